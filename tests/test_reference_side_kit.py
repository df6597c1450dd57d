"""The reference-side kit -- tests/golden/make_reference_fixtures.jl and the Julia shim in INTEGRATION.md -- is text that has
never been executed (no Julia in the build image).  This makes it fail HERE rather than on a maintainer's machine wherever a
static check can: every Terrarium identifier the two texts use (types, functions, qualified names, keyword arguments, struct
fields) must be defined under /root/reference/src with that name.  CPU only; skipped where the reference is absent (the GPU box),
so it never travels.  Earns no parity credit -- it only raises the odds that the kit works on first contact."""
import os
import re

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference sources are not on this machine")


def _strip(code):
    """Julia source without comments, docstrings and string literals (interpolations go with them)."""
    code = re.sub(r'"""(?:.|\n)*?"""', '""', code)
    code = re.sub(r"#=(?:.|\n)*?=#", "", code)
    out = []
    for line in code.split("\n"):
        line = re.sub(r'"(?:\\.|[^"\\])*"', '""', line)
        out.append(line.split("#", 1)[0])
    return "\n".join(out)


def _julia_blocks(markdown):
    return "\n".join(m.group(1) for m in re.finditer(r"```julia\n(.*?)```", markdown, flags=re.S))


@pytest.fixture(scope="module")
def reference():
    text = {}
    for base, _, files in os.walk(os.path.join(REF, "src")):
        for f in files:
            if f.endswith(".jl"):
                text[os.path.join(base, f)] = _strip(open(os.path.join(base, f), encoding="utf-8").read())
    for base, _, files in os.walk(os.path.join(REF, "ext")):
        for f in files:
            if f.endswith(".jl"):
                text[os.path.join(base, f)] = _strip(open(os.path.join(base, f), encoding="utf-8").read())
    return text


@pytest.fixture(scope="module")
def kit():
    fixtures = _strip(open(os.path.join(ROOT, "tests", "golden", "make_reference_fixtures.jl"), encoding="utf-8").read())
    shim = _strip(_julia_blocks(open(os.path.join(ROOT, "INTEGRATION.md"), encoding="utf-8").read()))
    return {"make_reference_fixtures.jl": fixtures, "INTEGRATION.md": shim}


IDENT = r"[A-Za-z_Ͱ-Ͽ∀-⋿][A-Za-z_0-9!Ͱ-Ͽ₀-₟ᵢ-ᵪ′]*"


def _short_form_definition(name, src):
    """`name(args...; kwargs...) [where {...}] = body` at the start of a line: the argument list may hold defaults (`=`), so the
    parentheses are matched, not skipped by a character class."""
    for m in re.finditer(rf"^[ \t]*(?:@\w+[ \t]+)*(?:\w+\.)*{re.escape(name)}(?:\{{[^}}\n]*\}})?\(", src, flags=re.M):
        depth, i = 1, m.end()
        while i < len(src) and depth:
            depth += src[i] == "("
            depth -= src[i] == ")"
            i += 1
        if re.match(r"\s*(?:where\s*(?:\{[^}]*\}|\w+)\s*)?=(?!=)", src[i:]):
            return True
    return False


def _defined_in_reference(name, reference):
    n = re.escape(name)
    pats = [rf"\bstruct\s+{n}\b", rf"\babstract\s+type\s+{n}\b", rf"\bfunction\s+(?:\w+\.)*{n}\b",
            rf"\bconst\s+{n}\b", rf"^\s*{n}\s*=(?!=)", rf"\bmacro\s+{n}\b", rf"@kernel\s+(?:inbounds\s*=\s*true\s+)?function\s+{n}\b"]
    rx = [re.compile(p, re.M) for p in pats]
    return [path for path, src in reference.items() if any(r.search(src) for r in rx) or _short_form_definition(name, src)]


def _exports(reference):
    names = set()
    for src in reference.values():
        for m in re.finditer(r"^\s*export\s+(.+(?:,\s*\n.+)*)", src, flags=re.M):
            names.update(x.strip() for x in m.group(1).replace("\n", " ").split(",") if x.strip())
    return names


# names the kit takes from Julia Base / Core / stdlib, Oceananigans, RingGrids, FreezeCurves, JSON (third parties are NOT on disk:
# SURVEY Appendix B; what the kit assumes of them is listed in INTEGRATION.md) and names the two texts define themselves
THIRD_PARTY = {
    "Oceananigans": {"interior", "set!", "CPU", "GPU", "Center", "Face", "Field", "FieldTimeSeries", "Clamp", "Cyclical", "Linear", "InMemory", "time_step!", "update_state!",
                     "OutputReaders", "TimeSteppers", "Simulations", "Grids", "Fields", "BoundaryConditions", "Value", "Flux", "Gradient", "architecture", "on_architecture",
                     "tick!", "Clock", "Simulation", "run!", "Units", "xnodes", "reset!"},
    "FreezeCurves": {"VanGenuchten", "BrooksCorey"},
    "RingGrids": {"RingGrids"},
    "JSON": {"JSON", "parsefile"},
    "Libdl": {"Libdl", "dlopen", "dlsym"},
}
BASE = set("""Float64 Float32 Int Int32 Int64 UInt32 UInt64 Bool Cint Cdouble Cvoid Cstring Cfloat Clonglong Culonglong Ptr Ref Vector Matrix Array AbstractArray
    Dict String Symbol Tuple NamedTuple Nothing Number Type Function Val Colon Any Union C_NULL NaN Inf nothing true false undef missing
    ccall unsafe_string isa hasproperty getproperty setproperty! propertynames length size reshape prod read! write open joinpath error merge get split collect
    vec pairs enumerate eachindex findall first last min max abs sum all any zip map filter push! copy copyto! fill! fill zeros ones similar eltype ndims
    convert reverse cumsum vcat hcat isnothing something typeof string println print pointer sizeof unsafe_wrap GC invoke include import using module
    view haskey keys values isempty iszero one zero floor ceil round div rem mod Base Core Main throw ErrorException ArgumentError
    @info @warn @assert @__DIR__ @inline @kwdef @eval @static @generated mutable struct const function end return for while if else elseif begin let do
    in where abstract type export global local try catch finally macro quote new continue break""".split())


def _local_definitions(code):
    names = set(re.findall(rf"\b(?:mutable\s+)?struct\s+({IDENT})", code))
    names |= set(re.findall(rf"\bconst\s+({IDENT})", code))
    names |= set(re.findall(rf"\bfunction\s+(?:\w+\.)*({IDENT})", code))
    names |= set(re.findall(rf"^\s*({IDENT})\s*\([^=\n]*\)\s*=(?!=)", code, flags=re.M))
    names |= set(re.findall(rf"\bmodule\s+({IDENT})", code))
    names |= set(re.findall(rf"\babstract\s+type\s+({IDENT})", code))
    names |= set(re.findall(rf"^\s*({IDENT})\s*=(?!=)", code, flags=re.M))          # local variables (some hold callables)
    names |= set(re.findall(rf"\bfor\s+\(?\s*({IDENT})", code))
    return names


def test_every_type_and_function_the_kit_names_exists_in_the_reference(kit, reference):
    exported = _exports(reference)
    third = set().union(*THIRD_PARTY.values())
    problems = []
    for where, code in kit.items():
        local = _local_definitions(code)
        used = set()
        used |= {m.group(1) for m in re.finditer(rf"\bTerrarium\.({IDENT})", code)}                      # qualified names
        used |= {m.group(1) for m in re.finditer(rf"(?<![\w.:@])([A-Z]{IDENT[1:]})\b", code)}            # Capitalised: types and constructors
        used |= {m.group(1) for m in re.finditer(rf"(?<![\w.:@])({IDENT})\(", code)}                      # calls
        # field accesses `x.name` and keyword names `name = ...` are checked by the curated tests below
        for name in sorted(used):
            if name in BASE or name in third or name in local or name.startswith(("trm_", "TRM_", "Trm")):
                continue
            if re.fullmatch(r"[A-Z][A-Z0-9_]*", name) and name in local | {"HERE", "INPUTS", "LIB", "NF", "FIELD", "FIELDS", "STATE_FIELD", "INPUT_FIELD"}:
                continue
            if len(name) == 1 or name in {"NF", "LX", "LY", "LZ"}:                                         # type parameters, loop variables
                continue
            if not _defined_in_reference(name, reference) and name not in exported:
                problems.append((where, name))
    assert not problems, "not defined under /root/reference/src: " + ", ".join(f"{n} ({w})" for w, n in problems)


def _signature_text(name, reference):
    """Every definition head of `name` in the reference (function form and assignment form), as text up to the closing parenthesis."""
    heads = []
    rx = re.compile(rf"(?:\bfunction\s+(?:\w+\.)*|^\s*(?:\w+\.)*){re.escape(name)}\s*(?:\{{[^}}]*\}})?\(", re.M)
    for src in reference.values():
        for m in rx.finditer(src):
            depth, i = 1, m.end()
            while i < len(src) and depth:
                depth += src[i] == "("
                depth -= src[i] == ")"
                i += 1
            heads.append(src[m.start():i])
    return heads


def _struct_fields(name, reference):
    for src in reference.values():
        m = re.search(rf"\bstruct\s+{re.escape(name)}\b[^\n]*\n(.*?)\n\s*end\b", src, flags=re.S)
        if m:
            return set(re.findall(rf"^\s*({IDENT})\s*(?:::|=|$)", m.group(1), flags=re.M))
    return None


# keyword arguments the kit passes -> (callable, where the kit uses it)
KEYWORDS = {
    "initialize": ["boundary_conditions"],                                  # model_integrator.jl:145-161
    "LandModel": ["soil", "vegetation"],                                    # land_model.jl
    "SoilModel": ["soil"],
    "SoilEnergyWaterCarbon": ["hydrology"],
    "SoilHydrology": ["hydraulic_properties"],
    "ConstantSoilHydraulics": ["swrc", "unsat_hydraulic_cond"],
    "PrescribedSpacing": ["Δz"],
    "ForwardEuler": ["Δt"],
}


def test_keyword_arguments_exist_with_that_name(reference):
    missing = []
    for name, kws in KEYWORDS.items():
        heads = _signature_text(name, reference)
        fields = _struct_fields(name, reference) or set()
        for kw in kws:
            in_head = any(re.search(rf";[^)]*\b{re.escape(kw)}\b", h, flags=re.S) for h in heads)
            if not in_head and kw not in fields:          # (@kwdef structs take their fields as keywords)
                missing.append(f"{name}(; {kw})")
    assert not missing, "keyword not found in any method / @kwdef struct of the reference: " + ", ".join(missing)


def test_positional_forms_the_kit_calls(reference):
    """The positional shapes: ColumnGrid(arch, NF, spacing, Nh), SoilHydrology(NF, RichardsEq(); ...), PrescribedSurfaceTemperature(name, value),
    GeothermalHeatFlux(value), InfiltrationFlux(value), merge_boundary_conditions(a, b), timestep!(integrator, dt), Terrarium.initialize!(state, model)."""
    def has_method(name, min_positional):
        for h in _signature_text(name, reference):
            args = h[h.index("(") + 1:-1].split(";")[0]
            depth, n, cur = 0, 0, ""
            for ch in args:
                depth += ch in "([{"
                depth -= ch in ")]}"
                if ch == "," and depth == 0:
                    n += bool(cur.strip()); cur = ""
                else:
                    cur += ch
            n += bool(cur.strip())
            if n >= min_positional or "..." in args:
                return True
        return False
    for name, n in (("ColumnGrid", 4), ("SoilHydrology", 2), ("PrescribedSurfaceTemperature", 2), ("GeothermalHeatFlux", 1), ("InfiltrationFlux", 1),
                    ("merge_boundary_conditions", 2), ("timestep!", 2), ("initialize!", 2)):
        assert has_method(name, n), f"{name}: no method of the reference takes {n} positional argument(s)"
    # LandModel(grid; ...) / SoilModel(grid; ...): the @kwdef structs take `grid` positionally through the generic constructor of
    # abstract_model.jl:218-223, `(::Type{Model})(grid::AbstractLandGrid, args...; kwargs...) where {Model <: AbstractModel} = Model(args...; grid, kwargs...)`
    generic = re.compile(r"\(::Type\{Model\}\)\(grid::AbstractLandGrid,\s*args\.\.\.;\s*kwargs\.\.\.\)\s*where\s*\{Model\s*<:\s*AbstractModel\}\s*=\s*Model\(args\.\.\.;\s*grid,\s*kwargs\.\.\.\)")
    assert any(generic.search(src) for src in reference.values())
    for model in ("LandModel", "SoilModel"):
        fields = _struct_fields(model, reference)
        assert fields is not None and "grid" in fields, model


def test_struct_fields_the_kit_reads(reference):
    """`integrator.state`, `state.inputs`, `integrator.clock`-like accesses of the two texts against the reference's struct definitions."""
    wanted = {"ModelIntegrator": {"state", "model", "timestepper", "inputs"}, "StateVariables": {"inputs", "clock", "prognostic", "tendencies", "auxiliary"}}
    for struct, fields in wanted.items():
        have = _struct_fields(struct, reference)
        assert have is not None, struct
        assert fields <= have, (struct, sorted(fields - have))


def test_state_variable_names_of_the_fixture_manifest_are_reference_variables(reference):
    """Every field / input name the fixture inputs carry is declared by the reference as a prognostic, auxiliary or input variable."""
    import json
    manifest = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_inputs", "manifest.json")))
    src = "\n".join(reference.values())
    declared = set(re.findall(r"(?:prognostic|auxiliary|input)\(\s*:([A-Za-z_0-9]+)", src))
    names = set()
    for case in manifest.values():
        names |= set(case["fields"]) | set(case["inputs"]) | set(case["compared"])
    unknown = sorted(n for n in names if n not in declared)
    assert not unknown, unknown


# what `params_from(model)` of the shim reads, property by property: struct -> fields
PARAMETER_PATHS = {
    "PhysicalConstants": {"ρw", "ρi", "ρₐ", "cₐ", "Lsl", "Llg", "Lsg", "g", "Tref", "σ", "κ", "ε", "Rₐ"},
    "LandModel": {"constants", "soil", "surface_energy_balance", "atmosphere", "surface_hydrology", "vegetation", "grid"},
    "SoilModel": {"constants", "soil", "grid"},
    "SoilEnergyWaterCarbon": {"energy", "strat", "biogeochem", "hydrology"},
    "SoilEnergyBalance": {"thermal_properties"},
    "SoilThermalProperties": {"conductivities", "heat_capacities"},
    "SoilThermalConductivities": {"water", "ice", "air", "mineral", "organic"},
    "SoilHeatCapacities": {"water", "ice", "air", "mineral", "organic"},
    "HomogeneousStratigraphy": {"porosity", "texture"},
    "ConstantSoilPorosity": {"mineral_porosity", "organic_porosity"},
    "ConstantSoilCarbonDensity": {"ρ_soc", "ρ_org"},
    "SoilHydrology": {"hydraulic_properties", "vertical_flow", "vwc_forcing"},
    "ConstantSoilHydraulics": {"sat_hydraulic_cond", "swrc", "unsat_hydraulic_cond"},
    "UnsatKVanGenuchten": {"impedance"},
    "SurfaceEnergyBalance": {"albedo", "skin_temperature"},
    "ConstantAlbedo": {"albedo", "emissivity"},
    "ImplicitSkinTemperature": {"κₛ"},
    "PrescribedAtmosphere": {"aerodynamics", "min_windspeed"},
    "ConstantAerodynamics": {"Cₕ"},
    "SurfaceHydrology": {"surface_runoff", "evapotranspiration"},
    "DirectSurfaceRunoff": {"τ_r"},
    "BareGroundEvaporation": {"ground_resistance"},
    "ConstantEvaporationResistanceFactor": {"factor"},
}


def test_parameter_paths_of_the_shim_exist(kit, reference):
    problems = []
    for struct, fields in PARAMETER_PATHS.items():
        have = _struct_fields(struct, reference)
        if have is None:
            problems.append(f"struct {struct} not found")
        elif not fields <= have:
            problems.append(f"{struct}: no field(s) {sorted(fields - have)} (has {sorted(have)})")
    assert not problems, "; ".join(problems)
    # ... and every `.property` the shim's params_from reads is one of the fields above (a new access must be added to the table)
    shim = kit["INTEGRATION.md"]
    body = shim[shim.index("function params_from(model)"):]
    body = body[:body.index("\nend")]
    known = set().union(*PARAMETER_PATHS.values())
    read = set(re.findall(rf"(?<=[\w\]\)])\.({IDENT})", body)) - {"p"}
    written = set(re.findall(rf"\bp\.({IDENT})", body))                   # the fields of TrmParams the shim assigns
    unknown = sorted(n for n in read - written if n not in known and n not in {"RichardsEq", "VanGenuchten", "UnsatKVanGenuchten", "LandModel", "PrescribedAlbedo",
                                                                             "SoilMoistureResistanceFactor", "field_capacity", "α", "n", "ψₛ", "λ"})
    assert not unknown, unknown
