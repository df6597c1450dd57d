# make_reference_fixtures.jl -- REFERENCE-SIDE fixture generator.  NEVER EXECUTED IN THIS REPOSITORY'S BUILD ENVIRONMENT (there is
# no Julia in the image): it is written against the public interface of TUM-PIK-ESM/Terrarium.jl as read from its sources and is
# the one route to pin what the reference's own tests do not -- the BrooksCorey default, `compute_z_bcs!`'s sign and Az / V scale,
# Oceananigans' reciprocal-spacing operators, the Value-boundary halo (DESIGN.md section 2, "parity unpinned").  It earns no parity
# credit until someone runs it; a maintainer with Terrarium.jl installed does
#
#     julia --project=<Terrarium.jl checkout> tests/golden/make_reference_fixtures.jl
#
# which reads the synthetic cases tests/golden/make_reference_inputs.py wrote (tests/golden/reference_inputs/: raw little-endian
# Float64 arrays + manifest.json -- BASELINE configs C1 - C4 on a handful of columns), sets each case up as the reference's own
# tests and examples do (test/coupled_models/land_model_tests.jl:6-36, test/soil/soil_hydrology_tests.jl:152-188,
# examples/simulations/soil_heat_global.jl:44-100), steps it, and writes after each number of steps listed in the manifest
#
#     tests/golden/reference_<case>__<field>__<steps>.bin     raw little-endian Float64, [rows][Nh], row 0 = bottom layer
#
# tests/test_golden_vectors.py picks those files up when they exist (oracle on the CPU, HIP library on the GPU; skipped otherwise).
using Terrarium
using Oceananigans: interior, set!
using JSON                    # manifest only

const HERE = @__DIR__
const INPUTS = joinpath(HERE, "reference_inputs")
manifest = JSON.parsefile(joinpath(INPUTS, "manifest.json"))

"raw [rows][Nh] (row-major, row 0 = bottom layer) is, read column-major, the (Nh, rows) matrix interior(field)[:, 1, :]"
function read_array(entry)
    shape = Int.(entry["shape"])
    data = Vector{Float64}(undef, prod(shape))
    read!(joinpath(INPUTS, entry["file"]), data)
    return length(shape) == 1 ? data : reshape(data, shape[2], shape[1])
end

"set!(field, array (Nh, rows)) for a 3-D field, (Nh,) for a 2-D one"
function set_from_array!(field, a)
    if a isa Vector
        interior(field)[:, 1, 1] .= a
    else
        interior(field)[:, 1, :] .= a
    end
end

function write_field(path, field)
    a = Array(interior(field))[:, 1, :]                      # (Nh, rows); written column-major = row-major [rows][Nh]
    open(path, "w") do io
        write(io, Float64.(a))
    end
end

for (name, case) in manifest
    NF = Float64
    Nz, Nh, dt = Int(case["Nz"]), Int(case["Nh"]), Float64(case["dt"])
    thickness = read_array(case["thickness"])                 # index 1 = surface layer (get_spacing)
    grid = ColumnGrid(CPU(), NF, PrescribedSpacing(Δz = thickness), Nh)
    params = case["params"]
    richards = get(params, "flow", 0) == 1
    # hydraulics: the reference defaults (BrooksCorey + UnsatKLinear) or the variant of every reference test (soil_hydrology_tests.jl:127-129)
    hydraulic_properties = get(params, "swrc", 0) == 1 ?
        ConstantSoilHydraulics(NF; swrc = VanGenuchten(α = params["vg_alpha"], n = params["vg_n"]), unsat_hydraulic_cond = UnsatKVanGenuchten(NF)) :
        ConstantSoilHydraulics(NF)
    hydrology = richards ? SoilHydrology(NF, RichardsEq(); hydraulic_properties) : SoilHydrology(NF; hydraulic_properties)
    soil = SoilEnergyWaterCarbon(NF; hydrology)
    land = get(params, "seb", 0) == 1
    model = land ? LandModel(grid; soil, vegetation = nothing) : SoilModel(grid; soil)

    # boundary conditions: per-column arrays as Fields of the grid (src/models/soil/soil_model_bcs.jl)
    bcs = (;)
    bc_arrays = Dict{String, Any}()
    for (key, bc) in case["bcs"]
        var, side = split(key, ":")
        values = read_array(bc)
        bc_arrays[key] = values
        f = Terrarium.Field(grid, XY())                      # a 2-D field holding the per-column boundary values
        set_from_array!(f, values)
        if var == "temperature" && side == "top" && bc["kind"] == "value"
            bcs = merge_boundary_conditions(bcs, PrescribedSurfaceTemperature(:T_ub, f))
        elseif var == "internal_energy" && side == "bottom" && bc["kind"] == "flux"
            bcs = merge_boundary_conditions(bcs, GeothermalHeatFlux(f))
        elseif var == "saturation_water_ice" && side == "top" && bc["kind"] == "flux"
            bcs = merge_boundary_conditions(bcs, InfiltrationFlux(f))
        else
            error("boundary condition $key ($(bc["kind"])) has no alias in this script: extend it")
        end
    end

    integrator = initialize(model, ForwardEuler(NF; Δt = dt); boundary_conditions = bcs)
    state = integrator.state
    # initial fields and inputs AS ARRAYS (the same numbers the HIP library and the oracle receive), then the process initialisers
    for (fname, entry) in case["fields"]
        set_from_array!(getproperty(state, Symbol(fname)), read_array(entry))
    end
    for (iname, entry) in case["inputs"]
        set_from_array!(getproperty(state.inputs, Symbol(iname)), read_array(entry))
    end
    Terrarium.initialize!(state, model)                      # soil_coupled.jl:45-54: hydraulics, water table, sat -> psi, T -> U

    done = 0
    for nsteps in Int.(case["steps"])
        while done < nsteps
            timestep!(integrator, dt)
            done += 1
        end
        for fname in case["compared"]
            hasproperty(state, Symbol(fname)) || continue
            write_field(joinpath(HERE, "reference_$(name)__$(fname)__$(nsteps).bin"), getproperty(state, Symbol(fname)))
        end
    end
    @info "wrote reference fixtures" name steps = case["steps"]
end
