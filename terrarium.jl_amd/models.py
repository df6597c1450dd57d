"""Model configuration structs -- host-side mirror of the reference's immutable
`@kwdef` parameter structs on the SoilModel hot path.  They
hold no arrays; `flatten()` turns a model into the flat `trm_params` POD of
include/terrarium_hip.h.  Names and defaults follow the reference."""
import numpy as np
from dataclasses import dataclass, field
from typing import Optional, Union

from . import _capi
from .grids import ColumnGrid


# ---- src/processes/physical_constants.jl:9-51 -------------------------------
@dataclass
class PhysicalConstants:
    rho_w: float = 1000.0
    rho_i: float = 916.2
    rho_a: float = 1.293
    c_a: float = 1005.7
    Lsl: float = 3.34e5
    Llg: float = 2.257e6
    Lsg: float = 2.834e6
    g: float = 9.80665
    Tref: float = 273.15
    sigma: float = 5.6704e-8
    kappa: float = 0.4
    epsilon: float = 0.622
    R_a: float = 287.058
    C_mass: float = 12.0


# ---- src/processes/soil/energy/soil_thermal_properties.jl:14-46 -------------
@dataclass
class SoilThermalConductivities:
    water: float = 0.57
    ice: float = 2.2
    air: float = 0.025
    mineral: float = 3.8
    organic: float = 0.25


@dataclass
class SoilHeatCapacities:
    water: float = 4.2e6
    ice: float = 1.9e6
    air: float = 0.00125e6
    mineral: float = 2.0e6
    organic: float = 2.5e6


@dataclass
class SoilThermalProperties:
    """Bulk conductivity = InverseQuadratic, freeze curve = FreeWater (the only
    ones the reference implements, soil_thermal_properties.jl:80-83,119-123)."""
    conductivities: SoilThermalConductivities = field(default_factory=SoilThermalConductivities)
    heat_capacities: SoilHeatCapacities = field(default_factory=SoilHeatCapacities)


@dataclass
class SoilEnergyBalance:
    """src/processes/soil/energy/soil_energy.jl:22-43 (ExplicitTwoPhaseHeatConduction,
    SoilEnergyTemperatureClosure)."""
    thermal_properties: SoilThermalProperties = field(default_factory=SoilThermalProperties)


# ---- stratigraphy / biogeochemistry -----------------------------------------
@dataclass
class ConstantSoilPorosity:
    """soil_porosity.jl:7-13"""
    mineral_porosity: float = 0.49
    organic_porosity: float = 0.9


@dataclass
class SoilPorositySURFEX:
    """soil_porosity.jl:30-50 (Noilhan & Mahfouf 1996): mineral porosity = porosity_default + porosity_sand_coef * sand.
    (The reference's `organic_porosity` method for this type reads an undefined variable, soil_porosity.jl:43, so the type
    cannot be evaluated there; the evident intent -- `porosity_organic` -- is what `flatten` uses.)"""
    porosity_default: float = 0.49
    porosity_sand_coef: float = -0.11
    porosity_organic: float = 0.9

    def mineral(self, texture):
        return self.porosity_default + self.porosity_sand_coef * texture.sand


@dataclass
class SoilTexture:
    """src/processes/soil/stratigraphy/soil_texture.jl:6-20: fractional mixture of sand, silt and clay."""
    sand: float = 1.0
    clay: float = 0.0
    silt: Optional[float] = None

    def __post_init__(self):
        if self.silt is None:
            self.silt = 1 - self.sand - self.clay
        for name in ("sand", "silt", "clay"):
            if not (0.0 <= getattr(self, name) <= 1.0):
                raise AssertionError(f"{name} fraction must lie in [0, 1]")
        if abs(self.sand + self.silt + self.clay - 1.0) > 1.5e-8:      # Julia's `≈` (rtol = sqrt(eps))
            raise AssertionError("sand, silt, and clay fractions must sum to unity")


@dataclass
class HomogeneousStratigraphy:
    """homogeneous_strat.jl:9-15 (the texture only matters to SURFEX field capacity / wilting point, off the step path)."""
    porosity: Union[ConstantSoilPorosity, "SoilPorositySURFEX"] = field(default_factory=ConstantSoilPorosity)
    texture: SoilTexture = field(default_factory=SoilTexture)


@dataclass
class ConstantSoilCarbonDensity:
    """constant_soil_carbon.jl:10-16"""
    rho_soc: float = 0.0
    rho_org: float = 1300.0


# ---- hydrology ---------------------------------------------------------------
@dataclass
class BrooksCorey:
    """FreezeCurves.BrooksCorey(psi_s = 0.01 m, lambda = 0.2); theta_res = 0."""
    psi_s: float = 0.01
    lam: float = 0.2
    theta_res: float = 0.0


@dataclass
class VanGenuchten:
    """FreezeCurves.VanGenuchten(alpha = 1/m, n = 2); theta_res = 0."""
    alpha: float = 1.0
    n: float = 2.0
    theta_res: float = 0.0


@dataclass
class UnsatKLinear:
    """soil_hydraulic_properties.jl:166-181"""


@dataclass
class UnsatKVanGenuchten:
    """soil_hydraulic_properties.jl:196-221"""
    impedance: float = 7.0


@dataclass
class ConstantSoilHydraulics:
    """soil_hydraulic_properties.jl:66-97: prescribed saturated conductivity, field capacity and wilting point."""
    swrc: Union[BrooksCorey, VanGenuchten] = field(default_factory=BrooksCorey)
    unsat_hydraulic_cond: Union[UnsatKLinear, UnsatKVanGenuchten] = field(default_factory=UnsatKLinear)
    sat_hydraulic_cond: float = 1.0e-5
    field_capacity_value: float = 0.25
    wilting_point_value: float = 0.05

    def saturated_hydraulic_conductivity(self, *args):
        return self.sat_hydraulic_cond

    def wilting_point(self, *args):
        return self.wilting_point_value

    def field_capacity(self, *args):
        return self.field_capacity_value


@dataclass
class SoilHydraulicsSURFEX:
    """soil_hydraulic_properties.jl:112-156: field capacity and wilting point from the clay content (Noilhan & Mahfouf
    1996, eqs. 28-29).  The step itself reads the SWRC, the unsaturated-K scheme and K_sat, exactly as for
    ConstantSoilHydraulics; field capacity / wilting point are consumed by the vegetation processes only."""
    swrc: Union[BrooksCorey, VanGenuchten] = field(default_factory=BrooksCorey)
    unsat_hydraulic_cond: Union[UnsatKLinear, UnsatKVanGenuchten] = field(default_factory=UnsatKLinear)
    sat_hydraulic_cond: float = 1.0e-5
    wilting_point_coef: float = 37.13e-3
    field_capacity_coef: float = 89.0e-3
    field_capacity_exp: float = 0.35

    def saturated_hydraulic_conductivity(self, *args):
        return self.sat_hydraulic_cond

    def wilting_point(self, texture: SoilTexture):
        return self.wilting_point_coef * (texture.clay * 100) ** 0.5

    def field_capacity(self, texture: SoilTexture):
        return self.field_capacity_coef * (texture.clay * 100) ** self.field_capacity_exp


@dataclass
class NoFlow:
    """soil_hydrology.jl:14"""


@dataclass
class RichardsEq:
    """soil_hydrology_rre.jl:18"""


@dataclass
class SoilHydrology:
    """soil_hydrology.jl:21-53.  `vwc_forcing` [1/s]: a number (spatially constant source/sink), a per-cell array
    `[Nz][Nh]` / vertical profile `[Nz]`, a function (x, z) evaluated once per cell (an Oceananigans `Forcing`
    that depends on position only), or a `StateFunction(func(fields, clock, parameters))` -- a forcing that depends on
    the evolving fields (forcings.jl:13-15), evaluated on the device before every step into the per-cell forcing field."""
    vertical_flow: Union[NoFlow, RichardsEq] = field(default_factory=NoFlow)
    hydraulic_properties: Union[ConstantSoilHydraulics, SoilHydraulicsSURFEX] = field(default_factory=ConstantSoilHydraulics)
    vwc_forcing: Optional[object] = None


@dataclass
class SoilEnergyWaterCarbon:
    """soil_coupled.jl:7-39"""
    strat: HomogeneousStratigraphy = field(default_factory=HomogeneousStratigraphy)
    energy: SoilEnergyBalance = field(default_factory=SoilEnergyBalance)
    hydrology: SoilHydrology = field(default_factory=SoilHydrology)
    biogeochem: ConstantSoilCarbonDensity = field(default_factory=ConstantSoilCarbonDensity)


# ---- surface energy balance / atmosphere / surface hydrology -----------------
@dataclass
class ConstantAlbedo:
    """albedo.jl:22-28"""
    albedo: float = 0.3
    emissivity: float = 0.97


@dataclass
class PrescribedAlbedo:
    """albedo.jl:8-14: `albedo` and `emissivity` are 2-D inputs, set like any other input (`inputs = dict(albedo = ...,
    emissivity = ...)`, a number, a per-column array, a FieldTimeSeries or a raster)."""


@dataclass
class ImplicitSkinTemperature:
    """skin_temperature.jl:49-52"""
    kappa_s: float = 2.0


@dataclass
class SurfaceEnergyBalance:
    """surface_energy_balance.jl:9-37 (DiagnosedRadiativeFluxes, DiagnosedTurbulentFluxes)."""
    skin_temperature: ImplicitSkinTemperature = field(default_factory=ImplicitSkinTemperature)
    albedo: Union[ConstantAlbedo, PrescribedAlbedo] = field(default_factory=ConstantAlbedo)


@dataclass
class ConstantAerodynamics:
    """aerodynamics.jl:6-9"""
    C_h: float = 1.2e-3


@dataclass
class PrescribedAtmosphere:
    """prescribed_atmosphere.jl:45-99"""
    altitude: float = 10.0
    min_windspeed: float = 0.01
    aerodynamics: ConstantAerodynamics = field(default_factory=ConstantAerodynamics)


@dataclass
class DirectSurfaceRunoff:
    """direct_surface_runoff.jl:15-18"""
    tau_r: float = 3600.0


@dataclass
class ConstantEvaporationResistanceFactor:
    """ground_resistance_factor.jl:6-12"""
    factor: float = 1.0


@dataclass
class SoilMoistureResistanceFactor:
    """ground_resistance_factor.jl:14-56 (Lee & Pielke 1992): beta = (1 - cos(pi theta_1 / theta_fc))^2 / 4 below the field
    capacity of the soil's hydraulic properties, 1 above."""


@dataclass
class BareGroundEvaporation:
    """bare_ground_evaporation.jl:12-24: E = beta dq / r_a with the ground resistance factor beta."""
    ground_resistance: Union[ConstantEvaporationResistanceFactor, SoilMoistureResistanceFactor] = field(default_factory=ConstantEvaporationResistanceFactor)

    @property
    def factor(self):
        return getattr(self.ground_resistance, "factor", 1.0)


@dataclass
class NoCanopyInterception:
    """canopy_interception.jl:1-23: all the rain reaches the ground."""


@dataclass
class PALADYNCanopyInterception:
    """canopy_interception.jl:37-49: interception I = alpha_int P (1 - exp(-k_ext (LAI + SAI))), a canopy water store of
    capacity w_can_max (LAI + SAI) emptied over tau_w."""
    alpha_int: float = 0.2
    k_ext: float = 0.5
    w_can_max: float = 2.0e-4
    tau_w: float = 86400.0


@dataclass
class PALADYNCanopyEvapotranspiration:
    """canopy_evapotranspiration.jl:25-40: transpiration, ground evaporation below the canopy, evaporation of intercepted water."""
    C_can: float = 0.006
    ground_resistance: Union[ConstantEvaporationResistanceFactor, SoilMoistureResistanceFactor] = field(default_factory=ConstantEvaporationResistanceFactor)

    @property
    def factor(self):
        return getattr(self.ground_resistance, "factor", 1.0)


@dataclass
class SurfaceHydrology:
    """surface_hydrology.jl:10-34.  Defaults: the `vegetation = nothing` configuration (land_model.jl:119-125),
    BareGroundEvaporation + NoCanopyInterception; `SurfaceHydrology.canopy()` gives the reference's own default
    (PALADYNCanopyInterception + PALADYNCanopyEvapotranspiration, surface_hydrology.jl:26-33) that a LandModel with
    vegetation takes."""
    evapotranspiration: Union[BareGroundEvaporation, PALADYNCanopyEvapotranspiration] = field(default_factory=BareGroundEvaporation)
    surface_runoff: DirectSurfaceRunoff = field(default_factory=DirectSurfaceRunoff)
    canopy_interception: Union[NoCanopyInterception, PALADYNCanopyInterception] = field(default_factory=NoCanopyInterception)

    @classmethod
    def canopy(cls, **kw):
        return cls(evapotranspiration=kw.pop("evapotranspiration", PALADYNCanopyEvapotranspiration()),
                   canopy_interception=kw.pop("canopy_interception", PALADYNCanopyInterception()), **kw)


# ---- initialisers (src/models/soil/soil_model_init.jl) ------------------------
@dataclass
class DefaultInitializer:
    """initializers.jl:29-34: leave every Field at its default (zero)."""


@dataclass
class ConstantSoilTemperature:
    T0: float = 0.0


@dataclass
class QuasiThermalSteadyState:
    T0: float = 0.0
    Qgeo: float = 0.02
    k_eff: float = 1.0


def piecewise_linear(*knots):
    """piecewise_linear(knots...) (src/utils/interpolation_utils.jl:6-14): f(z) linear between the (z, value) knots, given with
    z in DESCENDING order, flat beyond the ends."""
    zs = [float(k[0]) for k in knots]
    ys = [float(k[1]) for k in knots]
    if any(b > a for a, b in zip(zs, zs[1:])):
        raise ValueError("depths must be sorted in descending order")
    zr, yr = np.array(zs[::-1]), np.array(ys[::-1])
    return lambda z: np.interp(z, zr, yr)


@dataclass
class PiecewiseLinearInitialSoilTemperature:
    """soil_model_init.jl:86-114: temperature (degC) from (depth, value) knots, z in descending order."""
    knots: tuple = ()

    def __init__(self, *knots):
        self.knots = tuple(knots[0]) if len(knots) == 1 and knots and isinstance(knots[0][0], (tuple, list)) else tuple(knots)


@dataclass
class ConstantSaturation:
    sat: float = 1.0


@dataclass
class SaturationWaterTable:
    vadose_zone_saturation: float = 0.5
    water_table_depth: float = 5.0


@dataclass
class SoilInitializer:
    energy: object = field(default_factory=QuasiThermalSteadyState)
    hydrology: object = field(default_factory=SaturationWaterTable)


# ---- models -------------------------------------------------------------------
@dataclass
class SoilModel:
    """src/models/soil/soil_model.jl:9-27"""
    grid: ColumnGrid
    soil: SoilEnergyWaterCarbon = field(default_factory=SoilEnergyWaterCarbon)
    constants: PhysicalConstants = field(default_factory=PhysicalConstants)
    initializer: object = field(default_factory=DefaultInitializer)
    halo_policy: str = "reference_zero"

    coupled_surface = False


@dataclass
class LandModel:
    """src/models/coupled/land_model.jl:10-44: soil + surface energy balance + surface hydrology + PrescribedAtmosphere,
    ground_heat_flux / -infiltration wired as the soil's top flux BCs (:46-66).  `vegetation = None` (the default HERE; the
    reference's is VegetationCarbon) is the bare-ground configuration of the hot path; with `vegetation = VegetationCarbon()`
    the defaults follow land_model.jl:108-125: Richards soil, canopy interception and canopy evapotranspiration."""
    grid: ColumnGrid
    soil: SoilEnergyWaterCarbon = None
    surface_energy_balance: SurfaceEnergyBalance = field(default_factory=SurfaceEnergyBalance)
    surface_hydrology: SurfaceHydrology = None
    atmosphere: PrescribedAtmosphere = field(default_factory=PrescribedAtmosphere)
    constants: PhysicalConstants = field(default_factory=PhysicalConstants)
    initializer: object = field(default_factory=DefaultInitializer)
    halo_policy: str = "reference_zero"
    vegetation: object = None

    coupled_surface = True

    def __post_init__(self):
        if self.soil is None:      # default_soil (land_model.jl:108-109)
            self.soil = SoilEnergyWaterCarbon() if self.vegetation is None else SoilEnergyWaterCarbon(hydrology=SoilHydrology(vertical_flow=RichardsEq()))
        if self.surface_hydrology is None:   # default_surface_hydrology (land_model.jl:115-125)
            self.surface_hydrology = SurfaceHydrology() if self.vegetation is None else SurfaceHydrology.canopy()
        canopy = isinstance(self.surface_hydrology.evapotranspiration, PALADYNCanopyEvapotranspiration)
        if canopy != (self.vegetation is not None) or canopy != isinstance(self.surface_hydrology.canopy_interception, PALADYNCanopyInterception):
            raise ValueError("LandModel: canopy interception / canopy evapotranspiration go with a vegetation scheme, "
                             "bare-ground evaporation without one")


# ---- vegetation (src/processes/vegetation/, needleleaf-tree PFT defaults) ----------------------------------------------
@dataclass
class LUEPhotosynthesis:
    """photosynthesis.jl:17-68"""
    tau25: float = 2600.0
    Kc25: float = 30.0
    Ko25: float = 3.0e4
    q10_tau: float = 0.57
    q10_Kc: float = 2.1
    q10_Ko: float = 1.2
    alpha_leaf: float = 0.17
    alpha_a: float = 0.5
    alpha_C3: float = 0.08
    cq: float = 4.6e-6
    k_ext: float = 0.5
    T_CO2_high: float = 42.0
    T_CO2_low: float = -4.0
    T_photos_high: float = 30.0
    T_photos_low: float = 15.0
    theta_r: float = 0.7


@dataclass
class MedlynStomatalConductance:
    """stomatal_conductance.jl:16-24"""
    g1: float = 2.3
    g_min: float = 0.5


@dataclass
class PALADYNAutotrophicRespiration:
    """autotrophic_respiration.jl:14-23"""
    cn_sapwood: float = 330.0
    cn_root: float = 29.0
    aws: float = 10.0


@dataclass
class PALADYNPhenology:
    """phenology.jl:14-18 (evergreen placeholder: f_deciduous = 0, phenology factor 1)"""


@dataclass
class PALADYNCarbonDynamics:
    """carbon_dynamics.jl:18-43"""
    SLA: float = 10.0
    awl: float = 2.0
    LAI_min: float = 1.0
    LAI_max: float = 6.0
    gamma_L: float = 0.3
    gamma_R: float = 0.3
    gamma_S: float = 0.05


@dataclass
class PALADYNVegetationDynamics:
    """vegetation_dynamics.jl:15-22"""
    nu_seed: float = 0.001
    gamma_v_min: float = 0.002


@dataclass
class StaticExponentialRootDistribution:
    """root_distribution.jl:23-29"""
    a: float = 7.0
    b: float = 2.0


@dataclass
class FieldCapacityLimitedPAW:
    """plant_available_water.jl:17-19"""


@dataclass
class VegetationCarbon:
    """vegetation_carbon.jl:6-64"""
    photosynthesis: LUEPhotosynthesis = field(default_factory=LUEPhotosynthesis)
    stomatal_conductance: MedlynStomatalConductance = field(default_factory=MedlynStomatalConductance)
    autotrophic_respiration: PALADYNAutotrophicRespiration = field(default_factory=PALADYNAutotrophicRespiration)
    phenology: PALADYNPhenology = field(default_factory=PALADYNPhenology)
    carbon_dynamics: PALADYNCarbonDynamics = field(default_factory=PALADYNCarbonDynamics)
    vegetation_dynamics: PALADYNVegetationDynamics = field(default_factory=PALADYNVegetationDynamics)
    root_distribution: StaticExponentialRootDistribution = field(default_factory=StaticExponentialRootDistribution)
    plant_available_water: FieldCapacityLimitedPAW = field(default_factory=FieldCapacityLimitedPAW)


@dataclass
class VegetationModel:
    """src/models/vegetation/vegetation_model.jl:13-32: natural vegetation of a single plant functional type, driven by
    the prescribed atmosphere alone (soil moisture limitation and ground temperature are inputs)."""
    grid: ColumnGrid
    atmosphere: PrescribedAtmosphere = field(default_factory=PrescribedAtmosphere)
    vegetation: VegetationCarbon = field(default_factory=VegetationCarbon)
    constants: PhysicalConstants = field(default_factory=PhysicalConstants)
    initializer: object = field(default_factory=DefaultInitializer)
    halo_policy: str = "reference_zero"
    soil: SoilEnergyWaterCarbon = field(default_factory=SoilEnergyWaterCarbon)   # (the context's unused soil columns)


def flatten_vegetation(veg: VegetationCarbon, constants: PhysicalConstants = None, hydraulics=None, texture=None,
                       surface_hydrology=None) -> "_capi.TrmVegetationParams":
    """VegetationCarbon -> trm_vegetation_params (include/terrarium_hip.h)."""
    p = _capi.default_vegetation_params()
    for part in (veg.photosynthesis, veg.stomatal_conductance, veg.autotrophic_respiration, veg.carbon_dynamics, veg.vegetation_dynamics):
        for k, v in vars(part).items():
            setattr(p, k, v)
    p.root_a, p.root_b = veg.root_distribution.a, veg.root_distribution.b
    if hydraulics is not None:
        tex = texture or SoilTexture()
        p.wilting_point, p.field_capacity = hydraulics.wilting_point(tex), hydraulics.field_capacity(tex)
    if constants is not None:
        p.C_mass = getattr(constants, "C_mass", 12.0)
    if surface_hydrology is not None:
        ci, et = surface_hydrology.canopy_interception, surface_hydrology.evapotranspiration
        p.alpha_int, p.canopy_k_ext, p.w_can_max, p.tau_w = ci.alpha_int, ci.k_ext, ci.w_can_max, ci.tau_w
        p.C_can = et.C_can
    return p


def flatten(model) -> "_capi.TrmParams":
    """Model structs -> trm_params (include/terrarium_hip.h)."""
    p = _capi.TrmParams()
    c = model.constants
    p.rho_w, p.rho_i, p.rho_a, p.c_a = c.rho_w, c.rho_i, c.rho_a, c.c_a
    p.Lsl, p.Llg, p.Lsg, p.g, p.Tref, p.sigma = c.Lsl, c.Llg, c.Lsg, c.g, c.Tref, c.sigma
    p.kappa_vk, p.eps_mw, p.R_a = c.kappa, c.epsilon, c.R_a
    soil = model.soil
    k, h = soil.energy.thermal_properties.conductivities, soil.energy.thermal_properties.heat_capacities
    p.k_water, p.k_ice, p.k_air, p.k_mineral, p.k_organic = k.water, k.ice, k.air, k.mineral, k.organic
    p.c_water, p.c_ice, p.c_air, p.c_mineral, p.c_organic = h.water, h.ice, h.air, h.mineral, h.organic
    if isinstance(soil.strat.porosity, SoilPorositySURFEX):
        p.por_mineral = soil.strat.porosity.mineral(soil.strat.texture)
        p.por_organic = soil.strat.porosity.porosity_organic
    else:
        p.por_mineral = soil.strat.porosity.mineral_porosity
        p.por_organic = soil.strat.porosity.organic_porosity
    p.rho_soc, p.rho_org = soil.biogeochem.rho_soc, soil.biogeochem.rho_org
    hyd = soil.hydrology
    hp = hyd.hydraulic_properties
    p.K_sat = hp.sat_hydraulic_cond
    # defaults of the unused SWRC keep the struct fully defined
    p.bc_psi_s, p.bc_lambda, p.vg_alpha, p.vg_n, p.theta_res = 0.01, 0.2, 1.0, 2.0, 0.0
    if isinstance(hp.swrc, VanGenuchten):
        p.swrc = _capi.SWRC["van_genuchten"]
        p.vg_alpha, p.vg_n, p.theta_res = hp.swrc.alpha, hp.swrc.n, hp.swrc.theta_res
    else:
        p.swrc = _capi.SWRC["brooks_corey"]
        p.bc_psi_s, p.bc_lambda, p.theta_res = hp.swrc.psi_s, hp.swrc.lam, hp.swrc.theta_res
    p.impedance = 7.0
    if isinstance(hp.unsat_hydraulic_cond, UnsatKVanGenuchten):
        if not isinstance(hp.swrc, VanGenuchten):
            raise ValueError("UnsatKVanGenuchten requires a VanGenuchten SWRC (soil_hydraulic_properties.jl:203-206)")
        p.unsat_k = _capi.UNSATK["van_genuchten"]
        p.impedance = hp.unsat_hydraulic_cond.impedance
    else:
        p.unsat_k = _capi.UNSATK["linear"]
    p.flow = _capi.FLOW["richards"] if isinstance(hyd.vertical_flow, RichardsEq) else _capi.FLOW["noflow"]
    p.vwc_forcing = float(hyd.vwc_forcing) if isinstance(hyd.vwc_forcing, (int, float)) else 0.0  # arrays: uploaded by initialize
    # surface defaults (used only when seb = 1)
    p.albedo, p.emissivity, p.kappa_s, p.C_h = 0.3, 0.97, 2.0, 1.2e-3
    p.min_windspeed, p.tau_r, p.beta_evap, p.field_capacity = 0.01, 3600.0, 1.0, 0.25
    p.seb = 0
    if getattr(model, "coupled_surface", False):
        p.seb = 1
        seb = model.surface_energy_balance
        if isinstance(seb.albedo, PrescribedAlbedo):
            p.prescribed_albedo = 1
        else:
            p.albedo, p.emissivity = seb.albedo.albedo, seb.albedo.emissivity
        p.kappa_s = seb.skin_temperature.kappa_s
        p.C_h = model.atmosphere.aerodynamics.C_h
        p.min_windspeed = model.atmosphere.min_windspeed
        p.tau_r = model.surface_hydrology.surface_runoff.tau_r
        evap = model.surface_hydrology.evapotranspiration
        p.beta_evap = evap.factor
        if isinstance(evap.ground_resistance, SoilMoistureResistanceFactor):
            p.evap_resistance = 1
        p.field_capacity = hp.field_capacity(soil.strat.texture)
    p.halo_policy = _capi.HALO[model.halo_policy]
    return p
