/* terrarium_hip.h -- C ABI of libterrarium_hip.so
 *
 * MI355X-native (gfx950, hand-written HIP) implementation of ONE hot path of
 * TUM-PIK-ESM/Terrarium.jl: the explicit time step of SoilModel /
 * LandModel(vegetation = nothing) over laterally independent soil columns --
 * heat conduction with freeze/thaw, Richards water transport, bare-ground
 * surface energy balance, forward-Euler / Heun update and the closures.
 *
 * The reference is pure Julia and has NO FFI: its extension surface is multiple
 * dispatch on (state, model) (src/abstract_model.jl:52-95,175-215) and the
 * time-stepper hook timestep!(integrator, ::AbstractTimeStepper, dt)
 * (src/timesteppers/abstract_timestepper.jl:39).  Every entry point below names
 * the reference method it stands in for; INTEGRATION.md shows the ~200-line
 * Julia shim (`ccall`) a maintainer would add on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success, a TRM_E* code otherwise; the message
 *     is available from trm_last_error(ctx).  Nothing throws across the ABI.
 *   - the context owns all device memory; host pointers are borrowed for the
 *     duration of a call.  Calls are synchronous on return EXCEPT trm_step /
 *     trm_step_heun / the compute_* family when TRM_OPT_ASYNC is set.
 *   - one host thread drives a context; contexts are independent; no globals
 *     (reference rule AGENTS.md:44).
 *   - host arrays are [rows][num_columns], x (column) fastest, NO halos, row 0 =
 *     BOTTOM layer, row Nz-1 = surface layer: exactly `interior(field)` of the
 *     reference's (Nh, 1, Nz) Fields (src/grids/column_grid.jl:27-32).  Element
 *     type is double for TRM_F64 contexts and float for TRM_F32.
 *   - there is no CPU fallback: every entry point that computes needs a GPU.
 */
#ifndef TERRARIUM_HIP_H
#define TERRARIUM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRM_ABI_VERSION 19

typedef struct trm_ctx trm_ctx;

/* ---- status codes ------------------------------------------------------- */
enum { TRM_OK = 0, TRM_EINVAL = 1, TRM_EHIP = 2, TRM_ENOMEM = 3, TRM_EUNSUPPORTED = 4,
       TRM_ESTALE = 5, /* the field asked for is not materialised at this point (see trm_step) */
       TRM_ECOMM = 6   /* RCCL could not be loaded / a collective failed */ };

/* ---- number format NF (every reference struct is parameterised by it) ---- */
enum { TRM_F64 = 0, TRM_F32 = 1 };

/* ---- configuration enums -------------------------------------------------- */
enum { TRM_FLOW_NOFLOW = 0, TRM_FLOW_RICHARDS = 1 };        /* soil_hydrology.jl:14, soil_hydrology_rre.jl:18 */
enum { TRM_SWRC_BROOKS_COREY = 0, TRM_SWRC_VAN_GENUCHTEN = 1 }; /* FreezeCurves.jl SWRCs                     */
enum { TRM_UNSATK_LINEAR = 0, TRM_UNSATK_VAN_GENUCHTEN = 1 };   /* soil_hydraulic_properties.jl:166,196     */
/* under NoFlow the reference never fills the z halos of the
 * auxiliary saturation field (they stay 0); MIRROR copies the edge cell instead. */
enum { TRM_HALO_REFERENCE_ZERO = 0, TRM_HALO_MIRROR = 1 };

/* Boundary conditions (Oceananigans Value / Flux / Gradient / default NoFlux). */
enum { TRM_BC_NOFLUX = 0, TRM_BC_VALUE = 1, TRM_BC_FLUX = 2, TRM_BC_GRADIENT = 3 };
enum { TRM_BOTTOM = 0, TRM_TOP = 1 };
enum {
    TRM_BCV_INTERNAL_ENERGY = 0,
    TRM_BCV_SATURATION_WATER_ICE = 1,
    TRM_BCV_TEMPERATURE = 2,
    TRM_BCV_LIQUID_WATER_FRACTION = 3,
    TRM_BCV_PRESSURE_HEAD = 4,
    TRM_BCV_COUNT = 5
};

/* ---- state variables ---------------------------------- */
enum {
    /* 3-D, Nz rows */
    TRM_FIELD_INTERNAL_ENERGY = 0,          /* prognostic, J/m^3   soil_energy.jl:47            */
    TRM_FIELD_SATURATION_WATER_ICE = 1,     /* prognostic (Richards) / auxiliary (NoFlow)        */
    TRM_FIELD_TEMPERATURE = 2,              /* closure, degC      soil_energy_closures.jl:23   */
    TRM_FIELD_LIQUID_WATER_FRACTION = 3,    /* closure             soil_energy_closures.jl:24   */
    TRM_FIELD_PRESSURE_HEAD = 4,            /* closure, m          soil_hydraulic_closures.jl:15 */
    TRM_FIELD_HYDRAULIC_CONDUCTIVITY = 5,   /* Face field, Nz+1 rows  soil_hydrology.jl:81      */
    TRM_FIELD_TEND_INTERNAL_ENERGY = 6,
    TRM_FIELD_TEND_SATURATION_WATER_ICE = 7,
    /* 2-D, 1 row */
    TRM_FIELD_SURFACE_EXCESS_WATER = 8,     /* prognostic, m       soil_hydrology_rre.jl:22     */
    TRM_FIELD_TEND_SURFACE_EXCESS_WATER = 9,
    TRM_FIELD_WATER_TABLE = 10,             /* auxiliary, m        soil_hydrology.jl:80         */
    TRM_FIELD_SKIN_TEMPERATURE = 11,        /* skin_temperature.jl:85                           */
    TRM_FIELD_GROUND_HEAT_FLUX = 12,        /* skin_temperature.jl:86; top Flux BC of U in LandModel */
    TRM_FIELD_SURFACE_SHORTWAVE_UP = 13,    /* radiative_fluxes.jl:104-108                      */
    TRM_FIELD_SURFACE_LONGWAVE_UP = 14,
    TRM_FIELD_SURFACE_NET_RADIATION = 15,
    TRM_FIELD_SENSIBLE_HEAT_FLUX = 16,      /* turbulent_fluxes.jl:54-57                        */
    TRM_FIELD_LATENT_HEAT_FLUX = 17,
    TRM_FIELD_EVAPORATION_GROUND = 18,      /* bare_ground_evaporation.jl:27                    */
    TRM_FIELD_INFILTRATION = 19,            /* direct_surface_runoff.jl:65-68; -I = top Flux BC of sat */
    TRM_FIELD_SURFACE_RUNOFF = 20,
    /* PrescribedAtmosphere inputs (prescribed_atmosphere.jl:89-99) */
    TRM_FIELD_AIR_TEMPERATURE = 21,
    TRM_FIELD_AIR_PRESSURE = 22,
    TRM_FIELD_WINDSPEED = 23,
    TRM_FIELD_SPECIFIC_HUMIDITY = 24,
    TRM_FIELD_RAINFALL = 25,
    TRM_FIELD_SURFACE_SHORTWAVE_DOWN = 26,
    TRM_FIELD_SURFACE_LONGWAVE_DOWN = 27,
    /* 3-D, Nz rows: the user `vwc_forcing` (soil_hydrology.jl:37-38, forcings.jl:13-15) evaluated per cell by the
     * caller [1/s].  Uploading it switches the Richards tendency from the scalar trm_params.vwc_forcing to this
     * field; TRM_OPT_VWC_FORCING_FIELD = 0 switches back. */
    TRM_FIELD_VWC_FORCING = 28,
    /* 2-D inputs of PrescribedAlbedo (src/processes/surface_energy/albedo.jl:8-14): read by the surface energy balance
     * per column instead of trm_params.albedo / .emissivity when trm_params.prescribed_albedo = 1 */
    TRM_FIELD_ALBEDO = 29,
    TRM_FIELD_EMISSIVITY = 30,
    /* ---- vegetation (VegetationCarbon, src/processes/vegetation/; enabled by trm_set_vegetation) -- all 2-D ---- */
    TRM_FIELD_CARBON_VEGETATION = 31,        /* prognostic, kgC/m^2   carbon_dynamics.jl:55                          */
    TRM_FIELD_VEGETATION_AREA_FRACTION = 32, /* prognostic           vegetation_dynamics.jl:27                      */
    TRM_FIELD_TEND_CARBON_VEGETATION = 33,
    TRM_FIELD_TEND_VEGETATION_AREA_FRACTION = 34,
    TRM_FIELD_BALANCED_LEAF_AREA_INDEX = 35, /* carbon_dynamics.jl:56                                                */
    TRM_FIELD_PHENOLOGY_FACTOR = 36,         /* phenology.jl:24-25                                                   */
    TRM_FIELD_LEAF_AREA_INDEX = 37,
    TRM_FIELD_CANOPY_WATER_CONDUCTANCE = 38, /* m/s                  stomatal_conductance.jl:29-30                  */
    TRM_FIELD_LEAF_TO_AIR_CO2_RATIO = 39,
    TRM_FIELD_NET_ASSIMILATION = 40,         /* gC/m^2/s             photosynthesis.jl:73-75                        */
    TRM_FIELD_LEAF_RESPIRATION = 41,
    TRM_FIELD_GROSS_PRIMARY_PRODUCTION = 42, /* kgC/m^2/s                                                            */
    TRM_FIELD_AUTOTROPHIC_RESPIRATION = 43,  /* autotrophic_respiration.jl:33-34                                     */
    TRM_FIELD_NET_PRIMARY_PRODUCTION = 44,
    TRM_FIELD_CO2 = 45,                      /* input, ppm (default 380)   prescribed_atmosphere.jl:10-14           */
    TRM_FIELD_SOIL_MOISTURE_LIMITING_FACTOR = 46, /* input (default 1) without soil; the root-weighted integral of     */
                                             /* plant_available_water with it   plant_available_water.jl:21-36       */
    TRM_FIELD_DAILY_LEAF_RESPIRATION = 47,   /* input, gC/m^2/s      autotrophic_respiration.jl:36                  */
    TRM_FIELD_VEGETATION_GROUND_TEMPERATURE = 48, /* `ground_temperature` input of the standalone VegetationModel      */
                                             /* (default 10 degC, autotrophic_respiration.jl:38); with soil the top cell is read */
    /* 3-D, Nz rows (allocated by trm_set_vegetation) */
    TRM_FIELD_PLANT_AVAILABLE_WATER = 49,    /* plant_available_water.jl:22                                          */
    TRM_FIELD_ROOT_FRACTION = 50,            /* static, normalised   root_distribution.jl:45-63                     */
    /* LandModel with vegetation: PALADYNCanopyInterception / PALADYNCanopyEvapotranspiration, one value per column */
    TRM_FIELD_CANOPY_WATER = 51,             /* prognostic, m        canopy_interception.jl:55                      */
    TRM_FIELD_TEND_CANOPY_WATER = 52,
    TRM_FIELD_CANOPY_WATER_INTERCEPTION = 53, /* m/s                 canopy_interception.jl:56-59                   */
    TRM_FIELD_CANOPY_WATER_REMOVAL = 54,
    TRM_FIELD_SATURATION_CANOPY_WATER = 55,
    TRM_FIELD_RAINFALL_GROUND = 56,          /* rain reaching the ground, m/s: what the runoff scheme routes        */
    TRM_FIELD_EVAPORATION_CANOPY = 57,       /* m/s                  canopy_evapotranspiration.jl:90-92             */
    TRM_FIELD_TRANSPIRATION = 58,
    TRM_FIELD_STEM_AREA_INDEX = 59,          /* input `SAI` (default 0)   canopy_interception.jl:64                 */
    TRM_FIELD_COUNT = 60
};

/* ---- diagnostics ---------------------------------------------------------- */
enum { TRM_REDUCE_SUM = 0, TRM_REDUCE_MIN = 1, TRM_REDUCE_MAX = 2, TRM_REDUCE_HASNAN = 3, TRM_REDUCE_VOLUME_INTEGRAL_Z = 4 };
/* trm_status flag bits: replace the reference's CPU-side @assert / DEBUG NaN scans
 * (soil_volume.jl:26-28,85; diagnostics/debugging.jl:19-25). */
enum { TRM_STATUS_NAN = 1u, TRM_STATUS_COMPOSITION_OUT_OF_RANGE = 2u,
       TRM_STATUS_HANDOFF_TIMEOUT = 4u /* TRM_OPT_SURFACE_IN_LAUNCH: a column wave gave up waiting for its surface fluxes; its columns are NaN */ };

/* ---- options (trm_set_option) --------------------------------------------- */
enum {
    TRM_OPT_ASYNC = 0,          /* 1: trm_step & co. return after enqueueing on the context stream          */
    TRM_OPT_STEP_KERNEL = 1,    /* TRM_KERNEL_*: which implementation trm_step uses                          */
    TRM_OPT_WRITE_KF_EVERY_STEP = 2,/* 1 (default): hydraulic_conductivity is stored by every step launch;   */
                                    /* 0: only by launches that finalize (it is never an input of a step)   */
    TRM_OPT_VWC_FORCING_FIELD = 3,  /* 1 after TRM_FIELD_VWC_FORCING was uploaded: per-cell vwc_forcing; 0: scalar */
    TRM_OPT_PACKED_F32 = 4,         /* 1 (default): fp32 contexts with the reference-default hydraulics step two       */
                                    /* columns per lane with packed fp32 instructions (bit-identical results); 0: off */
    TRM_OPT_DERIVE_CLOSURE_FIELDS = 5, /* When the stored temperature / liquid_water_fraction are known to be the energy     */
                                    /* closure of the stored internal_energy / saturation (the library wrote them), a fused   */
                                    /* Euler step can re-derive them in registers instead of reading them: 2 of 5 field reads */
                                    /* less, bit-identical results.  0: never; 1: whenever legal; 2 (default): for fp64 states  */
                                    /* beyond the 256 MiB Infinity Cache or of >= 24 576 columns, and the liquid fraction alone */
                                    /* for fp32 states beyond the cache on the packed kernel: where it was measured to win      */
                                    /* (DESIGN 4.1, 4.3); 3: the liquid fraction alone (one read less); 4: the liquid fraction   */
                                    /* and, on the packed fp32 step with Richards, the pressure head from the stored saturation */
                                    /* and water table (two reads less; elsewhere as 3)                                          */
                                    /* 5: as 1 (until round 5 the fp64 column program had an instance that derived the pressure */
                                    /* head as well: measured slower twice, removed; EXPERIMENTS.md).  On the fp64 column       */
                                    /* program 3 and 4 select 1 as well: its "liquid fraction alone" instance went the same way */
    TRM_OPT_STEPS_PER_LAUNCH = 6,   /* trm_step keeps every column in registers for up to m steps per launch and writes the */
                                    /* fields once per launch (temporal blocking of run!'s loop, model_integrator.jl:72-88;  */
                                    /* bit-identical to m = 1).  0 (default): the library chooses -- 50 wherever the program  */
                                    /* is legal (branch-free boundary kinds, constants or device-resident series the program  */
                                    /* interpolates itself, no coupled vegetation; fp32 contexts included); 1: one launch per */
                                    /* step (state streams through memory every step: what bench.py's headline measures);     */
                                    /* m > 1: explicit                                                                          */
    TRM_OPT_PIPELINE_PARTS = 7      /* bare-ground LandModel, one launch per step and half: the columns are dealt to two      */
                                    /* halves and every launch covers the soil columns of one half AND the 0-D surface         */
                                    /* processes (land_model.jl:79-88) of the other, so that the latency-bound surface chain  */
                                    /* runs under the column program (columns are independent; bit-identical results).        */
                                    /* Applies to calls of >= 2 steps with constant inputs and the branch-free boundary kinds. */
                                    /* 0: off; 1: whenever legal; 2 (default): the library's rule -- currently never: measured  */
                                    /* within +-2 % at 812 500 columns and +10 % at N145 (every launch carries ~4 us of fixed    */
                                    /* cost)                                                                         */
    ,TRM_OPT_SINGLE_STEP_PROGRAM = 8 /* trm_step(ctx, dt, 1, .) of a bare-ground LandModel whose inputs change every step (a        */
                                    /* coupled atmosphere: speedy_dry_land.jl:45-68): 1 runs the resident column program with the   */
                                    /* surface processes inline for the single step -- ONE launch instead of the k_surface +        */
                                    /* k_column pair; 0: the launch pair; 2 (default): the library's rule (small shards, where the  */
                                    /* step is bound by launch latency: DESIGN 4.9)                                                  */
    ,TRM_OPT_BC_SIGNATURE = 9       /* 1 (default): a per-step ForwardEuler launch whose boundary kinds match one of the signatures    */
                                    /* compiled into the library (none; prescribed surface temperature; that + a bottom heat flux;    */
                                    /* that + an infiltration flux; the LandModel wiring) takes the program with the kinds as          */
                                    /* compile-time constants (-3 ... -10 %,                                                            */
                                    /* DESIGN 4.3); 0: always the program that reads the kinds at run time (same results; A/B, tests)  */
    ,TRM_OPT_ZERO_GRADIENT_FAST = 10 /* 1 (default): a Gradient condition on temperature or pressure head at the BOTTOM whose values   */
                                    /* trm_set_bc received as +0 everywhere -- the reference's FreeDrainage(), soil_model_bcs.jl:40 --  */
                                    /* forms the same halo, bit for bit, as no condition at all, and the context keeps the branch-free */
                                    /* programs (derivation, resident multi-step program, ...) instead of the generic-boundary kernels; */
                                    /* handing the value buffer out (trm_bc_device_ptr) or attaching a series ends it.  0: off (A/B)    */
    ,TRM_OPT_SURFACE_IN_LAUNCH = 11 /* bare-ground LandModel stepped one launch per step (fp64, Richards, the LandModel's boundary wiring,  */
                                    /* one level per lane): 1 = the 0-D surface processes (land_model.jl:79-88) run in the FIRST workgroups */
                                    /* of the step launch, one lane per column, and hand ground heat flux, infiltration and the skin        */
                                    /* temperature to the column workgroups of the same launch through tagged 8-byte words -- ONE launch     */
                                    /* per step instead of the k_surface + k_column pair, same operations per column (bit-identical); the   */
                                    /* inputs may change every step (nothing is evaluated ahead).  A column wave whose bounded wait for     */
                                    /* its words ends unanswered raises TRM_STATUS_HANDOFF_TIMEOUT instead of hanging.  0: the launch pair;  */
                                    /* 2 (default): the library's rule                                                          */
};
/* DIAGNOSTIC, read-only (trm_get_option): which fast paths the NEXT step will take -- what the library tracks about its own
 * buffers.  Tests pin them (a wrong value costs speed, never correctness, so nothing else would notice). */
enum {
    TRM_INFO_TOP_ARRAYS_CURRENT = 100, /* 1: the LandModel's next surface evaluation reads the compact top-cell arrays the last   */
                                       /* fused step wrote (coalesced) instead of gathering one word per column from the fields   */
    TRM_INFO_CLOSURE_CONSISTENT = 101,  /* 1: the stored temperature / liquid fraction are the closure of the stored state, so a  */
                                       /* step may re-derive them in registers (TRM_OPT_DERIVE_CLOSURE_FIELDS)                   */
    TRM_INFO_GENERIC_BOUNDARY_KERNELS = 103, /* 1: the context's boundary kinds need the generic-boundary kernels (k_step_wave, ...)   */
    TRM_INFO_BC_SIGNATURE = 102,       /* the boundary-condition signature of the context's current kinds (BCSIG bits: 1 / 2 Value on  */
                                       /* temperature bottom / top, 4 / 8 Flux on energy / saturation bottom, 16 / 32 top, 64 LandModel) */
    TRM_INFO_LAST_PROGRAM = 105        /* which kernel instance the last step launch of the context selected (TRM_PROGRAM_* below), 0     */
                                       /* before the first step: family in bits 0-7, then one field per selection rule                    */
};
/* kernel families reported in the low byte of TRM_INFO_LAST_PROGRAM; bits 8-9 the hydraulics instance (0 the reference default, 1 van
 * Genuchten n = 2, 2 run-time exponents), 10-11 lanes per column / 32, 12-14 which closure fields are derived, 15 per-column outputs
 * staged, 16 per-column inputs through the scalar path, 17-24 the compiled-in boundary signature + 1 (0: kinds read at run time) */
enum {
    TRM_PROGRAM_NONE = 0, TRM_PROGRAM_COLUMN_EULER = 1, TRM_PROGRAM_COLUMN_HEUN = 2, TRM_PROGRAM_COLUMN_MULTI = 3,
    TRM_PROGRAM_PACKED_F32 = 4, TRM_PROGRAM_GENERIC_EULER = 5, TRM_PROGRAM_GENERIC_HEUN = 6, TRM_PROGRAM_COLUMN_LAND = 7,
    TRM_PROGRAM_DEEP = 8, TRM_PROGRAM_WIDE = 9, TRM_PROGRAM_LAND_INTERLEAVED = 10, TRM_PROGRAM_UNFUSED = 11, TRM_PROGRAM_VEGETATION = 12,
    TRM_PROGRAM_PACKED_LAND = 13
};
enum {
    TRM_KERNEL_FUSED = 0,       /* one launch per step: lane = soil level, a column per (half-)wavefront,     */
                                /* wavefront shuffles for the vertical stencil (Nz <= 64; two levels per lane */
                                /* for 65 ... 128 levels, four for 129 ... 256: ForwardEuler and Heun with    */
                                /* every boundary kind; anything deeper takes the unfused kernels)            */
    TRM_KERNEL_UNFUSED = 1      /* one launch per reference kernel, in the reference's order (A/B comparator) */
};

/* ---- grid: ColumnGrid(arch, NF, vert, num_columns)  src/grids/column_grid.jl:20-34 */
typedef struct trm_grid {
    int32_t precision;        /* TRM_F64 | TRM_F32                                                        */
    int32_t num_layers;       /* Nz >= 2                                                                   */
    int64_t num_columns;      /* Nh >= 1 (this device's shard)                                             */
    const double* thickness;  /* [Nz] get_spacing(vert): index 0 = SURFACE layer (vertical_discretization.jl:20) */
    double dx;                /* x spacing of the underlying RectilinearGrid; <= 0 selects 1/Nh (x = (0,1)) */
    int32_t device;           /* HIP device ordinal                                                        */
    int32_t reserved;
} trm_grid;

/* ---- parameters: the flat POD of SURVEY 8(a12); defaults = reference defaults -- */
typedef struct trm_params {
    /* PhysicalConstants  src/processes/physical_constants.jl:9-51 */
    double rho_w, rho_i, rho_a, c_a, Lsl, Llg, Lsg, g, Tref, sigma, kappa_vk, eps_mw, R_a;
    /* SoilThermalConductivities / SoilHeatCapacities  soil_thermal_properties.jl:14-46 */
    double k_water, k_ice, k_air, k_mineral, k_organic;
    double c_water, c_ice, c_air, c_mineral, c_organic;
    /* ConstantSoilPorosity soil_porosity.jl:7-13; ConstantSoilCarbonDensity constant_soil_carbon.jl:10-16 */
    double por_mineral, por_organic, rho_soc, rho_org;
    /* ConstantSoilHydraulics / SoilHydraulicsSURFEX + SWRC + UnsatK  soil_hydraulic_properties.jl:66-221 */
    double K_sat, theta_res, bc_psi_s, bc_lambda, vg_alpha, vg_n, impedance;
    double vwc_forcing;       /* spatially constant vwc_forcing source/sink [1/s] (soil_hydrology.jl:37-38) */
    /* ConstantAlbedo, ImplicitSkinTemperature, ConstantAerodynamics, PrescribedAtmosphere,
     * DirectSurfaceRunoff, ConstantEvaporationResistanceFactor */
    double albedo, emissivity, kappa_s, C_h, min_windspeed, tau_r, beta_evap;
    double field_capacity;    /* field_capacity(hydraulic_properties, texture) (soil_hydraulic_properties.jl:93-97,150-155): read by
                                 the soil-moisture evaporation resistance only                                        */
    int32_t flow;             /* TRM_FLOW_*                                                               */
    int32_t swrc;             /* TRM_SWRC_*                                                               */
    int32_t unsat_k;          /* TRM_UNSATK_*                                                             */
    int32_t seb;              /* 0: SoilModel; 1: LandModel(vegetation = nothing) coupling (land_model.jl) */
    int32_t halo_policy;      /* TRM_HALO_*                                                               */
    int32_t prescribed_albedo;/* 0: ConstantAlbedo (the two scalars above); 1: PrescribedAlbedo -- per-column inputs
                                 TRM_FIELD_ALBEDO / TRM_FIELD_EMISSIVITY (albedo.jl:8-14, abstract_types.jl:120-131)   */
    int32_t evap_resistance;  /* ground evaporation resistance factor beta (ground_resistance_factor.jl): 0 = constant
                                 `beta_evap` (ConstantEvaporationResistanceFactor), 1 = SoilMoistureResistanceFactor:
                                 (1 - cos(pi * theta_1 / theta_fc))^2 / 4 below field capacity, 1 above (Lee & Pielke 1992) */
    int32_t reserved;
} trm_params;

/* Fill `p` with the reference defaults. */
int trm_default_params(trm_params* p);

/* ---- vegetation: the parameter structs of VegetationCarbon's processes (needleleaf-tree PFT defaults) --------------- */
typedef struct trm_vegetation_params {
    /* LUEPhotosynthesis  photosynthesis.jl:17-68 */
    double tau25, Kc25, Ko25, q10_tau, q10_Kc, q10_Ko, alpha_leaf, alpha_a, alpha_C3, cq, k_ext, T_CO2_high, T_CO2_low,
        T_photos_high, T_photos_low, theta_r;
    /* MedlynStomatalConductance  stomatal_conductance.jl:16-24 */
    double g1, g_min;
    /* PALADYNAutotrophicRespiration  autotrophic_respiration.jl:14-23 */
    double cn_sapwood, cn_root, aws;
    /* PALADYNCarbonDynamics  carbon_dynamics.jl:18-43 */
    double SLA, awl, LAI_min, LAI_max, gamma_L, gamma_R, gamma_S;
    /* PALADYNVegetationDynamics  vegetation_dynamics.jl:15-22 */
    double nu_seed, gamma_v_min;
    /* StaticExponentialRootDistribution  root_distribution.jl:23-29 */
    double root_a, root_b;
    /* wilting point / field capacity of the soil's hydraulic properties (FieldCapacityLimitedPAW) */
    double wilting_point, field_capacity;
    /* PhysicalConstants.C_mass  physical_constants.jl:50 */
    double C_mass;
    /* PALADYNCanopyInterception (canopy_interception.jl:37-49): interception factor, canopy extinction coefficient,
     * interception capacity [m], removal timescale [s]; PALADYNCanopyEvapotranspiration.C_can: ground-canopy drag
     * coefficient (canopy_evapotranspiration.jl:33-40).  Used by TRM_VEGETATION_COUPLED only. */
    double alpha_int, canopy_k_ext, w_can_max, tau_w, C_can;
} trm_vegetation_params;
int trm_default_vegetation_params(trm_vegetation_params* p);
/* Enables the vegetation processes of the context (allocating their 3-D fields, setting the input defaults and the static
 * root fractions).  TRM_VEGETATION_STANDALONE: the context IS a VegetationModel (src/models/vegetation/vegetation_model.jl):
 * trm_step / trm_step_heun / trm_update_state / trm_compute_auxiliary / trm_compute_tendencies / trm_explicit_step act on
 * the vegetation state alone; soil moisture limitation and ground temperature are inputs.  One launch covers
 * update_state! + explicit_step! for all `nsteps` (the 0-D column stays in registers).
 * TRM_VEGETATION_COUPLED (contexts with surface_energy_balance = 1): LandModel(grid; soil, vegetation) of
 * src/models/coupled/land_model.jl:79-97.  compute_auxiliary! becomes soil hydraulics -> plant available water from the
 * soil state, ground temperature = top soil cell, the vegetation processes -> canopy interception -> canopy
 * evapotranspiration (transpiration through the stomatal conductance, ground evaporation below the canopy, evaporation of
 * intercepted water) -> runoff of the rain reaching the ground -> the surface energy balance with the latent heat of all
 * three humidity fluxes; canopy_water, carbon_vegetation and vegetation_area_fraction are stepped with the soil.  The
 * 0-D part of one step is ONE launch in front of the soil column kernel; the resident multi-step program does not apply
 * (Euler steps run one launch pair each).  trm_step_heun takes four launches per step: the 0-D processes at the state, the
 * soil column with both stages in registers (storing only what the 0-D processes need of the stage), the 0-D processes at
 * the stage, the averaged 0-D update. */
enum { TRM_VEGETATION_OFF = 0, TRM_VEGETATION_STANDALONE = 1, TRM_VEGETATION_COUPLED = 2 };
int trm_set_vegetation(trm_ctx* ctx, const trm_vegetation_params* p, int mode);
/* FieldCapacityLimitedPAW (plant_available_water.jl:36-94): plant_available_water per cell from the soil state of the
 * context and soil_moisture_limiting_factor = sum_k PAW_k * root_fraction_k. */
int trm_compute_plant_available_water(trm_ctx* ctx);

/* initialize(model, timestepper) allocation part (src/state_variables.jl:303-314, 418-430):
 * creates all state buffers on `g->device`, zero-filled, inputs at their defaults. */
int trm_create(const trm_grid* g, const trm_params* p, trm_ctx** out);
int trm_destroy(trm_ctx* ctx);
const char* trm_last_error(const trm_ctx* ctx);
int trm_abi_version(void);

/* Geometry queries: rows of a field (Nz, Nz+1 or 1), and the grid the library derived
 * (z of faces [Nz+1] bottom first, z of centres [Nz], dz centre [Nz], dz face [Nz+1]). */
int trm_field_rows(const trm_ctx* ctx, int field, int64_t* rows);
int trm_get_grid(const trm_ctx* ctx, double* z_faces, double* z_centers, double* dz_center, double* dz_face);

/* set!(field, array) / Array(interior(field)) */
int trm_upload(trm_ctx* ctx, int field, const void* host);
int trm_download(trm_ctx* ctx, int field, void* host);
/* Rows [row0, row0 + nrows) of a field, host layout [nrows][num_columns]: `ground_temperature` (soil_energy.jl:52-57) is
 * row Nz - 1 of temperature -- one row through PCIe, not the field. */
int trm_download_rows(trm_ctx* ctx, int field, int row0, int nrows, void* host);
/* ---- ColumnRingGrid (src/grids/column_ring_grid.jl:37-59,102-149) -------------------------------------------------------
 * trm_set_ring_grid: the columns of the context are the points `mask_index[i]` (strictly increasing, one per column) of a
 * full grid of `num_points` points in ring order -- `findall(mask)`; for a context that holds one shard of the columns, the
 * shard's slice of that list.  Uploaded once; scatter / gather then run on the device:
 *   trm_download_ring / trm_scatter_ring_device   RingGrids.Field(field, grid; fill_value) (lines 102-115): rows of a field
 *       on the full grid, [nrows][num_points], `fill` outside the mask -- into host memory, or into a device buffer of a
 *       coupled model on the same device (speedy_dry_land.jl:45-66 reads the land state as ring-grid fields)
 *   trm_upload_ring / trm_gather_ring_device      Oceananigans.Field(ring_field, grid) (lines 117-149): the masked points of
 *       a full-grid array [rows][num_points] become the field (all rows of it) */
int trm_set_ring_grid(trm_ctx* ctx, int64_t num_points, const int64_t* mask_index);
int trm_download_ring(trm_ctx* ctx, int field, int row0, int nrows, double fill, void* host);
int trm_scatter_ring_device(trm_ctx* ctx, int field, int row0, int nrows, double fill, void* dev_out);
int trm_upload_ring(trm_ctx* ctx, int field, const void* host_full);
int trm_gather_ring_device(trm_ctx* ctx, int field, const void* dev_full);
/* Device address of a field, for zero-copy consumers on the same device.  The device layout is z-fastest:
 * element (column i, level k) of a 3-D field is at dev[i * pitch_elems + k] (pitch 32 for Nz <= 32, 64 for
 * Nz <= 64); 2-D fields are dev[i] (pitch 1).  The top face of hydraulic_conductivity is not part of this
 * buffer (use trm_download). */
int trm_field_device_ptr(trm_ctx* ctx, int field, void** dev, int64_t* pitch_elems);

/* Field boundary conditions (src/models/soil/soil_model_bcs.jl, src/boundary_conditions.jl:25-28):
 * `values` is a per-column array [Nh] or NULL to broadcast `scalar`. */
/* Device array [num_columns] of the boundary values of (var, side), as set by trm_set_bc / trm_set_bc_series: a coupled
 * model that lives on the same device writes the next coupling interval's values there itself (the SpeedyWeather
 * coupling of examples/simulations/speedy_dry_land.jl:45-66 does `set!(state.inputs.air_temperature, Tair)` per coupling
 * step).  Valid until the context is destroyed; writes must be ordered before the next trm_step on the context's stream
 * (trm_set_stream) or by a device synchronisation.  Fails if the condition carries no values (NoFlux) or a time series. */
int trm_bc_device_ptr(trm_ctx* ctx, int var, int side, void** dev);
int trm_set_bc(trm_ctx* ctx, int bc_var, int side, int kind, const void* values, double scalar);
/* update_inputs! (src/state_variables.jl:154-162): same as trm_upload on an input field. */
int trm_set_forcing(trm_ctx* ctx, int input_field, const void* per_column);
/* The same from DEVICE memory, stream-ordered: `dev_per_column` ([num_columns], context precision, on the context's device) is
 * copied into the input field on the context stream without synchronising the host -- a coupled model on the same device hands
 * over the next coupling interval's inputs between two asynchronous trm_step calls (speedy_dry_land.jl:45-68 does
 * `set!(state.inputs.air_temperature, Tair)` per coupling step).  The source must stay valid until the copy has executed
 * (trm_synchronize, or stream order on a shared stream: trm_set_stream).  Zero-copy alternative: write the field's own
 * buffer (trm_field_device_ptr) on the context stream. */
int trm_set_forcing_device(trm_ctx* ctx, int input_field, const void* dev_per_column);

/* ---- time series input sources --------------------------------------------------------------------------
 * FieldTimeSeriesInputSource (src/input_output/input_sources.jl:142-171): update_inputs! sets the input
 * field to `fts[Time(clock.time)]`, Oceananigans' time interpolation of a FieldTimeSeries.  Here the whole
 * series lives in HBM ([nt][Nh], uploaded once; 288 GB hold years of hourly forcing for a shard), and every
 * step of a trm_step / trm_step_heun loop evaluates it at the context clock before anything else runs, so one
 * call can span any number of forcing intervals.  `times` must be strictly increasing; nt >= 1.
 * The same mechanism drives time-dependent boundary values (the reference's functional / discrete-form
 * boundary conditions, evaluated on its own clock).
 *   TRM_TIME_LINEAR    linear interpolation, linear extrapolation outside [t_1, t_nt]   (Oceananigans `Linear`)
 *   TRM_TIME_CLAMP     linear interpolation, end values outside                         (`Clamp`)
 *   TRM_TIME_CYCLICAL  periodic with period (t_nt - t_1) + (t_nt - t_{nt-1})            (`Cyclical`)
 * Between nodes n1 < n2: value = v[n2] * f + v[n1] * (1 - f), f = (1 / (t[n2] - t[n1])) * (t - t[n1]), evaluated in
 * double and rounded once to the context precision; on an interior node the node's values are copied.
 *   TRM_TIME_RASTER    the Raster input source of ext/TerrariumRastersExt (lines 96-121): between nodes
 *                      v[n1] + (t - t[n1]) * (v[n2] - v[n1]) / (t[n2] - t[n1]) with (v[n2] - v[n1]) formed in the context
 *                      precision and the rest in double, the node's values on a node, the end values beyond the ends */
enum { TRM_TIME_LINEAR = 0, TRM_TIME_CLAMP = 1, TRM_TIME_CYCLICAL = 2, TRM_TIME_RASTER = 3 };
/* `values`: [nt][Nh] in the context precision.  Replaces any earlier series or constant of the same input. */
int trm_set_forcing_series(trm_ctx* ctx, int input_field, int nt, const double* times, const void* values, int time_indexing);
/* Boundary value series for (bc_var, side) with the given kind (VALUE / FLUX / GRADIENT). */
int trm_set_bc_series(trm_ctx* ctx, int bc_var, int side, int kind, int nt, const double* times, const void* values, int time_indexing);
/* Drops every series (inputs and boundary values keep what was last evaluated). */
int trm_clear_series(trm_ctx* ctx);
/* update_inputs!(state, clock): evaluates every series at the context clock (done implicitly by trm_step,
 * trm_step_heun and trm_update_state). */
int trm_update_inputs(trm_ctx* ctx);

/* ---- windowed time series -----------------------------------------------------------------------------
 * A forcing record that does not fit in HBM -- one hourly year of the 7 atmospheric inputs is ~199 GB for a 0.1-degree shard --
 * streams through a fixed window: create the series with its first levels (trm_set_forcing_series / trm_set_bc_series),
 * then alternate
 *     trm_series_trim_before(ctx, t)        -- releases the levels no evaluation at a time >= t can touch (the node at or
 *                                              before t and everything after it stay)
 *     trm_series_append(ctx, ...)           -- continues a series with `nt` further levels (`times` strictly increasing
 *                                              and beyond the last level held).  The values are staged through pinned
 *                                              memory and copied on a side stream UNDER whatever the context stream is
 *                                              running; the next step that evaluates the series waits for them.  Freed
 *                                              slots of the ring are reused; without free slots the ring grows.
 * `is_bc` = 0: `id` is the input field, `side` ignored; `is_bc` = 1: `id` is the bc_var, `side` the boundary.  Results
 * are bit-identical to the all-resident series as long as the window covers [t, t + dt] of every step taken (FieldTimeSeries
 * `InMemory(chunk)` backends play this role on the reference side).  TRM_TIME_CYCLICAL series cannot be windowed. */
int trm_series_append(trm_ctx* ctx, int is_bc, int id, int side, int nt, const double* times, const void* values);
/* Declares a series as WINDOWED (trm_series_append does the same implicitly): only windowed series are touched by
 * trm_series_trim_before -- a fully resident series that lives in the same context keeps its whole record -- and a windowed
 * series refuses (TRM_EINVAL) an evaluation at a time before the levels it still holds instead of extrapolating from its trimmed
 * head.  `levels` > the levels held reserves ring capacity for that many (0: leave the capacity as it is). */
int trm_series_window(trm_ctx* ctx, int is_bc, int id, int side, int levels);
int trm_series_trim_before(trm_ctx* ctx, double t);
int trm_series_info(const trm_ctx* ctx, int is_bc, int id, int side, int64_t* levels_held, int64_t* capacity, double* t_first, double* t_last);

/* reset!(state) + reset!(clock) of initialize!(integrator) (state_variables.jl:102-120, model_integrator.jl:96-99): every
 * prognostic, auxiliary and tendency field to zero, the status word cleared, the clock to (0, 0).  Inputs keep their values
 * (initialize!(state, inputs) re-evaluates their sources next), as do boundary values, series and options. */
int trm_reset(trm_ctx* ctx);

/* initialize!(state, model) process initialisers (soil_coupled.jl:45-54): hydraulics, water table,
 * sat -> psi, T -> U.  Call after uploading the initial temperature / saturation. */
int trm_initialize(trm_ctx* ctx);
/* update_state!(state, model, inputs; compute_tendencies)  src/state_variables.jl:72-80 */
int trm_update_state(trm_ctx* ctx, int compute_tendencies);
/* compute_auxiliary!(state, model)  soil_model.jl:39-42 / land_model.jl:79-88 */
int trm_compute_auxiliary(trm_ctx* ctx);
/* compute_tendencies!(state, model) soil_model.jl:44-47 (accumulates into the tendency fields) */
int trm_compute_tendencies(trm_ctx* ctx);
/* reset_tendencies!(state)  src/state_variables.jl:127-136 */
int trm_reset_tendencies(trm_ctx* ctx);
/* explicit_step!(state, grid, timestepper, dt)  abstract_timestepper.jl:65-77 */
int trm_explicit_step(trm_ctx* ctx, double dt);
/* closure!/invclosure!(state, model)  soil_coupled.jl:99-122 */
int trm_closure(trm_ctx* ctx);
int trm_invclosure(trm_ctx* ctx);

/* `nsteps` x timestep!(integrator, ForwardEuler, dt; finalize = false), then compute_auxiliary! once
 * if `finalize` (forward_euler.jl:19-31, model_integrator.jl:72-88,124-131).  nsteps = 1, finalize = 1
 * is the reference's timestep!(integrator, dt); finalize = 1 with nsteps = n is run!(steps = n).
 *
 * Tendency fields (TRM_FIELD_TEND_*): the reference leaves the last step's tendencies -- compute_tendencies! plus the
 * flux-boundary term compute_z_bcs! adds inside explicit_step!, averaged over the two stages for Heun -- in
 * state.tendencies.  The fused kernels (TRM_KERNEL_FUSED) keep tendencies in registers and store them only from the
 * launch that finalizes: after a call with finalize = 1 the tendency fields hold exactly what the reference's would;
 * after a call with finalize = 0 they are NOT materialised and trm_download / trm_reduce / trm_field_device_ptr of a
 * TRM_FIELD_TEND_* field fail with TRM_ESTALE until trm_update_state(ctx, 1), trm_reset_tendencies or a finalizing
 * step has run.  TRM_KERNEL_UNFUSED materialises them at every step. */
int trm_step(trm_ctx* ctx, double dt, int nsteps, int finalize);
/* Same for Heun (heun.jl:37-71): with TRM_KERNEL_FUSED and Nz <= 64 ONE launch per step (both stages on the column held in
 * registers; the stage never touches memory; every boundary kind), four for TRM_VEGETATION_COUPLED; ONE launch for 65 ... 128
 * levels with the branch-free boundary kinds as well; otherwise the reference-order kernels on a second copy of the state. */
int trm_step_heun(trm_ctx* ctx, double dt, int nsteps, int finalize);
/* DIAGNOSTIC (benchmarks): as trm_step, bracketed by HIP events on the context stream: *ms = device time of the launches. */
int trm_step_timed(trm_ctx* ctx, double dt, int nsteps, int finalize, float* ms);
/* The same for trm_step_heun. */
int trm_step_heun_timed(trm_ctx* ctx, double dt, int nsteps, int finalize, float* ms);

/* ---- Heun with state-dependent forcings / boundary values (heun.jl:37-71 with forcings.jl:13-19, boundary_conditions.jl:25-28) --
 * The reference evaluates a user's `forcing(i, j, k, grid, clock, fields)` / `getbc(..., clock, fields)` inside the tendency
 * kernel at BOTH Heun stages: at the state (clock t) and at the stage (the predicted state, clock t + dt).  A closure cannot
 * cross a C ABI; its values can, without leaving the device (see trm_field_device_ptr): the caller evaluates its function on
 * the state's buffers into the state's boundary / input / vwc_forcing buffers, then
 *     trm_heun_predict(ctx, dt)            -- heun.jl:41-52: update_state!(state), stage := state, explicit_step!(stage),
 *                                             closure!(stage), tick!(stage.clock); series are evaluated at t + dt for the stage
 *     (the caller evaluates its function on the STAGE's buffers -- trm_stage_field_device_ptr -- into the stage's boundary /
 *      input / vwc_forcing buffers -- trm_stage_bc_device_ptr, trm_stage_field_device_ptr of the input --, on the context stream)
 *     [trm_heun_stage_auxiliary(ctx)       -- the first half of update_state!(stage): reset tendencies, compute_auxiliary!(stage).
 *      (then the caller evaluates what the reference evaluates INSIDE compute_tendencies!(stage): a forcing function)]
 *     trm_heun_correct(ctx, dt, finalize)  -- heun.jl:54-71: update_state!(stage) (what is left of it), average_tendencies!,
 *                                             explicit_step!(state), closure!(state), tick!(clock) (+ compute_auxiliary! if `finalize`)
 * WHAT THE STAGE'S FIELDS HOLD when the caller reads them: after trm_heun_predict every field of the stage is the reference's at
 * the same point -- prognostic and closure fields of the predicted state, auxiliary fields (hydraulic_conductivity, the surface
 * fluxes, ...) still the STATE's copies (copyto!(stage, state), heun.jl:45): what a boundary-value function finds when
 * fill_halo_regions! evaluates it, BEFORE compute_auxiliary!(stage).  After trm_heun_stage_auxiliary the auxiliary fields are the
 * stage's own: what a forcing function finds inside the tendency kernel.  The stage's clock has ticked: time t + dt, iteration + 1.
 * The pair equals one trm_step_heun(ctx, dt, 1, finalize) bit for bit when the stage's values are the ones the library would
 * have used itself.  It runs on the reference-order kernels (the stage lives in memory between the two calls by construction).
 * A boundary condition or input whose stage buffer has been handed out keeps it; trm_step_heun refreshes such buffers from
 * the state's values at every step, so mixing the two forms is safe.
 * Any other call that writes the state, the clock, a boundary condition or an option between trm_heun_predict and trm_heun_correct
 * drops the predicted stage: trm_heun_correct then fails with TRM_EINVAL instead of correcting with a stage of another state. */
int trm_heun_predict(trm_ctx* ctx, double dt);
int trm_heun_stage_auxiliary(trm_ctx* ctx);
int trm_heun_correct(trm_ctx* ctx, double dt, int finalize);
/* Device address of a field of the Heun STAGE (layout as trm_field_device_ptr): the predicted state after trm_heun_predict --
 * to read --, or an input field / TRM_FIELD_VWC_FORCING of the stage -- to write before trm_heun_correct. */
int trm_stage_field_device_ptr(trm_ctx* ctx, int field, void** dev, int64_t* pitch_elems);
/* Device array [num_columns] of the STAGE's boundary values of (var, side) (allocated on first use as a copy of the state's) */
int trm_stage_bc_device_ptr(trm_ctx* ctx, int var, int side, void** dev);

/* ---- DIAGNOSTIC entry points (benchmarks, tests; not part of the reference interface) -----------------------------------
 * trm_step_timed / trm_step_heun_timed (above) and the device-side one-slot snapshot below exist for bench.py and the A/B
 * tools: a binder of the reference interface does not need them.  Restart files go through trm_download / trm_upload /
 * trm_set_clock ("Restart").
 * Device-side checkpoint of the whole state (every field, the clock, the status word): trm_save_state copies it into
 * a second set of buffers owned by the context, trm_restore_state copies it back.  One slot; no host traffic. */
int trm_save_state(trm_ctx* ctx);
int trm_restore_state(trm_ctx* ctx);

int trm_clock(const trm_ctx* ctx, double* time, int64_t* iteration);
int trm_set_clock(trm_ctx* ctx, double time, int64_t iteration);

/* Local (this device's columns) reduction of a field.  SUM/MIN/MAX/HASNAN give one value per row in
 * out[rows]; VOLUME_INTEGRAL_Z gives one value (sum over columns of sum_k f*dz).  trm_reduce_global (below)
 * combines the devices' results. */
int trm_reduce(trm_ctx* ctx, int field, int op, double* out);
int trm_status(trm_ctx* ctx, uint32_t* flags);
/* Sets the status word: a restart puts back the flags its checkpoint carried (the word is sticky: steps only OR into it), so a run
 * that had produced a NaN before the checkpoint does not report a clean status after the restart; 0 clears it. */
int trm_set_status(trm_ctx* ctx, uint32_t flags);

/* ---- multi-device diagnostics ----------------------------------------------------------------------------------------
 * The global grid of laterally independent columns is block-sharded over the devices of a node, one context (and one
 * host process or thread) per device; the step path has NO collective.  Diagnostics that span the shards -- the
 * reference's `sum(field)`, `minimum`, `maximum`, `any(isnan, ...)` on a whole Field, the CPU-side @assert scans -- are
 * combined inside the library with one RCCL all-reduce of <= 2 * (Nz + 1) doubles on the context's side stream (xGMI:
 * latency-bound, link bandwidth irrelevant), so a host in any language gets global values without a communication
 * layer of its own.  RCCL is opened lazily (dlopen) by the first of these calls.
 *   rank 0:      trm_comm_unique_id(id);  broadcast the 128 bytes to the other ranks by any means (MPI, a file, a pipe)
 *   every rank:  trm_comm_init(ctx, rank, world, id);    -- collective: returns once every rank has called it
 *                trm_reduce_global / trm_status_global   -- collective, same arguments on every rank
 * trm_reduce_global gives what trm_reduce would give on the unsharded grid: SUM / VOLUME_INTEGRAL_Z sums over ranks
 * (in RCCL's deterministic ring order), MIN / MAX / HASNAN exactly, a NaN on any rank propagating into MIN / MAX. */
int trm_comm_unique_id(void* id128);
int trm_comm_init(trm_ctx* ctx, int rank, int world_size, const void* id128);
int trm_comm_destroy(trm_ctx* ctx);
/* *world_size = 0 while the context has no communicator */
int trm_comm_info(const trm_ctx* ctx, int* rank, int* world_size);
int trm_reduce_global(trm_ctx* ctx, int field, int op, double* out);
int trm_status_global(trm_ctx* ctx, uint32_t* flags);
/* ---- ONE host process (thread) driving n contexts, one per device -- the reference's host is one Julia process
 * (column_grid.jl:32, model_integrator.jl:72-88).  trm_comm_init is a blocking collective: a single thread that holds every
 * context would wait in the first call for ranks it has not reached yet.  The *_all forms take all contexts at once:
 *   trm_comm_init_all     one RCCL communicator per context, created inside one ncclGroupStart / ncclGroupEnd; contexts must
 *                         sit on distinct devices (RCCL refuses two ranks per device); rank = position in `ctxs`
 *   trm_step_all          trm_step on every context without waiting in between (each on its own stream), then one wait for
 *                         all of them unless the contexts are asynchronous (TRM_OPT_ASYNC); trm_synchronize_all waits
 *   trm_reduce_global_all / trm_status_global_all
 *                         what trm_reduce_global / trm_status_global give, for all contexts from one thread: the per-device
 *                         partial results are combined by grouped RCCL all-reduces when the contexts carry communicators
 *                         (trm_comm_init_all), and on the host otherwise (one process needs no wire for <= 2 (Nz + 1)
 *                         doubles per device): same values, `out` written once. */
int trm_comm_init_all(trm_ctx** ctxs, int n);
int trm_step_all(trm_ctx** ctxs, int n, double dt, int nsteps, int finalize);
int trm_step_heun_all(trm_ctx** ctxs, int n, double dt, int nsteps, int finalize);
int trm_synchronize_all(trm_ctx** ctxs, int n);
int trm_reduce_global_all(trm_ctx** ctxs, int n, int field, int op, double* out);
int trm_status_global_all(trm_ctx** ctxs, int n, uint32_t* flags);

int trm_set_option(trm_ctx* ctx, int option, int value);
int trm_get_option(const trm_ctx* ctx, int option, int* value);
/* Adopt an external hipStream_t (e.g. torch's current stream); NULL restores the context's own. */
int trm_set_stream(trm_ctx* ctx, void* hip_stream);
int trm_synchronize(trm_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* TERRARIUM_HIP_H */
