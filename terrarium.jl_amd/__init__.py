"""MI355X-native SoilModel / LandModel(bare ground) time-stepping core.

Host-side mirror of the Terrarium.jl interface for the one hot path this
repository accelerates: the explicit time step of
`SoilModel` -- heat conduction with freeze/thaw, Richards water transport and
the bare-ground surface energy balance -- over laterally independent columns.
All arithmetic runs in hand-written HIP kernels (csrc/) behind the C ABI of
include/terrarium_hip.h; there is no CPU fallback.
"""
from .grids import (AbstractVerticalSpacing, UniformSpacing, ExponentialSpacing, PrescribedSpacing, ColumnGrid,
                    ColumnRingGrid)
from . import io
from . import masks
from . import parallel
from . import _capi
from .models import (PhysicalConstants, SoilThermalConductivities, SoilHeatCapacities, SoilThermalProperties,
                     SoilEnergyBalance, ConstantSoilPorosity, SoilPorositySURFEX, HomogeneousStratigraphy, ConstantSoilCarbonDensity,
                     BrooksCorey, VanGenuchten, UnsatKLinear, UnsatKVanGenuchten, ConstantSoilHydraulics,
                     SoilHydraulicsSURFEX, SoilTexture, NoFlow, RichardsEq, SoilHydrology, SoilEnergyWaterCarbon, ConstantAlbedo,
                     PrescribedAlbedo,
                     ImplicitSkinTemperature, SurfaceEnergyBalance, ConstantAerodynamics, PrescribedAtmosphere,
                     DirectSurfaceRunoff, BareGroundEvaporation, ConstantEvaporationResistanceFactor, SoilMoistureResistanceFactor,
                     SurfaceHydrology, NoCanopyInterception, PALADYNCanopyInterception, PALADYNCanopyEvapotranspiration, DefaultInitializer,
                     ConstantSoilTemperature, QuasiThermalSteadyState, PiecewiseLinearInitialSoilTemperature, piecewise_linear,
                     ConstantSaturation, SaturationWaterTable,
                     SoilInitializer, SoilModel, LandModel, flatten, LUEPhotosynthesis, MedlynStomatalConductance,
                     PALADYNAutotrophicRespiration, PALADYNPhenology, PALADYNCarbonDynamics, PALADYNVegetationDynamics,
                     StaticExponentialRootDistribution, FieldCapacityLimitedPAW, VegetationCarbon, VegetationModel, flatten_vegetation)
from .integrator import (ForwardEuler, Heun, PrescribedSurfaceTemperature, PrescribedBottomTemperature,
                         GroundHeatFlux, GeothermalHeatFlux, InfiltrationFlux, ImpermeableBoundary, FreeDrainage,
                         merge_boundary_conditions, DeviceState, ModelIntegrator, FieldTimeSeries, StateFunction, InputSource, InputSources, initialize, initialize_integrator,
                         timestep, run, current_time, compute_auxiliary, compute_tendencies, closure, invclosure,
                         update_state, default_dt, is_adaptive, iteration, time_step, reset, get_grid, znodes, zspacings,
                         checkpoint, restore, restart_fields, DeviceGroup)
from ._capi import TerrariumHipError
from .io import Hdf5File, RasterInputSource
from .simulation import Simulation, Callback, IterationInterval, TimeInterval, SnapshotWriter, run_simulation
