#!/usr/bin/env python3
"""A coupled land column -- the host-side mirror of the reference's examples/simulations/land_column.jl:

    ColumnGrid(ExponentialSpacing(dz_max = 1, N = 30))
    van Genuchten(alpha = 2, n = 2) retention + Mualem conductivity, Richards flow, VegetationCarbon
    LandModel(grid; soil, vegetation): canopy interception, canopy evapotranspiration, surface energy balance
    saturation min(1, 0.5 - 0.1 z) (water table at roughly 5 m), carbon_vegetation = 0.1;  one 60 s ForwardEuler step

    python examples/land_column.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import terrarium_jl_amd as trm  # noqa: E402


def build(num_columns=1, dtype=np.float64):
    grid = trm.ColumnGrid(trm.ExponentialSpacing(dz_max=1.0, N=30), num_columns, dtype=dtype)
    swrc = trm.VanGenuchten(alpha=2.0, n=2.0)
    hydraulic_properties = trm.ConstantSoilHydraulics(swrc=swrc, unsat_hydraulic_cond=trm.UnsatKVanGenuchten())
    hydrology = trm.SoilHydrology(vertical_flow=trm.RichardsEq(), hydraulic_properties=hydraulic_properties)
    soil = trm.SoilEnergyWaterCarbon(hydrology=hydrology)
    vegetation = trm.VegetationCarbon()
    land = trm.LandModel(grid, soil=soil, vegetation=vegetation)
    initializers = dict(saturation_water_ice=lambda x, z: min(1.0, 0.5 - 0.1 * z), carbon_vegetation=0.1)
    return trm.initialize(land, trm.ForwardEuler(), initializers=initializers)


def main():
    integrator = build()
    trm.timestep(integrator, 60.0)
    st = integrator.state
    print(f"after one 60 s step: status {st.status()}")
    for name in ("skin_temperature", "ground_heat_flux", "latent_heat_flux", "transpiration", "evaporation_ground", "evaporation_canopy",
                 "infiltration", "carbon_vegetation", "leaf_area_index", "soil_moisture_limiting_factor", "water_table"):
        print(f"  {name:32s} {float(st.get(name)[0]): .6e}")
    print("  saturation (top five cells)     ", np.array2string(st.saturation_water_ice[-5:, 0], precision=5))


if __name__ == "__main__":
    main()
