"""Pin the vegetation part of the CPU oracle (oracle/vegetation_oracle.hpp) against the reference's own unit tests under
test/vegetation/ (SURVEY 8(f) row 4).  Each test names the reference test it restates.  CPU only."""
import math

import numpy as np
import pytest

import oracle
from oracle import veg_scalar as V, default_vegetation_params

P = default_vegetation_params()


# test/vegetation/carbon_dynamics_tests.jl:5-60
def test_lambda_NPP():
    assert V("lambda_NPP", P.LAI_min / 2) == 0.0 and V("lambda_NPP", P.LAI_min) == 0.0
    assert 0 < V("lambda_NPP", (P.LAI_min + P.LAI_max) / 2) < 1
    assert V("lambda_NPP", P.LAI_max) == 1.0 and V("lambda_NPP", P.LAI_max * 2) == 1.0


def test_carbon_dynamics_finite_and_positive():
    assert math.isfinite(V("LAI_b", 0.5)) and V("LAI_b", 0.5) > 0
    assert V("LAI_b", 0.5) == 0.5 / ((2.0 / 10.0) + 2.0)                       # carbon_dynamics.jl:84-87
    mid = (P.LAI_min + P.LAI_max) / 2
    assert math.isfinite(V("Lambda_loc", mid)) and V("Lambda_loc", mid) > 0
    assert math.isfinite(V("C_veg_tend", mid, 0.5))


# test/vegetation/phenology_tests.jl:5-25
def test_phenology_placeholders():
    assert V("f_deciduous") == 0.0 and V("phenology_factor") == 1.0 and V("LAI", 5.0) == 5.0


# test/vegetation/vegetation_dynamics_tests.jl:5-41
def test_vegetation_dynamics():
    assert V("gamma_v") == P.gamma_v_min
    assert V("nu_star", P.nu_seed / 2) == P.nu_seed and V("nu_star", P.nu_seed * 2) == P.nu_seed * 2 and V("nu_star", P.nu_seed) == P.nu_seed
    mid = (P.LAI_min + P.LAI_max) / 2
    assert math.isfinite(V("nu_tendency", mid, 0.5, 1.0e-3, 0.3))


# test/vegetation/stomatal_conductance_tests.jl:5-17
def test_lambda_c():
    assert V("lambda_c", np.finfo(float).eps) == pytest.approx(1.0)
    assert 0.0 < V("lambda_c", 1000.0) < 1.0


# test/vegetation/photosynthesis_tests.jl:7-22 (kinetic parameters)
def test_kinetic_parameters():
    cold = [V(n, 20.0) for n in ("tau", "Kc", "Ko")]
    warm = [V(n, 30.0) for n in ("tau", "Kc", "Ko")]
    assert all(math.isfinite(x) and x > 0 for x in cold)
    assert warm[0] < cold[0] and warm[1] > cold[1] and warm[2] > cold[2]
    assert V("tau", 25.0) == 2600.0 and V("Kc", 25.0) == 30.0 and V("Ko", 25.0) == 3.0e4      # q10^0 = 1


# photosynthesis_tests.jl:24-37, 39-57, 59-77, 79-99
def test_gamma_star_par_apar_pres_i():
    g = V("Gamma_star", 3000.0, 20.9e3)
    assert math.isfinite(g) and g > 0 and V("Gamma_star", 2000.0, 20.9e3) > g
    assert V("PAR", 50.0) > 0 and V("PAR", 0.0) == 0 and V("PAR", 100.0) == pytest.approx(2 * V("PAR", 50.0))
    assert V("APAR", 50.0, 5.0) > 0 and V("APAR", 50.0, 0.0) == 0 and V("APAR", 50.0, math.inf) == P.alpha_a * V("PAR", 50.0)
    assert V("pres_i", 0.0, 40.0) == 0.0 and V("pres_i", 1.0, 40.0) == 40.0 and 0 < V("pres_i", 0.5, 40.0) < 40.0


# photosynthesis_tests.jl:101-127
def test_temperature_stress():
    assert V("temperature_stress", P.T_CO2_low * 2) == 0.0 and V("temperature_stress", P.T_CO2_low) == 0.0
    assert V("temperature_stress", P.T_CO2_high * 2) == 0.0 and V("temperature_stress", P.T_CO2_high) == 0.0
    assert 0.0 < V("temperature_stress", (P.T_CO2_low + P.T_CO2_high) / 2) < 1.0


# photosynthesis_tests.jl:129-157
def test_assimilation_factors():
    Gs, Kc, Ko, pO2 = 3.0, 20.0, 3.0e4, 20.9e3
    Ts = V("temperature_stress", 20.0)
    assert V("c_1", Gs, Ts, Kc, Ko, Gs, pO2) == 0 and V("c_2", Gs, Ts, Kc, Ko, Gs, pO2) == 0
    c1, c2 = V("c_1", Gs, Ts, Kc, Ko, Gs / 2, pO2), V("c_2", Gs, Ts, Kc, Ko, Gs / 2, pO2)
    assert math.isfinite(c1) and c1 <= 0 and math.isfinite(c2) and c2 < 0
    c1, c2 = V("c_1", Gs, Ts, Kc, Ko, Gs * 2, pO2), V("c_2", Gs, Ts, Kc, Ko, Gs * 2, pO2)
    assert math.isfinite(c1) and c1 >= 0 and math.isfinite(c2) and c2 > 0


# photosynthesis_tests.jl:159-177, 179-209, 211-232, 234-250
def test_vcmax_je_jc_rd_ag():
    assert V("Vc_max", 0.5, 0.0, 20.0, 3.0e4, 3.0, 20.0, 20.9e3) == 0.0
    assert math.isfinite(V("Vc_max", 0.5, 4.0, 20.0, 3.0e4, 3.0, 20.0, 20.9e3))
    assert V("JE", 0.0, 0.0, 4.0, 0.0) == 0.0 and V("JC", 0.0, 0.0, 4.0, 0.0) == 0.0
    assert V("JE", 0.5, 0.5, 0.0, 0.0) == 0.0 and V("JC", 0.5, 0.5, 0.0, 0.0) == 0.0
    assert V("JE", 0.5, 0.5, 4.0, 5.0) == 2.0 and V("JC", 0.5, 0.5, 4.0, 5.0) == 2.5
    assert V("Rd", 5.0, 0.0) == 0.0 and V("Rd", 5.0, 1.0) == P.alpha_C3 * 5.0 and 0.0 < V("Rd", 5.0, 0.5) < P.alpha_C3 * 5.0
    assert V("Ag", 0.5, 0.5, 4.0, 5.0, 0.0) == 0 and math.isfinite(V("Ag", 0.5, 0.5, 4.0, 5.0, 0.5))
    # the smoothed minimum of JE = 2 and JC = 2.5 (haxeltine eq. 2): below both, above neither by much
    ag = V("Ag", 0.5, 0.5, 4.0, 5.0, 1.0)
    s = 4.5
    assert ag == (s - math.sqrt(s * s - 4 * 0.7 * 2.0 * 2.5)) / (2 * 0.7) and 0 < ag < 2.0


# photosynthesis_tests.jl:268-299
def test_respiration_assimilation_switches():
    args = dict(swdown=50.0, pres=1.0e5, co2=400.0, lam=0.5, beta=1.0)
    call = lambda which, T, LAI: V(which, T, args["swdown"], args["pres"], args["co2"], LAI, args["lam"], args["beta"])
    assert call("resp_An", -5.0, 5.0) == 0.0 and call("resp_Rd", -5.0, 5.0) == 0.0          # T_air < -3
    assert call("resp_An", 20.0, 0.0) == 0.0 and call("resp_Rd", 20.0, 0.0) == 0.0          # LAI = 0
    assert math.isfinite(call("resp_An", 20.0, 5.0)) and math.isfinite(call("resp_Rd", 20.0, 5.0))
    assert call("resp_An", 20.0, 5.0) > 0 and call("resp_Rd", 20.0, 5.0) > 0


# test/vegetation/autotrophic_respiration_tests.jl:5-82
def test_autotrophic_respiration():
    assert V("f_temp_air", 10.0, 5.0) > 0 and V("f_temp_soil", 10.0, 5.0) == 0.0
    assert V("f_temp_air", 15.0, 10.0) > 0 and V("f_temp_soil", 15.0, 10.0) > 0
    assert V("f_temp_air", 10.0, 10.0) == V("f_temp_soil", 10.0, 10.0)
    assert V("f_temp_air", 10.0, 10.0) == pytest.approx(1.0, abs=1e-12)                    # exp(308.56 (1/56.02 - 1/56.02))
    assert V("resp10") == 0.066
    assert V("Rm", 20.0, 15.0, 0.2, 1.0, 0.5) > 0 and math.isfinite(V("Rg", 0.5, 0.2)) and V("Rg", 0.5, 0.2) == 0.25 * (0.5 - 0.2)
    assert math.isfinite(V("Ra", 20.0, 15.0, 0.2, 1.0, 0.5, 0.5))
    assert V("NPP", 0.5, 0.3) == 0.5 - 0.3


# test/vegetation/root_distribution_tests.jl:5-15: the root fractions sum to one on UniformSpacing(dz = 0.1, N = 10)
def test_root_fractions_sum_to_one():
    zc = -0.05 - 0.1 * np.arange(10)
    dens = np.array([V("root_density", z) for z in zc])
    R = dens * 0.1
    assert np.sum(R / np.sum(R)) == pytest.approx(1.0)
    assert np.all(np.diff(dens) < 0) and dens[0] == 0.5 * (7.0 * math.exp(7.0 * -0.05) + 2.0 * math.exp(2.0 * -0.05))


# test/vegetation/plant_available_water_tests.jl:38-77: the four cases (porosity 0.5, wp 0.05, fc 0.25)
def test_plant_available_water_cases():
    por = 0.5
    assert V("plant_available_water", por * 1.0 * 1.0) == pytest.approx(1.0)          # fully saturated
    assert V("plant_available_water", por * 1.0 * 0.0) == pytest.approx(0.0)          # dry
    assert V("plant_available_water", por * 0.0 * 1.0) == pytest.approx(0.0)          # frozen
    assert V("plant_available_water", por * 1.0 * 0.2) == pytest.approx(0.25)         # unsaturated


# test/vegetation/vegetation_model_tests.jl + run: the standalone model steps and stays finite; NPP = GPP - Ra etc.
@pytest.mark.parametrize("heun", [False, True])
def test_vegetation_model_steps(heun):
    o = oracle.VegetationOracle(5)
    o.set("carbon_vegetation", np.array([0.5, 1.0, 5.0, 10.0, 20.0]))
    o.set("vegetation_area_fraction", 0.3)
    o.set("air_temperature", np.array([-10.0, 5.0, 15.0, 25.0, 35.0]))
    o.set("surface_shortwave_down", np.array([0.0, 100.0, 300.0, 500.0, 800.0]))
    for _ in range(48):
        o.timestep(1800.0, True, heun)
    for name in oracle.VEG_FIELDS:
        assert np.all(np.isfinite(o.get(name))), name
    assert np.array_equal(o.get("net_primary_production"), o.get("gross_primary_production") - o.get("autotrophic_respiration"))
    assert np.array_equal(o.get("gross_primary_production"), o.get("net_assimilation") * 1.0e-3)
    assert o.get("net_assimilation")[0] == 0.0 and o.get("net_assimilation")[2] > 0       # no light, too cold / productive
    assert np.all(o.get("leaf_area_index") == o.get("balanced_leaf_area_index")) and np.all(o.get("phenology_factor") == 1.0)
    assert o.time() == 48 * 1800.0
