/* abi_host.c -- a plain C host of libterrarium_hip.so: the closest stand-in for a Julia `ccall` this image allows.
 * TEST INFRASTRUCTURE.  Built with gcc against include/terrarium_hip.h alone (no HIP, no C++, no Python); the library is
 * opened with dlopen like a foreign-language host would.
 *
 *   abi_host layout <libterrarium_hip.so>
 *       prints sizeof / offsetof of trm_grid, trm_params, trm_vegetation_params as this C compiler lays them out, one
 *       "struct field offset size" line each, and the library's trm_abi_version -- compared by tests/test_host_and_abi.py
 *       with the field order INTEGRATION.md's Julia structs declare and with the ctypes mirror.
 *   abi_host run <libterrarium_hip.so> <case.bin> <result.bin>
 *       reads a soil heat case written by the Python side (header of int64: Nh, Nz, nsteps, richards; doubles: dt,
 *       thickness[Nz], temperature[Nz][Nh], saturation[Nz][Nh], top temperature values[Nh]), runs
 *       trm_create -> trm_upload -> trm_set_bc -> trm_initialize -> trm_step(dt, nsteps, 1) -> trm_download and writes
 *       internal_energy, temperature, liquid_water_fraction [Nz][Nh] each, then the clock and the status word, as doubles.
 *       The Python side compares them with the oracle's numbers (tests/test_gpu_abi_host.py).
 */
#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/terrarium_hip.h"

#define FIELD(S, f) printf(#S " " #f " %zu %zu\n", offsetof(S, f), sizeof(((S*)0)->f))

static void* must_sym(void* lib, const char* name) {
    void* p = dlsym(lib, name);
    if (!p) { fprintf(stderr, "abi_host: %s missing: %s\n", name, dlerror()); exit(3); }
    return p;
}

static int layout(void* lib) {
    int (*abi)(void) = (int (*)(void))must_sym(lib, "trm_abi_version");
    printf("abi_version %d %d\n", abi(), TRM_ABI_VERSION);
    printf("sizeof trm_grid %zu\nsizeof trm_params %zu\nsizeof trm_vegetation_params %zu\n", sizeof(trm_grid), sizeof(trm_params),
           sizeof(trm_vegetation_params));
    FIELD(trm_grid, precision); FIELD(trm_grid, num_layers); FIELD(trm_grid, num_columns); FIELD(trm_grid, thickness);
    FIELD(trm_grid, dx); FIELD(trm_grid, device); FIELD(trm_grid, reserved);
#define P(f) FIELD(trm_params, f)
    P(rho_w); P(rho_i); P(rho_a); P(c_a); P(Lsl); P(Llg); P(Lsg); P(g); P(Tref); P(sigma); P(kappa_vk); P(eps_mw); P(R_a);
    P(k_water); P(k_ice); P(k_air); P(k_mineral); P(k_organic); P(c_water); P(c_ice); P(c_air); P(c_mineral); P(c_organic);
    P(por_mineral); P(por_organic); P(rho_soc); P(rho_org);
    P(K_sat); P(theta_res); P(bc_psi_s); P(bc_lambda); P(vg_alpha); P(vg_n); P(impedance); P(vwc_forcing);
    P(albedo); P(emissivity); P(kappa_s); P(C_h); P(min_windspeed); P(tau_r); P(beta_evap); P(field_capacity);
    P(flow); P(swrc); P(unsat_k); P(seb); P(halo_policy); P(prescribed_albedo); P(evap_resistance); P(reserved);
#undef P
#define V(f) FIELD(trm_vegetation_params, f)
    V(tau25); V(Kc25); V(Ko25); V(q10_tau); V(q10_Kc); V(q10_Ko); V(alpha_leaf); V(alpha_a); V(alpha_C3); V(cq); V(k_ext);
    V(T_CO2_high); V(T_CO2_low); V(T_photos_high); V(T_photos_low); V(theta_r); V(g1); V(g_min); V(cn_sapwood); V(cn_root); V(aws);
    V(SLA); V(awl); V(LAI_min); V(LAI_max); V(gamma_L); V(gamma_R); V(gamma_S); V(nu_seed); V(gamma_v_min); V(root_a); V(root_b);
    V(wilting_point); V(field_capacity); V(C_mass); V(alpha_int); V(canopy_k_ext); V(w_can_max); V(tau_w); V(C_can);
#undef V
    /* defaults through the ABI: a value in the middle and the last members, to catch a shifted layout */
    int (*defaults)(trm_params*) = (int (*)(trm_params*))must_sym(lib, "trm_default_params");
    trm_params p;
    memset(&p, 0xff, sizeof p);
    if (defaults(&p) != 0) return 4;
    printf("default rho_w %.17g\ndefault K_sat %.17g\ndefault field_capacity %.17g\ndefault halo_policy %d\ndefault reserved %d\n",
           p.rho_w, p.K_sat, p.field_capacity, p.halo_policy, p.reserved);
    int (*vdefaults)(trm_vegetation_params*) = (int (*)(trm_vegetation_params*))must_sym(lib, "trm_default_vegetation_params");
    trm_vegetation_params vp;
    if (vdefaults(&vp) != 0) return 4;
    printf("default tau25 %.17g\ndefault C_can %.17g\n", vp.tau25, vp.C_can);
    return 0;
}

#define CALL(ctx, expr)                                                                              \
    do {                                                                                             \
        int rc__ = (expr);                                                                           \
        if (rc__ != 0) {                                                                             \
            fprintf(stderr, "abi_host: %s -> %d: %s\n", #expr, rc__, last_error(ctx));               \
            return 5;                                                                                \
        }                                                                                            \
    } while (0)

static int run(void* lib, const char* in_path, const char* out_path) {
    int (*defaults)(trm_params*) = (int (*)(trm_params*))must_sym(lib, "trm_default_params");
    int (*create)(const trm_grid*, const trm_params*, trm_ctx**) = (int (*)(const trm_grid*, const trm_params*, trm_ctx**))must_sym(lib, "trm_create");
    int (*destroy)(trm_ctx*) = (int (*)(trm_ctx*))must_sym(lib, "trm_destroy");
    const char* (*last_error)(const trm_ctx*) = (const char* (*)(const trm_ctx*))must_sym(lib, "trm_last_error");
    int (*upload)(trm_ctx*, int, const void*) = (int (*)(trm_ctx*, int, const void*))must_sym(lib, "trm_upload");
    int (*download)(trm_ctx*, int, void*) = (int (*)(trm_ctx*, int, void*))must_sym(lib, "trm_download");
    int (*set_bc)(trm_ctx*, int, int, int, const void*, double) = (int (*)(trm_ctx*, int, int, int, const void*, double))must_sym(lib, "trm_set_bc");
    int (*initialize)(trm_ctx*) = (int (*)(trm_ctx*))must_sym(lib, "trm_initialize");
    int (*step)(trm_ctx*, double, int, int) = (int (*)(trm_ctx*, double, int, int))must_sym(lib, "trm_step");
    int (*clock_)(const trm_ctx*, double*, int64_t*) = (int (*)(const trm_ctx*, double*, int64_t*))must_sym(lib, "trm_clock");
    int (*status)(trm_ctx*, uint32_t*) = (int (*)(trm_ctx*, uint32_t*))must_sym(lib, "trm_status");

    FILE* f = fopen(in_path, "rb");
    if (!f) { perror(in_path); return 2; }
    int64_t head[4];
    double dt;
    if (fread(head, sizeof(int64_t), 4, f) != 4 || fread(&dt, sizeof(double), 1, f) != 1) return 2;
    const int64_t Nh = head[0], Nz = head[1], nsteps = head[2], richards = head[3];
    const size_t cells = (size_t)(Nh * Nz);
    double* thickness = (double*)malloc(sizeof(double) * (size_t)Nz);
    double* T = (double*)malloc(sizeof(double) * cells);
    double* sat = (double*)malloc(sizeof(double) * cells);
    double* Ttop = (double*)malloc(sizeof(double) * (size_t)Nh);
    if (fread(thickness, sizeof(double), (size_t)Nz, f) != (size_t)Nz || fread(T, sizeof(double), cells, f) != cells ||
        fread(sat, sizeof(double), cells, f) != cells || fread(Ttop, sizeof(double), (size_t)Nh, f) != (size_t)Nh)
        return 2;
    fclose(f);

    trm_params p;
    if (defaults(&p) != 0) return 4;
    p.flow = richards ? TRM_FLOW_RICHARDS : TRM_FLOW_NOFLOW;
    trm_grid g;
    memset(&g, 0, sizeof g);
    g.precision = TRM_F64;
    g.num_layers = (int32_t)Nz;
    g.num_columns = Nh;
    g.thickness = thickness;
    g.dx = 0.0;
    g.device = 0;
    trm_ctx* ctx = NULL;
    CALL(NULL, create(&g, &p, &ctx));
    CALL(ctx, upload(ctx, TRM_FIELD_TEMPERATURE, T));
    CALL(ctx, upload(ctx, TRM_FIELD_SATURATION_WATER_ICE, sat));
    CALL(ctx, set_bc(ctx, TRM_BCV_TEMPERATURE, TRM_TOP, TRM_BC_VALUE, Ttop, 0.0));
    CALL(ctx, initialize(ctx));
    CALL(ctx, step(ctx, dt, (int)nsteps, 1));
    double* out = (double*)malloc(sizeof(double) * cells);
    FILE* o = fopen(out_path, "wb");
    if (!o) { perror(out_path); return 2; }
    const int fields[3] = {TRM_FIELD_INTERNAL_ENERGY, TRM_FIELD_TEMPERATURE, TRM_FIELD_LIQUID_WATER_FRACTION};
    for (int n = 0; n < 3; ++n) {
        CALL(ctx, download(ctx, fields[n], out));
        fwrite(out, sizeof(double), cells, o);
    }
    double t;
    int64_t it;
    uint32_t flags;
    CALL(ctx, clock_(ctx, &t, &it));
    CALL(ctx, status(ctx, &flags));
    const double tail[3] = {t, (double)it, (double)flags};
    fwrite(tail, sizeof(double), 3, o);
    fclose(o);
    CALL(ctx, destroy(ctx));
    free(thickness); free(T); free(sat); free(Ttop); free(out);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: abi_host layout LIB | run LIB case.bin result.bin\n"); return 1; }
    void* lib = dlopen(argv[2], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "abi_host: %s\n", dlerror()); return 3; }
    if (!strcmp(argv[1], "layout")) return layout(lib);
    if (!strcmp(argv[1], "run") && argc == 5) return run(lib, argv[3], argv[4]);
    return 1;
}
