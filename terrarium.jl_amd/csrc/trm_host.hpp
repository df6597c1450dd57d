// trm_host.hpp -- host side shared by the translation units of libterrarium_hip.so: the context, the launch arguments and
// the launch policies (which kernel instance a context takes).  The kernel instantiations are spread over the trm_launch_*.hip
// files (one family per file, compiled in parallel); terrarium_hip.hip holds the context management, the step sequences and the
// C ABI.  Nothing here is part of the ABI.
#pragma once
#include "../../include/terrarium_hip.h"
#include "trm_kernels.hpp"
#include "trm_column.hpp"
#include "trm_vegetation.hpp"

#include <rccl/rccl.h>   // types only: the library is opened lazily by trm_comm_init (no link-time dependency)

#include <cmath>
#include <limits>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

struct FieldSet {
    void* f[TRM_FIELD_COUNT];
    void* kf_top;  // top face (face Nz) of the hydraulic_conductivity Face field, [Nh]
    void* raw[TRM_FIELD_COUNT];   // the allocations behind f[] (f = raw + the field's skew, see alloc_fields)
};

struct trm_ctx {
    int precision = TRM_F64;
    long Nh = 0;
    int Nz = 0, Nzp = 0, device = 0;  // Nzp: level pitch of the z-fastest device layout
    size_t esize = 8;
    trm_params params;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    FieldSet state{}, stage{}, saved{};
    bool has_stage = false, has_saved = false;
    double saved_time = 0.0;
    int64_t saved_iteration = 0;
    uint32_t saved_status = 0;
    bool saved_tend_valid = true;
    void* bc_value[TRM_BCV_COUNT][2] = {};
    int bc_kind[TRM_BCV_COUNT][2] = {};
    // A Gradient condition whose values the library KNOWS to be +0 everywhere (set through trm_set_bc from host values, the device
    // buffer never handed out since): at the BOTTOM its halo, edge + (+0) * (-dz) = edge + (-0), IS the edge value bit for bit -- the
    // halo the branch-free programs form where no condition is set.  The reference's FreeDrainage() (GradientBoundaryCondition(0) on
    // the pressure head, soil_model_bcs.jl:40) is that case, and need not take the generic-boundary kernels (Policy::generic_bcs).
    bool bc_zero_gradient[TRM_BCV_COUNT][2] = {};
    void *d_zC = nullptr, *d_zF = nullptr, *d_dzc = nullptr, *d_rdzc = nullptr, *d_rdzf = nullptr, *d_psiz = nullptr, *d_lvl = nullptr;
    void* d_rootf = nullptr;   // static root fraction per level [Nz] (root_distribution.jl:45-63)
    std::vector<double> h_zF, h_zC, h_dzc, h_dzf;  // as derived in NF, widened
    double dzf_bot = 0, dzf_top = 0, dzc_bot = 0, dzc_top = 0, Az = 1;
    uint32_t* d_status = nullptr;
    // time series input sources: whole series resident on the device, evaluated at the clock every step
    struct Series {
        bool is_bc = false;
        int field = 0, var = 0, side = 0, indexing = 0;
        std::vector<double> times;  // the time levels currently held, oldest first
        void* d_values = nullptr;   // [cap][Nh]: a ring of time levels, level n of `times` in slot (head + n) % cap
        long cap = 0, head = 0;
        long pending_from = -1;     // first level (index into `times`) whose copy may still be in flight, or -1
        bool windowed = false;      // levels have been appended (trm_series_append): trm_series_trim_before may release its head
        bool trimmed = false;       // ... and has: evaluations before `trimmed_before` would need levels that are gone
        double trimmed_before = 0.0;
        size_t slot(int n) const { return (size_t)((head + n) % cap); }
    };
    std::vector<Series> series;
    // trm_series_append: host values are staged through pinned memory and copied on a side stream under the running steps
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_done = nullptr;     // the last appended levels have reached the device
    hipEvent_t copy_order = nullptr;    // the context stream's work at the time levels were last released (trim)
    bool copy_pending = false, order_recorded = false;
    void* h_stage = nullptr;            // pinned staging buffer
    size_t h_stage_cap = 0;
    // ring grid (ColumnRingGrid, column_ring_grid.jl:37-59): column i <-> point ring_index[i] of the full grid
    long ring_points = 0;
    int32_t* d_ring_inv = nullptr;      // [ring_points] column of a grid point, -1 outside the mask
    int32_t* d_ring_idx = nullptr;      // [Nh] grid point of a column
    void* d_ring = nullptr;             // staging [rows][ring_points]
    size_t ring_cap = 0;
    void* bc_value_stage[TRM_BCV_COUNT][2] = {};  // Heun: the stage evaluates its boundary series at t + dt
    // the two-call Heun (trm_heun_predict / trm_heun_correct): stage buffers the caller writes between the two calls
    bool stage_bc_user[TRM_BCV_COUNT][2] = {};    // handed out by trm_stage_bc_device_ptr
    bool stage_vwc_own = false;                   // the stage reads its own per-cell vwc_forcing (else the state's)
    bool heun_pending = false;                    // trm_heun_predict has run, trm_heun_correct has not
    bool heun_stage_aux = false;                  // ... and trm_heun_stage_auxiliary has (compute_auxiliary!(stage) is done)
    double heun_dt = 0.0;
    void* d_top3 = nullptr;  // LandModel: [3][Nh] (T, sat, liq) of the top cell as left by the last fused step
    bool top_valid = false;  // ... and whether they still describe the state (any other writer clears it)
    bool top_escaped = false;  // a device pointer to T / sat / liq was handed out: never trust the copies again
    bool tend_valid = true;    // the tendency fields hold what the reference would (false after a fused step that did not finalize)
    // the stored (temperature, liquid_water_fraction) ARE the energy closure of the stored (internal_energy, saturation):
    // true after a fused step / closure!, false after anything else wrote one of the four fields.  Lets the step derive
    // them in registers instead of reading them (k_column<DERIVE>).
    bool closure_consistent = false, closure_escaped = false, saved_closure_consistent = false;
    void* d_zero = nullptr;  // [Nh] zeros: stands in for the value array of every unset boundary condition
    double* d_reduce = nullptr;  // scratch for trm_reduce
    size_t reduce_cap = 0;
    void* d_io = nullptr;        // staging buffer of trm_upload / trm_download (host layout [rows][Nh])
    size_t io_cap = 0;
    double time = 0.0;
    int64_t iteration = 0;
    int opt_packed = 1;   // fp32: two columns per lane with packed math where the path allows it
    int opt_async = 0, opt_kernel = TRM_KERNEL_FUSED, opt_write_kf = 1, opt_vwc_field = 0;
    int opt_derive = 2;
    int opt_steps_per_launch = 0;   // 0: chosen by the library (auto_steps_per_launch), 1: one launch per step, m > 1: up to m steps per launch
    // Two halves of the columns (TRM_OPT_PIPELINE_PARTS): the per-step LandModel path runs the latency-bound 0-D surface
    // processes of one half in the same launch as the soil columns of the other (k_land_euler).  Columns are independent.
    int opt_pipeline = 2;           // 0: off, 1: whenever legal, 2: auto (column threshold)
    int opt_zero_gradient_fast = 1; // TRM_OPT_ZERO_GRADIENT_FAST: FreeDrainage()-like conditions on the branch-free programs
    int opt_bc_signature = 1;       // TRM_OPT_BC_SIGNATURE: 1 the program with the boundary kinds compiled in where one matches
    int opt_single_step = 2;        // TRM_OPT_SINGLE_STEP_PROGRAM: 0 off, 1 whenever legal, 2 the library's rule
    int part = -1;                  // part the launch helpers currently address (-1: all columns)
    long part_lo[2] = {0, 0}, part_n[2] = {0, 0};
    // Launch arguments (DevParams, View of the state / the stage, StageView) are built once and reused by every launch;
    // any call that changes what they are built from (boundary conditions, options, lazily allocated buffers) clears
    // `args_valid` and the next launch rebuilds them.
    // multi-device diagnostics: one RCCL communicator per context, collectives on a side stream (never on the step path)
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    uint64_t comm_group = 0;    // hash of the ncclUniqueId the communicator was created from (0: none)
    hipStream_t comm_stream = nullptr;
    double* d_comm = nullptr;   // [2 * (Nz + 1) + 8] doubles: send | recv
    // vegetation (trm_set_vegetation)
    int veg_mode = TRM_VEGETATION_OFF;
    trm_vegetation_params veg_params{};
    // multi-step program with time series: device copies of the slot table and the per-step rows
    void* d_series_table = nullptr;
    void* d_series_rows = nullptr;
    size_t series_rows_cap = 0;
    // pinned staging for them: a ring, so that a launch never waits for the stream -- only for the copy that used the same
    // staging buffer four launches ago
    struct RowStage { void* h = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool pending = false; };
    RowStage row_stage[4];
    int row_stage_next = 0;
    // LandModel, TRM_OPT_SURFACE_IN_LAUNCH: the surface processes run in the first workgroups of the step launch (k_column_land) and
    // hand ground heat flux, infiltration and skin temperature to the column workgroups through granules tagged with the launch's epoch
    unsigned debug_handoff_tag_bias = 0;        // TRM_DEBUG_HANDOFF_TAG_BIAS (tests of the bounded wait)
    unsigned long long* d_gran = nullptr;   // [Nh][6], zero at allocation (epoch 0 is never used)
    uint32_t front_epoch = 0;               // epoch of the last such launch
    int opt_front = 2;                      // 0 off, 1 whenever legal, 2 the library's rule
    int last_program = 0;                   // TRM_INFO_LAST_PROGRAM: the kernel instance the last step launch selected
    bool args_valid = false;
    void* args = nullptr;   // LaunchArgs<NF>*, owned
    void (*args_free)(void*) = nullptr;
    std::string err;
};

namespace trmh {
using namespace trm;

// errors: the message stays with the context (trm_last_error); without one, with the calling thread
int fail(trm_ctx* ctx, int code, const std::string& msg);

#define TRM_HIP(ctx, call)                                                                               \
    do {                                                                                                 \
        hipError_t e__ = (call);                                                                         \
        if (e__ != hipSuccess)                                                                           \
            return ::trmh::fail(ctx, TRM_EHIP, std::string(#call) + ": " + hipGetErrorString(e__));     \
    } while (0)

inline long field_rows(const trm_ctx* c, int field) {
    if (field == TRM_FIELD_HYDRAULIC_CONDUCTIVITY) return c->Nz + 1;
    if (field <= TRM_FIELD_TEND_SATURATION_WATER_ICE || field == TRM_FIELD_VWC_FORCING) return c->Nz;
    if (field == TRM_FIELD_PLANT_AVAILABLE_WATER || field == TRM_FIELD_ROOT_FRACTION) return c->Nz;
    return 1;
}
inline bool valid_field(int f) { return f >= 0 && f < TRM_FIELD_COUNT; }
inline bool is_input_field(int f) {
    return (f >= TRM_FIELD_AIR_TEMPERATURE && f <= TRM_FIELD_SURFACE_LONGWAVE_DOWN) || f == TRM_FIELD_ALBEDO || f == TRM_FIELD_EMISSIVITY ||
           (f >= TRM_FIELD_CO2 && f <= TRM_FIELD_VEGETATION_GROUND_TEMPERATURE) || f == TRM_FIELD_STEM_AREA_INDEX;
}
inline bool is_3d(int field) {
    return field <= TRM_FIELD_TEND_SATURATION_WATER_ICE || field == TRM_FIELD_VWC_FORCING || field == TRM_FIELD_PLANT_AVAILABLE_WATER ||
           field == TRM_FIELD_ROOT_FRACTION;
}
// the 3-D vegetation fields exist only once trm_set_vegetation has run
inline bool is_lazy_field(int field) { return field == TRM_FIELD_PLANT_AVAILABLE_WATER || field == TRM_FIELD_ROOT_FRACTION; }
// elements of the device buffer of a field: [Nh][Nzp] for 3-D fields, [Nh] for 2-D fields
inline size_t field_elems(const trm_ctx* c, int field) { return is_3d(field) ? (size_t)c->Nh * c->Nzp : (size_t)c->Nh; }

template <class NF> struct StageView { const NF *bcT_bot, *bcT_top; };
template <class NF> struct LaunchArgs {
    DevParams<NF> p;
    View<NF> state, stage;
    View<NF> part[2];   // the state's view restricted to the two pipeline parts
    StageView<NF> w;
};
// columns the launch helpers currently address: all of them, or one pipeline part
inline long ncols(const trm_ctx* c) { return c->part >= 0 ? c->part_n[c->part] : c->Nh; }
inline long first_col(const trm_ctx* c) { return c->part >= 0 ? c->part_lo[c->part] : 0; }
// Launch arguments (DevParams, View of the state / the stage, StageView) are built once and reused by every launch (defined in
// terrarium_hip.hip, instantiated for double and float)
template <class NF> const LaunchArgs<NF>& launch_args(trm_ctx* c);
template <class NF> const View<NF>& cached_view(trm_ctx* c, const FieldSet& s) {
    const LaunchArgs<NF>& a = launch_args<NF>(c);
    if (&s == &c->stage) return a.stage;
    return c->part >= 0 ? a.part[c->part] : a.state;
}
// the state's view as the step launches see it (one pipeline part, or everything)
template <class NF> const View<NF>& state_view(trm_ctx* c) { return cached_view<NF>(c, c->state); }

inline dim3 cell_grid(const trm_ctx* c, long /*rows*/ = 0) { return dim3((unsigned)(((size_t)c->Nh * c->Nzp + 255) / 256), 1, 1); }
inline dim3 col_grid(const trm_ctx* c) { return dim3((unsigned)((ncols(c) + 255) / 256), 1, 1); }
// lane = level kernels: one column per LPC lanes, 4 waves per workgroup
inline dim3 wave_grid(const trm_ctx* c, int lpc) {
    long waves = (ncols(c) + (64 / lpc) - 1) / (64 / lpc);
    return dim3((unsigned)((waves + 3) / 4), 1, 1);
}
inline dim3 column_grid(const trm_ctx* c, int lpc) {
    dim3 grid = wave_grid(c, lpc);
    grid.x = (grid.x * 4 + (TRM_STEP_BLOCK / 64) - 1) / (TRM_STEP_BLOCK / 64);  // wave_grid counts 4-wave workgroups
    return grid;
}
// addresses one part of the columns for the lifetime of the scope
struct PartScope {
    trm_ctx* c;
    PartScope(trm_ctx* ctx, int q) : c(ctx) { c->part = q; }
    ~PartScope() { c->part = -1; }
};

// Time interpolation indices of a series at time t (defined in terrarium_hip.hip)
void series_time_indices(const std::vector<double>& times, int indexing, double t, int& n1, int& n2, double& f, double& g);

// ---- launch policies: which kernel instance a context takes (pure host logic, shared by every translation unit) -------------
template <class NF> struct Policy {
    static bool richards(const trm_ctx* c) { return c->params.flow == TRM_FLOW_RICHARDS; }
    static bool coupled(const trm_ctx* c) { return c->veg_mode == TRM_VEGETATION_COUPLED; }
    // hydraulics specialisation of this context (trm_device.hpp: HYD_*)
    static int hyd(const trm_ctx* c) {
        if (c->params.swrc == TRM_SWRC_BROOKS_COREY && c->params.unsat_k == TRM_UNSATK_LINEAR) {
            // the compile-time instance is lambda = 0.2 (-1/lambda = -5 exactly, Base's integer power); other lambda: generic
            const PowSpec<NF> spec = make_pow_spec<NF>(NF(-1) / (NF)c->params.bc_lambda);
            return (spec.kind == POW_INT && spec.n == -5) ? HYD_BC_LINEAR : HYD_GENERIC;
        }
        if (c->params.swrc == TRM_SWRC_VAN_GENUCHTEN && c->params.unsat_k == TRM_UNSATK_VAN_GENUCHTEN) {
            // the compile-time instance is van Genuchten's n = 2 (every reference test and example); other n: generic
            // (n = 2 exactly: -1/m = -2 {INT}, 1/n = (n-1)/n = 1/2 {HALVES, 1}, n/(n+1) = RN(2/3) {THIRDS, 2} in make_pow_spec)
            if (c->params.vg_n == 2.0) return HYD_VG_N2;
        }
        return HYD_GENERIC;
    }
    // the branch-free fused kernel covers Value on temperature and Flux on the prognostics; anything else is generic
    static bool generic_bcs(const trm_ctx* c) {
        bool generic = c->opt_vwc_field != 0;   // a per-cell vwc_forcing field is read by the generic instance only
        // (a zero gradient on temperature or pressure head at the bottom is the edge-value halo of the branch-free programs: see
        // trm_ctx::bc_zero_gradient.  Not on saturation / liquid fraction, whose unset halo follows the halo policy, and not at the top,
        // where edge + (+0) * dz turns an edge value of -0.0 into +0.0)
        auto plain = [&](int var, int side) { return side == 0 && c->opt_zero_gradient_fast && c->bc_kind[var][side] == TRM_BC_GRADIENT && c->bc_zero_gradient[var][side]; };
        for (int side = 0; side < 2; ++side) {
            generic = generic || (c->bc_kind[TRM_BCV_TEMPERATURE][side] == TRM_BC_GRADIENT && !plain(TRM_BCV_TEMPERATURE, side));
            for (int var : {TRM_BCV_SATURATION_WATER_ICE, TRM_BCV_LIQUID_WATER_FRACTION, TRM_BCV_PRESSURE_HEAD})
                generic = generic || c->bc_kind[var][side] == TRM_BC_VALUE ||
                          (c->bc_kind[var][side] == TRM_BC_GRADIENT && !(var == TRM_BCV_PRESSURE_HEAD && plain(var, side)));
        }
        return generic;
    }
    // Deriving T and liq in registers saves 2 of 11 field accesses and costs ~40 instructions per cell.  Measured on MI355X
    // (profiles/r03/exp4_ab_derive.log, interleaved medians on one box; fp64): 8 x N145 (HBM-resident) 215 vs 261 us, N145
    // 25.1 vs 27.3 us with the reference-default hydraulics, 33.6 vs 35.1 (LandModel), 34.4 vs 35.6 (LandModel, van Genuchten);
    // it loses on small grids (N72 heat-only: 7.1 vs 6.6 us, latency-bound) and for the packed fp32 kernel, which is not short of
    // bytes (C5: liquid fraction alone 523 vs 500 us, both 562 vs 533).  Deriving the liquid fraction alone (mode 3: one read
    // less, the temperature divide saved) sits between the two everywhere (8 x N145: 238 us) and is kept as an option only.
    // AUTO (2): fp64 states beyond the Infinity Cache, or of >= 24 576 columns; fp32 states beyond the cache on the packed kernel:
    // the liquid fraction alone (the numbers above for the packed kernel predate the store ordering of round 3; see below).
    template <bool RICH> static int derive_now(const trm_ctx* c) {
        // (the coupled vegetation reads T and liq of the whole column from memory every step)
        if (!c->closure_consistent || c->closure_escaped || coupled(c) || c->opt_derive == 0) return DERIVE_NONE;
        if (c->opt_derive == 1) return DERIVE_T_LIQ;
        // (3, 4: the packed fp32 step's modes -- the liquid fraction alone; that and the pressure head.  The fp64 column program had
        // instances for "liquid fraction alone" and "pressure head as well" (value 5) until round 5: both measured slower than
        // deriving T and liq, EXPERIMENTS.md; the values now select what the library offers there: both T and liq)
        const bool packed = std::is_same<NF, float>::value && packed_path(const_cast<trm_ctx*>(c));
        if (c->opt_derive == 3) return packed ? DERIVE_LIQ : DERIVE_T_LIQ;
        if (c->opt_derive == 5) return DERIVE_T_LIQ;
        if (c->opt_derive == 4) return packed ? (RICH ? DERIVE_LIQ_PSI : DERIVE_LIQ) : DERIVE_T_LIQ;
        const size_t state_bytes = (size_t)(RICH ? 6 : 4) * (size_t)c->Nh * (size_t)c->Nzp * sizeof(NF);
        const bool beyond_cache = state_bytes > ((size_t)256 << 20);
        const bool large = c->Nh >= 24576;
        // fp32 on the packed kernel, HBM-resident: the liquid fraction alone (r3, re-measured on the final kernels,
        // profiles/r03/exp21_derive_liq_fp32.log: C5 443.7 vs 457.9 us, C5-VG 472.5 vs 476.3; before the store ordering it lost)
        if (std::is_same<NF, float>::value) return (beyond_cache && packed_path(const_cast<trm_ctx*>(c))) ? DERIVE_LIQ : DERIVE_NONE;
        return (beyond_cache || large) ? DERIVE_T_LIQ : DERIVE_NONE;
    }
    // The per-column outputs of the column program through the workgroup's staging table (template parameter STAGED) or as direct 2-lane
    // stores.  Measured (profiles/r03/exp20_staged_small_stores.log, same box, alternating): staged wins where the state streams
    // from HBM (8 x N145: 201.7 vs 212.9 us, -5.3 %) and on the LandModel with its seven outputs (C4 33.7 vs 34.4), it loses
    // where the step is launch- and latency-bound (C3 25.2 vs 24.7, N72 heat-only 7.3 vs 6.7): the barrier in front of the
    // staged store.  TRM_STAGED_SMALL = 0 / 1 in the environment forces it (experiments).
    // The per-column inputs of the column program through the scalar memory path: cache-resident states (see column_program).
    // TRM_SCALAR_INPUTS = 0 / 1 in the environment forces it (experiments, tests).
    template <bool RICH> static int scalar_inputs_now(const trm_ctx* c) {
        static const int forced = [] { const char* e = std::getenv("TRM_SCALAR_INPUTS"); return e ? std::atoi(e) : -1; }();
        if (forced >= 0) return forced != 0;
        const size_t state_bytes = (size_t)(RICH ? 6 : 4) * (size_t)c->Nh * (size_t)c->Nzp * sizeof(NF);
        return state_bytes <= ((size_t)256 << 20) ? 1 : 0;
    }
    // The packed fp32 step does not gain (C5 472 vs 468 us, C5-VG 500 vs 487; exp20b): staging is off there unless forced.
    template <bool RICH> static int staged_now(const trm_ctx* c, bool packed = false) {
        static const int forced = [] { const char* e = std::getenv("TRM_STAGED_SMALL"); return e ? std::atoi(e) : -1; }();
        if (forced >= 0) return forced != 0;
        if (packed) return 0;
        const size_t state_bytes = (size_t)(RICH ? 6 : 4) * (size_t)c->Nh * (size_t)c->Nzp * sizeof(NF);
        const bool beyond_cache = state_bytes > ((size_t)256 << 20);
        return (beyond_cache || (c->params.seb != 0 && c->Nh >= 24576)) ? 1 : 0;
    }
    // The (STAGED, SCALAR_IN) combinations that have instances (round 5: the ones no rule selects were removed).  The rules above
    // give (0, 1) for cache-resident states, (1, 1) for large cache-resident LandModels, (1, 0) beyond the cache; the environment
    // switches of the tests can ask for anything: (0, 0) -- direct 2-lane stores AND vector loads of one address -- has no
    // instance anywhere (-> (0, 1)); (1, 1) exists where a surface energy balance can run: the LandModel signature and the
    // programs that read the kinds at run time (elsewhere -> (1, 0)).
    static void io_paths(bool land_or_runtime_kinds, int& staged, int& scalar_in) {
        if (!staged && !scalar_in) scalar_in = 1;
        if (staged && scalar_in && !land_or_runtime_kinds) scalar_in = 0;
    }
    // fp32: two columns per lane with packed math (trm_packed_f32.hpp) -- the reference-default hydraulics, and van
    // Genuchten retention with Mualem conductivity
    static bool packed_path(trm_ctx* c) {
        if (!std::is_same<NF, float>::value || !c->opt_packed || generic_bcs(c)) return false;
        if (hyd(c) == HYD_VG_N2) return true;
        return hyd(c) == HYD_BC_LINEAR;
    }
    // columns of 65 ... 128 levels: two levels per lane, one column per wavefront (trm_column_deep.hpp)
    static bool deep_columns(const trm_ctx* c) { return c->Nz > 64 && c->Nz <= 128; }
    // columns of 129 ... 256 levels: four levels per lane (trm_column_wide.hpp)
    static bool wide_columns(const trm_ctx* c) { return c->Nz > 128 && c->Nz <= 256; }
    // slot of the multi-step program a series feeds, or -1 when the program cannot take it (the step then runs per launch)
    static int series_slot(const trm_ctx* c, const trm_ctx::Series& sr) {
        if (sr.is_bc) {
            const int kind = c->bc_kind[sr.var][sr.side];
            if (sr.var == TRM_BCV_TEMPERATURE && kind == TRM_BC_VALUE) return sr.side == TRM_TOP ? SLOT_T_TOP : SLOT_T_BOT;
            if (sr.var == TRM_BCV_INTERNAL_ENERGY && kind == TRM_BC_FLUX && !(c->params.seb && sr.side == TRM_TOP)) return sr.side == TRM_TOP ? SLOT_FU_TOP : SLOT_FU_BOT;
            if (sr.var == TRM_BCV_SATURATION_WATER_ICE && kind == TRM_BC_FLUX && richards(c) && !(c->params.seb && sr.side == TRM_TOP)) return sr.side == TRM_TOP ? SLOT_FS_TOP : SLOT_FS_BOT;
            return -1;
        }
        if (!c->params.seb) return -1;     // (inputs nobody reads: leave them to update_inputs!)
        switch (sr.field) {
            case TRM_FIELD_AIR_TEMPERATURE: return SLOT_TAIR;
            case TRM_FIELD_AIR_PRESSURE: return SLOT_PRES;
            case TRM_FIELD_WINDSPEED: return SLOT_WIND;
            case TRM_FIELD_SPECIFIC_HUMIDITY: return SLOT_QAIR;
            case TRM_FIELD_RAINFALL: return SLOT_RAIN;
            case TRM_FIELD_SURFACE_SHORTWAVE_DOWN: return SLOT_SWD;
            case TRM_FIELD_SURFACE_LONGWAVE_DOWN: return SLOT_LWD;
            case TRM_FIELD_ALBEDO: return c->params.prescribed_albedo ? SLOT_ALBEDO : -1;
            case TRM_FIELD_EMISSIVITY: return c->params.prescribed_albedo ? SLOT_EMISSIVITY : -1;
            default: return -1;
        }
    }
    static bool series_fit_program(const trm_ctx* c) {
        for (const auto& sr : c->series)
            if (series_slot(c, sr) < 0) return false;
        return true;
    }
    static VegDev<NF> veg_dev(const trm_ctx* c) {
        VegDev<NF> p;
        const double* s = &c->veg_params.tau25;
        NF* t = &p.tau25;
        for (int n = 0; n < 40; ++n) t[n] = (NF)s[n];
        p.eps_mw = (NF)c->params.eps_mw;
        p.one_minus_eps_mw = NF(1) - p.eps_mw;
        p.sqrt_eps = std::sqrt(std::numeric_limits<NF>::epsilon());
        p.paw_span = p.field_capacity - p.wilting_point;
        p.rpaw_span = NF(1) / p.paw_span;
        p.ln_q10_tau = std::log(p.q10_tau); p.ln_q10_Kc = std::log(p.q10_Kc); p.ln_q10_Ko = std::log(p.q10_Ko);
        p.ts_k1 = NF(2) * std::log(NF(1) / NF(0.99) - NF(1)) / (p.T_CO2_low - p.T_photos_low);     // photosynthesis.jl:165-188
        p.ts_k2 = NF(0.5) * (p.T_CO2_low + p.T_photos_low);
        p.ts_k3 = std::log(NF(0.99) / NF(0.01)) / (p.T_CO2_high - p.T_photos_high);
        return p;
    }
    static VegView<NF> veg_view(const trm_ctx* c) { return veg_view(c, c->state); }
    static VegView<NF> veg_view(const trm_ctx* c, const FieldSet& s) {
        VegView<NF> v;
        auto F = [&](int id) { return (NF*)s.f[id]; };
        v.Nh = c->Nh;
        v.C_veg = F(TRM_FIELD_CARBON_VEGETATION); v.nu = F(TRM_FIELD_VEGETATION_AREA_FRACTION);
        v.G_C_veg = F(TRM_FIELD_TEND_CARBON_VEGETATION); v.G_nu = F(TRM_FIELD_TEND_VEGETATION_AREA_FRACTION);
        v.LAI_b = F(TRM_FIELD_BALANCED_LEAF_AREA_INDEX); v.phen = F(TRM_FIELD_PHENOLOGY_FACTOR); v.LAI = F(TRM_FIELD_LEAF_AREA_INDEX);
        v.gw_can = F(TRM_FIELD_CANOPY_WATER_CONDUCTANCE); v.lambda_c = F(TRM_FIELD_LEAF_TO_AIR_CO2_RATIO);
        v.An = F(TRM_FIELD_NET_ASSIMILATION); v.Rd = F(TRM_FIELD_LEAF_RESPIRATION); v.GPP = F(TRM_FIELD_GROSS_PRIMARY_PRODUCTION);
        v.Ra = F(TRM_FIELD_AUTOTROPHIC_RESPIRATION); v.NPP = F(TRM_FIELD_NET_PRIMARY_PRODUCTION);
        v.Tair = F(TRM_FIELD_AIR_TEMPERATURE); v.pres = F(TRM_FIELD_AIR_PRESSURE); v.qair = F(TRM_FIELD_SPECIFIC_HUMIDITY);
        v.swd = F(TRM_FIELD_SURFACE_SHORTWAVE_DOWN); v.CO2 = F(TRM_FIELD_CO2); v.smlf = F(TRM_FIELD_SOIL_MOISTURE_LIMITING_FACTOR);
        v.daily_Rd = F(TRM_FIELD_DAILY_LEAF_RESPIRATION);
        v.Tground = F(TRM_FIELD_VEGETATION_GROUND_TEMPERATURE);
        v.Tground_stride = 1;
        const bool canopy = coupled(c);
        auto G = [&](int id) { return canopy ? F(id) : (NF*)nullptr; };
        v.w_can = G(TRM_FIELD_CANOPY_WATER); v.G_w_can = G(TRM_FIELD_TEND_CANOPY_WATER); v.I_can = G(TRM_FIELD_CANOPY_WATER_INTERCEPTION);
        v.R_can = G(TRM_FIELD_CANOPY_WATER_REMOVAL); v.f_can = G(TRM_FIELD_SATURATION_CANOPY_WATER); v.rain_ground = G(TRM_FIELD_RAINFALL_GROUND);
        v.E_can = G(TRM_FIELD_EVAPORATION_CANOPY); v.transp = G(TRM_FIELD_TRANSPIRATION); v.SAI = G(TRM_FIELD_STEM_AREA_INDEX);
        v.paw = F(TRM_FIELD_PLANT_AVAILABLE_WATER);
        v.rootf = (const NF*)c->d_rootf;   // static: one copy serves the stage as well
        if (c->part >= 0 && &s == &c->state) {   // one pipeline part: columns [lo, lo + n)
            const long lo = first_col(c);
            v.Nh = ncols(c);
            for (NF** q : {&v.C_veg, &v.nu, &v.G_C_veg, &v.G_nu, &v.LAI_b, &v.phen, &v.LAI, &v.gw_can, &v.lambda_c, &v.An, &v.Rd, &v.GPP, &v.Ra, &v.NPP,
                           &v.w_can, &v.G_w_can, &v.I_can, &v.R_can, &v.f_can, &v.rain_ground, &v.E_can, &v.transp})
                if (*q) *q += lo;
            for (const NF** q : {&v.Tair, &v.pres, &v.qair, &v.swd, &v.CO2, &v.smlf, &v.daily_Rd, &v.Tground, &v.SAI})
                if (*q) *q += lo * (q == &v.Tground ? v.Tground_stride : 1);
            if (v.paw) v.paw += lo * c->Nzp;
        }
        return v;
    }
};
// the context's boundary kinds as a BCSIG signature (trm_kernels.hpp); meaningful where the branch-free kinds hold (Policy::generic_bcs)
inline int bc_signature_of(const trm_ctx* c) {
    const bool rich = c->params.flow == TRM_FLOW_RICHARDS;
    int sig = (c->bc_kind[TRM_BCV_TEMPERATURE][0] == TRM_BC_VALUE ? BCSIG_T_BOT : 0) | (c->bc_kind[TRM_BCV_TEMPERATURE][1] == TRM_BC_VALUE ? BCSIG_T_TOP : 0) |
              (c->bc_kind[TRM_BCV_INTERNAL_ENERGY][0] == TRM_BC_FLUX ? BCSIG_FU_BOT : 0) | ((rich && c->bc_kind[TRM_BCV_SATURATION_WATER_ICE][0] == TRM_BC_FLUX) ? BCSIG_FS_BOT : 0);
    if (c->params.seb) return sig | BCSIG_LAND;      // (the LandModel's wiring owns the top flux conditions: land_model.jl:56-61)
    return sig | (c->bc_kind[TRM_BCV_INTERNAL_ENERGY][1] == TRM_BC_FLUX ? BCSIG_FU_TOP : 0) | ((rich && c->bc_kind[TRM_BCV_SATURATION_WATER_ICE][1] == TRM_BC_FLUX) ? BCSIG_FS_TOP : 0);
}
// TRM_INFO_LAST_PROGRAM: which kernel instance a step launch selected -- family | HYD << 8 | (LPC / 32) << 10 | DERIVE << 12 |
// STAGED << 15 | SCALAR_IN << 16 | (BCSIG + 1) << 17 (0 there: the kinds are read at run time)
inline int program_id(int family, int hyd, int lpc, int derive, int staged, int scalar_in, int bcsig) {
    return family | (hyd << 8) | ((lpc / 32) << 10) | (derive << 12) | ((staged ? 1 : 0) << 15) | ((scalar_in ? 1 : 0) << 16) | ((bcsig + 1) << 17);
}
#define TRM_BY_HYD(c, CALL)                                   \
    switch (::trmh::Policy<NF>::hyd(c)) {                     \
        case HYD_BC_LINEAR: { constexpr int H = HYD_BC_LINEAR; CALL; } break; \
        case HYD_VG_N2: { constexpr int H = HYD_VG_N2; CALL; } break;         \
        default: { constexpr int H = HYD_GENERIC; CALL; } break;             \
    }

// ---- the launchers: declared here, defined and explicitly instantiated in the trm_launch_*.hip files ------------------------
// reference-order kernels, the 0-D surface kernel, update_inputs! of the time series (trm_launch_unfused.hip)
template <class NF> struct Unfused {
    static int await_levels(trm_ctx* c, trm_ctx::Series& sr, int last_level);
    static int update_inputs(trm_ctx* c, const FieldSet& s, double time);
    static int hydraulics(trm_ctx* c, const FieldSet& s);
    static int surface(trm_ctx* c, const FieldSet& s, bool from_state = false);
    static int compute_auxiliary(trm_ctx* c, const FieldSet& s);
    static int compute_tendencies(trm_ctx* c, const FieldSet& s);
    static int reset_tendencies(trm_ctx* c, const FieldSet& s);
    static int update_state(trm_ctx* c, const FieldSet& s, bool tendencies);
    static int explicit_step(trm_ctx* c, const FieldSet& s, double dt);
    static int closure_hydrology(trm_ctx* c, const FieldSet& s, bool with_psi, bool with_adjust = true);
    static int closure(trm_ctx* c, const FieldSet& s);
    static int invclosure(trm_ctx* c, const FieldSet& s);
    static int initialize(trm_ctx* c);
    static int average(trm_ctx* c, int field);
};
// vegetation and the vegetation-coupled surface kernel (trm_launch_vegetation.hip)
template <class NF> struct Veg {
    static int surface_veg(trm_ctx* c, const FieldSet& s, bool from_state, bool advance, double dt, bool store_paw = true);
    static int surface_veg_launch(trm_ctx* c, const View<NF>& v, const VegView<NF>& vv, const SurfaceVegArgs<NF>& a);
    static int vegetation(trm_ctx* c, const FieldSet& s, int mode, double dt, int nsteps, int finalize);   // k_vegetation<MODE>
    static int plant_available_water(trm_ctx* c, const FieldSet& s, bool store_paw);
    static int heun_average_0d(trm_ctx* c, const VegView<NF>& vs, const VegView<NF>& vg, double dt);
};
// the register-resident column programs k_column (trm_launch_column*.hip: one file per precision x program)
template <class NF, bool RICH, int PROG> struct ColumnLaunch { static int run(trm_ctx* c, double dt, int finalize, int nsteps); };
// the ForwardEuler program with the derivation of T / liq and the boundary-condition signature compiled in (BCSIG, trm_kernels.hpp):
// one explicit instantiation per signature in the trm_launch_column_sig_*.hip files; `supported` lists them.
template <class NF, bool RICH, int SIG> struct ColumnSigLaunch {
    static void run(trm_ctx* c, const View<NF>& v, const DevParams<NF>& p, const ColumnArgs<NF>& a, dim3 grid, dim3 block, int lpc, int derive, int staged, int scalar_in);
};
// the same for the one-launch Heun program (trm_launch_column_sig_heun_*.hip)
template <class NF, bool RICH, int SIG> struct ColumnSigHeunLaunch {
    static void run(trm_ctx* c, const View<NF>& v, const DevParams<NF>& p, const ColumnArgs<NF>& a, dim3 grid, dim3 block, int lpc);
};
// the signatures that have instances: launches L<NF, RICH, sig>::run(args...) and returns true, or returns false (the caller then takes
// the program that reads the kinds at run time)
template <template <class, bool, int> class L, class NF, bool RICH, class... A> bool launch_by_signature(int sig, A&&... args) {
    if constexpr (!std::is_same<NF, double>::value) return false;
    else switch (sig) {
        case 0: L<NF, RICH, 0>::run(args...); return true;
        case BCSIG_T_TOP: L<NF, RICH, BCSIG_T_TOP>::run(args...); return true;
        case BCSIG_T_TOP | BCSIG_FU_BOT: L<NF, RICH, BCSIG_T_TOP | BCSIG_FU_BOT>::run(args...); return true;
        case BCSIG_LAND:
            if constexpr (RICH) { L<NF, RICH, BCSIG_LAND>::run(args...); return true; }
            return false;
        case BCSIG_T_TOP | BCSIG_FS_TOP:      // prescribed surface temperature + InfiltrationFlux (soil_model_bcs.jl:28)
            if constexpr (RICH) { L<NF, RICH, BCSIG_T_TOP | BCSIG_FS_TOP>::run(args...); return true; }
            return false;
        default: return false;
    }
}
// the signature instances exist for the two compiled hydraulics (the reference default; van Genuchten n = 2): a context with
// run-time exponents takes the program that reads the kinds at run time too
#define TRM_BY_COMPILED_HYD(c, CALL)                          \
    switch (::trmh::Policy<NF>::hyd(c)) {                     \
        case HYD_BC_LINEAR: { constexpr int H = HYD_BC_LINEAR; CALL; } break; \
        default: { constexpr int H = HYD_VG_N2; CALL; } break;               \
    }
// generic boundary kinds: k_step_wave (Euler) and k_heun_generic (trm_launch_generic*.hip)
template <class NF> struct GenericLaunch {
    static int step(trm_ctx* c, double dt, int finalize);
    static int heun(trm_ctx* c, double dt, int finalize);
};
// columns of 65 ... 128 levels: k_column_deep (trm_launch_deep_f64.hip / _f32.hip)
template <class NF> struct DeepLaunch { static int run(trm_ctx* c, int prog, bool generic, double dt, int finalize, int nsteps); };
// columns of 129 ... 256 levels, four levels per lane: k_column_wide (trm_launch_wide_f64.hip / _f32.hip)
template <class NF> struct WideLaunch { static int run(trm_ctx* c, int prog, bool generic, double dt, int finalize); };
// interleaved LandModel launches: k_land_euler (fp64, trm_launch_land.hip) / k_land_pk (fp32, trm_launch_packed.hip)
template <class NF> struct LandLaunch { static int run(trm_ctx* c, int qcol, int qsurf, double dt, int finalize, bool top_arrays); };
template <> int LandLaunch<double>::run(trm_ctx* c, int qcol, int qsurf, double dt, int finalize, bool top_arrays);
template <> int LandLaunch<float>::run(trm_ctx* c, int qcol, int qsurf, double dt, int finalize, bool top_arrays);
// the packed fp32 step k_step_pk (trm_launch_packed.hip)
struct PackedLaunch {
    static int step(trm_ctx* c, double dt, int finalize);
    static int step_land(trm_ctx* c, double dt, int finalize);      // k_step_pk_land: the surface processes in the launch
};
// the granule buffer and the epoch of the next launch that carries its own surface processes (k_column_land, k_step_pk_land)
int front_epoch_next(trm_ctx* c);
// the LandModel's per-step launch with the surface processes in its first workgroups: k_column_land (fp64; trm_launch_column_land_*.hip)
struct FrontLaunch {
    static int run(trm_ctx* c, double dt, int finalize, bool heun = false);
    template <int H> static int run_hyd(trm_ctx* c, double dt, int finalize, bool heun);
};

}  // namespace trmh
