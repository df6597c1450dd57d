// trm_device.hpp -- device-side scalar arithmetic of the SoilModel step path.
//
// Every function names the reference lines whose arithmetic (operation order
// included) it carries.  The library is compiled with -ffp-contract=off: Julia
// never fuses a*b+c on its own, and fma is used only where Julia Base does
// (the compensated integer power).  Templated on the number format NF.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TRM_DEV __device__ __forceinline__
#define TRM_HD __host__ __device__ __forceinline__
// Phase markers for the per-phase instruction budget (profiles/tools/isa_phases.py).  With -DTRM_PHASE_MARKERS every marker
// leaves a comment line in the device assembly AND makes the values that are live at the boundary opaque to the optimiser
// (an empty volatile asm with "+v" operands), so that no arithmetic crosses it: the instructions between two comment lines are
// the phase's.  The marker build is a measuring instrument -- its schedule differs from the shipped kernel's, its instruction
// counts per phase are the shipped kernel's up to what the optimiser shares ACROSS phases (reported next to the shipped
// total).  Without the flag the markers vanish.
#ifdef TRM_PHASE_MARKERS
#define TRM_PHASE(name) asm volatile("; TRM_PHASE " name)
namespace trm {
template <class T> __device__ __forceinline__ void phase_opaque(T& x) { asm volatile("" : "+v"(x)); }
template <class... T> __device__ __forceinline__ void phase_fence_values(T&... x) { (phase_opaque(x), ...); }
}
// (the comment FIRST: volatile asms keep their order, so everything that depends on a fenced value follows the comment line)
#define TRM_PHASE_FENCE(name, ...)                 \
    do {                                           \
        asm volatile("; TRM_PHASE " name);         \
        ::trm::phase_fence_values(__VA_ARGS__);    \
    } while (0)
#else
#define TRM_PHASE(name) ((void)0)
#define TRM_PHASE_FENCE(name, ...) ((void)0)
#endif

// Round-4 instruction cuts, individually switchable for same-box A/B builds (profiles/tools/r04_exp2.sh, r04_exp3.sh).  Measured
// (profiles/r04/exp3_instruction_cuts.log): together they take 27 ... 40 vector instructions per wave out of the step
// kernels and move the step time by 0 ... -2 %; the two that put a NEW wave-uniform branch into the hot path (FLUX, POWRARE) cost
// the HBM-resident step +4 % and are off; the others are on.
#ifndef TRM_DEEP_SCALAR_INPUTS       // k_column_deep: the per-column inputs through the scalar memory path (one column per wave)
#define TRM_DEEP_SCALAR_INPUTS 1
#endif
#ifndef TRM_LOAD_POINTERS_UPFRONT    // the base pointers of the field reads fetched in one batch in front of the first load
#define TRM_LOAD_POINTERS_UPFRONT 1
#endif
#ifndef TRM_EARLY_INPUTS   // the per-column inputs requested right behind the field loads, in front of the derivation (trm_column.hpp)
#define TRM_EARLY_INPUTS 1
#endif
#ifndef TRM_CUT_MASKS      // lane predicates / active masks from wave-uniform scalar masks instead of lane-wise compares
#define TRM_CUT_MASKS 1
#endif
#ifndef TRM_CUT_WT         // water table search on the scalar unit
#define TRM_CUT_WT 1
#endif
#ifndef TRM_CUT_NSZ        // divides by launch constants without the sign-of-zero select where it cannot matter
#define TRM_CUT_NSZ 1
#endif
#ifndef TRM_CUT_FLUX       // flux boundary terms: a select per condition that is set (branches around the terms: OFF, see above)
#define TRM_CUT_FLUX 0
#endif
#ifndef TRM_CUT_CHECK      // composition check of the closure reduced to what is not known by construction
#define TRM_CUT_CHECK 1
#endif
#ifndef TRM_CUT_POWRARE    // the non-finite case of the compensated power as a wave-uniform rare branch (OFF, see above)
#define TRM_CUT_POWRARE 0
#endif
#ifndef TRM_CUT_FRAC       // volumetric fractions shared between a closure and the tendencies that follow it
#define TRM_CUT_FRAC 1
#endif
#ifndef TRM_CUT_SOFF       // scalar loads with a register byte offset instead of a 64-bit address each
#define TRM_CUT_SOFF 1
#endif

namespace trm {

// ---------------------------------------------------------------------------
// Julia Base float semantics the reference relies on
// ---------------------------------------------------------------------------
template <class NF> TRM_HD bool is_nan(NF x) { return x != x; }
// the wave's predicate mask straight from the compare (s_and with exec); HIP's __ballot takes the predicate through a vector
// register (v_cndmask 0/1 + v_cmp_ne)
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD b mod 8).  xcd_block<true> makes every XCD work on
// ONE contiguous eighth of the columns instead of every eighth workgroup-sized chunk.  Measured twice, same build, two boxes
// (profiles/r03/exp16_xcd_remap.log, exp16b_xcd_remap_pk.log): the packed fp32 step at C5 454.2 -> 437.4 us on one box and
// 452.3 -> 483.6 us on the other; the fp64 column program +1.6 % at 8 x N145, +2 % at C3.  A lever whose sign depends on the
// box is not a default: both off (TRM_XCD_REMAP / TRM_XCD_REMAP_PK = 1 build the variants).
#ifndef TRM_XCD_REMAP
#define TRM_XCD_REMAP 0
#endif
#ifndef TRM_XCD_REMAP_PK
#define TRM_XCD_REMAP_PK 0
#endif
template <bool ON> __device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned nb) {
    if (!ON) return b;
    const unsigned per = nb / 8u, rem = nb % 8u, x = b % 8u, j = b / 8u;
    return x * per + (x < rem ? x : rem) + j;
}
template <class NF> TRM_HD bool sign_bit(NF x) { return __builtin_signbit(x); }
TRM_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
TRM_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
TRM_HD double copysign_(double a, double b) { return __builtin_copysign(a, b); }
TRM_HD float copysign_(float a, float b) { return __builtin_copysignf(a, b); }
TRM_HD bool is_finite(double x) { return __builtin_isfinite(x); }
TRM_HD bool is_finite(float x) { return __builtin_isfinite(x); }

// Base.min / Base.max(x::Float, y::Float) are signed-zero aware (-0 < +0) and NaN-propagating.
// The hardware v_min/v_max_f64 have the same zero ordering; they differ only when an operand is
// already NaN (IEEE minNum returns the other operand).  A NaN state is reported through the status
// word either way, so the one-instruction form is used on the device.
// (Written as the instruction itself when both operands are variables: the builtin would first re-quiet each
// operand the compiler cannot prove canonical -- values that arrive by load or DPP move -- with a v_max x, x.)
TRM_DEV double jl_min(double x, double y) {
    if (__builtin_constant_p(x) || __builtin_constant_p(y)) return __builtin_fmin(x, y);
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
TRM_DEV double jl_max(double x, double y) {
    if (__builtin_constant_p(x) || __builtin_constant_p(y)) return __builtin_fmax(x, y);
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
TRM_DEV float jl_min(float x, float y) {
    if (__builtin_constant_p(x) || __builtin_constant_p(y)) return __builtin_fminf(x, y);
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
TRM_DEV float jl_max(float x, float y) {
    if (__builtin_constant_p(x) || __builtin_constant_p(y)) return __builtin_fmaxf(x, y);
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

// a / b for a divisor b that is constant over the launch, given rb = RN(1/b) formed on the host by
// an IEEE division.  Markstein's theorem: with q = RN(a*rb) and the exact residual r = a - b*q (one
// fma), RN(q + r*rb) IS the correctly rounded quotient -- bit-identical to a / b for finite
// operands (verified against 3.3e8 random cases), in 3 dependent ops instead of the 11-deep divide.
template <class NF> TRM_HD NF div_const(NF a, NF b, NF rb) {
    NF q = a * rb;
    NF r = fma_(-q, b, a);
    NF q2 = fma_(r, rb, q);
    return (q == NF(0)) ? q : q2;  // keeps the sign of a zero quotient
}
// div_const without the select that keeps the sign of a ZERO quotient: identical to div_const for every a except a == -0.0,
// where it gives +0.0 (q = -0, r = fma(+0, b, -0) = +0, q2 = fma(+0, rb, -0) = +0).  Used where that sign provably cannot
// reach a result: the quotient is added to a value that is never -0.0 (x + (+-0) = x), or the numerator cannot be -0.0.  Each
// call site says which.  (An underflowing quotient, a != 0, reproduces itself: q2 = RN(a * rb) = q.)  Saves 3 of 6 instructions.
template <class NF> TRM_HD NF div_const_nsz(NF a, NF b, NF rb) {
#if !TRM_CUT_NSZ
    return div_const(a, b, rb);
#endif
    const NF q = a * rb;
    const NF r = fma_(-q, b, a);
    return fma_(r, rb, q);
}
// n / d for VARIABLE operands in fp64 without the scaling steps of the full IEEE sequence.  The compiler expands `/` to
// v_div_scale x2, v_rcp, two Newton steps on the reciprocal, q = n * y, the residual fma, v_div_fmas, v_div_fixup.
// v_div_scale leaves its operand unchanged (and v_div_fmas is a plain fma) unless the divisor is denormal or beyond
// 2^1021, the numerator is below 2^-969, or the quotient leaves the normal range; v_div_fixup supplies the zero /
// infinity / NaN cases from the original operands.  Every divide of the step has a divisor that is normal and bounded
// for any legal composition -- heat capacity C >= 1e3, theta_sat ~ porosity, -L_theta + eps in [4.9e-32, 2e8],
// r = theta / span in (0, 1] -- and numerators (energies, water contents, 1) far above 2^-969, so the sequence below
// IS the hardware divide with its no-op steps removed: bit-identical (tests/test_gpu_parity.py::test_divide_sequences).
// Outside those ranges (|numerator| < 2e-292, saturation below 4.5e-308) the last bit / the overflow case may differ.
TRM_HD double div_nr(double n, double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    double q = n * y;
    const double r = __builtin_fma(-d, q, n);
    q = __builtin_fma(r, y, q);
    return __builtin_amdgcn_div_fixup(q, d, n);
#else
    return n / d;
#endif
}
// fp32: the compiler's sequence (v_rcp_f32, one Newton step, q = n * y, two residual corrections) without its two
// v_div_scale steps; they are no-ops unless the divisor is denormal or beyond 2^126, the numerator is below 2^-103 or the
// quotient leaves the normal range (fp32 denormals are enabled on gfx950, so there is no mode switch around it).
TRM_HD float div_nr(float n, float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    float y = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    float q = n * y;
    float r = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(r, y, q);
    return __builtin_amdgcn_div_fixupf(q, d, n);
#else
    return n / d;
#endif
}
// Bool * Float: `false` is a strong zero carrying the sign of x.
template <class NF> TRM_HD NF boolmul(bool b, NF x) { return b ? x : copysign_(NF(0), x); }

template <class NF> struct Limits;
template <> struct Limits<double> {
    static TRM_HD double eps() { return 2.220446049250313e-16; }
    static TRM_HD double inf() { return __builtin_huge_val(); }
    static TRM_HD double nan() { return __builtin_nan(""); }
};
template <> struct Limits<float> {
    static TRM_HD float eps() { return 1.1920929e-07f; }
    static TRM_HD float inf() { return __builtin_huge_valf(); }
    static TRM_HD float nan() { return __builtin_nanf(""); }
};

// Upwinded face conductivity of the Darcy flux (soil_hydrology_rre.jl:120-125):
//     K* = (g < 0) * min(K[k-1], K[k]) + (g >= 0) * min(K[k], K[k+1])
// The Bool factors are strong zeros, so for any non-NaN gradient exactly one term survives and the other
// contributes a signed zero; q = -K* g, the flux difference and the `+ forcing` that follows absorb the
// sign of a zero, and a NaN gradient gives q = NaN either way.  The select form is therefore
// value-identical downstream (verified bit for bit against the oracle, which keeps the product form).
template <class NF> TRM_DEV NF upwind_conductivity(NF g, NF Kdn, NF Kmid, NF Kup) {
    const NF other = (g < NF(0)) ? Kdn : Kup;   // min is symmetric bit for bit: select the operand, then one min
    return jl_min(Kmid, other);
}

// src/utils/utils.jl:25
template <class NF> TRM_HD NF safediv(NF x, NF y) { return (y == NF(0)) ? Limits<NF>::inf() : div_nr(x, y + Limits<NF>::eps()); }

// Base.Math.pow_body(x, n::Integer): compensated power by squaring.
template <class NF> TRM_HD NF pow_int(NF x, int n) {
    if (n == 0) return NF(1);
    NF y = NF(1), xnlo = NF(0), ynlo = NF(0);
    if (n == 3) return x * x * x;
    if (n == 2) return x * x;   // pow_body(x, 2): the compensation term is the exact residual of x * x, so RN(hi + lo) = hi
    if (n < 0) {
        NF rx = NF(1) / x;
        if (n == -2) return rx * rx;
        if (is_finite(x)) xnlo = -fma_(x, rx, NF(-1)) * rx;
        x = rx;
        n = -n;
    }
    while (n > 1) {
        if (n & 1) {
            NF err = fma_(y, xnlo, x * ynlo);
            NF hi = x * y;
            NF lo = fma_(x, y, -hi);
            y = hi;
            ynlo = lo + err;
        }
        NF err = x * NF(2) * xnlo;
        NF hi = x * x;
        NF lo = fma_(x, x, -hi);
        x = hi;
        xnlo = lo + err;
        n >>= 1;
    }
    NF err = fma_(y, xnlo, x * ynlo);
    return (is_finite(x) && is_finite(err)) ? fma_(x, y, err) : x * y;
}

// An exponent that is constant over the launch: Base.:^(x, y) takes the integer
// path for integer-valued y (decided once on the host), the generic pow otherwise.
// Two further classes are recognised on the host: y = n/2 and y = n/3 with small odd / non-multiple n
// (van Genuchten with n = 2 gives 1/n = (n-1)/n = 1/2 and n/(n+1) = 2/3).  These are evaluated as
// sqrt(x)^n and cbrt(x)^n: a correctly rounded root followed by the compensated integer power, a few ulp
// from Base's pow -- inside the 1e-10 tolerance stated for the pow paths -- in ~30 instead of ~230 instructions.
enum { POW_GENERIC = 0, POW_INT = 1, POW_HALVES = 2, POW_THIRDS = 3 };
template <class NF> struct PowSpec {
    NF y;
    int n;
    int kind;
};
template <class NF> inline PowSpec<NF> make_pow_spec(NF y) {
    PowSpec<NF> s;
    s.y = y;
    s.n = 0;
    s.kind = POW_GENERIC;
    double yd = (double)y;
    if (yd > -4096.5 && yd < 24576.5) {
        long long yi = (long long)yd;
        if ((double)yi == yd && yi >= -4096 && yi <= 24576) {
            s.kind = POW_INT;
            s.n = (int)yi;
            return s;
        }
    }
    for (int q = 2; q <= 3; ++q) {
        double t = yd * q;
        long long n = (long long)(t < 0 ? t - 0.5 : t + 0.5);
        if (n >= -16 && n <= 16 && n % q != 0 && (NF)((NF)n / (NF)q) == y) {
            s.kind = q == 2 ? POW_HALVES : POW_THIRDS;
            s.n = (int)n;
            return s;
        }
    }
    return s;
}
// The generic pow is ~230 instructions and sits on paths that are rarely (or, for a launch, never) taken.  At IR level
// it is a single side-effect-free intrinsic call, cheap enough in the optimiser's eyes to be speculated -- hoisted out
// of its branch and evaluated by every wave (measured: 1097 instead of 608 VALU per wave in the van Genuchten step).
// The empty volatile asm pins it to its branch.
TRM_DEV void rare_path() { asm volatile(""); }
TRM_DEV double pow_generic(double x, double y) { rare_path(); return pow(x, y); }
TRM_DEV float pow_generic(float x, float y) { rare_path(); return powf(x, y); }
TRM_DEV double exp_(double x) { return exp(x); }
TRM_DEV float exp_(float x) { return expf(x); }
TRM_DEV double sqrt_(double x) { return sqrt(x); }
TRM_DEV float sqrt_(float x) { return sqrtf(x); }
TRM_DEV double fabs_(double x) { return fabs(x); }
TRM_DEV float fabs_(float x) { return fabsf(x); }

// 10^y for the ice impedance of partially frozen cells (soil_hydraulic_properties.jl:201-221), y in [-Omega, 0] and not
// integer-valued: exp(y ln 10) with ln 10 in two pieces and the product's rounding error carried along (fma residual), so
// the result is the library exp's (< 1 ulp) times (1 + O(eps^2 |y|)) -- within 2 ulp of Base's 10.0^y, against ~230
// instructions for the generic pow.  A wave pays for this path whenever ANY of its 64 cells sits on a freezing front:
// about half the waves of the N145 van Genuchten workloads (measured: 608 -> see DESIGN 4.1 VALU per wave).
TRM_DEV double exp10_frac(double y) {
    const double hi = 2.302585092994046, lo = -2.1707562233822494e-16;
    const double p = y * hi, e = fma_(y, hi, -p) + y * lo, r = exp_(p);
    return fma_(r, e, r);
}
TRM_DEV float exp10_frac(float y) {
    const float hi = 2.3025851f, lo = -3.1975437e-08f;
    const float p = y * hi, e = fma_(y, hi, -p) + y * lo, r = exp_(p);
    return fma_(r, e, r);
}

// pow_body(x, -5) with the loop unrolled: the same operations in the same order as pow_int(x, -5) for
// finite x (x * 0 and 1 * x folded; they are exact), i.e. bit-identical -- the BrooksCorey default
// lambda = 0.2 evaluates r^(-1/lambda) = r^(-5.0) for every cell and step.
template <class NF> TRM_HD NF pow_int_m5(NF x) {
    const NF rx = div_nr(NF(1), x);
    const NF l0 = -fma_(x, rx, NF(-1)) * rx;     // low part of 1/x
    // n = 5 (odd): y = rx, ynlo = 0 + l0
    const NF ynlo = NF(0) + (l0 + NF(0));
    NF err = rx * NF(2) * l0;
    const NF x2 = rx * rx;
    const NF l2 = fma_(rx, rx, -x2) + err;
    // n = 2 (even)
    err = x2 * NF(2) * l2;
    const NF x4 = x2 * x2;
    const NF l4 = fma_(x2, x2, -x4) + err;
    // n = 1: combine
    err = fma_(rx, l4, x4 * ynlo);
#if defined(__HIP_DEVICE_COMPILE__) && TRM_CUT_POWRARE
    // Base's `isfinite(x) && isfinite(err) ? muladd(x, y, err) : x * y`: an overflowing power (r below ~1e-62) is the rare case,
    // decided per wave from two ballots -- two compares and the fma instead of two compares, the fma, a product and a select
    const NF full = fma_(x4, rx, err);
    const unsigned long long all_finite = wave_ballot(is_finite(x4)) & wave_ballot(is_finite(err));
    if (all_finite == wave_ballot(true)) return full;
    rare_path();
#endif
    return (is_finite(x4) && is_finite(err)) ? fma_(x4, rx, err) : x4 * rx;
}

TRM_DEV double cbrt_(double x) { return cbrt(x); }
TRM_DEV float cbrt_(float x) { return cbrtf(x); }
template <class NF> TRM_DEV NF jl_pow(NF x, const PowSpec<NF>& s) {
    // (Base.:^ returns 1.0 for x === 1.0 up front; the integer path gives exactly 1 there anyway)
    if (s.kind == POW_INT) return (s.n == -5) ? pow_int_m5(x) : pow_int(x, s.n);   // the bit-exact path
    if (x == NF(1)) return NF(1);
    if (s.kind == POW_HALVES) {
        const NF r = sqrt_(x);                       // NaN for x < 0, as the domain error of Base's pow
        return s.n == 1 ? r : pow_int(r, s.n);
    }
    if (s.kind == POW_THIRDS) {
        const NF r = (x < NF(0)) ? Limits<NF>::nan() : cbrt_(x);
        return s.n == 1 ? r : pow_int(r, s.n);
    }
    return pow_generic(x, s.y);
}

// ---------------------------------------------------------------------------
// Device parameter block: trm_params converted to NF plus the sub-expressions
// that are constant over the grid, evaluated ON THE HOST IN NF with the
// reference's operation order (IEEE basic ops round identically everywhere).
// ---------------------------------------------------------------------------
// Hydraulics specialisation of the kernels (dead-code elimination + register pressure):
enum { HYD_BC_LINEAR = 0,   // BrooksCorey SWRC + UnsatKLinear with the default lambda = 0.2: the reference default
                            // (soil_hydraulic_properties.jl:132-140).  r^(-1/lambda) = r^(-5.0) takes Base's integer path: the
                            // compensated power by squaring, unrolled (pow_int_m5) -- no run-time PowSpec dispatch and no
                            // generic pow in the kernel (its ~40 literal constants were being kept in registers across the
                            // loops of the multi-step / multi-group programs); any other lambda takes HYD_GENERIC
       HYD_VG_N2 = 1,       // VanGenuchten SWRC + UnsatKVanGenuchten with n = 2: the variant of every reference test and
                            // example (exponents -1/m = -2, 1/n = (n-1)/n = 1/2, n/(n+1) = 2/3 fixed at compile time: no
                            // run-time PowSpec dispatch, a straight-line cell); any other n takes HYD_GENERIC
       HYD_GENERIC = 2 };   // any combination, decided at run time

template <class NF> struct DevParams {
    // Member order = order of use in the fused step kernel: the kernel arguments are fetched by scalar loads of up
    // to 64 contiguous bytes, so what the reference-default path needs comes first and packed.
    // composition (homogeneous_strat.jl:34-61, soil_volume.jl:52-67,103-107)
    NF por, rpor;
    // thermal (soil_thermal_properties.jl:90-107,119-123)
    NF sk_water, sk_ice, sk_air;      // sqrt(k_i)
    NF kterm_mineral, kterm_organic;  // sqrt(k_m)*mineral, sqrt(k_o)*organic
    NF K_sat;
    int flow, swrc, unsat_k, seb, halo_policy, impedance_int;
    NF c_water, c_ice, c_air;
    NF cterm_mineral, cterm_organic;  // c_m*mineral, c_o*organic
    NF L;                             // rho_w * Lsl (soil_energy_closures.jl:107)
    // hydrology
    NF theta_res, theta_span, rtheta_span, bc_psi_s, vwc_forcing;  // theta_span = por - theta_res
    PowSpec<NF> bc_neg_inv_lambda, bc_lambda;             // -1/lambda, lambda
    NF vg_alpha, impedance;
    NF I_ice_frozen;                                      // 10^(-impedance) by the integer path (impedance_int != 0)
    NF neg_inv_alpha;                                     // -1/alpha
    PowSpec<NF> vg_n, vg_neg_m, vg_neg_inv_m, vg_inv_n;   // n, -m, -1/m, 1/n
    PowSpec<NF> vgk_e1, vgk_e2;                           // n/(n+1), (n-1)/n
    NF org, solid_frac, frac_organic, frac_mineral;
    // surface energy balance
    NF albedo, emissivity, one_minus_emissivity, eps_sigma, kappa_s2, rkappa_s2, C_h, min_windspeed, tau_r, rtau_r, beta_evap;
    NF sigma, field_capacity;
    int prescribed_albedo, evap_resistance;
    NF Tref, eps_mw, one_minus_eps_mw, ca_rhoa, Llg_rhoa;
};

template <class NF> struct Frac { NF water, ice, air; };

// volumetric_fractions (soil_volume.jl:52-67); the SoilVolume constructor's
// @assert bounds (soil_volume.jl:26-28) become a status flag.
template <class NF> TRM_DEV Frac<NF> fractions(const DevParams<NF>& p, NF sat, NF liq, uint32_t& viol) {
    bool ok = (NF(0) <= sat && sat <= NF(1)) && (NF(0) <= liq && liq <= NF(1));
    viol |= ok ? 0u : 2u;
    Frac<NF> f;
    NF wi = sat * p.por;
    f.water = wi * liq;
    f.ice = wi * (NF(1) - liq);
    f.air = (NF(1) - sat) * p.por;
    return f;
}
// the same without the bounds check (the caller flags the composition itself: column programs, where most of the check is
// known to hold by construction)
template <class NF> TRM_DEV Frac<NF> fractions_unchecked(const DevParams<NF>& p, NF sat, NF liq) {
    Frac<NF> f;
    const NF wi = sat * p.por;
    f.water = wi * liq;
    f.ice = wi * (NF(1) - liq);
    f.air = (NF(1) - sat) * p.por;
    return f;
}
// InverseQuadratic bulk conductivity (soil_thermal_properties.jl:119-123): left fold over
// (water, ice, air, mineral, organic) of sqrt(k_i)*theta_i, squared.
template <class NF> TRM_DEV NF conductivity(const DevParams<NF>& p, const Frac<NF>& f) {
    NF s = p.sk_water * f.water;
    s = s + p.sk_ice * f.ice;
    s = s + p.sk_air * f.air;
    s = s + p.kterm_mineral;
    s = s + p.kterm_organic;
    return s * s;
}
// heat_capacity (soil_thermal_properties.jl:102-107)
template <class NF> TRM_DEV NF heat_capacity(const DevParams<NF>& p, const Frac<NF>& f) {
    NF s = p.c_water * f.water;
    s = s + p.c_ice * f.ice;
    s = s + p.c_air * f.air;
    s = s + p.cterm_mineral;
    s = s + p.cterm_organic;
    return s;
}

// van Genuchten-Mualem conductivity for states OUTSIDE 0 <= x <= 1: the reference evaluates the
// formula in complex arithmetic so illegal states give a finite magnitude
// (soil_hydraulic_properties.jl:217-218).  Cold path, kept out of line.
template <class NF> __device__ __noinline__ NF vg_conductivity_complex(NF x, NF kk, NF e1f, NF e2f) {
    double e1 = (double)e1f, e2 = (double)e2f;
    double r = fabs((double)x), th = ((double)x < 0.0) ? 3.141592653589793 : 0.0;
    double rp = pow(r, e1);
    double ar = rp * cos(e1 * th), ai = rp * sin(e1 * th);
    double ir = 1.0 - ar, ii = -ai;
    double r2 = pow(sqrt(ir * ir + ii * ii), e2), th2 = atan2(ii, ir);
    double br = r2 * cos(e2 * th2), bi = r2 * sin(e2 * th2);
    double tr = 1.0 - br, ti = -bi;
    double t2r = tr * tr - ti * ti, t2i = 2.0 * tr * ti;
    double sr = sqrt(r) * cos(0.5 * th), si = sqrt(r) * sin(0.5 * th);
    double zr = (double)kk * sr * t2r - (double)kk * si * t2i, zi = (double)kk * sr * t2i + (double)kk * si * t2r;
    return (NF)sqrt(zr * zr + zi * zi);
}

// hydraulic_conductivity at a cell centre (soil_hydraulic_properties.jl:170-181, 203-221)
template <class NF> TRM_DEV NF conductivity_linear(const DevParams<NF>& p, const Frac<NF>& f) {
    NF theta_sat = f.water + f.ice + f.air;
    return div_nr(p.K_sat * f.water, theta_sat);
}
// COMPLEX_FALLBACK = false (the fused step): a state outside 0 <= x <= 1 is an illegal composition -- the reference's
// CPU path stops at its SoilVolume @assert (soil_volume.jl:26-28) before K is ever evaluated, here it is reported through
// TRM_STATUS_COMPOSITION_OUT_OF_RANGE -- and K is NaN instead of the complex magnitude.  Keeping the out-of-line
// complex path callable from the step kernel costs 29 VGPRs (96 vs 67, 5 vs 7 waves per SIMD) and 8 % of its time.
// N2: the exponents of n = 2 as compile-time forms -- x^(2/3) = cbrt(x)^2 and y^(1/2) = sqrt(y), exactly what jl_pow
// evaluates for the PowSpecs {THIRDS, 2} and {HALVES, 1} (its x == 1 shortcut returns what the roots return there).
template <class NF, bool COMPLEX_FALLBACK = true, bool N2 = false> TRM_DEV NF conductivity_vg(const DevParams<NF>& p, NF liq, const Frac<NF>& f) {
    NF x = div_const(f.water, p.por, p.rpor);
    // I_ice = 10^(-Omega (1 - f)): Base.:^ takes the integer path when the exponent is integer-valued
    NF y = -p.impedance * (NF(1) - liq);
    NF I_ice;
    if (liq == NF(1)) {
        I_ice = NF(1);                      // y = -0.0: the integer path returns 1
    } else if (liq == NF(0) && p.impedance_int) {
        I_ice = p.I_ice_frozen;             // y = -Omega, integer-valued: pow_body(10, -Omega), formed on the host
    } else {
        NF yt = (NF)(int)y;
        if (yt == y && y > NF(-4096) && y < NF(4096)) I_ice = pow_int(NF(10), (int)y);
        else I_ice = exp10_frac(y);
    }
    if (x >= NF(0) && x <= NF(1)) {
        NF inner, t;
        if (N2) {
            const NF c = cbrt_(x);
            inner = NF(1) - c * c;
            t = NF(1) - sqrt_(inner);
        } else {
            inner = NF(1) - jl_pow(x, p.vgk_e1);
            t = NF(1) - jl_pow(inner, p.vgk_e2);
        }
        return fabs_(p.K_sat * I_ice * sqrt_(x) * (t * t));
    }
    if (!COMPLEX_FALLBACK) return Limits<NF>::nan();
    return vg_conductivity_complex(x, p.K_sat * I_ice, p.vgk_e1.y, p.vgk_e2.y);
}
template <class NF, int HYD, bool COMPLEX_FALLBACK = true> TRM_DEV NF conductivity_hydraulic(const DevParams<NF>& p, NF liq, const Frac<NF>& f) {
    if (HYD == HYD_BC_LINEAR) return conductivity_linear(p, f);
    if (HYD == HYD_VG_N2) return conductivity_vg<NF, COMPLEX_FALLBACK, true>(p, liq, f);
    return p.unsat_k == 0 ? conductivity_linear(p, f) : conductivity_vg<NF, COMPLEX_FALLBACK>(p, liq, f);
}

// Free-water energy closure (soil_energy_closures.jl:99-159): (U, sat) -> (liq, T)
template <class NF> TRM_DEV void energy_closure(const DevParams<NF>& p, NF U, NF sat, NF& liq, NF& T, uint32_t& viol) {
    NF Lth = p.L * sat * p.por;
    liq = (U >= NF(0)) ? NF(1) : boolmul(U >= -Lth, NF(1) - safediv(U, -Lth));
    NF C = heat_capacity(p, fractions(p, sat, liq, viol));
    // (U < -Lth) ? (U + Lth) / C : (U >= 0 ? U / C : 0): one divide, operands selected first
    NF num = (U < -Lth) ? (U + Lth) : U;
    NF quo = div_nr(num, C);
    T = (U < -Lth || U >= NF(0)) ? quo : NF(0);
}
// inverse (initialisation only, soil_energy_closures.jl:64-97): (T, sat) -> (liq, U)
template <class NF> TRM_DEV void energy_invclosure(const DevParams<NF>& p, NF T, NF sat, NF& liq, NF& U, uint32_t& viol) {
    liq = (T >= NF(0)) ? NF(1) : NF(0);
    NF C = heat_capacity(p, fractions(p, sat, liq, viol));
    U = T * C - p.L * sat * p.por * (NF(1) - liq);
}

// FreezeCurves.jl 0.9 SWRCs (restated): psi_m(theta; theta_sat = por)
// NSZ: the caller guarantees theta is not -0.0 (a saturation that has been through the repair is +0.0 at least: `s + 0`, the
// bottom clamp max(s, +0)), so the numerator theta - theta_res cannot be -0.0 and the quotient's zero needs no sign fix
template <class NF, bool M5 = false, bool NSZ = false> TRM_DEV NF swrc_psi_bc(const DevParams<NF>& p, NF theta) {
    NF r = NSZ ? div_const_nsz(theta - p.theta_res, p.theta_span, p.rtheta_span) : div_const(theta - p.theta_res, p.theta_span, p.rtheta_span);
    NF v = -p.bc_psi_s * (M5 ? pow_int_m5(r) : jl_pow(r, p.bc_neg_inv_lambda));
    return (theta < p.por) ? v : -p.bc_psi_s;
}
template <class NF, bool N2 = false, bool NSZ = false> TRM_DEV NF swrc_psi_vg(const DevParams<NF>& p, NF theta) {
    if (theta < p.por) {
        NF r = NSZ ? div_const_nsz(theta - p.theta_res, p.theta_span, p.rtheta_span) : div_const(theta - p.theta_res, p.theta_span, p.rtheta_span);
        if (N2) {   // r^(-2) = (1 / r)^2 (pow_int, n = -2), then the square root
            const NF rr = div_nr(NF(1), r);
            return p.neg_inv_alpha * sqrt_(rr * rr - NF(1));
        }
        return p.neg_inv_alpha * jl_pow(jl_pow(r, p.vg_neg_inv_m) - NF(1), p.vg_inv_n);
    }
    return NF(0);
}
template <class NF, int HYD, bool NSZ = false> TRM_DEV NF swrc_psi(const DevParams<NF>& p, NF theta) {
    if (HYD == HYD_BC_LINEAR) return swrc_psi_bc<NF, true, NSZ>(p, theta);
    if (HYD == HYD_VG_N2) return swrc_psi_vg<NF, true, NSZ>(p, theta);
    return p.swrc == 1 ? swrc_psi_vg<NF, false, NSZ>(p, theta) : swrc_psi_bc<NF, false, NSZ>(p, theta);
}
template <class NF> TRM_DEV NF swrc_theta(const DevParams<NF>& p, NF psi, NF theta_sat) {
    if (p.swrc == 1) {
        if (psi <= NF(0))
            return p.theta_res + (theta_sat - p.theta_res) * jl_pow(NF(1) + jl_pow(-p.vg_alpha * psi, p.vg_n), p.vg_neg_m);
        return theta_sat;
    }
    if (psi < -p.bc_psi_s) return p.theta_res + (theta_sat - p.theta_res) * jl_pow(-p.bc_psi_s / psi, p.bc_lambda);
    return theta_sat;
}
// saturation_to_pressure! (soil_hydraulic_closures.jl:102-129): psi = (psi_h + psi_m) + psi_z,
// psi_z = z - z_ref is a per-level constant formed on the host.
template <class NF, int HYD, bool SAT_REPAIRED = false> TRM_DEV NF pressure_head(const DevParams<NF>& p, NF sat, NF z, NF psiz, NF z0) {
    NF psim = swrc_psi<NF, HYD, SAT_REPAIRED>(p, sat * p.por);
    NF psih = jl_max(NF(0), z0 - z);
    return psih + psim + psiz;
}

// ---- surface energy balance ---------------------------
template <class NF> TRM_DEV NF saturation_vapor_pressure(NF T) {  // physics_utils.jl:54,67-73
    // (div_nr: every divisor of the surface processes is normal and far from the range limits -- temperatures in degC
    // offset by ~250, pressures ~1e5, resistances in [1, 1e6] -- so the sequence IS the IEEE division, see div_nr)
    // (the branch of the reference selects the Magnus coefficients; one exp and one divide whatever the signs in the wave)
    const bool ice = T <= NF(0);
    const NF a = ice ? NF(22.46) : NF(17.62), b = ice ? NF(272.62) : NF(243.12);
    return NF(611.0) * exp_(div_nr(a * T, T + b));
}
template <class NF> TRM_DEV NF humidity_vpd(const DevParams<NF>& p, NF pres, NF q_air, NF Ts) {
    // physical_constants.jl:83-97 compute_vpd; physics_utils.jl:38
    NF e_sat = saturation_vapor_pressure(Ts);
    NF e_air = div_nr(q_air * pres, p.eps_mw + p.one_minus_eps_mw * q_air);
    NF vpd = jl_max(e_sat - e_air, NF(0.1));
    return div_nr(p.eps_mw * vpd, pres);
}
template <class NF> TRM_DEV NF aerodynamic_resistance(const DevParams<NF>& p, NF windspeed) {
    // prescribed_atmosphere.jl:110-116,137
    NF V = jl_max(windspeed, p.min_windspeed);
    NF Va = jl_max(V, NF(1.0e-6));
    return div_nr(NF(1), p.C_h * Va);
}

// (albedo, eps_sigma = emissivity * sigma, one_minus_emissivity: ConstantAlbedo's launch constants or PrescribedAlbedo's
// per-column inputs, see seb_radiation_inputs)
template <class NF> struct SebIn { NF Tair, pres, wind, qair, rain, swd, lwd, albedo, eps_sigma, one_minus_emissivity; };
template <class NF> struct SebOut { NF Ts, ghf, swu, lwu, rnet, Hs, Hl, evap, infil, runoff; };

// albedo / emissivity of a column (albedo.jl:37-44, abstract_types.jl:120-131): stefan_boltzmann is eps * sigma * T^4
template <class NF> TRM_DEV void seb_radiation_inputs(const DevParams<NF>& p, const NF* albedo, const NF* emissivity, unsigned byte_off, SebIn<NF>& in) {
    if (p.prescribed_albedo) {
        const NF a = *reinterpret_cast<const NF*>(reinterpret_cast<const char*>(albedo) + byte_off);
        const NF e = *reinterpret_cast<const NF*>(reinterpret_cast<const char*>(emissivity) + byte_off);
        in.albedo = a;
        in.eps_sigma = e * p.sigma;
        in.one_minus_emissivity = NF(1) - e;
    } else {
        in.albedo = p.albedo;
        in.eps_sigma = p.eps_sigma;
        in.one_minus_emissivity = p.one_minus_emissivity;
    }
}
// surface_energy_balance.jl:119-144 with the ET-coupled latent heat flux (turbulent_fluxes.jl:130-143); Q_h is the
// surface_humidity_flux of the evapotranspiration scheme
template <class NF> TRM_DEV void seb_fluxes_humidity(const DevParams<NF>& p, const SebIn<NF>& in, NF ra, NF Q_h, SebOut<NF>& o) {
    o.swu = in.albedo * in.swd;
    NF Tk = o.Ts + p.Tref;
    o.lwu = in.eps_sigma * pow_int(Tk, 4) + in.one_minus_emissivity * in.lwd;
    o.rnet = o.swu - in.swd + o.lwu - in.lwd;
    NF Q_T = div_nr(o.Ts - in.Tair, ra);
    o.Hs = p.ca_rhoa * Q_T;
    o.Hl = p.Llg_rhoa * Q_h;
    o.ghf = o.rnet - o.Hs - o.Hl;
}
template <class NF> TRM_DEV void seb_fluxes(const DevParams<NF>& p, const SebIn<NF>& in, NF ra, SebOut<NF>& o) {
    seb_fluxes_humidity(p, in, ra, o.evap, o);   // bare ground: the ground evaporation alone (bare_ground_evaporation.jl:29)
}
// compute_auxiliary! of the surface processes for one column (land_model.jl:79-88):
// bare-ground evaporation, direct runoff / infiltration, then the fused SEB kernel twice.
// 0-D per column; it runs in its own small launch (k_surface) so that its registers (exp, divides)
// do not inflate the per-cell kernels.
TRM_DEV double cos_(double x) { return cos(x); }
TRM_DEV float cos_(float x) { return cosf(x); }
// ground_evaporation_resistance_factor (ground_resistance_factor.jl:12,36-56): the constant factor, or Lee & Pielke's
// soil-moisture limitation from the liquid water content of the top cell against the field capacity
template <class NF> TRM_DEV NF evaporation_resistance_factor(const DevParams<NF>& p, NF sat_top, NF liq_top) {
    if (p.evap_resistance == 0) return p.beta_evap;
    const NF water = (sat_top * p.por) * liq_top;          // volumetric_fractions(soil).water
    if (water < p.field_capacity) {
        const NF t = NF(1) - cos_(NF(3.141592653589793) * water / p.field_capacity);
        return (t * t) / NF(4);
    }
    return NF(1);
}
// direct_surface_runoff.jl:87-117: infiltration and runoff of the rain that reaches the ground
template <class NF> TRM_DEV void surface_runoff(const DevParams<NF>& p, NF rain, NF sat_top, NF Kf_top, NF S, bool richards, SebOut<NF>& o) {
    NF excess = richards ? S : NF(0);
    bool unsat = sat_top < NF(1);
    NF drainage;
    if (excess > NF(0)) {
        drainage = div_const(jl_max(excess, NF(0)), p.tau_r, p.rtau_r);
        o.infil = boolmul(unsat, jl_min(drainage, Kf_top));
    } else {
        drainage = NF(0);
        o.infil = boolmul(unsat, jl_min(rain, Kf_top));
    }
    o.runoff = rain + drainage - o.infil;
}
template <class NF>
TRM_DEV void surface_processes(const DevParams<NF>& p, const SebIn<NF>& in, NF Ts_in, NF T_ground, NF sat_top, NF liq_top,
                                               NF Kf_top, NF S, bool richards, NF dz_top, SebOut<NF>& o) {
    o.Ts = Ts_in;
    NF ra = aerodynamic_resistance(p, in.wind);
    // bare_ground_evaporation.jl:49-62
    o.evap = div_nr(evaporation_resistance_factor(p, sat_top, liq_top) * humidity_vpd(p, in.pres, in.qair, o.Ts), ra);
    // direct_surface_runoff.jl:87-117
    NF excess = richards ? S : NF(0);
    bool unsat = sat_top < NF(1);
    NF drainage;
    if (excess > NF(0)) {
        drainage = div_const(jl_max(excess, NF(0)), p.tau_r, p.rtau_r);
        o.infil = boolmul(unsat, jl_min(drainage, Kf_top));
    } else {
        drainage = NF(0);
        o.infil = boolmul(unsat, jl_min(in.rain, Kf_top));
    }
    o.runoff = in.rain + drainage - o.infil;
    // surface_energy_balance.jl:95-110, executed twice (land_model.jl:85-86)
    for (int sweep = 0; sweep < 2; ++sweep) {
        seb_fluxes(p, in, ra, o);
        o.Ts = T_ground - div_const(o.ghf * dz_top, p.kappa_s2, p.rkappa_s2);  // skin_temperature.jl:62-68
        seb_fluxes(p, in, ra, o);
    }
}

// ---- z halos (Oceananigans fill_halo_regions!) ----------
struct BcSet {
    int kind[5][2];          // [bc_var][side]
    const void* value[5][2]; // per-column arrays (NF), may be null for NOFLUX
};
// Boundary-face geometry, constant over the launch (formed on the host in NF)
template <class NF> struct BcGeom {
    NF dzf_bot, dzf_top;             // face spacing at the two boundary faces
    NF hdzf_bot, hdzf_top;           // dzf / 2
    NF rhdzf_bot, rhdzf_top;         // 1 / (dzf / 2)
    NF Az, V_bot, V_top, rV_bot, rV_top;  // Az = dx (Flat y), V = Az * dz of the boundary cell
    NF zF_top, dzc_top;              // surface elevation zF[Nz], thickness of the top cell
};
// halo value above the top cell / below the bottom cell of a centre field
template <class NF> TRM_DEV NF halo_top(int kind, const NF* val, long i, NF c_edge, const BcGeom<NF>& g) {
    if (kind == 1) {  // Value: linear extrapolation through the boundary value
        NF grad = div_const(val[i] - c_edge, g.hdzf_top, g.rhdzf_top);
        return c_edge + grad * g.dzf_top;
    }
    if (kind == 3) return c_edge + val[i] * g.dzf_top;  // Gradient
    return c_edge;                                       // Flux / NoFlux / default
}
template <class NF> TRM_DEV NF halo_bottom(int kind, const NF* val, long i, NF c_edge, const BcGeom<NF>& g) {
    if (kind == 1) {
        NF grad = div_const(c_edge - val[i], g.hdzf_bot, g.rhdzf_bot);
        return c_edge + grad * (-g.dzf_bot);
    }
    if (kind == 3) return c_edge + val[i] * (-g.dzf_bot);
    return c_edge;
}
// compute_z_bcs!: flux * Az / V of a boundary cell
template <class NF> TRM_DEV NF flux_term_top(NF flux, const BcGeom<NF>& g) { return div_const(flux * g.Az, g.V_top, g.rV_top); }
template <class NF> TRM_DEV NF flux_term_bottom(NF flux, const BcGeom<NF>& g) { return div_const(flux * g.Az, g.V_bot, g.rV_bot); }
// (for terms that are ADDED to a tendency that is never -0.0: see div_const_nsz)
template <class NF> TRM_DEV NF flux_term_top_nsz(NF flux, const BcGeom<NF>& g) { return div_const_nsz(flux * g.Az, g.V_top, g.rV_top); }
template <class NF> TRM_DEV NF flux_term_bottom_nsz(NF flux, const BcGeom<NF>& g) { return div_const_nsz(flux * g.Az, g.V_bot, g.rV_bot); }

}  // namespace trm
