// launchfloor.hip -- what ONE dependent kernel launch per step costs on this device, whatever the kernel does: the period of
// back-to-back launches on one stream (HIP events around N launches) for
//   (a) an empty kernel, by grid size (1 ... 7 119 workgroups of 256 threads: the shard of BASELINE config 4 has 890 + 28,
//       C2 1 753, N145 7 119) and by the size of its by-value arguments (8 bytes ... 1.5 KB: the step kernels pass ~330 scalars),
//   (b) a kernel whose waves read one kernel argument from the END of the argument block (the scalar load a step kernel starts with),
//   (c) a kernel in which every thread does one dependent global load -> store (the shortest real memory round trip).
// The per-step LandModel shard (8.4 us), C2 (6.0 us) and the 1 780-column grid (7.1 us) are to be read against these floors.
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/launchfloor profiles/tools/microbench/launchfloor.hip && /tmp/launchfloor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int WORDS> struct Args { double w[WORDS]; };

template <int WORDS> __global__ void __launch_bounds__(256) k_empty(Args<WORDS> a, double* out) {
    if (a.w[0] == 12345.678 && out) out[0] = a.w[WORDS - 1];      // (never true: keeps the arguments alive)
}
template <int WORDS> __global__ void __launch_bounds__(256) k_last_arg(Args<WORDS> a, double* out) {
    const double x = a.w[WORDS - 1];
    if (x == 12345.678 && out) out[threadIdx.x] = x;
}
// every wave sums ALL of its arguments (scalar loads of the whole block, as a step kernel's s_load of its ~330 scalars): by value ...
// (integer sums of the 32-bit halves: scalar loads and scalar adds only -- the cost is the argument fetch, not arithmetic)
template <int WORDS> __global__ void __launch_bounds__(256) k_sum_args(Args<WORDS> a, double* out) {
    unsigned x = 0;
    _Pragma("unroll") for (int i = 0; i < WORDS; ++i) { const unsigned long long b = __builtin_bit_cast(unsigned long long, a.w[i]); x += (unsigned)b ^ (unsigned)(b >> 32); }
    if (x == 12345u && out) out[threadIdx.x] = x;
}
// ... and from a device-resident block written ONCE (what does not change between two steps: field pointers, parameters)
template <int WORDS> __global__ void __launch_bounds__(256) k_sum_resident(const Args<WORDS>* __restrict__ a, double* out) {
    const Args<WORDS>* c = (const Args<WORDS>*)__builtin_assume_aligned(a, 16);
    unsigned x = 0;
    _Pragma("unroll") for (int i = 0; i < WORDS; ++i) { const unsigned long long b = __builtin_bit_cast(unsigned long long, c->w[i]); x += (unsigned)b ^ (unsigned)(b >> 32); }
    if (x == 12345u && out) out[threadIdx.x] = x;
}
__global__ void k_spin(long long cycles, double* out) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (out && cycles < 0) out[0] = 1.0;
}
__global__ void __launch_bounds__(256) k_roundtrip(const double* in, double* out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1.0;
}

template <class F> static double period_us(F launch, hipStream_t s, int n = 2000, int reps = 7) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> t;
    for (int r = 0; r < reps; ++r) {
        for (int i = 0; i < 200; ++i) launch();
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < n; ++i) launch();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms * 1e3 / n);
    }
    std::sort(t.begin(), t.end());
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return t[t.size() / 2];
}

// GPU-side period: the N launches are enqueued while the stream is held up by a spinning kernel (100 MHz wall clock: `ms` of it),
// so what the events bracket is the device draining a queue that is already full -- no host enqueue cost in it
template <class F> static double drain_us(F launch, hipStream_t s, int n = 1500, int reps = 5, double ms = 12.0) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> t;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, (long long)(ms * 1e5), (double*)nullptr);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < n; ++i) launch();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float el = 0;
        CK(hipEventElapsedTime(&el, e0, e1));
        t.push_back(el * 1e3 / n);
    }
    std::sort(t.begin(), t.end());
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return t[t.size() / 2];
}
template <int WORDS> static void drained(hipStream_t s, double* out, int grid) {
    Args<WORDS> a{};
    Args<WORDS>* d;
    CK(hipMalloc(&d, sizeof(a)));
    CK(hipMemcpy(d, &a, sizeof(a), hipMemcpyHostToDevice));
    const double e = drain_us([&] { hipLaunchKernelGGL(k_empty<WORDS>, dim3(grid), dim3(256), 0, s, a, out); }, s);
    const double v = drain_us([&] { hipLaunchKernelGGL(k_sum_args<WORDS>, dim3(grid), dim3(256), 0, s, a, out); }, s);
    const double r = drain_us([&] { hipLaunchKernelGGL(k_sum_resident<WORDS>, dim3(grid), dim3(256), 0, s, d, out); }, s);
    std::printf("{\"queue\": \"full\", \"arg_bytes\": %d, \"workgroups\": %d, \"empty_us\": %.3f, \"sums_by_value_args_us\": %.3f, \"sums_resident_block_us\": %.3f}\n", WORDS * 8, grid, e, v, r);
    CK(hipFree(d));
}

template <int WORDS> static void by_args(hipStream_t s, double* out, int grid) {
    Args<WORDS> a{};
    const double e = period_us([&] { hipLaunchKernelGGL(k_empty<WORDS>, dim3(grid), dim3(256), 0, s, a, out); }, s);
    const double l = period_us([&] { hipLaunchKernelGGL(k_last_arg<WORDS>, dim3(grid), dim3(256), 0, s, a, out); }, s);
    std::printf("{\"kernel\": \"empty\", \"arg_bytes\": %d, \"workgroups\": %d, \"us_per_launch\": %.3f, \"reads_last_argument_us\": %.3f}\n", WORDS * 8 + 8, grid, e, l);
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    double *in, *out;
    const long nmax = 7119L * 256;
    CK(hipMalloc(&in, nmax * sizeof(double)));
    CK(hipMalloc(&out, nmax * sizeof(double)));
    CK(hipMemset(in, 0, nmax * sizeof(double)));
    for (int grid : {1, 918, 1753, 7119}) {
        drained<2>(s, out, grid);
        drained<64>(s, out, grid);
        drained<192>(s, out, grid);
    }
    for (int grid : {1, 256, 918, 1753, 7119}) {
        by_args<1>(s, out, grid);
        by_args<64>(s, out, grid);
        by_args<192>(s, out, grid);
    }
    for (int grid : {1, 256, 918, 1753, 7119}) {
        const long n = (long)grid * 256;
        const double t = period_us([&] { hipLaunchKernelGGL(k_roundtrip, dim3(grid), dim3(256), 0, s, in, out, n); }, s);
        std::printf("{\"kernel\": \"load+store per thread\", \"workgroups\": %d, \"us_per_launch\": %.3f}\n", grid, t);
    }
    return 0;
}
