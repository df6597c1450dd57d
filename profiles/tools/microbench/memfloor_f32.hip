// Memory-only floors of the packed fp32 LandModel step (C5: 812 500 columns x 64 levels; 4 field reads U, sat, T, psi and 6 field
// writes per cell, no arithmetic), by how a wave's accesses are shaped:
//   P  the shipped k_step_pk mapping: lane = level, the lane's two columns as TWO 4-byte accesses per field (256 contiguous bytes each)
//   Q  lane = two consecutive levels of one column: ONE 8-byte access per field, a wave covers two columns (512 contiguous bytes)
//   R  lane = four consecutive levels: one 16-byte access per field, a wave covers four columns (1 KB)
//   Ps / Qs  as P / Q with the fields' bases skewed by 65 x 256 B against each other (the library's layout)
// Build: hipcc --offload-arch=gfx950 -O3 -o memfloor_f32 memfloor_f32.hip ; run: ./memfloor_f32 [columns]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Ptrs { float* f[6]; };
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_p(Ptrs p, int Nh, float dt) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int i0 = wave * 2;
    if (i0 >= Nh) return;
    const int i1 = i0 + 1 < Nh ? i0 + 1 : i0;
    const size_t c0 = (size_t)i0 * 64 + lane, c1 = (size_t)i1 * 64 + lane;
    float a[4][2];
#pragma unroll
    for (int f = 0; f < 4; ++f) { a[f][0] = p.f[f][c0]; a[f][1] = p.f[f][c1]; }
#pragma unroll
    for (int f = 0; f < 4; ++f) { p.f[f][c0] = a[f][0] + dt; p.f[f][c1] = a[f][1] + dt; }
    p.f[4][c0] = a[0][0] * dt; p.f[4][c1] = a[0][1] * dt;
    p.f[5][c0] = a[1][0] * dt; p.f[5][c1] = a[1][1] * dt;
}
__global__ void __launch_bounds__(256) k_q(Ptrs p, int Nh, float dt) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int i = wave * 2 + (lane >> 5);
    if (i >= Nh) return;
    const size_t c = (size_t)i * 32 + (lane & 31);
    f2 a[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) a[f] = ((const f2*)p.f[f])[c];
#pragma unroll
    for (int f = 0; f < 4; ++f) ((f2*)p.f[f])[c] = a[f] + dt;
    ((f2*)p.f[4])[c] = a[0] * dt;
    ((f2*)p.f[5])[c] = a[1] * dt;
}
__global__ void __launch_bounds__(256) k_r(Ptrs p, int Nh, float dt) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int i = wave * 4 + (lane >> 4);
    if (i >= Nh) return;
    const size_t c = (size_t)i * 16 + (lane & 15);
    f4 a[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) a[f] = ((const f4*)p.f[f])[c];
#pragma unroll
    for (int f = 0; f < 4; ++f) ((f4*)p.f[f])[c] = a[f] + dt;
    ((f4*)p.f[4])[c] = a[0] * dt;
    ((f4*)p.f[5])[c] = a[1] * dt;
}

int main(int argc, char** argv) {
    const int Nh = argc > 1 ? atoi(argv[1]) : 812500;
    const size_t n = (size_t)Nh * 64;
    Ptrs p, ps;
    const size_t skew = 65 * 256;
    for (int f = 0; f < 6; ++f) {
        char* base;
        CK(hipMalloc(&base, n * 4 + 64 * skew));
        CK(hipMemset(base, 0, n * 4 + 64 * skew));
        p.f[f] = (float*)base;
        ps.f[f] = (float*)(base + f * skew);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 10.0 * n * 4;
    auto run = [&](const char* name, auto launch) {
        for (int w = 0; w < 200; ++w) launch();   // clocks settle
        CK(hipDeviceSynchronize());
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            for (int s = 0; s < 30; ++s) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("%-44s Nh %7d  %7.2f us/launch (best of 5 x 30)  %6.2f TB/s of %.1f MB\n", name, Nh, best * 1000 / 30, bytes / (best * 1e-3 / 30) / 1e12, bytes / 1e6);
    };
    const int w2 = (Nh + 1) / 2, w4 = (Nh + 3) / 4;
    run("P  lane = level, 2 x 4 B per field (shipped)", [&] { hipLaunchKernelGGL(k_p, dim3((w2 + 3) / 4), dim3(256), 0, 0, p, Nh, 1.0f); });
    run("Ps the same, skewed field bases", [&] { hipLaunchKernelGGL(k_p, dim3((w2 + 3) / 4), dim3(256), 0, 0, ps, Nh, 1.0f); });
    run("Q  lane = 2 levels, 8 B per field", [&] { hipLaunchKernelGGL(k_q, dim3((w2 + 3) / 4), dim3(256), 0, 0, p, Nh, 1.0f); });
    run("Qs the same, skewed field bases", [&] { hipLaunchKernelGGL(k_q, dim3((w2 + 3) / 4), dim3(256), 0, 0, ps, Nh, 1.0f); });
    run("R  lane = 4 levels, 16 B per field", [&] { hipLaunchKernelGGL(k_r, dim3((w4 + 3) / 4), dim3(256), 0, 0, p, Nh, 1.0f); });
    run("Rs the same, skewed field bases", [&] { hipLaunchKernelGGL(k_r, dim3((w4 + 3) / 4), dim3(256), 0, 0, ps, Nh, 1.0f); });
    return 0;
}
