// Memory-only floors of the fused step's access pattern (5 field reads, 6 field writes per cell, fp64):
//   A  lane = level, 8 B per lane per access (the shipped k_step_wave mapping), one column pair per wave
//   B  4 consecutive levels per lane (2 x 16 B per lane per field), 8 columns per wave
//   C  as A with G column-pair groups per wave, the next group's loads issued before the current group's stores
// Build: hipcc --offload-arch=gfx950 -O3 -o memfloor memfloor.hip ; run: ./memfloor <columns>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Ptrs { double* f[6]; };

__global__ void __launch_bounds__(256) k_a(Ptrs p, int Nh, double dt) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int i = wave * 2 + (lane >> 5), k = lane & 31;
    if (i >= Nh) return;
    const size_t c = (size_t)i * 32 + k;
    double a0 = p.f[0][c], a1 = p.f[1][c], a2 = p.f[2][c], a3 = p.f[3][c], a4 = p.f[4][c];
    p.f[0][c] = a0 + dt; p.f[1][c] = a1 + dt; p.f[2][c] = a2 + dt; p.f[3][c] = a3 + dt; p.f[4][c] = a4 + dt; p.f[5][c] = a0 + a1;
}
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_b(Ptrs p, int Nh, double dt) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int i = wave * 8 + (lane >> 3), k4 = lane & 7;
    if (i >= Nh) return;
    const size_t c = ((size_t)i * 32 + k4 * 4) / 2;
    d2 a[5][2];
#pragma unroll
    for (int f = 0; f < 5; ++f) { a[f][0] = ((const d2*)p.f[f])[c]; a[f][1] = ((const d2*)p.f[f])[c + 1]; }
#pragma unroll
    for (int f = 0; f < 5; ++f) { ((d2*)p.f[f])[c] = a[f][0] + dt; ((d2*)p.f[f])[c + 1] = a[f][1] + dt; }
    ((d2*)p.f[5])[c] = a[0][0] + a[1][0]; ((d2*)p.f[5])[c + 1] = a[0][1] + a[1][1];
}
template <int G> __global__ void __launch_bounds__(256) k_c(Ptrs p, int Nh, double dt) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int k = lane & 31;
    double a[G][5];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        int i = (wave * G + g) * 2 + (lane >> 5);
        i = i < Nh ? i : Nh - 1;
        const size_t c = (size_t)i * 32 + k;
#pragma unroll
        for (int f = 0; f < 5; ++f) a[g][f] = p.f[f][c];
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int i = (wave * G + g) * 2 + (lane >> 5);
        if (i >= Nh) continue;
        const size_t c = (size_t)i * 32 + k;
#pragma unroll
        for (int f = 0; f < 5; ++f) p.f[f][c] = a[g][f] + dt;
        p.f[5][c] = a[g][0] + a[g][1];
    }
}

// D: as A with only U, sat, psi read (T and liq derived in registers): 3 reads + 6 writes
__global__ void __launch_bounds__(256) k_d(Ptrs p, int Nh, double dt) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int i = wave * 2 + (lane >> 5), k = lane & 31;
    if (i >= Nh) return;
    const size_t c = (size_t)i * 32 + k;
    double a0 = p.f[0][c], a1 = p.f[1][c], a4 = p.f[4][c];
    p.f[0][c] = a0 + dt; p.f[1][c] = a1 + dt; p.f[2][c] = a0 * dt; p.f[3][c] = a1 * dt; p.f[4][c] = a4 + dt; p.f[5][c] = a0 + a1;
}
// E: 2 reads + 6 writes (psi derived as well)
__global__ void __launch_bounds__(256) k_e(Ptrs p, int Nh, double dt) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int i = wave * 2 + (lane >> 5), k = lane & 31;
    if (i >= Nh) return;
    const size_t c = (size_t)i * 32 + k;
    double a0 = p.f[0][c], a1 = p.f[1][c];
    p.f[0][c] = a0 + dt; p.f[1][c] = a1 + dt; p.f[2][c] = a0 * dt; p.f[3][c] = a1 * dt; p.f[4][c] = a0 - a1; p.f[5][c] = a0 + a1;
}

// F: as D (3 reads + 6 writes) on a TILED layout: the six fields of a workgroup's 8 columns adjacent in memory (8 x 32 x 8 B = 2 KB per
// field, 12 KB per tile) -- a workgroup's nine streams touch one 12 KB run instead of nine places ~115 MB apart
__global__ void __launch_bounds__(256) k_f(double* base, int Nh, double dt) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int i = wave * 2 + (lane >> 5), k = lane & 31;
    if (i >= Nh) return;
    const size_t tile = (size_t)(i >> 3) * (6 * 256), in_tile = (size_t)(i & 7) * 32 + k;
    double* f0 = base + tile + in_tile;
    double a0 = f0[0], a1 = f0[256], a4 = f0[4 * 256];
    f0[0] = a0 + dt; f0[256] = a1 + dt; f0[2 * 256] = a0 * dt; f0[3 * 256] = a1 * dt; f0[4 * 256] = a4 + dt; f0[5 * 256] = a0 + a1;
}

int main(int argc, char** argv) {
    const int Nh = argc > 1 ? atoi(argv[1]) : 56951;
    const size_t n = (size_t)Nh * 32;
    Ptrs p;
    for (int f = 0; f < 6; ++f) { CK(hipMalloc(&p.f[f], n * 8 + 64)); CK(hipMemset(p.f[f], 0, n * 8)); }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 11.0 * n * 8;
    auto run = [&](const char* name, auto launch) {
        for (int w = 0; w < 2000; ++w) launch();   // clocks settle
        CK(hipDeviceSynchronize());
        float best = 1e9, sum = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            for (int s = 0; s < 200; ++s) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; sum += ms;
        }
        printf("%-28s Nh %7d  %7.2f us/launch (best of 5 x 200)  %6.2f TB/s of %.1f MB\n", name, Nh, best * 1000 / 200, bytes / (best * 1e-3 / 200) / 1e12, bytes / 1e6);
    };
    const int wavesA = (Nh + 1) / 2, wavesB = (Nh + 7) / 8;
    run("A lane=level 8B/lane", [&] { hipLaunchKernelGGL(k_a, dim3((wavesA + 3) / 4), dim3(256), 0, 0, p, Nh, 1.0); });
    run("D 3 reads 6 writes (x9/11)", [&] { hipLaunchKernelGGL(k_d, dim3((wavesA + 3) / 4), dim3(256), 0, 0, p, Nh, 1.0); });
    run("E 2 reads 6 writes (x8/11)", [&] { hipLaunchKernelGGL(k_e, dim3((wavesA + 3) / 4), dim3(256), 0, 0, p, Nh, 1.0); });
    double* tiled; CK(hipMalloc(&tiled, 6 * n * 8 + (1 << 20))); CK(hipMemset(tiled, 0, 6 * n * 8));
    run("F 3 reads 6 writes, tiled layout (x9/11)", [&] { hipLaunchKernelGGL(k_f, dim3((wavesA + 3) / 4), dim3(256), 0, 0, tiled, Nh, 1.0); });
    run("B 4 levels/lane 2x16B", [&] { hipLaunchKernelGGL(k_b, dim3((wavesB + 3) / 4), dim3(256), 0, 0, p, Nh, 1.0); });
    run("C2 two groups per wave", [&] { hipLaunchKernelGGL(k_c<2>, dim3(((wavesA + 1) / 2 + 3) / 4), dim3(256), 0, 0, p, Nh, 1.0); });
    run("C4 four groups per wave", [&] { hipLaunchKernelGGL(k_c<4>, dim3(((wavesA + 3) / 4 + 3) / 4), dim3(256), 0, 0, p, Nh, 1.0); });
    return 0;
}
