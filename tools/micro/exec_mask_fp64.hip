// Does a wave64 fp64 VALU instruction get cheaper when whole 16-lane quarters of EXEC are zero?
// One wave per SIMD (1024 single-wave workgroups, large LDS request keeps it at 1 per SIMD is not needed: 1024 blocks
// on 1024 SIMDs). Lanes selected by `mode` run a long fp64 FMA chain, the others skip it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(64) void chain(double *out, int iters, unsigned long long mask) {
    const int lane = threadIdx.x;
    double a0 = lane * 1e-3 + 1.0, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3, a4 = a0 + 0.4, a5 = a0 + 0.5,
           a6 = a0 + 0.6, a7 = a0 + 0.7;
    const double m = 0.999999, c = 1e-7;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
            a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
            a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
        }
    }
    out[(size_t)blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    double *d;
    hipMalloc(&d, 1024 * 64 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct { const char *name; unsigned long long mask; } modes[] = {
        {"all 64 lanes", ~0ull},
        {"lanes 0-31", 0xffffffffull},
        {"lanes 0-15", 0xffffull},
        {"lanes 0-7", 0xffull},
        {"lane 0", 1ull},
        {"lanes 16-31", 0xffff0000ull},
        {"every 4th lane (16 lanes, all quarters)", 0x1111111111111111ull},
        {"lanes 0-15 + lane 63", 0x800000000000ffffull},
    };
    const int iters = 200000;
    for (auto &md : modes) {
        hipLaunchKernelGGL(chain, dim3(1024), dim3(64), 0, 0, d, 1000, md.mask);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(chain, dim3(1024), dim3(64), 0, 0, d, iters, md.mask);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-45s %8.3f ms  -> %.2f ns per wave-instruction\n", md.name, ms, ms * 1e6 / (8.0 * iters));
    }
    return 0;
}
