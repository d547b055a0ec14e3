// Issue cost of the instructions the filter kernels are made of, one wavefront per SIMD (1024 single-wave workgroups),
// eight independent chains per lane so that dependent-issue latency does not enter: ns per wave-instruction.
// Also the DEPENDENT cost (one chain) of fma, for the latency of a dependent chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHAINS8(OP) \
    a0 = OP(a0); a1 = OP(a1); a2 = OP(a2); a3 = OP(a3); a4 = OP(a4); a5 = OP(a5); a6 = OP(a6); a7 = OP(a7);

__device__ __forceinline__ double op_fma(double x) { return __builtin_fma(x, 0.999999, 1e-7); }
__device__ __forceinline__ double op_mul(double x) { return x * 0.999999; }
__device__ __forceinline__ double op_add(double x) { return x + 1e-7; }
__device__ __forceinline__ double op_rsq(double x) { return __builtin_amdgcn_rsq(x) + 1.0; } /* + add: measured separately */
__device__ __forceinline__ double op_rcp(double x) { return __builtin_amdgcn_rcp(x) + 1.0; }
__device__ __forceinline__ double op_rsq32(double x) { return (double)__builtin_amdgcn_rsqf((float)x) + 1.0; }
__device__ __forceinline__ double op_sqrt(double x) { return __builtin_amdgcn_sqrt(x) + 1.0; }
__device__ __forceinline__ double op_cvt(double x) { float f = (float)x; asm volatile("" : "+v"(f)); return (double)f; }

template <int WHICH>
__global__ __launch_bounds__(64) void k(double *out, int iters) {
    const int lane = threadIdx.x;
    double a0 = lane * 1e-3 + 1.0, a1 = a0 + 0.1, a2 = a0 + 0.2, a3 = a0 + 0.3, a4 = a0 + 0.4, a5 = a0 + 0.5,
           a6 = a0 + 0.6, a7 = a0 + 0.7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int rep = 0; rep < 8; ++rep) {
        if (WHICH == 0) { CHAINS8(op_fma) }
        if (WHICH == 1) { CHAINS8(op_mul) }
        if (WHICH == 2) { CHAINS8(op_add) }
        if (WHICH == 3) { CHAINS8(op_rsq) }
        if (WHICH == 4) { CHAINS8(op_rcp) }
        if (WHICH == 5) { CHAINS8(op_rsq32) }
        if (WHICH == 6) { CHAINS8(op_sqrt) }
        if (WHICH == 7) { CHAINS8(op_cvt) }
        if (WHICH == 8) { a0 = op_fma(a0); a0 = op_fma(a0); a0 = op_fma(a0); a0 = op_fma(a0); a0 = op_fma(a0); a0 = op_fma(a0); a0 = op_fma(a0); a0 = op_fma(a0); }
        if (WHICH == 9) { /* v_cndmask pairs: select between two doubles */
            const bool c = ((i + rep) & 1) != 0;
            a0 = c ? a0 : a1; a1 = c ? a1 : a2; a2 = c ? a2 : a3; a3 = c ? a3 : a4; a4 = c ? a4 : a5; a5 = c ? a5 : a6; a6 = c ? a6 : a7; a7 = c ? a7 : a0;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
      }
    }
    out[(size_t)blockIdx.x * 64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int W>
static float run(double *d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<W>, dim3(1024), dim3(64), 0, 0, d, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<W>, dim3(1024), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    double *d;
    hipMalloc(&d, 1024 * 64 * sizeof(double));
    const int iters = 20000;
    const double per = 1e6 / (64.0 * iters);
    const float fma = run<0>(d, iters), mul = run<1>(d, iters), add = run<2>(d, iters);
    printf("{\"v_fma_f64\": %.2f, \"v_mul_f64\": %.2f, \"v_add_f64\": %.2f", fma * per, mul * per, add * per);
    printf(", \"v_rsq_f64 (+ v_add_f64)\": %.2f", run<3>(d, iters) * per);
    printf(", \"v_rcp_f64 (+ v_add_f64)\": %.2f", run<4>(d, iters) * per);
    printf(", \"cvt f64->f32, v_rsq_f32, cvt f32->f64 (+ v_add_f64)\": %.2f", run<5>(d, iters) * per);
    printf(", \"v_sqrt_f64 (+ v_add_f64)\": %.2f", run<6>(d, iters) * per);
    printf(", \"cvt f64->f32->f64 (2 instr)\": %.2f", run<7>(d, iters) * per);
    printf(", \"v_fma_f64 dependent chain\": %.2f", run<8>(d, iters) * per);
    printf(", \"2 x v_cndmask_b32 (one double select)\": %.2f", run<9>(d, iters) * per);
    printf(", \"unit\": \"ns per wave-instruction group, one wavefront per SIMD\"}\n");
    return 0;
}
