// mfma16_turn.hip -- v_mfma_f32_16x16x4_f32 issued back to back against the same stream with one VALU instruction
// (a select feeding the next A operand) between every two: what an MFMA <-> VALU turn costs one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND> __global__ void __launch_bounds__(64, 1) rate(float *out, long long *cyc, int rep)
{
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = 1.f + 1e-3f * i; }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const bool sel = threadIdx.x < 40;
    long long t0 = clock64();
    for (int r = 0; r < rep; ++r) {
        if (KIND == 0) {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[m], acc[m & 3], 0, 0, 0);
        } else if (KIND == 1) {       // operands prepared first, then the block
            float s[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) s[m] = sel ? a[m] + (float)r : 0.f;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(s[m], b[m], acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {                      // select, MFMA, select, MFMA ...
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const float s = sel ? a[m] + (float)r : 0.f;
                __builtin_amdgcn_sched_barrier(0);
                acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(s, b[m], acc[m & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 1024 * 64 * 4);
    hipMalloc(&cyc, 1024 * 8);
    const int grid = 1024, rep = 2000;
    for (int kind = 0; kind < 3; ++kind) {
        for (int pass = 0; pass < 2; ++pass) {
            if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(grid), dim3(64), 0, 0, out, cyc, rep);
            else if (kind == 1) hipLaunchKernelGGL(rate<1>, dim3(grid), dim3(64), 0, 0, out, cyc, rep);
            else hipLaunchKernelGGL(rate<2>, dim3(grid), dim3(64), 0, 0, out, cyc, rep);
        }
        hipDeviceSynchronize();
        std::vector<long long> h(grid);
        hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += v; mean /= grid;
        printf("%-52s %7.2f clk per MFMA (8 per trip%s)\n", kind == 0 ? "16x16x4 f32 back to back" : kind == 1 ? "8 selects (2 VALU each), then 8 MFMAs" : "select, MFMA, select, MFMA, ...",
               mean / (8.0 * rep), kind ? ", 16 VALU" : "");
    }
    return 0;
}
