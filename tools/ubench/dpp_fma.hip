// Micro-benchmark: issue rate of v_fmac_f32 with a DPP row_newbcast operand vs a plain v_fmac_f32,
// one wave per SIMD (the regime of a 4-problems-per-wave LQR kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#include <utility>
#define ROW_NEWBCAST(n) (0x150 + (n))
template <int N> __device__ __forceinline__ float bcast(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ROW_NEWBCAST(N), 0xf, 0xf, true));
}
template <int MODE, int... I>
__device__ __forceinline__ void row(float (&a)[16], float vm, float f, std::integer_sequence<int, I...>)
{
    ((a[I] = fmaf(MODE == 0 ? vm : bcast<I>(vm), f, a[I])), ...);
}

template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, int iters)
{
    float a[16], v[12];
    const int lane = threadIdx.x;
    for (int i = 0; i < 16; ++i) a[i] = 0.f;
    for (int i = 0; i < 12; ++i) v[i] = 0.001f * (lane + i + 1);
    float f = 1.0f + 1e-3f * lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            row<MODE>(a, v[m], f, std::make_integer_sequence<int, 16>{});
        }
        f += 1e-6f;
#pragma unroll
        for (int m = 0; m < 12; ++m) v[m] += 1e-7f * a[m];
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * 64 + lane] = s;
}

int main()
{
    float *d;
    (void)hipMalloc(&d, 4096 * 64 * 4);
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
        const int grid = 1024 * waves_per_simd, iters = 2000;
        for (int mode = 0; mode < 2; ++mode) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, d, iters);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, d, iters);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
            }
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double)iters * 192;
            printf("waves/SIMD %d mode %s: %.3f ms, %.2f ns per FMA instr per wave, (%.2f cycles @2.0GHz)\n", waves_per_simd,
                   mode ? "dpp_newbcast" : "plain", ms, ms * 1e6 / instr, ms * 1e6 / instr * 2.0);
        }
    }
    return 0;
}
