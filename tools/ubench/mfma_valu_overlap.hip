// mfma_valu_overlap.hip -- does INDEPENDENT vector work run underneath v_mfma_f32_16x16x4_f32?
//  (a) inside one wave: a trip of 8 independent MFMAs with N independent v_fma between them (one wave per SIMD);
//  (b) across waves: two waves per SIMD, one issuing only MFMAs, the other only v_fma, against each of them alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV> __global__ void __launch_bounds__(64, 1) inwave(float *out, long long *cyc, int rep)
{
    float a[8], b[8], f[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = 1.f + 1e-3f * i; f[i] = 0.5f + i; }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const float m = 1.0001f, c = 1e-4f;
    long long t0 = clock64();
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k], acc[k], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) f[(k + j) & 7] = __builtin_fmaf(f[(k + j) & 7], m, c);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + f[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// role 0: MFMA only, role 1: VALU only, role 2: waves 0..3 MFMA + waves 4..7 VALU (two per SIMD)
__global__ void __launch_bounds__(512, 1) crosswave(float *out, long long *cyc, int rep, int role)
{
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = role == 0 || (role == 2 && wave < 4), do_valu = role == 1 || (role == 2 && wave >= 4);
    float a[8], b[8], f[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = 1.f + 1e-3f * i; f[i] = 0.5f + i; }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const float m = 1.0001f, c = 1e-4f;
    long long t0 = clock64();
    if (do_mfma) {
        for (int r = 0; r < rep; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k], acc[k], 0, 0, 0);
    }
    if (do_valu) {
        for (int r = 0; r < rep; ++r)
#pragma unroll
            for (int k = 0; k < 64; ++k) f[k & 7] = __builtin_fmaf(f[k & 7], m, c);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + f[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 512 * 4 * 4);
    hipMalloc(&cyc, 1024 * 8 * 8);
    const int rep = 2000;
    auto mean = [&](int n, int stride, int first, int count) {
        std::vector<long long> h(n);
        hipMemcpy(h.data(), cyc, n * 8, hipMemcpyDeviceToHost);
        double s = 0; int c = 0;
        for (int i = 0; i < n; ++i) if ((i % stride) >= first && (i % stride) < first + count) { s += h[i]; ++c; }
        return s / c;
    };
#define RUN(NV)                                                                                            \
    for (int pass = 0; pass < 2; ++pass) hipLaunchKernelGGL(inwave<NV>, dim3(1024), dim3(64), 0, 0, out, cyc, rep); \
    hipDeviceSynchronize();                                                                                \
    printf("one wave per SIMD: 8 MFMAs + 8 x %2d independent v_fma per trip: %7.1f clk per trip (%5.1f per MFMA)\n", NV, \
           mean(1024, 1, 0, 1) / rep, mean(1024, 1, 0, 1) / rep / 8);
    RUN(0) RUN(1) RUN(2) RUN(4) RUN(6) RUN(8) RUN(12)
    for (int role = 0; role < 3; ++role) {
        for (int pass = 0; pass < 2; ++pass) hipLaunchKernelGGL(crosswave, dim3(256), dim3(512), 0, 0, out, cyc, rep, role);
        hipDeviceSynchronize();
        if (role == 0) printf("8 waves per CU, all MFMA (two per SIMD):        %7.1f clk per trip of 8 MFMAs per wave\n", mean(2048, 1, 0, 1) / rep);
        if (role == 1) printf("8 waves per CU, all v_fma (two per SIMD):       %7.1f clk per trip of 64 v_fma per wave\n", mean(2048, 1, 0, 1) / rep);
        if (role == 2) printf("4 MFMA waves + 4 v_fma waves per CU:            %7.1f clk per MFMA trip, %7.1f clk per v_fma trip\n",
                              mean(2048, 8, 0, 4) / rep, mean(2048, 8, 4, 4) / rep);
    }
    return 0;
}
