// mfma4_probe.hip -- layout and broadcast semantics of v_mfma_f32_4x4x1_16b_f32 on gfx950, and its issue cost
// for one wave per SIMD.  Prints, for cbsz=2 and each abid, which (lane, value) pairs feed D.
//   hipcc -O3 --offload-arch=gfx950 -o mfma4_probe tools/ubench/mfma4_probe.hip && ./mfma4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ABID> __global__ void probe(float *out)
{
    const int l = threadIdx.x;
    const float a = 1000.f + l;          // A operand: identifies the source lane
    const float b = 1.f;                 // B = 1: D[v] = A_src(v)
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 2, ABID, 0);
    for (int v = 0; v < 4; ++v) out[(ABID * 64 + l) * 4 + v] = d[v];
}
__global__ void probe_b(float *out)
{
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(1.f, 2000.f + l, c, 2, 0, 0);
    for (int v = 0; v < 4; ++v) out[l * 4 + v] = d[v];
}
// rounding: is D = fma(a, b, c) with one rounding?
__global__ void probe_fma(float *out)
{
    const float a = 1.f + 0x1p-12f, b = 1.f + 0x1p-12f;   // a*b = 1 + 2^-11 + 2^-24
    f32x4 c = {-1.f, -1.f, -1.f, -1.f};
    f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    out[threadIdx.x] = d[0];
    out[64 + threadIdx.x] = fmaf(a, b, -1.f);
    out[128 + threadIdx.x] = a * b - 1.f;
}
template <int KIND> __global__ void __launch_bounds__(64, 1) rate(float *out, long long *cyc, int rep)
{
    float a[12], b[12];
    for (int i = 0; i < 12; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = 1.f + 1e-3f * i; }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float x0 = 0, x1 = 0, x2 = 0, x3 = 0;
    int sc = 0;
    long long t0 = clock64();
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            if (KIND == 0) {       // four independent accumulators, as Q = C + F'Y would issue them
                acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m], b[m], acc[0], 2, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m], b[m], acc[1], 2, 1, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m], b[m], acc[2], 2, 2, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m], b[m], acc[3], 2, 3, 0);
            } else if (KIND == 2) {   // an independent v_fmac after every MFMA: does it fill the second pass?
                acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m], b[m], acc[0], 2, 0, 0);
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x0) : "v"(b[m]), "v"(b[0]));
                acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m], b[m], acc[1], 2, 1, 0);
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x1) : "v"(b[m]), "v"(b[1]));
                acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m], b[m], acc[2], 2, 2, 0);
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x2) : "v"(b[m]), "v"(b[2]));
                acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m], b[m], acc[3], 2, 3, 0);
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x3) : "v"(b[m]), "v"(b[3]));
            } else if (KIND == 3) {   // a scalar instruction after every MFMA
                acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m], b[m], acc[0], 2, 0, 0);
                asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
                acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m], b[m], acc[1], 2, 1, 0);
                asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
                acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m], b[m], acc[2], 2, 2, 0);
                asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
                acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m], b[m], acc[3], 2, 3, 0);
                asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
            } else {               // one accumulator: the dependent chain
                acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m], b[m], acc[0], 2, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m], b[m], acc[0], 2, 1, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m], b[m], acc[0], 2, 2, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[m], b[m], acc[0], 2, 3, 0);
            }
        }
    }
    long long t1 = clock64();
    float s = x0 + x1 + x2 + x3 + sc;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 4096 * 64 * 4);
    hipMalloc(&cyc, 4096 * 8);
    std::vector<float> h(4 * 64 * 4);
    hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, out);
    hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, out);
    hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, out);
    hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, out);
    hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int ab = 0; ab < 4; ++ab)
        for (int l = 0; l < 64; ++l)
            for (int v = 0; v < 4; ++v) {
                const float want = 1000.f + (l & ~15) + 4 * ab + v;      // lane (row base + 4 abid + v)
                if (h[(ab * 64 + l) * 4 + v] != want) ++bad;
            }
    printf("A broadcast (cbsz=2): D[v] of lane l = A of lane (l & ~15) + 4*abid + v : %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    if (bad) for (int l = 0; l < 64; l += 5) printf("  abid=1 lane %2d: %g %g %g %g\n", l, h[(64 + l) * 4], h[(64 + l) * 4 + 1], h[(64 + l) * 4 + 2], h[(64 + l) * 4 + 3]);
    hipLaunchKernelGGL(probe_b, dim3(1), dim3(64), 0, 0, out);
    hipMemcpy(h.data(), out, 64 * 4 * 4, hipMemcpyDeviceToHost);
    bad = 0;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) if (h[l * 4 + v] != 2000.f + l) ++bad;
    printf("B stays in its lane: D[v] of lane l = B of lane l : %s\n", bad ? "NO" : "yes");
    hipLaunchKernelGGL(probe_fma, dim3(1), dim3(64), 0, 0, out);
    hipMemcpy(h.data(), out, 192 * 4, hipMemcpyDeviceToHost);
    printf("rounding: mfma %.10g  fmaf %.10g  mul-then-add %.10g\n", h[0], h[64], h[128]);
    for (int grid : {1024}) {
        for (int kind = 0; kind < 4; ++kind) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int rep = 2000;
            auto go = [&]() {
                if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(grid), dim3(64), 0, 0, out, cyc, rep);
                else if (kind == 1) hipLaunchKernelGGL(rate<1>, dim3(grid), dim3(64), 0, 0, out, cyc, rep);
                else if (kind == 2) hipLaunchKernelGGL(rate<2>, dim3(grid), dim3(64), 0, 0, out, cyc, rep);
                else hipLaunchKernelGGL(rate<3>, dim3(grid), dim3(64), 0, 0, out, cyc, rep);
            };
            go();
            hipEventRecord(e0);
            go();
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> hc(grid);
            hipMemcpy(hc.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (auto v : hc) mean += v; mean /= grid;
            printf("v_mfma_f32_4x4x1 %s grid %4d: %6.2f clk/instr, %6.3f ns/instr\n", kind == 0 ? "four accumulators" : kind == 1 ? "one accumulator  " : kind == 2 ? "four acc + v_fmac each (per MFMA)" : "four acc + s_add each (per MFMA)", grid, mean / (48.0 * rep), ms * 1e6 / (48.0 * rep));
        }
    }
    return 0;
}
