// sector_probe.hip -- does skipping 32-byte sectors / 64-byte rows of a streamed block save HBM time on gfx950?
// 1024 waves, each streams `T` blocks of 4 KiB (4 problems x 16 rows x 64 B, the C stream of the LQR kernel) through
// global_load_lds, 8 stages in flight.  mode 0: every granule; 1: rows 8..15 skip their first 32 B (quarters 0, 1 alias
// quarters 2, 3: 25 % fewer distinct bytes, in 32-byte sectors); 2: rows 8..15 alias row 8 (44 % fewer bytes, whole
// 64-byte rows); 3: upper block triangle (quarter q >= row / 4: 37.5 % fewer bytes, 16-byte granules)
//   hipcc -O3 --offload-arch=gfx950 -o sector_probe tools/ubench/sector_probe.hip && ./sector_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
__shared__ __attribute__((aligned(16))) char buf[8 * 4096];
__global__ void __launch_bounds__(64, 1) k(const char *C, long tstride, int T, int mode, float *out)
{
    const int l = threadIdx.x;
    const char *src[4];
    for (int kk = 0; kk < 4; ++kk) {
        int g = l;                       // granule of problem kk: row g / 4, quarter g & 3
        const int row = g >> 2, q = g & 3;
        if (mode == 1 && row >= 8 && q < 2) g += 2;
        if (mode == 2 && row >= 8) g = 32 + q;
        if (mode == 3 && q < (row >> 2)) g = (row << 2) | 3;
        src[kk] = C + ((long)blockIdx.x * 4 + kk) * 1024 + 16 * g;
    }
    float acc = 0.f;
    for (int t = 0; t < T + 7; ++t) {
        if (t < T)
            for (int kk = 0; kk < 4; ++kk)
                __builtin_amdgcn_global_load_lds((glb_void_t *)(src[kk] + (long)t * tstride), (lds_void_t *)(buf + (t & 7) * 4096 + kk * 1024), 16, 0, 0);
        if (t >= 7) {
            asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
            acc += *(const float *)(buf + ((t - 7) & 7) * 4096 + 4 * l);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 64 + l] = acc;
}
int main()
{
    const int T = 50, W = 1024;
    const long tstride = (long)W * 4096;
    char *C; float *out;
    hipMalloc(&C, (size_t)T * tstride); hipMalloc(&out, W * 64 * 4);
    hipMemset(C, 0, (size_t)T * tstride);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 4; ++mode) {
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(W), dim3(64), 0, 0, C, tstride, T, mode, out);
        hipEventRecord(a);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k, dim3(W), dim3(64), 0, 0, C, tstride, T, mode, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("mode %d: %.2f us per launch, %.2f TB/s of the full %.0f MB\n", mode, ms * 1e3 / 200, (double)T * tstride / (ms * 1e-3 / 200) / 1e12, (double)T * tstride / 1e6);
    }
    return 0;
}
