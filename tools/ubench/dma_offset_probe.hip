// dma_offset_probe.hip -- does the immediate offset of global_load_lds_dwordx4 move the LDS destination as well as
// the global source?  (If it does, one M0 write serves every DMA of a stage.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void glb_void_t;
__shared__ __attribute__((aligned(16))) char lds[16384];

template <int OFF> __global__ void probe(const float *g, float *out)
{
    for (int i = threadIdx.x; i < 4096; i += 64) ((float *)lds)[i] = -1.f;
    __syncthreads();
    // lane l moves 16 bytes: source = g + 2048 floats + 4 l (+ OFF bytes if the offset applies to the source)
    const char *src = (const char *)(g + 2048) + 16 * threadIdx.x;
    __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)(lds + 8192), 16, OFF, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = ((float *)lds)[i];
}
int main()
{
    std::vector<float> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = i;
    float *g, *out;
    hipMalloc(&g, 8192 * 4); hipMalloc(&out, 4096 * 4);
    hipMemcpy(g, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    std::vector<float> r(4096);
    for (int off : {0, 1024, -2048}) {
        if (off == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, g, out);
        else if (off == 1024) hipLaunchKernelGGL(probe<1024>, dim3(1), dim3(64), 0, 0, g, out);
        else hipLaunchKernelGGL(probe<-2048>, dim3(1), dim3(64), 0, 0, g, out);
        hipMemcpy(r.data(), out, 4096 * 4, hipMemcpyDeviceToHost);
        int first = -1, n = 0;
        for (int i = 0; i < 4096; ++i) if (r[i] != -1.f) { if (first < 0) first = i; ++n; }
        printf("offset %5d: %d floats landed, first at LDS byte %d (base 8192), value %g (source float index; 2048 = no source shift)\n",
               off, n, first * 4, first >= 0 ? r[first] : -1.f);
    }
    return 0;
}
