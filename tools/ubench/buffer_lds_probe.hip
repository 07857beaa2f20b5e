#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
__global__ void probe(const float *src, int nbytes, const unsigned *voff, float *out)
{
    __shared__ __attribute__((aligned(16))) float lds[256];
    const int l = threadIdx.x;
    for (int i = l; i < 256; i += 64) lds[i] = -7.0f;           // sentinel: "untouched"
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, nbytes, 0x00020000);
    // 64 lanes x 4 B -> lds[0..63]; lanes with voff beyond nbytes are out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)lds, 4, voff[l], 0, 0, 0);
    // a second instruction with an immediate offset of 256 B: where does it land (source and/or destination moved)?
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(lds + 64), 4, voff[l], 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = l; i < 256; i += 64) out[i] = lds[i];
}
int main()
{
    const int N = 64;
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = 100.f + i;
    std::vector<unsigned> off(64);
    for (int l = 0; l < 64; ++l) off[l] = (l % 3 == 2) ? 0x7fffff00u : 4u * ((l * 7) % 40);    // every third lane out of range
    float *d, *o; unsigned *dv;
    hipMalloc(&d, N * 4); hipMalloc(&o, 256 * 4); hipMalloc(&dv, 64 * 4);
    hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(dv, off.data(), 64 * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, 40 * 4, dv, o);
    std::vector<float> r(256);
    hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        const float want = (l % 3 == 2) ? 0.f : 100.f + (l * 7) % 40;
        printf("lane %2d off %10u -> %8.1f / %8.1f (in range: %g)\n", l, off[l], r[l], r[64 + l], want);
    }
    return 0;
}
