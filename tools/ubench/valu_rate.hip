// valu_rate.hip -- issue cost (shader clocks per wave64 instruction) of the VALU forms the LQR kernels choose between.
// One wave per SIMD (grid = 1024 x 64), each test a straight block of 12 x REP independent-accumulator instructions.
//   hipcc -O3 --offload-arch=gfx950 -o valu_rate tools/ubench/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define DPPM " row_mask:0xf bank_mask:0xf\n"
#define R12(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__shared__ __attribute__((aligned(16))) float lds_buf[4096];

template <int KIND> __global__ void __launch_bounds__(64, 1) k(float *out, long long *cyc, int rep)
{
    float a[12], s[12];
    f32x2 p[12];
    for (int i = 0; i < 12; ++i) { a[i] = threadIdx.x * 0.001f + i; s[i] = 1.0f + i * 1e-3f; p[i] = f32x2{a[i], s[i]}; }
    float m = 1.0001f;
    f32x2 m2 = {1.0001f, 0.9999f};
    for (int i = threadIdx.x; i < 4096; i += 64) lds_buf[i] = i;
    __syncthreads();
    unsigned addr = (threadIdx.x >> 4) * 1024;      // one address per 16-lane row: a broadcast read
    f32x4 acc4 = {0, 0, 0, 0};
    long long t0 = clock64();
    for (int r = 0; r < rep; ++r) {
        if (KIND == 0) {
#define X(i) "v_fmac_f32 %" #i ", %12, %13\n"
            asm volatile(R12(X) R12(X) R12(X) R12(X)
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11])
                : "v"(s[0]), "v"(m));
#undef X
        } else if (KIND == 1) {
#define X(i) "v_fmac_f32_dpp %" #i ", %12, %13 row_newbcast:3" DPPM
            asm volatile("s_nop 1\n" R12(X) R12(X) R12(X) R12(X)
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11])
                : "v"(s[0]), "v"(m));
#undef X
        } else if (KIND == 2) {
#define X(i) "v_pk_fma_f32 %" #i ", %12, %13, %" #i "\n"
            asm volatile(R12(X) R12(X) R12(X) R12(X)
                : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(p[8]), "+v"(p[9]), "+v"(p[10]), "+v"(p[11])
                : "v"(m2), "v"(m2));
#undef X
        } else if (KIND == 3) {
#define X(i) "v_mov_b32_dpp %" #i ", %12 row_newbcast:3" DPPM
            asm volatile("s_nop 1\n" R12(X) R12(X) R12(X) R12(X)
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11])
                : "v"(s[0]));
#undef X
        } else if (KIND == 4) {
#define X(i) "v_fmac_f32_dpp %" #i ", %12, %13 quad_perm:[1,0,3,2]" DPPM
            asm volatile("s_nop 1\n" R12(X) R12(X) R12(X) R12(X)
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11])
                : "v"(s[0]), "v"(m));
#undef X
        } else if (KIND == 5) {
#define X(i) "v_fmac_f32_dpp %" #i ", %12, %13 row_shr:1" DPPM
            asm volatile("s_nop 1\n" R12(X) R12(X) R12(X) R12(X)
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11])
                : "v"(s[0]), "v"(m));
#undef X
        } else if (KIND == 6) {
            // 48 broadcast ds_read_b128 (every 16-lane row reads one address), consumed by 48 adds
#pragma unroll
            for (int q = 0; q < 48; ++q) {
                f32x4 v = *(const f32x4 *)((const char *)lds_buf + addr + 16 * (q & 31));
                acc4 += v;
            }
        } else if (KIND == 7) {
            // 48 x ds_read_b128, every lane its own 16 bytes (conflict-free streaming)
#pragma unroll
            for (int q = 0; q < 48; ++q) {
                f32x4 v = *(const f32x4 *)((const char *)lds_buf + threadIdx.x * 16 + 1024 * (q & 7));
                acc4 += v;
            }
        } else if (KIND == 9) {
            // one accumulator: every instruction waits for the previous one
#define X(i) "v_fmac_f32 %0, %12, %13\n"
            asm volatile(R12(X) R12(X) R12(X) R12(X)
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11])
                : "v"(s[0]), "v"(m));
#undef X
        } else if (KIND == 10) {
            // two accumulators alternating
#define X(i) "v_fmac_f32 %0, %12, %13\nv_fmac_f32 %1, %12, %13\n"
            asm volatile(R12(X) R12(X)
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11])
                : "v"(s[0]), "v"(m));
#undef X
        } else if (KIND == 11) {
            // dependent chain through v_rcp_f32
#define X(i) "v_rcp_f32 %0, %0\n"
            asm volatile(R12(X) R12(X) R12(X) R12(X)
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11])
                : "v"(s[0]), "v"(m));
#undef X
        } else if (KIND == 12) {
            // s_nop 0 stream
#define X(i) "s_nop 0\n"
            asm volatile(R12(X) R12(X) R12(X) R12(X) ::: "memory");
#undef X
        } else if (KIND == 13) {
            // salu stream
#define X(i) "s_add_u32 %0, %0, 1\n"
            int sreg = r;
            asm volatile(R12(X) R12(X) R12(X) R12(X) : "+s"(sreg));
            a[0] += sreg;
#undef X
        } else if (KIND == 14) {
            // dependent DPP chain
#define X(i) "v_fmac_f32_dpp %0, %0, %13 row_newbcast:3" DPPM "s_nop 1\n"
            asm volatile("s_nop 1\n" R12(X) R12(X)
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11])
                : "v"(s[0]), "v"(m));
#undef X
        } else if (KIND == 15) {
            unsigned long long w[4] = {(unsigned long long)a[0], 3, 5, 7};
#define X(i) "v_mad_u64_u32 %0, vcc, %4, %5, %0\nv_mad_u64_u32 %1, vcc, %4, %5, %1\nv_mad_u64_u32 %2, vcc, %4, %5, %2\nv_mad_u64_u32 %3, vcc, %4, %5, %3\n"
            asm volatile(R12(X) : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "v"(r), "v"(rep) : "vcc");
#undef X
            a[0] += (float)(w[0] + w[1] + w[2] + w[3]);
        } else if (KIND == 16) {
            unsigned long long w[4] = {(unsigned long long)a[0], 3, 5, 7};
#define X(i) "v_lshl_add_u64 %0, %4, 0, %0\nv_lshl_add_u64 %1, %4, 0, %1\nv_lshl_add_u64 %2, %4, 0, %2\nv_lshl_add_u64 %3, %4, 0, %3\n"
            unsigned long long inc = r;
            asm volatile(R12(X) : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "v"(inc));
#undef X
            a[0] += (float)(w[0] + w[1] + w[2] + w[3]);
        } else if (KIND == 17) {
            double w[4] = {a[0], a[1], a[2], a[3]};
            double inc = m;
#define X(i) "v_add_f64 %0, %0, %4\nv_add_f64 %1, %1, %4\nv_add_f64 %2, %2, %4\nv_add_f64 %3, %3, %4\n"
            asm volatile(R12(X) : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "v"(inc));
#undef X
            a[0] += (float)(w[0] + w[1] + w[2] + w[3]);
        } else if (KIND == 18) {
#define X(i) "v_cndmask_b32 %" #i ", %12, %13, vcc\n"
            asm volatile(R12(X) R12(X) R12(X) R12(X)
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11])
                : "v"(s[0]), "v"(m) : "vcc");
#undef X
        } else if (KIND == 19) {
            unsigned long long msk = 0x5555aaaa5555aaaaull ^ r;
#define X(i) "v_cndmask_b32_e64 %" #i ", %12, %13, %14\n"
            asm volatile(R12(X) R12(X) R12(X) R12(X)
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11])
                : "v"(s[0]), "v"(m), "s"(msk));
#undef X
        } else if (KIND == 20) {
            unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0;
#define X(i) "v_cmp_lt_f32_e64 %0, %4, %5\nv_cmp_lt_f32_e64 %1, %5, %4\nv_cmp_lt_f32_e64 %2, %4, %6\nv_cmp_lt_f32_e64 %3, %6, %4\n"
            asm volatile(R12(X) : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3) : "v"(a[0]), "v"(s[0]), "v"(m));
#undef X
            a[1] += (float)(m0 + m1 + m2 + m3);
        } else if (KIND == 21) {
            // the usual pair: compare into an SGPR mask, select on it
            unsigned long long m0 = 0;
#define X(i) "v_cmp_lt_f32_e64 %12, %13, %" #i "\nv_cndmask_b32_e64 %" #i ", %13, %14, %12\n"
            asm volatile(R12(X) R12(X)
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+s"(m0)
                : "v"(s[0]), "v"(m));
#undef X
        } else if (KIND == 22) {
            unsigned long long m0 = r, m1 = ~r;
#define X(i) "s_and_b64 %0, %0, %1\ns_or_b64 %1, %0, %1\ns_and_b64 %0, %0, %1\ns_or_b64 %1, %0, %1\n"
            asm volatile(R12(X) : "+s"(m0), "+s"(m1) :: "scc");
#undef X
            a[1] += (float)(m0 + m1);
        } else if (KIND == 23) {
            int sv = 0;
#define X(i) "v_readlane_b32 %1, %0, 3\nv_writelane_b32 %0, %1, 5\nv_readlane_b32 %1, %0, 7\nv_writelane_b32 %0, %1, 9\n"
            asm volatile(R12(X) : "+v"(a[0]), "+s"(sv));
#undef X
        } else if (KIND == 24) {
            // v_max / v_min as the branch-free select alternative
#define X(i) "v_max_f32 %" #i ", %12, %" #i "\n"
            asm volatile(R12(X) R12(X) R12(X) R12(X)
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11])
                : "v"(s[0]), "v"(m));
#undef X
        } else if (KIND == 25) {
            // the mask logic of the box QP: two compares into SGPR masks, a scalar AND, a select on the result
            unsigned long long m0, m1, m2;
#define X(i) "v_cmp_lt_f32_e64 %12, %15, %" #i "\nv_cmp_gt_f32_e64 %13, %16, %" #i "\ns_and_b64 %14, %12, %13\nv_cndmask_b32_e64 %" #i ", %15, %16, %14\n"
            asm volatile(R12(X)
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]),
                  "=&s"(m0), "=&s"(m1), "=&s"(m2)
                : "v"(s[0]), "v"(m) : "scc");
#undef X
        } else if (KIND == 26) {
            // the same decision without the scalar unit: compare, select 0/1, compare-and-select again
            unsigned long long m0, m1;
            float t;
#define X(i) "v_cmp_lt_f32_e64 %12, %15, %" #i "\nv_cndmask_b32_e64 %14, %16, %15, %12\nv_cmp_gt_f32_e64 %13, %14, %" #i "\nv_cndmask_b32_e64 %" #i ", %15, %16, %13\n"
            asm volatile(R12(X)
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]),
                  "=&s"(m0), "=&s"(m1), "=&v"(t)
                : "v"(s[0]), "v"(m));
#undef X
        } else if (KIND == 8) {
            // plain fmac reading a different source each time through 2 DPP-free alternates: v_fma_f32 (VOP3)
#define X(i) "v_fma_f32 %" #i ", %12, %13, %" #i "\n"
            asm volatile(R12(X) R12(X) R12(X) R12(X)
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11])
                : "v"(s[0]), "v"(m));
#undef X
        }
    }
    long long t1 = clock64();
    float r = acc4[0] + acc4[1] + acc4[2] + acc4[3];
    for (int i = 0; i < 12; ++i) r += a[i] + p[i][0] + p[i][1];
    out[blockIdx.x * 64 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND> static void run(const char *name, int grid, int rep, float *out, long long *cyc)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, out, cyc, rep);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, out, cyc, rep);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += v;
    mean /= grid;
    const double n = 48.0 * rep;
    printf("%-44s grid %5d: %7.2f clk/instr (clock64), %7.3f ns/instr (wall), %8.3f ms\n", name, grid, mean / n, ms * 1e6 / n, ms);
}

int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 4096 * 64 * sizeof(float));
    hipMalloc(&cyc, 4096 * sizeof(long long));
    const int rep = 2000;
    for (int grid : {1024}) {
        run<0>("v_fmac_f32", grid, rep, out, cyc);
        run<8>("v_fma_f32 (VOP3)", grid, rep, out, cyc);
        run<1>("v_fmac_f32_dpp row_newbcast", grid, rep, out, cyc);
        run<4>("v_fmac_f32_dpp quad_perm", grid, rep, out, cyc);
        run<5>("v_fmac_f32_dpp row_shr:1", grid, rep, out, cyc);
        run<3>("v_mov_b32_dpp row_newbcast", grid, rep, out, cyc);
        run<2>("v_pk_fma_f32", grid, rep, out, cyc);
        run<15>("v_mad_u64_u32", grid, rep, out, cyc);
        run<16>("v_lshl_add_u64", grid, rep, out, cyc);
        run<17>("v_add_f64", grid, rep, out, cyc);
        run<18>("v_cndmask_b32", grid, rep, out, cyc);
        run<19>("v_cndmask_b32_e64 (sgpr mask)", grid, rep, out, cyc);
        run<20>("v_cmp_lt_f32_e64 -> sgpr", grid, rep, out, cyc);
        run<21>("v_cmp -> sgpr, v_cndmask on it (per instr)", grid, rep, out, cyc);
        run<22>("s_and_b64 / s_or_b64 dependent", grid, rep, out, cyc);
        run<23>("v_readlane / v_writelane dependent", grid, rep, out, cyc);
        run<24>("v_max_f32", grid, rep, out, cyc);
        run<25>("cmp, cmp, s_and, cndmask (per instr, dependent)", grid, rep, out, cyc);
        run<26>("cmp, cndmask, cmp, cndmask (per instr, dependent)", grid, rep, out, cyc);
        run<9>("v_fmac_f32, one accumulator (dependent)", grid, rep, out, cyc);
        run<10>("v_fmac_f32, two accumulators", grid, rep, out, cyc);
        run<11>("v_rcp_f32 dependent", grid, rep, out, cyc);
        run<12>("s_nop 0", grid, rep, out, cyc);
        run<13>("s_add_u32 dependent", grid, rep, out, cyc);
        run<14>("v_fmac_dpp dependent + s_nop 1 (per 2 instr)", grid, rep, out, cyc);
        run<6>("ds_read_b128 row-broadcast + 4 v_add", grid, rep, out, cyc);
        run<7>("ds_read_b128 per-lane + 4 v_add", grid, rep, out, cyc);
    }
    return 0;
}
