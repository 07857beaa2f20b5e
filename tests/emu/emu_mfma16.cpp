// emu_mfma16.cpp -- TEST INFRASTRUCTURE: host-side lockstep emulation of one gfx950 wavefront,
// used to run the product kernel source mpc.pytorch_amd/csrc/lqr_mfma16_body.h on a CPU-only box.
//
// The kernel body is written per lane against the `wv::` interface.  Here every lane is a
// ucontext fiber; a cross-lane intrinsic (mfma, readlane, shfl, ballot) deposits its operands
// in a double-buffered exchange area and yields, the scheduler runs all 64 lanes up to the
// same intrinsic, and on resumption each lane computes its result exactly as the hardware
// defines it (v_mfma_f32_16x16x4_f32: lane 16k+i holds A[i][k], lane 16k+j holds B[k][j],
// lane 16g+j register r holds D[4g+r][j]; fp32 fma chain over k = 0..3).
//
// Built by tests/test_emu_mfma16.py with the host clang++ of the ROCm toolchain (or g++).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#define MPC_EMULATE 1
#define MPC_DEV static inline
#define MPC_DEVM inline
// statistics of the row-wise pnqp (lane 0 of each 16-lane row counts): trips, factorisations, Armijo entries ...
extern "C" { long emu_stats[16]; }
#define MPC_STAT(i) do { if ((mpclqr::wv::lane() & 15) == 0) ++emu_stats[i]; } while (0)

namespace emu {
enum { NL = 64, STACK = 256 * 1024 };
struct Wave {
    ucontext_t main_ctx, ctx[NL];
    char *stack[NL];
    bool done[NL];
    int cur;
    int problem;
    // exchange area, two generations
    float fa[2][NL], fb[2][NL];
    double da[2][NL], db[2][NL];   // ... for the float64 instantiation of the one-problem-per-wavefront kernel
    float fq[2][NL][4];
    double fd[2][NL];
    int ia[2][NL];
    unsigned seq[NL];
    // LDS of the (single-wave) workgroup + the in-order queue of LDS-DMA instructions in flight
    alignas(16) unsigned char lds[160 * 1024];   // a CU's whole LDS (the staged kernels' rings use 40 KB of it, lqr_wave1 up to 150)
    struct Dma { unsigned char data[NL][16]; bool act[NL]; unsigned char seen[NL]; unsigned off; int size; };
    Dma q[192];      // (the hardware counter holds 63: issuing beyond that stalls until the oldest lands; landing LATER, as here, is the stricter test)
    int qn;
    int region_start;     // queue entries from here on belong to the current lockstep region
};
static Wave W;
static void (*g_body)(void);
// 0: DMA data lands at issue (earliest legal time); 1: it lands only when a counted wait forces it
// (latest legal time).  A correct kernel gives identical results in both.
static int g_dma_late = 0;
static void dma_land(int keep)
{
    while (W.qn > keep) {
        Wave::Dma &d = W.q[0];
        for (int l = 0; l < NL; ++l)
            if (d.act[l]) memcpy(W.lds + d.off + (unsigned)l * d.size, d.data[l], d.size);
        memmove(&W.q[0], &W.q[1], sizeof(Wave::Dma) * (W.qn - 1));
        --W.qn;
        if (W.region_start > 0) --W.region_start;
    }
}

static void yield_lane() { swapcontext(&W.ctx[W.cur], &W.main_ctx); }
static void trampoline()
{
    g_body();
    W.done[W.cur] = true;
    swapcontext(&W.ctx[W.cur], &W.main_ctx);
}
static void run_wave(int problem, void (*body)(void))
{
    g_body = body;
    W.problem = problem;
    for (int l = 0; l < NL; ++l) {
        if (!W.stack[l]) W.stack[l] = (char *)malloc(STACK);
        getcontext(&W.ctx[l]);
        W.ctx[l].uc_stack.ss_sp = W.stack[l];
        W.ctx[l].uc_stack.ss_size = STACK;
        W.ctx[l].uc_link = &W.main_ctx;
        makecontext(&W.ctx[l], trampoline, 0);
        W.done[l] = false;
        W.seq[l] = 0;
    }
    W.qn = 0;
    W.region_start = 0;
    memset(W.lds, 0xff, sizeof(W.lds));     // NaN pattern: reading a slot before its DMA landed shows
    for (;;) {
        int ndone = 0;
        for (int l = 0; l < NL; ++l) {
            if (W.done[l]) { ++ndone; continue; }
            W.cur = l;
            swapcontext(&W.main_ctx, &W.ctx[l]);
        }
        if (ndone == NL) break;
        // every lane is now parked at the same lockstep point: the region's DMA instructions are complete
        if (!g_dma_late) dma_land(0);
        W.region_start = W.qn;
    }
}
}  // namespace emu

namespace mpclqr {
namespace wv {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <typename T> struct vec4_of;
template <> struct vec4_of<float> { typedef f32x4 type; };
template <> struct vec4_of<double> { typedef f64x4 type; };
static inline int lane() { return emu::W.cur; }
static inline int problem() { return emu::W.problem; }
static inline f32x4 mfma(float a, float b, f32x4 c)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.fa[gen][l] = a;
    w.fb[gen][l] = b;
    emu::yield_lane();
    const int g = l >> 4, j = l & 15;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w.fa[gen][16 * k + i], w.fb[gen][16 * k + j], acc);
        d[r] = acc;
    }
    return d;
}
static inline float readlane(float x, int src)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.fa[gen][l] = x;
    emu::yield_lane();
    return w.fa[gen][src];
}
// float64 (v_mfma_f64_16x16x4_f64: the operand layouts of the float32 instruction, element r of lane group g of D is row 4 r + g)
static inline f64x4 mfma(double a, double b, f64x4 c)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.da[gen][l] = a;
    w.db[gen][l] = b;
    emu::yield_lane();
    const int g = l >> 4, j = l & 15;
    f64x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * r + g;          // (NOT the float32 instruction's 4 g + r: measured, tools/ubench/mfma_f64_probe.hip)
        double acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fma(w.da[gen][16 * k + i], w.db[gen][16 * k + j], acc);
        d[r] = acc;
    }
    return d;
}
static inline double readlane(double x, int src)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.da[gen][l] = x;
    emu::yield_lane();
    return w.da[gen][src];
}
static inline double shfl_xor(double x, int m)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.da[gen][l] = x;
    emu::yield_lane();
    return w.da[gen][l ^ m];
}
static inline double rcp(double x) { return 1.0 / x; }
static inline void pin(double &) {}
static inline int readlane_i(int x, int src)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.ia[gen][l] = x;
    emu::yield_lane();
    return w.ia[gen][src];
}
static inline float shfl_xor(float x, int m)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.fa[gen][l] = x;
    emu::yield_lane();
    return w.fa[gen][l ^ m];
}
static inline float sum_rows(float x)
{
    x += shfl_xor(x, 32);
    x += shfl_xor(x, 16);
    return x;
}
static inline unsigned long long ballot(bool c)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.ia[gen][l] = c ? 1 : 0;
    emu::yield_lane();
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) if (w.ia[gen][i]) m |= 1ull << i;
    return m;
}
static inline float rcp(float x) { return 1.0f / x; }
static inline float rcp_fast(float x) { return 1.0f / x; }
static inline bool first_lane(bool c) { return readlane_i((int)c, 0) != 0; }
static inline void pin(float &) {}
static inline void sched_fence() {}
// v_mfma_f32_4x4x1_16b_f32 cbsz:2 abid:ABID (lqr_dpp16_body.h): rows 4*ABID..+3 of a per-row outer product
template <int ABID> static inline f32x4 mfma4(float a, float b, f32x4 c)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.fa[gen][l] = a;
    emu::yield_lane();
    f32x4 d;
    for (int v = 0; v < 4; ++v) d[v] = fmaf(w.fa[gen][(l & ~15) + 4 * ABID + v], b, c[v]);
    return d;
}
// ---- DPP row_newbcast family (lqr_dpp16_body.h): lane N of the caller's 16-lane row -------------
template <int N> static inline float bcast(float x)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.fa[gen][l] = x;
    emu::yield_lane();
    return w.fa[gen][(l & ~15) + N];
}
template <int N> static inline void fmac_bcast(float &acc, float src, float mul) { acc = fmaf(bcast<N>(src), mul, acc); }
template <int N> static inline void fmac_bcast_settled(float &acc, float src, float mul) { acc = fmaf(bcast<N>(src), mul, acc); }
// wv::rows01 of lqr_mfma40.hip: rows 0 and 1 of x copied to every 16-lane row
static inline void rows01(float x, float &r0, float &r1)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.fa[gen][l] = x;
    emu::yield_lane();
    r0 = w.fa[gen][l & 15];
    r1 = w.fa[gen][16 + (l & 15)];
}
// wv::swap16 of lqr_mfma40.hip: lo = {a.row0, b.row0, a.row2, b.row2}, hi = {a.row1, b.row1, a.row3, b.row3}
static inline void swap16(float a, float b, float &lo, float &hi)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.fa[gen][l] = a;
    w.fb[gen][l] = b;
    emu::yield_lane();
    const bool odd = ((l >> 4) & 1) != 0;
    lo = odd ? w.fb[gen][l - 16] : w.fa[gen][l];
    hi = odd ? w.fb[gen][l] : w.fa[gen][l + 16];
}
// wv::lower_halves of lqr_mfma40.hip: {a.lanes 0..31, b.lanes 0..31}
static inline float lower_halves(float a, float b)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.fa[gen][l] = a;
    w.fb[gen][l] = b;
    emu::yield_lane();
    return l < 32 ? w.fa[gen][l] : w.fb[gen][l - 32];
}
template <int NN> static inline void dot_bcast(float &acc, float src, const float (&m)[NN])
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.fa[gen][l] = src;
    emu::yield_lane();
    // the kernel's two accumulation chains: even terms onto acc, odd terms onto a temporary
    float t = w.fa[gen][(l & ~15) + 1] * m[1];
    acc = fmaf(w.fa[gen][(l & ~15) + 0], m[0], acc);
    for (int i = 2; i < NN; i += 2) {
        t = fmaf(w.fa[gen][(l & ~15) + i + 1], m[i + 1], t);
        acc = fmaf(w.fa[gen][(l & ~15) + i], m[i], acc);
    }
    acc += t;
}
static inline void dot_bcast16(float &acc, float src, const float (&m)[16]) { dot_bcast<16>(acc, src, m); }
static inline void dot_bcast_u4(float &acc, float src, const float (&m)[4])
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.fa[gen][l] = src;
    emu::yield_lane();
    const int r = l & ~15;
    float t = w.fa[gen][r + 13] * m[1];
    acc = fmaf(w.fa[gen][r + 12], m[0], acc);
    t = fmaf(w.fa[gen][r + 15], m[3], t);
    acc = fmaf(w.fa[gen][r + 14], m[2], acc);
    acc += t;
}
static inline void dot_bcast12(float &acc, float src, const float (&m)[12]) { dot_bcast<12>(acc, src, m); }
static inline float row_sum(float x)
{
    emu::Wave &w = emu::W;
    const int xors[4] = {1, 2, 0, 0};
    (void)xors;
    // quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror -- as the hardware adds them
    for (int step = 0; step < 4; ++step) {
        const int l = w.cur, gen = w.seq[l]++ & 1;
        w.fa[gen][l] = x;
        emu::yield_lane();
        const int r = l & ~15, j = l & 15;
        int src;
        if (step == 0) src = j ^ 1;
        else if (step == 1) src = j ^ 2;
        else if (step == 2) src = (j & 8) | (7 - (j & 7));
        else src = 15 - j;
        x += w.fa[gen][r + src];
    }
    return x;
}
// wv::row_sum8 of lqr_mfma40.hip: the first three steps of row_sum
static inline float row_sum8(float x)
{
    emu::Wave &w = emu::W;
    for (int step = 0; step < 3; ++step) {
        const int l = w.cur, gen = w.seq[l]++ & 1;
        w.fa[gen][l] = x;
        emu::yield_lane();
        const int r = l & ~15, j = l & 15;
        const int src = step == 0 ? (j ^ 1) : (step == 1 ? (j ^ 2) : ((j & 8) | (7 - (j & 7))));
        x += w.fa[gen][r + src];
    }
    return x;
}
static inline float row_max(float x)
{
    emu::Wave &w = emu::W;
    for (int step = 0; step < 4; ++step) {
        const int l = w.cur, gen = w.seq[l]++ & 1;
        w.fa[gen][l] = x;
        emu::yield_lane();
        const int r = l & ~15, j = l & 15;
        int src;
        if (step == 0) src = j ^ 1;
        else if (step == 1) src = j ^ 2;
        else if (step == 2) src = (j & 8) | (7 - (j & 7));
        else src = 15 - j;
        x = fmaxf(x, w.fa[gen][r + src]);
    }
    return x;
}
static inline float swap1(float x)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.fa[gen][l] = x;
    emu::yield_lane();
    return w.fa[gen][l ^ 1];
}
static inline void store_f32x2_out(float *g, float a, float b) { g[0] = a; g[1] = b; }
static inline void absmax3(float &acc, float a, float b) { acc = fmaxf(acc, fmaxf(fabsf(a), fabsf(b))); }
// register-resident gains of lqr_dpp16.hip (accumulation registers a[4t..4t+3] there): one array per lane here
static f32x4 g_rg[64][64];
static inline void rg_put(int t, f32x4 v) { g_rg[emu::W.cur][t] = v; }
static inline f32x4 rg_get(int t) { return g_rg[emu::W.cur][t]; }
// wv::quad_sums of lqr_dpp16.hip: lane j of a row gets the row sum of p_(j & 3)
static inline float quad_sums(float p0, float p1, float p2, float p3, int j)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    float *buf = &w.fq[gen][l][0];
    buf[0] = p0; buf[1] = p1; buf[2] = p2; buf[3] = p3;
    emu::yield_lane();
    const int r = l & ~15, a = j & 3;
    // the hardware's order of additions: pairs, quads, then the quads rotated in
    float q[4];
    for (int k = 0; k < 4; ++k) {
        const int b = r + 4 * k;
        const float lo = w.fq[gen][b + 0][a] + w.fq[gen][b + 1][a], hi = w.fq[gen][b + 2][a] + w.fq[gen][b + 3][a];
        q[k] = lo + hi;
    }
    const int me = (l & 15) >> 2;
    const float s1 = q[me] + q[(me + 1) & 3];
    const float s2 = q[(me + 2) & 3] + q[(me + 3) & 3];
    return s1 + s2;
}
// wv::ring_sum of lqr_dpp16.hip: x + row_ror:4, then + row_ror:8 (lane j receives lane j - n of its row)
static inline float ring_sum(float x)
{
    emu::Wave &w = emu::W;
    for (int n = 4; n <= 8; n += 4) {
        const int l = w.cur, gen = w.seq[l]++ & 1;
        w.fa[gen][l] = x;
        emu::yield_lane();
        const int r = l & ~15, j = l & 15;
        x += w.fa[gen][r + ((j - n) & 15)];
    }
    return x;
}
static inline double row_sum_f64(double x)
{
    emu::Wave &w = emu::W;
    for (int step = 0; step < 4; ++step) {
        const int l = w.cur, gen = w.seq[l]++ & 1;
        w.fd[gen][l] = x;
        emu::yield_lane();
        const int r = l & ~15, j = l & 15;
        const int src = step == 0 ? (j ^ 1) : step == 1 ? (j ^ 2) : step == 2 ? ((j & 8) | (7 - (j & 7))) : 15 - j;
        x += w.fd[gen][r + src];
    }
    return x;
}
// lqr_wave1.hip's extra primitives: the LDS as an array of floats, doubles across lanes
static inline float &sm(int i) { return reinterpret_cast<float *>(emu::W.lds)[i]; }
static inline double shfl_xor_f64(double x, int m)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.fd[gen][l] = x;
    emu::yield_lane();
    return w.fd[gen][l ^ m];
}
static inline double readlane_f64(double x, int src)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.fd[gen][l] = x;
    emu::yield_lane();
    return w.fd[gen][src];
}
static inline double wave_sum_f64(double x)
{
    for (int off = 32; off > 0; off >>= 1) x += shfl_xor_f64(x, off);
    return x;
}
static inline bool any(bool c)
{
    emu::Wave &w = emu::W;
    const int l = w.cur, gen = w.seq[l]++ & 1;
    w.ia[gen][l] = c ? 1 : 0;
    emu::yield_lane();
    for (int i = 0; i < 64; ++i) if (w.ia[gen][i]) return true;
    return false;
}
static inline void store_f32x4(float *g, f32x4 v) { memcpy(g, &v, 16); }
static inline void store_f32x4_grad(float *g, f32x4 v) { memcpy(g, &v, 16); }
// v_readfirstlane: what the hardware would hand every lane is lane 0's value -- a call site whose value is NOT wave-uniform then goes wrong
// here as it does on the GPU (round 6: st_buf with a per-lane row forced uniform was caught by the GPU fuzzer, not by the identity this was)
static inline unsigned uniform_u32(unsigned x) { return (unsigned)readlane_i((int)x, 0); }
static inline const void *uniform_ptr(const void *q)
{
    const unsigned long long v = (unsigned long long)q;
    const unsigned lo = uniform_u32((unsigned)v), hi = uniform_u32((unsigned)(v >> 32));
    return (const void *)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}
static inline void store_f32_out(float *g, float v) { *g = v; }
static inline void store_f32_grad(float *g, float v) { *g = v; }
static inline bool uniform(bool c)
{
    // must be wave-uniform: check it
    const int v = readlane_i((int)c, 0);
    if (v != (int)c) { fprintf(stderr, "emu: non-uniform branch condition in lane %d\n", emu::W.cur); abort(); }
    return c;
}
static inline int uniform(int v)
{
    const int v0 = readlane_i(v, 0);
    if (v0 != v) { fprintf(stderr, "emu: non-uniform value in lane %d\n", emu::W.cur); abort(); }
    return v;
}
static inline int ctz64(unsigned long long m) { return __builtin_ctzll(m); }
// ---- LDS-DMA.  Between two lockstep points every lane runs the same instruction sequence (lanes
// that branch around an instruction are inactive for it), and each DMA instruction of such a region
// has its own LDS destination, so (offset, size) identifies the instruction a lane is taking part in.
static inline void dma_n(const void *g, unsigned off, int size, bool active = true)
{
    emu::Wave &w = emu::W;
    const int l = w.cur;
    emu::Wave::Dma *d = nullptr;
    for (int i = w.region_start; i < w.qn; ++i)
        if (w.q[i].off == off && w.q[i].size == size && w.q[i].seen[l] == 0) { d = &w.q[i]; break; }
    if (!d) {
        if (w.qn >= 192) { fprintf(stderr, "emu: DMA queue overflow\n"); abort(); }
        d = &w.q[w.qn++];
        memset(d->act, 0, sizeof(d->act));
        memset(d->seen, 0, sizeof(d->seen));
        d->off = off;
        d->size = size;
    }
    d->seen[l] = 1;
    // an exec-masked lane still passes the instruction: it takes its place in program order
    if (!active) return;
    d->act[l] = true;
    memcpy(d->data[l], g, size);
}
static inline void dma16(const void *g, unsigned off) { dma_n(g, off, 16); }
// immediate offset: moves source and destination together (g is the pointer biased by -IMM)
enum { DMA_PLAIN = 0, DMA_C = 1, DMA_LAST = 2 };
template <int IMM, int KIND = DMA_PLAIN> static inline void dma16_at(const void *g, unsigned mid)
{
    dma_n((const char *)g + IMM, (unsigned)((int)mid + IMM), 16);
}
template <int IMM> static inline void dma16_once_at(const void *g, unsigned mid) { dma_n((const char *)g + IMM, (unsigned)((int)mid + IMM), 16); }
template <int IMM> static inline void dma16_at_if(bool active, const void *g, unsigned mid)
{
    dma_n((const char *)g + IMM, (unsigned)((int)mid + IMM), 16, active);
}
static inline void dma16_once(const void *g, unsigned off) { dma_n(g, off, 16); }
static inline void store_out(float *g, float v) { *g = v; }
static inline void dma16_if(bool active, const void *g, unsigned off) { dma_n(g, off, 16, active); }
static inline void dma4(const void *g, unsigned off) { dma_n(g, off, 4); }
static inline void dma4_if(bool active, const void *g, unsigned off) { dma_n(g, off, 4, active); }
// buffer_store_dword through a raw buffer of `nbytes`: out-of-range lanes store nothing
static inline void st_buf(float *base, unsigned nbytes, unsigned voff, float v)
{
    if ((unsigned long)voff + 4ul <= (unsigned long)nbytes) memcpy((char *)base + voff, &v, 4);
}
// (the hardware's v_readfirstlane under a partial exec mask reads the first ACTIVE lane; the emulator's cross-lane operations are wave-wide
// rendezvous, and these stores sit behind `if (store)` and the like: the uniformity of their rows is the GPU fuzzer's to check)
static inline void st_buf_u(float *base, unsigned nbytes, unsigned voff, float v) { st_buf(base, nbytes, voff, v); }
// buffer_load ... lds (the padded 32/8 instantiation's gathers): G bytes per lane from base + voff; a lane whose access lies
// beyond the buffer's `nbytes` writes ZERO into LDS (the hardware's range check, tools/ubench/buffer_lds_probe.hip)
template <int G> static inline void dma_buf(bool active, const void *base, unsigned nbytes, unsigned voff, unsigned off)
{
    static const char zero[16] = {0};
    const bool in = (unsigned long)voff + (unsigned long)G <= (unsigned long)nbytes;
    dma_n(in ? (const char *)base + voff : zero, off, G, active);
}
// ... with an immediate offset IMM (round 6, the padded 12/4 kernel): the instruction offset moves the LDS destination AND the buffer
// offset together, like dma16_at -- `voff` is the lane's source offset biased by -IMM + BIAS and `base_biased` the block's base minus BIAS
// (the bias keeps voff non-negative; the range check sees voff + IMM against nbytes + BIAS), `anchor` the LDS address IMM counts from
template <int IMM, int BIAS> static inline void dma_buf_at(const void *base_biased, unsigned nbytes_biased, unsigned voff, unsigned anchor)
{
    static const char zero[16] = {0};
    const unsigned long o = (unsigned long)voff + (unsigned long)IMM;        // what the hardware adds up and range-checks
    const bool in = o + 4ul <= (unsigned long)nbytes_biased && o >= (unsigned long)BIAS;
    dma_n(in ? (const char *)base_biased + o : zero, anchor + (unsigned)IMM, 4, true);
}
// global_load_lds_dword with an immediate offset (g = the lane's pointer biased by -IMM)
template <int IMM> static inline void dma4_at_if(bool active, const void *g, unsigned anchor)
{
    dma_n((const char *)g + IMM, anchor + (unsigned)IMM, 4, active);
}
template <int N> static inline void dma_wait()
{
    // lockstep point: every lane has issued its part of the preceding DMA instructions
    (void)readlane_i(0, 0);
    if (emu::W.cur == 0) emu::dma_land(emu::g_dma_late ? N : 0);
}
static inline unsigned load_uniform_u32(const unsigned *g) { return *g; }
static inline float bits_f32(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned f32_bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float lds_f32(unsigned off)
{
    float v;
    memcpy(&v, emu::W.lds + off, 4);
    return v;
}
static inline f32x4 lds_f32x4(unsigned off)
{
    if (off & 15) { fprintf(stderr, "emu: misaligned ds_read_b128\n"); abort(); }
    f32x4 v;
    memcpy(&v, emu::W.lds + off, 16);
    return v;
}
static inline void lds_store_f32(unsigned off, float v) { memcpy(emu::W.lds + off, &v, 4); }
static inline void lds_store_f32x4(unsigned off, f32x4 v)
{
    if (off & 15) { fprintf(stderr, "emu: misaligned ds_write_b128\n"); abort(); }
    memcpy(emu::W.lds + off, &v, 16);
}
// the type-parametrised accessors of lqr_mfma16_body.h (one LDS here: the type only picks the width)
template <typename T> static inline void dma16r(const void *g, unsigned off) { dma_n(g, off, 16); }
template <typename T> static inline void dma4r(const void *g, unsigned off) { dma_n(g, off, 4); }
template <typename T> static inline T lds_r(unsigned off) { T v; memcpy(&v, emu::W.lds + off, sizeof(T)); return v; }
template <typename T> static inline typename vec4_of<T>::type lds_r4(unsigned off)
{
    if (off & 15) { fprintf(stderr, "emu: misaligned ds_read_b128\n"); abort(); }
    typename vec4_of<T>::type v;
    memcpy(&v, emu::W.lds + off, 4 * sizeof(T));
    return v;
}
static inline void lds_store_r(unsigned off, float v) { memcpy(emu::W.lds + off, &v, 4); }
static inline void lds_store_r(unsigned off, double v) { memcpy(emu::W.lds + off, &v, 8); }
static inline void lds_store_r4(unsigned off, f32x4 v) { memcpy(emu::W.lds + off, &v, 16); }
static inline void lds_store_r4(unsigned off, f64x4 v) { memcpy(emu::W.lds + off, &v, 32); }
// the hardware executes a wave's DS instructions in order; the lane-serial emulator needs every
// lane to have passed the preceding stores/loads before any lane goes on
static inline void lds_sync() { (void)readlane_i(0, 0); }
static inline void fence_own_stores() {}
}  // namespace wv
}  // namespace mpclqr

#include "../../mpc.pytorch_amd/csrc/lqr_mfma16_body.h"
// ... and its float64 instantiation (namespace mfma16d)
#define MPC_M16_F64
#include "../../mpc.pytorch_amd/csrc/lqr_mfma16_body.h"
#undef MPC_M16_F64
#include "../../mpc.pytorch_amd/csrc/lqr_dpp16_body.h"
#include "../../mpc.pytorch_amd/csrc/lqr_mfma40_body.h"

static const mpclqr::StepParams<float> *g_p;
template <bool FULL> static void body()
{
    const mpclqr::StepParams<float> &p = *g_p;
    if (p.bound_mode != MPC_BOUND_NONE) mpclqr::mfma16::step_problem<FULL, 2>(p);
    else if (p.zero_mask) mpclqr::mfma16::step_problem<FULL, 1>(p);
    else mpclqr::mfma16::step_problem<FULL, 0>(p);
}

extern "C" void emu_set_dma_late(int late) { emu::g_dma_late = late; }

extern "C" int emu_lqr_step_mfma16(const mpc_lqr_problem *p, const mpc_lqr_options *o, const mpc_lqr_outputs *out,
                                   int force_general)
{
    if (p->dtype != MPC_F32) return MPC_E_DTYPE;
    mpclqr::StepParams<float> sp = mpclqr::make_params<float>(p, o, out);
    if (!(sp.ns >= 1 && sp.ns <= 12 && sp.nc >= 1 && sp.nc <= 4 && sp.max_ls >= 1 && sp.max_ls <= 16)) return MPC_E_DIMS;
    if (!sp.new_x || !sp.new_u) return MPC_E_NULL;
    static float *kk_buf = nullptr;
    static size_t kk_cap = 0;
    const size_t need = (size_t)sp.T * sp.B * 64;
    if (need > kk_cap) { free(kk_buf); kk_buf = (float *)malloc(need * sizeof(float)); kk_cap = need; }
    for (size_t i = 0; i < need; ++i) kk_buf[i] = NAN;
    sp.Kk = kk_buf;
    g_p = &sp;
    const bool full = sp.ns == 12 && sp.nc == 4 && !force_general;
    for (int b = 0; b < sp.B; ++b) emu::run_wave(b, full ? body<true> : body<false>);
    return 0;
}

static const mpclqr::StepParams<double> *g_pd;
template <bool FULL> static void body_f64()
{
    const mpclqr::StepParams<double> &p = *g_pd;
    if (p.bound_mode != MPC_BOUND_NONE) mpclqr::mfma16d::step_problem<FULL, 2>(p);
    else if (p.zero_mask) mpclqr::mfma16d::step_problem<FULL, 1>(p);
    else mpclqr::mfma16d::step_problem<FULL, 0>(p);
}

extern "C" int emu_lqr_step_mfma16_f64(const mpc_lqr_problem *p, const mpc_lqr_options *o, const mpc_lqr_outputs *out,
                                       int force_general)
{
    if (p->dtype != MPC_F64) return MPC_E_DTYPE;
    mpclqr::StepParams<double> sp = mpclqr::make_params<double>(p, o, out);
    if (!(sp.ns >= 1 && sp.ns <= 12 && sp.nc >= 1 && sp.nc <= 4 && sp.max_ls >= 1 && sp.max_ls <= 16)) return MPC_E_DIMS;
    if (!sp.new_x || !sp.new_u) return MPC_E_NULL;
    static double *kk_buf = nullptr;
    static size_t kk_cap = 0;
    const size_t need = (size_t)sp.T * sp.B * 64;
    if (need > kk_cap) { free(kk_buf); kk_buf = (double *)aligned_alloc(16, need * sizeof(double)); kk_cap = need; }
    for (size_t i = 0; i < need; ++i) kk_buf[i] = NAN;
    sp.Kk = kk_buf;
    g_pd = &sp;
    const bool full = sp.ns == 12 && sp.nc == 4 && !force_general;
    for (int b = 0; b < sp.B; ++b) emu::run_wave(b, full ? body_f64<true> : body_f64<false>);
    return 0;
}

// ---- the 4-problems-per-wave DPP kernel (lqr_dpp16_body.h) -------------------------------------
static void body_dpp16()
{
    const mpclqr::StepParams<float> &p = *g_p;
    if (p.bound_mode != MPC_BOUND_NONE) mpclqr::dpp16::step_wave<2>(p);
    else if (p.zero_mask) mpclqr::dpp16::step_wave<1>(p);
    else if (p.T <= mpclqr::dpp16::RG_STEPS) mpclqr::dpp16::step_wave<0>(p);
    else mpclqr::dpp16::step_wave<3>(p);
}

extern "C" int emu_lqr_step_dpp16(const mpc_lqr_problem *p, const mpc_lqr_options *o, const mpc_lqr_outputs *out)
{
    if (p->dtype != MPC_F32) return MPC_E_DTYPE;
    mpclqr::StepParams<float> sp = mpclqr::make_params<float>(p, o, out);
#ifdef MPC_DPP16_PAD          // the padded instantiation (csrc/Makefile: lqr_dpp16_pad.o): any n_state <= 12, n_ctrl <= 4
    if (!(sp.ns >= 1 && sp.ns <= 12 && sp.nc >= 1 && sp.nc <= 4 && sp.max_ls >= 1 && sp.max_ls <= 16)) return MPC_E_DIMS;
#else
    if (!(sp.ns == 12 && sp.nc == 4 && sp.max_ls >= 1 && sp.max_ls <= 16)) return MPC_E_DIMS;
#endif
    if (!sp.new_x || !sp.new_u) return MPC_E_NULL;
    static float *kk_buf = nullptr;
    static size_t kk_cap = 0;
    const size_t need = (size_t)sp.T * sp.B * (128 + 16) + 4;     // two records + the second trial's trajectory
    if (need > kk_cap) { free(kk_buf); kk_buf = (float *)aligned_alloc(16, (need * sizeof(float) + 15) / 16 * 16); kk_cap = need; }
    for (size_t i = 0; i < need; ++i) kk_buf[i] = NAN;
    sp.Kk = kk_buf;
    g_p = &sp;
    for (int w = 0; 4 * w < sp.B; ++w) emu::run_wave(w, body_dpp16);
    return 0;
}

static const mpclqr::dpp16::KktArgs *g_k;
static void body_kkt16() { mpclqr::dpp16::kkt_wave(*g_p, *g_k); }

extern "C" int emu_kkt_dpp16(const mpc_lqr_problem *p, const float *dx, const float *du, const float *dl_dx,
                             float *dC, float *dc, float *dF, float *df, float *dx_init)
{
    mpclqr::StepParams<float> sp = mpclqr::make_params<float>(p, nullptr, nullptr);
    if (!(sp.ns == 12 && sp.nc == 4)) return MPC_E_DIMS;
    mpclqr::dpp16::KktArgs k;
    k.dx = dx; k.du = du; k.dl_dx = dl_dx; k.dC = dC; k.dc = dc; k.dF = dF; k.df = df; k.dx_init = dx_init;
    g_p = &sp;
    g_k = &k;
    for (int w = 0; 4 * w < sp.B; ++w) emu::run_wave(w, body_kkt16);
    return 0;
}

static const mpclqr::dpp16::KktFusedArgs *g_kf;
template <bool MASKED> static void body_kkt_fused()
{
    if (g_p->T > mpclqr::dpp16::RG_STEPS) mpclqr::dpp16::kkt_fused_wave<MASKED, true>(*g_p, *g_kf);      // the long-horizon instantiation
    else mpclqr::dpp16::kkt_fused_wave<MASKED>(*g_p, *g_kf);
}

// the whole KKT backward in one (emulated) launch: lqr_dpp16_body.h, kkt_fused_wave
extern "C" int emu_kkt_fused(const mpc_lqr_problem *p, const mpc_lqr_options *o, const float *dl_dx, const float *dl_du,
                             float *dC, float *dc, float *dF, float *df, float *dx_init, float *dx_out, float *du_out,
                             int *status)
{
    mpc_lqr_outputs out;
    memset(&out, 0, sizeof(out));
    out.status = status;
    mpclqr::StepParams<float> sp = mpclqr::make_params<float>(p, o, &out);
#ifdef MPC_DPP16_PAD
    if (!(sp.ns >= 1 && sp.ns <= 12 && sp.nc >= 1 && sp.nc <= 4)) return MPC_E_DIMS;      // the padded instantiation (launch_kkt_fused_dpp16_pad)
    sp.c_symmetric = 1;
    sp.zero_mask = nullptr;
    sp.has_delta = 0;
#else
    if (!(sp.ns == 12 && sp.nc == 4)) return MPC_E_DIMS;
#endif
    const size_t need = (size_t)sp.T * sp.B * (mpclqr::dpp16::KF_VBLK + 24 + (sp.T > mpclqr::dpp16::RG_STEPS ? 64 : 0)) + 4;
    float *ws = (float *)aligned_alloc(16, (need * sizeof(float) + 15) / 16 * 16);
    for (size_t i = 0; i < need; ++i) ws[i] = NAN;
    mpclqr::dpp16::KktFusedArgs k;
    k.dl_dx = dl_dx; k.dl_du = dl_du; k.dC = dC; k.dc = dc; k.dF = dF; k.df = df; k.dx_init = dx_init;
    k.dx_out = dx_out; k.du_out = du_out; k.vws = ws;
    k.decay = o ? (float)o->linesearch_decay : 0.2f;
    k.max_ls = o ? o->max_linesearch_iter : 10;
    g_p = &sp;
    g_kf = &k;
    const bool masked = sp.bound_mode != MPC_BOUND_NONE;
    for (int w = 0; 4 * w < sp.B; ++w) emu::run_wave(w, masked ? body_kkt_fused<true> : body_kkt_fused<false>);
    free(ws);
    return 0;
}

// ---- env_dynamics.h on the host: the same source the linearisation / rollout kernels compile ----
template <typename real>
static void env_linearize_host(int kind, const real *params, double dt, double umax, long N, const real *x,
                               const real *u, real *nxt, real *F, real *f)
{
    mpclqr::EnvDesc<real> e;
    e.kind = kind; e.params = params; e.dt = (real)dt; e.u_max = (real)umax;
    const int ns = mpclqr::env_ns(kind), n = ns + 1;
    for (long i = 0; i < N; ++i) {
        real out[5], J[30];
        mpclqr::env_step<real>(e, x + i * ns, u[i], out, J);
        for (int r = 0; r < ns; ++r) {
            real acc = out[r];
            for (int j = 0; j < ns; ++j) acc -= J[r * n + j] * x[i * ns + j];
            acc -= J[r * n + ns] * u[i];
            f[i * ns + r] = acc;
            nxt[i * ns + r] = out[r];
            for (int j = 0; j < n; ++j) F[(i * ns + r) * n + j] = J[r * n + j];
        }
    }
}
extern "C" void emu_env_linearize_f64(int kind, const double *params, double dt, double umax, long N, const double *x,
                                      const double *u, double *nxt, double *F, double *f)
{
    env_linearize_host<double>(kind, params, dt, umax, N, x, u, nxt, F, f);
}
extern "C" void emu_env_linearize_f32(int kind, const float *params, double dt, double umax, long N, const float *x,
                                      const float *u, float *nxt, float *F, float *f)
{
    env_linearize_host<float>(kind, params, dt, umax, N, x, u, nxt, F, f);
}

// ---- lqr_tiny_body.h on the host: no cross-lane traffic, so a plain loop over the problems -------
#include "../../mpc.pytorch_amd/csrc/lqr_tiny_body.h"
template <typename real> static int tiny_host(const mpc_lqr_problem *p, const mpc_lqr_options *o, const mpc_lqr_outputs *out)
{
    mpclqr::StepParams<real> sp = mpclqr::make_params<real>(p, o, out);
    if (!mpclqr::tiny::shape_supported(sp.ns, sp.nc)) return MPC_E_DIMS;
    if (!sp.new_x || !sp.new_u) return MPC_E_NULL;
    const size_t gains = (size_t)sp.T * sp.B * (sp.ns + 1), need = 2 * gains;     // + one trial column per problem (G = 1)
    real *kw = (real *)malloc(need * sizeof(real));
    for (size_t i = 0; i < need; ++i) kw[i] = (real)NAN;
    for (int b = 0; b < sp.B; ++b) {
        switch (sp.ns) {
        case 1: mpclqr::tiny::lqr_step_problem<real, 1>(sp, b, kw, kw + gains, mpclqr::tiny::OneLane()); break;
        case 2: mpclqr::tiny::lqr_step_problem<real, 2>(sp, b, kw, kw + gains, mpclqr::tiny::OneLane()); break;
        case 3: mpclqr::tiny::lqr_step_problem<real, 3>(sp, b, kw, kw + gains, mpclqr::tiny::OneLane()); break;
        case 4: mpclqr::tiny::lqr_step_problem<real, 4>(sp, b, kw, kw + gains, mpclqr::tiny::OneLane()); break;
        case 5: mpclqr::tiny::lqr_step_problem<real, 5>(sp, b, kw, kw + gains, mpclqr::tiny::OneLane()); break;
        case 6: mpclqr::tiny::lqr_step_problem<real, 6>(sp, b, kw, kw + gains, mpclqr::tiny::OneLane()); break;
        }
    }
    free(kw);
    return 0;
}
extern "C" int emu_lqr_step_tiny(const mpc_lqr_problem *p, const mpc_lqr_options *o, const mpc_lqr_outputs *out)
{
    return p->dtype == MPC_F32 ? tiny_host<float>(p, o, out) : tiny_host<double>(p, o, out);
}


// ---- lqr_mfma40_body.h: the n_state = 32, n_ctrl = 8 sweep (one emulated wavefront per problem) ----
static int g_m40_full = 0;
template <int MODE> static void body_mfma40_mode()
{
    if (g_m40_full) mpclqr::mfma40::step_wave<MODE>(*g_p, g_p->K, g_p->k);
    else (void)mpclqr::mfma40::sweep_wave<MODE>(*g_p, g_p->K, g_p->k);
}
static void body_mfma40()
{
    if (g_p->bound_mode != MPC_BOUND_NONE) body_mfma40_mode<2>();
    else if (g_p->zero_mask) body_mfma40_mode<1>();
    else body_mfma40_mode<0>();
}
static int g_m40_record = 1;
extern "C" void emu_mfma40_full(int full) { g_m40_full = full; }
extern "C" void emu_mfma40_record(int on) { g_m40_record = on; }
extern "C" int emu_lqr_sweep_mfma40(const mpc_lqr_problem *p, const mpc_lqr_options *o, const mpc_lqr_outputs *out)
{
    if (p->dtype != MPC_F32) return MPC_E_DTYPE;
    mpclqr::StepParams<float> sp = mpclqr::make_params<float>(p, o, out);
#ifdef MPC_MFMA40_PAD
    // the padded instantiation (any n_state <= 32, n_ctrl <= 8): K / k are the kernel's own padded gains, the caller's go by K_user
    if (sp.ns < 1 || sp.ns > 32 || sp.nc < 1 || sp.nc > 8) return MPC_E_DIMS;
    if (MPC_MFMA40_PAD == 16 && (sp.ns % 4 || sp.nc % 4)) return MPC_E_DIMS;
    const size_t ngain = (size_t)sp.T * sp.B * (256 + 8);
    float *gains = (float *)aligned_alloc(16, (ngain * sizeof(float) + 15) / 16 * 16);
    for (size_t i = 0; i < ngain; ++i) gains[i] = NAN;
    sp.K_user = sp.K;
    sp.k_user = sp.k;
    sp.K = gains;
    sp.k = gains + (size_t)sp.T * sp.B * 256;
#else
    if (!(sp.ns == 32 && sp.nc == 8) || !sp.K || !sp.k) return MPC_E_DIMS;
#endif
    // the (M, Quu, m) record of the constrained modes' priced rollout (capi.hip takes it out of the workspace)
    const size_t need = (size_t)sp.T * sp.B * (mpclqr::mfma40::PREC + mpclqr::mfma40::PSCR) + 4;
    float *rec = (float *)aligned_alloc(16, (need * sizeof(float) + 15) / 16 * 16);
    for (size_t i = 0; i < need; ++i) rec[i] = NAN;
    sp.Kk = g_m40_record ? rec : nullptr;
    g_p = &sp;
    for (int b = 0; b < sp.B; ++b) emu::run_wave(b, body_mfma40);
    free(rec);
#ifdef MPC_MFMA40_PAD
    free(gains);
#endif
    return 0;
}

// the fused KKT backward of that shape (kkt_fused_wave of lqr_mfma40_body.h): dx, du, dx_init, df and the costates
// lambda_{t+1}, dlambda_{t+1} parked in the first 64 words of the dF_t blocks (the outer products are kkt_outer_kernel's)
static const mpclqr::mfma40::KktArgs40 *g_k40;
template <int MODE> static void body_kkt40() { mpclqr::mfma40::kkt_fused_wave<MODE>(*g_p, g_p->K, g_p->k, *g_k40); }
extern "C" int emu_kkt_fused_mfma40(const mpc_lqr_problem *p, const mpc_lqr_options *o, const float *dl_dx, const float *dl_du,
                                    float *dF, float *df, float *dx_init, float *dx_out, float *du_out, int *status)
{
    mpc_lqr_outputs out;
    memset(&out, 0, sizeof(out));
    out.status = status;
    mpclqr::StepParams<float> sp = mpclqr::make_params<float>(p, o, &out);
#ifdef MPC_MFMA40_PAD
    if (!(sp.ns >= 1 && sp.ns <= 32 && sp.nc >= 1 && sp.nc <= 8)) return MPC_E_DIMS;      // the padded instantiation (launch_kkt_fused_mfma40_pad)
    sp.zero_mask = nullptr;
    sp.has_delta = 0;
#else
    if (!(sp.ns == 32 && sp.nc == 8)) return MPC_E_DIMS;
#endif
    if (sizeof(emu::W.lds) < mpclqr::mfma40::KLDS_TOTAL) return MPC_E_DIMS;
    const size_t TB = (size_t)sp.T * sp.B, need = TB * (256 + 8 + 1024 + 64) + 4;
    float *ws = (float *)aligned_alloc(16, (need * sizeof(float) + 15) / 16 * 16);
    for (size_t i = 0; i < need; ++i) ws[i] = NAN;
    sp.K = ws;
    sp.k = ws + TB * 256;
    sp.new_x = dx_out;
    sp.new_u = du_out;
    sp.ls_decay = 0.2f;
    sp.max_ls = 10;
    sp.c_symmetric = 1;
    mpclqr::mfma40::KktArgs40 k;
    k.dl_dx = dl_dx; k.dl_du = dl_du; k.dF = dF; k.df = df; k.dx_init = dx_init;
    k.Vws = ws + TB * 264; k.vgws = ws + TB * (264 + 1024);
    g_p = &sp;
    g_k40 = &k;
    const bool masked = sp.bound_mode != MPC_BOUND_NONE;
    for (int b = 0; b < sp.B; ++b) emu::run_wave(b, masked ? body_kkt40<1> : body_kkt40<0>);
    free(ws);
    return 0;
}

// ---- lqr_wave1_body.h: one wavefront per problem, n_ctrl = 1, n_state <= 6, the problem in LDS ----
#include "../../mpc.pytorch_amd/csrc/lqr_wave1_body.h"
template <int NS> static void body_wave1_ns() { mpclqr::wave1::step_wave<NS>(*g_p); }
static void body_wave1()
{
    switch (g_p->ns) {
    case 1: body_wave1_ns<1>(); break;
    case 2: body_wave1_ns<2>(); break;
    case 3: body_wave1_ns<3>(); break;
    case 4: body_wave1_ns<4>(); break;
    case 5: body_wave1_ns<5>(); break;
    case 6: body_wave1_ns<6>(); break;
    }
}
extern "C" int emu_lqr_step_wave1(const mpc_lqr_problem *p, const mpc_lqr_options *o, const mpc_lqr_outputs *out)
{
    if (p->dtype != MPC_F32) return MPC_E_DTYPE;
    mpclqr::StepParams<float> sp = mpclqr::make_params<float>(p, o, out);
    if (!mpclqr::wave1::shape_supported(sp)) return MPC_E_DIMS;
    if (!sp.new_x || !sp.new_u) return MPC_E_NULL;
    if ((size_t)mpclqr::wave1::layout(sp).total * 16 > sizeof(emu::W.lds)) return MPC_E_DIMS;
    g_p = &sp;
    for (int w = 0; w < (sp.B + 3) / 4; ++w) emu::run_wave(w, body_wave1);
    return 0;
}

// ---- the loop-free scalar QP of lqr_wave1_body.h against the loop it replaces, 64 cases per emulated wavefront ----
static const float *g_qp_in;     // [n][5]: H, q, lb, ub, x0
static float *g_qp_out;          // [n][5]: x, Hfree, is_free, ret, conv
static int g_qp_n, g_qp_iter;
static void body_pnqp1_fast()
{
    const int i = mpclqr::wv::problem() * 64 + mpclqr::wv::lane();
    const int k = i < g_qp_n ? i : g_qp_n - 1;
    const float *in = g_qp_in + 5 * k;
    float x = in[4], Hf;
    bool fr, conv;
    const int ret = mpclqr::wave1::pnqp1_fast(in[0], in[1], in[2], in[3], x, Hf, fr, g_qp_iter, conv);
    if (i < g_qp_n) {
        float *o = g_qp_out + 5 * i;
        o[0] = x, o[1] = Hf, o[2] = fr, o[3] = (float)ret, o[4] = conv;
    }
}
extern "C" void emu_pnqp1_pair(int n, int n_iter, const float *in, float *fast, float *loop)
{
    g_qp_in = in, g_qp_out = fast, g_qp_n = n, g_qp_iter = n_iter;
    for (int w = 0; w < (n + 63) / 64; ++w) emu::run_wave(w, body_pnqp1_fast);
    for (int i = 0; i < n; ++i) {
        const float *a = in + 5 * i;
        float x = a[4], Hf;
        bool fr, conv;
        const int ret = mpclqr::tiny::pnqp1<float>(a[0], a[1], a[2], a[3], x, Hf, fr, n_iter, conv);
        float *o = loop + 5 * i;
        o[0] = x, o[1] = Hf, o[2] = fr, o[3] = (float)ret, o[4] = conv;
    }
}
