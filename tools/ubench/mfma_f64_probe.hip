// v_mfma_f64_16x16x4_f64: which lane holds which A / B / D entry?  (round 5: the float64 instantiation of lqr_mfma16_body.h
// assumes the float32 instruction's layout -- lane 16 g + j: A[i=j][k=g], B[k=g][j], D[4g+r][j] in element r.)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_f64_probe.hip -o /tmp/mfma_f64_probe && /tmp/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
// hypothesis H: a(lane) = A[lane % 16][lane / 16], b(lane) = B[lane / 16][lane % 16]
__global__ void probe(double *out, int kk)
{
    const int l = threadIdx.x, j = l & 15, g = l >> 4;
    // A[i][k] = (i + 1) if k == kk else 0 ; B[k][j] = 100 (j + 1) if k == kk else 0  ->  D[i][j] = 100 (i + 1)(j + 1)
    const double a = g == kk ? (double)(j + 1) : 0.0;
    const double b = g == kk ? 100.0 * (j + 1) : 0.0;
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[(kk * 64 + l) * 4 + r] = c[r];
}
int main()
{
    double *d;
    hipMalloc(&d, 4 * 64 * 4 * sizeof(double));
    for (int kk = 0; kk < 4; ++kk) probe<<<1, 64>>>(d, kk);
    double h[4 * 64 * 4];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int kk = 0; kk < 4; ++kk) {
        int same_as_f32 = 1;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const double v = h[(kk * 64 + l) * 4 + r];
                const int i = 4 * (l >> 4) + r, j = l & 15;
                if (v != 100.0 * (i + 1) * (j + 1)) same_as_f32 = 0;
            }
        printf("k = %d: D layout as float32's 16x16x4 (row 4 g + r, column j): %s\n", kk, same_as_f32 ? "yes" : "NO");
        if (!same_as_f32) {
            // decode: v / 100 = (i + 1)(j + 1); assume column j = lane % 16 and print the row each (lane group, element) holds
            for (int g = 0; g < 4; ++g) {
                printf("   lane group %d:", g);
                for (int r = 0; r < 4; ++r) {
                    const double v = h[(kk * 64 + 16 * g + 2) * 4 + r] / 100.0;      // lane j = 2: (i + 1) * 3
                    printf("  element %d -> row %g (col by lane 5: %g)", r, v / 3.0 - 1.0, h[(kk * 64 + 16 * g + 5) * 4 + r] / 100.0 / (v / 3.0) - 1.0);
                }
                printf("\n");
            }
        }
    }
    return 0;
}
