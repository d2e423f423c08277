// Layout probe for v_mfma_i32_16x16x64_i8 on gfx950 (integer-exact Gram matrix of the initial pass, wide cohorts):
// D = A B with A 16 x 64 and B 64 x 16 signed bytes.  Assumed: lane l = (i = l & 15, kb = l >> 4) holds A[i][16 kb .. 16 kb + 15]
// (byte b of its four registers = k index 16 kb + b), the same for B[k][j] with j = l & 15; D[4 (l >> 4) + r][l & 15] in register r.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libmfma_i8_test.so mfma_i8_test.hip   (extern "C" ubench_main)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const signed char *A, const signed char *B, int *D, long long *cyc)
{
    const int l = threadIdx.x, i = l & 15, kb = l >> 4;
    v4i a, b;
    for (int r = 0; r < 4; r++) {
        int wa = 0, wb = 0;
        for (int q = 0; q < 4; q++) {
            const int kk = 16 * kb + 4 * r + q;
            wa |= ((int) (unsigned char) A[i * 64 + kk]) << (8 * q);          // A[i][kk]
            wb |= ((int) (unsigned char) B[kk * 16 + i]) << (8 * q);          // B[kk][j = i]
        }
        a[r] = wa; b[r] = wb;
    }
    v4i acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(4 * kb + r) * 16 + i] = acc[r];
    v4i t = {0, 0, 0, 0};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 1000; it++) {
        t = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, t, 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    v4i u0 = {0, 0, 0, 0}, u1 = u0, u2 = u0, u3 = u0;
    const long long t2 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 1000; it++) {
        u0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, u0, 0, 0, 0);
        u1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, u1, 0, 0, 0);
        u2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, u2, 0, 0, 0);
        u3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, u3, 0, 0, 0);
    }
    const long long t3 = __builtin_amdgcn_s_memtime();
    if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t3 - t2; }
    if (t[0] + u0[0] + u1[1] + u2[2] + u3[3] == 12345) D[0] = t[1];
}
extern "C" int ubench_main()
{
    signed char hA[16 * 64], hB[64 * 16];
    int hD[256], ref[256];
    srand(7);
    for (int i = 0; i < 16 * 64; i++) { hA[i] = (signed char) (rand() % 256 - 128); hB[i] = (signed char) (rand() % 256 - 128); }
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { int s = 0; for (int kk = 0; kk < 64; kk++) s += (int) hA[i * 64 + kk] * (int) hB[kk * 16 + j]; ref[i * 16 + j] = s; }
    signed char *A, *B; int *D; long long *cyc;
    hipMalloc(&A, sizeof(hA)); hipMalloc(&B, sizeof(hB)); hipMalloc(&D, sizeof(hD)); hipMalloc(&cyc, 16);
    hipMemcpy(A, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(B, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, D, cyc);
    long long hc[2];
    hipMemcpy(hD, D, sizeof(hD), hipMemcpyDeviceToHost); hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; i++) bad += hD[i] != ref[i];
    printf("v_mfma_i32_16x16x64_i8 layout check: %d of 256 entries differ from the host product\n", bad);
    printf("cycles per MFMA: accumulate-chained %.1f, four independent accumulators %.1f\n", hc[0] / 4000.0, hc[1] / 4000.0);
    return bad;
}
int main() { return ubench_main(); }
