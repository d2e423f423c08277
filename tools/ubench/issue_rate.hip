// Microbenchmark: VALU issue rate of one wave alone on a SIMD vs two waves, for fp64 FMA, fp32 FMA, packed fp32 FMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void k(double *out, int iters)
{
    double a[16], b[16]; float f[16], g[16]; float2 p2[16];
    for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 1e-3 + i; f[i] = (float) a[i]; p2[i] = make_float2(f[i], f[i] + 1); b[i] = 1.0 + 1e-9 * (threadIdx.x + i); g[i] = (float) b[i]; }
    double m = 1.0000001; float mf = 1.0000001f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) a[i] = fma(a[i], m, 0.5);
            if (MODE == 1) f[i] = fmaf(f[i], mf, 0.5f);
            if (MODE == 2) { p2[i].x = fmaf(p2[i].x, mf, 0.5f); p2[i].y = fmaf(p2[i].y, mf, 0.5f); }
            if (MODE == 3) a[i] = fma(b[i], b[(i + 5) & 15], a[i]);      // three VGPR operands, like the Gram update
            if (MODE == 4) f[i] = fmaf(g[i], g[(i + 5) & 15], f[i]);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 16; i++) s += a[i] + f[i] + p2[i].x + p2[i].y + b[i] + g[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (double) (t1 - t0) / (16.0 * iters); }
    if (s == 12345.678) out[1] = s;
}
int main()
{
    double *d; hipMalloc(&d, 64);
    const char *names[] = {"v_fma_f64 1 vgpr src", "v_fma_f32 1 vgpr src", "2x v_fma_f32 (pk?)", "v_fma_f64 3 vgpr src", "v_fma_f32 3 vgpr src"};
    for (int mode = 0; mode < 5; mode++)
        for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
            int threads = 256 * waves_per_simd;   // one block per CU, waves_per_simd waves on each SIMD
            if (threads > 1024) continue;
            hipMemset(d, 0, 64);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, d, 20000);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, d, 20000);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, d, 20000);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, d, 20000);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 0, 0, d, 20000);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            double h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            double wave_instr = 256.0 * (threads / 64) * 16.0 * 20000;   // per chip
            printf("%-20s waves/SIMD=%d  s_memtime ticks/instr = %.2f  wall %.3f ms  => %.2f ns per wave-instr per SIMD\n", names[mode], waves_per_simd, h[0], ms, ms * 1e6 / (wave_instr / 1024.0));
        }
    return 0;
}
