// Microbenchmark: can a wave issue other vector instructions "in the shadow" of a dependent chain of
// v_mfma_f64_16x16x4_f64 (the eigen-solver's squarings: 64 cycles each, one wave per SIMD)?
// Per iteration: 12 accumulate-chained / operand-chained MFMAs, and between consecutive MFMAs FILL independent instructions of one kind:
//   mode 0 nothing, 1 v_accvgpr_read_b32, 2 v_and_b32, 3 v_cvt_f64_u32, 4 v_fma_f64 (independent accumulators), 5 v_mul_f64
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shadow mfma_shadow.hip && ./mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define FILL 12

template <int MODE>
__device__ __forceinline__ void fill(unsigned (&r)[FILL], double (&f)[FILL / 2], double c)
{
#pragma unroll
    for (int i = 0; i < FILL; i++) {
        if (MODE == 1) asm volatile("v_accvgpr_read_b32 %0, a100" : "=v"(r[i]));
        if (MODE == 2) asm volatile("v_and_b32 %0, 0xffff, %0" : "+v"(r[i]));
        if (MODE == 3) { double t; asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(t) : "v"(r[i])); if (i < FILL / 2) f[i] += 0.0 * t; }
    }
#pragma unroll
    for (int i = 0; i < FILL / 2; i++) {
        if (MODE == 4) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(f[i]) : "v"(c));
        if (MODE == 5) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[i]) : "v"(c));
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k(long long *cyc, double *sink, int iters)
{
    asm volatile("" ::: "a127");
    const int l = threadIdx.x & 63;
    double s0 = 1.0 + 1e-3 * l, s1 = 0.5, s2 = 0.25;
    unsigned r[FILL];
    double f[FILL / 2];
    for (int i = 0; i < FILL; i++) r[i] = l + i;
    for (int i = 0; i < FILL / 2; i++) f[i] = 1.0 + i;
    const double c = 1.0000001;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int sq = 0; sq < 4; sq++) {                                  // 4 x (3 accumulate-chained MFMAs whose result feeds the next): 12 per iteration
            d4 z = {0, 0, 0, 0};
            z = __builtin_amdgcn_mfma_f64_16x16x4f64(s0, s0, z, 0, 0, 0); fill<MODE>(r, f, c);
            z = __builtin_amdgcn_mfma_f64_16x16x4f64(s1, s1, z, 0, 0, 0); fill<MODE>(r, f, c);
            z = __builtin_amdgcn_mfma_f64_16x16x4f64(s2, s2, z, 0, 0, 0); fill<MODE>(r, f, c);
            s0 = z[0] * 1e-3; s1 = z[1] * 1e-3; s2 = z[2] * 1e-3;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    double acc = s0 + s1 + s2;
    for (int i = 0; i < FILL; i++) acc += r[i];
    for (int i = 0; i < FILL / 2; i++) acc += f[i];
    if (acc == 1.2345) sink[0] = acc;
}

template <int MODE> static void run(const char *name, long long *cyc, double *sink)
{
    const int iters = 2000;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, cyc, sink, iters); hipDeviceSynchronize(); }
    long long h[4];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-28s %7.1f cycles per iteration of 12 dependent MFMAs + 12 x %d fill instructions (%.1f per MFMA slot)\n", name, (double) h[0] / iters,
           MODE == 0 ? 0 : (MODE >= 4 ? FILL / 2 : FILL), (double) h[0] / iters / 12.0);
}

extern "C" int ubench_main()
{
    long long *cyc; double *sink;
    hipMalloc(&cyc, sizeof(long long) * 4096); hipMalloc(&sink, 64);
    run<0>("no fill", cyc, sink);
    run<1>("12 v_accvgpr_read_b32", cyc, sink);
    run<2>("12 v_and_b32", cyc, sink);
    run<3>("12 v_cvt_f64_u32", cyc, sink);
    run<4>("6 v_fma_f64 (independent)", cyc, sink);
    run<5>("6 v_mul_f64 (independent)", cyc, sink);
    return 0;
}
int main() { return ubench_main(); }
