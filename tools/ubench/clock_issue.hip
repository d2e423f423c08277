// Microbenchmark: fp64 VALU issue ceiling WITH the clock stated.  Every wave stamps s_memtime (shader cycles) and
// s_memrealtime (100 MHz) around its loop, so that the in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz and the
// per-SIMD cost of one wave-instruction = (slowest wave's cycles) / (instructions per wave x waves per SIMD) are both
// measured, for 1, 2 and 4 waves per SIMD.  Modes: fp64 FMA with one VGPR source (the update's form, u in SGPRs), with
// three VGPR sources (the Gram update), and the instruction mix of one column of the p = 10 pass (10 cvt + 10 mul +
// 10 + 10 + 10 fma + 10 max + 55 Gram fma).
//   hipcc --offload-arch=gfx950 -O3 -o clock_issue clock_issue.hip && ./clock_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int MODE>
__global__ void k(double *sink, long long *stamps, int iters)
{
    double a[16], b[16];
    for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 1e-3 + i; b[i] = 1.0 + 1e-9 * (threadIdx.x + i); }
    double G[55];
    for (int i = 0; i < 55; i++) G[i] = 0.0;
    float x[10];
    for (int i = 0; i < 10; i++) x[i] = (float) (threadIdx.x + i);
    const double m = 1.0000001;
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = fma(a[i], m, 0.5);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = fma(b[i], b[(i + 5) & 15], a[i]);
        } else {
            // one column of the pass, registers only
            double f[10], s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int i = 0; i < 10; i++) f[i] = (double) x[i] * (1.0 + 0.01 * i);
#pragma unroll
            for (int i = 0; i < 10; i += 2) { s0 = fma(0.3 + 0.01 * i, a[i], s0); s1 = fma(0.31 + 0.01 * i, a[i + 1], s1); }
            const double s = s0 + s1;
#pragma unroll
            for (int i = 0; i < 10; i++) {
                const double res = fma(0.3 + 0.01 * i, s, -f[i]);
                a[i] = fmax(fma(-0.1, res, a[i]), f[i]);
            }
#pragma unroll
            for (int i = 0; i < 10; i++)
#pragma unroll
                for (int j = 0; j <= i; j++) G[i * (i + 1) / 2 + j] = fma(a[i], a[j], G[i * (i + 1) / 2 + j]);
#pragma unroll
            for (int i = 0; i < 10; i++) x[i] += 1.0f;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    double sum = 0;
    for (int i = 0; i < 16; i++) sum += a[i] + b[i];
    for (int i = 0; i < 55; i++) sum += G[i];
    for (int i = 0; i < 10; i++) sum += x[i];
    if (sum == 12345.678) sink[0] = sum;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

int main()
{
    double *d; hipMalloc(&d, 64);
    long long *st; hipMalloc(&st, sizeof(long long) * 2 * 256 * 16);
    const char *names[] = {"v_fma_f64, 1 VGPR source", "v_fma_f64, 3 VGPR sources", "column of the p=10 pass (125 instr)"};
    const double per_iter[] = {16.0, 16.0, 125.0};
    for (int mode = 0; mode < 3; mode++)
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int threads = 256 * wps;
            const int iters = mode == 2 ? 40000 : 300000;          // a few ms per launch; three launches, the last is reported
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, d, st, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, d, st, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, d, st, iters);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            const int nw = 256 * threads / 64;
            std::vector<long long> h(2 * nw);
            hipMemcpy(h.data(), st, sizeof(long long) * 2 * nw, hipMemcpyDeviceToHost);
            std::vector<double> cyc(nw), clk(nw);
            for (int w = 0; w < nw; w++) { cyc[w] = (double) h[2 * w]; clk[w] = (double) h[2 * w] / (double) h[2 * w + 1] * 0.1; }
            std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
            const double instr = per_iter[mode] * iters;
            printf("%-36s waves/SIMD=%d  wall %.3f ms  in-kernel clock median %.3f GHz  cycles per wave-instr: fastest wave %.2f, median %.2f, slowest %.2f"
                   "  => per SIMD %.2f cycles (slowest wave / (instr x waves)), %.2f ns wall per wave-instr per SIMD\n",
                   names[mode], wps, ms, clk[nw / 2], cyc[0] / instr, cyc[nw / 2] / instr, cyc[nw - 1] / instr,
                   cyc[nw - 1] / instr / wps, ms * 1e6 / (instr * wps));
        }
    return 0;
}
