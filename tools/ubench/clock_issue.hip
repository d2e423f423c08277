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

// MODE >= 4 keeps a column's state in a200 .. a219 by inline asm, behind the compiler's back: its own AGPR use (spill
// slots, allocated from a0 upwards) must stay below a200 -- check with the scan of the generated ISA before running
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k(double *sink, long long *stamps, int iters, int nlim)
{
    double a[16], b[16];
    for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 1e-3 + i; b[i] = 1.0 + 1e-9 * (threadIdx.x + i); }
    double G[55];
    for (int i = 0; i < 55; i++) G[i] = 0.0;
    float x[10];
    for (int i = 0; i < 10; i++) x[i] = (float) (threadIdx.x + i);
    const double m = 1.0000001;
    if (MODE == 7) {                                   // put the CU's waves out of phase in the unrolled code: every wave its own fetch stream
        const int w = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % 8;
        for (int d = 0; d < w * 531; d++) __builtin_amdgcn_s_sleep(3);
    }
    const long long r0 = __builtin_amdgcn_s_memrealtime();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = fma(a[i], m, 0.5);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = fma(b[i], b[(i + 5) & 15], a[i]);
        } else if (MODE == 4 || MODE == 5 || MODE == 6 || MODE == 7 || MODE == 8) {
#pragma unroll
          for (int rep = 0; rep < (MODE >= 6 ? 10 : 1); rep++) {
            if (MODE == 8 && !((int) threadIdx.x + rep * 256 < nlim)) continue;       // the kernel's per-column exec mask (never false here)
            // the scaled column with its state in AGPRs: 20 v_accvgpr_read before, 20 v_accvgpr_write after (MODE 5: reads only)
            double f[10], s0 = 0.0, s1 = 0.0;
            int lo[10], hi[10];
#define RD(I) asm volatile("v_accvgpr_read_b32 %0, a" #I : "=v"(lo[I / 2])); asm volatile("v_accvgpr_read_b32 %0, a" #I "+1" : "=v"(hi[I / 2]));
            asm volatile("v_accvgpr_read_b32 %0, a200\n\tv_accvgpr_read_b32 %1, a201" : "=v"(lo[0]), "=v"(hi[0]));
            asm volatile("v_accvgpr_read_b32 %0, a202\n\tv_accvgpr_read_b32 %1, a203" : "=v"(lo[1]), "=v"(hi[1]));
            asm volatile("v_accvgpr_read_b32 %0, a204\n\tv_accvgpr_read_b32 %1, a205" : "=v"(lo[2]), "=v"(hi[2]));
            asm volatile("v_accvgpr_read_b32 %0, a206\n\tv_accvgpr_read_b32 %1, a207" : "=v"(lo[3]), "=v"(hi[3]));
            asm volatile("v_accvgpr_read_b32 %0, a208\n\tv_accvgpr_read_b32 %1, a209" : "=v"(lo[4]), "=v"(hi[4]));
            asm volatile("v_accvgpr_read_b32 %0, a210\n\tv_accvgpr_read_b32 %1, a211" : "=v"(lo[5]), "=v"(hi[5]));
            asm volatile("v_accvgpr_read_b32 %0, a212\n\tv_accvgpr_read_b32 %1, a213" : "=v"(lo[6]), "=v"(hi[6]));
            asm volatile("v_accvgpr_read_b32 %0, a214\n\tv_accvgpr_read_b32 %1, a215" : "=v"(lo[7]), "=v"(hi[7]));
            asm volatile("v_accvgpr_read_b32 %0, a216\n\tv_accvgpr_read_b32 %1, a217" : "=v"(lo[8]), "=v"(hi[8]));
            asm volatile("v_accvgpr_read_b32 %0, a218\n\tv_accvgpr_read_b32 %1, a219" : "=v"(lo[9]), "=v"(hi[9]));
#pragma unroll
            for (int i = 0; i < 10; i++) a[i] = __hiloint2double(hi[i], lo[i]);
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
            if (MODE == 4 || MODE >= 6) {
#define WR(A0, A1, I) asm volatile("v_accvgpr_write_b32 a" #A0 ", %0\n\tv_accvgpr_write_b32 a" #A1 ", %1" : : "v"(__double2loint(a[I])), "v"(__double2hiint(a[I])));
                WR(200, 201, 0) WR(202, 203, 1) WR(204, 205, 2) WR(206, 207, 3) WR(208, 209, 4) WR(210, 211, 5) WR(212, 213, 6) WR(214, 215, 7) WR(216, 217, 8) WR(218, 219, 9)
            }
#pragma unroll
            for (int i = 0; i < 10; i++) x[i] += 1.0f;
          }
        } else if (MODE == 3) {
            // the same column in raw count units (col_update_raw): 4 instead of 5 fp64 instructions per element
            double f[10], s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int i = 0; i < 10; i++) f[i] = (double) x[i];
#pragma unroll
            for (int i = 0; i < 10; i += 2) { s0 = fma(0.3 + 0.01 * i, a[i], s0); s1 = fma(0.31 + 0.01 * i, a[i + 1], s1); }
            const double s = s0 + s1;
#pragma unroll
            for (int i = 0; i < 10; i++) a[i] = fmax(fma(-(0.03 + 0.001 * i), s, fma(0.1, f[i], a[i])), f[i]);
#pragma unroll
            for (int i = 0; i < 10; i++)
#pragma unroll
                for (int j = 0; j <= i; j++) G[i * (i + 1) / 2 + j] = fma(a[i], a[j], G[i * (i + 1) / 2 + j]);
#pragma unroll
            for (int i = 0; i < 10; i++) x[i] += 1.0f;
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
    if (MODE >= 4) asm volatile("" ::: "a219");      // a200 .. a219 are ours (the allocator stays far below: checked on the ISA)
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
    const char *names[] = {"v_fma_f64, 1 VGPR source", "v_fma_f64, 3 VGPR sources", "column of the p=10 pass (125 instr)", "column, raw units (115 instr)", "column + 20 accvgpr reads + 20 writes (165)", "column + 20 accvgpr reads (145)", "10 unrolled copies of column + reads + writes (1650)", "the same, waves out of phase", "10 unrolled copies, each under the per-column exec mask"};
    const double per_iter[] = {16.0, 16.0, 125.0, 115.0, 165.0, 145.0, 1650.0, 1650.0, 1650.0};
    for (int mode = 0; mode < 9; mode++)
        for (int wps = 1; wps <= 4; wps *= 2) {
            if (mode >= 2 && wps > 2) continue;            // the column body needs > 128 registers
            if (mode >= 4 && wps > 1) continue;            // a200 .. a219 exist only at one wave per SIMD
            const int threads = 256 * wps;
            const int iters = mode >= 6 ? 4000 : (mode >= 2 ? 40000 : 300000);          // a few ms per launch; three launches, the last is reported
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0, 0);
#define DN_LAUNCH(M, T) hipLaunchKernelGGL((k<M, T>), dim3(256), dim3(T), 0, 0, d, st, iters, 1 << 30)
                if (mode == 0) { if (wps == 1) DN_LAUNCH(0, 256); else if (wps == 2) DN_LAUNCH(0, 512); else DN_LAUNCH(0, 1024); }
                if (mode == 1) { if (wps == 1) DN_LAUNCH(1, 256); else if (wps == 2) DN_LAUNCH(1, 512); else DN_LAUNCH(1, 1024); }
                if (mode == 2) { if (wps == 1) DN_LAUNCH(2, 256); else DN_LAUNCH(2, 512); }
                if (mode == 3) { if (wps == 1) DN_LAUNCH(3, 256); else DN_LAUNCH(3, 512); }
                if (mode == 4) { if (wps == 1) DN_LAUNCH(4, 256); else DN_LAUNCH(4, 512); }
                if (mode == 5) { if (wps == 1) DN_LAUNCH(5, 256); else DN_LAUNCH(5, 512); }
                if (mode == 6) { if (wps == 1) DN_LAUNCH(6, 256); else DN_LAUNCH(6, 512); }
                if (mode == 7) { if (wps == 1) DN_LAUNCH(7, 256); else DN_LAUNCH(7, 512); }
                if (mode == 8) { if (wps == 1) DN_LAUNCH(8, 256); else DN_LAUNCH(8, 512); }
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
