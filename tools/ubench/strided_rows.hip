// What the memory system gives the access patterns of the wide-cohort initial pass (gen::k_ratio_svd_mg, config 4): a gene is
// R = 50 rows of L = 2 750 fp32 counts (rows 11 KB apart), 512 resident workgroups of 4 waves each walk one gene at a time, a wave
// takes blocks of columns round-robin and reads the block's segment of EVERY row before it goes on.  Nothing is computed (a running
// sum keeps the loads alive), so every number is the pattern's own ceiling:
//   A  pass 2 as shipped: one dword per lane -- 256 contiguous bytes per row and instruction, 50 instructions in flight
//   B  8 bytes per lane   -- 512 B per row and instruction
//   C  16 bytes per lane  -- 1 KB per row and instruction
//   D  pass 1 as shipped: lane (i, kb) reads 16 bytes, an instruction covers 16 rows x 64 contiguous bytes, four instructions per row tile
//   E  the gene as one contiguous run, 16 bytes per lane (what k_row_max does): 1 KB per instruction, 8 in flight
// each with default and with non-temporal loads.   usage: bash tools/exp_ub.sh <tag> strided_rows.so
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
constexpr int R = 50, L = 2750, W = 4;

template <bool NT, typename T> __device__ __forceinline__ T ld(const T *p) { return NT ? __builtin_nontemporal_load(p) : *p; }

template <int VW, bool NT>              // VW floats per lane and row
__global__ __launch_bounds__(256, 2) void k_cols(const float *base, int n_genes, float *out)
{
    typedef float vt __attribute__((ext_vector_type(VW), aligned(4)));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float s = 0.0f;
    for (int g = blockIdx.x; g < n_genes; g += gridDim.x) {
        const float *x = base + (size_t) g * R * L;
        constexpr int BC = 64 * VW;                                      // columns per wave and trip
        for (int k0 = BC * w; k0 + BC <= L; k0 += BC * W) {
            vt v[R];
#pragma unroll
            for (int i = 0; i < R; i++) v[i] = ld<NT>((const vt *) (x + (size_t) i * L + k0 + VW * lane));
#pragma unroll
            for (int i = 0; i < R; i++) {
#pragma unroll
                for (int e = 0; e < VW; e++) s += v[i][e];
            }
        }
    }
    if (s == 1.2345f) out[0] = s;
}

template <bool NT, bool ROWSEG>
__global__ __launch_bounds__(256, 2) void k_tiles(const float *base, int n_genes, float *out)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    float s = 0.0f;
    for (int g = blockIdx.x; g < n_genes; g += gridDim.x) {
        const float *x = base + (size_t) g * R * L;
        for (int k0 = 64 * w; k0 + 64 <= L; k0 += 64 * W) {
            f4 v[16];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int row = 16 * t + li < R ? 16 * t + li : 0;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    v[4 * t + q] = ld<NT>((const f4 *) (x + (size_t) row * L + k0 + (ROWSEG ? 16 * q + 4 * lk : 16 * lk + 4 * q)));
            }
#pragma unroll
            for (int i = 0; i < 16; i++) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    if (s == 1.2345f) out[0] = s;
}

template <bool NT>
__global__ __launch_bounds__(256, 2) void k_flat(const float *base, int n_genes, float *out)
{
    float s = 0.0f;
    for (int g = blockIdx.x; g < n_genes; g += gridDim.x) {
        const f4 *x = (const f4 *) (base + (size_t) g * R * L);
        const int n4 = R * L / 4;
        for (int i = threadIdx.x; i + 7 * 256 < n4; i += 8 * 256) {
            f4 v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = ld<NT>(x + i + 256 * q);
#pragma unroll
            for (int q = 0; q < 8; q++) s += v[q][0] + v[q][1] + v[q][2] + v[q][3];
        }
    }
    if (s == 1.2345f) out[0] = s;
}

// both passes of the initial pass back to back per gene (tiles as pass 1, then columns as pass 2, from the gene's end or from its
// start), on rows of LL floats: LL = 2 750 is the fp32 gene, LL = 1 375 the same gene with 2-byte counts (half the bytes in flight
// between a byte's two uses: 140 MB instead of 280 MB for the 512 resident workgroups, against the 256 MB Infinity Cache)
template <int LL, bool REV, bool NT2>
__global__ __launch_bounds__(256, 2) void k_two(const float *base, int n_genes, float *out)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    float s = 0.0f;
    for (int g = blockIdx.x; g < n_genes; g += gridDim.x) {
        const float *x = base + (size_t) g * R * LL;
        for (int k0 = 64 * w; k0 + 64 <= LL; k0 += 64 * W) {
            f4 v[16];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int row = 16 * t + li < R ? 16 * t + li : 0;
#pragma unroll
                for (int q = 0; q < 4; q++) v[4 * t + q] = *(const f4 *) (x + (size_t) row * LL + k0 + 16 * q + 4 * lk);
            }
#pragma unroll
            for (int i = 0; i < 16; i++) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
        __syncthreads();
        constexpr int NB = LL / 64;
        for (int b = w; b < NB; b += W) {
            const int k0 = 64 * (REV ? NB - 1 - b : b);
            float v[R];
#pragma unroll
            for (int i = 0; i < R; i++) v[i] = ld<NT2>(x + (size_t) i * LL + k0 + lane);
#pragma unroll
            for (int i = 0; i < R; i++) s += v[i];
        }
        __syncthreads();
    }
    if (s == 1.2345f) out[0] = s;
}

// the same two passes WITH the arithmetic of the real kernel between the loads (pass 1: counts -> offset bytes -> 40 i8 MFMAs per
// 64-column group; pass 2: dot product with u from LDS, clamped sums into 56 per-lane accumulators), without its fixed parts (tile
// sums, solve, reductions): what the load -> wait -> compute shape of a wave costs against the pure patterns above
typedef int i4 __attribute__((ext_vector_type(4)));
template <int LL, bool REV, bool NT2, bool WORK1, bool WORK2>
__global__ __launch_bounds__(256, 2) void k_two_work(const float *base, int n_genes, float *out)
{
    __shared__ double u_lds[64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    if (threadIdx.x < 64) u_lds[threadIdx.x] = 0.125 + 0.001 * threadIdx.x;
    __syncthreads();
    float s = 0.0f;
    double sd = 0.0;
    for (int g = blockIdx.x; g < n_genes; g += gridDim.x) {
        const float *x = base + (size_t) g * R * LL;
        i4 hh[10], ll[10], hl[10];
#pragma unroll
        for (int i = 0; i < 10; i++) { hh[i] = i4{0, 0, 0, 0}; ll[i] = hh[i]; hl[i] = hh[i]; }
#pragma clang loop unroll(disable)
        for (int k0 = 64 * w; k0 + 64 <= LL; k0 += 64 * W) {
            i4 H[4], Lo[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int row = 16 * t + li < R ? 16 * t + li : 0;
                f4 v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = *(const f4 *) (x + (size_t) row * LL + k0 + 16 * q + 4 * lk);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if constexpr (WORK1) {
                        const unsigned u0 = (unsigned) v[q][0], u1 = (unsigned) v[q][1], u2 = (unsigned) v[q][2], u3 = (unsigned) v[q][3];
                        const unsigned t01 = __builtin_amdgcn_perm(u1, u0, 0x05010400u), t23 = __builtin_amdgcn_perm(u3, u2, 0x05010400u);
                        Lo[t][q] = (int) (__builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u);
                        H[t][q] = (int) (__builtin_amdgcn_perm(t23, t01, 0x07060302u) ^ 0x80808080u);
                    } else s += v[q][0] + v[q][1] + v[q][2] + v[q][3];
                }
            }
            if constexpr (WORK1) {
                int tix = 0;
#pragma unroll
                for (int t1 = 0; t1 < 4; t1++)
#pragma unroll
                    for (int t2 = 0; t2 <= t1; t2++, tix++) {
                        hh[tix] = __builtin_amdgcn_mfma_i32_16x16x64_i8(H[t1], H[t2], hh[tix], 0, 0, 0);
                        ll[tix] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Lo[t1], Lo[t2], ll[tix], 0, 0, 0);
                        hl[tix] = __builtin_amdgcn_mfma_i32_16x16x64_i8(H[t1], Lo[t2], hl[tix], 0, 0, 0);
                        hl[tix] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Lo[t1], H[t2], hl[tix], 0, 0, 0);
                    }
            }
        }
        if constexpr (WORK1) {
#pragma unroll
            for (int i = 0; i < 10; i++) s += (float) (hh[i][0] + ll[i][1] + hl[i][2] + hh[i][3]);
        }
        __syncthreads();
        constexpr int NB = LL / 64, RC = 56;
        double acc[RC];
#pragma unroll
        for (int i = 0; i < RC; i++) acc[i] = 0.0;
#pragma clang loop unroll(disable)
        for (int b = w; b < NB; b += W) {
            const int k0 = 64 * (REV ? NB - 1 - b : b);
            float v[RC];
#pragma unroll
            for (int i = 0; i < RC; i++) v[i] = ld<NT2>(x + (size_t) (i < R ? i : R - 1) * LL + k0 + lane);
            if constexpr (WORK2) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int i = 0; i < RC; i += 2) { s0 = fma(u_lds[i], (double) v[i], s0); s1 = fma(u_lds[i + 1], (double) v[i + 1], s1); }
                const double sj = s0 + s1;
#pragma unroll
                for (int i = 0; i < RC; i++) {
                    asm volatile("" : "+v"(v[i]));
                    acc[i] += fmax(u_lds[i] * sj, (double) v[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < RC; i++) s += v[i];
            }
        }
        if constexpr (WORK2) {
#pragma unroll
            for (int i = 0; i < RC; i++) sd += acc[i];
        }
        __syncthreads();
    }
    if (s == 1.2345f || sd == 1.2345) out[0] = s + (float) sd;
}

// ... and on genes of DIFFERENT lengths (uniform on 501 .. 5 000 like config 4, or all 2 750), fetched from a queue (longest
// first) like the real kernel, rows at whatever alignment the length gives or padded to 16 bytes
__global__ __launch_bounds__(256, 2) void k_two_var(const float *base, const long long *goff, const int *glen, const int *gpitch,
                                                    int n_genes, int *counter, float *out)
{
    __shared__ double u_lds[64];
    __shared__ int q_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    if (threadIdx.x < 64) u_lds[threadIdx.x] = 0.125 + 0.001 * threadIdx.x;
    __syncthreads();
    float s = 0.0f;
    double sd = 0.0;
    for (;;) {
        if (threadIdx.x == 0) q_s = atomicAdd(counter, 1);
        __syncthreads();
        const int g = q_s;
        __syncthreads();
        if (g >= n_genes) break;
        const float *x = base + goff[g];
        const int LL = glen[g], P = gpitch[g];
        i4 hh[10], ll[10], hl[10];
#pragma unroll
        for (int i = 0; i < 10; i++) { hh[i] = i4{0, 0, 0, 0}; ll[i] = hh[i]; hl[i] = hh[i]; }
#pragma clang loop unroll(disable)
        for (int k0 = 64 * w; k0 + 64 <= LL; k0 += 64 * W) {
            i4 H[4], Lo[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int row = 16 * t + li < R ? 16 * t + li : 0;
                f4 v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = *(const f4 *) (x + (size_t) row * P + k0 + 16 * q + 4 * lk);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const unsigned u0 = (unsigned) v[q][0], u1 = (unsigned) v[q][1], u2 = (unsigned) v[q][2], u3 = (unsigned) v[q][3];
                    const unsigned t01 = __builtin_amdgcn_perm(u1, u0, 0x05010400u), t23 = __builtin_amdgcn_perm(u3, u2, 0x05010400u);
                    Lo[t][q] = (int) (__builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u);
                    H[t][q] = (int) (__builtin_amdgcn_perm(t23, t01, 0x07060302u) ^ 0x80808080u);
                }
            }
            int tix = 0;
#pragma unroll
            for (int t1 = 0; t1 < 4; t1++)
#pragma unroll
                for (int t2 = 0; t2 <= t1; t2++, tix++) {
                    hh[tix] = __builtin_amdgcn_mfma_i32_16x16x64_i8(H[t1], H[t2], hh[tix], 0, 0, 0);
                    ll[tix] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Lo[t1], Lo[t2], ll[tix], 0, 0, 0);
                    hl[tix] = __builtin_amdgcn_mfma_i32_16x16x64_i8(H[t1], Lo[t2], hl[tix], 0, 0, 0);
                    hl[tix] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Lo[t1], H[t2], hl[tix], 0, 0, 0);
                }
        }
#pragma unroll
        for (int i = 0; i < 10; i++) s += (float) (hh[i][0] + ll[i][1] + hl[i][2] + hh[i][3]);
        __syncthreads();
        constexpr int RC = 56;
        const int NB = LL / 64;
        double acc[RC];
#pragma unroll
        for (int i = 0; i < RC; i++) acc[i] = 0.0;
#pragma clang loop unroll(disable)
        for (int b = w; b < NB; b += W) {
            const int k0 = 64 * (NB - 1 - b);
            float v[RC];
#pragma unroll
            for (int i = 0; i < RC; i++) v[i] = ld<true>(x + (size_t) (i < R ? i : R - 1) * P + k0 + lane);
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int i = 0; i < RC; i += 2) { s0 = fma(u_lds[i], (double) v[i], s0); s1 = fma(u_lds[i + 1], (double) v[i + 1], s1); }
            const double sj = s0 + s1;
#pragma unroll
            for (int i = 0; i < RC; i++) {
                asm volatile("" : "+v"(v[i]));
                acc[i] += fmax(u_lds[i] * sj, (double) v[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < RC; i++) sd += acc[i];
        __syncthreads();
    }
    if (s == 1.2345f || sd == 1.2345) out[0] = s + (float) sd;
}

template <typename K> static void run(const char *name, K launch, double bytes)
{
    hipEvent_t e0, e1;
    (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        (void) hipEventRecord(e0, 0);
        launch();
        (void) hipEventRecord(e1, 0);
        (void) hipEventSynchronize(e1);
        float ms; (void) hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    printf("%-58s %7.3f ms  %6.2f TB/s\n", name, best, bytes / best * 1e-9);
    fflush(stdout);
}

extern "C" int ubench_main()
{
    const int n_genes = 16000;                                          // 8.8 GB, as the 16 000-gene slice of tools/c4_ab.py
    const size_t bytes = (size_t) n_genes * R * L * 4;
    const size_t alloc = (size_t) n_genes * R * 2900 * 4 + 4096;           // room for the longest rows and the variable-length tables below
    float *buf, *out;
    if (hipMalloc(&buf, alloc) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void) hipMemset(buf, 0, alloc);
    const dim3 grid(512), blk(256);
    const double full = (double) bytes;
    // the column patterns leave the tail of a row that does not fill a whole trip unread: count what is read
    auto frac = [](int bc) { return (double) (L / bc * bc) / L; };
    const double flat = (double) ((R * L / 4 - 1793) / 2048 + 1) * 2048 / (R * L / 4);
#define COLS(VW, NT, name) run(name, [&] { hipLaunchKernelGGL((k_cols<VW, NT>), grid, blk, 0, 0, buf, n_genes, out); }, full * frac(64 * VW))
    COLS(1, false, "A  4 B per lane: 256 B per row and instruction");
    COLS(1, true,  "A  ... non-temporal");
    COLS(2, false, "B  8 B per lane: 512 B per row and instruction");
    COLS(2, true,  "B  ... non-temporal");
    COLS(4, false, "C  16 B per lane: 1 KB per row and instruction");
    COLS(4, true,  "C  ... non-temporal");
    run("D  tiles, 64 B per LANE (round 3's pass 1)", [&] { hipLaunchKernelGGL((k_tiles<false, false>), grid, blk, 0, 0, buf, n_genes, out); }, full * frac(64));
    run("D  tiles, 64 B per row and instruction (round 4)", [&] { hipLaunchKernelGGL((k_tiles<false, true>), grid, blk, 0, 0, buf, n_genes, out); }, full * frac(64));
    run("D  ... non-temporal", [&] { hipLaunchKernelGGL((k_tiles<true, true>), grid, blk, 0, 0, buf, n_genes, out); }, full * frac(64));
    run("E  the gene as one contiguous run, 16 B per lane", [&] { hipLaunchKernelGGL((k_flat<false>), grid, blk, 0, 0, buf, n_genes, out); }, full * flat);
    run("E  ... non-temporal", [&] { hipLaunchKernelGGL((k_flat<true>), grid, blk, 0, 0, buf, n_genes, out); }, full * flat);
#define TWO(LL, REV, NT2, name) run(name, [&] { hipLaunchKernelGGL((k_two<LL, REV, NT2>), grid, blk, 0, 0, buf, n_genes, out); }, 2.0 * n_genes * R * (LL / 64 * 64) * 4.0)
    TWO(2750, false, false, "two passes per gene, fp32 rows, pass 2 forwards");
    TWO(2750, true, false,  "two passes per gene, fp32 rows, pass 2 from the end");
    TWO(2750, true, true,   "two passes per gene, fp32 rows, from the end, nt");
    TWO(1375, false, false, "two passes per gene, 2-byte rows, pass 2 forwards");
    TWO(1375, true, false,  "two passes per gene, 2-byte rows, pass 2 from the end");
    TWO(1375, true, true,   "two passes per gene, 2-byte rows, from the end, nt");
#define TWOW(LL, W1, W2, name) run(name, [&] { hipLaunchKernelGGL((k_two_work<LL, true, true, W1, W2>), grid, blk, 0, 0, buf, n_genes, out); }, 2.0 * n_genes * R * (LL / 64 * 64) * 4.0)
    TWOW(2750, false, false, "two passes, from the end, nt, 56 rows in pass 2, no arithmetic");
    TWOW(2750, true, false,  "  + pass 1's conversions and MFMAs");
    TWOW(2750, false, true,  "  + pass 2's dot product and clamped sums");
    TWOW(2750, true, true,   "  + both");
    TWOW(2752, true, true,   "  + both, rows of 2 752 counts (every row 16-byte aligned)");
    TWOW(2751, true, true,   "  + both, rows of 2 751 counts (rows at every 4-byte alignment)");
    TWOW(2753, true, true,   "  + both, rows of 2 753 counts");
    {   // variable lengths
        long long *goff_h = (long long *) malloc(sizeof(long long) * n_genes), *goff_d;
        int *glen_h = (int *) malloc(sizeof(int) * n_genes), *gp_h = (int *) malloc(sizeof(int) * n_genes), *glen_d, *gp_d, *cnt_d;
        (void) hipMalloc(&goff_d, sizeof(long long) * n_genes); (void) hipMalloc(&glen_d, sizeof(int) * n_genes);
        (void) hipMalloc(&gp_d, sizeof(int) * n_genes); (void) hipMalloc(&cnt_d, 4);
        for (int mode = 0; mode < 5; mode++) {              // 0: all 2 750, pitch = length; 1: uniform lengths, longest first, laid out in that order; 2: ... pitch padded to 4 counts; 3: unsorted; 4: laid out unsorted, WALKED longest first (the real kernel's case)
            unsigned long long st = 88172645463325252ull;
            double cnt = 0.0;
            for (int g = 0; g < n_genes; g++) {
                st ^= st << 13; st ^= st >> 7; st ^= st << 17;
                glen_h[g] = mode == 0 ? 2750 : 501 + (int) (st % 4500);
            }
            auto sort_desc = [&](bool with_off) {
                for (int i = 1; i < n_genes; i++) {
                    const int v = glen_h[i], pv = gp_h[i]; const long long o = goff_h[i]; int j = i - 1;
                    while (j >= 0 && glen_h[j] < v) { glen_h[j + 1] = glen_h[j]; if (with_off) { goff_h[j + 1] = goff_h[j]; gp_h[j + 1] = gp_h[j]; } j--; }
                    glen_h[j + 1] = v; if (with_off) { goff_h[j + 1] = o; gp_h[j + 1] = pv; }
                }
            };
            if (mode == 1 || mode == 2) sort_desc(false);
            long long off = 0;
            for (int g = 0; g < n_genes; g++) {
                gp_h[g] = mode == 2 ? (glen_h[g] + 3) & ~3 : glen_h[g];
                goff_h[g] = off; off += (long long) R * gp_h[g];
                if (mode == 2) off = (off + 3) & ~3ll;
                cnt += 2.0 * R * (glen_h[g] / 64 * 64) * 4.0;
            }
            if (mode == 4) sort_desc(true);
            if ((size_t) off * 4 + 4096 > alloc) { printf("variable-length table does not fit\n"); break; }
            (void) hipMemcpy(goff_d, goff_h, sizeof(long long) * n_genes, hipMemcpyHostToDevice);
            (void) hipMemcpy(glen_d, glen_h, sizeof(int) * n_genes, hipMemcpyHostToDevice);
            (void) hipMemcpy(gp_d, gp_h, sizeof(int) * n_genes, hipMemcpyHostToDevice);
            const char *names[5] = { "queue, every gene 2 750 counts long", "queue, lengths uniform on 501 .. 5 000, longest first",
                                     "  ... rows padded to 16 bytes", "  ... rows as they come, genes in random order",
                                     "  ... laid out in random order, walked longest first" };
            run(names[mode], [&] { (void) hipMemsetAsync(cnt_d, 0, 4, 0); hipLaunchKernelGGL(k_two_var, grid, blk, 0, 0, buf, goff_d, glen_d, gp_d, n_genes, cnt_d, out); }, cnt);
        }
    }
    (void) hipFree(buf); (void) hipFree(out);
    return 0;
}
