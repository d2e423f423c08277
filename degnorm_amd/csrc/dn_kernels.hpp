// dn_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) for DegNorm's NMF over-approximation hot path.
//
// What runs here (reference lines relative to the DegNorm checkout):
//   k_baseline<P,NT>   adjust_coverage_curves + baseline_selection per gene   nmf.py:142-146, :189-372
//                      (-> nmf() :78-107 -> rank_one_approx :55-64, get_high_coverage_idx :66-76,
//                       split_into_chunks utils.py:176-192, shift_bins nmf.py:160-187)
//   k_ratio_svd<P,NT>  ratio_svd + row sums                                    nmf.py:109-121, :524-525
//   k_estimates<P>     the `estimate` output of baseline_selection             nmf.py:327-365
//
// Mapping: one workgroup of NT threads (NT/64 wavefronts; NT = 64 is "one wavefront per gene") owns one
// gene at a time and walks the whole baseline-selection state machine for it; workgroups are persistent
// and pull genes from a longest-first queue.  Columns (base positions) are spread over lanes, the p
// samples of a column live in one lane's registers, so every global / LDS access is lane-contiguous.
//
// The rank-1 SVD: est = K E = u u^T (x + lambda) is column-local once u is known, so one pass per inner
// NMF-OA iteration updates lambda AND accumulates the p x p Gram matrix of the *next* x + lambda in
// registers (fp64).  The Gram partials are reduced through LDS (transposed tree, no atomics, fixed
// order => deterministic), and every lane then runs the same warm-started power iteration on the tiny
// matrix to machine precision (the reference's ARPACK call is tol=0).  No MFMA: p << 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dn {

constexpr int TRACE_LEN = 48;
constexpr int MAX_BINS = 64;
constexpr int P_MAX = 16;

enum { EXIT_LOW_COV = 0, EXIT_ZERO_SAMPLE = 1, EXIT_MEDIAN = 2, EXIT_NO_LOOP = 3,
       EXIT_REFINED = 4, EXIT_REFINE_FALLBACK = 5, EXIT_NOT_FOUND_FALLBACK = 6 };
enum { LOOP_NATURAL = 0, LOOP_PERFECT = 1, LOOP_VALUE_ERROR = 2, LOOP_ZERO_ROWSUM = 3, LOOP_MIN_BINS = 4,
       LOOP_NOT_ENTERED = 5 };
enum { ST_OK = 0, ST_ARPACK = -1, ST_EMPTY_MIN = -2, ST_VALUE_ERROR = -3 };
// how k_estimates rebuilds a gene's estimate
enum { EM_INPUT = 0,        // defaults: estimate = F                                   nmf.py:217
       EM_EXPAND = 1,       // n0 < L: K fixed-up, E = max_i F/K, est = max(K E, F)      nmf.py:358-365
       EM_REFINED = 2,      // n0 == L, baseline found: est = K E                        nmf.py:329-334
       EM_CLAMPED = 3,      // n0 == L, fallback: est = max(K_start E_start, F)          nmf.py:343-345, :350-352
       EM_RAW = 4 };        // n0 == L, no loop: est = K_start E_start                   nmf.py:251

struct IterArgs {
    const float   *cov;        // packed coverage, gene g at goff[g], p rows of glen[g]
    const int64_t *goff;
    const int32_t *glen;
    const int32_t *order;      // work queue: gene ids, longest first
    int32_t       *counter;    // queue head
    const int64_t *ds_start;   // per gene, or nullptr
    double        *ws;         // scratch: slots x slot_stride doubles
    double        *rho;        // n x p
    int32_t       *flags;      // n
    int32_t       *trace;      // n x TRACE_LEN
    double        *kfin;       // n x p   (estimates support)
    int32_t       *emode;      // n
    double        *svec;       // per gene s_start vectors (only when want_est), at svoff[g]
    const int64_t *svoff;
    int64_t        slot_stride;
    int32_t        n_genes;
    int32_t        S;          // column stride of the scratch arrays (>= longest gene, multiple of 64)
    int32_t        T, bins, min_hc, rate, skip, want_est;
    double         scale[P_MAX];
};

struct InitArgs {
    const float   *cov;
    const int64_t *goff;
    const int32_t *glen;
    const int32_t *order;
    int32_t       *counter;
    double        *est_sums;   // n x p
    double        *cov_sums;   // n x p
    int32_t       *status;     // n
    int32_t        n_genes;
};

struct EstArgs {
    const float   *cov;
    const int64_t *goff;
    const int32_t *glen;
    const double  *kfin;
    const int32_t *emode;
    const double  *svec;
    const int64_t *svoff;
    double        *out;        // same offsets as cov (goff), float64
    int32_t        n_genes;
    double         scale[P_MAX];
};

// ---------------------------------------------------------------------------------------------------
// LDS layout of one workgroup.
// ---------------------------------------------------------------------------------------------------
constexpr int RED_ROWS = 32;                 // entries reduced per round
constexpr int RED_LD = 66;                   // row stride in doubles: 528 B, 16-B aligned, bank-skewed by 4 dwords
constexpr int WAVE_RED_DOUBLES = RED_ROWS * RED_LD;

template <int P, int NT>
struct Smem {
    static constexpr int W = NT / 64;
    static constexpr int NG = P * (P + 1) / 2;
    double red[W][WAVE_RED_DOUBLES];         // per-wave transposed-reduce tile
    double xw[W][NG > 64 ? NG : 64];         // per-wave totals (cross-wave combine / broadcast)
    double ss[MAX_BINS];                     // per-bin mean squared residual
    int32_t alive[MAX_BINS];                 // original ids of the surviving bins, in order
    int32_t cnt[W];                          // per-wave hi-coverage counts
    int32_t gene;                            // current queue item
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

template <int NT>
__device__ __forceinline__ void block_sync() { __syncthreads(); }

// Sum each of the N register values over the 64 lanes of a wave and leave the totals in every lane.
// Transposed tree through LDS: lane l writes its N partials as column l of an [entry][lane] tile; lane t
// then adds 32 lanes of entry (t & 31) in a fixed order, the two halves meet with one shuffle, totals go
// back through LDS as broadcast reads.  ~3 LDS ops per value instead of 12 ds_bpermute for a butterfly.
template <int N>
__device__ __forceinline__ void wave_sum_bcast(double (&g)[N], double *tile, double *tot)
{
    constexpr int ROUNDS = (N + RED_ROWS - 1) / RED_ROWS;
    const int lane = lane_id();
    const int e = lane & 31, h = lane >> 5;
    double t[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
#pragma unroll
        for (int q = 0; q < RED_ROWS; q++) {
            if (r * RED_ROWS + q < N) tile[q * RED_LD + lane] = g[r * RED_ROWS + q];
        }
        __syncthreads();
        double s = 0.0;
        if (r * RED_ROWS + e < N) {
            const double2 *row = reinterpret_cast<const double2 *>(tile + e * RED_LD + h * 32);
#pragma unroll
            for (int c = 0; c < 16; c++) { double2 v = row[c]; s += v.x; s += v.y; }
        }
        s += __shfl_xor(s, 32);
        t[r] = s;
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; r++)
        if (lane < 32 && r * RED_ROWS + lane < N) tot[r * RED_ROWS + lane] = t[r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; i++) g[i] = tot[i];
}

// Block-wide sum of N values, result in every thread.  All waves add the per-wave totals in the same order,
// so every thread holds bit-identical results (the uniform branches below rely on that).
template <int N, int P, int NT>
__device__ __forceinline__ void block_sum(double (&g)[N], Smem<P, NT> &sm)
{
    static_assert(N <= (Smem<P, NT>::NG > 64 ? Smem<P, NT>::NG : 64), "xw too small");
    constexpr int W = NT / 64;
    const int w = wave_id();
    wave_sum_bcast<N>(g, sm.red[w], sm.xw[w]);
    if constexpr (W > 1) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < N; i++) {
            double s = sm.xw[0][i];
#pragma unroll
            for (int ww = 1; ww < W; ww++) s += sm.xw[ww][i];
            g[i] = s;
        }
    }
    __syncthreads();
}

template <int N, int P, int NT>
__device__ __forceinline__ void block_max_f(float (&m)[N], Smem<P, NT> &sm)
{
    constexpr int W = NT / 64;
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m[i] = fmaxf(m[i], __shfl_xor(m[i], o));
    }
    if constexpr (W > 1) {
        const int w = wave_id();
        if (lane_id() == 0) {
#pragma unroll
            for (int i = 0; i < N; i++) sm.xw[w][i] = (double) m[i];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < N; i++) {
            float s = (float) sm.xw[0][i];
#pragma unroll
            for (int ww = 1; ww < W; ww++) s = fmaxf(s, (float) sm.xw[ww][i]);
            m[i] = s;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ double wave_sum1(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------------------
// Top eigenpair of the symmetric PSD p x p matrix G (packed lower triangle, idx(i,j) = i(i+1)/2 + j).
// Power iteration, warm-started from u, run by every lane on identical data until ||u_new - u||^2 is at
// the fp64 rounding floor; the reference's ARPACK call (tol = 0) converges to machine precision too.
// Entries of x + lambda are non-negative, so the Perron vector is non-negative and a positive start is
// never orthogonal to it.  Returns the number of steps; theta = u^T G u (= sigma^2).
// ---------------------------------------------------------------------------------------------------
template <int P>
__device__ __forceinline__ int top_eig(const double (&G)[P * (P + 1) / 2], double (&u)[P], double &theta)
{
    int steps = 0;
    double th = 0.0;
    for (; steps < 4000;) {
        double y[P];
#pragma unroll
        for (int i = 0; i < P; i++) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < P; j++) {
                const int a = i > j ? i : j, b = i > j ? j : i;
                acc = fma(G[a * (a + 1) / 2 + b], u[j], acc);
            }
            y[i] = acc;
        }
        double n2 = 0.0;
        th = 0.0;
#pragma unroll
        for (int i = 0; i < P; i++) { n2 = fma(y[i], y[i], n2); th = fma(u[i], y[i], th); }
        steps++;
        if (!(n2 > 0.0)) { theta = 0.0; return steps; }
        const double inv = 1.0 / sqrt(n2);
        double d2 = 0.0;
#pragma unroll
        for (int i = 0; i < P; i++) {
            const double un = y[i] * inv;
            const double d = un - u[i];
            d2 = fma(d, d, d2);
            u[i] = un;
        }
        if (d2 <= 1e-27) break;
    }
    // Rayleigh quotient of the final vector
    {
        double t2 = 0.0;
#pragma unroll
        for (int i = 0; i < P; i++) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < P; j++) {
                const int a = i > j ? i : j, b = i > j ? j : i;
                acc = fma(G[a * (a + 1) / 2 + b], u[j], acc);
            }
            t2 = fma(u[i], acc, t2);
        }
        th = t2;
    }
    theta = th;
    return steps;
}

template <int P>
__device__ __forceinline__ void gram_add(double (&G)[P * (P + 1) / 2], const double (&a)[P])
{
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) G[i * (i + 1) / 2 + j] = fma(a[i], a[j], G[i * (i + 1) / 2 + j]);
}

template <int P> __device__ __forceinline__ double vmax(const double (&v)[P])
{ double m = v[0];
#pragma unroll
  for (int i = 1; i < P; i++) m = v[i] > m ? v[i] : m; return m; }
template <int P> __device__ __forceinline__ double vmin(const double (&v)[P])
{ double m = v[0];
#pragma unroll
  for (int i = 1; i < P; i++) m = v[i] < m ? v[i] : m; return m; }

// K = abs(K); K[K < 1e-5] = min(K[K >= 1e-5])    nmf.py:329-330, :361-362
template <int P> __device__ __forceinline__ int fix_k(double (&K)[P])
{
    double mn = INFINITY;
#pragma unroll
    for (int i = 0; i < P; i++) { K[i] = fabs(K[i]); if (K[i] >= 1e-5 && K[i] < mn) mn = K[i]; }
    if (mn == INFINITY) return ST_EMPTY_MIN;
#pragma unroll
    for (int i = 0; i < P; i++) if (K[i] < 1e-5) K[i] = mn;
    return ST_OK;
}

// np.nanmedian(1 - rho) > 1   nmf.py:257   (rank selection without sorting; p is tiny)
template <int P> __device__ __forceinline__ double median_of(const double (&v)[P])
{
    double lo = 0.0, hi = 0.0;
#pragma unroll
    for (int i = 0; i < P; i++) {
        int less = 0, eq = 0;
#pragma unroll
        for (int j = 0; j < P; j++) { less += v[j] < v[i]; eq += v[j] == v[i]; }
        // v[i] occupies sorted ranks [less, less + eq)
        const int r_lo = (P - 1) / 2, r_hi = P / 2;
        if (less <= r_lo && r_lo < less + eq) lo = v[i];
        if (less <= r_hi && r_hi < less + eq) hi = v[i];
    }
    return 0.5 * (lo + hi);
}

// ---------------------------------------------------------------------------------------------------
// One nmf() call on the compacted working matrix Fb (p x n, row stride S) -- nmf.py:78-107.
// On return u, theta describe the last SVD; sums[] = { sum_j s_j, clamped row sums (P), row sums of Fb (P) };
// rs[k] = squared relative residual of column k (nmf.py:280-282), sv[k] = s_k when store_s.
// ---------------------------------------------------------------------------------------------------
template <int P, int NT>
__device__ __forceinline__ int nmf_call(const double *__restrict__ Fb, double *__restrict__ Lm,
                                        double *__restrict__ rs, double *__restrict__ sv,
                                        int n, int S, int T, bool first, Smem<P, NT> &sm,
                                        double (&u)[P], double &theta, double (&sums)[2 * P + 1], int &steps)
{
    constexpr int NG = P * (P + 1) / 2;
    const int tid = threadIdx.x;
    double G[NG];

    // cold start: SVD of x itself (nmf.py:88)
#pragma unroll
    for (int i = 0; i < NG; i++) G[i] = 0.0;
    for (int k = tid; k < n; k += NT) {
        double x[P];
#pragma unroll
        for (int i = 0; i < P; i++) x[i] = Fb[(size_t) i * S + k];
        gram_add<P>(G, x);
    }
    block_sum<NG, P, NT>(G, sm);
    {
        double tr = 0.0;
#pragma unroll
        for (int i = 0; i < P; i++) tr += G[i * (i + 1) / 2 + i];
        if (!(tr > 0.0)) return ST_ARPACK;
    }
    const double u0 = 1.0 / sqrt((double) P);
#pragma unroll
    for (int i = 0; i < P; i++) u[i] = u0;
    steps += top_eig<P>(G, u, theta);

    const double c = 1.0 / sqrt((double) T);                         // nmf.py:91
    for (int t = 0; t < T; t++) {
#pragma unroll
        for (int i = 0; i < NG; i++) G[i] = 0.0;
        for (int k = tid; k < n; k += NT) {
            double x[P], l[P], a[P];
#pragma unroll
            for (int i = 0; i < P; i++) x[i] = Fb[(size_t) i * S + k];
            if (t > 0) {
#pragma unroll
                for (int i = 0; i < P; i++) l[i] = Lm[(size_t) i * S + k];
            } else {
#pragma unroll
                for (int i = 0; i < P; i++) l[i] = 0.0;                // lmbda = zeros, nmf.py:90
            }
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < P; i++) s = fma(u[i], x[i] + l[i], s);  // E_j * sigma = u . (x + lambda)_j
#pragma unroll
            for (int i = 0; i < P; i++) {
                const double res = fma(u[i], s, -x[i]);                // est - x          nmf.py:94
                double ln = fma(-c, res, l[i]);                        // lmbda -= c * res nmf.py:95
                ln = ln < 0.0 ? 0.0 : ln;                              //                  nmf.py:96
                Lm[(size_t) i * S + k] = ln;
                a[i] = x[i] + ln;                                      // x + lmbda        nmf.py:97
            }
            gram_add<P>(G, a);
        }
        block_sum<NG, P, NT>(G, sm);
        steps += top_eig<P>(G, u, theta);
    }

    // final pass: K E of the last SVD, its row sums, the clamped row sums and the residual profile.
    double acc[2 * P + 1];
#pragma unroll
    for (int i = 0; i < 2 * P + 1; i++) acc[i] = 0.0;
    for (int k = tid; k < n; k += NT) {
        double x[P];
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < P; i++) {
            x[i] = Fb[(size_t) i * S + k];
            s = fma(u[i], x[i] + Lm[(size_t) i * S + k], s);
        }
        acc[0] += s;
        double rmax = 0.0;
#pragma unroll
        for (int i = 0; i < P; i++) {
            const double ke = u[i] * s;
            acc[1 + i] += ke < x[i] ? x[i] : ke;                       // KE[KE < F] = F     nmf.py:318
            acc[1 + P + i] += x[i];
            double d = ke - x[i];
            if (!first) d = d < 0.0 ? 0.0 : d;                         // residual of the clamped KE on later trips
            const double r = d / (x[i] + 1.0);                         // (KE - F) / (F + 1) nmf.py:282
            const double r2 = r * r;
            rmax = r2 > rmax ? r2 : rmax;
        }
        rs[k] = rmax;
        if (first) sv[k] = s;
    }
    block_sum<2 * P + 1, P, NT>(acc, sm);
#pragma unroll
    for (int i = 0; i < 2 * P + 1; i++) sums[i] = acc[i];
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------------
// k_baseline: the per-gene state machine.  Gene-level vectors (rho, K, ...) are wave-uniform and live in
// LDS (GeneState) so that the registers belong to the Gram accumulators of the inner passes.
// ---------------------------------------------------------------------------------------------------
template <int P>
struct GeneState {
    double sumF[P];      // row sums of F_start                                   nmf.py:241, :337
    double rho[P];       // current DI vector
    double K[P];         // current K = u * sigma
    double us[P];        // u of the first nmf() call (K_start / sigma)           nmf.py:250
    double rho_fb[P];    // DI of max(K_start E_start, F_start)                   nmf.py:345-346, :352-353
    double sig0;         // sigma of the first call
};

template <int P> __device__ __forceinline__ double lds_max(const double *v)
{ double m = v[0];
#pragma unroll
  for (int i = 1; i < P; i++) { const double t = v[i]; m = t > m ? t : m; } return m; }
template <int P> __device__ __forceinline__ double lds_min(const double *v)
{ double m = v[0];
#pragma unroll
  for (int i = 1; i < P; i++) { const double t = v[i]; m = t < m ? t : m; } return m; }

template <int P, int NT>
__global__ __launch_bounds__(NT) void k_baseline(IterArgs A)
{
    constexpr int W = NT / 64;
    __shared__ Smem<P, NT> sm;
    __shared__ GeneState<P> gs;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int S = A.S;
    double *Fs = A.ws + (size_t) blockIdx.x * A.slot_stride;
    double *Fb = Fs + (size_t) P * S;
    double *Lm = Fb + (size_t) P * S;
    double *sv = Lm + (size_t) P * S;
    double *rs = sv + S;

    for (;;) {
        if (tid == 0) sm.gene = atomicAdd(A.counter, 1);
        __syncthreads();
        const int q = sm.gene;
        __syncthreads();
        if (q >= A.n_genes) break;
        const int g = A.order[q];
        const int L = A.glen[g];
        const float *x = A.cov + A.goff[g];

        int n0 = 0, n_calls = 0, n_drops = 0, exit_code = EXIT_LOW_COV, loop_reason = LOOP_NOT_ENTERED;
        int status = ST_OK, flag = 0, steps = 0, emode = EM_INPUT;
        long long sum_cols = 0;
        int32_t *tr = A.trace + (size_t) g * TRACE_LEN;
        if (tid < P) { gs.rho[tid] = 0.0; gs.K[tid] = 0.0; gs.us[tid] = 0.0; }

        // ---- get_high_coverage_idx (nmf.py:66-76) on F = x / s (nmf.py:146) -------------------------
        // max_j fl(x_ij / s_i) = fl((max_j x_ij) / s_i): division by a positive scalar is monotone.
        double thr;
        {
            float rmx[P];
#pragma unroll
            for (int i = 0; i < P; i++) rmx[i] = 0.0f;
            for (int j = tid; j < L; j += NT) {
#pragma unroll
                for (int i = 0; i < P; i++) rmx[i] = fmaxf(rmx[i], x[(size_t) i * L + j]);
            }
            block_max_f<P, P, NT>(rmx, sm);
            double gmax = (double) rmx[0] / A.scale[0];
#pragma unroll
            for (int i = 1; i < P; i++) { const double v = (double) rmx[i] / A.scale[i]; gmax = v > gmax ? v : gmax; }
            thr = 0.1 * gmax;
        }

        const int rate = A.rate;
        const long long ds0 = (rate > 1 && A.ds_start) ? A.ds_start[g] : -1;
        const int seg = ((L + W - 1) / W + 63) & ~63;
        const int jb = w * seg, je = (jb + seg < L) ? jb + seg : L;

        // pass 1: count per wave segment; pass 2: ordered compaction into Fs / Fb (nmf.py:236-238)
        {
            double sumF[P];
#pragma unroll
            for (int i = 0; i < P; i++) sumF[i] = 0.0;
            int base = 0;
            for (int pass = 0; pass < 2; pass++) {
                int run = 0;
                for (int c = jb; c < je; c += 64) {
                    const int j = c + lane;
                    bool hi = false;
                    double f[P];
                    if (j < je) {
                        double cm = 0.0;
#pragma unroll
                        for (int i = 0; i < P; i++) { f[i] = (double) x[(size_t) i * L + j] / A.scale[i]; cm = f[i] > cm ? f[i] : cm; }
                        hi = cm > thr;
                        if (ds0 >= 0) hi = hi && (j >= ds0) && ((j - ds0) % rate == 0);   // nmf.py:223-227
                    }
                    const unsigned long long mask = __ballot(hi);
                    if (pass == 1 && hi) {
                        const int pos = base + run + __popcll(mask & ((1ull << lane) - 1ull));
#pragma unroll
                        for (int i = 0; i < P; i++) {
                            Fs[(size_t) i * S + pos] = f[i];
                            Fb[(size_t) i * S + pos] = f[i];
                            sumF[i] += f[i];
                        }
                    }
                    run += __popcll(mask);
                }
                if (pass == 0) {
                    if (lane == 0) sm.cnt[w] = run;
                    __syncthreads();
                    n0 = 0;
#pragma unroll
                    for (int ww = 0; ww < W; ww++) { if (ww < w) base += sm.cnt[ww]; n0 += sm.cnt[ww]; }
                    __syncthreads();
                    if (n0 < A.min_hc) break;                                          // nmf.py:232
                }
            }
            if (n0 >= A.min_hc) {
                block_sum<P, P, NT>(sumF, sm);
                if (tid == 0) {
#pragma unroll
                    for (int i = 0; i < P; i++) gs.sumF[i] = sumF[i];
                }
            }
            __syncthreads();
        }

        int n = n0;                   // current width of Fb
        if (n0 >= A.min_hc) {
            if (!(lds_min<P>(gs.sumF) > 0.0)) exit_code = EXIT_ZERO_SAMPLE;            // nmf.py:241
            else {
                double u[P], theta = 0.0, sums[2 * P + 1];
                status = nmf_call<P, NT>(Fb, Lm, rs, sv, n, S, A.T, true, sm, u, theta, sums, steps);  // nmf.py:245
                n_calls = 1; sum_cols = n;
                if (status == ST_OK) {
                    if (tid == 0) {
                        const double sig = sqrt(theta);
                        gs.sig0 = sig;
#pragma unroll
                        for (int i = 0; i < P; i++) {
                            gs.us[i] = u[i];                                               // nmf.py:250
                            gs.K[i] = u[i] * sig;
                            gs.rho[i] = 1.0 - sums[1 + P + i] / (u[i] * sums[0] + 1.0);   // nmf.py:254
                            gs.rho_fb[i] = 1.0 - gs.sumF[i] / (sums[1 + i] + 1.0);
                        }
                    }
                    __syncthreads();
                    double med;
                    {
                        double om[P];
#pragma unroll
                        for (int i = 0; i < P; i++) om[i] = 1.0 - gs.rho[i];
                        med = median_of<P>(om);
                    }
                    if (med > 1.0) {                                                   // nmf.py:257
                        exit_code = EXIT_MEDIAN;
                    } else {
                        const double min_gene_len = fmax(2.0, ceil(200.0 * (1.0 / (double) rate)));       // nmf.py:261
                        const double min_bins = ceil((double) A.bins * 0.2);                             // nmf.py:35
                        emode = (n0 < L) ? EM_EXPAND : EM_RAW;
                        exit_code = EXIT_NO_LOOP;
                        if ((double) n0 >= min_gene_len && lds_min<P>(gs.rho) <= 0.2 && !A.skip) {      // nmf.py:265
                            // split_into_chunks(range(n0), bins)   utils.py:176-192, nmf.py:269-271
                            const int csize = (n0 + A.bins - 1) / A.bins;
                            int n_bins = (n0 + csize - 1) / csize;
                            if (tid < n_bins) sm.alive[tid] = tid;
                            __syncthreads();
                            while (lds_max<P>(gs.rho) > 0.1) {                                           // nmf.py:273
                                flag = 1;                                                                // nmf.py:276
                                loop_reason = LOOP_NATURAL;
                                // per-bin mean of rs[] (nmf.py:283): one wave per bin, fixed order
                                for (int b = w; b < n_bins; b += W) {
                                    const int kb = b * csize, ke = (kb + csize < n) ? kb + csize : n;
                                    double part = 0.0;
                                    for (int k = kb + lane; k < ke; k += 64) part += rs[k];
                                    part = wave_sum1(part);
                                    if (lane == 0) sm.ss[b] = part / (double) (ke - kb);
                                }
                                __syncthreads();
                                double best = -INFINITY; int drop = 0;
                                for (int b = 0; b < n_bins; b++) { const double v = sm.ss[b]; if (v > best) { best = v; drop = b; } }   // nmf.py:291
                                __syncthreads();
                                if (best == 0.0) { loop_reason = LOOP_PERFECT; break; }                  // nmf.py:286
                                // drop the bin, renumber (nmf.py:292-302); Fb is rebuilt from the pristine Fs
                                const int kb = drop * csize;
                                const int dlen = ((kb + csize < n) ? kb + csize : n) - kb;
                                if (tid == 0) {
                                    for (int b = drop; b < n_bins - 1; b++) sm.alive[b] = sm.alive[b + 1];
                                    if (n_drops < 32) tr[8 + n_drops] = drop;
                                }
                                n_bins--;
                                n -= dlen;
                                n_drops++;
                                __syncthreads();
                                for (int k = tid; k < n; k += NT) {
                                    const int a = k / csize;
                                    const int ko = sm.alive[a] * csize + (k - a * csize);
#pragma unroll
                                    for (int i = 0; i < P; i++) Fb[(size_t) i * S + k] = Fs[(size_t) i * S + ko];
                                }
                                __syncthreads();
                                if (n < 2) { loop_reason = LOOP_VALUE_ERROR; break; }                    // svds ValueError, nmf.py:306-310
                                const int st = nmf_call<P, NT>(Fb, Lm, rs, sv, n, S, A.T, false, sm, u, theta, sums, steps);
                                if (st != ST_OK) { status = st; break; }
                                n_calls++; sum_cols += n;
                                bool zero_row = false;
#pragma unroll
                                for (int i = 0; i < P; i++) zero_row = zero_row || (u[i] * sums[0] == 0.0);
                                if (tid == 0) {
                                    const double sg = sqrt(theta);
#pragma unroll
                                    for (int i = 0; i < P; i++) {
                                        gs.K[i] = u[i] * sg;                                             // nmf.py:307
                                        if (!zero_row) gs.rho[i] = 1.0 - sums[1 + P + i] / (sums[1 + i] + 1.0);   // nmf.py:318-321
                                    }
                                }
                                __syncthreads();
                                if (zero_row) { loop_reason = LOOP_ZERO_ROWSUM; break; }                 // nmf.py:315
                                if ((double) n_bins <= min_bins || (double) n < min_gene_len) { loop_reason = LOOP_MIN_BINS; break; }  // nmf.py:323
                            }
                            if (status == ST_OK) {
                                bool fallback = false;
                                if (lds_max<P>(gs.rho) < 0.2) {                                          // nmf.py:327
                                    double K[P];
#pragma unroll
                                    for (int i = 0; i < P; i++) K[i] = gs.K[i];
                                    status = fix_k<P>(K);                                                // nmf.py:329-330
                                    if (status == ST_OK) {
                                        double se[1] = {0.0};
                                        for (int k = tid; k < n0; k += NT) {                             // nmf.py:333
                                            double m = -INFINITY;
#pragma unroll
                                            for (int i = 0; i < P; i++) { const double qv = Fs[(size_t) i * S + k] / K[i]; m = qv > m ? qv : m; }
                                            se[0] += m;
                                        }
                                        block_sum<1, P, NT>(se, sm);
                                        double rmax = -INFINITY;
#pragma unroll
                                        for (int i = 0; i < P; i++) {
                                            const double r = 1.0 - gs.sumF[i] / (K[i] * se[0] + 1.0);     // nmf.py:334-337
                                            rmax = r > rmax ? r : rmax;
                                        }
                                        __syncthreads();
                                        if (rmax > 0.9) { fallback = true; exit_code = EXIT_REFINE_FALLBACK; }          // nmf.py:342
                                        else {
                                            exit_code = EXIT_REFINED; emode = (n0 < L) ? EM_EXPAND : EM_REFINED;
                                            if (tid == 0) {
#pragma unroll
                                                for (int i = 0; i < P; i++) { gs.K[i] = K[i]; gs.rho[i] = 1.0 - gs.sumF[i] / (K[i] * se[0] + 1.0); }
                                            }
                                        }
                                    }
                                } else { fallback = true; exit_code = EXIT_NOT_FOUND_FALLBACK; }          // nmf.py:349
                                if (fallback && status == ST_OK) {
                                    if (tid == 0) {
#pragma unroll
                                        for (int i = 0; i < P; i++) { gs.K[i] = gs.us[i] * gs.sig0; gs.rho[i] = gs.rho_fb[i]; }
                                    }
                                    emode = (n0 < L) ? EM_EXPAND : EM_CLAMPED;
                                }
                                __syncthreads();
                            }
                        }
                        // the re-expansion fix-up runs (and may raise) whenever the estimate is narrower than F  nmf.py:358-362
                        if (status == ST_OK && n0 < L) {
                            double K[P];
#pragma unroll
                            for (int i = 0; i < P; i++) K[i] = gs.K[i];
                            status = fix_k<P>(K);
                            __syncthreads();
                            if (tid == 0) {
#pragma unroll
                                for (int i = 0; i < P; i++) gs.K[i] = K[i];
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();

        const bool zero_out = (status != ST_OK) || exit_code <= EXIT_MEDIAN;
        if (zero_out) { emode = EM_INPUT; if (status != ST_OK) flag = 0; }

        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < P; i++) {
                A.rho[(size_t) g * P + i] = zero_out ? 0.0 : gs.rho[i];
                // EM_CLAMPED / EM_RAW rebuild K_start E_start as us[i] * s_j; the others use K
                A.kfin[(size_t) g * P + i] = (emode == EM_CLAMPED || emode == EM_RAW) ? gs.us[i] : gs.K[i];
            }
            A.flags[g] = flag;
            A.emode[g] = emode;
            tr[0] = n0; tr[1] = n_calls; tr[2] = (int32_t) sum_cols; tr[3] = exit_code; tr[4] = loop_reason;
            tr[5] = n_drops; tr[6] = status; tr[7] = steps;
        }
        if (A.want_est && (emode == EM_CLAMPED || emode == EM_RAW)) {
            double *dst = A.svec + A.svoff[g];
            for (int k = tid; k < n0; k += NT) dst[k] = sv[k];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// k_ratio_svd: initial DI pass on the raw coverage -- ratio_svd (nmf.py:109-121) and the row sums of
// nmf.py:524-525.  Two streaming passes over the packed fp32 coverage per gene.
// ---------------------------------------------------------------------------------------------------
template <int P, int NT>
__global__ __launch_bounds__(NT) void k_ratio_svd(InitArgs A)
{
    constexpr int NG = P * (P + 1) / 2;
    __shared__ Smem<P, NT> sm;
    const int tid = threadIdx.x;
    for (;;) {
        if (tid == 0) sm.gene = atomicAdd(A.counter, 1);
        __syncthreads();
        const int q = sm.gene;
        __syncthreads();
        if (q >= A.n_genes) break;
        const int g = A.order[q];
        const int L = A.glen[g];
        const float *x = A.cov + A.goff[g];
        int status = ST_OK;
        double acc[2 * P];
#pragma unroll
        for (int i = 0; i < 2 * P; i++) acc[i] = 0.0;
        if (L < 2) status = ST_VALUE_ERROR;
        else {
            double G[NG];
#pragma unroll
            for (int i = 0; i < NG; i++) G[i] = 0.0;
            for (int j = tid; j < L; j += NT) {
                double v[P];
#pragma unroll
                for (int i = 0; i < P; i++) v[i] = (double) x[(size_t) i * L + j];
                gram_add<P>(G, v);
            }
            block_sum<NG, P, NT>(G, sm);
            double trc = 0.0;
#pragma unroll
            for (int i = 0; i < P; i++) trc += G[i * (i + 1) / 2 + i];
            if (!(trc > 0.0)) status = ST_ARPACK;
            else {
                double u[P], theta;
#pragma unroll
                for (int i = 0; i < P; i++) u[i] = 1.0 / sqrt((double) P);
                top_eig<P>(G, u, theta);
                for (int j = tid; j < L; j += NT) {
                    double v[P], s = 0.0;
#pragma unroll
                    for (int i = 0; i < P; i++) { v[i] = (double) x[(size_t) i * L + j]; s = fma(u[i], v[i], s); }
#pragma unroll
                    for (int i = 0; i < P; i++) {
                        const double ke = u[i] * s;
                        acc[i] += ke < v[i] ? v[i] : ke;              // est[est < x] = x[est < x]   nmf.py:119
                        acc[P + i] += v[i];
                    }
                }
                block_sum<2 * P, P, NT>(acc, sm);
            }
        }
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < P; i++) {
                A.est_sums[(size_t) g * P + i] = status == ST_OK ? acc[i] : 0.0;
                A.cov_sums[(size_t) g * P + i] = status == ST_OK ? acc[P + i] : 0.0;
            }
            A.status[g] = status;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// k_estimates: streaming rebuild of the estimated coverage matrices (float64 out), one block per
// (gene, 256-column tile).  blockIdx.x walks a host-built tile list: tile_gene[t], tile_col0[t].
// ---------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void k_estimates(EstArgs A, const int32_t *__restrict__ tile_gene,
                                                   const int32_t *__restrict__ tile_col0)
{
    const int g = tile_gene[blockIdx.x];
    const int j = tile_col0[blockIdx.x] + threadIdx.x;
    const int L = A.glen[g];
    if (j >= L) return;
    const float *x = A.cov + A.goff[g];
    double *o = A.out + A.goff[g];
    const int em = A.emode[g];
    double f[P], K[P];
#pragma unroll
    for (int i = 0; i < P; i++) { f[i] = (double) x[(size_t) i * L + j] / A.scale[i]; K[i] = A.kfin[(size_t) g * P + i]; }
    if (em == EM_INPUT) {
#pragma unroll
        for (int i = 0; i < P; i++) o[(size_t) i * L + j] = f[i];
    } else if (em == EM_EXPAND || em == EM_REFINED) {
        double m = -INFINITY;
#pragma unroll
        for (int i = 0; i < P; i++) { const double q = f[i] / K[i]; m = q > m ? q : m; }
#pragma unroll
        for (int i = 0; i < P; i++) {
            double v = K[i] * m;
            if (em == EM_EXPAND) v = v < f[i] ? f[i] : v;
            o[(size_t) i * L + j] = v;
        }
    } else {
        const double s = A.svec[A.svoff[g] + j];
#pragma unroll
        for (int i = 0; i < P; i++) {
            double v = K[i] * s;
            if (em == EM_CLAMPED) v = v < f[i] ? f[i] : v;
            o[(size_t) i * L + j] = v;
        }
    }
}

// Launchers instantiated per P in dn_inst.hip ---------------------------------------------------------
typedef void (*baseline_launch_fn)(const IterArgs &, int grid, hipStream_t);
typedef void (*init_launch_fn)(const InitArgs &, int grid, hipStream_t);
typedef void (*est_launch_fn)(const EstArgs &, const int32_t *, const int32_t *, int n_tiles, hipStream_t);
typedef int (*occupancy_fn)(int which);

struct KernelSet {
    int p;
    int nt;
    baseline_launch_fn baseline;
    init_launch_fn init;
    est_launch_fn est;
    occupancy_fn blocks_per_cu;       // which: 0 baseline, 1 init
    size_t slot_doubles_per_col;      // scratch doubles per column of stride S
    const char *baseline_name;
};

const KernelSet *kernel_set_for(int p);   // dn_api.hip

}  // namespace dn
