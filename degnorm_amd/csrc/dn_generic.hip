// dn_generic.hip -- run-time-p variants of the kernels for sample counts above the register-resident range
// (any p <= 64; BASELINE config 4 is p = 50 with take-every 500, active matrices 50 x <= 10).
//
// Same state machine and outputs as k_baseline<P,...> (dn_kernels.hpp; reference nmf.py:189-372), but the
// p x p Gram matrix no longer fits in registers, so the top singular vector of x + lambda is found by power
// iteration on A A^T applied as two passes (s = A^T u per column, y = A s in row tiles of 8 through the block
// reduction), with the x + lambda state in the workgroup's scratch slot (L2-resident).  For p > 24 this is also
// cheaper than forming the Gram (2 p n per step against p^2 n / 2).  When n < p (the down-sampled regime) the
// same iteration converges to the same top singular triplet; scipy switches to the n x n Gram there, the result
// is the identical rank-1 factor -- and so does nmf_rows below, which serves active matrices of at most 12 columns
// (all of config 4) from registers, one wave per gene, with the machinery of the templated kernels.  The block-wide
// path (nmf_gen) is about coverage of the interface, not about speed.
//
// The file is compiled twice (build.py): -DDN_GEN_NT=256 is the general family (kernel_set_generic), -DDN_GEN_NT=64 the
// one-wavefront-per-gene family (kernel_set_rows) that serves data sets in which no gene can keep more than NSM_MAX
// active columns: there only nmf_rows runs, a single wave does all the work of a gene, and 64-thread workgroups let a
// CU keep 16 genes in flight instead of 2.
#include <cstdio>
#ifndef DN_GEN_NT
#define DN_GEN_NT 256
#endif
#define DN_P 8          // sizes the shared reduction scratch only (Smem<8, NT>)
#define DN_NT DN_GEN_NT
#include "dn_kernels.hpp"

#if DN_GEN_NT == 256
#define DN_GEN_NS gen
#undef DN_GEN_MINW
#define DN_GEN_MINW 2           // two workgroups per CU
#elif DN_GEN_NT == 64
#define DN_GEN_NS gen_rows
#ifndef DN_GEN_MINW
#define DN_GEN_MINW 3           // <= 168 registers, three waves per SIMD: the loop of nmf_rows is bound by instruction issue, not by
#endif                          // latency, and at 128 registers (four waves) its Gram products spill (config 4 sweep 6.4 -> 4.8 ms with
                                // the per-column-count instantiations; four waves: 4.9)
#else
#error "DN_GEN_NT must be 256 or 64"
#endif

namespace dn {
namespace DN_GEN_NS {

constexpr int GP = P_MAX;       // 64
constexpr int NT = DN_GEN_NT;
constexpr int TI = 8;           // rows per reduction tile

struct GState {
    double scale[GP], inv[GP], sumF[GP], rho[GP], K[GP], us[GP], rho_fb[GP], u[GP], y[GP], csum[GP], rsum[GP];
    double S, theta, sig0;
    int32_t status, steps;
    int32_t max_steps;          // cap of one eigen-solve in MFMA-solver units (IterArgs::max_steps); the plain power
};                              // iterations below (one step per pass) get 5x that many
__shared__ GState g_st;
__shared__ double g_rows_tot[256];     // nmf_rows: the wave's Gram totals (<= 78) and the solver's zero slot
__shared__ double g_rows_dsel[80];     // nmf_rows: 1.0 at the packed indices of the Gram diagonal (the reducing lane applies the shift there)

__device__ __forceinline__ void tile_sum(double (&part)[TI])          // totals left in g_sm.tot[0..TI)
{
    block_sum_lds<TI, DN_P, NT, double>(part, g_sm);
}

// y = A (A^T u) with the current u; returns ||y||^2 (every thread) and leaves y in g_st.y.  sj receives A^T u.
__device__ __forceinline__ double apply_gram(const double *A, double *sj, int n, int S, int p)
{
    const int tid = threadIdx.x;
    for (int k = tid; k < n; k += NT) {
        double s = 0.0;
        for (int i = 0; i < p; i++) s = fma(g_st.u[i], A[(size_t) i * S + k], s);
        sj[k] = s;
    }
    for (int i0 = 0; i0 < p; i0 += TI) {
        double part[TI];
#pragma unroll
        for (int r = 0; r < TI; r++) part[r] = 0.0;
        for (int k = tid; k < n; k += NT) {
            const double s = sj[k];
#pragma unroll
            for (int r = 0; r < TI; r++)
                if (i0 + r < p) part[r] = fma(A[(size_t) (i0 + r) * S + k], s, part[r]);
        }
        tile_sum(part);
        if (tid < TI && i0 + tid < p) g_st.y[i0 + tid] = g_sm.tot[tid];
        __syncthreads();
    }
    double n2 = 0.0;
    for (int i = 0; i < p; i++) n2 = fma(g_st.y[i], g_st.y[i], n2);
    return n2;
}

// Power iteration on A A^T from the current g_st.u to fp64 round-off (the reference's svds is tol = 0).
__device__ __forceinline__ int top_singular(const double *A, double *sj, int n, int S, int p)
{
    const int tid = threadIdx.x;
    const int cap = 5 * g_st.max_steps;
    for (int it = 0; ; it++) {
        if (it >= cap) return ST_NO_CONVERGENCE;
        const double n2 = apply_gram(A, sj, n, S, p);
        if (tid == 0) g_st.steps++;
        if (!(n2 > 0.0)) return ST_ARPACK;
        const double inv = 1.0 / sqrt(n2);
        double d2 = 0.0;
        for (int i = 0; i < p; i++) { const double d = g_st.y[i] * inv - g_st.u[i]; d2 = fma(d, d, d2); }
        __syncthreads();
        if (tid < p) g_st.u[tid] = g_st.y[tid] * inv;
        if (tid == 0) g_st.theta = sqrt(n2);              // ||A A^T u|| -> sigma^2 (error second order in the residual)
        __syncthreads();
        if (d2 <= 1e-27) break;
    }
    return ST_OK;
}

// One nmf() call (nmf.py:78-107) on Fb (raw counts of the active columns); results in g_st.
__device__ __attribute__((noinline)) void nmf_gen(const float *Fb, double *A, double *rs, double *sv, double *sj,
                                                   int n, int S, int T, int first_i, int p)
{
    const int tid = threadIdx.x;
    const bool first = first_i != 0;
    for (int k = tid; k < n; k += NT)                                    // lmbda = 0: state a = x
        for (int i = 0; i < p; i++) A[(size_t) i * S + k] = (double) Fb[(size_t) i * S + k] * g_st.inv[i];
    if (tid < p) g_st.u[tid] = 1.0 / sqrt((double) p);
    __syncthreads();
    int st = top_singular(A, sj, n, S, p);                                // SVD of x (nmf.py:88)
    bool cold_every_solve = false;                                        // dn_kernels.hpp, warm_start_unsafe
    for (int i = 0; i < p; i++) cold_every_solve = cold_every_solve || g_st.u[i] < WARM_START_MIN_COMPONENT;
    const double c = 1.0 / sqrt((double) T);
    for (int t = 0; t < T && st == ST_OK; t++) {
        for (int k = tid; k < n; k += NT) {
            double s = 0.0;
            for (int i = 0; i < p; i++) s = fma(g_st.u[i], A[(size_t) i * S + k], s);
            for (int i = 0; i < p; i++) {
                const double f = (double) Fb[(size_t) i * S + k] * g_st.inv[i];
                const double a = A[(size_t) i * S + k];
                A[(size_t) i * S + k] = fmax(fma(-c, fma(g_st.u[i], s, -f), a), f);     // nmf.py:94-97
            }
        }
        __syncthreads();
        if (cold_every_solve) {
            if (tid < p) g_st.u[tid] = 1.0 / sqrt((double) p);
            __syncthreads();
        }
        st = top_singular(A, sj, n, S, p);
    }
    if (st != ST_OK) { if (tid == 0) g_st.status = st; __syncthreads(); return; }
    // final pass: K E, its row sums, the clamped row sums, the residual profile
    double accS[TI];
#pragma unroll
    for (int r = 0; r < TI; r++) accS[r] = 0.0;
    for (int k = tid; k < n; k += NT) {
        double s = 0.0;
        for (int i = 0; i < p; i++) s = fma(g_st.u[i], A[(size_t) i * S + k], s);
        sj[k] = s;
        accS[0] += s;
        double rmax = 0.0;
        for (int i = 0; i < p; i++) {
            const double f = (double) Fb[(size_t) i * S + k] * g_st.inv[i];
            double d = g_st.u[i] * s - f;
            if (!first) d = d < 0.0 ? 0.0 : d;
            const double r = d / (f + 1.0);                                             // nmf.py:282
            rmax = r * r > rmax ? r * r : rmax;
        }
        rs[k] = rmax;
        if (first) sv[k] = s;
    }
    tile_sum(accS);
    if (tid == 0) g_st.S = g_sm.tot[0];
    __syncthreads();
    for (int i0 = 0; i0 < p; i0 += TI) {
        double pc[TI], pf[TI];
#pragma unroll
        for (int r = 0; r < TI; r++) { pc[r] = 0.0; pf[r] = 0.0; }
        for (int k = tid; k < n; k += NT) {
            const double s = sj[k];
#pragma unroll
            for (int r = 0; r < TI; r++) {
                if (i0 + r < p) {
                    const double f = (double) Fb[(size_t) (i0 + r) * S + k] * g_st.inv[i0 + r];
                    const double ke = g_st.u[i0 + r] * s;
                    pc[r] += ke < f ? f : ke;                                             // nmf.py:318
                    pf[r] += f;
                }
            }
        }
        tile_sum(pc);
        if (tid < TI && i0 + tid < p) g_st.csum[i0 + tid] = g_sm.tot[tid];
        __syncthreads();
        tile_sum(pf);
        if (tid < TI && i0 + tid < p) g_st.rsum[i0 + tid] = g_sm.tot[tid];
        __syncthreads();
    }
    if (tid == 0) g_st.status = ST_OK;
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------
// One nmf() call when the active matrix has at most NSM columns -- the down-sampled regime of BASELINE config 4
// (p = 50, take-every 500: 50 x <= 10).  There the small dimension is the column count, and a block-wide pass per
// inner iteration (nmf_gen: ~37 us each, almost all of it barriers) is the wrong shape.  Here sample s is lane s of
// wave 0 and keeps its row of x + lambda in registers; the right singular vector v comes from the n x n Gram matrix
// A^T A = sum over the lanes of a_s a_s^T (what scipy's svds does when n < p) through the same register reduce-scatter
// and MFMA squaring solver as the templated kernels, with no barrier inside the loop: ~2 us per inner iteration.
// K E = (A v) v^T, so u_s sigma = a_s . v.  Same outputs as nmf_gen.  The other waves of the workgroup wait.
// ---------------------------------------------------------------------------------------------------
constexpr int NSM_MAX = 12;                   // widest active matrix the row-wise routine takes
#ifndef DN_ROWS_EXACT
#define DN_ROWS_EXACT 1         // one instantiation of nmf_rows per column count 2..12 (0: capacities 4 / 6 / 8 / 10 / 12)
#endif
constexpr int ROWS_ZSLOT = 255;               // last double of g_rows_tot, kept at 0.0 (padding lanes of the solver read it)

// NSM: compiled column capacity (4, 6, 8, 10 or 12: the Gram matrix and the solver's tile shrink with it)
// SAFE: the repeat of a call whose result shows samples decoupled from the top block (dn_kernels.hpp, warm_start_unsafe): every solve
// block by block.  A separate instantiation -- inside the hot one the rare path cost 32 % (17.6 against 13.3 ms per launch on config 4)
template <int NSM, bool SAFE = false>
__device__ __attribute__((noinline)) void nmf_rows(const float *Fb, double *rs, double *sv, double *sj,
                                                   int n, int S, int T, int first_i, int p)
{
    constexpr int NGS = NSM * (NSM + 1) / 2;  // packed Gram entries: 10, 36, 78
    const bool first = first_i != 0;
    // arguments of a non-inlined function arrive in vector registers: as scalars the loop's tests (last iteration, cold solve, step
    // cap) are scalar branches, not exec masks with a branch around either side
    T = __builtin_amdgcn_readfirstlane(T);
    n = __builtin_amdgcn_readfirstlane(n);
    p = __builtin_amdgcn_readfirstlane(p);
    if (wave_id() == 0) {
        const int lane = lane_id();
        const bool live = lane < p;
        double *tot = g_rows_tot;
        const double inv_s = live ? g_st.inv[lane] : 0.0;
        double f[NSM], a[NSM], v[NSM];
#pragma unroll
        for (int j = 0; j < NSM; j++) {
            const float xv = (live && j < n) ? Fb[(size_t) lane * S + j] : 0.0f;
            f[j] = (double) xv * inv_s;
            a[j] = f[j];                                                // lmbda = 0 (nmf.py:90)
            v[j] = 0.0;
        }
        if (lane == 0) tot[ROWS_ZSLOT] = 0.0;
        for (int e = lane; e < NGS; e += 64) {
            bool diag = false;
#pragma unroll
            for (int i = 0; i < NSM; i++) diag = diag || (e == i * (i + 1) / 2 + i);
            g_rows_dsel[e] = diag ? 1.0 : 0.0;
        }
        wave_fence();
        const double c = 1.0 / sqrt((double) T);                        // nmf.py:91
        Solver<NSM> sol;                                                // carried solver state (iterate, scale, shift)
        double theta = 0.0;
        int steps = 0, st = ST_OK;
        bool noconv = false;
        const int maxs = __builtin_amdgcn_readfirstlane(g_st.max_steps);
#pragma clang loop unroll(disable)
        for (int t = -1; t < T; t++) {                                  // t = -1: SVD of x itself (nmf.py:88)
            if (__builtin_expect(t >= 0, 1)) {
                double ts = 0.0;
#pragma unroll
                for (int j = 0; j < NSM; j++) ts = fma(a[j], v[j], ts);  // u_s sigma
#pragma unroll
                for (int j = 0; j < NSM; j++) {
                    const double res = fma(ts, v[j], -f[j]);            // K E - x                       nmf.py:94
                    a[j] = fmax(fma(-c, res, a[j]), f[j]);              // x + max(lambda - c res, 0)    nmf.py:95-97
                }
            }
            double C[NGS];
#pragma unroll
            for (int i = 0; i < NSM; i++)
#pragma unroll
                for (int j = 0; j <= i; j++) C[i * (i + 1) / 2 + j] = a[i] * a[j];
            // ceil(NGS / 64) reduce-scatter rounds; the lane that ends up with a diagonal entry subtracts the call's shift as it stores
            // it (round 4: the shift is fixed after the cold solve -- the trace, its test and the read-modify-write of the diagonal
            // with its two fences are the cold solve's alone; x + lambda >= x entry by entry, so a matrix that was not zero stays so)
            wave_fence();
            wave_round_store<NGS, 0, double>(C, tot, lane, t < 0 ? 0.0 : sol.shift(), g_rows_dsel, true);
            wave_fence();
            if (__builtin_expect(t < 0, 0)) {
                double tr = 0.0;
#pragma unroll
                for (int i = 0; i < NSM; i++) tr += tot[i * (i + 1) / 2 + i];
                if (!(tr > 0.0)) { st = ST_ARPACK; break; }
                sol.cold(tr, v);
            }
            int r;
            if constexpr (SAFE) {                                       // the safe repeat (k_baseline_gen decides): block by block, unshifted
                r = solve_by_blocks<NSM>(sol, tot, ROWS_ZSLOT, v, theta, maxs, n);
            } else r = sol.run(tot, ROWS_ZSLOT, v, theta, t == T - 1, maxs, t < 0);
            steps += r;
            noconv = noconv || r > maxs;
            wave_fence();
        }
        if (st == ST_OK) {
            const double sig = sqrt(theta);
            double ts = 0.0;
#pragma unroll
            for (int j = 0; j < NSM; j++) ts = fma(a[j], v[j], ts);      // u_s sigma for the final state
            double cs = 0.0, fs = 0.0, ssum = 0.0;
            double r2[NSM];
#pragma unroll
            for (int j = 0; j < NSM; j++) {
                const double ke = ts * v[j];                            // (K E)_sj
                cs += ke < f[j] ? f[j] : ke;                            // nmf.py:318
                fs += f[j];
                ssum += v[j];
                double d = ke - f[j];
                if (!first) d = d < 0.0 ? 0.0 : d;
                const double r = d / (f[j] + 1.0);                      // nmf.py:282
                r2[j] = r * r;
            }
#pragma unroll
            for (int j = 0; j < NSM; j++) {
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) r2[j] = fmax(r2[j], __shfl_xor(r2[j], o));
            }
#pragma unroll
            for (int j = 0; j < NSM; j++) {
                if (lane == j && j < n) {
                    const double sk = sig * v[j];                       // s_k = (A^T u)_k
                    sj[j] = sk;
                    rs[j] = r2[j];
                    if (first) sv[j] = sk;
                }
            }
            if (live) { g_st.u[lane] = ts / sig; g_st.csum[lane] = cs; g_st.rsum[lane] = fs; }
            if (lane == 0) { g_st.theta = theta; g_st.S = sig * ssum; }
        }
        if (st == ST_OK && noconv) st = ST_NO_CONVERGENCE;
        if (lane == 0) { g_st.status = st; g_st.steps += steps; }
    }
    __syncthreads();
}

__device__ __forceinline__ double st_max(const double *v, int p) { double m = v[0]; for (int i = 1; i < p; i++) m = v[i] > m ? v[i] : m; return m; }
__device__ __forceinline__ double st_min(const double *v, int p) { double m = v[0]; for (int i = 1; i < p; i++) m = v[i] < m ? v[i] : m; return m; }

// K = abs(K); K[K < 1e-5] = min(K[K >= 1e-5])   nmf.py:329-330, :361-362   (thread 0 only, g_st.K in place)
__device__ __forceinline__ int fix_k_lds(int p)
{
    double mn = INFINITY;
    for (int i = 0; i < p; i++) { const double k = fabs(g_st.K[i]); if (k >= 1e-5 && k < mn) mn = k; }
    __syncthreads();
    if (mn == INFINITY) return ST_EMPTY_MIN;
    if (threadIdx.x == 0) for (int i = 0; i < p; i++) { const double k = fabs(g_st.K[i]); g_st.K[i] = k < 1e-5 ? mn : k; }
    return ST_OK;
}

__global__ __launch_bounds__(NT, DN_GEN_MINW) void k_baseline_gen(IterArgs A)
{
    constexpr int W = NT / 64;
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const int p = A.p, S = A.S;
    char *slot = A.ws + (size_t) blockIdx.x * A.slot_bytes;
    float *Fs = reinterpret_cast<float *>(slot);
    float *Fb = Fs + (size_t) p * S;
    double *Ast = reinterpret_cast<double *>(Fb + (size_t) p * S);
    double *sv = Ast + (size_t) p * S;
    double *rs = sv + S;
    double *sj = rs + S;
    if (tid < p) { g_st.scale[tid] = A.scale[tid]; g_st.inv[tid] = A.inv_scale[tid]; }
    if (tid == 0) g_st.max_steps = A.max_steps > 0 ? A.max_steps : EIG_MAX_STEPS_DEFAULT;
    __syncthreads();

    for (;;) {
        if (tid == 0) g_sm.gene = atomicAdd(A.counter, 1);
        __syncthreads();
        const int q = g_sm.gene;
        __syncthreads();
        if (q >= A.n_genes) break;
        const int g = A.order[q];
        const int L = A.glen[g];
        const float *x = A.cov + A.goff[g];
        int n0 = 0, n_calls = 0, n_drops = 0, exit_code = EXIT_LOW_COV, loop_reason = LOOP_NOT_ENTERED;
        int status = ST_OK, flag = 0, emode = EM_INPUT;
        long long sum_cols = 0;
        int32_t *tr = A.trace + (size_t) g * TRACE_LEN;
        if (tid < p) { g_st.rho[tid] = 0.0; g_st.K[tid] = 0.0; g_st.us[tid] = 0.0; }
        if (tid == 0) g_st.steps = 0;
        __syncthreads();

        // get_high_coverage_idx (nmf.py:66-76) on F = x / s, true quotients.  max_j fl(x_ij / s_i) = fl((max_j x_ij) / s_i)
        // (division by a positive scalar is monotone): the threshold needs only the row maxima of the raw counts (k_row_max).
        double gm = 0.0;
        {
            const float *rmx = A.rowmax + (size_t) g * p;
            for (int i = 0; i < p; i++) { const double f = (double) rmx[i] / g_st.scale[i]; gm = f > gm ? f : gm; }
        }
        const double thr = 0.1 * gm;
        // candidate columns: every base, or only the systematic sample ds0 + m * rate when down-sampling (nmf.py:223-227)
        const int rate = A.rate;
        const long long ds0 = (rate > 1 && A.ds_start) ? A.ds_start[g] : -1;
        const int M = ds0 >= 0 ? (ds0 < L ? (int) ((L - 1 - ds0) / rate) + 1 : 0) : L;
        const int seg = ((M + W - 1) / W + 63) & ~63;
        const int jb = w * seg, je = (jb + seg < M) ? jb + seg : M;
        int base = 0;
        for (int pass = 0; pass < 2; pass++) {
            int run = 0;
            for (int cc = jb; cc < je; cc += 64) {
                const int m = cc + lane;
                const long long j = ds0 >= 0 ? ds0 + (long long) m * rate : (long long) m;
                bool hi = false;
                if (m < je) {
                    double cm = 0.0;
                    for (int i = 0; i < p; i++) { const double f = (double) x[(size_t) i * L + j] / g_st.scale[i]; cm = f > cm ? f : cm; }
                    hi = cm > thr;
                }
                const unsigned long long mask = __ballot(hi);
                if (pass == 1 && hi) {
                    const int pos = base + run + __popcll(mask & ((1ull << lane) - 1ull));
                    for (int i = 0; i < p; i++) { const float v = x[(size_t) i * L + j]; Fs[(size_t) i * S + pos] = v; Fb[(size_t) i * S + pos] = v; }
                }
                run += __popcll(mask);
            }
            if (pass == 0) {
                if (lane == 0) g_sm.cnt[w] = run;
                __syncthreads();
                n0 = 0;
                for (int ww = 0; ww < W; ww++) { if (ww < w) base += g_sm.cnt[ww]; n0 += g_sm.cnt[ww]; }
                __syncthreads();
                if (n0 < A.min_hc) break;
            }
        }
        __syncthreads();
        int n = n0;
        if (n0 >= A.min_hc) {
            // row sums of F_start (true quotients), row tiles of 8
            for (int i0 = 0; i0 < p; i0 += TI) {
                double part[TI];
#pragma unroll
                for (int r = 0; r < TI; r++) part[r] = 0.0;
                for (int k = tid; k < n0; k += NT) {
#pragma unroll
                    for (int r = 0; r < TI; r++)
                        if (i0 + r < p) part[r] += (double) Fs[(size_t) (i0 + r) * S + k] / g_st.scale[i0 + r];
                }
                tile_sum(part);
                if (tid < TI && i0 + tid < p) g_st.sumF[i0 + tid] = g_sm.tot[tid];
                __syncthreads();
            }
            if (!(st_min(g_st.sumF, p) > 0.0)) exit_code = EXIT_ZERO_SAMPLE;
            else {
                const double min_gene_len = fmax(2.0, ceil(200.0 * (1.0 / (double) rate)));
                const double min_bins = ceil((double) A.bins * 0.2);
                int csize = 1, n_bins = 0;
                bool first = true, in_loop = false;
                for (;;) {
                    // one instantiation per column count (DN_ROWS_EXACT) or per even capacity: the n x n Gram matrix has
                    // NSM (NSM + 1) / 2 entries to multiply and reduce-scatter over the lanes every inner iteration (the
                    // bulk of the loop's ~1 000 instructions; at most 64 entries go in ONE round) and the solver
                    // ceil(NSM / 4) MFMAs per product -- config 4's first calls have 9-10 columns
#if DN_ROWS_EXACT
#define DN_ROWS_CASE(N) case N: nmf_rows<N>(Fb, rs, sv, sj, n, S, A.T, first ? 1 : 0, p); break;
                    if (n <= NSM_MAX) {
                        switch (n) {
                            DN_ROWS_CASE(3) DN_ROWS_CASE(4) DN_ROWS_CASE(5) DN_ROWS_CASE(6) DN_ROWS_CASE(7) DN_ROWS_CASE(8)
                            DN_ROWS_CASE(9) DN_ROWS_CASE(10) DN_ROWS_CASE(11) DN_ROWS_CASE(12)
                            default: nmf_rows<2>(Fb, rs, sv, sj, n, S, A.T, first ? 1 : 0, p); break;
                        }
                    }
#undef DN_ROWS_CASE
#else
                    if (n <= 4) nmf_rows<4>(Fb, rs, sv, sj, n, S, A.T, first ? 1 : 0, p);
                    else if (n <= 6) nmf_rows<6>(Fb, rs, sv, sj, n, S, A.T, first ? 1 : 0, p);
                    else if (n <= 8) nmf_rows<8>(Fb, rs, sv, sj, n, S, A.T, first ? 1 : 0, p);
                    else if (n <= 10) nmf_rows<10>(Fb, rs, sv, sj, n, S, A.T, first ? 1 : 0, p);
                    else if (n <= NSM_MAX) nmf_rows<12>(Fb, rs, sv, sj, n, S, A.T, first ? 1 : 0, p);
#endif
                    else nmf_gen(Fb, Ast, rs, sv, sj, n, S, A.T, first ? 1 : 0, p);
                    if (n <= NSM_MAX) {
                        // the safe repeat (dn_kernels.hpp, warm_start_unsafe): a covered sample whose component of u is (nearly) zero is
                        // decoupled from the block the call ended on, a solve in its step cap is most likely two blocks racing
                        bool unsafe = g_st.status == ST_NO_CONVERGENCE;
                        if (g_st.status == ST_OK) {                     // one sample per lane, one ballot (every wave sees all p <= 64 samples)
                            const int l = threadIdx.x & 63;
                            const bool mine = l < p && g_st.rsum[l] > 0.0 && g_st.u[l] < WARM_START_MIN_COMPONENT;
                            unsafe = __ballot(mine) != 0ull;
                        }
                        if (unsafe) {
                            __syncthreads();
#if DN_ROWS_EXACT
#define DN_ROWS_CASE(N) case N: nmf_rows<N, true>(Fb, rs, sv, sj, n, S, A.T, first ? 1 : 0, p); break;
                            switch (n) {
                                DN_ROWS_CASE(3) DN_ROWS_CASE(4) DN_ROWS_CASE(5) DN_ROWS_CASE(6) DN_ROWS_CASE(7) DN_ROWS_CASE(8)
                                DN_ROWS_CASE(9) DN_ROWS_CASE(10) DN_ROWS_CASE(11) DN_ROWS_CASE(12)
                                default: nmf_rows<2, true>(Fb, rs, sv, sj, n, S, A.T, first ? 1 : 0, p); break;
                            }
#undef DN_ROWS_CASE
#else
                            nmf_rows<12, true>(Fb, rs, sv, sj, n, S, A.T, first ? 1 : 0, p);
#endif
                        }
                    }
                    if (g_st.status != ST_OK) { status = g_st.status; break; }
                    n_calls++; sum_cols += n;
                    if (first) {
                        __syncthreads();
                        if (tid == 0) {
                            const double sig = sqrt(g_st.theta);
                            g_st.sig0 = sig;
                            for (int i = 0; i < p; i++) {
                                g_st.us[i] = g_st.u[i];
                                g_st.K[i] = g_st.u[i] * sig;
                                g_st.rho[i] = 1.0 - g_st.rsum[i] / (g_st.u[i] * g_st.S + 1.0);
                                g_st.rho_fb[i] = 1.0 - g_st.sumF[i] / (g_st.csum[i] + 1.0);
                            }
                        }
                        __syncthreads();
                        // nanmedian(1 - rho) > 1   (nmf.py:257): rank selection
                        double lo = 0.0, hi_v = 0.0;
                        for (int i = 0; i < p; i++) {
                            const double vi = 1.0 - g_st.rho[i];
                            int less = 0, eq = 0;
                            for (int j2 = 0; j2 < p; j2++) { const double vj = 1.0 - g_st.rho[j2]; less += vj < vi; eq += vj == vi; }
                            const int r_lo = (p - 1) / 2, r_hi = p / 2;
                            if (less <= r_lo && r_lo < less + eq) lo = vi;
                            if (less <= r_hi && r_hi < less + eq) hi_v = vi;
                        }
                        if (0.5 * (lo + hi_v) > 1.0) { exit_code = EXIT_MEDIAN; break; }
                        emode = (n0 < L) ? EM_EXPAND : EM_RAW;
                        exit_code = EXIT_NO_LOOP;
                        if (!((double) n0 >= min_gene_len && st_min(g_st.rho, p) <= 0.2 && !A.skip)) break;
                        in_loop = true;
                        csize = (n0 + A.bins - 1) / A.bins;
                        n_bins = (n0 + csize - 1) / csize;
                        if (tid < n_bins) g_sm.alive[tid] = tid;
                        __syncthreads();
                        first = false;
                    } else {
                        bool zero_row = false;
                        for (int i = 0; i < p; i++) zero_row = zero_row || (g_st.rsum[i] == 0.0);   // see dn_kernels.hpp, nmf.py:315
                        __syncthreads();
                        if (tid == 0) {
                            const double sg = sqrt(g_st.theta);
                            for (int i = 0; i < p; i++) {
                                g_st.K[i] = g_st.u[i] * sg;
                                if (!zero_row) g_st.rho[i] = 1.0 - g_st.rsum[i] / (g_st.csum[i] + 1.0);
                            }
                        }
                        __syncthreads();
                        if (zero_row) { loop_reason = LOOP_ZERO_ROWSUM; break; }
                        if ((double) n_bins <= min_bins || (double) n < min_gene_len) { loop_reason = LOOP_MIN_BINS; break; }
                    }
                    if (!(st_max(g_st.rho, p) > 0.1)) break;
                    flag = 1;
                    loop_reason = LOOP_NATURAL;
                    for (int b = w; b < n_bins; b += W) {
                        const int kb = b * csize, ke = (kb + csize < n) ? kb + csize : n;
                        double part = 0.0;
                        for (int k = kb + lane; k < ke; k += 64) part += rs[k];
                        part = wave_sum1(part);
                        if (lane == 0) g_sm.ss[b] = part / (double) (ke - kb);
                    }
                    __syncthreads();
                    double best = -INFINITY; int drop = 0;
                    for (int b = 0; b < n_bins; b++) { const double v = g_sm.ss[b]; if (v > best) { best = v; drop = b; } }
                    __syncthreads();
                    if (best == 0.0) { loop_reason = LOOP_PERFECT; break; }
                    const int kb = drop * csize;
                    const int dlen = ((kb + csize < n) ? kb + csize : n) - kb;
                    if (tid == 0) {
                        for (int b = drop; b < n_bins - 1; b++) g_sm.alive[b] = g_sm.alive[b + 1];
                        if (n_drops < 32) tr[8 + n_drops] = drop;
                    }
                    n_bins--; n -= dlen; n_drops++;
                    __syncthreads();
                    for (int k = tid; k < n; k += NT) {
                        const int a = k / csize;
                        const int ko = g_sm.alive[a] * csize + (k - a * csize);
                        for (int i = 0; i < p; i++) Fb[(size_t) i * S + k] = Fs[(size_t) i * S + ko];
                    }
                    __syncthreads();
                    if (n < 2) { loop_reason = LOOP_VALUE_ERROR; break; }
                }

                if (in_loop && status == ST_OK) {
                    bool fallback = false;
                    if (st_max(g_st.rho, p) < 0.2) {
                        __syncthreads();
                        status = fix_k_lds(p);
                        __syncthreads();
                        if (status == ST_OK) {
                            double se[TI];
#pragma unroll
                            for (int r = 0; r < TI; r++) se[r] = 0.0;
                            for (int k = tid; k < n0; k += NT) {
                                double m = -INFINITY;
                                for (int i = 0; i < p; i++) { const double qv = ((double) Fs[(size_t) i * S + k] * g_st.inv[i]) / g_st.K[i]; m = qv > m ? qv : m; }
                                se[0] += m;
                            }
                            tile_sum(se);
                            const double sE = g_sm.tot[0];
                            double rmax = -INFINITY;
                            for (int i = 0; i < p; i++) { const double r = 1.0 - g_st.sumF[i] / (g_st.K[i] * sE + 1.0); rmax = r > rmax ? r : rmax; }
                            __syncthreads();
                            if (rmax > 0.9) { fallback = true; exit_code = EXIT_REFINE_FALLBACK; }
                            else {
                                exit_code = EXIT_REFINED; emode = (n0 < L) ? EM_EXPAND : EM_REFINED;
                                if (tid == 0) for (int i = 0; i < p; i++) g_st.rho[i] = 1.0 - g_st.sumF[i] / (g_st.K[i] * sE + 1.0);
                            }
                        }
                    } else { fallback = true; exit_code = EXIT_NOT_FOUND_FALLBACK; }
                    if (fallback && status == ST_OK) {
                        __syncthreads();
                        if (tid == 0) for (int i = 0; i < p; i++) { g_st.K[i] = g_st.us[i] * g_st.sig0; g_st.rho[i] = g_st.rho_fb[i]; }
                        emode = (n0 < L) ? EM_EXPAND : EM_CLAMPED;
                    }
                    __syncthreads();
                }
                if (status == ST_OK && exit_code >= EXIT_NO_LOOP && n0 < L) {
                    __syncthreads();
                    status = fix_k_lds(p);
                    __syncthreads();
                }
            }
        }
        __syncthreads();
        const bool zero_out = (status != ST_OK) || exit_code <= EXIT_MEDIAN;
        if (zero_out) { emode = EM_INPUT; if (status != ST_OK) flag = 0; }
        if (tid < p) {
            A.rho[(size_t) g * p + tid] = zero_out ? 0.0 : g_st.rho[tid];
            A.kfin[(size_t) g * p + tid] = (emode == EM_CLAMPED || emode == EM_RAW) ? g_st.us[tid] : g_st.K[tid];
        }
        if (tid == 0) {
            A.flags[g] = flag; A.emode[g] = emode;
            tr[0] = n0; tr[1] = n_calls; tr[2] = (int32_t) sum_cols; tr[3] = exit_code; tr[4] = loop_reason;
            tr[5] = n_drops; tr[6] = status; tr[7] = g_st.steps;
        }
        if (A.want_est && (emode == EM_CLAMPED || emode == EM_RAW)) {
            double *dst = A.svec + A.svoff[g];
            for (int k = tid; k < n0; k += NT) dst[k] = sv[k];
        }
        __syncthreads();
    }
}

#if DN_GEN_NT == 256
// ratio_svd + row sums (nmf.py:109-121, :524-525) for run-time p; uses the same scratch slots.
// y = A (A^T u) on the RAW fp32 coverage (row stride L) in ONE pass: every thread forms s = u . a_j for its columns and
// adds s a_j to per-thread partials of all p rows (registers), which are then block-reduced in tiles of 8.  The
// block-wide apply_gram above reads the matrix twice per step from an fp64 copy; the initial pass works on the whole
// transcript of every gene, so here the bytes are what counts: 4 B per element and step.
__device__ __forceinline__ double apply_gram_raw(const float *x, int L, int p)
{
    const int tid = threadIdx.x;
    double y[GP];
#pragma unroll
    for (int i = 0; i < GP; i++) y[i] = 0.0;
    for (int k = tid; k < L; k += NT) {
        float xv[GP];
#pragma unroll
        for (int i = 0; i < GP; i++) xv[i] = i < p ? x[(size_t) i * L + k] : 0.0f;
        double sdot = 0.0;
#pragma unroll
        for (int i = 0; i < GP; i++) if (i < p) sdot = fma(g_st.u[i], (double) xv[i], sdot);
#pragma unroll
        for (int i = 0; i < GP; i++) if (i < p) y[i] = fma((double) xv[i], sdot, y[i]);
    }
#pragma unroll
    for (int i0 = 0; i0 < GP; i0 += TI) {
        if (i0 < p) {                                                   // uniform
            double part[TI];
#pragma unroll
            for (int r = 0; r < TI; r++) part[r] = y[i0 + r];
            tile_sum(part);
            if (tid < TI && i0 + tid < p) g_st.y[i0 + tid] = g_sm.tot[tid];
            __syncthreads();
        }
    }
    double n2 = 0.0;
    for (int i = 0; i < p; i++) n2 = fma(g_st.y[i], g_st.y[i], n2);
    return n2;
}

__device__ __forceinline__ int top_singular_raw(const float *x, int L, int p)
{
    const int tid = threadIdx.x;
    const int cap = 5 * g_st.max_steps;
    for (int it = 0; ; it++) {
        if (it >= cap) return ST_NO_CONVERGENCE;
        const double n2 = apply_gram_raw(x, L, p);
        if (tid == 0) g_st.steps++;
        if (!(n2 > 0.0)) return ST_ARPACK;
        const double inv = 1.0 / sqrt(n2);
        double d2 = 0.0;
        for (int i = 0; i < p; i++) { const double d = g_st.y[i] * inv - g_st.u[i]; d2 = fma(d, d, d2); }
        __syncthreads();
        if (tid < p) g_st.u[tid] = g_st.y[tid] * inv;
        __syncthreads();
        if (d2 <= 1e-27) break;
    }
    return ST_OK;
}

// Initial DI pass (nmf.py:109-121, :522-525): top left singular vector of the raw coverage, then the clamped and plain
// row sums in one more pass (per-thread partials of all rows, reduced in tiles).
// ---------------------------------------------------------------------------------------------------
// k_ratio_svd_mg: the initial DI pass (nmf.py:109-121, :522-525) for 17 <= p <= 64 in TWO streaming passes over the raw
// fp32 coverage -- SURVEY 8(d)'s algorithmic 8 p L bytes per gene -- instead of one pass per power step (k_ratio_svd_gen
// below read a gene ~20 times).
//   pass 1  Gram matrix G = X X^T of the p x L matrix on the fp64 matrix cores.  v_mfma_f64_16x16x4_f64 contracts over 4
//           columns; WHICH 4 columns share an instruction does not matter for a sum over all columns, so lane
//           (i = l & 15, k = l >> 4) loads ONE float4 = columns c0 + 4k .. 4k + 3 of row 16 t + i (64 contiguous bytes
//           per row and k-group, no transposition), and round m = 0..3 multiplies component m of every lane:
//           D[t1][t2] += X_t1(:, {c0 + m, c0 + 4 + m, c0 + 8 + m, c0 + 12 + m}) X_t2(...)^T.  The same register is the A
//           and the B operand.  Row p of the padded matrix is a row of ones, so that G[p][i] = sum_j x_ij: the plain
//           row sums (cov_sums) fall out of the same products.  Waves take 16-column groups round-robin; their tiles are
//           added in LDS one wave after the other (fixed order: deterministic).  (Where every count fits 16 bits the Gram
//           matrix is formed EXACTLY on the i8 matrix cores instead: mg_gram_pass_i8 below; this fp64 form serves the rest.)
//   solve   top eigenvector of the p x p block by shifted power iteration, the matrix in LDS, ONE wave (lane = row, no barrier
//           inside the iteration), two steps between convergence checks, the same stopping rule as top_eig_rows.
//   pass 2  one column per lane: s_j = u . x_j from the p counts of the column (registers), per-lane partial sums of
//           max(u_i s_j, x_ij) for every row kept in registers for the whole gene, reduced once at the end.
// ---------------------------------------------------------------------------------------------------
// The two streaming passes are inlined into the kernel (round 4): as functions of their own they saved and restored ~110 callee-saved
// vector registers per call -- 55 KB of scratch traffic per wave and call, 440 KB per gene next to the gene's own 2 x 550 KB
// (3.94 against 4.29 ms on the 16 000-gene slice; DN_MG_INLINE=0 builds the functions).
#ifndef DN_MG_INLINE
#define DN_MG_INLINE 1
#endif
#if DN_MG_INLINE
#define DN_MG_FN __device__ __forceinline__
#else
#define DN_MG_FN __device__ __attribute__((noinline))
#endif
constexpr int MG_ROWS = 80;                  // 5 tiles of 16: p <= 64 samples + the row of ones
constexpr int MG_LD = MG_ROWS + 1;           // LDS row stride in doubles (odd: the column walk of the solver is conflict-light)
__shared__ double g_mg[MG_ROWS * MG_LD];
__shared__ double g_mv[2][MG_ROWS];
#ifdef DN_STAMP
__shared__ long long g_mg_ts[8];          // diagnostic build: clock reads inside pass 1 (thread 0)
#define DN_MG_TS(k) do { if (threadIdx.x == 0) g_mg_ts[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DN_MG_TS(k) do { } while (0)
#endif

template <int TR>
__device__ __forceinline__ void mg_gram_pass(const float *x, int L, int p)
{
    constexpr int NTILE = TR * (TR + 1) / 2;
    constexpr int W = NT / 64;
    const int lane = lane_id(), w = wave_id();
    const int li = lane & 15, lk = lane >> 4;
    dn_double4 acc[NTILE];
#pragma unroll
    for (int i = 0; i < NTILE; i++) acc[i] = dn_double4{0.0, 0.0, 0.0, 0.0};
    // per tile row: where this lane reads (rows >= p read row 0 and are masked afterwards: no branch around the loads)
    const float *rowp[TR];
    float keep[TR], fill[TR];
#pragma unroll
    for (int t = 0; t < TR; t++) {
        const int row = 16 * t + li;
        rowp[t] = x + (size_t) (row < p ? row : 0) * L + 4 * lk;
        keep[t] = row < p ? 1.0f : 0.0f;
        fill[t] = row == p ? 1.0f : 0.0f;                                   // the row of ones
    }
    auto load_full = [&](int g, dn_f4 (&xf)[TR]) {                          // g < L / 16: all four columns exist
#pragma unroll
        for (int t = 0; t < TR; t++) xf[t] = *(const dn_f4u *) (rowp[t] + 16 * g);
    };
    auto products = [&](const dn_f4 (&xf)[TR]) {
#pragma unroll
        for (int m = 0; m < 4; m++) {
            double xd[TR];
#pragma unroll
            for (int t = 0; t < TR; t++) xd[t] = (double) fmaf(xf[t][m], keep[t], fill[t]);   // counts are exact in fp32: x * 1 + 0, or 0 * x + {0, 1}
            int tix = 0;
#pragma unroll
            for (int t1 = 0; t1 < TR; t1++)
#pragma unroll
                for (int t2 = 0; t2 <= t1; t2++, tix++)
                    acc[tix] = __builtin_amdgcn_mfma_f64_16x16x4f64(xd[t1], xd[t2], acc[tix], 0, 0, 0);
        }
    };
    const int nfull = L / 16;
    dn_f4 xa[TR], xb[TR];
    int g = w;
    if (g < nfull) load_full(g, xa);
#pragma clang loop unroll(disable)
    for (; g < nfull; g += 2 * W) {                                          // ping-pong: the next group's loads fly during the products
        const bool more = g + W < nfull;
        if (more) load_full(g + W, xb);
        products(xa);
        if (!more) break;
        if (g + 2 * W < nfull) load_full(g + 2 * W, xa);
        products(xb);
    }
    if ((L & 15) && w == nfull % W) {                                        // the partial last group: guarded element loads
        dn_f4 xt[TR];
        const int c = 16 * nfull + 4 * lk;
#pragma unroll
        for (int t = 0; t < TR; t++) {
            const float *src = rowp[t] + 16 * nfull;
#pragma unroll
            for (int m = 0; m < 4; m++) xt[t][m] = (c + m < L) ? src[m] : 0.0f;
        }
        // columns beyond L must contribute nothing, also to the row of ones
        float keep_t[4];
#pragma unroll
        for (int m = 0; m < 4; m++) keep_t[m] = (c + m < L) ? 1.0f : 0.0f;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            double xd[TR];
#pragma unroll
            for (int t = 0; t < TR; t++) xd[t] = (double) (fmaf(xt[t][m], keep[t], fill[t]) * keep_t[m]);
            int tix = 0;
#pragma unroll
            for (int t1 = 0; t1 < TR; t1++)
#pragma unroll
                for (int t2 = 0; t2 <= t1; t2++, tix++)
                    acc[tix] = __builtin_amdgcn_mfma_f64_16x16x4f64(xd[t1], xd[t2], acc[tix], 0, 0, 0);
        }
    }
    // tile (t1, t2): register r of lane (c = l & 15, q = l >> 4) holds D[16 t1 + q + 4 r][16 t2 + c]
    for (int ww = 0; ww < W; ww++) {
        if (w == ww) {
            int tix = 0;
#pragma unroll
            for (int t1 = 0; t1 < TR; t1++)
#pragma unroll
                for (int t2 = 0; t2 <= t1; t2++, tix++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = 16 * t1 + lk + 4 * r, col = 16 * t2 + li;
                        const double v = acc[tix][r];
                        if (ww == 0) g_mg[row * MG_LD + col] = v;
                        else g_mg[row * MG_LD + col] += v;
                    }
        }
        __syncthreads();
    }
    // the tiles below the diagonal were accumulated once; their mirror images are copied now (the solver walks whole rows)
    constexpr int R = 16 * TR;
    for (int idx = threadIdx.x; idx < R * R; idx += NT) {
        const int r = idx / R, c = idx - r * R;
        if ((c >> 4) > (r >> 4)) g_mg[r * MG_LD + c] = g_mg[c * MG_LD + r];
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------
// Pass 1, integer-exact, on the i8 matrix cores (round 3).  Coverage is whole-number counts; when every count of the gene fits
// 16 bits (k_row_max's x16 flag) a count is x = 256 h + l with two bytes, stored offset by 128 as SIGNED bytes h', l', and
//     sum_c x_i x_j = 65536 HH_ij + 256 (HL_ij + HL_ji) + LL_ij + K (r_i + r_j) + n K^2,     K = 32 896, r_i = 256 sum_c h'_i + sum_c l'_i
// with three byte Gram matrices accumulated EXACTLY in int32 by v_mfma_i32_16x16x64_i8 (16 cycles for 64 columns of a 16 x 16
// tile; the fp64 form takes 64 cycles for 4 columns: 160 MFMA-cycles per column of a 64-row matrix against 10 here).  p <= 63.  As in the
// fp64 pass which columns share an instruction does not matter, so load q = 0..3 of lane (i = l & 15, kb = l >> 4) takes the four
// counts c0 + 16 q + 4 kb .. + 3 of row 16 t + i: the four lanes of a row read 64 CONTIGUOUS bytes per load instruction (one whole
// sector per row and instruction; the per-lane-contiguous form of round 3 touched 64 sectors for a quarter each), a row's 256 bytes
// per wave in four instructions, and the same registers serve as A and as B operand.  Row p is a row of ones in l' (zeros in h'):
// r_j falls out of the same products.  Every wave takes 64-column groups round-robin; a wave's int32 tiles hold at most
// 2 x 16 384 x (L / 4) -- genes longer than 2^17 bases take the fp64 pass.  The waves' tiles are added up as 64-bit integers in
// LDS (ds_add_u64, any order: exact) where the matrix goes, and mg_finalize_i8 turns the sums into the fp64 matrix (round 4).
// ---------------------------------------------------------------------------------------------------
typedef int dn_int4 __attribute__((ext_vector_type(4)));

// The 64-bit integer sums of the byte Gram matrices (mg_gram_pass_i8 left them where the matrix goes) -> the matrix in fp64: every
// entry converted, the offsets of the byte representation added, the mirror image written.  r_i = 256 sum h'_i + sum l'_i sits in row p
// (the row of ones); x = 256 h' + l' + K.  Wave w takes rows w, w + 4, ..., lane = column; all LDS reads first, one barrier, then the
// writes.  A function of its own (like mg_solve): inside the kernel's body it shares the register allocation of the streaming loops,
// which parks loop-invariant values in scratch and fetches them back here one dependent round trip at a time (19 k cycles per gene).
template <int TR>
__device__ __attribute__((noinline)) void mg_finalize_i8(int L, int p)
{
    constexpr int W = NT / 64;
    constexpr int R = 16 * TR, NK = R / W;
    constexpr double K = 32896.0;
    const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane(wave_id());
    L = __builtin_amdgcn_readfirstlane(L);
    p = __builtin_amdgcn_readfirstlane(p);
    const long long *g_mg64 = reinterpret_cast<const long long *>(g_mg);
    const double nK2 = (double) L * K * K, LK = (double) L * K;
    const int c = lane < R ? lane : R - 1, ct = c >> 4;
    const bool col_live = c < p;
    long long raw[NK], rawr[NK];
    const long long rawc = g_mg64[p * MG_LD + c];
#pragma unroll
    for (int k = 0; k < NK; k++) {
        raw[k] = g_mg64[(w + W * k) * MG_LD + c];
        rawr[k] = g_mg64[p * MG_LD + (w + W * k)];
    }
    __syncthreads();                                                         // every raw sum is in registers: the doubles go where they were
    DN_MG_TS(3);
    const double rc = (double) rawc;
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const int r = w + W * k;                                             // wave-uniform; its tile row is k / (16 / W) whatever w is
        constexpr int dummy = 0; (void) dummy;
        const int rt = (W * k) >> 4;
        // rows below p: offsets; row p: the plain row sums (cov_sums); rows beyond: as they are.  r is uniform: scalar selects
        const double add_live = r < p ? fma(K, (double) rawr[k] + rc, nK2) : (r == p ? LK : 0.0);      // integers below 2^53: exact
        const double v = (double) raw[k] + (col_live ? add_live : 0.0);
        if (lane < R && ct <= rt) {                                          // the tiles on and below the diagonal were accumulated
            g_mg[r * MG_LD + c] = v;
            if (ct < rt) g_mg[c * MG_LD + r] = v;                            // mirror image: the solver walks whole rows
        }
    }
    __syncthreads();
}

#ifndef DN_I8_ROWSEG
#define DN_I8_ROWSEG 1          // 1: a load instruction reads 64 contiguous bytes per row (0: round 3's 64 contiguous bytes per LANE)
#endif
constexpr int I8_LK = DN_I8_ROWSEG ? 4 : 16, I8_Q = DN_I8_ROWSEG ? 16 : 4;     // column offset per lane group / per load
#if defined(DN_PASS1_NT) && DN_PASS1_NT
#define DN_P1_LOAD(ptr) __builtin_nontemporal_load(ptr)
#else
#define DN_P1_LOAD(ptr) (*(ptr))
#endif
#ifndef DN_PASS2_NT
#define DN_PASS2_NT 1           // pass 2 is the last use of the gene's bytes: non-temporal loads (-5 % on the kernel)
#endif
#if DN_PASS2_NT
#define DN_P2_LOAD(ptr) __builtin_nontemporal_load(ptr)
#else
#define DN_P2_LOAD(ptr) (*(ptr))
#endif

template <int TR>
DN_MG_FN void mg_gram_pass_i8(const float *x, int L, int p)
{
    constexpr int NTILE = TR * (TR + 1) / 2;
    constexpr int W = NT / 64;
    const int lane = lane_id(), w = wave_id();
    const int li = lane & 15, lk = lane >> 4;
    dn_int4 hh[NTILE], ll[NTILE], hl[NTILE];
#pragma unroll
    for (int i = 0; i < NTILE; i++) { hh[i] = dn_int4{0, 0, 0, 0}; ll[i] = hh[i]; hl[i] = hh[i]; }
    // the matrix's place in LDS collects the waves' tiles as 64-bit integers first (below): zero it now, while nothing else is ready
    long long *const g_mg64 = reinterpret_cast<long long *>(g_mg);
    for (int idx = threadIdx.x; idx < 16 * TR * MG_LD; idx += NT) g_mg64[idx] = 0ll;
    __syncthreads();
    gF_cptr rowp[TR];
    unsigned keep[TR], ones[TR];
#pragma unroll
    for (int t = 0; t < TR; t++) {
        const int row = 16 * t + li;
        rowp[t] = (gF_cptr) x + (size_t) (row < p ? row : 0) * L + I8_LK * lk;
        keep[t] = row < p ? 0xffffffffu : 0u;                               // padding rows: zero bytes (after the offset)
        ones[t] = row == p ? 0x01010101u : 0u;                              // the row of ones (in l')
    }
    // bytes of four counts -> one dword of low bytes, one of high bytes (both offset by 128: x ^ 0x80 per byte)
    auto pack4 = [](const dn_f4 &v, unsigned &lo, unsigned &hi) {
        const unsigned u0 = (unsigned) v.x, u1 = (unsigned) v.y, u2 = (unsigned) v.z, u3 = (unsigned) v.w;
        const unsigned t01 = __builtin_amdgcn_perm(u1, u0, 0x05010400u);   // u0.b0 u1.b0 u0.b1 u1.b1
        const unsigned t23 = __builtin_amdgcn_perm(u3, u2, 0x05010400u);
        lo = __builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u;
        hi = __builtin_amdgcn_perm(t23, t01, 0x07060302u) ^ 0x80808080u;
    };
    auto products = [&](const dn_int4 (&H)[TR], const dn_int4 (&Lo)[TR]) {
        int tix = 0;
#pragma unroll
        for (int t1 = 0; t1 < TR; t1++)
#pragma unroll
            for (int t2 = 0; t2 <= t1; t2++, tix++) {
                hh[tix] = __builtin_amdgcn_mfma_i32_16x16x64_i8(H[t1], H[t2], hh[tix], 0, 0, 0);
                ll[tix] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Lo[t1], Lo[t2], ll[tix], 0, 0, 0);
                hl[tix] = __builtin_amdgcn_mfma_i32_16x16x64_i8(H[t1], Lo[t2], hl[tix], 0, 0, 0);      // HL + LH: both into one tile
                hl[tix] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Lo[t1], H[t2], hl[tix], 0, 0, 0);
            }
    };
    const int nfull = L / 64;
    // Round 4: the loop is rotated like pass 2's -- the loads of the wave's NEXT group follow the conversion of each row tile into the
    // registers it just freed, so a group's 16 loads are in flight while this one's 40 matrix instructions run (two waves per SIMD:
    // without it a wave's share of the memory pipe stood empty through its conversions and products).  Only the last row tile has
    // rows beyond the samples (padding, the row of ones): the others take their bytes as they are.
    if (w < nfull) {
        dn_f4 raw[TR][4];
#pragma unroll
        for (int t = 0; t < TR; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) raw[t][q] = DN_P1_LOAD((gF4_cptr) (rowp[t] + 64 * w + I8_Q * q));
        auto trip = [&](auto more_c, int gn) {
            constexpr bool MORE = decltype(more_c)::value;
            dn_int4 H[TR], Lo[TR];
#pragma unroll
            for (int t = 0; t < TR; t++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    unsigned lo, hi;
                    pack4(raw[t][q], lo, hi);
                    if (t == TR - 1) { lo = (lo & keep[t]) | ones[t]; hi &= keep[t]; }
                    Lo[t][q] = (int) lo;
                    H[t][q] = (int) hi;
                }
                if constexpr (MORE) {
#pragma unroll
                    for (int q = 0; q < 4; q++) raw[t][q] = DN_P1_LOAD((gF4_cptr) (rowp[t] + 64 * gn + I8_Q * q));
                }
            }
            if constexpr (MORE) __builtin_amdgcn_sched_barrier(0);         // the loads stay in front of the products (the scheduler sinks them to save registers)
            products(H, Lo);
        };
        int g = w;
#pragma clang loop unroll(disable)
        for (; g + W < nfull; g += W) trip(std::true_type{}, g + W);
        trip(std::false_type{}, 0);
    }
    DN_MG_TS(0);
    if ((L & 63) && w == nfull % W) {                                        // the partial last group
        dn_int4 H[TR], Lo[TR];
        if (L >= 64) {
            // the LAST 64 columns of the gene in whole loads, the bytes of the columns the full groups have counted masked out (zero
            // bytes add nothing to any of the sums): round 3 fetched every count of the group with a guarded load of its own, 10 k cycles
            // during which the other three waves waited at the barrier
            const int thr = 64 - (L & 63);                                   // columns thr .. 63 of the window are new
            unsigned m[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int d = thr - (I8_Q * q + I8_LK * lk);                 // elements d .. 3 of this lane's four are new
                m[q] = d <= 0 ? 0xffffffffu : (d >= 4 ? 0u : (0xffffffffu << (8 * d)));
            }
#pragma unroll
            for (int t = 0; t < TR; t++) {
                dn_f4 v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = DN_P1_LOAD((gF4_cptr) (rowp[t] + (L - 64) + I8_Q * q));
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    unsigned lo, hi;
                    pack4(v[q], lo, hi);
                    Lo[t][q] = (int) (((lo & keep[t]) | ones[t]) & m[q]);
                    H[t][q] = (int) (hi & keep[t] & m[q]);
                }
            }
        } else {                                                             // a gene shorter than one group: guarded element loads, columns beyond L are zero bytes
            const int c = 64 * nfull + I8_LK * lk;
#pragma unroll
            for (int t = 0; t < TR; t++) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    dn_f4 v;
                    unsigned m = 0;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int col = c + I8_Q * q + e;
                        v[e] = col < L ? rowp[t][64 * nfull + I8_Q * q + e] : 0.0f;
                        m |= col < L ? (0xffu << (8 * e)) : 0u;
                    }
                    unsigned lo, hi;
                    pack4(v, lo, hi);
                    Lo[t][q] = (int) (((lo & keep[t]) | ones[t]) & m);
                    H[t][q] = (int) (hi & keep[t] & m);
                }
            }
        }
        products(H, Lo);
    }
    DN_MG_TS(1);
    // Round 4: the waves' tiles are added up as 64-BIT INTEGERS in LDS (ds_add_u64, all four waves at once, any order: exact), then
    // one pass turns every entry into its double, adds the offsets and writes the mirror image -- with all its LDS reads up front.
    // Round 3 added the tiles in fp64 one wave after the other (four phases, a barrier each) and ran the offsets and the mirror as
    // loops of one dependent LDS round trip per trip: 67 k cycles per gene between the last column and the solve, 24 % of a gene
    // (stamps of the diagnostic build, profiles/round4/init_phases_*.txt).  Same numbers: sums of integers below 2^53.
    // tile (t1, t2): register r of lane (c = l & 15, q = l >> 4) holds entry [16 t1 + 4 q + r][16 t2 + c]
    {
        int tix = 0;
#pragma unroll
        for (int t1 = 0; t1 < TR; t1++)
#pragma unroll
            for (int t2 = 0; t2 <= t1; t2++, tix++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = 16 * t1 + 4 * lk + r, col = 16 * t2 + li;
                    const long long v = 65536ll * (long long) hh[tix][r] + 256ll * (long long) hl[tix][r] + (long long) ll[tix][r];
                    __hip_atomic_fetch_add(g_mg64 + row * MG_LD + col, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
    }
    __syncthreads();
    DN_MG_TS(2);
    mg_finalize_i8<TR>(L, p);
    DN_MG_TS(4);
}

// Top eigenvector of the p x p block of g_mg (p <= 64) by ONE wave: lane r keeps to row r of the matrix in LDS (row stride
// MG_LD = 81 doubles: the 64 lanes hit different banks), the vectors live in g_mv and are read with broadcast loads, norms and
// the Rayleigh quotient are all-reduced over the lanes in registers -- no barrier inside the iteration (the block-wide form
// this replaces paid two barriers per matrix-vector product plus two block reductions per check: 53 k cycles per gene for 7
// steps).  Shifted power iteration, two plain steps between convergence checks, the stopping rule of top_eig_rows.  The other
// waves wait at the closing barrier; with two workgroups per CU their SIMDs run the neighbour's passes meanwhile.
__device__ __attribute__((noinline)) int mg_solve(int p, int maxs, double *u_out)
{
    int status = ST_OK;
    if (wave_id() == 0) {
        const int lane = lane_id();
        const bool live = lane < p;
        const int r = live ? lane : p - 1;
        const double *Gr = g_mg + r * MG_LD;
        double *uv = g_mv[0], *vv = g_mv[1];
        // entries p .. 63 of the vectors are 0 and the matrix has 16 TR >= p rows / columns of finite numbers: the products run
        // over whole groups of 8 columns, eight LDS reads in flight and four independent sums
        const int p8 = (p + 7) & ~7;
        auto row_dot = [&](const double *vec) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma clang loop unroll(disable)
            for (int j = 0; j < p8; j += 8) {
                double gq[8], vq[8];
#pragma unroll
                for (int k = 0; k < 8; k++) { gq[k] = Gr[j + k]; vq[k] = vec[j + k]; }
                a0 = fma(gq[0], vq[0], a0); a1 = fma(gq[1], vq[1], a1); a2 = fma(gq[2], vq[2], a2); a3 = fma(gq[3], vq[3], a3);
                a0 = fma(gq[4], vq[4], a0); a1 = fma(gq[5], vq[5], a1); a2 = fma(gq[6], vq[6], a2); a3 = fma(gq[7], vq[7], a3);
            }
            return live ? (a0 + a1) + (a2 + a3) : 0.0;
        };
        const double tr = wave_allsum(live ? Gr[r] : 0.0);
        double ul = live ? 1.0 / sqrt((double) p) : 0.0;
        uv[lane] = ul; vv[lane] = 0.0;                                  // lanes >= p hold 0 (MG_ROWS >= 64 entries)
        wave_fence();
        int steps = 1;
        if (!(tr > 0.0)) status = ST_ARPACK;
        else {
            double y = row_dot(uv);                                     // (G u)_l
            const double th = wave_allsum(ul * y);
            if (!(th > 0.0)) status = ST_ARPACK;
            else {
                double mu = (tr - th) / (double) (p > 1 ? p - 1 : 1);
                mu = (mu > 0.0 && mu < 0.5 * th) ? mu : 0.0;
                double vl = fma(-mu, ul, y);                            // first shifted step
                double d2_prev = -1.0;
                for (;;) {
                    const double n2 = wave_allsum(vl * vl);
                    if (!(n2 > 0.0)) { status = ST_ARPACK; break; }
                    const double inv = 1.0 / sqrt(n2);
                    const double un = vl * inv;
                    const double d = un - ul;
                    const double d2 = wave_allsum(d * d);
                    ul = un;
                    wave_fence();
                    uv[lane] = ul;
                    wave_fence();
                    if (d2 <= 1e-26 || (d2_prev > 0.0 && 4.0 * d2 < d2_prev && 4.0 * d2 * d2 <= 1e-26 * d2_prev)) break;
                    if (steps >= maxs) { status = ST_NO_CONVERGENCE; break; }
                    d2_prev = d2;
                    const double wl = fma(-mu, ul, row_dot(uv));        // two plain shifted steps, no normalisation in between
                    vv[lane] = wl;
                    wave_fence();
                    vl = fma(-mu, wl, row_dot(vv));
                    wave_fence();
                    steps += 2;
                }
            }
        }
        if (live) u_out[lane] = ul;
        if (lane == 0) { g_st.steps = steps; g_st.status = status; }
    }
    __syncthreads();
    status = g_st.status;
    __syncthreads();
    return status;
}

// Pass 2 of the initial DI pass: a wave takes 64 columns at a time, lane = column.  The p counts of the column are loaded once
// (p independent 256-byte row segments per wave in flight), s_j = u . x_j is formed from them, and every row keeps a per-lane
// partial sum of max(u_i s_j, x_ij) in registers for the whole gene (RC doubles, RC = p rounded up to 8; with the column: 3 RC
// registers, <= 192);
// the block adds them up once per gene (register reduce-scatter per wave, 32 rows per round).  u is read from LDS with
// broadcast loads.  One read of the gene instead of the two of the row-tiled form this replaces (s_j staged in LDS, rows
// walked in tiles of 8: 27.5 GB more traffic on config 4).
// Round 4 (tools/ubench/strided_rows.hip, tools/init_ab.py): the memory system gives this pair of passes 6.5-7 TB/s of algorithmic
// bytes -- with the arithmetic of both loops in place -- where the kernel's loops reached 4.4; what was missing are loads in flight
// while a wave computes (two waves per SIMD: a wave that issues, waits, then computes for ~3 000 cycles leaves its share of the
// memory pipe empty for that long).  So (i) the loop is ROTATED: the clamped-sum statement of row i is followed by the load of row
// i of the wave's NEXT block into the register it just freed -- a full block of loads in flight through the second half of every
// trip at no register cost; (ii) the blocks are walked from the gene's END: what pass 1 read last is read first, while it still
// sits in the Infinity Cache / L2 (the 512 resident workgroups read ~280 MB between a byte's two uses otherwise), with
// non-temporal loads (the last use of these bytes); (iii) a row's address is one 32-bit vector add away from the previous row's
// (scalar block base + lane offset; the running 64-bit scalar base cost four scalar instructions per load); (iv) the block that
// the gene's end leaves partial is done first, on its own, under the one exec mask it needs.
template <int RC>                                                    // row capacity of this instantiation (p <= RC, p > RC - 8), a multiple of 8
DN_MG_FN void mg_pass2(const float *x_, int L, int p)
{
    constexpr int W = NT / 64;
    const int tid = threadIdx.x, lane = lane_id(), w = __builtin_amdgcn_readfirstlane(wave_id());
    const float *x = uniform_ptr(x_);                                   // block bases stay in scalar registers: loads take saddr + lane offset
    L = __builtin_amdgcn_readfirstlane(L);
    p = __builtin_amdgcn_readfirstlane(p);
    // rows p .. RC - 1 are walked like the others, without a branch: their u is 0 (no part in s_j), they re-read row p - 1
    // (the line is in L2) and their sums are never looked at
    if (tid >= p && tid < RC) g_st.u[tid] = 0.0;
    __syncthreads();
    double acc[RC];
#pragma unroll
    for (int i = 0; i < RC; i++) acc[i] = 0.0;
    typedef const char __attribute__((address_space(1))) *gbyte_cptr;
    const unsigned Lb = 4u * (unsigned) L;                              // row pitch in bytes (a gene's RC rows stay below 4 GB: L <= 2^24, checked at upload)
    // offset of row i + 1 from row i: the pitch, or 0 once the rows run out (only the last seven rows of an instantiation can)
    auto step = [&](int i) -> unsigned { return (i + 1 <= RC - 8 || i + 1 < p) ? Lb : 0u; };
    float xv[RC];
    auto load_block = [&](int k0, unsigned off) {                       // all rows of the 64 columns from k0 (off: this lane's byte offset in the block)
        gbyte_cptr blk = (gbyte_cptr) (x + k0);
        asm volatile("" : "+v"(off));                                   // (the row offsets are formed as the rows are walked: hoisted, they are 2 RC registers)
#pragma unroll
        for (int i = 0; i < RC; i++) { xv[i] = DN_P2_LOAD((gF_cptr) (blk + off)); off += step(i); }
    };
    auto dot = [&]() {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < RC; i += 2) {
            s0 = fma(g_st.u[i], (double) xv[i], s0);
            s1 = fma(g_st.u[i + 1], (double) xv[i + 1], s1);
        }
        return s0 + s1;
    };
    const int nfull = L >> 6;
    if ((L & 63) && w == nfull % W) {                                   // the partial last block: lanes beyond the gene's end re-read its last column and add nothing
        const int k0 = 64 * nfull;
        const bool valid = k0 + lane < L;
        load_block(k0, 4u * (unsigned) (valid ? lane : L - 1 - k0));
        const double sj = dot();
        if (valid) {
#pragma unroll
            for (int i = 0; i < RC; i++) {
                asm volatile("" : "+v"(xv[i]));                         // convert again here: keeping the RC doubles of the dot product alive spills them
                acc[i] += fmax(g_st.u[i] * sj, (double) xv[i]);         // est[est < x] = x   nmf.py:119
            }
        }
    }
    if (w < nfull) {                                                    // the full blocks w, w + W, ... from the last one down
        int b = (nfull - 1 - w) / W * W + w;
        const unsigned off0 = 4u * (unsigned) lane;
        load_block(64 * b, off0);
        // one trip: the sums of block b, and (MORE) the loads of the wave's next block b - W behind them row by row
        auto trip = [&](auto more_c, int k0n) {
            constexpr bool MORE = decltype(more_c)::value;
            const double sj = dot();
            gbyte_cptr blk = (gbyte_cptr) (x + k0n);
            unsigned off = off0;
            asm volatile("" : "+v"(off));
#pragma unroll
            for (int i = 0; i < RC; i++) {
                asm volatile("" : "+v"(xv[i]));
                const double xd = (double) xv[i];
                if constexpr (MORE) { xv[i] = DN_P2_LOAD((gF_cptr) (blk + off)); off += step(i); }
                acc[i] += fmax(g_st.u[i] * sj, xd);                     // est[est < x] = x   nmf.py:119
            }
        };
#pragma clang loop unroll(disable)
        for (; b >= W; b -= W) trip(std::true_type{}, 64 * (b - W));
        trip(std::false_type{}, 0);
    }
#pragma unroll
    for (int i0 = 0; i0 < RC; i0 += 32) {
        constexpr int dummy = 0; (void) dummy;
        double part[32];
#pragma unroll
        for (int r = 0; r < 32; r++) part[r] = i0 + r < RC ? acc[i0 + r < RC ? i0 + r : 0] : 0.0;
        block_sum_lds<32, DN_P, NT, double>(part, g_sm);
        if (tid < 32 && i0 + tid < p) g_st.csum[i0 + tid] = g_sm.tot[tid];
        __syncthreads();
    }
}

__global__ __launch_bounds__(NT, 2) void k_ratio_svd_mg(InitArgs A)
{
    const int tid = threadIdx.x;
    const int p = A.p;
    const int maxs = 5 * (A.max_steps > 0 ? A.max_steps : EIG_MAX_STEPS_DEFAULT);       // plain power steps, like top_singular
    const int TR = (p + 1 + 15) / 16;
#ifdef DN_STAMP
    long long ts_prev = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        if (tid == 0) g_sm.gene = atomicAdd(A.counter, 1);
        __syncthreads();
        const int q = g_sm.gene;
        __syncthreads();
        if (q >= A.n_genes) break;
        const int g = A.order[q];
        const int L = A.glen[g];
        const float *x = A.cov + A.goff[g];
        int status = ST_OK;
#ifdef DN_STAMP
        const long long ts0 = __builtin_amdgcn_s_memtime();
        long long ts1 = ts0, ts2 = ts0;
#endif
        if (L < 2 || L > (1 << 24)) status = ST_VALUE_ERROR;             // (pass 2 addresses a gene's rows by 32-bit byte offsets)
        else {
            const bool bytes_ok = TR <= 4 && A.x16 && A.x16[g] != 0 && L <= (1 << 17) && !A.force_fp64;     // else: the fp64 matrix cores
            if (bytes_ok) {
                if (TR <= 2) mg_gram_pass_i8<2>(x, L, p);
                else if (TR == 3) mg_gram_pass_i8<3>(x, L, p);
                else mg_gram_pass_i8<4>(x, L, p);
            } else if (TR <= 2) mg_gram_pass<2>(x, L, p);
            else if (TR == 3) mg_gram_pass<3>(x, L, p);
            else if (TR == 4) mg_gram_pass<4>(x, L, p);
            else mg_gram_pass<5>(x, L, p);
#ifdef DN_STAMP
            ts1 = __builtin_amdgcn_s_memtime();
#endif
            status = mg_solve(p, maxs, g_st.u);
#ifdef DN_STAMP
            ts2 = __builtin_amdgcn_s_memtime();
#endif
        }
        // pass 2 (nmf.py:117-121, :524-525): est = max(K E, x) summed per row, in ONE more read of the gene.  The plain row
        // sums come from the Gram matrix's row of ones.
        if (tid < p) { g_st.rsum[tid] = g_mg[p * MG_LD + tid]; g_st.csum[tid] = 0.0; }
        __syncthreads();
        if (status == ST_OK) {
            if (p <= 24) mg_pass2<24>(x, L, p);
            else if (p <= 32) mg_pass2<32>(x, L, p);
            else if (p <= 40) mg_pass2<40>(x, L, p);
            else if (p <= 48) mg_pass2<48>(x, L, p);
            else if (p <= 56) mg_pass2<56>(x, L, p);
            else mg_pass2<64>(x, L, p);
        }
        if (tid < p) {
            A.est_sums[(size_t) g * p + tid] = status == ST_OK ? g_st.csum[tid] : 0.0;
            A.cov_sums[(size_t) g * p + tid] = status == ST_OK ? g_st.rsum[tid] : 0.0;
        }
        if (tid == 0) A.status[g] = status;
#ifdef DN_STAMP          // diagnostic build: the first four sums of the gene are REPLACED by cycles of pass 1 / solve / pass 2 and the solver's steps
        if (tid == 0) {
            const long long ts3 = __builtin_amdgcn_s_memtime();
            A.est_sums[(size_t) g * p + 0] = (double) (ts1 - ts0); A.est_sums[(size_t) g * p + 1] = (double) (ts2 - ts1);
            A.est_sums[(size_t) g * p + 2] = (double) (ts3 - ts2); A.est_sums[(size_t) g * p + 3] = (double) g_st.steps;
            // slots 4 .. 9: inside pass 1 (main loop, partial group, tile sums, offsets, mirror) and the time between the genes
            A.est_sums[(size_t) g * p + 4] = (double) (g_mg_ts[0] - ts0);
            for (int k = 1; k < 5; k++) A.est_sums[(size_t) g * p + 4 + k] = (double) (g_mg_ts[k] - g_mg_ts[k - 1]);
            A.est_sums[(size_t) g * p + 9] = (double) (ts0 - ts_prev);
        }
        ts_prev = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
    }
}

__global__ __launch_bounds__(NT) void k_ratio_svd_gen(InitArgs A)
{
    const int tid = threadIdx.x;
    const int p = A.p;
    if (tid == 0) g_st.max_steps = A.max_steps > 0 ? A.max_steps : EIG_MAX_STEPS_DEFAULT;
    __syncthreads();
    for (;;) {
        if (tid == 0) g_sm.gene = atomicAdd(A.counter, 1);
        __syncthreads();
        const int q = g_sm.gene;
        __syncthreads();
        if (q >= A.n_genes) break;
        const int g = A.order[q];
        const int L = A.glen[g];
        const float *x = A.cov + A.goff[g];
        int status = ST_OK;
        if (L < 2) status = ST_VALUE_ERROR;
        else {
            if (tid < p) g_st.u[tid] = 1.0 / sqrt((double) p);
            if (tid == 0) g_st.steps = 0;
            __syncthreads();
            status = top_singular_raw(x, L, p);
        }
        double pe[GP], pc[GP];
#pragma unroll
        for (int i = 0; i < GP; i++) { pe[i] = 0.0; pc[i] = 0.0; }
        if (status == ST_OK) {
            for (int k = tid; k < L; k += NT) {
                float xv[GP];
#pragma unroll
                for (int i = 0; i < GP; i++) xv[i] = i < p ? x[(size_t) i * L + k] : 0.0f;
                double sdot = 0.0;
#pragma unroll
                for (int i = 0; i < GP; i++) if (i < p) sdot = fma(g_st.u[i], (double) xv[i], sdot);
#pragma unroll
                for (int i = 0; i < GP; i++) {
                    if (i < p) {
                        const double v = (double) xv[i];
                        const double ke = g_st.u[i] * sdot;
                        pe[i] += ke < v ? v : ke;                         // nmf.py:119
                        pc[i] += v;
                    }
                }
            }
        }
#pragma unroll
        for (int i0 = 0; i0 < GP; i0 += TI) {
            if (i0 < p) {
                double part[TI];
#pragma unroll
                for (int r = 0; r < TI; r++) part[r] = pe[i0 + r];
                tile_sum(part);
                if (tid < TI && i0 + tid < p) A.est_sums[(size_t) g * p + i0 + tid] = g_sm.tot[tid];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < TI; r++) part[r] = pc[i0 + r];
                tile_sum(part);
                if (tid < TI && i0 + tid < p) A.cov_sums[(size_t) g * p + i0 + tid] = g_sm.tot[tid];
                __syncthreads();
            }
        }
        if (tid == 0) A.status[g] = status;
        __syncthreads();
    }
}

#endif  // DN_GEN_NT == 256

__global__ __launch_bounds__(256) void k_estimates_gen(EstArgs A, const int32_t *__restrict__ tile_gene,
                                                       const int32_t *__restrict__ tile_col0)
{
    const int g = tile_gene[blockIdx.x];
    const int j = tile_col0[blockIdx.x] + threadIdx.x;
    const int L = A.glen[g], p = A.p;
    if (j >= L) return;
    const float *x = A.cov + A.goff[g];
    double *o = A.out + (A.ooff ? A.ooff[g] : A.goff[g]);
    const int em = A.emode[g];
    const double *K = A.kfin + (size_t) g * p;
    if (em == EM_INPUT) {
        for (int i = 0; i < p; i++) o[(size_t) i * L + j] = (double) x[(size_t) i * L + j] / A.scale[i];
    } else if (em == EM_EXPAND || em == EM_REFINED) {
        double m = -INFINITY;
        for (int i = 0; i < p; i++) { const double qv = ((double) x[(size_t) i * L + j] / A.scale[i]) / K[i]; m = qv > m ? qv : m; }
        for (int i = 0; i < p; i++) {
            const double f = (double) x[(size_t) i * L + j] / A.scale[i];
            double v = K[i] * m;
            if (em == EM_EXPAND) v = v < f ? f : v;
            o[(size_t) i * L + j] = v;
        }
    } else {
        const double s = A.svec[A.svoff[g] + j];
        for (int i = 0; i < p; i++) {
            const double f = (double) x[(size_t) i * L + j] / A.scale[i];
            double v = K[i] * s;
            if (em == EM_CLAMPED) v = v < f ? f : v;
            o[(size_t) i * L + j] = v;
        }
    }
}

static int launch_baseline(const IterArgs &a, int grid, size_t, hipStream_t s)
{
    hipLaunchKernelGGL(k_baseline_gen, dim3(grid), dim3(NT), 0, s, a);
    return (int) hipGetLastError();
}
static void launch_est(const EstArgs &a, const int32_t *tg, const int32_t *tc, int n_tiles, hipStream_t s)
{
    hipLaunchKernelGGL(k_estimates_gen, dim3(n_tiles), dim3(256), 0, s, a, tg, tc);
}
#if DN_GEN_NT == 256
static void launch_init(const InitArgs &a, int grid, hipStream_t s)
{
    // 17 <= p <= 64: two passes over the coverage with the Gram matrix on the matrix cores; DN_INIT_POWER=1 keeps the
    // one-pass-per-power-step kernel (cross-check in the tests); below 17 samples only the tests come here (DN_FORCE_GENERIC)
    const char *pw = getenv("DN_INIT_POWER");
    if (a.p >= 17 && !(pw && pw[0] == '1')) hipLaunchKernelGGL(k_ratio_svd_mg, dim3(grid), dim3(NT), 0, s, a);
    else hipLaunchKernelGGL(k_ratio_svd_gen, dim3(grid), dim3(NT), 0, s, a);
}
static int blocks_per_cu(int which)
{
    int nb = 0;
    // which: 0 the iteration kernel, 1 the initial pass by power iteration (p <= 16 here), 2 the initial pass with the Gram
    // matrix on the matrix cores (k_ratio_svd_mg, 17 <= p <= 64: what launch_init starts for those sample counts)
    hipError_t e = which == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_baseline_gen, NT, 0)
                 : which == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_ratio_svd_mg, NT, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_ratio_svd_gen, NT, 0);
    return e == hipSuccess ? nb : 0;
}
#else
// the initial ratio-SVD pass works on whole transcripts: it stays with the 256-thread family
static void launch_init(const InitArgs &a, int grid, hipStream_t s) { kernel_set_generic()->init(a, grid, s); }
static int blocks_per_cu(int which)
{
    if (which != 0) return kernel_set_generic()->blocks_per_cu(which);
    int nb = 0;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_baseline_gen, NT, 0) == hipSuccess ? nb : 0;
}
#endif

}  // namespace gen / gen_rows

#if DN_GEN_NT == 256
const KernelSet *kernel_set_generic()
{
    static const KernelSet ks = {
        0, gen::NT, gen::launch_baseline, gen::launch_init, gen::launch_est, gen::blocks_per_cu,
        (size_t) 160 * 1024,               // no LDS tier: "static" covers the CU so that lds_cols comes out 0
        "gen::k_baseline_gen",          // as rocprofv3 prints it (dn::gen::k_baseline_gen)
        0, 0, 1,
    };
    return &ks;
}
#else
const KernelSet *kernel_set_rows()
{
    static const KernelSet ks = {
        0, gen_rows::NT, gen_rows::launch_baseline, gen_rows::launch_init, gen_rows::launch_est, gen_rows::blocks_per_cu,
        (size_t) 160 * 1024,
        "gen_rows::k_baseline_gen",     // dn::gen_rows::k_baseline_gen
        0, 0, 1,
    };
    return &ks;
}
#endif

}  // namespace dn
