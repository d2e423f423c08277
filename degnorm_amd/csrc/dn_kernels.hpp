// dn_kernels.hpp -- hand-written HIP kernels (gfx950 / CDNA4) for DegNorm's NMF over-approximation hot path.
//
// What runs here (reference lines relative to the DegNorm checkout):
//   k_baseline<P,NT>   adjust_coverage_curves + baseline_selection per gene   nmf.py:142-146, :189-372
//                      (-> nmf() :78-107 -> rank_one_approx :55-64, get_high_coverage_idx :66-76,
//                       split_into_chunks utils.py:176-192, shift_bins nmf.py:160-187)
//   k_ratio_svd<P,NT>  ratio_svd + row sums                                    nmf.py:109-121, :524-525
//   k_estimates<P>     the `estimate` output of baseline_selection             nmf.py:327-365
//
// Mapping: one workgroup of NT threads (NT / 64 wavefronts) owns one gene at a time and walks the whole
// baseline-selection state machine for it; workgroups are persistent and pull genes from a work queue.  (The DN_PAIR build
// gives every wavefront of a 128-thread workgroup its own gene: "workgroup" below then reads "wavefront".)  Columns (base
// positions) are spread over lanes, the p samples of a column live in one lane's registers, so every global / LDS
// access is lane-contiguous.
//
// The rank-1 SVD: est = K E = u u^T (x + lambda) is column-local once u is known, so one pass per inner NMF-OA
// iteration updates the state a = x + lambda AND accumulates the p x p Gram matrix of the *next* a in registers (fp64).
// The per-lane Gram partials are reduce-scattered inside each wave in registers (dn_reduce.hpp: v_permlane swaps + DPP,
// fixed order => deterministic, no atomics), added across waves through LDS, and the top eigenvector of the p x p
// matrix comes from repeated squaring on the fp64 matrix cores (top_eig_mfma, p <= 16; v_mfma_f64_16x16x4_f64) or a
// row-distributed shifted power iteration (p > 16), to fp64 round-off like the reference's ARPACK call with tol = 0.
// For p >= DN_MG_MIN_P the Gram matrix itself is accumulated on the matrix cores (mg_core).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "dn_reduce.hpp"
#include "dn_dpp_ops.hpp"

// Measured dead ends are recorded in DESIGN.md (section 4, "tried and not kept"), not kept here as compile-time branches:
// non-temporal spill accesses, fp32 Gram accumulators, raw-count-units state, a lane-major tier save area, the register
// tier as a masked loop for genes that fill it.
// Diagnostic ISA builds (-DDN_MARKS, tools/isa_regions.py): named comment lines in the generated assembly, so that the
// instructions of each phase of an inner iteration can be counted statically.  Never defined for the product.
#ifdef DN_MARKS
#define DN_MARK(name) asm volatile("; DN_MARK " name)
#else
#define DN_MARK(name)
#endif

namespace dn {

constexpr int TRACE_LEN = 48;
constexpr int MAX_BINS = 64;
constexpr int P_MAX = 64;      // largest sample count any kernel accepts (templated and run-time-p kernels alike)

enum { EXIT_LOW_COV = 0, EXIT_ZERO_SAMPLE = 1, EXIT_MEDIAN = 2, EXIT_NO_LOOP = 3,
       EXIT_REFINED = 4, EXIT_REFINE_FALLBACK = 5, EXIT_NOT_FOUND_FALLBACK = 6 };
enum { LOOP_NATURAL = 0, LOOP_PERFECT = 1, LOOP_VALUE_ERROR = 2, LOOP_ZERO_ROWSUM = 3, LOOP_MIN_BINS = 4,
       LOOP_NOT_ENTERED = 5 };
enum { ST_OK = 0, ST_ARPACK = -1, ST_EMPTY_MIN = -2, ST_VALUE_ERROR = -3,
       ST_NO_CONVERGENCE = -4 };     // an eigen-solve left through its step cap (ARPACK would raise ArpackNoConvergence)
constexpr int EIG_MAX_STEPS_DEFAULT = 4000;
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
    const float   *rowmax;     // n x p: max_j x[i][j] of the raw coverage (k_row_max, once per upload)
    const int32_t *x16;        // n: every count of the gene is a whole number <= 65535 (the register tier may pack them)
    char          *ws;         // scratch: slots x slot_bytes
    double        *rho;        // n x p
    int32_t       *flags;      // n
    int32_t       *trace;      // n x TRACE_LEN
    double        *kfin;       // n x p   (estimates support)
    int32_t       *emode;      // n
    double        *svec;       // per gene s_start vectors (only when want_est), at svoff[g]
    const int64_t *svoff;
    int64_t        slot_bytes;
    int32_t        n_genes;
    int32_t        S;          // column stride of the scratch arrays (>= longest gene, multiple of 64)
    int32_t        lds_cols;   // lambda columns held in LDS (dynamic shared memory = 8 * p * lds_cols bytes)
    int32_t        T, bins, min_hc, rate, skip, want_est;
    int32_t        p;          // samples (run-time copy; the templated kernels take it from the template)
    int32_t        max_steps;  // step cap of one eigen-solve (power-step equivalents); beyond it the gene gets ST_NO_CONVERGENCE
    double         scale[P_MAX];
    double         inv_scale[P_MAX];
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
    int32_t        p;
    int32_t        max_steps;
    char          *ws;         // scratch slots (generic kernels only)
    int64_t        slot_bytes;
    int32_t        S;
    const int32_t *x16;        // n: every count of the gene is a whole number <= 65535 (k_row_max): the integer-exact Gram pass may run
    int32_t        force_fp64; // diagnostics (DN_INIT_FP64=1): keep the fp64 Gram pass also where the integer-exact one applies
};

struct EstArgs {
    const float   *cov;
    const int64_t *goff;
    const int32_t *glen;
    const double  *kfin;
    const int32_t *emode;
    const double  *svec;
    const int64_t *svoff;
    double        *out;        // float64; gene g at ooff[g] (or at goff[g], the coverage offsets, when ooff is null)
    const int64_t *ooff;
    int32_t        n_genes;
    int32_t        p;
    double         scale[P_MAX];
};

// ---------------------------------------------------------------------------------------------------
// LDS layout of one workgroup (static part; the lambda tile is dynamic shared memory behind it).
// ---------------------------------------------------------------------------------------------------
// From this sample count on the Gram matrix is accumulated by the fp64 matrix cores (mg_core below) instead of in
// per-lane registers swept several times over the state.
#ifndef DN_MG_MIN_P
#define DN_MG_MIN_P 25
#endif

template <int P, int NT>
struct Smem {
    static constexpr int W = NT / 64;
    static constexpr int MG_TR = (P + 15) / 16;                 // 16-row tiles covering the samples
    static constexpr int MG_STR = 16 * MG_TR;                   // doubles per staged column (zero-padded rows)
    static constexpr int NG = P * (P + 1) / 2;
    static constexpr int NX = NG >= 64 ? NG + 2 : 64;
    static constexpr int ZSLOT = NX - 1;     // tot[ZSLOT] is written once (0.0): the load target of padding lanes
    double xw[W][NX];                        // per-wave totals (cross-wave combine)
    double tot[NX];                          // block totals (broadcast)
    double dsel[NX];                         // 1.0 at the packed indices of the Gram diagonal, else 0.0
    // round 4 (top_eig_dpp): the block total of the Gram matrix once more as a SQUARE matrix, rows contiguous and 16-byte aligned,
    // already multiplied by the solver's scale: lane r of the solver reads its row with 128-bit loads from ONE address (the packed
    // triangle needs a gather of p addresses held in p registers through the whole T loop).  Rows >= p stay zero.
    // Every WAVE builds its own copy from the per-wave totals (one barrier per reduction instead of two: nobody waits for a
    // combining wave), which is why xw is double-buffered for these reductions (xw2: the T loop alternates between the two).
    static constexpr int SQ_STR = P + (P & 1);
    static constexpr int SQ_LEN = 16 * SQ_STR;
    alignas(16) double sq[P <= 16 ? W * SQ_LEN : 2];
    double xw2[(P <= 16 && W > 1) ? W : 1][(P <= 16 && W > 1) ? NX : 1];
    int32_t sqi[P <= 16 ? NX : 2];           // packed entry e = (a, b): (a SQ_STR + b) | (b SQ_STR + a) << 16
    double stage[P >= DN_MG_MIN_P ? W * 16 * MG_STR : 2];      // per wave: 16 updated columns in the MFMA operand layout
    double eigv[P >= DN_MG_MIN_P ? W * 2 * 64 : 2];            // per wave: current eigenvector u and a work vector
    double ss[MAX_BINS];                     // per-bin mean squared residual
    int32_t alive[MAX_BINS];                 // original ids of the surviving bins, in order
    int32_t cnt[W];                          // per-wave hi-coverage counts
    int32_t gene;                            // current queue item
};

// DN_PAIR: a 128-thread workgroup carries TWO genes, one per wavefront ("unit").  Everything the kernels call a workgroup --
// thread index, barriers, the LDS objects, the scratch slot -- is then a wavefront's: the template runs with NT = 64 and the
// two units never synchronise with each other (they walk different genes through different control flow).  One wavefront
// per gene pays reduce + eigen-solve on one SIMD and needs no cross-wave step; pairing two of them in a workgroup of the
// narrow class's shape (threads, LDS block) keeps the CU's LDS free of the holes that 64-thread workgroups leave between
// 128-thread ones (third gene class).
#ifdef DN_PAIR
#define DN_UNITS 2
#define DN_TIDX ((int) (threadIdx.x & 63))
#define DN_UNIT ((int) (threadIdx.x >> 6))
#else
#define DN_UNITS 1
#define DN_TIDX ((int) threadIdx.x)
#define DN_UNIT 0
#endif
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return DN_UNITS > 1 ? 0 : (int) (threadIdx.x >> 6); }   // wave within the unit

// LDS traffic inside one wave needs no s_barrier: a wave's DS instructions execute in order.  This only stops
// the compiler from moving LDS accesses across the point.
__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// barrier of the unit: the workgroup's s_barrier, or -- one wavefront per unit -- just the ordering of its LDS traffic
__device__ __forceinline__ void dn_sync()
{
    if constexpr (DN_UNITS > 1) wave_fence(); else __syncthreads();
}

template <typename T>
__device__ __forceinline__ T *uniform_ptr(T *q)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) a), hi = __builtin_amdgcn_readfirstlane((unsigned) (a >> 32));
    return reinterpret_cast<T *>(((unsigned long long) hi << 32) | lo);
}

__device__ __forceinline__ double uniform(double v)
{
    // value is identical in every lane: move it to scalar registers so it costs no VGPRs
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Block-wide sum of N register values; the totals are left in sm.tot[0..N), bit-identical for every reader
// (the wave-uniform branches of the state machine rely on that).  Per wave the values are reduce-scattered in
// registers (dn_reduce.hpp: lane l ends up with the wave total of entry bitrev6(l)), one LDS write per lane hands
// them to the cross-wave add.
// One round of the per-wave reduction: entries [OFF, OFF + CNT) of g, CNT <= 64, totals into dst[OFF + entry].
// the block total of a packed Gram entry as the solver wants it: minus the shift on the diagonal (dsel: 1 there, else 0); with the
// Gram matrix accumulated in raw count units (RAWSC) dsel carries the entry's scale 1 / (s_a s_b), negative on the diagonal
template <bool RAWSC>
__device__ __forceinline__ double shift_total(double s, double d, double diag_shift)
{
    if constexpr (RAWSC) return fma(s, fabs(d), d < 0.0 ? -diag_shift : 0.0);
    else return fma(-diag_shift, d, s);
}

// the square, scaled copy of a shifted block total for top_eig_dpp (Smem::sq): both triangles
__device__ __forceinline__ void sq_store(double *sq, int packed_idx, double t)
{
    sq[packed_idx & 0xffff] = t;
    sq[packed_idx >> 16] = t;
}

template <int N, int OFF, typename VT, bool RAWSC = false, bool SQ = false>
__device__ __forceinline__ void wave_round_store(const VT (&g)[N], double *dst, int lane, double diag_shift, const double *dsel,
                                                 bool shift_here, double *sq = nullptr, const int32_t *sqi = nullptr, double sq_scale = 1.0)
{
    constexpr int CNT = (N - OFF) < 64 ? (N - OFF) : 64;
    VT part[CNT];
#pragma unroll
    for (int i = 0; i < CNT; i++) part[i] = g[OFF + i];
    double s = wave_reduce_scatter<CNT, VT>(part, lane);
    const int e = reduce_scatter_entry(lane);
    if (e < CNT) {
        if (shift_here) s = shift_total<RAWSC>(s, dsel[OFF + e], diag_shift);
        dst[OFF + e] = s;
        if constexpr (SQ) { if (shift_here) sq_store(sq, sqi[OFF + e], s * sq_scale); }
    }
    if constexpr (OFF + 64 < N) wave_round_store<N, OFF + 64, VT, RAWSC, SQ>(g, dst, lane, diag_shift, dsel, shift_here, sq, sqi, sq_scale);
}

template <int N, int P, int NT, typename VT, bool SHIFT = false, int OFF = 0, bool RAWSC = false, bool SQ = false>    // OFF: the totals go to tot[OFF .. OFF + N)
__device__ __forceinline__ void block_sum_lds(const VT (&g)[N], Smem<P, NT> &sm, double diag_shift = 0.0, double sq_scale = 1.0, int parity = 0)
{
    static_assert(!RAWSC || SHIFT, "the raw-unit scale rides on the shift step");
    static_assert(!SQ || (SHIFT && P <= 16), "the square copy is the eigen-solver's view of the shifted Gram matrix");
    static_assert(OFF + N <= Smem<P, NT>::NX - 1, "xw too small");
    static_assert(!SHIFT || OFF + N <= P * (P + 1) / 2, "the shift is for the packed Gram matrix");
    constexpr int W = NT / 64;
    const int lane = lane_id(), w = wave_id();
    if constexpr (SQ && W > 1) {
        // The eigen-solver's input only (T loop): per-wave totals into the xw buffer of this call's parity, ONE barrier, then every
        // wave adds the W totals of each entry itself (same order in every wave: same bits) and writes its own square, scaled copy.
        // The next reduction writes the other buffer, so a wave still reading here is never overtaken; sm.tot is not written.
        constexpr int NXW = Smem<P, NT>::NX;
        double *xwb = parity ? &sm.xw2[0][0] : &sm.xw[0][0];
        wave_round_store<N, 0, VT, RAWSC, false>(g, xwb + w * NXW + OFF, lane, diag_shift, sm.dsel + OFF, false);
        DN_MARK("wave_reduced");
        __syncthreads();
        double *sqw = sm.sq + w * Smem<P, NT>::SQ_LEN;
        for (int e = OFF + lane; e < OFF + N; e += 64) {            // one trip unless N > 64
            double t = xwb[e];
#pragma unroll
            for (int ww = 1; ww < W; ww++) t += xwb[ww * NXW + e];
            t = shift_total<RAWSC>(t, sm.dsel[e], diag_shift);
            sq_store(sqw, sm.sqi[e], t * sq_scale);
        }
        wave_fence();
        return;
    }
    double *dst = ((W > 1) ? sm.xw[w] : sm.tot) + OFF;
    wave_round_store<N, 0, VT, RAWSC, SQ>(g, dst, lane, diag_shift, sm.dsel + OFF, SHIFT && W == 1, sm.sq, sm.sqi + OFF, sq_scale);     // ceil(N / 64) rounds
    DN_MARK("wave_reduced");
    if constexpr (W > 1) {
        __syncthreads();
        for (int e = OFF + DN_TIDX; e < OFF + N; e += NT) {         // one trip unless N > NT
            double t = sm.xw[0][e];
#pragma unroll
            for (int ww = 1; ww < W; ww++) t += sm.xw[ww][e];
            if constexpr (SHIFT) t = shift_total<RAWSC>(t, sm.dsel[e], diag_shift);  // G - mu I for the eigen-solver
            sm.tot[e] = t;
        }
    }
    dn_sync();
    // totals are in sm.tot[0..N); no trailing barrier: the next writer of tot / xw sits behind the next
    // call's first barrier
}

template <int N, int P, int NT>
__device__ __forceinline__ void block_sum(double (&g)[N], Smem<P, NT> &sm)
{
    block_sum_lds<N, P, NT, double>(g, sm);
#pragma unroll
    for (int i = 0; i < N; i++) g[i] = sm.tot[i];
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

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, I1)
template <int I, int I1, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < I1) { f(std::integral_constant<int, I>{}); static_for<I + 1, I1>(f); }
}

// ---------------------------------------------------------------------------------------------------
// Top eigenpair of the symmetric PSD p x p matrix G (packed lower triangle, idx(i,j) = i(i+1)/2 + j).
// Shifted power iteration, warm-started from u, run by every lane on identical data until
// ||u_new - u||^2 <= 1e-24 (the step after that shrinks the error by the convergence ratio again, so u is at
// fp64 round-off like the reference's ARPACK call with tol = 0).  The shift mu = mean of the non-dominant
// eigenvalues ~ (trace - theta) / (p - 1) moves the noise cluster to ~0 and never slows convergence while
// mu <= lambda_2.  Entries of x + lambda are non-negative, so the Perron vector is non-negative and a
// positive start is never orthogonal to it.  Returns the number of steps; theta = u^T G u (= sigma^2).
// ---------------------------------------------------------------------------------------------------
template <int P>
__device__ __forceinline__ void sym_matvec(const double (&G)[P * (P + 1) / 2], const double (&u)[P], double (&y)[P])
{
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
}

template <int P>
__device__ __forceinline__ int top_eig(const double (&G)[P * (P + 1) / 2], double (&u)[P], double &theta, int maxs = EIG_MAX_STEPS_DEFAULT)
{
    double tr = 0.0;
#pragma unroll
    for (int i = 0; i < P; i++) tr += G[i * (i + 1) / 2 + i];
    double y[P];
    sym_matvec<P>(G, u, y);
    double th = 0.0;
#pragma unroll
    for (int i = 0; i < P; i++) th = fma(u[i], y[i], th);
    int steps = 1;
    for (;;) {
        // y = G u is current; th = u.y
        double mu = (tr - th) / (double) (P > 1 ? P - 1 : 1);
        mu = (mu > 0.0 && mu < 0.5 * th) ? mu : 0.0;
        double n2 = 0.0;
#pragma unroll
        for (int i = 0; i < P; i++) { y[i] = fma(-mu, u[i], y[i]); n2 = fma(y[i], y[i], n2); }
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
        sym_matvec<P>(G, u, y);
        th = 0.0;
#pragma unroll
        for (int i = 0; i < P; i++) th = fma(u[i], y[i], th);
        steps++;
        if (d2 <= 1e-24) break;
        if (steps >= maxs) { steps = maxs + 1; break; }          // > maxs tells the caller the cap was hit
    }
    theta = th;
    return steps;
}

// The hot loop finds the same eigenpair through the fp64 matrix cores: repeated squaring instead of single power steps.
// v_mfma_f64_16x16x4_f64 (gfx950, 64 cycles) takes A as lane (i = l & 15, k = l >> 4) -> A[i][k], B as lane
// (j, k) -> B[k][j] and returns D[q + 4 r][j] in register r of lane (j, q = l >> 4).  For a symmetric matrix
// held as h[kb] = H[l & 15][(l >> 4) + 4 kb] the product H H = sum_kb mfma(h[kb], h[kb]) therefore comes back in
// the layout it went in, so M squarings cost M * ceil(p / 4) MFMAs and no data movement, and one step with
// H = ((G - mu I) / tr)^(2^M) equals 2^M shifted power steps.  The iterate lives in the B layout (register kb of
// lane group q holds v[q + 4 kb], every column alike), which is also what the step returns; norms are all-reduced
// over the four 16-lane rows with the permlane swaps, so every lane sees identical bits.  The shift uses the
// eigenvalue of the previous solve (theta on entry, 0 on a cold start).  theta = u^T G u of the unshifted matrix.
typedef double dn_double4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double allsum_rows(double v)
{
    v = swap32_add(v, v);
    return swap16_add(v, v);
}

#ifndef DN_EIG_SQUARINGS
#define DN_EIG_SQUARINGS 2
#endif

__device__ __forceinline__ double rsqrt_newton(double n2)
{
    // 1/sqrt(n2): hardware estimate + two Newton steps
    double inv = __builtin_amdgcn_rsq(n2);
    inv = inv * fma(-0.5 * n2 * inv, inv, 1.5);
    inv = inv * fma(-0.5 * n2 * inv, inv, 1.5);
    return inv;
}

__device__ __forceinline__ double rcp_newton(double x)
{
    // 1 / x for x >= 1: hardware estimate + two Newton steps (error ~1 ulp; an IEEE division costs 15 instructions, and the
    // residual profile it feeds is compared between bins whose values differ by far more)
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    y = fma(fma(-x, y, 1.0), y, y);
    return y;
}

template <int KB>
__device__ __forceinline__ dn_double4 mfma_sym(const double (&h)[KB], const double (&v)[KB])
{
    dn_double4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kb = 0; kb < KB; kb++) y = __builtin_amdgcn_mfma_f64_16x16x4f64(h[kb], v[kb], y, 0, 0, 0);
    return y;
}

// State carried from one solve to the next inside an nmf() call: the iterate in the MFMA B layout (no rebuild
// from the broadcast u), and the scale / shift the next solve starts from (they only steer convergence).
template <int P>
struct EigState {
    static constexpr int KB = (P + 3) / 4;
    double v[KB];        // v[kb] of lane group q = u[q + 4 kb]
    double sc;           // ~ 1 / trace(G)
    double mu;           // shift in the units of G (0: none); the caller subtracts it from the diagonal in LDS
};

template <int P>
__device__ __forceinline__ void eig_state_cold(EigState<P> &st, double tr)
{
    const int q = lane_id() >> 4;
    const double u0 = 1.0 / sqrt((double) P);
#pragma unroll
    for (int kb = 0; kb < EigState<P>::KB; kb++) st.v[kb] = (q + 4 * kb < P) ? u0 : 0.0;
    st.sc = __builtin_amdgcn_rcp(tr);
    st.mu = 0.0;
}

// tot holds G - st.mu I (packed, plus a zero at zslot).  On return u is the unit top eigenvector (broadcast),
// theta the eigenvalue of G: exact (Rayleigh quotient) when asked for, else an estimate that only feeds the next shift.
template <int P>
__device__ __forceinline__ int top_eig_mfma(const double *tot, int zslot, double (&u)[P], double &theta, EigState<P> &st,
                                            bool exact_theta, int maxs = EIG_MAX_STEPS_DEFAULT)
{
    static_assert(P <= 16, "one 16 x 16 MFMA tile");
    constexpr int KB = EigState<P>::KB;
    constexpr int M = DN_EIG_SQUARINGS;
    const int c = lane_id() & 15, q = lane_id() >> 4;
    const double sc = st.sc, mu = st.mu;
    double g[KB], h[KB], v[KB];
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
        const int k = q + 4 * kb;
        const int a = c > k ? c : k, b = c > k ? k : c;
        g[kb] = tot[a < P ? a * (a + 1) / 2 + b : zslot] * sc;          // rows / columns >= P are zero and stay zero
        h[kb] = g[kb];
        v[kb] = st.v[kb];
    }
    DN_MARK("solver_loaded");
#pragma unroll
    for (int m = 0; m < M; m++) {
        const dn_double4 d = mfma_sym<KB>(h, h);
#pragma unroll
        for (int kb = 0; kb < KB; kb++) h[kb] = d[kb];
    }
    // two steps back to back; the norm of the first iterate is formed in the shadow of the second product
    const dn_double4 y1 = mfma_sym<KB>(h, v);
    double y1r[KB];
#pragma unroll
    for (int kb = 0; kb < KB; kb++) y1r[kb] = y1[kb];
    const dn_double4 y2 = mfma_sym<KB>(h, y1r);
    // trace of G for the next solve's scale and shift: independent of the products above
    double dg[P];
#pragma unroll
    for (int i = 0; i < P; i++) dg[i] = tot[i * (i + 1) / 2 + i];
#pragma unroll
    for (int w = 1; w < P; w *= 2) {
#pragma unroll
        for (int i = 0; i + w < P; i += 2 * w) dg[i] += dg[i + w];
    }
    const double tr = fma((double) P, mu, dg[0]);
    double p1 = 0.0, p2 = 0.0;
#pragma unroll
    for (int kb = 0; kb < KB; kb++) { p1 = fma(y1[kb], y1[kb], p1); p2 = fma(y2[kb], y2[kb], p2); }
    const double n1 = allsum_rows(p1), n2 = allsum_rows(p2);
    if (!(n1 > 0.0) || !(n2 > 0.0)) { theta = 0.0; return 1; }
    const double i1 = rsqrt_newton(n1), i2 = rsqrt_newton(n2);
    double e1p = 0.0, e2p = 0.0;
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
        const double u1 = y1[kb] * i1, u2 = y2[kb] * i2;
        const double da = u1 - v[kb], db = u2 - u1;
        e1p = fma(da, da, e1p); e2p = fma(db, db, e2p);
        v[kb] = u2;
    }
    double d2_prev = allsum_rows(e1p), d2 = allsum_rows(e2p);
    DN_MARK("solver_checked");
    double n_last = n2 * i2, i_prev = i1;                               // |y2| and 1 / |y1|: y2 = H y1
    int steps = 1 + (2 << M);
    // as in top_eig_rows: stop when the error predicted from the contraction between two checks is ~1e-13
    // (ratio = d2 / d2_prev < 0.25 and 4 d2 ratio <= 1e-26, written without the division)
    bool conv;
    while (!(conv = (d2 <= 1e-26 || (4.0 * d2 < d2_prev && 4.0 * d2 * d2 <= 1e-26 * d2_prev))) && steps < maxs) {
        const dn_double4 y = mfma_sym<KB>(h, v);
        double part = 0.0;
#pragma unroll
        for (int kb = 0; kb < KB; kb++) part = fma(y[kb], y[kb], part);
        const double nn = allsum_rows(part);
        if (!(nn > 0.0)) { theta = 0.0; return steps; }
        const double inv = rsqrt_newton(nn);
        n_last = nn * inv; i_prev = 1.0;                               // v was a unit vector here
        double dpart = 0.0;
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
            const double un = y[kb] * inv;
            const double d = un - v[kb];
            dpart = fma(d, d, dpart);
            v[kb] = un;
        }
        d2_prev = d2;
        d2 = allsum_rows(dpart);
        steps += 1 << M;
    }
#pragma unroll
    for (int i = 0; i < P; i++) {
        const double t = v[i >> 2];
        u[i] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t), 16 * (i & 3)),
                                __builtin_amdgcn_readlane(__double2loint(t), 16 * (i & 3)));
    }
    if (exact_theta) {
        const dn_double4 w = mfma_sym<KB>(g, v);                       // Rayleigh quotient: v^T (G - mu I) v / |v|^2 + mu
        double tp = 0.0;
#pragma unroll
        for (int kb = 0; kb < KB; kb++) tp = fma(v[kb], w[kb], tp);
        theta = allsum_rows(tp) / sc + mu;
    } else {
        // dominant eigenvalue of H ~ |H y| / |y|, and H = ((G - mu I) sc)^(2^M)
        double lam = n_last * i_prev;
#pragma unroll
        for (int m = 0; m < M; m++) lam = __builtin_amdgcn_sqrt(lam);
        theta = fma(lam, __builtin_amdgcn_rcp(sc), mu);
    }
#pragma unroll
    for (int kb = 0; kb < KB; kb++) st.v[kb] = v[kb];
    st.sc = __builtin_amdgcn_rcp(tr);
    double mn = (tr - theta) * (1.0 / (double) (P > 1 ? P - 1 : 1));
    st.mu = (theta > 0.0 && mn > 0.0 && mn < 0.5 * theta) ? mn : 0.0;
    return conv ? (steps < maxs ? steps + 1 : maxs) : maxs + 1;         // > maxs: left through the step cap
}

// ---------------------------------------------------------------------------------------------------
// Round 4: the same eigenpair by a warm-started, shifted power iteration on the vector pipe, lane = row, with the iterate
// broadcast INSIDE the multiply: gfx90a+ has DPP forms of two fp64 instructions, v_fmac_f64 and v_mov_b64, restricted to
// row_newbcast:j ("every lane of a 16-lane row reads lane j of the row").  With row r of H = (G - mu I) sc in the registers of
// lane r (every 16-lane row of the wave keeps its own copy of the matrix, p <= 16), one power step is p instructions
//     y_r += bcast_j(v) * H[r][j]          v_fmac_f64_dpp acc, v, h_j row_newbcast:j          (dn_dpp_ops.hpp, generated)
// and leaves y distributed as v was -- no v_readlane, no LDS, no matrix instruction; a dot product over the rows costs the same p
// instructions (bcast_j(a b) * 1.0).  What decides the design (tools/ubench/solver_instr_cost.hip, solver_ab.hip; profiles/round4):
// at one wave per SIMD EVERY instruction -- vector, scalar, s_nop -- takes a 4-cycle issue slot (v_fmac_f64_dpp too: 4.08 cycles,
// also as a dependent chain; fp64 rcp / rsq / sqrt 16; v_permlane32_swap 8), so a solve is priced by its instruction COUNT.  The
// squaring solver of rounds 1-3 spends 12 chained 64-cycle MFMAs + ~220 vector + ~100 scalar instructions whatever the matrix
// (2 286 cycles warm); config 2's matrices need 5.0 plain power steps per warm solve on average (2 ... 10; numpy restatement of the
// T loop with this stopping rule), i.e. ~5 x 12 instructions of stepping.
// Schedule of a warm solve: k0 "blind" steps without normalisation (k0 follows what the previous solves of the call needed),
// normalise (u_a), one step, normalise (u_b) and measure d2 = |u_b - u_a|^2 ~ |e_a|^2.  u_b's error is rho |e_a| with rho the
// contraction ratio of the shifted matrix; rho^2 is MEASURED whenever a solve looks twice (d2 of consecutive looks: the cold solve of
// a call always does, a solve that had to continue does, and every 8th solve is made to by looking one step early) and carried to the
// solves in between; stop at 4 d2 rho^2 <= 1e-26, i.e. ~1e-13 like the other solvers, the 4 being the margin for rho's drift.
// Shift and scale (mu = mean of the non-dominant eigenvalues, sc = 1 / (theta - mu)) are set by the call's cold solve and kept for
// its T warm solves: they only steer convergence, and refreshing them costs a trace reduction per solve.
// Every lane of every wave runs the same instructions on the same bits: the result is wave-uniform.
// ---------------------------------------------------------------------------------------------------
template <int P>
struct EigStateD {
    double vl;           // component (lane & 15) of the iterate (unit vector; 0 in the lanes of rows >= P)
    double sc;           // ~ 1 / (theta - mu): un-normalised steps keep the iterate's length
    double mu;           // shift in the units of G (0: none); the caller subtracts it from the diagonal in LDS
    double rho2;         // contraction of d2 per step, as last measured (1: not yet)
    double q;            // min(1, 4 rho2): a look has converged when d2 q <= 1e-26
    int k0;              // blind steps of the next solve
    int age;             // solves since rho2 was measured
};

template <int P>
__device__ __forceinline__ void eig_state_cold(EigStateD<P> &st, double tr)
{
    st.vl = (lane_id() & 15) < P ? 1.0 / sqrt((double) P) : 0.0;
    st.sc = __builtin_amdgcn_rcp(tr);
    st.mu = 0.0;
    st.rho2 = 1.0;
    st.q = 1.0;
    st.k0 = 0;
    st.age = 0;
}

// a condition every lane evaluates on identical bits, as a scalar (the branch on it is s_cbranch_scc, not an exec mask)
__device__ __forceinline__ bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }

template <int P>
__device__ __forceinline__ double bcast_lane(double t, int i)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t), i), __builtin_amdgcn_readlane(__double2loint(t), i));
}

constexpr int EIG_RHO_MAX_AGE = 8;

// Same contract as top_eig_mfma: tot holds G - st.mu I (packed, plus a zero at zslot); u the unit top eigenvector (broadcast),
// theta the eigenvalue of G -- the exact Rayleigh quotient when asked for (the last solve of a call), an estimate after a cold
// solve (it sets shift and scale), untouched otherwise.  COLD: the first solve of an nmf() call.
// ss (diagnostics, tools/ubench/solver_ab.hip): cycles per phase { load, blind steps, first normalisation, looks, epilogue }
template <int P, bool STAMP = false, bool SQUARE = false>
__device__ __forceinline__ int top_eig_dpp(const double *tot, int zslot, double (&u)[P], double &theta, EigStateD<P> &st,
                                           bool exact_theta, bool cold, int maxs = EIG_MAX_STEPS_DEFAULT, long long *ss = nullptr,
                                           const double *sq = nullptr)
{
    static_assert(P <= 16, "one row of the matrix per lane of a 16-lane DPP row");
    long long t_last = 0;
    if constexpr (STAMP) t_last = __builtin_amdgcn_s_memtime();
    auto stamp = [&](int slot) {
        if constexpr (STAMP) {
            __builtin_amdgcn_sched_barrier(0);                             // the clock read stays where it is written
            const long long t = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
            ss[slot] += t - t_last; t_last = t;
        }
    };
    const int r = lane_id() & 15;
    const double sc = st.sc, mu = st.mu;
    double Hr[P];
    if constexpr (SQUARE) {
        // the caller's reduction left (G - mu I) sc as a square matrix (Smem::sq): this lane's row in 128-bit loads from one address
        constexpr int STR = P + (P & 1);
        const double *row = sq + r * STR;
#pragma unroll
        for (int j = 0; j + 1 < P; j += 2) { const double2 h2 = *reinterpret_cast<const double2 *>(row + j); Hr[j] = h2.x; Hr[j + 1] = h2.y; }
        if constexpr (P & 1) Hr[P - 1] = row[P - 1];
    } else {
#pragma unroll
        for (int j = 0; j < P; j++) {
            const int a = r > j ? r : j, b = r > j ? j : r;
            Hr[j] = tot[r < P ? a * (a + 1) / 2 + b : zslot] * sc;        // rows >= P are zero and stay zero
        }
    }
    DN_MARK("solver_loaded");
    double v = st.vl;
    const int k0 = __builtin_amdgcn_readfirstlane(st.k0);
    const int age = __builtin_amdgcn_readfirstlane(st.age);
    const bool probe = age >= EIG_RHO_MAX_AGE && k0 > 0;                  // look one step early, and twice: rho^2 is measured again
    const int kb = k0 - (probe ? 1 : 0);
    int steps = kb;
    stamp(0);
    // the length stays ~1: sc ~ 1 / (theta - mu).  Two steps per trip: a taken branch costs a wave alone on its SIMD ~10 issue slots
    if (kb & 1) v = dpp_matvec<P>(Hr, v);
#pragma clang loop unroll(disable)
    for (int k = kb >> 1; k > 0; k--) v = dpp_matvec<P>(Hr, dpp_matvec<P>(Hr, v));
    stamp(1);
    // (a zero matrix -- a sample block without coverage in solve_by_blocks -- gives NaN from here on: no look converges on NaN, and
    // the slow path below sorts it out; the common path carries no test for it)
    double ua = v * rsqrt_newton(dpp_rowdot<P>(v, v));
    double rho2 = st.rho2;
    double d2 = 0.0, ylen = 1.0, m2 = 1.0;
    // one look: a step from the unit vector u_a, the new unit vector, the squared distance between the two
    auto look = [&]() {
        const double y = dpp_matvec<P>(Hr, ua);
        steps++;
        m2 = dpp_rowdot<P>(y, y);
        const double inv = rsqrt_newton(m2);
        ylen = m2 * inv;                                                   // |H u_a| ~ (theta - mu) sc
        const double ub = y * inv;
        const double d = ub - ua;
        d2 = dpp_rowdot<P>(d, d);
        ua = ub;
    };
    stamp(2);
    look();
    // the common case is over here: converged by the carried contraction (no short-circuit logic: one compare, one scalar branch)
    double q = st.q;
    bool conv = wave_any(d2 * q <= 1e-26);
    int extra = 0;                                                         // looks beyond the first
    if (__builtin_expect(!conv || probe, 0)) {
        if (!wave_any(m2 > 0.0)) { theta = 0.0; return steps; }           // H v = 0: nothing to find (theta = 0 tells the caller)
        while (steps < maxs) {
            const double d2_prev = d2;
            extra++;
            look();
            if (!wave_any(m2 > 0.0)) { theta = 0.0; return steps; }
            const double m = d2 * __builtin_amdgcn_rcp(d2_prev);           // the contraction, measured; the floor keeps a difference
            rho2 = m > 1e-8 ? m : 1e-8;                                    // of round-off from passing for fast convergence
            q = 4.0 * rho2 < 1.0 ? 4.0 * rho2 : 1.0;
            conv = wave_any(d2 * q <= 1e-26);
            if (conv) break;
        }
    }
    stamp(3);
    DN_MARK("solver_checked");
#pragma unroll
    for (int i = 0; i < P; i++) u[i] = bcast_lane<P>(ua, i);
    if (__builtin_expect(exact_theta, 0)) theta = dpp_rowdot<P>(dpp_matvec<P>(Hr, ua), ua) / sc + mu;     // Rayleigh quotient u^T (G - mu I) u + mu
    else if (__builtin_expect(cold, 0)) theta = fma(ylen, __builtin_amdgcn_rcp(sc), mu);
    if (__builtin_expect(cold, 0)) {                                       // shift and scale of this call's warm solves
        const double dgl = tot[r < P ? r * (r + 1) / 2 + r : zslot];
        const double tr = fma((double) P, mu, dpp_rowdot<P>(dgl, 1.0));
        const double mn = (tr - theta) * (1.0 / (double) (P > 1 ? P - 1 : 1));
        const double mu_n = (theta > 0.0 && mn > 0.0 && mn < 0.5 * theta) ? mn : 0.0;
        st.mu = mu_n;
        st.sc = __builtin_amdgcn_rcp(theta - mu_n);
    }
    st.vl = ua;
    st.rho2 = rho2;
    st.q = q;
    st.age = extra > 0 ? 0 : age + 1;
    // the next solve takes one blind step less when this one converged with a whole step to spare (4 d2_(k-1) rho^2 = 4 d2 <= 1e-26,
    // with a margin of 4), and as many more as this one needed looks beyond its first
    {
        const bool spare = extra == 0 && wave_any(16.0 * d2 <= 1e-26);
        const int kn = kb + extra - (spare ? 1 : 0);
        st.k0 = kn < 0 ? 0 : (kn > 24 ? 24 : kn);
    }
    stamp(4);
    return conv ? (steps < maxs ? steps + 1 : maxs) : maxs + 1;           // > maxs: left through the step cap
}

// Sample counts above 16 (more than one MFMA tile): shifted power iteration with the matrix distributed by rows --
// lane l keeps row l of G, computes one component of G v per step, v_readlane broadcasts the p components.  Two
// unnormalised steps between convergence checks; the check predicts the current error from the contraction between
// consecutive checks and stops at ~1e-13.  Warm-started from u.  All waves run it redundantly on identical data.
template <int P>
__device__ __forceinline__ void bcast_rows(double y, double (&yb)[P])
{
    const int lo = __double2loint(y), hi = __double2hiint(y);
#pragma unroll
    for (int j = 0; j < P; j++)
        yb[j] = __hiloint2double(__builtin_amdgcn_readlane(hi, j), __builtin_amdgcn_readlane(lo, j));
}

template <int P>
__device__ __forceinline__ double row_dot(const double (&Gr)[P], const double (&v)[P])
{
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int j = 0; j + 1 < P; j += 2) { a0 = fma(Gr[j], v[j], a0); a1 = fma(Gr[j + 1], v[j + 1], a1); }
    if (P & 1) a0 = fma(Gr[P - 1], v[P - 1], a0);
    return a0 + a1;
}

template <int P>
__device__ __forceinline__ int top_eig_rows(const double *tot, double (&u)[P], double &theta, int maxs = EIG_MAX_STEPS_DEFAULT)
{
    static_assert(P <= 64, "one row per lane");
    const int r = lane_id() < P ? lane_id() : P - 1;
    double Gr[P];
#pragma unroll
    for (int j = 0; j < P; j++) {
        const int a = r > j ? r : j, b = r > j ? j : r;
        Gr[j] = tot[a * (a + 1) / 2 + b];
    }
    double tr = 0.0;
#pragma unroll
    for (int i = 0; i < P; i++) tr += tot[i * (i + 1) / 2 + i];

    double v[P], w[P];
    bcast_rows<P>(row_dot<P>(Gr, u), w);                               // w = G u
    double th = 0.0;
#pragma unroll
    for (int i = 0; i < P; i++) th = fma(u[i], w[i], th);
    if (!(th > 0.0)) { theta = 0.0; return 1; }
    double mu = (tr - th) * (1.0 / (double) (P > 1 ? P - 1 : 1));
    mu = (mu > 0.0 && mu < 0.5 * th) ? mu : 0.0;
#pragma unroll
    for (int j = 0; j < P; j++) Gr[j] = (j == r) ? Gr[j] - mu : Gr[j];  // G - mu I
#pragma unroll
    for (int i = 0; i < P; i++) v[i] = fma(-mu, u[i], w[i]);             // first shifted step
    int steps = 1;
    bool capped = false;
    double d2_prev = -1.0;
    for (;;) {
        double n2 = 0.0;
#pragma unroll
        for (int i = 0; i < P; i++) n2 = fma(v[i], v[i], n2);
        if (!(n2 > 0.0)) { theta = 0.0; return steps; }
        const double inv = rsqrt_newton(n2);
        double d2 = 0.0;
#pragma unroll
        for (int i = 0; i < P; i++) {
            const double un = v[i] * inv;
            const double d = un - u[i];
            d2 = fma(d, d, d2);
            u[i] = un;
        }
        if (d2 <= 1e-26 || (d2_prev > 0.0 && 4.0 * d2 < d2_prev && 4.0 * d2 * d2 <= 1e-26 * d2_prev)) break;
        if (steps >= maxs) { capped = true; break; }
        d2_prev = d2;
        bcast_rows<P>(row_dot<P>(Gr, u), w);                           // two plain steps, no normalisation in between
        bcast_rows<P>(row_dot<P>(Gr, w), v);
        steps += 2;
    }
    bcast_rows<P>(row_dot<P>(Gr, u), w);                               // Rayleigh quotient of the unshifted matrix
    th = mu;
#pragma unroll
    for (int i = 0; i < P; i++) th = fma(u[i], w[i], th);
    theta = th;
    return capped ? maxs + 1 : (steps < maxs ? steps + 1 : maxs);
}

// The same iteration for wide cohorts: lane l keeps row l of G in registers (p doubles) and ITS component of the
// vectors; the vectors live in wave-private LDS (uv: current eigenvector, warm start of the next solve; vv: work vector)
// and are read back with broadcast loads, so nothing of size p sits in registers besides the row.  Norms and the
// Rayleigh quotient are all-reduced over the 64 lanes in registers (identical bits in every lane).
__device__ __forceinline__ double wave_allsum(double v)
{
    v += dpp_mov<DPP_QX1>(v);
    v += dpp_mov<DPP_QX2>(v);
    v += dpp_mov<0x141>(v);          // row_half_mirror
    v += dpp_mov<0x140>(v);          // row_mirror
    return allsum_rows(v);
}

template <int P>
__device__ __forceinline__ double lds_row_dot(const double (&Gr)[P], const double *vec)
{
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int j = 0; j + 1 < P; j += 2) {
        const double2 v2 = *reinterpret_cast<const double2 *>(vec + j);
        a0 = fma(Gr[j], v2.x, a0); a1 = fma(Gr[j + 1], v2.y, a1);
    }
    if (P & 1) a0 = fma(Gr[P - 1], vec[P - 1], a0);
    return a0 + a1;
}

template <int P>
__device__ __forceinline__ int top_eig_rows_lds(const double *tot, double *uv, double *vv, double &theta, int maxs = EIG_MAX_STEPS_DEFAULT)
{
    static_assert(P <= 64, "one row per lane");
    const int lane = lane_id();
    const bool live = lane < P;
    const int r = live ? lane : P - 1;
    double Gr[P];
#pragma unroll
    for (int j = 0; j < P; j++) {
        const int a = r > j ? r : j, b = r > j ? j : r;
        Gr[j] = live ? tot[a * (a + 1) / 2 + b] : 0.0;
    }
    const double tr = wave_allsum(live ? tot[r * (r + 1) / 2 + r] : 0.0);
    double ul = live ? uv[r] : 0.0;
    double y = lds_row_dot<P>(Gr, uv);                                  // (G u)_l
    double th = wave_allsum(ul * y);
    if (!(th > 0.0)) { theta = 0.0; return 1; }
    double mu = (tr - th) * (1.0 / (double) (P > 1 ? P - 1 : 1));
    mu = (mu > 0.0 && mu < 0.5 * th) ? mu : 0.0;
    double vl = fma(-mu, ul, y);                                        // first shifted step
    int steps = 1;
    bool capped = false;
    double d2_prev = -1.0;
    for (;;) {
        const double n2 = wave_allsum(vl * vl);
        if (!(n2 > 0.0)) { theta = 0.0; return steps; }
        const double inv = rsqrt_newton(n2);
        const double un = vl * inv;
        const double d = un - ul;
        const double d2 = wave_allsum(d * d);
        ul = un;
        wave_fence();
        if (live) uv[r] = ul;
        wave_fence();
        if (d2 <= 1e-26 || (d2_prev > 0.0 && 4.0 * d2 < d2_prev && 4.0 * d2 * d2 <= 1e-26 * d2_prev)) break;
        if (steps >= maxs) { capped = true; break; }
        d2_prev = d2;
        const double wl = fma(-mu, ul, lds_row_dot<P>(Gr, uv));        // two plain shifted steps, no normalisation in between
        if (live) vv[r] = wl;
        wave_fence();
        vl = fma(-mu, wl, lds_row_dot<P>(Gr, vv));
        wave_fence();
        steps += 2;
    }
    y = lds_row_dot<P>(Gr, uv);                                         // Rayleigh quotient of the unshifted matrix
    theta = wave_allsum(ul * y);
    return capped ? maxs + 1 : (steps < maxs ? steps + 1 : maxs);
}

// A warm start is only safe while the previous eigenvector overlaps the next one.  When the active matrix falls apart into
// blocks of samples whose supports share no column (sparse, low-count genes: x + lambda keeps the zero pattern of x, so the Gram
// matrix stays block-diagonal through all T iterations), the top eigenvector lives on ONE block, its components on the others
// are round-off (<= 1e-13 after a converged solve), and when another block's singular value overtakes -- it does, lambda grows
// fastest where K E is zero -- a power iteration started from the old vector sees no change above its stopping threshold and
// stays on the wrong block (the reference's ARPACK starts from a random vector every time).  The same holds for blocks coupled
// so weakly that the components are tiny.  Test, once per nmf() call ("warm_start_unsafe"): any component of the eigenvector
// below 1e-7 -> the call is done by the safe path, which solves block by block (solve_by_blocks below).  A stuck iterate keeps
// its tiny components, so the test catches it wherever in the call it got stuck; it costs such genes only.
constexpr double WARM_START_MIN_COMPONENT = 1e-7;
// One interface over the two solvers: the MFMA squaring solver with its carried state for p <= 16, the row-distributed
// power iteration above it.
template <int P, bool MFMA = (P <= 16)> struct Solver;
#ifndef DN_SOLVER_DPP
#define DN_SOLVER_DPP 1          // 1: top_eig_dpp (round 4), 0: top_eig_mfma (rounds 1-3)
#endif
template <int P> struct Solver<P, true> {
#if DN_SOLVER_DPP
    EigStateD<P> st;
    // component i of the iterate sits in lane i (of every 16-lane row)
    __device__ __forceinline__ double iterate_of_row() const { return st.vl; }
    static __device__ __forceinline__ int row_of_lane() { return lane_id() & 15; }
    static __device__ __forceinline__ double bcast_component(double t, int i) { return bcast_lane<P>(t, i); }
#else
    EigState<P> st;
#endif
    __device__ __forceinline__ void cold(double tr, double (&u)[P]) { (void) u; eig_state_cold<P>(st, tr); }
    // the next solve starts from the normalised indicator vector of the rows in `rows` (scale and shift stay): solve_by_blocks
    __device__ __forceinline__ void start_from(unsigned long long rows, double (&u)[P])
    {
        (void) u;
        const double u0 = 1.0 / sqrt((double) __builtin_popcountll(rows));
#if DN_SOLVER_DPP
        st.vl = ((rows >> (lane_id() & 15)) & 1ull) ? u0 : 0.0;
        st.k0 = 0; st.rho2 = 1.0; st.q = 1.0; st.age = 0;    // nothing is known about this start: look after every step
#else
        const int q = lane_id() >> 4;
#pragma unroll
        for (int kb = 0; kb < EigState<P>::KB; kb++) st.v[kb] = ((rows >> (q + 4 * kb)) & 1ull) ? u0 : 0.0;
#endif
    }
    // the shift of the carried state is taken from the top block's eigenvalue: mu < theta / 2 keeps the TOP of the matrix on top,
    // but inside another block it can put the bottom of that block's spectrum on top -- block-by-block solves run unshifted
    __device__ __forceinline__ void no_shift()
    {
        st.mu = 0.0;
    }
    __device__ __forceinline__ double shift() const { return st.mu; }
    __device__ __forceinline__ double scale() const { return st.sc; }
    static constexpr bool SHIFTED = true;
    static constexpr bool SQUARE = DN_SOLVER_DPP != 0;                  // takes the scaled square copy of the matrix
    // cold: the first solve of an nmf() call (it may set what the call's warm solves keep: shift, scale)
    // sq: the matrix once more as the scaled square copy a block_sum_lds<..., SQ> left (Smem::sq), or null
    template <bool SQ = false>
    __device__ __forceinline__ int run(const double *tot, int zslot, double (&u)[P], double &theta, bool exact, int maxs, bool cold = false,
                                       const double *sq = nullptr)
    {
#if DN_SOLVER_DPP
        return top_eig_dpp<P, false, SQ>(tot, zslot, u, theta, st, exact, cold, maxs, nullptr, sq);
#else
        (void) cold; (void) sq;
        return top_eig_mfma<P>(tot, zslot, u, theta, st, exact, maxs);
#endif
    }
};
template <int P> struct Solver<P, false> {
    __device__ __forceinline__ void cold(double tr, double (&u)[P])
    {
        (void) tr;
        const double u0 = 1.0 / sqrt((double) P);
#pragma unroll
        for (int i = 0; i < P; i++) u[i] = u0;
    }
    __device__ __forceinline__ void start_from(unsigned long long rows, double (&u)[P])
    {
        const double u0 = 1.0 / sqrt((double) __builtin_popcountll(rows));
#pragma unroll
        for (int i = 0; i < P; i++) u[i] = ((rows >> i) & 1ull) ? u0 : 0.0;
    }
    __device__ __forceinline__ void no_shift() {}                     // top_eig_rows derives its shift from the start vector's own quotient
    __device__ __forceinline__ double shift() const { return 0.0; }
    __device__ __forceinline__ double scale() const { return 1.0; }
    static constexpr bool SHIFTED = false;
    static constexpr bool SQUARE = false;
    template <bool SQ = false>
    __device__ __forceinline__ int run(const double *tot, int zslot, double (&u)[P], double &theta, bool exact, int maxs, bool cold = false,
                                       const double *sq = nullptr)
    { (void) zslot; (void) exact; (void) cold; (void) sq; return top_eig_rows<P>(tot, u, theta, maxs); }
};

// The block-by-block solve of the safe path.  Rows i, j of the Gram matrix are linked when G_ij > 0 (sums of non-negative
// products): the connected components are the blocks of samples whose supports share columns, the matrix is block-diagonal
// over them, and its top eigenpair is the best of the blocks' top eigenpairs.  Each block is solved from ITS indicator vector
// (the iterate then never leaves the block: products with exact zeros), so the race between two blocks whose singular values
// the iteration itself drives towards each other no longer decides convergence; components outside the winning block are exact
// zeros.  tot: packed lower triangle in LDS; only off-diagonal entries are read here (the diagonal may carry the shift).
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long x)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) x), hi = __builtin_amdgcn_readfirstlane((unsigned) (x >> 32));
    return ((unsigned long long) hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long gram_component(const double *tot, int seed, int nrows)
{
    const int lane = lane_id();
    unsigned long long comp = 1ull << seed, frontier = comp;
    while (frontier) {
        const int i = __builtin_ctzll(frontier);
        frontier &= frontier - 1ull;
        const int j = lane < nrows ? lane : 0;
        const int a = i > j ? i : j, b = i > j ? j : i;
        const bool link = lane < nrows && j != i && tot[a * (a + 1) / 2 + b] > 0.0;
        const unsigned long long fresh = uniform_u64(__ballot(link)) & ~comp;
        comp |= fresh;
        frontier |= fresh;
    }
    return comp;
}

template <int P>
__device__ __forceinline__ int solve_by_blocks(Solver<P> &solver, const double *tot, int zslot, double (&u)[P], double &theta,
                                               int maxs, int nrows = P)
{
    unsigned long long todo = nrows >= 64 ? ~0ull : ((1ull << nrows) - 1ull);
    const Solver<P> entry = solver;                      // scale and shift of this solve (the shift is already inside tot)
    Solver<P> best = solver;
    double bu[P], bth = -1.0;
#pragma unroll
    for (int i = 0; i < P; i++) bu[i] = 0.0;
    int steps = 0;
    bool capped = false;
    while (todo) {
        const unsigned long long comp = gram_component(tot, __builtin_ctzll(todo), nrows);
        todo &= ~comp;
        Solver<P> s = entry;
        double uu[P], th = 0.0;
#pragma unroll
        for (int i = 0; i < P; i++) uu[i] = 0.0;
        s.start_from(comp, uu);
        const int r = s.run(tot, zslot, uu, th, true, maxs);
        steps += r < maxs ? r : maxs;
        capped = capped || r > maxs;
        if (th > bth) {
            bth = th;
            best = s;
#pragma unroll
            for (int i = 0; i < P; i++) bu[i] = uu[i];
        }
    }
    solver = best;
    solver.no_shift();                                   // the next Gram matrix of this call is solved block by block again
#pragma unroll
    for (int i = 0; i < P; i++) u[i] = bu[i];
    theta = bth > 0.0 ? bth : 0.0;
    return capped ? maxs + 1 : (steps < maxs ? steps : maxs);
}

typedef double gram_t;         // per-lane Gram accumulators

// Packed entries [LO, LO + CNT) of the Gram update only (a sweep of a Gram matrix too large for one register set).
template <int P, int LO, int CNT, typename T>
__device__ __forceinline__ void gram_add_range(T (&G)[CNT], const double (&a)[P])
{
    T b[P];
#pragma unroll
    for (int i = 0; i < P; i++) b[i] = (T) a[i];
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            constexpr int dummy = 0; (void) dummy;
            const int idx = i * (i + 1) / 2 + j;
            if (idx >= LO && idx < LO + CNT) G[idx - LO] = fma(b[i], b[j], G[idx - LO]);
        }
}

// G = a a^T on the packed entries [LO, LO + CNT): the first column of a pass STARTS the accumulators
template <int P, int LO, int CNT, typename T>
__device__ __forceinline__ void gram_set_range(T (&G)[CNT], const double (&a)[P])
{
    T b[P];
#pragma unroll
    for (int i = 0; i < P; i++) b[i] = (T) a[i];
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) {
            const int idx = i * (i + 1) / 2 + j;
            if (idx >= LO && idx < LO + CNT) G[idx - LO] = b[i] * b[j];
        }
}

template <int P, typename T>
__device__ __forceinline__ void gram_add(T (&G)[P * (P + 1) / 2], const double (&a)[P])
{
    T b[P];
#pragma unroll
    for (int i = 0; i < P; i++) b[i] = (T) a[i];
#pragma unroll
    for (int i = 0; i < P; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) G[i * (i + 1) / 2 + j] = fma(b[i], b[j], G[i * (i + 1) / 2 + j]);
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

template <int P>
struct GeneState {
    double sumF[P];      // row sums of F_start                                   nmf.py:241, :337
    double rho[P];       // current DI vector
    double K[P];         // current K = u * sigma
    double us[P];        // u of the first nmf() call (K_start / sigma)           nmf.py:250
    double rho_fb[P];    // DI of max(K_start E_start, F_start)                   nmf.py:345-346, :352-353
    double sig0;         // sigma of the first call
    double inv[P];       // 1 / scale factors
    double scl[P];       // the scale factors again (1 / inv): u_i s_i of the raw-unit pass (DN_RAW_UNITS)
    double u[P];         // outputs of the last nmf() call: top left singular vector,
    double theta;        //   sigma^2,
    double sums[2 * P + 1];   // { sum_j s_j, clamped row sums (P), row sums of Fb (P) }
    int32_t steps;       //   power-iteration steps spent (accumulated over the gene's calls)
    int32_t status;
    int32_t max_steps;   // step cap of one eigen-solve (IterArgs::max_steps)
    long long stamp[6];  // diagnostic builds (DN_STAMP): cycles in pass / reduction / eigen-solver / whole nmf() calls / final pass / cold start
};

// The workgroup's LDS objects live at namespace scope so that the out-of-line nmf_call() addresses them with
// ds_* instructions (a reference parameter would decay to a flat pointer).  One translation unit = one (p, NT).
#ifdef DN_P
__shared__ Smem<DN_P, DN_NT> g_sm_u[DN_UNITS];
__shared__ GeneState<DN_P> g_gs_u[DN_UNITS];
extern __shared__ __attribute__((aligned(16))) double g_lam[];      // lambda LDS tier: [units][lds_cols][p]
#define g_sm g_sm_u[DN_UNIT]
#define g_gs g_gs_u[DN_UNIT]
typedef const float __attribute__((address_space(1))) *gF_cptr;      // compacted raw counts (fp32, exact)
typedef double __attribute__((address_space(1))) *gdouble_ptr;

// ---------------------------------------------------------------------------------------------------
// One nmf() call on the compacted working matrix (p x n) -- nmf.py:78-107.
//
// Where the data lives.  Fb: the raw fp32 counts of the n active columns (column-contiguous, p floats per column) in
// the workgroup's scratch slot -- read-only inside the call, ~40 B per column per pass, served by the XCD's L2 /
// Infinity Cache because the slot is touched by this CU only.  F = x * (1/s_i) is formed in registers.  The state
// (fp64, read AND written every pass) never leaves the chip when the gene fits: column k of lane tid = k % NT sits
//     k <  lds_cols         in the dynamic LDS tile lam[k * PS + i]                           "LDS tier"
//     else                  in the slot's global spill array, spill_ptr(Lg, k)[i * 64]        "spill tier"
// (a register-resident tier in front of these was measured slower and is gone: DESIGN.md, dead ends)
// On return u, theta describe the last SVD; sums[] = { sum_j s_j, clamped row sums (P), row sums of Fb (P) };
// rs[k] = squared relative residual of column k (nmf.py:280-282), sv[k] = s_k when `first`.
// ---------------------------------------------------------------------------------------------------
// The on-chip state of a column is a = x + lambda rather than lambda itself:
//     lambda' = max(lambda - c (K E - x), 0)   (nmf.py:94-96)   <=>   a' = x + lambda' = max(a - c (u s - x), x),
// with s = u . a (E_j sigma), so a pass costs 3 fp64 ops per element plus the p(p+1)/2 Gram products and
// x + lambda is never re-formed (nmf.py:97).  lambda is not needed by itself anywhere.
template <int P>
__device__ __forceinline__ void col_update(const double (&f)[P], double (&a)[P], const double (&u)[P], const double (&ur)[P], double c)
{
    // u: the coefficients of the dot product, ur: those of the residual -- the same vector, except in raw count units (DN_RAW_UNITS:
    // u_i / s_i and u_i s_i)
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int i = 0; i + 1 < P; i += 2) { s0 = fma(u[i], a[i], s0); s1 = fma(u[i + 1], a[i + 1], s1); }
    if (P & 1) s0 = fma(u[P - 1], a[P - 1], s0);
    const double s = s0 + s1;                                          // E_j * sigma = u . (x + lambda)_j
#pragma unroll
    for (int i = 0; i < P; i++) {
        const double res = fma(ur[i], s, -f[i]);                       // est - x                       nmf.py:94
        a[i] = fmax(fma(-c, res, a[i]), f[i]);                         // x + max(lambda - c res, 0)    nmf.py:95-97
    }
}

// Wide cohorts: u and 1/s come from LDS (broadcast loads) and F = x / s is formed element by element, so that next to
// the state only the raw counts of the column are live (at p = 64: 128 + 64 registers instead of 4 x 128).
template <int P>
__device__ __forceinline__ void col_update_lds(const float (&x)[P], const double *invp, double (&a)[P], const double *uv, double c)
{
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int i = 0; i + 1 < P; i += 2) {
        const double2 u2 = *reinterpret_cast<const double2 *>(uv + i);
        s0 = fma(u2.x, a[i], s0); s1 = fma(u2.y, a[i + 1], s1);
    }
    if (P & 1) s0 = fma(uv[P - 1], a[P - 1], s0);
    const double s = s0 + s1;
#pragma unroll
    for (int i = 0; i < P; i++) {
        const double fi = (double) x[i] * invp[i];
        const double res = fma(uv[i], s, -fi);
        a[i] = fmax(fma(-c, res, a[i]), fi);
    }
}

template <int P>
__device__ __forceinline__ void col_final(const double (&f)[P], const double (&a)[P], const double (&u)[P], bool first,
                                          double (&acc)[2 * P + 1], double &s_out, double &r_out)
{
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < P; i++) s = fma(u[i], a[i], s);
    acc[0] += s;
    double rmax = 0.0;
#pragma unroll
    for (int i = 0; i < P; i++) {
        const double ke = u[i] * s;
        acc[1 + i] += ke < f[i] ? f[i] : ke;                           // KE[KE < F] = F     nmf.py:318
        acc[1 + P + i] += f[i];
        double d = ke - f[i];
        if (!first) d = d < 0.0 ? 0.0 : d;                             // residual of the clamped KE on later trips
        const double r = d * rcp_newton(f[i] + 1.0);                   // (KE - F) / (F + 1) nmf.py:282 (denominator >= 1)
        const double r2 = r * r;
        rmax = r2 > rmax ? r2 : rmax;
    }
    s_out = s; r_out = rmax;
}

template <int PS>
__device__ __forceinline__ void lds_col_read(const double *col, double (&a)[PS])
{
#pragma unroll
    for (int i = 0; i < PS; i += 2) { const double2 v = *reinterpret_cast<const double2 *>(col + i); a[i] = v.x; a[i + 1] = v.y; }
}
template <int PS>
__device__ __forceinline__ void lds_col_write(double *col, const double (&a)[PS])
{
#pragma unroll
    for (int i = 0; i < PS; i += 2) *reinterpret_cast<double2 *>(col + i) = make_double2(a[i], a[i + 1]);
}

// Column k of the compacted counts is P contiguous floats (Fb[k * P + i]): three vector loads at p = 10 instead of
// ten scalar ones; a wave reads 64 * 4P contiguous bytes.  Columns are only 4-byte aligned, which global loads allow.
typedef float dn_f4 __attribute__((ext_vector_type(4)));
typedef float dn_f2 __attribute__((ext_vector_type(2)));
typedef dn_f4 dn_f4u __attribute__((aligned(4)));
typedef dn_f2 dn_f2u __attribute__((aligned(4)));
typedef const dn_f4u __attribute__((address_space(1))) *gF4_cptr;
typedef const dn_f2u __attribute__((address_space(1))) *gF2_cptr;

template <int P>
__device__ __forceinline__ void load_x(gF_cptr Fb, int k, float (&x)[P])
{
    gF_cptr col = Fb + (size_t) k * P;
#pragma unroll
    for (int i = 0; i + 4 <= P; i += 4) {
        const dn_f4 v = *(gF4_cptr) (col + i);
        x[i] = v.x; x[i + 1] = v.y; x[i + 2] = v.z; x[i + 3] = v.w;
    }
    if constexpr ((P & 3) >= 2) {
        const dn_f2 v = *(gF2_cptr) (col + (P & ~3));
        x[P & ~3] = v.x; x[(P & ~3) + 1] = v.y;
    }
    if constexpr (P & 1) x[P - 1] = col[P - 1];
}

template <int P>
__device__ __forceinline__ void load_f(gF_cptr Fb, int k, const double (&inv)[P], double (&f)[P])
{
    float x[P];
    load_x<P>(Fb, k, x);
#pragma unroll
    for (int i = 0; i < P; i++) f[i] = (double) x[i] * inv[i];
}

// the counts of a column as the pass wants them: scaled (x / s_i), or left raw when the state is kept in raw units (DN_RAW_UNITS)
template <int P, bool RAW>
__device__ __forceinline__ void pass_counts(const float (&x)[P], const double (&inv)[P], double (&f)[P])
{
#pragma unroll
    for (int i = 0; i < P; i++) f[i] = RAW ? (double) x[i] : (double) x[i] * inv[i];
}

// Spill-tier layout: blocks of 64 columns, inside a block the p rows of 64 doubles back to back.  A wave (64 consecutive
// columns) reads or writes 512 contiguous bytes per row exactly as with whole rows [p][S], but the ten accesses of a
// column differ only by the immediate offset i * 512 from ONE address -- with whole rows every row needs its own 64-bit
// address register (20 VGPRs at p = 10), which pushed the prefetched column into AGPRs (40 extra moves per column).
template <int P>
__device__ __forceinline__ gdouble_ptr spill_ptr(gdouble_ptr Lg, int k)
{
    return Lg + ((size_t) (k >> 6) * (64 * P) + (size_t) (k & 63));
}

// ---------------------------------------------------------------------------------------------------
// Wide cohorts (p >= DN_MG_MIN_P): cold start and the T inner iterations of one nmf() call with the Gram matrix on the
// matrix cores.  A per-lane register Gram needs p (p + 1) / 2 accumulators and, beyond ~100 of them, several sweeps over
// the state; a 16 x 16 fp64 MFMA tile needs 8 registers, so all (p / 16)^2 / 2 tiles of the matrix stay resident and ONE
// pass per inner iteration is enough.  fp64 MFMA runs at the fp64 vector rate on gfx950 (and does not overlap with it),
// so the arithmetic costs the same; what goes away is the re-reading of the state, the AGPR traffic of the accumulators
// and the per-wave reduce-scatter (an MFMA sums over the 64 columns of a wave by itself).
// Every wave walks its 64-column steps in lock-step: each lane updates its column as in the narrow-cohort kernels, the
// updated columns are staged 16 at a time in LDS (zero-padded to 16 x MG_TR rows), re-read in the operand layout
// (lane (i = l & 15, kq = l >> 4) <- row 16 tr + i of staged column 4 g + kq) and multiplied tile row by tile row:
// D[tr1][tr2] += X[tr1] X[tr2]^T over four columns per instruction.
// ---------------------------------------------------------------------------------------------------
template <int P, int NT>
__device__ __forceinline__ int mg_core(gF_cptr Fb, gdouble_ptr Lg, double *lam, int n, int nL, int T,
                                       double (&u)[P], double &theta, int &steps, int maxs, bool &noconv)
{
    constexpr int W = NT / 64;
    constexpr int NG = P * (P + 1) / 2;
    constexpr int PS = P + (P & 1);
    constexpr int TR = Smem<P, NT>::MG_TR, STR = Smem<P, NT>::MG_STR;
    constexpr int NTILE = TR * (TR + 1) / 2;
    Smem<P, NT> &sm = g_sm;
    const int tid = DN_TIDX, lane = lane_id(), w = wave_id();
    const int nLe = (n < nL) ? n : nL;
    double *stage = &sm.stage[w * 16 * STR];
    double *uv = &sm.eigv[w * 128], *vv = uv + 64;                     // this wave's copy of u and a work vector
    const double *invp = g_gs.inv;
    for (int e = lane; e < 16 * STR; e += 64) stage[e] = 0.0;          // rows >= p of a staged column stay zero
    uv[lane] = lane < P ? 1.0 / sqrt((double) P) : 0.0;                // cold start of the solver
    vv[lane] = 0.0;
    wave_fence();
    const int oi = lane & 15, okq = lane >> 4;
    dn_double4 acc[NTILE];

    auto feed = [&](const double (&a)[PS], bool valid) {
#pragma unroll
        for (int cq = 0; cq < 4; cq++) {
            if ((lane >> 4) == cq) {
                double *dst = stage + (lane & 15) * STR;
#pragma unroll
                for (int i = 0; i < PS; i += 2) *reinterpret_cast<double2 *>(dst + i) = make_double2(valid ? a[i] : 0.0, valid ? a[i + 1] : 0.0);
            }
            wave_fence();
#pragma unroll
            for (int g = 0; g < 4; g++) {
                double x[TR];
#pragma unroll
                for (int tr = 0; tr < TR; tr++) x[tr] = stage[(4 * g + okq) * STR + 16 * tr + oi];
                int tix = 0;
#pragma unroll
                for (int t1 = 0; t1 < TR; t1++)
#pragma unroll
                    for (int t2 = 0; t2 <= t1; t2++, tix++)
                        acc[tix] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[t1], x[t2], acc[tix], 0, 0, 0);
            }
            wave_fence();
        }
    };

    const double c = 1.0 / sqrt((double) T);                           // nmf.py:91
    // one pass over the active columns: cold (a = F, Gram of F: nmf.py:88) or inner iteration t (nmf.py:93-99)
    auto pass = [&](bool cold, int t) {
#pragma unroll
        for (int i = 0; i < NTILE; i++) acc[i] = dn_double4{0.0, 0.0, 0.0, 0.0};
        const int wb = w * 64;
        const int steps_l = wb < nLe ? (nLe - wb + NT - 1) / NT : 0;    // wave-uniform trip counts
        const int steps_s = nL + wb < n ? (n - nL - wb + NT - 1) / NT : 0;
#pragma clang loop unroll(disable)
        for (int j = 0; j < steps_l; j++) {
            const int k = tid + j * NT;
            const bool valid = k < nLe;
            double a[PS];
#pragma unroll
            for (int i = 0; i < PS; i++) a[i] = 0.0;
            if (valid) {
                float xr[P];
                load_x<P>(Fb, k, xr);
                if (cold) {
#pragma unroll
                    for (int i = 0; i < P; i++) a[i] = (double) xr[i] * invp[i];
                } else {
                    double al[PS], aa[P];
                    lds_col_read<PS>(lam + (size_t) k * PS, al);
#pragma unroll
                    for (int i = 0; i < P; i++) aa[i] = al[i];
                    col_update_lds<P>(xr, invp, aa, uv, c);
#pragma unroll
                    for (int i = 0; i < P; i++) a[i] = aa[i];
                }
                lds_col_write<PS>(lam + (size_t) k * PS, a);            // lmbda = zeros on the cold pass: state a = x
            }
            feed(a, valid);
        }
#pragma clang loop unroll(disable)
        for (int j = 0; j < steps_s; j++) {
            const int k = nL + tid + j * NT;
            const bool valid = k < n;
            double a[PS];
#pragma unroll
            for (int i = 0; i < PS; i++) a[i] = 0.0;
            if (valid) {
                float xr[P];
                double aa[P];
                load_x<P>(Fb, k, xr);
                if (cold || t == 0) {
#pragma unroll
                    for (int i = 0; i < P; i++) aa[i] = (double) xr[i] * invp[i];
                } else {
#pragma unroll
                    for (int i = 0; i < P; i++) aa[i] = *(spill_ptr<P>(Lg, k) + i * 64);
                }
                if (!cold) {
                    col_update_lds<P>(xr, invp, aa, uv, c);
#pragma unroll
                    for (int i = 0; i < P; i++) *(spill_ptr<P>(Lg, k) + i * 64) = aa[i];
                }
#pragma unroll
                for (int i = 0; i < P; i++) a[i] = aa[i];
            }
            feed(a, valid);
        }
        // tile (t1, t2), register r of lane (c = l & 15, q = l >> 4) holds G[16 t1 + q + 4 r][16 t2 + c]: this wave's
        // totals go to xw, the cross-wave add packs them into tot
        {
            double *dst = (W > 1) ? sm.xw[w] : sm.tot;
            int tix = 0;
#pragma unroll
            for (int t1 = 0; t1 < TR; t1++)
#pragma unroll
                for (int t2 = 0; t2 <= t1; t2++, tix++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = 16 * t1 + okq + 4 * r, col = 16 * t2 + oi;
                        if (row < P && col <= row) dst[row * (row + 1) / 2 + col] = acc[tix][r];
                    }
        }
        if constexpr (W > 1) {
            dn_sync();
            for (int e = tid; e < NG; e += NT) {
                double tsum = sm.xw[0][e];
#pragma unroll
                for (int ww = 1; ww < W; ww++) tsum += sm.xw[ww][e];
                sm.tot[e] = tsum;
            }
        }
        dn_sync();
    };

    pass(true, 0);
    {
        const double tr = wave_allsum(lane < P ? sm.tot[lane * (lane + 1) / 2 + lane] : 0.0);
        if (!(tr > 0.0)) return ST_ARPACK;
    }
    { const int r = top_eig_rows_lds<P>(sm.tot, uv, vv, theta, maxs); steps += r; noconv = noconv || r > maxs; }
    bool cold_every_solve;                                             // see warm_start_unsafe
    {
        double um = lane < P ? uv[lane] : 1.0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) um = fmin(um, __shfl_xor(um, o));
        cold_every_solve = __builtin_amdgcn_readfirstlane((int) (um < WARM_START_MIN_COMPONENT)) != 0;
    }
#pragma clang loop unroll(disable)
    for (int t = 0; t < T; t++) {
        pass(false, t);
        int r;
        if (!cold_every_solve) r = top_eig_rows_lds<P>(sm.tot, uv, vv, theta, maxs);
        else {                                                          // solve_by_blocks with this solver's vectors
            unsigned long long todo = P >= 64 ? ~0ull : ((1ull << P) - 1ull);
            double bul = 0.0, bth = -1.0;
            int sum = 0;
            bool capped = false;
            while (todo) {
                const unsigned long long comp = gram_component(sm.tot, __builtin_ctzll(todo), P);
                todo &= ~comp;
                wave_fence();
                uv[lane] = ((comp >> lane) & 1ull) ? 1.0 / sqrt((double) __builtin_popcountll(comp)) : 0.0;
                wave_fence();
                double th = 0.0;
                const int rr = top_eig_rows_lds<P>(sm.tot, uv, vv, th, maxs);
                sum += rr < maxs ? rr : maxs;
                capped = capped || rr > maxs;
                const double ul = uv[lane];
                if (th > bth) { bth = th; bul = ul; }
            }
            wave_fence();
            uv[lane] = bul;
            wave_fence();
            theta = bth > 0.0 ? bth : 0.0;
            r = capped ? maxs + 1 : (sum < maxs ? sum : maxs);
        }
        steps += r; noconv = noconv || r > maxs;
    }
    bcast_rows<P>(lane < P ? uv[lane] : 0.0, u);                        // the final pass wants u as a uniform array
    return ST_OK;
}

// ---------------------------------------------------------------------------------------------------
// Register tier of the x + lambda state (narrow cohorts, DN_RT_MIN_P <= p <= DN_RT_MAX_P).
// At one wave per SIMD a lane owns 512 registers; the pass needs ~250 of them (Gram accumulators, the column in flight,
// the prefetched counts), the other half sat idle while the state of long genes went out to the fabric (spill tier).
// Here the kernel is compiled for a budget of 256 registers (__launch_bounds__(NT, 2): the allocator then uses
// architectural VGPRs only and never touches an accumulation register -- AGPRs: 0 in the resource report of such a
// build), and ALL 256 AGPRs hold state: column r of lane tid is column tid + r * NT of the gene, its p doubles sit in
// a[2 p r ...).  They are reached only through the v_accvgpr_read / v_accvgpr_write below, with literal register
// numbers (the code is unrolled over r); the "a255" clobber in k_baseline makes the kernel descriptor ask for all 512
// registers, so the hardware still runs one wave per SIMD.  tests/test_host.py checks on the generated ISA that no
// compiler-generated instruction names an AGPR.
// Order of the tiers along a gene: registers [0, RT * NT), then LDS, then the spill array.
// p = 10, 256 threads: 12 x 256 = 3 072 columns in registers + 1 950 in LDS >= the longest config-2 gene.
// ---------------------------------------------------------------------------------------------------
#ifndef DN_RT_MAX_COLS
#define DN_RT_MAX_COLS 12
#endif
#ifndef DN_RT_MAX_P
#define DN_RT_MAX_P 12           // above it the Gram accumulators alone need more than 256 registers
#endif
#ifndef DN_RT_MIN_P
#define DN_RT_MIN_P 2            // round 3: the tier (and with it the pair class) serves every p <= 12.  Round 2 stopped at 8 ("below it a
#endif                           // workgroup needs so few registers that several share a SIMD") -- measured on config-2-shaped genes: p = 3
                                 // +31 %, 4 +7 %, 5 +31 %, 6 +26 %, 7 +47 % genes/s (tools/p_sweep.py, profiles/round3/p_sweep.txt)
// DN_RAW_UNITS: inside the T loop of the register-tier cohorts the state is kept in RAW count units (a~ = x + lambda / s_i instead
// of x / s_i + lambda): the counts need no scaling multiply per element and pass (-10 of the 171 vector instructions of a
// register-tier column at p = 10); u is handed to the pass as u_i / s_i (dot product) and u_i s_i (residual), the Gram matrix is
// accumulated in raw units and scaled entry by entry as the block total is written (block_sum_lds), so the solver sees what it saw
#ifndef DN_RAW_UNITS
#define DN_RAW_UNITS 1
#endif
#if defined(DN_P) && DN_P >= DN_RT_MIN_P && DN_P <= DN_RT_MAX_P && !defined(DN_NO_REG_TIER)
#define DN_REG_TIER 1
#define DN_KERNEL_WAVES 2        // the register ALLOCATOR's budget: 512 / 2 registers (the kernel really runs one wave per SIMD)
// The caller's view: nmf_call() visibly uses no AGPR (it must not name one in a constraint or clobber: a function that
// visibly uses AGPRs gets half of its budget as AGPRs, 128 + 128), so inter-procedural register allocation lets
// k_baseline park values in accumulation registers across the call.  nmf_call() therefore SAVES the registers of the
// tier to the workgroup's scratch slot on entry and restores them on every exit (rt_save / rt_restore: 2 x 250 moves
// and as many 4-byte accesses per lane and call, against >= 100 passes over the gene in between).
// k_baseline names a255 once so that the kernel descriptor asks for all 512 registers of a lane.
#define DN_RT_CLAIM() asm volatile("" ::: "a255")
#else
#define DN_REG_TIER 0
#define DN_KERNEL_WAVES 1
#define DN_RT_CLAIM()
#endif
#ifndef DN_RAW_MAX_P
#if DN_SOLVER_DPP
#define DN_RAW_MAX_P 12          // every register-tier cohort (round 4)
#else
#define DN_RAW_MAX_P 11          // with the squaring solver of rounds 1-3 (DN_SOLVER_DPP=0) p = 12 stays in scaled units: its raw-unit pair build
                                 // failed the edge-shape parity test on genes that fill the register tier -- the two coefficient sets of the pass
                                 // came out wrong when they were scaled in that solver's iterate layout (lane group q holds v[q + 4 kb], 3 registers
                                 // at p = 12 with NO padding lane) and broadcast from there; with top_eig_dpp the iterate is one register, component i
                                 // in lane i, the same two multiplies and readlanes pass (tests: pair class and edge shapes, p = 2 .. 12, tier-filling genes)
#endif
#endif
template <int P> constexpr bool raw_units() { return DN_RAW_UNITS != 0 && DN_REG_TIER != 0 && P <= DN_RAW_MAX_P && P < DN_MG_MIN_P; }

// agpr_get1<N>() / agpr_put1<N>(v): the 32-bit value held in aN.  Register names must be literal text, hence the list.
template <int IDX> __device__ __forceinline__ int agpr_get1();
template <int IDX> __device__ __forceinline__ void agpr_put1(int v);
#define DN_AGPR_ACCESSORS(N)                                                                                              \
    template <> __device__ __forceinline__ int agpr_get1<N>() { int v; asm volatile("v_accvgpr_read_b32 %0, a" #N : "=v"(v)); return v; } \
    template <> __device__ __forceinline__ void agpr_put1<N>(int v) { asm volatile("v_accvgpr_write_b32 a" #N ", %0" : : "v"(v)); }
#define DN_AGPR_REGS(X) \
    X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) \
    X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) \
    X(32) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40) X(41) X(42) X(43) X(44) X(45) X(46) X(47) \
    X(48) X(49) X(50) X(51) X(52) X(53) X(54) X(55) X(56) X(57) X(58) X(59) X(60) X(61) X(62) X(63) \
    X(64) X(65) X(66) X(67) X(68) X(69) X(70) X(71) X(72) X(73) X(74) X(75) X(76) X(77) X(78) X(79) \
    X(80) X(81) X(82) X(83) X(84) X(85) X(86) X(87) X(88) X(89) X(90) X(91) X(92) X(93) X(94) X(95) \
    X(96) X(97) X(98) X(99) X(100) X(101) X(102) X(103) X(104) X(105) X(106) X(107) X(108) X(109) X(110) X(111) \
    X(112) X(113) X(114) X(115) X(116) X(117) X(118) X(119) X(120) X(121) X(122) X(123) X(124) X(125) X(126) X(127) \
    X(128) X(129) X(130) X(131) X(132) X(133) X(134) X(135) X(136) X(137) X(138) X(139) X(140) X(141) X(142) X(143) \
    X(144) X(145) X(146) X(147) X(148) X(149) X(150) X(151) X(152) X(153) X(154) X(155) X(156) X(157) X(158) X(159) \
    X(160) X(161) X(162) X(163) X(164) X(165) X(166) X(167) X(168) X(169) X(170) X(171) X(172) X(173) X(174) X(175) \
    X(176) X(177) X(178) X(179) X(180) X(181) X(182) X(183) X(184) X(185) X(186) X(187) X(188) X(189) X(190) X(191) \
    X(192) X(193) X(194) X(195) X(196) X(197) X(198) X(199) X(200) X(201) X(202) X(203) X(204) X(205) X(206) X(207) \
    X(208) X(209) X(210) X(211) X(212) X(213) X(214) X(215) X(216) X(217) X(218) X(219) X(220) X(221) X(222) X(223) \
    X(224) X(225) X(226) X(227) X(228) X(229) X(230) X(231) X(232) X(233) X(234) X(235) X(236) X(237) X(238) X(239) \
    X(240) X(241) X(242) X(243) X(244) X(245) X(246) X(247) X(248) X(249) X(250) X(251) X(252) X(253) X(254) X(255)
DN_AGPR_REGS(DN_AGPR_ACCESSORS)
#undef DN_AGPR_ACCESSORS
#undef DN_AGPR_REGS
template <int IDX> __device__ __forceinline__ double agpr_get()          // the double in a[IDX : IDX + 1] (any IDX)
{
    const int lo = agpr_get1<IDX>(), hi = agpr_get1<IDX + 1>();
    return __hiloint2double(hi, lo);
}
template <int IDX> __device__ __forceinline__ void agpr_put(double v)
{
    agpr_put1<IDX>(__double2loint(v));
    agpr_put1<IDX + 1>(__double2hiint(v));
}

// Column r of a lane: registers [BASE, BASE + 2 P) hold the p doubles of x + lambda; with X16 the next ceil(p / 2)
// registers hold the column's raw counts packed as 16-bit integers (exact up to 65 535; a gene with a larger count takes
// the variant without them), so that a register-tier column needs no memory access at all in a pass.
template <int P, int BASE> __device__ __forceinline__ void rt_read(double (&a)[P])
{
    static_for<0, P>([&](auto ic) { constexpr int I = decltype(ic)::value; a[I] = agpr_get<BASE + 2 * I>(); });
}
template <int P, int BASE> __device__ __forceinline__ void rt_write(const double (&a)[P])
{
    static_for<0, P>([&](auto ic) { constexpr int I = decltype(ic)::value; agpr_put<BASE + 2 * I>(a[I]); });
}
template <int P, int BASE> __device__ __forceinline__ void rt_read_counts(double (&xd)[P])
{
    static_for<0, (P + 1) / 2>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        const unsigned w = (unsigned) agpr_get1<BASE + J>();
        xd[2 * J] = (double) (w & 0xffffu);
        if constexpr (2 * J + 1 < P) xd[2 * J + 1] = (double) (w >> 16);
    });
}
template <int P, int BASE> __device__ __forceinline__ void rt_write_counts(const float (&x)[P])
{
    static_for<0, (P + 1) / 2>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        unsigned w = (unsigned) x[2 * J];
        if constexpr (2 * J + 1 < P) w |= ((unsigned) x[2 * J + 1]) << 16;
        agpr_put1<BASE + J>((int) w);
    });
}
template <int P, bool X16> constexpr int rt_col_regs() { return 2 * P + (X16 ? (P + 1) / 2 : 0); }
template <int P, bool X16> constexpr int rt_cols()
{
    return (P >= DN_RT_MIN_P && P <= DN_RT_MAX_P) ? (256 / rt_col_regs<P, X16>() < DN_RT_MAX_COLS ? 256 / rt_col_regs<P, X16>() : DN_RT_MAX_COLS) : 0;
}
template <int P> constexpr int rt_regs_used()          // registers a0 .. a(N - 1) that either variant touches
{
    return rt_cols<P, true>() * rt_col_regs<P, true>() > rt_cols<P, false>() * rt_col_regs<P, false>()
               ? rt_cols<P, true>() * rt_col_regs<P, true>() : rt_cols<P, false>() * rt_col_regs<P, false>();
}
// The caller's contents of the tier's registers, parked in the scratch slot for the duration of one nmf() call:
// register i of lane tid at save[i * NT + tid] (register-major: the 64 lanes of a wave write 256 contiguous bytes; with
// the save area's base in scalar registers one address register and immediate offsets still serve every access).  The
// earlier lane-major layout (save[tid * N + i], 16-byte pieces 1 000 bytes apart) measured 0.5 % slower on config 2.
// Batches of RT_BATCH registers: the loads of a batch are all in flight together (one memory latency per batch, not per
// register), and the compiler barrier between batches keeps it from gathering every register first.
constexpr int RT_BATCH = 48;
#define DN_RT_SAVE_IDX(r) ((size_t) (r) * NT + DN_TIDX)      // register-major: a wave's 64 lanes write 256 contiguous bytes
// `need`: registers a0 .. a(need - 1) are the only ones this call can write (columns beyond the gene's width are never
// touched), so only whole batches below it are parked.
template <int N, int NT> __device__ __forceinline__ void rt_save(int *save, int need)
{
    static_for<0, (N + RT_BATCH - 1) / RT_BATCH>([&](auto bc) {
        constexpr int B = decltype(bc)::value * RT_BATCH;
        constexpr int CNT = (N - B) < RT_BATCH ? (N - B) : RT_BATCH;
        if (B < need) {
            int v[CNT];
            static_for<0, CNT>([&](auto ic) { constexpr int I = decltype(ic)::value; v[I] = agpr_get1<B + I>(); });
#pragma unroll
            for (int i = 0; i < CNT; i++) save[DN_RT_SAVE_IDX(B + i)] = v[i];
            asm volatile("" ::: "memory");
        }
    });
}
template <int N, int NT> __device__ __forceinline__ void rt_restore(const int *save, int need)
{
    static_for<0, (N + RT_BATCH - 1) / RT_BATCH>([&](auto bc) {
        constexpr int B = decltype(bc)::value * RT_BATCH;
        constexpr int CNT = (N - B) < RT_BATCH ? (N - B) : RT_BATCH;
        if (B < need) {
            int v[CNT];
#pragma unroll
            for (int i = 0; i < CNT; i++) v[i] = save[DN_RT_SAVE_IDX(B + i)];
            static_for<0, CNT>([&](auto ic) { constexpr int I = decltype(ic)::value; agpr_put1<B + I>(v[I]); });
            asm volatile("" ::: "memory");
        }
    });
}
template <int P, int NT> constexpr size_t rt_save_bytes() { return (size_t) rt_regs_used<P>() * 4 * NT; }

// FULL: the gene fills the register tier (n >= RT * NT): every lane owns all RT columns and the tier is walked as straight-line code
// ONCHIP (with FULL): the gene also fits the register + LDS tiers (n <= RT * NT + lds_cols): the body carries no spill tier,
// which is where the register pressure of the pass peaks -- the registers that frees let the first tier column start the Gram
// accumulators (no zeroing, two-source multiplies) without a spill
template <int P, int NT, bool X16, bool FULL = false, bool ONCHIP = false, bool SAFE = false>
__device__ __forceinline__ void nmf_body(const float *Fb_, double *Lg_, double *rs_, double *sv_,
                                         int n, int S, int nL, int T, int first_i)
{
    Smem<P, NT> &sm = g_sm;
    double *lam = g_lam + (DN_UNITS > 1 ? (size_t) DN_UNIT * (size_t) nL * (P + (P & 1)) : 0);    // this unit's share of the tile
    // every argument is wave-uniform but arrives in vector registers (calling convention): move them to
    // scalar registers so that all row-base arithmetic below is SALU and costs no VGPRs
    gF_cptr Fb = (gF_cptr) uniform_ptr(Fb_);
    gdouble_ptr Lg = (gdouble_ptr) uniform_ptr(Lg_);
    gdouble_ptr rs = (gdouble_ptr) uniform_ptr(rs_);
    gdouble_ptr sv = (gdouble_ptr) uniform_ptr(sv_);
    n = __builtin_amdgcn_readfirstlane(n);
    S = __builtin_amdgcn_readfirstlane(S);
    nL = __builtin_amdgcn_readfirstlane(nL);
    T = __builtin_amdgcn_readfirstlane(T);
    const bool first = __builtin_amdgcn_readfirstlane(first_i) != 0;
    double inv[P], u[P], theta = 0.0;
    int steps = 0;
    const int maxs = __builtin_amdgcn_readfirstlane(g_gs.max_steps);
    bool noconv = false;
#pragma unroll
    for (int i = 0; i < P; i++) inv[i] = uniform(g_gs.inv[i]);
    long long stamp[4] = {0, 0, 0, 0};
#ifdef DN_STAMP
#define DN_T0() const long long t0_ = __builtin_amdgcn_s_memtime()
#define DN_T1(slot) stamp[slot] += __builtin_amdgcn_s_memtime() - t0_
#else
#define DN_T0()
#define DN_T1(slot) (void) stamp
#endif
    constexpr int NG = P * (P + 1) / 2;
    // The Gram matrix is accumulated in SW sweeps of CH packed entries each: one register set holds ~105-120 fp64
    // accumulators next to the column in flight (p <= 15, the last of them already through AGPR copies).  From p = 16 on
    // the first sweep updates the state and later sweeps re-read it (read-only) for the remaining entries.  ~90 entries
    // per sweep measured best at p = 20 and p = 32 alike (36 / 53 / 70 / 90 / 105 / 132 / 176 tried: fewer, fatter
    // sweeps trade state re-reads for AGPR copies; beyond ~105 the accumulators reach scratch memory).
    constexpr int CH_MAX = P <= 15 ? 120 : 90;
    constexpr int SW = (NG + CH_MAX - 1) / CH_MAX;
    constexpr int CH = (NG + SW - 1) / SW;
    constexpr int PS = P + (P & 1);                        // LDS column stride in doubles
    const int tid = DN_TIDX;
    constexpr int RT = DN_REG_TIER ? rt_cols<P, X16>() : 0;   // columns per lane held in AGPRs (register tier)
    constexpr int CS = rt_col_regs<P, X16>();              // registers per column: the state, and with X16 the packed raw counts
    constexpr int NR = RT * NT;                            // the gene's first NR columns
    constexpr bool RAW = raw_units<P>();                   // the state of the T loop in raw count units (DN_RAW_UNITS)
    double ud[P], ur[P];                                   // u as the pass takes it: dot-product and residual coefficients (col_update)
    const int nLe = (n < NR + nL) ? n : NR + nL;           // end of the LDS tier (absolute column); LDS slot of column k: k - NR
    const int kS0 = NR + nL;                               // first column of the spill tier
    if constexpr (P >= DN_MG_MIN_P) {
        const int st = mg_core<P, NT>(Fb, Lg, lam, n, nL, T, u, theta, steps, maxs, noconv);
        if (st != ST_OK) { if (tid == 0) g_gs.status = st; dn_sync(); return; }
    } else {
    gram_t G[CH];

    // cold start: SVD of x itself (nmf.py:88)
#ifdef DN_STAMP
    const long long t_cold0 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int i = 0; i < CH; i++) G[i] = 0.0;
    {
        float xq[P];                                                  // the next column's counts fly during this column's products
        if (tid < n) load_x<P>(Fb, tid, xq);
#pragma clang loop unroll(disable)
        for (int k = tid; k < n; k += NT) {
            double f[P];
#pragma unroll
            for (int i = 0; i < P; i++) f[i] = (double) xq[i] * inv[i];
            load_x<P>(Fb, k + NT < n ? k + NT : k, xq);
            gram_add_range<P, 0, CH>(G, f);
        }
    }
    block_sum_lds<CH, P, NT, gram_t>(G, sm);
    static_for<1, SW>([&](auto qc) {
        constexpr int Q = decltype(qc)::value;
        constexpr int NQ = (NG - Q * CH) < CH ? (NG - Q * CH) : CH;
        gram_t Gq[NQ];
#pragma unroll
        for (int i = 0; i < NQ; i++) Gq[i] = 0.0;
#pragma clang loop unroll(disable)
        for (int k = tid; k < n; k += NT) {
            double f[P];
            load_f<P>(Fb, k, inv, f);
            gram_add_range<P, Q * CH, NQ>(Gq, f);
        }
        block_sum_lds<NQ, P, NT, gram_t, false, Q * CH>(Gq, g_sm);
    });
    {
        double tr = 0.0;
#pragma unroll
        for (int i = 0; i < P; i++) tr += sm.tot[i * (i + 1) / 2 + i];
        if (!(tr > 0.0)) { if (tid == 0) g_gs.status = ST_ARPACK; dn_sync(); return; }
    }
    Solver<P> solver;
    {
        double tr = 0.0;
#pragma unroll
        for (int i = 0; i < P; i++) tr += sm.tot[i * (i + 1) / 2 + i];
        solver.cold(tr, u);
    }
    {
        const int r = SAFE ? solve_by_blocks<P>(solver, sm.tot, Smem<P, NT>::ZSLOT, u, theta, maxs)
                           : solver.run(sm.tot, Smem<P, NT>::ZSLOT, u, theta, T == 0, maxs, true);
        steps += r; noconv = noconv || r > maxs;
    }
    const double c = 1.0 / sqrt((double) T);                         // nmf.py:91
    // u for the pass, in scalar registers (two-VGPR-source FMAs).  Raw units: u_i / s_i and u_i s_i, scaled per lane in the solver's
    // iterate layout (lane group q holds v[q + 4 kb]) and broadcast from there
    auto pass_u = [&]() {
        if constexpr (RAW) {
#if DN_SOLVER_DPP
            const int k = Solver<P>::row_of_lane() < P ? Solver<P>::row_of_lane() : P - 1;
            const double vt = solver.iterate_of_row() * g_gs.inv[k], vh = solver.iterate_of_row() * g_gs.scl[k];
#pragma unroll
            for (int i = 0; i < P; i++) { ud[i] = Solver<P>::bcast_component(vt, i); ur[i] = Solver<P>::bcast_component(vh, i); }
#else
            constexpr int KB = EigState<P>::KB;
            const int q = lane_id() >> 4;
            double vt[KB], vh[KB];
#pragma unroll
            for (int kb = 0; kb < KB; kb++) {
                const int k = q + 4 * kb < P ? q + 4 * kb : P - 1;
                vt[kb] = solver.st.v[kb] * g_gs.inv[k];
                vh[kb] = solver.st.v[kb] * g_gs.scl[k];
            }
#pragma unroll
            for (int i = 0; i < P; i++) {
                const double a_ = vt[i >> 2], b_ = vh[i >> 2];
                ud[i] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a_), 16 * (i & 3)),
                                         __builtin_amdgcn_readlane(__double2loint(a_), 16 * (i & 3)));
                ur[i] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(b_), 16 * (i & 3)),
                                         __builtin_amdgcn_readlane(__double2loint(b_), 16 * (i & 3)));
            }
#endif
        } else {
#pragma unroll
            for (int i = 0; i < P; i++) { u[i] = uniform(u[i]); ud[i] = u[i]; ur[i] = u[i]; }
        }
    };
    pass_u();

    static_for<0, RT>([&](auto rc) {                             // lmbda = zeros (nmf.py:90): state a = x
        constexpr int R = decltype(rc)::value;
        const int k = tid + R * NT;
        if (k < n) {
            float xf[P];
            double f[P];
            load_x<P>(Fb, k, xf);
            pass_counts<P, RAW>(xf, inv, f);
            rt_write<P, CS * R>(f);
            if constexpr (X16) rt_write_counts<P, CS * R + 2 * P>(xf);
        }
    });
    for (int k = NR + tid; k < nLe; k += NT) {
        double f[P], a[PS];
        { float xf[P]; load_x<P>(Fb, k, xf); pass_counts<P, RAW>(xf, inv, f); }
#pragma unroll
        for (int i = 0; i < PS; i++) a[i] = i < P ? f[i] : 0.0;
        lds_col_write<PS>(lam + (size_t) (k - NR) * PS, a);
    }
#ifdef DN_STAMP
    stamp[3] += __builtin_amdgcn_s_memtime() - t_cold0;
#endif
    constexpr bool G0 = FULL && ONCHIP && SW == 1;          // the first tier column STARTS the Gram accumulators (no zeroing)
    constexpr bool SQ = Solver<P>::SQUARE && SW == 1 && !SAFE;   // the solver reads the scaled square copy of the Gram matrix (Smem::sq)
#pragma clang loop unroll(disable)
    for (int t = 0; t < T; t++) {
        DN_MARK("iter_begin");
        if constexpr (!G0) {
#pragma unroll
            for (int i = 0; i < CH; i++) G[i] = 0.0;
        }
        DN_MARK("zeroed");
        { DN_T0();
        // The columns are walked forwards on even passes and backwards on odd ones (spill tier first, then the LDS tier):
        // what the previous pass touched last -- the end of the spill state and of the counts -- is still in the XCD's
        // L2 when the next pass starts there, where a cyclic walk over more than the L2 share never hits (+2 % on
        // config 2, -7 % pass time for the longest genes under full load).  Keeping a separate forward-only loop for
        // genes that fit in LDS (with the first column starting the accumulators instead of zeroing them) made every
        // loop slower: three loop variants in one function cost more in register allocation than the zeroing does.
        const int dir = (t & 1) ? -1 : 1;

        // LDS tier.  Column o keeps its p doubles contiguously (stride PS = p rounded up to even, 16-B aligned):
        // 128-bit LDS accesses, conflict-free because the lane stride (20 dwords at p = 10) is 4 x odd.
        // The next column's raw counts (10 VGPRs) are requested before this column's arithmetic starts, so the L2
        // latency of the only global read of the tier hides behind ~120 fp64 operations.  Prefetching the fp64 state
        // as well measured 1.7x SLOWER per column -- the extra live registers end up in AGPRs and every use pays a copy
        // -- and so did a register-resident tier in front of it (tools/trace_stats.py, profiles/round1).
        // register tier: unrolled over the lane's RT columns, the next column's counts requested before this one's arithmetic
        auto reg_tier = [&]() {
            if constexpr (RT > 0 && X16) {
                // counts AND state in registers: a column of this tier touches no memory in a pass
                auto column = [&](auto rc) {
                    constexpr int R = decltype(rc)::value;
                    double f[P], a[P];
                    rt_read_counts<P, CS * R + 2 * P>(f);
                    if constexpr (!RAW) {
#pragma unroll
                        for (int i = 0; i < P; i++) f[i] *= inv[i];
                    }
                    rt_read<P, CS * R>(a);
                    col_update<P>(f, a, ud, ur, c);
                    if constexpr (G0 && R == 0) gram_set_range<P, 0, CH>(G, a); else gram_add_range<P, 0, CH>(G, a);
                    rt_write<P, CS * R>(a);
                };
                if constexpr (FULL) {
                    // every lane owns all RT columns: one straight line of code, no exec mask and no branch per column (the
                    // per-column mask costs ~90 cycles per column at one wave per SIMD: tools/ubench/clock_issue.hip); the
                    // scheduling barrier keeps the columns apart so that the register pressure stays that of one column
                    static_for<0, RT>([&](auto rc) { column(rc); __builtin_amdgcn_sched_barrier(0); });
                } else {
                    static_for<0, RT>([&](auto rc) {
                        constexpr int R = decltype(rc)::value;
                        if (tid + R * NT < n) column(rc);
                    });
                }
            } else if constexpr (RT > 0) {
                // the next column's counts are requested before this column's arithmetic starts (two columns ahead with two
                // buffers measured no faster: 341.9 vs 339.0 ms per sweep)
                float xq[P];
                if (tid < n) load_x<P>(Fb, tid, xq);
                static_for<0, RT>([&](auto rc) {
                    constexpr int R = decltype(rc)::value;
                    const int k = tid + R * NT;
                    if (k < n) {
                        double f[P], a[P];
                        pass_counts<P, RAW>(xq, inv, f);
                        if constexpr (R + 1 < RT) load_x<P>(Fb, k + NT < n ? k + NT : k, xq);
                        rt_read<P, CS * R>(a);
                        col_update<P>(f, a, ud, ur, c);
                        gram_add_range<P, 0, CH>(G, a);
                        rt_write<P, CS * R>(a);
                    }
                });
            }
        };
        auto lds_tier = [&]() {
            const int first = NR + tid;
            const int cnt = first < nLe ? (nLe - first + NT - 1) / NT : 0;
            const int step = dir * NT;
            int k = dir > 0 ? first : first + (cnt - 1) * NT;
            float xq[P];
            if (cnt > 0) load_x<P>(Fb, k, xq);
#pragma clang loop unroll(disable)
            for (int j = 0; j < cnt; j++, k += step) {
                double f[P], a[PS];
                lds_col_read<PS>(lam + (size_t) (k - NR) * PS, a);
                pass_counts<P, RAW>(xq, inv, f);
                load_x<P>(Fb, j + 1 < cnt ? k + step : k, xq);      // unconditional (clamped): no branch around the loads
                double aa[P];
#pragma unroll
                for (int i = 0; i < P; i++) aa[i] = a[i];
                col_update<P>(f, aa, ud, ur, c);
                gram_add_range<P, 0, CH>(G, aa);
#pragma unroll
                for (int i = 0; i < P; i++) a[i] = aa[i];
                lds_col_write<PS>(lam + (size_t) (k - NR) * PS, a);
            }
        };
        // spill tier: x + lambda of the columns that do not fit in LDS lives in the slot (L2 / Infinity Cache).
        // Here the loads are far away, and prefetching the next column's counts and state does pay (1.15x).
        auto spill_tier = [&]() {
            if constexpr (ONCHIP) return;
            const int first = kS0 + tid;
            const int cnt = first < n ? (n - first + NT - 1) / NT : 0;
            const int step = dir * NT;
            int k = dir > 0 ? first : first + (cnt - 1) * NT;
            float xn[P];
            double an[P];
            if (cnt > 0) {
                // (the first column's index is opaque here: as a value fixed for the whole call the compiler keeps the addresses of its rows
                // beyond the 4 095-byte reach of an immediate offset -- rows 8 and 9 at p = 10 -- in registers of their own, has no room
                // for them next to the Gram accumulators and fetches each back from scratch in front of its load: reload, s_waitcnt
                // vmcnt(0), load -- two memory round trips in a row at the head of every pass over a spill tier)
                int kq = k;
                asm volatile("" : "+v"(kq));
                load_x<P>(Fb, kq, xn);
                if (t > 0) {
#pragma unroll
                    for (int i = 0; i < P; i++) an[i] = *(spill_ptr<P>(Lg, kq) + i * 64);
                }
            }
#pragma clang loop unroll(disable)
            for (int j = 0; j < cnt; j++, k += step) {
                double f[P], a[P];
                pass_counts<P, RAW>(xn, inv, f);
#pragma unroll
                for (int i = 0; i < P; i++) a[i] = t > 0 ? an[i] : f[i];
                if (j + 1 < cnt) {
                    load_x<P>(Fb, k + step, xn);
                    if (t > 0) {
#pragma unroll
                        for (int i = 0; i < P; i++) an[i] = *(spill_ptr<P>(Lg, k + step) + i * 64);
                    }
                }
                col_update<P>(f, a, ud, ur, c);
                gram_add_range<P, 0, CH>(G, a);
#pragma unroll
                for (int i = 0; i < P; i++) *(spill_ptr<P>(Lg, k) + i * 64) = a[i];
            }
        };
#if defined(DN_STAMP) && defined(DN_EXP_TIER)           // diagnostic: only tier DN_EXP_TIER (1 register, 2 LDS, 3 spill) is timed into the pass slot
#define DN_TIER(id, call) do { if (DN_EXP_TIER == id) { const long long tt_ = __builtin_amdgcn_s_memtime(); call; stamp[0] += __builtin_amdgcn_s_memtime() - tt_; } else { call; } } while (0)
#else
#define DN_TIER(id, call) call
#endif
        if constexpr (G0) { DN_MARK("reg_tier"); DN_TIER(1, reg_tier()); DN_MARK("lds_tier"); DN_TIER(2, lds_tier()); }       // the register tier touches no memory: its place is free
        else if (dir > 0) { DN_MARK("reg_tier"); DN_TIER(1, reg_tier()); DN_MARK("lds_tier"); DN_TIER(2, lds_tier()); DN_MARK("spill_tier"); DN_TIER(3, spill_tier()); }
        else { DN_MARK("spill_tier_b"); DN_TIER(3, spill_tier()); DN_MARK("lds_tier_b"); DN_TIER(2, lds_tier()); DN_MARK("reg_tier_b"); DN_TIER(1, reg_tier()); }
        DN_MARK("pass_end");
#if !(defined(DN_STAMP) && defined(DN_EXP_TIER))
        DN_T1(0);
#endif
        }
        { DN_T0(); block_sum_lds<CH, P, NT, gram_t, Solver<P>::SHIFTED, 0, RAW, SQ>(G, sm, solver.shift(), solver.scale(), t & 1); DN_T1(1); }   // tot = G - mu I
        DN_MARK("reduced");
        // later sweeps (p >= 16): the remaining Gram entries from the updated state, read-only
        static_for<1, SW>([&](auto qc) {
            constexpr int Q = decltype(qc)::value;
            constexpr int NQ = (NG - Q * CH) < CH ? (NG - Q * CH) : CH;
            gram_t Gq[NQ];
#pragma unroll
            for (int i = 0; i < NQ; i++) Gq[i] = 0.0;
            static_assert(SW == 1 || RT == 0, "the later sweeps do not read the register tier");
#pragma clang loop unroll(disable)
            for (int k = NR + tid; k < nLe; k += NT) {
                double a[PS], aa[P];
                lds_col_read<PS>(lam + (size_t) (k - NR) * PS, a);
#pragma unroll
                for (int i = 0; i < P; i++) aa[i] = a[i];
                gram_add_range<P, Q * CH, NQ>(Gq, aa);
            }
#pragma clang loop unroll(disable)
            for (int k = kS0 + tid; k < n; k += NT) {
                double aa[P];
#pragma unroll
                for (int i = 0; i < P; i++) aa[i] = *(spill_ptr<P>(Lg, k) + i * 64);
                gram_add_range<P, Q * CH, NQ>(Gq, aa);
            }
            block_sum_lds<NQ, P, NT, gram_t, Solver<P>::SHIFTED, Q * CH>(Gq, g_sm, solver.shift());
        });
        { DN_T0();
        // raw units: the pass takes u from the solver's iterate (pass_u); the broadcast copy is rebuilt once, after the loop
        double u_unused[P];
        double (&us_)[P] = RAW ? u_unused : u;
        const int r = SAFE ? solve_by_blocks<P>(solver, sm.tot, Smem<P, NT>::ZSLOT, us_, theta, maxs)
                           : solver.template run<SQ>(sm.tot, Smem<P, NT>::ZSLOT, us_, theta, t == T - 1, maxs, false,
                                                     sm.sq + wave_id() * Smem<P, NT>::SQ_LEN);   // sigma^2 is only read after the last solve
        steps += r; noconv = noconv || r > maxs;
        pass_u();
        DN_MARK("solved");
        DN_T1(2);
        }
    }
    if constexpr (SQ && NT > 64) dn_sync();                 // the loop's last reduction had one barrier: no wave is still reading xw / xw2
    if constexpr (RAW) {
#pragma unroll
        for (int i = 0; i < P; i++) {
#if DN_SOLVER_DPP
            u[i] = Solver<P>::bcast_component(solver.iterate_of_row(), i);
#else
            const double t_ = solver.st.v[i >> 2];
            u[i] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t_), 16 * (i & 3)),
                                    __builtin_amdgcn_readlane(__double2loint(t_), 16 * (i & 3)));
#endif
        }
    }
    }   // narrow cohorts

    // final pass: K E of the last SVD, its row sums, the clamped row sums and the residual profile.
#ifdef DN_STAMP
    const long long t_fin0 = __builtin_amdgcn_s_memtime();
#endif
    // raw units: the T loop did not need the scale factors; they come back from LDS here instead of sitting in 20 scalar registers
    // through the loop (next to the two coefficient sets of the pass that made the allocator spill scalars every iteration)
    double invf[P];
#pragma unroll
    for (int i = 0; i < P; i++) invf[i] = RAW ? uniform(g_gs.inv[i]) : inv[i];
    double acc[2 * P + 1];
#pragma unroll
    for (int i = 0; i < 2 * P + 1; i++) acc[i] = 0.0;
    auto fin = [&](int k, const double (&f)[P], const double (&l_)[P]) {
        double s, r, l0[P];
#pragma unroll
        for (int i = 0; i < P; i++) l0[i] = RAW ? l_[i] * invf[i] : l_[i];        // raw-unit state -> x / s + lambda
        col_final<P>(f, l0, u, first, acc, s, r);
        rs[k] = r;
        if (first) sv[k] = s;
    };
    static_for<0, RT>([&](auto rc) {
        constexpr int R = decltype(rc)::value;
        const int k = tid + R * NT;
        if (k < n) {
            double f[P], l[P];
            if constexpr (X16) {
                rt_read_counts<P, CS * R + 2 * P>(f);
#pragma unroll
                for (int i = 0; i < P; i++) f[i] *= invf[i];
            } else load_f<P>(Fb, k, invf, f);
            rt_read<P, CS * R>(l);
            fin(k, f, l);
        }
    });
    float xqf[P];
    if (NR + tid < n) load_x<P>(Fb, NR + tid, xqf);
#pragma clang loop unroll(disable)
    for (int k = NR + tid; k < n; k += NT) {
        double f[P], l[P];
#pragma unroll
        for (int i = 0; i < P; i++) f[i] = (double) xqf[i] * invf[i];
        load_x<P>(Fb, k + NT < n ? k + NT : k, xqf);
        if (k < nLe) {
            double al[PS];
            lds_col_read<PS>(lam + (size_t) (k - NR) * PS, al);
#pragma unroll
            for (int i = 0; i < P; i++) l[i] = al[i];
        } else {
#pragma unroll
            for (int i = 0; i < P; i++) l[i] = *(spill_ptr<P>(Lg, k) + i * 64);
        }
        fin(k, f, l);
    }
    block_sum_lds<2 * P + 1, P, NT, double>(acc, sm);
    for (int e = tid; e < 2 * P + 1; e += NT) g_gs.sums[e] = sm.tot[e];      // 2 p + 1 can exceed the workgroup (p = 64, 128 threads)
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < P; i++) g_gs.u[i] = u[i];
        g_gs.theta = theta;
        g_gs.steps += steps;
        g_gs.status = noconv ? ST_NO_CONVERGENCE : ST_OK;
#ifdef DN_STAMP
        g_gs.stamp[0] += stamp[0]; g_gs.stamp[1] += stamp[1]; g_gs.stamp[2] += stamp[2];
        g_gs.stamp[4] += __builtin_amdgcn_s_memtime() - t_fin0; g_gs.stamp[5] += stamp[3];
#endif
    }
    dn_sync();
}

// Out of line on purpose: the call has its own register allocation (Gram accumulators + one column in flight),
// independent of what the state machine keeps live.
template <int P, int NT, bool SAFE = false>
__device__ __attribute__((noinline)) void nmf_call(const float *Fb_, double *Lg_, double *rs_, double *sv_, double *rtsave_,
                                                   int n, int S, int nL, int T, int first_i, int x16_i)
{
    constexpr int NSAVE = DN_REG_TIER ? rt_regs_used<P>() : 0;
    int *rtsave = reinterpret_cast<int *>(uniform_ptr(rtsave_));
#ifdef DN_STAMP
    const long long t_call0 = __builtin_amdgcn_s_memtime();
#endif
    // columns of the tier this gene can reach: ceil(n / NT); their registers in the wider of the two layouts
    const int n_u = __builtin_amdgcn_readfirstlane(n);
    const int need = DN_REG_TIER ? ((n_u + NT - 1) / NT) * rt_col_regs<P, true>() : 0;
    rt_save<NSAVE, NT>(rtsave, need);
    // counts up to 65 535 are carried in the register tier next to the state (X16); a gene with a larger count runs the
    // variant that reads them from the scratch slot (two more state columns per lane instead)
    if constexpr (SAFE) {
        // the safe repeat of a call (k_baseline decides, right after the call): the general body, block-by-block solves.  Its own out-of-line
        // function, so that the hot bodies of the function above keep the registers and the schedule they have without it
        // (as a fifth body of ONE function it cost each T loop a scratch reload: 294.3 against 291.5 ms per sweep on config 2)
        nmf_body<P, NT, false, false, false, true>(Fb_, Lg_, rs_, sv_, n, S, nL, T, first_i);
    } else if (DN_REG_TIER && __builtin_amdgcn_readfirstlane(x16_i) != 0) {
        // a third body for the genes that fill the register tier: a second variant of the tier INSIDE one body costs registers
        // the pass does not have (the allocator starts spilling Gram accumulators in the loop)
        // and a fourth (ONCHIP) for those that also fit the register + LDS tiers: no spill tier in it
        constexpr bool X = DN_REG_TIER != 0;
        const int nL_u = __builtin_amdgcn_readfirstlane(nL);
        if (X && n_u >= rt_cols<P, X>() * NT && n_u <= rt_cols<P, X>() * NT + nL_u)
            nmf_body<P, NT, X, X, X>(Fb_, Lg_, rs_, sv_, n, S, nL, T, first_i);
        else if (X && n_u >= rt_cols<P, X>() * NT) nmf_body<P, NT, X, X>(Fb_, Lg_, rs_, sv_, n, S, nL, T, first_i);
        else nmf_body<P, NT, X, false>(Fb_, Lg_, rs_, sv_, n, S, nL, T, first_i);
    } else nmf_body<P, NT, false>(Fb_, Lg_, rs_, sv_, n, S, nL, T, first_i);
    rt_restore<NSAVE, NT>(rtsave, need);
#ifdef DN_STAMP
    if (DN_TIDX == 0) g_gs.stamp[3] += __builtin_amdgcn_s_memtime() - t_call0;
#endif
}

// ---------------------------------------------------------------------------------------------------
// k_baseline: the per-gene state machine.  Gene-level vectors (rho, K, ...) are wave-uniform and live in
// LDS (GeneState) so that the registers belong to the Gram accumulators and the column in flight.
// ---------------------------------------------------------------------------------------------------

template <int P> __device__ __forceinline__ double lds_max(const double *v)
{ double m = v[0];
#pragma unroll
  for (int i = 1; i < P; i++) { const double t = v[i]; m = t > m ? t : m; } return m; }
template <int P> __device__ __forceinline__ double lds_min(const double *v)
{ double m = v[0];
#pragma unroll
  for (int i = 1; i < P; i++) { const double t = v[i]; m = t < m ? t : m; } return m; }

template <int P, int NT>
__global__ __launch_bounds__(NT * DN_UNITS, DN_KERNEL_WAVES) void k_baseline(IterArgs A)
{
    constexpr int W = NT / 64;
    Smem<P, NT> &sm = g_sm;
    GeneState<P> &gs = g_gs;
    const int tid = DN_TIDX, lane = lane_id(), w = wave_id();
    const int S = A.S;
    const int nL = A.lds_cols;
    char *slot = A.ws + ((size_t) blockIdx.x * DN_UNITS + DN_UNIT) * A.slot_bytes;
    float *Fs = reinterpret_cast<float *>(slot);                      // pristine compacted raw counts  [P][S]
    float *Fb = Fs + (size_t) P * S;                                  // working copy after bin drops   [S][P]
    double *Lg = reinterpret_cast<double *>(Fb + (size_t) P * S);     // x + lambda, spill tier         [S / 64][P][64]
    double *sv = Lg + (size_t) P * S;                                 // s_start                        [S]
    double *rs = sv + S;                                              // residual profile               [S]
    double *rtsave = rs + 2 * (size_t) S;                             // caller's AGPRs during an nmf() call  [RT * P][NT]
    DN_RT_CLAIM();
    if (tid < P) { gs.inv[tid] = A.inv_scale[tid]; gs.scl[tid] = 1.0 / A.inv_scale[tid]; }
    if (tid == 0) gs.max_steps = A.max_steps > 0 ? A.max_steps : EIG_MAX_STEPS_DEFAULT;
    if constexpr (P <= 16) { for (int t = tid; t < W * Smem<P, NT>::SQ_LEN; t += NT) sm.sq[t] = 0.0; }    // rows >= p of the square copies stay zero
    for (int t = tid; t < Smem<P, NT>::NX; t += NT) {                  // constants of the eigen-solver
        bool diag = false;
        int ra = 0, rb = 0;                                             // packed entry t = (ra, rb), rb <= ra
#pragma unroll
        for (int i = 0; i < P; i++) {
            diag = diag || (t == i * (i + 1) / 2 + i);
            if (t >= i * (i + 1) / 2) { ra = i; rb = t - i * (i + 1) / 2; }
        }
        if constexpr (P <= 16) sm.sqi[t] = (t < P * (P + 1) / 2) ? (ra * Smem<P, NT>::SQ_STR + rb) | ((rb * Smem<P, NT>::SQ_STR + ra) << 16) : 0;
        if constexpr (raw_units<P>())                                   // raw-unit Gram totals: the entry's scale, negative on the diagonal (shift_total)
            sm.dsel[t] = (t < P * (P + 1) / 2) ? (diag ? -1.0 : 1.0) * A.inv_scale[ra] * A.inv_scale[rb < P ? rb : 0] : 0.0;
        else sm.dsel[t] = diag ? 1.0 : 0.0;
        if (t == Smem<P, NT>::ZSLOT) sm.tot[t] = 0.0;
    }

    for (;;) {
        if (tid == 0) sm.gene = atomicAdd(A.counter, 1);
        dn_sync();
        const int q = sm.gene;
        dn_sync();
        if (q >= A.n_genes) break;
        const int g = A.order[q];
        const int L = A.glen[g];
        const float *x = A.cov + A.goff[g];

        int n0 = 0, n_calls = 0, n_drops = 0, exit_code = EXIT_LOW_COV, loop_reason = LOOP_NOT_ENTERED;
        int status = ST_OK, flag = 0, emode = EM_INPUT;
        long long sum_cols = 0;
#ifdef DN_STAMP
        const long long t_gene0 = __builtin_amdgcn_s_memtime();
        long long t_ret = 0, t_book = 0;
#endif
        int32_t *tr = A.trace + (size_t) g * TRACE_LEN;
        if (tid < P) { gs.rho[tid] = 0.0; gs.K[tid] = 0.0; gs.us[tid] = 0.0; }
        if (tid == 0) { gs.steps = 0; gs.stamp[0] = gs.stamp[1] = gs.stamp[2] = gs.stamp[3] = gs.stamp[4] = gs.stamp[5] = 0; }

        // ---- get_high_coverage_idx (nmf.py:66-76) on F = x / s (nmf.py:146) -------------------------
        // max_j fl(x_ij / s_i) = fl((max_j x_ij) / s_i): division by a positive scalar is monotone, so the global maximum
        // of the scaled matrix needs only the p row maxima of the RAW counts, computed once per upload (k_row_max).
        double thr;
        int x16;
        {
            const float *rmx = A.rowmax + (size_t) g * P;
            double gmax = (double) rmx[0] / A.scale[0];
#pragma unroll
            for (int i = 1; i < P; i++) { const double v = (double) rmx[i] / A.scale[i]; gmax = v > gmax ? v : gmax; }
            thr = 0.1 * gmax;
            x16 = A.x16[g];                                          // every raw count of the gene is a whole number that fits 16 bits
        }

        // Candidate columns: every base, or -- when down-sampling -- only the systematic sample ds0 + m * rate
        // (nmf.py:223-227: the high-coverage test is intersected with the sample, so columns outside it are never read).
        const int rate = A.rate;
        const long long ds0 = (rate > 1 && A.ds_start) ? A.ds_start[g] : -1;
        const int M = ds0 >= 0 ? (ds0 < L ? (int) ((L - 1 - ds0) / rate) + 1 : 0) : L;
        const int seg = ((M + W - 1) / W + 63) & ~63;
        const int jb = w * seg, je = (jb + seg < M) ? jb + seg : M;

        // pass 1: count per wave segment; pass 2: ordered compaction of the raw counts into Fs / Fb (nmf.py:236-238).
        // The threshold test uses the true quotient x / s so that ties resolve exactly as in the reference.
        {
            double sumF[P];
#pragma unroll
            for (int i = 0; i < P; i++) sumF[i] = 0.0;
            int base = 0;
            for (int pass = 0; pass < 2; pass++) {
                int run = 0;
                for (int c = jb; c < je; c += 64) {
                    const int m = c + lane;
                    bool hi = false;
                    float xv[P];
                    double f[P];
                    if (m < je) {
                        const long long j = ds0 >= 0 ? ds0 + (long long) m * rate : (long long) m;
                        double cm = 0.0;
#pragma unroll
                        for (int i = 0; i < P; i++) { xv[i] = x[(size_t) i * L + j]; f[i] = (double) xv[i] / A.scale[i]; cm = f[i] > cm ? f[i] : cm; }
                        hi = cm > thr;
                    }
                    const unsigned long long mask = __ballot(hi);
                    if (pass == 1 && hi) {
                        const int pos = base + run + __popcll(mask & ((1ull << lane) - 1ull));
#pragma unroll
                        for (int i = 0; i < P; i++) {
                            Fs[(size_t) i * S + pos] = xv[i];
                            Fb[(size_t) pos * P + i] = xv[i];
                            sumF[i] += f[i];
                        }
                    }
                    run += __popcll(mask);
                }
                if (pass == 0) {
                    if (lane == 0) sm.cnt[w] = run;
                    dn_sync();
                    n0 = 0;
#pragma unroll
                    for (int ww = 0; ww < W; ww++) { if (ww < w) base += sm.cnt[ww]; n0 += sm.cnt[ww]; }
                    dn_sync();
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
            dn_sync();
        }

#ifdef DN_STAMP
        const long long t_scan1 = __builtin_amdgcn_s_memtime();       // end of the candidate scan + compaction
#endif
        int n = n0;                   // current width of Fb
        if (n0 >= A.min_hc) {
            if (!(lds_min<P>(gs.sumF) > 0.0)) exit_code = EXIT_ZERO_SAMPLE;            // nmf.py:241
            else {
                const double min_gene_len = fmax(2.0, ceil(200.0 * (1.0 / (double) rate)));   // nmf.py:261
                const double min_bins = ceil((double) A.bins * 0.2);                         // nmf.py:35
                int csize = 1, n_bins = 0;
                bool first = true, in_loop = false;
                // One nmf() call per trip: the first on F_start (nmf.py:245), the others inside the
                // `while max(rho) > 0.1` loop of nmf.py:273-324 after a bin has been dropped.
                for (;;) {
#ifdef DN_STAMP
                    if (t_ret != 0) t_book += __builtin_amdgcn_s_memtime() - t_ret;      // between two nmf() calls
#endif
                    nmf_call<P, NT>(Fb, Lg, rs, sv, rtsave, n, S, nL, A.T, first ? 1 : 0, x16);       // results in gs (LDS)
                    if constexpr (P < DN_MG_MIN_P) {
                        // every call ends behind a barrier with its results in LDS.  When the eigenvector it ended with has a
                        // component below the threshold (warm_start_unsafe, above the solvers), some samples are (nearly) decoupled
                        // from the block it lives on and the warm-started solves may have stayed on a block that was overtaken; a
                        // solve that ran into its step cap is most likely two such blocks racing each other.  The call is repeated
                        // with block-by-block solves
                        const int st_call = __builtin_amdgcn_readfirstlane(gs.status);
                        bool unsafe = st_call == ST_NO_CONVERGENCE;
#ifdef DN_FORCE_SAFE                                                        // diagnostic: every call is repeated by the safe path
                        unsafe = unsafe || st_call == ST_OK;
#endif
                        if (st_call == ST_OK) {
#pragma unroll
                            for (int i = 0; i < P; i++)                      // a sample WITHOUT coverage in the active columns has u_i = 0 by right
                                unsafe = unsafe || (gs.u[i] < WARM_START_MIN_COMPONENT && gs.sums[1 + P + i] > 0.0);
                        }
                        if (__builtin_amdgcn_readfirstlane((int) unsafe) != 0) {
                            dn_sync();
                            nmf_call<P, NT, true>(Fb, Lg, rs, sv, rtsave, n, S, nL, A.T, first ? 1 : 0, x16);
                        }
                    }
#ifdef DN_STAMP
                    t_ret = __builtin_amdgcn_s_memtime();
#endif
                    if (gs.status != ST_OK) { status = gs.status; break; }
                    const double *u = gs.u, *sums = gs.sums;
                    const double theta = gs.theta;
                    n_calls++; sum_cols += n;
                    if (first) {
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
                        dn_sync();
                        double med;
                        {
                            double om[P];
#pragma unroll
                            for (int i = 0; i < P; i++) om[i] = 1.0 - gs.rho[i];
                            med = median_of<P>(om);
                        }
                        if (med > 1.0) { exit_code = EXIT_MEDIAN; break; }                              // nmf.py:257
                        emode = (n0 < L) ? EM_EXPAND : EM_RAW;
                        exit_code = EXIT_NO_LOOP;
                        if (!((double) n0 >= min_gene_len && lds_min<P>(gs.rho) <= 0.2 && !A.skip)) break;   // nmf.py:265
                        in_loop = true;
                        // split_into_chunks(range(n0), bins)   utils.py:176-192, nmf.py:269-271
                        csize = (n0 + A.bins - 1) / A.bins;
                        n_bins = (n0 + csize - 1) / csize;
                        if (tid < n_bins) sm.alive[tid] = tid;
                        dn_sync();
                        first = false;
                    } else {
                        // nmf.py:315, min row sum of K E == 0: in the reference K_i is EXACTLY zero precisely for a sample without
                        // coverage in the remaining columns (ARPACK takes its start vector through the operator once; u = A v / sigma
                        // when there are fewer columns than samples), and round-off-sized, not zero, for samples that are merely decoupled
                        // from the top block (tests/golden/sparse.npz, generated with the reference).  The solvers here leave something
                        // tiny for the former and exact zeros for the latter (solve_by_blocks), so test what the reference's zero means
                        bool zero_row = false;
#pragma unroll
                        for (int i = 0; i < P; i++) zero_row = zero_row || (sums[1 + P + i] == 0.0);
                        if (tid == 0) {
                            const double sg = sqrt(theta);
#pragma unroll
                            for (int i = 0; i < P; i++) {
                                gs.K[i] = u[i] * sg;                                             // nmf.py:307
                                if (!zero_row) gs.rho[i] = 1.0 - sums[1 + P + i] / (sums[1 + i] + 1.0);   // nmf.py:318-321
                            }
                        }
                        dn_sync();
                        if (zero_row) { loop_reason = LOOP_ZERO_ROWSUM; break; }                 // nmf.py:315
                        if ((double) n_bins <= min_bins || (double) n < min_gene_len) { loop_reason = LOOP_MIN_BINS; break; }  // nmf.py:323
                    }
                    if (!(lds_max<P>(gs.rho) > 0.1)) break;                                      // nmf.py:273
                    flag = 1;                                                                    // nmf.py:276
                    loop_reason = LOOP_NATURAL;
                    // per-bin mean of rs[] (nmf.py:283): one wave per bin, fixed order
                    // (the loads of four 64-column strips are issued together, then added in the order a one-by-one walk would
                    // use: the profile is in the scratch slot, and a dependent load -> add chain paid its latency per strip)
                    for (int b = w; b < n_bins; b += W) {
                        const int kb = b * csize, ke = (kb + csize < n) ? kb + csize : n;
                        double part = 0.0;
                        for (int k = kb + lane; k < ke; k += 4 * 64) {
                            double v[4];
#pragma unroll
                            for (int c = 0; c < 4; c++) v[c] = (k + 64 * c < ke) ? rs[k + 64 * c] : 0.0;
#pragma unroll
                            for (int c = 0; c < 4; c++) if (k + 64 * c < ke) part += v[c];
                        }
                        part = wave_sum1(part);
                        if (lane == 0) sm.ss[b] = part / (double) (ke - kb);
                    }
                    dn_sync();
                    double best = -INFINITY; int drop = 0;
                    for (int b = 0; b < n_bins; b++) { const double v = sm.ss[b]; if (v > best) { best = v; drop = b; } }   // nmf.py:291
                    dn_sync();
                    if (best == 0.0) { loop_reason = LOOP_PERFECT; break; }                      // nmf.py:286
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
                    dn_sync();
                    // columns in front of the dropped bin keep their place; the others are fetched again from the pristine
                    // copy, four columns per thread in flight
                    for (int k0 = kb + tid; k0 < n; k0 += 4 * NT) {
                        float v[4][P];
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const int k = (k0 + c * NT < n) ? k0 + c * NT : k0;
                            const int a = k / csize;
                            const int ko = sm.alive[a] * csize + (k - a * csize);
#pragma unroll
                            for (int i = 0; i < P; i++) v[c][i] = Fs[(size_t) i * S + ko];
                        }
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            if (k0 + c * NT < n) {
#pragma unroll
                                for (int i = 0; i < P; i++) Fb[(size_t) (k0 + c * NT) * P + i] = v[c][i];
                            }
                        }
                    }
                    dn_sync();
                    if (n < 2) { loop_reason = LOOP_VALUE_ERROR; break; }                        // svds ValueError, nmf.py:306-310
                }

                if (in_loop && status == ST_OK) {
                    bool fallback = false;
                    if (lds_max<P>(gs.rho) < 0.2) {                                              // nmf.py:327
                        double K[P];
#pragma unroll
                        for (int i = 0; i < P; i++) K[i] = gs.K[i];
                        status = fix_k<P>(K);                                                    // nmf.py:329-330
                        if (status == ST_OK) {
                            double se[1] = {0.0};
                            for (int k = tid; k < n0; k += NT) {                                 // nmf.py:333
                                double m = -INFINITY;
#pragma unroll
                                for (int i = 0; i < P; i++) {
                                    const double qv = ((double) Fs[(size_t) i * S + k] * gs.inv[i]) / K[i];
                                    m = qv > m ? qv : m;
                                }
                                se[0] += m;
                            }
                            block_sum<1, P, NT>(se, sm);
                            double rmax = -INFINITY;
#pragma unroll
                            for (int i = 0; i < P; i++) {
                                const double r = 1.0 - gs.sumF[i] / (K[i] * se[0] + 1.0);         // nmf.py:334-337
                                rmax = r > rmax ? r : rmax;
                            }
                            dn_sync();
                            if (rmax > 0.9) { fallback = true; exit_code = EXIT_REFINE_FALLBACK; }              // nmf.py:342
                            else {
                                exit_code = EXIT_REFINED; emode = (n0 < L) ? EM_EXPAND : EM_REFINED;
                                if (tid == 0) {
#pragma unroll
                                    for (int i = 0; i < P; i++) { gs.K[i] = K[i]; gs.rho[i] = 1.0 - gs.sumF[i] / (K[i] * se[0] + 1.0); }
                                }
                            }
                        }
                    } else { fallback = true; exit_code = EXIT_NOT_FOUND_FALLBACK; }              // nmf.py:349
                    if (fallback && status == ST_OK) {
                        if (tid == 0) {
#pragma unroll
                            for (int i = 0; i < P; i++) { gs.K[i] = gs.us[i] * gs.sig0; gs.rho[i] = gs.rho_fb[i]; }
                        }
                        emode = (n0 < L) ? EM_EXPAND : EM_CLAMPED;
                    }
                    dn_sync();
                }
                // the re-expansion fix-up runs (and may raise) whenever the estimate is narrower than F  nmf.py:358-362
                if (status == ST_OK && exit_code >= EXIT_NO_LOOP && n0 < L) {
                    double K[P];
#pragma unroll
                    for (int i = 0; i < P; i++) K[i] = gs.K[i];
                    status = fix_k<P>(K);
                    dn_sync();
                    if (tid == 0) {
#pragma unroll
                        for (int i = 0; i < P; i++) gs.K[i] = K[i];
                    }
                }
            }
        }
        dn_sync();

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
            tr[5] = n_drops; tr[6] = status; tr[7] = gs.steps;
#ifdef DN_STAMP
            // diagnostic build only: kilo-cycles spent in the pass / reduction / eigen-solver / whole gene
            tr[40] = (int32_t) (gs.stamp[0] >> 10); tr[41] = (int32_t) (gs.stamp[1] >> 10); tr[42] = (int32_t) (gs.stamp[2] >> 10);
            tr[44] = (int32_t) (gs.stamp[3] >> 10); tr[45] = (int32_t) (gs.stamp[4] >> 10); tr[46] = (int32_t) (gs.stamp[5] >> 10);
            tr[43] = (int32_t) ((__builtin_amdgcn_s_memtime() - t_gene0) >> 10);
            tr[47] = (int32_t) ((t_scan1 - t_gene0) >> 10);
            tr[39] = (int32_t) (t_book >> 10);
            tr[38] = (int32_t) ((t_ret != 0 ? __builtin_amdgcn_s_memtime() - t_ret : 0) >> 10);
#endif
        }
        if (A.want_est && (emode == EM_CLAMPED || emode == EM_RAW)) {
            double *dst = A.svec + A.svoff[g];
            for (int k = tid; k < n0; k += NT) dst[k] = sv[k];
        }
        dn_sync();
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
    Smem<P, NT> &sm = g_sm;
    const int tid = DN_TIDX;
    for (;;) {
        if (tid == 0) sm.gene = atomicAdd(A.counter, 1);
        dn_sync();
        const int q = sm.gene;
        dn_sync();
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
                const int maxs = A.max_steps > 0 ? A.max_steps : EIG_MAX_STEPS_DEFAULT;
                if (top_eig<P>(G, u, theta, maxs) > maxs) status = ST_NO_CONVERGENCE;
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
        dn_sync();
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
    double *o = A.out + (A.ooff ? A.ooff[g] : A.goff[g]);
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

#endif  // DN_P

// Launchers instantiated per P in dn_inst.hip ---------------------------------------------------------
typedef int (*baseline_launch_fn)(const IterArgs &, int grid, size_t dyn_lds_bytes, hipStream_t);
typedef void (*init_launch_fn)(const InitArgs &, int grid, hipStream_t);
typedef void (*est_launch_fn)(const EstArgs &, const int32_t *, const int32_t *, int n_tiles, hipStream_t);
typedef int (*occupancy_fn)(int which);

struct KernelSet {
    int p;
    int nt;
    baseline_launch_fn baseline;
    init_launch_fn init;
    est_launch_fn est;
    occupancy_fn blocks_per_cu;       // which: 0 baseline (no dynamic LDS), 1 init (power iteration), 2 init (k_ratio_svd_mg, p >= 17)
    size_t static_lds_bytes;          // static LDS of k_baseline
    const char *baseline_name;
    size_t slot_extra_bytes;          // per scratch slot, behind the S-sized arrays (register-tier save area)
    int reg_tier_cols;                // columns of a gene a workgroup (a unit) keeps in registers (0: no register tier)
    int units;                        // genes a workgroup carries at once (2: the pair build, one gene per wavefront; else 1).
                                      // `nt` threads, the LDS tile and a scratch slot belong to ONE unit; `static_lds_bytes`,
                                      // blocks_per_cu() and the dynamic LDS of a launch to the workgroup
};

const KernelSet *kernel_set_for(int p);   // dn_api.hip
const KernelSet *kernel_set_generic();    // dn_generic.hip (run-time p, any p <= P_MAX)

}  // namespace dn
