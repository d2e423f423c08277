// dn_api.hip -- C ABI (include/degnorm_amd.h) over the kernels of dn_kernels.hpp: device handle, host packer,
// resident buffers, launches.  Host side only; the kernels are instantiated per sample count in dn_inst.hip.
#include "dn_kernels.hpp"
#include "../../include/degnorm_amd.h"

#include <dlfcn.h>
#include <rccl/rccl.h>          // types and constants only: the entry points are resolved at run time (rccl_api below)

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

// sample counts with compiled kernels (dn_inst.hip is built once per entry; keep in sync with build.py)
#ifndef DN_P_MAX_TEMPLATED
#define DN_P_MAX_TEMPLATED 64        // templated kernels for 2 .. this many samples (build.py compiles the same list)
#endif
#define DN_P_2_16(X)  X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
#define DN_P_17_24(X) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)
#define DN_P_25_32(X) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)
#define DN_P_33_48(X) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40) X(41) X(42) X(43) X(44) X(45) X(46) X(47) X(48)
#define DN_P_49_64(X) X(49) X(50) X(51) X(52) X(53) X(54) X(55) X(56) X(57) X(58) X(59) X(60) X(61) X(62) X(63) X(64)
#if DN_P_MAX_TEMPLATED >= 64
#define DN_FOR_EACH_P(X) DN_P_2_16(X) DN_P_17_24(X) DN_P_25_32(X) DN_P_33_48(X) DN_P_49_64(X)
#elif DN_P_MAX_TEMPLATED >= 48
#define DN_FOR_EACH_P(X) DN_P_2_16(X) DN_P_17_24(X) DN_P_25_32(X) DN_P_33_48(X)
#elif DN_P_MAX_TEMPLATED >= 32
#define DN_FOR_EACH_P(X) DN_P_2_16(X) DN_P_17_24(X) DN_P_25_32(X)
#elif DN_P_MAX_TEMPLATED >= 24
#define DN_FOR_EACH_P(X) DN_P_2_16(X) DN_P_17_24(X)
#elif DN_P_MAX_TEMPLATED >= 16
#define DN_FOR_EACH_P(X) DN_P_2_16(X)
#else
#error "DN_P_MAX_TEMPLATED must be 16, 24, 32, 48 or 64"
#endif

namespace dn {
#ifndef DN_WIDE_NT
#define DN_WIDE_NT 256               // threads per workgroup of the wide gene class (build.py compiles this variant)
#endif
#define DN_CAT4_(a, b, c, d) a##b##c##d
#define DN_CAT4(a, b, c, d) DN_CAT4_(a, b, c, d)
#define DN_WIDE_SET(P) DN_CAT4(kernel_set_p, P, _nt, DN_WIDE_NT)
#define DN_DECL(P) const KernelSet *DN_WIDE_SET(P)(); const KernelSet *kernel_set_p##P##_nt128();
DN_FOR_EACH_P(DN_DECL)
#undef DN_DECL

// Pair build (dn_inst.hip -DDN_PAIR): 128-thread workgroups carrying two genes, one per wavefront, for the shortest genes;
// compiled where the register tier exists
#ifndef DN_P_PAIR                 // build.py passes the list it compiles (-D'DN_P_PAIR(X)=X(8) ...'): ONE source of truth
#define DN_P_PAIR(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12)
#endif
#define DN_DECL(P) const KernelSet *kernel_set_p##P##_pair();
DN_P_PAIR(DN_DECL)
#undef DN_DECL
const KernelSet *kernel_set_pair(int p)
{
    switch (p) {
#define DN_CASE(P) case P: return kernel_set_p##P##_pair();
        DN_P_PAIR(DN_CASE)
#undef DN_CASE
        default: return nullptr;
    }
}

// 128-thread workgroups (two per CU) for the narrow gene class
const KernelSet *kernel_set_narrow(int p)
{
    switch (p) {
#define DN_CASE(P) case P: return kernel_set_p##P##_nt128();
        DN_FOR_EACH_P(DN_CASE)
#undef DN_CASE
        default: return nullptr;
    }
}

const KernelSet *kernel_set_rows();       // dn_generic.hip compiled with DN_GEN_NT=64: one wavefront per gene

const KernelSet *kernel_set_for(int p)
{
    // DN_FORCE_GENERIC=1 routes every supported p through the run-time-p kernels (used by the tests to
    // cross-check the two kernel families on the same inputs)
    const char *force = getenv("DN_FORCE_GENERIC");
    if (force && force[0] == '1' && p >= 2 && p <= P_MAX) return kernel_set_generic();
    switch (p) {
#define DN_CASE(P) case P: return DN_WIDE_SET(P)();
        DN_FOR_EACH_P(DN_CASE)
#undef DN_CASE
        default: return (p > DN_P_MAX_TEMPLATED && p <= P_MAX) ? kernel_set_generic() : nullptr;
    }
}
}  // namespace dn

// Row maxima of the raw coverage, once per upload: max_j fl(x_ij / s_i) = fl((max_j x_ij) / s_i) for s_i > 0, so the
// 0.1 * max(F) threshold of get_high_coverage_idx (nmf.py:66-76) needs these p numbers per gene instead of a scan of
// the whole scaled matrix in every outer iteration (SURVEY 8(d), config-4 regime note).  One workgroup per gene, one wave per
// row at a time.  Round 4: a pure max-reduce has to stream -- a wave keeps FOUR 16-byte loads per lane in flight (4 KiB per wave
// and trip; rows are only 4-byte aligned, so a scalar head brings the wave to a 16-byte boundary first) instead of one dword
// (round 3: 2.77 TB/s on config 4's 27.5 GB).
__device__ __forceinline__ void row_max_acc(float v, float &m, bool &ok)
{
    m = fmaxf(m, v);
    ok = ok && (v >= 0.0f) && (v <= 65535.0f) && (v == truncf(v));
}

__global__ __launch_bounds__(256) void k_row_max(const float *__restrict__ cov, const int64_t *__restrict__ goff,
                                                 const int32_t *__restrict__ glen, float *__restrict__ rowmax,
                                                 int32_t *__restrict__ x16, int n, int p)
{
    __shared__ int ok_w[4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int g = blockIdx.x; g < n; g += gridDim.x) {
        const int L = glen[g];
        bool ok = true;                                    // every value a whole number in [0, 65535]: fits 16 bits exactly
        for (int i = w; i < p; i += 4) {
            const float *row = cov + goff[g] + (size_t) i * L;
            float m = row[0];
            const int head = (int) (((16u - (unsigned) ((size_t) row & 15u)) & 15u) >> 2);     // floats up to the next 16-byte boundary
            const int h = head < L ? head : L;
            if (lane < h) row_max_acc(row[lane], m, ok);
            const float4 *q = reinterpret_cast<const float4 *>(row + h);
            const int n4 = (L - h) >> 2;
            int j = lane;
            typedef float vf4 __attribute__((ext_vector_type(4)));
            const vf4 *qv = reinterpret_cast<const vf4 *>(q);
            for (; j + 448 < n4; j += 512) {               // eight independent 16-byte loads per lane in flight, read once: non-temporal
                vf4 v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(qv + j + 64 * k);
#pragma unroll
                for (int k = 0; k < 8; k++) { row_max_acc(v[k].x, m, ok); row_max_acc(v[k].y, m, ok); row_max_acc(v[k].z, m, ok); row_max_acc(v[k].w, m, ok); }
            }
            for (; j + 192 < n4; j += 256) {               // four
                const float4 a = q[j], b = q[j + 64], c = q[j + 128], d = q[j + 192];
                row_max_acc(a.x, m, ok); row_max_acc(a.y, m, ok); row_max_acc(a.z, m, ok); row_max_acc(a.w, m, ok);
                row_max_acc(b.x, m, ok); row_max_acc(b.y, m, ok); row_max_acc(b.z, m, ok); row_max_acc(b.w, m, ok);
                row_max_acc(c.x, m, ok); row_max_acc(c.y, m, ok); row_max_acc(c.z, m, ok); row_max_acc(c.w, m, ok);
                row_max_acc(d.x, m, ok); row_max_acc(d.y, m, ok); row_max_acc(d.z, m, ok); row_max_acc(d.w, m, ok);
            }
            for (; j < n4; j += 64) {
                const float4 a = q[j];
                row_max_acc(a.x, m, ok); row_max_acc(a.y, m, ok); row_max_acc(a.z, m, ok); row_max_acc(a.w, m, ok);
            }
            const int t0 = h + 4 * n4;                     // scalar tail (< 4 floats)
            if (t0 + lane < L) row_max_acc(row[t0 + lane], m, ok);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
            if (lane == 0) rowmax[(size_t) g * p + i] = m;
        }
        const int all_ok = __all(ok);
        if (lane == 0) ok_w[w] = all_ok;
        __syncthreads();
        if (threadIdx.x == 0) x16[g] = (ok_w[0] && ok_w[1] && ok_w[2] && ok_w[3]) ? 1 : 0;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// The outer DegNorm update on the device (nmf.py:398-399, :148-158, :575-590; nmf_mpi.py:821-838).  Per outer iteration
// the host needs 3p + 4 numbers, not the n x p DI matrix: one wave per gene, lane i = sample i.
//   k_outer_partials  clip rho to [0, 0.9]; untouched = (max_i rho == 0); per-sample partial sums
//                     A = sum_touched x_w / (1 - rho), B = sum_untouched x_w, W = sum x_w; counts of untouched / failed /
//                     unconverged genes.  Block partials, then k_outer_reduce adds them in block order (deterministic).
//   k_outer_apply     rho[untouched] = avg_di; x_adj = x_w / (1 - rho); x_w /= norm; ran_baseline_selection[:, iter].
// ---------------------------------------------------------------------------------------------------
constexpr int OUT_BLOCKS = 512;
constexpr int OUT_STRIDE = 3 * 64 + 4;          // per block: A[64] B[64] W[64] n_untouched n_failed n_noconv n_flagged

__device__ __forceinline__ double wave_max_d(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const double t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}

__global__ __launch_bounds__(256) void k_outer_partials(const double *__restrict__ rho_raw, double *__restrict__ rho_c,
                                                        const double *__restrict__ xw, const int32_t *__restrict__ trace,
                                                        const int32_t *__restrict__ flags, double *__restrict__ part, int n, int p)
{
    __shared__ double sm[4][3][64];
    __shared__ double cnt[4][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nw = gridDim.x * 4;
    double a = 0.0, b = 0.0, ws = 0.0, nu = 0.0, nf = 0.0, nc = 0.0, nr = 0.0;
    for (int g = blockIdx.x * 4 + w; g < n; g += nw) {
        double r = 0.0, x = 0.0;
        if (lane < p) {
            r = rho_raw[(size_t) g * p + lane];
            r = r > 0.9 ? 0.9 : r;                                        // nmf.py:398
            r = r < 0.0 ? 0.0 : r;                                        // nmf.py:399
            rho_c[(size_t) g * p + lane] = r;
            x = xw[(size_t) g * p + lane];
        }
        const bool untouched = wave_max_d(r) == 0.0;                      // nmf.py:155
        a += untouched ? 0.0 : x / (1.0 - r);                             // nmf.py:575
        b += untouched ? x : 0.0;
        ws += x;
        if (lane == 0) {
            const int st = trace[(size_t) g * dn::TRACE_LEN + 6];
            nu += untouched ? 1.0 : 0.0; nf += st != 0 ? 1.0 : 0.0; nc += st == dn::ST_NO_CONVERGENCE ? 1.0 : 0.0;
            nr += flags[g] != 0 ? 1.0 : 0.0;                              // ran_baseline_selection[:, i].sum() (nmf.py:571)
        }
    }
    sm[w][0][lane] = a; sm[w][1][lane] = b; sm[w][2][lane] = ws;
    if (lane == 0) { cnt[w][0] = nu; cnt[w][1] = nf; cnt[w][2] = nc; cnt[w][3] = nr; }
    __syncthreads();
    if (threadIdx.x < 192) {
        const int k = threadIdx.x >> 6, i = threadIdx.x & 63;
        part[(size_t) blockIdx.x * OUT_STRIDE + 64 * k + i] = ((sm[0][k][i] + sm[1][k][i]) + sm[2][k][i]) + sm[3][k][i];
    } else if (threadIdx.x < 196) {
        const int k = threadIdx.x - 192;
        part[(size_t) blockIdx.x * OUT_STRIDE + 192 + k] = ((cnt[0][k] + cnt[1][k]) + cnt[2][k]) + cnt[3][k];
    }
}

// The initial normalisation on the device (nmf.py:524-531): rho0 = 1 - cov_sums / (est_sums + 1) per gene, `low` =
// (max_i rho0 < 0.1), per-sample sums of the read counts over the low genes and over all genes, the number of low genes and
// of genes whose initial SVD failed.  Same block-partial layout as k_outer_partials (A = low sums, B = all sums, W unused).
__global__ __launch_bounds__(256) void k_init_partials(const double *__restrict__ est, const double *__restrict__ cov,
                                                       const int32_t *__restrict__ status, const double *__restrict__ x,
                                                       double *__restrict__ part, int n, int p)
{
    __shared__ double sm[4][3][64];
    __shared__ double cnt[4][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nw = gridDim.x * 4;
    double a = 0.0, b = 0.0, nl = 0.0, nb = 0.0;
    for (int g = blockIdx.x * 4 + w; g < n; g += nw) {
        double r = -INFINITY, xv = 0.0;
        if (lane < p) {
            r = 1.0 - cov[(size_t) g * p + lane] / (est[(size_t) g * p + lane] + 1.0);       // nmf.py:526
            xv = x[(size_t) g * p + lane];
        }
        const bool low = wave_max_d(r) < 0.1;                             // nmf.py:529
        a += low ? xv : 0.0;
        b += xv;
        if (lane == 0) { nl += low ? 1.0 : 0.0; nb += status[g] != 0 ? 1.0 : 0.0; }
    }
    sm[w][0][lane] = a; sm[w][1][lane] = b; sm[w][2][lane] = 0.0;
    if (lane == 0) { cnt[w][0] = nl; cnt[w][1] = nb; cnt[w][2] = 0.0; cnt[w][3] = 0.0; }
    __syncthreads();
    if (threadIdx.x < 192) {
        const int k = threadIdx.x >> 6, i = threadIdx.x & 63;
        part[(size_t) blockIdx.x * OUT_STRIDE + 64 * k + i] = ((sm[0][k][i] + sm[1][k][i]) + sm[2][k][i]) + sm[3][k][i];
    } else if (threadIdx.x < 196) {
        const int k = threadIdx.x - 192;
        part[(size_t) blockIdx.x * OUT_STRIDE + 192 + k] = ((cnt[0][k] + cnt[1][k]) + cnt[2][k]) + cnt[3][k];
    }
}

// x_weighted = x / norm (nmf.py:533)
__global__ __launch_bounds__(256) void k_scale_reads(const double *__restrict__ x, const double *__restrict__ norm,
                                                     double *__restrict__ xw, long long np, int p)
{
    for (long long i = (long long) blockIdx.x * 256 + threadIdx.x; i < np; i += (long long) gridDim.x * 256)
        xw[i] = x[i] / norm[i % p];
}

__global__ __launch_bounds__(256) void k_outer_reduce(const double *__restrict__ part, double *__restrict__ out, int nblocks, int p)
{
    const int t = threadIdx.x;
    if (t >= 196) return;
    // eight independent running sums (block b goes to sum b % 8: eight loads in flight instead of a chain of nblocks dependent ones
    // -- 117 us for 512 blocks in round 3), combined in one fixed order: the result does not depend on timing
    double a[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
    int b = 0;
    for (; b + 8 <= nblocks; b += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] += part[(size_t) (b + k) * OUT_STRIDE + t];
    }
    for (int k = 0; b < nblocks; b++, k++) a[k] += part[(size_t) b * OUT_STRIDE + t];
    const double s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    const int k = t >> 6, i = t & 63;
    if (t < 192) { if (i < p) out[k * p + i] = s; }
    else out[3 * p + (t - 192)] = s;
}

__global__ __launch_bounds__(256) void k_outer_apply(double *__restrict__ rho_c, double *__restrict__ xw, double *__restrict__ xadj,
                                                     const int32_t *__restrict__ flags, uint8_t *__restrict__ ran,
                                                     const double *__restrict__ avg_di, const double *__restrict__ norm,
                                                     int have_avg, int n, int p, int iter, int n_iter)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nw = gridDim.x * 4;
    for (int g = blockIdx.x * 4 + w; g < n; g += nw) {
        double r = lane < p ? rho_c[(size_t) g * p + lane] : 0.0;
        const bool untouched = wave_max_d(r) == 0.0;
        if (lane < p) {
            if (untouched && have_avg) r = avg_di[lane];                  // correct_di_scores, nmf.py:157
            const double x = xw[(size_t) g * p + lane];
            rho_c[(size_t) g * p + lane] = r;
            xadj[(size_t) g * p + lane] = x / (1.0 - r);                  // nmf.py:581
            xw[(size_t) g * p + lane] = x / norm[lane];                   // nmf.py:587
        }
        if (lane == 0 && iter < n_iter) ran[(size_t) g * n_iter + iter] = flags[g] != 0;      // nmf.py:403
    }
}

__global__ __launch_bounds__(256) void k_gather_rows(const double *__restrict__ rho_raw, const int32_t *__restrict__ flags,
                                                     const int64_t *__restrict__ rows, double *__restrict__ out_rho,
                                                     int32_t *__restrict__ out_flags, int n_rows, int p)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * p) return;
    const int k = t / p, i = t - k * p;
    out_rho[t] = rho_raw[(size_t) rows[k] * p + i];
    if (i == 0) out_flags[k] = flags[rows[k]];
}

// the first `cols` counters of every gene, packed: what comes back to the host per iteration when the caller does not want the
// whole trace (dn_set_trace_columns)
__global__ __launch_bounds__(256) void k_trace_head(const int32_t *__restrict__ trace, int32_t *__restrict__ head, long long n_cols_total, int cols)
{
    for (long long t = (long long) blockIdx.x * 256 + threadIdx.x; t < n_cols_total; t += (long long) gridDim.x * 256) {
        const long long g = t / cols;
        head[t] = trace[g * dn::TRACE_LEN + (t - g * cols)];
    }
}

static thread_local std::string g_err;

static int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(DN_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));              \
    } while (0)

struct dn_handle_s {
    int device = -1;
    int n_cus = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const dn::KernelSet *ks = nullptr;

    int64_t n = 0;
    int32_t p = 0;
    int64_t total = 0;            // floats in the packed coverage
    int32_t lmax = 0;
    std::vector<int64_t> goff;    // n + 1
    std::vector<int32_t> glen;
    std::vector<int64_t> svoff;   // n + 1, in columns

    float   *d_cov = nullptr;
    int64_t *d_goff = nullptr;
    int32_t *d_glen = nullptr;
    int32_t *d_order = nullptr;
    int32_t *d_counter = nullptr;
    int64_t *d_ds = nullptr;
    char    *d_ws = nullptr;          // alias of cls[0].d_ws
    double  *d_rho = nullptr;
    int32_t *d_flags = nullptr;
    int32_t *d_trace = nullptr;
    double  *d_kfin = nullptr;
    int32_t *d_emode = nullptr;
    double  *d_svec = nullptr;
    int64_t *d_svoff = nullptr;
    float   *d_rowmax = nullptr;      // n x p row maxima of the raw coverage (k_row_max at upload)
    int32_t *d_x16 = nullptr;         // n: 1 when every count of the gene is a whole number <= 65535 (packable into 16 bits)
    // outer-update state (dn_outer_begin): clipped / corrected DI, x_weighted, x_adj, ran_baseline_selection, partial sums
    double  *d_rhoc = nullptr, *d_xw = nullptr, *d_xadj = nullptr, *d_part = nullptr, *d_pvec = nullptr;
    double  *d_x = nullptr;           // the read counts (n x p), resident for the device-side initial normalisation (dn_init_begin)
    uint8_t *d_ran = nullptr;
    int32_t  n_iter = 0;
    double  *d_est_sums = nullptr, *d_cov_sums = nullptr;
    int32_t *d_status = nullptr;
    double  *d_est = nullptr;
    int32_t *d_tile_gene = nullptr, *d_tile_col = nullptr;
    int64_t n_tiles = 0;

    // Gene classes: [0] wide genes on the main kernel set (256-thread workgroups, one per CU, the whole LDS for one
    // gene), [1] narrow genes (length <= split_len) on 128-thread workgroups, two per CU, so that the serial
    // reduction / eigen-solver phases of one gene overlap the pass of another.  Each class has its own longest-first
    // queue, scratch slots and stream; class 0 is launched first and class 1 fills the CUs as they drain.
    struct GeneClass {
        const dn::KernelSet *ks = nullptr;
        int32_t n = 0;
        int32_t longest = 0;              // longest gene of the class
        int32_t *d_order = nullptr;
        std::vector<int32_t> order;       // host copy of the work queue (gene ids)
        int32_t *d_counter = nullptr;
        char *d_ws = nullptr;
        int slots = 0;
        int32_t S = 0;
        int64_t slot_bytes = 0;
        int32_t lds_cols = 0;
        size_t dyn_lds = 0;
        float last_ms = 0.f;
    };
    static constexpr int NCLS = 3;   // [2]: the shortest genes (length <= tiny_len), one wavefront per gene, two genes per 128-thread
                                     // workgroup (the pair build): no cross-wave step, the serial phases are paid by ONE SIMD
    GeneClass cls[NCLS];
    int32_t split_len = 0, tiny_len = 0;
    hipStream_t stream2 = nullptr, stream3 = nullptr;
    hipEvent_t ev2a = nullptr, ev2b = nullptr, ev3a = nullptr, ev3b = nullptr, ev_ready = nullptr;
    hipStream_t class_stream(int c) const { return c == 0 ? stream : (c == 1 ? stream2 : stream3); }
    hipEvent_t class_ev_a(int c) const { return c == 0 ? ev0 : (c == 1 ? ev2a : ev3a); }
    hipEvent_t class_ev_b(int c) const { return c == 0 ? ev1 : (c == 1 ? ev2b : ev3b); }
    int slots = 0;              // class 0 (kept for the run-time-p init kernel)
    int32_t S = 0;
    int64_t slot_bytes = 0;
    double last_scale[dn::P_MAX] = {0};
    bool have_estimate_state = false;
    float last_ms = 0.f;
    float last_span_ms = 0.f;     // first launch to last end of the class kernels of the most recent dn_baseline_iteration
    float last_init_ms = 0.f;     // device time of the most recent dn_ratio_svd_sums kernel
    float last_rowmax_ms = 0.f;   // device time of k_row_max at the most recent upload
    char init_name[64] = {0};
    hipEvent_t ev_i0 = nullptr, ev_i1 = nullptr;
    // the collective inside the library (dn_comm_*): one RCCL communicator per handle, all-reduces on the handle's stream
    void *comm = nullptr;         // ncclComm_t
    int32_t comm_rank = 0, comm_size = 0;
    double *d_comm = nullptr;     // 256 doubles of scratch (ranks without genes, small host vectors)
    int32_t ds_hint = 1;          // take-every rate the caller intends to use (dn_set_downsample_hint); 1 = none
    int32_t max_steps = dn::EIG_MAX_STEPS_DEFAULT;   // step cap of one eigen-solve (dn_set_solver_step_cap)
    // per-gene counters of the previous dn_baseline_iteration: the narrow class orders its queue by the work they predict
    int32_t *host_trace = nullptr;        // pinned (hipHostMalloc): the per-gene counters come back every iteration (n x trace_cols)
    size_t host_trace_len = 0;
    int32_t trace_cols = dn::TRACE_LEN;   // leading trace columns copied back per iteration (dn_set_trace_columns)
    int32_t *d_trace_head = nullptr;      // n x trace_cols, packed (only when trace_cols < TRACE_LEN)
    bool have_trace = false;

};

// ---------------------------------------------------------------------------------------------------
// RCCL, resolved at run time.  The library carries no link-time dependency on librccl: a host process that already
// has one mapped (PyTorch ships its own copy) must share THAT copy -- two RCCL instances in one process each bring their
// own proxy threads and IPC state --, and a host without any multi-GPU use never loads it.  Order: a copy already in
// the process (RTLD_NOLOAD), DN_RCCL_PATH, the system's librccl.so.1.
// ---------------------------------------------------------------------------------------------------
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    std::string path;
};

static RcclApi *rccl_api(std::string &err)
{
    static RcclApi api;
    static bool tried = false;
    static std::string load_err;
    if (!tried) {
        tried = true;
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names) if (!api.lib) { api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (api.lib) api.path = std::string(n) + " (already in the process)"; }
        if (!api.lib) { const char *e = getenv("DN_RCCL_PATH"); if (e && *e) { api.lib = dlopen(e, RTLD_NOW | RTLD_LOCAL); if (api.lib) api.path = e; } }
        const char *sys[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *n : sys) if (!api.lib) { api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (api.lib) api.path = n; }
        if (!api.lib) load_err = std::string("librccl not found (") + (dlerror() ? dlerror() : "?") + "); set DN_RCCL_PATH";
        else {
            api.GetUniqueId = (decltype(api.GetUniqueId)) dlsym(api.lib, "ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank)) dlsym(api.lib, "ncclCommInitRank");
            api.CommDestroy = (decltype(api.CommDestroy)) dlsym(api.lib, "ncclCommDestroy");
            api.AllReduce = (decltype(api.AllReduce)) dlsym(api.lib, "ncclAllReduce");
            api.GetErrorString = (decltype(api.GetErrorString)) dlsym(api.lib, "ncclGetErrorString");
            api.GetVersion = (decltype(api.GetVersion)) dlsym(api.lib, "ncclGetVersion");
            if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GetErrorString) {
                load_err = "librccl (" + api.path + ") lacks an entry point of the nccl API";
                api.lib = nullptr;
            }
        }
    }
    if (!api.lib) { err = load_err; return nullptr; }
    return &api;
}

static void free_device(dn_handle h)
{
    void *ptrs[] = {h->d_cov, h->d_goff, h->d_glen, h->d_order, h->d_counter, h->d_ds, h->d_ws, h->d_rho, h->d_flags,
                    h->d_trace, h->d_kfin, h->d_emode, h->d_svec, h->d_svoff, h->d_est_sums, h->d_cov_sums,
                    h->d_status, h->d_est, h->d_tile_gene, h->d_tile_col, h->d_rowmax, h->d_x16, h->d_rhoc, h->d_xw, h->d_xadj,
                    h->d_part, h->d_pvec, h->d_ran, h->d_x};
    for (void *q : ptrs) if (q && q != (void *) h->cls[0].d_ws) (void) hipFree(q);
    if (h->host_trace) { (void) hipHostFree(h->host_trace); h->host_trace = nullptr; h->host_trace_len = 0; }
    if (h->d_trace_head) { (void) hipFree(h->d_trace_head); h->d_trace_head = nullptr; }
    for (auto &c : h->cls) {
        if (c.d_order) (void) hipFree(c.d_order);
        if (c.d_counter) (void) hipFree(c.d_counter);
        if (c.d_ws) (void) hipFree(c.d_ws);
        c = dn_handle_s::GeneClass();
    }
    h->d_cov = nullptr; h->d_goff = nullptr; h->d_glen = nullptr; h->d_order = nullptr; h->d_counter = nullptr;
    h->d_ds = nullptr; h->d_ws = nullptr; h->d_rho = nullptr; h->d_flags = nullptr; h->d_trace = nullptr;
    h->d_kfin = nullptr; h->d_emode = nullptr; h->d_svec = nullptr; h->d_svoff = nullptr; h->d_est_sums = nullptr;
    h->d_cov_sums = nullptr; h->d_status = nullptr; h->d_est = nullptr; h->d_tile_gene = nullptr; h->d_tile_col = nullptr;
    h->d_rowmax = nullptr; h->d_x16 = nullptr;
    h->d_rhoc = nullptr; h->d_xw = nullptr; h->d_xadj = nullptr; h->d_part = nullptr; h->d_pvec = nullptr; h->d_ran = nullptr;
    h->d_x = nullptr;
    h->n_iter = 0;
    h->have_estimate_state = false;
}

extern "C" {

#ifndef DN_BUILD_STAMP
#define DN_BUILD_STAMP "unknown build"
#endif
// the library, its target, and the build it is: compiler version and code-generation flags (degnorm_amd/build.py build_stamp).
// Results are bit-reproducible per binary; what may differ between two builds is named here.
const char *dn_version(void) { return "degnorm_amd 0.2.0 (gfx950) [" DN_BUILD_STAMP "]"; }
const char *dn_last_error(void) { return g_err.c_str(); }

int dn_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int dn_p_supported(int p) { return dn::kernel_set_for(p) != nullptr; }

static int create_streams(dn_handle h)
{
    // The class kernels of a sweep are queued together on three streams and must start in class order: the wide class needs
    // whole CUs, and a CU that a narrow or pair workgroup reaches first is lost to it until that (persistent) workgroup has
    // emptied its queue (seen: the wide kernel then ends LAST, sweep 291 -> 307 ms).  Stream priorities make the dispatcher
    // place pending wide workgroups first, then narrow ones, then pairs.
    int prio_least = 0, prio_greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    const int prio_mid = (prio_least + prio_greatest) / 2;
    HIP_TRY(hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prio_greatest));
    HIP_TRY(hipEventCreate(&h->ev0));
    HIP_TRY(hipEventCreate(&h->ev1));
    HIP_TRY(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, prio_mid));
    HIP_TRY(hipEventCreate(&h->ev2a));
    HIP_TRY(hipEventCreate(&h->ev2b));
    HIP_TRY(hipStreamCreateWithPriority(&h->stream3, hipStreamNonBlocking, prio_least));
    HIP_TRY(hipEventCreate(&h->ev3a));
    HIP_TRY(hipEventCreate(&h->ev3b));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&h->ev_i0));
    HIP_TRY(hipEventCreate(&h->ev_i1));
    return DN_OK;
}

int dn_create(int device, dn_handle *out)
{
    if (!out) return fail(DN_E_INVALID, "dn_create: out is null");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DN_E_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(DN_E_INVALID, "dn_create: device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    dn_handle h = new dn_handle_s();
    h->device = device;
    h->n_cus = prop.multiProcessorCount;
    const int rc = create_streams(h);
    if (rc != DN_OK) {                       // nothing leaks: dn_destroy releases whatever was created
        const std::string msg = g_err;
        (void) dn_destroy(h);
        return fail(rc, msg);
    }
    *out = h;
    return DN_OK;
}

int dn_destroy(dn_handle h)
{
    if (!h) return DN_OK;
    (void) hipSetDevice(h->device);
    if (h->stream) (void) hipStreamSynchronize(h->stream);
    (void) dn_comm_destroy(h);
    free_device(h);
    if (h->ev0) (void) hipEventDestroy(h->ev0);
    if (h->ev1) (void) hipEventDestroy(h->ev1);
    if (h->ev2a) (void) hipEventDestroy(h->ev2a);
    if (h->ev2b) (void) hipEventDestroy(h->ev2b);
    if (h->ev3a) (void) hipEventDestroy(h->ev3a);
    if (h->ev3b) (void) hipEventDestroy(h->ev3b);
    if (h->stream3) (void) hipStreamDestroy(h->stream3);
    if (h->ev_ready) (void) hipEventDestroy(h->ev_ready);
    if (h->ev_i0) (void) hipEventDestroy(h->ev_i0);
    if (h->ev_i1) (void) hipEventDestroy(h->ev_i1);
    if (h->stream2) (void) hipStreamDestroy(h->stream2);
    if (h->stream) (void) hipStreamDestroy(h->stream);
    delete h;
    return DN_OK;
}

// Scratch slots and LDS tier of one gene class for genes of up to `cols` active columns.  The new scratch is allocated
// FIRST and swapped in only on success: a failed (re)size leaves the class -- and the handle's alias of class 0's slots --
// exactly as it was, still valid.
static int size_class(dn_handle h, dn_handle_s::GeneClass &C, int32_t cols)
{
    const int32_t p = h->p;
    int per_cu = C.ks->blocks_per_cu(0);
    if (per_cu < 1) per_cu = 1;
    const int units = std::max(1, C.ks->units);          // genes a workgroup carries at once: a slot and an LDS tile per unit
    int slots = (int) std::min<int64_t>(((int64_t) C.n + units - 1) / units * units, (int64_t) per_cu * h->n_cus * units);
    const int32_t S = (cols + 63) & ~63;
    // slot: Fs, Fb (fp32, p x S) + x + lambda spill (fp64 [p][S]) + s_start, residual profile, A^T u (fp64 [S])
    const int64_t slot_bytes = (int64_t) S * ((int64_t) p * (2 * sizeof(float) + sizeof(double)) + 3 * sizeof(double)) + (int64_t) C.ks->slot_extra_bytes;
    {
        // one very long gene sizes every slot of its class: keep the scratch within a share of free HBM by
        // running fewer persistent workgroups rather than failing (the old scratch, still allocated, counts as free)
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const int64_t budget = (int64_t) ((free_b + (C.d_ws ? (size_t) C.slot_bytes * (size_t) std::max(C.slots, 1) : 0)) / 3);
        if (slot_bytes > budget) return fail(DN_E_INVALID, "a gene is too long for the device scratch (" + std::to_string(S) + " columns)");
        slots = (int) std::max<int64_t>(units, std::min<int64_t>(slots, budget / slot_bytes / units * units));
    }
    char *ws = nullptr;
    HIP_TRY(hipMalloc(&ws, (size_t) slot_bytes * (size_t) std::max(slots, 1)));
    const bool aliased = (h->d_ws == C.d_ws);
    if (C.d_ws) (void) hipFree(C.d_ws);
    C.d_ws = ws; C.slots = slots; C.S = S; C.slot_bytes = slot_bytes;
    if (aliased && &C == &h->cls[0]) h->d_ws = ws;
    // lambda LDS tier: whatever of the CU's 160 KiB is left per resident workgroup after the static part
    const int64_t lds_per_block = (160 * 1024) / per_cu - (int64_t) C.ks->static_lds_bytes - 256;
    const int64_t ps = p + (p & 1);                        // LDS column stride in doubles (16-B aligned)
    int64_t lcols = lds_per_block > 0 ? lds_per_block / units / (8 * ps) : 0;
    lcols = std::min<int64_t>(lcols, C.S) & ~(int64_t) 1;
    C.lds_cols = (int32_t) lcols;                          // per unit
    C.dyn_lds = (size_t) units * (size_t) lcols * 8 * (size_t) ps;
    return DN_OK;
}

// Where the coverage comes from: already packed host memory (one copy), or a routine that packs the genes [g0, g1) into a
// staging buffer (dn_upload_ragged: float64 dict values -> float32, threads), called chunk by chunk while the previous
// chunk's copy is in flight.
struct CoverageSource {
    const float *packed = nullptr;
    const void *const *genes = nullptr;
    int32_t is_f32 = 0, n_threads = 1;
    std::atomic<int64_t> *bad = nullptr;
};

static void pack_genes(dn_handle h, const CoverageSource &src, int64_t g0, int64_t g1, float *stage)
{
    const int32_t p = h->p;
    const int64_t base = h->goff[g0];
    std::atomic<int64_t> next(g0);
    auto work = [&]() {
        int64_t local_bad = 0;
        for (;;) {
            const int64_t a = next.fetch_add(16);
            if (a >= g1) break;
            const int64_t b = std::min(g1, a + 16);
            for (int64_t g = a; g < b; g++) {
                const int64_t cnt = (int64_t) p * h->glen[g];
                float *dst = stage + (h->goff[g] - base);
                if (src.is_f32) std::memcpy(dst, src.genes[g], sizeof(float) * (size_t) cnt);
                else {
                    const double *s = (const double *) src.genes[g];
                    for (int64_t k = 0; k < cnt; k++) { const float f = (float) s[k]; dst[k] = f; local_bad += ((double) f != s[k]); }
                }
            }
        }
        if (src.bad) *src.bad += local_bad;
    };
    const int nt = (int) std::max<int64_t>(1, std::min<int64_t>(src.n_threads, (g1 - g0 + 15) / 16));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

// Chunked, double-buffered upload of a ragged data set: two pinned staging buffers of ~128 MB; the host threads pack chunk
// k + 1 while chunk k travels (a single 2 GB pinned buffer cost more to allocate than the whole copy takes).
static int upload_ragged_chunks(dn_handle h, const CoverageSource &src)
{
    const int64_t n = h->n;
    int64_t chunk = (int64_t) 32 << 20;                                     // floats per staging buffer
    if (const char *env = getenv("DN_UPLOAD_CHUNK_FLOATS")) chunk = std::max<int64_t>(1, atoll(env));    // tests: many small chunks
    for (int64_t g = 0; g < n; g++) chunk = std::max(chunk, (int64_t) h->p * h->glen[g]);
    chunk = std::min(chunk, std::max<int64_t>(h->total, 1));
    float *stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    int rc = DN_OK;
    auto cleanup = [&]() {
        for (int b = 0; b < 2; b++) { if (stage[b]) (void) hipHostFree(stage[b]); if (done[b]) (void) hipEventDestroy(done[b]); }
    };
    for (int b = 0; b < 2 && rc == DN_OK; b++) {
        if (hipHostMalloc(&stage[b], sizeof(float) * (size_t) chunk, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&done[b], hipEventDisableTiming) != hipSuccess) rc = fail(DN_E_HIP, "upload: pinned staging buffer");
    }
    int64_t g0 = 0;
    for (int k = 0; rc == DN_OK && g0 < n; k++) {
        int64_t g1 = g0, fl = 0;
        while (g1 < n && fl + (int64_t) h->p * h->glen[g1] <= chunk) { fl += (int64_t) h->p * h->glen[g1]; g1++; }
        const int b = k & 1;
        if (k >= 2 && hipEventSynchronize(done[b]) != hipSuccess) { rc = fail(DN_E_HIP, "upload: event"); break; }
        pack_genes(h, src, g0, g1, stage[b]);
        if (hipMemcpyAsync(h->d_cov + h->goff[g0], stage[b], sizeof(float) * (size_t) fl, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipEventRecord(done[b], h->stream) != hipSuccess) { rc = fail(DN_E_HIP, "upload: copy"); break; }
        g0 = g1;
    }
    if (rc == DN_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(DN_E_HIP, "upload: synchronize");
    if (rc != DN_OK) (void) hipStreamSynchronize(h->stream);                // nothing may still read the staging buffers
    const std::string msg = g_err;
    cleanup();
    return rc == DN_OK ? DN_OK : fail(rc, msg);
}

// Class boundaries of a cohort of p samples served by kernel set `ks` (see dn_handle_s::GeneClass): genes longer than
// split_len run one per CU (wide class), the others two per CU (narrow), those of at most tiny_len bases one wavefront each
// (pair class; `pair` = its kernel set or null).  0 = the class does not exist.  Needs the device (occupancy queries).
static void class_lengths(const dn::KernelSet *ks, int32_t p, int32_t &split_len, int32_t &tiny_len, const dn::KernelSet *&pair)
{
    {
        const char *env = getenv("DN_SPLIT_LEN");
        const dn::KernelSet *narrow = (ks->p != 0) ? dn::kernel_set_narrow(p) : nullptr;
        split_len = 0;
        if (env) split_len = atoi(env);
        else if (narrow) {
            // Narrow class (128-thread workgroups, two genes per CU): its fixed cost per inner iteration (reduce +
            // eigen-solve) is paid by half a CU instead of a whole one, so it wins for every gene it can
            // keep mostly on chip.  On-chip columns of a narrow workgroup = register tier + LDS tier (p = 10: 1 280 + 975);
            // measured optimum of the boundary on config 2: 3 600-4 000 bases, i.e. ~1.7 x that capacity (split 2 600 /
            // 3 000 / 3 400 / 3 800 / 4 200 -> 11 510 / 11 740 / 11 920 / 11 950 / 11 890 genes/s).  Without a register
            // tier: ~2.1 x the LDS columns (round 1: 2 000-2 200 at 975 columns).
            const int per_cu_n = std::max(1, narrow->blocks_per_cu(0));
            const int64_t lds_n = (160 * 1024) / per_cu_n - (int64_t) narrow->static_lds_bytes - 256;
            const int64_t lds_cols_n = std::max<int64_t>(0, lds_n / (8 * (int64_t) (p + (p & 1))));
            const int64_t reg_cols_n = narrow->reg_tier_cols;
            // round 3: 1.775 x capacity (4 000 bases at p = 10) instead of 1.7 x (3 831): the same on the full configuration (-0.2 %), but
            // +2.4 % at the shard sizes of 2 and 8 GPUs (10 000 / 2 500 genes: 727 vs 745 ms, 191 vs 196 ms per run) -- with few genes
            // per GPU the wide class (one gene per CU, launched first) is what quantises: 500 instead of 595 genes on 256 slots
            split_len = (int32_t) (reg_cols_n > 0 ? (int64_t) (1.775 * (double) (reg_cols_n + lds_cols_n)) : (int64_t) (2.1 * (double) lds_cols_n));
        }
        if (!narrow) split_len = 0;
        if (!env && p >= 25) split_len = 0;     // wide cohorts (MFMA Gram): one class measured 3-5 % faster than two
        // Pair class: one wavefront per gene pays reduce + eigen-solve on ONE SIMD instead of two and needs no cross-wave step;
        // two such genes share a workgroup of the narrow class's shape.  A wavefront keeps its register tier plus its half of
        // the workgroup's LDS tile on chip (p = 10: 640 + ~480 columns); as for the narrow class the measured optimum of the
        // boundary is ~1.7 x that capacity (config 2, sweep in ms at 900 / 1 120 / 1 300 / 1 500 / 1 700 / 2 000 / 2 300 /
        // 2 600 / 3 000: 300.1 / 298.3 / 295.4 / 293.5 / 292.2 / 291.7 / 293.0 / 296.1 / 304.7; without the class 307.7).
        // DN_TINY_LEN overrides the boundary (0: no such class).
        pair = (ks->p != 0 && split_len > 0) ? dn::kernel_set_pair(p) : nullptr;
        tiny_len = 0;
        if (pair) {
            const char *tenv = getenv("DN_TINY_LEN");
            if (tenv) tiny_len = atoi(tenv);
            else {
                const int per_cu_t = std::max(1, pair->blocks_per_cu(0));
                const int64_t lds_t = ((160 * 1024) / per_cu_t - (int64_t) pair->static_lds_bytes - 256) / std::max(1, pair->units);
                const int64_t lds_cols_t = std::max<int64_t>(0, lds_t / (8 * (int64_t) (p + (p & 1))));
                tiny_len = (int32_t) (1.7 * (double) (pair->reg_tier_cols + lds_cols_t));
            }
            tiny_len = std::min(tiny_len, split_len);
            if (tiny_len <= 0) { pair = nullptr; tiny_len = 0; }
        }
    }
}

static int finish_upload_impl(dn_handle h, const CoverageSource &src)
{
    const int64_t n = h->n;
    const int32_t p = h->p;
    free_device(h);
    HIP_TRY(hipSetDevice(h->device));

    // work queue: longest gene first (a 17-call gene costs ~17x a 1-call gene)
    h->have_trace = false;
    std::vector<int32_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return h->glen[a] > h->glen[b]; });

    h->svoff.assign(n + 1, 0);
    for (int64_t g = 0; g < n; g++) h->svoff[g + 1] = h->svoff[g] + h->glen[g];

    // estimate tiles: (gene, first column) per 256 columns
    std::vector<int32_t> tg, tc;
    for (int64_t g = 0; g < n; g++)
        for (int32_t c = 0; c < h->glen[g]; c += 256) { tg.push_back((int32_t) g); tc.push_back(c); }
    h->n_tiles = (int64_t) tg.size();

    HIP_TRY(hipMalloc(&h->d_cov, sizeof(float) * (size_t) std::max<int64_t>(h->total, 1)));
    HIP_TRY(hipMalloc(&h->d_goff, sizeof(int64_t) * (size_t) (n + 1)));
    HIP_TRY(hipMalloc(&h->d_glen, sizeof(int32_t) * (size_t) n));
    HIP_TRY(hipMalloc(&h->d_order, sizeof(int32_t) * (size_t) n));
    HIP_TRY(hipMalloc(&h->d_counter, sizeof(int32_t) * 4));
    HIP_TRY(hipMalloc(&h->d_ds, sizeof(int64_t) * (size_t) n));
    HIP_TRY(hipMalloc(&h->d_rho, sizeof(double) * (size_t) n * p));
    HIP_TRY(hipMalloc(&h->d_flags, sizeof(int32_t) * (size_t) n));
    HIP_TRY(hipMalloc(&h->d_trace, sizeof(int32_t) * (size_t) n * dn::TRACE_LEN));
    HIP_TRY(hipMalloc(&h->d_kfin, sizeof(double) * (size_t) n * p));
    HIP_TRY(hipMalloc(&h->d_emode, sizeof(int32_t) * (size_t) n));
    HIP_TRY(hipMalloc(&h->d_svoff, sizeof(int64_t) * (size_t) (n + 1)));
    HIP_TRY(hipMalloc(&h->d_est_sums, sizeof(double) * (size_t) n * p));
    HIP_TRY(hipMalloc(&h->d_cov_sums, sizeof(double) * (size_t) n * p));
    HIP_TRY(hipMalloc(&h->d_status, sizeof(int32_t) * (size_t) n));
    HIP_TRY(hipMalloc(&h->d_rowmax, sizeof(float) * (size_t) n * p));
    HIP_TRY(hipMalloc(&h->d_x16, sizeof(int32_t) * (size_t) n));
    HIP_TRY(hipMalloc(&h->d_tile_gene, sizeof(int32_t) * (size_t) std::max<int64_t>(h->n_tiles, 1)));
    HIP_TRY(hipMalloc(&h->d_tile_col, sizeof(int32_t) * (size_t) std::max<int64_t>(h->n_tiles, 1)));

    if (src.packed) HIP_TRY(hipMemcpyAsync(h->d_cov, src.packed, sizeof(float) * (size_t) h->total, hipMemcpyHostToDevice, h->stream));
    else { const int rcu = upload_ragged_chunks(h, src); if (rcu != DN_OK) return rcu; }
    HIP_TRY(hipMemcpyAsync(h->d_goff, h->goff.data(), sizeof(int64_t) * (size_t) (n + 1), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_glen, h->glen.data(), sizeof(int32_t) * (size_t) n, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_order, order.data(), sizeof(int32_t) * (size_t) n, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_svoff, h->svoff.data(), sizeof(int64_t) * (size_t) (n + 1), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_tile_gene, tg.data(), sizeof(int32_t) * tg.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_tile_col, tc.data(), sizeof(int32_t) * tc.size(), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipEventRecord(h->ev_i0, h->stream));
    hipLaunchKernelGGL(k_row_max, dim3((unsigned) std::min<int64_t>(n, (int64_t) h->n_cus * 8)), dim3(256), 0, h->stream,
                       h->d_cov, h->d_goff, h->d_glen, h->d_rowmax, h->d_x16, (int) n, (int) p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev_i1, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    (void) hipEventElapsedTime(&h->last_rowmax_ms, h->ev_i0, h->ev_i1);

    // gene classes
    {
        const dn::KernelSet *pair = nullptr;
        class_lengths(h->ks, p, h->split_len, h->tiny_len, pair);
        std::vector<int32_t> ord[dn_handle_s::NCLS];
        for (int32_t g : order) {                                                                           // stays longest-first
            const int32_t L = h->glen[g];
            ord[(h->split_len > 0 && L <= h->split_len) ? ((pair && L <= h->tiny_len) ? 2 : 1) : 0].push_back(g);
        }
        h->cls[0].ks = h->ks;
        h->cls[1].ks = (h->ks->p != 0 && h->split_len > 0) ? dn::kernel_set_narrow(p) : nullptr;
        h->cls[2].ks = pair;
        {
            // Queue order of a class: longest first, then zigzagged (longest, shortest, 2nd longest, 2nd shortest, ...).
            // Long genes keep most of their state in the spill tier and pull ~60 GB/s per CU through the fabric, short
            // ones a quarter of that; running only long genes everywhere at the start of a launch saturates the fabric
            // (their pass is 1.4x slower than alone on the chip, tools/contention.sh).  The zigzag keeps the demand level
            // over the launch: +2.9 % on config 2 and more on short queues (+6 % at 5 000 genes per GPU against ordering
            // by the cost predicted from the previous iteration's counters, which clusters the heavy genes even more).
            // DN_ORDER_MIX=0 restores plain longest-first, bit c selects class c.
            const char *mix = getenv("DN_ORDER_MIX");
            const int mask = mix ? atoi(mix) : 1;
            for (int c = 0; c < dn_handle_s::NCLS; c++) {
                if (!((mask >> c) & 1) || ord[c].size() < 8) continue;
                const size_t n0 = ord[c].size();
                std::vector<int32_t> mixed;
                mixed.reserve(n0);
                for (size_t i = 0, j = n0 - 1; i < j; i++, j--) { mixed.push_back(ord[c][i]); mixed.push_back(ord[c][j]); }
                if (n0 & 1) mixed.push_back(ord[c][n0 / 2]);
                ord[c] = mixed;
            }
        }
        for (int c = 0; c < dn_handle_s::NCLS; c++) {
            auto &C = h->cls[c];
            C.n = (int32_t) ord[c].size();
            if (C.n == 0 || !C.ks) { C.n = 0; continue; }
            HIP_TRY(hipMalloc(&C.d_order, sizeof(int32_t) * (size_t) C.n));
            HIP_TRY(hipMalloc(&C.d_counter, sizeof(int32_t) * 4));
            HIP_TRY(hipMemcpy(C.d_order, ord[c].data(), sizeof(int32_t) * (size_t) C.n, hipMemcpyHostToDevice));
            C.order = ord[c];
            // columns a scratch slot must hold: the longest gene of the class, or -- for the one-wave-per-gene family,
            // chosen because the announced take-every rate leaves every gene at most 12 active columns -- that bound
            int32_t cols = C.longest = h->glen[ord[c][0]];
            if (C.ks == dn::kernel_set_rows()) cols = (cols + h->ds_hint - 1) / h->ds_hint;
            const int rc = size_class(h, C, cols);
            if (rc != DN_OK) return rc;
        }
        if (h->cls[0].n == 0 && (h->cls[1].n > 0 || h->cls[2].n > 0) && h->ks->p == 0) return fail(DN_E_STATE, "internal: empty wide class for the generic kernels");
        h->slots = h->cls[0].slots; h->S = h->cls[0].S; h->slot_bytes = h->cls[0].slot_bytes; h->d_ws = h->cls[0].d_ws;
    }
    return DN_OK;
}

// A failed upload leaves the handle EMPTY (no resident coverage, later calls return DN_E_STATE), never half-sized.
static int finish_upload(dn_handle h, const CoverageSource &src)
{
    const int rc = finish_upload_impl(h, src);
    if (rc != DN_OK) {
        const std::string msg = g_err;
        free_device(h);
        h->n = 0; h->total = 0;
        return fail(rc, msg);
    }
    return DN_OK;
}

static int check_shape(dn_handle h, int64_t n_genes, int32_t p, const int64_t *lengths)
{
    if (!h) return fail(DN_E_INVALID, "null handle");
    if (n_genes <= 0 || n_genes > INT32_MAX) return fail(DN_E_INVALID, "n_genes must be in [1, 2^31)");
    if (p < 2) return fail(DN_E_INVALID, "need at least 2 samples (svds(k=1) requires 1 < min(shape), nmf.py:63)");
    const dn::KernelSet *ks = dn::kernel_set_for(p);
    if (!ks) return fail(DN_E_UNSUPPORTED, "no kernels compiled for p = " + std::to_string(p));
    // validate into locals; the handle is only touched once everything checks out
    std::vector<int64_t> goff((size_t) n_genes + 1, 0);
    std::vector<int32_t> glen((size_t) n_genes, 0);
    int32_t lmax = 0;
    for (int64_t g = 0; g < n_genes; g++) {
        if (lengths[g] < 1 || lengths[g] > (int64_t) 1 << 26) return fail(DN_E_INVALID, "gene length out of range at gene " + std::to_string(g));
        glen[g] = (int32_t) lengths[g];
        goff[g + 1] = goff[g] + (int64_t) p * lengths[g];
        lmax = std::max(lmax, glen[g]);
    }
    // Down-sampled regime announced by the caller: when no gene can keep more than 12 active columns the run-time-p
    // kernels serve it row-wise, one wavefront per gene (dn_generic.hip, nmf_rows) -- faster than the column-parallel
    // templated kernels from p ~ 8 on (p = 16: 76 000 vs 41 000 genes/s per run, p = 32: 5x).
    const bool rows_regime = h->ds_hint > 1 && p >= 8 && (lmax + h->ds_hint - 1) / h->ds_hint <= 12;
    const char *force = getenv("DN_FORCE_GENERIC");
    if (rows_regime && !(force && force[0] == '2')) ks = dn::kernel_set_rows();      // DN_FORCE_GENERIC=2: keep the 256-thread family
    else if (rows_regime) ks = dn::kernel_set_generic();
    // commit: the previous data set (if any) is released first, so that a later failure cannot leave old device
    // buffers behind new host-side shapes
    free_device(h);
    h->ks = ks;
    h->n = n_genes; h->p = p;
    h->goff.swap(goff); h->glen.swap(glen);
    h->total = h->goff[n_genes];
    h->lmax = lmax;
    return DN_OK;
}

int dn_set_downsample_hint(dn_handle h, int32_t rate)
{
    if (!h) return fail(DN_E_INVALID, "null handle");
    if (rate < 1) return fail(DN_E_INVALID, "downsample rate must be >= 1");
    h->ds_hint = rate;
    return DN_OK;
}

int dn_set_trace_columns(dn_handle h, int32_t cols)
{
    if (!h) return fail(DN_E_INVALID, "null handle");
    if (cols < 8 || cols > dn::TRACE_LEN) return fail(DN_E_INVALID, "trace columns must be in [8, DN_TRACE_LEN]");
    h->trace_cols = cols;
    h->have_trace = false;                                // the host copy changes shape: no stale counters for the queue order
    return DN_OK;
}

int dn_set_solver_step_cap(dn_handle h, int32_t max_steps)
{
    if (!h) return fail(DN_E_INVALID, "null handle");
    if (max_steps < 1) return fail(DN_E_INVALID, "the solver step cap must be >= 1");
    h->max_steps = max_steps;
    return DN_OK;
}

int dn_upload_packed(dn_handle h, int64_t n_genes, int32_t p, const float *packed, const int64_t *lengths)
{
    if (!packed || !lengths) return fail(DN_E_INVALID, "dn_upload_packed: null argument");
    int rc = check_shape(h, n_genes, p, lengths);
    if (rc != DN_OK) return rc;
    CoverageSource src;
    src.packed = packed;
    return finish_upload(h, src);
}

int dn_upload_ragged(dn_handle h, int64_t n_genes, int32_t p, const void *const *genes, const int64_t *lengths,
                     int32_t is_f32, int32_t n_threads, int64_t *inexact)
{
    if (!genes || !lengths) return fail(DN_E_INVALID, "dn_upload_ragged: null argument");
    int rc = check_shape(h, n_genes, p, lengths);
    if (rc != DN_OK) return rc;
    HIP_TRY(hipSetDevice(h->device));
    if (n_threads < 1) n_threads = (int32_t) std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::atomic<int64_t> bad(0);
    CoverageSource src;
    src.genes = genes; src.is_f32 = is_f32; src.n_threads = n_threads; src.bad = &bad;
    rc = finish_upload(h, src);
    if (inexact) *inexact = bad.load();
    return rc;
}

int dn_ratio_svd_sums(dn_handle h, double *est_sums, double *cov_sums, int32_t *status)
{
    if (!h || !h->d_cov) return fail(DN_E_STATE, "dn_ratio_svd_sums: nothing uploaded");
    if ((est_sums == nullptr) != (cov_sums == nullptr)) return fail(DN_E_INVALID, "dn_ratio_svd_sums: the two sums are fetched together or not at all");
    HIP_TRY(hipSetDevice(h->device));
    dn::InitArgs a;
    a.cov = h->d_cov; a.goff = h->d_goff; a.glen = h->d_glen; a.order = h->d_order; a.counter = h->d_counter;
    a.est_sums = h->d_est_sums; a.cov_sums = h->d_cov_sums; a.status = h->d_status; a.n_genes = (int32_t) h->n;
    a.p = h->p; a.ws = h->d_ws; a.slot_bytes = h->slot_bytes; a.S = h->S; a.max_steps = h->max_steps;
    a.x16 = h->d_x16;
    { const char *f64 = getenv("DN_INIT_FP64"); a.force_fp64 = (f64 && f64[0] == '1') ? 1 : 0; }
    HIP_TRY(hipMemsetAsync(h->d_counter, 0, sizeof(int32_t) * 4, h->stream));
    // occupancy of the kernel that ks->init() will start: from 17 samples on it is the matrix-core variant (round 2 asked for
    // the power-iteration kernel's figure here and ran k_ratio_svd_mg at ONE workgroup per CU instead of two)
    const char *pw = getenv("DN_INIT_POWER");
    const int which_init = (h->p >= 17 && !(pw && pw[0] == '1')) ? 2 : 1;
    int per_cu = std::max(1, h->ks->blocks_per_cu(which_init));
    int grid = (int) std::min<int64_t>(h->n, (int64_t) per_cu * h->n_cus);
    if (h->ks->p == 0) grid = std::min(grid, h->slots);          // generic kernels work in the scratch slots
    HIP_TRY(hipEventRecord(h->ev_i0, h->stream));
    h->ks->init(a, grid, h->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev_i1, h->stream));
    const size_t np = (size_t) h->n * h->p;
    if (est_sums) {                     // null: the sums stay on the device (dn_init_partials reduces them there)
        HIP_TRY(hipMemcpyAsync(est_sums, h->d_est_sums, sizeof(double) * np, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(cov_sums, h->d_cov_sums, sizeof(double) * np, hipMemcpyDeviceToHost, h->stream));
    }
    if (status) HIP_TRY(hipMemcpyAsync(status, h->d_status, sizeof(int32_t) * (size_t) h->n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipEventElapsedTime(&h->last_init_ms, h->ev_i0, h->ev_i1));
    if (h->ks->p >= 2 && h->ks->p <= 16) snprintf(h->init_name, sizeof(h->init_name), "k_ratio_svd<%d,%d>", h->ks->p, h->ks->nt);
    else snprintf(h->init_name, sizeof(h->init_name), (h->p >= 17 && !(getenv("DN_INIT_POWER") && getenv("DN_INIT_POWER")[0] == '1')) ? "gen::k_ratio_svd_mg" : "gen::k_ratio_svd_gen");
    return DN_OK;
}

int dn_baseline_iteration(dn_handle h, const double *scale, const dn_params *prm, const int64_t *ds_start,
                          double *rho, int32_t *flags, int32_t *trace)
{
    if (!h || !h->d_cov) return fail(DN_E_STATE, "dn_baseline_iteration: nothing uploaded");
    if (!scale || !prm) return fail(DN_E_INVALID, "dn_baseline_iteration: null argument");
    if ((rho == nullptr) != (flags == nullptr)) return fail(DN_E_INVALID, "dn_baseline_iteration: rho and flags are fetched together or not at all");
    if (prm->nmf_iter < 1) return fail(DN_E_INVALID, "nmf_iter must be >= 1");
    if (prm->bins < 1 || prm->bins > dn::MAX_BINS) return fail(DN_E_INVALID, "bins must be in [1, 64]");
    if (prm->min_high_coverage < 2) return fail(DN_E_INVALID, "min_high_coverage must be >= 2 (nmf.py:34)");
    if (prm->downsample_rate < 1) return fail(DN_E_INVALID, "downsample_rate must be >= 1");
    if (prm->downsample_rate > 1) {
        if (!ds_start) return fail(DN_E_INVALID, "downsample_rate > 1 needs per-gene start offsets");
        for (int64_t g = 0; g < h->n; g++) {
            // nmf.py:443-444 / :479-481: cannot downsample at a rate >= gene length
            if (h->glen[g] <= prm->downsample_rate) return fail(DN_E_INVALID, "downsample_rate is too large; take-every size > at least one gene.");
            if (ds_start[g] < 0 || ds_start[g] >= prm->downsample_rate) return fail(DN_E_INVALID, "ds_start out of [0, rate)");
        }
    }
    for (int i = 0; i < h->p; i++) if (!(scale[i] > 0.0) || !std::isfinite(scale[i])) return fail(DN_E_INVALID, "scale factors must be positive and finite");
    HIP_TRY(hipSetDevice(h->device));

    {
        // the scratch slots were sized at upload (for the announced take-every rate in the one-wave-per-gene family):
        // grow them if this iteration's rate leaves more active columns than they hold
        const int32_t rate = prm->downsample_rate;
        for (auto &C : h->cls) {
            if (C.n == 0 || !C.ks) continue;
            const int32_t longest = C.longest;
            const int32_t need = rate > 1 ? (longest + rate - 1) / rate : longest;
            if (need > C.S) {
                HIP_TRY(hipStreamSynchronize(h->stream));
                const int rc = size_class(h, C, need);
                if (rc != DN_OK) return rc;
                if (&C == &h->cls[0]) { h->slots = C.slots; h->S = C.S; h->slot_bytes = C.slot_bytes; h->d_ws = C.d_ws; }
            }
        }
    }
    if (prm->want_estimates && !h->d_svec)
        HIP_TRY(hipMalloc(&h->d_svec, sizeof(double) * (size_t) std::max<int64_t>(h->svoff[h->n], 1)));

    dn::IterArgs a;
    std::memset(&a, 0, sizeof(a));
    a.cov = h->d_cov; a.goff = h->d_goff; a.glen = h->d_glen; a.order = h->d_order; a.counter = h->d_counter;
    a.ds_start = nullptr;
    a.ws = h->d_ws; a.rho = h->d_rho; a.flags = h->d_flags; a.trace = h->d_trace; a.kfin = h->d_kfin; a.emode = h->d_emode;
    a.svec = h->d_svec; a.svoff = h->d_svoff; a.slot_bytes = h->slot_bytes; a.n_genes = (int32_t) h->n; a.S = h->S;
    a.p = h->p; a.rowmax = h->d_rowmax; a.x16 = h->d_x16; a.max_steps = h->max_steps;
    a.T = prm->nmf_iter; a.bins = prm->bins; a.min_hc = prm->min_high_coverage; a.rate = prm->downsample_rate;
    a.skip = prm->skip_baseline_selection ? 1 : 0; a.want_est = prm->want_estimates ? 1 : 0;
    for (int i = 0; i < h->p; i++) { a.scale[i] = scale[i]; a.inv_scale[i] = 1.0 / scale[i]; h->last_scale[i] = scale[i]; }
    for (int i = h->p; i < dn::P_MAX; i++) { a.scale[i] = 1.0; a.inv_scale[i] = 1.0; }
    if (prm->downsample_rate > 1) {
        HIP_TRY(hipMemcpyAsync(h->d_ds, ds_start, sizeof(int64_t) * (size_t) h->n, hipMemcpyHostToDevice, h->stream));
        a.ds_start = h->d_ds;
    }
    // The narrow class (state in LDS, little fabric traffic) orders its queue most expensive first from the second
    // iteration on, the cost of a gene predicted from the previous iteration's counters (sum of active columns over its
    // nmf() calls plus a fixed part per call worth ~4 columns per lane): its genes are what fills the end of a launch.
    // The wide class keeps the zigzag (it is bound by the fabric, see upload).  DN_NARROW_WORK_ORDER=0 disables.
    {
        const char *wo = getenv("DN_NARROW_WORK_ORDER");
        for (int wc = 1; wc < dn_handle_s::NCLS; wc++) {
        auto &C = h->cls[wc];
        if (h->have_trace && C.n > 0 && C.ks && !(wo && wo[0] == '0')) {
            const double per_call = 4.0 * (double) (C.ks->nt > 0 ? C.ks->nt : 128);
            std::vector<std::pair<double, int32_t>> key((size_t) C.n);
            for (int32_t k = 0; k < C.n; k++) {
                const int32_t g = C.order[k];
                const int32_t *tr = &h->host_trace[(size_t) g * h->trace_cols];
                key[k] = {(double) tr[2] + per_call * (double) tr[1] + 1e-3 * (double) h->glen[g], g};
            }
            std::stable_sort(key.begin(), key.end(), [](const std::pair<double, int32_t> &a, const std::pair<double, int32_t> &b) { return a.first > b.first; });
            for (int32_t k = 0; k < C.n; k++) C.order[k] = key[k].second;
            HIP_TRY(hipMemcpyAsync(C.d_order, C.order.data(), sizeof(int32_t) * (size_t) C.n, hipMemcpyHostToDevice, h->stream));
        }
        }
    }
    HIP_TRY(hipMemsetAsync(h->d_trace, 0, sizeof(int32_t) * (size_t) h->n * dn::TRACE_LEN, h->stream));
    for (auto &C : h->cls) if (C.n > 0) HIP_TRY(hipMemsetAsync(C.d_counter, 0, sizeof(int32_t) * 4, h->stream));
    HIP_TRY(hipEventRecord(h->ev_ready, h->stream));
    int first_cls = -1;
    for (int c = 0; c < dn_handle_s::NCLS; c++) {
        auto &C = h->cls[c];
        C.last_ms = 0.f;
        if (C.n == 0) continue;
        if (first_cls < 0) first_cls = c;
        hipStream_t st = h->class_stream(c);
        if (c > 0) HIP_TRY(hipStreamWaitEvent(st, h->ev_ready, 0));
        a.order = C.d_order; a.counter = C.d_counter; a.ws = C.d_ws; a.slot_bytes = C.slot_bytes; a.S = C.S;
        a.lds_cols = C.lds_cols; a.n_genes = C.n;
        HIP_TRY(hipEventRecord(h->class_ev_a(c), st));
        const int lrc = C.ks->baseline(a, C.slots, C.dyn_lds, st);
        if (lrc != 0) return fail(DN_E_HIP, std::string("k_baseline launch: ") + hipGetErrorString((hipError_t) lrc));
        HIP_TRY(hipEventRecord(h->class_ev_b(c), st));
    }
    for (int c = 1; c < dn_handle_s::NCLS; c++)                                   // results are copied on the main stream
        if (h->cls[c].n > 0) HIP_TRY(hipStreamWaitEvent(h->stream, h->class_ev_b(c), 0));
    if (rho) {                          // null: the DI rows stay on the device (dn_outer_partials / dn_outer_apply / dn_fetch_outer)
        HIP_TRY(hipMemcpyAsync(rho, h->d_rho, sizeof(double) * (size_t) h->n * h->p, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipMemcpyAsync(flags, h->d_flags, sizeof(int32_t) * (size_t) h->n, hipMemcpyDeviceToHost, h->stream));
    }
    const int32_t tcols = h->trace_cols;
    const size_t trace_ints = (size_t) h->n * (size_t) tcols;
    if (h->host_trace_len < trace_ints) {
        if (h->host_trace) { (void) hipHostFree(h->host_trace); h->host_trace = nullptr; h->host_trace_len = 0; }
        HIP_TRY(hipHostMalloc((void **) &h->host_trace, sizeof(int32_t) * trace_ints, hipHostMallocDefault));
        h->host_trace_len = trace_ints;
    }
    if (tcols == dn::TRACE_LEN)
        HIP_TRY(hipMemcpyAsync(h->host_trace, h->d_trace, sizeof(int32_t) * trace_ints, hipMemcpyDeviceToHost, h->stream));
    else {
        if (!h->d_trace_head) HIP_TRY(hipMalloc(&h->d_trace_head, sizeof(int32_t) * (size_t) h->n * dn::TRACE_LEN));
        hipLaunchKernelGGL(k_trace_head, dim3((unsigned) std::min<size_t>(1024, (trace_ints + 255) / 256)), dim3(256), 0, h->stream,
                           h->d_trace, h->d_trace_head, (long long) trace_ints, (int) tcols);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h->host_trace, h->d_trace_head, sizeof(int32_t) * trace_ints, hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->have_trace = (prm->downsample_rate <= 1);          // with down-sampling the active columns are redrawn every iteration
    if (trace) std::memcpy(trace, h->host_trace, sizeof(int32_t) * trace_ints);
    h->last_span_ms = 0.f;
    for (int c = 0; c < dn_handle_s::NCLS; c++) {
        if (h->cls[c].n == 0) continue;
        HIP_TRY(hipEventElapsedTime(&h->cls[c].last_ms, h->class_ev_a(c), h->class_ev_b(c)));
        float span = 0.f;                                                // the classes are launched in order: the first start opens the span
        HIP_TRY(hipEventElapsedTime(&span, h->class_ev_a(first_cls), h->class_ev_b(c)));
        h->last_span_ms = std::max(h->last_span_ms, span);
    }
    h->last_ms = first_cls >= 0 ? h->cls[first_cls].last_ms : 0.f;
    h->have_estimate_state = prm->want_estimates != 0;
    return DN_OK;
}

static int outer_alloc(dn_handle h, int32_t degnorm_iter)
{
    const size_t np = (size_t) h->n * h->p;
    if (!h->d_rhoc) {
        HIP_TRY(hipMalloc(&h->d_rhoc, sizeof(double) * np));
        HIP_TRY(hipMalloc(&h->d_xw, sizeof(double) * np));
        HIP_TRY(hipMalloc(&h->d_xadj, sizeof(double) * np));
    }
    if (!h->d_part) {
        HIP_TRY(hipMalloc(&h->d_part, sizeof(double) * (size_t) OUT_BLOCKS * OUT_STRIDE));
        HIP_TRY(hipMalloc(&h->d_pvec, sizeof(double) * (3 * dn::P_MAX + 4 + 2 * dn::P_MAX)));
    }
    if (degnorm_iter > 0 && h->n_iter != degnorm_iter) {
        if (h->d_ran) { (void) hipFree(h->d_ran); h->d_ran = nullptr; }
        HIP_TRY(hipMalloc(&h->d_ran, (size_t) h->n * degnorm_iter));
        h->n_iter = degnorm_iter;
    }
    return DN_OK;
}

int dn_init_begin(dn_handle h, const double *reads)
{
    if (!h || !h->d_cov) return fail(DN_E_STATE, "dn_init_begin: nothing uploaded");
    if (!reads) return fail(DN_E_INVALID, "dn_init_begin: null argument");
    HIP_TRY(hipSetDevice(h->device));
    const size_t np = (size_t) h->n * h->p;
    if (!h->d_x) HIP_TRY(hipMalloc(&h->d_x, sizeof(double) * np));
    HIP_TRY(hipMemcpyAsync(h->d_x, reads, sizeof(double) * np, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return DN_OK;
}

int dn_init_partials(dn_handle h, double *partials)
{
    if (!h || !h->d_x) return fail(DN_E_STATE, "dn_init_partials: dn_init_begin has not been called");
    if (!partials) return fail(DN_E_INVALID, "dn_init_partials: null output");
    HIP_TRY(hipSetDevice(h->device));
    { const int rc = outer_alloc(h, 0); if (rc != DN_OK) return rc; }
    const int blocks = (int) std::min<int64_t>(OUT_BLOCKS, (h->n + 3) / 4);
    hipLaunchKernelGGL(k_init_partials, dim3(blocks), dim3(256), 0, h->stream, h->d_est_sums, h->d_cov_sums, h->d_status, h->d_x, h->d_part,
                       (int) h->n, (int) h->p);
    hipLaunchKernelGGL(k_outer_reduce, dim3(1), dim3(256), 0, h->stream, h->d_part, h->d_pvec, blocks, (int) h->p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(partials, h->d_pvec, sizeof(double) * (size_t) (3 * h->p + 4), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return DN_OK;
}

int dn_outer_begin_scaled(dn_handle h, const double *norm, int32_t degnorm_iter)
{
    if (!h || !h->d_x) return fail(DN_E_STATE, "dn_outer_begin_scaled: dn_init_begin has not been called");
    if (!norm || degnorm_iter < 1) return fail(DN_E_INVALID, "dn_outer_begin_scaled: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    { const int rc = outer_alloc(h, degnorm_iter); if (rc != DN_OK) return rc; }
    const size_t np = (size_t) h->n * h->p;
    double *d_norm = h->d_pvec + (3 * dn::P_MAX + 4) + dn::P_MAX;
    HIP_TRY(hipMemsetAsync(h->d_ran, 0, (size_t) h->n * degnorm_iter, h->stream));
    HIP_TRY(hipMemcpyAsync(d_norm, norm, sizeof(double) * (size_t) h->p, hipMemcpyHostToDevice, h->stream));
    const int blocks = (int) std::min<size_t>(2048, (np + 255) / 256);
    hipLaunchKernelGGL(k_scale_reads, dim3(blocks), dim3(256), 0, h->stream, h->d_x, d_norm, h->d_xw, (long long) np, (int) h->p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    return DN_OK;
}

int dn_outer_begin(dn_handle h, const double *x_weighted, int32_t degnorm_iter)
{
    if (!h || !h->d_cov) return fail(DN_E_STATE, "dn_outer_begin: nothing uploaded");
    if (!x_weighted || degnorm_iter < 1) return fail(DN_E_INVALID, "dn_outer_begin: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    const size_t np = (size_t) h->n * h->p;
    { const int rc = outer_alloc(h, degnorm_iter); if (rc != DN_OK) return rc; }
    HIP_TRY(hipMemsetAsync(h->d_ran, 0, (size_t) h->n * degnorm_iter, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_xw, x_weighted, sizeof(double) * np, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return DN_OK;
}

int dn_outer_partials(dn_handle h, double *partials)
{
    if (!h || !h->d_xw) return fail(DN_E_STATE, "dn_outer_partials: dn_outer_begin has not been called");
    if (!partials) return fail(DN_E_INVALID, "dn_outer_partials: null output");
    HIP_TRY(hipSetDevice(h->device));
    const int blocks = (int) std::min<int64_t>(OUT_BLOCKS, (h->n + 3) / 4);
    hipLaunchKernelGGL(k_outer_partials, dim3(blocks), dim3(256), 0, h->stream, h->d_rho, h->d_rhoc, h->d_xw, h->d_trace, h->d_flags, h->d_part,
                       (int) h->n, (int) h->p);
    hipLaunchKernelGGL(k_outer_reduce, dim3(1), dim3(256), 0, h->stream, h->d_part, h->d_pvec, blocks, (int) h->p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(partials, h->d_pvec, sizeof(double) * (size_t) (3 * h->p + 4), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return DN_OK;
}

int dn_outer_partials_device(dn_handle h, double **d_partials)
{
    if (!h || !h->d_xw) return fail(DN_E_STATE, "dn_outer_partials_device: dn_outer_begin has not been called");
    if (!d_partials) return fail(DN_E_INVALID, "dn_outer_partials_device: null output");
    HIP_TRY(hipSetDevice(h->device));
    const int blocks = (int) std::min<int64_t>(OUT_BLOCKS, (h->n + 3) / 4);
    hipLaunchKernelGGL(k_outer_partials, dim3(blocks), dim3(256), 0, h->stream, h->d_rho, h->d_rhoc, h->d_xw, h->d_trace, h->d_flags, h->d_part,
                       (int) h->n, (int) h->p);
    hipLaunchKernelGGL(k_outer_reduce, dim3(1), dim3(256), 0, h->stream, h->d_part, h->d_pvec, blocks, (int) h->p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));           // the collective runs on the caller's stream: the sums must be there
    *d_partials = h->d_pvec;
    return DN_OK;
}

int dn_outer_apply(dn_handle h, const double *avg_di, const double *norm, int32_t iter)
{
    if (!h || !h->d_xw) return fail(DN_E_STATE, "dn_outer_apply: dn_outer_begin has not been called");
    if (!norm || iter < 0) return fail(DN_E_INVALID, "dn_outer_apply: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    double *d_avg = h->d_pvec + (3 * dn::P_MAX + 4), *d_norm = d_avg + dn::P_MAX;
    if (avg_di) HIP_TRY(hipMemcpyAsync(d_avg, avg_di, sizeof(double) * (size_t) h->p, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(d_norm, norm, sizeof(double) * (size_t) h->p, hipMemcpyHostToDevice, h->stream));
    const int blocks = (int) std::min<int64_t>(OUT_BLOCKS, (h->n + 3) / 4);
    hipLaunchKernelGGL(k_outer_apply, dim3(blocks), dim3(256), 0, h->stream, h->d_rhoc, h->d_xw, h->d_xadj, h->d_flags, h->d_ran, d_avg, d_norm,
                       avg_di ? 1 : 0, (int) h->n, (int) h->p, (int) iter, (int) h->n_iter);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));           // the host buffers behind avg_di / norm may go away
    return DN_OK;
}

// ---------------------------------------------------------------------------------------------------
// The collective of the sharded run inside the library (nmf_mpi.py:758-815 moves whole coverage chunks and DI matrices
// through rank 0; here 3p + 4 doubles are summed over the GPUs in place, on the handle's own stream, by RCCL over xGMI).
// ---------------------------------------------------------------------------------------------------
#define RCCL_TRY(api, expr)                                                                        \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess)                                                                     \
            return fail(DN_E_HIP, std::string(#expr) + ": " + (api)->GetErrorString(r_));          \
    } while (0)

int dn_comm_unique_id(uint8_t *id)
{
    if (!id) return fail(DN_E_INVALID, "dn_comm_unique_id: null argument");
    std::string err;
    RcclApi *api = rccl_api(err);
    if (!api) return fail(DN_E_STATE, "dn_comm_unique_id: " + err);
    static_assert(sizeof(ncclUniqueId) == DN_COMM_ID_BYTES, "DN_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
    ncclUniqueId u;
    RCCL_TRY(api, api->GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return DN_OK;
}

int dn_comm_create(dn_handle h, const uint8_t *id, int32_t rank, int32_t size)
{
    if (!h) return fail(DN_E_INVALID, "dn_comm_create: null handle");
    if (!id || size < 1 || rank < 0 || rank >= size) return fail(DN_E_INVALID, "dn_comm_create: bad argument");
    if (h->comm) return fail(DN_E_STATE, "dn_comm_create: this handle already has a communicator");
    std::string err;
    RcclApi *api = rccl_api(err);
    if (!api) return fail(DN_E_STATE, "dn_comm_create: " + err);
    HIP_TRY(hipSetDevice(h->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    RCCL_TRY(api, api->CommInitRank(&c, size, u, rank));
    h->comm = c; h->comm_rank = rank; h->comm_size = size;
    if (!h->d_comm) HIP_TRY(hipMalloc(&h->d_comm, sizeof(double) * 256));
    return DN_OK;
}

int dn_comm_destroy(dn_handle h)
{
    if (!h || !h->comm) return DN_OK;
    std::string err;
    RcclApi *api = rccl_api(err);
    (void) hipSetDevice(h->device);
    if (h->stream) (void) hipStreamSynchronize(h->stream);
    if (api) (void) api->CommDestroy((ncclComm_t) h->comm);
    h->comm = nullptr; h->comm_size = 0; h->comm_rank = 0;
    if (h->d_comm) { (void) hipFree(h->d_comm); h->d_comm = nullptr; }
    return DN_OK;
}

int32_t dn_comm_size(dn_handle h) { return h ? h->comm_size : 0; }

const char *dn_comm_library(void)
{
    std::string err;
    RcclApi *api = rccl_api(err);
    static std::string text;
    if (!api) { text = ""; return text.c_str(); }
    int v = 0;
    if (api->GetVersion) (void) api->GetVersion(&v);
    text = api->path + ", nccl api " + std::to_string(v);
    return text.c_str();
}

// sum d_buf[0 .. count) over the ranks in place on the handle's stream, then copy the totals to the host
static int allreduce_to_host(dn_handle h, double *d_buf, int32_t count, double *totals)
{
    std::string err;
    RcclApi *api = rccl_api(err);
    if (!api) return fail(DN_E_STATE, err);
    RCCL_TRY(api, api->AllReduce(d_buf, d_buf, (size_t) count, ncclDouble, ncclSum, (ncclComm_t) h->comm, h->stream));
    HIP_TRY(hipMemcpyAsync(totals, d_buf, sizeof(double) * (size_t) count, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return DN_OK;
}

int dn_comm_allreduce(dn_handle h, double *buf, int32_t count)
{
    if (!h || !h->comm) return fail(DN_E_STATE, "dn_comm_allreduce: dn_comm_create has not been called");
    if (!buf || count < 1 || count > 256) return fail(DN_E_INVALID, "dn_comm_allreduce: 1 .. 256 doubles");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->d_comm, buf, sizeof(double) * (size_t) count, hipMemcpyHostToDevice, h->stream));
    return allreduce_to_host(h, h->d_comm, count, buf);
}

int dn_init_allreduce(dn_handle h, double *totals)
{
    if (!h || !h->comm) return fail(DN_E_STATE, "dn_init_allreduce: dn_comm_create has not been called");
    if (!h->d_x) return fail(DN_E_STATE, "dn_init_allreduce: dn_init_begin has not been called");
    if (!totals) return fail(DN_E_INVALID, "dn_init_allreduce: null output");
    HIP_TRY(hipSetDevice(h->device));
    { const int rc = outer_alloc(h, 0); if (rc != DN_OK) return rc; }
    const int blocks = (int) std::min<int64_t>(OUT_BLOCKS, (h->n + 3) / 4);
    hipLaunchKernelGGL(k_init_partials, dim3(blocks), dim3(256), 0, h->stream, h->d_est_sums, h->d_cov_sums, h->d_status, h->d_x, h->d_part,
                       (int) h->n, (int) h->p);
    hipLaunchKernelGGL(k_outer_reduce, dim3(1), dim3(256), 0, h->stream, h->d_part, h->d_pvec, blocks, (int) h->p);
    HIP_TRY(hipGetLastError());
    return allreduce_to_host(h, h->d_pvec, 3 * h->p + 4, totals);
}

int dn_outer_allreduce(dn_handle h, double *totals)
{
    if (!h || !h->comm) return fail(DN_E_STATE, "dn_outer_allreduce: dn_comm_create has not been called");
    if (!h->d_xw) return fail(DN_E_STATE, "dn_outer_allreduce: dn_outer_begin has not been called");
    if (!totals) return fail(DN_E_INVALID, "dn_outer_allreduce: null output");
    HIP_TRY(hipSetDevice(h->device));
    const int blocks = (int) std::min<int64_t>(OUT_BLOCKS, (h->n + 3) / 4);
    hipLaunchKernelGGL(k_outer_partials, dim3(blocks), dim3(256), 0, h->stream, h->d_rho, h->d_rhoc, h->d_xw, h->d_trace, h->d_flags, h->d_part,
                       (int) h->n, (int) h->p);
    hipLaunchKernelGGL(k_outer_reduce, dim3(1), dim3(256), 0, h->stream, h->d_part, h->d_pvec, blocks, (int) h->p);
    HIP_TRY(hipGetLastError());
    return allreduce_to_host(h, h->d_pvec, 3 * h->p + 4, totals);
}

int dn_fetch_outer(dn_handle h, double *rho, double *x_adj, double *x_weighted, uint8_t *ran)
{
    if (!h || !h->d_xw) return fail(DN_E_STATE, "dn_fetch_outer: dn_outer_begin has not been called");
    HIP_TRY(hipSetDevice(h->device));
    const size_t np = (size_t) h->n * h->p;
    if (rho) HIP_TRY(hipMemcpyAsync(rho, h->d_rhoc, sizeof(double) * np, hipMemcpyDeviceToHost, h->stream));
    if (x_adj) HIP_TRY(hipMemcpyAsync(x_adj, h->d_xadj, sizeof(double) * np, hipMemcpyDeviceToHost, h->stream));
    if (x_weighted) HIP_TRY(hipMemcpyAsync(x_weighted, h->d_xw, sizeof(double) * np, hipMemcpyDeviceToHost, h->stream));
    if (ran) HIP_TRY(hipMemcpyAsync(ran, h->d_ran, (size_t) h->n * h->n_iter, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return DN_OK;
}

int dn_fetch_rows(dn_handle h, int64_t n_rows, const int64_t *rows, double *rho_raw, int32_t *flags)
{
    if (!h || !h->d_cov) return fail(DN_E_STATE, "dn_fetch_rows: nothing uploaded");
    if (n_rows <= 0 || !rows || !rho_raw || !flags) return fail(DN_E_INVALID, "dn_fetch_rows: bad argument");
    for (int64_t k = 0; k < n_rows; k++) if (rows[k] < 0 || rows[k] >= h->n) return fail(DN_E_INVALID, "dn_fetch_rows: row out of range");
    HIP_TRY(hipSetDevice(h->device));
    int64_t *d_rows = nullptr; double *d_out = nullptr; int32_t *d_fl = nullptr;
    HIP_TRY(hipMalloc(&d_rows, sizeof(int64_t) * (size_t) n_rows));
    hipError_t e = hipMalloc(&d_out, sizeof(double) * (size_t) n_rows * h->p);
    if (e == hipSuccess) e = hipMalloc(&d_fl, sizeof(int32_t) * (size_t) n_rows);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rows, rows, sizeof(int64_t) * (size_t) n_rows, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        const int tot = (int) (n_rows * h->p);
        hipLaunchKernelGGL(k_gather_rows, dim3((tot + 255) / 256), dim3(256), 0, h->stream, h->d_rho, h->d_flags, d_rows, d_out, d_fl, (int) n_rows, (int) h->p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(rho_raw, d_out, sizeof(double) * (size_t) n_rows * h->p, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(flags, d_fl, sizeof(int32_t) * (size_t) n_rows, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void) hipFree(d_rows); if (d_out) (void) hipFree(d_out); if (d_fl) (void) hipFree(d_fl);
    if (e != hipSuccess) return fail(DN_E_HIP, std::string("dn_fetch_rows: ") + hipGetErrorString(e));
    return DN_OK;
}

// First touch of a large, freshly allocated host buffer from several threads at once (one write per 4 KiB page; the caller
// overwrites every byte afterwards): a single thread faults ~4 GB of new pages in 0.15-0.2 s on the GPU box, which is more
// than the copy that fills them takes.
static void prefault_pages(void *ptr, size_t bytes, int n_threads)
{
    if (!ptr || bytes < ((size_t) 64 << 20)) return;
    char *base = (char *) ptr;
    const size_t page = 4096, n_pages = (bytes + page - 1) / page;
    n_threads = std::max(1, std::min(n_threads, 16));
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++)
        th.emplace_back([=]() {
            const size_t lo = n_pages * (size_t) t / (size_t) n_threads, hi = n_pages * (size_t) (t + 1) / (size_t) n_threads;
            for (size_t k = lo; k < hi; k++) { volatile char *q = base + std::min(k * page, bytes - 1); *q = 0; }
        });
    for (auto &t : th) t.join();
}

int dn_fetch_estimates(dn_handle h, double *out)
{
    if (!h || !h->d_cov) return fail(DN_E_STATE, "dn_fetch_estimates: nothing uploaded");
    if (!h->have_estimate_state) return fail(DN_E_STATE, "dn_fetch_estimates: last iteration did not run with want_estimates = 1");
    if (!out) return fail(DN_E_INVALID, "dn_fetch_estimates: null output");
    HIP_TRY(hipSetDevice(h->device));
    if (!h->d_est) HIP_TRY(hipMalloc(&h->d_est, sizeof(double) * (size_t) h->total));
    dn::EstArgs a;
    std::memset(&a, 0, sizeof(a));
    a.cov = h->d_cov; a.goff = h->d_goff; a.glen = h->d_glen; a.kfin = h->d_kfin; a.emode = h->d_emode;
    a.svec = h->d_svec; a.svoff = h->d_svoff; a.out = h->d_est; a.n_genes = (int32_t) h->n; a.p = h->p;
    for (int i = 0; i < dn::P_MAX; i++) a.scale[i] = i < h->p ? h->last_scale[i] : 1.0;
    h->ks->est(a, h->d_tile_gene, h->d_tile_col, (int) h->n_tiles, h->stream);
    HIP_TRY(hipGetLastError());
    prefault_pages(out, sizeof(double) * (size_t) h->total, (int) std::thread::hardware_concurrency());     // while the kernel runs
    HIP_TRY(hipMemcpyAsync(out, h->d_est, sizeof(double) * (size_t) h->total, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return DN_OK;
}

int dn_fetch_estimates_subset(dn_handle h, int64_t n_sel, const int64_t *gene_ids, double *out)
{
    if (!h || !h->d_cov) return fail(DN_E_STATE, "dn_fetch_estimates_subset: nothing uploaded");
    if (!h->have_estimate_state) return fail(DN_E_STATE, "dn_fetch_estimates_subset: last iteration did not run with want_estimates = 1");
    if (n_sel <= 0 || !gene_ids || !out) return fail(DN_E_INVALID, "dn_fetch_estimates_subset: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    std::vector<int64_t> ooff(h->n, -1);
    std::vector<int32_t> tg, tc;
    int64_t total = 0;
    for (int64_t k = 0; k < n_sel; k++) {
        const int64_t g = gene_ids[k];
        if (g < 0 || g >= h->n) return fail(DN_E_INVALID, "dn_fetch_estimates_subset: gene id out of range");
        if (ooff[g] >= 0) return fail(DN_E_INVALID, "dn_fetch_estimates_subset: duplicate gene id");
        ooff[g] = total;
        total += (int64_t) h->p * h->glen[g];
        for (int32_t c = 0; c < h->glen[g]; c += 256) { tg.push_back((int32_t) g); tc.push_back(c); }
    }
    double *d_out = nullptr; int64_t *d_ooff = nullptr; int32_t *d_tg = nullptr, *d_tc = nullptr;
    hipError_t le = hipMalloc(&d_out, sizeof(double) * (size_t) total);            // nothing leaks on a failure half-way
    if (le == hipSuccess) le = hipMalloc(&d_ooff, sizeof(int64_t) * (size_t) h->n);
    if (le == hipSuccess) le = hipMalloc(&d_tg, sizeof(int32_t) * tg.size());
    if (le == hipSuccess) le = hipMalloc(&d_tc, sizeof(int32_t) * tc.size());
    if (le == hipSuccess) le = hipMemcpyAsync(d_ooff, ooff.data(), sizeof(int64_t) * (size_t) h->n, hipMemcpyHostToDevice, h->stream);
    if (le == hipSuccess) le = hipMemcpyAsync(d_tg, tg.data(), sizeof(int32_t) * tg.size(), hipMemcpyHostToDevice, h->stream);
    if (le == hipSuccess) le = hipMemcpyAsync(d_tc, tc.data(), sizeof(int32_t) * tc.size(), hipMemcpyHostToDevice, h->stream);
    if (le != hipSuccess) {
        if (d_out) (void) hipFree(d_out);
        if (d_ooff) (void) hipFree(d_ooff);
        if (d_tg) (void) hipFree(d_tg);
        if (d_tc) (void) hipFree(d_tc);
        return fail(DN_E_HIP, std::string("dn_fetch_estimates_subset: ") + hipGetErrorString(le));
    }
    dn::EstArgs a;
    std::memset(&a, 0, sizeof(a));
    a.cov = h->d_cov; a.goff = h->d_goff; a.glen = h->d_glen; a.kfin = h->d_kfin; a.emode = h->d_emode;
    a.svec = h->d_svec; a.svoff = h->d_svoff; a.out = d_out; a.ooff = d_ooff; a.n_genes = (int32_t) h->n; a.p = h->p;
    for (int i = 0; i < dn::P_MAX; i++) a.scale[i] = i < h->p ? h->last_scale[i] : 1.0;
    h->ks->est(a, d_tg, d_tc, (int) tg.size(), h->stream);
    le = hipGetLastError();
    if (le == hipSuccess) le = hipMemcpyAsync(out, d_out, sizeof(double) * (size_t) total, hipMemcpyDeviceToHost, h->stream);
    if (le == hipSuccess) le = hipStreamSynchronize(h->stream);
    (void) hipFree(d_out); (void) hipFree(d_ooff); (void) hipFree(d_tg); (void) hipFree(d_tc);
    if (le != hipSuccess) return fail(DN_E_HIP, std::string("dn_fetch_estimates_subset: ") + hipGetErrorString(le));
    return DN_OK;
}

double dn_last_kernel_ms(dn_handle h) { return h ? (double) h->last_ms : 0.0; }
double dn_last_init_ms(dn_handle h) { return h ? (double) h->last_init_ms : 0.0; }
double dn_last_span_ms(dn_handle h) { return h ? (double) h->last_span_ms : 0.0; }
double dn_last_rowmax_ms(dn_handle h) { return h ? (double) h->last_rowmax_ms : 0.0; }
const char *dn_init_kernel_name(dn_handle h) { return h ? h->init_name : ""; }
const char *dn_main_kernel_name(dn_handle h) { return (h && h->ks) ? h->ks->baseline_name : ""; }
double dn_class_kernel_ms(dn_handle h, int cls) { return (h && cls >= 0 && cls < dn_handle_s::NCLS) ? (double) h->cls[cls].last_ms : 0.0; }
const char *dn_class_kernel_name(dn_handle h, int cls) { return (h && cls >= 0 && cls < dn_handle_s::NCLS && h->cls[cls].ks && h->cls[cls].n > 0) ? h->cls[cls].ks->baseline_name : ""; }
int32_t dn_tiny_length(dn_handle h) { return h ? h->tiny_len : 0; }
int dn_class_lengths(dn_handle h, int32_t p, int32_t downsample_rate, int32_t *split_len, int32_t *tiny_len)
{
    if (!h || !split_len || !tiny_len) return fail(DN_E_INVALID, "dn_class_lengths: null argument");
    const dn::KernelSet *ks = dn::kernel_set_for(p);
    if (!ks) return fail(DN_E_UNSUPPORTED, "no kernels compiled for p = " + std::to_string(p));
    HIP_TRY(hipSetDevice(h->device));
    *split_len = 0; *tiny_len = 0;
    if (downsample_rate > 1 && p >= 8) return DN_OK;        // the one-wavefront-per-gene family of the down-sampled regime has one class
    const dn::KernelSet *pair = nullptr;
    class_lengths(ks, p, *split_len, *tiny_len, pair);
    return DN_OK;
}
int32_t dn_split_length(dn_handle h) { return h ? h->split_len : 0; }
int dn_synchronize(dn_handle h)
{
    if (!h) return fail(DN_E_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return DN_OK;
}

}  // extern "C"

// stream-copy ceiling ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_copy4(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n4)
{
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t) gridDim.x * blockDim.x) dst[i] = src[i];
}

// read-only stream: every lane keeps four 16-byte loads in flight and folds them into one number (nothing is written but a
// few partial sums): the ceiling a streaming READ kernel can reach on this device ("measured stream-read ceiling")
__global__ __launch_bounds__(256) void k_read4(const float4 *__restrict__ src, float *__restrict__ sink, size_t n4)
{
    float acc = 0.f;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);
    }
    for (; i < n4; i += stride) { const float4 a = src[i]; acc += a.x + a.y + a.z + a.w; }
    if (acc == 12345.678f) sink[0] = acc;                  // never true for the memset pattern: keeps the loads alive
}

// the same with eight non-temporal 16-byte loads per lane in flight
__global__ __launch_bounds__(256) void k_read8nt(const float4 *__restrict__ src, float *__restrict__ sink, size_t n4)
{
    float acc = 0.f;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        typedef float vf4 __attribute__((ext_vector_type(4)));
        vf4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(src + i + k * stride));
#pragma unroll
        for (int k = 0; k < 8; k++) acc += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    for (; i < n4; i += stride) { const float4 a = src[i]; acc += a.x + a.y + a.z + a.w; }
    if (acc == 12345.678f) sink[0] = acc;
}

extern "C" double dn_measure_read_gbps(dn_handle h, int64_t bytes, int reps)
{
    if (!h || bytes < (1 << 20)) return 0.0;
    if (hipSetDevice(h->device) != hipSuccess) return 0.0;
    float4 *a = nullptr; float *sink = nullptr;
    const size_t n4 = (size_t) bytes / sizeof(float4);
    if (hipMalloc(&a, n4 * sizeof(float4)) != hipSuccess) return 0.0;
    if (hipMalloc(&sink, 256) != hipSuccess) { (void) hipFree(a); return 0.0; }
    (void) hipMemsetAsync(a, 1, n4 * sizeof(float4), h->stream);
    double best = 0.0;
    for (int mult : {4, 8, 16, 32}) {                      // workgroups per CU: the best grid is the ceiling
        for (int r = 0; r < reps + 1; r++) {
            (void) hipEventRecord(h->ev0, h->stream);
            hipLaunchKernelGGL(k_read4, dim3(h->n_cus * mult), dim3(256), 0, h->stream, a, sink, n4);
            (void) hipEventRecord(h->ev1, h->stream);
            (void) hipStreamSynchronize(h->stream);
            float ms = 0.f;
            (void) hipEventElapsedTime(&ms, h->ev0, h->ev1);
            if (r > 0 && ms > 0.f) best = std::max(best, (double) n4 * sizeof(float4) / (ms * 1e-3) / 1e9);
        }
        for (int r = 0; r < reps + 1; r++) {
            (void) hipEventRecord(h->ev0, h->stream);
            hipLaunchKernelGGL(k_read8nt, dim3(h->n_cus * mult), dim3(256), 0, h->stream, a, sink, n4);
            (void) hipEventRecord(h->ev1, h->stream);
            (void) hipStreamSynchronize(h->stream);
            float ms = 0.f;
            (void) hipEventElapsedTime(&ms, h->ev0, h->ev1);
            if (r > 0 && ms > 0.f) best = std::max(best, (double) n4 * sizeof(float4) / (ms * 1e-3) / 1e9);
        }
    }
    (void) hipFree(a); (void) hipFree(sink);
    return best;
}

extern "C" double dn_measure_copy_gbps(dn_handle h, int64_t bytes, int reps)
{
    if (!h || bytes < (1 << 20)) return 0.0;
    if (hipSetDevice(h->device) != hipSuccess) return 0.0;
    float4 *a = nullptr, *b = nullptr;
    const size_t n4 = (size_t) bytes / sizeof(float4);
    if (hipMalloc(&a, n4 * sizeof(float4)) != hipSuccess) return 0.0;
    if (hipMalloc(&b, n4 * sizeof(float4)) != hipSuccess) { (void) hipFree(a); return 0.0; }
    (void) hipMemsetAsync(a, 1, n4 * sizeof(float4), h->stream);
    double best = 0.0;
    for (int r = 0; r < reps + 1; r++) {
        (void) hipEventRecord(h->ev0, h->stream);
        hipLaunchKernelGGL(k_copy4, dim3(h->n_cus * 8), dim3(256), 0, h->stream, a, b, n4);
        (void) hipEventRecord(h->ev1, h->stream);
        (void) hipStreamSynchronize(h->stream);
        float ms = 0.f;
        (void) hipEventElapsedTime(&ms, h->ev0, h->ev1);
        if (r > 0 && ms > 0.f) best = std::max(best, 2.0 * (double) n4 * sizeof(float4) / (ms * 1e-3) / 1e9);
    }
    (void) hipFree(a); (void) hipFree(b);
    return best;
}
