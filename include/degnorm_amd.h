/*
 * degnorm_amd.h -- C ABI of the MI355X-native NMF over-approximation core of DegNorm.
 *
 * This is the drop-in boundary for the hot path named in BASELINE.json:
 * the reference is pure Python, so "the reference's FFI for this path" is the set of calls its
 * GeneNMFOA.run() / run_gene_nmfoa_mpi() bodies make per gene; each entry point below names the
 * reference lines (relative to the DegNorm checkout, v0.1.4) it replaces.  The only caller is
 * degnorm_amd/_lib.py (ctypes); INTEGRATION.md shows the binding a DegNorm maintainer would add.
 *
 * Conventions
 *   - plain C types only; the caller allocates every host output; nothing throws across the boundary.
 *   - every function returns DN_OK (0) or a negative DN_E_* code; dn_last_error() gives the text.
 *   - per-gene problems the reference would raise on (ArpackError, empty np.min, svds ValueError:
 *     SURVEY H8) are reported in the per-gene trace status instead of aborting the batch.
 *   - a handle owns one HIP device, one stream, the resident coverage and all scratch; it is not
 *     thread-safe (the reference is called once from the main thread, nmf.py:483).
 *   - genes keep the caller's order on the boundary; the library permutes internally
 *     (longest-first work queue) and un-permutes on output.
 */
#ifndef DEGNORM_AMD_H
#define DEGNORM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DN_OK                 0
#define DN_E_INVALID         -1   /* bad argument / shape (reference: ValueError, nmf.py:469-481)            */
#define DN_E_HIP             -2   /* HIP runtime error (text in dn_last_error)                               */
#define DN_E_STATE           -3   /* call out of order (e.g. iterate before upload)                          */
#define DN_E_UNSUPPORTED     -4   /* p outside the compiled range                                            */
#define DN_E_NO_DEVICE       -5   /* no usable gfx950 device                                                 */

#define DN_TRACE_LEN          48  /* int32 per gene; layout mirrors oracle/nmfoa_oracle.c                    */
/* trace[0] n_hi_cov  [1] #nmf() calls  [2] sum of active columns over calls  [3] exit code
 * [4] loop-exit reason  [5] #dropped bins  [6] status (0 ok; -1 ArpackError, -2 empty min, -3 ValueError,
 *     -4 an eigen-solve left through its step cap: ARPACK's ArpackNoConvergence; the gene's DI row is zeroed like
 *     the other failures, never filled from an unconverged vector)
 * [7] total power-iteration steps of the on-chip eigen-solver  [8..40) drop_idx sequence                    */

typedef struct dn_handle_s *dn_handle;

/* Algorithm parameters of one outer iteration: GeneNMFOA.__init__ (nmf.py:12-53) after its own
 * normalisation (abs/int, min_high_coverage = max(2,.), forced to 2 when downsample_rate > 1). */
typedef struct {
    int32_t nmf_iter;                /* nmf.py:31  T, inner NMF-OA iterations per nmf() call               */
    int32_t bins;                    /* nmf.py:33  B                                                        */
    int32_t min_high_coverage;       /* nmf.py:34,52-53                                                     */
    int32_t downsample_rate;         /* nmf.py:36  take-every rate, 1 = none                                */
    int32_t skip_baseline_selection; /* nmf.py:48                                                           */
    int32_t want_estimates;          /* keep what dn_fetch_estimates needs (last outer iteration, nmf.py:601) */
    int32_t reserved[2];
} dn_params;

/* Library / device ------------------------------------------------------------------------------- */
const char *dn_version(void);
const char *dn_last_error(void);
int  dn_device_count(void);
int  dn_p_supported(int p);                         /* 1 if kernels for p samples are compiled in          */

int  dn_create(int device, dn_handle *out);
int  dn_destroy(dn_handle h);

/* Upload --------------------------------------------------------------------------------------------
 * Replaces: cov_mats = list(cov_dat.values()) held as float64 host arrays (nmf.py:499) and the
 * per-iteration re-scaled copies adjust_coverage_curves makes (nmf.py:142-146, nmf_mpi.py:745-760).
 * Coverage is packed once into HBM as float32 (sample-major p x L_g per gene, genes back to back) and
 * stays resident; scaling by 1/s_i is folded into the kernels' loads.
 *
 * dn_upload_ragged: genes[g] points at a C-contiguous p x lengths[g] matrix of float64 (is_f32 = 0)
 * or float32 (is_f32 = 1).  *inexact (nullable) receives the number of values that are not exactly
 * representable in float32 (DegNorm coverage is integer counts, reads.py:714,773, so normally 0).
 * dn_upload_packed: the same data already packed (offsets in elements, offsets[n] = total).          */
/* Optional, before an upload: the take-every rate the following dn_baseline_iteration calls will use (GeneNMFOA's
 * downsample_rate, nmf.py:36).  Only steers which kernel family serves the data (results do not depend on it): when no
 * gene can keep more than 12 active columns, the row-wise one-wave-per-gene kernels are chosen from p = 8 on.          */
int  dn_set_downsample_hint(dn_handle h, int32_t rate);
/* Step cap of one on-chip eigen-solve in power-step equivalents (default 4000; the reference's ARPACK call has
 * maxiter = 10 n, scipy eigsh via svds, nmf.py:63).  A gene whose solve hits the cap gets status -4.                    */
int  dn_set_solver_step_cap(dn_handle h, int32_t max_steps);
/* Leading columns of the per-gene trace that dn_baseline_iteration copies back (8 .. DN_TRACE_LEN, default all): the
 * counters [0..8) are what a production run reads; the dropped-bin sequence [8..40) is diagnostics (at 50 000 genes the whole
 * trace is 9.6 MB per iteration).  `trace` of dn_baseline_iteration then holds n x cols int32.                            */
int  dn_set_trace_columns(dn_handle h, int32_t cols);
int  dn_upload_ragged(dn_handle h, int64_t n_genes, int32_t p, const void *const *genes,
                      const int64_t *lengths, int32_t is_f32, int32_t n_threads, int64_t *inexact);
int  dn_upload_packed(dn_handle h, int64_t n_genes, int32_t p, const float *packed,
                      const int64_t *lengths);

/* Initialisation pass -------------------------------------------------------------------------------
 * Replaces: par_apply(run_ratio_svd_serial) + est_sums / cov_sums (nmf.py:522-525; nmf_mpi.py:681-703).
 * est_sums[g*p+i] = sum_j max(K_i E_j, x_ij), cov_sums[g*p+i] = sum_j x_ij on the raw coverage.
 * status[g] (nullable): 0 or the per-gene error code (a gene of fewer than 2 columns: the reference's
 * ValueError; for p >= 17 also a gene of more than 2^24 columns -- its rows are addressed by 32-bit offsets).      */
int  dn_ratio_svd_sums(dn_handle h, double *est_sums, double *cov_sums, int32_t *status);

/* One outer DegNorm iteration over all resident genes ------------------------------------------------
 * Replaces: adjust_coverage_curves + par_apply_baseline_selection's per-gene work
 * (nmf.py:563-566 -> :142-146, :189-372; nmf_mpi.py:745-785), without the rho clip (host, nmf.py:398-399).
 * scale[p]: current scale factors.  ds_start (nullable unless downsample_rate > 1): per-gene
 * systematic-sample start offset in [0, rate) (nmf.py:422).
 * rho[n*p] (unclipped), flags[n] (ran_baseline_selection) -- both NULL: kept on the device for the dn_outer_* calls;
 * trace[n*DN_TRACE_LEN] (nullable).                                                                  */
int  dn_baseline_iteration(dn_handle h, const double *scale, const dn_params *prm, const int64_t *ds_start,
                           double *rho, int32_t *flags, int32_t *trace);

/* The outer DegNorm update on the device ---------------------------------------------------------------
 * Replaces: the O(n p) host arithmetic between two baseline-selection sweeps -- the DI clip (nmf.py:398-399),
 * correct_di_scores (:148-158), x_adj (:575, :581), the normalisation of the weighted counts (:584-587) and the
 * ran_baseline_selection column (:403); nmf_mpi.py:821-838 on rank 0.  With it dn_baseline_iteration may be called with
 * rho = flags = NULL: the n x p DI matrix stays in HBM and the host sees 3p + 4 numbers per iteration.
 *   dn_outer_begin     x_weighted (n x p, after the initial normalisation, nmf.py:534) -> device; degnorm_iter columns of flags
 *   dn_outer_partials  after dn_baseline_iteration: partials[0:p] = sum over touched genes of x_w / (1 - rho),
 *                      [p:2p] = sum over untouched genes (rho.max() == 0) of x_w, [2p:3p] = sum of x_w,
 *                      [3p] = #untouched, [3p+1] = #genes with trace status != 0, [3p+2] = #genes with status -4,
 *                      [3p+3] = #genes sent through baseline selection (ran_baseline_selection[:, iter].sum(), nmf.py:571)
 *                      -- 3p + 4 doubles
 *                      (sums in a fixed order; the caller all-reduces them over the GPUs)
 *   dn_outer_apply     rho[untouched] = avg_di (NULL: none untouched); x_adj = x_w / (1 - rho); x_w /= norm; flags -> column iter
 *   dn_fetch_outer     final rho / x_adj / x_weighted (n x p each) and ran_baseline_selection (n x degnorm_iter bytes); any may be NULL
 *   dn_fetch_rows      raw (unclipped) DI rows and flags of a few genes of the last dn_baseline_iteration (diagnostics)          */
int  dn_outer_begin(dn_handle h, const double *x_weighted, int32_t degnorm_iter);
/* The initial normalisation on the device -------------------------------------------------------------
 * Replaces: rho0 = 1 - cov_sums / (est_sums + 1), the `low` genes (rho0.max() < 0.1) and the per-sample sums of their read
 * counts, nmf.py:524-531 (nmf_mpi.py:681-718 on rank 0) -- O(n p) host arithmetic on two n x p matrices that
 * dn_ratio_svd_sums would otherwise have to copy back -- and x_weighted = x / norm (:533).
 *   dn_init_begin          reads (n x p float64 read counts) -> device, once per upload
 *   dn_ratio_svd_sums      may then be called with est_sums = cov_sums = NULL (the sums stay in HBM)
 *   dn_init_partials       partials[0:p] = sum over the low genes of x, [p:2p] = sum over all genes of x, [3p] = #low genes,
 *                          [3p+1] = #genes whose initial SVD failed ([2p:3p], [3p+2], [3p+3] unused; 3p + 4 doubles); fixed order, for the all-reduce
 *   dn_outer_begin_scaled  dn_outer_begin with x_weighted = reads / norm formed on the device                        */
int  dn_init_begin(dn_handle h, const double *reads);
int  dn_init_partials(dn_handle h, double *partials);
int  dn_outer_begin_scaled(dn_handle h, const double *norm, int32_t degnorm_iter);
int  dn_outer_partials(dn_handle h, double *partials);
/* The same sums left ON THE DEVICE for a device-side collective (RCCL all-reduce over xGMI on the buffer itself, no host
 * hop; nmf_mpi.py:796-838 moves the whole DI matrix through rank 0 instead): *d_partials receives the device address of the
 * 3p + 4 doubles, valid until the next dn_outer_* call on this handle; the library's stream has been synchronised.          */
int  dn_outer_partials_device(dn_handle h, double **d_partials);
int  dn_outer_apply(dn_handle h, const double *avg_di, const double *norm, int32_t iter);
int  dn_fetch_outer(dn_handle h, double *rho, double *x_adj, double *x_weighted, uint8_t *ran);

/* The collective of the sharded run, inside the library ----------------------------------------------------
 * Replaces: the per-iteration traffic of run_gene_nmfoa_mpi -- rank 0 re-scales and re-sends every coverage chunk
 * (nmf_mpi.py:745-760), every worker returns its estimates and DI rows (:796-815) and rank 0 alone updates the scale
 * factors (:821-838).  Here every GPU keeps its genes, and ONE all-reduce of 3p + 4 float64 per outer iteration (RCCL
 * over xGMI, in place on the library's device buffer, on the library's stream) gives every rank what it needs to
 * compute the new scale factors itself.  A host written in any language shards a run with these calls alone:
 *   dn_comm_unique_id   rank 0: DN_COMM_ID_BYTES opaque bytes (ncclGetUniqueId); the host hands them to every rank by
 *                       whatever it has (MPI_Bcast, a file, a socket)
 *   dn_comm_create      every rank, collectively: joins the communicator with the handle's GPU (ncclCommInitRank)
 *   dn_init_allreduce   dn_init_partials summed over the ranks: totals[3p + 4], same layout
 *   dn_outer_allreduce  dn_outer_partials summed over the ranks: totals[3p + 4], same layout; dn_outer_apply follows
 *   dn_comm_allreduce   sums a small host vector (<= 256 doubles) over the ranks in place: a rank WITHOUT genes joins the
 *                       two collectives above with zeros through this call (same count 3p + 4), error counts, ...
 *   dn_comm_library     which librccl was bound ("" if none could be loaded).  RCCL is resolved at run time (a copy the
 *                       process already holds, e.g. PyTorch's, else DN_RCCL_PATH, else the system's): no link-time dependency. */
#define DN_COMM_ID_BYTES 128
int  dn_comm_unique_id(uint8_t *id);
int  dn_comm_create(dn_handle h, const uint8_t *id, int32_t rank, int32_t size);
int  dn_comm_destroy(dn_handle h);
int32_t dn_comm_size(dn_handle h);
const char *dn_comm_library(void);
int  dn_comm_allreduce(dn_handle h, double *buf, int32_t count);
int  dn_init_allreduce(dn_handle h, double *totals);
int  dn_outer_allreduce(dn_handle h, double *totals);
int  dn_fetch_rows(dn_handle h, int64_t n_rows, const int64_t *rows, double *rho_raw, int32_t *flags);

/* Estimated coverage matrices of the last dn_baseline_iteration run with want_estimates = 1 ----------
 * Replaces: the `estimate` output of baseline_selection (nmf.py:355-369), returned by run() (nmf.py:601).
 * out: float64, gene g at element offset p * sum(lengths[:g]), p x L_g row-major.                     */
int  dn_fetch_estimates(dn_handle h, double *out);
/* The same for a chosen subset (plots and reports only read a handful of genes, report.py:97-113,
 * __main__.py:291-316): gene_ids[n_sel] in upload order; out holds the selected genes back to back, in that order. */
int  dn_fetch_estimates_subset(dn_handle h, int64_t n_sel, const int64_t *gene_ids, double *out);

/* Coverage-matrix assembly -------------------------------------------------------------
 * Replaces: the densify-and-slice loop of merge_chrom_coverage (reads_coverage_merge.py:283-353).
 * Per sample i the chromosome coverage is the CSR row written by reads.py:785-786: nnz[i] positions indices[i][]
 * (0-based) with values[i][] (nnz[i] = 0: file missing, imputed as zeros, :309-316).  Each gene's union of exons is
 * given as chunks (<= any length): chunk c copies chunk_len[c] positions starting at chromosome position chunk_src[c]
 * to column chunk_dst_in_gene[c] of gene chunk_gene[c].  out_packed: float32, gene g at p * sum(lengths[:g]),
 * p rows of lengths[g] -- the layout dn_upload_packed takes.  device_ms (nullable): device time of the assembly. */
int  dn_assemble_coverage(int device, int64_t chrom_len, int32_t p, const int64_t *nnz,
                          const int32_t *const *indices, const float *const *values,
                          int64_t n_genes, const int64_t *lengths,
                          int64_t n_chunks, const int32_t *chunk_gene, const int64_t *chunk_src,
                          const int64_t *chunk_dst_in_gene, const int32_t *chunk_len,
                          float *out_packed, double *device_ms);
const char *dn_assemble_last_error(void);

/* Measurement hooks (bench.py) -------------------------------------------------------------------- */
/* Device time in ms of the most recent dn_baseline_iteration's main kernel, measured with HIP events
 * on the library's own stream; kernel name via dn_main_kernel_name().                               */
double dn_last_kernel_ms(dn_handle h);
const char *dn_main_kernel_name(dn_handle h);
/* The same for the most recent dn_ratio_svd_sums (the initial pass over the whole transcripts).       */
double dn_last_init_ms(dn_handle h);
/* Device time in ms of the row-maxima kernel of the most recent upload (one read of the whole packed coverage).          */
double dn_last_rowmax_ms(dn_handle h);
const char *dn_init_kernel_name(dn_handle h);
/* Genes are run in up to three classes: class 0 = genes longer than dn_split_length() (256-thread workgroups, one per CU),
 * class 1 = the others (128-thread workgroups, two per CU), class 2 = genes of at most dn_tiny_length() bases (one wavefront
 * per gene, two genes per 128-thread workgroup; 0 when the class does not exist for this sample count); one kernel launch
 * per non-empty class and outer iteration. */
int32_t dn_split_length(dn_handle h);
/* The same two boundaries for ANY cohort of p samples before anything is uploaded (they depend on p and on the device's
 * register / LDS capacity, not on the data): what a host needs to deal genes to GPUs by predicted cost so that every GPU gets
 * the same share of every class (the reference deals contiguous equal-count chunks, nmf_mpi.py:605).  0 / 0: one class.   */
int  dn_class_lengths(dn_handle h, int32_t p, int32_t downsample_rate, int32_t *split_len, int32_t *tiny_len);
int32_t dn_tiny_length(dn_handle h);
double dn_class_kernel_ms(dn_handle h, int cls);
/* First launch to last end of the class kernels of the most recent dn_baseline_iteration (they overlap).          */
double dn_last_span_ms(dn_handle h);
const char *dn_class_kernel_name(dn_handle h, int cls);
int  dn_synchronize(dn_handle h);
/* Stream-copy ceiling of this device (GB/s, float4 copy of `bytes` bytes, best of `reps`).          */
double dn_measure_copy_gbps(dn_handle h, int64_t bytes, int reps);
/* Stream-READ ceiling of this device (GB/s: `bytes` bytes read with four 16-byte loads per lane in flight and folded into
 * one number, best of `reps`): what a read-only streaming kernel -- the row maxima, the two passes of the initial DI pass --
 * is measured against, beside the 8 TB/s of the data sheet.                                              */
double dn_measure_read_gbps(dn_handle h, int64_t bytes, int reps);

#ifdef __cplusplus
}
#endif
#endif /* DEGNORM_AMD_H */
