// dn_assemble.hip -- coverage-matrix assembly on the device.
//
// Replaces the densify-and-slice loop of merge_chrom_coverage (reference reads_coverage_merge.py:283-353): per sample
// the chromosome coverage vector arrives as the index / value arrays of its 1 x N CSR row (reads.py:785-786 writes it
// with scipy.sparse.save_npz), is expanded into a dense fp32 vector in HBM, and every gene's exon intervals
// (union of its exons, ascending) are gathered straight into the packed layout the NMF-OA kernels read
// (gene g at p * sum(lengths[:g]), p rows of lengths[g]).  No float64 host dictionary is needed on the way to HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/degnorm_amd.h"

namespace {

__global__ __launch_bounds__(256) void k_scatter_csr(const int32_t *__restrict__ idx, const float *__restrict__ val,
                                                     int64_t nnz, float *__restrict__ dense, int64_t n)
{
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < nnz; i += (int64_t) gridDim.x * 256) {
        const int64_t j = idx[i];
        if (j >= 0 && j < n) dense[j] = val[i];
    }
}

// one block per (interval, 1024-column chunk): out[dst + t] = dense[src + t]
__global__ __launch_bounds__(256) void k_gather_intervals(const float *__restrict__ dense, int64_t n,
                                                          const int64_t *__restrict__ c_src, const int64_t *__restrict__ c_dst,
                                                          const int32_t *__restrict__ c_len, float *__restrict__ out)
{
    const int64_t s = c_src[blockIdx.x], d = c_dst[blockIdx.x];
    const int len = c_len[blockIdx.x];
    for (int t = threadIdx.x; t < len; t += 256) {
        const int64_t j = s + t;
        out[d + t] = (j >= 0 && j < n) ? dense[j] : 0.0f;
    }
}

thread_local std::string g_asm_err;

}  // namespace

#define ASM_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) { g_asm_err = std::string(#expr) + ": " + hipGetErrorString(e_); rc = DN_E_HIP; goto done; } \
    } while (0)

extern "C" const char *dn_assemble_last_error(void) { return g_asm_err.c_str(); }

extern "C" int dn_assemble_coverage(int device, int64_t chrom_len, int32_t p, const int64_t *nnz,
                                    const int32_t *const *indices, const float *const *values,
                                    int64_t n_genes, const int64_t *lengths,
                                    int64_t n_chunks, const int32_t *chunk_gene, const int64_t *chunk_src,
                                    const int64_t *chunk_dst_in_gene, const int32_t *chunk_len,
                                    float *out_packed, double *device_ms)
{
    int rc = DN_OK;
    float *d_dense = nullptr, *d_out = nullptr, *d_val = nullptr;
    int32_t *d_idx = nullptr, *d_clen = nullptr;
    int64_t *d_csrc = nullptr, *d_cdst = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int64_t total = 0, max_nnz = 1;
    if (chrom_len <= 0 || p < 1 || n_genes < 1 || !lengths || !out_packed || n_chunks < 0) { g_asm_err = "dn_assemble_coverage: bad argument"; return DN_E_INVALID; }
    {
        std::string offs_err;
        // absolute destination of every chunk for sample 0; sample i adds i * L_gene
        int64_t *goff = new int64_t[n_genes + 1];
        goff[0] = 0;
        for (int64_t g = 0; g < n_genes; g++) goff[g + 1] = goff[g] + (int64_t) p * lengths[g];
        total = goff[n_genes];
        int64_t *cdst = new int64_t[n_chunks > 0 ? n_chunks : 1];
        for (int64_t c = 0; c < n_chunks; c++) cdst[c] = goff[chunk_gene[c]] + chunk_dst_in_gene[c];
        for (int i = 0; i < p; i++) if (nnz[i] > max_nnz) max_nnz = nnz[i];

        ASM_TRY(hipSetDevice(device));
        ASM_TRY(hipStreamCreate(&st));
        ASM_TRY(hipEventCreate(&e0));
        ASM_TRY(hipEventCreate(&e1));
        ASM_TRY(hipMalloc(&d_dense, sizeof(float) * (size_t) chrom_len));
        ASM_TRY(hipMalloc(&d_out, sizeof(float) * (size_t) (total > 0 ? total : 1)));
        ASM_TRY(hipMalloc(&d_idx, sizeof(int32_t) * (size_t) max_nnz));
        ASM_TRY(hipMalloc(&d_val, sizeof(float) * (size_t) max_nnz));
        ASM_TRY(hipMalloc(&d_csrc, sizeof(int64_t) * (size_t) (n_chunks > 0 ? n_chunks : 1)));
        ASM_TRY(hipMalloc(&d_cdst, sizeof(int64_t) * (size_t) (n_chunks > 0 ? n_chunks : 1)));
        ASM_TRY(hipMalloc(&d_clen, sizeof(int32_t) * (size_t) (n_chunks > 0 ? n_chunks : 1)));
        if (n_chunks > 0) {
            ASM_TRY(hipMemcpyAsync(d_csrc, chunk_src, sizeof(int64_t) * (size_t) n_chunks, hipMemcpyHostToDevice, st));
            ASM_TRY(hipMemcpyAsync(d_clen, chunk_len, sizeof(int32_t) * (size_t) n_chunks, hipMemcpyHostToDevice, st));
        }
        ASM_TRY(hipEventRecord(e0, st));
        for (int i = 0; i < p; i++) {
            // per-sample destinations: row i of every gene
            int64_t *cd = new int64_t[n_chunks > 0 ? n_chunks : 1];
            for (int64_t c = 0; c < n_chunks; c++) cd[c] = cdst[c] + (int64_t) i * lengths[chunk_gene[c]];
            hipError_t ee = n_chunks > 0 ? hipMemcpyAsync(d_cdst, cd, sizeof(int64_t) * (size_t) n_chunks, hipMemcpyHostToDevice, st) : hipSuccess;
            if (ee == hipSuccess) ee = hipStreamSynchronize(st);
            delete[] cd;
            ASM_TRY(ee);
            ASM_TRY(hipMemsetAsync(d_dense, 0, sizeof(float) * (size_t) chrom_len, st));        // missing file: all zeros (reference :309-316)
            if (nnz[i] > 0) {
                ASM_TRY(hipMemcpyAsync(d_idx, indices[i], sizeof(int32_t) * (size_t) nnz[i], hipMemcpyHostToDevice, st));
                ASM_TRY(hipMemcpyAsync(d_val, values[i], sizeof(float) * (size_t) nnz[i], hipMemcpyHostToDevice, st));
                const int grid = (int) ((nnz[i] + 255) / 256 < 65536 ? (nnz[i] + 255) / 256 : 65536);
                hipLaunchKernelGGL(k_scatter_csr, dim3(grid), dim3(256), 0, st, d_idx, d_val, nnz[i], d_dense, chrom_len);
            }
            if (n_chunks > 0)
                hipLaunchKernelGGL(k_gather_intervals, dim3((unsigned) n_chunks), dim3(256), 0, st, d_dense, chrom_len, d_csrc, d_cdst, d_clen, d_out);
            ASM_TRY(hipGetLastError());
        }
        ASM_TRY(hipEventRecord(e1, st));
        ASM_TRY(hipMemcpyAsync(out_packed, d_out, sizeof(float) * (size_t) total, hipMemcpyDeviceToHost, st));
        ASM_TRY(hipStreamSynchronize(st));
        if (device_ms) { float ms = 0.f; ASM_TRY(hipEventElapsedTime(&ms, e0, e1)); *device_ms = ms; }
    done:
        delete[] goff;
        delete[] cdst;
    }
    if (d_dense) (void) hipFree(d_dense);
    if (d_out) (void) hipFree(d_out);
    if (d_idx) (void) hipFree(d_idx);
    if (d_val) (void) hipFree(d_val);
    if (d_csrc) (void) hipFree(d_csrc);
    if (d_cdst) (void) hipFree(d_cdst);
    if (d_clen) (void) hipFree(d_clen);
    if (e0) (void) hipEventDestroy(e0);
    if (e1) (void) hipEventDestroy(e1);
    if (st) (void) hipStreamDestroy(st);
    return rc;
}
