// Microbenchmark (round 4): what ONE wave per SIMD pays per instruction of the kinds the EIGEN-SOLVE and the reduce are made of --
// the fp64 DPP multiply-add (v_fmac_f64_dpp row_newbcast), the 32-bit DPP moves of a row butterfly, v_readlane, the fp64
// transcendentals (v_rcp / v_rsq / v_sqrt), permlane swaps -- as independent streams (issue cost) and as dependent chains (latency),
// in straight-line streams with literal registers, 8 instructions x 64 repeats per loop trip (same frame as instr_cost.hip).
//   hipcc --offload-arch=gfx950 -O3 -o solver_instr_cost solver_instr_cost.hip && ./solver_instr_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51", \
             "v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71", \
             "v72","v73","v74","v75","s20","s21","s22","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35","s36","s37","s38","s39","vcc","a0","a1","a2","a3","a4","a5","a6","a7","scc","memory"

#define INIT "v_mov_b32 v64, 0\n v_mov_b32 v65, 0x3ff00000\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0x3ff00000\n" \
             "v_mov_b32 v68, 0\n v_mov_b32 v69, 0x3ff00000\n v_mov_b32 v70, 0x00030004\n v_mov_b32 v71, 7\n" \
             "v_mov_b32 v72, 0\n v_mov_b32 v73, 0x3ff00000\n v_mov_b32 v74, 0\n v_mov_b32 v75, 0x3ff00000\n" \
             "s_mov_b32 s20, 0\n s_mov_b32 s21, 0x3ff00000\n" \
             ".irp r,32,34,36,38,40,42,44,46,48,50,52,54,56,58,60,62\n v_mov_b32 v\\r, 0\n .endr\n" \
             ".irp r,33,35,37,39,41,43,45,47,49,51,53,55,57,59,61,63\n v_mov_b32 v\\r, 0x3ff00000\n .endr\n"

#define KERNEL(N, BODY)                                                                                                     \
    __global__ __launch_bounds__(256) void k##N(long long *cyc, int iters)                                                   \
    {                                                                                                                       \
        long long t0, t1;                                                                                                   \
        asm volatile(INIT "s_mov_b32 s22, %2\n s_memtime %0\n s_waitcnt lgkmcnt(0)\n"                                          \
                     "Lloop" #N ":\n .rept 64\n" BODY ".endr\n s_sub_u32 s22, s22, 1\n s_cmp_lg_u32 s22, 0\n s_cbranch_scc1 Lloop" #N "\n" \
                     "s_memtime %1\n s_waitcnt lgkmcnt(0)\n"                                                                 \
                     : "=&s"(t0), "=&s"(t1) : "s"(iters) : CLOB);                                                            \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                     \
    }

KERNEL(0,
          "v_fmac_f64_dpp v[32:33], v[64:65], v[68:69] row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[34:35], v[64:65], v[68:69] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[36:37], v[64:65], v[68:69] row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[38:39], v[64:65], v[68:69] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[40:41], v[64:65], v[68:69] row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[42:43], v[64:65], v[68:69] row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[44:45], v[64:65], v[68:69] row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[46:47], v[64:65], v[68:69] row_newbcast:7 row_mask:0xf bank_mask:0xf\n")

KERNEL(1,
          "v_fmac_f64_dpp v[32:33], v[64:65], v[68:69] row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[32:33], v[64:65], v[68:69] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[32:33], v[64:65], v[68:69] row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[32:33], v[64:65], v[68:69] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[32:33], v[64:65], v[68:69] row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[32:33], v[64:65], v[68:69] row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[32:33], v[64:65], v[68:69] row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[32:33], v[64:65], v[68:69] row_newbcast:7 row_mask:0xf bank_mask:0xf\n")

KERNEL(2,
          "v_fmac_f64_dpp v[32:33], v[64:65], v[68:69] row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[34:35], v[64:65], v[68:69] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[32:33], v[64:65], v[68:69] row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[34:35], v[64:65], v[68:69] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[32:33], v[64:65], v[68:69] row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[34:35], v[64:65], v[68:69] row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[32:33], v[64:65], v[68:69] row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
          "v_fmac_f64_dpp v[34:35], v[64:65], v[68:69] row_newbcast:7 row_mask:0xf bank_mask:0xf\n")

KERNEL(3,
          "s_nop 1\n v_fmac_f64_dpp v[34:35], v[32:33], v[68:69] row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
          "s_nop 1\n v_fmac_f64_dpp v[32:33], v[34:35], v[68:69] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
          "s_nop 1\n v_fmac_f64_dpp v[34:35], v[32:33], v[68:69] row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
          "s_nop 1\n v_fmac_f64_dpp v[32:33], v[34:35], v[68:69] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
          "s_nop 1\n v_fmac_f64_dpp v[34:35], v[32:33], v[68:69] row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
          "s_nop 1\n v_fmac_f64_dpp v[32:33], v[34:35], v[68:69] row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
          "s_nop 1\n v_fmac_f64_dpp v[34:35], v[32:33], v[68:69] row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
          "s_nop 1\n v_fmac_f64_dpp v[32:33], v[34:35], v[68:69] row_newbcast:7 row_mask:0xf bank_mask:0xf\n")

KERNEL(4,
          "v_mov_b64_dpp v[32:33], v[64:65] row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
          "v_mov_b64_dpp v[34:35], v[64:65] row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
          "v_mov_b64_dpp v[36:37], v[64:65] row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
          "v_mov_b64_dpp v[38:39], v[64:65] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
          "v_mov_b64_dpp v[40:41], v[64:65] row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
          "v_mov_b64_dpp v[42:43], v[64:65] row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
          "v_mov_b64_dpp v[44:45], v[64:65] row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
          "v_mov_b64_dpp v[46:47], v[64:65] row_newbcast:7 row_mask:0xf bank_mask:0xf\n")

KERNEL(5,
          "v_mov_b32_dpp v32, v64 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_mov_b32_dpp v33, v64 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_mov_b32_dpp v34, v64 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_mov_b32_dpp v35, v64 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_mov_b32_dpp v36, v64 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_mov_b32_dpp v37, v64 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_mov_b32_dpp v38, v64 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_mov_b32_dpp v39, v64 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n")

KERNEL(6,
          "s_nop 1\n v_mov_b32_dpp v34, v32 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v35, v33 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f64 v[32:33], v[32:33], v[34:35]\n"
          "s_nop 1\n v_mov_b32_dpp v34, v32 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp v35, v33 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f64 v[32:33], v[32:33], v[34:35]\n")

KERNEL(7,
          "v_rcp_f64 v[32:33], v[64:65]\n"
          "v_rcp_f64 v[34:35], v[64:65]\n"
          "v_rcp_f64 v[36:37], v[64:65]\n"
          "v_rcp_f64 v[38:39], v[64:65]\n"
          "v_rcp_f64 v[40:41], v[64:65]\n"
          "v_rcp_f64 v[42:43], v[64:65]\n"
          "v_rcp_f64 v[44:45], v[64:65]\n"
          "v_rcp_f64 v[46:47], v[64:65]\n")

KERNEL(8,
          "v_rsq_f64 v[32:33], v[64:65]\n"
          "v_rsq_f64 v[34:35], v[64:65]\n"
          "v_rsq_f64 v[36:37], v[64:65]\n"
          "v_rsq_f64 v[38:39], v[64:65]\n"
          "v_rsq_f64 v[40:41], v[64:65]\n"
          "v_rsq_f64 v[42:43], v[64:65]\n"
          "v_rsq_f64 v[44:45], v[64:65]\n"
          "v_rsq_f64 v[46:47], v[64:65]\n")

KERNEL(9,
          "v_sqrt_f64 v[32:33], v[64:65]\n"
          "v_sqrt_f64 v[34:35], v[64:65]\n"
          "v_sqrt_f64 v[36:37], v[64:65]\n"
          "v_sqrt_f64 v[38:39], v[64:65]\n"
          "v_sqrt_f64 v[40:41], v[64:65]\n"
          "v_sqrt_f64 v[42:43], v[64:65]\n"
          "v_sqrt_f64 v[44:45], v[64:65]\n"
          "v_sqrt_f64 v[46:47], v[64:65]\n")

KERNEL(10,
          "v_rcp_f64 v[32:33], v[32:33]\n"
          "v_rcp_f64 v[32:33], v[32:33]\n"
          "v_rcp_f64 v[32:33], v[32:33]\n"
          "v_rcp_f64 v[32:33], v[32:33]\n"
          "v_rcp_f64 v[32:33], v[32:33]\n"
          "v_rcp_f64 v[32:33], v[32:33]\n"
          "v_rcp_f64 v[32:33], v[32:33]\n"
          "v_rcp_f64 v[32:33], v[32:33]\n")

KERNEL(11,
          "v_rsq_f64 v[32:33], v[64:65]\n v_mul_f64 v[64:65], v[32:33], v[68:69]\n"
          "v_rsq_f64 v[32:33], v[64:65]\n v_mul_f64 v[64:65], v[32:33], v[68:69]\n"
          "v_rsq_f64 v[32:33], v[64:65]\n v_mul_f64 v[64:65], v[32:33], v[68:69]\n"
          "v_rsq_f64 v[32:33], v[64:65]\n v_mul_f64 v[64:65], v[32:33], v[68:69]\n")

KERNEL(12,
          "v_readlane_b32 s24, v64, 0\n"
          "v_readlane_b32 s25, v64, 1\n"
          "v_readlane_b32 s26, v64, 2\n"
          "v_readlane_b32 s27, v64, 3\n"
          "v_readlane_b32 s28, v64, 4\n"
          "v_readlane_b32 s29, v64, 5\n"
          "v_readlane_b32 s30, v64, 6\n"
          "v_readlane_b32 s31, v64, 7\n")

KERNEL(13,
          "v_readlane_b32 s24, v64, 0\n v_readlane_b32 s25, v65, 0\n v_fma_f64 v[32:33], s[24:25], v[68:69], v[32:33]\n"
          "v_readlane_b32 s24, v64, 1\n v_readlane_b32 s25, v65, 1\n v_fma_f64 v[34:35], s[24:25], v[68:69], v[34:35]\n"
          "v_readlane_b32 s26, v64, 3\n v_readlane_b32 s27, v65, 3\n")

KERNEL(14,
          "v_permlane32_swap_b32 v32, v33\n"
          "v_permlane32_swap_b32 v34, v35\n"
          "v_permlane32_swap_b32 v36, v37\n"
          "v_permlane32_swap_b32 v38, v39\n"
          "v_permlane32_swap_b32 v40, v41\n"
          "v_permlane32_swap_b32 v42, v43\n"
          "v_permlane32_swap_b32 v44, v45\n"
          "v_permlane32_swap_b32 v46, v47\n")

KERNEL(15,
          "v_cmp_lt_f64 vcc, v[64:65], v[68:69]\n"
          "v_cmp_lt_f64 vcc, v[64:65], v[68:69]\n"
          "v_cmp_lt_f64 vcc, v[64:65], v[68:69]\n"
          "v_cmp_lt_f64 vcc, v[64:65], v[68:69]\n"
          "v_cmp_lt_f64 vcc, v[64:65], v[68:69]\n"
          "v_cmp_lt_f64 vcc, v[64:65], v[68:69]\n"
          "v_cmp_lt_f64 vcc, v[64:65], v[68:69]\n"
          "v_cmp_lt_f64 vcc, v[64:65], v[68:69]\n")

KERNEL(16,
          "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n"
          "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n"
          "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n"
          "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n"
          "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n"
          "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n"
          "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n"
          "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n")

KERNEL(17,
          "v_add_f64 v[32:33], v[32:33], v[68:69]\n"
          "v_add_f64 v[32:33], v[32:33], v[68:69]\n"
          "v_add_f64 v[32:33], v[32:33], v[68:69]\n"
          "v_add_f64 v[32:33], v[32:33], v[68:69]\n"
          "v_add_f64 v[32:33], v[32:33], v[68:69]\n"
          "v_add_f64 v[32:33], v[32:33], v[68:69]\n"
          "v_add_f64 v[32:33], v[32:33], v[68:69]\n"
          "v_add_f64 v[32:33], v[32:33], v[68:69]\n")

typedef void (*kern_t)(long long *, int);

int main()
{
    long long *st; (void) hipMalloc(&st, sizeof(long long) * 1024);
    const kern_t ks[] = {k0, k1, k2, k3, k4, k5, k6, k7, k8, k9, k10, k11, k12, k13, k14, k15, k16, k17};
    const char *names[] = {
        "v_fmac_f64_dpp row_newbcast, 8 independent accumulators",
        "v_fmac_f64_dpp row_newbcast, ONE dependent chain",
        "v_fmac_f64_dpp row_newbcast, two interleaved chains",
        "v_fmac_f64_dpp reading the PREVIOUS result (v := chain through src0, s_nop 1 each)",
        "v_mov_b64_dpp row_newbcast, independent",
        "v_mov_b32_dpp quad_perm, independent",
        "row butterfly level as compiled: s_nop 1, 2 x v_mov_b32_dpp, v_add_f64 -- dependent chain (cost per LEVEL = 8 x the figure / 2 ... see note)",
        "v_rcp_f64 independent",
        "v_rsq_f64 independent",
        "v_sqrt_f64 independent",
        "v_rcp_f64 dependent chain",
        "v_rsq_f64 then a dependent v_mul_f64 (pairs)",
        "v_readlane_b32 independent (to 8 SGPRs)",
        "v_readlane_b32 + v_fma_f64 using that SGPR pair (pairs of 2 readlanes + 1 fma)",
        "v_permlane32_swap independent",
        "v_cmp_lt_f64 to vcc, independent",
        "v_fma_f64 dependent chain (reference)",
        "v_add_f64 dependent chain"};
    const double per_trip[] = {8.0, 8.0, 8.0, 16.0, 8.0, 8.0, 8.0, 8.0, 8.0, 8.0, 8.0, 8.0, 8.0, 8.0, 8.0, 8.0, 8.0, 8.0};      // instructions of a BODY (the frame repeats it 64 times per loop trip)
    const int iters = 2000;
    for (int m = 0; m < (int) (sizeof(ks) / sizeof(ks[0])); m++) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(ks[m], dim3(256), dim3(256), 0, 0, st, iters); (void) hipDeviceSynchronize(); }
        std::vector<long long> h(1024);
        (void) hipMemcpy(h.data(), st, sizeof(long long) * 1024, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double bodies = 64.0 * iters;
        printf("%-100s cycles per BODY %.1f, per instruction %.2f (median; fastest wave %.2f)\n", names[m], h[512] / bodies, h[512] / bodies / per_trip[m], h[0] / bodies / per_trip[m]);
    }
    return 0;
}
