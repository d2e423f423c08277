// Microbenchmark: what ONE wave per SIMD pays per vector instruction of each kind the c2 column body is made of, in long
// straight-line streams with literal registers (no compiler in the way): 8 independent instructions x 64 repeats per loop
// trip.  Answers (a) whether the three 64-bit sources of the Gram update collide on VGPR banks (register number mod 4) and
// whether the assignment matters, (b) what an SGPR-pair source costs, (c) the cost of the moves / conversions / unpacks,
// (d) the latency of a dependent fp64 chain.
//   hipcc --offload-arch=gfx950 -O3 -o instr_cost instr_cost.hip && ./instr_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51", \
             "v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71", \
             "v72","v73","v74","v75","s20","s21","s22","a0","a1","a2","a3","a4","a5","a6","a7","scc","memory"

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

// accumulators alternate between the two bank pairs; both multiplicands in banks {0,1}
KERNEL(0, "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n v_fma_f64 v[34:35], v[64:65], v[68:69], v[34:35]\n"
          "v_fma_f64 v[36:37], v[64:65], v[68:69], v[36:37]\n v_fma_f64 v[38:39], v[64:65], v[68:69], v[38:39]\n"
          "v_fma_f64 v[40:41], v[64:65], v[68:69], v[40:41]\n v_fma_f64 v[42:43], v[64:65], v[68:69], v[42:43]\n"
          "v_fma_f64 v[44:45], v[64:65], v[68:69], v[44:45]\n v_fma_f64 v[46:47], v[64:65], v[68:69], v[46:47]\n")
// all three sources in banks {0,1}
KERNEL(1, "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n v_fma_f64 v[36:37], v[64:65], v[68:69], v[36:37]\n"
          "v_fma_f64 v[40:41], v[64:65], v[68:69], v[40:41]\n v_fma_f64 v[44:45], v[64:65], v[68:69], v[44:45]\n"
          "v_fma_f64 v[48:49], v[64:65], v[68:69], v[48:49]\n v_fma_f64 v[52:53], v[64:65], v[68:69], v[52:53]\n"
          "v_fma_f64 v[56:57], v[64:65], v[68:69], v[56:57]\n v_fma_f64 v[60:61], v[64:65], v[68:69], v[60:61]\n")
// accumulator in {2,3}, multiplicands in {0,1} and {2,3}
KERNEL(2, "v_fma_f64 v[34:35], v[64:65], v[66:67], v[34:35]\n v_fma_f64 v[38:39], v[64:65], v[66:67], v[38:39]\n"
          "v_fma_f64 v[42:43], v[64:65], v[66:67], v[42:43]\n v_fma_f64 v[46:47], v[64:65], v[66:67], v[46:47]\n"
          "v_fma_f64 v[50:51], v[64:65], v[66:67], v[50:51]\n v_fma_f64 v[54:55], v[64:65], v[66:67], v[54:55]\n"
          "v_fma_f64 v[58:59], v[64:65], v[66:67], v[58:59]\n v_fma_f64 v[62:63], v[64:65], v[66:67], v[62:63]\n")
// accumulator in {2,3}, both multiplicands in {0,1}
KERNEL(3, "v_fma_f64 v[34:35], v[64:65], v[68:69], v[34:35]\n v_fma_f64 v[38:39], v[64:65], v[68:69], v[38:39]\n"
          "v_fma_f64 v[42:43], v[64:65], v[68:69], v[42:43]\n v_fma_f64 v[46:47], v[64:65], v[68:69], v[46:47]\n"
          "v_fma_f64 v[50:51], v[64:65], v[68:69], v[50:51]\n v_fma_f64 v[54:55], v[64:65], v[68:69], v[54:55]\n"
          "v_fma_f64 v[58:59], v[64:65], v[68:69], v[58:59]\n v_fma_f64 v[62:63], v[64:65], v[68:69], v[62:63]\n")
// diagonal entry: one multiplicand register pair, used twice
KERNEL(4, "v_fma_f64 v[32:33], v[64:65], v[64:65], v[32:33]\n v_fma_f64 v[34:35], v[64:65], v[64:65], v[34:35]\n"
          "v_fma_f64 v[36:37], v[64:65], v[64:65], v[36:37]\n v_fma_f64 v[38:39], v[64:65], v[64:65], v[38:39]\n"
          "v_fma_f64 v[40:41], v[64:65], v[64:65], v[40:41]\n v_fma_f64 v[42:43], v[64:65], v[64:65], v[42:43]\n"
          "v_fma_f64 v[44:45], v[64:65], v[64:65], v[44:45]\n v_fma_f64 v[46:47], v[64:65], v[64:65], v[46:47]\n")
// SGPR-pair multiplicand (u_i, 1/s_i, the step)
KERNEL(5, "v_fma_f64 v[32:33], s[20:21], v[64:65], v[32:33]\n v_fma_f64 v[34:35], s[20:21], v[64:65], v[34:35]\n"
          "v_fma_f64 v[36:37], s[20:21], v[64:65], v[36:37]\n v_fma_f64 v[38:39], s[20:21], v[64:65], v[38:39]\n"
          "v_fma_f64 v[40:41], s[20:21], v[64:65], v[40:41]\n v_fma_f64 v[42:43], s[20:21], v[64:65], v[42:43]\n"
          "v_fma_f64 v[44:45], s[20:21], v[64:65], v[44:45]\n v_fma_f64 v[46:47], s[20:21], v[64:65], v[46:47]\n")
KERNEL(6, "v_mul_f64 v[32:33], v[64:65], v[68:69]\n v_mul_f64 v[34:35], v[64:65], v[68:69]\n v_mul_f64 v[36:37], v[64:65], v[68:69]\n v_mul_f64 v[38:39], v[64:65], v[68:69]\n"
          "v_mul_f64 v[40:41], v[64:65], v[68:69]\n v_mul_f64 v[42:43], v[64:65], v[68:69]\n v_mul_f64 v[44:45], v[64:65], v[68:69]\n v_mul_f64 v[46:47], v[64:65], v[68:69]\n")
KERNEL(7, "v_mul_f64 v[32:33], s[20:21], v[64:65]\n v_mul_f64 v[34:35], s[20:21], v[64:65]\n v_mul_f64 v[36:37], s[20:21], v[64:65]\n v_mul_f64 v[38:39], s[20:21], v[64:65]\n"
          "v_mul_f64 v[40:41], s[20:21], v[64:65]\n v_mul_f64 v[42:43], s[20:21], v[64:65]\n v_mul_f64 v[44:45], s[20:21], v[64:65]\n v_mul_f64 v[46:47], s[20:21], v[64:65]\n")
KERNEL(8, "v_max_f64 v[32:33], v[64:65], v[68:69]\n v_max_f64 v[34:35], v[64:65], v[68:69]\n v_max_f64 v[36:37], v[64:65], v[68:69]\n v_max_f64 v[38:39], v[64:65], v[68:69]\n"
          "v_max_f64 v[40:41], v[64:65], v[68:69]\n v_max_f64 v[42:43], v[64:65], v[68:69]\n v_max_f64 v[44:45], v[64:65], v[68:69]\n v_max_f64 v[46:47], v[64:65], v[68:69]\n")
KERNEL(9, "v_cvt_f64_u32 v[32:33], v71\n v_cvt_f64_u32 v[34:35], v71\n v_cvt_f64_u32 v[36:37], v71\n v_cvt_f64_u32 v[38:39], v71\n"
          "v_cvt_f64_u32 v[40:41], v71\n v_cvt_f64_u32 v[42:43], v71\n v_cvt_f64_u32 v[44:45], v71\n v_cvt_f64_u32 v[46:47], v71\n")
KERNEL(10, "v_accvgpr_read_b32 v32, a0\n v_accvgpr_read_b32 v33, a1\n v_accvgpr_read_b32 v34, a2\n v_accvgpr_read_b32 v35, a3\n"
           "v_accvgpr_read_b32 v36, a4\n v_accvgpr_read_b32 v37, a5\n v_accvgpr_read_b32 v38, a6\n v_accvgpr_read_b32 v39, a7\n")
KERNEL(11, "v_accvgpr_write_b32 a0, v32\n v_accvgpr_write_b32 a1, v33\n v_accvgpr_write_b32 a2, v34\n v_accvgpr_write_b32 a3, v35\n"
           "v_accvgpr_write_b32 a4, v36\n v_accvgpr_write_b32 a5, v37\n v_accvgpr_write_b32 a6, v38\n v_accvgpr_write_b32 a7, v39\n")
KERNEL(12, "v_and_b32 v32, 0xffff, v70\n v_lshrrev_b32 v33, 16, v70\n v_and_b32 v34, 0xffff, v70\n v_lshrrev_b32 v35, 16, v70\n"
           "v_and_b32 v36, 0xffff, v70\n v_lshrrev_b32 v37, 16, v70\n v_and_b32 v38, 0xffff, v70\n v_lshrrev_b32 v39, 16, v70\n")
// the update of one element as a dependent chain: latency of back-to-back dependent fp64 instructions
KERNEL(13, "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n"
           "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n"
           "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n"
           "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n")
// two interleaved dependent chains
KERNEL(14, "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n v_fma_f64 v[34:35], v[64:65], v[68:69], v[34:35]\n"
           "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n v_fma_f64 v[34:35], v[64:65], v[68:69], v[34:35]\n"
           "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n v_fma_f64 v[34:35], v[64:65], v[68:69], v[34:35]\n"
           "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n v_fma_f64 v[34:35], v[64:65], v[68:69], v[34:35]\n")
// Gram row as the kernel writes it: a_i x a_j, a_j walking, accumulators wherever they fall
KERNEL(15, "v_fma_f64 v[32:33], v[64:65], v[66:67], v[32:33]\n v_fma_f64 v[34:35], v[64:65], v[68:69], v[34:35]\n"
           "v_fma_f64 v[36:37], v[64:65], v[72:73], v[36:37]\n v_fma_f64 v[38:39], v[64:65], v[74:75], v[38:39]\n"
           "v_fma_f64 v[40:41], v[66:67], v[68:69], v[40:41]\n v_fma_f64 v[42:43], v[66:67], v[72:73], v[42:43]\n"
           "v_fma_f64 v[44:45], v[66:67], v[74:75], v[44:45]\n v_fma_f64 v[46:47], v[68:69], v[72:73], v[46:47]\n")
// fp64 with an inline constant / a 32-bit literal
KERNEL(16, "v_fma_f64 v[32:33], v[64:65], 0.5, v[32:33]\n v_fma_f64 v[34:35], v[64:65], 0.5, v[34:35]\n"
           "v_fma_f64 v[36:37], v[64:65], 0.5, v[36:37]\n v_fma_f64 v[38:39], v[64:65], 0.5, v[38:39]\n"
           "v_fma_f64 v[40:41], v[64:65], 0.5, v[40:41]\n v_fma_f64 v[42:43], v[64:65], 0.5, v[42:43]\n"
           "v_fma_f64 v[44:45], v[64:65], 0.5, v[44:45]\n v_fma_f64 v[46:47], v[64:65], 0.5, v[46:47]\n")
// alternating SGPR-source and VGPR-only fp64 instructions (does the penalty hide behind a neighbour?)
KERNEL(17, "v_fma_f64 v[32:33], s[20:21], v[64:65], v[32:33]\n v_fma_f64 v[34:35], v[66:67], v[64:65], v[34:35]\n"
           "v_fma_f64 v[36:37], s[20:21], v[64:65], v[36:37]\n v_fma_f64 v[38:39], v[66:67], v[64:65], v[38:39]\n"
           "v_fma_f64 v[40:41], s[20:21], v[64:65], v[40:41]\n v_fma_f64 v[42:43], v[66:67], v[64:65], v[42:43]\n"
           "v_fma_f64 v[44:45], s[20:21], v[64:65], v[44:45]\n v_fma_f64 v[46:47], v[66:67], v[64:65], v[46:47]\n")
// 32-bit moves between the fp64 instructions: accvgpr read, fma, accvgpr write, fma
KERNEL(18, "v_accvgpr_read_b32 v48, a0\n v_fma_f64 v[34:35], v[66:67], v[64:65], v[34:35]\n"
           "v_accvgpr_write_b32 a1, v70\n v_fma_f64 v[38:39], v[66:67], v[64:65], v[38:39]\n"
           "v_accvgpr_read_b32 v49, a2\n v_fma_f64 v[42:43], v[66:67], v[64:65], v[42:43]\n"
           "v_accvgpr_write_b32 a3, v70\n v_fma_f64 v[46:47], v[66:67], v[64:65], v[46:47]\n")

typedef void (*kern_t)(long long *, int);

int main()
{
    long long *st; hipMalloc(&st, sizeof(long long) * 1024);
    const kern_t ks[] = {k0, k1, k2, k3, k4, k5, k6, k7, k8, k9, k10, k11, k12, k13, k14, k15, k16, k17, k18};
    const char *names[] = {
        "v_fma_f64 v,v,v  acc alternating bank pairs, a_i a_j both in banks {0,1}",
        "v_fma_f64 v,v,v  all three sources in banks {0,1}",
        "v_fma_f64 v,v,v  acc {2,3}, a_i {0,1}, a_j {2,3}",
        "v_fma_f64 v,v,v  acc {2,3}, a_i a_j both {0,1}",
        "v_fma_f64 v,v,v  diagonal (a_i twice)",
        "v_fma_f64 s,v,v  SGPR-pair multiplicand",
        "v_mul_f64 v,v",
        "v_mul_f64 s,v",
        "v_max_f64 v,v",
        "v_cvt_f64_u32",
        "v_accvgpr_read_b32",
        "v_accvgpr_write_b32",
        "v_and_b32 / v_lshrrev_b32 (unpack)",
        "v_fma_f64 dependent chain (1 chain)",
        "v_fma_f64 dependent chains (2 interleaved)",
        "v_fma_f64 Gram-like operand walk",
        "v_fma_f64 v,const,v  inline constant",
        "alternating s,v,v and v,v,v fma",
        "alternating accvgpr move and v,v,v fma"};
    const int iters = 2000;
    for (int m = 0; m < (int) (sizeof(ks) / sizeof(ks[0])); m++) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(ks[m], dim3(256), dim3(256), 0, 0, st, iters); hipDeviceSynchronize(); }
        std::vector<long long> h(1024);
        hipMemcpy(h.data(), st, sizeof(long long) * 1024, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double instr = 8.0 * 64.0 * iters;
        printf("%-76s cycles per wave-instruction: fastest %.3f  median %.3f  slowest %.3f\n", names[m], h[0] / instr, h[512] / instr, h[1023] / instr);
    }
    return 0;
}
