// Microbenchmark + correctness check of the two eigen-solvers of the hot loop on a REAL sequence of Gram matrices (the T + 1
// matrices of one nmf() call, produced on the host by tools/ubench/solver_ab.py): dn::top_eig_mfma (rounds 1-3: squaring on the
// fp64 matrix cores) against dn::top_eig_dpp (round 4: warm-started shifted power iteration, v_fmac_f64_dpp row_newbcast).
// One wave per SIMD as in the kernel (256 threads, __launch_bounds__(256, 2) + 512-register claim is not needed here: the solver
// alone is far from the register limit), matrix in LDS, carried state, u broadcast at the end.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -DDN_P=10 -DDN_NT=256 -I../../degnorm_amd/csrc -I../../include -o libsolver_ab.so solver_ab.hip
#include "dn_kernels.hpp"
#include <cstdio>
using namespace dn;
constexpr int P = DN_P;
constexpr int NG = P * (P + 1) / 2;

template <int MODE>          // 0: top_eig_mfma, 1: top_eig_dpp, 2: top_eig_dpp with phase stamps
__global__ __launch_bounds__(256) void k(const double *Gs, int nmat, long long *out, double *uout, int reps)
{
    __shared__ double tot[64 > NG + 2 ? 64 : NG + 2];
    constexpr int ZS = (64 > NG + 2 ? 64 : NG + 2) - 1;
    const int tid = threadIdx.x;
    constexpr bool DPP = MODE != 0;
    typename std::conditional<DPP, EigStateD<P>, EigState<P>>::type st;
    double u[P], theta = 0.0;
    long long cyc = 0;
    long long ss[5] = {0, 0, 0, 0, 0};
    int steps = 0;
    for (int rep = 0; rep < reps; rep++) {
        for (int m = 0; m < nmat; m++) {
            __syncthreads();
            if (tid < NG) tot[tid] = Gs[(size_t) m * NG + tid];
            if (tid == 0) tot[ZS] = 0.0;
            __syncthreads();
            if (m == 0) {
                double tr = 0.0;
                for (int i = 0; i < P; i++) tr += tot[i * (i + 1) / 2 + i];
                eig_state_cold<P>(st, tr);
            }
            __syncthreads();
            if (tid < P) tot[tid * (tid + 1) / 2 + tid] -= st.mu;       // the caller hands over G - mu I
            __syncthreads();
            const long long t0 = __builtin_amdgcn_s_memtime();
            int r;
            if constexpr (MODE == 2) r = top_eig_dpp<P, true>(tot, ZS, u, theta, st, m == nmat - 1, m == 0, 4000, ss);
            else if constexpr (MODE == 1) r = top_eig_dpp<P, false>(tot, ZS, u, theta, st, m == nmat - 1, m == 0, 4000, ss);
            else r = top_eig_mfma<P>(tot, ZS, u, theta, st, m == nmat - 1, 4000);
#pragma unroll
            for (int i = 0; i < P; i++) u[i] = uniform(u[i]);
            const long long t1 = __builtin_amdgcn_s_memtime();
            if (m > 0) { cyc += t1 - t0; steps += r; }
            if (rep == 0 && blockIdx.x == 0 && tid == 0) {
                for (int i = 0; i < P; i++) uout[(size_t) m * (P + 1) + i] = u[i];
                uout[(size_t) m * (P + 1) + P] = theta;
            }
        }
    }
    if ((tid & 63) == 0) { out[2 * (blockIdx.x * 4 + (tid >> 6))] = cyc; out[2 * (blockIdx.x * 4 + (tid >> 6)) + 1] = steps; }
    if (blockIdx.x == 0 && tid == 0) for (int i = 0; i < 5; i++) out[2048 + i] = ss[i];
}

// Gs: nmat x NG packed lower triangles; u_out: 2 x nmat x (P + 1) (mfma, dpp: u and theta); res: { cycles per warm solve, steps per warm solve } x 2,
// then the dpp solver's cycles per solve in its phases { load, blind steps, first normalisation, looks, epilogue } (cold solve included)
extern "C" int solver_ab(const double *Gs, int nmat, double *u_out, double *res)
{
    double *dG, *du; long long *out;
    hipMalloc(&dG, sizeof(double) * nmat * NG); hipMalloc(&du, sizeof(double) * nmat * (P + 1)); hipMalloc(&out, sizeof(long long) * (2 * 1024 + 8));
    hipMemcpy(dG, Gs, sizeof(double) * nmat * NG, hipMemcpyHostToDevice);
    const int reps = 20;
    for (int v = 0; v < 3; v++) {
        for (int warm = 0; warm < 2; warm++) {
            if (v == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, dG, nmat, out, du, reps);
            else if (v == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, dG, nmat, out, du, reps);
            else hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, dG, nmat, out, du, reps);
            if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        }
        long long h[2048 + 8];
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        if (v < 2) {
            hipMemcpy(u_out + (size_t) v * nmat * (P + 1), du, sizeof(double) * nmat * (P + 1), hipMemcpyDeviceToHost);
            res[2 * v] = (double) h[0] / ((double) reps * (nmat - 1));
            res[2 * v + 1] = (double) h[1] / ((double) reps * (nmat - 1));
        } else for (int i = 0; i < 5; i++) res[4 + i] = (double) h[2048 + i] / ((double) reps * nmat);
    }
    hipFree(dG); hipFree(du); hipFree(out);
    return 0;
}
