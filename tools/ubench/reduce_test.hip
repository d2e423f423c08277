// Checks dn::wave_reduce_scatter (csrc/dn_reduce.hpp) against host sums for several N, and prints its cost.
#include "../../degnorm_amd/csrc/dn_reduce.hpp"
#include <cstdio>
#include <cmath>
#include <vector>
template <int N>
__global__ void k(const double *in, double *out, int *ent, long long *cyc)
{
    const int lane = threadIdx.x;
    double g[N];
    for (int e = 0; e < N; e++) g[e] = in[e * 64 + lane];
    const long long t0 = __builtin_amdgcn_s_memtime();
    double s = dn::wave_reduce_scatter<N, double>(g, lane);
    for (int r = 0; r < 15; r++) { for (int e = 0; e < N; e++) g[e] = g[e] * 0.5 + s; s = dn::wave_reduce_scatter<N, double>(g, lane); }
    const long long t1 = __builtin_amdgcn_s_memtime();
    for (int e = 0; e < N; e++) g[e] = in[e * 64 + lane];
    out[lane] = dn::wave_reduce_scatter<N, double>(g, lane) + (s == 1.234e300 ? 1.0 : 0.0);
    ent[lane] = dn::reduce_scatter_entry(lane);
    if (lane == 0) cyc[0] = (t1 - t0) / 16;
}
template <int N> int run()
{
    std::vector<double> h(N * 64), ref(N, 0.0);
    for (int e = 0; e < N; e++) for (int l = 0; l < 64; l++) { h[e * 64 + l] = (double) ((e * 131 + l * 17) % 97) + 0.25 * ((l * 7 + e) % 5); ref[e] += h[e * 64 + l]; }
    double *in, *out; int *ent; long long *cyc;
    hipMalloc(&in, h.size() * 8); hipMalloc(&out, 64 * 8); hipMalloc(&ent, 64 * 4); hipMalloc(&cyc, 8);
    hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, in, out, ent, cyc);
    double o[64]; int en[64]; long long c;
    hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost); hipMemcpy(en, ent, sizeof(en), hipMemcpyDeviceToHost); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    int bad = 0, seen = 0;
    for (int l = 0; l < 64; l++) if (en[l] < N) { seen++; if (o[l] != ref[en[l]]) { if (bad < 4) printf("  N=%d lane %d entry %d: got %.3f want %.3f\n", N, l, en[l], o[l], ref[en[l]]); bad++; } }
    printf("N=%2d: %d entries checked, %d wrong, %lld ticks per reduction\n", N, seen, bad, c);
    return bad + (seen != N);
}
int main()
{
    int bad = run<55>() + run<21>() + run<10>() + run<3>() + run<64>() + run<1>() + run<36>();
    printf(bad ? "FAILED\n" : "all good\n");
    return bad != 0;
}
