// dn_reduce.hpp -- register-only reduce-scatter of N <= 64 per-lane fp64 values over one wavefront (gfx950).
//
// Six halving levels, one per lane-id bit, each turning two registers into one:
//   bit 5  v_permlane32_swap     lanes 0-31 keep a[l] + a[l+32], lanes 32-63 keep b[l-32] + b[l]
//   bit 4  v_permlane16_swap     16-lane rows 0..3 keep a.row0+a.row1, b.row0+b.row1, a.row2+a.row3, b.row2+b.row3
//   bit 3  DPP row_ror:8         written per bank (4 lanes): lanes 0-7 of a row keep a, lanes 8-15 keep b
//   bit 2  DPP row_shl/shr:4     banks 0,2 keep a, banks 1,3 keep b
//   bit 1  quad_perm [2,3,0,1]   lane pairs inside a quad, chosen with v_cndmask
//   bit 0  quad_perm [1,0,3,2]
// After the last level lane l holds the wave total of entry bitrev6(l) (meaningful where that index is < N); an
// unpaired register at a level is simply folded onto itself.  ~3.3 VALU operations per value, no LDS traffic, and a
// fixed summation order (so every wave that reduces the same data gets the same bits).
#pragma once
#include <hip/hip_runtime.h>

namespace dn {

typedef unsigned dn_uint2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double swap32_add(double a, double b)
{
    const dn_uint2 lo = __builtin_amdgcn_permlane32_swap((unsigned) __double2loint(a), (unsigned) __double2loint(b), false, false);
    const dn_uint2 hi = __builtin_amdgcn_permlane32_swap((unsigned) __double2hiint(a), (unsigned) __double2hiint(b), false, false);
    return __hiloint2double((int) hi.x, (int) lo.x) + __hiloint2double((int) hi.y, (int) lo.y);
}
__device__ __forceinline__ double swap16_add(double a, double b)
{
    const dn_uint2 lo = __builtin_amdgcn_permlane16_swap((unsigned) __double2loint(a), (unsigned) __double2loint(b), false, false);
    const dn_uint2 hi = __builtin_amdgcn_permlane16_swap((unsigned) __double2hiint(a), (unsigned) __double2hiint(b), false, false);
    return __hiloint2double((int) hi.x, (int) lo.x) + __hiloint2double((int) hi.y, (int) lo.y);
}

// Cross-lane moves inside a row of 16 lanes (DPP).  bound_ctrl with full masks: every lane that matters has a valid
// source, so no "old" value has to be materialised first.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// lanes of the banks in BANKS take src (permuted by CTRL), the others keep old
template <int CTRL, int BANKS>
__device__ __forceinline__ double dpp_merge(double old, double src)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xf, BANKS, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xf, BANKS, false);
    return __hiloint2double(hi, lo);
}

constexpr int DPP_IDENT = 0xE4;          // quad_perm [0,1,2,3]
constexpr int DPP_QX1 = 0xB1;            // quad_perm [1,0,3,2]
constexpr int DPP_QX2 = 0x4E;            // quad_perm [2,3,0,1]
constexpr int DPP_SHL4 = 0x104;          // row_shl:4   dst[l] = src[l + 4]
constexpr int DPP_SHR4 = 0x114;          // row_shr:4   dst[l] = src[l - 4]
constexpr int DPP_ROR8 = 0x128;          // row_ror:8   dst[l] = src[l ^ 8]

// lanes of BANKS_A: a[l] + a[partner]; other lanes: b[l] + b[partner]
template <int TO_A, int TO_B, int BANKS_A>
__device__ __forceinline__ double rs_banks(double a, double b)
{
    double y = dpp_mov<TO_A>(a);
    y = dpp_merge<TO_B, (~BANKS_A) & 0xf>(y, b);
    const double x = dpp_merge<DPP_IDENT, BANKS_A>(b, a);
    return x + y;
}
// lanes with (lane & BIT) == 0: a[l] + a[l ^ BIT]; the others: b[l] + b[l ^ BIT]     (BIT = 1 or 2, inside a quad)
template <int QP>
__device__ __forceinline__ double rs_quad(double a, double b, bool take_b)
{
    const double x = take_b ? b : a, z = take_b ? a : b;
    return x + dpp_mov<QP>(z);
}

__device__ __forceinline__ int reduce_scatter_entry(int lane) { return (int) (__brev((unsigned) lane) >> 26); }

template <int N, typename VT>
__device__ __forceinline__ double wave_reduce_scatter(const VT (&g)[N], int lane)
{
    static_assert(N >= 1 && N <= 64, "one entry per lane at most");
    constexpr int R1 = (N + 1) / 2, R2 = (R1 + 1) / 2, R3 = (R2 + 1) / 2, R4 = (R3 + 1) / 2, R5 = (R4 + 1) / 2;
    static_assert((R5 + 1) / 2 == 1, "six levels reduce 64 entries to one register");
    double h1[R1], h2[R2], h3[R3], h4[R4], h5[R5];
#pragma unroll
    for (int j = 0; j < R1; j++) h1[j] = swap32_add((double) g[2 * j], 2 * j + 1 < N ? (double) g[2 * j + 1] : 0.0);
#pragma unroll
    for (int j = 0; j < R2; j++) h2[j] = swap16_add(h1[2 * j], 2 * j + 1 < R1 ? h1[2 * j + 1] : 0.0);
#pragma unroll
    for (int j = 0; j < R3; j++)
        h3[j] = 2 * j + 1 < R2 ? rs_banks<DPP_ROR8, DPP_ROR8, 0x3>(h2[2 * j], h2[2 * j + 1]) : h2[2 * j] + dpp_mov<DPP_ROR8>(h2[2 * j]);
#pragma unroll
    for (int j = 0; j < R4; j++)
        h4[j] = 2 * j + 1 < R3 ? rs_banks<DPP_SHL4, DPP_SHR4, 0x5>(h3[2 * j], h3[2 * j + 1]) : h3[2 * j] + dpp_mov<DPP_SHL4>(h3[2 * j]);
    const bool b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
#pragma unroll
    for (int j = 0; j < R5; j++)
        h5[j] = 2 * j + 1 < R4 ? rs_quad<DPP_QX2>(h4[2 * j], h4[2 * j + 1], b1) : h4[2 * j] + dpp_mov<DPP_QX2>(h4[2 * j]);
    return R5 > 1 ? rs_quad<DPP_QX1>(h5[0], h5[R5 > 1 ? 1 : 0], b0) : h5[0] + dpp_mov<DPP_QX1>(h5[0]);
}

}  // namespace dn
