// dn_inst.hip -- one translation unit per sample count: hipcc -DDN_P=<p> [-DDN_NT=<threads>] -c dn_inst.hip
// Instantiates the kernels of dn_kernels.hpp for p = DN_P and exports their launchers as dn_kernel_set_p<DN_P>.
#include <cstdio>
#include <mutex>
#include "dn_kernels.hpp"

#ifndef DN_P
#error "compile with -DDN_P=<number of samples>"
#endif
#ifndef DN_NT
#define DN_NT 64
#endif

#if defined(DN_PAIR) && !DN_REG_TIER
#error "the pair build exists only where the register tier does (DN_RT_MIN_P <= p <= DN_RT_MAX_P): fix PAIR_P_LIST in build.py"
#endif

#define DN_MAX_DEVICES 64
#define DN_CAT_(a, b) a##b
#define DN_CAT(a, b) DN_CAT_(a, b)
#define DN_STR_(a) #a
#define DN_STR(a) DN_STR_(a)

namespace dn {

static int launch_baseline(const IterArgs &a, int grid, size_t dyn_lds, hipStream_t s)
{
    // The dynamic-LDS opt-in is a property of the function ON A DEVICE (each device loads its own copy of the code
    // object), so it is tracked per device: a process that opens handles on two GPUs configures both.
    static std::mutex mu;
    static size_t configured[DN_MAX_DEVICES] = {0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int) e;
    if (dev < 0 || dev >= DN_MAX_DEVICES) return (int) hipErrorInvalidDevice;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (dyn_lds > configured[dev]) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_baseline<DN_P, DN_NT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int) dyn_lds);
            if (e != hipSuccess) return (int) e;
            configured[dev] = dyn_lds;
        }
    }
    // (pair build: `grid` counts scratch slots = units; a workgroup carries DN_UNITS of them)
    hipLaunchKernelGGL((k_baseline<DN_P, DN_NT>), dim3((grid + DN_UNITS - 1) / DN_UNITS), dim3(DN_NT * DN_UNITS), dyn_lds, s, a);
    return (int) hipGetLastError();
}

#ifndef DN_PAIR            // the pair build serves baseline iterations only (the initial pass and the estimates run on the main kernel set)
static void launch_init(const InitArgs &a, int grid, hipStream_t s)
{
#if DN_P <= 16
    hipLaunchKernelGGL((k_ratio_svd<DN_P, DN_NT>), dim3(grid), dim3(DN_NT), 0, s, a);
#else
    kernel_set_generic()->init(a, grid, s);      // the per-lane p x p Gram of k_ratio_svd does not fit in registers above 16
#endif
}

static void launch_est(const EstArgs &a, const int32_t *tg, const int32_t *tc, int n_tiles, hipStream_t s)
{
    hipLaunchKernelGGL((k_estimates<DN_P>), dim3(n_tiles), dim3(256), 0, s, a, tg, tc);
}

#endif

static int blocks_per_cu(int which)
{
    int nb = 0;
    hipError_t e;
    if (which == 0) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_baseline<DN_P, DN_NT>, DN_NT * DN_UNITS, 0);
#ifdef DN_PAIR
    else return 0;
#elif DN_P <= 16
    else            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_ratio_svd<DN_P, DN_NT>, DN_NT, 0);
#else
    else return kernel_set_generic()->blocks_per_cu(which);
#endif
    return e == hipSuccess ? nb : 0;
}

#define DN_CAT3(a, b, c, d) DN_CAT(DN_CAT(DN_CAT(a, b), c), d)
#ifdef DN_PAIR
const KernelSet *DN_CAT(DN_CAT(kernel_set_p, DN_P), _pair)()
#else
const KernelSet *DN_CAT3(kernel_set_p, DN_P, _nt, DN_NT)()
#endif
{
    static char name[64];
    snprintf(name, sizeof(name), "k_baseline<%d,%d>", (int) DN_P, (int) DN_NT);
    static const KernelSet ks = {
        DN_P, DN_NT, launch_baseline,
#ifdef DN_PAIR
        nullptr, nullptr,
#else
        launch_init, launch_est,
#endif
        blocks_per_cu,
        DN_UNITS * (sizeof(Smem<DN_P, DN_NT>) + sizeof(GeneState<DN_P>)),
        name,
        DN_REG_TIER ? rt_save_bytes<DN_P, DN_NT>() : 0,
        DN_REG_TIER ? rt_cols<DN_P, true>() * DN_NT : 0,
        DN_UNITS,
    };
    return &ks;
}

}  // namespace dn
