// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the access widths the kernels use (4 B and 8 B per lane).
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T> __global__ __launch_bounds__(256) void k_read(const T *p, size_t n, double *out)
{
    double s = 0;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) s += (double) p[i];
    if (s == 1.2345) out[0] = s;
}
template <typename T> __global__ __launch_bounds__(256) void k_write(T *p, size_t n)
{
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) p[i] = (T) i;
}
int main()
{
    const size_t bytes = 4ull << 30;
    void *buf; double *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void) hipMemset(buf, 0, bytes);
    hipLaunchKernelGGL(k_read<float>, dim3(2048), dim3(256), 0, 0, (const float *) buf, bytes / 4, out);
    hipLaunchKernelGGL(k_read<double>, dim3(2048), dim3(256), 0, 0, (const double *) buf, bytes / 8, out);
    hipLaunchKernelGGL(k_write<float>, dim3(2048), dim3(256), 0, 0, (float *) buf, bytes / 4);
    hipLaunchKernelGGL(k_write<double>, dim3(2048), dim3(256), 0, 0, (double *) buf, bytes / 8);
    (void) hipDeviceSynchronize();
    printf("each kernel moved %zu bytes\n", bytes);
    return 0;
}
