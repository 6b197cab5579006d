// Sustained fp64 rates of the device: (a) independent v_mfma_f64_16x16x4 back to back, (b) v_mfma_f64_4x4x4,
// (c) vector v_fma_f64.  build: hipcc --offload-arch=gfx950 -O3 mfma_f64_peak.hip -o mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void peak16(double *out, int iters)
{
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x * 1e-3 + i; b[i] = 1.0 + threadIdx.x * 1e-4 * i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void peak4(double *out, int iters)
{
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = 0.0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void peakv(double *out, int iters)
{
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = i;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1e-7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0.0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char **argv)
{
    const int wps = argc > 1 ? atoi(argv[1]) : 1;          // workgroups of 4 waves per CU
    const int blocks = 256 * wps, iters = 20000;
    double *d; hipMalloc(&d, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(peak16<16>, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("mfma_f64_16x16x4 x16 acc: %.2f ms -> %.1f TFLOP/s\n", ms, (double)blocks * 4 * iters * 16 * 2048.0 / ms * 1e-9);
        hipEventRecord(e0);
        hipLaunchKernelGGL(peak16<8>, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("mfma_f64_16x16x4 x8 acc : %.2f ms -> %.1f TFLOP/s\n", ms, (double)blocks * 4 * iters * 8 * 2048.0 / ms * 1e-9);
        hipEventRecord(e0);
        hipLaunchKernelGGL(peak4, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("mfma_f64_4x4x4 x16      : %.2f ms -> %.1f TFLOP/s\n", ms, (double)blocks * 4 * iters * 16 * 512.0 / ms * 1e-9);
        hipEventRecord(e0);
        hipLaunchKernelGGL(peakv, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("v_fma_f64 x16           : %.2f ms -> %.1f TFLOP/s\n", ms, (double)blocks * 4 * iters * 16 * 128.0 / ms * 1e-9);
    }
    return 0;
}
