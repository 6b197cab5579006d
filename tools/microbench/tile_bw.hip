// HBM bandwidth of the syr2k-like access pattern: every workgroup reads and writes back 128 x 128 fp64 tiles of
// column-major matrices (128 segments of 1 KB, `ld` doubles apart).  Is ld = 4096 (32 KB stride) a bad stride?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256, 2) void rw_tiles(double *A, long ld, long bs, int nt, int rw)
{
    // tile (bx, by) of matrix bz; thread layout as the GEMM epilogue: 16 lanes along a column (128 B), 4 row groups
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *C = A + (long)blockIdx.z * bs + ((long)blockIdx.y * 128) * ld + (long)blockIdx.x * 128;
    double v[64];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = (wave >> 1) * 64 + i * 16 + (lane >> 4) + 4 * r, gj = (wave & 1) * 64 + j * 16 + (lane & 15);
                v[(i * 4 + j) * 4 + r] = C[(long)gi * ld + gj];
            }
    if (rw) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = (wave >> 1) * 64 + i * 16 + (lane >> 4) + 4 * r, gj = (wave & 1) * 64 + j * 16 + (lane & 15);
                    C[(long)gi * ld + gj] = v[(i * 4 + j) * 4 + r] * 1.0000001;
                }
    } else {
        double s = 0;
        for (int q = 0; q < 64; ++q) s += v[q];
        if (s == 1.2345) C[0] = s;
    }
}
int main()
{
    const int n = 4096, nt = n / 128, batch = 48;
    for (long pad : {0L, 16L, 32L, 64L, 272L}) {
        const long ld = n + pad, bs = ld * n + 1024;
        double *A; if (hipMalloc(&A, sizeof(double) * bs * batch) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipMemset(A, 0, sizeof(double) * bs * batch);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rw = 0; rw < 2; ++rw) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(rw_tiles, dim3(nt, nt, batch), dim3(256), 0, 0, A, ld, bs, nt, rw);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double bytes = (double)batch * n * n * 8 * (rw ? 2 : 1);
            printf("ld = %ld (+%ld): %s %.2f ms -> %.2f TB/s\n", ld, pad, rw ? "read+write" : "read only ", ms, bytes / ms * 1e-9);
        }
        hipFree(A);
    }
    return 0;
}
