// Rate of the GEMM k-loop alone (operands resident in LDS, no global traffic): which part of the kernel bounds it?
// variants: V=0 v_mfma_f64_16x16x4 (8 LDS reads / 16 MFMA per k-step), V=1 v_mfma_f64_4x4x4 (20 reads / 64 MFMA),
// V=2 4x4x4 with the A strips replicated from the 16x16x4-layout register by DPP row shifts instead of LDS reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int BK = 16, BM = 128, BN = 128, LDA = BM + 16, LDB = BN + 16;
template <int V, int SYNC>
__global__ __launch_bounds__(256, 2) void core(double *out, int ktiles)
{
    __shared__ double As[2][BK * LDA];
    __shared__ double Bs[2][BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * BK * LDA; i += 256) { (&As[0][0])[i] = 1e-3 * (i % 97); (&Bs[0][0])[i] = 1e-3 * (i % 89); }
    __syncthreads();
    double4_t acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0, 0, 0, 0};
    for (int t = 0; t < ktiles; ++t) {
        const int cur = t & 1;
#pragma unroll
        for (int k4 = 0; k4 < BK / 4; ++k4) {
            const int kr = k4 * 4 + (lane >> 4);
            const double *Arow = &As[cur][kr * LDA + wm * 64], *Brow = &Bs[cur][kr * LDB + wn * 64];
            double b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Brow[j * 16 + (lane & 15)];
            if (V == 0) {
                double a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = Arow[i * 16 + (lane & 15)];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    double a4[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) a4[s] = Arow[i * 16 + 4 * s + (lane & 3)];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            acc[i][j][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4[s], b[j], acc[i][j][s], 0, 0, 0);
                }
            }
        }
        if (SYNC) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + tid] = s;
}
template <int V, int SYNC>
void run(const char *name, int blocks, double *d)
{
    const int ktiles = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((core<V, SYNC>), dim3(blocks), dim3(256), 0, 0, d, ktiles);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-44s blocks %4d: %7.2f ms -> %5.1f TFLOP/s\n", name, blocks, ms, (double)blocks * ktiles * 2.0 * BM * BN * BK / ms * 1e-9);
}
int main()
{
    double *d; hipMalloc(&d, sizeof(double) * 1024 * 256);
    for (int blocks : {256, 512}) {
        run<0, 1>("16x16x4, barrier per k-tile", blocks, d);
        run<0, 0>("16x16x4, no barrier", blocks, d);
        run<1, 1>("4x4x4 (LDS strips), barrier per k-tile", blocks, d);
        run<1, 0>("4x4x4 (LDS strips), no barrier", blocks, d);
    }
    return 0;
}
