// What a lock-step of a workgroup costs on MI355X: NW waves, one workgroup per CU (137 KB of LDS), ITER rounds of
// { LOADS ds_read_b64 per wave ; STORES ds_write_b64 ; s_waitcnt lgkmcnt(0) ; s_barrier }, for several access patterns:
//   PAT 0: the tiles of sbr2.hip's sb16r_kernel: lane = (row g = lane >> 4, j = lane & 15), address base + 1568 g + j + 31 i
//          (the four rows of a wave 49 columns of 32 doubles apart: the same banks)
//   PAT 1: the same with the rows 16 doubles further apart each (1584 g)
//   PAT 2: linear, address base + lane + 64 i
//   PAT 3: column stride 33 instead of 32: base + 49 * 33 g + j + 32 i
//   PAT 4: the same address in all 16 lanes of a row (broadcast), rows 32 doubles apart
// Prints ns and shader ticks per round.   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int PAT, int LOADS, int STORES>
__global__ void k(double *out, int iters, long long *ticks)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    int base = wv * 997, step = 31;
    if (PAT == 0) base += g * 1568 + j;
    if (PAT == 1) base += g * 1584 + j;
    if (PAT == 2) { base += lane; step = 64; }
    if (PAT == 3) { base += g * 49 * 33 + j; step = 32; }
    if (PAT == 4) { base += g * 32; step = 1; }
    double acc = 0.0, x[16];
    for (int i = 0; i < 16; ++i) x[i] = 0.0;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 1e-9;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) x[i & 15] = lds[(base + step * i) & 16383];
        double y = acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) y += x[i];
        acc = y;
#pragma unroll
        for (int i = 0; i < STORES; ++i) lds[(base + step * i) & 16383] = x[i & 15] + acc;
        base = (base + 512) & 8191;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}
template <int P, int L, int S>
void run(int nw, const char *what)
{
    double *out; long long *tk, h;
    (void)hipMalloc(&out, 256 * 1024 * 8); (void)hipMalloc(&tk, 8);
    const int iters = 20000, lds = 137 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<P, L, S>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<P, L, S>), dim3(256), dim3(64 * nw), lds, 0, out, 100, tk);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<P, L, S>), dim3(256), dim3(64 * nw), lds, 0, out, iters, tk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost);
    printf("pattern %d: %2d loads %2d stores, waves %d: %7.1f ns per round, %7.1f ticks\n", P, L, S, nw, ms * 1e6 / iters, (double)h / iters);
    (void)hipFree(out); (void)hipFree(tk);
}
template <int P>
void pat()
{
    for (int nw : {1, 7}) run<P, 0, 0>(nw, "");
    for (int nw : {1, 7}) run<P, 16, 0>(nw, "");
    for (int nw : {1, 7}) run<P, 32, 0>(nw, "");
    for (int nw : {1, 7}) run<P, 16, 16>(nw, "");
}
int main()
{
    pat<0>(); pat<1>(); pat<2>(); pat<3>(); pat<4>();
    return 0;
}
