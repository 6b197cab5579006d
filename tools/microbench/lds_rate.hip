// LDS read throughput on MI355X for the access shapes of sbr2.hip's sb16r_kernel: NW waves of one workgroup (137 KB of LDS: one
// workgroup per CU) issue 64 reads each per round with immediate offsets from one address register (no address arithmetic in
// the loop), then s_waitcnt + s_barrier.  Reports shader ticks per round and per wave-instruction.
//   0: ds_read_b64, lanes linear (lane * 8 bytes)                      1: ds_read_b64, tile rows: lane = (g, j): 1568 g + j doubles
//   2: ds_read_b64, the same address in the 16 lanes of a row (rows 32 doubles apart)
//   3: ds_read2_b64 (two consecutive doubles per lane), broadcast per row   4: ds_read_b128, broadcast per row
//   5: ds_read2_b64, lane j: two consecutive doubles of column j (lane stride 31 doubles)
//   6: ds_read_b64, lane stride 32 doubles (one bank)                  7: ds_write_b64 tile rows (as 1)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int PAT>
__global__ void k(double *out, int iters, long long *ticks)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 17000; i += blockDim.x) lds[i] = i * 1e-9;
    __syncthreads();
    int idx = wv * 1000;
    if (PAT == 0) idx += lane;
    if (PAT == 1 || PAT == 7) idx += 1568 * g + j;
    if (PAT == 2 || PAT == 3 || PAT == 4) idx += 32 * g;
    if (PAT == 5) idx += 1568 * g + 31 * j;
    if (PAT == 6) idx += 32 * lane;
    idx &= ~1;                                             // 16-byte aligned for the b128 case
    const double *p = lds + idx;
    double acc = 0.0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // 16 loads in flight, then their sum: the loads of a group are independent (a wave issues them back to back)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (PAT == 3 || PAT == 4 || PAT == 5) {
                d2 x[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (PAT == 4) x[i] = *reinterpret_cast<const d2 *>(p + 2 * i + 64 * r);
                    else { x[i].x = p[2 * i + 64 * r]; x[i].y = p[2 * i + 1 + 64 * r]; }
                }
                d2 t = x[0];
#pragma unroll
                for (int i = 1; i < 16; ++i) t += x[i];
                acc += t.x + t.y;
            } else if (PAT == 7) {
#pragma unroll
                for (int i = 0; i < 16; ++i) const_cast<double *>(p)[31 * i + 512 * r] = acc;
            } else {
                double x[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = p[31 * i + 512 * r];
#pragma unroll
                for (int i = 0; i < 16; ++i) acc += x[i];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}
template <int P>
void run(int nw)
{
    double *out; long long *tk, h;
    (void)hipMalloc(&out, 256 * 1024 * 8); (void)hipMalloc(&tk, 8);
    const int iters = 5000, lds = 137 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<P>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((k<P>), dim3(256), dim3(64 * nw), lds, 0, out, 50, tk);
    hipLaunchKernelGGL((k<P>), dim3(256), dim3(64 * nw), lds, 0, out, iters, tk);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost);
    printf("pattern %d, waves %d: %8.1f ticks per round of 64 instructions per wave = %5.1f ticks per instruction of the workgroup\n", P, nw,
           (double)h / iters, (double)h / iters / (64.0 * nw));
    (void)hipFree(out); (void)hipFree(tk);
}
template <int P> void pat() { for (int nw : {1, 2, 4, 7}) run<P>(nw); }
int main() { pat<0>(); pat<1>(); pat<2>(); pat<3>(); pat<4>(); pat<5>(); pat<6>(); pat<7>(); return 0; }
