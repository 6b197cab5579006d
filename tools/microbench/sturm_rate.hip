// Where do the cycles of the Sturm-count loop of bisect3_kernel go?  The loop of csrc/tridiag.hip::sturm_counts3 with
// parts switched off: V0 = as shipped (3 fp64 ops + sign-history alignbit per row, renormalisation every 8 rows),
// V1 = no sign history, V2 = no renormalisation, V3 = the three fp64 instructions alone.
// build: hipcc --offload-arch=gfx950 -O3 sturm_rate.hip -o sturm_rate ; run: ./sturm_rate [threads per block]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int EPT = 4, HW = 32;
template <int V, int RS>
__global__ __launch_bounds__(512) void k(const double2 *__restrict__ deg, int np, int rounds, double *out, int *outc)
{
    extern __shared__ double2 de[];
    for (int i = threadIdx.x; i <= np; i += blockDim.x) de[i] = deg[i];
    __syncthreads();
    double x[EPT];
    for (int c = 0; c < EPT; ++c) x[c] = -0.9 + 1.8 * ((threadIdx.x * EPT + c) % 997) / 997.0;
    int total = 0; double sink = 0.0;
    for (int rd = 0; rd < rounds; ++rd) {
        double p0[EPT], p1[EPT]; unsigned h[EPT], hp[EPT]; int cnt[EPT];
        const double d0 = de[0].x;
        for (int c = 0; c < EPT; ++c) { p0[c] = 1.0; p1[c] = d0 - x[c]; hp[c] = (unsigned)__double2hiint(p1[c]) >> 31; cnt[c] = hp[c]; h[c] = 0u; }
        for (int ib = 0; ib < np; ib += HW) {
#pragma unroll
            for (int sb = 0; sb < HW / RS; ++sb) {
#pragma unroll
                for (int r = 0; r < RS; ++r) {
                    const double2 v = de[ib + sb * RS + r + 1];
#pragma unroll
                    for (int c = 0; c < EPT; ++c) {
                        const double pn = __builtin_fma(v.x - x[c], p1[c], -(v.y * p0[c]));
                        if (V == 0 || V == 2) h[c] = __builtin_amdgcn_alignbit(h[c], (unsigned)__double2hiint(pn), 31);
                        if (V == 4) h[c] = (h[c] >> 1) | ((unsigned)__double2hiint(pn) & 0x80000000u);     // two full-rate ops?
                        if (V == 5) h[c] += (unsigned)(__double2hiint(pn) ^ __double2hiint(p1[c])) >> 31;     // direct count
                        p0[c] = p1[c]; p1[c] = pn;
                    }
                }
                if (V == 0 || V == 1 || V >= 4) {
#pragma unroll
                    for (int c = 0; c < EPT; ++c) {
                        const int ea = __builtin_amdgcn_frexp_exp(p1[c]), eb = __builtin_amdgcn_frexp_exp(p0[c]);
                        const int ex = -max(ea, eb);
                        p1[c] = __builtin_amdgcn_ldexp(p1[c], ex); p0[c] = __builtin_amdgcn_ldexp(p0[c], ex);
                    }
                }
            }
            if (V == 5) { for (int c = 0; c < EPT; ++c) { cnt[c] += (int)h[c]; h[c] = 0u; } }
            if (V == 0 || V == 2 || V == 4) {
#pragma unroll
                for (int c = 0; c < EPT; ++c) { const unsigned t = __builtin_amdgcn_alignbit(hp[c], h[c], 1); cnt[c] += __builtin_popcount(h[c] ^ t); hp[c] = h[c]; }
            }
        }
        for (int c = 0; c < EPT; ++c) { total += cnt[c]; sink += p1[c]; x[c] += 1e-9; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink; outc[blockIdx.x * blockDim.x + threadIdx.x] = total;
}
template <int V, int RS = 8>
void run(const char *nm, int tpb, int blocks, const double2 *d_de, int np, double *d_o, int *d_c)
{
    const int rounds = 8;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<V, RS>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<V, RS>), dim3(blocks), dim3(tpb), (np + 1) * 16, 0, d_de, np, rounds, d_o, d_c);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double rowchains = (double)blocks * tpb / 64 * EPT * np * rounds;          // (row, chain) steps per wave, all waves
    const double simd_cycles = ms * 1e-3 * 2.4e9 * 1024;
    printf("%-34s %4d thr x %4d blocks: %7.2f ms -> %.1f cycles per (row, chain) per wave at 2.4 GHz\n", nm, tpb, blocks, ms, simd_cycles / rowchains);
}
int main(int argc, char **argv)
{
    const int tpb = argc > 1 ? atoi(argv[1]) : 512, np = 4096;
    double2 *h = (double2 *)malloc((np + 1) * 16);
    for (int i = 0; i <= np; ++i) { h[i].x = 0.3 * sin(0.37 * i); h[i].y = 0.04 + 0.03 * cos(0.11 * i); }
    double2 *d_de; double *d_o; int *d_c;
    hipMalloc(&d_de, (np + 1) * 16); hipMemcpy(d_de, h, (np + 1) * 16, hipMemcpyHostToDevice);
    hipMalloc(&d_o, 8 * 1024 * 1024); hipMalloc(&d_c, 4 * 1024 * 1024);
    for (int blocks : {256, 512}) {
        run<0>("as shipped", tpb, blocks, d_de, np, d_o, d_c);
        run<1>("no sign history", tpb, blocks, d_de, np, d_o, d_c);
        run<2>("no renormalisation", tpb, blocks, d_de, np, d_o, d_c);
        run<3>("three fp64 instructions only", tpb, blocks, d_de, np, d_o, d_c);
        run<4>("history by shift + and_or", tpb, blocks, d_de, np, d_o, d_c);
        run<5>("direct count xor/shift/add", tpb, blocks, d_de, np, d_o, d_c);
        run<0, 16>("as shipped, renorm every 16 rows", tpb, blocks, d_de, np, d_o, d_c);
        run<4, 16>("shift + and_or, renorm every 16", tpb, blocks, d_de, np, d_o, d_c);
    }
    return 0;
}
