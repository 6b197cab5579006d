// Probe the operand layout of v_mfma_f64_4x4x4f64 on gfx950: which (block, row, k) does lane l supply for A,
// (block, k, col) for B, and which (block, row, col) does it receive in D.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const double *a, const double *b, double *d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
int main()
{
    double ha[64], hb[64], hd[64], *da, *db, *dd;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
    // experiment 1: A lane la = 1, B lane lb = 1 -> which D lanes are non-zero?  scan la, lb
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d pairs with:", la);
        for (int lb = 0; lb < 64; ++lb) {
            for (int i = 0; i < 64; ++i) { ha[i] = (i == la); hb[i] = (i == lb); }
            hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
            hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
            for (int i = 0; i < 64; ++i) if (hd[i] != 0.0) printf(" [B%2d->D%2d]", lb, i);
        }
        printf("\n");
        if (la == 7) la = 15;       // lanes 0-7, then 16-23 ... enough to see the pattern
        if (la == 23) break;
    }
    return 0;
}
