// Does the matrix pipe run UNDER a wave's own integer multiplies?  ("what is left in the row pass": a two-tile row kernel
// would issue tile k's MFMAs between the multiplies of tile k + 1's sweeps.)  One workgroup of 512 threads per CU; per iteration every
// thread does MUL dependent secp256k1 table multiplies (register operands, no memory) and NM independent v_mfma_i32_32x32x32_i8
// (4 accumulators round robin, register operands).  If the pipes overlap inside a wave, time(MUL, NM) ~ max(time(MUL, 0), time(0, NM)).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../ecfft_amd/csrc/field_secp256k1.h"
using namespace ecfft;
using F = Secp256k1;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MUL, int NM>
__global__ __launch_bounds__(512, 2) void k(const Te256* __restrict__ tb, Fe256* out, int iters) {
    const uint32_t tid = threadIdx.x;
    Te256 t = tb[tid & 255];
    Fe256 x = F::zero(); x.l[0] = tid * 2654435761u + 1; x.l[5] = blockIdx.x;
    v4i A = {(int)tid, 2, 3, 4}, B = {5, (int)blockIdx.x, 7, 8};
    v16i acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < (MUL > NM ? MUL : NM); ++m) {
            if (m < NM) acc[m & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc[m & 3], 0, 0, 0);
            if (m < MUL) x = F::tmul_add(t, x, x);
        }
    }
    int s = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[a][i];
    x.l[7] ^= (uint32_t)s & 1u;
    out[(size_t)blockIdx.x * 512 + tid] = x;
}

template <int MUL, int NM>
double run(const Te256* tb, Fe256* out, int blocks) {
    const int iters = 2000;
    k<MUL, NM><<<blocks, 512>>>(tb, out, iters); (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); k<MUL, NM><<<blocks, 512>>>(tb, out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters;
    printf("  %d multiplies + %2d MFMAs per iteration, %4d workgroups: %.3f us per iteration\n", MUL, NM, blocks, us);
    return us;
}

int main() {
    Te256* tb; Fe256* out;
    (void)hipMalloc(&tb, 256 * sizeof(Te256)); (void)hipMemset(tb, 0x5a, 256 * sizeof(Te256));
    (void)hipMalloc(&out, (size_t)512 * 512 * sizeof(Fe256));
    for (int blocks : {256, 512}) {
        printf("%d workgroups of 512 threads (%d per CU):\n", blocks, blocks / 256);
        const double a = run<2, 0>(tb, out, blocks), b = run<0, 8>(tb, out, blocks), c = run<2, 8>(tb, out, blocks);
        const double d = run<2, 4>(tb, out, blocks), e = run<0, 4>(tb, out, blocks), f = run<2, 16>(tb, out, blocks), g = run<0, 16>(tb, out, blocks);
        printf("  -> 2 mul + 8 MFMA: %.3f against max %.3f / sum %.3f;  2 + 4: %.3f against max %.3f / sum %.3f;  2 + 16: %.3f against max %.3f / sum %.3f\n",
               c, a > b ? a : b, a + b, d, a > e ? a : e, a + e, f, a > g ? a : g, a + g);
    }
    return 0;
}
