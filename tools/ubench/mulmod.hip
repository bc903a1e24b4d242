// mulmod throughput microbenchmark (gfx950): dependent chain x <- t*x + c mod p per lane, ITER times,
// 8 waves/SIMD.  Reports cycles per wave-level mul_add per SIMD (@2.4 GHz nominal) and field-mul/s.
// Variants are selected with -DVARIANT=n; results are cross-checked against the host implementation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../ecfft_amd/csrc/field_secp256k1.h"
using namespace ecfft;
using F = Secp256k1;

#ifndef ITER
#define ITER 512
#endif

__device__ unsigned long long g_cycles[2];
__global__ __launch_bounds__(256) void k_chain(const Fe256* t, const Fe256* c, Fe256* x) {
    size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fe256 tv = t[g], cv = c[g], xv = x[g];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < ITER; ++i) xv = F::mul_add(tv, xv, cv);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    x[g] = xv;
    if (g == 0) { g_cycles[0] = t1 - t0; }
}

__global__ __launch_bounds__(256) void k_chain2(const Fe256* t, const Fe256* u, const Fe256* c, Fe256* x) {
    size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fe256 tv = t[g], uv = u[g], cv = c[g], xv = x[g];
#pragma unroll 1
    for (int i = 0; i < ITER; ++i) xv = F::mul2_add(tv, uv, xv, cv);
    x[g] = xv;
}

int main() {
    const int blocks = 256 * 8, n = blocks * 256;
    std::vector<Fe256> ht(n), hc(n), hx(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); };
    for (int i = 0; i < n; ++i) for (int l = 0; l < 8; ++l) { ht[i].l[l] = rnd(); hc[i].l[l] = rnd(); hx[i].l[l] = rnd(); }
    for (int i = 0; i < n; ++i) { ht[i].l[7] &= 0x7FFFFFFF; hc[i].l[7] &= 0x7FFFFFFF; hx[i].l[7] &= 0x7FFFFFFF; }
    // a few edge values
    for (int l = 0; l < 8; ++l) { ht[0].l[l] = 0xFFFFFFFF; hx[0].l[l] = 0xFFFFFFFF; hc[0].l[l] = 0xFFFFFFFF; }
    ht[0].l[0] = hx[0].l[0] = hc[0].l[0] = 0xFFFFFC2E; ht[0].l[1] = hx[0].l[1] = hc[0].l[1] = 0xFFFFFFFE;   // p - 1
    Fe256 *dt, *dc, *dx;
    hipMalloc(&dt, n * 32); hipMalloc(&dc, n * 32); hipMalloc(&dx, n * 32);
    hipMemcpy(dt, ht.data(), n * 32, hipMemcpyHostToDevice); hipMemcpy(dc, hc.data(), n * 32, hipMemcpyHostToDevice);
    hipMemcpy(dx, hx.data(), n * 32, hipMemcpyHostToDevice);
    k_chain<<<blocks, 256>>>(dt, dc, dx);
    std::vector<Fe256> out(n);
    hipMemcpy(out.data(), dx, n * 32, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 4096; ++i) {
        Fe256 v = hx[i];
        for (int k = 0; k < ITER; ++k) v = F::mul_add(ht[i], v, hc[i]);
        if (!F::eq(v, out[i])) ++bad;
    }
    printf("verify vs host (4096 lanes x %d iters): %s\n", ITER, bad ? "MISMATCH" : "ok");
    {   // two-table variant: u = t * 2^128 mod p
        std::vector<Fe256> hu(n);
        Fe256 k128 = F::zero(); k128.l[4] = 1;
        for (int i = 0; i < n; ++i) hu[i] = F::mul(ht[i], k128);
        Fe256* du; hipMalloc(&du, n * 32);
        hipMemcpy(du, hu.data(), n * 32, hipMemcpyHostToDevice);
        hipMemcpy(dx, hx.data(), n * 32, hipMemcpyHostToDevice);
        k_chain2<<<blocks, 256>>>(dt, du, dc, dx);
        std::vector<Fe256> out2(n);
        hipMemcpy(out2.data(), dx, n * 32, hipMemcpyDeviceToHost);
        int bad2 = 0;
        for (int i = 0; i < n; ++i) if (!F::eq(out2[i], out[i])) ++bad2;
        printf("two-table variant vs one-table (all %d lanes): %s (%d)\n", n, bad2 ? "MISMATCH" : "ok", bad2);
        bad += bad2;
        hipEvent_t f0, f1; hipEventCreate(&f0); hipEventCreate(&f1);
        for (int wpb = 1; wpb <= 8; wpb *= 2) {
            int grid = 256 * wpb; float best = 1e30f;
            for (int r = 0; r < 5; ++r) {
                (void)hipEventRecord(f0); k_chain2<<<grid, 256>>>(dt, du, dc, dx); (void)hipEventRecord(f1); (void)hipEventSynchronize(f1);
                float ms; (void)hipEventElapsedTime(&ms, f0, f1); if (ms < best) best = ms;
            }
            printf("two-table waves/SIMD %d: %.3f ms, %.3e field-mul/s chip-wide\n", wpb, best, (double)grid * 256 * ITER / (best * 1e-3));
        }
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpb = 1; wpb <= 8; ++wpb) {           // resident 256-thread blocks per CU (4 waves each => wpb waves per SIMD)
        int grid = 256 * wpb;
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            (void)hipEventRecord(e0); k_chain<<<grid, 256>>>(dt, dc, dx); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        double wave_ops = (double)grid * 4 * ITER;
        double cyc = best * 1e-3 * 2.4e9 / (wave_ops / 1024.0);
        unsigned long long cyc_dev[2]; (void)hipMemcpyFromSymbol(cyc_dev, HIP_SYMBOL(g_cycles), sizeof(cyc_dev));
        printf("   [s_memtime delta of wave 0: %llu ticks over ~%.3f ms => %.0f MHz if 1 tick = 1 shader clock]\n", cyc_dev[0], best, cyc_dev[0] / (best * 1e3));
        printf("waves/SIMD %d: %.3f ms, %.1f cycles/wave-op/SIMD @2.4GHz, %.3e field-mul/s chip-wide\n", wpb, best, cyc,
               (double)grid * 256 * ITER / (best * 1e-3));
    }
    return bad != 0;
}
