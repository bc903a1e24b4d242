// Candidate secp256k1 table multiply in radix 2^29 (9 limbs, lazy) against the shipped radix-2^32 one (field_secp256k1.h):
// 64-bit column accumulators never overflow (9 products of 29 x <=31.8 bits), so no carry instruction follows any
// v_mad_u64_u32; columns are chained through the accumulator (acc >> 29 seeds the next column).  Bare dependent chains
// x <- t*x + c per lane on the whole chip; prints multiplies/s of both and a few lanes for an exactness check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../ecfft_amd/csrc/field_secp256k1.h"
using namespace ecfft;

struct L9 { uint32_t l[9]; };
struct T29 { uint32_t t[9], u[9]; };   // t and t * 2^(29*5) mod p, canonical 29-bit limbs
static constexpr uint32_t M29 = (1u << 29) - 1;

template <bool HAS_C>
__device__ __forceinline__ L9 tmul29(const T29& T, const L9& x, const L9& c, uint32_t one) {
    uint32_t l[14];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        if (HAS_C && k < 9) acc += (uint64_t)c.l[k] * one;
#pragma unroll
        for (int i = 0; i < 9; ++i) { const int j = k - i; if (j >= 0 && j <= 4) acc += (uint64_t)T.t[i] * x.l[j]; }
#pragma unroll
        for (int i = 0; i < 9; ++i) { const int j = k - i; if (j >= 0 && j <= 3) acc += (uint64_t)T.u[i] * x.l[5 + j]; }
        l[k] = (uint32_t)acc & M29; acc >>= 29;
    }
    l[13] = (uint32_t)acc;
    L9 o; acc = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {      // 2^261 = 2^37 + 31264 (mod p): h_k*R^k*(256*R + 31264)
        if (k < 5) acc += (uint64_t)l[9 + k] * 31264u;
        if (k >= 1) acc += (uint64_t)l[9 + k - 1] * 256u;
        o.l[k] = l[k] + ((uint32_t)acc & M29); acc >>= 29;
    }
    o.l[6] = l[6] + (uint32_t)acc; o.l[7] = l[7]; o.l[8] = l[8];
    return o;
}

__global__ __launch_bounds__(256) void k_chain29(const T29* t, L9* x, const L9* c, int iters) {
    size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    T29 tv = t[g]; L9 xv = x[g], cv = c[g];
    uint32_t one; asm volatile("v_mov_b32 %0, 1" : "=v"(one));
#pragma unroll 1
    for (int i = 0; i < iters; ++i) xv = tmul29<true>(tv, xv, cv, one);
    x[g] = xv;
}
__global__ __launch_bounds__(256) void k_chain29_noc(const T29* t, L9* x, const L9* c, int iters) {
    size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    T29 tv = t[g]; L9 xv = x[g], cv = c[g];
#pragma unroll 1
    for (int i = 0; i < iters; ++i) xv = tmul29<false>(tv, xv, cv, 1);
    x[g] = xv;
}
__global__ __launch_bounds__(256) void k_chain32(const Fe256* t, Fe256* x, int iters) {
    size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    Te256 tv = Secp256k1::to_table(t[g]);
    Fe256 xv = x[g], cv = t[g];
#pragma unroll 1
    for (int i = 0; i < iters; ++i) xv = Secp256k1::tmul_add(tv, xv, cv);
    x[g] = Secp256k1::canon(xv);
}

template <class Fn>
static float timed(Fn fn) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) { (void)hipEventRecord(e0); fn(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms; }
    return best;
}

int main() {
    const int iters = 2000;
    for (int waves : {1, 2, 4, 8}) {
        const size_t n = (size_t)256 * 4 * waves * 64, blocks = n / 256;
        std::vector<T29> ht(n); std::vector<L9> hx(n), hc(n); std::vector<Fe256> ft(n);
        srand(7);
        for (size_t i = 0; i < n; ++i) {
            for (int k = 0; k < 9; ++k) { ht[i].t[k] = (uint32_t)rand() & M29; ht[i].u[k] = (uint32_t)rand() & M29; hx[i].l[k] = (uint32_t)rand() & M29; hc[i].l[k] = (uint32_t)rand() & M29; }
            ht[i].t[8] &= 0xffffff; ht[i].u[8] &= 0xffffff;
            for (int k = 0; k < 8; ++k) ft[i].l[k] = (uint32_t)rand() * 3u + k; ft[i].l[7] &= 0x7fffffffu;
        }
        T29* dt; L9 *dx, *dc; Fe256 *d32t, *d32x;
        (void)hipMalloc(&dt, n * sizeof(T29)); (void)hipMalloc(&dx, n * sizeof(L9)); (void)hipMalloc(&dc, n * sizeof(L9)); (void)hipMalloc(&d32t, n * 32); (void)hipMalloc(&d32x, n * 32);
        (void)hipMemcpy(dt, ht.data(), n * sizeof(T29), hipMemcpyHostToDevice); (void)hipMemcpy(dc, hc.data(), n * sizeof(L9), hipMemcpyHostToDevice);
        (void)hipMemcpy(d32t, ft.data(), n * 32, hipMemcpyHostToDevice); (void)hipMemcpy(d32x, ft.data(), n * 32, hipMemcpyHostToDevice);
        if (waves == 1) {   // exactness sample: one multiply, first 4 lanes
            (void)hipMemcpy(dx, hx.data(), n * sizeof(L9), hipMemcpyHostToDevice);
            k_chain29<<<blocks, 256>>>(dt, dx, dc, 1); (void)hipDeviceSynchronize();
            std::vector<L9> o(8); (void)hipMemcpy(o.data(), dx, 8 * sizeof(L9), hipMemcpyDeviceToHost);
            for (int i = 0; i < 8; ++i) {
                printf("CHK");
                for (int k = 0; k < 9; ++k) printf(" %u", ht[i].t[k]);
                for (int k = 0; k < 9; ++k) printf(" %u", ht[i].u[k]);
                for (int k = 0; k < 9; ++k) printf(" %u", hx[i].l[k]);
                for (int k = 0; k < 9; ++k) printf(" %u", hc[i].l[k]);
                for (int k = 0; k < 9; ++k) printf(" %u", o[i].l[k]);
                printf("\n");
            }
        }
        (void)hipMemcpy(dx, hx.data(), n * sizeof(L9), hipMemcpyHostToDevice);
        float a = timed([&] { k_chain29<<<blocks, 256>>>(dt, dx, dc, iters); });
        float b = timed([&] { k_chain29_noc<<<blocks, 256>>>(dt, dx, dc, iters); });
        float c = timed([&] { k_chain32<<<blocks, 256>>>(d32t, d32x, iters); });
        printf("waves/SIMD %d: radix 2^29 t*x+c %.3e mul/s   t*x %.3e mul/s   shipped radix 2^32 t*x+c %.3e mul/s\n", waves, (double)n * iters / (a * 1e-3), (double)n * iters / (b * 1e-3), (double)n * iters / (c * 1e-3));
        (void)hipFree(dt); (void)hipFree(dx); (void)hipFree(dc); (void)hipFree(d32t); (void)hipFree(d32x);
    }
    return 0;
}
