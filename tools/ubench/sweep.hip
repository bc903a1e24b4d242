// Where does the time of ONE LDS butterfly sweep of the secp256k1 low-level kernels go?  One workgroup per CU (the small-n
// regime), 512 threads, 512 pairs of a 1024-element LDS tile, decompose sweeps in a loop; variants remove one ingredient
// at a time.  Reports shader cycles per sweep (s_memtime of wave 0).
//   0 full sweep: table loads from global (L2-resident), LDS reads, sub + 2 dependent multiplies, LDS writes, barrier
//   1 tables from registers (no global load)
//   2 no barrier (incorrect, timing only)
//   3 one multiply instead of two
//   4 no LDS traffic (operands stay in registers)
//   5 table loads issued BEFORE the barrier of the previous sweep (software prefetch)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../ecfft_amd/csrc/field_secp256k1.h"
using namespace ecfft;
using F = Secp256k1;

template <int V>
__global__ __launch_bounds__(512, 2) void k_sweep(const Te256* __restrict__ ta, const Te256* __restrict__ tb, Fe256* out, unsigned long long* cyc, int sweeps) {
    __shared__ Fe256 tile[1024];
    const uint32_t tid = threadIdx.x;
    for (uint32_t j = tid; j < 1024; j += 512) { Fe256 v = F::zero(); v.l[0] = j * 2654435761u; v.l[3] = tid; tile[j] = v; }
    __syncthreads();
    Te256 ra = ta[tid], rb = tb[tid];
    Fe256 ka = tile[tid], kb = tile[tid + 512];
    Te256 pa = ta[tid & 255], pb = tb[tid & 255];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int s = 0; s < sweeps; ++s) {
        const uint32_t lh = 8 - (s & 7), h = 1u << lh, i = tid & (h - 1), idx = ((tid >> lh) << (lh + 1)) + i;
        Te256 t0v, t1v;
        if (V == 1) { t0v = ra; t1v = rb; }
        else if (V == 5) { t0v = pa; t1v = pb; }
        else { t0v = ta[(s & 7) * 512 + i]; t1v = tb[(s & 7) * 512 + i]; }
        Fe256 a, b;
        if (V == 4) { a = ka; b = kb; } else { a = tile[idx]; b = tile[idx + h]; }
        Fe256 q1 = F::tmul(t1v, F::sub(b, a));
        Fe256 q0 = V == 3 ? a : F::tmul_add(t0v, q1, a);
        if (V == 4) { ka = q0; kb = q1; } else { tile[idx] = q0; tile[idx + h] = q1; }
        if (V == 5) { const uint32_t lh2 = 8 - ((s + 1) & 7), i2 = tid & ((1u << lh2) - 1); pa = ta[((s + 1) & 7) * 512 + i2]; pb = tb[((s + 1) & 7) * 512 + i2]; }
        if (V != 2) __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (V == 4) { tile[tid] = ka; tile[tid + 512] = kb; }
    __syncthreads();
    out[blockIdx.x * 1024 + tid] = tile[tid]; out[blockIdx.x * 1024 + tid + 512] = tile[tid + 512];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char* name, const Te256* ta, const Te256* tb, Fe256* out, unsigned long long* cyc, int blocks) {
    const int sweeps = 4000;
    k_sweep<V><<<blocks, 512>>>(ta, tb, out, cyc, sweeps); (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); k_sweep<V><<<blocks, 512>>>(ta, tb, out, cyc, sweeps); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks); (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += (double)v; c /= blocks;
    printf("%-44s blocks %4d: %7.1f cycles/sweep  %.3f us/sweep (wall)\n", name, blocks, c / sweeps, ms * 1e3 / sweeps);
}

int main() {
    std::vector<Te256> h(8 * 512);
    for (size_t i = 0; i < h.size(); ++i) for (int l = 0; l < 8; ++l) { h[i].t.l[l] = (uint32_t)(i * 40503u + l * 977u); h[i].u.l[l] = (uint32_t)(i * 69069u + l); }
    for (auto& t : h) { t.t.l[7] &= 0x7fffffffu; t.u.l[7] &= 0x7fffffffu; }
    Te256 *ta, *tb; Fe256* out; unsigned long long* cyc;
    (void)hipMalloc(&ta, h.size() * sizeof(Te256)); (void)hipMalloc(&tb, h.size() * sizeof(Te256)); (void)hipMalloc(&out, 1024 * 1024 * sizeof(Fe256)); (void)hipMalloc(&cyc, 1024 * 8);
    (void)hipMemcpy(ta, h.data(), h.size() * sizeof(Te256), hipMemcpyHostToDevice); (void)hipMemcpy(tb, h.data(), h.size() * sizeof(Te256), hipMemcpyHostToDevice);
    for (int blocks : {64, 256, 512}) {
        run<0>("0 full sweep", ta, tb, out, cyc, blocks);
        run<1>("1 tables in registers", ta, tb, out, cyc, blocks);
        run<2>("2 no barrier", ta, tb, out, cyc, blocks);
        run<3>("3 one multiply", ta, tb, out, cyc, blocks);
        run<4>("4 no LDS traffic", ta, tb, out, cyc, blocks);
        run<5>("5 tables prefetched one sweep ahead", ta, tb, out, cyc, blocks);
    }
    return 0;
}
