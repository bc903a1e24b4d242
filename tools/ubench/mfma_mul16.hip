// Round 4: the 16-point map of mfma_blk16.h in the SMALL-LAUNCH regime — 256-element tiles (16 blocks) on
// v_mfma_i32_16x16x64_i8 (Blk16::phase256) — checked bit-exactly against host field arithmetic and timed next to the seven
// pair-split VALU sweeps it would replace on such a tile (stage_sweep's latency form: 256 threads, one multiply per thread
// per sweep, two barriers).  The constant tables are the ones the 32 x 32 x 32 form already uses.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_mul16 mfma_mul16.hip   (-DSTAMPS: per-segment cycle stamps; -DDA=n: K-steps of
// constant matrices in flight)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../ecfft_amd/csrc/field_secp256k1.h"
using namespace ecfft;
using F = Secp256k1;

#ifdef STAMPS
__device__ unsigned long long g_stamp[8][8];       // [wave][point]
#define BLK16_STAMP(k) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_stamp[threadIdx.x >> 6][k] = __builtin_amdgcn_s_memtime(); }
#endif
#include "../../ecfft_amd/csrc/mfma_blk16.h"
constexpr int NB = 16;
constexpr int TILE = 256;
constexpr int BLK = 256;
#ifndef DA
#define DA 2
#endif
#ifndef MINW
#define MINW 1
#endif

__global__ __launch_bounds__(BLK, MINW) void k_block16_256(const Fe256* __restrict__ in, Fe256* __restrict__ out, const uint8_t* __restrict__ Amat,
                                                            const unsigned long long* __restrict__ Kc, int reps) {
    __shared__ Fe256 tile[TILE];
    const uint32_t tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * TILE;
    tile[tid] = in[base + tid];
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#ifdef STAMPS
        BLK16_STAMP(0)
#endif
        Blk16::phase_n16<4, false, DA>(tile, Amat, Kc, tid, [&] { Blk16::to_operand_form<BLK>(tile, TILE, tid); });
        Blk16::from_swizzled<BLK>(tile, TILE, tid);
#ifdef STAMPS
        BLK16_STAMP(5)
#endif
    }
    out[base + tid] = tile[tid];
}

// what the phase replaces on a 256-element tile: 7 PAIR-SPLIT sweeps (threads [0,128): low output of each pair, [128,256): high
// one; one table multiply each; a barrier between the reads and the in-place writes and one after them)
__global__ __launch_bounds__(BLK, MINW) void k_valu7_256(const Fe256* __restrict__ in, Fe256* __restrict__ out, const Te256* __restrict__ tab, int reps) {
    __shared__ Fe256 tile[TILE];
    const uint32_t tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * TILE;
    tile[tid] = in[base + tid];
    __syncthreads();
    const uint32_t npairs = TILE / 2;
    const bool hi = tid >= npairs; const uint32_t g = hi ? tid - npairs : tid;
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#pragma unroll 1
        for (int s = 0; s < 7; ++s) {
            const uint32_t lh = s < 4 ? 3 - s : s - 3, hh = 1u << lh;
            const uint32_t i = g & (hh - 1), idx = ((g >> lh) << (lh + 1)) + i;
            const Fe256 x = tile[idx], y = tile[idx + hh];
            const Te256 t = tab[(hi ? 32 : 0) + 2 * hh + i];
            __syncthreads();
            if (s < 4) { const Fe256 d = F::sub(y, x); if (hi) tile[idx + hh] = F::tmul(t, d); else tile[idx] = F::tmul_add(t, d, x); }
            else tile[idx + (hi ? hh : 0)] = F::tmul_add(t, y, x);
            __syncthreads();
        }
    }
    out[base + tid] = tile[tid];
}

// lane semantics of the two swaps (printed once: the transpose in phase256 depends on them)
__global__ void k_swap_probe(unsigned* out) {
    const unsigned l = threadIdx.x;
    auto a = __builtin_amdgcn_permlane32_swap(l, 100u + l, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
    out[l] = a[0]; out[64 + l] = a[1]; out[128 + l] = b[0]; out[192 + l] = b[1];
}

// ---------------------------------------------------------------- host side -------------------------------------
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 16); }
static Fe256 rnd_elem() {
    Fe256 r; for (int l = 0; l < 8; ++l) r.l[l] = rnd();
    const uint32_t zero[8] = {r.l[0], r.l[1], r.l[2], r.l[3], r.l[4], r.l[5], r.l[6], r.l[7]};
    return F::finish(zero, 0);
}
static Fe256 pow2(int k) { Fe256 r = F::one(); for (int i = 0; i < k; ++i) r = F::add(r, r); return r; }
static void signed_digits(const Fe256& c, int8_t d[32]) {
    uint8_t u[32]; memcpy(u, c.l, 32);
    bool big = false;
    for (int j = 31; j >= 0; --j) { if (u[j] != 0x7f) { big = u[j] > 0x7f; break; } }
    uint32_t w[8]; memcpy(w, c.l, 32);
    if (big) { uint64_t cy = 977; for (int i = 0; i < 8; ++i) { cy += (uint64_t)w[i] + (i == 1 ? 1u : 0u); w[i] = (uint32_t)cy; cy >>= 32; } }
    uint64_t cy = 0; for (int i = 0; i < 8; ++i) { cy += (uint64_t)w[i] + 0x80808080u; w[i] = (uint32_t)cy; cy >>= 32; }
    memcpy(u, w, 32);
    for (int j = 0; j < 32; ++j) d[j] = (int8_t)(u[j] ^ 0x80);
}

int main(int argc, char** argv) {
    const int tiles = argc > 1 ? atoi(argv[1]) : 1024, reps = argc > 2 ? atoi(argv[2]) : 50;
    {
        unsigned* d; (void)hipMalloc(&d, 256 * 4); k_swap_probe<<<1, 64>>>(d); unsigned h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("permlane32_swap(l, 100+l): r0 lanes 0,16,32,48 = %u %u %u %u   r1 = %u %u %u %u\n", h[0], h[16], h[32], h[48], h[64], h[80], h[96], h[112]);
        printf("permlane16_swap(l, 100+l): r0 lanes 0,16,32,48 = %u %u %u %u   r1 = %u %u %u %u\n", h[128], h[144], h[160], h[176], h[192], h[208], h[224], h[240]);
        (void)hipFree(d);
    }
    std::vector<Fe256> T(NB * NB);
    for (auto& t : T) t = rnd_elem();
    T[0] = F::zero(); T[1] = F::one(); T[2] = F::neg(F::one());
    for (int l = 0; l < 8; ++l) T[3].l[l] = 0x7f7f7f7f;
    T[4] = T[3]; T[4].l[0] += 1;
    std::vector<uint8_t> Amat((size_t)NB * NB * 1024);
    std::vector<unsigned long long> Kc(NB * 8);
    Fe256 off = F::zero();
    for (int g = 0; g < 8; ++g) off = F::add(off, pow2(50 + 32 * g));
    const Fe256 f256 = F::from_u32(256), f128 = F::from_u32(128);
    for (int o = 0; o < NB; ++o) {
        Fe256 sum = F::zero();
        for (int i = 0; i < NB; ++i) {
            Fe256 c = T[o * NB + i];
            int8_t dig[32][32];
            for (int j = 0; j < 32; ++j) { signed_digits(c, dig[j]); sum = F::add(sum, c); c = F::mul(c, f256); }
            uint8_t* A = &Amat[((size_t)o * NB + i) * 1024];
            for (int lane = 0; lane < 64; ++lane) {
                const int m = lane & 31, hh = lane >> 5;
                const int b = 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3);
                for (int q = 0; q < 16; ++q) A[lane * 16 + q] = (uint8_t)dig[16 * hh + q][b];
            }
        }
        const Fe256 kap = F::sub(F::mul(sum, f128), off);
        for (int g = 0; g < 8; ++g) Kc[o * 8 + g] = (1ull << 50) + kap.l[g];
    }
    const size_t n = (size_t)tiles * TILE;
    std::vector<Fe256> hin(n), hout(n);
    for (auto& x : hin) x = rnd_elem();
    for (int l = 0; l < 8; ++l) { hin[0].l[l] = 0; hin[1].l[l] = 0xFFFFFFFFu; }
    hin[1].l[0] = 0xFFFFFC2Eu; hin[1].l[1] = 0xFFFFFFFEu;                   // p - 1
    Fe256 *din, *dout; uint8_t* dA; unsigned long long* dK;
    (void)hipMalloc(&din, n * 32); (void)hipMalloc(&dout, n * 32); (void)hipMalloc(&dA, Amat.size()); (void)hipMalloc(&dK, Kc.size() * 8);
    (void)hipMemcpy(din, hin.data(), n * 32, hipMemcpyHostToDevice);
    (void)hipMemcpy(dA, Amat.data(), Amat.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(dK, Kc.data(), Kc.size() * 8, hipMemcpyHostToDevice);

    for (int check_reps = 1; check_reps <= 2; ++check_reps) {
        k_block16_256<<<tiles, BLK>>>(din, dout, dA, dK, check_reps);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(e)); return 1; }
        (void)hipMemcpy(hout.data(), dout, n * 32, hipMemcpyDeviceToHost);
        size_t bad = 0, checked = 0;
        for (size_t blk = 0; blk < n / NB; blk += (blk < 256 ? 1 : 97)) {
            Fe256 x[NB], y[NB];
            for (int i = 0; i < NB; ++i) x[i] = hin[blk * NB + i];
            for (int r = 0; r < check_reps; ++r) {
                for (int o = 0; o < NB; ++o) { Fe256 a = F::zero(); for (int i = 0; i < NB; ++i) a = F::add(a, F::mul(T[o * NB + i], x[i])); y[o] = a; }
                for (int i = 0; i < NB; ++i) x[i] = y[i];
            }
            for (int o = 0; o < NB; ++o) { ++checked; if (!F::eq(x[o], hout[blk * NB + o])) { if (bad < 6) printf("  mismatch blk %zu out %d: got %08x..%08x want %08x..%08x\n", blk, o, hout[blk * NB + o].l[7], hout[blk * NB + o].l[0], x[o].l[7], x[o].l[0]); ++bad; } }
        }
        printf("MFMA 16x16x64 block-16 map on 256-element tiles, %d application(s): %zu / %zu outputs %s\n", check_reps, checked - bad, checked, bad ? "MISMATCH" : "bit-exact vs host");
    }
    {   // the other small-launch forms (test-hook kernels of mfma_blk16.h), one application each, same host check
        auto check = [&](const char* what) {
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) { printf("%s: kernel failed: %s\n", what, hipGetErrorString(e)); return; }
            (void)hipMemcpy(hout.data(), dout, n * 32, hipMemcpyDeviceToHost);
            size_t bad = 0, checked = 0;
            for (size_t blk = 0; blk < n / NB; blk += (blk < 256 ? 1 : 97)) {
                for (int o = 0; o < NB; ++o) {
                    Fe256 a = F::zero(); for (int i = 0; i < NB; ++i) a = F::add(a, F::mul(T[o * NB + i], hin[blk * NB + i]));
                    ++checked; if (!F::eq(a, hout[blk * NB + o])) { if (bad < 4) printf("  %s mismatch blk %zu out %d\n", what, blk, o); ++bad; }
                }
            }
            printf("%s: %zu / %zu outputs %s\n", what, checked - bad, checked, bad ? "MISMATCH" : "bit-exact vs host");
        };
        (void)hipMemcpy(dout, din, n * 32, hipMemcpyDeviceToDevice); k_blk16_apply_n16<1><<<n / 256, 256>>>(dout, dA, dK); check("mode 1 (256 elements, 4 waves, LDS-resident)");
        (void)hipMemcpy(dout, din, n * 32, hipMemcpyDeviceToDevice); k_blk16_apply_n16<2><<<n / 256, 128>>>(dout, dA, dK); check("mode 2 (256 elements, 2 waves, two results per lane)");
        (void)hipMemcpy(dout, din, n * 32, hipMemcpyDeviceToDevice); k_blk16_apply_n16<3><<<n / 128, 128>>>(dout, dA, dK); check("mode 3 (128 elements = 8 blocks, 2 waves, registers)");
        (void)hipMemcpy(dout, din, n * 32, hipMemcpyDeviceToDevice); k_blk16_apply_n16<4><<<n / 256, 256>>>(dout, dA, dK); check("mode 4 (256 elements, 4 waves, registers)");
    }
#ifdef STAMPS
    for (int tl : {1, 256}) {
        k_block16_256<<<tl, BLK>>>(din, dout, dA, dK, 3); (void)hipDeviceSynchronize();
        unsigned long long st[8][8]; (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
        printf("stamps (shader cycles, block 0, last of 3 reps; grid %d): start -> operand form + requests -> MFMA loop -> swaps+normalise -> barrier -> write+barrier+un-swizzle\n", tl);
        for (int w = 0; w < 4; ++w) printf("  wave %d: conv %6llu  mfma %6llu  norm %6llu  bar %6llu  write %6llu   total %6llu\n", w, st[w][1] - st[w][0], st[w][2] - st[w][1], st[w][3] - st[w][2], st[w][4] - st[w][3], st[w][5] - st[w][4], st[w][5] - st[w][0]);
    }
#endif
    std::vector<Te256> htab(64);
    for (auto& t : htab) { t.t = rnd_elem(); t.u = F::mul(t.t, pow2(128)); }
    Te256* dtab; (void)hipMalloc(&dtab, htab.size() * sizeof(Te256)); (void)hipMemcpy(dtab, htab.data(), htab.size() * sizeof(Te256), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time_it = [&](auto launch) { float best = 1e30f; for (int r = 0; r < 4; ++r) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; } return best; };
    for (int tl : {128, 256, 512, 1024}) {
        if ((size_t)tl * TILE > n) continue;
        const float m1 = time_it([&] { k_block16_256<<<tl, BLK>>>(din, dout, dA, dK, reps); });
        const float m0 = time_it([&] { k_block16_256<<<tl, BLK>>>(din, dout, dA, dK, 0); });
        const float v1 = time_it([&] { k_valu7_256<<<tl, BLK>>>(din, dout, dtab, reps); });
        const float v0 = time_it([&] { k_valu7_256<<<tl, BLK>>>(din, dout, dtab, 0); });
        const double us_m = (m1 - m0) * 1e3 / reps, us_v = (v1 - v0) * 1e3 / reps;
        printf("tiles %4d (DA=%d): MFMA phase (operand form + 16x16x64 phase + un-swizzle) %7.2f us per pass, 7 pair-split VALU sweeps %7.2f us  -> ratio %.2fx\n", tl, DA, us_m, us_v, us_v / us_m);
    }
    return 0;
}
