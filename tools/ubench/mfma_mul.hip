// Round 3, candidate (b) of the "cheaper 256-bit table multiply" question: the secp256k1 constant multiply on the int8 matrix
// cores (v_mfma_i32_32x32x32_i8), applied where the butterfly constants are shared by >= 32 data elements.
//
// What is computed.  The innermost NB = 16 points of every block of an EXTEND see the decompose stages h = 8,4,2,1 and the
// recombine stages h = 1,2,4,8 back to back: one LINEAR map out_o = sum_i T[o][i] * x_i (mod p) with a 16 x 16 matrix of
// field constants that is the same for every 16-block of the level (it depends on the tree and the parity only).  Each
// constant c = T[o][i] becomes a 32 x 32 int8 matrix
//        C[b][j] = digit b (radix 256, signed, in [-128,127]) of  c * 2^(8j) mod p,
// so that with x = sum_j x_j 2^(8j) (bytes)     c * x  ==  sum_b 2^(8b) * sum_j C[b][j] x_j     (mod p).
// One MFMA handles the same constant for 32 different blocks (N = 32 columns = 32 data elements, K = 32 data bytes,
// M = 32 result digits); the 16 inputs of a block are 16 MFMAs accumulating into the same 32 x 32 int32 tile.
// Data bytes are unsigned: x'_j = x_j - 128 (xor 0x80) goes into the MFMA and the constant 128 * sum C[b][j] is part of
// the per-output additive constant K.  What is left for the VALU is ONE carry normalisation per output element
// (32 column sums of < 2^24 -> eight 32-bit words + fold of the 19-bit top) instead of 7 modular multiplies.
//
// This file: (1) checks the scheme bit-exactly against host field arithmetic, (2) measures the cost of the MFMA phase
// of a 1024-element LDS tile next to the same tile's 7 VALU sweeps it would replace.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_mul mfma_mul.hip
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
constexpr int NB = 16;            // points of the composite map
constexpr int TILE = 1024;        // elements per workgroup tile (32 KiB of LDS)
constexpr int BLK = 512;
#ifndef MINW
#define MINW 4
#endif

__global__ __launch_bounds__(BLK, MINW) void k_block16(const Fe256* __restrict__ in, Fe256* __restrict__ out, const uint8_t* __restrict__ Amat,
                                                    const unsigned long long* __restrict__ Kc, int reps) {
    __shared__ Fe256 tile[TILE];
    const uint32_t tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * TILE;
    for (uint32_t j = tid; j < TILE; j += BLK) tile[j] = in[base + j];
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#ifdef STAMPS
        BLK16_STAMP(0)
#endif
        Blk16::APre pre = Blk16::prefetch(Amat, tid); __builtin_amdgcn_sched_barrier(0); Blk16::to_operand_form<BLK>(tile, TILE, tid); Blk16::phase(tile, Amat, Kc, tid, pre); Blk16::from_swizzled<BLK>(tile, TILE, tid);
#ifdef STAMPS
        BLK16_STAMP(5)
#endif
    }
    for (uint32_t j = tid; j < TILE; j += BLK) out[base + j] = tile[j];
}

// what the phase replaces: 7 LDS sweeps of the same tile with the shipped table multiply (2 multiplies per pair per sweep;
// constants from a 16-entry table as the innermost stages read them)
__global__ __launch_bounds__(BLK, 4) void k_valu7(const Fe256* __restrict__ in, Fe256* __restrict__ out, const Te256* __restrict__ tab, int reps) {
    __shared__ Fe256 tile[TILE];
    const uint32_t tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * TILE;
    for (uint32_t j = tid; j < TILE; j += BLK) tile[j] = in[base + j];
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
#pragma unroll 1
        for (int s = 0; s < 7; ++s) {
            const uint32_t lh = s < 4 ? 3 - s : s - 3, hh = 1u << lh;
            const uint32_t g = tid, i = g & (hh - 1), idx = ((g >> lh) << (lh + 1)) + i;
            Fe256 a = tile[idx], b = tile[idx + hh];
            const Te256 t0 = tab[2 * hh + i], t1 = tab[32 + 2 * hh + i];
            if (s < 4) { Fe256 q1 = F::tmul(t1, F::sub(b, a)); tile[idx] = F::tmul_add(t0, q1, a); tile[idx + hh] = q1; }
            else { tile[idx] = F::tmul_add(t0, b, a); tile[idx + hh] = F::tmul_add(t1, b, a); }
            __syncthreads();
        }
    }
    for (uint32_t j = tid; j < TILE; j += BLK) out[base + j] = tile[j];
}

// ---------------------------------------------------------------- host side -------------------------------------
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 16); }
static Fe256 rnd_elem() {
    Fe256 r; for (int l = 0; l < 8; ++l) r.l[l] = rnd();
    const uint32_t zero[8] = {r.l[0], r.l[1], r.l[2], r.l[3], r.l[4], r.l[5], r.l[6], r.l[7]};
    return F::finish(zero, 0);                       // canonical
}
static Fe256 pow2(int k) { Fe256 r = F::one(); for (int i = 0; i < k; ++i) r = F::add(r, r); return r; }

// signed radix-256 digits of a canonical c: value c or c - p, whichever lies in [-0x8080..80, 0x7f7f..7f]
static void signed_digits(const Fe256& c, int8_t d[32]) {
    uint8_t u[32]; memcpy(u, c.l, 32);
    bool big = false;                                // c > 0x7f7f...7f ?
    for (int j = 31; j >= 0; --j) { if (u[j] != 0x7f) { big = u[j] > 0x7f; break; } }
    uint32_t w[8]; memcpy(w, c.l, 32);
    if (big) {                                       // two's complement pattern of c - p = c + 2^32 + 977 - 2^256
        uint64_t cy = 977; for (int i = 0; i < 8; ++i) { cy += (uint64_t)w[i] + (i == 1 ? 1u : 0u); w[i] = (uint32_t)cy; cy >>= 32; }
    }
    uint64_t cy = 0; for (int i = 0; i < 8; ++i) { cy += (uint64_t)w[i] + 0x80808080u; w[i] = (uint32_t)cy; cy >>= 32; }
    memcpy(u, w, 32);
    for (int j = 0; j < 32; ++j) d[j] = (int8_t)(u[j] ^ 0x80);
}

int main(int argc, char** argv) {
    const int tiles = argc > 1 ? atoi(argv[1]) : 512, reps = argc > 2 ? atoi(argv[2]) : 50;
    // the composite matrix (random constants stand in for a tree's: the arithmetic does not care)
    std::vector<Fe256> T(NB * NB);
    for (auto& t : T) t = rnd_elem();
    T[0] = F::zero(); T[1] = F::one(); T[2] = F::neg(F::one());           // edge constants
    for (int l = 0; l < 8; ++l) T[3].l[l] = 0x7f7f7f7f;                    // the recoding threshold itself
    T[4] = T[3]; T[4].l[0] += 1;
    std::vector<uint8_t> Amat((size_t)NB * NB * 1024);
    std::vector<unsigned long long> Kc(NB * 8);
    Fe256 off = F::zero();
    for (int g = 0; g < 8; ++g) off = F::add(off, pow2(50 + 32 * g));
    const Fe256 f256 = F::from_u32(256), f128 = F::from_u32(128);
    int bad_digits = 0;
    for (int o = 0; o < NB; ++o) {
        Fe256 sum = F::zero();
        for (int i = 0; i < NB; ++i) {
            Fe256 c = T[o * NB + i];
            int8_t dig[32][32];
            for (int j = 0; j < 32; ++j) {
                signed_digits(c, dig[j]);
                {   // self-check: digits reassemble to c (mod p)
                    Fe256 acc = F::zero();
                    for (int b = 31; b >= 0; --b) { acc = F::mul(acc, f256); int v = dig[j][b]; acc = v >= 0 ? F::add(acc, F::from_u32((uint32_t)v)) : F::sub(acc, F::from_u32((uint32_t)(-v))); }
                    if (!F::eq(acc, c)) ++bad_digits;
                }
                sum = F::add(sum, c);
                c = F::mul(c, f256);
            }
            uint8_t* A = &Amat[((size_t)o * NB + i) * 1024];
            for (int lane = 0; lane < 64; ++lane) {
                const int m = lane & 31, hh = lane >> 5;
                const int b = 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3);   // == inverse of Blk16::row_of_digit
                for (int q = 0; q < 16; ++q) A[lane * 16 + q] = (uint8_t)dig[16 * hh + q][b];
            }
        }
        const Fe256 kap = F::sub(F::mul(sum, f128), off);
        for (int g = 0; g < 8; ++g) Kc[o * 8 + g] = (1ull << 50) + kap.l[g];
    }
    printf("signed-digit self-check: %s\n", bad_digits ? "MISMATCH" : "ok");

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
        k_block16<<<tiles, BLK>>>(din, dout, dA, dK, check_reps);
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
            for (int o = 0; o < NB; ++o) { ++checked; if (!F::eq(x[o], hout[blk * NB + o])) { if (bad < 4) printf("  mismatch blk %zu out %d: got %08x.. want %08x..\n", blk, o, hout[blk * NB + o].l[7], x[o].l[7]); ++bad; } }
        }
        printf("MFMA block-16 map, %d application(s): %zu / %zu outputs %s\n", check_reps, checked - bad, checked, bad ? "MISMATCH" : "bit-exact vs host");
    }

#ifdef STAMPS
    for (int tl : {1, 256, 512}) {
        k_block16<<<tl, BLK>>>(din, dout, dA, dK, 3); (void)hipDeviceSynchronize();
        unsigned long long st[8][8]; (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
        printf("stamps (shader cycles, block 0, last of 3 reps; grid %d): start -> A req + operand form -> MFMA loop -> swap+normalise -> barrier -> write+barrier\n", tl);
        for (int w = 0; w < 8; ++w) printf("  wave %d: conv %6llu  mfma %6llu  norm %6llu  bar %6llu  write %6llu   total %6llu\n", w, st[w][1] - st[w][0], st[w][2] - st[w][1], st[w][3] - st[w][2], st[w][4] - st[w][3], st[w][5] - st[w][4], st[w][5] - st[w][0]);
    }
#endif
    // timing
    std::vector<Te256> htab(64);
    for (auto& t : htab) { t.t = rnd_elem(); t.u = F::mul(t.t, pow2(128)); }
    Te256* dtab; (void)hipMalloc(&dtab, htab.size() * sizeof(Te256)); (void)hipMemcpy(dtab, htab.data(), htab.size() * sizeof(Te256), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time_it = [&](auto launch) { float best = 1e30f; for (int r = 0; r < 4; ++r) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; } return best; };
    for (int tl : {256, 512, 1024, 2048}) {
        if ((size_t)tl * TILE > n) continue;
        const float m1 = time_it([&] { k_block16<<<tl, BLK>>>(din, dout, dA, dK, reps); });
        const float m0 = time_it([&] { k_block16<<<tl, BLK>>>(din, dout, dA, dK, 0); });
        const float v1 = time_it([&] { k_valu7<<<tl, BLK>>>(din, dout, dtab, reps); });
        const float v0 = time_it([&] { k_valu7<<<tl, BLK>>>(din, dout, dtab, 0); });
        const double phases = (double)tl * reps;
        const double us_m = (m1 - m0) * 1e3 / reps, us_v = (v1 - v0) * 1e3 / reps;
        printf("tiles %4d: MFMA phase %8.2f us per pass over all tiles (%6.3f us per tile-phase per CU-slot), VALU 7 sweeps %8.2f us  -> ratio %.2fx;  "
               "MFMA path %.3e element-maps/s = %.3e replaced field-mul/s (VALU path: %.3e field-mul/s)\n",
               tl, us_m, us_m / (tl / 256.0), us_v, us_v / us_m, phases * TILE / ((m1 - m0) * 1e-3), phases * TILE * 7 / ((m1 - m0) * 1e-3),
               phases * TILE * 7 / ((v1 - v0) * 1e-3));
    }
    return 0;
}
