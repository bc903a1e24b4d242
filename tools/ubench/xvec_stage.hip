// Round 6, DESIGN.md 8.2: what would ONE butterfly stage cost on the matrix cores when the constant is shared ACROSS VECTORS?
//
// Inside one transform only pair distances <= 8 of a tile share a constant 32-fold (mfma_blk16.h).  Across the vectors of a level
// (or the polynomials of a batch) EVERY stage constant is shared: level m of an ENTER has 2n/m vectors on the same tables.  This
// micro-benchmark measures the per-stage cost of that form next to the shipped 169-instruction table multiply on the same work:
//
//   tile = 32 positions x 32 vectors of secp256k1 elements, position-major, held in LDS in the MFMA operand form (bytes ^ 0x80);
//   a recombine stage at pair distance h:  (a, b) = (x[i], x[i + h])  ->  (a + s0 * b, a + s1 * b)   for all 32 vectors at once:
//        acc_o = I * a_bytes + C(s_o) * b_bytes        two v_mfma_i32_32x32x32_i8 per output (N = 32 vectors, K = 32 data bytes,
//                                                      M = 32 result digits), C(s) = the 32 x 32 int8 digit matrix of s (1 KiB),
//        one v_permlane32_swap per accumulator register pairs the two outputs, ONE carry normalisation per output element
//        (Blk16::normalise), result written back in operand form — the tile never leaves the byte form between stages.
//   Every (position pair, stage) has its own two constants; NG position groups of constants are cycled over the tiles (NG small:
//   the matrices are L2 hits, NG large: they stream), as the columns of a level would.
//
// The VALU kernel does the same stage with F::tmul_add on plain elements (2 telems = 128 B per pair and stage).
// Both results are compared bit for bit (and a sample against host arithmetic).
// Build: hipcc --offload-arch=gfx950 -O3 -o xvec_stage xvec_stage.hip      Run: ./xvec_stage [tiles] [stage-reps] [NG]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../ecfft_amd/csrc/field_secp256k1.h"
#include "../../ecfft_amd/csrc/mfma_blk16.h"
using namespace ecfft;
using F = Secp256k1;

constexpr int P = 32, V = 32, NST = 5;             // positions, vectors, stages of one pass (distances 16, 8, 4, 2, 1)
constexpr int BLK = 256;                           // 4 waves
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// constants of (group g, stage s, pair q, output o): index ((g * NST + s) * 16 + q) * 2 + o
__device__ __forceinline__ uint32_t pair_lo(uint32_t q, uint32_t lh) { return ((q >> lh) << (lh + 1)) + (q & ((1u << lh) - 1)); }

#ifndef MINB
#define MINB 2
#endif
__global__ __launch_bounds__(BLK, MINB) void k_xvec(const Fe256* __restrict__ in, Fe256* __restrict__ out, const uint8_t* __restrict__ Cmat,
                                                 const unsigned long long* __restrict__ Kc, int ng, int reps) {
    __shared__ uint4 lds[P * 2 * V];               // [pos][half][vec] 16-byte chunks: a wave's operand read is 1 KiB contiguous
    const uint32_t tid = threadIdx.x, L = tid & 63, w = tid >> 6, n = L & 31, h = L >> 5;
    const size_t base = (size_t)blockIdx.x * P * V;
    const uint32_t X = 0x80808080u;
    for (uint32_t e = tid; e < P * V; e += BLK) {       // global layout [pos][vec]
        const Fe256 x = in[base + e];
        const uint32_t pos = e >> 5, vec = e & 31;
        lds[(pos * 2 + 0) * V + vec] = make_uint4(x.l[0] ^ X, x.l[1] ^ X, x.l[2] ^ X, x.l[3] ^ X);
        lds[(pos * 2 + 1) * V + vec] = make_uint4(x.l[4] ^ X, x.l[5] ^ X, x.l[6] ^ X, x.l[7] ^ X);
    }
    // identity in the A layout: lane (m = L & 31, half h): byte q is 1 where result digit b(m) == 16 h + q
    v4i Id;
    {
        const uint32_t m = L & 31, b = 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3);
        uint32_t wv[4] = {0, 0, 0, 0};
        if ((b >> 4) == h) wv[(b & 15) >> 2] = 1u << (8 * (b & 3));
        Id = (v4i){(int)wv[0], (int)wv[1], (int)wv[2], (int)wv[3]};
    }
    const uint32_t g = blockIdx.x % (uint32_t)ng;
    typedef const __attribute__((address_space(1))) char* gchar;
    typedef const __attribute__((address_space(1))) v4i* gv4;
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        const uint32_t s = (uint32_t)r % NST, lh = 4 - s;
        const size_t cbase = ((size_t)g * NST + s) * 16;
        // a wave's four pairs; the matrices of the next pair are requested while this one runs
        v4i C0 = *(gv4)((gchar)Cmat + ((cbase + 4 * w) * 2 + 0) * 1024 + L * 16), C1 = *(gv4)((gchar)Cmat + ((cbase + 4 * w) * 2 + 1) * 1024 + L * 16);
        Fe256 z[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t q = 4 * w + k, i = pair_lo(q, lh), j = i + (1u << lh);
            v4i nC0 = C0, nC1 = C1;
            if (k + 1 < 4) { nC0 = *(gv4)((gchar)Cmat + ((cbase + q + 1) * 2 + 0) * 1024 + L * 16); nC1 = *(gv4)((gchar)Cmat + ((cbase + q + 1) * 2 + 1) * 1024 + L * 16); }
            const uint4 ua = lds[(i * 2 + h) * V + n], ub = lds[(j * 2 + h) * V + n];
            const v4i Ba = {(int)ua.x, (int)ua.y, (int)ua.z, (int)ua.w}, Bb = {(int)ub.x, (int)ub.y, (int)ub.z, (int)ub.w};
            __builtin_amdgcn_sched_barrier(0);
            v16i acc0 = {0}, acc1 = {0};
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(Id, Ba, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(Id, Ba, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(C0, Bb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(C1, Bb, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            int lo[16], hi[16];
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                auto p = __builtin_amdgcn_permlane32_swap((unsigned)acc0[rr], (unsigned)acc1[rr], false, false);
                lo[rr] = (int)p[0]; hi[rr] = (int)p[1];
            }
            z[k] = Blk16::normalise<true>(lo, hi, Kc + ((cbase + q) * 2 + h) * 8);     // lanes < 32: output 0 (position i), lanes >= 32: output 1 (position j)
            C0 = nC0; C1 = nC1;
        }
        __syncthreads();                               // every operand read of the stage is done
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t q = 4 * w + k, i = pair_lo(q, lh), pos = h ? i + (1u << lh) : i;
            lds[(pos * 2 + 0) * V + n] = make_uint4(z[k].l[0] ^ X, z[k].l[1] ^ X, z[k].l[2] ^ X, z[k].l[3] ^ X);
            lds[(pos * 2 + 1) * V + n] = make_uint4(z[k].l[4] ^ X, z[k].l[5] ^ X, z[k].l[6] ^ X, z[k].l[7] ^ X);
        }
        __syncthreads();
    }
    for (uint32_t e = tid; e < P * V; e += BLK) {
        const uint32_t pos = e >> 5, vec = e & 31;
        const uint4 a = lds[(pos * 2 + 0) * V + vec], b = lds[(pos * 2 + 1) * V + vec];
        Fe256 x; x.l[0] = a.x ^ X; x.l[1] = a.y ^ X; x.l[2] = a.z ^ X; x.l[3] = a.w ^ X; x.l[4] = b.x ^ X; x.l[5] = b.y ^ X; x.l[6] = b.z ^ X; x.l[7] = b.w ^ X;
        out[base + e] = x;
    }
}

// the shipped multiply on the same stage: 512 butterflies per stage and tile, two per thread (vector-minor: a wave = 2 pairs x 32 vectors)
__global__ __launch_bounds__(BLK, 4) void k_valu(const Fe256* __restrict__ in, Fe256* __restrict__ out, const Te256* __restrict__ tab, int ng, int reps) {
    __shared__ Fe256 tile[P * V];
    const uint32_t tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * P * V;
    for (uint32_t e = tid; e < P * V; e += BLK) tile[e] = in[base + e];
    const uint32_t g = blockIdx.x % (uint32_t)ng;
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        const uint32_t s = (uint32_t)r % NST, lh = 4 - s;
        const size_t cbase = ((size_t)g * NST + s) * 16;
#pragma unroll 1
        for (int k = 0; k < 2; ++k) {
            const uint32_t bf = tid + BLK * k, q = bf >> 5, vec = bf & 31, i = pair_lo(q, lh), j = i + (1u << lh);
            const Fe256 a = tile[i * V + vec], b = tile[j * V + vec];
            const Te256 t0 = tab[(cbase + q) * 2 + 0], t1 = tab[(cbase + q) * 2 + 1];
            const Fe256 o0 = F::tmul_add(t0, b, a), o1 = F::tmul_add(t1, b, a);
            tile[i * V + vec] = o0; tile[j * V + vec] = o1;       // a butterfly owns its two elements: no barrier inside the stage
        }
        __syncthreads();
    }
    for (uint32_t e = tid; e < P * V; e += BLK) out[base + e] = tile[e];
}

// ---------------------------------------------------------------- host side -------------------------------------
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 16); }
static Fe256 rnd_elem() {
    Fe256 r; for (int l = 0; l < 8; ++l) r.l[l] = rnd();
    const uint32_t w[8] = {r.l[0], r.l[1], r.l[2], r.l[3], r.l[4], r.l[5], r.l[6], r.l[7]};
    return F::finish(w, 0);
}
static Fe256 pow2(int k) { Fe256 r = F::one(); for (int i = 0; i < k; ++i) r = F::add(r, r); return r; }
static void signed_digits(const Fe256& c, int8_t d[32]) {     // as tools/ubench/mfma_mul.hip
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
    const int tiles = argc > 1 ? atoi(argv[1]) : 2048, reps = argc > 2 ? atoi(argv[2]) : 50, ng = argc > 3 ? atoi(argv[3]) : 64;
    const size_t ncon = (size_t)ng * NST * 16 * 2;
    std::vector<Fe256> cs(ncon);
    for (auto& c : cs) c = rnd_elem();
    cs[0] = F::zero(); cs[1] = F::one(); cs[2] = F::neg(F::one());
    for (int l = 0; l < 8; ++l) cs[3].l[l] = 0x7f7f7f7f;
    std::vector<uint8_t> Cmat(ncon * 1024);
    std::vector<unsigned long long> Kc(ncon * 8);
    std::vector<Te256> tab(ncon);
    Fe256 off = F::zero();
    for (int g = 0; g < 8; ++g) off = F::add(off, pow2(50 + 32 * g));
    const Fe256 f256 = F::from_u32(256), f128 = F::from_u32(128);
    for (size_t x = 0; x < ncon; ++x) {
        Fe256 sum = F::zero();
        for (int inp = 0; inp < 2; ++inp) {                       // inputs: a with constant 1, b with constant cs[x]
            Fe256 c = inp ? cs[x] : F::one();
            int8_t dig[32][32];
            for (int j = 0; j < 32; ++j) { signed_digits(c, dig[j]); sum = F::add(sum, c); c = F::mul(c, f256); }
            if (inp) {
                uint8_t* A = &Cmat[x * 1024];
                for (int lane = 0; lane < 64; ++lane) {
                    const int m = lane & 31, hh = lane >> 5, b = 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3);
                    for (int q = 0; q < 16; ++q) A[lane * 16 + q] = (uint8_t)dig[16 * hh + q][b];
                }
            } else {                                             // the kernel's register identity must BE the digit matrix of 1
                for (int j = 0; j < 32; ++j) for (int b = 0; b < 32; ++b) if (dig[j][b] != (j == b ? 1 : 0)) { printf("identity digits unexpected\n"); return 1; }
            }
        }
        const Fe256 kap = F::sub(F::mul(sum, f128), off);
        for (int g = 0; g < 8; ++g) Kc[x * 8 + g] = (1ull << 50) + kap.l[g];
        tab[x].t = cs[x]; tab[x].u = F::mul(cs[x], pow2(128));
    }
    const size_t n = (size_t)tiles * P * V;
    std::vector<Fe256> hin(n), h1(n), h2(n);
    for (auto& x : hin) x = rnd_elem();
    for (int l = 0; l < 8; ++l) { hin[0].l[l] = 0; hin[1].l[l] = 0xFFFFFFFFu; }
    hin[1].l[0] = 0xFFFFFC2Eu; hin[1].l[1] = 0xFFFFFFFEu;
    Fe256 *din, *dout; uint8_t* dC; unsigned long long* dK; Te256* dtab;
    (void)hipMalloc(&din, n * 32); (void)hipMalloc(&dout, n * 32); (void)hipMalloc(&dC, Cmat.size()); (void)hipMalloc(&dK, Kc.size() * 8); (void)hipMalloc(&dtab, tab.size() * sizeof(Te256));
    (void)hipMemcpy(din, hin.data(), n * 32, hipMemcpyHostToDevice);
    (void)hipMemcpy(dC, Cmat.data(), Cmat.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(dK, Kc.data(), Kc.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dtab, tab.data(), tab.size() * sizeof(Te256), hipMemcpyHostToDevice);
    for (int cr : {1, NST, 2 * NST + 1}) {
        k_xvec<<<tiles, BLK>>>(din, dout, dC, dK, ng, cr);
        hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) { printf("k_xvec failed: %s\n", hipGetErrorString(e)); return 1; }
        (void)hipMemcpy(h1.data(), dout, n * 32, hipMemcpyDeviceToHost);
        k_valu<<<tiles, BLK>>>(din, dout, dtab, ng, cr);
        e = hipDeviceSynchronize(); if (e != hipSuccess) { printf("k_valu failed: %s\n", hipGetErrorString(e)); return 1; }
        (void)hipMemcpy(h2.data(), dout, n * 32, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < n; ++i) if (!F::eq(h1[i], h2[i])) { if (bad < 4) printf("  mismatch elem %zu (tile %zu pos %zu vec %zu)\n", i, i / (P * V), (i / V) % P, i % V); ++bad; }
        // host check of tile 0 and the last tile
        size_t hbad = 0;
        for (size_t t : {(size_t)0, (size_t)tiles - 1}) {
            std::vector<Fe256> x(hin.begin() + t * P * V, hin.begin() + (t + 1) * P * V);
            for (int r = 0; r < cr; ++r) {
                const uint32_t s = r % NST, lh = 4 - s; const size_t cbase = ((size_t)(t % ng) * NST + s) * 16;
                for (uint32_t q = 0; q < 16; ++q) for (uint32_t v = 0; v < V; ++v) {
                    const uint32_t i = ((q >> lh) << (lh + 1)) + (q & ((1u << lh) - 1)), j = i + (1u << lh);
                    const Fe256 a = x[i * V + v], b = x[j * V + v];
                    x[i * V + v] = F::add(a, F::mul(cs[(cbase + q) * 2 + 0], b)); x[j * V + v] = F::add(a, F::mul(cs[(cbase + q) * 2 + 1], b));
                }
            }
            for (size_t i = 0; i < (size_t)P * V; ++i) if (!F::eq(x[i], h1[t * P * V + i])) ++hbad;
        }
        printf("%d stage(s): matrix-core form vs VALU form %zu / %zu elements equal, host check of 2 tiles: %s\n", cr, n - bad, n, hbad ? "MISMATCH" : "bit-exact");
        if (bad || hbad) return 1;
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time_it = [&](auto launch) { float best = 1e30f; for (int r = 0; r < 4; ++r) { (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; } return best; };
    printf("constant groups %d: %.1f MiB of digit matrices / %.2f MiB of telems per %d-stage pass\n", ng, Cmat.size() / 1048576.0, tab.size() * 64 / 1048576.0, NST);
    for (int tl : {512, 1024, 2048, 4096, 8192}) {
        if (tl > tiles) continue;
        const float m1 = time_it([&] { k_xvec<<<tl, BLK>>>(din, dout, dC, dK, ng, reps); }), m0 = time_it([&] { k_xvec<<<tl, BLK>>>(din, dout, dC, dK, ng, 0); });
        const float v1 = time_it([&] { k_valu<<<tl, BLK>>>(din, dout, dtab, ng, reps); }), v0 = time_it([&] { k_valu<<<tl, BLK>>>(din, dout, dtab, ng, 0); });
        const double muls = (double)tl * reps * P * V;        // one multiply per element and stage
        printf("tiles %5d: matrix-core stage %7.3f us per 1024-element tile-stage-slot, VALU stage %7.3f  -> %.2fx;  %.3e vs %.3e field-mul/s\n", tl,
               (m1 - m0) * 1e3 / reps / (tl / 512.0), (v1 - v0) * 1e3 / reps / (tl / 512.0), (v1 - v0) / (m1 - m0), muls / ((m1 - m0) * 1e-3), muls / ((v1 - v0) * 1e-3));
    }
    return 0;
}
