// What does ONE chained pass cost when the kernel boundary is replaced by per-tile dependency flags?  
//
// The latency regime is ~100 dependent launches of 128-256 workgroups, each a 256-element tile (8 KiB) that is
// loaded, worked on for a few dependent sweeps and stored; consecutive passes alternate between a contiguous ("row") and a
// strided ("column": 64 rows x 4 elements) footprint, so every tile of a pass reads from 64 tiles of the pass before it.
// This benchmark runs exactly that data flow — G tiles of 256 x 32 B, P passes, alternating footprints, K dependent multiply-adds of
// filler work per pass — in four forms and checks the result of every element (a counter word that each pass increments: a stale
// read shows up as a wrong count):
//   launch     one kernel launch per pass (what the library does today)
//   flags-wt   ONE persistent launch; a tile's stores are write-through (sc1), every storing wave drains vmcnt, lane 0 publishes the
//              tile's flag (relaxed agent store of the pass number); a consumer's first wave polls the 64 flags it depends on (one per
//              lane, relaxed agent loads), then the workgroup reads its tile with sc1 loads (L2-served: no acquire fence)
//   flags-rel  the same with plain stores + agent release fence before the flag and an agent acquire fence + plain loads after the poll
//   counter    ONE persistent launch; one arrival counter per pass (atomicAdd), every tile polls it for G: the "grid barrier" form
//   flags-spin flags-wt polling without s_sleep and with one flag PER STORING WAVE (no workgroup barrier on the publish path; the
//              consumer's four waves poll 64 flags each)
//   granule    the floor of any in-kernel hand-off: the data IS the flag — every 32-bit limb travels as an 8-byte {tag = pass, limb}
//              granule (64 B per element instead of 32, sc1 stores, no drain, no flag); the consumer re-reads its own eight granules
//              until every tag matches (guide G16 form R2)
// build: hipcc --offload-arch=gfx950 -O3 -o flag_wait flag_wait.hip        run: ./flag_wait [G=256] [P=200]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned int gu32;

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ u32x4 ld_sc1(const u32x4* p) {
    u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ void st_sc1(u32x4* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// position of element `tid` of tile `w` in pass `p` (even passes: contiguous tiles; odd passes: 64 rows x 4 columns, row stride n/64)
__device__ __forceinline__ unsigned pos_of(unsigned p, unsigned w, unsigned tid, unsigned n) {
    return (p & 1) ? (tid >> 2) * (n >> 6) + (w << 2) + (tid & 3) : (w << 8) + tid;
}
// the 64 tiles of pass p-1 that tile w of pass p reads: lane l polls dep(l)
__device__ __forceinline__ unsigned dep_of(unsigned p, unsigned w, unsigned lane, unsigned G) {
    // odd pass (column tile w) reads rows r = 0..63 at r*(n/64) + 4w + c: row tile (r*(n/64) + 4w) / 256 = r*(G/64) + (4w >> 8)
    // even pass (row tile w) reads [256w, 256w + 256): row r = 256w / (n/64) = 64w / G (G >= 64), columns (256w mod n/64)/4 .. +64
    return (p & 1) ? lane * (G >> 6) + (w >> 6) : ((w << 6) % G) + lane;
}

__device__ __forceinline__ void work(u32x4& a, u32x4& b, int K) {
    unsigned long long acc = ((unsigned long long)a.y << 32) | a.z;
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int j = 0; j < 32; ++j) acc = (unsigned long long)(unsigned)acc * (b.x | 1u) + (acc >> 32) + b.y;      // dependent v_mad_u64_u32 chain
    }
    a.y = (unsigned)(acc >> 32); a.z = (unsigned)acc;
    a.x += 1;                                                                // the counter word the check reads
}

__global__ __launch_bounds__(256) void k_pass(const u32x4* src, u32x4* dst, unsigned p, unsigned n, int K) {
    const unsigned pos = pos_of(p, blockIdx.x, threadIdx.x, n);
    u32x4 a = src[2 * pos], b = src[2 * pos + 1];
    work(a, b, K);
    dst[2 * pos] = a; dst[2 * pos + 1] = b;
}

// MODE 1: flags-wt, 2: flags-rel, 3: counter
template <int MODE>
__global__ __launch_bounds__(256) void k_chain(u32x4* buf, gu32* flags, unsigned P, unsigned n, int K, unsigned epoch0, unsigned* tmo) {
    const unsigned w = blockIdx.x, tid = threadIdx.x, G = gridDim.x;
    for (unsigned p = 0; p < P; ++p) {
        const unsigned ep = epoch0 + p;                                       // value a finished tile of pass p publishes
        if (p > 0) {
            if (tid < 64) {
                if (MODE == 3) {
                    gu32* c = flags + (p - 1);
                    unsigned spins = 0;
                    while (__hip_atomic_load(c, RLX_AGENT) != G) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 22)) { *tmo = 1; break; } }
                } else {
                    gu32* f = flags + (size_t)(p - 1) * G + dep_of(p, w, tid, G);
                    unsigned spins = 0;
                    for (;;) {
                        const bool ok = __hip_atomic_load(f, RLX_AGENT) == ep - 1;
                        if (__all(ok)) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1u << 22)) { *tmo = 1; break; }
                    }
                }
                if (MODE != 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        }
        const unsigned pos = pos_of(p, w, tid, n);
        u32x4 a, b;
        if (MODE == 1) { a = ld_sc1(buf + 2 * pos); b = ld_sc1(buf + 2 * pos + 1); drain(); }
        else { a = buf[2 * pos]; b = buf[2 * pos + 1]; }
        work(a, b, K);
        if (MODE == 1) { st_sc1(buf + 2 * pos, a); st_sc1(buf + 2 * pos + 1, b); drain(); }
        else { buf[2 * pos] = a; buf[2 * pos + 1] = b; drain(); }
        __syncthreads();
        if (tid == 0) {
            if (MODE != 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); drain(); }
            if (MODE == 3) __hip_atomic_fetch_add(flags + p, 1u, RLX_AGENT);
            else __hip_atomic_store(flags + (size_t)p * G + w, ep, RLX_AGENT);
        }
    }
}

// flags-spin: one flag per (pass, tile, wave)
__global__ __launch_bounds__(256) void k_chain_spin(u32x4* buf, gu32* flags, unsigned P, unsigned n, int K, unsigned epoch0, unsigned* tmo) {
    const unsigned w = blockIdx.x, tid = threadIdx.x, G = gridDim.x, lane = tid & 63, wv = tid >> 6;
    for (unsigned p = 0; p < P; ++p) {
        const unsigned ep = epoch0 + p;
        if (p > 0) {
            // wave wv of the consumer reads the elements tid = 64 wv + lane; every wave polls the 4 wave flags of 16 producer tiles
            gu32* f = flags + ((size_t)(p - 1) * G + dep_of(p, w, (lane >> 2) + 16 * wv, G)) * 4 + (lane & 3);
            unsigned spins = 0;
            for (;;) {
                const bool ok = __hip_atomic_load(f, RLX_AGENT) == ep - 1;
                if (__all(ok)) break;
                if (++spins > (1u << 24)) { *tmo = 1; break; }
            }
        }
        const unsigned pos = pos_of(p, w, tid, n);
        u32x4 a = ld_sc1(buf + 2 * pos), b = ld_sc1(buf + 2 * pos + 1); drain();
        work(a, b, K);
        st_sc1(buf + 2 * pos, a); st_sc1(buf + 2 * pos + 1, b); drain();
        if (lane == 0) __hip_atomic_store(flags + ((size_t)p * G + w) * 4 + wv, ep, RLX_AGENT);
    }
}

// granule: element = 8 granules {tag, limb}; the two buffers alternate so that a tile never overwrites granules somebody may still poll
typedef __attribute__((address_space(1))) unsigned long long gu64;
__global__ __launch_bounds__(256) void k_chain_granule(unsigned long long* g0, unsigned long long* g1, unsigned P, unsigned n, int K, unsigned epoch0, unsigned* tmo) {
    const unsigned w = blockIdx.x, tid = threadIdx.x;
    for (unsigned p = 0; p < P; ++p) {
        const unsigned ep = epoch0 + p;
        const unsigned pos = pos_of(p, w, tid, n);
        gu64* src = (gu64*)((p & 1) ? g1 : g0) + (size_t)pos * 8;
        gu64* dst = (gu64*)((p & 1) ? g0 : g1) + (size_t)pos * 8;
        unsigned v[8];
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 8; ++k) { const unsigned long long x = __hip_atomic_load(src + k, RLX_AGENT); v[k] = (unsigned)x; ok &= (unsigned)(x >> 32) == ep - 1; }
            if (ok) break;                                                  // per lane: a lane works on as soon as ITS element is there
            if (++spins > (1u << 24)) { *tmo = 1; break; }
        }
        u32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        work(a, b, K);
        const unsigned o[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) __hip_atomic_store(dst + k, ((unsigned long long)ep << 32) | o[k], RLX_AGENT);
    }
}
__global__ void k_granule_init(unsigned long long* g, unsigned n8, unsigned tag) { const unsigned i = blockIdx.x * 256 + threadIdx.x; if (i < n8) g[i] = (unsigned long long)tag << 32; }
__global__ void k_granule_read(const unsigned long long* g, unsigned* out, unsigned n) { const unsigned i = blockIdx.x * 256 + threadIdx.x; if (i < n) out[8 * i] = (unsigned)g[8 * (size_t)i]; }

int main(int argc, char** argv) {
    const unsigned G = argc > 1 ? (unsigned)atoi(argv[1]) : 256u, P = argc > 2 ? (unsigned)atoi(argv[2]) : 200u;   // P <= 1023
    const unsigned n = G * 256u;
    if (G < 64 || (G & (G - 1))) { fprintf(stderr, "G must be a power of two >= 64\n"); return 2; }
    u32x4 *a, *b; gu32* flags; unsigned* tmo;
    (void)hipMalloc(&a, (size_t)n * 32); (void)hipMalloc(&b, (size_t)n * 32);
    (void)hipMalloc((void**)&flags, (size_t)P * G * 16);
    unsigned long long *g0, *g1; (void)hipMalloc(&g0, (size_t)n * 64); (void)hipMalloc(&g1, (size_t)n * 64); (void)hipMalloc(&tmo, 4);
    std::vector<unsigned> h((size_t)n * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int REP = 11;
    printf("G = %u tiles of 256 x 32 B, P = %u passes, alternating row / column (64 x 4) footprints; us per pass, median of %d chains\n", G, P, REP);
    printf("%-6s %10s %10s %10s %10s %10s %10s\n", "K", "launch", "flags-wt", "flags-rel", "counter", "flags-spin", "granule");
    for (int K : {0, 2, 8, 32}) {
        double res[6];
        for (int mode = 0; mode < 6; ++mode) {
            std::vector<float> t;
            bool bad = false;
            for (int rep = 0; rep < REP + 2; ++rep) {
                (void)hipMemset(a, 0, (size_t)n * 32);
                if (mode == 3) (void)hipMemset((void*)flags, 0, (size_t)P * 4);
                (void)hipMemset(tmo, 0, 4);
                static unsigned chain = 0;
                if (mode == 5) k_granule_init<<<(n * 8 + 255) / 256, 256>>>(g0, n * 8, (chain + 1) * 1024u);      // tag = epoch0 - 1: "pass -1" of this chain
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(e0); const unsigned epoch0 = (++chain) * 1024u + 1u;             // flags need no reset: every chain of the run has its own epochs
                if (mode == 0) for (unsigned p = 0; p < P; ++p) k_pass<<<G, 256>>>(a, a, p, n, K);  // in place: a tile reads and writes its own footprint
                else if (mode == 1) k_chain<1><<<G, 256>>>(a, flags, P, n, K, epoch0, tmo);
                else if (mode == 2) k_chain<2><<<G, 256>>>(a, flags, P, n, K, epoch0, tmo);
                else if (mode == 3) k_chain<3><<<G, 256>>>(a, flags, P, n, K, epoch0, tmo);
                else if (mode == 4) k_chain_spin<<<G, 256>>>(a, flags, P, n, K, epoch0, tmo);
                else k_chain_granule<<<G, 256>>>(g0, g1, P, n, K, epoch0, tmo);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2) t.push_back(ms);
                if (mode == 5) k_granule_read<<<(n + 255) / 256, 256>>>((P & 1) ? g1 : g0, (unsigned*)a, n);
                (void)hipMemcpy(h.data(), a, (size_t)n * 32, hipMemcpyDeviceToHost);
                unsigned hto; (void)hipMemcpy(&hto, tmo, 4, hipMemcpyDeviceToHost);
                size_t wrong = 0;
                for (size_t i = 0; i < n; ++i) wrong += h[8 * i] != P;
                if (wrong || hto) { bad = true; fprintf(stderr, "mode %d K %d rep %d: %zu of %u elements WRONG, timeout %u\n", mode, K, rep, wrong, n, hto); }
            }
            std::sort(t.begin(), t.end());
            res[mode] = bad ? -1.0 : t[t.size() / 2] * 1e3 / P;
        }
        printf("%-6d %10.2f %10.2f %10.2f %10.2f %10.2f %10.2f\n", K, res[0], res[1], res[2], res[3], res[4], res[5]);
    }
    (void)b;
    return 0;
}
