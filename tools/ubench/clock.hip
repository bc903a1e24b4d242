// Effective shader clock and TRUE issue cost (in shader cycles) of the integer VALU instructions the secp256k1 multiply is
// made of, on MI355X.  Reconciles tools/ubench/intops.hip (which prices wall time at the nominal 2.4 GHz) with the
// guide's "2 cycles per plain wave64 VALU instruction": each kernel brackets its instruction loop with s_memtime
// (shader-clock ticks) and wall_clock64() (constant 100 MHz), so
//     effective clock = d(s_memtime) / d(wall_clock64) * 100 MHz,
//     cycles per wave-instruction per SIMD = launch wall time x effective clock / (wave-instructions per SIMD of the launch).
// Build: hipcc --offload-arch=gfx950 -O3 -o clock clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "../../ecfft_amd/csrc/field_secp256k1.h"
#include "../../ecfft_amd/csrc/field_m31.h"
using namespace ecfft;

constexpr int ITER = 32768;   // long kernels (several ms): launch ramp and block stagger are negligible
struct Stamp { unsigned long long cyc, wall; };

#define KERNEL_BODY(NAME, DECL, OPS, FOLD)                                                         \
__global__ __launch_bounds__(256) void NAME(uint32_t* out, Stamp* st, uint32_t seed) {              \
    uint32_t t = threadIdx.x + blockIdx.x * 256 + seed;                                             \
    DECL                                                                                            \
    unsigned long long c_0 = __builtin_amdgcn_s_memtime(), w_0 = wall_clock64();                    \
    for (int it = 0; it < ITER; ++it) { OPS }                                                       \
    unsigned long long c_1 = __builtin_amdgcn_s_memtime(), w_1 = wall_clock64();                    \
    out[threadIdx.x + blockIdx.x * 256] = FOLD;                                                     \
    if (threadIdx.x == 0) { st[blockIdx.x].cyc = c_1 - c_0; st[blockIdx.x].wall = w_1 - w_0; }      \
}
#define DECL64 uint64_t c0=t,c1=t+1,c2=t+2,c3=t+3,c4=t+4,c5=t+5,c6=t+6,c7=t+7; uint32_t a=t*2654435761u|1u, b=t^0x9e3779b9u;
#define FOLD64 (uint32_t)(c0^c1^c2^c3^c4^c5^c6^c7) ^ (uint32_t)((c0^c1^c2^c3^c4^c5^c6^c7)>>32)
#define DECL32 uint32_t c0=t,c1=t+1,c2=t+2,c3=t+3,c4=t+4,c5=t+5,c6=t+6,c7=t+7; uint32_t a=t*2654435761u|1u, b=t^0x9e3779b9u;
#define FOLD32 (c0^c1^c2^c3^c4^c5^c6^c7)
#define OP3(INS,c) asm volatile(INS " %0, %0, %1" : "+v"(c) : "v"(a));
#define ALL8(M, INS) M(INS,c0) M(INS,c1) M(INS,c2) M(INS,c3) M(INS,c4) M(INS,c5) M(INS,c6) M(INS,c7)
#define OP_MAD64(c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "vcc");
#define OPCO(c) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc" : "+v"(c) : "v"(a), "v"(b) : "vcc");
#define OPMADCO(c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32_e32 %3, vcc, 0, %3, vcc" : "+v"(c), "+v"(x) : "v"(a), "v"(b) : "vcc");
#define DECLF32 float c0=t,c1=t+1,c2=t+2,c3=t+3,c4=t+4,c5=t+5,c6=t+6,c7=t+7; float a=1.0000001f, b=1e-9f;
#define OPF32(c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c) : "v"(a), "v"(b));
KERNEL_BODY(k_fma_f32, DECLF32, OPF32(c0) OPF32(c1) OPF32(c2) OPF32(c3) OPF32(c4) OPF32(c5) OPF32(c6) OPF32(c7), (uint32_t)(c0+c1+c2+c3+c4+c5+c6+c7))
KERNEL_BODY(k_add_u32, DECL32, ALL8(OP3, "v_add_u32"), FOLD32)
KERNEL_BODY(k_mul_lo_u32, DECL32, ALL8(OP3, "v_mul_lo_u32"), FOLD32)
KERNEL_BODY(k_add_co_addc, DECL32, OPCO(c0) OPCO(c1) OPCO(c2) OPCO(c3) OPCO(c4) OPCO(c5) OPCO(c6) OPCO(c7), FOLD32)
KERNEL_BODY(k_mad_u64_u32, DECL64, OP_MAD64(c0) OP_MAD64(c1) OP_MAD64(c2) OP_MAD64(c3) OP_MAD64(c4) OP_MAD64(c5) OP_MAD64(c6) OP_MAD64(c7), FOLD64)
KERNEL_BODY(k_mad_addc, DECL64 uint32_t x = t;, OPMADCO(c0) OPMADCO(c1) OPMADCO(c2) OPMADCO(c3) OPMADCO(c4) OPMADCO(c5) OPMADCO(c6) OPMADCO(c7), (FOLD64) ^ x)


// ---- round 3: the instruction classes of the two candidate replacements for the integer table multiply -------------------
#define DECLF64 double c0=t,c1=t+1,c2=t+2,c3=t+3,c4=t+4,c5=t+5,c6=t+6,c7=t+7; double a=1.0000001, b=1e-9;
#define OPF64(c) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(c) : "v"(a), "v"(b));
#define OPADDF64(c) asm volatile("v_add_f64 %0, %0, %1" : "+v"(c) : "v"(b));
#define OP_MADI64(c) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "vcc");
#define OP_LSHLADD64(c) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(c) : "v"(c7));
#define OP_SWAP(c) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(a));
#define OP_MUL24(c) asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(c) : "v"(a));
#define OP_XOR(c) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(c) : "v"(a));
KERNEL_BODY(k_fma_f64, DECLF64, OPF64(c0) OPF64(c1) OPF64(c2) OPF64(c3) OPF64(c4) OPF64(c5) OPF64(c6) OPF64(c7), (uint32_t)(c0+c1+c2+c3+c4+c5+c6+c7))
KERNEL_BODY(k_add_f64, DECLF64, OPADDF64(c0) OPADDF64(c1) OPADDF64(c2) OPADDF64(c3) OPADDF64(c4) OPADDF64(c5) OPADDF64(c6) OPADDF64(c7), (uint32_t)(c0+c1+c2+c3+c4+c5+c6+c7))
KERNEL_BODY(k_mad_i64_i32, DECL64, OP_MADI64(c0) OP_MADI64(c1) OP_MADI64(c2) OP_MADI64(c3) OP_MADI64(c4) OP_MADI64(c5) OP_MADI64(c6) OP_MADI64(c7), FOLD64)
KERNEL_BODY(k_lshl_add_u64, DECL64, OP_LSHLADD64(c0) OP_LSHLADD64(c1) OP_LSHLADD64(c2) OP_LSHLADD64(c3) OP_LSHLADD64(c4) OP_LSHLADD64(c5) OP_LSHLADD64(c6) OP_LSHLADD64(c0), FOLD64)
KERNEL_BODY(k_permlane32_swap, DECL32, OP_SWAP(c0) OP_SWAP(c1) OP_SWAP(c2) OP_SWAP(c3) OP_SWAP(c4) OP_SWAP(c5) OP_SWAP(c6) OP_SWAP(c7), FOLD32 ^ a)
KERNEL_BODY(k_mul_u32_u24, DECL32, ALL8(OP3, "v_mul_u32_u24_e32"), FOLD32)
KERNEL_BODY(k_xor_b32, DECL32, ALL8(OP3, "v_xor_b32_e32"), FOLD32)

// v_mfma_i32_32x32x32_i8: NACC independent accumulators per wave (1: dependent chain on one accumulator)
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_i8(uint32_t* out, Stamp* st, uint32_t seed) {
    uint32_t t = threadIdx.x + blockIdx.x * 256 + seed;
    v4i_t A = {(int)t, (int)(t * 3u), (int)(t * 5u), (int)(t * 7u)}, B = {(int)(t ^ 0x55u), (int)(t * 11u), (int)(t * 13u), (int)(t * 17u)};
    v16i_t acc[NACC];
    for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = (int)t + k + r;
    unsigned long long c_0 = __builtin_amdgcn_s_memtime(), w_0 = wall_clock64();
#pragma unroll 1
    for (int it = 0; it < ITER / 8; ++it) {
#pragma unroll
        for (int u = 0; u < 8 / NACC; ++u)
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc[k], 0, 0, 0);
    }
    unsigned long long c_1 = __builtin_amdgcn_s_memtime(), w_1 = wall_clock64();
    uint32_t f = 0; for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) f ^= (uint32_t)acc[k][r];
    out[threadIdx.x + blockIdx.x * 256] = f;
    if (threadIdx.x == 0) { st[blockIdx.x].cyc = c_1 - c_0; st[blockIdx.x].wall = w_1 - w_0; }
}

// the kernels' table multiply (169 instructions) as a dependent chain
__global__ __launch_bounds__(256) void k_tmul_chain(uint32_t* out, Stamp* st, uint32_t seed) {
    uint32_t t = threadIdx.x + blockIdx.x * 256 + seed;
    Fe256 x, c; Te256 T;
    for (int i = 0; i < 8; ++i) { x.l[i] = t * (2654435761u + i) ^ 0x9e3779b9u; c.l[i] = t + i; T.t.l[i] = t * 40503u + i * 977u; T.u.l[i] = t * 69069u + i; }
    x.l[7] &= 0x7fffffffu; c.l[7] &= 0x7fffffffu; T.t.l[7] &= 0x7fffffffu; T.u.l[7] &= 0x7fffffffu;
    unsigned long long c_0 = __builtin_amdgcn_s_memtime(), w_0 = wall_clock64();
#pragma unroll 1
    for (int it = 0; it < ITER / 8; ++it) x = Secp256k1::tmul_add(T, x, c);
    unsigned long long c_1 = __builtin_amdgcn_s_memtime(), w_1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * 256] = x.l[0] ^ x.l[7];
    if (threadIdx.x == 0) { st[blockIdx.x].cyc = c_1 - c_0; st[blockIdx.x].wall = w_1 - w_0; }
}

// M31 table multiply (7 instructions: v_mad_u64_u32, v_and, v_alignbit, v_add, v_and, v_lshrrev, v_add), K independent
// chains per lane: the butterfly kernels' multiply with and without instruction-level parallelism inside a wave
template <int K>
__global__ __launch_bounds__(256) void k_m31_chain(uint32_t* out, Stamp* st, uint32_t seed) {
    uint32_t t = threadIdx.x + blockIdx.x * 256 + seed;
    uint32_t x[K], c[K], T[K];
    for (int k = 0; k < K; ++k) { x[k] = (t * (2654435761u + k)) & 0x7fffffffu; c[k] = (t + k) & 0x7fffffffu; T[k] = (t * 40503u + k * 977u) & 0x3fffffffu; }
    unsigned long long c_0 = __builtin_amdgcn_s_memtime(), w_0 = wall_clock64();
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int k = 0; k < K; ++k) x[k] = M31::tmul_add(T[k], x[k], c[k]);
    }
    unsigned long long c_1 = __builtin_amdgcn_s_memtime(), w_1 = wall_clock64();
    uint32_t f = 0; for (int k = 0; k < K; ++k) f ^= x[k];
    out[threadIdx.x + blockIdx.x * 256] = f;
    if (threadIdx.x == 0) { st[blockIdx.x].cyc = c_1 - c_0; st[blockIdx.x].wall = w_1 - w_0; }
}

template <class K>
void run(const char* name, K kern, uint32_t* d_out, Stamp* d_st, double ops_per_iter, int iters) {
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = 256 * wps;                       // one 256-thread block = one wave per SIMD of a CU
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        kern<<<blocks, 256>>>(d_out, d_st, 1); (void)hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            (void)hipEventRecord(e0); kern<<<blocks, 256>>>(d_out, d_st, r); (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        static Stamp h[256 * 8];
        (void)hipMemcpy(h, d_st, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0; for (int i = 0; i < blocks; ++i) { cyc += h[i].cyc; wall += h[i].wall; }
        cyc /= blocks; wall /= blocks;
        double mhz = cyc / wall * 100.0;
        // cycles per wave-instruction per SIMD from the WALL time of the whole launch at the clock measured inside it.  (Round 2 also
        // printed d(s_memtime) / (wps * iters * ops) per block, which assumes all `wps` requested blocks of a CU are resident together:
        // kernels whose registers allow fewer waves per SIMD run their blocks in turns and that column read up to 2.3x too low.)
        double insts_per_simd = (double)blocks * 4 * iters * ops_per_iter / 1024.0;
        double per_inst = best * 1e-3 * mhz * 1e6 / insts_per_simd;
        double nominal = best * 1e-3 * 2.4e9 / insts_per_simd;
        printf("%-18s blocks/CU %d: %7.3f ms  eff.clock %6.0f MHz  %5.2f cyc/wave-inst/SIMD at that clock  (%5.2f if priced at 2.4 GHz)\n",
               name, wps, best, mhz, per_inst, nominal);
    }
}

int main() {
    uint32_t* d_out; Stamp* d_st;
    (void)hipMalloc(&d_out, 256 * 8 * 256 * 4); (void)hipMalloc(&d_st, sizeof(Stamp) * 256 * 8);
    run("v_fma_f64", k_fma_f64, d_out, d_st, 8, ITER);
    run("v_add_f64", k_add_f64, d_out, d_st, 8, ITER);
    run("v_mad_i64_i32", k_mad_i64_i32, d_out, d_st, 8, ITER);
    run("v_lshl_add_u64", k_lshl_add_u64, d_out, d_st, 8, ITER);
    run("v_permlane32_swap", k_permlane32_swap, d_out, d_st, 8, ITER);
    run("v_mul_u32_u24", k_mul_u32_u24, d_out, d_st, 8, ITER);
    run("v_xor_b32", k_xor_b32, d_out, d_st, 8, ITER);
    run("mfma_i32_32x32x32_i8 x1 (dependent)", k_mfma_i8<1>, d_out, d_st, 8, ITER / 8);
    run("mfma_i32_32x32x32_i8 x2", k_mfma_i8<2>, d_out, d_st, 8, ITER / 8);
    run("mfma_i32_32x32x32_i8 x4", k_mfma_i8<4>, d_out, d_st, 8, ITER / 8);
    if (getenv("CLOCK_R3_ONLY")) return 0;
    run("v_fma_f32", k_fma_f32, d_out, d_st, 8, ITER);
    run("v_add_u32", k_add_u32, d_out, d_st, 8, ITER);
    run("v_mul_lo_u32", k_mul_lo_u32, d_out, d_st, 8, ITER);
    run("add_co+addc", k_add_co_addc, d_out, d_st, 16, ITER);
    run("v_mad_u64_u32", k_mad_u64_u32, d_out, d_st, 8, ITER);
    run("mad_u64+addc", k_mad_addc, d_out, d_st, 16, ITER);
    run("tmul_add (169 inst)", k_tmul_chain, d_out, d_st, 169, ITER / 8);
    run("m31 tmul_add x1 (7)", k_m31_chain<1>, d_out, d_st, 7, ITER);
    run("m31 tmul_add x2 (7)", k_m31_chain<2>, d_out, d_st, 14, ITER);
    run("m31 tmul_add x4 (7)", k_m31_chain<4>, d_out, d_st, 28, ITER);
    run("m31 tmul_add x8 (7)", k_m31_chain<8>, d_out, d_st, 56, ITER);
    return 0;
}
