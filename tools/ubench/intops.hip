// Integer-ALU issue-rate microbenchmark for gfx950 (MI355X).
// Decides the limb width / instruction mix of the 256-bit modular multiply
// (DESIGN.md "Field arithmetic on the CDNA4 VALU").  Each kernel runs ITER
// iterations of 8 independent dependency chains of ONE instruction per lane;
// reported: wave-instructions per cycle per SIMD assuming 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 4096;

#define KERNEL_BODY(NAME, DECL, OPS, FOLD)                                       \
__global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {      \
    uint32_t t = threadIdx.x + blockIdx.x * 256 + seed;                          \
    DECL                                                                         \
    for (int it = 0; it < ITER; ++it) { OPS }                                    \
    out[threadIdx.x + blockIdx.x * 256] = FOLD;                                  \
}

// 64-bit accumulators c0..c7, multiplicands a,b
#define DECL64 uint64_t c0=t,c1=t+1,c2=t+2,c3=t+3,c4=t+4,c5=t+5,c6=t+6,c7=t+7; uint32_t a=t*2654435761u|1u, b=t^0x9e3779b9u;
#define FOLD64 (uint32_t)(c0^c1^c2^c3^c4^c5^c6^c7) ^ (uint32_t)((c0^c1^c2^c3^c4^c5^c6^c7)>>32)
#define OP_MAD64(c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "vcc");
KERNEL_BODY(k_mad_u64_u32, DECL64,
    OP_MAD64(c0) OP_MAD64(c1) OP_MAD64(c2) OP_MAD64(c3) OP_MAD64(c4) OP_MAD64(c5) OP_MAD64(c6) OP_MAD64(c7), FOLD64)

#define DECL32 uint32_t c0=t,c1=t+1,c2=t+2,c3=t+3,c4=t+4,c5=t+5,c6=t+6,c7=t+7; uint32_t a=t*2654435761u|1u, b=t^0x9e3779b9u;
#define FOLD32 (c0^c1^c2^c3^c4^c5^c6^c7)
#define OP3(INS,c) asm volatile(INS " %0, %0, %1" : "+v"(c) : "v"(a));
#define OP4(INS,c) asm volatile(INS " %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#define ALL8(M, INS) M(INS,c0) M(INS,c1) M(INS,c2) M(INS,c3) M(INS,c4) M(INS,c5) M(INS,c6) M(INS,c7)
KERNEL_BODY(k_mul_lo_u32, DECL32, ALL8(OP3, "v_mul_lo_u32"), FOLD32)
KERNEL_BODY(k_mul_hi_u32, DECL32, ALL8(OP3, "v_mul_hi_u32"), FOLD32)
KERNEL_BODY(k_mul_u32_u24, DECL32, ALL8(OP3, "v_mul_u32_u24"), FOLD32)
KERNEL_BODY(k_mul_hi_u32_u24, DECL32, ALL8(OP3, "v_mul_hi_u32_u24"), FOLD32)
KERNEL_BODY(k_mad_u32_u24, DECL32, ALL8(OP4, "v_mad_u32_u24"), FOLD32)
KERNEL_BODY(k_add_u32, DECL32, ALL8(OP3, "v_add_u32"), FOLD32)
KERNEL_BODY(k_add3_u32, DECL32, ALL8(OP4, "v_add3_u32"), FOLD32)
KERNEL_BODY(k_lshl_add_u32, DECL32, ALL8(OP4, "v_lshl_add_u32"), FOLD32)
KERNEL_BODY(k_alignbit, DECL32, ALL8(OP4, "v_alignbit_b32"), FOLD32)
#define OPCO(c) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc" : "+v"(c) : "v"(a), "v"(b) : "vcc");
KERNEL_BODY(k_add_co_addc_pair, DECL32, OPCO(c0) OPCO(c1) OPCO(c2) OPCO(c3) OPCO(c4) OPCO(c5) OPCO(c6) OPCO(c7), FOLD32)
#define OPCO64Z(c) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32_e64 %0, vcc, 0, 0, vcc" : "+v"(c) : "v"(a) : "vcc");
KERNEL_BODY(k_addc_e64_zero, DECL32, OPCO64Z(c0) OPCO64Z(c1) OPCO64Z(c2) OPCO64Z(c3) OPCO64Z(c4) OPCO64Z(c5) OPCO64Z(c6) OPCO64Z(c7), FOLD32)
#define OPCO32Z(c) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32_e32 %0, vcc, 0, %2, vcc" : "+v"(c) : "v"(a), "v"(b) : "vcc");
KERNEL_BODY(k_addc_e32_zero, DECL32, OPCO32Z(c0) OPCO32Z(c1) OPCO32Z(c2) OPCO32Z(c3) OPCO32Z(c4) OPCO32Z(c5) OPCO32Z(c6) OPCO32Z(c7), FOLD32)
#define OPCND(c) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(c) : "v"(a) : "s10", "s11");
KERNEL_BODY(k_cndmask_e64, DECL32, OPCND(c0) OPCND(c1) OPCND(c2) OPCND(c3) OPCND(c4) OPCND(c5) OPCND(c6) OPCND(c7), FOLD32)
#define OPMADCO(c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_addc_co_u32_e32 %3, vcc, 0, %3, vcc" : "+v"(c), "+v"(x) : "v"(a), "v"(b) : "vcc");
KERNEL_BODY(k_mad_addc_pair, DECL64 uint32_t x = t;, OPMADCO(c0) OPMADCO(c1) OPMADCO(c2) OPMADCO(c3) OPMADCO(c4) OPMADCO(c5) OPMADCO(c6) OPMADCO(c7), (FOLD64) ^ x)
#define OPLA64(c) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c) : "v"(c7));
KERNEL_BODY(k_lshl_add_u64, DECL64, OPLA64(c0) OPLA64(c1) OPLA64(c2) OPLA64(c3) OPLA64(c4) OPLA64(c5) OPLA64(c6) c7 += a;, FOLD64)
#define OPMADU16(c) asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
KERNEL_BODY(k_mad_u32_u16, DECL32, OPMADU16(c0) OPMADU16(c1) OPMADU16(c2) OPMADU16(c3) OPMADU16(c4) OPMADU16(c5) OPMADU16(c6) OPMADU16(c7), FOLD32)
#define OPDOT4(c) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
KERNEL_BODY(k_dot4_u32_u8, DECL32, OPDOT4(c0) OPDOT4(c1) OPDOT4(c2) OPDOT4(c3) OPDOT4(c4) OPDOT4(c5) OPDOT4(c6) OPDOT4(c7), FOLD32)

#define DECLF64 double c0=t,c1=t+1,c2=t+2,c3=t+3,c4=t+4,c5=t+5,c6=t+6,c7=t+7; double a=1.0000001, b=1e-9;
#define FOLDF64 (uint32_t)(c0+c1+c2+c3+c4+c5+c6+c7)
#define OPF64(c) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(c) : "v"(a), "v"(b));
KERNEL_BODY(k_fma_f64, DECLF64, OPF64(c0) OPF64(c1) OPF64(c2) OPF64(c3) OPF64(c4) OPF64(c5) OPF64(c6) OPF64(c7), FOLDF64)
#define DECLF32 float c0=t,c1=t+1,c2=t+2,c3=t+3,c4=t+4,c5=t+5,c6=t+6,c7=t+7; float a=1.0000001f, b=1e-9f;
#define OPF32(c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c) : "v"(a), "v"(b));
KERNEL_BODY(k_fma_f32, DECLF32, OPF32(c0) OPF32(c1) OPF32(c2) OPF32(c3) OPF32(c4) OPF32(c5) OPF32(c6) OPF32(c7), FOLDF64)

template <class K>
int run(const char* name, K kern, uint32_t* d_out, int ops_per_iter) {
    const int blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves/SIMD
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    kern<<<blocks, 256>>>(d_out, 1); CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHK(hipEventRecord(e0)); kern<<<blocks, 256>>>(d_out, r); CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1)); float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double wave_insts = (double)blocks * 4 * ITER * ops_per_iter;
    double per_simd = wave_insts / 1024.0;                 // 256 CUs x 4 SIMDs
    double cycles = best * 1e-3 * 2.4e9;
    printf("%-22s %8.3f ms  %6.2f cyc/wave-inst/SIMD (@2.4GHz)  %8.2f Tlane-op/s\n", name, best,
           cycles / per_simd, wave_insts * 64 / (best * 1e-3) / 1e12);
    return 0;
}

int main() {
    uint32_t* d_out; CHK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
    run("v_fma_f32", k_fma_f32, d_out, 8);
    run("v_fma_f64", k_fma_f64, d_out, 8);
    run("v_add_u32", k_add_u32, d_out, 8);
    run("v_add3_u32", k_add3_u32, d_out, 8);
    run("v_lshl_add_u32", k_lshl_add_u32, d_out, 8);
    run("v_alignbit_b32", k_alignbit, d_out, 8);
    run("add_co+addc (2 inst)", k_add_co_addc_pair, d_out, 16);
    run("add_co+addc_e64 0,0 (2)", k_addc_e64_zero, d_out, 16);
    run("add_co+addc_e32 0,v (2)", k_addc_e32_zero, d_out, 16);
    run("v_cndmask_b32_e64", k_cndmask_e64, d_out, 8);
    run("mad_u64+addc (2 inst)", k_mad_addc_pair, d_out, 16);
    run("v_lshl_add_u64 (7)", k_lshl_add_u64, d_out, 7);
    run("v_mul_lo_u32", k_mul_lo_u32, d_out, 8);
    run("v_mul_hi_u32", k_mul_hi_u32, d_out, 8);
    run("v_mad_u64_u32", k_mad_u64_u32, d_out, 8);
    run("v_mul_u32_u24", k_mul_u32_u24, d_out, 8);
    run("v_mul_hi_u32_u24", k_mul_hi_u32_u24, d_out, 8);
    run("v_mad_u32_u24", k_mad_u32_u24, d_out, 8);
    run("v_mad_u32_u16", k_mad_u32_u16, d_out, 8);
    run("v_dot4_u32_u8", k_dot4_u32_u8, d_out, 8);
    return 0;
}
