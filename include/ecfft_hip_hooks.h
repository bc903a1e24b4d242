/* TEST AND MEASUREMENT HOOKS of libecfft_hip — NOT part of the shipped library.
 *
 * The reference exposes no such surface (/root/reference/src/fftree.rs:23-38, 123, 164, 227: tables and three methods), and the
 * product build of ecfft_amd/csrc/ecfft_capi.hip exports none of the symbols below.  They exist only in a build with
 * -DECFFT_TEST_HOOKS (tests/hooks/libecfft_hip_hooks.so, made by tests/hooks/build_hooks.py and __graft_entry__.build()), which
 * the tests and the measurement tools load explicitly.  The same build also reads the A/B switches of the tuning experiments from
 * the environment (ECFFT_NO_MFMA, ECFFT_NO_LOW16, ECFFT_LOW32, ECFFT_NO_ROW256, ECFFT_NO_COL256, ECFFT_NO_SMALL_TILES,
 * ECFFT_SMALL_TILES_MAX, ECFFT_SMALL_LOW_MAX, ECFFT_SMALL_MIN_LOGC, ECFFT_NO_FULL_CYCLIC, ECFFT_SPLIT_GATHER_MAX_LOG,
 * ECFFT_SPLIT_Q2_SPLIT); the shipped library reads no environment variable at all. */
#ifndef ECFFT_HIP_HOOKS_H
#define ECFFT_HIP_HOOKS_H
#include "ecfft_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* test hook: entries of z0_s1 / z1_s0 of the subtree with m leaves (built as the reference does, src/fftree.rs:386-397) that
 * differ from the pointwise isogeny-chain formula the sharded builds use; 0 = identical, -1 = error */
long ecfft_selfcheck_pointwise_z(ecfft_ctx* ctx, size_t m);

/* MEASUREMENT ONLY: one rank of a `world`-rank job timed on its own.  Every exchange with a remote peer costs delay_us + (bytes on
 * the exchange's most loaded LINK: the messages to one peer add up) / link_gbps GB/s as a spinning kernel on the caller's stream, and the rank's own send buffers are copied
 * into its receive buffers: the stream's timeline is that of a rank whose peers answer after exactly the modelled time; the
 * RESULTS of a sharded call on such a communicator are meaningless (tools/split_project.py: per-rank compute, exchanges, bytes and
 * exposed communication time of the split transforms without multi-GPU hardware).  link_gbps = 0: latency only. */
int ecfft_comm_init_projection(int world, int rank, int device, double delay_us, double link_gbps, ecfft_comm** out);

/* test hooks of the agreement protocol (ecfft_hip.h "Failure on ONE rank"): the local preparation of the context's next sharded
 * call of a NEW shape reports failure */
int ecfft_test_fail_next_collective(ecfft_ctx* ctx);
/* test hook, process wide: the local part of the next collective ecfft_build_exit_shard reports failure on rank `rank` (-1: off) */
int ecfft_test_fail_build_rank(int rank);

/* Test hook: the DEVICE field arithmetic on raw residues (plain integers < p, no Montgomery interpretation), host buffers.
 * op 0: out = a*b + c mod p   1: a*b   2: a - b   3: a + b   4 / 5: a*b + c / a*b with a taken as a TABLE constant, i.e. the
 * multiply of the butterfly kernels, result in the kernels' internal (lazy) range, not canonicalised.  Lets the tests drive the hand-written gfx950 multiply with
 * directed operands (results next to p and 2^256, carry-out of the second fold) that random data never reaches. */
int ecfft_selftest_field(int field, int op, const void* a, const void* b, const void* c, void* out, size_t n, int device);

/* Test hook (secp256k1): the matrix-core form of the innermost 16-point map (ecfft_amd/csrc/mfma_blk16.h) with an EXPLICIT map —
 * matrix256 = 16 x 16 plain residues < p, row-major [output][input]; x / out = n raw residues (n a multiple of 1024), every aligned
 * block of 16 is mapped to out_o = sum_i matrix[o][i] * x_i mod p.  Lets the tests reach the carry-out and canonicalisation
 * branches of the normalisation (identity / -1 / 0 constants, inputs next to 0, p and 2^32 + 977) that a tree's constants never hit. */
int ecfft_selftest_blk16(const void* matrix256, const void* x, void* out, size_t n, int device);
/* ... the SMALL-LAUNCH forms of the same map (v_mfma_i32_16x16x64_i8; 256-element tiles of the latency regime).
 * mode 1: 256-element arrays, 4 waves, LDS-resident (k_enter_low<8,256>'s low16)   2: 256-element arrays, 2 waves, two results per
 * lane (k_exit_low<8,128>'s low16)   3: 128-element arrays = 8 blocks, 2 waves, element in registers (k_exit_low<8,128>'s half-tiles)
 * 4: 256-element arrays, 4 waves, element in registers (k_stages_row256, k_enter_low<8,256>'s EXTEND cores).  n: a multiple of 256. */
int ecfft_selftest_blk16_small(const void* matrix256, const void* x, void* out, size_t n, int mode, int device);
/* ... the 32-point form (round 4: the five lowest ENTER / EXIT levels of the 1024-element low-level kernels as one map): matrix1024 =
 * 32 x 32 plain residues, row-major [output][input]; every aligned block of 32 of x is mapped; n a multiple of 1024. */
int ecfft_selftest_blk32(const void* matrix1024, const void* x, void* out, size_t n, int device);
/* measurement / test hook: which composite map the context's 1024-element low-level kernels run for the lowest levels of ENTER
 * (dir 0) / EXIT (dir 1): 32 = levels 1..5, 16 = levels 1..4, 0 = level code (secp256k1 only; A/B: ECFFT_LOW32, ECFFT_NO_LOW16) */
int ecfft_ctx_low_map(const ecfft_ctx* ctx, int dir);

#ifdef __cplusplus
}
#endif
#endif
