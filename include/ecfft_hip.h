/* ecfft_hip.h — C ABI of the MI355X (gfx950) ECFFT hot path.
 *
 * Drop-in boundary for the EXTEND / ENTER / EXIT path of andrewmilson/ecfft.  The reference has no
 * FFI layer (pure generic Rust); each entry point below names the Rust item whose body a thin
 * `impl` would forward to it (see INTEGRATION.md for the binding a maintainer would add):
 *
 *   ecfft_build_fftree            <-> FftreeField::build_fftree(n) -> Option<FFTree<Self>>   src/lib.rs:14-16, 39-85, 198-215
 *   ecfft_fftree_new              <-> FFTree::new(leaves, rational_maps)                       src/fftree.rs:42-70
 *   ecfft_enter                   <-> FFTree::enter(&self, &[F]) -> Vec<F>                     src/fftree.rs:164-167
 *   ecfft_exit                    <-> FFTree::exit(&self, &[F]) -> Vec<F>                      src/fftree.rs:227-230
 *   ecfft_extend                  <-> FFTree::extend(&self, &[F], Moiety) -> Vec<F>            src/fftree.rs:123-126
 *   ecfft_tree_size / _table      <-> the pub fields of FFTree<F> / subtree_with_size          src/fftree.rs:24-38, 489-496
 *   ecfft_fftree_serialize / _deserialize <-> impl CanonicalSerialize / CanonicalDeserialize  src/fftree.rs:507-660
 *   ecfft_ctx_destroy             <-> Drop
 *
 * Element representation = the crate's in-memory one, so Rust slices pass through untouched:
 *   ECFFT_FIELD_SECP256K1: 32 bytes = [u64; 4] little-endian limbs of x * 2^256 mod p (ark-ff
 *                          Fp256<MontBackend<FqConfig, 4>>, src/lib.rs:37), fully reduced;
 *   ECFFT_FIELD_M31:       4 bytes  = u32 canonical residue (ark_ff_optimized::fp31::Fp, src/lib.rs:196).
 * Outputs are fully reduced, so equality of bytes == equality of field elements (assert_eq! in the
 * reference's tests).
 *
 * Errors: the reference panics (src/fftree.rs:40 "TODO: errors"); the ABI returns a status instead:
 *   - length not a power of two   (assert!, src/fftree.rs:490, src/lib.rs:41)  -> ECFFT_ERR_NOT_POW2
 *   - "FFTree is too small"       (panic!, src/fftree.rs:494)                   -> ECFFT_ERR_TREE_TOO_SMALL
 *   - build_fftree returning None (src/lib.rs:62-64, src/ec.rs:513-515)         -> ECFFT_ERR_TREE_TOO_LARGE, *out = NULL
 * There is no CPU fallback: without a usable HIP device every call fails with ECFFT_ERR_HIP.
 *
 * Threading: a context is immutable after creation (like &FFTree); transform calls on one context
 * serialise on its scratch buffers (a host mutex orders the enqueues, a HIP event orders the device
 * work across streams); use one context per host thread/stream for concurrency.
 * Ownership: the caller owns every buffer; `stream` is a hipStream_t passed as void* (NULL = default).
 */
#ifndef ECFFT_HIP_H
#define ECFFT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ecfft_ctx ecfft_ctx;

enum { ECFFT_FIELD_SECP256K1 = 0, ECFFT_FIELD_M31 = 1 };
enum { ECFFT_S0 = 0, ECFFT_S1 = 1 };                 /* enum Moiety, src/fftree.rs:17-21 (the TARGET moiety) */
enum { ECFFT_MEM_HOST = 0, ECFFT_MEM_DEVICE = 1 };   /* where in/out pointers live */
enum {
    ECFFT_OK = 0,
    ECFFT_ERR_NOT_POW2 = 1,
    ECFFT_ERR_TREE_TOO_SMALL = 2,
    ECFFT_ERR_TREE_TOO_LARGE = 3,
    ECFFT_ERR_HIP = 4,
    ECFFT_ERR_BAD_ARG = 5
};
/* tables of FFTree<F> exported by ecfft_tree_table (src/fftree.rs:25-37) */
enum {
    ECFFT_TBL_F = 0,              /* BinaryTree<F>, 2m entries, heap order          */
    ECFFT_TBL_RECOMBINE = 1,      /* BinaryTree<Mat2x2<F>>, m matrices = 4m elements, row-major, heap order (src/fftree.rs:26) */
    ECFFT_TBL_DECOMPOSE = 2,      /* likewise (src/fftree.rs:27); rebuilt on demand from the normalised tables */
    ECFFT_TBL_XNN_S = 3, ECFFT_TBL_XNN_S_INV = 4, ECFFT_TBL_Z0_S1 = 5, ECFFT_TBL_Z1_S0 = 6,
    ECFFT_TBL_Z0_INV_S1 = 7, ECFFT_TBL_Z1_INV_S0 = 8, ECFFT_TBL_Z0Z0_REM_XNN_S = 9, ECFFT_TBL_Z1Z1_REM_XNN_S = 10
};

/* size in bytes of one field element of `field` (32 or 4); 0 for an unknown field */
size_t ecfft_elem_size(int field);

/* FftreeField::build_fftree(n): builds the whole subtree chain T_1..T_n on `device`. */
int ecfft_build_fftree(int field, size_t n, int device, ecfft_ctx** out);

/* FFTree::new(leaves, rational_maps): n leaves and log2(n) maps, each map given as 3 numerator and
 * 3 denominator coefficients (low -> high, zero padded) — all in the element representation above. */
int ecfft_fftree_new(int field, const void* leaves, size_t n, const void* map_num3, const void* map_den3,
                     int device, ecfft_ctx** out);

void ecfft_ctx_destroy(ecfft_ctx* ctx);

size_t ecfft_tree_size(const ecfft_ctx* ctx);   /* number of leaves of the top tree */
int ecfft_field(const ecfft_ctx* ctx);
size_t ecfft_ctx_device_bytes(const ecfft_ctx* ctx);   /* HBM the context holds between calls: tables + transform scratch + pooled temporaries (+ gathered cyclic tables) */
/* The algorithm wrappers (ecfft_redc, ecfft_vanish, ecfft_degree, the sharded transforms ...) keep their temporaries in a
 * per-context pool between calls; the pool is capped (idle blocks beyond twice the transform scratch are freed at the end of a
 * call) and this call returns every idle block — except the pinned temporaries of sharded call shapes the ranks have agreed on, which
 * stay so that those calls keep running without allocation — and the host-call staging buffer to the device.  Call between transforms.
 * A full context that has served sharded EXTENDs also holds compact copies of the cyclic stages' table entries (gathered on
 * first use, counted by ecfft_ctx_device_bytes); they are returned too and gathered again when needed. */
int ecfft_ctx_trim(ecfft_ctx* ctx);

/* coefficients -> evaluations on the leaves of T_n (n = len; any power of two <= tree size) */
int ecfft_enter(ecfft_ctx* ctx, const void* coeffs, void* evals, size_t n, int mem, void* stream);
/* evaluations -> coefficients */
int ecfft_exit(ecfft_ctx* ctx, const void* evals, void* coeffs, size_t n, int mem, void* stream);
/* batched forms (no reference counterpart): `count` independent polynomials of length n, laid end to end, share every
 * kernel launch and every table read — the throughput mode for provers that transform many columns.  (An even batch of at
 * least 2^20 elements runs as two half-batches of whole polynomials on two streams, joined on `stream` before the call's work
 * is complete in stream order: same results, 4-7 % faster) */
int ecfft_enter_many(ecfft_ctx* ctx, const void* coeffs, void* evals, size_t n, size_t count, int mem, void* stream);
int ecfft_exit_many(ecfft_ctx* ctx, const void* evals, void* coeffs, size_t n, size_t count, int mem, void* stream);
/* `count` vectors of `e` evaluations on the moiety opposite to `moiety` -> evaluations on `moiety`
 * of T_{2e}; vectors are laid end to end (count = 1 is FFTree::extend).  (An even batch of >= 2^20 elements of vectors with
 * e >= 2^19 runs as two half-batches on two streams, like the batched ENTER / EXIT.) */
int ecfft_extend(ecfft_ctx* ctx, const void* in, void* out, size_t e, int moiety, size_t count, int mem, void* stream);

/* The remaining FFTree algorithms, composed from the same GPU kernels.  Synchronous.
 *   ecfft_mextend         <-> FFTree::mextend(&self, &[F], Moiety)      src/fftree.rs:138-141
 *   ecfft_redc            <-> FFTree::redc_z0 / redc_z1(&self, evals, a)  src/fftree.rs:264-275  (moiety S0 / S1)
 *   ecfft_modular_reduce  <-> FFTree::modular_reduce(&self, evals, a, c)  src/fftree.rs:286-289
 *   ecfft_vanish          <-> FFTree::vanish(&self, domain) -> 2*nd evals  src/fftree.rs:313-316
 *   ecfft_degree          <-> FFTree::degree(&self, evals) -> usize        src/fftree.rs:195-198 */
int ecfft_mextend(ecfft_ctx* ctx, const void* in, void* out, size_t e, int moiety, size_t count, int mem, void* stream);
int ecfft_redc(ecfft_ctx* ctx, const void* evals, const void* a, void* out, size_t n, int moiety, int mem, void* stream);
int ecfft_modular_reduce(ecfft_ctx* ctx, const void* evals, const void* a, const void* c, void* out, size_t n, int mem, void* stream);
int ecfft_vanish(ecfft_ctx* ctx, const void* domain, void* out, size_t nd, int mem, void* stream);
int ecfft_degree(ecfft_ctx* ctx, const void* evals, size_t n, int mem, void* stream, size_t* degree);

/* Building blocks of ONE EXTEND of e evaluations (tree T_{2e}) split over P = 2^log_p GPUs, for hosts that drive the
 * exchanges themselves (ecfft_extend_sharded below does the whole thing); see DESIGN.md section 8.  No reference counterpart (the reference is
 * single-process); together they compute exactly FFTree::extend (src/fftree.rs:123-126).
 *   ecfft_extend_top_cyclic : buf = the rank's CYCLIC shard (local j' <-> global j'*P + rank, e/P elements).
 *                             recombine = 0: multiply by 1/W_src, then decompose stages 0..log_p-1;
 *                             recombine = 1: recombine stages log_p-1..0, then multiply by W_target.
 *   ecfft_extend_local_block: buf = the rank's BLOCK shard (global [rank*e/P, (rank+1)*e/P)): every stage
 *                             k >= log_p, decompose then recombine, with the fused single-GPU kernels. */
int ecfft_extend_top_cyclic(ecfft_ctx* ctx, void* buf, size_t e, int moiety, unsigned log_p, unsigned rank, int recombine,
                            int mem, void* stream);
int ecfft_extend_local_block(ecfft_ctx* ctx, void* buf, size_t e, int moiety, unsigned log_p, int mem, void* stream);

/* ---- ONE transform split over several GPUs, one process per GPU (no reference counterpart: the reference is single
 * threaded; together the ranks compute exactly FFTree::extend / enter / exit, src/fftree.rs:123-126, 164-167, 227-230).
 * The loops being split are the butterfly stage loops src/fftree.rs:83-97 and 104-118: stage k pairs (i, i + e >> (k+1)), so
 * stages k >= log2 P are local when the vector is BLOCK distributed (rank r holds [r*len/P, (r+1)*len/P)) and stages
 * k < log2 P are local when it is CYCLIC (position j on rank j mod P); an all-to-all inside the group switches between the
 * two.  Data moves GPU to GPU through an `ecfft_comm`:
 *   ecfft_comm_get_unique_id + ecfft_comm_init_rank   RCCL (ncclGetUniqueId / ncclCommInitRank; librccl.so is loaded at run
 *       time): grouped ncclSend / ncclRecv over xGMI on the caller's stream.  Rank 0 creates the 128-byte id and hands it to
 *       the other ranks by any means (MPI, TCP, a file, torch.distributed); every context of the job is built for its own GPU.
 *   ecfft_comm_init_callback   the host moves the device buffers itself (tests: several ranks sharing one GPU over gloo).
 * Arguments: device pointers only; `in` / `out` = this rank's BLOCK shard (len / world elements, may alias); world = 2^k;
 * len / world >= 2 * world.  Every rank of the communicator makes the same call with the same len.  Asynchronous on `stream`.
 * Exchanges (grouped send / receive calls) per transform: EXTEND 4 (2 per cyclic side saved, ecfft_extend_sharded_layout); ENTER
 * 3 per top level + 1 (Q = 2: 1); EXIT 1 + 9 per top level above the pairs + 1 — inside a level every vector stays cyclic over its
 * group; the level of the PAIRS of ranks (blocks of 2 len / world) runs redundantly on both ranks of a pair from one exchange
 * (round 4), so EXIT takes 2 / 11 / 20 exchanges at world = 2 / 4 / 8 (before: 10 / 19 / 28).  A FULL context (tables replicated) splits an EXIT of at
 * most 2^21 (ECFFT_SPLIT_GATHER_MAX_LOG) differently: one all-gather, then every top level redundantly on the block that contains the
 * rank's chunk — ONE exchange per EXIT (the split top levels are latency bound at such sizes, tools/split_project.py). */
typedef struct ecfft_comm ecfft_comm;
#define ECFFT_COMM_ID_BYTES 128
/* n sends and n receives of device buffers that must progress together; return 0 on success */
typedef int (*ecfft_exchange_fn)(void* user, int n_send, const int* send_peer, const void* const* send_ptr, const size_t* send_bytes,
                                 int n_recv, const int* recv_peer, void* const* recv_ptr, const size_t* recv_bytes, void* stream);
int ecfft_comm_get_unique_id(void* id_out);                                                      /* ECFFT_COMM_ID_BYTES bytes */
int ecfft_comm_init_rank(const void* id, int world, int rank, int device, ecfft_comm** out);
/* The RCCL library ecfft_comm_get_unique_id / ecfft_comm_init_rank bind (dlopen, RTLD_LOCAL) instead of the copy already mapped into
 * the process or /opt/rocm/lib/librccl.so: a differently named RCCL build, or the tests' stand-in (tests/stub_rccl).  NULL or "" =
 * default.  Must precede the first communicator of the process (the binding is made once): ECFFT_ERR_BAD_ARG afterwards.  The library
 * reads no environment variable for this (or for anything else). */
int ecfft_comm_set_rccl_library(const char* path);
int ecfft_comm_init_callback(int world, int rank, int device, ecfft_exchange_fn fn, void* user, ecfft_comm** out);
void ecfft_comm_destroy(ecfft_comm* comm);
/* RCCL transports: ncclCommAbort — unblocks the exchanges in flight (a peer died or never arrived) and makes every later sharded
 * call on this communicator return ECFFT_ERR_HIP; may be called from another host thread than the blocked one.  ECFFT_ERR_HIP for a
 * callback transport (the host owns its exchanges).  The librccl that is bound: ecfft_comm_set_rccl_library. */
int ecfft_comm_abort(ecfft_comm* comm);
/* Link striping of the big pairwise exchanges of a split ENTER / EXIT (round 5): a message travels as `world` slices, slice k via
 * rank k, in two grouped exchanges, so that every link of the xGMI mesh carries 1/world of it per phase — applied to an exchange only
 * when its most loaded link gets lighter by at least `min_gain_bytes` over both phases.  OFF by default (SIZE_MAX = never; round 6:
 * bit-exact over every transport the tests have, never yet timed on xGMI — it doubles the bytes a rank injects and adds an exchange).
 * 4 MiB is the threshold the projection suggests (>= 85 us at 48 GB/s against one more exchange latency); 0 = whenever striping moves
 * fewer bytes over the most loaded link.  Every rank of the communicator must use the same value (each rank decides locally from the
 * call's message pattern): the ranks compare it in the agreement that precedes the first call of every sharded call shape, and a
 * mismatch fails that call on all of them with ECFFT_ERR_HIP.  Only BEFORE the communicator has carried its first exchange:
 * ECFFT_ERR_BAD_ARG afterwards. */
int ecfft_comm_set_link_striping(ecfft_comm* comm, size_t min_gain_bytes);
int ecfft_comm_rank(const ecfft_comm* comm);
int ecfft_comm_world(const ecfft_comm* comm);
/* communication time: while enabled every exchange is bracketed by HIP events on its stream; _read synchronises the device */
int ecfft_comm_stats_enable(ecfft_comm* comm, int on);
int ecfft_comm_stats_read(ecfft_comm* comm, double* comm_ms, double* exchanges, double* bytes_sent);
int ecfft_extend_sharded(ecfft_ctx* ctx, ecfft_comm* comm, const void* in, void* out, size_t e, int moiety, void* stream);
/* Sharded EXTEND-ONLY context: one rank's share of the tables ONE EXTEND of e evaluations over `world` GPUs reads (tree
 * T_2e of build_fftree(2e); SURVEY 8(e) "matrix tables shard the same way").  Holds 26 e/world table constants — the
 * entries i = rank (mod world) of the stage tables for the cyclic stages, the last e/world entries for the block-local stages,
 * the normalisation weights of the rank's positions — instead of the ~84 e elements of the full chain T_1 .. T_2e; no tree
 * is ever materialised on any GPU.  Accepted by ecfft_extend_sharded / ecfft_extend_sharded_layout only (same e, world and rank in the communicator, either
 * moiety), plus ecfft_tree_size / ecfft_field / ecfft_ctx_device_bytes / ecfft_profile_* / ecfft_ctx_destroy; every other call returns
 * ECFFT_ERR_BAD_ARG.  Results are bit-identical to ecfft_extend on a full context.  world = 2^k <= 64, e / world >= 2 * world;
 * ECFFT_ERR_TREE_TOO_LARGE when T_2e exceeds the curve's 2-adicity, as ecfft_build_fftree(2e). */
int ecfft_build_extend_shard(int field, size_t e, int device, int world, int rank, ecfft_ctx** out);
/* Sharded ENTER-ONLY context for ONE ENTER of n coefficients over `world` GPUs (ecfft_enter_sharded): the full chain T_1 .. T_c,
 * c = n / world, for the rank-local low levels, and for each of the log2(world) top levels only the rank's share of that tree —
 * the EXTEND tables of the split over its half-group and the c entries of xnn_s its combine step reads (src/fftree.rs:155-159).
 * All of it is pointwise in the point set: no tree above T_c is materialised on any GPU (~1/world of a full context's HBM), so
 * an ENTER can be larger than one GPU's table capacity.  Accepted by ecfft_enter_sharded only (same n, world and rank), plus
 * the informational calls listed above.  world = 2^k, 2 <= world <= 64, n / world >= 2 * world. */
int ecfft_build_enter_shard(int field, size_t n, int device, int world, int rank, ecfft_ctx** out);
/* Sharded EXIT-ONLY context for ONE EXIT of n evaluations over the ranks of `comm` (ecfft_exit_sharded) — a COLLECTIVE call: every
 * rank of the communicator makes it with the same field and n.  Holds the full chain T_1 .. T_c (c = n / world) and, for each of the
 * log2(world) top levels, only the rank's share of that tree: the EXTEND tables of the split over its group (both directions), its
 * entries of xnn_s, 1 / xnn_s, 1 / z0_s1 (pointwise in the point set) and of z0z0_rem_xnn_s — which is built distributed, level
 * by level, as the reference builds it (src/fftree.rs:418-452) but with the split EXIT's own operators and exchanges over `comm`.
 * No tree above T_c is materialised on any GPU — except T_2c for the redundant pair level, see ecfft_build_exit_shard_opts.
 * Accepted by ecfft_exit_sharded only (same n and communicator shape). */
int ecfft_build_exit_shard(int field, size_t n, int device, ecfft_comm* comm, ecfft_ctx** out);
/* ... with options.  The level of the PAIRS of ranks (blocks of 2c) has two forms: REDUNDANT — each rank also keeps the full tree
 * T_2c and both ranks of a pair run the level on the whole block from one exchange (1 exchange instead of 9, twice that level's
 * arithmetic; at world = 2 the context is then as large as a full one) — or SPLIT over the pair's shares, with no tree above T_c
 * anywhere.  Default (flags 0, = ecfft_build_exit_shard): redundant when T_2c fits the free memory of EVERY rank — the ranks agree
 * before anything is allocated — split otherwise.  ECFFT_EXIT_SHARD_MIN_MEMORY: always split (the largest n a node can reach). */
#define ECFFT_EXIT_SHARD_MIN_MEMORY 1
int ecfft_build_exit_shard_opts(int field, size_t n, int device, ecfft_comm* comm, int flags, ecfft_ctx** out);
/* ecfft_extend_sharded with a choice of distribution for the rank's shard on each side.  ECFFT_LAYOUT_CYCLIC: local element j'
 * is global position j' * world + rank.  A cyclic input saves the first of the four exchanges, a cyclic output the last one —
 * for hosts that chain split EXTENDs or that produce / consume the cyclic order anyway.  (BLOCK, BLOCK) == ecfft_extend_sharded. */
#define ECFFT_LAYOUT_BLOCK 0
#define ECFFT_LAYOUT_CYCLIC 1
int ecfft_extend_sharded_layout(ecfft_ctx* ctx, ecfft_comm* comm, const void* in, void* out, size_t e, int moiety, int in_layout,
                                int out_layout, void* stream);
/* Failure on ONE rank: the first time a context sees a sharded call of a given shape (op, size, layout, world) every rank
 * prepares its temporaries and the ranks AGREE on the outcome (one int each way, host wait) before the first exchange — if any
 * rank failed, all of them return ECFFT_ERR_HIP and none is left blocked in ncclRecv.  That first call is therefore synchronous;
 * later calls of the shape reuse the pinned temporaries, cannot fail locally and are asynchronous.  ecfft_build_exit_shard votes
 * after its local part, at every level and on its final status.  (Failure injection for the tests: ecfft_hip_hooks.h, test builds only.) */
int ecfft_enter_sharded(ecfft_ctx* ctx, ecfft_comm* comm, const void* coeffs, void* evals, size_t n, void* stream);
int ecfft_exit_sharded(ecfft_ctx* ctx, ecfft_comm* comm, const void* evals, void* coeffs, size_t n, void* stream);

/* Pointwise building block of the multi-GPU ENTER / EXIT (no reference counterpart): with T = table `which` (one of
 * ECFFT_TBL_XNN_S .. ECFFT_TBL_Z1Z1_REM_XNN_S) of the subtree with m leaves,
 *     mode 0: out[i] = x[i]*T[j]    1: x[i]*T[j] + y[i]    2: y[i] - x[i]*T[j]    3: (y[i] - x[i])*T[j],   j = t_off + i*t_stride.
 * These are the loops src/fftree.rs:155-159, 217-219, 238, 253-255, 279 restricted to an index range. */
int ecfft_table_fma(ecfft_ctx* ctx, void* out, const void* x, const void* y, size_t cnt, size_t m, int which, size_t t_off,
                    size_t t_stride, int mode, int mem, void* stream);

/* copy one table of the subtree with m leaves into host memory (element representation above);
 * returns the number of elements through *count; cap = capacity of host_out in elements. */
int ecfft_tree_table(ecfft_ctx* ctx, size_t m, int which, void* host_out, size_t cap, size_t* count);

/* FFTree wire format — impl CanonicalSerialize / CanonicalDeserialize for FFTree<F>, src/fftree.rs:507-660 (ark-serialize 0.4:
 * Vec = u64 LE length + elements, field element = standard-form integer LE, bool = 1 byte).  `compress` != 0 is
 * Compress::Yes: the three inverse tables are left out (:536-541) and regenerated on load (:620-628).
 *   ecfft_fftree_serialize   <-> FFTree::serialize_compressed / serialize_uncompressed + serialized_size (:556-590): *len receives
 *       the byte count; buf == NULL only asks for it; cap < *len is ECFFT_ERR_BAD_ARG.  Tables are copied out of HBM as they lie
 *       there (plain residues = the standard form).
 *   ecfft_fftree_deserialize <-> FFTree::deserialize_compressed / deserialize_uncompressed (:600-660): bounds-checked parse (a
 *       truncated, non-canonical or inconsistent file is ECFFT_ERR_BAD_ARG), then FFTree::new on the file's leaves and maps —
 *       every other table is recomputed on the GPU.  verify != 0 compares each table of the file with the recomputed one and
 *       rejects the file on a mismatch (the reference trusts the file: Valid::check is a no-op, :592-597).  verify == 0 still checks the
 *       internal layers of `f`; the file's OTHER tables are then neither used nor checked — a file whose tables disagree with its
 *       point set loads as the tree of its point set, where the reference would use the file's tables verbatim.  Bindings should
 *       default to verify = 1. */
int ecfft_fftree_serialize(ecfft_ctx* ctx, int compress, void* buf, size_t cap, size_t* len);
/* the pub field rational_maps (src/fftree.rs:28) of the top tree: log2(n) maps, 3 numerator + 3 denominator coefficients each
 * (low -> high, zero padded), element representation as everywhere; either output may be NULL */
int ecfft_tree_rational_maps(ecfft_ctx* ctx, void* map_num3_out, void* map_den3_out);
int ecfft_fftree_deserialize(int field, const void* bytes, size_t len, int compress, int device, int verify, ecfft_ctx** out);

/* Host-only front end of build_fftree (src/lib.rs:66-81) + the layer fill of FFTree::new
 * (src/fftree.rs:49-67): writes f (2n elements, heap order: f[n..2n) = leaves x(coset_offset + i*G))
 * and the log2(n) isogeny x-maps (3 + 3 coefficients each).  Needs no GPU; used to cross-check the
 * construction against an ark-built tree. */
int ecfft_build_points(int field, size_t n, void* f_out, void* map_num3_out, void* map_den3_out);

/* Per-launch timing for benchmarks (no reference counterpart): while enabled, every hot-path kernel
 * launch is bracketed by HIP events on its stream.  ecfft_profile_read synchronises the device and
 * returns, for kernel class `cls` (0 <= cls < ecfft_profile_classes()), its name, number of launches,
 * summed event time in ms and summed ALGORITHMIC bytes (stage-streaming model). */
int ecfft_profile_enable(ecfft_ctx* ctx, int on);
int ecfft_profile_classes(void);
int ecfft_profile_read(ecfft_ctx* ctx, int cls, char* name, size_t cap, uint64_t* launches, double* ms_total,
                       double* alg_bytes_total);

/* Element representation converters (host buffers, no GPU needed): the crate's in-memory form <-> the STANDARD-form
 * little-endian integer that ark-serialize writes (32 bytes for secp256k1, 4 for M31).  Used by the FFTree wire-format
 * reader/writer (ecfft_amd/serialize.py, reference: src/fftree.rs:507-660). */
int ecfft_elems_to_standard(int field, const void* in, void* out, size_t n);
int ecfft_elems_from_standard(int field, const void* in, void* out, size_t n);

/* Measurement hook: field multiplies per second of the butterfly kernels' table multiply run as a bare dependent chain
 * (x <- T*x + c per lane, `waves_per_simd` resident waves per SIMD, whole chip) — the VALU ceiling bench.py prices the hot
 * path against beside the HBM roofline. */
int ecfft_mul_ceiling(int field, int device, int waves_per_simd, double* mul_per_s);

/* Measurement hook: effective shader clock in MHz while every SIMD of the chip runs the kernels' table multiply (ratio of
 * s_memtime ticks to the constant 100 MHz wall clock inside that kernel) — the clock DVFS grants this instruction mix, needed
 * to turn rocprofv3 instruction counts into issue-cycle fractions. */
int ecfft_shader_clock(int field, int device, double* mhz);

/* Device-buffer helpers for hosts without HIP bindings (examples/sharded_extend.cpp is plain C++ over this ABI): allocate / free
 * HBM on `device`, wait for the device. */
int ecfft_device_alloc(int device, size_t bytes, void** out);
int ecfft_device_free(void* ptr);
int ecfft_device_sync(int device);

/* synchronous copy on the CURRENT device: kind 0 device -> host, 1 host -> device, 2 device -> device.  Lets a host language
 * without HIP bindings implement the exchange callback above — ecfft_amd/distributed.py does, over gloo. */
int ecfft_device_copy(void* dst, const void* src, size_t bytes, int kind);

/* library / device identification for logs: writes a NUL-terminated string */
int ecfft_device_info(int device, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
