// ecfft_fftree.hpp — C++ host-side mirror of the reference's `FFTree<F>` / `FftreeField` surface
// (/root/reference/src/lib.rs:14-16, src/fftree.rs:17-38, 42, 123, 164, 227, 489) over the C ABI of
// ecfft_hip.h.  Header-only; link with libecfft_hip.so.  Same names, argument meaning and error
// behaviour: where the Rust code panics this throws (std::invalid_argument for non powers of two,
// std::length_error("FFTree is too small")), `build_fftree` returns std::nullopt where Rust returns None.
#pragma once
#include <array>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>
#include "ecfft_hip.h"

namespace ecfft_host {

enum class Moiety { S0 = ECFFT_S0, S1 = ECFFT_S1 };                       // src/fftree.rs:17-21

struct Secp256k1Fp { static constexpr int id = ECFFT_FIELD_SECP256K1; using Elem = std::array<uint64_t, 4>; };  // Montgomery limbs
struct M31Fp { static constexpr int id = ECFFT_FIELD_M31; using Elem = uint32_t; };

inline void check(int rc) {
    switch (rc) {
        case ECFFT_OK: return;
        case ECFFT_ERR_NOT_POW2: throw std::invalid_argument("length must be a power of two");
        case ECFFT_ERR_TREE_TOO_SMALL: throw std::length_error("FFTree is too small");
        case ECFFT_ERR_HIP: throw std::runtime_error("ecfft: HIP failure (no usable device?) - there is no CPU fallback");
        default: throw std::runtime_error("ecfft: error " + std::to_string(rc));
    }
}

inline unsigned log2_floor(size_t n) { unsigned l = 0; while (n > 1) { n >>= 1; ++l; } return l; }
inline void require(bool cond, const char* what) { if (!cond) throw std::invalid_argument(what); }

// `ecfft_comm`: inter-GPU transport of the sharded transforms, one process per GPU (RCCL underneath, ecfft_hip.h).
class Comm {
public:
    using UniqueId = std::array<unsigned char, ECFFT_COMM_ID_BYTES>;
    Comm(const Comm&) = delete;
    Comm& operator=(const Comm&) = delete;
    Comm(Comm&& o) noexcept : c_(o.c_) { o.c_ = nullptr; }
    ~Comm() { if (c_) ecfft_comm_destroy(c_); }
    // rank 0 creates the id and hands it to the other ranks by any means (MPI, a file, a socket)
    static UniqueId unique_id() { UniqueId id{}; check(ecfft_comm_get_unique_id(id.data())); return id; }
    static Comm init_rank(const UniqueId& id, int world, int rank, int device) {
        ecfft_comm* c = nullptr;
        check(ecfft_comm_init_rank(id.data(), world, rank, device, &c));
        return Comm(c);
    }
    // the RCCL library the first communicator of the process binds (ecfft_comm_set_rccl_library); "" / nullptr = the default
    static void set_rccl_library(const char* path) { check(ecfft_comm_set_rccl_library(path)); }
    // ncclCommAbort: unblocks exchanges in flight (from another host thread); later sharded calls on this communicator fail
    bool abort() { return ecfft_comm_abort(c_) == ECFFT_OK; }
    // threshold of the link striping of the big pairwise exchanges (ecfft_comm_set_link_striping); the same value on every rank
    void set_link_striping(size_t min_gain_bytes) { check(ecfft_comm_set_link_striping(c_, min_gain_bytes)); }
    int rank() const { return ecfft_comm_rank(c_); }
    int world() const { return ecfft_comm_world(c_); }
    ecfft_comm* raw() const { return c_; }
private:
    explicit Comm(ecfft_comm* c) : c_(c) {}
    ecfft_comm* c_;
};

template <class F>
class FFTree {
public:
    using Elem = typename F::Elem;
    FFTree(const FFTree&) = delete;
    FFTree& operator=(const FFTree&) = delete;
    FFTree(FFTree&& o) noexcept : ctx_(o.ctx_) { o.ctx_ = nullptr; }
    ~FFTree() { if (ctx_) ecfft_ctx_destroy(ctx_); }

    // FftreeField::build_fftree (src/lib.rs:14-16)
    static std::optional<FFTree> build_fftree(size_t n, int device = 0) {
        ecfft_ctx* c = nullptr;
        int rc = ecfft_build_fftree(F::id, n, device, &c);
        if (rc == ECFFT_ERR_TREE_TOO_LARGE) return std::nullopt;
        check(rc);
        return FFTree(c);
    }
    // sharded EXTEND-only context (ecfft_build_extend_shard): this rank's share of the tables of ONE EXTEND of e evaluations
    // over `world` GPUs; only extend_sharded works on it
    static std::optional<FFTree> build_extend_shard(size_t e, int world, int rank, int device = 0) {
        ecfft_ctx* c = nullptr;
        int rc = ecfft_build_extend_shard(F::id, e, device, world, rank, &c);
        if (rc == ECFFT_ERR_TREE_TOO_LARGE) return std::nullopt;
        check(rc);
        return FFTree(c);
    }
    // sharded ENTER-only context (ecfft_build_enter_shard): only enter_sharded works on it
    static std::optional<FFTree> build_enter_shard(size_t n, int world, int rank, int device = 0) {
        ecfft_ctx* c = nullptr;
        int rc = ecfft_build_enter_shard(F::id, n, device, world, rank, &c);
        if (rc == ECFFT_ERR_TREE_TOO_LARGE) return std::nullopt;
        check(rc);
        return FFTree(c);
    }
    // sharded EXIT-only context (ecfft_build_exit_shard), a COLLECTIVE build over `comm`: only exit_sharded works on it
    // min_memory: never keep the full tree T_2n/world for the redundant pair level (ECFFT_EXIT_SHARD_MIN_MEMORY)
    static std::optional<FFTree> build_exit_shard(size_t n, const Comm& comm, int device = 0, bool min_memory = false) {
        ecfft_ctx* c = nullptr;
        int rc = ecfft_build_exit_shard_opts(F::id, n, device, comm.raw(), min_memory ? ECFFT_EXIT_SHARD_MIN_MEMORY : 0, &c);
        if (rc == ECFFT_ERR_TREE_TOO_LARGE) return std::nullopt;
        check(rc);
        return FFTree(c);
    }
    // FFTree::new (src/fftree.rs:42-70): maps as 3 numerator + 3 denominator coefficients each
    static FFTree from_leaves(const std::vector<Elem>& leaves, const std::vector<Elem>& num3, const std::vector<Elem>& den3, int device = 0) {
        // the C ABI reads 3*log2(n) numerator and denominator coefficients: a short vector would be a host out-of-bounds read
        require(!leaves.empty() && (leaves.size() & (leaves.size() - 1)) == 0, "leaves: length must be a power of two");
        require(num3.size() == 3 * (size_t)log2_floor(leaves.size()) && den3.size() == num3.size(), "rational maps: need 3*log2(n) numerator and denominator coefficients");
        ecfft_ctx* c = nullptr;
        check(ecfft_fftree_new(F::id, leaves.data(), leaves.size(), num3.data(), den3.data(), device, &c));
        return FFTree(c);
    }
    size_t size() const { return ecfft_tree_size(ctx_); }

    std::vector<Elem> enter(const std::vector<Elem>& coeffs) const {            // src/fftree.rs:164-167
        std::vector<Elem> out(coeffs.size());
        check(ecfft_enter(ctx_, coeffs.data(), out.data(), coeffs.size(), ECFFT_MEM_HOST, nullptr));
        return out;
    }
    std::vector<Elem> exit(const std::vector<Elem>& evals) const {              // src/fftree.rs:227-230
        std::vector<Elem> out(evals.size());
        check(ecfft_exit(ctx_, evals.data(), out.data(), evals.size(), ECFFT_MEM_HOST, nullptr));
        return out;
    }
    std::vector<Elem> extend(const std::vector<Elem>& evals, Moiety moiety) const {   // src/fftree.rs:123-126
        std::vector<Elem> out(evals.size());
        check(ecfft_extend(ctx_, evals.data(), out.data(), evals.size(), (int)moiety, 1, ECFFT_MEM_HOST, nullptr));
        return out;
    }
    // the remaining algorithms (src/fftree.rs:138-141, 195-198, 264-275, 286-289, 313-316)
    std::vector<Elem> mextend(const std::vector<Elem>& evals, Moiety moiety) const {
        std::vector<Elem> out(evals.size());
        check(ecfft_mextend(ctx_, evals.data(), out.data(), evals.size(), (int)moiety, 1, ECFFT_MEM_HOST, nullptr));
        return out;
    }
    std::vector<Elem> redc_z0(const std::vector<Elem>& evals, const std::vector<Elem>& a) const { return redc(evals, a, Moiety::S0); }
    std::vector<Elem> redc_z1(const std::vector<Elem>& evals, const std::vector<Elem>& a) const { return redc(evals, a, Moiety::S1); }
    std::vector<Elem> modular_reduce(const std::vector<Elem>& evals, const std::vector<Elem>& a, const std::vector<Elem>& c) const {
        require(a.size() == evals.size() && c.size() == evals.size(), "modular_reduce: a and c must have evals.len() entries");
        std::vector<Elem> out(evals.size());
        check(ecfft_modular_reduce(ctx_, evals.data(), a.data(), c.data(), out.data(), evals.size(), ECFFT_MEM_HOST, nullptr));
        return out;
    }
    std::vector<Elem> vanish(const std::vector<Elem>& domain) const {
        std::vector<Elem> out(2 * domain.size());
        check(ecfft_vanish(ctx_, domain.data(), out.data(), domain.size(), ECFFT_MEM_HOST, nullptr));
        return out;
    }
    size_t degree(const std::vector<Elem>& evals) const {
        size_t d = 0;
        check(ecfft_degree(ctx_, evals.data(), evals.size(), ECFFT_MEM_HOST, nullptr, &d));
        return d;
    }
    size_t device_bytes() const { return ecfft_ctx_device_bytes(ctx_); }     // HBM held between calls: tables + scratch
    // device-resident variants (pointers into HBM, caller's stream)
    void enter_device(const Elem* coeffs, Elem* evals, size_t n, void* stream) const { check(ecfft_enter(ctx_, coeffs, evals, n, ECFFT_MEM_DEVICE, stream)); }
    void exit_device(const Elem* evals, Elem* coeffs, size_t n, void* stream) const { check(ecfft_exit(ctx_, evals, coeffs, n, ECFFT_MEM_DEVICE, stream)); }
    void extend_device(const Elem* in, Elem* out, size_t e, Moiety m, size_t count, void* stream) const { check(ecfft_extend(ctx_, in, out, e, (int)m, count, ECFFT_MEM_DEVICE, stream)); }

    // ONE transform split over the ranks of `comm` (device pointers: this rank's block shard of len / world elements)
    void extend_sharded(const Comm& comm, const Elem* in, Elem* out, size_t e, Moiety m, void* stream) const { check(ecfft_extend_sharded(ctx_, comm.raw(), in, out, e, (int)m, stream)); }
    // the same with a CYCLIC shard (local j' = global j' * world + rank) on either side: one exchange fewer per cyclic side
    void extend_sharded_layout(const Comm& comm, const Elem* in, Elem* out, size_t e, Moiety m, bool cyclic_in, bool cyclic_out, void* stream) const {
        check(ecfft_extend_sharded_layout(ctx_, comm.raw(), in, out, e, (int)m, cyclic_in ? ECFFT_LAYOUT_CYCLIC : ECFFT_LAYOUT_BLOCK,
                                          cyclic_out ? ECFFT_LAYOUT_CYCLIC : ECFFT_LAYOUT_BLOCK, stream));
    }
    void enter_sharded(const Comm& comm, const Elem* coeffs, Elem* evals, size_t n, void* stream) const { check(ecfft_enter_sharded(ctx_, comm.raw(), coeffs, evals, n, stream)); }
    void exit_sharded(const Comm& comm, const Elem* evals, Elem* coeffs, size_t n, void* stream) const { check(ecfft_exit_sharded(ctx_, comm.raw(), evals, coeffs, n, stream)); }

    // pub tables of the subtree with m leaves (src/fftree.rs:24-38, 489-496)
    std::vector<Elem> table(int which, size_t m) const {
        size_t cnt = 0;
        check(ecfft_tree_table(ctx_, m, which, nullptr, 0, &cnt));
        std::vector<Elem> out(cnt);
        check(ecfft_tree_table(ctx_, m, which, out.data(), cnt, &cnt));
        return out;
    }
    ecfft_ctx* raw() const { return ctx_; }

private:
    std::vector<Elem> redc(const std::vector<Elem>& evals, const std::vector<Elem>& a, Moiety m) const {
        require(a.size() == evals.size(), "redc: a must have evals.len() entries");
        std::vector<Elem> out(evals.size());
        check(ecfft_redc(ctx_, evals.data(), a.data(), out.data(), evals.size(), (int)m, ECFFT_MEM_HOST, nullptr));
        return out;
    }
    explicit FFTree(ecfft_ctx* c) : ctx_(c) {}
    ecfft_ctx* ctx_;
};

}  // namespace ecfft_host
