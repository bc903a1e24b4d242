// ONE EXTEND / ENTER / EXIT split over the GPUs of a node from a plain C++ host — no Python, no HIP headers: only the C ABI of
// include/ecfft_hip.h (through the C++ mirror).  One process per GPU:
//     ECFFT_WORLD=8 ECFFT_RANK=r ECFFT_ID_FILE=/tmp/ecfft.id ./sharded_extend [log2 e]     (r = 0..7, device = r)
// Rank 0 creates the RCCL unique id (ecfft_comm_get_unique_id) and writes it to ECFFT_ID_FILE; the other ranks read it (any
// out-of-band channel would do).  Without the variables it runs with world = 1 — still through RCCL: communicator creation
// and all exchanges are real (self send / receive) — which is what tests/test_examples.py does on the one-GPU box.
// Every rank checks its shard against the single-GPU transform of the whole vector.
//   g++ -O2 -std=c++17 -Iinclude examples/sharded_extend.cpp -Lecfft_amd -lecfft_hip -Wl,-rpath,$PWD/ecfft_amd -o sharded_extend
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include "ecfft_fftree.hpp"

using namespace ecfft_host;
using Elem = Secp256k1Fp::Elem;

static int env_int(const char* k, int dflt) { const char* v = getenv(k); return v ? atoi(v) : dflt; }

struct DevBuf {                                   // HBM buffer through the ABI's helpers
    void* p = nullptr; int device;
    DevBuf(int dev, size_t bytes) : device(dev) { check(ecfft_device_alloc(dev, bytes, &p)); }
    ~DevBuf() { ecfft_device_free(p); }
    void put(const void* h, size_t bytes) { check(ecfft_device_copy(p, h, bytes, 1)); }
    void get(void* h, size_t bytes) const { check(ecfft_device_copy(h, p, bytes, 0)); }
};

int main(int argc, char** argv) {
    const int world = env_int("ECFFT_WORLD", 1), rank = env_int("ECFFT_RANK", 0), device = env_int("ECFFT_DEVICE", world > 1 ? rank : 0);
    const size_t e = (size_t)1 << (argc > 1 ? atoi(argv[1]) : 14), c = e / world;
    // --- communicator
    Comm::UniqueId id{};
    const char* idf = getenv("ECFFT_ID_FILE");
    if (rank == 0) {
        id = Comm::unique_id();
        if (world > 1 && idf) { FILE* f = fopen(idf, "wb"); fwrite(id.data(), 1, id.size(), f); fclose(f); }
    } else {
        for (int tries = 0;; ++tries) {
            FILE* f = idf ? fopen(idf, "rb") : nullptr;
            if (f && fread(id.data(), 1, id.size(), f) == id.size()) { fclose(f); break; }
            if (f) fclose(f);
            if (tries > 600) { printf("rank %d: no id file\n", rank); return 1; }
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
    }
    Comm comm = Comm::init_rank(id, world, rank, device);
    // --- tree (every rank builds the chain for its own GPU) and the same global input on every rank
    auto tree = FFTree<Secp256k1Fp>::build_fftree(2 * e, device);
    if (!tree) { printf("e exceeds the curve's 2-adicity\n"); return 1; }
    std::mt19937_64 rng(7);
    std::vector<Elem> x(e);
    for (auto& v : x) v = {rng(), rng(), rng(), rng() >> 1};
    bool ok = true;
    DevBuf in(device, c * sizeof(Elem)), out(device, c * sizeof(Elem));
    std::vector<Elem> got(c);
    auto shard_eq = [&](const std::vector<Elem>& want, const char* what) {
        out.get(got.data(), c * sizeof(Elem));
        bool good = memcmp(got.data(), want.data() + (size_t)rank * c, c * sizeof(Elem)) == 0;
        printf("rank %d/%d: %s %s\n", rank, world, what, good ? "== single-GPU result" : "FAILED");
        ok = ok && good;
    };
    in.put(x.data() + (size_t)rank * c, c * sizeof(Elem));
    for (Moiety m : {Moiety::S1, Moiety::S0}) {
        tree->extend_sharded(comm, (const Elem*)in.p, (Elem*)out.p, e, m, nullptr);
        check(ecfft_device_sync(device));
        shard_eq(tree->extend(x, m), m == Moiety::S1 ? "EXTEND -> S1" : "EXTEND -> S0");
    }
    {   // the same EXTEND on a sharded EXTEND-only context: this rank's 1/world share of the tables of T_2e and nothing else
        auto shard = FFTree<Secp256k1Fp>::build_extend_shard(e, world, rank, device);
        if (!shard) { printf("e exceeds the curve's 2-adicity\n"); return 1; }
        shard->extend_sharded(comm, (const Elem*)in.p, (Elem*)out.p, e, Moiety::S1, nullptr);
        check(ecfft_device_sync(device));
        shard_eq(tree->extend(x, Moiety::S1), "EXTEND -> S1 on a shard context");
        printf("rank %d/%d: tables %.1f MiB (shard context) vs %.1f MiB (full context)\n", rank, world, shard->device_bytes() / 1048576.0, tree->device_bytes() / 1048576.0);
    }
    if (world > 1) {   // ENTER / EXIT on sharded contexts (the EXIT one is a collective build over the communicator)
        auto esh = FFTree<Secp256k1Fp>::build_enter_shard(e, world, rank, device);
        auto xsh = FFTree<Secp256k1Fp>::build_exit_shard(e, comm, device);
        if (!esh || !xsh) { printf("shard context build failed\n"); return 1; }
        esh->enter_sharded(comm, (const Elem*)in.p, (Elem*)out.p, e, nullptr);
        check(ecfft_device_sync(device));
        shard_eq(tree->enter(x), "ENTER on a shard context");
        xsh->exit_sharded(comm, (const Elem*)in.p, (Elem*)out.p, e, nullptr);
        check(ecfft_device_sync(device));
        shard_eq(tree->exit(x), "EXIT on a shard context");
    }
    tree->enter_sharded(comm, (const Elem*)in.p, (Elem*)out.p, e, nullptr);
    check(ecfft_device_sync(device));
    shard_eq(tree->enter(x), "ENTER");
    tree->exit_sharded(comm, (const Elem*)in.p, (Elem*)out.p, e, nullptr);
    check(ecfft_device_sync(device));
    shard_eq(tree->exit(x), "EXIT");
    return ok ? 0 : 1;
}
