// TEST INFRASTRUCTURE (CPU only, no GPU call): Transport::exchange_striped (ecfft_amd/csrc/transport.h) with W ranks as host threads
// and a transport that moves HOST bytes between them — grouped exchange semantics (all sends and receives of a call progress
// together, messages between one pair of ranks match in the order given).  For each of the patterns the split ENTER / EXIT use
// (level re-distribution, re-blocking, pair level, small-group all-to-all) and for random patterns: every rank's receive buffers
// must hold exactly what the plain exchange delivers, with a threshold of 0 (every eligible message striped) and with the default.
// build: g++ -O1 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include striping_host.cpp -L/opt/rocm/lib -lamdhip64 -lpthread
#include "../../ecfft_amd/csrc/transport.h"
#include <condition_variable>
#include <cstdint>
#include <random>
#include <thread>
using namespace ecfft;

struct Barrier {
    std::mutex m; std::condition_variable cv; int n, count = 0, gen = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait() { std::unique_lock<std::mutex> l(m); int g = gen; if (++count == n) { count = 0; ++gen; cv.notify_all(); } else cv.wait(l, [&] { return g != gen; }); }
};
struct Board { std::vector<std::vector<P2P>> sends; Barrier bar; std::atomic<int> calls{0}; explicit Board(int W) : sends((size_t)W), bar(W) {} };

class HostTransport : public Transport {
public:
    HostTransport(Board& b, int W, int r) : board(b) { world = W; rank = r; }
protected:
    bool do_exchange(const P2P* snd, int ns, const P2P* rcv, int nr, hipStream_t) override {
        board.sends[(size_t)rank].assign(snd, snd + ns);
        board.bar.wait();
        bool ok = true;
        for (int i = 0; i < nr; ++i) {                                  // my i-th receive from peer p = p's k-th send to me, k = number of earlier receives from p
            int p = rcv[i].peer, k = 0, seen = 0; const P2P* hit = nullptr;
            for (int j = 0; j < i; ++j) k += rcv[j].peer == p;
            for (const P2P& q : board.sends[(size_t)p]) if (q.peer == rank && seen++ == k) { hit = &q; break; }
            if (!hit || hit->bytes != rcv[i].bytes) { ok = false; continue; }
            memcpy(rcv[i].ptr, hit->ptr, rcv[i].bytes);
        }
        { int mine = 0; for (int p = 0; p < world; ++p) for (const P2P& q : board.sends[(size_t)p]) mine += q.peer == rank; if (mine != nr) ok = false; }
        board.bar.wait();
        if (rank == 0) ++board.calls;
        return ok;
    }
    Board& board;
};

typedef std::vector<std::vector<Transport::MsgDesc>> Pattern;     // [rank] -> its sends

static int run(int W, const Pattern& pat, size_t min_gain, const char* what, int expect_calls) {
    // buffers: message i of rank q = bytes filled with a hash of (q, i, offset)
    std::vector<std::vector<std::vector<uint8_t>>> sbuf((size_t)W), rbuf((size_t)W);
    std::vector<std::vector<P2P>> snd((size_t)W), rcv((size_t)W);
    for (int q = 0; q < W; ++q) {
        sbuf[(size_t)q].resize(pat[(size_t)q].size());
        for (size_t i = 0; i < pat[(size_t)q].size(); ++i) {
            auto& b = sbuf[(size_t)q][i]; b.resize(pat[(size_t)q][i].bytes);
            for (size_t o = 0; o < b.size(); ++o) b[o] = (uint8_t)((q * 131 + i * 31 + o * 7 + (o >> 8)) & 0xFF);
        }
    }
    for (int d = 0; d < W; ++d)
        for (int q = 0; q < W; ++q)
            for (size_t i = 0; i < pat[(size_t)q].size(); ++i) if (pat[(size_t)q][i].dst == d) rbuf[(size_t)d].emplace_back(pat[(size_t)q][i].bytes, (uint8_t)0xEE);
    for (int q = 0; q < W; ++q) {
        for (size_t i = 0; i < pat[(size_t)q].size(); ++i) snd[(size_t)q].push_back({pat[(size_t)q][i].dst, sbuf[(size_t)q][i].data(), pat[(size_t)q][i].bytes});
        size_t k = 0;
        for (int p = 0; p < W; ++p) for (size_t i = 0; i < pat[(size_t)p].size(); ++i) if (pat[(size_t)p][i].dst == q) { rcv[(size_t)q].push_back({p, rbuf[(size_t)q][k].data(), pat[(size_t)p][i].bytes}); ++k; }
    }
    Board board(W);
    std::vector<int> ok((size_t)W, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < W; ++r) th.emplace_back([&, r] {
        HostTransport t(board, W, r);
        t.stripe_min_gain = min_gain;
        std::vector<uint8_t> stage((size_t)64 << 20);
        auto fn = [&](int q, std::vector<Transport::MsgDesc>& m) { m = pat[(size_t)q]; };
        ok[(size_t)r] = t.exchange_striped(fn, snd[(size_t)r].data(), (int)snd[(size_t)r].size(), rcv[(size_t)r].data(), (int)rcv[(size_t)r].size(), stage.data(), stage.size(), nullptr);
    });
    for (auto& t : th) t.join();
    int bad = 0;
    for (int r = 0; r < W; ++r) if (!ok[(size_t)r]) ++bad;
    for (int d = 0; d < W; ++d) {
        size_t k = 0;
        for (int p = 0; p < W; ++p) for (size_t i = 0; i < pat[(size_t)p].size(); ++i) if (pat[(size_t)p][i].dst == d) { if (rbuf[(size_t)d][k] != sbuf[(size_t)p][i]) ++bad; ++k; }
    }
    const int calls = board.calls.load();
    if (expect_calls && calls != expect_calls) { printf("FAIL %s (W = %d): %d grouped exchanges, expected %d\n", what, W, calls, expect_calls); return 1; }
    printf("%s %s (W = %d, min gain %zu): %d grouped exchange(s)\n", bad ? "FAIL" : "ok  ", what, W, min_gain, calls);
    return bad ? 1 : 0;
}

int main() {
    int fails = 0;
    const size_t M = (size_t)1 << 20;
    for (int W : {4, 8}) {
        Pattern enter((size_t)W), reblock((size_t)W), pair((size_t)W), group((size_t)W), small((size_t)W);
        const int Q = W, half = Q / 2;
        for (int q = 0; q < W; ++q) {
            const int b = (q / Q) * Q, a = q - b, ap = a % half;
            enter[(size_t)q] = {{b + 2 * ap, 8 * M}, {b + 2 * ap + 1, 8 * M}};            // api_enter_split: cur / ext shares
            reblock[(size_t)q] = {{b + a / 2, 4 * M}, {b + half + a / 2, 4 * M}};         // api_exit_split: u0 / v0 shares
            pair[(size_t)q] = {{q ^ 1, 4 * M}, {q ^ 1, 4 * M}};                           // pair level: both halves to the partner
            const int gb = (q / 2) * 2;
            group[(size_t)q] = {{gb, 4 * M}, {gb + 1, 4 * M}};                            // all-to-all inside groups of two (one piece to itself)
            small[(size_t)q] = {{q ^ 1, 4096}, {(q + 1) % W, 100}};                       // too small / not divisible: never striped
        }
        // two messages to two peers: per phase a link carries 2 x 1/W of a message — lighter than the direct link only for W > 4
        fails += run(W, enter, 0, "ENTER re-distribution", W > 4 ? 2 : 1) + run(W, reblock, 0, "EXIT re-blocking", W > 4 ? 2 : 1) + run(W, pair, 0, "pair level", 2) + run(W, group, 0, "groups of two", 2);
        fails += run(W, small, 0, "small messages", 1);
        fails += run(W, pair, (size_t)4 << 20, "pair level (2 x 4 MiB), default threshold", 2);       // 8 MiB on one link against 2 x 8/W MiB per phase: striped
        Pattern pair2((size_t)W);
        for (int q = 0; q < W; ++q) pair2[(size_t)q] = {{q ^ 1, 2 * M}, {q ^ 1, 2 * M}};
        fails += run(W, pair2, (size_t)4 << 20, "pair level (2 x 2 MiB), default threshold", 1);       // 4 MiB on one link: the gain stays below 4 MiB, direct
        std::mt19937 rng(1234 + W);
        for (int t = 0; t < 20; ++t) {                                                               // random patterns, mixed sizes, self messages
            Pattern p((size_t)W);
            for (int q = 0; q < W; ++q) { int nm = rng() % 4; for (int i = 0; i < nm; ++i) p[(size_t)q].push_back({(int)(rng() % W), (size_t)(((rng() % 3) == 0 ? 1000 + rng() % 5000 : (size_t)(1 + rng() % 6) * 16 * W * 512))}); }
            fails += run(W, p, 0, "random pattern", 0);
        }
    }
    printf(fails ? "STRIPING_HOST_FAILED\n" : "STRIPING_HOST_OK\n");
    return fails ? 1 : 0;
}
