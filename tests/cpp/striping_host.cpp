// TEST INFRASTRUCTURE (CPU only, no GPU call): Transport::exchange_striped (ecfft_amd/csrc/transport.h) with W ranks as host threads
// and a transport that moves HOST bytes between them — grouped exchange semantics (all sends and receives of a call progress
// together, messages between one pair of ranks match in the order given).  For each of the patterns the split ENTER / EXIT use
// (level re-distribution, re-blocking, pair level, small-group all-to-all) and for random patterns: every rank's receive buffers
// must hold exactly what the plain exchange delivers, with a threshold of 0 (every eligible message striped) and with 4 MiB.
// Round 6: built with -fsanitize=address,undefined by tests/test_striping_host.py and driven with seeded RANDOM patterns at
// W = 4, 8, 16, 64 — zero-length messages, sizes that do not divide by 16 W, self messages, several messages per pair, ranks that send
// or receive nothing — plus the error path: a pattern that disagrees with the call's sends / receives must FAIL on that rank.
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
            if (rcv[i].bytes) memcpy(rcv[i].ptr, hit->ptr, rcv[i].bytes);       // (a zero-length message of an empty vector has a null pointer)
        }
        { int mine = 0; for (int p = 0; p < world; ++p) for (const P2P& q : board.sends[(size_t)p]) mine += q.peer == rank; if (mine != nr) ok = false; }
        board.bar.wait();
        if (rank == 0) ++board.calls;
        return ok;
    }
    Board& board;
};

typedef std::vector<std::vector<Transport::MsgDesc>> Pattern;     // [rank] -> its sends

static size_t g_stage_bytes = (size_t)64 << 20;
static bool g_quiet = false;
static int run(int W, const Pattern& pat, size_t min_gain, const char* what, int expect_calls, int* calls_out = nullptr) {
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
        std::vector<uint8_t> stage(g_stage_bytes);
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
    if (calls_out) *calls_out = calls;
    if (expect_calls && calls != expect_calls) { printf("FAIL %s (W = %d): %d grouped exchanges, expected %d\n", what, W, calls, expect_calls); return 1; }
    if (bad || !g_quiet) printf("%s %s (W = %d, min gain %zu): %d grouped exchange(s)\n", bad ? "FAIL" : "ok  ", what, W, min_gain, calls);
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
        fails += run(W, pair, (size_t)4 << 20, "pair level (2 x 4 MiB), 4 MiB threshold", 2);       // 8 MiB on one link against 2 x 8/W MiB per phase: striped
        Pattern pair2((size_t)W);
        for (int q = 0; q < W; ++q) pair2[(size_t)q] = {{q ^ 1, 2 * M}, {q ^ 1, 2 * M}};
        fails += run(W, pair2, (size_t)4 << 20, "pair level (2 x 2 MiB), 4 MiB threshold", 1);       // 4 MiB on one link: the gain stays below 4 MiB, direct
        std::mt19937 rng(1234 + W);
        for (int t = 0; t < 20; ++t) {                                                               // random patterns, mixed sizes, self messages
            Pattern p((size_t)W);
            for (int q = 0; q < W; ++q) { int nm = rng() % 4; for (int i = 0; i < nm; ++i) p[(size_t)q].push_back({(int)(rng() % W), (size_t)(((rng() % 3) == 0 ? 1000 + rng() % 5000 : (size_t)(1 + rng() % 6) * 16 * W * 512))}); }
            fails += run(W, p, 0, "random pattern", 0);
        }
    }
    // ---- round 6: seeded random patterns at W = 4 .. 64 (sanitizer build) ------------------------------------------------------
    g_quiet = true;
    for (int W : {4, 8, 16, 64}) {
        g_stage_bytes = (size_t)4 << 20;
        std::mt19937 rng(977u * (unsigned)W + 6u);
        const int rounds = W == 64 ? 8 : 40;
        int striped = 0, plain = 0;
        for (int t = 0; t < rounds; ++t) {
            Pattern p((size_t)W);
            const unsigned shape = rng() % 4;      // 0: sparse, 1: pairwise big, 2: dense small, 3: mixed
            for (int q = 0; q < W; ++q) {
                int nm = shape == 2 ? (int)(rng() % 6) : (int)(rng() % 3);
                if (shape == 1) nm = 1 + (int)(rng() % 2);
                if (rng() % 11 == 0) nm = 0;                                      // a rank that sends nothing
                for (int i = 0; i < nm; ++i) {
                    int dst = shape == 1 ? (q ^ 1) : (int)(rng() % (unsigned)W);
                    if (rng() % 13 == 0) dst = q;                                  // self message
                    size_t b;
                    switch (shape == 2 ? rng() % 4 : rng() % 6) {                         // dense small: nothing eligible, never striped
                        case 0: b = 0; break;                                      // zero-length
                        case 1: b = 1 + rng() % 4000; break;                       // tiny, odd
                        case 2: b = ((size_t)64 << 10) + 1 + rng() % 1000; break;  // above the 64 KiB floor but not a multiple of 16 W
                        case 3: b = (size_t)16 * (size_t)W * (64 + rng() % 64) - 16; break;   // multiple of 16, not of 16 W
                        default: b = (size_t)16 * (size_t)W * (size_t)((W == 64 ? 64 : 256) + rng() % (W == 64 ? 64 : 1024)); break;   // eligible: >= 64 KiB, multiple of 16 W
                    }
                    p[(size_t)q].push_back({dst, b});
                }
            }
            int calls = 0;
            fails += run(W, p, 0, "random pattern (threshold 0)", 0, &calls);
            (calls == 2 ? striped : plain)++;
            if (calls != 1 && calls != 2) { printf("FAIL random pattern (W = %d): %d grouped exchanges\n", W, calls); ++fails; }
            fails += run(W, p, (size_t)1 << 20, "random pattern (threshold 1 MiB)", 0, &calls);
            fails += run(W, p, ~(size_t)0, "random pattern (striping off)", 1, &calls);
        }
        printf("ok   %d random patterns at W = %d: %d striped, %d direct, all delivered\n", rounds, W, striped, plain);
        if (!striped || !plain) { printf("FAIL random patterns at W = %d never took %s path\n", W, striped ? "the direct" : "the striped"); ++fails; }
    }
    // ---- the error path: a pattern that disagrees with what the rank really posts is an ERROR on that rank, not a silent
    // plain exchange next to peers that stripe.  One thread, no peers needed: the check precedes every exchange. ------------------------
    {
        struct Dead : Transport { Dead() { world = 4; rank = 1; } bool do_exchange(const P2P*, int, const P2P*, int, hipStream_t) override { ++n; return true; } int n = 0; } t;
        t.stripe_min_gain = 0;
        std::vector<uint8_t> a((size_t)1 << 20), b((size_t)1 << 20), stage((size_t)4 << 20);
        P2P snd[1] = {{0, a.data(), a.size()}}, rcv[1] = {{0, b.data(), b.size()}};
        auto good = [&](int q, std::vector<Transport::MsgDesc>& m) { m.push_back({q ^ 1, (size_t)1 << 20}); };
        auto wrong_size = [&](int q, std::vector<Transport::MsgDesc>& m) { m.push_back({q ^ 1, (size_t)(q == 1 ? 2 : 1) << 20}); };
        auto wrong_peer = [&](int q, std::vector<Transport::MsgDesc>& m) { m.push_back({q == 1 ? 2 : (q ^ 1), (size_t)1 << 20}); };
        auto wrong_count = [&](int q, std::vector<Transport::MsgDesc>& m) { m.push_back({q ^ 1, (size_t)1 << 20}); if (q == 1) m.push_back({0, 16}); };
        auto missing_recv = [&](int q, std::vector<Transport::MsgDesc>& m) { if (q != 0) m.push_back({q ^ 1, (size_t)1 << 20}); };
        auto out_of_range = [&](int q, std::vector<Transport::MsgDesc>& m) { m.push_back({q == 3 ? 7 : (q ^ 1), (size_t)1 << 20}); };
        fprintf(stderr, "(the five diagnostics below are the expected ones of the error-path checks)\n");
        const bool g = t.exchange_striped(good, snd, 1, rcv, 1, stage.data(), stage.size(), nullptr);
        const int n_good = t.n;
        const bool e1 = t.exchange_striped(wrong_size, snd, 1, rcv, 1, stage.data(), stage.size(), nullptr);
        const bool e2 = t.exchange_striped(wrong_peer, snd, 1, rcv, 1, stage.data(), stage.size(), nullptr);
        const bool e3 = t.exchange_striped(wrong_count, snd, 1, rcv, 1, stage.data(), stage.size(), nullptr);
        const bool e4 = t.exchange_striped(missing_recv, snd, 1, rcv, 1, stage.data(), stage.size(), nullptr);
        const bool e5 = t.exchange_striped(out_of_range, snd, 1, rcv, 1, stage.data(), stage.size(), nullptr);
        if (!g || n_good != 2) { printf("FAIL error path: the consistent pattern did not stripe (ok %d, %d exchanges)\n", (int)g, n_good); ++fails; }
        if (e1 || e2 || e3 || e4 || e5 || t.n != n_good) { printf("FAIL error path: an inconsistent pattern went through (%d %d %d %d %d, %d exchanges issued)\n", e1, e2, e3, e4, e5, t.n - n_good); ++fails; }
        else printf("ok   inconsistent patterns (size, peer, count, missing receive, rank out of range) fail without issuing an exchange\n");
    }
    printf(fails ? "STRIPING_HOST_FAILED\n" : "STRIPING_HOST_OK\n");
    return fails ? 1 : 0;
}
