// C++ counterpart of the reference's criterion bench (/root/reference/benches/fftree.rs:19-62): the same eight algorithms
// at the same size (input length 2048 on a 4096-leaf tree), both fields, through the C ABI with HOST buffers — what a Rust
// caller of the drop-in would see, staging included.  Prints the median of 10 samples like criterion's sample_size(10).
//   g++ -O2 -std=c++17 -Iinclude examples/bench_fftree.cpp -Lecfft_amd -lecfft_hip -Wl,-rpath,$PWD/ecfft_amd -o bench_fftree
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <functional>
#include <random>
#include "ecfft_fftree.hpp"

using namespace ecfft_host;
using Clock = std::chrono::steady_clock;

static double median_ms(const std::function<void()>& f) {
    f();                                                     // warm-up (criterion warms up too)
    std::vector<double> t;
    for (int i = 0; i < 10; ++i) {
        auto t0 = Clock::now(); f();
        t.push_back(std::chrono::duration<double, std::milli>(Clock::now() - t0).count());
    }
    std::sort(t.begin(), t.end());
    return 0.5 * (t[4] + t[5]);
}

template <class F>
static typename F::Elem random_elem(std::mt19937_64& rng);
template <>
Secp256k1Fp::Elem random_elem<Secp256k1Fp>(std::mt19937_64& rng) { return {rng(), rng(), rng(), rng() >> 1}; }   // < 2^255 < p
template <>
M31Fp::Elem random_elem<M31Fp>(std::mt19937_64& rng) { return (M31Fp::Elem)(rng() % 0x7FFFFFFFu); }

template <class F>
static int bench_field(const char* description) {
    const size_t n = 2048;                                   // BENCHMARK_INPUT_SIZES (:14)
    std::mt19937_64 rng(1);
    std::vector<typename F::Elem> vals(n);
    for (auto& v : vals) v = random_elem<F>(rng);
    auto tree = FFTree<F>::build_fftree(n * 2);              // :26
    if (!tree) { printf("tree too large\n"); return 1; }
    auto xnn = tree->table(ECFFT_TBL_XNN_S, n), c = tree->table(ECFFT_TBL_Z0Z0_REM_XNN_S, n);
    std::vector<typename F::Elem> half(vals.begin(), vals.begin() + n / 2);
    printf("ECFFT algorithms (%s), n = %zu\n", description, n);
    printf("  ENTER   %8.3f ms\n", median_ms([&] { tree->enter(vals); }));
    printf("  EXIT    %8.3f ms\n", median_ms([&] { tree->exit(vals); }));
    printf("  DEGREE  %8.3f ms\n", median_ms([&] { tree->degree(vals); }));
    printf("  EXTEND  %8.3f ms\n", median_ms([&] { tree->extend(vals, Moiety::S1); }));
    printf("  MEXTEND %8.3f ms\n", median_ms([&] { tree->mextend(vals, Moiety::S1); }));
    printf("  MOD     %8.3f ms\n", median_ms([&] { tree->modular_reduce(vals, xnn, c); }));
    printf("  REDC    %8.3f ms\n", median_ms([&] { tree->redc_z0(vals, xnn); }));
    printf("  VANISH  %8.3f ms\n", median_ms([&] { tree->vanish(half); }));
    // the round trip the reference's tests assert
    if (tree->exit(tree->enter(vals)) != vals) { printf("round trip FAILED\n"); return 1; }
    return 0;
}

int main() {
    int rc = bench_field<M31Fp>("31 bit Mersenne prime field");
    rc |= bench_field<Secp256k1Fp>("secp256k1's prime field");
    return rc;
}
