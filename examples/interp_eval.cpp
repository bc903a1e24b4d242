// C++ counterpart of the reference's examples/interp_eval.rs (/root/reference/examples/interp_eval.rs:9-34):
// build the secp256k1 FFTree for n = 2^14, ENTER a polynomial, EXIT it again, check the round trip and
// print wall times.  (The reference's naive O(n^2) evaluation is the job of tests/, which have the oracle.)
//   g++ -O2 -std=c++17 -Iinclude examples/interp_eval.cpp -Lecfft_amd -lecfft_hip -Wl,-rpath,$PWD/ecfft_amd -o interp_eval
#include <chrono>
#include <cstdio>
#include <random>
#include "ecfft_fftree.hpp"

using namespace ecfft_host;
using Clock = std::chrono::steady_clock;

int main(int argc, char** argv) {
    size_t n = (size_t)1 << (argc > 1 ? atoi(argv[1]) : 14);
    auto t0 = Clock::now();
    auto tree = FFTree<Secp256k1Fp>::build_fftree(n);
    if (!tree) { printf("n exceeds the curve's 2-adicity\n"); return 1; }
    printf("FFTree generation time: %.3f s\n", std::chrono::duration<double>(Clock::now() - t0).count());

    std::mt19937_64 rng(1);
    std::vector<Secp256k1Fp::Elem> coeffs(n);
    for (auto& c : coeffs) { c = {rng(), rng(), rng(), rng() >> 1}; }      // < 2^255 < p: valid Montgomery-form residues

    t0 = Clock::now();
    auto evals = tree->enter(coeffs);
    printf("evaluation time (fft): %.3f ms\n", std::chrono::duration<double, std::milli>(Clock::now() - t0).count());
    t0 = Clock::now();
    auto back = tree->exit(evals);
    printf("interpolation time (ifft): %.3f ms\n", std::chrono::duration<double, std::milli>(Clock::now() - t0).count());
    if (back != coeffs) { printf("round trip FAILED\n"); return 1; }
    printf("round trip ok (assert_eq!(coeffs, exit(enter(coeffs))))\n");
    return 0;
}
