// TEST INFRASTRUCTURE (CPU only): the FFTree wire-format PARSER (ecfft_amd/csrc/wire_parse.h — the code ecfft_fftree_deserialize runs
// before anything touches the GPU) under AddressSanitizer + UBSan with seeded mutations of valid files.  Counterpart in the reference:
// impl CanonicalDeserialize for FFTree<F>, /root/reference/src/fftree.rs:600-660 (which trusts its input).
// usage: wire_fuzz FIELD(0|1) COMPRESS(0|1) VALID_FILE OTHER_VALID_FILE CASES
// Every mutated buffer is copied into an EXACT-SIZE heap allocation, so a read one byte past the file is an ASan error; the parser
// must return ECFFT_OK / ECFFT_ERR_BAD_ARG / ECFFT_ERR_NOT_POW2, and whenever it returns OK every table pointer must lie inside the
// buffer with its full length.
// build: g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -std=c++17 wire_fuzz.cpp -o wire_fuzz
#include "../../ecfft_amd/csrc/wire_parse.h"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

static std::vector<uint8_t> slurp(const char* path) {
    std::vector<uint8_t> v; FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    uint8_t buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f); return v;
}

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: wire_fuzz FIELD COMPRESS VALID OTHER CASES\n"); return 2; }
    const int field = atoi(argv[1]), compress = atoi(argv[2]);
    const std::vector<uint8_t> good = slurp(argv[3]), other = slurp(argv[4]);
    const long cases = atol(argv[5]);
    const size_t eb = ecfft::wire::elem_bytes(field);
    ecfft::wire::File file;
    if (ecfft::wire::parse(field, good.data(), good.size(), compress, file) != ECFFT_OK) { printf("FAIL: the valid file does not parse\n"); return 1; }
    const size_t top = file.levels[0].n;
    if (ecfft::wire::parse(field, good.data(), good.size(), !compress, file) == ECFFT_OK && top > 1) { printf("FAIL: a file parses in the wrong mode\n"); return 1; }
    std::mt19937_64 rng(0x5EED0F00ull + 2 * field + compress);
    long counts[6] = {0, 0, 0, 0, 0, 0};
    for (long c = 0; c < cases; ++c) {
        std::vector<uint8_t> b = good;
        const int reps = 1 + (int)(rng() % 3);                       // up to three stacked mutations
        for (int r = 0; r < reps; ++r) {
            const size_t n = b.size();
            if (n < 16) break;
            switch (rng() % 11) {
                case 0: for (int k = 0, m = 1 + (int)(rng() % 4); k < m; ++k) b[rng() % n] ^= (uint8_t)(1u << (rng() % 8)); break;
                case 1: { size_t i = rng() % n, m = 1 + rng() % 16; for (size_t k = i; k < i + m && k < n; ++k) b[k] = (uint8_t)rng(); } break;
                case 2: {                                            // a u64 anywhere (8-aligned or not) becomes a chosen length
                    static const uint64_t vals[] = {0, 1, 2, 3, 7, 8, 16, 17, 1ull << 20, (1ull << 61) + 1, 1ull << 63, ~0ull, 0x2000000000000000ull, 0x0800000000000001ull};
                    uint64_t v = rng() % 4 ? vals[rng() % (sizeof(vals) / sizeof(vals[0]))] : rng();
                    size_t i = rng() % 2 ? 8 * (rng() % (n / 8)) : rng() % (n - 8);
                    if (rng() % 3 == 0) i = 0;                        // the first prefix: the length of f
                    for (int k = 0; k < 8; ++k) b[i + k] = (uint8_t)(v >> (8 * k));
                } break;
                case 3: b.resize(rng() % n); break;                   // truncate
                case 4: { size_t i = rng() % (n + 1), m = 1 + rng() % 64; std::vector<uint8_t> ins(m); for (auto& x : ins) x = (uint8_t)rng(); b.insert(b.begin() + (long)i, ins.begin(), ins.end()); } break;
                case 5: { size_t i = rng() % n, m = 1 + rng() % 200; if (i + m > n) m = n - i; std::vector<uint8_t> d(b.begin() + (long)i, b.begin() + (long)(i + m)); b.insert(b.begin() + (long)i, d.begin(), d.end()); } break;
                case 6: { size_t ne = n / eb; if (ne > 2) { size_t i = eb * (rng() % (ne - 1)), j = eb * (rng() % (ne - 1)); for (size_t k = 0; k < eb; ++k) std::swap(b[i + k], b[j + k]); } } break;
                case 7: { size_t ne = (n - 8) / eb; if (ne > 1) { size_t i = 8 + eb * (rng() % (ne - 1)); uint8_t v = rng() % 2 ? 0xFF : 0x00; for (size_t k = 0; k < eb; ++k) b[i + k] = v; } } break;
                case 8: { size_t i = rng() % n; b.resize(i); size_t j = i < other.size() ? i : other.size() - 1; b.insert(b.end(), other.begin() + (long)j, other.end()); } break;
                case 9: b[n - 1 - rng() % (n < 10 ? n : 10)] = (uint8_t)rng(); break;
                default: { size_t i = rng() % n, m = 1 + rng() % 100; if (i + m > n) m = n - i; b.erase(b.begin() + (long)i, b.begin() + (long)(i + m)); } break;   // cut a range out
            }
        }
        // exact-size heap copy: one byte past the end is poisoned for ASan
        uint8_t* heap = (uint8_t*)malloc(b.size() ? b.size() : 1);
        if (b.size()) memcpy(heap, b.data(), b.size());
        const int rc = ecfft::wire::parse(field, heap, b.size(), compress, file);
        if (rc != ECFFT_OK && rc != ECFFT_ERR_BAD_ARG && rc != ECFFT_ERR_NOT_POW2) { printf("FAIL: case %ld returned %d\n", c, rc); return 1; }
        ++counts[rc];
        if (rc == ECFFT_OK) {
            for (const auto& lv : file.levels)
                for (int w = 0; w < 11; ++w)
                    if (lv.cnt[w]) {
                        if (lv.tbl[w] < heap || lv.tbl[w] + lv.cnt[w] * eb > heap + b.size()) { printf("FAIL: case %ld: table %d of level %zu points outside the file\n", c, w, lv.n); return 1; }
                        volatile uint8_t sink = 0;
                        for (size_t k = 0; k < lv.cnt[w] * eb; k += 7) sink ^= lv.tbl[w][k];       // touch it: ASan checks every byte we would hand to the GPU
                        sink ^= lv.tbl[w][lv.cnt[w] * eb - 1];
                        (void)sink;
                    }
            if (file.levels.empty() || file.maps.size() != ecfft::wire::ilog2_sz(file.levels[0].n)) { printf("FAIL: case %ld: accepted with inconsistent maps\n", c); return 1; }
        }
        free(heap);
    }
    printf("WIRE_FUZZ_OK field %d compress %d: %ld cases: %ld accepted, %ld bad arg, %ld not pow2\n", field, compress, cases, counts[ECFFT_OK], counts[ECFFT_ERR_BAD_ARG], counts[ECFFT_ERR_NOT_POW2]);
    return 0;
}
