// Host-side construction of the FFTree's point sets: the leaves x(coset_offset + i*G), the isogeny
// x-maps and the layers L_{k+1} = psi_k(L_k).  This is the `FftreeField::build_fftree` front end
// (/root/reference/src/lib.rs:39-85 for secp256k1, src/lib.rs:198-215 + src/ec.rs:498-554 for M31)
// and `FFTree::new` (src/fftree.rs:42-70).  Construction only — O(n) field operations, done once;
// the O(n log^2 n) table precompute that follows runs on the GPU (device_tree.h).  For ecfft_build_fftree and the shard
// builders only the O(log n) part runs here (curve, generator, isogeny maps); leaves and layers are then computed by the same
// formulas on the GPU (DeviceChain::points_on_device).  The full host path below serves ecfft_build_points (hosts without a
// GPU, the CPU tests that pin it to the oracle and the golden vectors) and is what the GPU point set is compared with.
//
// Everything here is in PLAIN (non-Montgomery) form; see field_secp256k1.h.  Sequential affine
// additions with one inversion each (reference: src/lib.rs:73-78) are replaced by log(n) rounds of
// batched additions sharing one inversion per round — the resulting x-coordinates are the same
// field elements.
#pragma once
#include <array>
#include <vector>
#include <algorithm>
#include <cstdio>
#include "field_secp256k1.h"
#include "field_m31.h"

namespace ecfft {

template <class F>
struct RatMap {  // numerator / denominator coefficients, low -> high (src/utils.rs:367-371)
    typename F::elem num[3], den[3];
};

// affine points on y^2 = x^3 + a2 x^2 + a4 x + a6 (a1 = a3 = 0; src/ec.rs:376-424)
template <class F>
struct Pt { typename F::elem x, y; bool inf; };

template <class F>
struct Curve { typename F::elem a2, a4, a6; };

template <class F>
struct HostTree {  // what FFTree::new derives before from_tree: f layers (heap order, 2n) + maps
    size_t n = 0;
    std::vector<typename F::elem> f;          // empty when the point set is left to the GPU (have_gen)
    std::vector<RatMap<F>> maps;
    // what the leaves are generated from: leaf i = x(off + i*gen) on `curve` — the device front end (device_tree.h
    // points_on_device) recomputes leaves and layers from these instead of uploading them
    bool have_gen = false;
    Curve<F> curve{}; Pt<F> off{}, gen{};
    bool leaves_only = false;                 // f holds the leaves (f[n..2n)) only: the layers are computed on the GPU (FFTree::new)
};

template <class F>
static void batch_inverse(std::vector<typename F::elem>& v) {  // all entries non-zero
    using E = typename F::elem;
    size_t n = v.size();
    if (!n) return;
    std::vector<E> pre(n);
    E acc = F::one();
    for (size_t i = 0; i < n; ++i) { pre[i] = acc; acc = F::mul(acc, v[i]); }
    acc = F::inv(acc);
    for (size_t i = n; i-- > 0;) { E t = F::mul(acc, pre[i]); acc = F::mul(acc, v[i]); v[i] = t; }
}

template <class F>
static typename F::elem poly3(const typename F::elem c[3], const typename F::elem& x) {
    return F::mul_add(F::mul_add(c[2], x, c[1]), x, c[0]);
}

// FFTree::new (src/fftree.rs:42-70): fill the inner layers of f from the leaves
template <class F>
static bool fill_layers(HostTree<F>& t) {
    using E = typename F::elem;
    size_t n = t.n;
    for (size_t k = 0, sz = n; sz > 1; ++k, sz >>= 1) {
        const E* prev = t.f.data() + sz;
        size_t half = sz / 2;
        E* layer = t.f.data() + half;
        std::vector<E> den(half);
        for (size_t j = 0; j < half; ++j) {
            den[j] = poly3<F>(t.maps[k].den, prev[j]);
            if (F::is_zero(den[j])) return false;
        }
        batch_inverse<F>(den);
        for (size_t j = 0; j < half; ++j) layer[j] = F::mul(poly3<F>(t.maps[k].num, prev[j]), den[j]);
    }
    return true;
}

template <class F>
static Pt<F> pt_add(const Curve<F>& c, const Pt<F>& p, const Pt<F>& q) {
    using E = typename F::elem;
    if (p.inf) return q;
    if (q.inf) return p;
    E lambda;
    if (F::eq(p.x, q.x)) {
        if (F::is_zero(F::add(p.y, q.y))) { Pt<F> r; r.inf = true; r.x = r.y = F::zero(); return r; }
        E xx = F::sqr(p.x);
        E num = F::add(F::add(F::add(xx, xx), xx), F::add(F::mul(F::add(c.a2, c.a2), p.x), c.a4));
        lambda = F::mul(num, F::inv(F::add(p.y, p.y)));
    } else {
        lambda = F::mul(F::sub(q.y, p.y), F::inv(F::sub(q.x, p.x)));
    }
    Pt<F> r; r.inf = false;
    r.x = F::sub(F::sub(F::sub(F::sqr(lambda), c.a2), p.x), q.x);
    r.y = F::sub(F::mul(lambda, F::sub(p.x, r.x)), p.y);
    return r;
}
template <class F>
static int pt_two_adicity(const Curve<F>& c, Pt<F> p) {  // src/utils.rs:356-365
    for (int i = 0; i < 64; ++i) { if (p.inf) return i; p = pt_add(c, p, p); }
    return -1;
}

// leaves[i] = x(offset + i*gen), i < n  (src/lib.rs:72-78 / src/ec.rs:545-551)
template <class F>
static void compute_leaves(const Curve<F>& c, const Pt<F>& offset, const Pt<F>& gen, size_t n, typename F::elem* leaves) {
    using E = typename F::elem;
    std::vector<E> px(n), py(n);   // i*gen for i >= 1
    if (n > 1) { px[1] = gen.x; py[1] = gen.y; }
    for (size_t r = 1; 2 * r <= n && r < n; r <<= 1) {
        // P_{2r} by doubling, then P_{r+j} = P_r + P_j for 0 < j < r with one shared inversion
        size_t cnt = r - 1;
        std::vector<E> den(cnt);
        for (size_t j = 1; j < r; ++j) den[j - 1] = F::sub(px[j], px[r]);
        batch_inverse<F>(den);
        for (size_t j = 1; j < r; ++j) {
            E lambda = F::mul(F::sub(py[j], py[r]), den[j - 1]);
            E x3 = F::sub(F::sub(F::sub(F::sqr(lambda), c.a2), px[r]), px[j]);
            px[r + j] = x3;
            py[r + j] = F::sub(F::mul(lambda, F::sub(px[r], x3)), py[r]);
        }
        if (2 * r < n) {
            Pt<F> pr{px[r], py[r], false};
            Pt<F> d = pt_add(c, pr, pr);
            px[2 * r] = d.x; py[2 * r] = d.y;
        }
    }
    leaves[0] = offset.x;
    if (n > 1) {
        std::vector<E> den(n - 1);
        for (size_t i = 1; i < n; ++i) den[i - 1] = F::sub(px[i], offset.x);
        batch_inverse<F>(den);
        for (size_t i = 1; i < n; ++i) {
            E lambda = F::mul(F::sub(py[i], offset.y), den[i - 1]);
            leaves[i] = F::sub(F::sub(F::sub(F::sqr(lambda), c.a2), offset.x), px[i]);
        }
    }
}

// ---- secp256k1: good curve + good isogeny chain (src/ec.rs:38-45, 61-90, 177-189) ----
// returns 0 ok, 1 = n too large for the curve's 2-adicity (build_fftree -> None), 2 = internal error
// points = false: only the maps and the generator data (the GPU computes leaves and layers)
static inline int build_secp256k1(unsigned log_n, HostTree<Secp256k1>& t, bool points = true) {
    using F = Secp256k1; using E = F::elem;
    const unsigned two_adicity = 36;
    if (log_n >= two_adicity) return 1;                         // src/lib.rs:62-64
    E a = F::from_dec("31172306031375832341232376275243462303334845584808513005362718476441963632613");
    E bb = F::from_dec("45508371059383884471556188660911097844526467659576498497548207627741160623272");
    E b; if (!F::sqrt(bb, &b)) return 2;                        // GoodCurve::new_odd
    Curve<F> c{a, bb, F::zero()};
    Pt<F> off{F::from_dec("105623886150579165427389078198493427091405550492761682382732004625374789850161"),
              F::from_dec("7709812624542158994629670452026922591039826164720902911013234773380889499231"), false};
    Pt<F> gen{F::from_dec("41293412487153066667050767300223451435019201659857889215769525847559135483332"),
              F::from_dec("73754924733368840065089190002333366411120578552679996887076912271884749237510"), false};
    for (unsigned i = 0; i < two_adicity - log_n; ++i) gen = pt_add(c, gen, gen);   // src/lib.rs:67-70
    size_t n = (size_t)1 << log_n;
    t.n = n; t.maps.resize(log_n);
    t.have_gen = true; t.curve = c; t.off = off; t.gen = gen;
    if (points) { t.f.assign(2 * n, F::zero()); compute_leaves<F>(c, off, gen, n, t.f.data() + n); }
    for (unsigned k = 0; k < log_n; ++k) {                      // good_isogeny: psi(x) = (x - b)^2 / x
        E b2 = F::sqr(b);
        RatMap<F>& m = t.maps[k];
        m.num[0] = b2; m.num[1] = F::neg(F::add(b, b)); m.num[2] = F::one();
        m.den[0] = F::zero(); m.den[1] = F::one(); m.den[2] = F::zero();
        E a_next = F::add(a, F::add(F::add(F::add(b, b), F::add(b, b)), F::add(b, b)));      // a + 6b
        E ab = F::mul(a, b);
        E B_next = F::add(F::add(F::add(ab, ab), F::add(ab, ab)), F::add(F::add(F::add(b2, b2), F::add(b2, b2)), F::add(F::add(b2, b2), F::add(b2, b2))));  // 4ab + 8b^2
        E b_next; if (!F::sqrt(B_next, &b_next)) return 2;
        a = a_next; b = b_next;
    }
    if (!points) return 0;
    return fill_layers<F>(t) ? 0 : 2;
}

// ---- Mersenne-31: Velu 2-isogenies of short Weierstrass curves (src/ec.rs:209-259, 498-554) ----
namespace m31detail {
struct Poly { uint32_t c[8]; int deg; };   // small dense polynomial over M31, deg = -1 for zero
static inline void norm(Poly& p) { while (p.deg >= 0 && p.c[p.deg] == 0) --p.deg; }
static inline Poly rem(Poly a, const Poly& m) {
    uint32_t li = M31::inv(m.c[m.deg]);
    while (a.deg >= m.deg) {
        uint32_t q = M31::mul(a.c[a.deg], li); int sh = a.deg - m.deg;
        for (int i = 0; i <= m.deg; ++i) a.c[i + sh] = M31::sub(a.c[i + sh], M31::mul(q, m.c[i]));
        norm(a);
    }
    return a;
}
static inline Poly mulmod(const Poly& a, const Poly& b, const Poly& m) {
    Poly r{}; r.deg = -1;
    if (a.deg < 0 || b.deg < 0) return r;
    for (int i = 0; i <= a.deg; ++i) for (int j = 0; j <= b.deg; ++j) r.c[i + j] = M31::add(r.c[i + j], M31::mul(a.c[i], b.c[j]));
    r.deg = a.deg + b.deg; norm(r);
    return rem(r, m);
}
static inline Poly powmod(Poly base, uint64_t e, const Poly& m) {
    Poly r{}; r.c[0] = 1; r.deg = 0; r = rem(r, m); base = rem(base, m);
    for (; e; e >>= 1) { if (e & 1) r = mulmod(r, base, m); base = mulmod(base, base, m); }
    return r;
}
static inline Poly monic(Poly a) { if (a.deg >= 0) { uint32_t li = M31::inv(a.c[a.deg]); for (int i = 0; i <= a.deg; ++i) a.c[i] = M31::mul(a.c[i], li); } return a; }
static inline Poly gcd(Poly a, Poly b) { while (b.deg >= 0) { Poly r = rem(a, b); a = b; b = r; } return monic(a); }
static inline Poly quo(Poly a, const Poly& b) {
    Poly q{}; q.deg = a.deg - b.deg; uint32_t li = M31::inv(b.c[b.deg]);
    while (a.deg >= b.deg) {
        uint32_t t = M31::mul(a.c[a.deg], li); int sh = a.deg - b.deg; q.c[sh] = t;
        for (int i = 0; i <= b.deg; ++i) a.c[i + sh] = M31::sub(a.c[i + sh], M31::mul(t, b.c[i]));
        norm(a);
    }
    return q;
}
static inline void linear_roots(const Poly& g, std::vector<uint32_t>& out) {
    if (g.deg <= 0) return;
    if (g.deg == 1) { out.push_back(M31::neg(M31::mul(g.c[0], M31::inv(g.c[1])))); return; }
    for (uint32_t s = 1;; ++s) {
        Poly h{}; h.c[0] = s; h.c[1] = 1; h.deg = 1;
        Poly w = powmod(h, (M31::P - 1) / 2, g);
        if (w.deg < 0) w.deg = 0;
        w.c[0] = M31::sub(w.c[0], 1); norm(w);
        if (w.deg < 0) continue;
        Poly d = gcd(g, w);
        if (d.deg > 0 && d.deg < g.deg) { linear_roots(d, out); linear_roots(monic(quo(g, d)), out); return; }
    }
}
// roots in F_p of x^3 + a x + b, ascending (find_roots + sort, src/utils.rs:25-44)
static inline std::vector<uint32_t> cubic_roots(uint32_t a, uint32_t b) {
    Poly f{}; f.c[0] = b; f.c[1] = a; f.c[2] = 0; f.c[3] = 1; f.deg = 3;
    Poly x{}; x.c[1] = 1; x.deg = 1;
    Poly xp = powmod(x, M31::P, f);
    if (xp.deg < 1) xp.deg = 1;
    xp.c[1] = M31::sub(xp.c[1], 1); norm(xp);
    Poly g = xp.deg < 0 ? f : gcd(f, xp);
    std::vector<uint32_t> r; linear_roots(g, r); std::sort(r.begin(), r.end());
    return r;
}
}  // namespace m31detail

static inline int build_m31(unsigned log_n, HostTree<M31>& t, bool points = true) {
    using F = M31;
    const unsigned two_adicity = 28;
    if (log_n > two_adicity) return 1;                          // src/ec.rs:513-515
    uint32_t ca = 1, cb = 0;                                    // y^2 = x^3 + x  (src/lib.rs:201)
    Curve<F> c{0, ca, cb};
    Pt<F> off{1048755163u, 279503108u, false}, gen{1273083559u, 804329170u, false};
    for (unsigned i = 0; i < two_adicity - log_n; ++i) gen = pt_add(c, gen, gen);
    size_t n = (size_t)1 << log_n;
    t.n = n; t.maps.resize(log_n);
    t.have_gen = true; t.curve = c; t.off = off; t.gen = gen;
    Pt<F> g = gen; Curve<F> cur = c;
    for (unsigned k = 0; k < log_n; ++k) {                      // src/ec.rs:526-543
        std::vector<uint32_t> roots = m31detail::cubic_roots(cur.a4, cur.a6);
        int tg = pt_two_adicity(cur, g); bool found = false;
        for (uint32_t x0 : roots) {
            uint32_t tt = F::add(F::mul(F::mul(3, x0), x0), cur.a4);          // t = 3 x0^2 + a
            Curve<F> cod{0, F::sub(cur.a4, F::mul(5, tt)), F::sub(cur.a6, F::mul(F::mul(7, x0), tt))};
            RatMap<F> m;
            m.num[0] = tt; m.num[1] = F::neg(x0); m.num[2] = 1;
            m.den[0] = F::neg(x0); m.den[1] = 1; m.den[2] = 0;
            // phi(g) = (r(x), h(x) y), h = ((x - x0)^2 - t)/(x - x0)^2
            uint32_t dx = F::sub(g.x, x0);
            Pt<F> gp{0, 0, true};                                // vanishing denominator -> Point::zero (src/ec.rs:354-357)
            if (dx != 0) {
                uint32_t dx2 = F::sqr(dx), idx2 = F::inv(dx2);
                gp = Pt<F>{F::mul(poly3<F>(m.num, g.x), F::inv(dx)), F::mul(F::mul(F::sub(dx2, tt), idx2), g.y), false};
            }
            int tp = pt_two_adicity(cod, gp);
            if (tg >= 0 && tp >= 0 && tg == tp + 1) { t.maps[k] = m; cur = cod; g = gp; found = true; break; }
        }
        if (!found) { fprintf(stderr, "ecfft: cannot find a suitable isogeny\n"); return 2; }
    }
    if (!points) return 0;
    t.f.assign(2 * n, 0);
    compute_leaves<F>(c, off, gen, n, t.f.data() + n);
    return fill_layers<F>(t) ? 0 : 2;
}

template <class F> int build_host_tree(unsigned log_n, HostTree<F>& t, bool points = true);
template <> inline int build_host_tree<Secp256k1>(unsigned log_n, HostTree<Secp256k1>& t, bool points) { return build_secp256k1(log_n, t, points); }
template <> inline int build_host_tree<M31>(unsigned log_n, HostTree<M31>& t, bool points) { return build_m31(log_n, t, points); }

}  // namespace ecfft
