/* TEST INFRASTRUCTURE — CPU oracle, not product code.
 *
 * Mersenne-31 instance of the oracle: `impl FftreeField for m31::Fp` (/root/reference/src/lib.rs:198-215)
 * through build_ec_fftree (src/ec.rs:498-554), Velu 2-isogenies of short Weierstrass curves
 * (src/ec.rs:209-259) and root finding of the 2-torsion cubic (src/utils.rs:25-44; the reference
 * factors with distinct/equal-degree factorisation and then SORTS the roots, so any correct root
 * finder followed by a sort gives the same list).
 */
#include "field_m31.h"
#define ORA(name) ora_m31_##name
#include "fftree_generic.h"

/* ---- tiny dense polynomials, degree <= 6, coefficients low -> high ---- */
typedef struct { fe c[8]; int deg; } spoly; /* deg = -1 for the zero polynomial */

static void sp_norm(spoly* p) { while (p->deg >= 0 && fe_is_zero(p->c[p->deg])) p->deg--; }
static spoly sp_zero(void) { spoly p; memset(&p, 0, sizeof p); p.deg = -1; return p; }
static spoly sp_rem(spoly a, const spoly* m) { /* a mod m, m != 0 */
    fe lead_inv = fe_inv(m->c[m->deg]);
    while (a.deg >= m->deg) {
        fe q = fe_mul(a.c[a.deg], lead_inv);
        int sh = a.deg - m->deg;
        for (int i = 0; i <= m->deg; ++i) a.c[i + sh] = fe_sub(a.c[i + sh], fe_mul(q, m->c[i]));
        sp_norm(&a);
    }
    return a;
}
static spoly sp_mulmod(const spoly* a, const spoly* b, const spoly* m) {
    spoly r = sp_zero();
    if (a->deg < 0 || b->deg < 0) return r;
    for (int i = 0; i <= a->deg; ++i)
        for (int j = 0; j <= b->deg; ++j) r.c[i + j] = fe_add(r.c[i + j], fe_mul(a->c[i], b->c[j]));
    r.deg = a->deg + b->deg; sp_norm(&r);
    return sp_rem(r, m);
}
static spoly sp_powmod(spoly base, uint64_t e, const spoly* m) {
    spoly r = sp_zero(); r.c[0] = fe_one(); r.deg = 0; r = sp_rem(r, m);
    base = sp_rem(base, m);
    while (e) { if (e & 1) r = sp_mulmod(&r, &base, m); base = sp_mulmod(&base, &base, m); e >>= 1; }
    return r;
}
static spoly sp_monic(spoly a) {
    if (a.deg < 0) return a;
    fe li = fe_inv(a.c[a.deg]);
    for (int i = 0; i <= a.deg; ++i) a.c[i] = fe_mul(a.c[i], li);
    return a;
}
static spoly sp_gcd(spoly a, spoly b) { /* monic gcd (src/utils.rs:132-141) */
    while (b.deg >= 0) { spoly r = sp_rem(a, &b); a = b; b = r; }
    return sp_monic(a);
}
static spoly sp_div_exact(spoly a, const spoly* b) { /* quotient of a / b */
    spoly q = sp_zero(); fe li = fe_inv(b->c[b->deg]);
    q.deg = a.deg - b->deg;
    while (a.deg >= b->deg) {
        fe t = fe_mul(a.c[a.deg], li); int sh = a.deg - b->deg; q.c[sh] = t;
        for (int i = 0; i <= b->deg; ++i) a.c[i + sh] = fe_sub(a.c[i + sh], fe_mul(t, b->c[i]));
        a.c[a.deg] = 0; sp_norm(&a);
    }
    return q;
}
/* roots of a monic product of distinct linear factors (equal-degree splitting, deterministic shifts) */
static void split_roots(spoly g, fe* roots, int* nroots) {
    if (g.deg <= 0) return;
    if (g.deg == 1) { roots[(*nroots)++] = fe_neg(fe_mul(g.c[0], fe_inv(g.c[1]))); return; }
    for (uint32_t shift = 1;; ++shift) {
        spoly h = sp_zero(); h.c[0] = shift; h.c[1] = 1; h.deg = 1;
        spoly w = sp_powmod(h, (M31_P - 1) / 2, &g);
        w.c[0] = fe_sub(w.c[0], fe_one()); if (w.deg < 0) w.deg = 0; sp_norm(&w);
        if (w.deg < 0) continue;
        spoly d = sp_gcd(g, w);
        if (d.deg > 0 && d.deg < g.deg) {
            spoly q = sp_monic(sp_div_exact(g, &d));
            split_roots(d, roots, nroots); split_roots(q, roots, nroots);
            return;
        }
    }
}
/* find_roots (src/utils.rs:25-44) for a monic cubic x^3 + c2 x^2 + c1 x + c0; output sorted ascending */
int ORA(find_roots_cubic)(fe c0, fe c1, fe c2, fe* roots) {
    spoly f = sp_zero(); f.c[0] = c0; f.c[1] = c1; f.c[2] = c2; f.c[3] = 1; f.deg = 3;
    /* square-free part (src/utils.rs:118-127): f / gcd(f, f') */
    spoly fp = sp_zero(); fp.c[0] = c1; fp.c[1] = fe_dbl(c2); fp.c[2] = 3; fp.deg = 2; sp_norm(&fp);
    if (fp.deg >= 0) { spoly g = sp_gcd(f, fp); if (g.deg > 0) f = sp_monic(sp_div_exact(f, &g)); }
    spoly x = sp_zero(); x.c[1] = 1; x.deg = 1;
    spoly xp = sp_powmod(x, M31_P, &f);
    spoly diff = xp; diff.c[1] = fe_sub(diff.c[1], fe_one()); if (diff.deg < 1) diff.deg = 1; sp_norm(&diff);
    spoly g = (diff.deg < 0) ? f : sp_gcd(f, diff);
    int n = 0; split_roots(g, roots, &n);
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) if (roots[j] < roots[i]) { fe t = roots[i]; roots[i] = roots[j]; roots[j] = t; }
    return n;
}

typedef struct { fe a, b; } swcurve; /* y^2 = x^3 + a x + b (src/ec.rs:204-207) */
static wcurve swcurve_w(const swcurve* c) { wcurve w; w.a1 = w.a2 = w.a3 = 0; w.a4 = c->a; w.a6 = c->b; return w; }

/* ShortWeierstrassCurve::two_isogenies (src/ec.rs:214-242), one candidate per 2-torsion root x0 */
static void velu2(const swcurve* c, fe x0, swcurve* codomain, ratmap* r, fe hnum[3], fe hden[3]) {
    fe t = fe_add(fe_mul(fe_mul(3, x0), x0), c->a);
    codomain->a = fe_sub(c->a, fe_mul(5, t));
    codomain->b = fe_sub(c->b, fe_mul(fe_mul(7, x0), t));
    r->num[0] = t; r->num[1] = fe_neg(x0); r->num[2] = 1; r->nnum = 3;
    r->den[0] = fe_neg(x0); r->den[1] = 1; r->den[2] = 0; r->nden = 2;
    fe x0x0 = fe_sqr(x0), m2x0 = fe_neg(fe_add(x0, x0));
    hnum[0] = fe_sub(x0x0, t); hnum[1] = m2x0; hnum[2] = 1;
    hden[0] = x0x0; hden[1] = m2x0; hden[2] = 1;
}
static ecpoint velu2_map(const ratmap* r, const fe hnum[3], const fe hden[3], ecpoint p) { /* src/ec.rs:344-358 */
    ecpoint q; q.inf = 1; q.x = q.y = 0;
    if (p.inf) return q;
    fe rx; if (!ratmap_map(r, p.x, &rx)) return q;
    fe hd = poly_eval(hden, 3, p.x); if (fe_is_zero(hd)) return q;
    fe hx = fe_mul(poly_eval(hnum, 3, p.x), fe_inv(hd));
    q.inf = 0; q.x = rx; q.y = fe_mul(hx, p.y);
    return q;
}

/* m31::Fp::build_fftree (src/lib.rs:199-214) -> build_ec_fftree (src/ec.rs:498-554) */
static void* build_impl(unsigned log_n, int check_chain, int extend_only) {
    (void)check_chain; /* the two-adicity test is part of the algorithm here (src/ec.rs:534) */
    swcurve curve = {1, 0};
    ecpoint offset = {1048755163u, 279503108u, 0}, gen = {1273083559u, 804329170u, 0};
    const unsigned two_adicity = 28;
    if (log_n >= 32) return NULL;                              /* assert!(log_n < 32) (:510) */
    if (log_n > two_adicity) return NULL;                      /* :513-515 */
    wcurve w = swcurve_w(&curve);
    for (unsigned i = 0; i < two_adicity - log_n; ++i) gen = ec_add(&w, gen, gen); /* :518-521 */
    ratmap* maps = (ratmap*)calloc(log_n ? log_n : 1, sizeof(ratmap));
    swcurve cur = curve; ecpoint g = gen;
    for (unsigned k = 0; k < log_n; ++k) {                     /* :526-543 */
        fe roots[3]; int nr = ORA(find_roots_cubic)(cur.b, cur.a, 0, roots);
        wcurve wc = swcurve_w(&cur);
        int tg = ec_two_adicity(&wc, g), found = 0;
        for (int i = 0; i < nr && !found; ++i) {
            swcurve cod; ratmap r; fe hn[3], hd[3];
            velu2(&cur, roots[i], &cod, &r, hn, hd);
            ecpoint gp = velu2_map(&r, hn, hd, g);
            wcurve wn = swcurve_w(&cod);
            int tp = ec_two_adicity(&wn, gp);
            if (tg >= 0 && tp >= 0 && tg == tp + 1) { maps[k] = r; cur = cod; g = gp; found = 1; }
        }
        if (!found) { fprintf(stderr, "cannot find a suitable isogeny\n"); free(maps); return NULL; }
    }
    size_t n = (size_t)1 << log_n;
    fe* leaves = fe_alloc(n);
    ec_leaves(&w, offset, gen, leaves, n);                     /* :546-551 */
    ora_extend_only = extend_only;
    fftree* t = tree_new(leaves, n, maps, (int)log_n);
    ora_extend_only = 0;
    free(maps); fe_free(leaves);
    return t;
}
void* ORA(build_fftree)(unsigned log_n, int check_chain) { return build_impl(log_n, check_chain, 0); }
/* TEST INFRASTRUCTURE: the tree with 2^log_n leaves holding only what FFTree::extend of 2^(log_n - 1) evaluations reads (fftree_generic.h,
 * ora_extend_only); every other call on it fails or crashes — the Python wrapper exposes extend only */
void* ORA(build_extend_tree)(unsigned log_n) { return build_impl(log_n, 0, 1); }
/* leaves idx[0..k) of the n = 2^log_n point set of build_fftree above (src/lib.rs:201-206, src/ec.rs:518-521, 545-551) */
int ORA(leaves_at)(unsigned log_n, const uint64_t* idx, size_t k, void* out) {
    swcurve curve = {1, 0};
    ecpoint offset = {1048755163u, 279503108u, 0}, gen = {1273083559u, 804329170u, 0};
    const unsigned two_adicity = 28;
    if (log_n > two_adicity) return -1;
    wcurve w = swcurve_w(&curve);
    for (unsigned i = 0; i < two_adicity - log_n; ++i) gen = ec_add(&w, gen, gen);
    ec_leaves_at(&w, offset, gen, idx, k, (fe*)out);
    return 0;
}
void ORA(from_std)(const void* in, void* out, size_t n) { memcpy(out, in, n * sizeof(fe)); }
void ORA(to_std)(const void* in, void* out, size_t n) { memcpy(out, in, n * sizeof(fe)); }
