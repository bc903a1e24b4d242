/* TEST INFRASTRUCTURE — CPU oracle, not product code.
 *
 * secp256k1 instance of the oracle: `impl FftreeField for Fp` (/root/reference/src/lib.rs:39-85)
 * on top of the GoodCurve / good-isogeny layer (src/ec.rs:24-91, 177-189).
 */
#include "field_secp256k1.h"
#define ORA(name) ora_secp_##name
#include "fftree_generic.h"


typedef struct { fe a, b; } goodcurve; /* GoodCurve::Odd { a, b }: y^2 = x^3 + a x^2 + b^2 x (src/ec.rs:28-35) */

/* GoodCurve::new_odd (src/ec.rs:38-45) */
static int goodcurve_new_odd(fe a, fe bb, goodcurve* out) {
    fe disc = fe_sub(fe_sqr(a), fe_dbl(fe_dbl(bb)));
    if (fe_is_zero(bb) || fe_is_zero(disc)) return 0;
    fe b; if (!fe_sqrt(bb, &b)) return 0;
    fe tmp; if (!fe_sqrt(fe_add(fe_add(a, b), b), &tmp)) return 0;
    out->a = a; out->b = b; return 1;
}
static wcurve goodcurve_w(const goodcurve* c) { /* WeierstrassCurve for GoodCurve::Odd (src/ec.rs:142-173) */
    wcurve w; w.a1 = fe_zero(); w.a2 = c->a; w.a3 = fe_zero(); w.a4 = fe_sqr(c->b); w.a6 = fe_zero(); return w;
}
/* GoodCurve::good_isogeny, odd case (src/ec.rs:75-88): r(x) = (x^2 - 2b x + b^2)/x,
 * h(x) = (x^2 - b^2)/x^2, codomain (a + 6b, 4ab + 8b^2) */
static int good_isogeny(const goodcurve* c, goodcurve* codomain, ratmap* r) {
    fe a = c->a, b = c->b, bb = fe_sqr(b);
    fe a_prime = fe_add(fe_add(a, fe_dbl(fe_dbl(b))), fe_dbl(b));
    fe b_prime = fe_add(fe_dbl(fe_dbl(fe_mul(a, b))), fe_dbl(fe_dbl(fe_dbl(bb))));
    if (!goodcurve_new_odd(a_prime, b_prime, codomain)) return 0;
    r->num[0] = bb; r->num[1] = fe_neg(fe_dbl(b)); r->num[2] = fe_one(); r->nnum = 3;
    r->den[0] = fe_zero(); r->den[1] = fe_one(); r->den[2] = fe_zero(); r->nden = 2;
    return 1;
}
/* Isogeny::map on a point (src/ec.rs:344-358) with g = 0, h = (x^2 - b^2)/x^2 */
static ecpoint good_isogeny_map(const goodcurve* c, const ratmap* r, ecpoint p) {
    ecpoint q; q.inf = 1; q.x = q.y = fe_zero();
    if (p.inf) return q;
    fe rx; if (!ratmap_map(r, p.x, &rx)) return q;
    fe xx = fe_sqr(p.x); if (fe_is_zero(xx)) return q;
    fe hx = fe_mul(fe_sub(xx, fe_sqr(c->b)), fe_inv(xx));
    q.inf = 0; q.x = rx; q.y = fe_mul(hx, p.y);
    return q;
}

/* decimal string -> field element */
static fe fe_from_dec(const char* s) {
    fe r = fe_zero(); fe ten = fe_from_u64(10);
    for (; *s; ++s) r = fe_add(fe_mul(r, ten), fe_from_u64((uint64_t)(*s - '0')));
    return r;
}

/* Fp::build_fftree (src/lib.rs:40-84). check_chain != 0 also runs the two-adicity assertion of
 * find_isogeny_chain (src/ec.rs:184), which costs O(log^2 n) point doublings. */
static void* build_impl(unsigned log_n, int check_chain, int extend_only) {

    goodcurve curve;
    if (!goodcurve_new_odd(
            fe_from_dec("31172306031375832341232376275243462303334845584808513005362718476441963632613"),
            fe_from_dec("45508371059383884471556188660911097844526467659576498497548207627741160623272"), &curve))
        return NULL;
    ecpoint offset, gen; offset.inf = gen.inf = 0;
    offset.x = fe_from_dec("105623886150579165427389078198493427091405550492761682382732004625374789850161");
    offset.y = fe_from_dec("7709812624542158994629670452026922591039826164720902911013234773380889499231");
    gen.x = fe_from_dec("41293412487153066667050767300223451435019201659857889215769525847559135483332");
    gen.y = fe_from_dec("73754924733368840065089190002333366411120578552679996887076912271884749237510");
    const unsigned two_adicity = 36;
    if (log_n >= two_adicity) return NULL;                      /* :62-64 */
    wcurve w = goodcurve_w(&curve);
    for (unsigned i = 0; i < two_adicity - log_n; ++i) gen = ec_add(&w, gen, gen); /* :67-70 */
    size_t n = (size_t)1 << log_n;
    fe* leaves = fe_alloc(n);
    ec_leaves(&w, offset, gen, leaves, n);                      /* :73-78 */
    /* find_isogeny_chain (src/ec.rs:177-189) */
    ratmap* maps = (ratmap*)calloc(log_n ? log_n : 1, sizeof(ratmap));
    goodcurve cur = curve; ecpoint g = gen;
    for (unsigned k = 0; k < log_n; ++k) {
        goodcurve next;
        if (!good_isogeny(&cur, &next, &maps[k])) { free(maps); fe_free(leaves); return NULL; }
        if (check_chain) {
            wcurve wc = goodcurve_w(&cur), wn = goodcurve_w(&next);
            ecpoint gp = good_isogeny_map(&cur, &maps[k], g);
            int t0 = ec_two_adicity(&wc, g), t1 = ec_two_adicity(&wn, gp);
            if (t0 != t1 + 1) { fprintf(stderr, "isogeny chain check failed at %u\n", k); abort(); }
            g = gp;
        }
        cur = next;
    }
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
/* leaves idx[0..k) of the n = 2^log_n point set of build_fftree above (same constants, src/lib.rs:45-78); 0 on success */
int ORA(leaves_at)(unsigned log_n, const uint64_t* idx, size_t k, void* out) {
    goodcurve curve;
    if (!goodcurve_new_odd(
            fe_from_dec("31172306031375832341232376275243462303334845584808513005362718476441963632613"),
            fe_from_dec("45508371059383884471556188660911097844526467659576498497548207627741160623272"), &curve))
        return -1;
    ecpoint offset, gen; offset.inf = gen.inf = 0;
    offset.x = fe_from_dec("105623886150579165427389078198493427091405550492761682382732004625374789850161");
    offset.y = fe_from_dec("7709812624542158994629670452026922591039826164720902911013234773380889499231");
    gen.x = fe_from_dec("41293412487153066667050767300223451435019201659857889215769525847559135483332");
    gen.y = fe_from_dec("73754924733368840065089190002333366411120578552679996887076912271884749237510");
    const unsigned two_adicity = 36;
    if (log_n >= two_adicity) return -1;
    wcurve w = goodcurve_w(&curve);
    for (unsigned i = 0; i < two_adicity - log_n; ++i) gen = ec_add(&w, gen, gen);
    ec_leaves_at(&w, offset, gen, idx, k, (fe*)out);
    return 0;
}
/* Montgomery <-> standard form (little-endian 32 bytes per element) */
void ORA(from_std)(const void* in, void* out, size_t n) {
    for (size_t i = 0; i < n; ++i) ((fe*)out)[i] = fe_from_std(((const uint64_t*)in) + 4 * i);
}
void ORA(to_std)(const void* in, void* out, size_t n) {
    for (size_t i = 0; i < n; ++i) fe_to_std(((const fe*)in)[i], ((uint64_t*)out) + 4 * i);
}
