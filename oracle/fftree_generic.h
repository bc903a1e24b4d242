/* TEST INFRASTRUCTURE — CPU oracle, not product code.
 *
 * Field-generic CPU restatement of the reference's FFTree algorithms.  Include AFTER one of
 * field_secp256k1.h / field_m31.h (which define `fe` and the fe_* operations) and after defining
 * ORA(name) to the exported symbol prefix.  Every function cites the reference lines it follows
 * (paths relative to /root/reference).  The recursion, the per-call allocation pattern and the
 * per-REDC batch inversion of the reference are kept on purpose: this file is also the CPU
 * baseline that bench.py times ("cpu_baseline", kind "port").
 *
 * Parity status: the reference (Rust, un-vendored ark-* crates) cannot be built in this image, so
 * this oracle is pinned against (1) the identities the reference's own tests assert — ENTER ==
 * naive evaluation on the leaves (src/lib.rs:108-120, 239-251), EXTEND == evaluation on the other
 * moiety (src/lib.rs:122-152), EXIT(ENTER(c)) == c (src/lib.rs:253-264, examples/interp_eval.rs:33),
 * DEGREE (src/lib.rs:266-278), the cubic-roots known answer (src/utils.rs:401-413) — and (2) golden
 * vectors from an independent Python big-integer implementation (tests/golden/gen_golden.py) that
 * derives the leaves from the curve constants of src/lib.rs:45-59 / 201-206.  The ark-ff internal
 * limb encoding (x*2^256 mod p, little-endian u64) is restated from the crate's documentation and is
 * not confirmed against a Rust-produced buffer ("encoding unpinned").
 */
#ifndef ORA
#error "define ORA(name) before including fftree_generic.h"
#endif
#include <stdio.h>
#include <assert.h>

/* ------------------------------------------------------------------------------------------
 * containers (src/utils.rs:228-347, 367-390)
 * ---------------------------------------------------------------------------------------- */
typedef struct { fe m[2][2]; } mat2; /* Mat2x2: row-major [[F;2];2], src/utils.rs:317-318 */

typedef struct { /* RationalMap, src/utils.rs:367-371; coefficients low -> high */
    fe num[3]; int nnum;
    fe den[3]; int nden;
} ratmap;

typedef struct fftree {
    size_t n;            /* number of leaves                                           */
    fe* f;               /* BinaryTree<F>, 2n entries in heap order (src/fftree.rs:25) */
    mat2* recombine;     /* BinaryTree<Mat2x2>, n entries (src/fftree.rs:26)           */
    mat2* decompose;     /* (src/fftree.rs:27)                                         */
    ratmap* maps; int nmaps;
    struct fftree* subtree;
    fe *xnn_s, *xnn_s_inv;             /* n   (src/fftree.rs:30-31) */
    fe *z0_s1, *z1_s0;                 /* n/2 (src/fftree.rs:32-33) */
    fe *z0_inv_s1, *z1_inv_s0;         /* n/2 (src/fftree.rs:34-35) */
    fe *z0z0_rem_xnn_s, *z1z1_rem_xnn_s; /* n (src/fftree.rs:36-37) */
} fftree;

enum { MOIETY_S0 = 0, MOIETY_S1 = 1 }; /* src/fftree.rs:17-21 */

/* Every vector of the recursion is a fresh zero-initialised allocation that is released when its function returns, like
 * the reference's Vec<F>s.  With glibc that is an mmap / page-fault / munmap cycle per large vector, which serialises on the
 * kernel's address-space lock when many threads run transforms at once (the 64-thread socket baseline of bench.py scaled
 * 13x).  A per-thread size-class cache keeps the allocation PATTERN (same calls, same zeroing) but takes the system calls
 * out of the timed path, so the socket figure is bound by the field arithmetic.  ECFFT_ORACLE_POOL=0 restores plain
 * calloc / free. */
typedef struct ora_blk { struct ora_blk* next; size_t bin; } ora_blk;          /* 16-byte header in front of the payload */
#define ORA_NBINS 48
static __thread ora_blk* ora_bins[ORA_NBINS];
static int ora_pool_enabled(void) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("ECFFT_ORACLE_POOL"); on = !(e && e[0] == '0'); }
    return on;
}
static fe* fe_alloc(size_t n) {
    size_t bytes = (n ? n : 1) * sizeof(fe);
    unsigned bin = 0; while (((size_t)1 << bin) < bytes) ++bin;
    ora_blk* b = NULL;
    if (ora_pool_enabled() && bin < ORA_NBINS && ora_bins[bin]) { b = ora_bins[bin]; ora_bins[bin] = b->next; }
    else { b = (ora_blk*)malloc(sizeof(ora_blk) + ((size_t)1 << bin)); if (!b) abort(); }
    b->bin = bin; b->next = NULL;
    memset(b + 1, 0, bytes);
    return (fe*)(b + 1);
}
static void fe_free(fe* p) {
    if (!p) return;
    ora_blk* b = (ora_blk*)p - 1;
    if (ora_pool_enabled() && b->bin < ORA_NBINS) { b->next = ora_bins[b->bin]; ora_bins[b->bin] = b; }
    else free(b);
}

static unsigned ilog2_sz(size_t n) { unsigned l = 0; while (n > 1) { n >>= 1; ++l; } return l; }

/* BinaryTree::get_layer for a heap-ordered array with `len` entries (src/utils.rs:248-252) */
static size_t layer_off(size_t len, unsigned i) { return (len / 2) >> i; }

/* DensePolynomial::evaluate (Horner), used at src/utils.rs:384 and src/fftree.rs:358 */
static fe poly_eval(const fe* c, int n, fe x) {
    fe r = fe_zero();
    for (int i = n - 1; i >= 0; --i) r = fe_add(fe_mul(r, x), c[i]);
    return r;
}
/* RationalMap::map (src/utils.rs:383-385); returns 0 if the denominator vanishes */
static int ratmap_map(const ratmap* m, fe x, fe* out) {
    fe d = poly_eval(m->den, m->nden, x);
    if (fe_is_zero(d)) return 0;
    *out = fe_mul(poly_eval(m->num, m->nnum, x), fe_inv(d));
    return 1;
}
/* &Mat2x2 * &[F;2]  (src/utils.rs:338-347) */
static void mat2_apply(const mat2* m, fe a, fe b, fe* o0, fe* o1) {
    *o0 = fe_add(fe_mul(m->m[0][0], a), fe_mul(m->m[0][1], b));
    *o1 = fe_add(fe_mul(m->m[1][0], a), fe_mul(m->m[1][1], b));
}
/* Mat2x2::inverse (src/utils.rs:325-335) */
static mat2 mat2_inverse(const mat2* m) {
    fe det = fe_sub(fe_mul(m->m[0][0], m->m[1][1]), fe_mul(m->m[0][1], m->m[1][0]));
    fe di = fe_inv(det);
    mat2 r;
    r.m[0][0] = fe_mul(m->m[1][1], di);
    r.m[0][1] = fe_mul(fe_neg(m->m[0][1]), di);
    r.m[1][0] = fe_mul(fe_neg(m->m[1][0]), di);
    r.m[1][1] = fe_mul(m->m[0][0], di);
    return r;
}
/* ark_ff::batch_inversion (Montgomery's trick; zero entries are left untouched) */
static void batch_inversion(fe* v, size_t n) {
    fe* prod = fe_alloc(n);
    size_t np = 0;
    fe acc = fe_one();
    for (size_t i = 0; i < n; ++i) { if (fe_is_zero(v[i])) continue; acc = fe_mul(acc, v[i]); prod[np++] = acc; }
    if (np == 0) { fe_free(prod); return; }
    acc = fe_inv(acc);
    for (size_t i = n; i-- > 0;) {
        if (fe_is_zero(v[i])) continue;
        --np;                                   /* prod[np] = product up to and including v[i] */
        fe below = np ? prod[np - 1] : fe_one();
        fe newacc = fe_mul(acc, v[i]);
        v[i] = fe_mul(acc, below);
        acc = newacc;
    }
    fe_free(prod);
}

/* ------------------------------------------------------------------------------------------
 * subtree selection (src/fftree.rs:484-496)
 * ---------------------------------------------------------------------------------------- */
static const fftree* subtree_with_size(const fftree* t, size_t n) {
    assert(n && (n & (n - 1)) == 0);
    while (t && n < t->n) t = t->subtree;
    if (!t || n > t->n) { fprintf(stderr, "FFTree is too small\n"); return NULL; }
    return t;
}

/* ------------------------------------------------------------------------------------------
 * EXTEND (src/fftree.rs:72-126)
 * ---------------------------------------------------------------------------------------- */
static fe* extend_impl(const fftree* t, const fe* evals, size_t n, int moiety) {
    fe* res = fe_alloc(n);
    if (n == 1) { res[0] = evals[0]; return res; }            /* :74-76 */
    unsigned layer = ilog2_sz(2 * t->n) - 2 - ilog2_sz(n);     /* :78  (f.num_layers() = log2(2n)) */
    size_t h = n / 2;
    fe* e0 = fe_alloc(h); fe* e1 = fe_alloc(h);                /* :81-82 */
    const mat2* D = t->decompose + layer_off(t->n, layer);     /* get_layer(layer) */
    size_t skip = (moiety == MOIETY_S0) ? 1 : 0;               /* :87-90 */
    for (size_t i = 0; i < h; ++i)                             /* :83-97 */
        mat2_apply(&D[2 * i + skip], evals[i], evals[i + h], &e0[i], &e1[i]);
    fe* e0p = extend_impl(t, e0, h, moiety);                   /* :100 */
    fe* e1p = extend_impl(t, e1, h, moiety);                   /* :101 */
    const mat2* R = t->recombine + layer_off(t->n, layer);
    skip = (moiety == MOIETY_S0) ? 0 : 1;                      /* :108-111 */
    for (size_t i = 0; i < h; ++i)                             /* :104-118 */
        mat2_apply(&R[2 * i + skip], e0p[i], e1p[i], &res[i], &res[i + h]);
    fe_free(e0); fe_free(e1); fe_free(e0p); fe_free(e1p);
    return res;
}
static fe* tree_extend(const fftree* t, const fe* evals, size_t n, int moiety) { /* :123-126 */
    const fftree* s = subtree_with_size(t, n * 2);
    return s ? extend_impl(s, evals, n, moiety) : NULL;
}
/* MEXTEND (src/fftree.rs:128-141) */
static fe* mextend_impl(const fftree* t, const fe* evals, size_t n, int moiety) {
    fe* e = extend_impl(t, evals, n, moiety);
    const fe* z = (moiety == MOIETY_S1) ? t->z0_s1 : t->z1_s0;
    for (size_t i = 0; i < n; ++i) e[i] = fe_add(e[i], z[i]);
    return e;
}
static fe* tree_mextend(const fftree* t, const fe* evals, size_t n, int moiety) {
    const fftree* s = subtree_with_size(t, n * 2);
    return s ? mextend_impl(s, evals, n, moiety) : NULL;
}

/* ------------------------------------------------------------------------------------------
 * ENTER (src/fftree.rs:143-167)
 * ---------------------------------------------------------------------------------------- */
static fe* tree_enter(const fftree* t, const fe* coeffs, size_t n);
static fe* enter_impl(const fftree* t, const fe* coeffs, size_t n) {
    fe* res = fe_alloc(n);
    if (n == 1) { res[0] = coeffs[0]; return res; }            /* :145-147 */
    const fftree* st = t->subtree;
    size_t h = n / 2;
    fe* u0 = tree_enter(st, coeffs, h);                        /* :150 */
    fe* v0 = tree_enter(st, coeffs + h, h);                    /* :151 */
    fe* u1 = tree_extend(t, u0, h, MOIETY_S1);                 /* :152 */
    fe* v1 = tree_extend(t, v0, h, MOIETY_S1);                 /* :153 */
    for (size_t i = 0; i < h; ++i) {                           /* :155-159 */
        res[2 * i] = fe_add(u0[i], fe_mul(v0[i], t->xnn_s[2 * i]));
        res[2 * i + 1] = fe_add(u1[i], fe_mul(v1[i], t->xnn_s[2 * i + 1]));
    }
    fe_free(u0); fe_free(v0); fe_free(u1); fe_free(v1);
    return res;
}
static fe* tree_enter(const fftree* t, const fe* coeffs, size_t n) { /* :164-167 */
    const fftree* s = subtree_with_size(t, n);
    return s ? enter_impl(s, coeffs, n) : NULL;
}

/* ------------------------------------------------------------------------------------------
 * REDC / MOD (src/fftree.rs:232-289)
 * ---------------------------------------------------------------------------------------- */
static fe* redc_impl(const fftree* t, const fe* evals, const fe* a, size_t n, int moiety) {
    size_t h = n / 2;
    fe* e0 = fe_alloc(h); fe* e1 = fe_alloc(h); fe* a0_inv = fe_alloc(h); fe* a1 = fe_alloc(h);
    for (size_t i = 0; i < h; ++i) {                           /* :233-234 */
        e0[i] = evals[2 * i]; e1[i] = evals[2 * i + 1];
        a0_inv[i] = a[2 * i]; a1[i] = a[2 * i + 1];
    }
    batch_inversion(a0_inv, h);                                /* :235 */
    fe* t0 = fe_alloc(h);
    for (size_t i = 0; i < h; ++i) t0[i] = fe_mul(e0[i], a0_inv[i]); /* :238 */
    fe* g1 = extend_impl(t, t0, h, moiety == MOIETY_S1 ? MOIETY_S0 : MOIETY_S1); /* :239-245 */
    const fe* z_inv = (moiety == MOIETY_S0) ? t->z0_inv_s1 : t->z1_inv_s0;       /* :247-250 */
    fe* h1 = fe_alloc(h);
    for (size_t i = 0; i < h; ++i)                             /* :253-255 */
        h1[i] = fe_mul(fe_sub(e1[i], fe_mul(g1[i], a1[i])), z_inv[i]);
    fe* h0 = extend_impl(t, h1, h, moiety);                    /* :256 */
    fe* res = fe_alloc(n);
    for (size_t i = 0; i < h; ++i) { res[2 * i] = h0[i]; res[2 * i + 1] = h1[i]; } /* :258 */
    fe_free(e0); fe_free(e1); fe_free(a0_inv); fe_free(a1); fe_free(t0); fe_free(g1); fe_free(h1); fe_free(h0);
    return res;
}
static fe* modular_reduce_impl(const fftree* t, const fe* evals, const fe* a, const fe* c, size_t n) {
    fe* h = redc_impl(t, evals, a, n, MOIETY_S0);              /* :278 */
    for (size_t i = 0; i < n; ++i) h[i] = fe_mul(h[i], c[i]);  /* :279 */
    fe* r = redc_impl(t, h, a, n, MOIETY_S0);                  /* :280 */
    fe_free(h);
    return r;
}
static fe* tree_modular_reduce(const fftree* t, const fe* evals, const fe* a, const fe* c, size_t n) {
    const fftree* s = subtree_with_size(t, n);                 /* :286-289 */
    return s ? modular_reduce_impl(s, evals, a, c, n) : NULL;
}

/* ------------------------------------------------------------------------------------------
 * EXIT (src/fftree.rs:200-230)
 * ---------------------------------------------------------------------------------------- */
static fe* exit_impl(const fftree* t, const fe* evals, size_t n) {
    fe* res = fe_alloc(n);
    if (n == 1) { res[0] = evals[0]; return res; }             /* :202-204 */
    size_t h = n / 2;
    fe* mr = modular_reduce_impl(t, evals, t->xnn_s, t->z0z0_rem_xnn_s, n); /* :206-207 */
    fe* u0 = fe_alloc(h);
    for (size_t i = 0; i < h; ++i) u0[i] = mr[2 * i];          /* :208-210 */
    fe_free(mr);
    const fftree* st = t->subtree;
    fe* a = exit_impl(st, u0, h);                              /* :213 */
    fe* v0 = fe_alloc(h);
    for (size_t i = 0; i < h; ++i)                             /* :215-219 */
        v0[i] = fe_mul(fe_sub(evals[2 * i], u0[i]), t->xnn_s_inv[2 * i]);
    fe* b = exit_impl(st, v0, h);                              /* :220 */
    memcpy(res, a, h * sizeof(fe)); memcpy(res + h, b, h * sizeof(fe)); /* :222 */
    fe_free(u0); fe_free(a); fe_free(v0); fe_free(b);
    return res;
}
static fe* tree_exit(const fftree* t, const fe* evals, size_t n) { /* :227-230 */
    const fftree* s = subtree_with_size(t, n);
    return s ? exit_impl(s, evals, n) : NULL;
}

/* ------------------------------------------------------------------------------------------
 * DEGREE (src/fftree.rs:169-198)
 * ---------------------------------------------------------------------------------------- */
static size_t degree_impl(const fftree* t, const fe* evals, size_t n) {
    if (n == 1) return 0;
    size_t h = n / 2;
    fe* e0 = fe_alloc(h); fe* e1 = fe_alloc(h);
    for (size_t i = 0; i < h; ++i) { e0[i] = evals[2 * i]; e1[i] = evals[2 * i + 1]; }
    fe* g1 = extend_impl(t, e0, h, MOIETY_S1);
    size_t r;
    if (memcmp(g1, e1, h * sizeof(fe)) == 0) {
        r = degree_impl(t->subtree, e0, h);
    } else {
        fe* t1 = fe_alloc(h);
        for (size_t i = 0; i < h; ++i) t1[i] = fe_mul(fe_sub(e1[i], g1[i]), t->z0_inv_s1[i]);
        fe* t0 = extend_impl(t, t1, h, MOIETY_S0);
        r = h + degree_impl(t->subtree, t0, h);
        fe_free(t1); fe_free(t0);
    }
    fe_free(e0); fe_free(e1); fe_free(g1);
    return r;
}

/* ------------------------------------------------------------------------------------------
 * VANISH (src/fftree.rs:291-316)
 * ---------------------------------------------------------------------------------------- */
static fe* vanish_impl(const fftree* t, const fe* dom, size_t n) {
    fe* res = fe_alloc(2 * n);
    if (n == 1) {                                              /* :293-298 */
        const fe* l = t->f + t->n; assert(t->n == 2);
        res[0] = fe_sub(dom[0], l[0]); res[1] = fe_sub(dom[0], l[1]);
        return res;
    }
    const fftree* st = t->subtree;
    size_t h = n / 2;
    fe* qp = vanish_impl(st, dom, h);                          /* :301 */
    fe* qpp = vanish_impl(st, dom + h, h);                     /* :302 */
    fe* q_s0 = fe_alloc(n);
    for (size_t i = 0; i < n; ++i) q_s0[i] = fe_mul(qp[i], qpp[i]); /* :303 */
    fe* q_s1 = tree_mextend(t, q_s0, n, MOIETY_S1);            /* :304 */
    for (size_t i = 0; i < n; ++i) { res[2 * i] = q_s0[i]; res[2 * i + 1] = q_s1[i]; } /* :305-307 */
    fe_free(qp); fe_free(qpp); fe_free(q_s0); fe_free(q_s1);
    return res;
}
static fe* tree_vanish(const fftree* t, const fe* dom, size_t n) { /* :313-316 */
    const fftree* s = subtree_with_size(t, n * 2);
    return s ? vanish_impl(s, dom, n) : NULL;
}

/* ------------------------------------------------------------------------------------------
 * construction (src/fftree.rs:42-70, 318-482)
 * ---------------------------------------------------------------------------------------- */
static void tree_free(fftree* t) {
    if (!t) return;
    tree_free(t->subtree);
    fe_free(t->f); free(t->recombine); free(t->decompose); free(t->maps);
    fe_free(t->xnn_s); fe_free(t->xnn_s_inv); fe_free(t->z0_s1); fe_free(t->z1_s0);
    fe_free(t->z0_inv_s1); fe_free(t->z1_inv_s0); fe_free(t->z0z0_rem_xnn_s); fe_free(t->z1z1_rem_xnn_s);
    free(t);
}

static fftree* from_tree(fe* f, size_t n, const ratmap* maps, int nmaps);

/* derive_subtree (src/fftree.rs:465-482) */
static fftree* derive_subtree(const fe* f, size_t leaves, const ratmap* maps, int nmaps) {
    size_t n = leaves / 2;
    if (n == 0) return NULL;
    fe* fp = fe_alloc(2 * n);
    /* every layer of f', from the leaves up, takes every second element of the parent's layer */
    unsigned layers_p = ilog2_sz(2 * n); /* depth of f' */
    for (unsigned i = 0; i < layers_p; ++i) {
        size_t off_p = layer_off(2 * n, i), sz_p = off_p;       /* f' layer i: [n>>i, 2(n>>i)) */
        size_t off = layer_off(2 * leaves, i);
        for (size_t j = 0; j < sz_p; ++j) fp[off_p + j] = f[off + 2 * j];
    }
    int nm = nmaps > 0 ? nmaps - 1 : 0;                         /* split_last (:480) */
    return from_tree(fp, n, maps, nm);
}

/* from_tree (src/fftree.rs:318-463); takes ownership of f */
/* TEST INFRASTRUCTURE switch (per thread): build ONLY what extend_impl (src/fftree.rs:72-120) reads of ONE tree — the f layers and the
 * recombine / decompose matrices (:341-363) — and none of the subtree chain or the xnn / z tables.  An EXTEND of e evaluations runs on
 * T_2e alone (:123-126), so BASELINE configs[3] (e = 2^22 on T_2^23) can be checked element for element without the quarter of an hour
 * the whole 2^23 chain would take.  The matrix code below is the same code either way. */
static __thread int ora_extend_only = 0;
static fftree* from_tree(fe* f, size_t n, const ratmap* maps, int nmaps) {
    fftree* t = (fftree*)calloc(1, sizeof(fftree));
    t->n = n; t->f = f;
    t->nmaps = nmaps;
    t->maps = (ratmap*)calloc(nmaps ? nmaps : 1, sizeof(ratmap));
    if (nmaps) memcpy(t->maps, maps, nmaps * sizeof(ratmap));
    if (!ora_extend_only) t->subtree = derive_subtree(f, n, maps, nmaps);            /* :319 */
    const fe* s = f + n;                                       /* f_layers[0] */
    uint64_t nn = n / 2, nnnn = n / 4;                         /* :322-323 */

    fe* xnnnn_s = fe_alloc(n); fe* xnnnn_s_inv = fe_alloc(n);  /* :328-330 */
    t->xnn_s = fe_alloc(n); t->xnn_s_inv = fe_alloc(n);        /* :331-333 */
    for (size_t i = 0; i < (ora_extend_only ? 0 : n); ++i) {
        xnnnn_s[i] = fe_pow_u64(s[i], nnnn); xnnnn_s_inv[i] = xnnnn_s[i];
        t->xnn_s[i] = fe_pow_u64(s[i], nn); t->xnn_s_inv[i] = t->xnn_s[i];
    }
    if (!ora_extend_only) { batch_inversion(xnnnn_s_inv, n); batch_inversion(t->xnn_s_inv, n); }

    size_t hn = n / 2;
    fe* s0 = fe_alloc(hn); fe* s1 = fe_alloc(hn);              /* :336 */
    for (size_t i = 0; i < hn; ++i) { s0[i] = s[2 * i]; s1[i] = s[2 * i + 1]; }

    /* matrices (:341-363): identity everywhere, then per layer k with d = |L_k|/2 > 1 */
    t->recombine = (mat2*)calloc(n ? n : 1, sizeof(mat2));
    t->decompose = (mat2*)calloc(n ? n : 1, sizeof(mat2));
    for (size_t i = 0; i < n; ++i) {
        t->recombine[i].m[0][0] = t->recombine[i].m[1][1] = fe_one();
        t->decompose[i].m[0][0] = t->decompose[i].m[1][1] = fe_one();
    }
    unsigned mat_layers = n ? ilog2_sz(n) : 0;                 /* depth of a len-n BinaryTree */
    for (unsigned k = 0; k < mat_layers && (int)k < nmaps; ++k) {
        const fe* l = f + layer_off(2 * n, k); size_t lsz = layer_off(2 * n, k);
        size_t d = lsz / 2;
        if (d == 1) continue;                                  /* :350-352 */
        const ratmap* map = &maps[k];
        mat2* R = t->recombine + layer_off(n, k);
        mat2* D = t->decompose + layer_off(n, k);
        for (size_t i = 0; i < d; ++i) {                       /* :355-362 */
            fe p0 = l[i], p1 = l[i + d];
            fe v0 = fe_pow_u64(poly_eval(map->den, map->nden, p0), (uint64_t)(d / 2 - 1));
            fe v1 = fe_pow_u64(poly_eval(map->den, map->nden, p1), (uint64_t)(d / 2 - 1));
            R[i].m[0][0] = v0; R[i].m[0][1] = fe_mul(p0, v0);
            R[i].m[1][0] = v1; R[i].m[1][1] = fe_mul(p1, v1);
            D[i] = mat2_inverse(&R[i]);
        }
    }

    if (ora_extend_only) { fe_free(xnnnn_s); fe_free(xnnnn_s_inv); fe_free(s0); fe_free(s1); return t; }
    /* z0_s1, z1_s0 (:384-405) */
    t->z0_s1 = fe_alloc(hn); t->z1_s0 = fe_alloc(hn);
    if (n > 2) {
        const fftree* st = t->subtree;
        size_t q = hn / 2;
        fe* st_z0_s0 = fe_alloc(hn); fe* st_z1_s0 = fe_alloc(hn); /* :389-390 */
        for (size_t i = 0; i < q; ++i) {
            st_z0_s0[2 * i] = fe_zero(); st_z0_s0[2 * i + 1] = st->z0_s1[i];
            st_z1_s0[2 * i] = st->z1_s0[i]; st_z1_s0[2 * i + 1] = fe_zero();
        }
        fe* st_z0_s1 = tree_extend(t, st_z0_s0, hn, MOIETY_S1); /* :391 */
        fe* st_z1_s1 = tree_extend(t, st_z1_s0, hn, MOIETY_S1); /* :392 */
        for (size_t i = 0; i < hn; ++i) t->z0_s1[i] = fe_mul(st_z0_s1[i], st_z1_s1[i]); /* :393 */
        fe_free(st_z0_s0); fe_free(st_z1_s0); fe_free(st_z0_s1); fe_free(st_z1_s1);
        fe* z1_s = tree_vanish(t, s1, hn);                     /* :396 */
        for (size_t i = 0; i < hn; ++i) t->z1_s0[i] = z1_s[2 * i]; /* :397 */
        fe_free(z1_s);
    } else if (n == 2) {
        t->z0_s1[0] = fe_sub(s1[0], s0[0]);                    /* :401 */
        t->z1_s0[0] = fe_sub(s0[0], s1[0]);                    /* :402 */
    }
    t->z0_inv_s1 = fe_alloc(hn); t->z1_inv_s0 = fe_alloc(hn);  /* :407-410 */
    memcpy(t->z0_inv_s1, t->z0_s1, hn * sizeof(fe)); memcpy(t->z1_inv_s0, t->z1_s0, hn * sizeof(fe));
    batch_inversion(t->z0_inv_s1, hn); batch_inversion(t->z1_inv_s0, hn);

    /* z0z0_rem_xnn_s, z1z1_rem_xnn_s (:417-460) */
    t->z0z0_rem_xnn_s = fe_alloc(n); t->z1z1_rem_xnn_s = fe_alloc(n);
    if (n > 2) {
        const fftree* st = t->subtree;
        fe* sq_s0 = fe_alloc(hn);                              /* :421-423 */
        for (size_t i = 0; i < hn; ++i) sq_s0[i] = fe_mul(st->z0z0_rem_xnn_s[i], st->z1z1_rem_xnn_s[i]);
        fe* zz_nnnn_s0 = tree_modular_reduce(st, sq_s0, st->xnn_s, st->z0z0_rem_xnn_s, hn); /* :424-425 */
        fe* zz_nnnn_s1 = tree_extend(t, zz_nnnn_s0, hn, MOIETY_S1); /* :426 */
        fe* zz_nnnn_s = fe_alloc(n);                           /* :427-429 */
        for (size_t i = 0; i < hn; ++i) { zz_nnnn_s[2 * i] = zz_nnnn_s0[i]; zz_nnnn_s[2 * i + 1] = zz_nnnn_s1[i]; }
        fe* tmp = fe_alloc(n);                                 /* :430-438 */
        for (size_t i = 0; i < n; ++i) {
            fe z0 = (i & 1) ? t->z0_s1[i / 2] : fe_zero();
            fe y = fe_sub(z0, t->xnn_s[i]);
            fe ysq = fe_sqr(y);
            tmp[i] = fe_mul(fe_sub(ysq, zz_nnnn_s[i]), xnnnn_s_inv[i]);
        }
        fe* div_rem = tree_modular_reduce(t, tmp, xnnnn_s, zz_nnnn_s, n); /* :439-440 */
        for (size_t i = 0; i < n; ++i)                         /* :441-446 */
            t->z0z0_rem_xnn_s[i] = fe_add(zz_nnnn_s[i], fe_mul(xnnnn_s[i], div_rem[i]));
        fe* z1z1 = fe_alloc(n);                                /* :449-451 */
        for (size_t i = 0; i < n; ++i) {
            fe z1 = (i & 1) ? fe_zero() : t->z1_s0[i / 2];
            z1z1[i] = fe_sqr(fe_sub(z1, t->xnn_s[i]));
        }
        fe* r = tree_modular_reduce(t, z1z1, t->xnn_s, t->z0z0_rem_xnn_s, n); /* :452 */
        memcpy(t->z1z1_rem_xnn_s, r, n * sizeof(fe));
        fe_free(sq_s0); fe_free(zz_nnnn_s0); fe_free(zz_nnnn_s1); fe_free(zz_nnnn_s); fe_free(tmp); fe_free(div_rem);
        fe_free(z1z1); fe_free(r);
    } else if (n == 2) {
        t->z0z0_rem_xnn_s[0] = t->z0z0_rem_xnn_s[1] = fe_sqr(s0[0]); /* :456 */
        t->z1z1_rem_xnn_s[0] = t->z1z1_rem_xnn_s[1] = fe_sqr(s1[0]); /* :457 */
    }
    fe_free(xnnnn_s); fe_free(xnnnn_s_inv); fe_free(s0); fe_free(s1);
    return t;
}

/* FFTree::new (src/fftree.rs:42-70) */
static fftree* tree_new(const fe* leaves, size_t n, const ratmap* maps, int nmaps) {
    assert(n && (n & (n - 1)) == 0);
    assert((int)ilog2_sz(n) == nmaps);
    fe* f = fe_alloc(2 * n);
    memcpy(f + n, leaves, n * sizeof(fe));
    for (int k = 0; k < nmaps; ++k) {                          /* :56-67 */
        const fe* prev = f + layer_off(2 * n, k);
        size_t lsz = layer_off(2 * n, k + 1);
        fe* layer = f + lsz;
        for (size_t i = 0; i < lsz; ++i) {
            fe v; int ok = ratmap_map(&maps[k], prev[i], &v); assert(ok); (void)ok;
            layer[i] = v;
#ifdef ORACLE_DEBUG_ASSERT
            fe w; ratmap_map(&maps[k], prev[i + lsz], &w); assert(fe_eq(v, w)); /* :65 */
#endif
        }
    }
    return from_tree(f, n, maps, nmaps);
}

/* ------------------------------------------------------------------------------------------
 * elliptic-curve layer, construction only (src/ec.rs)
 * general Weierstrass y^2 + a1 x y + a3 y = x^3 + a2 x^2 + a4 x + a6 (src/ec.rs:291-312)
 * ---------------------------------------------------------------------------------------- */
typedef struct { fe a1, a2, a3, a4, a6; } wcurve;
typedef struct { fe x, y; int inf; } ecpoint;

/* Point::add (src/ec.rs:376-424) */
static ecpoint ec_add(const wcurve* c, ecpoint p, ecpoint q) {
    if (p.inf) return q;
    if (q.inf) return p;
    ecpoint r; r.inf = 0;
    fe x1 = p.x, y1 = p.y, x2 = q.x, y2 = q.y;
    if (fe_eq(x1, x2) && fe_is_zero(fe_add(fe_add(fe_add(y1, y2), fe_mul(c->a1, x2)), c->a3))) {
        r.inf = 1; r.x = r.y = fe_zero(); return r;            /* :400-401 */
    }
    fe lambda, nu;
    if (fe_eq(x1, x2)) {                                       /* :405-412 tangent */
        fe x1x1 = fe_sqr(x1), a2x1 = fe_mul(c->a2, x1), a1x1 = fe_mul(c->a1, x1);
        fe den = fe_inv(fe_add(fe_add(fe_add(y1, y1), a1x1), c->a3));
        fe num = fe_sub(fe_add(fe_add(fe_add(fe_add(fe_add(x1x1, x1x1), x1x1), a2x1), a2x1), c->a4),
                        fe_mul(c->a1, y1));
        lambda = fe_mul(num, den);
        fe nnu = fe_sub(fe_add(fe_add(fe_add(fe_neg(fe_mul(x1x1, x1)), fe_mul(c->a4, x1)), c->a6), c->a6),
                        fe_mul(c->a3, y1));
        nu = fe_mul(nnu, den);
    } else {                                                   /* :413-417 chord */
        fe den = fe_inv(fe_sub(x2, x1));
        lambda = fe_mul(fe_sub(y2, y1), den);
        nu = fe_mul(fe_sub(fe_mul(y1, x2), fe_mul(y2, x1)), den);
    }
    r.x = fe_sub(fe_sub(fe_sub(fe_add(fe_sqr(lambda), fe_mul(c->a1, lambda)), c->a2), x1), x2); /* :418 */
    r.y = fe_sub(fe_sub(fe_mul(fe_neg(fe_add(lambda, c->a1)), r.x), nu), c->a3);                  /* :419 */
    return r;
}
/* two_adicity (src/utils.rs:356-365) */
static int ec_two_adicity(const wcurve* c, ecpoint p) {
    for (int i = 0; i < 2048; ++i) { if (p.inf) return i; p = ec_add(c, p, p); }
    return -1;
}
/* leaves: x(coset_offset + i*G), src/lib.rs:72-78 and src/ec.rs:545-551 */
static void ec_leaves(const wcurve* c, ecpoint offset, ecpoint gen, fe* leaves, size_t n) {
    ecpoint acc; acc.inf = 1; acc.x = acc.y = fe_zero();
    for (size_t i = 0; i < n; ++i) { leaves[i] = ec_add(c, offset, acc).x; acc = ec_add(c, acc, gen); }
}

/* individual leaves x(coset_offset + idx*G) of the same point set (src/lib.rs:72-78, src/ec.rs:545-551) without walking the whole
 * coset: idx*G by double-and-add on the reference's affine group law.  Test infrastructure: gives the spot checks at sizes whose
 * oracle tree would take minutes (n >= 2^22) leaves that do not come from the library under test. */
static void ec_leaves_at(const wcurve* c, ecpoint offset, ecpoint gen, const uint64_t* idx, size_t k, fe* out) {
    for (size_t j = 0; j < k; ++j) {
        ecpoint acc; acc.inf = 1; acc.x = acc.y = fe_zero();
        ecpoint dbl = gen;
        for (uint64_t e = idx[j]; e; e >>= 1) { if (e & 1) acc = ec_add(c, acc, dbl); dbl = ec_add(c, dbl, dbl); }
        out[j] = ec_add(c, offset, acc).x;
    }
}

/* ------------------------------------------------------------------------------------------
 * exported C interface (loaded by tests/ and bench.py through ctypes)
 * ---------------------------------------------------------------------------------------- */
enum {
    ORA_T_F = 0, ORA_T_RECOMBINE, ORA_T_DECOMPOSE, ORA_T_XNN_S, ORA_T_XNN_S_INV, ORA_T_Z0_S1, ORA_T_Z1_S0,
    ORA_T_Z0_INV_S1, ORA_T_Z1_INV_S0, ORA_T_Z0Z0, ORA_T_Z1Z1
};

size_t ORA(elem_size)(void) { return sizeof(fe); }
void ORA(free_tree)(void* t) { tree_free((fftree*)t); }
size_t ORA(tree_size)(const void* t) { return ((const fftree*)t)->n; }

/* pointer to a table of the subtree with m leaves; *count receives the number of `fe` in it */
const void* ORA(table)(const void* tv, size_t m, int which, size_t* count) {
    const fftree* t = subtree_with_size((const fftree*)tv, m);
    if (!t) return NULL;
    size_t n = t->n; const void* p = NULL; size_t c = 0;
    switch (which) {
        case ORA_T_F: p = t->f; c = 2 * n; break;
        case ORA_T_RECOMBINE: p = t->recombine; c = 4 * n; break;
        case ORA_T_DECOMPOSE: p = t->decompose; c = 4 * n; break;
        case ORA_T_XNN_S: p = t->xnn_s; c = n; break;
        case ORA_T_XNN_S_INV: p = t->xnn_s_inv; c = n; break;
        case ORA_T_Z0_S1: p = t->z0_s1; c = n / 2; break;
        case ORA_T_Z1_S0: p = t->z1_s0; c = n / 2; break;
        case ORA_T_Z0_INV_S1: p = t->z0_inv_s1; c = n / 2; break;
        case ORA_T_Z1_INV_S0: p = t->z1_inv_s0; c = n / 2; break;
        case ORA_T_Z0Z0: p = t->z0z0_rem_xnn_s; c = n < 2 ? 0 : n; break;   /* Ordering::Less => left empty (src/fftree.rs:459) */
        case ORA_T_Z1Z1: p = t->z1z1_rem_xnn_s; c = n < 2 ? 0 : n; break;
        default: return NULL;
    }
    if (count) *count = c;
    return p;
}
/* rational map k of the full tree: 3 numerator + 3 denominator coefficients (zero padded) */
int ORA(ratmap)(const void* tv, int k, void* num3, void* den3) {
    const fftree* t = (const fftree*)tv;
    if (k < 0 || k >= t->nmaps) return -1;
    fe z[3] = {fe_zero(), fe_zero(), fe_zero()};
    memcpy(num3, z, sizeof z); memcpy(den3, z, sizeof z);
    memcpy(num3, t->maps[k].num, t->maps[k].nnum * sizeof(fe));
    memcpy(den3, t->maps[k].den, t->maps[k].nden * sizeof(fe));
    return 0;
}
void* ORA(tree_new)(const void* leaves, size_t n, const void* nums3, const void* dens3) {
    int nm = (int)ilog2_sz(n);
    ratmap* maps = (ratmap*)calloc(nm ? nm : 1, sizeof(ratmap));
    for (int k = 0; k < nm; ++k) {
        memcpy(maps[k].num, (const fe*)nums3 + 3 * k, 3 * sizeof(fe)); maps[k].nnum = 3;
        memcpy(maps[k].den, (const fe*)dens3 + 3 * k, 3 * sizeof(fe)); maps[k].nden = 3;
    }
    fftree* t = tree_new((const fe*)leaves, n, maps, nm);
    free(maps);
    return t;
}
#define ORA_WRAP_OUT(expr, count)                                         \
    do { fe* r_ = (expr); if (!r_) return -1;                             \
         memcpy(out, r_, (count) * sizeof(fe)); fe_free(r_); return 0; } while (0)

int ORA(extend)(const void* t, const void* in, void* out, size_t e, int moiety) {
    if (!e || (e & (e - 1))) return -2;
    ORA_WRAP_OUT(tree_extend((const fftree*)t, (const fe*)in, e, moiety), e);
}
int ORA(mextend)(const void* t, const void* in, void* out, size_t e, int moiety) {
    if (!e || (e & (e - 1))) return -2;
    ORA_WRAP_OUT(tree_mextend((const fftree*)t, (const fe*)in, e, moiety), e);
}
int ORA(enter)(const void* t, const void* in, void* out, size_t n) {
    if (!n || (n & (n - 1))) return -2;
    ORA_WRAP_OUT(tree_enter((const fftree*)t, (const fe*)in, n), n);
}
int ORA(exit)(const void* t, const void* in, void* out, size_t n) {
    if (!n || (n & (n - 1))) return -2;
    ORA_WRAP_OUT(tree_exit((const fftree*)t, (const fe*)in, n), n);
}
int ORA(redc)(const void* tv, const void* in, const void* a, void* out, size_t n, int moiety) {
    if (!n || (n & (n - 1))) return -2;
    const fftree* t = subtree_with_size((const fftree*)tv, n); if (!t) return -1;
    ORA_WRAP_OUT(redc_impl(t, (const fe*)in, (const fe*)a, n, moiety), n);
}
int ORA(modular_reduce)(const void* t, const void* in, const void* a, const void* c, void* out, size_t n) {
    if (!n || (n & (n - 1))) return -2;
    ORA_WRAP_OUT(tree_modular_reduce((const fftree*)t, (const fe*)in, (const fe*)a, (const fe*)c, n), n);
}
int ORA(vanish)(const void* t, const void* dom, void* out, size_t n) {
    if (!n || (n & (n - 1))) return -2;
    ORA_WRAP_OUT(tree_vanish((const fftree*)t, (const fe*)dom, n), 2 * n);
}
long ORA(degree)(const void* tv, const void* in, size_t n) {
    if (!n || (n & (n - 1))) return -2;
    const fftree* t = subtree_with_size((const fftree*)tv, n); if (!t) return -1;
    return (long)degree_impl(t, (const fe*)in, n);
}
/* naive evaluation of sum c_j x^j — the reference tests' expected side (DensePolynomial::evaluate) */
void ORA(horner)(const void* coeffs, size_t n, const void* xs, size_t nx, void* out) {
    for (size_t i = 0; i < nx; ++i) ((fe*)out)[i] = poly_eval((const fe*)coeffs, (int)n, ((const fe*)xs)[i]);
}
void ORA(mul_vec)(const void* a, const void* b, void* out, size_t n) {
    for (size_t i = 0; i < n; ++i) ((fe*)out)[i] = fe_mul(((const fe*)a)[i], ((const fe*)b)[i]);
}
void ORA(add_vec)(const void* a, const void* b, void* out, size_t n) {
    for (size_t i = 0; i < n; ++i) ((fe*)out)[i] = fe_add(((const fe*)a)[i], ((const fe*)b)[i]);
}
void ORA(sub_vec)(const void* a, const void* b, void* out, size_t n) {
    for (size_t i = 0; i < n; ++i) ((fe*)out)[i] = fe_sub(((const fe*)a)[i], ((const fe*)b)[i]);
}
void ORA(inv_vec)(const void* a, void* out, size_t n) {
    for (size_t i = 0; i < n; ++i) ((fe*)out)[i] = fe_inv(((const fe*)a)[i]);
}
