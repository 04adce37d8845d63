/*
 * oracle/cpu_ref.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED (see oracle/bn256_ref.py header and DESIGN.md): the reference repository
 * (/root/reference) holds no MSM/FFT fixtures, and the implementing crate -- halo2_proofs git tag
 * v2023_02_02 (/root/reference/Cargo.toml:10) with its halo2curves dependency -- is not in this
 * container and cannot be built (no Rust toolchain, no network).  This file restates that crate's
 * published algorithm from SURVEY.md §3.3 / §3.4 [UPSTREAM-RECALLED]:
 *
 *   best_multiexp / multiexp_serial   (halo2_proofs/src/arithmetic.rs)
 *       n > T: split into chunks of n/T, one thread each; per chunk c = 1 (m<4), 3 (m<32),
 *       else ceil(ln m); segments = 256/c + 1, MSB first; c doublings per segment; 2^c - 1
 *       buckets of {None, Affine, Projective}; zero-digit skip; running-sum reduction;
 *       chunk results summed serially.
 *   best_fft / recursive_butterfly_arithmetic
 *       serial bit-reversal swap; serial n/2-entry twiddle table; iterative stages when
 *       log_n <= log2(T), otherwise recursive radix-2 DIT with a two-way fork per level and a
 *       serial butterfly sweep per task.
 *
 * and the BN256 field/curve types of halo2curves (4 x u64 Montgomery limbs, y^2 = x^3 + 3,
 * Jacobian G1).  Reference call sites of the path: /root/reference/src/circuits/utils.rs:28-48.
 * Outputs are canonical (affine point / fully reduced Montgomery limbs), so any correct
 * implementation is bit-identical to upstream's.  It is validated against oracle/bn256_ref.py
 * (independent big-integer mathematics) in tests/test_oracle.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * It doubles as the timed CPU baseline ("port") and states the thread count it used.
 *
 * Build: gcc -O3 -march=native -shared -fPIC -pthread -o oracle/libcpu_ref.so oracle/cpu_ref.c -lm
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;          /* field element, Montgomery form */
typedef struct { fe x, y; } g1a;               /* affine; identity = (0,0) */
typedef struct { fe x, y, z; } g1j;            /* Jacobian; identity has z = 0 */

typedef struct { fe mod; uint64_t inv; fe r; fe r2; } field;

static const field FQ = {
    {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}},
    0x87d20782e4866389ULL,
    {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}},
    {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}}};
static const field FR = {
    {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}},
    0xc2e1f593efffffffULL,
    {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}},
    {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}}};

/* ------------------------------------------------------------------ field arithmetic ------- */
static inline int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe *a, const fe *b) {
    return ((a->l[0] ^ b->l[0]) | (a->l[1] ^ b->l[1]) | (a->l[2] ^ b->l[2]) | (a->l[3] ^ b->l[3])) == 0;
}
static inline int fe_geq(const fe *a, const fe *b) {
    for (int i = 3; i >= 0; --i) {
        if (a->l[i] > b->l[i]) return 1;
        if (a->l[i] < b->l[i]) return 0;
    }
    return 1;
}
static inline void fe_sub_raw(fe *o, const fe *a, const fe *b) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 t = (u128)a->l[i] - b->l[i] - br;
        o->l[i] = (uint64_t)t;
        br = (t >> 64) & 1;
    }
}
static inline void fe_add(const field *F, fe *o, const fe *a, const fe *b) {
    u128 c = 0;
    fe t;
    for (int i = 0; i < 4; ++i) {
        c += (u128)a->l[i] + b->l[i];
        t.l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (c || fe_geq(&t, &F->mod)) fe_sub_raw(&t, &t, &F->mod);
    *o = t;
}
static inline void fe_sub(const field *F, fe *o, const fe *a, const fe *b) {
    fe t;
    if (fe_geq(a, b)) {
        fe_sub_raw(&t, a, b);
    } else {
        fe u;
        fe_sub_raw(&u, b, a);
        fe_sub_raw(&t, &F->mod, &u);
    }
    *o = t;
}
static inline void fe_neg(const field *F, fe *o, const fe *a) {
    if (fe_is_zero(a)) { *o = *a; return; }
    fe_sub_raw(o, &F->mod, a);
}
static inline void fe_dbl(const field *F, fe *o, const fe *a) { fe_add(F, o, a, a); }

/* CIOS Montgomery multiplication, 4 x 64-bit limbs (scalar temporaries so they stay in registers) */
static inline __attribute__((always_inline)) void fe_mul(const field *F, fe *o, const fe *a, const fe *b) {
    const uint64_t m0 = F->mod.l[0], m1 = F->mod.l[1], m2 = F->mod.l[2], m3 = F->mod.l[3], inv = F->inv;
    const uint64_t a0 = a->l[0], a1 = a->l[1], a2 = a->l[2], a3 = a->l[3];
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#pragma GCC unroll 4
    for (int i = 0; i < 4; ++i) {
        const uint64_t bi = b->l[i];
        u128 c = (u128)a0 * bi + t0; t0 = (uint64_t)c; c >>= 64;
        c += (u128)a1 * bi + t1; t1 = (uint64_t)c; c >>= 64;
        c += (u128)a2 * bi + t2; t2 = (uint64_t)c; c >>= 64;
        c += (u128)a3 * bi + t3; t3 = (uint64_t)c; c >>= 64;
        c += t4; t4 = (uint64_t)c;
        const uint64_t t5 = (uint64_t)(c >> 64);
        const uint64_t m = t0 * inv;
        c = (u128)m * m0 + t0; c >>= 64;
        c += (u128)m * m1 + t1; t0 = (uint64_t)c; c >>= 64;
        c += (u128)m * m2 + t2; t1 = (uint64_t)c; c >>= 64;
        c += (u128)m * m3 + t3; t2 = (uint64_t)c; c >>= 64;
        c += t4; t3 = (uint64_t)c; t4 = t5 + (uint64_t)(c >> 64);
    }
    fe r = {{t0, t1, t2, t3}};
    if (t4 || fe_geq(&r, &F->mod)) fe_sub_raw(&r, &r, &F->mod);
    *o = r;
}
static inline void fe_sqr(const field *F, fe *o, const fe *a) { fe_mul(F, o, a, a); }
static void fe_from_mont(const field *F, fe *o, const fe *a) {
    fe one = {{1, 0, 0, 0}};
    fe_mul(F, o, a, &one);
}
static void fe_to_mont(const field *F, fe *o, const fe *a) { fe_mul(F, o, a, &F->r2); }
static void fe_pow(const field *F, fe *o, const fe *a, const uint64_t e[4]) {
    fe acc = F->r;
    for (int i = 255; i >= 0; --i) {
        fe_sqr(F, &acc, &acc);
        if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(F, &acc, &acc, a);
    }
    *o = acc;
}
static void fe_inv(const field *F, fe *o, const fe *a) {
    fe two = {{2, 0, 0, 0}}, e;
    fe_sub_raw(&e, &F->mod, &two);
    fe_pow(F, o, a, e.l);
}

/* ------------------------------------------------------------------ curve arithmetic ------- */
static inline int g1a_is_identity(const g1a *p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static inline int g1j_is_identity(const g1j *p) { return fe_is_zero(&p->z); }
static inline void g1j_set_identity(g1j *p) { memset(p, 0, sizeof *p); }
static inline void g1j_from_affine(g1j *o, const g1a *p) {
    if (g1a_is_identity(p)) { g1j_set_identity(o); return; }
    o->x = p->x; o->y = p->y; o->z = FQ.r;
}

/* dbl-2009-l, a = 0 */
static void g1j_double(g1j *o, const g1j *p) {
    if (g1j_is_identity(p)) { g1j_set_identity(o); return; }
    fe A, B, C, D, E, Fv, t, x3, y3, z3;
    fe_sqr(&FQ, &A, &p->x);
    fe_sqr(&FQ, &B, &p->y);
    fe_sqr(&FQ, &C, &B);
    fe_add(&FQ, &t, &p->x, &B);
    fe_sqr(&FQ, &t, &t);
    fe_sub(&FQ, &t, &t, &A);
    fe_sub(&FQ, &t, &t, &C);
    fe_dbl(&FQ, &D, &t);
    fe_dbl(&FQ, &E, &A);
    fe_add(&FQ, &E, &E, &A);
    fe_sqr(&FQ, &Fv, &E);
    fe_dbl(&FQ, &t, &D);
    fe_sub(&FQ, &x3, &Fv, &t);
    fe_mul(&FQ, &z3, &p->y, &p->z);
    fe_dbl(&FQ, &z3, &z3);
    fe_sub(&FQ, &t, &D, &x3);
    fe_mul(&FQ, &y3, &E, &t);
    fe_dbl(&FQ, &C, &C); fe_dbl(&FQ, &C, &C); fe_dbl(&FQ, &C, &C);
    fe_sub(&FQ, &y3, &y3, &C);
    o->x = x3; o->y = y3; o->z = z3;
}

/* add-2007-bl with the exceptional cases handled */
static void g1j_add(g1j *o, const g1j *p, const g1j *q) {
    if (g1j_is_identity(p)) { *o = *q; return; }
    if (g1j_is_identity(q)) { *o = *p; return; }
    fe z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t, x3, y3, z3;
    fe_sqr(&FQ, &z1z1, &p->z);
    fe_sqr(&FQ, &z2z2, &q->z);
    fe_mul(&FQ, &u1, &p->x, &z2z2);
    fe_mul(&FQ, &u2, &q->x, &z1z1);
    fe_mul(&FQ, &s1, &p->y, &q->z); fe_mul(&FQ, &s1, &s1, &z2z2);
    fe_mul(&FQ, &s2, &q->y, &p->z); fe_mul(&FQ, &s2, &s2, &z1z1);
    if (fe_eq(&u1, &u2)) {
        if (fe_eq(&s1, &s2)) { g1j_double(o, p); return; }
        g1j_set_identity(o);
        return;
    }
    fe_sub(&FQ, &h, &u2, &u1);
    fe_dbl(&FQ, &i, &h); fe_sqr(&FQ, &i, &i);
    fe_mul(&FQ, &j, &h, &i);
    fe_sub(&FQ, &r, &s2, &s1); fe_dbl(&FQ, &r, &r);
    fe_mul(&FQ, &v, &u1, &i);
    fe_sqr(&FQ, &x3, &r);
    fe_sub(&FQ, &x3, &x3, &j);
    fe_dbl(&FQ, &t, &v);
    fe_sub(&FQ, &x3, &x3, &t);
    fe_sub(&FQ, &t, &v, &x3);
    fe_mul(&FQ, &y3, &r, &t);
    fe_mul(&FQ, &t, &s1, &j); fe_dbl(&FQ, &t, &t);
    fe_sub(&FQ, &y3, &y3, &t);
    fe_add(&FQ, &z3, &p->z, &q->z); fe_sqr(&FQ, &z3, &z3);
    fe_sub(&FQ, &z3, &z3, &z1z1); fe_sub(&FQ, &z3, &z3, &z2z2);
    fe_mul(&FQ, &z3, &z3, &h);
    o->x = x3; o->y = y3; o->z = z3;
}

/* madd-2007-bl (q affine) with the exceptional cases handled */
static void g1j_add_mixed(g1j *o, const g1j *p, const g1a *q) {
    if (g1a_is_identity(q)) { *o = *p; return; }
    if (g1j_is_identity(p)) { g1j_from_affine(o, q); return; }
    fe z1z1, u2, s2, h, hh, i, j, r, v, t, x3, y3, z3;
    fe_sqr(&FQ, &z1z1, &p->z);
    fe_mul(&FQ, &u2, &q->x, &z1z1);
    fe_mul(&FQ, &s2, &q->y, &p->z); fe_mul(&FQ, &s2, &s2, &z1z1);
    if (fe_eq(&p->x, &u2)) {
        if (fe_eq(&p->y, &s2)) { g1j_double(o, p); return; }
        g1j_set_identity(o);
        return;
    }
    fe_sub(&FQ, &h, &u2, &p->x);
    fe_sqr(&FQ, &hh, &h);
    fe_dbl(&FQ, &i, &hh); fe_dbl(&FQ, &i, &i);
    fe_mul(&FQ, &j, &h, &i);
    fe_sub(&FQ, &r, &s2, &p->y); fe_dbl(&FQ, &r, &r);
    fe_mul(&FQ, &v, &p->x, &i);
    fe_sqr(&FQ, &x3, &r);
    fe_sub(&FQ, &x3, &x3, &j);
    fe_dbl(&FQ, &t, &v);
    fe_sub(&FQ, &x3, &x3, &t);
    fe_sub(&FQ, &t, &v, &x3);
    fe_mul(&FQ, &y3, &r, &t);
    fe_mul(&FQ, &t, &p->y, &j); fe_dbl(&FQ, &t, &t);
    fe_sub(&FQ, &y3, &y3, &t);
    fe_add(&FQ, &z3, &p->z, &h); fe_sqr(&FQ, &z3, &z3);
    fe_sub(&FQ, &z3, &z3, &z1z1); fe_sub(&FQ, &z3, &z3, &hh);
    o->x = x3; o->y = y3; o->z = z3;
}

static void g1j_to_affine(g1a *o, const g1j *p) {
    if (g1j_is_identity(p)) { memset(o, 0, sizeof *o); return; }
    fe zi, zi2, zi3;
    fe_inv(&FQ, &zi, &p->z);
    fe_sqr(&FQ, &zi2, &zi);
    fe_mul(&FQ, &zi3, &zi2, &zi);
    fe_mul(&FQ, &o->x, &p->x, &zi2);
    fe_mul(&FQ, &o->y, &p->y, &zi3);
}

/* ------------------------------------------------------------------ best_multiexp ---------- */
enum { BK_NONE = 0, BK_AFFINE = 1, BK_PROJ = 2 };
typedef struct { int tag; g1j p; } bucket;   /* affine state keeps (x,y) in p.x, p.y */

static inline void bucket_add_assign(bucket *b, const g1a *q) {
    if (b->tag == BK_NONE) {
        b->tag = BK_AFFINE; b->p.x = q->x; b->p.y = q->y;
    } else if (b->tag == BK_AFFINE) {
        g1a a = {b->p.x, b->p.y};
        g1j t;
        g1j_from_affine(&t, &a);
        g1j_add_mixed(&b->p, &t, q);
        b->tag = BK_PROJ;
    } else {
        g1j_add_mixed(&b->p, &b->p, q);
    }
}
static inline void bucket_add_to(const bucket *b, g1j *other) {   /* other = bucket + other */
    if (b->tag == BK_NONE) return;
    if (b->tag == BK_AFFINE) {
        g1a a = {b->p.x, b->p.y};
        g1j_add_mixed(other, other, &a);
    } else {
        g1j_add(other, other, &b->p);
    }
}

static inline unsigned get_at(unsigned seg, unsigned c, const uint8_t bytes[32]) {
    unsigned skip_bits = seg * c, skip_bytes = skip_bits / 8;
    if (skip_bytes >= 32) return 0;
    uint64_t v = 0;
    for (unsigned i = 0; i < 8 && skip_bytes + i < 32; ++i) v |= (uint64_t)bytes[skip_bytes + i] << (8 * i);
    v >>= skip_bits - skip_bytes * 8;
    return (unsigned)(v % (1ULL << c));
}

static void multiexp_serial(const fe *coeffs, const g1a *bases, size_t m, g1j *acc) {
    unsigned c = m < 4 ? 1 : m < 32 ? 3 : (unsigned)ceil(log((double)m));
    unsigned segments = 256 / c + 1;
    uint8_t (*reprs)[32] = malloc(m * 32 + 32);
    for (size_t i = 0; i < m; ++i) {     /* to_repr(): Montgomery -> canonical LE bytes */
        fe t;
        fe_from_mont(&FR, &t, &coeffs[i]);
        memcpy(reprs[i], t.l, 32);
    }
    size_t nb = ((size_t)1 << c) - 1;
    bucket *buckets = malloc(nb * sizeof(bucket));
    for (int seg = (int)segments - 1; seg >= 0; --seg) {
        for (unsigned k = 0; k < c; ++k) g1j_double(acc, acc);
        for (size_t b = 0; b < nb; ++b) buckets[b].tag = BK_NONE;
        for (size_t i = 0; i < m; ++i) {
            unsigned d = get_at((unsigned)seg, c, reprs[i]);
            if (d != 0) bucket_add_assign(&buckets[d - 1], &bases[i]);
        }
        g1j running;
        g1j_set_identity(&running);
        for (size_t b = nb; b-- > 0;) {
            bucket_add_to(&buckets[b], &running);
            g1j_add(acc, acc, &running);
        }
    }
    free(buckets);
    free(reprs);
}

typedef struct { const fe *coeffs; const g1a *bases; size_t m; g1j acc; } msm_task;
static void *msm_worker(void *arg) {
    msm_task *t = arg;
    g1j_set_identity(&t->acc);
    multiexp_serial(t->coeffs, t->bases, t->m, &t->acc);
    return NULL;
}

/* scalars: n x 4 u64 Montgomery Fr; bases: n x 8 u64 Montgomery affine; out: 12 u64 Jacobian */
int ref_best_multiexp(const uint64_t *scalars, const uint64_t *bases, size_t n, int threads, uint64_t *out_jac) {
    const fe *cs = (const fe *)scalars;
    const g1a *bs = (const g1a *)bases;
    g1j acc;
    g1j_set_identity(&acc);
    if (threads < 1) threads = 1;
    if (n > (size_t)threads) {
        size_t chunk = n / (size_t)threads;
        size_t ntask = (n + chunk - 1) / chunk;
        msm_task *tasks = calloc(ntask, sizeof *tasks);
        pthread_t *th = calloc(ntask, sizeof *th);
        for (size_t t = 0; t < ntask; ++t) {
            size_t lo = t * chunk, hi = lo + chunk > n ? n : lo + chunk;
            tasks[t].coeffs = cs + lo; tasks[t].bases = bs + lo; tasks[t].m = hi - lo;
            pthread_create(&th[t], NULL, msm_worker, &tasks[t]);
        }
        for (size_t t = 0; t < ntask; ++t) {
            pthread_join(th[t], NULL);
            g1j_add(&acc, &acc, &tasks[t].acc);
        }
        free(tasks); free(th);
    } else {
        multiexp_serial(cs, bs, n, &acc);
    }
    memcpy(out_jac, &acc, sizeof acc);
    return 0;
}

/* ------------------------------------------------------------------ best_fft --------------- */
static inline uint32_t bitreverse32(uint32_t n, uint32_t l) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < l; ++i) { r = (r << 1) | (n & 1); n >>= 1; }
    return r;
}

typedef struct { fe *a; size_t n; size_t tc; const fe *tw; int fork_levels; } fft_task;
static void recursive_butterfly(fe *a, size_t n, size_t tc, const fe *tw, int fork_levels);
static void *fft_worker(void *arg) {
    fft_task *t = arg;
    recursive_butterfly(t->a, t->n, t->tc, t->tw, t->fork_levels);
    return NULL;
}
static void recursive_butterfly(fe *a, size_t n, size_t tc, const fe *tw, int fork_levels) {
    if (n == 2) {
        fe t = a[1];
        fe_sub(&FR, &a[1], &a[0], &t);
        fe_add(&FR, &a[0], &a[0], &t);
        return;
    }
    fe *left = a, *right = a + n / 2;
    if (fork_levels > 0) {   /* rayon::join: the two halves run concurrently */
        fft_task rt = {right, n / 2, tc * 2, tw, fork_levels - 1};
        pthread_t th;
        pthread_create(&th, NULL, fft_worker, &rt);
        recursive_butterfly(left, n / 2, tc * 2, tw, fork_levels - 1);
        pthread_join(th, NULL);
    } else {
        recursive_butterfly(left, n / 2, tc * 2, tw, 0);
        recursive_butterfly(right, n / 2, tc * 2, tw, 0);
    }
    /* twiddle = 1 */
    fe t = right[0];
    fe_sub(&FR, &right[0], &left[0], &t);
    fe_add(&FR, &left[0], &left[0], &t);
    for (size_t i = 1; i < n / 2; ++i) {
        fe_mul(&FR, &t, &right[i], &tw[i * tc]);
        fe_sub(&FR, &right[i], &left[i], &t);
        fe_add(&FR, &left[i], &left[i], &t);
    }
}

/* a: n x 4 u64 Montgomery Fr, in place; omega: 4 u64 Montgomery */
int ref_best_fft(uint64_t *a_, const uint64_t *omega_, uint32_t log_n, int threads) {
    fe *a = (fe *)a_;
    fe omega;
    memcpy(&omega, omega_, sizeof omega);
    size_t n = (size_t)1 << log_n;
    if (threads < 1) threads = 1;
    int log_threads = 0;
    while ((2 << log_threads) <= threads) ++log_threads;
    for (size_t k = 0; k < n; ++k) {
        size_t rk = bitreverse32((uint32_t)k, log_n);
        if (k < rk) { fe t = a[rk]; a[rk] = a[k]; a[k] = t; }
    }
    if (log_n == 0) return 0;
    size_t half = n / 2;
    fe *tw = malloc((half ? half : 1) * sizeof(fe));
    fe w = FR.r;
    for (size_t i = 0; i < half; ++i) { tw[i] = w; fe_mul(&FR, &w, &w, &omega); }
    if ((int)log_n <= log_threads) {
        size_t chunk = 2, tchunk = n / 2;
        for (uint32_t s = 0; s < log_n; ++s) {
            for (size_t base = 0; base < n; base += chunk) {
                fe *left = a + base, *right = a + base + chunk / 2;
                fe t = right[0];
                fe_sub(&FR, &right[0], &left[0], &t);
                fe_add(&FR, &left[0], &left[0], &t);
                for (size_t i = 1; i < chunk / 2; ++i) {
                    fe_mul(&FR, &t, &right[i], &tw[i * tchunk]);
                    fe_sub(&FR, &right[i], &left[i], &t);
                    fe_add(&FR, &left[i], &left[i], &t);
                }
            }
            chunk *= 2; tchunk /= 2;
        }
    } else {
        recursive_butterfly(a, n, 1, tw, log_threads);
    }
    free(tw);
    return 0;
}

/* ------------------------------------------------------------------ helpers for the tests -- */
void ref_fr_mul(const uint64_t *a, const uint64_t *b, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; ++i) fe_mul(&FR, (fe *)o + i, (const fe *)a + i, (const fe *)b + i);
}
void ref_fq_mul(const uint64_t *a, const uint64_t *b, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; ++i) fe_mul(&FQ, (fe *)o + i, (const fe *)a + i, (const fe *)b + i);
}
void ref_fr_add(const uint64_t *a, const uint64_t *b, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; ++i) fe_add(&FR, (fe *)o + i, (const fe *)a + i, (const fe *)b + i);
}
void ref_fr_sub(const uint64_t *a, const uint64_t *b, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; ++i) fe_sub(&FR, (fe *)o + i, (const fe *)a + i, (const fe *)b + i);
}
void ref_fr_from_mont(const uint64_t *a, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; ++i) fe_from_mont(&FR, (fe *)o + i, (const fe *)a + i);
}
void ref_fr_to_mont(const uint64_t *a, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; ++i) fe_to_mont(&FR, (fe *)o + i, (const fe *)a + i);
}
void ref_fr_inv(const uint64_t *a, uint64_t *o, size_t n) {
    for (size_t i = 0; i < n; ++i) fe_inv(&FR, (fe *)o + i, (const fe *)a + i);
}
/* Horner evaluation of sum coeffs[i] x^i (all Montgomery Fr) */
void ref_fr_horner(const uint64_t *coeffs, size_t n, const uint64_t *x, uint64_t *out) {
    fe acc = {{0, 0, 0, 0}};
    for (size_t i = n; i-- > 0;) {
        fe_mul(&FR, &acc, &acc, (const fe *)x);
        fe_add(&FR, &acc, &acc, (const fe *)coeffs + i);
    }
    memcpy(out, &acc, sizeof acc);
}
/* dot product sum a[i]*b[i] in Fr (Montgomery in, Montgomery out) */
void ref_fr_dot(const uint64_t *a, const uint64_t *b, size_t n, uint64_t *out) {
    fe acc = {{0, 0, 0, 0}}, t;
    for (size_t i = 0; i < n; ++i) {
        fe_mul(&FR, &t, (const fe *)a + i, (const fe *)b + i);
        fe_add(&FR, &acc, &acc, &t);
    }
    memcpy(out, &acc, sizeof acc);
}
/* powers[i] = base^i, Montgomery */
void ref_fr_powers(const uint64_t *base, size_t n, uint64_t *out) {
    fe w = FR.r;
    for (size_t i = 0; i < n; ++i) { ((fe *)out)[i] = w; fe_mul(&FR, &w, &w, (const fe *)base); }
}
/* out_affine = [k]P by double-and-add; k Montgomery Fr, P Montgomery affine */
void ref_g1_mul(const uint64_t *k_mont, const uint64_t *p_affine, uint64_t *out_affine) {
    fe k;
    fe_from_mont(&FR, &k, (const fe *)k_mont);
    g1j acc;
    g1j_set_identity(&acc);
    for (int i = 255; i >= 0; --i) {
        g1j_double(&acc, &acc);
        if ((k.l[i >> 6] >> (i & 63)) & 1) g1j_add_mixed(&acc, &acc, (const g1a *)p_affine);
    }
    g1j_to_affine((g1a *)out_affine, &acc);
}
void ref_g1_to_affine(const uint64_t *jac, uint64_t *out_affine, size_t n) {
    for (size_t i = 0; i < n; ++i) g1j_to_affine((g1a *)out_affine + i, (const g1j *)jac + i);
}
/* out_jac = sum of n Jacobian points */
void ref_g1_sum(const uint64_t *jac, size_t n, uint64_t *out_jac) {
    g1j acc;
    g1j_set_identity(&acc);
    for (size_t i = 0; i < n; ++i) g1j_add(&acc, &acc, (const g1j *)jac + i);
    memcpy(out_jac, &acc, sizeof acc);
}
/* out_affine = a + b (affine inputs) */
void ref_g1_add_affine(const uint64_t *a, const uint64_t *b, uint64_t *out_affine) {
    g1j t;
    g1j_from_affine(&t, (const g1a *)a);
    g1j_add_mixed(&t, &t, (const g1a *)b);
    g1j_to_affine((g1a *)out_affine, &t);
}
/* naive MSM by double-and-add: the definition (slow; small n only) */
void ref_msm_naive(const uint64_t *scalars, const uint64_t *bases, size_t n, uint64_t *out_jac) {
    g1j acc;
    g1j_set_identity(&acc);
    for (size_t i = 0; i < n; ++i) {
        fe k;
        fe_from_mont(&FR, &k, (const fe *)scalars + i);
        g1j t;
        g1j_set_identity(&t);
        for (int b = 255; b >= 0; --b) {
            g1j_double(&t, &t);
            if ((k.l[b >> 6] >> (b & 63)) & 1) g1j_add_mixed(&t, &t, (const g1a *)bases + i);
        }
        g1j_add(&acc, &acc, &t);
    }
    memcpy(out_jac, &acc, sizeof acc);
}
/* bases[i] = [s^i]G (affine, Montgomery): the KZG SRS of ParamsKZG::setup, serial */
void ref_srs(const uint64_t *s_mont, size_t n, uint64_t *out_affine) {
    fe pw = FR.r;
    g1a gen;
    fe one = {{1, 0, 0, 0}}, two = {{2, 0, 0, 0}};
    fe_to_mont(&FQ, &gen.x, &one);
    fe_to_mont(&FQ, &gen.y, &two);
    for (size_t i = 0; i < n; ++i) {
        ref_g1_mul(pw.l, (const uint64_t *)&gen, out_affine + 8 * i);
        fe_mul(&FR, &pw, &pw, (const fe *)s_mont);
    }
}
