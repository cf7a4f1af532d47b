/* ecm_oracle.c — TEST INFRASTRUCTURE ONLY (see ecm_oracle.h).
 *
 * Scalar restatement of the reference's hot path, one curve at a time.  Arithmetic is on
 * DIGITBITS-bit limbs held in uint64_t with unsigned __int128 products; host-side big-integer
 * work (inversions, gcd, hex) uses GMP exactly where the reference does.  Compile with
 * -ffp-contract=off: the PRAC multiplier choice depends on plain IEEE double rounding
 * (ecm.c:486, 584).
 */
#include "ecm_oracle.h"
#include <gmp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef unsigned __int128 u128;
typedef uint64_t fe_t[ORC_MAXW];

struct orc_ctx {
    int digitbits, nwords, maxbits, nbits;
    uint64_t mask;
    fe_t n, one;       /* N and R mod N (monty->n, monty->one; main.c:629-634) */
    uint64_t rho;      /* -N^-1 mod 2^DIGITBITS (main.c:627-628, 636-640) */
    mpz_t N;
};

typedef struct {
    fe_t X, Z;
} orc_pt;

/* ecm_work, avx_ecm.h:218-262 (scalar) */
typedef struct {
    fe_t sum1, diff1, sum2, diff2, tt1, tt2, tt3, tt4, tt5, s;
    orc_pt pt1, pt2, pt3, pt4, pt5;
    uint32_t *map;
    orc_pt *Pa, *Pb;
    fe_t *Pa_inv, *Paprod, *Pbprod;
    orc_pt Pad, Pdnorm;
    fe_t stg2acc;
    uint32_t paired, ptadds, ptdups, numinv;
    uint64_t A;
    uint32_t amin, U, L, D, R;
    int found_during_inv;
    mpz_t inv_factor;
} orc_work;

/* ------------------------------------------------------------------------------------------- */
static void fe_from_mpz(const orc_ctx *c, uint64_t *r, const mpz_t x)
{
    /* insert_mpz_to_vec / broadcast_mpz_to_vec, main.c:95-138 (all NWORDS limbs written) */
    mpz_t t;
    mpz_init_set(t, x);
    for (int i = 0; i < c->nwords; i++) {
        r[i] = mpz_get_ui(t) & c->mask;
        mpz_tdiv_q_2exp(t, t, (mp_bitcnt_t)c->digitbits);
    }
    mpz_clear(t);
}

static void fe_to_mpz(const orc_ctx *c, mpz_t x, const uint64_t *a)
{
    /* extract_bignum_from_vec_to_mpz, main.c:63-93 */
    mpz_set_ui(x, 0);
    for (int i = c->nwords - 1; i >= 0; i--) {
        mpz_mul_2exp(x, x, (mp_bitcnt_t)c->digitbits);
        mpz_add_ui(x, x, a[i]);
    }
}

static void fe_copy(const orc_ctx *c, uint64_t *r, const uint64_t *a)
{
    memcpy(r, a, sizeof(uint64_t) * (size_t)c->nwords);   /* vecCopy, vec_common.c:57-65 */
}

orc_ctx *orc_create(const char *n_str, int digitbits)
{
    if (digitbits != 52 && digitbits != 32) return NULL;
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof *c);
    mpz_init(c->N);
    int base = (n_str[0] == '0' && (n_str[1] == 'x' || n_str[1] == 'X')) ? 16 : 10;
    if (mpz_set_str(c->N, base == 16 ? n_str + 2 : n_str, base) || mpz_even_p(c->N) || mpz_cmp_ui(c->N, 3) < 0) {
        mpz_clear(c->N);
        free(c);
        return NULL;
    }
    c->digitbits = digitbits;
    c->mask = digitbits == 52 ? 0xfffffffffffffULL : 0xffffffffULL;
    c->nbits = (int)mpz_sizeinbase(c->N, 2);
    /* main.c:465-483 */
    int step = digitbits == 52 ? 208 : 128;
    c->maxbits = step;
    while (c->maxbits <= c->nbits) c->maxbits += step;
    c->nwords = c->maxbits / digitbits;
    if (c->nwords > ORC_MAXW) {
        mpz_clear(c->N);
        free(c);
        return NULL;
    }
    /* main.c:620-640 */
    mpz_t r, nhat;
    mpz_init(r);
    mpz_init(nhat);
    mpz_set_ui(r, 1);
    mpz_mul_2exp(r, r, (mp_bitcnt_t)(digitbits * c->nwords));
    mpz_invert(nhat, c->N, r);
    mpz_sub(nhat, r, nhat);
    c->rho = mpz_get_ui(nhat) & c->mask;
    fe_from_mpz(c, c->n, c->N);
    mpz_tdiv_r(r, r, c->N);
    fe_from_mpz(c, c->one, r);
    mpz_clear(r);
    mpz_clear(nhat);
    return c;
}

void orc_destroy(orc_ctx *c)
{
    if (!c) return;
    mpz_clear(c->N);
    free(c);
}

int orc_nwords(const orc_ctx *c) { return c->nwords; }
int orc_maxbits(const orc_ctx *c) { return c->maxbits; }

/* ------------------------------------------------------------------------------------------- L0
 * c = a*b*R^-1 mod N, canonical in [0,N): the contract of vecmulmod52 (vecarith52.c:2438-3074;
 * per-column Montgomery digit s_j = lo(acc*rho) :2659; final t-N with borrow, keep t iff it
 * borrowed :3048-3070) and of vecmulmod (vecarith.c:221-887).  Word-serial (CIOS) evaluation of
 * the same sums; the result is the unique canonical residue either way. */
static void orc_mulmod(const orc_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out)
{
    const int n = c->nwords, db = c->digitbits;
    const uint64_t mask = c->mask;
    uint64_t t[ORC_MAXW + 2];
    memset(t, 0, sizeof(uint64_t) * (size_t)(n + 2));
    for (int i = 0; i < n; i++) {
        u128 cy = 0;
        for (int j = 0; j < n; j++) {
            u128 cur = (u128)a[j] * b[i] + t[j] + cy;
            t[j] = (uint64_t)cur & mask;
            cy = cur >> db;
        }
        u128 cur = (u128)t[n] + cy;
        t[n] = (uint64_t)cur & mask;
        t[n + 1] += (uint64_t)(cur >> db);
        uint64_t m = (t[0] * c->rho) & mask;
        cy = ((u128)m * c->n[0] + t[0]) >> db;
        for (int j = 1; j < n; j++) {
            cur = (u128)m * c->n[j] + t[j] + cy;
            t[j - 1] = (uint64_t)cur & mask;
            cy = cur >> db;
        }
        cur = (u128)t[n] + cy;
        t[n - 1] = (uint64_t)cur & mask;
        t[n] = t[n + 1] + (uint64_t)(cur >> db);
        t[n + 1] = 0;
    }
    /* conditional subtract */
    uint64_t d[ORC_MAXW];
    uint64_t borrow = 0;
    for (int j = 0; j < n; j++) {
        uint64_t x = t[j] - c->n[j] - borrow;
        borrow = (x >> 63) & 1;
        d[j] = x & mask;
    }
    int ge = t[n] != 0 || !borrow;
    for (int j = 0; j < n; j++) out[j] = ge ? d[j] : t[j];
}

/* vecsqrmod52 (vecarith52.c:3317-4548) returns the same value as vecmulmod52(a,a) */
static void orc_sqrmod(const orc_ctx *c, const uint64_t *a, uint64_t *out) { orc_mulmod(c, a, a, out); }

/* vecaddmod52, vecarith52.c:4550-4611: c = a+b; if (carry or c >= N) c -= N */
static void orc_addmod(const orc_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out)
{
    const int n = c->nwords, db = c->digitbits;
    const uint64_t mask = c->mask;
    uint64_t s[ORC_MAXW], d[ORC_MAXW];
    uint64_t cy = 0, borrow = 0;
    for (int j = 0; j < n; j++) {
        uint64_t x = a[j] + b[j] + cy;
        s[j] = x & mask;
        cy = x >> db;
    }
    for (int j = 0; j < n; j++) {
        uint64_t x = s[j] - c->n[j] - borrow;
        borrow = (x >> 63) & 1;
        d[j] = x & mask;
    }
    int ge = cy || !borrow;
    for (int j = 0; j < n; j++) out[j] = ge ? d[j] : s[j];
}

/* vecsubmod52, vecarith52.c:4684-4723: c = a-b; if borrow c += N */
static void orc_submod(const orc_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out)
{
    const int n = c->nwords, db = c->digitbits;
    const uint64_t mask = c->mask;
    uint64_t d[ORC_MAXW];
    uint64_t borrow = 0, cy = 0;
    for (int j = 0; j < n; j++) {
        uint64_t x = a[j] - b[j] - borrow;
        borrow = (x >> 63) & 1;
        d[j] = x & mask;
    }
    if (borrow)
        for (int j = 0; j < n; j++) {
            uint64_t x = d[j] + c->n[j] + cy;
            d[j] = x & mask;
            cy = x >> db;
        }
    for (int j = 0; j < n; j++) out[j] = d[j];
}

/* vec_simul_addsub52, vecarith52.c:4877-4968 */
static void orc_addsubmod(const orc_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *sum, uint64_t *diff)
{
    fe_t s, d;
    orc_addmod(c, a, b, s);
    orc_submod(c, a, b, d);
    fe_copy(c, sum, s);
    fe_copy(c, diff, d);
}

int orc_l0_hex(orc_ctx *c, int op, const char *a_hex, const char *b_hex, char *out_hex)
{
    mpz_t x;
    fe_t a, b, r;
    mpz_init(x);
    mpz_set_str(x, a_hex, 16);
    fe_from_mpz(c, a, x);
    mpz_set_str(x, b_hex, 16);
    fe_from_mpz(c, b, x);
    switch (op) {
    case 0: orc_mulmod(c, a, b, r); break;
    case 1: orc_sqrmod(c, a, r); break;
    case 2: orc_addmod(c, a, b, r); break;
    case 3: orc_submod(c, a, b, r); break;
    default: mpz_clear(x); return -1;
    }
    fe_to_mpz(c, x, r);
    mpz_get_str(out_hex, 16, x);
    mpz_clear(x);
    return 0;
}

/* ------------------------------------------------------------------------------------------- L1 */
/* vec_add, ecm.c:407-443 */
static void orc_vec_add(const orc_ctx *c, orc_work *w, const orc_pt *Pin, orc_pt *Pout)
{
    orc_mulmod(c, w->diff1, w->sum2, w->tt1);          /* U */
    orc_mulmod(c, w->sum1, w->diff2, w->tt2);          /* V */
    orc_addsubmod(c, w->tt1, w->tt2, w->tt3, w->tt4);
    orc_sqrmod(c, w->tt3, w->tt1);                     /* (U+V)^2 */
    orc_sqrmod(c, w->tt4, w->tt2);                     /* (U-V)^2 */
    fe_t x, z;
    orc_mulmod(c, w->tt1, Pin->Z, x);                  /* Z * (U+V)^2 */
    orc_mulmod(c, w->tt2, Pin->X, z);                  /* X * (U-V)^2 */
    fe_copy(c, Pout->X, x);                            /* (the in-place branch :427-435 ends in the same state) */
    fe_copy(c, Pout->Z, z);
    w->ptadds++;
}

/* vec_duplicate, ecm.c:445-457 */
static void orc_vec_duplicate(const orc_ctx *c, orc_work *w, const uint64_t *insum, const uint64_t *indiff, orc_pt *P)
{
    orc_sqrmod(c, indiff, w->tt1);               /* V */
    orc_sqrmod(c, insum, w->tt2);                /* U */
    orc_mulmod(c, w->tt1, w->tt2, P->X);         /* X = U*V */
    orc_submod(c, w->tt2, w->tt1, w->tt3);       /* w = U-V */
    orc_mulmod(c, w->tt3, w->s, w->tt2);         /* t = s*w */
    orc_addmod(c, w->tt2, w->tt1, w->tt2);       /* t += V */
    orc_mulmod(c, w->tt2, w->tt3, P->Z);         /* Z = t*w */
    w->ptdups++;
}

#define ADD 5.5 /* ecm.c:459 */
#define DUP 4.5 /* ecm.c:460 */
#define NV 10
static const double val[NV] = {0.61803398874989485, 0.72360679774997897, 0.58017872829546410,
                               0.63283980608870629, 0.61242994950949500, 0.62018198080741576,
                               0.61721461653440386, 0.61834711965622806, 0.61791440652881789,
                               0.61807966846989581}; /* ecm.c:473-477 */

/* lucas_cost, ecm.c:479-563 (ORIG_PRAC undefined, ecm.c:467) */
double orc_lucas_cost(uint64_t n, double v)
{
    uint64_t d, e, r;
    double c;
    d = n;
    r = (uint64_t)((double)d * v + 0.5);
    if (r >= n) return (ADD * (double)n);
    d = n - r;
    e = 2 * r - n;
    c = DUP + ADD;
    while (d != e) {
        if (d < e) { r = d; d = e; e = r; }
        if ((d + 3) / 4 <= e) { d -= e; c += ADD; }
        else if ((d + e) % 2 == 0) { d = (d - e) / 2; c += ADD + DUP; }
        else if (d % 2 == 0) { d /= 2; c += ADD + DUP; }
        else { e /= 2; c += ADD + DUP; }
    }
    if (d != 1) return 999999999.;
    return c;
}

int orc_prac_choice(uint64_t c)
{
    int i = 0;
    double cmin = ADD * (double)c;        /* ecm.c:574 */
    for (int d = 0; d < NV; d++) {
        double cost = orc_lucas_cost(c, val[d]);
        if (cost < cmin) { cmin = cost; i = d; }   /* strict <, ecm.c:577 */
    }
    return i;
}

static void pt_swap(orc_pt *a, orc_pt *b)
{
    orc_pt t = *a;
    *a = *b;
    *b = t;
}

/* prac, ecm.c:565-884 */
static void orc_prac(const orc_ctx *c, orc_work *w, orc_pt *P, uint64_t cc)
{
    uint64_t d, e, r;
    int i = orc_prac_choice(cc);
    d = cc;
    r = (uint64_t)((double)d * val[i] + 0.5);   /* ecm.c:584 */
    d = cc - r;
    e = 2 * r - cc;
    w->pt1 = *P;                                /* ecm.c:603-608 */
    w->pt2 = *P;
    w->pt3 = *P;
    orc_submod(c, w->pt1.X, w->pt1.Z, w->diff1);
    orc_addmod(c, w->pt1.X, w->pt1.Z, w->sum1);
    orc_vec_duplicate(c, w, w->sum1, w->diff1, &w->pt1);   /* ecm.c:613 */
    while (d != e) {
        if (d < e) {                            /* ecm.c:617-630 */
            r = d; d = e; e = r;
            pt_swap(&w->pt1, &w->pt2);
        }
        if ((d + 3) / 4 <= e) {                 /* rule 3, ecm.c:683-713 */
            d -= e;
            orc_addsubmod(c, w->pt2.X, w->pt2.Z, w->sum1, w->diff1);
            orc_addsubmod(c, w->pt1.X, w->pt1.Z, w->sum2, w->diff2);
            orc_vec_add(c, w, &w->pt3, &w->pt4);
            orc_pt t = w->pt2;                  /* circular permutation (B,T,C) */
            w->pt2 = w->pt4;
            w->pt4 = w->pt3;
            w->pt3 = t;
        } else if ((d + e) % 2 == 0) {          /* rule 4, ecm.c:714-726 */
            d = (d - e) / 2;
            orc_addsubmod(c, w->pt2.X, w->pt2.Z, w->sum1, w->diff1);
            orc_addsubmod(c, w->pt1.X, w->pt1.Z, w->sum2, w->diff2);
            orc_vec_add(c, w, &w->pt3, &w->pt2);
            orc_vec_duplicate(c, w, w->sum2, w->diff2, &w->pt1);
        } else if (d % 2 == 0) {                /* rule 5, ecm.c:728-740 */
            d /= 2;
            orc_addsubmod(c, w->pt3.X, w->pt3.Z, w->sum1, w->diff1);
            orc_addsubmod(c, w->pt1.X, w->pt1.Z, w->sum2, w->diff2);
            orc_vec_add(c, w, &w->pt2, &w->pt3);
            orc_vec_duplicate(c, w, w->sum2, w->diff2, &w->pt1);
        } else {                                /* rule 9, ecm.c:853-865 */
            e /= 2;
            orc_addsubmod(c, w->pt3.X, w->pt3.Z, w->sum1, w->diff1);
            orc_addsubmod(c, w->pt2.X, w->pt2.Z, w->sum2, w->diff2);
            orc_vec_add(c, w, &w->pt1, &w->pt3);
            orc_vec_duplicate(c, w, w->sum2, w->diff2, &w->pt2);
        }
    }
    orc_submod(c, w->pt1.X, w->pt1.Z, w->diff1);   /* ecm.c:868-873 */
    orc_addmod(c, w->pt1.X, w->pt1.Z, w->sum1);
    orc_submod(c, w->pt2.X, w->pt2.Z, w->diff2);
    orc_addmod(c, w->pt2.X, w->pt2.Z, w->sum2);
    orc_vec_add(c, w, &w->pt3, P);
}

/* next_pt_vec, ecm.c:886-976 */
static void orc_next_pt(const orc_ctx *c, orc_work *w, orc_pt *P, uint64_t cc)
{
    uint64_t mask;
    if (cc == 1) return;
    w->pt1 = *P;
    orc_submod(c, P->X, P->Z, w->diff1);
    orc_addmod(c, P->X, P->Z, w->sum1);
    orc_vec_duplicate(c, w, w->sum1, w->diff1, &w->pt2);
    if (cc == 2) { *P = w->pt2; return; }
    mask = 1ULL << (64 - __builtin_clzll(cc) - 2);
    while (mask > 0) {
        orc_addsubmod(c, w->pt2.X, w->pt2.Z, w->sum2, w->diff2);
        orc_addsubmod(c, w->pt1.X, w->pt1.Z, w->sum1, w->diff1);
        if (cc & mask) {
            orc_vec_add(c, w, P, &w->pt1);
            orc_vec_duplicate(c, w, w->sum2, w->diff2, &w->pt2);
        } else {
            orc_vec_add(c, w, P, &w->pt2);
            orc_vec_duplicate(c, w, w->sum1, w->diff1, &w->pt1);
        }
        mask >>= 1;
    }
    *P = w->pt1;
}

/* primes in [lo, hi): plain sieve */
static uint64_t *orc_primes(uint64_t lo, uint64_t hi, size_t *count)
{
    uint8_t *comp = (uint8_t *)calloc((size_t)hi + 1, 1);
    size_t n = 0, cap = 1024;
    uint64_t *out = (uint64_t *)malloc(cap * sizeof(uint64_t));
    for (uint64_t i = 2; i * i < hi; i++)
        if (!comp[i])
            for (uint64_t j = i * i; j < hi; j += i) comp[j] = 1;
    for (uint64_t i = lo < 2 ? 2 : lo; i < hi; i++)
        if (!comp[i]) {
            if (n == cap) { cap *= 2; out = (uint64_t *)realloc(out, cap * sizeof(uint64_t)); }
            out[n++] = i;
        }
    free(comp);
    *count = n;
    return out;
}

/* ecm_stage1, ecm.c:1806-1854 */
static void orc_stage1(const orc_ctx *c, orc_work *w, orc_pt *P, uint64_t B1, const uint64_t *primes, size_t np)
{
    uint64_t q = 2;
    while (q < B1) {                                   /* ecm.c:1815-1822 */
        orc_submod(c, P->X, P->Z, w->diff1);
        orc_addmod(c, P->X, P->Z, w->sum1);
        orc_vec_duplicate(c, w, w->sum1, w->diff1, P);
        q *= 2;
    }
    for (size_t i = 1; i < np && primes[i] < B1; i++) {   /* ecm.c:1824-1832 */
        uint64_t cc = 1;
        q = primes[i];
        do {
            orc_prac(c, w, P, q);
            cc *= q;
        } while ((cc * q) < B1);
    }
}

/* build_one_curve, ecm.c:1548-1803 (Suyama branch :1711-1772) */
static void orc_build_curve(const orc_ctx *c, uint64_t sigma, orc_pt *P, uint64_t *s)
{
    mpz_t u, v, X, Z, A, t1, t2, t3, t4;
    mpz_inits(u, v, X, Z, A, t1, t2, t3, t4, NULL);
    mpz_set_ui(v, sigma);           /* one limb: sigma is 64-bit, unsigned long is 64-bit here */
    mpz_mul_2exp(v, v, 2);
    mpz_set_ui(u, sigma);
    mpz_mul(u, u, u);
    mpz_sub_ui(u, u, 5);
    mpz_mul(X, u, u); mpz_mul(X, X, u); mpz_tdiv_r(X, X, c->N);
    mpz_mul(Z, v, v); mpz_mul(Z, Z, v); mpz_tdiv_r(Z, Z, c->N);
    if (mpz_cmp(u, v) > 0) { mpz_sub(t1, v, u); mpz_add(t1, t1, c->N); }
    else mpz_sub(t1, v, u);
    mpz_mul(t2, t1, t1); mpz_tdiv_r(t2, t2, c->N);
    mpz_mul(t4, t2, t1); mpz_tdiv_r(t4, t4, c->N);
    mpz_mul_ui(t1, u, 3); mpz_add(t3, t1, v); mpz_tdiv_r(t3, t3, c->N);
    mpz_mul(t1, t3, t4); mpz_tdiv_r(t1, t1, c->N);
    mpz_mul_ui(t2, X, 16); mpz_mul(t4, t2, v); mpz_tdiv_r(t4, t4, c->N);
    mpz_invert(t2, t4, c->N);
    mpz_mul(A, t1, t2); mpz_tdiv_r(A, A, c->N);
    mpz_invert(t1, Z, c->N);
    mpz_mul(X, X, t1);
    mpz_set_ui(Z, 1);
    mp_bitcnt_t rb = (mp_bitcnt_t)(c->digitbits * c->nwords);
    mpz_mul_2exp(X, X, rb); mpz_tdiv_r(X, X, c->N);
    mpz_mul_2exp(Z, Z, rb); mpz_tdiv_r(Z, Z, c->N);
    mpz_mul_2exp(A, A, rb); mpz_tdiv_r(A, A, c->N);
    if (mpz_sgn(X) < 0) mpz_add(X, X, c->N);   /* only for sigma^2 > N, where the reference misbehaves */
    if (mpz_sgn(A) < 0) mpz_add(A, A, c->N);
    fe_from_mpz(c, P->X, X);
    fe_from_mpz(c, P->Z, Z);
    fe_from_mpz(c, s, A);
    mpz_clears(u, v, X, Z, A, t1, t2, t3, t4, NULL);
}

/* check_factor, ecm.c:2542-2557 */
static int orc_check_factor(const orc_ctx *c, const uint64_t *z, mpz_t f)
{
    mpz_t t;
    mpz_init(t);
    fe_to_mpz(c, t, z);
    mpz_gcd(f, t, c->N);
    mpz_clear(t);
    if (mpz_cmp_ui(f, 1) > 0) {
        if (mpz_cmp(f, c->N) == 0) { mpz_set_ui(f, 0); return 0; }
        return 1;
    }
    return 0;
}

static orc_work *work_new(void)
{
    orc_work *w = (orc_work *)calloc(1, sizeof *w);
    mpz_init(w->inv_factor);
    return w;
}

static void work_free(orc_work *w)
{
    free(w->map); free(w->Pa); free(w->Pb); free(w->Pa_inv); free(w->Paprod); free(w->Pbprod);
    mpz_clear(w->inv_factor);
    free(w);
}

int orc_stage1_line(orc_ctx *c, uint64_t sigma, uint64_t B1, char *line, size_t linelen, char *factor_dec,
                    size_t faclen, uint64_t *counts)
{
    orc_work *w = work_new();
    orc_pt P;
    size_t np;
    uint64_t *primes = orc_primes(0, B1 + 1000, &np);
    memset(&P, 0, sizeof P);
    orc_build_curve(c, sigma, &P, w->s);
    orc_stage1(c, w, &P, B1, primes, np);
    free(primes);
    if (counts) { counts[0] = w->ptadds; counts[1] = w->ptdups; }
    /* ecm.c:1327-1331: X*1, Z*1 */
    fe_t one, x, z;
    memset(one, 0, sizeof one);
    one[0] = 1;
    orc_mulmod(c, P.X, one, x);
    orc_mulmod(c, P.Z, one, z);
    mpz_t f, mx, mz;
    mpz_inits(f, mx, mz, NULL);
    int found = orc_check_factor(c, P.Z, f);
    if (factor_dec && faclen) {
        factor_dec[0] = 0;
        if (found) gmp_snprintf(factor_dec, faclen, "%Zd", f);
    }
    fe_to_mpz(c, mx, x);
    fe_to_mpz(c, mz, z);
    /* ecm.c:1372-1380 */
    int n = gmp_snprintf(line, linelen, "METHOD=ECM; SIGMA=%lu; B1=%lu; N=0x%Zx; X=0x%Zx; Z=0x%Zx; PROGRAM=AVX-ECM;\n",
                         (unsigned long)sigma, (unsigned long)B1, c->N, mx, mz);
    mpz_clears(f, mx, mz, NULL);
    work_free(w);
    return n;
}

/* vececm's loop over prime ranges for one sigma (ecm.c:1209-1234): ecm_stage1 is called once per range of
 * `prime_range` (PRIME_RANGE = 1e8 in the reference, main.c:581; smaller values let tests walk the same path
 * cheaply) with PRIMES = the primes of [rangemin, rangemax], rangemax = min(B2 + 1000, rangemin + prime_range)
 * (ecm.c:1215-1216).  Every call repeats the 2-power doublings and starts at PRIMES[1] (orc_stage1 above).
 * Stops after `stop_after` calls (0 = all of them).  The resume line carries b1_field, or — b1_field = 0 — the last
 * prime processed, PRIMES[last_pid - 1], as the checkpoint lines of ecm.c:1295-1305 do (the final save line: pass B1).
 * *checkpoint (if not NULL) = 1 when the reference would write checkpoint.txt after the last call made: it tests
 * PRIMES[last_pid] < B1 (ecm.c:1237) where last_pid is one past the list if the range ran out — a zero word of the
 * freshly mapped allocation. */
int orc_stage1_ranges_line(orc_ctx *c, uint64_t sigma, uint64_t B1, uint64_t B2, uint64_t prime_range, int stop_after,
                           uint64_t b1_field, char *line, size_t linelen, char *factor_dec, size_t faclen,
                           uint64_t *counts, int *checkpoint)
{
    orc_work *w = work_new();
    orc_pt P;
    memset(&P, 0, sizeof P);
    orc_build_curve(c, sigma, &P, w->s);
    if (B2 < B1) B2 = B1;
    uint64_t last_prime = 0;
    int calls = 0, ckpt = 0;
    for (uint64_t p = 0; p < B1; p += prime_range) {    /* ecm.c:1209 */
        const uint64_t rangemax = B2 + 1000 < p + prime_range ? B2 + 1000 : p + prime_range;
        size_t np;
        uint64_t *primes = orc_primes(p, rangemax + 1, &np);
        orc_stage1(c, w, &P, B1, primes, np);
        size_t i = 1;
        while (i < np && primes[i] < B1) i++;           /* work->last_pid, ecm.c:1844 */
        last_prime = np ? primes[i - 1] : 0;
        ckpt = i >= np;
        free(primes);
        if (++calls == stop_after) break;
    }
    if (counts) { counts[0] = w->ptadds; counts[1] = w->ptdups; counts[2] = last_prime; }
    if (checkpoint) *checkpoint = ckpt;
    fe_t one, x, z;
    memset(one, 0, sizeof one);
    one[0] = 1;
    orc_mulmod(c, P.X, one, x);
    orc_mulmod(c, P.Z, one, z);
    mpz_t f, mx, mz;
    mpz_inits(f, mx, mz, NULL);
    int found = orc_check_factor(c, P.Z, f);
    if (factor_dec && faclen) {
        factor_dec[0] = 0;
        if (found) gmp_snprintf(factor_dec, faclen, "%Zd", f);
    }
    fe_to_mpz(c, mx, x);
    fe_to_mpz(c, mz, z);
    int n = gmp_snprintf(line, linelen, "METHOD=ECM; SIGMA=%lu; B1=%lu; N=0x%Zx; X=0x%Zx; Z=0x%Zx; PROGRAM=AVX-ECM;\n",
                         (unsigned long)sigma, (unsigned long)(b1_field ? b1_field : last_prime), c->N, mx, mz);
    mpz_clears(f, mx, mz, NULL);
    work_free(w);
    return n;
}

double orc_time_stage1(orc_ctx *c, uint64_t sigma0, int curves, uint64_t B1)
{
    size_t np;
    uint64_t *primes = orc_primes(0, B1 + 1000, &np);
    struct timespec t0, t1;
    orc_work *w = work_new();
    volatile uint64_t sink = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int k = 0; k < curves; k++) {
        orc_pt P;
        memset(&P, 0, sizeof P);
        orc_build_curve(c, sigma0 + (uint64_t)k, &P, w->s);
        orc_stage1(c, w, &P, B1, primes, np);
        sink += P.X[0];
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    work_free(w);
    free(primes);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ------------------------------------------------------------------------------------------- stage 2 */
static uint32_t gcd32(uint32_t a, uint32_t b)
{
    while (b) { uint32_t t = a % b; a = b; b = t; }
    return a;
}

/* the parts of thread_init (main.c:834-882) and ecm_work_init (ecm.c:248-340) stage 2 needs */
static void work_init_stage2(orc_work *w, uint32_t D, uint32_t U)
{
    uint32_t i, j, m;
    w->D = D;
    w->U = U;
    w->L = 2 * U;                                     /* main.c:950 */
    for (j = 0, i = 0; i < 2 * D; i++)
        if (gcd32(i, 2 * D) == 1) j++;
    w->R = j + 3;                                     /* main.c:874-882 */
    w->Pa = (orc_pt *)calloc(2 * w->L, sizeof(orc_pt));
    w->Pa_inv = (fe_t *)calloc(2 * w->L, sizeof(fe_t));
    w->Paprod = (fe_t *)calloc(2 * w->L, sizeof(fe_t));
    w->Pb = (orc_pt *)calloc((size_t)U * (w->R + 1), sizeof(orc_pt));
    w->Pbprod = (fe_t *)calloc((size_t)U * (w->R + 1), sizeof(fe_t));
    w->map = (uint32_t *)calloc((size_t)U * (D + 1) + 3, sizeof(uint32_t));
    w->map[0] = 0; w->map[1] = 1; w->map[2] = 2;      /* ecm.c:301-329 */
    m = 3;
    for (i = 0; i < U; i++) {
        j = (i == 0) ? 3 : 1;
        for (; j < D; j++) {
            if (gcd32(j, D) == 1) w->map[i * D + j] = m++;
            else w->map[i * D + j] = 0;
        }
        if (i == 0) w->map[i * D + j] = m++;
    }
}

/* the shared core of batch_invert_pt_inplace (ecm.c:1869-2001) and batch_invert_pt_to_bignum
 * (ecm.c:2003-2136): given Z_0..Z_{num-1} returns Montgomery-form inverses in out[0..num-1]. */
static void batch_invert(const orc_ctx *c, orc_work *w, const uint64_t *const *Zs, fe_t *prefix, fe_t *out, int num)
{
    fe_t *B = (fe_t *)calloc((size_t)num, sizeof(fe_t));
    fe_t one;
    mpz_t g, inv;
    mpz_inits(g, inv, NULL);
    w->numinv++;
    fe_copy(c, prefix[0], Zs[0]);
    for (int i = 1; i < num; i++) orc_mulmod(c, Zs[i], prefix[i - 1], prefix[i]);
    memset(one, 0, sizeof one);
    one[0] = 1;
    orc_mulmod(c, prefix[num - 1], one, B[num - 1]);          /* out of Montgomery form, :1903-1912 */
    fe_to_mpz(c, g, B[num - 1]);
    if (mpz_invert(inv, g, c->N) == 0) {                      /* :1925-1939 */
        /* The reference stores gcd(product, N) into stg2acc — every time an inversion fails, so the LAST failing
         * batch is the one whose gcd the accumulator carries to the end (each later cross product only multiplies
         * it) — and goes on with whatever its destination variable held: GMP documents rop as undefined when
         * mpz_invert fails; in practice it is the previous lane's inverse (lane-order-dependent garbage).  The
         * garbage is not restated: the gcd of the last failing batch is recorded and reported as the factor, the
         * inverse is taken as 0.  The reference's result differs from this only if one of the garbage cross
         * products happens to vanish modulo another prime factor of N. */
        mpz_gcd(w->inv_factor, g, c->N);
        w->found_during_inv = 1;
        mpz_set_ui(inv, 0);
    }
    mpz_mul_2exp(inv, inv, (mp_bitcnt_t)c->maxbits);          /* :1944-1945 */
    mpz_tdiv_r(inv, inv, c->N);
    fe_from_mpz(c, B[num - 1], inv);
    for (int i = num - 2; i >= 0; i--) orc_mulmod(c, Zs[i + 1], B[i + 1], B[i]);   /* :1965-1968 */
    fe_copy(c, out[0], B[0]);
    for (int i = 1; i < num; i++) orc_mulmod(c, B[i], prefix[i - 1], out[i]);     /* :1974-1978 */
    mpz_clears(g, inv, NULL);
    free(B);
}

/* ecm_stage2_init, ecm.c:2201-2340 */
static void orc_stage2_init(const orc_ctx *c, orc_work *w, const orc_pt *P, uint64_t B1)
{
    uint32_t wD = w->D, U = w->U;
    orc_pt *Pb = w->Pb;
    int lastMapID = 0;
    w->amin = (uint32_t)((B1 + wD) / (2 * wD));
    w->paired = 0; w->ptadds = 0; w->ptdups = 0; w->numinv = 0;
    Pb[1] = *P;
    Pb[2] = *P;
    orc_addsubmod(c, P->X, P->Z, w->sum1, w->diff1);
    orc_vec_duplicate(c, w, w->sum1, w->diff1, &Pb[2]);
    w->pt2 = Pb[1];
    w->pt1 = Pb[2];
    for (uint32_t j = 3; j <= U * wD; j++) {
        orc_pt *P1 = &w->pt1, *P2 = &Pb[1], *P3 = &w->pt2, *Pout = &Pb[w->map[j]];
        if (w->map[j] > 0) lastMapID = (int)w->map[j];
        orc_addsubmod(c, P1->X, P1->Z, w->sum1, w->diff1);
        orc_addsubmod(c, P2->X, P2->Z, w->sum2, w->diff2);
        orc_mulmod(c, w->diff1, w->sum2, w->tt1);
        orc_mulmod(c, w->sum1, w->diff2, w->tt2);
        orc_addsubmod(c, w->tt1, w->tt2, Pout->X, Pout->Z);
        orc_sqrmod(c, Pout->X, w->tt1);
        orc_sqrmod(c, Pout->Z, w->tt2);
        orc_mulmod(c, w->tt1, P3->Z, Pout->X);
        orc_mulmod(c, w->tt2, P3->X, Pout->Z);
        w->ptadds++;
        *P3 = *P1;
        *P1 = *Pout;
    }
    fe_copy(c, w->stg2acc, c->one);                         /* ecm.c:2318 */
    {   /* batch_invert_pt_inplace(Pb, Pbprod, ..., lastMapID + 1), indices 1..num-1 */
        int num = lastMapID + 1;
        const uint64_t **Zs = (const uint64_t **)malloc(sizeof(uint64_t *) * (size_t)num);
        fe_t *inv = (fe_t *)calloc((size_t)num, sizeof(fe_t));
        for (int i = 1; i < num; i++) Zs[i - 1] = Pb[i].Z;
        batch_invert(c, w, Zs, w->Pbprod, inv, num - 1);
        for (int i = 1; i < num; i++) {
            fe_copy(c, Pb[i].Z, inv[i - 1]);
            orc_mulmod(c, Pb[i].X, Pb[i].Z, Pb[i].X);       /* :1983-1987 */
        }
        free(inv);
        free(Zs);
    }
    w->Pdnorm = *P;
    orc_next_pt(c, w, &w->Pdnorm, wD);                      /* Pd = [w]Q, ecm.c:2332-2334 */
}

static void invert_Pa_range(const orc_ctx *c, orc_work *w, int start, int stop)
{
    int num = stop - start;
    const uint64_t **Zs = (const uint64_t **)malloc(sizeof(uint64_t *) * (size_t)num);
    fe_t *inv = (fe_t *)calloc((size_t)num, sizeof(fe_t));
    for (int i = 0; i < num; i++) Zs[i] = w->Pa[start + i].Z;
    batch_invert(c, w, Zs, w->Paprod, inv, num);
    for (int i = 0; i < num; i++) orc_mulmod(c, w->Pa[start + i].X, inv[i], w->Pa_inv[start + i]);  /* :2115-2119 */
    free(inv);
    free(Zs);
}

/* ecm_stage2_pair, ecm.c:2342-2540 */
static void orc_stage2_pair(const orc_ctx *c, orc_work *w, const orc_pt *P, uint32_t steps, const uint32_t *pm_v,
                            const uint32_t *pm_u)
{
    uint32_t wD = w->D, U = w->U, L = w->L;
    uint32_t amin = w->amin;
    orc_pt *Pa = w->Pa, *Pd = &w->Pdnorm;
    w->A = (uint64_t)amin * (uint64_t)wD * 2;
    Pa[0] = *P;
    orc_next_pt(c, w, &Pa[0], w->A);
    w->Pad = *P;
    orc_next_pt(c, w, &w->Pad, w->A - wD);
    orc_addmod(c, Pa[0].X, Pa[0].Z, w->sum1);
    orc_addmod(c, Pd->X, Pd->Z, w->sum2);
    orc_submod(c, Pa[0].X, Pa[0].Z, w->diff1);
    orc_submod(c, Pd->X, Pd->Z, w->diff2);
    orc_vec_add(c, w, &w->Pad, &Pa[1]);
    w->A += wD;
    for (uint32_t i = 2; i < 2 * L; i++) {
        orc_addsubmod(c, Pa[i - 1].X, Pa[i - 1].Z, w->sum1, w->diff1);
        orc_addsubmod(c, Pd->X, Pd->Z, w->sum2, w->diff2);
        orc_vec_add(c, w, &Pa[i - 2], &Pa[i]);
        w->A += wD;
    }
    invert_Pa_range(c, w, 0, (int)(2 * L));
    w->numinv++;                                            /* counted twice, ecm.c:2428-2429 */
    for (uint32_t mapid = 0; mapid < steps; mapid++) {
        if (pm_u[mapid] == 0 && pm_v[mapid] == 0) {         /* ecm.c:2458-2502 */
            uint32_t shift = 2 * U;
            for (uint32_t i = 0; i < 2 * L - shift; i++) {
                Pa[i] = Pa[i + shift];
                fe_copy(c, w->Pa_inv[i], w->Pa_inv[i + shift]);
            }
            for (uint32_t i = 2 * L - shift; i < 2 * L; i++) {
                orc_addsubmod(c, Pa[i - 1].X, Pa[i - 1].Z, w->sum1, w->diff1);
                orc_addsubmod(c, Pd->X, Pd->Z, w->sum2, w->diff2);
                orc_vec_add(c, w, &Pa[i - 2], &Pa[i]);
                w->A += wD;
            }
            amin += U;
            invert_Pa_range(c, w, (int)(2 * L - shift), (int)(2 * L));
        } else {                                            /* CROSS_PRODUCT_INV, ecm.c:1857-1859 */
            uint32_t pa = pm_v[mapid] - amin, pb = pm_u[mapid];
            orc_submod(c, w->Pa_inv[pa], w->Pb[w->map[pb]].X, w->tt1);
            orc_mulmod(c, w->stg2acc, w->tt1, w->stg2acc);
            w->paired++;
        }
    }
    w->amin = amin;
}

/* ---- PAIR, ecm.c:2559-2910, with the queues of queue.c:32-102 -------------------------------- */
typedef struct {
    uint32_t *Q;
    uint32_t len, sz, head, tail;
} orc_queue;

static void q_enqueue(orc_queue *q, uint32_t e)
{
    q->Q[q->tail++] = e;
    q->len++;
    if (q->tail == q->sz) q->tail = 0;
}

static uint32_t q_dequeue(orc_queue *q)
{
    uint32_t e = q->Q[q->head++];
    q->len--;
    if (q->head == q->sz) q->head = 0;
    return e;
}

uint32_t orc_pair(uint64_t B1, uint64_t B2, uint32_t D, uint32_t U, uint32_t **pm_v_out, uint32_t **pm_u_out,
                  uint32_t *amin_out, uint32_t *pairs_out, uint32_t *nump_out)
{
    int64_t w = D;
    uint32_t L = 2 * U;
    int64_t umax = (int64_t)D * U;
    /* Qmap / Qrmap, main.c:723-748 */
    uint32_t *Qmap = (uint32_t *)malloc(2 * D * sizeof(uint32_t));
    uint32_t *Qrmap = (uint32_t *)malloc(2 * D * sizeof(uint32_t));
    uint32_t R = 0;
    for (uint32_t k = 0; k < 2 * D; k++) {
        if (gcd32(k, 2 * D) == 1) { Qmap[k] = R; Qrmap[R++] = k; }
        else Qmap[k] = (uint32_t)-1;
    }
    orc_queue *Q = (orc_queue *)calloc(R, sizeof(orc_queue));
    for (uint32_t k = 0; k < R; k++) { Q[k].Q = (uint32_t *)malloc(D * sizeof(uint32_t)); Q[k].sz = D; }
    size_t np;
    uint64_t *primes = orc_primes(B1 > 1000 ? B1 - 1000 : 0, B2 + 1000, &np);
    size_t cap = np + np / 4 + 1024, mapid = 0;
    uint32_t *pm_v = (uint32_t *)malloc(cap * sizeof(uint32_t)), *pm_u = (uint32_t *)malloc(cap * sizeof(uint32_t));
    uint64_t amin = (B1 + (uint64_t)w) / (2 * (uint64_t)w);
    uint32_t pairs = 0, nump = 0;
    size_t pid = 0;
    *amin_out = (uint32_t)amin;
#define EMIT(v, u)                                                                                   \
    do {                                                                                             \
        if (mapid == cap) { cap *= 2; pm_v = (uint32_t *)realloc(pm_v, cap * 4); pm_u = (uint32_t *)realloc(pm_u, cap * 4); } \
        pm_v[mapid] = (uint32_t)(v); pm_u[mapid] = (uint32_t)(u); mapid++;                           \
    } while (0)
    while (pid < np && primes[pid] < B1) pid++;
    while (pid < np && primes[pid] < B2) {
        uint64_t s = primes[pid], a = (s + (uint64_t)w) / (2 * (uint64_t)w), ap;
        int64_t q, mq, u;
        nump++;
        while (a >= amin + L) {                             /* ecm.c:2611-2685 */
            uint64_t oldmin = amin;
            amin = amin + L - U;
            for (uint32_t i = 0; i < R; i++) {
                uint32_t len = Q[i].len;
                int64_t qq = (Qrmap[i] > (uint32_t)w) ? 2 * w - (int64_t)Qrmap[i] : (int64_t)Qrmap[i];
                for (uint32_t j = 0; j < len; j++) {
                    ap = q_dequeue(&Q[i]);
                    if ((uint32_t)ap < amin) { EMIT(2 * ap - oldmin, qq); pairs++; }
                    else q_enqueue(&Q[i], (uint32_t)ap);
                }
            }
            EMIT(0, 0);
        }
        q = (int64_t)s - 2 * (int64_t)a * w;                /* ecm.c:2687-2691 */
        if (q < 0) mq = -q; else mq = 2 * w - q;
        do {
            if (Q[Qmap[mq]].len > 0) {
                ap = q_dequeue(&Q[Qmap[mq]]);
                if (q < 0) u = w * (int64_t)(a - ap) + q; else u = w * (int64_t)(a - ap) + q;
                if (u > umax) {
                    int64_t qq = q < 0 ? -q : q;
                    if (q >= 0 && qq >= w) qq = 2 * w - qq;
                    EMIT(2 * ap - amin, qq);
                    pairs++;
                } else {
                    EMIT(a + ap - amin, u);
                    pairs++;
                }
            } else {
                if (q < 0) q_enqueue(&Q[Qmap[2 * w + q]], (uint32_t)a);
                else q_enqueue(&Q[Qmap[q]], (uint32_t)a);
                u = 0;
            }
        } while (u > umax);
        pid++;
    }
    for (uint32_t i = 0; i < R; i++) {                      /* ecm.c:2796-2843 */
        uint32_t len = Q[i].len;
        int64_t qq = (Qrmap[i] > (uint32_t)w) ? 2 * w - (int64_t)Qrmap[i] : (int64_t)Qrmap[i];
        for (uint32_t j = 0; j < len; j++) {
            uint64_t ap = q_dequeue(&Q[i]);
            EMIT(2 * ap - amin, qq);
            pairs++;
        }
    }
#undef EMIT
    for (uint32_t k = 0; k < R; k++) free(Q[k].Q);
    free(Q); free(Qmap); free(Qrmap); free(primes);
    *pm_v_out = pm_v;
    *pm_u_out = pm_u;
    if (pairs_out) *pairs_out = pairs;
    if (nump_out) *nump_out = nump;
    return (uint32_t)mapid;
}

int orc_stage2(orc_ctx *c, uint64_t sigma, uint64_t B1, uint64_t B2, uint32_t D, uint32_t U, char *acc_hex,
               char *factor_dec, size_t faclen, uint64_t *counts)
{
    const uint64_t PRIME_RANGE = 100000000ULL;              /* avx_ecm.h / main.c:581 */
    orc_work *w = work_new();
    orc_pt P;
    size_t np;
    uint64_t *primes = orc_primes(0, B1 + 1000, &np);
    memset(&P, 0, sizeof P);
    orc_build_curve(c, sigma, &P, w->s);
    orc_stage1(c, w, &P, B1, primes, np);
    free(primes);
    work_init_stage2(w, D, U);
    orc_stage2_init(c, w, &P, B1);
    for (uint64_t p = B1; p < B2; p += PRIME_RANGE) {       /* ecm.c:1424-1476 */
        uint32_t *pm_v, *pm_u, amin, steps;
        uint64_t hi = p + PRIME_RANGE < B2 ? p + PRIME_RANGE : B2;
        steps = orc_pair(p, hi, D, U, &pm_v, &pm_u, &amin, NULL, NULL);
        w->amin = amin;
        orc_stage2_pair(c, w, &P, steps, pm_v, pm_u);
        free(pm_v);
        free(pm_u);
    }
    if (counts) { counts[0] = w->ptadds; counts[1] = w->numinv; counts[2] = w->paired; }
    mpz_t f, t;
    mpz_inits(f, t, NULL);
    int found = orc_check_factor(c, w->stg2acc, f);         /* ecm.c:1489-1490 */
    if (w->found_during_inv) {
        mpz_set(f, w->inv_factor);
        found = mpz_cmp_ui(f, 1) > 0 && mpz_cmp(f, c->N) != 0;
    }
    if (factor_dec && faclen) {
        factor_dec[0] = 0;
        if (found) gmp_snprintf(factor_dec, faclen, "%Zd", f);
    }
    if (acc_hex) {
        fe_to_mpz(c, t, w->stg2acc);
        mpz_get_str(acc_hex, 16, t);
    }
    mpz_clears(f, t, NULL);
    work_free(w);
    return found;
}
