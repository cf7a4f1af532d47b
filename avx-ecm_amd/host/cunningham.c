/* cunningham.c — what the reference's main() does to its input before any curve is built
 * (main.c:403-457): recognise N | 2^k - 1, N | 2^k + 1 or 2^k = c (mod N) with a one-limb c, and for the
 * first two strip the algebraic factors by intersecting N with the "primitive" part of 2^k -/+ 1.
 *
 * The reference then either multiplies modulo 2^k -/+ 1 with special folding routines
 * (vecarith52.c:284-2436) or — when the cofactor is under 0.7 of k in limbs — stays with REDC
 * (main.c:505-527).  libgecm always uses REDC on the cofactor: every residue it produces is
 * the reference's residue reduced modulo N.  What must match exactly is the N the run is made on and the
 * lines printed about it; that is this file.
 */
#include "cunningham.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

void cunningham_detect(cunningham_form *f, const mpl_t *N, int digitbits)
{
    f->form = 0;
    f->k = 0;
    f->c = 0;
    const int size_n = mpl_bits(N);
    mpl_t t, one, nm1;
    mpl_set_u64(&one, 1);
    mpl_sub(&nm1, N, &one);
    /* t = 2^i mod N, starting at i = size_n - 1 (main.c:408) */
    mpl_shl(&t, &one, (unsigned)(size_n - 1));
    mpl_mod(&t, &t, N);
    for (int i = size_n - 1; i < 2048; i++) {
        if (mpl_cmp(&t, &one) == 0) {              /* 2^i - 1 = 0 (mod N)   main.c:410-419 */
            f->form = 1;
            f->k = i;
            return;
        }
        if (mpl_cmp(&t, &nm1) == 0) {              /* 2^i + 1 = 0 (mod N)   main.c:421-430 */
            f->form = -1;
            f->k = i;
            return;
        }
        if ((mpl_is_zero(&t) ? 1 : mpl_bits(&t)) < digitbits) {   /* main.c:432-441 */
            f->form = 2;
            f->k = i;
            f->c = mpl_get_u64(&t);
            return;
        }
        mpl_add(&t, &t, &t);
        if (mpl_cmp(&t, N) >= 0) mpl_sub(&t, &t, N);
    }
}

static void logf_(char *log, size_t loglen, const char *fmt, ...)
{
    size_t used = strlen(log);
    if (used + 1 >= loglen) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(log + used, loglen - used, fmt, ap);
    va_end(ap);
}

/* 2^x -/+ 1 */
static int two_pow_pm1(mpl_t *r, int x, int sign)
{
    if (x + 2 > MPL_MAXL * 32) return -1;
    mpl_t one;
    mpl_set_u64(&one, 1);
    mpl_shl(r, &one, (unsigned)x);
    if (sign > 0) mpl_sub(r, r, &one);
    else mpl_add(r, r, &one);
    return 0;
}

/* The primitive part of 2^e -/+ 1 (find_primitive_factor, main.c:187-353) by inclusion-exclusion over the distinct odd
 * primes p_0 < ... < p_(r-1) of e:
 *
 *      prod over subsets S of { 2^(m * prod_{i in S} p_i) -/+ 1 }^((-1)^(r - |S|)),      m = e / (p_0 ... p_(r-1)).
 *
 * One enumeration does every r: subsets are listed by size, and within a size in lexicographic order of their index
 * tuples.  Sizes of the parity of r multiply, largest first; then the other sizes divide, largest first.  That is also the
 * order of the reference's "gen:" lines, which are reproduced here (the reference names the subsets of size k "rank k"
 * and gives up beyond three distinct odd primes; so does this). */
#define CUN_MAXP 3

typedef struct {
    int prod[1 << CUN_MAXP];      /* product of the chosen primes, subsets of one size after the other */
    int first[CUN_MAXP + 2];      /* subsets of size k: prod[first[k]] .. prod[first[k+1]-1] */
} cun_subsets;

/* append the size-`want` subsets of p[from..r) that extend a partial product */
static void subsets_of_size(cun_subsets *s, int *count, const int *p, int r, int from, int want, int partial)
{
    if (want == 0) {
        s->prod[(*count)++] = partial;
        return;
    }
    for (int i = from; i + want <= r; i++) subsets_of_size(s, count, p, r, i + 1, want - 1, partial * p[i]);
}

static int distinct_odd_primes_below_1000(int *p, int cap, int e)
{
    /* tdiv_int (main.c:163-184) walks the primes below 1000 in ascending order; only which odd ones divide e matters */
    int r = 0;
    for (int q = 3; q < 1000 && e > 1; q += 2) {
        int prime = 1;
        for (int d = 3; d * d <= q; d += 2)
            if (q % d == 0) { prime = 0; break; }
        if (!prime || e % q) continue;
        while (e % q == 0) e /= q;
        if (r < cap) p[r] = q;
        r++;
    }
    return r;
}

int cunningham_primitive(mpl_t *prim, int e, int sign, char *log, size_t loglen)
{
    int p[32];
    const int r = distinct_odd_primes_below_1000(p, 32, e);
    logf_(log, loglen, "gen: rank 1 terms: ");
    for (int i = 0; i < r && i < 32; i++) logf_(log, loglen, "%d ", p[i]);
    logf_(log, loglen, "\n");
    if (r > CUN_MAXP) {
        logf_(log, loglen, "gen: too many distinct odd factors in exponent!\n");
        return -1;
    }
    cun_subsets s;
    int count = 0, m = e;
    for (int k = 0; k <= r; k++) {
        s.first[k] = count;
        subsets_of_size(&s, &count, p, r, 0, k, 1);
    }
    s.first[r + 1] = count;
    for (int k = 2; k <= r; k++) {
        const int n = s.first[k + 1] - s.first[k];
        if (n == 1) {
            logf_(log, loglen, "gen: rank %d term: %d\n", k, s.prod[s.first[k]]);
            continue;
        }
        logf_(log, loglen, "gen: rank %d terms: ", k);
        for (int t = s.first[k]; t < s.first[k + 1]; t++) logf_(log, loglen, "%d ", s.prod[t]);
        logf_(log, loglen, "\n");
    }
    for (int i = 0; i < r; i++) m /= p[i];
    logf_(log, loglen, "gen: base exponent multiplier: %d\n", m);

    static char dec[MPL_MAXL * 10 + 16];
    mpl_t acc, term, quo, rem;
    mpl_set_u64(&acc, 1);
    for (int pass = 0; pass < 2; pass++)                     /* 0: the factors of the numerator, 1: of the denominator */
        for (int k = r; k >= 0; k--) {
            if (((r - k) & 1) != pass) continue;
            for (int t = s.first[k]; t < s.first[k + 1]; t++) {
                const int x = s.prod[t] * m;
                if (two_pow_pm1(&term, x, sign)) return -1;
                mpl_get_dec(dec, &term);
                logf_(log, loglen, "gen: %s by 2^%d %c 1 = %s\n", pass ? "dividing" : "multiplying", x, sign > 0 ? '-' : '+', dec);
                if (!pass) {
                    if (acc.n + term.n > MPL_MAXL - 2) return -1;
                    mpl_mul(&acc, &acc, &term);
                    continue;
                }
                mpl_divrem(&quo, &rem, &acc, &term);
                if (!mpl_is_zero(&rem)) logf_(log, loglen, "gen: error, term doesn't divide n!\n");
                else acc = quo;
            }
        }
    *prim = acc;
    return 0;
}

/* ---- the public entry point (include/gecm.h) ------------------------------------------------- */
#include "../../include/gecm.h"
#include "calc_lite.h"

static int words_for(int bits, int digitbits)
{
    /* main.c:464-483: MAXBITS = 208 (128 for 32-bit limbs), grown in steps of itself while <= bits */
    const int step = digitbits == 52 ? 208 : 128;
    int maxbits = step;
    while (maxbits <= bits) maxbits += step;
    return maxbits / digitbits;
}

int gecm_prepare_input(const char *expr, int digitbits, char *n_dec, size_t n_len, gecm_input_info *info,
                       char *log, size_t loglen)
{
    static char dec[MPL_MAXL * 10 + 16];
    if (!expr || !n_dec || !log || loglen == 0 || (digitbits != 52 && digitbits != 32)) return GECM_ERR_ARG;
    log[0] = 0;
    mpl_t N;
    if (calc_lite(&N, expr) || mpl_cmp_u64(&N, 3) < 0 || !mpl_is_odd(&N)) return GECM_ERR_ARG;
    cunningham_form f;
    cunningham_detect(&f, &N, digitbits);
    int size_n = f.form ? f.k : mpl_bits(&N);
    int is_m = f.form == 2 ? (int)(uint32_t)f.c : f.form;       /* `int isMersenne = mpz_get_ui(g)`, main.c:439 */
    if (f.form == 1 || f.form == -1) {                          /* main.c:445-457 */
        mpl_t g, r, rem;
        if (cunningham_primitive(&g, size_n, f.form, log, loglen)) return GECM_ERR_ARG;
        mpl_divrem(&r, &rem, &N, &g);
        mpl_get_dec(dec, &r);
        logf_(log, loglen, "removing algebraic %s%d factor %s\n", mpl_probab_prime(&g, 3) ? "PRP" : "C",
              mpl_sizeinbase10(&r), dec);
        mpl_gcd(&N, &N, &g);
        if (mpl_cmp_u64(&N, 3) < 0) return GECM_ERR_ARG;        /* nothing left to factor */
    }
    const int nwords = words_for(mpl_bits(&N), digitbits);
    const int mwords = words_for(size_n, digitbits);
    mpl_get_dec(dec, &N);
    logf_(log, loglen, "commencing parallel ecm on %s\n", dec);                  /* main.c:503 */
    int special = 0;
    if (is_m && (double)nwords / (double)mwords < 0.7) {                          /* main.c:505-516 */
        logf_(log, loglen, "Mersenne input 2^%d %c %d determined to be faster by REDC\n", size_n,
              is_m > 0 ? '-' : '+', is_m);
    } else if (is_m) {
        special = 1;
    }
    if (strlen(dec) + 1 > n_len) return GECM_ERR_ARG;
    strcpy(n_dec, dec);
    if (info) {
        info->form = is_m;
        info->k = f.form ? f.k : 0;
        info->c = f.c;
        info->nbits = mpl_bits(&N);
        info->ref_special_reduction = special;
    }
    return GECM_OK;
}

/* the digit count the reference prints next to a factor: mpz_sizeinbase(f, 10) (ecm.c:1346, 1494) */
int gecm_sizeinbase10(const char *dec)
{
    mpl_t v;
    if (!dec || mpl_set_str(&v, dec)) return GECM_ERR_ARG;
    return mpl_sizeinbase10(&v);
}

/* the hash of the host sources this object was compiled from (Makefile: H_SHA); gecm_version() compares them */
#ifdef GECM_MANIFEST_FN
const char *GECM_MANIFEST_FN(void) { return GECM_MANIFEST; }
#endif
