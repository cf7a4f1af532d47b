/* gecm_plan.c — prime supply and stage-1 tape compiler (host, plain C).
 *
 * The reference runs prac() inside the per-thread hot loop and re-derives the same Lucas chain
 * for every batch of 8 curves (ecm.c:1824-1832).  The chain depends only on B1, so here it is
 * evaluated ONCE and recorded as a byte tape that every lane of every GPU replays.
 *
 * Bit-exactness traps reproduced on purpose:
 *   - the multiplier choice uses IEEE double arithmetic, r = (uint64_t)((double)d * v + 0.5)
 *     (ecm.c:486, 584); this file must be compiled with -ffp-contract=off so no FMA is formed;
 *   - ties keep the lowest index (strict <, ecm.c:577), initial cmin = ADD*c (ecm.c:574);
 *   - prime powers use strict <  (ecm.c:1816, 1832).
 */
#include "gecm_plan.h"
#include "../csrc/gecm_tape.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC optimize("fp-contract=off")
#endif
#pragma STDC FP_CONTRACT OFF

/* ------------------------------------------------------------------ primes */
uint64_t *gecm_primes_range(uint64_t lo, uint64_t hi, size_t *count)
{
    *count = 0;
    if (hi < 2 || hi <= lo) return (uint64_t *)calloc(1, sizeof(uint64_t));
    /* base primes up to sqrt(hi) */
    uint64_t root = 1;
    while ((root + 1) * (root + 1) < hi) root++;
    root += 1;
    uint8_t *small = (uint8_t *)calloc(root + 1, 1);
    if (!small) return NULL;
    size_t nbase = 0;
    uint32_t *base = (uint32_t *)malloc((root / 2 + 16) * sizeof(uint32_t));
    if (!base) { free(small); return NULL; }
    for (uint64_t i = 2; i <= root; i++) {
        if (!small[i]) {
            base[nbase++] = (uint32_t)i;
            for (uint64_t j = i * i; j <= root; j += i) small[j] = 1;
        }
    }
    free(small);
    /* upper bound on pi(hi)-pi(lo): generous */
    size_t cap = (size_t)((hi - lo) / 2 + 1024);
    if (hi > 100000) {
        /* pi(x) < 1.26 x / ln x; ln via integer log2 */
        double lg = 0; uint64_t t = hi; while (t > 1) { lg += 1; t >>= 1; }
        size_t est = (size_t)(1.3 * (double)hi / (lg * 0.6931)) + 1024;
        if (est < cap) cap = est;
    }
    uint64_t *out = (uint64_t *)malloc(cap * sizeof(uint64_t));
    if (!out) { free(base); return NULL; }
    size_t n = 0;
    const uint64_t SEG = 1u << 18;
    uint8_t *seg = (uint8_t *)malloc(SEG);
    if (!seg) { free(base); free(out); return NULL; }
    for (uint64_t s = lo; s < hi; s += SEG) {
        uint64_t e = s + SEG < hi ? s + SEG : hi;
        memset(seg, 0, (size_t)(e - s));
        for (size_t k = 0; k < nbase; k++) {
            uint64_t p = base[k];
            if (p * p >= e) break;
            uint64_t start = (s + p - 1) / p * p;
            if (start < p * p) start = p * p;
            for (uint64_t j = start; j < e; j += p) seg[j - s] = 1;
        }
        for (uint64_t v = s; v < e; v++) {
            if (v >= 2 && !seg[v - s]) {
                if (n == cap) {
                    cap = cap * 2;
                    uint64_t *t2 = (uint64_t *)realloc(out, cap * sizeof(uint64_t));
                    if (!t2) { free(out); free(base); free(seg); return NULL; }
                    out = t2;
                }
                out[n++] = v;
            }
        }
    }
    free(seg);
    free(base);
    *count = n;
    return out;
}

/* ------------------------------------------------------------------ PRAC */
#define ADD 5.5 /* ecm.c:459 */
#define DUP 4.5 /* ecm.c:460 */
#define NV 10
/* ecm.c:473-477 */
static const double val[NV] = {0.61803398874989485, 0.72360679774997897, 0.58017872829546410,
                               0.63283980608870629, 0.61242994950949500, 0.62018198080741576,
                               0.61721461653440386, 0.61834711965622806, 0.61791440652881789,
                               0.61807966846989581};

double gecm_lucas_cost(uint64_t n, double v)
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
        if ((d + 3) / 4 <= e) { d -= e; c += ADD; }                        /* rule 3 */
        else if ((d + e) % 2 == 0) { d = (d - e) / 2; c += ADD + DUP; }    /* rule 4 */
        else if (d % 2 == 0) { d /= 2; c += ADD + DUP; }                   /* rule 5 */
        else { e /= 2; c += ADD + DUP; }                                   /* rule 9 */
    }
    if (d != 1) return 999999999.;
    return c;
}

int gecm_prac_best_multiplier(uint64_t c)
{
    int i = 0;
    double cmin = ADD * (double)c;
    for (int d = 0; d < NV; d++) {
        double cost = gecm_lucas_cost(c, val[d]);
        if (cost < cmin) { cmin = cost; i = d; }
    }
    return i;
}

static int tape_push(gecm_tape_t *t, uint8_t op)
{
    if ((t->len & 0xffff) == 0) {
        uint8_t *p = (uint8_t *)realloc(t->ops, t->len + 0x10000 + 8);
        if (!p) return -1;
        t->ops = p;
    }
    t->ops[t->len++] = op;
    return 0;
}

int gecm_tape_append_prac(gecm_tape_t *t, uint64_t c)
{
    uint64_t d, e, r;
    int i = gecm_prac_best_multiplier(c);
    d = c;
    r = (uint64_t)((double)d * val[i] + 0.5);     /* ecm.c:584 */
    d = c - r;                                      /* ecm.c:592 */
    e = 2 * r - c;
    if (tape_push(t, GECM_OP_PRAC_BEGIN)) return -1;    /* ecm.c:603-613 */
    t->ptdups++;
    t->prac_calls++;
    while (d != e) {
        uint8_t op = GECM_OP_STEP;
        if (d < e) { r = d; d = e; e = r; op |= GECM_OP_SWAP; t->swaps++; }
        if ((d + 3) / 4 <= e) { d -= e; op |= GECM_OP_RULE3; t->ptadds++; t->rule_count[0]++; }
        else if ((d + e) % 2 == 0) { d = (d - e) / 2; op |= GECM_OP_RULE4; t->ptadds++; t->ptdups++; t->rule_count[1]++; }
        else if (d % 2 == 0) { d /= 2; op |= GECM_OP_RULE5; t->ptadds++; t->ptdups++; t->rule_count[2]++; }
        else { e /= 2; op |= GECM_OP_RULE9; t->ptadds++; t->ptdups++; t->rule_count[3]++; }
        if (tape_push(t, op)) return -1;
    }
    if (tape_push(t, GECM_OP_PRAC_END)) return -1;      /* ecm.c:868-873 */
    t->ptadds++;
    return d == 1 ? 0 : -2;                              /* ecm.c:877-880 */
}

/* ---- one prime range of stage 1 -------------------------------------------------------------
 * vececm calls ecm_stage1 once per range of PRIME_RANGE = 1e8 (ecm.c:1209-1234) with PRIMES = the primes of
 * [rangemin, rangemax]; every call (ecm.c:1806-1854)
 *   - runs the 2-power doublings again: one per power of two below B1 (ecm.c:1815-1822), in EVERY range;
 *   - starts at PRIMES[1]: the range's first prime is never processed (2 in the first range, where the doublings
 *     stand for it; the first prime above a multiple of 1e8 in the later ones);
 *   - processes prac(q) for the primes q < B1 of the range, repeated while q^k < B1.
 * The first range alone is the whole of stage 1 for B1 <= 1e8.  The chains of different primes are independent, so
 * the range's primes are cut into slices compiled by worker threads and the pieces joined in order. */
static uint64_t g_prime_range = GECM_PRIME_RANGE;
/* test hook: walk the multi-range path with short ranges (tests compare with the oracle run the same way) */
void gecm_plan_set_prime_range_for_tests(uint64_t range) { g_prime_range = range ? range : GECM_PRIME_RANGE; }
/* the same hook for a whole process (the command-line driver under test): GECM_TEST_PRIME_RANGE=n, read once */
__attribute__((constructor)) static void prime_range_from_env(void)
{
    const char *e = getenv("GECM_TEST_PRIME_RANGE");
    if (e && atoll(e) >= 16) g_prime_range = (uint64_t)atoll(e);
}

typedef struct {
    const uint64_t *primes;
    size_t lo, hi;
    uint64_t B1;
    gecm_tape_t t;
    int rc;
} tape_slice;

static void *tape_slice_run(void *arg)
{
    tape_slice *s = (tape_slice *)arg;
    memset(&s->t, 0, sizeof s->t);
    s->rc = 0;
    for (size_t k = s->lo; k < s->hi && !s->rc; k++) {  /* ecm.c:1824-1832 */
        const uint64_t q = s->primes[k];
        uint64_t c = 1;
        do {
            s->rc = gecm_tape_append_prac(&s->t, q);
            c *= q;
        } while (!s->rc && (c * q) < s->B1);
    }
    return NULL;
}

uint32_t gecm_stage1_ranges_plan(uint64_t B1)
{
    return B1 <= g_prime_range ? 1u : (uint32_t)((B1 + g_prime_range - 1) / g_prime_range);   /* ecm.c:1209 */
}

int gecm_tape_build_stage1_range(gecm_tape_t *t, uint64_t B1, uint32_t range, int threads)
{
    memset(t, 0, sizeof *t);
    if (range >= gecm_stage1_ranges_plan(B1)) return -2;
    /* PRIMES = the primes of [rangemin, rangemax], both ends included (GetPRIMESRange; a prime AT a range boundary —
     * impossible with the reference's 1e8, possible with the short ranges tests use — ends one list and heads the next,
     * where it is skipped), of which the call uses those below B1 */
    const uint64_t lo = (uint64_t)range * g_prime_range;
    const uint64_t hi = lo + g_prime_range + 1 < B1 ? lo + g_prime_range + 1 : B1;
    uint64_t q = 2;
    while (q < B1) {                                    /* ecm.c:1815-1822 */
        if (tape_push(t, GECM_OP_PRAC_BEGIN)) return -1;
        t->ptdups++;
        q *= 2;
    }
    size_t np = 0;
    uint64_t *primes = gecm_primes_range(lo, hi, &np);
    if (!primes) { gecm_tape_free(t); return -1; }
    /* "Stage 1 completed at prime PRIMES[last_pid - 1]" (ecm.c:1849): the last prime below B1, or PRIMES[0] itself
     * when the loop never ran */
    if (np) t->last_prime = primes[np - 1];
    else {
        size_t n2 = 0;
        uint64_t *nx = gecm_primes_range(lo, lo + 2000, &n2);     /* prime gaps here are far below 2000 */
        if (nx && n2) t->last_prime = nx[0];
        free(nx);
    }
    if (np > 1) {
        int nt = threads < 1 ? 1 : threads > 64 ? 64 : threads;
        if ((size_t)nt > (np - 1) / 4096 + 1) nt = (int)((np - 1) / 4096 + 1);
        tape_slice sl[64];
        pthread_t th[64];
        for (int i = 0; i < nt; i++) {
            sl[i].primes = primes; sl[i].B1 = B1;
            sl[i].lo = 1 + (np - 1) * (size_t)i / (size_t)nt;
            sl[i].hi = 1 + (np - 1) * (size_t)(i + 1) / (size_t)nt;
        }
        for (int i = 1; i < nt; i++)
            if (pthread_create(&th[i], NULL, tape_slice_run, &sl[i])) { tape_slice_run(&sl[i]); th[i] = 0; }
        tape_slice_run(&sl[0]);
        int rc = 0;
        size_t total = t->len;
        for (int i = 0; i < nt; i++) {
            if (i > 0 && th[i]) pthread_join(th[i], NULL);
            if (sl[i].rc) rc = sl[i].rc;
            total += sl[i].t.len;
        }
        uint8_t *all = rc ? NULL : (uint8_t *)realloc(t->ops, total + 8);
        if (!all) {
            for (int i = 0; i < nt; i++) free(sl[i].t.ops);
            free(primes);
            gecm_tape_free(t);
            return rc ? rc : -1;
        }
        t->ops = all;
        for (int i = 0; i < nt; i++) {
            if (sl[i].t.len) memcpy(t->ops + t->len, sl[i].t.ops, sl[i].t.len);
            t->len += sl[i].t.len;
            t->ptadds += sl[i].t.ptadds; t->ptdups += sl[i].t.ptdups; t->prac_calls += sl[i].t.prac_calls;
            t->swaps += sl[i].t.swaps;
            for (int r = 0; r < 4; r++) t->rule_count[r] += sl[i].t.rule_count[r];
            free(sl[i].t.ops);
        }
        memset(t->ops + t->len, 0, 8);
    }
    free(primes);
    if (!t->ops) { t->ops = (uint8_t *)calloc(8, 1); if (!t->ops) return -1; }
    return 0;
}

int gecm_tape_build_stage1(gecm_tape_t *t, uint64_t B1)
{
    if (B1 > g_prime_range) { memset(t, 0, sizeof *t); return -2; }     /* several ranges: one tape each */
    int rc = gecm_tape_build_stage1_range(t, B1, 0, 1);
    if (!rc && B1 > 2 && !t->last_prime) t->last_prime = 2;
    return rc;
}

/* What vececm prints and decides around one range (ecm.c:1215-1247): the sieved interval [rangemin, rangemax] with
 * rangemax = min(B2 + 1000, rangemin + 1e8), its prime count and first prime, the last prime below B1, and whether
 * the range ends before B1 — the reference then appends the batch to checkpoint.txt (it tests PRIMES[last_pid] <
 * B1 with last_pid = NUM_P, one past the list: the allocation is 1.25x an estimate and freshly mapped, so the word
 * read is 0 and the test holds whenever the list is exhausted, also for B1 in (99999989, 1e8] in a single range). */
int gecm_stage1_range_info(gecm_range_info *ri, uint64_t B1, uint64_t B2, uint32_t range)
{
    memset(ri, 0, sizeof *ri);
    if (range >= gecm_stage1_ranges_plan(B1)) return -2;
    if (B2 < B1) B2 = B1;
    ri->lo = (uint64_t)range * g_prime_range;
    ri->hi = B2 + 1000 < ri->lo + g_prime_range ? B2 + 1000 : ri->lo + g_prime_range;
    size_t np = 0;
    uint64_t *pr = gecm_primes_range(ri->lo, ri->hi + 1, &np);
    if (!pr) return -1;
    ri->nprimes = np;
    ri->first_prime = np ? pr[0] : 0;
    size_t i = 1;
    while (i < np && pr[i] < B1) i++;                    /* ecm.c:1824: last_pid */
    ri->last_prime = np ? pr[i - 1] : 0;
    ri->exhausted = (i >= np);
    free(pr);
    return 0;
}

void gecm_tape_free(gecm_tape_t *t)
{
    free(t->ops);
    memset(t, 0, sizeof *t);
}

/* the hash of the host sources this object was compiled from (Makefile: H_SHA); gecm_version() compares them */
#ifdef GECM_MANIFEST_FN
const char *GECM_MANIFEST_FN(void) { return GECM_MANIFEST; }
#endif
