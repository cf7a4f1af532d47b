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

int gecm_tape_build_stage1(gecm_tape_t *t, uint64_t B1)
{
    memset(t, 0, sizeof *t);
    uint64_t q = 2;
    while (q < B1) {                                    /* ecm.c:1815-1822 */
        if (tape_push(t, GECM_OP_PRAC_BEGIN)) return -1;
        t->ptdups++;
        q *= 2;
    }
    if (B1 > 2) t->last_prime = 2;
    size_t np = 0;
    uint64_t *primes = gecm_primes_range(0, B1, &np);
    if (!primes) return -1;
    for (size_t k = 1; k < np; k++) {                   /* ecm.c:1824-1832 */
        uint64_t c = 1;
        q = primes[k];
        do {
            int rc = gecm_tape_append_prac(t, q);
            if (rc) { free(primes); return rc; }
            c *= q;
        } while ((c * q) < B1);
        t->last_prime = q;
    }
    free(primes);
    if (!t->ops) { t->ops = (uint8_t *)calloc(8, 1); if (!t->ops) return -1; }
    return 0;
}

void gecm_tape_free(gecm_tape_t *t)
{
    free(t->ops);
    memset(t, 0, sizeof *t);
}
