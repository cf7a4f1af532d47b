/* gecm_pair.c — see gecm_pair.h. */
#include "gecm_pair.h"
#include "gecm_plan.h"
#include <stdlib.h>
#include <string.h>

static uint32_t gcd_u32(uint32_t a, uint32_t b)
{
    while (b) { uint32_t t = a % b; a = b; b = t; }
    return a;
}

uint32_t gecm_s2_default_D(uint64_t B1)
{
    /* main.c:838-872 */
    if (B1 <= 60) return 30;
    if (B1 <= 128) return 60;
    if (B1 <= 256) return 120;
    if (B1 <= 512) return 210;
    if (B1 <= 2048) return 385;
    if (B1 <= 4096) return 1155;
    return 2310;
}

int gecm_s2_plan_init(gecm_s2_plan *p, uint32_t D, uint32_t U)
{
    memset(p, 0, sizeof *p);
    if (D < 6 || U < 1) return -1;
    p->D = D; p->U = U; p->L = 2 * U; p->umax = U * D;
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < 2 * D; i++) cnt += gcd_u32(i, 2 * D) == 1;
    p->R = cnt + 3;
    p->map = (uint32_t *)calloc((size_t)U * (D + 1) + 3, sizeof(uint32_t));
    p->keep_words = ((size_t)p->umax + 32) / 32 + 1;
    p->keep = (uint32_t *)calloc(p->keep_words, sizeof(uint32_t));
    if (!p->map || !p->keep) { gecm_s2_plan_free(p); return -1; }
    /* ecm.c:301-329: entries 1, 2 = Q, 2Q; then every j coprime to D (and j = D) in order */
    uint32_t m = 3;
    p->map[1] = 1;
    p->map[2] = 2;
    for (uint32_t blk = 0; blk < U; blk++)
        for (uint32_t j = blk ? 1 : 3; j <= (blk ? D - 1 : D); j++) {
            if (gcd_u32(j, D) == 1 || (blk == 0 && j == D)) p->map[blk * D + j] = m++;
        }
    p->npb = m;
    for (uint32_t j = 1; j <= p->umax; j++)
        if (p->map[j]) p->keep[j >> 5] |= 1u << (j & 31);
    return 0;
}

void gecm_s2_plan_free(gecm_s2_plan *p)
{
    free(p->map);
    free(p->keep);
    memset(p, 0, sizeof *p);
}

/* ---- PAIR ------------------------------------------------------------------------------- */
/* FIFO of giant-step indices waiting for a partner, one per residue class (queue.c:32-102) */
typedef struct {
    uint32_t *buf;
    uint32_t cap, head, count;
} fifo;

static int fifo_push(fifo *f, uint32_t x)
{
    if (f->count == f->cap) {
        uint32_t ncap = f->cap ? f->cap * 2 : 16;
        uint32_t *nb = (uint32_t *)malloc(ncap * sizeof(uint32_t));
        if (!nb) return -1;
        for (uint32_t i = 0; i < f->count; i++) nb[i] = f->buf[(f->head + i) % f->cap];
        free(f->buf);
        f->buf = nb; f->cap = ncap; f->head = 0;
    }
    f->buf[(f->head + f->count) % f->cap] = x;
    f->count++;
    return 0;
}

static uint32_t fifo_pop(fifo *f)
{
    uint32_t x = f->buf[f->head];
    f->head = (f->head + 1) % f->cap;
    f->count--;
    return x;
}

typedef struct {
    gecm_pairmap *pm;
    size_t cap;
} emitter;

static int emit(emitter *e, uint64_t v, uint64_t u)
{
    gecm_pairmap *pm = e->pm;
    if (pm->steps == e->cap) {
        size_t nc = e->cap ? e->cap * 2 : 4096;
        uint32_t *nv = (uint32_t *)realloc(pm->v, nc * sizeof(uint32_t));
        if (!nv) return -1;
        pm->v = nv;
        uint32_t *nu = (uint32_t *)realloc(pm->u, nc * sizeof(uint32_t));
        if (!nu) return -1;
        pm->u = nu;
        e->cap = nc;
    }
    pm->v[pm->steps] = (uint32_t)v;
    pm->u[pm->steps] = (uint32_t)u;
    pm->steps++;
    return 0;
}

int gecm_pair(gecm_pairmap *out, uint64_t B1, uint64_t B2, uint32_t D, uint32_t U)
{
    memset(out, 0, sizeof *out);
    const int64_t w = D;
    const uint64_t L = 2ull * U;
    const int64_t umax = (int64_t)D * U;
    /* residue classes coprime to 2w (Qmap/Qrmap, main.c:723-748) */
    uint32_t *cls = (uint32_t *)malloc(2 * (size_t)D * sizeof(uint32_t));   /* residue -> class */
    uint32_t *res = (uint32_t *)malloc(2 * (size_t)D * sizeof(uint32_t));   /* class -> residue */
    if (!cls || !res) { free(cls); free(res); return -1; }
    uint32_t ncls = 0;
    for (uint32_t k = 0; k < 2 * D; k++) {
        if (gcd_u32(k, 2 * D) == 1) { cls[k] = ncls; res[ncls++] = k; }
        else cls[k] = UINT32_MAX;
    }
    fifo *Q = (fifo *)calloc(ncls, sizeof(fifo));
    size_t np = 0;
    uint64_t *primes = gecm_primes_range(B1, B2, &np);
    if (!Q || !primes) { free(cls); free(res); free(Q); free(primes); return -1; }
    emitter em = {out, 0};
    uint64_t amin = (B1 + (uint64_t)w) / (2 * (uint64_t)w);      /* ecm.c:2571 */
    out->amin = (uint32_t)amin;
    int rc = 0;
    for (size_t pid = 0; pid < np && !rc; pid++) {
        const uint64_t s = primes[pid];
        const uint64_t a = (s + (uint64_t)w) / (2 * (uint64_t)w);
        out->nump++;
        while (a >= amin + L && !rc) {                             /* ecm.c:2611-2685 */
            const uint64_t oldmin = amin;
            amin = amin + L - U;
            for (uint32_t i = 0; i < ncls && !rc; i++) {
                const int64_t r = res[i];
                const int64_t qq = r > w ? 2 * w - r : r;
                uint32_t len = Q[i].count;
                for (uint32_t j = 0; j < len && !rc; j++) {
                    uint32_t ap = fifo_pop(&Q[i]);
                    if (ap < (uint32_t)amin) { rc = emit(&em, 2ull * ap - oldmin, (uint64_t)qq); out->pairs++; }
                    else rc = fifo_push(&Q[i], ap);
                }
            }
            if (!rc) rc = emit(&em, 0, 0);
        }
        const int64_t q = (int64_t)s - 2 * (int64_t)a * w;       /* ecm.c:2687-2691 */
        const int64_t mq = q < 0 ? -q : 2 * w - q;
        int64_t u;
        do {
            fifo *partner = &Q[cls[mq]];
            if (partner->count > 0) {                              /* ecm.c:2696-2776 */
                uint64_t ap = fifo_pop(partner);
                u = w * (int64_t)(a - ap) + q;
                if (u > umax) {
                    int64_t qq = q < 0 ? -q : q;
                    if (q >= 0 && qq >= w) qq = 2 * w - qq;
                    rc = emit(&em, 2 * ap - amin, (uint64_t)qq);
                } else {
                    rc = emit(&em, a + ap - amin, (uint64_t)u);
                }
                out->pairs++;
            } else {                                               /* ecm.c:2777-2791 */
                rc = fifo_push(&Q[cls[q < 0 ? 2 * w + q : q]], (uint32_t)a);
                u = 0;
            }
        } while (u > umax && !rc);
    }
    for (uint32_t i = 0; i < ncls && !rc; i++) {                   /* ecm.c:2796-2843 */
        const int64_t r = res[i];
        const int64_t qq = r > w ? 2 * w - r : r;
        while (Q[i].count && !rc) {
            uint64_t ap = fifo_pop(&Q[i]);
            rc = emit(&em, 2 * ap - amin, (uint64_t)qq);
            out->pairs++;
        }
    }
    for (uint32_t i = 0; i < ncls; i++) free(Q[i].buf);
    free(Q); free(cls); free(res); free(primes);
    if (rc) gecm_pairmap_free(out);
    return rc;
}

void gecm_pairmap_free(gecm_pairmap *pm)
{
    free(pm->v);
    free(pm->u);
    memset(pm, 0, sizeof *pm);
}

/* ---- the device tape of one range (see gecm_pair.h) ---------------------------------------------------------- */
static int cmp_pair_slot(const void *a, const void *b)
{
    const uint32_t *x = (const uint32_t *)a, *y = (const uint32_t *)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    return x[1] < y[1] ? -1 : (x[1] > y[1]);
}

int gecm_s2_tape_build(gecm_s2_tape *out, const gecm_s2_plan *p, uint32_t steps, const uint32_t *pm_v, const uint32_t *pm_u,
                       uint32_t amin, uint32_t chunk, uint32_t ring, uint32_t *bad)
{
    memset(out, 0, sizeof *out);
    if (bad) *bad = 0;
    if (!chunk || (ring & (ring - 1)) || 4ull * p->U + chunk > ring) return -2;
    uint64_t shifts = 0;
    for (uint32_t i = 0; i < steps; i++) shifts += (pm_v[i] == 0 && pm_u[i] == 0);
    const uint64_t E = 2ull * p->L + 2ull * p->U * shifts;
    const uint64_t g0 = shifts ? E - 2ull * p->U : 0;
    /* marks: ceil(g0 / chunk) + ceil((E - g0) / chunk), each one pair of words */
    const size_t max_marks = (size_t)(g0 / chunk) + (size_t)((E - g0) / chunk) + 4;
    uint32_t *tape = (uint32_t *)malloc(((size_t)steps + max_marks) * 2 * sizeof(uint32_t));
    if (!tape) return -1;
    const size_t cap = ((size_t)steps + max_marks) * 2;
    size_t nt = 0;
    uint32_t run_amin = amin;
    uint64_t adds = 2ull * p->L - 1, inv = 2, paired = 0, devinv = 0;    /* ecm.c:2401-2429 */
    uint64_t generated = 0;                      /* giant steps the device will have made so far */
    const uint64_t base = 2ull * amin;           /* absolute number of giant step 0 (in units of D) */
#define NEED(upto)                                                                                       \
    while (generated < (upto)) {                                                                         \
        const uint64_t lim = generated < g0 ? g0 : E;                                                    \
        const uint64_t n = lim - generated < chunk ? lim - generated : chunk;                            \
        if (nt + 2 > cap) { free(tape); return -1; }                                                     \
        tape[nt++] = GECM_S2_GEN;                                                                        \
        tape[nt++] = (uint32_t)n | ((generated >= g0 || amin == 0) ? 0x80000000u : 0u);                  \
        generated += n; devinv++;                                                                        \
    }
    NEED(2ull * p->L);
    for (uint32_t i = 0; i < steps; i++) {
        if (pm_v[i] == 0 && pm_u[i] == 0) {
            run_amin += p->U;                                            /* ecm.c:2496 */
            adds += 2ull * p->U; inv++;
            NEED(2ull * run_amin - base + 2ull * p->L);
        } else {
            const uint32_t pa = pm_v[i] - run_amin, pb = pm_u[i];
            if (pm_v[i] < run_amin || pa >= 2 * p->L || pb > p->umax || p->map[pb] == 0) {     /* ecm.c:2508-2517 */
                free(tape);
                if (bad) *bad = i;
                return -2;
            }
            const uint64_t absidx = 2ull * run_amin - base + pa;
            if (nt + 2 > cap) { free(tape); return -1; }
            tape[nt++] = (uint32_t)(absidx & (ring - 1));
            tape[nt++] = p->map[pb];
            paired++;
        }
    }
#undef NEED
    for (size_t i = 0; i < nt;) {
        if (tape[i] == GECM_S2_GEN) { i += 2; continue; }
        size_t j = i;
        while (j < nt && tape[j] != GECM_S2_GEN) j += 2;
        qsort(tape + i, (j - i) / 2, 2 * sizeof(uint32_t), cmp_pair_slot);
        i = j;
    }
    out->words = tape; out->nwords = nt;
    out->adds = adds; out->inv = inv; out->paired = paired; out->devinv = devinv; out->amin_last = run_amin;
    return 0;
}

/* the hash of the host sources this object was compiled from (Makefile: H_SHA); gecm_version() compares them */
#ifdef GECM_MANIFEST_FN
const char *GECM_MANIFEST_FN(void) { return GECM_MANIFEST; }
#endif
