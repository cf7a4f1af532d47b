/* gecm_pair.h — host-side stage-2 planning: wheel parameters, baby-step index map and
 * Montgomery's PAIR prime pairing (reference: main.c:834-951, ecm.c:248-340, ecm.c:2559-2910).
 * Identical for every curve, thread and GPU; computed once per B2 range on the host. */
#ifndef GECM_PAIR_H
#define GECM_PAIR_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* D (= w) chosen from B1 as thread_init does (main.c:838-872) */
uint32_t gecm_s2_default_D(uint64_t B1);
/* The reference picks U through an uninitialised variable (main.c:912, 943); every run observed
 * chose 16.  This build makes it explicit. */
#define GECM_S2_DEFAULT_U 16u

typedef struct {
    uint32_t D, U, L;        /* L = 2U (main.c:950) */
    uint32_t R;              /* phi(2D) + 3 (main.c:874-882) */
    uint32_t umax;           /* U * D */
    uint32_t *map;           /* size U*(D+1)+3: j -> table index, 0 = not stored (ecm.c:301-329) */
    uint32_t npb;            /* number of table entries incl. unused entry 0 = lastMapID + 1 */
    uint32_t *keep;          /* bitmap over j in [0, umax]: bit set iff map[j] > 0 */
    size_t keep_words;
} gecm_s2_plan;

int gecm_s2_plan_init(gecm_s2_plan *p, uint32_t D, uint32_t U);
void gecm_s2_plan_free(gecm_s2_plan *p);

typedef struct {
    uint32_t *v, *u;         /* pairmap_v / pairmap_u (ecm.c:2559); (0,0) = advance the window */
    uint32_t steps;
    uint32_t amin;           /* (B1 + w) / (2w) at entry (ecm.c:2571) */
    uint32_t pairs, nump;    /* printed at ecm.c:2904-2905 */
} gecm_pairmap;

/* pair() of the reference for primes in [B1, B2) */
int gecm_pair(gecm_pairmap *out, uint64_t B1, uint64_t B2, uint32_t D, uint32_t U);
void gecm_pairmap_free(gecm_pairmap *pm);

#ifdef __cplusplus
}
#endif
#endif
