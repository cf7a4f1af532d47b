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

/* ---- the device's stage-2 tape for one range (ecm_stage2_pair, ecm.c:2342-2540, resolved on the host) ----
 * words[2i], words[2i+1]: (GECM_S2_GEN, n | flag) = make the next n giant steps (flag = bit 31: as ONE chain with
 * ONE inversion), else (ring slot, table index) = one pair.  A pair (v,u) of the map refers to giant step number
 * 2*amin_now + (v - amin_now) counted from [2*amin*D]Q in steps of D (ecm.c:2378, 2505) and to table entry map[u].
 * The reference makes E = 2L + 2U * (window shifts) giant steps in the range and inverts them in batches: the first
 * 2L together, then the 2U new ones of every shift (ecm.c:2425, 2499).  The tape's chunks (at most `chunk` steps)
 * never go beyond E, and the last one is exactly the reference's last batch [g0, E), flagged: when a curve's
 * inversions fail (it found its factor already: every Z is 0 modulo it) the gcd the reference's accumulator ends up
 * carrying is the one of its last failing batch (ecm.c:1925-1939 overwrites stg2acc each time), and the device's
 * record (the last failure wins) is then taken over the same points.  A range that starts at amin = 0 (B1 < D: the
 * reference's first giant steps are [0]Q and a ladder with a negative multiplier) has every chunk flagged: only the
 * plain chain reproduces those.  Pairs between two marks are sorted by ring slot (their product is order-
 * independent; the device re-reads a ring row only when it changes).
 * adds / inv / paired: the reference's counters for the range (ecm.c:2401-2429, 2496; without the two ladders);
 * devinv: inversions the device makes.  Returns 0, -1 (out of memory) or -2 (bad entry, index in *bad: ecm.c:2508-2517;
 * or U too large for the ring: 4U + chunk must not exceed ring). */
#define GECM_S2_GEN 0xffffffffu
typedef struct {
    uint32_t *words;
    size_t nwords;
    uint64_t adds, inv, paired, devinv;
    uint32_t amin_last;
} gecm_s2_tape;
int gecm_s2_tape_build(gecm_s2_tape *out, const gecm_s2_plan *p, uint32_t steps, const uint32_t *pm_v, const uint32_t *pm_u,
                       uint32_t amin, uint32_t chunk, uint32_t ring, uint32_t *bad);

#ifdef __cplusplus
}
#endif
#endif
