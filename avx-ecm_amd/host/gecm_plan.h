/* gecm_plan.h — host-side planning for the device hot path: prime supply and the stage-1
 * op tape (the reference's ecm_stage1 + prac control flow, evaluated once per B1). */
#ifndef GECM_PLAN_H
#define GECM_PLAN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* All primes p with lo <= p < hi (segmented sieve of Eratosthenes; hi <= 2^40).
 * Returns a malloc'ed array and its length in *count; NULL on allocation failure. */
uint64_t *gecm_primes_range(uint64_t lo, uint64_t hi, size_t *count);

typedef struct {
    uint8_t *ops;        /* one byte per event, see csrc/gecm_tape.h */
    size_t len;
    uint64_t ptadds;     /* point additions  (reference counter work->ptadds, ecm.c:441)  */
    uint64_t ptdups;     /* point doublings  (work->ptdups, ecm.c:455)                    */
    uint64_t prac_calls; /* number of prac() invocations                                  */
    uint64_t last_prime; /* largest prime processed (printed at ecm.c:1849)               */
    uint64_t rule_count[4]; /* rule 3, 4, 5, 9 */
    uint64_t swaps;
} gecm_tape_t;

/* PRAC multiplier cost model, ecm.c:479-563 (exposed for tests). */
double gecm_lucas_cost(uint64_t n, double v);
/* index (0..9) of the multiplier prac() selects for c, ecm.c:574-582 */
int gecm_prac_best_multiplier(uint64_t c);

/* Append the chain of prac(c) (ecm.c:565-884) to the tape. */
int gecm_tape_append_prac(gecm_tape_t *t, uint64_t c);
/* Whole stage 1 for bound B1 (ecm.c:1806-1854): one doubling per power of two below B1, then
 * prac(q) for every odd prime q < B1, repeated while q^k < B1. */
int gecm_tape_build_stage1(gecm_tape_t *t, uint64_t B1);
void gecm_tape_free(gecm_tape_t *t);

/* Stage 1 above one prime range (ecm.c:1209-1312): vececm calls ecm_stage1 once per range of PRIME_RANGE primes. */
#define GECM_PRIME_RANGE 100000000ull                   /* main.c:581 */
uint32_t gecm_stage1_ranges_plan(uint64_t B1);               /* ceil(B1 / 1e8), at least 1 */
/* The tape of ONE ecm_stage1 call: the 2-power doublings (again in every range), then prac(q) for the primes q < B1
 * of [range * 1e8, (range + 1) * 1e8) except the first one of the list (ecm.c:1815-1832).  last_prime, ptadds,
 * ptdups are those of this call.  `threads` worker threads compile slices of the range. */
int gecm_tape_build_stage1_range(gecm_tape_t *t, uint64_t B1, uint32_t range, int threads);
typedef struct {
    uint64_t lo, hi;        /* the sieved interval, printed as "Found %lu primes in range [lo : hi]" (ecm.c:1228) */
    uint64_t nprimes;
    uint64_t first_prime;   /* P_MIN: "Commencing Stage 1 @ prime" (ecm.c:1233) */
    uint64_t last_prime;    /* PRIMES[last_pid - 1]: "Stage 1 completed at prime", the checkpoint's B1 (ecm.c:1849, 1244) */
    int exhausted;          /* the range holds no prime >= B1: the reference writes checkpoint.txt after it (ecm.c:1237) */
} gecm_range_info;
int gecm_stage1_range_info(gecm_range_info *ri, uint64_t B1, uint64_t B2, uint32_t range);
/* Test hook (not in include/gecm.h): another PRIME_RANGE for the three functions above, so that tests walk the
 * multi-range path at small B1 next to the oracle run the same way; 0 restores 1e8.  Process-wide. */
void gecm_plan_set_prime_range_for_tests(uint64_t range);

#ifdef __cplusplus
}
#endif
#endif
