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

#ifdef __cplusplus
}
#endif
#endif
