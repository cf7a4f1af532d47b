/* cunningham.h — see cunningham.c */
#ifndef CUNNINGHAM_H
#define CUNNINGHAM_H
#include "mpl.h"
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int form;        /* 0 none; +1: N | 2^k - 1; -1: N | 2^k + 1; 2: pseudo-Mersenne, 2^k mod N = c < 2^digitbits */
    int k;
    uint64_t c;
} cunningham_form;

/* main.c:405-441 */
void cunningham_detect(cunningham_form *f, const mpl_t *N, int digitbits);
/* main.c:187-352 (find_primitive_factor with base 2): prim = the "primitive" part of 2^e -/+ 1 with respect
 * to the odd primes < 1000 of e.  sign = +1 for 2^e - 1, -1 for 2^e + 1.  The reference's "gen:" progress
 * lines are appended to log.  Returns 0, or -1 if e has more than 3 distinct odd prime factors (the reference
 * exits there) or a number outgrows mpl_t. */
int cunningham_primitive(mpl_t *prim, int e, int sign, char *log, size_t loglen);

#ifdef __cplusplus
}
#endif
#endif
