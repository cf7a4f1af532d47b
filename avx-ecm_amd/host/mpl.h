/* mpl.h — "multi-precision lite": the small unsigned big-integer kit the host side of libgecm
 * needs (curve setup, Montgomery constants, gcd, hex/decimal I/O).  The reference uses GMP for
 * these (mpz_invert ecm.c:1745,1759; mpz_gcd ecm.c:2545; gmp_fprintf %Zx ecm.c:1374-1380); they
 * are exact integer functions with unique results, so any correct implementation gives
 * identical output.  Self-contained on purpose: the product must not depend on a GMP being
 * installed on the GPU box.
 */
#ifndef MPL_H
#define MPL_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MPL_MAXL 136 /* 32-bit limbs: 4352 bits, enough for products of two 2176-bit values */

typedef struct {
    int n;                 /* used limbs; d[n-1] != 0; n == 0 means zero */
    uint32_t d[MPL_MAXL];
} mpl_t;

void mpl_set_u64(mpl_t *r, uint64_t v);
uint64_t mpl_get_u64(const mpl_t *a);           /* low 64 bits */
int mpl_set_str(mpl_t *r, const char *s);       /* decimal, or hex with 0x prefix; 0 on success */
/* returns number of chars written (excluding NUL); buf must hold MPL_MAXL*10+2 bytes */
int mpl_get_hex(char *buf, const mpl_t *a);     /* lower case, no prefix, "0" for zero (as %Zx) */
int mpl_get_dec(char *buf, const mpl_t *a);
int mpl_bits(const mpl_t *a);
/* what GMP's mpz_sizeinbase(a, 10) returns: floor(bits * log10(2)) + 1, exact or one too large; 1 for zero */
int mpl_sizeinbase10(const mpl_t *a);
int mpl_is_zero(const mpl_t *a);
int mpl_is_odd(const mpl_t *a);
int mpl_cmp(const mpl_t *a, const mpl_t *b);
int mpl_cmp_u64(const mpl_t *a, uint64_t v);
void mpl_add(mpl_t *r, const mpl_t *a, const mpl_t *b);
void mpl_add_u64(mpl_t *r, const mpl_t *a, uint64_t v);
void mpl_sub(mpl_t *r, const mpl_t *a, const mpl_t *b);      /* requires a >= b */
void mpl_mul(mpl_t *r, const mpl_t *a, const mpl_t *b);
void mpl_mul_u64(mpl_t *r, const mpl_t *a, uint64_t v);
void mpl_shl(mpl_t *r, const mpl_t *a, unsigned bits);
void mpl_shr(mpl_t *r, const mpl_t *a, unsigned bits);
void mpl_divrem(mpl_t *q, mpl_t *r, const mpl_t *a, const mpl_t *b); /* q or r may be NULL */
void mpl_mod(mpl_t *r, const mpl_t *a, const mpl_t *m);
void mpl_mulmod(mpl_t *r, const mpl_t *a, const mpl_t *b, const mpl_t *m);
void mpl_addmod(mpl_t *r, const mpl_t *a, const mpl_t *b, const mpl_t *m); /* a,b < m */
void mpl_submod(mpl_t *r, const mpl_t *a, const mpl_t *b, const mpl_t *m); /* a,b < m */
void mpl_powmod(mpl_t *r, const mpl_t *a, const mpl_t *e, const mpl_t *m);
void mpl_gcd(mpl_t *r, const mpl_t *a, const mpl_t *b);
/* r = a^-1 mod m; returns 1 if it exists, else 0 (r undefined) — like mpz_invert */
int mpl_invmod(mpl_t *r, const mpl_t *a, const mpl_t *m);
/* Miller-Rabin, `reps` fixed small bases + trial division; 1 = probable prime */
int mpl_probab_prime(const mpl_t *a, int reps);

/* fixed-width limb import/export: value <-> count limbs of `bits` bits each stored in
 * consecutive elements `stride` apart (uint32_t or uint64_t containers) */
void mpl_to_limbs32(uint32_t *out, size_t stride, int count, int bits, const mpl_t *a);
void mpl_from_limbs32(mpl_t *r, const uint32_t *in, size_t stride, int count, int bits);
void mpl_to_limbs64(uint64_t *out, size_t stride, int count, int bits, const mpl_t *a);
void mpl_from_limbs64(mpl_t *r, const uint64_t *in, size_t stride, int count, int bits);

#ifdef __cplusplus
}
#endif
#endif
