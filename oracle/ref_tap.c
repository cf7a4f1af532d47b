/* oracle/ref_tap.c — TEST INFRASTRUCTURE ONLY.
 *
 * A tap on the reference's own stage-2 accumulator.  oracle/Makefile target `reftap` compiles the reference's
 * ecm.c (from where it lies, unmodified) with -Dextract_bignum_from_vec_to_mpz=gecm_tap_extract, so every lane
 * vececm extracts (ecm.c:1256, 1335, 1489, ...) passes through the function below, which calls the reference's real
 * extract_bignum_from_vec_to_mpz (main.c:63) and, if GECM_TAP_FILE is set, appends "vector-address lane value-hex"
 * to that file.  The last VECLEN lines of a run with curves == VECLEN are work->stg2acc as the reference reads it at
 * ecm.c:1489, i.e. the accumulator of ecm_stage2_init (ecm.c:2201-2340) + ecm_stage2_pair (ecm.c:2342-2540),
 * CROSS_PRODUCT_INV (ecm.c:1857-1859), in the reference's Montgomery radix.
 * tests/golden/make_golden.py --only stage2acc turns them into tests/golden/stage2_acc.json.
 */
#include "avx_ecm.h"

void gecm_tap_extract(mpz_t dest, bignum *vec_src, int num, int sz)
{
    extract_bignum_from_vec_to_mpz(dest, vec_src, num, sz);
    const char *f = getenv("GECM_TAP_FILE");
    if (f) {
        FILE *o = fopen(f, "a");
        if (o) {
            gmp_fprintf(o, "%p %d %Zx\n", (void *)vec_src, num, dest);
            fclose(o);
        }
    }
}
