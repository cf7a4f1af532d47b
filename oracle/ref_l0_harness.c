/* oracle/ref_l0_harness.c — TEST INFRASTRUCTURE ONLY.
 *
 * My own driver around the REFERENCE's L0 vector arithmetic, used (in the build
 * container, where /root/reference exists) to generate tests/golden/l0_*.json.
 * It includes the reference's header and links the reference's vecarith52.c / vecarith.c /
 * vec_common.c / util.c compiled from where they lie (oracle/Makefile target `harness`);
 * nothing of the reference is copied here.
 *
 * The five operators exercised are the function pointers bound in main.c:642-702:
 *   vecmulmod52 (vecarith52.c:2438), vecsqrmod52 (:3317), vecaddmod52 (:4550),
 *   vecsubmod52 (:4684), vec_simul_addsub52 (:4877)   [DIGITBITS=52]
 *   vecmulmod (vecarith.c:221), vecsqrmod (:889), vecaddmod (:2806), vecsubmod (:2870),
 *   vec_simul_addsub (:2726)                           [DIGITBITS=32]
 * Montgomery constants are set up as main.c:620-640 does.
 *
 * usage: ref_l0_harness <nwords> <N hex>   < stdin: per vector VECLEN lines "a_hex b_hex"
 * output: per input line: "mul sqr add sub asum adiff" (hex), where asum/adiff come from
 * the fused add+sub operator.
 */
#include "avx_ecm.h"

static void put_lane(bignum *v, mpz_t x, int lane)
{
    /* layout data[lane + limb*VECLEN] (main.c:117-138), all NWORDS limbs written */
    mpz_t t; mpz_init_set(t, x);
    for (uint32_t i = 0; i < NWORDS; i++) {
        v->data[lane + i * VECLEN] = (base_t)(mpz_get_ui(t) & MAXDIGIT);
        mpz_tdiv_q_2exp(t, t, DIGITBITS);
    }
    mpz_clear(t);
}

static void get_lane(mpz_t x, bignum *v, int lane)
{
    mpz_set_ui(x, 0);
    for (int i = (int)NWORDS - 1; i >= 0; i--) {
        mpz_mul_2exp(x, x, DIGITBITS);
        mpz_add_ui(x, x, v->data[lane + i * VECLEN]);
    }
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s nwords Nhex\n", argv[0]); return 2; }
    NWORDS = atoi(argv[1]);
    NBLOCKS = NWORDS / BLOCKWORDS;
    MAXBITS = NWORDS * DIGITBITS;
    mpz_t n, r, x, y;
    mpz_init(n); mpz_init(r); mpz_init(x); mpz_init(y);
    mpz_set_str(n, argv[2], 16);

    monty *m = monty_alloc();
    m->isMersenne = 0;
    m->nbits = mpz_sizeinbase(n, 2);
    mpz_set_ui(r, 1);
    mpz_mul_2exp(r, r, DIGITBITS * NWORDS);
    mpz_invert(m->nhat, n, r);
    mpz_sub(m->nhat, r, m->nhat);
    for (int i = 0; i < VECLEN; i++) {
        put_lane(m->n, n, i);
        m->vrho[i] = mpz_get_ui(m->nhat) & MAXDIGIT;
    }

    bignum *a = vecInit(), *b = vecInit(), *c = vecInit(), *s = vecInit();
    bignum *q = vecInit(), *ad = vecInit(), *sb = vecInit(), *as = vecInit(), *df = vecInit();
    static char la[8192], lb[8192];
    for (;;) {
        int got = 0;
        for (int i = 0; i < VECLEN; i++) {
            if (scanf("%8191s %8191s", la, lb) != 2) break;
            mpz_set_str(x, la, 16); mpz_set_str(y, lb, 16);
            put_lane(a, x, i); put_lane(b, y, i);
            got++;
        }
        if (got == 0) break;
#if DIGITBITS == 52
        vecmulmod52(a, b, c, m->n, s, m);
        vecsqrmod52(a, q, m->n, s, m);
        vecaddmod52(a, b, ad, m);
        vecsubmod52(a, b, sb, m);
        vec_simul_addsub52(a, b, as, df, m);
#else
        vecmulmod(a, b, c, m->n, s, m);
        vecsqrmod(a, q, m->n, s, m);
        vecaddmod(a, b, ad, m);
        vecsubmod(a, b, sb, m);
        vec_simul_addsub(a, b, as, df, m);
#endif
        for (int i = 0; i < got; i++) {
            get_lane(x, c, i);  gmp_printf("%Zx ", x);
            get_lane(x, q, i);  gmp_printf("%Zx ", x);
            get_lane(x, ad, i); gmp_printf("%Zx ", x);
            get_lane(x, sb, i); gmp_printf("%Zx ", x);
            get_lane(x, as, i); gmp_printf("%Zx ", x);
            get_lane(x, df, i); gmp_printf("%Zx\n", x);
        }
        if (got < VECLEN) break;
    }
    return 0;
}
