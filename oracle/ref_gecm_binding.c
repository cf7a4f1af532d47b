/* oracle/ref_gecm_binding.c — TEST INFRASTRUCTURE ONLY.
 *
 * The reference-side binding of INTEGRATION.md §2, compiled FOR REAL against the reference in the
 * build container: the reference's own main()/vececm()/prac()/vec_add() (unchanged, built from where
 * they lie under /root/reference) run with their five operator pointers (avx_ecm.h:205-209, bound in
 * main.c:642-702) served by libgecm's C ABI on the GPU.  oracle/Makefile target `refgpu` compiles the
 * reference's main.c with -Dvecmulmod52=gecm_bind_mulmod52 ... so the addresses main.c stores into
 * vecmulmod_ptr & co. are the functions below; nothing of the reference is copied or edited.
 *
 * The resulting binary (oracle/_ref/avx-ecm-52-gecm, git-ignored) travels to the GPU box, where
 * tests/test_gpu_dropin.py runs it and compares its save_b1.txt with the fixture the pure reference
 * wrote: the drop-in claim at the operator seam, end to end.  One operator call on 8 lanes is a
 * PCIe round trip, so this is a correctness demonstration at small B1, not a fast path.
 */
#include "avx_ecm.h"
#include "gecm.h"

static gecm_ctx *g_ctx;
static unsigned long g_calls;

static void report(void)
{
    char name[128] = "?";
    if (g_ctx) gecm_device_name(g_ctx, name, sizeof name);
    fprintf(stderr, "gecm binding: %lu operator calls served on %s\n", g_calls, name);
}

static gecm_ctx *ctx_for(monty *mdata, bignum *nvec)
{
    if (!g_ctx) {
        mpz_t n;
        char *s;
        mpz_init(n);
        extract_bignum_from_vec_to_mpz(n, nvec ? nvec : mdata->n, 0, NWORDS);
        s = mpz_get_str(NULL, 10, n);
        if (gecm_create(&g_ctx, 0, s, DIGITBITS)) {
            fprintf(stderr, "gecm binding: %s\n", gecm_last_error());
            exit(3);
        }
        gecm_config cfg;
        gecm_get_config(g_ctx, &cfg);
        if (cfg.nwords != (int)NWORDS) {
            fprintf(stderr, "gecm binding: NWORDS mismatch %d vs %u\n", cfg.nwords, NWORDS);
            exit(3);
        }
        free(s);
        mpz_clear(n);
        atexit(report);
    }
    g_calls++;
    return g_ctx;
}

#define CHECK(call)                                                          \
    do {                                                                     \
        if ((call) < 0) {                                                    \
            fprintf(stderr, "gecm binding: %s\n", gecm_last_error());        \
            exit(3);                                                         \
        }                                                                    \
    } while (0)

void gecm_bind_mulmod52(bignum *a, bignum *b, bignum *c, bignum *n, bignum *s, monty *mdata)
{
    (void)s;
    CHECK(gecm_vecmulmod(ctx_for(mdata, n), a->data, b->data, c->data, VECLEN));
}

void gecm_bind_sqrmod52(bignum *a, bignum *c, bignum *n, bignum *s, monty *mdata)
{
    (void)s;
    CHECK(gecm_vecsqrmod(ctx_for(mdata, n), a->data, c->data, VECLEN));
}

void gecm_bind_addmod52(bignum *a, bignum *b, bignum *c, monty *mdata)
{
    CHECK(gecm_vecaddmod(ctx_for(mdata, NULL), a->data, b->data, c->data, VECLEN));
}

void gecm_bind_submod52(bignum *a, bignum *b, bignum *c, monty *mdata)
{
    CHECK(gecm_vecsubmod(ctx_for(mdata, NULL), a->data, b->data, c->data, VECLEN));
}

void gecm_bind_addsub52(bignum *a, bignum *b, bignum *sum, bignum *diff, monty *mdata)
{
    CHECK(gecm_vecaddsubmod(ctx_for(mdata, NULL), a->data, b->data, sum->data, diff->data, VECLEN));
}
