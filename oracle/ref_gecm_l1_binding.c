/* oracle/ref_gecm_l1_binding.c — TEST INFRASTRUCTURE ONLY.
 *
 * The reference-side binding of INTEGRATION.md §3 — the production seam — compiled FOR REAL against the reference in
 * the build container.  vececm (ecm.c:1077-1544, unchanged, built from where it lies under /root/reference) drives
 * its four per-thread phases through a thread pool: it registers ecm_build_curve_work_fcn, ecm_stage1_work_fcn,
 * ecm_stage2_init_work_fcn and ecm_stage2_work_fcn with tpool_add_work_fcn (ecm.c:1130-1133) and runs them with
 * tpool_go.  oracle/Makefile target `refl1` compiles the reference's ecm.c with
 * -Dtpool_add_work_fcn=gecm_bind_add_work_fcn, so those four registrations arrive here, and this file registers the
 * GPU versions below in their place.  Everything else is the reference: main(), the expression parser, the sieve,
 * pair(), the thread pool, build_one_curve, the save_b1.txt / checkpoint.txt / ecm_results.txt writers and the
 * factor scan (which reads P->X, P->Z and work->stg2acc, where the GPU versions leave their results).
 *
 *   phase 0  the reference's own ecm_build_curve_work_fcn (host), then gecm_upload_points(P->X, P->Z, work->s)
 *   phase 1  gecm_stage1_range(STAGE1_MAX, range of the PRIMES list vececm has just sieved), results back into P
 *   phase 2  gecm_stage2_init(work->D, work->U)
 *   phase 3  gecm_stage2_pair(pairmap_steps, pairmap_v, pairmap_u, work->amin) with the map the reference's own
 *            pair() made, accumulator back into work->stg2acc
 * In the reference's special-form mode (isMersenne != 0) the context is made on mdata->n = 2^k -/+ 1 or 2^k - c as in
 * every other mode; only the vectors differ there (plain residues instead of Montgomery forms) and are converted.
 *
 * The binary (oracle/_ref/avx-ecm-52-l1, git-ignored) travels to the GPU box; tests/test_gpu_dropin.py runs it and
 * compares the files it writes with the ones the pure reference wrote.
 */
#include "avx_ecm.h"
#include "threadpool.h"
#include "eratosthenes/soe.h"
#include "gecm.h"

void tpool_add_work_fcn(tpool_t *tdata, void *work_fcn);      /* the reference's, threadpool.c:427 */

#define MAX_THREADS 64
static gecm_ctx *g_ctx[MAX_THREADS];
static void (*g_ref_fcn[4])(void *);
static int g_registered;
static unsigned long g_calls[4];

#define CHECK(call)                                                          \
    do {                                                                     \
        if ((call) < 0) {                                                    \
            fprintf(stderr, "gecm L1 binding: %s\n", gecm_last_error());     \
            exit(3);                                                         \
        }                                                                    \
    } while (0)

static void report(void)
{
    char name[128] = "?";
    if (g_ctx[0]) gecm_device_name(g_ctx[0], name, sizeof name);
    fprintf(stderr, "gecm L1 binding: %lu curve uploads, %lu stage-1 ranges, %lu stage-2 inits, %lu stage-2 ranges served on %s (%s)\n",
            g_calls[0], g_calls[1], g_calls[2], g_calls[3], name, gecm_version());
}

static gecm_ctx *ctx_of(thread_data_t *t, uint32_t tid)
{
    if (tid >= MAX_THREADS) { fprintf(stderr, "gecm L1 binding: more than %d threads\n", MAX_THREADS); exit(3); }
    if (!g_ctx[tid]) {
        mpz_t n;
        mpz_init(n);
        extract_bignum_from_vec_to_mpz(n, t->mdata->n, 0, NWORDS);
        char *s = mpz_get_str(NULL, 10, n);
        int devs = gecm_device_count();
        if (devs < 1) { fprintf(stderr, "gecm L1 binding: no HIP device\n"); exit(3); }
        CHECK(gecm_create(&g_ctx[tid], (int)(tid % (uint32_t)devs), s, DIGITBITS));
        gecm_config cfg;
        gecm_get_config(g_ctx[tid], &cfg);
        if (cfg.nwords != (int)NWORDS) { fprintf(stderr, "gecm L1 binding: NWORDS mismatch %d vs %u\n", cfg.nwords, NWORDS); exit(3); }
        free(s);
        mpz_clear(n);
        if (tid == 0) atexit(report);
    }
    return g_ctx[tid];
}

/* The reference's special-form mode (mdata->isMersenne != 0: mdata->n is 2^k -/+ 1 or 2^k - c, main.c:642-684) keeps PLAIN
 * residues in its vectors — ecm.c:1763-1772 skips the conversion to Montgomery form — where the ABI's vector operands
 * are x * 2^MAXBITS mod n.  to_abi / from_abi multiply a whole vector by 2^MAXBITS or its inverse modulo mdata->n. */
static void scale_vec(thread_data_t *t, bignum *v, int to_montgomery)
{
    mpz_t n, x, r;
    mpz_inits(n, x, r, NULL);
    extract_bignum_from_vec_to_mpz(n, t->mdata->n, 0, NWORDS);
    mpz_set_ui(r, 1);
    mpz_mul_2exp(r, r, MAXBITS);
    mpz_mod(r, r, n);
    if (!to_montgomery) mpz_invert(r, r, n);
    for (int k = 0; k < VECLEN; k++) {
        extract_bignum_from_vec_to_mpz(x, v, k, NWORDS);
        mpz_mul(x, x, r);
        mpz_mod(x, x, n);
        for (uint32_t j = 0; j < NWORDS; j++) v->data[j * VECLEN + k] = 0;
        insert_mpz_to_vec(v, x, k);
    }
    mpz_clears(n, x, r, NULL);
}

/* phase 0: ecm_build_curve_work_fcn (ecm.c:201-246) stays on the host, as in the reference; its output goes up */
static void gpu_build_curve(void *vptr)
{
    tpool_t *tp = (tpool_t *)vptr;
    thread_data_t *ud = (thread_data_t *)tp->user_data;
    const uint32_t tid = (uint32_t)tp->tindex;
    g_ref_fcn[0](vptr);
    const int plain = ud[tid].mdata->isMersenne != 0;
    if (plain) { scale_vec(&ud[tid], ud[tid].P->X, 1); scale_vec(&ud[tid], ud[tid].P->Z, 1); scale_vec(&ud[tid], ud[tid].work->s, 1); }
    CHECK(gecm_upload_points(ctx_of(&ud[tid], tid), ud[tid].P->X->data, ud[tid].P->Z->data, ud[tid].work->s->data, VECLEN));
    if (plain) { scale_vec(&ud[tid], ud[tid].P->X, 0); scale_vec(&ud[tid], ud[tid].P->Z, 0); scale_vec(&ud[tid], ud[tid].work->s, 0); }
    __sync_fetch_and_add(&g_calls[0], 1);
}

/* phase 1: ecm_stage1_work_fcn -> ecm_stage1 (ecm.c:167-176, 1806-1854) on the PRIMES list vececm has set up */
static void gpu_stage1(void *vptr)
{
    tpool_t *tp = (tpool_t *)vptr;
    thread_data_t *ud = (thread_data_t *)tp->user_data;
    const uint32_t tid = (uint32_t)tp->tindex;
    gecm_ctx *c = ctx_of(&ud[tid], tid);
    const uint32_t range = P_MIN <= 2 ? 0u : (uint32_t)(P_MIN / 100000000ULL);       /* ecm.c:1215: rangemin */
    CHECK(gecm_stage1_range(c, STAGE1_MAX, range));
    CHECK(gecm_sync(c));
    if (ud[tid].mdata->isMersenne != 0) CHECK(gecm_download_points_plain(c, ud[tid].P->X->data, ud[tid].P->Z->data));
    else CHECK(gecm_download_points(c, ud[tid].P->X->data, ud[tid].P->Z->data));
    gecm_stage1_stats st;
    gecm_get_stage1_stats(c, &st);
    ecm_work *w = ud[tid].work;
    /* what ecm_stage1 leaves behind for vececm: the counters (cumulative: cleared at curve build, ecm.c:1177-1178) and
     * last_pid, the index of the first prime it did not process (ecm.c:1824, 1844) */
    uint64_t i = 1;
    while (i < NUM_P && PRIMES[i] < STAGE1_MAX) i++;
    w->last_pid = (uint32_t)i;
    w->ptadds = (uint32_t)st.ptadds;
    w->ptdups = (uint32_t)st.ptdups;
    if (tid == 0) {                                                                   /* ecm.c:1847-1852 */
        printf("\nStage 1 completed at prime %lu with %u point-adds and %u point-doubles\n", PRIMES[i - 1], w->ptadds, w->ptdups);
        fflush(stdout);
    }
    __sync_fetch_and_add(&g_calls[1], 1);
}

/* the accumulator as the reference's scan reads it (ecm.c:1489): where an inversion met a non-invertible product
 * the reference holds the gcd itself in that lane (ecm.c:1927-1939) */
static void fetch_acc(gecm_ctx *c, ecm_work *w)
{
    CHECK(gecm_download_acc(c, w->stg2acc->data));       /* Montgomery form also in special-form mode: the reference only
                                                            takes its gcd with the input number, and 2^MAXBITS is a unit */
    for (int k = 0; k < VECLEN; k++) {
        char dec[2048];
        if (gecm_stage2_factor(c, (size_t)k, dec, sizeof dec, NULL) == 1) {
            mpz_t g;
            mpz_init_set_str(g, dec, 10);
            for (uint32_t j = 0; j < NWORDS; j++) w->stg2acc->data[j * VECLEN + k] = 0;
            insert_mpz_to_vec(w->stg2acc, g, k);
            mpz_clear(g);
        }
    }
}

static void put_stage2_counters(gecm_ctx *c, ecm_work *w)
{
    gecm_stage2_stats s2;
    if (gecm_get_stage2_stats(c, &s2) == 0) {
        w->ptadds = (uint32_t)s2.ptadds;
        w->numinv = (uint32_t)s2.numinv;
        w->paired = (uint32_t)s2.paired;
    }
}

/* phase 2: ecm_stage2_init_work_fcn -> ecm_stage2_init (ecm.c:178-186, 2201-2340) */
static void gpu_stage2_init(void *vptr)
{
    tpool_t *tp = (tpool_t *)vptr;
    thread_data_t *ud = (thread_data_t *)tp->user_data;
    const uint32_t tid = (uint32_t)tp->tindex;
    gecm_ctx *c = ctx_of(&ud[tid], tid);
    ecm_work *w = ud[tid].work;
    w->amin = (uint32_t)((STAGE1_MAX + w->D) / (2 * w->D));                           /* ecm.c:2208 */
    w->paired = 0; w->numprimes = 0; w->ptadds = 0; w->ptdups = 0; w->numinv = 0;     /* ecm.c:2228-2232 */
    if (tid == 0) printf("\n");
    CHECK(gecm_stage2_init(c, w->D, w->U));
    CHECK(gecm_sync(c));
    put_stage2_counters(c, w);
    fetch_acc(c, w);
    __sync_fetch_and_add(&g_calls[2], 1);
}

/* phase 3: ecm_stage2_work_fcn -> ecm_stage2_pair (ecm.c:188-199, 2342-2540) with the reference's own pair map */
static void gpu_stage2_pair(void *vptr)
{
    tpool_t *tp = (tpool_t *)vptr;
    thread_data_t *ud = (thread_data_t *)tp->user_data;
    const uint32_t tid = (uint32_t)tp->tindex;
    gecm_ctx *c = ctx_of(&ud[tid], tid);
    ecm_work *w = ud[tid].work;
    if (tid == 0)                                                                     /* ecm.c:2372, 2440-2445 */
        printf("\ncommencing stage 2 at A=%lu\nw = %u, R = %u, L = %u, U = %d, umax = %u, amin = %u\n",
               2 * (uint64_t)w->amin * (uint64_t)w->D, w->D, w->R - 3, w->L, (int)w->U, w->U * w->D, w->amin);
    CHECK(gecm_stage2_pair(c, ud[tid].pairmap_steps, ud[tid].pairmap_v, ud[tid].pairmap_u, w->amin));
    CHECK(gecm_sync(c));
    gecm_stage2_stats s2;
    gecm_get_stage2_stats(c, &s2);
    put_stage2_counters(c, w);
    w->amin = s2.amin_last;                                                           /* ecm.c:2535-2537 */
    w->last_pid = (uint32_t)NUM_P;
    fetch_acc(c, w);
    __sync_fetch_and_add(&g_calls[3], 1);
}

/* what ecm.c:1130-1133 calls instead of tpool_add_work_fcn */
void gecm_bind_add_work_fcn(tpool_t *tdata, void *work_fcn)
{
    static void (*const gpu[4])(void *) = {gpu_build_curve, gpu_stage1, gpu_stage2_init, gpu_stage2_pair};
    if (g_registered < 4) {
        g_ref_fcn[g_registered] = (void (*)(void *))work_fcn;
        tpool_add_work_fcn(tdata, (void *)gpu[g_registered]);
        g_registered++;
    } else
        tpool_add_work_fcn(tdata, work_fcn);
}
