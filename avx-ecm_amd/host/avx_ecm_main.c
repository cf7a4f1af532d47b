/* avx_ecm_main.c — the avx-ecm command line on top of libgecm.
 *
 *     avx-ecm input curves B1 [threads] [B2] [sigma]
 *
 * Same positional arguments as the reference (main.c:380-384, 459-460, 537-559).  The 4th argument keeps the
 * reference's meaning for everything an existing script can observe: it only enters the rounding of the curve
 * count (main.c:585-589: curves per thread, whole 8-lane vectors per thread) and the banner.  It does NOT select
 * GPUs: the run uses every visible HIP device (or the first GECM_GPUS of them), one host thread and one gecm_ctx
 * per GPU, and the result files do not depend on how many there are.  One deliberate difference: with a fixed
 * sigma and more than one thread the reference gives every thread the SAME eight sigmas per step (ecm.c:1187 adds
 * the step to thread 0's sigmas for all threads), i.e. it runs every curve `threads` times; here curve k of the run
 * gets sigma + k, which for threads = 1 is exactly the reference's assignment.
 * Reproduces vececm's sequence (ecm.c:1077-1544) and its
 * observable protocol: the banner lines, "Stage 1 completed at prime ..." counters, the factor
 * lines on stdout and in ecm_results.txt (ecm.c:1356-1367, 1510-1522) and the GMP-ECM resume
 * lines appended to save_b1.txt (ecm.c:1372-1380), in global curve order (sigma ascending), so the
 * file is the same for any number of GPUs.  GMP-ECM then resumes with
 *     ecm -resume save_b1.txt <B1> <B2>
 * Defaults as the reference: B2 = 100*B1 (main.c:462), B2 <= B1 disables stage 2
 * (main.c:548-552), sigma = 0 draws random 64-bit sigmas >= 6 (ecm.c:1564-1570).
 * Curves are processed in batches (here up to 131072 per GPU, in the reference 8 per thread); the run
 * stops after the batch in which a factor is found (ecm.c:1531-1532).  The curve count is rounded
 * up to a multiple of 8 as the reference does (main.c:585-589).
 */
#include "../../include/gecm.h"
#include "calc_lite.h"
#include "gecm_pair.h"
#include "mpl.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <unistd.h>

#define MAX_GPUS 16
/* curves per GPU per pass: 2 wavefronts on each of the 1024 SIMDs of an MI355X */
#define FULL_BATCH 131072u

static double now(void)
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return (double)tv.tv_sec + 1e-6 * (double)tv.tv_usec;
}

/* lcg_rand, main.c:993-998 */
static uint64_t lcg_rand(uint64_t *state)
{
    *state = 6364136223846793005ULL * (*state) + 1442695040888963407ULL;
    return *state;
}

/* the pair map of the first prime range of stage 2 (ecm.c:1441-1443): identical for every batch, GPU and pass, so it
 * is made once, on the main thread, while the GPUs run the first stage 1 */
typedef struct {
    uint64_t lo, hi;
    uint32_t D, U;
    gecm_pairs pm;
    int valid, rc;
    int settled;               /* made or failed: the job threads wait for this before they prepare their tapes */
    pthread_mutex_t mu;
    pthread_cond_t cv;
} first_range_t;


typedef struct {
    int gpu;
    gecm_ctx *ctx;
    uint64_t *sigma;
    size_t ncurves, first;     /* this GPU's slice of the batch: global indices first .. first+ncurves */
    uint64_t B1, B2;
    int do_stage2;
    int rc;
    char err[512];
    double t_build, t_stage1, t_s2;
    const gecm_pairs *pm;      /* the pair map of the current prime range (made once, shared) */
    first_range_t *fr;         /* stage 1: the first range's map, being made on the main thread meanwhile (or NULL) */
} job_t;

static void *job_build(void *p)
{
    job_t *j = (job_t *)p;
    double t = now();
    j->rc = j->ncurves ? gecm_build_curves(j->ctx, j->sigma, j->ncurves) : 0;
    if (j->rc < 0) snprintf(j->err, sizeof j->err, "%s", gecm_last_error());
    j->t_build = now() - t;
    return NULL;
}

static void *job_stage1(void *p)
{
    job_t *j = (job_t *)p;
    double t = now();
    j->rc = 0;
    if (j->ncurves) {
        j->rc = gecm_stage1(j->ctx, j->B1);                           /* returns after the launch */
        if (j->rc == 0 && j->fr) {
            /* while the kernel runs: this context's launch tape of the first stage-2 range, as soon as the main
             * thread has the pair map (0.13 s of host time; later passes find it kept) */
            first_range_t *fr = j->fr;
            pthread_mutex_lock(&fr->mu);
            while (!fr->settled) pthread_cond_wait(&fr->cv, &fr->mu);
            pthread_mutex_unlock(&fr->mu);
            if (fr->valid)
                (void)gecm_stage2_pair_prepare(j->ctx, fr->D, fr->U, fr->pm.steps, fr->pm.pairmap_v, fr->pm.pairmap_u, fr->pm.amin);
        }
        if (j->rc == 0) j->rc = gecm_sync(j->ctx);
        if (j->rc < 0) snprintf(j->err, sizeof j->err, "%s", gecm_last_error());
    }
    j->t_stage1 = now() - t;
    return NULL;
}

static void *job_stage2(void *p)
{
    job_t *j = (job_t *)p;
    double t = now();
    j->rc = 0;
    if (j->ncurves) {
        j->rc = gecm_stage2_init(j->ctx, 0, 0);                       /* ecm.c:1401-1407 */
        if (j->rc == 0) j->rc = gecm_sync(j->ctx);
        if (j->rc < 0) snprintf(j->err, sizeof j->err, "%s", gecm_last_error());
    }
    j->t_s2 = now() - t;
    return NULL;
}

static void *job_stage2_pair(void *p)
{
    job_t *j = (job_t *)p;
    j->rc = 0;
    if (j->ncurves) {
        j->rc = gecm_stage2_pair(j->ctx, j->pm->steps, j->pm->pairmap_v, j->pm->pairmap_u, j->pm->amin);   /* ecm.c:1460 */
        if (j->rc == 0) j->rc = gecm_sync(j->ctx);
        if (j->rc < 0) snprintf(j->err, sizeof j->err, "%s", gecm_last_error());
    }
    return NULL;
}

static void make_first_range(first_range_t *fr)
{
    if (!fr->valid) {
        fr->rc = gecm_pair_primes(&fr->pm, fr->lo, fr->hi, fr->D, fr->U);
        fr->valid = fr->rc == 0;
    }
    pthread_mutex_lock(&fr->mu);
    fr->settled = 1;
    pthread_cond_broadcast(&fr->cv);
    pthread_mutex_unlock(&fr->mu);
}

static int run_all(job_t *jobs, int n, void *(*fn)(void *), first_range_t *meanwhile)
{
    pthread_t th[MAX_GPUS];
    if (meanwhile) {                         /* every job on a thread of its own, the host work here */
        for (int i = 0; i < n; i++) jobs[i].fr = meanwhile;
        for (int i = 0; i < n; i++) pthread_create(&th[i], NULL, fn, &jobs[i]);
        make_first_range(meanwhile);
        for (int i = 0; i < n; i++) pthread_join(th[i], NULL);
    } else {
        for (int i = 0; i < n; i++) jobs[i].fr = NULL;
        for (int i = 1; i < n; i++) pthread_create(&th[i], NULL, fn, &jobs[i]);
        fn(&jobs[0]);
        for (int i = 1; i < n; i++) pthread_join(th[i], NULL);
    }
    for (int i = 0; i < n; i++)
        if (jobs[i].rc < 0) {
            fprintf(stderr, "GPU %d: %s\n", jobs[i].gpu, jobs[i].err);
            return -1;
        }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 4) {
        printf("usage: avx-ecm $input $numcurves $B1 [$threads] [$B2] [$sigma]\n");   /* main.c:382 */
        return 1;
    }
    double t_start = now();
    printf("starting process %d\n", (int)getpid());                               /* main.c:391 */
    /* main.c:393-457: evaluate the expression, recognise Cunningham-type inputs, strip algebraic factors */
    static char ndec[MPL_MAXL * 10 + 16], prep_log[65536];
    gecm_input_info inf;
    if (gecm_prepare_input(argv[1], 52, ndec, sizeof ndec, &inf, prep_log, sizeof prep_log)) {
        fputs(prep_log, stdout);
        printf("input must evaluate to an odd integer >= 3 (operators + - * / %% ^ ! # fib() luc())\n");
        return 1;
    }
    size_t numcurves = strtoul(argv[2], NULL, 10);
    uint64_t B1 = strtoull(argv[3], NULL, 10);
    uint64_t B2 = 100ULL * B1;                                                    /* main.c:462 */
    int threads = 1, do_stage2 = 1;
    uint64_t sigma0 = 0;
    if (argc > 4) threads = atoi(argv[4]);
    if (argc > 5) B2 = strtoull(argv[5], NULL, 10);
    if (argc > 6) sigma0 = strtoull(argv[6], NULL, 10);
    if (B2 <= B1) { do_stage2 = 0; B2 = B1; }                                      /* main.c:548-552 */
    if (threads < 1) threads = 1;
    int have = gecm_device_count();
    if (have < 1) { fprintf(stderr, "no HIP device visible\n"); return 2; }
    int gpus = have;
    if (getenv("GECM_GPUS") && atoi(getenv("GECM_GPUS")) > 0 && atoi(getenv("GECM_GPUS")) < gpus) gpus = atoi(getenv("GECM_GPUS"));
    /* GECM_CONTEXTS_PER_GPU=k (rehearsal knob): k contexts, each with its host thread, on every device used — the
     * multi-context path of this driver on a box with one GPU.  Nothing is gained by it. */
    int per_gpu = 1;
    if (getenv("GECM_CONTEXTS_PER_GPU") && atoi(getenv("GECM_CONTEXTS_PER_GPU")) > 1) per_gpu = atoi(getenv("GECM_CONTEXTS_PER_GPU"));
    const int devices = gpus;
    gpus *= per_gpu;
    if (gpus > MAX_GPUS) gpus = MAX_GPUS;
    if (numcurves == 0 || B1 < 2 || B1 > 100000000ULL) { printf("need curves >= 1 and 2 <= B1 <= 1e8\n"); return 1; }
    /* main.c:585-589: at least one curve per thread, the same number on every thread; ecm.c:1151: every thread
     * runs whole vectors of VECLEN = 8, so "10 curves" on one thread writes 16 resume lines there and here */
    if (numcurves < (size_t)threads) numcurves = (size_t)threads;
    const size_t per_thread = numcurves / (size_t)threads + (numcurves % (size_t)threads != 0);
    numcurves = (per_thread + 7) / 8 * 8 * (size_t)threads;

    fputs(prep_log, stdout);          /* "gen: ...", "removing algebraic ...", "commencing parallel ecm on ..." */
    job_t jobs[MAX_GPUS];
    memset(jobs, 0, sizeof jobs);
    for (int g = 0; g < gpus; g++) {
        jobs[g].gpu = g;
        if (gecm_create(&jobs[g].ctx, g % devices, ndec, 52)) { fprintf(stderr, "%s\n", gecm_last_error()); return 2; }
    }
    gecm_config cfg;
    gecm_get_config(jobs[0].ctx, &cfg);
    char devname[256];
    gecm_device_name(jobs[0].ctx, devname, sizeof devname);
    /* main.c:529-533, verbatim: DIGITBITS and VECLEN describe the vector format at the boundary (curves come in
     * groups of 8, limbs of 52 bits); the device's own numbers follow on a line of their own */
    printf("ECM has been configured with DIGITBITS = %d, VECLEN = %d, GMP_LIMB_BITS = %d\n", cfg.digitbits, 8, 64);
    printf("Choosing MAXBITS = %d, NWORDS = %d, NBLOCKS = %d based on input size %d\n", cfg.maxbits, cfg.nwords,
           cfg.nwords / 4, cfg.nbits);
    printf("%s: %d GPU(s) [%s], residues of %d limbs x 28 bits on the device\n", gecm_version(), gpus, devname,
           cfg.dev_limbs);
    if (inf.ref_special_reduction) {
        /* main.c:644-670 prints "Using special Mersenne mod for factor of: 2^k-1" here */
        int fk = 0, fl = 0;
        if (gecm_get_special_form(jobs[0].ctx, &fk, &fl) >= 1)
            printf("REDC modulo 2^%d%c1 (%d limbs, special reduction) serves stage 1 of this factor of 2^%d%c1 when the "
                   "batch is large enough for it; residues are reduced modulo N\n", abs(fk), fk > 0 ? '-' : '+', fl,
                   abs(fk), fk > 0 ? '-' : '+');
        else
            printf("Input divides 2^%d %c %d: running REDC on the %d-bit cofactor (residues = the reference's modulo N)\n",
                   inf.k, inf.form > 0 ? '-' : '+', inf.form, inf.nbits);
    }
    if (argc > 6) printf("starting with sigma = %lu\n", (unsigned long)sigma0);   /* main.c:558 */
    size_t per_pass = (size_t)FULL_BATCH * (size_t)gpus;
    printf("Input has %d bits, using %d threads (%d curves/thread)\n", cfg.nbits, threads, (int)per_thread);   /* main.c:591-592 */
    printf("Processing in batches of %u primes\n", 100000000u);                  /* main.c:593 */
    printf("Initialization took %1.4f seconds.\n", now() - t_start);              /* main.c:776 */

    first_range_t first_range;
    memset(&first_range, 0, sizeof first_range);
    pthread_mutex_init(&first_range.mu, NULL);
    pthread_cond_init(&first_range.cv, NULL);
    first_range.lo = B1;
    first_range.hi = B1 + 100000000ULL < B2 ? B1 + 100000000ULL : B2;
    first_range.D = gecm_s2_default_D(B1);
    first_range.U = GECM_S2_DEFAULT_U;
    uint64_t lcg = (uint64_t)(t_start * 1e6) * 0x9E3779B97F4A7C15ULL + (uint64_t)getpid();
    int found = 0;
    static char line[16384], fac[4096];
    for (size_t done = 0; done < numcurves && !found; done += per_pass) {
        size_t batch = numcurves - done < per_pass ? numcurves - done : per_pass;
        printf("\nCommencing curves %zu-%zu of %zu\n", done, done + batch - 1, numcurves);   /* ecm.c:1201 */
        /* host-side split: GPU g owns global indices [batch*g/G, batch*(g+1)/G) of this pass */
        uint64_t *sig = (uint64_t *)malloc(batch * sizeof(uint64_t));
        for (size_t k = 0; k < batch; k++) {
            if (sigma0) sig[k] = sigma0 + done + k;                              /* main.c:761, ecm.c:1187 */
            else do { sig[k] = lcg_rand(&lcg); } while (sig[k] < 6);             /* ecm.c:1564-1570 */
        }
        for (int g = 0; g < gpus; g++) {
            size_t lo = batch * (size_t)g / (size_t)gpus, hi = batch * (size_t)(g + 1) / (size_t)gpus;
            jobs[g].first = lo;
            jobs[g].ncurves = hi - lo;
            jobs[g].sigma = sig + lo;
            jobs[g].B1 = B1;
            jobs[g].B2 = B2;
            jobs[g].do_stage2 = do_stage2;
        }
        double t = now();
        if (run_all(jobs, gpus, job_build, NULL)) return 2;
        printf("Building curves took %1.4f seconds.\n", now() - t);              /* ecm.c:1204 */
        printf("Commencing Stage 1 @ prime 2\n");                                /* ecm.c:1233 */
        t = now();
        if (run_all(jobs, gpus, job_stage1, do_stage2 ? &first_range : NULL)) return 2;
        gecm_stage1_stats st;
        gecm_get_stage1_stats(jobs[0].ctx, &st);
        printf("\nStage 1 completed at prime %lu with %lu point-adds and %lu point-doubles\n",
               (unsigned long)st.last_prime, (unsigned long)st.ptadds, (unsigned long)st.ptdups);   /* ecm.c:1849 */
        double t1 = now() - t;
        printf("Stage 1 took %1.4f seconds\n", t1);                              /* ecm.c:1317 */
        printf("(%.1f curves/sec; kernel %.1f ms on GPU 0)\n", (double)batch / t1, gecm_last_kernel_ms(jobs[0].ctx));
        /* save + factor scan in global curve order, ecm.c:1319-1388 */
        FILE *save = fopen("save_b1.txt", "a");
        if (!save) printf("could not open save_b1.txt for appending, Stage 1 data will not be saved\n");
        for (int g = 0; g < gpus; g++) {
            /* whole-batch gcd scan on the device, then format only the flagged curves */
            if (jobs[g].ncurves && gecm_scan_factors(jobs[g].ctx, 1, NULL) < 0) { fprintf(stderr, "%s\n", gecm_last_error()); return 2; }
            for (size_t k = 0; k < jobs[g].ncurves; k++) {
                int prp = 0;
                int r = gecm_curve_flag(jobs[g].ctx, 1, k) ? gecm_stage1_factor(jobs[g].ctx, k, fac, sizeof fac, &prp) : 0;
                if (r == 1) {
                    size_t curve = done + jobs[g].first + k;
                    printf("\nfound %s%d factor %s in stage 1 (B1 = %lu): thread %d, vec %zu, sigma %lu\n",
                           prp ? "PRP" : "C", gecm_sizeinbase10(fac), fac, (unsigned long)B1, g, k,
                           (unsigned long)jobs[g].sigma[k]);
                    FILE *out = fopen("ecm_results.txt", "a");
                    if (out) {
                        fprintf(out, "\nfound %s%d factor %s in stage 1 (B1 = %lu): curve %zu, thread %d, vec %zu, sigma %lu\n",
                                prp ? "PRP" : "C", gecm_sizeinbase10(fac), fac, (unsigned long)B1, curve, g, k,
                                (unsigned long)jobs[g].sigma[k]);
                        fclose(out);
                    }
                    found = 1;
                }
                if (save && gecm_format_save_line(jobs[g].ctx, k, line, sizeof line) > 0) fputs(line, save);
            }
        }
        if (save) fclose(save);
        fflush(stdout);
        if (do_stage2) {                                                         /* ecm.c:1394-1528 */
            t = now();
            if (run_all(jobs, gpus, job_stage2, NULL)) return 2;                 /* stage-2 init, ecm.c:1401-1421 */
            printf("Stage 2 Init took %1.4f seconds\n", now() - t);              /* ecm.c:1421 */
            gecm_stage2_stats s2;
            gecm_get_stage2_stats(jobs[0].ctx, &s2);
            uint32_t rcount = 0;
            for (uint32_t i = 0; i < 2 * s2.D; i++) {                            /* main.c:874-882: R - 3 */
                uint32_t a = i, b = 2 * s2.D;
                while (b) { uint32_t r = a % b; a = b; b = r; }
                rcount += a == 1;
            }
            for (uint64_t p = B1; p < B2; p += 100000000ULL) {                   /* ecm.c:1424-1476 */
                const uint64_t hi = p + 100000000ULL < B2 ? p + 100000000ULL : B2;
                gecm_pairs pm;
                const int shared = first_range.valid && p == first_range.lo && hi == first_range.hi &&
                                   s2.D == first_range.D && s2.U == first_range.U;
                printf("commencing pair on range %lu:%lu\n", (unsigned long)p, (unsigned long)hi);   /* ecm.c:2568 */
                if (shared) pm = first_range.pm;
                else if (gecm_pair_primes(&pm, p, hi, s2.D, s2.U)) { fprintf(stderr, "%s\n", gecm_last_error()); return 2; }
                printf("%u pairs found from %u primes (ratio = %1.2f)\n", pm.pairs, pm.primes,
                       pm.primes ? (double)pm.pairs / (double)pm.primes : 0.0);   /* ecm.c:2904-2905 */
                printf("\ncommencing stage 2 at A=%lu\nw = %u, R = %u, L = %u, U = %d, umax = %u, amin = %u\n",
                       2ul * (unsigned long)pm.amin * s2.D, s2.D, rcount, s2.L, (int)s2.U, s2.U * s2.D, pm.amin);   /* ecm.c:2440-2442 */
                for (int g = 0; g < gpus; g++) jobs[g].pm = &pm;
                if (run_all(jobs, gpus, job_stage2_pair, NULL)) return 2;
                if (!shared) gecm_pairmap_release(&pm);
                gecm_get_stage2_stats(jobs[0].ctx, &s2);
                printf("\nlast amin: %u\n", s2.amin_last);                       /* ecm.c:1462 */
            }
            printf("\nStage 2 took %1.4f seconds\n", now() - t);                 /* ecm.c:1481 */
            printf("performed %lu pt-adds, %lu inversions, and %lu pair-muls in stage 2\n",
                   (unsigned long)s2.ptadds, (unsigned long)s2.numinv, (unsigned long)s2.paired);   /* ecm.c:1482 */
            for (int g = 0; g < gpus; g++) {
                if (jobs[g].ncurves && gecm_scan_factors(jobs[g].ctx, 2, NULL) < 0) { fprintf(stderr, "%s\n", gecm_last_error()); return 2; }
                for (size_t k = 0; k < jobs[g].ncurves; k++) {
                    int prp = 0;
                    if (gecm_curve_flag(jobs[g].ctx, 2, k) && gecm_stage2_factor(jobs[g].ctx, k, fac, sizeof fac, &prp) == 1) {
                        size_t curve = done + jobs[g].first + k;
                        printf("\nfound %s%d factor %s in stage 2 (B2 = %lu): thread %d, vec %zu, sigma %lu\n",
                               prp ? "PRP" : "C", gecm_sizeinbase10(fac), fac, (unsigned long)B2, g, k,
                               (unsigned long)jobs[g].sigma[k]);
                        FILE *out = fopen("ecm_results.txt", "a");
                        if (out) {
                            fprintf(out, "\nfound %s%d factor %s in stage 2 (B2 = %lu): curve %zu, thread %d, vec %zu, sigma %lu\n",
                                    prp ? "PRP" : "C", gecm_sizeinbase10(fac), fac, (unsigned long)B2, curve, g, k,
                                    (unsigned long)jobs[g].sigma[k]);
                            fclose(out);
                        }
                        found = 1;
                    }
                }
            }
        }
        free(sig);
    }
    if (first_range.valid) gecm_pairmap_release(&first_range.pm);
    for (int g = 0; g < gpus; g++) gecm_destroy(jobs[g].ctx);
    printf("Process took %1.4f seconds.\n", now() - t_start);                    /* ecm.c:1538 */
    return 0;
}
