/* avx_ecm_main.c — the avx-ecm command line on top of libgecm.
 *
 *     avx-ecm input curves B1 [threads] [B2] [sigma]
 *
 * Same positional arguments as the reference (main.c:380-384, 459-460, 537-559) and the same FILES, byte for byte:
 * save_b1.txt (GMP-ECM resume lines, ecm.c:1372-1380), ecm_results.txt (factor lines, ecm.c:1356-1367, 1510-1522)
 * and, for B1 above one prime range of 1e8, checkpoint.txt (ecm.c:1236-1312).  GMP-ECM then resumes with
 *     ecm -resume save_b1.txt <B1> <B2>
 *
 * How the reference's run maps onto the GPU.  vececm (ecm.c:1077-1544) works in BATCHES of 8 curves per thread:
 * for every batch it builds the curves, runs stage 1 (one ecm_stage1 call per prime range), appends the batch to
 * save_b1.txt, runs stage 2, and stops after the first batch in which any curve found a factor (ecm.c:1531-1532).
 * Line j*8 + i of a batch belongs to thread j, vector lane i; its label in ecm_results.txt is
 * "curve threads*curve + j*8 + i, thread j, vec i" (ecm.c:1356-1366).  With a sigma on the command line every thread
 * of a batch runs the SAME eight sigmas sigma + curve + i (ecm.c:1187 copies thread 0's): the batch has 8 distinct
 * curves and 8*threads lines.  Without one every lane draws its own (ecm.c:1564-1570).
 *
 * Here a PASS puts many reference batches on the GPU(s) at once — up to 131072 distinct curves per GPU — and then
 * writes exactly what the reference would have written for those batches one after the other: the lines of every
 * batch up to and including the first one with a factor, that batch's factor lines, nothing after it.  With a fixed
 * sigma the 8 distinct curves of a batch are computed once and written `threads` times.  The 4th argument therefore
 * keeps everything a script can observe (curve-count rounding main.c:585-589, banner, labels, line count); it does not
 * select GPUs: the run uses every visible HIP device (or the first GECM_GPUS), one host thread and one gecm_ctx per
 * GPU, and the files do not depend on how many there are.
 *
 * Passes are pipelined when there are several (and B1 is within one prime range): two sets of contexts alternate, so
 * that the curve construction of pass k+1 and the formatting and writing of pass k-1 (host work) run while the
 * kernels of pass k do; files are written in pass order.  stdout carries the reference's lines per PASS, not per
 * batch (its timings are per pass too).
 *
 * Defaults as the reference: B2 = 100*B1 (main.c:462), B2 <= B1 disables stage 2 (main.c:548-552), no sigma = random
 * 64-bit sigmas >= 6.
 */
#include "../../include/gecm.h"
#include "calc_lite.h"
#include "gecm_pair.h"
#include "mpl.h"
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <unistd.h>

#define MAX_GPUS 16
/* The reference is built in two flavours (avx_ecm.h:65-93): DIGITBITS = 52 with VECLEN = 8 curves per vector, and
 * DIGITBITS = 32 with VECLEN = 16.  The flavour decides the NWORDS/MAXBITS rule and its banner lines, the size of a batch
 * (VECLEN x threads lines) and therefore where a run stops, never a residue.  This source builds both: avx-ecm and
 * avx-ecm-32 (Makefile: -DGECM_CLI_DIGITBITS=32). */
#ifndef GECM_CLI_DIGITBITS
#define GECM_CLI_DIGITBITS 52
#endif
#define VECLEN (GECM_CLI_DIGITBITS == 52 ? 8 : 16)
/* distinct curves per GPU per pass: 2 wavefronts on each of the 1024 SIMDs of an MI355X */
#define FULL_BATCH 131072u
#define PRIME_RANGE 100000000ULL

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
 * is made once, on the thread of the first pass, while the GPUs run the first stage 1 */
typedef struct {
    uint64_t lo, hi;
    uint32_t D, U;
    gecm_pairs pm;
    int valid, rc;
    int settled;               /* made or failed: the job threads wait for this before they prepare their tapes */
    int claimed;               /* some pass has taken on making it */
    pthread_mutex_t mu;
    pthread_cond_t cv;
} first_range_t;

/* ---- run-wide state ------------------------------------------------------------------------- */
typedef struct {
    uint64_t B1, B2, sigma0;
    int do_stage2, threads, fixed_sigma;
    int gpus;                  /* contexts per slot */
    int nranges;               /* ecm_stage1 calls per batch (prime ranges of 1e8) */
    size_t per_thread;         /* tdata[0].curves */
    size_t nbatches;           /* reference batches of the run: ceil(per_thread / 8) */
    size_t ub;                 /* distinct curves per reference batch: 8 (fixed sigma) or 8*threads */
    first_range_t fr;
    gecm_stage1_range_desc *rd;   /* what vececm prints and decides around each prime range (made once, on a helper thread) */
    pthread_t rd_thread;
    int rd_pending, rd_rc;
    /* pipeline: passes take the GPUs and write their output in pass order */
    pthread_mutex_t mu;
    pthread_cond_t cv;
    size_t gpu_turn, out_turn;
    size_t found_pass;         /* the pass that found a factor (SIZE_MAX: none yet): later passes are not run, not written */
    int failed;
    uint64_t lcg;
    double t_start;
} run_t;

typedef struct {
    int gpu;
    gecm_ctx *ctx;
    uint64_t *sigma;
    size_t ncurves, first;     /* this GPU's slice of the pass: distinct curves first .. first+ncurves */
    uint64_t B1;
    uint32_t range;
    int rc;
    char err[512];
    const gecm_pairs *pm;      /* the pair map of the current prime range (made once, shared) */
    first_range_t *fr;         /* stage 1: the first range's map, being made on the pass thread meanwhile (or NULL) */
    double kernel_ms;
    int progress;              /* print the launches of a long stage 1 as they finish (GPU 0 of a pass that prints live) */
} job_t;

typedef struct {
    char *buf;
    size_t len, cap;
} text_t;

typedef struct {
    run_t *run;
    size_t index;              /* pass number */
    int slot;
    size_t b0, nb;             /* reference batches b0 .. b0+nb of the run */
    size_t ucurves;            /* distinct curves of the pass = nb * ub */
    uint64_t *sigma;
    job_t jobs[MAX_GPUS];
    text_t log;                /* this pass's stdout, released in pass order */
    int live;                  /* not pipelined: print as it happens */
    /* checkpoints (several prime ranges): per range written, its lines in (batch, thread, lane) order and the
     * factor lines of the first flagged batch */
    int nck;
    struct ck_t {
        uint64_t last_prime;
        char **lines;          /* ucurves resume lines (distinct curves) */
        size_t first_flagged;  /* batch (within the pass), or nb if none */
        text_t res, out;       /* ecm_results.txt lines / stdout lines of that batch */
    } *ck;
    long ck_offset;            /* where this pass's part of checkpoint.txt starts */
    pthread_t th;
    double t_build, t_stage1, t_s2init, t_s2;
} pass_t;

static void text_add(text_t *t, const char *s, size_t n)
{
    if (t->len + n + 1 > t->cap) {
        size_t cap = t->cap ? t->cap * 2 : 4096;
        while (cap < t->len + n + 1) cap *= 2;
        char *p = (char *)realloc(t->buf, cap);
        if (!p) { fprintf(stderr, "out of memory\n"); exit(2); }
        t->buf = p;
        t->cap = cap;
    }
    memcpy(t->buf + t->len, s, n);
    t->len += n;
    t->buf[t->len] = 0;
}

static void text_printf(text_t *t, const char *fmt, ...)
{
    char tmp[8192];
    va_list ap;
    va_start(ap, fmt);
    int n = vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    if (n > 0) text_add(t, tmp, (size_t)n < sizeof tmp ? (size_t)n : sizeof tmp - 1);
}

/* a line of this pass's stdout */
static void plog(pass_t *ps, const char *fmt, ...)
{
    char tmp[8192];
    va_list ap;
    va_start(ap, fmt);
    int n = vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    if (n <= 0) return;
    if (ps->live) { fputs(tmp, stdout); fflush(stdout); }
    else text_add(&ps->log, tmp, (size_t)n < sizeof tmp ? (size_t)n : sizeof tmp - 1);
}

/* the sieved interval, prime count, first and last prime and checkpoint decision of every stage-1 range: 0.25 s of
 * sieving per range, done once while the contexts are made and the first curves built */
static void *describe_ranges(void *arg)
{
    run_t *R = (run_t *)arg;
    for (int r = 0; r < R->nranges && !R->rd_rc; r++)
        R->rd_rc = gecm_stage1_describe_range(R->B1, R->B2, (uint32_t)r, &R->rd[r]);
    return NULL;
}

/* ---- per-GPU jobs of a pass ----------------------------------------------------------------- */
static void *job_build(void *p)
{
    job_t *j = (job_t *)p;
    j->rc = j->ncurves ? gecm_build_curves(j->ctx, j->sigma, j->ncurves) : 0;
    if (j->rc < 0) snprintf(j->err, sizeof j->err, "%s", gecm_last_error());
    return NULL;
}

static void *job_stage1(void *p)
{
    job_t *j = (job_t *)p;
    j->rc = 0;
    if (j->ncurves) {
        j->rc = gecm_stage1_range(j->ctx, j->B1, j->range);               /* returns after the launch */
        if (j->rc == 0 && j->fr) {
            /* while the kernel runs: this context's launch tape of the first stage-2 range, as soon as the pass
             * thread has the pair map (0.13 s of host time; later passes find it kept) */
            first_range_t *fr = j->fr;
            pthread_mutex_lock(&fr->mu);
            while (!fr->settled) pthread_cond_wait(&fr->cv, &fr->mu);
            pthread_mutex_unlock(&fr->mu);
            if (fr->valid)
                (void)gecm_stage2_pair_prepare(j->ctx, fr->D, fr->U, fr->pm.steps, fr->pm.pairmap_v, fr->pm.pairmap_u, fr->pm.amin);
        }
        if (j->rc == 0 && j->progress) {
            /* a 1e8 prime range is 13 launches of up to minutes each: say where it is (the reference prints
             * "accumulating prime" every 8192 primes, ecm.c:1834-1842) */
            uint32_t done = 0, total = 0, shown = 0;
            while (gecm_stage1_progress(j->ctx, &done, &total) == 0 && total > 1 && done < total) {
                if (done != shown) { printf("stage 1, range %u: launch %u of %u done\r", j->range, done, total); fflush(stdout); shown = done; }
                usleep(200000);
            }
        }
        if (j->rc == 0) j->rc = gecm_sync(j->ctx);
        if (j->rc == 0) j->kernel_ms += gecm_last_kernel_ms(j->ctx);
        /* the device factor scan and the download of x, z belong to the GPU's turn (the next pass's kernel would
         * keep them waiting); formatting happens later, off the GPU */
        if (j->rc == 0 && gecm_scan_factors(j->ctx, 1, NULL) < 0) j->rc = -1;
        if (j->rc < 0) snprintf(j->err, sizeof j->err, "%s", gecm_last_error());
    }
    return NULL;
}

static void *job_stage2_init(void *p)
{
    job_t *j = (job_t *)p;
    j->rc = 0;
    if (j->ncurves) {
        j->rc = gecm_stage2_init(j->ctx, 0, 0);                           /* ecm.c:1401-1407 */
        if (j->rc == 0) j->rc = gecm_sync(j->ctx);
        if (j->rc < 0) snprintf(j->err, sizeof j->err, "%s", gecm_last_error());
    }
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

static void *job_stage2_scan(void *p)
{
    job_t *j = (job_t *)p;
    j->rc = 0;
    if (j->ncurves && gecm_scan_factors(j->ctx, 2, NULL) < 0) {
        j->rc = -1;
        snprintf(j->err, sizeof j->err, "%s", gecm_last_error());
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

static int run_all(job_t *jobs, int n, void *(*fn)(void *), first_range_t *meanwhile, int make_it)
{
    pthread_t th[MAX_GPUS];
    for (int i = 0; i < n; i++) jobs[i].fr = meanwhile;
    if (meanwhile && make_it) {              /* every job on a thread of its own, the host work here */
        for (int i = 0; i < n; i++) pthread_create(&th[i], NULL, fn, &jobs[i]);
        make_first_range(meanwhile);
        for (int i = 0; i < n; i++) pthread_join(th[i], NULL);
    } else {
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

/* ---- where a line of the reference's files comes from ------------------------------------------
 * Line `l` (0 .. 8*threads) of batch b of a pass is distinct curve b*ub + (fixed sigma ? l % 8 : l) of the pass:
 * which job holds it, and at which index. */
static void locate(const pass_t *ps, size_t b, size_t l, int *g, size_t *k)
{
    const run_t *R = ps->run;
    const size_t u = b * R->ub + (R->fixed_sigma ? l % VECLEN : l);
    for (int i = 0; i < R->gpus; i++)
        if (u >= ps->jobs[i].first && u < ps->jobs[i].first + ps->jobs[i].ncurves) { *g = i; *k = u - ps->jobs[i].first; return; }
    *g = 0; *k = 0;
}

static int batch_flagged(const pass_t *ps, int stage, size_t b)
{
    const run_t *R = ps->run;
    for (size_t u = b * R->ub; u < (b + 1) * R->ub; u++) {
        int g; size_t k;
        locate(ps, b, u - b * R->ub, &g, &k);
        if (gecm_curve_flag(ps->jobs[g].ctx, stage, k)) return 1;
    }
    return 0;
}

/* the factor lines of batch b (ecm.c:1336-1367 for stage 1, 1485-1522 for stage 2): stdout and ecm_results.txt text.
 * b1_label: the number printed as "B1 = " / "B2 = ". */
static void factor_lines(pass_t *ps, int stage, size_t b, uint64_t label, text_t *res, text_t *out)
{
    const run_t *R = ps->run;
    static __thread char fac[4096];
    for (size_t l = 0; l < (size_t)VECLEN * (size_t)R->threads; l++) {
        int g; size_t k;
        locate(ps, b, l, &g, &k);
        if (!gecm_curve_flag(ps->jobs[g].ctx, stage, k)) continue;
        int prp = 0;
        int r = stage == 1 ? gecm_stage1_factor(ps->jobs[g].ctx, k, fac, sizeof fac, &prp)
                           : gecm_stage2_factor(ps->jobs[g].ctx, k, fac, sizeof fac, &prp);
        if (r != 1) continue;
        const size_t j = l / VECLEN, i = l % VECLEN;
        const size_t curve = (size_t)R->threads * VECLEN * (ps->b0 + b) + l;       /* threads*curve + j*VECLEN + i */
        const unsigned long sg = (unsigned long)ps->jobs[g].sigma[k];
        text_printf(out, "\nfound %s%d factor %s in stage %d (B%d = %lu): thread %zu, vec %zu, sigma %lu\n", prp ? "PRP" : "C",
                    gecm_sizeinbase10(fac), fac, stage, stage, (unsigned long)label, j, i, sg);
        text_printf(res, "\nfound %s%d factor %s in stage %d (B%d = %lu): curve %zu, thread %zu, vec %zu, sigma %lu\n",
                    prp ? "PRP" : "C", gecm_sizeinbase10(fac), fac, stage, stage, (unsigned long)label, curve, j, i, sg);
    }
}

/* resume lines of the pass's distinct curves, formatted by worker threads (gecm_format_resume_line only reads the
 * downloaded x, z) */
typedef struct {
    pass_t *ps;
    uint64_t b1_field;
    size_t lo, hi;
    char **lines;
} fmt_job;

static void *fmt_run(void *arg)
{
    fmt_job *f = (fmt_job *)arg;
    static __thread char line[16384];
    const run_t *R = f->ps->run;
    for (size_t u = f->lo; u < f->hi; u++) {
        int g = 0;
        while (g + 1 < R->gpus && u >= f->ps->jobs[g].first + f->ps->jobs[g].ncurves) g++;
        int n = gecm_format_resume_line(f->ps->jobs[g].ctx, u - f->ps->jobs[g].first, f->b1_field, line, sizeof line);
        f->lines[u] = n > 0 ? strdup(line) : NULL;
    }
    return NULL;
}

static char **format_lines(pass_t *ps, uint64_t b1_field, size_t upto)
{
    char **lines = (char **)calloc(ps->ucurves ? ps->ucurves : 1, sizeof(char *));
    if (!lines) return NULL;
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    int nt = ncpu > 16 ? 16 : ncpu < 1 ? 1 : (int)ncpu;
    if ((size_t)nt > upto / 512 + 1) nt = (int)(upto / 512 + 1);
    fmt_job fj[16];
    pthread_t th[16];
    for (int t = 0; t < nt; t++) {
        fj[t].ps = ps; fj[t].b1_field = b1_field; fj[t].lines = lines;
        fj[t].lo = upto * (size_t)t / (size_t)nt;
        fj[t].hi = upto * (size_t)(t + 1) / (size_t)nt;
    }
    for (int t = 1; t < nt; t++)
        if (pthread_create(&th[t], NULL, fmt_run, &fj[t])) { fmt_run(&fj[t]); th[t] = 0; }
    fmt_run(&fj[0]);
    for (int t = 1; t < nt; t++)
        if (th[t]) pthread_join(th[t], NULL);
    return lines;
}

/* batches 0 .. nbatches of `lines` to f in the reference's order: per batch, thread by thread, lane by lane */
static void write_batches(const pass_t *ps, FILE *f, char **lines, size_t b_from, size_t b_to)
{
    const run_t *R = ps->run;
    for (size_t b = b_from; b < b_to; b++)
        for (size_t l = 0; l < (size_t)VECLEN * (size_t)R->threads; l++) {
            const size_t u = b * R->ub + (R->fixed_sigma ? l % VECLEN : l);
            if (lines[u]) fputs(lines[u], f);
        }
}

static void free_lines(char **lines, size_t n)
{
    if (!lines) return;
    for (size_t i = 0; i < n; i++) free(lines[i]);
    free(lines);
}

/* ---- one pass --------------------------------------------------------------------------------- */
static void pass_fail(pass_t *ps)
{
    run_t *R = ps->run;
    pthread_mutex_lock(&R->mu);
    R->failed = 1;
    if (R->gpu_turn <= ps->index) R->gpu_turn = ps->index + 1;
    if (R->out_turn <= ps->index) R->out_turn = ps->index + 1;
    pthread_cond_broadcast(&R->cv);
    pthread_mutex_unlock(&R->mu);
}

static void *pass_run(void *arg)
{
    pass_t *ps = (pass_t *)arg;
    run_t *R = ps->run;
    const int G = R->gpus;
    const size_t lines_per_batch = (size_t)VECLEN * (size_t)R->threads;
    double t;

    /* host: the curves (ecm.c:1177-1204) */
    t = now();
    if (run_all(ps->jobs, G, job_build, NULL, 0)) { pass_fail(ps); return NULL; }
    ps->t_build = now() - t;

    /* the GPUs, in pass order */
    pthread_mutex_lock(&R->mu);
    while (R->gpu_turn != ps->index && !R->failed) pthread_cond_wait(&R->cv, &R->mu);
    const int stop = R->found_pass < ps->index || R->failed;
    pthread_mutex_unlock(&R->mu);
    if (stop) {                                   /* an earlier pass found a factor: this one is not run at all */
        pthread_mutex_lock(&R->mu);
        R->gpu_turn = ps->index + 1;
        pthread_cond_broadcast(&R->cv);
        while (R->out_turn != ps->index && !R->failed) pthread_cond_wait(&R->cv, &R->mu);
        R->out_turn = ps->index + 1;
        pthread_cond_broadcast(&R->cv);
        pthread_mutex_unlock(&R->mu);
        return NULL;
    }
    plog(ps, "\nCommencing curves %zu-%zu of %zu\n", lines_per_batch * ps->b0, lines_per_batch * (ps->b0 + ps->nb) - 1,
         (size_t)R->threads * R->per_thread);                                                      /* ecm.c:1201 */
    plog(ps, "Building curves took %1.4f seconds.\n", ps->t_build);                                /* ecm.c:1204 */
    t = now();
    gecm_stage1_stats st;
    memset(&st, 0, sizeof st);
    pthread_mutex_lock(&R->mu);
    if (R->rd_pending) { pthread_join(R->rd_thread, NULL); R->rd_pending = 0; }
    pthread_mutex_unlock(&R->mu);
    if (R->rd_rc) { fprintf(stderr, "%s\n", gecm_last_error()); pass_fail(ps); return NULL; }
    for (int r = 0; r < R->nranges; r++) {                                                         /* ecm.c:1209-1312 */
        const gecm_stage1_range_desc rd = R->rd[r];
        /* the reference sieves range 0 once before its first batch (ecm.c:1139-1146) and again whenever a batch
         * starts after a later range was loaded (ecm.c:1160-1173) */
        if (r > 0 || R->nranges > 1)
            plog(ps, "Found %lu primes in range [%lu : %lu]\n", (unsigned long)rd.nprimes, (unsigned long)rd.lo, (unsigned long)rd.hi);   /* ecm.c:1228 */
        plog(ps, "Commencing Stage 1 @ prime %lu\n", (unsigned long)rd.first_prime);               /* ecm.c:1233 */
        for (int g = 0; g < G; g++) { ps->jobs[g].B1 = R->B1; ps->jobs[g].range = (uint32_t)r; ps->jobs[g].progress = ps->live && g == 0; }
        int make_map = 0;
        first_range_t *fr = NULL;
        if (R->do_stage2 && r == R->nranges - 1) {
            fr = &R->fr;
            pthread_mutex_lock(&fr->mu);
            if (!fr->claimed) { fr->claimed = 1; make_map = 1; }
            pthread_mutex_unlock(&fr->mu);
        }
        if (run_all(ps->jobs, G, job_stage1, fr, make_map)) { pass_fail(ps); return NULL; }
        gecm_get_stage1_stats(ps->jobs[0].ctx, &st);
        plog(ps, "\nStage 1 completed at prime %lu with %lu point-adds and %lu point-doubles\n",
             (unsigned long)st.last_prime, (unsigned long)st.ptadds, (unsigned long)st.ptdups);     /* ecm.c:1849 */
        if (rd.checkpoint) {
            /* ecm.c:1236-1312: the batch goes to checkpoint.txt with the last prime in the B1 field; factors are
             * looked for and reported as after stage 1 proper.  Written now (a checkpoint is for the crash that
             * may follow), all batches of the pass; put into the reference's order when the pass is over. */
            struct ck_t *ck = &ps->ck[ps->nck];
            memset(ck, 0, sizeof *ck);
            ck->last_prime = rd.last_prime;
            ck->lines = format_lines(ps, rd.last_prime, ps->ucurves);
            ck->first_flagged = ps->nb;
            for (size_t b = 0; b < ps->nb; b++)
                if (batch_flagged(ps, 1, b)) { ck->first_flagged = b; break; }
            if (ck->first_flagged < ps->nb) {
                factor_lines(ps, 1, ck->first_flagged, rd.last_prime, &ck->res, &ck->out);
                if (ck->out.len) plog(ps, "%s", ck->out.buf);
            }
            FILE *cf = fopen("checkpoint.txt", "a");
            if (cf) {
                plog(ps, "Saving checkpoint after p=%lu\n", (unsigned long)rd.last_prime);         /* ecm.c:1244 */
                if (ps->nck == 0) { fseek(cf, 0, SEEK_END); ps->ck_offset = ftell(cf); }
                if (ck->lines) write_batches(ps, cf, ck->lines, 0, ps->nb);
                fclose(cf);
            } else
                plog(ps, "could not open checkpoint.txt for appending, Stage 1 data will not be saved\n");
            ps->nck++;
        }
    }
    ps->t_stage1 = now() - t;
    plog(ps, "Stage 1 took %1.4f seconds\n", ps->t_stage1);                                         /* ecm.c:1317 */
    plog(ps, "(%.1f curves/sec; kernel %.1f ms on GPU 0)\n", (double)ps->ucurves / ps->t_stage1, ps->jobs[0].kernel_ms);

    gecm_stage2_stats s2;
    memset(&s2, 0, sizeof s2);
    text_t s2log = {0, 0, 0};
    if (R->do_stage2) {                                                                            /* ecm.c:1394-1528 */
        t = now();
        if (run_all(ps->jobs, G, job_stage2_init, NULL, 0)) { pass_fail(ps); return NULL; }        /* ecm.c:1401-1421 */
        ps->t_s2init = now() - t;
        text_printf(&s2log, "Stage 2 Init took %1.4f seconds\n", ps->t_s2init);                    /* ecm.c:1421 */
        gecm_get_stage2_stats(ps->jobs[0].ctx, &s2);
        uint32_t rcount = 0;
        for (uint32_t i = 0; i < 2 * s2.D; i++) {                                                  /* main.c:874-882: R - 3 */
            uint32_t a = i, b = 2 * s2.D;
            while (b) { uint32_t rr = a % b; a = b; b = rr; }
            rcount += a == 1;
        }
        first_range_t *fr = &R->fr;
        for (uint64_t p = R->B1; p < R->B2; p += PRIME_RANGE) {                                    /* ecm.c:1424-1476 */
            const uint64_t hi = p + PRIME_RANGE < R->B2 ? p + PRIME_RANGE : R->B2;
            gecm_pairs pm;
            const int shared = fr->valid && p == fr->lo && hi == fr->hi && s2.D == fr->D && s2.U == fr->U;
            text_printf(&s2log, "commencing pair on range %lu:%lu\n", (unsigned long)p, (unsigned long)hi);   /* ecm.c:2568 */
            if (shared) pm = fr->pm;
            else if (gecm_pair_primes(&pm, p, hi, s2.D, s2.U)) { fprintf(stderr, "%s\n", gecm_last_error()); pass_fail(ps); return NULL; }
            text_printf(&s2log, "%u pairs found from %u primes (ratio = %1.2f)\n", pm.pairs, pm.primes,
                        pm.primes ? (double)pm.pairs / (double)pm.primes : 0.0);                   /* ecm.c:2904-2905 */
            text_printf(&s2log, "\ncommencing stage 2 at A=%lu\nw = %u, R = %u, L = %u, U = %d, umax = %u, amin = %u\n",
                        2ul * (unsigned long)pm.amin * s2.D, s2.D, rcount, s2.L, (int)s2.U, s2.U * s2.D, pm.amin);   /* ecm.c:2440-2442 */
            for (int g = 0; g < G; g++) ps->jobs[g].pm = &pm;
            if (run_all(ps->jobs, G, job_stage2_pair, NULL, 0)) { pass_fail(ps); return NULL; }
            if (!shared) gecm_pairmap_release(&pm);
            gecm_get_stage2_stats(ps->jobs[0].ctx, &s2);
            text_printf(&s2log, "\nlast amin: %u\n", s2.amin_last);                                /* ecm.c:1462 */
        }
        if (run_all(ps->jobs, G, job_stage2_scan, NULL, 0)) { pass_fail(ps); return NULL; }
        ps->t_s2 = now() - t;
        text_printf(&s2log, "\nStage 2 took %1.4f seconds\n", ps->t_s2);                           /* ecm.c:1481 */
        text_printf(&s2log, "performed %lu pt-adds, %lu inversions, and %lu pair-muls in stage 2\n",
                    (unsigned long)s2.ptadds, (unsigned long)s2.numinv, (unsigned long)s2.paired); /* ecm.c:1482 */
    }
    /* What the reference would have written for these batches, one after the other: the first batch in which anything
     * was found — at a checkpoint, after stage 1 or after stage 2 — is the last one written.  Known from the device
     * scans alone, so it is settled before the GPUs go on: a pass behind a factor is not even started. */
    size_t bstar = ps->nb;
    for (int c = 0; c < ps->nck; c++)
        if (ps->ck[c].first_flagged < bstar) bstar = ps->ck[c].first_flagged;
    for (size_t b = 0; b < bstar; b++)
        if (batch_flagged(ps, 1, b) || (R->do_stage2 && batch_flagged(ps, 2, b))) { bstar = b; break; }
    const int found = bstar < ps->nb;
    /* the GPUs go to the next pass */
    pthread_mutex_lock(&R->mu);
    if (found && R->found_pass > ps->index) R->found_pass = ps->index;
    R->gpu_turn = ps->index + 1;
    pthread_cond_broadcast(&R->cv);
    pthread_mutex_unlock(&R->mu);

    const size_t nwrite = found ? bstar + 1 : ps->nb;
    char **lines = format_lines(ps, R->B1, nwrite * R->ub);
    text_t res = {0, 0, 0}, out1 = {0, 0, 0}, out2 = {0, 0, 0};
    if (found) {
        for (int c = 0; c < ps->nck; c++)
            if (ps->ck[c].first_flagged == bstar && ps->ck[c].res.len) text_add(&res, ps->ck[c].res.buf, ps->ck[c].res.len);
        factor_lines(ps, 1, bstar, R->B1, &res, &out1);
        if (R->do_stage2) factor_lines(ps, 2, bstar, R->B2, &res, &out2);
    }

    /* files and stdout, in pass order */
    pthread_mutex_lock(&R->mu);
    while (R->out_turn != ps->index && !R->failed) pthread_cond_wait(&R->cv, &R->mu);
    const int skip = R->found_pass < ps->index || R->failed;
    pthread_mutex_unlock(&R->mu);
    if (!skip) {
        if (!ps->live && ps->log.len) fputs(ps->log.buf, stdout);
        if (out1.len) fputs(out1.buf, stdout);
        FILE *save = fopen("save_b1.txt", "a");
        if (save) { write_batches(ps, save, lines, 0, nwrite); fclose(save); }
        else printf("could not open save_b1.txt for appending, Stage 1 data will not be saved\n");
        if (s2log.len) fputs(s2log.buf, stdout);
        if (out2.len) fputs(out2.buf, stdout);
        if (res.len) {
            FILE *o = fopen("ecm_results.txt", "a");
            if (o) { fputs(res.buf, o); fclose(o); }
        }
        /* checkpoint.txt of a pass of several batches: the reference has them batch by batch (all ranges of batch
         * 0, then batch 1, ...) and nothing after the batch that found a factor */
        if (ps->nck && (ps->nb > 1 || found)) {
            FILE *cf = fopen("checkpoint.txt", "r+");
            if (cf) {
                if (ftruncate(fileno(cf), ps->ck_offset) == 0) {
                    fseek(cf, 0, SEEK_END);
                    for (size_t b = 0; b < nwrite; b++)
                        for (int c = 0; c < ps->nck; c++)
                            if (ps->ck[c].lines) write_batches(ps, cf, ps->ck[c].lines, b, b + 1);
                }
                fclose(cf);
            }
        }
        fflush(stdout);
    }
    pthread_mutex_lock(&R->mu);
    R->out_turn = ps->index + 1;
    pthread_cond_broadcast(&R->cv);
    pthread_mutex_unlock(&R->mu);
    free_lines(lines, ps->ucurves);
    free(res.buf); free(out1.buf); free(out2.buf); free(s2log.buf);
    return NULL;
}

static void pass_release(pass_t *ps)
{
    for (int c = 0; c < ps->nck; c++) {
        free_lines(ps->ck[c].lines, ps->ucurves);
        free(ps->ck[c].res.buf);
        free(ps->ck[c].out.buf);
    }
    free(ps->ck);
    free(ps->sigma);
    free(ps->log.buf);
    memset(ps, 0, sizeof *ps);
}

int main(int argc, char **argv)
{
    if (argc < 4) {
        printf("usage: avx-ecm $input $numcurves $B1 [$threads] [$B2] [$sigma]\n");   /* main.c:382 */
        return 1;
    }
    static run_t R;
    R.t_start = now();
    printf("starting process %d\n", (int)getpid());                               /* main.c:391 */
    /* main.c:393-457: evaluate the expression, recognise Cunningham-type inputs, strip algebraic factors */
    static char ndec[MPL_MAXL * 10 + 16], prep_log[65536];
    gecm_input_info inf;
    if (gecm_prepare_input(argv[1], GECM_CLI_DIGITBITS, ndec, sizeof ndec, &inf, prep_log, sizeof prep_log)) {
        fputs(prep_log, stdout);
        printf("input must evaluate to an odd integer >= 3 (operators + - * / %% ^ ! # fib() luc())\n");
        return 1;
    }
    size_t numcurves = strtoul(argv[2], NULL, 10);
    R.B1 = strtoull(argv[3], NULL, 10);
    R.B2 = 100ULL * R.B1;                                                         /* main.c:462 */
    R.threads = 1;
    R.do_stage2 = 1;
    if (argc > 4) R.threads = atoi(argv[4]);
    if (argc > 5) R.B2 = strtoull(argv[5], NULL, 10);
    if (argc > 6) R.sigma0 = strtoull(argv[6], NULL, 10);
    if (R.B2 <= R.B1) { R.do_stage2 = 0; R.B2 = R.B1; }                           /* main.c:548-552 */
    if (R.threads < 1) R.threads = 1;
    R.fixed_sigma = R.sigma0 > 0;                                                 /* main.c:754-770 */
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
    R.gpus = gpus;
    if (numcurves == 0 || R.B1 < 2 || R.B1 > 1000000000000ULL) { printf("need curves >= 1 and 2 <= B1 <= 1e12\n"); return 1; }
    R.nranges = gecm_stage1_ranges(R.B1);
    /* main.c:585-589: at least one curve per thread, the same number on every thread; ecm.c:1151: every thread
     * runs whole vectors of VECLEN curves, so "10 curves" on one thread writes 16 resume lines there and here */
    if (numcurves < (size_t)R.threads) numcurves = (size_t)R.threads;
    R.per_thread = numcurves / (size_t)R.threads + (numcurves % (size_t)R.threads != 0);
    R.nbatches = (R.per_thread + VECLEN - 1) / VECLEN;
    R.ub = R.fixed_sigma ? VECLEN : (size_t)VECLEN * (size_t)R.threads;

    fputs(prep_log, stdout);          /* "gen: ...", "removing algebraic ...", "commencing parallel ecm on ..." */
    R.rd = (gecm_stage1_range_desc *)calloc((size_t)R.nranges, sizeof *R.rd);
    if (!R.rd) { fprintf(stderr, "out of memory\n"); return 2; }
    R.rd_pending = pthread_create(&R.rd_thread, NULL, describe_ranges, &R) == 0;
    if (!R.rd_pending) describe_ranges(&R);
    /* Special-form inputs for which the reference leaves REDC (main.c:505-527, 642-684): it then works modulo
     * Mw = 2^k - 1, 2^k + 1 or 2^k - c throughout, curve construction included, and keeps the number given for the "N="
     * of its files and for its factor checks (ecm.c:1111-1118).  Same here: the contexts are made on Mw and report
     * against N (gecm_set_report_modulus); the files come out as the reference's, byte for byte. */
    static char mwdec[MPL_MAXL * 10 + 16];
    const char *modulus = ndec;
    if (inf.ref_special_reduction) {
        mpl_t mw, t;
        mpl_set_u64(&mw, 1);
        mpl_shl(&mw, &mw, (unsigned)inf.k);
        if (inf.form < 0) { mpl_set_u64(&t, 1); mpl_add(&mw, &mw, &t); }
        else { mpl_set_u64(&t, inf.form > 1 ? (uint64_t)inf.c : 1); mpl_sub(&mw, &mw, &t); }
        mpl_get_dec(mwdec, &mw);
        modulus = mwdec;
    }
    static gecm_ctx *ctx[2][MAX_GPUS];
    for (int g = 0; g < gpus; g++) {
        if (gecm_create(&ctx[0][g], g % devices, modulus, GECM_CLI_DIGITBITS)) { fprintf(stderr, "%s\n", gecm_last_error()); return 2; }
        if (inf.ref_special_reduction && gecm_set_report_modulus(ctx[0][g], ndec)) { fprintf(stderr, "%s\n", gecm_last_error()); return 2; }
    }
    /* passes: as many reference batches as fit FULL_BATCH distinct curves per GPU — or what the device's memory takes
     * (the stage-2 table of 1024-bit curves is 1.1 MB per curve: 149 GB for a full batch).  GECM_PASS_CURVES (distinct
     * curves per pass over all GPUs) overrides it for tests. */
    uint64_t mem_free = 0, mem_total = 0;
    (void)gecm_device_memory(ctx[0][0], &mem_free, &mem_total);
    const uint64_t budget = mem_free / 10 * 9 / (uint64_t)per_gpu;
    size_t fit = FULL_BATCH;
    while (fit > 64 && mem_free && gecm_batch_bytes(ctx[0][0], fit, R.do_stage2, R.B1, 0, 0) > budget) fit = fit / 2 / 64 * 64;
    if (fit < 64) fit = 64;
    size_t cap = fit * (size_t)gpus;
    if (getenv("GECM_PASS_CURVES") && atol(getenv("GECM_PASS_CURVES")) > 0) cap = (size_t)atol(getenv("GECM_PASS_CURVES"));
    size_t batches_per_pass = cap / R.ub;
    if (batches_per_pass < 1) batches_per_pass = 1;
    const size_t npasses = (R.nbatches + batches_per_pass - 1) / batches_per_pass;
    /* two sets of contexts when there is more than one pass to overlap (one prime range only: with several, a pass
     * takes minutes to hours and its checkpoints are written as it goes) — and when two passes' worth of device memory
     * is there: a full pass that needs more than 45 % of it stays alone on its GPU */
    const size_t pass_curves_per_gpu = (batches_per_pass * R.ub + (size_t)gpus - 1) / (size_t)gpus;
    const int room = !mem_free || 2 * gecm_batch_bytes(ctx[0][0], pass_curves_per_gpu, R.do_stage2, R.B1, 0, 0) <= budget;
    /* (B1 in (99999989, 1e8] is one range WITH a checkpoint, ecm.c:1237: checkpoint.txt is appended to inside a pass's
     * turn on the GPU and put in order when the pass is written, which two passes in flight would do to each other) */
    const int slots = (npasses > 1 && R.nranges == 1 && R.B1 <= 99999989ULL && room && !getenv("GECM_NO_PIPELINE")) ? 2 : 1;
    for (int s = 1; s < slots; s++)
        for (int g = 0; g < gpus; g++) {
            if (gecm_create(&ctx[s][g], g % devices, modulus, GECM_CLI_DIGITBITS)) { fprintf(stderr, "%s\n", gecm_last_error()); return 2; }
            if (inf.ref_special_reduction && gecm_set_report_modulus(ctx[s][g], ndec)) { fprintf(stderr, "%s\n", gecm_last_error()); return 2; }
        }
    gecm_config cfg;
    gecm_get_config(ctx[0][0], &cfg);
    char devname[256];
    gecm_device_name(ctx[0][0], devname, sizeof devname);
    /* main.c:529-533, verbatim: DIGITBITS and VECLEN describe the vector format at the boundary (curves come in
     * groups of 8, limbs of 52 bits); the device's own numbers follow on a line of their own.  For a special-form run
     * the "input size" is k, the size of 2^k -/+ c (main.c:465-483 on size_n = k). */
    printf("ECM has been configured with DIGITBITS = %d, VECLEN = %d, GMP_LIMB_BITS = %d\n", cfg.digitbits, VECLEN, 64);
    printf("Choosing MAXBITS = %d, NWORDS = %d, NBLOCKS = %d based on input size %d\n", cfg.maxbits, cfg.nwords,
           cfg.nwords / 4, inf.ref_special_reduction ? inf.k : cfg.nbits);
    printf("%s: %d GPU(s) [%s], residues of %d limbs x 28 bits on the device\n", gecm_version(), gpus, devname,
           cfg.dev_limbs);
    if (argc > 6) printf("starting with sigma = %lu\n", (unsigned long)R.sigma0);  /* main.c:558 */
    printf("Input has %d bits, using %d threads (%d curves/thread)\n", inf.ref_special_reduction ? inf.nbits : cfg.nbits,
           R.threads, (int)R.per_thread);                                          /* main.c:591-592 */
    printf("Processing in batches of %u primes\n", 100000000u);                   /* main.c:593 */
    if (inf.ref_special_reduction) {                                               /* main.c:644-670 */
        if (inf.form > 1) printf("Using special pseudo-Mersenne mod for factor of: 2^%d-%d\n", inf.k, inf.form);
        else printf("Using special Mersenne mod for factor of: 2^%d%c1\n", inf.k, inf.form > 0 ? '-' : '+');
        int fk = 0, fl = 0;
        if (gecm_get_special_form(ctx[0][0], &fk, &fl) >= 1)
            printf("(batches that fill the GPU multiply modulo it with a special reduction, %d limbs; smaller ones by REDC)\n", fl);
    }
    printf("Initialization took %1.4f seconds.\n", now() - R.t_start);             /* main.c:776 */
    fflush(stdout);

    pthread_mutex_init(&R.fr.mu, NULL);
    pthread_cond_init(&R.fr.cv, NULL);
    pthread_mutex_init(&R.mu, NULL);
    pthread_cond_init(&R.cv, NULL);
    R.fr.lo = R.B1;
    R.fr.hi = R.B1 + PRIME_RANGE < R.B2 ? R.B1 + PRIME_RANGE : R.B2;
    R.fr.D = gecm_s2_default_D(R.B1);
    R.fr.U = GECM_S2_DEFAULT_U;
    R.lcg = (uint64_t)(R.t_start * 1e6) * 0x9E3779B97F4A7C15ULL + (uint64_t)getpid();
    R.found_pass = (size_t)-1;

    if (R.rd_pending) { pthread_join(R.rd_thread, NULL); R.rd_pending = 0; }
    if (!R.rd_rc && R.nranges == 1)                                                /* ecm.c:1139-1146 */
        printf("Found %lu primes in range [%lu : %lu]\n", (unsigned long)R.rd[0].nprimes, (unsigned long)R.rd[0].lo, (unsigned long)R.rd[0].hi);
    static pass_t pass[2];
    int running[2] = {0, 0};
    for (size_t pi = 0; pi < npasses; pi++) {
        const int s = (int)(pi % (size_t)slots);
        if (running[s]) { pthread_join(pass[s].th, NULL); pass_release(&pass[s]); running[s] = 0; }
        pthread_mutex_lock(&R.mu);
        const int stop = R.found_pass != (size_t)-1 || R.failed;
        pthread_mutex_unlock(&R.mu);
        if (stop) break;
        pass_t *ps = &pass[s];
        memset(ps, 0, sizeof *ps);
        ps->run = &R;
        ps->index = pi;
        ps->slot = s;
        ps->live = slots == 1;
        ps->b0 = pi * batches_per_pass;
        ps->nb = R.nbatches - ps->b0 < batches_per_pass ? R.nbatches - ps->b0 : batches_per_pass;
        ps->ucurves = ps->nb * R.ub;
        ps->sigma = (uint64_t *)malloc(ps->ucurves * sizeof(uint64_t));
        ps->ck = (struct ck_t *)calloc((size_t)R.nranges + 1, sizeof(struct ck_t));
        if (!ps->sigma || !ps->ck) { fprintf(stderr, "out of memory\n"); return 2; }
        for (size_t u = 0; u < ps->ucurves; u++) {
            /* fixed sigma: lane i of every thread of batch b runs sigma + 8 b + i (main.c:761, ecm.c:1187) */
            if (R.fixed_sigma) ps->sigma[u] = R.sigma0 + VECLEN * ps->b0 + u;
            else do { ps->sigma[u] = lcg_rand(&R.lcg); } while (ps->sigma[u] < 6);   /* ecm.c:1564-1570 */
        }
        /* host-side split: GPU g owns distinct curves [n*g/G, n*(g+1)/G) of this pass */
        for (int g = 0; g < gpus; g++) {
            const size_t lo = ps->ucurves * (size_t)g / (size_t)gpus, hi = ps->ucurves * (size_t)(g + 1) / (size_t)gpus;
            ps->jobs[g].gpu = g;
            ps->jobs[g].ctx = ctx[s][g];
            ps->jobs[g].first = lo;
            ps->jobs[g].ncurves = hi - lo;
            ps->jobs[g].sigma = ps->sigma + lo;
        }
        if (pthread_create(&ps->th, NULL, pass_run, ps)) { pass_run(ps); pass_release(ps); }
        else running[s] = 1;
    }
    for (size_t k = 0; k < 2; k++) {
        /* in pass order: the older of the two first */
        const int s = (int)((npasses + k) % 2);
        if (s < slots && running[s]) { pthread_join(pass[s].th, NULL); pass_release(&pass[s]); running[s] = 0; }
    }
    for (int s = 0; s < 2; s++)
        if (running[s]) { pthread_join(pass[s].th, NULL); pass_release(&pass[s]); }
    if (R.fr.valid) gecm_pairmap_release(&R.fr.pm);
    for (int s = 0; s < slots; s++)
        for (int g = 0; g < gpus; g++) gecm_destroy(ctx[s][g]);
    printf("Process took %1.4f seconds.\n", now() - R.t_start);                    /* ecm.c:1538 */
    return R.failed ? 2 : 0;
}
