/* gecm_api.c — public C ABI of libgecm (include/gecm.h): host logic in C above the HIP device
 * layer (csrc/gecm_dev.h).  Mirrors the reference's setup and phase functions:
 *   Montgomery constants        main.c:597-640
 *   NWORDS/MAXBITS rule         main.c:465-483
 *   build_one_curve             ecm.c:1548-1803
 *   ecm_stage1                  ecm.c:1806-1854 (tape built by gecm_plan.c, run by the device)
 *   save line / check_factor    ecm.c:1319-1388, 2542-2557
 */
#include "../../include/gecm.h"
#include "../csrc/gecm_dev.h"
#include "gecm_plan.h"
#include "gecm_pair.h"
#include "mpl.h"
#include "cunningham.h"
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define LIMB_BITS 28
/* stage 2: giant steps per device chunk (one inversion each) and ring size (power of two >= chunk + 2L) */
#define S2_GIANT_CHUNK 512u
#define S2_RING 1024u
/* largest B1: the 32-bit offsets of a range's tape and uint32 range indices are nowhere near it; the cap is the
 * reference's own (its prime sieve serves ranges below 10^13 or so; ecm.c keeps primes in 64 bits) kept at a size one
 * can still test */
#define GECM_B1_MAX 1000000000000ull
#define gecm_stage1_ranges_u(b1) gecm_stage1_ranges_plan(b1)

static __thread char g_err[512];
static void set_err(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *gecm_last_error(void) { return g_err; }
/* "libgecm 0.3 (gfx950) K:<hash> R:<hash> D:<hash> H:<hash>": the hashes of the sources the objects inside this
 * library were compiled from (avx-ecm_amd/Makefile: kernels, 32-lane kernels, device layer, host C), MIXED where
 * objects of one group disagree.  tests/test_abi_cpu.py and __graft_entry__.smoke() recompute them from the tree. */
#ifndef GECM_MANIFEST
#define GECM_MANIFEST "unset"
#endif
const char *gecm_manifest_host_gecm_plan(void);
const char *gecm_manifest_host_gecm_pair(void);
const char *gecm_manifest_host_mpl(void);
const char *gecm_manifest_host_calc_lite(void);
const char *gecm_manifest_host_cunningham(void);
const char *gecm_version(void)
{
    static char v[256];
    if (!v[0]) {
        const char *h[5] = {gecm_manifest_host_gecm_plan(), gecm_manifest_host_gecm_pair(), gecm_manifest_host_mpl(),
                            gecm_manifest_host_calc_lite(), gecm_manifest_host_cunningham()};
        int mixed = 0;
        for (int i = 0; i < 5; i++) mixed |= strcmp(h[i], GECM_MANIFEST) != 0;
        snprintf(v, sizeof v, "libgecm 0.3 (gfx950) %s H:%s", gecm_dev_manifest(), mixed ? "MIXED" : GECM_MANIFEST);
    }
    return v;
}
int gecm_device_count(void) { return gecm_dev_count(); }

struct gecm_ctx {
    int device, digitbits, nwords, maxbits, nbits, nl;
    mpl_t N;
    mpl_t N_report;      /* gecm_set_report_modulus: the number the save lines name and factors are looked for in */
    int have_report;
    mpl_t rref_mod_n;    /* 2^(digitbits*nwords) mod N  = the reference's "one" */
    mpl_t rint_mod_n;    /* 2^(28*nl) mod N */
    mpl_t ref_to_int;    /* Rint * Rref^-1 mod N : x*Rref -> x*Rint by plain modular multiply */
    mpl_t int_to_ref;    /* Rref * Rint^-1 mod N */
    uint64_t rho_ref;
    uint32_t rho28;
    uint32_t *n28, *kp28, *one28, *fix28; /* fix28 = Rint^2/Rref mod N (see gecm_dev_l0) */
    gecm_dev *dev, *dev_l0;
    /* current batch */
    size_t batch;
    uint64_t *sigma;
    uint8_t *bad;
    uint32_t *hx, *hz;   /* last downloaded plain x, z: [nl][batch] */
    int have_plain;
    uint64_t B1;
    /* tape cache: the tape of (tape_B1, tape_range), one ecm_stage1 call of the reference */
    gecm_tape_t tape;
    uint64_t tape_B1;
    uint32_t tape_range;
    int tape_on_dev;
    double last_ms;
    /* counters of the stage 1 in progress: summed over the ranges run since range 0 (work->ptadds / ptdups are
     * cleared when the curves are built, ecm.c:1177-1178, and grow through every ecm_stage1 call) */
    uint64_t s1_ptadds, s1_ptdups, s1_last_prime, s1_tape_len;
    /* the next range's tape, compiled by a helper thread while the device runs the current range */
    pthread_t pf_thread;
    int pf_active, pf_rc;
    gecm_tape_t pf_tape;
    uint64_t pf_B1;
    uint32_t pf_range;
    /* stage 2 */
    uint32_t *r3_28;
    gecm_s2_plan s2;
    int s2_ready;
    uint32_t *hacc, *hfail;
    uint32_t fail_planes;    /* planes of hfail: 1, or 1 + sub-sequences (gecm_dev_s2_fail_planes) */
    gecm_pairs pm;           /* pair map of the last single-range gecm_stage2 call (pm_valid), reused while (range, D, U) match */
    int pm_valid;
    gecm_s2_tape tp;         /* the device tape made from pm (tp_valid): it depends on (pm, D, U) only, so it is kept with it */
    int tp_valid;
    uint64_t tp_id;          /* identifies the kept tape to the device side, which then keeps its copy too */
    gecm_s2_tape ptp;        /* the tape of the last gecm_stage2_pair call (ptp_valid), found again by a fingerprint of the map */
    int ptp_valid;
    uint64_t ptp_fp, ptp_fp2, ptp_id;
    uint32_t ptp_steps, ptp_amin, ptp_D, ptp_U;
    uint64_t pm_lo, pm_hi;
    uint32_t pm_D, pm_U;
    int have_acc;
    uint32_t *flags[2];      /* per-curve result of the last device factor scan: stage 1, stage 2 */
    uint32_t *hg[2];         /* and the gcds it computed, [nl][batch] */
    int scan_valid[2];       /* the cached scan belongs to the current stage-1 / stage-2 result */
    uint64_t s2_ptadds, s2_numinv, s2_paired, s2_devinv;
    uint32_t s2_amin_last;
    int lanes_per_curve;     /* 0 = auto, 1, 2, 8, 32 (gecm_set_lanes_per_curve) */
    /* F-form stage 1 for N | 2^k - 1: a second device context working modulo Mw = 2^k - 1
     * (csrc/gecm_field.hpp); results are brought back modulo N by ff_settle() */
    gecm_dev *dev_f;
    int ff_k, ff_sign, ff_nl, ff_on, ff_pending, ff_loaded, last_on_f;
    uint64_t ff_c;           /* Mw = 2^ff_k - ff_c for ff_sign > 0 (1: Mersenne form), 2^ff_k + 1 for ff_sign < 0 */
    uint64_t ff_tape_B1;
    uint32_t ff_tape_range;
    mpl_t ff_M, ff_r_mod_m;  /* Mw; 2^(28 ff_nl) mod Mw */
    uint32_t *ff_n28;        /* n, kp, one for dev_f */
};

static int pick_nl(int nbits)
{
    int need = (nbits + 5 + LIMB_BITS - 1) / LIMB_BITS;   /* R = 2^(28 nl) >= 32 N */
    for (const int *p = gecm_dev_supported_nl(); *p; p++)
        if (*p >= need) return *p;
    return 0;
}

/* K' for the lazy subtraction (csrc/gecm_field.hpp): K = 2^j * mod in [R/32, R/16), written with every
 * limb in [2^28-1, 2^29): +2^28 at limb 0, +2^28-1 in the middle, -1 at the top. */
static int make_kp(uint32_t *kp, const mpl_t *mod, int nl)
{
    mpl_t K;
    uint32_t *kl = (uint32_t *)calloc((size_t)nl, sizeof(uint32_t));
    if (!kl) return -1;
    mpl_shl(&K, mod, (unsigned)(LIMB_BITS * nl - 4 - mpl_bits(mod)));
    mpl_to_limbs32(kl, 1, nl, LIMB_BITS, &K);
    for (int i = 0; i < nl; i++) {
        if (i == 0) kp[i] = kl[i] + (1u << LIMB_BITS);
        else if (i < nl - 1) kp[i] = kl[i] + (1u << LIMB_BITS) - 1;
        else kp[i] = kl[i] - 1;
    }
    free(kl);
    return 0;
}

static void pow2_mod(mpl_t *r, unsigned e, const mpl_t *m)
{
    mpl_t t;
    mpl_set_u64(&t, 1);
    mpl_shl(&t, &t, e);
    mpl_mod(r, &t, m);
}

/* N | 2^k - 1, N | 2^k + 1 or N | 2^k - c with c below one reference limb (the reference's isMersenne == +1 / -1 / c,
 * main.c:410-441): open a second device context modulo Mw = 2^k -/+ 1 or 2^k - c for the special stage-1 multiply
 * (csrc/gecm_field.hpp, F-form / P-form / C-form) when that is the cheaper multiply.  Failure to set it up is not an error: stage 1 then runs modulo N
 * like everything else. */
static void ff_setup(gecm_ctx *c)
{
    cunningham_form f;
    cunningham_detect(&f, &c->N, c->digitbits);
    if ((f.form != 1 && f.form != -1 && f.form != 2) || f.k < 64) return;
    if (f.form == 2 && (f.c < 3 || !(f.c & 1))) return;              /* 2^k - c with c odd, c > 1 (c = 1 is form +1) */
    const int mbits = f.form < 0 ? f.k + 1 : f.k;
    const int nlf = pick_nl(mbits);
    if (!nlf) return;
    const int G = gecm_dev_fform_generic_limbs(nlf);
    if (G < 0 || f.k < LIMB_BITS * (nlf - G)) return;                 /* limbs below nl-G must be F..F / 1,0..0 */
    if (f.form == 2 && nlf - G < 3) return;                          /* limbs 0, 1 carry c - 1: one pure F limb above */
    /* multiply-adds per modular multiplication: about nl^2 + G*nl (+ 3 nl for 2^k - c) against 2 nl^2 + nl */
    if ((double)(nlf * nlf + (G + (f.form == 2 ? 3 : 0)) * nlf) * 1.15 > (double)(2 * c->nl * c->nl + c->nl)) return;
    mpl_t one, t;
    mpl_set_u64(&one, 1);
    mpl_shl(&c->ff_M, &one, (unsigned)f.k);
    if (f.form == 2) { mpl_set_u64(&t, f.c); mpl_sub(&c->ff_M, &c->ff_M, &t); }
    else if (f.form > 0) mpl_sub(&c->ff_M, &c->ff_M, &one);
    else mpl_add(&c->ff_M, &c->ff_M, &one);
    { mpl_t r; mpl_mod(&r, &c->ff_M, &c->N); if (!mpl_is_zero(&r)) return; }   /* N | Mw, or nothing below holds */
    pow2_mod(&c->ff_r_mod_m, (unsigned)(LIMB_BITS * nlf), &c->ff_M);
    c->ff_n28 = (uint32_t *)calloc((size_t)nlf * 3, sizeof(uint32_t));
    if (!c->ff_n28) return;
    mpl_to_limbs32(c->ff_n28, 1, nlf, LIMB_BITS, &c->ff_M);
    mpl_to_limbs32(c->ff_n28 + 2 * nlf, 1, nlf, LIMB_BITS, &c->ff_r_mod_m);
    /* rho = -Mw^-1 mod 2^28: 1 for 2^k - 1, 2^28 - 1 for 2^k + 1, (c mod 2^28)^-1 for 2^k - c */
    uint32_t rho = f.form == 1 ? 1u : (1u << LIMB_BITS) - 1u;
    if (f.form == 2) {
        mpl_t two28, inv;
        mpl_set_u64(&two28, 1u << LIMB_BITS);
        if (!mpl_invmod(&inv, &c->ff_M, &two28)) { free(c->ff_n28); c->ff_n28 = NULL; return; }
        mpl_sub(&inv, &two28, &inv);
        rho = (uint32_t)mpl_get_u64(&inv);
    }
    if (make_kp(c->ff_n28 + nlf, &c->ff_M, nlf) ||
        gecm_dev_open(&c->dev_f, c->device, nlf, c->ff_n28, c->ff_n28 + nlf, c->ff_n28 + 2 * nlf, rho)) {
        free(c->ff_n28);
        c->ff_n28 = NULL;
        c->dev_f = NULL;
        return;
    }
    gecm_dev_set_fform(c->dev_f, f.form);
    c->ff_k = f.k;
    c->ff_sign = f.form < 0 ? -1 : 1;
    c->ff_c = f.form == 2 ? f.c : 1;
    c->ff_nl = nlf;
    c->ff_on = 1;
}

/* Constants of the 32-lanes-per-curve stage-1 kernel (csrc/gecm_row.hpp): it works modulo N' = m*N, the multiple
 * of N that is -1 modulo 2^28 (the Montgomery digit is then the low limb itself), on L limbs (nl + 1 rounded up
 * to whole lanes of nq limbs) with R' = 2^(28 L) >= 32 N'.  Entry factor R'^2/R mod N turns the buffers' x*R into x*R'; the exit multiply by R mod N
 * (modulo N itself) turns it back.  Not an error if it cannot be set up: the other layouts cover every N. */
static void row_setup(gecm_ctx *c)
{
    const int nl = c->nl;
    int nq, L;                                        /* L = rows of a multiply = limbs in use: 28 (nl + 1) >= bits(N') + 5 */
    gecm_row_shape(nl, &nq, &L);
    if (nq > GECM_ROW_MAXNQ) return;
    uint32_t w[GECM_ROW_KINDS * GECM_ROW_WORDS];
    memset(w, 0, sizeof w);
    mpl_t two28, inv, m, np, t;
    mpl_set_u64(&two28, 1u << LIMB_BITS);
    if (!mpl_invmod(&inv, &c->N, &two28)) return;
    mpl_sub(&m, &two28, &inv);                        /* m = -N^-1 mod 2^28, in [1, 2^28) */
    mpl_mul(&np, &m, &c->N);
    if (mpl_bits(&np) + 5 > LIMB_BITS * L) return;
    mpl_to_limbs32(w + 0 * GECM_ROW_WORDS, 1, L, LIMB_BITS, &np);
    if (w[0] != (1u << LIMB_BITS) - 1u) return;
    memcpy(w + 1 * GECM_ROW_WORDS, c->n28, (size_t)nl * sizeof(uint32_t));
    pow2_mod(&t, (unsigned)(2 * LIMB_BITS * L - LIMB_BITS * nl), &c->N);
    mpl_to_limbs32(w + 2 * GECM_ROW_WORDS, 1, nl, LIMB_BITS, &t);
    memcpy(w + 3 * GECM_ROW_WORDS, c->one28, (size_t)nl * sizeof(uint32_t));
    memcpy(w + 4 * GECM_ROW_WORDS, c->kp28, (size_t)nl * sizeof(uint32_t));
    (void)gecm_dev_set_rowconst(c->dev, nq, L, w);
}

int gecm_create(gecm_ctx **out, int device, const char *n_str, int digitbits)
{
    if (!out || !n_str || (digitbits != 52 && digitbits != 32)) {
        set_err("gecm_create: bad argument (digitbits must be 52 or 32)");
        return GECM_ERR_ARG;
    }
    gecm_ctx *c = (gecm_ctx *)calloc(1, sizeof *c);
    if (!c) return GECM_ERR_NOMEM;
    if (mpl_set_str(&c->N, n_str) || !mpl_is_odd(&c->N) || mpl_cmp_u64(&c->N, 3) < 0) {
        set_err("gecm_create: N must be an odd integer >= 3 (decimal or 0x-hex)");
        free(c);
        return GECM_ERR_ARG;
    }
    c->device = device;
    c->digitbits = digitbits;
    c->nbits = mpl_bits(&c->N);
    /* main.c:465-483: MAXBITS = smallest multiple of 208 (128) strictly greater than bitlen */
    int step = digitbits == 52 ? 208 : 128;
    c->maxbits = step;
    while (c->maxbits <= c->nbits) c->maxbits += step;
    c->nwords = c->maxbits / digitbits;
    c->nl = pick_nl(c->nbits);
    if (!c->nl || c->maxbits + 64 > MPL_MAXL * 16) {
        set_err("gecm_create: N of %d bits is larger than this build supports", c->nbits);
        free(c);
        return GECM_ERR_ARG;
    }
    int nl = c->nl;
    unsigned rint_bits = (unsigned)(LIMB_BITS * nl), rref_bits = (unsigned)c->maxbits;
    mpl_t t, inv;
    pow2_mod(&c->rref_mod_n, rref_bits, &c->N);
    pow2_mod(&c->rint_mod_n, rint_bits, &c->N);
    mpl_invmod(&inv, &c->rref_mod_n, &c->N);
    mpl_mulmod(&c->ref_to_int, &c->rint_mod_n, &inv, &c->N);
    mpl_invmod(&inv, &c->rint_mod_n, &c->N);
    mpl_mulmod(&c->int_to_ref, &c->rref_mod_n, &inv, &c->N);
    /* rho = -N^-1 mod 2^digitbits (main.c:627-628, 636-640) and mod 2^28 */
    mpl_t two64;
    mpl_set_u64(&two64, 1);
    mpl_shl(&two64, &two64, 64);
    mpl_invmod(&inv, &c->N, &two64);
    mpl_sub(&t, &two64, &inv);
    uint64_t nhat = mpl_get_u64(&t);
    c->rho_ref = digitbits == 52 ? (nhat & 0xfffffffffffffull) : (nhat & 0xffffffffull);
    c->rho28 = (uint32_t)(nhat & ((1u << LIMB_BITS) - 1));
    c->n28 = (uint32_t *)calloc((size_t)nl * 4, sizeof(uint32_t));
    if (!c->n28) { free(c); return GECM_ERR_NOMEM; }
    c->kp28 = c->n28 + nl;
    c->one28 = c->kp28 + nl;
    c->fix28 = c->one28 + nl;
    mpl_to_limbs32(c->n28, 1, nl, LIMB_BITS, &c->N);
    mpl_to_limbs32(c->one28, 1, nl, LIMB_BITS, &c->rint_mod_n);
    if (make_kp(c->kp28, &c->N, nl)) { free(c->n28); free(c); return GECM_ERR_NOMEM; }
    /* fix = Rint^2 / Rref mod N = Rint * ref_to_int */
    mpl_mulmod(&t, &c->rint_mod_n, &c->ref_to_int, &c->N);
    mpl_to_limbs32(c->fix28, 1, nl, LIMB_BITS, &t);
    if (gecm_dev_open(&c->dev, device, nl, c->n28, c->kp28, c->one28, c->rho28)) {
        set_err("gecm_create: %s", gecm_dev_error());
        free(c->n28);
        free(c);
        return GECM_ERR_DEVICE;
    }
    /* R^3 mod N for the device inversion (csrc/gecm_stage2.hpp: fe_inv_mont) */
    c->r3_28 = (uint32_t *)calloc((size_t)nl, sizeof(uint32_t));
    if (!c->r3_28) { gecm_destroy(c); return GECM_ERR_NOMEM; }
    mpl_mulmod(&t, &c->rint_mod_n, &c->rint_mod_n, &c->N);
    mpl_mulmod(&t, &t, &c->rint_mod_n, &c->N);
    mpl_to_limbs32(c->r3_28, 1, nl, LIMB_BITS, &t);
    /* batches of 28 division steps after which the device inversion has converged for a modulus of nbits bits:
     * the bound of the "half-delta" variant, floor((45907 bits + 26313) / 19929), +1, rounded up to whole batches */
    gecm_dev_set_s2const(c->dev, c->r3_28, (uint32_t)((((45907ull * (unsigned)c->nbits + 26313ull) / 19929ull + 1) + 27) / 28));
    row_setup(c);
    ff_setup(c);
    *out = c;
    return GECM_OK;
}

static void free_batch(gecm_ctx *c)
{
    free(c->sigma); free(c->bad); free(c->hx); free(c->hz); free(c->hacc); free(c->hfail);
    free(c->flags[0]); free(c->flags[1]); free(c->hg[0]); free(c->hg[1]);
    c->flags[0] = c->flags[1] = NULL;
    c->hg[0] = c->hg[1] = NULL;
    c->scan_valid[0] = c->scan_valid[1] = 0;
    c->sigma = NULL; c->bad = NULL; c->hx = c->hz = NULL; c->hacc = c->hfail = NULL;
    c->have_acc = 0; c->s2_ready = 0;
    c->batch = 0;
    c->have_plain = 0;
}

/* the kept pair map of a single-range stage 2 and the device tape made from it */
static void drop_kept_pairmap(gecm_ctx *c)
{
    if (c->pm_valid) gecm_pairmap_release(&c->pm);
    c->pm_valid = 0;
    if (c->tp_valid) free(c->tp.words);
    c->tp_valid = 0;
}

static uint64_t next_tape_id(void)
{
    static uint64_t next_id = 1;                     /* ids only tell tapes apart on the device side */
    return __atomic_fetch_add(&next_id, 1, __ATOMIC_RELAXED);
}

void gecm_destroy(gecm_ctx *c)
{
    if (!c) return;
    gecm_dev_close(c->dev);
    gecm_dev_close(c->dev_l0);
    gecm_dev_close(c->dev_f);
    free(c->ff_n28);
    if (c->pf_active) { pthread_join(c->pf_thread, NULL); c->pf_active = 0; gecm_tape_free(&c->pf_tape); }
    gecm_tape_free(&c->tape);
    free_batch(c);
    gecm_s2_plan_free(&c->s2);
    drop_kept_pairmap(c);
    if (c->ptp_valid) free(c->ptp.words);
    free(c->r3_28);
    free(c->n28);
    free(c);
}

/* The reference's special-form runs (main.c:505-527, 642-684) work modulo Mw = 2^k -/+ 1 or 2^k - c — curve construction,
 * stage 1, stage 2 — while the number given, N | Mw, stays the one its files name and its factor checks use
 * (ecm.c:1111-1118: gmpn = vnhat).  A context on Mw with N as its report modulus writes those files byte for byte. */
int gecm_set_report_modulus(gecm_ctx *c, const char *n_str)
{
    if (!c) return GECM_ERR_ARG;
    if (!n_str) { c->have_report = 0; return GECM_OK; }
    mpl_t r, m;
    if (mpl_set_str(&r, n_str) || mpl_cmp_u64(&r, 1) <= 0) { set_err("gecm_set_report_modulus: bad number"); return GECM_ERR_ARG; }
    mpl_mod(&m, &c->N, &r);
    if (!mpl_is_zero(&m)) { set_err("gecm_set_report_modulus: the number must divide the context's modulus"); return GECM_ERR_ARG; }
    c->N_report = r;
    c->have_report = 1;
    c->scan_valid[0] = c->scan_valid[1] = 0;
    return GECM_OK;
}

static const mpl_t *report_n(const gecm_ctx *c) { return c->have_report ? &c->N_report : &c->N; }

/* g = gcd(value, context modulus) -> the factor of the report modulus it holds */
static void to_report(const gecm_ctx *c, mpl_t *g)
{
    if (c->have_report && !mpl_is_zero(g)) { mpl_t t = *g; mpl_gcd(g, &t, &c->N_report); }
}

int gecm_get_config(const gecm_ctx *c, gecm_config *cfg)
{
    if (!c || !cfg) return GECM_ERR_ARG;
    cfg->digitbits = c->digitbits;
    cfg->nwords = c->nwords;
    cfg->maxbits = c->maxbits;
    cfg->nbits = c->nbits;
    cfg->dev_limbs = c->nl;
    cfg->device = c->device;
    cfg->rho = c->rho_ref;
    return GECM_OK;
}

int gecm_device_memory(gecm_ctx *c, uint64_t *free_bytes, uint64_t *total_bytes)
{
    if (!c) return GECM_ERR_ARG;
    if (gecm_dev_memory(c->dev, free_bytes, total_bytes)) { set_err("%s", gecm_dev_error()); return GECM_ERR_DEVICE; }
    return GECM_OK;
}

/* Device bytes a batch of `curves` curves takes: the stage-1 arrays and, with_stage2, the stage-2 allocations for wheel D
 * and height U (0 = the defaults for B1) — the baby-step table is the largest allocation of the path.
 * The same sums the device layer allocates by. */
uint64_t gecm_batch_bytes(const gecm_ctx *c, size_t curves, int with_stage2, uint64_t B1, uint32_t D, uint32_t U)
{
    if (!c || !curves) return 0;
    uint32_t npb = 0;
    if (with_stage2) {
        if (!D) D = gecm_s2_default_D(B1 ? B1 : 1000000);
        if (!U) U = GECM_S2_DEFAULT_U;
        gecm_s2_plan p;
        memset(&p, 0, sizeof p);
        if (gecm_s2_plan_init(&p, D, U) == 0) {
            npb = p.npb;
            gecm_s2_plan_free(&p);
        }
    }
    uint64_t b = gecm_dev_batch_bytes(c->dev, curves, npb, S2_GIANT_CHUNK, S2_RING);
    if (c->dev_f) b += gecm_dev_batch_bytes(c->dev_f, curves, 0, 0, 0);
    return b;
}

int gecm_device_name(gecm_ctx *c, char *buf, size_t len)
{
    if (gecm_dev_device_name(c->dev, buf, len)) { set_err("%s", gecm_dev_error()); return GECM_ERR_DEVICE; }
    return GECM_OK;
}

/* ---- reference vec layout <-> mpl ---------------------------------------------------------- */
static void vec_get(const gecm_ctx *c, mpl_t *r, const void *vec, size_t batch, size_t lane)
{
    if (c->digitbits == 52) mpl_from_limbs64(r, (const uint64_t *)vec + lane, batch, c->nwords, 52);
    else mpl_from_limbs32(r, (const uint32_t *)vec + lane, batch, c->nwords, 32);
}

static void vec_put(const gecm_ctx *c, void *vec, size_t batch, size_t lane, const mpl_t *v)
{
    if (c->digitbits == 52) mpl_to_limbs64((uint64_t *)vec + lane, batch, c->nwords, 52, v);
    else mpl_to_limbs32((uint32_t *)vec + lane, batch, c->nwords, 32, v);
}

int gecm_get_one(const gecm_ctx *c, void *one_limbs)
{
    if (!c || !one_limbs) return GECM_ERR_ARG;
    vec_put(c, one_limbs, 1, 0, &c->rref_mod_n);
    return GECM_OK;
}

/* ---- L0 ------------------------------------------------------------------------------------ */
static int l0_call(gecm_ctx *c, int op, const void *a, const void *b, void *r0, void *r1, size_t batch)
{
    if (!c || !a || !r0 || batch == 0) { set_err("L0: bad argument"); return GECM_ERR_ARG; }
    if (!c->dev_l0 &&
        gecm_dev_open(&c->dev_l0, c->device, c->nl, c->n28, c->kp28, c->one28, c->rho28)) {
        set_err("L0: %s", gecm_dev_error());
        return GECM_ERR_DEVICE;
    }
    int nl = c->nl;
    size_t words = (size_t)nl * batch;
    uint32_t *ha = (uint32_t *)malloc(words * 4 * 4);
    if (!ha) return GECM_ERR_NOMEM;
    uint32_t *hb = ha + words, *hc = hb + words, *hd = hc + words;
    mpl_t v;
    for (size_t i = 0; i < batch; i++) {
        vec_get(c, &v, a, batch, i);
        if (mpl_cmp(&v, &c->N) >= 0) { free(ha); set_err("L0: operand a[%zu] not < N", i); return GECM_ERR_ARG; }
        mpl_to_limbs32(ha + i, batch, nl, LIMB_BITS, &v);
        if (b) {
            vec_get(c, &v, b, batch, i);
            if (mpl_cmp(&v, &c->N) >= 0) { free(ha); set_err("L0: operand b[%zu] not < N", i); return GECM_ERR_ARG; }
        }
        mpl_to_limbs32(hb + i, batch, nl, LIMB_BITS, &v);
    }
    int rc = gecm_dev_l0(c->dev_l0, op, ha, hb, hc, hd, batch, c->fix28);
    if (rc) { free(ha); set_err("L0: %s", gecm_dev_error()); return GECM_ERR_DEVICE; }
    for (size_t i = 0; i < batch; i++) {
        mpl_from_limbs32(&v, hc + i, batch, nl, LIMB_BITS);
        vec_put(c, r0, batch, i, &v);
        if (op == GECM_L0_ADDSUB) {
            mpl_from_limbs32(&v, hd + i, batch, nl, LIMB_BITS);
            vec_put(c, r1, batch, i, &v);
        }
    }
    free(ha);
    return GECM_OK;
}

int gecm_vecmulmod(gecm_ctx *c, const void *a, const void *b, void *r, size_t batch)
{
    return b ? l0_call(c, GECM_L0_MUL, a, b, r, NULL, batch) : GECM_ERR_ARG;
}
int gecm_vecsqrmod(gecm_ctx *c, const void *a, void *r, size_t batch)
{
    return l0_call(c, GECM_L0_SQR, a, NULL, r, NULL, batch);
}
int gecm_vecaddmod(gecm_ctx *c, const void *a, const void *b, void *r, size_t batch)
{
    return b ? l0_call(c, GECM_L0_ADD, a, b, r, NULL, batch) : GECM_ERR_ARG;
}
int gecm_vecsubmod(gecm_ctx *c, const void *a, const void *b, void *r, size_t batch)
{
    return b ? l0_call(c, GECM_L0_SUB, a, b, r, NULL, batch) : GECM_ERR_ARG;
}
int gecm_vecaddsubmod(gecm_ctx *c, const void *a, const void *b, void *sum, void *diff, size_t batch)
{
    return (b && diff) ? l0_call(c, GECM_L0_ADDSUB, a, b, sum, diff, batch) : GECM_ERR_ARG;
}

/* ---- phase 0 -------------------------------------------------------------------------------- */
static int alloc_batch(gecm_ctx *c, size_t batch)
{
    free_batch(c);
    c->sigma = (uint64_t *)calloc(batch, sizeof(uint64_t));
    c->bad = (uint8_t *)calloc(batch, 1);
    c->hx = (uint32_t *)calloc(batch * (size_t)c->nl, 4);
    c->hz = (uint32_t *)calloc(batch * (size_t)c->nl, 4);
    c->hacc = (uint32_t *)calloc(batch * (size_t)c->nl, 4);
    c->hfail = (uint32_t *)calloc(batch * (size_t)c->nl, 4);
    c->fail_planes = 1;
    if (!c->sigma || !c->bad || !c->hx || !c->hz || !c->hacc || !c->hfail) { free_batch(c); return GECM_ERR_NOMEM; }
    c->batch = batch;
    if (gecm_dev_resize(c->dev, batch)) { set_err("%s", gecm_dev_error()); return GECM_ERR_DEVICE; }
    c->ff_pending = 0;
    c->ff_loaded = 0;
    if (c->dev_f && gecm_dev_resize(c->dev_f, batch)) { set_err("%s", gecm_dev_error()); return GECM_ERR_DEVICE; }
    return GECM_OK;
}

/* Suyama curve for one sigma up to (but not including) the two modular inversions
 * (ecm.c:1587-1641, 1717-1722): outputs x3 = u^3 mod n, z3 = v^3 mod n, num = (v-u)^3 (3u+v) mod n,
 * den = 16 u^3 v mod n. */
static void suyama_pre(const mpl_t *n, uint64_t sigma, mpl_t *x3, mpl_t *z3, mpl_t *num, mpl_t *den)
{
    mpl_t u, v, t1, t2, t3, t4;
    mpl_set_u64(&v, sigma);
    mpl_shl(&v, &v, 2);                 /* v = 4 sigma            ecm.c:1588-1589 */
    mpl_set_u64(&u, sigma);
    mpl_mul(&u, &u, &u);
    mpl_set_u64(&t1, 5);
    mpl_sub(&u, &u, &t1);               /* u = sigma^2 - 5        ecm.c:1596-1598 */
    mpl_mul(&t1, &u, &u);
    mpl_mul(&t1, &t1, &u);
    mpl_mod(x3, &t1, n);                /* x = u^3                ecm.c:1601-1603 */
    mpl_mul(&t1, &v, &v);
    mpl_mul(&t1, &t1, &v);
    mpl_mod(z3, &t1, n);                /* z = v^3                ecm.c:1607-1609 */
    /* (v - u) mod n                                               ecm.c:1615-1623 */
    mpl_t um, vm;
    mpl_mod(&um, &u, n);
    mpl_mod(&vm, &v, n);
    mpl_submod(&t1, &vm, &um, n);
    mpl_mulmod(&t2, &t1, &t1, n);
    mpl_mulmod(&t4, &t2, &t1, n);       /* (v-u)^3                ecm.c:1626-1629 */
    mpl_mul_u64(&t3, &u, 3);
    mpl_add(&t3, &t3, &v);
    mpl_mod(&t3, &t3, n);               /* 3u + v                 ecm.c:1632-1634 */
    mpl_mulmod(num, &t3, &t4, n);       /* a = (v-u)^3 (3u+v)     ecm.c:1637-1638 */
    mpl_mul_u64(&t2, x3, 16);
    mpl_mul(&t2, &t2, &v);
    mpl_mod(den, &t2, n);               /* 16 u^3 v               ecm.c:1718-1720 */
}

/* One worker's slice [lo, hi) of the batch: the whole Suyama construction for those curves.
 * The two inversions per curve, mpz_invert(16u^3v) ecm.c:1745 and mpz_invert(v^3) ecm.c:1759, share the
 * modulus, so each slice does Montgomery's simultaneous inversion: one extended Euclid per slice
 * instead of two per curve.  Inverses mod N are unique, so the values are the ones GMP returns. */
typedef struct {
    gecm_ctx *c;
    const uint64_t *sigma;
    size_t batch, lo, hi;
    uint32_t *hX, *hZ, *hS;
    uint32_t *fX, *fZ, *fS;   /* the same three values for the F-form context (NULL if unused) */
    int anybad, rc;
} build_job;

static void *build_slice(void *arg)
{
    build_job *j = (build_job *)arg;
    gecm_ctx *c = j->c;
    const size_t cnt = j->hi - j->lo, m = 2 * cnt, batch = j->batch;
    const int nl = c->nl;
    j->rc = 0;
    j->anybad = 0;
    if (cnt == 0) return NULL;
    mpl_t *x3 = (mpl_t *)malloc(cnt * sizeof(mpl_t) * 2);
    mpl_t *dens = (mpl_t *)malloc(m * sizeof(mpl_t));
    mpl_t *pref = (mpl_t *)malloc(m * sizeof(mpl_t));
    if (!x3 || !dens || !pref) { free(x3); free(dens); free(pref); j->rc = GECM_ERR_NOMEM; return NULL; }
    mpl_t *num = x3 + cnt;
    for (size_t i = 0; i < cnt; i++)
        suyama_pre(&c->N, j->sigma[j->lo + i], &x3[i], &dens[2 * i + 1], &num[i], &dens[2 * i]);
    int batch_ok = 1;
    pref[0] = dens[0];
    for (size_t i = 1; i < m; i++) mpl_mulmod(&pref[i], &pref[i - 1], &dens[i], &c->N);
    mpl_t inv, t;
    if (!mpl_invmod(&inv, &pref[m - 1], &c->N)) batch_ok = 0;
    mpl_t *invs = pref;   /* overwritten back to front */
    if (batch_ok) {
        for (size_t i = m - 1; i > 0; i--) {
            mpl_mulmod(&t, &inv, &pref[i - 1], &c->N);      /* dens[i]^-1 */
            mpl_mulmod(&inv, &inv, &dens[i], &c->N);
            invs[i] = t;
        }
        invs[0] = inv;
    } else {
        /* Some denominator shares a factor with N.  The reference ignores mpz_invert's return
         * value (ecm.c:1745, 1759); GMP leaves the destination untouched on failure, so the
         * reference goes on with the STALE operand: t2 = 16*u^3 (ecm.c:1718) in place of
         * (16u^3v)^-1 and t1 = (v-u)^3(3u+v) (ecm.c:1637) in place of (v^3)^-1.  Reproduced here so
         * that such curves still give the reference's residues bit for bit; the lane is also
         * flagged (a non-invertible denominator means gcd(denominator, N) is a factor). */
        for (size_t i = 0; i < m; i++)
            if (!mpl_invmod(&invs[i], &dens[i], &c->N)) {
                c->bad[j->lo + i / 2] = 1;
                j->anybad = 1;
                if ((i & 1) == 0) { mpl_mul_u64(&t, &x3[i / 2], 16); mpl_mod(&invs[i], &t, &c->N); }
                else invs[i] = num[i / 2];
            }
    }
    for (size_t i = 0; i < cnt; i++) {
        mpl_t A, X, Xm, Sm;
        const size_t k = j->lo + i;
        mpl_mulmod(&A, &num[i], &invs[2 * i], &c->N);          /* b = a / 16u^3v   ecm.c:1752-1753 */
        mpl_mulmod(&X, &x3[i], &invs[2 * i + 1], &c->N);       /* X = u^3 / v^3, Z = 1  ecm.c:1759-1761 */
        /* into Montgomery form (ecm.c:1763-1772), internal radix */
        mpl_mulmod(&Xm, &X, &c->rint_mod_n, &c->N);
        mpl_mulmod(&Sm, &A, &c->rint_mod_n, &c->N);
        mpl_to_limbs32(j->hX + k, batch, nl, LIMB_BITS, &Xm);
        mpl_to_limbs32(j->hZ + k, batch, nl, LIMB_BITS, &c->rint_mod_n);
        mpl_to_limbs32(j->hS + k, batch, nl, LIMB_BITS, &Sm);
        if (j->fX) {           /* plain residues mod N, lifted to Montgomery form modulo Mw = 2^k - 1 */
            mpl_mulmod(&Xm, &X, &c->ff_r_mod_m, &c->ff_M);
            mpl_mulmod(&Sm, &A, &c->ff_r_mod_m, &c->ff_M);
            mpl_to_limbs32(j->fX + k, batch, c->ff_nl, LIMB_BITS, &Xm);
            mpl_to_limbs32(j->fZ + k, batch, c->ff_nl, LIMB_BITS, &c->ff_r_mod_m);
            mpl_to_limbs32(j->fS + k, batch, c->ff_nl, LIMB_BITS, &Sm);
        }
    }
    free(x3); free(dens); free(pref);
    return NULL;
}

static int host_threads(void)
{
    /* worker threads for host-side batch work: GECM_HOST_THREADS, else min(8, online CPUs) */
    const char *e = getenv("GECM_HOST_THREADS");
    long n = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
    if (n < 1) n = 1;
    if (!e && n > 8) n = 8;
    if (n > 64) n = 64;
    return (int)n;
}

int gecm_build_curves(gecm_ctx *c, const uint64_t *sigma, size_t batch)
{
    if (!c || !sigma || batch == 0) { set_err("gecm_build_curves: bad argument"); return GECM_ERR_ARG; }
    /* inputs are checked before the context takes the new batch: after an error it holds no batch at all (the
     * phase functions then return GECM_ERR_STATE instead of running on memory nothing was uploaded to) */
    for (size_t i = 0; i < batch; i++)
        if (sigma[i] < 6) { set_err("gecm_build_curves: sigma[%zu] < 6", i); return GECM_ERR_ARG; }
    int rc = alloc_batch(c, batch);
    if (rc) return rc;
    memcpy(c->sigma, sigma, batch * sizeof(uint64_t));
    size_t words = (size_t)c->nl * batch;
    uint32_t *hX = (uint32_t *)calloc(words * 3, 4);
    if (!hX) { free_batch(c); return GECM_ERR_NOMEM; }
    const size_t fwords = c->dev_f ? (size_t)c->ff_nl * batch : 0;
    uint32_t *fX = fwords ? (uint32_t *)calloc(fwords * 3, 4) : NULL;
    if (fwords && !fX) { free(hX); free_batch(c); return GECM_ERR_NOMEM; }
    int nt = host_threads();
    if ((size_t)nt > batch / 256 + 1) nt = (int)(batch / 256 + 1);
    build_job jobs[64];
    pthread_t th[64];
    for (int t = 0; t < nt; t++) {
        jobs[t].c = c; jobs[t].sigma = sigma; jobs[t].batch = batch;
        jobs[t].lo = batch * (size_t)t / (size_t)nt;
        jobs[t].hi = batch * (size_t)(t + 1) / (size_t)nt;
        jobs[t].hX = hX; jobs[t].hZ = hX + words; jobs[t].hS = hX + 2 * words;
        jobs[t].fX = fX; jobs[t].fZ = fX ? fX + fwords : NULL; jobs[t].fS = fX ? fX + 2 * fwords : NULL;
    }
    for (int t = 1; t < nt; t++)
        if (pthread_create(&th[t], NULL, build_slice, &jobs[t])) { build_slice(&jobs[t]); th[t] = 0; }
    build_slice(&jobs[0]);
    int anybad = 0;
    for (int t = 0; t < nt; t++) {
        if (t > 0 && th[t]) pthread_join(th[t], NULL);
        if (jobs[t].rc) rc = jobs[t].rc;
        anybad |= jobs[t].anybad;
    }
    if (rc) { free(hX); free(fX); free_batch(c); return rc; }
    rc = gecm_dev_upload(c->dev, hX, hX + words, hX + 2 * words);
    if (!rc && fX) {
        rc = gecm_dev_upload(c->dev_f, fX, fX + fwords, fX + 2 * fwords);
        c->ff_loaded = !rc;
    }
    free(hX);
    free(fX);
    if (rc) { set_err("%s", gecm_dev_error()); free_batch(c); return GECM_ERR_DEVICE; }
    return anybad ? 1 : GECM_OK;
}

int gecm_upload_points(gecm_ctx *c, const void *X, const void *Z, const void *s, size_t batch)
{
    if (!c || !X || !Z || !s || batch == 0) { set_err("gecm_upload_points: bad argument"); return GECM_ERR_ARG; }
    int rc = alloc_batch(c, batch);
    if (rc) return rc;
    int nl = c->nl;
    size_t words = (size_t)nl * batch;
    uint32_t *h = (uint32_t *)calloc(words * 3, 4);
    if (!h) { free_batch(c); return GECM_ERR_NOMEM; }
    const void *src[3] = {X, Z, s};
    for (int k = 0; k < 3; k++)
        for (size_t i = 0; i < batch; i++) {
            mpl_t v;
            vec_get(c, &v, src[k], batch, i);
            if (mpl_cmp(&v, &c->N) >= 0) {
                free(h);
                free_batch(c);                 /* the context holds no batch after a rejected upload */
                set_err("gecm_upload_points: operand not < N");
                return GECM_ERR_ARG;
            }
            mpl_mulmod(&v, &v, &c->ref_to_int, &c->N);
            mpl_to_limbs32(h + (size_t)k * words + i, batch, nl, LIMB_BITS, &v);
        }
    rc = gecm_dev_upload(c->dev, h, h + words, h + 2 * words);
    if (!rc && c->dev_f) {
        /* x*Rint mod N -> x -> x*Rf mod Mw */
        const size_t fwords = (size_t)c->ff_nl * batch;
        uint32_t *f = (uint32_t *)calloc(fwords * 3, 4);
        mpl_t rinv;
        if (f && mpl_invmod(&rinv, &c->rint_mod_n, &c->N)) {
            for (int k = 0; k < 3; k++)
                for (size_t i = 0; i < batch; i++) {
                    mpl_t v;
                    mpl_from_limbs32(&v, h + (size_t)k * words + i, batch, nl, LIMB_BITS);
                    mpl_mulmod(&v, &v, &rinv, &c->N);
                    mpl_mulmod(&v, &v, &c->ff_r_mod_m, &c->ff_M);
                    mpl_to_limbs32(f + (size_t)k * fwords + i, batch, c->ff_nl, LIMB_BITS, &v);
                }
            c->ff_loaded = !gecm_dev_upload(c->dev_f, f, f + fwords, f + 2 * fwords);
        }
        free(f);
    }
    free(h);
    if (rc) { set_err("%s", gecm_dev_error()); free_batch(c); return GECM_ERR_DEVICE; }
    return GECM_OK;
}

/* ---- the F-form detour of stage 1 ------------------------------------------------------------
 * ff_settle: wait for the stage-1 kernel that ran modulo Mw = 2^k - 1, fetch its X, Z (canonical,
 * de-Montgomeryised), reduce them modulo N, put them back into Montgomery form modulo N and store them
 * in the main context as if stage 1 had run there.  Everything after stage 1 (save lines, factor scan,
 * stage 2) then works on residues modulo N as always. */
typedef struct {
    gecm_ctx *c;
    const uint32_t *fx, *fz;
    uint32_t *hX, *hZ;
    size_t lo, hi;
} settle_job;

static void *settle_slice(void *arg)
{
    settle_job *j = (settle_job *)arg;
    gecm_ctx *c = j->c;
    const size_t batch = c->batch;
    for (size_t i = j->lo; i < j->hi; i++) {
        mpl_t v;
        mpl_from_limbs32(&v, j->fx + i, batch, c->ff_nl, LIMB_BITS);
        mpl_mod(&v, &v, &c->N);
        mpl_mulmod(&v, &v, &c->rint_mod_n, &c->N);
        mpl_to_limbs32(j->hX + i, batch, c->nl, LIMB_BITS, &v);
        mpl_from_limbs32(&v, j->fz + i, batch, c->ff_nl, LIMB_BITS);
        mpl_mod(&v, &v, &c->N);
        mpl_mulmod(&v, &v, &c->rint_mod_n, &c->N);
        mpl_to_limbs32(j->hZ + i, batch, c->nl, LIMB_BITS, &v);
    }
    return NULL;
}

static int ff_settle(gecm_ctx *c)
{
    if (!c->ff_pending) return GECM_OK;
    c->ff_pending = 0;
    if (gecm_dev_sync(c->dev_f)) { set_err("%s", gecm_dev_error()); return GECM_ERR_DEVICE; }
    c->last_ms = gecm_dev_last_kernel_ms(c->dev_f);
    const size_t batch = c->batch, fwords = (size_t)c->ff_nl * batch, words = (size_t)c->nl * batch;
    uint32_t *f = (uint32_t *)calloc(2 * fwords + 2 * words, 4);
    if (!f) return GECM_ERR_NOMEM;
    if (gecm_dev_download_plain(c->dev_f, f, f + fwords)) { free(f); set_err("%s", gecm_dev_error()); return GECM_ERR_DEVICE; }
    int nt = host_threads();
    if ((size_t)nt > batch / 256 + 1) nt = (int)(batch / 256 + 1);
    settle_job jobs[64];
    pthread_t th[64];
    for (int t = 0; t < nt; t++) {
        jobs[t].c = c; jobs[t].fx = f; jobs[t].fz = f + fwords;
        jobs[t].hX = f + 2 * fwords; jobs[t].hZ = f + 2 * fwords + words;
        jobs[t].lo = batch * (size_t)t / (size_t)nt;
        jobs[t].hi = batch * (size_t)(t + 1) / (size_t)nt;
    }
    for (int t = 1; t < nt; t++)
        if (pthread_create(&th[t], NULL, settle_slice, &jobs[t])) { settle_slice(&jobs[t]); th[t] = 0; }
    settle_slice(&jobs[0]);
    for (int t = 1; t < nt; t++)
        if (th[t]) pthread_join(th[t], NULL);
    int rc = gecm_dev_upload_xz(c->dev, f + 2 * fwords, f + 2 * fwords + words);
    free(f);
    if (rc) { set_err("%s", gecm_dev_error()); return GECM_ERR_DEVICE; }
    return GECM_OK;
}

/* ---- phase 1 -------------------------------------------------------------------------------- */
static void *prefetch_run(void *arg)
{
    gecm_ctx *c = (gecm_ctx *)arg;
    c->pf_rc = gecm_tape_build_stage1_range(&c->pf_tape, c->pf_B1, c->pf_range, host_threads());
    return NULL;
}

/* the tape of ecm_stage1's call number `range` for bound B1 into c->tape: kept from the last call, taken from the
 * helper thread that compiled it while the device ran the range before, or compiled now */
static int tape_for(gecm_ctx *c, uint64_t B1, uint32_t range)
{
    if (c->tape.ops && c->tape_B1 == B1 && c->tape_range == range) return GECM_OK;
    int rc;
    if (c->pf_active) {
        pthread_join(c->pf_thread, NULL);
        c->pf_active = 0;
        if (!c->pf_rc && c->pf_B1 == B1 && c->pf_range == range) {
            gecm_tape_free(&c->tape);
            c->tape = c->pf_tape;
            memset(&c->pf_tape, 0, sizeof c->pf_tape);
            c->tape_B1 = B1; c->tape_range = range; c->tape_on_dev = 0;
            return GECM_OK;
        }
        gecm_tape_free(&c->pf_tape);
    }
    gecm_tape_free(&c->tape);
    rc = gecm_tape_build_stage1_range(&c->tape, B1, range, host_threads());
    if (rc) { set_err("gecm_stage1: tape build failed (%d)", rc); return rc == -1 ? GECM_ERR_NOMEM : GECM_ERR_STATE; }
    c->tape_B1 = B1; c->tape_range = range; c->tape_on_dev = 0;
    return GECM_OK;
}

int gecm_stage1_ranges(uint64_t B1) { return (int)gecm_stage1_ranges_u(B1); }

int gecm_stage1_describe_range(uint64_t B1, uint64_t B2, uint32_t range, gecm_stage1_range_desc *out)
{
    gecm_range_info ri;
    if (!out || B1 < 2 || B1 > GECM_B1_MAX) { set_err("gecm_stage1_describe_range: bad argument"); return GECM_ERR_ARG; }
    int rc = gecm_stage1_range_info(&ri, B1, B2, range);
    if (rc) { set_err("gecm_stage1_describe_range: %s", rc == -1 ? "out of memory" : "no such range"); return rc == -1 ? GECM_ERR_NOMEM : GECM_ERR_ARG; }
    out->lo = ri.lo; out->hi = ri.hi; out->nprimes = ri.nprimes; out->first_prime = ri.first_prime;
    out->last_prime = ri.last_prime; out->checkpoint = ri.exhausted;
    return GECM_OK;
}

int gecm_stage1_range(gecm_ctx *c, uint64_t B1, uint32_t range)
{
    if (c && c->ff_pending) { int rcs = ff_settle(c); if (rcs) return rcs; }
    if (!c || c->batch == 0) { set_err("gecm_stage1: no curves uploaded"); return GECM_ERR_STATE; }
    if (B1 < 2 || B1 > GECM_B1_MAX) { set_err("gecm_stage1: B1 must be in [2, %llu]", (unsigned long long)GECM_B1_MAX); return GECM_ERR_ARG; }
    const uint32_t nranges = gecm_stage1_ranges_u(B1);
    if (range >= nranges) { set_err("gecm_stage1_range: B1 = %llu has %u prime range(s)", (unsigned long long)B1, nranges); return GECM_ERR_ARG; }
    int rc = tape_for(c, B1, range);
    if (rc) return rc;
    if (!c->tape_on_dev) {
        /* stream-ordered after the kernel of the range before; returns when the copy is done */
        if (gecm_dev_set_tape(c->dev, c->tape.ops, c->tape.len)) { set_err("%s", gecm_dev_error()); return GECM_ERR_DEVICE; }
        c->tape_on_dev = 1;
    }
    if (range == 0) c->s1_ptadds = c->s1_ptdups = 0;
    c->s1_ptadds += c->tape.ptadds;
    c->s1_ptdups += c->tape.ptdups;
    c->s1_last_prime = c->tape.last_prime;
    c->s1_tape_len = c->tape.len;
    c->B1 = B1;
    c->have_plain = 0;
    c->have_acc = 0;
    c->s2_ready = 0;
    c->scan_valid[0] = c->scan_valid[1] = 0;
    /* For a batch small enough for the eight-lane layout (generic moduli only) that layout beats the special
     * multiply in its two-lane form: 1.5x against 1.4x at 15 limbs, 2.2-2.4x at 30-37 limbs. */
    const int want = c->lanes_per_curve ? c->lanes_per_curve : gecm_dev_auto_lanes(c->dev);
    const int small_batch = want == 8 || want == 32;
    if (c->dev_f && c->ff_on && c->ff_loaded && !small_batch) {
        /* N | 2^k - 1: run the chain modulo 2^k - 1 with the F-form multiply; ff_settle brings X, Z back */
        if (c->ff_tape_B1 != B1 || c->ff_tape_range != range) {
            if (gecm_dev_set_tape(c->dev_f, c->tape.ops, c->tape.len)) { set_err("%s", gecm_dev_error()); return GECM_ERR_DEVICE; }
            c->ff_tape_B1 = B1;
            c->ff_tape_range = range;
        }
        if (gecm_dev_stage1(c->dev_f, c->lanes_per_curve)) { set_err("%s", gecm_dev_error()); return GECM_ERR_DEVICE; }
        c->ff_pending = 1;
        c->last_on_f = 1;
    } else {
        c->last_on_f = 0;
        c->ff_loaded = 0;       /* the F-form copy of the points no longer matches */
        if (gecm_dev_stage1(c->dev, c->lanes_per_curve)) { set_err("%s", gecm_dev_error()); return GECM_ERR_DEVICE; }
    }
    if (range + 1 < nranges && !c->pf_active) {
        /* while the device runs this range: the next one's tape (2 s of host time per 1e8 primes on 8 threads) */
        c->pf_B1 = B1;
        c->pf_range = range + 1;
        c->pf_rc = 0;
        c->pf_active = pthread_create(&c->pf_thread, NULL, prefetch_run, c) == 0;
    }
    return GECM_OK;
}

int gecm_stage1(gecm_ctx *c, uint64_t B1)
{
    if (B1 < 2 || B1 > GECM_B1_MAX) { set_err("gecm_stage1: B1 must be in [2, %llu]", (unsigned long long)GECM_B1_MAX); return GECM_ERR_ARG; }
    const uint32_t nranges = gecm_stage1_ranges_u(B1);
    for (uint32_t r = 0; r < nranges; r++) {             /* ecm.c:1209-1234 */
        int rc = gecm_stage1_range(c, B1, r);
        if (rc) return rc;
    }
    return GECM_OK;
}

int gecm_set_special_form(gecm_ctx *c, int on)
{
    if (!c) return GECM_ERR_ARG;
    c->ff_on = on != 0;
    return GECM_OK;
}

int gecm_get_special_form(const gecm_ctx *c, int *k, int *limbs)
{
    if (!c) return GECM_ERR_ARG;
    if (k) *k = c->dev_f ? c->ff_k * c->ff_sign : 0;
    if (limbs) *limbs = c->dev_f ? c->ff_nl : 0;
    if (!c->dev_f || !c->ff_on) return 0;
    return c->last_on_f ? 2 : 1;
}

int gecm_set_lanes_per_curve(gecm_ctx *c, int lanes)
{
    if (!c || (lanes != 0 && lanes != 1 && lanes != 2 && lanes != 8 && lanes != 32)) {
        set_err("gecm_set_lanes_per_curve: lanes must be 0 (auto), 1, 2, 8 or 32");
        return GECM_ERR_ARG;
    }
    c->lanes_per_curve = lanes;
    return GECM_OK;
}

int gecm_get_lanes_per_curve(const gecm_ctx *c)
{
    if (!c) return GECM_ERR_ARG;
    return gecm_dev_last_lanes(c->last_on_f ? c->dev_f : c->dev);
}

int gecm_stage1_progress(const gecm_ctx *c, uint32_t *done, uint32_t *total)
{
    if (!c) return GECM_ERR_ARG;
    return gecm_dev_stage1_progress(c->last_on_f ? c->dev_f : c->dev, done, total) ? GECM_ERR_DEVICE : GECM_OK;
}

int gecm_last_kernel_name(const gecm_ctx *c, char *buf, size_t len)
{
    if (!c || !buf || !len) return GECM_ERR_ARG;
    snprintf(buf, len, "%s", gecm_dev_last_kernel(c->last_on_f ? c->dev_f : c->dev));
    return GECM_OK;
}

int gecm_sync(gecm_ctx *c)
{
    if (!c) return GECM_ERR_ARG;
    if (c->ff_pending) return ff_settle(c);
    if (gecm_dev_sync(c->dev)) { set_err("%s", gecm_dev_error()); return GECM_ERR_DEVICE; }
    c->last_ms = gecm_dev_last_kernel_ms(c->dev);
    return GECM_OK;
}

double gecm_last_kernel_ms(const gecm_ctx *c) { return c ? c->last_ms : 0.0; }

int gecm_get_stage1_stats(const gecm_ctx *c, gecm_stage1_stats *st)
{
    if (!c || !st || !c->tape.ops) return GECM_ERR_STATE;
    st->ptadds = c->s1_ptadds;
    st->ptdups = c->s1_ptdups;
    st->last_prime = c->s1_last_prime;
    st->tape_len = c->s1_tape_len;
    return GECM_OK;
}

int gecm_download_points(gecm_ctx *c, void *X, void *Z)
{
    if (c && c->ff_pending) { int rcs = ff_settle(c); if (rcs) return rcs; }
    if (!c || !X || !Z || c->batch == 0) return GECM_ERR_ARG;
    size_t batch = c->batch, words = (size_t)c->nl * batch;
    uint32_t *h = (uint32_t *)malloc(words * 2 * 4);
    if (!h) return GECM_ERR_NOMEM;
    if (gecm_dev_download_mont(c->dev, h, h + words)) { free(h); set_err("%s", gecm_dev_error()); return GECM_ERR_DEVICE; }
    void *dst[2] = {X, Z};
    for (int k = 0; k < 2; k++)
        for (size_t i = 0; i < batch; i++) {
            mpl_t v;
            mpl_from_limbs32(&v, h + (size_t)k * words + i, batch, c->nl, LIMB_BITS);
            mpl_mulmod(&v, &v, &c->int_to_ref, &c->N);
            vec_put(c, dst[k], batch, i, &v);
        }
    free(h);
    return GECM_OK;
}

static int fetch_plain(gecm_ctx *c)
{
    if (c->ff_pending) { int rcs = ff_settle(c); if (rcs) return rcs; }
    if (c->have_plain) return GECM_OK;
    if (c->batch == 0) { set_err("no batch"); return GECM_ERR_STATE; }
    if (gecm_dev_download_plain(c->dev, c->hx, c->hz)) { set_err("%s", gecm_dev_error()); return GECM_ERR_DEVICE; }
    c->have_plain = 1;
    return GECM_OK;
}

int gecm_download_points_plain(gecm_ctx *c, void *x, void *z)
{
    if (!c || !x || !z) return GECM_ERR_ARG;
    int rc = fetch_plain(c);
    if (rc) return rc;
    for (size_t i = 0; i < c->batch; i++) {
        mpl_t v;
        mpl_from_limbs32(&v, c->hx + i, c->batch, c->nl, LIMB_BITS);
        vec_put(c, x, c->batch, i, &v);
        mpl_from_limbs32(&v, c->hz + i, c->batch, c->nl, LIMB_BITS);
        vec_put(c, z, c->batch, i, &v);
    }
    return GECM_OK;
}

int gecm_format_save_line(gecm_ctx *c, size_t k, char *buf, size_t buflen)
{
    return gecm_format_resume_line(c, k, c ? c->B1 : 0, buf, buflen);
}

int gecm_format_resume_line(gecm_ctx *c, size_t k, uint64_t b1_label, char *buf, size_t buflen)
{
    if (!c || !buf || k >= c->batch) return GECM_ERR_ARG;
    int rc = fetch_plain(c);
    if (rc) return rc;
    static __thread char hn[MPL_MAXL * 10 + 2], hxs[MPL_MAXL * 10 + 2], hzs[MPL_MAXL * 10 + 2];
    mpl_t v;
    mpl_get_hex(hn, report_n(c));
    mpl_from_limbs32(&v, c->hx + k, c->batch, c->nl, LIMB_BITS);
    mpl_get_hex(hxs, &v);
    mpl_from_limbs32(&v, c->hz + k, c->batch, c->nl, LIMB_BITS);
    mpl_get_hex(hzs, &v);
    /* ecm.c:1372-1380 */
    int n = snprintf(buf, buflen, "METHOD=ECM; SIGMA=%llu; B1=%llu; N=0x%s; X=0x%s; Z=0x%s; PROGRAM=AVX-ECM;\n",
                     (unsigned long long)c->sigma[k], (unsigned long long)b1_label, hn, hxs, hzs);
    if (n < 0 || (size_t)n >= buflen) { set_err("gecm_format_save_line: buffer too small"); return GECM_ERR_ARG; }
    return n;
}

int gecm_stage1_factor(gecm_ctx *c, size_t k, char *dec, size_t declen, int *is_prp)
{
    if (!c || k >= c->batch) return GECM_ERR_ARG;
    int rc = fetch_plain(c);
    if (rc) return rc;
    mpl_t z, g;
    /* check_factor, ecm.c:2542-2557: gcd(Z, N); the reference passes Z in Montgomery form, and
     * gcd(z R mod N, N) = gcd(z, N) because R is a power of two and N is odd.  If the device scan
     * of this batch has run, its gcd is used; otherwise it is computed here. */
    if (c->scan_valid[0] && c->hg[0]) {
        mpl_from_limbs32(&g, c->hg[0] + k, c->batch, c->nl, LIMB_BITS);
    } else {
        mpl_from_limbs32(&z, c->hz + k, c->batch, c->nl, LIMB_BITS);
        mpl_gcd(&g, &z, &c->N);
    }
    to_report(c, &g);
    if (mpl_cmp_u64(&g, 1) > 0 && mpl_cmp(&g, report_n(c)) != 0) {
        static __thread char tmp[MPL_MAXL * 10 + 16];
        int n = mpl_get_dec(tmp, &g);
        if (dec && declen) {
            if ((size_t)n >= declen) { set_err("gecm_stage1_factor: buffer too small"); return GECM_ERR_ARG; }
            memcpy(dec, tmp, (size_t)n + 1);
        }
        if (is_prp) *is_prp = mpl_probab_prime(&g, 3);   /* ecm.c:1346 */
        return 1;
    }
    return 0;
}

/* ---- stage 2 -------------------------------------------------------------------------------- */
/* point additions next_pt_vec performs for multiplier c (one per bit below the top one, ecm.c:939-966) */
static uint64_t ladder_adds(uint64_t c)
{
    uint64_t n = 0;
    if (c <= 2) return 0;
    while (c > 1) { n++; c >>= 1; }
    return n;
}

int gecm_stage2_init(gecm_ctx *c, uint32_t D, uint32_t U)
{
    if (c && c->ff_pending) { int rcs = ff_settle(c); if (rcs) return rcs; }
    if (!c || c->batch == 0 || c->B1 == 0) { set_err("gecm_stage2_init: run stage 1 first"); return GECM_ERR_STATE; }
    if (!D) D = gecm_s2_default_D(c->B1);
    if (!U) U = GECM_S2_DEFAULT_U;
    /* the ring holds the 2L = 4U steps of the window plus one chunk being generated */
    if (U > (S2_RING - S2_GIANT_CHUNK) / 4) {
        set_err("gecm_stage2_init: U = %u is more than this build's giant-step ring takes (U <= %u)", U,
                (S2_RING - S2_GIANT_CHUNK) / 4);
        return GECM_ERR_ARG;
    }
    if (c->s2.D != D || c->s2.U != U) {
        gecm_s2_plan_free(&c->s2);
        if (gecm_s2_plan_init(&c->s2, D, U)) { set_err("gecm_stage2_init: bad D/U"); return GECM_ERR_ARG; }
    }
    c->have_acc = 0;
    c->s2_ptadds = (uint64_t)c->s2.umax - 2 + ladder_adds(D);   /* ecm.c:2263: j = 3..U*w; Pd ladder :2334 */
    c->s2_numinv = 1;                                        /* ecm.c:2322 */
    c->s2_devinv = (c->s2.npb - 1 + GECM_S2_BLK - 1) / GECM_S2_BLK;
    c->s2_paired = 0;
    /* small batches: K interleaved sub-sequences per curve (csrc/gecm_stage2.hpp, s2_init_k): the table indices of
     * the kept members of sub-sequence r, j = r, r+K, ... (r = 0: K, 2K, ...), in order */
    const uint32_t K = gecm_dev_s2_subseq(c->dev);
    uint32_t *tgt = NULL, toff[33];
    memset(toff, 0, sizeof toff);
    if (K > 1) {
        tgt = (uint32_t *)malloc(((size_t)c->s2.npb + 1) * sizeof(uint32_t));
        if (!tgt) return GECM_ERR_NOMEM;
        uint32_t n = 0;
        for (uint32_t r = 0; r < K; r++) {
            toff[r] = n;
            for (uint32_t j = r ? r : K; j <= c->s2.umax; j += K)
                if (c->s2.map[j]) tgt[n++] = c->s2.map[j];
        }
        toff[K] = n;
    }
    /* the host copy of the failure planes is sized before the device is touched: after an allocation failure the
     * context still has its old stage-2 state, untouched */
    const uint32_t planes = K > 1 ? K + 1 : 1;
    if (planes != c->fail_planes || !c->hfail) {
        uint32_t *nf = (uint32_t *)calloc(c->batch * (size_t)c->nl * planes, 4);
        if (!nf) { free(tgt); return GECM_ERR_NOMEM; }
        free(c->hfail);
        c->hfail = nf;
        c->fail_planes = planes;
        c->have_acc = 0;
    }
    c->s2_ready = 0;
    int drc = gecm_dev_s2_init(c->dev, c->s2.keep, c->s2.keep_words, c->s2.umax, D, c->s2.npb, S2_GIANT_CHUNK, S2_RING, tgt,
                               toff, K);
    free(tgt);
    if (drc) {
        set_err("gecm_stage2_init: %s", gecm_dev_error());
        return GECM_ERR_DEVICE;
    }
    if (gecm_dev_s2_fail_planes(c->dev) != planes) { set_err("gecm_stage2_init: failure planes out of step"); return GECM_ERR_STATE; }
    c->s2_ready = 1;
    return GECM_OK;
}

int gecm_pair_primes(gecm_pairs *out, uint64_t B1, uint64_t B2, uint32_t D, uint32_t U)
{
    gecm_pairmap pm;
    if (!out || B2 <= B1 || !D || !U) { set_err("gecm_pair_primes: bad argument"); return GECM_ERR_ARG; }
    if (gecm_pair(&pm, B1, B2, D, U)) { set_err("gecm_pair_primes: out of memory"); return GECM_ERR_NOMEM; }
    out->pairmap_v = pm.v; out->pairmap_u = pm.u; out->steps = pm.steps; out->amin = pm.amin;
    out->pairs = pm.pairs; out->primes = pm.nump;
    return GECM_OK;
}

void gecm_pairmap_release(gecm_pairs *p)
{
    if (!p) return;
    free(p->pairmap_v);
    free(p->pairmap_u);
    memset(p, 0, sizeof *p);
}

/* launch the giant steps and the pair walk of one range from a built tape; tape_id != 0: the device keeps its copy */
static int s2_run_tape(gecm_ctx *c, const gecm_s2_tape *t, uint32_t amin, uint64_t tape_id)
{
    const gecm_s2_plan *p = &c->s2;
    const uint64_t A0 = (uint64_t)amin * p->D * 2;                       /* ecm.c:2378 */
    int rc = gecm_dev_s2_pair(c->dev, t->words, (uint32_t)(t->nwords / 2), p->D, S2_GIANT_CHUNK, S2_RING, A0, tape_id);
    if (rc) { set_err("gecm_stage2_pair: %s", gecm_dev_error()); return GECM_ERR_DEVICE; }
    c->s2_ptadds += t->adds + ladder_adds(A0) + ladder_adds(A0 - p->D);  /* ecm.c:2383, 2390 */
    c->s2_numinv += t->inv; c->s2_paired += t->paired; c->s2_devinv += t->devinv;
    c->s2_amin_last = t->amin_last;
    c->have_acc = 0;
    c->scan_valid[1] = 0;
    return GECM_OK;
}

static int s2_build_tape(gecm_ctx *c, gecm_s2_tape *t, uint32_t steps, const uint32_t *pm_v, const uint32_t *pm_u, uint32_t amin)
{
    uint32_t bad = 0;
    int trc = gecm_s2_tape_build(t, &c->s2, steps, pm_v, pm_u, amin, S2_GIANT_CHUNK, S2_RING, &bad);
    if (trc == -1) return GECM_ERR_NOMEM;
    if (trc) {                                                           /* ecm.c:2508-2517 */
        set_err("gecm_stage2_pair: invalid pair map entry %u: (%u,%u)", bad, pm_v[bad], pm_u[bad]);
        return GECM_ERR_ARG;
    }
    return GECM_OK;
}

/* Two independent 64-bit hashes of the pair map (FNV-1a over the words; a multiply-rotate mix over the 64-bit
 * pairs (v, u) with their position), 8 ms for the 3.0 M entries of a 1e8 range */
static void pairmap_fingerprint(uint32_t steps, const uint32_t *pm_v, const uint32_t *pm_u, uint64_t *h1, uint64_t *h2)
{
    uint64_t a = 1469598103934665603ull, b = 0x9E3779B97F4A7C15ull;
    for (uint32_t i = 0; i < steps; i++) {
        a = (a ^ pm_v[i]) * 1099511628211ull;
        a = (a ^ pm_u[i]) * 1099511628211ull;
        uint64_t w = (((uint64_t)pm_v[i] << 32) | pm_u[i]) + (uint64_t)i * 0xD6E8FEB86659FD93ull;
        w *= 0xBF58476D1CE4E5B9ull;
        b = ((b << 27) | (b >> 37)) ^ w;
        b *= 0x94D049BB133111EBull;
    }
    *h1 = a;
    *h2 = b;
}

/* The tape of a pair map is the same for every batch and costs more host time than reading the map once (130 ms
 * against 8 ms): the last one is kept and recognised by (steps, amin, D, U) and both hashes of the map. */
static int s2_tape_for(gecm_ctx *c, uint32_t steps, const uint32_t *pm_v, const uint32_t *pm_u, uint32_t amin)
{
    uint64_t fp, fp2;
    pairmap_fingerprint(steps, pm_v, pm_u, &fp, &fp2);
    if (c->ptp_valid && c->ptp_steps == steps && c->ptp_amin == amin && c->ptp_D == c->s2.D && c->ptp_U == c->s2.U &&
        c->ptp_fp == fp && c->ptp_fp2 == fp2)
        return GECM_OK;
    if (c->ptp_valid) free(c->ptp.words);
    c->ptp_valid = 0;
    int rc = s2_build_tape(c, &c->ptp, steps, pm_v, pm_u, amin);
    if (rc) return rc;
    c->ptp_valid = 1;
    c->ptp_fp = fp; c->ptp_fp2 = fp2;
    c->ptp_steps = steps; c->ptp_amin = amin; c->ptp_D = c->s2.D; c->ptp_U = c->s2.U;
    c->ptp_id = next_tape_id();
    return GECM_OK;
}

int gecm_stage2_pair(gecm_ctx *c, uint32_t steps, const uint32_t *pm_v, const uint32_t *pm_u, uint32_t amin)
{
    if (!c || !c->s2_ready) { set_err("gecm_stage2_pair: gecm_stage2_init has not run"); return GECM_ERR_STATE; }
    if (steps && (!pm_v || !pm_u)) return GECM_ERR_ARG;
    int rc = s2_tape_for(c, steps, pm_v, pm_u, amin);
    if (rc) return rc;
    return s2_run_tape(c, &c->ptp, amin, c->ptp_id);
}

/* Optional, for callers of the phase functions: make (and keep) the tape gecm_stage2_pair will need for this pair map
 * and (D, U) ahead of time — e.g. while the device runs stage 1.  Touches nothing on the device. */
int gecm_stage2_pair_prepare(gecm_ctx *c, uint32_t D, uint32_t U, uint32_t steps, const uint32_t *pm_v, const uint32_t *pm_u,
                             uint32_t amin)
{
    if (!c || (steps && (!pm_v || !pm_u))) return GECM_ERR_ARG;
    if (!D || !U || U > (S2_RING - S2_GIANT_CHUNK) / 4) { set_err("gecm_stage2_pair_prepare: bad D/U"); return GECM_ERR_ARG; }
    if (c->s2.D != D || c->s2.U != U) {                                  /* the plan gecm_stage2_init(D, U) would make */
        gecm_s2_plan_free(&c->s2);
        c->s2_ready = 0;
        if (gecm_s2_plan_init(&c->s2, D, U)) { set_err("gecm_stage2_pair_prepare: bad D/U"); return GECM_ERR_ARG; }
    }
    return s2_tape_for(c, steps, pm_v, pm_u, amin);
}

/* keep `pm` (ownership passes to the context) and make the tape that goes with it; the plan of (D, U) must be c->s2 */
static int keep_pairmap(gecm_ctx *c, gecm_pairs *pm, uint64_t lo, uint64_t hi)
{
    drop_kept_pairmap(c);
    c->pm = *pm;
    c->pm_valid = 1; c->pm_lo = lo; c->pm_hi = hi; c->pm_D = c->s2.D; c->pm_U = c->s2.U;
    int rc = s2_build_tape(c, &c->tp, c->pm.steps, c->pm.pairmap_v, c->pm.pairmap_u, c->pm.amin);
    if (rc) return rc;                                                   /* the map stays; the tape is made (and refused) again later */
    c->tp_valid = 1;
    c->tp_id = next_tape_id();
    return GECM_OK;
}

/* The pair map of [B1, B2) and the device tape made from it depend on nothing the device computes: a caller can have
 * them made while stage 1 runs (gecm_stage1 returns after the launch).  Kept in the context; gecm_stage2 with the same
 * (B2, D, U) finds them, and the device keeps its copy of the tape from one batch to the next. */
int gecm_stage2_prepare(gecm_ctx *c, uint64_t B2, uint32_t D, uint32_t U)
{
    const uint64_t PRIME_RANGE = 100000000ull;
    if (!c || c->B1 == 0 || B2 <= c->B1) { set_err("gecm_stage2_prepare: call gecm_stage1 first; B2 > B1"); return GECM_ERR_ARG; }
    if (!D) D = gecm_s2_default_D(c->B1);
    if (!U) U = GECM_S2_DEFAULT_U;
    if (B2 - c->B1 > PRIME_RANGE) return GECM_OK;                        /* several ranges: made range by range later */
    if (U > (S2_RING - S2_GIANT_CHUNK) / 4) return GECM_OK;              /* gecm_stage2_init will refuse it */
    if (c->pm_valid && c->pm_lo == c->B1 && c->pm_hi == B2 && c->pm_D == D && c->pm_U == U) return GECM_OK;
    if (c->s2.D != D || c->s2.U != U) {                                  /* the plan gecm_stage2_init(D, U) would make */
        gecm_s2_plan_free(&c->s2);
        c->s2_ready = 0;
        if (gecm_s2_plan_init(&c->s2, D, U)) { set_err("gecm_stage2_prepare: bad D/U"); return GECM_ERR_ARG; }
    }
    gecm_pairs pm;
    int rc = gecm_pair_primes(&pm, c->B1, B2, D, U);
    if (rc) return rc;
    return keep_pairmap(c, &pm, c->B1, B2);
}

int gecm_stage2(gecm_ctx *c, uint64_t B2, uint32_t D, uint32_t U)
{
    const uint64_t PRIME_RANGE = 100000000ull;                           /* main.c:581 */
    if (!c || c->B1 == 0 || B2 <= c->B1) { set_err("gecm_stage2: need B2 > B1 and a finished stage 1"); return GECM_ERR_ARG; }
    int rc = gecm_stage2_init(c, D, U);
    if (rc) return rc;
    for (uint64_t p = c->B1; p < B2; p += PRIME_RANGE) {                 /* ecm.c:1424-1476 */
        uint64_t hi = p + PRIME_RANGE < B2 ? p + PRIME_RANGE : B2;
        /* the pair map depends on (range, D, U) only: a run of many batches (the reference: one per 8 curves and
         * thread) computes it once; the last single-range map is kept in the context, with its tape */
        const int cacheable = (p == c->B1 && hi == B2);
        const int kept = c->pm_valid && c->pm_lo == p && c->pm_hi == hi && c->pm_D == c->s2.D && c->pm_U == c->s2.U;
        if (cacheable && !kept) {
            gecm_pairs pm;
            rc = gecm_pair_primes(&pm, p, hi, c->s2.D, c->s2.U);
            if (rc) return rc;
            rc = keep_pairmap(c, &pm, p, hi);
            if (rc) return rc;
        }
        if (cacheable) {
            if (!c->tp_valid) {                                          /* a map kept without its tape: make it now */
                gecm_pairs pm = c->pm;
                c->pm_valid = 0;
                rc = keep_pairmap(c, &pm, p, hi);
                if (rc) return rc;
            }
            rc = s2_run_tape(c, &c->tp, c->pm.amin, c->tp_id);
            if (rc) return rc;
            continue;
        }
        gecm_pairs pm;
        rc = gecm_pair_primes(&pm, p, hi, c->s2.D, c->s2.U);
        if (rc) return rc;
        rc = gecm_stage2_pair(c, pm.steps, pm.pairmap_v, pm.pairmap_u, pm.amin);
        gecm_pairmap_release(&pm);
        if (rc) return rc;
    }
    return gecm_sync(c);
}

int gecm_get_stage2_stats(const gecm_ctx *c, gecm_stage2_stats *st)
{
    if (!c || !st || !c->s2.D) return GECM_ERR_STATE;
    st->ptadds = c->s2_ptadds; st->numinv = c->s2_numinv; st->paired = c->s2_paired;
    st->device_inversions = c->s2_devinv;
    st->D = c->s2.D; st->U = c->s2.U; st->L = c->s2.L; st->amin_last = c->s2_amin_last;
    return GECM_OK;
}

static int fetch_acc(gecm_ctx *c)
{
    if (c->have_acc) return GECM_OK;
    if (!c->s2_ready) { set_err("no stage-2 state"); return GECM_ERR_STATE; }
    if (gecm_dev_s2_download(c->dev, c->hacc, c->hfail)) { set_err("%s", gecm_dev_error()); return GECM_ERR_DEVICE; }
    c->have_acc = 1;
    return GECM_OK;
}

/* The failed-inversion record of curve k.  The reference overwrites its accumulator with gcd(product of the batch, N)
 * every time a batch inversion fails (ecm.c:1925-1939): what its scan finds in the end is the gcd of the LAST failing
 * batch (times later cross products).  Plane 0 holds that gcd for the single-chain inversions — after gecm_stage2_pair
 * the last chunk of the range, cut to be exactly the reference's last batch — and decides when it holds one.  Otherwise
 * the sub-sequences' planes stand for one batch inverted in K pieces: the gcd of N with the PRODUCT of their records
 * is the gcd of the whole batch's product (for a product that covers N that is N itself — "no factor", which is
 * what the reference finds then too: its batch product is 0 modulo N).  Every record is passed through gcd(., N)
 * first: what comes out divides N. */
static void fail_record(gecm_ctx *c, size_t k, mpl_t *g)
{
    const size_t plane = c->batch * (size_t)c->nl;
    mpl_t t, prod, gp;
    mpl_from_limbs32(&t, c->hfail + k, c->batch, c->nl, LIMB_BITS);
    if (!mpl_is_zero(&t)) { mpl_gcd(g, &t, &c->N); return; }
    mpl_set_u64(g, 0);
    if (c->fail_planes <= 1) return;
    mpl_set_u64(&prod, 0);
    for (uint32_t p = 1; p < c->fail_planes; p++) {
        mpl_from_limbs32(&t, c->hfail + p * plane + k, c->batch, c->nl, LIMB_BITS);
        if (mpl_is_zero(&t)) continue;
        mpl_gcd(&gp, &t, &c->N);
        if (mpl_is_zero(&prod)) prod = gp;
        else mpl_mulmod(&prod, &prod, &gp, &c->N);
        if (mpl_is_zero(&prod)) { prod = c->N; break; }       /* the product covers N: gcd = N, "no factor" */
    }
    if (!mpl_is_zero(&prod)) mpl_gcd(g, &prod, &c->N);
}

int gecm_download_acc(gecm_ctx *c, void *acc)
{
    if (!c || !acc) return GECM_ERR_ARG;
    int rc = fetch_acc(c);
    if (rc) return rc;
    for (size_t i = 0; i < c->batch; i++) {
        mpl_t v;
        mpl_from_limbs32(&v, c->hacc + i, c->batch, c->nl, LIMB_BITS);
        mpl_mulmod(&v, &v, &c->int_to_ref, &c->N);
        vec_put(c, acc, c->batch, i, &v);
    }
    return GECM_OK;
}

int gecm_stage2_factor(gecm_ctx *c, size_t k, char *dec, size_t declen, int *is_prp)
{
    if (!c || k >= c->batch) return GECM_ERR_ARG;
    int rc = fetch_acc(c);
    if (rc) return rc;
    mpl_t a, g;
    fail_record(c, k, &g);
    if (mpl_is_zero(&g)) {
        if (c->scan_valid[1] && c->hg[1]) {
            mpl_from_limbs32(&g, c->hg[1] + k, c->batch, c->nl, LIMB_BITS);
        } else {
            mpl_from_limbs32(&a, c->hacc + k, c->batch, c->nl, LIMB_BITS);
            mpl_gcd(&g, &a, &c->N);                  /* check_factor, ecm.c:2542-2557 */
        }
    }
    to_report(c, &g);
    if (mpl_cmp_u64(&g, 1) > 0 && mpl_cmp(&g, report_n(c)) != 0) {
        static __thread char tmp[MPL_MAXL * 10 + 16];
        int n = mpl_get_dec(tmp, &g);
        if (dec && declen) {
            if ((size_t)n >= declen) { set_err("gecm_stage2_factor: buffer too small"); return GECM_ERR_ARG; }
            memcpy(dec, tmp, (size_t)n + 1);
        }
        if (is_prp) *is_prp = mpl_probab_prime(&g, 3);
        return 1;
    }
    return 0;
}

/* ---- device factor scan ------------------------------------------------------------------- */
int gecm_scan_factors(gecm_ctx *c, int stage, size_t *first)
{
    if (c && c->ff_pending) { int rcs = ff_settle(c); if (rcs) return rcs; }
    if (!c || c->batch == 0 || (stage != 1 && stage != 2)) { set_err("gecm_scan_factors: bad argument"); return GECM_ERR_ARG; }
    if (stage == 2 && !c->s2_ready) { set_err("gecm_scan_factors: no stage-2 state"); return GECM_ERR_STATE; }
    uint32_t **f = &c->flags[stage - 1], **hg = &c->hg[stage - 1];
    if (!*f) *f = (uint32_t *)calloc(c->batch, sizeof(uint32_t));
    if (!*hg) *hg = (uint32_t *)calloc(c->batch * (size_t)c->nl, sizeof(uint32_t));
    if (!*f || !*hg) return GECM_ERR_NOMEM;
    if (gecm_dev_gcd_scan(c->dev, stage - 1, *f, *hg)) { set_err("gecm_scan_factors: %s", gecm_dev_error()); return GECM_ERR_DEVICE; }
    size_t n = 0, lo = c->batch;
    if (stage == 1) {
        /* x, z come to the host with the scan: everything a caller does next (save lines, factors of the flagged
         * curves) is then host work, off the device's queue */
        int rc = fetch_plain(c);
        if (rc) return rc;
    }
    if (stage == 2) {
        /* a failed batch inversion also marks its curve (ecm.c:1927-1939) */
        int rc = fetch_acc(c);
        if (rc) return rc;
        for (size_t k = 0; k < c->batch; k++) {
            mpl_t g;
            fail_record(c, k, &g);
            if (!mpl_is_zero(&g)) {
                to_report(c, &g);
                (*f)[k] = (mpl_cmp_u64(&g, 1) > 0 && mpl_cmp(&g, report_n(c)) != 0);
            }
        }
    }
    if (c->have_report)
        /* the device looked for factors of the context's modulus: keep the curves whose gcd shares one with the
         * report modulus (gcd(gcd(v, Mw), N) = gcd(v, N) for N | Mw) */
        for (size_t k = 0; k < c->batch; k++)
            if ((*f)[k]) {
                mpl_t g;
                mpl_from_limbs32(&g, *hg + k, c->batch, c->nl, LIMB_BITS);
                if (mpl_is_zero(&g)) continue;            /* flagged by a failure record, settled above */
                to_report(c, &g);
                if (!(mpl_cmp_u64(&g, 1) > 0 && mpl_cmp(&g, report_n(c)) != 0)) {
                    /* stage 2: the failure record decides if there is one */
                    mpl_t fr;
                    mpl_set_u64(&fr, 0);
                    if (stage == 2) fail_record(c, k, &fr);
                    if (mpl_is_zero(&fr)) (*f)[k] = 0;
                }
            }
    for (size_t k = 0; k < c->batch; k++)
        if ((*f)[k]) { n++; if (k < lo) lo = k; }
    c->scan_valid[stage - 1] = 1;
    if (first) *first = lo;
    return (int)(n > 0x7fffffff ? 0x7fffffff : n);
}

int gecm_curve_flag(const gecm_ctx *c, int stage, size_t k)
{
    if (!c || (stage != 1 && stage != 2) || k >= c->batch || !c->flags[stage - 1]) return 0;
    return (int)c->flags[stage - 1][k];
}

/* the hash of the host sources this object was compiled from (Makefile: H_SHA); gecm_version() compares them */
#ifdef GECM_MANIFEST_FN
const char *GECM_MANIFEST_FN(void) { return GECM_MANIFEST; }
#endif
