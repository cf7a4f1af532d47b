/* gecm.h — public C ABI of libgecm, the MI355X-native engine for the hot path of bbuhrow/avx-ecm.
 *
 * Drop-in boundary.  The reference reaches its hot path through
 *   (L0) five global function pointers bound in main.c:642-702 and declared in avx_ecm.h:205-209:
 *          vecmulmod_ptr / vecsqrmod_ptr / vecaddmod_ptr / vecsubmod_ptr / vecaddsubmod_ptr
 *   (L1) four per-thread phase functions dispatched by vececm through tpool_go
 *        (ecm.c:1195, 1234, 1407, 1460):
 *          ecm_build_curve_work_fcn  ecm.c:201-246   -> build_one_curve ecm.c:1548-1803
 *          ecm_stage1_work_fcn       ecm.c:167-176   -> ecm_stage1      ecm.c:1806-1854
 *          ecm_stage2_init_work_fcn  ecm.c:178-186   -> ecm_stage2_init ecm.c:2201-2340
 *          ecm_stage2_work_fcn       ecm.c:188-199   -> ecm_stage2_pair ecm.c:2342-2540
 * This header exports exactly those seams, on a batch of B curves instead of VECLEN=8/16 lanes.
 * Every entry point is extern "C", takes plain pointers and sizes, returns an int status
 * (0 = ok, < 0 = error; text from gecm_last_error()), and never exits the process (the reference
 * printf+exit()s: util.c:56-59, ecm.c:969-970).
 *
 * Vector operands ("vec") use the REFERENCE's memory layout (avx_ecm.h:111-116, main.c:117-138):
 *   data[lane + limb * batch], limb-major / curve-minor, limbs of DIGITBITS bits held in
 *   uint64_t (DIGITBITS = 52) or uint32_t (DIGITBITS = 32), NWORDS limbs per value, values in the
 *   reference's Montgomery form x * 2^(DIGITBITS*NWORDS) mod N, canonical in [0, N).
 * So a maintainer can pass bignum->data straight through (with batch = VECLEN) — INTEGRATION.md.
 *
 * Ownership: the context owns all device memory; every host array is caller-owned and copied.
 * Threading: one context per GPU, used by one host thread at a time; contexts are independent.
 */
#ifndef GECM_H
#define GECM_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct gecm_ctx gecm_ctx;

#define GECM_OK 0
#define GECM_ERR_ARG (-2)
#define GECM_ERR_DEVICE (-1)
#define GECM_ERR_NOMEM (-3)
#define GECM_ERR_STATE (-4)

const char *gecm_last_error(void);
int gecm_device_count(void);
const char *gecm_version(void);

/* ---- configuration (replaces monty_alloc + main.c:465-483, 597-640) ------------------------
 * n_str: the number to factor, decimal or 0x-hex (odd, > 1).
 * digitbits: 52 or 32 — the reference limb format used at THIS boundary (vec operands,
 *            NWORDS rule: smallest multiple of 208 (resp. 128) bits strictly greater than
 *            bitlen(N), main.c:465-483).  The device arithmetic is the same either way.
 * device: HIP device ordinal.                                                                 */
int gecm_create(gecm_ctx **out, int device, const char *n_str, int digitbits);
void gecm_destroy(gecm_ctx *ctx);

typedef struct {
    int digitbits;   /* 52 | 32                                   (avx_ecm.h:65-93)   */
    int nwords;      /* NWORDS                                    (main.c:482)        */
    int maxbits;     /* MAXBITS = DIGITBITS * NWORDS                                  */
    int nbits;       /* bitlen(N)                                                     */
    int dev_limbs;   /* 28-bit limbs per residue on the device                         */
    int device;
    uint64_t rho;    /* -N^-1 mod 2^DIGITBITS  (monty->vrho, main.c:636-640)           */
} gecm_config;
int gecm_get_config(const gecm_ctx *ctx, gecm_config *cfg);
int gecm_device_name(gecm_ctx *ctx, char *buf, size_t len);
/* free and total device memory right now (hipMemGetInfo), and the device bytes a batch of `curves` curves takes (stage 1
 * only, or with the stage-2 tables of wheel D and height U; 0 = the defaults for B1): what a caller sizes its batches
 * with.  The reference has no counterpart (its tables are per thread, ecm_work_init).                                */
int gecm_device_memory(gecm_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);
uint64_t gecm_batch_bytes(const gecm_ctx *ctx, size_t curves, int with_stage2, uint64_t B1, uint32_t D, uint32_t U);
/* "one" = R mod N in the reference limb format, single value (monty->one, main.c:633-634) */
int gecm_get_one(const gecm_ctx *ctx, void *one_limbs);

/* ---- L0: the five vector operators (avx_ecm.h:205-209) -------------------------------------
 * Test-level exports: same semantics as vecmulmod52/vecsqrmod52/vecaddmod52/vecsubmod52/
 * vec_simul_addsub52 (vecarith52.c:2438, 3317, 4550, 4684, 4877) and their 32-bit twins
 * (vecarith.c:221, 889, 2806, 2870, 2726): inputs canonical, outputs canonical, Montgomery
 * radix 2^(DIGITBITS*NWORDS).  `c` may alias `a` or `b`.  batch >= 1, any size.              */
int gecm_vecmulmod(gecm_ctx *ctx, const void *a, const void *b, void *c, size_t batch);
int gecm_vecsqrmod(gecm_ctx *ctx, const void *a, void *c, size_t batch);
int gecm_vecaddmod(gecm_ctx *ctx, const void *a, const void *b, void *c, size_t batch);
int gecm_vecsubmod(gecm_ctx *ctx, const void *a, const void *b, void *c, size_t batch);
int gecm_vecaddsubmod(gecm_ctx *ctx, const void *a, const void *b, void *sum, void *diff, size_t batch);

/* ---- L1 phase 0: curve construction (build_one_curve, ecm.c:1548-1803) ---------------------
 * Suyama parametrisation from sigma[0..batch): the context computes X, Z (=1), s = (A+2)/4 on the
 * host and uploads them.  sigma values must be >= 6 (the reference redraws below 6,
 * ecm.c:1564-1570).  Returns GECM_OK, or 1 if some curve's setup inversion failed because
 * gcd(denominator, N) > 1; such lanes continue exactly as the reference does (it ignores
 * mpz_invert's return value, ecm.c:1745, 1759, and goes on with the stale operand).           */
int gecm_build_curves(gecm_ctx *ctx, const uint64_t *sigma, size_t batch);
/* Alternative phase 0: caller supplies P=(X,Z) and s as vec operands (reference layout and
 * Montgomery radix), e.g. the output of the reference's own build_one_curve.                 */
int gecm_upload_points(gecm_ctx *ctx, const void *X, const void *Z, const void *s, size_t batch);

/* ---- input preparation (main.c:393-527) -----------------------------------------------------
 * What the reference's main() does to its first argument before any curve is built: evaluate the
 * expression (calc.c), recognise N | 2^k - 1, N | 2^k + 1 or 2^k = c (mod N) with c below one limb
 * (main.c:405-441), and for the first two replace N by gcd(N, primitive part of 2^k -/+ 1)
 * (find_primitive_factor, main.c:187-352, 445-457).  n_dec receives the decimal N the run is made on;
 * log receives the lines the reference prints on the way ("gen: ...", "removing algebraic ...",
 * "commencing parallel ecm on ...", "Mersenne input ... determined to be faster by REDC"), byte for byte.
 * When the reference would fold modulo Mw = 2^k -/+ 1 or 2^k - c (ref_special_reduction = 1; main.c:505-527, 642-684,
 * vecarith52.c:284-2436) it works modulo Mw THROUGHOUT — curve construction included — and its files hold residues
 * modulo Mw next to "N=" the number given, in which it also looks for factors (ecm.c:1111-1118).  To write those
 * files byte for byte, create the context on Mw and name N with gecm_set_report_modulus (the command-line driver does;
 * nine reference runs in tests/golden/special.json).  A context created on N itself gives the residues modulo N of a
 * run modulo N — with the cheaper special-form multiply where it pays, gecm_set_special_form — which is what a
 * caller wants who is not after the reference's bytes.                                                              */
typedef struct {
    int form;                   /* the reference's isMersenne: 0, +1 (2^k - 1), -1 (2^k + 1), else c of 2^k - c */
    int k;
    uint64_t c;
    int nbits;                  /* bit length of the prepared N */
    int ref_special_reduction;  /* 1 = the reference would not use REDC for this input (main.c:505-527) */
} gecm_input_info;
int gecm_prepare_input(const char *expr, int digitbits, char *n_dec, size_t n_len, gecm_input_info *info,
                       char *log, size_t loglen);
/* The size the reference prints next to a factor ("found PRP45 factor ...", ecm.c:1346-1366, 1494-1520) is
 * mpz_sizeinbase(f, 10) = floor(bits * log10 2) + 1, which is the digit count or one more (a 12-digit
 * factor of 40 bits is labelled C13).  This returns that number for a decimal string.                 */
int gecm_sizeinbase10(const char *dec);
/* The number the save lines name ("N=0x...") and factors are reported of, when it is not the context's modulus but a
 * divisor of it: the reference's gmpn = vnhat for special-form inputs (ecm.c:1111-1118).  n_str decimal or 0x-hex, must
 * divide the modulus the context was created on; NULL restores the default.  gecm_stage1_factor, gecm_stage2_factor and
 * gecm_scan_factors then report gcd(value, n) as the reference's check_factor(value, gmpn) does.                      */
int gecm_set_report_modulus(gecm_ctx *ctx, const char *n_str);

/* ---- L1 phase 1: stage 1 (ecm_stage1, ecm.c:1806-1854) -------------------------------------
 * P <- [prod of prime powers < B1] P for every curve of the batch.  Asynchronous: returns after
 * the (last) launch; gecm_sync waits.  For B1 > 10^8 this is the whole loop of ecm.c:1209-1234:
 * gecm_stage1_range for range 0, 1, ... in turn.                                               */
int gecm_stage1(gecm_ctx *ctx, uint64_t B1);
int gecm_sync(gecm_ctx *ctx);
/* Stage 1 above one prime range.  vececm re-sieves every PRIME_RANGE = 10^8 and calls ecm_stage1 once per range
 * (ecm.c:1209-1234); between the calls it appends the batch to checkpoint.txt (ecm.c:1236-1312).  One
 * gecm_stage1_range call is one of those ecm_stage1 calls, with the reference's behaviour there reproduced so that
 * the residues stay bit-identical: every call runs the 2-power doublings again (ecm.c:1815-1822) and starts at the
 * SECOND prime of its range (ecm.c:1824: i = 1; the first prime above each multiple of 10^8 is never processed).
 * range = 0 .. gecm_stage1_ranges(B1) - 1, in order, on the points the previous range left on the device;
 * asynchronous like gecm_stage1.  While the device runs range r the library compiles the tape of range r + 1 on a
 * helper thread.  B1 <= 10^12. */
int gecm_stage1_ranges(uint64_t B1);                   /* ceil(B1 / 10^8), at least 1 */
int gecm_stage1_range(gecm_ctx *ctx, uint64_t B1, uint32_t range);
/* What vececm prints and decides around one range (ecm.c:1215-1247, 1849): the sieved interval [lo, hi] with
 * hi = min(B2 + 1000, lo + 10^8) and its prime count ("Found %lu primes in range [lo : hi]"), P_MIN ("Commencing
 * Stage 1 @ prime"), the last prime the call processes ("Stage 1 completed at prime", the B1 field of the
 * checkpoint lines), and whether the reference writes checkpoint.txt after this range: it does when the range
 * holds no prime >= B1 (ecm.c:1237 reads PRIMES[last_pid] one past the list, a zero word of the fresh
 * allocation) — every range but the last, and also the last (or only) one when B1 lies above its last prime.   */
typedef struct {
    uint64_t lo, hi, nprimes, first_prime, last_prime;
    int checkpoint;
} gecm_stage1_range_desc;
int gecm_stage1_describe_range(uint64_t B1, uint64_t B2, uint32_t range, gecm_stage1_range_desc *out);
/* How stage 1 maps curves to lanes.  1 = one curve per lane (64 per wavefront): the throughput layout,
 * full speed from 2 wavefronts per SIMD, i.e. 128 x (4 x CUs) = 131072 curves on MI355X.  2 = the X and
 * the Z coordinate of a curve on two adjacent lanes (32 curves per wavefront): each point operation's
 * independent halves (ecm.c:417-440, 447-454) run side by side, so a curve finishes in half the time
 * and a batch fills the chip at half the size (1.83x the curves/s up to 32768 curves on MI355X).
 * 8 = X and Z on two adjacent quads of lanes and the limbs of each residue spread over the four lanes of
 * its quad (8 curves per wavefront), with a row-wise Montgomery multiply whose digit and operand limbs are
 * broadcast inside the quad: another 1.5x (416-bit) to 2.4x (1024-bit) for batches up to 8192 curves.
 * 0 (default) = chosen per launch from the batch size, the limb count and the CU count
 * (gecm_dev_auto_lanes).  Results are identical in every layout.
 * gecm_get_lanes_per_curve returns what the last gecm_stage1 launch used (0 before the first). */
int gecm_set_lanes_per_curve(gecm_ctx *ctx, int lanes);
/* N | 2^k - 1, N | 2^k + 1 (Cunningham cofactors) or N | 2^k - c with c odd and below one reference limb (pseudo-
 * Mersenne inputs); the reference's isMersenne == +1 / -1 / c, main.c:410-441, for which it switches to
 * vecmulmod52_mersenne: stage 1 runs modulo 2^k -/+ 1 or 2^k - c with a multiply whose reduction
 * half needs almost no multiplications, and X, Z are reduced modulo N when they come back.  Chosen at
 * gecm_create when it is the cheaper multiply; outputs are the same residues modulo N either way.
 * gecm_set_special_form(ctx, 0) keeps everything on the generic REDC path (takes effect at the next
 * gecm_build_curves / gecm_upload_points).  A batch small enough for the eight-lane layout runs there, on
 * generic REDC, which is faster still.  gecm_get_special_form returns 0 if the special multiply is off or not
 * available, 1 if it is enabled, 2 if the last gecm_stage1 launch actually used it, and stores +k for
 * 2^k - 1 and 2^k - c (c: gecm_prepare_input's info.c), -k for 2^k + 1, and the limb count of that modulus (0, 0 if N
 * has no such form or it would not pay). */
int gecm_set_special_form(gecm_ctx *ctx, int on);
int gecm_get_special_form(const gecm_ctx *ctx, int *k, int *limbs);
int gecm_get_lanes_per_curve(const gecm_ctx *ctx);
/* Progress of the stage-1 call in flight: a long op tape (200 MB for a 1e8 prime range) runs as several kernel launches,
 * cut between two prac() calls; *done of *total have finished.  Callable from the launching thread between
 * gecm_stage1 / gecm_stage1_range and gecm_sync (the reference prints "accumulating prime" every 8192 primes). */
int gecm_stage1_progress(const gecm_ctx *ctx, uint32_t *done, uint32_t *total);
/* the stage-1 kernel the last launch ran, by the name rocprofv3 prints for it ("k_stage1_rowp<1, 16>", "k_stage1<15>"):
 * what a profile of the run has to be matched against */
int gecm_last_kernel_name(const gecm_ctx *ctx, char *buf, size_t len);
/* milliseconds of the last stage-1 kernel, from HIP events on the context's stream */
double gecm_last_kernel_ms(const gecm_ctx *ctx);

typedef struct {
    uint64_t ptadds, ptdups;   /* ecm.c:441, 455; printed at ecm.c:1849-1850: summed over the ranges run so far */
    uint64_t last_prime;       /* ecm.c:1849: of the last range run */
    uint64_t tape_len;         /* of the last range run */
} gecm_stage1_stats;
int gecm_get_stage1_stats(const gecm_ctx *ctx, gecm_stage1_stats *st);

/* P after stage 1 as vec operands (reference layout, Montgomery radix, canonical). */
int gecm_download_points(gecm_ctx *ctx, void *X, void *Z);
/* The de-Montgomeryised X*1, Z*1 the reference writes to save_b1.txt (ecm.c:1327-1331). */
int gecm_download_points_plain(gecm_ctx *ctx, void *x, void *z);

/* ---- save / factor path (ecm.c:1319-1388, check_factor ecm.c:2542-2557) --------------------
 * Formats curve k's resume line exactly as ecm.c:1372-1380:
 *   "METHOD=ECM; SIGMA=%lu; B1=%lu; N=0x%Zx; X=0x%Zx; Z=0x%Zx; PROGRAM=AVX-ECM;\n"
 * from the last downloaded stage-1 result.  Returns the line length, or < 0.                  */
int gecm_format_save_line(gecm_ctx *ctx, size_t k, char *buf, size_t buflen);
/* The same line with another B1 field: the checkpoint.txt lines of ecm.c:1295-1305 carry the last prime of the
 * range just finished (PRIMES[last_pid - 1]) instead of B1.                                              */
int gecm_format_resume_line(gecm_ctx *ctx, size_t k, uint64_t b1_field, char *buf, size_t buflen);
/* gcd(Z_k, N) after stage 1 (ecm.c:1336-1344): returns 1 and the factor as a decimal string if
 * 1 < g < N, else 0 (g == N is "no factor", ecm.c:2549-2553).                                 */
int gecm_stage1_factor(gecm_ctx *ctx, size_t k, char *dec, size_t declen, int *is_prp);

/* Whole-batch factor scan on the device (the reference scans lane by lane on the host,
 * ecm.c:1323-1370, 1485-1528): stage = 1 checks gcd(Z_k, N) after stage 1, stage = 2 checks
 * gcd(acc_k, N) after stage 2, for every curve at once.  Returns the number of curves with a factor
 * (>= 0) and, if first != NULL, the lowest such curve index (or batch if none).  The per-curve
 * functions above/below then format the factors of the flagged curves.                          */
int gecm_scan_factors(gecm_ctx *ctx, int stage, size_t *first);
/* 1 if curve k was flagged by the last gecm_scan_factors call of that stage, else 0 */
int gecm_curve_flag(const gecm_ctx *ctx, int stage, size_t k);

/* ---- L1 phase 2: stage-2 init (ecm_stage2_init, ecm.c:2201-2340) --------------------------
 * Q = the stage-1 result resident on the device.  Builds the baby-step table Pb[map[j]] = [j]Q for
 * j <= U*D with gcd(j, D) = 1 (X/Z-normalised by batch inversion, ecm.c:2322), Pd = [D]Q and
 * acc = one.  D = 0 picks the reference's wheel for the current B1 (main.c:838-872); U = 0 picks 16
 * (the value every reference run chose through its uninitialised `paircost`, main.c:912, 943).
 * Asynchronous; gecm_sync waits.                                                                */
int gecm_stage2_init(gecm_ctx *ctx, uint32_t D, uint32_t U);

/* ---- host pair map (pair, ecm.c:2559-2910) --------------------------------------------------
 * Montgomery's PAIR over the primes in [B1, B2) for wheel D and table height U.  Arrays are
 * malloc'ed; release with gecm_pairmap_release.  (0,0) entries mean "advance the window".       */
typedef struct {
    uint32_t *pairmap_v, *pairmap_u;   /* ecm.c:2559 */
    uint32_t steps;                    /* return value of pair() */
    uint32_t amin;                     /* work->amin set at ecm.c:2571 */
    uint32_t pairs, primes;            /* printed at ecm.c:2904-2905 */
} gecm_pairs;
int gecm_pair_primes(gecm_pairs *out, uint64_t B1, uint64_t B2, uint32_t D, uint32_t U);
void gecm_pairmap_release(gecm_pairs *p);

/* ---- L1 phase 3: stage-2 pair (ecm_stage2_pair, ecm.c:2342-2540) ----------------------------
 * One B2 range: giant steps from A = 2*amin*D, window of 2L = 4U steps, walk of the pair map,
 * acc <- acc * (Xa/Za - Xb/Zb) per pair (CROSS_PRODUCT_INV, ecm.c:1857-1859).  Same arguments as the
 * reference (pairmap_steps, pairmap_v, pairmap_u, work->amin).  Asynchronous.                   */
int gecm_stage2_pair(gecm_ctx *ctx, uint32_t steps, const uint32_t *pairmap_v, const uint32_t *pairmap_u,
                     uint32_t amin);
/* Optional: the host-side preparation gecm_stage2_pair makes for a pair map (its launch tape) ahead of time, e.g.
 * while the device runs stage 1; D, U as for gecm_stage2_init (explicit, not 0).  The reference has no counterpart
 * (its ecm_stage2_pair walks the map directly, ecm.c:2448-2533).  gecm_stage2_pair keeps the tape of the last map it
 * saw either way and recognises the map (its length, amin, D, U and two independent hashes of its words), so a run of
 * many batches prepares it once.  Called with a (D, U) other than the one of the last gecm_stage2_init it replaces the
 * context's stage-2 plan: results of a finished stage 2 (accumulator, factors) must have been read before.  The same
 * holds for gecm_stage2_prepare below. */
int gecm_stage2_pair_prepare(gecm_ctx *ctx, uint32_t D, uint32_t U, uint32_t steps, const uint32_t *pairmap_v,
                             const uint32_t *pairmap_u, uint32_t amin);

/* Convenience: the whole stage-2 sequence of vececm (ecm.c:1401-1476) for primes in [B1, B2):
 * init, then pair + stage2_pair per range of 1e8.  Synchronous.                                  */
int gecm_stage2(gecm_ctx *ctx, uint64_t B2, uint32_t D, uint32_t U);
/* Optional: make (and keep) the pair map gecm_stage2(ctx, B2, D, U) will need, e.g. between gecm_stage1 and
 * gecm_sync, while the device runs stage 1 (the reference computes it between the stages, on the main thread:
 * ecm.c:1441-1443).  Also kept after a gecm_stage2 call, so a run of many batches computes it once. */
int gecm_stage2_prepare(gecm_ctx *ctx, uint64_t B2, uint32_t D, uint32_t U);

typedef struct {
    uint64_t ptadds, numinv, paired;   /* the reference's counters, ecm.c:1482-1483 (numinv as the
                                          reference counts it; the device normalises the baby-step
                                          table in blocks, see stage2_device_inversions) */
    uint64_t device_inversions;
    uint32_t D, U, L, amin_last;
} gecm_stage2_stats;
int gecm_get_stage2_stats(const gecm_ctx *ctx, gecm_stage2_stats *st);

/* stg2acc as a vec operand (reference layout, Montgomery radix, canonical), ecm.c:1489 */
int gecm_download_acc(gecm_ctx *ctx, void *acc);
/* Factor check after stage 2 (ecm.c:1485-1528): gcd(acc_k, N), or — when a batch inversion met a
 * non-invertible product (ecm.c:1927-1939) — the gcd recorded then.  Returns 1 and the decimal
 * factor if 1 < g < N, else 0.                                                                  */
int gecm_stage2_factor(gecm_ctx *ctx, size_t k, char *dec, size_t declen, int *is_prp);

#ifdef __cplusplus
}
#endif
#endif
