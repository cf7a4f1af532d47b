/* ecm_oracle.h — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or executed from the
 * product (avx-ecm_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may use it, and only as the checker.
 *
 * Scalar-C restatement of the reference's hot path (bbuhrow/avx-ecm), one curve at a time, in the
 * reference's own limb radix (DIGITBITS = 52 or 32) and Montgomery radix R = 2^(DIGITBITS*NWORDS).
 * Each function cites the reference lines it follows.  Pinned against the reference's own
 * outputs: tests/test_oracle.py checks it against tests/golden/{l0,stage1}.json, which were
 * produced by running the reference itself (oracle/_ref, built from /root/reference by
 * oracle/Makefile).
 */
#ifndef ECM_ORACLE_H
#define ECM_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#define ORC_MAXW 80

typedef struct orc_ctx orc_ctx;

/* main.c:465-483 (NWORDS rule) and main.c:597-640 (Montgomery constants).  n_str decimal or 0x-hex. */
orc_ctx *orc_create(const char *n_str, int digitbits);
void orc_destroy(orc_ctx *c);
int orc_nwords(const orc_ctx *c);
int orc_maxbits(const orc_ctx *c);

/* ---- L0 (values as hex strings, canonical, reference Montgomery radix) ---------------------- */
/* op: 0 mul (vecarith52.c:2438), 1 sqr (:3317), 2 add (:4550), 3 sub (:4684); out must hold 2*ORC_MAXW*16 */
int orc_l0_hex(orc_ctx *c, int op, const char *a_hex, const char *b_hex, char *out_hex);

/* ---- L1 --------------------------------------------------------------------------------------- */
/* build_one_curve (ecm.c:1548-1803) + ecm_stage1 (ecm.c:1806-1854) + the save line of
 * ecm.c:1372-1380 for one sigma.  Returns the line length.  If factor_dec != NULL it receives the
 * decimal gcd(Z,N) when check_factor (ecm.c:2542-2557) reports one, else "".  counts[0..1] =
 * point adds / doublings (ecm.c:441, 455). */
int orc_stage1_line(orc_ctx *c, uint64_t sigma, uint64_t B1, char *line, size_t linelen, char *factor_dec,
                    size_t faclen, uint64_t *counts);

/* The same through vececm's loop over prime ranges (ecm.c:1209-1312; B1 above prime_range = 1e8 in the reference):
 * ecm_stage1 once per range, stopping after `stop_after` calls (0 = all).  b1_field = 0 writes the last prime
 * processed into the line's B1 field, as the checkpoint.txt lines do (ecm.c:1295-1305).  counts[0..2] = point adds,
 * doublings, last prime.  *checkpoint = the reference appends checkpoint.txt after the last call made. */
int orc_stage1_ranges_line(orc_ctx *c, uint64_t sigma, uint64_t B1, uint64_t B2, uint64_t prime_range, int stop_after,
                           uint64_t b1_field, char *line, size_t linelen, char *factor_dec, size_t faclen,
                           uint64_t *counts, int *checkpoint);

/* Stage 1 then stage 2 (ecm_stage2_init ecm.c:2201-2340, pair ecm.c:2559-2910, ecm_stage2_pair
 * ecm.c:2342-2540, driver loop ecm.c:1401-1476) for one sigma with explicit D and U (the reference
 * picks U through an uninitialised variable, main.c:912, 943; observed value 16).
 * acc_hex: stg2acc in Montgomery form, canonical.  factor_dec: decimal factor from gcd(acc, N) or "".
 * counts[0..2] = stage-2 point adds, inversions, pair multiplications (ecm.c:1482-1483). */
int orc_stage2(orc_ctx *c, uint64_t sigma, uint64_t B1, uint64_t B2, uint32_t D, uint32_t U, char *acc_hex,
               char *factor_dec, size_t faclen, uint64_t *counts);

/* pair map for one range (ecm.c:2559-2910); returns number of steps; arrays malloc'ed by callee */
uint32_t orc_pair(uint64_t B1, uint64_t B2, uint32_t D, uint32_t U, uint32_t **pm_v, uint32_t **pm_u,
                  uint32_t *amin_out, uint32_t *pairs_out, uint32_t *nump_out);

/* PRAC cost model and multiplier choice (ecm.c:479-563, 574-582), exposed for tests */
double orc_lucas_cost(uint64_t n, double v);
int orc_prac_choice(uint64_t c);

/* timing leg for bench.py's cpu_baseline (kind "port"): stage 1 of `curves` curves, one after the
 * other, single thread; returns seconds */
double orc_time_stage1(orc_ctx *c, uint64_t sigma0, int curves, uint64_t B1);

#endif
