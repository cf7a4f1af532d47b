/* gecm_dev.h — internal device layer (HIP side) of libgecm.  Plain C interface so that the
 * host logic (C, gcc) never sees HIP types.  Not part of the public ABI (that is include/gecm.h).
 *
 * All residues crossing this layer are NL limbs of 28 bits in uint32_t, struct-of-arrays
 * [limb][curve] (curve index fastest: consecutive lanes of a wavefront read consecutive words),
 * in the engine's internal Montgomery form (R = 2^(28*NL)) unless a function says otherwise.
 */
#ifndef GECM_DEV_H
#define GECM_DEV_H
#include <stddef.h>
#include <stdint.h>
#include "gecm_ops.h"
#include "gecm_rowk.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct gecm_dev gecm_dev;


int gecm_dev_count(void);
/* hashes of the sources the device objects were compiled from: "K:.. R:.. D:.." (Makefile; K = MIXED if the kernel
 * objects disagree) */
const char *gecm_dev_manifest(void);
const char *gecm_dev_error(void);
/* limb counts for which kernels are instantiated, ascending, 0-terminated */
const int *gecm_dev_supported_nl(void);

int gecm_dev_open(gecm_dev **out, int device, int nl, const uint32_t *n, const uint32_t *kp,
                  const uint32_t *one, uint32_t rho);
void gecm_dev_close(gecm_dev *d);
int gecm_dev_device_name(gecm_dev *d, char *buf, size_t len);
int gecm_dev_memory(gecm_dev *d, uint64_t *free_bytes, uint64_t *total_bytes);   /* hipMemGetInfo */
/* device bytes of a batch: stage-1 arrays, plus (npb != 0) the stage-2 allocations for that table / chunk / ring size */
uint64_t gecm_dev_batch_bytes(gecm_dev *d, size_t curves, uint32_t npb, uint32_t G, uint32_t ring_size);

/* (re)allocate state for ncurves curves: X, Z, S (+ scratch for downloads) */
int gecm_dev_resize(gecm_dev *d, size_t ncurves);
size_t gecm_dev_stride(gecm_dev *d);
int gecm_dev_upload(gecm_dev *d, const uint32_t *X, const uint32_t *Z, const uint32_t *S);
int gecm_dev_upload_xz(gecm_dev *d, const uint32_t *X, const uint32_t *Z);   /* X, Z only; S untouched */
int gecm_dev_set_tape(gecm_dev *d, const uint8_t *tape, size_t len);
/* stage 1: asynchronous on the context's stream; HIP events bracket the kernel */
/* lanes_per_curve: 1 = one curve per lane, 2 = X and Z of a curve on two adjacent lanes (for batches
 * too small to fill the chip), 0 = let the device layer choose from the batch size and CU count */
int gecm_dev_stage1(gecm_dev *d, int lanes_per_curve);
int gecm_dev_auto_lanes(gecm_dev *d);
/* launches of the last gecm_dev_stage1 finished so far / made (a long tape is cut into several, gecm_dev_set_tape) */
int gecm_dev_stage1_progress(gecm_dev *d, uint32_t *done, uint32_t *total);
/* constants of the 32-lanes-per-curve kernel (csrc/gecm_row.hpp): nq limbs per lane, rows per multiply
 * (gecm_row_shape), GECM_ROW_KINDS x
 * GECM_ROW_WORDS words (N' = m*N = -1 mod 2^28; N; entry factor; R mod N; K' of N), limb j at word j */
int gecm_dev_set_rowconst(gecm_dev *d, int nq, int rows, const uint32_t *words);
/* F-form (modulus 2^k - 1, csrc/gecm_field.hpp): number of top limbs the kernel for `nl` limbs reads from
 * the modulus (all limbs below must be 2^28 - 1), and the switch that makes gecm_dev_stage1 use it. */
int gecm_dev_fform_generic_limbs(int nl);
void gecm_dev_set_fform(gecm_dev *d, int form);   /* +1: 2^k - 1, -1: 2^k + 1, 2: 2^k - c (limbs 0, 1 below F), 0: off */
int gecm_dev_last_lanes(gecm_dev *d);
/* name of the stage-1 kernel the last launch ran, as rocprofv3 prints it ("k_stage1_rowp<1, 16>") */
const char *gecm_dev_last_kernel(gecm_dev *d);
int gecm_dev_sync(gecm_dev *d);
float gecm_dev_last_kernel_ms(gecm_dev *d);
/* canonical Montgomery-form X, Z (what P holds after ecm_stage1 in the reference, modulo R) */
int gecm_dev_download_mont(gecm_dev *d, uint32_t *X, uint32_t *Z);
/* canonical de-Montgomeryised x, z (the reference's X*1, Z*1 of ecm.c:1327-1331) */
int gecm_dev_download_plain(gecm_dev *d, uint32_t *x, uint32_t *z);

/* test-level L0 operators on `count` independent residues.  Inputs canonical (< N), internal
 * Montgomery form; outputs canonical.  For MUL/SQR the product is additionally multiplied by
 * the constant `fix` (internal Montgomery form; pass `one` for none): this is how the public
 * ABI returns results in the reference's own Montgomery radix. */
int gecm_dev_l0(gecm_dev *d, int op, const uint32_t *a, const uint32_t *b, uint32_t *c, uint32_t *dd,
                size_t count, const uint32_t *fix);

/* ---- stage 2 (csrc/gecm_stage2.hpp) ----
 * r3 = R^3 mod N (28-bit limbs); inv_iters = batches of 28 division steps of the device inversion (fe_invert). */
int gecm_dev_set_s2const(gecm_dev *d, const uint32_t *r3, uint32_t inv_iters);
/* ecm_stage2_init: baby-step table (npb entries, X/Z normalised), Pd = [D]Q, acc = one.
 * keep: bitmap over j in [0, umax], bit set iff j is stored.  L = ring half-size (2L giant steps). */
/* K = gecm_dev_s2_subseq(d) sub-sequences per curve (1: the plain chain).  For K > 1: tgt_off[r] .. tgt_off[r+1] is
 * the range of tgt[] holding the table indices of the kept members j = r, r+K, ... (r = 0: K, 2K, ...) of
 * sub-sequence r, in order; tgt_off has K + 1 entries. */
uint32_t gecm_dev_s2_subseq(gecm_dev *d);
int gecm_dev_s2_init(gecm_dev *d, const uint32_t *keep, size_t keep_words, uint32_t umax, uint32_t D,
                     uint32_t npb, uint32_t G, uint32_t ring_size, const uint32_t *tgt, const uint32_t *tgt_off,
                     uint32_t K);
/* failure records come in planes of [limb][curve]: plane 0 from the single-chain inversions, plane 1 + r from
 * sub-sequence r; gecm_dev_s2_download fills all of them */
uint32_t gecm_dev_s2_fail_planes(gecm_dev *d);
/* ecm_stage2_pair for one range: steps = nsteps words pairs: (0xffffffff, n) = generate the next n
 * giant steps (n <= G), else (ring slot, table index).  G = chunk size, ring_size = power of two. */
int gecm_dev_s2_pair(gecm_dev *d, const uint32_t *steps, uint32_t nsteps, uint32_t D, uint32_t G,
                     uint32_t ring_size, uint64_t A0, uint64_t tape_id);
#define GECM_S2_BLK 256   /* baby-step normalisation block (= S2_BLK in gecm_stage2.hpp) */
/* canonical Montgomery-form accumulator and the failed-inversion gcd records ([limb][curve]) */
int gecm_dev_s2_download(gecm_dev *d, uint32_t *acc, uint32_t *fail);
/* factor scan on the device: which = 0 -> stage-1 Z, 1 -> stage-2 accumulator.  flags[curve] = 1 iff
 * 1 < gcd(value, N) < N; g = the gcds ([limb][curve]); either output may be NULL. */
int gecm_dev_gcd_scan(gecm_dev *d, int which, uint32_t *flags, uint32_t *g);

#ifdef __cplusplus
}
#endif
#endif
