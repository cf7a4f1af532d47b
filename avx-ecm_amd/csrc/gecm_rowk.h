/* gecm_rowk.h — launcher of the 32-lanes-per-curve stage-1 kernels (csrc/gecm_row.hpp, gecm_rowk.hip): one entry
 * point for every limb count. */
#ifndef GECM_ROWK_H
#define GECM_ROWK_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 32 lanes per curve (csrc/gecm_row.hpp, gecm_rowk.hip): nq = limbs per lane, rows = rows of a multiply = limbs in use
 * (the nl + 1 limbs of N' = m*N; round 3: no longer rounded up to a multiple of nq — 831 bits are 31 rows, not 32),
 * nl = limbs per residue of the device buffers, rc =
 * device array of GECM_ROW_KINDS x GECM_ROW_WORDS constants.  Leaves lazy values in X, Z (run gecm_launch_canon_<nl>
 * afterwards).  a_lds != 0: operand limbs are broadcast through the LDS crossbar instead of DPP (faster from 3
 * wavefronts per SIMD up).  Returns -1 if (nq, rows) is not built. */
#define GECM_ROW_WORDS 48     /* words per constant array: limbs 0 .. 16*nq-1, zero padded */
#define GECM_ROW_KINDS 5
#define GECM_ROW_MAXNQ 3
/* the shapes built: one per built limb count nl (gecm_launch.h's list): nq = ceil((nl+1)/16), rows = nl + 1 */
#define GECM_ROW_SHAPES(X) X(1, 9) X(1, 11) X(1, 13) X(1, 15) X(1, 16) X(2, 18) X(2, 20) X(2, 22) X(2, 24) X(2, 27) X(2, 29) \
    X(2, 31) X(3, 33) X(3, 35) X(3, 38)
static inline void gecm_row_shape(int nl, int *nq, int *rows)
{
    *nq = (nl + 1 + 15) / 16;
    *rows = nl + 1;
}
int gecm_launch_stage1_row(void *stream, int nq, int rows, const uint32_t *tape, uint32_t tape_len, uint32_t *X, uint32_t *Z,
                           const uint32_t *S, size_t stride, uint32_t nl, const uint32_t *rc, uint32_t rho_n,
                           int a_lds);

#ifdef __cplusplus
}
#endif
#endif
