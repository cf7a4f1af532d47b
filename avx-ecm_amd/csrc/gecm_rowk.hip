// gecm_rowk.hip — the 32-lanes-per-curve stage-1 kernels (gecm_row.hpp).  One translation unit for all limb
// counts: the kernel is templated on the limbs per lane (NQ = 1, 2, 3 cover up to 16, 32, 48 limbs of 28 bits) and on
// the rows of a multiply (ROWS = the limbs of N' = m*N rounded up to whole lanes); the limb count of the device
// buffers is a run-time argument.
#include "gecm_row.hpp"
#include <hip/hip_runtime.h>

// Four wavefronts per workgroup (they do not interact).  With one, the first launch of a process was 41 % slower than
// every later one in 13 of 18 runs (381 ms against 270 ms at 4096 curves, B1 = 1e5; 3.8 s against 2.7 s at B1 = 1e6 in
// the command-line driver, whose only stage-1 launch is its first): these kernels need few registers, a SIMD can take
// eight of their wavefronts, and nothing makes a cold dispatcher spread 2048 one-wavefront workgroups two per SIMD.
// A workgroup of four puts one wavefront on each SIMD of a CU: 14 first launches of 14 at full speed
// (profiles/r02_first_launch_workgroup_size.txt).
// (GECM_ROW_WG_WAVES: gecm_row.hpp)
template <int NQ, int ROWS, bool ALDS>
__global__ void __launch_bounds__(64 * GECM_ROW_WG_WAVES, 2)
k_stage1_row(const uint32_t *__restrict__ tape, uint32_t tape_len, uint32_t *__restrict__ X, uint32_t *__restrict__ Z,
             const uint32_t *__restrict__ S, size_t stride, uint32_t nl, const uint32_t *__restrict__ rc, uint32_t rho_n)
{
    stage1_row<NQ, ROWS, ALDS ? 1 : 0>(tape, tape_len, X, Z, S, stride, nl, rc, rho_n);
}

// the LDS-prefetch variant (gecm_row.hpp, run_tape_row_lds): point forms in LDS, 15 KB per limb per lane and workgroup.
// An experiment that lost: built only with -DGECM_ROW_LDS_VARIANT (tools/ab_row_libs.py).
#ifdef GECM_ROW_LDS_VARIANT
template <int NQ, int ROWS>
__global__ void __launch_bounds__(64 * GECM_ROW_WG_WAVES, 2)
k_stage1_rowp(const uint32_t *__restrict__ tape, uint32_t tape_len, uint32_t *__restrict__ X, uint32_t *__restrict__ Z,
              const uint32_t *__restrict__ S, size_t stride, uint32_t nl, const uint32_t *__restrict__ rc, uint32_t rho_n)
{
    __shared__ __attribute__((aligned(16))) RowSlots<NQ> slots[GECM_ROW_WG_ROWS];
    stage1_row<NQ, ROWS, 2>(tape, tape_len, X, Z, S, stride, nl, rc, rho_n, slots);
}
#define GECM_ROWP_LAUNCH(q, r) hipLaunchKernelGGL((k_stage1_rowp<q, r>), grid, block, 0, (hipStream_t)stream, tape, tape_len, X, Z, S, stride, nl, rc, rho_n)
#else
#define GECM_ROWP_LAUNCH(q, r) return -1
#endif

/* rc = device array of GECM_ROW_KINDS x GECM_ROW_WORDS words (gecm_row.hpp).  Leaves lazy values (limbs < 2^28 + 4,
 * value < K + 2N) in X, Z: the caller runs k_canon afterwards.  a_lds = 1: operand broadcasts through the LDS crossbar
 * (for launches of 3 or more wavefronts per SIMD); 2: the LDS-prefetch variant k_stage1_rowp.  (nq, rows) must be one of the built pairs — gecm_row_shape() of a
 * built limb count; returns -1 otherwise. */
extern "C" int gecm_launch_stage1_row(void *stream, int nq, int rows, const uint32_t *tape, uint32_t tape_len, uint32_t *X,
                                      uint32_t *Z, const uint32_t *S, size_t stride, uint32_t nl, const uint32_t *rc,
                                      uint32_t rho_n, int a_lds)
{
    const dim3 grid((unsigned)(stride / (2 * GECM_ROW_WG_WAVES))), block(64 * GECM_ROW_WG_WAVES);   // stride: a multiple of 64
#define GECM_ROW_LAUNCH(q, r)                                                                                               \
    if (nq == q && rows == r) {                                                                                             \
        if (a_lds == 2)                                                                                                     \
            GECM_ROWP_LAUNCH(q, r);                                                                                         \
        else if (a_lds)                                                                                                     \
            hipLaunchKernelGGL((k_stage1_row<q, r, true>), grid, block, 0, (hipStream_t)stream, tape, tape_len, X, Z, S,    \
                               stride, nl, rc, rho_n);                                                                      \
        else                                                                                                                \
            hipLaunchKernelGGL((k_stage1_row<q, r, false>), grid, block, 0, (hipStream_t)stream, tape, tape_len, X, Z, S,   \
                               stride, nl, rc, rho_n);                                                                      \
        return 0;                                                                                                           \
    }
    GECM_ROW_SHAPES(GECM_ROW_LAUNCH)
#undef GECM_ROW_LAUNCH
    return -1;
}

// the hash of the sources this object was compiled from (Makefile: R_SHA)
#ifndef GECM_MANIFEST
#define GECM_MANIFEST "unset"
#endif
extern "C" const char *gecm_manifest_rowk(void) { return GECM_MANIFEST; }
