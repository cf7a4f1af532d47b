/* gecm_launch.h — per-limb-count kernel launchers (one object file per GECM_NL). */
#ifndef GECM_LAUNCH_H
#define GECM_LAUNCH_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* limb counts built into the library; keep in sync with the Makefile's NLS */
#ifdef GECM_DEV_NL15
#define GECM_NL_LIST(X) X(15)      /* `make DEV=1`: quick developer build, 416-bit class only */
#else
#define GECM_NL_LIST(X) X(8) X(10) X(12) X(14) X(15) X(17) X(19) X(21) X(23) X(26) X(28) X(30) X(32) X(34) X(37)
#endif

typedef struct {
    const uint32_t *n, *kp, *one, *r3;
    uint32_t rho, inv_iters;
} gecm_modconst;

typedef struct {
    const uint32_t *X, *Z, *S;
    uint32_t *PbX, *bx, *bz, *bp, *PdX, *PdZ, *acc, *fail;
    const uint32_t *keep;
    uint32_t umax, D, npb;
    size_t stride;
    /* K > 1: K sub-sequences per curve (csrc/gecm_stage2.hpp, s2_init_k) */
    uint32_t K;
    const uint32_t *tgt, *tgt_off;
    uint32_t *kbx, *kbz, *kbp, *PdKX, *PdKZ;
} gecm_s2_init_args;

typedef struct {
    const uint32_t *X, *Z, *S, *PbX, *PdX, *PdZ;
    uint32_t npb;
    uint32_t *gx, *gz, *gp, *ring, *acc, *fail;
    const uint32_t *steps;        /* device copy of the tape */
    const uint32_t *host_steps;   /* host copy: the launcher splits it at the "generate" marks */
    uint32_t nsteps, D, G, ring_size;
    uint32_t slices;              /* accumulators per curve: each run of pairs is cut into this many slices */
    uint64_t A0;
    size_t stride;
    /* K > 1: giant steps with K sub-sequences per curve (giant_chunk_k); Gs = entries per sub-sequence and chunk */
    uint32_t K, Gs;
    uint32_t *kgx, *kgz, *kgp;
    const uint32_t *PdKX, *PdKZ;
} gecm_s2_pair_args;

#define GECM_DECL(nl)                                                                                     \
    void gecm_launch_stage1_##nl(void *stream, const gecm_modconst *mc, const uint32_t *tape,             \
                                 uint32_t tape_len, uint32_t *X, uint32_t *Z, const uint32_t *S,          \
                                 size_t stride);                                                          \
    void gecm_launch_stage1_pair_##nl(void *stream, const gecm_modconst *mc, const uint32_t *tape,        \
                                      uint32_t tape_len, uint32_t *X, uint32_t *Z, const uint32_t *S,     \
                                      size_t stride);                                                     \
    void gecm_launch_stage1_f_##nl(void *stream, const gecm_modconst *mc, const uint32_t *tape,           \
                                   uint32_t tape_len, uint32_t *X, uint32_t *Z, const uint32_t *S,        \
                                   size_t stride, int lanes, int form);                                   \
    int gecm_launch_stage1_quad_##nl(void *stream, const gecm_modconst *mc, const uint32_t *tape,         \
                                     uint32_t tape_len, uint32_t *X, uint32_t *Z, const uint32_t *S,      \
                                     size_t stride, const uint32_t *modq);                                \
    void gecm_launch_canon_##nl(void *stream, const gecm_modconst *mc, uint32_t *X, uint32_t *Z,          \
                                size_t stride);                                                           \
    int gecm_fform_generic_limbs_##nl(void);                                                              \
    void gecm_launch_from_mont_##nl(void *stream, const gecm_modconst *mc, const uint32_t *X,             \
                                    const uint32_t *Z, uint32_t *ox, uint32_t *oz, size_t stride);        \
    void gecm_launch_l0_##nl(void *stream, const gecm_modconst *mc, int op, const uint32_t *A,            \
                             const uint32_t *B, uint32_t *C, uint32_t *D, size_t stride,                  \
                             const uint32_t *fix);                                                        \
    void gecm_launch_s2_init_##nl(void *stream, const gecm_modconst *mc, const gecm_s2_init_args *h);     \
    void gecm_launch_s2_pair_##nl(void *stream, const gecm_modconst *mc, const gecm_s2_pair_args *h);     \
    void gecm_launch_s2_acc_init_##nl(void *stream, const gecm_modconst *mc, uint32_t *acc,               \
                                      uint32_t slices, size_t stride);                                    \
    void gecm_launch_gcd_scan_##nl(void *stream, const gecm_modconst *mc, const uint32_t *V, uint32_t *G, \
                                   uint32_t *flags, size_t stride);
GECM_NL_LIST(GECM_DECL)
#undef GECM_DECL

#ifdef __cplusplus
}
#endif
#endif
