// gecm_kernels.hip — HIP kernels of libgecm for ONE limb count (compile with -DGECM_NL=<n>).
// One translation unit per limb count so the (large, fully unrolled) kernels build in parallel.
// See gecm_field.hpp / gecm_curve.hpp for the arithmetic; DESIGN.md for the layout.
#include "gecm_ops.h"
#include "gecm_launch.h"
#include "gecm_curve.hpp"
#include "gecm_stage2.hpp"
#include "gecm_quad.hpp"
#include <hip/hip_runtime.h>

#ifndef GECM_NL
#error "compile with -DGECM_NL=<limbs>"
#endif
// GECM_PART splits one limb count over two objects so the build parallelises better:
//   1 = stage 1, de-Montgomeryisation, L0 operators, factor scan;  2 = stage 2;  unset = both.
#ifndef GECM_PART
#define GECM_PART 0
#endif
#define GECM_HAS_PART(p) (GECM_PART == 0 || GECM_PART == (p))

template <int NL>
struct ModArgs {
    ModK<NL> m;
    Fe<NL> one;   // R mod N, canonical
};
#if GECM_HAS_PART(1)
// ---------------------------------------------------------------- stage 1
// One curve per lane.  64-thread blocks (one wave): a CU holds 8 of them at 2 waves/SIMD, the
// occupancy at which v_mad_u64_u32 issues back-to-back (profiles/r01_valu_ubench_gfx950.txt).
template <int NL>
__global__ void __launch_bounds__(64, 2)
k_stage1(const uint32_t *__restrict__ tape, uint32_t tape_len, uint32_t *__restrict__ X,
         uint32_t *__restrict__ Z, const uint32_t *__restrict__ S, size_t stride, ModArgs<NL> a)
{
    uint32_t idx = blockIdx.x * 64u + threadIdx.x;
    Pt<NL> P;
    fe_load(P.X, X, stride, idx);
    fe_load(P.Z, Z, stride, idx);
    constexpr bool CL = TapePolicy<NL>::c_in_lds;
    __shared__ uint32_t lds_c[CL ? 2 * NL * 64 : 1];
    CStore<NL, CL> cst;
    if constexpr (CL) cst.lds = lds_c + threadIdx.x;
    run_tape<NL>(tape, tape_len, P, S, stride, idx, a.m, cst);
    Fe<NL> ox, oz;
    fe_canonical_mont(ox, P.X, a.one, a.m);
    fe_canonical_mont(oz, P.Z, a.one, a.m);
    fe_store(X, stride, idx, ox);
    fe_store(Z, stride, idx, oz);
}

// Stage 1 modulo Mw = 2^k - 1 (gecm_field.hpp, "F-form"): the same interpreter, the REDC half of every
// multiply replaced by the shift-and-subtract form.  Used by the host for N | 2^k - 1.
template <int NL, class MOD>
struct ModArgsS {
    MOD m;
    Fe<NL> one;
};

// MOD = ModF<NL> (2^k - 1), ModP<NL> (2^k + 1) or ModC<NL> (2^k - c)
template <int NL, class MOD>
__global__ void __launch_bounds__(64, 2)
k_stage1_f(const uint32_t *__restrict__ tape, uint32_t tape_len, uint32_t *__restrict__ X,
           uint32_t *__restrict__ Z, const uint32_t *__restrict__ S, size_t stride, ModArgsS<NL, MOD> a)
{
    uint32_t idx = blockIdx.x * 64u + threadIdx.x;
    Pt<NL> P;
    fe_load(P.X, X, stride, idx);
    fe_load(P.Z, Z, stride, idx);
    constexpr bool CL = TapePolicy<NL>::c_in_lds;
    __shared__ uint32_t lds_c[CL ? 2 * NL * 64 : 1];
    CStore<NL, CL> cst;
    if constexpr (CL) cst.lds = lds_c + threadIdx.x;
    run_tape<NL>(tape, tape_len, P, S, stride, idx, a.m, cst);
    Fe<NL> ox, oz;
    fe_canonical_mont(ox, P.X, a.one, a.m);
    fe_canonical_mont(oz, P.Z, a.one, a.m);
    fe_store(X, stride, idx, ox);
    fe_store(Z, stride, idx, oz);
}

// Two lanes per curve (gecm_curve.hpp, "split-coordinate"): lane 2j works on X, lane 2j+1 on Z of
// curve blockIdx.x*32 + j.  Chosen by the device layer for batches that leave SIMDs under-occupied.
template <int NL>
__global__ void __launch_bounds__(64, 2)
k_stage1_pair(const uint32_t *__restrict__ tape, uint32_t tape_len, uint32_t *__restrict__ X,
              uint32_t *__restrict__ Z, const uint32_t *__restrict__ S, size_t stride, ModArgs<NL> a)
{
    const uint32_t cidx = blockIdx.x * 32u + (threadIdx.x >> 1);
    const bool isZ = (threadIdx.x & 1u) != 0;
    uint32_t *mine = isZ ? Z : X;
    Fe<NL> P;
    fe_load(P, mine, stride, cidx);
    run_tape_pair<NL>(tape, tape_len, P, S, stride, cidx, isZ, a.m);
    Fe<NL> o;
    fe_canonical_mont(o, P, a.one, a.m);
    fe_store(mine, stride, cidx, o);
}

template <int NL, class MOD>
__global__ void __launch_bounds__(64, 2)
k_stage1_pair_f(const uint32_t *__restrict__ tape, uint32_t tape_len, uint32_t *__restrict__ X,
                uint32_t *__restrict__ Z, const uint32_t *__restrict__ S, size_t stride, ModArgsS<NL, MOD> a)
{
    const uint32_t cidx = blockIdx.x * 32u + (threadIdx.x >> 1);
    const bool isZ = (threadIdx.x & 1u) != 0;
    uint32_t *mine = isZ ? Z : X;
    Fe<NL> P;
    fe_load(P, mine, stride, cidx);
    run_tape_pair<NL>(tape, tape_len, P, S, stride, cidx, isZ, a.m);
    Fe<NL> o;
    fe_canonical_mont(o, P, a.one, a.m);
    fe_store(mine, stride, cidx, o);
}

// Eight lanes per curve (gecm_quad.hpp).
#define GECM_HAS_QUAD 1
#if GECM_HAS_QUAD
// 256-thread workgroups (four independent wavefronts): this kernel needs about 65 registers, and one-wavefront
// workgroups of such a kernel are not spread evenly by a cold dispatcher — see gecm_rowk.hip; its first launch of a
// process measured 745 ms against 483 ms at 8192 curves.
template <int NL>
__global__ void __launch_bounds__(256, 2)
k_stage1_quad(const uint32_t *__restrict__ tape, uint32_t tape_len, uint32_t *__restrict__ X,
              uint32_t *__restrict__ Z, const uint32_t *__restrict__ S, size_t stride, const uint32_t *__restrict__ modq,
              uint32_t rho)
{
    const uint32_t cidx = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const uint32_t l = threadIdx.x & 3u;
    const bool isZ = (threadIdx.x & 4u) != 0;
    uint32_t *mine = isZ ? Z : X;
    constexpr int NQ = QuadShape<NL>::NQ;
    QuadMod<NL> m;
#pragma unroll
    for (int t = 0; t < NQ; t++) {         // modq: [n limbs 0..39 | K' limbs 0..39], zero padded
        m.n[t] = modq[NQ * l + t];
        m.kp[t] = modq[40 + NQ * l + t];
    }
    m.rho = rho;
    m.is0 = l == 0;
    m.top_mask = l == 3 ? 0u : 0xffffffffu;
    FeQn<NQ> P;
    feq_load<NL>(P, mine, stride, cidx, l);
    run_tape_quad<NL>(tape, tape_len, P, S, stride, cidx, l, isZ, m);
    feq_store<NL>(mine, stride, cidx, l, P);    // lazy representative; k_canon makes it canonical
}
#endif

// canonical Montgomery form of X, Z in place (the tail of k_stage1, for kernels that leave lazy values)
template <int NL>
__global__ void __launch_bounds__(64)
k_canon(uint32_t *__restrict__ X, uint32_t *__restrict__ Z, size_t stride, ModArgs<NL> a)
{
    uint32_t idx = blockIdx.x * 64u + threadIdx.x;
    Fe<NL> x, z, r;
    fe_load(x, X, stride, idx);
    fe_load(z, Z, stride, idx);
    fe_canonical_mont(r, x, a.one, a.m);
    fe_store(X, stride, idx, r);
    fe_canonical_mont(r, z, a.one, a.m);
    fe_store(Z, stride, idx, r);
}

template <int NL>
__global__ void __launch_bounds__(64)
k_from_mont(const uint32_t *__restrict__ X, const uint32_t *__restrict__ Z, uint32_t *__restrict__ ox,
            uint32_t *__restrict__ oz, size_t stride, ModArgs<NL> a)
{
    uint32_t idx = blockIdx.x * 64u + threadIdx.x;
    Fe<NL> x, z, r;
    fe_load(x, X, stride, idx);
    fe_load(z, Z, stride, idx);
    fe_from_mont_canonical(r, x, a.m);
    fe_store(ox, stride, idx, r);
    fe_from_mont_canonical(r, z, a.m);
    fe_store(oz, stride, idx, r);
}

// ---------------------------------------------------------------- L0 test-level operators
template <int NL>
__global__ void __launch_bounds__(64)
k_l0(int op, const uint32_t *__restrict__ A, const uint32_t *__restrict__ B, uint32_t *__restrict__ C,
     uint32_t *__restrict__ D, size_t stride, ModArgs<NL> a, Fe<NL> fix)
{
    uint32_t idx = blockIdx.x * 64u + threadIdx.x;
    Fe<NL> x, y, r, t;
    fe_load(x, A, stride, idx);
    fe_load(y, B, stride, idx);
    if (op == GECM_L0_MUL) {
        fe_mul(t, x, y, a.m);
        fe_canonical_mont(r, t, fix, a.m);
        fe_store(C, stride, idx, r);
    } else if (op == GECM_L0_SQR) {
        fe_sqr(t, x, a.m);
        fe_canonical_mont(r, t, fix, a.m);
        fe_store(C, stride, idx, r);
    } else {
        if (op == GECM_L0_ADD || op == GECM_L0_ADDSUB) {
            fe_add(t, x, y);
            fe_canonical_mont(r, t, a.one, a.m);
            fe_store(C, stride, idx, r);
        }
        if (op == GECM_L0_SUB || op == GECM_L0_ADDSUB) {
            fe_sub(t, x, y, a.m);
            fe_canonical_mont(r, t, a.one, a.m);
            fe_store(op == GECM_L0_SUB ? C : D, stride, idx, r);
        }
    }
}


// ---------------------------------------------------------------- factor scan
// check_factor (ecm.c:2542-2557) for every curve on the device: g = gcd(v, N) by the same
// fixed-iteration binary algorithm the stage-2 inversion uses; flag = 1 iff 1 < g < N.
// v is any representative (Montgomery form or not: R is a power of two, N is odd).
template <int NL>
__global__ void __launch_bounds__(64, 2)
k_gcd_scan(const uint32_t *__restrict__ V, uint32_t *__restrict__ G, uint32_t *__restrict__ flags, size_t stride,
           S2Const<NL> k)
{
    uint32_t idx = blockIdx.x * 64u + threadIdx.x;
    Fe<NL> v, c, t, g;
    fe_load(v, V, stride, idx);
    fe_canonical_mont(c, v, k.one, k.m);
    fe_invert(t, g, c, k.m, k.inv_iters);
    bool is_one = g.v[0] == 1u, is_n = true;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        if (i > 0) is_one = is_one && g.v[i] == 0;
        is_n = is_n && g.v[i] == k.m.n[i];
    }
    fe_store(G, stride, idx, g);
    flags[idx] = (!is_one && !is_n) ? 1u : 0u;
}

#endif
#if GECM_HAS_PART(2)
// ---------------------------------------------------------------- stage 2
template <int NL>
__global__ void __launch_bounds__(64, 2) k_s2_init(S2InitArgs a, S2Const<NL> k)
{
    s2_init<NL>(a, k, blockIdx.x * 64u + threadIdx.x);
}

// K sub-sequences per curve (small batches): block b works on curve block b / K, sub-sequence b % K
template <int NL>
__global__ void __launch_bounds__(64, 2) k_s2_init_k(S2InitArgs a, S2Const<NL> k)
{
    s2_init_k<NL>(a, k, (blockIdx.x / a.K) * 64u + threadIdx.x, blockIdx.x % a.K);
}

template <int NL>
__global__ void __launch_bounds__(64, 2) k_s2_gen_k(S2PairArgs a, uint32_t first_abs, uint32_t n, S2Const<NL> k)
{
    giant_chunk_k<NL>(a, first_abs, n, k, (blockIdx.x / a.K) * 64u + threadIdx.x, blockIdx.x % a.K);
}

// giant steps [first_abs, first_abs+n): generate + normalise into the ring
template <int NL>
__global__ void __launch_bounds__(64, 2) k_s2_gen(S2PairArgs a, uint32_t first_abs, uint32_t n, uint32_t kprev, S2Const<NL> k)
{
    giant_chunk<NL>(a, first_abs, n, first_abs == 0, k, blockIdx.x * 64u + threadIdx.x, kprev);
}

// pair walk over tape entries [first, first+count)
template <int NL>
__global__ void __launch_bounds__(64, 2) k_s2_pairs(S2PairArgs a, uint32_t first, uint32_t count, S2Const<NL> k)
{
    // gridDim.y slices of the segment, one accumulator each (see s2_pairs)
    const uint32_t per = (count + gridDim.y - 1) / gridDim.y;
    const uint32_t off = blockIdx.y * per;
    if (off >= count) return;
    const uint32_t n = count - off < per ? count - off : per;
    s2_pairs<NL>(a, first + off, n, k, blockIdx.x * 64u + threadIdx.x, a.acc + (size_t)blockIdx.y * NL * a.stride);
}

template <int NL>
__global__ void __launch_bounds__(64, 2) k_s2_merge(uint32_t *acc, uint32_t slices, size_t stride, int init_only,
                                                    S2Const<NL> k)
{
    s2_merge<NL>(acc, slices, stride, init_only != 0, k, blockIdx.x * 64u + threadIdx.x);
}

#endif
// ---------------------------------------------------------------- launchers (C linkage)
template <int NL>
static ModArgs<NL> make_args(const gecm_modconst *mc)
{
    ModArgs<NL> a;
    for (int i = 0; i < NL; i++) {
        a.m.n[i] = mc->n[i];
        a.m.kp[i] = mc->kp[i];
        a.one.v[i] = mc->one[i];
    }
    a.m.rho = mc->rho;
    return a;
}

template <int NL>
static S2Const<NL> make_s2(const gecm_modconst *mc)
{
    S2Const<NL> k;
    for (int i = 0; i < NL; i++) {
        k.m.n[i] = mc->n[i];
        k.m.kp[i] = mc->kp[i];
        k.one.v[i] = mc->one[i];
        k.r3.v[i] = mc->r3[i];
    }
    k.m.rho = mc->rho;
    k.inv_iters = mc->inv_iters;
    return k;
}

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

// the hash of the sources this object was compiled from (Makefile: K_SHA); gecm_dev.hip collects them
#ifndef GECM_MANIFEST
#define GECM_MANIFEST "unset"
#endif
extern "C" const char *CAT(CAT(CAT(gecm_manifest_k_, GECM_NL), _p), GECM_PART)(void) { return GECM_MANIFEST; }

#if GECM_HAS_PART(1)
extern "C" void CAT(gecm_launch_stage1_, GECM_NL)(void *stream, const gecm_modconst *mc, const uint32_t *tape,
                                                   uint32_t tape_len, uint32_t *X, uint32_t *Z,
                                                   const uint32_t *S, size_t stride)
{
    hipLaunchKernelGGL(k_stage1<GECM_NL>, dim3((unsigned)(stride / 64)), dim3(64), 0, (hipStream_t)stream, tape,
                       tape_len, X, Z, S, stride, make_args<GECM_NL>(mc));
}

template <class MOD>
static void launch_stage1_special(void *stream, const gecm_modconst *mc, const uint32_t *tape, uint32_t tape_len,
                                  uint32_t *X, uint32_t *Z, const uint32_t *S, size_t stride, int lanes)
{
    ModArgsS<GECM_NL, MOD> a;
    for (int i = 0; i < GECM_NL; i++) {
        a.m.n[i] = mc->n[i];
        a.m.kp[i] = mc->kp[i];
        a.one.v[i] = mc->one[i];
    }
    a.m.rho = mc->rho;
    if (lanes == 2)
        hipLaunchKernelGGL((k_stage1_pair_f<GECM_NL, MOD>), dim3((unsigned)(stride / 32)), dim3(64), 0,
                           (hipStream_t)stream, tape, tape_len, X, Z, S, stride, a);
    else
        hipLaunchKernelGGL((k_stage1_f<GECM_NL, MOD>), dim3((unsigned)(stride / 64)), dim3(64), 0, (hipStream_t)stream,
                           tape, tape_len, X, Z, S, stride, a);
}

/* form: +1 = modulus 2^k - 1 (F-form), -1 = modulus 2^k + 1 (P-form), 2 = modulus 2^k - c (C-form) */
extern "C" void CAT(gecm_launch_stage1_f_, GECM_NL)(void *stream, const gecm_modconst *mc, const uint32_t *tape,
                                                     uint32_t tape_len, uint32_t *X, uint32_t *Z,
                                                     const uint32_t *S, size_t stride, int lanes, int form)
{
    if (form == 2) {
        if constexpr (FPolicy<GECM_NL>::NF >= 3) launch_stage1_special<ModC<GECM_NL>>(stream, mc, tape, tape_len, X, Z, S, stride, lanes);
    } else if (form > 0) launch_stage1_special<ModF<GECM_NL>>(stream, mc, tape, tape_len, X, Z, S, stride, lanes);
    else launch_stage1_special<ModP<GECM_NL>>(stream, mc, tape, tape_len, X, Z, S, stride, lanes);
}

extern "C" int CAT(gecm_fform_generic_limbs_, GECM_NL)(void) { return FPolicy<GECM_NL>::G; }

extern "C" void CAT(gecm_launch_stage1_pair_, GECM_NL)(void *stream, const gecm_modconst *mc, const uint32_t *tape,
                                                        uint32_t tape_len, uint32_t *X, uint32_t *Z,
                                                        const uint32_t *S, size_t stride)
{
    hipLaunchKernelGGL(k_stage1_pair<GECM_NL>, dim3((unsigned)(stride / 32)), dim3(64), 0, (hipStream_t)stream,
                       tape, tape_len, X, Z, S, stride, make_args<GECM_NL>(mc));
}

/* returns 0 if launched, -1 if this limb count has no eight-lane kernel.  modq = device array of 80 words:
 * limbs 0..39 of N then of K' (zero padded), read per lane. */
extern "C" int CAT(gecm_launch_stage1_quad_, GECM_NL)(void *stream, const gecm_modconst *mc, const uint32_t *tape,
                                                      uint32_t tape_len, uint32_t *X, uint32_t *Z,
                                                      const uint32_t *S, size_t stride, const uint32_t *modq)
{
#if GECM_HAS_QUAD
    hipLaunchKernelGGL(k_stage1_quad<GECM_NL>, dim3((unsigned)(stride / 32)), dim3(256), 0, (hipStream_t)stream, tape,   // stride: a multiple of 64
                       tape_len, X, Z, S, stride, modq, mc->rho);
    hipLaunchKernelGGL(k_canon<GECM_NL>, dim3((unsigned)(stride / 64)), dim3(64), 0, (hipStream_t)stream, X, Z, stride,
                       make_args<GECM_NL>(mc));
    return 0;
#else
    (void)stream; (void)mc; (void)tape; (void)tape_len; (void)X; (void)Z; (void)S; (void)stride; (void)modq;
    return -1;
#endif
}

extern "C" void CAT(gecm_launch_canon_, GECM_NL)(void *stream, const gecm_modconst *mc, uint32_t *X, uint32_t *Z,
                                                  size_t stride)
{
    hipLaunchKernelGGL(k_canon<GECM_NL>, dim3((unsigned)(stride / 64)), dim3(64), 0, (hipStream_t)stream, X, Z, stride,
                       make_args<GECM_NL>(mc));
}

extern "C" void CAT(gecm_launch_from_mont_, GECM_NL)(void *stream, const gecm_modconst *mc, const uint32_t *X,
                                                      const uint32_t *Z, uint32_t *ox, uint32_t *oz,
                                                      size_t stride)
{
    hipLaunchKernelGGL(k_from_mont<GECM_NL>, dim3((unsigned)(stride / 64)), dim3(64), 0, (hipStream_t)stream, X, Z,
                       ox, oz, stride, make_args<GECM_NL>(mc));
}

extern "C" void CAT(gecm_launch_l0_, GECM_NL)(void *stream, const gecm_modconst *mc, int op, const uint32_t *A,
                                               const uint32_t *B, uint32_t *C, uint32_t *D, size_t stride,
                                               const uint32_t *fix)
{
    Fe<GECM_NL> f;
    for (int i = 0; i < GECM_NL; i++) f.v[i] = fix[i];
    hipLaunchKernelGGL(k_l0<GECM_NL>, dim3((unsigned)(stride / 64)), dim3(64), 0, (hipStream_t)stream, op, A, B, C,
                       D, stride, make_args<GECM_NL>(mc), f);
}
#endif
#if GECM_HAS_PART(2)
extern "C" void CAT(gecm_launch_s2_init_, GECM_NL)(void *stream, const gecm_modconst *mc, const gecm_s2_init_args *h)
{
    S2InitArgs a;
    a.X = h->X; a.Z = h->Z; a.S = h->S; a.PbX = h->PbX; a.bx = h->bx; a.bz = h->bz; a.bp = h->bp;
    a.PdX = h->PdX; a.PdZ = h->PdZ; a.acc = h->acc; a.fail = h->fail; a.keep = h->keep;
    a.umax = h->umax; a.D = h->D; a.npb = h->npb; a.stride = h->stride;
    a.K = h->K; a.tgt = h->tgt; a.tgt_off = h->tgt_off; a.kbx = h->kbx; a.kbz = h->kbz; a.kbp = h->kbp;
    a.PdKX = h->PdKX; a.PdKZ = h->PdKZ;
    if (h->K > 1)
        hipLaunchKernelGGL(k_s2_init_k<GECM_NL>, dim3((unsigned)(h->stride / 64 * h->K)), dim3(64), 0, (hipStream_t)stream, a,
                           make_s2<GECM_NL>(mc));
    else
        hipLaunchKernelGGL(k_s2_init<GECM_NL>, dim3((unsigned)(h->stride / 64)), dim3(64), 0, (hipStream_t)stream, a,
                           make_s2<GECM_NL>(mc));
}

extern "C" void CAT(gecm_launch_s2_pair_, GECM_NL)(void *stream, const gecm_modconst *mc, const gecm_s2_pair_args *h)
{
    S2PairArgs a;
    a.X = h->X; a.Z = h->Z; a.S = h->S; a.PbX = h->PbX; a.npb = h->npb; a.PdX = h->PdX; a.PdZ = h->PdZ;
    a.gx = h->gx; a.gz = h->gz; a.gp = h->gp; a.ring = h->ring; a.acc = h->acc; a.fail = h->fail;
    a.steps = h->steps; a.nsteps = h->nsteps; a.D = h->D; a.G = h->G; a.ring_size = h->ring_size; a.A0 = h->A0;
    a.stride = h->stride;
    a.K = h->K; a.Gs = h->Gs; a.kgx = h->kgx; a.kgz = h->kgz; a.kgp = h->kgp; a.PdKX = h->PdKX; a.PdKZ = h->PdKZ;
    // the tape on the host decides the launch sequence: one k_s2_gen per "generate" mark, one
    // k_s2_pairs per run of pairs between marks (~84 + 84 launches per 1e8 range)
    const dim3 grid((unsigned)(h->stride / 64)), block(64);
    const dim3 pgrid((unsigned)(h->stride / 64), h->slices ? h->slices : 1);
    const S2Const<GECM_NL> k = make_s2<GECM_NL>(mc);
    // a "generate" mark with bit 31 set in its count is a single-chain chunk (the reference's last batch of the range,
    // gecm_stage2_pair); the others use K sub-sequences per curve when the batch is small (h->K > 1)
    const dim3 kgrid((unsigned)(h->stride / 64 * (h->K ? h->K : 1)));
    uint32_t generated = 0, i = 0, kprev = 1;
    while (i < h->nsteps) {
        if (h->host_steps[2 * i] == S2_STEP_GEN) {
            const uint32_t word = h->host_steps[2 * i + 1];
            const uint32_t n = word & 0x7fffffffu;
            if (h->K > 1 && !(word & 0x80000000u)) {
                hipLaunchKernelGGL(k_s2_gen_k<GECM_NL>, kgrid, block, 0, (hipStream_t)stream, a, generated, n, k);
                kprev = h->K;
            } else {
                hipLaunchKernelGGL(k_s2_gen<GECM_NL>, grid, block, 0, (hipStream_t)stream, a, generated, n, generated ? kprev : 1u, k);
                kprev = 1;
            }
            generated += n;
            i++;
        } else {
            uint32_t j = i;
            while (j < h->nsteps && h->host_steps[2 * j] != S2_STEP_GEN) j++;
            hipLaunchKernelGGL(k_s2_pairs<GECM_NL>, pgrid, block, 0, (hipStream_t)stream, a, i, j - i, k);
            i = j;
        }
    }
    if (h->slices > 1)
        hipLaunchKernelGGL(k_s2_merge<GECM_NL>, grid, block, 0, (hipStream_t)stream, h->acc, h->slices, h->stride, 0, k);
}

extern "C" void CAT(gecm_launch_s2_acc_init_, GECM_NL)(void *stream, const gecm_modconst *mc, uint32_t *acc,
                                                        uint32_t slices, size_t stride)
{
    hipLaunchKernelGGL(k_s2_merge<GECM_NL>, dim3((unsigned)(stride / 64)), dim3(64), 0, (hipStream_t)stream, acc, slices,
                       stride, 1, make_s2<GECM_NL>(mc));
}
#endif
#if GECM_HAS_PART(1)
extern "C" void CAT(gecm_launch_gcd_scan_, GECM_NL)(void *stream, const gecm_modconst *mc, const uint32_t *V,
                                                     uint32_t *G, uint32_t *flags, size_t stride)
{
    hipLaunchKernelGGL(k_gcd_scan<GECM_NL>, dim3((unsigned)(stride / 64)), dim3(64), 0, (hipStream_t)stream, V, G, flags,
                       stride, make_s2<GECM_NL>(mc));
}
#endif
