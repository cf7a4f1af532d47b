// gecm_field.hpp — device-side modular arithmetic for the MI355X ECM engine (gfx950 only).
//
// Replaces, for the GPU, the reference's L0 layer:
//   vecmulmod52 (vecarith52.c:2438-3074), vecsqrmod52 (:3317-4548), vecaddmod52 (:4550-4611),
//   vecsubmod52 (:4684-4723), vec_simul_addsub52 (:4877-4968) and the 32-bit twins in vecarith.c.
//
// Design: ONE CURVE PER LANE, 64 curves per wavefront, no cross-lane traffic.
// A residue is NL limbs of 28 bits held in 32-bit VGPRs; the modulus N, the subtraction bias K'
// and rho are wave-uniform (kernel arguments -> SGPRs).  The multiply is product-scanning
// (column-wise) Montgomery with ONE 64-bit column accumulator fed by v_mad_u64_u32:
//   column c:  acc += sum_{i+j=c} a_i*b_j + sum_{i+j=c} q_i*N_j ;  q_c = (acc*rho) mod 2^28
// 28-bit limbs leave 8 bits of headroom in the 64-bit accumulator, so a column of up to 25
// (lazy operands) / 51 (normalised operands) limbs never overflows and needs no carry
// instructions; measured on gfx950 v_mad_u64_u32 issues every ~4.7 cycles per wave64 at
// >=2 waves/SIMD, the same rate as v_fma_f64, which makes this denser than the reference's
// FP64 hi/lo 52-bit trick (profiles/r01_valu_ubench_gfx950.txt).
//
// Lazy reduction: R = 2^(28*NL) >= 32*N.  K = 2^k*N in [R/32, R/16).  Every multiply output
// is < 0.675*K with normalised limbs (< 2^28); add is limb-wise (no carry), sub is
// a + K' - b limb-wise where K' is K written with every limb in [2^28-1, 2^29) so no limb
// goes negative.  Values stay bounded (fixed point of M = (M+K)^2/R + N), so there is no
// conditional subtraction anywhere in the ladder; canonical residues in [0,N) — what the
// reference returns after every operation — are produced once, when results leave the device.
// Because every intermediate is the same element of Z/N as in the reference, outputs are
// bit-identical (SURVEY.md §8a semantics note).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GECM_LIMB_BITS 28
#define GECM_LIMB_MASK 0x0FFFFFFFu

template <int NL>
struct ModK {
    uint32_t n[NL];   // N, 28-bit limbs
    uint32_t kp[NL];  // K' = bias for subtraction (multiple of N, limbs in [2^28-1, 2^29))
    uint32_t rho;     // -N^-1 mod 2^28
};

template <int NL>
struct Fe {
    uint32_t v[NL];
};

__device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b, uint64_t c)
{
    // lowers to v_mad_u64_u32
    return (uint64_t)a * (uint64_t)b + c;
}

// r = a*b/R mod N (lazy: r < 0.675K, limbs < 2^28).  r may alias a or b.
template <int NL>
__device__ __forceinline__ void fe_mul(Fe<NL> &r, const Fe<NL> &a, const Fe<NL> &b, const ModK<NL> &m)
{
    uint32_t q[NL];
    Fe<NL> o;
    uint64_t acc = 0;
#pragma unroll
    for (int c = 0; c < NL; c++) {
#pragma unroll
        for (int i = 0; i <= c; i++) acc = mad64(a.v[i], b.v[c - i], acc);
#pragma unroll
        for (int i = 0; i < c; i++) acc = mad64(q[i], m.n[c - i], acc);
        q[c] = ((uint32_t)acc * m.rho) & GECM_LIMB_MASK;
        acc = mad64(q[c], m.n[0], acc);
        acc >>= GECM_LIMB_BITS;
    }
#pragma unroll
    for (int c = NL; c < 2 * NL; c++) {
#pragma unroll
        for (int i = c - NL + 1; i < NL; i++) acc = mad64(a.v[i], b.v[c - i], acc);
#pragma unroll
        for (int i = c - NL + 1; i < NL; i++) acc = mad64(q[i], m.n[c - i], acc);
        o.v[c - NL] = (c == 2 * NL - 1) ? (uint32_t)acc : ((uint32_t)acc & GECM_LIMB_MASK);
        acc >>= GECM_LIMB_BITS;
    }
    r = o;
}

// r = a*a/R mod N; cross terms computed once with a doubled operand (same value as fe_mul(a,a),
// as vecsqrmod52 == vecmulmod52(a,a) in the reference).
template <int NL>
__device__ __forceinline__ void fe_sqr(Fe<NL> &r, const Fe<NL> &a, const ModK<NL> &m)
{
    uint32_t q[NL];
    uint32_t a2[NL];
    Fe<NL> o;
#pragma unroll
    for (int i = 0; i < NL; i++) a2[i] = a.v[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int c = 0; c < NL; c++) {
#pragma unroll
        for (int i = 0; 2 * i < c; i++) acc = mad64(a.v[i], a2[c - i], acc);
        if ((c & 1) == 0) acc = mad64(a.v[c / 2], a.v[c / 2], acc);
#pragma unroll
        for (int i = 0; i < c; i++) acc = mad64(q[i], m.n[c - i], acc);
        q[c] = ((uint32_t)acc * m.rho) & GECM_LIMB_MASK;
        acc = mad64(q[c], m.n[0], acc);
        acc >>= GECM_LIMB_BITS;
    }
#pragma unroll
    for (int c = NL; c < 2 * NL; c++) {
#pragma unroll
        for (int i = c - NL + 1; 2 * i < c; i++) acc = mad64(a.v[i], a2[c - i], acc);
        if ((c & 1) == 0 && c / 2 < NL) acc = mad64(a.v[c / 2], a.v[c / 2], acc);
#pragma unroll
        for (int i = c - NL + 1; i < NL; i++) acc = mad64(q[i], m.n[c - i], acc);
        o.v[c - NL] = (c == 2 * NL - 1) ? (uint32_t)acc : ((uint32_t)acc & GECM_LIMB_MASK);
        acc >>= GECM_LIMB_BITS;
    }
    r = o;
}

// Parallel (carry-save) renormalisation: limbs < 2^30 in, limbs < 2^28 + 4 out, value unchanged.
template <int NL>
__device__ __forceinline__ void fe_weak_norm(Fe<NL> &r)
{
    Fe<NL> o;
    o.v[0] = r.v[0] & GECM_LIMB_MASK;
#pragma unroll
    for (int i = 1; i < NL - 1; i++) o.v[i] = (r.v[i] & GECM_LIMB_MASK) + (r.v[i - 1] >> GECM_LIMB_BITS);
    o.v[NL - 1] = r.v[NL - 1] + (r.v[NL - 2] >> GECM_LIMB_BITS);
    r = o;
}

// Column sums of fe_mul are bounded by NL*(La*Lb + 2^56); with lazy operands
// (sub output limbs < 3*2^28) that is safe up to NL = 25.  Above that, sub outputs are
// renormalised (3 cheap VALU ops per limb) so all operands are < 2^29: safe up to NL = 51.
template <int NL>
struct LazyPolicy {
    static constexpr bool norm_sub = (NL > 25);
};

// r = a + b (limb-wise, no carry).  Operands must have normalised limbs.
template <int NL>
__device__ __forceinline__ void fe_add(Fe<NL> &r, const Fe<NL> &a, const Fe<NL> &b)
{
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = a.v[i] + b.v[i];
}

// r = a - b + K  (limb-wise, never negative).  b must have normalised limbs.
template <int NL>
__device__ __forceinline__ void fe_sub(Fe<NL> &r, const Fe<NL> &a, const Fe<NL> &b, const ModK<NL> &m)
{
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = a.v[i] + m.kp[i] - b.v[i];
    if (LazyPolicy<NL>::norm_sub) fe_weak_norm(r);
}

// Full carry propagation: limbs < 2^28 (top limb takes the rest).
template <int NL>
__device__ __forceinline__ void fe_full_norm(Fe<NL> &r)
{
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        uint32_t t = r.v[i] + c;
        r.v[i] = t & GECM_LIMB_MASK;
        c = t >> GECM_LIMB_BITS;
    }
    r.v[NL - 1] += c;
}

// Canonical residue in [0, N) of a fully normalised value < 2N.
template <int NL>
__device__ __forceinline__ void fe_cond_sub_n(Fe<NL> &r, const ModK<NL> &m)
{
    Fe<NL> t;
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        uint32_t d = r.v[i] - m.n[i] - borrow;
        borrow = (d >> 31) & 1u;              // limbs < 2^29 so bit 31 set <=> negative
        t.v[i] = (i == NL - 1) ? d : (d & GECM_LIMB_MASK);
    }
    if (!borrow) r = t;
}

// x*R -> x canonical: the reference's "vecmulmod(P->X, one)" de-Montgomeryisation
// (ecm.c:1327-1331) followed by the reduction to [0,N) every reference op performs.
template <int NL>
__device__ __forceinline__ void fe_from_mont_canonical(Fe<NL> &r, const Fe<NL> &a, const ModK<NL> &m)
{
    Fe<NL> one;
#pragma unroll
    for (int i = 0; i < NL; i++) one.v[i] = (i == 0) ? 1u : 0u;
    fe_mul(r, a, one, m);          // < N + 1, normalised limbs
    fe_cond_sub_n(r, m);
}

// Canonical residue of a lazy Montgomery-form value, staying in Montgomery form:
// mont(a, R mod N) = a, < N + K/16... then one conditional subtract.  rmodn must be canonical.
template <int NL>
__device__ __forceinline__ void fe_canonical_mont(Fe<NL> &r, const Fe<NL> &a, const Fe<NL> &rmodn, const ModK<NL> &m)
{
    fe_mul(r, a, rmodn, m);        // a*(R mod N)/R = a mod N, value < M*N/R + N < 2N
    fe_cond_sub_n(r, m);
}

// coalesced SoA access: element [limb][curve], consecutive lanes -> consecutive curves
template <int NL>
__device__ __forceinline__ void fe_load(Fe<NL> &r, const uint32_t *__restrict__ base, size_t stride, size_t idx)
{
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = base[(size_t)i * stride + idx];
}

template <int NL>
__device__ __forceinline__ void fe_store(uint32_t *__restrict__ base, size_t stride, size_t idx, const Fe<NL> &r)
{
#pragma unroll
    for (int i = 0; i < NL; i++) base[(size_t)i * stride + idx] = r.v[i];
}
