// gecm_field.hpp — device-side modular arithmetic for the MI355X ECM engine (gfx950 only).
//
// Replaces, for the GPU, the reference's L0 layer:
//   vecmulmod52 (vecarith52.c:2438-3074), vecsqrmod52 (:3317-4548), vecaddmod52 (:4550-4611),
//   vecsubmod52 (:4684-4723), vec_simul_addsub52 (:4877-4968) and the 32-bit twins in vecarith.c.
//
// Design: ONE CURVE PER LANE, 64 curves per wavefront, no cross-lane traffic.
// (Batches too small to fill the chip use the same arithmetic with a curve's X and Z on two adjacent lanes —
// gecm_curve.hpp, run_tape_pair — or a row-wise multiply with a residue spread over four lanes — gecm_quad.hpp.)
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

// ---- the multiply-accumulate chain --------------------------------------------------------
// acc += sum x[k]*y[k] as K back-to-back v_mad_u64_u32 in ONE asm statement.  Inline asm on
// purpose: from C++ LLVM reassociates the column sums into partial chains joined by 64-bit adds
// (v_lshl_add_u64, one or two per column, +7% instructions) and stretches live ranges.  A serial
// dependent chain costs nothing on gfx950 (a dependent v_mad_u64_u32 issues as fast as an
// independent one: tools/mad_chain_ubench.hip), and grouping up to 8 per statement avoids the
// s_nop the compiler pads after every asm statement.  The carry-out goes to vcc and is ignored
// (28-bit limbs: the 64-bit column sum cannot overflow).  `_s`: y[] are wave-uniform (limbs of N)
// and are read straight from SGPRs.
#ifndef GECM_MAD_CHUNK
#define GECM_MAD_CHUNK 14     // mads per asm statement (<= 14: an asm statement takes at most 30 operands)
#endif

// The statements for K = 1 .. 14 multiply-adds are spelled by the preprocessor: GECM_MAD_TEXT_k is the text of k
// instructions (operands %1,%2 / %3,%4 / ...), GECM_MAD_OPS_k(Y) their operand list with constraint Y on the second
// factor ("v": a register per lane, "s": a wave-uniform value read straight from an SGPR).
#define GECM_MAD_LINE(a, b) "v_mad_u64_u32 %0, vcc, %" #a ", %" #b ", %0\n\t"
#define GECM_MAD_TEXT_1 GECM_MAD_LINE(1, 2)
#define GECM_MAD_TEXT_2 GECM_MAD_TEXT_1 GECM_MAD_LINE(3, 4)
#define GECM_MAD_TEXT_3 GECM_MAD_TEXT_2 GECM_MAD_LINE(5, 6)
#define GECM_MAD_TEXT_4 GECM_MAD_TEXT_3 GECM_MAD_LINE(7, 8)
#define GECM_MAD_TEXT_5 GECM_MAD_TEXT_4 GECM_MAD_LINE(9, 10)
#define GECM_MAD_TEXT_6 GECM_MAD_TEXT_5 GECM_MAD_LINE(11, 12)
#define GECM_MAD_TEXT_7 GECM_MAD_TEXT_6 GECM_MAD_LINE(13, 14)
#define GECM_MAD_TEXT_8 GECM_MAD_TEXT_7 GECM_MAD_LINE(15, 16)
#define GECM_MAD_TEXT_9 GECM_MAD_TEXT_8 GECM_MAD_LINE(17, 18)
#define GECM_MAD_TEXT_10 GECM_MAD_TEXT_9 GECM_MAD_LINE(19, 20)
#define GECM_MAD_TEXT_11 GECM_MAD_TEXT_10 GECM_MAD_LINE(21, 22)
#define GECM_MAD_TEXT_12 GECM_MAD_TEXT_11 GECM_MAD_LINE(23, 24)
#define GECM_MAD_TEXT_13 GECM_MAD_TEXT_12 GECM_MAD_LINE(25, 26)
#define GECM_MAD_TEXT_14 GECM_MAD_TEXT_13 GECM_MAD_LINE(27, 28)
#define GECM_MAD_OPS_1(Y) "v"(x[0]), Y(y[0])
#define GECM_MAD_OPS_2(Y) GECM_MAD_OPS_1(Y), "v"(x[1]), Y(y[1])
#define GECM_MAD_OPS_3(Y) GECM_MAD_OPS_2(Y), "v"(x[2]), Y(y[2])
#define GECM_MAD_OPS_4(Y) GECM_MAD_OPS_3(Y), "v"(x[3]), Y(y[3])
#define GECM_MAD_OPS_5(Y) GECM_MAD_OPS_4(Y), "v"(x[4]), Y(y[4])
#define GECM_MAD_OPS_6(Y) GECM_MAD_OPS_5(Y), "v"(x[5]), Y(y[5])
#define GECM_MAD_OPS_7(Y) GECM_MAD_OPS_6(Y), "v"(x[6]), Y(y[6])
#define GECM_MAD_OPS_8(Y) GECM_MAD_OPS_7(Y), "v"(x[7]), Y(y[7])
#define GECM_MAD_OPS_9(Y) GECM_MAD_OPS_8(Y), "v"(x[8]), Y(y[8])
#define GECM_MAD_OPS_10(Y) GECM_MAD_OPS_9(Y), "v"(x[9]), Y(y[9])
#define GECM_MAD_OPS_11(Y) GECM_MAD_OPS_10(Y), "v"(x[10]), Y(y[10])
#define GECM_MAD_OPS_12(Y) GECM_MAD_OPS_11(Y), "v"(x[11]), Y(y[11])
#define GECM_MAD_OPS_13(Y) GECM_MAD_OPS_12(Y), "v"(x[12]), Y(y[12])
#define GECM_MAD_OPS_14(Y) GECM_MAD_OPS_13(Y), "v"(x[13]), Y(y[13])
#define GECM_MAD_CASE(k, Y) \
    if constexpr (K == k) asm(GECM_MAD_TEXT_##k : "+v"(acc) : GECM_MAD_OPS_##k(Y) : "vcc");
#define GECM_MAD_CASES(Y)                                                                                              \
    GECM_MAD_CASE(1, Y) GECM_MAD_CASE(2, Y) GECM_MAD_CASE(3, Y) GECM_MAD_CASE(4, Y) GECM_MAD_CASE(5, Y)                    \
    GECM_MAD_CASE(6, Y) GECM_MAD_CASE(7, Y) GECM_MAD_CASE(8, Y) GECM_MAD_CASE(9, Y) GECM_MAD_CASE(10, Y)                   \
    GECM_MAD_CASE(11, Y) GECM_MAD_CASE(12, Y) GECM_MAD_CASE(13, Y) GECM_MAD_CASE(14, Y)

// acc += sum x[k] * y[k], both factors in vector registers
template <int K>
__device__ __forceinline__ void mad_chain_v(uint64_t &acc, const uint32_t (&x)[K], const uint32_t (&y)[K])
{
    static_assert(K >= 1 && K <= 14, "one asm statement takes at most 30 operands");
#ifdef GECM_CXX_MAD
#pragma unroll
    for (int k = 0; k < K; k++) acc += (uint64_t)x[k] * (uint64_t)y[k];
#else
    GECM_MAD_CASES("v")
#endif
}

// the same with wave-uniform y[k] (limbs of the modulus)
template <int K>
__device__ __forceinline__ void mad_chain_s(uint64_t &acc, const uint32_t (&x)[K], const uint32_t (&y)[K])
{
    static_assert(K >= 1 && K <= 14, "one asm statement takes at most 30 operands");
#ifdef GECM_CXX_MAD
#pragma unroll
    for (int k = 0; k < K; k++) acc += (uint64_t)x[k] * (uint64_t)y[k];
#else
    GECM_MAD_CASES("s")
#endif
}

template <int K> struct IC { static constexpr int value = K; };
template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I0 < I1) {
        f(IC<I0>{});
        static_for<I0 + 1, I1>(f);
    }
}

// acc += sum_{i in [I0,I1)} a[i] * b[C-i]
template <int C, int I0, int I1, int NA, int NB>
__device__ __forceinline__ void col_vv(uint64_t &acc, const uint32_t (&a)[NA], const uint32_t (&b)[NB])
{
    constexpr int K = I1 - I0;
    if constexpr (K > GECM_MAD_CHUNK) {
        col_vv<C, I0, I0 + GECM_MAD_CHUNK>(acc, a, b);
        col_vv<C, I0 + GECM_MAD_CHUNK, I1>(acc, a, b);
    } else if constexpr (K > 0) {
        uint32_t x[K], y[K];
#pragma unroll
        for (int k = 0; k < K; k++) { x[k] = a[I0 + k]; y[k] = b[C - I0 - k]; }
        mad_chain_v<K>(acc, x, y);
    }
}

// acc += sum_{i in [I0,I1)} q[i] * n[C-i]   (n wave-uniform)
template <int C, int I0, int I1, int NA, int NB>
__device__ __forceinline__ void col_vs(uint64_t &acc, const uint32_t (&q)[NA], const uint32_t (&n)[NB])
{
    constexpr int K = I1 - I0;
    if constexpr (K > GECM_MAD_CHUNK) {
        col_vs<C, I0, I0 + GECM_MAD_CHUNK>(acc, q, n);
        col_vs<C, I0 + GECM_MAD_CHUNK, I1>(acc, q, n);
    } else if constexpr (K > 0) {
        uint32_t x[K], y[K];
#pragma unroll
        for (int k = 0; k < K; k++) { x[k] = q[I0 + k]; y[k] = n[C - I0 - k]; }
        mad_chain_s<K>(acc, x, y);
    }
}

// r = a*b/R mod N (lazy: r < 0.675K, limbs < 2^28).  r may alias a or b.
template <int NL>
__device__ __forceinline__ void fe_mul(Fe<NL> &r, const Fe<NL> &a, const Fe<NL> &b, const ModK<NL> &m)
{
    uint32_t q[NL];
    Fe<NL> o;
    uint64_t acc = 0;
    static_for<0, NL>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        col_vv<c, 0, c + 1>(acc, a.v, b.v);
        col_vs<c, 0, c>(acc, q, m.n);
        q[c] = ((uint32_t)acc * m.rho) & GECM_LIMB_MASK;
        col_vs<c, c, c + 1>(acc, q, m.n);
        acc >>= GECM_LIMB_BITS;
    });
    static_for<NL, 2 * NL>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        col_vv<c, c - NL + 1, NL>(acc, a.v, b.v);
        col_vs<c, c - NL + 1, NL>(acc, q, m.n);
        o.v[c - NL] = (c == 2 * NL - 1) ? (uint32_t)acc : ((uint32_t)acc & GECM_LIMB_MASK);
        acc >>= GECM_LIMB_BITS;
    });
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
    static_for<0, NL>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        col_vv<c, 0, (c + 1) / 2>(acc, a.v, a2);                     // i < c-i
        if constexpr ((c & 1) == 0) col_vv<c, c / 2, c / 2 + 1>(acc, a.v, a.v);
        col_vs<c, 0, c>(acc, q, m.n);
        q[c] = ((uint32_t)acc * m.rho) & GECM_LIMB_MASK;
        col_vs<c, c, c + 1>(acc, q, m.n);
        acc >>= GECM_LIMB_BITS;
    });
    static_for<NL, 2 * NL>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        col_vv<c, c - NL + 1, (c + 1) / 2>(acc, a.v, a2);
        if constexpr ((c & 1) == 0 && c / 2 < NL) col_vv<c, c / 2, c / 2 + 1>(acc, a.v, a.v);
        col_vs<c, c - NL + 1, NL>(acc, q, m.n);
        o.v[c - NL] = (c == 2 * NL - 1) ? (uint32_t)acc : ((uint32_t)acc & GECM_LIMB_MASK);
        acc >>= GECM_LIMB_BITS;
    });
    r = o;
}

// ---- moduli of the form 2^k - 1 ("F-form") ---------------------------------------------------
// For N | 2^k - 1 stage 1 may run modulo Mw = 2^k - 1 instead of N (every residue mod Mw reduces to the
// right residue mod N; the host reduces once at the end).  Mw in 28-bit limbs is F = 2^28 - 1 in every
// limb below the one that holds bit k, so in the REDC half of the product scanning
//     sum_i q_i * n_(c-i)  =  F * (sum of the q_i whose partner limb is an F)  +  (at most G generic terms)
// with F*T one multiply-add per class sum (FPolicy): the q*N triangle of NL^2 multiply-adds shrinks to
// about (G+NC)*2NL, and rho = -Mw^-1 mod 2^28 = 1 makes the Montgomery digit a mask.  This is the SAME REDC with the
// same digits q_i, so every result is the integer the generic fe_mul returns for this modulus; bounds,
// lazy add/sub and K' are unchanged.  The reference switches such inputs to a folding multiply
// (vecmulmod52_mersenne, vecarith52.c:284-1031); this is the MI355X counterpart, kept inside REDC.
// G = number of top limbs read from m.n (they hold the partial limb and possibly zeros): the limb count
// is the smallest BUILT size with 28*NL >= k+5, so bit k can sit up to (NL - previous built size) limbs
// below the top.  The host enables the F-form only when limbs 0 .. NL-G-1 of Mw are all F.
template <int NL>
struct ModF : ModK<NL> {};

template <int NL>
struct FPolicy {
    static constexpr int G = (NL == 26 || NL == 37) ? 4 : (NL == 8 || NL == 15) ? 2 : 3;
    static constexpr int NF = NL - G;          // limbs 0 .. NF-1 of the modulus are 2^28 - 1
    // The digits paired with F limbs are summed in NC classes (digit i in class i % NC) of at most 15
    // digits, so each class sum fits 32 bits and F*T is ONE v_mad_u64_u32 per class (64-bit shift/subtract
    // sequences cost as much as 4-5 multiply-adds on gfx950: tools/fform_check.py history in DESIGN.md).
    static constexpr int NC = (NF + 14) / 15;
};

template <int NL>
struct FSums {
    uint32_t t[FPolicy<NL>::NC];
};

// REDC part of column C.  T = class sums of the digits currently paired with an F limb (excluding q_C).
template <int C, int NL>
__device__ __forceinline__ void redc_f_col(uint64_t &acc, FSums<NL> &T, const uint32_t (&q)[NL], const uint32_t (&n)[NL])
{
    constexpr int NF = FPolicy<NL>::NF, NC = FPolicy<NL>::NC;
    if constexpr (C - NF >= 0 && C - NF <= NL - 1) T.t[(C - NF) % NC] -= q[C - NF];   // its partner is now limb NF: generic
    // class sums that can be non-empty here: digits i in [max(0, C-NF+1), min(C-1, NL-1)] (consecutive);
    // generic terms: digits i in [max(0, C-NL+1), min(C-NF, NL-1)], partner limb C-i in [NF, NL-1].
    // All of them go into ONE multiply-add chain (one asm statement: no s_nop padding in between).
    constexpr int w_lo = (C - NF + 1 > 0) ? C - NF + 1 : 0;
    constexpr int w_hi = (C - 1 < NL - 1) ? C - 1 : NL - 1;
    constexpr int wn = (w_hi >= w_lo) ? w_hi - w_lo + 1 : 0;
    constexpr int cnt = wn < NC ? wn : NC;
    constexpr int i_lo = (C - NL + 1 > 0) ? C - NL + 1 : 0;
    constexpr int i_hi = (C - NF < NL - 1) ? C - NF : NL - 1;
    constexpr int gen = (i_hi >= i_lo) ? i_hi - i_lo + 1 : 0;
    if constexpr (cnt + gen > 0) {
        uint32_t x[cnt + gen], y[cnt + gen];
#pragma unroll
        for (int j = 0; j < cnt; j++) {
            x[j] = T.t[(w_lo + j) % NC];
            y[j] = GECM_LIMB_MASK;                                              // F = 2^28 - 1
        }
#pragma unroll
        for (int j = 0; j < gen; j++) {
            x[cnt + j] = q[i_lo + j];
            y[cnt + j] = n[C - i_lo - j];
        }
        mad_chain_s<cnt + gen>(acc, x, y);
    }
}

template <int NL>
__device__ __forceinline__ void fe_mul(Fe<NL> &r, const Fe<NL> &a, const Fe<NL> &b, const ModF<NL> &m)
{
    uint32_t q[NL];
    Fe<NL> o;
    uint64_t acc = 0;
    FSums<NL> T;
#pragma unroll
    for (int j = 0; j < FPolicy<NL>::NC; j++) T.t[j] = 0;
    static_for<0, NL>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        col_vv<c, 0, c + 1>(acc, a.v, b.v);
        redc_f_col<c>(acc, T, q, m.n);
        q[c] = (uint32_t)acc & GECM_LIMB_MASK;                 // rho = 1
        acc = (acc >> GECM_LIMB_BITS) + q[c];                  // (acc + q*F) >> 28
        T.t[c % FPolicy<NL>::NC] += q[c];
    });
    static_for<NL, 2 * NL>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        col_vv<c, c - NL + 1, NL>(acc, a.v, b.v);
        redc_f_col<c>(acc, T, q, m.n);
        o.v[c - NL] = (c == 2 * NL - 1) ? (uint32_t)acc : ((uint32_t)acc & GECM_LIMB_MASK);
        acc >>= GECM_LIMB_BITS;
    });
    r = o;
}

template <int NL>
__device__ __forceinline__ void fe_sqr(Fe<NL> &r, const Fe<NL> &a, const ModF<NL> &m)
{
    uint32_t q[NL];
    uint32_t a2[NL];
    Fe<NL> o;
#pragma unroll
    for (int i = 0; i < NL; i++) a2[i] = a.v[i] << 1;
    uint64_t acc = 0;
    FSums<NL> T;
#pragma unroll
    for (int j = 0; j < FPolicy<NL>::NC; j++) T.t[j] = 0;
    static_for<0, NL>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        col_vv<c, 0, (c + 1) / 2>(acc, a.v, a2);
        if constexpr ((c & 1) == 0) col_vv<c, c / 2, c / 2 + 1>(acc, a.v, a.v);
        redc_f_col<c>(acc, T, q, m.n);
        q[c] = (uint32_t)acc & GECM_LIMB_MASK;
        acc = (acc >> GECM_LIMB_BITS) + q[c];
        T.t[c % FPolicy<NL>::NC] += q[c];
    });
    static_for<NL, 2 * NL>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        col_vv<c, c - NL + 1, (c + 1) / 2>(acc, a.v, a2);
        if constexpr ((c & 1) == 0 && c / 2 < NL) col_vv<c, c / 2, c / 2 + 1>(acc, a.v, a.v);
        redc_f_col<c>(acc, T, q, m.n);
        o.v[c - NL] = (c == 2 * NL - 1) ? (uint32_t)acc : ((uint32_t)acc & GECM_LIMB_MASK);
        acc >>= GECM_LIMB_BITS;
    });
    r = o;
}

// ---- moduli of the form 2^k - c, c odd, 1 < c < 2^52 ("C-form": the reference's pseudo-Mersenne inputs, main.c:432-441,
// for which it folds with vecmulmod52_mersenne, vecarith52.c:284-1031) ------------------------------------------------
// Mw = 2^k - c = (2^k - 1) - (c - 1): in 28-bit limbs the F-form modulus with e0 = (c-1) mod 2^28 taken off limb 0 and
// e1 = (c-1) >> 28 off limb 1 (no borrow: e < 2^28), so in the REDC half of column C
//     sum_i q_i n_(C-i) = [the F-form sum] - e0 * q_C - e1 * q_(C-1),
// and rho = -Mw^-1 = (c mod 2^28)^-1 mod 2^28 is a real multiplier.  Per column that is, on top of the F-form's class
// sums: one v_mul_lo for the digit, one multiply-add q_C * n_0 (n_0 = F - e0, read from the modulus) instead of the
// F-form's "+ q", and one signed multiply-add -e1 * q_(C-1).  Same REDC, same digits, same integers as the generic
// multiply for this modulus; the host reduces modulo N when the points come back.
template <int NL>
struct ModC : ModK<NL> {};

__device__ __forceinline__ void mad_neg(uint64_t &acc, uint32_t e, uint32_t q)      // acc -= e * q   (e < 2^28, q < 2^28)
{
    const int32_t ne = -(int32_t)e;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(ne), "v"(q) : "vcc");
}

// REDC part of column C before the digit: the F-form part with the digits paired with limbs 1 .. NF-1 (limb 0's
// partner is the digit of this column, added by the caller), then the correction for limb 1.
template <int C, int NL>
__device__ __forceinline__ void redc_c_col(uint64_t &acc, FSums<NL> &T, const uint32_t (&q)[NL], const uint32_t (&n)[NL], uint32_t e1)
{
    redc_f_col<C>(acc, T, q, n);
    if constexpr (C >= 1 && C - 1 <= NL - 1) mad_neg(acc, e1, q[C - 1]);
}

template <int NL, bool SQR>
__device__ __forceinline__ void fe_mulsqr_c(Fe<NL> &r, const Fe<NL> &a, const Fe<NL> &b, const ModC<NL> &m)
{
    static_assert(FPolicy<NL>::NF >= 3, "limbs 0, 1 carry the c - 1 correction; at least one pure F limb above them");
    uint32_t q[NL];
    uint32_t a2[NL];
    Fe<NL> o;
    if constexpr (SQR) {
#pragma unroll
        for (int i = 0; i < NL; i++) a2[i] = a.v[i] << 1;
    }
    const uint32_t e1 = GECM_LIMB_MASK - m.n[1];
    uint32_t n0[1] = {m.n[0]};
    uint64_t acc = 0;
    FSums<NL> T;
#pragma unroll
    for (int j = 0; j < FPolicy<NL>::NC; j++) T.t[j] = 0;
    static_for<0, NL>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        if constexpr (SQR) {
            col_vv<c, 0, (c + 1) / 2>(acc, a.v, a2);
            if constexpr ((c & 1) == 0) col_vv<c, c / 2, c / 2 + 1>(acc, a.v, a.v);
        } else {
            col_vv<c, 0, c + 1>(acc, a.v, b.v);
        }
        redc_c_col<c>(acc, T, q, m.n, e1);
        q[c] = ((uint32_t)acc * m.rho) & GECM_LIMB_MASK;
        uint32_t qc[1] = {q[c]};
        mad_chain_s<1>(acc, qc, n0);                            // + q_c * n_0: the column is now 0 mod 2^28
        acc >>= GECM_LIMB_BITS;
        T.t[c % FPolicy<NL>::NC] += q[c];
    });
    static_for<NL, 2 * NL>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        if constexpr (SQR) {
            col_vv<c, c - NL + 1, (c + 1) / 2>(acc, a.v, a2);
            if constexpr ((c & 1) == 0 && c / 2 < NL) col_vv<c, c / 2, c / 2 + 1>(acc, a.v, a.v);
        } else {
            col_vv<c, c - NL + 1, NL>(acc, a.v, b.v);
        }
        redc_c_col<c>(acc, T, q, m.n, e1);
        o.v[c - NL] = (c == 2 * NL - 1) ? (uint32_t)acc : ((uint32_t)acc & GECM_LIMB_MASK);
        acc >>= GECM_LIMB_BITS;
    });
    r = o;
}

template <int NL>
__device__ __forceinline__ void fe_mul(Fe<NL> &r, const Fe<NL> &a, const Fe<NL> &b, const ModC<NL> &m)
{
    fe_mulsqr_c<NL, false>(r, a, b, m);
}

template <int NL>
__device__ __forceinline__ void fe_sqr(Fe<NL> &r, const Fe<NL> &a, const ModC<NL> &m)
{
    fe_mulsqr_c<NL, true>(r, a, a, m);
}

// ---- moduli of the form 2^k + 1 ("P-form") ---------------------------------------------------
// Mw = 2^k + 1 in 28-bit limbs is 1, 0, ..., 0 below the limb that holds bit k, and rho = -Mw^-1 mod 2^28
// = 2^28 - 1.  The REDC half of a column is then: the digit q_c = (-acc) mod 2^28, its product with limb 0
// (an addition), and the generic terms with the top G limbs (only one of which is non-zero).  Same REDC,
// same digits, same integers as the generic multiply for this modulus.  The reference's counterpart is
// the isMersenne == -1 branch of vecmulmod52_mersenne (vecarith52.c:973-1027).
template <int NL>
struct ModP : ModK<NL> {};

template <int C, int NL>
__device__ __forceinline__ void redc_p_col(uint64_t &acc, const uint32_t (&q)[NL], const uint32_t (&n)[NL])
{
    constexpr int NF = FPolicy<NL>::NF;
    constexpr int i_lo = (C - NL + 1 > 0) ? C - NL + 1 : 0;
    constexpr int i_hi = (C - NF < NL - 1) ? C - NF : NL - 1;              // partner limb index C-i in [NF, NL-1]
    if constexpr (i_hi >= i_lo) col_vs<C, i_lo, i_hi + 1>(acc, q, n);
}

template <int NL>
__device__ __forceinline__ void fe_mul(Fe<NL> &r, const Fe<NL> &a, const Fe<NL> &b, const ModP<NL> &m)
{
    uint32_t q[NL];
    Fe<NL> o;
    uint64_t acc = 0;
    static_for<0, NL>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        col_vv<c, 0, c + 1>(acc, a.v, b.v);
        redc_p_col<c>(acc, q, m.n);
        q[c] = (0u - (uint32_t)acc) & GECM_LIMB_MASK;          // rho = -1
        acc = (acc + q[c]) >> GECM_LIMB_BITS;                  // limb 0 of the modulus is 1
    });
    static_for<NL, 2 * NL>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        col_vv<c, c - NL + 1, NL>(acc, a.v, b.v);
        redc_p_col<c>(acc, q, m.n);
        o.v[c - NL] = (c == 2 * NL - 1) ? (uint32_t)acc : ((uint32_t)acc & GECM_LIMB_MASK);
        acc >>= GECM_LIMB_BITS;
    });
    r = o;
}

template <int NL>
__device__ __forceinline__ void fe_sqr(Fe<NL> &r, const Fe<NL> &a, const ModP<NL> &m)
{
    uint32_t q[NL];
    uint32_t a2[NL];
    Fe<NL> o;
#pragma unroll
    for (int i = 0; i < NL; i++) a2[i] = a.v[i] << 1;
    uint64_t acc = 0;
    static_for<0, NL>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        col_vv<c, 0, (c + 1) / 2>(acc, a.v, a2);
        if constexpr ((c & 1) == 0) col_vv<c, c / 2, c / 2 + 1>(acc, a.v, a.v);
        redc_p_col<c>(acc, q, m.n);
        q[c] = (0u - (uint32_t)acc) & GECM_LIMB_MASK;
        acc = (acc + q[c]) >> GECM_LIMB_BITS;
    });
    static_for<NL, 2 * NL>([&](auto ic) {
        constexpr int c = decltype(ic)::value;
        col_vv<c, c - NL + 1, (c + 1) / 2>(acc, a.v, a2);
        if constexpr ((c & 1) == 0 && c / 2 < NL) col_vv<c, c / 2, c / 2 + 1>(acc, a.v, a.v);
        redc_p_col<c>(acc, q, m.n);
        o.v[c - NL] = (c == 2 * NL - 1) ? (uint32_t)acc : ((uint32_t)acc & GECM_LIMB_MASK);
        acc >>= GECM_LIMB_BITS;
    });
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
template <int NL, class MOD>
__device__ __forceinline__ void fe_from_mont_canonical(Fe<NL> &r, const Fe<NL> &a, const MOD &m)
{
    Fe<NL> one;
#pragma unroll
    for (int i = 0; i < NL; i++) one.v[i] = (i == 0) ? 1u : 0u;
    fe_mul(r, a, one, m);          // < N + 1, normalised limbs
    fe_cond_sub_n(r, m);
}

// Canonical residue of a lazy Montgomery-form value, staying in Montgomery form:
// mont(a, R mod N) = a, < N + K/16... then one conditional subtract.  rmodn must be canonical.
template <int NL, class MOD>
__device__ __forceinline__ void fe_canonical_mont(Fe<NL> &r, const Fe<NL> &a, const Fe<NL> &rmodn, const MOD &m)
{
    fe_mul(r, a, rmodn, m);        // a*(R mod N)/R = a mod N, value < M*N/R + N < 2N
    fe_cond_sub_n(r, m);
}

// coalesced SoA access: element [limb][curve], consecutive lanes -> consecutive curves.
// The row pointer (base + limb*stride) is wave-uniform and the lane index is a 32-bit offset, so
// each access is `global_load_dword v, v_off, s[row]` (SGPR base + VGPR offset): no per-limb 64-bit
// address pairs are kept live in VGPRs.
template <int NL>
__device__ __forceinline__ void fe_load(Fe<NL> &r, const uint32_t *__restrict__ base, size_t stride, uint32_t idx)
{
    const uint32_t boff = idx * 4u;       // 32-bit byte offset: lets the SGPR-base addressing form match
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const char *row = (const char *)(base + (size_t)i * stride);
        r.v[i] = *(const uint32_t *)(row + boff);
    }
}

template <int NL>
__device__ __forceinline__ void fe_store(uint32_t *__restrict__ base, size_t stride, uint32_t idx, const Fe<NL> &r)
{
    const uint32_t boff = idx * 4u;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        char *row = (char *)(base + (size_t)i * stride);
        *(uint32_t *)(row + boff) = r.v[i];
    }
}
