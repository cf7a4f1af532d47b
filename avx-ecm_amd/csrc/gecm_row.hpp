// gecm_row.hpp — stage 1 with THIRTY-TWO lanes per curve: the X and the Z coordinate of a curve's points on two
// adjacent DPP rows (16 lanes each) of a wavefront, and inside a row each lane holds NQ consecutive limbs of the
// residue (lane l: limbs NQ*l .. NQ*l+NQ-1; NQ = 1 for the 416-bit class).
//
// Why: BASELINE configs[1] is a batch of 4096 curves.  MI355X has 1024 SIMDs and v_mad_u64_u32 only issues at
// its full rate from two wavefronts per SIMD (profiles/r01_valu_ubench_gfx950.txt: 8.2 cycles with one, 4.7
// with two), so the batch has to become 2048 wavefronts: 32 lanes per curve.  The eight-lane layout
// (gecm_quad.hpp) leaves half of the SIMDs without a wavefront and the other half at the slow rate.
//
// The multiply is row-wise (operand scanning) Montgomery, as in the eight-lane layout, re-thought so that a row
// costs 3 multiply-adds and 3 DPP moves per lane at NQ = 1 (5 + 3 at NQ = 2):
//   * arithmetic is modulo N' = m*N with N' = -1 (mod 2^28), so the Montgomery digit is the low limb itself
//     (rho = 1: no multiplication on the dependent path); N' is 28 bits longer than N, which is exactly the room
//     16 lanes x 28 bits leave above a 415-bit N plus the 5 bits of lazy-reduction headroom;
//   * every accumulator is kept multiplied by 16, so that its high register IS the part above 28 bits and its
//     low register IS the low limb (times 16): no shift or mask instructions; the operands are pre-multiplied by
//     4 each, the digit is used as it comes (times 16), and the hand-over "high part into the next slot" is one
//     more v_mad (x16);
//   * digit and operand limbs are broadcast inside the row by DPP row_newbcast, the window moves down by DPP
//     row_shl:1;
//   * limbs are SIGNED and balanced ([-2^27, 2^27] after a multiply): subtraction is limb-wise without a bias,
//     and the pre-multiplied operands stay inside 32 bits.
// The residues are the same elements of Z/N as in every other layout (N | N'), so results are identical; the
// kernel converts from and to the R = 2^(28*NL) Montgomery form of the device buffers at entry and exit
// (R' = 2^(448*NQ) inside), and k_canon makes the exit values canonical.
#pragma once
#include "gecm_curve.hpp"
#include "gecm_launch.h"

template <int NQ>
struct FeR {
    int32_t v[NQ];
};

template <int NQ>
struct RowMod {
    uint32_t n[NQ];   // this lane's limbs of the modulus
    uint32_t rho;     // -modulus^-1 mod 2^28 (unused when RHO1)
};

#define GECM_DPP_ROW_SHL1 0x101        /* lane l <- lane l+1 of the row (lane 15: 0) */
#define GECM_DPP_ROW_SHR1 0x111        /* lane l <- lane l-1 of the row (lane 0: 0) */
#define GECM_DPP_ROW_NEWBCAST 0x150    /* + i: every lane <- lane i of its row (gfx90a+) */

template <int CTRL>
__device__ __forceinline__ uint32_t row_dpp(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);
}
template <int I>
__device__ __forceinline__ uint32_t row_bcast(uint32_t x)
{
    return row_dpp<GECM_DPP_ROW_NEWBCAST + I>(x);
}
__device__ __forceinline__ uint32_t other_row(uint32_t x)       // lane <-> lane ^ 16: the other coordinate
{
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x401F /* bit mode: and 0x1f, or 0, xor 0x10 */);
}

__device__ __forceinline__ void smad(int64_t &acc, int32_t x, int32_t y)
{
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
}
__device__ __forceinline__ void umad(int64_t &acc, uint32_t x, uint32_t y)
{
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
}
__device__ __forceinline__ int64_t smad16(int32_t x, int64_t add)    // 16 * x + add
{
    int64_t r;
    asm("v_mad_i64_i32 %0, vcc, %1, 16, %2" : "=v"(r) : "v"(x), "v"(add) : "vcc");
    return r;
}

// r = a*b/R' mod (modulus of m), R' = 2^(448*NQ).  Operand limbs |.| < 2^29; result limbs in [-2^27-4, 2^27+4]
// (top limb: whatever the value needs), |result| < |a||b|/R' + modulus.
template <int NQ, bool RHO1>
__device__ __forceinline__ void fer_mul(FeR<NQ> &r, const FeR<NQ> &a, const FeR<NQ> &b, const RowMod<NQ> &m)
{
    int32_t a4[NQ], b4[NQ];
#pragma unroll
    for (int t = 0; t < NQ; t++) {
        a4[t] = (int32_t)((uint32_t)a.v[t] << 2);
        b4[t] = (int32_t)((uint32_t)b.v[t] << 2);
    }
    int64_t T[NQ];                    // 16 x the window; logical slot t of row i lives in T[(t + i) % NQ]
#pragma unroll
    for (int t = 0; t < NQ; t++) T[t] = 0;
    static_for<0, 16 * NQ>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int rot = i % NQ, nxt = (i + 1) % NQ;
        const int32_t A = (int32_t)row_bcast<i / NQ>((uint32_t)a4[i % NQ]);
#pragma unroll
        for (int t = 0; t < NQ; t++) smad(T[(t + rot) % NQ], A, b4[t]);
        uint32_t qs = (uint32_t)T[rot];                     // 16 x (column mod 2^28)
        if (!RHO1) qs *= m.rho;                             // 16 x the digit (mod 2^32)
        const uint32_t Q = row_bcast<0>(qs);
#pragma unroll
        for (int t = 0; t < NQ; t++) umad(T[(t + rot) % NQ], Q, m.n[t]);
        // the window moves down one limb: the low register of the lowest slot (zero on lane 0 by the choice of
        // the digit) becomes the top slot of the lane below, its high register (the part above 28 bits) is
        // added, times 16, to this lane's next slot
        const int32_t hi = (int32_t)(T[rot] >> 32);
        const uint32_t lo = row_dpp<GECM_DPP_ROW_SHL1>((uint32_t)T[rot]);
        T[rot] = (int64_t)(uint64_t)lo;
        T[nxt] = smad16(hi, T[nxt]);
    });
    // balanced normalisation.  Inside the lane the slots still hold whole column sums (only the lowest slot is
    // folded per row), so the carry runs through them in 64 bits; the top slot is a fresh 28-bit hand-over, so
    // what leaves the lane is small and goes to the lane above carry-save: limb = centred low 28 bits + carry.
    int32_t lo[NQ];
    int64_t carry = 0;
#pragma unroll
    for (int t = 0; t < NQ - 1; t++) {
        const int64_t u = (T[t] >> 4) + carry + (1 << 27);
        carry = u >> GECM_LIMB_BITS;
        lo[t] = (int32_t)((uint32_t)u & GECM_LIMB_MASK) - (1 << 27);
    }
    const int32_t ut = (int32_t)(T[NQ - 1] >> 4) + (int32_t)carry + (1 << 27);
    lo[NQ - 1] = (int32_t)((uint32_t)ut & GECM_LIMB_MASK) - (1 << 27);
    const int32_t below = (int32_t)row_dpp<GECM_DPP_ROW_SHR1>((uint32_t)(ut >> GECM_LIMB_BITS));
    r.v[0] = lo[0] + below;
#pragma unroll
    for (int t = 1; t < NQ; t++) r.v[t] = lo[t];
}

template <int NQ>
__device__ __forceinline__ void fer_other(FeR<NQ> &r, const FeR<NQ> &a)
{
#pragma unroll
    for (int t = 0; t < NQ; t++) r.v[t] = (int32_t)other_row((uint32_t)a.v[t]);
}

// r = x + y on lanes with neg == false, x - y on lanes with neg == true
template <int NQ>
__device__ __forceinline__ void fer_addsub_lane(FeR<NQ> &r, const FeR<NQ> &x, const FeR<NQ> &y, bool neg)
{
#pragma unroll
    for (int t = 0; t < NQ; t++) r.v[t] = x.v[t] + (neg ? -y.v[t] : y.v[t]);
}

// the point arithmetic of gecm_quad.hpp (ecm.c:407-457 split into its X and Z halves), on rows
template <int NQ>
__device__ __forceinline__ void row_sum_diff(FeR<NQ> &r, const FeR<NQ> &own, bool isZ)
{
    FeR<NQ> oth;
    fer_other<NQ>(oth, own);
    fer_addsub_lane<NQ>(r, oth, own, isZ);            // X rows: Z + X      Z rows: X - Z
}

template <int NQ>
__device__ __forceinline__ void row_diff_sum(FeR<NQ> &r, const FeR<NQ> &own, bool isZ)
{
    FeR<NQ> oth;
    fer_other<NQ>(oth, own);
    fer_addsub_lane<NQ>(r, own, oth, !isZ);           // X rows: X - Z      Z rows: Z + X
}

template <int NQ>
__device__ __forceinline__ void row_add(FeR<NQ> &T, const FeR<NQ> &fB, const FeR<NQ> &fA, const FeR<NQ> &c, bool isZ,
                                        const RowMod<NQ> &m)
{
    FeR<NQ> w, t, e;
    fer_mul<NQ, true>(w, fB, fA, m);                  // X: U      Z: V
    fer_other<NQ>(t, w);
    fer_addsub_lane<NQ>(e, t, w, isZ);                // X: V + U  Z: U - V
    fer_mul<NQ, true>(e, e, e, m);
    fer_other<NQ>(t, c);                              // X: C.Z    Z: C.X
    fer_mul<NQ, true>(T, e, t, m);
}

template <int NQ>
__device__ __forceinline__ void row_dup(FeR<NQ> &D, const FeR<NQ> &fA, const FeR<NQ> &s4, bool isZ, const RowMod<NQ> &m)
{
    FeR<NQ> q, t, w, p1, p2, r1;
    fer_mul<NQ, true>(q, fA, fA, m);                  // X: U = (x+z)^2    Z: V = (x-z)^2
    fer_other<NQ>(t, q);                              // X: V              Z: U
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        w.v[i] = t.v[i] - q.v[i];                     // Z: w = U - V
        p1.v[i] = isZ ? s4.v[i] : q.v[i];
        p2.v[i] = isZ ? w.v[i] : t.v[i];
    }
    fer_mul<NQ, true>(r1, p1, p2, m);                 // X: U*V            Z: s*w
#pragma unroll
    for (int i = 0; i < NQ; i++) t.v[i] = r1.v[i] + q.v[i];     // Z: s*w + V
    fer_mul<NQ, true>(t, t, w, m);                    // Z: (s*w + V)*w
#pragma unroll
    for (int i = 0; i < NQ; i++) D.v[i] = isZ ? t.v[i] : r1.v[i];
}

template <int NQ>
__device__ __forceinline__ void fer_load(FeR<NQ> &r, const uint32_t *__restrict__ base, size_t stride, uint32_t cidx,
                                         uint32_t l, uint32_t nl)
{
#pragma unroll
    for (int t = 0; t < NQ; t++) {
        const uint32_t limb = (uint32_t)NQ * l + (uint32_t)t;
        r.v[t] = limb < nl ? (int32_t)base[(size_t)limb * stride + cidx] : 0;
    }
}

// run_tape_quad of gecm_quad.hpp on rows: A, B, C are this lane's limbs of its coordinate.
template <int NQ>
__device__ __forceinline__ void run_tape_row(const uint32_t *__restrict__ tape, uint32_t tape_len, FeR<NQ> &A,
                                             const FeR<NQ> &s4, bool isZ, const RowMod<NQ> &m)
{
    FeR<NQ> B = A, C = A;
    auto fetch = [&](uint32_t pc) -> uint32_t {
        uint32_t w = tape[pc >> 2];
        return __builtin_amdgcn_readfirstlane((w >> ((pc & 3u) * 8u)) & 0xffu);
    };
    uint32_t nxt = tape_len ? fetch(0) : GECM_OP_NOP;
    for (uint32_t pc = 0; pc < tape_len; pc++) {
        uint32_t op = nxt;
        nxt = (pc + 1 < tape_len) ? fetch(pc + 1) : GECM_OP_NOP;
        while ((op & ~GECM_OP_SWAP) == (GECM_OP_STEP | GECM_OP_RULE3)) {
            if (op & GECM_OP_SWAP) {
                FeR<NQ> t = A;
                A = B;
                B = t;
            }
            FeR<NQ> fA, fB, T;
            row_diff_sum<NQ>(fB, B, isZ);
            row_sum_diff<NQ>(fA, A, isZ);
            row_add<NQ>(T, fB, fA, C, isZ, m);
            C = B;
            B = T;
            pc++;
            op = nxt;
            nxt = (pc + 1 < tape_len) ? fetch(pc + 1) : GECM_OP_NOP;
        }
        if (op == GECM_OP_NOP) continue;
        const uint32_t rule = op & GECM_OP_RULE_MASK;
        const bool is_step = op >= GECM_OP_STEP;
        const bool do_add = op != GECM_OP_PRAC_BEGIN;
        const bool do_dup = op != GECM_OP_PRAC_END;
        if (is_step && (op & GECM_OP_SWAP)) {
            FeR<NQ> t = A;
            A = B;
            B = t;
        }
        if (is_step && rule == GECM_OP_RULE5) {
            FeR<NQ> t = B;
            B = C;
            C = t;
        } else if (is_step && rule == GECM_OP_RULE9) {
            FeR<NQ> t = A;
            A = B;
            B = C;
            C = t;
        } else if (op == GECM_OP_PRAC_BEGIN) {
            B = A;
            C = A;
        }
        FeR<NQ> T, D;
        {
            FeR<NQ> fA;
            row_sum_diff<NQ>(fA, A, isZ);
            if (do_add) {
                FeR<NQ> fB;
                row_diff_sum<NQ>(fB, B, isZ);
                row_add<NQ>(T, fB, fA, C, isZ, m);
            }
            if (do_dup) row_dup<NQ>(D, fA, s4, isZ, m);
        }
        if (op == GECM_OP_PRAC_END) {
            A = T;
        } else if (op == GECM_OP_PRAC_BEGIN) {
            A = D;
        } else if (rule == GECM_OP_RULE4) {
            B = T;
            A = D;
        } else if (rule == GECM_OP_RULE5) {
            FeR<NQ> t = C;
            C = T;
            B = t;
            A = D;
        } else {
            FeR<NQ> oldA = C;
            C = T;
            B = D;
            A = oldA;
        }
    }
}

// Constants of the row kernel, one array of GECM_ROW_WORDS words per kind, limb j at word j (zero padded):
//   [0] N' = m*N, = -1 mod 2^28      [1] N      [2] c_in = 2^28 * R' mod N (entry conversion)
//   [3] R mod N (exit conversion)    [4] K' of N (bias that makes the exit value's limbs non-negative)
// (GECM_ROW_WORDS, GECM_ROW_KINDS: gecm_launch.h)

// The whole stage-1 kernel body for one lane.  nl = limbs per residue in the device buffers (R = 2^(28*nl)).
template <int NQ>
__device__ __forceinline__ void stage1_row(const uint32_t *__restrict__ tape, uint32_t tape_len, uint32_t *__restrict__ X,
                                           uint32_t *__restrict__ Z, const uint32_t *__restrict__ S, size_t stride,
                                           uint32_t nl, const uint32_t *__restrict__ rc, uint32_t rho_n)
{
    const uint32_t cidx = blockIdx.x * 2u + (threadIdx.x >> 5);
    const uint32_t l = threadIdx.x & 15u;
    const bool isZ = (threadIdx.x & 16u) != 0;
    uint32_t *mine = isZ ? Z : X;
    RowMod<NQ> mp, mn;
    FeR<NQ> cin, one, kp;
#pragma unroll
    for (int t = 0; t < NQ; t++) {
        const uint32_t j = (uint32_t)NQ * l + (uint32_t)t;
        mp.n[t] = rc[0 * GECM_ROW_WORDS + j];
        mn.n[t] = rc[1 * GECM_ROW_WORDS + j];
        cin.v[t] = (int32_t)rc[2 * GECM_ROW_WORDS + j];
        one.v[t] = (int32_t)rc[3 * GECM_ROW_WORDS + j];
        kp.v[t] = (int32_t)rc[4 * GECM_ROW_WORDS + j];
    }
    mp.rho = 1u;
    mn.rho = rho_n;
    FeR<NQ> P, s4, t;
    fer_load<NQ>(t, mine, stride, cidx, l, nl);
    fer_mul<NQ, true>(P, t, cin, mp);                       // x*R -> x*R' (mod N')
    fer_load<NQ>(t, S, stride, cidx, l, nl);
    fer_mul<NQ, true>(s4, t, cin, mp);
    run_tape_row<NQ>(tape, tape_len, P, s4, isZ, mp);
    fer_mul<NQ, false>(t, P, one, mn);                      // x*R' -> x*R (mod N), in (-N/16, 17N/16)
    // + K' (a multiple of N with every limb >= 2^28 - 1): all limbs positive; then one carry-save pass
    uint32_t u[NQ];
#pragma unroll
    for (int i = 0; i < NQ; i++) u[i] = (uint32_t)(t.v[i] + kp.v[i]);
    const uint32_t below = row_dpp<GECM_DPP_ROW_SHR1>(u[NQ - 1] >> GECM_LIMB_BITS);
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        const uint32_t limb = (uint32_t)NQ * l + (uint32_t)i;
        const uint32_t v = (u[i] & GECM_LIMB_MASK) + (i == 0 ? below : (u[i - 1] >> GECM_LIMB_BITS));
        if (limb < nl) mine[(size_t)limb * stride + cidx] = (limb == nl - 1) ? v + (u[i] & ~GECM_LIMB_MASK) : v;
    }
}
