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
// costs 3 multiply-adds and 2.5 DPP moves per lane at NQ = 1 (5 + 2.5 at NQ = 2, 7 + 2.67 at NQ = 3):
//   * arithmetic is modulo N' = m*N with N' = -1 (mod 2^28), so the Montgomery digit is the low limb itself
//     (rho = 1: no multiplication on the dependent path); N' is 28 bits longer than N, which is exactly the room
//     16 lanes x 28 bits leave above a 415-bit N plus the 5 bits of lazy-reduction headroom;
//   * every accumulator is kept multiplied by 16, so that its high register IS the part above 28 bits and its
//     low register IS the low limb (times 16): no shift or mask instructions; the operands are pre-multiplied by
//     4 each, the digit is used as it comes (times 16), and the hand-over "high part into the next slot" is one
//     more v_mad (x16);
//   * digit and operand limbs are broadcast inside the row by DPP row_newbcast — the operand limbs TWO per move
//     (v_mov_b64_dpp, which knows exactly this control) —, the window moves down by DPP row_shl:1;
//   * a multiply ends in the accumulators' own scale: T + 2^31 holds the balanced limb (+ 2^27) in bits 4..31 of its
//     low register and the carry for the lane above AS its high register;
//   * limbs are SIGNED and balanced ([-2^27, 2^27] after a multiply): subtraction is limb-wise without a bias,
//     and the pre-multiplied operands stay inside 32 bits.
// The residues are the same elements of Z/N as in every other layout (N | N'), so results are identical; the
// kernel converts from and to the R = 2^(28*NL) Montgomery form of the device buffers at entry and exit
// (R' = 2^(448*NQ) inside), and k_canon makes the exit values canonical.
#pragma once
#include "gecm_curve.hpp"
#include "gecm_rowk.h"

#ifndef GECM_ROW_DPP_ROWS
#define GECM_ROW_DPP_ROWS 48    /* rows of a multiply whose operand limb is broadcast by DPP (the rest: ds_swizzle).
                                   Measured at 4096 curves x 415 bits, B1 = 1e5: all rows 271 ms, 4 rows 279 ms */
#endif

template <int NQ>
struct FeR {
    int32_t v[NQ];
};

template <int NQ>
struct RowMod {
    uint32_t n[NQ];   // this lane's limbs of the modulus
    uint32_t rho;     // -modulus^-1 mod 2^28 (unused when RHO1)
    int32_t c16;      // 16, in a scalar register, opaque to the compiler (row_c16)
};

// An empty asm statement that READS x: a second use of the value stops the compiler from re-associating the sum it
// belongs to (and moving its instruction) and it stays where it is written; no instruction, no register written.
__device__ __forceinline__ void row_pin(const int64_t &x)
{
    asm volatile("" ::"v"(x));
}
__device__ __forceinline__ int32_t row_c16()
{
    int32_t c;
    asm volatile("s_mov_b32 %0, 16" : "=s"(c));
    return c;
}

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
template <int I>
__device__ __forceinline__ int64_t row_bcast64(int64_t x)       // two registers in one move (v_mov_b64_dpp)
{
    return __builtin_amdgcn_update_dpp((int64_t)0, x, GECM_DPP_ROW_NEWBCAST + I, 0xF, 0xF, true);
}
// every lane <- lane I of its row, through the LDS crossbar (ds_swizzle, bit mode: lane = (lane & 0x10) | I):
// no VALU issue slot, the latency is covered by issuing all broadcasts of a multiply before its first row
template <int I>
__device__ __forceinline__ uint32_t row_bcast_lds(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x10 | (I << 5));
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
__device__ __forceinline__ int64_t smad0(int32_t x, int32_t y)       // x * y
{
    int64_t r;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(x), "v"(y) : "vcc");
    return r;
}
__device__ __forceinline__ int64_t smad16(int32_t x, int64_t add)    // 16 * x + add
{
    int64_t r;
    asm("v_mad_i64_i32 %0, vcc, %1, 16, %2" : "=v"(r) : "v"(x), "v"(add) : "vcc");
    return r;
}

// 16 * x + add, then + y * z: the hand-over at the end of a row and the first multiply-add of the next row in one
// asm statement (the compiler pads a wait state between two asm statements)
__device__ __forceinline__ int64_t smad16_smad(int32_t x, int64_t add, int32_t y, int32_t z)
{
    int64_t r;
    asm("v_mad_i64_i32 %0, vcc, %1, 16, %2\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0" : "=&v"(r) : "v"(x), "v"(add), "v"(y), "v"(z) : "vcc");
    return r;
}

// r = a*b/R' mod (modulus of m), R' = 2^(28*ROWS): ROWS = the limbs in use (a multiple of NQ, at most 16*NQ; limbs
// from ROWS up are zero in every operand and in the modulus, so their rows would add nothing).  Operand limbs |.| < 2^29; result limbs in [-2^27-4, 2^27+4]
// (top limb: whatever the value needs), |result| < |a||b|/R' + modulus.
// ALDS: the limbs of a are broadcast through the LDS crossbar (ds_swizzle: no VALU issue slot; with ONE limb per lane best
// from 3 wavefronts per SIMD up) instead of DPP row_newbcast (best at 2, and at every batch size with 2-3 limbs per lane).
template <int NQ, int ROWS, bool RHO1, bool ALDS>
__device__ __forceinline__ void fer_mul(FeR<NQ> &r, const FeR<NQ> &a, const FeR<NQ> &b, const RowMod<NQ> &m)
{
    int32_t a4[NQ], b4[NQ];
#pragma unroll
    for (int t = 0; t < NQ; t++) {
        a4[t] = (int32_t)((uint32_t)a.v[t] << 2);
        b4[t] = (int32_t)((uint32_t)b.v[t] << 2);
    }
    // The limbs of 4a reach the lanes of the row one per row of the multiply.  A DPP read of a register that a
    // VALU instruction has just written needs two independent instructions in between; a row has two such places
    // (multiply-add -> digit broadcast, multiply-add -> hand-over), i.e. four instruction slots, and the requests
    // for later rows' limbs are what fills them:
    //   * rows 0 .. ND-1 get their limb by DPP row_newbcast, requested one row ahead (a VALU instruction, ready at
    //     once: the operand a is only known when the multiply starts);
    //   * rows ND .. get theirs by ds_swizzle through the LDS crossbar (no VALU issue slot, but ~70 cycles), all
    //     requested in the slots of the first rows.
    // ALDS (3 or more wavefronts per SIMD hide the start-up latency): every limb by ds_swizzle, requested up front.
    static_assert(ROWS > 16 * (NQ - 1) && ROWS <= 16 * NQ, "the limbs in use fill the lanes but for the last one");
    constexpr int ND = ALDS ? 0 : (GECM_ROW_DPP_ROWS < ROWS ? GECM_ROW_DPP_ROWS : ROWS);
#ifndef GECM_ROW_OLD_NQ2
    // (ROWS need not be a multiple of NQ here — 31 rows for 831 bits, 38 for 1023 —; the crossbar variant below wants whole
    // lanes, so a shape with a partly used last lane runs these rows whatever ALDS says)
    constexpr bool CROWS = (NQ == 2 || NQ == 3) && ((!ALDS && ND == ROWS) || ROWS % NQ != 0);
    if constexpr (CROWS) {
        // Two or three limbs per lane: the multiply-adds are written in C and the compiler places them (it knows how
        // many instructions lie between a result and the DPP move that reads it; an asm statement counts as none and
        // is padded).  Two things keep a row at 3 + 2 NQ multiply-adds and 3 DPP moves: the hand-over "16 x high
        // part" multiplies by m.c16, a 16 the compiler cannot see through (it would otherwise spend a shift, a mask
        // and a 64-bit add on it), and the slot the window shift has emptied starts from {lo, 0} as the addend of its
        // first product.  Per row at NQ = 2: 8 VALU instructions and no s_nop where the round-2 form had 9 and two.
        int64_t T[NQ];
        // 4 x limb j of a in every lane of the row: limbs 0 and 1 of a lane travel as ONE 64-bit DPP move
        // (v_mov_b64_dpp knows row_newbcast only, which is the one needed), limb 2 (NQ = 3) on its own
        const int64_t a01 = (int64_t)(((uint64_t)(uint32_t)a4[1] << 32) | (uint32_t)a4[0]);
        int64_t Ap = row_bcast64<0>(a01);
        int32_t As = 0;
        auto limb = [&](auto jc) -> int32_t {     // limb j, which the last request that covers it has brought
            constexpr int q = decltype(jc)::value % NQ;
            return q == 0 ? (int32_t)(uint32_t)Ap : q == 1 ? (int32_t)(Ap >> 32) : As;
        };
#pragma unroll
        for (int t = 0; t < NQ; t++) T[t] = (int64_t)limb(IC<0>{}) * b4[t];
        // The order inside a row is pinned (sched_barrier, row_pin): the digit broadcast and the window shift each read
        // a result two instructions old, so no wait state is left to pad:
        //   digit | q*n (lowest slot first) | a*b into the next lowest slot (and the middle one) | shift | hand-over
        //   | a*b into the emptied slot | request for what the next row's products need
        static_for<0, ROWS>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int rot = i % NQ, nxt = (i + 1) % NQ;
            const uint32_t Q = row_bcast<0>(RHO1 ? (uint32_t)T[rot] : (uint32_t)T[rot] * m.rho);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NQ; t++) {
                T[(t + rot) % NQ] = (int64_t)((uint64_t)T[(t + rot) % NQ] + (uint64_t)Q * m.n[t]);
                __builtin_amdgcn_sched_barrier(0);
            }
            const int32_t Ac = limb(IC<i + 1>{});                  // row i adds limb i + 1 of a
            if constexpr (i + 1 < ROWS) {
#pragma unroll
                for (int t = 0; t < NQ - 1; t++) {
                    T[(t + nxt) % NQ] += (int64_t)Ac * b4[t];
                    row_pin(T[(t + nxt) % NQ]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const int32_t hi = (int32_t)(T[rot] >> 32);
            const uint32_t lo = row_dpp<GECM_DPP_ROW_SHL1>((uint32_t)T[rot]);
            __builtin_amdgcn_sched_barrier(0);
            T[nxt] += (int64_t)hi * m.c16;
            row_pin(T[nxt]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i + 1 < ROWS) {
                T[rot] = (int64_t)Ac * b4[NQ - 1] + (int64_t)(uint64_t)lo;
                row_pin(T[rot]);
            } else {
                T[rot] = (int64_t)(uint64_t)lo;
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i + 2 < ROWS) {
                if constexpr ((i + 2) % NQ == 0) Ap = row_bcast64<(i + 2) / NQ>(a01);
                else if constexpr ((i + 2) % NQ == 2) As = (int32_t)row_bcast<(i + 2) / NQ>((uint32_t)a4[NQ - 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        // balanced normalisation as below, in the accumulators' own scale: u16 = 16 x (value + carry + 2^27) has the limb
        // (+ 2^27) in bits 4..31 of its low register and the carry AS its high register
        int32_t lo[NQ];
        int32_t carry = 0;
#pragma unroll
        for (int t = 0; t < NQ - 1; t++) {
            int64_t u16 = T[(t + ROWS) % NQ] + (int64_t)(1u << 31);
            if (t > 0) u16 += (int64_t)carry * m.c16;
            carry = (int32_t)(u16 >> 32);
            lo[t] = (int32_t)(((uint32_t)u16 >> 4) & GECM_LIMB_MASK) - (1 << 27);
        }
        const int32_t ut = (int32_t)(T[(NQ - 1 + ROWS) % NQ] >> 4) + carry + (1 << 27);
        lo[NQ - 1] = (int32_t)((uint32_t)ut & GECM_LIMB_MASK) - (1 << 27);
        const int32_t below = (int32_t)row_dpp<GECM_DPP_ROW_SHR1>((uint32_t)(ut >> GECM_LIMB_BITS));
        r.v[0] = lo[0] + below;
#pragma unroll
        for (int t = 1; t < NQ; t++) r.v[t] = lo[t];
        return;
    }
#endif
    int32_t Ab[ROWS];
#ifndef GECM_ROW_OLD_NQ2
    static_assert(CROWS || ROWS % NQ == 0, "these rows rotate the slots back to where they started: whole lanes");
#endif
#ifndef GECM_ROW_NO_PAIR_BCAST
    // One limb per lane, all rows by DPP: the limbs travel TWO per move.  Every lane first takes the limb of the lane
    // above next to its own (one row_shl:1 per multiply); a 64-bit move (v_mov_b64_dpp, row_newbcast — the one control
    // it knows) from lane 2k then brings limbs 2k and 2k + 1: 8 moves + 1 instead of 16 for the 416-bit class.
    constexpr bool PAIRS = NQ == 1 && !ALDS && ND == ROWS;
    int64_t a01 = 0;
    if constexpr (PAIRS)
        a01 = (int64_t)(((uint64_t)row_dpp<GECM_DPP_ROW_SHL1>((uint32_t)a4[0]) << 32) | (uint32_t)a4[0]);
#else
    constexpr bool PAIRS = false;
    const int64_t a01 = 0;
#endif
    auto request = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (PAIRS) {
            if constexpr (j < ROWS && j % 2 == 0) {
                const int64_t P = row_bcast64<j>(a01);
                Ab[j] = (int32_t)(uint32_t)P;
                if constexpr (j + 1 < ROWS) Ab[j + 1] = (int32_t)(P >> 32);
            }
        } else if constexpr (j < ND) Ab[j] = (int32_t)row_bcast<j / NQ>((uint32_t)a4[j % NQ]);
        else if constexpr (j < ROWS) Ab[j] = (int32_t)row_bcast_lds<j / NQ>((uint32_t)a4[j % NQ]);
    };
    // slot plan: row i < ND: first place = DPP request for row i+1 (if that is a DPP row) else one LDS request;
    // second place = two LDS requests.  Rows >= ND: one LDS request in the first place, two in the second, until
    // all are out.  LDS requests go out in row order (the crossbar answers in order).
    if constexpr (ALDS) static_for<0, ROWS>(request);
    else request(IC<0>{});
    int64_t T[NQ];                    // 16 x the window; logical slot t of row i lives in T[(t + i) % NQ]
#pragma unroll
    for (int t = 0; t < NQ; t++) T[t] = 0;
    static_for<0, ROWS>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int rot = i % NQ, nxt = (i + 1) % NQ;
        // LDS requests issued before this row: 2 per row in the second place, plus 1 in the first place of every
        // row from ND-1 on (whose next row is not a DPP row)
        constexpr int first_lds = (i + 1 < ND) ? 0 : 1;
        constexpr int before = ND + 2 * i + (i >= ND ? i - ND + 1 : 0);
        // (the multiply-add into the lowest slot was fused with the previous row's hand-over, except in row 0)
        if constexpr (i == 0) {
            if constexpr (NQ == 1) T[0] = smad0(Ab[0], b4[0]);
            else smad(T[rot], Ab[0], b4[0]);
        }
#pragma unroll
        for (int t = 1; t < NQ; t++) smad(T[(t + rot) % NQ], Ab[i], b4[t]);
        if constexpr (!ALDS) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i + 1 < ND) request(IC<i + 1>{});
            else request(IC<before>{});
            __builtin_amdgcn_sched_barrier(0);
        }
        uint32_t qs = (uint32_t)T[rot];                     // 16 x (column mod 2^28)
        if (!RHO1) qs *= m.rho;                             // 16 x the digit (mod 2^32)
        const uint32_t Q = row_bcast<0>(qs);
#pragma unroll
        for (int t = 0; t < NQ; t++) umad(T[(t + rot) % NQ], Q, m.n[t]);
        if constexpr (!ALDS) {
            __builtin_amdgcn_sched_barrier(0);
            request(IC<before + first_lds>{});
            request(IC<before + first_lds + 1>{});
            __builtin_amdgcn_sched_barrier(0);
        }
        // the window moves down one limb: the low register of the lowest slot (zero on lane 0 by the choice of
        // the digit) becomes the top slot of the lane below, its high register (the part above 28 bits) is
        // added, times 16, to this lane's next slot — which is the next row's lowest slot and gets that row's
        // first product in the same statement
        const int32_t hi = (int32_t)(T[rot] >> 32);
        const uint32_t lo = row_dpp<GECM_DPP_ROW_SHL1>((uint32_t)T[rot]);
        T[rot] = (int64_t)(uint64_t)lo;
        if constexpr (i + 1 < ROWS) T[nxt] = smad16_smad(hi, T[nxt], Ab[i + 1], b4[0]);
        else T[nxt] = smad16(hi, T[nxt]);
    });
    // balanced normalisation.  Inside the lane the slots still hold whole column sums (only the lowest slot is
    // folded per row), so the carry runs through them in 64 bits; the top slot is a fresh 28-bit hand-over, so
    // what leaves the lane is small and goes to the lane above carry-save: limb = centred low 28 bits + carry.
    if constexpr (NQ == 1) {
        // in the accumulator's own scale: u16 = 16 x (value + 2^27) has the limb (+ 2^27) in bits 4..31 of its low register
        // and what leaves the lane AS its high register; the neighbour's carry comes in on the add itself (DPP)
        const int64_t u16 = T[0] + (int64_t)(1u << 31);
        const int32_t lim = (int32_t)(((uint32_t)u16 >> 4) & GECM_LIMB_MASK) - (1 << 27);
        r.v[0] = lim + (int32_t)row_dpp<GECM_DPP_ROW_SHR1>((uint32_t)(u16 >> 32));
        return;
    }
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

// The same multiply with the limbs of 4a ALREADY in every lane of the row (Ab[j] = 4 x limb j of a): no operand
// broadcast in the rows at all.  This is what the multiplies of the LDS-prefetch kernel run (run_tape_row_lds below):
// their broadcast operand is a point form that was written to LDS when the point was made and is read back, four
// limbs per ds_read_b128 with every lane of a row reading the same address, one multiply ahead of its use.
template <int NQ, int ROWS, bool RHO1>
__device__ __forceinline__ void fer_mul_pre(FeR<NQ> &r, const int32_t (&Ab)[ROWS], const FeR<NQ> &b, const RowMod<NQ> &m)
{
    static_assert(ROWS % NQ == 0 && ROWS <= 16 * NQ, "whole lanes, at most 16 of them");
    int32_t b4[NQ];
#pragma unroll
    for (int t = 0; t < NQ; t++) b4[t] = (int32_t)((uint32_t)b.v[t] << 2);
    int64_t T[NQ];
#pragma unroll
    for (int t = 0; t < NQ; t++) T[t] = 0;
    static_for<0, ROWS>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int rot = i % NQ, nxt = (i + 1) % NQ;
        if constexpr (i == 0) smad(T[rot], Ab[0], b4[0]);
#pragma unroll
        for (int t = 1; t < NQ; t++) smad(T[(t + rot) % NQ], Ab[i], b4[t]);
        uint32_t qs = (uint32_t)T[rot];
        if (!RHO1) qs *= m.rho;
        const uint32_t Q = row_bcast<0>(qs);
#pragma unroll
        for (int t = 0; t < NQ; t++) umad(T[(t + rot) % NQ], Q, m.n[t]);
        const int32_t hi = (int32_t)(T[rot] >> 32);
        const uint32_t lo = row_dpp<GECM_DPP_ROW_SHL1>((uint32_t)T[rot]);
        T[rot] = (int64_t)(uint64_t)lo;
        if constexpr (i + 1 < ROWS) T[nxt] = smad16_smad(hi, T[nxt], Ab[i + 1], b4[0]);
        else T[nxt] = smad16(hi, T[nxt]);
    });
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

// A point coordinate as the tape interpreter keeps it: its own limbs and the three combinations with the other
// coordinate that the point formulas read (ecm.c:407-457 split into X and Z halves as in gecm_quad.hpp), all made
// ONCE when the point is created, from one exchange between the X row and the Z row.  The exchange goes through
// the LDS crossbar (ds_swizzle, lane ^ 16): no VALU issue slot, which is what a batch at 2 wavefronts per SIMD is
// short of (v_permlane16_swap, measured: 7% slower at 4096 curves).
template <int NQ>
struct PtR {
    FeR<NQ> own;   // X rows: X        Z rows: Z
    FeR<NQ> sd;    // X rows: Z + X    Z rows: X - Z      (the operand this point gives as "A")
    FeR<NQ> ds;    // X rows: X - Z    Z rows: Z + X      (as "B")
    FeR<NQ> oth;   // X rows: Z        Z rows: X          (as the difference point "C")
};

struct RowSign {
    uint32_t mz, bz;   // Z rows: -1, 1    X rows: 0, 0      (x ^ mz) + bz = -x on Z rows
    uint32_t mx, bx;   // X rows: -1, 1    Z rows: 0, 0
};

template <int NQ>
__device__ __forceinline__ void fer_other(FeR<NQ> &r, const FeR<NQ> &a)
{
#pragma unroll
    for (int t = 0; t < NQ; t++) r.v[t] = (int32_t)other_row((uint32_t)a.v[t]);
}

// r = oth + own on X rows, oth - own on Z rows
template <int NQ>
__device__ __forceinline__ void fer_sum_diff(FeR<NQ> &r, const FeR<NQ> &oth, const FeR<NQ> &own, const RowSign &g)
{
#pragma unroll
    for (int t = 0; t < NQ; t++) r.v[t] = (int32_t)((uint32_t)oth.v[t] + ((uint32_t)own.v[t] ^ g.mz) + g.bz);
}

template <int NQ>
__device__ __forceinline__ void row_forms(PtR<NQ> &p, const RowSign &g)
{
    fer_other<NQ>(p.oth, p.own);
    fer_sum_diff<NQ>(p.sd, p.oth, p.own, g);
#pragma unroll
    for (int t = 0; t < NQ; t++)                  // own - oth on X rows, own + oth on Z rows
        p.ds.v[t] = (int32_t)((uint32_t)p.own.v[t] + ((uint32_t)p.oth.v[t] ^ g.mx) + g.bx);
}

// T = A + B with difference C (vec_add, ecm.c:407-443): fB = B.ds, fA = A.sd, c = C.oth
template <int NQ, int ROWS, bool ALDS>
__device__ __forceinline__ void row_add(PtR<NQ> &T, const FeR<NQ> &fB, const FeR<NQ> &fA, const FeR<NQ> &c, bool isZ,
                                        const RowSign &g, const RowMod<NQ> &m)
{
    FeR<NQ> w, t, e;
    fer_mul<NQ, ROWS, true, ALDS>(w, fB, fA, m);            // X: U      Z: V
    fer_other<NQ>(t, w);
    fer_sum_diff<NQ>(e, t, w, g);                     // X: V + U  Z: U - V
    fer_mul<NQ, ROWS, true, ALDS>(e, e, e, m);
    fer_mul<NQ, ROWS, true, ALDS>(T.own, e, c, m);
    row_forms<NQ>(T, g);
}

// D = 2A (vec_duplicate, ecm.c:445-457): fA = A.sd, s4 = (A+2)/4 of the curve
template <int NQ, int ROWS, bool ALDS>
__device__ __forceinline__ void row_dup(PtR<NQ> &D, const FeR<NQ> &fA, const FeR<NQ> &s4, bool isZ, const RowSign &g,
                                        const RowMod<NQ> &m)
{
    FeR<NQ> q, t, w, p1, p2, r1;
    fer_mul<NQ, ROWS, true, ALDS>(q, fA, fA, m);            // X: U = (x+z)^2    Z: V = (x-z)^2
    fer_other<NQ>(t, q);                              // X: V              Z: U
#pragma unroll
    for (int i = 0; i < NQ; i++) {
        w.v[i] = t.v[i] - q.v[i];                     // Z: w = U - V
        p1.v[i] = isZ ? s4.v[i] : q.v[i];
        p2.v[i] = isZ ? w.v[i] : t.v[i];
    }
    fer_mul<NQ, ROWS, true, ALDS>(r1, p1, p2, m);           // X: U*V            Z: s*w
#pragma unroll
    for (int i = 0; i < NQ; i++) t.v[i] = r1.v[i] + q.v[i];     // Z: s*w + V
    fer_mul<NQ, ROWS, true, ALDS>(t, t, w, m);              // Z: (s*w + V)*w
#pragma unroll
    for (int i = 0; i < NQ; i++) D.own.v[i] = isZ ? t.v[i] : r1.v[i];
    row_forms<NQ>(D, g);
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

// run_tape_quad of gecm_quad.hpp on rows: A, B, C are this lane's limbs of its coordinate (and their forms).
template <int NQ, int ROWS, bool ALDS>
__device__ __forceinline__ void run_tape_row(const uint32_t *__restrict__ tape, uint32_t tape_len, PtR<NQ> &A,
                                             const FeR<NQ> &s4, bool isZ, const RowSign &g, const RowMod<NQ> &m)
{
    PtR<NQ> B = A, C = A;
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
                PtR<NQ> t = A;
                A = B;
                B = t;
            }
            PtR<NQ> T;
            row_add<NQ, ROWS, ALDS>(T, B.ds, A.sd, C.oth, isZ, g, m);
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
            PtR<NQ> t = A;
            A = B;
            B = t;
        }
        if (is_step && rule == GECM_OP_RULE5) {
            PtR<NQ> t = B;
            B = C;
            C = t;
        } else if (is_step && rule == GECM_OP_RULE9) {
            PtR<NQ> t = A;
            A = B;
            B = C;
            C = t;
        } else if (op == GECM_OP_PRAC_BEGIN) {
            B = A;
            C = A;
        }
        PtR<NQ> T, D;
        if (do_add) row_add<NQ, ROWS, ALDS>(T, B.ds, A.sd, C.oth, isZ, g, m);
        if (do_dup) row_dup<NQ, ROWS, ALDS>(D, A.sd, s4, isZ, g, m);
        if (op == GECM_OP_PRAC_END) {
            A = T;
        } else if (op == GECM_OP_PRAC_BEGIN) {
            A = D;
        } else if (rule == GECM_OP_RULE4) {
            B = T;
            A = D;
        } else if (rule == GECM_OP_RULE5) {
            PtR<NQ> t = C;
            C = T;
            B = t;
            A = D;
        } else {
            PtR<NQ> oldA = C;
            C = T;
            B = D;
            A = oldA;
        }
    }
}

// ==== experiment kept for the record (round 3; off unless GECM_ROW_LDS_VARIANT is defined: tools/ab build) ============
// Measured slower than the DPP kernel above by 3-4 % at 415, 623 and 831 bits and equal at 1023 (4096 curves,
// profiles/r03/rowp_lds_prefetch_ab_4096_curves.txt, profiles/r03/ab_tape_and_lds_variants.txt): DESIGN.md §5c.
#ifndef GECM_ROW_WG_WAVES
#define GECM_ROW_WG_WAVES 4
#endif
#ifdef GECM_ROW_LDS_VARIANT
// The op tape, read two events ahead through the vector memory path: a buffer load with a wave-uniform offset is
// counted by vmcnt, which nothing else in the loop uses, so it is requested at the top of an event and waited for at its
// end, for free.  (A scalar load shares lgkmcnt with the LDS requests and returns out of order: with one in flight
// every LDS wait of the event would have to drain the whole queue, the operand limbs requested for later included.)
struct TapeAhead {
#ifndef GECM_ROW_TAPE_SMEM
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t len;
    uint32_t pend;        // the word holding the byte requested last
    __device__ __forceinline__ void open(const uint32_t *tape, uint32_t tape_len)
    {
        len = tape_len;
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)tape, 0, (int)((tape_len + 3u) & ~3u), 0x00020000);
        pend = 0;
    }
    __device__ __forceinline__ void request(uint32_t pc) { pend = __builtin_amdgcn_raw_buffer_load_b32(rsrc, 0, (int)(pc & ~3u), 0); }
    __device__ __forceinline__ uint32_t take(uint32_t pc) const      // the byte at pc (the one requested), NOP past the end
    {
        const uint32_t b = __builtin_amdgcn_readfirstlane((pend >> ((pc & 3u) * 8u)) & 0xffu);
        return pc < len ? b : GECM_OP_NOP;
    }
#else   // A/B: the scalar load of rounds 1-2, issued when the byte is taken
    const uint32_t *tp;
    uint32_t len;
    __device__ __forceinline__ void open(const uint32_t *tape, uint32_t tape_len) { tp = tape; len = tape_len; }
    __device__ __forceinline__ void request(uint32_t) {}
    __device__ __forceinline__ uint32_t take(uint32_t pc) const
    {
        if (pc >= len) return GECM_OP_NOP;
        const uint32_t w = tp[pc >> 2];
        return __builtin_amdgcn_readfirstlane((w >> ((pc & 3u) * 8u)) & 0xffu);
    }
#endif
};

// ---- the LDS-prefetch variant (BC = 2) ----------------------------------------------------------------------------
// In a multiply a*b/R' the limbs of ONE operand have to reach all 16 lanes of the row: 16 v_mov_b32_dpp per multiply,
// a sixth of its VALU instructions, in a kernel that is bound by VALU issue.  Two of the three
// multiplies of a point addition have an operand that is OLD when the multiply starts:
//   level 1  (x_B -+ z_B)(x_A +- z_A): one of A, B is the point the previous step made, the other is older;
//   level 3  (U +- V)^2 * (z_C | x_C): C is two steps old.
// (The product is the same integer whichever operand is scanned, and so is everything computed from it.)  Those
// operands are therefore taken from LDS: when a point is made, the three forms the formulas read it in (sd, ds, oth,
// times 4) go to a slot of LDS — one ds_write per form, LDS pipe — and the multiply that needs one as its broadcast
// operand reads it back with ds_read_b128, every lane of the row reading the same 16 bytes (a broadcast, no bank
// conflict), requested ONE MULTIPLY AHEAD so that nothing waits for it.  The squaring in the middle (both operands
// fresh) and the 14 % of tape events outside the rule-3 loop keep the DPP multiply.
// tools/lds_bcast_ubench.hip (profiles/r03/lds_bcast_ubench_gfx950.txt): a 16-row body costs 900 cycles per wavefront
// at two wavefronts per SIMD with the DPP broadcast, 700 with the limbs in registers, 824 with the four reads, their
// wait and 16 register copies this kernel does not need.
// LDS holds, per DPP row of the workgroup, GECM_ROW_SLOTS point slots of 3 forms of 16*NQ limbs; a slot is written
// when its point is made and never changed; A, B, C name slots (wave-uniform numbers) as they name registers.
#ifndef GECM_ROW_WG_WAVES
#define GECM_ROW_WG_WAVES 4
#endif
#define GECM_ROW_SLOTS 5
#define GECM_ROW_WG_ROWS (4 * GECM_ROW_WG_WAVES)
template <int NQ>
struct RowSlots {
    int32_t f[GECM_ROW_SLOTS][3][16 * NQ];      // [slot][sd, ds, oth][limb]: one DPP row's points
};

template <int NQ>
__device__ __forceinline__ void lds_put_forms(RowSlots<NQ> *L, uint32_t slot, uint32_t l, const PtR<NQ> &p)
{
#pragma unroll
    for (int t = 0; t < NQ; t++) {
        L->f[slot][0][NQ * l + t] = (int32_t)((uint32_t)p.sd.v[t] << 2);
        L->f[slot][1][NQ * l + t] = (int32_t)((uint32_t)p.ds.v[t] << 2);
        L->f[slot][2][NQ * l + t] = (int32_t)((uint32_t)p.oth.v[t] << 2);
    }
}

// the first ROWS limbs of one form into every lane: ceil(ROWS/4) reads of 16 bytes at the same address in all lanes
template <int NQ, int ROWS>
__device__ __forceinline__ void lds_get_form(int32_t (&Ab)[ROWS], const RowSlots<NQ> *L, uint32_t slot, uint32_t form)
{
    typedef int32_t v4 __attribute__((ext_vector_type(4)));
    const v4 *src = reinterpret_cast<const v4 *>(&L->f[slot][form][0]);
    constexpr int N4 = (ROWS + 3) / 4;
#pragma unroll
    for (int k = 0; k < N4; k++) {
        const v4 v = src[k];
        if (4 * k + 0 < ROWS) Ab[4 * k + 0] = v.x;
        if (4 * k + 1 < ROWS) Ab[4 * k + 1] = v.y;
        if (4 * k + 2 < ROWS) Ab[4 * k + 2] = v.z;
        if (4 * k + 3 < ROWS) Ab[4 * k + 3] = v.w;
    }
}

// two slots none of A, B, C names (five slots, at most three in use)
__device__ __forceinline__ void free_slots(uint32_t sA, uint32_t sB, uint32_t sC, uint32_t &f0, uint32_t &f1)
{
    uint32_t mask = ~((1u << sA) | (1u << sB) | (1u << sC)) & ((1u << GECM_ROW_SLOTS) - 1u);
    f0 = (uint32_t)__builtin_ctz(mask);
    mask &= mask - 1u;
    f1 = (uint32_t)__builtin_ctz(mask);
}

template <int NQ, int ROWS>
__device__ __forceinline__ void run_tape_row_lds(const uint32_t *__restrict__ tape, uint32_t tape_len, PtR<NQ> &A,
                                                 const FeR<NQ> &s4, bool isZ, const RowSign &g, const RowMod<NQ> &m,
                                                 RowSlots<NQ> *L, uint32_t l)
{
    PtR<NQ> B = A, C = A;
    uint32_t sA = 0, sB = 0, sC = 0;
    lds_put_forms<NQ>(L, 0, l, A);
    auto is_fast = [](uint32_t op) { return (op & ~GECM_OP_SWAP) == (GECM_OP_STEP | GECM_OP_RULE3); };
    TapeAhead rd;
    rd.open(tape, tape_len);
    rd.request(0);
    uint32_t op = rd.take(0);
    rd.request(1);
    uint32_t nxt = rd.take(1);
    uint32_t pc = 0;
    while (pc < tape_len) {
        if (is_fast(op)) {
            // T = B + A (C); (B, T, C) <- (T, C, B), after the optional exchange of A and B (ecm.c:617-630, 683-713).
            // Level 1's scanned operand is the OLDER of the two points: the A of before the exchange — its ds form if
            // the exchange happens (it is B then), its sd form if not.
            int32_t Ab1[ROWS], Ab3[ROWS];
            lds_get_form<NQ, ROWS>(Ab1, L, sA, (op & GECM_OP_SWAP) ? 1u : 0u);
            do {
                rd.request(pc + 2);
                const bool sw = (op & GECM_OP_SWAP) != 0;
                if (sw) {
                    PtR<NQ> t = A;
                    A = B;
                    B = t;
                    const uint32_t ts = sA;
                    sA = sB;
                    sB = ts;
                }
                lds_get_form<NQ, ROWS>(Ab3, L, sC, 2u);              // level 3's operand, two multiplies ahead
                __builtin_amdgcn_sched_barrier(0);
                FeR<NQ> b1, w, t, e;
#pragma unroll
                for (int i = 0; i < NQ; i++) b1.v[i] = sw ? A.sd.v[i] : B.ds.v[i];
                fer_mul_pre<NQ, ROWS, true>(w, Ab1, b1, m);          // X: U      Z: V
                fer_other<NQ>(t, w);
                fer_sum_diff<NQ>(e, t, w, g);                        // X: V + U  Z: U - V
                __builtin_amdgcn_sched_barrier(0);
                // the next step's level-1 operand: A stays where it is, the form follows that step's exchange.
                // Requested whatever the next tape byte is (a branch here would leave the number of LDS requests in
                // flight unknown to the compiler's wait placement, which then waits for all of them at once).
                lds_get_form<NQ, ROWS>(Ab1, L, sA, (nxt & GECM_OP_SWAP) ? 1u : 0u);
                __builtin_amdgcn_sched_barrier(0);
                fer_mul<NQ, ROWS, true, false>(e, e, e, m);
                PtR<NQ> T;
                fer_mul_pre<NQ, ROWS, true>(T.own, Ab3, e, m);
                row_forms<NQ>(T, g);
                uint32_t f0, f1;
                free_slots(sA, sB, sC, f0, f1);
                lds_put_forms<NQ>(L, f0, l, T);
                C = B;
                sC = sB;
                B = T;
                sB = f0;
                pc++;
                op = nxt;
                nxt = rd.take(pc + 1);
            } while (is_fast(op));
            continue;
        }
        rd.request(pc + 2);
        if (op != GECM_OP_NOP) {
            const uint32_t rule = op & GECM_OP_RULE_MASK;
            const bool is_step = op >= GECM_OP_STEP;
            const bool do_add = op != GECM_OP_PRAC_BEGIN;
            const bool do_dup = op != GECM_OP_PRAC_END;
            if (is_step && (op & GECM_OP_SWAP)) {
                PtR<NQ> t = A;
                A = B;
                B = t;
                const uint32_t ts = sA;
                sA = sB;
                sB = ts;
            }
            if (is_step && rule == GECM_OP_RULE5) {
                PtR<NQ> t = B;
                B = C;
                C = t;
                const uint32_t ts = sB;
                sB = sC;
                sC = ts;
            } else if (is_step && rule == GECM_OP_RULE9) {
                PtR<NQ> t = A;
                A = B;
                B = C;
                C = t;
                const uint32_t ts = sA;
                sA = sB;
                sB = sC;
                sC = ts;
            } else if (op == GECM_OP_PRAC_BEGIN) {
                B = A;
                C = A;
                sB = sA;
                sC = sA;
            }
            PtR<NQ> T, D;
            uint32_t sT, sD;
            free_slots(sA, sB, sC, sT, sD);
            if (do_add) {
                row_add<NQ, ROWS, false>(T, B.ds, A.sd, C.oth, isZ, g, m);
                lds_put_forms<NQ>(L, sT, l, T);
            }
            if (do_dup) {
                row_dup<NQ, ROWS, false>(D, A.sd, s4, isZ, g, m);
                lds_put_forms<NQ>(L, sD, l, D);
            }
            if (op == GECM_OP_PRAC_END) {
                A = T;
                sA = sT;
            } else if (op == GECM_OP_PRAC_BEGIN) {
                A = D;
                sA = sD;
            } else if (rule == GECM_OP_RULE4) {
                B = T;
                sB = sT;
                A = D;
                sA = sD;
            } else if (rule == GECM_OP_RULE5) {
                PtR<NQ> t = C;
                const uint32_t ts = sC;
                C = T;
                sC = sT;
                B = t;
                sB = ts;
                A = D;
                sA = sD;
            } else {
                PtR<NQ> oldA = C;
                const uint32_t so = sC;
                C = T;
                sC = sT;
                B = D;
                sB = sD;
                A = oldA;
                sA = so;
            }
        }
        pc++;
        op = nxt;
        nxt = rd.take(pc + 1);
    }
}

#endif   // GECM_ROW_LDS_VARIANT

// Constants of the row kernel, one array of GECM_ROW_WORDS words per kind, limb j at word j (zero padded):
//   [0] N' = m*N, = -1 mod 2^28      [1] N      [2] c_in = R'^2 / R mod N (entry conversion: x R -> x R')
//   [3] R mod N (exit conversion)    [4] K' of N (bias that makes the exit value's limbs non-negative)
// (GECM_ROW_WORDS, GECM_ROW_KINDS: gecm_rowk.h)

// The whole stage-1 kernel body for one lane.  nl = limbs per residue in the device buffers (R = 2^(28*nl)).
// BC: how the scanned operand of a multiply reaches the lanes of its row — 0: DPP row_newbcast in the rows' wait
// states; 1: ds_swizzle, all requested at the top of the multiply (3 or more wavefronts per SIMD); 2: from LDS, read one
// multiply ahead, in the two multiplies of a point addition whose operand is old (run_tape_row_lds; `slots` = this
// workgroup's LDS, otherwise unused).
template <int NQ, int ROWS, int BC>
__device__ __forceinline__ void stage1_row(const uint32_t *__restrict__ tape, uint32_t tape_len, uint32_t *__restrict__ X,
                                           uint32_t *__restrict__ Z, const uint32_t *__restrict__ S, size_t stride,
                                           uint32_t nl, const uint32_t *__restrict__ rc, uint32_t rho_n,
                                           void *slots_ = nullptr)
{
#ifdef GECM_ROW_LDS_VARIANT
    RowSlots<NQ> *slots = static_cast<RowSlots<NQ> *>(slots_);
#else
    (void)slots_;
#endif
    constexpr bool ALDS = BC == 1;
#ifdef GECM_ROW_PRIO
    // experiment (tools/build_row_variant.sh): with 8 wavefronts per workgroup, wavefronts w and w + 4 share a SIMD;
    // give one of each pair a higher issue priority so that it runs as if alone and the other fills its gaps
    if ((threadIdx.x >> 8) & 1u) __builtin_amdgcn_s_setprio(GECM_ROW_PRIO);
#endif
    const uint32_t cidx = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;      // wavefronts of a workgroup are independent
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
    mp.c16 = mn.c16 = row_c16();
    PtR<NQ> P;
    FeR<NQ> s4, t;
    fer_load<NQ>(t, mine, stride, cidx, l, nl);
    fer_mul<NQ, ROWS, true, ALDS>(P.own, t, cin, mp);             // x*R -> x*R' (mod N')
    RowSign g;
    g.mz = isZ ? 0xffffffffu : 0u;
    g.bz = isZ ? 1u : 0u;
    g.mx = ~g.mz;
    g.bx = 1u - g.bz;
    row_forms<NQ>(P, g);
    fer_load<NQ>(t, S, stride, cidx, l, nl);
    fer_mul<NQ, ROWS, true, ALDS>(s4, t, cin, mp);
#ifdef GECM_ROW_LDS_VARIANT
    if constexpr (BC == 2) run_tape_row_lds<NQ, ROWS>(tape, tape_len, P, s4, isZ, g, mp, slots + (threadIdx.x >> 4), l);
    else
#endif
        run_tape_row<NQ, ROWS, ALDS>(tape, tape_len, P, s4, isZ, g, mp);
    fer_mul<NQ, ROWS, false, ALDS>(t, P.own, one, mn);                    // x*R' -> x*R (mod N), in (-N/16, 17N/16)
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
