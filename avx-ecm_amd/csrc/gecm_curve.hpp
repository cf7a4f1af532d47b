// gecm_curve.hpp — Montgomery-curve XZ point arithmetic and the stage-1 tape interpreter.
//
// Replaces, for the GPU, the reference's L1 layer:
//   vec_add        ecm.c:407-443   (4 mul + 2 sqr + 1 addsub)
//   vec_duplicate  ecm.c:445-457   (3 mul + 2 sqr + 1 sub + 1 add)
//   prac           ecm.c:565-884   (rules 3,4,5,9 of Montgomery's Table 4; ORIG_PRAC undefined)
//   ecm_stage1     ecm.c:1806-1854
// The sequence of field operations per point operation is exactly the reference's, so every
// intermediate is the same element of Z/N (projective X,Z are not normalised, so the formula
// sequence matters — SURVEY.md §8a).
//
// All lanes of all waves execute the same chain for the same B1, so the host walks
// ecm_stage1/prac ONCE and emits a byte tape (gecm_tape.h); the kernel is an interpreter whose
// control flow is wave-uniform (tape byte -> SGPR -> s_cbranch), with the three live points
// A=pt1, B=pt2, C=pt3 of prac() resident in VGPRs for the whole stage.  The reference's pointer
// swaps (ecm.c:624-629, 704-711) become register renames at the end of each tape step.
#pragma once
#include "gecm_field.hpp"
#include "gecm_tape.h"

template <int NL>
struct Pt {
    Fe<NL> X, Z;
};

// P = 2*(point whose sum/diff are s, d).  ecm.c:447-454.
template <int NL, class MOD>
__device__ __forceinline__ void pt_dup(Pt<NL> &out, const Fe<NL> &s, const Fe<NL> &d, const Fe<NL> &s4,
                                       const MOD &m)
{
    Fe<NL> t1, t2, t3;
    fe_sqr(t1, d, m);        // V = (x-z)^2
    fe_sqr(t2, s, m);        // U = (x+z)^2
    fe_mul(out.X, t1, t2, m);  // X = U*V
    fe_sub(t3, t2, t1, m);   // w = U - V
    fe_mul(t2, t3, s4, m);   // t = (A+2)/4 * w
    fe_add(t2, t2, t1);      // t = t + V
    fe_mul(out.Z, t2, t3, m);  // Z = t*w
}

template <int NL>
__device__ __forceinline__ void pt_sumdiff(Fe<NL> &s, Fe<NL> &d, const Pt<NL> &p, const ModK<NL> &m)
{
    fe_add(s, p.X, p.Z);
    fe_sub(d, p.X, p.Z, m);
}

// First half of pt_add: pp = (U+V)^2, mm = (U-V)^2  (ecm.c:417-422)
template <int NL, class MOD>
__device__ __forceinline__ void pt_add_uv(Fe<NL> &pp, Fe<NL> &mm, const Fe<NL> &s1, const Fe<NL> &d1,
                                          const Fe<NL> &s2, const Fe<NL> &d2, const MOD &m)
{
    Fe<NL> u, v;
    fe_mul(u, d1, s2, m);   // U
    fe_mul(v, s1, d2, m);   // V
    fe_add(pp, u, v);       // U + V
    fe_sub(mm, u, v, m);    // U - V
    fe_sqr(pp, pp, m);      // (U+V)^2
    fe_sqr(mm, mm, m);      // (U-V)^2
}

// Where the third PRAC point C lives.  Up to NL = 19 all of A, B, C fit in the 256 VGPRs that
// 2 waves/SIMD allow.  Above that C is parked in LDS, [coord][limb][lane] (conflict-free: lane l hits
// bank l): C is only read for the last two multiplies of an addition and written once per step, and a
// wave's 2*NL*256 bytes (15 KB at NL=30) leave room for the 8 waves of a CU in the 160 KB.
template <int NL, bool IN_LDS>
struct CStore;

template <int NL>
struct CStore<NL, false> {
    Pt<NL> c;
    __device__ __forceinline__ void put(const Pt<NL> &p) { c = p; }
    __device__ __forceinline__ void get(Pt<NL> &p) const { p = c; }
    __device__ __forceinline__ void getX(Fe<NL> &x) const { x = c.X; }
    __device__ __forceinline__ void getZ(Fe<NL> &z) const { z = c.Z; }
};

template <int NL>
struct CStore<NL, true> {
    uint32_t *lds;   // this lane's column: word (coord*NL + limb)*64
    __device__ __forceinline__ void put(const Pt<NL> &p)
    {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            lds[i * 64] = p.X.v[i];
            lds[(NL + i) * 64] = p.Z.v[i];
        }
    }
    __device__ __forceinline__ void getX(Fe<NL> &x) const
    {
#pragma unroll
        for (int i = 0; i < NL; i++) x.v[i] = lds[i * 64];
    }
    __device__ __forceinline__ void getZ(Fe<NL> &z) const
    {
#pragma unroll
        for (int i = 0; i < NL; i++) z.v[i] = lds[(NL + i) * 64];
    }
    __device__ __forceinline__ void get(Pt<NL> &p) const
    {
        getX(p.X);
        getZ(p.Z);
    }
};

#ifndef GECM_C_LDS_ABOVE
#define GECM_C_LDS_ABOVE 19
#endif
template <int NL>
struct TapePolicy {
    static constexpr bool c_in_lds = (NL > GECM_C_LDS_ABOVE);
};

// Run a tape on point P (held in A).  Returns with the result in A.
//
// The loop body contains exactly ONE inlined point addition and ONE inlined doubling (~55 KB of
// straight-line code at NL=15, inside the 64 KB instruction cache); each tape op selects their
// operands and destinations with wave-uniform branches and register moves (<2% of a step).
// Register budget (<=256 VGPRs for 2 waves/SIMD): the difference point is only copied after
// the first half of the addition, and s = (A+2)/4 is re-read from memory for each doubling
// (doublings are ~10% of the steps) instead of occupying NL registers throughout.
template <int NL, class MOD>
__device__ __forceinline__ void run_tape(const uint32_t *__restrict__ tape, uint32_t tape_len, Pt<NL> &A,
                                         const uint32_t *__restrict__ S, size_t stride, uint32_t idx,
                                         const MOD &m, CStore<NL, TapePolicy<NL>::c_in_lds> &cst)
{
    Pt<NL> B = A;
    cst.put(A);
    auto fetch = [&](uint32_t pc) -> uint32_t {
        // one tape byte; past the end reads as NOP (the host pads the tape with zero words)
        uint32_t w = tape[pc >> 2];
        return __builtin_amdgcn_readfirstlane((w >> ((pc & 3u) * 8u)) & 0xffu);
    };
    // The tape byte of the NEXT event is fetched at the start of the current one, so its scalar-load
    // latency hides behind ~13k cycles of arithmetic instead of being paid at every step.
    uint32_t nxt = tape_len ? fetch(0) : GECM_OP_NOP;
    for (uint32_t pc = 0; pc < tape_len; pc++) {
        uint32_t op = nxt;
        nxt = (pc + 1 < tape_len) ? fetch(pc + 1) : GECM_OP_NOP;
        // Fast path, as its own inner loop: rule 3 is 93% of the point additions at B1=1e6
        // (1,762,907 of 1,902,102 steps) and comes in long runs.  Straight-line code with no
        // operand selection; only A, B, C are live around it.
        // ecm.c:617-630 (swap), 683-713: T = B + A (C); (B,T,C) <- (T,C,B)
#ifndef GECM_NO_FASTPATH
        while ((op & ~GECM_OP_SWAP) == (GECM_OP_STEP | GECM_OP_RULE3)) {
            if (op & GECM_OP_SWAP) {
                Pt<NL> t = A;
                A = B;
                B = t;
            }
            Fe<NL> s1, d1, s2, d2, pp, mm;
            pt_sumdiff(s1, d1, B, m);
            pt_sumdiff(s2, d2, A, m);
            pt_add_uv(pp, mm, s1, d1, s2, d2, m);
            Pt<NL> T;
            {
                Fe<NL> cz;
                cst.getZ(cz);
                fe_mul(T.X, pp, cz, m);
            }
            {
                Fe<NL> cx;
                cst.getX(cx);
                fe_mul(T.Z, mm, cx, m);
            }
            cst.put(B);
            B = T;
            pc++;
            op = nxt;
            nxt = (pc + 1 < tape_len) ? fetch(pc + 1) : GECM_OP_NOP;
        }
#endif
        if (op == GECM_OP_NOP) continue;
        // Slow path (rules 4, 5, 9, PRAC_BEGIN, PRAC_END: 14% of the events).  Every one of them is
        // "T = B' + A' (difference C'), D = 2A'" for a renaming (A',B',C') of the three points, so
        // the points are permuted into that canonical order with register moves, ONE straight-line
        // add+double core runs, and the results are written back through the inverse renaming.
        // No operand is selected inside the arithmetic, which keeps register pressure low.
        const uint32_t rule = op & GECM_OP_RULE_MASK;
        const bool is_step = op >= GECM_OP_STEP;
        const bool do_add = op != GECM_OP_PRAC_BEGIN;
        const bool do_dup = op != GECM_OP_PRAC_END;
        Pt<NL> C;
        cst.get(C);
        if (is_step && (op & GECM_OP_SWAP)) {       // ecm.c:617-630
            Pt<NL> t = A;
            A = B;
            B = t;
        }
        if (is_step && rule == GECM_OP_RULE5) {     // C = C + A (B); A = 2A  (ecm.c:728-740): B <-> C
            Pt<NL> t = B;
            B = C;
            C = t;
        } else if (is_step && rule == GECM_OP_RULE9) {   // C = C + B (A); B = 2B  (ecm.c:853-865): (A,B,C) <- (B,C,A)
            Pt<NL> t = A;
            A = B;
            B = C;
            C = t;
        } else if (op == GECM_OP_PRAC_BEGIN) {      // B = C = A  (ecm.c:603-608)
            B = A;
            C = A;
        }
        Pt<NL> T, D;
        {
            Fe<NL> s1, d1, s2, d2;
            pt_sumdiff(s2, d2, A, m);
            if (do_add) {                            // T = B + A, difference C  (ecm.c:417-440)
                Fe<NL> pp, mm;
                pt_sumdiff(s1, d1, B, m);
                pt_add_uv(pp, mm, s1, d1, s2, d2, m);
                fe_mul(T.X, pp, C.Z, m);
                fe_mul(T.Z, mm, C.X, m);
            }
            if (do_dup) {                            // D = 2A  (ecm.c:447-454)
                Fe<NL> s4;
                fe_load(s4, S, stride, idx);
                pt_dup(D, s2, d2, s4, m);
            }
        }
        if (op == GECM_OP_PRAC_END) {               // P = A + B (C)   ecm.c:868-873
            A = T;
        } else if (op == GECM_OP_PRAC_BEGIN) {      // A = 2A          ecm.c:613
            A = D;
        } else if (rule == GECM_OP_RULE4) {         // B = T; A = D    ecm.c:721-722
            B = T;
            A = D;
        } else if (rule == GECM_OP_RULE5) {         // C = T (sits in B); A = D; undo B <-> C
            Pt<NL> t = C;
            C = T;
            B = t;
            A = D;
        } else {                                    // rule 9: C = T, B = D; undo (A,B,C) <- (B,C,A)
            Pt<NL> oldA = C;
            C = T;
            B = D;
            A = oldA;
        }
        cst.put(C);
    }
}

// ======================================================================================
// Two lanes per curve ("split-coordinate" stage 1) — for batches that do not fill the chip.
//
// MI355X has 1024 SIMDs; with one curve per lane a batch of B curves is B/64 wavefronts, so below
// 65536 curves some SIMDs have no wave at all, and between full rounds of 2 waves/SIMD (131072 curves)
// the last round runs part-empty.  Stage 1 is one long dependent chain per curve, so the only way to
// use the idle SIMDs is to split a curve.  The XZ formulas are two-way parallel at every level: in
// vec_add (ecm.c:407-443)
// U=(x1-z1)(x2+z2) | V=(x1+z1)(x2-z2), then (U+V)^2 | (U-V)^2, then *z3 | *x3; in vec_duplicate
// (ecm.c:445-457) (x+z)^2 | (x-z)^2, then U*V | s*(U-V), then - | (s*w+V)*w.  So lane 2j holds the X
// coordinate and lane 2j+1 the Z coordinate of curve j's points A, B, C; each stage is ONE multiply
// executed by both lanes on their own operands, and the halves are exchanged with a DPP quad
// permutation (v_mov_b32_dpp quad_perm:[1,0,3,2], no LDS, no memory).  A point addition is 3
// multiply-times instead of 6, a doubling 3 instead of 5, with twice the waves for the same batch:
// measured 1.83x the curves/s for batches <= 32768, equal at 65536, 3% slower at a full 131072
// (the idle third multiply of a doubling and the exchanges) — tools/lanes_bench.py, DESIGN.md §5.
// The field operations and their operands are exactly those of the one-lane-per-curve kernel, so
// the results are the same integers.  The sign of each add/sub differs between the two lanes of a
// pair; it is a per-lane select on the subtrahend (fe_addsub_lane), not a branch.
template <int NL>
__device__ __forceinline__ void fe_partner(Fe<NL> &r, const Fe<NL> &a)
{
#pragma unroll
    for (int i = 0; i < NL; i++)
        r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, true);
}

// r = x + y on lanes with neg == false, x - y + K on lanes with neg == true (limb-wise, lazy).
template <int NL>
__device__ __forceinline__ void fe_addsub_lane(Fe<NL> &r, const Fe<NL> &x, const Fe<NL> &y, bool neg,
                                               const ModK<NL> &m)
{
#pragma unroll
    for (int i = 0; i < NL; i++) {
        uint32_t t = neg ? m.kp[i] - y.v[i] : y.v[i];
        r.v[i] = x.v[i] + t;
    }
    if (LazyPolicy<NL>::norm_sub) fe_weak_norm(r);
}

// own coordinate of P -> X lane: P.X + P.Z, Z lane: P.X - P.Z
template <int NL>
__device__ __forceinline__ void pair_sum_diff(Fe<NL> &r, const Fe<NL> &own, bool isZ, const ModK<NL> &m)
{
    Fe<NL> oth;
    fe_partner(oth, own);
    fe_addsub_lane(r, oth, own, isZ, m);
}

// own coordinate of P -> X lane: P.X - P.Z, Z lane: P.X + P.Z
template <int NL>
__device__ __forceinline__ void pair_diff_sum(Fe<NL> &r, const Fe<NL> &own, bool isZ, const ModK<NL> &m)
{
    Fe<NL> oth;
    fe_partner(oth, own);
    fe_addsub_lane(r, own, oth, !isZ, m);
}

// T = P1 + P2 with difference C: fB = pair_diff_sum(P1), fA = pair_sum_diff(P2).  ecm.c:417-440
template <int NL, class MOD>
__device__ __forceinline__ void pair_add(Fe<NL> &T, const Fe<NL> &fB, const Fe<NL> &fA, const Fe<NL> &c, bool isZ,
                                         const MOD &m)
{
    Fe<NL> w, t, e;
    fe_mul(w, fB, fA, m);              // X lane: U = (x1-z1)(x2+z2)     Z lane: V = (x1+z1)(x2-z2)
    fe_partner(t, w);
    fe_addsub_lane(e, t, w, isZ, m);   // X lane: V + U                  Z lane: U - V
    fe_sqr(e, e, m);
    fe_partner(t, c);                  // X lane: C.Z                    Z lane: C.X
    fe_mul(T, e, t, m);                // X lane: (U+V)^2 * z3           Z lane: (U-V)^2 * x3
}

// D = 2P: fA = pair_sum_diff(P), s4 = (A+2)/4 of this curve.  ecm.c:447-454
template <int NL, class MOD>
__device__ __forceinline__ void pair_dup(Fe<NL> &D, const Fe<NL> &fA, const Fe<NL> &s4, bool isZ,
                                         const MOD &m)
{
    Fe<NL> q, t, w, p1, p2, r1;
    fe_sqr(q, fA, m);                  // X lane: U = (x+z)^2            Z lane: V = (x-z)^2
    fe_partner(t, q);                  // X lane: V                      Z lane: U
    fe_sub(w, t, q, m);                // Z lane: w = U - V              (X lane: unused)
#pragma unroll
    for (int i = 0; i < NL; i++) {
        p1.v[i] = isZ ? s4.v[i] : q.v[i];
        p2.v[i] = isZ ? w.v[i] : t.v[i];
    }
    fe_mul(r1, p1, p2, m);             // X lane: X = U*V                Z lane: t = s*w
    fe_add(t, r1, q);                  // Z lane: t + V
    fe_mul(t, t, w, m);                // Z lane: Z = (t+V)*w            (X lane: idle multiply)
#pragma unroll
    for (int i = 0; i < NL; i++) D.v[i] = isZ ? t.v[i] : r1.v[i];
}

// run_tape for the split-coordinate layout: A, B, C are this lane's coordinate of prac()'s three
// points.  Same tape, same renamings as run_tape above.
template <int NL, class MOD>
__device__ __forceinline__ void run_tape_pair(const uint32_t *__restrict__ tape, uint32_t tape_len, Fe<NL> &A,
                                              const uint32_t *__restrict__ S, size_t stride, uint32_t cidx,
                                              bool isZ, const MOD &m)
{
    Fe<NL> B = A, C = A;
    auto fetch = [&](uint32_t pc) -> uint32_t {
        uint32_t w = tape[pc >> 2];
        return __builtin_amdgcn_readfirstlane((w >> ((pc & 3u) * 8u)) & 0xffu);
    };
    uint32_t nxt = tape_len ? fetch(0) : GECM_OP_NOP;
    for (uint32_t pc = 0; pc < tape_len; pc++) {
        uint32_t op = nxt;
        nxt = (pc + 1 < tape_len) ? fetch(pc + 1) : GECM_OP_NOP;
        // rule 3: T = B + A (C); (B,T,C) <- (T,C,B)
        while ((op & ~GECM_OP_SWAP) == (GECM_OP_STEP | GECM_OP_RULE3)) {
            if (op & GECM_OP_SWAP) {
                Fe<NL> t = A;
                A = B;
                B = t;
            }
            Fe<NL> fA, fB, T;
            pair_diff_sum(fB, B, isZ, m);
            pair_sum_diff(fA, A, isZ, m);
            pair_add(T, fB, fA, C, isZ, m);
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
            Fe<NL> t = A;
            A = B;
            B = t;
        }
        if (is_step && rule == GECM_OP_RULE5) {
            Fe<NL> t = B;
            B = C;
            C = t;
        } else if (is_step && rule == GECM_OP_RULE9) {
            Fe<NL> t = A;
            A = B;
            B = C;
            C = t;
        } else if (op == GECM_OP_PRAC_BEGIN) {
            B = A;
            C = A;
        }
        Fe<NL> T, D;
        {
            Fe<NL> fA;
            pair_sum_diff(fA, A, isZ, m);
            if (do_add) {
                Fe<NL> fB;
                pair_diff_sum(fB, B, isZ, m);
                pair_add(T, fB, fA, C, isZ, m);
            }
            if (do_dup) {
                Fe<NL> s4;
                fe_load(s4, S, stride, cidx);
                pair_dup(D, fA, s4, isZ, m);
            }
        }
        if (op == GECM_OP_PRAC_END) {
            A = T;
        } else if (op == GECM_OP_PRAC_BEGIN) {
            A = D;
        } else if (rule == GECM_OP_RULE4) {
            B = T;
            A = D;
        } else if (rule == GECM_OP_RULE5) {
            Fe<NL> t = C;
            C = T;
            B = t;
            A = D;
        } else {
            Fe<NL> oldA = C;
            C = T;
            B = D;
            A = oldA;
        }
    }
}
