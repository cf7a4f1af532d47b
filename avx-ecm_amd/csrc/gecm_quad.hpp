// gecm_quad.hpp — stage 1 with EIGHT lanes per curve: the X and the Z coordinate of a curve's points on two
// adjacent quads of lanes, and inside a quad each lane holds NQ = ceil(NL/4) of the residue's limbs (lane l:
// limbs NQ*l .. NQ*l+NQ-1).
//
// For batches of a few thousand curves (BASELINE configs[1]: 4096) even two lanes per curve leave 7 SIMDs of
// 8 without a wavefront.  The last place left to split is the multiplication itself.  It is done row-wise
// (operand scanning, "CIOS"): for each limb a_i of the first operand, broadcast to the quad by a DPP
// quad_perm, every lane adds a_i * b[own 4 limbs] and q_i * N[own 4 limbs] into its four 64-bit window
// accumulators (q_i = Montgomery digit, computed by lane 0 and broadcast the same way), then the window
// moves down one limb: each lane folds the upper part of its lowest accumulator into its next one and hands
// the low 28 bits to the lane below.  That is 8 multiply-adds and ~13
// cheap operations per row and lane, 15 rows for a 420-bit radix: about half the time of the full
// product-scanning multiply, with four times the lanes.  The digits q_i and the result are the integers
// the generic fe_mul computes (same REDC, R = 2^(28*NL)), so the three layouts are interchangeable on the
// same device buffers; the exit canonicalisation is left to a separate one-lane kernel (k_canon).
#pragma once
#include "gecm_curve.hpp"

template <int NL>
struct QuadShape {
    static constexpr int NQ = (NL + 3) / 4;      // limbs per lane: lane l of a quad holds limbs NQ*l .. NQ*l+NQ-1
};

template <int NQ>
struct FeQn {
    uint32_t v[NQ];
};

template <int NL>
struct QuadMod {
    uint32_t n[QuadShape<NL>::NQ];    // this lane's limbs of N
    uint32_t kp[QuadShape<NL>::NQ];   // this lane's limbs of K'
    uint32_t rho;
    uint32_t top_mask;   // 0 on the last lane of the quad, ~0 elsewhere
    bool is0;            // first lane of the quad
};

template <int S>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, S * 0x55 /* quad_perm:[S,S,S,S] */, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t quad_from_above(uint32_t x)   // lane l <- lane l+1 (lane 3: undefined, masked by caller)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xF9 /* quad_perm:[1,2,3,3] */, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t quad_from_below(uint32_t x)   // lane l <- lane l-1 (lane 0: undefined, masked by caller)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x90 /* quad_perm:[0,0,1,2] */, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t other_coord(uint32_t x)       // lane <-> lane ^ 4
{
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)x, 0x101F /* bit mode: and 0x1f, or 0, xor 4 */);
}

// T[j] += x * y[j] for 1..4 accumulators in one asm statement
__device__ __forceinline__ void mad1(uint64_t &t0, uint32_t x, uint32_t y0)
{
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(t0) : "v"(x), "v"(y0) : "vcc");
}
__device__ __forceinline__ void mad2(uint64_t &t0, uint64_t &t1, uint32_t x, uint32_t y0, uint32_t y1)
{
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %2, %4, %1" : "+v"(t0), "+v"(t1) : "v"(x), "v"(y0), "v"(y1) : "vcc");
}
__device__ __forceinline__ void mad3(uint64_t &t0, uint64_t &t1, uint64_t &t2, uint32_t x, uint32_t y0, uint32_t y1, uint32_t y2)
{
    asm("v_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_mad_u64_u32 %1, vcc, %3, %5, %1\n\tv_mad_u64_u32 %2, vcc, %3, %6, %2"
        : "+v"(t0), "+v"(t1), "+v"(t2) : "v"(x), "v"(y0), "v"(y1), "v"(y2) : "vcc");
}
__device__ __forceinline__ void mad4(uint64_t &t0, uint64_t &t1, uint64_t &t2, uint64_t &t3, uint32_t x, uint32_t y0, uint32_t y1,
                                     uint32_t y2, uint32_t y3)
{
    asm("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %6, %1\n\t"
        "v_mad_u64_u32 %2, vcc, %4, %7, %2\n\tv_mad_u64_u32 %3, vcc, %4, %8, %3"
        : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(x), "v"(y0), "v"(y1), "v"(y2), "v"(y3) : "vcc");
}

// one row: T[(t + ROT) % NQ] += x * y[t] for t = T0 .. NQ-1
template <int NQ, int ROT, int T0 = 0>
__device__ __forceinline__ void mad_row(uint64_t (&T)[NQ], uint32_t x, const uint32_t (&y)[NQ])
{
    constexpr int left = NQ - T0;
    if constexpr (left >= 4) {
        mad4(T[(T0 + ROT) % NQ], T[(T0 + 1 + ROT) % NQ], T[(T0 + 2 + ROT) % NQ], T[(T0 + 3 + ROT) % NQ], x, y[T0], y[T0 + 1],
             y[T0 + 2], y[T0 + 3]);
        mad_row<NQ, ROT, T0 + 4>(T, x, y);
    } else if constexpr (left == 3) {
        mad3(T[(T0 + ROT) % NQ], T[(T0 + 1 + ROT) % NQ], T[(T0 + 2 + ROT) % NQ], x, y[T0], y[T0 + 1], y[T0 + 2]);
    } else if constexpr (left == 2) {
        mad2(T[(T0 + ROT) % NQ], T[(T0 + 1 + ROT) % NQ], x, y[T0], y[T0 + 1]);
    } else if constexpr (left == 1) {
        mad1(T[(T0 + ROT) % NQ], x, y[T0]);
    }
}

// r = a*b/R mod N (lazy, limbs < 2^28): the same integer fe_mul<NL> returns.
template <int NL>
__device__ __forceinline__ void feq_mul(FeQn<QuadShape<NL>::NQ> &r, const FeQn<QuadShape<NL>::NQ> &a,
                                        const FeQn<QuadShape<NL>::NQ> &b, const QuadMod<NL> &m)
{
    constexpr int NQ = QuadShape<NL>::NQ;
    uint64_t T[NQ];
#pragma unroll
    for (int t = 0; t < NQ; t++) T[t] = 0;
    static_for<0, NL>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int rot = i % NQ;                       // logical slot t of the window lives in T[(t + i) % NQ]
        const uint32_t ai = quad_bcast<i / NQ>(a.v[i % NQ]);
        mad_row<NQ, rot>(T, ai, b.v);
        const uint32_t q = quad_bcast<0>(((uint32_t)T[rot] * m.rho) & GECM_LIMB_MASK);
        mad_row<NQ, rot>(T, q, m.n);
        // the window moves down one limb: every lane folds the part of its lowest accumulator that lies above
        // 28 bits into its next one (it has that weight) and hands the low 28 bits to the lane below, whose
        // new top slot they are.  Lane 0's low 28 bits are zero by construction of q and go nowhere.
        const uint64_t old0 = T[rot];
        T[(1 + i) % NQ] += old0 >> GECM_LIMB_BITS;
        const uint32_t lo = quad_from_above((uint32_t)old0 & GECM_LIMB_MASK);
        T[rot] = (uint64_t)(lo & m.top_mask);                                         // new top slot (0 on lane 3)
    });
    // carry propagation to limbs < 2^28: inside the lane, then one hand-over to the lane above, then
    // (almost never) single-bit ripples until no lane has a carry left
    uint64_t carry = 0;
    uint32_t o[NQ];
    static_for<0, NQ>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        const uint64_t v = T[(t + NL) % NQ] + carry;
        o[t] = (uint32_t)v & GECM_LIMB_MASK;
        carry = v >> GECM_LIMB_BITS;
    });
    uint32_t clo = quad_from_below((uint32_t)carry), chi = quad_from_below((uint32_t)(carry >> 32));
    uint64_t cin = m.is0 ? 0ull : (((uint64_t)chi << 32) | clo);
    for (;;) {
        uint64_t c = cin;
#pragma unroll
        for (int t = 0; t < NQ; t++) {
            const uint64_t v = (uint64_t)o[t] + c;
            o[t] = (uint32_t)v & GECM_LIMB_MASK;
            c = v >> GECM_LIMB_BITS;
        }
        // a carry out of lane 3 cannot happen (the result is < R); carries out of lanes 0..2 are 0 or 1
        const uint32_t up = quad_from_below((uint32_t)c);
        cin = m.is0 ? 0ull : (uint64_t)up;
        if (__builtin_amdgcn_ballot_w64(cin != 0) == 0) break;
    }
#pragma unroll
    for (int t = 0; t < NQ; t++) r.v[t] = o[t];
}

// fe_weak_norm across the quad: limbs < 2^30 in, limbs < 2^28 + 4 out, value unchanged (needed above 25 limbs,
// LazyPolicy, so that accumulator sums stay below 2^64)
template <int NL>
__device__ __forceinline__ void feq_weak_norm(FeQn<QuadShape<NL>::NQ> &r, const QuadMod<NL> &m)
{
    constexpr int NQ = QuadShape<NL>::NQ;
    const uint32_t below = quad_from_below(r.v[NQ - 1] >> GECM_LIMB_BITS);     // carry of the lane below's top limb
    FeQn<NQ> o;
    o.v[0] = (r.v[0] & GECM_LIMB_MASK) + (m.is0 ? 0u : below);
#pragma unroll
    for (int t = 1; t < NQ; t++) o.v[t] = (r.v[t] & GECM_LIMB_MASK) + (r.v[t - 1] >> GECM_LIMB_BITS);
    // the very top limb of the residue (limb NL-1, on lane 3) keeps its upper bits, as in fe_weak_norm, and
    // the padding slots above it stay as they are (zero)
    constexpr int t_top = (NL - 1) - 3 * NQ;
    static_assert(t_top >= 0 && t_top < NQ, "limb NL-1 must sit on the last lane of the quad");
    const bool last = m.top_mask == 0u;
#pragma unroll
    for (int t = 0; t < NQ; t++) {
        if (t == t_top) o.v[t] = last ? o.v[t] + (r.v[t] & ~GECM_LIMB_MASK) : o.v[t];
        else if (t > t_top) o.v[t] = last ? r.v[t] : o.v[t];
    }
    r = o;
}

template <int NL>
__device__ __forceinline__ void feq_add(FeQn<QuadShape<NL>::NQ> &r, const FeQn<QuadShape<NL>::NQ> &a, const FeQn<QuadShape<NL>::NQ> &b)
{
#pragma unroll
    for (int t = 0; t < QuadShape<NL>::NQ; t++) r.v[t] = a.v[t] + b.v[t];
}

template <int NL>
__device__ __forceinline__ void feq_sub(FeQn<QuadShape<NL>::NQ> &r, const FeQn<QuadShape<NL>::NQ> &a, const FeQn<QuadShape<NL>::NQ> &b, const QuadMod<NL> &m)
{
#pragma unroll
    for (int t = 0; t < QuadShape<NL>::NQ; t++) r.v[t] = a.v[t] + m.kp[t] - b.v[t];
    if (LazyPolicy<NL>::norm_sub) feq_weak_norm<NL>(r, m);
}

// r = x + y on lanes with neg == false, x - y + K on lanes with neg == true
template <int NL>
__device__ __forceinline__ void feq_addsub_lane(FeQn<QuadShape<NL>::NQ> &r, const FeQn<QuadShape<NL>::NQ> &x, const FeQn<QuadShape<NL>::NQ> &y, bool neg, const QuadMod<NL> &m)
{
#pragma unroll
    for (int t = 0; t < QuadShape<NL>::NQ; t++) {
        const uint32_t s = neg ? m.kp[t] - y.v[t] : y.v[t];
        r.v[t] = x.v[t] + s;
    }
    if (LazyPolicy<NL>::norm_sub) feq_weak_norm<NL>(r, m);
}

template <int NL>
__device__ __forceinline__ void feq_other(FeQn<QuadShape<NL>::NQ> &r, const FeQn<QuadShape<NL>::NQ> &a)
{
#pragma unroll
    for (int t = 0; t < QuadShape<NL>::NQ; t++) r.v[t] = other_coord(a.v[t]);
}

// the point arithmetic of gecm_curve.hpp's two-lane layout, on quads
template <int NL>
__device__ __forceinline__ void quad_sum_diff(FeQn<QuadShape<NL>::NQ> &r, const FeQn<QuadShape<NL>::NQ> &own, bool isZ, const QuadMod<NL> &m)
{
    FeQn<QuadShape<NL>::NQ> oth;
    feq_other<NL>(oth, own);
    feq_addsub_lane<NL>(r, oth, own, isZ, m);          // X lanes: Z + X      Z lanes: X - Z
}

template <int NL>
__device__ __forceinline__ void quad_diff_sum(FeQn<QuadShape<NL>::NQ> &r, const FeQn<QuadShape<NL>::NQ> &own, bool isZ, const QuadMod<NL> &m)
{
    FeQn<QuadShape<NL>::NQ> oth;
    feq_other<NL>(oth, own);
    feq_addsub_lane<NL>(r, own, oth, !isZ, m);         // X lanes: X - Z      Z lanes: Z + X
}

template <int NL>
__device__ __forceinline__ void quad_add(FeQn<QuadShape<NL>::NQ> &T, const FeQn<QuadShape<NL>::NQ> &fB, const FeQn<QuadShape<NL>::NQ> &fA, const FeQn<QuadShape<NL>::NQ> &c, bool isZ, const QuadMod<NL> &m)
{
    FeQn<QuadShape<NL>::NQ> w, t, e;
    feq_mul<NL>(w, fB, fA, m);                         // X: U      Z: V
    feq_other<NL>(t, w);
    feq_addsub_lane<NL>(e, t, w, isZ, m);              // X: V + U  Z: U - V
    feq_mul<NL>(e, e, e, m);                           // squares (no symmetry saving in the row-wise form)
    feq_other<NL>(t, c);                               // X: C.Z    Z: C.X
    feq_mul<NL>(T, e, t, m);
}

template <int NL>
__device__ __forceinline__ void quad_dup(FeQn<QuadShape<NL>::NQ> &D, const FeQn<QuadShape<NL>::NQ> &fA, const FeQn<QuadShape<NL>::NQ> &s4, bool isZ, const QuadMod<NL> &m)
{
    FeQn<QuadShape<NL>::NQ> q, t, w, p1, p2, r1;
    feq_mul<NL>(q, fA, fA, m);                         // X: U = (x+z)^2    Z: V = (x-z)^2
    feq_other<NL>(t, q);                               // X: V              Z: U
    feq_sub<NL>(w, t, q, m);                           // Z: w = U - V
#pragma unroll
    for (int i = 0; i < QuadShape<NL>::NQ; i++) {
        p1.v[i] = isZ ? s4.v[i] : q.v[i];
        p2.v[i] = isZ ? w.v[i] : t.v[i];
    }
    feq_mul<NL>(r1, p1, p2, m);                        // X: U*V            Z: s*w
    feq_add<NL>(t, r1, q);                         // Z: s*w + V
    feq_mul<NL>(t, t, w, m);                           // Z: (s*w + V)*w
#pragma unroll
    for (int i = 0; i < QuadShape<NL>::NQ; i++) D.v[i] = isZ ? t.v[i] : r1.v[i];
}

template <int NL>
__device__ __forceinline__ void feq_load(FeQn<QuadShape<NL>::NQ> &r, const uint32_t *__restrict__ base, size_t stride, uint32_t cidx, uint32_t l)
{
#pragma unroll
    for (int t = 0; t < QuadShape<NL>::NQ; t++) {
        const uint32_t limb = (uint32_t)QuadShape<NL>::NQ * l + (uint32_t)t;
        r.v[t] = limb < (uint32_t)NL ? base[(size_t)limb * stride + cidx] : 0u;
    }
}

template <int NL>
__device__ __forceinline__ void feq_store(uint32_t *__restrict__ base, size_t stride, uint32_t cidx, uint32_t l, const FeQn<QuadShape<NL>::NQ> &r)
{
#pragma unroll
    for (int t = 0; t < QuadShape<NL>::NQ; t++) {
        const uint32_t limb = (uint32_t)QuadShape<NL>::NQ * l + (uint32_t)t;
        if (limb < (uint32_t)NL) base[(size_t)limb * stride + cidx] = r.v[t];
    }
}

// run_tape_pair of gecm_curve.hpp on quads: A, B, C are this lane's 4 limbs of its coordinate.
template <int NL>
__device__ __forceinline__ void run_tape_quad(const uint32_t *__restrict__ tape, uint32_t tape_len, FeQn<QuadShape<NL>::NQ> &A,
                                              const uint32_t *__restrict__ S, size_t stride, uint32_t cidx, uint32_t l,
                                              bool isZ, const QuadMod<NL> &m)
{
    FeQn<QuadShape<NL>::NQ> B = A, C = A;
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
                FeQn<QuadShape<NL>::NQ> t = A;
                A = B;
                B = t;
            }
            FeQn<QuadShape<NL>::NQ> fA, fB, T;
            quad_diff_sum<NL>(fB, B, isZ, m);
            quad_sum_diff<NL>(fA, A, isZ, m);
            quad_add<NL>(T, fB, fA, C, isZ, m);
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
            FeQn<QuadShape<NL>::NQ> t = A;
            A = B;
            B = t;
        }
        if (is_step && rule == GECM_OP_RULE5) {
            FeQn<QuadShape<NL>::NQ> t = B;
            B = C;
            C = t;
        } else if (is_step && rule == GECM_OP_RULE9) {
            FeQn<QuadShape<NL>::NQ> t = A;
            A = B;
            B = C;
            C = t;
        } else if (op == GECM_OP_PRAC_BEGIN) {
            B = A;
            C = A;
        }
        FeQn<QuadShape<NL>::NQ> T, D;
        {
            FeQn<QuadShape<NL>::NQ> fA;
            quad_sum_diff<NL>(fA, A, isZ, m);
            if (do_add) {
                FeQn<QuadShape<NL>::NQ> fB;
                quad_diff_sum<NL>(fB, B, isZ, m);
                quad_add<NL>(T, fB, fA, C, isZ, m);
            }
            if (do_dup) {
                FeQn<QuadShape<NL>::NQ> s4;
                feq_load<NL>(s4, S, stride, cidx, l);
                quad_dup<NL>(D, fA, s4, isZ, m);
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
            FeQn<QuadShape<NL>::NQ> t = C;
            C = T;
            B = t;
            A = D;
        } else {
            FeQn<QuadShape<NL>::NQ> oldA = C;
            C = T;
            B = D;
            A = oldA;
        }
    }
}
