// gecm_stage2.hpp — stage 2 (standard continuation) on the device.
//
// Replaces, for the GPU:
//   next_pt_vec                 ecm.c:886-976    binary Montgomery ladder, uniform multiplier
//   ecm_stage2_init             ecm.c:2201-2340  baby-step table Pb[map[j]], j <= U*D, gcd(j,D)=1
//   batch_invert_pt_inplace     ecm.c:1869-2001  Montgomery's trick + one inversion
//   batch_invert_pt_to_bignum   ecm.c:2003-2136
//   ecm_stage2_pair             ecm.c:2342-2540  giant-step window + CROSS_PRODUCT_INV (ecm.c:1857-1859)
//
// One curve per lane as in stage 1; control flow is wave-uniform (the pair map, the keep-bitmap of
// the baby steps and the ladder bits are the same for every curve).  Tables live in HBM,
// [entry][limb][curve], so every access is a coalesced 256-byte row per limb.
//
// Differences from the reference that do not change any result (inverses mod N are unique and
// the accumulator is a product in a commutative ring):
//   * the per-lane host mpz_invert (ecm.c:1919-1950, 2054-2085) becomes a fixed-iteration inversion
//     on the device (fe_invert: division steps in batches of 28); a non-invertible product records gcd(product, N) per
//     curve instead of writing it into stg2acc (ecm.c:1927-1939);
//   * the baby-step table is normalised in blocks of S2_BLK entries (one inversion per block)
//     rather than in one 7.7k-entry pass, so only the normalised X of each entry is kept in HBM;
//   * giant steps are produced in chunks of G steps with one inversion per chunk into a ring of
//     normalised X/Z (the reference re-inverts 2U new steps at every window shift,
//     ecm.c:2458-2502: 1,341 inversions at B2 = 1e8; here 84 + 31 for the table).
#pragma once
#include "gecm_curve.hpp"

#define S2_BLK 256

template <int NL>
struct S2Const {
    ModK<NL> m;
    Fe<NL> one;   // R mod N
    Fe<NL> r3;    // R^3 mod N  (plain inverse -> Montgomery form of the inverse)
    uint32_t inv_iters;   // batches of 28 division steps of the inversion (fe_invert)
};

// Tables are tiled per wavefront: [wave][entry][limb][lane].  One entry of one wave is NL*256
// contiguous bytes (3.84 KB at NL=15), so a pair step reads one contiguous chunk of the 60-GB
// baby-step table instead of NL 256-byte pieces half a megabyte apart ([entry][limb][curve] order):
// one DRAM page and one TLB entry per access instead of NL.
template <int NL>
struct Tab {
    uint32_t *base;
    uint32_t nent;      // entries per wave
};
template <int NL>
__device__ __forceinline__ const uint32_t *tb_ptr(const uint32_t *base, uint32_t nent, uint32_t idx, size_t e)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(idx >> 6);
    return base + ((size_t)wave * nent + e) * (size_t)(NL * 64);
}
template <int NL>
__device__ __forceinline__ void tb_load(Fe<NL> &r, const uint32_t *__restrict__ base, uint32_t nent, uint32_t idx, size_t e)
{
    const char *p = (const char *)tb_ptr<NL>(base, nent, idx, e);
    const uint32_t boff = (idx & 63u) * 4u;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = *(const uint32_t *)(p + i * 256 + boff);
}
template <int NL>
__device__ __forceinline__ void tb_store(uint32_t *__restrict__ base, uint32_t nent, uint32_t idx, size_t e, const Fe<NL> &r)
{
    char *p = (char *)const_cast<uint32_t *>(tb_ptr<NL>(base, nent, idx, e));
    const uint32_t boff = (idx & 63u) * 4u;
#pragma unroll
    for (int i = 0; i < NL; i++) *(uint32_t *)(p + i * 256 + boff) = r.v[i];
}

// Table rows whose arrival the pair walk waits for by hand.  The compiler's own s_waitcnt placement drains the whole
// queue (vmcnt(0)) at the top of every pair — the row load that is conditional on the giant step changing makes its
// count unknown — so the rows requested for the NEXT pairs were waited for too and the lookahead hid nothing.  These
// loads are inline asm (the compiler does not count them); tb_wait() waits until at most `newer` x NL loads issued
// after the row are still out (loads complete in order; compiler-issued loads in between only make the wait longer)
// and ties the row's registers to that point.  vmcnt holds 6 bits: above 63 the wait is simply for more.
// No "memory" clobber on these statements: the tables are read-only in the kernels that use them, and a clobber would
// stop the compiler from fetching the wave-uniform tape words with scalar loads (its vector loads come with vmcnt(0)).
#ifndef GECM_S2_ASYNC_MAXNL
#define GECM_S2_ASYNC_MAXNL 30
#endif
template <int NL>
__device__ __forceinline__ void tb_load_async(Fe<NL> &r, const uint32_t *__restrict__ base, uint32_t nent, uint32_t idx, size_t e)
{
    // Only while every row in flight stays in registers: a row the compiler moves to scratch would be moved before it
    // has arrived.  Larger residues keep compiler-visible loads.  tests/test_abi_cpu.py checks the built objects: the
    // pair-walk kernels up to GECM_S2_ASYNC_MAXNL limbs use no scratch and spill nothing; tools/check_async_rows.py
    // checks their ISA (no instruction but the loads touches a row register between request and wait).
    if constexpr (NL > GECM_S2_ASYNC_MAXNL) {
        tb_load(r, base, nent, idx, e);
        return;
    }
    // the row address is wave-uniform; say so in a way that survives the table pointer arriving in vector registers
    const uint64_t pv = (uint64_t)tb_ptr<NL>(base, nent, idx, e);
    const uint32_t phi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pv >> 32));
    const uint32_t plo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)pv);      // the builtin returns int: no sign extension
    const uint32_t *p = (const uint32_t *)(((uint64_t)phi << 32) | (uint64_t)plo);
    const uint32_t boff = (idx & 63u) * 4u;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const uint32_t *pc = p + (i / 16) * 1024;              // the immediate offset reaches 4095 bytes
        asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(r.v[i]) : "v"(boff), "s"(pc), "n"((i % 16) * 256));
    }
}
template <int NL, int I0, int N>
__device__ __forceinline__ void tb_tie(Fe<NL> &r)
{
    if constexpr (NL > GECM_S2_ASYNC_MAXNL) return;
    if constexpr (N >= 8)
        asm volatile("" : "+v"(r.v[I0]), "+v"(r.v[I0 + 1]), "+v"(r.v[I0 + 2]), "+v"(r.v[I0 + 3]), "+v"(r.v[I0 + 4]), "+v"(r.v[I0 + 5]),
                     "+v"(r.v[I0 + 6]), "+v"(r.v[I0 + 7]));
    else if constexpr (N >= 1)
        asm volatile("" : "+v"(r.v[I0]));
    if constexpr (N >= 8) tb_tie<NL, I0 + 8, N - 8>(r);
    else if constexpr (N >= 2) tb_tie<NL, I0 + 1, N - 1>(r);
}
template <int NL, int NEWER>
__device__ __forceinline__ void tb_wait_cnt()
{
    constexpr int cnt = NL * NEWER > 63 ? 63 : NL * NEWER;
    if constexpr (NL <= GECM_S2_ASYNC_MAXNL) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(cnt));
}
template <int NL, int NEWER>
__device__ __forceinline__ void tb_wait(Fe<NL> &r)
{
    tb_wait_cnt<NL, NEWER>();
    tb_tie<NL, 0, NL>(r);
}

// ---- exact helpers on fully normalised values ----------------------------------------------
// r = a - b, returns borrow (1 if a < b); limbs normalised in and out (result mod 2^(28 NL))
template <int NL>
__device__ __forceinline__ uint32_t fe_sub_borrow(Fe<NL> &r, const Fe<NL> &a, const Fe<NL> &b)
{
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        uint32_t d = a.v[i] - b.v[i] - borrow;
        borrow = d >> 31;
        r.v[i] = d & GECM_LIMB_MASK;
    }
    return borrow;
}
template <int NL>
__device__ __forceinline__ void fe_add_carry(Fe<NL> &r, const Fe<NL> &a, const uint32_t (&b)[NL])
{
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        uint32_t t = a.v[i] + b[i] + c;
        c = t >> GECM_LIMB_BITS;
        r.v[i] = (i == NL - 1) ? t : (t & GECM_LIMB_MASK);
    }
}
template <int NL>
__device__ __forceinline__ void fe_shr1(Fe<NL> &r)
{
#pragma unroll
    for (int i = 0; i < NL - 1; i++) r.v[i] = (r.v[i] >> 1) | ((r.v[i + 1] & 1u) << (GECM_LIMB_BITS - 1));
    r.v[NL - 1] >>= 1;
}
template <int NL>
__device__ __forceinline__ void fe_select(Fe<NL> &r, bool c, const Fe<NL> &a, const Fe<NL> &b)
{
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = c ? a.v[i] : b.v[i];
}

// x^-1 mod N for canonical x (plain integers, not Montgomery form) and gcd(x, N), in a fixed number of steps
// (branch-free per lane).  Bernstein-Yang division steps in the batched form of libsecp256k1's modinv32, on 28-bit
// limbs: a batch runs 28 division steps on the low words of (f, g) = (N, x) with cheap 32-bit instructions and
// yields a 2x2 transition matrix (entries below 2^28 in size), which is then applied to the full (f, g) and — modulo
// N, with the exact division by 2^28 done the Montgomery way — to the cofactors (d, e), d*x = f, e*x = g (mod N),
// with v_mad_i64_i32 rows.  After `batches` >= ceil((floor((45907 bits + 26313) / 19929) + 1) / 28) batches (the
// published bound for the "half-delta" variant) g = 0 and f = +-gcd(x, N).  About 26k instructions at 15 limbs where
// the bit-by-bit binary inversion this replaces took 500k (one inversion cost as much as a thousand multiplications;
// tools/s2_subseq_check.py history in DESIGN.md §7).  Returns true iff the inverse exists; g receives the gcd.
__device__ __forceinline__ void imad(int64_t &acc, int32_t x, int32_t y)
{
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
}

template <int NL>
__device__ __noinline__ bool fe_invert(Fe<NL> &r, Fe<NL> &gout, const Fe<NL> &x, const ModK<NL> &m, uint32_t batches)
{
    constexpr int32_t M28 = (int32_t)GECM_LIMB_MASK;
    int32_t f[NL], g[NL], d[NL], e[NL];     // low limbs in [0, 2^28), the top limb carries the sign
#pragma unroll
    for (int i = 0; i < NL; i++) {
        f[i] = (int32_t)m.n[i];
        g[i] = (int32_t)x.v[i];
        d[i] = 0;
        e[i] = (i == 0) ? 1 : 0;
    }
    const uint32_t ninv = (0u - m.rho) & GECM_LIMB_MASK;        // N^-1 mod 2^28
    int32_t zeta = -1;
    for (uint32_t bt = 0; bt < batches; bt++) {
        // 28 division steps on the low words
        uint32_t u = 1, v = 0, q = 0, w = 1, fl = (uint32_t)f[0], gl = (uint32_t)g[0];
#pragma unroll
        for (int i = 0; i < GECM_LIMB_BITS; i++) {
            uint32_t c1 = (uint32_t)(zeta >> 31);
            const uint32_t c2 = 0u - (gl & 1u);
            const uint32_t xx = (fl ^ c1) - c1, yy = (u ^ c1) - c1, zz = (v ^ c1) - c1;
            gl += xx & c2; q += yy & c2; w += zz & c2;
            c1 &= c2;
            zeta = (zeta ^ (int32_t)c1) - 1;
            fl += gl & c1; u += q & c1; v += w & c1;
            gl >>= 1; u <<= 1; v <<= 1;
        }
        const int32_t U = (int32_t)u, V = (int32_t)v, Q = (int32_t)q, W = (int32_t)w;
        // (d, e) <- (U d + V e, Q d + W e) / 2^28 mod N
        {
            const int32_t sd = d[NL - 1] >> 31, se = e[NL - 1] >> 31;
            int32_t md = (U & sd) + (V & se), me = (Q & sd) + (W & se);
            int64_t cd = 0, ce = 0;
            imad(cd, U, d[0]); imad(cd, V, e[0]);
            imad(ce, Q, d[0]); imad(ce, W, e[0]);
            md -= (int32_t)((ninv * (uint32_t)cd + (uint32_t)md) & GECM_LIMB_MASK);
            me -= (int32_t)((ninv * (uint32_t)ce + (uint32_t)me) & GECM_LIMB_MASK);
            imad(cd, (int32_t)m.n[0], md);
            imad(ce, (int32_t)m.n[0], me);
            cd >>= GECM_LIMB_BITS;
            ce >>= GECM_LIMB_BITS;
#pragma unroll
            for (int i = 1; i < NL; i++) {
                imad(cd, U, d[i]); imad(cd, V, e[i]); imad(cd, (int32_t)m.n[i], md);
                imad(ce, Q, d[i]); imad(ce, W, e[i]); imad(ce, (int32_t)m.n[i], me);
                d[i - 1] = (int32_t)cd & M28; cd >>= GECM_LIMB_BITS;
                e[i - 1] = (int32_t)ce & M28; ce >>= GECM_LIMB_BITS;
            }
            d[NL - 1] = (int32_t)cd;
            e[NL - 1] = (int32_t)ce;
        }
        // (f, g) <- (U f + V g, Q f + W g) / 2^28   (exact)
        {
            int64_t cf = 0, cg = 0;
            imad(cf, U, f[0]); imad(cf, V, g[0]);
            imad(cg, Q, f[0]); imad(cg, W, g[0]);
            cf >>= GECM_LIMB_BITS;
            cg >>= GECM_LIMB_BITS;
#pragma unroll
            for (int i = 1; i < NL; i++) {
                imad(cf, U, f[i]); imad(cf, V, g[i]);
                imad(cg, Q, f[i]); imad(cg, W, g[i]);
                f[i - 1] = (int32_t)cf & M28; cf >>= GECM_LIMB_BITS;
                g[i - 1] = (int32_t)cg & M28; cg >>= GECM_LIMB_BITS;
            }
            f[NL - 1] = (int32_t)cf;
            g[NL - 1] = (int32_t)cg;
        }
    }
    // f = +-gcd: make it positive, and give d the same sign change; then d into [0, N)
    const int32_t sf = f[NL - 1] >> 31;
    auto negate_if = [&](int32_t (&a)[NL], int32_t mask) {
        int32_t c = mask & 1;
#pragma unroll
        for (int i = 0; i < NL - 1; i++) {
            const int32_t t = ((a[i] ^ mask) & M28) + c;
            a[i] = t & M28;
            c = t >> GECM_LIMB_BITS;
        }
        a[NL - 1] = (a[NL - 1] ^ mask) + c;
    };
    auto add_n_if = [&](int32_t (&a)[NL], int32_t mask) {
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < NL - 1; i++) {
            const int32_t t = a[i] + ((int32_t)m.n[i] & mask) + c;
            a[i] = t & M28;
            c = t >> GECM_LIMB_BITS;
        }
        a[NL - 1] += ((int32_t)m.n[NL - 1] & mask) + c;
    };
    negate_if(f, sf);
    negate_if(d, sf);                         // d in (-2N, 2N)
    add_n_if(d, d[NL - 1] >> 31);
    add_n_if(d, d[NL - 1] >> 31);             // d in [0, 2N)
    Fe<NL> dv, dn;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        dv.v[i] = (uint32_t)d[i];
        gout.v[i] = (uint32_t)f[i];
    }
    Fe<NL> nn;
#pragma unroll
    for (int i = 0; i < NL; i++) nn.v[i] = m.n[i];
    const uint32_t lt = fe_sub_borrow(dn, dv, nn);          // dv - N
    fe_select(r, lt != 0, dv, dn);
    bool ok = (gout.v[0] == 1u);
#pragma unroll
    for (int i = 1; i < NL; i++) ok = ok && (gout.v[i] == 0);
    return ok;
}

// Montgomery form of the inverse of a Montgomery-form value: a = x R  ->  x^-1 R.
// On failure r = 0 and *fail receives gcd(x R mod N, N) = gcd(x, N).  The LAST failure wins, as in the reference,
// which overwrites its accumulator with the gcd every time an inversion fails (ecm.c:1925-1939); the host makes the
// last chunk of giant steps coincide with the reference's last batch (gecm_stage2_pair).
template <int NL>
__device__ __forceinline__ void fe_inv_mont(Fe<NL> &r, const Fe<NL> &a, const S2Const<NL> &k, uint32_t *__restrict__ fail,
                                            size_t stride, uint32_t idx)
{
    Fe<NL> c, t, g;
    fe_canonical_mont(c, a, k.one, k.m);          // canonical x R
    bool ok = fe_invert(t, g, c, k.m, k.inv_iters);   // (x R)^-1
    fe_mul(r, t, k.r3, k.m);                      // (xR)^-1 R^3 / R = x^-1 R
    if (!ok) {
#pragma unroll
        for (int i = 0; i < NL; i++) r.v[i] = 0;
        bool nz = false;
#pragma unroll
        for (int i = 0; i < NL; i++) nz = nz || g.v[i] != 0;
        if (nz) fe_store(fail, stride, idx, g);
    }
}

// Montgomery's trick on n <= S2_BLK points whose X, Z sit in bx, bz (block tables):
// out[e0 + i] = X_i / Z_i (Montgomery form, lazy-normalised).  bp = scratch for prefix products.
// tgt == nullptr: entry i of the block goes to table entry e0 + i; else to tgt[e0 + i] (a sub-sequence's kept
// entries are not consecutive in the table).  sidx = index used for the block scratch (a virtual curve per
// sub-sequence), idx = the curve's own index (table, failure record).
template <int NL>
__device__ __forceinline__ void block_normalise(uint32_t *__restrict__ out, uint32_t out_nent, size_t e0,
                                                const uint32_t *__restrict__ bx, const uint32_t *__restrict__ bz,
                                                uint32_t *__restrict__ bp, uint32_t n, const S2Const<NL> &k,
                                                uint32_t *__restrict__ fail, size_t stride, uint32_t idx,
                                                const uint32_t *__restrict__ tgt = nullptr, uint32_t sidx = 0xffffffffu)
{
    if (sidx == 0xffffffffu) sidx = idx;
    auto where = [&](size_t i) -> size_t {
        return tgt ? (size_t)__builtin_amdgcn_readfirstlane(tgt[e0 + i]) : e0 + i;
    };
    Fe<NL> acc, z, x, t;
    tb_load(acc, bz, S2_BLK, sidx, 0);
    tb_store(bp, S2_BLK, sidx, 0, acc);
    for (uint32_t i = 1; i < n; i++) {            // prefix products  (ecm.c:1889-1893)
        tb_load(z, bz, S2_BLK, sidx, i);
        fe_mul(acc, acc, z, k.m);
        tb_store(bp, S2_BLK, sidx, i, acc);
    }
    Fe<NL> inv;
    fe_inv_mont(inv, acc, k, fail, stride, idx);  // (prod Z)^-1
    for (uint32_t i = n - 1; i > 0; i--) {        // suffix walk  (ecm.c:1965-1987)
        tb_load(t, bp, S2_BLK, sidx, i - 1);
        fe_mul(t, t, inv, k.m);                   // Z_i^-1
        tb_load(z, bz, S2_BLK, sidx, i);
        fe_mul(inv, inv, z, k.m);
        tb_load(x, bx, S2_BLK, sidx, i);
        fe_mul(x, x, t, k.m);
        tb_store(out, out_nent, idx, where(i), x);
    }
    tb_load(x, bx, S2_BLK, sidx, 0);
    fe_mul(x, x, inv, k.m);
    tb_store(out, out_nent, idx, where(0), x);
}

// The ladders of the stage-2 set-up run a handful of times per launch, but there are eight call sites of them and each
// holds eleven residue multiplies: inlined, the 37-limb translation unit took over five minutes to compile.  From
// GECM_LADDER_OL_NL limbs on the ladder's multiplies are calls to one out-of-line multiply and one square (operands
// through memory: ~3*NL words against NL^2 multiply-adds).  (The whole ladder out of line was tried first: with this
// ROCm it gave wrong points, and from 28 limbs on stopped the code generator inside k_s2_init_k.)
#ifndef GECM_LADDER_OL_NL
#define GECM_LADDER_OL_NL 20
#endif
template <int NL>
struct ModKOut {
    const ModK<NL> &m;
};
template <int NL>
__device__ __noinline__ void fe_mul_ol(Fe<NL> &r, const Fe<NL> &a, const Fe<NL> &b, const ModK<NL> &m)
{
    Fe<NL> x = a, y = b, t;
    fe_mul(t, x, y, m);
    r = t;
}
template <int NL>
__device__ __noinline__ void fe_sqr_ol(Fe<NL> &r, const Fe<NL> &a, const ModK<NL> &m)
{
    Fe<NL> x = a, t;
    fe_sqr(t, x, m);
    r = t;
}
template <int NL>
__device__ __forceinline__ void fe_mul(Fe<NL> &r, const Fe<NL> &a, const Fe<NL> &b, const ModKOut<NL> &o) { fe_mul_ol(r, a, b, o.m); }
template <int NL>
__device__ __forceinline__ void fe_sqr(Fe<NL> &r, const Fe<NL> &a, const ModKOut<NL> &o) { fe_sqr_ol(r, a, o.m); }
template <int NL>
__device__ __forceinline__ void fe_sub(Fe<NL> &r, const Fe<NL> &a, const Fe<NL> &b, const ModKOut<NL> &o) { fe_sub(r, a, b, o.m); }

// P <- [c]P, binary ladder (next_pt_vec, ecm.c:886-976); c is wave-uniform.  MM = ModK<NL> or ModKOut<NL>.
template <int NL, class MM>
__device__ __forceinline__ void pt_ladder_body(Pt<NL> &P, uint64_t c, const Fe<NL> &s4, const ModK<NL> &m, const MM &mm_)
{
    if (c == 1) return;
    Fe<NL> s1, d1, s2, d2;
    Pt<NL> p1 = P, p2;
    pt_sumdiff(s1, d1, P, m);
    pt_dup(p2, s1, d1, s4, mm_);
    if (c == 2) { P = p2; return; }
    int top = 63 - __builtin_clzll(c);
    for (int bit = top - 1; bit >= 0; bit--) {
        uint32_t b = (uint32_t)((c >> bit) & 1);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b) { Pt<NL> t = p1; p1 = p2; p2 = t; }     // so that p2 is always the one added-to
        // now: add into p2 <- p1 + p2 (diff P), double p1   [bit 0]; with the swap this is the
        // reference's "add x1, duplicate x2" for bit 1 (ecm.c:945-960)
        pt_sumdiff(s2, d2, p2, m);
        pt_sumdiff(s1, d1, p1, m);
        Fe<NL> pp, mm;
        pt_add_uv(pp, mm, s1, d1, s2, d2, mm_);
        Pt<NL> T, D;
        fe_mul(T.X, pp, P.Z, mm_);
        fe_mul(T.Z, mm, P.X, mm_);
        pt_dup(D, s1, d1, s4, mm_);
        p2 = T;
        p1 = D;
        if (b) { Pt<NL> t = p1; p1 = p2; p2 = t; }
    }
    P = p1;
}
template <int NL>
__device__ __forceinline__ void pt_ladder(Pt<NL> &P, uint64_t c, const Fe<NL> &s4, const ModK<NL> &m)
{
    if constexpr (NL >= GECM_LADDER_OL_NL) pt_ladder_body(P, c, s4, m, ModKOut<NL>{m});
    else pt_ladder_body(P, c, s4, m, m);
}

struct S2InitArgs {
    const uint32_t *X, *Z, *S;       // Q = P after stage 1 (Montgomery form), s = (A+2)/4
    uint32_t *PbX;                   // out: normalised baby steps, entries 0..npb-1 (0 unused)
    uint32_t *bx, *bz, *bp;          // block scratch, S2_BLK entries each
    uint32_t *PdX, *PdZ;             // out: Pd = [D]Q
    uint32_t *acc;                   // out: accumulator = one
    uint32_t *fail;                  // per-curve gcd record of a failed inversion (zeroed by host)
    const uint32_t *keep;            // bitmap over j: bit j set iff map[j] > 0
    uint32_t umax, D, npb;
    size_t stride;
    // K sub-sequences per curve (small batches, s2_init_k): table index of the i-th kept member of sub-sequence r
    // at tgt[tgt_off[r] + i]; block scratch kbx/kbz/kbp per (curve block, r); PdK = [K*D]Q for the giant steps
    uint32_t K;
    const uint32_t *tgt, *tgt_off;
    uint32_t *kbx, *kbz, *kbp;
    uint32_t *PdKX, *PdKZ;
};

// ecm_stage2_init, ecm.c:2201-2340
template <int NL>
__device__ __forceinline__ void s2_init(const S2InitArgs &a, const S2Const<NL> &k, uint32_t idx)
{
    const ModK<NL> &m = k.m;
    const size_t stride = a.stride;
    Pt<NL> Q, P1, P3;
    Fe<NL> s4, sQ, dQ;
    fe_load(Q.X, a.X, stride, idx);
    fe_load(Q.Z, a.Z, stride, idx);
    fe_load(s4, a.S, stride, idx);
    pt_sumdiff(sQ, dQ, Q, m);
    pt_dup(P1, sQ, dQ, s4, m);                 // [2]Q     ecm.c:2243-2244
    P3 = Q;
    // entries 1 and 2
    tb_store(a.bx, S2_BLK, idx, 0, Q.X);  tb_store(a.bz, S2_BLK, idx, 0, Q.Z);
    tb_store(a.bx, S2_BLK, idx, 1, P1.X); tb_store(a.bz, S2_BLK, idx, 1, P1.Z);
    uint32_t nblk = 2, e0 = 1;
    for (uint32_t j = 3; j <= a.umax; j++) {    // ecm.c:2263-2313
        Fe<NL> s1, d1, pp, mm;
        pt_sumdiff(s1, d1, P1, m);
        pt_add_uv(pp, mm, s1, d1, sQ, dQ, m);
        Pt<NL> T;
        fe_mul(T.X, pp, P3.Z, m);
        fe_mul(T.Z, mm, P3.X, m);
        uint32_t kb = (a.keep[j >> 5] >> (j & 31)) & 1u;
        kb = __builtin_amdgcn_readfirstlane(kb);
        if (kb) {
            tb_store(a.bx, S2_BLK, idx, nblk, T.X);
            tb_store(a.bz, S2_BLK, idx, nblk, T.Z);
            nblk++;
            if (nblk == S2_BLK) {
                block_normalise<NL>(a.PbX, a.npb, e0, a.bx, a.bz, a.bp, nblk, k, a.fail, stride, idx);
                e0 += nblk;
                nblk = 0;
            }
        }
        P3 = P1;
        P1 = T;
    }
    if (nblk) block_normalise<NL>(a.PbX, a.npb, e0, a.bx, a.bz, a.bp, nblk, k, a.fail, stride, idx);
    Pt<NL> Pd = Q;
    pt_ladder(Pd, (uint64_t)a.D, s4, m);        // Pd = [w]Q   ecm.c:2332-2334
    Fe<NL> c;
    fe_canonical_mont(c, Pd.X, k.one, m); fe_store(a.PdX, stride, idx, c);
    fe_canonical_mont(c, Pd.Z, k.one, m); fe_store(a.PdZ, stride, idx, c);
    fe_store(a.acc, stride, idx, k.one);        // acc = one   ecm.c:2318
}

// ecm_stage2_init with K sub-sequences per curve: sub-sequence r holds the multiples j = r, r+K, r+2K, ... (j >= 1;
// r = 0 starts at K) and steps by [K]Q: P_(j+K) = P_j + [K]Q with difference P_(j-K) — the same points [j]Q as the
// reference's chain j -> j+1 (ecm.c:2263-2313), as other projective representatives, and only X/Z of them is kept.
// A batch of a few thousand curves is a few dozen wavefronts; with K = 32 chains per curve it fills the chip.
// r is wave-uniform (blockIdx.x % K); idx is the curve.
template <int NL>
__device__ __forceinline__ void s2_init_k(const S2InitArgs &a, const S2Const<NL> &k, uint32_t idx, uint32_t r)
{
    const ModK<NL> &m = k.m;
    const size_t stride = a.stride;
    const uint32_t K = a.K;
    const uint32_t sidx = ((idx >> 6) * K + r) * 64u + (idx & 63u);      // virtual curve: scratch of (curve block, r)
    Pt<NL> Q, PK, p1, p2;
    Fe<NL> s4, sK, dK;
    fe_load(Q.X, a.X, stride, idx);
    fe_load(Q.Z, a.Z, stride, idx);
    fe_load(s4, a.S, stride, idx);
    PK = Q;
    pt_ladder(PK, (uint64_t)K, s4, m);
    pt_sumdiff(sK, dK, PK, m);
    const uint32_t j0 = r ? r : K;                  // first member
    // a failing inversion goes to plane 1 + r of the record, as in giant_chunk_k: the K sub-sequences of a curve are
    // K wavefronts, and two of them failing with different gcds must not write the same limbs (plane 0 belongs to
    // the single-chain inversions)
    uint32_t *const failp = a.fail + (size_t)(1 + r) * NL * stride;
    const uint32_t toff = __builtin_amdgcn_readfirstlane(a.tgt_off[r]);
    uint32_t nblk = 0, e0 = toff;
    uint32_t mi = 0;
    for (uint32_t j = j0; j <= a.umax; j += K, mi++) {
        Pt<NL> T;
        if (mi < 2) {
            T = Q;
            pt_ladder(T, (uint64_t)j, s4, m);       // the first two members by the ladder (next_pt_vec)
        } else {
            Fe<NL> s1, d1, pp, mm;
            pt_sumdiff(s1, d1, p1, m);
            pt_add_uv(pp, mm, s1, d1, sK, dK, m);
            fe_mul(T.X, pp, p2.Z, m);
            fe_mul(T.Z, mm, p2.X, m);
        }
        uint32_t kb = (a.keep[j >> 5] >> (j & 31)) & 1u;
        kb = __builtin_amdgcn_readfirstlane(kb);
        if (kb) {
            tb_store(a.kbx, S2_BLK, sidx, nblk, T.X);
            tb_store(a.kbz, S2_BLK, sidx, nblk, T.Z);
            nblk++;
            if (nblk == S2_BLK) {
                block_normalise<NL>(a.PbX, a.npb, e0, a.kbx, a.kbz, a.kbp, nblk, k, failp, stride, idx, a.tgt, sidx);
                e0 += nblk;
                nblk = 0;
            }
        }
        p2 = p1;
        p1 = T;
    }
    if (nblk) block_normalise<NL>(a.PbX, a.npb, e0, a.kbx, a.kbz, a.kbp, nblk, k, failp, stride, idx, a.tgt, sidx);
    if (r == 0) {
        Pt<NL> Pd = Q;
        pt_ladder(Pd, (uint64_t)a.D, s4, m);        // Pd = [w]Q   ecm.c:2332-2334
        Fe<NL> c;
        fe_canonical_mont(c, Pd.X, k.one, m); fe_store(a.PdX, stride, idx, c);
        fe_canonical_mont(c, Pd.Z, k.one, m); fe_store(a.PdZ, stride, idx, c);
        pt_ladder(Pd, (uint64_t)K, s4, m);          // [K*D]Q: the giant steps' stride (giant_chunk_k)
        fe_canonical_mont(c, Pd.X, k.one, m); fe_store(a.PdKX, stride, idx, c);
        fe_canonical_mont(c, Pd.Z, k.one, m); fe_store(a.PdKZ, stride, idx, c);
        fe_store(a.acc, stride, idx, k.one);        // acc = one   ecm.c:2318
    }
}

struct S2PairArgs {
    const uint32_t *X, *Z, *S;       // Q, s
    const uint32_t *PbX;             // normalised baby steps
    uint32_t npb;
    const uint32_t *PdX, *PdZ;       // Pd = [D]Q
    uint32_t *gx, *gz;               // chunk scratch: X, Z of the giant steps being generated, G+2 entries
                                     // (entries 0,1 = the last two steps of the previous chunk)
    uint32_t *gp;                    // prefix products, G entries
    uint32_t *ring;                  // X/Z of the giant steps, ring of `ring_size` entries (power of two)
    uint32_t *acc;                   // in/out accumulator
    uint32_t *fail;
    const uint32_t *steps;           // pair tape, 2 words per step (see S2_STEP_GEN)
    uint32_t nsteps, D, G, ring_size;
    uint64_t A0;                     // multiplier of the first giant step: 2*amin*D   ecm.c:2378
    size_t stride;
    // K sub-sequences per curve (giant_chunk_k): scratch per (curve block, r) with Gs + 2 / Gs entries, the stride
    // point [K*D]Q, and one failure plane per sub-sequence after plane 0
    uint32_t K, Gs;
    uint32_t *kgx, *kgz, *kgp;
    const uint32_t *PdKX, *PdKZ;
};

// tape word 0 == S2_STEP_GEN: generate the next `word 1` giant steps (continuing the sequence) and
// normalise them into the ring.  Otherwise (slot, pb): acc *= ring[slot] - PbX[pb].
#define S2_STEP_GEN 0xffffffffu

// Giant steps.  The reference keeps a window of 2L = 4U steps and, at every window shift, makes 2U new
// ones and inverts them (ecm.c:2458-2502): 1,341 inversions at B2 = 1e8.  Here they are produced in
// chunks of G >> 2U steps with ONE inversion per chunk; the ring holds the normalised X/Z of every
// step a pair can still ask for.  Same points, same (unique) inverses, same accumulator.
// kprev > 1: the steps before first_abs were made by giant_chunk_k with kprev sub-sequences; the two previous steps are
// then the latest members of sub-sequences (first_abs-1) % kprev and (first_abs-2) % kprev.
template <int NL>
__device__ __forceinline__ void giant_chunk(const S2PairArgs &a, uint32_t first_abs, uint32_t n, bool very_first,
                                            const S2Const<NL> &k, uint32_t idx, uint32_t kprev = 1)
{
    // its own kernel launch (k_s2_gen, ~84 per curve batch at B2=1e8): everything it needs is read from
    // memory, so the pair-walk kernel keeps only the accumulator and two operand pairs live
    const ModK<NL> &m = k.m;
    const size_t stride = a.stride;
    Pt<NL> Q, Pd;
    Fe<NL> s4, sD, dD;
    fe_load(Pd.X, a.PdX, stride, idx);
    fe_load(Pd.Z, a.PdZ, stride, idx);
    pt_sumdiff(sD, dD, Pd, m);
    if (very_first) {
        fe_load(Q.X, a.X, stride, idx);
        fe_load(Q.Z, a.Z, stride, idx);
        fe_load(s4, a.S, stride, idx);
    }
    // The two most recent giant steps stay in registers across the loop (p1 = latest, p2 = the one
    // before); the scratch arrays only receive stores here, so no step waits on a store->load round
    // trip.  Scratch slot i+2 holds chunk entry i; slots 0,1 hold the two steps before the chunk.
    uint32_t start = 0;
    Pt<NL> p1, p2;
    if (very_first) {
        Pt<NL> P0 = Q, Pad = Q, T;
        pt_ladder(P0, a.A0, s4, m);                   // Pa[0] = [A]Q      ecm.c:2380-2383
        pt_ladder(Pad, a.A0 - a.D, s4, m);            // Pad = [A-D]Q     ecm.c:2388-2390
        Fe<NL> s1, d1, pp, mm;
        pt_sumdiff(s1, d1, P0, m);
        pt_add_uv(pp, mm, s1, d1, sD, dD, m);
        fe_mul(T.X, pp, Pad.Z, m);                    // Pa[1] = Pa[0] + Pd (Pad)   ecm.c:2395-2401
        fe_mul(T.Z, mm, Pad.X, m);
        tb_store(a.gx, a.G + 2, idx, 2, P0.X); tb_store(a.gz, a.G + 2, idx, 2, P0.Z);
        tb_store(a.gx, a.G + 2, idx, 3, T.X);  tb_store(a.gz, a.G + 2, idx, 3, T.Z);
        p2 = P0;
        p1 = T;
        start = 2;
    } else if (kprev > 1) {
        const uint32_t r1 = (first_abs - 1) % kprev, r2 = (first_abs - 2) % kprev;
        const uint32_t v1 = ((idx >> 6) * kprev + r1) * 64u + (idx & 63u), v2 = ((idx >> 6) * kprev + r2) * 64u + (idx & 63u);
        tb_load(p1.X, a.kgx, a.Gs + 2, v1, 1); tb_load(p1.Z, a.kgz, a.Gs + 2, v1, 1);
        tb_load(p2.X, a.kgx, a.Gs + 2, v2, 1); tb_load(p2.Z, a.kgz, a.Gs + 2, v2, 1);
    } else {
        tb_load(p2.X, a.gx, a.G + 2, idx, 0); tb_load(p2.Z, a.gz, a.G + 2, idx, 0);
        tb_load(p1.X, a.gx, a.G + 2, idx, 1); tb_load(p1.Z, a.gz, a.G + 2, idx, 1);
    }
    for (uint32_t i = start; i < n; i++) {            // Pa[i] = Pa[i-1] + Pd, difference Pa[i-2]  ecm.c:2412-2416
        Pt<NL> T;
        Fe<NL> s1, d1, pp, mm;
        pt_sumdiff(s1, d1, p1, m);
        pt_add_uv(pp, mm, s1, d1, sD, dD, m);
        fe_mul(T.X, pp, p2.Z, m);
        fe_mul(T.Z, mm, p2.X, m);
        tb_store(a.gx, a.G + 2, idx, i + 2, T.X);
        tb_store(a.gz, a.G + 2, idx, i + 2, T.Z);
        p2 = p1;
        p1 = T;
    }
    // the last two steps seed the next chunk
    tb_store(a.gx, a.G + 2, idx, 0, p2.X); tb_store(a.gz, a.G + 2, idx, 0, p2.Z);
    tb_store(a.gx, a.G + 2, idx, 1, p1.X); tb_store(a.gz, a.G + 2, idx, 1, p1.Z);
    // Montgomery's trick over the chunk (ecm.c:2003-2136), results into the ring
    Fe<NL> acc, z, x, t;
    tb_load(acc, a.gz, a.G + 2, idx, 2);
    tb_store(a.gp, a.G, idx, 0, acc);
    for (uint32_t i = 1; i < n; i++) {
        tb_load(z, a.gz, a.G + 2, idx, i + 2);
        fe_mul(acc, acc, z, m);
        tb_store(a.gp, a.G, idx, i, acc);
    }
    Fe<NL> inv;
    fe_inv_mont(inv, acc, k, a.fail, stride, idx);
    const uint32_t rmask = a.ring_size - 1;
    for (uint32_t i = n - 1; i > 0; i--) {
        tb_load(t, a.gp, a.G, idx, i - 1);
        fe_mul(t, t, inv, m);
        tb_load(z, a.gz, a.G + 2, idx, i + 2);
        fe_mul(inv, inv, z, m);
        tb_load(x, a.gx, a.G + 2, idx, i + 2);
        fe_mul(x, x, t, m);
        tb_store(a.ring, a.ring_size, idx, (first_abs + i) & rmask, x);
    }
    tb_load(x, a.gx, a.G + 2, idx, 2);
    fe_mul(x, x, inv, m);
    tb_store(a.ring, a.ring_size, idx, first_abs & rmask, x);
}

// Giant steps [first_abs, first_abs + n) with K sub-sequences per curve: sub-sequence r makes the steps s = r (mod K),
// P_s = P_(s-K) + [K*D]Q with difference P_(s-2K) (its first two members by the ladder), normalises ITS members with
// its own inversion and writes them to the ring.  Same points, same X/Z.  State between chunks: the latest two members
// in scratch entries 0, 1 of (curve block, r).  A failing inversion is recorded in plane 1 + r of `fail` (plane 0
// belongs to the single-chain chunks, whose last one reproduces the reference's last batch).
template <int NL>
__device__ __forceinline__ void giant_chunk_k(const S2PairArgs &a, uint32_t first_abs, uint32_t n, const S2Const<NL> &k,
                                              uint32_t idx, uint32_t r)
{
    const ModK<NL> &m = k.m;
    const size_t stride = a.stride;
    const uint32_t K = a.K;
    const uint32_t s0 = first_abs + (r + K - first_abs % K) % K;            // first step of this sub-sequence in the chunk
    if (s0 >= first_abs + n) return;                                        // wave-uniform
    const uint32_t cnt = (first_abs + n - 1 - s0) / K + 1;
    const uint32_t m0 = s0 / K;                                             // member number of s0 (s = r + member*K)
    const uint32_t v = ((idx >> 6) * K + r) * 64u + (idx & 63u);
    Pt<NL> Q, PdK, p1, p2;
    Fe<NL> s4, sD, dD;
    fe_load(PdK.X, a.PdKX, stride, idx);
    fe_load(PdK.Z, a.PdKZ, stride, idx);
    pt_sumdiff(sD, dD, PdK, m);
    if (m0 < 2) {
        fe_load(Q.X, a.X, stride, idx);
        fe_load(Q.Z, a.Z, stride, idx);
        fe_load(s4, a.S, stride, idx);
    }
    if (m0 >= 1) { tb_load(p1.X, a.kgx, a.Gs + 2, v, 1); tb_load(p1.Z, a.kgz, a.Gs + 2, v, 1); }
    if (m0 >= 2) { tb_load(p2.X, a.kgx, a.Gs + 2, v, 0); tb_load(p2.Z, a.kgz, a.Gs + 2, v, 0); }
    for (uint32_t j = 0; j < cnt; j++) {
        Pt<NL> T;
        if (m0 + j < 2) {
            T = Q;
            pt_ladder(T, a.A0 + (uint64_t)(s0 + j * K) * a.D, s4, m);       // [A + s*D]Q   ecm.c:2380-2383
        } else {
            Fe<NL> s1, d1, pp, mm;
            pt_sumdiff(s1, d1, p1, m);
            pt_add_uv(pp, mm, s1, d1, sD, dD, m);
            fe_mul(T.X, pp, p2.Z, m);
            fe_mul(T.Z, mm, p2.X, m);
        }
        tb_store(a.kgx, a.Gs + 2, v, j + 2, T.X);
        tb_store(a.kgz, a.Gs + 2, v, j + 2, T.Z);
        p2 = p1;
        p1 = T;
    }
    if (m0 + cnt >= 2) { tb_store(a.kgx, a.Gs + 2, v, 0, p2.X); tb_store(a.kgz, a.Gs + 2, v, 0, p2.Z); }
    tb_store(a.kgx, a.Gs + 2, v, 1, p1.X); tb_store(a.kgz, a.Gs + 2, v, 1, p1.Z);
    Fe<NL> acc, z, x, t;
    tb_load(acc, a.kgz, a.Gs + 2, v, 2);
    tb_store(a.kgp, a.Gs, v, 0, acc);
    for (uint32_t i = 1; i < cnt; i++) {
        tb_load(z, a.kgz, a.Gs + 2, v, i + 2);
        fe_mul(acc, acc, z, m);
        tb_store(a.kgp, a.Gs, v, i, acc);
    }
    Fe<NL> inv;
    fe_inv_mont(inv, acc, k, a.fail + (size_t)(1 + r) * NL * stride, stride, idx);
    const uint32_t rmask = a.ring_size - 1;
    for (uint32_t i = cnt - 1; i > 0; i--) {
        tb_load(t, a.kgp, a.Gs, v, i - 1);
        fe_mul(t, t, inv, m);
        tb_load(z, a.kgz, a.Gs + 2, v, i + 2);
        fe_mul(inv, inv, z, m);
        tb_load(x, a.kgx, a.Gs + 2, v, i + 2);
        fe_mul(x, x, t, m);
        tb_store(a.ring, a.ring_size, idx, (s0 + i * K) & rmask, x);
    }
    tb_load(x, a.kgx, a.Gs + 2, v, 2);
    fe_mul(x, x, inv, m);
    tb_store(a.ring, a.ring_size, idx, s0 & rmask, x);
}

// The pair walk of ecm_stage2_pair (ecm.c:2448-2533) over tape entries [first, first+count): every
// entry is a pair (ring slot, table index); "generate" marks are separate launches of giant_chunk.
// The product over the pairs is order-independent, so a segment may be cut into slices walked by
// different wavefronts into separate accumulators (accbuf = this slice's accumulator, [limb][curve]);
// s2_merge multiplies them together.  Each slice starts from `one` = R mod N, so the merged Montgomery
// product one * prod(d_i) * R^-n is the same residue as the reference's single running accumulator.
#ifndef GECM_S2_DEPTH
#define GECM_S2_DEPTH 4
#endif
#ifndef GECM_S2_DEPTH_MAXNL
#define GECM_S2_DEPTH_MAXNL 23            // above: two rows in flight (registers)
#endif
template <int NL>
__device__ __forceinline__ void s2_pairs(const S2PairArgs &a, uint32_t first, uint32_t count, const S2Const<NL> &k,
                                         uint32_t idx, uint32_t *__restrict__ accbuf)
{
    const ModK<NL> &m = k.m;
    const size_t stride = a.stride;
    if (count == 0) return;
    Fe<NL> acc;
    fe_load(acc, accbuf, stride, idx);
    // Lookahead: the baby-step rows of the next DEPTH pairs are in flight while the current pair is
    // multiplied, so HBM latency (2-3 us under load) hides behind the arithmetic (tb_load_async / tb_wait_cnt).  The host sorts the
    // pairs of a segment by giant step (the product is order-independent), so the ring row is
    // re-read only when the giant step changes: per pair one 4*NL-byte row of the 461-KB-per-curve
    // baby-step table comes from HBM instead of two rows (the walk is HBM-bound otherwise:
    // 2*60 B x 131,072 curves x 3.0 M pairs = 47 TB at B2 = 1e8).
    const uint32_t *st = a.steps + 2 * (size_t)first;
    constexpr int DEPTH = NL <= GECM_S2_DEPTH_MAXNL ? GECM_S2_DEPTH : 2;     // table rows in flight per lane
    static_assert(64 % DEPTH == 0, "a block of 64 tape entries is walked in groups of DEPTH");
    Fe<NL> x, yq[DEPTH];
    // The tape is read 64 entries at a time, one entry per lane (coalesced), and handed out with v_readlane: `cs` =
    // giant-step slots of this block's pairs, `ca` = table indices of the pairs DEPTH ahead of them (the rows to
    // request next).  The words of the next block (ns, na) are requested at the top of a block and first touched at the
    // top of the next one, behind an empty asm, so that the compiler's wait for them — a vmcnt(0): it does not count
    // the asm loads — sits there, once per 64 pairs, and nowhere inside the walk.  Entries past the segment repeat
    // its last one.
    const uint32_t lane = idx & 63u;
    auto ld = [&](uint32_t j, uint32_t w) -> uint32_t {
        j = j < count ? j : count - 1;
        return st[2 * (size_t)j + w];
    };
    auto rl = [](uint32_t v, uint32_t l) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); };
    uint32_t ns = ld(lane, 0), na = ld(lane + DEPTH, 1), p0 = ld(lane, 1);
    tb_tie<NL, 0, NL>(acc);                               // the accumulator has arrived before the first request
    asm volatile("" : "+v"(ns), "+v"(na), "+v"(p0));
    uint32_t slot = rl(ns, 0);
    tb_load_async(x, a.ring, a.ring_size, idx, slot);
    tb_wait<NL, 0>(x);
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if ((uint32_t)d < count) tb_load_async(yq[d], a.PbX, a.npb, idx, rl(p0, (uint32_t)d));
    for (uint32_t b0 = 0; b0 < count; b0 += 64) {
        uint32_t cs = ns, ca = na;
        asm volatile("" : "+v"(cs), "+v"(ca));
        ns = ld(b0 + 64 + lane, 0);
        na = ld(b0 + 64 + lane + DEPTH, 1);
        for (uint32_t jj = 0; jj < 64; jj += DEPTH) {
            if (b0 + jj >= count) break;
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                const uint32_t l = jj + (uint32_t)d;
                const uint32_t j = b0 + l;
                if (j < count) {
                    const uint32_t sl = rl(cs, l);
                    if (sl != slot) {                     // next giant step (pairs are sorted by it)
                        slot = sl;
                        tb_load_async(x, a.ring, a.ring_size, idx, slot);   // once per giant step: waited for in full
                        tb_wait<NL, 0>(x);
                    }
                    // rows of the pairs j+1 .. j+DEPTH-1 that exist were requested after this one
                    const uint32_t newer = count - 1 - j;
                    if (newer >= (uint32_t)(DEPTH - 1)) tb_wait_cnt<NL, DEPTH - 1>();
                    else if (newer == 2) tb_wait_cnt<NL, (DEPTH > 2 ? 2 : DEPTH - 1)>();
                    else if (newer == 1) tb_wait_cnt<NL, 1>();
                    else tb_wait_cnt<NL, 0>();
                    tb_tie<NL, 0, NL>(yq[d]);
                    Fe<NL> t;
                    fe_sub(t, x, yq[d], m);               // CROSS_PRODUCT_INV  ecm.c:1857-1859
                    fe_mul(acc, acc, t, m);
                    if (j + DEPTH < count) tb_load_async(yq[d], a.PbX, a.npb, idx, rl(ca, l));
                }
            }
        }
    }
    Fe<NL> c;
    fe_canonical_mont(c, acc, k.one, m);
    fe_store(accbuf, stride, idx, c);
}

// acc[0] <- product of the `slices` accumulators (skipped when init_only); acc[1..] <- one.
template <int NL>
__device__ __forceinline__ void s2_merge(uint32_t *__restrict__ acc, uint32_t slices, size_t stride, bool init_only,
                                         const S2Const<NL> &k, uint32_t idx)
{
    const size_t slice_words = (size_t)NL * stride;
    Fe<NL> r;
    if (!init_only) fe_load(r, acc, stride, idx);
    for (uint32_t p = 1; p < slices; p++) {
        uint32_t *sp = acc + p * slice_words;
        if (!init_only) {
            Fe<NL> t;
            fe_load(t, sp, stride, idx);
            fe_mul(r, r, t, k.m);
        }
        fe_store(sp, stride, idx, k.one);
    }
    if (!init_only) {
        Fe<NL> c;
        fe_canonical_mont(c, r, k.one, k.m);
        fe_store(acc, stride, idx, c);
    }
}
