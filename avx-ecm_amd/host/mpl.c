/* mpl.c — see mpl.h.  Schoolbook algorithms on 32-bit limbs; sizes here are <= ~2200 bits and
 * the calls are off the hot path (a few per curve), so clarity wins over speed. */
#include "mpl.h"
#include <string.h>

static void norm(mpl_t *a)
{
    while (a->n > 0 && a->d[a->n - 1] == 0) a->n--;
}

void mpl_set_u64(mpl_t *r, uint64_t v)
{
    r->n = 0;
    if (v) r->d[r->n++] = (uint32_t)v;
    if (v >> 32) r->d[r->n++] = (uint32_t)(v >> 32);
}

uint64_t mpl_get_u64(const mpl_t *a)
{
    uint64_t v = 0;
    if (a->n > 0) v = a->d[0];
    if (a->n > 1) v |= (uint64_t)a->d[1] << 32;
    return v;
}

int mpl_is_zero(const mpl_t *a) { return a->n == 0; }
int mpl_is_odd(const mpl_t *a) { return a->n > 0 && (a->d[0] & 1); }

int mpl_bits(const mpl_t *a)
{
    if (a->n == 0) return 0;
    uint32_t t = a->d[a->n - 1];
    int b = 0;
    while (t) { b++; t >>= 1; }
    return (a->n - 1) * 32 + b;
}

int mpl_cmp(const mpl_t *a, const mpl_t *b)
{
    if (a->n != b->n) return a->n < b->n ? -1 : 1;
    for (int i = a->n - 1; i >= 0; i--)
        if (a->d[i] != b->d[i]) return a->d[i] < b->d[i] ? -1 : 1;
    return 0;
}

int mpl_cmp_u64(const mpl_t *a, uint64_t v)
{
    mpl_t t;
    mpl_set_u64(&t, v);
    return mpl_cmp(a, &t);
}

void mpl_add(mpl_t *r, const mpl_t *a, const mpl_t *b)
{
    const mpl_t *x = a->n >= b->n ? a : b, *y = a->n >= b->n ? b : a;
    uint64_t c = 0;
    int i, xn = x->n, yn = y->n;
    for (i = 0; i < yn; i++) { c += (uint64_t)x->d[i] + y->d[i]; r->d[i] = (uint32_t)c; c >>= 32; }
    for (; i < xn; i++) { c += x->d[i]; r->d[i] = (uint32_t)c; c >>= 32; }
    r->n = xn;
    if (c && r->n < MPL_MAXL) r->d[r->n++] = (uint32_t)c;
}

void mpl_add_u64(mpl_t *r, const mpl_t *a, uint64_t v)
{
    mpl_t t;
    mpl_set_u64(&t, v);
    mpl_add(r, a, &t);
}

void mpl_sub(mpl_t *r, const mpl_t *a, const mpl_t *b)
{
    int64_t c = 0;
    int i, an = a->n, bn = b->n;
    for (i = 0; i < bn; i++) { c += (int64_t)a->d[i] - b->d[i]; r->d[i] = (uint32_t)c; c >>= 32; }
    for (; i < an; i++) { c += a->d[i]; r->d[i] = (uint32_t)c; c >>= 32; }
    r->n = an;
    norm(r);
}

void mpl_mul(mpl_t *r, const mpl_t *a, const mpl_t *b)
{
    mpl_t t;
    int an = a->n, bn = b->n;
    if (an == 0 || bn == 0) { r->n = 0; return; }
    int rn = an + bn;
    if (rn > MPL_MAXL) rn = MPL_MAXL;
    memset(t.d, 0, (size_t)rn * 4);
    for (int i = 0; i < an; i++) {
        uint64_t c = 0, ai = a->d[i];
        int j;
        for (j = 0; j < bn && i + j < rn; j++) {
            c += ai * b->d[j] + t.d[i + j];
            t.d[i + j] = (uint32_t)c;
            c >>= 32;
        }
        if (i + j < rn) t.d[i + j] = (uint32_t)c;
    }
    t.n = rn;
    norm(&t);
    *r = t;
}

void mpl_mul_u64(mpl_t *r, const mpl_t *a, uint64_t v)
{
    mpl_t t;
    mpl_set_u64(&t, v);
    mpl_mul(r, a, &t);
}

void mpl_shl(mpl_t *r, const mpl_t *a, unsigned bits)
{
    mpl_t t;
    int ws = (int)(bits / 32), bs = (int)(bits % 32);
    if (a->n == 0) { r->n = 0; return; }
    int n = a->n + ws + 1;
    if (n > MPL_MAXL) n = MPL_MAXL;
    memset(t.d, 0, (size_t)n * 4);
    for (int i = 0; i < a->n; i++) {
        uint64_t v = (uint64_t)a->d[i] << bs;
        if (i + ws < n) t.d[i + ws] |= (uint32_t)v;
        if (i + ws + 1 < n) t.d[i + ws + 1] |= (uint32_t)(v >> 32);
    }
    t.n = n;
    norm(&t);
    *r = t;
}

void mpl_shr(mpl_t *r, const mpl_t *a, unsigned bits)
{
    mpl_t t;
    int ws = (int)(bits / 32), bs = (int)(bits % 32);
    if (ws >= a->n) { r->n = 0; return; }
    int n = a->n - ws;
    for (int i = 0; i < n; i++) {
        uint64_t v = a->d[i + ws];
        if (i + ws + 1 < a->n) v |= (uint64_t)a->d[i + ws + 1] << 32;
        t.d[i] = (uint32_t)(v >> bs);
    }
    t.n = n;
    norm(&t);
    *r = t;
}

/* Knuth algorithm D */
void mpl_divrem(mpl_t *q, mpl_t *r, const mpl_t *a, const mpl_t *b)
{
    mpl_t qq, u, v;
    if (b->n == 0) { if (q) q->n = 0; if (r) r->n = 0; return; }
    if (mpl_cmp(a, b) < 0) { mpl_t t = *a; if (q) q->n = 0; if (r) *r = t; return; }
    if (b->n == 1) {
        uint64_t rem = 0, d = b->d[0];
        for (int i = a->n - 1; i >= 0; i--) {
            uint64_t cur = (rem << 32) | a->d[i];
            qq.d[i] = (uint32_t)(cur / d);
            rem = cur % d;
        }
        qq.n = a->n;
        norm(&qq);
        if (q) *q = qq;
        if (r) mpl_set_u64(r, rem);
        return;
    }
    int s = 0;
    { uint32_t t = b->d[b->n - 1]; while (!(t & 0x80000000u)) { t <<= 1; s++; } }
    mpl_shl(&v, b, (unsigned)s);
    mpl_shl(&u, a, (unsigned)s);
    int n = v.n, m = a->n - b->n;
    /* u needs a->n + 1 limbs */
    for (int i = u.n; i <= a->n; i++) u.d[i] = 0;
    u.n = a->n + 1;
    memset(qq.d, 0, (size_t)(m + 1) * 4);
    for (int j = m; j >= 0; j--) {
        uint64_t num = ((uint64_t)u.d[j + n] << 32) | u.d[j + n - 1];
        uint64_t qhat = num / v.d[n - 1], rhat = num % v.d[n - 1];
        while (qhat >= 0x100000000ull || qhat * v.d[n - 2] > ((rhat << 32) | u.d[j + n - 2])) {
            qhat--;
            rhat += v.d[n - 1];
            if (rhat >= 0x100000000ull) break;
        }
        int64_t borrow = 0;
        uint64_t carry = 0;
        for (int i = 0; i < n; i++) {
            uint64_t p = qhat * v.d[i] + carry;
            carry = p >> 32;
            int64_t t = (int64_t)u.d[i + j] - (int64_t)(uint32_t)p + borrow;
            u.d[i + j] = (uint32_t)t;
            borrow = t >> 32;
        }
        int64_t t = (int64_t)u.d[j + n] - (int64_t)carry + borrow;
        u.d[j + n] = (uint32_t)t;
        if (t < 0) {
            qhat--;
            uint64_t c = 0;
            for (int i = 0; i < n; i++) {
                c += (uint64_t)u.d[i + j] + v.d[i];
                u.d[i + j] = (uint32_t)c;
                c >>= 32;
            }
            u.d[j + n] += (uint32_t)c;
        }
        qq.d[j] = (uint32_t)qhat;
    }
    qq.n = m + 1;
    norm(&qq);
    if (q) *q = qq;
    if (r) {
        u.n = n;
        norm(&u);
        mpl_shr(r, &u, (unsigned)s);
    }
}

void mpl_mod(mpl_t *r, const mpl_t *a, const mpl_t *m) { mpl_divrem(NULL, r, a, m); }

void mpl_mulmod(mpl_t *r, const mpl_t *a, const mpl_t *b, const mpl_t *m)
{
    mpl_t t;
    mpl_mul(&t, a, b);
    mpl_divrem(NULL, r, &t, m);
}

void mpl_addmod(mpl_t *r, const mpl_t *a, const mpl_t *b, const mpl_t *m)
{
    mpl_t t;
    mpl_add(&t, a, b);
    if (mpl_cmp(&t, m) >= 0) mpl_sub(&t, &t, m);
    *r = t;
}

void mpl_submod(mpl_t *r, const mpl_t *a, const mpl_t *b, const mpl_t *m)
{
    mpl_t t;
    if (mpl_cmp(a, b) >= 0) mpl_sub(&t, a, b);
    else { mpl_add(&t, a, m); mpl_sub(&t, &t, b); }
    *r = t;
}

void mpl_powmod(mpl_t *r, const mpl_t *a, const mpl_t *e, const mpl_t *m)
{
    mpl_t base, acc;
    mpl_mod(&base, a, m);
    mpl_set_u64(&acc, 1);
    mpl_mod(&acc, &acc, m);
    int nb = mpl_bits(e);
    for (int i = nb - 1; i >= 0; i--) {
        mpl_mulmod(&acc, &acc, &acc, m);
        if ((e->d[i / 32] >> (i % 32)) & 1) mpl_mulmod(&acc, &acc, &base, m);
    }
    *r = acc;
}

void mpl_gcd(mpl_t *r, const mpl_t *a, const mpl_t *b)
{
    mpl_t x = *a, y = *b, t;
    while (y.n) {
        mpl_divrem(NULL, &t, &x, &y);
        x = y;
        y = t;
    }
    *r = x;
}

int mpl_invmod(mpl_t *r, const mpl_t *a, const mpl_t *m)
{
    /* extended Euclid on (m, a mod m) tracking only the coefficient of a, with explicit signs */
    mpl_t r0 = *m, r1, t0, t1, q, tmp, prod;
    int s0 = 0, s1 = 0; /* signs of t0, t1 (1 = negative) */
    mpl_mod(&r1, a, m);
    t0.n = 0;
    mpl_set_u64(&t1, 1);
    if (mpl_cmp_u64(m, 1) == 0) { r->n = 0; return 1; }
    while (r1.n) {
        mpl_divrem(&q, &tmp, &r0, &r1);
        r0 = r1;
        r1 = tmp;
        /* t2 = t0 - q*t1 */
        mpl_mul(&prod, &q, &t1);
        mpl_t t2;
        int s2;
        if (s0 != s1) { mpl_add(&t2, &t0, &prod); s2 = s0; }
        else if (mpl_cmp(&t0, &prod) >= 0) { mpl_sub(&t2, &t0, &prod); s2 = s0; }
        else { mpl_sub(&t2, &prod, &t0); s2 = !s0; }
        t0 = t1; s0 = s1;
        t1 = t2; s1 = s2;
    }
    if (mpl_cmp_u64(&r0, 1) != 0) return 0;
    mpl_mod(&t0, &t0, m);
    if (s0 && t0.n) mpl_sub(&t0, m, &t0);
    *r = t0;
    return 1;
}

int mpl_probab_prime(const mpl_t *a, int reps)
{
    static const uint32_t small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71,
                                     73, 79, 83, 89, 97, 101, 103, 107, 109, 113, 127, 131, 137, 139, 149, 151};
    const int ns = (int)(sizeof small / sizeof small[0]);
    if (a->n == 0 || mpl_cmp_u64(a, 1) == 0) return 0;
    mpl_t t, rem;
    for (int i = 0; i < ns; i++) {
        if (mpl_cmp_u64(a, small[i]) == 0) return 1;
        mpl_set_u64(&t, small[i]);
        mpl_divrem(NULL, &rem, a, &t);
        if (rem.n == 0) return 0;
    }
    mpl_t one, nm1, d, x, base;
    mpl_set_u64(&one, 1);
    mpl_sub(&nm1, a, &one);
    int s = 0;
    d = nm1;
    while (!mpl_is_odd(&d)) { mpl_shr(&d, &d, 1); s++; }
    if (reps < 1) reps = 1;
    if (reps > ns) reps = ns;
    for (int i = 0; i < reps + 9; i++) {     /* a few more bases than asked: still cheap */
        if (i >= ns) break;
        mpl_set_u64(&base, small[i]);
        mpl_powmod(&x, &base, &d, a);
        if (mpl_cmp(&x, &one) == 0 || mpl_cmp(&x, &nm1) == 0) continue;
        int comp = 1;
        for (int r = 1; r < s; r++) {
            mpl_mulmod(&x, &x, &x, a);
            if (mpl_cmp(&x, &nm1) == 0) { comp = 0; break; }
        }
        if (comp) return 0;
    }
    return 1;
}

int mpl_set_str(mpl_t *r, const char *s)
{
    mpl_t acc;
    acc.n = 0;
    while (*s == ' ' || *s == '\t' || *s == '\n') s++;
    int hex = 0;
    if (s[0] == '0' && (s[1] == 'x' || s[1] == 'X')) { hex = 1; s += 2; }
    if (!*s) return -1;
    for (; *s; s++) {
        int v;
        char c = *s;
        if (c >= '0' && c <= '9') v = c - '0';
        else if (hex && c >= 'a' && c <= 'f') v = c - 'a' + 10;
        else if (hex && c >= 'A' && c <= 'F') v = c - 'A' + 10;
        else if (c == ' ' || c == '\n' || c == '\r' || c == '\t') continue;
        else return -1;
        if (acc.n >= MPL_MAXL - 1) return -2;
        mpl_mul_u64(&acc, &acc, hex ? 16 : 10);
        mpl_add_u64(&acc, &acc, (uint64_t)v);
    }
    *r = acc;
    return 0;
}

int mpl_get_hex(char *buf, const mpl_t *a)
{
    static const char H[] = "0123456789abcdef";
    if (a->n == 0) { buf[0] = '0'; buf[1] = 0; return 1; }
    int k = 0, started = 0;
    for (int i = a->n - 1; i >= 0; i--)
        for (int sft = 28; sft >= 0; sft -= 4) {
            int v = (a->d[i] >> sft) & 15;
            if (v || started) { buf[k++] = H[v]; started = 1; }
        }
    buf[k] = 0;
    return k;
}

int mpl_get_dec(char *buf, const mpl_t *a)
{
    if (a->n == 0) { buf[0] = '0'; buf[1] = 0; return 1; }
    mpl_t t = *a, q, r, ten9;
    char tmp[MPL_MAXL * 10 + 16];
    int k = 0;
    mpl_set_u64(&ten9, 1000000000u);
    while (t.n) {
        mpl_divrem(&q, &r, &t, &ten9);
        uint32_t v = r.n ? r.d[0] : 0;
        for (int i = 0; i < 9; i++) { tmp[k++] = (char)('0' + v % 10); v /= 10; }
        t = q;
    }
    while (k > 1 && tmp[k - 1] == '0') k--;
    for (int i = 0; i < k; i++) buf[i] = tmp[k - 1 - i];
    buf[k] = 0;
    return k;
}

static uint32_t getbits(const mpl_t *a, int pos, int bits)
{
    /* bits <= 32 */
    uint64_t v = 0;
    int w = pos / 32, o = pos % 32;
    if (w < a->n) v = a->d[w];
    if (w + 1 < a->n) v |= (uint64_t)a->d[w + 1] << 32;
    v >>= o;
    return (uint32_t)(bits >= 32 ? v : (v & ((1ull << bits) - 1)));
}

void mpl_to_limbs32(uint32_t *out, size_t stride, int count, int bits, const mpl_t *a)
{
    for (int i = 0; i < count; i++) out[(size_t)i * stride] = getbits(a, i * bits, bits);
}

void mpl_to_limbs64(uint64_t *out, size_t stride, int count, int bits, const mpl_t *a)
{
    for (int i = 0; i < count; i++) {
        int pos = i * bits;
        uint64_t lo = getbits(a, pos, bits > 32 ? 32 : bits);
        uint64_t hi = bits > 32 ? getbits(a, pos + 32, bits - 32) : 0;
        out[(size_t)i * stride] = lo | (hi << 32);
    }
}

static void orbits(mpl_t *r, int pos, uint64_t v)
{
    /* add v << pos into r (r wide enough, limbs pre-zeroed); handles overlapping limbs by addition */
    int w = pos / 32, o = pos % 32;
    uint64_t lo = v << o, hi = o ? (v >> (64 - o)) : 0;
    uint64_t c = 0;
    uint64_t parts[3] = {lo & 0xffffffffu, lo >> 32, hi};
    for (int k = 0; k < 3 || c; k++) {
        if (w + k >= MPL_MAXL) break;
        c += (uint64_t)r->d[w + k] + (k < 3 ? parts[k] : 0);
        r->d[w + k] = (uint32_t)c;
        c >>= 32;
    }
}

void mpl_from_limbs32(mpl_t *r, const uint32_t *in, size_t stride, int count, int bits)
{
    mpl_t t;
    memset(t.d, 0, sizeof t.d);
    for (int i = 0; i < count; i++) orbits(&t, i * bits, in[(size_t)i * stride]);
    t.n = MPL_MAXL;
    norm(&t);
    *r = t;
}

void mpl_from_limbs64(mpl_t *r, const uint64_t *in, size_t stride, int count, int bits)
{
    mpl_t t;
    memset(t.d, 0, sizeof t.d);
    for (int i = 0; i < count; i++) orbits(&t, i * bits, in[(size_t)i * stride]);
    t.n = MPL_MAXL;
    norm(&t);
    *r = t;
}

/* mpz_sizeinbase(a, 10) of GMP 6.x (MPN_SIZEINBASE with mp_bases[10].logb2 = floor(2^64 log10 2)):
 * the high word of (logb2 + 1) * bits, plus one.  The reference prints factor sizes with it
 * (ecm.c:1346, 1494; main.c:455), so "C13" can label a 12-digit factor. */
int mpl_sizeinbase10(const mpl_t *a)
{
    if (mpl_is_zero(a)) return 1;
    const unsigned __int128 p = (unsigned __int128)(0x4d104d427de7fbccULL + 1) * (unsigned)mpl_bits(a);
    return (int)(uint64_t)(p >> 64) + 1;
}

/* the hash of the host sources this object was compiled from (Makefile: H_SHA); gecm_version() compares them */
#ifdef GECM_MANIFEST_FN
const char *GECM_MANIFEST_FN(void) { return GECM_MANIFEST; }
#endif
