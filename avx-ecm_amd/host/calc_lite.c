/* calc_lite.c — input-expression evaluator for the avx-ecm command line.
 *
 * The reference accepts its first argument as an expression (calc.c:683-1104, README.md:30), e.g.
 * "fib(791)/13/677/216416017".  This is a small recursive-descent evaluator over the mpl integers
 * covering the operators a factoring command line uses:
 *     + - * / % ^      binary, usual precedence, ^ right-associative, / exact or truncating
 *     n!  n#           factorial and primorial (postfix)
 *     << >>            shifts (lowest precedence, as in the reference's table calc.c:1106-1126)
 *     fib(n) luc(n)    Fibonacci and Lucas numbers
 *     gcd(a,b) sqrt(a) nroot(a,n) modinv(a,m) modexp(a,e,m) lg2(a) log(a) abs(a) xor(a,b) and(a,b) or(a,b)
 *                      (lg2 / log = mpz_sizeinbase(a, 2) / (a, 10), calc.c:1250-1259)
 *     ( )              grouping; decimal or 0x-hex literals
 * Values are non-negative and bounded by the mpl capacity (4352 bits); a subtraction that would go
 * negative or an overflow is an error (return code != 0), never a wrong number.
 */
#include "calc_lite.h"
#include <ctype.h>
#include <string.h>

typedef struct {
    const char *s;
    int err;
} P;

static void skip(P *p) { while (*p->s == ' ' || *p->s == '\t') p->s++; }
static void expr(P *p, mpl_t *r);

static int fits(const mpl_t *a, const mpl_t *b) { return a->n + b->n <= MPL_MAXL - 2; }

static void fibluc(P *p, mpl_t *r, uint64_t n, int lucas)
{
    mpl_t a, b, t;
    mpl_set_u64(&a, lucas ? 2 : 0);
    mpl_set_u64(&b, 1);
    for (uint64_t i = 0; i < n; i++) {
        if (b.n >= MPL_MAXL - 2) { p->err = 1; return; }
        mpl_add(&t, &a, &b);
        a = b;
        b = t;
    }
    *r = a;
}

static void primary(P *p, mpl_t *r)
{
    skip(p);
    if (p->err) return;
    if (*p->s == '(') {
        p->s++;
        expr(p, r);
        skip(p);
        if (*p->s != ')') { p->err = 1; return; }
        p->s++;
    } else if (isalpha((unsigned char)*p->s)) {
        char name[8];
        int k = 0;
        while (isalnum((unsigned char)*p->s) && k < 7) name[k++] = (char)tolower((unsigned char)*p->s++);   /* lg2 */
        name[k] = 0;
        skip(p);
        if (*p->s != '(') { p->err = 1; return; }
        p->s++;
        mpl_t arg[3];
        int na = 0;
        for (;;) {
            if (na == 3) { p->err = 1; return; }
            expr(p, &arg[na++]);
            skip(p);
            if (p->err) return;
            if (*p->s == ',') { p->s++; continue; }
            break;
        }
        if (*p->s != ')') { p->err = 1; return; }
        p->s++;
        if ((!strcmp(name, "fib") || !strcmp(name, "luc")) && na == 1) {
            if (arg[0].n > 1 || mpl_get_u64(&arg[0]) > 100000) { p->err = 1; return; }
            fibluc(p, r, mpl_get_u64(&arg[0]), name[0] == 'l');
        } else if (!strcmp(name, "gcd") && na == 2) {
            mpl_gcd(r, &arg[0], &arg[1]);
        } else if (!strcmp(name, "modinv") && na == 2) {
            if (mpl_is_zero(&arg[1]) || !mpl_invmod(r, &arg[0], &arg[1])) p->err = 1;
        } else if (!strcmp(name, "modexp") && na == 3) {
            if (mpl_is_zero(&arg[2])) p->err = 1;
            else mpl_powmod(r, &arg[0], &arg[1], &arg[2]);
        } else if ((!strcmp(name, "sqrt") && na == 1) || (!strcmp(name, "nroot") && na == 2)) {
            /* integer k-th root by bisection on the bit length (mpz_sqrt / mpz_root, calc.c:1306, 1322) */
            uint64_t kth = 2;
            if (na == 2) {
                if (arg[1].n > 1 || mpl_get_u64(&arg[1]) < 1 || mpl_get_u64(&arg[1]) > 4096) { p->err = 1; return; }
                kth = mpl_get_u64(&arg[1]);
            }
            mpl_t lo, hi, mid, pw, one;
            mpl_set_u64(&lo, 0);
            mpl_set_u64(&one, 1);
            mpl_shl(&hi, &one, (unsigned)(mpl_bits(&arg[0]) / (int)kth + 1));
            while (mpl_cmp(&lo, &hi) < 0) {           /* invariant: lo^k <= a < (hi+1)^k */
                mpl_add(&mid, &lo, &hi);
                mpl_add(&mid, &mid, &one);
                mpl_shr(&mid, &mid, 1);
                int over = 0;
                pw = one;
                for (uint64_t e = 0; e < kth && !over; e++) {
                    if (!fits(&pw, &mid)) { over = 1; break; }
                    mpl_mul(&pw, &pw, &mid);
                    if (mpl_cmp(&pw, &arg[0]) > 0) over = 1;
                }
                if (!over) lo = mid;
                else mpl_sub(&hi, &mid, &one);
            }
            *r = lo;
        } else if (!strcmp(name, "lg2") && na == 1) {
            mpl_set_u64(r, mpl_is_zero(&arg[0]) ? 1 : (uint64_t)mpl_bits(&arg[0]));
        } else if (!strcmp(name, "log") && na == 1) {
            mpl_set_u64(r, (uint64_t)mpl_sizeinbase10(&arg[0]));
        } else if (!strcmp(name, "abs") && na == 1) {
            *r = arg[0];
        } else if ((!strcmp(name, "xor") || !strcmp(name, "and") || !strcmp(name, "or")) && na == 2) {
            mpl_t o;
            const int n = arg[0].n > arg[1].n ? arg[0].n : arg[1].n;
            for (int i = 0; i < n; i++) {
                const uint32_t x = i < arg[0].n ? arg[0].d[i] : 0, y = i < arg[1].n ? arg[1].d[i] : 0;
                o.d[i] = name[0] == 'x' ? (x ^ y) : name[0] == 'a' ? (x & y) : (x | y);
            }
            o.n = n;
            while (o.n > 0 && o.d[o.n - 1] == 0) o.n--;
            *r = o;
        } else {
            p->err = 1;
        }
    } else if (isdigit((unsigned char)*p->s)) {
        char buf[2048];
        int k = 0;
        if (p->s[0] == '0' && (p->s[1] == 'x' || p->s[1] == 'X')) {
            buf[k++] = *p->s++;
            buf[k++] = *p->s++;
            while (isxdigit((unsigned char)*p->s) && k < 2040) buf[k++] = *p->s++;
        } else {
            while (isdigit((unsigned char)*p->s) && k < 2040) buf[k++] = *p->s++;
        }
        buf[k] = 0;
        if (mpl_set_str(r, buf)) p->err = 1;
    } else {
        p->err = 1;
    }
    /* postfix ! and # */
    for (;;) {
        skip(p);
        if (p->err) return;
        if (*p->s == '!' || *p->s == '#') {
            int prim = *p->s == '#';
            p->s++;
            if (r->n > 1 || mpl_get_u64(r) > 100000) { p->err = 1; return; }
            uint64_t n = mpl_get_u64(r);
            mpl_t acc;
            mpl_set_u64(&acc, 1);
            for (uint64_t i = 2; i <= n; i++) {
                if (prim) {
                    int isp = 1;
                    for (uint64_t d = 2; d * d <= i; d++)
                        if (i % d == 0) { isp = 0; break; }
                    if (!isp) continue;
                }
                if (acc.n >= MPL_MAXL - 3) { p->err = 1; return; }
                mpl_mul_u64(&acc, &acc, i);
            }
            *r = acc;
        } else {
            return;
        }
    }
}

static void power(P *p, mpl_t *r)
{
    primary(p, r);
    skip(p);
    if (!p->err && *p->s == '^') {
        p->s++;
        mpl_t e, base = *r, acc;
        power(p, &e);                       /* right-associative */
        if (p->err || e.n > 1 || mpl_get_u64(&e) > 100000) { p->err = 1; return; }
        uint64_t n = mpl_get_u64(&e);
        mpl_set_u64(&acc, 1);
        for (uint64_t i = 0; i < n; i++) {
            if (!fits(&acc, &base)) { p->err = 1; return; }
            mpl_mul(&acc, &acc, &base);
        }
        *r = acc;
    }
}

static void term(P *p, mpl_t *r)
{
    power(p, r);
    for (;;) {
        skip(p);
        if (p->err) return;
        char op = *p->s;
        if (op != '*' && op != '/' && op != '%') return;
        p->s++;
        mpl_t b, q, rem;
        power(p, &b);
        if (p->err) return;
        if (op == '*') {
            if (!fits(r, &b)) { p->err = 1; return; }
            mpl_mul(r, r, &b);
        } else {
            if (mpl_is_zero(&b)) { p->err = 1; return; }
            mpl_divrem(&q, &rem, r, &b);
            *r = op == '/' ? q : rem;
        }
    }
}

static void sum(P *p, mpl_t *r)
{
    term(p, r);
    for (;;) {
        skip(p);
        if (p->err) return;
        char op = *p->s;
        if (op != '+' && op != '-') return;
        p->s++;
        mpl_t b;
        term(p, &b);
        if (p->err) return;
        if (op == '+') mpl_add(r, r, &b);
        else {
            if (mpl_cmp(r, &b) < 0) { p->err = 1; return; }
            mpl_sub(r, r, &b);
        }
    }
}

static void expr(P *p, mpl_t *r)
{
    sum(p, r);
    for (;;) {
        skip(p);
        if (p->err) return;
        if (!((p->s[0] == '<' && p->s[1] == '<') || (p->s[0] == '>' && p->s[1] == '>'))) return;
        int left = p->s[0] == '<';
        p->s += 2;
        mpl_t b;
        sum(p, &b);
        if (p->err || b.n > 1 || mpl_get_u64(&b) > 4000) { p->err = 1; return; }
        unsigned k = (unsigned)mpl_get_u64(&b);
        if (left) {
            if (mpl_bits(r) + (int)k > MPL_MAXL * 32 - 64) { p->err = 1; return; }
            mpl_shl(r, r, k);
        } else {
            mpl_shr(r, r, k);
        }
    }
}

int calc_lite(mpl_t *out, const char *s)
{
    P p = {s, 0};
    expr(&p, out);
    skip(&p);
    if (p.err || *p.s) return -1;
    return 0;
}

/* the hash of the host sources this object was compiled from (Makefile: H_SHA); gecm_version() compares them */
#ifdef GECM_MANIFEST_FN
const char *GECM_MANIFEST_FN(void) { return GECM_MANIFEST; }
#endif
