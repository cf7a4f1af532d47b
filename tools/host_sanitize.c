/* tools/host_sanitize.c — the host-side logic of libgecm (everything that runs without a device) under
 * AddressSanitizer + UBSan.  Built and run by tools/host_sanitize.sh on the CPU; GPU sanitizers are not
 * available on the pool.  Exercises: expression evaluator and input preparation with random and
 * hostile strings, mpl arithmetic identities, the stage-1 tape compiler, the stage-2 plan and PAIR. */
#include "../avx-ecm_amd/host/calc_lite.h"
#include "../avx-ecm_amd/host/cunningham.h"
#include "../avx-ecm_amd/host/gecm_pair.h"
#include "../avx-ecm_amd/host/gecm_plan.h"
#include "../avx-ecm_amd/host/mpl.h"
#include "../include/gecm.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t rng = 88172645463325252ull;
static uint64_t rnd(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }

int main(void)
{
    static char ndec[2000], log[65536], buf[4096];
    gecm_input_info inf;
    /* 1. hostile and random expressions */
    const char *alphabet = "0123456789+-*/%^!#()<>, xfibluc";
    int ok = 0;
    for (int it = 0; it < 20000; it++) {
        int len = (int)(rnd() % 40);
        for (int i = 0; i < len; i++) buf[i] = alphabet[rnd() % strlen(alphabet)];
        buf[len] = 0;
        if (gecm_prepare_input(buf, it & 1 ? 52 : 32, ndec, sizeof ndec, &inf, log, sizeof log) == 0) ok++;
    }
    const char *fixed[] = {"2^4000", "2^4351-1", "fib(100000)", "100000!", "99999#", "1<<4000", "2^2^2^2^2", "((((((((1))))))))",
                           "modexp(2,1000000,2^1277-1)", "modinv(3,2^1000+1)", "sqrt(2^4000)", "gcd(2^600-1,2^900-1)",
                           "2^1155-1", "2^2047-1", "2^2047+1", "2^1024+1", "0x", "0xffffffffffffffffffffffffffffffff",
                           "3", "1", "", "2^2100+1"};
    for (size_t i = 0; i < sizeof fixed / sizeof *fixed; i++) {
        int rc = gecm_prepare_input(fixed[i], 52, ndec, sizeof ndec, &inf, log, sizeof log);
        printf("%-36s rc=%d form=%d k=%d bits=%d\n", fixed[i], rc, rc ? 0 : inf.form, rc ? 0 : inf.k, rc ? 0 : inf.nbits);
    }
    /* tiny output buffers must be refused, not overrun */
    char small[4], slog[8];
    gecm_prepare_input("2^210-1", 52, small, sizeof small, &inf, slog, sizeof slog);
    printf("random expressions accepted: %d of 20000\n", ok);

    /* 2. mpl identities on random operands */
    for (int it = 0; it < 3000; it++) {
        mpl_t a, b, m, q, r, t, u;
        int la = 1 + (int)(rnd() % 60), lb = 1 + (int)(rnd() % 60);
        a.n = la; b.n = lb;
        for (int i = 0; i < la; i++) a.d[i] = (uint32_t)rnd();
        for (int i = 0; i < lb; i++) b.d[i] = (uint32_t)rnd();
        if (!a.d[la - 1]) a.d[la - 1] = 1;
        if (!b.d[lb - 1]) b.d[lb - 1] = 1;
        mpl_divrem(&q, &r, &a, &b);
        mpl_mul(&t, &q, &b);
        mpl_add(&t, &t, &r);
        if (mpl_cmp(&t, &a) || mpl_cmp(&r, &b) >= 0) { printf("divrem identity failed\n"); return 1; }
        m = b; m.d[0] |= 1;
        if (mpl_cmp_u64(&m, 3) >= 0) {
            mpl_mod(&u, &a, &m);
            if (mpl_invmod(&t, &u, &m)) {
                mpl_mulmod(&t, &t, &u, &m);
                if (mpl_cmp_u64(&t, 1)) { printf("invmod identity failed\n"); return 1; }
            }
            mpl_gcd(&t, &a, &m);
            mpl_mod(&u, &m, &t);
            if (!mpl_is_zero(&u)) { printf("gcd does not divide\n"); return 1; }
        }
        mpl_get_dec(buf, &a);
        mpl_set_str(&t, buf);
        if (mpl_cmp(&t, &a)) { printf("decimal round trip failed\n"); return 1; }
    }
    /* 3. tape compiler */
    const uint64_t b1s[] = {2, 3, 4, 5, 100, 1000, 65537, 1000000};
    for (size_t i = 0; i < sizeof b1s / sizeof *b1s; i++) {
        gecm_tape_t t;
        memset(&t, 0, sizeof t);
        int rc = gecm_tape_build_stage1(&t, b1s[i]);
        printf("tape B1=%lu rc=%d len=%zu adds=%lu dups=%lu\n", (unsigned long)b1s[i], rc, t.len, (unsigned long)t.ptadds,
               (unsigned long)t.ptdups);
        gecm_tape_free(&t);
    }
    /* 4. stage-2 plan and PAIR for several wheels */
    const uint32_t Ds[] = {30, 210, 1155, 2310};
    for (size_t i = 0; i < sizeof Ds / sizeof *Ds; i++)
        for (uint32_t U = 2; U <= 16; U *= 2) {
            gecm_s2_plan p;
            if (gecm_s2_plan_init(&p, Ds[i], U)) { printf("plan %u/%u refused\n", Ds[i], U); continue; }
            gecm_pairmap pm;
            int rc = gecm_pair(&pm, 1000, 300000, Ds[i], U);
            printf("D=%u U=%u npb=%u pair rc=%d steps=%u pairs=%u nump=%u\n", Ds[i], U, p.npb, rc, rc ? 0 : pm.steps,
                   rc ? 0 : pm.pairs, rc ? 0 : pm.nump);
            if (!rc) gecm_pairmap_free(&pm);
            gecm_s2_plan_free(&p);
        }
    /* 5. the device tape of a range: real pair maps, then hostile ones (nothing but window shifts with a tall table —
     * the case that overran a fixed-size tape in round 1 —, random entries, a table too tall for the ring) */
    {
        const uint32_t chunk = 512, ring = 1024;
        for (uint32_t U = 2; U <= 256; U *= 2) {
            gecm_s2_plan p;
            if (gecm_s2_plan_init(&p, 210, U)) continue;
            gecm_pairmap pm;
            gecm_s2_tape t;
            uint32_t bad = 0;
            if (!gecm_pair(&pm, 1000, 200000, 210, U)) {
                int rc = gecm_s2_tape_build(&t, &p, pm.steps, pm.v, pm.u, pm.amin, chunk, ring, &bad);
                uint64_t gen = 0, pairs = 0;
                if (!rc) {
                    for (size_t i = 0; i < t.nwords; i += 2)
                        if (t.words[i] == GECM_S2_GEN) gen += t.words[i + 1] & 0x7fffffffu; else pairs++;
                    if (pairs != pm.pairs || pairs != t.paired) { printf("tape: pair count mismatch\n"); return 1; }
                    free(t.words);
                }
                printf("tape U=%u rc=%d steps=%u marks+pairs=%zu giant steps=%lu\n", U, rc, pm.steps, rc ? 0 : t.nwords / 2,
                       (unsigned long)gen);
                gecm_pairmap_free(&pm);
            }
            /* 5000 window shifts and nothing else */
            uint32_t *z = (uint32_t *)calloc(5000, sizeof(uint32_t));
            int rc = gecm_s2_tape_build(&t, &p, 5000, z, z, 7, chunk, ring, &bad);
            if (!rc) {
                uint64_t gen = 0;
                for (size_t i = 0; i < t.nwords; i += 2) gen += t.words[i + 1] & 0x7fffffffu;
                if (gen != 2ull * p.L + 2ull * U * 5000) { printf("tape: wrong number of giant steps\n"); return 1; }
                if (!(t.words[t.nwords - 1] & 0x80000000u)) { printf("tape: last chunk not flagged\n"); return 1; }
                free(t.words);
            }
            printf("tape U=%u all-shifts rc=%d\n", U, rc);
            /* random entries: refused with the index of the first bad one, never a crash */
            for (int it = 0; it < 200; it++) {
                uint32_t v[64], u[64];
                for (int i = 0; i < 64; i++) { v[i] = (uint32_t)(rnd() % 40); u[i] = (uint32_t)(rnd() % (p.umax + 50)); }
                rc = gecm_s2_tape_build(&t, &p, 64, v, u, (uint32_t)(rnd() % 8), chunk, ring, &bad);
                if (!rc) free(t.words);
            }
            free(z);
            gecm_s2_plan_free(&p);
        }
    }
    size_t np;
    uint64_t *pr = gecm_primes_range(999999000ull, 1000001000ull, &np);
    printf("primes in [999999000, 1000001000): %zu\n", np);
    free(pr);
    printf("host sanitizer run complete\n");
    return 0;
}
