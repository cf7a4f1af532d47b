#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference compiled by `make -C oracle ref harness`
into oracle/_ref/).  The fixtures are data (inputs + outputs); no reference source is stored.

  stage1.json : (N, sigma0, curves, B1[, B2]) -> the save_b1.txt lines the reference wrote
                (ecm.c:1372-1380) + factor lines from ecm_results.txt (ecm.c:1362-1366, 1517-1520)
  l0.json     : per-operator vectors (a, b, N -> mulmod, sqrmod, addmod, submod, addsub) from
                the reference's vecarith52.c / vecarith.c via oracle/ref_l0_harness.c

  inputs.json : Cunningham-type command-line inputs -> the lines the reference prints while it
                prepares N (main.c:393-527) and, for two of them, the save lines of a short run

  stage2_acc.json : (N, sigma0, B1, B2) -> work->stg2acc of all VECLEN lanes as the reference reads it at
                ecm.c:1489, taken by oracle/ref_tap.c (the reference's ecm.c compiled with a tap on
                extract_bignum_from_vec_to_mpz; `make -C oracle reftap`), plus the D the reference chose
                (main.c:838-872) and its stage-2 counters

  multirange.json : B1 > 1e8 (several prime ranges, ecm.c:1209-1312): checkpoint.txt, save_b1.txt, the factor
                lines and the stdout lines of the reference for a 204-bit N at B1 = 1.1e8 (3 minutes of
                reference time per case)

  batches.json : runs of more than one reference batch (8 x threads curves, ecm.c:1151, 1531-1532): what the
                reference writes when a batch finds a factor (it stops after that batch) and how it labels
                curve / thread / vec with more than one thread (ecm.c:1356-1366; with a fixed sigma every
                thread runs the same eight sigmas, ecm.c:1187)

  special.json : special-form inputs end to end (stage 1 and stage 2): the reference works modulo 2^k -/+ 1 or 2^k - c

usage: python tests/golden/make_golden.py [--only stage1|l0|inputs|stage2acc|multirange|batches|special|degenerate] [--quick]
"""
import json, os, random, re, subprocess, sys, tempfile, hashlib

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from xladder import true_stage1_point  # noqa: E402


def rand_n(bits):
    """SURVEY.md §8c/§8d input generator."""
    return random.Random(bits).getrandbits(bits) | (1 << (bits - 1)) | 1



def test_csh_case(sigma):
    """(N, curves, B1, threads, B2, sigma) of the test.csh line that uses this sigma."""
    for l in open("/root/reference/test.csh").read().splitlines():
        f = l.split()
        if len(f) >= 7 and f[6] == str(sigma):
            return int(f[1]), int(f[3]), int(f[5]), int(f[6])
    raise KeyError(sigma)



def run_ref(digitbits, n_expr, curves, b1, b2, sigma, threads=1, keep_stdout=False):
    exe = os.path.join(REFDIR, "avx-ecm-%d" % digitbits)
    with tempfile.TemporaryDirectory() as d:
        cmd = [exe, str(n_expr), str(curves), str(b1), str(threads), str(b2), str(sigma)]
        p = subprocess.run(cmd, cwd=d, capture_output=True, text=True, timeout=3600)
        out = p.stdout
        save = []
        if os.path.exists(os.path.join(d, "save_b1.txt")):
            save = open(os.path.join(d, "save_b1.txt")).read().splitlines()
        ckpt = None
        if os.path.exists(os.path.join(d, "checkpoint.txt")):
            ckpt = open(os.path.join(d, "checkpoint.txt")).read().splitlines()
        res = []
        if os.path.exists(os.path.join(d, "ecm_results.txt")):
            res = [l for l in open(os.path.join(d, "ecm_results.txt")).read().splitlines() if l.strip()]
    m = re.search(r"Choosing MAXBITS = (\d+), NWORDS = (\d+)", out)
    cnt = re.search(r"with (\d+) point-adds and (\d+) point-doubles", out)
    s2 = re.search(r"performed (\d+) pt-adds, (\d+) inversions, and (\d+) pair-muls", out)
    t1 = re.search(r"Stage 1 took ([0-9.]+) seconds", out)
    return {
        "digitbits": digitbits, "N": str(n_expr), "curves": curves, "B1": b1, "B2": b2, "sigma0": sigma,
        "maxbits": int(m.group(1)) if m else None, "nwords": int(m.group(2)) if m else None,
        "ptadds": int(cnt.group(1)) if cnt else None, "ptdups": int(cnt.group(2)) if cnt else None,
        "stage2_counts": [int(x) for x in s2.groups()] if s2 else None,
        "stage1_seconds_container": float(t1.group(1)) if t1 else None,
        "save_lines": save, "results_lines": res,
        "save_sha256": hashlib.sha256(("\n".join(save) + "\n").encode()).hexdigest() if save else None,
        **({"checkpoint_lines": ckpt} if ckpt is not None else {}),
        **({"stdout_lines": stdout_protocol(out)} if keep_stdout else {}),
    }


def stdout_protocol(out):
    """the reference's stdout without what changes from run to run: progress lines (\\r), pid, timings"""
    keep = []
    for l in out.replace("\r", "\n").splitlines():
        if l.startswith(("accumulating prime", "starting process")) or re.search(r"took [0-9.]+ seconds", l):
            continue
        keep.append(l)
    return keep


def gen_stage1(quick):
    cases = []

    def add(name, *a, **k):
        print("stage1:", name, flush=True)
        c = run_ref(*a, **k)
        c["name"] = name
        cases.append(c)

    n415, n623, n831, n1023 = rand_n(415), rand_n(623), rand_n(831), rand_n(1023)
    # stage-1-only runs: B2 = B1 disables stage 2 (main.c:548-552)
    for b1 in (1000, 10000, 100000) + (() if quick else (1000000,)):
        add("n415_b1_%d" % b1, 52, n415, 8, b1, b1, 1000)
    for b1 in (1000, 10000) + (() if quick else (1000000,)):
        add("n623_b1_%d" % b1, 52, n623, 8, b1, b1, 1000)
        add("n831_b1_%d" % b1, 52, n831, 8, b1, b1, 1000)
    for b1 in (1000, 10000) + (() if quick else (100000,)):
        add("n1023_d32_b1_%d" % b1, 32, n1023, 16, b1, b1, 1000)
    # K3: limb-width independence (same N through the 32-bit build)
    add("n415_d32_b1_10000", 32, n415, 16, 10000, 10000, 1000)
    # 64-bit sigma values (test_t35.csh style) and small / odd sizes
    add("n415_bigsigma_b1_1000", 52, n415, 8, 1000, 1000, 11919771003873180376)
    add("n200_b1_1000", 52, rand_n(200), 8, 1000, 1000, 42)
    add("n64_b1_500", 52, rand_n(64), 8, 500, 500, 7)
    # two batches of 8 (curves 16, 1 thread): sigma0..sigma0+15 unless a factor stops it early
    add("n415_two_batches_b1_1000", 52, rand_n(414) , 16, 1000, 1000, 5000)
    # KATs from test.csh / test_inputs.txt: K1 stage-1 factor, K2 stage-2 factor
    if not quick:
        n, b1, b2, sg = test_csh_case(7372562557)
        add("K1", 52, n, 8, b1, b1, sg)
        # K1's N has no small factors: 16 curves at a tiny B1 run as two batches of 8 (ecm.c:1151)
        add("K1N_two_full_batches_b1_500", 52, n, 16, 500, 500, 100)
        add("K1N_d32_b1_2000", 32, n, 16, 2000, 2000, 77)
        n, b1, b2, sg = test_csh_case(3018506502)
        add("K2", 52, n, 8, b1, b2, sg)
        t35 = open("/root/reference/test_t35.csh").read().splitlines()[45].split()
        add("T35_46", 52, int(t35[1]), 8, int(t35[3]), int(t35[5]), int(t35[6]))
        # config 1 with a pinned sigma, B2 default 100*B1 given explicitly
        add("config1_fib791", 52, "fib(791)/13/677/216416017", 8, 1000000, 100000000, 1000)
    # stage-2 small cases (factor found or not; accumulators are not observable from the reference)
    add("n415_b1_10000_b2_1e6", 52, n415, 8, 10000, 1000000, 1000)
    json.dump(cases, open(os.path.join(HERE, "stage1.json"), "w"), indent=1)


def gen_l0():
    rng = random.Random(20261004)
    out = []
    for digitbits, veclen, sizes in ((52, 8, ((8, 415), (8, 412), (12, 623), (16, 831), (4, 200), (8, 400))),
                                     (32, 16, ((32, 1023), (16, 415), (4, 127)))):
        exe = os.path.join(REFDIR, "l0_harness-%d" % digitbits)
        for nwords, bits in sizes:
            n = rand_n(bits)
            pairs = []
            edge = [0, 1, 2, n - 1, n - 2, (n + 1) // 2, (n - 1) // 2, (1 << (bits - 1)) - 1]
            for a in edge:
                for b in (0, 1, n - 1, (n + 1) // 2):
                    pairs.append((a, b))
            while len(pairs) % veclen:
                pairs.append((rng.randrange(n), rng.randrange(n)))
            for _ in range(4 * veclen):
                pairs.append((rng.randrange(n), rng.randrange(n)))
            inp = "".join("%x %x\n" % p for p in pairs)
            p = subprocess.run([exe, str(nwords), "%x" % n], input=inp, capture_output=True, text=True, check=True)
            rows = [l.split() for l in p.stdout.splitlines()]
            assert len(rows) == len(pairs), (len(rows), len(pairs))
            out.append({"digitbits": digitbits, "nwords": nwords, "N": "%x" % n,
                        "a": ["%x" % x[0] for x in pairs], "b": ["%x" % x[1] for x in pairs],
                        "mul": [r[0] for r in rows], "sqr": [r[1] for r in rows],
                        "add": [r[2] for r in rows], "sub": [r[3] for r in rows],
                        "asum": [r[4] for r in rows], "adiff": [r[5] for r in rows]})
            print("l0:", digitbits, nwords, bits, len(pairs), flush=True)
    json.dump(out, open(os.path.join(HERE, "l0.json"), "w"), indent=0)


def gen_inputs():
    """Cunningham-type inputs (main.c:405-457, 505-527)."""
    def phi2(m):                               # cyclotomic polynomial value Phi_m(2)
        def mu(k):
            r, p = 1, 2
            while p * p <= k:
                if k % p == 0:
                    k //= p
                    if k % p == 0:
                        return 0
                    r = -r
                p += 1
            return -r if k > 1 else r
        num = den = 1
        for d in range(1, m + 1):
            if m % d == 0:
                if mu(m // d) == 1:
                    num *= 2 ** d - 1
                elif mu(m // d) == -1:
                    den *= 2 ** d - 1
        return num // den
    redc = phi2(105) * phi2(210)               # | 2^210 - 1, 97 bits: the reference stays with REDC
    cof251 = (2 ** 251 - 1) // 503 // 54217    # cofactor of M251: the reference folds modulo 2^251 - 1
    exprs = [str(redc), str(cof251), "2^210-1", "2^300+1", "2^251-1", "2^1009-1", "2^945+1", "2^127-1",
             "(2^61-1)*(2^89-1)", "2^1155-1", "2^64+13", "2^400-593", "11526466273339081241",
             "fib(791)/13/677/216416017", "(2^499-1)/20959", "(2^523+1)/3",
             # the calculator (calc.c): functions and precedence
             "nroot(10^300,3)+1", "nroot(2^500+12345,7)*2+1", "lg2(2^400)*2^200+1", "log(10^77)*10^50+1",
             "xor(2^200+5,2^100+2)", "and(2^200-1,2^150+2^60+1)", "or(2^120,2^60+1)", "abs(2^150+1)", "sqrt(10^100+7)+3",
             "modinv(17,2^200+1)*2+1", "modexp(3,10^40,2^255-19)", "gcd(fib(601)*3,fib(900)*3)", "100!+1", "149#+1",
             "(2^300>>100)+1", "(1<<250)+1", "luc(401)", "7^180 % 10^120 + 10^121", "2^2^3+1", "(10^59-1)/9"]
    exe = os.path.join(REFDIR, "avx-ecm-52")
    banner, runs = [], []
    for e in exprs:
        with tempfile.TemporaryDirectory() as d:
            p = subprocess.run([exe, e, "8", "100", "1", "100", "1000"], cwd=d, capture_output=True, text=True, timeout=600)
        keep = [l for l in p.stdout.splitlines() if l.startswith(("gen:", "removing", "commencing", "Mersenne input"))]
        m = re.search(r"commencing parallel ecm on (\d+)", p.stdout)
        banner.append({"expr": e, "lines": keep, "N": m.group(1) if m else None,
                       "special_reduction": "Using special" in p.stdout})
        print("inputs:", e[:40], len(keep), flush=True)
    for name, e in (("redc_phi105_phi210", str(redc)), ("special_m251_cofactor", str(cof251))):
        c = run_ref(52, e, 8, 2000, 2000, 1000)
        c["name"] = name
        n = int(c["save_lines"][0].split("N=0x")[1].split(";")[0], 16)
        ok = []
        for l in c["save_lines"]:
            g = lambda key: int(l.split(key + "=0x")[1].split(";")[0], 16)
            X, Z = true_stage1_point(n, int(l.split("SIGMA=")[1].split(";")[0]), 2000)
            ok.append((X * g("Z") - g("X") * Z) % n == 0)
        c["reference_lane_is_the_true_point"] = ok
        runs.append(c)
        print("inputs run:", name, ok, flush=True)
    json.dump({"banner": banner, "runs": runs}, open(os.path.join(HERE, "inputs.json"), "w"), indent=1)


def gen_stage2_acc():
    """stg2acc of the reference itself for a few small (B1, B2): moduli without small factors (every inversion
    succeeds, so the accumulator is a well-defined product), VECLEN curves on one thread."""
    k1n = int([l for l in open("/root/reference/test.csh").read().splitlines() if "7372562557" in l][0].split()[1])
    t35 = int(open("/root/reference/test_t35.csh").read().splitlines()[45].split()[1]) if os.path.exists("/root/reference/test_t35.csh") else k1n
    M = lambda e: (1 << e) - 1
    path = os.path.join(HERE, "stage2_acc.json")
    cases = json.load(open(path)) if os.path.exists(path) and "--keep" in sys.argv else []
    have = {c["name"] for c in cases}
    t35_sigma = int(open("/root/reference/test_t35.csh").read().splitlines()[45].split()[6]) if os.path.exists("/root/reference/test_t35.csh") else 100
    for name, digitbits, n, b1, b2, sigma0 in (("K1N_b1_2000_b2_1e5", 52, k1n, 2000, 100000, 100),
                                               # BASELINE's own size (configs[3]: B1 = 1e6, B2 = 1e8, D = 2310, U = 16): the
                                               # reference's stage-2 KAT (test_t35.csh line 46: lane 0 finds its PRP31) and
                                               # the 412-bit N of test.csh (416-bit class, no factor found)
                                               ("T35_46_b1_1e6_b2_1e8", 52, t35, 1000000, 100000000, t35_sigma),
                                               ("K1N_b1_1e6_b2_1e8", 52, k1n, 1000000, 100000000, 4000),
                                               ("K1N_b1_5000_b2_3e5", 52, k1n, 5000, 300000, 200),
                                               ("K1N_b1_3000_b2_150000", 52, k1n, 3000, 150000, 500),
                                               ("T35N_b1_1000_b2_50000", 52, t35, 1000, 50000, 42),
                                               ("K1N_d32_b1_300_b2_20000", 32, k1n, 300, 20000, 300),
                                               # the larger size classes: products of Mersenne primes (no small factors)
                                               ("M607xM127xM89_b1_800_b2_40000", 52, M(607) * M(127) * M(89), 800, 40000, 700),
                                               ("M607xM127xM107xM89xM61_d32_b1_500_b2_30000", 32,
                                                M(607) * M(127) * M(107) * M(89) * M(61), 500, 30000, 900),
                                               ("M521xM127_b1_1200_b2_60000", 52, M(521) * M(127), 1200, 60000, 1100)):
        if name in have:
            continue
        print("stage2acc:", name, flush=True)
        exe = os.path.join(REFDIR, "avx-ecm-%d-tap" % digitbits)
        veclen = 8 if digitbits == 52 else 16
        with tempfile.TemporaryDirectory() as d:
            tap = os.path.join(d, "tap.txt")
            p = subprocess.run([exe, str(n), str(veclen), str(b1), "1", str(b2), str(sigma0)], cwd=d, capture_output=True,
                               text=True, timeout=3600, env=dict(os.environ, GECM_TAP_FILE=tap))
            rows = [l.split() for l in open(tap).read().splitlines()]
            res = [l for l in open(os.path.join(d, "ecm_results.txt")).read().splitlines() if l.strip()] \
                if os.path.exists(os.path.join(d, "ecm_results.txt")) else []
        last = rows[-veclen:]
        assert len({r[0] for r in last}) == 1 and [int(r[1]) for r in last] == list(range(veclen)), "tap tail is not stg2acc"
        s2 = re.search(r"performed (\d+) pt-adds, (\d+) inversions, and (\d+) pair-muls", p.stdout)
        w = re.search(r"w = (\d+), R = \d+, L = (\d+), U = (\d+)", p.stdout)
        m = re.search(r"Choosing MAXBITS = (\d+), NWORDS = (\d+)", p.stdout)
        cases.append({"name": name, "digitbits": digitbits, "N": str(n), "B1": b1, "B2": b2, "sigma0": sigma0,
                      "curves": veclen, "maxbits": int(m.group(1)), "nwords": int(m.group(2)),
                      "D": int(w.group(1)) if w else None, "L": int(w.group(2)) if w else None,
                      "U": int(w.group(3)) if w else None,
                      "stage2_counts": [int(x) for x in s2.groups()], "acc_hex": [r[2] for r in last],
                      "results_lines": res})
    json.dump(cases, open(path, "w"), indent=1)


def semiprime_204():
    """a 98-bit prime times a 106-bit prime (NWORDS = 4: the cheapest class), no small factors"""
    def isprime(n):
        if n < 2:
            return False
        for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            if n % p == 0:
                return n == p
        d, s = n - 1, 0
        while d % 2 == 0:
            d //= 2
            s += 1
        for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            x = pow(a, d, n)
            if x in (1, n - 1):
                continue
            for _ in range(s - 1):
                x = x * x % n
                if x == n - 1:
                    break
            else:
                return False
        return True

    def nextprime(n):
        n |= 1
        while not isprime(n):
            n += 2
        return n
    r = random.Random(20261004)
    p = nextprime(r.getrandbits(98) | 1 << 97)
    q = nextprime(r.getrandbits(106) | 1 << 105)
    return p * q


def gen_multirange():
    """B1 above one prime range: the reference re-sieves per range of 1e8, runs ecm_stage1 once per range and appends
    a checkpoint.txt line per curve after every range but the last (ecm.c:1209-1312)."""
    n = semiprime_204()
    path = os.path.join(HERE, "multirange.json")
    cases = json.load(open(path)) if os.path.exists(path) and "--keep" in sys.argv else []
    have = {c["name"] for c in cases}
    for name, b1, b2, sigma0 in (("n204_b1_1.1e8", 110000000, 110000000, 1000),
                                 ("n204_b1_1.1e8_b2_1.3e8", 110000000, 130000000, 2000),
                                 # one range whose primes all lie below B1: the reference writes a checkpoint even so
                                 # (ecm.c:1237 reads one past its prime list)
                                 ("n204_b1_1e8_single_range_checkpoint", 100000000, 100000000, 3000)):
        if name in have:
            continue
        print("multirange:", name, flush=True)
        c = run_ref(52, n, 8, b1, b2, sigma0, keep_stdout=True)
        c["name"] = name
        cases.append(c)
    json.dump(cases, open(path, "w"), indent=1)


def gen_special():
    """Inputs for which the reference leaves REDC (main.c:505-527, 642-684): N | 2^k - 1, N | 2^k + 1, 2^k = c (mod N).
    It then works modulo Mw = 2^k -/+ 1 or 2^k - c throughout — curve construction included — and its files hold
    residues modulo Mw next to N= the number given (ecm.c:1111-1118)."""
    cof251 = (2 ** 251 - 1) // 503 // 54217
    cases = []
    for name, expr, b1, b2, sigma0 in (("M251_cofactor", str(cof251), 2000, 50000, 1000),
                                       ("M251_cofactor_b1_20000_stage2", str(cof251), 20000, 1000000, 7000),
                                       ("M499_cofactor", "(2^499-1)/20959", 2000, 50000, 1000),
                                       ("P523_cofactor", "(2^523+1)/3", 2000, 50000, 1000),
                                       ("pseudo_2^400-593", "2^400-593", 2000, 50000, 1000),
                                       ("pseudo_2^64+13", "2^64+13", 500, 5000, 1000),
                                       ("M127", "2^127-1", 1000, 20000, 1000),
                                       ("M1009", "2^1009-1", 500, 10000, 1000),
                                       ("F8_2^2^3+1", "2^2^3+1", 500, 5000, 1000)):
        print("special:", name, flush=True)
        c = run_ref(52, expr, 8, b1, b2, sigma0, keep_stdout=True)
        c["name"] = name
        m = re.search(r"Using special (?:pseudo-)?Mersenne mod for factor of: 2\^(\d+)([-+])(\d+)", "\n".join(c["stdout_lines"]))
        c["special"] = {"k": int(m.group(1)), "sign": m.group(2), "c": int(m.group(3))} if m else None
        cases.append(c)
    json.dump(cases, open(os.path.join(HERE, "special.json"), "w"), indent=1)


def gen_degenerate():
    """A modulus made of many small primes (the cofactor of 2^496 - 1 times a 98-bit prime, so that the reference stays
    with REDC): stage-2 inversions fail batch after batch, with different gcds, differential additions degenerate modulo
    the small primes, and the reference's accumulator ends as the gcd of its last failing batch (ecm.c:1925-1939).  Found
    by tools/soak_fuzz.py; pins the oracle's restatement of that behaviour and the HIP path's contract there."""
    n = 3121796185145477483418392554386586511231252428132929722174497757015462917147969171896000316935515523337612234042458296148257855102126147338567681
    p = 297467847534123075601765177943
    c = run_ref(52, n * p, 8, 65, 50085, 768295079280089151, keep_stdout=True)
    c["name"] = "many_small_primes_b1_65_b2_50085"
    json.dump([c], open(os.path.join(HERE, "degenerate.json"), "w"), indent=1)


def gen_batches():
    """More curves than one reference batch (8 x threads)."""
    n415 = rand_n(415)
    k1n = int([l for l in open("/root/reference/test.csh").read().splitlines() if "7372562557" in l][0].split()[1])
    cases = []
    for name, n, curves, b1, b2, sigma0, threads in (
            # small factors: the first batch finds one, the reference writes 8 lines and stops (ecm.c:1531-1532)
            ("n415_64_curves_stops_after_first_batch", n415, 64, 1000, 1000, 1000, 1),
            # two threads: batches of 16 lines, both threads on the same eight sigmas (ecm.c:1187), thread/vec labels
            ("n415_64_curves_2_threads", n415, 64, 1000, 1000, 1000, 2),
            # no factor anywhere: every batch is written
            ("K1N_32_curves_2_threads_b1_300", k1n, 32, 300, 300, 100, 2),
            ("K1N_40_curves_3_threads_b1_300", k1n, 40, 300, 300, 100, 3),
            # a factor in the second batch only, found in stage 2 (lane 3 of batch 1)
            ("n415_stage2_24_curves", n415, 24, 2000, 100000, 1000, 1)):
        print("batches:", name, flush=True)
        c = run_ref(52, n, curves, b1, b2, sigma0, threads=threads, keep_stdout=True)
        c["name"] = name
        c["threads"] = threads
        cases.append(c)
    json.dump(cases, open(os.path.join(HERE, "batches.json"), "w"), indent=1)


if __name__ == "__main__":
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    quick = "--quick" in sys.argv
    if only in (None, "l0"):
        gen_l0()
    if only in (None, "stage1"):
        gen_stage1(quick)
    if only in (None, "inputs"):
        gen_inputs()
    if only in (None, "stage2acc"):
        gen_stage2_acc()
    if only in (None, "multirange"):
        gen_multirange()
    if only in (None, "batches"):
        gen_batches()
    if only in (None, "special"):
        gen_special()
    if only in (None, "degenerate"):
        gen_degenerate()
