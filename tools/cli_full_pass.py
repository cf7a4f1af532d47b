#!/usr/bin/env python3
"""The command-line driver at full scale, end to end (outside the suite: a minute of GPU time).

A. 262,144 curves (two full passes of 131,072 with stage 2) on the modulus of the reference's stage-2 KAT (test_t35.csh
   line 46), B1 = 1e6, B2 = 1e8.  Some curve of the first pass finds the 31-digit factor; the reference would have run
   batch after batch of 8 and stopped after the first batch with a factor: the driver must write exactly the batches up to
   that one, that batch's factor lines with the reference's labels, and must not even start the second pass.  What
   "the first batch with a factor" is, is checked with the library on those curves.
B. 262,144 curves on a modulus without small factors, B1 = 1e5, B2 = 1e7: both passes run (pipelined, stage-2 tables of
   two sets of contexts resident), all lines written in sigma order, samples equal to a small run of the library."""
import json, os, re, subprocess, sys, tempfile, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "avx-ecm_amd"))
import pyecm
S1 = {x["name"]: x for x in json.load(open(os.path.join(ROOT, "tests", "golden", "stage1.json")))}
exe = os.path.join(ROOT, "avx-ecm_amd", "avx-ecm")


def run(args):
    with tempfile.TemporaryDirectory() as d:
        t = time.time()
        p = subprocess.run([exe] + [str(a) for a in args], cwd=d, capture_output=True, text=True)
        wall = time.time() - t
        assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
        save = open(os.path.join(d, "save_b1.txt")).read().splitlines()
        res = [l for l in open(os.path.join(d, "ecm_results.txt")).read().splitlines() if l.strip()] \
            if os.path.exists(os.path.join(d, "ecm_results.txt")) else []
    out = p.stdout.replace("\r", "\n")
    print("\n".join(l for l in out.splitlines() if re.match(r"(Commencing curves|Stage 1 took|Stage 2 took|performed|found|Process took|\(.*curves/sec)", l)), flush=True)
    return out, save, res, wall


# ---- A ----
c = S1["T35_46"]
sigma0 = 471960146
out, save, res, wall = run([c["N"], 262144, c["B1"], 1, c["B2"], sigma0])
n = int(save[0].split("N=0x")[1].split(";")[0], 16)
assert len(save) % 8 == 0 and 8 <= len(save) < 131072
assert [int(l.split("SIGMA=")[1].split(";")[0]) for l in save] == list(range(sigma0, sigma0 + len(save)))
assert out.count("Commencing curves") == 1, "the second pass must not start behind a factor"
eng = pyecm.Engine(n)
eng.build_curves(list(range(sigma0, sigma0 + len(save))))
eng.stage1(c["B1"])
assert [l.rstrip("\n") for l in eng.save_lines()] == save
f1 = [k for k in range(len(save)) if eng.stage1_factor(k)]
eng.stage2(c["B2"])
f2 = [k for k in range(len(save)) if eng.stage2_factor(k)]
first = min(f1 + f2)
assert first // 8 == len(save) // 8 - 1, (first, len(save))          # the batch of the first finder is the last one written
want = []
for k in f1:
    f, prp = eng.stage1_factor(k)
    want.append("found %s%d factor %d in stage 1 (B1 = %d): curve %d, thread 0, vec %d, sigma %d" % ("PRP" if prp else "C", len(str(f)), f, c["B1"], k, k % 8, sigma0 + k))
for k in f2:
    f, prp = eng.stage2_factor(k)
    want.append("found %s%d factor %d in stage 2 (B2 = %d): curve %d, thread 0, vec %d, sigma %d" % ("PRP" if prp else "C", len(str(f)), f, c["B2"], k, k % 8, sigma0 + k))
eng.close()
strip = lambda l: re.sub(r"found (PRP|C)\d+ factor", "found factor", l)      # the size label is mpz_sizeinbase's, one off at times
assert [strip(l) for l in res] == [strip(l) for l in want], (res, want)
print("A ok: first factor at curve %d -> %d save lines, %d factor line(s), one pass; wall %.1f s" % (first, len(save), len(res), wall), flush=True)

# ---- B ----
c = S1["K1N_two_full_batches_b1_500"]
sigma0, b1, b2 = 2000000, 100000, 10000000
out, save, res, wall = run([c["N"], 262144, b1, 1, b2, sigma0])
n = int(save[0].split("N=0x")[1].split(";")[0], 16)
assert len(save) == 262144 and res == []
assert [int(l.split("SIGMA=")[1].split(";")[0]) for l in save] == list(range(sigma0, sigma0 + 262144))
assert out.count("Commencing curves") == 2 and "Commencing curves 131072-262143 of 262144" in out
eng = pyecm.Engine(n)
pick = [0, 1, 63, 64, 131071, 131072, 200000, 262143]
eng.build_curves([sigma0 + k for k in pick])
eng.stage1(b1)
assert [l.rstrip("\n") for l in eng.save_lines()] == [save[k] for k in pick]
eng.close()
kern = [float(x) for x in re.findall(r"kernel ([0-9.]+) ms on GPU 0", out)]
s2 = [float(x) for x in re.findall(r"Stage 2 took ([0-9.]+) seconds", out)]
took = float(re.search(r"Process took ([0-9.]+) seconds", out).group(1))
print("B ok: 262144 lines in sigma order, two pipelined passes with stage 2; stage-1 kernels %s ms, stage 2 %s s, process %.2f s" % (kern, s2, took), flush=True)
