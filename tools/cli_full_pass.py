#!/usr/bin/env python3
"""The command-line driver at full scale, end to end: the reference's stage-2 KAT (test_t35.csh line 46: sigma 472060146
finds a PRP31 in stage 2 at B1 = 1e6, B2 = 1e8) hidden at curve 100,000 of a 262,144-curve run — two full passes of
131,072 curves with stage 2, pipelined.  The reference would run 12,501 batches of 8 and stop after the one that finds
the factor; the driver must write exactly those batches (100,008 save lines), the one factor line with the reference's
labels, and nothing of the second pass.  Sample lines are checked against a small run of the library."""
import json, os, re, subprocess, sys, tempfile, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "avx-ecm_amd"))
import pyecm
c = [x for x in json.load(open(os.path.join(ROOT, "tests", "golden", "stage1.json"))) if x["name"] == "T35_46"][0]
kat = int(c["save_lines"][0].split("SIGMA=")[1].split(";")[0])
at = 100000
sigma0 = kat - at
exe = os.path.join(ROOT, "avx-ecm_amd", "avx-ecm")
with tempfile.TemporaryDirectory() as d:
    t = time.time()
    p = subprocess.run([exe, c["N"], "262144", str(c["B1"]), "1", str(c["B2"]), str(sigma0)], cwd=d, capture_output=True, text=True)
    wall = time.time() - t
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    save = open(os.path.join(d, "save_b1.txt")).read().splitlines()
    res = [l for l in open(os.path.join(d, "ecm_results.txt")).read().splitlines() if l.strip()]
out = p.stdout.replace("\r", "\n")
print("\n".join(l for l in out.splitlines() if re.match(r"(Commencing curves|Stage 1 took|Stage 2 took|performed|found|Process took|\(.*curves/sec)", l)))
assert len(save) == 8 * (at // 8 + 1), len(save)
assert [int(l.split("SIGMA=")[1].split(";")[0]) for l in save] == list(range(sigma0, sigma0 + len(save)))
assert save[at] == c["save_lines"][0]
want = c["results_lines"][0].replace("curve 0,", "curve %d," % at)
assert res == [want], (res, want)
assert "performed 79886 pt-adds, 1341 inversions, and 3008627 pair-muls in stage 2" in out
eng = pyecm.Engine(int(save[0].split("N=0x")[1].split(";")[0], 16))
pick = [0, 1, 63, 64, 65535, 99999, 100007]
eng.build_curves([sigma0 + k for k in pick])
eng.stage1(c["B1"])
assert [l.rstrip("\n") for l in eng.save_lines()] == [save[k] for k in pick]
eng.close()
print("cli_full_pass ok: %d save lines, factor at curve %d reported as the reference labels it, second pass not written; wall %.1f s" % (len(save), at, wall))
