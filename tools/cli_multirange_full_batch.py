#!/usr/bin/env python3
"""B1 above one prime range AT BATCH SIZE: the command-line driver on 4096 curves of the multi-range fixture's 204-bit N at
B1 = 1.1e8 (two prime ranges, 26 stage-1 launches, checkpoint.txt after the first range).  Curves 0-7 are the eight
curves of tests/golden/multirange.json (sigma 1000..1007, made by the reference in three minutes): their checkpoint and
save lines must be the reference's byte for byte inside the big batch.  checkpoint.txt holds the whole batch (it is written
before anybody looks for factors); save_b1.txt ends with the reference batch of eight in which the first factor turned
up (ecm.c:1531-1532) — with 4096 curves on a 204-bit semiprime at this B1 one does (sigma 1057).
usage (GPU box): cli_multirange_full_batch.py [curves]      stdout of the driver goes to gpurun_out/ as it is written"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
curves = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
c = [x for x in json.load(open(os.path.join(ROOT, "tests", "golden", "multirange.json"))) if x["name"] == "n204_b1_1.1e8"][0]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
log = os.path.join(ROOT, "gpurun_out", "cli_multirange_full_batch_stdout.txt")
with tempfile.TemporaryDirectory() as d:
    t0 = time.time()
    with open(log, "w") as f:
        rc = subprocess.call([os.path.join(ROOT, "avx-ecm_amd", "avx-ecm"), c["N"], str(curves), str(c["B1"]), "1", str(c["B2"]), str(c["sigma0"])],
                             cwd=d, stdout=f, stderr=subprocess.STDOUT, timeout=1000)
    wall = time.time() - t0
    assert rc == 0, rc
    save = open(os.path.join(d, "save_b1.txt")).read().splitlines()
    ckpt = open(os.path.join(d, "checkpoint.txt")).read().splitlines()
out = open(log).read().replace("\r", "\n").splitlines()
found = [l for l in out if l.startswith("found ")]
assert len(save) % 8 == 0 and (len(save) == curves or found), (len(save), len(ckpt), found)
assert len(ckpt) in (curves, len(save)), (len(save), len(ckpt))
assert ckpt[:8] == c["checkpoint_lines"], "checkpoint lines of curves 0-7 differ from the reference's"
assert save[:8] == c["save_lines"], "save lines of curves 0-7 differ from the reference's"
assert len(set(save)) == len(save)
for l in c["stdout_lines"]:
    if l.startswith(("Found ", "Saving checkpoint")):
        assert l in out, l
ms = [l for l in out if "Stage 1 completed" in l or "kernel" in l.lower()]
print("%d curves x 204 bits, B1 = %d: wall %.1f s; checkpoint.txt holds %d lines, save_b1.txt %d; lines 0-7 of both equal "
      "the reference's (tests/golden/multirange.json n204_b1_1.1e8)" % (curves, c["B1"], wall, len(ckpt), len(save)))
for l in found:
    print("  " + l)
for l in ms[:8]:
    print("  " + l.strip())
