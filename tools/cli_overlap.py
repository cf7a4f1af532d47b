#!/usr/bin/env python3
"""How much of the command-line driver's wall time is kernel time: `avx-ecm N 262144 1e5 1 1e5 sigma` (two passes of
131072 curves, stage 1 only) pipelined and not pipelined.  Prints the per-pass kernel times, their sum and the
program's own "Process took"."""
import os, re, subprocess, sys, tempfile, time

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
exe = os.path.join(root, "avx-ecm_amd", "avx-ecm")
n = "7908926676514675413083853032827063880118980193445471625562601469958414706043143581401715516956542424923236530406833110566233"
curves = sys.argv[1] if len(sys.argv) > 1 else "262144"
b1 = sys.argv[2] if len(sys.argv) > 2 else "100000"
for label, env in (("pipelined (default)", {}), ("GECM_NO_PIPELINE=1", {"GECM_NO_PIPELINE": "1"})):
    with tempfile.TemporaryDirectory() as d:
        t = time.time()
        p = subprocess.run([exe, n, curves, b1, "1", b1, "1000"], cwd=d, capture_output=True, text=True, env=dict(os.environ, **env))
        wall = time.time() - t
        lines = sum(1 for _ in open(os.path.join(d, "save_b1.txt")))
    k = [float(x) for x in re.findall(r"kernel ([0-9.]+) ms on GPU 0", p.stdout)]
    took = float(re.search(r"Process took ([0-9.]+) seconds", p.stdout).group(1))
    init = float(re.search(r"Initialization took ([0-9.]+) seconds", p.stdout).group(1))
    print("%-22s %s curves B1=%s: kernels %s ms, sum %.3f s; Process took %.3f s (of which initialisation %.3f s); "
          "wall %.3f s; %d save lines; process/kernels = %.3f" % (label, curves, b1, ["%.0f" % x for x in k], sum(k) / 1e3, took, init, wall, lines, took / (sum(k) / 1e3)), flush=True)
