"""Interleaved A/B timing of stage-1 kernel variants in separate processes on the SAME GPU box
(variants are separate shared libraries; each measurement = one stage-1 pass at B1=1e5)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, random
sys.path.insert(0, os.path.join(%r, "avx-ecm_amd"))
import pyecm
n = random.Random(415).getrandbits(415) | (1 << 414) | 1
eng = pyecm.Engine(n)
eng.build_curves(list(range(1000, 1000 + 131072)))
out = []
for _ in range(%d):
    eng.stage1(%d)
    out.append(eng.last_kernel_ms())
print(out)
''' 
libs = sys.argv[1:]
b1 = 100000
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ, GECM_LIB=os.path.join(ROOT, "avx-ecm_amd", l))
        p = subprocess.run([sys.executable, "-c", CODE % (ROOT, 2, b1)], env=env, capture_output=True, text=True)
        ms = eval(p.stdout.strip().splitlines()[-1])
        res[l] += ms
        print(rnd, l, ["%.1f" % x for x in ms], flush=True)
for l in libs:
    v = sorted(res[l])
    print("%-28s min %.1f  median %.1f ms  (B1=%d, 131072 curves)" % (l, v[0], v[len(v) // 2], b1))
