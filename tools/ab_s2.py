"""Interleaved A/B of stage-2 variants (separate libraries, separate processes, same box)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:]
for rnd in range(2):
    for l in libs:
        env = dict(os.environ, GECM_LIB=os.path.join(ROOT, "avx-ecm_amd", l))
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "s2_small.py"), "100000000", "131072"],
                           env=env, capture_output=True, text=True)
        print(rnd, l, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:], flush=True)
