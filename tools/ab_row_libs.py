#!/usr/bin/env python3
"""Interleaved A/B of the 32-lane stage-1 kernel between shared libraries (GECM_LIB), separate processes on the same
GPU box: 4096 curves (AB_CURVES=n for another batch), B1 = 1e5, three passes each (the first discarded), repeated three
times; save lines compared.
usage: ab_row_libs.py libA.so libB.so@2 ... [-- bits ...]      (lib@m: GECM_ROW_ALDS=m, the operand-broadcast variant)"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, random, hashlib
sys.path.insert(0, os.path.join(%r, "avx-ecm_amd"))
import pyecm
bits = %d
n = random.Random(bits).getrandbits(bits) | (1 << (bits - 1)) | 1
eng = pyecm.Engine(n)
out = []
for _ in range(3):
    eng.build_curves(list(range(1000, 1000 + int(os.environ.get('AB_CURVES', '4096')))))
    eng.stage1(100000)
    out.append(eng.last_kernel_ms())
print(out[1:], hashlib.sha256("".join(eng.save_lines()[::16]).encode()).hexdigest()[:12], eng.lanes_per_curve())
'''
args = sys.argv[1:]
libs = args[:args.index("--")] if "--" in args else args
bits_list = [int(x) for x in args[args.index("--") + 1:]] if "--" in args else [415, 831]
for bits in bits_list:
    res = {l: [] for l in libs}
    shas = set()
    for rnd in range(3):
        for l in libs:
            env = dict(os.environ, GECM_LIB=os.path.join(ROOT, "avx-ecm_amd", l.split("@")[0]))
            if "@" in l:
                env["GECM_ROW_ALDS"] = l.split("@")[1]
            p = subprocess.run([sys.executable, "-c", CODE % (ROOT, bits)], env=env, capture_output=True, text=True)
            last = p.stdout.strip().splitlines()[-1]
            ms = eval(last.split("]")[0] + "]")
            shas.add(last.split()[-2])
            res[l] += ms
    for l in libs:
        v = sorted(res[l])
        print("%4d bits %-22s min %.1f  median %.1f  max %.1f ms   %s" % (bits, l, v[0], v[len(v) // 2], v[-1], "same residues" if len(shas) == 1 else "RESIDUES DIFFER"), flush=True)
