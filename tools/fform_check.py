"""F-form (2^k - 1) stage 1 against the generic REDC path: same save lines, and kernel time of both.
usage: python tools/fform_check.py k [curves] [B1] [lanes] [c]     (negative k: N = 2^|k| + 1; c: N = 2^k - c)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "avx-ecm_amd"))
import pyecm
k = int(sys.argv[1]) if len(sys.argv) > 1 else 401
curves = int(sys.argv[2]) if len(sys.argv) > 2 else 200
b1 = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cc = int(sys.argv[5]) if len(sys.argv) > 5 else 1
n = (1 << k) - cc if k > 0 else (1 << -k) + 1
eng = pyecm.Engine(n, digitbits=52)
print("N = 2^%d %s %d, REDC limbs %d, special form:" % (abs(k), "-" if k > 0 else "+", cc, eng.cfg.dev_limbs), eng.special_form(), flush=True)
sig = list(range(1000, 1000 + curves))
res = {}
for on in (True, False):
    eng.set_special_form(on)
    eng.set_lanes_per_curve(lanes)
    eng.build_curves(sig)
    eng.stage1(b1)
    res[on] = (eng.save_lines(), eng.last_kernel_ms(), eng.special_form()[0])
    print("special=%s: kernel %.1f ms, %d lane(s) per curve" % (on, res[on][1], eng.lanes_per_curve()), flush=True)
same = res[True][0] == res[False][0]
print("save lines identical:", same, " speed-up %.2fx" % (res[False][1] / res[True][1]))
if not same:
    bad = [i for i, (a, b) in enumerate(zip(res[True][0], res[False][0])) if a != b]
    print("differing curves:", len(bad), bad[:10])
    print(res[True][0][bad[0]][:300]); print(res[False][0][bad[0]][:300])
eng.close()
sys.exit(0 if same else 1)
