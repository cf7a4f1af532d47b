"""Stage 2 with K sub-sequences per curve (GECM_S2_SUBSEQ) against the plain chain: accumulators must be identical.
usage: python tools/s2_subseq_check.py [batch] [B1] [B2] [D] [U]"""
import os, sys, json, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "avx-ecm_amd"))
import pyecm
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
K1 = next(c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "stage1.json"))) if c["name"] == "K1")
n = int(K1["save_lines"][0].split("N=0x")[1].split(";")[0], 16)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 200
b1 = int(sys.argv[2]) if len(sys.argv) > 2 else 200
b2 = int(sys.argv[3]) if len(sys.argv) > 3 else 60000
D = int(sys.argv[4]) if len(sys.argv) > 4 else 210
U = int(sys.argv[5]) if len(sys.argv) > 5 else 4
sig = list(range(5000, 5000 + batch))
ref = None
ok = True
for K in (1, 2, 4, 8, 16, 32):
    os.environ["GECM_S2_SUBSEQ"] = str(K)
    eng = pyecm.Engine(n)
    eng.build_curves(sig)
    eng.stage1(b1)
    t = time.time()
    eng.stage2(b2, D, U)
    acc = eng.download_acc()
    dt = time.time() - t
    eng.close()
    if ref is None:
        ref = acc
    bad = [i for i, (a, b) in enumerate(zip(acc, ref)) if a != b]
    print("K=%2d: %.3f s, %d of %d accumulators differ from K=1 %s" % (K, dt, len(bad), batch, bad[:6]), flush=True)
    ok &= not bad
sys.exit(0 if ok else 1)
