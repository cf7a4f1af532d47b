"""Eight lanes per curve against two lanes per curve: same save lines, kernel times over batch sizes.
usage: python tools/quad_check.py [B1] [bits] [batch,batch,...]"""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "avx-ecm_amd"))
import pyecm
b1 = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 415
n = random.Random(bits).getrandbits(bits) | (1 << (bits - 1)) | 1
eng = pyecm.Engine(n)
ok = True
batches = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [8, 70, 1024, 4096, 8192, 12288, 16384]
for batch in batches:
    sig = list(range(1000, 1000 + batch))
    res = {}
    for lanes in (2, 8):
        eng.set_lanes_per_curve(lanes)
        eng.build_curves(sig)
        eng.stage1(b1)
        res[lanes] = (eng.save_lines() if batch <= 4096 else None, eng.last_kernel_ms())
    same = res[2][0] == res[8][0]
    ok &= same
    print("batch %6d: lanes=2 %9.1f ms  lanes=8 %9.1f ms  ratio %.2f  identical=%s" % (batch, res[2][1], res[8][1], res[2][1] / res[8][1], same), flush=True)
    if not same:
        bad = [i for i, (a, b) in enumerate(zip(res[2][0], res[8][0])) if a != b]
        print("  differing curves:", len(bad), bad[:8])
        print("  ", res[2][0][bad[0]][:200]); print("  ", res[8][0][bad[0]][:200])
        break
eng.close()
sys.exit(0 if ok else 1)
