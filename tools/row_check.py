"""32 lanes per curve (csrc/gecm_row.hpp) against the other layouts: same save lines, kernel times over batch sizes.
usage: python tools/row_check.py [B1] [bits] [batch,batch,...] [other-lanes,...]"""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "avx-ecm_amd"))
import pyecm
b1 = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 415
n = random.Random(bits).getrandbits(bits) | (1 << (bits - 1)) | 1
eng = pyecm.Engine(n, digitbits=52 if bits < 1000 else 32)
ok = True
batches = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [8, 70, 1024, 4096, 8192, 16384]
others = [int(x) for x in sys.argv[4].split(',')] if len(sys.argv) > 4 else [8]
for batch in batches:
    sig = list(range(1000, 1000 + batch))
    res = {}
    for lanes in others + [32]:
        eng.set_lanes_per_curve(lanes)
        eng.build_curves(sig)
        eng.stage1(b1)
        res[lanes] = (eng.save_lines() if batch <= 4096 else None, eng.last_kernel_ms())
    same = all(res[l][0] == res[32][0] for l in others)
    ok &= same
    print("bits %d batch %6d: " % (bits, batch) + "  ".join("lanes=%d %9.1f ms" % (l, res[l][1]) for l in others + [32])
          + "  ratio %.2f  identical=%s" % (res[others[0]][1] / res[32][1], same), flush=True)
    if not same:
        a0, b0 = res[others[0]][0], res[32][0]
        bad = [i for i, (a, b) in enumerate(zip(a0, b0)) if a != b]
        print("  differing curves:", len(bad), bad[:8])
        print("  ", a0[bad[0]][:300]); print("  ", b0[bad[0]][:300])
        break
eng.close()
sys.exit(0 if ok else 1)
