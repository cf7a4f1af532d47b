"""Stage-1 kernel time, one curve per lane vs two lanes per curve, over batch sizes.
usage: python tools/lanes_bench.py [B1] [bits]   (needs a GPU)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "avx-ecm_amd"))
import pyecm  # noqa: E402

b1 = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 415
import random
n = random.Random(bits).getrandbits(bits) | (1 << (bits - 1)) | 1     # generic: not of the form 2^k -/+ 1
eng = pyecm.Engine(n, digitbits=52)
print("N = 2^%d-ish, NL = %d, B1 = %d, %s" % (bits, eng.cfg.dev_limbs, b1, eng.device_name()))
for batch in (1024, 4096, 16384, 32768, 49152, 65536, 98304, 131072):
    eng.build_curves(list(range(1000, 1000 + batch)))
    row = []
    for lanes in (1, 2):
        eng.set_lanes_per_curve(lanes)
        best = 1e30
        for _ in range(2):       # stage 1 runs in place: the second run continues from [k]P, same work
            eng.stage1(b1)
            best = min(best, eng.last_kernel_ms())
        row.append(best)
    print("batch %7d  lanes=1 %9.1f ms %8.0f curves/s   lanes=2 %9.1f ms %8.0f curves/s   ratio %.2f"
          % (batch, row[0], batch / row[0] * 1e3, row[1], batch / row[1] * 1e3, row[0] / row[1]), flush=True)
eng.close()
