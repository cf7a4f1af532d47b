"""Full-batch stage-1 kernel time, one curve per lane vs two lanes per curve, for every limb count.
usage: python tools/lanes_sizes.py [B1] [batch] [nl,nl,...]   (needs a GPU)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "avx-ecm_amd"))
import pyecm  # noqa: E402

b1 = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
nls = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [8, 10, 12, 14, 15, 17, 19, 21, 23, 26, 28, 30, 32, 34, 37]
for nl in nls:
    bits = 28 * nl - 5
    import random
    n = random.Random(bits).getrandbits(bits) | (1 << (bits - 1)) | 1     # generic: not of the form 2^k -/+ 1
    eng = pyecm.Engine(n, digitbits=52)
    assert eng.cfg.dev_limbs == nl, (eng.cfg.dev_limbs, nl)
    eng.build_curves(list(range(1000, 1000 + batch)))
    row = [1e30, 1e30]
    for _ in range(2):                       # interleaved: 1, 2, 1, 2 (clock drift hits both alike)
        for lanes in (1, 2):
            eng.set_lanes_per_curve(lanes)
            eng.stage1(b1)
            row[lanes - 1] = min(row[lanes - 1], eng.last_kernel_ms())
    print("NL %2d (%4d bits) batch %d  lanes=1 %8.1f ms  lanes=2 %8.1f ms  ratio %.3f"
          % (nl, bits, batch, row[0], row[1], row[0] / row[1]), flush=True)
    eng.close()
