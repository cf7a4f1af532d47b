#!/usr/bin/env python3
"""Stage-1 kernel time between 4096 and 16,384 curves for the 32-lane layout with the operand limbs by DPP (two per
move) and through the LDS crossbar, and for the eight-lane layout: the data behind gecm_dev_auto_lanes / row_a_lds
after the row changes of round 3.   usage: mid_batch_rows.py [bits ...]   (B1 = 1e5 at 415 bits, 2e4 above)"""
import os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "avx-ecm_amd"))
import pyecm

for bits in [int(x) for x in sys.argv[1:]] or [415, 623, 831, 1023]:
    n = random.Random(bits).getrandbits(bits) | (1 << (bits - 1)) | 1
    eng = pyecm.Engine(n)
    b1 = 100000 if bits < 500 else 20000
    print("%d bits (%d limbs), B1 = %d: kernel ms       32 lanes DPP   32 lanes crossbar   8 lanes   auto" % (bits, eng.cfg.dev_limbs, b1), flush=True)
    for curves in (4096, 6144, 8192, 10240, 12288, 16384):
        row = []
        for lanes, alds in ((32, "0"), (32, "1"), (8, None), (0, None)):
            if alds is None:
                os.environ.pop("GECM_ROW_ALDS", None)
            else:
                os.environ["GECM_ROW_ALDS"] = alds
            eng.set_lanes_per_curve(lanes)
            ms = []
            for _ in range(3):
                eng.build_curves(list(range(1000, 1000 + curves)))
                eng.stage1(b1)
                ms.append(eng.last_kernel_ms())
            row.append("%8.1f" % min(ms[1:]) + (" (%s)" % eng.last_kernel_name() if lanes == 0 else ""))
        print("  %6d curves   %s" % (curves, "   ".join(row)), flush=True)
