"""Stage-2 wall time on small and full batches.  usage: python tools/s2_small.py [B2] [batches...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "avx-ecm_amd"))
import pyecm
b2 = int(sys.argv[1]) if len(sys.argv) > 1 else 100000000
batches = [int(x) for x in sys.argv[2:]] or [4096, 32768, 131072]
import random
n = random.Random(415).getrandbits(415) | (1 << 414) | 1               # generic: not of the form 2^k -/+ 1
eng = pyecm.Engine(n, digitbits=52)
for b in batches:
    eng.build_curves(list(range(1000, 1000 + b)))
    eng.stage1(10000)
    t = time.perf_counter()
    eng.stage2(b2)
    nf, _ = eng.scan_factors(2)
    t = time.perf_counter() - t
    s2 = eng.stage2_stats()
    print("batch %7d  B2=%d  stage 2 %.2f s  (%d pair muls, %d adds, %d inversions)  acc[0] %x"
          % (b, b2, t, s2.paired, s2.ptadds, s2.numinv, eng.download_acc()[0]), flush=True)
eng.close()
