"""One stage-1 launch on a small batch, for profiling: python3 tools/run_small.py [curves] [B1] [lanes]"""
import os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "avx-ecm_amd"))
import pyecm
curves = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b1 = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = random.Random(415).getrandbits(415) | (1 << 414) | 1
eng = pyecm.Engine(n)
eng.set_lanes_per_curve(lanes)
eng.build_curves(list(range(1000, 1000 + curves)))
eng.stage1(b1)
print("curves %d B1 %d lanes per curve %d kernel %.1f ms" % (curves, b1, eng.lanes_per_curve(), eng.last_kernel_ms()))
eng.close()
