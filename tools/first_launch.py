"""The first large launch of a process against later ones: python3 tools/first_launch.py [curves] [B1] [lanes] [warm]
warm = 0: time the first stage-1 launch of the process;  warm = B: run a stage 1 to B1 = B first (a short launch), then time;
prints the kernel times of three timed launches in a row."""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "avx-ecm_amd"))
import pyecm
curves = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b1 = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 0
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 0
sleep = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
bits = int(os.environ.get("BITS", "415"))                   # BITS=831, 1023 (32-bit reference limbs): the other size classes
n = random.Random(bits).getrandbits(bits) | (1 << (bits - 1)) | 1
eng = pyecm.Engine(n, digitbits=32 if bits > 1000 else 52)
eng.set_lanes_per_curve(lanes)
sig = list(range(1000, 1000 + curves))
eng.build_curves(sig)
wms = None
if warm:
    eng.stage1(warm)
    wms = eng.last_kernel_ms()
    eng.build_curves(sig)
if sleep:
    time.sleep(sleep)
ms = []
for _ in range(3):
    eng.stage1(b1)
    ms.append(eng.last_kernel_ms())
    eng.build_curves(sig)
print("bits %d lib %s curves %d B1 %d lanes %d warm %s sleep %.1f: %s ms" % (bits, os.path.basename(os.environ.get("GECM_LIB", "main")), curves, b1, eng.lanes_per_curve(),
      ("B1=%d (%.1f ms)" % (warm, wms)) if warm else "none", sleep, " ".join("%.1f" % m for m in ms)), flush=True)
eng.close()
