"""time one 1e8 prime range on a handful of curves per lane layout (decides what the library picks for tiny batches)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "avx-ecm_amd"))
import pyecm
n = 16674785985932905097902908042144874074120022918183327030138969
b1 = int(sys.argv[1]) if len(sys.argv) > 1 else 3000000
for lanes in (32, 8, 2):
    eng = pyecm.Engine(n)
    eng.set_lanes_per_curve(lanes)
    eng.build_curves(list(range(1000, 1008)))
    t = time.time()
    eng.stage1(b1)
    print("lanes", lanes, "B1", b1, "wall %.2f s" % (time.time() - t), "kernel %.1f ms" % eng.last_kernel_ms(), eng.save_line(0)[:100], flush=True)
    eng.close()
