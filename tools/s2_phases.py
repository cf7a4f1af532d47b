"""Where the wall time of stage 2 goes on a small batch: init / pair map / tape + walk, phase by phase.
usage: python3 tools/s2_phases.py [curves] [B1] [B2]"""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "avx-ecm_amd"))
import pyecm
curves = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b1 = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
b2 = int(sys.argv[3]) if len(sys.argv) > 3 else 100000000
n = random.Random(415).getrandbits(415) | (1 << 414) | 1
eng = pyecm.Engine(n)
eng.build_curves(list(range(1000, 1000 + curves)))
eng.stage1(b1)
for rep in range(3):
    t0 = time.perf_counter(); eng.stage2_init(2310, 16, sync=False); t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
    p = pyecm.pair_primes(b1, b2, 2310, 16); t3 = time.perf_counter()
    eng.stage2_pair(p, sync=False); t4 = time.perf_counter(); eng.sync(); t5 = time.perf_counter()
    print("rep %d: stage2_init call %.1f ms + sync %.1f ms | pair map %.1f ms | stage2_pair call %.1f ms + sync %.1f ms" %
          (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3), flush=True)
t0 = time.perf_counter(); eng.stage2(b2); t1 = time.perf_counter()
print("gecm_stage2 whole: %.1f ms" % ((t1 - t0) * 1e3))
for _ in range(2):
    t0 = time.perf_counter(); eng.stage2(b2); t1 = time.perf_counter()
    print("gecm_stage2 again (pair map and tape kept): %.1f ms" % ((t1 - t0) * 1e3))
acc1 = eng.download_acc()
eng.stage1(b1 + 1)        # another B1: nothing kept applies
t0 = time.perf_counter(); eng.stage2_prepare(b2); t1 = time.perf_counter(); eng.stage2(b2); t2 = time.perf_counter()
print("after a new stage 1: gecm_stage2_prepare %.1f ms, gecm_stage2 %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
eng.close()
