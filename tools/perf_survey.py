"""Performance survey on one MI355X: stage-1 rate per operand size, and stage-2 time (not the bench)."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "avx-ecm_amd"))
import pyecm

PEAK = 256 * 4 * 16 * 2.4e9


def mads(adds, dups, nl):
    mul = 4 * adds + 3 * dups
    sqr = 2 * adds + 2 * dups
    return mul * (2 * nl * nl + nl) + sqr * (nl * (nl + 1) // 2 + nl * nl + nl)


what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("all", "stage1"):
    for bits, curves, b1 in ((415, 131072, 100000), (623, 131072, 100000), (831, 131072, 100000), (1023, 131072, 100000)):
        n = random.Random(bits).getrandbits(bits) | (1 << (bits - 1)) | 1
        eng = pyecm.Engine(n)
        t = time.time(); eng.build_curves(list(range(1000, 1000 + curves))); tb = time.time() - t
        eng.stage1(b1)
        st = eng.stage1_stats(); ms = eng.last_kernel_ms(); nl = eng.cfg.dev_limbs
        w = mads(st.ptadds, st.ptdups, nl) * curves
        print("stage1 bits=%d NL=%d curves=%d B1=%d: kernel %.1f ms, %.1f curves/s (B1=1e6-equivalent %.1f), %.2f Tmad/s = %.3f of peak; host curve build %.2f s"
              % (bits, nl, curves, b1, ms, curves / ms * 1e3, curves / ms * 1e3 * mads(195448, 23269, nl) / mads(1980817, 217929, nl), w / ms / 1e9, w / ms * 1e3 / PEAK, tb), flush=True)
        eng.close()
if what in ("all", "stage2"):
    bits, curves, b1, b2 = 415, 32768, 100000, 10000000
    n = random.Random(bits).getrandbits(bits) | (1 << (bits - 1)) | 1
    eng = pyecm.Engine(n)
    eng.build_curves(list(range(1000, 1000 + curves)))
    eng.stage1(b1)
    t = time.time(); eng.stage2_init(0, 0); ti = time.time() - t; msi = eng.last_kernel_ms()
    st = eng.stage2_stats()
    pm = pyecm.pair_primes(b1, b2, st.D, st.U)
    t = time.time(); eng.stage2_pair(pm); tp = time.time() - t; msp = eng.last_kernel_ms()
    st = eng.stage2_stats()
    mulmods = 6 * st.ptadds + 4 * (7683 + 64 + (st.numinv - 2) * 32) + st.paired
    print("stage2 bits=%d curves=%d B1=%d B2=%d D=%d U=%d: init kernel %.1f ms, pair kernel %.1f ms (%d pairs, %d pt-adds, %d inv (device %d)); ~%.2f M mulmods/curve -> %.2f Tmad/s-equivalent"
          % (bits, curves, b1, b2, st.D, st.U, msi, msp, st.paired, st.ptadds, st.numinv, st.device_inversions, mulmods / 1e6,
             mulmods * 465 * curves / ((msi + msp) * 1e-3) / 1e12), flush=True)
    eng.close()
