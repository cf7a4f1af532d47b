"""Is the stage-2 pair walk sensitive to the size of the baby-step table?  Same D, different U (table entries per
curve), full batch: kernel time of one gecm_stage2_pair range per pair multiply.
usage: python tools/s2_table_size.py [B2] [curves]"""
import os, sys, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "avx-ecm_amd"))
import pyecm
b2 = int(sys.argv[1]) if len(sys.argv) > 1 else 30000000
curves = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
n = random.Random(415).getrandbits(415) | (1 << 414) | 1
eng = pyecm.Engine(n)
eng.build_curves(list(range(1000, 1000 + curves)))
b1 = 100000
eng.stage1(b1)
for U in (16, 8, 4, 2):
    eng.stage2_init(2310, U)
    t_init = eng.last_kernel_ms()
    pm = pyecm.pair_primes(b1, b2, 2310, U)
    eng.stage2_pair(pm)
    ms = eng.last_kernel_ms()
    st = eng.stage2_stats()
    print("U=%2d: table init %.0f ms; range kernels %.0f ms for %d pair muls + %d adds -> %.3f us per pair mul (all kernels)"
          % (U, t_init, ms, st.paired, st.ptadds, ms * 1e3 / st.paired), flush=True)
eng.close()
