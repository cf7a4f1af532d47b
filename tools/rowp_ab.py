#!/usr/bin/env python3
"""A/B of the 32-lane stage-1 kernel's three operand-broadcast variants (GECM_ROW_ALDS = 0: DPP, 1: ds_swizzle up
front, 2: point forms in LDS read one multiply ahead) on one device, interleaved, identical save lines required.
usage: rowp_ab.py [curves] [B1] [bits ...]"""
import hashlib, os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "avx-ecm_amd"))
import pyecm

curves = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b1 = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
bits_list = [int(x) for x in sys.argv[3:]] or [415, 623, 831, 1023]
for bits in bits_list:
    n = random.Random(bits).getrandbits(bits) | (1 << (bits - 1)) | 1
    eng = pyecm.Engine(n)
    eng.set_lanes_per_curve(32)
    sig = list(range(1000, 1000 + curves))
    res = {}
    for rep in range(3):
        for mode in ("0", "2", "1"):
            os.environ["GECM_ROW_ALDS"] = mode
            eng.build_curves(sig)
            eng.stage1(b1)
            ms = eng.last_kernel_ms()
            sha = hashlib.sha256("".join(eng.save_lines()[:: max(1, curves // 256)]).encode()).hexdigest()[:12]
            res.setdefault(mode, []).append((ms, sha, eng.last_kernel_name()))
    shas = {r[1] for v in res.values() for r in v}
    for mode in ("0", "1", "2"):
        v = res[mode]
        print("%4d bits %5d curves B1=%d  mode %s %-28s kernel ms %s   %s" % (bits, curves, b1, mode, v[0][2], " ".join("%.1f" % r[0] for r in v),
              "same residues" if len(shas) == 1 else "RESIDUES DIFFER %s" % v[0][1]), flush=True)
    best0, best2 = min(r[0] for r in res["0"][1:]), min(r[0] for r in res["2"][1:])
    print("     -> LDS-prefetch / DPP = %.3f (%.1f%% %s)" % (best2 / best0, abs(1 - best2 / best0) * 100, "faster" if best2 < best0 else "SLOWER"), flush=True)
    eng.close()
