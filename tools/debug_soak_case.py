import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "avx-ecm_amd"))
import pyecm
n = 1562582277569665971075393519297452663206766820364091277895193160794817586493056690102396454782254530005271337632771673839228338391936863723921683087062957223268339396209838074545385906765490591968698971492357749390388905
b1, prange, digitbits = 2013, 503, 52
sig = [667, 1086180218, 2390745089]
L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libecm_oracle.so"))
L.orc_create.restype = ctypes.c_void_p
L.orc_create.argtypes = [ctypes.c_char_p, ctypes.c_int]
L.orc_stage1_ranges_line.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                     ctypes.c_int, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                                     ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]
hook = pyecm.lib.gecm_plan_set_prime_range_for_tests
hook.argtypes = [ctypes.c_uint64]
hook.restype = None
o = L.orc_create(str(n).encode(), digitbits)
buf = ctypes.create_string_buffer(16384)
def orc(s, pr, stop):
    cnt = (ctypes.c_uint64 * 3)()
    L.orc_stage1_ranges_line(o, s, b1, b1, pr if pr else 100000000, stop, b1, buf, len(buf), None, 0, cnt, None)
    return buf.value.decode(), list(cnt)
print("bits", n.bit_length())
for pr in (0, 503):
    hook(pr)
    nr = pyecm.stage1_ranges(b1)
    for lanes in (1, 2, 8, 32):
        eng = pyecm.Engine(n, digitbits=digitbits)
        eng.set_lanes_per_curve(lanes)
        eng.build_curves(sig)
        for r in range(nr):
            eng.stage1_range(b1, r)
            st = eng.stage1_stats()
            got = [eng.save_line(k) for k in range(3)]
            want = [orc(s, pr, r + 1) for s in sig]
            ok = [g == w[0] for g, w in zip(got, want)]
            print("prange", pr, "lanes", lanes, "dev_limbs", eng.cfg.dev_limbs, "range", r, "of", nr, "ok", ok, "counters", (st.ptadds, st.ptdups, st.last_prime), want[0][1], flush=True)
        eng.close()
