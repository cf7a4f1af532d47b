"""BASELINE-size cross-check: 131,072 curves at B1 = 1e6 through both lane layouts; the save lines of all
curves must agree (sha256 of the whole file), and a sample of lanes must equal the oracle's lines.
usage: python tools/full_size_crosscheck.py [bits] [B1] [curves] [lanes,lanes,...]"""
import ctypes, hashlib, os, random, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "avx-ecm_amd"))
import pyecm
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 415
b1 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
curves = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
layouts = [int(x) for x in sys.argv[4].split(',')] if len(sys.argv) > 4 else [1, 2]
n = random.Random(bits).getrandbits(bits) | (1 << (bits - 1)) | 1
sig = list(range(1000, 1000 + curves))
eng = pyecm.Engine(n)
sha = {}
keep = {}
for lanes in layouts:
    eng.set_lanes_per_curve(lanes)
    eng.build_curves(sig)
    t = time.time()
    eng.stage1(b1)
    lines = eng.save_lines()
    sha[lanes] = hashlib.sha256("".join(lines).encode()).hexdigest()
    keep[lanes] = lines
    print("lanes=%d: kernel %.1f ms, %d lines, sha256 %s" % (lanes, eng.last_kernel_ms(), len(lines), sha[lanes]), flush=True)
eng.close()
assert len(set(sha.values())) == 1, "layouts disagree"
L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libecm_oracle.so"))
L.orc_create.restype = ctypes.c_void_p
L.orc_create.argtypes = [ctypes.c_char_p, ctypes.c_int]
L.orc_stage1_line.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t,
                              ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
c = L.orc_create(str(n).encode(), 52)
buf = ctypes.create_string_buffer(16384)
for k in (0, 63, 64, 4095, 65537, curves - 1):
    if k >= curves:
        continue
    L.orc_stage1_line(c, sig[k], b1, buf, len(buf), None, 0, None)
    assert buf.value.decode() == keep[layouts[0]][k], k
    print("lane %d == oracle" % k, flush=True)
print("full-size cross-check passed: %d curves, B1=%d, %d-bit N" % (curves, b1, bits))
