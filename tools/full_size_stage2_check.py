"""Stage 2 at the bench's size: 131,072 curves, B2 = 1e8 on the device; the accumulators of sample lanes must equal
the oracle's (oracle/ecm_oracle.c orc_stage2), and the factor flags must match gcd(acc, N) on the host.
usage: python tools/full_size_stage2_check.py [B1] [B2] [curves] [modulus]
modulus: K1 (default: the 412-bit K1 of the reference's tests) or a product of Mersenne primes "521x127", "607x127x89", ...
for the other limb counts (no small factors)"""
import ctypes, math, os, random, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "avx-ecm_amd"))
import pyecm
b1 = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
b2 = int(sys.argv[2]) if len(sys.argv) > 2 else 100000000
curves = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
import json
K1 = next(c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "stage1.json"))) if c["name"] == "K1")
n = int(K1["save_lines"][0].split("N=0x")[1].split(";")[0], 16)      # two large prime factors: accumulators are generic
if len(sys.argv) > 4 and sys.argv[4] != "K1":
    n = 1
    for e in sys.argv[4].split("x"):
        n *= (1 << int(e)) - 1
sig = list(range(1000, 1000 + curves))
eng = pyecm.Engine(n)
print("modulus of %d bits, %d limbs on the device" % (n.bit_length(), eng.cfg.dev_limbs), flush=True)
eng.build_curves(sig)
eng.stage1(b1)
t = time.time()
eng.stage2(b2)
nf, first = eng.scan_factors(2)
print("stage 2 to B2=%d on %d curves: %.1f s, %d curves with a factor (first %s)" % (b2, curves, time.time() - t, nf, first), flush=True)
st = eng.stage2_stats()
print("D=%d U=%d pt-adds %d inversions %d pair-muls %d" % (st.D, st.U, st.ptadds, st.numinv, st.paired), flush=True)
lanes = [0, 63, 64, 4095, 65537, curves - 1]
lanes = [k for k in lanes if k < curves]
acc = eng.download_acc()
flags = [eng.curve_flag(2, k) for k in lanes]
eng.close()
L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libecm_oracle.so"))
L.orc_create.restype = ctypes.c_void_p
L.orc_create.argtypes = [ctypes.c_char_p, ctypes.c_int]
L.orc_stage2.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                         ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
c = L.orc_create(str(n).encode(), 52)
buf = ctypes.create_string_buffer(8192)
for k, f in zip(lanes, flags):
    L.orc_stage2(c, sig[k], b1, b2, st.D, st.U, buf, None, 0, None)
    assert int(buf.value, 16) == acc[k], k
    g = math.gcd(acc[k], n)
    if acc[k] != 0:                      # acc = 0 marks a failed batch inversion; its factor is in the fail record
        assert f == (1 < g < n), (k, f, g)
    print("lane %d: accumulator == oracle, gcd(acc, N) = %d, factor flag %s" % (k, g if g < n else 0, f), flush=True)
print("full-size stage-2 check passed")
