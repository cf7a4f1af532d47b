#!/usr/bin/env python3
"""Check of csrc/gecm_row.hpp's multiply against Python integers (run on the GPU box):
   python3 tools/row_mul_check.py [nq] [rho1] [iters]
builds tools/row_mul_check (hipcc) if needed, feeds it random operands with limbs over the whole allowed
range, and verifies value (a*b/R' mod modulus), limb range and value bound of every result."""
import os, random, struct, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rho1 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iters = sys.argv[3] if len(sys.argv) > 3 else "20000"
exe = os.path.join(ROOT, "tools", "row_mul_check")
if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(exe + ".hip"):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", exe + ".hip", "-o", exe])
rng = random.Random(7 + nq * 2 + rho1)
L = 16 * nq
Rp = 1 << (28 * L)
nbits = 28 * L - 33 if rho1 else 28 * L - 5
N = rng.getrandbits(nbits) | (1 << (nbits - 1)) | 1
if rho1:
    m = (-pow(N, -1, 1 << 28)) % (1 << 28)
    M = m * N
    assert M % (1 << 28) == (1 << 28) - 1
else:
    M = N
rho = (-pow(M, -1, 1 << 28)) % (1 << 28)
count = 4096
def limbs_u(x):
    return [(x >> (28 * j)) & 0xFFFFFFF for j in range(L)]
def val(l):
    return sum(v << (28 * j) for j, v in enumerate(l))
ops = []
for k in range(count):
    kind = k % 4
    # operands of the kernel are sums or differences of two multiply outputs: |limb| <= 2^28 + 16.  One limb per
    # lane has room for twice that; with several limbs per lane a slot collects nq rows before it is folded
    lim = ((1 << 29) - 1 if nq == 1 else (1 << 28) + 16) if kind < 3 else (1 << 27)
    a = [rng.randint(-lim, lim) for _ in range(L)]
    b = [rng.randint(-lim, lim) for _ in range(L)]
    if kind == 1:      # extremes
        a = [rng.choice((-lim, lim)) for _ in range(L)]
        b = [rng.choice((-lim, lim)) for _ in range(L)]
    if kind == 2:      # canonical unsigned inputs
        a = limbs_u(rng.randrange(M)); b = limbs_u(rng.randrange(M))
    # keep |value| < 4*M so that the result bound is meaningful: scale the top limb down
    for v in (a, b):
        top = 4 * M >> (28 * (L - 1))
        v[L - 1] = rng.randint(-top // 2, top // 2) if kind != 2 else v[L - 1]
    ops.append((a, b))
with open("/tmp/row_in.bin", "wb") as f:
    f.write(struct.pack("<4I", nq, rho1, rho, count))
    f.write(struct.pack("<%dI" % L, *limbs_u(M)))
    for a, b in ops:
        f.write(struct.pack("<%di" % L, *a)); f.write(struct.pack("<%di" % L, *b))
out = subprocess.run([exe, "/tmp/row_in.bin", "/tmp/row_out.bin", iters], capture_output=True, text=True)
print(out.stdout, out.stderr)
assert out.returncode == 0
res = open("/tmp/row_out.bin", "rb").read()
Rinv = pow(Rp, -1, M)
bad = 0
for k, (a, b) in enumerate(ops):
    r = list(struct.unpack_from("<%di" % L, res, k * 4 * L))
    va, vb, vr = val(a), val(b), val(r)
    ok = (vr - va * vb * Rinv) % M == 0
    ok &= all(abs(x) <= (1 << 27) + 8 for x in r[:-1])
    ok &= abs(vr) <= abs(va * vb) // Rp + M + 1
    if not ok:
        bad += 1
        if bad < 4:
            print("MISMATCH case", k, "kind", k % 4, [hex(x) for x in r][:4])
print("row_mul_check nq=%d rho1=%d: %d cases, %d bad" % (nq, rho1, count, bad))
sys.exit(1 if bad else 0)
