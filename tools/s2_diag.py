import os, sys, ctypes, random, subprocess, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "avx-ecm_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import pyecm
    from test_gpu_sizes import CASES
    name, n, nl = [c for c in CASES if c[2] == int(sys.argv[2])][0]
    rng = random.Random(nl)
    sig = [rng.randrange(6, 1 << 63) for _ in range(65)]
    b1, b2, D, U = 400, int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    eng = pyecm.Engine(n); eng.build_curves(sig); eng.stage1(b1); eng.stage2(b2, D, U); acc = eng.download_acc(); FACS = {k: eng.stage2_factor(k) for k in (0, 1, 64)}; eng.close()
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libecm_oracle.so"))
    L.orc_create.restype = ctypes.c_void_p; L.orc_create.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.orc_stage2.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
    c = L.orc_create(str(n).encode(), 52); acch = ctypes.create_string_buffer(8192)
    import math
    eng2 = None
    fac = ctypes.create_string_buffer(4096)
    for k in (0, 1, 64):
        found = L.orc_stage2(c, sig[k], b1, b2, D, U, acch, fac, len(fac), None)
        oa = int(acch.value, 16)
        print("k", k, "equal", oa == acc[k], "orc found", found, fac.value.decode()[:40], "gcd(orc acc,N)", math.gcd(oa, n) if math.gcd(oa, n) < 10**30 else "big",
              "gcd(dev acc,N)", math.gcd(acc[k], n) if math.gcd(acc[k], n) < 10**30 else "big", "dev factor", FACS[k])
else:
    for nl in (32,):
        for env in ({}, {"GECM_S2_SUBSEQ": "1"}):
            for (b2, D, U) in ((400000, 2310, 8), (30000, 385, 2)):
                e = dict(os.environ); e.update(env)
                r = subprocess.run([sys.executable, __file__, "child", str(nl), str(b2), str(D), str(U)], env=e, capture_output=True, text=True)
                print(nl, env, (b2, D, U), r.stdout.strip() or r.stderr.strip()[-300:], flush=True)
