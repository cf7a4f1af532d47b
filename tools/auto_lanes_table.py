import os, sys, random
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "avx-ecm_amd"))
import pyecm
for bits, db in ((415, 52), (831, 52), (1023, 32), (250, 52)):
    n = random.Random(bits).getrandbits(bits) | (1 << (bits - 1)) | 1
    eng = pyecm.Engine(n, digitbits=db)
    out = []
    for c in (1024, 4096, 7680, 8192, 10240, 12288, 12800, 14336, 16384, 24576, 65536, 131072):
        eng.build_curves(list(range(1000, 1000 + c)))
        eng.stage1(200)
        out.append("%d:%d" % (c, eng.lanes_per_curve()))
    print(bits, " ".join(out), flush=True)
    eng.close()
