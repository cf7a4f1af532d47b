"""N>1 path on CPU: world_size-2 gloo run of the host-side curve split, the single found-flag
all-reduce and the ordered gather of save lines (pyecm/shard.py), with the oracle standing in for the
device engine.  The merged save file must equal the single-process one for any G."""
import ctypes
import json
import os
import random
import subprocess
import sys
import tempfile

import pytest

from conftest import ROOT

WORKER = r'''
import ctypes, json, os, sys
sys.path.insert(0, os.path.join(%(root)r, "avx-ecm_amd"))
sys.path.insert(0, os.path.join(%(root)r, "avx-ecm_amd", "pyecm"))
import torch.distributed as dist
import shard
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=world)
L = ctypes.CDLL(os.path.join(%(root)r, "oracle", "libecm_oracle.so"))
L.orc_create.restype = ctypes.c_void_p
L.orc_create.argtypes = [ctypes.c_char_p, ctypes.c_int]
L.orc_stage1_line.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t,
                              ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
n, total, sigma0, b1 = %(n)d, %(total)d, %(sigma0)d, %(b1)d
c = L.orc_create(str(n).encode(), 52)
lo, hi = shard.shard_bounds(total, rank, world)
lines, first = [], None
line = ctypes.create_string_buffer(8192); fac = ctypes.create_string_buffer(2048)
for k, s in zip(range(lo, hi), shard.shard_sigmas(sigma0, total, rank, world)):
    L.orc_stage1_line(c, s, b1, line, len(line), fac, len(fac), None)
    lines.append(line.value.decode())
    if fac.value and first is None:
        first = k
found = shard.allreduce_found(dist, first, total)
allx = shard.gather_lines(dist, lines)
if rank == 0:
    json.dump({"found": found, "lines": allx}, open(%(out)r, "w"))
dist.barrier()
dist.destroy_process_group()
'''


def _run(world, n, total, sigma0, b1, port):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "out.json")
        src = os.path.join(d, "w.py")
        open(src, "w").write(WORKER % dict(root=ROOT, port=port, n=n, total=total, sigma0=sigma0, b1=b1, out=out))
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, src], env=env))
        for p in procs:
            assert p.wait(timeout=300) == 0
        return json.load(open(out))


def test_shard_bounds_cover_and_order():
    sys.path.insert(0, os.path.join(ROOT, "avx-ecm_amd", "pyecm"))
    import shard
    for total in (0, 1, 7, 8, 4096, 32768 + 5):
        for world in (1, 2, 3, 8):
            segs = [shard.shard_bounds(total, r, world) for r in range(world)]
            assert segs[0][0] == 0 and segs[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(segs, segs[1:]))
            assert max(h - l for l, h in segs) - min(h - l for l, h in segs) <= 1
    assert shard.decode_found(shard.encode_found(None, 10), 10) is None
    assert shard.decode_found(max(shard.encode_found(7, 10), shard.encode_found(3, 10)), 10) == 3


def test_world2_equals_world1_save_file_and_found_flag():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], stdout=subprocess.DEVNULL)
    n = random.Random(415).getrandbits(415) | (1 << 414) | 1     # has small factors: some curves "find" them
    one = _run(1, n, 6, 1000, 300, 29611)
    two = _run(2, n, 6, 1000, 300, 29613)
    assert one["lines"] == two["lines"] and len(one["lines"]) == 6
    assert one["found"] == two["found"]
    # ragged split (7 curves over 2 ranks) with an N whose first factor shows up late or never
    k1 = 7908926676514675413083853032827063880118980193445471625562601469958414706043143581401715516956542424923236530406833110566233
    one = _run(1, k1, 7, 100, 200, 29615)
    two = _run(2, k1, 7, 100, 200, 29617)
    assert one["lines"] == two["lines"] and len(two["lines"]) == 7
    assert one["found"] == two["found"]
