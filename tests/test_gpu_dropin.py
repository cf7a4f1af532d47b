"""The drop-in claim at the operator seam, exercised with the reference itself.

oracle/_ref/avx-ecm-52-gecm is the REFERENCE program (its own main, vececm, prac, vec_add, vec_duplicate,
stage 2 — compiled from /root/reference in the build container by `make -C oracle refgpu`) with the five
operator pointers of avx_ecm.h:205-209 bound to libgecm's gecm_vec*mod entry points through
oracle/ref_gecm_binding.c (the INTEGRATION.md §2 binding).  Every field operation of the run is a GPU
kernel launch; the save file must be byte-identical to the one the pure AVX-512 reference wrote.

One launch per operator on 8 lanes is a PCIe round trip, so the cases are small."""
import hashlib
import json
import os
import re
import subprocess
import tempfile

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "oracle", "_ref", "avx-ecm-52-gecm")
S1 = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "stage1.json")))}


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/avx-ecm-52-gecm not built (needs /root/reference at build time)")
@pytest.mark.parametrize("name", ["n64_b1_500", "K1N_two_full_batches_b1_500", "n415_b1_1000"])
def test_reference_driver_with_gpu_operators_writes_the_reference_save_file(name):
    c = S1[name]
    with tempfile.TemporaryDirectory() as d:
        p = subprocess.run([EXE, c["N"], str(c["curves"]), str(c["B1"]), "1", str(c["B2"]), str(c["sigma0"])],
                           cwd=d, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        save = open(os.path.join(d, "save_b1.txt")).read()
        res = [l for l in open(os.path.join(d, "ecm_results.txt")).read().splitlines() if l.strip()] \
            if os.path.exists(os.path.join(d, "ecm_results.txt")) else []
    m = re.search(r"gecm binding: (\d+) operator calls served on (.*)", p.stderr)
    assert m, p.stderr[-2000:]
    # every modular operation of the run went through the library: at least 6 mul/sqr per point-add
    assert int(m.group(1)) >= 6 * c["ptadds"] * (c["curves"] // 8)
    assert hashlib.sha256(save.encode()).hexdigest() == c["save_sha256"]
    assert save.splitlines() == c["save_lines"]
    if c["curves"] == 8:
        assert res == c["results_lines"]
