"""The host-side logic of libgecm (expression evaluator, input preparation, big integers, tape compiler,
stage-2 plan, PAIR) under AddressSanitizer + UBSan with random and hostile inputs: tools/host_sanitize.sh.
CPU only; GPU sanitizers are not available on the pool."""
import os
import subprocess

from conftest import ROOT


def test_host_logic_is_clean_under_asan_ubsan():
    p = subprocess.run([os.path.join(ROOT, "tools", "host_sanitize.sh")], capture_output=True, text=True, timeout=600)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-3000:]
    assert "host sanitizer run complete" in out
    assert "ERROR: AddressSanitizer" not in out and "runtime error" not in out
    # counters of the reference at B1 = 1e6 (ecm.c:1849) come out of the tape compiler
    assert "tape B1=1000000 rc=0 len=2059551 adds=1980817 dups=217929" in out
