"""Error behaviour of the C ABI on a GPU box: misuse returns negative codes with a message, never a
crash, never a silent fallback (the reference printf+exit()s: util.c:56-59, ecm.c:969-970)."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu


def test_argument_and_state_errors():
    import pyecm
    L = pyecm.lib
    n = (1 << 127) - 1
    eng = pyecm.Engine(n * ((1 << 89) - 1))
    with pytest.raises(pyecm.GecmError, match="no curves"):
        eng.stage1(1000)                                   # nothing uploaded
    arr = (ctypes.c_uint64 * 1)(1000)
    assert L.gecm_build_curves(eng._h, arr, 0) == -2       # empty batch
    with pytest.raises(pyecm.GecmError, match="sigma"):
        eng.build_curves([5])                              # ecm.c:1564-1570: sigma >= 6
    eng.build_curves([6, 7, 8])
    with pytest.raises(pyecm.GecmError):
        eng.stage1(1)                                      # B1 < 2
    with pytest.raises(pyecm.GecmError):
        eng.stage1(10 ** 13)                               # beyond the 1e12 the ABI takes (1e9 is ten prime ranges: fine)
    with pytest.raises(pyecm.GecmError, match="prime range"):
        eng.stage1_range(3 * 10 ** 8, 3)                   # ranges 0, 1, 2 only
    with pytest.raises(pyecm.GecmError, match="stage 1"):
        eng.stage2_init()                                  # stage 2 before stage 1
    eng.stage1(100)
    with pytest.raises(pyecm.GecmError):
        eng.stage2(50)                                     # B2 <= B1
    with pytest.raises(pyecm.GecmError, match="gecm_stage2_init has not run"):
        pm = pyecm.pair_primes(100, 5000, 210, 2)
        eng.stage2_pair(pm)
    eng.stage2_init(210, 2)
    bad_v = (ctypes.c_uint32 * 1)(10 ** 6)
    bad_u = (ctypes.c_uint32 * 1)(1)
    assert L.gecm_stage2_pair(eng._h, 1, bad_v, bad_u, 1) == -2   # pair outside the window (ecm.c:2508-2511)
    assert b"invalid pair map entry" in L.gecm_last_error()
    eng.close()


def test_unsupported_sizes_and_inputs():
    import pyecm
    with pytest.raises(pyecm.GecmError, match="larger than this build supports"):
        pyecm.Engine((1 << 1100) + 1)
    with pytest.raises(pyecm.GecmError, match="odd"):
        pyecm.Engine(1 << 100)
    with pytest.raises(pyecm.GecmError):
        pyecm.Engine(12345678901234567891, digitbits=40)
    # largest supported class: 1031 bits (37 limbs of 28 bits, R >= 32 N)
    e = pyecm.Engine((1 << 1030) + 1)
    assert e.cfg.dev_limbs == 37
    e.close()


def test_tiny_bounds_and_b1_change():
    """B1 = 2, 3, 4 (empty / one-doubling tapes) and changing B1 on a live context"""
    import pyecm
    n = ((1 << 127) - 1) * ((1 << 107) - 1)
    eng = pyecm.Engine(n)
    eng.build_curves([11, 12])
    eng.stage1(2)
    x0, z0 = eng.download_points_plain()
    assert z0 == [1, 1]                                    # empty tape: Z still 1
    st = eng.stage1_stats()
    assert (st.ptadds, st.ptdups, st.tape_len) == (0, 0, 0)
    eng.stage1(3)
    assert (eng.stage1_stats().ptdups, eng.stage1_stats().ptadds) == (1, 0)
    eng.stage1(1000)
    assert eng.stage1_stats().ptadds == 1796
    eng.close()


def test_rejected_batch_leaves_no_batch_behind():
    """an input error in gecm_build_curves / gecm_upload_points must not leave the context claiming a batch
    nothing was uploaded for: either the previous batch is untouched (inputs refused before anything is
    allocated) or the context holds no batch and the phase functions answer GECM_ERR_STATE (-4)"""
    import pyecm
    L = pyecm.lib
    n = ((1 << 127) - 1) * ((1 << 107) - 1)
    eng = pyecm.Engine(n)
    eng.build_curves([11, 12, 13])
    eng.stage1(50)
    before = eng.save_lines()
    with pytest.raises(pyecm.GecmError, match="sigma"):
        eng.build_curves([11, 5, 13, 14])                  # sigma < 6 (ecm.c:1564-1570 redraws; the ABI refuses)
    eng.batch = 3
    assert eng.save_lines() == before                      # refused before the batch was replaced
    buf = ctypes.create_string_buffer(4096)
    assert L.gecm_format_save_line(eng._h, 3, buf, len(buf)) < 0
    one = eng.pack([1, 1])
    big = eng.pack([n, 1])                                 # operand not < N
    assert L.gecm_upload_points(eng._h, big, one, one, 2) == -2
    assert L.gecm_stage1(eng._h, 50) == -4                 # refused after the old batch was dropped: no batch now
    assert L.gecm_format_save_line(eng._h, 0, buf, len(buf)) < 0
    eng.build_curves([11, 12, 13])
    eng.stage1(50)                                         # and the context is usable again
    assert eng.save_lines() == before
    eng.close()
