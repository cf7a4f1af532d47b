"""Stage 1 modulo 2^k -/+ 1 for N | 2^k -/+ 1 (the "F-form" / "P-form" multiplies, csrc/gecm_field.hpp; the
reference's isMersenne == +1 / -1 inputs).  The kernel runs the same REDC with the same digits modulo
Mw = 2^k -/+ 1 and the host reduces modulo N, so the save lines must be byte-identical to the generic
path's for every curve."""
import pytest

pytestmark = pytest.mark.gpu


def _lines(n, sig, b1, special):
    import pyecm
    eng = pyecm.Engine(n, digitbits=52)
    avail = eng.special_form()
    eng.set_special_form(special)
    eng.build_curves(sig)
    eng.stage1(b1)
    used = eng.special_form()[0] and special
    lines = eng.save_lines()
    facs = [eng.stage1_factor(k) for k in range(len(sig))]
    eng.close()
    return lines, facs, avail, used


@pytest.mark.parametrize("k,cof", [(170, 1), (251, 503 * 54217), (401, 1), (521, 1), (607, 1), (701, 1), (929, 1), (1009, 1),
                                   (-171, 3), (-256, 1), (-401, 3), (-523, 3), (-600, 1), (-809, 3), (-1024, 1)])
def test_special_form_stage1_equals_generic_path(k, cof):
    n = (((1 << k) - 1) if k > 0 else ((1 << -k) + 1)) // cof
    assert (((1 << abs(k)) - (1 if k > 0 else -1)) % cof) == 0
    sig = list(range(2000, 2070))
    a, fa, avail, used = _lines(n, sig, 3000, True)
    assert avail[0] and avail[1] == k and used, (avail, used)
    b, fb, _, used_b = _lines(n, sig, 3000, False)
    assert not used_b
    assert a == b and fa == fb


def test_special_form_is_not_used_when_it_cannot_be():
    import pyecm
    # k = 150: bit k sits lower than the kernel's generic top limbs reach -> plain REDC
    eng = pyecm.Engine((1 << 150) - 1)
    assert eng.special_form() == (False, 0, 0)
    eng.close()
    # a generic N has no such form
    eng = pyecm.Engine((1 << 300) + 157)
    assert eng.special_form() == (False, 0, 0)
    eng.close()
    # a small cofactor of a large 2^k - 1: REDC on the cofactor is the cheaper multiply (as main.c:505-516 decides too)
    eng = pyecm.Engine(((1 << 89) - 1) * ((1 << 107) - 1) * 13367)      # divides 2^9523-1: k beyond the detection loop
    assert eng.special_form()[0] is False
    eng.close()


def test_special_form_survives_a_second_stage1_and_stage2():
    """stage 1 twice on the same batch continues from [k]P in both paths; stage 2 then runs modulo N"""
    import pyecm
    n = (1 << 401) - 1
    sig = list(range(3000, 3064))
    out = {}
    for special in (True, False):
        eng = pyecm.Engine(n)
        eng.set_special_form(special)
        eng.build_curves(sig)
        eng.stage1(500)
        eng.stage1(500)
        lines = eng.save_lines()
        eng.stage2(20000)
        out[special] = (lines, eng.download_acc())
        eng.close()
    assert out[True] == out[False]


@pytest.mark.parametrize("k", [401, -523])
def test_special_form_through_upload_points(k):
    """phase 0 by the caller (gecm_upload_points, operands in the reference's Montgomery radix): the points
    are lifted to the 2^k -/+ 1 context as well, and come back in the reference's radix modulo N"""
    import random
    import pyecm
    n = (1 << k) - 1 if k > 0 else ((1 << -k) + 1) // 3
    rng = random.Random(k)
    xs = [rng.randrange(1, n) for _ in range(40)]
    ss = [rng.randrange(1, n) for _ in range(40)]
    out = {}
    for special in (True, False):
        eng = pyecm.Engine(n)
        R = 1 << eng.cfg.maxbits
        eng.set_special_form(special)
        eng.upload_points([x * R % n for x in xs], [R % n] * 40, [s * R % n for s in ss])
        eng.stage1(700)
        assert eng.special_form()[0] == special
        out[special] = (eng.download_points(), eng.download_points_plain())
        eng.close()
    assert out[True] == out[False]
    assert all(v < n for v in out[True][0][0] + out[True][0][1])
