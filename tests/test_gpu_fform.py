"""Stage 1 modulo 2^k -/+ 1 for N | 2^k -/+ 1 (the "F-form" / "P-form" multiplies, csrc/gecm_field.hpp; the
reference's isMersenne == +1 / -1 inputs).  The kernel runs the same REDC with the same digits modulo
Mw = 2^k -/+ 1 and the host reduces modulo N, so the save lines must be byte-identical to the generic
path's for every curve."""
import pytest

pytestmark = pytest.mark.gpu


def _lines(n, sig, b1, special, lanes=2):
    import pyecm
    eng = pyecm.Engine(n, digitbits=52)
    avail = eng.special_form()
    eng.set_special_form(special)
    eng.set_lanes_per_curve(lanes)          # explicit: left to itself a batch this small takes the eight-lane generic kernel
    eng.build_curves(sig)
    eng.stage1(b1)
    used = eng.special_form_used()
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
        eng.set_lanes_per_curve(1)
        eng.build_curves(sig)
        eng.stage1(500)
        eng.stage1(500)
        assert eng.special_form_used() == special
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
        eng.set_lanes_per_curve(2)
        eng.upload_points([x * R % n for x in xs], [R % n] * 40, [s * R % n for s in ss])
        eng.stage1(700)
        assert eng.special_form_used() == special
        out[special] = (eng.download_points(), eng.download_points_plain())
        eng.close()
    assert out[True] == out[False]
    assert all(v < n for v in out[True][0][0] + out[True][0][1])


BUILT = [8, 10, 12, 14, 15, 17, 19, 21, 23, 26, 28, 30, 32, 34, 37]


def _edge_cases():
    out = []
    for i, nl in enumerate(BUILT):
        prev = BUILT[i - 1] if i else 7
        for sign in (+1, -1):
            extra = 0 if sign > 0 else 1                      # 2^k + 1 has k+1 bits
            hi = 28 * nl - 5 - extra                          # largest k this limb count takes
            lo = 28 * prev - 4 - extra                        # smallest k that needs it: bit k sits in limb prev-1,
            out.append((nl, sign * hi))                       # the lowest of the top limbs the kernel reads from n[]
            out.append((nl, sign * lo))
    return out


@pytest.mark.parametrize("nl,k", _edge_cases(), ids=["nl%d_%+d" % c for c in _edge_cases()])
def test_special_form_at_both_ends_of_every_limb_count(nl, k):
    """for each built limb count: the largest and the smallest exponent that maps to it, 2^k - 1 and 2^k + 1"""
    import pyecm
    n = (1 << k) - 1 if k > 0 else (1 << -k) + 1
    sig = list(range(4000, 4033))
    eng = pyecm.Engine(n)
    avail, kk, limbs = eng.special_form()
    assert avail and kk == k and limbs == nl, (avail, kk, limbs)
    out = []
    for special in (True, False):
        for lanes in (1, 2):
            eng.set_special_form(special)
            eng.set_lanes_per_curve(lanes)
            eng.build_curves(sig)
            eng.stage1(1200)
            assert eng.special_form_used() == special
            out.append(eng.save_lines())
    eng.close()
    assert out[0] == out[1] == out[2] == out[3]


def test_small_batches_prefer_the_many_lane_generic_kernel():
    """left to itself (lanes = 0) a small batch of a 2^k - 1 cofactor runs the 32-lane generic kernel, a large
    one the special multiply; same save lines either way"""
    import pyecm
    n = (1 << 401) - 1
    eng = pyecm.Engine(n)
    assert eng.special_form() == (True, 401, 15)
    eng.build_curves(list(range(7000, 7032)))
    eng.stage1(400)
    assert eng.lanes_per_curve() == 32 and not eng.special_form_used()
    small = eng.save_lines()
    eng.build_curves(list(range(7000, 7000 + 20000)))
    eng.stage1(400)
    assert eng.lanes_per_curve() == 2 and eng.special_form_used()
    assert eng.save_lines()[:32] == small
    eng.close()


def test_results_can_be_read_without_an_explicit_sync():
    """gecm_stage1 is asynchronous; every consumer of its result (save lines, factor scan, stage 2) waits for it
    and, for a special-form launch, first brings X and Z back modulo N"""
    import pyecm
    n = (1 << 401) - 1
    sig = list(range(9000, 9040))
    ref = None
    for special, consumer in ((False, "lines"), (True, "lines"), (True, "scan"), (True, "stage2")):
        eng = pyecm.Engine(n)
        eng.set_special_form(special)
        eng.set_lanes_per_curve(2)
        eng.build_curves(sig)
        eng.stage1(2000, sync=False)
        if consumer == "scan":
            eng.scan_factors(1)
        elif consumer == "stage2":
            eng.stage2_init()
        lines = eng.save_lines()
        assert eng.special_form_used() == special
        eng.close()
        if ref is None:
            ref = lines
        assert lines == ref
