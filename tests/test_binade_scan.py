"""The integer "binade scan" reproduces a sequential fp64 sum of non-negative addends bit for bit
(model of the parallel exact seed-row chain; see tests/binade_scan_model.py)."""
import random
import struct

from tests import binade_scan_model as m
from tests.binade_scan_model import addend_func, chain_sum, bits, from_bits


def seq_sum(xs):
    s = 0.0
    for x in xs:
        s = s + x
    return s


def test_random_mixed_magnitudes():
    rng = random.Random(1234)
    for trial in range(300):
        n = rng.randint(1, 700)
        lo, hi = rng.choice([(-40, 10), (-5, 5), (-1074, -1000), (-60, -50), (0, 30)])
        xs = []
        for _ in range(n):
            r = rng.random()
            if r < 0.15:
                xs.append(0.0)
            elif r < 0.25:
                xs.append(float(2 ** rng.randint(max(lo, -1074), hi)))      # exact powers: ties and exact hits
            else:
                xs.append(rng.random() * 2.0 ** rng.randint(lo, hi))
        got, _ = chain_sum(xs, block=rng.choice([1, 7, 64, 256]))
        assert bits(got) == bits(seq_sum(xs)), (trial, got, seq_sum(xs))


def test_half_way_addends_follow_parity():
    # s = 2^52 + m (ulp 1): adding x.5 must round to even, which depends on the running parity
    base = float(2 ** 52)
    xs = [base] + [0.5] * 50 + [1.5] * 50 + [2.5, 0.5, 3.5] * 20
    got, fb = chain_sum(xs, block=64)
    assert bits(got) == bits(seq_sum(xs))
    xs = [base + 1.0] + [0.5, 1.5] * 100
    got, fb = chain_sum(xs, block=32)
    assert bits(got) == bits(seq_sum(xs))


def test_rank_like_distribution():
    # one big value then a long tail of small ones, the shape of an RWR restart chain
    rng = random.Random(7)
    n = 20000
    xs = [6.0e6 * 0.15] + [rng.expovariate(1.0) * 0.15 * rng.choice([1e-9, 1e-3, 1.0]) for _ in range(n)]
    got, fb = chain_sum(xs, block=256)
    assert bits(got) == bits(seq_sum(xs))
    assert fb < 40
    xs = xs[::-1]
    got, fb = chain_sum(xs, block=256)
    assert bits(got) == bits(seq_sum(xs))


def test_addend_func_edges():
    eb = 1023 + 10
    assert addend_func(0.0, eb) == (0, 0)
    assert addend_func(2.0 ** 11, eb)[0] >= 1 << 53            # larger than the binade: forces the fall-back
    u = 2.0 ** (10 - 52)
    assert addend_func(u, eb) == (1, 1)
    assert addend_func(u / 2, eb) == (0, 1)                      # exact half: to even
    assert addend_func(u * 1.5, eb) == (2, 1)
    assert addend_func(u * 0.75, eb) == (1, 1)
    assert addend_func(5e-324, eb) == (0, 0)


def test_split_blocks_are_exact_or_refused():
    """chain_scan.hip's CS_SPLIT blocks: located with an APPROXIMATE incoming sum, applied to the EXACT one.  Whenever the
    carry's checks pass the result is the sequential sum bit for bit; a perturbed prediction may only make them fail."""
    rng = random.Random(20260105)
    accepted = refused = 0
    for trial in range(400):
        n_front = rng.randrange(50, 400)
        front = [rng.random() * 1e-3 for _ in range(n_front)]
        s = seq_sum(front)
        # a block that takes the sum across exactly one binade boundary
        eb = (m.bits(s) >> 52) & 0x7FF
        top = m.from_bits((eb + 1) << 52)
        need = top - s
        rows = rng.choice([64, 256, 1024])
        block = [need * 1.3 / rows * (0.5 + rng.random()) for _ in range(rows)]
        if trial % 7 == 0:                                   # half-way cases: addends that are multiples of half an ulp
            u = m.from_bits((eb - 52) << 52) if eb > 52 else 0.0
            block = [round(a / (u / 2)) * (u / 2) if u else a for a in block]
        want = s
        for a in block:
            want = want + a
        if ((m.bits(want) >> 52) & 0x7FF) != eb + 1:
            continue
        # the prediction: exact, or off by a few thousand ulps either way, or grossly off
        err = rng.choice([0.0, 1e-13, -1e-13, 3e-12, -3e-12, 1e-6, -1e-6, 2e-3, -2e-3, 3e-2, -3e-2])
        pre = s * (1.0 + err)
        if ((m.bits(pre) >> 52) & 0x7FF) != eb:
            continue
        sp = m.split_block(pre, block, run=rng.choice([4, 8]))     # (4: the chain's blocks, 8: the hub rows' segments -- spmv.hip)
        got = m.apply_split(s, sp)
        if got is None:
            refused += 1
            assert abs(err) > 0 or sp is None
        else:
            accepted += 1
            assert m.bits(got) == m.bits(want), (trial, err)
    assert accepted > 150 and refused > 0


def test_double_split_blocks_are_exact_or_refused():
    """CS_SPLIT2: a block over which the sum crosses TWO binade boundaries (it more than doubles: the ranks are still
    concentrated around the seed) -- located with approximate sums, applied to the exact one, every part checked."""
    rng = random.Random(77)
    accepted = refused = 0
    for trial in range(300):
        front = [rng.random() * 1e-3 for _ in range(rng.randrange(20, 200))]
        s = seq_sum(front)
        eb = (m.bits(s) >> 52) & 0x7FF
        top2 = m.from_bits((eb + 2) << 52)
        rows = rng.choice([64, 256, 1024])
        need = (top2 - s) * (1.05 + 0.6 * rng.random())          # ends inside binade e + 2
        block = [need / rows * (0.5 + rng.random()) for _ in range(rows)]
        want = s
        for a in block:
            want = want + a
        if ((m.bits(want) >> 52) & 0x7FF) != eb + 2:
            continue
        err = rng.choice([0.0, 1e-13, -1e-13, 1e-9, -1e-9, 2e-3, -2e-3, 5e-2])
        pre = s * (1.0 + err)
        if ((m.bits(pre) >> 52) & 0x7FF) != eb:
            continue
        sp = m.split_block2(pre, block, run=4)
        got = m.apply_split2(s, sp)
        if got is None:
            refused += 1
        else:
            accepted += 1
            assert m.bits(got) == m.bits(want), (trial, err)
    assert accepted > 100 and refused > 0
