"""Executable model of the one-workgroup radix sort's place arithmetic (recommendersystems_amd/csrc/sort_small.h): every wave
owns one contiguous range of the array, a pass turns the (wave, digit) counts into output places digit-major / wave-minor,
and each wave then scatters its range 64 keys at a time, ranking equal digits inside the step and moving its own row of
places on.  The model follows the kernel phase by phase and must give numpy's stable sort -- the property every caller
(link sort, row orders, item order of the one-launch build; candidate ranking of the mid-size call) rests on.  The kernel
itself is checked on the GPU by the bitwise parity tests that go through those callers."""
import numpy as np
import pytest

NW, WAVE, RADIX = 16, 64, 256


def one_pass(keys, vals, shift):
    m = len(keys)
    per = ((m + NW - 1) // NW + WAVE - 1) // WAVE * WAVE
    digit = ((keys >> np.uint64(shift)) & np.uint64(RADIX - 1)).astype(np.int64)
    bounds = [(min(m, w * per), min(m, min(m, w * per) + per)) for w in range(NW)]
    wcnt = np.zeros((NW, RADIX), dtype=np.int64)
    for w, (lo, hi) in enumerate(bounds):                      # phase 1: each wave counts its own range
        np.add.at(wcnt[w], digit[lo:hi], 1)
    total = wcnt.sum(axis=0)                                   # phase 2: thread t owns digit t
    if (total == m).any():
        return keys.copy(), vals.copy(), True                  # constant digit: the pass is a copy
    at = np.cumsum(total) - total
    for q in range(NW):
        c = wcnt[q].copy()
        wcnt[q] = at
        at = at + c
    kout = np.empty_like(keys)
    vout = np.empty_like(vals)
    filled = np.zeros(m, dtype=bool)
    for w, (lo, hi) in enumerate(bounds):                      # phase 3: the wave walks its range in index order
        for base in range(lo, hi, WAVE):
            idx = np.arange(base, min(hi, base + WAVE))
            dg = digit[idx]
            place = wcnt[w][dg].copy()                         # every lane reads its place ...
            rank = np.array([int((dg[:l] == dg[l]).sum()) for l in range(len(idx))])
            for l in np.nonzero(rank == 0)[0]:                 # ... before a digit's first lane moves it on
                wcnt[w][dg[l]] = place[l] + int((dg == dg[l]).sum())
            pos = place + rank
            assert not filled[pos].any()
            filled[pos] = True
            kout[pos] = keys[idx]
            vout[pos] = vals[idx]
    assert filled.all()
    return kout, vout, False


def model_sort(keys, key_bits):
    vals = np.arange(len(keys), dtype=np.uint32)
    copies = 0
    for shift in range(0, key_bits, 8):
        keys, vals, copied = one_pass(keys, vals, shift)
        copies += copied
    return keys, vals, copies


@pytest.mark.parametrize("m", [0, 1, 63, 64, 65, 1023, 1024, 1025, 4097, 10000, 20480])
def test_wave_range_sort_is_numpy_stable_sort(m):
    rng = np.random.default_rng(m + 7)
    keys = rng.integers(0, 1 << 20, size=m, dtype=np.uint64)       # many equal keys: stability matters
    mask = np.uint64((1 << 24) - 1)
    got_k, got_v, _ = model_sort(keys.copy(), 24)
    want_v = np.argsort(keys & mask, kind="stable").astype(np.uint32)
    assert (got_v == want_v).all()
    assert (got_k == keys[want_v]).all()


def test_constant_digit_pass_is_a_copy_and_keeps_the_order():
    rng = np.random.default_rng(3)
    # score-shaped keys: the top byte (sign + high exponent bits) is the same for every key
    keys = (np.uint64(0x3F) << np.uint64(56)) | rng.integers(0, 1 << 40, size=5000, dtype=np.uint64)
    got_k, got_v, copies = model_sort(keys.copy(), 64)
    assert copies >= 2                                              # bits 40..55 are zero, bits 56..63 constant
    want_v = np.argsort(keys, kind="stable").astype(np.uint32)
    assert (got_v == want_v).all() and (got_k == keys[want_v]).all()


def test_skewed_digits_one_wave_holds_almost_everything():
    keys = np.zeros(3000, dtype=np.uint64)
    keys[::7] = 5
    keys[-1] = 200
    got_k, got_v, _ = model_sort(keys.copy(), 8)
    want_v = np.argsort(keys, kind="stable").astype(np.uint32)
    assert (got_v == want_v).all() and (got_k == keys[want_v]).all()
