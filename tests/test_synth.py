"""The synthetic generator is integer-only and must be identical everywhere."""
import numpy as np

from recommendersystems_amd import synth


def test_splitmix_and_mulhi_against_python_ints():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 2 ** 64, 2000, dtype=np.uint64)
    b = rng.integers(0, 2 ** 64, 2000, dtype=np.uint64)
    with np.errstate(over="ignore"):
        hi = synth.mulhi(a, b)
        sm = synth.splitmix64(a)
    for x, y, h, s in zip(a.tolist(), b.tolist(), hi.tolist(), sm.tolist()):
        assert h == (x * y) >> 64
        z = (x + 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
        assert s == z ^ (z >> 31)
    assert int(synth.splitmix64(np.array([0], dtype=np.uint64))[0]) == 0xE220A8397B1DCDAF   # published test vector


def test_bipartite_structure():
    g = synth.bipartite(0, 200, 700, 5000)
    U, I, E = 200, 700, g["likes"]
    n = U + I
    rp, dst = g["rowptr"], g["dst"]
    assert rp[0] == 0 and rp[-1] == 2 * E and len(dst) == 2 * E and E <= 5000
    fwd = set()
    for u in range(U):
        row = dst[rp[u]:rp[u + 1]]
        assert (np.diff(row) > 0).all() and (row >= U).all()          # ascending, de-duplicated, items only
        fwd |= {(u, int(t)) for t in row}
    back = set()
    for j in range(U, n):
        row = dst[rp[j]:rp[j + 1]]
        assert (np.diff(row) > 0).all() and (row < U).all()
        back |= {(int(s), j) for s in row}
    assert fwd == back and len(fwd) == E                               # every like is stored both ways
    # skew: low-index users/items are the active/popular ones
    deg_u = np.diff(rp[:U + 1])
    assert deg_u[:20].sum() > deg_u[-20:].sum()
    assert (synth.seeds_for(U, 8, 0, 8) == np.array([0, 25, 50, 75, 100, 125, 150, 175])).all()


def test_byte_formula():
    # SURVEY.md section 8d worked size: C2, K = 1, v = 8  ->  ~255 MB
    b = synth.algorithmic_bytes_per_step(600_000, 20_000_000, 1, 8)
    assert abs(b - 255e6) < 1e6
