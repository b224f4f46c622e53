"""Reads the per-wave cycle counts the stamped experiments build of the sweep kernel leaves (make exp CXXFLAGS+=-DRWR_SWEEP_STAMPS,
RWR_X_SWEEP_STAMPS=path): per wave [cycles inside block changes (barrier + LDS refill + barrier), cycles of the whole kernel,
word-rows of its stream, blocks]."""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1], dtype=np.int64)
a = a[a[:, 1] > 0]
ent, tot, T, B = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
print(f"waves {len(a)}, blocks {B[0]}, T mean {T.mean():.1f} max {T.max()}")
print(f"whole kernel per wave: mean {tot.mean():.0f} max {tot.max()} cycles")
print(f"inside block changes : mean {ent.mean():.0f} max {ent.max()} cycles  = {ent.mean() / B[0]:.0f} per block ({100 * ent.mean() / tot.mean():.0f} % of the wave's time)")
o = np.argsort(-T)[:5]
for w in o:
    print(f"  heavy wave {w}: T {T[w]} total {tot[w]} block changes {ent[w]} -> summing {tot[w] - ent[w]} = {(tot[w] - ent[w]) / max(T[w], 1):.0f} cycles per word-row")
o = np.argsort(T)[:3]
for w in o:
    print(f"  light wave {w}: T {T[w]} total {tot[w]} block changes {ent[w]}")
