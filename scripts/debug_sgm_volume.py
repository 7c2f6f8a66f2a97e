"""Compare the GPU's aggregated SGM volume with the oracle's and report where they differ."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import rand_pair
from cameracalibration_amd import api
from oracle import orc

cases = [(24, 40, 0, 16, 8, 32), (37, 53, 0, 64, 8, 32), (16, 130, 0, 256, 8, 32), (40, 70, 3, 200, 5, 30)]
for (H, W, minD, D, P1, P2) in cases:
    L, R = rand_pair(H * 31 + W, H, W, min(D, W))
    S = api.sgbm_volume(L, R, api.sgbm_params(minD, D, P1, P2, 10, 1, 8, 1)).astype(np.int64)
    _, E = orc.sgm(L, R, minD, D, P1, P2, 10, 1, 8, 1, want_S=True)
    E = E.astype(np.int64)
    bad = S != E
    print(f"case {H}x{W} D={D}: mismatches {bad.sum()} of {bad.size}")
    if bad.any():
        ys, xs, ds = np.nonzero(bad)
        print("  rows", np.unique(ys)[:20], "cols", np.unique(xs)[:40], "d", np.unique(ds)[:20])
        for i in range(min(8, len(ys))):
            print("   ", ys[i], xs[i], ds[i], "gpu", S[ys[i], xs[i], ds[i]], "orc", E[ys[i], xs[i], ds[i]])
        print("  diff stats: max", np.abs(S - E).max(), "mean", (S - E)[bad].mean())
