"""debug: rolling sequence with a small LDS window (many bins) vs stand-alone renders, channel by channel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from beifong_amd import capi, scenes
from tests.test_gpu_rolling import _Sequence, _launch_like

for bins, npaths, iters in ((4096, 1 << 17, "1"), (4096, 1 << 17, ""), (256, 1 << 17, "1")):
    if iters:
        os.environ["BF_ROLL_ITERS"] = iters
    else:
        os.environ.pop("BF_ROLL_ITERS", None)
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=npaths, bins=bins, dr=25.6 / bins)
    g = capi.Scene(sd)
    seeds = list(range(100, 112))
    seq = _Sequence(g, lp, seeds)
    seq.issue()
    g.flush()
    h, recs = seq.results()
    for k, seed in enumerate(seeds):
        hs, rs, _ = g.render(_launch_like(lp, seed), records=True)
        same = all(np.array_equal(recs[k][f].view(np.uint32), rs[f].view(np.uint32)) for f in ("L", "aux"))
        d = h[k].astype(np.float64) - hs
        bad = np.nonzero(np.abs(d) > 2e-5 * np.abs(hs) + npaths * 2.0 ** -24 * 4)[0]
        print(bins, iters or "auto", k, "records", same, "W", h[k][4], hs[4], "base diff", d[:5], "n bad bins", len(bad), bad[:8], d[bad[:4]])
