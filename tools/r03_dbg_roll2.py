import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from beifong_amd import capi, scenes
from tests.test_gpu_rolling import _launch_like
sd, lp = scenes.bus_receive(n_tris=20000, n_paths=20000, t_bins=256, dr=0.1)
K, calls = int(os.environ.get("K", 4)), 3
rng = np.random.default_rng(3)
offsets = np.concatenate([np.zeros((1, 3)), rng.uniform(-0.02, 0.02, (5, 3)), rng.uniform(-2.5, 2.5, (6, 3))]).astype(np.float32)[:K * calls]
seeds = [int(x) for x in rng.integers(1, 1 << 40, K * calls)]
use_off = os.environ.get("OFF", "1") == "1"
g = capi.Scene(sd)
n = g.channels(lp)
hist = torch.zeros((K * calls, n), dtype=torch.float32, device="cuda")
lr = _launch_like(lp, lp.seed, flags=capi.BF_FLAG_ROLLING)
for c in range(calls):
    k0 = c * K
    g.render_batch_device(lr, K, hist[k0].data_ptr(), seeds=seeds[k0:k0 + K], offsets=offsets[k0:k0 + K] if use_off else None)
g.flush(); g.sync()
h = hist.cpu().numpy()
g2 = capi.Scene(sd)
hb, rb, _ = g2.render_batch(lp, K * calls, seeds=seeds, offsets=offsets if use_off else None, records=True)
for k in range(K * calls):
    print(k, "W total rolling", h[k][2::3].sum(), "batch", hb[k][2::3].sum(), "A", h[k][1::3].sum(), hb[k][1::3].sum(), "Y", h[k][0::3].sum(), hb[k][0::3].sum())
