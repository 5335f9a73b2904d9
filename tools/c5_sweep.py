"""BASELINE configs[4]: C2 geometry, target translating at v = (-5, 0, 0) m/s, 64 pulses at
PRI = 1 ms, 2^20 paths per pulse, gen-3 receive in BF_MODE_RECEIVE_IQ with a 1024-bin fast-time ADC ->
cube [64, 1024, (I, Q, W)] -> slow-time FFT -> range-Doppler map.  Prints the timing and where the map peaks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from beifong_amd import capi, scenes, sweep

n_pulses, pri, v = 64, 1e-3, np.array([-5.0, 0.0, 0.0])
n_paths = int(os.environ.get("PATHS", 1 << 20))
sd, lp = scenes.bus_receive(n_tris=200_000, n_paths=n_paths, t_bins=1024, dr=0.03, seed=4)
lp.mode = capi.BF_MODE_RECEIVE_IQ
offsets = (np.arange(n_pulses)[:, None] * pri * v[None, :]).astype(np.float32)
sweep.render_pulse_sweep(sd, lp, offsets[:3], n_streams=3)          # warm-up (library, allocator)
for n_streams in (1, 3):
    t = time.time()
    cube = sweep.render_pulse_sweep(sd, lp, offsets, n_streams=n_streams)
    dt = time.time() - t
    print(f"streams={n_streams}: {n_pulses} pulses x {n_paths} paths in {dt * 1e3:.1f} ms  ({dt / n_pulses * 1e3:.2f} ms per pulse, "
          f"incl. {n_streams} scene builds)", flush=True)
rd = np.abs(sweep.range_doppler(cube))
k, r = np.unravel_index(np.argmax(rd[1:]), rd[1:].shape)
lam = 0.5 * (sd.physics.lambda_min_nm + sd.physics.lambda_max_nm) * 1e-9
print(f"cube {cube.shape}, W per cell sum {cube[:, :, 2].sum():.0f}; strongest moving line: Doppler bin {k + 1}, range bin {r} "
      f"({r * 0.03:.2f} m); expected Doppler for the band centre: {(2 * 5.0 * pri / lam * n_pulses) % n_pulses:.1f} bins")
