"""BASELINE configs[4]: C2 geometry, target translating at v = (-5, 0, 0) m/s, 64 pulses at
PRI = 1 ms, 2^20 paths per pulse, gen-3 receive in BF_MODE_RECEIVE_IQ with a 1024-bin fast-time ADC ->
cube [64, 1024, (I, Q, W)] -> slow-time FFT -> range-Doppler map.  Prints the timing and where the map peaks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import beifong_amd
beifong_amd.configure_runtime()
from beifong_amd import capi, scenes, sweep

speed = float(os.environ.get("SPEED", 5.0))      # m/s towards the radar; 5 m/s walks 10 range bins in the 64 pulses,
n_pulses, pri, v = 64, 1e-3, np.array([-speed, 0.0, 0.0])   # 0.5 m/s stays within one bin (a clean Doppler line)
n_paths = int(os.environ.get("PATHS", 1 << 20))
lam0 = 8.6e6                                     # nm; +-0.1 % band: a narrow-band radar, so the Doppler line stays one bin wide
sd, lp = scenes.bus_receive(n_tris=200_000, n_paths=n_paths, t_bins=1024, dr=0.03, seed=4,
                            lambda_band_nm=(lam0 * 0.999, lam0 * 1.001))
lp.mode = capi.BF_MODE_RECEIVE_IQ
offsets = (np.arange(n_pulses)[:, None] * pri * v[None, :]).astype(np.float32)
sweep.render_pulse_sweep(sd, lp, offsets[:3], n_streams=2)          # warm-up (library, allocator)
t = time.time()
h = capi.Scene(sd)
print(f"one scene build (BVH + upload): {(time.time() - t) * 1e3:.1f} ms", flush=True)
h.close()
for n_streams in [int(x) for x in os.environ.get("STREAMS", "1,3").split(",")]:
    t = time.time()
    sw = sweep.PulseSweeper(sd, lp, n_streams)
    t_setup = time.time() - t
    times = []
    for rep in range(3):                         # the first sweep allocates the path pools and learns the launch plans
        t = time.time()
        cube = sw.render(offsets)
        times.append(time.time() - t)
    sw.close()
    print(f"streams={n_streams}: setup {t_setup * 1e3:.0f} ms (one BVH build, {n_streams - 1} clones); {n_pulses} pulses x {n_paths} paths in "
          f"{n_streams} batches: first sweep {times[0] * 1e3:.1f} ms, then {min(times[1:]) * 1e3:.1f} ms = {min(times[1:]) / n_pulses * 1e3:.2f} ms per pulse", flush=True)
    if os.environ.get("PER_PULSE"):
        t = time.time()
        sw2 = sweep.PulseSweeper(sd, lp, n_streams)
        sw2.render(offsets, per_pulse=True)
        t = time.time()
        cube_pp = sw2.render(offsets, per_pulse=True)
        print(f"   per-pulse loop (round 1): {(time.time() - t) * 1e3:.1f} ms; max |cube - cube_pp| / max|cube| = "
              f"{np.abs(cube - cube_pp).max() / np.abs(cube).max():.2e}", flush=True)
        sw2.close()
rd = np.abs(sweep.range_doppler(cube))
lam = 0.5 * (sd.physics.lambda_min_nm + sd.physics.lambda_max_nm) * 1e-9
far = rd[:, 200:]                            # fast-time cells beyond 6 m: the bus, not the ground under the antenna
prof = far[2:-1].sum(1)                      # Doppler profile without the static clutter line (bins 63, 0, 1)
k = int(np.argmax(prof)) + 2
r = int(np.argmax(far[k])) + 200
print(f"cube {cube.shape}, samples {cube[:, :, 2].sum():.0f}; static clutter (Doppler 0, beyond 6 m) energy {far[0].sum():.3e}; moving target: "
      f"Doppler bin {k} (energy {prof.max():.3e}, median of the other bins {np.median(prof):.3e}), strongest range bin {r} "
      f"({r * 0.03:.2f} m); band-centre prediction {(2 * speed * pri / lam * n_pulses) % n_pulses:.1f} bins")
