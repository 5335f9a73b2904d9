"""Parity soak: full-size renders of every configuration family against the oracle, ALL per-path records compared
bit for bit (radiance, path length / time, validity, ray count) plus the counters.  A few minutes on the GPU box
(16 oracle threads); prints one line per case and exits non-zero on the first mismatch.
    python tools/soak_parity.py [n_seeds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from beifong_amd import capi, scenes
from tests.oracle_lib import OracleScene
from tests import test_gpu_parity as T

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
fails = 0


def check(name, sd, lp, flags=(0,)):
    global fails
    t = time.time()
    ho, ro, so = OracleScene(sd).render(lp, records=True, threads=16)
    t_o = time.time() - t
    g = capi.Scene(sd)
    for fl in flags:
        lp.flags = fl
        hg, rg, sg = g.render(lp, records=True)
        bad = {k: int((rg[k] != ro[k]).sum()) for k in ("n_rays", "valid")}
        bad.update({k: int((rg[k].view(np.uint32) != ro[k].view(np.uint32)).sum()) for k in ("aux", "L")})
        ctr = (sg.n_rays_closest, sg.n_rays_shadow, sg.n_bounces, sg.n_invalid) == (so.n_rays_closest, so.n_rays_shadow, so.n_bounces, so.n_invalid)
        ok = not any(bad.values()) and ctr
        fails += 0 if ok else 1
        print(f"{'ok  ' if ok else 'FAIL'} {name:44s} flags {fl} paths {lp.n_paths:9d} rays {sg.n_rays_closest + sg.n_rays_shadow:10d} "
              f"invalid {sg.n_invalid:3d} nan {int(np.isnan(ro['L']).sum()):3d} mismatches {bad} counters {ctr}  oracle {t_o:5.1f} s", flush=True)
        if not ok:
            i = np.flatnonzero(rg["L"].view(np.uint32) != ro["L"].view(np.uint32))[:3]
            print("     first:", i, rg[i], ro[i], flush=True)
    lp.flags = 0
    g.close()


if len(sys.argv) > 2 and sys.argv[2] == "fuzz":
    # the random scenes of tests/test_gpu_parity.py::_fuzz_scene at 2^21 paths each
    n = 1 << 21
    for seed in range(n_seeds):
        for receive in (False, True):
            sd, lp = T._fuzz_scene(seed, receive=receive)
            if lp.spp:
                lp.spp = n // (lp.film_width * lp.film_height)
                lp.n_paths = lp.spp * lp.film_width * lp.film_height
            else:
                lp.n_paths = n
            check(f"fuzz {'receive' if receive else 'render'} scene {seed} mode {lp.mode} depth {lp.max_depth} rr {lp.rr_depth}", sd, lp,
                  flags=(0, capi.BF_FLAG_MEGAKERNEL))
    print("FAILED" if fails else "all cases bit-exact")
    sys.exit(1 if fails else 0)

if len(sys.argv) > 2 and sys.argv[2] == "misc":
    # corners of the pipeline at scale: multi-pixel films, a pool much smaller than the launch (run with BF_WF_POOL=262144
    # to exercise regeneration into freed slots), depth / roulette limits, sharded launches, moved geometry
    from beifong_amd.scenedesc import Transform4f as TF
    for seed in range(n_seeds):
        sd, _ = T._zoo_scene(two_emitters=True, uv=True)
        film = (96, 64)
        sd.set_perspective(TF.translate([0, 0, 0.3]) * TF.rotate([1, 0, 0], 90) * TF.rotate([0, 1, 0], 90), fov=60.0, near_clip=0.1,
                           far_clip=100.0, film=film)
        sd.finalize()
        for mode, bins, bw in ((capi.BF_MODE_RANGE, 128, 0.1), (capi.BF_MODE_TIME, 40, 1e-9), (capi.BF_MODE_PATH, 0, 0.0)):
            lp = capi.make_launch(mode, film[0] * film[1] * 512, seed=1000 + seed, bins=bins, bin_width=bw, film=film, spp=512)
            check(f"film 96x64 mode {mode} seed {1000 + seed}", sd, lp, flags=(0, capi.BF_FLAG_MEGAKERNEL))
        for md, rr in ((1, 5), (2, 1), (3, 50), (12, 3), (-1, 1), (-1, 200)):
            sd, lp = scenes.bus_radar(n_tris=200_000, n_paths=1 << 22, seed=1100 + seed)
            lp.max_depth, lp.rr_depth = md, rr
            check(f"C2 max_depth {md} rr_depth {rr} seed {1100 + seed}", sd, lp)
        sd, lp = scenes.bus_radar(n_tris=200_000, n_paths=(1 << 22) + 12345, seed=1200 + seed)
        lp.path_offset = (1 << 33) + 7                       # a shard far into a > 2^32-path job
        check(f"C2 shard at path 2^33 seed {1200 + seed}", sd, lp)
        sd, lp = scenes.trans_rad(spp=1 << 22)
        lp.seed = 1300 + seed
        check(f"C1 trans_rad fluxmeter + spot seed {1300 + seed}", sd, lp, flags=(0, capi.BF_FLAG_MEGAKERNEL))
        # round 4: FMCW — a resample_freq chirp transmitter against the Wigner receiver's own local oscillator ("mix_resample"),
        # 200 k-triangle bus, 2^22 paths, a 256 x 64 time / beat-frequency ADC
        lam = 8.6e6
        sd, lp = scenes.bus_receive(n_tris=200_000, n_paths=1 << 22, t_bins=256, dr=0.1, seed=1400 + seed, receiver="wigner",
                                    lambda_band_nm=(lam * 0.999, lam * 1.001))
        Tw, f_c = sd.sensor.t_bandwidth, sd.emitters[0].freq_centre
        e, sn = sd.emitters[0], sd.sensor
        e.signal_type, e.freq_ext, e.pulse_len, e.prf, e.resample_freq = capi.BF_SIGNAL_LINFMCW, 0.002 * f_c, Tw, 1.0 / Tw, 1
        sn.freq_centre, sn.freq_ext, sn.rx_sig_is_delta = f_c, 0.002 * f_c, 1
        sn.rx_signal_type, sn.rx_pulse_len, sn.rx_prf = capi.BF_SIGNAL_LINFMCW, Tw, 1.0 / Tw
        sn.f_bins, sn.f_bandwidth = 64, 0.002 * f_c
        sd.finalize()
        lp.bins_y = 64
        lp.flags = capi.BF_FLAG_MIX_RESAMPLE                 # (the oracle's render takes the launch as it stands)
        check(f"C2-recv FMCW de-chirp (resample_freq + mix_resample, Wigner receiver) seed {1400 + seed}", sd, lp,
              flags=(capi.BF_FLAG_MIX_RESAMPLE, capi.BF_FLAG_MIX_RESAMPLE | capi.BF_FLAG_MEGAKERNEL))
    print("FAILED" if fails else "all cases bit-exact")
    sys.exit(1 if fails else 0)

if len(sys.argv) > 2 and sys.argv[2] == "batch":
    # round 2: batched launches (per-render seeds, on-the-fly mesh offsets), the Doppler hook, clones — every path of
    # every render of the batch against the oracle on a scene BUILT from the shifted vertices
    from beifong_amd import meshgen
    from tests.test_gpu_batch import _bus_receive_with_mesh

    def check_batch(name, g, lp, seeds, offsets, oracle_scene_of):
        global fails
        hb, rb, sb = g.render_batch(lp, len(seeds), seeds=seeds, offsets=offsets, records=True)
        bad_total = 0
        for k, sd_k in enumerate(oracle_scene_of):
            l1 = capi.bf_launch()
            import ctypes as C
            C.memmove(C.byref(l1), C.byref(lp), C.sizeof(capi.bf_launch))
            l1.seed = int(seeds[k])
            _, ro, _ = OracleScene(sd_k).render(l1, records=True, threads=16)
            bad = sum(int((rb[k][key].view(np.uint32) != ro[key].view(np.uint32)).sum()) for key in ("L", "aux"))
            bad += int((rb[k]["n_rays"] != ro["n_rays"]).sum()) + int((rb[k]["valid"] != ro["valid"]).sum())
            bad_total += bad
        ok = bad_total == 0
        fails += 0 if ok else 1
        print(f"{'ok  ' if ok else 'FAIL'} {name:52s} renders {len(seeds)} x {lp.n_paths} paths, rays {sb.n_rays_closest + sb.n_rays_shadow}, "
              f"tail rays {sb.n_rays_tail}, mismatching records {bad_total}", flush=True)

    v0, f0 = meshgen.bus(200_000, seed=1)
    v0 = meshgen.place(v0, yaw_deg=-20.0, translate=(10.0, 3.0, 1.7)).astype(np.float32)
    lam = (8.6e6 * 0.999, 8.6e6 * 1.001)
    for seed in range(n_seeds):
        rng = np.random.default_rng(seed)
        # C5-like: I/Q receive, 6 pulses with their own offsets (mm to metres) and seeds
        sd, lp = scenes.bus_receive(n_tris=200_000, n_paths=1 << 20, t_bins=1024, dr=0.03, seed=4, lambda_band_nm=lam)
        lp.mode = capi.BF_MODE_RECEIVE_IQ
        offs = np.concatenate([np.zeros((1, 3)), rng.uniform(-0.01, 0.01, (2, 3)), rng.uniform(-2.0, 2.0, (3, 3))]).astype(np.float32)
        seeds = [int(x) for x in rng.integers(1, 1 << 40, len(offs))]
        scenes_k = [_bus_receive_with_mesh(np.ascontiguousarray((v0 + o[None, :]).astype(np.float32)), f0, t_bins=1024, dr=0.03, lambda_band_nm=lam)
                    for o in offs]
        g = capi.Scene(sd)
        check_batch(f"C5 I/Q batch with offsets, soak seed {seed}", g, lp, seeds, offs, scenes_k)
        # the same through a clone, with the Doppler hook on
        c = g.clone()
        lp.mode = capi.BF_MODE_RECEIVE_RAW
        lp.flags = capi.BF_FLAG_DOPPLER
        check_batch(f"C2-recv + Doppler hook, clone, soak seed {seed}", c, lp, seeds[:3], offs[:3], scenes_k[:3])
        lp.flags = capi.BF_FLAG_DOPPLER | capi.BF_FLAG_MIX_RESAMPLE       # receive_type "mix_resample": beat-frequency rows
        check_batch(f"C2-recv mix_resample + Doppler, clone, soak seed {seed}", c, lp, seeds[3:5], offs[3:5], scenes_k[3:5])
        c.close()
        g.close()
        # range mode, no offsets: 4 renders x 2^22 paths with their own seeds (LDS-privatised batch histogram)
        sd, lp = scenes.bus_radar(n_tris=200_000, n_paths=1 << 22, seed=1)
        g = capi.Scene(sd)
        check_batch(f"C2 range batch of seeds, soak seed {seed}", g, lp, seeds[:4], None, [sd] * 4)
        g.close()
    print("FAILED" if fails else "all cases bit-exact")
    sys.exit(1 if fails else 0)

if len(sys.argv) > 2 and sys.argv[2] == "rolling":
    # round 3: rolling sequences at full size — the bench step (C2, 2^24 paths), C3, a C4 shard and C2-recv I/Q, several
    # renders per sequence with their own seeds, EVERY per-path record of every render against the oracle
    import torch
    from tests.test_gpu_rolling import _Sequence, _launch_like

    def check_rolling(name, sd, lp, seeds, env=None):
        global fails
        for k, v in (env or {}).items():
            os.environ[k] = v
        g = capi.Scene(sd)
        for k in (env or {}):
            os.environ.pop(k)
        seq = _Sequence(g, lp, seeds, extra_flags=capi.BF_FLAG_STATS)
        seq.issue()
        st = g.flush(want_stats=True)
        h, recs = seq.results()
        o = OracleScene(sd)
        bad_total, rays = 0, 0
        for k, seed in enumerate(seeds):
            _, ro, so = o.render(_launch_like(lp, seed), records=True, threads=16)
            rays += so.n_rays_closest + so.n_rays_shadow
            bad_total += sum(int((recs[k][key].view(np.uint32) != ro[key].view(np.uint32)).sum()) for key in ("L", "aux"))
            bad_total += int((recs[k]["n_rays"] != ro["n_rays"]).sum()) + int((recs[k]["valid"] != ro["valid"]).sum())
        ok = bad_total == 0 and rays == st.n_rays_closest + st.n_rays_shadow and st.n_guard == 0
        fails += 0 if ok else 1
        print(f"{'ok  ' if ok else 'FAIL'} {name:46s} {len(seeds)} renders x {lp.n_paths} paths, rays {st.n_rays_closest + st.n_rays_shadow} "
              f"(oracle {rays}), tail rays {st.n_rays_tail}, iterations {st.n_bounce_iters}, mismatching records {bad_total}", flush=True)
        g.close()
        del seq
        torch.cuda.empty_cache()

    for seed in range(n_seeds):
        rng = np.random.default_rng(1000 + seed)
        seeds = [int(x) for x in rng.integers(1, 1 << 40, 5)]
        sd, lp = scenes.bus_radar(n_tris=200_000, n_paths=1 << 24, seed=1)
        check_rolling(f"C2 bench step, soak seed {seed}", sd, lp, seeds[:3])
        sd, lp = scenes.car_radar(n_tris=1_000_000, n_paths=1 << 20, bins=1024, dr=0.03, seed=2)
        check_rolling(f"C3 car, soak seed {seed}", sd, lp, seeds)
        check_rolling(f"C3 car, one iteration per call, soak seed {seed}", sd, lp, seeds, env={"BF_ROLL_ITERS": "1"})
        sd, lp = scenes.multi_mesh_radar(n_paths=1 << 19)
        check_rolling(f"C4 shard, soak seed {seed}", sd, lp, seeds)
        sd, lp = scenes.bus_receive(n_tris=200_000, n_paths=1 << 20, t_bins=1024, dr=0.03, seed=4, lambda_band_nm=(8.6e6 * 0.999, 8.6e6 * 1.001))
        lp.mode = capi.BF_MODE_RECEIVE_IQ
        check_rolling(f"C2-recv I/Q, soak seed {seed}", sd, lp, seeds[:4])
    # round 4: a sweep whose RADAR TURNS every frame as ONE rolling sequence (bf_scene_update_endpoints joins it): every per-path
    # record of every frame against the oracle on the scene rebuilt for that frame
    def check_turning(name, builder, n, yaws, mode=None, env=None):
        global fails
        frames = [builder(y) for y in yaws]
        if mode is not None:
            for _, l in frames:
                l.mode = mode
        for k, v in (env or {}).items():
            os.environ[k] = v
        g = capi.Scene(frames[0][0])
        for k in (env or {}):
            os.environ.pop(k)
        K = len(frames)
        hist = torch.zeros((K, g.channels(frames[0][1])), dtype=torch.float32, device="cuda")
        rec = torch.zeros((K, n, 4), dtype=torch.int32, device="cuda")
        for k, (sd, lp) in enumerate(frames):
            if k:
                g.update_endpoints(sd)
            g.render_device(_launch_like(lp, 4000 + k, flags=capi.BF_FLAG_ROLLING | capi.BF_FLAG_COUNT), hist[k].data_ptr(), records_ptr=rec[k].data_ptr())
        st = g.flush(want_stats=True)
        torch.cuda.synchronize()
        r = rec.cpu().numpy().view(np.uint32).reshape(K, -1, 4)
        bad_total, rays = 0, 0
        for k, (sd, lp) in enumerate(frames):
            rk = np.ascontiguousarray(r[k]).view(capi.PATH_RECORD_DTYPE).reshape(-1)
            _, ro, so = OracleScene(sd).render(_launch_like(lp, 4000 + k), records=True, threads=16)
            rays += so.n_rays_closest + so.n_rays_shadow
            bad_total += sum(int((rk[key].view(np.uint32) != ro[key].view(np.uint32)).sum()) for key in ("L", "aux"))
            bad_total += int((rk["n_rays"] != ro["n_rays"]).sum()) + int((rk["valid"] != ro["valid"]).sum())
        ok = bad_total == 0 and rays == st.n_rays_closest + st.n_rays_shadow and st.n_guard == 0 and st.n_launches_tail <= 1
        fails += 0 if ok else 1
        print(f"{'ok  ' if ok else 'FAIL'} {name:46s} {K} frames x {n} paths (radar turned per frame, ONE sequence: {st.n_launches_tail} tail), rays "
              f"{st.n_rays_closest + st.n_rays_shadow} (oracle {rays}), mismatching records {bad_total}", flush=True)
        g.close()
        torch.cuda.empty_cache()

    from beifong_amd import meshgen
    bus = scenes.bus_mesh(200_000)
    bus_rx = meshgen.bus(200_000, seed=1)
    for seed in range(n_seeds):
        yaws = [float(y) for y in np.linspace(-20.0, 20.0, 6) + seed]
        n = 1 << 20
        check_turning(f"C2 turning radar, soak seed {seed}", lambda y: scenes.bus_radar(n_paths=n, bins=256, dr=0.1, radar_yaw_deg=y, mesh=bus), n, yaws)
        check_turning(f"C2 turning radar, one iteration per call, seed {seed}", lambda y: scenes.bus_radar(n_paths=n, bins=256, dr=0.1, radar_yaw_deg=y, mesh=bus),
                      n, yaws, env={"BF_ROLL_ITERS": "1"})
        check_turning(f"C2-recv I/Q turning radar, soak seed {seed}",
                      lambda y: scenes.bus_receive(n_paths=n, t_bins=1024, dr=0.03, radar_yaw_deg=y, mesh=bus_rx, lambda_band_nm=(8.6e6 * 0.999, 8.6e6 * 1.001)),
                      n, yaws, mode=capi.BF_MODE_RECEIVE_IQ)
    print("FAILED" if fails else "all cases bit-exact")
    sys.exit(1 if fails else 0)

for seed in range(n_seeds):
    sd, lp = scenes.bus_radar(n_tris=200_000, n_paths=1 << 24, seed=100 + seed)
    check(f"C2 range bus seed {100 + seed}", sd, lp)
    sd, lp = scenes.bus_radar(n_tris=200_000, n_paths=1 << 22, seed=200 + seed)
    lp.mode, lp.bin_width, lp.bins = capi.BF_MODE_TIME, 1e-9, 128
    check(f"C2 time bus seed {200 + seed}", sd, lp, flags=(0, capi.BF_FLAG_MEGAKERNEL))
    sd, lp = scenes.car_radar(n_tris=1_000_000, n_paths=1 << 22, seed=300 + seed)
    check(f"C3 car seed {300 + seed}", sd, lp)
    sd, lp = scenes.multi_mesh_radar(n_paths=1 << 22, seed=400 + seed)
    check(f"C4 multi-mesh seed {400 + seed}", sd, lp)
    for tx, rx, sig in (("wigner", "omnidirectional", "pulse"), ("wigner", "wigner", "linfmcw"), ("area", "omnidirectional", "pulse")):
        sd, lp = scenes.bus_receive(n_tris=200_000, n_paths=1 << 22, t_bins=1024, dr=0.03, seed=500 + seed, transmitter=tx, receiver=rx,
                                    signaltype=sig)
        check(f"C2-recv {tx}/{rx}/{sig} seed {500 + seed}", sd, lp)
    sd, lp = scenes.bus_receive(n_tris=200_000, n_paths=1 << 22, t_bins=256, dr=0.1, seed=600 + seed)
    lp.phase_bins = 16
    check(f"C2-recv phase AOVs seed {600 + seed}", sd, lp)
    sd, lp = scenes.bus_receive(n_tris=200_000, n_paths=1 << 22, t_bins=1024, dr=0.03, seed=700 + seed, lambda_band_nm=(8.6e6 * 0.999, 8.6e6 * 1.001))
    lp.mode = capi.BF_MODE_RECEIVE_IQ
    check(f"C5 pulse I/Q seed {700 + seed}", sd, lp, flags=(0, capi.BF_FLAG_MEGAKERNEL))
    sd, lp = scenes.phased_receive(n_tris=100_000, n_paths=1 << 21, steer_deg=(10.0, 0.0, 0.0))
    lp.seed = 800 + seed
    check(f"phased tx/rx seed {800 + seed}", sd, lp)
    sd, lp = T._zoo_scene(two_emitters=True, uv=True)
    lp.n_paths, lp.seed = 1 << 22, 900 + seed
    check(f"zoo (uv, two emitters) seed {900 + seed}", sd, lp)
print("FAILED" if fails else "all cases bit-exact")
sys.exit(1 if fails else 0)
