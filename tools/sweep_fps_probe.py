"""Frames per second of a sweep whose radar turns every frame (the loop shape the reference ships:
python_scripts/animated_trans_rad.py:307-384 — 73 frames —, Receive.ipynb cell 30 — 256 angles), on ONE scene handle:
  stand-alone : bf_scene_update_endpoints + a plain render per frame (one tail per frame)
  flush/frame : rolling renders, BF_ROLL_JOIN=0 — round 3: the endpoint update flushes the sequence (still one tail per frame)
  one sequence: rolling renders, the updates JOIN the sequence (round 4): one tail per sweep
and the same sweep dealt to FOUR handles (one rolling sequence each, as beifong_amd.sweep.render_sweep does).
usage: python tools/sweep_fps_probe.py [frames] [log2 paths per frame ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

torch.cuda.init()
import beifong_amd

beifong_amd.configure_runtime()
from beifong_amd import capi, scenes

F = int(sys.argv[1]) if len(sys.argv) > 1 else 73
logs = [int(a) for a in sys.argv[2:]] or [16, 18, 20]
mesh = scenes.bus_mesh(200_000)
yaws = np.linspace(-30.0, 30.0, F)


def run(kind, frames, n):
    os.environ["BF_ROLL_JOIN"] = "0" if kind == "flush/frame" else "1"
    g = capi.Scene(frames[0][0])
    nch = g.channels(frames[0][1])
    hist = torch.zeros((F, nch), dtype=torch.float32, device="cuda")
    s = torch.cuda.Stream()
    flags = 0 if kind == "stand-alone" else capi.BF_FLAG_ROLLING
    best = 1e9
    for rep in range(3):
        hist.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k, (sd, lp) in enumerate(frames):
            if k:
                g.update_endpoints(sd, stream=s.cuda_stream)
            l = capi.make_launch(lp.mode, n, seed=100 + k, bins=lp.bins, bin_width=lp.bin_width, color_mode=lp.color_mode, flags=flags)
            g.render_device(l, hist[k].data_ptr(), stream=s.cuda_stream)
        g.flush(stream=s.cuda_stream)
        s.synchronize()
        best = min(best, time.perf_counter() - t0)
        g.update_endpoints(frames[0][0], stream=s.cuda_stream)
        s.synchronize()
    w = hist[:, 4].sum().item()
    g.close()
    return best, w


def run_handles(frames, n, n_handles=4):
    """the same sweep dealt round-robin to `n_handles` clones of the scene, one stream and ONE rolling sequence each (what
    beifong_amd.sweep.render_sweep does): the handles' small launches share the GPU"""
    os.environ["BF_ROLL_JOIN"] = "1"
    first = capi.Scene(frames[0][0])
    hs = [first] + [first.clone() for _ in range(n_handles - 1)]
    nch = first.channels(frames[0][1])
    hist = torch.zeros((F, nch), dtype=torch.float32, device="cuda")
    ss = [torch.cuda.Stream() for _ in hs]
    best = 1e9
    for rep in range(3):
        hist.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k, (sd, lp) in enumerate(frames):
            j = k % n_handles
            hs[j].update_endpoints(sd, stream=ss[j].cuda_stream)
            l = capi.make_launch(lp.mode, n, seed=100 + k, bins=lp.bins, bin_width=lp.bin_width, color_mode=lp.color_mode, flags=capi.BF_FLAG_ROLLING)
            hs[j].render_device(l, hist[k].data_ptr(), stream=ss[j].cuda_stream)
        for j, h in enumerate(hs):
            h.flush(stream=ss[j].cuda_stream)
        for s_ in ss:
            s_.synchronize()
        best = min(best, time.perf_counter() - t0)
    w = hist[:, 4].sum().item()
    for h in hs[1:] + hs[:1]:
        h.close()
    return best, w


for lg in logs:
    n = 1 << lg
    frames = [scenes.bus_radar(n_paths=n, bins=256, dr=0.1, radar_yaw_deg=float(y), mesh=mesh) for y in yaws]
    row = []
    for kind in ("stand-alone", "flush/frame", "one sequence"):
        dt, w = run(kind, frames, n)
        assert w == F * n, (kind, w, F * n)
        row.append("%s %7.1f frames/s (%6.2f ms per frame)" % (kind, F / dt, dt / F * 1e3))
    dt, w = run_handles(frames, n)
    assert w == F * n, ("four handles", w, F * n)
    row.append("%s %7.1f frames/s (%6.2f ms per frame)" % ("one sequence on each of 4 handles", F / dt, dt / F * 1e3))
    print("%d frames x 2^%d paths, 200 k-triangle bus, radar turning per frame:  %s" % (F, lg, "   ".join(row)), flush=True)
