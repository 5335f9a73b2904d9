#!/usr/bin/env python3
"""bench.py — Mrays/s + roofline fractions of the radar path-tracing hot path, per BASELINE.json config.

Contract (task brief): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per
GPU through torch.distributed.run (RCCL).  A step is one pass of the hot path over one batch of synthetic input:

  --config c2 (default, the config BASELINE.json's metric is quoted on): the Bus.obj-class monostatic radar scene
            (200 k-triangle synthetic bus + ground, gen-2 range(pathlength), 256 range bins, 64 spp per pulse) rendered
            for a batch of pulses: 2^24 paths per GPU per step;
  --config c3: Car-body.ply-class shell (1 M triangles, vertex normals) + ground, 1024 bins, 2^20 paths per step;
  --config c4shard: bus + car + motorbike (1.49 M triangles), 4096 bins, ONE of the 8 shards of C4 per GPU per step
            (2^19 paths); --config c4 --scaling strong: the configured 2^22 paths split over the ranks;
  --config c5: the 64-pulse coherent sweep (C2 geometry through gen-3 receive, I/Q ADC of 1024 fast-time bins, target
            moving 5 mm per pulse): one step = one sweep = 64 x 2^20 paths, rendered as batched launches
            (bf_render_batch_device).

followed, for N > 1, by the RCCL all-reduce of the per-GPU histograms.  Weak scaling: per-GPU work fixed, GPU g renders
global path indices [g*P, (g+1)*P) via bf_launch.path_offset (the union is one sample set); --scaling strong: the total
is fixed and split with beifong_amd.dist.shard_range.

Consecutive steps are independent renders (successive coherent processing intervals), issued round-robin on
`--streams` HIP streams — one handle per stream, all clones of ONE scene (bf_scene_clone: one BVH on the device) — so
the latency-bound tail of one step overlaps the heads of the next ones.  The HIP runtime maps a process's streams onto 4
hardware queues by default, which caps that overlap at three renders in flight (4 and 8 streams are even slower than 3:
two streams then share a queue with the default stream's work); the bench raises GPU_MAX_HW_QUEUES to 16 before HIP starts
and keeps 8 renders in flight (C5: 4 batches) — C2 8.9 -> 8.5 ms per step, C3 0.87 -> 0.73, C4 shard 1.16 -> 0.98
(profiles/r02_hw_queues_streams.txt).  The timed region carries no instrumentation; ray
counts, per-kernel algorithmic bytes (BF_FLAG_STATS counters) and per-kernel HIP-event durations come from two untimed
serial passes over the SAME steps (same seeds => same rays).

Rank 0 prints ONE JSON line.  `roofline`: headline = the whole path as SURVEY.md §8(d) defines it (traversal bytes of
ALL rays of a step / time of a step), `kernels` = one entry per kernel (wf_trace, wf_shade, tail) ranked by its share of
the GPU time, each with its own algorithmic bytes per launch / average launch duration, against the HBM peak and against
the Infinity-Cache gather rate (the scenes are cache-resident: DESIGN.md §3).  `cpu_baseline`: the CPU oracle (a port of
the reference's scalar path with its own SAH BVH; neither embree nor TBB exist offline), -O3 -march=native, one warm-up
+ best of 3, all host cores and one core, on a bounded sample of the same workload.
"""
import argparse
import os

# before anything loads the HIP runtime: hardware queues per process (default 4), see the note on streams above
# (= beifong_amd.configure_runtime(), spelled out here because nothing of the package is imported yet)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
CACHE_GATHER_GBS = 8600.0    # MI355X_MICROARCH.md "Indexed rows": 38 MB table gathered from the Infinity Cache, chip-wide
L2_GATHER_GBS = 17800.0      # same table: rows shared by every workgroup, served by the XCDs' L2s (16.8-18.8 TB/s)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4shard", "c4", "c5"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--paths", type=int, default=0, help="paths per GPU per step (per pulse for c5); 0 = the config's")
    ap.add_argument("--tris", type=int, default=0, help="c2 / c5: triangles of the synthetic bus; 0 = 200 000")
    ap.add_argument("--pulses", type=int, default=64, help="c5: pulses per sweep")
    ap.add_argument("--cpu-paths", type=int, default=0, help="bounded sample for the CPU baseline; 0 = the config's")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-iso", action="store_true", help="skip the stand-alone latency probe (profiling runs: only the sequence's launches)")
    ap.add_argument("--streams", type=int, default=0, help="HIP streams (scene handles) the steps rotate over; 0 = the config's")
    ap.add_argument("--rolling", type=int, default=1, help="1: the steps of a handle form a rolling sequence (BF_FLAG_ROLLING), flushed at the "
                    "end of the timed region; 0: every step is a stand-alone render with its own tail (round 2's scheme)")
    return ap.parse_args()


class Workload:
    """One BASELINE.json config as a scene, a launch and a way to issue step i."""

    def __init__(self, args, rank, world, capi, scenes):
        import numpy as np
        cfg = args.config
        self.cfg = cfg
        self.sweep = cfg == "c5"
        tris = args.tris or 200_000
        if cfg == "c2":
            paths = args.paths or (1 << 24)
            self.sd, self.lp = scenes.bus_radar(n_tris=tris, n_paths=paths, bins=256, dr=0.1, seed=1)
            self.label = ("C2 bus_radar: synthetic %d-triangle bus (Bus.obj stand-in) + 20x20 m ground, monostatic 20x50 mm TX "
                          "aperture + perspective RX, gen-2 range(pathlength) integrator, 256 range bins dr=0.1 m, 64 spp x %d "
                          "pulses = %d paths per GPU per step")
            self.streams = args.streams or (2 if args.rolling else 8)
            self.cpu_paths = args.cpu_paths or (1 << 24)
        elif cfg == "c3":
            paths = args.paths or (1 << 20)
            self.sd, self.lp = scenes.car_radar(n_tris=1_000_000, n_paths=paths, bins=1024, dr=0.03, seed=2)
            self.label = ("C3 car_radar: synthetic %d-triangle car-body shell with vertex normals (Car-body.ply stand-in) + "
                          "ground, gen-2 range(pathlength), 1024 range bins dr=0.03 m, 2^20 primary rays: %d x 64 = %d paths per GPU per step")
            self.streams = args.streams or (4 if args.rolling else 8)
            self.cpu_paths = args.cpu_paths or (1 << 22)
        elif cfg in ("c4shard", "c4"):
            total = 4096 << 10
            paths = args.paths or (total // 8 if cfg == "c4shard" else total)
            self.sd, self.lp = scenes.multi_mesh_radar(n_paths=paths, bins=4096, dr=0.01, seed=3)
            self.label = ("C4 multi_mesh_radar: bus + car + motorbike (%d triangles) on the ground, gen-2 range(pathlength), 4096 "
                          "range bins; %d x 64 = %d paths per step (the configured 4096 spp x 2^10 = 2^22 paths are 8 such shards)")
            self.streams = args.streams or (4 if args.rolling else 8)
            self.cpu_paths = args.cpu_paths or (1 << 22)
        else:
            paths = args.paths or (1 << 20)
            lam0 = 8.6e6        # nm; +-0.1 % band (tools/c5_sweep.py)
            self.sd, self.lp = scenes.bus_receive(n_tris=tris, n_paths=paths, t_bins=1024, dr=0.03, seed=4,
                                                  lambda_band_nm=(lam0 * 0.999, lam0 * 1.001))
            self.lp.mode = capi.BF_MODE_RECEIVE_IQ
            self.n_pulses = args.pulses
            v = np.array([-5.0, 0.0, 0.0])
            self.offsets = np.ascontiguousarray((np.arange(self.n_pulses)[:, None] * 1e-3 * v[None, :]).astype(np.float32))
            self.label = ("C5 pulse sweep: C2 geometry (%d triangles) through gen-3 receive (wigner transmitter, omnidirectional "
                          "receiver, BF_MODE_RECEIVE_IQ), 1024 fast-time bins, target at -5 m/s, PRI 1 ms; one step = one sweep of "
                          + str(self.n_pulses) + " pulses x %d x 64 = %d paths per pulse per GPU, batched launches")
            self.streams = args.streams or 4
            self.cpu_paths = args.cpu_paths or (1 << 22)
        self.total_paths = int(self.lp.n_paths)
        if args.scaling == "strong":
            from beifong_amd.dist import shard_range
            self.path_offset, self.paths = shard_range(self.total_paths, rank, world)
        else:
            self.path_offset, self.paths = rank * self.total_paths, self.total_paths
        self.receive = self.lp.mode in (capi.BF_MODE_RECEIVE_RAW, capi.BF_MODE_RECEIVE_IQ)
        self.capi = capi

    def launch(self, i, flags=0):
        c, lp = self.capi, self.lp
        if self.receive:
            l = c.make_launch(lp.mode, self.paths, seed=lp.seed + 1000003 * i, path_offset=self.path_offset, bins=lp.bins,
                              bins_y=lp.bins_y, flags=flags)
        else:
            l = c.make_launch(lp.mode, self.paths, seed=lp.seed + 1000003 * i, path_offset=self.path_offset, bins=lp.bins,
                              bin_width=lp.bin_width, color_mode=lp.color_mode, flags=flags)
        if os.environ.get("BENCH_MAX_DEPTH"):        # developer probe (what would a shorter tail buy); never set by the driver
            l.max_depth = int(os.environ["BENCH_MAX_DEPTH"])
        return l


STAT_KEYS = ("n_paths", "n_rays_closest", "n_rays_shadow", "n_nodes_visited", "n_tris_tested", "n_bounces", "n_rays_tail", "n_rays_traced",
             "n_nodes_lds", "n_nodes_tail", "n_wnodes_tail", "n_tris_tail", "n_bounces_tail", "n_shade_loads", "n_shade_stores",
             "n_shade_shadow", "n_shade_rays", "n_launches_trace", "n_bounce_iters", "kernel_ms", "trace_ms", "shade_ms", "tail_ms")


def add_stats(acc, st):
    for k in STAT_KEYS:
        acc[k] = acc.get(k, 0) + getattr(st, k)
    acc["n_tail_launches"] = acc.get("n_tail_launches", 0) + (st.n_launches_tail or (1 if st.tail_ms > 0 else 0))
    acc["n_shade_launches"] = acc.get("n_shade_launches", 0) + (st.n_launches_shade or st.n_bounce_iters)
    acc["kernel_variant"] = st.kernel_variant


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE {world}")

    import numpy as np
    import torch                      # first: its libamdhip64.so is the one HIP runtime of the process
    import torch.distributed as dist

    from beifong_amd import capi, scenes

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # BENCH_SHARE_DEVICE=1 / BENCH_DIST_BACKEND=gloo: rehearsal of the N > 1 code path on a one-GPU box (all ranks on
    # cuda:0, host-staged collectives); the driver's multi-GPU runs use neither
    dev_index = 0 if os.environ.get("BENCH_SHARE_DEVICE") else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    lib = capi.load_library()
    capi.check(lib, lib.bf_set_device(dev_index), "bf_set_device")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")       # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    w = Workload(args, rank, world, capi, scenes)
    n_streams = max(1, w.streams)
    rolling = bool(args.rolling)
    first = capi.Scene(w.sd, lib)                                  # ONE BVH build / upload ...
    handles = [first] + [first.clone() for _ in range(n_streams - 1)]   # ... shared by the stream handles
    info = first.info()
    n_chan = first.channels(w.lp)
    # one histogram PER STEP (a rolling render's histogram is complete only after the flush), one cube per sweep
    n_rows = max(args.steps, args.warmup, n_streams if not w.sweep else 1, 2)
    hists = torch.zeros((n_rows, n_chan * (w.n_pulses if w.sweep else 1)), dtype=torch.float32, device=dev)
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    roll_flag = capi.BF_FLAG_ROLLING if rolling else 0

    def step(i, acc=None, flags=0, only=None):
        """Issue step i (on handle `only`, else i mod n_streams).  Stand-alone renders with `acc` (a dict) are synchronous
        and add their bf_stats to it; rolling renders report through finish()."""
        want = acc is not None and not rolling
        l = w.launch(i, flags | roll_flag)
        if not w.sweep:
            j = i % n_streams if only is None else only
            with torch.cuda.stream(streams[j]):
                st = handles[j].render_device(l, hists[i].data_ptr(), stream=streams[j].cuda_stream, want_stats=want)
            if want:
                add_stats(acc, st)
            return
        # c5: one sweep = n_streams batches of pulses (one batched launch each: K pulses join the handle's rolling sequence
        # per call, or — stand-alone — form one launch sequence with its own tail), cube of step i in hists[i]
        want = acc is not None and not rolling
        cube = hists[i]
        # batches of at most 2^24 paths (a handle's pool), dealt round-robin to the handles
        kmax = max(1, (1 << 24) // w.paths)
        n_chunks = max(n_streams, (w.n_pulses + kmax - 1) // kmax)
        while w.n_pulses % n_chunks:          # equal batches: a handle's rolling sequence keeps one shape (K renders per call)
            n_chunks += 1
        bounds = [w.n_pulses * j // n_chunks for j in range(n_chunks + 1)]
        for c in range(n_chunks):
            k0, k1 = bounds[c], bounds[c + 1]
            if k1 == k0:
                continue
            j = c % n_streams if only is None else only
            with torch.cuda.stream(streams[j]):
                st = handles[j].render_batch_device(l, k1 - k0, cube.data_ptr() + 4 * n_chan * k0, offsets=w.offsets[k0:k1],
                                                    stream=streams[j].cuda_stream, want_stats=want)
            if want:
                add_stats(acc, st)
        if world > 1 and not rolling:
            cur = torch.cuda.current_stream(dev)
            for s in streams:
                cur.wait_stream(s)
            dist.all_reduce(cube)             # one all-reduce of the whole slow-time x fast-time cube

    def begin(n):
        """Zero the histograms of the next n steps (stream-ordered before every handle's work)."""
        hists[:n].zero_()
        for s in streams:
            s.wait_stream(torch.cuda.current_stream(dev))

    def finish(n, acc=None):
        """End of a region of n steps: flush every handle's rolling sequence (the ONE tail per handle), then, for N > 1, one
        RCCL all-reduce of all n per-step range histograms over xGMI (they are complete only now)."""
        for j in range(n_streams if rolling else 0):
            st = handles[j].flush(stream=streams[j].cuda_stream, want_stats=acc is not None)
            if st is not None and st.n_paths:
                add_stats(acc, st)
        if world > 1 and (rolling or not w.sweep):      # (stand-alone sweeps reduce their cube per step)
            cur = torch.cuda.current_stream(dev)
            for s in streams:
                cur.wait_stream(s)
            dist.all_reduce(hists[:n])

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # every scene handle allocates its path pool and learns its launch plan (and its flush plan) on first use: touch each
    # once (untimed, before the W warm-up steps) so neither pass below pays for that
    for rep in range(2 if rolling else 1):
        begin(n_streams)
        for j in range(n_streams if not w.sweep else 1):
            step(j)
        finish(n_streams)
        sync()
    begin(args.warmup)
    for i in range(args.warmup):
        step(i)
    finish(args.warmup)
    sync()

    # (A) counters of the timed region's steps: BF_FLAG_STATS counts node visits and triangle tests per kernel (the
    #     instrumented kernel variants; untimed).  (B) per-kernel HIP-event durations of the same steps, serial,
    #     product kernels.  Same seeds as the timed region below => same rays.  With rolling sequences both passes run the
    #     K steps as ONE sequence on ONE handle (every launch alone on the GPU) and read the sequence's totals at its flush.
    cnt, tim = {}, {}
    one = 0 if rolling else None
    per_step = w.sweep          # a sweep is dozens of renders: its sequence is read out (and flushed) sweep by sweep
    begin(args.steps)
    for i in range(args.steps):
        step(i, acc=cnt, flags=capi.BF_FLAG_STATS, only=one)
        if per_step:
            finish(args.steps, acc=cnt)
    finish(args.steps, acc=cnt)
    sync()
    begin(args.steps)
    for i in range(args.steps):
        step(i, acc=tim, flags=capi.BF_FLAG_TIMING if rolling else 0, only=one)
        if per_step:
            finish(args.steps, acc=tim)
    finish(args.steps, acc=tim)
    sync()

    # (C) a few stand-alone renders (their own tail each, nothing else on the GPU): the latency of ONE render, which is what
    #     an 8-GPU strong-scaled render of this config cannot go below
    iso = {}
    if not w.sweep and not args.no_iso:
        for i in range(min(args.steps, 5)):
            with torch.cuda.stream(streams[0]):
                add_stats(iso, handles[0].render_device(w.launch(i, 0), hists[i].data_ptr(), stream=streams[0].cuda_stream, want_stats=True))
        iso["n"] = min(args.steps, 5)
        # a stand-alone render closes handle 0's rolling sequence and resets what the handle had learnt about it (iterations per
        # call, flush plan): learn it again, untimed, as before the warm-up steps
        for rep in range(2 if rolling else 0):
            begin(n_streams)
            for j in range(n_streams):
                step(j)
            finish(n_streams)
            sync()

    # timed region: EXACTLY `steps` steps, no instrumentation, barrier + synchronize on both sides; every histogram of the
    # region is complete (flushed, and reduced for N > 1) when the clock stops
    sync()
    t0 = time.perf_counter()
    begin(args.steps)
    for i in range(args.steps):
        step(i)
    finish(args.steps)
    sync()
    dt = time.perf_counter() - t0

    # (D) the same steps as STAND-ALONE renders (every render its own tail; 8 handles in flight: round 2's scheme, what
    #     `--rolling 0` times): the figure a caller gets who cannot let consecutive renders share a sequence (the
    #     headline is the rolling sequence's; both belong on one line).  Untimed region of its own, after the timed one.
    standalone = None
    if rolling and not w.sweep and world == 1 and not args.no_iso:
        sa_n = 8
        sa_handles = handles + [first.clone() for _ in range(max(0, sa_n - n_streams))]
        sa_streams = streams + [torch.cuda.Stream(dev) for _ in range(max(0, sa_n - n_streams))]
        ks = min(args.steps, 16)

        def sa_pass(n):
            for i in range(n):
                j = i % sa_n
                with torch.cuda.stream(sa_streams[j]):
                    sa_handles[j].render_device(w.launch(i, 0), hists[i % n_rows].data_ptr(), stream=sa_streams[j].cuda_stream)
            for s_ in sa_streams:
                s_.synchronize()

        sa_pass(2 * sa_n)                      # pools, launch plans (the first render of a shape is synchronous)
        hists.zero_()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        sa_pass(ks)
        standalone = {"steps": ks, "handles": sa_n, "ms_per_step": (time.perf_counter() - t1) / ks * 1e3}
        for h_ in sa_handles[n_streams:]:
            h_.close()

    if "n_rays_closest" not in cnt:
        raise SystemExit("bench.py: no ray counters came back — a step of %d paths does not fit the handle's path pool, so its renders did "
                         "not join a rolling sequence (BF_WF_POOL?): run with --rolling 0" % w.paths)
    rays = cnt["n_rays_closest"] + cnt["n_rays_shadow"]
    paths = cnt["n_paths"]
    tot = torch.tensor([dt, float(rays), float(paths)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0])
        rays_all, paths_all = float(sm[1]), float(sm[2])
    else:
        rays_all, paths_all = float(rays), float(paths)

    if rank == 0:
        K = args.steps
        S_n, S_w, S_t, S_q = float(info.node_bytes), 512.0, float(info.tri_bytes), 48.0
        row = 100.0 + (16.0 if w.receive else 0.0) + (4.0 if w.sweep else 0.0)      # path-state row: ray0 ray1 sa sb sd [se] + hit + hit_prim [+ render]
        n4_tail, n16_tail = cnt["n_nodes_tail"], cnt["n_wnodes_tail"]
        n4_trace = cnt["n_nodes_visited"] - n4_tail - n16_tail
        t_trace = cnt["n_tris_tested"] - cnt["n_tris_tail"]
        # algorithmic bytes: traversal = V_n S_n + V_t S_t + S_q per ray with the node size of
        # the tree that was walked; shading = state rows read / written + triangle & normals of the shaded hit (96 B) +
        # the root node (128 B) each new ray is tested against in wf_shade
        # Node visits of wf_trace served from its LDS copy of the tree's top 85 nodes never reach the memory pipe: they are
        # NOT priced (pricing them at 128 B each made the headline 0.47 instead of 0.32); the figure with them
        # is printed next to it (frac_with_lds_served_nodes: the letter of SURVEY 8d, every visit x node size)
        n4_lds = cnt["n_nodes_lds"]
        b_trace_lds = n4_lds * S_n
        b_trace = (n4_trace - n4_lds) * S_n + t_trace * S_t + cnt["n_rays_traced"] * S_q
        b_tail = (n4_tail * S_n + n16_tail * S_w + cnt["n_tris_tail"] * S_t + cnt["n_rays_tail"] * S_q + cnt["n_bounces_tail"] * 96.0)
        b_shade = (cnt["n_shade_loads"] * row + cnt["n_shade_stores"] * row + cnt["n_shade_shadow"] * 36.0 +
                   (cnt["n_bounces"] - cnt["n_bounces_tail"]) * 96.0 + cnt["n_shade_rays"] * S_n)
        b_traversal_all = (n4_trace - n4_lds + n4_tail) * S_n + n16_tail * S_w + cnt["n_tris_tested"] * S_t + rays * S_q
        ms = {"wf_trace": tim["trace_ms"], "wf_shade": tim["shade_ms"], "tail": tim["tail_ms"]}
        launches = {"wf_trace": max(tim["n_launches_trace"], 1), "wf_shade": max(tim["n_shade_launches"], 1),
                    "tail": max(tim["n_tail_launches"], 1)}
        bytes_ = {"wf_trace": b_trace, "wf_shade": b_shade, "tail": b_tail}
        # the symbols rocprofv3 prints (template arguments: wf_trace<STATS, W, SHIFT, QUANT>, wf_shade<FIRST, W, RX> with
        # FIRST = 0 alive masks / 1 first bounce / 2 wake launch / 3 alive masks + eviction, bf_render_kernel<STATS, RESUME, SPILL, TW, VX>)
        # RX = mode class (0 render modes, 1 receive modes) | 8 for the lean build; the tail's fifth argument likewise (absent = 0)
        lean = tim.get("kernel_variant", 0) == capi.BF_VARIANT_LEAN
        names = {"wf_trace": "bfd::wf_trace<false, 5, %s, false>" % ("true" if w.sweep else "false"),
                 "wf_shade": "bfd::wf_shade<0|1|2|3, 3, %d>" % ((1 if w.receive else 0) | (8 if lean else 0)),
                 "tail": "bfd::bf_render_kernel<false, true, %s, 3%s> (tail)" % ("true" if info.bvh_stack_need > 32 else "false",
                                                                                   ", 8" if lean else "")}
        traffic = pmc_traffic(w, args, rolling, n_streams)
        total_ms = sum(ms.values()) or 1.0
        kernels = []
        for k in sorted(ms, key=lambda k: -ms[k]):
            per_launch = bytes_[k] / launches[k]
            avg_s = ms[k] / launches[k] / 1e3
            ach = per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
            tr = traffic.get(k) if traffic else None
            kernels.append({
                "kernel": names[k], "share_of_gpu_time": round(ms[k] / total_ms, 4), "ms_per_step": round(ms[k] / K, 4),
                "launches_per_step": round(launches[k] / K, 2), "avg_launch_ms": round(avg_s * 1e3, 4),
                "algorithmic_bytes_per_launch": round(per_launch), "achieved": round(ach, 2), "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "frac_of_cache_gather_ceiling": round(ach / CACHE_GATHER_GBS, 5),
                "traffic": tr, "traffic_over_algorithmic": (round(tr / per_launch, 3) if tr and per_launch else None),
                "frac_by_counter_traffic": (round(tr / avg_s / 1e9 / HBM_PEAK_GBS, 5) if tr and avg_s > 0 else None),
            })
            if k == "wf_trace":
                kernels[-1]["frac_with_lds_served_nodes"] = round((per_launch + b_trace_lds / launches[k]) / avg_s / 1e9 / HBM_PEAK_GBS, 5) if avg_s > 0 else 0.0
        step_s = dt / K
        whole = b_traversal_all / K / step_s / 1e9
        whole_serial = b_traversal_all / (tim["kernel_ms"] / 1e3) / 1e9
        whole_with_lds = (b_traversal_all + b_trace_lds) / K / step_s / 1e9
        traffic_step = (sum(kk["traffic"] * kk["launches_per_step"] for kk in kernels) if traffic and all(kk["traffic"] for kk in kernels) else None)
        mrays = rays_all / dt / 1e6
        out = {
            "metric": "Mrays/s (closest + any-hit BVH queries), %s" % {"c2": "Bus.obj-class radar scene", "c3": "Car-body.ply-class scene",
                       "c4shard": "bus+car+motorbike scene, one of 8 shards per GPU", "c4": "bus+car+motorbike scene", "c5": "64-pulse coherent sweep"}[w.cfg],
            "value": round(mrays, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 4),
            "ms_per_step_serial": round(tim["kernel_ms"] / K, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            # the dominant kernel and ITS fraction of the HBM roofline, beside the whole-path `roofline.frac` (both also inside `roofline`)
            "dominant_kernel": kernels[0]["kernel"],
            "dominant_kernel_frac": kernels[0]["frac"],
            "config": {
                "workload": w.label % (info.n_triangles, w.paths // 64, w.paths),
                "name": w.cfg,
                "paths_per_gpu_per_step": w.paths * (w.n_pulses if w.sweep else 1),
                "range_bins": int(w.lp.bins),
                "triangles": int(info.n_triangles),
                "bvh_nodes": int(info.n_bvh_nodes),
                "parallelism": "sample-sharded x%d (%s), RCCL all-reduce of the histogram" % (world, args.scaling),
                "streams": n_streams,
                "rolling": rolling,
                # which build of wf_shade / the tail ran (bf_stats.kernel_variant): "lean" = everything outside the radar scenes'
                # profile compiled out, chosen by the library per scene and launch; same per-path results
                "kernels": {0: "general", 1: "lean", 2: "wide-filter"}.get(tim.get("kernel_variant", 0), "general"),
                "mpaths_per_s": round(paths_all / dt / 1e6, 2),
                "rays_per_path": round(rays_all / paths_all, 3),
            },
            "roofline": {
                "bound": "hbm",
                "scope": "whole path: traversal bytes of ALL rays of a step (V_n S_n + V_t S_t + S_q, SURVEY 8d; node visits served from "
                         "wf_trace's LDS copy of the top of the tree are not priced) / ms_per_step",
                "kernel": kernels[0]["kernel"],                       # the dominant kernel (largest share of the GPU time) ...
                "kernel_frac": kernels[0]["frac"],                    # ... and ITS fraction of the HBM peak (kernels[0], repeated here)
                "kernel_share_of_gpu_time": kernels[0]["share_of_gpu_time"],
                "achieved": round(whole, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(whole / HBM_PEAK_GBS, 5),               # the WHOLE PATH (scope above), not one kernel
                "frac_serial": round(whole_serial / HBM_PEAK_GBS, 5),
                "frac_with_lds_served_nodes": round(whole_with_lds / HBM_PEAK_GBS, 5),
                "frac_by_counter_traffic": (round(traffic_step / step_s / 1e9 / HBM_PEAK_GBS, 5) if traffic_step else None),
                "traffic": (round(traffic_step) if traffic_step else None),
                "traffic_note": "L2 fabric-side bytes per step / per launch (Infinity-Cache hits INCLUDED: the counters sit between L2 and "
                                "the fabric, so this is L2-miss traffic, an upper bound of HBM traffic): rocprofv3 --pmc FETCH_SIZE (x the "
                                "correction factor measured for these kernels' access pattern, profiles/README.md) + WRITE_SIZE, separate "
                                "passes, profiles/r04_pmc_traffic.json (tools/profile_r04.sh), stamped with a hash of beifong_amd/csrc: null "
                                "when the kernels, the workload shape or the scheduling (rolling, streams) differ from the profiled run",
                "bytes_per_ray": round(b_traversal_all / rays, 1),
                "nodes_per_ray": round((n4_trace + n4_tail + n16_tail) / rays, 2),
                "tris_per_ray": round(cnt["n_tris_tested"] / rays, 2),
                "resolved_in_wf_shade_frac": round(1.0 - (cnt["n_rays_traced"] + cnt["n_rays_tail"]) / rays, 4),
                "ceiling_cache": {"note": "the scene (%.0f MB of nodes + triangles + normals) stays in the 256 MB Infinity Cache and partly in "
                                          "the 8 x 4 MB L2s, so the traversal's physical ceiling is the scattered-gather rate out of those caches "
                                          "(MI355X_MICROARCH.md, Indexed rows), not HBM" % (info.device_bytes / 1e6),
                                  "infinity_cache_gather_GBs": CACHE_GATHER_GBS, "l2_gather_GBs": L2_GATHER_GBS,
                                  "frac_of_infinity_cache_gather": round(whole / CACHE_GATHER_GBS, 5)},
                "measured": "HIP events around every launch (serial pass over the timed region's steps); counters from a BF_FLAG_STATS pass over the same steps",
                "kernels": kernels,
                "state_row_bytes": row,
                "with_shading_state": {"bytes_per_step": round((b_trace + b_tail + b_shade) / K),
                                       "achieved": round((b_trace + b_tail + b_shade) / K / step_s / 1e9, 2)},
            },
        }
        if world == 1 and w.cfg in ("c3", "c4shard", "c4", "c2"):
            # what an 8-GPU strong-scaled render of this config would see: per-GPU time of 1/8 of the paths cannot drop
            # below the tail
            if iso:
                out["config"]["isolated_step_ms"] = round(iso["kernel_ms"] / iso["n"], 3)
                out["config"]["isolated_tail_ms"] = round(iso["tail_ms"] / iso["n"], 3)
            out["config"]["tail_ms_per_step"] = round(tim["tail_ms"] / K, 3)
        if standalone:
            # stand-alone renders of the same steps (round 2's scheme): like-for-like with a caller's one-render-per-call loop
            sa_ms = standalone["ms_per_step"]
            out["value_standalone"] = round(rays / K / (sa_ms / 1e3) / 1e6, 2)
            out["standalone"] = {"ms_per_step": round(sa_ms, 4), "steps": standalone["steps"], "handles": standalone["handles"],
                                 "note": "the same steps as stand-alone renders (own tail each, %d handles in flight); `value` is the rolling "
                                         "sequence of %d steps with ONE tail per handle, flushed inside the timed region" % (standalone["handles"], K)}
        if not args.no_cpu and world == 1:          # (rank 0 at N = 1 only: the task's contract; an N > 1 run is not kept waiting for it)
            out["cpu_baseline"] = cpu_baseline(w, args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def csrc_hash():
    """sha256 over the kernel and host sources of libbeifong_hip.so (what the committed PMC summary was measured on)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "beifong_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_traffic(w, args, rolling, n_streams):
    """HBM bytes per launch of each kernel from the committed PMC summary of the same workload shape (PMC counters cannot
    be read from inside this process: separate rocprofv3 --pmc passes, tools/profile_r04.sh).  The summary carries a hash
    of beifong_amd/csrc: a kernel change makes it stale and the entry null, not silently wrong."""
    try:
        with open(os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")) as f:
            pt = json.load(f)
        e = pt.get(w.cfg)
        if pt.get("csrc_sha16") != csrc_hash() or not e or e.get("paths") != w.paths or args.tris not in (0, 200_000):
            return None
        if bool(e.get("rolling")) != bool(rolling):
            return None
        return {k: float(v) for k, v in e["traffic_bytes_per_launch"].items()}
    except (OSError, ValueError, KeyError):
        return None


def fast_oracle_path():
    """oracle/_fast/libbf_oracle_fast_<cpu>.so: the oracle built -O3 -march=native FOR THE CPU THIS RUNS ON (the file
    name carries a hash of the CPU's model and flags, so a library built on another box is never loaded)."""
    try:
        with open("/proc/cpuinfo") as f:
            txt = f.read()
        key = "".join(l for l in txt.splitlines() if l.startswith(("model name", "flags")))[:20000]
    except OSError:
        key = "unknown"
    tag = hashlib.sha256(key.encode()).hexdigest()[:10]
    out = os.path.join(ROOT, "oracle", "_fast", "libbf_oracle_fast_%s.so" % tag)
    newest = max(os.path.getmtime(os.path.join(ROOT, "oracle", "bf_oracle.cpp")), os.path.getmtime(os.path.join(ROOT, "include", "beifong_hip.h")))
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "fast", "FAST_OUT=" + out], check=True)
    return out


def cpu_baseline(w, args):
    """the oracle (kind "port": a CPU restatement of the reference's scalar path; the reference itself
    cannot be built offline and neither embree nor TBB exist here) built -O3 -march=native -ffp-contract=off, with its own
    binned-SAH BVH, on this box's host cores: one warm-up + best of 3, all cores and one core, bounded samples of the same
    scene and launch."""
    import ctypes as C
    from beifong_amd import capi
    from tests import oracle_lib
    lib = oracle_lib.load_from(fast_oracle_path())
    cores = min(16, len(os.sched_getaffinity(0)))       # the GPU box's CPU share for one GPU is 16 cores
    t0 = time.perf_counter()
    o = oracle_lib.OracleScene(w.sd, accel=2, lib=lib)
    t_build = time.perf_counter() - t0

    def run(n_paths, threads):
        l = w.launch(0)
        l.n_paths = n_paths
        l.path_offset = 0
        best, rays = None, 0
        for k in range(4):                                  # first = warm-up
            t = time.perf_counter()
            _, _, st = o.render(l, rng_mode=0, threads=threads)
            d = time.perf_counter() - t
            rays = st.n_rays_closest + st.n_rays_shadow
            if k and (best is None or d < best):
                best = d
        return rays, best

    n_all = w.cpu_paths
    n_one = max(1 << 14, n_all // 16)
    rays_all, t_all = run(n_all, cores)
    rays_one, t_one = run(n_one, 1)
    return {
        "value": round(rays_all / t_all / 1e6, 3),
        "unit": "Mrays/s",
        "cores": cores,
        "kind": "port",
        "what": "SAH BVH2 port of the reference's scalar path (oracle/bf_oracle.cpp, accel = binned SAH, 16 bins, leaves <= 4); no embree, no TBB "
                "(neither exists offline); reported baseline, not the optimisation target",
        "build": "g++ -O3 -march=native -std=c++17 -ffp-contract=off -fno-fast-math (oracle/Makefile: fast), built on this host",
        "protocol": "1 warm-up + best of 3, wall clock around bfo_render",
        "sample": "%d paths of the same scene and launch (seed of step 0, per-path PCG32 streams), %d std::threads, best %.2f s; BVH build %.2f s"
                  % (n_all, cores, t_all, t_build),
        "value_1core": round(rays_one / t_one / 1e6, 3),
        "sample_1core": "%d paths, 1 thread, best %.2f s" % (n_one, t_one),
    }


if __name__ == "__main__":
    main()
