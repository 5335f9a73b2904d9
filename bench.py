#!/usr/bin/env python3
"""bench.py — Mrays/s + HBM-roofline fraction of the radar path-tracing hot path.

Contract (see the task brief): `python bench.py --gpus N --steps K --warmup W`;
for N > 1 the driver launches one rank per GPU through torch.distributed.run
(RCCL).  A step is one pass of the hot path over one batch of synthetic input:
BASELINE.json configs[1] — the Bus.obj-class monostatic radar scene (200 k
triangle synthetic bus + ground, 256 range bins, 64 spp per pulse) — rendered
for a batch of pulses (2^24 paths per GPU per step) and followed, for N > 1, by
the RCCL all-reduce of the per-GPU range histograms.  Weak scaling: per-GPU
work is fixed; GPU g renders global path indices [g*P, (g+1)*P) via
bf_launch.path_offset, so the union is one sample set.

Consecutive steps are independent renders (successive coherent processing
intervals of a pulse sweep), so they are issued round-robin on `--streams`
HIP streams (default 3, one bf_scene handle each): the latency-bound deep-path
tail of one step overlaps the head of the next.  The timed region carries no
instrumentation; ray counts and per-kernel HIP-event durations come from an
instrumented serial pass over the SAME steps (same seeds => same rays), which
also yields `ms_per_step_serial`.

Rank 0 prints ONE JSON line with `roofline` (dominant kernel: wf_trace)
and `cpu_baseline` (the CPU oracle, a port of the reference's scalar path,
timed on this box's host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--paths", type=int, default=1 << 24, help="paths per GPU per step")
    ap.add_argument("--tris", type=int, default=200_000)
    ap.add_argument("--cpu-paths", type=int, default=1 << 26, help="bounded sample for the CPU baseline")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams (scene handles) the steps rotate over")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE {world}")

    import numpy as np
    import torch                      # first: its libamdhip64.so.7 is the one HIP runtime of the process
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

    sd, lp = scenes.bus_radar(n_tris=args.tris, n_paths=args.paths, bins=256, dr=0.1, seed=1)
    n_streams = max(1, args.streams)
    handles = [capi.Scene(sd, lib) for _ in range(n_streams)]      # one render in flight per handle
    scene = handles[0]
    info = scene.info()
    n_chan = scene.channels(lp)
    hists = [torch.zeros(n_chan, dtype=torch.float32, device=dev) for _ in range(n_streams)]
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]

    def step(i, want_stats=False, flags=0):
        j = i % n_streams
        l = capi.make_launch(lp.mode, args.paths, seed=lp.seed + 1000003 * i, path_offset=rank * args.paths,
                             bins=lp.bins, bin_width=lp.bin_width, color_mode=lp.color_mode, flags=flags)
        with torch.cuda.stream(streams[j]):
            hists[j].zero_()
            st = handles[j].render_device(l, hists[j].data_ptr(), stream=streams[j].cuda_stream, want_stats=want_stats)
            if world > 1:
                dist.all_reduce(hists[j])     # RCCL sum of the per-GPU range histograms over xGMI
        return st

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # algorithmic bytes per ray: V_n * S_n + V_t * 48 + 48 (S_n = 128-byte four-wide nodes), with
    # V_n / V_t counted by the instrumented kernels on exactly this workload.
    # wf_trace only sees the rays that enter the mesh BVH (wf_shade resolves the
    # others against the rectangles + the BVH root boxes), so its roofline uses
    # ITS rays and ITS nodes/triangles (the tail kernel's share is taken out
    # pro rata by rays).
    st = step(0, want_stats=True, flags=capi.BF_FLAG_STATS)
    rays0 = st.n_rays_closest + st.n_rays_shadow
    v_n = st.n_nodes_visited / rays0
    v_t = st.n_tris_tested / rays0
    b_ray = v_n * info.node_bytes + v_t * info.tri_bytes + 48.0
    bvh_rays0 = st.n_rays_traced + st.n_rays_tail            # rays that walked the BVH at all
    share = st.n_rays_traced / max(bvh_rays0, 1)
    v_n_tr = st.n_nodes_visited * share / max(st.n_rays_traced, 1)
    v_t_tr = st.n_tris_tested * share / max(st.n_rays_traced, 1)
    b_ray_tr = v_n_tr * info.node_bytes + v_t_tr * info.tri_bytes + 48.0
    # node visits wf_trace serves from its LDS copy of the tree's top levels (they are algorithmic bytes of the
    # traversal all the same; the figure without them is reported next to the headline one)
    v_n_lds = st.n_nodes_lds / max(st.n_rays_traced, 1)
    b_ray_tr_mem = (v_n_tr - v_n_lds) * info.node_bytes + v_t_tr * info.tri_bytes + 48.0
    # every scene handle allocates its wavefront pool and learns its launch plan on first use:
    # touch each once (untimed, before the W warm-up steps) so neither pass below pays for that
    for j in range(n_streams):
        step(j)
    sync()
    for i in range(args.warmup):
        step(i)

    # instrumented serial pass over the steps of the timed region: ray counts (exact — the
    # same seeds are rendered again below) and HIP-event durations of every kernel launch
    sync()
    rays = 0
    paths = 0
    kernel_ms = 0.0
    trace_ms = shade_ms = tail_ms = 0.0
    trace_launches = 0
    rays_trace = 0
    for i in range(args.steps):
        st = step(i, want_stats=True)
        rays += st.n_rays_closest + st.n_rays_shadow
        paths += st.n_paths
        kernel_ms += st.kernel_ms
        trace_ms += st.trace_ms
        shade_ms += st.shade_ms
        tail_ms += st.tail_ms
        trace_launches += st.n_launches_trace
        rays_trace += st.n_rays_traced

    # timed region: EXACTLY `steps` steps, no instrumentation, barrier + synchronize on both sides
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync()
    dt = time.perf_counter() - t0

    tot = torch.tensor([dt, float(rays), float(paths), kernel_ms], dtype=torch.float64, device=dev)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0])
        rays_all, paths_all = float(sm[1]), float(sm[2])
    else:
        rays_all, paths_all = float(rays), float(paths)

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the
    # committed summary of the separate rocprofv3 --pmc passes over this workload supplies it
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_final_pmc_traffic.json")) as f:
            pt = json.load(f)
        if pt.get("kernel") == "wf_trace" and args.paths == (1 << 24) and args.tris == 200_000:
            traffic = float(pt["traffic_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        traffic = None

    out = None
    if rank == 0:
        mrays = rays_all / dt / 1e6
        # dominant kernel = wf_trace (BVH traversal): HIP events bracket every one
        # of its launches on the launch stream; a step issues one launch per bounce
        avg_launch_s = trace_ms / max(trace_launches, 1) / 1e3
        rays_per_launch = rays_trace / max(trace_launches, 1)
        achieved = b_ray_tr * rays_per_launch / avg_launch_s / 1e9
        out = {
            "metric": "Mrays/s (closest + any-hit BVH queries), Bus.obj-class radar scene",
            "value": round(mrays, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "ms_per_step_serial": round(kernel_ms / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "C2 bus_radar: synthetic %d-triangle bus (Bus.obj stand-in) + 20x20 m ground, monostatic "
                            "20x50 mm TX aperture + perspective RX, gen-2 range(pathlength) integrator, 256 range "
                            "bins dr=0.1 m, 64 spp x %d pulses = %d paths per GPU per step"
                            % (info.n_triangles, args.paths // 64, args.paths),
                "paths_per_gpu_per_step": args.paths,
                "range_bins": 256,
                "triangles": int(info.n_triangles),
                "bvh_nodes": int(info.n_bvh_nodes),
                "parallelism": "sample-sharded x%d, RCCL all-reduce of the range histogram" % world,
                "streams": n_streams,
                "mpaths_per_s": round(paths_all / dt / 1e6, 2),
                "rays_per_path": round(rays_all / paths_all, 3),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "wf_trace",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic,
                "traffic_note": "HBM bytes per wf_trace launch from rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, "
                                "profiles/r01_final_pmc.txt; algorithmic bytes per launch = %.3e" % (b_ray_tr * rays_per_launch),
                "measured": "HIP events around every launch, instrumented serial pass over the timed region's steps",
                "bytes_per_ray": round(b_ray_tr, 1),
                "nodes_per_ray": round(v_n_tr, 2),
                "tris_per_ray": round(v_t_tr, 2),
                "nodes_per_ray_from_lds": round(v_n_lds, 2),
                "achieved_without_lds_served_nodes": round(achieved * b_ray_tr_mem / b_ray_tr, 2),
                "all_rays": {"bytes_per_ray": round(b_ray, 1), "nodes_per_ray": round(v_n, 2), "tris_per_ray": round(v_t, 2),
                             "resolved_in_wf_shade_frac": round(1.0 - bvh_rays0 / rays0, 4)},
                "rays_per_launch": int(rays_per_launch),
                "launches_per_step": round(trace_launches / args.steps, 2),
                "avg_launch_ms": round(avg_launch_s * 1e3, 4),
                "per_step_ms": {"wf_trace": round(trace_ms / args.steps, 3), "wf_shade": round(shade_ms / args.steps, 3),
                                "tail": round(tail_ms / args.steps, 3), "all_kernels": round(kernel_ms / args.steps, 3)},
                "whole_pipeline_frac": round(b_ray * rays / (kernel_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5),
            },
        }
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(sd, lp, args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(sd, lp, args):
    """The oracle (kind "port": a CPU restatement of the reference's scalar
    path; the reference itself cannot be built offline) on this box's host
    cores, on a bounded sample of the same workload."""
    from beifong_amd import capi
    from tests.oracle_lib import OracleScene
    # the GPU box's CPU share for one GPU is 16 cores (task brief); affinity may list the whole host
    cores = min(16, len(os.sched_getaffinity(0)))
    o = OracleScene(sd)
    l = capi.make_launch(lp.mode, args.cpu_paths, seed=lp.seed, bins=lp.bins, bin_width=lp.bin_width, color_mode=lp.color_mode)
    t0 = time.perf_counter()
    _, _, st = o.render(l, rng_mode=0, threads=cores)
    dt = time.perf_counter() - t0
    rays = st.n_rays_closest + st.n_rays_shadow
    # single core (BASELINE.md §3 asks for both): 1/16 of the sample, the reference's own 1x1-film renders are one thread
    n1 = max(1 << 16, args.cpu_paths // 16)
    l1 = capi.make_launch(lp.mode, n1, seed=lp.seed, bins=lp.bins, bin_width=lp.bin_width, color_mode=lp.color_mode)
    t1 = time.perf_counter()
    _, _, st1 = o.render(l1, rng_mode=0, threads=1)
    dt1 = time.perf_counter() - t1
    return {
        "value": round(rays / dt / 1e6, 3),
        "unit": "Mrays/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d paths of the same C2 scene (same seed, per-path PCG32 streams), oracle/bf_oracle.cpp with its "
                  "own median-split BVH, %d std::threads, %.1f s wall" % (args.cpu_paths, cores, dt),
        "value_1core": round((st1.n_rays_closest + st1.n_rays_shadow) / dt1 / 1e6, 3),
        "sample_1core": "%d paths, 1 thread, %.1f s wall" % (n1, dt1),
    }


if __name__ == "__main__":
    main()
