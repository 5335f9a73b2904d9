"""world_size-2 gloo test of the sample-sharding + histogram all-reduce logic
(the N > 1 path of bench.py / beifong_amd.dist), with the CPU oracle standing in
for the per-rank renderer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from beifong_amd import capi, scenes
from beifong_amd.dist import render_cube_sharded, render_sharded, shard_range


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == n
            pos = 0
            for off, cnt in spans:
                assert off == pos
                pos += cnt


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_paths, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.oracle_lib import OracleScene
    sd, lp = scenes.trans_rad(spp=n_paths)
    o = OracleScene(sd)
    n_floats = o.lib.bfo_launch_channels(lp)

    def render(off, cnt, out):
        l = capi.make_launch(lp.mode, cnt, seed=lp.seed, path_offset=off, bins=lp.bins, bin_width=lp.bin_width,
                             time_c=lp.time_c, color_mode=lp.color_mode)
        h, _, _ = o.render(l)
        out += torch.from_numpy(h)

    hist, span = render_sharded(render, n_paths, n_floats)
    np.save(os.path.join(out_dir, f"hist_{rank}.npy"), hist.numpy())
    np.save(os.path.join(out_dir, f"span_{rank}.npy"), np.array(span))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_all_reduce_equals_single_run(tmp_path):
    n_paths, world = 6001, 2
    mp.spawn(_worker, args=(world, _free_port(), n_paths, str(tmp_path)), nprocs=world, join=True)
    from tests.oracle_lib import OracleScene
    sd, lp = scenes.trans_rad(spp=n_paths)
    ref, _, _ = OracleScene(sd).render(lp)
    h0 = np.load(tmp_path / "hist_0.npy")
    h1 = np.load(tmp_path / "hist_1.npy")
    assert np.array_equal(h0, h1)                       # all-reduce leaves every rank with the sum
    assert np.allclose(h0, ref, rtol=1e-6, atol=1e-6)   # union of shards == one sample set
    assert h0[4] == n_paths
    s0, s1 = np.load(tmp_path / "span_0.npy"), np.load(tmp_path / "span_1.npy")
    assert s0[0] == 0 and s0[0] + s0[1] == s1[0] and s1[0] + s1[1] == n_paths


def _pulse_scene(k):
    """Pulse k of a tiny coherent sweep: the plate 4 mm closer per pulse (scene rebuilt — the oracle has no refit)."""
    sd, lp = scenes.plate_doppler(wavelength_m=0.1, n_paths=3001, plate_x=5.0 - 0.004 * k)
    return sd, lp


def _cube_worker(rank, world, port, n_pulses, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.oracle_lib import OracleScene
    _, lp0 = _pulse_scene(0)

    def render(off, cnt, cube):
        from beifong_amd.sweep import _shard_launch
        for k in range(n_pulses):
            sd, lp = _pulse_scene(k)
            h, _, _ = OracleScene(sd).render(_shard_launch(lp, off, cnt))
            cube[k] += torch.from_numpy(h)

    cube, _ = render_cube_sharded(render, int(lp0.n_paths), (n_pulses, 3))
    np.save(os.path.join(out_dir, f"cube_{rank}.npy"), cube.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_pulse_cube_equals_single_run(tmp_path):
    """C5 sharding: paths of every pulse split over the ranks, ONE all-reduce of the (I, Q, W) cube."""
    n_pulses, world = 4, 2
    mp.spawn(_cube_worker, args=(world, _free_port(), n_pulses, str(tmp_path)), nprocs=world, join=True)
    from tests.oracle_lib import OracleScene
    ref = np.stack([OracleScene(sd).render(lp)[0] for sd, lp in map(_pulse_scene, range(n_pulses))])
    c0, c1 = np.load(tmp_path / "cube_0.npy"), np.load(tmp_path / "cube_1.npy")
    assert np.array_equal(c0, c1)
    assert np.allclose(c0, ref, rtol=1e-5, atol=1e-6 * np.abs(ref).max())
    assert np.array_equal(c0[:, 2], ref[:, 2])             # W: the count of samples put, exact
    assert np.abs(ref[:, :2]).max() > 0 and not np.allclose(ref[0, :2], ref[1, :2])   # the phase moves with the plate


def test_shard_range_of_the_c_abi_is_the_partition_dist_uses():
    """bf_shard_range (what bf_render_sharded_device cuts a render with) == dist.shard_range (what bench.py --scaling
    strong and the torch path use): contiguous, exhaustive, non-overlapping, for sizes up to 2^62."""
    import random
    from beifong_amd import capi, dist
    lib = capi.load_library()
    rnd = random.Random(7)
    for _ in range(500):
        n = rnd.choice([0, 1, 7, 2 ** 19, 2 ** 22, 4096 << 10, rnd.randrange(1, 2 ** 40), 2 ** 62 + 12345])
        w = rnd.choice([1, 2, 3, 4, 7, 8, 64, 1000])
        tot = 0
        for r in range(w):
            a = capi.shard_range(n, r, w, lib)
            assert a == dist.shard_range(n, r, w) and a[0] == tot
            tot += a[1]
        assert tot == n
