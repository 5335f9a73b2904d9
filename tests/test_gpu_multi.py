"""Multi-GPU behind the C ABI (include/beifong_hip.h: bf_shard_range, bf_render_sharded_device, bf_render_sharded,
bf_allreduce_device; SURVEY 8b "device_mask", 8e).  A GPU box of this pool has ONE MI355X, so what runs here is the
one-device degenerate case of every entry point — same code path up to the size of the communicator — plus the
partition arithmetic; the multi-device legs are covered by construction (bf_shard_range == dist.shard_range, the 2-rank
gloo tests, test_render_sharding_by_path_offset)."""
import ctypes as C
import subprocess
import os

import numpy as np
import pytest

from beifong_amd import capi, scenes

pytestmark = pytest.mark.gpu


def _close_hist(hb, hs, n_paths, amax):
    atol = n_paths * 2.0 ** -24 * max(amax, 1.0) * 4
    assert np.allclose(hb, hs, rtol=2e-5, atol=atol), float(np.abs(hb - hs).max())


def test_sharded_render_on_one_device_is_the_plain_render(hiplib):
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=1 << 16, bins=256, dr=0.1)
    g = capi.Scene(sd)
    assert g.info().device == 0
    hs, rs, ss = g.render(lp, records=True)
    hm, sm = capi.render_sharded([g], lp)
    _close_hist(hm, hs, lp.n_paths, float(np.abs(rs["L"]).max()))
    assert hm[3] == hs[3] and hm[4] == hs[4] == lp.n_paths           # alpha and weight channels: integer sums, exact
    assert (sm.n_paths, sm.n_rays_closest, sm.n_rays_shadow, sm.n_bounces) == (ss.n_paths, ss.n_rays_closest, ss.n_rays_shadow, ss.n_bounces)


def test_sharded_device_entry_and_a_communicator_of_one(hiplib):
    """bf_render_sharded_device on one GPU + bf_allreduce_device over a one-GPU communicator: RCCL is found, initialised
    (ncclCommInitAll) and run; the sum over one rank is the identity."""
    import torch
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=1 << 15, bins=256, dr=0.1)
    g = capi.Scene(sd)
    hs, rs, _ = g.render(lp, records=True)
    d = torch.zeros(g.channels(lp), dtype=torch.float32, device="cuda")
    s = torch.cuda.Stream()
    capi.render_sharded_device([g], lp, [d.data_ptr()], streams=[s.cuda_stream])
    bufs = (C.c_void_p * 1)(C.c_void_p(d.data_ptr()))
    streams = (C.c_void_p * 1)(C.c_void_p(s.cuda_stream))
    devs = (C.c_int * 1)(0)
    capi.check(hiplib, hiplib.bf_allreduce_device(devs, 1, bufs, d.numel(), streams), "bf_allreduce_device")
    s.synchronize()
    _close_hist(d.cpu().numpy(), hs, lp.n_paths, float(np.abs(rs["L"]).max()))
    # the same device twice is a caller error, not a hang inside RCCL
    devs2 = (C.c_int * 2)(0, 0)
    bufs2 = (C.c_void_p * 2)(C.c_void_p(d.data_ptr()), C.c_void_p(d.data_ptr()))
    assert hiplib.bf_allreduce_device(devs2, 2, bufs2, d.numel(), None) == capi.BF_ERR_INVALID


def test_sharded_rolling_shards_need_a_flush_and_reduce_by_hand(hiplib):
    """BF_FLAG_ROLLING through the sharded entry: no collective is issued; flush, then bf_allreduce_device."""
    import torch
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=1 << 15, bins=256, dr=0.1)
    g = capi.Scene(sd)
    hs, rs, _ = g.render(lp, records=True)
    d = torch.zeros(g.channels(lp), dtype=torch.float32, device="cuda")
    lr = capi.make_launch(lp.mode, lp.n_paths, seed=lp.seed, bins=lp.bins, bin_width=lp.bin_width, color_mode=lp.color_mode,
                          flags=capi.BF_FLAG_ROLLING)
    capi.render_sharded_device([g], lr, [d.data_ptr()])
    g.flush()
    g.sync()
    _close_hist(d.cpu().numpy(), hs, lp.n_paths, float(np.abs(rs["L"]).max()))


def test_entry_points_run_on_the_handles_device(hiplib):
    """Every call switches to its handle's device and back (one host thread drives the handles of several GPUs)."""
    import torch
    sd, lp = scenes.bus_radar(n_tris=5000, n_paths=4096, bins=64, dr=0.4)
    g = capi.Scene(sd)
    before = torch.cuda.current_device()
    h0, r0, _ = g.render(lp, records=True)
    assert torch.cuda.current_device() == before
    if hiplib.bf_device_count() >= 2:
        # a handle that lives on ANOTHER GPU than the caller's current one: the host-buffer entries (bf_render, bf_trace_*:
        # staging buffers allocated before the kernels are launched) must allocate and run there too
        capi.check(hiplib, hiplib.bf_set_device(1), "bf_set_device")
        try:
            g1 = capi.Scene(sd)
        finally:
            capi.check(hiplib, hiplib.bf_set_device(before), "bf_set_device")
        assert g1.info().device == 1
        h1, r1, _ = g1.render(lp, records=True)
        assert torch.cuda.current_device() == before
        assert np.array_equal(r1["L"].view(np.uint32), r0["L"].view(np.uint32)) and np.array_equal(r1["n_rays"], r0["n_rays"])
        rays = np.zeros((64, 8), dtype=np.float32)
        rays[:, 0:3] = (0.0, 0.0, 0.3)
        rays[:, 3] = 1e-4
        rays[:, 4:7] = (1.0, 0.0, -0.05)
        rays[:, 7] = np.inf
        t0 = g.trace_closest(rays)[0]
        t1 = g1.trace_closest(rays)[0]
        assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
        assert torch.cuda.current_device() == before


def test_host_layer_gpu_count_and_cli(hiplib, tmp_path):
    """set_gpu_count / bfrender --gpus: 1 is the plain entry, more than the box has is an error (not a silent fallback)."""
    from beifong_amd import mitsuba
    from beifong_amd.mitsuba import _host
    from tests.test_host import TRANS_RAD_LIKE, HOST
    assert mitsuba.gpu_count() == 1
    with pytest.raises(_host.HostError):
        mitsuba.set_gpu_count(0)
    p = tmp_path / "scene.xml"
    p.write_text(TRANS_RAD_LIKE)
    out = tmp_path / "out.npy"
    r = subprocess.run([os.path.join(HOST, "bfrender"), "-m", "scalar_rgb", "-Dspp=5000", "--gpus", "1", "-o", str(out), str(p)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert np.load(out)[0, 0, 4] == 5000
    n = hiplib.bf_device_count()
    r = subprocess.run([os.path.join(HOST, "bfrender"), "-m", "scalar_rgb", "-Dspp=5000", "--gpus", str(n + 1), "-o", str(out), str(p)],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "GPUs requested" in r.stderr
    if n >= 2:          # a multi-GPU box: the sharded histogram is the one-GPU histogram
        mitsuba.set_variant("scalar_rgb")
        from beifong_amd.mitsuba.core.xml import load_string
        scene = load_string(TRANS_RAD_LIKE, spp=20000)
        sensor = scene.sensors()[0]
        scene.integrator().render(scene, sensor)
        one = np.array(sensor.film().bitmap(raw=True)).copy()
        mitsuba.set_gpu_count(2)
        try:
            scene.integrator().render(scene, sensor)
            two = np.array(sensor.film().bitmap(raw=True))
        finally:
            mitsuba.set_gpu_count(1)
        assert two[0, 0, 4] == one[0, 0, 4] and np.allclose(two, one, rtol=1e-4, atol=1e-3)
