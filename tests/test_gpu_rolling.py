"""Rolling sequences (BF_FLAG_ROLLING + bf_scene_flush): consecutive renders of one handle leave their long paths alive in
the pool, the next renders' launches carry them, one tail runs per sequence (include/beifong_hip.h).

The reference runs such loops one render() / receive() per frame (python_scripts/animated_trans_rad.py:307-384,
Receive.ipynb cell 30; sample loop src/librender/integrator.cpp:659-663).  Whatever launch advances a path, it must be
the path a stand-alone render traces: every per-path record of every render of a sequence is held to the stand-alone
render's and to the oracle's, bit for bit, and every histogram is complete after the flush."""
import ctypes as C
import os

import numpy as np
import pytest

from beifong_amd import capi, scenes
from tests.oracle_lib import OracleScene

pytestmark = pytest.mark.gpu


def _same_records(a, b):
    for k in ("L", "aux"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    assert np.array_equal(a["n_rays"], b["n_rays"]) and np.array_equal(a["valid"], b["valid"])


def _close_hist(hb, hs, n_paths, amax):
    atol = n_paths * 2.0 ** -24 * max(amax, 1.0) * 4
    assert np.allclose(hb, hs, rtol=2e-5, atol=atol), float(np.abs(hb - hs).max())


def _launch_like(lp, seed, flags=0, n_paths=None, path_offset=0):
    return capi.make_launch(lp.mode, int(lp.n_paths if n_paths is None else n_paths), seed=seed, path_offset=path_offset, bins=lp.bins,
                            bins_y=lp.bins_y, bin_width=lp.bin_width, color_mode=lp.color_mode, max_depth=lp.max_depth,
                            rr_depth=lp.rr_depth, time_c=lp.time_c, phase_bins=lp.phase_bins, flags=flags)


class _Sequence:
    """K rolling renders of one handle with device buffers (torch), then a flush."""

    def __init__(self, g, lp, seeds, offsets=None, extra_flags=0, records=True):
        import torch
        self.torch = torch
        self.g, self.lp, self.seeds = g, lp, list(seeds)
        K, n = len(self.seeds), g.channels(lp)
        self.hist = torch.zeros((K, n), dtype=torch.float32, device="cuda")
        self.rec = torch.zeros((K, int(lp.n_paths), 4), dtype=torch.int32, device="cuda") if records else None
        self.offsets = list(offsets) if offsets is not None else [0] * K
        self.flags = capi.BF_FLAG_ROLLING | extra_flags

    def issue(self, ks=None, stream=0):
        for k in (range(len(self.seeds)) if ks is None else ks):
            l = _launch_like(self.lp, self.seeds[k], flags=self.flags | self.lp.flags, path_offset=self.offsets[k])
            self.g.render_device(l, self.hist[k].data_ptr(), stream=stream,
                                 records_ptr=self.rec[k].data_ptr() if self.rec is not None else None)

    def results(self):
        self.torch.cuda.synchronize()
        h = self.hist.cpu().numpy()
        r = self.rec.cpu().numpy().view(np.uint32).reshape(len(self.seeds), -1, 4) if self.rec is not None else None
        recs = None
        if r is not None:
            recs = [np.ascontiguousarray(r[k]).view(capi.PATH_RECORD_DTYPE).reshape(-1) for k in range(len(self.seeds))]
        return h, recs


def _check_against_stand_alone(g, lp, seq, h, recs, oracle=None, offsets=None):
    rays = 0
    for k, seed in enumerate(seq.seeds):
        l1 = _launch_like(lp, seed, flags=lp.flags, path_offset=seq.offsets[k])
        hs, rs, ss = g.render(l1, records=True)
        _same_records(recs[k], rs)
        _close_hist(h[k], hs, lp.n_paths, float(np.abs(rs["L"]).max()))
        rays += ss.n_rays_closest + ss.n_rays_shadow
        if oracle is not None:
            l1.flags = 0
            _, ro, _ = oracle.render(l1, records=True, threads=8)
            _same_records(recs[k], ro)
    return rays


@pytest.mark.parametrize("iters", ["", "1", "5"])
def test_rolling_range_sequence_every_path_bit_exact(hiplib, iters, monkeypatch):
    """C2-class scene, range mode: eight rolling renders with their own seeds; per-path records == stand-alone renders ==
    the oracle, histograms complete after the flush, the sequence's counters == the sum of the renders'.  BF_ROLL_ITERS=1
    lets the backlog of late-started paths grow until the flush; 5 runs most iterations over a nearly idle pool."""
    if iters:
        monkeypatch.setenv("BF_ROLL_ITERS", iters)
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=1 << 16, bins=256, dr=0.1)
    g = capi.Scene(sd)
    seeds = [11, 7, 123456789, 11, 2 ** 40 + 5, 3, 4, 5]
    seq = _Sequence(g, lp, seeds, extra_flags=capi.BF_FLAG_STATS)
    seq.issue()
    st = g.flush(want_stats=True)
    h, recs = seq.results()
    rays = _check_against_stand_alone(g, lp, seq, h, recs, oracle=OracleScene(sd))
    assert st.n_rays_closest + st.n_rays_shadow == rays and st.n_paths == lp.n_paths * len(seeds)
    assert st.n_guard == 0 and st.n_launches_tail <= 1
    assert np.array_equal(recs[0]["L"], recs[3]["L"])          # same seed, same render
    assert all(hk[4] == lp.n_paths for hk in h)                # the weight channel: every path of every render landed


@pytest.mark.parametrize("bins,iters", [(4096, "1"), (4096, ""), (256, "1")])
def test_rolling_heavy_eviction_loses_no_path(hiplib, bins, iters, monkeypatch):
    """Twelve renders of 2^17 paths with one bounce iteration per call: a third of every render's paths moves to the
    survivor area.  Regression: with fewer survivor batches than shading waves two waves could claim the same batch in
    one launch and hand its free slots out twice (~600 of 131072 paths per render lost).  4096 bins leave an LDS window
    of two renders, so most late samples also take the base-channel table / global-atomic route."""
    if iters:
        monkeypatch.setenv("BF_ROLL_ITERS", iters)
    n = 1 << 17
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=n, bins=bins, dr=25.6 / bins)
    g = capi.Scene(sd)
    seeds = list(range(100, 112))
    seq = _Sequence(g, lp, seeds)
    seq.issue()
    g.flush()
    h, recs = seq.results()
    assert [float(hk[4]) for hk in h] == [float(n)] * len(seeds)
    for k in (0, 5, 11):
        hs, rs, _ = g.render(_launch_like(lp, seeds[k]), records=True)
        _same_records(recs[k], rs)
        _close_hist(h[k], hs, n, float(np.abs(rs["L"]).max()))


def test_rolling_survivor_area_overrun_is_loud_and_loses_nothing(hiplib, monkeypatch):
    """The invariant behind the regression above — the survivor batches ONE launch claims are distinct — is checked on the
    device (surv_take: a per-launch claim count; CTR_SURV_GUARD).  BF_DEBUG_SURV_BATCHES shrinks the survivor area to 16
    batches and switches the sizing rule off: the same heavy-eviction workload now asks for far more claims per launch than
    the area has batches.  The claims beyond the area are REFUSED (those paths stay in their slots), so every path is still
    the stand-alone render's, and the flush with statistics reports the violation as BF_ERR_DEVICE."""
    monkeypatch.setenv("BF_ROLL_ITERS", "1")
    monkeypatch.setenv("BF_DEBUG_SURV_BATCHES", "16")
    n = 1 << 17
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=n, bins=256, dr=0.1)
    g = capi.Scene(sd)
    seeds = list(range(100, 106))
    seq = _Sequence(g, lp, seeds)
    seq.issue()
    with pytest.raises(capi.BeifongError) as e:
        g.flush(want_stats=True)
    assert "survivor" in str(e.value)
    g.sync()                                   # reported once: the sticky word is clear again
    h, recs = seq.results()
    assert [float(hk[4]) for hk in h] == [float(n)] * len(seeds)          # nothing lost
    monkeypatch.delenv("BF_DEBUG_SURV_BATCHES")
    g2 = capi.Scene(sd)
    for k in (0, 3, 5):
        hs, rs, _ = g2.render(_launch_like(lp, seeds[k]), records=True)
        _same_records(recs[k], rs)


def test_rolling_flush_statistics_count_every_path(hiplib):
    """bf_scene_flush with statistics checks that every path of every render of the sequence was binned exactly once
    (CTR_FILM == paths supplied); a healthy sequence passes and reports the sequence's totals."""
    n = 1 << 15
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=n, bins=256, dr=0.1)
    g = capi.Scene(sd)
    seq = _Sequence(g, lp, [1, 2, 3, 4, 5], extra_flags=capi.BF_FLAG_COUNT)
    seq.issue()
    st = g.flush(want_stats=True)
    assert st.n_paths == 5 * n and st.n_guard == 0 and st.n_rays_closest >= 5 * n


def test_rolling_receive_sequence_and_second_sequence(hiplib):
    """gen-3 receive (the RX = 1 kernels) with I/Q; a second sequence on the same handle after the flush starts afresh."""
    sd, lp = scenes.bus_receive(n_tris=20000, n_paths=20000, t_bins=256, dr=0.1)
    lp.mode = capi.BF_MODE_RECEIVE_IQ
    g = capi.Scene(sd)
    o = OracleScene(sd)
    for seeds in ([5, 6, 7, 8, 9], [100, 5]):
        seq = _Sequence(g, lp, seeds)
        seq.issue()
        g.flush()
        h, recs = seq.results()
        _check_against_stand_alone(g, lp, seq, h, recs, oracle=o)


@pytest.mark.parametrize("n_paths", [1, 63, 1000, 4097])
def test_rolling_ragged_pools(hiplib, n_paths):
    """Path counts that are no multiple of 64: the pool's spare slots belong to the NEXT render's paths (slot i renders the
    global paths i, i + n_slots, ...), so renders overlap inside the pool."""
    sd, lp = scenes.bus_radar(n_tris=5000, n_paths=n_paths, bins=64, dr=0.4)
    g = capi.Scene(sd)
    seeds = list(range(50, 57))
    seq = _Sequence(g, lp, seeds)
    seq.issue()
    g.flush()
    h, recs = seq.results()
    _check_against_stand_alone(g, lp, seq, h, recs, oracle=OracleScene(sd))


def test_rolling_sequence_longer_than_the_descriptor_ring(hiplib):
    """300 renders: the 257th flushes the first 256 behind the scenes (kRollRing descriptors); every render is complete."""
    sd, lp = scenes.bus_radar(n_tris=5000, n_paths=512, bins=64, dr=0.4)
    g = capi.Scene(sd)
    seeds = [1000 + 17 * k for k in range(300)]
    seq = _Sequence(g, lp, seeds)
    seq.issue()
    g.flush()
    h, recs = seq.results()
    assert all(hk[4] == lp.n_paths for hk in h)
    o = OracleScene(sd)
    for k in (0, 1, 255, 256, 257, 299):
        l1 = _launch_like(lp, seeds[k])
        _, ro, _ = o.render(l1, records=True, threads=4)
        _same_records(recs[k], ro)


def test_rolling_path_offsets_shard_a_render(hiplib):
    """The shards of one render as a rolling sequence (path_offset per render): the union is the unsharded sample set."""
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=1 << 14, bins=256, dr=0.1)
    g = capi.Scene(sd)
    K = 4
    seq = _Sequence(g, lp, [lp.seed] * K, offsets=[k * int(lp.n_paths) for k in range(K)])
    seq.issue()
    g.flush()
    h, recs = seq.results()
    whole = _launch_like(lp, lp.seed, n_paths=K * int(lp.n_paths))
    hw, rw, _ = g.render(whole, records=True)
    _same_records(np.concatenate(recs), rw)
    _close_hist(h.sum(axis=0), hw, K * lp.n_paths, float(np.abs(rw["L"]).max()))


def test_everything_else_flushes_an_open_sequence(hiplib):
    """A plain render, an endpoint update, a translation and a clone each finish the open sequence first: its paths see the
    scene they were issued for, and its histograms are complete once the other call's stream work is."""
    import torch
    sd, lp = scenes.bus_receive(n_tris=20000, n_paths=8192, t_bins=256, dr=0.1)
    g = capi.Scene(sd)
    ref_h, ref_r = {}, {}
    for seed in (1, 2, 3, 4):
        ref_h[seed], ref_r[seed], _ = g.render(_launch_like(lp, seed), records=True)

    def run(seed, then):
        seq = _Sequence(g, lp, [seed])
        seq.issue()
        then()
        h, recs = seq.results()              # no explicit flush
        _same_records(recs[0], ref_r[seed])
        _close_hist(h[0], ref_h[seed], lp.n_paths, float(np.abs(ref_r[seed]["L"]).max()))

    run(1, lambda: g.render(_launch_like(lp, 77)))
    sd2, _ = scenes.bus_receive(n_tris=20000, n_paths=8192, t_bins=256, dr=0.1)
    # (round 4: an endpoint update JOINS the open sequence instead of finishing it — test_rolling_sequence_across_endpoint_updates —
    # unless BF_ROLL_JOIN=0 or the scene cannot: here the explicit flush ends it)
    run(2, lambda: (g.update_endpoints(sd2), g.flush()))
    run(3, lambda: g.translate_meshes((0.0, 0.0, 0.0)))
    run(4, lambda: g.clone())
    # a render of another shape opens a new sequence behind the flushed one
    seq_a = _Sequence(g, lp, [1])
    seq_a.issue()
    lp_b = _launch_like(lp, 2, n_paths=4096)
    seq_b = _Sequence(g, lp_b, [2])
    seq_b.issue()
    g.flush()
    ha, ra = seq_a.results()
    _same_records(ra[0], ref_r[1])
    hb, rb = seq_b.results()
    _, rs, _ = g.render(_launch_like(lp_b, 2), records=True)
    _same_records(rb[0], rs)
    torch.cuda.synchronize()


@pytest.mark.parametrize("mode,iters", [("range", ""), ("range", "1"), ("receive", ""), ("receive_iq", "1")])
def test_rolling_sequence_across_endpoint_updates(hiplib, mode, iters, monkeypatch):
    """The loop the reference ships (python_scripts/animated_trans_rad.py:307-384: 73 frames, Receive.ipynb cell 30: 256 angles):
    the radar TURNS between frames.  bf_scene_update_endpoints joins the open rolling sequence — the new endpoint tables
    become the next version in the handle's pool, every path reads the tables of ITS frame through the descriptor ring
    (kMulti kernels) — so the whole sweep is ONE sequence with ONE tail.  Every per-path record of every frame equals the
    oracle's on the scene REBUILT for that frame; histograms equal stand-alone renders of fresh scenes; after the flush the
    handle renders with the last frame's endpoints.  BF_ROLL_ITERS=1 keeps many paths of earlier frames alive while later
    frames start (versions mixed inside the waves, the survivor area and the flush's tail)."""
    import torch
    if iters:
        monkeypatch.setenv("BF_ROLL_ITERS", iters)
    n = 1 << 15
    yaws = [0.0, 6.0, 12.0, -9.0, 3.0]
    if mode == "range":
        mesh = scenes.bus_mesh(20000)
        frames = [scenes.bus_radar(n_paths=n, bins=256, dr=0.1, radar_yaw_deg=y, mesh=mesh) for y in yaws]
    else:
        from beifong_amd import meshgen
        mesh = meshgen.bus(20000, seed=1)
        frames = [scenes.bus_receive(n_paths=n, t_bins=256, dr=0.1, radar_yaw_deg=y, mesh=mesh) for y in yaws]
        if mode == "receive_iq":
            for _, l in frames:
                l.mode = capi.BF_MODE_RECEIVE_IQ
    sd0, lp0 = frames[0]
    g = capi.Scene(sd0)
    K, nch = len(frames), g.channels(lp0)
    hist = torch.zeros((K, nch), dtype=torch.float32, device="cuda")
    rec = torch.zeros((K, n, 4), dtype=torch.int32, device="cuda")
    for k, (sd, lp) in enumerate(frames):
        if k:
            g.update_endpoints(sd)
        l = _launch_like(lp, 1000 + k, flags=capi.BF_FLAG_ROLLING | capi.BF_FLAG_COUNT)
        g.render_device(l, hist[k].data_ptr(), records_ptr=rec[k].data_ptr())
    st = g.flush(want_stats=True)
    if os.environ.get("BF_ROLL_JOIN", "1") != "0":         # (BF_ROLL_JOIN=0, round 3's behaviour: every update flushes; same results)
        assert st.n_paths == K * n and st.n_launches_tail <= 1          # ONE sequence: the updates did not flush it
    torch.cuda.synchronize()
    h = hist.cpu().numpy()
    r = rec.cpu().numpy().view(np.uint32).reshape(K, -1, 4)
    for k, (sd, lp) in enumerate(frames):
        recs = np.ascontiguousarray(r[k]).view(capi.PATH_RECORD_DTYPE).reshape(-1)
        l1 = _launch_like(lp, 1000 + k)
        _, ro, _ = OracleScene(sd).render(l1, records=True, threads=8)          # the scene REBUILT for frame k
        _same_records(recs, ro)
        hs, rs, _ = capi.Scene(sd).render(l1, records=True)                      # a fresh handle, stand-alone
        _same_records(recs, rs)
        _close_hist(h[k], hs, n, float(np.abs(rs["L"]).max()))
    # the handle's tables are the last frame's again (home buffers): a plain render sees them
    _, rl, _ = g.render(_launch_like(frames[-1][1], 77), records=True)
    _, rf, _ = capi.Scene(frames[-1][0]).render(_launch_like(frames[-1][1], 77), records=True)
    _same_records(rl, rf)


def test_endpoint_update_flushes_when_it_cannot_join(hiplib, monkeypatch):
    """BF_ROLL_JOIN=0 (round 3's behaviour) and scenes that cannot carry table versions (phased arrays: their element
    tables are replaced in place) finish the open sequence at the update: same results, one tail per frame."""
    import torch
    monkeypatch.setenv("BF_ROLL_JOIN", "0")
    n = 1 << 14
    mesh = scenes.bus_mesh(20000)
    frames = [scenes.bus_radar(n_paths=n, bins=256, dr=0.1, radar_yaw_deg=y, mesh=mesh) for y in (0.0, 8.0, -8.0)]
    g = capi.Scene(frames[0][0])
    hist = torch.zeros((3, g.channels(frames[0][1])), dtype=torch.float32, device="cuda")
    rec = torch.zeros((3, n, 4), dtype=torch.int32, device="cuda")
    for k, (sd, lp) in enumerate(frames):
        if k:
            g.update_endpoints(sd)
        g.render_device(_launch_like(lp, 5 + k, flags=capi.BF_FLAG_ROLLING), hist[k].data_ptr(), records_ptr=rec[k].data_ptr())
    g.flush()
    torch.cuda.synchronize()
    r = rec.cpu().numpy().view(np.uint32).reshape(3, -1, 4)
    for k, (sd, lp) in enumerate(frames):
        _, rs, _ = capi.Scene(sd).render(_launch_like(lp, 5 + k), records=True)
        _same_records(np.ascontiguousarray(r[k]).view(capi.PATH_RECORD_DTYPE).reshape(-1), rs)


def test_rolling_rejects_what_cannot_roll(hiplib):
    sd, lp = scenes.bus_radar(n_tris=5000, n_paths=4096, bins=64, dr=0.4)
    g = capi.Scene(sd)
    l = _launch_like(lp, 1, flags=capi.BF_FLAG_ROLLING)
    hist = np.zeros(g.channels(l), np.float32)
    st = capi.bf_stats()
    # host-buffer renders and stats requests are synchronous by nature
    assert hiplib.bf_render(g.handle, C.byref(l), hist.ctypes.data_as(C.c_void_p), None, C.byref(st)) == capi.BF_ERR_INVALID
    l2 = _launch_like(lp, 1, flags=capi.BF_FLAG_ROLLING | capi.BF_FLAG_MEGAKERNEL)
    import torch
    d = torch.zeros(g.channels(l), dtype=torch.float32, device="cuda")
    assert hiplib.bf_render_device(g.handle, C.byref(l2), C.c_void_p(d.data_ptr()), None, None, None) == capi.BF_ERR_UNSUPPORTED
    assert g.flush(want_stats=True).n_paths == 0           # nothing open: a no-op


def test_guard_trip_of_the_last_render_is_reported_by_sync(hiplib):
    """a planned render returns BF_OK before its kernels ran, so a guard trip of the LAST render of a sequence
    used to go unreported.  The guard word is sticky now and bf_scene_sync reports it (test hook: pre-load the word)."""
    import torch
    sd, lp = scenes.bus_radar(n_tris=5000, n_paths=1 << 14, bins=64, dr=0.4)
    g = capi.Scene(sd)
    d = torch.zeros(g.channels(lp), dtype=torch.float32, device="cuda")
    for k in range(3):                                      # the first learns the plan, the others are planned (asynchronous)
        g.render_device(_launch_like(lp, k), d.data_ptr())
    g.sync()                                                # clean
    hiplib.bfdbg_preload_guard.argtypes = [C.c_void_p, C.c_ulonglong]
    assert hiplib.bfdbg_preload_guard(g.handle, 5) == capi.BF_OK
    g.render_device(_launch_like(lp, 9), d.data_ptr())      # planned: returns BF_OK, does not clear the word
    with pytest.raises(capi.BeifongError, match="dropped 5 rays"):
        g.sync()
    g.sync()                                                # reported once, then clean again
    # the synchronous path (stats) reports it too
    assert hiplib.bfdbg_preload_guard(g.handle, 2) == capi.BF_OK
    with pytest.raises(capi.BeifongError, match="dropped 2 rays"):
        g.render(_launch_like(lp, 1))
    g.render(_launch_like(lp, 1))


def test_one_host_thread_per_handle(hiplib):
    """A second host thread inside a call on the same handle gets BF_ERR_INVALID instead of a race on the handle's pool
    (test hook holds the flag the way a concurrent call would)."""
    sd, lp = scenes.bus_radar(n_tris=5000, n_paths=4096, bins=64, dr=0.4)
    g = capi.Scene(sd)
    hiplib.bfdbg_hold_busy.argtypes = [C.c_void_p, C.c_int]
    assert hiplib.bfdbg_hold_busy(g.handle, 1) == capi.BF_OK
    with pytest.raises(capi.BeifongError, match="in use by another host thread"):
        g.render(_launch_like(lp, 1))
    with pytest.raises(capi.BeifongError, match="in use by another host thread"):
        g.flush()
    with pytest.raises(capi.BeifongError, match="in use by another host thread"):
        g.clone()
    assert hiplib.bfdbg_hold_busy(g.handle, 0) == capi.BF_OK
    g.render(_launch_like(lp, 1))
    c = g.clone()                                           # the clone has its own flag
    assert hiplib.bfdbg_hold_busy(g.handle, 1) == capi.BF_OK
    c.render(_launch_like(lp, 1))
    assert hiplib.bfdbg_hold_busy(g.handle, 0) == capi.BF_OK


def test_renders_of_one_handle_on_two_streams_are_ordered(hiplib):
    """Successive renders of ONE handle on different streams used to race on the pool (a header comment asked callers not
    to); the second now waits on the device for the first."""
    import torch
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=1 << 18, bins=256, dr=0.1)
    g = capi.Scene(sd)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    n = g.channels(lp)
    d = torch.zeros((6, n), dtype=torch.float32, device="cuda")
    r = torch.zeros((6, int(lp.n_paths), 4), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for k in range(6):
        s = s1 if k % 2 == 0 else s2
        g.render_device(_launch_like(lp, 40 + k), d[k].data_ptr(), stream=s.cuda_stream, records_ptr=r[k].data_ptr())
    g.sync()
    torch.cuda.synchronize()
    rr = r.cpu().numpy().view(np.uint32)
    for k in (0, 1, 5):
        _, rs, _ = g.render(_launch_like(lp, 40 + k), records=True)
        _same_records(np.ascontiguousarray(rr[k]).view(capi.PATH_RECORD_DTYPE).reshape(-1), rs)


@pytest.mark.parametrize("iq", [False, True])
def test_rolling_batches_with_mesh_offsets(hiplib, iq):
    """Batched calls join a rolling sequence (the pulses of a sweep, K per call, each with its own seed and mesh offset:
    BASELINE configs[4]): every path equals the plain batched launch's, which test_gpu_batch.py pins to the oracle on
    scenes BUILT from the shifted vertices."""
    import torch
    sd, lp = scenes.bus_receive(n_tris=20000, n_paths=20000, t_bins=256, dr=0.1)
    if iq:
        lp.mode = capi.BF_MODE_RECEIVE_IQ
    K, calls = 4, 3
    rng = np.random.default_rng(3)
    offsets = np.concatenate([np.zeros((1, 3)), rng.uniform(-0.02, 0.02, (5, 3)), rng.uniform(-2.5, 2.5, (6, 3))]).astype(np.float32)
    seeds = [int(x) for x in rng.integers(1, 1 << 40, K * calls)]
    g = capi.Scene(sd)
    n = g.channels(lp)
    hist = torch.zeros((K * calls, n), dtype=torch.float32, device="cuda")
    rec = torch.zeros((K * calls, int(lp.n_paths), 4), dtype=torch.int32, device="cuda")
    lr = _launch_like(lp, lp.seed, flags=capi.BF_FLAG_ROLLING)
    for c in range(calls):
        k0 = c * K
        g.render_batch_device(lr, K, hist[k0].data_ptr(), seeds=seeds[k0:k0 + K], offsets=offsets[k0:k0 + K],
                              records_ptr=rec[k0].data_ptr())
    g.flush()
    g.sync()
    h = hist.cpu().numpy()
    rr = rec.cpu().numpy().view(np.uint32)
    g2 = capi.Scene(sd)
    hb, rb, _ = g2.render_batch(lp, K * calls, seeds=seeds, offsets=offsets, records=True)
    for k in range(K * calls):
        got = np.ascontiguousarray(rr[k]).view(capi.PATH_RECORD_DTYPE).reshape(-1)
        _same_records(got, rb[k])
        _close_hist(h[k], hb[k], lp.n_paths, float(np.abs(rb[k]["L"]).max()))
    assert not np.array_equal(rb[0]["L"], rb[7]["L"])
    # a rolling batch without offsets after one with offsets opens a new sequence (the shape differs) and is correct too
    g.render_batch_device(lr, K, hist[0].data_ptr(), seeds=seeds[:K], records_ptr=rec[0].data_ptr())
    g.flush()
    g.sync()
    hb0, rb0, _ = g2.render_batch(lp, K, seeds=seeds[:K], records=True)
    got = np.ascontiguousarray(rec.cpu().numpy().view(np.uint32)[2]).view(capi.PATH_RECORD_DTYPE).reshape(-1)
    _same_records(got, rb0[2])


@pytest.mark.parametrize("seed", range(10))
@pytest.mark.parametrize("receive", [False, True])
def test_rolling_fuzz_scenes(hiplib, seed, receive):
    """The random scenes of test_gpu_parity (every integrator mode, depth and roulette limits, several emitters, phase
    bins, the Doppler hook / mix_resample on odd receive seeds) as rolling sequences of five renders: records and
    histograms of every render against the stand-alone render (which the fuzz tests hold to the oracle), and the first and
    last against the oracle directly."""
    from tests.test_gpu_parity import _fuzz_scene
    sd, lp = _fuzz_scene(seed, receive=receive)
    if lp.spp:
        pytest.skip("multi-pixel films do not roll")
    if receive and seed % 2:
        lp.flags = capi.BF_FLAG_DOPPLER | (capi.BF_FLAG_MIX_RESAMPLE if sd.sensor.type == capi.BF_RECEIVER_OMNI else 0)
    g = capi.Scene(sd)
    seeds = [int(lp.seed) + 1000 * k for k in range(5)]
    seq = _Sequence(g, lp, seeds)
    seq.issue()
    g.flush()
    h, recs = seq.results()
    o = OracleScene(sd)
    for k, s in enumerate(seeds):
        l1 = _launch_like(lp, s, flags=lp.flags)
        hs, rs, _ = g.render(l1, records=True)
        _same_records(recs[k], rs)
        amax = float(np.nanmax(np.abs(rs["L"]))) if len(rs) else 0.0
        assert np.allclose(h[k], hs, rtol=2e-5, atol=lp.n_paths * 2.0 ** -24 * max(amax, 1.0) * 4, equal_nan=True), (seed, k)
        if k in (0, 4):
            _, ro, _ = o.render(l1, records=True, threads=8)
            _same_records(recs[k], ro)


@pytest.mark.parametrize("share", ["", "1", "8"])
def test_clones_rolling_side_by_side_share_the_grids(hiplib, share, monkeypatch):
    """Small pools of handles that roll at the same time launch a share of the persistent grids each (bf_api.cpp: wf_setup,
    BF_GRID_SHARE: default 3, 1 = off, 8 = an eighth of the grids with eight peers... here four): three clones and their source,
    one stream and one rolling sequence each, renders issued round-robin — every per-path record equals the oracle's whatever
    grid a launch ran on, and every path of every render lands."""
    import torch
    if share:
        monkeypatch.setenv("BF_GRID_SHARE", share)
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=1 << 15, bins=256, dr=0.1)
    first = capi.Scene(sd)
    hs = [first] + [first.clone() for _ in range(3)]
    streams = [torch.cuda.Stream() for _ in hs]
    seeds = [[100 * j + k for k in range(3)] for j in range(len(hs))]
    seqs = [_Sequence(h, lp, seeds[j], extra_flags=capi.BF_FLAG_COUNT) for j, h in enumerate(hs)]
    for k in range(3):
        for j, q in enumerate(seqs):
            q.issue(ks=[k], stream=streams[j].cuda_stream)
    stats = [h.flush(stream=streams[j].cuda_stream, want_stats=True) for j, h in enumerate(hs)]
    oracle = OracleScene(sd)
    for j, q in enumerate(seqs):
        h, recs = q.results()
        assert stats[j].n_paths == 3 * lp.n_paths and stats[j].n_guard == 0
        for k, seed in enumerate(seeds[j]):
            l1 = _launch_like(lp, seed)
            _, ro, _ = oracle.render(l1, records=True, threads=8)
            _same_records(recs[k], ro)
            assert h[k][4] == lp.n_paths
    for h in hs[1:] + hs[:1]:
        h.close()
