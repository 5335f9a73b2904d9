"""Batched launches (bf_render_batch_device: many renders of one scene in ONE launch sequence) and scene clones
(bf_scene_clone: several handles on one geometry) against stand-alone renders and the oracle.

The reference runs sweeps one render() / receive() call per frame and rebuilds the scene in between
(python_scripts/animated_trans_rad.py:307-384, Receive.ipynb cell 30).  A batch must give every path of every render the
very record the stand-alone render gives (bit for bit), whichever kernel finishes the path (wf_shade chains, wf_trace,
the tail kernel's lane / quad / row traversals), with per-render seeds and per-render mesh offsets."""
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


@pytest.mark.parametrize("mega", [False, True])
def test_batch_of_seeds_equals_stand_alone_renders(hiplib, mega):
    """Range mode, five renders with their own seeds: records and histograms of the batch == the five renders, ==
    the oracle; the batch's LDS-privatised histogram covers all renders (5 x 261 channels)."""
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=30000, bins=256, dr=0.1)
    if mega:
        lp.flags |= capi.BF_FLAG_MEGAKERNEL
    g = capi.Scene(sd)
    seeds = [11, 7, 123456789, 11, 2 ** 40 + 5]
    hb, rb, sb = g.render_batch(lp, len(seeds), seeds=seeds, records=True)
    o = OracleScene(sd)
    rays = 0
    for k, seed in enumerate(seeds):
        l1 = capi.make_launch(lp.mode, lp.n_paths, seed=seed, bins=lp.bins, bin_width=lp.bin_width, color_mode=lp.color_mode,
                              flags=lp.flags)
        hs, rs, ss = g.render(l1, records=True)
        _same_records(rb[k], rs)
        _close_hist(hb[k], hs, lp.n_paths, float(np.abs(rs["L"]).max()))
        assert hb[k][4] == hs[4]
        rays += ss.n_rays_closest + ss.n_rays_shadow
        l1.flags = 0
        ho, ro, _ = o.render(l1, records=True, threads=8)
        _same_records(rb[k], ro)
    assert sb.n_rays_closest + sb.n_rays_shadow == rays and sb.n_paths == lp.n_paths * len(seeds)
    assert np.array_equal(rb[0]["L"], rb[3]["L"])          # same seed, same render


def test_batch_global_atomics_and_single_render(hiplib):
    """A batch too large for LDS privatisation (global float atomics), the explicit flag, and n_renders = 1."""
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=6000, bins=1024, dr=0.03)
    g = capi.Scene(sd)
    seeds = np.arange(16, dtype=np.uint64) * 1000003 + 5        # 16 x 1029 channels > 12288 floats of LDS
    hb, rb, _ = g.render_batch(lp, 16, seeds=seeds, records=True)
    lp2 = capi.make_launch(lp.mode, lp.n_paths, seed=lp.seed, bins=lp.bins, bin_width=lp.bin_width, color_mode=lp.color_mode,
                           flags=capi.BF_FLAG_GLOBAL_ATOMICS)
    hb2, rb2, _ = g.render_batch(lp2, 3, seeds=seeds[:3], records=True)
    for k in (0, 7, 15):
        l1 = capi.make_launch(lp.mode, lp.n_paths, seed=int(seeds[k]), bins=lp.bins, bin_width=lp.bin_width, color_mode=lp.color_mode)
        hs, rs, _ = g.render(l1, records=True)
        _same_records(rb[k], rs)
        _close_hist(hb[k], hs, lp.n_paths, float(np.abs(rs["L"]).max()))
        if k < 3:
            _same_records(rb2[k], rs)
            _close_hist(hb2[k], hs, lp.n_paths, float(np.abs(rs["L"]).max()))
    h1, r1, _ = g.render_batch(lp, 1, seeds=[99], records=True)
    l1 = capi.make_launch(lp.mode, lp.n_paths, seed=99, bins=lp.bins, bin_width=lp.bin_width, color_mode=lp.color_mode)
    hs, rs, _ = g.render(l1, records=True)
    _same_records(r1[0], rs)
    _close_hist(h1[0], hs, lp.n_paths, float(np.abs(rs["L"]).max()))


@pytest.mark.parametrize("iq", [False, True])
def test_batch_with_mesh_offsets_equals_translated_scenes(hiplib, iq):
    """Receive mode (the pulse sweep's): every render of the batch carries its own mesh offset, applied to the triangles on
    the fly; every path must equal the render of the scene translated by bf_scene_translate_meshes — which
    test_translate_meshes_equals_rebuilt_scene pins to a scene BUILT from the shifted vertices — and the oracle on
    that rebuilt scene."""
    mesh = scenes.bus_mesh(20000)
    sd, lp = scenes.bus_receive(n_tris=20000, n_paths=20000, t_bins=256, dr=0.1)
    if iq:
        lp.mode = capi.BF_MODE_RECEIVE_IQ
    offsets = np.array([[0, 0, 0], [0.013, -0.2, 0.05], [-3.0, 1.5, 0.25], [-0.005, 0, 0], [2.5, -4.0, 0.0]], np.float32)
    g = capi.Scene(sd)
    hb, rb, sb = g.render_batch(lp, len(offsets), offsets=offsets, records=True)
    g2 = capi.Scene(sd)
    for k, off in enumerate(offsets):
        g2.translate_meshes(off)
        hs, rs, _ = g2.render(lp, records=True)
        _same_records(rb[k], rs)
        _close_hist(hb[k], hs, lp.n_paths, float(np.abs(rs["L"]).max()))
    assert not np.array_equal(rb[0]["L"], rb[2]["L"])
    # per-render seeds AND offsets, against the oracle on a rebuilt scene
    from beifong_amd import meshgen
    seeds = [5, 6, 7, 8, 9]
    hb, rb, _ = g.render_batch(lp, len(offsets), seeds=seeds, offsets=offsets, records=True)
    k = 2
    v, f = meshgen.bus(20000, seed=1)
    v = meshgen.place(v, yaw_deg=-20.0, translate=(10.0, 3.0, 1.7)).astype(np.float32)
    v1 = np.ascontiguousarray((v + offsets[k][None, :]).astype(np.float32))
    sd1 = _bus_receive_with_mesh(v1, f)
    l1 = capi.make_launch(lp.mode, lp.n_paths, seed=seeds[k], bins=lp.bins, bins_y=1)
    _, ro, _ = OracleScene(sd1).render(l1, records=True, threads=8)
    _same_records(rb[k], ro)


def _bus_receive_with_mesh(v, f, t_bins=256, dr=0.1, lambda_band_nm=None):
    """scenes.bus_receive with the bus vertices replaced (same endpoints, materials, ADC)."""
    from beifong_amd import meshgen
    orig_bus, orig_place = meshgen.bus, meshgen.place
    try:
        meshgen.bus = lambda n, seed=1: (v, f)
        meshgen.place = lambda vv, yaw_deg=0.0, translate=(0, 0, 0): vv
        sd, _ = scenes.bus_receive(n_tris=len(f), n_paths=64, t_bins=t_bins, dr=dr, lambda_band_nm=lambda_band_nm)
    finally:
        meshgen.bus, meshgen.place = orig_bus, orig_place
    return sd


def test_batch_deep_tail_paths_with_offsets(hiplib):
    """Enough paths that the wavefront iterations hand a populated pool to the tail kernel (lane, quad and row traversals
    all see shifted meshes), range mode with vertex normals (make_si interpolates them at the shifted hit)."""
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=1 << 18, bins=256, dr=0.1)
    offsets = np.array([[0.0, 0.0, 0.0], [-0.35, 0.1, 0.0], [1.25, -0.5, 0.02]], np.float32)
    g = capi.Scene(sd)
    hb, rb, sb = g.render_batch(lp, 3, offsets=offsets, records=True)
    assert sb.n_rays_tail > 0
    g2 = capi.Scene(sd)
    for k, off in enumerate(offsets):
        g2.translate_meshes(off)
        hs, rs, _ = g2.render(lp, records=True)
        _same_records(rb[k], rs)
        _close_hist(hb[k], hs, lp.n_paths, float(np.abs(rs["L"]).max()))


def test_batch_rejects_what_it_cannot_do(hiplib):
    sd, lp = scenes.film_half_lit(film=(4, 2), spp=16)
    g = capi.Scene(sd)
    with pytest.raises(capi.BeifongError, match="multi-pixel"):
        g.render_batch(lp, 2)
    sd, lp = scenes.bus_radar(n_tris=2000, n_paths=100)
    g = capi.Scene(sd)
    with pytest.raises(capi.BeifongError, match="non-finite"):
        g.render_batch(lp, 2, offsets=[[0, 0, 0], [np.nan, 0, 0]])
    with pytest.raises(capi.BeifongError, match="n_renders"):
        g.render_batch(lp, 0)


def test_clone_shares_geometry_and_keeps_its_own_endpoints(hiplib):
    """bf_scene_clone: same renders as the original; device memory of a clone is the small tables only; an endpoint update
    or a translation of one handle does not leak into the other (copy on write); the clone survives its source."""
    torch = pytest.importorskip("torch")
    mesh = scenes.bus_mesh(20000)
    sd, lp = scenes.bus_radar(n_paths=20000, mesh=mesh)
    g = capi.Scene(sd)
    c = g.clone()
    _, r0, _ = g.render(lp, records=True)
    _, rc, _ = c.render(lp, records=True)
    _same_records(r0, rc)
    # endpoints of the clone move: the original is untouched
    sd_turned, _ = scenes.bus_radar(n_paths=20000, mesh=mesh, radar_yaw_deg=10.0)
    c.update_endpoints(sd_turned)
    _, rc2, _ = c.render(lp, records=True)
    _, r1, _ = g.render(lp, records=True)
    _same_records(r0, r1)
    _, rf, _ = capi.Scene(sd_turned).render(lp, records=True)
    _same_records(rc2, rf)
    # the original is translated: copy on write, the clone keeps the geometry as created
    off = [-0.4, 0.2, 0.0]
    g.translate_meshes(off)
    _, rt, _ = g.render(lp, records=True)
    assert not np.array_equal(rt["L"], r0["L"])
    _, rc3, _ = c.render(lp, records=True)
    _same_records(rc3, rc2)
    ref = capi.Scene(sd)
    ref.translate_meshes(off)
    _, rr, _ = ref.render(lp, records=True)
    _same_records(rt, rr)
    # a clone of the translated handle starts from what that handle renders now
    c2 = g.clone()
    _, rc4, _ = c2.render(lp, records=True)
    _same_records(rc4, rt)
    g.translate_meshes([0, 0, 0])
    _, rb, _ = g.render(lp, records=True)
    _same_records(rb, r0)
    _, rc5, _ = c2.render(lp, records=True)
    _same_records(rc5, rt)
    g.close()
    _, rc6, _ = c.render(lp, records=True)
    _same_records(rc6, rc2)


def test_clone_allocates_no_second_copy_of_the_geometry(hiplib):
    """C2's bus: triangles + vertex normals + four-wide nodes are ~26 MB; a clone pays for its small tables and its
    traversal-spill columns only."""
    torch = pytest.importorskip("torch")
    sd, lp = scenes.bus_radar(n_tris=200_000, n_paths=64)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    g = capi.Scene(sd)
    free1 = torch.cuda.mem_get_info()[0]
    c = g.clone()
    free2 = torch.cuda.mem_get_info()[0]
    info = g.info()
    geometry = info.n_triangles * 96 + info.n_bvh_nodes * 128
    assert (free0 - free1) - (free1 - free2) >= 0.9 * geometry, (free0 - free1, free1 - free2, geometry)
    _, r0, _ = g.render(lp, records=True)
    _, r1, _ = c.render(lp, records=True)
    _same_records(r0, r1)


def test_concurrent_renders_on_clones(hiplib):
    """One BVH, three handles, three streams: what bench.py and PulseSweeper do."""
    torch = pytest.importorskip("torch")
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=1 << 18)
    first = capi.Scene(sd)
    handles = [first, first.clone(), first.clone()]
    streams = [torch.cuda.Stream() for _ in handles]
    nch = first.channels(lp)
    ref, _, _ = first.render(lp)
    hists = [torch.zeros(nch, device="cuda") for _ in range(9)]
    for k, h in enumerate(hists):
        j = k % 3
        with torch.cuda.stream(streams[j]):
            h.zero_()
            handles[j].render_device(lp, h.data_ptr(), stream=streams[j].cuda_stream)
    torch.cuda.synchronize()
    for h in hists:
        assert np.allclose(h.cpu().numpy(), ref, rtol=1e-4, atol=1e-2)


def test_pulse_sweeper_batched_equals_per_pulse(hiplib):
    """PulseSweeper renders the sweep as batches; per_pulse=True is the translate-and-render loop of round 1."""
    pytest.importorskip("torch")
    from beifong_amd import sweep
    sd, lp = scenes.bus_receive(n_tris=20000, n_paths=1 << 15, t_bins=256, dr=0.1, seed=4)
    lp.mode = capi.BF_MODE_RECEIVE_IQ
    offsets = np.zeros((12, 3), np.float32)
    offsets[:, 0] = -0.005 * np.arange(12)
    sw = sweep.PulseSweeper(sd, lp, n_streams=2)
    a = sw.render(offsets)
    b = sw.render(offsets, per_pulse=True)
    c = sw.render(offsets)
    sw.close()
    assert a.shape == (12, 256, 3) and np.all(a[:, :, 2].sum(1) == lp.n_paths)
    scale = np.abs(b).max()
    assert np.allclose(a, b, rtol=1e-4, atol=1e-5 * scale) and np.allclose(a, c, rtol=1e-4, atol=1e-5 * scale)
