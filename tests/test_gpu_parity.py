"""Parity of the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs.  Bit-exact for t / primitive / barycentrics / ray counts
and per-path radiance; fp32-summation tolerance for the atomically accumulated
histograms (stated at each assert)."""
import math

import os

import numpy as np
import pytest

from beifong_amd import capi, meshgen, scenes
from beifong_amd.scenedesc import SceneDesc, Transform4f
from tests.oracle_lib import OracleScene

pytestmark = pytest.mark.gpu
f32 = np.float32
EPS = f32(1500 * 2.0 ** -24)


def _rays(n, seed, extent=1.5):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-extent, extent, (n, 3))
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, np.full((n, 1), EPS), d, np.full((n, 1), np.inf)], 1).astype(f32)


def _compare_trace(sd, rays):
    g = capi.Scene(sd)
    o = OracleScene(sd)
    tg, pg, sg, ug = g.trace_closest(rays)
    to, po, so, uo = o.trace_closest(rays)
    assert np.array_equal(tg.view(np.uint32), to.view(np.uint32))
    assert np.array_equal(pg, po) and np.array_equal(sg, so)
    hit = np.isfinite(to)
    assert np.array_equal(ug[hit].view(np.uint32), uo[hit].view(np.uint32))
    assert np.array_equal(g.trace_any(rays), o.trace_any(rays))
    return hit.sum()


def test_trace_stairs_known_answer(hiplib):
    # src/librender/tests/test_kdtrees.py:25-57 through the HIP traversal
    n_steps = 20
    v, f = meshgen.stairs(n_steps)
    g = capi.Scene(scenes.single_mesh(v, f))
    n = 128
    inv_n = 1.0 / (n - 1)
    rays, exp = [], []
    for x in range(n - 1):
        for y in range(n - 1):
            rays.append([x * inv_n, y * inv_n, 2, 0, 0, 0, -1, 100])
            exp.append(2.0 - math.floor((y * inv_n) * n_steps) / n_steps)
    rays = np.array(rays, f32)
    t, prim, shape, uv = g.trace_closest(rays)
    assert np.all(g.trace_any(rays) == 1)
    assert np.allclose(t, np.array(exp, f32), atol=1e-6)


def test_trace_rectangle_known_answer(hiplib):
    # src/shapes/tests/test_rectangle.py:37-63
    sd = SceneDesc()
    sd.add_rectangle(Transform4f.scale([2.0, 0.5, 1.0]), sd.add_diffuse(0.5))
    sd.set_perspective(Transform4f())
    g = capi.Scene(sd.finalize())
    coords = np.linspace(-1, 1, 15, dtype=f32)
    rays = np.array([[a, a, 5, EPS, 0, 0, -1, np.inf] for a in coords], f32)
    t, prim, shape, uv = g.trace_closest(rays)
    assert np.array_equal(np.isfinite(t), np.abs(coords) <= 0.5)
    assert np.isfinite(t).sum() == 7 and np.array_equal(g.trace_any(rays).astype(bool), np.isfinite(t))


@pytest.mark.parametrize("n_tris,seed", [(1, 1), (7, 2), (2000, 3), (50000, 4)])
def test_trace_soup_bit_exact(hiplib, n_tris, seed):
    v, f = meshgen.triangle_soup(n_tris, seed=seed)
    nhit = _compare_trace(scenes.single_mesh(v, f), _rays(20000, seed + 100))
    assert n_tris < 100 or nhit > 1000


def test_trace_empty_scene_and_axis_aligned_rays(hiplib):
    sd = SceneDesc()
    sd.add_diffuse(0.5)
    sd.set_perspective(Transform4f())
    g = capi.Scene(sd.finalize())
    t, prim, shape, uv = g.trace_closest(_rays(100, 1))
    assert np.all(np.isinf(t)) and np.all(prim == 0xffffffff)
    # axis-aligned directions have zero components (inf reciprocals in the slab test)
    v, f = meshgen.stairs(8)
    rays = np.array([[0.5, 0.5, 2, 0, 0, 0, -1, 100], [0.5, -1, 0.01, 0, 0, 1, 0, 100], [-1, 0.3, 0.2, 0, 1, 0, 0, 100]], f32)
    _compare_trace(scenes.single_mesh(v, f), rays)


def test_trace_mixed_rect_and_mesh_scene(hiplib):
    sd, _ = scenes.bus_radar(n_tris=20000, n_paths=1)
    rays = _rays(20000, 9, extent=3.0)
    rays[:, 0] += 8.0
    rays[:, 2] = np.abs(rays[:, 2]) + 0.2
    assert _compare_trace(sd, rays) > 2000


def _render_compare(sd, lp, hist_rtol=2e-5):
    """Both device pipelines (wavefront = default, megakernel = ablation flag)
    against the oracle."""
    o = OracleScene(sd)
    oracle_out = o.render(lp, records=True, threads=8)
    g = capi.Scene(sd)
    flags0 = lp.flags
    lp.flags = flags0 | capi.BF_FLAG_MEGAKERNEL
    _render_compare_one(g, lp, oracle_out, hist_rtol)
    lp.flags = flags0
    return _render_compare_one(g, lp, oracle_out, hist_rtol)


def _render_compare_one(g, lp, oracle_out, hist_rtol):
    hg, rg, sg = g.render(lp, records=True)
    ho, ro, so = oracle_out
    # integer counters and per-path results: exact
    assert np.array_equal(rg["n_rays"], ro["n_rays"])
    assert np.array_equal(rg["valid"], ro["valid"])
    assert np.array_equal(rg["aux"].view(np.uint32), ro["aux"].view(np.uint32))
    assert np.array_equal(rg["L"].view(np.uint32), ro["L"].view(np.uint32))
    assert sg.n_rays_closest == so.n_rays_closest and sg.n_rays_shadow == so.n_rays_shadow
    assert sg.n_bounces == so.n_bounces and sg.n_invalid == so.n_invalid
    # histogram: same addends, different summation order (fp32 atomics vs
    # double accumulation): relative tolerance hist_rtol of the channel value
    # plus N * 2^-24 * max addend absolute.
    n = float(lp.n_paths)
    amax = float(np.abs(ro["L"]).max()) if len(ro) else 0.0
    atol = n * 2.0 ** -24 * max(amax, 1.0) * 4
    assert np.allclose(hg, ho, rtol=hist_rtol, atol=atol), np.abs(hg - ho).max()
    rmse = float(np.sqrt(np.mean((hg / n - ho / n) ** 2)))
    # BASELINE.json target: per-range-bin RMSE < 1e-4 (relative to the largest bin when bins exceed 1)
    assert rmse < 1e-4 * max(1.0, float(np.abs(ho / n).max()))
    if lp.mode in (capi.BF_MODE_RECEIVE_RAW, capi.BF_MODE_RECEIVE_IQ):
        assert hg.reshape(-1, 3 + lp.phase_bins)[:, 2].sum() == n - sg.n_invalid
    else:
        assert hg[4] == n - sg.n_invalid      # weight channel counts the samples put
    return hg, ho, sg


def test_render_c1_trans_rad(hiplib):
    sd, lp = scenes.trans_rad(spp=16)          # BASELINE configs[0] literally
    _render_compare(sd, lp)
    sd, lp = scenes.trans_rad(spp=20000)
    hg, ho, st = _render_compare(sd, lp)
    assert (hg[5:] > 0).sum() > 100


@pytest.mark.parametrize("mode", [capi.BF_MODE_PATH, capi.BF_MODE_RANGE, capi.BF_MODE_TIME])
def test_render_modes_rect_scene(hiplib, mode):
    sd, lp = scenes.trans_rad(spp=5000)
    lp.mode = mode
    if mode == capi.BF_MODE_RANGE:
        lp.bins, lp.bin_width = 64, 0.25
    _render_compare(sd, lp)


@pytest.mark.parametrize("color", [capi.BF_COLOR_RGB, capi.BF_COLOR_MONO])
def test_render_bus_small(hiplib, color):
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=20000, bins=256, dr=0.1)
    lp.color_mode = color
    hg, ho, st = _render_compare(sd, lp)
    assert (hg[5:] != 0).sum() > 20


def test_render_c2_literal_64_paths(hiplib):
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=64)
    _render_compare(sd, lp)


def test_render_car_with_vertex_normals(hiplib):
    sd, lp = scenes.car_radar(n_tris=30000, n_paths=20000, bins=1024, dr=0.03)
    _render_compare(sd, lp)


@pytest.mark.parametrize("tx,rx,sig", [("wigner", "omnidirectional", "pulse"), ("area", "omnidirectional", "pulse"),
                                       ("wigner", "wigner", "pulse"), ("wigner", "omnidirectional", "linfmcw"),
                                       ("wigner", "omnidirectional", "cw")])
def test_receive_gen3(hiplib, tx, rx, sig):
    """Integrator::receive o PathTimeFrequencyIntegrator with Transmitter /
    Receiver / ADC plugins (C2-recv)."""
    sd, lp = scenes.bus_receive(n_tris=20000, n_paths=20000, transmitter=tx, receiver=rx, signaltype=sig)
    hg, ho, st = _render_compare(sd, lp)
    assert (hg.reshape(-1, 3)[:, 0] != 0).sum() > 5


def test_receive_2d_adc_uses_global_atomics(hiplib):
    """A 64 x 32 time-frequency ADC: frequency rows get populated and the
    histogram no longer fits the LDS budget check path (n_chan = 6144 does)."""
    sd, lp = scenes.bus_receive(n_tris=5000, n_paths=20000, t_bins=64)
    sd.sensor.f_bins = 32
    # spread the received band over the 32 rows: f in [c/lmax, c/lmin]
    c, lmin, lmax = sd.physics.c, sd.physics.lambda_min_nm, sd.physics.lambda_max_nm
    sd.sensor.f_bandwidth = c / (lmin * 1e-9)
    sd.finalize()
    lp.bins_y = 32
    hg, ho, st = _render_compare(sd, lp)
    rows = hg.reshape(32, 64, 3)[:, :, 2].sum(1)
    assert (rows > 0).sum() >= 5


def test_render_ragged_and_empty_launches(hiplib):
    sd, lp = scenes.trans_rad(spp=1)
    _render_compare(sd, lp)
    lp.n_paths = 0
    g = capi.Scene(sd)
    h, _, st = g.render(lp)
    assert np.all(h == 0)
    lp.n_paths = 333                      # not a multiple of the wave size
    _render_compare(sd, lp)


def test_render_sharding_by_path_offset(hiplib):
    """Multi-GPU contract: shards [g*N/G, (g+1)*N/G) by path_offset reproduce
    the 1-GPU sample set."""
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=8192)
    g = capi.Scene(sd)
    _, rec_all, st_all = g.render(lp, records=True)
    parts = []
    rays = 0
    for k in range(4):
        lpk = capi.make_launch(lp.mode, 2048, seed=lp.seed, path_offset=2048 * k, bins=lp.bins, bin_width=lp.bin_width,
                               color_mode=lp.color_mode)
        _, r, st = g.render(lpk, records=True)
        parts.append(r)
        rays += st.n_rays_closest + st.n_rays_shadow
    rec = np.concatenate(parts)
    assert np.array_equal(rec["L"].view(np.uint32), rec_all["L"].view(np.uint32))
    assert np.array_equal(rec["aux"].view(np.uint32), rec_all["aux"].view(np.uint32))
    assert rays == st_all.n_rays_closest + st_all.n_rays_shadow


def test_render_global_atomics_flag_matches_lds(hiplib):
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=20000)
    g = capi.Scene(sd)
    h1, _, _ = g.render(lp)
    lp.flags = capi.BF_FLAG_GLOBAL_ATOMICS | capi.BF_FLAG_STATS
    h2, _, st = g.render(lp)
    assert np.allclose(h1, h2, rtol=2e-5, atol=1e-3)
    assert st.n_nodes_visited > 0 and st.n_tris_tested > 0


def test_full_size_properties_c2(hiplib):
    """BASELINE-size run (200 k triangles, 2^20 paths): size-independent
    properties instead of an oracle diff."""
    sd, lp = scenes.bus_radar(n_tris=200_000, n_paths=1 << 20)
    g = capi.Scene(sd)
    h, _, st = g.render(lp)
    n = float(lp.n_paths)
    assert h[4] == n - st.n_invalid and st.n_invalid < 10
    assert 0 < h[3] <= n
    assert np.all(np.isfinite(h)) and np.all(h[5:] >= 0)
    assert st.n_rays_closest >= lp.n_paths
    # linearity: two half launches add up to the full one (same sample set)
    lpa = capi.make_launch(lp.mode, 1 << 19, seed=lp.seed, bins=lp.bins, bin_width=lp.bin_width)
    lpb = capi.make_launch(lp.mode, 1 << 19, seed=lp.seed, path_offset=1 << 19, bins=lp.bins, bin_width=lp.bin_width)
    ha, _, _ = g.render(lpa)
    hb, _, _ = g.render(lpb)
    assert np.allclose(ha + hb, h, rtol=1e-4, atol=1e-2)
    # idempotence: same launch twice gives the same counters
    h2, _, st2 = g.render(lp)
    assert st2.n_rays_closest == st.n_rays_closest and st2.n_rays_shadow == st.n_rays_shadow
    assert np.allclose(h2, h, rtol=1e-4, atol=1e-2)


@pytest.mark.timeout(600)
def test_bench_step_every_path_bit_exact(hiplib):
    """The bench workload itself — BASELINE configs[1] at the throughput size: 196 564 triangles, 2^24 paths, 45 M
    rays — through the default pipeline (16 Mi-slot pool, planned launches, tail kernel): every one of the 16.7 M
    per-path records (radiance bits, path length bits, validity, ray count) and every counter equal to the oracle's."""
    sd, lp = scenes.bus_radar(n_tris=200_000, n_paths=1 << 24, bins=256, dr=0.1, seed=1)
    ho, ro, so = OracleScene(sd).render(lp, records=True, threads=16)
    g = capi.Scene(sd)
    for rep in range(2):                          # the second render of a shape is the planned (host-sync-free) one
        hg, rg, sg = g.render(lp, records=True)
        assert np.array_equal(rg["n_rays"], ro["n_rays"]) and np.array_equal(rg["valid"], ro["valid"])
        assert np.array_equal(rg["aux"].view(np.uint32), ro["aux"].view(np.uint32))
        assert np.array_equal(rg["L"].view(np.uint32), ro["L"].view(np.uint32))
        assert (sg.n_rays_closest, sg.n_rays_shadow, sg.n_bounces, sg.n_invalid) == (so.n_rays_closest, so.n_rays_shadow, so.n_bounces, so.n_invalid)
        n = float(lp.n_paths)
        assert hg[4] == n - sg.n_invalid
        rmse = float(np.sqrt(np.mean((hg / n - ho / n) ** 2)))
        assert rmse < 1e-4                           # BASELINE.json's per-range-bin target
        assert np.allclose(hg, ho, rtol=2e-4, atol=n * 2.0 ** -24 * float(np.abs(ro["L"]).max()) * 4)
    assert sg.n_rays_closest + sg.n_rays_shadow > 44_000_000


@pytest.mark.timeout(600)
@pytest.mark.parametrize("iq", [False, True])
def test_full_size_c2_recv_every_path_bit_exact(hiplib, iq):
    """C2-recv / one C5 pulse at full size: the 200 k-triangle bus through gen-3 receive() (Wigner transmitter, 1024
    fast-time ADC bins, 2^22 paths), raw and coherent I/Q — every per-path record against the oracle."""
    sd, lp = scenes.bus_receive(n_tris=200_000, n_paths=1 << 22, t_bins=1024, dr=0.03, seed=4,
                                lambda_band_nm=(8.6e6 * 0.999, 8.6e6 * 1.001) if iq else None)
    if iq:
        lp.mode = capi.BF_MODE_RECEIVE_IQ
    ho, ro, so = OracleScene(sd).render(lp, records=True, threads=16)
    hg, rg, sg = capi.Scene(sd).render(lp, records=True)
    assert np.array_equal(rg["n_rays"], ro["n_rays"]) and np.array_equal(rg["valid"], ro["valid"])
    assert np.array_equal(rg["aux"].view(np.uint32), ro["aux"].view(np.uint32))
    assert np.array_equal(rg["L"].view(np.uint32), ro["L"].view(np.uint32))
    assert (sg.n_rays_closest, sg.n_rays_shadow, sg.n_bounces, sg.n_invalid) == (so.n_rays_closest, so.n_rays_shadow, so.n_bounces, so.n_invalid)
    a, b = hg.reshape(-1, 3), ho.reshape(-1, 3)
    assert np.array_equal(a[:, 2], b[:, 2])                                   # W: samples per ADC cell
    scale = max(float(np.abs(b[:, :2]).max()), 1e-30)
    assert np.abs(a[:, :2] - b[:, :2]).max() < 2e-4 * scale


def test_occluded_sample_with_non_finite_bsdf_value_poisons_the_path(hiplib):
    """Scene::sample_transmitter_direction zeroes the VALUE of an occluded sample (scene.cpp:220-224, 283-287) and the
    integrator still adds mis * throughput * bsdf_val * 0 (path.cpp:161, pathtimefrequency.cpp:232-240): path 1 949 725 of
    the C2-recv scene meets a NaN BSDF value on an occluded transmitter sample, so its radiance is NaN and
    ImageBlock / SignalBlock::put drops the sample.  (Found by the full-size test below; a contribution released only by
    an unoccluded shadow ray reports 0 here.)"""
    sd, lp = scenes.bus_receive(n_tris=200_000, n_paths=64, t_bins=1024, dr=0.03, seed=4)
    lp.path_offset = 1949725 - 17
    ho, ro, so = OracleScene(sd).render(lp, records=True)
    assert np.isnan(ro["L"][17]) and so.n_invalid == 1
    g = capi.Scene(sd)
    for flags in (0, capi.BF_FLAG_MEGAKERNEL):
        lp.flags = flags
        hg, rg, sg = g.render(lp, records=True)
        assert np.array_equal(rg["L"].view(np.uint32), ro["L"].view(np.uint32)) and sg.n_invalid == 1
        assert np.array_equal(hg.reshape(-1, 3)[:, 2], ho.reshape(-1, 3)[:, 2])


def test_irradiancemeter_sensor(hiplib):
    """irradiancemeter.cpp: area-sampled sensor with ray weight pi / surface_area (python_scripts/trans_image.xml)."""
    sd, lp = scenes.trans_rad(spp=20000)
    sd.set_irradiancemeter(sd.sensor.shape)
    sd.finalize()
    hg, ho, _ = _render_compare(sd, lp)
    sdf, _ = scenes.trans_rad(spp=20000)
    hf = capi.Scene(sdf).render(lp)[0]
    o = OracleScene(sd)
    area = o.lib.bfo_rect_area(o.handle, sd.sensor.shape)        # trans_rad holds rectangles only: shape index == rectangle index
    assert hg[1] > 0 and np.isclose(hg[1] / hf[1], 1.0 / area, rtol=1e-3)


def test_radiancemeter_sensor(hiplib):
    """radiancemeter.cpp: every path starts with the same ray (a pencil beam at the bus); no aperture sample is drawn."""
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=20000)
    sd.set_radiancemeter(Transform4f.look_at([0, 0, 0.3], [10.0, 3.0, 1.5], [0, 0, 1]))
    sd.finalize()
    hg, ho, sg = _render_compare(sd, lp)
    assert hg[3] == lp.n_paths                      # the beam hits the bus every time


def test_point_light(hiplib):
    """point.cpp: an isotropic point light next to the zoo's other emitters (uniform emitter selection over three kinds)."""
    sd, lp = _zoo_scene(two_emitters=True)
    sd.add_point([1.0, -2.0, 3.0], intensity=40.0)
    sd.finalize()
    hg, ho, _ = _render_compare(sd, lp)
    sd0, _ = _zoo_scene(two_emitters=True)
    assert not np.allclose(capi.Scene(sd0).render(lp)[0], hg)


def test_elementary_functions_bit_equal(hiplib):
    """The fp32 sin/cos/acos/exp/log/erf/tan specification evaluates to the same bits on the
    device as in the oracle (the oracle's accuracy against libm is checked on the CPU in
    test_oracle_known_answers.py::test_elementary_functions)."""
    from tests import oracle_lib
    olib = oracle_lib.load()
    rng = np.random.default_rng(11)
    n = 1 << 20
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, np.inf, -np.inf, np.nan, 1e-30, 88.8, -87.2, 4.0, 0.8, 1.6], f32)
    ranges = {0: (-4000, 4000), 1: (-4000, 4000), 2: (-1, 1), 3: (-90, 90), 4: (0, 1e6), 5: (-5, 5), 6: (-1.55, 1.55)}
    for op, (lo, hi) in ranges.items():
        x = np.concatenate([rng.uniform(lo, hi, n).astype(f32), rng.uniform(-1, 1, n).astype(f32), special])
        x = np.ascontiguousarray(x)
        yo = np.empty_like(x)
        yg = np.empty_like(x)
        olib.bfo_elementary(op, x.size, x.ctypes.data, yo.ctypes.data)
        capi.check(hiplib, hiplib.bf_eval_elementary(op, x.size, x.ctypes.data, yg.ctypes.data), "bf_eval_elementary")
        nan_o, nan_g = np.isnan(yo), np.isnan(yg)
        assert np.array_equal(nan_o, nan_g), op
        assert np.array_equal(yo[~nan_o].view(np.uint32), yg[~nan_g].view(np.uint32)), op


def test_full_size_c3_car_bit_exact(hiplib):
    """BASELINE configs[2] at full size: 1 M-triangle car body with vertex normals + ground, 1024 range
    bins, 2^20 primary rays — every path's radiance / length / ray count against the oracle."""
    sd, lp = scenes.car_radar(n_tris=1_000_000, n_paths=1 << 20, bins=1024, dr=0.03)
    hg, ho, st = _render_compare(sd, lp)
    assert (hg[5:] != 0).sum() > 50


def test_full_size_c4_multi_mesh_sharded(hiplib):
    """BASELINE configs[3] AT ITS CONFIGURED SIZE (bus 200 k + car 1 M + motorbike 300 k triangles, 4096 spp x 2^10 = 2^22
    paths in 8 shards of 2^19): one whole shard against the oracle path by path; the 8 shards' histograms sum to the
    unsharded 2^22-path render; and the 8 shards issued as ONE rolling sequence (what a GPU that renders several
    shards back to back does) give every shard the stand-alone shard's records."""
    torch = pytest.importorskip("torch")
    n = 4096 << 10
    sd, lp = scenes.multi_mesh_radar(n_paths=n)
    assert lp.n_paths == n and lp.bins == 4096
    shard = capi.make_launch(lp.mode, n // 8, seed=lp.seed, path_offset=3 * (n // 8), bins=lp.bins, bin_width=lp.bin_width,
                             color_mode=lp.color_mode)
    _render_compare(sd, shard)
    g = capi.Scene(sd)
    h_all, _, st_all = g.render(lp)
    acc = np.zeros_like(h_all, dtype=np.float64)
    rays = 0
    recs = []
    for k in range(8):
        lpk = capi.make_launch(lp.mode, n // 8, seed=lp.seed, path_offset=k * (n // 8), bins=lp.bins, bin_width=lp.bin_width,
                               color_mode=lp.color_mode)
        h, r, st = g.render(lpk, records=k in (0, 7))
        recs.append(r)
        acc += h
        rays += st.n_rays_closest + st.n_rays_shadow
    assert rays == st_all.n_rays_closest + st_all.n_rays_shadow
    assert acc[4] == n and h_all[4] == n
    assert np.allclose(acc, h_all, rtol=1e-4, atol=1e-2)
    # the shards as a rolling sequence on one handle
    hist = torch.zeros((8, g.channels(lp)), dtype=torch.float32, device="cuda")
    rec = torch.zeros((8, n // 8, 4), dtype=torch.int32, device="cuda")
    for k in range(8):
        lpk = capi.make_launch(lp.mode, n // 8, seed=lp.seed, path_offset=k * (n // 8), bins=lp.bins, bin_width=lp.bin_width,
                               color_mode=lp.color_mode, flags=capi.BF_FLAG_ROLLING)
        g.render_device(lpk, hist[k].data_ptr(), records_ptr=rec[k].data_ptr())
    g.flush()
    g.sync()
    hr = hist.cpu().numpy()
    assert np.allclose(hr.astype(np.float64).sum(axis=0), h_all, rtol=1e-4, atol=1e-2) and all(hk[4] == n // 8 for hk in hr)
    rr = rec.cpu().numpy().view(np.uint32)
    for k in (0, 7):
        got = np.ascontiguousarray(rr[k]).view(capi.PATH_RECORD_DTYPE).reshape(-1)
        for f in ("L", "aux"):
            assert np.array_equal(got[f].view(np.uint32), recs[k][f].view(np.uint32)), (k, f)
        assert np.array_equal(got["n_rays"], recs[k]["n_rays"])


def test_concurrent_renders_on_two_streams(hiplib):
    """bench.py issues consecutive steps on two HIP streams (one bf_scene handle each) so that one
    render's deep-path tail overlaps the next one's head: results must not depend on that."""
    torch = pytest.importorskip("torch")
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=1 << 18)
    handles = [capi.Scene(sd), capi.Scene(sd)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    nch = handles[0].channels(lp)
    ref, _, _ = handles[0].render(lp)
    hists = [torch.zeros(nch, device="cuda") for _ in range(6)]
    for k, h in enumerate(hists):
        j = k & 1
        with torch.cuda.stream(streams[j]):
            h.zero_()
            handles[j].render_device(lp, h.data_ptr(), stream=streams[j].cuda_stream)
    torch.cuda.synchronize()
    for h in hists:
        assert np.allclose(h.cpu().numpy(), ref, rtol=1e-4, atol=1e-2)


def test_planned_render_matches_first_render(hiplib):
    """The first wavefront render of a launch shape is driven bounce by bounce from the host and
    records a plan; later renders of that shape are enqueued without a host round trip.  Both must
    give every path the same result (and the oracle's)."""
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=1 << 16, bins=256, dr=0.1)
    o = OracleScene(sd)
    oracle_out = o.render(lp, records=True, threads=8)
    g = capi.Scene(sd)
    for _ in range(3):                       # 1st: synchronous, 2nd/3rd: planned (3rd after feedback)
        _render_compare_one(g, lp, oracle_out, 2e-5)
    lp2 = capi.make_launch(lp.mode, 1 << 16, seed=lp.seed + 17, bins=lp.bins, bin_width=lp.bin_width, color_mode=lp.color_mode)
    _render_compare_one(g, lp2, o.render(lp2, records=True, threads=8), 2e-5)     # new seed, same plan
    lp3 = capi.make_launch(lp.mode, 3000, seed=lp.seed, bins=lp.bins, bin_width=lp.bin_width, color_mode=lp.color_mode)
    _render_compare_one(g, lp3, o.render(lp3, records=True, threads=8), 2e-5)     # other shape: plan not applicable


def test_update_endpoints_equals_fresh_scene(hiplib):
    """bf_scene_update_endpoints (radar turned, BVH kept) gives every path the result a freshly
    created scene gives; a layout change or an endpoint outside the padded bound is refused."""
    mesh = scenes.bus_mesh(20000)
    sd0, lp = scenes.bus_radar(n_paths=20000, mesh=mesh)
    g = capi.Scene(sd0)
    for yaw in (12.5, -30.0):
        sd1, _ = scenes.bus_radar(n_paths=20000, mesh=mesh, radar_yaw_deg=yaw)
        g.update_endpoints(sd1)
        h_u, r_u, st_u = g.render(lp, records=True)
        h_f, r_f, st_f = capi.Scene(sd1).render(lp, records=True)
        assert np.array_equal(r_u["L"].view(np.uint32), r_f["L"].view(np.uint32))
        assert np.array_equal(r_u["aux"].view(np.uint32), r_f["aux"].view(np.uint32))
        assert np.array_equal(r_u["n_rays"], r_f["n_rays"])
        o = OracleScene(sd1)
        _render_compare_one(g, lp, o.render(lp, records=True, threads=8), 2e-5)
    h0, _, _ = capi.Scene(sd0).render(lp)
    assert not np.allclose(h0, h_u)                       # the radar really moved
    sd_other, _ = scenes.trans_rad(spp=16)
    with pytest.raises(capi.BeifongError):
        g.update_endpoints(sd_other)                      # different layout
    sd_far, _ = scenes.bus_radar(n_paths=20000, mesh=mesh, radar_position=(5000.0, 0.0, 0.3))
    with pytest.raises(capi.BeifongError, match="padded"):
        g.update_endpoints(sd_far)                        # same layout, but outside the bound the boxes allow for


def test_render_sweep_reuses_device_scenes(hiplib):
    """The frame loop of animated_trans_rad.py: radar yaw sweep over a static scene, frames rotated
    over HIP streams, BVH built once per stream handle."""
    pytest.importorskip("torch")
    from beifong_amd import sweep
    mesh = scenes.bus_mesh(20000)
    yaws = np.linspace(-20, 20, 9)
    frames = [scenes.bus_radar(n_paths=1 << 14, mesh=mesh, radar_yaw_deg=float(y)) for y in yaws]
    cube = sweep.render_sweep(frames, n_streams=3)
    assert cube.shape == (9, 5 + 256)
    assert sweep.render_sweep.last_stats == {"created": 3, "updated": 6}
    for k in (0, 4, 8):
        h, _, _ = capi.Scene(frames[k][0]).render(frames[k][1])
        assert np.allclose(cube[k], h, rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("tx,P", [("wigner", 16), ("area", 7), ("area", 1)])
def test_receive_phase_integrator(hiplib, tx, P):
    """receive o phase o pathtimefrequency: PhaseIntegrator's S{k}.Y channels after Y, A, W
    (phase.cpp:93-147; the last-segment phase of ray.h:89-93 / interaction.h:61-64)."""
    sd, lp = scenes.bus_receive(n_tris=20000, n_paths=40000, transmitter=tx)
    lp.phase_bins = P
    hg, ho, st = _render_compare(sd, lp)
    s = hg.reshape(256, 3 + P)[:, 3:]
    assert np.count_nonzero(s.sum(0)) == P if tx == "area" else np.count_nonzero(s) > 0
    # wide band: every finite phase wraps into the last bin (see tests/test_oracle_gen3.py)
    sd.physics.lambda_min_nm = 0.0
    sd.physics.lambda_max_nm = 2.0e12
    sd.finalize()
    hg, ho, st = _render_compare(sd, lp)
    s = hg.reshape(256, 3 + P)[:, 3:]
    assert np.all(s[:, :P - 1] == 0) and np.count_nonzero(s[:, P - 1]) > 0


def test_small_pool_regenerates_paths_into_freed_slots(hiplib, monkeypatch):
    """More paths than wavefront slots: slot i renders paths i, i + n_slots, ... (static assignment),
    in the host-driven first render, in planned renders and across the tail kernel."""
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=50000, bins=256, dr=0.1)
    o = OracleScene(sd)
    oracle_out = o.render(lp, records=True, threads=8)
    for pool, tail in (("4096", "512"), ("8192", "100000"), ("1024", "0")):
        monkeypatch.setenv("BF_WF_POOL", pool)
        monkeypatch.setenv("BF_WF_TAIL", tail)
        g = capi.Scene(sd)
        for _ in range(3):
            _render_compare_one(g, lp, oracle_out, 2e-5)


@pytest.mark.parametrize("tx", ["wigner", "area"])
def test_receive_iq_mode(hiplib, tx):
    """BF_MODE_RECEIVE_IQ: per-contribution phasors, I / Q / W per ADC cell — every path's (I, Q) bit-identical
    to the oracle, in the wavefront pipeline (deferred NEE contributions released by wf_trace), its planned
    renders, the tail kernel and the one-kernel variant."""
    sd, lp = scenes.bus_receive(n_tris=20000, n_paths=40000, transmitter=tx)
    lp.mode = capi.BF_MODE_RECEIVE_IQ
    hg, ho, st = _render_compare(sd, lp)
    cells = hg.reshape(256, 3)
    assert np.count_nonzero(cells[:, 0]) > 5 and np.count_nonzero(cells[:, 1]) > 5
    g = capi.Scene(sd)
    o = OracleScene(sd)
    out = o.render(lp, records=True, threads=8)
    for _ in range(2):
        _render_compare_one(g, lp, out, 2e-5)              # second render: planned


def test_translate_meshes_equals_rebuilt_scene(hiplib):
    """bf_scene_translate_meshes: vertices fl(p0 + offset), four-wide BVH re-fitted in place — every path as in
    a scene built from the shifted vertices; offsets are absolute (back to 0 restores the original)."""
    mesh = scenes.bus_mesh(20000)
    sd0, lp = scenes.bus_radar(n_paths=20000, mesh=mesh)
    g = capi.Scene(sd0)
    _, r0, _ = g.render(lp, records=True)
    for off in ([0.013, -0.2, 0.05], [-3.0, 1.5, 0.25]):
        g.translate_meshes(off)
        _, r_t, st_t = g.render(lp, records=True)
        v1 = (mesh[0] + np.asarray(off, np.float32)[None, :]).astype(np.float32)
        sd1, _ = scenes.bus_radar(n_paths=20000, mesh=(np.ascontiguousarray(v1), mesh[1], mesh[2]))
        _, r_f, st_f = capi.Scene(sd1).render(lp, records=True)
        for k in ("L", "aux"):
            assert np.array_equal(r_t[k].view(np.uint32), r_f[k].view(np.uint32))
        assert np.array_equal(r_t["n_rays"], r_f["n_rays"])
        assert not np.array_equal(r_t["L"], r0["L"])
        _render_compare_one(g, lp, OracleScene(sd1).render(lp, records=True, threads=8), 2e-5)
    g.translate_meshes([0, 0, 0])
    _, r_b, _ = g.render(lp, records=True)
    assert np.array_equal(r_b["L"].view(np.uint32), r0["L"].view(np.uint32))


def test_pulse_sweep_range_doppler_peak(hiplib):
    """BASELINE configs[4] in miniature: a plate approaching by dx per pulse rotates every path's phasor by
    2 pi * 2 dx / lambda per pulse; with common random numbers the slow-time FFT of (I + jQ) peaks at
    that Doppler bin, the static ground stays at zero Doppler."""
    pytest.importorskip("torch")
    from beifong_amd import sweep
    lam, n_pulses, dx = 0.1, 64, -0.004                 # approaching: optical length shrinks by 2 |dx| per pulse
    sd, lp = scenes.plate_doppler(wavelength_m=lam, n_paths=1 << 16, ground=True)
    offsets = np.zeros((n_pulses, 3), np.float32)
    offsets[:, 0] = dx * np.arange(n_pulses)
    cube = sweep.render_pulse_sweep(sd, lp, offsets, n_streams=3)
    assert cube.shape == (n_pulses, 1, 3) and np.all(cube[:, 0, 2] == lp.n_paths)
    rd = np.abs(sweep.range_doppler(cube)[:, 0])
    # phase of pulse k: -2 pi (L0 + 2 dx k) / lambda  ->  frequency +2 |dx| / lambda cycles per pulse
    expect = int(round(2 * abs(dx) / lam * n_pulses)) % n_pulses
    order = np.argsort(rd)[::-1]
    top = {int(order[0]), int(order[1]), int(order[2]), int(order[3])}
    assert expect in top or (expect + 1) % n_pulses in top
    assert 0 in top or 1 in top or n_pulses - 1 in top           # ground: zero Doppler
    # the moving-plate line stands well above the Doppler floor
    floor = np.median(rd)
    assert max(rd[expect], rd[(expect + 1) % n_pulses]) > 10 * floor
    # one pulse of the sweep against a stand-alone render of the same offset
    k = 17
    g = capi.Scene(sd)
    g.translate_meshes(offsets[k])
    h_k, _, _ = g.render(lp)
    assert np.allclose(h_k.reshape(1, 3), cube[k], rtol=1e-4, atol=1e-6 * np.abs(cube[k]).max())


def _planar_uv(v):
    """Texture coordinates for the zoo meshes: an oblique planar projection, with a patch collapsed to one point so
    that some triangles carry a degenerate parameterisation (mesh.cpp:500-502 keeps coordinate_system(n) there)."""
    v = np.asarray(v, np.float32)
    uv = np.stack([0.37 * v[:, 0] + 0.11 * v[:, 2], 0.29 * v[:, 1] - 0.2 * v[:, 2]], 1).astype(np.float32)
    uv[v[:, 2] > np.quantile(v[:, 2], 0.9)] = [0.25, 0.75]
    return uv


def _zoo_scene(two_emitters=True, receive=False, uv=False):
    """Small scene exercising the branches the radar configs do not: several emitters (uniform emitter
    selection, scene.cpp:180-230 / 249-299), one-sided materials, a mesh with and a mesh without normals."""
    sd = SceneDesc()
    T = Transform4f
    d0 = T.rotate([1, 0, 0], 90) * T.rotate([0, 1, 0], 90)
    ap = T.translate([0, 0, 0.3]) * d0 * T.scale([20e-3, 50e-3, 1])
    c, lmin, lmax = sd.physics.c, sd.physics.lambda_min_nm, sd.physics.lambda_max_nm
    if receive:
        txa = sd.add_rectangle(ap, sd.add_diffuse(0.0))
        rxa = sd.add_rectangle(ap, sd.add_diffuse(0.5))
        tau = 2.0 * 0.1 / c
        f_c = c / (0.5 * (lmin + lmax) * 1e-9)
        sd.add_wigner_transmitter(txa, signaltype="pulse", amplitude=1.0, freq_centre=f_c, freq_ext=1.0 / tau, pulse_len=tau,
                                  prf=1.0 / (64 * tau), gain=1.0)
        if two_emitters:
            tx2 = sd.add_rectangle(T.translate([0.0, 1.0, 0.6]) * d0 * T.scale([0.1, 0.1, 1]), sd.add_diffuse(0.0))
            sd.add_area_transmitter(tx2, 0.5)
        sd.set_receiver(rxa, kind="omnidirectional", adc_sampling_start=0.0, adc_sampling_end=64 * tau, t_bins=64, f_bins=1,
                        t_bandwidth=64 * tau, f_bandwidth=2.0 * c / (lmin * 1e-9), freq_centre=f_c,
                        freq_ext=c / (lmin * 1e-9) - c / (lmax * 1e-9))
        lp = capi.make_launch(capi.BF_MODE_RECEIVE_RAW, 30000, seed=9, bins=64, bins_y=1)
    else:
        txa = sd.add_rectangle(ap, sd.add_diffuse(0.0))
        sd.add_area_emitter(txa, 500.0)
        if two_emitters:
            sd.add_spot(T.look_at([0.5, -1.0, 2.0], [4.0, 0.0, 0.0], [0, 0, 1]), intensity=30.0, cutoff_angle=30.0, beam_width=20.0)
            tx3 = sd.add_rectangle(T.translate([2.0, 2.0, 2.5]) * T.rotate([1, 0, 0], 180) * T.scale([0.3, 0.3, 1]), sd.add_diffuse(0.0))
            sd.add_area_emitter(tx3, 20.0)
        sd.set_perspective(T.translate([0, 0, 0.3]) * d0, fov=60.0, near_clip=0.1, far_clip=100.0)
        lp = capi.make_launch(capi.BF_MODE_RANGE, 30000, seed=9, bins=128, bin_width=0.1, color_mode=capi.BF_COLOR_RGB)
    sd.add_rectangle(T.scale([20, 20, 1]), sd.add_diffuse(0.4, twosided=False))
    v, f, n = meshgen.car_body(6000, seed=3)
    vc = meshgen.place(v, 25.0, (4.0, 0.5, 0.8))
    # uv=True: anisotropic roughness, so that the shading frame's s (from dp_du, interaction.h:159-162) shapes the lobe
    sd.add_mesh(vc, f, sd.add_roughconductor(alpha=0.3, alpha_v=0.05 if uv else None, twosided=False, specular_reflectance=0.7),
                normals=meshgen.vertex_normals(vc, f), texcoords=_planar_uv(vc) if uv else None)
    v, f = meshgen.bus(4000, seed=6)
    vb = meshgen.place(v, -40.0, (7.0, -3.0, 1.7), scale=0.5)
    sd.add_mesh(vb, f, sd.add_diffuse(0.9, twosided=True), texcoords=_planar_uv(vb) if uv else None)
    sd.finalize()
    return sd, lp


@pytest.mark.parametrize("receive", [False, True])
@pytest.mark.parametrize("two", [False, True])
def test_zoo_multi_emitter_one_sided(hiplib, receive, two):
    sd, lp = _zoo_scene(two_emitters=two, receive=receive)
    for max_depth, rr_depth in ((-1, 5), (3, 5), (-1, 1), (1, 5)):
        lp.max_depth, lp.rr_depth = max_depth, rr_depth
        hg, ho, st = _render_compare(sd, lp)
        if max_depth == -1:
            assert np.count_nonzero(hg) > 3


def test_ray_intersect_full_surface_interaction(hiplib):
    """bf_ray_intersect: every field of the SurfaceInteraction (mesh.cpp:452-548, rectangle.cpp:265-298,
    interaction.h:159-162) bit-equal to the oracle, with and without vertex normals / texture coordinates."""
    for uv in (False, True):
        sd, _ = _zoo_scene(uv=uv)
        g, o = capi.Scene(sd), OracleScene(sd)
        rays = _rays(3000, 21 + uv, extent=1.0)
        rays[:, 0:3] += [4.0, 0.0, 1.5]
        r = g.ray_intersect(rays)
        t, prim, shape, _ = o.trace_closest(rays)
        assert np.array_equal(r["t"].view(np.uint32), t.view(np.uint32))
        hit = np.isfinite(t)
        assert np.array_equal(r["prim"][hit], prim[hit]) and np.array_equal(r["shape"][hit], shape[hit])
        assert len(set(shape[hit])) >= 3 and hit.sum() > 1500
        assert not r["raw"][~hit, 1:].any()
        n_uv_tangent = 0
        for i in np.flatnonzero(hit):
            ref = o.intersect_full(rays[i])
            for k in ("t", "p", "n", "sh_n", "sh_s", "sh_t", "wi", "prim_uv", "dp_du", "dp_dv"):
                a, b = np.atleast_1d(r[k][i]), np.atleast_1d(ref[k])
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (uv, i, k, a, b)
            # dp_du x dp_dv is parallel to n unless coordinate_system(n) was kept: count the UV-parameterised ones
            n_uv_tangent += shape[i] >= len(sd.shapes) - 2 and abs(np.linalg.norm(ref["dp_du"]) - 1.0) > 1e-3
        assert (n_uv_tangent > 500) if uv else n_uv_tangent == 0


def test_render_uv_tangents_shape_anisotropic_lobe(hiplib):
    sd, lp = _zoo_scene(uv=True)
    hist, _, _ = _render_compare(sd, lp)
    # the same scene without texture coordinates renders differently: the tangent frame is observable
    sd0, _ = _zoo_scene(uv=True)
    for s in sd0.shapes:
        s.texcoords = None
    sd0.finalize()
    h0, _, _ = capi.Scene(sd0).render(lp)
    assert not np.array_equal(h0, hist)


@pytest.mark.parametrize("phased_tx,phased_rx,steer", [(True, True, (0.0, 0.0, 0.0)), (True, False, (12.0, 0.0, 0.0)),
                                                        (False, True, (0.0, 0.0, 0.0)), (True, True, (20.0, 5.0, 0.0))])
def test_receive_phased_array_endpoints(hiplib, phased_tx, phased_rx, steer):
    """src/transmitters/phasedtransmitter.cpp + src/receivers/phasedreceiver.cpp: the array's Wigner function (sum over
    n_elems^2 virtual elements with complex steering phasors) in eval / sample_direction / pdf_direction / sample_ray."""
    sd, lp = scenes.phased_receive(n_tris=20000, n_paths=30000, n_elems=4, steer_deg=steer, phased_tx=phased_tx,
                                   phased_rx=phased_rx)
    hg, ho, st = _render_compare(sd, lp)
    assert np.count_nonzero(hg.reshape(64, 3)[:, 0]) > 5
    lp.mode = capi.BF_MODE_RECEIVE_IQ
    _render_compare(sd, lp)


def test_phased_array_steering_update(hiplib):
    """Beam steering between frames: bf_scene_update_endpoints replaces the element tables in place."""
    sd0, lp = scenes.phased_receive(n_tris=20000, n_paths=20000, steer_deg=(0.0, 0.0, 0.0))
    sd1, _ = scenes.phased_receive(n_tris=20000, n_paths=20000, steer_deg=(15.0, 0.0, 0.0))
    g = capi.Scene(sd0)
    h0, _, _ = g.render(lp)
    g.update_endpoints(sd1)
    _render_compare_one(g, lp, OracleScene(sd1).render(lp, records=True, threads=8), 2e-5)
    h1, _, _ = g.render(lp)
    assert not np.allclose(h0, h1)
    sd2, _ = scenes.phased_receive(n_tris=20000, n_paths=20000, n_elems=3)
    with pytest.raises(capi.BeifongError):
        g.update_endpoints(sd2)                      # another array size: not an in-place update


def test_cross_seed_statistical_agreement(hiplib):
    """two independent seeds agree per range bin within Monte-Carlo error (3-sigma band from the
    per-path records, a few bins may stray), and the integer bookkeeping is seed-independent in expectation."""
    n = 1 << 18
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=n, bins=256, dr=0.1, seed=11)
    g = capi.Scene(sd)
    ha, ra, sa = g.render(lp, records=True)
    lp2 = capi.make_launch(lp.mode, n, seed=987654321, bins=lp.bins, bin_width=lp.bin_width, color_mode=lp.color_mode)
    hb, rb, sb = g.render(lp2, records=True)
    assert not np.array_equal(ra["L"], rb["L"])
    ba, bb = ha[5:] / n, hb[5:] / n
    # per-bin variance of the estimator from seed A's records (the AOV holds the radiance without the sensor weight)
    idx = np.floor(ra["aux"] / np.float32(lp.bin_width)).astype(np.int64)
    ok = (idx >= 0) & (idx < lp.bins) & (ra["L"] != 0)
    s1 = np.bincount(idx[ok], ra["L"][ok].astype(np.float64), minlength=lp.bins)
    s2 = np.bincount(idx[ok], ra["L"][ok].astype(np.float64) ** 2, minlength=lp.bins)
    scale = np.where(s1 != 0, ba.astype(np.float64) * n / np.where(s1 != 0, s1, 1), 1.0)       # records carry L incl. weight
    var = (s2 * scale ** 2 / n - (s1 * scale / n) ** 2) / n
    sigma = np.sqrt(2 * np.maximum(var, 0))
    lit = (ba > 0) | (bb > 0)
    z = np.abs(ba - bb)[lit] / np.maximum(sigma[lit], 1e-12)
    assert np.mean(z < 3.0) > 0.9 and np.median(z) < 1.5
    assert abs(sa.n_rays_closest / n - sb.n_rays_closest / n) < 0.02
    assert abs(float(ha[3]) - float(hb[3])) < 5 * np.sqrt(n)            # alpha: hit fraction


def _film_compare(sd, lp, film, chan):
    """Multi-pixel film: per-path records exact, per-pixel histograms to summation order, both pipelines."""
    ho, ro, so = OracleScene(sd).render(lp, records=True, threads=8)
    g = capi.Scene(sd)
    out = None
    for flags in (capi.BF_FLAG_MEGAKERNEL, 0, capi.BF_FLAG_GLOBAL_ATOMICS):
        lp.flags = flags
        hg, rg, sg = g.render(lp, records=True)
        for k in ("n_rays", "valid"):
            assert np.array_equal(rg[k], ro[k])
        assert np.array_equal(rg["aux"].view(np.uint32), ro["aux"].view(np.uint32))
        assert np.array_equal(rg["L"].view(np.uint32), ro["L"].view(np.uint32))
        assert sg.n_rays_closest == so.n_rays_closest and sg.n_rays_shadow == so.n_rays_shadow and sg.n_invalid == so.n_invalid
        a, b = hg.reshape(film[1], film[0], chan), ho.reshape(film[1], film[0], chan)
        assert np.array_equal(a[:, :, 3:5], b[:, :, 3:5])                    # alpha and weight: integer counts
        amax = float(np.abs(ro["L"]).max())
        assert np.allclose(a, b, rtol=2e-5, atol=lp.spp * 2.0 ** -24 * max(amax, 1.0) * 4)
        out = a
    lp.flags = 0
    return out


def test_multi_pixel_film_known_answer_on_device(hiplib):
    film, spp = (4, 2), 256
    sd, lp = scenes.film_half_lit(film, spp, 3.0)
    img = _film_compare(sd, lp, film, 5)
    assert np.array_equal(img[:, :, 4], np.full((2, 4), spp)) and np.array_equal(img[:, :, 3], [[spp, spp, 0, 0]] * 2)
    assert np.allclose(img[:, :2, :3], 3.0 * spp, rtol=1e-6)
    # spp = 0 on a 1 x 1 film is the same launch
    sd1, lp1 = scenes.film_half_lit((1, 1), 4096, 3.0)
    g = capi.Scene(sd1)
    a = g.render(lp1)[0]
    lp1.film_width = lp1.film_height = lp1.spp = 0
    assert np.allclose(a, g.render(lp1)[0], rtol=1e-6)
    # the launch has to name the sensor's film
    lp.film_width = 3
    with pytest.raises(capi.BeifongError, match="film"):
        capi.Scene(sd).render(lp)


@pytest.mark.parametrize("mode,film", [(capi.BF_MODE_RANGE, (8, 6)), (capi.BF_MODE_TIME, (5, 3)), (capi.BF_MODE_RANGE, (96, 64))])
def test_multi_pixel_range_image_of_the_zoo(hiplib, mode, film):
    """A transient (range / time resolved) IMAGE: the zoo scene through a perspective camera with a W x H film.
    96 x 64 x (5 + 128) floats exceed the LDS-privatised histogram: global atomics."""
    sd, lp0 = _zoo_scene(two_emitters=True)
    T = Transform4f
    sd.set_perspective(T.translate([0, 0, 0.3]) * T.rotate([1, 0, 0], 90) * T.rotate([0, 1, 0], 90), fov=60.0, near_clip=0.1,
                       far_clip=100.0, film=film)
    sd.finalize()
    spp = 64 if film[0] < 32 else 4
    bins = 128 if mode == capi.BF_MODE_RANGE else 40
    lp = capi.make_launch(mode, film[0] * film[1] * spp, seed=5, bins=bins, bin_width=0.1 if mode == capi.BF_MODE_RANGE else 1e-9,
                          color_mode=capi.BF_COLOR_RGB, film=film, spp=spp)
    chan = 5 + bins * (1 if mode == capi.BF_MODE_RANGE else 3)
    img = _film_compare(sd, lp, film, chan)
    assert np.array_equal(img[:, :, 4], np.full((film[1], film[0]), spp))
    assert (img[:, :, 5:].sum(axis=2) > 0).mean() > 0.3                     # the scene is in view
    # sharding by path_offset tiles the film's sample set exactly
    if film == (8, 6):
        g = capi.Scene(sd)
        full = g.render(lp, records=True)[1]
        half = lp.n_paths // 2 + 7
        parts = []
        for off, cnt in ((0, half), (half, lp.n_paths - half)):
            l2 = capi.make_launch(mode, cnt, seed=5, path_offset=off, bins=bins, bin_width=lp.bin_width, color_mode=capi.BF_COLOR_RGB,
                                  film=film, spp=spp)
            parts.append(g.render(l2, records=True))
        assert np.array_equal(np.concatenate([p[1] for p in parts]), full)
        assert np.allclose(parts[0][0] + parts[1][0], img.ravel(), rtol=2e-5, atol=1e-4)


def _fuzz_receive_endpoints(sd, rng):
    """gen-3 endpoints for _fuzz_scene: one or two transmitters (wigner pulse / linfmcw, area), an omnidirectional or
    Wigner receiver with a random ADC, RECEIVE_RAW (with or without phase AOVs) or RECEIVE_IQ."""
    T = Transform4f
    c, lmin, lmax = sd.physics.c, sd.physics.lambda_min_nm, sd.physics.lambda_max_nm
    tau = float(rng.uniform(0.5, 3.0)) * 0.1 / c
    f_c = c / (0.5 * (lmin + lmax) * 1e-9)
    t_bins, f_bins = int(rng.integers(1, 96)), int(rng.choice([1, 1, 2, 5]))
    pose = T.translate([0.2, -0.3, 1.2]) * T.rotate([1, 0, 0], float(rng.uniform(100, 260))) * T.scale([0.05, 0.08, 1])
    for k in range(int(rng.integers(1, 3))):
        tx = sd.add_rectangle(pose if k == 0 else T.translate([float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)), 3.0]) *
                              T.rotate([1, 0, 0], 180) * T.scale([0.1, 0.2, 1]), sd.add_diffuse(0.0))
        if rng.random() < 0.6:
            sd.add_wigner_transmitter(tx, signaltype="linfmcw" if rng.random() < 0.4 else "pulse", amplitude=float(rng.uniform(0.5, 2)),
                                      freq_centre=f_c, freq_ext=1.0 / tau, pulse_len=tau, prf=1.0 / (t_bins * tau), gain=float(rng.uniform(0.5, 2)))
        else:
            sd.add_area_transmitter(tx, float(rng.uniform(0.5, 5)))
    rx = sd.add_rectangle(pose, sd.add_diffuse(0.5))
    sd.set_receiver(rx, kind="wigner" if rng.random() < 0.4 else "omnidirectional", adc_sampling_start=float(rng.choice([0.0, 2 * tau])),
                    adc_sampling_end=t_bins * tau, t_bins=t_bins, f_bins=f_bins, t_bandwidth=t_bins * tau,
                    f_bandwidth=2.0 * c / (lmin * 1e-9), freq_centre=f_c, freq_ext=c / (lmin * 1e-9) - c / (lmax * 1e-9),
                    gain=float(rng.uniform(0.5, 2)), sig_is_delta=bool(rng.integers(2)))
    sd.finalize()
    iq = rng.random() < 0.3
    lp = capi.make_launch(capi.BF_MODE_RECEIVE_IQ if iq else capi.BF_MODE_RECEIVE_RAW, 6000, seed=int(rng.integers(1 << 30)), bins=t_bins,
                          bins_y=f_bins, max_depth=int(rng.choice([-1, 2, 3, 8])), rr_depth=int(rng.choice([1, 3, 5, 50])),
                          phase_bins=0 if iq or rng.random() < 0.5 else int(rng.integers(1, 20)))
    return sd, lp


def _fuzz_scene(seed, receive=False):
    """A random small scene of the render modes: a box of 3-6 rectangles and 1-3 meshes with random materials
    (diffuse / rough conductor, Beckmann / GGX, one- and two-sided, isotropic or not, visible-normal sampling or
    not), a spot or area emitter (or both), fluxmeter or perspective sensor (possibly with a small film), random
    mode, colour mode, depth limits and bin widths."""
    rng = np.random.default_rng(1000 + seed + (500 if receive else 0))
    sd = SceneDesc()
    T = Transform4f

    def material():
        two = bool(rng.integers(2))
        if rng.random() < 0.45:
            return sd.add_diffuse(float(rng.uniform(0.05, 0.95)), twosided=two)
        au = float(rng.choice([0.05, 0.15, 0.4, 0.8]))
        return sd.add_roughconductor(alpha=au, alpha_v=float(rng.choice([0.05, 0.3])) if rng.random() < 0.3 else None, twosided=two,
                                     distribution="ggx" if rng.random() < 0.5 else "beckmann", sample_visible=bool(rng.integers(2)),
                                     specular_reflectance=float(rng.uniform(0.3, 1.0)) if rng.random() < 0.6 else None)

    # floor + a few random walls
    sd.add_rectangle(T.scale([6, 6, 1]), material())
    for _ in range(int(rng.integers(2, 6))):
        ax = rng.standard_normal(3)
        ax /= np.linalg.norm(ax)
        sd.add_rectangle(T.translate(list(rng.uniform(-3, 3, 2)) + [float(rng.uniform(0.3, 3))]) * T.rotate(list(ax), float(rng.uniform(0, 360))) *
                         T.scale([float(rng.uniform(0.3, 2.5)), float(rng.uniform(0.3, 2.5)), 1]), material())
    for k in range(int(rng.integers(1, 4))):
        kind = int(rng.integers(3))
        if kind == 0:
            v, f = meshgen.triangle_soup(int(rng.integers(50, 3000)), seed=seed * 7 + k, extent=1.0, size=float(rng.uniform(0.05, 0.5)))
            n = None
        elif kind == 1:
            v, f, n = meshgen.car_body(int(rng.integers(500, 4000)), seed=seed * 7 + k)
            v = v * 0.4
        else:
            v, f = meshgen.bus(int(rng.integers(500, 4000)), seed=seed * 7 + k)
            v = v * 0.25
            n = None
        v = meshgen.place(v, float(rng.uniform(0, 360)), tuple(rng.uniform(-2, 2, 2)) + (float(rng.uniform(0.5, 2.0)),))
        if kind == 1 or rng.random() < 0.4:
            n = meshgen.vertex_normals(v, f)
        sd.add_mesh(v, f, material(), normals=n, texcoords=_planar_uv(v) if rng.random() < 0.4 else None)
    if receive:
        return _fuzz_receive_endpoints(sd, rng)
    em = int(rng.integers(4))
    if em == 3:
        sd.add_point(list(rng.uniform(-3, 3, 2)) + [float(rng.uniform(1.5, 4))], intensity=float(rng.uniform(5, 50)))
        if rng.random() < 0.5:
            em = 1                                     # ... plus an area light
    if em in (0, 2):
        sd.add_spot(T.look_at(list(rng.uniform(-3, 3, 2)) + [float(rng.uniform(2, 4))], list(rng.uniform(-1, 1, 3)), [0, 0, 1]),
                    intensity=float(rng.uniform(5, 50)), cutoff_angle=float(rng.uniform(15, 60)), beam_width=float(rng.uniform(5, 14)))
    if em in (1, 2):
        r = sd.add_rectangle(T.translate([float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)), 3.5]) * T.rotate([1, 0, 0], 180) *
                             T.scale([float(rng.uniform(0.05, 1.0)), float(rng.uniform(0.05, 1.0)), 1]), sd.add_diffuse(0.0))
        sd.add_area_emitter(r, float(rng.uniform(1, 40)))
    film, spp = None, 0
    kind = rng.random()
    if kind < 0.4:
        rx = sd.add_rectangle(T.translate([0.2, -0.3, 1.0]) * T.rotate([1, 0, 0], float(rng.uniform(90, 270))) * T.scale([0.05, 0.08, 1]),
                              sd.add_diffuse(0.5))
        if rng.random() < 0.5:
            sd.set_fluxmeter(rx)
        else:
            sd.set_irradiancemeter(rx)
    elif kind < 0.5:
        sd.set_radiancemeter(T.look_at(list(rng.uniform(-3, 3, 2)) + [float(rng.uniform(0.5, 3))], list(rng.uniform(-1, 1, 2)) + [0.5], [0, 0, 1]))
    else:
        if rng.random() < 0.5:
            film = (int(rng.integers(1, 7)), int(rng.integers(1, 5)))
        sd.set_perspective(T.look_at(list(rng.uniform(-3, 3, 2)) + [float(rng.uniform(0.5, 3))], [0, 0, 0.8], [0, 0, 1]),
                           fov=float(rng.uniform(30, 100)), near_clip=0.05, far_clip=100.0, film=film or (1, 1))
    sd.finalize()
    mode = int(rng.choice([capi.BF_MODE_PATH, capi.BF_MODE_RANGE, capi.BF_MODE_TIME]))
    n_paths = 6000
    if film:
        spp = n_paths // (film[0] * film[1])
        n_paths = spp * film[0] * film[1]
    lp = capi.make_launch(mode, n_paths, seed=int(rng.integers(1 << 30)), bins=int(rng.integers(1, 200)),
                          bin_width=float(rng.uniform(0.02, 0.5)) if mode == capi.BF_MODE_RANGE else float(rng.uniform(1e-10, 2e-9)),
                          color_mode=int(rng.integers(2)), max_depth=int(rng.choice([-1, 1, 2, 3, 8])), rr_depth=int(rng.choice([1, 3, 5, 50])),
                          film=film, spp=spp)
    return sd, lp


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_random_scenes_render(hiplib, seed):
    sd, lp = _fuzz_scene(seed)
    ho, ro, so = OracleScene(sd).render(lp, records=True, threads=8)
    g = capi.Scene(sd)
    for flags in (capi.BF_FLAG_MEGAKERNEL, 0):
        lp.flags = flags
        hg, rg, sg = g.render(lp, records=True)
        assert np.array_equal(rg["n_rays"], ro["n_rays"]) and np.array_equal(rg["valid"], ro["valid"]), seed
        assert np.array_equal(rg["aux"].view(np.uint32), ro["aux"].view(np.uint32)), seed
        assert np.array_equal(rg["L"].view(np.uint32), ro["L"].view(np.uint32)), seed
        assert (sg.n_rays_closest, sg.n_rays_shadow, sg.n_bounces, sg.n_invalid) == (so.n_rays_closest, so.n_rays_shadow, so.n_bounces, so.n_invalid)
        amax = float(np.nanmax(np.abs(ro["L"])))           # seed 2 has a NaN sample: dropped by both, bit-equal in the records
        assert np.allclose(hg, ho, rtol=2e-5, atol=lp.n_paths * 2.0 ** -24 * max(amax, 1.0) * 4), seed


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("hook", [False, True])
def test_fuzz_random_scenes_receive(hiplib, seed, hook):
    """hook: the same scenes with the Doppler hook on (every shape's default velocity) and, on the omnidirectional
    receiver, receive_type "mix_resample" (odd seeds): records AND the ADC rows against the oracle."""
    sd, lp = _fuzz_scene(seed, receive=True)
    extra = 0
    if hook:
        extra = capi.BF_FLAG_DOPPLER
        if sd.sensor.type == capi.BF_RECEIVER_OMNI and seed % 2:
            extra |= capi.BF_FLAG_MIX_RESAMPLE
    lp.flags = extra
    ho, ro, so = OracleScene(sd).render(lp, records=True, threads=8)
    g = capi.Scene(sd)
    for flags in (capi.BF_FLAG_MEGAKERNEL | extra, extra):
        lp.flags = flags
        hg, rg, sg = g.render(lp, records=True)
        assert np.array_equal(rg["n_rays"], ro["n_rays"]) and np.array_equal(rg["valid"], ro["valid"]), seed
        assert np.array_equal(rg["aux"].view(np.uint32), ro["aux"].view(np.uint32)), seed
        assert np.array_equal(rg["L"].view(np.uint32), ro["L"].view(np.uint32)), seed
        assert (sg.n_rays_closest, sg.n_rays_shadow, sg.n_bounces, sg.n_invalid) == (so.n_rays_closest, so.n_rays_shadow, so.n_bounces, so.n_invalid)
        amax = float(np.nanmax(np.abs(ro["L"])))
        assert np.allclose(hg, ho, rtol=2e-5, atol=lp.n_paths * 2.0 ** -24 * max(amax, 1.0) * 4), seed


@pytest.mark.timeout(300)
def test_trace_scheduling_extremes_complete_without_guard(hiplib, monkeypatch):
    """wf_trace's straggler rule postpones the node steps of a few lanes while others hold leaves; its progress condition
    ("only if some lane HAS a leaf to intersect") is what the one GPU hang of round 1 lacked.  BF_TRACE_STRAGGLERS=64 /
    BF_TRACE_REFILL=0 is the configuration that rule protects: every node step is postponable and no lane is refilled
    before the whole wave is idle.  The render must complete, the iteration guard must not have dropped a ray
    (bf_stats.n_guard == 0; a non-zero count makes bf_render fail with BF_ERR_DEVICE), every path as the oracle's."""
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=1 << 16, bins=256, dr=0.1)
    out = OracleScene(sd).render(lp, records=True, threads=8)
    for strag, refill in (("64", "0"), ("1", "63"), ("12", "44")):
        monkeypatch.setenv("BF_TRACE_STRAGGLERS", strag)
        monkeypatch.setenv("BF_TRACE_REFILL", refill)
        g = capi.Scene(sd)
        for _ in range(2):                                   # synchronous, then planned
            _, _, st = _render_compare_one(g, lp, out, 2e-5)
            assert st.n_guard == 0 and st.n_rays_traced > 0


@pytest.mark.timeout(900)
def test_c5_full_size_sweep(hiplib):
    """BASELINE configs[4] at its configured size on one GPU: 200 k-triangle bus, 64 pulses x 2^20 paths, 1024 fast-time
    bins, I/Q ADC, through PulseSweeper (batched launches; the loop it replaces: python_scripts/animated_trans_rad.py:307-384).
    Pulses 0, 31 and 63 of the cube equal stand-alone renders of the translated scene; every path of those pulses equals
    the oracle's on a scene BUILT from the shifted vertices; the Doppler line of a 0.5 m/s target lands in the predicted bin."""
    pytest.importorskip("torch")
    from beifong_amd import sweep
    n_pulses, pri, lam0 = 64, 1e-3, 8.6e6
    sd, lp = scenes.bus_receive(n_tris=200_000, n_paths=1 << 20, t_bins=1024, dr=0.03, seed=4,
                                lambda_band_nm=(lam0 * 0.999, lam0 * 1.001))
    lp.mode = capi.BF_MODE_RECEIVE_IQ
    speed = 0.5
    offsets = np.zeros((n_pulses, 3), np.float32)
    offsets[:, 0] = (-speed * pri * np.arange(n_pulses)).astype(np.float32)
    sw = sweep.PulseSweeper(sd, lp, n_streams=2)
    cube = sw.render(offsets)
    sw.close()
    assert cube.shape == (n_pulses, 1024, 3)
    assert np.all(cube[:, :, 2].sum(1) == lp.n_paths)                      # W channel: every path of every pulse binned
    # the Doppler line: 2 v PRI / lambda cycles per pulse (the analysis of tools/c5_sweep.py)
    rd = np.abs(sweep.range_doppler(cube))
    far = rd[:, 200:]
    prof = far[2:-1].sum(1)
    k = int(np.argmax(prof)) + 2
    # A scatterer in direction u from the radar, moving with velocity w, advances its round-trip phase by 2 (w . u) PRI /
    # lambda cycles per pulse.  The bus is an extended target seen under 17 .. 50 degrees from the radar's axis, so its line
    # lies between the Doppler bins of its extreme directions (the band-centre figure 2 v PRI / lambda * n_pulses = 7.4 is
    # the head-on bound), and stands far above the Doppler floor.
    lam = lam0 * 1e-9
    v, f = meshgen.bus(200_000, seed=1)
    v = meshgen.place(v, yaw_deg=-20.0, translate=(10.0, 3.0, 1.7)).astype(np.float32)
    to_bus = v.astype(np.float64) - np.array([0.0, 0.0, 0.3])
    ux = to_bus[:, 0] / np.linalg.norm(to_bus, axis=1)
    head_on = 2 * speed * pri / lam * n_pulses
    assert head_on * ux.min() - 1.0 <= k <= head_on * ux.max() + 1.0, (k, head_on * ux.min(), head_on * ux.max())
    assert prof.max() > 20 * np.median(prof)
    # three pulses: stand-alone renders of the translated scene, and the oracle per path
    pick = [0, 31, 63]
    g = capi.Scene(sd)
    hb, rb, _ = g.render_batch(lp, len(pick), offsets=offsets[pick], records=True)
    g2 = capi.Scene(sd)
    for j, kk in enumerate(pick):
        scale = float(np.abs(hb[j]).max())
        assert np.allclose(hb[j].reshape(1024, 3), cube[kk], rtol=1e-4, atol=1e-5 * scale)
        g2.translate_meshes(offsets[kk])
        hs, rs, _ = g2.render(lp, records=True)
        for key in ("L", "aux"):
            assert np.array_equal(rb[j][key].view(np.uint32), rs[key].view(np.uint32))
        assert np.array_equal(rb[j]["n_rays"], rs["n_rays"])
        assert np.allclose(hb[j], hs, rtol=1e-4, atol=1e-5 * scale)
        v1 = np.ascontiguousarray((v + offsets[kk][None, :]).astype(np.float32))
        from tests.test_gpu_batch import _bus_receive_with_mesh
        sd1 = _bus_receive_with_mesh(v1, f, t_bins=1024, dr=0.03, lambda_band_nm=(lam0 * 0.999, lam0 * 1.001))
        _, ro, _ = OracleScene(sd1).render(lp, records=True, threads=8)
        for key in ("L", "aux"):
            assert np.array_equal(rb[j][key].view(np.uint32), ro[key].view(np.uint32))
        assert np.array_equal(rb[j]["n_rays"], ro["n_rays"]) and np.array_equal(rb[j]["valid"], ro["valid"])


def _doppler_scene(scale):
    """C2-recv geometry with a 64 x 32 time-frequency ADC whose rows span the received band, and a `velocity` transform
    (scale * identity, so Shape::doppler = 2 scale dot(wi, to_local(p)) / c * lambda) on the bus."""
    sd, lp = scenes.bus_receive(n_tris=5000, n_paths=30000, t_bins=64)
    sd.sensor.f_bins = 32
    c, lmin = sd.physics.c, sd.physics.lambda_min_nm
    sd.sensor.f_bandwidth = c / (lmin * 1e-9)
    vel = np.eye(4, dtype=np.float32) * np.float32(scale)
    vel[3, 3] = 1.0
    for s in sd.shapes:
        if s.type == capi.BF_SHAPE_MESH:
            s.velocity = (capi.M16)(*vel.reshape(-1).tolist())
    sd.finalize()
    lp.bins_y = 32
    return sd, lp


def test_doppler_hook_off_by_default_and_bit_exact_when_on(hiplib):
    """Shape::doppler (shape.cpp:375-389), which the reference carries with its call sites commented out
    (pathtimefrequency.cpp:124-126, 141-144, 180-183).  BF_FLAG_DOPPLER enables the hook in oracle and kernels: the
    shifted wavelength selects the ADC's frequency row.  Off (the default, the reference's HEAD) nothing changes; on,
    every path's record and row equals the oracle's; the shape's velocity transform scales the shift."""
    sd, lp = _doppler_scene(3.0)
    h_off, _, _ = _render_compare(sd, lp)
    lp_on = capi.make_launch(lp.mode, lp.n_paths, seed=lp.seed, bins=lp.bins, bins_y=lp.bins_y, flags=capi.BF_FLAG_DOPPLER)
    h_on, ho, _ = _render_compare(sd, lp_on)
    rows_off = h_off.reshape(32, 64, 3)[:, :, 2].sum(1)
    rows_on = h_on.reshape(32, 64, 3)[:, :, 2].sum(1)
    assert rows_off.sum() > 0 and not np.array_equal(rows_off, rows_on)        # the hook moves samples between frequency rows
    # identity velocity on every shape: another shift, still the oracle's
    sd1, _ = _doppler_scene(1.0)
    h1, _, _ = _render_compare(sd1, lp_on)
    assert not np.array_equal(h1, h_on)
    # the planned (asynchronous) second render of a shape keeps the per-slot shift
    g = capi.Scene(sd)
    out = OracleScene(sd).render(lp_on, records=True, threads=8)
    for _ in range(3):
        _render_compare_one(g, lp_on, out, 2e-5)


def test_mix_resample_receive_type(hiplib):
    """BF_FLAG_MIX_RESAMPLE (receive_type "mix_resample", integrator.cpp:1588-1603): the frequency row is that of the beat
    |c / lambda_after - f_rx|.  With the Doppler hook every path's record and row equals the oracle's (both pipelines);
    without it the beat is 0 and every sample is dropped, as at the reference's HEAD; render modes refuse the flag, and so does a
    Wigner receiver whose signal is no delta (test_fmcw_dechirp_with_the_receivers_local_oscillator has the delta case)."""
    sd, lp = _doppler_scene(3.0)
    sd.sensor.f_bandwidth = 0.5 * sd.physics.c / (sd.physics.lambda_min_nm * 1e-9)
    sd.finalize()
    both = capi.BF_FLAG_DOPPLER | capi.BF_FLAG_MIX_RESAMPLE
    lp_mix = capi.make_launch(lp.mode, lp.n_paths, seed=lp.seed, bins=lp.bins, bins_y=lp.bins_y, flags=both)
    h_mix, _, _ = _render_compare(sd, lp_mix)
    rows = h_mix.reshape(32, 64, 3)[:, :, 2].sum(1)
    assert rows.sum() > 0.3 * lp.n_paths and np.count_nonzero(rows) > 3
    lp_dop = capi.make_launch(lp.mode, lp.n_paths, seed=lp.seed, bins=lp.bins, bins_y=lp.bins_y, flags=capi.BF_FLAG_DOPPLER)
    h_dop, _, _ = _render_compare(sd, lp_dop)
    assert not np.array_equal(h_dop, h_mix)
    lp_only = capi.make_launch(lp.mode, lp.n_paths, seed=lp.seed, bins=lp.bins, bins_y=lp.bins_y, flags=capi.BF_FLAG_MIX_RESAMPLE)
    h0, _, _ = _render_compare(sd, lp_only)
    assert not h0.any()
    # a batch keeps the flag per render
    g = capi.Scene(sd)
    hb, rb, _ = g.render_batch(lp_mix, 2, seeds=[lp.seed, lp.seed + 1], records=True)
    assert np.allclose(hb[0], h_mix, rtol=1e-4, atol=1e-3)
    sdw, lpw = scenes.bus_receive(n_tris=500, n_paths=256, t_bins=4, receiver="wigner")
    sdw.sensor.rx_signal_type, sdw.sensor.rx_sig_is_delta = capi.BF_SIGNAL_PULSE, 1          # (a "pulse" that is a delta: uninitialised in the reference)
    sdw.finalize()
    lpw.flags = capi.BF_FLAG_MIX_RESAMPLE
    with pytest.raises(capi.BeifongError, match="delta"):
        capi.Scene(sdw).render(lpw)
    sdr, lpr = scenes.bus_radar(n_tris=500, n_paths=256, bins=16, dr=0.5)
    lpr.flags = capi.BF_FLAG_MIX_RESAMPLE
    with pytest.raises(capi.BeifongError):
        capi.Scene(sdr).render(lpr)


def test_quantised_nodes_opt_in_is_bit_exact(hiplib, monkeypatch):
    """BF_QUANT_BVH=1: wf_trace walks 64-byte nodes with 8-bit child boxes (bf_bvh.h: Node4Q; measured slower, hence
    opt-in).  Quantised boxes contain the fp32 boxes and box tests only select triangles: every path must still equal the
    oracle's, also after bf_scene_translate_meshes re-quantises the tree on the device, and in a batch with mesh offsets."""
    monkeypatch.setenv("BF_QUANT_BVH", "1")
    sd, lp = scenes.bus_radar(n_tris=20000, n_paths=1 << 16, bins=256, dr=0.1)
    g = capi.Scene(sd)
    assert g.info().trace_node_bytes == 64 and g.info().node_bytes == 128
    out = OracleScene(sd).render(lp, records=True, threads=8)
    _, _, st = _render_compare_one(g, lp, out, 2e-5)
    assert st.n_rays_traced > 0
    monkeypatch.delenv("BF_QUANT_BVH")
    ref = capi.Scene(sd)
    assert ref.info().trace_node_bytes == 128
    off = [-0.37, 0.21, 0.02]
    g.translate_meshes(off)
    ref.translate_meshes(off)
    _, rq, _ = g.render(lp, records=True)
    _, rr, _ = ref.render(lp, records=True)
    for k in ("L", "aux"):
        assert np.array_equal(rq[k].view(np.uint32), rr[k].view(np.uint32))
    assert np.array_equal(rq["n_rays"], rr["n_rays"])
    g.translate_meshes([0, 0, 0])
    hb, rb, _ = g.render_batch(lp, 2, offsets=[[0, 0, 0], off], records=True)
    for k in ("L", "aux"):
        assert np.array_equal(rb[1][k].view(np.uint32), rr[k].view(np.uint32))
        assert np.array_equal(rb[0][k].view(np.uint32), out[1][k].view(np.uint32))


@pytest.mark.parametrize("case", ["bus_range", "car_normals", "bus_receive", "bus_receive_iq", "not_lean_two_emitters"])
def test_lean_and_general_kernels_agree(hiplib, monkeypatch, case):
    """Scenes that fit the lean profile (bf_stats.kernel_variant) run shading / tail kernels with everything outside the profile
    compiled out; BF_LEAN=0 (read when the scene is created) forces the general kernels.  Same per-path records, bit for bit, from
    both — plain, planned and rolling renders — and both equal the oracle's; a scene outside the profile never gets the lean ones."""
    if case == "bus_range":
        sd, lp = scenes.bus_radar(n_tris=20000, n_paths=30000, bins=256, dr=0.1)
    elif case == "car_normals":
        sd, lp = scenes.car_radar(n_tris=30000, n_paths=30000, bins=1024, dr=0.03)
    elif case == "not_lean_two_emitters":
        sd, lp = _zoo_scene(two_emitters=True)
    else:
        sd, lp = scenes.bus_receive(n_tris=20000, n_paths=30000)
        if case == "bus_receive_iq":
            lp.mode = capi.BF_MODE_RECEIVE_IQ
    ho, ro, so = OracleScene(sd).render(lp, records=True, threads=8)
    out = {}
    for lean in ("1", "0"):
        monkeypatch.setenv("BF_LEAN", lean)
        g = capi.Scene(sd)
        h1, r1, s1 = g.render(lp, records=True)
        want = capi.BF_VARIANT_LEAN if (lean == "1" and case != "not_lean_two_emitters") else 0
        if os.environ.get("BF_SHADE_WAVES", "3") != "3" or os.environ.get("BF_TAIL_WAVES", "3") != "3":
            want = s1.kernel_variant      # (developer knobs: the lean builds exist for three waves per SIMD only; tools/r04_knob_matrix.sh)
        assert s1.kernel_variant == want
        h2, r2, s2 = g.render(lp, records=True)                 # the planned render (no host synchronisation inside)
        assert np.array_equal(r1, r2) and s2.kernel_variant == want
        lp.flags |= capi.BF_FLAG_MEGAKERNEL                       # the one-kernel variant is always the general build
        assert g.render(lp, records=True)[2].kernel_variant == 0
        lp.flags &= ~capi.BF_FLAG_MEGAKERNEL
        for k in ("n_rays", "valid"):
            assert np.array_equal(r1[k], ro[k])
        assert np.array_equal(r1["aux"].view(np.uint32), ro["aux"].view(np.uint32))
        assert np.array_equal(r1["L"].view(np.uint32), ro["L"].view(np.uint32))
        assert s1.n_rays_closest == so.n_rays_closest and s1.n_rays_shadow == so.n_rays_shadow and s1.n_invalid == so.n_invalid
        out[lean] = (h1, r1)
    assert np.array_equal(out["1"][1], out["0"][1])
    assert np.allclose(out["1"][0], out["0"][0], rtol=2e-5, atol=1e-3)


@pytest.mark.parametrize("receive", [False, True])
def test_twosided_with_two_nested_bsdfs(hiplib, receive):
    """<bsdf type="twosided"> with a rough conductor in front and a bright diffuse BSDF behind (twosided.cpp:62-178): a plate in
    front of the camera that shows it its BACK, and the zoo scene's conductor mesh given a diffuse inside.  Every path against
    the oracle; and the back side really is the other BSDF (records differ from the same scene with the front BSDF on both sides)."""
    T = Transform4f
    out = {}
    for two in (False, True):
        sd, lp = _zoo_scene(two_emitters=not receive, receive=receive)
        front = sd.add_roughconductor(alpha=0.2, twosided=True, specular_reflectance=0.6)
        # the rectangle's normal is +z; turned to +x it points AWAY from the camera at the origin, which looks along +x
        sd.add_rectangle(T.translate([2.0, -0.2, 0.9]) * T.rotate([0, 1, 0], 90) * T.scale([1.0, 1.2, 1]), front)
        if two:
            sd.set_back_material(front, sd.add_diffuse(reflectance=0.85, twosided=True))
            for i in range(len(sd.materials)):
                if sd.materials[i].type == capi.BF_BSDF_ROUGHCONDUCTOR and i != front:
                    sd.set_back_material(i, sd.add_diffuse(reflectance=0.3, twosided=True))
        sd.finalize()
        hg, ho, st = _render_compare(sd, lp)
        assert st.kernel_variant == 0 or not two            # general kernels: the lean profile has one BSDF per material
        out[two] = capi.Scene(sd).render(lp, records=True)[1]
    changed = (out[True]["L"].view(np.uint32) != out[False]["L"].view(np.uint32)).mean()
    assert changed > 0.003, changed      # the paths that met the plate from behind (the receiver's cosine lobe sees less of it)
    bad = sd.materials[front].back_material
    sd.materials[front].back_material = len(sd.materials) + 5
    sd.finalize()
    with pytest.raises(capi.BeifongError, match="back_material"):
        capi.Scene(sd)
    sd.materials[front].back_material = bad


def _fmcw_scene(n_paths=30000, resample=True, signal="linfmcw", transmitter="wigner", f_bins=32):
    """C2-recv geometry with an FMCW transmitter: one chirp of sweep B over the whole ADC window T (crf = 1 / T), a receiver
    band equal to the chirp's, a 64 x f_bins ADC whose frequency axis spans B (the beat frequencies of "mix_resample")."""
    lam = 8.6e6                                                    # nm: ~35 GHz
    if transmitter == "phased":
        sd, lp = scenes.phased_receive(n_tris=5000, n_paths=n_paths, phased_rx=False, phased_tx=True)
    else:
        sd, lp = scenes.bus_receive(n_tris=5000, n_paths=n_paths, t_bins=64, lambda_band_nm=(lam * 0.999, lam * 1.001))
    c, lmin, lmax = sd.physics.c, sd.physics.lambda_min_nm, sd.physics.lambda_max_nm
    e = sd.emitters[0]
    e.signal_type = {"linfmcw": capi.BF_SIGNAL_LINFMCW, "cw": capi.BF_SIGNAL_CW, "pulse": capi.BF_SIGNAL_PULSE}[signal]
    band = c / (lmin * 1e-9) - c / (lmax * 1e-9)
    e.freq_ext = band
    e.pulse_len = sd.sensor.t_bandwidth                            # chirp_len = T
    e.prf = 1.0 / sd.sensor.t_bandwidth
    e.resample_freq = int(resample)
    sd.sensor.f_bins = f_bins
    sd.sensor.f_bandwidth = band
    sd.finalize()
    lp.bins_y = f_bins
    return sd, lp


def test_resample_freq_transmitter(hiplib):
    """m_resample_freq (wignertransmitter.cpp:211-221, 430-441; phasedtransmitter.cpp likewise): Transmitter::eval and
    sample_direction overwrite the interaction's wavelength with the signal's instantaneous frequency at the (retarded) time,
    spawn_ray carries it on and the caller's ray ends with it (pathtimefrequency.cpp:451) — the ADC's frequency row is the
    transmitter's frequency at emission ("raw") or its beat with what the receiver sampled ("mix_resample": the de-chirped FMCW
    return).  Every per-path record and every histogram cell equals the oracle's: both pipelines, I/Q mode, a rolling sequence
    (the receiver's sample rides in the path state), the phased transmitter; "pulse" and the Doppler hook are refused."""
    sd, lp = _fmcw_scene()
    h_raw, _, st = _render_compare(sd, lp)                          # ("raw": absolute frequencies, far above this ADC's rows)
    assert st.kernel_variant == 0                                   # general kernels: the lean profile has no re-sampling
    sd0, _ = _fmcw_scene(resample=False)
    _, r_on, _ = capi.Scene(sd).render(lp, records=True)
    _, r_off, _ = capi.Scene(sd0).render(lp, records=True)
    assert not np.array_equal(r_on["L"], r_off["L"])                # the Wigner gains see the re-sampled wavelength, power 1
    # "mix_resample": |f_tx(t_emit) - f_rx| — rows spread over the sweep
    lp_mix = capi.make_launch(lp.mode, lp.n_paths, seed=lp.seed, bins=lp.bins, bins_y=lp.bins_y, flags=capi.BF_FLAG_MIX_RESAMPLE)
    h_mix, _, _ = _render_compare(sd, lp_mix)
    rows = h_mix.reshape(32, 64, 3)[:, :, 2].sum(1)
    assert rows.sum() > 0.3 * lp.n_paths and np.count_nonzero(rows) > 8 and not np.array_equal(h_mix, h_raw)
    # I/Q mode and "cw" (the carrier: every return at f_centre)
    lp_iq = capi.make_launch(capi.BF_MODE_RECEIVE_IQ, lp.n_paths, seed=lp.seed, bins=lp.bins, bins_y=lp.bins_y)
    _render_compare(sd, lp_iq)
    sd_cw, _ = _fmcw_scene(signal="cw")
    h_cw, _, _ = _render_compare(sd_cw, lp_mix)                     # |f_centre - f_rx|: half the sweep at most
    rows_cw = h_cw.reshape(32, 64, 3)[:, :, 2].sum(1)
    assert rows_cw[:16].sum() > 0 and rows_cw[17:].sum() == 0
    # the phased transmitter shares the signal model
    sd_ph, lp_ph = _fmcw_scene(n_paths=8000, transmitter="phased")
    _render_compare(sd_ph, lp_ph)
    # a rolling sequence: paths wait in the pool between launches with BOTH wavelengths in their state
    import torch
    g = capi.Scene(sd)
    seeds = [5, 6, 7]
    nch = g.channels(lp_mix)
    hist = torch.zeros((len(seeds), nch), dtype=torch.float32, device="cuda")
    rec = torch.zeros((len(seeds), lp.n_paths, 4), dtype=torch.int32, device="cuda")
    for k, seed in enumerate(seeds):
        l = capi.make_launch(lp.mode, lp.n_paths, seed=seed, bins=lp.bins, bins_y=lp.bins_y,
                             flags=capi.BF_FLAG_MIX_RESAMPLE | capi.BF_FLAG_ROLLING)
        g.render_device(l, hist[k].data_ptr(), records_ptr=rec[k].data_ptr())
    g.flush()
    torch.cuda.synchronize()
    o = OracleScene(sd)
    for k, seed in enumerate(seeds):
        l = capi.make_launch(lp.mode, lp.n_paths, seed=seed, bins=lp.bins, bins_y=lp.bins_y, flags=capi.BF_FLAG_MIX_RESAMPLE)
        ho, ro, _ = o.render(l, records=True, threads=8)
        rg = rec[k].cpu().numpy().view(np.uint32).reshape(-1, 4)
        rg = np.ascontiguousarray(rg).view(capi.PATH_RECORD_DTYPE).reshape(-1)
        assert np.array_equal(rg["L"].view(np.uint32), ro["L"].view(np.uint32)) and np.array_equal(rg["n_rays"], ro["n_rays"])
        assert np.allclose(hist[k].cpu().numpy(), ho, rtol=2e-5, atol=lp.n_paths * 2.0 ** -22 * max(1.0, float(np.abs(ro["L"]).max())))
    # refused: "pulse" (sample_delta_frequency leaves the frequency uninitialised), the Doppler hook next to re-sampling
    sd_p, _ = _fmcw_scene(signal="pulse")
    with pytest.raises(capi.BeifongError, match="pulse"):
        capi.Scene(sd_p)
    lp_d = capi.make_launch(lp.mode, lp.n_paths, seed=lp.seed, bins=lp.bins, bins_y=lp.bins_y, flags=capi.BF_FLAG_DOPPLER)
    with pytest.raises(capi.BeifongError, match="DOPPLER"):
        capi.Scene(sd).render(lp_d)


@pytest.mark.parametrize("receiver", ["wigner", "phased"])
def test_fmcw_dechirp_with_the_receivers_local_oscillator(hiplib, receiver):
    """receive_type "mix_resample" on the Wigner / phased receiver (wignerreceiver.cpp:72-110, 149-189): the receiver's frequency
    sample is its own chirp's instantaneous frequency at the receive time; against a resample_freq transmitter with the same chirp
    the ADC's frequency axis is the de-chirped beat |f_tx(t - delay) - f_lo(t)| = (sweep / chirp_len) * delay — every return of
    one range in ONE frequency row, rows proportional to the time bin of the same return.  Per-path parity with the oracle in
    both pipelines; the "raw" receive types of the same scene do not read the local oscillator."""
    lam = 8.6e6
    if receiver == "phased":
        sd, lp = scenes.phased_receive(n_tris=5000, n_paths=20000, phased_rx=True, phased_tx=False)
    else:
        sd, lp = scenes.bus_receive(n_tris=5000, n_paths=30000, t_bins=64, receiver="wigner", lambda_band_nm=(lam * 0.999, lam * 1.001))
    c = sd.physics.c
    T = sd.sensor.t_bandwidth
    f_c = sd.emitters[0].freq_centre
    sweep = 0.002 * f_c
    e = sd.emitters[0]
    e.signal_type, e.freq_ext, e.pulse_len, e.prf, e.resample_freq = capi.BF_SIGNAL_LINFMCW, sweep, T, 1.0 / T, 1
    s = sd.sensor
    s.freq_centre, s.freq_ext, s.rx_sig_is_delta = f_c, sweep, 1
    s.rx_signal_type, s.rx_pulse_len, s.rx_prf = capi.BF_SIGNAL_LINFMCW, T, 1.0 / T
    s.f_bins, s.f_bandwidth = 64, sweep                   # beat = sweep * delay / T: the frequency row IS the delay's time bin
    sd.finalize()
    lp.bins_y = 64
    lp_mix = capi.make_launch(lp.mode, lp.n_paths, seed=lp.seed, bins=lp.bins, bins_y=64, flags=capi.BF_FLAG_MIX_RESAMPLE)
    h, _, st = _render_compare(sd, lp_mix)
    assert st.kernel_variant == 0
    cube = h.reshape(64, lp.bins, 3)[:, :, 2]                                # [frequency row][time bin] sample counts
    assert cube.sum() > 0.2 * lp.n_paths
    # de-chirped: the beat is the delay's, whenever the return is received — the row profile (the scene's range profile: ground and
    # bus) is the same in every part of the receive window (past the first returns' wrap into the previous chirp)
    nt = lp.bins
    p1, p2 = cube[:, nt // 4:nt // 2].sum(1), cube[:, nt // 2:].sum(1)
    assert p1.sum() > 0 and p2.sum() > 0
    assert np.corrcoef(p1 / p1.sum(), p2 / p2.sum())[0, 1] > 0.95
    # "raw" on the same scene: parity too, and another histogram (absolute frequencies: outside this ADC)
    h_raw, _, _ = _render_compare(sd, lp)
    assert not np.array_equal(h_raw, h)


@pytest.mark.parametrize("signal", ["pulse", "linfmcw", "cw"])
def test_mix_resample_receiver_signal_that_is_no_delta(hiplib, signal):
    """The other branch of the Wigner receiver's sample_frequency under "mix_resample" (wignerreceiver.cpp:179-186): a uniform
    frequency from [f_centre - f_ext / 2, f_centre + f_ext / 2] weighted with the receiver's eval_signal(time, f) (:118-142 —
    the pulse's / chirp's Wigner function, or amplitude^2 for "cw"), and the band's extent in the ray weight (:258).  Per-path
    parity with the oracle against a resample_freq transmitter."""
    sd, lp = _fmcw_scene(n_paths=20000)
    s = sd.sensor
    s.type = capi.BF_RECEIVER_WIGNER
    s.rx_sig_is_delta = 0
    s.rx_signal_type = {"pulse": capi.BF_SIGNAL_PULSE, "linfmcw": capi.BF_SIGNAL_LINFMCW, "cw": capi.BF_SIGNAL_CW}[signal]
    s.rx_pulse_len, s.rx_prf, s.rx_amplitude = 0.25 * s.t_bandwidth, 1.0 / s.t_bandwidth, 1.5
    s.freq_centre, s.freq_ext = sd.emitters[0].freq_centre, sd.emitters[0].freq_ext
    sd.finalize()
    lp_mix = capi.make_launch(lp.mode, lp.n_paths, seed=lp.seed, bins=lp.bins, bins_y=lp.bins_y, flags=capi.BF_FLAG_MIX_RESAMPLE)
    h, _, _ = _render_compare(sd, lp_mix)
    assert h.reshape(32, 64, 3)[:, :, 2].sum() > 0


def test_fmcw_beat_row_is_the_range_of_a_plate(hiplib):
    """A physical known answer for the de-chirp path (resample_freq transmitter + the Wigner receiver's local oscillator under
    "mix_resample"): a 0.5 m plate R = 5 m in front of coincident TX / RX apertures; both ends sweep B over the receive window T, so
    a return delayed by 2 R / c beats at B / T * 2 R / c whenever it is received.  With an ADC whose 64 frequency rows span the beat
    of 12.8 m, the plate's energy sits in rows 24 / 25 (SignalBlock::put: ceil(64 r / 12.8 - 1), r = 5.00 .. 5.01 m) — and the
    engine's every path equals the oracle's."""
    sd, lp = scenes.fmcw_plate(plate_x=5.0, r_max=12.8, n_paths=1 << 18)
    h, _, _ = _render_compare(sd, lp)
    energy = np.abs(h.reshape(64, 8, 3)[:, :, 0]).sum(1)               # per beat row, over the receive window
    assert energy.sum() > 0
    # slant ranges 5.000 .. 5.013 m (the plate is 0.5 m wide): pos = 64 * r / 12.8 = 25.0 .. 25.06 -> rows 24 (r = 5 exactly) and 25
    assert int(np.argmax(energy)) in (24, 25), energy[20:30]
    assert energy[24:26].sum() > 0.95 * energy.sum()
