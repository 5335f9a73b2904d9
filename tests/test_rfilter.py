"""Reconstruction filters wider than a pixel (ImageBlock::put / SignalBlock::put, filtered branch).

CPU part: the oracle's restatement of src/rfilters/*.cpp + ReconstructionFilter::init_discretization is pinned by the
reference's own spot checks (src/rfilters/tests/test_rfilter.py:8-56) and by its put() test with a Gaussian filter
(src/librender/tests/test_imageblock.py:145-213, the scalar half); the host plugins (the product's filters) must produce
the oracle's tables bit for bit.  GPU part: HIP renders against the oracle with such filters on the film / the ADC."""
import ctypes as C

import numpy as np
import pytest

from beifong_amd import capi, scenes
from beifong_amd.scenedesc import Transform4f
from tests import oracle_lib

KINDS = {"box": 0, "tent": 1, "gaussian": 2, "mitchell": 3, "catmullrom": 4, "lanczos": 5}
DEFAULTS = {"box": (0.5, 0.0), "tent": (0.0, 0.0), "gaussian": (0.5, 0.0), "mitchell": (1 / 3, 1 / 3), "catmullrom": (0.0, 0.0),
            "lanczos": (3.0, 0.0)}


@pytest.fixture(scope="module")
def oracle():
    lib = oracle_lib.load()
    lib.bfo_rfilter.argtypes = [C.c_int, C.c_float, C.c_float, C.POINTER(capi.bf_rfilter)]
    lib.bfo_rfilter_eval.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_int]
    lib.bfo_rfilter_eval.restype = C.c_float
    lib.bfo_imageblock_put_filtered.argtypes = [C.POINTER(capi.bf_rfilter), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                                C.c_float, C.c_float, C.c_void_p]
    return lib


class OFilter:
    """the oracle's filter with the reference's Python face (eval / eval_discretized / radius / border_size)"""

    def __init__(self, lib, kind, p0=None, p1=None):
        d = DEFAULTS[kind]
        self.lib, self.k = lib, KINDS[kind]
        self.p0, self.p1 = d[0] if p0 is None else p0, d[1] if p1 is None else p1
        self.flat = capi.bf_rfilter()
        lib.bfo_rfilter(self.k, self.p0, self.p1, C.byref(self.flat))

    def eval(self, x):
        return self.lib.bfo_rfilter_eval(self.k, self.p0, self.p1, x, 0)

    def eval_discretized(self, x):
        return self.lib.bfo_rfilter_eval(self.k, self.p0, self.p1, x, 1)

    def radius(self):
        return self.flat.radius

    def border_size(self):
        return self.flat.border


def _host_filter(kind, **props):
    from beifong_amd import mitsuba
    mitsuba.set_variant("scalar_rgb")
    from beifong_amd.mitsuba.core.xml import load_string
    body = "".join("<%s name='%s' value='%s'/>" % ("integer" if isinstance(v, int) else "float", k, v) for k, v in props.items())
    return load_string("<rfilter version='2.0.0' type='%s'>%s</rfilter>" % (kind, body))


def _spot_checks(make):
    # src/rfilters/tests/test_rfilter.py:8-56, test by test (the reference's own tolerances; test02 there reads
    # `assert ek,allclose(...)`, a tuple that is always true — the value it names is checked here all the same)
    f = make("box")
    assert f.eval(0.49) == 1 and f.eval(0.51) == 0
    assert f.eval_discretized(0.49) == 1 and f.eval_discretized(0.51) == 0
    f = make("gaussian")
    assert abs(f.eval(0.2) - 0.9227) < 8e-3 and abs(f.eval_discretized(0.2) - 0.9227) < 8e-3
    assert f.eval(2.1) == 0 and f.eval_discretized(2.1) == 0
    f = make("lanczos")
    assert abs(f.eval(1.4) - -0.14668) < 1e-2 and abs(f.eval_discretized(1.4) - -0.14668) < 1e-2
    assert f.eval(3.1) == 0 and f.eval_discretized(3.1) == 0
    f = make("mitchell")
    assert abs(f.eval(0) - 0.8888) < 1e-3 and abs(f.eval_discretized(0) - 0.8888) < 1e-3
    assert f.eval(2.1) == 0 and f.eval_discretized(2.1) == 0
    f = make("catmullrom")
    assert abs(f.eval(0) - 0.9765) < 5e-2 and abs(f.eval_discretized(0) - 0.9765) < 5e-2
    assert f.eval(2.1) == 0 and f.eval_discretized(2.1) == 0
    f = make("tent")
    assert abs(f.eval(0.1) - 0.903) < 5e-2 and abs(f.eval_discretized(0.1) - 0.903) < 5e-2
    assert f.eval(1.1) == 0 and f.eval_discretized(1.1) == 0


def test_reference_spot_checks_on_the_oracle(oracle):
    _spot_checks(lambda kind: OFilter(oracle, kind))


def test_reference_spot_checks_on_the_host_plugins():
    _spot_checks(_host_filter)


def test_closed_forms(oracle):
    # the filters' definitions evaluated in double (gaussian.cpp:41-47, tent.cpp:35, mitchell.cpp:41-55, lanczos.cpp:43-52)
    g = OFilter(oracle, "gaussian", 0.7)
    m = OFilter(oracle, "mitchell")
    la = OFilter(oracle, "lanczos", 2)
    for x in np.linspace(-3.2, 3.2, 41):
        a = -1.0 / (2 * 0.7 * 0.7)
        assert abs(g.eval(x) - max(0.0, np.exp(a * x * x) - np.exp(a * 2.8 * 2.8))) < 2e-7
        assert abs(OFilter(oracle, "tent").eval(x) - max(0.0, 1 - abs(x))) < 1e-7
        ax, B, Cc = abs(x), 1 / 3, 1 / 3
        mm = ((12 - 9 * B - 6 * Cc) * ax ** 3 + (-18 + 12 * B + 6 * Cc) * ax ** 2 + (6 - 2 * B)) / 6 if ax < 1 else \
            ((-B - 6 * Cc) * ax ** 3 + (6 * B + 30 * Cc) * ax ** 2 + (-12 * B - 48 * Cc) * ax + (8 * B + 24 * Cc)) / 6 if ax < 2 else 0.0
        assert abs(m.eval(x) - mm) < 2e-6
        ll = 1.0 if ax < 1e-9 else (0.0 if ax > 2 else np.sin(np.pi * ax) * np.sin(np.pi * ax / 2) / (np.pi * ax * np.pi * ax / 2))
        assert abs(la.eval(x) - ll) < 2e-6
    # init_discretization (rfilter.cpp:9-21): 31 samples of eval on [0, radius), a closing zero, scale and border
    for kind, p0 in (("gaussian", 0.5), ("gaussian", 1.3), ("lanczos", 3), ("box", 0.5), ("box", 0.85)):
        f = OFilter(oracle, kind, p0)
        r = np.float32(f.radius())
        for i in range(31):
            assert f.flat.values[i] == f.eval(float(np.float32(r * np.float32(i)) / np.float32(31)))
        assert f.flat.values[31] == 0 and f.flat.scale == np.float32(31) / r
        assert f.border_size() == int(np.ceil(np.float32(r - np.float32(0.5)) - np.float32(2 * 1500 * 2.0 ** -24)))
    assert OFilter(oracle, "box").border_size() == 0 and OFilter(oracle, "gaussian").border_size() == 2


@pytest.mark.parametrize("kind,props", [("box", {}), ("box", {"radius": 0.85}), ("tent", {}), ("gaussian", {}), ("gaussian", {"stddev": 1.25}),
                                        ("mitchell", {}), ("mitchell", {"B": 0.2, "C": 0.4}), ("catmullrom", {}), ("lanczos", {}),
                                        ("lanczos", {"lobes": 2})])
def test_host_plugins_produce_the_oracles_tables(oracle, kind, props):
    h = _host_filter(kind, **props).flatten(7)
    vals = list(props.values())
    o = OFilter(oracle, kind, *[float(v) for v in vals]).flat
    assert h.block_size == 7 and o.block_size == 0
    assert (h.radius, h.scale, h.border) == (o.radius, o.scale, o.border)
    assert np.array_equal(np.array(h.values[:]).view(np.uint32), np.array(o.values[:]).view(np.uint32))


M_SRGB_TO_XYZ = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]], dtype=np.float32)


def test_put_with_filter_like_the_reference_test(oracle):
    """test_imageblock.py:145-213 (test05_put_with_filter), scalar half: ten samples through a Gaussian filter (stddev 0.5)
    into a 12 x 12 block with its border, against the window sum the reference's test writes out in numpy."""
    rng = np.random.default_rng(11)
    f = OFilter(oracle, "gaussian", 0.5)
    size = (12, 12)
    border = f.border_size()
    radius = int(np.ceil(f.radius()))
    positions = np.array([[5, 6], [0, 1], [5, 6], [1, 11], [11, 11], [0, 1], [2, 5], [4, 1], [0, 11], [5, 4]], dtype=np.float64)
    n = positions.shape[0]
    positions += rng.uniform(size=positions.shape, low=0, high=0.95)
    spectra = np.arange(n * 3).reshape((n, 3))
    block = np.zeros((size[1] + 2 * border, size[0] + 2 * border, 5))
    ref = np.zeros_like(block)
    for i in range(n):
        xyz = M_SRGB_TO_XYZ @ spectra[i].astype(np.float32)
        value = np.array([xyz[0], xyz[1], xyz[2], 1.0, 1.0], dtype=np.float32)
        assert oracle.bfo_imageblock_put_filtered(C.byref(f.flat), block.ctypes.data, size[0], size[1], 5, 0, 0, positions[i, 0], positions[i, 1],
                                                  value.ctypes.data) == 1
        pos = positions[i] - 0.5 + border
        lo = np.ceil(pos - radius).astype(int)
        hi = np.floor(pos + radius).astype(int)
        for dy in range(lo[1], hi[1] + 1):
            for dx in range(lo[0], hi[0] + 1):
                r_pos = np.array([dx, dy])
                if np.any(r_pos < 0) or np.any(r_pos >= ref.shape[:2]):
                    continue
                w_pos = r_pos - pos
                weight = f.eval_discretized(w_pos[0]) * f.eval_discretized(w_pos[1])
                ref[dy, dx, :3] += weight * xyz
                ref[dy, dx, 3] += weight
                ref[dy, dx, 4] += weight
    assert np.abs(block).max() > 1 and np.allclose(block, ref, atol=1e-6 * max(1.0, np.abs(ref).max()))
    # a non-finite value drops the sample (warn_invalid, imageblock.cpp:85-111)
    bad = np.array([1, np.inf, 1, 1, 1], dtype=np.float32)
    before = block.copy()
    assert oracle.bfo_imageblock_put_filtered(C.byref(f.flat), block.ctypes.data, 12, 12, 5, 0, 0, 3.3, 3.3, bad.ctypes.data) == 0
    assert np.array_equal(block, before)
    # the weights of one sample sum to (sum of the discretised taps)^2, whatever its sub-pixel position well inside the block
    one = np.zeros((12 + 2 * border, 12 + 2 * border, 1))
    v = np.ones(1, dtype=np.float32)
    oracle.bfo_imageblock_put_filtered(C.byref(f.flat), one.ctypes.data, 12, 12, 1, 0, 0, 6.25, 5.75, v.ctypes.data)
    # pos = 6.25 + 1.5 = 7.75 -> lo = 6, taps at -1.75 .. 1.25; 5.75 + 1.5 = 7.25 -> lo = 6, taps at -1.25 .. 1.75 (n = 4)
    tx = [f.eval_discretized(-1.75 + k) for k in range(4)]
    ty = [f.eval_discretized(-1.25 + k) for k in range(4)]
    assert (one != 0).sum() == 16 and np.array_equal(one[6:10, 6:10, 0], np.outer(np.float32(ty), np.float32(tx)).astype(np.float64))


def test_xml_default_filter_is_gaussian_and_reaches_the_flat_scene():
    """film.cpp / adc.cpp:70-75: no <rfilter> child means the Gaussian; the flattened sensor carries its table."""
    from beifong_amd import mitsuba
    mitsuba.set_variant("scalar_rgb")
    from beifong_amd.mitsuba.core.xml import load_string
    xml = """<scene version='2.0.0'>
        <integrator type='path'/>
        <sensor type='perspective'><film type='hdrfilm'><integer name='width' value='8'/><integer name='height' value='4'/>%s</film>
            <sampler type='independent'><integer name='sample_count' value='4'/></sampler></sensor>
        <shape type='rectangle'><emitter type='area'><spectrum name='radiance' value='1'/></emitter></shape>
    </scene>"""
    for child, radius, border in (("", 2.0, 2), ("<rfilter type='box'/>", np.float32(0.5) + np.float32(1500 * 2.0 ** -24), 0),
                                  ("<rfilter type='lanczos'><integer name='lobes' value='2'/></rfilter>", 2.0, 2)):
        sc = load_string(xml % child)
        d = sc.flat_desc(sc.sensors()[0]).desc
        assert d.sensor.rfilter.radius == radius and d.sensor.rfilter.border == border and d.sensor.rfilter.block_size == 32


def test_load_dict_builds_filters_and_embeds_them():
    """src/python/python/xml.py load_dict: a filter dictionary gives the same object as the XML, stands on its own, and can be
    embedded in a film / ADC dictionary."""
    from beifong_amd import mitsuba
    mitsuba.set_variant("scalar_rgb")
    from beifong_amd.mitsuba.core.xml import load_dict
    f = load_dict({"type": "gaussian", "stddev": 0.7})
    x = _host_filter("gaussian", stddev=0.7)
    assert (f.radius(), f.border_size(), f.eval(0.3), f.eval_discretized(1.1)) == (x.radius(), x.border_size(), x.eval(0.3), x.eval_discretized(1.1))
    scene = load_dict({"type": "scene", "integrator": {"type": "path"},
                       "sensor": {"type": "perspective", "film": {"type": "hdrfilm", "width": 4, "height": 2, "rfilter": load_dict({"type": "lanczos", "lobes": 2})},
                                  "sampler": {"type": "independent", "sample_count": 4}},
                       "light": {"type": "rectangle", "emitter": {"type": "area", "radiance": {"type": "spectrum", "value": 1.0}}}})
    d = scene.flat_desc(scene.sensors()[0]).desc
    assert d.sensor.rfilter.radius == 2.0 and d.sensor.rfilter.border == 2
    assert list(d.sensor.rfilter.values) == list(_host_filter("lanczos", lobes=2).flatten().values)


# ---------------------------------------------------------------------------------------------------------------------
# HIP against the oracle
# ---------------------------------------------------------------------------------------------------------------------
def _records_equal(rg, ro):
    for k in ("n_rays", "valid"):
        assert np.array_equal(rg[k], ro[k])
    assert np.array_equal(rg["aux"].view(np.uint32), ro["aux"].view(np.uint32))
    assert np.array_equal(rg["L"].view(np.uint32), ro["L"].view(np.uint32))


def _hist_close(hg, ho, n, amax):
    # same addends (value * wy * wx in fp32), different summation order (fp32 atomics against double sums)
    assert np.allclose(hg, ho, rtol=3e-5, atol=n * 2.0 ** -24 * max(amax, 1.0) * 4), np.abs(hg - ho).max()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,props,block", [("gaussian", {}, 32), ("gaussian", {"stddev": 0.9}, 4), ("lanczos", {}, 0), ("mitchell", {}, 2),
                                              ("tent", {}, 32), ("box", {"radius": 0.85}, 3)])
def test_film_with_a_wide_filter(hiplib, oracle, kind, props, block):
    """A W x H range image through a perspective camera (the zoo scene of test_gpu_parity) with a filter that spreads every
    sample over its neighbours — negative lobes included; both device pipelines and the global-atomics fallback."""
    from tests.test_gpu_parity import _zoo_scene
    from tests.oracle_lib import OracleScene
    film, spp, bins = (9, 6), 48, 64
    sd, _ = _zoo_scene(two_emitters=True)
    T = Transform4f
    sd.set_perspective(T.translate([0, 0, 0.3]) * T.rotate([1, 0, 0], 90) * T.rotate([0, 1, 0], 90), fov=60.0, near_clip=0.1,
                       far_clip=100.0, film=film)
    sd.sensor.rfilter = _host_filter(kind, **props).flatten(block)
    sd.finalize()
    lp = capi.make_launch(capi.BF_MODE_RANGE, film[0] * film[1] * spp, seed=5, bins=bins, bin_width=0.2, color_mode=capi.BF_COLOR_RGB, film=film,
                          spp=spp)
    ho, ro, so = OracleScene(sd).render(lp, records=True, threads=8)
    g = capi.Scene(sd)
    chan = 5 + bins
    for flags in (capi.BF_FLAG_MEGAKERNEL, 0, capi.BF_FLAG_GLOBAL_ATOMICS):
        lp.flags = flags
        hg, rg, sg = g.render(lp, records=True)
        _records_equal(rg, ro)
        assert sg.n_invalid == so.n_invalid == 0
        _hist_close(hg, ho, spp * 16, float(np.abs(ro["L"]).max()))
    img = hg.reshape(film[1], film[0], chan)
    # the filter really spread the samples: the weight channel is no longer the integer sample count of the box filter
    assert not np.array_equal(img[:, :, 4], np.full((film[1], film[0]), float(spp)))
    assert (img[:, :, 5:].sum(axis=2) != 0).mean() > 0.3
    # against the same scene with the box filter: the per-path records do not depend on the filter
    sd.sensor.rfilter = capi.bf_rfilter()
    sd.finalize()
    lp.flags = 0
    _records_equal(capi.Scene(sd).render(lp, records=True)[1], ro)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [capi.BF_MODE_PATH, capi.BF_MODE_RANGE, capi.BF_MODE_TIME])
def test_single_pixel_film_with_the_default_gaussian(hiplib, oracle, mode):
    """The radar scenes' 1 x 1 film under the film plugins' DEFAULT filter: every sample lands in the one pixel with the weight
    of its sub-pixel offset (all channels, the range / time bins included); plain launch, shards and a rolling sequence."""
    from tests.oracle_lib import OracleScene
    sd, lp = scenes.trans_rad(spp=6000)
    lp.mode = mode
    if mode == capi.BF_MODE_RANGE:
        lp.bins, lp.bin_width = 64, 0.25
    sd.sensor.rfilter = _host_filter("gaussian").flatten(32)
    sd.finalize()
    ho, ro, so = OracleScene(sd).render(lp, records=True, threads=8)
    g = capi.Scene(sd)
    n = lp.n_paths
    amax = float(np.abs(ro["L"]).max())
    for flags in (capi.BF_FLAG_MEGAKERNEL, 0):
        lp.flags = flags
        hg, rg, sg = g.render(lp, records=True)
        _records_equal(rg, ro)
        _hist_close(hg, ho, n, amax)
    assert 0 < hg[4] < n and hg[4] != np.round(hg[4])           # W: the sum of the samples' weights
    # two shards as a rolling sequence on one handle
    half = n // 2 + 5
    parts = np.zeros((2, len(ho)), dtype=np.float32)
    import torch
    dev = torch.zeros((2, len(ho)), dtype=torch.float32, device="cuda")
    for k, (off, cnt) in enumerate(((0, half), (half, n - half))):
        l2 = capi.make_launch(mode, cnt, seed=lp.seed, path_offset=off, bins=lp.bins, bin_width=lp.bin_width, color_mode=lp.color_mode,
                              max_depth=lp.max_depth, rr_depth=lp.rr_depth)
        l2.time_c = lp.time_c
        l2.flags = capi.BF_FLAG_ROLLING
        g.render_device(l2, dev[k].data_ptr())
    g.flush()
    torch.cuda.synchronize()
    parts = dev.cpu().numpy()
    _hist_close(parts.sum(0), ho, n, amax)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,f_bins,iq", [("gaussian", 1, False), ("gaussian", 16, False), ("tent", 16, True), ("lanczos", 1, False)])
def test_adc_with_a_wide_filter(hiplib, oracle, kind, f_bins, iq):
    """Integrator::receive with the ADC's DEFAULT (Gaussian) filter and others: SignalBlock::put spreads a return over the
    neighbouring fast-time (and frequency) cells; returns near the window's edge lose the part that falls outside."""
    from tests.oracle_lib import OracleScene
    sd, lp = scenes.bus_receive(n_tris=5000, n_paths=20000, t_bins=64)
    if f_bins > 1:
        sd.sensor.f_bins = f_bins
        c, lmin = sd.physics.c, sd.physics.lambda_min_nm
        sd.sensor.f_bandwidth = c / (lmin * 1e-9)
        lp.bins_y = f_bins
    if iq:
        lp.mode = capi.BF_MODE_RECEIVE_IQ
    sd.sensor.rfilter = _host_filter(kind).flatten(0)
    sd.finalize()
    ho, ro, so = OracleScene(sd).render(lp, records=True, threads=8)
    g = capi.Scene(sd)
    for flags in (capi.BF_FLAG_MEGAKERNEL, 0):
        lp.flags = flags
        hg, rg, sg = g.render(lp, records=True)
        _records_equal(rg, ro)
        assert sg.n_invalid == so.n_invalid
        _hist_close(hg, ho, lp.n_paths, float(np.abs(ro["L"]).max()))
    cells = hg.reshape(f_bins, 64, 3)
    assert (cells[:, :, 2] != 0).sum() > 8 and not np.array_equal(cells[:, :, 2], np.round(cells[:, :, 2]))


@pytest.mark.gpu
def test_batched_and_rolling_batched_renders_with_a_wide_filter(hiplib, oracle):
    """The filtered put re-derives a path's film position from ITS render's seed: batched launches (one launch sequence for
    several renders, per-render seeds) and rolling batches must agree with stand-alone oracle renders of those seeds."""
    import torch
    from tests.oracle_lib import OracleScene
    sd, lp = scenes.trans_rad(spp=3000)
    lp.mode, lp.bins, lp.bin_width = capi.BF_MODE_RANGE, 48, 0.25
    sd.sensor.rfilter = _host_filter("gaussian", stddev=0.8).flatten(32)
    sd.finalize()
    seeds = [11, 12, 13, 14]
    o = OracleScene(sd)
    want = []
    for sdd in seeds:
        lp.seed = sdd
        want.append(o.render(lp, records=True, threads=8))
    g = capi.Scene(sd)
    lp.seed = 0
    hist, rec, st = g.render_batch(lp, len(seeds), seeds=seeds, records=True)
    assert st.kernel_variant == capi.BF_VARIANT_WIDE
    for k in range(len(seeds)):
        _records_equal(rec[k], want[k][1])
        _hist_close(hist[k], want[k][0], lp.n_paths, float(np.abs(want[k][1]["L"]).max()))
    # the same four renders as two rolling batch calls of two
    dev = torch.zeros((4, hist.shape[1]), dtype=torch.float32, device="cuda")
    lp.flags = capi.BF_FLAG_ROLLING
    for c in range(2):
        g.render_batch_device(lp, 2, dev[2 * c].data_ptr(), seeds=seeds[2 * c: 2 * c + 2])
    g.flush()
    g.sync()
    got = dev.cpu().numpy()
    for k in range(len(seeds)):
        _hist_close(got[k], want[k][0], lp.n_paths, float(np.abs(want[k][1]["L"]).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_fuzz_filters_films_and_modes(hiplib, oracle, seed):
    """Random filter (kind, parameters, block size) x film (1 x 1 ... 13 x 9) x mode (path / range / time) x launch form (plain,
    one-kernel, global atomics, two path_offset shards) on the zoo scene: records per path and the filtered histograms against the
    oracle."""
    from tests.test_gpu_parity import _zoo_scene
    from tests.oracle_lib import OracleScene
    rng = np.random.default_rng(1000 + seed)
    kind = ["gaussian", "tent", "mitchell", "catmullrom", "lanczos", "box"][seed % 6]
    props = {"gaussian": {"stddev": float(rng.uniform(0.3, 1.4))}, "tent": {}, "mitchell": {"B": float(rng.uniform(0, 1)), "C": float(rng.uniform(0, 1))},
             "catmullrom": {}, "lanczos": {"lobes": int(rng.integers(1, 5))}, "box": {"radius": float(rng.uniform(0.55, 2.2))}}[kind]
    film = (int(rng.integers(1, 14)), int(rng.integers(1, 10))) if seed % 4 else (1, 1)
    mode = [capi.BF_MODE_RANGE, capi.BF_MODE_PATH, capi.BF_MODE_TIME][seed % 3]
    block = int(rng.choice([0, 1, 2, 5, 32]))
    sd, _ = _zoo_scene(two_emitters=bool(seed & 1))
    T = Transform4f
    sd.set_perspective(T.translate([0, 0, 0.3]) * T.rotate([1, 0, 0], 90) * T.rotate([0, 1, 0], 90), fov=float(rng.uniform(30, 80)), near_clip=0.1,
                       far_clip=100.0, film=film)
    sd.sensor.rfilter = _host_filter(kind, **props).flatten(block)
    sd.finalize()
    spp = int(rng.integers(40, 200)) if film != (1, 1) else 6000
    bins = int(rng.integers(8, 70))
    kw = dict(seed=int(rng.integers(1, 1 << 30)), bins=bins, bin_width=0.2 if mode == capi.BF_MODE_RANGE else 1e-9, color_mode=int(rng.integers(0, 2)))
    if film != (1, 1):
        kw.update(film=film, spp=spp)
    n = film[0] * film[1] * spp
    lp = capi.make_launch(mode, n, **kw)
    ho, ro, so = OracleScene(sd).render(lp, records=True, threads=8)
    g = capi.Scene(sd)
    amax = float(np.abs(ro["L"]).max())
    for flags in (0, capi.BF_FLAG_MEGAKERNEL, capi.BF_FLAG_GLOBAL_ATOMICS):
        lp.flags = flags
        hg, rg, sg = g.render(lp, records=True)
        _records_equal(rg, ro)
        assert sg.n_invalid == so.n_invalid and (sg.kernel_variant == capi.BF_VARIANT_WIDE or flags == capi.BF_FLAG_MEGAKERNEL)
        _hist_close(hg, ho, max(spp, 64) * 16, amax)
    lp.flags = 0
    half = n // 2 + 3
    parts = []
    for off, cnt in ((0, half), (half, n - half)):
        l2 = capi.make_launch(mode, cnt, path_offset=off, **kw)
        parts.append(g.render(l2)[0])
    _hist_close(parts[0] + parts[1], ho, max(spp, 64) * 16, amax)


WINDOW_XML = """<scene version='2.0.0'>
    <integrator type='pathtimefrequency'/>
    <shape type='rectangle'><receiver type='omnidirectional'>
        <adc type='hdradc'><integer name='t_bins' value='64'/><integer name='f_bins' value='8'/>%s<rfilter type='box'/></adc>
        <sampler type='independent'><integer name='sample_count' value='4'/></sampler></receiver></shape>
    <shape type='rectangle'><transform name='to_world'><translate z='3'/></transform>
        <transmitter type='areatransmitter'><spectrum name='radiance' value='1'/></transmitter></shape>
</scene>"""


def test_adc_window_properties_like_the_reference():
    """adc.cpp:26-38,80-91 (the fork's ADC; src/films/tests/test_hdrfilm.py:35-72 is the film's twin): window_{t,f}_bins and
    window_offset_{t,f}; a window that leaves the ADC is an error; the flattened sensor carries it and the launch names the window."""
    from beifong_amd import mitsuba
    mitsuba.set_variant("scalar_spectral")
    from beifong_amd.mitsuba.core.xml import load_string
    from beifong_amd.mitsuba._host import HostError
    win = "<integer name='window_t_bins' value='20'/><integer name='window_f_bins' value='3'/><integer name='window_offset_t' value='7'/><integer name='window_offset_f' value='2'/>"
    sc = load_string(WINDOW_XML % win)
    rx = sc.receivers()[0]
    d = sc.flat_desc(rx).desc.sensor
    assert (d.t_bins, d.f_bins) == (64, 8)
    assert (d.window_t_bins, d.window_f_bins, d.window_offset_t, d.window_offset_f) == (20, 3, 7, 2)
    lp = sc.integrator().launch_for(rx)
    assert (lp.bins, lp.bins_y) == (20, 3)
    d0 = load_string(WINDOW_XML % "")
    s0 = d0.flat_desc(d0.receivers()[0]).desc.sensor
    assert (s0.window_t_bins, s0.window_f_bins, s0.window_offset_t, s0.window_offset_f) == (0, 0, 0, 0)
    with pytest.raises(HostError, match="Invalid window specification"):
        load_string(WINDOW_XML % "<integer name='window_offset_t' value='60'/>")              # 60 + 64 > 64: the size does not adjust
    load_string(WINDOW_XML % "<integer name='window_offset_t' value='60'/><integer name='window_t_bins' value='4'/>")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["box", "gaussian"])
def test_adc_window(hiplib, oracle, kind):
    """receive() into an ADC window (integrator.cpp:624-628): the launch bins into the window, the time / frequency scaling stays
    the full ADC's.  Against the oracle; and, for the box filter, the window equals that part of the whole ADC's render."""
    from tests.oracle_lib import OracleScene
    sd, lp = scenes.bus_receive(n_tris=5000, n_paths=30000, t_bins=64)
    sd.sensor.f_bins = 8
    sd.sensor.f_bandwidth = sd.physics.c / (sd.physics.lambda_min_nm * 1e-9)
    if kind != "box":
        sd.sensor.rfilter = _host_filter(kind).flatten(0)
    sd.finalize()
    lp.bins_y = 8
    full = capi.Scene(sd).render(lp, records=True)
    ot, of, wt, wf = 5, 1, 40, 6
    sd.sensor.window_offset_t, sd.sensor.window_offset_f, sd.sensor.window_t_bins, sd.sensor.window_f_bins = ot, of, wt, wf
    sd.finalize()
    g = capi.Scene(sd)
    with pytest.raises(capi.BeifongError, match="window"):
        g.render(lp)                                            # the launch still names the whole ADC
    lp.bins, lp.bins_y = wt, wf
    ho, ro, so = OracleScene(sd).render(lp, records=True, threads=8)
    for flags in (0, capi.BF_FLAG_MEGAKERNEL):
        lp.flags = flags
        hg, rg, sg = g.render(lp, records=True)
        _records_equal(rg, ro)
        assert sg.n_invalid == so.n_invalid and sg.kernel_variant in (0, capi.BF_VARIANT_WIDE)       # a window off the origin: general kernels
        _hist_close(hg, ho, lp.n_paths, float(np.abs(ro["L"]).max()))
    _records_equal(rg, full[1])                                 # the window changes where samples land, not the paths
    crop = full[0].reshape(8, 64, 3)[of:of + wf, ot:ot + wt]
    assert np.allclose(hg.reshape(wf, wt, 3), crop, rtol=3e-5, atol=lp.n_paths * 2.0 ** -24 * 4 * max(1.0, float(np.abs(ro["L"]).max())))
    assert (crop[:, :, 2] != 0).sum() > 5
    sd.sensor.window_offset_t = 30                              # 30 + 40 > 64
    sd.finalize()
    with pytest.raises(capi.BeifongError, match="Invalid window"):
        capi.Scene(sd)


CROP_XML = """<scene version='2.0.0'><integrator type='path'/>
    <sensor type='perspective'><float name='fov' value='50'/>
        <film type='hdrfilm'><integer name='width' value='32'/><integer name='height' value='21'/>%s<rfilter type='box'/></film>
        <sampler type='independent'><integer name='sample_count' value='4'/></sampler></sensor>
    <shape type='rectangle'><emitter type='area'><spectrum name='radiance' value='1'/></emitter></shape></scene>"""


def test_film_crop_window_like_the_reference_test():
    """src/films/tests/test_hdrfilm.py:35-72 (test02_crops): size / crop_size / crop_offset; a crop window that leaves the film
    is an error because the crop size does not adjust.  And what the crop does to the camera (perspective_projection,
    sensor.h:196-231): sample (u, v) of the crop is the full film's sample (offset + (u, v) crop_size) / film_size."""
    from beifong_amd import mitsuba
    mitsuba.set_variant("scalar_rgb")
    from beifong_amd.mitsuba.core.xml import load_string
    from beifong_amd.mitsuba._host import HostError
    crop = ("<integer name='crop_width' value='11'/><integer name='crop_height' value='5'/><integer name='crop_offset_x' value='2'/>"
            "<integer name='crop_offset_y' value='3'/><boolean name='high_quality_edges' value='true'/>")
    sc = load_string(CROP_XML % crop)
    film = sc.sensors()[0].film()
    assert film.size() == (32, 21) and film.crop_size() == (11, 5) and film.crop_offset() == (2, 3)
    d = sc.flat_desc(sc.sensors()[0]).desc.sensor
    assert (d.film_width, d.film_height, d.crop_offset_x, d.crop_offset_y) == (11, 5, 2, 3)
    lp = sc.integrator().launch_for(sc.sensors()[0])
    assert (lp.film_width, lp.film_height, lp.spp, lp.n_paths) == (11, 5, 4, 11 * 5 * 4)
    incomplete = "<integer name='crop_offset_x' value='30'/><integer name='crop_offset_y' value='20'/>"
    with pytest.raises(HostError, match="Invalid crop window"):
        load_string(CROP_XML % incomplete)
    sc2 = load_string(CROP_XML % (incomplete + "<integer name='crop_width' value='2'/><integer name='crop_height' value='1'/>"))
    f2 = sc2.sensors()[0].film()
    assert f2.size() == (32, 21) and f2.crop_size() == (2, 1) and f2.crop_offset() == (30, 20)
    # the camera: crop sample -> the same near-plane point as the corresponding full-film sample
    full = load_string(CROP_XML % "")
    m_full = np.array(full.flat_desc(full.sensors()[0]).desc.sensor.sample_to_camera[:], dtype=np.float64).reshape(4, 4)
    m_crop = np.array(d.sample_to_camera[:], dtype=np.float64).reshape(4, 4)
    for u, v in ((0.0, 0.0), (0.3, 0.8), (1.0, 1.0)):
        a = m_crop @ np.array([u, v, 0, 1.0])
        b = m_full @ np.array([(2 + u * 11) / 32, (3 + v * 5) / 21, 0, 1.0])
        assert np.allclose(a[:3] / a[3], b[:3] / b[3], rtol=2e-6, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,block", [("box", 0), ("gaussian", 32), ("lanczos", 3)])
def test_film_crop_window(hiplib, oracle, kind, block):
    """A crop window of a 40 x 30 film through the perspective camera: per-path records and the (filtered) histogram against the
    oracle, both pipelines; with the box filter every pixel of the crop holds its spp samples."""
    from tests.test_gpu_parity import _zoo_scene
    from tests.oracle_lib import OracleScene
    film, crop, spp, bins = (40, 30), (7, 4, 17, 9), 40, 32
    sd, _ = _zoo_scene(two_emitters=True)
    T = Transform4f
    sd.set_perspective(T.translate([0, 0, 0.3]) * T.rotate([1, 0, 0], 90) * T.rotate([0, 1, 0], 90), fov=60.0, near_clip=0.1, far_clip=100.0,
                       film=film, crop=crop)
    if kind != "box":
        sd.sensor.rfilter = _host_filter(kind).flatten(block)
    sd.finalize()
    cw, ch = crop[2], crop[3]
    lp = capi.make_launch(capi.BF_MODE_RANGE, cw * ch * spp, seed=9, bins=bins, bin_width=0.2, color_mode=capi.BF_COLOR_RGB, film=(cw, ch), spp=spp)
    ho, ro, so = OracleScene(sd).render(lp, records=True, threads=8)
    g = capi.Scene(sd)
    for flags in (0, capi.BF_FLAG_MEGAKERNEL, capi.BF_FLAG_GLOBAL_ATOMICS):
        lp.flags = flags
        hg, rg, sg = g.render(lp, records=True)
        _records_equal(rg, ro)
        assert sg.n_invalid == so.n_invalid and not (sg.kernel_variant & capi.BF_VARIANT_LEAN)
        _hist_close(hg, ho, spp * 16, float(np.abs(ro["L"]).max()))
    img = hg.reshape(ch, cw, 5 + bins)
    if kind == "box":
        assert np.array_equal(img[:, :, 4], np.full((ch, cw), float(spp)))
    assert (img[:, :, 5:].sum(axis=2) != 0).mean() > 0.3
    # the crop shows what that part of the full film shows: the same camera rays, different random numbers
    sd.set_perspective(T.translate([0, 0, 0.3]) * T.rotate([1, 0, 0], 90) * T.rotate([0, 1, 0], 90), fov=60.0, near_clip=0.1, far_clip=100.0,
                       film=film)
    sd.sensor.rfilter = capi.bf_rfilter()
    sd.finalize()
    lf = capi.make_launch(capi.BF_MODE_RANGE, film[0] * film[1] * spp, seed=10, bins=bins, bin_width=0.2, color_mode=capi.BF_COLOR_RGB, film=film, spp=spp)
    full = capi.Scene(sd).render(lf)[0].reshape(film[1], film[0], 5 + bins)[crop[1]:crop[1] + ch, crop[0]:crop[0] + cw]
    if kind == "box":
        hit_c, hit_f = img[:, :, 3] / spp, full[:, :, 3] / spp                  # alpha: fraction of the pixel's rays that hit something
        assert np.abs(hit_c - hit_f).mean() < 0.08 and np.corrcoef(hit_c.ravel(), hit_f.ravel())[0, 1] > 0.9
