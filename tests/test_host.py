"""Host layer (beifong_amd/host): the Mitsuba-shaped plugin surface, the XML
loader and the flattening into the C ABI.  CPU-only: results are checked by
feeding the flattened bf_scene_desc to the oracle."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

from beifong_amd import capi, meshgen, scenes
from tests.oracle_lib import OracleScene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "beifong_amd", "host")
REF_XML = "/root/reference/python_scripts/trans_rad.xml"


@pytest.fixture(scope="module")
def mitsuba():
    if not os.path.exists(os.path.join(HOST, "libbeifong_host.so")):
        import __graft_entry__ as g
        g.build()
    from beifong_amd import mitsuba as m
    m.set_variant("scalar_rgb")
    return m


TRANS_RAD_LIKE = """
<scene version="2.1.0">
    <default name="spp" value="100"/>
    <integrator type="time"><integrator type="pathtime"/></integrator>
    <shape type="rectangle" id="rx">
        <transform name="to_world"><scale x="0.05" y="0.05"/><lookat origin="0, 0, 0" target="0, -1, 0" up="0, 0, 1"/></transform>
        <sensor type="fluxmeter">
            <film type="hdrfilm"><integer name="width" value="1"/><integer name="height" value="1"/><rfilter type="box"/></film>
            <sampler type="independent"><integer name="sample_count" value="$spp"/></sampler>
        </sensor>
    </shape>
    <bsdf type="twosided" id="material"><bsdf type="diffuse"><spectrum value="1.0" name="reflectance"/></bsdf></bsdf>
    <emitter type="spot">
        <spectrum value="1.0" name="intensity"/><float name="cutoff_angle" value="25"/><float name="beam_width" value="20"/>
        <transform name="to_world"><lookat origin="0, 0, 0" target="0, -1, 0" up="0, 0, 1"/></transform>
    </emitter>
    <shape type="rectangle" id="target">
        <transform name="to_world"><scale x="1" y="1"/><lookat origin="0, -4, 0" target="0, 0, 0" up="0, 0, 1"/></transform>
        <ref id="material" name="bsdf"/>
    </shape>
    <shape type="rectangle" id="gnd">
        <transform name="to_world"><scale x="20" y="20"/><lookat origin="0, 0, -0.5" target="0, 0, 0.5"/></transform>
        <ref id="material" name="bsdf"/>
    </shape>
</scene>
"""


def _oracle_on_host_scene(scene, endpoint):
    lp = scene.integrator().launch_for(endpoint)
    h, rec, st = OracleScene(scene.flat_desc(endpoint)).render(lp, records=True)
    return lp, h, rec


def test_every_plugin_exports_the_reference_plugin_abi(mitsuba):
    # include/mitsuba/core/class.h:195-211: extern "C" plugin_name() / plugin_descr()
    sos = sorted(glob.glob(os.path.join(HOST, "plugins", "*.so")))
    names = set()
    for so in sos:
        lib = C.CDLL(so)
        lib.plugin_name.restype = C.c_char_p
        lib.plugin_descr.restype = C.c_char_p
        n = lib.plugin_name().decode()
        assert n == os.path.basename(so)[:-3]
        assert len(lib.plugin_descr()) > 0
        names.add(n)
    for need in ("path", "pathlength", "range", "pathtime", "time", "pathtimefrequency", "rectangle", "obj", "ply", "diffuse",
                 "twosided", "roughconductor", "spot", "point", "area", "areatransmitter", "wignertransmitter", "fluxmeter",
                 "irradiancemeter", "radiancemeter", "perspective", "omnidirectional", "wignerreceiver", "hdrfilm", "hdradc", "box", "independent", "phase",
                 "phasedtransmitter", "phasedreceiver"):
        assert need in names, need


@pytest.mark.skipif(not os.path.exists(REF_XML), reason="reference tree not present (GPU box)")
def test_reference_trans_rad_xml_loads_unchanged(mitsuba):
    """python_scripts/trans_rad.xml, -Dspp=16 (BASELINE configs[0]): the loaded
    scene flattens to the same description as the hand-built one."""
    from beifong_amd.mitsuba.core.xml import load_file
    scene = load_file(REF_XML, spp=16)
    sen = scene.sensors()[0]
    assert sen.sampler().sample_count() == 16
    lp, h, rec = _oracle_on_host_scene(scene, sen)
    assert (lp.mode, lp.bins, lp.n_paths) == (capi.BF_MODE_TIME, 50, 16)
    sd, lp2 = scenes.trans_rad(16)
    h2, rec2, _ = OracleScene(sd).render(lp2, records=True)
    assert np.array_equal(h, h2) and np.array_equal(rec["L"], rec2["L"])
    assert h.shape == (5 + 150,)            # float32[1,1,155]


def test_xml_transform_composition_and_defaults(mitsuba):
    from beifong_amd.mitsuba.core.xml import load_string
    scene = load_string(TRANS_RAD_LIKE, spp=2000)
    sen = scene.sensors()[0]
    assert sen.sampler().sample_count() == 2000           # caller overrides <default>
    assert [round(s.surface_area(), 4) for s in scene.shapes()] == [0.01, 4.0, 1600.0]
    lp, h, rec = _oracle_on_host_scene(scene, sen)
    sd, lp2 = scenes.trans_rad(2000)
    h2, _, _ = OracleScene(sd).render(lp2)
    assert np.array_equal(h, h2)
    scene = load_string(TRANS_RAD_LIKE)
    assert scene.sensors()[0].sampler().sample_count() == 100


def test_xml_errors_are_reported_like_the_reference(mitsuba):
    from beifong_amd.mitsuba._host import HostError
    from beifong_amd.mitsuba.core.xml import load_string
    with pytest.raises(HostError, match="unreferenced property"):
        load_string(TRANS_RAD_LIKE.replace('<float name="cutoff_angle" value="25"/>',
                                           '<float name="cutoff_angle" value="25"/><float name="bogus" value="1"/>'))
    with pytest.raises(HostError, match="not found"):
        load_string(TRANS_RAD_LIKE.replace('type="spot"', 'type="nosuchplugin"'))
    with pytest.raises(HostError, match="Type mismatch"):
        load_string(TRANS_RAD_LIKE.replace('<emitter type="spot">', '<emitter type="diffuse">').replace(
            '<spectrum value="1.0" name="intensity"/><float name="cutoff_angle" value="25"/><float name="beam_width" value="20"/>', "")
            .replace('<transform name="to_world"><lookat origin="0, 0, 0" target="0, -1, 0" up="0, 0, 1"/></transform>\n    </emitter>', "</emitter>"))
    with pytest.raises(HostError, match="undefined parameter"):
        load_string(TRANS_RAD_LIKE.replace('<default name="spp" value="100"/>', ""))
    with pytest.raises(HostError, match="1x1"):
        load_string(TRANS_RAD_LIKE.replace('name="width" value="1"', 'name="width" value="4"'))


def _write_obj(path, v, f, with_normals=False, texcoords=None):
    with open(path, "w") as fh:
        for p in v:
            fh.write("v %r %r %r\n" % tuple(float(x) for x in p))
        if texcoords is not None:
            for uv in texcoords:
                fh.write("vt %r %r\n" % (float(uv[0]), float(uv[1])))
            for t in f:
                fh.write("f %d/%d %d/%d %d/%d\n" % tuple(int(i) + 1 for i in t for _ in (0, 1)))
            return
        if with_normals:
            fh.write("vn 0 0 1\n")
        for t in f:
            if with_normals:
                fh.write("f %d//1 %d//1 %d//1\n" % tuple(int(i) + 1 for i in t))
            else:
                fh.write("f %d %d %d\n" % tuple(int(i) + 1 for i in t))


MESH_SCENE = """
<scene version="2.1.0">
    <integrator type="range"><integrator type="pathlength"/><float name="dr" value="0.1"/><integer name="bins" value="64"/></integrator>
    <sensor type="perspective">
        <float name="fov" value="45"/><float name="near_clip" value="0.1"/><float name="far_clip" value="100"/>
        <transform name="to_world"><lookat origin="0, 0, -3" target="0, 0, 0" up="0, 1, 0"/></transform>
        <film type="hdrfilm"><integer name="width" value="1"/><integer name="height" value="1"/><rfilter type="box"/></film>
        <sampler type="independent"><integer name="sample_count" value="3000"/></sampler>
    </sensor>
    <shape type="rectangle">
        <transform name="to_world"><scale x="0.2" y="0.2"/><lookat origin="0.5, 0.5, -3" target="0, 0, 0" up="0, 1, 0"/></transform>
        <emitter type="area"><spectrum name="radiance" value="100"/></emitter>
    </shape>
    <shape type="%s">
        <string name="filename" value="%s"/>%s
        <bsdf type="twosided"><bsdf type="roughconductor"><float name="alpha" value="0.2"/></bsdf></bsdf>
    </shape>
</scene>
"""


def test_obj_loader_known_answers_and_normals(mitsuba, tmp_path):
    """src/librender/tests/test_mesh.py:257-298 on a synthesised rectangle.obj,
    through the obj plugin (to_world at load, fan triangulation, vertex de-dup,
    normals recomputed when the file has none: obj.cpp:339-344)."""
    from beifong_amd.mitsuba.core.xml import load_string
    v, f = meshgen.rectangle_obj()
    _write_obj(tmp_path / "rectangle.obj", v, f)
    scene = load_string(MESH_SCENE % ("obj", "rectangle.obj", ""), base_dir=str(tmp_path))
    shape = scene.shapes()[1]
    assert shape.primitive_count() == 2 and np.isclose(shape.surface_area(), 4.0)
    desc = scene.flat_desc(scene.sensors()[0])
    sh = desc.desc.shapes[1]
    assert sh.n_vertices == 4 and sh.n_faces == 2 and bool(sh.normals)
    nrm = np.ctypeslib.as_array(sh.normals, shape=(4, 3))
    assert np.allclose(nrm, [[0, 0, 1]] * 4, atol=1e-6)            # recomputed vertex normals of a flat quad
    o = OracleScene(desc)
    eps = np.float32(1500 * 2.0 ** -24)
    r = o.intersect_full([-0.3, -0.3, -10, eps, 0, 0, 1, np.inf])
    t, prim, _, uv = o.trace_closest([[-0.3, -0.3, -10, eps, 0, 0, 1, np.inf]])
    assert np.isclose(r["t"], 10) and np.allclose(r["prim_uv"], [0.35, 0.3], atol=1e-6)
    assert prim[0] == 1                  # global primitive index: the emitter rectangle is primitive 0
    # texture coordinates (flip_tex_coords defaults to true: obj.cpp:99,199) reach the boundary and give the tangents
    # test_mesh.py:284-285,296-297 expect
    uv = (v[:, :2] + 1) / 2
    _write_obj(tmp_path / "rectangle_uv.obj", v, f, texcoords=np.stack([uv[:, 0], 1 - uv[:, 1]], 1))
    scene = load_string(MESH_SCENE % ("obj", "rectangle_uv.obj", ""), base_dir=str(tmp_path))
    desc = scene.flat_desc(scene.sensors()[0])
    sh = desc.desc.shapes[1]
    assert bool(sh.texcoords) and sh.n_vertices == 4
    pos = np.ctypeslib.as_array(sh.positions, shape=(4, 3))
    assert np.allclose(np.ctypeslib.as_array(sh.texcoords, shape=(4, 2)), (pos[:, :2] + 1) / 2, atol=1e-6)
    r = OracleScene(desc).intersect_full([-0.3, -0.3, -10, eps, 0, 0, 1, np.inf])
    assert np.allclose(r["dp_du"], [2, 0, 0], atol=1e-6) and np.allclose(r["dp_dv"], [0, 2, 0], atol=1e-6)
    # quad face + face_normals=true: one polygon fan-triangulated, no normals kept
    with open(tmp_path / "quad.obj", "w") as fh:
        fh.write("v -1 -1 0\nv 1 -1 0\nv 1 1 0\nv -1 1 0\nvn 0 0 1\nf 1//1 2//1 3//1 4//1\n")
    scene = load_string(MESH_SCENE % ("obj", "quad.obj", '<boolean name="face_normals" value="true"/>'), base_dir=str(tmp_path))
    sh = scene.flat_desc(scene.sensors()[0]).desc.shapes[1]
    assert sh.n_faces == 2 and sh.n_vertices == 4 and not bool(sh.normals)
    # to_world is applied at load time
    scene = load_string(MESH_SCENE % ("obj", "rectangle.obj", '<transform name="to_world"><scale value="2"/><translate x="1"/></transform>'),
                        base_dir=str(tmp_path))
    sh = scene.flat_desc(scene.sensors()[0]).desc.shapes[1]
    pos = np.ctypeslib.as_array(sh.positions, shape=(4, 3))
    assert np.allclose(sorted(pos[:, 0]), [-1, -1, 3, 3])


def test_ply_loader_on_the_reference_fixtures(mitsuba):
    """tests/golden/triangle*.ply are the reference's own data files
    (src/librender/tests/data/): ASCII, and binary LE with normals + extra face
    properties."""
    from beifong_amd.mitsuba.core.xml import load_string
    gold = os.path.join(ROOT, "tests", "golden")
    for name, has_n in (("triangle.ply", False), ("triangle_face_colors.ply", True)):
        scene = load_string(MESH_SCENE % ("ply", name, ""), base_dir=gold)
        sh = scene.flat_desc(scene.sensors()[0]).desc.shapes[1]
        assert sh.n_vertices == 3 and sh.n_faces == 1
        pos = np.ctypeslib.as_array(sh.positions, shape=(3, 3))
        assert np.allclose(pos, [[0, 0, 0], [0, 0, 1], [0, 1, 0]])
        nrm = np.ctypeslib.as_array(sh.normals, shape=(3, 3))
        assert np.allclose(np.abs(nrm), [[1, 0, 0]] * 3, atol=1e-6)
        assert np.isclose(scene.shapes()[1].surface_area(), 0.5)


def test_ply_binary_big_endian_and_ascii_agree(mitsuba, tmp_path):
    from beifong_amd.mitsuba.core.xml import load_string
    v, f = meshgen.triangle_soup(50, seed=3)
    hdr = "ply\nformat %s 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\n" \
          "element face %d\nproperty list uchar int vertex_indices\nend_header\n"
    with open(tmp_path / "a.ply", "w") as fh:
        fh.write(hdr % ("ascii", len(v), len(f)))
        for p in v:
            fh.write("%r %r %r 7\n" % tuple(float(x) for x in p))
        for t in f:
            fh.write("3 %d %d %d\n" % tuple(int(i) for i in t))
    for fmt, end in (("binary_big_endian", ">"), ("binary_little_endian", "<")):
        with open(tmp_path / (fmt + ".ply"), "wb") as fh:
            fh.write((hdr % (fmt, len(v), len(f))).encode())
            for p in v:
                fh.write(p.astype(end + "f4").tobytes() + b"\x07")
            for t in f:
                fh.write(b"\x03" + t.astype(end + "i4").tobytes())
    out = []
    for name in ("a.ply", "binary_big_endian.ply", "binary_little_endian.ply"):
        scene = load_string(MESH_SCENE % ("ply", name, '<boolean name="face_normals" value="true"/>'), base_dir=str(tmp_path))
        sh = scene.flat_desc(scene.sensors()[0]).desc.shapes[1]
        out.append((np.ctypeslib.as_array(sh.positions, shape=(sh.n_vertices, 3)).copy(),
                    np.ctypeslib.as_array(sh.indices, shape=(sh.n_faces, 3)).copy()))
    for p, i in out[1:]:
        assert np.array_equal(p, out[0][0]) and np.array_equal(i, out[0][1])
    assert np.array_equal(out[0][0], v) and np.array_equal(out[0][1], f)


def test_mesh_scene_flattens_like_the_python_builder(mitsuba, tmp_path):
    from beifong_amd.mitsuba.core.xml import load_string
    v, f = meshgen.triangle_soup(300, seed=5, extent=0.8, size=0.3)
    _write_obj(tmp_path / "soup.obj", v, f)
    scene = load_string(MESH_SCENE % ("obj", "soup.obj", '<boolean name="face_normals" value="true"/>'), base_dir=str(tmp_path))
    lp, h, rec = _oracle_on_host_scene(scene, scene.sensors()[0])
    assert lp.mode == capi.BF_MODE_RANGE and lp.bins == 64 and h.shape == (69,)
    assert h[4] == 3000 and h[3] > 0 and np.abs(h[5:]).sum() > 0


MESH_SCENE_RECT = """
<scene version="2.1.0">
    <integrator type="range"><integrator type="pathlength"/><float name="dr" value="0.1"/><integer name="bins" value="64"/></integrator>
    <sensor type="perspective">
        <float name="fov" value="45"/><float name="near_clip" value="0.1"/><float name="far_clip" value="100"/>
        <transform name="to_world"><lookat origin="0, 0, -3" target="0, 0, 0" up="0, 1, 0"/></transform>
        <film type="hdrfilm"><integer name="width" value="1"/><integer name="height" value="1"/><rfilter type="box"/></film>
        <sampler type="independent"><integer name="sample_count" value="100"/></sampler>
    </sensor>
    <shape type="rectangle">
        <transform name="to_world"><scale x="0.2" y="0.2"/><lookat origin="0.5, 0.5, -3" target="0, 0, 0" up="0, 1, 0"/></transform>
        <emitter type="area"><spectrum name="radiance" value="100"/></emitter>
    </shape>
    <shape type="rectangle"><bsdf type="diffuse"/></shape>
</scene>
"""

RECEIVE_SCENE = """
<scene version="2.1.0">
    <integrator type="pathtimefrequency"/>
    <shape type="rectangle">
        <transform name="to_world"><scale x="0.02" y="0.05"/><lookat origin="0, 0, 0.3" target="1, 0, 0.3" up="0, 0, 1"/></transform>
        <transmitter type="wignertransmitter">
            <string name="signaltype" value="pulse"/><float name="amplitude" value="1"/>
            <float name="pulse_len" value="0.000588235"/><float name="prf" value="6.640625"/>
            <float name="freq_centre" value="39375"/><float name="freq_ext" value="1700"/>
        </transmitter>
    </shape>
    <shape type="rectangle">
        <transform name="to_world"><scale x="0.02" y="0.05"/><lookat origin="0, 0, 0.3" target="1, 0, 0.3" up="0, 0, 1"/></transform>
        <receiver type="omnidirectional">
            <float name="adc_sampling_start" value="0"/><float name="adc_sampling_end" value="0.150588"/>
            <adc type="hdradc"><integer name="t_bins" value="256"/><integer name="f_bins" value="1"/>
                <float name="t_bandwidth" value="0.150588"/><float name="f_bandwidth" value="90000"/><rfilter type="box"/></adc>
            <sampler type="independent"><integer name="sample_count" value="4000"/></sampler>
        </receiver>
    </shape>
    <shape type="rectangle">
        <transform name="to_world"><scale x="20" y="20"/></transform>
        <bsdf type="twosided"><bsdf type="diffuse"><spectrum name="reflectance" value="0.5"/></bsdf></bsdf>
    </shape>
</scene>
"""


def test_receive_scene_with_fork_plugins(mitsuba):
    """<transmitter>, <receiver>, <adc> tags (class aliases of the fork's base
    classes: transmitter.cpp:10, receiver.cpp:205, adc.cpp:106)."""
    from beifong_amd.mitsuba.core.xml import load_string
    scene = load_string(RECEIVE_SCENE)
    assert len(scene.receivers()) == 1 and len(scene.sensors()) == 0
    rx = scene.receivers()[0]
    lp, h, rec = _oracle_on_host_scene(scene, rx)
    assert (lp.mode, lp.bins, lp.bins_y, lp.n_paths) == (capi.BF_MODE_RECEIVE_RAW, 256, 1, 4000)
    h = h.reshape(1, 256, 3)
    assert h[0, :, 2].sum() == 4000 and np.abs(h[0, :, 0]).sum() > 0
    d = scene.flat_desc(rx).desc
    assert d.n_emitters == 1 and d.emitters[0].type == capi.BF_TRANSMITTER_WIGNER and d.emitters[0].signal_type == capi.BF_SIGNAL_PULSE
    assert d.shapes[0].emitter == 0 and d.sensor.type == capi.BF_RECEIVER_OMNI and d.sensor.shape == 1


def test_receive_type_mix_resample_and_doppler_property(mitsuba):
    """receiver.cpp:21 receive_type: "mix_resample" (integrator.cpp:1588-1603) -> BF_FLAG_MIX_RESAMPLE on the omnidirectional
    receiver; the integrator's "doppler" switch (not a reference property; the reference carries the calls commented out)
    -> BF_FLAG_DOPPLER; "mixer" (an empty branch in the reference) is refused; mix_resample on the Wigner receiver reads the
    receiver's own local-oscillator properties (delta signals only)."""
    from beifong_amd.mitsuba.core.xml import load_string
    from beifong_amd.mitsuba._host import HostError
    mix = RECEIVE_SCENE.replace('<receiver type="omnidirectional">',
                                '<receiver type="omnidirectional"><string name="receive_type" value="mix_resample"/>')
    scene = load_string(mix)
    rx = scene.receivers()[0]
    lp, h, _ = _oracle_on_host_scene(scene, rx)
    assert lp.flags == capi.BF_FLAG_MIX_RESAMPLE and not h.any()          # HEAD: beat 0, every sample outside the ADC
    dop = mix.replace('<integrator type="pathtimefrequency"/>',
                      '<integrator type="pathtimefrequency"><boolean name="doppler" value="true"/></integrator>')
    scene = load_string(dop)
    rx = scene.receivers()[0]
    lp, h, _ = _oracle_on_host_scene(scene, rx)
    assert lp.flags == capi.BF_FLAG_MIX_RESAMPLE | capi.BF_FLAG_DOPPLER
    assert h.reshape(1, 256, 3)[0, :, 2].sum() > 0                         # identity velocity: the ground's returns beat
    # the AOV wrapper forwards the switch of its nested integrator
    scene = load_string(dop.replace('<integrator type="pathtimefrequency"><boolean name="doppler" value="true"/></integrator>',
                                    '<integrator type="phase"><integer name="bins" value="4"/><integrator type="pathtimefrequency">'
                                    '<boolean name="doppler" value="true"/></integrator></integrator>'))
    assert scene.integrator().launch_for(scene.receivers()[0]).flags == capi.BF_FLAG_MIX_RESAMPLE | capi.BF_FLAG_DOPPLER
    scene = load_string(RECEIVE_SCENE.replace('<receiver type="omnidirectional">',
                                              '<receiver type="omnidirectional"><string name="receive_type" value="mixer"/>'))
    with pytest.raises(HostError, match="mixer"):
        scene.integrator().launch_for(scene.receivers()[0])
    # the Wigner receiver under mix_resample has a local oscillator of its own: delta signals "linfmcw" / "cw" (wignerreceiver.cpp:72-110)
    wig = RECEIVE_SCENE.replace('<receiver type="omnidirectional">',
                                '<receiver type="wignerreceiver"><string name="receive_type" value="mix_resample"/>'
                                '<string name="signaltype" value="linfmcw"/><float name="chirp_len" value="0.150588"/>'
                                '<float name="crf" value="6.640625"/><float name="freq_centre" value="39375"/><float name="freq_sweep" value="1700"/>')
    scene = load_string(wig)
    rx = scene.receivers()[0]
    d = scene.flat_desc(rx).desc
    assert d.sensor.type == capi.BF_RECEIVER_WIGNER and d.sensor.rx_signal_type == capi.BF_SIGNAL_LINFMCW and d.sensor.rx_sig_is_delta == 1
    assert abs(d.sensor.rx_pulse_len - 0.150588) < 1e-8 and abs(d.sensor.rx_prf - 6.640625) < 1e-6 and d.sensor.freq_ext == 1700.0
    assert scene.integrator().launch_for(rx).flags == capi.BF_FLAG_MIX_RESAMPLE
    # "pulse" is no delta by default: a uniform frequency weighted with the receiver's eval_signal; as a delta it is refused
    pul = load_string(RECEIVE_SCENE.replace('<receiver type="omnidirectional">',
                                            '<receiver type="wignerreceiver"><string name="receive_type" value="mix_resample"/>'
                                            '<string name="signaltype" value="pulse"/><float name="amplitude" value="2"/>'
                                            '<float name="pulse_len" value="0.01"/><float name="prf" value="6.640625"/>'
                                            '<float name="freq_centre" value="39375"/><float name="freq_ext" value="1700"/>'))
    ds = pul.flat_desc(pul.receivers()[0]).desc.sensor
    assert (ds.rx_signal_type, ds.rx_sig_is_delta, ds.rx_amplitude) == (capi.BF_SIGNAL_PULSE, 0, 2.0) and abs(ds.rx_pulse_len - 0.01) < 1e-9
    with pytest.raises(HostError, match="delta"):
        load_string(RECEIVE_SCENE.replace('<receiver type="omnidirectional">',
                                          '<receiver type="wignerreceiver"><string name="receive_type" value="mix_resample"/>'
                                          '<string name="signaltype" value="pulse"/><boolean name="sig_is_delta" value="true"/>'))


def test_resample_freq_property_of_the_transmitters(mitsuba):
    """wignertransmitter.cpp:431 / :211-221 `resample_freq`: accepted for "linfmcw" and "cw" (sample_delta_frequency defines the
    frequency for those), flattened into bf_emitter.resample_freq, and the oracle then bins every return of a de-chirping
    receiver ("mix_resample") at the beat frequency; "pulse" reads an uninitialised frequency in the reference: refused."""
    from beifong_amd.mitsuba.core.xml import load_string
    from beifong_amd.mitsuba._host import HostError
    fmcw = RECEIVE_SCENE.replace('<string name="signaltype" value="pulse"/><float name="amplitude" value="1"/>\n'
                                 '            <float name="pulse_len" value="0.000588235"/><float name="prf" value="6.640625"/>\n'
                                 '            <float name="freq_centre" value="39375"/><float name="freq_ext" value="1700"/>',
                                 '<string name="signaltype" value="linfmcw"/><float name="amplitude" value="1"/>'
                                 '<float name="chirp_len" value="0.150588"/><float name="crf" value="6.640625"/>'
                                 '<float name="freq_centre" value="39375"/><float name="freq_sweep" value="1700"/>'
                                 '<boolean name="resample_freq" value="true"/>')
    assert 'resample_freq' in fmcw
    fmcw = fmcw.replace('<receiver type="omnidirectional">',
                        '<receiver type="omnidirectional"><string name="receive_type" value="mix_resample"/>')
    scene = load_string(fmcw)
    rx = scene.receivers()[0]
    d = scene.flat_desc(rx).desc
    assert d.emitters[0].resample_freq == 1 and d.emitters[0].signal_type == capi.BF_SIGNAL_LINFMCW
    lp, h, _ = _oracle_on_host_scene(scene, rx)
    assert lp.flags == capi.BF_FLAG_MIX_RESAMPLE
    assert h.reshape(1, 256, 3)[0, :, 2].sum() > 0          # beats inside the ADC's 90 kHz (without re-sampling: none, see above)
    with pytest.raises(HostError, match="pulse"):
        load_string(RECEIVE_SCENE.replace('<string name="signaltype" value="pulse"/>',
                                          '<string name="signaltype" value="pulse"/><boolean name="resample_freq" value="true"/>'))


def test_phase_integrator_plugin_and_nested_depths(mitsuba):
    """phase.cpp (built at HEAD) wraps pathtimefrequency: `bins` S{k}.Y channels after Y, A, W; the
    MonteCarloIntegrator parameters are those of the NESTED integrator (integrator.cpp:1713-1728)."""
    from beifong_amd.mitsuba.core.xml import load_string
    xml = RECEIVE_SCENE.replace('<integrator type="pathtimefrequency"/>',
                                '<integrator type="phase"><integer name="bins" value="12"/>'
                                '<integrator type="pathtimefrequency"><integer name="max_depth" value="3"/>'
                                '<integer name="rr_depth" value="2"/></integrator></integrator>')
    scene = load_string(xml)
    rx = scene.receivers()[0]
    lp, h, rec = _oracle_on_host_scene(scene, rx)
    assert (lp.mode, lp.phase_bins, lp.max_depth, lp.rr_depth) == (capi.BF_MODE_RECEIVE_RAW, 12, 3, 2)
    assert h.shape == (256 * (3 + 12),)
    assert "phase" in {os.path.basename(p)[:-3] for p in glob.glob(os.path.join(HOST, "plugins", "*.so"))}
    # same for the gen-2 wrapper
    scene = load_string(MESH_SCENE_RECT.replace('<integrator type="pathlength"/>',
                                                '<integrator type="pathlength"><integer name="max_depth" value="2"/></integrator>'))
    lp = scene.integrator().launch_for(scene.sensors()[0])
    assert (lp.mode, lp.max_depth) == (capi.BF_MODE_RANGE, 2)


PHASED_SCENE = """
<scene version="2.1.0">
    <integrator type="pathtimefrequency"/>
    <shape type="rectangle">
        <transform name="to_world"><scale x="0.05" y="0.025"/><lookat origin="0, 0, 0.3" target="1, 0, 0.3" up="0, 0, 1"/></transform>
        <transmitter type="phasedtransmitter">
            <string name="signaltype" value="pulse"/><float name="amplitude" value="1"/>
            <float name="pulse_len" value="0.000588235"/><float name="prf" value="26.5625"/>
            <float name="freq_centre" value="39375"/><float name="freq_ext" value="1700"/>
            <integer name="n_elems" value="4"/>
            <vector name="steering_vector" x="0.2" y="0" z="0"/>
            <transform name="array_loc"><lookat origin="0, 0, 0.3" target="1, 0, 0.3" up="0, 0, 1"/></transform>
            <vector name="elem_dims" x="0.02" y="0.05" z="1"/>
            <vector name="elem_spacing" x="0.025" y="0" z="0"/>
            <vector name="elem_axis" x="1" y="0" z="0"/>
        </transmitter>
    </shape>
    <shape type="rectangle">
        <transform name="to_world"><scale x="0.05" y="0.025"/><lookat origin="0, 0, 0.3" target="1, 0, 0.3" up="0, 0, 1"/></transform>
        <receiver type="phasedreceiver">
            <float name="adc_sampling_start" value="0"/><float name="adc_sampling_end" value="0.037647"/>
            <float name="freq_centre" value="39375"/><float name="freq_ext" value="10000"/>
            <integer name="n_elems" value="3"/>
            <transform name="array_loc"><lookat origin="0, 0, 0.3" target="1, 0, 0.3" up="0, 0, 1"/></transform>
            <vector name="elem_dims" x="0.02" y="0.05" z="1"/>
            <vector name="elem_spacing" x="0.03" y="0" z="0"/>
            <vector name="elem_axis" x="1" y="0" z="0"/>
            <adc type="hdradc"><integer name="t_bins" value="64"/><integer name="f_bins" value="1"/>
                <float name="t_bandwidth" value="0.037647"/><float name="f_bandwidth" value="90000"/><rfilter type="box"/></adc>
            <sampler type="independent"><integer name="sample_count" value="4000"/></sampler>
        </receiver>
    </shape>
    <shape type="rectangle">
        <transform name="to_world"><scale x="20" y="20"/></transform>
        <bsdf type="twosided"><bsdf type="diffuse"><spectrum name="reflectance" value="0.5"/></bsdf></bsdf>
    </shape>
</scene>
"""


def test_phased_array_plugins_build_the_reference_element_tables(mitsuba):
    """phasedtransmitter / phasedreceiver (src/transmitters, src/receivers of the fork): the constructors'
    n_elems^2 virtual-element tables (phasedtransmitter.cpp:108-165) from the C++ plugins agree with the numpy builder
    of beifong_amd/scenedesc.py, and the flattened scene renders through the oracle."""
    from beifong_amd.mitsuba.core.xml import load_string
    from beifong_amd.scenedesc import SceneDesc, Transform4f as T
    scene = load_string(PHASED_SCENE)
    rx = scene.receivers()[0]
    d = scene.flat_desc(rx).desc
    assert d.emitters[0].type == capi.BF_TRANSMITTER_PHASED and d.sensor.type == capi.BF_RECEIVER_PHASED
    assert d.emitters[0].array.n_velems == 16 and d.sensor.array.n_velems == 9
    tab_tx = np.ctypeslib.as_array(d.emitters[0].array.velems, shape=(16, capi.BF_VELEM_FLOATS)).copy()
    tab_rx = np.ctypeslib.as_array(d.sensor.array.velems, shape=(9, capi.BF_VELEM_FLOATS)).copy()
    sd = SceneDesc()
    pose = T.look_at([0, 0, 0.3], [1, 0, 0.3], [0, 0, 1])
    a_tx = sd.phased_array(4, [0.02, 0.05, 1], [0.025, 0, 0], [1, 0, 0], steering_vector=[0.2, 0, 0], array_loc=pose)
    a_rx = sd.phased_array(3, [0.02, 0.05, 1], [0.03, 0, 0], [1, 0, 0], array_loc=pose)
    ref_tx = np.ctypeslib.as_array(a_tx.velems, shape=(16, capi.BF_VELEM_FLOATS))
    ref_rx = np.ctypeslib.as_array(a_rx.velems, shape=(9, capi.BF_VELEM_FLOATS))
    # the steering phasor's argument is K * (r' . sin(steer)) with K ~ 9e2: float32 in one, float64 in the other
    assert np.allclose(tab_tx[:, :28], ref_tx[:, :28], rtol=1e-5, atol=1e-6)
    assert np.allclose(tab_tx[:, 28:30], ref_tx[:, 28:30], atol=2e-4)
    assert np.allclose(tab_rx, ref_rx, rtol=1e-5, atol=1e-6)
    assert np.allclose(tab_rx[:, 28], 1.0) and np.allclose(tab_rx[:, 29], 0.0)          # no steering: psi' = 1
    assert list(d.emitters[0].array.elem_dims) == [np.float32(0.02), np.float32(0.05), 1.0]
    lp, h, rec = _oracle_on_host_scene(scene, rx)
    h = h.reshape(64, 3)
    assert h[:, 2].sum() == 4000 and np.all(np.isfinite(h)) and np.count_nonzero(h[:, 0]) > 3


def test_exr_writer_round_trip(mitsuba, tmp_path):
    """Film / ADC develop(): multi-channel float32 OpenEXR (hdrfilm.cpp:213-249, hdradc.cpp:259-295), uncompressed.
    Channels are stored alphabetically, as the format requires."""
    from beifong_amd.mitsuba import _host
    rng = np.random.default_rng(3)
    a = rng.standard_normal((3, 5, 6)).astype(np.float32)
    names = ["Y", "A", "W", "S0.Y", "S1.Y", "S10.Y"]
    path = str(tmp_path / "adc.exr")
    _host.write_exr(path, a, names)
    back, order = _host.read_exr(path)
    assert order == sorted(names) and back.shape == a.shape
    for k, n in enumerate(order):
        assert np.array_equal(back[:, :, k].view(np.uint32), a[:, :, names.index(n)].view(np.uint32))
    raw = open(path, "rb").read()
    assert raw[:4] == b"\x76\x2f\x31\x01" and len(raw) > a.nbytes
    with pytest.raises(_host.HostError):
        _host.write_exr(path, a[:, :, :2], ["Y", "Y"])


def test_load_dict_matches_load_string(mitsuba):
    """animated_trans_rad.py-style dictionaries (python_scripts/animated_trans_rad.py:100-230)."""
    from beifong_amd.mitsuba.core import Transform4f
    from beifong_amd.mitsuba.core.xml import load_dict
    bsdfs = load_dict({"type": "twosided", "id": "material", "bsdf": {"type": "diffuse", "reflectance": {"type": "spectrum", "value": 1}}})
    targ = load_dict({"type": "rectangle", "to_world": Transform4f.look_at([0, -4, 0], [0, 0, 0], [0, 0, 1]), "bsdf": bsdfs})
    scene = load_dict({
        "type": "scene",
        "integrator": {"type": "range", "integrator": {"type": "pathlength"}, "dr": 0.2, "bins": 50},
        "sensor": {"type": "perspective", "near_clip": 0.2, "far_clip": 10.2, "fov_axis": "x", "fov": 45,
                   "to_world": Transform4f.look_at([0, 0, 0], [0, -1, 0], [0, 0, 1]),
                   "sampler": {"type": "independent", "sample_count": 500},
                   "film": {"type": "hdrfilm", "rfilter": {"type": "box"}, "width": 1, "height": 1}},
        "emitter": {"type": "spot", "cutoff_angle": 25, "beam_width": 20, "intensity": {"type": "spectrum", "value": 1000},
                    "to_world": Transform4f.look_at([0, 0, 0], [0, -1, 0], [0, 0, 1])},
        "so": targ,
    })
    lp, h, rec = _oracle_on_host_scene(scene, scene.sensors()[0])
    assert lp.mode == capi.BF_MODE_RANGE and lp.bins == 50 and np.isclose(lp.bin_width, 0.2)
    assert h[4] == 500 and h[5:].sum() > 0
    assert np.nonzero(h[5:])[0].min() >= 19          # target plate 4 m away, dr 0.2 m: first return at bin >= 19


def test_bfrender_cli_without_gpu_fails_cleanly(mitsuba, tmp_path):
    lib = capi.load_library()
    if lib.bf_device_count() > 0:
        pytest.skip("a GPU is present")
    p = tmp_path / "s.xml"
    p.write_text(TRANS_RAD_LIKE)
    r = subprocess.run([os.path.join(HOST, "bfrender"), "-m", "scalar_rgb", "-Dspp=16", str(p)], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr


@pytest.mark.parametrize("direction", [[0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])
@pytest.mark.parametrize("fov", [34, 80])
def test_perspective_sensor_known_answers(mitsuba, direction, fov):
    """src/sensors/tests/test_perspective.py:61-175 through the perspective plugin and the boundary: the ray of the film
    centre leaves the camera origin along the camera direction, and a film sample at the extremity of the fov axis makes
    an angle of fov / 2 with it — for fov_axis x / larger (512 x 256 film: x), y / smaller (y) and diagonal (corners)."""
    from beifong_amd.mitsuba.core import Transform4f
    from beifong_amd.mitsuba.core.xml import load_dict
    origin = [1.0, 0.0, 1.5]
    target = [origin[k] + direction[k] for k in range(3)]

    def camera(fov_axis):
        scene = load_dict({
            "type": "scene",
            "integrator": {"type": "path"},
            "sensor": {"type": "perspective", "near_clip": 1.0, "far_clip": 35.0, "fov": fov, "fov_axis": fov_axis,
                       "to_world": Transform4f.look_at(origin, target, [0, 1, 0]),
                       "sampler": {"type": "independent", "sample_count": 4},
                       "film": {"type": "hdrfilm", "rfilter": {"type": "box"}, "width": 512, "height": 256}},
            "so": {"type": "rectangle", "bsdf": {"type": "diffuse"}},
        })
        desc = scene.flat_desc(scene.sensors()[0])
        assert (desc.desc.sensor.film_width, desc.desc.sensor.film_height) == (512, 256)
        return OracleScene(desc)

    def angle(cam, sample):
        r = cam.sensor_sample_ray(*sample)
        assert np.allclose(r["o"], origin, atol=1e-6)
        return np.degrees(np.arccos(np.clip(np.dot(r["d"], direction), -1, 1)))

    cam = camera("x")
    r = cam.sensor_sample_ray(0.5, 0.5)
    assert np.allclose(r["d"], direction, atol=1e-6) and np.allclose(r["o"], origin, atol=1e-6) and r["weight"] == 1.0
    for axis in ("x", "larger"):
        cam = camera(axis)
        for sample in ([0.0, 0.5], [1.0, 0.5]):
            assert np.isclose(angle(cam, sample), fov / 2, atol=1e-3)
    for axis in ("y", "smaller"):
        cam = camera(axis)
        for sample in ([0.5, 0.0], [0.5, 1.0]):
            assert np.isclose(angle(cam, sample), fov / 2, atol=1e-3)
    cam = camera("diagonal")
    for sample in ([0.0, 0.0], [0.0, 1.0], [1.0, 0.0], [1.0, 1.0]):
        assert np.isclose(angle(cam, sample), fov / 2, atol=1e-3)


def test_irradiancemeter_is_the_flux_meter_per_unit_area(mitsuba):
    """src/sensors/irradiancemeter.cpp:63-105 (the sensor python_scripts/trans_image.xml names): the flux meter's rays with
    weight pi / surface_area instead of pi — same paths, radiance scaled by 1 / area, AOV bins untouched."""
    from beifong_amd.mitsuba.core.xml import load_string
    out = {}
    for kind in ("fluxmeter", "irradiancemeter"):
        scene = load_string(TRANS_RAD_LIKE.replace('type="fluxmeter"', 'type="%s"' % kind), spp=3000)
        lp, h, rec = _oracle_on_host_scene(scene, scene.sensors()[0])
        out[kind] = (h, rec)
    hf, rf = out["fluxmeter"]
    hi, ri = out["irradiancemeter"]
    area = 4 * 0.05 * 0.05                                    # the rectangle [-1, 1]^2 scaled by 0.05: 0.1 m x 0.1 m
    assert np.array_equal(rf["n_rays"], ri["n_rays"]) and np.array_equal(rf["aux"], ri["aux"])
    assert np.allclose(ri["L"], rf["L"] / area, rtol=1e-5) and rf["L"].max() > 0
    assert np.allclose(hi[:3], hf[:3] / area, rtol=1e-4) and np.array_equal(hi[3:], hf[3:])


def test_radiancemeter_is_a_pencil_beam(mitsuba):
    """src/sensors/radiancemeter.cpp:49-114 (src/sensors/tests/test_radiancemeter.py): every sample is the one ray
    origin + t * direction, weight 1, whatever the film / aperture samples are; neither origin-only nor a non-1x1 film is accepted."""
    from beifong_amd.mitsuba.core.xml import load_string
    from beifong_amd.mitsuba._host import HostError
    xml = TRANS_RAD_LIKE.replace('<sensor type="fluxmeter">', '<sensor type="radiancemeter"><point name="origin" x="0" y="0" z="0.5"/>'
                                 '<vector name="direction" x="0" y="-2" z="0"/>')
    # the sensor is a child of the rx rectangle in that scene: lift it to scene level
    scene = load_string(xml, spp=500)
    sensor = scene.sensors()[0]
    o = OracleScene(scene.flat_desc(sensor))
    for sample in ((0.1, 0.9), (0.5, 0.5), (0.99, 0.0)):
        r = o.sensor_sample_ray(*sample)
        assert np.allclose(r["o"], [0, 0, 0.5], atol=1e-6) and np.allclose(r["d"], [0, -1, 0], atol=1e-6) and r["weight"] == 1.0
    lp, h, rec = _oracle_on_host_scene(scene, sensor)
    assert h[4] == 500 and np.all(rec["valid"] == 1)            # the target plate 4 m down the -y axis is hit every time
    with pytest.raises(HostError, match="both values"):
        load_string(xml.replace('<vector name="direction" x="0" y="-2" z="0"/>', ''), spp=4)


def test_script_level_names_of_mitsuba_core(mitsuba, tmp_path):
    """The names the reference's radar scripts import from mitsuba.core (python_scripts/animated_trans_rad.py:11-12,304-311;
    Render.py:22-30,467-469; trans_rad.py:13,24): Vector3f, Transform4f / ScalarTransform4f, Bitmap, Struct, Thread."""
    import struct
    import zlib
    from beifong_amd.mitsuba.core import Vector3f, Point3f, Transform4f, ScalarTransform4f, Bitmap, Struct, Thread
    # the antenna sweep of animated_trans_rad.py:304-311
    lorigin, boresight = Vector3f(0, 0, 0), Vector3f(0, -1, 0)
    rot = Transform4f.rotate(Vector3f(0, 0, 1), 30.0)
    new_boresight, new_up = rot.transform_vector(boresight), rot.transform_vector(Vector3f(0, 0, 1))
    assert np.allclose(new_boresight, [0.5, -np.sqrt(0.75), 0], atol=1e-6) and np.allclose(new_up, [0, 0, 1])
    to_world = Transform4f.look_at(lorigin, new_boresight, new_up)
    assert np.allclose(to_world.transform_vector([0, 0, 1]), new_boresight, atol=1e-6)      # the camera looks along +z
    assert ScalarTransform4f is Transform4f and Point3f(1, 2, 3).z == 3 and Vector3f(2.0)[1] == 2
    Thread.thread().file_resolver().append(str(tmp_path))
    # Render.py:467-469: Bitmap(array, XYZAW).convert(RGB, UInt8, srgb_gamma=True).write(...)
    img = np.zeros((2, 3, 5), dtype=np.float32)
    img[:, :, :3] = [0.950456 * 4, 1.0 * 4, 1.08875 * 4]       # D65 white, accumulated with weight 4
    img[:, :, 3], img[:, :, 4] = 4, 4
    img[1, 2, :3] = 0
    out = Bitmap(img, Bitmap.PixelFormat.XYZAW).convert(Bitmap.PixelFormat.RGB, Struct.Type.UInt8, srgb_gamma=True)
    a = np.array(out)
    assert a.dtype == np.uint8 and a.shape == (2, 3, 3) and out.pixel_format() == "RGB"
    assert np.all(a[0, 0] >= 254) and np.all(a[1, 2] == 0)     # white / weight -> sRGB white; the empty pixel stays black
    lin = Bitmap(img, Bitmap.PixelFormat.XYZAW).convert(Bitmap.PixelFormat.Y, Struct.Type.Float32)
    assert np.allclose(np.array(lin)[0, 0, 0], 1.0, atol=1e-6)
    out.write(tmp_path / "frame.png")
    raw = (tmp_path / "frame.png").read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n" and struct.unpack(">II", raw[16:24]) == (3, 2)
    idat = raw[raw.index(b"IDAT") + 4: raw.index(b"IEND") - 8]
    rows = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(2, 1 + 3 * 3)
    assert np.array_equal(rows[:, 1:].reshape(2, 3, 3), a)
    out.write(tmp_path / "frame.jpg")                            # no libjpeg here: lands as PNG next to it
    assert (tmp_path / "frame.jpg.png").exists()
    Bitmap(img, Bitmap.PixelFormat.XYZAW).write(tmp_path / "frame.exr")
    from beifong_amd.mitsuba import _host
    back, names = _host.read_exr(str(tmp_path / "frame.exr"))
    assert sorted(names) == ["A", "W", "X", "Y", "Z"] and np.allclose(back[:, :, names.index("X")], img[:, :, 0])


def test_load_dict_objects_stand_for_their_instances_in_the_scene(mitsuba):
    """animated_trans_rad.py / Receive.ipynb keep using the objects load_dict gave them BEFORE the scene existed (render(scene,
    sen), film.bitmap(), adc.bitmap()): such an object is its dictionary until a scene embeds it, then the scene's instance."""
    from beifong_amd.mitsuba.core import Transform4f
    from beifong_amd.mitsuba.core.xml import load_dict, dict_to_xml
    film = load_dict({"type": "hdrfilm", "width": 1, "height": 1, "rfilter": {"type": "box"}})
    sampler = load_dict({"type": "independent", "sample_count": 16})
    with pytest.raises(AttributeError, match="outside a scene"):
        film.bitmap(raw=True)
    mat = load_dict({"type": "twosided", "id": "material", "bsdf": {"type": "diffuse"}})
    a = load_dict({"type": "rectangle", "bsdf": mat})
    b = load_dict({"type": "rectangle", "to_world": Transform4f.translate([0, 0, 2]), "bsdf": mat})
    sen = load_dict({"type": "perspective", "fov": 45, "sampler": sampler, "film": film})
    d = {"type": "scene", "integrator": {"type": "path"}, "sensor": sen, "a": a, "b": b,
         "light": {"type": "rectangle", "emitter": {"type": "area", "radiance": {"type": "spectrum", "value": 1.0}}}}
    text = dict_to_xml(d)
    assert text.count('id="material"') == 2 and text.count("<ref ") == 1          # one shared instance: written once, referenced once
    scene = load_dict(d)
    assert sen.class_name() == scene.sensors()[0].class_name() and sampler.sample_count() == 16
    assert type(film._resolve()) is type(scene.sensors()[0].film())
    # a later scene re-binds the same objects (the per-frame scenes of the sweep scripts)
    scene2 = load_dict(dict(d, integrator={"type": "path", "max_depth": 3}))
    ptr = lambda o: getattr(o._ptr, "value", o._ptr)
    assert ptr(sen._resolve()) == ptr(scene2.sensors()[0]) != ptr(scene.sensors()[0])


def test_twosided_with_two_nested_bsdfs_flattens_to_a_front_and_a_back_entry(mitsuba):
    """src/bsdfs/tests/test_twosided.py:33-41 (test01_create, second half): roughconductor in front, diffuse behind."""
    from beifong_amd.mitsuba.core.xml import load_string
    from beifong_amd.mitsuba._host import HostError
    xml = """<scene version='2.0.0'><integrator type='path'/>
        <sensor type='perspective'><film type='hdrfilm'><integer name='width' value='1'/><integer name='height' value='1'/><rfilter type='box'/></film>
            <sampler type='independent'><integer name='sample_count' value='4'/></sampler></sensor>
        <shape type='rectangle'><bsdf type='twosided'><bsdf type='roughconductor'/><bsdf type='diffuse'><spectrum name='reflectance' value='0.9'/></bsdf></bsdf></shape>
        <shape type='rectangle'><transform name='to_world'><translate z='2'/></transform><bsdf type='twosided'><bsdf type='diffuse'/></bsdf></shape>
        <shape type='rectangle'><transform name='to_world'><translate z='5'/></transform><emitter type='area'><spectrum name='radiance' value='1'/></emitter></shape>
    </scene>"""
    sc = load_string(xml)
    d = sc.flat_desc(sc.sensors()[0]).desc
    mats = [d.materials[i] for i in range(d.n_materials)]
    front = mats[d.shapes[0].material]
    assert front.type == capi.BF_BSDF_ROUGHCONDUCTOR and front.twosided == 1 and front.back_material != 0
    back = mats[front.back_material - 1]
    assert back.type == capi.BF_BSDF_DIFFUSE and back.twosided == 1 and back.back_material == 0 and abs(back.reflectance - 0.9) < 1e-7
    single = mats[d.shapes[1].material]
    assert single.type == capi.BF_BSDF_DIFFUSE and single.twosided == 1 and single.back_material == 0
    with pytest.raises(HostError, match="At most two nested BSDFs"):
        load_string(xml.replace("<bsdf type='roughconductor'/>", "<bsdf type='roughconductor'/><bsdf type='diffuse'/>"))


def test_hdrfilm_and_hdradc_constructor_checks(mitsuba):
    """src/films/tests/test_hdrfilm.py:7-32 (test01_construct): default and given reconstruction filter; component_format "uint8"
    and pixel_format "brga" are errors (hdrfilm.cpp:100-172).  The fork's hdradc: OpenEXR, luminance only (hdradc.cpp:107-147)."""
    from beifong_amd.mitsuba.core.xml import load_string
    from beifong_amd.mitsuba._host import HostError
    assert load_string("<film version='2.0.0' type='hdrfilm'></film>") is not None
    scene = """<scene version='2.0.0'><integrator type='path'/><sensor type='perspective'>%s
        <sampler type='independent'/></sensor><shape type='rectangle'><emitter type='area'><spectrum name='radiance' value='1'/></emitter></shape></scene>"""
    sc = load_string(scene % "<film type='hdrfilm'><rfilter type='gaussian'><float name='stddev' value='18.5'/></rfilter></film>")
    assert sc.flat_desc(sc.sensors()[0]).desc.sensor.rfilter.radius == 4 * 18.5
    sc = load_string(scene % "<film type='hdrfilm'/>")
    assert sc.flat_desc(sc.sensors()[0]).desc.sensor.rfilter.radius == 2.0            # the default: gaussian, stddev 0.5
    for bad in ("<string name='component_format' value='uint8'/>", "<string name='pixel_format' value='brga'/>",
                "<string name='file_format' value='png'/>"):
        with pytest.raises(HostError, match="parameter must"):
            load_string("<film version='2.0.0' type='hdrfilm'>%s</film>" % bad)
    for ok in ("<string name='pixel_format' value='XYZA'/>", "<string name='component_format' value='float32'/>", "<string name='file_format' value='exr'/>"):
        load_string("<film version='2.0.0' type='hdrfilm'>%s</film>" % ok)
    load_string("<adc version='2.0.0' type='hdradc'><string name='component_format' value='uint32'/></adc>")
    for bad in ("<string name='pixel_format' value='rgb'/>", "<string name='file_format' value='pfm'/>", "<string name='component_format' value='uint8'/>"):
        with pytest.raises(HostError, match="parameter must"):
            load_string("<adc version='2.0.0' type='hdradc'>%s</adc>" % bad)


def test_sampling_integrator_base_properties(mitsuba):
    """SamplingIntegrator (integrator.cpp:26-43,66-75): block_size (rounded up to a power of two; the blocks a film is rendered
    in — it reaches the flattened reconstruction filter), samples_per_pass (sample_count must be a multiple), timeout and
    hide_emitters (accepted: the radar integrators never read the latter, as in the reference)."""
    from beifong_amd.mitsuba.core.xml import load_string
    from beifong_amd.mitsuba._host import HostError
    xml = """<scene version='2.0.0'><integrator type='range'><integer name='bins' value='4'/><float name='dr' value='1'/>%s<integrator type='pathlength'/></integrator>
     <sensor type='perspective'><film type='hdrfilm'><integer name='width' value='4'/><integer name='height' value='4'/></film>
     <sampler type='independent'><integer name='sample_count' value='12'/></sampler></sensor>
     <shape type='rectangle'><emitter type='area'><spectrum name='radiance' value='1'/></emitter></shape></scene>"""
    sc = load_string(xml % "<integer name='block_size' value='5'/><integer name='samples_per_pass' value='4'/><float name='timeout' value='3'/>"
                           "<boolean name='hide_emitters' value='true'/>")
    assert sc.flat_desc(sc.sensors()[0]).desc.sensor.rfilter.block_size == 8
    assert load_string(xml % "").flat_desc(load_string(xml % "").sensors()[0]).desc.sensor.rfilter.block_size == 32
    bad = load_string(xml % "<integer name='samples_per_pass' value='5'/>")
    with pytest.raises(HostError, match=r"sample_count \(12\) must be a multiple of samples_per_pass \(5\)"):
        bad.integrator().render(bad, bad.sensors()[0])
