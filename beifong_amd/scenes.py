"""The BASELINE.json configurations as flat scene descriptions.

C1 follows python_scripts/trans_rad.xml tag by tag; C2-C4 follow the
geometry/materials of python_scripts/Render.py:196-392 and the gen-2
integrator of python_scripts/animated_trans_rad.py:163-169, with seeded
synthetic meshes standing in for the stripped scans.
"""
import numpy as np

from . import capi, meshgen
from .scenedesc import SceneDesc, Transform4f

T = Transform4f


def trans_rad(spp=16):
    """C1: python_scripts/trans_rad.xml (gen-1 `time` o `pathtime`)."""
    sd = SceneDesc()
    mat = sd.add_diffuse(1.0, twosided=True)                     # :35-39
    rx_mat = sd.add_diffuse(0.5)                                 # shape.cpp:89-98 default bsdf
    # receiveAntenna :16-31 — <scale .05 .05/> then <lookat/> => lookat * scale
    rx = sd.add_rectangle(T.look_at([0, 0, 0], [0, -1, 0], [0, 0, 1]) * T.scale([0.05, 0.05, 1]), rx_mat)
    sd.set_fluxmeter(rx)
    # spot :43-50
    sd.add_spot(T.look_at([0, 0, 0], [0, -1, 0], [0, 0, 1]), intensity=1.0, cutoff_angle=25.0, beam_width=20.0)
    # target :65-71, gnd :73-79 (lookat without up: xml.cpp:911-913)
    sd.add_rectangle(T.look_at([0, -4, 0], [0, 0, 0], [0, 0, 1]) * T.scale([1, 1, 1]), mat)
    sd.add_rectangle(T.look_at([0, 0, -0.5], [0, 0, 0.5], [0, 0, 0]) * T.scale([20, 20, 1]), mat)
    sd.finalize()
    launch = capi.make_launch(capi.BF_MODE_TIME, spp, seed=0, bins=50, bin_width=0.5e-9, time_c=3.0e8,
                              color_mode=capi.BF_COLOR_RGB)
    return sd, launch


def _radar_frontend(sd, radiance=1000.0, yaw_deg=0.0, position=(0.0, 0.0, 0.3)):
    """Monostatic front end of Render.py:196-271: a 20 x 50 mm TX aperture at
    (0,0,0.3) looking +x carrying an area emitter (gen-2 stand-in for the
    wignertransmitter), and a perspective RX at the same position.  `yaw_deg` turns the radar about
    the vertical axis like the frame loop of animated_trans_rad.py:307-373 turns sensor and emitter."""
    d0 = T.rotate([0, 0, 1], yaw_deg) * T.rotate([1, 0, 0], 90) * T.rotate([0, 1, 0], 90)        # txa_d0 = align_x * align_z
    tx_mat = sd.add_diffuse(0.0)                                  # emitter shape: rho = 0 (shape.cpp:89-98)
    txa = sd.add_rectangle(T.translate(list(position)) * d0 * T.scale([20e-3, 50e-3, 1]), tx_mat)
    sd.add_area_emitter(txa, radiance)
    sd.set_perspective(T.translate(list(position)) * d0, fov=45.0, near_clip=0.1, far_clip=100.0)


def _ground(sd):
    gnd = sd.add_diffuse(0.5, twosided=True)                      # B_GND Render.py:322-334
    sd.add_rectangle(T.translate([0, 0, 0]) * T.scale([20, 20, 1]), gnd)


def bus_mesh(n_tris=200_000):
    """The placed bus mesh of C2 as contiguous arrays (reusable across the frames of a sweep):
    positions, faces and the vertex normals the `obj` loader computes for a scan without `vn` lines
    (Render.py:386-392 loads Bus.obj through src/shapes/obj.cpp; mesh.cpp:201-249)."""
    v, f = meshgen.bus(n_tris, seed=1)
    # car_trafo Render.py:305-316: translate(10,3,1) * yaw(-20) (scan-axis alignment folded into the generator)
    v = meshgen.place(v, yaw_deg=-20.0, translate=(10.0, 3.0, 1.7))
    v = np.ascontiguousarray(v, dtype=np.float32)
    f = np.ascontiguousarray(f, dtype=np.uint32)
    return v, f, np.ascontiguousarray(meshgen.vertex_normals(v, f))


def bus_radar(n_tris=200_000, n_paths=64, bins=256, dr=0.1, seed=1, radar_yaw_deg=0.0, mesh=None,
              radar_position=(0.0, 0.0, 0.3)):
    """C2: Bus.obj-class monostatic radar scene, gen-2 `range` o `pathlength`."""
    sd = SceneDesc()
    _radar_frontend(sd, yaw_deg=radar_yaw_deg, position=radar_position)
    _ground(sd)
    car = sd.add_roughconductor(alpha=0.1, twosided=True, specular_reflectance=1.0)   # B_CAR :353-363
    v, f, n = mesh if mesh is not None else bus_mesh(n_tris)
    sd.add_mesh(v, f, car, normals=n)
    sd.finalize()
    launch = capi.make_launch(capi.BF_MODE_RANGE, n_paths, seed=seed, bins=bins, bin_width=dr, color_mode=capi.BF_COLOR_RGB)
    return sd, launch


def car_radar(n_tris=1_000_000, n_paths=1 << 20, bins=1024, dr=0.03, seed=2):
    """C3: Car-body.ply-class shell (vertex normals) + ground."""
    sd = SceneDesc()
    _radar_frontend(sd)
    _ground(sd)
    car = sd.add_roughconductor(alpha=0.1, twosided=True, specular_reflectance=1.0)
    v, f, n = meshgen.car_body(n_tris, seed=2)
    a = np.radians(-20.0)
    r = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    v = meshgen.place(v, yaw_deg=-20.0, translate=(8.0, 1.5, 0.75))
    n = (n.astype(np.float64) @ r.T).astype(np.float32)
    sd.add_mesh(v, f, car, normals=n)
    sd.finalize()
    launch = capi.make_launch(capi.BF_MODE_RANGE, n_paths, seed=seed, bins=bins, bin_width=dr, color_mode=capi.BF_COLOR_RGB)
    return sd, launch


def multi_mesh_radar(n_paths=4096 << 10, bins=4096, dr=0.01, seed=3, scale=1.0):
    """C4: bus + car + motorbike on the ground plane."""
    sd = SceneDesc()
    _radar_frontend(sd)
    _ground(sd)
    mat = sd.add_roughconductor(alpha=0.1, twosided=True, specular_reflectance=1.0)
    v, f = meshgen.bus(int(200_000 * scale), seed=1)
    sd.add_mesh(meshgen.place(v, -20.0, (12.0, 3.0, 1.7)), f, mat)
    v, f, n = meshgen.car_body(int(1_000_000 * scale), seed=2, with_normals=False)
    sd.add_mesh(meshgen.place(v, 15.0, (7.0, -2.5, 0.75)), f, mat)
    v, f = meshgen.motorbike(int(300_000 * scale), seed=5)
    sd.add_mesh(meshgen.place(v, 40.0, (5.0, 1.0, 0.0)), f, mat)
    sd.finalize()
    launch = capi.make_launch(capi.BF_MODE_RANGE, n_paths, seed=seed, bins=bins, bin_width=dr, color_mode=capi.BF_COLOR_RGB)
    return sd, launch


def bus_receive(n_tris=200_000, n_paths=64, t_bins=256, dr=0.1, seed=1, transmitter="wigner", receiver="omnidirectional",
                signaltype="pulse", lambda_band_nm=None, radar_yaw_deg=0.0, mesh=None):
    """C2-recv: C2 geometry through gen-3 receive():
    wignertransmitter (pulse tau = 2 dr / c, prf = 1/T) on the TX aperture,
    omnidirectional receiver on a coincident RX aperture, ADC t_bins x 1 with
    t_bandwidth = T = t_bins * tau, f_bandwidth = 2 c / lambda_min (one frequency row)."""
    sd = SceneDesc()
    if lambda_band_nm is not None:      # MTS_WAVELENGTH_MIN / MAX are compile-time choices of the fork (spectrum.h:15-30)
        sd.physics.lambda_min_nm, sd.physics.lambda_max_nm = lambda_band_nm
    c, lmin, lmax = sd.physics.c, sd.physics.lambda_min_nm, sd.physics.lambda_max_nm
    d0 = T.rotate([1, 0, 0], 90) * T.rotate([0, 1, 0], 90)
    # radar_yaw_deg: the radar turned about the vertical axis (the sweep loops of Receive.ipynb cell 30 rebuild TX / RX per angle)
    aperture = T.translate([0, 0, 0.3]) * T.rotate([0, 0, 1], radar_yaw_deg) * d0 * T.scale([20e-3, 50e-3, 1])
    tx_mat = sd.add_diffuse(0.0)        # transmitter shape: rho = 0 (shape.cpp:89-98)
    rx_mat = sd.add_diffuse(0.5)
    txa = sd.add_rectangle(aperture, tx_mat)
    rxa = sd.add_rectangle(aperture, rx_mat)
    tau = 2.0 * dr / c
    t_total = t_bins * tau
    f_c = c / (0.5 * (lmin + lmax) * 1e-9)
    if transmitter == "wigner":
        sd.add_wigner_transmitter(txa, signaltype=signaltype, amplitude=1.0, freq_centre=f_c, freq_ext=1.0 / tau,
                                  pulse_len=tau, prf=1.0 / t_total, gain=1.0)
    else:
        sd.add_area_transmitter(txa, 1.0)
    sd.set_receiver(rxa, kind=receiver, adc_sampling_start=0.0, adc_sampling_end=t_total, t_bins=t_bins, f_bins=1,
                    t_bandwidth=t_total, f_bandwidth=2.0 * c / (lmin * 1e-9), freq_centre=f_c,
                    freq_ext=c / (lmin * 1e-9) - c / (lmax * 1e-9))
    _ground(sd)
    car = sd.add_roughconductor(alpha=0.1, twosided=True, specular_reflectance=1.0)
    v, f = mesh if mesh is not None else meshgen.bus(n_tris, seed=1)
    sd.add_mesh(meshgen.place(v, yaw_deg=-20.0, translate=(10.0, 3.0, 1.7)), f, car)
    sd.finalize()
    launch = capi.make_launch(capi.BF_MODE_RECEIVE_RAW, n_paths, seed=seed, bins=t_bins, bins_y=1)
    return sd, launch


def plate_doppler(wavelength_m=0.1, n_paths=1 << 16, t_bins=1, plate_x=5.0, plate_size=0.5, n_grid=8, seed=4,
                  ground=False):
    """Narrow-band coherent scene for pulse sweeps: coincident 20 x 50 mm TX / RX apertures at (0,0,0.3) looking
    +x, an area transmitter (no signal gating) and an omnidirectional receiver, a diffuse mesh plate facing
    them at x = plate_x (the moving target), optionally the 20 x 20 m ground.  The band is +-1e-6 around
    `wavelength_m`, so every ray carries that wavelength."""
    sd = SceneDesc()
    lam_nm = wavelength_m * 1e9
    sd.physics.lambda_min_nm = lam_nm * (1 - 1e-6)
    sd.physics.lambda_max_nm = lam_nm * (1 + 1e-6)
    c = sd.physics.c
    d0 = T.rotate([1, 0, 0], 90) * T.rotate([0, 1, 0], 90)
    aperture = T.translate([0, 0, 0.3]) * d0 * T.scale([20e-3, 50e-3, 1])
    txa = sd.add_rectangle(aperture, sd.add_diffuse(0.0))
    rxa = sd.add_rectangle(aperture, sd.add_diffuse(0.5))
    sd.add_area_transmitter(txa, 1.0)
    t_total = 2.0e-7
    f_c = c / wavelength_m
    sd.set_receiver(rxa, kind="omnidirectional", adc_sampling_start=0.0, adc_sampling_end=t_total, t_bins=t_bins, f_bins=1,
                    t_bandwidth=t_total, f_bandwidth=2.0 * f_c, freq_centre=f_c, freq_ext=f_c * 2e-6)
    if ground:
        _ground(sd)
    # plate in the y-z plane, normal -x, n_grid x n_grid quads
    g = np.linspace(-0.5 * plate_size, 0.5 * plate_size, n_grid + 1)
    yy, zz = np.meshgrid(g, g, indexing="ij")
    v = np.stack([np.full(yy.size, plate_x), yy.ravel(), zz.ravel() + 0.3], -1).astype(np.float32)
    idx = np.arange((n_grid + 1) ** 2).reshape(n_grid + 1, n_grid + 1)
    a, b, cc, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    f = np.concatenate([np.stack([a, b, cc], -1), np.stack([a, cc, d], -1)]).astype(np.uint32)
    sd.add_mesh(np.ascontiguousarray(v), np.ascontiguousarray(f), sd.add_diffuse(0.8, twosided=True))
    sd.finalize()
    launch = capi.make_launch(capi.BF_MODE_RECEIVE_IQ, n_paths, seed=seed, bins=t_bins, bins_y=1)
    return sd, launch


def fmcw_plate(plate_x=5.0, r_max=12.8, n_paths=1 << 18, t_bins=8, f_bins=64, plate_size=0.5, seed=9):
    """The de-chirped FMCW return of a plate (round 4): coincident 20 x 50 mm TX / RX apertures at (0, 0, 0.3) looking +x, a Wigner
    transmitter with resample_freq that sweeps B = 1e-3 f_c over the receive window T = 4 * (2 r_max / c), a Wigner receiver whose
    local oscillator is the same chirp (receive_type "mix_resample", a delta signal), a diffuse plate facing them at x = plate_x.
    The ADC's f_bins frequency rows span the beat of r_max: a return from range r lands in row ceil(f_bins * r / r_max - 1)
    whenever it is received (launch: BF_MODE_RECEIVE_RAW | BF_FLAG_MIX_RESAMPLE)."""
    sd = SceneDesc()
    lam_nm = 0.1 * 1e9
    sd.physics.lambda_min_nm, sd.physics.lambda_max_nm = lam_nm * (1 - 1e-3), lam_nm * (1 + 1e-3)
    c = sd.physics.c
    d0 = T.rotate([1, 0, 0], 90) * T.rotate([0, 1, 0], 90)
    aperture = T.translate([0, 0, 0.3]) * d0 * T.scale([20e-3, 50e-3, 1])
    txa = sd.add_rectangle(aperture, sd.add_diffuse(0.0))
    rxa = sd.add_rectangle(aperture, sd.add_diffuse(0.5))
    t_win, f_c = 4.0 * (2.0 * r_max / c), c / 0.1
    sweep = 1e-3 * f_c
    sd.add_wigner_transmitter(txa, signaltype="linfmcw", amplitude=1.0, freq_centre=f_c, freq_ext=sweep, pulse_len=t_win,
                              prf=1.0 / t_win, resample_freq=True)
    sd.set_receiver(rxa, kind="wigner", adc_sampling_start=0.0, adc_sampling_end=t_win, t_bins=t_bins, f_bins=f_bins,
                    t_bandwidth=t_win, f_bandwidth=sweep * (2.0 * r_max / c) / t_win, freq_centre=f_c, freq_ext=sweep,
                    sig_is_delta=True, rx_signaltype="linfmcw", rx_chirp_len=t_win, rx_crf=1.0 / t_win)
    g = np.linspace(-0.5 * plate_size, 0.5 * plate_size, 9)
    yy, zz = np.meshgrid(g, g, indexing="ij")
    v = np.stack([np.full(yy.size, plate_x), yy.ravel(), zz.ravel() + 0.3], -1).astype(np.float32)
    idx = np.arange(81).reshape(9, 9)
    a, b, cc, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    f = np.concatenate([np.stack([a, b, cc], -1), np.stack([a, cc, d], -1)]).astype(np.uint32)
    sd.add_mesh(np.ascontiguousarray(v), np.ascontiguousarray(f), sd.add_diffuse(0.8, twosided=True))
    sd.finalize()
    launch = capi.make_launch(capi.BF_MODE_RECEIVE_RAW, n_paths, seed=seed, bins=t_bins, bins_y=f_bins, flags=capi.BF_FLAG_MIX_RESAMPLE)
    return sd, launch


def single_mesh(v, f, normals=None, texcoords=None):
    """Bare mesh scene for Scene::ray_intersect tests (test_kdtrees.py style)."""
    sd = SceneDesc()
    m = sd.add_diffuse(0.5)
    sd.add_mesh(v, f, m, normals=normals, texcoords=texcoords)
    sd.set_perspective(T.translate([0, 0, 0]), fov=45.0, near_clip=0.1, far_clip=100.0)
    sd.finalize()
    return sd


def phased_receive(n_tris=20000, n_paths=20000, n_elems=4, steer_deg=(0.0, 0.0, 0.0), phased_rx=True, phased_tx=True,
                   t_bins=64, dr=0.1, seed=6):
    """Gen-3 scene with the fork's phased-array endpoints (src/transmitters/phasedtransmitter.cpp,
    src/receivers/phasedreceiver.cpp): n_elems elements of 20 x 50 mm spaced 25 mm along the aperture's local x axis,
    steered by `steer_deg`; pulse signal as in bus_receive; the bus + ground as targets."""
    sd = SceneDesc()
    c, lmin, lmax = sd.physics.c, sd.physics.lambda_min_nm, sd.physics.lambda_max_nm
    d0 = T.rotate([1, 0, 0], 90) * T.rotate([0, 1, 0], 90)
    pose = T.translate([0, 0, 0.3]) * d0
    aperture = pose * T.scale([0.5 * n_elems * 25e-3, 25e-3, 1])
    txa = sd.add_rectangle(aperture, sd.add_diffuse(0.0))
    rxa = sd.add_rectangle(aperture, sd.add_diffuse(0.5))
    tau = 2.0 * dr / c
    t_total = t_bins * tau
    f_c = c / (0.5 * (lmin + lmax) * 1e-9)
    arr = lambda: sd.phased_array(n_elems, elem_dims=[20e-3, 50e-3, 1.0], elem_spacing=[25e-3, 0.0, 0.0], elem_axis=[1.0, 0.0, 0.0],
                                  steering_vector=np.radians(steer_deg), array_loc=pose)
    if phased_tx:
        sd.add_phased_transmitter(txa, arr(), signaltype="pulse", amplitude=1.0, freq_centre=f_c, freq_ext=1.0 / tau,
                                  pulse_len=tau, prf=1.0 / t_total, gain=1.0)
    else:
        sd.add_wigner_transmitter(txa, signaltype="pulse", amplitude=1.0, freq_centre=f_c, freq_ext=1.0 / tau, pulse_len=tau,
                                  prf=1.0 / t_total, gain=1.0)
    sd.set_receiver(rxa, kind="phased" if phased_rx else "omnidirectional", adc_sampling_start=0.0, adc_sampling_end=t_total,
                    t_bins=t_bins, f_bins=1, t_bandwidth=t_total, f_bandwidth=2.0 * c / (lmin * 1e-9), freq_centre=f_c,
                    freq_ext=c / (lmin * 1e-9) - c / (lmax * 1e-9), array=arr() if phased_rx else None)
    _ground(sd)
    car = sd.add_roughconductor(alpha=0.1, twosided=True, specular_reflectance=1.0)
    v, f = meshgen.bus(n_tris, seed=1)
    sd.add_mesh(meshgen.place(v, yaw_deg=-20.0, translate=(10.0, 3.0, 1.7)), f, car)
    sd.finalize()
    launch = capi.make_launch(capi.BF_MODE_RECEIVE_RAW, n_paths, seed=seed, bins=t_bins, bins_y=1)
    return sd, launch


def film_half_lit(film=(4, 2), spp=256, radiance=3.0, mode=None, bins=0, dr=0.0):
    """Multi-pixel film known answer: a perspective camera at the origin looking along +z (up +y, horizontal
    fov 90 deg) and an area light at z = 2 that fills exactly the half of the view on the camera's +x side —
    the LEFT half of the image (sensor.h:196-231: sample.x = (1 - x_clip) / 2)."""
    sd = SceneDesc()
    w, h = film
    half_h = 2.0 * h / w                                    # view half-extents at z = 2: 2 horizontally, 2 h / w vertically
    light = sd.add_rectangle(T.translate([1.0, 0.0, 2.0]) * T.rotate([0, 1, 0], 180) * T.scale([1.0, half_h, 1.0]), sd.add_diffuse(0.0))
    sd.add_area_emitter(light, radiance)
    sd.set_perspective(T.look_at([0, 0, 0], [0, 0, 1], [0, 1, 0]), fov=90.0, near_clip=0.1, far_clip=100.0, film=film)
    sd.finalize()
    lp = capi.make_launch(capi.BF_MODE_PATH if mode is None else mode, w * h * spp, seed=11, bins=bins, bin_width=dr,
                          color_mode=capi.BF_COLOR_MONO, film=film, spp=spp)
    return sd, lp
