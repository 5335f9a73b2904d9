"""Build a flat bf_scene_desc (include/beifong_hip.h) from Python.

Transform4f mirrors include/mitsuba/core/transform.h: a transform carries its
matrix AND its inverse, both composed analytically in float32 exactly as the
reference composes them (scale^-1 = 1/s, translate^-1 = -t, look_at^-1 built
from the basis rows), so `to_object` is not a numerical inverse.
"""
import ctypes as C
import math

import numpy as np

from . import capi

f32 = np.float32


class Transform4f:
    def __init__(self, matrix=None, inverse=None):
        if matrix is None:
            matrix = np.eye(4, dtype=f32)
        self.matrix = np.asarray(matrix, dtype=f32).reshape(4, 4)
        if inverse is None:
            # Transform(const Matrix &value): inverse_transpose = transpose(inverse(value))
            inverse = np.linalg.inv(self.matrix.astype(np.float64)).astype(f32)
        self.inv = np.asarray(inverse, dtype=f32).reshape(4, 4)

    def __mul__(self, other):
        # transform.h operator*: (matrix * other.matrix, inverse_transpose * other.inverse_transpose)
        return Transform4f(_matmul32(self.matrix, other.matrix), _matmul32(other.inv, self.inv))

    def inverse(self):
        return Transform4f(self.inv, self.matrix)

    @staticmethod
    def translate(v):
        m = np.eye(4, dtype=f32)
        i = np.eye(4, dtype=f32)
        m[:3, 3] = np.asarray(v, dtype=f32)
        i[:3, 3] = -np.asarray(v, dtype=f32)
        return Transform4f(m, i)

    @staticmethod
    def scale(v):
        v = np.asarray(v, dtype=f32)
        m = np.diag(np.concatenate([v, [f32(1)]]).astype(f32))
        i = np.diag(np.concatenate([f32(1) / v, [f32(1)]]).astype(f32))
        return Transform4f(m, i)

    @staticmethod
    def rotate(axis, angle_deg):
        # enoki::rotate<Matrix>(axis, deg_to_rad(angle)); inverse = transpose
        a = np.asarray(axis, dtype=np.float64)
        a = a / np.linalg.norm(a)
        ang = math.radians(angle_deg)
        s, c = math.sin(ang), math.cos(ang)
        x, y, z = a
        r = np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s, 0],
                      [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s, 0],
                      [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c), 0],
                      [0, 0, 0, 1]], dtype=np.float64).astype(f32)
        return Transform4f(r, r.T.copy())

    @staticmethod
    def look_at(origin, target, up):
        # include/mitsuba/core/transform.h:241-268
        origin = np.asarray(origin, dtype=f32)
        target = np.asarray(target, dtype=f32)
        up = np.asarray(up, dtype=f32)
        d = _normalize32(target - origin)
        d = _normalize32(d)
        if float(np.dot(up, up)) == 0.0:
            # xml.cpp:911-913: up = coordinate_system(dir).first
            up = _coordinate_system(d)[0]
        left = _normalize32(np.cross(up, d).astype(f32))
        new_up = np.cross(d, left).astype(f32)
        m = np.eye(4, dtype=f32)
        m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, origin
        inv = np.eye(4, dtype=f32)
        inv[0, :3], inv[1, :3], inv[2, :3] = left, new_up, d
        inv[0, 3] = -np.dot(left, origin)
        inv[1, 3] = -np.dot(new_up, origin)
        inv[2, 3] = -np.dot(d, origin)
        return Transform4f(m, inv)

    @staticmethod
    def perspective(fov_deg, near, far):
        # include/mitsuba/core/transform.h:203-220
        near, far = f32(near), f32(far)
        recip = f32(1) / (far - near)
        tan = f32(math.tan(math.radians(float(f32(fov_deg) * f32(0.5)))))
        cot = f32(1) / tan
        m = np.zeros((4, 4), dtype=f32)
        m[0, 0], m[1, 1], m[2, 2] = cot, cot, far * recip
        m[2, 3] = -near * far * recip
        m[3, 2] = 1
        inv = np.zeros((4, 4), dtype=f32)
        inv[0, 0], inv[1, 1], inv[3, 3] = tan, tan, f32(1) / near
        inv[2, 3] = 1
        inv[3, 2] = (near - far) / (far * near)
        return Transform4f(m, inv)


def _matmul32(a, b):
    return (a.astype(f32) @ b.astype(f32)).astype(f32)


def _normalize32(v):
    v = v.astype(f32)
    return (v * (f32(1) / np.sqrt(np.dot(v, v), dtype=f32))).astype(f32)


def _coordinate_system(n):
    # include/mitsuba/core/vector.h:116-136
    n = n.astype(f32)
    sign = f32(math.copysign(1.0, float(n[2])))
    a = f32(-1) / (sign + n[2])
    b = n[0] * n[1] * a
    s = np.array([sign * (n[0] * n[0] * a) + f32(1), sign * b, -sign * n[0]], dtype=f32)
    t = np.array([b, sign + n[1] * n[1] * a, -n[1]], dtype=f32)
    return s, t


def perspective_sample_to_camera(fov_x_deg, near, far, aspect=1.0, rel_size=(1.0, 1.0), rel_offset=(0.0, 0.0)):
    """m_sample_to_camera for a film of aspect = width / height; rel_size / rel_offset: crop size and offset over the film size.

    include/mitsuba/render/sensor.h:196-231: camera_to_sample =
    scale(1/rel) * translate(-off) * scale(-.5, -.5*aspect, 1) *
    translate(-1, -1/aspect, 0) * perspective(fov, near, far); inverse taken
    through Transform's analytic inverse composition.
    """
    aspect = float(f32(aspect))
    t = (Transform4f.scale([1.0 / float(f32(rel_size[0])), 1.0 / float(f32(rel_size[1])), 1]) *
         Transform4f.translate([-float(f32(rel_offset[0])), -float(f32(rel_offset[1])), 0]) * Transform4f.scale([-0.5, -0.5 * aspect, 1.0]) *
         Transform4f.translate([-1.0, -1.0 / aspect, 0.0]) * Transform4f.perspective(fov_x_deg, near, far))
    return t.inv.copy()


def _m16(m):
    return capi.M16(*[float(x) for x in np.asarray(m, dtype=f32).reshape(16)])


class SceneDesc:
    """Accumulates shapes / materials / emitters / sensor and owns the arrays."""

    def __init__(self, c=340.0, lambda_min_nm=7555556.0, lambda_max_nm=9714286.0):
        self.shapes, self.materials, self.emitters = [], [], []
        self.sensor = capi.bf_sensor()
        self.sensor.film_width = self.sensor.film_height = 1
        self.sensor.shape = -1
        self.physics = capi.bf_physics(c, lambda_min_nm, lambda_max_nm)
        self._keep = []
        self.desc = None

    # ---- materials ----
    def add_diffuse(self, reflectance=0.5, twosided=False):
        m = capi.bf_material(capi.BF_BSDF_DIFFUSE, int(twosided), reflectance, 0.1, 0.1, capi.BF_MF_BECKMANN, 1, 0.0, 1.0, 0)
        self.materials.append(m)
        return len(self.materials) - 1

    def add_roughconductor(self, alpha=0.1, twosided=False, specular_reflectance=None, distribution="beckmann",
                           eta=0.0, k=1.0, sample_visible=True, alpha_v=None):
        m = capi.bf_material(capi.BF_BSDF_ROUGHCONDUCTOR, int(twosided),
                             1.0 if specular_reflectance is None else specular_reflectance,
                             alpha, alpha if alpha_v is None else alpha_v,
                             capi.BF_MF_GGX if distribution == "ggx" else capi.BF_MF_BECKMANN,
                             int(sample_visible), eta, k, 0 if specular_reflectance is None else 1)
        self.materials.append(m)
        return len(self.materials) - 1

    def set_back_material(self, front, back):
        """<bsdf type="twosided"> with TWO nested BSDFs (twosided.cpp:62-92): material `back` shades the back side of `front`."""
        self.materials[front].twosided = self.materials[back].twosided = 1
        self.materials[front].back_material = back + 1
        return front

    # ---- shapes ----
    def add_rectangle(self, to_world, material, emitter=-1, is_sensor=False, velocity=None):
        s = capi.bf_shape()
        s.type, s.material, s.emitter, s.is_sensor = capi.BF_SHAPE_RECTANGLE, material, emitter, int(is_sensor)
        s.to_world, s.to_object = _m16(to_world.matrix), _m16(to_world.inv)
        s.velocity = _m16(np.eye(4) if velocity is None else velocity.matrix)      # Shape "velocity" transform (shape.cpp:42)
        self.shapes.append(s)
        return len(self.shapes) - 1

    def add_mesh(self, positions, indices, material, normals=None, emitter=-1, texcoords=None, velocity=None):
        pos = np.ascontiguousarray(positions, dtype=f32).reshape(-1, 3)
        idx = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        nrm = None if normals is None else np.ascontiguousarray(normals, dtype=f32).reshape(-1, 3)
        tex = None if texcoords is None else np.ascontiguousarray(texcoords, dtype=f32).reshape(-1, 2)
        self._keep += [pos, idx, nrm, tex]
        s = capi.bf_shape()
        s.type, s.material, s.emitter, s.is_sensor = capi.BF_SHAPE_MESH, material, emitter, 0
        s.to_world, s.to_object = _m16(np.eye(4)), _m16(np.eye(4))
        s.velocity = _m16(np.eye(4) if velocity is None else velocity.matrix)
        s.positions = pos.ctypes.data_as(C.POINTER(C.c_float))
        s.normals = nrm.ctypes.data_as(C.POINTER(C.c_float)) if nrm is not None else None
        s.texcoords = tex.ctypes.data_as(C.POINTER(C.c_float)) if tex is not None else None
        s.indices = idx.ctypes.data_as(C.POINTER(C.c_uint32))
        s.n_vertices, s.n_faces = pos.shape[0], idx.shape[0]
        self.shapes.append(s)
        return len(self.shapes) - 1

    # ---- emitters ----
    def add_spot(self, to_world, intensity=1.0, cutoff_angle=20.0, beam_width=None):
        e = capi.bf_emitter()
        e.type, e.shape = capi.BF_EMITTER_SPOT, -1
        e.to_world, e.to_object = _m16(to_world.matrix), _m16(to_world.inv)
        e.radiance, e.cutoff_angle_deg = intensity, cutoff_angle
        e.beam_width_deg = cutoff_angle * 3.0 / 4.0 if beam_width is None else beam_width
        self.emitters.append(e)
        return len(self.emitters) - 1

    def add_point(self, position, intensity=1.0):
        """point.cpp: isotropic point light at `position`."""
        e = capi.bf_emitter()
        e.type, e.shape, e.radiance = capi.BF_EMITTER_POINT, -1, intensity
        t = Transform4f.translate(list(position))
        e.to_world, e.to_object = _m16(t.matrix), _m16(t.inv)
        self.emitters.append(e)
        return len(self.emitters) - 1

    def add_area_emitter(self, shape, radiance=1.0):
        e = capi.bf_emitter()
        e.type, e.shape, e.radiance = capi.BF_EMITTER_AREA, shape, radiance
        e.to_world, e.to_object = _m16(np.eye(4)), _m16(np.eye(4))
        self.emitters.append(e)
        idx = len(self.emitters) - 1
        self.shapes[shape].emitter = idx
        return idx

    # ---- transmitters (gen-3) ----
    def add_area_transmitter(self, shape, radiance=1.0):
        e = capi.bf_emitter()
        e.type, e.shape, e.radiance = capi.BF_TRANSMITTER_AREA, shape, radiance
        e.to_world, e.to_object = _m16(np.eye(4)), _m16(np.eye(4))
        self.emitters.append(e)
        self.shapes[shape].emitter = len(self.emitters) - 1
        return len(self.emitters) - 1

    def add_wigner_transmitter(self, shape, signaltype="pulse", amplitude=1.0, freq_centre=1.0, freq_ext=1.0,
                               pulse_len=1.0, prf=1.0, gain=1.0, resample_freq=False):
        """wignertransmitter.cpp:53-110; chirp_len/crf/freq_sweep of "linfmcw" map onto pulse_len/prf/freq_ext.
        resample_freq (:211-221, 430-441): eval / sample_direction overwrite the path's wavelength with the signal's
        instantaneous frequency at the (retarded) time, signal power 1 ("linfmcw" and "cw" only)."""
        e = capi.bf_emitter()
        e.type, e.shape, e.radiance = capi.BF_TRANSMITTER_WIGNER, shape, 1.0
        e.to_world, e.to_object = _m16(np.eye(4)), _m16(np.eye(4))
        e.signal_type = {"cw": capi.BF_SIGNAL_CW, "pulse": capi.BF_SIGNAL_PULSE, "linfmcw": capi.BF_SIGNAL_LINFMCW}[signaltype]
        e.amplitude, e.freq_centre, e.freq_ext, e.pulse_len, e.prf, e.gain = amplitude, freq_centre, freq_ext, pulse_len, prf, gain
        e.resample_freq = int(bool(resample_freq))
        self.emitters.append(e)
        self.shapes[shape].emitter = len(self.emitters) - 1
        return len(self.emitters) - 1

    # ---- receivers + ADC (gen-3) ----
    def set_receiver(self, shape, kind="omnidirectional", adc_sampling_start=0.0, adc_sampling_end=0.0, t_bins=1024,
                     f_bins=1024, t_bandwidth=3.81e-6, f_bandwidth=250e6, freq_centre=1.0, freq_ext=1.0, gain=1.0,
                     sig_is_delta=False, array=None, rx_signaltype="cw", rx_chirp_len=0.0, rx_crf=0.0, rx_amplitude=1.0):
        """receiver.cpp:16-62 + adc.cpp:18-46 (box rfilter, full window).  rx_signaltype / rx_chirp_len / rx_crf: the local
        oscillator a Wigner / phased receiver samples its frequency from under receive_type "mix_resample"
        (wignerreceiver.cpp:72-110; freq_ext is then the sweep, sig_is_delta must be set)."""
        s = self.sensor
        s.type = {"omnidirectional": capi.BF_RECEIVER_OMNI, "wigner": capi.BF_RECEIVER_WIGNER,
                  "phased": capi.BF_RECEIVER_PHASED}[kind]
        if array is not None:
            s.array = array
        s.shape = shape
        self.shapes[shape].is_sensor = 1
        s.adc_sampling_start = adc_sampling_start
        s.adc_sampling_time = f32(adc_sampling_end) - f32(adc_sampling_start)
        s.t_bins, s.f_bins, s.t_bandwidth, s.f_bandwidth = t_bins, f_bins, t_bandwidth, f_bandwidth
        s.freq_centre, s.freq_ext, s.gain, s.rx_sig_is_delta = freq_centre, freq_ext, gain, int(sig_is_delta)
        s.rx_signal_type = {"cw": capi.BF_SIGNAL_CW, "pulse": capi.BF_SIGNAL_PULSE, "linfmcw": capi.BF_SIGNAL_LINFMCW}[rx_signaltype]
        s.rx_pulse_len, s.rx_prf, s.rx_amplitude = rx_chirp_len, rx_crf, rx_amplitude

    # ---- phased arrays (phasedtransmitter.cpp:108-165 == phasedreceiver.cpp:115-172) ----
    def phased_array(self, n_elems, elem_dims, elem_spacing, elem_axis, steering_vector=(0.0, 0.0, 0.0), array_loc=None):
        """The n_elems^2 virtual elements of the constructors: pairs (i, j) of physical elements placed at
        -spacing * axis * (i - (n - 1) / 2) around the array centre; per pair the transform of the virtual
        element (array_loc * translate((r_i + r_j) / 2) * scale(w_x / 2, w_y / 2, w_z)), its inverse, the frame
        built from it, r' = r_i - r_j and the steering phasor exp(j K (0 - r') . sin(steer)),
        K = 1 / ((lambda_max - lambda_min) * 1e-9 / 2)."""
        n = int(n_elems)
        loc = array_loc if array_loc is not None else Transform4f()
        wid = np.asarray(elem_dims, f32)
        spacing, axis = np.asarray(elem_spacing, f32), np.asarray(elem_axis, f32)
        steer = np.sin(np.asarray(steering_vector, f32)).astype(f32)
        step = (spacing * axis).astype(f32)
        locs = [(-(step * f32(i - (n / 2.0) + 0.5 if n % 2 == 0 else i - (n - 1.0) / 2.0))).astype(f32) for i in range(n)]
        k = f32(1.0 / ((np.float64(f32(self.physics.lambda_max_nm) - f32(self.physics.lambda_min_nm))) * 1e-9 / 2))
        tab = np.zeros((n * n, capi.BF_VELEM_FLOATS), f32)
        for i in range(n):
            for j in range(n):
                r_v = ((locs[i] + locs[j]) / f32(2)).astype(f32)
                r_dash = (locs[i] - locs[j]).astype(f32)
                t = loc * Transform4f.translate(r_v) * Transform4f.scale([wid[0] / f32(2), wid[1] / f32(2), wid[2]])
                m, inv = t.matrix.astype(f32), t.inv.astype(f32)
                dp_du, dp_dv = m[:3, :3] @ np.array([2, 0, 0], f32), m[:3, :3] @ np.array([0, 2, 0], f32)
                nrm = inv[:3, :3].T @ np.array([0, 0, 1], f32)          # normals transform with the inverse transpose
                cols = [v / np.linalg.norm(v) for v in (dp_du.astype(np.float64), dp_dv.astype(np.float64), nrm.astype(np.float64))]
                d2l = np.stack(cols, 0).astype(f32)                       # Transform::from_frame: ROWS s, t, n (transform.h:284-295)
                theta = f32(k * f32(np.dot((-r_dash).astype(f32), steer)))
                row = tab[i * n + j]
                row[0:12] = inv[:3, :4].reshape(-1)
                row[12:24] = np.concatenate([d2l, np.zeros((3, 1), f32)], 1).reshape(-1)
                row[24:27] = r_dash
                row[28], row[29] = np.cos(np.float64(theta)), np.sin(np.float64(theta))
        tab = np.ascontiguousarray(tab)
        self._keep.append(tab)
        a = capi.bf_phased_array()
        a.velems = tab.ctypes.data_as(C.POINTER(C.c_float))
        a.n_velems = n * n
        a.elem_dims = (C.c_float * 3)(*[float(x) for x in wid])
        return a

    def add_phased_transmitter(self, shape, array, signaltype="cw", amplitude=1.0, freq_centre=1.0, freq_ext=0.0, pulse_len=1.0,
                               prf=1.0, gain=1.0, resample_freq=False):
        """phasedtransmitter.cpp: the signal model of the Wigner transmitter + the array's Wigner function."""
        i = self.add_wigner_transmitter(shape, signaltype=signaltype, amplitude=amplitude, freq_centre=freq_centre,
                                        freq_ext=freq_ext, pulse_len=pulse_len, prf=prf, gain=gain, resample_freq=resample_freq)
        self.emitters[i].type = capi.BF_TRANSMITTER_PHASED
        self.emitters[i].array = array
        return i

    # ---- sensors ----
    def set_fluxmeter(self, shape):
        self.sensor.type, self.sensor.shape = capi.BF_SENSOR_FLUXMETER, shape
        self.shapes[shape].is_sensor = 1

    def set_radiancemeter(self, to_world):
        """radiancemeter.cpp: one ray from to_world * 0 along to_world * +z (a pencil beam), weight 1."""
        s = self.sensor
        s.type, s.shape = capi.BF_SENSOR_RADIANCEMETER, -1
        s.film_width = s.film_height = 1
        s.to_world = _m16(to_world.matrix)

    def set_irradiancemeter(self, shape):
        """irradiancemeter.cpp: the flux meter's rays, weighted by pi / surface_area."""
        self.sensor.type, self.sensor.shape = capi.BF_SENSOR_IRRADIANCEMETER, shape
        self.shapes[shape].is_sensor = 1

    def set_perspective(self, to_world, fov=45.0, near_clip=0.01, far_clip=10000.0, film=(1, 1), crop=None):
        """film = (width, height) in pixels; fov is the horizontal field of view (fov_axis "x"); crop = (offset_x, offset_y,
        width, height): the film's crop window (film.cpp:17-27) — the launch then names the crop size."""
        s = self.sensor
        s.type, s.shape = capi.BF_SENSOR_PERSPECTIVE, -1
        ox, oy, cw, ch = crop if crop is not None else (0, 0, int(film[0]), int(film[1]))
        s.film_width, s.film_height = int(cw), int(ch)
        s.crop_offset_x, s.crop_offset_y = int(ox), int(oy)
        s.to_world = _m16(to_world.matrix)
        fw, fh = f32(film[0]), f32(film[1])
        s.sample_to_camera = _m16(perspective_sample_to_camera(fov, near_clip, far_clip, film[0] / film[1], (f32(cw) / fw, f32(ch) / fh),
                                                               (f32(ox) / fw, f32(oy) / fh)))
        s.fov_x_deg, s.near_clip, s.far_clip = fov, near_clip, far_clip

    def finalize(self):
        d = capi.bf_scene_desc()
        self._shapes_arr = (capi.bf_shape * max(1, len(self.shapes)))(*self.shapes)
        self._mat_arr = (capi.bf_material * max(1, len(self.materials)))(*self.materials)
        self._em_arr = (capi.bf_emitter * max(1, len(self.emitters)))(*self.emitters)
        d.shapes, d.n_shapes = self._shapes_arr, len(self.shapes)
        d.materials, d.n_materials = self._mat_arr, len(self.materials)
        d.emitters, d.n_emitters = self._em_arr, len(self.emitters)
        d.sensor, d.physics = self.sensor, self.physics
        self.desc = d
        return self
