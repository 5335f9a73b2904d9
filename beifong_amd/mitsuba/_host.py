"""ctypes binding of libbeifong_host.so (beifong_amd/host/capi.cpp)."""
import ctypes as C
import os

import numpy as np

from .. import capi

_HOST_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "host")
LIB_PATH = os.path.join(_HOST_DIR, "libbeifong_host.so")
_lib = None


class HostError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HostError(f"{LIB_PATH} not found: run __graft_entry__.build() first")
    capi.load_library()            # libbeifong_hip.so first (same instance for both)
    l = C.CDLL(LIB_PATH)
    vp, cp = C.c_void_p, C.c_char_p
    # ABI handshake: the host library embeds bf_launch / bf_stats (Integrator::m_stats) and hands them to this binding
    for f in ("bfh_abi_version", "bfh_sizeof_launch", "bfh_sizeof_stats"):
        if not hasattr(l, f):
            raise HostError(f"{LIB_PATH} is stale (no {f}): rebuild the host layer (make -C beifong_amd/host)")
    if (l.bfh_abi_version() != capi.BF_ABI_VERSION or l.bfh_sizeof_launch() != C.sizeof(capi.bf_launch)
            or l.bfh_sizeof_stats() != C.sizeof(capi.bf_stats)):
        raise HostError(f"{LIB_PATH}: ABI version {l.bfh_abi_version()}, sizeof(bf_launch) {l.bfh_sizeof_launch()}, sizeof(bf_stats) "
                        f"{l.bfh_sizeof_stats()}; this binding has {capi.BF_ABI_VERSION}, {C.sizeof(capi.bf_launch)}, "
                        f"{C.sizeof(capi.bf_stats)} — rebuild the host layer (make -C beifong_amd/host)")
    l.bfh_last_error.restype = cp
    l.bfh_variant.restype = cp
    l.bfh_set_variant.argtypes = [cp]
    l.bfh_load_file.argtypes = [cp, C.c_int, C.POINTER(cp), C.POINTER(cp), C.POINTER(vp)]
    l.bfh_load_string.argtypes = [cp, cp, C.c_int, C.POINTER(cp), C.POINTER(cp), C.POINTER(vp)]
    l.bfh_release.argtypes = [vp]
    l.bfh_class_name.argtypes = [vp]
    l.bfh_class_name.restype = cp
    l.bfh_scene_counts.argtypes = [vp] + [C.POINTER(C.c_int)] * 5
    for f in ("bfh_scene_integrator",):
        getattr(l, f).argtypes = [vp]
        getattr(l, f).restype = vp
    for f in ("bfh_scene_sensor", "bfh_scene_receiver", "bfh_scene_shape"):
        getattr(l, f).argtypes = [vp, C.c_int]
        getattr(l, f).restype = vp
    l.bfh_shape_info.argtypes = [vp, C.POINTER(C.c_uint), C.POINTER(C.c_float)]
    l.bfh_scene_flat_desc.argtypes = [vp, vp]
    l.bfh_scene_flat_desc.restype = C.POINTER(capi.bf_scene_desc)
    l.bfh_integrator_launch.argtypes = [vp, vp, C.POINTER(capi.bf_launch)]
    l.bfh_integrator_render.argtypes = [vp, vp, vp]
    l.bfh_integrator_receive.argtypes = [vp, vp, vp]
    l.bfh_integrator_stats.argtypes = [vp, C.POINTER(capi.bf_stats), C.POINTER(C.c_double)]
    l.bfh_sensor_sample_count.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    l.bfh_develop.argtypes = [vp, cp]
    l.bfh_write_exr.argtypes = [cp, C.c_uint, C.c_uint, C.c_uint, C.POINTER(cp), C.POINTER(C.c_float)]
    l.bfh_bitmap.argtypes = [vp, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    l.bfh_channel_name.argtypes = [vp, C.c_uint]
    l.bfh_channel_name.restype = cp
    l.bfh_film_geometry.argtypes = [vp, C.POINTER(C.c_uint)]
    l.bfh_rfilter_eval.argtypes = [vp, C.c_float, C.c_int, C.POINTER(C.c_float)]
    l.bfh_rfilter_flatten.argtypes = [vp, C.c_uint, C.POINTER(capi.bf_rfilter)]
    l.bfh_loaded_plugins.argtypes = [C.c_char_p, C.c_int]
    _lib = l
    return l


def check(status):
    if status != 0:
        raise HostError(lib().bfh_last_error().decode())


def params(kwargs):
    n = len(kwargs)
    keys = (C.c_char_p * max(n, 1))(*[str(k).encode() for k in kwargs])
    vals = (C.c_char_p * max(n, 1))(*[str(v).encode() for v in kwargs.values()])
    return n, keys, vals


class Struct:
    """mitsuba.core.Struct: only the component types Bitmap.convert is asked for in the reference's scripts."""

    class Type:
        UInt8, Float16, Float32, Float64 = "uint8", "float16", "float32", "float64"


class Bitmap:
    """mitsuba.core.Bitmap, as far as the reference's radar scripts use it: the raw float32 storage of a film / ADC
    (np.array(bitmap) gives [rows, cols, channels]; film.bitmap(raw=True)), construction from an array with a pixel format,
    convert() to RGB / luminance with optional sRGB gamma, write() as .exr / .npy / .png (src/libcore/bitmap.cpp; no libjpeg
    here: a '.jpg' destination is written as PNG next to it, with a note on stderr)."""

    class PixelFormat:
        Y, YA, RGB, RGBA, XYZ, XYZA, XYZAW, MultiChannel = "Y", "YA", "RGB", "RGBA", "XYZ", "XYZA", "XYZAW", "MultiChannel"

    _CHANNELS = {"Y": ["Y"], "YA": ["Y", "A"], "RGB": ["R", "G", "B"], "RGBA": ["R", "G", "B", "A"], "XYZ": ["X", "Y", "Z"],
                 "XYZA": ["X", "Y", "Z", "A"], "XYZAW": ["X", "Y", "Z", "A", "W"]}
    _XYZ_TO_RGB = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]],
                           dtype=np.float32)        # include/mitsuba/core/spectrum.h:289-295

    def __init__(self, arr, names=None):
        arr = np.asarray(arr)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        self._arr = arr
        if isinstance(names, str):                        # a PixelFormat
            self._format = names
            names = Bitmap._CHANNELS.get(names)
            if names is None or len(names) != arr.shape[2]:
                raise ValueError(f"Bitmap: pixel format {self._format} does not match {arr.shape[2]} channels")
        else:
            self._format = Bitmap.PixelFormat.MultiChannel
            if names is None:
                names = [f"ch{i}" for i in range(arr.shape[2])]
        self._names = list(names)

    def __array__(self, dtype=None, copy=None):
        return self._arr if dtype is None else self._arr.astype(dtype)

    def channel_names(self):
        return list(self._names)

    def size(self):
        return (self._arr.shape[1], self._arr.shape[0])

    def width(self):
        return self._arr.shape[1]

    def height(self):
        return self._arr.shape[0]

    def channel_count(self):
        return self._arr.shape[2]

    def pixel_format(self):
        return self._format

    def convert(self, pixel_format, component_format=Struct.Type.Float32, srgb_gamma=False):
        """Bitmap::convert for the colour formats: XYZ[A[W]] / RGB[A] / Y[A] -> RGB / RGBA / Y / XYZ; W divides the colour
        channels (normalisation by the accumulated weight) as the reference's struct converter does."""
        a = self._arr.astype(np.float32)
        f = self._format
        alpha = None
        if f in ("XYZAW",):
            w = a[:, :, 4:5]
            a = np.concatenate([np.where(w != 0, a[:, :, :3] / np.where(w != 0, w, 1), 0), a[:, :, 3:4]], axis=2)
            f = "XYZA"
        if f in ("XYZA", "RGBA", "YA"):
            alpha = a[:, :, -1:]
            a = a[:, :, :-1]
            f = f[:-1]
        if f == "MultiChannel":
            raise ValueError("Bitmap.convert: a multi-channel bitmap has no colour interpretation; pick channels first")
        if f == "Y":
            xyz = np.concatenate([a * np.float32(0.950456), a, a * np.float32(1.08875)], axis=2)
        elif f == "RGB":
            xyz = a @ np.linalg.inv(Bitmap._XYZ_TO_RGB).T.astype(np.float32)
        else:
            xyz = a
        target = pixel_format[:-1] if pixel_format in ("RGBA", "XYZA", "YA") else pixel_format
        if target == "RGB":
            out = xyz @ Bitmap._XYZ_TO_RGB.T
        elif target == "XYZ":
            out = xyz
        elif target == "Y":
            out = xyz[:, :, 1:2]
        else:
            raise ValueError(f"Bitmap.convert: unsupported target format {pixel_format}")
        if srgb_gamma:
            o = np.clip(out, 0, None)
            out = np.where(o <= 0.0031308, 12.92 * o, 1.055 * np.power(o, 1 / 2.4) - 0.055)
        if pixel_format in ("RGBA", "XYZA", "YA"):
            out = np.concatenate([out, alpha if alpha is not None else np.ones_like(out[:, :, :1])], axis=2)
        if component_format == Struct.Type.UInt8:
            out = np.clip(np.rint(out * 255.0), 0, 255).astype(np.uint8)
        else:
            out = out.astype(component_format)
        return Bitmap(out, pixel_format)

    def write(self, path):
        path = str(path)
        ext = os.path.splitext(path)[1].lower()
        if ext == ".npy":
            np.save(path, self._arr)
        elif ext == ".exr":
            write_exr(path, self._arr.astype(np.float32), self._names)
        else:
            if ext in (".jpg", ".jpeg"):
                import sys
                print(f"[beifong] no JPEG encoder in this build: writing {path}.png instead", file=sys.stderr)
                path += ".png"
            elif ext != ".png":
                raise ValueError(f"Bitmap.write: unsupported file format '{ext}'")
            _write_png(path, self._arr)


def _write_png(path, arr):
    """8-bit grey / grey+alpha / RGB / RGBA PNG (zlib + CRC32 from the standard library)."""
    import struct
    import zlib
    a = np.asarray(arr)
    if a.dtype != np.uint8:
        a = np.clip(np.rint(a.astype(np.float32) * 255.0), 0, 255).astype(np.uint8)
    h, w, c = a.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}.get(c)
    if ctype is None:
        raise ValueError("PNG needs 1-4 channels")
    raw = b"".join(b"\x00" + a[r].tobytes() for r in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


class _Handle:
    def __init__(self, ptr, owner=None, owned=False):
        self._ptr = C.c_void_p(ptr) if not isinstance(ptr, C.c_void_p) else ptr
        self._owner = owner        # keeps the parent (scene) alive
        self._owned = owned

    def __del__(self):
        try:
            if self._owned and self._ptr:
                lib().bfh_release(self._ptr)
                self._ptr = None
        except Exception:
            pass

    def class_name(self):
        return lib().bfh_class_name(self._ptr).decode()

    def __repr__(self):
        return f"<{self.class_name()}>"


class Sampler:
    def __init__(self, endpoint):
        self._e = endpoint

    def sample_count(self):
        n = C.c_ulonglong()
        check(lib().bfh_sensor_sample_count(self._e._ptr, C.byref(n)))
        return n.value


class _Storage:
    def __init__(self, endpoint):
        self._e = endpoint

    def bitmap(self, raw=True):
        data = C.POINTER(C.c_float)()
        r, c, ch = C.c_uint(), C.c_uint(), C.c_uint()
        check(lib().bfh_bitmap(self._e._ptr, C.byref(data), C.byref(r), C.byref(c), C.byref(ch)))
        arr = np.ctypeslib.as_array(data, shape=(r.value, c.value, ch.value)).copy()
        names = [lib().bfh_channel_name(self._e._ptr, i).decode() for i in range(ch.value)]
        return Bitmap(arr, names)

    def set_destination_file(self, path):
        self._dest = str(path)

    def develop(self):
        """Write the raw storage as a multi-channel float32 OpenEXR file (hdrfilm.cpp:213-249, hdradc.cpp:259-295)."""
        dest = getattr(self, "_dest", None)
        if dest is None:
            raise RuntimeError("develop(): call set_destination_file() first")
        check(lib().bfh_develop(self._e._ptr, dest.encode()))

    def _geometry(self):
        if isinstance(self._e, Sensor):
            g = (C.c_uint * 6)()
            check(lib().bfh_film_geometry(self._e._ptr, g))
            return list(g)
        w, h = self.bitmap().size()              # an ADC: its window is what it stores
        return [w, h, w, h, 0, 0]

    def size(self):
        """Film::size(): the full film (film.cpp:10-14)."""
        g = self._geometry()
        return (g[0], g[1])

    def crop_size(self):
        g = self._geometry()
        return (g[2], g[3])

    def crop_offset(self):
        g = self._geometry()
        return (g[4], g[5])


class Sensor(_Handle):
    def film(self):
        return _Storage(self)

    def sampler(self):
        return Sampler(self)


class Receiver(_Handle):
    def adc(self):
        return _Storage(self)

    def sampler(self):
        return Sampler(self)


class Shape(_Handle):
    def primitive_count(self):
        p, a = C.c_uint(), C.c_float()
        check(lib().bfh_shape_info(self._ptr, C.byref(p), C.byref(a)))
        return p.value

    def surface_area(self):
        p, a = C.c_uint(), C.c_float()
        check(lib().bfh_shape_info(self._ptr, C.byref(p), C.byref(a)))
        return a.value


def _bound(obj):
    """An object made by load_dict outside a scene stands for its instance in the scene that was loaded with it last."""
    return obj._resolve() if hasattr(obj, "_resolve") else obj


class Integrator(_Handle):
    def render(self, scene, sensor):
        check(lib().bfh_integrator_render(self._ptr, scene._ptr, _bound(sensor)._ptr))
        return True

    def receive(self, scene, receiver):
        check(lib().bfh_integrator_receive(self._ptr, scene._ptr, _bound(receiver)._ptr))
        return True

    def launch_for(self, endpoint):
        lp = capi.bf_launch()
        check(lib().bfh_integrator_launch(self._ptr, endpoint._ptr, C.byref(lp)))
        return lp

    def stats(self):
        st, ms = capi.bf_stats(), C.c_double()
        check(lib().bfh_integrator_stats(self._ptr, C.byref(st), C.byref(ms)))
        return st, ms.value


class Scene(_Handle):
    def _counts(self):
        v = [C.c_int() for _ in range(5)]
        check(lib().bfh_scene_counts(self._ptr, *[C.byref(x) for x in v]))
        return [x.value for x in v]

    def integrator(self):
        return Integrator(lib().bfh_scene_integrator(self._ptr), owner=self)

    def sensors(self):
        return [Sensor(lib().bfh_scene_sensor(self._ptr, i), owner=self) for i in range(self._counts()[1])]

    def receivers(self):
        return [Receiver(lib().bfh_scene_receiver(self._ptr, i), owner=self) for i in range(self._counts()[2])]

    def shapes(self):
        return [Shape(lib().bfh_scene_shape(self._ptr, i), owner=self) for i in range(self._counts()[0])]

    def flat_desc(self, endpoint):
        """The bf_scene_desc the integrator hands to the HIP library (tests feed
        the same pointer to the CPU oracle)."""
        p = lib().bfh_scene_flat_desc(self._ptr, endpoint._ptr)
        if not p:
            raise HostError(lib().bfh_last_error().decode())

        class Holder:
            pass

        h = Holder()
        h.desc = p.contents
        h._scene = self
        return h


class ReconstructionFilter(_Handle):
    """mitsuba.core.ReconstructionFilter (src/libcore/python/rfilter.cpp): eval / eval_discretized / radius / border_size."""

    def _eval(self, x, discretized):
        out = C.c_float()
        check(lib().bfh_rfilter_eval(self._ptr, float(x), discretized, C.byref(out)))
        return out.value

    def eval(self, x):
        return self._eval(x, 0)

    def eval_discretized(self, x):
        return self._eval(x, 1)

    def flatten(self, block_size=0):
        """The filter as the C ABI carries it (capi.bf_rfilter: goes into bf_sensor.rfilter)."""
        f = capi.bf_rfilter()
        check(lib().bfh_rfilter_flatten(self._ptr, block_size, C.byref(f)))
        return f

    def radius(self):
        return self.flatten().radius

    def border_size(self):
        return self.flatten().border


def wrap(ptr, dict_src=None):
    name = lib().bfh_class_name(ptr).decode()
    cls = {"Scene": Scene, "ReconstructionFilter": ReconstructionFilter}.get(name, _Handle)
    o = cls(ptr, owned=True)
    o._dict = dict_src
    return o


def write_exr(path, array, channel_names):
    """Write float32 [rows, cols, channels] as an uncompressed multi-channel OpenEXR file."""
    a = np.ascontiguousarray(array, dtype=np.float32)
    if a.ndim != 3 or a.shape[2] != len(channel_names):
        raise ValueError("write_exr: array must be [rows, cols, len(channel_names)]")
    names = (C.c_char_p * len(channel_names))(*[n.encode() for n in channel_names])
    check(lib().bfh_write_exr(str(path).encode(), a.shape[1], a.shape[0], a.shape[2], names,
                              a.ctypes.data_as(C.POINTER(C.c_float))))


def read_exr(path):
    """Minimal reader for the files write_exr produces (single part, scanline, no compression, FLOAT channels):
    returns (float32 [rows, cols, channels], channel names in file order)."""
    import struct
    b = open(path, "rb").read()
    if b[:4] != b"\x76\x2f\x31\x01" or struct.unpack_from("<i", b, 4)[0] != 2:
        raise ValueError("not a plain OpenEXR 2 scanline file")
    pos, attrs = 8, {}
    while b[pos] != 0:
        e = b.index(b"\0", pos)
        name = b[pos:e].decode()
        e2 = b.index(b"\0", e + 1)
        size = struct.unpack_from("<i", b, e2 + 1)[0]
        attrs[name] = b[e2 + 5:e2 + 5 + size]
        pos = e2 + 5 + size
    pos += 1
    if attrs["compression"] != b"\0":
        raise ValueError("compressed EXR not supported")
    names, p, ch = [], 0, attrs["channels"]
    while ch[p] != 0:
        e = ch.index(b"\0", p)
        names.append(ch[p:e].decode())
        if struct.unpack_from("<i", ch, e + 1)[0] != 2:
            raise ValueError("only FLOAT channels supported")
        p = e + 1 + 16
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    offs = struct.unpack_from("<%dQ" % h, b, pos)
    out = np.empty((h, w, len(names)), np.float32)
    for y in range(h):
        yy, nbytes = struct.unpack_from("<2i", b, offs[y])
        line = np.frombuffer(b, np.float32, w * len(names), offs[y] + 8).reshape(len(names), w)
        out[yy - y0] = line.T
    return out, names
