"""ctypes loader for oracle/libbf_oracle.so (the CPU checker; test infrastructure)."""
import ctypes as C
import os
import subprocess

import numpy as np

from beifong_amd import capi

ORACLE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
LIB = os.path.join(ORACLE_DIR, "libbf_oracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def load():
    global _lib
    if _lib is not None:
        return _lib
    build()          # make: a no-op unless bf_oracle.cpp or include/beifong_hip.h is newer than the library
    _lib = load_from(LIB)
    return _lib


def load_from(path):
    """Bind an oracle build (the checker's -O2 library, or bench.py's -O3 -march=native timing build of the same source)."""
    lib = C.CDLL(path)
    vp = C.c_void_p
    # the checker shares the C-ABI structs (bf_launch, bf_stats, bf_scene_desc ...) with the product: same header or bust
    lib.bfo_abi_fingerprint.restype = C.c_uint64
    prod = capi.load_library()
    prod.bf_abi_fingerprint.restype = C.c_uint64
    if lib.bfo_abi_fingerprint() != prod.bf_abi_fingerprint():
        raise RuntimeError(f"{path} was built against another include/beifong_hip.h than libbeifong_hip.so: make -C oracle")
    lib.bfo_last_error.restype = C.c_char_p
    lib.bfo_scene_create.argtypes = [C.POINTER(capi.bf_scene_desc), C.c_int, C.POINTER(vp)]
    lib.bfo_scene_destroy.argtypes = [vp]
    lib.bfo_launch_channels.argtypes = [C.POINTER(capi.bf_launch)]
    lib.bfo_launch_channels.restype = C.c_uint32
    lib.bfo_render.argtypes = [vp, C.POINTER(capi.bf_launch), C.c_int, C.c_int, vp, vp, C.POINTER(capi.bf_stats)]
    lib.bfo_trace_closest.argtypes = [vp, C.c_uint64, vp, vp, vp, vp, vp]
    lib.bfo_trace_any.argtypes = [vp, C.c_uint64, vp, vp]
    lib.bfo_ray_intersect_full.argtypes = [vp, vp, vp]
    lib.bfo_tea_float32.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
    lib.bfo_tea_float32.restype = C.c_float
    lib.bfo_tea_float64.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
    lib.bfo_tea_float64.restype = C.c_double
    lib.bfo_pcg32_u32.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, vp]
    lib.bfo_sampler_floats.argtypes = [C.c_uint64, C.c_uint32, vp]
    for name in ("bfo_square_to_uniform_disk_concentric", "bfo_square_to_cosine_hemisphere"):
        getattr(lib, name).argtypes = [C.c_float, C.c_float, vp]
    lib.bfo_square_to_uniform_cone.argtypes = [C.c_float, C.c_float, C.c_float, vp]
    lib.bfo_coordinate_system.argtypes = [vp, vp, vp]
    lib.bfo_frame_from_normal.argtypes = [vp, vp, vp]
    lib.bfo_frame_from_normal.restype = None
    lib.bfo_bsdf_eval.argtypes = [C.POINTER(capi.bf_material), vp, vp]
    lib.bfo_bsdf_eval.restype = C.c_float
    lib.bfo_bsdf_pdf.argtypes = [C.POINTER(capi.bf_material), vp, vp]
    lib.bfo_bsdf_pdf.restype = C.c_float
    lib.bfo_bsdf_sample.argtypes = [C.POINTER(capi.bf_material), vp, C.c_float, C.c_float, C.c_float, vp, vp]
    lib.bfo_bsdf_sample.restype = C.c_float
    lib.bfo_bsdf_sample_n.argtypes = [C.POINTER(capi.bf_material), vp, C.c_uint64, vp, vp, vp]
    lib.bfo_bsdf_sample_n.restype = None
    lib.bfo_bsdf_pdf_n.argtypes = [C.POINTER(capi.bf_material), vp, C.c_uint64, vp, vp]
    lib.bfo_bsdf_pdf_n.restype = None
    lib.bfo_fresnel_conductor.argtypes = [C.c_float, C.c_float, C.c_float]
    lib.bfo_fresnel_conductor.restype = C.c_float
    lib.bfo_direction_sample.argtypes = [vp, vp, vp]
    lib.bfo_direction_sample.restype = None
    lib.bfo_erfinv.argtypes = [C.c_float]
    lib.bfo_erfinv.restype = C.c_float
    lib.bfo_elementary.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.bfo_elementary.restype = None
    lib.bfo_rect_area.argtypes = [vp, C.c_uint32]
    lib.bfo_rect_area.restype = C.c_float
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleScene:
    def __init__(self, holder, brute_force=False, accel=None, lib=None):
        """accel: 0 median-split BVH (default), 1 brute force (== brute_force=True), 2 binned-SAH BVH (the CPU baseline's)."""
        self.lib = lib or load()
        self.holder = holder
        h = C.c_void_p()
        st = self.lib.bfo_scene_create(C.byref(holder.desc), int(brute_force) if accel is None else int(accel), C.byref(h))
        if st != 0:
            raise RuntimeError(f"bfo_scene_create failed: {self.lib.bfo_last_error().decode()}")
        self.handle = h

    def __del__(self):
        try:
            if self.handle:
                self.lib.bfo_scene_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def render(self, launch, rng_mode=0, threads=1, records=False):
        n = self.lib.bfo_launch_channels(C.byref(launch))
        hist = np.zeros(n, np.float32)
        rec = np.zeros(launch.n_paths, capi.PATH_RECORD_DTYPE) if records else None
        st = capi.bf_stats()
        s = self.lib.bfo_render(self.handle, C.byref(launch), rng_mode, threads, _ptr(hist), _ptr(rec), C.byref(st))
        if s != 0:
            raise RuntimeError(f"bfo_render failed: {self.lib.bfo_last_error().decode()}")
        return hist, rec, st

    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        n = rays.shape[0]
        t = np.empty(n, np.float32)
        prim = np.empty(n, np.uint32)
        shape = np.empty(n, np.uint32)
        uv = np.empty((n, 2), np.float32)
        self.lib.bfo_trace_closest(self.handle, n, _ptr(rays), _ptr(t), _ptr(prim), _ptr(shape), _ptr(uv))
        return t, prim, shape, uv

    def emitter_sample_direction(self, index, ref_p, sample=(0.0, 0.0)):
        out = np.zeros(8, np.float32)
        p = np.ascontiguousarray(ref_p, np.float32)
        self.lib.bfo_emitter_sample_direction.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        assert self.lib.bfo_emitter_sample_direction(self.handle, index, _ptr(p), sample[0], sample[1], _ptr(out)) == 0
        return dict(d=out[0:3], dist=out[3], pdf=out[4], delta=bool(out[5]), spec=out[6], pdf_direction=out[7])

    def sensor_sample_ray(self, fx, fy, ax=0.5, ay=0.5):
        out = np.zeros(8, np.float32)
        self.lib.bfo_sensor_sample_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
        self.lib.bfo_sensor_sample_ray(self.handle, fx, fy, ax, ay, _ptr(out))
        return dict(o=out[0:3], mint=out[3], d=out[4:7], weight=out[7])

    def trace_any(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hit = np.empty(rays.shape[0], np.uint8)
        self.lib.bfo_trace_any(self.handle, rays.shape[0], _ptr(rays), _ptr(hit))
        return hit

    def intersect_full(self, ray):
        ray = np.ascontiguousarray(ray, np.float32).reshape(8)
        out = np.zeros(27, np.float32)
        self.lib.bfo_ray_intersect_full(self.handle, _ptr(ray), _ptr(out))
        return dict(t=out[0], p=out[1:4], n=out[4:7], sh_n=out[7:10], sh_s=out[10:13], sh_t=out[13:16],
                    wi=out[16:19], prim_uv=out[19:21], dp_du=out[21:24], dp_dv=out[24:27])
