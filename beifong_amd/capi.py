"""ctypes mirror of include/beifong_hip.h and loader for libbeifong_hip.so.

This is plumbing: the product is the HIP library behind the C ABI.  Loading
fails loudly when the extension is missing — there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libbeifong_hip.so")

BF_OK, BF_ERR_INVALID, BF_ERR_DEVICE, BF_ERR_NOMEM, BF_ERR_UNSUPPORTED = range(5)
BF_BSDF_DIFFUSE, BF_BSDF_ROUGHCONDUCTOR, BF_BSDF_NULL = range(3)
BF_MF_BECKMANN, BF_MF_GGX = range(2)
BF_SHAPE_RECTANGLE, BF_SHAPE_MESH = range(2)
BF_EMITTER_SPOT, BF_EMITTER_AREA, BF_TRANSMITTER_AREA, BF_TRANSMITTER_WIGNER, BF_TRANSMITTER_PHASED, BF_EMITTER_POINT = range(6)
BF_SIGNAL_CW, BF_SIGNAL_PULSE, BF_SIGNAL_LINFMCW = range(3)
BF_SENSOR_FLUXMETER, BF_SENSOR_PERSPECTIVE, BF_RECEIVER_OMNI, BF_RECEIVER_WIGNER, BF_RECEIVER_PHASED, BF_SENSOR_IRRADIANCEMETER, BF_SENSOR_RADIANCEMETER = range(7)
BF_ABI_VERSION = 4          # include/beifong_hip.h: BF_ABI_VERSION
BF_VELEM_FLOATS = 32
BF_SI_FLOATS = 27
BF_MODE_PATH, BF_MODE_RANGE, BF_MODE_TIME, BF_MODE_RECEIVE_RAW, BF_MODE_RECEIVE_IQ = range(5)
BF_COLOR_RGB, BF_COLOR_MONO = range(2)
BF_FLAG_STATS, BF_FLAG_GLOBAL_ATOMICS, BF_FLAG_MEGAKERNEL, BF_FLAG_DOPPLER, BF_FLAG_MIX_RESAMPLE = 1, 2, 4, 8, 16
BF_FLAG_ROLLING, BF_FLAG_TIMING, BF_FLAG_COUNT = 32, 64, 128

M16 = C.c_float * 16


class bf_material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("twosided", C.c_uint32), ("reflectance", C.c_float),
                ("alpha_u", C.c_float), ("alpha_v", C.c_float), ("distribution", C.c_uint32),
                ("sample_visible", C.c_uint32), ("eta", C.c_float), ("k", C.c_float),
                ("has_specular_reflectance", C.c_uint32), ("back_material", C.c_uint32)]


class bf_shape(C.Structure):
    _fields_ = [("type", C.c_uint32), ("material", C.c_uint32), ("emitter", C.c_int32),
                ("is_sensor", C.c_uint32), ("to_world", M16), ("to_object", M16),
                ("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
                ("texcoords", C.POINTER(C.c_float)), ("indices", C.POINTER(C.c_uint32)), ("n_vertices", C.c_uint32), ("n_faces", C.c_uint32),
                ("velocity", M16)]


class bf_phased_array(C.Structure):
    _fields_ = [("velems", C.POINTER(C.c_float)), ("n_velems", C.c_uint32), ("elem_dims", C.c_float * 3)]


class bf_emitter(C.Structure):
    _fields_ = [("type", C.c_uint32), ("shape", C.c_int32), ("to_world", M16), ("to_object", M16),
                ("radiance", C.c_float), ("cutoff_angle_deg", C.c_float), ("beam_width_deg", C.c_float),
                ("signal_type", C.c_uint32), ("amplitude", C.c_float), ("freq_centre", C.c_float),
                ("freq_ext", C.c_float), ("pulse_len", C.c_float), ("prf", C.c_float), ("gain", C.c_float),
                ("resample_freq", C.c_uint32), ("array", bf_phased_array)]


BF_FILTER_RESOLUTION = 31
BF_VARIANT_LEAN, BF_VARIANT_WIDE = 1, 2


class bf_rfilter(C.Structure):
    _fields_ = [("radius", C.c_float), ("scale", C.c_float), ("border", C.c_uint32), ("block_size", C.c_uint32),
                ("values", C.c_float * (BF_FILTER_RESOLUTION + 1))]


class bf_sensor(C.Structure):
    _fields_ = [("type", C.c_uint32), ("shape", C.c_int32), ("to_world", M16), ("sample_to_camera", M16),
                ("fov_x_deg", C.c_float), ("near_clip", C.c_float), ("far_clip", C.c_float),
                ("film_width", C.c_uint32), ("film_height", C.c_uint32),
                ("shutter_open", C.c_float), ("shutter_open_time", C.c_float),
                ("adc_sampling_start", C.c_float), ("adc_sampling_time", C.c_float),
                ("t_bins", C.c_uint32), ("f_bins", C.c_uint32),
                ("t_bandwidth", C.c_float), ("f_bandwidth", C.c_float),
                ("freq_centre", C.c_float), ("freq_ext", C.c_float), ("gain", C.c_float), ("rx_sig_is_delta", C.c_uint32),
                ("array", bf_phased_array), ("rfilter", bf_rfilter),
                ("window_offset_t", C.c_uint32), ("window_offset_f", C.c_uint32), ("window_t_bins", C.c_uint32), ("window_f_bins", C.c_uint32),
                ("crop_offset_x", C.c_uint32), ("crop_offset_y", C.c_uint32),
                ("rx_signal_type", C.c_uint32), ("rx_pulse_len", C.c_float), ("rx_prf", C.c_float), ("rx_amplitude", C.c_float)]


class bf_physics(C.Structure):
    _fields_ = [("c", C.c_float), ("lambda_min_nm", C.c_float), ("lambda_max_nm", C.c_float)]


class bf_scene_desc(C.Structure):
    _fields_ = [("shapes", C.POINTER(bf_shape)), ("n_shapes", C.c_uint32),
                ("materials", C.POINTER(bf_material)), ("n_materials", C.c_uint32),
                ("emitters", C.POINTER(bf_emitter)), ("n_emitters", C.c_uint32),
                ("sensor", bf_sensor), ("physics", bf_physics)]


class bf_launch(C.Structure):
    _fields_ = [("mode", C.c_uint32), ("color_mode", C.c_uint32), ("n_paths", C.c_uint64),
                ("path_offset", C.c_uint64), ("seed", C.c_uint64), ("max_depth", C.c_int32),
                ("rr_depth", C.c_int32), ("bins", C.c_uint32), ("bins_y", C.c_uint32), ("bin_width", C.c_float),
                ("time_c", C.c_float), ("flags", C.c_uint32), ("phase_bins", C.c_uint32),
                ("film_width", C.c_uint32), ("film_height", C.c_uint32), ("spp", C.c_uint32)]


class bf_path_record(C.Structure):
    _fields_ = [("L", C.c_float), ("aux", C.c_float), ("valid", C.c_uint32), ("n_rays", C.c_uint32)]


PATH_RECORD_DTYPE = np.dtype([("L", "<f4"), ("aux", "<f4"), ("valid", "<u4"), ("n_rays", "<u4")])


class bf_stats(C.Structure):
    _fields_ = [("n_paths", C.c_uint64), ("n_rays_closest", C.c_uint64), ("n_rays_shadow", C.c_uint64),
                ("n_nodes_visited", C.c_uint64), ("n_tris_tested", C.c_uint64), ("n_invalid", C.c_uint64),
                ("n_bounces", C.c_uint64), ("kernel_ms", C.c_float), ("trace_ms", C.c_float),
                ("shade_ms", C.c_float), ("tail_ms", C.c_float), ("n_launches_trace", C.c_uint32),
                ("n_bounce_iters", C.c_uint32), ("n_rays_tail", C.c_uint64), ("n_rays_traced", C.c_uint64),
                ("n_nodes_lds", C.c_uint64), ("n_nodes_tail", C.c_uint64), ("n_wnodes_tail", C.c_uint64),
                ("n_tris_tail", C.c_uint64), ("n_bounces_tail", C.c_uint64), ("n_shade_loads", C.c_uint64),
                ("n_shade_stores", C.c_uint64), ("n_shade_shadow", C.c_uint64), ("n_shade_rays", C.c_uint64),
                ("n_guard", C.c_uint64), ("n_launches_tail", C.c_uint32), ("n_launches_shade", C.c_uint32),
                ("kernel_variant", C.c_uint32), ("reserved_", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class bf_scene_info(C.Structure):
    _fields_ = [("n_shapes", C.c_uint32), ("n_rects", C.c_uint32), ("n_triangles", C.c_uint32),
                ("n_bvh_nodes", C.c_uint32), ("node_bytes", C.c_uint32), ("tri_bytes", C.c_uint32),
                ("device_bytes", C.c_uint64), ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3),
                ("bvh_depth", C.c_uint32), ("bvh_stack_need", C.c_uint32), ("trace_node_bytes", C.c_uint32), ("device", C.c_int32)]


class bf_batch(C.Structure):
    _fields_ = [("n_renders", C.c_uint32), ("seeds", C.POINTER(C.c_uint64)), ("mesh_offsets", C.POINTER(C.c_float))]


# include/beifong_hip.h: BF_ABI_MATERIAL .. BF_ABI_BATCH, in that order
ABI_STRUCTS = [bf_material, bf_shape, bf_emitter, bf_sensor, bf_scene_desc, bf_launch, bf_path_record, bf_stats, bf_scene_info, bf_batch]

# every symbol include/beifong_hip.h declares
EXPORTED_SYMBOLS = [
    "bf_abi_sizeof", "bf_abi_fingerprint", "bf_version", "bf_last_error", "bf_device_count", "bf_set_device", "bf_scene_create",
    "bf_scene_destroy", "bf_scene_update_endpoints", "bf_scene_translate_meshes", "bf_scene_get_info", "bf_scene_clone", "bf_launch_channels", "bf_render_device", "bf_render",
    "bf_scene_flush", "bf_scene_sync", "bf_shard_range", "bf_render_sharded_device", "bf_render_sharded", "bf_allreduce_device",
    "bf_render_batch_device", "bf_render_batch",
    "bf_trace_closest", "bf_trace_any", "bf_ray_intersect", "bf_eval_elementary",
]

_lib = None


class BeifongError(RuntimeError):
    pass


def load_library(path=None):
    """dlopen libbeifong_hip.so; raise if it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("BF_HIP_LIB") or LIB_PATH      # BF_HIP_LIB: developer A/B builds
    if not os.path.exists(p):
        raise BeifongError(
            f"{p} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback.")
    lib = C.CDLL(p)
    vp = C.c_void_p
    lib.bf_version.restype = C.c_int
    if lib.bf_version() != BF_ABI_VERSION:
        raise BeifongError(f"{p}: ABI version {lib.bf_version()}, this binding is for {BF_ABI_VERSION} (include/beifong_hip.h) — rebuild")
    # struct sizes as the library was compiled against this binding's ctypes mirrors (include/beifong_hip.h: bf_abi_sizeof)
    lib.bf_abi_sizeof.argtypes = [C.c_uint32]
    lib.bf_abi_sizeof.restype = C.c_uint32
    for k, t in enumerate(ABI_STRUCTS):
        if lib.bf_abi_sizeof(k) != C.sizeof(t):
            raise BeifongError(f"{p}: sizeof({t.__name__}) is {lib.bf_abi_sizeof(k)} in the library, {C.sizeof(t)} in this binding — "
                               "rebuild (python -c 'import __graft_entry__ as g; g.build()')")
    lib.bf_last_error.restype = C.c_char_p
    lib.bf_device_count.restype = C.c_int
    lib.bf_set_device.argtypes = [C.c_int]
    lib.bf_scene_create.argtypes = [C.POINTER(bf_scene_desc), C.POINTER(vp)]
    lib.bf_scene_destroy.argtypes = [vp]
    lib.bf_scene_get_info.argtypes = [vp, C.POINTER(bf_scene_info)]
    lib.bf_scene_clone.argtypes = [vp, C.POINTER(vp)]
    lib.bf_scene_update_endpoints.argtypes = [vp, C.POINTER(bf_scene_desc), vp]
    lib.bf_scene_translate_meshes.argtypes = [vp, C.POINTER(C.c_float * 3), vp]
    lib.bf_launch_channels.argtypes = [C.POINTER(bf_launch)]
    lib.bf_launch_channels.restype = C.c_uint32
    lib.bf_render_device.argtypes = [vp, C.POINTER(bf_launch), vp, vp, vp, C.POINTER(bf_stats)]
    lib.bf_render.argtypes = [vp, C.POINTER(bf_launch), vp, vp, C.POINTER(bf_stats)]
    lib.bf_scene_flush.argtypes = [vp, vp, C.POINTER(bf_stats)]
    lib.bf_scene_sync.argtypes = [vp]
    lib.bf_shard_range.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.bf_shard_range.restype = None
    lib.bf_render_sharded_device.argtypes = [C.POINTER(vp), C.c_uint32, C.POINTER(bf_launch), C.POINTER(vp), C.POINTER(vp), C.POINTER(bf_stats)]
    lib.bf_render_sharded.argtypes = [C.POINTER(vp), C.c_uint32, C.POINTER(bf_launch), vp, C.POINTER(bf_stats)]
    lib.bf_allreduce_device.argtypes = [C.POINTER(C.c_int), C.c_uint32, C.POINTER(vp), C.c_uint64, C.POINTER(vp)]
    lib.bf_render_batch_device.argtypes = [vp, C.POINTER(bf_launch), C.POINTER(bf_batch), vp, vp, vp, C.POINTER(bf_stats)]
    lib.bf_render_batch.argtypes = [vp, C.POINTER(bf_launch), C.POINTER(bf_batch), vp, vp, C.POINTER(bf_stats)]
    lib.bf_trace_closest.argtypes = [vp, C.c_uint64, vp, vp, vp, vp, vp]
    lib.bf_trace_any.argtypes = [vp, C.c_uint64, vp, vp]
    lib.bf_ray_intersect.argtypes = [vp, C.c_uint64, vp, vp, vp, vp]
    lib.bf_eval_elementary.argtypes = [C.c_int, C.c_uint64, vp, vp]
    if path is None:
        _lib = lib
    return lib


def check(lib, status, what):
    if status != BF_OK:
        msg = lib.bf_last_error()
        raise BeifongError(f"{what} failed (status {status}): {msg.decode() if msg else ''}")


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_launch(mode, n_paths, seed=0, path_offset=0, bins=0, bin_width=0.0, color_mode=BF_COLOR_RGB,
                max_depth=-1, rr_depth=5, time_c=3.0e8, flags=0, bins_y=0, phase_bins=0, film=None, spp=0):
    """film=(width, height), spp: multi-pixel film of the render modes; n_paths is then this call's share of the
    width * height * spp paths of the whole film (path g samples pixel g // spp)."""
    lp = bf_launch()
    if film is not None:
        lp.film_width, lp.film_height, lp.spp = int(film[0]), int(film[1]), int(spp)
    lp.mode, lp.color_mode, lp.n_paths, lp.path_offset, lp.seed = mode, color_mode, n_paths, path_offset, seed
    lp.max_depth, lp.rr_depth, lp.bins, lp.bin_width, lp.time_c, lp.flags = max_depth, rr_depth, bins, bin_width, time_c, flags
    lp.bins_y = bins_y
    lp.phase_bins = phase_bins
    return lp


def shard_range(n_paths, shard, n_shards, lib=None):
    """bf_shard_range: (offset, count) of shard `shard` of `n_shards` — the partition every multi-GPU driver uses
    (beifong_amd.dist.shard_range is the same arithmetic in Python)."""
    lib = lib or load_library()
    off, cnt = C.c_uint64(), C.c_uint64()
    lib.bf_shard_range(n_paths, shard, n_shards, C.byref(off), C.byref(cnt))
    return off.value, cnt.value


def render_sharded(scenes, launch, lib=None):
    """bf_render_sharded: ONE render split over the GPUs of `scenes` (one handle per GPU), host histogram + summed stats."""
    lib = lib or load_library()
    n = scenes[0].channels(launch)
    hist = np.zeros(n, dtype=np.float32)
    st = bf_stats()
    handles = (C.c_void_p * len(scenes))(*[s.handle for s in scenes])
    check(lib, lib.bf_render_sharded(handles, len(scenes), C.byref(launch), _ptr(hist), C.byref(st)), "bf_render_sharded")
    return hist, st


def render_sharded_device(scenes, launch, hist_ptrs, streams=None, lib=None):
    """bf_render_sharded_device: hist_ptrs[g] is a device pointer on scenes[g]'s GPU; all-reduced on completion."""
    lib = lib or load_library()
    handles = (C.c_void_p * len(scenes))(*[s.handle for s in scenes])
    hp = (C.c_void_p * len(scenes))(*[C.c_void_p(int(p)) for p in hist_ptrs])
    sp = (C.c_void_p * len(scenes))(*[C.c_void_p(int(x)) if x else None for x in streams]) if streams is not None else None
    check(lib, lib.bf_render_sharded_device(handles, len(scenes), C.byref(launch), hp, sp, None), "bf_render_sharded_device")


class Scene:
    """Device-resident immutable scene (bf_scene)."""

    def __init__(self, desc_holder, lib=None):
        self.lib = lib or load_library()
        self.holder = desc_holder          # keeps numpy arrays alive
        h = C.c_void_p()
        check(self.lib, self.lib.bf_scene_create(C.byref(desc_holder.desc), C.byref(h)), "bf_scene_create")
        self.handle = h

    def close(self):
        if self.handle:
            self.lib.bf_scene_destroy(self.handle)
            self.handle = None

    def clone(self):
        """bf_scene_clone: another handle on the same geometry (own endpoint tables, path pool, counters) for another stream."""
        other = Scene.__new__(Scene)
        other.lib = self.lib
        other.holder = self.holder
        h = C.c_void_p()
        check(self.lib, self.lib.bf_scene_clone(self.handle, C.byref(h)), "bf_scene_clone")
        other.handle = h
        return other

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update_endpoints(self, desc_holder, stream=0):
        """bf_scene_update_endpoints: same geometry, new rectangle / emitter / sensor records (no BVH rebuild)."""
        check(self.lib, self.lib.bf_scene_update_endpoints(self.handle, C.byref(desc_holder.desc),
                                                           C.c_void_p(stream) if stream else None),
              "bf_scene_update_endpoints")
        self.holder = desc_holder

    def translate_meshes(self, offset, stream=0):
        """bf_scene_translate_meshes: all triangles to fl(p0 + offset), BVH re-fitted in place."""
        off = (C.c_float * 3)(*[float(x) for x in offset])
        check(self.lib, self.lib.bf_scene_translate_meshes(self.handle, C.byref(off), C.c_void_p(stream) if stream else None),
              "bf_scene_translate_meshes")

    def info(self):
        i = bf_scene_info()
        check(self.lib, self.lib.bf_scene_get_info(self.handle, C.byref(i)), "bf_scene_get_info")
        return i

    def channels(self, launch):
        return self.lib.bf_launch_channels(C.byref(launch))

    def render(self, launch, records=False):
        """bf_render: host histogram float32[channels] (+ per-path records, stats)."""
        n = self.channels(launch)
        hist = np.zeros(n, dtype=np.float32)
        rec = np.zeros(launch.n_paths, dtype=PATH_RECORD_DTYPE) if records else None
        st = bf_stats()
        check(self.lib, self.lib.bf_render(self.handle, C.byref(launch), _ptr(hist), _ptr(rec), C.byref(st)), "bf_render")
        return hist, rec, st

    def render_device(self, launch, hist_ptr, stream=0, records_ptr=None, want_stats=False):
        """bf_render_device: accumulate into a device buffer (e.g. a torch tensor's data_ptr)."""
        st = bf_stats() if want_stats else None
        check(self.lib, self.lib.bf_render_device(self.handle, C.byref(launch), C.c_void_p(hist_ptr),
                                                  C.c_void_p(records_ptr) if records_ptr else None,
                                                  C.c_void_p(stream) if stream else None,
                                                  C.byref(st) if st is not None else None), "bf_render_device")
        return st

    def flush(self, stream=0, want_stats=False):
        """bf_scene_flush: finish the paths the handle's rolling renders (BF_FLAG_ROLLING) left alive; with want_stats the
        call waits and returns the whole sequence's bf_stats."""
        st = bf_stats() if want_stats else None
        check(self.lib, self.lib.bf_scene_flush(self.handle, C.c_void_p(stream) if stream else None,
                                                C.byref(st) if st is not None else None), "bf_scene_flush")
        return st

    def sync(self):
        """bf_scene_sync: flush, wait for the handle's work and raise if any render since the last check dropped rays."""
        check(self.lib, self.lib.bf_scene_sync(self.handle), "bf_scene_sync")

    @staticmethod
    def _batch(n_renders, seeds, offsets):
        """bf_batch + the arrays it points to (keep the tuple alive for the duration of the call)."""
        b = bf_batch()
        b.n_renders = int(n_renders)
        sa = oa = None
        if seeds is not None:
            sa = np.ascontiguousarray(seeds, dtype=np.uint64).reshape(-1)
            assert sa.size == b.n_renders, "one seed per render"
            b.seeds = sa.ctypes.data_as(C.POINTER(C.c_uint64))
        if offsets is not None:
            oa = np.ascontiguousarray(offsets, dtype=np.float32).reshape(-1, 3)
            assert oa.shape[0] == b.n_renders, "one mesh offset per render"
            b.mesh_offsets = oa.ctypes.data_as(C.POINTER(C.c_float))
        return b, sa, oa

    def render_batch(self, launch, n_renders, seeds=None, offsets=None, records=False):
        """bf_render_batch: n_renders renders of `launch` in one launch sequence -> float32[n_renders, channels]
        (+ records [n_renders, n_paths], stats)."""
        n = self.channels(launch)
        hist = np.zeros((n_renders, n), dtype=np.float32)
        rec = np.zeros((n_renders, launch.n_paths), dtype=PATH_RECORD_DTYPE) if records else None
        st = bf_stats()
        b, sa, oa = self._batch(n_renders, seeds, offsets)
        check(self.lib, self.lib.bf_render_batch(self.handle, C.byref(launch), C.byref(b), _ptr(hist), _ptr(rec), C.byref(st)),
              "bf_render_batch")
        return hist, rec, st

    def render_batch_device(self, launch, n_renders, hist_ptr, seeds=None, offsets=None, stream=0, records_ptr=None,
                            want_stats=False):
        """bf_render_batch_device: accumulate n_renders renders into hist_ptr[n_renders * channels] (device memory)."""
        st = bf_stats() if want_stats else None
        b, sa, oa = self._batch(n_renders, seeds, offsets)
        check(self.lib, self.lib.bf_render_batch_device(self.handle, C.byref(launch), C.byref(b), C.c_void_p(hist_ptr),
                                                        C.c_void_p(records_ptr) if records_ptr else None,
                                                        C.c_void_p(stream) if stream else None,
                                                        C.byref(st) if st is not None else None), "bf_render_batch_device")
        return st

    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        t = np.empty(n, np.float32)
        prim = np.empty(n, np.uint32)
        shape = np.empty(n, np.uint32)
        uv = np.empty((n, 2), np.float32)
        check(self.lib, self.lib.bf_trace_closest(self.handle, n, _ptr(rays), _ptr(t), _ptr(prim), _ptr(shape), _ptr(uv)),
              "bf_trace_closest")
        return t, prim, shape, uv

    def ray_intersect(self, rays):
        """Scene::ray_intersect -> dict of SurfaceInteraction fields (arrays over rays), prim, shape."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        si = np.empty((n, BF_SI_FLOATS), np.float32)
        prim = np.empty(n, np.uint32)
        shape = np.empty(n, np.uint32)
        check(self.lib, self.lib.bf_ray_intersect(self.handle, n, _ptr(rays), _ptr(si), _ptr(prim), _ptr(shape)),
              "bf_ray_intersect")
        return dict(t=si[:, 0], p=si[:, 1:4], n=si[:, 4:7], sh_n=si[:, 7:10], sh_s=si[:, 10:13], sh_t=si[:, 13:16],
                    wi=si[:, 16:19], prim_uv=si[:, 19:21], dp_du=si[:, 21:24], dp_dv=si[:, 24:27], prim=prim, shape=shape,
                    raw=si)

    def trace_any(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        hit = np.empty(n, np.uint8)
        check(self.lib, self.lib.bf_trace_any(self.handle, n, _ptr(rays), _ptr(hit)), "bf_trace_any")
        return hit
