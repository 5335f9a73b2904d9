"""Frame sweeps: the reference's radar scripts render one frame per radar pose or pulse in a Python
loop that rebuilds the whole scene every time (python_scripts/animated_trans_rad.py:307-384,
Receive.ipynb cell 30).  Here the frames of a sweep

  * rotate over `n_streams` HIP streams, one device scene handle each, so that the latency-bound
    deep-path tail of one frame overlaps the heads of the next ones;
  * keep the BVH on the device while only endpoints move: a frame whose mesh arrays are the very
    arrays of the handle's previous frame is applied with bf_scene_update_endpoints (rectangle
    transforms, emitter / transmitter, sensor / receiver records — microseconds instead of a rebuild).

Needs torch only for device buffers and streams.
"""
import ctypes as C

import numpy as np

from . import capi


def geometry_key(sd):
    """Identity of a description's meshes: the addresses and sizes of the arrays it points to."""
    key = []
    for s in sd.shapes:
        if s.type == capi.BF_SHAPE_MESH:
            key.append((C.cast(s.positions, C.c_void_p).value, s.n_vertices, C.cast(s.indices, C.c_void_p).value, s.n_faces,
                        C.cast(s.normals, C.c_void_p).value))
        else:
            key.append(None)
    return tuple(key)


def render_sweep(frames, n_streams=4, lib=None, device=None, rolling=True):
    """Render `frames` — an iterable of (SceneDesc, bf_launch) — and return float32[n_frames, channels].

    All frames must produce the same number of channels.  Frames are independent renders; their
    order in the output is the input order.  rolling (round 4): the frames a handle renders form ONE rolling
    sequence — an endpoint update between two frames joins it (bf_scene_update_endpoints) — flushed once at the end,
    so a sweep whose radar turns per frame (animated_trans_rad.py:307-384) pays one tail per handle, not one per frame."""
    import torch
    lib = lib or capi.load_library()
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    handles = [None] * n_streams
    keys = [None] * n_streams
    hists = []
    stats = {"created": 0, "updated": 0}
    for k, (sd, lp) in enumerate(frames):
        j = k % n_streams
        key = geometry_key(sd)
        with torch.cuda.stream(streams[j]):
            if handles[j] is not None and keys[j] == key:
                handles[j].update_endpoints(sd, stream=streams[j].cuda_stream)
                stats["updated"] += 1
            else:
                if handles[j] is not None:
                    streams[j].synchronize()          # the old handle's last render must be done before it goes
                    handles[j].close()
                handles[j] = capi.Scene(sd, lib)
                keys[j] = key
                stats["created"] += 1
            h = torch.zeros(handles[j].channels(lp), dtype=torch.float32, device=dev)
            multi_pixel = lp.spp and lp.film_width and lp.film_height
            if rolling and not multi_pixel and not (lp.flags & capi.BF_FLAG_MEGAKERNEL):
                lr = capi.bf_launch()
                C.memmove(C.byref(lr), C.byref(lp), C.sizeof(capi.bf_launch))
                lr.flags |= capi.BF_FLAG_ROLLING
                lp = lr
            handles[j].render_device(lp, h.data_ptr(), stream=streams[j].cuda_stream)
            hists.append(h)
    for j, hd in enumerate(handles):
        if hd is not None:
            hd.flush(stream=streams[j].cuda_stream)
    for s in streams:
        s.synchronize()
    out = np.stack([h.cpu().numpy() for h in hists]) if hists else np.zeros((0, 0), np.float32)
    for h in handles:
        if h is not None:
            h.close()
    render_sweep.last_stats = stats
    return out


def _shard_launch(launch, off, cnt):
    lp = capi.bf_launch()
    C.memmove(C.byref(lp), C.byref(launch), C.sizeof(capi.bf_launch))
    lp.path_offset = launch.path_offset + off
    lp.n_paths = cnt
    return lp


class PulseSweeper:
    """Device-resident state of a pulse sweep: ONE scene (one BVH build) and `n_streams` HIP streams, kept across
    sweeps.  The pulses of a sweep are rendered as `n_streams` BATCHES (bf_render_batch_device): each batch is one
    launch sequence over its pulses — the pulse's mesh offset is added to the triangles on the fly and the cube is
    accumulated with atomics — so a sweep pays one latency-bound tail per batch instead of one per pulse, and the
    batches' tails overlap each other's heads.  See render_pulse_sweep."""

    def __init__(self, sd, launch, n_streams=2, lib=None, device=None):
        from . import configure_runtime
        configure_runtime()          # more hardware queues than the default 4 (no effect once HIP has started)
        import torch
        self.lib = lib or capi.load_library()
        self.dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.launch = launch
        self.n_streams = max(1, int(n_streams))
        self.streams = [torch.cuda.Stream(self.dev) for _ in range(self.n_streams)]
        # one handle per stream: a handle owns the path pool its renders run in (the BVH build is shared work only
        # through bf_scene_clone, see capi.Scene.clone)
        first = capi.Scene(sd, self.lib)
        self.handles = [first] + [first.clone() for _ in range(self.n_streams - 1)]
        self.n_chan = self.handles[0].channels(launch)

    def render(self, offsets, group=None, per_pulse=False):
        """float32[n_pulses, f_bins * t_bins, 3] of (I, Q, W) for target offsets float[n_pulses, 3].
        per_pulse=True renders pulse by pulse (bf_scene_translate_meshes + one render each): the round-1 path, kept
        as the reference the batched path is tested against."""
        import torch
        import torch.distributed as tdist
        from .dist import render_cube_sharded
        offsets = np.ascontiguousarray(np.asarray(offsets, dtype=np.float32).reshape(-1, 3))
        n = len(offsets)

        def render(path_off, count, cube):
            lp = _shard_launch(self.launch, path_off, count)
            # the cube was zero-filled on the current stream: the side streams must not start before that
            for s in self.streams:
                s.wait_stream(torch.cuda.current_stream(self.dev))
            if per_pulse:
                for k, off in enumerate(offsets):
                    j = k % self.n_streams
                    with torch.cuda.stream(self.streams[j]):
                        self.handles[j].translate_meshes(off, stream=self.streams[j].cuda_stream)
                        self.handles[j].render_device(lp, cube[k].data_ptr(), stream=self.streams[j].cuda_stream)
                for j in range(self.n_streams):         # leave the handles as built
                    with torch.cuda.stream(self.streams[j]):
                        self.handles[j].translate_meshes((0.0, 0.0, 0.0), stream=self.streams[j].cuda_stream)
            else:
                bounds = [n * j // self.n_streams for j in range(self.n_streams + 1)]
                for j in range(self.n_streams):
                    k0, k1 = bounds[j], bounds[j + 1]
                    if k1 == k0:
                        continue
                    with torch.cuda.stream(self.streams[j]):
                        self.handles[j].render_batch_device(lp, k1 - k0, cube[k0].data_ptr(), offsets=offsets[k0:k1],
                                                            stream=self.streams[j].cuda_stream)
            for s in self.streams:
                s.synchronize()

        cube, _ = render_cube_sharded(render, int(self.launch.n_paths), (n, self.n_chan), device=self.dev,
                                      group=group if tdist.is_initialized() else None)
        return cube.cpu().numpy().reshape(n, -1, 3)

    def close(self):
        for h in reversed(self.handles):
            h.close()
        self.handles = []


def render_pulse_sweep(sd, launch, offsets, n_streams=2, lib=None, device=None, group=None):
    """Coherent pulse sweep over a rigidly moving target (BASELINE configs[4]).

    `sd`      scene description whose meshes are the target (rectangles — ground, antennas — stay put);
    `launch`  a BF_MODE_RECEIVE_IQ launch; every pulse uses the SAME seed (common random numbers), so a
              path keeps its geometry from pulse to pulse and only its optical length — its phase — moves;
    `offsets` float[n_pulses, 3]: target offset of each pulse, relative to the scene as built.

    The BVH is built once; the pulses are rendered in `n_streams` batches (bf_render_batch_device: the pulse's offset
    is applied to the triangles on the fly, every path bit-identical to a render of the translated scene).  Returns the slow-time x fast-time
    cube float32[n_pulses, f_bins * t_bins, 3] of (I, Q, W).  One-shot form of PulseSweeper (which keeps the
    handles for further sweeps).

    Under an initialised torch.distributed job (one process per GPU) every rank renders its contiguous share of
    each pulse's paths (bf_launch.path_offset) and the cube is summed with one all-reduce at the end
    (beifong_amd.dist.render_cube_sharded); every rank returns the full cube."""
    n = len(np.asarray(offsets, dtype=np.float32).reshape(-1, 3))
    sw = PulseSweeper(sd, launch, max(1, min(n_streams, n)), lib, device)
    try:
        return sw.render(offsets, group)
    finally:
        sw.close()


def range_doppler(cube, window=True):
    """Slow-time FFT of a pulse-sweep cube: complex64[n_doppler, cells], zero Doppler at row 0 (numpy.fft order)."""
    z = cube[:, :, 0].astype(np.complex64) + 1j * cube[:, :, 1].astype(np.complex64)
    if window:
        z = z * np.hanning(z.shape[0])[:, None].astype(np.float32)
    return np.fft.fft(z, axis=0)
