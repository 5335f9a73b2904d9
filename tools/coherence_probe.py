"""Developer probe: how much does ray ORDER matter to a one-ray-per-lane traversal of the 200 k bus?  Traces the same ray
sets in random order and sorted (primary rays by direction, secondary rays by origin cell then direction octant) through
bf_trace_closest; run under `rocprofv3 --kernel-trace` and read the bf_trace_kernel durations in dispatch order
(tools/coherence_probe_report.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from beifong_amd import capi, scenes

N = int(os.environ.get("RAYS", 1 << 22))
sd, lp = scenes.bus_radar(n_tris=200_000, n_paths=64)
g = capi.Scene(sd)
info = g.info()
rng = np.random.default_rng(1)
# the bus: roughly x 4..16, y 0..6, z 0..3.5 (placed at (10, 3, 1.7), yaw -20 deg)
lo, hi = np.array([4.0, 0.0, 0.2]), np.array([16.0, 6.0, 3.4])
o = np.array([0.0, 0.0, 0.3])
p = lo + rng.random((N, 3)) * (hi - lo)
d = p - o
d /= np.linalg.norm(d, axis=1, keepdims=True)
prim_rays = np.zeros((N, 8), np.float32)
prim_rays[:, 0:3] = o
prim_rays[:, 3] = 1e-4
prim_rays[:, 4:7] = d
prim_rays[:, 7] = np.inf


def morton2(u, v, bits):
    k = np.zeros(len(u), np.uint64)
    for b in range(bits):
        k |= ((u >> b) & 1).astype(np.uint64) << (2 * b) | ((v >> b) & 1).astype(np.uint64) << (2 * b + 1)
    return k


def morton3(x, y, z, bits):
    k = np.zeros(len(x), np.uint64)
    for b in range(bits):
        k |= ((x >> b) & 1).astype(np.uint64) << (3 * b) | ((y >> b) & 1).astype(np.uint64) << (3 * b + 1) | \
             ((z >> b) & 1).astype(np.uint64) << (3 * b + 2)
    return k


def dir_key(d, bits):
    # octahedral map of the direction, Morton-interleaved
    a = np.abs(d).sum(1, keepdims=True)
    q = d / a
    u, v = q[:, 0].copy(), q[:, 1].copy()
    neg = q[:, 2] < 0
    u2 = (1 - np.abs(q[:, 1])) * np.sign(q[:, 0] + 1e-30)
    v2 = (1 - np.abs(q[:, 0])) * np.sign(q[:, 1] + 1e-30)
    u[neg], v[neg] = u2[neg], v2[neg]
    s = (1 << bits) - 1
    return morton2(((u * .5 + .5) * s).astype(np.uint32), ((v * .5 + .5) * s).astype(np.uint32), bits)


t, prim, shape, uv = g.trace_closest(prim_rays)                     # dispatch 0: primary, random order
hitm = np.isfinite(t)
print(f"primary: {hitm.mean():.3f} hit")
order = np.argsort(dir_key(prim_rays[:, 4:7].astype(np.float64), 12), kind="stable")
g.trace_closest(prim_rays[order])                                     # dispatch 1: primary, sorted by direction

# secondary rays: from the hit points, uniformly random directions (half of them leave through the surface they start on)
hp = (prim_rays[:, 0:3] + t[:, None] * prim_rays[:, 4:7])[hitm]
M = len(hp)
dd = rng.normal(size=(M, 3))
dd /= np.linalg.norm(dd, axis=1, keepdims=True)
sec = np.zeros((M, 8), np.float32)
sec[:, 0:3] = hp
sec[:, 3] = 1e-3
sec[:, 4:7] = dd
sec[:, 7] = np.inf
perm = rng.permutation(M)
sec = sec[perm]
g.trace_closest(sec)                                                  # dispatch 2: secondary, random order
blo, bhi = hp.min(0), hp.max(0)
for bits in (5, 10):
    c = ((sec[:, 0:3] - blo) / (bhi - blo + 1e-9) * ((1 << bits) - 1)).astype(np.uint32)
    key = morton3(c[:, 0], c[:, 1], c[:, 2], bits)
    if bits == 5:
        octant = ((sec[:, 4] < 0).astype(np.uint64) | (sec[:, 5] < 0).astype(np.uint64) << 1 | (sec[:, 6] < 0).astype(np.uint64) << 2)
        key = (key << 3) | octant
    g.trace_closest(sec[np.argsort(key, kind="stable")])             # dispatch 3: 15-bit origin + octant; dispatch 4: 30-bit origin
key = (dir_key(sec[:, 4:7].astype(np.float64), 3) << 30) | morton3(*(((sec[:, 0:3] - blo) / (bhi - blo + 1e-9) * 1023).astype(np.uint32).T), 10)
g.trace_closest(sec[np.argsort(key, kind="stable")])                 # dispatch 5: coarse direction (64 cells) major, origin minor

# candidate 18-bit bin keys for an on-device counting sort (entry point into the mesh bounds / origin if inside, + direction)
def entry_point(rays):
    o_, d_ = rays[:, 0:3].astype(np.float64), rays[:, 4:7].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (blo - o_) / d_
        t1 = (bhi - o_) / d_
    tn = np.nanmax(np.minimum(t0, t1), axis=1)
    tn = np.maximum(tn, 0.0)
    return o_ + tn[:, None] * d_


def key18(rays, variant):
    e = entry_point(rays)
    if variant == "dir6_pos12":
        c = np.clip((e - blo) / (bhi - blo + 1e-9) * 15, 0, 15).astype(np.uint32)
        return (dir_key(rays[:, 4:7].astype(np.float64), 3) << 12) | morton3(c[:, 0], c[:, 1], c[:, 2], 4)
    c = np.clip((e - blo) / (bhi - blo + 1e-9) * 31, 0, 31).astype(np.uint32)
    octant = ((rays[:, 4] < 0).astype(np.uint64) | (rays[:, 5] < 0).astype(np.uint64) << 1 | (rays[:, 6] < 0).astype(np.uint64) << 2)
    if variant == "oct3_pos15":
        return (octant << 15) | morton3(c[:, 0], c[:, 1], c[:, 2], 5)
    return (morton3(c[:, 0], c[:, 1], c[:, 2], 5) << 3) | octant        # pos15_oct3


# shadow-like rays: from the hit points towards a point next to the sensor
sh = np.zeros((M, 8), np.float32)
sh[:, 0:3] = hp[perm]
tgt = np.array([0.0, 0.1, 0.3])
dv = tgt - sh[:, 0:3]
dist = np.linalg.norm(dv, axis=1)
sh[:, 4:7] = dv / dist[:, None]
sh[:, 3] = 1e-3
sh[:, 7] = dist * 0.999
for rays in (prim_rays, sec, sh):
    g.trace_closest(rays)
    for v in ("dir6_pos12", "oct3_pos15", "pos15_oct3"):
        g.trace_closest(rays[np.argsort(key18(rays, v), kind="stable")])
print("rays", N, "secondary", M)
