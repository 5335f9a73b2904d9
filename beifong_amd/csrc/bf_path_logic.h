// The integrator's per-path logic, shared verbatim by the wavefront shade
// kernel (bf_wavefront.hip) and the megakernel / tail kernel (bf_kernels.hip):
//
//   generate_path : SamplingIntegrator::render_sample / receive_sample head
//                   (integrator.cpp:259-283, 1538-1572) — sampler draws and the
//                   sensor / receiver ray
//   shade_vertex  : one iteration of PathIntegrator::sample (path.cpp:121-209),
//                   PathLengthIntegrator (pathlength.cpp), PathTimeIntegrator
//                   (pathtime.cpp) or PathTimeFrequencyIntegrator
//                   (pathtimefrequency.cpp:154-447), given the closest hit of the
//                   path's current ray
//   film_put      : render_sample / receive_sample tail + range/time AOVs +
//                   ImageBlock::put / SignalBlock::put (box filter)
//
// The NEE contribution is computed BEFORE its shadow ray is traced and handed
// to the caller (ShadowReq), which adds it to PathState::result iff the ray is
// unoccluded — the same value the reference adds (scene.cpp:220-224 zeroes the
// emitter value of an occluded sample).
#pragma once
#include "bf_device_core.h"
#include "bf_wavefront.h"

namespace bfd {

constexpr uint32_t kFlagValid = 1u << 24, kFlagFilmOk = 1u << 25, kFlagTermPending = 1u << 26, kDepthMask = 0xffffffu;
// multi-pixel films: the sample landed in the left / upper neighbour of the pixel it was drawn in (ImageBlock::put)
constexpr uint32_t kFlagFilmLeft = 1u << 27, kFlagFilmUp = 1u << 28;

struct PathState {
    V3 ro, rd;
    float rmint, rmaxt;
    float throughput, eta, emission_weight, result;
    float aux, bs_pdf;
    V3 prev_p;
    uint32_t flags;      // depth | kFlag*
    uint32_t n_rays;
    Rng rng;
    uint64_t path_i;     // index within the launch; batched launches: global index over all renders (render * batch_paths + local)
    uint32_t render;     // batched launches: which render of the batch this path belongs to (0 otherwise)
    // gen-3 (receive) only
    float time;          // ray.time: retarded time carried along the path (ray.h:89-93)
    float t_rx;          // sampled receive time (integrator.cpp:1556-1561)
    float lambda0;       // ray.wavelengths[0] in nm
    float phase;         // ray.phase of the working ray: last traced segment only (ray.h:89-93, interaction.h:61-64)
    float dlambda;       // BF_FLAG_DOPPLER: what the Doppler hook has added to the caller's ray.wavelengths[0] (nm)
};

BF_DEV void load_state(const WF &wf, uint32_t i, bool receive, PathState &s) {
    float4 r0 = wf.ray0(i), r1 = wf.ray1(i), a = wf.sa(i), bb = wf.sb(i);
    uint4 d = wf.sd(i);
    s.ro = mk(r0.x, r0.y, r0.z);
    s.rmint = r0.w;
    s.rd = mk(r1.x, r1.y, r1.z);
    s.rmaxt = r1.w;
    s.throughput = a.x;
    s.eta = a.y;
    s.emission_weight = a.z;
    s.result = a.w;
    s.aux = bb.x;
    s.bs_pdf = bb.y;
    s.prev_p = s.ro;             // the previous vertex IS the origin of the ray in flight (spawn_ray, interaction.h:61-64)
    s.flags = __float_as_uint(bb.z);
    s.n_rays = __float_as_uint(bb.w);
    s.rng.state = ((uint64_t) d.y << 32) | d.x;
    s.path_i = ((uint64_t) d.w << 32) | d.z;
    s.time = s.t_rx = s.lambda0 = s.phase = 0.f;
    s.render = wf.has_render ? wf.render(i) : 0u;
    s.dlambda = wf.has_dop ? wf.dop(i) : 0.f;
    if (receive) {
        float4 e = wf.se(i);
        s.time = e.x;
        s.t_rx = e.y;
        s.lambda0 = e.z;
        s.phase = e.w;
    }
}
BF_DEV void store_state(const WF &wf, uint32_t j, bool receive, const PathState &s) {
    wf.ray0(j) = make_float4(s.ro.x, s.ro.y, s.ro.z, s.rmint);
    wf.ray1(j) = make_float4(s.rd.x, s.rd.y, s.rd.z, s.rmaxt);
    wf.sa(j) = make_float4(s.throughput, s.eta, s.emission_weight, s.result);
    wf.sb(j) = make_float4(s.aux, s.bs_pdf, __uint_as_float(s.flags), __uint_as_float(s.n_rays));
    wf.sd(j) = make_uint4((uint32_t) s.rng.state, (uint32_t) (s.rng.state >> 32), (uint32_t) s.path_i,
                          (uint32_t) (s.path_i >> 32));
    if (receive) wf.se(j) = make_float4(s.time, s.t_rx, s.lambda0, s.phase);
    if (wf.has_render) wf.render(j) = s.render;
    if (wf.has_dop) wf.dop(j) = s.dlambda;
}
// Shape::doppler — src/librender/shape.cpp:375-389 (call sites commented out at the reference's HEAD:
// pathtimefrequency.cpp:141-144, 180-183): 2 dot(si.wi, m_velocity * Point3f(si.to_local(si.p))) / MTS_C * wavelength
BF_DEV float shape_doppler(const DScene &sc, const SI &si, float lambda_nm) {
    V3 q = xf_point(c_shapes(sc)[si.shape].velocity, to_local(si.sh, si.p));
    return 2.f * dot(si.wi, q) / sc.c * lambda_nm;
}
// The scene a path of render `render` sees.  Plain launches and sequences whose endpoints stand still: the launch's own
// (wave-uniform: the tables are read through the scalar cache).  kMulti: the table pointers and physics of the path's render out
// of the descriptor ring — a per-lane record, so everything read through them is a vector load.
template <int V> BF_DEV DScene path_scene(const DScene &sc, const DLaunch &lp, uint32_t render) {
    if (!(V & kMulti)) return sc;
    DScene r = sc;
    const BF_CAS DRoll &e = as_const(lp.roll)[render & (kRollRing - 1u)];
    r.rects = e.rects;
    r.shapes = e.shapes;
    r.emitters = e.emitters;
    r.materials = e.materials;
    r.sensor = e.sensor;
    r.c = e.c;
    r.lambda_min = e.lambda_min;
    r.lambda_max = e.lambda_max;
    r.tab_on = 0u;                 // (the workgroup's LDS copies hold one version)
    return r;
}
// the mesh shift of the path's render (batched launches with moving meshes; off otherwise)
BF_DEV Shift path_shift(const DLaunch &lp, uint32_t render) { return make_shift(lp.batch_offsets, render, lp.box_slack); }

// ---------------------------------------------------------------------------
// on-the-fly compaction: a cursor over a segment of 64-bit batch masks hands the
// next set bits (= slots that need work) to the lanes that ask for one
// ---------------------------------------------------------------------------
BF_DEV uint32_t nth_set_bit(unsigned long long m, uint32_t r) {
    // position of the r-th (0-based) set bit of m; caller guarantees r < popc(m)
    uint32_t pos = 0;
#pragma unroll
    for (int shift = 32; shift >= 1; shift >>= 1) {
        unsigned long long lowmask = (1ull << shift) - 1ull;
        uint32_t cnt = (uint32_t) __popcll(m & lowmask);
        if (r >= cnt) {
            r -= cnt;
            m >>= shift;
            pos += shift;
        } else {
            m &= lowmask;
        }
    }
    return pos;
}

// Walks the set bits of a wave's share of a batch-mask array.  The words are fetched 64 at a
// time, one per lane, so skipping the empty words of a sparse pool costs one load and a few
// scalar bit operations per 64 batches instead of one dependent memory round trip per word
// (which made every late, nearly empty bounce iteration cost 200-400 us whatever little work
// it held).
//
// A wave's share of the mask array is INTERLEAVED, not contiguous: wave w of W owns batches w, w + W, w + 2 W, ...
// Live slots are not spread evenly over a pool — a rolling sequence keeps one render in each half of the main slots and
// all its old paths in the survivor area behind them (bf_wavefront.h) — and with contiguous segments the few waves that
// own the busy region did all the work while the others idled; batch-granular interleaving balances any distribution to
// within one batch per wave.  The words are still fetched 64 at a time (lane l reads the wave's (64 k + l)-th batch: a
// strided gather, one 8-byte word per cache line — the masks are a few MB per launch, the state rows GBs).
struct MaskCursor {
    const unsigned long long *masks;
    const unsigned long long *masks2;   // optional: the walk covers masks & masks2 (sel = 1) or masks & ~masks2 (sel = 2)
    uint32_t sel;
    uint32_t b, b_end;          // current batch / number of batches (wave-uniform)
    unsigned long long m;       // unconsumed bits of batch b (wave-uniform)
    uint32_t base, stride;      // this wave's first batch / distance between its batches (the number of waves sharing the array)
    uint32_t k;                 // chunk: lane l holds the word of batch base + (64 k + l) stride
    unsigned long long sub;     // slots of a batch this wave serves (all ones, or a 32- / 16-slot share: the tail's spreading)
    unsigned long long w;       // this lane's word of the chunk, & sub
    unsigned long long nz;      // wave-uniform: chunk words that are non-zero and not consumed yet
    // the sources of the generating launches (wf_shade<1>, <2> with lane refill): the walk covers `b_end` batches starting at
    // physical batch `rot` (wrapping at `mod`), of the complemented words (inv: slots WITHOUT a live path) or — masks == nullptr —
    // of every slot.  Plain cursors: rot = 0, inv = 0, pb == b.
    uint32_t rot, mod, inv, pb;
};
BF_DEV unsigned long long wave_read_u64(unsigned long long v, int src) {
    unsigned lo = (unsigned) __shfl((int) (unsigned) v, src), hi = (unsigned) __shfl((int) (unsigned) (v >> 32), src);
    return ((unsigned long long) hi << 32) | lo;
}
BF_DEV void cursor_fetch(MaskCursor &c, int lane) {
    const uint32_t idx = c.base + (c.k * 64u + (uint32_t) lane) * c.stride;
    uint32_t p = idx + c.rot;
    if (p >= c.mod) p -= c.mod;
    c.w = 0ull;
    if (idx < c.b_end) {
        c.w = c.masks ? c.masks[p] : ~0ull;
        if (c.inv) c.w = ~c.w;
        c.w &= c.sub;
        if (c.sel) c.w &= c.sel == 1u ? c.masks2[p] : ~c.masks2[p];
    }
    c.nz = __ballot(c.w != 0ull);
}
// position the cursor on the wave's next non-empty batch (or at the end)
BF_DEV void cursor_seek(MaskCursor &c, int lane) {
    while (true) {
        if (c.nz) {
            const int j = __ffsll((unsigned long long) c.nz) - 1;
            c.nz &= c.nz - 1ull;
            c.b = c.base + (c.k * 64u + (uint32_t) j) * c.stride;
            c.pb = c.b + c.rot;
            if (c.pb >= c.mod) c.pb -= c.mod;
            c.m = wave_read_u64(c.w, j);
            return;
        }
        ++c.k;
        if (c.base + c.k * 64u * c.stride >= c.b_end) {
            c.b = c.b_end;
            c.m = 0ull;
            return;
        }
        cursor_fetch(c, lane);
    }
}
// `share` (1, 2 or 4): that many consecutive waves serve the same batches, each one its own 64 / share slots of every batch
BF_DEV void cursor_init(MaskCursor &c, const unsigned long long *masks, uint32_t wave_id, uint32_t n_waves, uint32_t n_batches, int lane,
                        uint32_t share = 1u, const unsigned long long *masks2 = nullptr, uint32_t sel = 0u, uint32_t rot = 0u,
                        uint32_t mod = 0xffffffffu, uint32_t inv = 0u) {
    c.rot = rot;
    c.mod = mod;
    c.inv = inv;
    c.pb = 0u;
    c.masks2 = masks2;
    c.sel = masks2 ? sel : 0u;
    c.sub = ~0ull;
    if (share > 1u) {
        const uint32_t width = 64u / share, q = wave_id % share;
        c.sub = ((1ull << width) - 1ull) << (width * q);
        wave_id /= share;
        n_waves = max(1u, n_waves / share);
    }
    c.masks = masks;
    c.b_end = n_batches;
    c.base = wave_id;
    c.stride = n_waves;
    c.k = 0u;
    c.m = 0ull;
    c.w = 0ull;
    c.nz = 0ull;
    if (wave_id >= n_batches) {
        c.b = n_batches;
        return;
    }
    cursor_fetch(c, lane);
    cursor_seek(c, lane);
}
BF_DEV bool cursor_empty(const MaskCursor &c) { return c.b >= c.b_end; }
BF_DEV void cursor_skip_empty(MaskCursor &c, int lane) {
    if (c.b < c.b_end && c.m == 0ull) cursor_seek(c, lane);
}
// Hands out up to `want` slots: the requesting lane of rank r (0-based among the
// requesters) receives a slot iff r < return value.  All control flow is uniform.
BF_DEV uint32_t cursor_take(MaskCursor &c, uint32_t want, bool requesting, uint32_t rank, uint32_t &slot, int lane) {
    uint32_t taken = 0;
    while (taken < want && c.b < c.b_end) {
        if (c.m == 0ull) {
            cursor_seek(c, lane);
            continue;
        }
        uint32_t cnt = (uint32_t) __popcll(c.m);
        uint32_t take = min(want - taken, cnt);
        if (requesting && rank >= taken && rank < taken + take) slot = c.pb * 64u + nth_set_bit(c.m, rank - taken);
        if (take == cnt) {
            c.m = 0ull;
        } else {
            uint32_t p = nth_set_bit(c.m, take - 1u);        // drop the `take` lowest set bits
            c.m &= ~((2ull << p) - 1ull);
        }
        taken += take;
    }
    return taken;
}

// ---------------------------------------------------------------------------
// gen-3 building blocks: "Jacob functions" (math.h:62-131), the rectangular
// aperture Wigner gain (rectangle.cpp:132-220), the transmitter signal model
// (wignertransmitter.cpp:111-146).  A literal `1e-9` is a double in the
// reference, so those products run in double here too.
// ---------------------------------------------------------------------------
BF_DEV float jabs(float x) { return x >= 0.f ? x : -x; }
// Ray::update_state, phase part (ray.h:89-93): float product, double quotient and sum, float store;
// the divisor is HALF THE BAND WIDTH in metres (Q4)
BF_DEV float phase_update(float phase, float t, float lambda_min_nm, float lambda_max_nm) {
    float num = 6.28318530717958647692f * t;
    double den = (double) ((lambda_max_nm - lambda_min_nm) / 2.f) * 1e-9;
    return (float) ((double) phase + (double) num / den);
}
BF_DEV float sinc_j(float x) { return jabs(x) > kEpsilon ? sin_cr(x) / x : 1.f; }
BF_DEV float tri_j(float x) { return jabs(x) < 0.5f ? 1.f - 2.f * jabs(x) : 0.f; }
BF_DEV float rect_j(float x) { return jabs(x) < 0.5f ? 1.f : 0.f; }
BF_DEV float fmodulo_j(float a, float b) {
    float result = jabs(a);
    int guard = 0;
    while (result - jabs(b) >= kEpsilon && guard++ < (1 << 22)) result -= jabs(b);
    result = (a < 0.f) ? jabs(b) - result : result;
    result += (b < 0.f) ? b : 0.f;
    return result;
}
BF_DEV float wchirp_j(float t, float f, float w, float a) {
    return 2 * a * a * w * tri_j(t / w) * sinc_j(6.28318530717958647692f * f * w * tri_j(t / w));
}
BF_DEV float rect_sample_wigner(CRect &rc, V3 p, V3 d, float lambda_nm) {
    const float kTwoPi = 6.28318530717958647692f;
    V3 fs = mk(rc.s[0], rc.s[1], rc.s[2]), ft = mk(rc.t[0], rc.t[1], rc.t[2]), fn = mk(rc.n[0], rc.n[1], rc.n[2]);
    float wid_x = norm(fs), wid_y = norm(ft);
    V3 r_hat = xf_point(rc.to_object, p) / 2.f;
    V3 ns = normalize(fs), nt = normalize(ft);
    (void) fn;
    float lx = dot(ns, d), ly = dot(nt, d);
    double inv = 1.0 / ((double) lambda_nm * 1e-9);
    float nu_x = (float) ((double) lx * inv), nu_y = (float) ((double) ly * inv);
    return 4 * tri_j(r_hat.x) * tri_j(r_hat.y) * sinc_j(kTwoPi * nu_x * wid_x * tri_j(r_hat.x)) *
           sinc_j(kTwoPi * nu_y * wid_y * tri_j(r_hat.y));
}
// PhasedTransmitter / Phasedreceiver::sample_wigner — phasedtransmitter.cpp:273-291, phasedreceiver.cpp:279-297:
// real part of the sum over the n^2 virtual elements of
//   W_rect_2D(r, nu, wid) * exp(j 2 pi nu . r') * psi',   r = velem_to_object * p / 2 (only |r.x|, |r.y| <= 0.5),
//   nu = dir_to_local * d / (lambda 1e-9)
template <class W> BF_DEV float phased_sample_wigner(const float *__restrict__ tab, uint32_t n, W wid, V3 p, V3 d, float lambda_nm) {
    const float kTwoPi = 6.28318530717958647692f;
    const double inv = 1.0 / ((double) lambda_nm * 1e-9);
    float w_re = 0.f;
    for (uint32_t i = 0; i < n; ++i) {
        const float *e = tab + (size_t) BF_VELEM_FLOATS * i;
        V3 r_hat = xf_point(e, p) / 2.f;
        if (jabs(r_hat.x) <= 0.5f && jabs(r_hat.y) <= 0.5f) {
            V3 md = xf_vector(e + 12, d);
            V3 nu = mk((float) ((double) md.x * inv), (float) ((double) md.y * inv), (float) ((double) md.z * inv));
            float tx = tri_j(r_hat.x), ty = tri_j(r_hat.y);
            float wr = 4 * wid[0] * wid[1] * tx * ty * sinc_j(kTwoPi * nu.x * wid[0] * tx) * sinc_j(kTwoPi * nu.y * wid[1] * ty);
            float sn, cs;
            bf_sincos(kTwoPi * dot(nu, mk(e[24], e[25], e[26])), sn, cs);
            float a = wr * cs, b = wr * sn;                       // W_rect * exp(j ...)
            w_re += a * e[28] - b * e[29];                        // ... * psi', real part
        }
    }
    return w_re;
}
BF_DEV float tx_eval_signal(CEmitter &e, float time, float frequency) {
    if (e.signal_type == BF_SIGNAL_LINFMCW) {
        float t = fmodulo_j(time, rcp(e.prf));
        float ti = 0 + e.pulse_len / 2;
        float fi = e.freq_centre + (e.freq_ext / e.pulse_len) * (t - ti);
        return rect_j((t - ti) / e.pulse_len) > 0.f ? wchirp_j(t - ti, frequency - fi, e.pulse_len, e.amplitude) : 0.f;
    } else if (e.signal_type == BF_SIGNAL_PULSE) {
        float t = fmodulo_j(time, rcp(e.prf));
        float ti = 0 + e.pulse_len / 2;
        float fi = e.freq_centre;
        return rect_j((t - ti) / e.pulse_len) > 0.f ? wchirp_j(t - ti, frequency - fi, e.pulse_len, e.amplitude) : 0.f;
    }
    return e.amplitude * e.amplitude;
}
BF_DEV float freq_of(float c, float lambda_nm) { return (float) ((double) c * (1.0 / ((double) lambda_nm * 1e-9))); }
// WignerTransmitter::sample_delta_frequency (wignertransmitter.cpp:152-168), the frequency only: the instantaneous frequency of the
// chirp ("linfmcw") or the carrier ("cw") at `time`; the weight it returns is 1 whatever eval_signal says (:165).  "pulse" leaves
// `frequencies` uninitialised there: refused at scene creation (bf_api.cpp).
BF_DEV float tx_delta_frequency(CEmitter &e, float time) {
    if (e.signal_type == BF_SIGNAL_LINFMCW) {
        float t = fmodulo_j(time, rcp(e.prf));
        float ti = 0 + e.pulse_len / 2;
        return e.freq_centre + (e.freq_ext / e.pulse_len) * (t - ti);
    }
    return e.freq_centre;
}
// m_resample_freq (wignertransmitter.cpp:211-221, 430-441; phasedtransmitter.cpp likewise): the interaction's wavelength is
// overwritten with MTS_C * rcp(frequency) * 1e9 (a double product rounded to Float) before anything else reads it, and the signal
// power is 1
BF_DEV float tx_resampled_lambda(const DScene &sc, CEmitter &e, float time) {
    return (float) ((double) (sc.c * rcp(tx_delta_frequency(e, time))) * 1e9);
}

// Transmitter::eval — areatransmitter.cpp:65-73, wignertransmitter.cpp:193-271
template <int V = 0> BF_DEV float transmitter_eval(const DScene &sc, CEmitter &e, const SI &si, float si_time, float &lambda0) {
    CRect &rc = c_rects(sc)[e.rect];
    if (e.type == BF_TRANSMITTER_AREA) return (si.wi.z > 0.f) ? e.radiance * rc.area : 0.f;
    float signal_power;
    if (rare<V>(e.resample != 0u)) {
        lambda0 = tx_resampled_lambda(sc, e, si_time);           // si.wavelengths = ... (the path carries it on: spawn_ray, :451)
        signal_power = 1.f;
    } else {
        signal_power = tx_eval_signal(e, si_time, freq_of(sc.c, lambda0));
    }
    if (rare<V>(e.type == BF_TRANSMITTER_PHASED)) {
        // phasedtransmitter.cpp:296-381: geom_gain = antenna / area * sample_wigner(ds with the uninitialised d, Q5)
        float geom_gain = 1.f * rcp(rc.area);
        geom_gain *= phased_sample_wigner(e.velems, e.n_velems, e.wid, si.p, mk(-0.f, -0.f, -0.f), lambda0);
        return (si.wi.z > 0.f) ? signal_power * e.gain * geom_gain : 0.f;
    }
    float ws = rect_sample_wigner(rc, si.p, mk(0.f, 0.f, 0.f), lambda0);   // Q5: ds.d never initialised -> 0
    return (si.wi.z > 0.f) ? signal_power * e.gain * (1.f * ws) * 6.28318530717958647692f : 0.f;
}
// Transmitter::sample_direction — areatransmitter.cpp:117-165, wignertransmitter.cpp:373-534
template <int V = 0>
BF_DEV float transmitter_sample_direction(const DScene &sc, CEmitter &e, V3 ref_p, float ref_time, float &lambda0,
                                          float sx, float sy, DirSample &ds) {
    CRect &rc = c_rects(sc)[e.rect];
    V3 p = xf_point(rc.to_world, mk(sx * 2.f - 1.f, sy * 2.f - 1.f, 0.f));
    V3 n = mk(rc.n[0], rc.n[1], rc.n[2]);
    ds.pdf = rc.inv_area;
    ds.delta = false;
    ds.d = p - ref_p;
    float dist_squared = squared_norm(ds.d);
    ds.dist = __builtin_sqrtf(dist_squared);
    ds.d = ds.d / ds.dist;
    float dp = __builtin_fabsf(dot(ds.d, n));
    ds.pdf *= (dp != 0.f) ? dist_squared / dp : 0.f;
    bool active = dot(ds.d, n) < 0.f && ds.pdf != 0.f;
    if (e.type == BF_TRANSMITTER_AREA) {
        float spec = e.radiance / ds.pdf;
        return active ? spec : 0.f;
    }
    float geom_gain = 1.f / ds.pdf;
    float t = ref_time;
    if ((double) ds.dist > 5e-7) t += -ds.dist / sc.c;                      // retarded time :422-425
    float signal_power;
    if (rare<V>(e.resample != 0u)) {
        lambda0 = tx_resampled_lambda(sc, e, t);                 // it.wavelengths = ... : whether or not the sample is used
        signal_power = 1.f;
    } else {
        signal_power = tx_eval_signal(e, t, freq_of(sc.c, lambda0));
    }
    if (rare<V>(e.type == BF_TRANSMITTER_PHASED)) {
        // phasedtransmitter.cpp:560-585: geom_gain *= W; ds.pdf *= W; ds.pdf = sqrt(ds.pdf^2); extents = 1
        float w = phased_sample_wigner(e.velems, e.n_velems, e.wid, p, -ds.d, lambda0);
        geom_gain *= w;
        ds.pdf *= w;
        ds.pdf = __builtin_sqrtf(ds.pdf * ds.pdf);
        return active ? signal_power * e.gain * geom_gain * 1.f : 0.f;
    }
    float ws = rect_sample_wigner(rc, p, -ds.d, lambda0);
    geom_gain *= ws;
    ds.pdf *= ws;
    float extents = rcp(rc.area) * 6.28318530717958647692f;
    return active ? signal_power * e.gain * geom_gain * extents : 0.f;
}
// Transmitter::pdf_direction — areatransmitter.cpp:167-186, wignertransmitter.cpp:540-577
template <int V = 0> BF_DEV float transmitter_pdf_direction(const DScene &sc, CEmitter &e, V3 p_ref, V3 p_hit, V3 n_hit, float lambda0) {
    CRect &rc = c_rects(sc)[e.rect];
    V3 d = p_hit - p_ref;
    float dist = norm(d);
    d = d / dist;
    float dp = dot(d, n_hit);
    float value = rc.inv_area, adp = __builtin_fabsf(dot(d, n_hit));
    value *= (adp != 0.f) ? (dist * dist) / adp : 0.f;
    if (e.type == BF_TRANSMITTER_WIGNER) value *= rect_sample_wigner(rc, p_hit, -d, lambda0);
    if (rare<V>(e.type == BF_TRANSMITTER_PHASED)) {                // phasedtransmitter.cpp:606-620
        value *= phased_sample_wigner(e.velems, e.n_velems, e.wid, p_hit, -d, lambda0);
        value = __builtin_sqrtf(value * value);
    }
    return (dp < 0.f) ? value : 0.f;
}

// Receiver::sample_ray_differential — omnidirectional.cpp:72-107, wignerreceiver.cpp:208-269
template <int V = 0>
BF_DEV float receiver_sample_ray(const DScene &sc, float time, bool mix, float wl_sample, float px, float py, float ax, float ay, V3 &o, V3 &d,
                                 float &mint, float &maxt, float &lambda0) {
    CSensor &s = c_sensor(sc);
    CRect &rc = c_rects(sc)[s.rect];
    o = xf_point(rc.to_world, mk(px * 2.f - 1.f, py * 2.f - 1.f, 0.f));
    V3 local = square_to_cosine_hemisphere(ax, ay);
    Frame f;
    f.n = mk(rc.n[0], rc.n[1], rc.n[2]);
    coordinate_system(f.n, f.s, f.t);
    d = to_world(f, local);
    mint = kRayEpsilon;
    maxt = BF_INF;
    if ((V & kLean) || s.type == BF_RECEIVER_OMNI) {
        float lo = sc.lambda_min, hi = sc.lambda_max;
        lambda0 = wl_sample * (hi - lo) + lo;            // sample_uniform_spectrum (spectrum.h:312-316), lane 0
        return (hi - lo) * rc.area;
    }
    float freq = wl_sample * s.freq_ext + (s.freq_centre - s.freq_ext / 2);
    float signal_power = 1.f;
    if (mix) {
        // receive_type "mix_resample" (wignerreceiver.cpp:172-189): the receiver's own local oscillator
        if (s.rx_sig_is_delta) {
            // sample_delta_frequency(time) (:149-166): the instantaneous frequency at the sampled receive time, weight 1
            freq = s.freq_centre;
            if (s.rx_signal == BF_SIGNAL_LINFMCW) {
                float t = fmodulo_j(time, rcp(s.rx_prf));
                float ti = 0 + s.rx_pulse_len / 2;
                freq = s.freq_centre + (s.freq_ext / s.rx_pulse_len) * (t - ti);
            }
        } else {
            // the uniform sample above, weighted with eval_signal(time, f) (:118-142)
            signal_power = s.rx_amplitude * s.rx_amplitude;
            if (s.rx_signal != BF_SIGNAL_CW) {
                float t = fmodulo_j(time, rcp(s.rx_prf));
                float ti = 0 + s.rx_pulse_len / 2;
                float fi = s.rx_signal == BF_SIGNAL_LINFMCW ? s.freq_centre + (s.freq_ext / s.rx_pulse_len) * (t - ti) : s.freq_centre;
                signal_power = rect_j((t - ti) / s.rx_pulse_len) > 0.f ? wchirp_j(t - ti, freq - fi, s.rx_pulse_len, s.rx_amplitude) : 0.f;
            }
        }
    }
    lambda0 = (float) ((double) (sc.c * rcp(freq)) * 1e9);
    if (s.type == BF_RECEIVER_PHASED) {
        // phasedreceiver.cpp:299-365: geom_gain = W(ds) * pdf * (1 - (ds.d . ds.n)^4), ds.d the LOCAL cosine direction
        float w = phased_sample_wigner(s.velems, s.n_velems, s.wid, o, local, lambda0);
        float dn = dot(local, f.n);
        float geom = w * rc.inv_area * (1 - dn * dn * dn * dn);
        float ext = rc.area * kPi;
        if (!s.rx_sig_is_delta) ext = (float) ((double) (ext * (sc.c * rcp(s.freq_ext))) * 1e9);
        return signal_power * s.gain * geom * ext;
    }
    float ws = rect_sample_wigner(rc, o, local, lambda0);   // ds.d is the LOCAL cosine direction (:249-252)
    float geom_gain = ws * rc.inv_area;
    float extents = rc.area * kPi;
    if (!s.rx_sig_is_delta) extents = (float) ((double) (extents * (sc.c * rcp(s.freq_ext))) * 1e9);
    return signal_power * s.gain * geom_gain * extents;
}

// Mode specialisation: RX = 0 compiles the render modes only (path / range / time), RX = 1 the receive modes only,
// RX = 2 decides at run time (the one-kernel variant and the tail).  The receive branches carry the Wigner / phased-array
// / signal-model code; a shading kernel that cannot reach them allocates fewer registers.
template <int RX> BF_DEV bool mode_receive(const DLaunch &lp) { return (RX & kModeMask) == 2 ? lp.mode == BF_MODE_RECEIVE_RAW : (RX & kModeMask) == 1; }
// RX | kWide: the kernel variant for films / ADCs whose reconstruction filter is wider than a pixel (DLaunch::wide).  A variant
// of its own, not a branch: the filtered put's live values cost the box-filter kernels 3 % of wf_shade when both were compiled
// into one (profiles/r03_wide_filter_ab.txt), and every radar scene of the reference uses the box filter.

// ---------------------------------------------------------------------------
// path generation
// ---------------------------------------------------------------------------
template <int RX = 2> BF_DEV void generate_path(const DScene &sc0, const DLaunch &lp, uint64_t path_i, PathState &s) {
    const bool receive = mode_receive<RX>(lp);
    s.path_i = path_i;
    s.render = 0u;
    uint64_t seed = lp.seed, path_offset = lp.path_offset;
    if (lp.batch != 0u) {
        // batched launch: global index -> (render, local path); every render is an ordinary render of its own seed
        s.render = (uint32_t) (path_i / lp.batch_paths);
        path_i -= (uint64_t) s.render * lp.batch_paths;
        if (lp.roll) {                      // rolling sequence: the render's own seed and shard offset
            const DRoll &rr = lp.roll[s.render & (kRollRing - 1u)];
            seed = rr.seed;
            path_offset = rr.path_offset;
        } else if (lp.batch_seeds) {
            seed = lp.batch_seeds[s.render];
        }
    }
    const DScene sc = path_scene<RX>(sc0, lp, s.render);
    // per-path stream: sampler->seed(base_seed + path) (sampler.cpp:83-96)
    pcg_seed(s.rng, seed + path_offset + path_i);
    float fx = next_1d(s.rng), fy = next_1d(s.rng);
    float ax = .5f, ay = .5f;
    s.time = s.t_rx = s.lambda0 = s.phase = 0.f;
    s.dlambda = 0.f;
    bool film_ok = true, film_left = false, film_up = false;
    if (receive) {
        // receive_sample — integrator.cpp:1544-1572
        ax = next_1d(s.rng);
        ay = next_1d(s.rng);
        float time = c_sensor(sc).adc_sampling_start;
        if (c_sensor(sc).adc_sampling_time > 0.f)
            time += next_1d(s.rng) * c_sensor(sc).adc_sampling_time;
        else
            time = 0.f;
        float wl = next_1d(s.rng);
        s.t_rx = time;
        s.time = time;
        float w = receiver_sample_ray<RX>(sc, time, rare<RX>(lp.mix != 0u), wl, fx, fy, ax, ay, s.ro, s.rd, s.rmint, s.rmaxt, s.lambda0);
        s.aux = w;                 // receive has no path-length scalar: aux carries |ray_weight|'s operand
        if (rare<RX>(lp.resample != 0u)) s.dlambda = s.lambda0;      // the wavelength the receiver sampled (DLaunch::resample)
    } else {
        // render_sample — integrator.cpp:263-283
        if (rare<RX>(c_sensor(sc).type != BF_SENSOR_PERSPECTIVE && c_sensor(sc).type != BF_SENSOR_RADIANCEMETER)) {   // endpoint.h:241, perspective.cpp:130, radiancemeter.cpp:86-87
            ax = next_1d(s.rng);
            ay = next_1d(s.rng);
        }
        if (c_sensor(sc).shutter_open_time > 0.f) (void) next_1d(s.rng);
        (void) next_1d(s.rng);     // wavelength sample (consumed in RGB mode too)
        // position_sample = pos + next_2d; adjusted_position = position_sample / crop_size (integrator.cpp:263,276-278)
        uint32_t px = 0, py = 0;
        if (rare<RX>(lp.spp != 0u)) {
            const uint64_t q = (lp.path_offset + path_i) / lp.spp;
            px = (uint32_t) (q % lp.film_w);
            py = (uint32_t) (q / lp.film_w);
        }
        // film crop window (film.cpp:17-27): pixels are counted inside the crop, positions in the full film
        const uint32_t cx = rare<RX>(c_sensor(sc).crop_x != 0u) ? c_sensor(sc).crop_x : 0u, cy = rare<RX>(c_sensor(sc).crop_y != 0u) ? c_sensor(sc).crop_y : 0u;
        const float posx = (float) (px + cx) + fx, posy = (float) (py + cy) + fy;
        (void) sensor_sample_ray<RX>(sc, (posx - (float) cx) / (float) lp.film_w, (posy - (float) cy) / (float) lp.film_h, ax, ay, s.ro, s.rd,
                                     s.rmint, s.rmaxt);
        // ImageBlock::put, box branch (imageblock.cpp:113,166-172): pos = pos_ - (offset + .5) with the block's offset (the crop's,
        // plus whole blocks), the sample lands in pixel lo = ceil(pos - .5), i.e. the pixel it was drawn in or, when next_2d
        // returned exactly 0, its left / upper neighbour
        const float lx = __builtin_ceilf((posx - ((float) cx + .5f)) - .5f), ly = __builtin_ceilf((posy - ((float) cy + .5f)) - .5f);
        film_ok = lx >= 0.f && lx < (float) lp.film_w && ly >= 0.f && ly < (float) lp.film_h;
        film_left = lx < (float) px;
        film_up = ly < (float) py;
        s.aux = 0.f;
    }
    s.throughput = 1.f;
    s.eta = 1.f;
    s.emission_weight = 1.f;
    s.result = 0.f;
    s.bs_pdf = 0.f;
    s.prev_p = mk(0, 0, 0);
    s.flags = (film_ok ? kFlagFilmOk : 0u) | (film_left ? kFlagFilmLeft : 0u) | (film_up ? kFlagFilmUp : 0u);
    s.n_rays = 1;
}

// ---------------------------------------------------------------------------
// one integrator iteration
// ---------------------------------------------------------------------------
struct ShadowReq {
    bool want;
    V3 o, d;
    float mint, maxt;
    float c;             // NEE contribution released by an unoccluded shadow ray
    float c_im;          // BF_MODE_RECEIVE_IQ: its imaginary part (c is the real part)
};

// BF_MODE_RECEIVE_IQ: unit phasor exp(-j 2 pi L / lambda) of an optical path of `length` metres at
// wavelength `lambda_nm`; the cycle count is reduced to its fractional part before the sine / cosine.
BF_DEV void path_phasor(float length, float lambda_nm, float &re, float &im) {
    float cycles = length / (lambda_nm * 1e-9f);
    float frac = cycles - __builtin_floorf(cycles);
    float s, c;
    bf_sincos(-6.28318530717958647692f * frac, s, c);
    re = c;
    im = s;
}

// Returns true if the path continues with a new closest-hit ray (s.ro/rd/...),
// false if it ended at the head of the iteration (film_put is due now).  When
// the BSDF sample kills the path (path.cpp:171-173) the function returns true
// with kFlagTermPending set and an empty ray interval: the film write has to
// wait for the shadow ray of this very iteration.
#if defined(BF_TAIL_PROF) || defined(BF_SHADE_PROF)
#define BF_VERTEX_PROF 1
#endif
#ifdef BF_VERTEX_PROF
struct ShadeProf {
    unsigned long long si, head, nee, bsdf;
};
#define BF_SHADEPROF_ARG , ShadeProf *spf = nullptr
#define BF_SHADEPROF_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define BF_SHADEPROF_ARG
#define BF_SHADEPROF_STAMP(v)
#endif
// developer build (-DBF_SHADE_PROF, tools/shade_profile.py): how many lanes of a wave enter each section of wf_shade.
// SLP(sec, cond) sits in wave-uniform control flow right before `if (cond)`: one wave entry and popcount(cond) lanes.
#ifdef BF_SHADE_PROF
constexpr int kShadeProfSections = 32;
static __device__ unsigned long long g_lane_prof[2 * kShadeProfSections];      // [sec] = wave entries, [32 + sec] = lanes
#define BF_LANEPROF_ARG , bool lpf = false
#define SLP(sec, cond)                                                                      \
    do {                                                                                    \
        if (lpf) {                                                                          \
            const unsigned long long slp_b = __ballot(cond);                                \
            if (slp_b && (int) (threadIdx.x & 63) == __ffsll((long long) slp_b) - 1) {      \
                atomicAdd(&g_lane_prof[sec], 1ull);                                         \
                atomicAdd(&g_lane_prof[kShadeProfSections + (sec)], (unsigned long long) __popcll(slp_b)); \
            }                                                                               \
        }                                                                                   \
    } while (0)
#else
#define BF_LANEPROF_ARG
#define SLP(sec, cond)
#endif
template <int RX = 2>
BF_DEV bool shade_vertex(const DScene &sc0, const DLaunch &lp, PathState &s, const Hit &hit, ShadowReq &sh,
                         uint32_t &c_bounces BF_SHADEPROF_ARG BF_LANEPROF_ARG) {
    BF_SHADEPROF_STAMP(spf_t0);
    const DScene sc = path_scene<RX>(sc0, lp, s.render);
    const bool receive = mode_receive<RX>(lp);
    const bool is_range = !receive && lp.mode == BF_MODE_RANGE, is_time = rare<RX>(!receive && lp.mode == BF_MODE_TIME);
    const bool iq = receive && lp.iq != 0u;
    const bool phase_bins = rare<RX>(receive && lp.phase_bins != 0u);
    const bool doppler = rare<RX>(receive && lp.doppler != 0u);
    const uint32_t n_emit = (RX & kLean) ? 1u : sc.n_emitters;
    sh.want = false;
    SI si;
    const bool si_valid = hit.t != BF_INF;
    int emitter = -1;
    SLP(10, si_valid);
    bf_material mat;
    mat.type = ~0u;
    mat.back_material = 0u;
    if (si_valid) {
        make_si<false, RX>(sc, s.ro, s.rd, hit, si, nullptr, path_shift(lp, s.render));
        emitter = si.emitter;
        mat = load_material(sc, si.material);      // issued here, waited for where NEE / BSDF sampling first read it
    }
#ifdef BF_VERTEX_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    BF_SHADEPROF_STAMP(spf_t1);
    uint32_t depth = s.flags & kDepthMask;
    if (depth == 0) {
        // first intersection — path.cpp:115-117, pathlength.cpp:138-146,
        // pathtime.cpp:136-140, pathtimefrequency.cpp:131-153
        if (si_valid) s.flags |= kFlagValid;
        if (is_range) s.aux += si_valid ? si.t : 0.f;
        if (is_time) s.aux = si_valid ? si.t / lp.time_c : 0.f;
        if (si_valid && doppler) s.dlambda += shape_doppler(sc, si, s.lambda0);   // :141-144
        if (receive && si_valid) {                                 // ray.update_state(-si.t); si.time = ray.time
            s.time += -si.t / sc.c;
            if (phase_bins) s.phase = phase_update(s.phase, -si.t, sc.lambda_min, sc.lambda_max);
        }
        depth = 1;
    } else {
        // tail of the previous iteration — path.cpp:184-209, pathtimefrequency.cpp:363-399
        if (receive) {                                             // :368-371, also for a miss (Q3)
            s.time += -hit.t / sc.c;
            if (phase_bins) s.phase = phase_update(0.f, -hit.t, sc.lambda_min, sc.lambda_max);   // spawn_ray: phase restarts at 0
        }
        SLP(11, emitter >= 0);
        if (emitter >= 0) {
            CEmitter &e = c_emitters(sc)[(RX & kLean) ? 0 : emitter];      // lean: ONE emitter, a wave-uniform record
            float emitter_pdf = receive ? transmitter_pdf_direction<RX>(sc, e, s.prev_p, si.p, si.sh.n, s.lambda0)
                                        : emitter_pdf_direction<RX>(sc, e, s.prev_p, si.p, si.sh.n);
            if (n_emit != 1) emitter_pdf *= 1.f / (float) n_emit;
            s.emission_weight = mis_weight(s.bs_pdf, emitter_pdf);
        }
        if (is_range) s.aux += si_valid ? si.t : 0.f;
        if (is_time) s.aux += si_valid ? si.t / lp.time_c : 0.f;
        ++depth;
    }
    s.flags = (s.flags & ~kDepthMask) | (depth & kDepthMask);
    // head of iteration `depth` — path.cpp:121-145
    SLP(12, emitter >= 0);
    if (emitter >= 0) {
        CEmitter &e = c_emitters(sc)[(RX & kLean) ? 0 : emitter];
        if (doppler) s.dlambda += shape_doppler(sc, si, s.lambda0);               // :180-183
        float ev;
        if (receive)
            ev = transmitter_eval<RX>(sc, e, si, s.time, s.lambda0);
        else
            ev = ((RX & kLean) || e.type == BF_EMITTER_AREA) ? ((si.wi.z > 0.f) ? e.radiance : 0.f) : 0.f;
        float contrib = s.emission_weight * s.throughput * ev;
        if (iq) {
            // optical length receiver -> ... -> this transmitter point: c * (t_rx - retarded time)
            float re, im;
            path_phasor((s.t_rx - s.time) * sc.c, s.lambda0, re, im);
            s.result += contrib * re;
            s.phase += contrib * im;                               // imaginary accumulator (phase bins are off in IQ mode)
        } else {
            s.result += contrib;
        }
        if (is_range) s.aux += si_valid ? si.t : 0.f;              // pathlength.cpp:161
    }
    bool active = si_valid;
    if ((int) depth > lp.rr_depth) {
        float q = __builtin_fminf(s.throughput * sqr(s.eta), .95f);
        active = (next_1d(s.rng) < q) && active;
        s.throughput *= rcp(q);
    }
    if (depth >= (uint32_t) lp.max_depth || !active) return false;

    // TwoSidedBRDF with two nested BSDFs (twosided.cpp:108-178): the second one answers for incident directions below the surface
    if (rare<RX>(mat.back_material != 0u) && si.wi.z < 0.f) mat = load_material(sc, mat.back_material - 1u);
    ++c_bounces;
    BF_SHADEPROF_STAMP(spf_t2);
    SLP(13, true);                                        // lanes that survive to NEE + BSDF sampling
    SLP(15, mat.type == BF_BSDF_ROUGHCONDUCTOR);
    SLP(16, mat.type != BF_BSDF_ROUGHCONDUCTOR);
    if (bsdf_smooth(mat)) {
        // Scene::sample_emitter_direction / sample_transmitter_direction — scene.cpp:180-230, 249-299
        float sx = next_1d(s.rng), sy = next_1d(s.rng);
        DirSample ds;
        ds.d = mk(0, 0, 1);
        ds.pdf = 0.f;
        ds.dist = 0.f;
        ds.delta = false;
        float emitter_val = 0.f;
        if (n_emit >= 1) {
            uint32_t index = 0;
            float emitter_pdf = 1.f;
            if (n_emit > 1) {
                emitter_pdf = 1.f / (float) n_emit;
                index = min((uint32_t) (sx * (float) n_emit), n_emit - 1u);
                sx = (sx - index * emitter_pdf) * (float) n_emit;
            }
            CEmitter &e = c_emitters(sc)[index];
            if (receive)
                emitter_val = transmitter_sample_direction<RX>(sc, e, si.p, s.time, s.lambda0, sx, sy, ds);
            else
                emitter_val = emitter_sample_direction<RX>(sc, e, si.p, sx, sy, ds);
            if (n_emit > 1) {
                ds.pdf *= emitter_pdf;
                emitter_val *= rcp(emitter_pdf);
            }
        }
        SLP(14, ds.pdf != 0.f);
        if (ds.pdf != 0.f) {
            V3 wo = to_local(si.sh, ds.d);
            float bsdf_val, bsdf_pdf;
            bsdf_eval_pdf(mat, si.wi, wo, bsdf_val, bsdf_pdf);
            float mis = ds.delta ? 1.f : mis_weight(ds.pdf, bsdf_pdf);
            sh.c = mis * s.throughput * bsdf_val * emitter_val;
            sh.c_im = 0.f;
            if (iq) {
                float re, im;
                path_phasor((s.t_rx - s.time) * sc.c + ds.dist, s.lambda0, re, im);
                sh.c_im = sh.c * im;
                sh.c = sh.c * re;
            }
            sh.o = si.p;
            sh.d = ds.d;
            sh.mint = kRayEpsilon * (1.f + hmax_abs(si.p));
            sh.maxt = ds.dist * (1.f - kShadowEpsilon);
            sh.want = true;
            ++s.n_rays;
        }
        if (is_range) s.aux += si.t;                                // pathlength.cpp:209
    }
    BF_SHADEPROF_STAMP(spf_t3);
    (void) next_1d(s.rng);                                          // sample1 (unused by these BSDFs)
    float s2x = next_1d(s.rng), s2y = next_1d(s.rng);
    BSDFSample bs;
    float bsdf_val = bsdf_sample(mat, si.wi, s2x, s2y, bs);
    s.throughput = s.throughput * bsdf_val;
    if (s.throughput == 0.f) {
        s.flags |= kFlagTermPending;
        s.rmint = BF_INF;      // empty interval: no closest-hit query for this slot
        s.rmaxt = 0.f;
        return true;
    }
    s.eta *= bs.eta;
    // si.spawn_ray — interaction.h:61-64
    s.ro = si.p;
    s.rd = to_world(si.sh, bs.wo);
    s.rmint = (1.f + hmax_abs(si.p)) * kRayEpsilon;
    s.rmaxt = BF_INF;
    s.prev_p = si.p;
    s.bs_pdf = bs.pdf;
    ++s.n_rays;
#ifdef BF_VERTEX_PROF
    if (spf) {
        const unsigned long long spf_t4 = __builtin_amdgcn_s_memtime();
        spf->si += spf_t1 - spf_t0;
        spf->head += spf_t2 - spf_t1;
        spf->nee += spf_t3 - spf_t2;
        spf->bsdf += spf_t4 - spf_t3;
    }
#endif
    return true;
}

// ---------------------------------------------------------------------------
// film / ADC
// ---------------------------------------------------------------------------
// The two destinations are named by address space: through generic pointers the compiler merges both branches into ONE
// flat_atomic_add_f32 on a selected pointer — a vector-memory instruction that finds out per lane whether it addresses LDS
// (round 4: every histogram sample of rounds 1-3 went that way, 20 % of wf_shade's cycles).
typedef __attribute__((address_space(3))) float *lds_float_ptr;
typedef __attribute__((address_space(1))) float *glb_float_ptr;
BF_DEV void lds_add(float *p, float v) {          // ds_add_f32
    (void) __hip_atomic_fetch_add((lds_float_ptr) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
BF_DEV void glb_add(float *p, float v) {          // global_atomic_add_f32
    (void) __hip_atomic_fetch_add((glb_float_ptr) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#ifndef BF_FLAT_HIST
#define BF_FLAT_HIST 0
#endif
BF_DEV void hist_add(float *s_hist, float *g_hist, bool lds, uint32_t idx, float v) {
#if BF_FLAT_HIST
    if (lds)
        atomicAdd(&s_hist[idx], v);
    else
        atomicAdd(&g_hist[idx], v);
#else
    if (lds)
        lds_add(s_hist + idx, v);
    else
        glb_add(g_hist + idx, v);
#endif
}
// Where the samples of render `render` go: plain launch = the histogram; batched launch = block `render` of it (LDS and
// global alike); rolling sequence = the render's own histogram (DRoll::hist), privatised in LDS only for the newest
// kRollWindow renders (the few stragglers of older renders take global atomics).
struct HistDst {
    float *s, *g;
    bool lds;
};
BF_DEV HistDst hist_dst(const DLaunch &lp, uint32_t render, float *s_hist, float *g_hist, bool lds_hist) {
    HistDst h;
    if (lp.roll) {
        h.lds = lds_hist && render >= lp.roll_lo;
        h.g = h.lds ? nullptr : lp.roll[render & (kRollRing - 1u)].hist;      // (no descriptor fetch for the samples that stay in LDS)
        h.s = s_hist + (h.lds ? (render - lp.roll_lo) * lp.n_chan : 0u);
    } else {
        const uint32_t hb = lp.batch != 0u ? render * lp.n_chan : 0u;
        h.g = g_hist + hb;
        h.s = s_hist + hb;
        h.lds = lds_hist;
    }
    return h;
}

struct FilmAcc {
    float X, Y, Z, A, W;    // base channels of the 1x1 film (render modes)
    uint32_t invalid;
    uint32_t n_put;         // paths binned by this lane (CTR_FILM: the loud "no path was lost" check of the host)
};

// ImageBlock::put / SignalBlock::put, the branch for reconstruction filters wider than a pixel (imageblock.cpp:115-165,
// signalblock.cpp:117-161): every channel of the sample goes, times wy * wx, to the cells within `radius` of its position;
// the weights are the host's discretised filter (bf_rfilter), looked up as eval_discretized does (rfilter.h:62-65).
// The sample sits in the block that rendered it: offset `off` (a multiple of bf_rfilter.block_size, integrator.cpp:139-142;
// the ADC is one block), size `bsz`, a border of filt_border cells around it; what lands in the border or outside the
// storage is dropped when the block is added to the film (imageblock.cpp:56-74 put(block): accumulate_2d clips).
struct WideSample {
    float posx, posy;         // pos_ as put() receives it
    int offx, offy;           // block offset
    int bw, bh;               // block size (without border)
    int W, H;                 // storage extent in cells
    int cropx, cropy;         // block coordinate of the storage's first cell (the ADC's window offset; 0 for films)
    uint32_t C;               // channels per cell
    float v0, v1, v2, v3, v4; // base channels (scalars, no indexed array: the hot kernels must stay free of scratch)
    bool five;                // five base channels (render modes) or three (receive modes)
    int ech;                  // channel of the first candidate extra bin (may be negative: that candidate is masked out)
    uint32_t emask;           // candidates that take the sample (bit i: bin i of the three)
    bool e3;                  // three values per bin (time mode) or one
    float e0, e1, e2;
};
BF_DEV float filt_eval(CSensor &se, float x) {
    const int idx = min((int) __builtin_fabsf(x * se.filt_scale), 31);
    return se.filt_tab[idx];
}
BF_DEV void put_wide_add(const HistDst &hd, uint32_t idx, float v, float w) {
    const float a = v * w;          // value[k] * weight (imageblock.cpp:160)
    if (a != 0.f) hist_add(hd.s, hd.g, hd.lds, idx, a);
}
BF_DEV void put_wide(CSensor &se, const WideSample &ws, const HistDst &hd) {
    const int border = (int) se.filt_border, n = (int) se.filt_n;
    const float r = se.filt_radius;
    // pos = pos_ - (m_offset - m_border_size + .5f)
    const float px = ws.posx - ((float) (ws.offx - border) + .5f), py = ws.posy - ((float) (ws.offy - border) + .5f);
    const int sx = ws.bw + 2 * border, sy = ws.bh + 2 * border;
    const int lox = max((int) __builtin_ceilf(px - r), 0), loy = max((int) __builtin_ceilf(py - r), 0);
    const int hix = min((int) __builtin_floorf(px + r), sx - 1), hiy = min((int) __builtin_floorf(py + r), sy - 1);
    const float basex = (float) lox - px, basey = (float) loy - py;
    const int estep = ws.e3 ? 3 : 1;
    for (int yr = 0; yr < n; ++yr) {
        const int y = loy + yr;
        if (y > hiy) break;
        const float wy = filt_eval(se, basey + (float) yr);
        const int gy = ws.offy + y - border - ws.cropy;
        for (int xr = 0; xr < n; ++xr) {
            const int x = lox + xr;
            if (x > hix) break;
            const float w = wy * filt_eval(se, basex + (float) xr);
            const int gx = ws.offx + x - border - ws.cropx;
            if (gx < 0 || gx >= ws.W || gy < 0 || gy >= ws.H) continue;
            const uint32_t cell = ws.C * ((uint32_t) gy * (uint32_t) ws.W + (uint32_t) gx);
            put_wide_add(hd, cell + 0u, ws.v0, w);
            put_wide_add(hd, cell + 1u, ws.v1, w);
            put_wide_add(hd, cell + 2u, ws.v2, w);
            if (ws.five) {
                put_wide_add(hd, cell + 3u, ws.v3, w);
                put_wide_add(hd, cell + 4u, ws.v4, w);
            }
            for (int i = 0; i < 3; ++i) {
                if (!(ws.emask >> i & 1u)) continue;
                const uint32_t c = cell + (uint32_t) (ws.ech + i * estep);
                put_wide_add(hd, c, ws.e0, w);
                if (ws.e3) {
                    put_wide_add(hd, c + 1u, ws.e1, w);
                    put_wide_add(hd, c + 2u, ws.e2, w);
                }
            }
        }
    }
}
// position_sample of a path's render_sample (integrator.cpp:263): the pixel it was drawn for and the first next_2d of its stream
BF_DEV void film_position(const DLaunch &lp, const PathState &s, uint32_t &px, uint32_t &py, float &fx, float &fy) {
    uint64_t seed = lp.seed, path_offset = lp.path_offset, path_i = s.path_i;
    if (lp.batch != 0u) {
        path_i -= (uint64_t) s.render * lp.batch_paths;
        if (lp.roll) {
            const DRoll &rr = lp.roll[s.render & (kRollRing - 1u)];
            seed = rr.seed;
            path_offset = rr.path_offset;
        } else if (lp.batch_seeds) {
            seed = lp.batch_seeds[s.render];
        }
    }
    Rng rng;
    pcg_seed(rng, seed + path_offset + path_i);
    fx = next_1d(rng);
    fy = next_1d(rng);
    px = py = 0u;
    if (lp.spp) {
        const uint64_t q = (lp.path_offset + path_i) / lp.spp;
        px = (uint32_t) (q % lp.film_w);
        py = (uint32_t) (q / lp.film_w);
    }
}

template <int RX = 2>
BF_DEV void film_put(const DScene &sc0, const DLaunch &lp, const PathState &s, float *s_hist, float *g_hist, bool lds_hist,
                     FilmAcc &acc, bf_path_record *records) {
    const DScene sc = path_scene<RX>(sc0, lp, s.render);
    const bool valid = (s.flags & kFlagValid) != 0;
#ifndef BF_NO_FILM_CTR
    ++acc.n_put;
#endif
    float rec_L, rec_aux;
    float *const s_base = s_hist + lp.base_off;                               // rolling launches: base-channel table (DLaunch::base_off)
    const HistDst hd = hist_dst(lp, s.render, s_hist, g_hist, lds_hist);      // this render's block of the histogram
    s_hist = hd.s;
    g_hist = hd.g;
    lds_hist = hd.lds;
    constexpr bool wide = (RX & kWide) != 0;       // reconstruction filter wider than a pixel (uniform; the radar scenes use box)
    if (mode_receive<RX>(lp)) {
        // receive_sample tail — integrator.cpp:1625-1665; SignalBlock::put — signalblock.cpp:162-169
        CSensor &se = c_sensor(sc);
        float tf0 = s.t_rx - se.adc_sampling_start;
        float tf1 = freq_of(sc.c, rare<RX>(lp.doppler != 0u) ? s.lambda0 + s.dlambda : s.lambda0);
        // "mix_resample": |f_after - f_rx| (integrator.cpp:1590-1601); with a re-sampling transmitter lambda0 is the wavelength the
        // path ENDS with (ray_.wavelengths = si.wavelengths, pathtimefrequency.cpp:451) and dlambda the one the receiver drew
        if (rare<RX>(lp.mix != 0u)) tf1 = __builtin_fabsf(tf1 - freq_of(sc.c, rare<RX>(lp.resample != 0u) ? s.dlambda : s.lambda0));
        tf0 *= (float) se.t_bins / se.t_bandwidth;
        tf1 *= (float) se.f_bins / se.f_bandwidth;
        float L = __builtin_fabsf(s.aux) * s.result;          // aux holds ray_weight in receive mode
        float a0 = valid ? 4.f * L : 0.f;                     // hsum over 4 identical spectral lanes
        float a1 = valid ? 1.f : 0.f;
        if (lp.iq) a1 = valid ? 4.f * (__builtin_fabsf(s.aux) * s.phase) : 0.f;   // I, Q, W instead of Y, A, W
        bool ok = __builtin_isfinite(a0) && __builtin_isfinite(a1);
        // PhaseIntegrator::sample (phase.cpp:93-141): S{k}.Y takes hsum(L), before the receiver weight,
        // iff rect((phase - centre_k) / width) > 0; evaluated exactly as written there for the (at most
        // three) candidate bins around phase / width
        const uint32_t P = (RX & kLean) ? 0u : lp.phase_bins;
        const float pv = valid ? 4.f * s.result : 0.f;
        int pk0 = 0;
        uint32_t pmask = 0u;            // bit i: bin pk0 - 1 + i takes the sample
        if (P) {
            const float two_pi = 6.28318530717958647692f;
            const float width = two_pi / (float) (int) P;
            float phase = __builtin_fmodf(valid ? 0.f + s.phase : 0.f, two_pi);
            phase += (phase < 0.f) ? two_pi : 0.f;
            const float fk = __builtin_floorf(phase / width);             // NaN phases (a miss: -inf) select no bin
            pk0 = (fk >= 0.f && fk < (float) P) ? (int) fk : ((fk >= (float) P) ? (int) P - 1 : 0);
            for (int i = 0; i < 3; ++i) {
                const int k = pk0 - 1 + i;
                if (k < 0 || k >= (int) P) continue;
                float centre = (float) ((double) width * ((double) k + 0.5));
                if (jabs((phase - centre) / width) < 0.5f) pmask |= 1u << i;
            }
            ok = ok && __builtin_isfinite(pv);
        }
        // pos = tf - (m_offset - m_border_size + .5f), lo = ceil(pos - .5f) (signalblock.cpp:115,163): the block sits at the ADC's
        // window offset (integrator.cpp:627; 0 without a window), the histogram is the window
        const uint32_t wot = rare<RX>(se.win_off_t != 0u) ? se.win_off_t : 0u, wof = rare<RX>(se.win_off_f != 0u) ? se.win_off_f : 0u;
        float lx = __builtin_ceilf((tf0 - ((float) wot + .5f)) - .5f), ly = __builtin_ceilf((tf1 - ((float) wof + .5f)) - .5f);
        if (wide) {
            if (ok) {
                WideSample ws;
                ws.posx = tf0;
                ws.posy = tf1;
                ws.offx = (int) wot;                         // receive(): ONE block, the ADC's window (integrator.cpp:624-628)
                ws.offy = (int) wof;
                ws.bw = ws.W = (int) lp.bins;
                ws.bh = ws.H = (int) lp.bins_y;
                ws.cropx = (int) wot;
                ws.cropy = (int) wof;
                ws.C = 3u + P;
                ws.v0 = a0;
                ws.v1 = a1;
                ws.v2 = 1.f;
                ws.v3 = ws.v4 = 0.f;
                ws.five = false;
                ws.ech = 3 + pk0 - 1;
                ws.emask = pmask;
                ws.e3 = false;
                ws.e0 = pv;
                ws.e1 = ws.e2 = 0.f;
                put_wide(se, ws, hd);
            } else {
                ++acc.invalid;
            }
        } else if ((ok = ok && lx >= 0.f && lx < (float) lp.bins && ly >= 0.f && ly < (float) lp.bins_y)) {
            uint32_t off = (3u + P) * ((uint32_t) ly * lp.bins + (uint32_t) lx);
            if (a0 != 0.f) hist_add(s_hist, g_hist, lds_hist, off + 0u, a0);
            if (a1 != 0.f) hist_add(s_hist, g_hist, lds_hist, off + 1u, a1);
            hist_add(s_hist, g_hist, lds_hist, off + 2u, 1.f);
            for (int i = 0; i < 3; ++i)
                if ((pmask >> i & 1u) && pv != 0.f) hist_add(s_hist, g_hist, lds_hist, off + 3u + (uint32_t) (pk0 - 1 + i), pv);
            acc.W += 1.f;
        } else {
            ++acc.invalid;
        }
        rec_L = a0;
        rec_aux = lp.iq ? a1 : s.t_rx - se.adc_sampling_start;
    } else {
        const bool is_range = lp.mode == BF_MODE_RANGE, is_time = rare<RX>(lp.mode == BF_MODE_TIME);
        // ray weight: fluxmeter.cpp:84 (wav_weight * pi), irradiancemeter.cpp:82 (wav_weight * pi / surface_area),
        // perspective.cpp:198 (wav_weight)
        float sensor_w = 1.f;
        if (rare<RX>(c_sensor(sc).type == BF_SENSOR_FLUXMETER)) sensor_w = 1.f * kPi;
        if (rare<RX>(c_sensor(sc).type == BF_SENSOR_IRRADIANCEMETER)) sensor_w = 1.f * kPi / c_rects(sc)[c_sensor(sc).rect].area;
        float L = sensor_w * s.result;                        // integrator.cpp:286
        float X, Y, Z;
        if (lp.color_mode == BF_COLOR_RGB)
            srgb_to_xyz_grey(L, X, Y, Z);
        else
            X = Y = Z = L;
        float a0 = s.result, a1 = s.result, a2 = s.result;    // AOVs see the unweighted radiance
        if (is_time && lp.color_mode == BF_COLOR_RGB) srgb_to_xyz_grey(s.result, a0, a1, a2);
        bool ok = (s.flags & kFlagFilmOk) && __builtin_isfinite(X) && __builtin_isfinite(Y) && __builtin_isfinite(Z);
        if (is_range || is_time) ok = ok && __builtin_isfinite(a0) && __builtin_isfinite(a1) && __builtin_isfinite(a2);
        // multi-pixel film: every channel of the sample goes to its pixel's block of the histogram; the 1 x 1 film
        // keeps the five base channels in registers until film_flush
        uint32_t pix = 0u;
        // the five base channels of a 1 x 1 film are summed in registers while every lane of the wave feeds the same
        // histogram: a plain launch, or the newest render of a rolling sequence
        const bool use_acc = lp.batch == 0u || (lp.roll != nullptr && s.render == lp.roll_newest);
        if (rare<RX>(lp.spp != 0u)) {
            const uint64_t q = (lp.path_offset + s.path_i) / lp.spp;
            const uint32_t px = (uint32_t) (q % lp.film_w) - ((s.flags & kFlagFilmLeft) ? 1u : 0u);
            const uint32_t py = (uint32_t) (q / lp.film_w) - ((s.flags & kFlagFilmUp) ? 1u : 0u);
            pix = (py * lp.film_w + px) * lp.chan_px;          // only used when kFlagFilmOk
        }
        if (wide) {
            // the filtered branch takes every finite sample (where it lands is decided cell by cell)
            bool fin = __builtin_isfinite(X) && __builtin_isfinite(Y) && __builtin_isfinite(Z);
            if (is_range || is_time) fin = fin && __builtin_isfinite(a0) && __builtin_isfinite(a1) && __builtin_isfinite(a2);
            if (fin) {
                WideSample ws;
                uint32_t qx, qy;
                float fx, fy;
                film_position(lp, s, qx, qy, fx, fy);
                ws.cropx = (int) c_sensor(sc).crop_x;         // blocks tile the crop window from its offset (spiral.cpp: offset += m_offset)
                ws.cropy = (int) c_sensor(sc).crop_y;
                ws.posx = (float) (qx + (uint32_t) ws.cropx) + fx;
                ws.posy = (float) (qy + (uint32_t) ws.cropy) + fy;
                const uint32_t B = c_sensor(sc).filt_block;
                const int bx0 = B ? (int) (qx / B * B) : 0, by0 = B ? (int) (qy / B * B) : 0;
                ws.offx = ws.cropx + bx0;
                ws.offy = ws.cropy + by0;
                ws.W = (int) lp.film_w;
                ws.H = (int) lp.film_h;
                ws.bw = B ? min((int) B, ws.W - bx0) : ws.W;
                ws.bh = B ? min((int) B, ws.H - by0) : ws.H;
                ws.C = lp.chan_px;
                ws.v0 = X;
                ws.v1 = Y;
                ws.v2 = Z;
                ws.v3 = valid ? 1.f : 0.f;
                ws.v4 = 1.f;
                ws.five = true;
                ws.emask = 0u;
                ws.e3 = is_time;
                ws.ech = 5;
                ws.e0 = a0;
                ws.e1 = a1;
                ws.e2 = a2;
                if (is_range || is_time) {
                    float w = lp.bin_width;
                    int k = (int) __builtin_floorf(s.aux / w);
                    ws.ech = 5 + (k - 1) * (is_time ? 3 : 1);
                    for (int i = k - 1; i <= k + 1; ++i) {
                        if (i < 0 || i >= (int) lp.bins) continue;
                        float lo = (float) i * w, hi = (float) i * w + w;
                        if (s.aux >= lo && s.aux < hi) ws.emask |= 1u << (i - (k - 1));
                    }
                }
                put_wide(c_sensor(sc), ws, hd);
            } else {
                ++acc.invalid;
            }
        } else if (ok) {
            if (rare<RX>(lp.spp != 0u)) {
                if (X != 0.f) hist_add(s_hist, g_hist, lds_hist, pix + 0u, X);
                if (Y != 0.f) hist_add(s_hist, g_hist, lds_hist, pix + 1u, Y);
                if (Z != 0.f) hist_add(s_hist, g_hist, lds_hist, pix + 2u, Z);
                if (valid) hist_add(s_hist, g_hist, lds_hist, pix + 3u, 1.f);
                hist_add(s_hist, g_hist, lds_hist, pix + 4u, 1.f);
            } else if (!use_acc) {
                // batched launch: the wave's lanes hold paths of different renders, so the base channels cannot be
                // summed in registers; each sample goes to its render's block.  A rolling sequence's older renders (behind
                // the LDS window) would hit the same five GLOBAL addresses from every wave: they go through the
                // workgroup's base-channel table instead (kRollBase renders; film_flush adds it up)
                float *bs = s_hist;
                bool bl = lds_hist;
                if (lp.roll && !lds_hist && lp.roll_newest - s.render < kRollBase) {
                    bs = s_base + 5u * (lp.roll_newest - s.render);
                    bl = true;
                }
                if (X != 0.f) hist_add(bs, g_hist, bl, 0u, X);
                if (Y != 0.f) hist_add(bs, g_hist, bl, 1u, Y);
                if (Z != 0.f) hist_add(bs, g_hist, bl, 2u, Z);
                if (valid) hist_add(bs, g_hist, bl, 3u, 1.f);
                hist_add(bs, g_hist, bl, 4u, 1.f);
            } else {
                acc.X += X;
                acc.Y += Y;
                acc.Z += Z;
                acc.A += valid ? 1.f : 0.f;
                acc.W += 1.f;
            }
            if (is_range || is_time) {
                // range.cpp:141-161 / time.cpp:134-153: bin i takes the sample iff
                // (float)i*w <= aux < (float)i*w + w, evaluated exactly as written
                // there for the (at most three) candidate bins
                float w = lp.bin_width;
                int k = (int) __builtin_floorf(s.aux / w);
                for (int i = k - 1; i <= k + 1; ++i) {
                    if (i < 0 || i >= (int) lp.bins) continue;
                    float lo = (float) i * w, hi = (float) i * w + w;
                    if (s.aux >= lo && s.aux < hi) {
                        if (is_range) {
                            if (a0 != 0.f) hist_add(s_hist, g_hist, lds_hist, pix + 5u + (uint32_t) i, a0);
                        } else if (a0 != 0.f || a1 != 0.f || a2 != 0.f) {
                            hist_add(s_hist, g_hist, lds_hist, pix + 5u + 3u * (uint32_t) i + 0u, a0);
                            hist_add(s_hist, g_hist, lds_hist, pix + 5u + 3u * (uint32_t) i + 1u, a1);
                            hist_add(s_hist, g_hist, lds_hist, pix + 5u + 3u * (uint32_t) i + 2u, a2);
                        }
                    }
                }
            }
        } else {
            ++acc.invalid;
        }
        rec_L = L;
        rec_aux = s.aux;
    }
    uint64_t rec_i = s.path_i;
    if (lp.roll) {              // rolling sequence: the render's own record array, indexed by the local path
        records = lp.has_records ? lp.roll[s.render & (kRollRing - 1u)].records : nullptr;     // (no per-path fetch of a null pointer)
        rec_i -= (uint64_t) s.render * lp.batch_paths;
    }
    if (records) {
        // bf_path_record {L, aux, valid, n_rays} as one 16-byte store through a global-address-space pointer (a rolling render's
        // array comes out of its descriptor: a generic pointer would make this a flat store)
        static_assert(sizeof(bf_path_record) == 16, "one 16-byte store per record");
        typedef __attribute__((address_space(1))) bf_f4 *glb_f4_ptr;
        bf_f4 r;
        r.x = rec_L;
        r.y = rec_aux;
        r.z = __uint_as_float(valid ? 1u : 0u);
        r.w = __uint_as_float(s.n_rays);
        ((glb_f4_ptr) records)[rec_i] = r;
    }
}

// wave-reduce the base channels into the histogram (render modes only) and
// flush the LDS-privatised histogram: one global atomic per non-empty bin
template <int RX = 2> BF_DEV void film_flush(const DLaunch &lp, FilmAcc &acc, float *s_hist, float *g_hist, bool lds_hist, int tid) {
    const int lane = tid & 63;
    if (!mode_receive<RX>(lp)) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            acc.X += __shfl_down(acc.X, off);
            acc.Y += __shfl_down(acc.Y, off);
            acc.Z += __shfl_down(acc.Z, off);
            acc.A += __shfl_down(acc.A, off);
            acc.W += __shfl_down(acc.W, off);
        }
        if (lane == 0 && acc.W != 0.f && !lp.spp && (lp.batch == 0u || lp.roll != nullptr)) {
            const HistDst hd = hist_dst(lp, lp.roll ? lp.roll_newest : 0u, s_hist, g_hist, lds_hist);
            hist_add(hd.s, hd.g, hd.lds, 0, acc.X);
            hist_add(hd.s, hd.g, hd.lds, 1, acc.Y);
            hist_add(hd.s, hd.g, hd.lds, 2, acc.Z);
            hist_add(hd.s, hd.g, hd.lds, 3, acc.A);
            hist_add(hd.s, hd.g, hd.lds, 4, acc.W);
        }
    }
#ifdef BF_NO_FLUSH      // developer timing probe: what the flush of the LDS histograms costs (the histograms stay empty)
    return;
#endif
    if (lds_hist) {
        __syncthreads();
        if (lp.roll) {
            // one block per render of the LDS window, each flushed into that render's own histogram
            for (uint32_t r = lp.roll_lo; r <= lp.roll_newest; ++r) {
                float *gh = lp.roll[r & (kRollRing - 1u)].hist;
                const float *sh = s_hist + (r - lp.roll_lo) * lp.n_chan;
                for (uint32_t i = tid; i < lp.n_chan; i += kBlock) {
                    float v = sh[i];
                    if (v != 0.f) glb_add(gh + i, v);
                }
            }
        } else {
            for (uint32_t i = tid; i < lp.n_chan_all; i += kBlock) {
                float v = s_hist[i];
                if (v != 0.f) glb_add(g_hist + i, v);
            }
        }
    }
    if (lp.roll && !mode_receive<RX>(lp)) {
        // base-channel table of the renders behind the window: entry [age][channel], age = roll_newest - render
        if (!lds_hist) __syncthreads();
        const float *s_base = s_hist + lp.base_off;
        for (uint32_t i = tid; i < 5u * kRollBase; i += kBlock) {
            const float v = s_base[i];
            const uint32_t age = i / 5u;
            if (v != 0.f && age <= lp.roll_newest) glb_add(lp.roll[(lp.roll_newest - age) & (kRollRing - 1u)].hist + (i - 5u * age), v);
        }
    }
}

}  // namespace bfd
