// Device-resident scene layout shared by the host API (bf_api) and the
// kernels (bf_kernels.hip).  Everything is read-only during a render.
//
// HBM layout (DESIGN.md "Data layout"):
//   nodes   : bf::Node4[n_nodes]     128 B each, four child boxes (SoA) + four child references
//   tris    : float4[kTriStride * n_tris]   48 B per triangle (kTriStride = 3), BVH leaf order:
//               q0 = (p0.xyz, bits(global prim index))
//               q1 = (p1.xyz, bits(shape index))
//               q2 = (p2.xyz, tag: normals / texcoords bits, material, emitter)
//             (-DBF_TRI_STRIDE=4 pads every record to 64 bytes so that none straddles a sector: measured, no difference
//              on any config — profiles/r03_tri_record_ab.txt — so the 48-byte records of SURVEY 8d stay)
//   normals : float4[3 * n_tris]     only if some mesh carries vertex normals
//   uvs     : float4[n_tris]         (uv1 - uv0, uv2 - uv0), only if some mesh carries texture coordinates
//   rects, shapes, materials, emitters : small tables (scenes hold a handful)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/beifong_hip.h"

namespace bfd {

#ifndef BF_TRI_STRIDE
#define BF_TRI_STRIDE 3
#endif
constexpr uint32_t kTriStride = BF_TRI_STRIDE;      // float4 per triangle record
constexpr int kBlock = 256;          // threads per workgroup (4 waves)
constexpr int kStackDepth = 32;      // per-lane traversal stack entries in LDS
constexpr int kMaxLdsHist = 12288;   // floats of LDS-privatised histogram (48 KiB)
constexpr int kWideStack = 512;      // row-traversal stack entries per 16-lane row (aliases the 16 lanes' LDS stack columns)
constexpr uint32_t kTopNodes = 85;   // == bf::kTopNodes (bf_bvh.h): nodes of the tree's top levels kept in LDS by wf_trace

struct DRect {
    float to_world[12];   // 3x4 row-major affine
    float to_object[12];
    float s[3], t[3], n[3];   // Rectangle::update frame (dp_du, dp_dv, normal)
    float inv_area;
    float area;           // Rectangle::surface_area()
    uint32_t shape;
    uint32_t prim;        // global primitive index
    uint32_t material;    // of the carrying shape
    int32_t emitter;      // of the carrying shape, or -1
};

static_assert(sizeof(DRect) == 156 && offsetof(DRect, s) == 96 && offsetof(DRect, n) == 120 && offsetof(DRect, shape) == 140 &&
                  offsetof(DRect, material) == 148 && offsetof(DRect, emitter) == 152,
              "make_si reads the LDS copy of a rectangle by dword index");

struct DShape {
    uint32_t type, material;
    int32_t emitter;
    int32_t rect;         // index into rects or -1
    float velocity[12];   // Shape "velocity" transform (shape.cpp:42), 3x4 row-major: BF_FLAG_DOPPLER only
};

// bf_material padded to 48 bytes: a vertex loads its material as three aligned 16-byte words (bf_device_core.h: load_material)
struct alignas(16) DMaterial {
    bf_material m;
    uint32_t pad;
};
static_assert(sizeof(bf_material) == 44 && sizeof(DMaterial) == 48, "load_material reads eleven dwords of a 48-byte record");

struct DEmitter {
    uint32_t type;
    int32_t rect;         // area types: rectangle index
    float to_world[12], to_object[12];
    float radiance, cutoff, beam, inv_transition, cos_cutoff, cos_beam;
    // wigner transmitter signal model (wignertransmitter.cpp:53-110)
    uint32_t signal_type;
    float amplitude, freq_centre, freq_ext, pulse_len, prf, gain;
    uint32_t resample;    // m_resample_freq (wignertransmitter.cpp:211-221, 430-441): eval / sample_direction re-draw the path's wavelength from the signal
    // phased array (phasedtransmitter.cpp:108-165): n_velems virtual elements, BF_VELEM_FLOATS floats each (device copy)
    const float *velems;
    uint32_t n_velems;
    float wid[3];
};

// Kernel variant word V — the template parameter (`RX`) of the path logic and of the kernels built on it:
//   bits 0-1  mode class: 0 render modes (path / range / time), 1 receive modes, 2 decided at run time (tail, one-kernel variant)
//   kWide     the sensor's reconstruction filter is wider than a pixel (DLaunch::wide)
//   kLean     scene and launch fit the LEAN PROFILE (DLaunch::lean, bf_api.cpp: lean_profile) — what every radar scene of the
//             reference's scripts and all BASELINE configs use: ONE emitter of an area type (area light, area / Wigner
//             transmitter: no spot, point or phased-array source), no texture coordinates, a perspective camera (render
//             modes) or the omnidirectional receiver (receive modes), the 1 x 1 film, no time-resolved mode, no phase bins, no
//             Doppler hook, no mix_resample.  The code for everything outside the profile is compiled out: wf_shade 168 -> 159
//             VGPRs, a quarter fewer scalar spills, -4.5 % of its time on C2 (profiles/r03_lean_variant_ab.txt).
//   kMulti    a rolling sequence whose ENDPOINTS moved between its renders (bf_scene_update_endpoints joined it: DLaunch::multi):
//             every path reads the rectangle / shape / emitter / material / sensor tables of ITS render through the
//             sequence's descriptor ring (DRoll) — per-lane pointers, so those reads are vector loads — instead of the
//             launch's kernel arguments
constexpr int kModeMask = 3, kWide = 4, kLean = 8, kMulti = 16;
// rare<V>(c): a condition the lean profile guarantees to be false
template <int V> __device__ __forceinline__ constexpr bool rare(bool c) { return (V & kLean) ? false : c; }

struct DSensor {
    uint32_t type;
    int32_t rect;
    float to_world[12];
    float sample_to_camera[16];
    float near_clip, far_clip, shutter_open, shutter_open_time;
    // receiver + ADC (receiver.cpp:16-62, adc.cpp:18-46, wignerreceiver.cpp)
    float adc_sampling_start, adc_sampling_time;
    uint32_t t_bins, f_bins;
    float t_bandwidth, f_bandwidth;
    float freq_centre, freq_ext, gain;
    uint32_t rx_sig_is_delta;
    uint32_t rx_signal;       // "mix_resample" on the Wigner / phased receiver: its local oscillator (bf_sensor::rx_signal_type, ...)
    float rx_pulse_len, rx_prf, rx_amplitude;
    const float *velems;      // BF_RECEIVER_PHASED (phasedreceiver.cpp:115-172)
    uint32_t n_velems;
    float wid[3];
    // reconstruction filter of the film / ADC (bf_rfilter); filt_n = 0: put()'s box branch
    uint32_t filt_n;          // ceil((radius - 2 RayEpsilon) * 2): weights per axis (imageblock.cpp:121)
    uint32_t filt_border, filt_block;
    float filt_radius, filt_scale;
    float filt_tab[32];
    uint32_t crop_x, crop_y;            // film crop offset (render modes): position sample = (pixel + crop) + next_2d
    uint32_t win_off_t, win_off_f;      // ADC window offset (receive modes; DLaunch::bins / bins_y are the window's size)
};

struct DScene {
    const float4 *qnodes;     // the same nodes quantised to 64 bytes (bf_bvh.h: Node4Q, 4 float4 each): wf_trace's; nullptr = use `nodes`
    const float4 *nodes;      // 8 float4 per node
    const float4 *tris;       // 3 float4 per triangle
    const float4 *normals;    // 3 float4 per triangle or nullptr
    const float4 *uvs;        // 1 float4 per triangle or nullptr
    const DRect *rects;
    const DShape *shapes;
    const DMaterial *materials;
    const DEmitter *emitters;
    uint32_t n_tris, n_rects, n_emitters, n_nodes;
    int32_t root;             // child reference of the BVH root
    // traversal-stack overflow: entry k of thread g at spill[k * spill_stride + g]; every kernel launches at most
    // spill_stride threads and keeps its first 16 (wavefront) or 32 entries in LDS
    int *spill;
    uint32_t spill_stride;
    uint32_t stack_need;      // BVH4::stack_need: kernels whose LDS stack holds that many entries compile the overflow path out
    float c, lambda_min, lambda_max;   // MTS_C, MTS_WAVELENGTH_MIN/MAX as run-time physics
    // sixteen-wide collapse of the same tree (bf_bvh.h: Node16, 32 float4 per node) for the tail kernel's row traversal;
    // nullptr when the scene has no triangles or its worst-case stack exceeds kWideStack
    const float4 *wnodes;
    int32_t wroot;
    uint32_t n_wnodes;
    uint32_t wrows_log;       // log2 of the most rows a gang may have: 16 * rows * depth stack entries must fit kWideStack
    const DSensor *sensor;    // device copy (kept out of the kernel arguments: 44 dwords of scalar registers)
    // LDS copies of the tables a vertex reads through a PER-LANE index (its material; the rectangle it hit): wf_shade and the
    // tail kernel copy them behind their histogram when the scene is small enough (tab_cache, set by the host, reserves
    // kTabBytes of dynamic LDS) and switch tab_on in their own copy of this struct (bf_device_core.h: load_tables_lds)
    uint32_t n_materials;
    uint32_t tab_cache;       // host: 1 = n_materials <= kTabMaxMaterials and n_rects <= kTabMaxRects
    uint32_t tab_on;          // device only: the tables are in LDS at byte offsets lds_mat / lds_rect of the dynamic segment
    uint32_t lds_mat, lds_rect;
};
constexpr uint32_t kTabMaxMaterials = 16, kTabMaxRects = 8;
constexpr uint32_t kRectDwords = sizeof(DRect) / 4;       // 39: an odd stride, lanes at different rectangles fall into different banks
constexpr uint32_t kTabBytes = (kTabMaxMaterials * 48u + kTabMaxRects * (uint32_t) sizeof(DRect) + 15u) & ~15u;

// The scene's SMALL TABLES (rectangles, shapes, materials, emitters, the sensor record, the rolling ring) are read through
// constant-address-space pointers: the loads are invariant for the compiler, and one whose address is wave-uniform — the
// rectangle loop of presolve_ray, the only emitter of the lean profile, the sensor — becomes a scalar load (s_load into
// SGPRs: scalar cache, no vector-memory instruction, no vmcnt wait).  Through generic pointers every such read inside the
// persistent loops was a uniform VECTOR load followed by s_waitcnt vmcnt(0) (the kernels store in between, so the compiler
// could not prove the tables constant): ~50 serialised waits per shaded vertex in wf_shade.
#define BF_CAS __attribute__((address_space(4)))
typedef const BF_CAS DRect CRect;
typedef const BF_CAS DShape CShape;
typedef const BF_CAS DEmitter CEmitter;
typedef const BF_CAS DSensor CSensor;
template <class T> __device__ __forceinline__ const BF_CAS T *as_const(const T *p) { return (const BF_CAS T *) (uintptr_t) p; }
__device__ __forceinline__ CRect *c_rects(const DScene &sc) { return as_const(sc.rects); }
__device__ __forceinline__ CShape *c_shapes(const DScene &sc) { return as_const(sc.shapes); }
__device__ __forceinline__ CEmitter *c_emitters(const DScene &sc) { return as_const(sc.emitters); }
__device__ __forceinline__ CSensor &c_sensor(const DScene &sc) { return *as_const(sc.sensor); }

// One render of a ROLLING SEQUENCE (bf_render_device with BF_FLAG_ROLLING): what differs between the renders of a
// sequence.  The sequence is one batched launch whose path supply grows by one render per call: global path index
// g = render * batch_paths + local path, slot i renders g = i, i + n_slots, ... for as long as the supply lasts, so the
// long paths of one render are carried by the launches of the next ones instead of a tail kernel per render.
struct DRoll {
    uint64_t seed, path_offset;
    float *hist;                // this render's histogram (device)
    bf_path_record *records;    // this render's per-path records (device) or nullptr
    // the endpoint tables and physics the render was issued with (kMulti kernels: the radar turns between the frames of a
    // sweep — python_scripts/animated_trans_rad.py:307-384, Receive.ipynb cell 30 — while its earlier frames' long paths
    // are still in flight)
    const DRect *rects;
    const DShape *shapes;
    const DEmitter *emitters;
    const DMaterial *materials;
    const DSensor *sensor;
    float c, lambda_min, lambda_max;
    uint32_t pad;
};
static_assert(sizeof(DRoll) == 88, "descriptor ring entry");
constexpr uint32_t kRollRing = 256;    // renders per sequence (descriptor ring; the host flushes a longer one in between)
constexpr uint32_t kRollWindow = 4;    // newest renders whose histogram blocks a workgroup privatises in LDS; older ones take global atomics
constexpr uint32_t kRollBase = 32;     // newest renders whose five BASE channels (X, Y, Z, alpha, weight: one address each per render, so
                                       // global float atomics on them serialise at L2) a workgroup sums in LDS behind the window

struct DLaunch {
    uint32_t mode, color_mode;
    uint64_t n_paths, path_offset, seed;
    int32_t max_depth, rr_depth;
    uint32_t bins, bins_y, phase_bins;
    float bin_width, time_c;
    uint32_t n_chan;
    uint32_t lds_hist;        // 1: histogram privatised in LDS
    uint32_t iq;              // 1: BF_MODE_RECEIVE_IQ (mode is RECEIVE_RAW inside the kernels): contributions are phasors
    uint32_t film_w, film_h;  // render modes: film size in pixels (>= 1)
    uint32_t spp;             // paths per pixel; 0 = single-pixel film (every path samples pixel 0)
    uint32_t chan_px;         // channels per pixel (n_chan = film_w * film_h * chan_px)
    // Batched launch (bf_render_batch_device): `batch` renders of batch_paths paths each in ONE launch sequence.
    // n_paths = batch * batch_paths global path indices g; render k = g / batch_paths renders its local path
    // g - k * batch_paths with seed batch_seeds[k] into g_hist + k * n_chan, its meshes shifted by batch_offsets[k].
    uint32_t batch;                 // 0: a plain launch
    uint32_t n_chan_all;            // batch * n_chan (the LDS-privatised histogram covers all renders when it fits)
    uint64_t batch_paths;
    const uint64_t *batch_seeds;    // device [batch], or nullptr: `seed` for every render (common random numbers)
    const float4 *batch_offsets;    // device [batch] (x, y, z, -), or nullptr: meshes as built
    uint32_t mix;                   // BF_FLAG_MIX_RESAMPLE (receive modes): the ADC's frequency axis is the beat frequency
    uint32_t doppler;               // BF_FLAG_DOPPLER (receive modes): Shape::doppler shifts the path's wavelength
    uint32_t resample;        // some transmitter has resample_freq set: the path's wavelength changes on the way (PathState::lambda0) and
                              // PathState::dlambda holds the wavelength the RECEIVER sampled (what "mix_resample" subtracts)
    float box_slack;                // offsets only: node boxes widened by this much on the ray's side (bf_device_core.h: RayBox)
    // Rolling sequence (batch = 1, batch_paths = paths per render, n_paths = supply so far): descriptor ring [kRollRing]
    const DRoll *roll;              // nullptr: not a rolling launch
    uint32_t roll_newest;           // render index of the call this launch belongs to
    uint32_t roll_lo;               // oldest render whose histogram block is in the LDS window [roll_lo, roll_newest]
    uint32_t lds_floats;            // floats of dynamic LDS the histogram code zeroes: the privatised histogram (n_chan_all, if lds_hist)
                                    // followed, in a rolling launch, by the base-channel table [kRollBase][5] at float offset base_off
    uint32_t base_off;
    uint32_t has_records;           // rolling sequence: some render of it writes per-path records (DRoll::records)
    uint32_t lean;                  // 1: scene and launch fit the lean profile: the kernels' kLean variants
    uint32_t wide;                  // 1: the sensor's reconstruction filter is wider than a pixel (DSensor::filt_n != 0): the kernels' kWide variants
    uint32_t multi;                 // 1: the rolling sequence carries more than one version of the endpoint tables: the kernels' kMulti variants
    uint32_t count;                 // 1: somebody will read the statistics counters (BF_FLAG_STATS, or a render with stats_out): wf_shade and
                                    // the tail add theirs up — ten same-line atomics per WAVE, ~15 us each per launch at the end of a
                                    // persistent grid whose waves all finish together; 0: only the live count the host steers by
};

// device counters (uint64 each)
enum {
    CTR_NEXT_PATH = 0, CTR_CLOSEST, CTR_SHADOW, CTR_NODES, CTR_TRIS, CTR_INVALID, CTR_BOUNCES, CTR_TAIL_RAYS, CTR_STARTED, CTR_TRACED,
    CTR_NODES_LDS,
    // per-kernel breakdown of the algorithmic bytes (bench.py's roofline entries)
    CTR_TAIL_NODES,        // four-wide node visits of the tail kernel (BF_FLAG_STATS)
    CTR_TAIL_WNODES,       // sixteen-wide node visits of the tail kernel (BF_FLAG_STATS)
    CTR_TAIL_TRIS,         // triangle tests of the tail kernel (BF_FLAG_STATS)
    CTR_TAIL_BOUNCES,      // vertices shaded by the tail kernel
    CTR_SHADE_LOADS,       // path-state rows wf_shade read (slots visited)
    CTR_SHADE_STORES,      // path-state rows wf_shade wrote back
    CTR_SHADE_SHADOW,      // shadow requests wf_shade queued for wf_trace
    CTR_SHADE_RAYS,        // rays generated by wf_shade (each one tests the rectangles and the root node's boxes)
    CTR_FILM,              // paths binned (film_put calls): == the paths supplied once a render / sequence has ended, else paths were lost
    // the STICKY words (never cleared by a render, reported once): keep them last
    CTR_GUARD,             // rays dropped by wf_trace's iteration guard (a bug if ever non-zero: bf_render reports BF_ERR_DEVICE)
    CTR_SURV_GUARD,        // survivor batches a launch of a rolling sequence tried to claim beyond the area's size (surv_take: the
                           // claim is refused, so nothing is lost, and reported: the sizing rule of bf_api.cpp: wf_setup was violated)
    CTR_COUNT
};

}  // namespace bfd
