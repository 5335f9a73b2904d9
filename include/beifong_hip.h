/*
 * beifong_hip.h — flat C ABI of the MI355X radar path-tracing core
 * (libbeifong_hip.so).  This is the drop-in boundary for beifong's
 * transient-radar hot path:
 *
 *   SamplingIntegrator::render / render_sample   src/librender/integrator.cpp:58-310
 *   SamplingIntegrator::receive / receive_sample src/librender/integrator.cpp:315-768,1538-1667
 *   PathIntegrator::sample                       src/integrators/path.cpp:100-226
 *   PathLengthIntegrator / RangeIntegrator       src/integrators/pathlength.cpp:114-337, range.cpp:89-190
 *   PathTimeIntegrator / TimeIntegrator          src/integrators/pathtime.cpp:114-302, time.cpp:86-190
 *   PathTimeFrequencyIntegrator::sample          src/integrators/pathtimefrequency.cpp:103-460
 *   Scene::ray_intersect / ray_test              src/librender/scene.cpp:129-178
 *   ImageBlock::put / SignalBlock::put           src/librender/imageblock.cpp:79+, signalblock.cpp:79-172
 *
 * The reference's plugin ABI is C++ templates over enoki types and cannot be
 * bound binary-for-binary; the host layer (beifong_amd/host) keeps the
 * source-level plugin surface and flattens a loaded scene into the POD
 * description below.  Everything that crosses this boundary is plain C:
 * pointers, sizes, integer status codes.  No exceptions, no torch types.
 *
 * Ownership: the caller owns every input array for the duration of
 * bf_scene_create (they are deep-copied to the device) and owns every output
 * buffer.  A bf_scene handle runs ONE render at a time (it owns the path pool
 * the render's state lives in); concurrent renders of one scene on several
 * streams use one handle each — bf_scene_clone shares the geometry.  This is
 * enforced, not just asked for: a second HOST THREAD entering a call on a
 * handle that is inside one gets BF_ERR_INVALID, and a call whose stream
 * differs from the handle's previous call waits (on the device) for that
 * call's work, so successive renders of one handle are always ordered.
 *
 * Conventions: all arithmetic fp32, indices uint32, RNG state uint64.
 * Matrices are row-major float[16].  Spectra are a single grey lane (all
 * radar scenes use uniform spectra; see DESIGN.md "Spectrum").
 */
#ifndef BEIFONG_HIP_H
#define BEIFONG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BF_ABI_VERSION 4

typedef int bf_status;
enum {
    BF_OK = 0,
    BF_ERR_INVALID = 1,   /* bad argument / inconsistent description        */
    BF_ERR_DEVICE = 2,    /* HIP runtime failure (message in bf_last_error)  */
    BF_ERR_NOMEM = 3,
    BF_ERR_UNSUPPORTED = 4
};

/* ---------------- materials: BSDF plugins flattened ----------------------
 * diffuse.cpp:78-135, roughconductor.cpp:145-392, twosided.cpp:62-180 */
enum { BF_BSDF_DIFFUSE = 0, BF_BSDF_ROUGHCONDUCTOR = 1, BF_BSDF_NULL = 2 };
enum { BF_MF_BECKMANN = 0, BF_MF_GGX = 1 };

typedef struct bf_material {
    uint32_t type;           /* BF_BSDF_*                                    */
    uint32_t twosided;       /* 1: wrapped in <bsdf type="twosided">         */
    float reflectance;       /* diffuse: reflectance; conductor: specular_reflectance */
    float alpha_u, alpha_v;  /* roughconductor roughness                     */
    uint32_t distribution;   /* BF_MF_*                                      */
    uint32_t sample_visible; /* roughconductor sample_visible (default 1)    */
    float eta, k;            /* conductor complex IOR (defaults 0, 1)        */
    uint32_t has_specular_reflectance;
    uint32_t back_material;  /* twosided with TWO nested BSDFs (twosided.cpp:62-92: the second one shades the back side):
                                0 = the same BSDF on both sides; k + 1 = table entry k (itself twosided, back_material 0)
                                is used when the incident direction is below the surface                            */
} bf_material;

/* ---------------- shapes ------------------------------------------------- */
enum { BF_SHAPE_RECTANGLE = 0, BF_SHAPE_MESH = 1 };

typedef struct bf_shape {
    uint32_t type;           /* BF_SHAPE_*                                   */
    uint32_t material;       /* index into materials                         */
    int32_t  emitter;        /* index into emitters if an area emitter / surface
                                transmitter is attached, else -1             */
    uint32_t is_sensor;      /* shape carries the sensor / receiver          */
    /* rectangle (src/shapes/rectangle.cpp): unit square [-1,1]^2, z=0.
       The reference's Transform carries its own inverse (transform.h), so the
       host layer supplies both; the core never inverts a matrix.            */
    float to_world[16];
    float to_object[16];
    /* mesh (include/mitsuba/render/mesh.h:344-348): world-space vertices
       (to_world already applied, as obj.cpp / ply.cpp do at load time)      */
    const float *positions;  /* [3 * n_vertices]                             */
    const float *normals;    /* [3 * n_vertices] or NULL                     */
    const float *texcoords;  /* [2 * n_vertices] or NULL: with them dp_du follows the
                                UV parameterisation (mesh.cpp:493-512) instead of
                                coordinate_system(n), which turns the shading frame */
    const uint32_t *indices; /* [3 * n_faces]                                */
    uint32_t n_vertices;
    uint32_t n_faces;
    /* Shape "velocity" transform (src/librender/shape.cpp:42, default identity) read by
       Shape::doppler (shape.cpp:375-389); only used with BF_FLAG_DOPPLER.  All zeros = identity. */
    float velocity[16];
} bf_shape;

/* ---------------- emitters / transmitters -------------------------------- */
enum {
    BF_EMITTER_SPOT = 0,         /* src/emitters/spot.cpp:64-170             */
    BF_EMITTER_AREA = 1,         /* src/emitters/area.cpp:64-150             */
    BF_TRANSMITTER_AREA = 2,     /* src/transmitters/areatransmitter.cpp     */
    BF_TRANSMITTER_WIGNER = 3,   /* src/transmitters/wignertransmitter.cpp   */
    BF_TRANSMITTER_PHASED = 4,   /* src/transmitters/phasedtransmitter.cpp   */
    BF_EMITTER_POINT = 5         /* src/emitters/point.cpp:60-118: position = to_world
                                    translation, `radiance` = intensity       */
};
enum { BF_SIGNAL_CW = 0, BF_SIGNAL_PULSE = 1, BF_SIGNAL_LINFMCW = 2 };

/* Phased array (phasedtransmitter.cpp:108-165, phasedreceiver.cpp:115-172): the
 * n_elems^2 VIRTUAL elements the constructors precompute, BF_VELEM_FLOATS floats
 * each:  [0..11] m_velem_to_object (3x4 row-major),  [12..23] m_dir_to_local_velem
 * (3x4 row-major, no translation),  [24..26] m_r_dash,  [28..29] m_psi_dash (re, im),
 * the rest 0.  The host layer (plugins/phased*.cpp) and beifong_amd/scenedesc.py
 * build them. */
#define BF_VELEM_FLOATS 32
typedef struct bf_phased_array {
    const float *velems;     /* n_velems * BF_VELEM_FLOATS floats, or NULL   */
    uint32_t n_velems;       /* n_elems * n_elems                            */
    float elem_dims[3];      /* m_wid                                        */
} bf_phased_array;

typedef struct bf_emitter {
    uint32_t type;
    int32_t  shape;          /* area types: index of the carrying shape      */
    float to_world[16];      /* spot                                         */
    float to_object[16];     /* spot: trafo.inverse() (spot.cpp:153)         */
    float radiance;          /* spot: intensity; area: radiance              */
    float cutoff_angle_deg;  /* spot                                         */
    float beam_width_deg;    /* spot                                         */
    /* wigner transmitter signal model (wignertransmitter.cpp:53-146)        */
    uint32_t signal_type;
    float amplitude, freq_centre, freq_ext, pulse_len, prf, gain;
    uint32_t resample_freq;  /* m_resample_freq (:211-221, 430-441): eval / sample_direction overwrite the path's wavelength with
                                MTS_C / f * 1e9, f the signal's instantaneous frequency ("linfmcw", "cw") at the interaction's /
                                the retarded time, signal power 1; "pulse": BF_ERR_UNSUPPORTED (uninitialised in the reference) */
    bf_phased_array array;   /* BF_TRANSMITTER_PHASED                        */
} bf_emitter;

/* ---------------- sensor / receiver + film / adc ------------------------- */
enum {
    BF_SENSOR_FLUXMETER = 0,     /* src/sensors/fluxmeter.cpp:63-105         */
    BF_SENSOR_PERSPECTIVE = 1,   /* src/sensors/perspective.cpp:95-199       */
    BF_RECEIVER_OMNI = 2,        /* src/receivers/omnidirectional.cpp:51-139 */
    BF_RECEIVER_WIGNER = 3,      /* src/receivers/wignerreceiver.cpp:43-299  */
    BF_RECEIVER_PHASED = 4,      /* src/receivers/phasedreceiver.cpp         */
    BF_SENSOR_IRRADIANCEMETER = 5, /* src/sensors/irradiancemeter.cpp:63-105: the flux
                                     meter's rays, weight pi / surface_area      */
    BF_SENSOR_RADIANCEMETER = 6   /* src/sensors/radiancemeter.cpp:49-114: ONE ray from
                                    to_world * origin along to_world * +z, weight 1 */
};

/* Reconstruction filter of the film / ADC (include/mitsuba/core/rfilter.h:10,55-66, src/libcore/rfilter.cpp:9-21).  The
 * host evaluates the filter (src/rfilters/{box,tent,gaussian,mitchell,catmullrom,lanczos}.cpp) into the reference's
 * discretisation; the kernels only look the table up, exactly as ImageBlock::put / SignalBlock::put do
 * (imageblock.cpp:109-165, signalblock.cpp:111-161): eval_discretized(x) = values[min(int(|x * scale|), 31)].
 * radius <= 0.5 + RayEpsilon (e.g. the zero-initialised struct) selects put()'s box branch — one pixel, weight 1 — and
 * nothing else of the struct is read.                                                                                   */
#define BF_FILTER_RESOLUTION 31              /* MTS_FILTER_RESOLUTION */
typedef struct bf_rfilter {
    float radius;                            /* m_radius                                                  */
    float scale;                             /* m_scale_factor = MTS_FILTER_RESOLUTION / m_radius          */
    uint32_t border;                         /* m_border_size = ceil(radius - .5 - 2 RayEpsilon)           */
    uint32_t block_size;                     /* render(): edge of the image blocks a film is rendered in (integrator.cpp:
                                                101-114, MTS_BLOCK_SIZE 32); a sample's position is taken relative to ITS
                                                block's offset - border (imageblock.cpp:113), which matters to the last bit
                                                only.  0: the film / ADC is one block (receive(): integrator.cpp:624-627)  */
    float values[BF_FILTER_RESOLUTION + 1];  /* m_values; values[31] = 0                                   */
} bf_rfilter;

typedef struct bf_sensor {
    uint32_t type;
    int32_t  shape;          /* fluxmeter / receivers: carrying shape        */
    float to_world[16];      /* perspective: camera-to-world                 */
    float sample_to_camera[16]; /* perspective: m_sample_to_camera
                                   (perspective.cpp:104-109, sensor.h:196-231) */
    float fov_x_deg;         /* perspective (already resolved to the x axis) */
    float near_clip, far_clip;
    uint32_t film_width, film_height;     /* the film's CROP size (film.cpp:22-27; = its size without a crop window): what
                                             the sensor samples, the launch names and the histogram holds               */
    float shutter_open, shutter_open_time;
    /* receiver / ADC (receiver.cpp:16-62, adc.cpp:18-46)                    */
    float adc_sampling_start, adc_sampling_time;
    uint32_t t_bins, f_bins;               /* the ADC's FULL size: tf is scaled by size / bandwidth (integrator.cpp:1639)   */
    float t_bandwidth, f_bandwidth;
    float freq_centre, freq_ext, gain;     /* wigner receiver                */
    uint32_t rx_sig_is_delta;  /* wignerreceiver.cpp:258 reads an uninitialised
                                  m_sig_is_delta in raw mode; made explicit  */
    bf_phased_array array;   /* BF_RECEIVER_PHASED                           */
    bf_rfilter rfilter;      /* film->reconstruction_filter() / adc->reconstruction_filter() */
    /* ADC window (adc.cpp:26-38, set_window :80-91): receive() bins into a SignalBlock of window size at the window's offset
     * (integrator.cpp:627-628) and the ADC stores just that (hdradc.cpp:166-167): the histogram of a receive-mode launch is
     * [window_f_bins][window_t_bins][channels] and bf_launch.bins / bins_y name the WINDOW.  All four zero: the whole ADC.      */
    uint32_t window_offset_t, window_offset_f, window_t_bins, window_f_bins;
    /* Film crop window (film.cpp:17-27): the offset of the crop inside the full film.  The position sample of pixel p is
     * (p + crop_offset) + next_2d and the sensor takes (position - crop_offset) / crop_size (integrator.cpp:263,276-278); a
     * perspective camera's sample_to_camera already contains the crop (sensor.h:196-231).  Both zero without a crop window.   */
    uint32_t crop_offset_x, crop_offset_y;
    /* Wigner / phased receiver under receive_type "mix_resample" (BF_FLAG_MIX_RESAMPLE): the receiver samples its frequency from a
     * local-oscillator signal of its own (wignerreceiver.cpp:72-110, sample_frequency :172-189, sample_delta_frequency :149-166):
     * BF_SIGNAL_LINFMCW — the chirp's instantaneous frequency freq_centre + (freq_ext / rx_pulse_len) * (t - rx_pulse_len / 2),
     * t = fmodulo(time, 1 / rx_prf), freq_ext the sweep — or BF_SIGNAL_CW (freq_centre), at the sampled receive time, weight 1
     * (sig_is_delta, the plugins' default for both).  Against a resample_freq transmitter the ADC's frequency axis is then the
     * de-chirped beat.  A signal that is NO delta (rx_sig_is_delta = 0: the plugins' default for "pulse") draws the frequency
     * uniformly from [freq_centre - freq_ext / 2, freq_centre + freq_ext / 2] and weights the ray with eval_signal(time, f)
     * (:118-142: the chirp's / pulse's Wigner function with rx_amplitude, or rx_amplitude^2 for "cw").  "pulse" as a delta reads an
     * uninitialised frequency in the reference: BF_ERR_UNSUPPORTED.  Not read in the "raw" receive types; the omnidirectional
     * receiver has no signal.  */
    uint32_t rx_signal_type;
    float rx_pulse_len, rx_prf, rx_amplitude;
} bf_sensor;

/* ---------------- scene --------------------------------------------------- */
typedef struct bf_physics {
    float c;                 /* MTS_C (spectrum.h:32-35); gen-1 uses 3.0e8   */
    float lambda_min_nm;     /* MTS_WAVELENGTH_MIN (spectrum.h:15-21)        */
    float lambda_max_nm;     /* MTS_WAVELENGTH_MAX (spectrum.h:23-29)        */
} bf_physics;

typedef struct bf_scene_desc {
    const bf_shape *shapes;       uint32_t n_shapes;
    const bf_material *materials; uint32_t n_materials;
    const bf_emitter *emitters;   uint32_t n_emitters;
    bf_sensor sensor;
    bf_physics physics;
} bf_scene_desc;

typedef struct bf_scene bf_scene;   /* opaque, device resident */

/* ---------------- launch --------------------------------------------------- */
enum {
    BF_MODE_PATH = 0,         /* path.cpp; channels X,Y,Z,A,W                 */
    BF_MODE_RANGE = 1,        /* range.cpp o pathlength.cpp; +bins channels   */
    BF_MODE_TIME = 2,         /* time.cpp o pathtime.cpp; +3*bins channels    */
    BF_MODE_RECEIVE_RAW = 3,  /* receive() o pathtimefrequency.cpp; Y,A,W     */
    BF_MODE_RECEIVE_IQ = 4    /* as RECEIVE_RAW, but every contribution is a phasor
                                 c * exp(-j 2 pi L / lambda) of its own optical length L
                                 (receiver -> ... -> transmitter); ADC cells hold I, Q, W.
                                 Coherent pulse sweeps (range-Doppler); the
                                 reference has no counterpart: parity is to the oracle */
};
enum { BF_COLOR_RGB = 0, BF_COLOR_MONO = 1 };

typedef struct bf_launch {
    uint32_t mode;            /* BF_MODE_*                                    */
    uint32_t color_mode;      /* BF_COLOR_*: scalar_rgb -> XYZ via srgb_to_xyz */
    uint64_t n_paths;         /* paths rendered by THIS call                  */
    uint64_t path_offset;     /* first global path index (multi-GPU sharding) */
    uint64_t seed;            /* sampler base seed (sampler.cpp:83-96)        */
    int32_t  max_depth;       /* -1 = infinite (integrator.cpp:1713-1728)     */
    int32_t  rr_depth;        /* default 5                                    */
    uint32_t bins;            /* range/time bins; receive: ADC t_bins         */
    uint32_t bins_y;          /* receive: ADC f_bins (else ignored)           */
    float    bin_width;       /* dr [m] (range) or dt [s] (time)              */
    float    time_c;          /* gen-1 divides by (Float)3.0e8 (pathtime.cpp:140) */
    uint32_t flags;           /* BF_FLAG_*                                    */
    uint32_t phase_bins;      /* receive mode: PhaseIntegrator AOVs S{k}.Y after Y,A,W
                                 (phase.cpp:80-141); 0 = plain pathtimefrequency */
    /* Film of the render modes (integrator.cpp:58-204, Film::crop_size with crop_offset 0):
     * global path g (= path_offset + local index) samples pixel g / spp,
     * pixels in row-major order as in the reference's wavefront branch (integrator.cpp:171-187);
     * its film position is pixel + next_2d and, under the box filter, it lands in pixel ceil(pos - 1)
     * per axis (ImageBlock::put, imageblock.cpp:166-172); a wider reconstruction filter
     * (bf_sensor.rfilter) spreads it over the pixels within its radius (imageblock.cpp:115-165).
     * The histogram becomes
     * [film_height][film_width][channels].  spp == 0 (or a 0 x 0 film) means the 1 x 1 film of
     * the radar scenes: every path samples pixel 0.  path_offset + n_paths must not exceed
     * film_width * film_height * spp.  Receive modes have an ADC instead and need these 0 or 1. */
    uint32_t film_width, film_height;
    uint32_t spp;
} bf_launch;

enum {
    BF_FLAG_STATS = 1u,       /* count BVH nodes visited / triangles tested   */
    BF_FLAG_GLOBAL_ATOMICS = 2u, /* skip LDS privatisation (debug / ablation) */
    BF_FLAG_MEGAKERNEL = 4u,   /* single persistent megakernel instead of the
                                  wavefront pipeline (ablation; same results) */
    BF_FLAG_MIX_RESAMPLE = 16u, /* receive modes: receive_type "mix_resample" (integrator.cpp:1588-1603): the ADC's frequency
                                  coordinate is the BEAT frequency |c / lambda_after - c / lambda_rx| between the wavelength the
                                  path ends with and the one the receiver sampled, instead of c / lambda ("raw" / "raw_resample",
                                  :1604-1623).  The two differ by the Doppler hook (without it the beat is exactly 0 and every
                                  sample falls outside the ADC, ceil(0 - 1) = -1, as at the reference's HEAD) and by resample_freq
                                  transmitters (bf_emitter): the beat of the transmitter's instantaneous frequency with what the
                                  receiver sampled — uniformly from the band (omnidirectional) or from its own local oscillator
                                  (Wigner / phased receiver: bf_sensor.rx_signal_type ..., delta signals only).
                                  receive_type "mixer" (:1624-1634) is an empty branch there and has no counterpart here. */
    BF_FLAG_ROLLING = 32u,     /* bf_render_device only: the render joins the handle's ROLLING SEQUENCE (below, bf_scene_flush) */
    BF_FLAG_TIMING = 64u,      /* rolling sequences: HIP events around every kernel launch; bf_scene_flush's statistics carry
                                  the per-kernel sums (trace_ms / shade_ms / tail_ms).  Set it on every render of the sequence. */
    BF_FLAG_COUNT = 128u,      /* keep the ray / bounce / path counters of a render that returns no bf_stats of its own (a rolling
                                  sequence whose flush will be asked for statistics, the shards of bf_render_sharded_device).
                                  Implied by BF_FLAG_STATS and by a non-NULL stats_out; without any of them the counters stay
                                  zero — adding them up costs every shading launch ~10 same-line atomics per wave.  Set it on
                                  every render of a sequence. */
    BF_FLAG_DOPPLER = 8u       /* receive modes: the Doppler hook the reference carries commented out
                                  ("Took doppler out to test", pathtimefrequency.cpp:124-126,141-144,180-183):
                                  the path's wavelength is shifted by Shape::doppler(si) =
                                  2 dot(si.wi, velocity * to_local(si.p)) / c * lambda (shape.cpp:388) at its first
                                  intersection and at every direct transmitter hit, and the shifted wavelength selects
                                  the ADC's frequency row.  Off by default, as at the reference's HEAD. */
};

/* per-path record for exact parity tests (optional output) */
typedef struct bf_path_record {
    float L;                  /* sensor-weighted grey radiance of the path    */
    float aux;                /* pathlength [m] / pathtime [s] / receive time */
    uint32_t valid;           /* first hit valid (alpha)                      */
    uint32_t n_rays;          /* closest + any-hit queries issued             */
} bf_path_record;

typedef struct bf_stats {
    uint64_t n_paths;
    uint64_t n_rays_closest;
    uint64_t n_rays_shadow;
    uint64_t n_nodes_visited;  /* only with BF_FLAG_STATS                     */
    uint64_t n_tris_tested;    /* only with BF_FLAG_STATS                     */
    uint64_t n_invalid;        /* samples dropped by ImageBlock::put's checks */
    uint64_t n_bounces;        /* path vertices shaded                        */
    float    kernel_ms;        /* HIP-event time of the whole render (all kernels) */
    float    trace_ms;         /* sum of the BVH traversal kernel launches (wf_trace) */
    float    shade_ms;         /* sum of the shading kernel launches (wf_shade)     */
    float    tail_ms;          /* tail kernel                                       */
    uint32_t n_launches_trace; /* wf_trace launches in this render                  */
    uint32_t n_bounce_iters;   /* wavefront iterations executed                     */
    uint64_t n_rays_tail;      /* rays traced by the tail kernel (not by wf_trace)   */
    uint64_t n_rays_traced;    /* rays that entered wf_trace (the others were resolved
                                  by wf_shade: rectangles + BVH root-box test)       */
    uint64_t n_nodes_lds;      /* of n_nodes_visited: served from wf_trace's LDS copy of
                                  the tree's top levels (only with BF_FLAG_STATS)     */
    /* per-kernel breakdown (what bench.py prices each kernel's roofline entry with) */
    uint64_t n_nodes_tail;     /* four-wide node visits of the tail kernel (BF_FLAG_STATS)   */
    uint64_t n_wnodes_tail;    /* sixteen-wide (512-byte) node visits of the tail kernel's row
                                  traversal (BF_FLAG_STATS); included in n_nodes_visited      */
    uint64_t n_tris_tail;      /* triangle tests of the tail kernel (BF_FLAG_STATS)           */
    uint64_t n_bounces_tail;   /* vertices shaded by the tail kernel                          */
    uint64_t n_shade_loads;    /* path-state rows wf_shade read                               */
    uint64_t n_shade_stores;   /* path-state rows wf_shade wrote back                         */
    uint64_t n_shade_shadow;   /* shadow requests wf_shade queued                             */
    uint64_t n_shade_rays;     /* rays generated by wf_shade (rectangle + root-box test each) */
    uint64_t n_guard;          /* rays dropped by wf_trace's iteration guard: always 0, else the
                                  render call fails with BF_ERR_DEVICE                        */
    uint32_t n_launches_tail;  /* tail kernel launches (timed renders / sequences)            */
    uint32_t n_launches_shade; /* wf_shade launches (timed renders / sequences: a rolling call has one more than
                                  bounce iterations, its wake launch)                          */
    uint32_t kernel_variant;   /* which build of the shading / tail kernels ran: BF_VARIANT_LEAN = scene and launch fit the
                                  lean profile (one area-type emitter, perspective camera or omnidirectional receiver, 1 x 1
                                  film, ...: every radar scene of the reference) and everything else is compiled out of the
                                  kernels; BF_VARIANT_WIDE = reconstruction filter wider than a pixel; 0 = general kernels.
                                  Same results either way (BF_LEAN=0 in the environment forces the general ones)            */
    uint32_t reserved_;
} bf_stats;
enum { BF_VARIANT_LEAN = 1, BF_VARIANT_WIDE = 2 };

typedef struct bf_scene_info {
    uint32_t n_shapes, n_rects, n_triangles, n_bvh_nodes;
    uint32_t node_bytes, tri_bytes;
    uint64_t device_bytes;
    float bbox_min[3], bbox_max[3];
    uint32_t bvh_depth;       /* levels of the four-wide tree                  */
    uint32_t bvh_stack_need;  /* worst-case traversal stack entries (<= 31)    */
    uint32_t trace_node_bytes; /* bytes per node as the throughput traversal kernel (wf_trace) reads them: node_bytes (128,
                                  fp32 child boxes) by default, 64 with the opt-in quantised nodes (BF_QUANT_BVH=1) */
    int32_t  device;          /* HIP device the scene lives on (the current device when it was created)              */
} bf_scene_info;

/* ---------------- entry points --------------------------------------------- */
/* ABI handshake.  bf_version() is BF_ABI_VERSION as the LIBRARY was compiled and bf_abi_sizeof(k) the size of struct k
 * there; a caller compares both with its own build before the first call (BF_ABI_MATCHES below; the host layer, every
 * plugin and the ctypes binding do) — a component built against an older header would otherwise have the library
 * write a larger bf_stats / read a larger bf_launch than the caller allocated. */
enum {
    BF_ABI_MATERIAL = 0, BF_ABI_SHAPE, BF_ABI_EMITTER, BF_ABI_SENSOR, BF_ABI_SCENE_DESC, BF_ABI_LAUNCH,
    BF_ABI_PATH_RECORD, BF_ABI_STATS, BF_ABI_SCENE_INFO, BF_ABI_BATCH, BF_ABI_STRUCTS
};
uint32_t bf_abi_sizeof(uint32_t which);          /* 0 for an unknown index */
/* the caller's side of the handshake: a word every component compiled against THIS header agrees on */
#define BF_ABI_FINGERPRINT                                                                                              \
    ((uint64_t) BF_ABI_VERSION << 48 ^ (uint64_t) sizeof(bf_launch) << 36 ^ (uint64_t) sizeof(bf_stats) << 24 ^        \
     (uint64_t) sizeof(bf_scene_desc) << 12 ^ (uint64_t) sizeof(bf_shape) << 6 ^ (uint64_t) sizeof(bf_emitter) ^        \
     (uint64_t) sizeof(bf_scene_info) << 18 ^ (uint64_t) sizeof(bf_batch) << 30)
uint64_t bf_abi_fingerprint(void);               /* BF_ABI_FINGERPRINT as the library was compiled */
int bf_version(void);
const char *bf_last_error(void);                 /* thread-local              */
int bf_device_count(void);
bf_status bf_set_device(int device);

bf_status bf_scene_create(const bf_scene_desc *desc, bf_scene **out);
bf_status bf_scene_destroy(bf_scene *scene);

/* Move / retune the endpoints of an existing scene WITHOUT rebuilding the BVH:
 * `desc` must describe the same layout (shape, rectangle, emitter, material
 * and triangle counts; the mesh arrays are not read) — rectangle transforms,
 * emitters / transmitters, the sensor / receiver + ADC, materials and physics
 * are replaced.  This is one frame of the reference's sweep loops, which
 * rebuild the whole scene per frame just to rotate the radar
 * (python_scripts/animated_trans_rad.py:307-373, Receive.ipynb cell 30).
 * Stream-ordered: renders enqueued on `stream` afterwards see the new
 * endpoints; renders of this scene on other streams must have completed.
 * Fails with BF_ERR_UNSUPPORTED if an endpoint moves further from the origin
 * than the bound the BVH boxes were padded for (recreate the scene then).
 * An open rolling sequence (BF_FLAG_ROLLING, below) on the same stream is NOT
 * finished first: the update joins it — the renders issued so far keep the
 * endpoints they were issued with (every path reads the tables of its own
 * render), the ones issued afterwards see the new ones — so a sweep whose
 * radar turns every frame is one sequence with one tail.  Scenes with phased
 * arrays or a reconstruction filter wider than a pixel, another stream, or
 * the 255th update of a sequence flush it instead (same results). */
bf_status bf_scene_update_endpoints(bf_scene *scene, const bf_scene_desc *desc, void *stream);

/* Rigidly translate ALL mesh triangles of the scene to `offset` (metres, relative
 * to the positions the scene was created with — absolute, so a sweep does not
 * accumulate rounding): vertices become fl(p0 + offset), the four-wide BVH is
 * re-fitted in place (boxes shifted and re-padded), rectangles and endpoints
 * stay.  A moving target between the pulses of a coherent sweep (SURVEY 8f-1;
 * the reference rebuilds the scene per frame, animated_trans_rad.py:307-373).
 * Stream-ordered like bf_scene_update_endpoints. */
bf_status bf_scene_translate_meshes(bf_scene *scene, const float offset[3], void *stream);
bf_status bf_scene_get_info(const bf_scene *scene, bf_scene_info *info);

/* A second handle on the same scene for another stream: the big read-only arrays (BVH, triangles,
 * normals, texture coordinates) are SHARED with `scene` and freed with the last handle; the clone
 * has its own endpoint tables (rectangles, emitters / transmitters, sensor / receiver, materials —
 * bf_scene_update_endpoints on one handle does not touch the others), its own path pool and
 * counters, so renders on different handles may run concurrently on different streams.  The
 * reference shares one Scene object between its worker threads the same way
 * (src/librender/integrator.cpp:125-159: every thread renders blocks of the same scene).
 * bf_scene_translate_meshes on a handle that shares its geometry copies on write.  A clone of a
 * scene that has been translated starts from the geometry that scene renders at that moment. */
bf_status bf_scene_clone(const bf_scene *scene, bf_scene **out);

/* number of floats the given launch accumulates into: 5 (+bins | +3*bins) for
 * the 1x1 film modes, f_bins*t_bins*3 ([y=f][x=t][Y,A,W]) for receive */
uint32_t bf_launch_channels(const bf_launch *launch);

/* Render into a DEVICE buffer hist_dev[film_h*film_w*channels] (accumulates;
 * caller zeroes).  stream is a hipStream_t (NULL = default stream).  The call
 * is asynchronous with respect to the host unless stats_out/records are
 * requested.  This is the entry the multi-GPU driver uses: the histogram stays
 * in HBM for the RCCL reduce. */
bf_status bf_render_device(const bf_scene *scene, const bf_launch *launch,
                           float *hist_dev, bf_path_record *records_dev,
                           void *stream, bf_stats *stats_out);

/* ROLLING SEQUENCES.  Every render ends in a latency-bound tail: a few Russian-roulette survivors of 100+ bounces
 * that a nearly empty GPU finishes one dependent bounce after the other (a quarter of a 2^24-path render's time, three
 * quarters of a 2^20-path one).  Loops that render the SAME scene again and again — the accumulation passes of a long
 * Monte-Carlo render, the pulses of a coherent interval (python_scripts/animated_trans_rad.py:307-384, Receive.ipynb
 * cell 30; the reference's own sample loop is src/librender/integrator.cpp:659-663) — need not pay it per render:
 * with BF_FLAG_ROLLING, bf_render_device enqueues only the throughput part of the render and LEAVES ITS LONG PATHS ALIVE
 * in the handle's pool, where the launches of the handle's next rolling renders carry them along (a path is the same
 * path whichever launch advances it: its own PCG32 stream, sampler.cpp:83-96).  One tail runs per sequence:
 *
 *   bf_scene_flush(scene, stream, stats)   finishes every path still alive (stream-ordered; asynchronous unless
 *                                          `stats` is given).  After it — and a stream synchronisation — every
 *                                          histogram of the sequence is complete.
 *
 * Rules: the renders of a sequence share mode, n_paths (at most the handle's pool, 2^24), bins, depth limits and flags;
 * they may differ in seed, path_offset, hist_dev and records_dev (one histogram / record array PER RENDER, all of which
 * must stay valid until the flush has completed).  stats_out must be NULL.  A render that does not fit the open sequence
 * (or the 256th of a sequence), a plain or batched render, bf_scene_translate_meshes and bf_scene_clone flush the
 * sequence first (bf_scene_update_endpoints joins it where it can: see there), so no path ever sees another scene than
 * the one its render was issued for;
 * bf_scene_destroy abandons it.  Renders without the flag behave exactly as before. */
bf_status bf_scene_flush(bf_scene *scene, void *stream, bf_stats *stats_out);

/* Flush, wait for everything enqueued on the handle and report a device-side failure of ANY render since the last check:
 * a planned render (the second and later ones of a launch shape run without a host round trip) returns BF_OK before its
 * kernels have run, so wf_trace's iteration guard — dropped rays, i.e. a wrong histogram: never expected — can only be
 * reported afterwards: here (BF_ERR_DEVICE; the error refers to an EARLIER render of the handle), by the handle's next
 * render, or by a render with stats_out.  Call it at the end of a sweep. */
bf_status bf_scene_sync(bf_scene *scene);

/* ONE PROCESS, SEVERAL GPUs (SURVEY 8b "device_mask", 8e).  The reference has no multi-device path (TBB over image
 * blocks of one host, src/librender/integrator.cpp:125-159); paths are i.i.d., so GPU g of G renders the global path
 * indices bf_shard_range(launch->n_paths, g, G) of ONE render through bf_launch.path_offset — the union is the sample
 * set of a one-GPU render — and the per-GPU histograms are summed by one ncclAllReduce(float, sum) over xGMI.
 *
 *   scenes[g]   a handle of the scene created on GPU g (bf_set_device(g); bf_scene_create(...)): every entry point of
 *               this header runs on its handle's device, whatever the caller's current device is
 *   hist_dev[g] float[bf_launch_channels(launch)] on that GPU, zeroed by the caller; on completion of streams[g] it
 *               holds the histogram of the WHOLE render (all-reduced), on every GPU
 *   streams[g]  a hipStream_t of that GPU (the array, or an entry, may be NULL: default stream)
 *
 * All GPUs' launches are enqueued by the calling thread before anything is waited for.  launch->n_paths is the TOTAL.
 * With BF_FLAG_ROLLING the shards join their handles' rolling sequences and NO all-reduce is issued: flush every handle,
 * then call bf_allreduce_device on the histograms.  RCCL (librccl.so) is loaded on first use; without it the calls that
 * need a collective fail with BF_ERR_UNSUPPORTED (one GPU needs none).  bf_render_sharded is the host-buffer form
 * (histogram zeroed by the callee; statistics summed over the GPUs, times = the slowest GPU's). */
void bf_shard_range(uint64_t n_paths, uint32_t shard, uint32_t n_shards, uint64_t *offset, uint64_t *count);
bf_status bf_render_sharded_device(bf_scene *const *scenes, uint32_t n_devices, const bf_launch *launch,
                                   float *const *hist_dev, void *const *streams, bf_stats *stats_out);
bf_status bf_render_sharded(bf_scene *const *scenes, uint32_t n_devices, const bf_launch *launch,
                            float *hist_out, bf_stats *stats_out);
/* in-place sum of count floats over the GPUs `devices` (bufs[g] on devices[g]), stream-ordered on streams[g] */
bf_status bf_allreduce_device(const int *devices, uint32_t n_devices, float *const *bufs, uint64_t count,
                              void *const *streams);

/* Convenience: render into a HOST buffer (zeroed by the callee). */
bf_status bf_render(const bf_scene *scene, const bf_launch *launch,
                    float *hist_out, bf_path_record *records_out,
                    bf_stats *stats_out);

/* MANY renders of one scene in ONE launch sequence: the frames of a sweep, the pulses of a
 * coherent processing interval, the shards of a sharded render.  The reference runs such loops
 * one render() / receive() call per frame and rebuilds the scene in between
 * (python_scripts/animated_trans_rad.py:307-384, Receive.ipynb cell 30; the sample loop itself
 * is src/librender/integrator.cpp:659-663); on the GPU every render ends in a latency-bound tail
 * of a few long paths, so K renders issued one by one pay K tails.  Here render k of
 * batch->n_renders is an ordinary render of `launch` (same n_paths, path_offset, mode, bins)
 *   - with sampler seed batch->seeds[k]          (NULL: launch->seed for every render — common
 *                                                 random numbers, what a coherent sweep wants),
 *   - with all meshes at batch->mesh_offsets[3k..] (NULL: as built) — the vertices
 *     bf_scene_translate_meshes(offset) would store, added on the fly while the BVH stays put,
 *   - accumulating into hist + k * bf_launch_channels(launch),
 * and every path of every render is bit-identical to that stand-alone render.  Path records
 * (optional) are [n_renders][n_paths].  Multi-pixel films are not batched.  The arrays of
 * `batch` are host memory and are free again when the call returns. */
typedef struct bf_batch {
    uint32_t n_renders;
    const uint64_t *seeds;        /* [n_renders] or NULL                         */
    const float *mesh_offsets;    /* [n_renders][3] metres, or NULL              */
} bf_batch;
bf_status bf_render_batch_device(const bf_scene *scene, const bf_launch *launch,
                                 const bf_batch *batch, float *hist_dev,
                                 bf_path_record *records_dev, void *stream,
                                 bf_stats *stats_out);
bf_status bf_render_batch(const bf_scene *scene, const bf_launch *launch,
                          const bf_batch *batch, float *hist_out,
                          bf_path_record *records_out, bf_stats *stats_out);

/* Scene::ray_intersect / ray_test over a batch of HOST rays (tests, tools).
 * rays: [n][8] = o.xyz, mint, d.xyz, maxt.  Outputs may be NULL.
 * out_t = +inf on miss; out_prim = global primitive index (shape prefix sum,
 * kdtree.h:2335-2355); out_uv = prim_uv. */
bf_status bf_trace_closest(const bf_scene *scene, uint64_t n, const float *rays,
                           float *out_t, uint32_t *out_prim, uint32_t *out_shape,
                           float *out_uv);
bf_status bf_trace_any(const bf_scene *scene, uint64_t n, const float *rays,
                       uint8_t *out_hit);

/* Scene::ray_intersect returning the whole SurfaceInteraction3f
 * (scene.cpp:129-146 -> PreliminaryIntersection::compute_surface_interaction,
 * interaction.h:613-644; Mesh::compute_surface_interaction mesh.cpp:452-548;
 * Rectangle::compute_surface_interaction rectangle.cpp:265-298).
 * out_si: [n][BF_SI_FLOATS] = t, p.xyz, n.xyz, sh_frame.n.xyz, sh_frame.s.xyz,
 * sh_frame.t.xyz, wi.xyz (local), prim_uv.xy, dp_du.xyz, dp_dv.xyz.
 * A miss has t = +inf and zeros elsewhere.  out_prim / out_shape may be NULL. */
#define BF_SI_FLOATS 27
bf_status bf_ray_intersect(const bf_scene *scene, uint64_t n, const float *rays,
                           float *out_si, uint32_t *out_prim, uint32_t *out_shape);

/* Evaluate the engine's fp32 elementary functions ON THE DEVICE over a HOST
 * array (conformance checks: the kernels use these in place of the libm calls
 * the reference's scalar variants make, e.g. spot.cpp:136, microfacet.h:160-190,
 * wignertransmitter.cpp signal models).  op: 0 sin, 1 cos, 2 acos, 3 exp, 4 log,
 * 5 erf, 6 tan.  Needs no scene. */
bf_status bf_eval_elementary(int op, uint64_t n, const float *x, float *y);

#ifdef __cplusplus
}
#endif
#endif /* BEIFONG_HIP_H */
