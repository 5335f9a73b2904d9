// C ABI of libbeifong_hip.so (include/beifong_hip.h): scene flattening, BVH
// build, device upload and kernel launches.  Plain pointers and sizes in,
// integer status out; no exceptions cross the boundary.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include <dlfcn.h>
// RCCL is dlopen'ed on first use (bf_allreduce_device); only the handful of types and prototypes below are needed, so a ROCm
// install without the RCCL development headers still builds the core (the values are rccl.h's: ncclSuccess 0, ncclFloat32 7,
// ncclSum 0)
#if defined(__has_include) && __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
const char *ncclGetErrorString(ncclResult_t result);
}
#endif

#include <map>
#include <mutex>

#include "bf_bvh.h"
#include "bf_device.h"
#include "bf_wavefront.h"

extern "C" hipError_t bfk_launch_render(const bfd::DScene *sc, const bfd::DLaunch *lp, float *g_hist, bf_path_record *records,
                                        unsigned long long *counters, int stats, unsigned grid, size_t lds_bytes,
                                        hipStream_t stream);
extern "C" float bfk_host_cos(float x);
extern "C" hipError_t bfk_launch_translate(const float4 *tris0, float4 *tris, uint32_t n_tri_rows, const float4 *nodes0,
                                           float4 *nodes, float4 *qnodes, uint32_t n_nodes, const float4 *wnodes0, float4 *wnodes,
                                           uint32_t n_wchildren, const float *d, hipStream_t stream);
extern "C" hipError_t bfk_launch_elementary(int op, uint64_t n, const float *x, float *y);
extern "C" hipError_t bfk_launch_trace(const bfd::DScene *sc, uint64_t n, const float *rays, int any_hit, float *out_t,
                                       uint32_t *out_prim, uint32_t *out_shape, float *out_uv, uint8_t *out_hit,
                                       float *out_si, hipStream_t stream);

extern "C" hipError_t bfk_wf_shade(const bfd::DScene *sc, const bfd::DLaunch *lp, const bfd::WF *wf, uint32_t it, int first,
                                   float *g_hist, bf_path_record *records, unsigned grid, size_t lds_bytes,
                                   hipStream_t stream, int waves);
extern "C" hipError_t bfk_wf_trace(const bfd::DScene *sc, const bfd::WF *wf, uint32_t it, int stats, unsigned grid,
                                   hipStream_t stream, int waves);
extern "C" hipError_t bfk_launch_tail(const bfd::DScene *sc, const bfd::DLaunch *lp, const bfd::WF *wf, uint32_t it,
                                      uint32_t n_slots, float *g_hist, bf_path_record *records, int stats, size_t lds_bytes,
                                      hipStream_t stream, int tail_waves, unsigned spread, unsigned block_cap);
extern "C" hipError_t bfk_roll_set(bfd::DRoll *ring, float4 *offsets, uint32_t idx, const bfd::DRoll *d, const float *offset3, hipStream_t stream);

namespace {

thread_local std::string g_err;

bf_status fail(bf_status st, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return st;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(BF_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <typename T> bf_status upload(const std::vector<T> &v, const T **out, std::vector<void *> &owned, uint64_t &bytes) {
    *out = nullptr;
    if (v.empty()) return BF_OK;
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, v.size() * sizeof(T)));
    owned.push_back(p);
    HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    bytes += v.size() * sizeof(T);
    *out = reinterpret_cast<const T *>(p);
    return BF_OK;
}

void m34(const float *m16, float *out12) { std::memcpy(out12, m16, 12 * sizeof(float)); }

inline float fmaf_(float a, float b, float c) { return std::fmaf(a, b, c); }
struct V3 {
    float x, y, z;
};
// same conventions as the device code (bf_device_math.h)
inline V3 xf_vector(const float *m, V3 v) {
    V3 r = {m[0] * v.x, m[4] * v.x, m[8] * v.x};
    r = {fmaf_(m[1], v.y, r.x), fmaf_(m[5], v.y, r.y), fmaf_(m[9], v.y, r.z)};
    r = {fmaf_(m[2], v.z, r.x), fmaf_(m[6], v.z, r.y), fmaf_(m[10], v.z, r.z)};
    return r;
}
inline float dot(V3 a, V3 b) { return fmaf_(a.z, b.z, fmaf_(a.y, b.y, a.x * b.x)); }
inline V3 cross(V3 a, V3 b) {
    return {fmaf_(a.y, b.z, -(a.z * b.y)), fmaf_(a.z, b.x, -(a.x * b.z)), fmaf_(a.x, b.y, -(a.y * b.x))};
}
inline V3 normalize(V3 a) {
    float s = 1.f / std::sqrt(dot(a, a));
    return {a.x * s, a.y * s, a.z * s};
}

}  // namespace

// The big read-only arrays of a scene (BVH nodes, triangles, normals, texture coordinates): shared by a scene and its
// clones (bf_scene_clone), freed with the last of them.
struct bf_geometry {
    std::vector<void *> owned;
    ~bf_geometry() {
        for (void *p : owned) (void) hipFree(p);
    }
};

// Developer overrides, read from the environment ONCE, when a scene is created; clones inherit them.
struct bf_tunables {
    uint32_t pool = 1u << 24;                // BF_WF_POOL
    int64_t tail = -1;                       // BF_WF_TAIL (-1: by pool size)
    uint32_t trace_refill = bfd::kTraceRefill, trace_stragglers = bfd::kTraceStragglers;
    uint32_t shade_chain = bfd::kShadeChain, row_jobs = bfd::kTailRowJobs;
    int shade_waves = 3, trace_waves = 5, tail_waves = 3;
    unsigned tail_spread = 1, tail_blocks = 0;
    int tail_share = -1;                     // BF_TAIL_SHARE: waves per batch in a stand-alone render's tail (-1: by pool size)
    bool allow_plan = true;                  // BF_WF_SYNC=1 turns launch plans off
    uint32_t roll_iters = 0;                 // BF_ROLL_ITERS: bounce iterations per call of a rolling sequence (0: adaptive)
    uint32_t roll_live = 0;                  // BF_ROLL_LIVE: a rolling call stops iterating once at most this many slots are alive (0: max(1.5 x 2^20, main slots / 4))
    bool no_wide = false, quant = false;
    int wide_rows_log = -1;
    bool lean = true;                        // BF_LEAN=0: never use the kernels' lean variants (bf_device.h: kLean)
    bool tab_cache = true;                   // BF_TAB_CACHE=0: materials / rectangles stay in device memory (no LDS copies)
    bool shade_split = false;                // BF_SHADE_SPLIT=1: wf_shade walks the alive masks twice: slots without a real hit first, real hits second (measured: no net gain)
    uint32_t chain_min = 16;                 // BF_CHAIN_MIN: resolved real hits chain only while at least this many lanes hold one (0: always)
    uint32_t rf_min = 16, rf_th = 44, rf_tm = 24;      // BF_RF_MIN / BF_RF_TH / BF_RF_TM: wf_shade's lane refill and phase vote (bf_wavefront.h)
    uint32_t grid_share = 3;                 // BF_GRID_SHARE: small pools (< grid_small slots) of handles that roll side by side launch 1 / min(peers, this) of the persistent grids (0 / 1: off)
    uint32_t grid_small = 1u << 22;          // BF_GRID_SMALL
    bool roll_join = true;                   // BF_ROLL_JOIN=0: bf_scene_update_endpoints flushes an open rolling sequence (round 3's behaviour)
    uint32_t debug_surv_batches = 0;         // BF_DEBUG_SURV_BATCHES (tests): size of the survivor area in batches, sizing rule off
};
static bf_tunables read_tunables() {
    bf_tunables t;
    auto num = [](const char *name, long long dflt) -> long long {
        const char *e = getenv(name);
        return e ? strtoll(e, nullptr, 10) : dflt;
    };
    t.pool = (uint32_t) std::max<long long>(1024, std::min<long long>(num("BF_WF_POOL", 1ll << 24), 1ll << 26));
    t.tail = num("BF_WF_TAIL", -1);
    t.trace_refill = (uint32_t) num("BF_TRACE_REFILL", bfd::kTraceRefill);
    t.trace_stragglers = (uint32_t) num("BF_TRACE_STRAGGLERS", bfd::kTraceStragglers);
    t.shade_chain = (uint32_t) std::max<long long>(1, num("BF_SHADE_CHAIN", bfd::kShadeChain));
    t.row_jobs = (uint32_t) num("BF_TAIL_ROWJOBS", bfd::kTailRowJobs);
    t.shade_waves = (int) std::max<long long>(1, std::min<long long>(4, num("BF_SHADE_WAVES", 3)));
    {
        const long long w = num("BF_TRACE_WAVES", 5);
        t.trace_waves = w < 5 ? 4 : (w > 5 ? 6 : 5);
    }
    t.tail_waves = num("BF_TAIL_WAVES", 3) == 2 ? 2 : 3;
    t.tail_spread = (unsigned) std::max<long long>(1, num("BF_TAIL_SPREAD", 1));
    t.tail_blocks = (unsigned) std::max<long long>(0, num("BF_TAIL_BLOCKS", 0));
    t.tail_share = (int) num("BF_TAIL_SHARE", -1);
    t.allow_plan = num("BF_WF_SYNC", 0) == 0;
    t.roll_iters = (uint32_t) std::max<long long>(0, std::min<long long>(32, num("BF_ROLL_ITERS", 0)));
    t.roll_live = (uint32_t) std::max<long long>(0, std::min<long long>(num("BF_ROLL_LIVE", 0), 1ll << 30));
    t.no_wide = getenv("BF_NO_WIDE_BVH") != nullptr;
    t.quant = num("BF_QUANT_BVH", 0) != 0;
    t.wide_rows_log = (int) num("BF_WIDE_ROWS_LOG", -1);
    t.lean = num("BF_LEAN", 1) != 0;
    t.tab_cache = num("BF_TAB_CACHE", 1) != 0;
    t.shade_split = num("BF_SHADE_SPLIT", 0) != 0;
    t.chain_min = (uint32_t) std::max<long long>(0, std::min<long long>(num("BF_CHAIN_MIN", 16), 64));
    t.rf_min = (uint32_t) std::max<long long>(1, std::min<long long>(num("BF_RF_MIN", 16), 64));
    t.rf_th = (uint32_t) std::max<long long>(1, std::min<long long>(num("BF_RF_TH", 44), 64));
    t.rf_tm = (uint32_t) std::max<long long>(0, std::min<long long>(num("BF_RF_TM", 24), 64));
    t.grid_share = (uint32_t) std::max<long long>(0, std::min<long long>(num("BF_GRID_SHARE", 3), 16));
    t.grid_small = (uint32_t) std::max<long long>(0, std::min<long long>(num("BF_GRID_SMALL", 1ll << 22), 1ll << 30));
    t.roll_join = num("BF_ROLL_JOIN", 1) != 0;
    t.debug_surv_batches = (uint32_t) std::max<long long>(0, std::min<long long>(num("BF_DEBUG_SURV_BATCHES", 0), 1 << 14));
    return t;
}

struct bf_scene {
    bfd::DScene d;
    bf_tunables tun;
    std::shared_ptr<bf_geometry> geom;     // nodes / wnodes / tris / normals / uvs as created
    // handles that render the SAME triangle / node arrays hold the same token (a clone that took its own snapshot of a
    // translated scene does not): bf_scene_translate_meshes copies on write only while the token is shared
    std::shared_ptr<char> geom_token;
    // handles cloned from one another are meant to be in flight together (one per stream): how many of them have a rolling sequence
    // open right now — small pools then launch a share of the persistent grids each (wf_setup: grid_share)
    std::shared_ptr<std::atomic<int>> peers_rolling;
    bool geom_private = false;             // d.tris / d.nodes / d.wnodes point at this handle's own translated copies
    // one host thread at a time per handle (the handle owns the path pool its render's state lives in)
    mutable std::atomic_flag busy = ATOMIC_FLAG_INIT;
    // stream order between successive renders of the handle: a render on another stream than the previous one waits for it
    mutable hipStream_t last_stream = nullptr;
    mutable hipEvent_t last_done = nullptr;
    mutable bool has_last = false;
    std::vector<void *> owned;             // this handle's own allocations (small tables, spill columns, private geometry)
    bf_scene_info info;
    int device = 0;
    int n_cus = 256;
    std::vector<uint32_t> emitter_types;
    // per-scene scratch for bf_render_device (counters), allocated once
    unsigned long long *counters = nullptr;
    // wavefront workspace, allocated on first use (mutable: lazily grown cache)
    mutable bfd::WF wf;
    mutable std::vector<void *> wf_owned;
    uint32_t n_materials = 0;
    bool any_back_material = false;        // some twosided material has a second nested BSDF (general kernels)
    bool any_resample = false;             // some transmitter re-samples the path's wavelength (resample_freq: general kernels, DLaunch::resample)
    bfd::DSensor sensor_host;              // host copy of the device sensor record
    mutable uint32_t last_variant = 0;     // BF_VARIANT_* of the latest render (bf_stats.kernel_variant)
    uint32_t film_w = 1, film_h = 1;       // the sensor's film (bf_sensor.film_width / film_height)
    uint32_t adc_t = 0, adc_f = 0;         // what a receive-mode launch bins into: the ADC's window, or the whole ADC
    float4 *tris0 = nullptr, *nodes0 = nullptr, *wnodes0 = nullptr;   // pristine geometry, kept once bf_scene_translate_meshes is used
    // device copies of the phased-array element tables: one per emitter (nullptr if none) + the receiver's
    std::vector<bfd::DShape> shapes_host;         // as created: mesh triangles carry their shape's material / emitter index
    std::vector<float *> array_dev;
    std::vector<uint32_t> array_n;
    float *sensor_array_dev = nullptr;
    uint32_t sensor_array_n = 0;
    float origin_scale_built = 0.f;        // ray-origin bound the BVH boxes were padded for (bf_bvh.h)
    mutable uint32_t *wf_host = nullptr;   // pinned read-back of queue counters
    mutable hipEvent_t wf_event = nullptr;
    mutable unsigned long long *wf_masks = nullptr;
    mutable std::vector<hipEvent_t> wf_timing;   // event pool for per-kernel timing (stats only): pair k = events 2k, 2k + 1
    mutable std::vector<int> wf_ev_kind;         // kind of every recorded pair: 0 trace, 1 shade, 2 tail
    mutable float wf_ms[3] = {0, 0, 0};          // trace, shade, tail of the last stats render / rolling sequence
    mutable uint32_t wf_iters = 0, wf_trace_launches = 0, wf_tail_launches = 0, wf_shade_launches = 0;
    // Rolling sequence (bf_render_device with BF_FLAG_ROLLING, bf_scene_flush): see wf_roll_render
    struct Roll {
        bool open = false;
        uint32_t count = 0;                      // renders issued since the sequence was opened
        uint32_t it = 0;                         // bounce-iteration counter (mask parity runs on across calls)
        bf_launch shape;                         // launch of the first render: later ones may differ in seed / path_offset only
        bfd::DLaunch lp;                         // device launch of the sequence (n_paths = supply so far)
        hipStream_t stream = nullptr;
        bool count_nodes = false, timed = false;
        uint32_t per_call = 1;                   // renders every call adds (bf_render_batch_device: the batch size)
        bool offsets = false;                    // the renders carry mesh offsets (batched calls with moving meshes)
        float dmax = 0.f;                        // largest |offset component| so far (box slack of the SHIFT traversal)
        uint32_t window = 1;                     // renders of the LDS histogram window
        uint32_t iters = 0;                      // bounce iterations per call (adapted from the live counts)
        uint32_t flush_iters = 0;                // planned bounce iterations of a flush before its tail (learned)
        uint32_t flush_live = 0;                 // slots alive at the flush's tail (learned: sizes its grid)
        bool multi = false;                      // the endpoints moved between the renders of the sequence (kMulti kernels from then on)
        uint32_t fb_call_iters = 0;              // iterations of the call whose live counts are in flight to wf_feedback
        bool fb_is_flush = false;
    };
    mutable Roll roll;
    // Endpoint-table versions of an open rolling sequence: bf_scene_update_endpoints writes the new tables into the next block of
    // a pool instead of flushing the sequence (the renders issued so far keep reading theirs through the descriptor ring:
    // bf_device.h: DRoll, kMulti); the home buffers (as created) hold the tables whenever no sequence is open.
    struct TabLayout {
        size_t o_rects = 0, o_shapes = 0, o_emit = 0, o_mat = 0, o_sensor = 0, stride = 0;
    };
    mutable TabLayout tab;
    mutable char *tab_pool = nullptr;            // device: kRollRing blocks of tab.stride bytes (allocated on first use)
    mutable uint32_t tab_next = 0;               // next free block
    mutable bool tables_in_pool = false;         // d.rects ... d.sensor point into the pool
    const bfd::DRect *home_rects = nullptr;
    const bfd::DShape *home_shapes = nullptr;
    const bfd::DEmitter *home_emitters = nullptr;
    const bfd::DMaterial *home_materials = nullptr;
    const bfd::DSensor *home_sensor = nullptr;
    mutable bfd::DRoll *roll_ring = nullptr;     // device [kRollRing]
    mutable float4 *roll_offsets = nullptr;      // device [kRollRing]: mesh offset of every render of the sequence
    // Launch plan learned from the previous render of the same shape (wf_render): how many bounce
    // iterations precede the tail and how many slots are then alive.  With a plan the whole render is
    // enqueued without a host round trip; the live counts come back through a pinned buffer afterwards.
    struct WfPlan {
        bool valid = false;
        uint64_t n_paths = 0;
        uint32_t mode = 0, max_depth = 0, n_slots = 0, tail_max = 0;
        uint32_t iters = 0, tail_live = 0;
    };
    mutable WfPlan wf_plan;
    // Pinned staging for small host tables that travel with a launch (batch seeds / mesh offsets, endpoint records):
    // a ring of slots, each with its own device mirror and an event recorded behind the copy, so the caller's arrays
    // and our stack locals are free again when the call returns and nothing blocks unless kStageSlots launches are in
    // flight on this scene.
    static constexpr int kStageSlots = 8;
    struct Stage {
        void *host = nullptr, *dev = nullptr;
        size_t cap = 0;
        hipEvent_t ev = nullptr;
        bool busy = false;
    };
    mutable Stage stage[kStageSlots];
    mutable int stage_next = 0;
    mutable uint32_t *wf_feedback = nullptr;     // pinned: n_live[0 .. wf_fb_iters) of the last planned render
    mutable hipEvent_t wf_fb_event = nullptr;
    mutable bool wf_fb_pending = false;
    mutable uint32_t wf_fb_iters = 0;
};

// One host thread at a time per handle: the second one gets BF_ERR_INVALID instead of a race on the handle's pool.
namespace {
struct BusyGuard {
    const bf_scene *s;
    bool ok;
    int prev_device = -1;
    // ... and every call runs on the handle's own device, whatever the caller's current one is (one host thread may
    // drive the handles of several GPUs: bf_render_sharded_device), restored on return
    explicit BusyGuard(const bf_scene *sc) : s(sc), ok(sc && !sc->busy.test_and_set(std::memory_order_acquire)) {
        if (ok) {
            int cur = -1;
            if (hipGetDevice(&cur) == hipSuccess && cur != sc->device && hipSetDevice(sc->device) == hipSuccess) prev_device = cur;
        }
    }
    ~BusyGuard() {
        if (prev_device >= 0) (void) hipSetDevice(prev_device);
        if (ok) s->busy.clear(std::memory_order_release);
    }
};
// the handle's device for the calls that allocate or launch before (or without) taking the busy flag
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int device) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != device && hipSetDevice(device) == hipSuccess) prev = cur;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void) hipSetDevice(prev);
    }
};
#define BF_ENTER(scene)                                                                                                   \
    BusyGuard busy_guard_(scene);                                                                                         \
    if (!busy_guard_.ok)                                                                                                  \
        return fail(BF_ERR_INVALID, "%s: the scene handle is in use by another host thread (one call at a time per handle; " \
                                    "bf_scene_clone gives every thread / stream its own)", __func__)
}  // namespace

static bf_status order_after_last(const bf_scene *scene, hipStream_t stream);
static bf_status mark_last(const bf_scene *scene, hipStream_t stream);
static bf_status close_sequence(const bf_scene *scene, hipStream_t stream);

extern "C" {

int bf_version(void) { return BF_ABI_VERSION; }
uint32_t bf_abi_sizeof(uint32_t which) {
    static const uint32_t sizes[BF_ABI_STRUCTS] = {sizeof(bf_material), sizeof(bf_shape), sizeof(bf_emitter), sizeof(bf_sensor),
                                                   sizeof(bf_scene_desc), sizeof(bf_launch), sizeof(bf_path_record), sizeof(bf_stats),
                                                   sizeof(bf_scene_info), sizeof(bf_batch)};
    return which < BF_ABI_STRUCTS ? sizes[which] : 0u;
}
uint64_t bf_abi_fingerprint(void) { return BF_ABI_FINGERPRINT; }
const char *bf_last_error(void) { return g_err.c_str(); }

int bf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

bf_status bf_set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return BF_OK;
}

static uint32_t film_pixels(const bf_launch *lp) {
    return (lp->spp && lp->film_width && lp->film_height) ? lp->film_width * lp->film_height : 1u;
}

uint32_t bf_launch_channels(const bf_launch *lp) {
    if (!lp) return 0;
    switch (lp->mode) {
        case BF_MODE_PATH: return 5 * film_pixels(lp);
        case BF_MODE_RANGE: return (5 + lp->bins) * film_pixels(lp);
        case BF_MODE_TIME: return (5 + 3 * lp->bins) * film_pixels(lp);
        case BF_MODE_RECEIVE_RAW: return (3 + lp->phase_bins) * lp->bins * lp->bins_y;
        case BF_MODE_RECEIVE_IQ: return 3 * lp->bins * lp->bins_y;
    }
    return 0;
}

bf_status bf_scene_destroy(bf_scene *s) {
    if (!s) return BF_OK;
    // kernels of this handle that are still in flight read the arrays freed below (an open rolling sequence is simply
    // abandoned: its histograms stay incomplete, as documented)
    if (s->has_last) (void) hipEventSynchronize(s->last_done);
    if (s->roll.open && s->peers_rolling) s->peers_rolling->fetch_sub(1, std::memory_order_relaxed);
    for (void *p : s->owned) (void) hipFree(p);
    for (void *p : s->wf_owned) (void) hipFree(p);
    if (s->wf_host) (void) hipHostFree(s->wf_host);
    if (s->wf_event) (void) hipEventDestroy(s->wf_event);
    if (s->wf_feedback) (void) hipHostFree(s->wf_feedback);
    if (s->wf_fb_event) (void) hipEventDestroy(s->wf_fb_event);
    if (s->last_done) (void) hipEventDestroy(s->last_done);
    for (hipEvent_t e : s->wf_timing) (void) hipEventDestroy(e);
    if (s->counters) (void) hipFree(s->counters);
    if (s->tab_pool) (void) hipFree(s->tab_pool);
    for (auto &st : s->stage) {
        if (st.ev) {
            (void) hipEventSynchronize(st.ev);
            (void) hipEventDestroy(st.ev);
        }
        if (st.host) (void) hipHostFree(st.host);
        if (st.dev) (void) hipFree(st.dev);
    }
    delete s;
    return BF_OK;
}

// Next staging slot with room for `bytes` (see bf_scene::Stage): *host is pinned memory the caller fills, *dev its
// device mirror; stage_commit() enqueues the copy and the slot's event.
static bf_status stage_acquire(const bf_scene *sc, size_t bytes, bf_scene::Stage **out) {
    bf_scene::Stage &st = sc->stage[sc->stage_next];
    sc->stage_next = (sc->stage_next + 1) % bf_scene::kStageSlots;
    if (st.busy) {
        HIP_TRY(hipEventSynchronize(st.ev));
        st.busy = false;
    }
    if (st.cap < bytes) {
        if (st.host) (void) hipHostFree(st.host);
        if (st.dev) (void) hipFree(st.dev);
        st.host = st.dev = nullptr;
        st.cap = 0;
        const size_t cap = std::max<size_t>(4096, (bytes + 4095) & ~size_t(4095));
        HIP_TRY(hipHostMalloc(&st.host, cap));
        HIP_TRY(hipMalloc(&st.dev, cap));
        st.cap = cap;
    }
    if (!st.ev) HIP_TRY(hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
    *out = &st;
    return BF_OK;
}
static bf_status stage_commit(bf_scene::Stage *st, size_t bytes, hipStream_t stream) {
    HIP_TRY(hipMemcpyAsync(st->dev, st->host, bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(st->ev, stream));
    st->busy = true;
    return BF_OK;
}
// the slot only lends its pinned buffer: copies to other device addresses were enqueued by the caller
static bf_status stage_release_after(bf_scene::Stage *st, hipStream_t stream) {
    HIP_TRY(hipEventRecord(st->ev, stream));
    st->busy = true;
    return BF_OK;
}

static int32_t bfd_no_node() { return INT32_MIN; }
static constexpr size_t kTriPad = 4;      // float4 rows of padding behind the triangle array (bf_wavefront.hip: the if-if step of wf_trace)
static_assert(bf::kTopNodes == bfd::kTopNodes, "the builder's breadth-first prefix is what wf_trace caches");
static_assert(bfd::CTR_GUARD + 2 == bfd::CTR_COUNT && bfd::CTR_SURV_GUARD + 1 == bfd::CTR_COUNT,
              "the two sticky guard words are the last counters: renders clear the ones before them");

namespace {
struct TriMeta {
    uint32_t prim, shape;
    const float *n0, *n1, *n2;
    const float *uv0, *uv1, *uv2;
};
// Everything of a scene description except the BVH: shape / rectangle / emitter tables and the sensor record.
struct Flat {
    std::vector<bfd::DShape> shapes;
    std::vector<bfd::DRect> rects;
    std::vector<bf::BuildTri> btris;      // filled only when with_meshes
    std::vector<TriMeta> meta;
    bool any_normals = false, any_uvs = false;
    uint32_t window_t = 0, window_f = 0;      // ADC window size (0: the whole ADC)
    std::vector<bfd::DEmitter> emitters;
    bfd::DSensor sensor;
    uint32_t n_tris = 0;
    float origin_scale = 0.f;             // largest |coordinate| of a rectangle corner, emitter or sensor position
};
}  // namespace

// Phased-array tables arrive as host pointers inside the flattened records: copy them to the device (allocating on
// scene creation, in place — same sizes required — on bf_scene_update_endpoints: beam steering between frames) and
// patch the records with the device addresses.
static bf_status bind_arrays(bf_scene *sc, Flat &f, hipStream_t stream, bool creating) {
    auto put = [&](const float *host, uint32_t n, float *&dev, uint32_t &dev_n) -> bf_status {
        const size_t bytes = (size_t) n * BF_VELEM_FLOATS * sizeof(float);
        if (creating) {
            void *p = nullptr;
            HIP_TRY(hipMalloc(&p, bytes));
            sc->owned.push_back(p);
            dev = (float *) p;
            dev_n = n;
        } else if (!dev || dev_n != n) {
            return fail(BF_ERR_INVALID, "bf_scene_update_endpoints: phased array size changed (%u -> %u virtual elements)", dev_n, n);
        }
        if (creating) {
            HIP_TRY(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
        } else {
            // the caller's table goes through the scene's pinned staging ring: free again when the call returns
            bf_scene::Stage *stg = nullptr;
            bf_status sst = stage_acquire(sc, bytes, &stg);
            if (sst != BF_OK) return sst;
            std::memcpy(stg->host, host, bytes);
            HIP_TRY(hipMemcpyAsync(dev, stg->host, bytes, hipMemcpyHostToDevice, stream));
            if ((sst = stage_release_after(stg, stream)) != BF_OK) return sst;
        }
        return BF_OK;
    };
    if (creating) {
        sc->array_dev.assign(f.emitters.size(), nullptr);
        sc->array_n.assign(f.emitters.size(), 0u);
    }
    for (size_t i = 0; i < f.emitters.size(); ++i) {
        if (f.emitters[i].type != BF_TRANSMITTER_PHASED) continue;
        if (i >= sc->array_dev.size()) return fail(BF_ERR_INVALID, "emitter layout changed");
        bf_status st = put(f.emitters[i].velems, f.emitters[i].n_velems, sc->array_dev[i], sc->array_n[i]);
        if (st != BF_OK) return st;
        f.emitters[i].velems = sc->array_dev[i];
    }
    if (f.sensor.type == BF_RECEIVER_PHASED) {
        bf_status st = put(f.sensor.velems, f.sensor.n_velems, sc->sensor_array_dev, sc->sensor_array_n);
        if (st != BF_OK) return st;
        f.sensor.velems = sc->sensor_array_dev;
    }
    return BF_OK;
}

static bf_status flatten(const bf_scene_desc *desc, Flat &f, bool with_meshes) {
    std::vector<bfd::DShape> &shapes = f.shapes;
    std::vector<bfd::DRect> &rects = f.rects;
    std::vector<bf::BuildTri> &btris = f.btris;
    std::vector<TriMeta> &meta = f.meta;
    bool &any_normals = f.any_normals;
    bool &f_any_uvs = f.any_uvs;
    any_normals = false;
    f.any_uvs = false;
    uint32_t prim = 0;
    uint64_t n_tris_total = 0;
    for (uint32_t i = 0; i < desc->n_shapes; ++i) {
        const bf_shape &s = desc->shapes[i];
        if (s.material >= desc->n_materials) return fail(BF_ERR_INVALID, "shape %u: material index out of range", i);
        if (s.emitter >= (int32_t) desc->n_emitters) return fail(BF_ERR_INVALID, "shape %u: emitter index out of range", i);
        bfd::DShape ds;
        ds.type = s.type;
        ds.material = s.material;
        ds.emitter = s.emitter;
        ds.rect = -1;
        {
            bool any = false;
            for (int k = 0; k < 16; ++k) any = any || s.velocity[k] != 0.f;
            for (int k = 0; k < 12; ++k) ds.velocity[k] = any ? s.velocity[k] : ((k % 5 == 0) ? 1.f : 0.f);    // all zeros = identity
        }
        if (s.type == BF_SHAPE_RECTANGLE) {
            bfd::DRect rc;
            m34(s.to_world, rc.to_world);
            m34(s.to_object, rc.to_object);
            // Rectangle::update — src/shapes/rectangle.cpp:83-92
            V3 dp_du = xf_vector(rc.to_world, V3{2.f, 0.f, 0.f});
            V3 dp_dv = xf_vector(rc.to_world, V3{0.f, 2.f, 0.f});
            V3 n = normalize(V3{s.to_object[8], s.to_object[9], s.to_object[10]});   // inverse-transpose * (0,0,1)
            rc.s[0] = dp_du.x; rc.s[1] = dp_du.y; rc.s[2] = dp_du.z;
            rc.t[0] = dp_dv.x; rc.t[1] = dp_dv.y; rc.t[2] = dp_dv.z;
            rc.n[0] = n.x; rc.n[1] = n.y; rc.n[2] = n.z;
            V3 c = cross(dp_du, dp_dv);
            float area = std::sqrt(dot(c, c));
            if (!(area > 0.f) || !std::isfinite(area)) return fail(BF_ERR_INVALID, "shape %u: degenerate rectangle", i);
            rc.inv_area = 1.f / area;
            rc.area = area;
            rc.shape = i;
            rc.prim = prim;
            rc.material = s.material;
            rc.emitter = s.emitter;
            ds.rect = (int32_t) rects.size();
            rects.push_back(rc);
            prim += 1;
        } else if (s.type == BF_SHAPE_MESH) {
            if (s.n_faces && (!s.positions || !s.indices)) return fail(BF_ERR_INVALID, "shape %u: null mesh arrays", i);
            n_tris_total += s.n_faces;
            for (uint32_t f = 0; with_meshes && f < s.n_faces; ++f) {
                uint32_t i0 = s.indices[3 * f], i1 = s.indices[3 * f + 1], i2 = s.indices[3 * f + 2];
                if (i0 >= s.n_vertices || i1 >= s.n_vertices || i2 >= s.n_vertices)
                    return fail(BF_ERR_INVALID, "shape %u face %u: vertex index out of range", i, f);
                bf::BuildTri t;
                std::memcpy(t.p0, s.positions + 3 * i0, 12);
                std::memcpy(t.p1, s.positions + 3 * i1, 12);
                std::memcpy(t.p2, s.positions + 3 * i2, 12);
                btris.push_back(t);
                TriMeta m{prim + f, i, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
                if (s.texcoords) {
                    m.uv0 = s.texcoords + 2 * i0;
                    m.uv1 = s.texcoords + 2 * i1;
                    m.uv2 = s.texcoords + 2 * i2;
                    f_any_uvs = true;
                }
                if (s.normals) {
                    m.n0 = s.normals + 3 * i0;
                    m.n1 = s.normals + 3 * i1;
                    m.n2 = s.normals + 3 * i2;
                    any_normals = true;
                }
                meta.push_back(m);
            }
            prim += s.n_faces;
        } else {
            return fail(BF_ERR_UNSUPPORTED, "shape %u: unknown type %u", i, s.type);
        }
        shapes.push_back(ds);
    }
    if (n_tris_total >= (1u << 28)) return fail(BF_ERR_UNSUPPORTED, "too many triangles");
    f.n_tris = (uint32_t) n_tris_total;

    std::vector<bfd::DEmitter> &emitters = f.emitters;
    for (uint32_t i = 0; i < desc->n_emitters; ++i) {
        const bf_emitter &e = desc->emitters[i];
        bfd::DEmitter de;
        std::memset(&de, 0, sizeof(de));
        de.type = e.type;
        de.rect = -1;
        de.radiance = e.radiance;
        if (e.type == BF_EMITTER_POINT) {
            m34(e.to_world, de.to_world);
        } else if (e.type == BF_EMITTER_SPOT) {
            m34(e.to_world, de.to_world);
            m34(e.to_object, de.to_object);
            // SpotLight ctor — src/emitters/spot.cpp:83-93
            const float pi = 3.14159265358979323846f;
            de.cutoff = e.cutoff_angle_deg * (pi / 180.f);
            de.beam = e.beam_width_deg * (pi / 180.f);
            de.inv_transition = 1.0f / (de.cutoff - de.beam);
            de.cos_cutoff = bfk_host_cos(de.cutoff);
            de.cos_beam = bfk_host_cos(de.beam);
        } else if (e.type == BF_EMITTER_AREA || e.type == BF_TRANSMITTER_AREA || e.type == BF_TRANSMITTER_WIGNER ||
                   e.type == BF_TRANSMITTER_PHASED) {
            if (e.shape < 0 || e.shape >= (int32_t) desc->n_shapes || desc->shapes[e.shape].type != BF_SHAPE_RECTANGLE)
                return fail(BF_ERR_UNSUPPORTED, "emitter %u: area emitters / transmitters must sit on a rectangle", i);
            de.rect = shapes[e.shape].rect;
            if (e.type == BF_TRANSMITTER_PHASED) {
                if (!e.array.velems || e.array.n_velems == 0) return fail(BF_ERR_INVALID, "emitter %u: phased transmitter without array elements", i);
                de.velems = e.array.velems;          // host pointer for now; replaced by the device copy on upload
                de.n_velems = e.array.n_velems;
                for (int k = 0; k < 3; ++k) de.wid[k] = e.array.elem_dims[k];
            }
            if (e.type == BF_TRANSMITTER_WIGNER || e.type == BF_TRANSMITTER_PHASED) {
                if (e.signal_type > BF_SIGNAL_LINFMCW) return fail(BF_ERR_INVALID, "emitter %u: unknown signal type", i);
                // sample_delta_frequency (wignertransmitter.cpp:152-168) defines the frequency for "linfmcw" and "cw" only
                if (e.resample_freq && e.signal_type == BF_SIGNAL_PULSE)
                    return fail(BF_ERR_UNSUPPORTED, "emitter %u: resample_freq=true with signaltype \"pulse\" reads an uninitialised frequency in the "
                                                    "reference (wignertransmitter.cpp:152-168); use \"linfmcw\" or \"cw\"", i);
                de.resample = e.resample_freq ? 1u : 0u;
                de.signal_type = e.signal_type;
                de.amplitude = e.amplitude;
                de.freq_centre = e.freq_centre;
                de.freq_ext = e.freq_ext;
                de.pulse_len = e.pulse_len;
                de.prf = e.prf;
                de.gain = e.gain;
            }
        } else {
            return fail(BF_ERR_UNSUPPORTED, "emitter %u: type %u not supported by this build", i, e.type);
        }
        emitters.push_back(de);
    }

    bfd::DSensor &sen = f.sensor;
    std::memset(&sen, 0, sizeof(sen));
    sen.type = desc->sensor.type;
    sen.rect = -1;
    if (desc->sensor.type == BF_SENSOR_FLUXMETER || desc->sensor.type == BF_SENSOR_IRRADIANCEMETER || desc->sensor.type == BF_RECEIVER_OMNI ||
        desc->sensor.type == BF_RECEIVER_WIGNER || desc->sensor.type == BF_RECEIVER_PHASED) {
        int32_t sh = desc->sensor.shape;
        if (sh < 0 || sh >= (int32_t) desc->n_shapes || desc->shapes[sh].type != BF_SHAPE_RECTANGLE) {
            return fail(BF_ERR_UNSUPPORTED, "fluxmeter / receiver must sit on a rectangle");
        }
        sen.rect = shapes[sh].rect;
        sen.adc_sampling_start = desc->sensor.adc_sampling_start;
        sen.adc_sampling_time = desc->sensor.adc_sampling_time;
        sen.t_bins = desc->sensor.t_bins;
        sen.f_bins = desc->sensor.f_bins;
        sen.t_bandwidth = desc->sensor.t_bandwidth;
        sen.f_bandwidth = desc->sensor.f_bandwidth;
        sen.freq_centre = desc->sensor.freq_centre;
        sen.freq_ext = desc->sensor.freq_ext;
        sen.gain = desc->sensor.gain;
        sen.rx_sig_is_delta = desc->sensor.rx_sig_is_delta;
        if (desc->sensor.rx_signal_type > BF_SIGNAL_LINFMCW) return fail(BF_ERR_INVALID, "sensor: unknown rx_signal_type %u", desc->sensor.rx_signal_type);
        sen.rx_signal = desc->sensor.rx_signal_type;
        sen.rx_pulse_len = desc->sensor.rx_pulse_len;
        sen.rx_prf = desc->sensor.rx_prf;
        sen.rx_amplitude = desc->sensor.rx_amplitude;
        {
            const bf_sensor &ds = desc->sensor;
            if (ds.window_t_bins || ds.window_f_bins || ds.window_offset_t || ds.window_offset_f) {      // adc.cpp:80-91
                if (ds.window_t_bins == 0 || ds.window_f_bins == 0 || (uint64_t) ds.window_offset_t + ds.window_t_bins > ds.t_bins ||
                    (uint64_t) ds.window_offset_f + ds.window_f_bins > ds.f_bins)
                    return fail(BF_ERR_INVALID, "Invalid window specification! offset (%u, %u) + window size (%u, %u) vs full size (%u, %u)",
                                ds.window_offset_t, ds.window_offset_f, ds.window_t_bins, ds.window_f_bins, ds.t_bins, ds.f_bins);
                sen.win_off_t = ds.window_offset_t;
                sen.win_off_f = ds.window_offset_f;
                f.window_t = ds.window_t_bins;
                f.window_f = ds.window_f_bins;
            }
        }
        if (desc->sensor.type == BF_RECEIVER_PHASED) {
            if (!desc->sensor.array.velems || desc->sensor.array.n_velems == 0)
                return fail(BF_ERR_INVALID, "phased receiver without array elements");
            sen.velems = desc->sensor.array.velems;      // host pointer for now (see bind_arrays)
            sen.n_velems = desc->sensor.array.n_velems;
            for (int k = 0; k < 3; ++k) sen.wid[k] = desc->sensor.array.elem_dims[k];
        }
    } else if (desc->sensor.type == BF_SENSOR_RADIANCEMETER) {
        m34(desc->sensor.to_world, sen.to_world);
    } else if (desc->sensor.type == BF_SENSOR_PERSPECTIVE) {
        m34(desc->sensor.to_world, sen.to_world);
        std::memcpy(sen.sample_to_camera, desc->sensor.sample_to_camera, 16 * sizeof(float));
    } else {
        return fail(BF_ERR_UNSUPPORTED, "sensor type %u not supported by this build", desc->sensor.type);
    }
    {
        // ImageBlock::put / SignalBlock::put take the filtered branch iff radius > 0.5 + RayEpsilon (imageblock.cpp:115)
        const bf_rfilter &rf = desc->sensor.rfilter;
        const float ray_eps = 1500.f * 5.9604644775390625e-8f;          // math::RayEpsilon<float> = Epsilon * 1500
        if (!(rf.radius >= 0.f) || !std::isfinite(rf.radius) || rf.radius > 64.f) return fail(BF_ERR_INVALID, "reconstruction filter radius %g", rf.radius);
        if (rf.radius > .5f + ray_eps) {
            sen.filt_n = (uint32_t) std::ceil((rf.radius - 2.f * ray_eps) * 2.f);
            sen.filt_border = rf.border;
            sen.filt_block = rf.block_size;
            sen.filt_radius = rf.radius;
            sen.filt_scale = rf.scale;
            for (int k = 0; k <= BF_FILTER_RESOLUTION; ++k) sen.filt_tab[k] = rf.values[k];
            if (rf.border > 64u || !(rf.scale > 0.f)) return fail(BF_ERR_INVALID, "reconstruction filter: border %u, scale %g", rf.border, rf.scale);
        }
    }
    sen.crop_x = desc->sensor.crop_offset_x;
    sen.crop_y = desc->sensor.crop_offset_y;
    if (sen.crop_x > (1u << 20) || sen.crop_y > (1u << 20)) return fail(BF_ERR_INVALID, "film crop offset (%u, %u) out of range", sen.crop_x, sen.crop_y);
    sen.near_clip = desc->sensor.near_clip;
    sen.far_clip = desc->sensor.far_clip;
    sen.shutter_open = desc->sensor.shutter_open;
    sen.shutter_open_time = desc->sensor.shutter_open_time;

    // BVH over all mesh triangles; triangles stored in leaf order
    // rays start on scene surfaces, sensors or emitters: bound |origin| for the builder's padding
    float &origin_scale = f.origin_scale;
    origin_scale = 0.f;
    auto grow_scale = [&](const float *m /* 3x4 */, float ex, float ey) {
        for (int r = 0; r < 3; ++r)
            origin_scale = std::max(origin_scale, std::fabs(m[4 * r + 3]) + std::fabs(m[4 * r + 0]) * ex + std::fabs(m[4 * r + 1]) * ey);
    };
    for (const auto &r : rects) grow_scale(r.to_world, 1.f, 1.f);
    for (const auto &e : emitters) grow_scale(e.to_world, 0.f, 0.f);
    grow_scale(sen.to_world, 0.f, 0.f);
    return BF_OK;
}

bf_status bf_scene_create(const bf_scene_desc *desc, bf_scene **out) {
    if (!desc || !out) return fail(BF_ERR_INVALID, "null argument");
    *out = nullptr;
    if (desc->n_shapes && !desc->shapes) return fail(BF_ERR_INVALID, "shapes is null");
    if (desc->n_materials == 0 || !desc->materials) return fail(BF_ERR_INVALID, "at least one material is required");
    for (uint32_t i = 0; i < desc->n_materials; ++i) {
        const uint32_t b = desc->materials[i].back_material;
        if (b == 0) continue;
        if (b > desc->n_materials || !desc->materials[i].twosided || !desc->materials[b - 1].twosided || desc->materials[b - 1].back_material != 0)
            return fail(BF_ERR_INVALID, "material %u: back_material %u must name a twosided table entry without a back side of its own", i, b);
    }
    if (desc->sensor.film_width == 0 || desc->sensor.film_height == 0)
        return fail(BF_ERR_INVALID, "sensor film is %u x %u", desc->sensor.film_width, desc->sensor.film_height);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(BF_ERR_DEVICE, "no HIP device available: the HIP path has no CPU fallback");

    Flat flat;
    {
        bf_status fst = flatten(desc, flat, true);
        if (fst != BF_OK) return fst;
    }
    std::vector<bfd::DShape> &shapes = flat.shapes;
    std::vector<bfd::DRect> &rects = flat.rects;
    std::vector<bf::BuildTri> &btris = flat.btris;
    std::vector<TriMeta> &meta = flat.meta;
    const bool any_normals = flat.any_normals;
    std::vector<bfd::DEmitter> &emitters = flat.emitters;
    const float origin_scale = flat.origin_scale;

    bf_scene *sc = new (std::nothrow) bf_scene();
    if (!sc) return fail(BF_ERR_NOMEM, "out of host memory");
    std::memset(&sc->d, 0, sizeof(sc->d));
    std::memset(&sc->info, 0, sizeof(sc->info));
    std::memset(&sc->wf, 0, sizeof(sc->wf));
    sc->tun = read_tunables();
    sc->geom = std::make_shared<bf_geometry>();
    sc->geom_token = std::make_shared<char>(0);
    sc->peers_rolling = std::make_shared<std::atomic<int>>(0);
    {
        bf_status ast = bind_arrays(sc, flat, nullptr, true);
        if (ast != BF_OK) {
            bf_scene_destroy(sc);
            return ast;
        }
    }
    sc->sensor_host = flat.sensor;
    sc->film_w = desc->sensor.film_width;
    sc->adc_t = flat.window_t ? flat.window_t : flat.sensor.t_bins;
    sc->adc_f = flat.window_f ? flat.window_f : flat.sensor.f_bins;
    sc->film_h = desc->sensor.film_height;
    sc->origin_scale_built = origin_scale;
    bf::BVH bvh;
    bf::build_bvh(btris, bvh, origin_scale);
    bf::BVH4 bvh4;
    bf::collapse_bvh4(bvh, bvh4);
    // sixteen-wide collapse of the same tree for the tail kernel's row traversal (bf_bvh.h: Node16)
    bf::BVH16 bvh16;
    bf::collapse_bvh16(bvh, bvh16);
    // a gang of R rows pops R entries per step: its stack holds at most one block of 16 R children per tree level
    // (blocks are consumed last-in first-out and a block's shallowest entry lies deeper than the block below it)
    uint32_t wide_rlog = 2;
    while (wide_rlog > 0 && (16u << wide_rlog) * std::max(1u, bvh16.max_depth) > (uint32_t) bfd::kWideStack) --wide_rlog;
    const bool use_wide = !btris.empty() && 16u * std::max(1u, bvh16.max_depth) <= (uint32_t) bfd::kWideStack &&
                          btris.size() < (1u << 27) && !sc->tun.no_wide;
    if (sc->tun.wide_rows_log >= 0) wide_rlog = std::min<uint32_t>(wide_rlog, (uint32_t) sc->tun.wide_rows_log);
    for (int k = 0; k < 3 && !btris.empty(); ++k)
        sc->origin_scale_built = std::max({sc->origin_scale_built, std::fabs(bvh.lo[k]), std::fabs(bvh.hi[k])});
    // (+ kTriPad rows behind the last triangle: wf_trace's if-if step reads seven rows from a leaf's first triangle)
    std::vector<float4> tri_data(bfd::kTriStride * btris.size() + (btris.empty() ? 0 : kTriPad), make_float4(0, 0, 0, 0)), nrm_data;
    if (any_normals) nrm_data.resize(3 * btris.size());
    std::vector<float4> uv_data;
    if (flat.any_uvs) uv_data.assign(btris.size(), make_float4(0, 0, 0, 0));
    for (size_t slot = 0; slot < btris.size(); ++slot) {
        uint32_t src = bvh.order[slot];
        const bf::BuildTri &t = btris[src];
        const TriMeta &m = meta[src];
        if (desc->n_materials > 0xfffu || desc->n_emitters > 0xffeu) {
            bf_scene_destroy(sc);
            return fail(BF_ERR_UNSUPPORTED, "more than 4095 materials or 4094 emitters");
        }
        // tag word: bit 0 = has vertex normals, bit 1 = has texture coordinates, bits 8..19 = material,
        // bits 20..31 = emitter + 1 (bf_device_core.h)
        const bf_shape &msh = desc->shapes[m.shape];
        uint32_t has_n = (m.n0 ? 1u : 0u) | (m.uv0 ? 2u : 0u) | ((uint32_t) msh.material << 8) | ((uint32_t) (msh.emitter + 1) << 20);
        if (m.uv0)   // mesh.cpp:494-499: duv0 = uv1 - uv0, duv1 = uv2 - uv0
            uv_data[slot] = make_float4(m.uv1[0] - m.uv0[0], m.uv1[1] - m.uv0[1], m.uv2[0] - m.uv0[0], m.uv2[1] - m.uv0[1]);
        float w0, w1, w2;
        std::memcpy(&w0, &m.prim, 4);
        std::memcpy(&w1, &m.shape, 4);
        std::memcpy(&w2, &has_n, 4);
        tri_data[bfd::kTriStride * slot + 0] = make_float4(t.p0[0], t.p0[1], t.p0[2], w0);
        tri_data[bfd::kTriStride * slot + 1] = make_float4(t.p1[0], t.p1[1], t.p1[2], w1);
        tri_data[bfd::kTriStride * slot + 2] = make_float4(t.p2[0], t.p2[1], t.p2[2], w2);
        if (any_normals) {
            if (m.n0) {
                nrm_data[3 * slot + 0] = make_float4(m.n0[0], m.n0[1], m.n0[2], 0.f);
                nrm_data[3 * slot + 1] = make_float4(m.n1[0], m.n1[1], m.n1[2], 0.f);
                nrm_data[3 * slot + 2] = make_float4(m.n2[0], m.n2[1], m.n2[2], 0.f);
            } else {
                nrm_data[3 * slot + 0] = nrm_data[3 * slot + 1] = nrm_data[3 * slot + 2] = make_float4(0, 0, 0, 0);
            }
        }
    }
    std::vector<float4> node_data(8 * bvh4.nodes.size());
    if (!bvh4.nodes.empty()) std::memcpy(node_data.data(), bvh4.nodes.data(), bvh4.nodes.size() * sizeof(bf::Node4));
    // 64-byte quantised copy of the four-wide nodes for wf_trace (bf_bvh.h: Node4Q).  OPT-IN (BF_QUANT_BVH=1): four loads
    // per node step instead of seven, but +39 VALU operations and 1.3 % more node visits — measured 3 % SLOWER on C2
    // (wf_trace 4.33 -> 4.45 ms per step; the kernel waits on dependent fetches and on its half-busy VALU, not on the
    // number of vector-memory instructions: DESIGN.md 3.1), so the fp32 nodes stay the default.
    std::vector<float4> qnode_data;
    if (!bvh4.nodes.empty() && sc->tun.quant) {
        std::vector<bf::Node4Q> q;
        bf::quantise_bvh4(bvh4, q);
        qnode_data.resize(4 * q.size());
        std::memcpy(qnode_data.data(), q.data(), q.size() * sizeof(bf::Node4Q));
    }
    std::vector<float4> wnode_data;
    if (use_wide) {
        // one spare node of padding: a row's speculative third load of a child record may touch the next 16 bytes
        wnode_data.assign(32 * (bvh16.nodes.size() + 1), make_float4(0, 0, 0, 0));
        if (!bvh16.nodes.empty()) std::memcpy(wnode_data.data(), bvh16.nodes.data(), bvh16.nodes.size() * sizeof(bf::Node16));
    }
    std::vector<bfd::DMaterial> mats(desc->n_materials);      // 48-byte device records (bf_device.h: DMaterial)
    for (uint32_t i = 0; i < desc->n_materials; ++i) {
        mats[i].m = desc->materials[i];
        mats[i].pad = 0u;
    }

    (void) hipGetDevice(&sc->device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, sc->device) == hipSuccess) sc->n_cus = prop.multiProcessorCount;

    uint64_t bytes = 0;
    bf_status st;
    // traversal-stack overflow columns: kernels keep >= 16 entries in LDS and launch at most
    // n_cus * kTraceBlocksPerCU workgroups
    {
        const uint32_t stride = (uint32_t) sc->n_cus * bfd::kTraceBlocksPerCU * bfd::kBlock;
        const uint32_t depth = bvh4.stack_need > 16 ? bvh4.stack_need - 16 : 1;
        void *p = nullptr;
        hipError_t he = hipMalloc(&p, (size_t) stride * depth * sizeof(int));
        if (he != hipSuccess) {
            bf_scene_destroy(sc);
            return fail(BF_ERR_NOMEM, "hipMalloc(traversal spill, %zu bytes): %s", (size_t) stride * depth * sizeof(int), hipGetErrorString(he));
        }
        sc->owned.push_back(p);
        bytes += (uint64_t) stride * depth * sizeof(int);
        sc->d.spill = (int *) p;
        sc->d.spill_stride = stride;
        sc->d.stack_need = bvh4.stack_need;
    }
#define UP(vec, field)                                              \
    if ((st = upload(vec, &sc->d.field, sc->owned, bytes)) != BF_OK) { \
        bf_scene_destroy(sc);                                       \
        return st;                                                  \
    }
#define UPG(vec, field)                                                   \
    if ((st = upload(vec, &sc->d.field, sc->geom->owned, bytes)) != BF_OK) { \
        bf_scene_destroy(sc);                                             \
        return st;                                                        \
    }
    UPG(node_data, nodes);
    UPG(qnode_data, qnodes);
    UPG(wnode_data, wnodes);
    UPG(tri_data, tris);
    UPG(nrm_data, normals);
    UPG(uv_data, uvs);
#undef UPG
    UP(rects, rects);
    UP(shapes, shapes);
    UP(mats, materials);
    UP(emitters, emitters);
    std::vector<bfd::DSensor> sensor_vec(1, flat.sensor);
    UP(sensor_vec, sensor);
#undef UP
    sc->home_rects = sc->d.rects;
    sc->home_shapes = sc->d.shapes;
    sc->home_emitters = sc->d.emitters;
    sc->home_materials = sc->d.materials;
    sc->home_sensor = sc->d.sensor;
    sc->n_materials = desc->n_materials;
    sc->d.n_materials = desc->n_materials;
    sc->d.tab_cache = (sc->tun.tab_cache && desc->n_materials <= bfd::kTabMaxMaterials && rects.size() <= bfd::kTabMaxRects) ? 1u : 0u;
    sc->any_back_material = false;
    for (uint32_t i = 0; i < desc->n_materials; ++i) sc->any_back_material = sc->any_back_material || desc->materials[i].back_material != 0;
    sc->any_resample = false;
    for (const auto &e : emitters) sc->any_resample = sc->any_resample || e.resample != 0u;
    sc->shapes_host = shapes;
    sc->d.n_tris = (uint32_t) btris.size();
    sc->d.n_rects = (uint32_t) rects.size();
    sc->d.n_emitters = (uint32_t) emitters.size();
    for (const auto &e : emitters) sc->emitter_types.push_back(e.type);
    sc->d.n_nodes = (uint32_t) bvh4.nodes.size();
    sc->d.root = bvh4.root_child;
    sc->d.wroot = use_wide ? bvh16.root_child : bfd_no_node();
    sc->d.n_wnodes = use_wide ? (uint32_t) bvh16.nodes.size() : 0u;
    sc->d.wrows_log = wide_rlog;
    sc->d.c = desc->physics.c;
    sc->d.lambda_min = desc->physics.lambda_min_nm;
    sc->d.lambda_max = desc->physics.lambda_max_nm;

    hipError_t e = hipMalloc((void **) &sc->counters, sizeof(unsigned long long) * bfd::CTR_COUNT);
    if (e == hipSuccess) e = hipMemset(sc->counters, 0, sizeof(unsigned long long) * bfd::CTR_COUNT);
    if (e != hipSuccess) {
        bf_scene_destroy(sc);
        return fail(BF_ERR_DEVICE, "hipMalloc(counters): %s", hipGetErrorString(e));
    }

    bf_scene_info &inf = sc->info;
    inf.n_shapes = desc->n_shapes;
    inf.n_rects = sc->d.n_rects;
    inf.n_triangles = sc->d.n_tris;
    inf.n_bvh_nodes = sc->d.n_nodes;
    inf.node_bytes = (uint32_t) sizeof(bf::Node4);
    inf.trace_node_bytes = sc->d.qnodes ? (uint32_t) sizeof(bf::Node4Q) : (uint32_t) sizeof(bf::Node4);
    inf.tri_bytes = 16 * bfd::kTriStride;
    inf.bvh_depth = bvh4.max_depth;
    inf.bvh_stack_need = bvh4.stack_need;
    inf.device_bytes = bytes;
    for (int k = 0; k < 3; ++k) {
        inf.bbox_min[k] = bvh.lo[k];
        inf.bbox_max[k] = bvh.hi[k];
    }
    *out = sc;
    return BF_OK;
}

bf_status bf_scene_update_endpoints(bf_scene *scene, const bf_scene_desc *desc, void *stream_) {
    if (!scene || !desc) return fail(BF_ERR_INVALID, "null argument");
    if (desc->n_shapes && !desc->shapes) return fail(BF_ERR_INVALID, "shapes is null");
    Flat f;
    bf_status st = flatten(desc, f, false);
    if (st != BF_OK) return st;
    if (f.shapes.size() != scene->info.n_shapes || f.rects.size() != scene->d.n_rects || f.emitters.size() != scene->d.n_emitters ||
        f.n_tris != scene->d.n_tris || desc->n_materials != scene->n_materials)
        return fail(BF_ERR_INVALID, "bf_scene_update_endpoints: the description has a different layout than the scene "
                                    "(shapes %zu/%u, rectangles %zu/%u, emitters %zu/%u, triangles %u/%u)",
                    f.shapes.size(), scene->info.n_shapes, f.rects.size(), scene->d.n_rects, f.emitters.size(),
                    scene->d.n_emitters, f.n_tris, scene->d.n_tris);
    for (size_t i = 0; i < f.shapes.size(); ++i)
        if (f.shapes[i].rect < 0 && (f.shapes[i].material != scene->shapes_host[i].material || f.shapes[i].emitter != scene->shapes_host[i].emitter))
            return fail(BF_ERR_UNSUPPORTED, "bf_scene_update_endpoints: mesh shape %zu changed its material / emitter index (the "
                                            "triangle records carry them); create a new scene", i);
    if (scene->d.n_tris && f.origin_scale > scene->origin_scale_built)
        return fail(BF_ERR_UNSUPPORTED, "bf_scene_update_endpoints: an endpoint moved to |coordinate| %g, outside the bound %g the "
                                        "BVH boxes were padded for; create a new scene", (double) f.origin_scale,
                    (double) scene->origin_scale_built);
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BF_ENTER(scene);
    // The paths of an open rolling sequence belong to the endpoints as they are.  Round 3 finished them first (a flush: one
    // tail per frame of a sweep whose radar turns — the loop the reference ships).  Now the update JOINS the sequence: the
    // new tables go into the next block of the handle's pool, the renders issued so far keep reading theirs through the
    // descriptor ring (kMulti kernels).  Phased arrays (their element tables are replaced in place), wide reconstruction
    // filters (no kMulti | kWide kernels), another stream or a full pool fall back to the flush.
    bool phased = f.sensor.type == BF_RECEIVER_PHASED || scene->sensor_array_dev != nullptr;
    for (const auto &e : f.emitters) phased = phased || e.type == BF_TRANSMITTER_PHASED;
    for (float *p : scene->array_dev) phased = phased || p != nullptr;
    bool resample_new = false;
    for (const auto &e : f.emitters) resample_new = resample_new || e.resample != 0u;
    const bool join = scene->roll.open && scene->roll.stream == stream && !phased && scene->sensor_host.filt_n == 0u && f.sensor.filt_n == 0u &&
                      scene->tab_next + 1u < bfd::kRollRing && scene->tun.roll_join && resample_new == scene->any_resample;
    {
        bf_status ost = order_after_last(scene, stream);
        if (ost == BF_OK && !join) ost = close_sequence(scene, stream);
        if (ost != BF_OK) return ost;
    }
    {
        bf_status ast = bind_arrays(scene, f, stream, false);
        if (ast != BF_OK) return ast;
    }
    // small tables: packed into one pinned staging slot owned by the scene (the flattened records above are stack
    // locals and `desc` is the caller's), then copied to their device tables in stream order — no host-blocking copy,
    // nothing read after this call returns
    {
        const size_t b_rects = f.rects.size() * sizeof(bfd::DRect), b_shapes = f.shapes.size() * sizeof(bfd::DShape);
        const size_t b_emit = f.emitters.size() * sizeof(bfd::DEmitter), b_mat = (size_t) desc->n_materials * sizeof(bfd::DMaterial);
        const size_t b_sensor = sizeof(bfd::DSensor);
        auto up16 = [](size_t v) { return (v + 15) & ~size_t(15); };
        const size_t o_rects = 0, o_shapes = o_rects + up16(b_rects), o_emit = o_shapes + up16(b_shapes), o_mat = o_emit + up16(b_emit),
                     o_sensor = o_mat + up16(b_mat), total = o_sensor + up16(b_sensor);
        // where the tables go: the home buffers, or — joining an open sequence — the next block of the pool (same layout as the
        // staging slot)
        char *dst_rects = (char *) scene->home_rects, *dst_shapes = (char *) scene->home_shapes, *dst_emit = (char *) scene->home_emitters,
             *dst_mat = (char *) scene->home_materials, *dst_sensor = (char *) scene->home_sensor;
        if (join) {
            if (!scene->tab_pool) {
                scene->tab.o_rects = o_rects;
                scene->tab.o_shapes = o_shapes;
                scene->tab.o_emit = o_emit;
                scene->tab.o_mat = o_mat;
                scene->tab.o_sensor = o_sensor;
                scene->tab.stride = (total + 255) & ~size_t(255);
                HIP_TRY(hipMalloc((void **) &scene->tab_pool, scene->tab.stride * bfd::kRollRing));
            }
            char *blk = scene->tab_pool + scene->tab.stride * scene->tab_next;
            dst_rects = blk + o_rects;
            dst_shapes = blk + o_shapes;
            dst_emit = blk + o_emit;
            dst_mat = blk + o_mat;
            dst_sensor = blk + o_sensor;
        }
        bf_scene::Stage *stg = nullptr;
        bf_status sst = stage_acquire(scene, total, &stg);
        if (sst != BF_OK) return sst;
        char *h = (char *) stg->host;
        if (b_rects) std::memcpy(h + o_rects, f.rects.data(), b_rects);
        if (b_shapes) std::memcpy(h + o_shapes, f.shapes.data(), b_shapes);
        if (b_emit) std::memcpy(h + o_emit, f.emitters.data(), b_emit);
        for (uint32_t i = 0; i < desc->n_materials; ++i) {
            bfd::DMaterial dm;
            dm.m = desc->materials[i];
            dm.pad = 0u;
            std::memcpy(h + o_mat + (size_t) i * sizeof(dm), &dm, sizeof(dm));
        }
        std::memcpy(h + o_sensor, &f.sensor, b_sensor);
        if (join) {
            HIP_TRY(hipMemcpyAsync(dst_rects, h, total, hipMemcpyHostToDevice, stream));      // one block, the staging slot's layout
        } else {
            if (b_rects) HIP_TRY(hipMemcpyAsync(dst_rects, h + o_rects, b_rects, hipMemcpyHostToDevice, stream));
            if (b_shapes) HIP_TRY(hipMemcpyAsync(dst_shapes, h + o_shapes, b_shapes, hipMemcpyHostToDevice, stream));
            if (b_emit) HIP_TRY(hipMemcpyAsync(dst_emit, h + o_emit, b_emit, hipMemcpyHostToDevice, stream));
            if (b_mat) HIP_TRY(hipMemcpyAsync(dst_mat, h + o_mat, b_mat, hipMemcpyHostToDevice, stream));
            HIP_TRY(hipMemcpyAsync(dst_sensor, h + o_sensor, b_sensor, hipMemcpyHostToDevice, stream));
        }
        sst = stage_release_after(stg, stream);
        if (sst != BF_OK) return sst;
        scene->d.rects = b_rects ? (const bfd::DRect *) dst_rects : scene->d.rects;
        scene->d.shapes = b_shapes ? (const bfd::DShape *) dst_shapes : scene->d.shapes;
        scene->d.emitters = b_emit ? (const bfd::DEmitter *) dst_emit : scene->d.emitters;
        scene->d.materials = b_mat ? (const bfd::DMaterial *) dst_mat : scene->d.materials;
        scene->d.sensor = (const bfd::DSensor *) dst_sensor;
        if (join) {
            ++scene->tab_next;
            scene->tables_in_pool = true;
            scene->roll.multi = true;
        }
    }
    scene->sensor_host = f.sensor;
    scene->any_resample = resample_new;
    scene->film_w = desc->sensor.film_width;
    scene->adc_t = f.window_t ? f.window_t : f.sensor.t_bins;
    scene->adc_f = f.window_f ? f.window_f : f.sensor.f_bins;
    scene->film_h = desc->sensor.film_height;
    scene->emitter_types.clear();
    for (const auto &e : f.emitters) scene->emitter_types.push_back(e.type);
    scene->d.c = desc->physics.c;
    scene->d.lambda_min = desc->physics.lambda_min_nm;
    scene->d.lambda_max = desc->physics.lambda_max_nm;
    return mark_last(scene, stream);
}

bf_status bf_scene_translate_meshes(bf_scene *scene, const float offset[3], void *stream_) {
    if (!scene || !offset) return fail(BF_ERR_INVALID, "null argument");
    if (!(std::isfinite(offset[0]) && std::isfinite(offset[1]) && std::isfinite(offset[2])))
        return fail(BF_ERR_INVALID, "bf_scene_translate_meshes: non-finite offset");
    if (scene->d.n_tris == 0) return BF_OK;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    BF_ENTER(scene);
    {
        bf_status ost = order_after_last(scene, stream);
        if (ost == BF_OK) ost = close_sequence(scene, stream);
        if (ost != BF_OK) return ost;
    }
    const size_t tri_bytes = ((size_t) scene->d.n_tris * bfd::kTriStride + kTriPad) * sizeof(float4), node_bytes = (size_t) scene->d.n_nodes * 8 * sizeof(float4);
    const size_t wnode_bytes = scene->d.wnodes ? (size_t) scene->d.n_wnodes * 32 * sizeof(float4) : 0;
    const bool shared = scene->geom_token.use_count() > 1 && !scene->geom_private;
    // all-or-nothing allocation: the handle's pointers change only once every copy exists (a failed hipMalloc half way
    // must not leave tris0 set and nodes0 null for the next call to trip over)
    struct Copies {
        float4 *p[4] = {nullptr, nullptr, nullptr, nullptr};
    } cp;
    auto alloc_all = [&](const size_t (&bytes)[4]) -> bf_status {
        for (int k = 0; k < 4; ++k) {
            if (!bytes[k]) continue;
            void *q = nullptr;
            hipError_t he = hipMalloc(&q, bytes[k]);
            if (he != hipSuccess) {
                for (int j = 0; j < k; ++j)
                    if (cp.p[j]) (void) hipFree(cp.p[j]);
                return fail(BF_ERR_NOMEM, "bf_scene_translate_meshes: hipMalloc(%zu bytes): %s", bytes[k], hipGetErrorString(he));
            }
            cp.p[k] = (float4 *) q;
        }
        for (int k = 0; k < 4; ++k)
            if (cp.p[k]) scene->owned.push_back(cp.p[k]);
        return BF_OK;
    };
    if (shared) {
        // copy on write: the arrays are shared with clones (bf_scene_clone) — this handle gets its own translated
        // copies; the source of the translation is the geometry as created if this handle has it (it translated in
        // place before it was cloned), else the shared arrays themselves
        const size_t bytes[4] = {tri_bytes, node_bytes, wnode_bytes, scene->d.qnodes ? node_bytes / 2 : 0};
        bf_status cst = alloc_all(bytes);
        if (cst != BF_OK) return cst;
        if (!scene->tris0) {
            scene->tris0 = const_cast<float4 *>(scene->d.tris);
            scene->nodes0 = const_cast<float4 *>(scene->d.nodes);
            scene->wnodes0 = const_cast<float4 *>(scene->d.wnodes);
        }
        scene->d.tris = cp.p[0];
        scene->d.nodes = cp.p[1];
        scene->d.wnodes = cp.p[2];
        if (scene->d.qnodes) scene->d.qnodes = cp.p[3];       // re-quantised from the translated fp32 nodes by the kernel below
        scene->geom_private = true;       // (the token stays shared: tris0 / nodes0 may still READ the shared arrays)
    } else if (!scene->tris0) {
        // first use: keep the geometry as created, so that every later offset is applied to it (no drift)
        const size_t bytes[4] = {tri_bytes, node_bytes, wnode_bytes, 0};
        bf_status cst = alloc_all(bytes);
        if (cst != BF_OK) return cst;
        HIP_TRY(hipMemcpyAsync(cp.p[0], scene->d.tris, tri_bytes, hipMemcpyDeviceToDevice, stream));
        if (node_bytes) HIP_TRY(hipMemcpyAsync(cp.p[1], scene->d.nodes, node_bytes, hipMemcpyDeviceToDevice, stream));
        if (wnode_bytes) HIP_TRY(hipMemcpyAsync(cp.p[2], scene->d.wnodes, wnode_bytes, hipMemcpyDeviceToDevice, stream));
        scene->tris0 = cp.p[0];
        scene->nodes0 = cp.p[1];
        scene->wnodes0 = cp.p[2];
    }
    HIP_TRY(bfk_launch_translate(scene->tris0, const_cast<float4 *>(scene->d.tris), scene->d.n_tris * bfd::kTriStride, scene->nodes0,
                                 const_cast<float4 *>(scene->d.nodes), const_cast<float4 *>(scene->d.qnodes), scene->d.n_nodes, scene->wnodes0,
                                 const_cast<float4 *>(scene->d.wnodes), wnode_bytes ? scene->d.n_wnodes * 16u : 0u, offset, stream));
    return mark_last(scene, stream);
}

bf_status bf_scene_get_info(const bf_scene *scene, bf_scene_info *info) {
    if (!scene || !info) return fail(BF_ERR_INVALID, "null argument");
    *info = scene->info;
    info->device = scene->device;
    return BF_OK;
}

bf_status bf_scene_clone(const bf_scene *src, bf_scene **out) {
    if (!src || !out) return fail(BF_ERR_INVALID, "null argument");
    *out = nullptr;
    BF_ENTER(src);
    {
        bf_status cst = close_sequence(src, src->roll.stream);
        if (cst != BF_OK) return cst;
    }
    HIP_TRY(hipDeviceSynchronize());      // pending endpoint updates / translations of `src` are part of what is cloned
    bf_scene *sc = new (std::nothrow) bf_scene();
    if (!sc) return fail(BF_ERR_NOMEM, "out of host memory");
    sc->d = src->d;                        // geometry pointers shared (as they stand now); the rest replaced below
    sc->tun = src->tun;
    sc->geom = src->geom;
    sc->geom_token = src->geom_token;      // replaced below if the clone takes its own snapshot
    sc->peers_rolling = src->peers_rolling;
    std::memset(&sc->wf, 0, sizeof(sc->wf));
    sc->info = src->info;
    sc->device = src->device;
    sc->n_cus = src->n_cus;
    sc->emitter_types = src->emitter_types;
    sc->n_materials = src->n_materials;
    sc->any_back_material = src->any_back_material;
    sc->any_resample = src->any_resample;
    sc->sensor_host = src->sensor_host;
    sc->film_w = src->film_w;
    sc->adc_t = src->adc_t;
    sc->adc_f = src->adc_f;
    sc->film_h = src->film_h;
    sc->shapes_host = src->shapes_host;
    sc->origin_scale_built = src->origin_scale_built;
    bf_status st = BF_OK;
    auto fail_out = [&](bf_status s) {
        bf_scene_destroy(sc);
        return s;
    };
    // own small tables (endpoints may differ per clone), own spill columns and counters
    auto dup = [&](const void *from, size_t bytes, const void **to) -> bf_status {
        *to = nullptr;
        if (!from || !bytes) return BF_OK;
        void *p = nullptr;
        HIP_TRY(hipMalloc(&p, bytes));
        sc->owned.push_back(p);
        HIP_TRY(hipMemcpy(p, from, bytes, hipMemcpyDeviceToDevice));
        *to = p;
        return BF_OK;
    };
    if (src->tris0 || src->geom_private) {
        // `src` has been translated (in place, or into its own copies): the clone takes a snapshot of the geometry
        // src renders now as ITS geometry "as created"; normals / texture coordinates stay shared
        const size_t tri_bytes = ((size_t) src->d.n_tris * bfd::kTriStride + kTriPad) * sizeof(float4), node_bytes = (size_t) src->d.n_nodes * 8 * sizeof(float4);
        const size_t wnode_bytes = src->d.wnodes ? (size_t) src->d.n_wnodes * 32 * sizeof(float4) : 0;
        if ((st = dup(src->d.tris, tri_bytes, (const void **) &sc->d.tris)) != BF_OK) return fail_out(st);
        if ((st = dup(src->d.nodes, node_bytes, (const void **) &sc->d.nodes)) != BF_OK) return fail_out(st);
        if ((st = dup(src->d.wnodes, wnode_bytes, (const void **) &sc->d.wnodes)) != BF_OK) return fail_out(st);
        if (src->d.qnodes && (st = dup(src->d.qnodes, node_bytes / 2, (const void **) &sc->d.qnodes)) != BF_OK) return fail_out(st);
        sc->geom_private = true;
        sc->geom_token = std::make_shared<char>(0);      // the snapshot is the clone's alone: `src` keeps translating in place
    }
    if ((st = dup(src->d.rects, sizeof(bfd::DRect) * src->d.n_rects, (const void **) &sc->d.rects)) != BF_OK) return fail_out(st);
    if ((st = dup(src->d.shapes, sizeof(bfd::DShape) * src->info.n_shapes, (const void **) &sc->d.shapes)) != BF_OK) return fail_out(st);
    if ((st = dup(src->d.materials, sizeof(bfd::DMaterial) * src->n_materials, (const void **) &sc->d.materials)) != BF_OK) return fail_out(st);
    if ((st = dup(src->d.sensor, sizeof(bfd::DSensor), (const void **) &sc->d.sensor)) != BF_OK) return fail_out(st);
    // emitters carry device pointers to their phased-array tables: duplicate the tables and re-point the records
    std::vector<bfd::DEmitter> em(src->d.n_emitters);
    if (!em.empty()) {
        hipError_t e = hipMemcpy(em.data(), src->d.emitters, sizeof(bfd::DEmitter) * em.size(), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail_out(fail(BF_ERR_DEVICE, "bf_scene_clone: %s", hipGetErrorString(e)));
    }
    sc->array_dev.assign(em.size(), nullptr);
    sc->array_n = src->array_n;
    sc->array_n.resize(em.size(), 0u);
    for (size_t i = 0; i < em.size(); ++i) {
        if (em[i].type != BF_TRANSMITTER_PHASED || i >= src->array_dev.size() || !src->array_dev[i]) continue;
        const void *p = nullptr;
        if ((st = dup(src->array_dev[i], sizeof(float) * BF_VELEM_FLOATS * src->array_n[i], &p)) != BF_OK) return fail_out(st);
        sc->array_dev[i] = (float *) const_cast<void *>(p);
        em[i].velems = sc->array_dev[i];
    }
    if (!em.empty()) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, sizeof(bfd::DEmitter) * em.size());
        if (e == hipSuccess) {
            sc->owned.push_back(p);
            e = hipMemcpy(p, em.data(), sizeof(bfd::DEmitter) * em.size(), hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) return fail_out(fail(BF_ERR_DEVICE, "bf_scene_clone: %s", hipGetErrorString(e)));
        sc->d.emitters = (const bfd::DEmitter *) p;
    }
    if (src->sensor_array_dev) {
        const void *p = nullptr;
        if ((st = dup(src->sensor_array_dev, sizeof(float) * BF_VELEM_FLOATS * src->sensor_array_n, &p)) != BF_OK) return fail_out(st);
        sc->sensor_array_dev = (float *) const_cast<void *>(p);
        sc->sensor_array_n = src->sensor_array_n;
        sc->sensor_host.velems = sc->sensor_array_dev;
        hipError_t e = hipMemcpy((void *) sc->d.sensor, &sc->sensor_host, sizeof(bfd::DSensor), hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail_out(fail(BF_ERR_DEVICE, "bf_scene_clone: %s", hipGetErrorString(e)));
    }
    {
        const uint32_t depth = src->d.stack_need > 16 ? src->d.stack_need - 16 : 1;
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, (size_t) src->d.spill_stride * depth * sizeof(int));
        if (e != hipSuccess) return fail_out(fail(BF_ERR_NOMEM, "bf_scene_clone: traversal spill: %s", hipGetErrorString(e)));
        sc->owned.push_back(p);
        sc->d.spill = (int *) p;
    }
    {
        hipError_t e = hipMalloc((void **) &sc->counters, sizeof(unsigned long long) * bfd::CTR_COUNT);
        if (e == hipSuccess) e = hipMemset(sc->counters, 0, sizeof(unsigned long long) * bfd::CTR_COUNT);
        if (e != hipSuccess) return fail_out(fail(BF_ERR_DEVICE, "bf_scene_clone: counters: %s", hipGetErrorString(e)));
    }
    sc->home_rects = sc->d.rects;
    sc->home_shapes = sc->d.shapes;
    sc->home_emitters = sc->d.emitters;
    sc->home_materials = sc->d.materials;
    sc->home_sensor = sc->d.sensor;
    *out = sc;
    return BF_OK;
}


// ---------------------------------------------------------------------------
// wavefront driver
// ---------------------------------------------------------------------------
static bf_status wf_ensure(const bf_scene *scene, uint32_t capacity) {
    bfd::WF &wf = scene->wf;
    if (wf.capacity >= capacity) return BF_OK;
    for (void *p : scene->wf_owned) (void) hipFree(p);
    scene->wf_owned.clear();
    std::memset(&wf, 0, sizeof(wf));
    scene->roll_ring = nullptr;
    scene->roll_offsets = nullptr;
    auto alloc = [&](void **p, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(p, bytes);
        if (e == hipSuccess) scene->wf_owned.push_back(*p);
        return e;
    };
    size_t n = capacity, nb = capacity / 64;
#if BF_STATE_AOS
#if BF_STATE_AOS == 2
    HIP_TRY(alloc((void **) &wf.recA, n * 128));
    wf.recB = wf.recA + 4;
#else
    HIP_TRY(alloc((void **) &wf.recA, n * 64));
    HIP_TRY(alloc((void **) &wf.recB, n * 64));
#endif
    HIP_TRY(alloc((void **) &wf.recC, n * 64));
#else
    HIP_TRY(alloc((void **) &wf.ray0_, n * 16));
    HIP_TRY(alloc((void **) &wf.ray1_, n * 16));
    HIP_TRY(alloc((void **) &wf.sa_, n * 16));
    HIP_TRY(alloc((void **) &wf.sb_, n * 16));
    HIP_TRY(alloc((void **) &wf.sd_, n * 16));
    HIP_TRY(alloc((void **) &wf.se_, n * 16));
    HIP_TRY(alloc((void **) &wf.hit_, n * 16));
    HIP_TRY(alloc((void **) &wf.hit_prim_, n * 4));
    HIP_TRY(alloc((void **) &wf.sh0_, n * 16));
    HIP_TRY(alloc((void **) &wf.sh1_, n * 16));
    HIP_TRY(alloc((void **) &wf.sh2_, n * 4));
    HIP_TRY(alloc((void **) &wf.sh3_, n * 4));
    HIP_TRY(alloc((void **) &wf.render_, n * 4));
    HIP_TRY(alloc((void **) &wf.dop_, n * 4));
#endif
    HIP_TRY(alloc((void **) &scene->wf_masks, 8 * nb * sizeof(unsigned long long)));
    HIP_TRY(alloc((void **) &wf.n_live, (bfd::kWfMaxIter + 2) * sizeof(uint32_t)));
    HIP_TRY(alloc((void **) &scene->roll_ring, bfd::kRollRing * sizeof(bfd::DRoll)));
    HIP_TRY(alloc((void **) &scene->roll_offsets, bfd::kRollRing * sizeof(float4)));
    HIP_TRY(alloc((void **) &wf.surv_cursor, 64));
    wf.counters = scene->counters;
    wf.capacity = capacity;
    if (!scene->wf_host) HIP_TRY(hipHostMalloc((void **) &scene->wf_host, 64));
    if (!scene->wf_event) HIP_TRY(hipEventCreateWithFlags(&scene->wf_event, hipEventDisableTiming));
    if (!scene->wf_feedback) HIP_TRY(hipHostMalloc((void **) &scene->wf_feedback, (bfd::kWfMaxIter + 2) * sizeof(uint32_t)));
    if (!scene->wf_fb_event) HIP_TRY(hipEventCreateWithFlags(&scene->wf_fb_event, hipEventDisableTiming));
    scene->wf_plan.valid = false;
    scene->wf_fb_pending = false;
    return BF_OK;
}

// Live slots at which the wavefront iterations hand over to the tail kernel.  A bounce iteration of a nearly empty pool
// costs ~0.5 ms of launch and latency floor whatever it holds, the tail ~25 us per bounce: large pools (the pipelined
// bench step, sweeps) switch at 2^17 live slots — more would keep the tail's 168-VGPR waves on the CUs the next renders'
// kernels want (measured: 2^18 costs 12 % of the pipelined C2 rate) — small pools, whose kernels never fill the chip,
// switch as soon as half the pool is done (C3: 1.21 -> 0.98 ms per render, C4 shard: 1.71 -> 1.30).
static uint32_t wf_tail_threshold(const bf_scene *scene, uint32_t n_slots) {
    if (scene->tun.tail >= 0) return (uint32_t) scene->tun.tail;
    if (n_slots >= bfd::kTailSmallPool) return 1u << 17;
    return std::max<uint32_t>(1u << 17, std::min<uint32_t>(1u << 19, n_slots / 2));
}

namespace {
// Everything the launches of one render (or of one call of a rolling sequence) share.
struct WfCtx {
    const bf_scene *scene;
    const bfd::DLaunch *lp;
    float *hist;
    bf_path_record *rec;
    hipStream_t stream;
    bool count_nodes, timed;
    size_t mask_bytes, lds_shade, lds_tail;
    unsigned grid_shade, grid_trace;
    uint32_t tail_max;
};
}  // namespace

// per-kernel timing (stats renders, BF_FLAG_TIMING sequences): one event pair around every launch
static hipError_t wf_tic(const WfCtx &c, int kind) {
    if (!c.timed) return hipSuccess;
    const bf_scene *sc = c.scene;
    while (sc->wf_timing.size() < 2 * (sc->wf_ev_kind.size() + 1)) {
        hipEvent_t e;
        hipError_t he = hipEventCreate(&e);
        if (he != hipSuccess) return he;
        sc->wf_timing.push_back(e);
    }
    sc->wf_ev_kind.push_back(kind);
    return hipEventRecord(sc->wf_timing[2 * (sc->wf_ev_kind.size() - 1)], c.stream);
}
static hipError_t wf_toc(const WfCtx &c) {
    if (!c.timed) return hipSuccess;
    return hipEventRecord(c.scene->wf_timing[2 * (c.scene->wf_ev_kind.size() - 1) + 1], c.stream);
}
// wait for the stream and add the recorded pairs up by kind (wf_ms), then forget them
static bf_status wf_collect_timing(const bf_scene *scene, hipStream_t stream) {
    scene->wf_ms[0] = scene->wf_ms[1] = scene->wf_ms[2] = 0.f;
    scene->wf_tail_launches = scene->wf_shade_launches = 0;
    if (scene->wf_ev_kind.empty()) return BF_OK;
    HIP_TRY(hipStreamSynchronize(stream));
    for (size_t k = 0; k < scene->wf_ev_kind.size(); ++k) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, scene->wf_timing[2 * k], scene->wf_timing[2 * k + 1]));
        scene->wf_ms[scene->wf_ev_kind[k]] += ms;
        if (scene->wf_ev_kind[k] == 2) ++scene->wf_tail_launches;
        if (scene->wf_ev_kind[k] == 1) ++scene->wf_shade_launches;
    }
    scene->wf_ev_kind.clear();
    return BF_OK;
}

// pool size, masks, scheduling knobs and grids for `lp` on this handle
static bf_status wf_setup(const bf_scene *scene, const bfd::DLaunch &lp, uint64_t pool_paths, float *hist_dev, bf_path_record *records_dev,
                          hipStream_t stream, bool count_nodes, bool timed, WfCtx &c, bool rolling = false) {
    uint64_t want = std::min<uint64_t>(scene->tun.pool, std::max<uint64_t>(pool_paths, 64));
    uint32_t n_main = (uint32_t) ((want + 63) & ~uint64_t(63)), n_surv = 0;
    if (rolling) {
        // two renders' worth of main slots (a slot's next path is supplied two calls after its current one) + the
        // survivor area for the paths that are still alive by then (bf_wavefront.h)
        n_main = (uint32_t) ((std::max<uint64_t>(2 * pool_paths, 128) + 63) & ~uint64_t(63));
        // at least one survivor batch per shading wave: a wave claims whole batches (surv_claims_max each per launch) and
        // all the claims of one launch must be distinct batches (wf_shade: surv_take)
        const uint32_t surv_min = (uint32_t) scene->n_cus * 4u * (uint32_t) std::max(2, scene->tun.shade_waves) * 64u;
        n_surv = std::max<uint32_t>(surv_min, std::min<uint32_t>(1u << 21, (n_main / 8 + 63) & ~63u));
        // test hook (BF_DEBUG_SURV_BATCHES): a survivor area far too small for the claims its waves may make, to see the loud
        // check of surv_take fire (tests/test_gpu_rolling.py)
        if (scene->tun.debug_surv_batches) n_surv = scene->tun.debug_surv_batches * 64u;
    }
    bf_status st = wf_ensure(scene, n_main + n_surv);
    if (st != BF_OK) return st;
    bfd::WF &wf = scene->wf;
    wf.n_main = rolling ? n_main : (uint32_t) ((std::min<uint64_t>(wf.capacity, pool_paths) + 63) & ~uint64_t(63));
    wf.n_surv = n_surv;
    wf.n_slots = wf.n_main + wf.n_surv;
    wf.trace_refill = scene->tun.trace_refill;
    wf.trace_stragglers = scene->tun.trace_stragglers;
    wf.shade_chain = scene->tun.shade_chain;
    wf.row_jobs = scene->tun.row_jobs;
    wf.iq = lp.iq;
    wf.has_render = lp.batch != 0u ? 1u : 0u;
    wf.offsets = lp.batch_offsets;
    wf.has_dop = (lp.doppler || lp.resample) ? 1u : 0u;
    wf.box_slack = lp.box_slack;
    const size_t nb = wf.n_slots / 64;
    for (int b = 0; b < 2; ++b) {      // alive | trace | shadow of one parity are contiguous: one memset per bounce
        wf.m_alive[b] = scene->wf_masks + (4 * b + 0) * nb;
        wf.m_trace[b] = scene->wf_masks + (4 * b + 1) * nb;
        wf.m_shadow[b] = scene->wf_masks + (4 * b + 2) * nb;
        wf.m_hit[b] = scene->wf_masks + (4 * b + 3) * nb;
    }
    c.scene = scene;
    c.lp = &lp;
    c.hist = hist_dev;
    c.rec = records_dev;
    c.stream = stream;
    c.count_nodes = count_nodes;
    c.timed = timed;
    c.mask_bytes = 4 * nb * sizeof(unsigned long long);
    wf.hit_split = scene->tun.shade_split ? 1u : 0u;
    wf.chain_min = scene->tun.chain_min;
    wf.rf_min = scene->tun.rf_min;
    wf.rf_th = scene->tun.rf_th;
    wf.rf_tm = scene->tun.rf_tm;
    c.lds_shade = ((sizeof(float) * lp.lds_floats + 15) & ~size_t(15)) + (scene->d.tab_cache ? bfd::kTabBytes : 0u);      // histogram | tables
    c.lds_tail = sizeof(int) * bfd::kStackDepth * bfd::kBlock + c.lds_shade;
    // persistent grids: shade is register-heavy (3 workgroups per CU at 168 VGPRs), trace runs
    // 5 workgroups per CU (28.6 KiB of LDS each: stacks + the tree's top levels; 96 VGPRs)
    const unsigned batches_per_block = bfd::kBlock / 64;
    const unsigned max_blocks = (unsigned) ((nb + batches_per_block - 1) / batches_per_block);
    c.grid_shade = std::max(1u, std::min((unsigned) scene->n_cus * (unsigned) std::max(2, scene->tun.shade_waves), max_blocks));
    c.grid_trace = std::max(1u, std::min((unsigned) scene->n_cus * (unsigned) scene->tun.trace_waves, max_blocks));
    // Small pools never fill the chip: their launches sit at latency floors with nearly idle waves, and a grid sized for the whole GPU
    // keeps the next handle's launch out until it has drained.  Handles that roll side by side (clones of one scene, one per stream)
    // therefore launch a SHARE of the persistent grids each, so that their launches overlap: C3 0.50 -> 0.45 ms per step, a C4 shard
    // 0.62 -> 0.57 with four handles (profiles/r04_grid_share_ab.txt); a handle that rolls alone keeps the full grids (a lone launch
    // is 20 % slower on a third of them).
    if (rolling && scene->tun.grid_share > 1u && wf.n_slots < scene->tun.grid_small) {
        const unsigned peers = (unsigned) std::max(1, scene->peers_rolling->load(std::memory_order_relaxed));
        const unsigned share = std::min(peers, (unsigned) scene->tun.grid_share);
        c.grid_shade = std::max(1u, c.grid_shade / share);
        c.grid_trace = std::max(1u, c.grid_trace / share);
    }
    c.tail_max = wf_tail_threshold(scene, rolling ? wf.n_main / 2 : wf.n_slots);
    wf.surv_claims_max = (wf.n_surv / 64u) / std::max(1u, c.grid_shade * batches_per_block);      // >= 1 by the sizing above
    if (rolling && scene->tun.debug_surv_batches) {                                                   // (the test hook: no rule at all)
        const char *e = getenv("BF_DEBUG_SURV_CLAIMS");
        wf.surv_claims_max = e ? (uint32_t) strtoul(e, nullptr, 10) : 1u << 20;
    }
    return BF_OK;
}
// One bounce iteration `it`: clear the next parity's masks, shade (first: 0 alive masks, 1 first bounce of a pool, 2 alive
// masks then the wake launch of a rolling call), trace.
static bf_status wf_iteration(const WfCtx &c, uint32_t it, int first) {
    const bf_scene *scene = c.scene;
    const bfd::WF &wf = scene->wf;
    const int nxt = (it & 1) ^ 1;
    HIP_TRY(hipMemsetAsync(wf.m_alive[nxt], 0, c.mask_bytes, c.stream));     // alive, trace, shadow, hit are contiguous
    if (first != 1) {
        // first launch of a rolling call (first == 2): the evicting variant — long paths make room for the new render's
        HIP_TRY(wf_tic(c, 1));
        HIP_TRY(bfk_wf_shade(&scene->d, c.lp, &wf, it, first == 2 ? 3 : 0, c.hist, c.rec, c.grid_shade, c.lds_shade, c.stream,
                             scene->tun.shade_waves));
        HIP_TRY(wf_toc(c));
    }
    if (first != 0) {
        HIP_TRY(wf_tic(c, 1));
        HIP_TRY(bfk_wf_shade(&scene->d, c.lp, &wf, it, first, c.hist, c.rec, c.grid_shade, c.lds_shade, c.stream, scene->tun.shade_waves));
        HIP_TRY(wf_toc(c));
    }
    return BF_OK;
}
static bf_status wf_trace_launch(const WfCtx &c, uint32_t it) {
    HIP_TRY(wf_tic(c, 0));
    HIP_TRY(bfk_wf_trace(&c.scene->d, &c.scene->wf, it, c.count_nodes ? 1 : 0, c.grid_trace, c.stream, c.scene->tun.trace_waves));
    HIP_TRY(wf_toc(c));
    return BF_OK;
}
static bf_status wf_tail_launch(const WfCtx &c, uint32_t it, uint32_t est_live, bool alone = false) {
    const bf_scene *scene = c.scene;
    // `alone`: nothing else wants the CUs (the flush of a rolling sequence): spread the paths thinly — up to four waves
    // share a batch, so a wave starts with <= 16 paths and walks four lanes per ray from its first bounce instead of
    // waiting for the longest of 64 lane-per-ray walks (half of a tail's cycles are those dense iterations)
    uint32_t share = 1;
    if (!alone && scene->tun.tail_share > 1) {
        share = scene->tun.tail_share >= 4 ? 4u : 2u;
        est_live = (uint32_t) std::min<uint64_t>((uint64_t) est_live * share, 1u << 30);
    }
    if (alone) {
        const uint32_t resident = (uint32_t) scene->n_cus * 4u * 3u;          // waves at 3 per SIMD
        while (share < 4u && (uint64_t) est_live * (share * 2u) / 64u <= resident) share *= 2u;
        est_live = (uint32_t) std::min<uint64_t>((uint64_t) est_live * share, 1u << 30);
    }
    scene->wf.tail_share = share;
    HIP_TRY(wf_tic(c, 2));
    HIP_TRY(bfk_launch_tail(&scene->d, c.lp, &scene->wf, it, est_live, c.hist, c.rec, c.count_nodes ? 1 : 0, c.lds_tail, c.stream,
                            scene->tun.tail_waves, scene->tun.tail_spread, scene->tun.tail_blocks));
    HIP_TRY(wf_toc(c));
    return BF_OK;
}
// The guard word of wf_trace (CTR_GUARD) is never cleared by a render, so it is sticky across the renders of a handle.
// So is the survivor-area word of rolling sequences (CTR_SURV_GUARD: wf_shade: surv_take refused a claim).
static bf_status wf_guard_error(unsigned long long lost, unsigned long long refused = 0) {
    if (lost)
        return fail(BF_ERR_DEVICE, "wf_trace's iteration guard dropped %llu rays in an earlier render of this scene: that render's "
                                   "histogram is wrong (a traversal bug; please report the scene)", lost);
    return fail(BF_ERR_DEVICE, "a launch of a rolling sequence of this scene tried to claim %llu survivor batches beyond the size of "
                               "its survivor area (the claims were refused: no path was lost, but the sizing rule of wf_setup is "
                               "violated; please report the scene and the launch)", refused);
}
// every path of a completed render / sequence was binned exactly once (CTR_FILM counts film_put calls): the loud form of the
// tests' "sum of the weight channel + invalid samples == paths"
static bf_status check_film_count(const unsigned long long *c, uint64_t n_paths) {
#ifdef BF_NO_FILM_CTR      // developer A/B build without the counter
    return BF_OK;
#endif
    if (c[bfd::CTR_FILM] != n_paths)
        return fail(BF_ERR_DEVICE, "%llu of %llu paths were binned: paths were lost or binned twice (a scheduling bug; please report the "
                                   "scene and the launch)", (unsigned long long) c[bfd::CTR_FILM], (unsigned long long) n_paths);
    return BF_OK;
}
// read + clear the sticky words of a handle whose work has completed; *lost / *refused as found
static bf_status read_guards(const bf_scene *scene, unsigned long long *lost, unsigned long long *refused) {
    unsigned long long g[2] = {0, 0};
    HIP_TRY(hipMemcpy(g, scene->counters + bfd::CTR_GUARD, sizeof(g), hipMemcpyDeviceToHost));
    if (g[0] || g[1]) {
        HIP_TRY(hipMemset(scene->counters + bfd::CTR_GUARD, 0, sizeof(g)));
        scene->wf_fb_pending = false;       // (a copy of the words may be in flight to the next planned render's feedback)
    }
    *lost = g[0];
    *refused = g[1];
    return BF_OK;
}

// Host control loop.  Per bounce `it`: [zero the next masks] -> wf_shade(it) ->
// [copy the live-slot counter] -> wf_trace(it).  The host waits for the counter of
// bounce `it` while wf_trace(it) is still running, so the device never idles on
// the decision; once at most wf_tail_threshold() slots are live, one tail launch
// finishes them (including the paths those slots still have to start).
static bf_status wf_render(const bf_scene *scene, const bfd::DLaunch &lp, float *hist_dev, bf_path_record *records_dev,
                           hipStream_t stream, bool count_nodes, bool stats) {
    WfCtx c;
    bf_status st = wf_setup(scene, lp, lp.n_paths, hist_dev, records_dev, stream, count_nodes, stats, c);
    if (st != BF_OK) return st;
    bfd::WF &wf = scene->wf;
    HIP_TRY(hipMemsetAsync(wf.n_live, 0, (bfd::kWfMaxIter + 2) * sizeof(uint32_t), stream));
    const uint32_t tail_max = c.tail_max;
    volatile uint32_t *hq = scene->wf_host;      // [0] = n_live[it]
    scene->wf_ev_kind.clear();
    auto finish = [&](uint32_t iters, uint32_t traces) -> bf_status {
        scene->wf_iters = iters;
        scene->wf_trace_launches = traces;
        if (!stats) {
            scene->wf_ms[0] = scene->wf_ms[1] = scene->wf_ms[2] = 0.f;
            return BF_OK;
        }
        return wf_collect_timing(scene, stream);
    };
    // ---- planned render: no host round trip ------------------------------------------------------
    bf_scene::WfPlan &plan = scene->wf_plan;
    const uint32_t depth_key = (uint32_t) lp.max_depth;
    const bool fb_ready = scene->wf_fb_pending && hipEventQuery(scene->wf_fb_event) == hipSuccess;
    (void) hipGetLastError();      // hipErrorNotReady from the query must not leak into the launch checks below
    if (fb_ready && scene->roll.fb_call_iters) {      // the counts in flight belong to a rolling call: not this plan's
        scene->wf_fb_pending = false;
        scene->roll.fb_call_iters = 0;
    } else if (fb_ready) {
        // live counts of the last planned render: move the switch to the tail to where it belongs
        scene->wf_fb_pending = false;
        const unsigned long long lost = reinterpret_cast<volatile unsigned long long *>(scene->wf_host)[1];
        const unsigned long long refused = reinterpret_cast<volatile unsigned long long *>(scene->wf_host)[2];
        if (lost || refused) {
            HIP_TRY(hipMemsetAsync(scene->counters + bfd::CTR_GUARD, 0, 2 * sizeof(unsigned long long), stream));
            return wf_guard_error(lost, refused);
        }
        const uint32_t *nl = scene->wf_feedback;
        uint32_t k = 0;
        while (k < scene->wf_fb_iters && nl[k] > plan.tail_max) ++k;
        if (k < scene->wf_fb_iters) {
            plan.iters = k + 1;
            plan.tail_live = nl[k];
        } else {                       // still above the threshold after the planned iterations: extend
            plan.iters = std::min<uint32_t>(scene->wf_fb_iters + 2, bfd::kWfMaxIter - 1);
            plan.tail_live = nl[scene->wf_fb_iters - 1];
        }
    }
    if (scene->tun.allow_plan && plan.valid && plan.n_paths == lp.n_paths && plan.mode == lp.mode &&
        plan.max_depth == depth_key && plan.n_slots == wf.n_slots && plan.tail_max == tail_max && plan.iters > 0) {
        for (uint32_t it = 0; it < plan.iters; ++it) {
            if ((st = wf_iteration(c, it, it == 0 ? 1 : 0)) != BF_OK) return st;
            if ((st = wf_trace_launch(c, it)) != BF_OK) return st;
        }
        // the tail kernel finishes whatever is alive, whatever the estimate: the estimate only sizes its grid
        const uint32_t est = std::max<uint32_t>(plan.tail_live + plan.tail_live / 4, 64u * bfd::kBlock);
        if ((st = wf_tail_launch(c, plan.iters, est)) != BF_OK) return st;
        if (!scene->wf_fb_pending) {
            HIP_TRY(hipMemcpyAsync(scene->wf_feedback, wf.n_live, plan.iters * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(reinterpret_cast<unsigned long long *>(scene->wf_host) + 1, wf.counters + bfd::CTR_GUARD,
                                   2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipEventRecord(scene->wf_fb_event, stream));
            scene->wf_fb_pending = true;
            scene->wf_fb_iters = plan.iters;
            scene->roll.fb_call_iters = 0;
        }
        return finish(plan.iters, plan.iters);
    }
    auto learn = [&](uint32_t iters, uint32_t live) {
        plan.valid = true;
        plan.n_paths = lp.n_paths;
        plan.mode = lp.mode;
        plan.max_depth = depth_key;
        plan.n_slots = wf.n_slots;
        plan.tail_max = tail_max;
        plan.iters = iters;
        plan.tail_live = live;
        scene->wf_fb_pending = false;
    };

    // ---- synchronous render (first render of a shape, or per-kernel statistics requested) ----------
    for (uint32_t it = 0; it < bfd::kWfMaxIter; ++it) {
        if ((st = wf_iteration(c, it, it == 0 ? 1 : 0)) != BF_OK) return st;
        HIP_TRY(hipMemcpyAsync((void *) &hq[0], wf.n_live + it, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipEventRecord(scene->wf_event, stream));
        if ((st = wf_trace_launch(c, it)) != BF_OK) return st;
        HIP_TRY(hipEventSynchronize(scene->wf_event));
        uint32_t n_live = hq[0];
        if (n_live == 0) {
            learn(it + 1, 0);
            return finish(it + 1, it + 1);
        }
        if (n_live <= tail_max) {
            learn(it + 1, n_live);
            if ((st = wf_tail_launch(c, it + 1, n_live)) != BF_OK) return st;
            return finish(it + 1, it + 1);
        }
    }
    return fail(BF_ERR_UNSUPPORTED, "path depth exceeded the wavefront iteration limit (%u bounces)", bfd::kWfMaxIter);
}

// ---------------------------------------------------------------------------
// Rolling sequences (BF_FLAG_ROLLING).  Every render ends in a latency-bound tail of a few long Russian-roulette
// survivors (126-174 bounces among 2^20 paths) that costs a quarter of a 2^24-path step and three quarters of a 2^20-path
// one.  Consecutive renders of one handle that only differ in seed / shard offset / output buffer — the steps of a
// Monte-Carlo accumulation, the pulses of a coherent interval (python_scripts/animated_trans_rad.py:307-384,
// Receive.ipynb cell 30 run such loops one render() / receive() per frame) — therefore form ONE batched launch whose
// path supply grows by one render per call: slot i renders global paths i, i + n_slots, ... (render = g / n_paths), a
// call enqueues a few bounce iterations over the whole pool (first the slots still alive, then a "wake" launch that
// starts the next path of every idle slot), and the survivors of render k simply ride along with the launches of
// renders k + 1, k + 2, ... .  One tail runs per sequence: bf_scene_flush (or anything that needs the pool: another kind
// of render, an endpoint update, a clone).  Every path is the path a stand-alone render would trace (own PCG32 stream).
// ---------------------------------------------------------------------------
static bool roll_same_shape(const bf_launch &a, const bf_launch &b) {
    return a.mode == b.mode && a.color_mode == b.color_mode && a.n_paths == b.n_paths && a.max_depth == b.max_depth &&
           a.rr_depth == b.rr_depth && a.bins == b.bins && a.bins_y == b.bins_y && a.bin_width == b.bin_width &&
           a.time_c == b.time_c && a.flags == b.flags && a.phase_bins == b.phase_bins;
}

static bf_status wf_roll_flush(const bf_scene *scene, hipStream_t stream, bool sync_timing);

static bf_status wf_roll_render(const bf_scene *scene, const bf_launch *launch, const bf_batch *batch, const bfd::DLaunch &lp_in,
                                float *hist_dev, bf_path_record *records_dev, hipStream_t stream) {
    bf_scene::Roll &r = scene->roll;
    bf_status st;
    const uint32_t K = batch ? batch->n_renders : 1u;          // renders this call adds to the sequence
    const bool with_offsets = batch && batch->mesh_offsets && scene->d.n_tris != 0;
    if (r.open && (!roll_same_shape(r.shape, *launch) || r.per_call != K || r.offsets != with_offsets ||
                   r.count + K > bfd::kRollRing || r.stream != stream)) {
        if ((st = wf_roll_flush(scene, r.stream, false)) != BF_OK) return st;
        if (r.stream != stream) {
            // the flush ran on the old stream: the new sequence's first launches reuse the pool behind it
            HIP_TRY(hipEventRecord(scene->wf_event, r.stream));
            HIP_TRY(hipStreamWaitEvent(stream, scene->wf_event, 0));
        }
    }
    const bool opening = !r.open;
    if (opening) {
        // what the previous sequence of this handle learned (iterations per call, the flush's plan) only fits its shape
        if (!roll_same_shape(r.shape, *launch) || r.per_call != K) r.iters = r.flush_iters = r.flush_live = 0;
        r.shape = *launch;
        r.lp = lp_in;
        r.lp.batch = 1u;
        r.lp.batch_paths = launch->n_paths;
        r.lp.batch_seeds = nullptr;              // seeds, offsets and buffers of a rolling render live in its descriptor
        r.lp.batch_offsets = nullptr;
        r.lp.box_slack = 0.f;
        r.lp.has_records = 0u;
        r.per_call = K;
        r.multi = false;
        r.offsets = with_offsets;
        r.dmax = 0.f;
        r.count = 0;
        r.it = 0;
        r.stream = stream;
        r.count_nodes = (launch->flags & BF_FLAG_STATS) != 0;
        r.timed = (launch->flags & BF_FLAG_TIMING) != 0;
        // LDS window: the newest renders' histogram blocks, as many as fit
        r.window = 1;
        r.lp.lds_hist = (lp_in.n_chan <= (uint32_t) bfd::kMaxLdsHist && !(launch->flags & BF_FLAG_GLOBAL_ATOMICS)) ? 1u : 0u;
        if (r.lp.lds_hist) r.window = std::max<uint32_t>(1u, std::min<uint32_t>(bfd::kRollWindow, (uint32_t) bfd::kMaxLdsHist / std::max(1u, lp_in.n_chan)));
        HIP_TRY(hipMemsetAsync(scene->counters, 0, sizeof(unsigned long long) * bfd::CTR_GUARD, stream));
        scene->wf_ev_kind.clear();
        scene->wf_iters = scene->wf_trace_launches = 0;
    }
    const bool fresh_pool = opening;
    const uint32_t k = r.count, newest = k + K - 1u;
    bfd::DLaunch &lp = r.lp;
    lp.n_paths = (uint64_t) (k + K) * lp.batch_paths;
    if (r.multi) {                 // the endpoints moved since the sequence was opened: per-path tables from now on (general kernels)
        lp.multi = 1u;
        lp.lean = 0u;
        scene->last_variant = 0u;
    }
    lp.roll_newest = newest;
    lp.roll_lo = newest + 1u > r.window ? newest + 1u - r.window : 0u;
    lp.n_chan_all = (lp.roll_newest - lp.roll_lo + 1u) * lp.n_chan;
    lp.base_off = lp.lds_hist ? r.window * lp.n_chan : 0u;        // fixed for the sequence: behind the full window
    lp.lds_floats = lp.base_off + 5u * bfd::kRollBase;
    WfCtx c;
    if (with_offsets) {
        for (uint32_t j = 0; j < K; ++j)
            for (int a = 0; a < 3; ++a) {
                const float q = batch->mesh_offsets[3 * j + a];
                if (!std::isfinite(q)) return fail(BF_ERR_INVALID, "bf_render_batch: non-finite mesh offset of render %u", j);
                r.dmax = std::max(r.dmax, std::fabs(q));
            }
        // bf_device_core.h: Shift — slack for the largest offset of the sequence so far (older paths just get wider boxes)
        lp.box_slack = 1e-6f * (scene->origin_scale_built + 2.f * r.dmax);
    }
    if ((st = wf_setup(scene, lp, (uint64_t) K * lp.batch_paths, nullptr, nullptr, stream, r.count_nodes, r.timed, c, true)) != BF_OK) return st;
    lp.roll = scene->roll_ring;           // wf_setup may have (re)allocated the pool and the ring with it
    lp.batch_offsets = with_offsets ? scene->roll_offsets : nullptr;
    scene->wf.offsets = lp.batch_offsets;
    {
        // the slots due for this call's paths: global paths [k P, (k + K) P) live in slots g mod n_main
        const uint32_t n_main = scene->wf.n_main, lo = (uint32_t) (((uint64_t) k * lp.batch_paths) % n_main);
        scene->wf.wake_b0 = lo / 64u;
        scene->wf.wake_nb = (uint32_t) std::min<uint64_t>(n_main / 64u, ((uint64_t) (lo % 64u) + (uint64_t) K * lp.batch_paths + 63u) / 64u);
    }
    if (fresh_pool) HIP_TRY(hipMemsetAsync(scene->wf.surv_cursor, 0, 2 * sizeof(uint32_t), stream));
    const uint32_t n_chan1 = lp.n_chan;
    if (records_dev) lp.has_records = 1u;
    for (uint32_t j = 0; j < K; ++j) {
        bfd::DRoll d;
        d.seed = (batch && batch->seeds) ? batch->seeds[j] : launch->seed;
        d.path_offset = launch->path_offset;
        d.hist = hist_dev + (size_t) j * n_chan1;
        d.records = records_dev ? records_dev + (size_t) j * lp.batch_paths : nullptr;
        d.rects = scene->d.rects;                 // the endpoint tables as they stand for THIS render (kMulti kernels)
        d.shapes = scene->d.shapes;
        d.emitters = scene->d.emitters;
        d.materials = scene->d.materials;
        d.sensor = scene->d.sensor;
        d.c = scene->d.c;
        d.lambda_min = scene->d.lambda_min;
        d.lambda_max = scene->d.lambda_max;
        d.pad = 0u;
        HIP_TRY(bfk_roll_set(scene->roll_ring, scene->roll_offsets, (k + j) & (bfd::kRollRing - 1u), &d,
                             with_offsets ? batch->mesh_offsets + 3 * j : nullptr, stream));
    }
    // ---- how many bounce iterations this call enqueues ------------------------------------------------
    // A launch that finds fewer live slots than fill the chip a few times over runs at its latency floor whatever it
    // holds, so a call stops iterating once that few are left (roll_live, 1.5 x 2^20 by default: tools/r03_probe9.sh) and leaves them to the next
    // call's launches: too few iterations and too many paths have to move to the survivor area, too many and the late ones
    // run over a nearly empty pool.  Steered by the live counts that come back (without ever waiting for them).
    const bool fb_ready = scene->wf_fb_pending && hipEventQuery(scene->wf_fb_event) == hipSuccess;
    (void) hipGetLastError();
    if (fb_ready) {
        scene->wf_fb_pending = false;
        const uint32_t n = r.fb_call_iters;
        if (n && !r.fb_is_flush && !scene->tun.roll_iters) {
            const uint32_t *nl = scene->wf_feedback;
            uint32_t k = 0;
            // round 4: with the shading launches' fixed costs gone (the statistics atomics) a launch over a FULL pool is what
            // pays: a call of a big render stops after its first iteration as long as at most a quarter of the main slots
            // is alive (C2 / C5: one iteration per call instead of three: wf_trace 3.80 -> 3.27 ms per step, profiles/
            // r04_roll_iterations.txt); what is still alive two calls later moves to the survivor area as before
            const uint32_t live_max = scene->tun.roll_live ? scene->tun.roll_live : std::max<uint32_t>(3u << 19, scene->wf.n_main / 4u);
            while (k < n && nl[k] > live_max) ++k;
            r.iters = std::min<uint32_t>(k < n ? k + 1u : n + 1u, 16u);
        }
        r.fb_call_iters = 0;
    }
    if (scene->tun.roll_iters) r.iters = scene->tun.roll_iters;
    if (r.iters == 0) r.iters = 2;
    if (r.it + 2u * 64u > bfd::kWfMaxIter) r.it &= 1u;       // the live-counter ring: keep the parity, restart the index
    const uint32_t it0 = r.it, I = r.iters;
    HIP_TRY(hipMemsetAsync(scene->wf.n_live + it0, 0, I * sizeof(uint32_t), stream));
    for (uint32_t j = 0; j < I; ++j) {
        const uint32_t it = r.it;
        if ((st = wf_iteration(c, it, j == 0 ? (opening ? 1 : 2) : 0)) != BF_OK) return st;
        if ((st = wf_trace_launch(c, it)) != BF_OK) return st;
        ++r.it;
    }
    scene->wf_iters += I;
    scene->wf_trace_launches += I;
    if (!scene->wf_fb_pending) {
        HIP_TRY(hipMemcpyAsync(scene->wf_feedback, scene->wf.n_live + it0, I * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipEventRecord(scene->wf_fb_event, stream));
        scene->wf_fb_pending = true;
        r.fb_call_iters = I;
        r.fb_is_flush = false;
    }
    if (!r.open) scene->peers_rolling->fetch_add(1, std::memory_order_relaxed);
    r.open = true;
    r.count += K;
    return BF_OK;
}

// Finish every path of the open sequence: bounce iterations until few slots are alive, then ONE tail launch.  The first
// flush of a sequence shape runs synchronously (the host reads the live count per iteration, as the first render of a
// shape does) and learns how many iterations that takes; later ones enqueue that many without a host round trip — the
// tail finishes whatever is alive, the estimate only sizes its grid.
static bf_status wf_roll_flush(const bf_scene *scene, hipStream_t stream, bool sync_timing) {
    bf_scene::Roll &r = scene->roll;
    if (!r.open) return BF_OK;
    bf_status st;
    if (stream != r.stream) {
        HIP_TRY(hipEventRecord(scene->wf_event, r.stream));
        HIP_TRY(hipStreamWaitEvent(stream, scene->wf_event, 0));
    }
    bfd::DLaunch &lp = r.lp;
    WfCtx c;
    if ((st = wf_setup(scene, lp, (uint64_t) r.per_call * lp.batch_paths, nullptr, nullptr, stream, r.count_nodes, r.timed, c, true)) != BF_OK) return st;
    bfd::WF &wf = scene->wf;
    volatile uint32_t *hq = scene->wf_host;
    const bool fb_ready = scene->wf_fb_pending && hipEventQuery(scene->wf_fb_event) == hipSuccess;
    (void) hipGetLastError();
    if (fb_ready) {
        scene->wf_fb_pending = false;
        if (r.fb_is_flush && r.fb_call_iters) {
            // live counts of the last planned flush: first iteration after which the tail threshold was met
            const uint32_t *nl = scene->wf_feedback, n = r.fb_call_iters;
            uint32_t k = 0;
            while (k < n && nl[k] > c.tail_max) ++k;
            if (k < n) {
                r.flush_iters = k + 1;
                r.flush_live = nl[k];
            } else {
                r.flush_iters = std::min<uint32_t>(n + 2, 48u);
                r.flush_live = nl[n - 1];
            }
        }
        r.fb_call_iters = 0;
    }
    if (r.it + 2u * 64u > bfd::kWfMaxIter) r.it &= 1u;
    const uint32_t it0 = r.it;
    HIP_TRY(hipMemsetAsync(wf.n_live + it0, 0, 64 * sizeof(uint32_t), stream));
    uint32_t done_iters = 0;
    if (scene->tun.allow_plan && r.flush_iters > 0) {
        for (uint32_t j = 0; j < r.flush_iters; ++j) {
            if ((st = wf_iteration(c, r.it, 0)) != BF_OK) return st;
            if ((st = wf_trace_launch(c, r.it)) != BF_OK) return st;
            ++r.it;
            ++done_iters;
        }
        const uint32_t est = std::max<uint32_t>(r.flush_live + r.flush_live / 4, 64u * bfd::kBlock);
        if ((st = wf_tail_launch(c, r.it, est, true)) != BF_OK) return st;
        if (!scene->wf_fb_pending && done_iters) {
            HIP_TRY(hipMemcpyAsync(scene->wf_feedback, wf.n_live + it0, done_iters * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipEventRecord(scene->wf_fb_event, stream));
            scene->wf_fb_pending = true;
            r.fb_call_iters = done_iters;
            r.fb_is_flush = true;
        }
    } else {
        uint32_t n_live = 0;
        for (uint32_t j = 0; j < 48u; ++j) {
            if ((st = wf_iteration(c, r.it, 0)) != BF_OK) return st;
            HIP_TRY(hipMemcpyAsync((void *) &hq[0], wf.n_live + r.it, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipEventRecord(scene->wf_event, stream));
            if ((st = wf_trace_launch(c, r.it)) != BF_OK) return st;
            HIP_TRY(hipEventSynchronize(scene->wf_event));
            ++r.it;
            ++done_iters;
            n_live = hq[0];
            if (n_live <= c.tail_max) break;
        }
        r.flush_iters = done_iters;
        r.flush_live = n_live;
        if (n_live) {
            if ((st = wf_tail_launch(c, r.it, n_live, true)) != BF_OK) return st;
        }
    }
    scene->wf_iters += done_iters;
    scene->wf_trace_launches += done_iters;
    r.open = false;
    scene->peers_rolling->fetch_sub(1, std::memory_order_relaxed);
    if (scene->tables_in_pool) {
        // the sequence's last table version becomes the handle's tables again (home buffers), behind the flush's kernels
        const bf_scene::TabLayout &t = scene->tab;
        const char *blk = (const char *) scene->d.rects - t.o_rects;
        const uint32_t nr = scene->d.n_rects, ne = scene->d.n_emitters, ns = scene->info.n_shapes, nm = scene->n_materials;
        if (nr) HIP_TRY(hipMemcpyAsync((void *) scene->home_rects, blk + t.o_rects, nr * sizeof(bfd::DRect), hipMemcpyDeviceToDevice, stream));
        if (ns) HIP_TRY(hipMemcpyAsync((void *) scene->home_shapes, blk + t.o_shapes, ns * sizeof(bfd::DShape), hipMemcpyDeviceToDevice, stream));
        if (ne) HIP_TRY(hipMemcpyAsync((void *) scene->home_emitters, blk + t.o_emit, ne * sizeof(bfd::DEmitter), hipMemcpyDeviceToDevice, stream));
        if (nm) HIP_TRY(hipMemcpyAsync((void *) scene->home_materials, blk + t.o_mat, nm * sizeof(bfd::DMaterial), hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipMemcpyAsync((void *) scene->home_sensor, blk + t.o_sensor, sizeof(bfd::DSensor), hipMemcpyDeviceToDevice, stream));
        bfd::DScene &d = const_cast<bf_scene *>(scene)->d;
        d.rects = scene->home_rects;
        d.shapes = scene->home_shapes;
        d.emitters = scene->home_emitters;
        d.materials = scene->home_materials;
        d.sensor = scene->home_sensor;
        scene->tables_in_pool = false;
        scene->tab_next = 0;
    }
    r.multi = false;
    lp.multi = 0u;
    if (sync_timing) return wf_collect_timing(scene, stream);
    return BF_OK;
}

// counters -> bf_stats (+ the per-kernel times of the last timed render / sequence)
static void fill_stats(const bf_scene *scene, const unsigned long long *c, uint64_t n_paths, bf_stats *st) {
    std::memset(st, 0, sizeof(*st));
    st->n_paths = n_paths;
    st->n_rays_closest = c[bfd::CTR_CLOSEST];
    st->n_rays_shadow = c[bfd::CTR_SHADOW];
    st->n_nodes_visited = c[bfd::CTR_NODES];
    st->n_nodes_lds = c[bfd::CTR_NODES_LDS];
    st->n_tris_tested = c[bfd::CTR_TRIS];
    st->n_invalid = c[bfd::CTR_INVALID];
    st->n_bounces = c[bfd::CTR_BOUNCES];
    st->n_rays_tail = c[bfd::CTR_TAIL_RAYS];
    st->n_rays_traced = c[bfd::CTR_TRACED];
    st->n_nodes_tail = c[bfd::CTR_TAIL_NODES];
    st->n_wnodes_tail = c[bfd::CTR_TAIL_WNODES];
    st->n_tris_tail = c[bfd::CTR_TAIL_TRIS];
    st->n_bounces_tail = c[bfd::CTR_TAIL_BOUNCES];
    st->n_shade_loads = c[bfd::CTR_SHADE_LOADS];
    st->n_shade_stores = c[bfd::CTR_SHADE_STORES];
    st->n_shade_shadow = c[bfd::CTR_SHADE_SHADOW];
    st->n_shade_rays = c[bfd::CTR_SHADE_RAYS];
    st->n_guard = c[bfd::CTR_GUARD];
    st->trace_ms = scene->wf_ms[0];
    st->shade_ms = scene->wf_ms[1];
    st->tail_ms = scene->wf_ms[2];
    st->n_launches_trace = scene->wf_trace_launches;
    st->n_bounce_iters = scene->wf_iters;
    st->n_launches_tail = scene->wf_tail_launches;
    st->n_launches_shade = scene->wf_shade_launches;
    st->kernel_variant = scene->last_variant;
}

// Stream order between the successive uses of a handle's pool: work enqueued on another stream than the previous
// call's waits for it (an event wait on the device, never on the host).
static bf_status order_after_last(const bf_scene *scene, hipStream_t stream) {
    if (scene->has_last && scene->last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, scene->last_done, 0));
    return BF_OK;
}
static bf_status mark_last(const bf_scene *scene, hipStream_t stream) {
    if (!scene->last_done) HIP_TRY(hipEventCreateWithFlags(&scene->last_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(scene->last_done, stream));
    scene->last_stream = stream;
    scene->has_last = true;
    return BF_OK;
}
// anything but another render of the open rolling sequence needs the pool (or the scene tables) to itself
static bf_status close_sequence(const bf_scene *scene, hipStream_t stream) {
    if (!scene->roll.open) return BF_OK;
    bf_status st = wf_roll_flush(scene, stream, false);
    if (st != BF_OK) return st;
    return mark_last(scene, stream);
}

// The lean profile (bf_device.h: kLean): what the scene and the launch must look like for the kernels that have everything else
// compiled out.  Every radar scene of the reference's scripts and every BASELINE config fits; anything else runs the
// general kernels (same results: tests/test_gpu_parity.py::test_lean_and_general_kernels_agree).
static bool lean_profile(const bf_scene *scene, const bf_launch *launch, bool receive_mode, bool multi_pixel) {
    // lean builds exist of the default register budgets only (three waves per SIMD), and not of the one-kernel variant
    if ((launch->flags & BF_FLAG_MEGAKERNEL) || scene->tun.shade_waves != 3 || scene->tun.tail_waves != 3) return false;
    if (!scene->tun.lean || scene->d.n_emitters != 1 || scene->d.uvs != nullptr || scene->sensor_host.filt_n != 0u) return false;
    if (scene->sensor_host.win_off_t || scene->sensor_host.win_off_f) return false;      // ADC window away from the origin
    if (scene->sensor_host.crop_x || scene->sensor_host.crop_y) return false;            // film crop window away from the origin
    if (scene->any_back_material) return false;                                            // twosided with two nested BSDFs
    if (scene->any_resample) return false;                                                 // resample_freq transmitters
    const uint32_t et = scene->emitter_types[0];
    if (receive_mode)
        return (et == BF_TRANSMITTER_AREA || et == BF_TRANSMITTER_WIGNER) && scene->sensor_host.type == BF_RECEIVER_OMNI &&
               launch->phase_bins == 0 && !(launch->flags & (BF_FLAG_DOPPLER | BF_FLAG_MIX_RESAMPLE));
    return et == BF_EMITTER_AREA && scene->sensor_host.type == BF_SENSOR_PERSPECTIVE && !multi_pixel && launch->mode != BF_MODE_TIME;
}

static bf_status render_common(const bf_scene *scene, const bf_launch *launch, const bf_batch *batch, float *hist_dev,
                               bf_path_record *records_dev, void *stream_, bf_stats *stats_out) {
    if (!scene || !launch || !hist_dev) return fail(BF_ERR_INVALID, "null argument");
    BF_ENTER(scene);
    const uint32_t n_renders = batch ? batch->n_renders : 1u;
    if (batch) {
        if (n_renders == 0) return fail(BF_ERR_INVALID, "bf_render_batch: n_renders is 0");
        if (launch->spp && launch->film_width && launch->film_height)
            return fail(BF_ERR_UNSUPPORTED, "bf_render_batch: multi-pixel films are rendered one launch at a time");
        if ((uint64_t) n_renders * bf_launch_channels(launch) > (1ull << 31) || (uint64_t) n_renders * launch->n_paths >= (1ull << 48))
            return fail(BF_ERR_UNSUPPORTED, "bf_render_batch: %u renders x %llu paths is too large", n_renders, (unsigned long long) launch->n_paths);
    }
    const bool is_rx = scene->sensor_host.type == BF_RECEIVER_OMNI || scene->sensor_host.type == BF_RECEIVER_WIGNER ||
                       scene->sensor_host.type == BF_RECEIVER_PHASED;
    const bool receive_mode = launch->mode == BF_MODE_RECEIVE_RAW || launch->mode == BF_MODE_RECEIVE_IQ;
    if (receive_mode) {
        if (!is_rx) return fail(BF_ERR_INVALID, "receive mode needs a receiver (omnidirectional / wigner)");
        if (launch->bins != scene->adc_t || launch->bins_y != scene->adc_f)
            return fail(BF_ERR_INVALID, "receive mode: launch bins (%u x %u) must equal the ADC size — its window, if it has one — (%u x %u)",
                        launch->bins, launch->bins_y, scene->adc_t, scene->adc_f);
        for (uint32_t i = 0; i < scene->d.n_emitters; ++i)
            if (scene->emitter_types[i] != BF_TRANSMITTER_AREA && scene->emitter_types[i] != BF_TRANSMITTER_WIGNER &&
                scene->emitter_types[i] != BF_TRANSMITTER_PHASED)
                return fail(BF_ERR_INVALID, "receive mode: emitter %u is not a transmitter", i);
        // the Wigner and phased receivers sample their own local-oscillator signal under "mix_resample" (wignerreceiver.cpp:72-110,
        // 172-189): a delta signal's instantaneous frequency at the receive time (sample_delta_frequency :149-166: a chirp's or a
        // carrier's; "pulse" leaves it uninitialised there: refused), or a uniform frequency weighted with eval_signal
        if ((launch->flags & BF_FLAG_MIX_RESAMPLE) && scene->sensor_host.type != BF_RECEIVER_OMNI) {
            if (scene->sensor_host.rx_sig_is_delta && scene->sensor_host.rx_signal == BF_SIGNAL_PULSE)
                return fail(BF_ERR_UNSUPPORTED, "BF_FLAG_MIX_RESAMPLE on the Wigner / phased receiver: a \"pulse\" local oscillator that is a delta "
                                                "signal reads an uninitialised frequency in the reference (wignerreceiver.cpp:149-166)");
            if (scene->sensor_host.rx_signal != BF_SIGNAL_CW && !(scene->sensor_host.rx_pulse_len > 0.f && scene->sensor_host.rx_prf > 0.f))
                return fail(BF_ERR_INVALID, "BF_FLAG_MIX_RESAMPLE: the receiver's chirp / pulse needs rx_pulse_len > 0 and rx_prf > 0");
        }
    } else {
        if (launch->flags & BF_FLAG_MIX_RESAMPLE) return fail(BF_ERR_INVALID, "BF_FLAG_MIX_RESAMPLE is a receive-mode flag");
        if (is_rx) return fail(BF_ERR_INVALID, "render modes need a sensor (fluxmeter / perspective), not a receiver");
        for (uint32_t i = 0; i < scene->d.n_emitters; ++i)
            if (scene->emitter_types[i] != BF_EMITTER_SPOT && scene->emitter_types[i] != BF_EMITTER_AREA &&
                scene->emitter_types[i] != BF_EMITTER_POINT)
                return fail(BF_ERR_INVALID, "render modes: emitter %u is a transmitter (use receive mode)", i);
    }
    if (launch->mode > BF_MODE_RECEIVE_IQ) return fail(BF_ERR_INVALID, "unknown mode %u", launch->mode);
    if (launch->mode != BF_MODE_RECEIVE_RAW && launch->phase_bins)
        return fail(BF_ERR_INVALID, "phase_bins needs receive mode (PhaseIntegrator wraps pathtimefrequency)");
    if (launch->phase_bins > 4096) return fail(BF_ERR_INVALID, "phase_bins %u out of range", launch->phase_bins);
    if ((launch->mode == BF_MODE_RANGE || launch->mode == BF_MODE_TIME) && (launch->bins == 0 || !(launch->bin_width > 0.f)))
        return fail(BF_ERR_INVALID, "range/time mode needs bins > 0 and bin_width > 0");
    const bool multi_pixel = launch->spp && launch->film_width && launch->film_height;
    if (multi_pixel ? (launch->film_width != scene->film_w || launch->film_height != scene->film_h)
                    : (scene->film_w != 1 || scene->film_h != 1) && !is_rx)
        return fail(BF_ERR_INVALID, "the sensor's film is %u x %u: the launch must name the same film and spp > 0 (it has %u x %u, spp %u)",
                    scene->film_w, scene->film_h, launch->film_width, launch->film_height, launch->spp);
    if (multi_pixel) {
        if (launch->mode == BF_MODE_RECEIVE_RAW || launch->mode == BF_MODE_RECEIVE_IQ)
            return fail(BF_ERR_INVALID, "receive modes bin into the ADC: film_width / film_height / spp must be 0");
        const uint64_t px = (uint64_t) launch->film_width * launch->film_height;
        if (px > (1u << 24) || px * (5ull + 3ull * launch->bins) > (1ull << 31))
            return fail(BF_ERR_UNSUPPORTED, "film %u x %u with %u bins is too large", launch->film_width, launch->film_height, launch->bins);
        if (launch->path_offset + launch->n_paths > px * launch->spp)
            return fail(BF_ERR_INVALID, "path_offset + n_paths = %llu exceeds film_width * film_height * spp = %llu",
                        (unsigned long long) (launch->path_offset + launch->n_paths), (unsigned long long) (px * launch->spp));
    }
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const bool rolling = (launch->flags & BF_FLAG_ROLLING) != 0u && launch->n_paths != 0u &&
                         (uint64_t) n_renders * launch->n_paths <= scene->tun.pool && n_renders <= bfd::kRollRing;
    if (launch->flags & BF_FLAG_ROLLING) {
        if (multi_pixel || (launch->flags & BF_FLAG_MEGAKERNEL))
            return fail(BF_ERR_UNSUPPORTED, "BF_FLAG_ROLLING: multi-pixel films and the one-kernel variant do not roll");
        if (stats_out)
            return fail(BF_ERR_INVALID, "BF_FLAG_ROLLING: a rolling render returns before its paths have ended, so it has no statistics "
                                        "of its own (pass stats_out = NULL; bf_scene_flush reports the sequence's)");
    }
    {
        bf_status ost = order_after_last(scene, stream);
        if (ost != BF_OK) return ost;
        if (!rolling && (ost = close_sequence(scene, stream)) != BF_OK) return ost;
    }
    bfd::DLaunch lp;
    std::memset(&lp, 0, sizeof(lp));
    lp.film_w = multi_pixel ? launch->film_width : 1u;
    lp.film_h = multi_pixel ? launch->film_height : 1u;
    lp.spp = multi_pixel ? launch->spp : 0u;
    lp.mode = launch->mode;
    lp.color_mode = launch->color_mode;
    lp.n_paths = launch->n_paths;
    lp.path_offset = launch->path_offset;
    lp.seed = launch->seed;
    lp.max_depth = launch->max_depth;
    lp.rr_depth = launch->rr_depth;
    lp.bins = launch->bins;
    lp.bins_y = launch->bins_y;
    lp.phase_bins = launch->mode == BF_MODE_RECEIVE_RAW ? launch->phase_bins : 0u;
    lp.iq = launch->mode == BF_MODE_RECEIVE_IQ ? 1u : 0u;
    if (lp.iq) lp.mode = BF_MODE_RECEIVE_RAW;          // the kernels see receive mode + the iq flag
    lp.bin_width = launch->bin_width;
    lp.time_c = launch->time_c;
    lp.n_chan = bf_launch_channels(launch);
    lp.chan_px = lp.n_chan / (lp.film_w * lp.film_h);
    lp.lean = lean_profile(scene, launch, receive_mode, multi_pixel) ? 1u : 0u;
    lp.wide = scene->sensor_host.filt_n != 0u ? 1u : 0u;
    scene->last_variant = (lp.lean ? (uint32_t) BF_VARIANT_LEAN : 0u) | (lp.wide ? (uint32_t) BF_VARIANT_WIDE : 0u);      // reconstruction filter wider than a pixel: the kernels' kWide variants
    lp.count = ((launch->flags & (BF_FLAG_STATS | BF_FLAG_COUNT)) || stats_out) ? 1u : 0u;
    lp.doppler = (receive_mode && (launch->flags & BF_FLAG_DOPPLER)) ? 1u : 0u;
    lp.resample = (receive_mode && scene->any_resample) ? 1u : 0u;
    if (lp.resample && lp.doppler)
        return fail(BF_ERR_UNSUPPORTED, "BF_FLAG_DOPPLER with a resample_freq transmitter: both rewrite the path's wavelength (one slot of path state)");
    lp.mix = (receive_mode && (launch->flags & BF_FLAG_MIX_RESAMPLE)) ? 1u : 0u;
    lp.n_chan_all = lp.n_chan * n_renders;
    lp.lds_hist = (lp.n_chan_all <= (uint32_t) bfd::kMaxLdsHist && !(launch->flags & BF_FLAG_GLOBAL_ATOMICS)) ? 1u : 0u;
    lp.lds_floats = lp.lds_hist ? lp.n_chan_all : 0u;
    size_t lds = sizeof(int) * bfd::kStackDepth * bfd::kBlock + (lp.lds_hist ? sizeof(float) * lp.n_chan_all : 0);
    lds = ((lds + 15) & ~size_t(15)) + (scene->d.tab_cache ? bfd::kTabBytes : 0u);      // stacks | histogram | tables (bf_device_core.h: load_tables_lds)
    if (batch && !rolling) {
        // one launch sequence over n_renders * n_paths global path indices (DLaunch::batch); the per-render seeds and
        // mesh offsets travel through the scene's pinned staging ring, so the caller's arrays are free on return
        lp.batch = n_renders;
        lp.batch_paths = launch->n_paths;
        lp.n_paths = launch->n_paths * n_renders;
        // the offsets are read as float4 on both sides: keep them 16-byte aligned behind the seeds
        const size_t seed_bytes = batch->seeds ? ((sizeof(uint64_t) * n_renders + 15) & ~size_t(15)) : 0;
        const size_t off_bytes = batch->mesh_offsets ? sizeof(float4) * n_renders : 0;
        if (seed_bytes + off_bytes) {
            bf_scene::Stage *stg = nullptr;
            bf_status sst = stage_acquire(scene, seed_bytes + off_bytes, &stg);
            if (sst != BF_OK) return sst;
            if (seed_bytes) {
                std::memcpy(stg->host, batch->seeds, sizeof(uint64_t) * n_renders);
                lp.batch_seeds = (const uint64_t *) stg->dev;
            }
            if (off_bytes) {
                float4 *o = (float4 *) ((char *) stg->host + seed_bytes);
                float dmax = 0.f;
                for (uint32_t k = 0; k < n_renders; ++k) {
                    const float *q = batch->mesh_offsets + 3 * k;
                    if (!(std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2])))
                        return fail(BF_ERR_INVALID, "bf_render_batch: non-finite mesh offset of render %u", k);
                    o[k] = make_float4(q[0], q[1], q[2], 0.f);
                    dmax = std::max({dmax, std::fabs(q[0]), std::fabs(q[1]), std::fabs(q[2])});
                }
                if (scene->d.n_tris) {
                    lp.batch_offsets = (const float4 *) ((char *) stg->dev + seed_bytes);
                    // bf_device_core.h: Shift — the roundings of o - d and p + d move a box plane by at most
                    // 1.8e-7 (S + 2 |d|), S = the bound the boxes were padded for
                    lp.box_slack = 1e-6f * (scene->origin_scale_built + 2.f * dmax);
                }
            }
            sst = stage_commit(stg, seed_bytes + off_bytes, stream);
            if (sst != BF_OK) return sst;
        }
    }

    // persistent grid: enough workgroups to fill the chip, never more than the work
    uint64_t want = (lp.n_paths + bfd::kBlock - 1) / bfd::kBlock;
    unsigned blocks_per_cu = (unsigned) std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / lds));
    unsigned grid = (unsigned) std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t) scene->n_cus * blocks_per_cu));

    if (rolling) {
        // more paths than the pool has slots would need several paths per slot and render: not a rolling shape (above)
        bf_status rst = wf_roll_render(scene, launch, batch, lp, hist_dev, records_dev, stream);
        if (rst != BF_OK) return rst;
        return mark_last(scene, stream);
    }
    // every counter but the sticky guard word (the last one)
    HIP_TRY(hipMemsetAsync(scene->counters, 0, sizeof(unsigned long long) * bfd::CTR_GUARD, stream));
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (stats_out) {
        HIP_TRY(hipEventCreate(&ev0));
        HIP_TRY(hipEventCreate(&ev1));
        HIP_TRY(hipEventRecord(ev0, stream));
    }
    if (launch->n_paths) {
        if (launch->flags & BF_FLAG_MEGAKERNEL) {
            HIP_TRY(bfk_launch_render(&scene->d, &lp, hist_dev, records_dev, scene->counters,
                                      (launch->flags & BF_FLAG_STATS) ? 1 : 0, grid, lds, stream));
        } else {
            bf_status wst = wf_render(scene, lp, hist_dev, records_dev, stream, (launch->flags & BF_FLAG_STATS) != 0,
                                      stats_out != nullptr);
            if (wst != BF_OK) return wst;
        }
    }
    if (stats_out) {
        HIP_TRY(hipEventRecord(ev1, stream));
        HIP_TRY(hipEventSynchronize(ev1));
        unsigned long long c[bfd::CTR_COUNT];
        HIP_TRY(hipMemcpy(c, scene->counters, sizeof(c), hipMemcpyDeviceToHost));
        if ((launch->flags & BF_FLAG_MEGAKERNEL) || !launch->n_paths) {
            scene->wf_ms[0] = scene->wf_ms[1] = scene->wf_ms[2] = 0.f;
            scene->wf_trace_launches = scene->wf_iters = scene->wf_tail_launches = scene->wf_shade_launches = 0;
        }
        fill_stats(scene, c, lp.n_paths, stats_out);
        float ms = 0.f;
        hipError_t he = hipEventElapsedTime(&ms, ev0, ev1);
        (void) hipEventDestroy(ev0);
        (void) hipEventDestroy(ev1);
        if (he != hipSuccess) return fail(BF_ERR_DEVICE, "hipEventElapsedTime: %s", hipGetErrorString(he));
        stats_out->kernel_ms = ms;
        if (c[bfd::CTR_GUARD] || c[bfd::CTR_SURV_GUARD]) {
            // reported here: clear the sticky words and the copy of them that may be in flight to the next planned render
            HIP_TRY(hipMemset(scene->counters + bfd::CTR_GUARD, 0, 2 * sizeof(unsigned long long)));
            scene->wf_fb_pending = false;
            return wf_guard_error(c[bfd::CTR_GUARD], c[bfd::CTR_SURV_GUARD]);
        }
        if (launch->n_paths && !(launch->flags & BF_FLAG_MEGAKERNEL)) {
            bf_status fst = check_film_count(c, lp.n_paths);
            if (fst != BF_OK) return fst;
        }
    }
    return mark_last(scene, stream);
}

bf_status bf_render_device(const bf_scene *scene, const bf_launch *launch, float *hist_dev, bf_path_record *records_dev,
                           void *stream, bf_stats *stats_out) {
    return render_common(scene, launch, nullptr, hist_dev, records_dev, stream, stats_out);
}

bf_status bf_render_batch_device(const bf_scene *scene, const bf_launch *launch, const bf_batch *batch, float *hist_dev,
                                 bf_path_record *records_dev, void *stream, bf_stats *stats_out) {
    if (!batch) return fail(BF_ERR_INVALID, "bf_render_batch_device: null batch");
    return render_common(scene, launch, batch, hist_dev, records_dev, stream, stats_out);
}

bf_status bf_scene_flush(bf_scene *scene, void *stream_, bf_stats *stats_out) {
    if (!scene) return fail(BF_ERR_INVALID, "null argument");
    BF_ENTER(scene);
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const bool was_open = scene->roll.open;
    const uint64_t n_paths = was_open ? scene->roll.lp.n_paths : 0;
    bf_status st = order_after_last(scene, stream);
    if (st != BF_OK) return st;
    if (was_open) {
        if ((st = wf_roll_flush(scene, stream, stats_out != nullptr)) != BF_OK) return st;
        if ((st = mark_last(scene, stream)) != BF_OK) return st;
    }
    if (stats_out) {
        std::memset(stats_out, 0, sizeof(*stats_out));
        if (!was_open) return BF_OK;
        HIP_TRY(hipStreamSynchronize(stream));
        unsigned long long c[bfd::CTR_COUNT];
        HIP_TRY(hipMemcpy(c, scene->counters, sizeof(c), hipMemcpyDeviceToHost));
        fill_stats(scene, c, n_paths, stats_out);
        stats_out->kernel_ms = scene->wf_ms[0] + scene->wf_ms[1] + scene->wf_ms[2];
        if (c[bfd::CTR_GUARD] || c[bfd::CTR_SURV_GUARD]) {
            HIP_TRY(hipMemset(scene->counters + bfd::CTR_GUARD, 0, 2 * sizeof(unsigned long long)));
            scene->wf_fb_pending = false;
            return wf_guard_error(c[bfd::CTR_GUARD], c[bfd::CTR_SURV_GUARD]);
        }
        return scene->roll.lp.count ? check_film_count(c, n_paths) : BF_OK;       // every path of every render of a COUNTED sequence
    }
    return BF_OK;
}

bf_status bf_scene_sync(bf_scene *scene) {
    if (!scene) return fail(BF_ERR_INVALID, "null argument");
    BF_ENTER(scene);
    bf_status st = close_sequence(scene, scene->roll.stream);
    if (st != BF_OK) return st;
    if (scene->has_last) HIP_TRY(hipEventSynchronize(scene->last_done));
    scene->wf_fb_pending = false;         // whatever feedback was in flight has landed; the next render re-learns from its own
    scene->roll.fb_call_iters = 0;
    unsigned long long lost = 0, refused = 0;
    bf_status gst = read_guards(scene, &lost, &refused);
    if (gst != BF_OK) return gst;
    if (lost || refused) return wf_guard_error(lost, refused);
    return BF_OK;
}

/* test hook (not part of the ABI): pre-load the sticky guard word, as if wf_trace had dropped `n` rays */
bf_status bfdbg_preload_guard(bf_scene *scene, unsigned long long n) {
    if (!scene) return fail(BF_ERR_INVALID, "null argument");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(scene->counters + bfd::CTR_GUARD, &n, sizeof(n), hipMemcpyHostToDevice));
    return BF_OK;
}

/* developer probe (tools/fetch_probe.py): `reps` launches of the gather pattern `mode` over a table of 2^log2_rows 16-byte rows
   (a second buffer of the same size is streamed in between, so every launch starts with a cold L2); returns the mean ms */
extern "C" hipError_t bfk_gather_probe(int mode, const float4 *table, uint32_t n_rows, float4 *out, hipStream_t stream);
bf_status bfdbg_gather_probe(int mode, uint32_t log2_rows, uint32_t reps, float *ms_out) {
    if (mode < 0 || mode > 2 || log2_rows < 10 || log2_rows > 28 || reps == 0) return fail(BF_ERR_INVALID, "bfdbg_gather_probe: bad arguments");
    const uint32_t n = 1u << log2_rows;
    float4 *table = nullptr, *flush = nullptr, *out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void **) &table, (size_t) n * 16);
    if (e == hipSuccess) e = hipMalloc((void **) &flush, (size_t) n * 16);
    if (e == hipSuccess) e = hipMalloc((void **) &out, 1024 * 16);
    if (e == hipSuccess) e = hipMemset(table, 0, (size_t) n * 16);
    if (e == hipSuccess) e = hipMemset(flush, 0, (size_t) n * 16);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float total = 0.f;
    for (uint32_t r = 0; r < reps && e == hipSuccess; ++r) {
        e = hipMemsetAsync(flush, (int) (r & 1u), (size_t) n * 16, nullptr);
        if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
        if (e == hipSuccess) e = bfk_gather_probe(mode, table, n, out, nullptr);
        if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        total += ms;
    }
    if (e0) (void) hipEventDestroy(e0);
    if (e1) (void) hipEventDestroy(e1);
    (void) hipFree(table);
    (void) hipFree(flush);
    (void) hipFree(out);
    if (e != hipSuccess) return fail(BF_ERR_DEVICE, "bfdbg_gather_probe: %s", hipGetErrorString(e));
    if (ms_out) *ms_out = total / (float) reps;
    return BF_OK;
}

/* test hook: take (1) / release (0) the handle's busy flag, as a call of another host thread would hold it */
bf_status bfdbg_hold_busy(bf_scene *scene, int on) {
    if (!scene) return fail(BF_ERR_INVALID, "null argument");
    if (on) {
        if (scene->busy.test_and_set(std::memory_order_acquire)) return fail(BF_ERR_INVALID, "bfdbg_hold_busy: already held");
    } else {
        scene->busy.clear(std::memory_order_release);
    }
    return BF_OK;
}


// ---------------------------------------------------------------------------
// One process, several GPUs (bf_launch.device_mask; 8e: sample shards + one all-reduce).  Paths are i.i.d.:
// GPU g of G renders the global path indices bf_shard_range(n, g, G) of the launch through bf_launch.path_offset — the
// union is the sample set of a one-GPU render — and the per-GPU range histograms are summed by ONE ncclAllReduce(float,
// sum) over xGMI.  RCCL is loaded on first use (dlopen: a process that never shards needs no librccl, and one that
// already carries a copy — PyTorch ships its own — keeps using that one).
// ---------------------------------------------------------------------------
void bf_shard_range(uint64_t n_paths, uint32_t shard, uint32_t n_shards, uint64_t *offset, uint64_t *count) {
    if (n_shards == 0) n_shards = 1;
    // floor(n s / S) without overflowing 64 bits for n < 2^63, S < 2^32
    auto cut = [&](uint64_t k) -> uint64_t { return (uint64_t) (((unsigned __int128) n_paths * k) / n_shards); };
    const uint64_t lo = cut(shard), hi = cut((uint64_t) shard + 1u);
    if (offset) *offset = lo;
    if (count) *count = hi - lo;
}

extern "C++" {
namespace {
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    std::string why;
};
Rccl &rccl() {
    static Rccl r = [] {
        Rccl q;
        // BF_RCCL_LIB names the library to load instead (a site build; the tests force the not-found path with it)
        const char *forced = getenv("BF_RCCL_LIB");
        const char *names[] = {forced && *forced ? forced : "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        const size_t n_names = forced && *forced ? 1 : 3;
        for (size_t i = 0; i < n_names && !q.lib; ++i)      // a copy that is already in the process (torch's) first
            q.lib = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
        std::string last;
        for (size_t i = 0; i < n_names && !q.lib; ++i) {
            q.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
            if (!q.lib) {
                const char *e = dlerror();      // ONE call: dlerror() clears the message it returns
                last = e ? e : "";
            }
        }
        if (!q.lib) {
            q.why = std::string(forced && *forced ? forced : "librccl.so") + " not found: " + last;
            return q;
        }
        q.comm_init_all = (decltype(q.comm_init_all)) dlsym(q.lib, "ncclCommInitAll");
        q.comm_destroy = (decltype(q.comm_destroy)) dlsym(q.lib, "ncclCommDestroy");
        q.all_reduce = (decltype(q.all_reduce)) dlsym(q.lib, "ncclAllReduce");
        q.group_start = (decltype(q.group_start)) dlsym(q.lib, "ncclGroupStart");
        q.group_end = (decltype(q.group_end)) dlsym(q.lib, "ncclGroupEnd");
        q.error_string = (decltype(q.error_string)) dlsym(q.lib, "ncclGetErrorString");
        if (!q.comm_init_all || !q.comm_destroy || !q.all_reduce || !q.group_start || !q.group_end || !q.error_string) {
            q.why = "librccl.so lacks ncclCommInitAll / ncclAllReduce / ncclGroupStart / ncclGroupEnd";
            q.lib = nullptr;
        }
        return q;
    }();
    return r;
}
// one communicator set per ordered device list, created on first use and kept for the life of the process
std::mutex g_comm_mutex;
std::map<std::vector<int>, std::vector<ncclComm_t>> g_comms;
}  // namespace
}  // extern "C++"

bf_status bf_allreduce_device(const int *devices, uint32_t n_devices, float *const *bufs, uint64_t count, void *const *streams) {
    if (!devices || !bufs || n_devices == 0) return fail(BF_ERR_INVALID, "bf_allreduce_device: null argument");
    if (count == 0) return BF_OK;
    for (uint32_t g = 0; g < n_devices; ++g)
        if (!bufs[g]) return fail(BF_ERR_INVALID, "bf_allreduce_device: buffer %u is null", g);
    Rccl &r = rccl();
    if (!r.lib) return fail(BF_ERR_UNSUPPORTED, "bf_allreduce_device: %s", r.why.c_str());
    std::vector<int> key(devices, devices + n_devices);
    for (uint32_t a = 0; a < n_devices; ++a)
        for (uint32_t b = a + 1; b < n_devices; ++b)
            if (key[a] == key[b]) return fail(BF_ERR_INVALID, "bf_allreduce_device: device %d is listed twice", key[a]);
    std::lock_guard<std::mutex> lock(g_comm_mutex);
    auto it = g_comms.find(key);
    if (it == g_comms.end()) {
        std::vector<ncclComm_t> comms(n_devices);
        ncclResult_t nr = r.comm_init_all(comms.data(), (int) n_devices, key.data());
        if (nr != ncclSuccess) return fail(BF_ERR_DEVICE, "ncclCommInitAll(%u devices): %s", n_devices, r.error_string(nr));
        it = g_comms.emplace(key, std::move(comms)).first;
    }
    int prev = -1;
    (void) hipGetDevice(&prev);
    ncclResult_t nr = r.group_start();
    for (uint32_t g = 0; g < n_devices && nr == ncclSuccess; ++g) {
        if (hipSetDevice(key[g]) != hipSuccess) {
            (void) r.group_end();
            if (prev >= 0) (void) hipSetDevice(prev);
            return fail(BF_ERR_DEVICE, "bf_allreduce_device: hipSetDevice(%d) failed", key[g]);
        }
        nr = r.all_reduce(bufs[g], bufs[g], (size_t) count, ncclFloat, ncclSum, it->second[g],
                          reinterpret_cast<hipStream_t>(streams ? streams[g] : nullptr));
    }
    ncclResult_t ne = r.group_end();
    if (prev >= 0) (void) hipSetDevice(prev);
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return fail(BF_ERR_DEVICE, "ncclAllReduce: %s", r.error_string(nr));
    return BF_OK;
}

bf_status bf_render_sharded_device(bf_scene *const *scenes, uint32_t n_devices, const bf_launch *launch, float *const *hist_dev,
                                   void *const *streams, bf_stats *stats_out) {
    if (!scenes || !launch || !hist_dev || n_devices == 0) return fail(BF_ERR_INVALID, "bf_render_sharded_device: null argument");
    std::vector<int> devices(n_devices);
    for (uint32_t g = 0; g < n_devices; ++g) {
        if (!scenes[g] || !hist_dev[g]) return fail(BF_ERR_INVALID, "bf_render_sharded_device: scene / histogram %u is null", g);
        devices[g] = scenes[g]->device;
    }
    if (stats_out) std::memset(stats_out, 0, sizeof(*stats_out));
    const bool rolling = (launch->flags & BF_FLAG_ROLLING) != 0u;
    if (stats_out && rolling)
        return fail(BF_ERR_INVALID, "BF_FLAG_ROLLING: a rolling render has no statistics of its own (bf_scene_flush reports the sequence's)");
    // Every GPU's launches are enqueued before any of them is waited for — also when statistics are asked for: a render
    // with stats_out is synchronous per device (it would make the GPUs take turns), so the shards are issued WITHOUT, each
    // between two events on its own stream, and the counters are read per device once all of them are in flight.
    std::vector<hipEvent_t> ev(stats_out ? 2 * (size_t) n_devices : 0, nullptr);
    std::vector<uint64_t> shard_paths(n_devices, 0);
    auto drop_events = [&]() {
        for (hipEvent_t e : ev)
            if (e) (void) hipEventDestroy(e);
    };
    for (uint32_t g = 0; g < n_devices; ++g) {
        bf_launch lg = *launch;
        uint64_t off = 0, cnt = 0;
        bf_shard_range(launch->n_paths, g, n_devices, &off, &cnt);
        lg.n_paths = cnt;
        lg.path_offset = launch->path_offset + off;
        shard_paths[g] = cnt;
        if (cnt == 0) continue;
        hipStream_t sg = reinterpret_cast<hipStream_t>(streams ? streams[g] : nullptr);
        if (stats_out) {
            DeviceGuard on_device(devices[g]);
            hipError_t he = hipEventCreate(&ev[2 * g]);
            if (he == hipSuccess) he = hipEventCreate(&ev[2 * g + 1]);
            if (he == hipSuccess) he = hipEventRecord(ev[2 * g], sg);
            if (he != hipSuccess) {
                drop_events();
                return fail(BF_ERR_DEVICE, "bf_render_sharded_device: device %d: %s", devices[g], hipGetErrorString(he));
            }
        }
        if (stats_out) lg.flags |= BF_FLAG_COUNT;      // (the counters are read below, once every device is in flight)
        bf_status st = bf_render_device(scenes[g], &lg, hist_dev[g], nullptr, sg, nullptr);
        if (st == BF_OK && stats_out) {
            DeviceGuard on_device(devices[g]);
            if (hipEventRecord(ev[2 * g + 1], sg) != hipSuccess) st = fail(BF_ERR_DEVICE, "bf_render_sharded_device: hipEventRecord failed");
        }
        if (st != BF_OK) {
            drop_events();
            return st;
        }
    }
    bf_status rst = BF_OK;
    // a rolling render's histogram is complete only after bf_scene_flush: the caller reduces then (bf_allreduce_device);
    // a communicator of one: the sum is the histogram itself
    if (!rolling && n_devices > 1) rst = bf_allreduce_device(devices.data(), n_devices, hist_dev, bf_launch_channels(launch), streams);
    if (stats_out) {
        unsigned long long lost = 0, refused = 0;
        for (uint32_t g = 0; g < n_devices && rst == BF_OK; ++g) {
            if (shard_paths[g] == 0) continue;
            BF_ENTER(scenes[g]);                 // (this handle's device; the counters are the handle's)
            hipError_t he = hipEventSynchronize(ev[2 * g + 1]);
            unsigned long long c[bfd::CTR_COUNT];
            if (he == hipSuccess) he = hipMemcpy(c, scenes[g]->counters, sizeof(c), hipMemcpyDeviceToHost);
            float ms = 0.f;
            if (he == hipSuccess) he = hipEventElapsedTime(&ms, ev[2 * g], ev[2 * g + 1]);
            if (he != hipSuccess) {
                rst = fail(BF_ERR_DEVICE, "bf_render_sharded_device: device %d: %s", devices[g], hipGetErrorString(he));
                break;
            }
            bf_stats p;
            fill_stats(scenes[g], c, shard_paths[g], &p);
            stats_out->n_paths += p.n_paths;
            stats_out->n_rays_closest += p.n_rays_closest;
            stats_out->n_rays_shadow += p.n_rays_shadow;
            stats_out->n_nodes_visited += p.n_nodes_visited;
            stats_out->n_tris_tested += p.n_tris_tested;
            stats_out->n_invalid += p.n_invalid;
            stats_out->n_bounces += p.n_bounces;
            stats_out->n_rays_tail += p.n_rays_tail;
            stats_out->n_rays_traced += p.n_rays_traced;
            stats_out->n_nodes_lds += p.n_nodes_lds;
            stats_out->kernel_variant = p.kernel_variant;
            // the devices run side by side: the slowest one's span (the per-kernel times need a synchronous render: 0 here)
            stats_out->kernel_ms = std::max(stats_out->kernel_ms, ms);
            if (c[bfd::CTR_GUARD] || c[bfd::CTR_SURV_GUARD]) {
                lost += c[bfd::CTR_GUARD];
                refused += c[bfd::CTR_SURV_GUARD];
                (void) hipMemset(scenes[g]->counters + bfd::CTR_GUARD, 0, 2 * sizeof(unsigned long long));
                scenes[g]->wf_fb_pending = false;
            }
        }
        drop_events();
        if (rst == BF_OK && (lost || refused)) rst = wf_guard_error(lost, refused);
    }
    return rst;
}

bf_status bf_render_sharded(bf_scene *const *scenes, uint32_t n_devices, const bf_launch *launch, float *hist_out, bf_stats *stats_out) {
    if (!scenes || !launch || !hist_out || n_devices == 0) return fail(BF_ERR_INVALID, "bf_render_sharded: null argument");
    if (launch->flags & BF_FLAG_ROLLING) return fail(BF_ERR_INVALID, "bf_render_sharded: host-buffer renders are synchronous (no BF_FLAG_ROLLING)");
    const uint64_t n = bf_launch_channels(launch);
    if (n == 0) return fail(BF_ERR_INVALID, "unknown mode");
    int prev = -1;
    (void) hipGetDevice(&prev);
    std::vector<float *> bufs(n_devices, nullptr);
    auto cleanup = [&]() {
        for (uint32_t g = 0; g < n_devices; ++g)
            if (bufs[g] && scenes[g] && hipSetDevice(scenes[g]->device) == hipSuccess) (void) hipFree(bufs[g]);
        if (prev >= 0) (void) hipSetDevice(prev);
    };
    for (uint32_t g = 0; g < n_devices; ++g) {
        if (!scenes[g]) {
            cleanup();
            return fail(BF_ERR_INVALID, "bf_render_sharded: scene %u is null", g);
        }
        hipError_t e = hipSetDevice(scenes[g]->device);
        if (e == hipSuccess) e = hipMalloc((void **) &bufs[g], n * sizeof(float));
        if (e == hipSuccess) e = hipMemset(bufs[g], 0, n * sizeof(float));
        if (e != hipSuccess) {
            cleanup();
            return fail(BF_ERR_DEVICE, "bf_render_sharded: device %d: %s", scenes[g]->device, hipGetErrorString(e));
        }
    }
    bf_status st = bf_render_sharded_device(scenes, n_devices, launch, bufs.data(), nullptr, stats_out);
    for (uint32_t g = 0; g < n_devices && st == BF_OK; ++g) {
        hipError_t e = hipSetDevice(scenes[g]->device);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) st = fail(BF_ERR_DEVICE, "bf_render_sharded: device %d: %s", scenes[g]->device, hipGetErrorString(e));
    }
    if (st == BF_OK) {
        hipError_t e = hipSetDevice(scenes[0]->device);
        if (e == hipSuccess) e = hipMemcpy(hist_out, bufs[0], n * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) st = fail(BF_ERR_DEVICE, "bf_render_sharded copy back: %s", hipGetErrorString(e));
    }
    cleanup();
    return st;
}

static bf_status render_host(const bf_scene *scene, const bf_launch *launch, const bf_batch *batch, float *hist_out,
                             bf_path_record *records_out, bf_stats *stats_out) {
    if (!scene || !launch || !hist_out) return fail(BF_ERR_INVALID, "null argument");
    const uint64_t n_renders = batch ? batch->n_renders : 1u;
    if (n_renders == 0) return fail(BF_ERR_INVALID, "bf_render_batch: n_renders is 0");
    const uint64_t nchan = (uint64_t) bf_launch_channels(launch) * n_renders, n_rec = launch->n_paths * n_renders;
    if (nchan == 0) return fail(BF_ERR_INVALID, "unknown mode");
    DeviceGuard on_device(scene->device);      // the staging buffers live where the kernels run, whatever the caller's current device
    float *d_hist = nullptr;
    bf_path_record *d_rec = nullptr;
    HIP_TRY(hipMalloc((void **) &d_hist, nchan * sizeof(float)));
    hipError_t e = hipMemset(d_hist, 0, nchan * sizeof(float));
    if (e == hipSuccess && records_out && n_rec)
        e = hipMalloc((void **) &d_rec, n_rec * sizeof(bf_path_record));
    if (e != hipSuccess) {
        (void) hipFree(d_hist);
        return fail(BF_ERR_DEVICE, "bf_render: %s", hipGetErrorString(e));
    }
    bf_stats local;
    bf_status st = render_common(scene, launch, batch, d_hist, d_rec, nullptr, stats_out ? stats_out : &local);
    if (st == BF_OK) {
        e = hipMemcpy(hist_out, d_hist, nchan * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess && d_rec)
            e = hipMemcpy(records_out, d_rec, n_rec * sizeof(bf_path_record), hipMemcpyDeviceToHost);
        if (e != hipSuccess) st = fail(BF_ERR_DEVICE, "bf_render copy back: %s", hipGetErrorString(e));
    }
    (void) hipFree(d_hist);
    if (d_rec) (void) hipFree(d_rec);
    return st;
}

bf_status bf_render(const bf_scene *scene, const bf_launch *launch, float *hist_out, bf_path_record *records_out,
                    bf_stats *stats_out) {
    return render_host(scene, launch, nullptr, hist_out, records_out, stats_out);
}

bf_status bf_render_batch(const bf_scene *scene, const bf_launch *launch, const bf_batch *batch, float *hist_out,
                          bf_path_record *records_out, bf_stats *stats_out) {
    if (!batch) return fail(BF_ERR_INVALID, "bf_render_batch: null batch");
    return render_host(scene, launch, batch, hist_out, records_out, stats_out);
}

static bf_status trace_common(const bf_scene *scene, uint64_t n, const float *rays, int any_hit, float *out_t,
                              uint32_t *out_prim, uint32_t *out_shape, float *out_uv, uint8_t *out_hit,
                              float *out_si = nullptr) {
    if (!scene || (n && !rays)) return fail(BF_ERR_INVALID, "null argument");
    if (n == 0) return BF_OK;
    DeviceGuard on_device(scene->device);
    float *d_rays = nullptr, *d_t = nullptr, *d_uv = nullptr, *d_si = nullptr;
    uint32_t *d_prim = nullptr, *d_shape = nullptr;
    uint8_t *d_hit = nullptr;
    std::vector<void *> tmp;
    auto alloc = [&](void **p, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(p, bytes);
        if (e == hipSuccess) tmp.push_back(*p);
        return e;
    };
    auto cleanup = [&]() {
        for (void *p : tmp) (void) hipFree(p);
    };
    hipError_t e = alloc((void **) &d_rays, n * 8 * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays, n * 8 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && any_hit) e = alloc((void **) &d_hit, n);
    if (e == hipSuccess && !any_hit) {
        e = alloc((void **) &d_t, n * 4);
        if (e == hipSuccess) e = alloc((void **) &d_prim, n * 4);
        if (e == hipSuccess) e = alloc((void **) &d_shape, n * 4);
        if (e == hipSuccess) e = alloc((void **) &d_uv, n * 8);
        if (e == hipSuccess && out_si) e = alloc((void **) &d_si, n * BF_SI_FLOATS * sizeof(float));
    }
    if (e == hipSuccess) e = bfk_launch_trace(&scene->d, n, d_rays, any_hit, d_t, d_prim, d_shape, d_uv, d_hit, d_si, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && any_hit && out_hit) e = hipMemcpy(out_hit, d_hit, n, hipMemcpyDeviceToHost);
    if (e == hipSuccess && !any_hit) {
        if (out_t) e = hipMemcpy(out_t, d_t, n * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && out_prim) e = hipMemcpy(out_prim, d_prim, n * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && out_shape) e = hipMemcpy(out_shape, d_shape, n * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && out_uv) e = hipMemcpy(out_uv, d_uv, n * 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess && out_si) e = hipMemcpy(out_si, d_si, n * BF_SI_FLOATS * sizeof(float), hipMemcpyDeviceToHost);
    }
    cleanup();
    if (e != hipSuccess) return fail(BF_ERR_DEVICE, "bf_trace: %s", hipGetErrorString(e));
    return BF_OK;
}

bf_status bf_trace_closest(const bf_scene *scene, uint64_t n, const float *rays, float *out_t, uint32_t *out_prim,
                           uint32_t *out_shape, float *out_uv) {
    return trace_common(scene, n, rays, 0, out_t, out_prim, out_shape, out_uv, nullptr);
}

bf_status bf_ray_intersect(const bf_scene *scene, uint64_t n, const float *rays, float *out_si, uint32_t *out_prim,
                           uint32_t *out_shape) {
    if (n && !out_si) return fail(BF_ERR_INVALID, "bf_ray_intersect: null output");
    return trace_common(scene, n, rays, 0, nullptr, out_prim, out_shape, nullptr, nullptr, out_si);
}

bf_status bf_trace_any(const bf_scene *scene, uint64_t n, const float *rays, uint8_t *out_hit) {
    return trace_common(scene, n, rays, 1, nullptr, nullptr, nullptr, nullptr, out_hit);
}

bf_status bf_eval_elementary(int op, uint64_t n, const float *x, float *y) {
    if (op < 0 || op > 6 || (n && (!x || !y))) return fail(BF_ERR_INVALID, "bf_eval_elementary: bad arguments");
    if (n == 0) return BF_OK;
    float *d_x = nullptr, *d_y = nullptr;
    hipError_t e = hipMalloc((void **) &d_x, n * 4);
    if (e == hipSuccess) e = hipMalloc((void **) &d_y, n * 4);
    if (e == hipSuccess) e = hipMemcpy(d_x, x, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = bfk_launch_elementary(op, n, d_x, d_y);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(y, d_y, n * 4, hipMemcpyDeviceToHost);
    (void) hipFree(d_x);
    (void) hipFree(d_y);
    if (e != hipSuccess) return fail(BF_ERR_DEVICE, "bf_eval_elementary: %s", hipGetErrorString(e));
    return BF_OK;
}

}  // extern "C"
