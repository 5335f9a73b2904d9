// Wavefront workspace: per-slot path state resident in HBM.
//
// The pool holds `n_slots` path slots (multiple of 64).  A slot's state stays IN
// PLACE for the whole render (no ping-pong, no queue compaction — device-wide
// queue counters serialise at ~88 returning atomics/us on MI355X, which made
// one atomic per 64-slot batch the bottleneck of the shading kernel); which
// slots need work is described by three bit masks per 64-slot batch:
//   alive  : the slot holds a live path (shade it next bounce)
//   trace  : its next closest-hit ray has to be traced
//   shadow : it queued an NEE shadow ray this bounce
// The kernels walk contiguous segments of batches and compact on the fly: a wave
// pulls the next set bits of its segment's masks into its idle lanes
// (__popcll / n-th-set-bit select), so sparse pools still run full waves.
//
// All per-slot arrays are SoA float4/uint4: a wave touching one dense batch reads
// and writes 1 KiB contiguous per array.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bfd {

constexpr uint32_t kWfMaxIter = 4096;   // ring of per-bounce live counters
constexpr uint32_t kShadeChain = 8;        // wf_shade: vertices per visit while rays resolve early (C5: 8 beats 3 by 4 %, C2-C4 indifferent; profiles/r02_chain_sweep.txt)
constexpr uint32_t kTraceRefill = 44;      // wf_trace refills idle lanes once at most this many still hold a ray
constexpr uint32_t kTraceStragglers = 12;  // ... and postpones node steps of fewer lanes than this while leaves wait
constexpr uint32_t kTraceBlocksPerCU = 8;
constexpr uint32_t kTailSmallPool = 1u << 22;   // pools below this never fill the chip: earlier hand-over to the tail (bf_api.cpp: wf_tail_threshold)
constexpr uint32_t kTailRowJobs = 12;      // tail: up to three passes of four row-traversed rays beat one quad pass of sixteen

struct WF {
    // path state [n_slots]
    float4 *ray0;       // o.xyz, mint
    float4 *ray1;       // d.xyz, maxt
    float4 *sa;         // throughput, eta, emission_weight, result
    float4 *sb;         // aux, bs_pdf, depth|flags (bits), n_rays (bits)
    uint4 *sd;          // rng state lo/hi, path index lo/hi
    float4 *se;         // receive mode only: ray.time, t_rx, lambda0, -
    float4 *hit;        // t, u, v, triangle slot
    uint32_t *hit_prim; // global primitive index of `hit` (tie rule) while a ray is in flight
    // NEE shadow ray of the slot (valid iff its shadow bit is set)
    float4 *sh0;        // o.xyz, mint
    float4 *sh1;        // d.xyz, maxt
    float *sh2;         // contribution released when unoccluded
    float *sh3;         // BF_MODE_RECEIVE_IQ: its imaginary part
    float *dop;         // BF_FLAG_DOPPLER: wavelength shift accumulated by the slot's path (nm); nullptr when the hook is off
    uint32_t *render;   // batched launches: render index of the slot's current path (selects the mesh offset)
    const float4 *offsets;   // batched launches with moving meshes: DLaunch::batch_offsets (wf_trace has no DLaunch)
    float box_slack;
    // batch masks, double buffered by bounce parity: [2][n_slots / 64]
    unsigned long long *m_alive[2];
    unsigned long long *m_trace[2];
    unsigned long long *m_shadow[2];
    uint32_t *n_live;               // [kWfMaxIter + 2] live slots after shading bounce `it`
    unsigned long long *counters;   // CTR_* (bf_device.h)
    uint32_t trace_refill, trace_stragglers;   // wf_trace scheduling thresholds (see bf_wavefront.hip)
    uint32_t iq;                               // BF_MODE_RECEIVE_IQ
    uint32_t shade_chain;                      // wf_shade: vertices a lane may shade per visit while its rays resolve early
    uint32_t row_jobs;                         // tail: waves holding at most this many rays trace them one per 16-lane row (traverse_row16)
    uint32_t n_slots;               // slots in use this render (multiple of 64): n_main + n_surv
    uint32_t capacity;              // slots allocated
    // Rolling sequences: slots [0, n_main) render the static path sequence g = slot, slot + n_main, ... (n_main = two
    // renders' worth, so a slot's next path is supplied two calls after its current one: plenty of time to finish);
    // slots [n_main, n_main + n_surv) are the SURVIVOR AREA: a path that is still alive when its slot's next path has
    // been supplied moves there (wf_shade<0> writes its state back to a free survivor slot instead of its own), so the
    // slot starts the new render's path on time and the wake launch stays one coherent generate pass.  Survivor slots
    // never regenerate.  Plain renders: n_main = n_slots, n_surv = 0.
    uint32_t n_main, n_surv;
    uint32_t *surv_cursor;          // survivor batches claimed so far (device; one returning atomic per claim)
    uint32_t wake_b0, wake_nb;      // wake launch: the batches that hold the slots due for the new render's paths: wake_nb batches from wake_b0, wrapping at n_main
    uint32_t tail_share;            // tail kernel: 1, 2 or 4 waves share every batch (16 paths per wave walk four lanes per ray from the first bounce)
    uint32_t surv_claims_max;       // claims one wave may make per launch: n_waves * max <= survivor batches, so no batch is claimed twice in a launch
};

}  // namespace bfd
