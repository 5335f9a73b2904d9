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
constexpr uint32_t kTraceRefill = 32;      // wf_trace refills idle lanes once at most this many still hold a ray (44 until round 4: -2 % at 32, profiles/r04_knob_resweep.txt)
constexpr uint32_t kTraceStragglers = 12;  // ... and postpones node steps of fewer lanes than this while leaves wait
constexpr uint32_t kTraceBlocksPerCU = 8;
constexpr uint32_t kTailSmallPool = 1u << 22;   // pools below this never fill the chip: earlier hand-over to the tail (bf_api.cpp: wf_tail_threshold)
constexpr uint32_t kTailRowJobs = 12;      // tail: up to three passes of four row-traversed rays beat one quad pass of sixteen

// Per-slot state layout.  BF_STATE_AOS = 1 (default since round 3): three 64-byte RECORDS per slot,
//   A = ray0, ray1, hit, (hit_prim, render, dop, -)      what wf_trace reads / writes for a closest-hit ray: ONE line
//   B = sa, sb, sd, se                                   the rest of the path state (wf_shade; wf_trace adds a released
//                                                        NEE contribution to sa.w / se.w)
//   C = sh0, sh1, (sh2, sh3, -, -), -                    the NEE shadow request
// Most launches GATHER sparse slots through the mask cursor (a few per cent to a third of a batch alive): with one
// array per row (BF_STATE_AOS = 0, rounds 1-2: "a wave touching one dense batch reads 1 KiB contiguous per array") a
// gathered lane touches seven cache lines for its state and six more to write it back; with records, two and two.
#ifndef BF_STATE_AOS
#define BF_STATE_AOS 1
#endif
#define BF_HD __host__ __device__ __forceinline__

struct WF {
#if BF_STATE_AOS
    // BF_STATE_AOS = 2: records A and B of a slot are the two halves of ONE 128-byte line (recB = recA + 4, stride 8)
    static constexpr size_t kAB = BF_STATE_AOS == 2 ? 8 : 4;
    float4 *recA, *recB, *recC;     // [n_slots][4] float4 each
    BF_HD float4 &ray0(uint32_t i) const { return recA[kAB * (size_t) i + 0]; }       // o.xyz, mint
    BF_HD float4 &ray1(uint32_t i) const { return recA[kAB * (size_t) i + 1]; }       // d.xyz, maxt
    BF_HD float4 &hit(uint32_t i) const { return recA[kAB * (size_t) i + 2]; }        // t, u, v, triangle slot
    BF_HD uint32_t &hit_prim(uint32_t i) const { return reinterpret_cast<uint32_t *>(recA + kAB * (size_t) i + 3)[0]; }
    BF_HD uint32_t &render(uint32_t i) const { return reinterpret_cast<uint32_t *>(recA + kAB * (size_t) i + 3)[1]; }
    BF_HD float &dop(uint32_t i) const { return reinterpret_cast<float *>(recA + kAB * (size_t) i + 3)[2]; }
    BF_HD float4 &sa(uint32_t i) const { return recB[kAB * (size_t) i + 0]; }         // throughput, eta, emission_weight, result
    BF_HD float4 &sb(uint32_t i) const { return recB[kAB * (size_t) i + 1]; }         // aux, bs_pdf, depth|flags (bits), n_rays (bits)
    BF_HD uint4 &sd(uint32_t i) const { return reinterpret_cast<uint4 *>(recB)[kAB * (size_t) i + 2]; }   // rng state lo/hi, path index lo/hi
    BF_HD float4 &se(uint32_t i) const { return recB[kAB * (size_t) i + 3]; }         // receive mode: ray.time, t_rx, lambda0, phase / Q
    BF_HD float4 &sh0(uint32_t i) const { return recC[4 * (size_t) i + 0]; }        // shadow ray o.xyz, mint
    BF_HD float4 &sh1(uint32_t i) const { return recC[4 * (size_t) i + 1]; }        // d.xyz, maxt
    BF_HD float &sh2(uint32_t i) const { return reinterpret_cast<float *>(recC + 4 * (size_t) i + 2)[0]; }   // contribution released when unoccluded
    BF_HD float &sh3(uint32_t i) const { return reinterpret_cast<float *>(recC + 4 * (size_t) i + 2)[1]; }   // BF_MODE_RECEIVE_IQ: its imaginary part
#else
    float4 *ray0_, *ray1_, *sa_, *sb_, *se_, *hit_, *sh0_, *sh1_;
    uint4 *sd_;
    uint32_t *hit_prim_, *render_;
    float *sh2_, *sh3_, *dop_;
    BF_HD float4 &ray0(uint32_t i) const { return ray0_[i]; }
    BF_HD float4 &ray1(uint32_t i) const { return ray1_[i]; }
    BF_HD float4 &hit(uint32_t i) const { return hit_[i]; }
    BF_HD uint32_t &hit_prim(uint32_t i) const { return hit_prim_[i]; }
    BF_HD uint32_t &render(uint32_t i) const { return render_[i]; }
    BF_HD float &dop(uint32_t i) const { return dop_[i]; }
    BF_HD float4 &sa(uint32_t i) const { return sa_[i]; }
    BF_HD float4 &sb(uint32_t i) const { return sb_[i]; }
    BF_HD uint4 &sd(uint32_t i) const { return sd_[i]; }
    BF_HD float4 &se(uint32_t i) const { return se_[i]; }
    BF_HD float4 &sh0(uint32_t i) const { return sh0_[i]; }
    BF_HD float4 &sh1(uint32_t i) const { return sh1_[i]; }
    BF_HD float &sh2(uint32_t i) const { return sh2_[i]; }
    BF_HD float &sh3(uint32_t i) const { return sh3_[i]; }
#endif
    uint32_t has_render;     // batched / rolling launches: render(i) = render index of the slot's current path (selects seed, histogram, mesh offset)
    uint32_t has_dop;        // BF_FLAG_DOPPLER: dop(i) = wavelength shift accumulated by the slot's path (nm)
    const float4 *offsets;   // batched launches with moving meshes: DLaunch::batch_offsets (wf_trace has no DLaunch)
    float box_slack;
    // batch masks, double buffered by bounce parity: [2][n_slots / 64]
    unsigned long long *m_alive[2];
    unsigned long long *m_trace[2];
    unsigned long long *m_shadow[2];
    // hit : the slot's pending vertex is a REAL hit (its closest-hit ray found a surface) — a packing hint, never a condition:
    //       wf_shade walks the alive slots WITHOUT the bit first (rays that left the scene, paths whose film write is due: a few
    //       instructions each) and then the ones WITH it, so that the expensive part of a vertex runs with full waves
    unsigned long long *m_hit[2];
    uint32_t hit_split;             // 1: two passes as above; 0: one walk of the alive masks (BF_SHADE_SPLIT=0)
    uint32_t chain_min;             // a chained round shades resolved REAL hits only while at least this many lanes hold one (the
                                    // others are stored and picked up — packed — by the next launch); 0: always chain
    // wf_shade with lane refill (bf_wavefront.hip, BF_SHADE_REFILL): settled lanes are written back and replaced once rf_min of a wave's
    // lanes are out of work; a pass over the real hits runs once rf_th lanes hold one, or fewer than rf_tm lanes hold cheap work
    uint32_t rf_min, rf_th, rf_tm;
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
    uint32_t *surv_cursor;          // [0] survivor batches claimed by the earlier launches of the sequence (where the next one starts
                                    // looking); [1] claims of the current evicting launch (one returning atomic per claim; folded
                                    // into [0] by the wake launch that follows: bf_wavefront.hip: surv_take)
    uint32_t wake_b0, wake_nb;      // wake launch: the batches that hold the slots due for the new render's paths: wake_nb batches from wake_b0, wrapping at n_main
    uint32_t tail_share;            // tail kernel: 1, 2 or 4 waves share every batch (16 paths per wave walk four lanes per ray from the first bounce)
    uint32_t surv_claims_max;       // claims one wave may make per launch: n_waves * max <= survivor batches, so no batch is claimed twice in a launch
};

}  // namespace bfd
