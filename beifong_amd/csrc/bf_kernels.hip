// Hand-written gfx950 kernels for beifong's transient-radar hot path: the one-kernel estimator loop, the batch
// ray queries, the mesh-translation kernel.
//
// bf_render_kernel<STATS, RESUME, SPILL> runs the whole estimator loop per lane (generate -> trace -> shade -> ...):
//   RESUME = false : the one-kernel variant of a render (BF_FLAG_MEGAKERNEL, an ablation of the wavefront pipeline
//                    of bf_wavefront.hip): lanes take path indices from wave-local pools of 256 refilled by one
//                    returning atomic, traverse with a 32-entry LDS stack per lane.
//   RESUME = true  : the wavefront pipeline's TAIL.  Once few paths survive, two launches per bounce cost more than
//                    they do; the survivors' slots are adopted through the alive masks and run to completion here.
//                    The tail is latency, not throughput — its length is the longest Russian-roulette survivor's
//                    bounce count times the time of one bounce of a nearly empty wave — so sparse waves trade lanes
//                    for serial depth: one ray per 16-lane DPP row on a sixteen-wide collapse of the same BVH
//                    (traverse_row16: gangs of up to four rows per ray), four lanes per ray on the four-wide tree
//                    (traverse_quad) above that, and idle lanes take over the shadow rays of busy ones.
// Path-length returns are binned into an LDS-privatised histogram flushed with one global atomic per non-empty bin
// per workgroup.  No MFMA: the path is pointer chasing and scalar shading.
//
// Reference semantics (file:line) are cited at each function; the oracle
// (oracle/bf_oracle.cpp) restates the same functions independently on the CPU.
#include <algorithm>
#include <cstring>

#include "bf_path_logic.h"

namespace bfd {

// RESUME = false: the whole render in one launch (ablation of the wavefront
// pipeline, BF_FLAG_MEGAKERNEL).
// RESUME = true : the wavefront pipeline's TAIL kernel — once the path supply is
// exhausted and only a few thousand long paths remain, every lane adopts one
// slot of the wavefront queue (state + the closest hit already traced for it)
// and runs that path to completion here, instead of paying two launches per
// bounce for a nearly empty chip.
// Register budget of the tail (waves per SIMD): 3 (168 VGPRs, some scratch) leaves room for the next renders' kernels when
// renders are pipelined over streams; 2 (224 VGPRs, no scratch) is ~10 % faster for the tail itself.  The launcher picks 2
// for small pools (a 2^20-path render never fills the chip: nothing to leave room for) and 3 for large ones.
#ifndef BF_TAIL_WAVES
#define BF_TAIL_WAVES 3
#endif
#ifdef BF_TAIL_PROF
// developer build (make prof): per-wave cycle breakdown of the tail kernel, read back by bfdbg_tail_profile
__device__ unsigned long long g_tail_prof[8192 * 24];
#define BF_PROF_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define BF_PROF_STAMP(var)
#endif
template <bool STATS, bool RESUME, bool SPILL, int TW = BF_TAIL_WAVES, int VX = 0>
__global__ __launch_bounds__(kBlock, RESUME ? TW : 3) void bf_render_kernel(DScene sc_arg, DLaunch lp, float *__restrict__ g_hist,
                                                           bf_path_record *__restrict__ records,
                                                           unsigned long long *__restrict__ counters, WF wf, uint32_t wf_it) {
    constexpr int kRX = 2 | VX;       // mode class decided at run time; VX: kWide (the filtered put) or kLean (bf_device.h: kernel variant word)
    extern __shared__ __align__(16) unsigned char s_raw[];
    int *s_stack = reinterpret_cast<int *>(s_raw);                         // [kStackDepth][kBlock]
    float *s_hist = reinterpret_cast<float *>(s_raw + sizeof(int) * kStackDepth * kBlock);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const bool lds_hist = lp.lds_hist != 0;
    DScene sc = sc_arg;
    if (VX & kMulti) sc.tab_cache = 0u;
    load_tables_lds(sc, (uint32_t) (sizeof(int) * kStackDepth * kBlock) + ((4u * lp.lds_floats + 15u) & ~15u), (uint32_t) tid);
    if (lp.lds_floats || sc.tab_on) {
        for (uint32_t i = tid; i < lp.lds_floats; i += kBlock) s_hist[i] = 0.f;
        __syncthreads();
    }
    int *stack = s_stack + tid;
    const bool receive = lp.mode == BF_MODE_RECEIVE_RAW;

    bool alive = false, done = false;
    bool need_closest = false;     // s holds a ray whose hit is not known yet
    bool film = false;             // the path ended at the last shaded vertex; binned after its shadow ray
    ShadowReq sh;
    sh.want = false;
    Hit hit;
    hit.t = BF_INF;
    hit.u = hit.v = 0.f;
    hit.prim = 0;
    hit.slot = 0;
    PathState s;
    s.flags = 0;
    s.render = 0u;                     // lanes without a path still index the batch tables (path_shift)
    s.dlambda = 0.f;
    s.rmint = 0.f;
    s.rmaxt = 0.f;
    FilmAcc acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0u, 0u};
    uint32_t c_closest = 0, c_shadow = 0, c_nodes = 0, c_wnodes = 0, c_tris = 0, c_bounces = 0;
    // wave-local pool of path indices (uniform across the wave)
    uint64_t pool_next = 0, pool_end = 0;

    // RESUME: this wave's segment of the pool's alive mask; lanes adopt live slots
    // from it whenever they are free (same on-the-fly compaction as wf_shade)
    uint32_t resume_slots = 0;
    bool regen_ok = true;          // RESUME: the adopted slot is a main slot (survivor-area slots of a rolling sequence never regenerate)
    MaskCursor rcur;
    rcur.masks = nullptr;
    rcur.masks2 = nullptr;
    rcur.sel = 0u;
    rcur.b = rcur.b_end = rcur.base = rcur.k = 0;
    rcur.stride = 1u;
    rcur.rot = rcur.inv = rcur.pb = 0u;
    rcur.mod = 0xffffffffu;
    rcur.sub = ~0ull;
    rcur.m = rcur.w = rcur.nz = 0ull;
    if (RESUME) {
        resume_slots = wf.n_main;
        const uint32_t n_batches = wf.n_slots >> 6;
        const uint32_t n_waves = gridDim.x * (kBlock / 64), wave_id = blockIdx.x * (kBlock / 64) + (tid >> 6);
        cursor_init(rcur, wf.m_alive[wf_it & 1], wave_id, n_waves, n_batches, lane, max(1u, wf.tail_share));
    }

#ifdef BF_TAIL_PROF
    unsigned long long pf_iters = 0, pf_regen = 0, pf_trav = 0, pf_film = 0, pf_shade = 0, pf_quad = 0, pf_rowpass = 0;
    unsigned long long pf_dense_iters = 0, pf_dense_trav = 0, pf_dense_shade = 0, pf_dense_rays = 0;
    RowProf pf_row = {0, 0, 0, 0};
    ShadeProf pf_sp = {0, 0, 0, 0};
    const unsigned long long pf_begin = __builtin_amdgcn_s_memtime();
#endif
    while (true) {
        BF_PROF_STAMP(pf_t0);
        // ---- 1. path regeneration: dead lanes pull new path indices -------
        unsigned long long need = __ballot(!alive && !done);
        if (need && RESUME) {
            uint32_t slot = 0;
            const uint32_t rank = __popcll(need & ((1ull << lane) - 1ull));
            const uint32_t got = cursor_take(rcur, (uint32_t) __popcll(need), !alive && !done, rank, slot, lane);
            if (!alive && !done) {
                if (rank < got) {
                    load_state(wf, slot, receive, s);
                    regen_ok = slot < wf.n_main;
                    alive = true;
                    need_closest = false;          // traced (and counted) by wf_trace already
                    film = false;
                    if (!(s.flags & kFlagTermPending)) {
                        float4 hq = wf.hit(slot);
                        hit.t = hq.x;
                        hit.u = hq.y;
                        hit.v = hq.z;
                        hit.slot = __float_as_int(hq.w);
                        hit.prim = 0;
                    }
                } else {
                    done = true;       // the segment has no live slot left for this lane
                }
            }
        } else if (need) {
            uint32_t n_need = __popcll(need);
            uint32_t rank = __popcll(need & ((1ull << lane) - 1ull));
            uint64_t path_i = 0;
            if (pool_end - pool_next < n_need) {
                // refill: one returning atomic per wave for 4 x 64 paths; the old
                // remainder is handed out first, so no index is ever skipped
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(&counters[CTR_NEXT_PATH], 256ull);
                base = __shfl(base, 0);
                uint64_t left = pool_end - pool_next;
                path_i = rank < left ? pool_next + rank : base + (rank - left);
                pool_next = base + (n_need - left);
                pool_end = base + 256ull;
            } else {
                path_i = pool_next + rank;
                pool_next += n_need;
            }
            if (!alive && !done) {
                if (path_i >= lp.n_paths) {
                    done = true;
                } else {
                    generate_path<kRX>(sc, lp, path_i, s);
                    alive = true;
                    need_closest = true;
                    film = false;
                }
            }
        }
        if (__ballot(alive) == 0ull) break;
        BF_PROF_STAMP(pf_t1);

        // ---- 2. traversal phase ------------------------------------------------------
        // Every lane may hold a closest-hit ray (new path, or the continuation ray of the vertex it
        // shaded last iteration) and a shadow ray (NEE of that same vertex).  The two are independent,
        // so lanes that have nothing to trace (their path is over, or the wave's paths have run out —
        // the normal state of the latency-bound tail) take over shadow rays of busy lanes: the wave
        // walks the BVH once per bounce instead of twice.
        const bool term_pending = alive && (s.flags & kFlagTermPending) != 0;
        const bool trace_closest = alive && need_closest && !term_pending;
        const unsigned long long want_mask = __ballot(sh.want);
        bool occluded = false;
        const unsigned long long closest_mask = __ballot(trace_closest);
        const uint32_t n_cl = (uint32_t) __popcll(closest_mask), n_sh = (uint32_t) __popcll(want_mask);
#ifdef BF_TAIL_PROF
        if (RESUME && n_cl + n_sh != 0u && n_cl + n_sh <= 16u) ++pf_quad;
        const bool pf_is_dense = n_cl + n_sh > 16u;
        pf_dense_rays += pf_is_dense ? n_cl + n_sh : 0u;
#endif
        if (RESUME && n_cl + n_sh != 0u && n_cl + n_sh <= wf.row_jobs && sc.wnodes != nullptr) {
            // ---- deep tail: ONE ray per 16-lane row on the sixteen-wide tree (traverse_row16), four rays per pass ------
            // job k < n_cl: closest-hit ray of the k-th lane of closest_mask; job n_cl + k: shadow ray of the k-th lane of
            // want_mask; row r of pass p serves job 4 p + r
            const uint32_t n_jobs = n_cl + n_sh;
            const unsigned long long below = (1ull << lane) - 1ull;
            const uint32_t my_cl_job = (uint32_t) __popcll(closest_mask & below);
            const uint32_t my_sh_job = n_cl + (uint32_t) __popcll(want_mask & below);
            // a lone ray gets all four rows, two rays two rows each (gangs pop several stack entries per step)
            const uint32_t rlog = min(n_jobs == 1u ? 2u : (n_jobs == 2u ? 1u : 0u), sc.wrows_log);
            const uint32_t per_pass = 4u >> rlog, gshift = 4u + rlog;
            for (uint32_t base = 0; base < n_jobs; base += per_pass) {
                const uint32_t job = base + ((uint32_t) lane >> gshift);
                const bool job_active = job < n_jobs;
                const bool job_any = job >= n_cl;
                int src = lane;
                if (job_active) src = (int) (job_any ? nth_set_bit(want_mask, job - n_cl) : nth_set_bit(closest_mask, job));
                const V3 co = mk(__shfl(s.ro.x, src), __shfl(s.ro.y, src), __shfl(s.ro.z, src));
                const V3 cd = mk(__shfl(s.rd.x, src), __shfl(s.rd.y, src), __shfl(s.rd.z, src));
                const float cmint = __shfl(s.rmint, src), cmaxt = __shfl(s.rmaxt, src);
                const V3 so = mk(__shfl(sh.o.x, src), __shfl(sh.o.y, src), __shfl(sh.o.z, src));
                const V3 sd = mk(__shfl(sh.d.x, src), __shfl(sh.d.y, src), __shfl(sh.d.z, src));
                const float smint = __shfl(sh.mint, src), smaxt = __shfl(sh.maxt, src);
                const uint32_t jrender = (uint32_t) __shfl((int) s.render, src);
                const Shift jshift = path_shift(lp, jrender);
                const DScene scj = path_scene<kRX>(sc, lp, jrender);      // (kMulti: the rectangles of the job's render)
                Hit qbest;
                bool qfound;
                traverse_row16<STATS>(scj, rlog, job_active, job_any, job_any ? so : co, job_any ? sd : cd, job_any ? smint : cmint,
                                      job_any ? smaxt : cmaxt, s_stack, qbest, qfound, c_wnodes, c_tris, jshift
#ifdef BF_TAIL_PROF
                                      , pf_row
#endif
                );
#ifdef BF_TAIL_PROF
                ++pf_rowpass;
#endif
                // deliver: closest hits to their lanes, occlusion verdicts to the requesters
                const int cl_lane = (int) (((my_cl_job - base) & (per_pass - 1u)) << gshift);
                const float ht = __shfl(qbest.t, cl_lane), hu = __shfl(qbest.u, cl_lane), hv = __shfl(qbest.v, cl_lane);
                const int hs = __shfl(qbest.slot, cl_lane);
                const uint32_t hp = (uint32_t) __shfl((int) qbest.prim, cl_lane);
                if (trace_closest && my_cl_job >= base && my_cl_job < base + per_pass) {
                    hit.t = ht;
                    hit.u = hu;
                    hit.v = hv;
                    hit.slot = hs;
                    hit.prim = hp;
                    ++c_closest;
                }
                const unsigned long long occl_mask = __ballot(job_active && job_any && qfound);
                if (sh.want && my_sh_job >= base && my_sh_job < base + per_pass) {
                    occluded = (occl_mask >> (((my_sh_job - base) & (per_pass - 1u)) << gshift)) & 1ull;
                    ++c_shadow;
                    // an occluded sample still contributes mis * throughput * bsdf * 0 (scene.cpp:220-224): c * 0
                    s.result += occluded ? sh.c * 0.f : sh.c;
                    if (lp.iq) s.phase += occluded ? sh.c_im * 0.f : sh.c_im;
                    sh.want = false;
                }
            }
        } else if (RESUME && n_cl + n_sh != 0u && n_cl + n_sh <= 16u) {
            // ---- sparse wave (the deep tail): four lanes per ray (traverse_quad) -----------------------
            // job k < n_cl: closest-hit ray of the k-th lane of closest_mask; job n_cl + k: shadow ray of the k-th
            // lane of want_mask; quad k = lanes 4k .. 4k + 3 serves job k
            const uint32_t job = (uint32_t) lane >> 2;
            const bool job_active = job < n_cl + n_sh;
            const bool job_any = job >= n_cl;
            int src = lane;
            if (job_active) src = (int) (job_any ? nth_set_bit(want_mask, job - n_cl) : nth_set_bit(closest_mask, job));
            const V3 co = mk(__shfl(s.ro.x, src), __shfl(s.ro.y, src), __shfl(s.ro.z, src));
            const V3 cd = mk(__shfl(s.rd.x, src), __shfl(s.rd.y, src), __shfl(s.rd.z, src));
            const float cmint = __shfl(s.rmint, src), cmaxt = __shfl(s.rmaxt, src);
            const V3 so = mk(__shfl(sh.o.x, src), __shfl(sh.o.y, src), __shfl(sh.o.z, src));
            const V3 sd = mk(__shfl(sh.d.x, src), __shfl(sh.d.y, src), __shfl(sh.d.z, src));
            const float smint = __shfl(sh.mint, src), smaxt = __shfl(sh.maxt, src);
            const uint32_t jrender = (uint32_t) __shfl((int) s.render, src);
            const Shift jshift = path_shift(lp, jrender);
            const DScene scj = path_scene<kRX>(sc, lp, jrender);
            Hit qbest;
            bool qfound;
            traverse_quad<STATS, SPILL>(scj, job_active, job_any, job_any ? so : co, job_any ? sd : cd, job_any ? smint : cmint,
                                        job_any ? smaxt : cmaxt, s_stack + (tid & ~3), qbest, qfound, c_nodes, c_tris, jshift);
            const unsigned long long below = (1ull << lane) - 1ull;
            // deliver: closest hits to their lanes, occlusion verdicts to the requesters
            const int my_cl_quad = (int) __popcll(closest_mask & below) * 4;
            const float ht = __shfl(qbest.t, my_cl_quad), hu = __shfl(qbest.u, my_cl_quad), hv = __shfl(qbest.v, my_cl_quad);
            const int hs = __shfl(qbest.slot, my_cl_quad);
            const uint32_t hp = (uint32_t) __shfl((int) qbest.prim, my_cl_quad);
            if (trace_closest) {
                hit.t = ht;
                hit.u = hu;
                hit.v = hv;
                hit.slot = hs;
                hit.prim = hp;
                ++c_closest;
            }
            const unsigned long long occl_mask = __ballot(job_active && job_any && qfound);
            if (sh.want) {
                const uint32_t leader = (n_cl + (uint32_t) __popcll(want_mask & below)) * 4u;
                occluded = (occl_mask >> leader) & 1ull;
                ++c_shadow;
                // an occluded sample still contributes mis * throughput * bsdf * 0 (scene.cpp:220-224): c * 0
                s.result += occluded ? sh.c * 0.f : sh.c;
                if (lp.iq) s.phase += occluded ? sh.c_im * 0.f : sh.c_im;
                sh.want = false;
            }
        } else if (want_mask) {
            const unsigned long long free_mask = __ballot(!trace_closest && !sh.want);
            const uint32_t n_del = min((uint32_t) __popcll(want_mask), (uint32_t) __popcll(free_mask));
            const unsigned long long below = (1ull << lane) - 1ull;
            const uint32_t want_rank = (uint32_t) __popcll(want_mask & below), free_rank = (uint32_t) __popcll(free_mask & below);
            const bool delegated = sh.want && want_rank < n_del;
            const bool helper = !trace_closest && !sh.want && free_rank < n_del;
            // helper f serves the f-th requester; requester r is served by the r-th free lane
            const int src = helper ? (int) nth_set_bit(want_mask, free_rank) : lane;
            V3 ho = mk(__shfl(sh.o.x, src), __shfl(sh.o.y, src), __shfl(sh.o.z, src));
            V3 hd = mk(__shfl(sh.d.x, src), __shfl(sh.d.y, src), __shfl(sh.d.z, src));
            const float hmint = __shfl(sh.mint, src), hmaxt = __shfl(sh.maxt, src);
            const uint32_t hrender = (uint32_t) __shfl((int) s.render, src);
            const Shift own_shift = path_shift(lp, s.render), helper_shift = path_shift(lp, hrender);
            const DScene sc_own = path_scene<kRX>(sc, lp, s.render), sc_first = path_scene<kRX>(sc, lp, trace_closest ? s.render : hrender);
            // first walk: closest-hit rays, delegated shadow rays (on their helpers), and the own shadow
            // ray of a lane that has no closest-hit ray to trace
            const bool own_first = sh.want && !delegated && !trace_closest;
            const bool any1 = helper || own_first;
            bool r1 = false;
            if (trace_closest || any1) {
                V3 o1 = trace_closest ? s.ro : ho, d1 = trace_closest ? s.rd : hd;
                float mint1 = trace_closest ? s.rmint : hmint, maxt1 = trace_closest ? s.rmaxt : hmaxt;
                Hit h1;
                r1 = traverse_dyn<STATS, SPILL>(sc_first, any1, o1, d1, mint1, maxt1, stack, h1, c_nodes, c_tris,
                                                trace_closest ? own_shift : helper_shift);
                if (trace_closest) {
                    hit = h1;
                    ++c_closest;
                } else {
                    ++c_shadow;
                }
            }
            const unsigned long long helper_occluded = __ballot(helper && r1);
            if (delegated) occluded = (helper_occluded >> nth_set_bit(free_mask, want_rank)) & 1ull;
            if (own_first) occluded = r1;
            // second walk: lanes that had both rays and found no helper
            const bool own_second = sh.want && !delegated && trace_closest;
            if (__ballot(own_second)) {
                if (own_second) {
                    Hit tmp;
                    occluded = traverse<true, STATS, SPILL>(sc_own, sh.o, sh.d, sh.mint, sh.maxt, stack, tmp, c_nodes, c_tris, own_shift);
                    ++c_shadow;
                }
            }
            if (sh.want) {                                  // occluded: c * 0 (scene.cpp:220-224)
                s.result += occluded ? sh.c * 0.f : sh.c;
                if (lp.iq) s.phase += occluded ? sh.c_im * 0.f : sh.c_im;
            }
            sh.want = false;
        } else if (__ballot(trace_closest)) {
            if (trace_closest) {
                traverse<false, STATS, SPILL>(path_scene<kRX>(sc, lp, s.render), s.ro, s.rd, s.rmint, s.rmaxt, stack, hit, c_nodes, c_tris,
                                              path_shift(lp, s.render));
                ++c_closest;
            }
        }
        if (trace_closest) need_closest = false;
        BF_PROF_STAMP(pf_t2);

        // ---- 3. film: paths that ended at the vertex shaded last iteration (their last shadow ray
        //         has resolved by now) ---------------------------------------------------------------
        if (alive && (film || term_pending)) {
            film_put<kRX>(sc, lp, s, s_hist, g_hist, lds_hist, acc, records);
            alive = false;
            film = false;
            if (RESUME) {
                // the slot's static path sequence: i, i + n_slots, i + 2 n_slots, ...
                uint64_t next_path = s.path_i + resume_slots;
                if (regen_ok && next_path < lp.n_paths) {
                    generate_path<kRX>(sc, lp, next_path, s);
                    alive = true;
                    need_closest = true;
                }
            }
        }

        BF_PROF_STAMP(pf_t3);
        // ---- 4. vertex logic for the lanes that hold a fresh hit ---------------------------------
        if (alive && !need_closest) {
            if (!shade_vertex<kRX>(sc, lp, s, hit, sh, c_bounces
#ifdef BF_TAIL_PROF
                              , &pf_sp
#endif
                              )) {
                film = true;                                   // ended at the head of the iteration: no new rays
            } else if (s.flags & kFlagTermPending) {
                film = true;                                   // ended by the BSDF sample: its NEE ray is still to trace
            } else {
                need_closest = true;
            }
        }
#ifdef BF_TAIL_PROF
        {
            const unsigned long long pf_t4 = __builtin_amdgcn_s_memtime();
            ++pf_iters;
            if (pf_is_dense) {
                ++pf_dense_iters;
                pf_dense_trav += pf_t2 - pf_t1;
                pf_dense_shade += pf_t4 - pf_t3;
            }
            pf_regen += pf_t1 - pf_t0;
            pf_trav += pf_t2 - pf_t1;
            pf_film += pf_t3 - pf_t2;
            pf_shade += pf_t4 - pf_t3;
        }
#endif
    }
#ifdef BF_TAIL_PROF
    {   // the lane that shaded most vertices (the wave's longest path) speaks for the shading breakdown
        unsigned long long key = pf_sp.si + pf_sp.head + pf_sp.nee + pf_sp.bsdf, best = key;
        for (int off = 32; off > 0; off >>= 1) {
            unsigned long long o = wave_read_u64(best, (lane + off) & 63);
            best = o > best ? o : best;
        }
        best = wave_read_u64(best, 0);
        // after the rotation-reduction lane 0 holds the max over all lanes only if every lane took part: recompute plainly
        unsigned long long m = key;
        for (int off = 1; off < 64; off <<= 1) {
            unsigned long long o = wave_read_u64(m, lane ^ off);
            m = o > m ? o : m;
        }
        const unsigned long long who = __ballot(key == m);
        const int srcl = __ffsll((unsigned long long) who) - 1;
        pf_sp.si = wave_read_u64(pf_sp.si, srcl);
        pf_sp.head = wave_read_u64(pf_sp.head, srcl);
        pf_sp.nee = wave_read_u64(pf_sp.nee, srcl);
        pf_sp.bsdf = wave_read_u64(pf_sp.bsdf, srcl);
        (void) best;
    }
    if (RESUME && lane == 0) {
        const uint32_t w = blockIdx.x * (kBlock / 64) + (tid >> 6);
        if (w < 8192u) {
            unsigned long long *q = g_tail_prof + 24u * w;
            q[16] = pf_dense_iters;
            q[17] = pf_dense_trav;
            q[18] = pf_dense_shade;
            q[19] = pf_dense_rays;
            q[8] = pf_rowpass;
            q[9] = pf_row.steps;
            q[10] = pf_row.rect;
            q[11] = pf_row.mem;
            q[12] = pf_row.cmp;
            q[13] = pf_sp.si;
            q[14] = pf_sp.head;
            q[15] = pf_sp.nee;
            q[7] = pf_sp.bsdf;
            q[0] = pf_iters;
            q[1] = pf_regen;
            q[2] = pf_trav;
            q[3] = pf_film;
            q[4] = pf_shade;
            q[5] = pf_quad;
            q[6] = __builtin_amdgcn_s_memtime() - pf_begin;
        }
    }
#endif

    film_flush<kRX>(lp, acc, s_hist, g_hist, lds_hist, tid);
    // statistics: wave-reduce then one atomic per counter per wave
    unsigned long long v_closest = c_closest, v_shadow = c_shadow, v_nodes = c_nodes, v_tris = c_tris,
                       v_invalid = acc.invalid, v_bounces = c_bounces, v_wnodes = c_wnodes, v_film = acc.n_put;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v_film += __shfl_down(v_film, off);
        v_wnodes += __shfl_down(v_wnodes, off);
        v_closest += __shfl_down(v_closest, off);
        v_shadow += __shfl_down(v_shadow, off);
        v_nodes += __shfl_down(v_nodes, off);
        v_tris += __shfl_down(v_tris, off);
        v_invalid += __shfl_down(v_invalid, off);
        v_bounces += __shfl_down(v_bounces, off);
    }
    if (lane == 0 && lp.count) {
        atomicAdd(&counters[CTR_CLOSEST], v_closest);
        atomicAdd(&counters[CTR_SHADOW], v_shadow);
        if (RESUME) atomicAdd(&counters[CTR_TAIL_RAYS], v_closest + v_shadow);
        if (STATS) {
            atomicAdd(&counters[CTR_NODES], v_nodes + v_wnodes);
            atomicAdd(&counters[CTR_TRIS], v_tris);
            if (RESUME) {
                atomicAdd(&counters[CTR_TAIL_NODES], v_nodes);
                atomicAdd(&counters[CTR_TAIL_WNODES], v_wnodes);
                atomicAdd(&counters[CTR_TAIL_TRIS], v_tris);
            }
        }
        if (RESUME) atomicAdd(&counters[CTR_TAIL_BOUNCES], v_bounces);
        atomicAdd(&counters[CTR_INVALID], v_invalid);
        atomicAdd(&counters[CTR_BOUNCES], v_bounces);
        if (v_film) atomicAdd(&counters[CTR_FILM], v_film);
    }
}

// Scene::ray_intersect / ray_test over a batch of rays (tests, tools)
template <bool SPILL>
__global__ __launch_bounds__(kBlock) void bf_trace_kernel(DScene sc, uint64_t n, const float *__restrict__ rays,
                                                          int any_hit, float *__restrict__ out_t,
                                                          uint32_t *__restrict__ out_prim, uint32_t *__restrict__ out_shape,
                                                          float *__restrict__ out_uv, uint8_t *__restrict__ out_hit,
                                                          float *__restrict__ out_si) {
    __shared__ int s_stack[kStackDepth * kBlock];
    int *stack = s_stack + threadIdx.x;
    for (uint64_t i = (uint64_t) blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t) gridDim.x * kBlock) {
    const float *r = rays + 8 * i;
    V3 o = mk(r[0], r[1], r[2]), d = mk(r[4], r[5], r[6]);
    float mint = r[3], maxt = r[7];
    Hit h;
    uint32_t a = 0, b = 0;
    if (any_hit) {
        out_hit[i] = traverse<true, false, SPILL>(sc, o, d, mint, maxt, stack, h, a, b) ? 1 : 0;
    } else {
        bool valid = traverse<false, false, SPILL>(sc, o, d, mint, maxt, stack, h, a, b);
        if (out_t) out_t[i] = h.t;
        if (out_prim) out_prim[i] = valid ? h.prim : 0xffffffffu;
        if (out_shape) {
            uint32_t s = 0xffffffffu;
            if (valid) s = h.slot < 0 ? c_rects(sc)[-h.slot - 1].shape : __float_as_uint(sc.tris[kTriStride * (size_t) h.slot + 1].w);
            out_shape[i] = s;
        }
        if (out_uv) {
            out_uv[2 * i] = h.u;
            out_uv[2 * i + 1] = h.v;
        }
        if (out_si) {
            // SurfaceInteraction record of bf_ray_intersect (BF_SI_FLOATS); misses report t = +inf and zeros
            float *q = out_si + (size_t) BF_SI_FLOATS * i;
            for (int k = 0; k < BF_SI_FLOATS; ++k) q[k] = 0.f;
            q[0] = h.t;
            if (valid) {
                SI si;
                SIGeom g;
                make_si<true>(sc, o, d, h, si, &g);
                const V3 v[8] = {si.p, g.n, si.sh.n, si.sh.s, si.sh.t, si.wi, g.dp_du, g.dp_dv};
                for (int k = 0; k < 6; ++k) {
                    q[1 + 3 * k] = v[k].x;
                    q[2 + 3 * k] = v[k].y;
                    q[3 + 3 * k] = v[k].z;
                }
                q[19] = h.u;
                q[20] = h.v;
                for (int k = 6; k < 8; ++k) {
                    q[3 + 3 * k] = v[k].x;
                    q[4 + 3 * k] = v[k].y;
                    q[5 + 3 * k] = v[k].z;
                }
            }
        }
    }
    }
}

}  // namespace bfd

// host-callable launchers (used by bf_api.cpp, which is plain C++)
extern "C" hipError_t bfk_launch_render(const bfd::DScene *sc, const bfd::DLaunch *lp, float *g_hist, bf_path_record *records,
                                        unsigned long long *counters, int stats, unsigned grid, size_t lds_bytes,
                                        hipStream_t stream) {
    bfd::WF none;
    memset(&none, 0, sizeof(none));
    const bool spill = sc->stack_need > (uint32_t) bfd::kStackDepth;
#define BF_RENDER_LAUNCH(S, R, P, WF_, IT_)                                                                                              \
    if (lp->wide)                                                                                                                        \
        hipLaunchKernelGGL((bfd::bf_render_kernel<S, R, P, BF_TAIL_WAVES, bfd::kWide>), dim3(grid), dim3(bfd::kBlock), lds_bytes, stream, *sc, \
                           *lp, g_hist, records, counters, WF_, IT_);                                                                    \
    else                                                                                                                                 \
        hipLaunchKernelGGL((bfd::bf_render_kernel<S, R, P>), dim3(grid), dim3(bfd::kBlock), lds_bytes, stream, *sc, *lp, g_hist,         \
                           records, counters, WF_, IT_)
    if (stats) {
        if (spill) { BF_RENDER_LAUNCH(true, false, true, none, 0u); }
        else { BF_RENDER_LAUNCH(true, false, false, none, 0u); }
    } else {
        if (spill) { BF_RENDER_LAUNCH(false, false, true, none, 0u); }
        else { BF_RENDER_LAUNCH(false, false, false, none, 0u); }
    }
    return hipGetLastError();
}

// wavefront tail: finish the `n_slots` paths of queue `it` in one launch
//   tail_waves : 3 waves per SIMD (168 VGPRs, some scratch) unless 2 asks for the scratch-free build: alone on the GPU a
//                small pool's tail is 2-3 % faster with 2, but with several renders in flight 3 leave room for the
//                neighbours (C3 0.74 -> 0.69 ms per render, C4 shard 0.96 -> 0.88; profiles/r02_retune_hw_queues.txt)
//   spread     : spread the survivors thinly while the chip has room: a wave that starts with ~16 paths instead of 64
//                runs them four lanes per ray (traverse_quad) from its first bounce and waits for the longest of 16
//   block_cap  : at most this many workgroups (0: no cap)
extern "C" hipError_t bfk_launch_tail(const bfd::DScene *sc, const bfd::DLaunch *lp, const bfd::WF *wf, uint32_t it,
                                      uint32_t n_slots, float *g_hist, bf_path_record *records, int stats, size_t lds_bytes,
                                      hipStream_t stream, int tail_waves, unsigned spread, unsigned block_cap) {
    // one lane per live slot (gathered from the alive masks), at most one thread per pool slot
    // (any grid finishes the job: waves loop over their segment of the alive masks; the cap keeps the
    // launch within the scene's traversal-spill columns)
    unsigned grid = (std::min(n_slots, wf->n_slots) + bfd::kBlock - 1) / bfd::kBlock;
    {
        const unsigned resident = (sc->spill_stride / bfd::kBlock) * 3u / bfd::kTraceBlocksPerCU;
        if (spread > 1 && grid < resident) grid = std::min(grid * spread, resident);
    }
    grid = std::min(grid, sc->spill_stride / bfd::kBlock);
    if (block_cap) grid = std::min(grid, block_cap);
    if (grid == 0) return hipSuccess;
    const bool spill = sc->stack_need > (uint32_t) bfd::kStackDepth;
    unsigned long long *counters = wf->counters;
    const bool two = tail_waves == 2;
#define BF_TAIL_LAUNCH(S, P)                                                                                                           \
    if (lp->multi)                                                                                                                     \
        hipLaunchKernelGGL((bfd::bf_render_kernel<S, true, P, 3, bfd::kMulti>), dim3(grid), dim3(bfd::kBlock), lds_bytes, stream, *sc, \
                           *lp, g_hist, records, counters, *wf, it);                                                                  \
    else if (lp->wide)                                                                                                                      \
        hipLaunchKernelGGL((bfd::bf_render_kernel<S, true, P, 3, bfd::kWide>), dim3(grid), dim3(bfd::kBlock), lds_bytes, stream, *sc,  \
                           *lp, g_hist, records, counters, *wf, it);                                                                  \
    else if (lp->lean && !two)                                                                                                         \
        hipLaunchKernelGGL((bfd::bf_render_kernel<S, true, P, 3, bfd::kLean>), dim3(grid), dim3(bfd::kBlock), lds_bytes, stream, *sc,  \
                           *lp, g_hist, records, counters, *wf, it);                                                                  \
    else if (two)                                                                                                                      \
        hipLaunchKernelGGL((bfd::bf_render_kernel<S, true, P, 2>), dim3(grid), dim3(bfd::kBlock), lds_bytes, stream, *sc, *lp, g_hist, \
                           records, counters, *wf, it);                                                                               \
    else                                                                                                                               \
        hipLaunchKernelGGL((bfd::bf_render_kernel<S, true, P, 3>), dim3(grid), dim3(bfd::kBlock), lds_bytes, stream, *sc, *lp, g_hist, \
                           records, counters, *wf, it)
    if (stats) {
        if (spill) { BF_TAIL_LAUNCH(true, true); }
        else { BF_TAIL_LAUNCH(true, false); }
    } else {
        if (spill) { BF_TAIL_LAUNCH(false, true); }
        else { BF_TAIL_LAUNCH(false, false); }
    }
#undef BF_TAIL_LAUNCH
    return hipGetLastError();
}

namespace bfd {
// Developer probe (tools/fetch_probe.py, profiles/README.md): what rocprofv3's FETCH_SIZE reports for the access patterns of
// this engine, on a table of known size that is read exactly once per launch.
//   MODE 0: streaming — thread g loads row g (16 B per lane, coalesced): every 128-byte line fully used by one wave
//   MODE 1: scattered rows — thread g loads row perm(g) (a bijection: an odd multiplier modulo the row count): every row
//           once, the eight rows of a line by eight different waves at different times (wf_trace's node / triangle fetches)
//   MODE 2: one 16-byte row per 128-byte line — thread g loads row 8 perm(g): 1/8 of the table's lines, each touched once
template <int MODE> __global__ void bf_gather_probe(const float4 *__restrict__ table, uint32_t n_rows, float4 *__restrict__ out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = MODE == 2 ? n_rows / 8u : n_rows;
    if (g >= n) return;
    uint32_t row = g;
    if (MODE >= 1) row = (g * 2654435761u) & (n - 1u);      // n is a power of two, the multiplier odd: a permutation
    if (MODE == 2) row *= 8u;
    const float4 v = table[row];
    if (v.x == 12345.678f) out[g & 1023u] = v;      // (never true: keeps the load)
}
}  // namespace bfd
extern "C" hipError_t bfk_gather_probe(int mode, const float4 *table, uint32_t n_rows, float4 *out, hipStream_t stream) {
    const uint32_t n = mode == 2 ? n_rows / 8u : n_rows;
    const dim3 grid((n + 255u) / 256u), block(256);
    if (mode == 0) hipLaunchKernelGGL(bfd::bf_gather_probe<0>, grid, block, 0, stream, table, n_rows, out);
    else if (mode == 1) hipLaunchKernelGGL(bfd::bf_gather_probe<1>, grid, block, 0, stream, table, n_rows, out);
    else hipLaunchKernelGGL(bfd::bf_gather_probe<2>, grid, block, 0, stream, table, n_rows, out);
    return hipGetLastError();
}

namespace bfd {
// one descriptor of a rolling sequence's ring (kernel arguments are captured at launch: no staging buffer to keep alive)
__global__ void bf_roll_set_kernel(DRoll *ring, float4 *offsets, uint32_t idx, DRoll d, float4 off) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ring[idx] = d;
        offsets[idx] = off;
    }
}
}  // namespace bfd
extern "C" hipError_t bfk_roll_set(bfd::DRoll *ring, float4 *offsets, uint32_t idx, const bfd::DRoll *d, const float *offset3, hipStream_t stream) {
    const float4 off = offset3 ? make_float4(offset3[0], offset3[1], offset3[2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    hipLaunchKernelGGL(bfd::bf_roll_set_kernel, dim3(1), dim3(64), 0, stream, ring, offsets, idx, *d, off);
    return hipGetLastError();
}

namespace bfd {
// bf_scene_translate_meshes: triangles and node boxes of the pristine copies shifted by `d`.
__global__ void bf_translate_kernel(const float4 *__restrict__ tris0, float4 *__restrict__ tris, uint32_t n_tri_rows,
                                    const float4 *__restrict__ nodes0, float4 *__restrict__ nodes, float4 *__restrict__ qnodes, uint32_t n_nodes,
                                    const float4 *__restrict__ wnodes0, float4 *__restrict__ wnodes, uint32_t n_wchildren,
                                    float dx, float dy, float dz) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_tri_rows) {                       // one float4 row (a vertex + tag) per thread
        float4 v = tris0[i];
        tris[i] = make_float4(v.x + dx, v.y + dy, v.z + dz, v.w);
    }
    if (i < n_nodes) {
        const float4 *s = nodes0 + 8u * i;
        float4 *o = nodes + 8u * i;
        float4 lx = s[0], ly = s[1], lz = s[2], hx = s[3], hy = s[4], hz = s[5];
        // shifted vertices are rounded to fp32 again (half an ulp each) and the fma slab test allows for
        // 2^-24 |origin| (bf_bvh.h): re-pad every box by 2.4e-7 of its largest shifted coordinate
#define BF_SHIFT(L, H, D, K)                                                                                  \
    {                                                                                                         \
        float lo = L.K + D, hi = H.K + D;                                                                     \
        float e = 2.4e-7f * __builtin_fmaxf(__builtin_fabsf(lo), __builtin_fabsf(hi));                        \
        L.K = lo - e;                                                                                         \
        H.K = hi + e;                                                                                         \
    }
        BF_SHIFT(lx, hx, dx, x) BF_SHIFT(lx, hx, dx, y) BF_SHIFT(lx, hx, dx, z) BF_SHIFT(lx, hx, dx, w)
        BF_SHIFT(ly, hy, dy, x) BF_SHIFT(ly, hy, dy, y) BF_SHIFT(ly, hy, dy, z) BF_SHIFT(ly, hy, dy, w)
        BF_SHIFT(lz, hz, dz, x) BF_SHIFT(lz, hz, dz, y) BF_SHIFT(lz, hz, dz, z) BF_SHIFT(lz, hz, dz, w)
#undef BF_SHIFT
        o[0] = lx; o[1] = ly; o[2] = lz; o[3] = hx; o[4] = hy; o[5] = hz;
        o[6] = s[6];
        o[7] = s[7];
        if (qnodes) {
            // re-quantise the shifted node for wf_trace: bf::quantise_node4 (bf_bvh.cpp) operation for operation
            const float cl[3][4] = {{lx.x, lx.y, lx.z, lx.w}, {ly.x, ly.y, ly.z, ly.w}, {lz.x, lz.y, lz.z, lz.w}};
            const float chh[3][4] = {{hx.x, hx.y, hx.z, hx.w}, {hy.x, hy.y, hy.z, hy.w}, {hz.x, hz.y, hz.z, hz.w}};
            const int child[4] = {__float_as_int(s[6].x), __float_as_int(s[6].y), __float_as_int(s[6].z), __float_as_int(s[6].w)};
            float nlo[3], scale[3];
            uint32_t exps = 0, qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
            for (int a = 0; a < 3; ++a) {
                float lo = BF_INF, hi = -BF_INF;
                for (int k = 0; k < 4; ++k)
                    if (child[k] != kNoNode) {
                        lo = __builtin_fminf(lo, cl[a][k]);
                        hi = __builtin_fmaxf(hi, chh[a][k]);
                    }
                nlo[a] = lo;
                int e = 0;
                (void) __builtin_frexpf((hi - lo) * (1.f / 255.f), &e);
                e = max(-100, min(100, e));
                if (!(lo + 255.f * __builtin_ldexpf(1.f, e) >= hi)) ++e;
                scale[a] = __builtin_ldexpf(1.f, e);
                exps |= (uint32_t) (e + 127) << (8 * a);
                for (int k = 0; k < 4; ++k) {
                    uint32_t ql = 255u, qh = 0u;
                    if (child[k] != kNoNode) {
                        const float inv = 1.f / scale[a];
                        float fl = __builtin_floorf((cl[a][k] - lo) * inv), fh = __builtin_ceilf((chh[a][k] - lo) * inv);
                        fl = __builtin_fminf(255.f, __builtin_fmaxf(0.f, fl));
                        fh = __builtin_fminf(255.f, __builtin_fmaxf(0.f, fh));
                        while (fl > 0.f && lo + fl * scale[a] > cl[a][k]) fl -= 1.f;
                        while (fh < 255.f && lo + fh * scale[a] < chh[a][k]) fh += 1.f;
                        ql = (uint32_t) fl;
                        qh = (uint32_t) fh;
                    }
                    qlo[a] |= ql << (8 * k);
                    qhi[a] |= qh << (8 * k);
                }
            }
            float4 *q = qnodes + 4u * i;
            q[0] = make_float4(nlo[0], nlo[1], nlo[2], __uint_as_float(exps));
            q[1] = s[6];
            q[2] = make_float4(__uint_as_float(qlo[0]), __uint_as_float(qlo[1]), __uint_as_float(qlo[2]), __uint_as_float(qhi[0]));
            q[3] = make_float4(__uint_as_float(qhi[1]), __uint_as_float(qhi[2]), 0.f, 0.f);
        }
    }
    if (i < n_wchildren) {                      // one child record of a sixteen-wide node (bf_bvh.h: Node16), same re-padding
        const float4 a = wnodes0[2u * i], b = wnodes0[2u * i + 1u];
        float lo[3] = {a.x + dx, a.y + dy, a.z + dz}, hi[3] = {a.w + dx, b.x + dy, b.y + dz};
        for (int k = 0; k < 3; ++k) {
            const float e = 2.4e-7f * __builtin_fmaxf(__builtin_fabsf(lo[k]), __builtin_fabsf(hi[k]));
            lo[k] -= e;
            hi[k] += e;
        }
        wnodes[2u * i] = make_float4(lo[0], lo[1], lo[2], hi[0]);
        wnodes[2u * i + 1u] = make_float4(hi[1], hi[2], b.z, b.w);
    }
}
}  // namespace bfd

extern "C" hipError_t bfk_launch_translate(const float4 *tris0, float4 *tris, uint32_t n_tri_rows, const float4 *nodes0,
                                           float4 *nodes, float4 *qnodes, uint32_t n_nodes, const float4 *wnodes0, float4 *wnodes,
                                           uint32_t n_wchildren, const float *d, hipStream_t stream) {
    uint32_t n = n_tri_rows > n_nodes ? n_tri_rows : n_nodes;
    n = n > n_wchildren ? n : n_wchildren;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(bfd::bf_translate_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, tris0, tris, n_tri_rows, nodes0, nodes,
                       qnodes, n_nodes, wnodes0, wnodes, n_wchildren, d[0], d[1], d[2]);
    return hipGetLastError();
}

#ifdef BF_TAIL_PROF
extern "C" int bfdbg_tail_profile(unsigned long long *out, int n_waves) {
    if (n_waves > 8192) n_waves = 8192;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(bfd::g_tail_prof), sizeof(unsigned long long) * 24 * (size_t) n_waves) != hipSuccess) return -1;
    unsigned long long zero[8] = {0};
    (void) zero;
    return n_waves;
}
extern "C" int bfdbg_tail_profile_clear(void) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(bfd::g_tail_prof)) != hipSuccess) return -1;
    return hipMemset(p, 0, sizeof(unsigned long long) * 24 * 8192) == hipSuccess ? 0 : -1;
}
#endif

/* Host-side evaluation of the engine's fp32 cosine (bf_device_math.h), used by scene setup so that
 * precomputed emitter constants follow the same specification as the kernels. */
extern "C" float bfk_host_cos(float x) { return bfd::bf_cos(x); }

namespace bfd {
__global__ void bf_elementary_kernel(int op, uint64_t n, const float *__restrict__ x, float *__restrict__ y) {
    uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = x[i], r;
    switch (op) {
        case 0: r = sin_cr(v); break;
        case 1: r = cos_cr(v); break;
        case 2: r = acos_cr(v); break;
        case 3: r = exp_cr(v); break;
        case 4: r = log_cr(v); break;
        case 5: r = erf_cr(v); break;
        default: r = tan_cr(v); break;
    }
    y[i] = r;
}
}  // namespace bfd

extern "C" hipError_t bfk_launch_elementary(int op, uint64_t n, const float *x, float *y) {
    unsigned grid = (unsigned) ((n + 255) / 256);
    hipLaunchKernelGGL(bfd::bf_elementary_kernel, dim3(grid), dim3(256), 0, 0, op, n, x, y);
    return hipGetLastError();
}

extern "C" hipError_t bfk_launch_trace(const bfd::DScene *sc, uint64_t n, const float *rays, int any_hit, float *out_t,
                                       uint32_t *out_prim, uint32_t *out_shape, float *out_uv, uint8_t *out_hit,
                                       float *out_si, hipStream_t stream) {
    unsigned grid = (unsigned) std::min<uint64_t>((n + bfd::kBlock - 1) / bfd::kBlock, sc->spill_stride / bfd::kBlock);
    if (grid == 0) return hipSuccess;
    if (sc->stack_need > (uint32_t) bfd::kStackDepth)
        hipLaunchKernelGGL(bfd::bf_trace_kernel<true>, dim3(grid), dim3(bfd::kBlock), 0, stream, *sc, n, rays, any_hit, out_t,
                           out_prim, out_shape, out_uv, out_hit, out_si);
    else
        hipLaunchKernelGGL(bfd::bf_trace_kernel<false>, dim3(grid), dim3(bfd::kBlock), 0, stream, *sc, n, rays, any_hit, out_t,
                           out_prim, out_shape, out_uv, out_hit, out_si);
    return hipGetLastError();
}
