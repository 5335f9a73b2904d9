// Wavefront path tracer for gfx950: the radar hot path as two persistent kernels per bounce iteration over a
// pool of path SLOTS whose state stays in place in HBM (bf_wavefront.h).  There are no queues and no device-wide
// counters on the data path: which slots need work is three 64-bit masks per 64-slot batch (alive / trace /
// shadow), every wave owns a contiguous segment of batches, and a MaskCursor (bf_path_logic.h) packs the set bits
// of its segment into full waves on the fly.
//
//   wf_shade : one lane per live slot.  Coalesced state load -> the integrator's vertex logic (emitter hit,
//              Russian roulette, next-event estimation, BSDF sampling: path.cpp:121-209 and the pathlength /
//              pathtime / pathtimefrequency variants, bf_path_logic.h: shade_vertex) -> state stored back in
//              place.  Finished paths are binned (film_put: LDS-privatised histogram) and the slot starts its next
//              path (static assignment: slot i renders paths i, i + n_slots, ...).  New rays are tested here
//              against the analytic rectangles and the four child boxes of the BVH root (presolve_ray); a lane whose
//              rays are all answered that way shades its next vertex in the same visit (up to kShadeChain).
//   wf_trace : persistent waves with dynamic ray replacement walk the four-wide BVH for the rays that do enter
//              the mesh: while-while traversal, 16-entry LDS stack per lane (+ HBM spill), the tree's top 85 nodes
//              in LDS.  Closest hits go to hit[slot]; an unoccluded shadow ray releases its NEE contribution into
//              the slot's result.
//
// The tail of a render (few, long paths) is finished by bf_render_kernel<.., RESUME = true> (bf_kernels.hip).
// All arithmetic is fp32 (the transcendentals are the engine's own fp32 specification, bf_device_math.h; the few
// double operations are where the reference itself multiplies by a double literal, e.g. 1e-9 in phase_update and
// freq_of).  Results are bit-identical per path to the one-kernel variant and to the oracle: same draws, same
// operations in the same order.
#include "bf_path_logic.h"

namespace bfd {

// OR-reduce per-lane bit contributions into the per-batch mask words.  Dense,
// aligned rounds (every lane holds slot batch*64 + lane) store the ballot
// directly; gathered rounds use atomicOr on the (pre-zeroed) words — spread over
// many addresses, non-returning.
BF_DEV void publish_masks(unsigned long long *m_out, bool aligned, bool merge, uint32_t batch0, uint32_t slot, bool valid_lane, bool bit) {
    if (aligned) {
        unsigned long long w = __ballot(bit);
        if ((threadIdx.x & 63) == 0 && w) {
            if (merge)
                atomicOr(&m_out[batch0], w);      // wake launch: wf_shade<0> of the same iteration wrote this word before
            else
                m_out[batch0] = w;
        }
    } else if (valid_lane && bit) {
        atomicOr(&m_out[slot >> 6], 1ull << (slot & 63u));
    }
}

// Lane-refill form of wf_shade (BF_SHADE_REFILL): the lanes of a wave hold slots of several batches and settle at different
// times, so the bits of one settle event are gathered per batch — transposed from lane order to slot order through 64 bytes of
// LDS — and OR-ed into the (pre-zeroed) words by ONE lane: three atomics per batch and event instead of three per slot.  Pools
// so sparse that every lane holds a different batch fall back to one atomic per lane after kPublishGroups batches.
// bits: 1 alive, 2 trace, 4 shadow.
constexpr uint32_t kPublishGroups = 4;
BF_DEV void publish_grouped(volatile unsigned char *tr, int lane, bool settle, uint32_t slot, uint32_t bits, unsigned long long *m_alive,
                            unsigned long long *m_trace, unsigned long long *m_shadow) {
    unsigned long long todo = __ballot(settle && bits != 0u);
    for (uint32_t n = 0; todo != 0ull && n < kPublishGroups; ++n) {
        const int j = __ffsll(todo) - 1;
        const uint32_t b = (uint32_t) __shfl((int) (slot >> 6), j);
        const bool mine = settle && bits != 0u && (slot >> 6) == b;
        tr[lane] = 0;
        if (mine) tr[slot & 63u] = (unsigned char) bits;
        const uint32_t f = tr[lane];
        const unsigned long long a = __ballot((f & 1u) != 0u), t = __ballot((f & 2u) != 0u), q = __ballot((f & 4u) != 0u);
        if (lane == 0) {
            if (a) atomicOr(&m_alive[b], a);
            if (t) atomicOr(&m_trace[b], t);
            if (q) atomicOr(&m_shadow[b], q);
        }
        todo &= ~__ballot(mine);
    }
    if ((todo >> lane) & 1ull) {
        const unsigned long long bit = 1ull << (slot & 63u);
        if (bits & 1u) atomicOr(&m_alive[slot >> 6], bit);
        if (bits & 2u) atomicOr(&m_trace[slot >> 6], bit);
        if (bits & 4u) atomicOr(&m_shadow[slot >> 6], bit);
    }
}

// Early resolution of a freshly spawned ray, done by the shading lane itself
// while the whole wave is active: the analytic rectangles (rectangle.cpp:229-263)
// are tested here, and if the ray misses all child boxes of the BVH root no
// triangle can be hit, so the query is already answered.  Only rays that enter
// the mesh BVH are handed to wf_trace (with the best rectangle hit so far as
// their starting point), which keeps that kernel's waves filled with comparable
// work instead of mixing 3-step misses with 60-step hits.
//   returns true if the ray still needs BVH traversal
BF_DEV bool presolve_ray(const DScene &sc, bool any, V3 o, V3 d, float mint, float maxt, Hit &best, bool &found, const Shift &shf) {
    best.t = BF_INF;
    best.u = best.v = 0.f;
    best.prim = 0;
    best.slot = 0;
    found = false;
    for (uint32_t i = 0; i < sc.n_rects; ++i) {
        CRect &rc = c_rects(sc)[i];
        float t, lx, ly;
        if (rect_intersect(rc, o, d, mint, maxt, t, lx, ly)) {
            if (any) {
                found = true;
                return false;
            }
            consider(best, t, lx, ly, rc.prim, -(int32_t) (i + 1));
        }
    }
    if (sc.n_tris == 0) return false;
    if (sc.root < 0) return true;                 // the whole mesh is one leaf
    // the root node's address is wave-uniform: scalar loads (constant address space, bf_device.h: BF_CAS)
    const c_f4_ptr np = (c_f4_ptr) (uintptr_t) (sc.nodes + 8u * (uint32_t) sc.root);
    const float4 lx = to_float4(np[0]), ly = to_float4(np[1]), lz = to_float4(np[2]), hx = to_float4(np[3]), hy = to_float4(np[4]),
                 hz = to_float4(np[5]), ch = to_float4(np[6]);
    V3 id, oid, ohi;
    ray_inverse_shift(o, d, shf, id, oid, ohi);
    float tmax = any ? maxt : __builtin_fminf(maxt, best.t), tn;
    return slab_fma(lx.x, ly.x, lz.x, hx.x, hy.x, hz.x, id, oid, ohi, mint, tmax, tn) ||
           slab_fma(lx.y, ly.y, lz.y, hx.y, hy.y, hz.y, id, oid, ohi, mint, tmax, tn) ||
           (__float_as_int(ch.z) != kNoNode && slab_fma(lx.z, ly.z, lz.z, hx.z, hy.z, hz.z, id, oid, ohi, mint, tmax, tn)) ||
           (__float_as_int(ch.w) != kNoNode && slab_fma(lx.w, ly.w, lz.w, hx.w, hy.w, hz.w, id, oid, ohi, mint, tmax, tn));
}

// Survivor-area allocation of a wave (rolling sequences, bf_wavefront.h: WF::n_surv).  The wave claims whole survivor
// batches with one returning atomic each and hands their free slots — bits not alive in the parity being consumed; the
// claim is exclusive for the launch — to the lanes that evict a path.  A wave that exhausts its claims (or finds only
// full batches) leaves the remaining paths where they are: they then run late, which is slower, never wrong.
struct SurvAlloc {
    unsigned long long free;    // unclaimed free slots of the current batch (wave-uniform)
    uint32_t batch, claims;
    uint32_t rot;               // claims of the sequence's earlier launches (surv_cursor[0], read once per kernel)
};
BF_DEV void surv_take(const WF &wf, int cur, SurvAlloc &sv, unsigned long long em, bool &evict, uint32_t &dst, int lane) {
    const uint32_t need = (uint32_t) __popcll(em), rank = (uint32_t) __popcll(em & ((1ull << lane) - 1ull));
    const uint32_t main_b = wf.n_main >> 6, surv_b = wf.n_surv >> 6;
    uint32_t served = 0;
    while (served < need) {
        if (sv.free == 0ull) {
            if (sv.claims >= wf.surv_claims_max || surv_b == 0u) break;
            // claim number `nth` of THIS launch (one returning atomic) picks batch (rot + nth) mod surv_b: distinct batches for
            // nth < surv_b whatever the order the waves arrive in.  rot = the claims of all earlier launches of the sequence
            // (surv_cursor[0], constant during the launch: the wake launch that follows folds this launch's count into it), so a
            // launch starts looking where the previous one stopped.
            uint32_t nth = 0;
            if (lane == 0) nth = atomicAdd(wf.surv_cursor + 1, 1u);
            nth = (uint32_t) __shfl((int) nth, 0);
            if (nth >= surv_b) {
                // The launch has handed out every survivor batch once: the next one would be a batch ANOTHER wave of this launch
                // holds, and two waves filling the same free slots lose paths silently (round 3, commit 129745c).  The sizing
                // rule (surv_claims_max x waves <= batches) keeps this from happening; if it ever does, the claim is refused —
                // the paths stay in their slots: slower, never wrong — and counted: the flush / sync of the handle fails.
                if (lane == 0) atomicAdd(&wf.counters[CTR_SURV_GUARD], 1ull);
                sv.claims = wf.surv_claims_max;
                break;
            }
            sv.batch = main_b + (sv.rot + nth) % surv_b;
            sv.free = ~wf.m_alive[cur][sv.batch];
            ++sv.claims;
            continue;
        }
        const uint32_t cnt = (uint32_t) __popcll(sv.free), take = min(need - served, cnt);
        if (evict && rank >= served && rank < served + take) dst = sv.batch * 64u + nth_set_bit(sv.free, rank - served);
        if (take == cnt) {
            sv.free = 0ull;
        } else {
            const uint32_t p = nth_set_bit(sv.free, take - 1u);
            sv.free &= ~((2ull << p) - 1ull);
        }
        served += take;
    }
    if (evict && rank >= served) evict = false;        // no room: the path stays in its slot
}

// wf_shade: one lane per live slot (see the file header of bf_wavefront.h).
//   FIRST = 0 : walk the alive masks of the current parity.
//   FIRST = 1 : bounce 0 of a render (or of a rolling sequence): every slot < n_slots starts its first path.
//   FIRST = 3 : as 0, and paths whose slot is due for its next path move to the survivor area (first launch of a rolling call).
//   FIRST = 2 : "wake" launch of a rolling sequence (bf_render_device with BF_FLAG_ROLLING): the path supply has just
//               grown by one render, so every slot that is NOT alive — it ran out of paths during an earlier call —
//               starts its next path (the one after the last it finished: wf.sd keeps that index for dead slots).
//               The slots that ARE alive belong to wf_shade<0> of the same iteration; both publish into the same
//               next-parity masks.
#ifdef BF_SHADE_PROF
// developer build (tools/shade_profile.py): cycles of the wave between consecutive stamps, by section (the stamps sit in
// wave-uniform control flow); SLP (bf_path_logic.h) counts wave entries and lanes per section
#define SLT(k)                                                            \
    do {                                                                  \
        const unsigned long long slt_now = __builtin_amdgcn_s_memtime();  \
        t_sec[k] += slt_now - slt_last;                                   \
        slt_last = slt_now;                                               \
    } while (0)
#else
#define SLT(k)
#endif
// BF_SHADE_REFILL = 1: wf_shade with lane refill and phase voting (below); 0: one visit per 64 slots with chained rounds.
#ifndef BF_SHADE_REFILL
#define BF_SHADE_REFILL 0
#endif
template <int FIRST, int W, int RX>
__global__ __launch_bounds__(kBlock, W) void wf_shade(DScene sc_arg, DLaunch lp, WF wf, uint32_t it, float *__restrict__ g_hist,
                                                      bf_path_record *__restrict__ records) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    float *s_hist = reinterpret_cast<float *>(s_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    DScene sc = sc_arg;
    if (RX & kMulti) sc.tab_cache = 0u;      // several versions of the tables in flight (path_scene): no LDS copy of one of them
    load_tables_lds(sc, (4u * lp.lds_floats + 15u) & ~15u, (uint32_t) tid);      // materials + rectangles behind the histogram
    constexpr bool WALK = FIRST == 0 || FIRST == 3;      // the launch walks the alive masks (else: whole batches of main slots)
    constexpr bool EVICT = FIRST == 3;                   // ... and moves long paths to the survivor area (own variant: the
                                                         // allocator's live values would cost wf_shade<0> its scratch-free build)
    const bool lds_hist = lp.lds_hist != 0;
    if (lp.lds_floats || sc.tab_on) {
        for (uint32_t i = tid; i < lp.lds_floats; i += kBlock) s_hist[i] = 0.f;
        __syncthreads();
    }
    const int cur = it & 1, nxt = cur ^ 1;
    const bool receive = mode_receive<RX>(lp);
    if (FIRST == 2 && blockIdx.x == 0 && tid == 0) {
        // the wake launch follows the evicting launch of its call (same stream): fold that launch's claims into the rotation
        wf.surv_cursor[0] += wf.surv_cursor[1];
        wf.surv_cursor[1] = 0u;
    }
    // the first / wake launches start paths: main slots only (the survivor area never regenerates)
    // (the wake launch: only the batches whose slots are due — the other half of the main slots holds the previous render)
    const uint32_t n_batches = WALK ? wf.n_slots >> 6 : (FIRST == 2 ? wf.wake_nb : wf.n_main >> 6);
    unsigned long long *m_alive = wf.m_alive[nxt], *m_trace = wf.m_trace[nxt], *m_shadow = wf.m_shadow[nxt], *m_hit = wf.m_hit[nxt];
    SurvAlloc sv = {0ull, 0u, 0u, EVICT ? wf.surv_cursor[0] : 0u};

    FilmAcc acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0u, 0u};
    uint32_t c_closest = 0, c_shadow = 0, c_bounces = 0, c_live = 0, c_traced = 0, c_loads = 0, c_shq = 0;
#ifdef BF_SHADE_PROF
    const bool lpf = BF_SHADE_PROF >= 2;      // -DBF_SHADE_PROF=2: lane counts per section too (an atomic pair per entry: the cycle shares are then perturbed)
    ShadeProf spf = {0, 0, 0, 0};
    unsigned long long t_sec[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // cursor, load, vertex + film, generate, presolve, chain, store
    unsigned long long slt_last = __builtin_amdgcn_s_memtime();
#endif

    // interleaved share of the batches per wave (MaskCursor; the first / wake launches: batches wave_id, wave_id + n_waves, ...)
    const uint32_t n_waves = gridDim.x * (kBlock / 64), wave_id = blockIdx.x * (kBlock / 64) + (tid >> 6);
    MaskCursor cur_alive;
    // two walks of the alive masks (WF::m_hit): first the slots whose pending vertex is no real hit, then the real hits
    const uint32_t n_pass = (WALK && wf.hit_split) ? 2u : 1u;
    uint32_t first_b = wave_id;
    const bool rolling = lp.roll != nullptr;

#if BF_SHADE_REFILL
    // ---- lane refill + phase voting --------------------------------------------------------------------------------------
    // A lane keeps a slot only while its path's next step needs no trace launch; the moment it settles (state written back, or the
    // slot out of paths) it takes the wave's next slot.  What a lane has pending is one of: a REAL hit to shade (the expensive
    // vertex: surface interaction, next-event estimation, BSDF sample), or cheap work (a ray that left the scene, a film write, the
    // slot's next path to start).  Each loop iteration the wave votes and runs ONE kind for all the lanes that hold it; the others
    // keep their registers and wait, so both kinds run with most of the wave instead of a decaying subset (tools/shade_profile.py:
    // 31 lanes per expensive-section entry, 22 per film write before this form).  Per-path results do not depend on the schedule.
    (void) n_pass;
    (void) first_b;
    (void) m_hit;
    __shared__ unsigned char s_tr[kBlock];
    volatile unsigned char *tr = s_tr + (tid & ~63);
    if (WALK)
        cursor_init(cur_alive, wf.m_alive[cur], wave_id, n_waves, n_batches, lane);
    else if (FIRST == 2)      // the due slots WITHOUT a live path (as of now: wf_shade<3> of this iteration has run)
        cursor_init(cur_alive, m_alive, wave_id, n_waves, wf.wake_nb, lane, 1u, nullptr, 0u, wf.wake_b0, wf.n_main >> 6, 1u);
    else
        cursor_init(cur_alive, nullptr, wave_id, n_waves, wf.n_main >> 6, lane);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    bool has = false, have_hit = false, need_film = false, need_gen = false, touched = false;
    uint32_t slot = 0, rounds = 0, guard = 0;
    PathState s;
    s.render = 0u;
    s.flags = 0u;
    s.path_i = 0ull;
    ShadowReq sh;
    sh.want = false;
    Hit hit;
    hit.t = BF_INF;
    hit.u = hit.v = 0.f;
    hit.prim = 0;
    hit.slot = 0;
    bool done = false;              // the lane's outcome for the launch is final: written back at the next exchange
    uint32_t out_bits = 0u;         // ... with these mask bits (1 alive, 2 trace, 4 shadow; 0: the slot has run out of paths)
    const uint32_t rf_min = wf.rf_min, rf_th = wf.rf_th, rf_tm = wf.rf_tm;
    while (true) {
        const bool pend_any = has && !done;
        const unsigned long long xm = __ballot(!pend_any);             // lanes without work: settled or free
        // ---- exchange: settled lanes write back, then every free lane takes the wave's next slot ---------------------
        // (lazily: once rf_min lanes are out of work, or nothing else is left to do — one store / publish / load round
        // trip for many lanes instead of one per loop iteration)
        if ((uint32_t) __popcll(xm) >= rf_min || xm == ~0ull) {
            if (__any(done)) {
                const bool keep = done && out_bits != 0u;          // a live path goes back to its slot (or to the survivor area)
                const bool tracing = (out_bits & 2u) != 0u, shadowing = (out_bits & 4u) != 0u;
                bool evict = false;
                uint32_t dst = slot;
                SLP(20, keep);
                if (EVICT) {
                    // only out of the batches THIS call's wake launch visits (see the batch-visit form below)
                    const uint32_t main_b = wf.n_main >> 6, b = slot >> 6;
                    const uint32_t rel = b >= wf.wake_b0 ? b - wf.wake_b0 : b + main_b - wf.wake_b0;
                    evict = keep && slot < wf.n_main && rel < wf.wake_nb && s.path_i + wf.n_main < lp.n_paths;
                    const unsigned long long em = __ballot(evict);
                    if (em) surv_take(wf, cur, sv, em, evict, dst, lane);
                    if (evict) wf.sd(slot) = make_uint4(0u, 0u, (uint32_t) s.path_i, (uint32_t) (s.path_i >> 32));
                }
                if (keep) {
                    if (!(s.flags & kFlagTermPending)) {
                        // resolved rays carry their final hit; the others start wf_trace from the rectangle hit
                        wf.hit(dst) = make_float4(hit.t, hit.u, hit.v, __int_as_float(hit.slot));
                        wf.hit_prim(dst) = hit.prim;
                    }
                    store_state(wf, dst, receive, s);
                    ++c_live;
                    if (shadowing) {
                        wf.sh0(dst) = make_float4(sh.o.x, sh.o.y, sh.o.z, sh.mint);
                        wf.sh1(dst) = make_float4(sh.d.x, sh.d.y, sh.d.z, sh.maxt);
                        wf.sh2(dst) = sh.c;
                        if (receive && lp.iq) wf.sh3(dst) = sh.c_im;
                    }
                    c_traced += (tracing ? 1u : 0u) + (shadowing ? 1u : 0u);
                    c_shq += shadowing ? 1u : 0u;
                } else if (done && rolling && touched) {
                    // the slot has run out of paths for now: remember the last one it rendered (the wake launch of the
                    // sequence's next call continues from there)
                    wf.sd(slot) = make_uint4(0u, 0u, (uint32_t) s.path_i, (uint32_t) (s.path_i >> 32));
                }
                publish_grouped(tr, lane, keep && !evict, slot, out_bits, m_alive, m_trace, m_shadow);
                if (evict) {                 // the moved path's bits go to its new slot (the batch's owner ORs its own in as well)
                    const unsigned long long bit = 1ull << (dst & 63u);
                    atomicOr(&m_alive[dst >> 6], bit);
                    if (tracing) atomicOr(&m_trace[dst >> 6], bit);
                    if (shadowing) atomicOr(&m_shadow[dst >> 6], bit);
                }
                if (done) {
                    done = false;
                    has = false;
                    sh.want = false;
                    out_bits = 0u;
                }
            }
            SLT(5);
            if (!cursor_empty(cur_alive)) {
                cursor_skip_empty(cur_alive, lane);
                const unsigned long long fm = __ballot(!has);
                const uint32_t rank = (uint32_t) __popcll(fm & lt_mask);
                uint32_t ns = 0;
                const uint32_t got = cursor_take(cur_alive, (uint32_t) __popcll(fm), !has, rank, ns, lane);
                const bool fresh = !has && rank < got;
                SLP(0, fresh);
                SLT(0);
                if (fresh) {
                    slot = ns;
                    has = true;
                    rounds = 0u;
                    sh.want = false;
                    have_hit = need_film = need_gen = false;
                    touched = FIRST != 2;
                    if (!WALK) {
                        need_gen = true;
                        s.render = 0u;
                        s.flags = 0u;
                        // the path BEFORE the one this slot starts now (see the batch-visit form below)
                        s.path_i = (uint64_t) slot - (uint64_t) wf.n_main;
                        if (FIRST == 2) {
                            const uint4 d = wf.sd(slot);
                            s.path_i = ((uint64_t) d.w << 32) | d.z;
                        }
                    } else {
                        load_state(wf, slot, receive, s);
                        ++c_loads;
                        if (s.flags & kFlagTermPending) {
                            need_film = true;      // ended after last bounce's BSDF sample; its NEE shadow ray has resolved by now
                        } else {
                            float4 hq = wf.hit(slot);
                            hit.t = hq.x;
                            hit.u = hq.y;
                            hit.v = hq.z;
                            hit.slot = __float_as_int(hq.w);
                            have_hit = true;
                        }
                    }
                }
                SLT(1);
            }
        }
        // ---- vote ---------------------------------------------------------------------------------------------------------
        const bool pend_h = has && !done && have_hit && hit.t != BF_INF;
        const bool pend_m = has && !done && !pend_h;           // (a lane with a slot that is not settled always has something pending)
        const uint32_t n_h = (uint32_t) __popcll(__ballot(pend_h)), n_m = (uint32_t) __popcll(__ballot(pend_m));
        if (n_h + n_m == 0u) break;                            // every lane free and the source exhausted
        if (++guard > (1u << 26)) {                            // (no schedule comes near this: a loud end instead of a hung GPU)
            if (lane == 0) atomicAdd(&wf.counters[CTR_GUARD], 1ull);
            break;
        }
        // real hits run once enough of them wait (or too little else is pending); cheap work runs in between and feeds them
        const bool run_h = n_h != 0u && (n_h >= rf_th || n_m < rf_tm);
        const bool go = run_h ? pend_h : pend_m;
        bool cont = false;
        // ---- vertex: the real hits (run_h) or the rays that left the scene ---------------------------------------------------
        SLP(2, go && have_hit);
        if (go && have_hit) {
            have_hit = false;
#ifdef BF_SHADE_PROF
            cont = shade_vertex<RX>(sc, lp, s, hit, sh, c_bounces, &spf, lpf);
#else
            cont = shade_vertex<RX>(sc, lp, s, hit, sh, c_bounces);
#endif
            if (!cont)
                need_film = true;
            else if (!(s.flags & kFlagTermPending))
                ++c_closest;
            if (sh.want) ++c_shadow;
        }
        SLT(2);
        // ---- film write and the slot's next path: cheap passes only (a path that ends in an expensive pass waits for one) ------
        const bool go_m = go && !run_h;
        SLP(5, go_m && need_film);
        if (go_m && need_film) {
            need_film = false;
            film_put<RX>(sc, lp, s, s_hist, g_hist, lds_hist, acc, records);
            need_gen = true;
        }
        SLP(6, go_m && need_gen);
        bool fin = false;
        if (go_m && need_gen) {
            // regeneration: slot i renders paths i, i + n_main, i + 2 n_main, ... (static assignment: no device-wide counter)
            need_gen = false;
            const uint64_t path_i = s.path_i + wf.n_main;
            if (slot < wf.n_main && path_i < lp.n_paths) {
                touched = true;
                generate_path<RX>(sc, lp, path_i, s);
                sh.want = false;
                ++c_closest;
                cont = true;
            } else {
                fin = true;
            }
        }
        SLT(3);
        // ---- early resolution of the new rays ---------------------------------------------------------------------------------
        bool tracing = false, shadowing = false;
        if (go) {
            tracing = cont && !(s.flags & kFlagTermPending);
            shadowing = cont && sh.want;
            const Shift shf = path_shift(lp, s.render);
            const DScene scp = path_scene<RX>(sc, lp, s.render);
            SLP(7, shadowing);
            SLP(8, tracing);
            if (shadowing) {
                Hit tmp;
                bool found;
                if (!presolve_ray(scp, true, sh.o, sh.d, sh.mint, sh.maxt, tmp, found, shf)) {
                    s.result += found ? sh.c * 0.f : sh.c;      // (c * 0: see the batch-visit form)
                    if (receive && lp.iq) s.phase += found ? sh.c_im * 0.f : sh.c_im;
                    shadowing = false;
                }
                sh.want = shadowing;
            }
            if (tracing) {
                bool found;
                tracing = presolve_ray(scp, false, s.ro, s.rd, s.rmint, s.rmaxt, hit, found, shf);
            }
        }
        SLT(4);
        // ---- keep the slot (both rays answered: the next vertex is known) or settle ------------------------------------------
        const bool resolved = go && cont && !tracing && !shadowing;
        const bool chain = resolved && rounds + 1u < wf.shade_chain;
        if (go) ++rounds;
        if (chain) {
            if (s.flags & kFlagTermPending)
                need_film = true;         // ended at its last BSDF sample and its NEE ray is answered: bin it in a cheap pass
            else
                have_hit = true;          // `hit` is the final answer of the continuation ray
        }
        if (go && !chain && (cont || fin)) {      // (neither: the path ended in an expensive pass, its film write is pending)
            done = true;
            out_bits = cont ? (1u | (tracing ? 2u : 0u) | (shadowing ? 4u : 0u)) : 0u;
        }
    }
#else
    for (uint32_t pass = 0; pass < n_pass; ++pass) {
    if (WALK) cursor_init(cur_alive, wf.m_alive[cur], wave_id, n_waves, n_batches, lane, 1u, n_pass == 2u ? wf.m_hit[cur] : nullptr, pass ? 1u : 2u);
    while (true) {
        // ---- gather up to 64 live slots of the segment into the lanes -----------
        uint32_t slot = 0, got;
        bool aligned;
        if (!WALK) {
            if (first_b >= n_batches) break;
            uint32_t fb = first_b;
            if (FIRST == 2) {
                fb += wf.wake_b0;
                if (fb >= wf.n_main >> 6) fb -= wf.n_main >> 6;
            }
            slot = fb * 64u + lane;
            got = 64;
            aligned = true;
            first_b += n_waves;
        } else {
            cursor_skip_empty(cur_alive, lane);
            if (cursor_empty(cur_alive)) break;
            const uint32_t batch_before = cur_alive.b;
            const bool whole = __popcll(cur_alive.m) == 64;
            got = cursor_take(cur_alive, 64u, true, (uint32_t) lane, slot, lane);
            if (got == 0) break;
            aligned = whole && got == 64 && slot == batch_before * 64u + (uint32_t) lane;
            aligned = __all(aligned);
        }
        bool has = (uint32_t) lane < got;
        const uint32_t batch0 = slot >> 6;
        // wake: only the slots without a live path — as of NOW: wf_shade<0> of this iteration has run, so a slot whose path
        // ended there, or moved to the survivor area, is free
        if (FIRST == 2) has = ((m_alive[batch0] >> lane) & 1ull) == 0ull;

        PathState s;
        s.render = 0u;                 // lanes without a path still index the batch tables (path_shift)
        ShadowReq sh;
        sh.want = false;
        bool need_gen = false, cont = false, have_hit = false;
        Hit hit;
        hit.t = BF_INF;
        hit.u = hit.v = 0.f;
        hit.prim = 0;
        hit.slot = 0;

        SLP(0, has);
        SLT(0);
        if (has) {
            if (!WALK) {
                need_gen = true;
                // the path BEFORE the one this slot starts now: slot - n_slots for a fresh pool (wraps: + n_slots = slot),
                // the slot's last finished path in a wake launch
                s.path_i = (uint64_t) slot - (uint64_t) wf.n_main;
                if (FIRST == 2) {
                    const uint4 d = wf.sd(slot);
                    s.path_i = ((uint64_t) d.w << 32) | d.z;
                }
            } else {
                load_state(wf, slot, receive, s);
                ++c_loads;
                SLP(1, (s.flags & kFlagTermPending) != 0);
                if (s.flags & kFlagTermPending) {
                    // ended after last bounce's BSDF sample; its NEE shadow ray has resolved by now
                    film_put<RX>(sc, lp, s, s_hist, g_hist, lds_hist, acc, records);
                    need_gen = true;
                } else {
                    float4 hq = wf.hit(slot);
                    hit.t = hq.x;
                    hit.u = hq.y;
                    hit.v = hq.z;
                    hit.slot = __float_as_int(hq.w);
                    have_hit = true;
                }
            }
        }
#ifdef BF_SHADE_PROF
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        SLT(1);
        // Chained shading: a lane whose new rays are both answered by the early resolution below
        // (rectangle hit or miss of the mesh, NEE ray clear of the mesh) needs no trace launch, so it
        // shades its next vertex — or starts its next path — right away, up to wf.shade_chain rounds per
        // visit.  Two thirds of all rays resolve that way; each chained round saves a state round trip
        // through HBM and, for paths that leave the scene, a whole bounce iteration.
        bool settled = !has;          // this lane's outcome for the launch is final
        // the slot's path index in wf.sd needs (re)writing if its path ends here: always, except for the idle slots a wake
        // launch finds not due yet (their entry stands); the first launch of a pool writes the initial entries
        bool touched = FIRST != 2;
        bool tracing = false, shadowing = false;
        for (uint32_t round = 0;; ++round) {
            // ---- vertex logic ---------------------------------------------------------------
            SLP(2 + min(round, 2u), !settled && have_hit);
            if (!settled && have_hit) {
                have_hit = false;
#ifdef BF_SHADE_PROF
                cont = shade_vertex<RX>(sc, lp, s, hit, sh, c_bounces, &spf, lpf);
#else
                cont = shade_vertex<RX>(sc, lp, s, hit, sh, c_bounces);
#endif
                SLT(7);
                SLP(5, !cont);
                if (!cont) {
                    film_put<RX>(sc, lp, s, s_hist, g_hist, lds_hist, acc, records);
                    need_gen = true;
                } else if (!(s.flags & kFlagTermPending)) {
                    ++c_closest;
                }
                if (sh.want) ++c_shadow;
            }
            // ---- regeneration: slot i renders paths i, i + n_slots, i + 2 n_slots, ... --------
            // (static assignment: no device-wide path counter to serialise on)
            SLT(2);
            SLP(6, !settled && need_gen);
            if (!settled && need_gen) {
                need_gen = false;
                cont = false;
                const uint64_t path_i = s.path_i + wf.n_main;
                if (slot < wf.n_main && path_i < lp.n_paths) {
                    touched = true;
                    generate_path<RX>(sc, lp, path_i, s);
                    sh.want = false;
                    ++c_closest;
                    cont = true;
                }
            }
            SLT(3);
            // ---- early resolution of the new rays ---------------------------------------------
            if (!settled) {
                tracing = cont && !(s.flags & kFlagTermPending);
                shadowing = cont && sh.want;
                const Shift shf = path_shift(lp, s.render);
                const DScene scp = path_scene<RX>(sc, lp, s.render);      // (kMulti: the rectangles of the path's own render)
                SLP(7, shadowing);
                SLP(8, tracing);
                if (shadowing) {
                    Hit tmp;
                    bool found;
                    if (!presolve_ray(scp, true, sh.o, sh.d, sh.mint, sh.maxt, tmp, found, shf)) {
                        // Scene::sample_emitter_direction zeroes the VALUE of an occluded sample (scene.cpp:220-224) and
                        // the integrator still adds mis * throughput * bsdf * 0: a NaN / inf BSDF value survives that
                        // product.  c * 0 is that term (+-0 for every finite c).
                        s.result += found ? sh.c * 0.f : sh.c;
                        if (receive && lp.iq) s.phase += found ? sh.c_im * 0.f : sh.c_im;
                        shadowing = false;
                    }
                    sh.want = shadowing;
                }
                if (tracing) {
                    bool found;
                    tracing = presolve_ray(scp, false, s.ro, s.rd, s.rmint, s.rmaxt, hit, found, shf);
                }
            }
            SLT(4);
            // ---- chain or settle ----------------------------------------------------------------
            const bool resolved = !settled && cont && !tracing && !shadowing;
            // a resolved REAL hit chains only in company (WF::chain_min): a handful of lanes would run the whole vertex at a
            // fraction of the wave; stored instead, they are shaded packed by the next launch's second pass
            const bool real_hit = !(s.flags & kFlagTermPending) && hit.t != BF_INF;
            const bool lonely = wf.chain_min != 0u && (uint32_t) __popcll(__ballot(resolved && real_hit)) < wf.chain_min;
            const bool chain = resolved && round + 1u < wf.shade_chain && !(lonely && real_hit);
            if (!__any(chain)) break;
            SLP(9, chain && (s.flags & kFlagTermPending));
            SLP(17 + min(round, 2u), chain);
            if (chain) {
                if (s.flags & kFlagTermPending) {
                    // the path ended at its last BSDF sample and its NEE ray is answered: bin it now
                    film_put<RX>(sc, lp, s, s_hist, g_hist, lds_hist, acc, records);
                    need_gen = true;
                    cont = false;
                } else {
                    have_hit = true;                    // `hit` is the final answer of the continuation ray
                }
            } else {
                settled = true;
            }
        }
        SLT(5);
        // ---- write back in place ----------------------------------------------------------------
        SLP(20, has && cont);
        SLP(21, has && cont && shadowing);
        // rolling sequence, first launch of a call: a path that goes on while its slot's NEXT path has been supplied moves
        // to the survivor area (its state is in registers anyway: the write-back simply goes to another slot)
        bool evict = false;
        uint32_t dst = slot;
        if (EVICT) {
            // ... and only out of the batches THIS call's wake launch visits (the slots due for the new render's paths): a
            // slot that has fallen behind (the survivor area was full when it was due: its path stayed in place) may sit in the
            // other half of the main slots, and a slot vacated there would stay dead — its pending paths unrendered — until its
            // half is woken again, or for good if the sequence is flushed first (round 4: found by the CTR_FILM check)
            const uint32_t main_b = wf.n_main >> 6, b = slot >> 6;
            const uint32_t rel = b >= wf.wake_b0 ? b - wf.wake_b0 : b + main_b - wf.wake_b0;
            evict = has && cont && slot < wf.n_main && rel < wf.wake_nb && s.path_i + wf.n_main < lp.n_paths;
            const unsigned long long em = __ballot(evict);
            if (em) surv_take(wf, cur, sv, em, evict, dst, lane);
            // the vacated slot's entry names the path that leaves it (the wake launch continues from there): the moved path may
            // have started within this very visit and never been stored here
            if (evict) wf.sd(slot) = make_uint4(0u, 0u, (uint32_t) s.path_i, (uint32_t) (s.path_i >> 32));
        }
        if (has && cont) {
            if (!(s.flags & kFlagTermPending)) {
                // resolved rays carry their final hit; the others start wf_trace from the rectangle hit
                wf.hit(dst) = make_float4(hit.t, hit.u, hit.v, __int_as_float(hit.slot));
                wf.hit_prim(dst) = hit.prim;
            }
            store_state(wf, dst, receive, s);
            ++c_live;
            if (shadowing) {
                wf.sh0(dst) = make_float4(sh.o.x, sh.o.y, sh.o.z, sh.mint);
                wf.sh1(dst) = make_float4(sh.d.x, sh.d.y, sh.d.z, sh.maxt);
                wf.sh2(dst) = sh.c;
                if (receive && lp.iq) wf.sh3(dst) = sh.c_im;
            }
        } else if (has && rolling && touched) {
            // the slot has run out of paths for now: remember the last one it rendered, so that the wake launch of the
            // sequence's next call continues from there (a path that started and ended within this visit was never stored)
            wf.sd(slot) = make_uint4(0u, 0u, (uint32_t) s.path_i, (uint32_t) (s.path_i >> 32));
        }
        cont = has && cont;
        tracing = cont && tracing;
        shadowing = cont && shadowing;
        publish_masks(m_alive, aligned, FIRST == 2, batch0, slot, has, cont && !evict);
        publish_masks(m_trace, aligned, FIRST == 2, batch0, slot, has, tracing && !evict);
        publish_masks(m_shadow, aligned, FIRST == 2, batch0, slot, has, shadowing && !evict);
        // the hit is final and real (answered by presolve_ray; wf_trace sets the bit for the rays it answers)
        const bool hit_known = cont && !tracing && !(s.flags & kFlagTermPending) && hit.t != BF_INF;
        if (wf.hit_split) publish_masks(m_hit, aligned, FIRST == 2, batch0, slot, has, hit_known && !evict);
        if (evict) {                 // the moved path's bits go to its new slot (the batch's owner ORs its own in as well)
            const unsigned long long bit = 1ull << (dst & 63u);
            atomicOr(&m_alive[dst >> 6], bit);
            if (tracing) atomicOr(&m_trace[dst >> 6], bit);
            if (shadowing) atomicOr(&m_shadow[dst >> 6], bit);
            if (wf.hit_split && hit_known) atomicOr(&m_hit[dst >> 6], bit);
        }
        c_traced += (tracing ? 1u : 0u) + (shadowing ? 1u : 0u);
        c_shq += shadowing ? 1u : 0u;
        SLT(6);
    }
    }
#endif

    film_flush<RX>(lp, acc, s_hist, g_hist, lds_hist, tid);
#ifdef BF_SHADE_PROF
    // (the stamps inside shade_vertex sit in divergent code: a lane holds the cycles of the sections IT went through; the
    // wave's figure is the largest)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        spf.si = max(spf.si, (unsigned long long) __shfl_down((long long) spf.si, off));
        spf.head = max(spf.head, (unsigned long long) __shfl_down((long long) spf.head, off));
        spf.nee = max(spf.nee, (unsigned long long) __shfl_down((long long) spf.nee, off));
        spf.bsdf = max(spf.bsdf, (unsigned long long) __shfl_down((long long) spf.bsdf, off));
    }
    if (lane == 0) {
        for (int k = 0; k < 8; ++k) atomicAdd(&g_lane_prof[24 + k], t_sec[k]);      // sections 24..31 of the wave-entry half: cycles
        // inside "vertex + film": surface interaction (incl. the wait for the triangle), head, next-event estimation, BSDF sampling
        atomicAdd(&g_lane_prof[kShadeProfSections + 24], spf.si);
        atomicAdd(&g_lane_prof[kShadeProfSections + 25], spf.head);
        atomicAdd(&g_lane_prof[kShadeProfSections + 26], spf.nee);
        atomicAdd(&g_lane_prof[kShadeProfSections + 27], spf.bsdf);
    }
#endif
    unsigned long long v_closest = c_closest, v_shadow = c_shadow, v_invalid = acc.invalid, v_bounces = c_bounces;
    uint32_t v_live = c_live, v_film = acc.n_put;
    unsigned long long v_traced = c_traced;
    uint32_t v_loads = c_loads, v_shq = c_shq;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v_loads += __shfl_down(v_loads, off);
        v_shq += __shfl_down(v_shq, off);
        v_traced += __shfl_down(v_traced, off);
        v_closest += __shfl_down(v_closest, off);
        v_shadow += __shfl_down(v_shadow, off);
        v_invalid += __shfl_down(v_invalid, off);
        v_bounces += __shfl_down(v_bounces, off);
        v_live += __shfl_down(v_live, off);
        v_film += __shfl_down(v_film, off);
    }
    // the live count steers the host (never optional): one atomic per WORKGROUP
    __shared__ uint32_t s_live[kBlock / 64];
    if (lane == 0) s_live[tid >> 6] = v_live;
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0;
        for (int k = 0; k < kBlock / 64; ++k) t += s_live[k];
        if (t) atomicAdd(&wf.n_live[it], t);
    }
    if (lane == 0 && lp.count) {
        if (v_film) atomicAdd(&wf.counters[CTR_FILM], (unsigned long long) v_film);
        if (v_closest) atomicAdd(&wf.counters[CTR_CLOSEST], v_closest);
        if (v_shadow) atomicAdd(&wf.counters[CTR_SHADOW], v_shadow);
        if (v_invalid) atomicAdd(&wf.counters[CTR_INVALID], v_invalid);
        if (v_bounces) atomicAdd(&wf.counters[CTR_BOUNCES], v_bounces);
        if (v_traced) atomicAdd(&wf.counters[CTR_TRACED], v_traced);
        if (v_loads) atomicAdd(&wf.counters[CTR_SHADE_LOADS], (unsigned long long) v_loads);
        if (v_live) atomicAdd(&wf.counters[CTR_SHADE_STORES], (unsigned long long) v_live);
        if (v_shq) atomicAdd(&wf.counters[CTR_SHADE_SHADOW], (unsigned long long) v_shq);
        if (v_closest + v_shadow) atomicAdd(&wf.counters[CTR_SHADE_RAYS], v_closest + v_shadow);
    }
}

// Persistent-wave traversal with dynamic ray replacement.  Each wave owns a
// contiguous segment of batches and walks first their shadow masks (any-hit
// rays) and then their trace masks (closest-hit rays); a lane whose ray has
// finished receives the next set bit as soon as the wave's occupancy drops to
// kRefill lanes (MaskCursor: __popcll / n-th-set-bit select, no atomics), so the
// wave's cost tracks the SUM of its rays' traversal steps instead of 64 x the
// longest one, and sparse pools still traverse with full waves.
//
// "while-while" form: lanes first descend through internal nodes together
// (postponing the leaf they reach), then all lanes holding a leaf intersect
// its triangles together, so the triangle code never runs for one straggler
// while 63 lanes wait at nodes.
//
// Stack: the first kLdsStack entries live in LDS (lane-strided, conflict-free),
// deeper entries — rare: one entry per tree level where BOTH children are hit —
// spill to a per-thread column in HBM.  16 KiB of LDS per workgroup instead of
// 32 lifts the kernel from 5 to 8 waves/SIMD.
constexpr int kLdsStack = 16;
// BF_TRACE_IFIF = 1: the "if-if" form of the traversal loop (one step for every lane per iteration, node and triangle fetches in
// flight together) instead of "while-while".  Built in round 4 as the structural experiment on this kernel, parity-green (216 GPU
// tests) and 20 % SLOWER (C2 wf_trace 3.79 -> 4.54 ms per step, C5 9.15 -> 10.25, profiles/r04_trace_ifif_ab.txt): every lane
// runs the node AND the leaf code every iteration and the unified loads move more bytes through the texture pipeline, which
// is this kernel's busiest unit — so it stays off.
#ifndef BF_TRACE_IFIF
#define BF_TRACE_IFIF 0
#endif
constexpr uint32_t kTraceGuard = 1u << 24;      // wf_trace: inner-loop iterations between two refills (see the guard below)

template <bool STATS, int W, bool SHIFT, bool QUANT>
__global__ __launch_bounds__(kBlock, W) void wf_trace(DScene sc, WF wf, uint32_t it) {
    __shared__ int s_stack[kLdsStack * kBlock];
    __shared__ float4 s_top[kTopNodes * (QUANT ? kTopStrideQ : kTopStride)];
    int *stack = s_stack + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int nxt = (it & 1) ^ 1;
    uint32_t c_nodes = 0, c_tris = 0, c_top = 0;
    const int n_top = (int) min(sc.n_nodes, kTopNodes);
    if (QUANT)
        load_top_qnodes(sc.qnodes, (uint32_t) n_top, s_top, threadIdx.x, kBlock);
    else
        load_top_nodes(sc.nodes, (uint32_t) n_top, s_top, threadIdx.x, kBlock);
    __syncthreads();
    const uint32_t n_waves = gridDim.x * (kBlock / 64), wave_id = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    LaneStack<kLdsStack, true> st = make_stack<kLdsStack, true>(sc, stack);
    const uint32_t n_batches = wf.n_slots >> 6;
    MaskCursor cursor;
    cursor_init(cursor, wf.m_shadow[nxt], wave_id, n_waves, n_batches, lane);
    bool phase_shadow = true;       // wave-uniform: which job list the cursor walks

    bool has = false, any = false;
    uint32_t job = 0;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1), id = mk(0, 0, 0), oid = mk(0, 0, 0), ohi = mk(0, 0, 0);
    Shift shf = no_shift();          // SHIFT: batched launch with moving meshes (the ray's render selects the offset)
    float mint = 0.f, maxt = 0.f;
    Hit best;
    best.t = BF_INF;
    best.u = best.v = 0.f;
    best.prim = 0;
    best.slot = 0;
    int node = kNoNode;
    bool found = false;      // any-hit result
    uint32_t guard = 0;

    auto work_left = [&]() -> bool { return phase_shadow || !cursor_empty(cursor); };

    while (true) {
        // ---- refill idle lanes from the wave's segment -------------------------
        unsigned long long idle = __ballot(!has);
        if (idle) {
            uint32_t want = (uint32_t) __popcll(idle);
            const uint32_t rank = (uint32_t) __popcll(idle & ((1ull << lane) - 1ull));
            uint32_t served = 0;
            while (want && work_left()) {
                uint32_t slot = 0;
                const bool req = !has && rank >= served;
                uint32_t got = cursor_take(cursor, want, req, rank - served, slot, lane);
                if (req && rank - served < got) {
                    job = slot;
                    float4 r0, r1;
                    if (phase_shadow) {
                        r0 = wf.sh0(slot);
                        r1 = wf.sh1(slot);
                        any = true;
                    } else {
                        r0 = wf.ray0(slot);
                        r1 = wf.ray1(slot);
                        any = false;
                    }
                    o = mk(r0.x, r0.y, r0.z);
                    d = mk(r1.x, r1.y, r1.z);
                    mint = r0.w;
                    maxt = r1.w;
                    found = false;
                    has = true;
                    best.t = BF_INF;
                    best.u = best.v = 0.f;
                    best.prim = 0;
                    best.slot = 0;
                    if (!phase_shadow) {
                        // closest-hit rays continue from the rectangle hit wf_shade found (presolve_ray)
                        float4 hq = wf.hit(slot);
                        best.t = hq.x;
                        best.u = hq.y;
                        best.v = hq.z;
                        best.slot = __float_as_int(hq.w);
                        best.prim = wf.hit_prim(slot);
                    }
                    if (SHIFT) {
                        shf = make_shift(wf.offsets, wf.render(slot), wf.box_slack);
                        ray_inverse_shift(o, d, shf, id, oid, ohi);
                    } else {
                        ray_inverse(o, d, id, oid);
                    }
                    node = sc.root;
                    st.sp = 0;
                }
                served += got;
                want -= got;
                if (want && cursor_empty(cursor) && phase_shadow) {
                    phase_shadow = false;                       // shadow rays done: closest-hit rays next
                    cursor_init(cursor, wf.m_trace[nxt], wave_id, n_waves, n_batches, lane);
                }
            }
        }
        if (__ballot(has) == 0ull) break;
        guard = 0;

        // ---- traversal until the wave thins out -----------------------------------
        while (true) {
            if (++guard > kTraceGuard) {
                // Safety net: a persistent wave must always drain.  The count restarts at every refill, so it bounds the
                // steps ONE set of rays may take (a ray visits each node and leaf at most once: far below the bound).  It
                // is never expected to trip; if it does, the rays are dropped LOUDLY: CTR_GUARD makes bf_render_device
                // fail with BF_ERR_DEVICE (bf_stats.n_guard) instead of returning a plausible histogram.
                const unsigned long long lost = __ballot(has);
                if (lane == 0) atomicAdd(&wf.counters[CTR_GUARD], (unsigned long long) __popcll(lost));
                has = false;
                break;
            }
#if BF_TRACE_IFIF
            if (!QUANT) {
                // "if-if": ONE step for every lane that holds a ray — a lane at an internal node fetches the node (LDS copy
                // or memory), a lane at a leaf its (one or two) triangles, into the same registers; the wave waits ONCE for
                // both kinds, then the node lanes decide and the leaf lanes intersect.  Every lane advances every iteration
                // and node and triangle fetches are in flight together (round 4: the kernel waits on dependent fetches with
                // 45 % of its lanes busy — in the while-while form below a lane at a leaf idles through the others' node steps)
                const bool at_node = has && node >= 0, at_leaf = has && node < 0 && node != kNoNode;
                // ONE generic pointer per lane — node record in memory, node record in the LDS copy of the tree's top, or the
                // leaf's first triangle — and the same seven loads for all of them (flat loads: a branch per source would give
                // every source its own destination registers, and the moves that merge them wait for the loads inside
                // their branch)
                const uint32_t enc = ~(uint32_t) node;
                const uint32_t l_first = enc >> 3, l_cnt = (enc & 7u) + 1u;
                const float4 *qp = sc.tris + kTriStride * l_first;
                if (at_node) qp = node < n_top ? (const float4 *) (s_top + kTopStride * (uint32_t) node) : sc.nodes + 8u * (uint32_t) node;
                // (unconditional: a leaf lane reads past its one or two triangles — the triangle array carries 64 bytes of padding
                // for the last one, bf_api.cpp — and a lane without a ray re-reads the root: straight-line loads are issued back
                // to back and waited for once)
                if (!at_node && !at_leaf) qp = sc.tris;
                const float4 q0 = qp[0], q1 = qp[1], q2 = qp[2], q3 = qp[3], q4 = qp[4], q5 = qp[5];
                float4 q6 = qp[6];
                // (keeps the child references' load up here with the others: the compiler would sink it into the node branch — a
                // second dependent round trip per node step)
                asm volatile("" : "+v"(q6.x), "+v"(q6.y), "+v"(q6.z), "+v"(q6.w));
                if (at_node) {
                    if (STATS) {
                        ++c_nodes;
                        c_top += node < n_top ? 1u : 0u;
                    }
                    node = node4_decide(q0, q1, q2, q3, q4, q5, q6, id, oid, SHIFT ? ohi : oid, mint, any ? maxt : __builtin_fminf(maxt, best.t), st);
                } else if (at_leaf) {
                    float t, u, v;
                    found = false;
                    if (STATS) c_tris += min(l_cnt, 2u);
                    if (tri_intersect(shifted(mk(q0.x, q0.y, q0.z), shf), shifted(mk(q1.x, q1.y, q1.z), shf), shifted(mk(q2.x, q2.y, q2.z), shf), o, d,
                                      mint, maxt, t, u, v)) {
                        found = any;
                        consider(best, t, u, v, __float_as_uint(q0.w), (int32_t) l_first);
                    }
                    if (l_cnt > 1u && !found &&
                        tri_intersect(shifted(mk(q3.x, q3.y, q3.z), shf), shifted(mk(q4.x, q4.y, q4.z), shf), shifted(mk(q5.x, q5.y, q5.z), shf), o, d,
                                      mint, maxt, t, u, v)) {
                        found = any;
                        consider(best, t, u, v, __float_as_uint(q3.w), (int32_t) (l_first + 1u));
                    }
                    for (uint32_t i = 2; i < l_cnt && !found; ++i) {          // (the builder's leaves hold at most two triangles)
                        const float4 *tp = sc.tris + kTriStride * (l_first + i);
                        const float4 a = tp[0], b = tp[1], c = tp[2];
                        if (STATS) ++c_tris;
                        if (tri_intersect(shifted(mk(a.x, a.y, a.z), shf), shifted(mk(b.x, b.y, b.z), shf), shifted(mk(c.x, c.y, c.z), shf), o, d, mint,
                                          maxt, t, u, v)) {
                            found = any;
                            consider(best, t, u, v, __float_as_uint(a.w), (int32_t) (l_first + i));
                        }
                    }
                    node = found ? kNoNode : st.pop_or_none();
                }
            } else
#endif
            {
            // (a) descend through internal nodes; a lane that reaches a leaf (node < 0) waits.
            // Stop descending once fewer than kStragglers lanes are still at internal nodes:
            // they resume after the others' leaves have been intersected.
            while (true) {
                const unsigned long long at_node = __ballot(has && node >= 0);
                if (!at_node) break;
                // progress guarantee: only postpone the stragglers if some lane has a leaf to intersect
                if ((uint32_t) __popcll(at_node) < wf.trace_stragglers && __ballot(has && node < 0 && node != kNoNode)) break;
                if (has && node >= 0) {
                    if (STATS) {
                        ++c_nodes;
                        c_top += node < n_top ? 1u : 0u;
                    }
                    if (QUANT)
                        node = node4q_step_top(sc.qnodes, s_top, n_top, node, id, oid, SHIFT ? ohi : oid, mint,
                                               any ? maxt : __builtin_fminf(maxt, best.t), st);
                    else
                        node = node4_step_top(sc.nodes, s_top, n_top, node, id, oid, SHIFT ? ohi : oid, mint,
                                              any ? maxt : __builtin_fminf(maxt, best.t), st);
                }
            }
            // (b) intersect the postponed leaves together
            if (has && node < 0 && node != kNoNode) {
                found = SHIFT ? leaf_intersect<STATS>(sc, node, any, o, d, mint, maxt, best, c_tris, shf)
                              : leaf_intersect<STATS>(sc, node, any, o, d, mint, maxt, best, c_tris);
                node = found ? kNoNode : st.pop_or_none();
            }
            }
            // (c) retire finished rays
            if (has && node == kNoNode) {
                if (any) {
                    // Scene::ray_test resolved: an unoccluded shadow ray releases its NEE contribution
                    // (an occluded sample contributes c * 0, scene.cpp:220-224: only a non-finite c leaves a trace)
                    const float c = wf.sh2(job);
                    if (!found || !__builtin_isfinite(c)) {
                        float4 a = wf.sa(job);
                        a.w += found ? c * 0.f : c;
                        wf.sa(job) = a;
                    }
                    if (wf.iq) {                     // BF_MODE_RECEIVE_IQ: imaginary accumulator lives in se.w
                        const float ci = wf.sh3(job);
                        if (!found || !__builtin_isfinite(ci)) {
                            float4 e = wf.se(job);
                            e.w += found ? ci * 0.f : ci;
                            wf.se(job) = e;
                        }
                    }
                } else {
                    wf.hit(job) = make_float4(best.t, best.u, best.v, __int_as_float(best.slot));
                    if (wf.hit_split && best.t != BF_INF) atomicOr(&wf.m_hit[nxt][job >> 6], 1ull << (job & 63u));
                }
                has = false;
            }
            unsigned long long act = __ballot(has);
            if (act == 0ull) break;
            if (work_left() && (uint32_t) __popcll(act) <= wf.trace_refill) break;
        }
    }
    if (STATS) {
        unsigned long long v_nodes = c_nodes, v_tris = c_tris, v_top = c_top;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            v_nodes += __shfl_down(v_nodes, off);
            v_tris += __shfl_down(v_tris, off);
            v_top += __shfl_down(v_top, off);
        }
        if (lane == 0) {
            atomicAdd(&wf.counters[CTR_NODES], v_nodes);
            atomicAdd(&wf.counters[CTR_TRIS], v_tris);
            atomicAdd(&wf.counters[CTR_NODES_LDS], v_top);
        }
    }
}

}  // namespace bfd

#ifdef BF_SHADE_PROF
// developer build: wave entries [0..31] (24..31: cycles between stamps) and lanes [32..63] per section of wf_shade
extern "C" int bfdbg_shade_lane_profile(unsigned long long *out, int clear) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(bfd::g_lane_prof), sizeof(unsigned long long) * 2 * bfd::kShadeProfSections) != hipSuccess) return -1;
    if (clear) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(bfd::g_lane_prof)) != hipSuccess) return -1;
        if (hipMemset(p, 0, sizeof(unsigned long long) * 2 * bfd::kShadeProfSections) != hipSuccess) return -1;
    }
    return 2 * bfd::kShadeProfSections;
}
#endif

extern "C" hipError_t bfk_wf_shade(const bfd::DScene *sc, const bfd::DLaunch *lp, const bfd::WF *wf, uint32_t it, int first,
                                   float *g_hist, bf_path_record *records, unsigned grid, size_t lds_bytes,
                                   hipStream_t stream, int waves) {
    // `waves`: register budget of the shading kernel (waves per SIMD); 3 is the sweet spot (168 VGPRs)
    const bool rx = lp->mode == BF_MODE_RECEIVE_RAW;
#define BF_SHADE_LAUNCH_RX(F, W, RX_)                                                                                              \
    hipLaunchKernelGGL((bfd::wf_shade<F, W, RX_>), dim3(grid), dim3(bfd::kBlock), lds_bytes, stream, *sc, *lp, *wf, it, g_hist, records)
#define BF_SHADE_LAUNCH(F, W)            \
    if (rx) BF_SHADE_LAUNCH_RX(F, W, 1); \
    else BF_SHADE_LAUNCH_RX(F, W, 0)
#define BF_SHADE_LAUNCH_WIDE(F)                          \
    if (rx) BF_SHADE_LAUNCH_RX(F, 3, 1 | bfd::kWide);    \
    else BF_SHADE_LAUNCH_RX(F, 3, 0 | bfd::kWide)
#define BF_SHADE_LAUNCH_LEAN(F)                          \
    if (rx) BF_SHADE_LAUNCH_RX(F, 3, 1 | bfd::kLean);    \
    else BF_SHADE_LAUNCH_RX(F, 3, 0 | bfd::kLean)
#define BF_SHADE_LAUNCH_MULTI(F)                         \
    if (rx) BF_SHADE_LAUNCH_RX(F, 3, 1 | bfd::kMulti);   \
    else BF_SHADE_LAUNCH_RX(F, 3, 0 | bfd::kMulti)
    if (lp->multi) {
        // the sequence's endpoints moved between its renders: per-path tables (bf_device.h: kMulti; general kernels, box filter)
        if (first == 3) { BF_SHADE_LAUNCH_MULTI(3); }
        else if (first == 2) { BF_SHADE_LAUNCH_MULTI(2); }
        else if (first) { BF_SHADE_LAUNCH_MULTI(1); }
        else { BF_SHADE_LAUNCH_MULTI(0); }
    } else if (lp->lean && !lp->wide && (first || waves == 3)) {
        // scene and launch fit the lean profile (bf_device.h: kLean; three waves per SIMD only)
        if (first == 3) { BF_SHADE_LAUNCH_LEAN(3); }
        else if (first == 2) { BF_SHADE_LAUNCH_LEAN(2); }
        else if (first) { BF_SHADE_LAUNCH_LEAN(1); }
        else { BF_SHADE_LAUNCH_LEAN(0); }
    } else if (lp->wide) {
        // reconstruction filter wider than a pixel: the kWide variants (three waves per SIMD only)
        if (first == 3) { BF_SHADE_LAUNCH_WIDE(3); }
        else if (first == 2) { BF_SHADE_LAUNCH_WIDE(2); }
        else if (first) { BF_SHADE_LAUNCH_WIDE(1); }
        else { BF_SHADE_LAUNCH_WIDE(0); }
    } else if (first == 3) {
        BF_SHADE_LAUNCH(3, 3);
    } else if (first == 2) {
        BF_SHADE_LAUNCH(2, 3);
    } else if (first) {
        BF_SHADE_LAUNCH(1, 3);
    } else {
        switch (waves) {
            case 2: BF_SHADE_LAUNCH(0, 2); break;
            case 3: BF_SHADE_LAUNCH(0, 3); break;
            case 4: BF_SHADE_LAUNCH(0, 4); break;
            default: BF_SHADE_LAUNCH(0, 1); break;
        }
    }
#undef BF_SHADE_LAUNCH
#undef BF_SHADE_LAUNCH_WIDE
#undef BF_SHADE_LAUNCH_LEAN
#undef BF_SHADE_LAUNCH_MULTI
#undef BF_SHADE_LAUNCH_RX
    return hipGetLastError();
}

extern "C" hipError_t bfk_wf_trace(const bfd::DScene *sc, const bfd::WF *wf, uint32_t it, int stats, unsigned grid,
                                   hipStream_t stream, int waves) {
    const bool shift = wf->offsets != nullptr, quant = sc->qnodes != nullptr;
#define BF_TRACE_LAUNCH2(S, W, SH, Q) \
    hipLaunchKernelGGL((bfd::wf_trace<S, W, SH, Q>), dim3(grid), dim3(bfd::kBlock), 0, stream, *sc, *wf, it)
#define BF_TRACE_LAUNCH1(S, W)                      \
    if (shift) {                                    \
        if (quant) BF_TRACE_LAUNCH2(S, W, true, true);  \
        else BF_TRACE_LAUNCH2(S, W, true, false);   \
    } else {                                        \
        if (quant) BF_TRACE_LAUNCH2(S, W, false, true); \
        else BF_TRACE_LAUNCH2(S, W, false, false);  \
    }
#define BF_TRACE_LAUNCH(W)   \
    if (stats) {             \
        BF_TRACE_LAUNCH1(true, W)  \
    } else {                 \
        BF_TRACE_LAUNCH1(false, W) \
    }
    // 28.6 KiB of LDS per workgroup (stacks + the tree's top levels): five workgroups per CU is the most that fit
    // (the quantised nodes' copy is 6.8 KiB: six workgroups per CU fit, if the register allocation follows)
    if (waves >= 6 && quant) {
        if (stats) {
            if (shift) BF_TRACE_LAUNCH2(true, 6, true, true);
            else BF_TRACE_LAUNCH2(true, 6, false, true);
        } else {
            if (shift) BF_TRACE_LAUNCH2(false, 6, true, true);
            else BF_TRACE_LAUNCH2(false, 6, false, true);
        }
    } else if (waves >= 5) {
        BF_TRACE_LAUNCH(5);
    } else {
        BF_TRACE_LAUNCH(4);
    }
#undef BF_TRACE_LAUNCH
#undef BF_TRACE_LAUNCH1
#undef BF_TRACE_LAUNCH2
    return hipGetLastError();
}
