// Wavefront path tracer for gfx950: the radar hot path split into two
// persistent kernels per bounce, with the path state streamed through HBM.
//
//   wf_shade : one lane per live path slot.  Reads the slot's state + closest
//              hit (coalesced, queue order), runs the integrator's vertex logic
//              (emitter hit, Russian roulette, next-event estimation, BSDF
//              sampling — path.cpp:121-209 and the pathlength / pathtime
//              variants), bins finished paths into the LDS-privatised range
//              histogram, REGENERATES finished slots with fresh paths from the
//              global path counter, and writes survivors compacted
//              (__ballot/__popcll prefix + one atomic per wave) into the next
//              queue together with their next ray; shadow rays go to a second
//              compacted queue carrying the NEE contribution they gate.
//   wf_trace : persistent waves pull batches of 64 rays (shadow rays first,
//              then closest-hit rays) from the queues and traverse the BVH with
//              per-lane LDS stacks; closest hits are written in queue order,
//              unoccluded shadow rays add their contribution to the path.
//
// The fat shading code (fp64 transcendentals, > 256 registers) and the lean
// traversal code (58 VGPRs) no longer share one register allocation, so the
// latency-bound traversal runs at 5 waves/SIMD instead of 1, and every lane
// of a trace wave holds a ray.  Results are bit-identical per path to the
// megakernel (bf_kernels.hip) and to the oracle: same draws, same arithmetic.
#include "bf_path_logic.h"

namespace bfd {

template <bool FIRST, int W>
__global__ __launch_bounds__(kBlock, W) void wf_shade(DScene sc, DLaunch lp, WF wf, uint32_t it, float *__restrict__ g_hist,
                                                   bf_path_record *__restrict__ records) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    float *s_hist = reinterpret_cast<float *>(s_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const bool lds_hist = lp.lds_hist != 0;
    if (lds_hist) {
        for (uint32_t i = tid; i < lp.n_chan; i += kBlock) s_hist[i] = 0.f;
        __syncthreads();
    }
    const int cur = it & 1, nxt = cur ^ 1;
    const uint32_t n_cur = wf.n_q[it];
    const bool receive = lp.mode == BF_MODE_RECEIVE_RAW;

    FilmAcc acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0u};
    uint32_t c_closest = 0, c_shadow = 0, c_bounces = 0, c_started = 0;

    // static wave-granular partition of the queue: a device-wide queue head
    // saturates at ~88 dequeues/us on MI355X (MI355X_MICROARCH.md "dequeue"),
    // which throttled 64-slot batches; batches are plentiful per wave, so a
    // grid-stride walk balances well without any atomic.
    const uint32_t n_waves = gridDim.x * (kBlock / 64), wave_id = blockIdx.x * (kBlock / 64) + (tid >> 6);
    for (uint32_t base = wave_id * 64u; base < n_cur; base += n_waves * 64u) {
        const uint32_t i = base + lane;
        const bool has = i < n_cur;

        PathState s;
        ShadowReq sh;
        sh.want = false;
        bool need_gen = false, cont = false;

        if (has) {
            if (FIRST) {
                need_gen = true;
            } else {
                load_state(wf, cur, i, receive, s);
                if (s.flags & kFlagTermPending) {
                    // ended after last bounce's BSDF sample; its NEE shadow ray has resolved by now
                    film_put(sc, lp, s, s_hist, g_hist, lds_hist, acc, records);
                    need_gen = true;
                } else {
                    float4 hq = wf.hit[i];
                    Hit hit;
                    hit.t = hq.x;
                    hit.u = hq.y;
                    hit.v = hq.z;
                    hit.slot = __float_as_int(hq.w);
                    hit.prim = 0;
                    cont = shade_vertex(sc, lp, s, hit, sh, c_bounces);
                    if (!cont) {
                        film_put(sc, lp, s, s_hist, g_hist, lds_hist, acc, records);
                        need_gen = true;
                    } else if (!(s.flags & kFlagTermPending)) {
                        ++c_closest;
                    }
                    if (sh.want) ++c_shadow;
                }
            }
        }

        // ---- regeneration: finished slots pull fresh paths -------------------
        unsigned long long gmask = __ballot(need_gen);
        if (gmask) {
            unsigned long long pbase = 0;
            if (lane == 0) pbase = atomicAdd(&wf.counters[CTR_NEXT_PATH], (unsigned long long) __popcll(gmask));
            pbase = __shfl(pbase, 0);
            if (need_gen) {
                uint64_t path_i = pbase + __popcll(gmask & ((1ull << lane) - 1ull));
                if (path_i < lp.n_paths) {
                    generate_path(sc, lp, path_i, s);
                    ++c_closest;
                    ++c_started;
                    cont = true;
                }
            }
        }

        // ---- compaction: survivors -> next queue, shadow rays -> shadow queue --
        unsigned long long cmask = __ballot(cont);
        uint32_t j = 0;
        if (cmask) {
            uint32_t qb = 0;
            if (lane == 0) qb = atomicAdd(&wf.n_q[it + 1], (uint32_t) __popcll(cmask));
            qb = __shfl(qb, 0);
            j = qb + __popcll(cmask & ((1ull << lane) - 1ull));
            if (cont) store_state(wf, nxt, j, receive, s);
        }
        unsigned long long smask = __ballot(sh.want);
        if (smask) {
            uint32_t sb = 0;
            if (lane == 0) sb = atomicAdd(&wf.n_sh[it], (uint32_t) __popcll(smask));
            sb = __shfl(sb, 0);
            if (sh.want) {
                uint32_t k = sb + __popcll(smask & ((1ull << lane) - 1ull));
                wf.sh0[k] = make_float4(sh.o.x, sh.o.y, sh.o.z, sh.mint);
                wf.sh1[k] = make_float4(sh.d.x, sh.d.y, sh.d.z, sh.maxt);
                wf.sh2[k] = make_uint2(j, __float_as_uint(sh.c));
            }
        }
    }

    film_flush(lp, acc, s_hist, g_hist, lds_hist, tid);
    unsigned long long v_closest = c_closest, v_shadow = c_shadow, v_invalid = acc.invalid, v_bounces = c_bounces,
                       v_started = c_started;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v_started += __shfl_down(v_started, off);
        v_closest += __shfl_down(v_closest, off);
        v_shadow += __shfl_down(v_shadow, off);
        v_invalid += __shfl_down(v_invalid, off);
        v_bounces += __shfl_down(v_bounces, off);
    }
    if (lane == 0) {
        if (v_closest) atomicAdd(&wf.counters[CTR_CLOSEST], v_closest);
        if (v_shadow) atomicAdd(&wf.counters[CTR_SHADOW], v_shadow);
        if (v_invalid) atomicAdd(&wf.counters[CTR_INVALID], v_invalid);
        if (v_bounces) atomicAdd(&wf.counters[CTR_BOUNCES], v_bounces);
        if (v_started) atomicAdd(&wf.counters[CTR_STARTED], v_started);   // host: supply exhausted iff == n_paths
    }
}

// Persistent-wave traversal with dynamic ray replacement: each wave owns a
// contiguous segment of the job list (shadow rays first, then closest-hit
// rays); a lane whose ray has finished takes the next job of the segment as
// soon as the wave's occupancy drops below kRefill lanes (__ballot/__popcll
// prefix, no atomics), so the wave's cost tracks the SUM of its rays'
// traversal steps instead of 64 x the longest one.
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
constexpr int kRefill = 44;
constexpr int kLdsStack = 16;
constexpr int kNoNode = INT32_MIN;

template <bool STATS>
__global__ __launch_bounds__(kBlock, 8) void wf_trace(DScene sc, WF wf, uint32_t it) {
    __shared__ int s_stack[kLdsStack * kBlock];
    int *stack = s_stack + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int nxt = (it & 1) ^ 1;
    const uint32_t n_sh = wf.n_sh[it], n_ext = wf.n_q[it + 1];
    const uint32_t total = n_sh + n_ext;
    uint32_t c_nodes = 0, c_tris = 0;
    const uint32_t n_threads = gridDim.x * kBlock;
    const uint32_t n_waves = gridDim.x * (kBlock / 64), wave_id = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    int *spill = wf.spill + (blockIdx.x * kBlock + threadIdx.x);    // entry k at spill[k * n_threads]
    // segment per wave, multiple of 64 so that the first fetch of each wave is a coalesced 1 KiB read
    uint32_t seg = (uint32_t) (((uint64_t) total + n_waves - 1) / n_waves);
    seg = (seg + 63u) & ~63u;
    uint64_t sb64 = (uint64_t) wave_id * seg;
    uint32_t next = (uint32_t) (sb64 < total ? sb64 : total);
    const uint32_t seg_end = (uint32_t) ((sb64 + seg) < total ? (sb64 + seg) : total);

    bool has = false, any = false;
    uint32_t job = 0;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1), id = mk(0, 0, 0);
    float mint = 0.f, maxt = 0.f;
    Hit best;
    best.t = BF_INF;
    best.u = best.v = 0.f;
    best.prim = 0;
    best.slot = 0;
    int node = kNoNode, sp = 0;
    bool found = false;      // any-hit result

    auto push = [&](int v) {
        if (sp < kLdsStack)
            stack[sp * kBlock] = v;
        else
            spill[(size_t) (sp - kLdsStack) * n_threads] = v;
        ++sp;
    };
    auto pop = [&]() -> int {
        --sp;
        return sp < kLdsStack ? stack[sp * kBlock] : spill[(size_t) (sp - kLdsStack) * n_threads];
    };

    while (true) {
        // ---- refill idle lanes from the wave's segment -------------------------
        unsigned long long idle = __ballot(!has);
        if (idle && next < seg_end) {
            uint32_t k = next + (uint32_t) __popcll(idle & ((1ull << lane) - 1ull));
            next += (uint32_t) __popcll(idle);
            if (!has && k < seg_end) {
                job = k;
                float4 r0, r1;
                if (k < n_sh) {
                    r0 = wf.sh0[k];
                    r1 = wf.sh1[k];
                    any = true;
                } else {
                    r0 = wf.ray0[nxt][k - n_sh];
                    r1 = wf.ray1[nxt][k - n_sh];
                    any = false;
                }
                o = mk(r0.x, r0.y, r0.z);
                d = mk(r1.x, r1.y, r1.z);
                mint = r0.w;
                maxt = r1.w;
                best.t = BF_INF;
                best.u = best.v = 0.f;
                best.prim = 0;
                best.slot = 0;
                found = false;
                has = true;
                bool live = mint <= maxt;       // TERM_PENDING slots carry an empty interval
                if (live) {
                    for (uint32_t i = 0; i < sc.n_rects; ++i) {
                        const DRect &rc = sc.rects[i];
                        float t, lx, ly;
                        if (rect_intersect(rc, o, d, mint, maxt, t, lx, ly)) {
                            if (any)
                                found = true;
                            else
                                consider(best, t, lx, ly, rc.prim, -(int32_t) (i + 1));
                        }
                    }
                }
                id = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
                node = sc.root;
                sp = 0;
                if (!live || sc.n_tris == 0 || (any && found)) node = kNoNode;   // nothing to traverse
            }
        }
        if (__ballot(has) == 0ull) break;

        // ---- traversal until the wave thins out -----------------------------------
        while (true) {
            // (a) descend through internal nodes; a lane that reaches a leaf (node < 0) waits
            while (__ballot(has && node >= 0)) {
                if (has && node >= 0) {
                    const float4 *np = sc.nodes + 4u * (uint32_t) node;
                    float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
                    if (STATS) ++c_nodes;
                    float tmax = any ? maxt : __builtin_fminf(maxt, best.t);
                    float tn0, tn1;
                    bool h0 = slab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, id, mint, tmax, tn0);
                    bool h1 = slab(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, id, mint, tmax, tn1);
                    int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
                    if (h0 && h1) {
                        if (tn1 < tn0) {
                            int tmp = c0;
                            c0 = c1;
                            c1 = tmp;
                        }
                        push(c1);
                        node = c0;
                    } else if (h0) {
                        node = c0;
                    } else if (h1) {
                        node = c1;
                    } else {
                        node = sp ? pop() : kNoNode;
                    }
                }
            }
            // (b) every lane now holds a leaf or nothing: intersect the leaves together
            if (has && node != kNoNode) {
                uint32_t enc = ~(uint32_t) node;
                uint32_t first = enc >> 3, cnt = (enc & 7u) + 1u;
                for (uint32_t i = 0; i < cnt; ++i) {
                    const float4 *tp = sc.tris + 3u * (first + i);
                    float4 a = tp[0], b = tp[1], c = tp[2];
                    if (STATS) ++c_tris;
                    float t, u, v;
                    if (tri_intersect(mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), o, d, mint, maxt, t, u, v)) {
                        if (any) {
                            found = true;
                            break;
                        }
                        consider(best, t, u, v, __float_as_uint(a.w), (int32_t) (first + i));
                    }
                }
                node = ((any && found) || sp == 0) ? kNoNode : pop();
            }
            // (c) retire finished rays
            if (has && node == kNoNode) {
                if (any) {
                    // Scene::ray_test resolved: an unoccluded shadow ray releases its NEE contribution
                    if (!found) {
                        uint2 e = wf.sh2[job];
                        float4 a = wf.sa[nxt][e.x];
                        a.w += __uint_as_float(e.y);
                        wf.sa[nxt][e.x] = a;
                    }
                } else {
                    wf.hit[job - n_sh] = make_float4(best.t, best.u, best.v, __int_as_float(best.slot));
                }
                has = false;
            }
            unsigned long long act = __ballot(has);
            if (act == 0ull) break;
            if (next < seg_end && __popcll(act) <= kRefill) break;
        }
    }
    if (STATS) {
        unsigned long long v_nodes = c_nodes, v_tris = c_tris;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            v_nodes += __shfl_down(v_nodes, off);
            v_tris += __shfl_down(v_tris, off);
        }
        if (lane == 0) {
            atomicAdd(&wf.counters[CTR_NODES], v_nodes);
            atomicAdd(&wf.counters[CTR_TRIS], v_tris);
        }
    }
}

}  // namespace bfd

extern "C" hipError_t bfk_wf_shade(const bfd::DScene *sc, const bfd::DLaunch *lp, const bfd::WF *wf, uint32_t it, int first,
                                   float *g_hist, bf_path_record *records, unsigned grid, size_t lds_bytes,
                                   hipStream_t stream, int waves) {
    // `waves`: register budget of the shading kernel (waves per SIMD): 1 = no spills, 2..4 trade
    // scratch spills of the fp64 transcendental code for occupancy
#define BF_SHADE_LAUNCH(F, W)                                                                                             \
    hipLaunchKernelGGL((bfd::wf_shade<F, W>), dim3(grid), dim3(bfd::kBlock), lds_bytes, stream, *sc, *lp, *wf, it, g_hist, \
                       records)
    if (first) {
        BF_SHADE_LAUNCH(true, 1);
    } else {
        switch (waves) {
            case 2: BF_SHADE_LAUNCH(false, 2); break;
            case 3: BF_SHADE_LAUNCH(false, 3); break;
            case 4: BF_SHADE_LAUNCH(false, 4); break;
            default: BF_SHADE_LAUNCH(false, 1); break;
        }
    }
#undef BF_SHADE_LAUNCH
    return hipGetLastError();
}

extern "C" hipError_t bfk_wf_trace(const bfd::DScene *sc, const bfd::WF *wf, uint32_t it, int stats, unsigned grid,
                                   hipStream_t stream) {
    if (stats)
        hipLaunchKernelGGL(bfd::wf_trace<true>, dim3(grid), dim3(bfd::kBlock), 0, stream, *sc, *wf, it);
    else
        hipLaunchKernelGGL(bfd::wf_trace<false>, dim3(grid), dim3(bfd::kBlock), 0, stream, *sc, *wf, it);
    return hipGetLastError();
}
