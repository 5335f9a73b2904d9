// Hand-written gfx950 kernels for beifong's transient-radar hot path.
//
// One persistent wave64 megakernel with path regeneration: every lane owns one
// path at a time; a lane whose path ended pulls the next global path index
// from a device-wide queue head (wave-aggregated with __ballot/__popcll, one
// returning atomic per refill) and starts over, so all 64 lanes enter every
// BVH traversal.  Traversal keeps a per-lane stack in LDS (lane-strided, so a
// wave's push/pop is bank-conflict free), reads 64-B two-child nodes and 48-B
// triangles from HBM, and path-length returns are binned into an
// LDS-privatised histogram that is flushed with one global atomic per
// non-empty bin per workgroup.  No MFMA: the path is pointer chasing.
//
// Reference semantics (file:line) are cited at each function; the oracle
// (oracle/bf_oracle.cpp) restates the same functions independently on the CPU.
#include <cstring>

#include "bf_device_core.h"
#include "bf_wf_state.h"

namespace bfd {

// ---------------------------------------------------------------------------
// the render kernel
// ---------------------------------------------------------------------------
BF_DEV void hist_add(float *s_hist, float *g_hist, bool lds, uint32_t idx, float v) {
    if (lds)
        atomicAdd(&s_hist[idx], v);     // ds_add_f32
    else
        atomicAdd(&g_hist[idx], v);     // global_atomic_add_f32
}

// RESUME = true is the wavefront pipeline's TAIL kernel: once the path supply is
// exhausted and only a few thousand long paths remain, every lane adopts one
// slot of the wavefront queue (state + the closest hit already traced for it)
// and runs that path to completion here, instead of paying two launches per
// bounce for a nearly empty chip.
template <bool STATS, bool RESUME>
__global__ __launch_bounds__(kBlock) void bf_render_kernel(DScene sc, DLaunch lp, float *__restrict__ g_hist,
                                                           bf_path_record *__restrict__ records,
                                                           unsigned long long *__restrict__ counters, WF wf, uint32_t wf_it) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    int *s_stack = reinterpret_cast<int *>(s_raw);                         // [kStackDepth][kBlock]
    float *s_hist = reinterpret_cast<float *>(s_raw + sizeof(int) * kStackDepth * kBlock);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const bool lds_hist = lp.lds_hist != 0;
    if (lds_hist) {
        for (uint32_t i = tid; i < lp.n_chan; i += kBlock) s_hist[i] = 0.f;
        __syncthreads();
    }
    int *stack = s_stack + tid;

    const bool is_range = lp.mode == BF_MODE_RANGE, is_time = lp.mode == BF_MODE_TIME;
    const uint32_t n_emit = sc.n_emitters;
    const bool aperture = sc.sensor.type != BF_SENSOR_PERSPECTIVE;   // endpoint.h:241, perspective.cpp:130

    // per-lane path state
    bool alive = false, done = false;
    Rng rng;
    rng.state = 0;
    uint64_t path_i = 0;
    float throughput = 1.f, eta = 1.f, emission_weight = 1.f, result = 0.f, aux = 0.f, sensor_w = 1.f;
    int depth = 0;
    bool valid_ray = false, film_ok = true;
    uint32_t n_rays = 0;
    V3 ro = mk(0, 0, 0), rd = mk(0, 0, 1);
    float rmint = 0.f, rmaxt = 0.f;
    V3 prev_p = mk(0, 0, 0);
    float bs_pdf = 0.f;
    // per-lane accumulators of the five base channels X,Y,Z,alpha,weight
    float accX = 0.f, accY = 0.f, accZ = 0.f, accA = 0.f, accW = 0.f;
    // statistics
    uint32_t c_closest = 0, c_shadow = 0, c_nodes = 0, c_tris = 0, c_invalid = 0, c_bounces = 0;
    // wave-local pool of path indices (uniform across the wave)
    uint64_t pool_next = 0, pool_end = 0;

    Hit resume_hit;
    resume_hit.t = BF_INF;
    resume_hit.u = resume_hit.v = 0.f;
    resume_hit.prim = 0;
    resume_hit.slot = 0;
    bool resume_first = false, resume_film = false;
    if (RESUME) {
        done = true;
        const uint32_t n_cur = wf.n_q[wf_it];
        const uint32_t i = blockIdx.x * kBlock + tid;
        if (i < n_cur) {
            PathState s;
            load_state(wf, wf_it & 1, i, s);
            ro = s.ro;
            rd = s.rd;
            rmint = s.rmint;
            rmaxt = s.rmaxt;
            throughput = s.throughput;
            eta = s.eta;
            emission_weight = s.emission_weight;
            result = s.result;
            aux = s.aux;
            bs_pdf = s.bs_pdf;
            prev_p = s.prev_p;
            depth = (int) (s.flags & kDepthMask);
            valid_ray = (s.flags & kFlagValid) != 0;
            film_ok = (s.flags & kFlagFilmOk) != 0;
            n_rays = s.n_rays;
            rng = s.rng;
            path_i = s.path_i;
            sensor_w = sc.sensor.type == BF_SENSOR_FLUXMETER ? 1.f * kPi : 1.f;
            alive = true;
            if (s.flags & kFlagTermPending) {
                resume_film = true;        // only the film write is left
            } else {
                float4 hq = wf.hit[i];
                resume_hit.t = hq.x;
                resume_hit.u = hq.y;
                resume_hit.v = hq.z;
                resume_hit.slot = __float_as_int(hq.w);
                resume_first = true;
            }
        }
    }

    while (true) {
        // ---- 1. path regeneration: dead lanes pull new path indices -------
        unsigned long long need = __ballot(!alive && !done);
        if (need) {
            uint32_t n_need = __popcll(need);
            if (pool_end - pool_next < n_need && pool_end != ~0ull) {
                // refill: one returning atomic per wave for 4 x 64 paths
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(&counters[CTR_NEXT_PATH], 256ull);
                base = __shfl(base, 0);
                // hand out what is left of the old pool first is not worth the
                // bookkeeping: indices are only claimed, never skipped, because
                // the remainder of the old pool is consumed below before the new one
                if (pool_next < pool_end) {
                    // serve the old remainder to the first lanes, rest from the new chunk
                    uint32_t rank = __popcll(need & ((1ull << lane) - 1ull));
                    uint64_t left = pool_end - pool_next;
                    if (!alive && !done) {
                        path_i = rank < left ? pool_next + rank : base + (rank - left);
                    }
                    pool_next = base + (n_need - left);
                    pool_end = base + 256ull;
                } else {
                    uint32_t rank = __popcll(need & ((1ull << lane) - 1ull));
                    if (!alive && !done) path_i = base + rank;
                    pool_next = base + n_need;
                    pool_end = base + 256ull;
                }
            } else {
                uint32_t rank = __popcll(need & ((1ull << lane) - 1ull));
                if (!alive && !done) path_i = pool_next + rank;
                pool_next += n_need;
            }
            if (!alive && !done) {
                if (path_i >= lp.n_paths) {
                    done = true;
                } else {
                    // SamplingIntegrator::render_sample — integrator.cpp:259-283,
                    // per-path stream seed(base_seed + path) (sampler.cpp:83-96)
                    pcg_seed(rng, lp.seed + lp.path_offset + path_i);
                    float fx = next_1d(rng), fy = next_1d(rng);
                    float ax = .5f, ay = .5f;
                    if (aperture) {
                        ax = next_1d(rng);
                        ay = next_1d(rng);
                    }
                    if (sc.sensor.shutter_open_time > 0.f) (void) next_1d(rng);
                    (void) next_1d(rng);   // wavelength sample (consumed in RGB mode too)
                    sensor_w = sensor_sample_ray(sc, fx, fy, ax, ay, ro, rd, rmint, rmaxt);
                    // ImageBlock::put box branch: lo = ceil(pos - .5 - .5) must be 0
                    film_ok = __builtin_ceilf((fx - .5f) - .5f) == 0.f && __builtin_ceilf((fy - .5f) - .5f) == 0.f;
                    throughput = 1.f;
                    eta = 1.f;
                    emission_weight = 1.f;
                    result = 0.f;
                    aux = 0.f;
                    depth = 0;
                    n_rays = 0;
                    alive = true;
                }
            }
        }
        if (__ballot(alive) == 0ull) break;

        // ---- 2. closest-hit traversal for every live lane ------------------
        Hit hit;
        hit.t = BF_INF;
        if (RESUME && resume_first) {
            hit = resume_hit;              // traced by wf_trace already (and counted there)
        } else if (alive && !(RESUME && resume_film)) {
            traverse<false, STATS>(sc, ro, rd, rmint, rmaxt, stack, hit, c_nodes, c_tris);
            ++c_closest;
            ++n_rays;
        }
        resume_first = false;

        // ---- 3. vertex logic up to the shadow ray ---------------------------
        SI si;
        bool si_valid = false;
        bool terminate = false, want_shadow = false, nee = false;
        DirSample ds;
        ds.d = mk(0, 0, 1);
        ds.pdf = 0.f;
        ds.dist = 0.f;
        ds.delta = false;
        float emitter_val = 0.f;
        uint32_t mat_id = 0;
        if (RESUME && resume_film) {
            terminate = true;              // kFlagTermPending: straight to the film
        } else if (alive) {
            si_valid = hit.t != BF_INF;
            int emitter = -1;
            if (si_valid) {
                make_si(sc, ro, rd, hit, si);
                emitter = sc.shapes[si.shape].emitter;
            }
            if (depth == 0) {
                // first intersection — path.cpp:115-117, pathlength.cpp:138-146, pathtime.cpp:136-140
                valid_ray = si_valid;
                if (is_range) aux += si_valid ? si.t : 0.f;
                if (is_time) aux = si_valid ? si.t / lp.time_c : 0.f;
                depth = 1;
            } else {
                // tail of the previous iteration — path.cpp:184-209
                if (emitter >= 0) {
                    const DEmitter &e = sc.emitters[emitter];
                    float emitter_pdf = emitter_pdf_direction(sc, e, prev_p, si.p, si.sh.n);
                    if (n_emit != 1) emitter_pdf *= 1.f / (float) n_emit;
                    emission_weight = mis_weight(bs_pdf, emitter_pdf);
                }
                if (is_range) aux += si_valid ? si.t : 0.f;
                if (is_time) aux += si_valid ? si.t / lp.time_c : 0.f;
                ++depth;
            }
            // head of iteration `depth` — path.cpp:121-145
            if (emitter >= 0) {
                const DEmitter &e = sc.emitters[emitter];
                float ev = (e.type == BF_EMITTER_SPOT) ? 0.f : ((si.wi.z > 0.f) ? e.radiance : 0.f);
                result += emission_weight * throughput * ev;
                if (is_range) aux += si_valid ? si.t : 0.f;       // pathlength.cpp:161
            }
            bool active = si_valid;
            if (depth > lp.rr_depth) {
                float q = __builtin_fminf(throughput * sqr(eta), .95f);
                active = (next_1d(rng) < q) && active;
                throughput *= rcp(q);
            }
            if ((uint32_t) depth >= (uint32_t) lp.max_depth || !active) {
                terminate = true;
            } else {
                mat_id = sc.shapes[si.shape].material;
                const bf_material &mat = sc.materials[mat_id];
                ++c_bounces;
                nee = bsdf_smooth(mat);
                if (nee) {
                    // Scene::sample_emitter_direction — scene.cpp:180-230
                    float sx = next_1d(rng), sy = next_1d(rng);
                    if (n_emit == 0) {
                        ds.pdf = 0.f;
                        emitter_val = 0.f;
                    } else if (n_emit == 1) {
                        emitter_val = emitter_sample_direction(sc, sc.emitters[0], si.p, sx, sy, ds);
                    } else {
                        float emitter_pdf = 1.f / (float) n_emit;
                        uint32_t index = min((uint32_t) (sx * (float) n_emit), n_emit - 1u);
                        sx = (sx - index * emitter_pdf) * (float) n_emit;
                        emitter_val = emitter_sample_direction(sc, sc.emitters[index], si.p, sx, sy, ds);
                        ds.pdf *= emitter_pdf;
                        emitter_val *= rcp(emitter_pdf);
                    }
                    want_shadow = ds.pdf != 0.f;
                }
            }
        }

        // ---- 4. shadow (any-hit) traversal ----------------------------------
        if (__ballot(want_shadow)) {
            if (want_shadow) {
                Hit sh;
                float smint = kRayEpsilon * (1.f + hmax_abs(si.p));
                float smaxt = ds.dist * (1.f - kShadowEpsilon);
                bool occluded = traverse<true, STATS>(sc, si.p, ds.d, smint, smaxt, stack, sh, c_nodes, c_tris);
                ++c_shadow;
                ++n_rays;
                if (occluded) emitter_val = 0.f;
            }
        }

        // ---- 5. NEE contribution, BSDF sampling, next ray --------------------
        if (alive && !terminate) {
            const bf_material &mat = sc.materials[mat_id];
            if (nee) {
                bool active_e = ds.pdf != 0.f;
                V3 wo = to_local(si.sh, ds.d);
                float bsdf_val, bsdf_pdf;
                bsdf_eval_pdf(mat, si.wi, wo, bsdf_val, bsdf_pdf);
                float mis = ds.delta ? 1.f : mis_weight(ds.pdf, bsdf_pdf);
                if (active_e) result += mis * throughput * bsdf_val * emitter_val;
                if (is_range) aux += si.t;                          // pathlength.cpp:209
            }
            (void) next_1d(rng);                                    // sample1 (unused by these BSDFs)
            float s2x = next_1d(rng), s2y = next_1d(rng);
            BSDFSample bs;
            float bsdf_val = bsdf_sample(mat, si.wi, s2x, s2y, bs);
            throughput = throughput * bsdf_val;
            if (throughput == 0.f) {
                terminate = true;
            } else {
                eta *= bs.eta;
                // si.spawn_ray — interaction.h:61-64
                ro = si.p;
                rd = to_world(si.sh, bs.wo);
                rmint = (1.f + hmax_abs(si.p)) * kRayEpsilon;
                rmaxt = BF_INF;
                prev_p = si.p;
                bs_pdf = bs.pdf;
            }
        }

        // ---- 6. film: render_sample tail + range/time AOVs + ImageBlock::put --
        if (alive && terminate) {
            float L = sensor_w * result;                            // integrator.cpp:286
            float X, Y, Z;
            if (lp.color_mode == BF_COLOR_RGB)
                srgb_to_xyz_grey(L, X, Y, Z);
            else
                X = Y = Z = L;
            float a0 = result, a1 = result, a2 = result;            // AOVs see the unweighted radiance
            if (is_time && lp.color_mode == BF_COLOR_RGB) srgb_to_xyz_grey(result, a0, a1, a2);
            bool ok = film_ok && __builtin_isfinite(X) && __builtin_isfinite(Y) && __builtin_isfinite(Z);
            if (is_range || is_time) ok = ok && __builtin_isfinite(a0) && __builtin_isfinite(a1) && __builtin_isfinite(a2);
            if (ok) {
                accX += X;
                accY += Y;
                accZ += Z;
                accA += valid_ray ? 1.f : 0.f;
                accW += 1.f;
                if (is_range || is_time) {
                    // range.cpp:141-161 / time.cpp:134-153: bin i takes the sample
                    // iff (float)i*w <= aux < (float)i*w + w, evaluated exactly
                    // as written there for the (at most three) candidate bins
                    float w = lp.bin_width;
                    int k = (int) __builtin_floorf(aux / w);
                    for (int i = k - 1; i <= k + 1; ++i) {
                        if (i < 0 || i >= (int) lp.bins) continue;
                        float lo = (float) i * w, hi = (float) i * w + w;
                        if (aux >= lo && aux < hi) {
                            if (is_range) {
                                if (a0 != 0.f) hist_add(s_hist, g_hist, lds_hist, 5u + (uint32_t) i, a0);
                            } else if (a0 != 0.f || a1 != 0.f || a2 != 0.f) {
                                hist_add(s_hist, g_hist, lds_hist, 5u + 3u * (uint32_t) i + 0u, a0);
                                hist_add(s_hist, g_hist, lds_hist, 5u + 3u * (uint32_t) i + 1u, a1);
                                hist_add(s_hist, g_hist, lds_hist, 5u + 3u * (uint32_t) i + 2u, a2);
                            }
                        }
                    }
                }
            } else {
                ++c_invalid;
            }
            if (records) {
                bf_path_record r;
                r.L = L;
                r.aux = aux;
                r.valid = valid_ray ? 1u : 0u;
                r.n_rays = n_rays;
                records[path_i] = r;
            }
            alive = false;
            resume_film = false;
        }
    }

    // ---- epilogue: wave-reduce the base channels, flush the histogram ------
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        accX += __shfl_down(accX, off);
        accY += __shfl_down(accY, off);
        accZ += __shfl_down(accZ, off);
        accA += __shfl_down(accA, off);
        accW += __shfl_down(accW, off);
    }
    if (lane == 0) {
        hist_add(s_hist, g_hist, lds_hist, 0, accX);
        hist_add(s_hist, g_hist, lds_hist, 1, accY);
        hist_add(s_hist, g_hist, lds_hist, 2, accZ);
        hist_add(s_hist, g_hist, lds_hist, 3, accA);
        hist_add(s_hist, g_hist, lds_hist, 4, accW);
    }
    if (lds_hist) {
        __syncthreads();
        for (uint32_t i = tid; i < lp.n_chan; i += kBlock) {
            float v = s_hist[i];
            if (v != 0.f) atomicAdd(&g_hist[i], v);
        }
    }
    // statistics: wave-reduce then one atomic per counter per wave
    unsigned long long v_closest = c_closest, v_shadow = c_shadow, v_nodes = c_nodes, v_tris = c_tris,
                       v_invalid = c_invalid, v_bounces = c_bounces;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v_closest += __shfl_down(v_closest, off);
        v_shadow += __shfl_down(v_shadow, off);
        v_nodes += __shfl_down(v_nodes, off);
        v_tris += __shfl_down(v_tris, off);
        v_invalid += __shfl_down(v_invalid, off);
        v_bounces += __shfl_down(v_bounces, off);
    }
    if (lane == 0) {
        atomicAdd(&counters[CTR_CLOSEST], v_closest);
        atomicAdd(&counters[CTR_SHADOW], v_shadow);
        if (RESUME) atomicAdd(&counters[CTR_TAIL_RAYS], v_closest + v_shadow);
        if (STATS) {
            atomicAdd(&counters[CTR_NODES], v_nodes);
            atomicAdd(&counters[CTR_TRIS], v_tris);
        }
        atomicAdd(&counters[CTR_INVALID], v_invalid);
        atomicAdd(&counters[CTR_BOUNCES], v_bounces);
    }
}

// Scene::ray_intersect / ray_test over a batch of rays (tests, tools)
__global__ __launch_bounds__(kBlock) void bf_trace_kernel(DScene sc, uint64_t n, const float *__restrict__ rays,
                                                          int any_hit, float *__restrict__ out_t,
                                                          uint32_t *__restrict__ out_prim, uint32_t *__restrict__ out_shape,
                                                          float *__restrict__ out_uv, uint8_t *__restrict__ out_hit) {
    __shared__ int s_stack[kStackDepth * kBlock];
    int *stack = s_stack + threadIdx.x;
    uint64_t i = (uint64_t) blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float *r = rays + 8 * i;
    V3 o = mk(r[0], r[1], r[2]), d = mk(r[4], r[5], r[6]);
    float mint = r[3], maxt = r[7];
    Hit h;
    uint32_t a = 0, b = 0;
    if (any_hit) {
        out_hit[i] = traverse<true, false>(sc, o, d, mint, maxt, stack, h, a, b) ? 1 : 0;
    } else {
        bool valid = traverse<false, false>(sc, o, d, mint, maxt, stack, h, a, b);
        if (out_t) out_t[i] = h.t;
        if (out_prim) out_prim[i] = valid ? h.prim : 0xffffffffu;
        if (out_shape) {
            uint32_t s = 0xffffffffu;
            if (valid) s = h.slot < 0 ? sc.rects[-h.slot - 1].shape : __float_as_uint(sc.tris[3 * (size_t) h.slot + 1].w);
            out_shape[i] = s;
        }
        if (out_uv) {
            out_uv[2 * i] = h.u;
            out_uv[2 * i + 1] = h.v;
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
    if (stats)
        hipLaunchKernelGGL((bfd::bf_render_kernel<true, false>), dim3(grid), dim3(bfd::kBlock), lds_bytes, stream, *sc, *lp,
                           g_hist, records, counters, none, 0u);
    else
        hipLaunchKernelGGL((bfd::bf_render_kernel<false, false>), dim3(grid), dim3(bfd::kBlock), lds_bytes, stream, *sc, *lp,
                           g_hist, records, counters, none, 0u);
    return hipGetLastError();
}

// wavefront tail: finish the `n_slots` paths of queue `it` in one launch
extern "C" hipError_t bfk_launch_tail(const bfd::DScene *sc, const bfd::DLaunch *lp, const bfd::WF *wf, uint32_t it,
                                      uint32_t n_slots, float *g_hist, bf_path_record *records, int stats, size_t lds_bytes,
                                      hipStream_t stream) {
    unsigned grid = (n_slots + bfd::kBlock - 1) / bfd::kBlock;
    if (grid == 0) return hipSuccess;
    if (stats)
        hipLaunchKernelGGL((bfd::bf_render_kernel<true, true>), dim3(grid), dim3(bfd::kBlock), lds_bytes, stream, *sc, *lp,
                           g_hist, records, wf->counters, *wf, it);
    else
        hipLaunchKernelGGL((bfd::bf_render_kernel<false, true>), dim3(grid), dim3(bfd::kBlock), lds_bytes, stream, *sc, *lp,
                           g_hist, records, wf->counters, *wf, it);
    return hipGetLastError();
}

extern "C" hipError_t bfk_launch_trace(const bfd::DScene *sc, uint64_t n, const float *rays, int any_hit, float *out_t,
                                       uint32_t *out_prim, uint32_t *out_shape, float *out_uv, uint8_t *out_hit,
                                       hipStream_t stream) {
    unsigned grid = (unsigned) ((n + bfd::kBlock - 1) / bfd::kBlock);
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(bfd::bf_trace_kernel, dim3(grid), dim3(bfd::kBlock), 0, stream, *sc, n, rays, any_hit, out_t,
                       out_prim, out_shape, out_uv, out_hit);
    return hipGetLastError();
}
