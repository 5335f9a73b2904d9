// Device functions shared by the megakernel (bf_kernels.hip) and the wavefront
// kernels (bf_wavefront.hip): BVH traversal, surface interaction, BSDFs,
// emitters, sensors.  Reference semantics (file:line) are cited at each
// function; the oracle (oracle/bf_oracle.cpp) restates the same functions
// independently on the CPU.
#pragma once
#include "bf_device.h"
#include "bf_device_math.h"

namespace bfd {

struct Hit {
    float t, u, v;
    uint32_t prim;      // global primitive index (tie rule)
    int32_t slot;       // triangle slot in leaf order, or -(rect+1)
};

// Mesh::ray_intersect_triangle — include/mitsuba/render/mesh.h:190-224
BF_DEV bool tri_intersect(V3 p0, V3 p1, V3 p2, V3 o, V3 d, float mint, float maxt, float &t, float &u, float &v) {
    V3 e1 = p1 - p0, e2 = p2 - p0;
    V3 pvec = cross(d, e2);
    float inv_det = rcp(dot(e1, pvec));
    V3 tvec = o - p0;
    u = dot(tvec, pvec) * inv_det;
    bool active = u >= 0.f && u <= 1.f;
    V3 qvec = cross(tvec, e1);
    v = dot(d, qvec) * inv_det;
    active = active && v >= 0.f && u + v <= 1.f;
    t = dot(e2, qvec) * inv_det;
    return active && t >= mint && t <= maxt;
}

// Rectangle::ray_intersect_preliminary — src/shapes/rectangle.cpp:229-249
// The z row of the object-space ray comes first: t = -o.z / d.z places most rays outside [mint, maxt] (every ray that
// points away from the rectangle's plane), and those never touch the x / y rows (the same operations in the same order as
// xf_point / xf_vector, so t and the local coordinates are the reference's bit for bit).
BF_DEV bool rect_intersect(CRect &rc, V3 o, V3 d, float mint, float maxt, float &t, float &lx, float &ly) {
    const float ooz = fmadd(rc.to_object[10], o.z, fmadd(rc.to_object[9], o.y, fmadd(rc.to_object[8], o.x, rc.to_object[11])));
    const float ddz = fmadd(rc.to_object[10], d.z, fmadd(rc.to_object[9], d.y, rc.to_object[8] * d.x));
    float d_rcp_z = rcp(ddz);
    t = -ooz * d_rcp_z;
    lx = ly = 0.f;
    if (!(t >= mint && t <= maxt)) return false;
    const float oox = fmadd(rc.to_object[2], o.z, fmadd(rc.to_object[1], o.y, fmadd(rc.to_object[0], o.x, rc.to_object[3])));
    const float ooy = fmadd(rc.to_object[6], o.z, fmadd(rc.to_object[5], o.y, fmadd(rc.to_object[4], o.x, rc.to_object[7])));
    const float ddx = fmadd(rc.to_object[2], d.z, fmadd(rc.to_object[1], d.y, rc.to_object[0] * d.x));
    const float ddy = fmadd(rc.to_object[6], d.z, fmadd(rc.to_object[5], d.y, rc.to_object[4] * d.x));
    lx = fmadd(ddx, t, oox);
    ly = fmadd(ddy, t, ooy);
    return __builtin_fabsf(lx) <= 1.f && __builtin_fabsf(ly) <= 1.f;
}

// Closest-hit tie rule: the reference shrinks ray.maxt and accepts t <= maxt
// (kdtree.h:2139-2156; Scene::ray_intersect_naive), so among equal t the
// primitive tested later — the larger global index in the naive order — wins.
// Fixing that rule makes the result independent of traversal order.
BF_DEV void consider(Hit &best, float t, float u, float v, uint32_t prim, int32_t slot) {
    if (t < best.t || (t == best.t && prim > best.prim)) {
        best.t = t;
        best.u = u;
        best.v = v;
        best.prim = prim;
        best.slot = slot;
    }
}

BF_DEV bool slab(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 o, V3 id, float mint, float tmax,
                 float &tn) {
    float t0x = (lox - o.x) * id.x, t1x = (hix - o.x) * id.x;
    float t0y = (loy - o.y) * id.y, t1y = (hiy - o.y) * id.y;
    float t0z = (loz - o.z) * id.z, t1z = (hiz - o.z) * id.z;
    tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                         __builtin_fmaxf(__builtin_fminf(t0z, t1z), mint));
    float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)),
                               __builtin_fminf(__builtin_fmaxf(t0z, t1z), tmax));
    return tn <= tf * 1.0000004f;
}

// Same test with the origin folded in: t = lo * (1/d) - o * (1/d), one fma per plane
// (12 instead of 24 VALU ops for the two child boxes).  The rounding of o * (1/d)
// adds an absolute error of ~6e-8 * |o| to the plane distances; node boxes are
// padded by 2e-6 * max|coordinate| (bf_bvh.cpp), 30x that, so the test stays
// conservative.  Box tests only decide WHICH triangles get tested, never a hit value.
//
// `oid` / `ohi`: the origin term of the lo planes and of the hi planes.  They are the same vector except in batched
// launches with moving meshes (Shift below), where oid = -o/d - slack/d and ohi = -o/d + slack/d widen every box by
// `slack` on the ray's side at no extra instruction.
BF_DEV bool slab_fma(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 id, V3 oid, V3 ohi, float mint, float tmax,
                     float &tn) {
    float t0x = fmadd(lox, id.x, oid.x), t1x = fmadd(hix, id.x, ohi.x);
    float t0y = fmadd(loy, id.y, oid.y), t1y = fmadd(hiy, id.y, ohi.y);
    float t0z = fmadd(loz, id.z, oid.z), t1z = fmadd(hiz, id.z, ohi.z);
    tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                         __builtin_fmaxf(__builtin_fminf(t0z, t1z), mint));
    float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)),
                               __builtin_fminf(__builtin_fmaxf(t0z, t1z), tmax));
    return tn <= tf * 1.0000004f;
}

// ---------------------------------------------------------------------------
// Four-wide BVH traversal step (bf_bvh.h: Node4, 128 B = one L2 line).
// ---------------------------------------------------------------------------
constexpr int kNoNode = INT32_MIN;          // "no node": also Node4's empty-child marker
constexpr uint32_t kMissKey = 0x7f000000u;  // sort key of a child the ray misses

// 1/d with zero components mapped to +-1e30 instead of +-inf, so that the folded form
// lo * (1/d) - o * (1/d) never meets inf - inf or 0 * inf (|o|, |lo| < 3e8).
BF_DEV void ray_inverse(V3 o, V3 d, V3 &id, V3 &oid) {
    id = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
    id.x = __builtin_fminf(__builtin_fmaxf(id.x, -1.0e30f), 1.0e30f);
    id.y = __builtin_fminf(__builtin_fmaxf(id.y, -1.0e30f), 1.0e30f);
    id.z = __builtin_fminf(__builtin_fmaxf(id.z, -1.0e30f), 1.0e30f);
    oid = mk(-o.x * id.x, -o.y * id.y, -o.z * id.z);
}

// Mesh shift of a batched launch (bf_render_batch_device, DLaunch::batch_offsets): the render's meshes stand at
// fl(p + d) — the very vertices bf_scene_translate_meshes stores, so hits are bit-equal to a render of the
// translated scene — while the BVH stays as built.  For the box tests the RAY is moved the other way (o - d) and
// every box is widened by `slack` on the ray's side (slab_fma): the roundings of o - d and of p + d move a plane by
// at most 2^-23 (|o| + |d|) + 2^-24 (|p| + |d|), and the host sets slack = 1e-6 (scene bound + largest |d|), several
// times that.  Box tests only select triangles.  `on` is wave-uniform.
struct Shift {
    bool on;
    V3 d;
    float slack;
};
BF_DEV Shift no_shift() { return Shift{false, mk(0.f, 0.f, 0.f), 0.f}; }
BF_DEV Shift make_shift(const float4 *__restrict__ offsets, uint32_t render, float slack) {
    if (!offsets) return no_shift();
    const float4 q = offsets[render];
    return Shift{true, mk(q.x, q.y, q.z), slack};
}
BF_DEV V3 shifted(V3 p, const Shift &sh) { return sh.on ? p + sh.d : p; }
BF_DEV void ray_inverse_shift(V3 o, V3 d, const Shift &sh, V3 &id, V3 &oid, V3 &ohi) {
    ray_inverse(sh.on ? o - sh.d : o, d, id, oid);
    ohi = oid;
    if (sh.on) {
        oid = mk(fnmadd(sh.slack, id.x, oid.x), fnmadd(sh.slack, id.y, oid.y), fnmadd(sh.slack, id.z, oid.z));
        ohi = mk(fmadd(sh.slack, id.x, ohi.x), fmadd(sh.slack, id.y, ohi.y), fmadd(sh.slack, id.z, ohi.z));
    }
}

// Entry distance and child slot packed into one sortable word (distances are >= mint >= 0,
// so their bit patterns order like the values; the two low mantissa bits carry the slot —
// the order of near-equal children is only a traversal heuristic, never a result).
BF_DEV uint32_t child_key(bool hit, float tn, uint32_t slot) {
    uint32_t b = min(__float_as_uint(tn), kMissKey - 4u);
    return hit ? ((b & ~3u) | slot) : (kMissKey | slot);
}
BF_DEV int pick_child(float4 ch, uint32_t key) {
    uint32_t i = key & 3u;
    return __float_as_int(i == 0u ? ch.x : (i == 1u ? ch.y : (i == 2u ? ch.z : ch.w)));
}

// Per-lane traversal stack: the first N_LDS entries in column `tid` of a lane-strided LDS array
// (entry k at lds[k * kBlock]: bank = tid mod 32, conflict-free), deeper entries — rare: the worst
// case of a four-wide tree is three per level — in a per-thread HBM column (DScene::spill, entry
// k at spill[k * spill_stride]).  SPILL = false: the host guarantees BVH4::stack_need <= N_LDS
// (bf_bvh.h) and the overflow path is compiled out.
// pop(): BF_FLAT_POP = 1 (default) reads the entry through ONE flat load on a selected pointer; 0 branches between an LDS and a
// global load.  The branch looked like the obvious win (a flat load is a vector-memory instruction and waits for both
// counters) and measured 1.3 % SLOWER in wf_trace (3.85 -> 3.90 ms per C2 step, profiles/r04_flat_ops_ab.txt): the extra
// divergence costs more than the flat path, so the select stays.
#ifndef BF_FLAT_POP
#define BF_FLAT_POP 1
#endif
template <int N_LDS, bool SPILL>
struct LaneStack {
    // The two homes of an entry are named by address space: with generic pointers the compiler turns pop()'s choice into a
    // select of two pointers and ONE flat_load — a vector-memory instruction (the busiest unit of the traversal kernels) that
    // also waits for every outstanding store (round 4: every pop of wf_trace in rounds 1-3 was one).
    __attribute__((address_space(3))) int *lds;
    __attribute__((address_space(1))) int *spill;
    uint32_t spill_stride;
    int sp;
    BF_DEV void push(int v) {
        if (!SPILL || sp < N_LDS)
            lds[sp * kBlock] = v;
        else
            spill[(size_t) (sp - N_LDS) * spill_stride] = v;
        ++sp;
    }
    BF_DEV int pop() {
        --sp;
#if BF_FLAT_POP
        return (!SPILL || sp < N_LDS) ? ((int *) lds)[sp * kBlock] : ((int *) spill)[(size_t) (sp - N_LDS) * spill_stride];
#else
        if (!SPILL || sp < N_LDS) return lds[sp * kBlock];
        return spill[(size_t) (sp - N_LDS) * spill_stride];
#endif
    }
    BF_DEV int pop_or_none() { return sp ? pop() : kNoNode; }
    // up to three entries a, b, c in that order (pa implies pb implies pc: the caller's entries are sorted, the absent ones
    // first).  The common case — all three would still land in the LDS part — is three predicated LDS writes at computed
    // offsets; push() one by one (a branch on the entry's home each) only near the spill boundary.
    BF_DEV void push3(int a, bool pa, int b, bool pb, int c, bool pc) {
        const int n = (pa ? 1 : 0) + (pb ? 1 : 0) + (pc ? 1 : 0);
        if (sp + 3 <= N_LDS) {
            // three UNCONDITIONAL writes: an absent entry goes to the slot above the new top (sp + n <= N_LDS - 1 when one is
            // absent), which holds nothing
            const int top = sp + n;
            lds[(pc ? top - 1 : top) * kBlock] = c;
            lds[(pb ? top - 2 : top) * kBlock] = b;
            lds[(pa ? top - 3 : top) * kBlock] = a;
            sp = top;
        } else if (!SPILL) {
            if (pc) lds[(sp + n - 1) * kBlock] = c;
            if (pb) lds[(sp + n - 2) * kBlock] = b;
            if (pa) lds[(sp + n - 3) * kBlock] = a;
            sp += n;
        } else {
            if (pa) push(a);
            if (pb) push(b);
            if (pc) push(c);
        }
    }
};
template <int N_LDS, bool SPILL>
BF_DEV LaneStack<N_LDS, SPILL> make_stack(const DScene &sc, int *lds_column) {
    LaneStack<N_LDS, SPILL> st;
    st.lds = (__attribute__((address_space(3))) int *) lds_column;
    st.spill = (__attribute__((address_space(1))) int *) (sc.spill + ((size_t) blockIdx.x * kBlock + threadIdx.x));
    st.spill_stride = sc.spill_stride;
    st.sp = 0;
    return st;
}

#ifndef BF_NODE_SORT_PAIRS
#define BF_NODE_SORT_PAIRS 1
#endif
// Visit internal node `node`: test its (up to) four child boxes against the ray segment
// [mint, tmax], push the hit children far-to-near and return the nearest one (or the next
// stack entry, or kNoNode when the traversal is finished).
template <class Stack>
BF_DEV int node4_decide(const float4 lx, const float4 ly, const float4 lz, const float4 hx, const float4 hy, const float4 hz,
                        const float4 ch, V3 id, V3 oid, V3 ohi, float mint, float tmax, Stack &st) {
    float t0, t1, t2, t3;
    const bool h0 = slab_fma(lx.x, ly.x, lz.x, hx.x, hy.x, hz.x, id, oid, ohi, mint, tmax, t0);
    const bool h1 = slab_fma(lx.y, ly.y, lz.y, hx.y, hy.y, hz.y, id, oid, ohi, mint, tmax, t1);
    const bool h2 = slab_fma(lx.z, ly.z, lz.z, hx.z, hy.z, hz.z, id, oid, ohi, mint, tmax, t2) && __float_as_int(ch.z) != kNoNode;
    const bool h3 = slab_fma(lx.w, ly.w, lz.w, hx.w, hy.w, hz.w, id, oid, ohi, mint, tmax, t3) && __float_as_int(ch.w) != kNoNode;
    uint32_t k0 = child_key(h0, t0, 0u), k1 = child_key(h1, t1, 1u), k2 = child_key(h2, t2, 2u), k3 = child_key(h3, t3, 3u);
#if BF_NODE_SORT_PAIRS
    // 5-comparator sorting network on (key, child) PAIRS: the children come out in traversal order (no selects by slot
    // afterwards), the three far ones go onto the stack with predicated writes (LaneStack::push3)
    int c0 = __float_as_int(ch.x), c1 = __float_as_int(ch.y), c2 = __float_as_int(ch.z), c3 = __float_as_int(ch.w);
#define BF_CSWAP(ka, ca, kb, cb)                   \
    {                                              \
        const bool sw = ka > kb;                   \
        const uint32_t kl = sw ? kb : ka, kh = sw ? ka : kb; \
        const int cl = sw ? cb : ca, chh = sw ? ca : cb;     \
        ka = kl;                                   \
        ca = cl;                                   \
        kb = kh;                                   \
        cb = chh;                                  \
    }
    BF_CSWAP(k0, c0, k1, c1)
    BF_CSWAP(k2, c2, k3, c3)
    BF_CSWAP(k0, c0, k2, c2)
    BF_CSWAP(k1, c1, k3, c3)
    BF_CSWAP(k1, c1, k2, c2)
#undef BF_CSWAP
    st.push3(c3, k3 < kMissKey, c2, k2 < kMissKey, c1, k1 < kMissKey);
    if (k0 < kMissKey) return c0;
    return st.pop_or_none();
#else
    // 5-comparator sorting network
    const uint32_t a = min(k0, k1), b = max(k0, k1), c = min(k2, k3), d = max(k2, k3);
    const uint32_t lo = min(a, c), x = max(a, c), y = min(b, d), hi = max(b, d);
    const uint32_t m1 = min(x, y), m2 = max(x, y);
    if (hi < kMissKey) st.push(pick_child(ch, hi));
    if (m2 < kMissKey) st.push(pick_child(ch, m2));
    if (m1 < kMissKey) st.push(pick_child(ch, m1));
    if (lo < kMissKey) return pick_child(ch, lo);
    return st.pop_or_none();
#endif
}
template <class Stack>
BF_DEV int node4_step(const float4 *__restrict__ nodes, int node, V3 id, V3 oid, V3 ohi, float mint, float tmax, Stack &st) {
    const float4 *np = nodes + 8u * (uint32_t) node;
    const float4 lx = np[0], ly = np[1], lz = np[2], hx = np[3], hy = np[4], hz = np[5], ch = np[6];
    return node4_decide(lx, ly, lz, hx, hy, hz, ch, id, oid, ohi, mint, tmax, st);
}

// The top of the tree — its first n_top nodes, breadth-first (bf_bvh.h: kTopNodes) — is walked by every ray; a
// workgroup keeps a copy in LDS (`top`, kTopStride float4 per node: 144-byte stride, so that lanes at different
// nodes fall into different bank groups while lanes at the same node broadcast) and only deeper nodes go through
// the vector-memory pipeline, the busiest unit of the traversal kernels.
constexpr uint32_t kTopStride = 9;
typedef float bf_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const bf_f4 *lds_f4_ptr;
typedef __attribute__((address_space(3))) const float *lds_f_ptr;
typedef const BF_CAS bf_f4 *c_f4_ptr;      // constant address space: a wave-uniform address loads through the scalar cache
BF_DEV float4 to_float4(bf_f4 v) { return make_float4(v.x, v.y, v.z, v.w); }
BF_DEV void load_top_nodes(const float4 *__restrict__ nodes, uint32_t n_top, float4 *top, uint32_t tid, uint32_t n_threads) {
    for (uint32_t i = tid; i < n_top * 7u; i += n_threads) {
        const uint32_t node = i / 7u, j = i - node * 7u;
        top[node * kTopStride + j] = nodes[8u * node + j];
    }
}
template <class Stack>
BF_DEV int node4_step_top(const float4 *__restrict__ nodes, const float4 *top, int n_top, int node, V3 id, V3 oid, V3 ohi, float mint,
                          float tmax, Stack &st) {
    float4 lx, ly, lz, hx, hy, hz, ch;
    if (node < n_top) {
        // read through an LDS-qualified pointer: with a generic one the compiler folds both branches into one
        // flat_load per row, which still occupies the vector-memory pipeline for every lane
        const lds_f4_ptr np = (lds_f4_ptr) top + kTopStride * (uint32_t) node;
        lx = to_float4(np[0]), ly = to_float4(np[1]), lz = to_float4(np[2]), hx = to_float4(np[3]), hy = to_float4(np[4]),
        hz = to_float4(np[5]), ch = to_float4(np[6]);
    } else {
        const float4 *np = nodes + 8u * (uint32_t) node;
        lx = np[0], ly = np[1], lz = np[2], hx = np[3], hy = np[4], hz = np[5], ch = np[6];
    }
    return node4_decide(lx, ly, lz, hx, hy, hz, ch, id, oid, ohi, mint, tmax, st);
}

// ---------------------------------------------------------------------------
// Quantised four-wide node step (bf_bvh.h: Node4Q, 64 bytes): four 16-byte loads instead of seven.  The child planes
// are lo_a + q 2^(e_a - 127); folded into the slab test, t = q (s / d) + (lo / d - o / d): one byte-to-float conversion
// and one fma per plane.  Every quantised box contains the fp32 box of the same child (the host rounds lower planes
// down and upper planes up against exactly this arithmetic), and box tests only select triangles.
// ---------------------------------------------------------------------------
constexpr uint32_t kTopStrideQ = 5;          // float4 per node of the LDS copy of the top levels (80-byte stride)
BF_DEV float qbyte(uint32_t w, int k) { return (float) ((w >> (8 * k)) & 0xffu); }      // v_cvt_f32_ubyteK
template <class Stack>
BF_DEV int node4q_decide(const float4 q0, const float4 q1, const float4 q2, const float4 q3, V3 id, V3 oid, V3 ohi, float mint,
                         float tmax, Stack &st) {
    const uint32_t ex = __float_as_uint(q0.w);
    const float ax = __uint_as_float((ex & 0xffu) << 23) * id.x, ay = __uint_as_float(((ex >> 8) & 0xffu) << 23) * id.y,
                az = __uint_as_float(((ex >> 16) & 0xffu) << 23) * id.z;
    const float blx = fmadd(q0.x, id.x, oid.x), bly = fmadd(q0.y, id.y, oid.y), blz = fmadd(q0.z, id.z, oid.z);
    const float bhx = fmadd(q0.x, id.x, ohi.x), bhy = fmadd(q0.y, id.y, ohi.y), bhz = fmadd(q0.z, id.z, ohi.z);
    const uint32_t lx = __float_as_uint(q2.x), ly = __float_as_uint(q2.y), lz = __float_as_uint(q2.z), hx = __float_as_uint(q2.w),
                   hy = __float_as_uint(q3.x), hz = __float_as_uint(q3.y);
    float tn[4];
    bool hit[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float t0x = fmadd(qbyte(lx, k), ax, blx), t1x = fmadd(qbyte(hx, k), ax, bhx);
        const float t0y = fmadd(qbyte(ly, k), ay, bly), t1y = fmadd(qbyte(hy, k), ay, bhy);
        const float t0z = fmadd(qbyte(lz, k), az, blz), t1z = fmadd(qbyte(hz, k), az, bhz);
        tn[k] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                                __builtin_fmaxf(__builtin_fminf(t0z, t1z), mint));
        const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)),
                                         __builtin_fminf(__builtin_fmaxf(t0z, t1z), tmax));
        hit[k] = tn[k] <= tf * 1.0000004f;
    }
    const float4 ch = q1;
    const bool h0 = hit[0] && __float_as_int(ch.x) != kNoNode, h1 = hit[1] && __float_as_int(ch.y) != kNoNode;
    const bool h2 = hit[2] && __float_as_int(ch.z) != kNoNode, h3 = hit[3] && __float_as_int(ch.w) != kNoNode;
    const uint32_t k0 = child_key(h0, tn[0], 0u), k1 = child_key(h1, tn[1], 1u), k2 = child_key(h2, tn[2], 2u), k3 = child_key(h3, tn[3], 3u);
    const uint32_t a = min(k0, k1), b = max(k0, k1), c = min(k2, k3), d = max(k2, k3);
    const uint32_t lo = min(a, c), x = max(a, c), y = min(b, d), hi = max(b, d);
    const uint32_t m1 = min(x, y), m2 = max(x, y);
    if (hi < kMissKey) st.push(pick_child(ch, hi));
    if (m2 < kMissKey) st.push(pick_child(ch, m2));
    if (m1 < kMissKey) st.push(pick_child(ch, m1));
    if (lo < kMissKey) return pick_child(ch, lo);
    return st.pop_or_none();
}
BF_DEV void load_top_qnodes(const float4 *__restrict__ qnodes, uint32_t n_top, float4 *top, uint32_t tid, uint32_t n_threads) {
    for (uint32_t i = tid; i < n_top * 4u; i += n_threads) {
        const uint32_t node = i >> 2, j = i & 3u;
        top[node * kTopStrideQ + j] = qnodes[4u * node + j];
    }
}
template <class Stack>
BF_DEV int node4q_step_top(const float4 *__restrict__ qnodes, const float4 *top, int n_top, int node, V3 id, V3 oid, V3 ohi, float mint,
                           float tmax, Stack &st) {
    float4 q0, q1, q2, q3;
    if (node < n_top) {
        const lds_f4_ptr np = (lds_f4_ptr) top + kTopStrideQ * (uint32_t) node;
        q0 = to_float4(np[0]), q1 = to_float4(np[1]), q2 = to_float4(np[2]), q3 = to_float4(np[3]);
    } else {
        const float4 *np = qnodes + 4u * (uint32_t) node;
        q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
    }
    return node4q_decide(q0, q1, q2, q3, id, oid, ohi, mint, tmax, st);
}

// all triangles of one leaf; returns true when an any-hit query is decided
template <bool STATS>
BF_DEV bool leaf_intersect(const DScene &sc, int node, bool any, V3 o, V3 d, float mint, float maxt, Hit &best, uint32_t &n_tris,
                           const Shift &sh = no_shift()) {
    const uint32_t enc = ~(uint32_t) node;
    const uint32_t first = enc >> 3, cnt = (enc & 7u) + 1u;
    for (uint32_t i = 0; i < cnt; ++i) {
        const float4 *tp = sc.tris + kTriStride * (first + i);
        const float4 a = tp[0], b = tp[1], c = tp[2];
        if (STATS) ++n_tris;
        float t, u, v;
        if (tri_intersect(shifted(mk(a.x, a.y, a.z), sh), shifted(mk(b.x, b.y, b.z), sh), shifted(mk(c.x, c.y, c.z), sh), o, d, mint, maxt, t, u, v)) {
            if (any) return true;
            consider(best, t, u, v, __float_as_uint(a.w), (int32_t) (first + i));
        }
    }
    return false;
}

// Scene::ray_intersect / ray_test — src/librender/scene.cpp:129-178.
// `stack` points at this lane's column of the workgroup's LDS stack
// (entry k at stack[k * kBlock], kStackDepth entries).  `any` may differ between the lanes of a
// wave (closest-hit and any-hit queries share the traversal loop).
template <bool STATS, bool SPILL>
BF_DEV bool traverse_dyn(const DScene &sc, bool any, V3 o, V3 d, float mint, float maxt, int *stack, Hit &best, uint32_t &n_nodes,
                         uint32_t &n_tris, const Shift &sh = no_shift()) {
    best.t = BF_INF;
    best.u = best.v = 0.f;
    best.prim = 0;
    best.slot = 0;
    // analytic rectangles (antennas, target plate, ground): a handful per scene
    for (uint32_t i = 0; i < sc.n_rects; ++i) {
        CRect &rc = c_rects(sc)[i];
        float t, lx, ly;
        if (rect_intersect(rc, o, d, mint, maxt, t, lx, ly)) {
            if (any) return true;
            consider(best, t, lx, ly, rc.prim, -(int32_t) (i + 1));
        }
    }
    if (sc.n_tris == 0) return best.t != BF_INF;

    V3 id, oid, ohi;
    ray_inverse_shift(o, d, sh, id, oid, ohi);
    LaneStack<kStackDepth, SPILL> st = make_stack<kStackDepth, SPILL>(sc, stack);
    int node = sc.root;
    while (node != kNoNode) {
        if (node >= 0) {
            if (STATS) ++n_nodes;
            node = node4_step(sc.nodes, node, id, oid, ohi, mint, any ? maxt : __builtin_fminf(maxt, best.t), st);
        } else {
            if (leaf_intersect<STATS>(sc, node, any, o, d, mint, maxt, best, n_tris, sh)) return true;
            node = st.pop_or_none();
        }
    }
    return best.t != BF_INF;
}
// ---------------------------------------------------------------------------
// Quad-cooperative traversal for sparse waves (the latency-bound tail): FOUR lanes per ray.  Lane q of an aligned
// quad tests child q of a node / triangle q of a leaf / rectangle q, q + 4, ...; keys, children and candidate hits
// are exchanged inside the quad, so every lane of it follows the same walk.  A lone wave issues one instruction
// per ~5-10 cycles however few lanes are live, so dividing the instructions per node step by ~3 is what shortens
// the serial depth of the longest paths.  Results are those of traverse_dyn bit for bit (same tests, same tie rule).
// All four lanes pass the same ray; `active` = the quad has a ray at all.  The stack lives in the LEADER's LDS column
// (and spill column): only q == 0 writes, all four read.
// ---------------------------------------------------------------------------
// lane exchanges inside an aligned quad: DPP quad_perm (a register move, no LDS round trip)
template <int CTRL> BF_DEV int quad_perm(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
template <int I> BF_DEV int quad_bcast(int v) { return quad_perm<I | (I << 2) | (I << 4) | (I << 6)>(v); }
template <int I> BF_DEV uint32_t quad_bcast(uint32_t v) { return (uint32_t) quad_bcast<I>((int) v); }
template <int CTRL> BF_DEV float quad_perm(float v) { return __int_as_float(quad_perm<CTRL>(__float_as_int(v))); }

BF_DEV void quad_merge_hit(Hit &best) {
    {   // partner lane ^ 1: quad_perm [1,0,3,2]
        const float t = quad_perm<0xB1>(best.t), u = quad_perm<0xB1>(best.u), v = quad_perm<0xB1>(best.v);
        const uint32_t prim = (uint32_t) quad_perm<0xB1>((int) best.prim);
        const int32_t slot = quad_perm<0xB1>(best.slot);
        if (t != BF_INF) consider(best, t, u, v, prim, slot);
    }
    {   // partner lane ^ 2: quad_perm [2,3,0,1]
        const float t = quad_perm<0x4E>(best.t), u = quad_perm<0x4E>(best.u), v = quad_perm<0x4E>(best.v);
        const uint32_t prim = (uint32_t) quad_perm<0x4E>((int) best.prim);
        const int32_t slot = quad_perm<0x4E>(best.slot);
        if (t != BF_INF) consider(best, t, u, v, prim, slot);
    }
}

template <bool STATS, bool SPILL>
BF_DEV void traverse_quad(const DScene &sc, bool active, bool any, V3 o, V3 d, float mint, float maxt, int *lds_leader_column,
                          Hit &best, bool &found, uint32_t &n_nodes, uint32_t &n_tris, const Shift &sh = no_shift()) {
    const int lane = threadIdx.x & 63, q = lane & 3, base = lane & ~3;
    best.t = BF_INF;
    best.u = best.v = 0.f;
    best.prim = 0;
    best.slot = 0;
    found = false;
    // rectangles: one per lane of the quad
    bool rect_hit = false;
    if (active) {
        for (uint32_t i = (uint32_t) q; i < sc.n_rects; i += 4u) {
            CRect &rc = c_rects(sc)[i];
            float t, lx, ly;
            if (rect_intersect(rc, o, d, mint, maxt, t, lx, ly)) {
                rect_hit = true;
                consider(best, t, lx, ly, rc.prim, -(int32_t) (i + 1));
            }
        }
    }
    {
        const unsigned long long m = __ballot(rect_hit);
        const bool quad_any = ((m >> base) & 0xFull) != 0ull;
        if (any && quad_any) found = true;
        quad_merge_hit(best);
    }
    int node = (active && !found && sc.n_tris != 0) ? sc.root : kNoNode;
    V3 id, oid, ohi;
    ray_inverse_shift(o, d, sh, id, oid, ohi);
    LaneStack<kStackDepth, SPILL> st;
    st.lds = (__attribute__((address_space(3))) int *) lds_leader_column;
    st.spill = (__attribute__((address_space(1))) int *) (sc.spill + ((size_t) blockIdx.x * kBlock + (threadIdx.x & ~3u)));
    st.spill_stride = sc.spill_stride;
    st.sp = 0;
    while (__ballot(node != kNoNode)) {
        if (node >= 0) {
            if (STATS && q == 0) ++n_nodes;
            const float *nf = reinterpret_cast<const float *>(sc.nodes + 8u * (uint32_t) node);
            const float lox = nf[q], loy = nf[4 + q], loz = nf[8 + q], hix = nf[12 + q], hiy = nf[16 + q], hiz = nf[20 + q];
            const int child = __float_as_int(nf[24 + q]);
            float tn;
            const float tmax = any ? maxt : __builtin_fminf(maxt, best.t);
            const bool h = slab_fma(lox, loy, loz, hix, hiy, hiz, id, oid, ohi, mint, tmax, tn) && child != kNoNode;
            const uint32_t key = child_key(h, tn, (uint32_t) q);
            const uint32_t k0 = quad_bcast<0>(key), k1 = quad_bcast<1>(key), k2 = quad_bcast<2>(key), k3 = quad_bcast<3>(key);
            const int c0 = quad_bcast<0>(child), c1 = quad_bcast<1>(child), c2 = quad_bcast<2>(child), c3 = quad_bcast<3>(child);
            const uint32_t a = min(k0, k1), b = max(k0, k1), c = min(k2, k3), dd = max(k2, k3);
            const uint32_t lo = min(a, c), x = max(a, c), y = min(b, dd), hi = max(b, dd);
            const uint32_t m1 = min(x, y), m2 = max(x, y);
            auto pick = [&](uint32_t k) -> int {
                const uint32_t i = k & 3u;
                return i == 0u ? c0 : (i == 1u ? c1 : (i == 2u ? c2 : c3));
            };
            // all four lanes keep sp in step; only the leader touches the memory
            auto push = [&](int v) {
                if (q == 0) {
                    st.push(v);
                } else {
                    ++st.sp;
                }
            };
            if (hi < kMissKey) push(pick(hi));
            if (m2 < kMissKey) push(pick(m2));
            if (m1 < kMissKey) push(pick(m1));
            node = (lo < kMissKey) ? pick(lo) : st.pop_or_none();
        } else if (node != kNoNode) {
            const uint32_t enc = ~(uint32_t) node;
            const uint32_t first = enc >> 3, cnt = (enc & 7u) + 1u;
            bool tri_hit = false;
            for (uint32_t i = (uint32_t) q; i < cnt; i += 4u) {
                const float4 *tp = sc.tris + kTriStride * (first + i);
                const float4 ta = tp[0], tb = tp[1], tc = tp[2];
                if (STATS) ++n_tris;
                float t, u, v;
                if (tri_intersect(shifted(mk(ta.x, ta.y, ta.z), sh), shifted(mk(tb.x, tb.y, tb.z), sh), shifted(mk(tc.x, tc.y, tc.z), sh), o, d,
                                  mint, maxt, t, u, v)) {
                    tri_hit = true;
                    consider(best, t, u, v, __float_as_uint(ta.w), (int32_t) (first + i));
                }
            }
            const unsigned long long m = __ballot(tri_hit);
            const bool quad_any = ((m >> base) & 0xFull) != 0ull;
            quad_merge_hit(best);
            if (any && quad_any) {
                found = true;
                node = kNoNode;
            } else {
                node = st.pop_or_none();
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Row-cooperative traversal for the deep tail: ONE ray per 16-lane DPP row on the sixteen-wide tree (bf_bvh.h:
// Node16).  Lane j of the row tests child j of a node, triangle j of a leaf (up to 16 per leaf), rectangle j.
// A lone path's bounce is a chain of dependent fetches; with four-wide nodes and two-triangle leaves that chain is
// ~25 steps long, here it is ~log16 of the triangle count.  Every loop iteration issues ONE batch of loads per row
// (child box or triangle, same registers), so rows at nodes and rows at leaves share a single memory round trip.
// Box tests only select triangles; triangle tests and the tie rule are those of traverse_dyn, so hits are bit-equal.
// The row's stack lives in the LDS columns of its 16 lanes (entry e of the row whose first thread is `b` at
// lds[(e / 16) * kBlock + b + e % 16]): kStackDepth * 16 = kWideStack entries; the host only enables the wide tree
// when its worst case fits.
// ---------------------------------------------------------------------------
template <int N> BF_DEV int row_ror(int v) { return __builtin_amdgcn_mov_dpp(v, 0x120 + N, 0xf, 0xf, true); }
template <int N> BF_DEV uint32_t row_ror(uint32_t v) { return (uint32_t) row_ror<N>((int) v); }
template <int N> BF_DEV float row_ror(float v) { return __int_as_float(row_ror<N>(__float_as_int(v))); }

template <int N> BF_DEV void row_merge_step(Hit &best) {
    const float t = row_ror<N>(best.t), u = row_ror<N>(best.u), v = row_ror<N>(best.v);
    const uint32_t prim = row_ror<N>(best.prim);
    const int32_t slot = row_ror<N>(best.slot);
    if (t != BF_INF) consider(best, t, u, v, prim, slot);
}
// after the four steps every lane of the row holds the row's best hit (consider() is a total order)
BF_DEV void row_merge_hit(Hit &best) {
    row_merge_step<8>(best);
    row_merge_step<4>(best);
    row_merge_step<2>(best);
    row_merge_step<1>(best);
}

#ifdef BF_TAIL_PROF
struct RowProf {
    unsigned long long steps, rect, mem, cmp;
};
#define BF_ROWPROF_ARG , RowProf &rp
#define BF_ROWPROF_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define BF_ROWPROF_ARG
#define BF_ROWPROF_STAMP(v)
#endif
// hit exchange between the rows of a gang (lane ^ 16, lane ^ 32): executed by every lane of the wave
template <int XOR> BF_DEV void gang_merge_step(Hit &best) {
    const float t = __shfl_xor(best.t, XOR), u = __shfl_xor(best.u, XOR), v = __shfl_xor(best.v, XOR);
    const uint32_t prim = (uint32_t) __shfl_xor((int) best.prim, XOR);
    const int32_t slot = __shfl_xor(best.slot, XOR);
    if (t != BF_INF) consider(best, t, u, v, prim, slot);
}

// `rlog`: log2 of the rows per ray (wave-uniform): a gang of 1, 2 or 4 rows shares one ray, one stack (in the LDS
// columns of its first row) and one best hit, and pops up to that many stack entries per step — the serial depth of a
// lone ray drops from "every node it visits" towards the depth of the tree, at the price of a little less culling.
// All lanes of a gang pass the same ray.
template <bool STATS>
BF_DEV void traverse_row16(const DScene &sc, uint32_t rlog, bool active, bool any, V3 o, V3 d, float mint, float maxt,
                           int *lds_block_stack, Hit &best, bool &found, uint32_t &n_nodes, uint32_t &n_tris, const Shift &sh BF_ROWPROF_ARG) {
    BF_ROWPROF_STAMP(rp_t0);
    const uint32_t lane = threadIdx.x & 63u, j = lane & 15u;
    const uint32_t rows = 1u << rlog, row = (lane >> 4) & (rows - 1u);          // this lane's row inside its gang
    const uint32_t gang_shift = lane & ~(16u * rows - 1u) & 63u;                // first lane of the gang
    const unsigned long long gang_mask = rows == 4u ? ~0ull : ((1ull << (16u * rows)) - 1ull);
    int *const stk = lds_block_stack + (threadIdx.x & ~(16u * rows - 1u));    // + (e >> 4) * kBlock + (e & 15)
    best.t = BF_INF;
    best.u = best.v = 0.f;
    best.prim = 0;
    best.slot = 0;
    found = false;
    bool rect_hit = false;
    if (active) {
        for (uint32_t i = j + 16u * row; i < sc.n_rects; i += 16u * rows) {
            CRect &rc = c_rects(sc)[i];
            float t, lx, ly;
            if (rect_intersect(rc, o, d, mint, maxt, t, lx, ly)) {
                rect_hit = true;
                consider(best, t, lx, ly, rc.prim, -(int32_t) (i + 1));
            }
        }
    }
    {
        const unsigned long long m = __ballot(rect_hit);
        if (m) {                                   // wave-uniform: some gang has a rectangle hit
            if (any && ((m >> gang_shift) & gang_mask)) found = true;
            row_merge_hit(best);
            if (rlog >= 1u) gang_merge_step<16>(best);
            if (rlog >= 2u) gang_merge_step<32>(best);
        }
    }
    V3 id, oid, ohi;
    ray_inverse_shift(o, d, sh, id, oid, ohi);
    int sp = (active && !found && sc.n_tris != 0) ? 1 : 0;
    if (sp && row == 0u && j == 0u) stk[0] = sc.wroot;
#ifdef BF_TAIL_PROF
    rp.rect += __builtin_amdgcn_s_memtime() - rp_t0;
#endif
    while (__ballot(sp > 0)) {
        BF_ROWPROF_STAMP(rp_t1);
        // ---- pop: row r of the gang takes the r-th entry from the top ---------------------------
        const int n_take = min((int) rows, sp);
        int node = kNoNode;
        if ((int) row < n_take) {
            const int e = sp - 1 - (int) row;
            node = stk[(e >> 4) * kBlock + (e & 15)];
        }
        sp -= n_take;
        const bool is_node = node >= 0, is_leaf = node < 0 && node != kNoNode;
        // ---- one batch of loads per row: child box j, or triangle j of the leaf ----------------
        const uint32_t enc = ~(uint32_t) node;
        const uint32_t first = enc >> 4, cnt = (enc & 15u) + 1u;
        const float4 *ptr = sc.tris;
        bool ld = false;
        if (is_node) {
            ptr = sc.wnodes + (32u * (uint32_t) node + 2u * j);
            ld = true;
        } else if (is_leaf) {
            ptr = sc.tris + kTriStride * (first + j);
            ld = j < cnt;
        }
        float4 qa = make_float4(0.f, 0.f, 0.f, 0.f), qb = qa, qc = qa;
        if (ld) {
            qa = ptr[0];
            qb = ptr[1];
            if (is_leaf) qc = ptr[2];
        }
#ifdef BF_TAIL_PROF
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BF_ROWPROF_STAMP(rp_t2);
#endif
        bool h = false, tri_hit = false;
        int child = kNoNode;
        uint32_t rank = 0;
        if (is_node) {
            if (STATS && j == 0u) ++n_nodes;
            child = __float_as_int(qb.z);
            float tn;
            const float tmax = any ? maxt : __builtin_fminf(maxt, best.t);
            h = slab_fma(qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, id, oid, ohi, mint, tmax, tn) && child != kNoNode;
            // distinct sortable keys: entry distance (non-negative: bit pattern orders like the value), child slot in the low bits
            const uint32_t key = h ? ((min(__float_as_uint(tn), 0x7f000000u) & ~15u) | j) : 0xffffffffu;
#define BF_RANK(N) rank += row_ror<N>(key) < key ? 1u : 0u;
            BF_RANK(1) BF_RANK(2) BF_RANK(3) BF_RANK(4) BF_RANK(5) BF_RANK(6) BF_RANK(7) BF_RANK(8)
            BF_RANK(9) BF_RANK(10) BF_RANK(11) BF_RANK(12) BF_RANK(13) BF_RANK(14) BF_RANK(15)
#undef BF_RANK
        }
        if (is_leaf && ld) {
            if (STATS) ++n_tris;
            float t, u, v;
            if (tri_intersect(shifted(mk(qa.x, qa.y, qa.z), sh), shifted(mk(qb.x, qb.y, qb.z), sh), shifted(mk(qc.x, qc.y, qc.z), sh), o, d, mint,
                              maxt, t, u, v)) {
                tri_hit = true;
                consider(best, t, u, v, __float_as_uint(qa.w), (int32_t) (first + j));
            }
        }
        // ---- all lanes again: triangle hits are merged over the gang, hit children pushed far-to-near ----
        const unsigned long long tm = __ballot(tri_hit);
        if (tm) {
            if (any && ((tm >> gang_shift) & gang_mask)) found = true;
            row_merge_hit(best);
            if (rlog >= 1u) gang_merge_step<16>(best);
            if (rlog >= 2u) gang_merge_step<32>(best);
        }
        const unsigned long long hm = (__ballot(h) >> gang_shift) & gang_mask;
        if (hm) {
            // the row that popped the top entry (row 0) pushes last, so the nearest child of the nearest node ends on top
            uint32_t above = 0, mine = 0, total = 0;
            for (uint32_t r = 0; r < rows; ++r) {
                const uint32_t c = (uint32_t) __popcll((hm >> (16u * r)) & 0xFFFFull);
                total += c;
                above += r > row ? c : 0u;
                mine = r == row ? c : mine;
            }
            if (h) {
                const int e = sp + (int) (above + mine - 1u - rank);
                stk[(e >> 4) * kBlock + (e & 15)] = child;
            }
            sp += (int) total;
        }
        if (found) sp = 0;
#ifdef BF_TAIL_PROF
        {
            const unsigned long long rp_t3 = __builtin_amdgcn_s_memtime();
            rp.steps += 1;
            rp.mem += rp_t2 - rp_t1;
            rp.cmp += rp_t3 - rp_t2;
        }
#endif
    }
}

template <bool ANY, bool STATS, bool SPILL>
BF_DEV bool traverse(const DScene &sc, V3 o, V3 d, float mint, float maxt, int *stack, Hit &best, uint32_t &n_nodes,
                     uint32_t &n_tris, const Shift &sh = no_shift()) {
    return traverse_dyn<STATS, SPILL>(sc, ANY, o, d, mint, maxt, stack, best, n_nodes, n_tris, sh);
}

// ---------------------------------------------------------------------------
// surface interaction: PreliminaryIntersection::compute_surface_interaction
// (interaction.h:613-644), Mesh::compute_surface_interaction
// (mesh.cpp:452-548), Rectangle::compute_surface_interaction
// (rectangle.cpp:265-298), initialize_sh_frame (interaction.h:159-162)
// ---------------------------------------------------------------------------
struct SI {
    float t;
    V3 p, wi;
    Frame sh;
    uint32_t shape;
    uint32_t material;      // index into DScene::materials (carried by the primitive record: no shapes[] hop)
    int32_t emitter;        // index into DScene::emitters or -1
};
// third tag word of a triangle record: bit 0 = has vertex normals, bit 1 = has texture coordinates,
// bits 8..19 = material, bits 20..31 = emitter + 1
BF_DEV uint32_t tri_tag_material(uint32_t tag) { return (tag >> 8) & 0xfffu; }
BF_DEV int32_t tri_tag_emitter(uint32_t tag) { return (int32_t) (tag >> 20) - 1; }

// the parts of a SurfaceInteraction the estimator never reads (bf_ray_intersect reports them)
struct SIGeom {
    V3 n, dp_du, dp_dv;
};

template <bool FULL = false, int V = 0> BF_DEV void make_si(const DScene &sc, V3 o, V3 d, const Hit &h, SI &si, SIGeom *geom = nullptr,
                                                 const Shift &sh = no_shift()) {
    si.t = h.t;
    V3 dp_du;
    if (h.slot < 0) {
        si.p = fmadd3(d, h.t, o);
        if (!FULL && sc.tab_on) {
            // the rectangle's record from the workgroup's LDS copy (per-lane index): DRect dwords 24.. = s, t, n, 35 = shape, 37 = material, 38 = emitter
            extern __shared__ __align__(16) unsigned char s_dyn[];
            const lds_f_ptr r = (lds_f_ptr) (s_dyn + sc.lds_rect) + kRectDwords * (uint32_t) (-h.slot - 1);
            si.shape = __float_as_uint(r[35]);
            si.material = __float_as_uint(r[37]);
            si.emitter = __float_as_int(r[38]);
            si.sh.n = mk(r[30], r[31], r[32]);
            dp_du = mk(r[24], r[25], r[26]);
        } else {
            CRect &rc = c_rects(sc)[-h.slot - 1];
            si.shape = rc.shape;
            si.material = rc.material;
            si.emitter = rc.emitter;
            si.sh.n = mk(rc.n[0], rc.n[1], rc.n[2]);
            dp_du = mk(rc.s[0], rc.s[1], rc.s[2]);
            if (FULL) {
                geom->n = si.sh.n;
                geom->dp_du = dp_du;
                geom->dp_dv = mk(rc.t[0], rc.t[1], rc.t[2]);
            }
        }
    } else {
        const float4 *tp = sc.tris + kTriStride * (size_t) h.slot;
        float4 a = tp[0], b = tp[1], c = tp[2];
        // the vertex normals are fetched together with the positions, not after the tag has arrived (one dependent
        // memory round trip less per shaded vertex); scenes without any normals skip the load
        float4 na = make_float4(0, 0, 0, 0), nb = na, nc = na;
        if (sc.normals) {
            const float4 *nq = sc.normals + 3 * (size_t) h.slot;
            na = nq[0];
            nb = nq[1];
            nc = nq[2];
        }
        float4 duv = make_float4(0, 0, 0, 0);   // (uv1 - uv0, uv2 - uv0), same speculation
        if (rare<V>(sc.uvs != nullptr)) duv = sc.uvs[h.slot];
        V3 p0 = shifted(mk(a.x, a.y, a.z), sh), p1 = shifted(mk(b.x, b.y, b.z), sh), p2 = shifted(mk(c.x, c.y, c.z), sh);
        si.shape = __float_as_uint(b.w);
        const uint32_t tag = __float_as_uint(c.w);
        si.material = tri_tag_material(tag);
        si.emitter = tri_tag_emitter(tag);
        float b1 = h.u, b2 = h.v, b0 = 1.f - b1 - b2;
        V3 dp0 = p1 - p0, dp1 = p2 - p0;
        si.p = p0 * b0 + p1 * b1 + p2 * b2;
        V3 n = normalize(cross(dp0, dp1));
        V3 dp_dv;
        coordinate_system(n, dp_du, dp_dv);
        if (rare<V>((tag & 2u) != 0u)) {
            // mesh.cpp:493-512: tangents of the UV parameterisation; a degenerate one keeps coordinate_system(n)
            float det = fmsub(duv.x, duv.w, duv.y * duv.z), inv_det = rcp(det);
            if (det != 0.f) {
                dp_du = mk(fmsub(duv.w, dp0.x, duv.y * dp1.x), fmsub(duv.w, dp0.y, duv.y * dp1.y), fmsub(duv.w, dp0.z, duv.y * dp1.z)) * inv_det;
                if (FULL)
                    dp_dv = mk(fnmadd(duv.z, dp0.x, duv.x * dp1.x), fnmadd(duv.z, dp0.y, duv.x * dp1.y), fnmadd(duv.z, dp0.z, duv.x * dp1.z)) * inv_det;
            }
        }
        if (FULL) {
            geom->n = n;
            geom->dp_du = dp_du;
            geom->dp_dv = dp_dv;
        }
        if (tag & 1u) {
            si.sh.n = normalize(mk(na.x, na.y, na.z) * b0 + mk(nb.x, nb.y, nb.z) * b1 + mk(nc.x, nc.y, nc.z) * b2);
        } else {
            si.sh.n = n;
        }
    }
    float dd = dot(si.sh.n, dp_du);
    si.sh.s = normalize(mk(fnmadd(si.sh.n.x, dd, dp_du.x), fnmadd(si.sh.n.y, dd, dp_du.y), fnmadd(si.sh.n.z, dd, dp_du.z)));
    si.sh.t = cross(si.sh.n, si.sh.s);
    si.wi = to_local(si.sh, -d);
}

// ---------------------------------------------------------------------------
// BSDFs: diffuse.cpp:78-135, roughconductor.cpp:196-392 (+ microfacet.h,
// fresnel.h:92-116), twosided.cpp:94-180
// ---------------------------------------------------------------------------
struct Microfacet {
    uint32_t type;
    float au, av;
    bool sample_visible;
};
// A shaded vertex reads its material ONCE, whole (three 16-byte loads issued together as soon as the hit's record names it,
// one wait at the first use) instead of field by field where the BSDF code happens to need one (a dozen dependent
// round trips per vertex through a per-lane address).  Device table: DMaterial, bf_material padded to 48 bytes.
// The copy comes from the workgroup's LDS tables when the scene fits them (DScene::tab_on: ~100 cycles instead of an L2
// round trip through a per-lane address), else from the device table.
BF_DEV void load_tables_lds(DScene &sc, uint32_t byte_off, uint32_t tid) {      // whole workgroup; the caller synchronises
    extern __shared__ __align__(16) unsigned char s_dyn[];
    sc.tab_on = 0u;
    if (!sc.tab_cache) return;
    float *dst = reinterpret_cast<float *>(s_dyn + byte_off);
    const float *gm = reinterpret_cast<const float *>(sc.materials), *gr = reinterpret_cast<const float *>(sc.rects);
    for (uint32_t i = tid; i < sc.n_materials * 12u; i += kBlock) dst[i] = gm[i];
    for (uint32_t i = tid; i < sc.n_rects * kRectDwords; i += kBlock) dst[kTabMaxMaterials * 12u + i] = gr[i];
    sc.lds_mat = byte_off;
    sc.lds_rect = byte_off + kTabMaxMaterials * 48u;
    sc.tab_on = 1u;
}
BF_DEV bf_material load_material(const DScene &sc, uint32_t index) {
    bf_f4 a, b, c;
    if (sc.tab_on) {
        extern __shared__ __align__(16) unsigned char s_dyn[];
        const lds_f4_ptr mp = (lds_f4_ptr) (s_dyn + sc.lds_mat) + 3u * index;
        a = mp[0], b = mp[1], c = mp[2];
    } else {
        const c_f4_ptr mp = (c_f4_ptr) (uintptr_t) (sc.materials + index);
        a = mp[0], b = mp[1], c = mp[2];
    }
    bf_material m;
    m.type = __float_as_uint(a.x);
    m.twosided = __float_as_uint(a.y);
    m.reflectance = a.z;
    m.alpha_u = a.w;
    m.alpha_v = b.x;
    m.distribution = __float_as_uint(b.y);
    m.sample_visible = __float_as_uint(b.z);
    m.eta = b.w;
    m.k = c.x;
    m.has_specular_reflectance = __float_as_uint(c.y);
    m.back_material = __float_as_uint(c.z);
    return m;
}
BF_DEV Microfacet mf_make(const bf_material &m) {
    Microfacet d;
    d.type = m.distribution;
    d.au = __builtin_fmaxf(m.alpha_u, 1e-4f);
    d.av = __builtin_fmaxf(m.alpha_v, 1e-4f);
    d.sample_visible = m.sample_visible != 0;
    return d;
}
BF_DEV float mf_eval(const Microfacet &d, V3 m) {
    float alpha_uv = d.au * d.av, cos_theta = m.z, cos_theta_2 = sqr(cos_theta), result;
    if (d.type == BF_MF_BECKMANN)
        result = exp_cr(-(sqr(m.x / d.au) + sqr(m.y / d.av)) / cos_theta_2) / (kPi * alpha_uv * sqr(cos_theta_2));
    else
        result = rcp(kPi * alpha_uv * sqr(sqr(m.x / d.au) + sqr(m.y / d.av) + sqr(m.z)));
    return (result * cos_theta > 1e-20f) ? result : 0.f;
}
BF_DEV float mf_smith_g1(const Microfacet &d, V3 v, V3 m) {
    float xy_alpha_2 = sqr(d.au * v.x) + sqr(d.av * v.y), tan_theta_alpha_2 = xy_alpha_2 / sqr(v.z), result;
    if (d.type == BF_MF_BECKMANN) {
        float a = 1.f / __builtin_sqrtf(tan_theta_alpha_2), a_sqr = sqr(a);
        result = (a >= 1.6f) ? 1.f : (3.535f * a + 2.181f * a_sqr) / (1.f + 2.276f * a + 2.577f * a_sqr);
    } else {
        result = 2.f / (1.f + __builtin_sqrtf(1.f + tan_theta_alpha_2));
    }
    if (xy_alpha_2 == 0.f) result = 1.f;
    if (dot(v, m) * v.z <= 0.f) result = 0.f;
    return result;
}
BF_DEV float mf_G(const Microfacet &d, V3 wi, V3 wo, V3 m) { return mf_smith_g1(d, wi, m) * mf_smith_g1(d, wo, m); }
BF_DEV void mf_sample_visible_11(const Microfacet &d, float cos_theta_i, float sx, float sy, float &ox, float &oy) {
    if (d.type == BF_MF_BECKMANN) {
        float tan_theta_i = safe_sqrt(fnmadd(cos_theta_i, cos_theta_i, 1.f)) / cos_theta_i;
        float cot_theta_i = rcp(tan_theta_i);
        float maxval = erf_cr(cot_theta_i);
        sx = __builtin_fmaxf(__builtin_fminf(sx, 1.f - 1e-6f), 1e-6f);
        sy = __builtin_fmaxf(__builtin_fminf(sy, 1.f - 1e-6f), 1e-6f);
        float x = maxval - (maxval + 1.f) * erf_cr(__builtin_sqrtf(-log_cr(sx)));
        sx *= 1.f + maxval + kInvSqrtPi * tan_theta_i * exp_cr(-sqr(cot_theta_i));
#pragma nounroll
        for (int i = 0; i < 3; ++i) {
            float slope = erfinv_giles(x);
            float value = 1.f + x + kInvSqrtPi * tan_theta_i * exp_cr(-sqr(slope)) - sx;
            float derivative = 1.f - slope * tan_theta_i;
            x -= value / derivative;
        }
        ox = erfinv_giles(x);
        oy = erfinv_giles(fmsub(2.f, sy, 1.f));
    } else {
        float px, py;
        square_to_uniform_disk_concentric(sx, sy, px, py);
        float s = .5f * (1.f + cos_theta_i);
        float a = safe_sqrt(1.f - sqr(px));
        py = fmadd(py, s, fnmadd(a, s, a));
        float x = px, y = py, z = safe_sqrt(1.f - fmadd(py, py, px * px));
        float sin_theta_i = safe_sqrt(1.f - sqr(cos_theta_i));
        float nrm = rcp(fmadd(sin_theta_i, y, cos_theta_i * z));
        ox = fmsub(cos_theta_i, y, sin_theta_i * z) * nrm;
        oy = x * nrm;
    }
}
BF_DEV void mf_sample(const Microfacet &d, V3 wi, float sx, float sy, V3 &m, float &pdf) {
    if (!d.sample_visible) {
        float sin_phi, cos_phi, cos_theta, cos_theta_2, alpha_2;
        if (d.au == d.av) {
            float ang = (2.f * kPi) * sy;
            sin_phi = sin_cr(ang);
            cos_phi = cos_cr(ang);
            alpha_2 = d.au * d.au;
        } else {
            float ratio = d.av / d.au, tmp = ratio * tan_cr((2.f * kPi) * sy);
            cos_phi = 1.f / __builtin_sqrtf(fmadd(tmp, tmp, 1.f));
            cos_phi = mulsign(cos_phi, __builtin_fabsf(sy - .5f) - .25f);
            sin_phi = cos_phi * tmp;
            alpha_2 = rcp(sqr(cos_phi / d.au) + sqr(sin_phi / d.av));
        }
        if (d.type == BF_MF_BECKMANN) {
            cos_theta = 1.f / __builtin_sqrtf(fnmadd(alpha_2, log_cr(1.f - sx), 1.f));
            cos_theta_2 = sqr(cos_theta);
            float cos_theta_3 = __builtin_fmaxf(cos_theta_2 * cos_theta, 1e-20f);
            pdf = (1.f - sx) / (kPi * d.au * d.av * cos_theta_3);
        } else {
            float tan_theta_m_2 = alpha_2 * sx / (1.f - sx);
            cos_theta = 1.f / __builtin_sqrtf(1.f + tan_theta_m_2);
            cos_theta_2 = sqr(cos_theta);
            float temp = 1.f + tan_theta_m_2 / alpha_2, cos_theta_3 = __builtin_fmaxf(cos_theta_2 * cos_theta, 1e-20f);
            pdf = rcp(kPi * d.au * d.av * cos_theta_3 * sqr(temp));
        }
        float sin_theta = __builtin_sqrtf(1.f - cos_theta_2);
        m = mk(cos_phi * sin_theta, sin_phi * sin_theta, cos_theta);
    } else {
        V3 wi_p = normalize(mk(d.au * wi.x, d.av * wi.y, wi.z));
        float sin_theta_2 = fmadd(wi_p.x, wi_p.x, sqr(wi_p.y));
        float inv_sin_theta = 1.f / __builtin_sqrtf(sin_theta_2);
        float sin_phi, cos_phi;
        if (__builtin_fabsf(sin_theta_2) <= 4.f * kEpsilon) {
            sin_phi = 0.f;
            cos_phi = 1.f;
        } else {
            sin_phi = __builtin_fminf(__builtin_fmaxf(wi_p.y * inv_sin_theta, -1.f), 1.f);
            cos_phi = __builtin_fminf(__builtin_fmaxf(wi_p.x * inv_sin_theta, -1.f), 1.f);
        }
        float slx, sly;
        mf_sample_visible_11(d, wi_p.z, sx, sy, slx, sly);
        float rx = fmsub(cos_phi, slx, sin_phi * sly) * d.au;
        float ry = fmadd(sin_phi, slx, cos_phi * sly) * d.av;
        m = normalize(mk(-rx, -ry, 1.f));
        pdf = mf_eval(d, m) * mf_smith_g1(d, wi, m) * __builtin_fabsf(dot(wi, m)) / wi.z;
    }
}
BF_DEV float mf_pdf(const Microfacet &d, V3 wi, V3 m) {
    float result = mf_eval(d, m);
    if (d.sample_visible)
        result *= mf_smith_g1(d, wi, m) * __builtin_fabsf(dot(wi, m)) / wi.z;
    else
        result *= m.z;
    return result;
}
BF_DEV float fresnel_conductor(float cos_theta_i, float eta_r, float eta_i) {
    float cos_theta_i_2 = cos_theta_i * cos_theta_i, sin_theta_i_2 = 1.f - cos_theta_i_2,
          sin_theta_i_4 = sin_theta_i_2 * sin_theta_i_2;
    float temp_1 = eta_r * eta_r - eta_i * eta_i - sin_theta_i_2,
          a_2_pb_2 = safe_sqrt(temp_1 * temp_1 + 4.f * eta_i * eta_i * eta_r * eta_r),
          a = safe_sqrt(.5f * (a_2_pb_2 + temp_1));
    float term_1 = a_2_pb_2 + cos_theta_i_2, term_2 = 2.f * cos_theta_i * a;
    float r_s = (term_1 - term_2) / (term_1 + term_2);
    float term_3 = a_2_pb_2 * cos_theta_i_2 + sin_theta_i_4, term_4 = term_2 * sin_theta_i_2;
    float r_p = r_s * (term_3 - term_4) / (term_3 + term_4);
    return .5f * (r_s + r_p);
}
BF_DEV V3 reflect(V3 wi, V3 m) {
    float d2 = 2.f * dot(wi, m);
    return mk(fmsub(m.x, d2, wi.x), fmsub(m.y, d2, wi.y), fmsub(m.z, d2, wi.z));
}

struct BSDFSample {
    V3 wo;
    float pdf, eta;
};

BF_DEV float bsdf_sample_1(const bf_material &mat, V3 wi, float s2x, float s2y, BSDFSample &bs) {
    bs.wo = mk(0.f, 0.f, 0.f);
    bs.pdf = 0.f;
    bs.eta = 1.f;
    float cos_theta_i = wi.z;
    if (!(cos_theta_i > 0.f)) return 0.f;
    if (mat.type == BF_BSDF_DIFFUSE) {
        bs.wo = square_to_cosine_hemisphere(s2x, s2y);
        bs.pdf = kInvPi * bs.wo.z;
        return (bs.pdf > 0.f) ? mat.reflectance : 0.f;
    } else if (mat.type == BF_BSDF_ROUGHCONDUCTOR) {
        Microfacet distr = mf_make(mat);
        V3 m;
        mf_sample(distr, wi, s2x, s2y, m, bs.pdf);
        bs.wo = reflect(wi, m);
        bool active = bs.pdf != 0.f && bs.wo.z > 0.f;
        float weight;
        if (distr.sample_visible)
            weight = mf_smith_g1(distr, bs.wo, m);
        else
            weight = mf_G(distr, wi, bs.wo, m) * dot(wi, m) / (cos_theta_i * m.z);
        bs.pdf /= 4.f * dot(bs.wo, m);
        float F = fresnel_conductor(dot(wi, m), mat.eta, mat.k);
        if (mat.has_specular_reflectance) weight *= mat.reflectance;
        return active ? F * weight : 0.f;
    }
    return 0.f;
}
BF_DEV float bsdf_eval_1(const bf_material &mat, V3 wi, V3 wo) {
    float cos_theta_i = wi.z, cos_theta_o = wo.z;
    bool active = cos_theta_i > 0.f && cos_theta_o > 0.f;
    if (mat.type == BF_BSDF_DIFFUSE) {
        float value = mat.reflectance * kInvPi * cos_theta_o;
        return active ? value : 0.f;
    } else if (mat.type == BF_BSDF_ROUGHCONDUCTOR) {
        if (!active) return 0.f;
        V3 H = normalize(wo + wi);
        Microfacet distr = mf_make(mat);
        float D = mf_eval(distr, H);
        active = active && D != 0.f;
        float G = mf_G(distr, wi, wo, H);
        float result = D * G / (4.f * wi.z);
        float F = fresnel_conductor(dot(wi, H), mat.eta, mat.k);
        if (mat.has_specular_reflectance) result *= mat.reflectance;
        return active ? F * result : 0.f;
    }
    return 0.f;
}
BF_DEV float bsdf_pdf_1(const bf_material &mat, V3 wi, V3 wo) {
    float cos_theta_i = wi.z, cos_theta_o = wo.z;
    if (mat.type == BF_BSDF_DIFFUSE) {
        float pdf = kInvPi * wo.z;
        return (cos_theta_i > 0.f && cos_theta_o > 0.f) ? pdf : 0.f;
    } else if (mat.type == BF_BSDF_ROUGHCONDUCTOR) {
        V3 m = normalize(wo + wi);
        bool active = cos_theta_i > 0.f && cos_theta_o > 0.f && dot(wi, m) > 0.f && dot(wo, m) > 0.f;
        if (!active) return 0.f;
        Microfacet distr = mf_make(mat);
        if (distr.sample_visible) return mf_eval(distr, m) * mf_smith_g1(distr, wi, m) / (4.f * cos_theta_i);
        return mf_pdf(distr, wi, m) / (4.f * dot(wo, m));
    }
    return 0.f;
}
// TwoSidedBRDF with one nested BSDF on both sides: flip wi.z / wo.z
BF_DEV float bsdf_sample(const bf_material &mat, V3 wi, float s2x, float s2y, BSDFSample &bs) {
    bool flip = mat.twosided && wi.z < 0.f;
    if (mat.twosided && wi.z == 0.f) {
        bs.wo = mk(0.f, 0.f, 0.f);
        bs.pdf = 0.f;
        bs.eta = 1.f;
        return 0.f;
    }
    if (flip) wi.z *= -1.f;
    float r = bsdf_sample_1(mat, wi, s2x, s2y, bs);
    if (flip) bs.wo.z *= -1.f;
    return r;
}
BF_DEV void bsdf_eval_pdf(const bf_material &mat, V3 wi, V3 wo, float &ev, float &pdf) {
    if (mat.twosided) {
        if (wi.z == 0.f) {
            ev = pdf = 0.f;
            return;
        }
        if (wi.z < 0.f) {
            wi.z *= -1.f;
            wo.z *= -1.f;
        }
    }
    ev = bsdf_eval_1(mat, wi, wo);
    pdf = bsdf_pdf_1(mat, wi, wo);
}
BF_DEV bool bsdf_smooth(const bf_material &mat) {
    return mat.type == BF_BSDF_DIFFUSE || mat.type == BF_BSDF_ROUGHCONDUCTOR;
}

// ---------------------------------------------------------------------------
// emitters: spot.cpp:97-164, area.cpp:66-186, shape.cpp:323-356,
// rectangle.cpp:111-125; scene.cpp:180-247
// ---------------------------------------------------------------------------
struct DirSample {
    V3 d;
    float pdf, dist;
    bool delta;
};

BF_DEV float spot_falloff(CEmitter &e, V3 d) {
    float result = e.radiance;
    V3 local_dir = normalize(d);
    float cos_theta = local_dir.z;
    float beam_res = (cos_theta >= e.cos_beam) ? result : result * ((e.cutoff - acos_cr(cos_theta)) * e.inv_transition);
    return (cos_theta <= e.cos_cutoff) ? 0.f : beam_res;
}

template <int V = 0> BF_DEV float emitter_sample_direction(const DScene &sc, CEmitter &e, V3 ref_p, float sx, float sy, DirSample &ds) {
    if (rare<V>(e.type == BF_EMITTER_SPOT || e.type == BF_EMITTER_POINT)) {
        V3 p = mk(e.to_world[3], e.to_world[7], e.to_world[11]);
        ds.pdf = 1.f;
        ds.delta = true;
        ds.d = p - ref_p;
        ds.dist = norm(ds.d);
        float inv_dist = rcp(ds.dist);
        ds.d = ds.d * inv_dist;
        if (e.type == BF_EMITTER_POINT) return e.radiance * sqr(inv_dist);      // point.cpp:100-103
        V3 local_d = xf_vector(e.to_object, -ds.d);
        return spot_falloff(e, local_d) * (inv_dist * inv_dist);
    } else {
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
        float spec = e.radiance / ds.pdf;
        return active ? spec : 0.f;
    }
}

// pdf_emitter_direction for the hit `p_hit` (normal n_hit) seen from `p_ref`
template <int V = 0> BF_DEV float emitter_pdf_direction(const DScene &sc, CEmitter &e, V3 p_ref, V3 p_hit, V3 n_hit) {
    if (rare<V>(e.type == BF_EMITTER_SPOT || e.type == BF_EMITTER_POINT)) return 0.f;
    CRect &rc = c_rects(sc)[e.rect];
    V3 d = p_hit - p_ref;
    float dist = norm(d);
    d = d / dist;
    float dp = dot(d, n_hit);
    float pdf = rc.inv_area, adp = __builtin_fabsf(dot(d, n_hit));
    pdf *= (adp != 0.f) ? (dist * dist) / adp : 0.f;
    return (dp < 0.f) ? pdf : 0.f;
}

BF_DEV float mis_weight(float pdf_a, float pdf_b) {   // path.cpp:222-226
    pdf_a *= pdf_a;
    pdf_b *= pdf_b;
    return pdf_a > 0.f ? pdf_a / (pdf_a + pdf_b) : 0.f;
}

// spectrum.h:281-287 applied to a grey colour: M * (l,l,l)
BF_DEV void srgb_to_xyz_grey(float l, float &X, float &Y, float &Z) {
    X = fmadd(0.180423f, l, fmadd(0.357580f, l, 0.412453f * l));
    Y = fmadd(0.072169f, l, fmadd(0.715160f, l, 0.212671f * l));
    Z = fmadd(0.950227f, l, fmadd(0.119193f, l, 0.019334f * l));
}

// sensor rays: fluxmeter.cpp:63-85, perspective.cpp:172-199
template <int V = 0>
BF_DEV float sensor_sample_ray(const DScene &sc, float px, float py, float ax, float ay, V3 &o, V3 &d, float &mint,
                               float &maxt) {
    CSensor &s = c_sensor(sc);
    if (rare<V>(s.type == BF_SENSOR_FLUXMETER || s.type == BF_SENSOR_IRRADIANCEMETER)) {
        CRect &rc = c_rects(sc)[s.rect];
        o = xf_point(rc.to_world, mk(px * 2.f - 1.f, py * 2.f - 1.f, 0.f));
        V3 local = square_to_cosine_hemisphere(ax, ay);
        Frame f;
        f.n = mk(rc.n[0], rc.n[1], rc.n[2]);
        coordinate_system(f.n, f.s, f.t);
        d = to_world(f, local);
        mint = kRayEpsilon;
        maxt = BF_INF;
        return 1.f * kPi;
    } else if (rare<V>(s.type == BF_SENSOR_RADIANCEMETER)) {     // radiancemeter.cpp:91-108: position and aperture samples unused
        o = xf_point(s.to_world, mk(0.f, 0.f, 0.f));
        d = xf_vector(s.to_world, mk(0.f, 0.f, 1.f));
        mint = kRayEpsilon;
        maxt = BF_INF;
        return 1.f;
    } else {
        V3 near_p = xf_point_proj(s.sample_to_camera, mk(px, py, 0.f));
        V3 dl = normalize(near_p);
        float inv_z = rcp(dl.z);
        mint = s.near_clip * inv_z;
        maxt = s.far_clip * inv_z;
        o = xf_point(s.to_world, mk(0.f, 0.f, 0.f));
        d = xf_vector(s.to_world, dl);
        return 1.f;
    }
}

}  // namespace bfd
