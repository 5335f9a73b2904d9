// Test harness (tests/ only): builds the product's host-side BVH (beifong_amd/csrc/bf_bvh.cpp) for a triangle soup
// and checks its structural invariants on the CPU.  Compiled by tests/test_bvh_host.py with g++.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../beifong_amd/csrc/bf_bvh.h"

namespace {
struct Ctx {
    const std::vector<bf::BuildTri> *tris;
    const bf::BVH *bvh2;
    const bf::BVH4 *bvh4;
    std::vector<uint32_t> seen;      // per leaf slot: times referenced
    uint32_t max_leaf = 0, n_leaves = 0, n_internal = 0, max_depth = 0, max_stack = 0;
    int error = 0;
};

bool tri_in_box(const bf::BuildTri &t, const float lo[3], const float hi[3]) {
    const float *p[3] = {t.p0, t.p1, t.p2};
    for (int v = 0; v < 3; ++v)
        for (int k = 0; k < 3; ++k)
            if (!(p[v][k] >= lo[k] && p[v][k] <= hi[k])) return false;
    return true;
}

// every triangle of subtree `ref` lies inside [lo, hi]; returns the worst-case stack entries below
uint32_t walk(Ctx &c, int32_t ref, const float lo[3], const float hi[3], uint32_t depth) {
    c.max_depth = std::max(c.max_depth, depth);
    if (ref < 0) {
        uint32_t enc = ~(uint32_t) ref, first = enc >> 3, cnt = (enc & 7u) + 1u;
        c.max_leaf = std::max(c.max_leaf, cnt);
        ++c.n_leaves;
        for (uint32_t i = 0; i < cnt; ++i) {
            if (first + i >= c.seen.size()) {
                c.error = 2;
                return 0;
            }
            ++c.seen[first + i];
            const bf::BuildTri &t = (*c.tris)[c.bvh2->order[first + i]];
            if (!tri_in_box(t, lo, hi)) c.error = 3;
        }
        return 0;
    }
    if ((size_t) ref >= c.bvh4->nodes.size()) {
        c.error = 4;
        return 0;
    }
    ++c.n_internal;
    const bf::Node4 &n = c.bvh4->nodes[(size_t) ref];
    uint32_t used = 0, below = 0;
    for (int k = 0; k < 4; ++k) {
        if (n.child[k] == bf::kEmptyChild) {
            if (k < 2) c.error = 5;      // slots 0 and 1 are always used
            continue;
        }
        ++used;
        float clo[3] = {n.lox[k], n.loy[k], n.loz[k]}, chi[3] = {n.hix[k], n.hiy[k], n.hiz[k]};
        // a child box may stick out of its parent's by the padding only
        for (int a = 0; a < 3; ++a) {
            float pad = 1e-5f * std::max({1.f, std::fabs(lo[a]), std::fabs(hi[a])});
            if (clo[a] < lo[a] - pad || chi[a] > hi[a] + pad) c.error = 6;
        }
        below = std::max(below, walk(c, n.child[k], clo, chi, depth + 1));
    }
    uint32_t need = used - 1 + below;
    c.max_stack = std::max(c.max_stack, need);
    return need;
}
}  // namespace

extern "C" int bvh_check(uint32_t n, const float *verts /* n*9 */, uint32_t out[8]) {
    std::vector<bf::BuildTri> tris(n);
    for (uint32_t i = 0; i < n; ++i) {
        std::memcpy(tris[i].p0, verts + 9 * i, 12);
        std::memcpy(tris[i].p1, verts + 9 * i + 3, 12);
        std::memcpy(tris[i].p2, verts + 9 * i + 6, 12);
    }
    bf::BVH bvh2;
    auto t0 = std::chrono::steady_clock::now();
    bf::build_bvh(tris, bvh2, 0.f);
    auto t1 = std::chrono::steady_clock::now();
    bf::BVH4 bvh4;
    bf::collapse_bvh4(bvh2, bvh4);
    auto t2 = std::chrono::steady_clock::now();
    if (getenv("BVH_CHECK_TIMING"))
        fprintf(stderr, "build %.1f ms, collapse %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(t2 - t1).count());
    Ctx c;
    c.tris = &tris;
    c.bvh2 = &bvh2;
    c.bvh4 = &bvh4;
    c.seen.assign(n, 0);
    if (n) {
        // permutation
        std::vector<uint32_t> o = bvh2.order;
        std::sort(o.begin(), o.end());
        for (uint32_t i = 0; i < n; ++i)
            if (o[i] != i) return 1;
        uint32_t need = walk(c, bvh4.root_child, bvh2.lo, bvh2.hi, 1);
        if (c.error) return c.error;
        for (uint32_t i = 0; i < n; ++i)
            if (c.seen[i] != 1) return 7;
        if (need != bvh4.stack_need && bvh4.root_child >= 0) return 8;
    }
    out[0] = (uint32_t) bvh4.nodes.size();
    out[1] = c.n_leaves;
    out[2] = c.max_leaf;
    out[3] = c.max_depth;
    out[4] = bvh4.stack_need;
    out[5] = bvh2.max_depth;
    out[6] = (uint32_t) bvh2.nodes.size();
    out[7] = (uint32_t) bf::kMaxLeaf;
    return 0;
}

// Sixteen-wide collapse (bf::collapse_bvh16): every triangle slot in exactly one leaf of at most 16 contiguous slots,
// triangles inside their leaf's box, child boxes inside their parent's (up to the padding), reported stack need =
// the worst case of "push every hit child" (sum over levels of the children count).
namespace {
struct Ctx16 {
    const std::vector<bf::BuildTri> *tris;
    const bf::BVH *bvh2;
    const bf::BVH16 *w;
    std::vector<uint32_t> seen;
    uint32_t max_leaf = 0, n_leaves = 0, n_internal = 0, max_depth = 0;
    int error = 0;
};
uint32_t walk16(Ctx16 &c, int32_t ref, const float lo[3], const float hi[3], uint32_t depth) {
    c.max_depth = std::max(c.max_depth, depth);
    if (ref < 0) {
        const uint32_t enc = ~(uint32_t) ref, first = enc >> 4, cnt = (enc & 15u) + 1u;
        c.max_leaf = std::max(c.max_leaf, cnt);
        ++c.n_leaves;
        for (uint32_t i = 0; i < cnt; ++i) {
            if (first + i >= c.seen.size()) {
                c.error = 12;
                return 0;
            }
            ++c.seen[first + i];
            if (!tri_in_box((*c.tris)[c.bvh2->order[first + i]], lo, hi)) c.error = 13;
        }
        return 0;
    }
    if ((size_t) ref >= c.w->nodes.size()) {
        c.error = 14;
        return 0;
    }
    ++c.n_internal;
    const bf::Node16 &n = c.w->nodes[(size_t) ref];
    uint32_t used = 0, below = 0;
    for (int k = 0; k < 16; ++k) {
        int32_t child;
        std::memcpy(&child, &n.c[k][6], 4);
        if (child == bf::kEmptyChild) {
            if (k < 2) c.error = 15;
            continue;
        }
        ++used;
        const float *clo = &n.c[k][0], *chi = &n.c[k][3];
        for (int a = 0; a < 3; ++a) {
            float pad = 1e-5f * std::max({1.f, std::fabs(lo[a]), std::fabs(hi[a])});
            if (clo[a] < lo[a] - pad || chi[a] > hi[a] + pad) c.error = 16;
        }
        below = std::max(below, walk16(c, child, clo, chi, depth + 1));
    }
    return used + below;
}
}  // namespace

extern "C" int bvh16_check(uint32_t n, const float *verts /* n*9 */, uint32_t out[8]) {
    std::vector<bf::BuildTri> tris(n);
    for (uint32_t i = 0; i < n; ++i) {
        std::memcpy(tris[i].p0, verts + 9 * i, 12);
        std::memcpy(tris[i].p1, verts + 9 * i + 3, 12);
        std::memcpy(tris[i].p2, verts + 9 * i + 6, 12);
    }
    bf::BVH bvh2;
    bf::build_bvh(tris, bvh2, 0.f);
    bf::BVH16 w;
    bf::collapse_bvh16(bvh2, w);
    Ctx16 c;
    c.tris = &tris;
    c.bvh2 = &bvh2;
    c.w = &w;
    c.seen.assign(n, 0);
    uint32_t need = 0;
    if (n) {
        need = walk16(c, w.root_child, bvh2.lo, bvh2.hi, 1);
        if (c.error) return c.error;
        for (uint32_t i = 0; i < n; ++i)
            if (c.seen[i] != 1) return 17;
        if (need != w.stack_need) return 18;
    } else if (w.root_child != bf::kEmptyChild) {
        return 19;
    }
    out[0] = (uint32_t) w.nodes.size();
    out[1] = c.n_leaves;
    out[2] = c.max_leaf;
    out[3] = c.max_depth;
    out[4] = w.stack_need;
    out[5] = c.n_internal;
    out[6] = 0;
    out[7] = 0;
    return 0;
}


// Quantised four-wide nodes (bf::quantise_bvh4, wf_trace's 64-byte nodes): every quantised child box, evaluated as the
// kernels do (fl32(lo + q * 2^(e - 127))), contains the fp32 child box; same child references; unused slots inverted.
extern "C" int bvh4q_check(uint32_t n, const float *verts /* n*9 */, uint32_t out[8]) {
    std::vector<bf::BuildTri> tris(n);
    for (uint32_t i = 0; i < n; ++i) {
        std::memcpy(tris[i].p0, verts + 9 * i, 12);
        std::memcpy(tris[i].p1, verts + 9 * i + 3, 12);
        std::memcpy(tris[i].p2, verts + 9 * i + 6, 12);
    }
    bf::BVH bvh2;
    bf::build_bvh(tris, bvh2, 0.f);
    bf::BVH4 bvh4;
    bf::collapse_bvh4(bvh2, bvh4);
    std::vector<bf::Node4Q> q;
    bf::quantise_bvh4(bvh4, q);
    if (q.size() != bvh4.nodes.size()) return 21;
    double worst = 0.0;       // largest growth of a box side relative to the node's extent
    for (size_t i = 0; i < q.size(); ++i) {
        const bf::Node4 &a = bvh4.nodes[i];
        const bf::Node4Q &b = q[i];
        const float *clo[3] = {a.lox, a.loy, a.loz}, *chi[3] = {a.hix, a.hiy, a.hiz};
        for (int k = 0; k < 4; ++k) {
            if (b.child[k] != a.child[k]) return 22;
            for (int ax = 0; ax < 3; ++ax) {
                const uint32_t e = (b.exps >> (8 * ax)) & 0xffu;
                uint32_t bits = e << 23;
                float s;
                std::memcpy(&s, &bits, 4);
                const float ql = (float) ((b.qlo[ax] >> (8 * k)) & 0xffu), qh = (float) ((b.qhi[ax] >> (8 * k)) & 0xffu);
                if (a.child[k] == bf::kEmptyChild) {
                    if (!(ql == 255.f && qh == 0.f)) return 23;
                    continue;
                }
                const float plo = b.lo[ax] + ql * s, phi = b.lo[ax] + qh * s;
                if (!(plo <= clo[ax][k]) || !(phi >= chi[ax][k])) return 24;
                const double grow = std::max((double) clo[ax][k] - plo, (double) phi - chi[ax][k]);
                if (grow > 1.0001 * (double) s) return 25;             // never more than one quantum
                worst = std::max(worst, grow / std::max(1e-30, 255.0 * (double) s));
            }
        }
    }
    out[0] = (uint32_t) q.size();
    out[1] = (uint32_t) (worst * 1e6);
    return 0;
}
