// Binned-SAH BVH2 builder (host).  See bf_bvh.h for the node layout and why a
// BVH replaces the reference's kd-tree (kdtree.h) on MI355X.
#include "bf_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <atomic>
#include <future>
#include <limits>

namespace bf {
namespace {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int i = 0; i < 3; ++i) {
            lo[i] = std::numeric_limits<float>::infinity();
            hi[i] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const Box &b) {
        for (int i = 0; i < 3; ++i) {
            lo[i] = std::min(lo[i], b.lo[i]);
            hi[i] = std::max(hi[i], b.hi[i]);
        }
    }
    void grow(const float *p) {
        for (int i = 0; i < 3; ++i) {
            lo[i] = std::min(lo[i], p[i]);
            hi[i] = std::max(hi[i], p[i]);
        }
    }
    float half_area() const {
        float d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        if (d[0] < 0 || d[1] < 0 || d[2] < 0) return 0.f;
        return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
    }
};

struct Builder {
    const std::vector<BuildTri> &tris;
    std::vector<Box> tb;          // per-triangle bounds
    std::vector<float> cent;      // per-triangle centroid [3n]
    std::vector<uint32_t> &order;
    std::atomic<uint32_t> max_depth{0};
    float abs_pad = 0.f;          // origin-rounding allowance (see bf_bvh.h)

    Builder(const std::vector<BuildTri> &t, BVH &out) : tris(t), order(out.order) {}

    // Subtrees of at least kParallelMin triangles are built by their own
    // task into their own node vector and spliced in afterwards (child indices shifted), so the node order —
    // hence the device layout — is the same depth-first order whatever the thread schedule.
    static constexpr uint32_t kParallelMin = 8192;
    std::atomic<int> task_budget{48};       // concurrent subtree tasks (SAH trees can be lopsided: no depth limit)
    static int32_t splice(std::vector<Node> &out, std::vector<Node> &sub, int32_t ref) {
        if (ref < 0) return ref;
        const int32_t off = (int32_t) out.size();
        for (Node &n : sub) {
            if (n.child[0] >= 0) n.child[0] += off;
            if (n.child[1] >= 0) n.child[1] += off;
        }
        out.insert(out.end(), sub.begin(), sub.end());
        return ref + off;
    }
    void note_depth(uint32_t d) {
        uint32_t cur = max_depth.load(std::memory_order_relaxed);
        while (d > cur && !max_depth.compare_exchange_weak(cur, d, std::memory_order_relaxed)) {
        }
    }

    void pad(Box &b) const {
        // fp32 hit distances can land a hair outside the exact box: widen by a
        // relative epsilon of the box scale (and of its distance from the origin)
        float m = 0.f;
        for (int i = 0; i < 3; ++i) m = std::max({m, b.hi[i] - b.lo[i], std::fabs(b.lo[i]), std::fabs(b.hi[i])});
        float e = 2e-6f * m + abs_pad + 1e-30f;
        for (int i = 0; i < 3; ++i) {
            b.lo[i] -= e;
            b.hi[i] += e;
        }
    }

    Box range_bounds(uint32_t first, uint32_t count) const {
        Box b;
        b.reset();
        for (uint32_t i = 0; i < count; ++i) b.grow(tb[order[first + i]]);
        return b;
    }

    // returns the child reference for [first, first+count)
    int32_t build(std::vector<Node> &nodes, uint32_t first, uint32_t count, const Box &bounds, uint32_t depth) {
        note_depth(depth);
        if (count <= (uint32_t) kMaxLeaf) return ~(int32_t) ((first << 3) | (count - 1));

        Box cb;
        cb.reset();
        for (uint32_t i = 0; i < count; ++i) cb.grow(&cent[3 * order[first + i]]);

#ifndef BF_SAH_BINS
#define BF_SAH_BINS 16
#endif
        constexpr int NB = BF_SAH_BINS;
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_bin = -1;
        // The traversal kernels keep a kStackDepth-entry stack per lane in LDS:
        // once the remaining budget only fits a balanced subtree, fall back to
        // median splits so the depth bound holds by construction.
        uint32_t need = 0;
        while ((uint32_t) kMaxLeaf << need < count) ++need;
        const bool force_median = depth + need + 1 >= (uint32_t) kMaxDepth;
        for (int ax = 0; ax < 3 && !force_median; ++ax) {
            float ext = cb.hi[ax] - cb.lo[ax];
            if (!(ext > 0.f)) continue;
            Box bb[NB];
            uint32_t bc[NB];
            for (int b = 0; b < NB; ++b) {
                bb[b].reset();
                bc[b] = 0;
            }
            float scale = NB / ext;
            for (uint32_t i = 0; i < count; ++i) {
                uint32_t t = order[first + i];
                int b = std::min(NB - 1, std::max(0, (int) ((cent[3 * t + ax] - cb.lo[ax]) * scale)));
                bb[b].grow(tb[t]);
                bc[b]++;
            }
            float right_area[NB];
            uint32_t right_cnt[NB];
            Box acc;
            acc.reset();
            uint32_t cnt = 0;
            for (int b = NB - 1; b > 0; --b) {
                acc.grow(bb[b]);
                cnt += bc[b];
                right_area[b] = acc.half_area();
                right_cnt[b] = cnt;
            }
            acc.reset();
            cnt = 0;
            for (int b = 0; b < NB - 1; ++b) {
                acc.grow(bb[b]);
                cnt += bc[b];
                if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                float cost = acc.half_area() * cnt + right_area[b + 1] * right_cnt[b + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = ax;
                    best_bin = b;
                }
            }
        }
        uint32_t mid;
        if (force_median) {
            int ax = 0;
            for (int k = 1; k < 3; ++k)
                if (cb.hi[k] - cb.lo[k] > cb.hi[ax] - cb.lo[ax]) ax = k;
            mid = count / 2;
            std::nth_element(order.begin() + first, order.begin() + first + mid, order.begin() + first + count,
                             [&](uint32_t a, uint32_t b) { return cent[3 * a + ax] < cent[3 * b + ax]; });
        } else if (best_axis < 0) {
            mid = count / 2;   // all centroids coincide: split by index
        } else {
            float ext = cb.hi[best_axis] - cb.lo[best_axis];
            float scale = NB / ext;
            auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) {
                int b = std::min(NB - 1, std::max(0, (int) ((cent[3 * t + best_axis] - cb.lo[best_axis]) * scale)));
                return b <= best_bin;
            });
            mid = (uint32_t) (it - (order.begin() + first));
            if (mid == 0 || mid == count) mid = count / 2;
        }
        Box lb = range_bounds(first, mid), rb = range_bounds(first + mid, count - mid);
        int32_t me = (int32_t) nodes.size();
        nodes.push_back(Node());
        int32_t l, r;
        bool parallel = std::min(mid, count - mid) >= kParallelMin;
        if (parallel && task_budget.fetch_sub(1, std::memory_order_relaxed) <= 0) {
            task_budget.fetch_add(1, std::memory_order_relaxed);
            parallel = false;
        }
        if (parallel) {
            std::vector<Node> ln, rn;
            auto fut = std::async(std::launch::async, [&]() { return build(ln, first, mid, lb, depth + 1); });
            const int32_t rr = build(rn, first + mid, count - mid, rb, depth + 1);
            const int32_t ll = fut.get();
            l = splice(nodes, ln, ll);
            r = splice(nodes, rn, rr);
            task_budget.fetch_add(1, std::memory_order_relaxed);
        } else {
            l = build(nodes, first, mid, lb, depth + 1);
            r = build(nodes, first + mid, count - mid, rb, depth + 1);
        }
        pad(lb);
        pad(rb);
        Node &n = nodes[me];
        for (int i = 0; i < 3; ++i) {
            n.c0lo[i] = lb.lo[i];
            n.c0hi[i] = lb.hi[i];
            n.c1lo[i] = rb.lo[i];
            n.c1hi[i] = rb.hi[i];
        }
        n.child[0] = l;
        n.child[1] = r;
        n.pad[0] = n.pad[1] = 0;
        (void) bounds;
        return me;
    }
};

}  // namespace

void build_bvh(const std::vector<BuildTri> &tris, BVH &out, float origin_scale) {
    out.nodes.clear();
    out.order.resize(tris.size());
    out.max_depth = 0;
    for (int i = 0; i < 3; ++i) {
        out.lo[i] = 0;
        out.hi[i] = 0;
    }
    out.root_child = 0;
    if (tris.empty()) return;
    Builder b(tris, out);
    size_t n = tris.size();
    b.tb.resize(n);
    b.cent.resize(3 * n);
    Box all;
    all.reset();
    for (size_t i = 0; i < n; ++i) {
        out.order[i] = (uint32_t) i;
        Box &x = b.tb[i];
        x.reset();
        x.grow(tris[i].p0);
        x.grow(tris[i].p1);
        x.grow(tris[i].p2);
        for (int k = 0; k < 3; ++k) b.cent[3 * i + k] = 0.5f * (x.lo[k] + x.hi[k]);
        all.grow(x);
    }
    out.nodes.reserve(n);
    float scale = origin_scale;
    for (int k = 0; k < 3; ++k) scale = std::max({scale, std::fabs(all.lo[k]), std::fabs(all.hi[k])});
    b.abs_pad = 2e-7f * scale;
    out.root_child = b.build(out.nodes, 0, (uint32_t) n, all, 0);
    b.pad(all);
    for (int i = 0; i < 3; ++i) {
        out.lo[i] = all.lo[i];
        out.hi[i] = all.hi[i];
    }
    out.max_depth = b.max_depth.load();
}

namespace {

struct ChildRef {
    int32_t ref;        // BVH2 child encoding
    float lo[3], hi[3];
    float area() const {
        float d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
    }
};

void children_of(const Node &n, ChildRef out[2]) {
    for (int k = 0; k < 3; ++k) {
        out[0].lo[k] = n.c0lo[k];
        out[0].hi[k] = n.c0hi[k];
        out[1].lo[k] = n.c1lo[k];
        out[1].hi[k] = n.c1hi[k];
    }
    out[0].ref = n.child[0];
    out[1].ref = n.child[1];
}

struct Collapser {
    const BVH &in;
    BVH4 &out;
    int width;
    // returns the stack need of the subtree; writes the node at index `self`
    uint32_t emit(int32_t node2, uint32_t self, uint32_t depth) {
        out.max_depth = std::max(out.max_depth, depth);
        ChildRef c[4];
        int n = 2;
        children_of(in.nodes[(size_t) node2], c);
        while (n < width) {
            int best = -1;
            float best_area = -1.f;
            for (int i = 0; i < n; ++i)
                if (c[i].ref >= 0 && c[i].area() > best_area) {
                    best_area = c[i].area();
                    best = i;
                }
            if (best < 0) break;
            ChildRef g[2];
            children_of(in.nodes[(size_t) c[best].ref], g);
            c[best] = g[0];
            c[n++] = g[1];
        }
        int32_t refs[4];
        uint32_t need_below = 0;
        for (int i = 0; i < n; ++i) {
            if (c[i].ref >= 0) {
                uint32_t idx = (uint32_t) out.nodes.size();
                out.nodes.emplace_back();
                refs[i] = (int32_t) idx;
                need_below = std::max(need_below, emit(c[i].ref, idx, depth + 1));
            } else {
                refs[i] = c[i].ref;
            }
        }
        Node4 &w = out.nodes[self];
        const float inf = std::numeric_limits<float>::infinity();
        for (int i = 0; i < 4; ++i) {
            bool used = i < n;
            w.lox[i] = used ? c[i].lo[0] : inf;
            w.loy[i] = used ? c[i].lo[1] : inf;
            w.loz[i] = used ? c[i].lo[2] : inf;
            w.hix[i] = used ? c[i].hi[0] : -inf;
            w.hiy[i] = used ? c[i].hi[1] : -inf;
            w.hiz[i] = used ? c[i].hi[2] : -inf;
            w.child[i] = used ? refs[i] : kEmptyChild;
            w.pad[i] = 0;
        }
        return (uint32_t) (n - 1) + need_below;
    }
};

}  // namespace

void collapse_bvh4(const BVH &in, BVH4 &out) {
    out.nodes.clear();
    out.root_child = in.root_child;
    out.stack_need = 0;
    out.max_depth = 0;
    if (in.nodes.empty() || in.root_child < 0) return;
    out.nodes.reserve(in.nodes.size() / 2 + 1);
    out.nodes.emplace_back();
    Collapser c{in, out, 4};
    out.root_child = 0;
    out.stack_need = c.emit(in.root_child, 0, 1);      // <= 3 * kMaxDepth (the binary depth bound)

    // top levels first, breadth-first (BVH4 comment); child references follow the move
    const size_t n = out.nodes.size();
    std::vector<int32_t> new_of(n, -1);
    std::vector<int32_t> order;
    order.reserve(n);
    order.push_back(0);
    new_of[0] = 0;
    for (size_t head = 0; head < order.size() && order.size() < kTopNodes; ++head)
        for (int i = 0; i < 4 && order.size() < kTopNodes; ++i) {
            const int32_t ch = out.nodes[(size_t) order[head]].child[i];
            if (ch >= 0) {
                new_of[(size_t) ch] = (int32_t) order.size();
                order.push_back(ch);
            }
        }
    for (size_t i = 0; i < n; ++i)
        if (new_of[i] < 0) {
            new_of[i] = (int32_t) order.size();
            order.push_back((int32_t) i);
        }
    std::vector<Node4> moved(n);
    for (size_t k = 0; k < n; ++k) {
        Node4 w = out.nodes[(size_t) order[k]];
        for (int i = 0; i < 4; ++i)
            if (w.child[i] >= 0) w.child[i] = new_of[(size_t) w.child[i]];
        moved[k] = w;
    }
    out.nodes.swap(moved);
}

namespace {

struct Collapser16 {
    const BVH &in;
    BVH16 &out;
    std::vector<uint32_t> first, count;      // triangle range of every binary node's subtree (contiguous by construction)

    void range_of(int32_t ref, uint32_t &f, uint32_t &c) const {
        if (ref >= 0) {
            f = first[(size_t) ref];
            c = count[(size_t) ref];
        } else {
            const uint32_t enc = ~(uint32_t) ref;
            f = enc >> 3;
            c = (enc & 7u) + 1u;
        }
    }
    void ranges(int32_t node) {
        const Node &n = in.nodes[(size_t) node];
        for (int k = 0; k < 2; ++k)
            if (n.child[k] >= 0) ranges(n.child[k]);
        uint32_t f0, c0, f1, c1;
        range_of(n.child[0], f0, c0);
        range_of(n.child[1], f1, c1);
        first[(size_t) node] = std::min(f0, f1);
        count[(size_t) node] = c0 + c1;
    }
    bool expandable(const ChildRef &c) const { return c.ref >= 0 && count[(size_t) c.ref] > kWideLeaf; }

    uint32_t emit(int32_t node2, uint32_t self, uint32_t depth) {
        out.max_depth = std::max(out.max_depth, depth);
        ChildRef c[16];
        int n = 2;
        children_of(in.nodes[(size_t) node2], c);
        while (n < 16) {
            int best = -1;
            float best_area = -1.f;
            for (int i = 0; i < n; ++i)
                if (expandable(c[i]) && c[i].area() > best_area) {
                    best_area = c[i].area();
                    best = i;
                }
            if (best < 0) break;
            ChildRef g[2];
            children_of(in.nodes[(size_t) c[best].ref], g);
            c[best] = g[0];
            c[n++] = g[1];
        }
        int32_t refs[16];
        uint32_t need_below = 0;
        for (int i = 0; i < n; ++i) {
            if (expandable(c[i])) {
                uint32_t idx = (uint32_t) out.nodes.size();
                out.nodes.emplace_back();
                refs[i] = (int32_t) idx;
                need_below = std::max(need_below, emit(c[i].ref, idx, depth + 1));
            } else {
                uint32_t f, k;
                range_of(c[i].ref, f, k);
                refs[i] = ~(int32_t) ((f << 4) | (k - 1u));
            }
        }
        Node16 &w = out.nodes[self];
        const float inf = std::numeric_limits<float>::infinity();
        for (int i = 0; i < 16; ++i) {
            const bool used = i < n;
            float *q = w.c[i];
            q[0] = used ? c[i].lo[0] : inf;
            q[1] = used ? c[i].lo[1] : inf;
            q[2] = used ? c[i].lo[2] : inf;
            q[3] = used ? c[i].hi[0] : -inf;
            q[4] = used ? c[i].hi[1] : -inf;
            q[5] = used ? c[i].hi[2] : -inf;
            const int32_t r = used ? refs[i] : kEmptyChild;
            std::memcpy(&q[6], &r, 4);
            q[7] = 0.f;
        }
        return (uint32_t) n + need_below;
    }
};

}  // namespace

void collapse_bvh16(const BVH &in, BVH16 &out) {
    out.nodes.clear();
    out.root_child = kEmptyChild;
    out.stack_need = 0;
    out.max_depth = 0;
    if (in.order.empty()) return;
    if (in.nodes.empty() || in.root_child < 0) {
        // a single binary leaf (<= kMaxLeaf triangles)
        const uint32_t enc = ~(uint32_t) in.root_child;
        out.root_child = ~(int32_t) (((enc >> 3) << 4) | (enc & 7u));
        return;
    }
    Collapser16 c{in, out, std::vector<uint32_t>(in.nodes.size(), 0u), std::vector<uint32_t>(in.nodes.size(), 0u)};
    c.ranges(in.root_child);
    if (c.count[(size_t) in.root_child] <= kWideLeaf) {
        out.root_child = ~(int32_t) ((c.first[(size_t) in.root_child] << 4) | (c.count[(size_t) in.root_child] - 1u));
        return;
    }
    out.nodes.reserve(in.nodes.size() / 8 + 1);
    out.nodes.emplace_back();
    out.root_child = 0;
    out.stack_need = c.emit(in.root_child, 0, 1);
}

// Quantisation of one four-wide node (bf_bvh.h: Node4Q).  Plain fp32 operations in a fixed order — the device repeats
// them when it re-fits the tree after a mesh translation (bf_kernels.hip: bf_translate_kernel).
void quantise_node4(const Node4 &in, Node4Q &out) {
    const float inf = std::numeric_limits<float>::infinity();
    float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    const float *clo[3] = {in.lox, in.loy, in.loz}, *chi[3] = {in.hix, in.hiy, in.hiz};
    for (int k = 0; k < 4; ++k) {
        if (in.child[k] == kEmptyChild) continue;
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], clo[a][k]);
            hi[a] = std::max(hi[a], chi[a][k]);
        }
    }
    uint32_t exps = 0;
    float scale[3];
    for (int a = 0; a < 3; ++a) {
        out.lo[a] = lo[a];
        // smallest power of two s with 255 s >= extent (one binade of head room for the rounding of lo + q s)
        const float ext = hi[a] - lo[a];
        int e = 0;
        (void) std::frexp(ext * (1.f / 255.f), &e);          // ext / 255 = m 2^e, m in [0.5, 1)  =>  2^e > ext / 255
        e = std::max(-100, std::min(100, e));
        if (!(lo[a] + 255.f * std::ldexp(1.f, e) >= hi[a])) ++e;      // the top plane must be reachable as the kernels round it
        scale[a] = std::ldexp(1.f, e);
        exps |= (uint32_t) (e + 127) << (8 * a);
    }
    out.exps = exps;
    for (int a = 0; a < 3; ++a) out.qlo[a] = out.qhi[a] = 0;
    for (int k = 0; k < 4; ++k) {
        out.child[k] = in.child[k];
        for (int a = 0; a < 3; ++a) {
            uint32_t ql = 255u, qh = 0u;
            if (in.child[k] != kEmptyChild) {
                const float inv = 1.f / scale[a];                         // exact: a power of two
                float fl = std::floor((clo[a][k] - lo[a]) * inv), fh = std::ceil((chi[a][k] - lo[a]) * inv);
                fl = std::min(255.f, std::max(0.f, fl));
                fh = std::min(255.f, std::max(0.f, fh));
                // containment as the kernels evaluate the planes: fl32(lo + q s)
                while (fl > 0.f && lo[a] + fl * scale[a] > clo[a][k]) fl -= 1.f;
                while (fh < 255.f && lo[a] + fh * scale[a] < chi[a][k]) fh += 1.f;
                ql = (uint32_t) fl;
                qh = (uint32_t) fh;
            }
            out.qlo[a] |= ql << (8 * k);
            out.qhi[a] |= qh << (8 * k);
        }
    }
    out.pad[0] = out.pad[1] = 0;
}

void quantise_bvh4(const BVH4 &in, std::vector<Node4Q> &out) {
    out.resize(in.nodes.size());
    for (size_t i = 0; i < in.nodes.size(); ++i) quantise_node4(in.nodes[i], out[i]);
}

}  // namespace bf
