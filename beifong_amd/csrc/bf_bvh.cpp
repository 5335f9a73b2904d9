// Binned-SAH BVH2 builder (host).  See bf_bvh.h for the node layout and why a
// BVH replaces the reference's kd-tree (kdtree.h) on MI355X.
#include "bf_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
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
    std::vector<Node> &nodes;
    uint32_t max_depth = 0;

    Builder(const std::vector<BuildTri> &t, BVH &out) : tris(t), order(out.order), nodes(out.nodes) {}

    static void pad(Box &b) {
        // fp32 hit distances can land a hair outside the exact box: widen by a
        // relative epsilon of the box scale (and of its distance from the origin)
        float m = 0.f;
        for (int i = 0; i < 3; ++i) m = std::max({m, b.hi[i] - b.lo[i], std::fabs(b.lo[i]), std::fabs(b.hi[i])});
        float e = 2e-6f * m + 1e-30f;
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
    int32_t build(uint32_t first, uint32_t count, const Box &bounds, uint32_t depth) {
        max_depth = std::max(max_depth, depth);
        if (count <= (uint32_t) kMaxLeaf) return ~(int32_t) ((first << 3) | (count - 1));

        Box cb;
        cb.reset();
        for (uint32_t i = 0; i < count; ++i) cb.grow(&cent[3 * order[first + i]]);

        constexpr int NB = 16;
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
        int32_t l = build(first, mid, lb, depth + 1);
        int32_t r = build(first + mid, count - mid, rb, depth + 1);
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

void build_bvh(const std::vector<BuildTri> &tris, BVH &out) {
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
    out.root_child = b.build(0, (uint32_t) n, all, 0);
    Builder::pad(all);
    for (int i = 0; i < 3; ++i) {
        out.lo[i] = all.lo[i];
        out.hi[i] = all.hi[i];
    }
    out.max_depth = b.max_depth;
}

}  // namespace bf
