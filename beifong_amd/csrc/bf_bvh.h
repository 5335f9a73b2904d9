// Host-side BVH construction for the HIP traversal kernels.
//
// The reference accelerates Scene::ray_intersect with an SAH kd-tree
// (include/mitsuba/render/kdtree.h, built in Scene::accel_init_cpu,
// src/librender/scene_native.inl:3-10) or embree/OptiX.  Closest-hit results do
// not depend on the accelerator (ties are resolved by primitive index, see
// bf_kernels.hip), so the MI355X build uses a structure that suits wave64
// pointer chasing instead: a binned-SAH binary BVH (64-byte nodes holding BOTH
// child boxes), collapsed into four-wide 128-byte nodes (Node4) for the device,
// so one L2-line fetch decides four subtrees.
#pragma once
#include <cstdint>
#include <vector>

namespace bf {

struct BuildTri {
    float p0[3], p1[3], p2[3];
};

// 64-byte node, 16-byte aligned; read by the kernels as four float4.
//   q0 = (c0.lo.x, c0.lo.y, c0.lo.z, c0.hi.x)
//   q1 = (c0.hi.y, c0.hi.z, c1.lo.x, c1.lo.y)
//   q2 = (c1.lo.z, c1.hi.x, c1.hi.y, c1.hi.z)
//   q3 = (child0, child1, 0, 0)  as int bits
// child >= 0: index of an internal node; child < 0: leaf, ~child =
// (first_triangle << 3) | (count - 1), triangles stored contiguously in leaf
// order.
struct alignas(16) Node {
    float c0lo[3], c0hi[3], c1lo[3], c1hi[3];
    int32_t child[2];
    int32_t pad[2];
};
static_assert(sizeof(Node) == 64, "node must be 64 bytes");

// triangles per leaf: 2 measured best on MI355X (wf_trace 5.0 ms per C2 step; 1: 5.4, 4: 5.3, 8: 6.2) — a triangle
// test costs ~100 lane-instructions for every lane of the wave, a four-box node step ~170
#ifndef BF_MAX_LEAF
#define BF_MAX_LEAF 2
#endif
constexpr int kMaxLeaf = BF_MAX_LEAF;
constexpr int kMaxDepth = 31;   // tree depth bound == traversal stack bound (kStackDepth 32)

struct BVH {
    std::vector<Node> nodes;          // nodes[0] is the root (if any triangles)
    std::vector<uint32_t> order;      // order[i] = input triangle stored at slot i
    float lo[3], hi[3];               // padded scene bounds of the triangles
    int32_t root_child;               // encoding of the root as a child reference
    uint32_t max_depth;
};

// 128-byte four-wide node (one L2 line), read by the kernels as eight float4:
//   q0..q2 = lo.x[4], lo.y[4], lo.z[4];  q3..q5 = hi.x[4], hi.y[4], hi.z[4]
//   q6     = child[4] as int bits (same encoding as Node::child)
//   q7     = unused
// Unused child slots hold an inverted box (lo = +inf, hi = -inf), which no ray
// segment can overlap, and child = kEmptyChild.
struct alignas(16) Node4 {
    float lox[4], loy[4], loz[4], hix[4], hiy[4], hiz[4];
    int32_t child[4];
    int32_t pad[4];
};
static_assert(sizeof(Node4) == 128, "wide node must be 128 bytes");
constexpr int32_t kEmptyChild = INT32_MIN;

// The first kTopNodes nodes of a BVH4 are its top levels in breadth-first order (the traversal kernels keep a copy
// of them in LDS: every ray walks through them); the rest keep the depth-first order of the collapse.
constexpr uint32_t kTopNodes = 85;    // 1 + 4 + 16 + 64: three full levels under the root

struct BVH4 {
    std::vector<Node4> nodes;         // nodes[0] is the root if root_child >= 0
    int32_t root_child;               // >= 0: node index, < 0: the whole mesh is one leaf
    uint32_t stack_need;              // worst-case traversal stack entries: sum over levels of (children - 1) <= 3 * kMaxDepth
    uint32_t max_depth;
};

// Collapse the binary tree into four-wide nodes: a node adopts its grandchildren
// (largest surface area first) until it has four children or only leaves.  Half
// the dependent node fetches per ray.  The worst-case traversal stack (sum over
// levels of children - 1) is reported so that the kernels' stacks can be sized.
void collapse_bvh4(const BVH &in, BVH4 &out);

// 64-byte quantised form of a Node4 for the throughput traversal kernel (wf_trace): the node's own bounds as an fp32
// origin `lo` plus one power-of-two scale per axis, the four child boxes as 8-bit offsets from it (lower planes rounded
// down, upper planes rounded up, so every quantised box CONTAINS the fp32 box it replaces — box tests only select
// triangles, hits stay bit-exact).  Four 16-byte loads per node step instead of seven, 14 instead of 28 live
// registers.  Read by the kernels as four float4:
//   q0 = (lo.x, lo.y, lo.z, bits(ex | ey << 8 | ez << 16))      scale of axis a = 2^(e_a - 127) (a float's exponent field)
//   q1 = child[4] as int bits (same references as Node4: the two arrays are index-compatible)
//   q2 = (qlo.x[4], qlo.y[4], qlo.z[4], qhi.x[4])   one byte per child, child k in bits 8k .. 8k+7
//   q3 = (qhi.y[4], qhi.z[4], 0, 0)
// plane of child k: lo_a + q * 2^(e_a - 127).  Unused slots: qlo = 255, qhi = 0 (inverted) and child = kEmptyChild.
struct alignas(16) Node4Q {
    float lo[3];
    uint32_t exps;
    int32_t child[4];
    uint32_t qlo[3], qhi[3];
    uint32_t pad[2];
};
static_assert(sizeof(Node4Q) == 64, "quantised node must be 64 bytes");
void quantise_node4(const Node4 &in, Node4Q &out);      // also used (same arithmetic, on the device) by the mesh translation
void quantise_bvh4(const BVH4 &in, std::vector<Node4Q> &out);

// Sixteen-wide collapse of the SAME binary tree (same leaf order, same padded boxes) for the tail kernel's row
// traversal: one ray per 16-lane DPP row, lane j tests child j of a node or triangle j of a leaf, so the serial
// depth of a lone ray is ~log16 instead of ~log4 of the triangle count.  512 bytes per node:
//   child j = float4 (lo.x, lo.y, lo.z, hi.x), float4 (hi.y, hi.z, bits(ref), 0)
// ref >= 0: node index; ref < 0: leaf, ~ref = (first_triangle << 4) | (count - 1) with up to 16 triangles in
// contiguous slots (a whole binary subtree); kEmptyChild: unused slot (inverted box).
struct alignas(16) Node16 {
    float c[16][8];
};
static_assert(sizeof(Node16) == 512, "sixteen-wide node must be 512 bytes");
constexpr uint32_t kWideLeaf = 16;

struct BVH16 {
    std::vector<Node16> nodes;        // nodes[0] is the root if root_child >= 0
    int32_t root_child;               // >= 0: node index, < 0: the whole mesh is one leaf (Node16 leaf encoding)
    uint32_t stack_need;              // worst-case traversal stack entries (all hit children are pushed: sum of children per level)
    uint32_t max_depth;
};
void collapse_bvh16(const BVH &in, BVH16 &out);

// Binned SAH build (16 bins, leaf <= kMaxLeaf triangles).  Boxes are padded by a few
// ulps so that a fp32 Moeller-Trumbore hit distance never falls outside the
// box that holds its triangle.  `origin_scale`: largest |coordinate| a ray origin
// can have (scene geometry, sensors, emitters); the kernels' slab test folds the
// origin into one fma per plane (t = lo * (1/d) - o * (1/d)), whose rounding
// shifts a plane by up to 2^-24 * |o|, so boxes are also padded by
// 2e-7 * origin_scale to keep the test conservative.
void build_bvh(const std::vector<BuildTri> &tris, BVH &out, float origin_scale = 0.f);

}  // namespace bf
