// Host-side BVH construction for the HIP traversal kernels.
//
// The reference accelerates Scene::ray_intersect with an SAH kd-tree
// (include/mitsuba/render/kdtree.h, built in Scene::accel_init_cpu,
// src/librender/scene_native.inl:3-10) or embree/OptiX.  Closest-hit results do
// not depend on the accelerator (ties are resolved by primitive index, see
// bf_kernels.hip), so the MI355X build uses a structure that suits wave64
// pointer chasing instead: a binned-SAH BVH2 flattened into 64-byte nodes that
// hold BOTH child boxes, so one 64-B fetch decides two subtrees.
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

constexpr int kMaxLeaf = 4;
constexpr int kMaxDepth = 31;   // tree depth bound == traversal stack bound (kStackDepth 32)

struct BVH {
    std::vector<Node> nodes;          // nodes[0] is the root (if any triangles)
    std::vector<uint32_t> order;      // order[i] = input triangle stored at slot i
    float lo[3], hi[3];               // padded scene bounds of the triangles
    int32_t root_child;               // encoding of the root as a child reference
    uint32_t max_depth;
};

// Binned SAH build (16 bins, leaf <= kMaxLeaf).  Boxes are padded by a few
// ulps so that a fp32 Moeller-Trumbore hit distance never falls outside the
// box that holds its triangle.
void build_bvh(const std::vector<BuildTri> &tris, BVH &out);

}  // namespace bf
