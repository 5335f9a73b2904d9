// src/rfilters/box.cpp:27-46 — box filter, radius 0.5 (+ RayEpsilon, box.cpp:33)
#include "../render.h"
using namespace bfh;
class BoxFilter final : public ReconstructionFilter {
public:
    explicit BoxFilter(const Properties &props) {
        m_radius = props.float_("radius", .5f) + 1500 * 5.9604644775390625e-8f;
        init_discretization();
    }
    float eval(float x) const override { return std::fabs(x) <= m_radius ? 1.f : 0.f; }
};
BF_EXPORT_PLUGIN(BoxFilter, "ReconstructionFilter", "box", "Box filter")
