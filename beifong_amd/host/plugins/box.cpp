// src/rfilters/box.cpp — box filter, radius 0.5
#include "../render.h"
using namespace bfh;
class BoxFilter final : public ReconstructionFilter {
public:
    explicit BoxFilter(const Properties &props) { m_radius = props.float_("radius", .5f); }
    float radius() const override { return m_radius; }
private:
    float m_radius;
};
BF_EXPORT_PLUGIN(BoxFilter, "ReconstructionFilter", "box", "Box filter")
