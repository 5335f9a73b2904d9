// src/rfilters/mitchell.cpp:26-68 — Mitchell-Netravali cubic, radius 2, B = C = 1/3 by default
#include "../render.h"
using namespace bfh;
class MitchellNetravaliFilter final : public ReconstructionFilter {
public:
    explicit MitchellNetravaliFilter(const Properties &props) {
        m_radius = 2.f;
        m_b = props.float_("B", 1.f / 3.f);
        m_c = props.float_("C", 1.f / 3.f);
        init_discretization();
    }
    float eval(float x) const override { return cubic(x, m_b, m_c); }
private:
    static float cubic(float x, float B, float C) {
        x = std::fabs(x);
        const float x2 = x * x, x3 = x2 * x;
        const float result = (1.f / 6.f) * (x < 1 ? (12.f - 9.f * B - 6.f * C) * x3 + (-18.f + 12.f * B + 6.f * C) * x2 + (6.f - 2.f * B)
                                                  : (-B - 6.f * C) * x3 + (6.f * B + 30.f * C) * x2 + (-12.f * B - 48.f * C) * x + (8.f * B + 24.f * C));
        return x < 2.f ? result : 0.f;
    }
    float m_b, m_c;
};
BF_EXPORT_PLUGIN(MitchellNetravaliFilter, "ReconstructionFilter", "mitchell", "Mitchell-Netravali filter")
