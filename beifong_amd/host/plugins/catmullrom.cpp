// src/rfilters/catmullrom.cpp:21-56 — the Mitchell-Netravali cubic with B = 0, C = 1/2
#include "../render.h"
using namespace bfh;
class CatmullRomFilter final : public ReconstructionFilter {
public:
    explicit CatmullRomFilter(const Properties &) {
        m_radius = 2.f;
        init_discretization();
    }
    float eval(float x) const override { return cubic(x, 0.f, .5f); }
private:
    static float cubic(float x, float B, float C) {
        x = std::fabs(x);
        const float x2 = x * x, x3 = x2 * x;
        const float result = (1.f / 6.f) * (x < 1 ? (12.f - 9.f * B - 6.f * C) * x3 + (-18.f + 12.f * B + 6.f * C) * x2 + (6.f - 2.f * B)
                                                  : (-B - 6.f * C) * x3 + (6.f * B + 30.f * C) * x2 + (-12.f * B - 48.f * C) * x + (8.f * B + 24.f * C));
        return x < 2.f ? result : 0.f;
    }
};
BF_EXPORT_PLUGIN(CatmullRomFilter, "ReconstructionFilter", "catmullrom", "Catmull-Rom filter")
