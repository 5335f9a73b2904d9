// src/rfilters/tent.cpp:23-47 — tent filter, radius 1
#include "../render.h"
using namespace bfh;
class TentFilter final : public ReconstructionFilter {
public:
    explicit TentFilter(const Properties &) {
        m_radius = 1.f;
        m_inv_radius = 1.f / m_radius;
        init_discretization();
    }
    float eval(float x) const override { return std::max(0.f, 1.f - std::fabs(x * m_inv_radius)); }
private:
    float m_inv_radius;
};
BF_EXPORT_PLUGIN(TentFilter, "ReconstructionFilter", "tent", "Tent filter")
