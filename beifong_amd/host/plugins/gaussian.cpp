// src/rfilters/gaussian.cpp:28-60 — windowed Gaussian, the default film / ADC filter
#include "../render.h"
using namespace bfh;
class GaussianFilter final : public ReconstructionFilter {
public:
    explicit GaussianFilter(const Properties &props) {
        m_stddev = props.float_("stddev", .5f);
        m_radius = 4 * m_stddev;                                  // cut off after 4 standard deviations
        m_alpha = -1.f / (2.f * m_stddev * m_stddev);
        m_bias = std::exp(m_alpha * (m_radius * m_radius));
        init_discretization();
    }
    float eval(float x) const override { return std::max(0.f, std::exp(m_alpha * (x * x)) - m_bias); }
private:
    float m_stddev, m_alpha, m_bias;
};
BF_EXPORT_PLUGIN(GaussianFilter, "ReconstructionFilter", "gaussian", "Gaussian reconstruction filter")
