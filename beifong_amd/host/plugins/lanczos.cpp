// src/rfilters/lanczos.cpp:31-63 — windowed sinc, `lobes` side lobes (3)
#include "../render.h"
using namespace bfh;
class LanczosSincFilter final : public ReconstructionFilter {
public:
    explicit LanczosSincFilter(const Properties &props) {
        m_radius = (float) props.int_("lobes", 3);
        init_discretization();
    }
    float eval(float x) const override {
        x = std::fabs(x);
        const float x1 = 3.14159265358979323846f * x, x2 = x1 / m_radius, result = (std::sin(x1) * std::sin(x2)) / (x1 * x2);
        return x < 5.9604644775390625e-8f ? 1.f : (x > m_radius ? 0.f : result);
    }
};
BF_EXPORT_PLUGIN(LanczosSincFilter, "ReconstructionFilter", "lanczos", "Lanczos Sinc filter")
