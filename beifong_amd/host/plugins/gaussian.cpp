// src/rfilters/gaussian.cpp — default film/ADC filter; the radar path only accepts box
#include "../render.h"
using namespace bfh;
class GaussianFilter final : public ReconstructionFilter {
public:
    explicit GaussianFilter(const Properties &props) { m_stddev = props.float_("stddev", .5f); }
    float radius() const override { return 4 * m_stddev; }
private:
    float m_stddev;
};
BF_EXPORT_PLUGIN(GaussianFilter, "ReconstructionFilter", "gaussian", "Gaussian reconstruction filter")
