// src/integrators/path.cpp — PathIntegrator
#include "../render.h"
using namespace bfh;
class PathIntegrator final : public SamplingIntegrator {
public:
    explicit PathIntegrator(const Properties &props) : SamplingIntegrator(props) {}
    void configure(bf_launch &lp) const override { lp.mode = BF_MODE_PATH; }
};
BF_EXPORT_PLUGIN(PathIntegrator, "SamplingIntegrator", "path", "Path Tracer integrator")
