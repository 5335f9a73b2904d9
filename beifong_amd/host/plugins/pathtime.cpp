// src/integrators/pathtime.cpp — PathTimeIntegrator
#include "../render.h"
using namespace bfh;
class PathTimeIntegrator final : public SamplingIntegrator {
public:
    explicit PathTimeIntegrator(const Properties &props) : SamplingIntegrator(props) {}
    void configure(bf_launch &lp) const override { lp.mode = BF_MODE_PATH; }
};
BF_EXPORT_PLUGIN(PathTimeIntegrator, "SamplingIntegrator", "pathtime", "Path time integrator (fork, gen-1)")
