// src/integrators/pathtimefrequency.cpp — PathTimeFrequencyIntegrator
#include "../render.h"
using namespace bfh;
class PathTimeFrequencyIntegrator final : public SamplingIntegrator {
public:
    explicit PathTimeFrequencyIntegrator(const Properties &props) : SamplingIntegrator(props) {}
    void configure(bf_launch &lp) const override { lp.mode = BF_MODE_RECEIVE_RAW; }
};
BF_EXPORT_PLUGIN(PathTimeFrequencyIntegrator, "SamplingIntegrator", "pathtimefrequency", "Path time-frequency integrator (fork, gen-3)")
