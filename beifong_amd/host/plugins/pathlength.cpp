// src/integrators/pathlength.cpp — PathLengthIntegrator
#include "../render.h"
using namespace bfh;
class PathLengthIntegrator final : public SamplingIntegrator {
public:
    explicit PathLengthIntegrator(const Properties &props) : SamplingIntegrator(props) {}
    // on its own it renders like `path`; RangeIntegrator reads its range output
    void configure(bf_launch &lp) const override { lp.mode = BF_MODE_PATH; }
};
BF_EXPORT_PLUGIN(PathLengthIntegrator, "SamplingIntegrator", "pathlength", "Path length integrator (fork, gen-2)")
