// src/samplers/independent.cpp:54-91 — IndependentSampler (PCG32; draws happen on the device)
#include "../render.h"
using namespace bfh;
class IndependentSampler final : public Sampler {
public:
    explicit IndependentSampler(const Properties &props) : Sampler(props) {}
};
BF_EXPORT_PLUGIN(IndependentSampler, "Sampler", "independent", "Independent Sampler")
