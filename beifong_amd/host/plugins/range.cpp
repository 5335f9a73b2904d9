// src/integrators/range.cpp:58-233 — RangeIntegrator (fork, gen-2): bins the nested
// PathLengthIntegrator's accumulated length into `bins` AOV channels of width `dr`
#include "../render.h"
using namespace bfh;
class RangeIntegrator final : public SamplingIntegrator {
public:
    explicit RangeIntegrator(const Properties &props) : SamplingIntegrator(props) {
        for (auto &kv : props.objects()) {
            auto *in = dynamic_cast<SamplingIntegrator *>(kv.second.get());
            if (!in) Throw("Child objects must be of type 'SamplingIntegrator'!");
            if (m_integrator) Throw("More than one sub-integrator specified!");
            m_integrator = in;
        }
        if (!m_integrator) Throw("Must specify a sub-integrator!");
        m_dr = props.float_("dr", -1.f);
        m_bins = (int) props.int_("bins", -1);
        if (m_bins <= 0 || !(m_dr > 0.f)) Throw("range: 'dr' and 'bins' must be positive");
    }
    std::vector<std::string> aov_names() const override {
        std::vector<std::string> r = m_integrator->aov_names();
        for (int i = 0; i < m_bins; ++i) r.insert(r.begin() + i, "S" + std::to_string(i) + ".Y");   // :216-221
        return r;
    }
    void configure(bf_launch &lp) const override {
        lp.mode = BF_MODE_RANGE;
        lp.bins = (uint32_t) m_bins;
        lp.bin_width = m_dr;
    }
    int max_depth() const override { return m_integrator->max_depth(); }
    bool doppler() const override { return m_integrator->doppler(); }
    int rr_depth() const override { return m_integrator->rr_depth(); }
private:
    ref<SamplingIntegrator> m_integrator;
    float m_dr;
    int m_bins;
};
BF_EXPORT_PLUGIN(RangeIntegrator, "SamplingIntegrator", "range", "Range integrator")
