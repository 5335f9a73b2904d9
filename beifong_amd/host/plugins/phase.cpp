// src/integrators/phase.cpp:58-159 — PhaseIntegrator (fork, built at HEAD): bins the phase the nested
// integrator leaves on the ray (ray.h:89-93 via pathtimefrequency.cpp:151,369,453) into `bins` AOV
// channels S{k}.Y of width 2 pi / bins; each sample adds hsum(L) of the nested result to its bin
#include "../render.h"
using namespace bfh;
class PhaseIntegrator final : public SamplingIntegrator {
public:
    explicit PhaseIntegrator(const Properties &props) : SamplingIntegrator(props) {
        for (auto &kv : props.objects()) {
            auto *in = dynamic_cast<SamplingIntegrator *>(kv.second.get());
            if (!in) Throw("Child objects must be of type 'SamplingIntegrator'!");
            if (m_integrator) Throw("More than one sub-integrator specified!");
            m_integrator = in;
        }
        if (!m_integrator) Throw("Must specify a sub-integrator!");
        m_bins = (int) props.int_("bins", 1);                                   // :80
        if (m_bins <= 0) Throw("phase: 'bins' must be positive");
    }
    std::vector<std::string> aov_names() const override {
        std::vector<std::string> r = m_integrator->aov_names();
        for (int i = 0; i < m_bins; ++i) r.insert(r.begin() + i, "S" + std::to_string(i) + ".Y");   // :142-147
        return r;
    }
    void configure(bf_launch &lp) const override {
        m_integrator->configure(lp);
        if (lp.mode != BF_MODE_RECEIVE_RAW) Throw("phase: the sub-integrator must be 'pathtimefrequency'");
        lp.phase_bins = (uint32_t) m_bins;
    }
    int max_depth() const override { return m_integrator->max_depth(); }
    bool doppler() const override { return m_integrator->doppler(); }
    int rr_depth() const override { return m_integrator->rr_depth(); }
private:
    ref<SamplingIntegrator> m_integrator;
    int m_bins;
};
BF_EXPORT_PLUGIN(PhaseIntegrator, "SamplingIntegrator", "phase", "Phase integrator")
