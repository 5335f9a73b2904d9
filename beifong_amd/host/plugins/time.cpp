// src/integrators/time.cpp:58-215 — TimeIntegrator (fork, gen-1): 50 bins of 0.5 ns,
// three channels S{i}.{R,G,B} per bin
#include "../render.h"
using namespace bfh;
class TimeIntegrator final : public SamplingIntegrator {
public:
    explicit TimeIntegrator(const Properties &props) : SamplingIntegrator(props) {
        for (auto &kv : props.objects()) {
            auto *in = dynamic_cast<SamplingIntegrator *>(kv.second.get());
            if (!in) Throw("Child objects must be of type 'SamplingIntegrator'!");
            if (m_integrator) Throw("More than one sub-integrator specified!");
            m_integrator = in;
        }
        if (!m_integrator) Throw("Must specify a sub-integrator!");
    }
    std::vector<std::string> aov_names() const override {
        std::vector<std::string> r = m_integrator->aov_names();
        for (int i = 0; i < 50; ++i)
            for (int j = 0; j < 3; ++j) r.insert(r.begin() + 3 * i + j, "S" + std::to_string(i) + "." + "RGB"[j]);   // :200-206
        return r;
    }
    void configure(bf_launch &lp) const override {
        lp.mode = BF_MODE_TIME;
        lp.bins = 50;                 // time.cpp:134
        lp.bin_width = 0.5e-9f;       // time.cpp:118
        lp.time_c = 3.0e8f;           // pathtime.cpp:140
    }
    int max_depth() const override { return m_integrator->max_depth(); }
    bool doppler() const override { return m_integrator->doppler(); }
    int rr_depth() const override { return m_integrator->rr_depth(); }
private:
    ref<SamplingIntegrator> m_integrator;
};
BF_EXPORT_PLUGIN(TimeIntegrator, "SamplingIntegrator", "time", "Time integrator")
