// src/receivers/omnidirectional.cpp:43-70 — Omnidirectional receiver (fork)
#include "../render.h"
using namespace bfh;
class Omnidirectional final : public Receiver {
public:
    explicit Omnidirectional(const Properties &props) : Receiver(props) {
        if (props.has_property("to_world"))
            Throw("Found a 'to_world' transformation -- this is not allowed. The omnidirectional receiver inherits "
                  "this transformation from its parent shape.");
        if (m_adc->reconstruction_filter()->radius() > 0.5f + 1500 * 5.9604644775390625e-8f)
            Log(Warn, "This sensor should only be used with a reconstruction filter of radius 0.5 or lower(e.g. default box)");
    }
    void flatten(bf_sensor &s, int32_t shape) const override {
        if (shape < 0) Throw("receiver must be the child of a shape");
        s.type = BF_RECEIVER_OMNI;
        s.shape = shape;
        s.film_width = s.film_height = 1;
        s.adc_sampling_start = m_adc_sampling_start;
        s.adc_sampling_time = m_adc_sampling_time;
        s.t_bins = m_adc->t_bins();
        s.f_bins = m_adc->f_bins();
        s.t_bandwidth = m_adc->t_bandwidth();
        s.f_bandwidth = m_adc->f_bandwidth();
    }
};
BF_EXPORT_PLUGIN(Omnidirectional, "Receiver", "omnidirectional", "Omnidirectional receiver")
