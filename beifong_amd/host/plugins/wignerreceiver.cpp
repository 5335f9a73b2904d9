// src/receivers/wignerreceiver.cpp:43-110 — Wignerreceiver (fork), receive_type "raw"
#include "../render.h"
using namespace bfh;
class Wignerreceiver final : public Receiver {
public:
    explicit Wignerreceiver(const Properties &props) : Receiver(props) {
        if (props.has_property("to_world"))
            Throw("Found a 'to_world' transformation -- this is not allowed. The wigner receiver inherits this "
                  "transformation from its parent shape.");
        // "mix_resample" gives this receiver a local-oscillator signal model of its own (wignerreceiver.cpp sample_frequency /
        // eval_signal, "signaltype" cw | pulse | linfmcw): not built; the omnidirectional receiver takes mix_resample
        if (m_receive_type == "mix_resample")
            Throw("wignerreceiver: receive_type \"mix_resample\" is not supported (\"raw\" and \"raw_resample\" are)");
        if (m_adc->reconstruction_filter()->radius() > 0.5f + 1500 * 5.9604644775390625e-8f)
            Log(Warn, "This sensor should only be used with a reconstruction filter of radius 0.5 or lower(e.g. default box)");
        m_f_centre = props.float_("freq_centre", 1.f);
        m_f_ext = props.float_("freq_ext", 1.f);
        m_gain = props.float_("gain", 1.f);
        // :258 reads m_sig_is_delta, which the raw branch never initialises; explicit here
        m_sig_is_delta = props.bool_("sig_is_delta", false);
    }
    void flatten(bf_sensor &s, int32_t shape) const override {
        if (shape < 0) Throw("receiver must be the child of a shape");
        s.type = BF_RECEIVER_WIGNER;
        s.shape = shape;
        s.film_width = s.film_height = 1;
        s.adc_sampling_start = m_adc_sampling_start;
        s.adc_sampling_time = m_adc_sampling_time;
        s.t_bins = m_adc->t_bins();
        s.f_bins = m_adc->f_bins();
        s.t_bandwidth = m_adc->t_bandwidth();
        s.f_bandwidth = m_adc->f_bandwidth();
        s.freq_centre = m_f_centre;
        s.freq_ext = m_f_ext;
        s.gain = m_gain;
        s.rx_sig_is_delta = m_sig_is_delta;
    }
private:
    float m_f_centre, m_f_ext, m_gain;
    bool m_sig_is_delta;
};
BF_EXPORT_PLUGIN(Wignerreceiver, "Receiver", "wignerreceiver", "Wignerreceiver")
