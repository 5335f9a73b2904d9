// src/receivers/wignerreceiver.cpp:43-110 — Wignerreceiver (fork), receive_types "raw" / "raw_resample" and — delta signals — "mix_resample"
#include "../render.h"
using namespace bfh;
class Wignerreceiver final : public Receiver {
public:
    explicit Wignerreceiver(const Properties &props) : Receiver(props) {
        if (props.has_property("to_world"))
            Throw("Found a 'to_world' transformation -- this is not allowed. The wigner receiver inherits this "
                  "transformation from its parent shape.");
        if (m_adc->reconstruction_filter()->radius() > 0.5f + 1500 * 5.9604644775390625e-8f)
            Log(Warn, "This sensor should only be used with a reconstruction filter of radius 0.5 or lower(e.g. default box)");
        m_signal = BF_SIGNAL_CW;
        m_t_ext = m_repfreq = 0.f;
        m_amplitude = 1.f;
        if (m_receive_type == "mix_resample") {
            // the receiver's local oscillator (wignerreceiver.cpp:72-110): its frequency sample is the signal's instantaneous frequency at
            // the receive time for a delta signal (sample_frequency -> sample_delta_frequency, :149-189), a uniform frequency weighted
            // with eval_signal (:118-142) otherwise; a "pulse" that is a delta reads an uninitialised frequency there: refused
            const std::string sig = props.string("signaltype", "cw");
            m_amplitude = props.float_("amplitude", 1.f);
            (void) props.float_("phase", 0.f);
            if (sig == "linfmcw") {
                m_signal = BF_SIGNAL_LINFMCW;
                m_repfreq = props.float_("crf", 1.f);
                m_t_ext = props.float_("chirp_len", 1.f);
                m_f_centre = props.float_("freq_centre", 1.f);
                m_f_ext = props.float_("freq_sweep", 1.f);
                m_sig_is_delta = props.bool_("sig_is_delta", true);
            } else if (sig == "pulse") {
                m_signal = BF_SIGNAL_PULSE;
                m_repfreq = props.float_("prf", 1.f);
                m_t_ext = props.float_("pulse_len", 1.f);
                m_f_centre = props.float_("freq_centre", 1.f);
                m_f_ext = props.float_("freq_ext", 1.f);
                m_sig_is_delta = props.bool_("sig_is_delta", false);
                if (m_sig_is_delta)
                    Throw("wignerreceiver: receive_type \"mix_resample\" with a \"pulse\" that is a delta signal reads an uninitialised frequency "
                          "in the reference");
            } else if (sig == "cw") {
                m_f_centre = props.float_("freq_centre", 1.f);
                m_f_ext = props.float_("freq_ext", 0.f);
                m_sig_is_delta = props.bool_("sig_is_delta", true);
            } else {
                // :100-109 — any other signaltype is a delta signal whose frequency sample_delta_frequency never sets
                Throw("wignerreceiver: receive_type \"mix_resample\" with signaltype \"%s\" is not supported (\"linfmcw\", \"cw\" and \"pulse\" are)", sig.c_str());
            }
            m_gain = props.float_("gain", 1.f);
        } else {
            m_f_centre = props.float_("freq_centre", 1.f);
            m_f_ext = props.float_("freq_ext", 1.f);
            m_gain = props.float_("gain", 1.f);
            // :258 reads m_sig_is_delta, which the raw branch never initialises; explicit here
            m_sig_is_delta = props.bool_("sig_is_delta", false);
        }
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
        s.rx_signal_type = m_signal;
        s.rx_pulse_len = m_t_ext;
        s.rx_prf = m_repfreq;
        s.rx_amplitude = m_amplitude;
    }
private:
    float m_f_centre, m_f_ext, m_gain, m_t_ext, m_repfreq, m_amplitude;
    uint32_t m_signal;
    bool m_sig_is_delta;
};
BF_EXPORT_PLUGIN(Wignerreceiver, "Receiver", "wignerreceiver", "Wignerreceiver")
