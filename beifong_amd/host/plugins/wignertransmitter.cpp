// src/transmitters/wignertransmitter.cpp:53-110 — WignerTransmitter (fork)
#include "../render.h"
using namespace bfh;
class WignerTransmitter final : public Transmitter {
public:
    explicit WignerTransmitter(const Properties &props) : Transmitter(props) {
        if (props.has_property("to_world"))
            Throw("Found a 'to_world' transformation -- this is not allowed. The area light inherits this "
                  "transformation from its parent shape.");
        (void) props.texture_value("antenna_texture", 1.f);
        (void) props.texture_value("radiance", 1.f);       // Render.py passes it; the plugin never reads it
        m_signal = props.string("signaltype", "cw");
        m_resample_freq = props.bool_("resample_freq", false);
        m_repfreq = 1.f;
        m_t_ext = 1.f;
        if (m_signal == "linfmcw") {
            m_amplitude = props.float_("amplitude", 1.f);
            m_repfreq = props.float_("crf", 1.f);
            m_t_ext = props.float_("chirp_len", 1.f);
            m_f_centre = props.float_("freq_centre", 1.f);
            m_f_ext = props.float_("freq_sweep", 1.f);
            (void) props.bool_("sig_is_delta", true);
        } else if (m_signal == "pulse") {
            m_amplitude = props.float_("amplitude", 1.f);
            m_repfreq = props.float_("prf", 1.f);
            m_t_ext = props.float_("pulse_len", 1.f);
            m_f_centre = props.float_("freq_centre", 1.f);
            m_f_ext = props.float_("freq_ext", 1.f);
            (void) props.bool_("sig_is_delta", false);
        } else if (m_signal == "cw") {
            m_amplitude = props.float_("amplitude", 1.f);
            m_f_centre = props.float_("freq_centre", 1.f);
            m_f_ext = props.float_("freq_ext", 0.f);
            (void) props.bool_("sig_is_delta", true);
        } else {
            // :96-105 — any other signaltype evaluates like cw (eval_signal's final else)
            m_amplitude = props.float_("amplitude", 1.f);
            m_repfreq = props.float_("prf", 1.f);
            m_t_ext = props.float_("pulse_len", 1.f);
            m_f_centre = props.float_("freq_centre", 1.f);
            m_f_ext = 1.f / m_t_ext;
            (void) props.bool_("sig_is_delta", true);
        }
        (void) props.float_("phase", 0.f);                   // eval_signal forces the phase output to 0 (:143)
        m_gain = props.float_("gain", 1.f);
        // sample_delta_frequency (wignertransmitter.cpp:152-168) defines the re-sampled frequency for "linfmcw" and "cw" only
        if (m_resample_freq && m_signal == "pulse")
            Throw("wignertransmitter: resample_freq=true with signaltype \"pulse\" reads an uninitialised frequency in the reference");
    }
    bf_emitter flatten(int32_t shape) const override {
        if (shape < 0) Throw("wigner transmitter without an associated Shape");
        bf_emitter e{};
        e.type = BF_TRANSMITTER_WIGNER;
        e.shape = shape;
        e.radiance = 1.f;
        e.signal_type = m_signal == "linfmcw" ? BF_SIGNAL_LINFMCW : (m_signal == "pulse" ? BF_SIGNAL_PULSE : BF_SIGNAL_CW);
        e.amplitude = m_amplitude;
        e.freq_centre = m_f_centre;
        e.freq_ext = m_f_ext;
        e.pulse_len = m_t_ext;
        e.prf = m_repfreq;
        e.gain = m_gain;
        e.resample_freq = m_resample_freq ? 1u : 0u;
        return e;
    }
private:
    std::string m_signal;
    bool m_resample_freq;
    float m_amplitude, m_repfreq, m_t_ext, m_f_centre, m_f_ext, m_gain;
};
BF_EXPORT_PLUGIN(WignerTransmitter, "Transmitter", "wignertransmitter", "Wigner transmitter")
