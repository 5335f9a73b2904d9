// src/emitters/spot.cpp:64-95 — SpotLight
#include "../render.h"
using namespace bfh;
class SpotLight final : public Emitter {
public:
    explicit SpotLight(const Properties &props) : Emitter(props) {
        m_intensity = props.texture_value("intensity", 1.f);
        if (props.has_property("texture")) Throw("spot: projection textures are not supported on the radar path");
        m_cutoff_angle = props.float_("cutoff_angle", 20.0f);
        m_beam_width = props.float_("beam_width", m_cutoff_angle * 3.0f / 4.0f);
        if (m_cutoff_angle < m_beam_width) Throw("spot: cutoff_angle must be >= beam_width");
    }
    bf_emitter flatten(int32_t) const override {
        bf_emitter e{};
        e.type = BF_EMITTER_SPOT;
        e.shape = -1;
        for (int i = 0; i < 16; ++i) {
            e.to_world[i] = m_to_world.matrix.m[i];
            e.to_object[i] = m_to_world.inverse.m[i];
        }
        e.radiance = m_intensity;
        e.cutoff_angle_deg = m_cutoff_angle;
        e.beam_width_deg = m_beam_width;
        return e;
    }
private:
    float m_intensity, m_cutoff_angle, m_beam_width;
};
BF_EXPORT_PLUGIN(SpotLight, "Emitter", "spot", "Spot emitter")
