// src/emitters/point.cpp:36-58 — PointLight
#include "../render.h"
using namespace bfh;
class PointLight final : public Emitter {
public:
    explicit PointLight(const Properties &props) : Emitter(props) {
        if (props.has_property("position")) {
            if (props.has_property("to_world"))
                Throw("Only one of the parameters 'position' and 'to_world' can be specified at the same time!'");
            m_to_world = Transform4f::translate(props.vector3f("position", Vector3f()));
        }
        m_intensity = props.texture_value("intensity", 1.f);
    }
    bf_emitter flatten(int32_t) const override {
        bf_emitter e{};
        e.type = BF_EMITTER_POINT;
        e.shape = -1;
        for (int i = 0; i < 16; ++i) {
            e.to_world[i] = m_to_world.matrix.m[i];
            e.to_object[i] = m_to_world.inverse.m[i];
        }
        e.radiance = m_intensity;
        return e;
    }
private:
    float m_intensity;
};
BF_EXPORT_PLUGIN(PointLight, "Emitter", "point", "Point emitter")
