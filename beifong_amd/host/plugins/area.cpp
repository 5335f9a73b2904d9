// src/emitters/area.cpp:40-62 — AreaLight
#include "../render.h"
using namespace bfh;
class AreaLight final : public Emitter {
public:
    explicit AreaLight(const Properties &props) : Emitter(props) {
        if (props.has_property("to_world"))
            Throw("Found a 'to_world' transformation -- this is not allowed. The area light inherits this "
                  "transformation from its parent shape.");
        m_radiance = props.texture_value("radiance", 1.f);
    }
    bf_emitter flatten(int32_t shape) const override {
        if (shape < 0) Throw("area emitter without an associated Shape");
        bf_emitter e{};
        e.type = BF_EMITTER_AREA;
        e.shape = shape;
        e.radiance = m_radiance;
        return e;
    }
private:
    float m_radiance;
};
BF_EXPORT_PLUGIN(AreaLight, "Emitter", "area", "Area emitter")
