// src/transmitters/areatransmitter.cpp:52-63 — AreaTransmitter (fork)
#include "../render.h"
using namespace bfh;
class AreaTransmitter final : public Transmitter {
public:
    explicit AreaTransmitter(const Properties &props) : Transmitter(props) {
        if (props.has_property("to_world"))
            Throw("Found a 'to_world' transformation -- this is not allowed. The area light inherits this "
                  "transformation from its parent shape.");
        m_radiance = props.texture_value("radiance", 1.f);
    }
    bf_emitter flatten(int32_t shape) const override {
        if (shape < 0) Throw("area transmitter without an associated Shape");
        bf_emitter e{};
        e.type = BF_TRANSMITTER_AREA;
        e.shape = shape;
        e.radiance = m_radiance;
        return e;
    }
private:
    float m_radiance;
};
BF_EXPORT_PLUGIN(AreaTransmitter, "Transmitter", "areatransmitter", "Area transmitter")
