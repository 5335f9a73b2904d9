// src/sensors/irradiancemeter.cpp:43-61 — IrradianceMeter (the sensor the fork's flux meter was derived from)
#include "../render.h"
using namespace bfh;
class IrradianceMeter final : public Sensor {
public:
    explicit IrradianceMeter(const Properties &props) : Sensor(props) {
        if (props.has_property("to_world"))
            Throw("Found a 'to_world' transformation -- this is not allowed. The irradiance meter inherits this "
                  "transformation from its parent shape.");
        if (m_film->width() != 1 || m_film->height() != 1) Throw("This sensor only supports films of size 1x1 Pixels!");
        if (m_film->reconstruction_filter()->radius() > 0.5f + 1500 * 5.9604644775390625e-8f)
            Log(Warn, "This sensor should only be used with a reconstruction filter of radius 0.5 or lower(e.g. default box)");
    }
    void flatten(bf_sensor &s, int32_t shape) const override {
        if (shape < 0) Throw("irradiancemeter must be the child of a shape");
        s.type = BF_SENSOR_IRRADIANCEMETER;
        s.shape = shape;
        s.film_width = m_film->width();
        s.film_height = m_film->height();
        s.shutter_open = m_shutter_open;
        s.shutter_open_time = m_shutter_open_time;
    }
};
BF_EXPORT_PLUGIN(IrradianceMeter, "Sensor", "irradiancemeter", "Irradiance meter")
