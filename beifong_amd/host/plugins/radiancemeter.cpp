// src/sensors/radiancemeter.cpp:49-114 — RadianceMeter: a single ray from `origin` along `direction` (or to_world)
#include <cmath>

#include "../render.h"
using namespace bfh;
class RadianceMeter final : public Sensor {
public:
    explicit RadianceMeter(const Properties &props) : Sensor(props) {
        if (!props.has_property("to_world")) {
            if (props.has_property("direction") != props.has_property("origin"))
                Throw("If the sensor is specified through origin and direction both values must be set!");
            if (props.has_property("direction")) {
                const Vector3f origin = props.vector3f("origin", Vector3f()), n = props.vector3f("direction", Vector3f());
                const Vector3f target{origin.x + n.x, origin.y + n.y, origin.z + n.z};
                // up = first vector of coordinate_system(direction) — vector.h:116-136 (Duff et al.)
                auto mulsign = [](float v, float w) { return std::signbit(w) ? -v : v; };
                const float sign = std::copysign(1.f, n.z), a = -1.f / (sign + n.z), b = n.x * n.y * a;
                const Vector3f up{mulsign(n.x * n.x * a, n.z) + 1.f, mulsign(b, n.z), -mulsign(n.x, n.z)};
                m_to_world = Transform4f::look_at(origin, target, up);
            }
        }
        if (m_film->width() != 1 || m_film->height() != 1) Throw("This sensor only supports films of size 1x1 Pixels!");
        if (m_film->reconstruction_filter()->radius() > 0.5f + 1500 * 5.9604644775390625e-8f)
            Log(Warn, "This sensor should be used with a reconstruction filter with a radius of 0.5 or lower (e.g. default box)");
    }
    void flatten(bf_sensor &s, int32_t) const override {
        s.type = BF_SENSOR_RADIANCEMETER;
        s.shape = -1;
        for (int i = 0; i < 16; ++i) s.to_world[i] = m_to_world.matrix.m[i];
        s.film_width = m_film->width();
        s.film_height = m_film->height();
        s.shutter_open = m_shutter_open;
        s.shutter_open_time = m_shutter_open_time;
    }
};
BF_EXPORT_PLUGIN(RadianceMeter, "Sensor", "radiancemeter", "Radiance meter")
