// src/sensors/perspective.cpp:95-132 — PerspectiveCamera
#include <cmath>
#include "../render.h"
using namespace bfh;
class PerspectiveCamera final : public Sensor {
public:
    explicit PerspectiveCamera(const Properties &props) : Sensor(props) {
        m_near_clip = props.float_("near_clip", 1e-2f);          // ProjectiveCamera, sensor.cpp
        m_far_clip = props.float_("far_clip", 1e4f);
        (void) props.float_("focus_distance", m_far_clip);
        if (m_near_clip <= 0.f) Throw("The 'near_clip' parameter must be greater than zero!");
        if (m_near_clip >= m_far_clip) Throw("The 'near_clip' parameter must be smaller than 'far_clip'.");
        // parse_fov — include/mitsuba/render/sensor.h
        float aspect = m_film->full_width() / (float) m_film->full_height();      // m_film->size(): the full film
        if (props.has_property("fov") && props.has_property("focal_length"))
            Throw("Please specify either a focal length ('focal_length') or a field of view ('fov')!");
        float fov = props.float_("fov", 0.f);
        if (!props.has_property("fov")) {
            std::string f = props.string("focal_length", "50mm");
            if (f.size() > 2 && f.substr(f.size() - 2) == "mm") f = f.substr(0, f.size() - 2);
            float value = std::stof(f);
            fov = 2.f * std::atan(std::sqrt((float) (36 * 36 + 24 * 24)) / (2.f * value)) * 180.f / 3.14159265358979323846f;
            m_fov_axis = "diagonal";
        }
        m_fov_axis = props.string("fov_axis", props.has_property("fov") ? "x" : "diagonal");
        if (m_fov_axis == "x" || (m_fov_axis == "smaller" && aspect <= 1) || (m_fov_axis == "larger" && aspect > 1)) {
            m_x_fov = fov;
        } else if (m_fov_axis == "y" || m_fov_axis == "smaller" || m_fov_axis == "larger") {
            m_x_fov = 2.f * std::atan(std::tan(fov * .5f * 3.14159265358979323846f / 180.f) * aspect) * 180.f / 3.14159265358979323846f;
        } else if (m_fov_axis == "diagonal") {
            float diag = 2.f * std::tan(.5f * fov * 3.14159265358979323846f / 180.f);
            float width = diag / std::sqrt(1.f + 1.f / (aspect * aspect));
            m_x_fov = 2.f * std::atan(width * .5f) * 180.f / 3.14159265358979323846f;
        } else {
            Throw("The 'fov_axis' parameter must be set to one of 'smaller', 'larger', 'diagonal', 'x', or 'y'!");
        }
        if (m_to_world.has_scale()) Throw("Scale factors in the camera-to-world transformation are not allowed!");
    }
    void flatten(bf_sensor &s, int32_t) const override {
        s.type = BF_SENSOR_PERSPECTIVE;
        s.shape = -1;
        // perspective_projection — sensor.h:196-231 (film size, crop size, crop offset) and its inverse
        const float fw = (float) m_film->full_width(), fh = (float) m_film->full_height(), aspect = fw / fh;
        const float rel_w = (float) m_film->width() / fw, rel_h = (float) m_film->height() / fh;
        const float rel_x = (float) m_film->crop_offset_x() / fw, rel_y = (float) m_film->crop_offset_y() / fh;
        Transform4f c2s = Transform4f::scale({1.f / rel_w, 1.f / rel_h, 1.f}) * Transform4f::translate({-rel_x, -rel_y, 0.f}) *
                          Transform4f::scale({-0.5f, -0.5f * aspect, 1.f}) * Transform4f::translate({-1.f, -1.f / aspect, 0.f}) *
                          Transform4f::perspective(m_x_fov, m_near_clip, m_far_clip);
        for (int i = 0; i < 16; ++i) {
            s.to_world[i] = m_to_world.matrix.m[i];
            s.sample_to_camera[i] = c2s.inverse.m[i];
        }
        s.fov_x_deg = m_x_fov;
        s.near_clip = m_near_clip;
        s.far_clip = m_far_clip;
        s.film_width = m_film->width();
        s.film_height = m_film->height();
        s.shutter_open = m_shutter_open;
        s.shutter_open_time = m_shutter_open_time;
    }
private:
    float m_near_clip, m_far_clip, m_x_fov;
    std::string m_fov_axis;
};
BF_EXPORT_PLUGIN(PerspectiveCamera, "Sensor", "perspective", "Perspective Camera")
