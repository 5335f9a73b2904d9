// src/shapes/rectangle.cpp:66-105 — Rectangle
#include <cmath>
#include "../render.h"
using namespace bfh;
class Rectangle final : public Shape {
public:
    explicit Rectangle(const Properties &props) : Shape(props) {
        if (props.bool_("flip_normals", false)) m_to_world = m_to_world * Transform4f::scale({1.f, 1.f, -1.f});
    }
    uint32_t primitive_count() const override { return 1; }
    bool is_rectangle() const override { return true; }
    float surface_area() const override {
        const float *m = m_to_world.matrix.m;
        float s[3] = {2 * m[0], 2 * m[4], 2 * m[8]}, t[3] = {2 * m[1], 2 * m[5], 2 * m[9]};
        float c[3] = {s[1] * t[2] - s[2] * t[1], s[2] * t[0] - s[0] * t[2], s[0] * t[1] - s[1] * t[0]};
        return std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    }
};
BF_EXPORT_PLUGIN(Rectangle, "Shape", "rectangle", "Rectangle intersection primitive")
