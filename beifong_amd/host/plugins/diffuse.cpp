// src/bsdfs/diffuse.cpp:67-76 — SmoothDiffuse
#include "../render.h"
using namespace bfh;
class SmoothDiffuse final : public BSDF {
public:
    explicit SmoothDiffuse(const Properties &props) { m_reflectance = props.texture_value("reflectance", .5f); }
    bf_material flatten() const override {
        bf_material m{};
        m.type = BF_BSDF_DIFFUSE;
        m.reflectance = m_reflectance;
        m.alpha_u = m.alpha_v = 0.1f;
        m.sample_visible = 1;
        m.k = 1.f;
        return m;
    }
private:
    float m_reflectance;
};
BF_EXPORT_PLUGIN(SmoothDiffuse, "BSDF", "diffuse", "Smooth diffuse material")
