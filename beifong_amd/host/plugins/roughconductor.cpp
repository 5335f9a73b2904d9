// src/bsdfs/roughconductor.cpp:145-193 — RoughConductor
#include <algorithm>
#include "../render.h"
using namespace bfh;
class RoughConductor final : public BSDF {
public:
    explicit RoughConductor(const Properties &props) {
        std::string material = props.string("material", "none");
        if (props.has_property("eta") || material == "none") {
            m_eta = props.texture_value("eta", 0.f);
            m_k = props.texture_value("k", 1.f);
            if (material != "none") Throw("Should specify either (eta, k) or material, not both.");
        } else {
            Throw("roughconductor: named materials need the IOR data files (resources/data is an empty submodule)");
        }
        m_type = BF_MF_BECKMANN;
        if (props.has_property("distribution")) {
            std::string d = props.string("distribution");
            std::transform(d.begin(), d.end(), d.begin(), ::tolower);
            if (d == "beckmann") m_type = BF_MF_BECKMANN;
            else if (d == "ggx") m_type = BF_MF_GGX;
            else Throw("Specified an invalid distribution \"%s\", must be \"beckmann\" or \"ggx\"!", d.c_str());
        }
        m_sample_visible = props.bool_("sample_visible", true);
        if (props.has_property("alpha_u") || props.has_property("alpha_v")) {
            if (!props.has_property("alpha_u") || !props.has_property("alpha_v"))
                Throw("Microfacet model: both 'alpha_u' and 'alpha_v' must be specified.");
            if (props.has_property("alpha")) Throw("Microfacet model: please specify either 'alpha' or 'alpha_u'/'alpha_v'.");
            m_alpha_u = props.texture_value("alpha_u", 0.1f);
            m_alpha_v = props.texture_value("alpha_v", 0.1f);
        } else {
            m_alpha_u = m_alpha_v = props.texture_value("alpha", 0.1f);
        }
        m_has_spec = props.has_property("specular_reflectance");
        m_spec = props.texture_value("specular_reflectance", 1.f);
    }
    bf_material flatten() const override {
        bf_material m{};
        m.type = BF_BSDF_ROUGHCONDUCTOR;
        m.reflectance = m_spec;
        m.has_specular_reflectance = m_has_spec;
        m.alpha_u = m_alpha_u;
        m.alpha_v = m_alpha_v;
        m.distribution = m_type;
        m.sample_visible = m_sample_visible;
        m.eta = m_eta;
        m.k = m_k;
        return m;
    }
private:
    float m_eta, m_k, m_alpha_u, m_alpha_v, m_spec;
    uint32_t m_type;
    bool m_sample_visible, m_has_spec;
};
BF_EXPORT_PLUGIN(RoughConductor, "BSDF", "roughconductor", "Rough conductor")
