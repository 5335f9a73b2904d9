// src/bsdfs/twosided.cpp:62-92 — TwoSidedBRDF (one nested BRDF used for both sides)
#include "../render.h"
using namespace bfh;
class TwoSidedBRDF final : public BSDF {
public:
    explicit TwoSidedBRDF(const Properties &props) {
        auto bsdfs = props.objects();
        if (bsdfs.size() > 0) m_brdf[0] = dynamic_cast<BSDF *>(bsdfs[0].second.get());
        if (bsdfs.size() == 2) m_brdf[1] = dynamic_cast<BSDF *>(bsdfs[1].second.get());
        else if (bsdfs.size() > 2) Throw("At most two nested BSDFs can be specified!");
        if (!m_brdf[0]) Throw("A nested one-sided material is required!");
        if (m_brdf[1] && m_brdf[1].get() != m_brdf[0].get())
            Throw("twosided: two different nested BSDFs are not supported on the radar path");
    }
    bf_material flatten() const override {
        bf_material m = m_brdf[0]->flatten();
        m.twosided = 1;
        return m;
    }
private:
    ref<BSDF> m_brdf[2];
};
BF_EXPORT_PLUGIN(TwoSidedBRDF, "BSDF", "twosided", "Two-sided material adapter")
