// src/bsdfs/twosided.cpp:62-92 — TwoSidedBRDF: one nested BRDF for both sides, or two (front, back)
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
        if (dynamic_cast<TwoSidedBRDF *>(m_brdf[0].get()) || dynamic_cast<TwoSidedBRDF *>(m_brdf[1].get()))
            Throw("Only materials with reflection components can be nested!");      // twosided.cpp: a nested twosided has both sides
    }
    bf_material flatten() const override {
        bf_material m = m_brdf[0]->flatten();
        m.twosided = 1;
        return m;
    }
    /// the second nested BSDF (the back side), or null: Scene::flatten gives it a table entry of its own (bf_material.back_material)
    const BSDF *back() const override { return (m_brdf[1] && m_brdf[1].get() != m_brdf[0].get()) ? m_brdf[1].get() : nullptr; }
private:
    ref<BSDF> m_brdf[2];
};
BF_EXPORT_PLUGIN(TwoSidedBRDF, "BSDF", "twosided", "Two-sided material adapter")
