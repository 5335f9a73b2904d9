// Hand-written gfx950 kernels for beifong's transient-radar hot path.
//
// One persistent wave64 megakernel with path regeneration: every lane owns one
// path at a time; a lane whose path ended pulls the next global path index
// from a device-wide queue head (wave-aggregated with __ballot/__popcll, one
// returning atomic per refill) and starts over, so all 64 lanes enter every
// BVH traversal.  Traversal keeps a per-lane stack in LDS (lane-strided, so a
// wave's push/pop is bank-conflict free), reads 64-B two-child nodes and 48-B
// triangles from HBM, and path-length returns are binned into an
// LDS-privatised histogram that is flushed with one global atomic per
// non-empty bin per workgroup.  No MFMA: the path is pointer chasing.
//
// Reference semantics (file:line) are cited at each function; the oracle
// (oracle/bf_oracle.cpp) restates the same functions independently on the CPU.
#include "bf_device.h"
#include "bf_device_math.h"

namespace bfd {

struct Hit {
    float t, u, v;
    uint32_t prim;      // global primitive index (tie rule)
    int32_t slot;       // triangle slot in leaf order, or -(rect+1)
};

// Mesh::ray_intersect_triangle — include/mitsuba/render/mesh.h:190-224
BF_DEV bool tri_intersect(V3 p0, V3 p1, V3 p2, V3 o, V3 d, float mint, float maxt, float &t, float &u, float &v) {
    V3 e1 = p1 - p0, e2 = p2 - p0;
    V3 pvec = cross(d, e2);
    float inv_det = rcp(dot(e1, pvec));
    V3 tvec = o - p0;
    u = dot(tvec, pvec) * inv_det;
    bool active = u >= 0.f && u <= 1.f;
    V3 qvec = cross(tvec, e1);
    v = dot(d, qvec) * inv_det;
    active = active && v >= 0.f && u + v <= 1.f;
    t = dot(e2, qvec) * inv_det;
    return active && t >= mint && t <= maxt;
}

// Rectangle::ray_intersect_preliminary — src/shapes/rectangle.cpp:229-249
BF_DEV bool rect_intersect(const DRect &rc, V3 o, V3 d, float mint, float maxt, float &t, float &lx, float &ly) {
    V3 oo = xf_point(rc.to_object, o);
    V3 dd = xf_vector(rc.to_object, d);
    float d_rcp_z = rcp(dd.z);
    t = -oo.z * d_rcp_z;
    V3 local = fmadd3(dd, t, oo);
    lx = local.x;
    ly = local.y;
    return t >= mint && t <= maxt && __builtin_fabsf(local.x) <= 1.f && __builtin_fabsf(local.y) <= 1.f;
}

// Closest-hit tie rule: the reference shrinks ray.maxt and accepts t <= maxt
// (kdtree.h:2139-2156; Scene::ray_intersect_naive), so among equal t the
// primitive tested later — the larger global index in the naive order — wins.
// Fixing that rule makes the result independent of traversal order.
BF_DEV void consider(Hit &best, float t, float u, float v, uint32_t prim, int32_t slot) {
    if (t < best.t || (t == best.t && prim > best.prim)) {
        best.t = t;
        best.u = u;
        best.v = v;
        best.prim = prim;
        best.slot = slot;
    }
}

BF_DEV bool slab(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 o, V3 id, float mint, float tmax,
                 float &tn) {
    float t0x = (lox - o.x) * id.x, t1x = (hix - o.x) * id.x;
    float t0y = (loy - o.y) * id.y, t1y = (hiy - o.y) * id.y;
    float t0z = (loz - o.z) * id.z, t1z = (hiz - o.z) * id.z;
    tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                         __builtin_fmaxf(__builtin_fminf(t0z, t1z), mint));
    float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)),
                               __builtin_fminf(__builtin_fmaxf(t0z, t1z), tmax));
    return tn <= tf * 1.0000004f;
}

// Scene::ray_intersect / ray_test — src/librender/scene.cpp:129-178.
// `stack` points at this lane's column of the workgroup's LDS stack
// (entry k at stack[k * kBlock]).
template <bool ANY, bool STATS>
BF_DEV bool traverse(const DScene &sc, V3 o, V3 d, float mint, float maxt, int *stack, Hit &best, uint32_t &n_nodes,
                     uint32_t &n_tris) {
    best.t = BF_INF;
    best.u = best.v = 0.f;
    best.prim = 0;
    best.slot = 0;
    // analytic rectangles (antennas, target plate, ground): a handful per scene
    for (uint32_t i = 0; i < sc.n_rects; ++i) {
        const DRect &rc = sc.rects[i];
        float t, lx, ly;
        if (rect_intersect(rc, o, d, mint, maxt, t, lx, ly)) {
            if (ANY) return true;
            consider(best, t, lx, ly, rc.prim, -(int32_t) (i + 1));
        }
    }
    if (sc.n_tris == 0) return best.t != BF_INF;

    V3 id = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
    int node = sc.root;
    int sp = 0;
    while (true) {
        if (node >= 0) {
            const float4 *np = sc.nodes + 4 * (size_t) node;
            float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
            if (STATS) ++n_nodes;
            float tmax = ANY ? maxt : __builtin_fminf(maxt, best.t);
            float tn0, tn1;
            bool h0 = slab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, o, id, mint, tmax, tn0);
            bool h1 = slab(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, o, id, mint, tmax, tn1);
            int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
            if (h0 && h1) {
                if (tn1 < tn0) {
                    int tmp = c0;
                    c0 = c1;
                    c1 = tmp;
                }
                stack[sp * kBlock] = c1;
                ++sp;
                node = c0;
                continue;
            } else if (h0) {
                node = c0;
                continue;
            } else if (h1) {
                node = c1;
                continue;
            }
        } else {
            uint32_t enc = ~(uint32_t) node;
            uint32_t first = enc >> 3, cnt = (enc & 7u) + 1u;
            for (uint32_t i = 0; i < cnt; ++i) {
                const float4 *tp = sc.tris + 3 * (size_t) (first + i);
                float4 a = tp[0], b = tp[1], c = tp[2];
                if (STATS) ++n_tris;
                float t, u, v;
                if (tri_intersect(mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), o, d, mint, maxt, t, u, v)) {
                    if (ANY) return true;
                    consider(best, t, u, v, __float_as_uint(a.w), (int32_t) (first + i));
                }
            }
        }
        if (sp == 0) break;
        --sp;
        node = stack[sp * kBlock];
    }
    return best.t != BF_INF;
}

// ---------------------------------------------------------------------------
// surface interaction: PreliminaryIntersection::compute_surface_interaction
// (interaction.h:613-644), Mesh::compute_surface_interaction
// (mesh.cpp:452-548), Rectangle::compute_surface_interaction
// (rectangle.cpp:265-298), initialize_sh_frame (interaction.h:159-162)
// ---------------------------------------------------------------------------
struct SI {
    float t;
    V3 p, wi;
    Frame sh;
    uint32_t shape;
};

BF_DEV void make_si(const DScene &sc, V3 o, V3 d, const Hit &h, SI &si) {
    si.t = h.t;
    V3 dp_du;
    if (h.slot < 0) {
        const DRect &rc = sc.rects[-h.slot - 1];
        si.shape = rc.shape;
        si.p = fmadd3(d, h.t, o);
        si.sh.n = mk(rc.n[0], rc.n[1], rc.n[2]);
        dp_du = mk(rc.s[0], rc.s[1], rc.s[2]);
    } else {
        const float4 *tp = sc.tris + 3 * (size_t) h.slot;
        float4 a = tp[0], b = tp[1], c = tp[2];
        V3 p0 = mk(a.x, a.y, a.z), p1 = mk(b.x, b.y, b.z), p2 = mk(c.x, c.y, c.z);
        si.shape = __float_as_uint(b.w);
        float b1 = h.u, b2 = h.v, b0 = 1.f - b1 - b2;
        V3 dp0 = p1 - p0, dp1 = p2 - p0;
        si.p = p0 * b0 + p1 * b1 + p2 * b2;
        V3 n = normalize(cross(dp0, dp1));
        V3 dp_dv;
        coordinate_system(n, dp_du, dp_dv);
        if (__float_as_uint(c.w) != 0u) {
            const float4 *nq = sc.normals + 3 * (size_t) h.slot;
            float4 na = nq[0], nb = nq[1], nc = nq[2];
            si.sh.n = normalize(mk(na.x, na.y, na.z) * b0 + mk(nb.x, nb.y, nb.z) * b1 + mk(nc.x, nc.y, nc.z) * b2);
        } else {
            si.sh.n = n;
        }
    }
    float dd = dot(si.sh.n, dp_du);
    si.sh.s = normalize(mk(fnmadd(si.sh.n.x, dd, dp_du.x), fnmadd(si.sh.n.y, dd, dp_du.y), fnmadd(si.sh.n.z, dd, dp_du.z)));
    si.sh.t = cross(si.sh.n, si.sh.s);
    si.wi = to_local(si.sh, -d);
}

// ---------------------------------------------------------------------------
// BSDFs: diffuse.cpp:78-135, roughconductor.cpp:196-392 (+ microfacet.h,
// fresnel.h:92-116), twosided.cpp:94-180
// ---------------------------------------------------------------------------
struct Microfacet {
    uint32_t type;
    float au, av;
    bool sample_visible;
};
BF_DEV Microfacet mf_make(const bf_material &m) {
    Microfacet d;
    d.type = m.distribution;
    d.au = __builtin_fmaxf(m.alpha_u, 1e-4f);
    d.av = __builtin_fmaxf(m.alpha_v, 1e-4f);
    d.sample_visible = m.sample_visible != 0;
    return d;
}
BF_DEV float mf_eval(const Microfacet &d, V3 m) {
    float alpha_uv = d.au * d.av, cos_theta = m.z, cos_theta_2 = sqr(cos_theta), result;
    if (d.type == BF_MF_BECKMANN)
        result = exp_cr(-(sqr(m.x / d.au) + sqr(m.y / d.av)) / cos_theta_2) / (kPi * alpha_uv * sqr(cos_theta_2));
    else
        result = rcp(kPi * alpha_uv * sqr(sqr(m.x / d.au) + sqr(m.y / d.av) + sqr(m.z)));
    return (result * cos_theta > 1e-20f) ? result : 0.f;
}
BF_DEV float mf_smith_g1(const Microfacet &d, V3 v, V3 m) {
    float xy_alpha_2 = sqr(d.au * v.x) + sqr(d.av * v.y), tan_theta_alpha_2 = xy_alpha_2 / sqr(v.z), result;
    if (d.type == BF_MF_BECKMANN) {
        float a = 1.f / __builtin_sqrtf(tan_theta_alpha_2), a_sqr = sqr(a);
        result = (a >= 1.6f) ? 1.f : (3.535f * a + 2.181f * a_sqr) / (1.f + 2.276f * a + 2.577f * a_sqr);
    } else {
        result = 2.f / (1.f + __builtin_sqrtf(1.f + tan_theta_alpha_2));
    }
    if (xy_alpha_2 == 0.f) result = 1.f;
    if (dot(v, m) * v.z <= 0.f) result = 0.f;
    return result;
}
BF_DEV float mf_G(const Microfacet &d, V3 wi, V3 wo, V3 m) { return mf_smith_g1(d, wi, m) * mf_smith_g1(d, wo, m); }
BF_DEV void mf_sample_visible_11(const Microfacet &d, float cos_theta_i, float sx, float sy, float &ox, float &oy) {
    if (d.type == BF_MF_BECKMANN) {
        float tan_theta_i = safe_sqrt(fnmadd(cos_theta_i, cos_theta_i, 1.f)) / cos_theta_i;
        float cot_theta_i = rcp(tan_theta_i);
        float maxval = erf_cr(cot_theta_i);
        sx = __builtin_fmaxf(__builtin_fminf(sx, 1.f - 1e-6f), 1e-6f);
        sy = __builtin_fmaxf(__builtin_fminf(sy, 1.f - 1e-6f), 1e-6f);
        float x = maxval - (maxval + 1.f) * erf_cr(__builtin_sqrtf(-log_cr(sx)));
        sx *= 1.f + maxval + kInvSqrtPi * tan_theta_i * exp_cr(-sqr(cot_theta_i));
#pragma nounroll
        for (int i = 0; i < 3; ++i) {
            float slope = erfinv_giles(x);
            float value = 1.f + x + kInvSqrtPi * tan_theta_i * exp_cr(-sqr(slope)) - sx;
            float derivative = 1.f - slope * tan_theta_i;
            x -= value / derivative;
        }
        ox = erfinv_giles(x);
        oy = erfinv_giles(fmsub(2.f, sy, 1.f));
    } else {
        float px, py;
        square_to_uniform_disk_concentric(sx, sy, px, py);
        float s = .5f * (1.f + cos_theta_i);
        float a = safe_sqrt(1.f - sqr(px));
        py = fmadd(py, s, fnmadd(a, s, a));
        float x = px, y = py, z = safe_sqrt(1.f - fmadd(py, py, px * px));
        float sin_theta_i = safe_sqrt(1.f - sqr(cos_theta_i));
        float nrm = rcp(fmadd(sin_theta_i, y, cos_theta_i * z));
        ox = fmsub(cos_theta_i, y, sin_theta_i * z) * nrm;
        oy = x * nrm;
    }
}
BF_DEV void mf_sample(const Microfacet &d, V3 wi, float sx, float sy, V3 &m, float &pdf) {
    if (!d.sample_visible) {
        float sin_phi, cos_phi, cos_theta, cos_theta_2, alpha_2;
        if (d.au == d.av) {
            float ang = (2.f * kPi) * sy;
            sin_phi = sin_cr(ang);
            cos_phi = cos_cr(ang);
            alpha_2 = d.au * d.au;
        } else {
            float ratio = d.av / d.au, tmp = ratio * tan_cr((2.f * kPi) * sy);
            cos_phi = 1.f / __builtin_sqrtf(fmadd(tmp, tmp, 1.f));
            cos_phi = mulsign(cos_phi, __builtin_fabsf(sy - .5f) - .25f);
            sin_phi = cos_phi * tmp;
            alpha_2 = rcp(sqr(cos_phi / d.au) + sqr(sin_phi / d.av));
        }
        if (d.type == BF_MF_BECKMANN) {
            cos_theta = 1.f / __builtin_sqrtf(fnmadd(alpha_2, log_cr(1.f - sx), 1.f));
            cos_theta_2 = sqr(cos_theta);
            float cos_theta_3 = __builtin_fmaxf(cos_theta_2 * cos_theta, 1e-20f);
            pdf = (1.f - sx) / (kPi * d.au * d.av * cos_theta_3);
        } else {
            float tan_theta_m_2 = alpha_2 * sx / (1.f - sx);
            cos_theta = 1.f / __builtin_sqrtf(1.f + tan_theta_m_2);
            cos_theta_2 = sqr(cos_theta);
            float temp = 1.f + tan_theta_m_2 / alpha_2, cos_theta_3 = __builtin_fmaxf(cos_theta_2 * cos_theta, 1e-20f);
            pdf = rcp(kPi * d.au * d.av * cos_theta_3 * sqr(temp));
        }
        float sin_theta = __builtin_sqrtf(1.f - cos_theta_2);
        m = mk(cos_phi * sin_theta, sin_phi * sin_theta, cos_theta);
    } else {
        V3 wi_p = normalize(mk(d.au * wi.x, d.av * wi.y, wi.z));
        float sin_theta_2 = fmadd(wi_p.x, wi_p.x, sqr(wi_p.y));
        float inv_sin_theta = 1.f / __builtin_sqrtf(sin_theta_2);
        float sin_phi, cos_phi;
        if (__builtin_fabsf(sin_theta_2) <= 4.f * kEpsilon) {
            sin_phi = 0.f;
            cos_phi = 1.f;
        } else {
            sin_phi = __builtin_fminf(__builtin_fmaxf(wi_p.y * inv_sin_theta, -1.f), 1.f);
            cos_phi = __builtin_fminf(__builtin_fmaxf(wi_p.x * inv_sin_theta, -1.f), 1.f);
        }
        float slx, sly;
        mf_sample_visible_11(d, wi_p.z, sx, sy, slx, sly);
        float rx = fmsub(cos_phi, slx, sin_phi * sly) * d.au;
        float ry = fmadd(sin_phi, slx, cos_phi * sly) * d.av;
        m = normalize(mk(-rx, -ry, 1.f));
        pdf = mf_eval(d, m) * mf_smith_g1(d, wi, m) * __builtin_fabsf(dot(wi, m)) / wi.z;
    }
}
BF_DEV float mf_pdf(const Microfacet &d, V3 wi, V3 m) {
    float result = mf_eval(d, m);
    if (d.sample_visible)
        result *= mf_smith_g1(d, wi, m) * __builtin_fabsf(dot(wi, m)) / wi.z;
    else
        result *= m.z;
    return result;
}
BF_DEV float fresnel_conductor(float cos_theta_i, float eta_r, float eta_i) {
    float cos_theta_i_2 = cos_theta_i * cos_theta_i, sin_theta_i_2 = 1.f - cos_theta_i_2,
          sin_theta_i_4 = sin_theta_i_2 * sin_theta_i_2;
    float temp_1 = eta_r * eta_r - eta_i * eta_i - sin_theta_i_2,
          a_2_pb_2 = safe_sqrt(temp_1 * temp_1 + 4.f * eta_i * eta_i * eta_r * eta_r),
          a = safe_sqrt(.5f * (a_2_pb_2 + temp_1));
    float term_1 = a_2_pb_2 + cos_theta_i_2, term_2 = 2.f * cos_theta_i * a;
    float r_s = (term_1 - term_2) / (term_1 + term_2);
    float term_3 = a_2_pb_2 * cos_theta_i_2 + sin_theta_i_4, term_4 = term_2 * sin_theta_i_2;
    float r_p = r_s * (term_3 - term_4) / (term_3 + term_4);
    return .5f * (r_s + r_p);
}
BF_DEV V3 reflect(V3 wi, V3 m) {
    float d2 = 2.f * dot(wi, m);
    return mk(fmsub(m.x, d2, wi.x), fmsub(m.y, d2, wi.y), fmsub(m.z, d2, wi.z));
}

struct BSDFSample {
    V3 wo;
    float pdf, eta;
};

BF_DEV float bsdf_sample_1(const bf_material &mat, V3 wi, float s2x, float s2y, BSDFSample &bs) {
    bs.wo = mk(0.f, 0.f, 0.f);
    bs.pdf = 0.f;
    bs.eta = 1.f;
    float cos_theta_i = wi.z;
    if (!(cos_theta_i > 0.f)) return 0.f;
    if (mat.type == BF_BSDF_DIFFUSE) {
        bs.wo = square_to_cosine_hemisphere(s2x, s2y);
        bs.pdf = kInvPi * bs.wo.z;
        return (bs.pdf > 0.f) ? mat.reflectance : 0.f;
    } else if (mat.type == BF_BSDF_ROUGHCONDUCTOR) {
        Microfacet distr = mf_make(mat);
        V3 m;
        mf_sample(distr, wi, s2x, s2y, m, bs.pdf);
        bs.wo = reflect(wi, m);
        bool active = bs.pdf != 0.f && bs.wo.z > 0.f;
        float weight;
        if (distr.sample_visible)
            weight = mf_smith_g1(distr, bs.wo, m);
        else
            weight = mf_G(distr, wi, bs.wo, m) * dot(wi, m) / (cos_theta_i * m.z);
        bs.pdf /= 4.f * dot(bs.wo, m);
        float F = fresnel_conductor(dot(wi, m), mat.eta, mat.k);
        if (mat.has_specular_reflectance) weight *= mat.reflectance;
        return active ? F * weight : 0.f;
    }
    return 0.f;
}
BF_DEV float bsdf_eval_1(const bf_material &mat, V3 wi, V3 wo) {
    float cos_theta_i = wi.z, cos_theta_o = wo.z;
    bool active = cos_theta_i > 0.f && cos_theta_o > 0.f;
    if (mat.type == BF_BSDF_DIFFUSE) {
        float value = mat.reflectance * kInvPi * cos_theta_o;
        return active ? value : 0.f;
    } else if (mat.type == BF_BSDF_ROUGHCONDUCTOR) {
        if (!active) return 0.f;
        V3 H = normalize(wo + wi);
        Microfacet distr = mf_make(mat);
        float D = mf_eval(distr, H);
        active = active && D != 0.f;
        float G = mf_G(distr, wi, wo, H);
        float result = D * G / (4.f * wi.z);
        float F = fresnel_conductor(dot(wi, H), mat.eta, mat.k);
        if (mat.has_specular_reflectance) result *= mat.reflectance;
        return active ? F * result : 0.f;
    }
    return 0.f;
}
BF_DEV float bsdf_pdf_1(const bf_material &mat, V3 wi, V3 wo) {
    float cos_theta_i = wi.z, cos_theta_o = wo.z;
    if (mat.type == BF_BSDF_DIFFUSE) {
        float pdf = kInvPi * wo.z;
        return (cos_theta_i > 0.f && cos_theta_o > 0.f) ? pdf : 0.f;
    } else if (mat.type == BF_BSDF_ROUGHCONDUCTOR) {
        V3 m = normalize(wo + wi);
        bool active = cos_theta_i > 0.f && cos_theta_o > 0.f && dot(wi, m) > 0.f && dot(wo, m) > 0.f;
        if (!active) return 0.f;
        Microfacet distr = mf_make(mat);
        if (distr.sample_visible) return mf_eval(distr, m) * mf_smith_g1(distr, wi, m) / (4.f * cos_theta_i);
        return mf_pdf(distr, wi, m) / (4.f * dot(wo, m));
    }
    return 0.f;
}
// TwoSidedBRDF with one nested BSDF on both sides: flip wi.z / wo.z
BF_DEV float bsdf_sample(const bf_material &mat, V3 wi, float s2x, float s2y, BSDFSample &bs) {
    bool flip = mat.twosided && wi.z < 0.f;
    if (mat.twosided && wi.z == 0.f) {
        bs.wo = mk(0.f, 0.f, 0.f);
        bs.pdf = 0.f;
        bs.eta = 1.f;
        return 0.f;
    }
    if (flip) wi.z *= -1.f;
    float r = bsdf_sample_1(mat, wi, s2x, s2y, bs);
    if (flip) bs.wo.z *= -1.f;
    return r;
}
BF_DEV void bsdf_eval_pdf(const bf_material &mat, V3 wi, V3 wo, float &ev, float &pdf) {
    if (mat.twosided) {
        if (wi.z == 0.f) {
            ev = pdf = 0.f;
            return;
        }
        if (wi.z < 0.f) {
            wi.z *= -1.f;
            wo.z *= -1.f;
        }
    }
    ev = bsdf_eval_1(mat, wi, wo);
    pdf = bsdf_pdf_1(mat, wi, wo);
}
BF_DEV bool bsdf_smooth(const bf_material &mat) {
    return mat.type == BF_BSDF_DIFFUSE || mat.type == BF_BSDF_ROUGHCONDUCTOR;
}

// ---------------------------------------------------------------------------
// emitters: spot.cpp:97-164, area.cpp:66-186, shape.cpp:323-356,
// rectangle.cpp:111-125; scene.cpp:180-247
// ---------------------------------------------------------------------------
struct DirSample {
    V3 d;
    float pdf, dist;
    bool delta;
};

BF_DEV float spot_falloff(const DEmitter &e, V3 d) {
    float result = e.radiance;
    V3 local_dir = normalize(d);
    float cos_theta = local_dir.z;
    float beam_res = (cos_theta >= e.cos_beam) ? result : result * ((e.cutoff - acos_cr(cos_theta)) * e.inv_transition);
    return (cos_theta <= e.cos_cutoff) ? 0.f : beam_res;
}

BF_DEV float emitter_sample_direction(const DScene &sc, const DEmitter &e, V3 ref_p, float sx, float sy, DirSample &ds) {
    if (e.type == BF_EMITTER_SPOT) {
        V3 p = mk(e.to_world[3], e.to_world[7], e.to_world[11]);
        ds.pdf = 1.f;
        ds.delta = true;
        ds.d = p - ref_p;
        ds.dist = norm(ds.d);
        float inv_dist = rcp(ds.dist);
        ds.d = ds.d * inv_dist;
        V3 local_d = xf_vector(e.to_object, -ds.d);
        return spot_falloff(e, local_d) * (inv_dist * inv_dist);
    } else {
        const DRect &rc = sc.rects[e.rect];
        V3 p = xf_point(rc.to_world, mk(sx * 2.f - 1.f, sy * 2.f - 1.f, 0.f));
        V3 n = mk(rc.n[0], rc.n[1], rc.n[2]);
        ds.pdf = rc.inv_area;
        ds.delta = false;
        ds.d = p - ref_p;
        float dist_squared = squared_norm(ds.d);
        ds.dist = __builtin_sqrtf(dist_squared);
        ds.d = ds.d / ds.dist;
        float dp = __builtin_fabsf(dot(ds.d, n));
        ds.pdf *= (dp != 0.f) ? dist_squared / dp : 0.f;
        bool active = dot(ds.d, n) < 0.f && ds.pdf != 0.f;
        float spec = e.radiance / ds.pdf;
        return active ? spec : 0.f;
    }
}

// pdf_emitter_direction for the hit `p_hit` (normal n_hit) seen from `p_ref`
BF_DEV float emitter_pdf_direction(const DScene &sc, const DEmitter &e, V3 p_ref, V3 p_hit, V3 n_hit) {
    if (e.type == BF_EMITTER_SPOT) return 0.f;
    const DRect &rc = sc.rects[e.rect];
    V3 d = p_hit - p_ref;
    float dist = norm(d);
    d = d / dist;
    float dp = dot(d, n_hit);
    float pdf = rc.inv_area, adp = __builtin_fabsf(dot(d, n_hit));
    pdf *= (adp != 0.f) ? (dist * dist) / adp : 0.f;
    return (dp < 0.f) ? pdf : 0.f;
}

BF_DEV float mis_weight(float pdf_a, float pdf_b) {   // path.cpp:222-226
    pdf_a *= pdf_a;
    pdf_b *= pdf_b;
    return pdf_a > 0.f ? pdf_a / (pdf_a + pdf_b) : 0.f;
}

// spectrum.h:281-287 applied to a grey colour: M * (l,l,l)
BF_DEV void srgb_to_xyz_grey(float l, float &X, float &Y, float &Z) {
    X = fmadd(0.180423f, l, fmadd(0.357580f, l, 0.412453f * l));
    Y = fmadd(0.072169f, l, fmadd(0.715160f, l, 0.212671f * l));
    Z = fmadd(0.950227f, l, fmadd(0.119193f, l, 0.019334f * l));
}

// sensor rays: fluxmeter.cpp:63-85, perspective.cpp:172-199
BF_DEV float sensor_sample_ray(const DScene &sc, float px, float py, float ax, float ay, V3 &o, V3 &d, float &mint,
                               float &maxt) {
    const DSensor &s = sc.sensor;
    if (s.type == BF_SENSOR_FLUXMETER) {
        const DRect &rc = sc.rects[s.rect];
        o = xf_point(rc.to_world, mk(px * 2.f - 1.f, py * 2.f - 1.f, 0.f));
        V3 local = square_to_cosine_hemisphere(ax, ay);
        Frame f;
        f.n = mk(rc.n[0], rc.n[1], rc.n[2]);
        coordinate_system(f.n, f.s, f.t);
        d = to_world(f, local);
        mint = kRayEpsilon;
        maxt = BF_INF;
        return 1.f * kPi;
    } else {
        V3 near_p = xf_point_proj(s.sample_to_camera, mk(px, py, 0.f));
        V3 dl = normalize(near_p);
        float inv_z = rcp(dl.z);
        mint = s.near_clip * inv_z;
        maxt = s.far_clip * inv_z;
        o = xf_point(s.to_world, mk(0.f, 0.f, 0.f));
        d = xf_vector(s.to_world, dl);
        return 1.f;
    }
}

// ---------------------------------------------------------------------------
// the render kernel
// ---------------------------------------------------------------------------
BF_DEV void hist_add(float *s_hist, float *g_hist, bool lds, uint32_t idx, float v) {
    if (lds)
        atomicAdd(&s_hist[idx], v);     // ds_add_f32
    else
        atomicAdd(&g_hist[idx], v);     // global_atomic_add_f32
}

template <bool STATS>
__global__ __launch_bounds__(kBlock) void bf_render_kernel(DScene sc, DLaunch lp, float *__restrict__ g_hist,
                                                           bf_path_record *__restrict__ records,
                                                           unsigned long long *__restrict__ counters) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    int *s_stack = reinterpret_cast<int *>(s_raw);                         // [kStackDepth][kBlock]
    float *s_hist = reinterpret_cast<float *>(s_raw + sizeof(int) * kStackDepth * kBlock);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const bool lds_hist = lp.lds_hist != 0;
    if (lds_hist) {
        for (uint32_t i = tid; i < lp.n_chan; i += kBlock) s_hist[i] = 0.f;
        __syncthreads();
    }
    int *stack = s_stack + tid;

    const bool is_range = lp.mode == BF_MODE_RANGE, is_time = lp.mode == BF_MODE_TIME;
    const uint32_t n_emit = sc.n_emitters;
    const bool aperture = sc.sensor.type != BF_SENSOR_PERSPECTIVE;   // endpoint.h:241, perspective.cpp:130

    // per-lane path state
    bool alive = false, done = false;
    Rng rng;
    rng.state = 0;
    uint64_t path_i = 0;
    float throughput = 1.f, eta = 1.f, emission_weight = 1.f, result = 0.f, aux = 0.f, sensor_w = 1.f;
    int depth = 0;
    bool valid_ray = false, film_ok = true;
    uint32_t n_rays = 0;
    V3 ro = mk(0, 0, 0), rd = mk(0, 0, 1);
    float rmint = 0.f, rmaxt = 0.f;
    V3 prev_p = mk(0, 0, 0);
    float bs_pdf = 0.f;
    // per-lane accumulators of the five base channels X,Y,Z,alpha,weight
    float accX = 0.f, accY = 0.f, accZ = 0.f, accA = 0.f, accW = 0.f;
    // statistics
    uint32_t c_closest = 0, c_shadow = 0, c_nodes = 0, c_tris = 0, c_invalid = 0, c_bounces = 0;
    // wave-local pool of path indices (uniform across the wave)
    uint64_t pool_next = 0, pool_end = 0;

    while (true) {
        // ---- 1. path regeneration: dead lanes pull new path indices -------
        unsigned long long need = __ballot(!alive && !done);
        if (need) {
            uint32_t n_need = __popcll(need);
            if (pool_end - pool_next < n_need && pool_end != ~0ull) {
                // refill: one returning atomic per wave for 4 x 64 paths
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(&counters[CTR_NEXT_PATH], 256ull);
                base = __shfl(base, 0);
                // hand out what is left of the old pool first is not worth the
                // bookkeeping: indices are only claimed, never skipped, because
                // the remainder of the old pool is consumed below before the new one
                if (pool_next < pool_end) {
                    // serve the old remainder to the first lanes, rest from the new chunk
                    uint32_t rank = __popcll(need & ((1ull << lane) - 1ull));
                    uint64_t left = pool_end - pool_next;
                    if (!alive && !done) {
                        path_i = rank < left ? pool_next + rank : base + (rank - left);
                    }
                    pool_next = base + (n_need - left);
                    pool_end = base + 256ull;
                } else {
                    uint32_t rank = __popcll(need & ((1ull << lane) - 1ull));
                    if (!alive && !done) path_i = base + rank;
                    pool_next = base + n_need;
                    pool_end = base + 256ull;
                }
            } else {
                uint32_t rank = __popcll(need & ((1ull << lane) - 1ull));
                if (!alive && !done) path_i = pool_next + rank;
                pool_next += n_need;
            }
            if (!alive && !done) {
                if (path_i >= lp.n_paths) {
                    done = true;
                } else {
                    // SamplingIntegrator::render_sample — integrator.cpp:259-283,
                    // per-path stream seed(base_seed + path) (sampler.cpp:83-96)
                    pcg_seed(rng, lp.seed + lp.path_offset + path_i);
                    float fx = next_1d(rng), fy = next_1d(rng);
                    float ax = .5f, ay = .5f;
                    if (aperture) {
                        ax = next_1d(rng);
                        ay = next_1d(rng);
                    }
                    if (sc.sensor.shutter_open_time > 0.f) (void) next_1d(rng);
                    (void) next_1d(rng);   // wavelength sample (consumed in RGB mode too)
                    sensor_w = sensor_sample_ray(sc, fx, fy, ax, ay, ro, rd, rmint, rmaxt);
                    // ImageBlock::put box branch: lo = ceil(pos - .5 - .5) must be 0
                    film_ok = __builtin_ceilf((fx - .5f) - .5f) == 0.f && __builtin_ceilf((fy - .5f) - .5f) == 0.f;
                    throughput = 1.f;
                    eta = 1.f;
                    emission_weight = 1.f;
                    result = 0.f;
                    aux = 0.f;
                    depth = 0;
                    n_rays = 0;
                    alive = true;
                }
            }
        }
        if (__ballot(alive) == 0ull) break;

        // ---- 2. closest-hit traversal for every live lane ------------------
        Hit hit;
        hit.t = BF_INF;
        if (alive) {
            traverse<false, STATS>(sc, ro, rd, rmint, rmaxt, stack, hit, c_nodes, c_tris);
            ++c_closest;
            ++n_rays;
        }

        // ---- 3. vertex logic up to the shadow ray ---------------------------
        SI si;
        bool si_valid = false;
        bool terminate = false, want_shadow = false, nee = false;
        DirSample ds;
        ds.d = mk(0, 0, 1);
        ds.pdf = 0.f;
        ds.dist = 0.f;
        ds.delta = false;
        float emitter_val = 0.f;
        uint32_t mat_id = 0;
        if (alive) {
            si_valid = hit.t != BF_INF;
            int emitter = -1;
            if (si_valid) {
                make_si(sc, ro, rd, hit, si);
                emitter = sc.shapes[si.shape].emitter;
            }
            if (depth == 0) {
                // first intersection — path.cpp:115-117, pathlength.cpp:138-146, pathtime.cpp:136-140
                valid_ray = si_valid;
                if (is_range) aux += si_valid ? si.t : 0.f;
                if (is_time) aux = si_valid ? si.t / lp.time_c : 0.f;
                depth = 1;
            } else {
                // tail of the previous iteration — path.cpp:184-209
                if (emitter >= 0) {
                    const DEmitter &e = sc.emitters[emitter];
                    float emitter_pdf = emitter_pdf_direction(sc, e, prev_p, si.p, si.sh.n);
                    if (n_emit != 1) emitter_pdf *= 1.f / (float) n_emit;
                    emission_weight = mis_weight(bs_pdf, emitter_pdf);
                }
                if (is_range) aux += si_valid ? si.t : 0.f;
                if (is_time) aux += si_valid ? si.t / lp.time_c : 0.f;
                ++depth;
            }
            // head of iteration `depth` — path.cpp:121-145
            if (emitter >= 0) {
                const DEmitter &e = sc.emitters[emitter];
                float ev = (e.type == BF_EMITTER_SPOT) ? 0.f : ((si.wi.z > 0.f) ? e.radiance : 0.f);
                result += emission_weight * throughput * ev;
                if (is_range) aux += si_valid ? si.t : 0.f;       // pathlength.cpp:161
            }
            bool active = si_valid;
            if (depth > lp.rr_depth) {
                float q = __builtin_fminf(throughput * sqr(eta), .95f);
                active = (next_1d(rng) < q) && active;
                throughput *= rcp(q);
            }
            if ((uint32_t) depth >= (uint32_t) lp.max_depth || !active) {
                terminate = true;
            } else {
                mat_id = sc.shapes[si.shape].material;
                const bf_material &mat = sc.materials[mat_id];
                ++c_bounces;
                nee = bsdf_smooth(mat);
                if (nee) {
                    // Scene::sample_emitter_direction — scene.cpp:180-230
                    float sx = next_1d(rng), sy = next_1d(rng);
                    if (n_emit == 0) {
                        ds.pdf = 0.f;
                        emitter_val = 0.f;
                    } else if (n_emit == 1) {
                        emitter_val = emitter_sample_direction(sc, sc.emitters[0], si.p, sx, sy, ds);
                    } else {
                        float emitter_pdf = 1.f / (float) n_emit;
                        uint32_t index = min((uint32_t) (sx * (float) n_emit), n_emit - 1u);
                        sx = (sx - index * emitter_pdf) * (float) n_emit;
                        emitter_val = emitter_sample_direction(sc, sc.emitters[index], si.p, sx, sy, ds);
                        ds.pdf *= emitter_pdf;
                        emitter_val *= rcp(emitter_pdf);
                    }
                    want_shadow = ds.pdf != 0.f;
                }
            }
        }

        // ---- 4. shadow (any-hit) traversal ----------------------------------
        if (__ballot(want_shadow)) {
            if (want_shadow) {
                Hit sh;
                float smint = kRayEpsilon * (1.f + hmax_abs(si.p));
                float smaxt = ds.dist * (1.f - kShadowEpsilon);
                bool occluded = traverse<true, STATS>(sc, si.p, ds.d, smint, smaxt, stack, sh, c_nodes, c_tris);
                ++c_shadow;
                ++n_rays;
                if (occluded) emitter_val = 0.f;
            }
        }

        // ---- 5. NEE contribution, BSDF sampling, next ray --------------------
        if (alive && !terminate) {
            const bf_material &mat = sc.materials[mat_id];
            if (nee) {
                bool active_e = ds.pdf != 0.f;
                V3 wo = to_local(si.sh, ds.d);
                float bsdf_val, bsdf_pdf;
                bsdf_eval_pdf(mat, si.wi, wo, bsdf_val, bsdf_pdf);
                float mis = ds.delta ? 1.f : mis_weight(ds.pdf, bsdf_pdf);
                if (active_e) result += mis * throughput * bsdf_val * emitter_val;
                if (is_range) aux += si.t;                          // pathlength.cpp:209
            }
            (void) next_1d(rng);                                    // sample1 (unused by these BSDFs)
            float s2x = next_1d(rng), s2y = next_1d(rng);
            BSDFSample bs;
            float bsdf_val = bsdf_sample(mat, si.wi, s2x, s2y, bs);
            throughput = throughput * bsdf_val;
            if (throughput == 0.f) {
                terminate = true;
            } else {
                eta *= bs.eta;
                // si.spawn_ray — interaction.h:61-64
                ro = si.p;
                rd = to_world(si.sh, bs.wo);
                rmint = (1.f + hmax_abs(si.p)) * kRayEpsilon;
                rmaxt = BF_INF;
                prev_p = si.p;
                bs_pdf = bs.pdf;
            }
        }

        // ---- 6. film: render_sample tail + range/time AOVs + ImageBlock::put --
        if (alive && terminate) {
            float L = sensor_w * result;                            // integrator.cpp:286
            float X, Y, Z;
            if (lp.color_mode == BF_COLOR_RGB)
                srgb_to_xyz_grey(L, X, Y, Z);
            else
                X = Y = Z = L;
            float a0 = result, a1 = result, a2 = result;            // AOVs see the unweighted radiance
            if (is_time && lp.color_mode == BF_COLOR_RGB) srgb_to_xyz_grey(result, a0, a1, a2);
            bool ok = film_ok && __builtin_isfinite(X) && __builtin_isfinite(Y) && __builtin_isfinite(Z);
            if (is_range || is_time) ok = ok && __builtin_isfinite(a0) && __builtin_isfinite(a1) && __builtin_isfinite(a2);
            if (ok) {
                accX += X;
                accY += Y;
                accZ += Z;
                accA += valid_ray ? 1.f : 0.f;
                accW += 1.f;
                if (is_range || is_time) {
                    // range.cpp:141-161 / time.cpp:134-153: bin i takes the sample
                    // iff (float)i*w <= aux < (float)i*w + w, evaluated exactly
                    // as written there for the (at most three) candidate bins
                    float w = lp.bin_width;
                    int k = (int) __builtin_floorf(aux / w);
                    for (int i = k - 1; i <= k + 1; ++i) {
                        if (i < 0 || i >= (int) lp.bins) continue;
                        float lo = (float) i * w, hi = (float) i * w + w;
                        if (aux >= lo && aux < hi) {
                            if (is_range) {
                                if (a0 != 0.f) hist_add(s_hist, g_hist, lds_hist, 5u + (uint32_t) i, a0);
                            } else if (a0 != 0.f || a1 != 0.f || a2 != 0.f) {
                                hist_add(s_hist, g_hist, lds_hist, 5u + 3u * (uint32_t) i + 0u, a0);
                                hist_add(s_hist, g_hist, lds_hist, 5u + 3u * (uint32_t) i + 1u, a1);
                                hist_add(s_hist, g_hist, lds_hist, 5u + 3u * (uint32_t) i + 2u, a2);
                            }
                        }
                    }
                }
            } else {
                ++c_invalid;
            }
            if (records) {
                bf_path_record r;
                r.L = L;
                r.aux = aux;
                r.valid = valid_ray ? 1u : 0u;
                r.n_rays = n_rays;
                records[path_i] = r;
            }
            alive = false;
        }
    }

    // ---- epilogue: wave-reduce the base channels, flush the histogram ------
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        accX += __shfl_down(accX, off);
        accY += __shfl_down(accY, off);
        accZ += __shfl_down(accZ, off);
        accA += __shfl_down(accA, off);
        accW += __shfl_down(accW, off);
    }
    if (lane == 0) {
        hist_add(s_hist, g_hist, lds_hist, 0, accX);
        hist_add(s_hist, g_hist, lds_hist, 1, accY);
        hist_add(s_hist, g_hist, lds_hist, 2, accZ);
        hist_add(s_hist, g_hist, lds_hist, 3, accA);
        hist_add(s_hist, g_hist, lds_hist, 4, accW);
    }
    if (lds_hist) {
        __syncthreads();
        for (uint32_t i = tid; i < lp.n_chan; i += kBlock) {
            float v = s_hist[i];
            if (v != 0.f) atomicAdd(&g_hist[i], v);
        }
    }
    // statistics: wave-reduce then one atomic per counter per wave
    unsigned long long v_closest = c_closest, v_shadow = c_shadow, v_nodes = c_nodes, v_tris = c_tris,
                       v_invalid = c_invalid, v_bounces = c_bounces;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        v_closest += __shfl_down(v_closest, off);
        v_shadow += __shfl_down(v_shadow, off);
        v_nodes += __shfl_down(v_nodes, off);
        v_tris += __shfl_down(v_tris, off);
        v_invalid += __shfl_down(v_invalid, off);
        v_bounces += __shfl_down(v_bounces, off);
    }
    if (lane == 0) {
        atomicAdd(&counters[CTR_CLOSEST], v_closest);
        atomicAdd(&counters[CTR_SHADOW], v_shadow);
        if (STATS) {
            atomicAdd(&counters[CTR_NODES], v_nodes);
            atomicAdd(&counters[CTR_TRIS], v_tris);
        }
        atomicAdd(&counters[CTR_INVALID], v_invalid);
        atomicAdd(&counters[CTR_BOUNCES], v_bounces);
    }
}

// Scene::ray_intersect / ray_test over a batch of rays (tests, tools)
__global__ __launch_bounds__(kBlock) void bf_trace_kernel(DScene sc, uint64_t n, const float *__restrict__ rays,
                                                          int any_hit, float *__restrict__ out_t,
                                                          uint32_t *__restrict__ out_prim, uint32_t *__restrict__ out_shape,
                                                          float *__restrict__ out_uv, uint8_t *__restrict__ out_hit) {
    __shared__ int s_stack[kStackDepth * kBlock];
    int *stack = s_stack + threadIdx.x;
    uint64_t i = (uint64_t) blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float *r = rays + 8 * i;
    V3 o = mk(r[0], r[1], r[2]), d = mk(r[4], r[5], r[6]);
    float mint = r[3], maxt = r[7];
    Hit h;
    uint32_t a = 0, b = 0;
    if (any_hit) {
        out_hit[i] = traverse<true, false>(sc, o, d, mint, maxt, stack, h, a, b) ? 1 : 0;
    } else {
        bool valid = traverse<false, false>(sc, o, d, mint, maxt, stack, h, a, b);
        if (out_t) out_t[i] = h.t;
        if (out_prim) out_prim[i] = valid ? h.prim : 0xffffffffu;
        if (out_shape) {
            uint32_t s = 0xffffffffu;
            if (valid) s = h.slot < 0 ? sc.rects[-h.slot - 1].shape : __float_as_uint(sc.tris[3 * (size_t) h.slot + 1].w);
            out_shape[i] = s;
        }
        if (out_uv) {
            out_uv[2 * i] = h.u;
            out_uv[2 * i + 1] = h.v;
        }
    }
}

}  // namespace bfd

// host-callable launchers (used by bf_api.cpp, which is plain C++)
extern "C" hipError_t bfk_launch_render(const bfd::DScene *sc, const bfd::DLaunch *lp, float *g_hist, bf_path_record *records,
                                        unsigned long long *counters, int stats, unsigned grid, size_t lds_bytes,
                                        hipStream_t stream) {
    if (stats)
        hipLaunchKernelGGL(bfd::bf_render_kernel<true>, dim3(grid), dim3(bfd::kBlock), lds_bytes, stream, *sc, *lp, g_hist,
                           records, counters);
    else
        hipLaunchKernelGGL(bfd::bf_render_kernel<false>, dim3(grid), dim3(bfd::kBlock), lds_bytes, stream, *sc, *lp, g_hist,
                           records, counters);
    return hipGetLastError();
}

extern "C" hipError_t bfk_launch_trace(const bfd::DScene *sc, uint64_t n, const float *rays, int any_hit, float *out_t,
                                       uint32_t *out_prim, uint32_t *out_shape, float *out_uv, uint8_t *out_hit,
                                       hipStream_t stream) {
    unsigned grid = (unsigned) ((n + bfd::kBlock - 1) / bfd::kBlock);
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(bfd::bf_trace_kernel, dim3(grid), dim3(bfd::kBlock), 0, stream, *sc, n, rays, any_hit, out_t,
                       out_prim, out_shape, out_uv, out_hit);
    return hipGetLastError();
}
