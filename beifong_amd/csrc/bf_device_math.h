// Device-side scalar/vector math for the radar path tracer (gfx950).
//
// fp32 operation order follows the reference source text: a fused multiply-add
// appears exactly where the reference writes fmadd/fmsub/fnmadd (or where
// enoki's generic array code does: dot = fma chain from lane 0, cross = fmsub
// form, normalize = v * (1/sqrt(dot))), and nowhere else — the translation
// unit is compiled with -ffp-contract=off.  Division and sqrt are IEEE
// (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt).  Scalar
// transcendentals (sin, cos, acos, exp, log, erf) are evaluated in double and
// rounded once, which is what the reference's scalar variant gets from libm to
// within rounding.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bfd {

#define BF_DEV __device__ __forceinline__

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kInvSqrtPi = 0.56418958354775628695f;
constexpr float kEpsilon = 5.9604644775390625e-8f;       // math.h Epsilon = FLT_EPSILON/2
constexpr float kRayEpsilon = kEpsilon * 1500.f;         // math.h RayEpsilon
constexpr float kShadowEpsilon = kRayEpsilon * 10.f;     // math.h ShadowEpsilon
#define BF_INF __builtin_huge_valf()

BF_DEV float fmadd(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
BF_DEV float fmsub(float a, float b, float c) { return __builtin_fmaf(a, b, -c); }
BF_DEV float fnmadd(float a, float b, float c) { return __builtin_fmaf(-a, b, c); }
BF_DEV float sqr(float x) { return x * x; }
BF_DEV float rcp(float x) { return 1.f / x; }
BF_DEV float safe_sqrt(float x) { return __builtin_sqrtf(__builtin_fmaxf(x, 0.f)); }
BF_DEV float sin_cr(float x) { return (float) ::sin((double) x); }
BF_DEV float cos_cr(float x) { return (float) ::cos((double) x); }
BF_DEV float acos_cr(float x) { return (float) ::acos((double) x); }
BF_DEV float exp_cr(float x) { return (float) ::exp((double) x); }
BF_DEV float log_cr(float x) { return (float) ::log((double) x); }
BF_DEV float erf_cr(float x) { return (float) ::erf((double) x); }
BF_DEV float tan_cr(float x) { return (float) ::tan((double) x); }
BF_DEV float mulsign(float a, float b) { return __builtin_signbitf(b) ? -a : a; }
BF_DEV float mulsign_neg(float a, float b) { return __builtin_signbitf(b) ? a : -a; }

struct V3 {
    float x, y, z;
};
BF_DEV V3 mk(float x, float y, float z) { return V3{x, y, z}; }
BF_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
BF_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
BF_DEV V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
BF_DEV V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
BF_DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
BF_DEV float dot(V3 a, V3 b) { return fmadd(a.z, b.z, fmadd(a.y, b.y, a.x * b.x)); }
BF_DEV float squared_norm(V3 a) { return dot(a, a); }
BF_DEV float norm(V3 a) { return __builtin_sqrtf(squared_norm(a)); }
BF_DEV V3 normalize(V3 a) { return a * (1.f / __builtin_sqrtf(squared_norm(a))); }
BF_DEV V3 cross(V3 a, V3 b) {
    return mk(fmsub(a.y, b.z, a.z * b.y), fmsub(a.z, b.x, a.x * b.z), fmsub(a.x, b.y, a.y * b.x));
}
BF_DEV float hmax_abs(V3 a) {
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z));
}
BF_DEV V3 fmadd3(V3 d, float t, V3 o) { return mk(fmadd(d.x, t, o.x), fmadd(d.y, t, o.y), fmadd(d.z, t, o.z)); }

// include/mitsuba/core/vector.h:116-136
BF_DEV void coordinate_system(V3 n, V3 &s, V3 &t) {
    float sign = __builtin_copysignf(1.f, n.z);
    float a = -rcp(sign + n.z);
    float b = n.x * n.y * a;
    s = mk(mulsign(sqr(n.x) * a, n.z) + 1.f, mulsign(b, n.z), mulsign_neg(n.x, n.z));
    t = mk(b, sign + sqr(n.y) * a, -n.y);
}

// include/mitsuba/core/frame.h:20-40
struct Frame {
    V3 s, t, n;
};
BF_DEV V3 to_local(const Frame &f, V3 v) { return mk(dot(v, f.s), dot(v, f.t), dot(v, f.n)); }
BF_DEV V3 to_world(const Frame &f, V3 v) { return f.s * v.x + f.t * v.y + f.n * v.z; }

// 3x4 row-major affine: Transform::transform_affine (include/mitsuba/core/transform.h)
BF_DEV V3 xf_point(const float *m, V3 p) {
    V3 r = mk(m[3], m[7], m[11]);
    r = mk(fmadd(m[0], p.x, r.x), fmadd(m[4], p.x, r.y), fmadd(m[8], p.x, r.z));
    r = mk(fmadd(m[1], p.y, r.x), fmadd(m[5], p.y, r.y), fmadd(m[9], p.y, r.z));
    r = mk(fmadd(m[2], p.z, r.x), fmadd(m[6], p.z, r.y), fmadd(m[10], p.z, r.z));
    return r;
}
BF_DEV V3 xf_vector(const float *m, V3 v) {
    V3 r = mk(m[0] * v.x, m[4] * v.x, m[8] * v.x);
    r = mk(fmadd(m[1], v.y, r.x), fmadd(m[5], v.y, r.y), fmadd(m[9], v.y, r.z));
    r = mk(fmadd(m[2], v.z, r.x), fmadd(m[6], v.z, r.y), fmadd(m[10], v.z, r.z));
    return r;
}
// 4x4 projective point transform (Transform::operator*(Point))
BF_DEV V3 xf_point_proj(const float *m, V3 p) {
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float acc = m[4 * i + 3];
        acc = fmadd(m[4 * i + 0], p.x, acc);
        acc = fmadd(m[4 * i + 1], p.y, acc);
        acc = fmadd(m[4 * i + 2], p.z, acc);
        r[i] = acc;
    }
    return mk(r[0] / r[3], r[1] / r[3], r[2] / r[3]);
}

// ---------------------------------------------------------------------------
// PCG32 (O'Neill) — enoki/random.h is absent from the reference tree; seeding
// per src/librender/sampler.cpp:83-96, next_1d/next_2d per
// src/samplers/independent.cpp:73-82.  Every path uses the default stream, so
// `inc` is a compile-time constant and only the 64-bit state lives in VGPRs.
// ---------------------------------------------------------------------------
constexpr uint64_t PCG32_DEFAULT_STREAM = 0xda3e39cb94b95bdbULL;
constexpr uint64_t PCG32_MULT = 0x5851f42d4c957f2dULL;
constexpr uint64_t PCG32_INC = (PCG32_DEFAULT_STREAM << 1) | 1ULL;
struct Rng {
    uint64_t state;
};
BF_DEV uint32_t pcg_next(Rng &r) {
    uint64_t old = r.state;
    r.state = old * PCG32_MULT + PCG32_INC;
    uint32_t xs = (uint32_t) (((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t) (old >> 59);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31));
}
BF_DEV void pcg_seed(Rng &r, uint64_t initstate) {
    r.state = 0;
    pcg_next(r);
    r.state += initstate;
    pcg_next(r);
}
BF_DEV float next_1d(Rng &r) { return __uint_as_float((pcg_next(r) >> 9) | 0x3f800000u) - 1.f; }

// ---------------------------------------------------------------------------
// warps — include/mitsuba/core/warp.h:54-90, 325-350, 446-490
// ---------------------------------------------------------------------------
BF_DEV void square_to_uniform_disk_concentric(float sx, float sy, float &ox, float &oy) {
    float x = fmsub(2.f, sx, 1.f), y = fmsub(2.f, sy, 1.f);
    bool is_zero = (x == 0.f) && (y == 0.f);
    bool q13 = __builtin_fabsf(x) < __builtin_fabsf(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = .25f * kPi * rp / r;
    if (q13) phi = .5f * kPi - phi;
    if (is_zero) phi = 0.f;
    double s, c;
    ::sincos((double) phi, &s, &c);
    ox = r * (float) c;
    oy = r * (float) s;
}
BF_DEV V3 square_to_cosine_hemisphere(float sx, float sy) {
    float px, py;
    square_to_uniform_disk_concentric(sx, sy, px, py);
    float z = safe_sqrt(1.f - fmadd(py, py, px * px));
    return mk(px, py, z);
}

// erfinv — Giles' single-precision polynomial (the algorithm enoki cites)
BF_DEV float erfinv_giles(float x) {
    float w = -log_cr((1.f - x) * (1.f + x));
    float p;
    if (w < 5.f) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = fmadd(p, w, 3.43273939e-07f);
        p = fmadd(p, w, -3.5233877e-06f);
        p = fmadd(p, w, -4.39150654e-06f);
        p = fmadd(p, w, 0.00021858087f);
        p = fmadd(p, w, -0.00125372503f);
        p = fmadd(p, w, -0.00417768164f);
        p = fmadd(p, w, 0.246640727f);
        p = fmadd(p, w, 1.50140941f);
    } else {
        w = __builtin_sqrtf(w) - 3.f;
        p = -0.000200214257f;
        p = fmadd(p, w, 0.000100950558f);
        p = fmadd(p, w, 0.00134934322f);
        p = fmadd(p, w, -0.00367342844f);
        p = fmadd(p, w, 0.00573950773f);
        p = fmadd(p, w, -0.0076224613f);
        p = fmadd(p, w, 0.00943887047f);
        p = fmadd(p, w, 1.00167406f);
        p = fmadd(p, w, 2.83297682f);
    }
    return p * x;
}

}  // namespace bfd
