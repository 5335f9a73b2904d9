// Device-side scalar/vector math for the radar path tracer (gfx950).
//
// fp32 operation order follows the reference source text: a fused multiply-add
// appears exactly where the reference writes fmadd/fmsub/fnmadd (or where
// enoki's generic array code does: dot = fma chain from lane 0, cross = fmsub
// form, normalize = v * (1/sqrt(dot))), and nowhere else — the translation
// unit is compiled with -ffp-contract=off.  Division and sqrt are IEEE
// (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt).  Scalar
// transcendentals (sin, cos, acos, exp, log, erf) follow the fp32 specification
// below (bf_exp, bf_log, ...): fixed IEEE operation sequences shared with the
// oracle, within 2.5 ulp of the libm calls the reference's scalar variant makes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bfd {

#define BF_DEV __device__ __forceinline__
#define BF_HD __host__ __device__ __forceinline__

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

// ---------------------------------------------------------------------------
// fp32 elementary functions — the engine's SPECIFICATION of sin/cos/acos/exp/
// log/erf/tan.  The reference's scalar variants call libm (glibc) for these;
// here each function is a fixed sequence of IEEE fp32 operations (+ - * / fma
// sqrt rint, integer bit tricks), so that the HIP kernels and the CPU oracle
// produce bit-identical values without fp64 anywhere on the device.  Algorithms:
// Cephes single precision (expf, logf, sinf/cosf with three-part pi/4 reduction,
// asinf/acosf); erf: three-range polynomial fit (tools/gen_erf_coeffs.py).
// Accuracy vs libm: <= 2.5 ulp (tests/test_oracle_known_answers.py::test_elementary_functions).
// ---------------------------------------------------------------------------
BF_HD float bf_bits_to_float(uint32_t u) { return __builtin_bit_cast(float, u); }
BF_HD uint32_t bf_float_to_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

#ifdef BF_ESTRIN_PROBE
// TIMING PROBE ONLY (tools/r04_estrin_probe.sh; never the product, never compared with the oracle): the polynomials of exp / log /
// asin / erf evaluated in Estrin form (depth log2 n instead of n) to MEASURE what re-association would buy a lone path's bounce.
template <int N> BF_HD float bf_estrin(const float *c, float x) {          // c[0] + c[1] x + ... + c[N-1] x^(N-1)
    if constexpr (N == 1) {
        return c[0];
    } else if constexpr (N == 2) {
        return __builtin_fmaf(c[1], x, c[0]);
    } else {
        constexpr int H = N > 8 ? 8 : (N > 4 ? 4 : 2);
        float xh = x * x;
        if constexpr (H >= 4) xh = xh * xh;
        if constexpr (H >= 8) xh = xh * xh;
        return __builtin_fmaf(bf_estrin<N - H>(c + H, x), xh, bf_estrin<H>(c, x));
    }
}
#define BF_POLY(var, x, ...)                                  \
    do {                                                      \
        const float bf_c_[] = {__VA_ARGS__};                  \
        var = bf_estrin<sizeof(bf_c_) / sizeof(float)>(bf_c_, x); \
    } while (0)
#endif
BF_HD float bf_exp(float x) {
    if (!(x >= -87.0f)) return (x != x) ? x : 0.f;           // results below FLT_MIN are flushed to 0
    if (x > 88.72283905f) return __builtin_huge_valf();
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
#ifdef BF_ESTRIN_PROBE
    float p;
    BF_POLY(p, r, 5.0000001201e-1f, 1.6666665459e-1f, 4.1665795894e-2f, 8.3334519073e-3f, 1.3981999507e-3f, 1.9875691500e-4f);
#else
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
#endif
    p = __builtin_fmaf(p, r * r, r);
    p = p + 1.f;
    int ni = (int) n;
    if (ni > 127) {
        p = p * 2.f;
        ni -= 1;
    }
    return p * bf_bits_to_float((uint32_t) (ni + 127) << 23);
}

BF_HD float bf_log(float x) {
    if (!(x > 0.f)) return (x == 0.f) ? -__builtin_huge_valf() : __builtin_nanf("");
    if (x == __builtin_huge_valf()) return x;
    int e_adj = 0;
    if (x < 1.17549435e-38f) {
        x = x * 8388608.f;
        e_adj = -23;
    }
    uint32_t b = bf_float_to_bits(x);
    int e = (int) ((b >> 23) & 0xffu) - 126 + e_adj;
    float m = bf_bits_to_float((b & 0x007fffffu) | 0x3f000000u);     // [0.5, 1)
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.f;
    } else {
        m = m - 1.f;
    }
    float z = m * m;
#ifdef BF_ESTRIN_PROBE
    float y;
    BF_POLY(y, m, 3.3333331174e-1f, -2.4999993993e-1f, 2.0000714765e-1f, -1.6668057665e-1f, 1.4249322787e-1f, -1.2420140846e-1f, 1.1676998740e-1f, -1.1514610310e-1f, 7.0376836292e-2f);
#else
    float y = 7.0376836292e-2f;
    y = __builtin_fmaf(y, m, -1.1514610310e-1f);
    y = __builtin_fmaf(y, m, 1.1676998740e-1f);
    y = __builtin_fmaf(y, m, -1.2420140846e-1f);
    y = __builtin_fmaf(y, m, 1.4249322787e-1f);
    y = __builtin_fmaf(y, m, -1.6668057665e-1f);
    y = __builtin_fmaf(y, m, 2.0000714765e-1f);
    y = __builtin_fmaf(y, m, -2.4999993993e-1f);
    y = __builtin_fmaf(y, m, 3.3333331174e-1f);
#endif
    y = (y * m) * z;
    float fe = (float) e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = m + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    return r;
}

// sin and cos of x together (three-part Cody-Waite reduction by pi/4; exact for |x| < ~5e4)
BF_HD void bf_sincos(float x, float &s_out, float &c_out) {
    float ax = __builtin_fabsf(x);
    if (!(ax < 3.0e9f)) {          // inf / nan / absurdly large: NaN like libm would for inf
        s_out = c_out = (ax != ax || ax == __builtin_huge_valf()) ? __builtin_nanf("") : 0.f;
        if (ax == ax && ax != __builtin_huge_valf()) c_out = 1.f;
        return;
    }
    uint32_t j = (uint32_t) (ax * 1.27323954473516f);
    if (j & 1u) j += 1u;
    float y = (float) j;
    float r = __builtin_fmaf(-y, 0.78515625f, ax);
    r = __builtin_fmaf(-y, 2.4187564849853515625e-4f, r);
    r = __builtin_fmaf(-y, 3.77489497744594108e-8f, r);
    float z = r * r;
    float ps = -1.9515295891e-4f;
    ps = __builtin_fmaf(ps, z, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
    ps = __builtin_fmaf(ps * z, r, r);                       // sin(r)
    float pc = 2.443315711809948e-5f;
    pc = __builtin_fmaf(pc, z, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
    pc = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.f));     // cos(r)
    uint32_t q = j & 7u;                           // octant pair: 0,2,4,6
    float sv = (q == 2u || q == 6u) ? pc : ps;
    float cv = (q == 2u || q == 6u) ? ps : pc;
    if (q == 4u || q == 6u) sv = -sv;
    if (q == 2u || q == 4u) cv = -cv;
    s_out = (x < 0.f) ? -sv : sv;
    c_out = cv;
}
BF_HD float bf_sin(float x) {
    float s, c;
    bf_sincos(x, s, c);
    return s;
}
BF_HD float bf_cos(float x) {
    float s, c;
    bf_sincos(x, s, c);
    return c;
}
BF_HD float bf_tan(float x) {
    float s, c;
    bf_sincos(x, s, c);
    return s / c;
}

BF_HD float bf_asin_core(float a) {               // a in [0, 1]
    bool flag = a > 0.5f;
    float z, x;
    if (flag) {
        z = 0.5f * (1.f - a);
        x = __builtin_sqrtf(z);
    } else {
        x = a;
        z = x * x;
    }
#ifdef BF_ESTRIN_PROBE
    float p;
    BF_POLY(p, z, 1.6666752422e-1f, 7.4953002686e-2f, 4.5470025998e-2f, 2.4181311049e-2f, 4.2163199048e-2f);
#else
    float p = 4.2163199048e-2f;
    p = __builtin_fmaf(p, z, 2.4181311049e-2f);
    p = __builtin_fmaf(p, z, 4.5470025998e-2f);
    p = __builtin_fmaf(p, z, 7.4953002686e-2f);
    p = __builtin_fmaf(p, z, 1.6666752422e-1f);
#endif
    p = __builtin_fmaf(p * z, x, x);
    if (flag) p = 1.5707963267948966f - (p + p);
    return p;
}
BF_HD float bf_acos(float x) {
    if (!(x >= -1.f && x <= 1.f)) return __builtin_nanf("");
    if (x < -0.5f) return 3.14159265358979323846f - 2.f * bf_asin_core(__builtin_sqrtf(0.5f * (1.f + x)));
    if (x > 0.5f) return 2.f * bf_asin_core(__builtin_sqrtf(0.5f * (1.f - x)));
    float a = bf_asin_core(__builtin_fabsf(x));
    return 1.5707963267948966f - ((x < 0.f) ? -a : a);
}

BF_HD float bf_erf(float x) {
    float a = __builtin_fabsf(x);
    if (a != a) return x;
    float r;
    if (a < 0.8f) {
        float z = x * x;
#ifdef BF_ESTRIN_PROBE
        float p;
        BF_POLY(p, z, 1.128379167e+00f, -3.761263888e-01f, 1.128379095e-01f, -2.686608035e-02f, 5.223417615e-03f, -8.529368140e-04f, 1.169606002e-04f, -1.128872449e-05f);
#else
        float p = -1.128872449e-05f;
        p = __builtin_fmaf(p, z, 1.169606002e-04f);
        p = __builtin_fmaf(p, z, -8.529368140e-04f);
        p = __builtin_fmaf(p, z, 5.223417615e-03f);
        p = __builtin_fmaf(p, z, -2.686608035e-02f);
        p = __builtin_fmaf(p, z, 1.128379095e-01f);
        p = __builtin_fmaf(p, z, -3.761263888e-01f);
        p = __builtin_fmaf(p, z, 1.128379167e+00f);
#endif
        return p * x;
    } else if (a < 1.6f) {
        float t = a - 1.2f;
#ifdef BF_ESTRIN_PROBE
        float p;
        BF_POLY(p, t, 9.103139784e-01f, 2.673443467e-01f, -3.208132755e-01f, 1.675358269e-01f, 6.419227034e-03f, -5.334181702e-02f, 1.957314866e-02f, 5.989846125e-03f, -5.621837162e-03f, 3.210465009e-04f);
#else
        float p = 3.210465009e-04f;
        p = __builtin_fmaf(p, t, -5.621837162e-03f);
        p = __builtin_fmaf(p, t, 5.989846125e-03f);
        p = __builtin_fmaf(p, t, 1.957314866e-02f);
        p = __builtin_fmaf(p, t, -5.334181702e-02f);
        p = __builtin_fmaf(p, t, 6.419227034e-03f);
        p = __builtin_fmaf(p, t, 1.675358269e-01f);
        p = __builtin_fmaf(p, t, -3.208132755e-01f);
        p = __builtin_fmaf(p, t, 2.673443467e-01f);
        p = __builtin_fmaf(p, t, 9.103139784e-01f);
#endif
        r = p;
    } else if (a < 4.0f) {
        float t = a - 2.8f;
#ifdef BF_ESTRIN_PROBE
        float p;
        BF_POLY(p, t, -9.497846531e+00f, -5.921730786e+00f, -9.526015505e-01f, -8.600132565e-03f, 1.626972312e-03f, -3.038591482e-04f, 5.432784894e-05f, -9.056785943e-06f, 1.344892106e-06f, -1.412586080e-07f, 1.378729715e-09f);
#else
        float p = 1.378729715e-09f;
        p = __builtin_fmaf(p, t, -1.412586080e-07f);
        p = __builtin_fmaf(p, t, 1.344892106e-06f);
        p = __builtin_fmaf(p, t, -9.056785943e-06f);
        p = __builtin_fmaf(p, t, 5.432784894e-05f);
        p = __builtin_fmaf(p, t, -3.038591482e-04f);
        p = __builtin_fmaf(p, t, 1.626972312e-03f);
        p = __builtin_fmaf(p, t, -8.600132565e-03f);
        p = __builtin_fmaf(p, t, -9.526015505e-01f);
        p = __builtin_fmaf(p, t, -5.921730786e+00f);
        p = __builtin_fmaf(p, t, -9.497846531e+00f);
#endif
        r = 1.f - bf_exp(p);
    } else {
        r = 1.f;
    }
    return (x < 0.f) ? -r : r;
}

BF_DEV float sin_cr(float x) { return bf_sin(x); }
BF_DEV float cos_cr(float x) { return bf_cos(x); }
BF_DEV float acos_cr(float x) { return bf_acos(x); }
BF_DEV float exp_cr(float x) { return bf_exp(x); }
BF_DEV float log_cr(float x) { return bf_log(x); }
BF_DEV float erf_cr(float x) { return bf_erf(x); }
BF_DEV float tan_cr(float x) { return bf_tan(x); }
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
// (the matrix pointer is a template parameter: the scene's small tables are read through constant-address-space pointers,
// bf_device.h: BF_CAS)
template <class M> BF_DEV V3 xf_point(M m, V3 p) {
    V3 r = mk(m[3], m[7], m[11]);
    r = mk(fmadd(m[0], p.x, r.x), fmadd(m[4], p.x, r.y), fmadd(m[8], p.x, r.z));
    r = mk(fmadd(m[1], p.y, r.x), fmadd(m[5], p.y, r.y), fmadd(m[9], p.y, r.z));
    r = mk(fmadd(m[2], p.z, r.x), fmadd(m[6], p.z, r.y), fmadd(m[10], p.z, r.z));
    return r;
}
template <class M> BF_DEV V3 xf_vector(M m, V3 v) {
    V3 r = mk(m[0] * v.x, m[4] * v.x, m[8] * v.x);
    r = mk(fmadd(m[1], v.y, r.x), fmadd(m[5], v.y, r.y), fmadd(m[9], v.y, r.z));
    r = mk(fmadd(m[2], v.z, r.x), fmadd(m[6], v.z, r.y), fmadd(m[10], v.z, r.z));
    return r;
}
// 4x4 projective point transform (Transform::operator*(Point))
template <class M> BF_DEV V3 xf_point_proj(M m, V3 p) {
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
    float s, c;
    bf_sincos(phi, s, c);
    ox = r * c;
    oy = r * s;
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
#ifdef BF_ESTRIN_PROBE
        BF_POLY(p, w, 1.50140941f, 0.246640727f, -0.00417768164f, -0.00125372503f, 0.00021858087f, -4.39150654e-06f, -3.5233877e-06f, 3.43273939e-07f, 2.81022636e-08f);
        return p * x;
#endif
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
#ifdef BF_ESTRIN_PROBE
        BF_POLY(p, w, 2.83297682f, 1.00167406f, 0.00943887047f, -0.0076224613f, 0.00573950773f, -0.00367342844f, 0.00134934322f, 0.000100950558f, -0.000200214257f);
        return p * x;
#endif
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
