// oracle/omath.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Scalar/vector arithmetic used by the CPU oracle.  The reference leans on glm (un-vendored,
// unpinned: /root/reference/xmake.lua:25) for every vector operation, so the exact operation order of
// glm 0.9.9's generic (non-SIMD) code paths is restated here by hand; each function names the glm
// routine it stands in for.  Compile with `-ffp-contract=off -fno-fast-math`: every `*`, `+`, `-`, `/`
// and sqrtf below is then one correctly-rounded IEEE-754 binary32 operation, which is what the HIP
// kernels execute as well (they carry their own, separately written, copy of these semantics in
// radish_pt_amd/csrc/device/rmath.h — nothing under oracle/ is included by the product).
#pragma once
#include <cstring>
#include <cfloat>
#include <cmath>
#include <cstdint>

namespace om {

struct vec2 {
    float x, y;
    vec2() : x(0.f), y(0.f) {}
    explicit vec2(float s) : x(s), y(s) {}
    vec2(float x_, float y_) : x(x_), y(y_) {}
};

struct vec3 {
    float x, y, z;
    vec3() : x(0.f), y(0.f), z(0.f) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    vec3(vec2 v, float z_) : x(v.x), y(v.y), z(z_) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

struct vec4 {
    float x, y, z, w;
};

// ---- vec2 -------------------------------------------------------------------------------------
inline vec2 operator+(vec2 a, vec2 b) { return {a.x + b.x, a.y + b.y}; }
inline vec2 operator-(vec2 a, vec2 b) { return {a.x - b.x, a.y - b.y}; }
inline vec2 operator*(vec2 a, vec2 b) { return {a.x * b.x, a.y * b.y}; }
inline vec2 operator*(vec2 a, float s) { return {a.x * s, a.y * s}; }
inline vec2 operator*(float s, vec2 a) { return {s * a.x, s * a.y}; }
inline vec2 operator/(float s, vec2 a) { return {s / a.x, s / a.y}; }
inline vec2 operator+(vec2 a, float s) { return {a.x + s, a.y + s}; }
inline vec2 operator-(float s, vec2 a) { return {s - a.x, s - a.y}; }
inline vec2 operator-(vec2 a) { return {-a.x, -a.y}; }
// glm::dot(vec2): tmp = a*b; tmp.x + tmp.y
inline float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }

// ---- vec3 -------------------------------------------------------------------------------------
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator/(vec3 a, vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 operator/(float s, vec3 a) { return {s / a.x, s / a.y, s / a.z}; }
inline vec3 operator+(vec3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
inline vec3 operator-(vec3 a, float s) { return {a.x - s, a.y - s, a.z - s}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 &operator+=(vec3 &a, vec3 b) { a = a + b; return a; }
inline vec3 &operator*=(vec3 &a, vec3 b) { a = a * b; return a; }
inline vec3 &operator*=(vec3 &a, float s) { a = a * s; return a; }
inline vec3 &operator/=(vec3 &a, float s) { a = a / s; return a; }

// glm::dot(vec3) — compute_dot<vec<3>>: tmp = a*b; return tmp.x + tmp.y + tmp.z  (left to right)
inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// glm::cross — (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)
inline vec3 cross(vec3 a, vec3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
// glm::length = sqrt(dot(v,v)); glm::inversesqrt(x) = 1/sqrt(x); glm::normalize(v) = v * inversesqrt(dot(v,v))
inline float length(vec3 v) { return sqrtf(dot(v, v)); }
inline float length(vec2 v) { return sqrtf(dot(v, v)); }
inline vec3 normalize(vec3 v) { return v * (1.f / sqrtf(dot(v, v))); }
// glm::distance(p0, p1) = length(p1 - p0)
inline float distance(vec3 p0, vec3 p1) { return length(p1 - p0); }
// glm::reflect(I, N) = I - N * dot(N, I) * 2
inline vec3 reflect(vec3 I, vec3 N) { return I - N * dot(N, I) * 2.f; }
// glm::mix(x, y, a) = x * (1 - a) + y * a
inline float mix(float x, float y, float a) { return x * (1.f - a) + y * a; }
inline vec3 mix(vec3 x, vec3 y, float a) { return x * (1.f - a) + y * a; }
inline vec3 mix(vec3 x, vec3 y, vec3 a) { return x * (vec3(1.f) - a) + y * a; }
// glm::min(x, y) = (y < x) ? y : x ; glm::max(x, y) = (x < y) ? y : x
inline float gmin(float x, float y) { return (y < x) ? y : x; }
inline float gmax(float x, float y) { return (x < y) ? y : x; }
inline int gmin(int x, int y) { return (y < x) ? y : x; }
inline vec3 gmin(vec3 a, vec3 b) { return {gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)}; }
inline vec3 gmax(vec3 a, vec3 b) { return {gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)}; }
inline vec3 gabs(vec3 a) { return {fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
// glm::fract(x) = x - floor(x)
inline float fract(float x) { return x - floorf(x); }
inline vec2 fract(vec2 v) { return {fract(v.x), fract(v.y)}; }
// glm::radians(d) = d * 0.01745329251994329576923690768489
inline float radians(float d) { return d * 0.01745329251994329576923690768489f; }

// C fminf/fmaxf as used by AABB::getDistMinMax/getDistMaxMin (/root/reference/src/bvh.h:72-86):
// a NaN operand yields the other operand.  Written out so the result does not depend on libm.
inline float c_fminf(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    return (b < a) ? b : a;
}
inline float c_fmaxf(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    return (a < b) ? b : a;
}

inline bool isnan_f(float x) { return x != x; }
inline bool isinf_f(float x) { return fabsf(x) == INFINITY; }

// ---- mat3 (column major, as glm) ---------------------------------------------------------------
struct mat3 {
    vec3 c[3];
    mat3() {}
    mat3(vec3 a, vec3 b, vec3 d) { c[0] = a; c[1] = b; c[2] = d; }
};
// glm mat3 * vec3: m[0][i]*v.x + m[1][i]*v.y + m[2][i]*v.z  (left to right)
inline vec3 operator*(const mat3 &m, vec3 v) {
    return {(m.c[0].x * v.x + m.c[1].x * v.y) + m.c[2].x * v.z,
            (m.c[0].y * v.x + m.c[1].y * v.y) + m.c[2].y * v.z,
            (m.c[0].z * v.x + m.c[1].z * v.y) + m.c[2].z * v.z};
}
// glm::inverse(mat3) — compute_inverse<3,3>: cofactors times one-over-determinant
inline mat3 inverse(const mat3 &m) {
    float m00 = m.c[0].x, m01 = m.c[0].y, m02 = m.c[0].z;
    float m10 = m.c[1].x, m11 = m.c[1].y, m12 = m.c[1].z;
    float m20 = m.c[2].x, m21 = m.c[2].y, m22 = m.c[2].z;
    float oneOverDet = 1.f / ((m00 * (m11 * m22 - m21 * m12) - m10 * (m01 * m22 - m21 * m02)) +
                              m20 * (m01 * m12 - m11 * m02));
    mat3 r;
    r.c[0].x = +(m11 * m22 - m21 * m12) * oneOverDet;
    r.c[1].x = -(m10 * m22 - m20 * m12) * oneOverDet;
    r.c[2].x = +(m10 * m21 - m20 * m11) * oneOverDet;
    r.c[0].y = -(m01 * m22 - m21 * m02) * oneOverDet;
    r.c[1].y = +(m00 * m22 - m20 * m02) * oneOverDet;
    r.c[2].y = -(m00 * m21 - m20 * m01) * oneOverDet;
    r.c[0].z = +(m01 * m12 - m11 * m02) * oneOverDet;
    r.c[1].z = -(m00 * m12 - m10 * m02) * oneOverDet;
    r.c[2].z = +(m00 * m11 - m10 * m01) * oneOverDet;
    return r;
}

// ---- sin/cos -----------------------------------------------------------------------------------
// The reference calls CUDA libdevice cosf/sinf (mathUtil.h:132-136), whose bits no other platform
// reproduces.  Oracle and HIP kernels therefore both evaluate this fixed recipe (Cephes sinf/cosf:
// 3-term Cody–Waite reduction by pi/2 and degree-7/8 minimax polynomials), using only binary32
// multiply/add/subtract in the written order — max error ≈ 1 ulp on [0, 2π].
inline void sincos_det(float x, float *s, float *c) {
    float kf = rintf(x * 0.63661977236758134f);  // round-half-even(x * 2/pi)
    int k = (int)kf;
    float r = ((x - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188216e-8f;
    float z = r * r;
    float sp = r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
    float cp = (1.f - z * 0.5f) +
               z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    switch (k & 3) {
    case 0: *s = sp; *c = cp; break;
    case 1: *s = cp; *c = -sp; break;
    case 2: *s = -sp; *c = -cp; break;
    default: *s = -cp; *c = sp; break;
    }
}

// atan2 for Math::toPlane (mathUtil.h:143-147; the reference calls CUDA atan2f through glm::atan(y, x)): Cephes atanf
// (range reduction at tan(pi/8), tan(3pi/8) + degree-9 odd polynomial) and its quadrant logic, binary32 operations in
// this order on both sides — max error ≈ 2 ulp.
inline float atan_det(float xx) {
    float x = xx < 0.f ? -xx : xx, y;
    if (x > 2.414213562373095f) {
        y = 1.5707963267948966f;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {
        y = 0.7853981633974483f;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y = 0.0f;
    }
    float z = x * x;
    y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
    return xx < 0.f ? -y : y;
}
inline float atan2_det(float y, float x) {
    const float PIF = 3.14159265358979323846f, PIO2F = 1.5707963267948966f;
    if (x == 0.f) {
        if (y < 0.f) return -PIO2F;
        if (y == 0.f) return 0.f;
        return PIO2F;
    }
    if (y == 0.f) return x < 0.f ? PIF : 0.f;
    float w = 0.f;
    if (x < 0.f) w = (y < 0.f) ? -PIF : PIF;
    return w + atan_det(y / x);
}


// ---- gamma --------------------------------------------------------------------------------------
// Math::gammaCorrection is glm::pow(color, 1/2.2) (mathUtil.h:124-126) = CUDA libdevice powf in the reference.
// Oracle and HIP both evaluate this fixed recipe instead: log2 of the mantissa folded into [sqrt(1/2), sqrt(2)) by the
// atanh series, times 1/2.2f, exp2 by a degree-6 polynomial on [-1/2, 1/2]; binary32 operations in the written order.
inline float pow_gamma_det(float x) {
    if (!(x > 0.f)) return x == 0.f ? 0.f : NAN;
    if (x == INFINITY) return x;
    int eAdj = 0;
    if (x < 1.17549435e-38f) {
        x = x * 16777216.f;
        eAdj = -24;
    }
    uint32_t bits;
    memcpy(&bits, &x, 4);
    int e = int((bits >> 23) & 0xffu) - 126 + eAdj;
    uint32_t mb = (bits & 0x007fffffu) | 0x3f000000u;
    float m;
    memcpy(&m, &mb, 4);
    if (m < 0.70710678118654752f) {
        m = m + m;
        e = e - 1;
    }
    float sN = (m - 1.f) / (m + 1.f);
    float z = sN * sN;
    float p = sN + sN * z * (0.333333333333f + z * (0.2f + z * (0.142857142857f + z * 0.111111111111f)));
    float l2 = p * 2.8853900817779268f;
    float t = (float(e) + l2) * (1.f / 2.2f);
    float n = rintf(t);
    float f = t - n;
    float q = 1.f + f * (0.69314718056f + f * (0.240226506959f + f * (0.0555041086648f + f * (0.00961812910763f +
                  f * (0.00133335581464f + f * 0.000154035303934f)))));
    int ni = (int)n;
    if (ni < -125) return 0.f;
    uint32_t qb;
    memcpy(&qb, &q, 4);
    qb += (uint32_t)ni << 23;
    float r;
    memcpy(&r, &qb, 4);
    return r;
}


// ---- exp / pow for the denoisers' weights (denoiser.cu:70-77,139-150; glm::exp / glm::pow = CUDA libdevice in the
// reference).  Fixed binary32 recipes, evaluated identically by the HIP kernels (device/kernels_denoise.h).
inline float exp_det(float x) {
    if (!(x == x)) return x;
    if (x > 88.7f) return INFINITY;
    if (x < -87.3f) return 0.f;
    float n = rintf(x * 1.44269504088896341f);
    float r = (x - n * 0.693359375f) - n * -2.12194440e-4f;
    float p = 1.f + r * (1.f + r * (0.5f + r * (0.166666671633720f + r * (0.0416666679084301f +
                  r * (0.00833333376795053f + r * 0.00138888892252f)))));
    uint32_t pb;
    memcpy(&pb, &p, 4);
    pb += (uint32_t)(int)n << 23;
    float out;
    memcpy(&out, &pb, 4);
    return out;
}
inline float pow_det(float x, float y) {
    if (!(x > 0.f)) return (x == 0.f && y > 0.f) ? 0.f : ((x == 0.f && y == 0.f) ? 1.f : NAN);
    if (x == INFINITY) return x;
    int eAdj = 0;
    if (x < 1.17549435e-38f) {
        x = x * 16777216.f;
        eAdj = -24;
    }
    uint32_t bits;
    memcpy(&bits, &x, 4);
    int e = int((bits >> 23) & 0xffu) - 126 + eAdj;
    uint32_t mb = (bits & 0x007fffffu) | 0x3f000000u;
    float m;
    memcpy(&m, &mb, 4);
    if (m < 0.70710678118654752f) {
        m = m + m;
        e = e - 1;
    }
    float sN = (m - 1.f) / (m + 1.f);
    float z = sN * sN;
    float p = sN + sN * z * (0.333333333333f + z * (0.2f + z * (0.142857142857f + z * 0.111111111111f)));
    float l2 = p * 2.8853900817779268f;
    float t = (float(e) + l2) * y;
    if (t > 127.9f) return INFINITY;
    if (t < -125.9f) return 0.f;
    float n = rintf(t);
    float f = t - n;
    float q = 1.f + f * (0.69314718056f + f * (0.240226506959f + f * (0.0555041086648f + f * (0.00961812910763f +
                  f * (0.00133335581464f + f * 0.000154035303934f)))));
    uint32_t qb;
    memcpy(&qb, &q, 4);
    qb += (uint32_t)(int)n << 23;
    float r;
    memcpy(&r, &qb, 4);
    return r;
}

}  // namespace om
