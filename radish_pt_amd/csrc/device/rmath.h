// radish_pt_amd/csrc/device/rmath.h — device-side scalar/vector arithmetic for the gfx950 kernels.
//
// Numerics contract: the whole library is compiled with -ffp-contract=off and without fast-math,
// so every +,-,*,/ and sqrt here is one correctly-rounded binary32 operation (HIP's default
// -fhip-fp32-correctly-rounded-divide-sqrt), evaluated in the order written.  The operation order of each
// helper follows glm's generic code path, which is what the reference's expressions expand to
// (/root/reference/src/mathUtil.h, material.h, … include glm for all vector math).  No libm/ocml transcendental
// is called on the device: sin/cos go through sincos_det below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RD_DEV __device__ __forceinline__

namespace rd {

struct v2 {
    float x, y;
};
struct v3 {
    float x, y, z;
};
struct v4 {
    float x, y, z, w;
};

RD_DEV v2 mk2(float x, float y) { return v2{x, y}; }
RD_DEV v3 mk3(float x, float y, float z) { return v3{x, y, z}; }
RD_DEV v3 mk3(float s) { return v3{s, s, s}; }

RD_DEV v2 operator+(v2 a, v2 b) { return {a.x + b.x, a.y + b.y}; }
RD_DEV v2 operator-(v2 a, v2 b) { return {a.x - b.x, a.y - b.y}; }
RD_DEV v2 operator*(v2 a, v2 b) { return {a.x * b.x, a.y * b.y}; }
RD_DEV v2 operator*(v2 a, float s) { return {a.x * s, a.y * s}; }
RD_DEV float dot(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }

RD_DEV v3 operator+(v3 a, v3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RD_DEV v3 operator-(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
RD_DEV v3 operator*(v3 a, v3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
RD_DEV v3 operator/(v3 a, v3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
RD_DEV v3 operator*(v3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
RD_DEV v3 operator/(v3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
RD_DEV v3 operator+(v3 a, float s) { return {a.x + s, a.y + s, a.z + s}; }
RD_DEV v3 operator-(v3 a) { return {-a.x, -a.y, -a.z}; }
RD_DEV v3 rdiv(float s, v3 a) { return {s / a.x, s / a.y, s / a.z}; }  // glm: scalar / vec

// glm::dot(vec3): (a.x*b.x + a.y*b.y) + a.z*b.z
RD_DEV float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// glm::cross
RD_DEV v3 cross(v3 a, v3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
RD_DEV float rsqrt_exact(float x) { return 1.f / __builtin_sqrtf(x); }  // glm::inversesqrt
RD_DEV float length(v3 v) { return __builtin_sqrtf(dot(v, v)); }
RD_DEV v3 normalize(v3 v) { return v * rsqrt_exact(dot(v, v)); }
RD_DEV v3 reflect(v3 I, v3 N) { return I - N * dot(N, I) * 2.f; }
RD_DEV float mixf(float x, float y, float a) { return x * (1.f - a) + y * a; }
RD_DEV v3 mix(v3 x, v3 y, float a) { return x * (1.f - a) + y * a; }
RD_DEV v3 mix(v3 x, v3 y, v3 a) { return x * (mk3(1.f) - a) + y * a; }
// glm::min(x,y) = (y < x) ? y : x ; glm::max(x,y) = (x < y) ? y : x  (NaN-order sensitive: keep as selects)
RD_DEV float gmin(float x, float y) { return (y < x) ? y : x; }
RD_DEV float gmax(float x, float y) { return (x < y) ? y : x; }
RD_DEV int imin(int x, int y) { return (y < x) ? y : x; }
RD_DEV v3 gmin(v3 a, v3 b) { return {gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)}; }
RD_DEV v3 gmax(v3 a, v3 b) { return {gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)}; }
RD_DEV float fabs_(float x) { return __builtin_fabsf(x); }
// C fminf/fmaxf semantics (a NaN operand yields the other one), as AABB::getDistMinMax/MaxMin use them.
RD_DEV float c_fminf(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    return (b < a) ? b : a;
}
RD_DEV float c_fmaxf(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    return (a < b) ? b : a;
}
RD_DEV bool isnan_(float x) { return x != x; }
RD_DEV bool isinf_(float x) { return __builtin_fabsf(x) == __builtin_inff(); }
RD_DEV bool isNanOrInf(float x) { return isnan_(x) || isinf_(x); }

struct m3 {  // column-major, as glm::mat3
    v3 c0, c1, c2;
};
RD_DEV v3 mul(const m3 &m, v3 v) {  // glm mat3 * vec3
    return {(m.c0.x * v.x + m.c1.x * v.y) + m.c2.x * v.z, (m.c0.y * v.x + m.c1.y * v.y) + m.c2.y * v.z,
            (m.c0.z * v.x + m.c1.z * v.y) + m.c2.z * v.z};
}
RD_DEV m3 inverse(const m3 &m) {  // glm::inverse(mat3): cofactors * 1/det
    float m00 = m.c0.x, m01 = m.c0.y, m02 = m.c0.z;
    float m10 = m.c1.x, m11 = m.c1.y, m12 = m.c1.z;
    float m20 = m.c2.x, m21 = m.c2.y, m22 = m.c2.z;
    float ood = 1.f / ((m00 * (m11 * m22 - m21 * m12) - m10 * (m01 * m22 - m21 * m02)) + m20 * (m01 * m12 - m11 * m02));
    m3 r;
    r.c0.x = +(m11 * m22 - m21 * m12) * ood;
    r.c1.x = -(m10 * m22 - m20 * m12) * ood;
    r.c2.x = +(m10 * m21 - m20 * m11) * ood;
    r.c0.y = -(m01 * m22 - m21 * m02) * ood;
    r.c1.y = +(m00 * m22 - m20 * m02) * ood;
    r.c2.y = -(m00 * m21 - m20 * m01) * ood;
    r.c0.z = +(m01 * m12 - m11 * m02) * ood;
    r.c1.z = -(m00 * m12 - m10 * m02) * ood;
    r.c2.z = +(m00 * m11 - m10 * m01) * ood;
    return r;
}

// sin/cos of x in roughly [0, 2*pi]: Cephes-style — 3-term Cody–Waite reduction by pi/2, degree-7/8 minimax
// polynomials, only binary32 mul/add/sub in this order (≈1 ulp).  Replaces libdevice cosf/sinf in
// Math::concentricSampleDisk (/root/reference/src/mathUtil.h:132-136).
RD_DEV void sincos_det(float x, float &s, float &c) {
    float kf = __builtin_rintf(x * 0.63661977236758134f);
    int k = (int)kf;
    float r = ((x - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188216e-8f;
    float z = r * r;
    float sp = r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
    float cp = (1.f - z * 0.5f) + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    int q = k & 3;
    float ss = (q & 1) ? cp : sp;
    float cc = (q & 1) ? sp : cp;
    s = (q & 2) ? -ss : ss;
    c = (q == 1 || q == 2) ? -cc : cc;
}

// atan2 for Math::toPlane (/root/reference/src/mathUtil.h:143-147, CUDA atan2f in the reference): Cephes atanf — range
// reduction at tan(pi/8) and tan(3pi/8), degree-9 odd polynomial — and its quadrant logic; binary32 operations in this
// order only (≈2 ulp).
RD_DEV float atan_det(float xx) {
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
RD_DEV float atan2_det(float y, float x) {
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
RD_DEV float fract_(float x) { return x - __builtin_floorf(x); }  // glm::fract

RD_DEV uint32_t utilhash(uint32_t a) {  // /root/reference/src/mathUtil.h:199-207
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

}  // namespace rd
