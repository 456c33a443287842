// ssf_math.hpp -- per-element arithmetic of the supersurfel hot path for gfx950 kernels.
//
// Everything here is `SSF_HD` (host + device) and uses only IEEE-754 +,-,*,/,sqrt, comparisons and
// integer ops, in a fixed written order; the library is built with -ffp-contract=off and
// -fhip-fp32-correctly-rounded-divide-sqrt, so results are bit-reproducible on host and device.
// Reference semantics (what is computed) are cited per function, paths relative to
// /root/reference/core; how it is computed (layout, fusion, exact integer sums) is this build's.
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#define SSF_HD __host__ __device__ __forceinline__
#else
#define SSF_HD inline
#endif

namespace ssf {

struct V3 { float x, y, z; };
struct Sym3 { float xx, xy, xz, yy, yz, zz; };   // symmetric 3x3 (Cov3, matrix_types.h:26-31)
struct M3 { V3 r0, r1, r2; };                    // rows (Mat33, matrix_types.h:33-36)

SSF_HD V3 v3(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
SSF_HD V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
SSF_HD V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
SSF_HD V3 scale(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
SSF_HD V3 negate(V3 a) { return v3(-a.x, -a.y, -a.z); }
SSF_HD float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }      // vector_math.cuh:235
SSF_HD V3 cross3(V3 a, V3 b) {                                                     // vector_math.cuh:117
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
SSF_HD float len3(V3 a) { return sqrtf(dot3(a, a)); }
// normalize with an exact reciprocal square root (the reference's rsqrtf is approximate on CUDA)
SSF_HD V3 unit3(V3 a) { float inv = 1.0f / sqrtf(dot3(a, a)); return v3(a.x * inv, a.y * inv, a.z * inv); }

SSF_HD Sym3 sym3(float xx, float xy, float xz, float yy, float yz, float zz) {
    Sym3 c; c.xx = xx; c.xy = xy; c.xz = xz; c.yy = yy; c.yz = yz; c.zz = zz; return c;
}
SSF_HD Sym3 sym_add(Sym3 a, Sym3 b) { return sym3(a.xx + b.xx, a.xy + b.xy, a.xz + b.xz, a.yy + b.yy, a.yz + b.yz, a.zz + b.zz); }
SSF_HD Sym3 sym_sub(Sym3 a, Sym3 b) { return sym3(a.xx - b.xx, a.xy - b.xy, a.xz - b.xz, a.yy - b.yy, a.yz - b.yz, a.zz - b.zz); }
SSF_HD Sym3 sym_scale(float s, Sym3 a) { return sym3(s * a.xx, s * a.xy, s * a.xz, s * a.yy, s * a.yz, s * a.zz); }
SSF_HD Sym3 sym_div(Sym3 a, float s) { return sym3(a.xx / s, a.xy / s, a.xz / s, a.yy / s, a.yz / s, a.zz / s); }
SSF_HD float sym_trace(Sym3 a) { return (a.xx + a.yy) + a.zz; }
SSF_HD V3 sym_mul(Sym3 m, V3 b) {                                                  // matrix_math.cuh:164
    return v3((m.xx * b.x + m.xy * b.y) + m.xz * b.z, (m.xy * b.x + m.yy * b.y) + m.yz * b.z,
              (m.xz * b.x + m.yz * b.y) + m.zz * b.z);
}
SSF_HD Sym3 sym_square(Sym3 a) {                                                   // matrix_math.cuh:184
    return sym3((a.xx * a.xx + a.xy * a.xy) + a.xz * a.xz, (a.xx * a.xy + a.xy * a.yy) + a.xz * a.yz,
                (a.xx * a.xz + a.xy * a.yz) + a.xz * a.zz, (a.xy * a.xy + a.yy * a.yy) + a.yz * a.yz,
                (a.xy * a.xz + a.yy * a.yz) + a.yz * a.zz, (a.xz * a.xz + a.yz * a.yz) + a.zz * a.zz);
}
SSF_HD Sym3 sym_outer(V3 v) { return sym3(v.x * v.x, v.x * v.y, v.x * v.z, v.y * v.y, v.y * v.z, v.z * v.z); }
// closed-form inverse, |det| > 1e-9 compared in double (matrix_math.cuh:41-63)
SSF_HD bool sym_inverse(Sym3 in, Sym3& out) {
    out.xx = in.zz * in.yy - in.yz * in.yz;
    out.xy = in.xz * in.yz - in.zz * in.xy;
    out.xz = in.xy * in.yz - in.xz * in.yy;
    out.yy = in.zz * in.xx - in.xz * in.xz;
    out.yz = in.xy * in.xz - in.xx * in.yz;
    out.zz = in.xx * in.yy - in.xy * in.xy;
    float det = (in.xx * out.xx + in.xy * out.xy) + in.xz * out.xz;
    if (fabs((double)det) > 1e-9) {
        out.xx /= det; out.xy /= det; out.xz /= det; out.yy /= det; out.yz /= det; out.zz /= det;
        return true;
    }
    return false;
}

SSF_HD M3 m3(V3 a, V3 b, V3 c) { M3 m; m.r0 = a; m.r1 = b; m.r2 = c; return m; }
SSF_HD M3 m3_identity() { return m3(v3(1, 0, 0), v3(0, 1, 0), v3(0, 0, 1)); }
SSF_HD M3 m3_transpose(M3 a) {
    return m3(v3(a.r0.x, a.r1.x, a.r2.x), v3(a.r0.y, a.r1.y, a.r2.y), v3(a.r0.z, a.r1.z, a.r2.z));
}
SSF_HD V3 m3_mulv(M3 a, V3 b) { return v3(dot3(a.r0, b), dot3(a.r1, b), dot3(a.r2, b)); }   // matrix_math.cuh:484
SSF_HD V3 row_mul(V3 a, M3 b) {                                                     // row vector * matrix
    return v3((a.x * b.r0.x + a.y * b.r1.x) + a.z * b.r2.x, (a.x * b.r0.y + a.y * b.r1.y) + a.z * b.r2.y,
              (a.x * b.r0.z + a.y * b.r1.z) + a.z * b.r2.z);
}
SSF_HD M3 m3_mul(M3 a, M3 b) { return m3(row_mul(a.r0, b), row_mul(a.r1, b), row_mul(a.r2, b)); }   // :381
SSF_HD Sym3 rot_sym(M3 A, Sym3 B) {                                                 // mult_ABAt, :442
    V3 b1 = v3(B.xx, B.xy, B.xz), b2 = v3(B.xy, B.yy, B.yz), b3 = v3(B.xz, B.yz, B.zz);
    V3 t0 = v3(dot3(b1, A.r0), dot3(b2, A.r0), dot3(b3, A.r0));
    V3 t1 = v3(dot3(b1, A.r1), dot3(b2, A.r1), dot3(b3, A.r1));
    V3 t2 = v3(dot3(b1, A.r2), dot3(b2, A.r2), dot3(b3, A.r2));
    return sym3(dot3(A.r0, t0), dot3(A.r0, t1), dot3(A.r0, t2), dot3(A.r1, t1), dot3(A.r1, t2), dot3(A.r2, t2));
}

// rotMatToQuat, matrix_math.cuh:529-618
SSF_HD void rot_to_quat(M3 m, float* q) {
    float s; const float tr = (m.r0.x + m.r1.y) + m.r2.z;
    if (tr > 0) {
        s = sqrtf(tr + 1); q[3] = 0.5f * s; s = 0.5f / s;
        q[0] = (m.r2.y - m.r1.z) * s; q[1] = (m.r0.z - m.r2.x) * s; q[2] = (m.r1.x - m.r0.y) * s;
    } else {
        int i = 0;
        if (m.r1.y > m.r0.x) i = 1;
        if (m.r2.z > m.r0.x || m.r2.z > m.r1.y) i = 2;
        if (i == 0) {
            s = sqrtf(((1.0f + m.r0.x) - m.r1.y) - m.r2.z); q[0] = 0.5f * s; s = 0.5f / s;
            q[3] = (m.r2.y - m.r1.z) * s; q[1] = (m.r0.y + m.r1.x) * s; q[2] = (m.r0.z + m.r2.x) * s;
        } else if (i == 1) {
            s = sqrtf(((1.0f + m.r1.y) - m.r0.x) - m.r2.z); q[1] = 0.5f * s; s = 0.5f / s;
            q[3] = (m.r0.z - m.r2.x) * s; q[0] = (m.r0.y + m.r1.x) * s; q[2] = (m.r1.z + m.r2.y) * s;
        } else {
            s = sqrtf(((1.0f + m.r2.z) - m.r0.x) - m.r1.y); q[2] = 0.5f * s; s = 0.5f / s;
            q[3] = (m.r1.x - m.r0.y) * s; q[0] = (m.r0.z + m.r2.x) * s; q[1] = (m.r1.z + m.r2.y) * s;
        }
    }
}
// quatToRotMat, matrix_math.cuh:512-527, with its `wy = q.w*q.z` (sic, :521) reproduced: applyDeformation's rotation
SSF_HD M3 quat_to_rot_quirk(const float* q /* x, y, z, w */) {
    const float x2 = q[0] * q[0], y2 = q[1] * q[1], z2 = q[2] * q[2];
    const float xy = q[0] * q[1], xz = q[0] * q[2], yz = q[1] * q[2];
    const float wx = q[3] * q[0], wy = q[3] * q[2] /* sic */, wz = q[3] * q[2];
    return m3(v3(1.0f - 2.0f * (y2 + z2), 2.0f * (xy - wz), 2.0f * (xz + wy)),
              v3(2.0f * (xy + wz), 1.0f - 2.0f * (x2 + z2), 2.0f * (yz - wx)),
              v3(2.0f * (xz - wy), 2.0f * (yz + wx), 1.0f - 2.0f * (x2 + y2)));
}

// ---- specified roots: IEEE double ops only (see DESIGN.md "arithmetic spec") -----------------
SSF_HD double bits_to_f64(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
SSF_HD uint64_t f64_to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }
// RN(x / 3.0) -- the IEEE quotient, bit for bit -- in three operations instead of the division's eleven: q = RN(x c) with
// c = RN(1/3) is within 1.25 ulp of x / 3, so e = x - 3 q is a multiple (at most 4) of ulp(q) and exact in one fused
// operation; q + e c = x / 3 + e (c - 1/3) differs from the true quotient by less than 2^-53 ulp, while x / 3 cannot lie
// closer than ulp / 6 to a midpoint between two doubles (x is an even multiple of ulp / 2, three times a midpoint an odd
// one): the final rounding returns RN(x / 3).  For normal x away from the ends of the exponent range -- the specified
// cube root works on [0.008, 1.2].  (tools/probe/div3_check.c: 1.5e9 values against the division, no difference.)
// Twelve of these per pixel in k_render_moments, whose time is this conversion's double-precision arithmetic.
SSF_HD double div3_exact(double x) {
    const double c = 0x1.5555555555555p-2;
    const double q = x * c;
    return fma(fma(-3.0, q, x), c, q);
}
// b / 3 for a 64-bit pattern from 32-bit pieces: two multiply-high and a handful of adds instead of the compiler's 64 x 64
// multiply-high (four quarter-rate 64-bit multiply-adds).  hi = 3 qh + r, lo = 3 t + s, 2^32 = 3 * 0x55555555 + 1:
// r 2^32 + lo = 3 (r * 0x55555555 + t) + (r + s).  (tools/probe/div3_check.c: 2e9 patterns.)
SSF_HD uint64_t div3_u64(uint64_t b) {
    const uint32_t hi = (uint32_t)(b >> 32), lo = (uint32_t)b;
    const uint32_t qh = (uint32_t)(((uint64_t)hi * 0xAAAAAAABull) >> 33), t = (uint32_t)(((uint64_t)lo * 0xAAAAAAABull) >> 33);
    uint32_t qh2 = qh + qh, t2 = t + t;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(qh2), "+v"(t2));       // (3 q as q + q + q: left to itself the compiler multiplies by -3 at a quarter of the rate)
#endif
    const uint32_t r = hi - (qh2 + qh), sm = lo - (t2 + t);            // remainders, 0 .. 2
    const uint32_t rc = ((0u - (r & 1u)) & 0x55555555u) | ((0u - (r >> 1)) & 0xAAAAAAAAu);      // r * 0x55555555
    const uint32_t ql = rc + t + ((r + sm) >= 3u ? 1u : 0u);
    return ((uint64_t)qh << 32) | ql;
}
// n / d, bit for bit, for POSITIVE NORMAL operands of moderate exponent (the cube root's: [0.008, 1.2] / [0.04, 1.2]): the
// IEEE division the compiler emits for gfx950 is v_div_scale x 2, v_rcp_f64, two Newton steps on the reciprocal, the
// quotient, one correction (v_div_fmas) and v_div_fixup; for such operands the scaling is the identity, v_div_fmas is a
// plain fused multiply-add and the fix-up returns its input -- the eight operations below are the same operations on the
// same values.  (Host code divides.)
SSF_HD double div_inrange(double n, double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0); r = fma(r, e, r);
    e = fma(-d, r, 1.0); r = fma(r, e, r);
    const double q = n * r;
    return fma(fma(-d, q, n), r, q);
#else
    return n / d;
#endif
}
SSF_HD double cbrt_spec(double a) {          // a > 0 (normal, moderate exponent: see div_inrange)
    double y = bits_to_f64(div3_u64(f64_to_bits(a)) + 0x2A9F7893782DA1CEull);
#pragma unroll
    for (int i = 0; i < 4; i++) y = div3_exact(2.0 * y + div_inrange(a, y * y));
    return y;
}
SSF_HD double root5_spec(double a) {         // a > 0
    double y = bits_to_f64(f64_to_bits(a) / 5 + 0x3325F8C2A7F1C29Aull);
#pragma unroll
    for (int i = 0; i < 5; i++) { double y2 = y * y; y = (4.0 * y + a / (y2 * y2)) / 5.0; }
    return y;
}
SSF_HD float pow24_spec(float x) { double a = (double)x, t = root5_spec(a); return (float)((a * a) * (t * t)); }
SSF_HD float pow_inv24_spec(float x) {
    double t = cbrt_spec(sqrt(sqrt((double)x)));
    double t2 = t * t;
    return (float)((t2 * t2) * t);
}
SSF_HD float cbrtf_spec(float x) { return (float)cbrt_spec((double)x); }

// exp(x) for x <= 0 as a specified sequence of IEEE single-precision operations (libm / OCML / CUDA exp are not
// bit-reproducible across platforms, and this one gates nothing: it only weights the taps of the depth pre-filter):
// x = k ln2 + r with a two-constant Cody-Waite reduction, degree-7 Taylor polynomial in r (|r| <= 0.347: truncation
// 5e-9), exact scaling by 2^k.  ~20 float operations, 1-2 ulp; no fused multiply-add (the build forbids contraction).
// (Round 1 evaluated this in double with eleven divisions per call: the pre-filter spent 0.4 ms per 640x480 frame in it.)
SSF_HD float exp_neg_spec(float x) {
    if (!(x > -87.0f)) return 0.0f;
    if (x > 0.0f) x = 0.0f;
    const float kf = rintf(x * 1.44269502f);                       // k in [-126, 0]
    // Cody-Waite: ln 2 = 0.693359375 (9 significant bits: kf * hi is exact) - 2.12194440e-4
    const float r = (x - kf * 0.693359375f) - kf * -2.12194440e-4f;
    float p = 1.98412701e-4f;                                      // 1/5040
    p = p * r + 1.38888892e-3f;                                    // 1/720
    p = p * r + 8.33333377e-3f;                                    // 1/120
    p = p * r + 4.16666679e-2f;                                    // 1/24
    p = p * r + 0.166666672f;                                      // 1/6
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    const uint32_t bits = (uint32_t)(127 + (int)kf) << 23;         // 2^k exactly (k >= -126: a normal number)
    float two_k; memcpy(&two_k, &bits, 4);
    return p * two_k;
}

// sRGB(0..255) -> CIE Lab, vector_math.cuh:566-585
SSF_HD float srgb_expand(float c) { return (c > 0.04045f) ? pow24_spec((c + 0.055f) / 1.055f) : c / 12.92f; }
SSF_HD float lab_f(float t) { return (t > 0.008856f) ? cbrtf_spec(t) : 7.787f * t + 16.0f / 116.0f; }
// (in pieces, so that a kernel can spread the three channels over lanes: update_group in ssf_track_fuse.hip)
SSF_HD V3 xyz_from_linear(float r, float g, float b) {
    return v3(((r * 0.4124f + g * 0.3575f) + b * 0.1805f) / 0.95047f, ((r * 0.2126f + g * 0.7152f) + b * 0.0722f),
              ((r * 0.0193f + g * 0.1192f) + b * 0.9505f) / 1.08883f);
}
SSF_HD V3 lab_from_f(float x, float y, float z) { return v3(116.0f * y - 16.0f, 500.0f * (x - y), 200.0f * (y - z)); }
SSF_HD V3 rgb_to_lab(V3 c) {
    const V3 q = xyz_from_linear(srgb_expand(c.x / 255.0f), srgb_expand(c.y / 255.0f), srgb_expand(c.z / 255.0f));
    return lab_from_f(lab_f(q.x), lab_f(q.y), lab_f(q.z));
}
// same conversion for 8-bit colours: the gamma expansion of the 256 possible channel values comes
// from a table built on the host with srgb_expand itself (identical bits), the rest is unchanged
SSF_HD V3 rgb8_to_lab(const float* __restrict__ expand_lut, unsigned r8, unsigned g8, unsigned b8) {
    const V3 q = xyz_from_linear(expand_lut[r8], expand_lut[g8], expand_lut[b8]);
    return lab_from_f(lab_f(q.x), lab_f(q.y), lab_f(q.z));
}
// CIE Lab -> sRGB(0..255), vector_math.cuh:543-564 (its two double literals promote g and b)
SSF_HD float lab_finv(float t) { float t3 = (t * t) * t; return (t3 > 0.008856f) ? t3 : (t - 16.0f / 116.0f) / 7.787f; }
SSF_HD float srgb_compress(float c) { return (c > 0.0031308f) ? (1.055f * pow_inv24_spec(c) - 0.055f) : 12.92f * c; }
SSF_HD V3 lab_to_linear_rgb(V3 c) {
    float y = (c.x + 16.0f) / 116.0f;
    float x = c.y / 500.0f + y;
    float z = y - c.z / 200.0f;
    x = 0.95047f * lab_finv(x); y = 1.0f * lab_finv(y); z = 1.08883f * lab_finv(z);
    const float r = (x * 3.2406f - y * 1.5372f) - z * 0.4986f;
    const float g = (float)(((double)(-x * 0.9689f) + (double)y * 1.8758) + (double)(z * 0.0415f));
    const float b = (float)((double)(x * 0.0557f - y * 0.2040f) + (double)z * 1.0570);
    return v3(r, g, b);
}
SSF_HD float srgb_to_255(float lin) { return fmaxf(0.0f, fminf(1.0f, srgb_compress(lin))) * 255.0f; }
SSF_HD V3 lab_to_rgb(V3 c) {
    const V3 l = lab_to_linear_rgb(c);
    return v3(srgb_to_255(l.x), srgb_to_255(l.y), srgb_to_255(l.z));
}

// principal frame by repeated squaring, supersurfel_fusion_kernels.cu:48-111
SSF_HD V3 dominant_column(Sym3 M) {
    float vmax = fmaxf(fmaxf(fmaxf(fmaxf(fmaxf(M.xx, M.xy), M.xz), M.yy), M.yz), M.zz);
    if (M.xx == vmax || M.xy == vmax || M.xz == vmax) return unit3(v3(M.xx, M.xy, M.xz));
    if (M.yy == vmax || M.yz == vmax) return unit3(v3(M.xy, M.yy, M.yz));
    return unit3(v3(M.xz, M.yz, M.zz));
}
SSF_HD float axis_eigenvalue(Sym3 A, V3 v) {
    float emax = fmaxf(fmaxf(v.x, v.y), v.z);
    if (v.x == emax) return ((A.xx * v.x + A.xy * v.y) + A.xz * v.z) / v.x;
    if (v.y == emax) return ((A.xy * v.x + A.yy * v.y) + A.yz * v.z) / v.y;
    return ((A.xz * v.x + A.yz * v.y) + A.zz * v.z) / v.z;
}
// dominant eigenvector of X (trace 1) by ten squarings
SSF_HD V3 principal_power(Sym3 X) {
    for (int i = 0; i < 10; ++i) { X = sym_square(X); X = sym_div(X, sym_trace(X)); }
    return dominant_column(X);
}
SSF_HD Sym3 principal_start(Sym3 A, bool smallest) {       // the two start matrices: A / tr A, and I - A / tr A
    const Sym3 P = sym_div(A, sym_trace(A));
    return smallest ? sym3(1.f - P.xx, -P.xy, -P.xz, 1.f - P.yy, -P.yz, 1.f - P.zz) : P;
}
SSF_HD void principal_finish(Sym3 A, V3 r0, V3 r2, M3& vecs, V3& vals) {
    vecs.r0 = r0; vecs.r2 = r2;
    vecs.r1 = cross3(vecs.r2, vecs.r0);
    vals = v3(axis_eigenvalue(A, vecs.r0), axis_eigenvalue(A, vecs.r1), axis_eigenvalue(A, vecs.r2));
}
SSF_HD void principal_frame(Sym3 A, M3& vecs, V3& vals) {
    principal_finish(A, principal_power(principal_start(A, false)), principal_power(principal_start(A, true)), vecs, vals);
}

// connectivity guard of a relabelling pass, isUnchangeable (TPS_RGBD_kernels.cuh:178-233): walk the ring NW, N, NE, E, SE, S,
// SW, W (not closed W -> NW); the pixel is a bridge -- its label may not change -- when "carries my label" flips more than
// twice along the walk.  `ring`: bit k = ring pixel k carries the pixel's own label; flips = set bits of ring ^ (ring >> 1)
// among the seven adjacent pairs.
SSF_HD unsigned int guard_ring(int index, int nw, int n, int ne, int e, int se, int s, int sw, int w) {
    return (nw == index ? 1u : 0u) | (n == index ? 2u : 0u) | (ne == index ? 4u : 0u) | (e == index ? 8u : 0u) |
           (se == index ? 16u : 0u) | (s == index ? 32u : 0u) | (sw == index ? 64u : 0u) | (w == index ? 128u : 0u);
}
SSF_HD bool guard_unchangeable(unsigned int ring) { return __builtin_popcount((ring ^ (ring >> 1)) & 0x7Fu) > 2; }

// 3x3 plane normal equations, TPS_RGBD_kernels.cu:27-59 (its guard only rejects -inf; kept)
SSF_HD bool plane_solve(float& ta, float& tb, float& tc, float x1, float y1, float z1, float d1,
                        float x2, float y2, float z2, float d2, float x3, float y3, float z3, float d3) {
    const float eps = 1e-20f;
    float denA = (x1 * z2 - x2 * z1) * (y2 * z3 - y3 * z2) - (x2 * z3 - x3 * z2) * (y1 * z2 - y2 * z1);
    if (!isfinite(denA) && denA < eps) return false;
    ta = ((z2 * d1 - z1 * d2) * (y2 * z3 - y3 * z2) - (z3 * d2 - z2 * d3) * (y1 * z2 - y2 * z1)) / denA;
    float denB = y1 * z2 - y2 * z1;
    if (denB > eps) tb = ((z2 * d1 - z1 * d2) - ta * (x1 * z2 - x2 * z1)) / denB;
    else { denB = y2 * z3 - y3 * z2; tb = ((z3 * d2 - z2 * d3) - ta * (x2 * z3 - x3 * z2)) / denB; }
    if (z1 > eps) tc = ((d1 - ta * x1) - tb * y1) / z1;
    else if (z2 > eps) tc = ((d2 - ta * x2) - tb * y2) / z2;
    else tc = ((d3 - ta * x3) - tb * y3) / z3;
    return true;
}

// round half away from zero for |v| < 2^23, -1 when v cannot be a pixel coordinate
SSF_HD int pixel_round(float v) {
    if (!(fabsf(v) < 8388608.0f)) return -1;
    float t = truncf(v);
    if (fabsf(v - t) >= 0.5f) t += (v < 0.0f) ? -1.0f : 1.0f;
    return (int)t;
}

// exact fixed-point terms (order-independent sums)
SSF_HD long long fx64(double v, double scale, double lim) {
#if defined(__HIP_DEVICE_COMPILE__)
    // The same value as the host branch below, in fewer instructions (twelve of these per pixel in k_render_moments):
    // clamp by min / max (a NaN is replaced at the end), and the round-to-nearest-even conversion of |t| < 2^51 by the
    // magic-number addition -- t + 1.5 * 2^52 is rounded to an integer by the addition itself (default rounding mode) and lies
    // in the binade whose unit is 1, so the integer is the difference of the bit patterns; there is no hardware
    // double -> int64 conversion, the general sequence is ~8 double-precision instructions.  With a compile-time `lim`
    // below 2^50 the general sequence is not even emitted.
    const double t0 = v * scale;
    const double t = __builtin_fmin(__builtin_fmax(t0, -lim), lim);
    long long r;
    if (lim <= 1125899906842624.0 || __builtin_fabs(t) < 2251799813685248.0)
        r = __double_as_longlong(t + 6755399441055744.0) - 0x4338000000000000LL;
    else r = __double2ll_rn(t);
    return t0 == t0 ? r : 0;
#else
    double t = v * scale;
    if (!(t == t)) return 0;
    if (t > lim) t = lim;
    if (t < -lim) t = -lim;
    return llrint(t);
#endif
}
// 32-bit fixed point of an already scaled value: round to nearest even, then convert SATURATING exactly as the hardware's
// v_cvt_i32_f32 does (ISA manual; checked on the part by tools/probe/cvt_i32_f32.hip): NaN -> 0, t >= 2^31 -> INT_MAX,
// t <= -2^31 -> INT_MIN, anything else is exact -- two instructions per term, no compare / select (27 terms per row in k_icp).
// (Until round 3 the specification also sent 2^31 - 128, the largest float below 2^31, to INT_MAX: one compare and one
// select per term for a value no term can take; oracle and product changed together.)
SSF_HD int fx32r(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float t = rintf(v);
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(t));
    return r;
#else
    float t = rintf(v);
    if (!(t == t)) return 0;
    if (t >= 2147483648.0f) return 2147483647;
    if (t <= -2147483648.0f) return (int)0x80000000;
    return (int)t;
#endif
}
SSF_HD int fx32(float v, float scale) { return fx32r(v * scale); }
#define SSF_DISP_SCALE 1073741824.0            /* 2^30 */
#define SSF_DISP_LIM 4503599627370496.0        /* 2^52 */
#define SSF_MOM_SCALE 16777216.0               /* 2^24 */
#define SSF_MOM_LIM 1099511627776.0            /* 2^40 */

// counter-based generator: splitmix64 finaliser over (seed, stream, counter)
SSF_HD uint32_t rng_draw(uint64_t seed, uint32_t stream, uint32_t& counter) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((((uint64_t)stream) << 32) | (uint64_t)counter);
    counter++;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}
SSF_HD float rng_unit(uint32_t r) { return (float)((r >> 8) + 1u) * 5.9604644775390625e-8f; }   // (0,1]

// superpixel table row: 12 floats, 48 B, 16-byte aligned (three dwordx4 gathers)
struct __attribute__((aligned(16))) SpRow { float cx, cy, r, g, b, ta, tb, tc, size, pad0, pad1, pad2; };

}  // namespace ssf
