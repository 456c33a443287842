// oracle_math.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Plain C++ restatement of the helper arithmetic of the reference's hot path.  Every function
// cites the reference lines it follows (paths relative to /root/reference/core).  fp32 unless the
// reference itself uses double.  Compiled with -ffp-contract=off and no fast-math, so every
// + - * / sqrt below is one correctly rounded IEEE-754 operation in the written order; the HIP
// product is built under the same rules, which is what makes bit-exact parity possible.
//
// Deliberate, documented deviations from the (fast-math, racy, unrunnable-here) CUDA reference:
//   * normalize(): 1/sqrtf(dot) instead of rsqrtf (approximate on CUDA) -- vector_math.cuh:247-252
//   * powf/cbrtf in the Lab conversions are replaced by a specified sequence of IEEE double ops
//     (Newton roots), because libm/OCML/CUDA transcendentals are not bit-reproducible across
//     platforms and the Lab distance gates integer decisions (match / ICP inliers).
//   * pow(x, 3) is x*x*x.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace orc {

// ---- STUDY ARMS (round 6; tools/cuda_tolerance_study.py, tests/test_replay.py) -------------------------------------------------
// The reference is compiled --use_fast_math (cmake/UseCUDA.cmake:15: approximate division / sqrt / rsqrt / powf / cbrtf, FMA
// contraction, flush-to-zero) and is racy by construction: "the CUDA path" is a FAMILY of executions, none of
// which can be produced here.  To state how far that family lies from this build's specification, a second build of these sources
// (-DSSF_ORACLE_ARMS -> oracle/_build/libssf_oracle_arms.so; and the same with -ffp-contract=fast -mfma -> ..._fma.so) routes every
// device-side float division, square root and transcendental through the hooks below, and the schedule decisions of Appendix A
// through g_arms; ssf_oracle_set_arm() (oracle_abi.cpp) switches them at run time.  In the DEFAULT build (the checker every parity
// test uses) the hooks are the plain IEEE operation and g_arms does not exist: fdiv(a, b) IS a / b.
#ifdef SSF_ORACLE_ARMS
struct StudyArms {
    int div_ulp;        // every device-side float division / sqrt result moved by this many ulp (any small integer; 3 = pseudo-random in [-2, 2])
    int pow_ulp;        // the same for the results of powf / cbrtf (the Lab conversions)
    int schedule;       // relabelling pass: 0 = every tile loads before any tile stores (the specification);
                        //                   1 = the reference's 32 x 32 tiles one after the other in raster order, in place
                        //                       ("every block loads after its predecessors stored": the other extreme valid CUDA outcome)
    int tie;            // association arg-min: 0 = ties to the lowest model id, 1 = to the highest
    int insert_rev;     // insertion: 0 = ascending frame id, 1 = descending (another atomic arrival order)
    int filter_gs;      // plane filter: 0 = Jacobi sweeps (the specification), 1 = in place in node order (Gauss-Seidel: the in-place outcome)
};
extern StudyArms g_arms;
static inline float nudge_ulp(float v, int mode, uint32_t key) {
    if (mode == 0 || !(std::fabs(v) > 1.0e-37f) || !(std::fabs(v) < 3.0e38f)) return v;      // (zero, denormal, inf, NaN: untouched)
    const int k = mode == 3 ? (int)(((key * 2654435761u) >> 16) % 5u) - 2 : mode;
    uint32_t b; std::memcpy(&b, &v, 4);
    b = (uint32_t)((int32_t)b + (v > 0.f ? k : -k) * 1);      // (sign-magnitude: + k ulp = towards larger magnitude for both signs)
    std::memcpy(&v, &b, 4);
    return v;
}
static inline uint32_t fbits(float v) { uint32_t b; std::memcpy(&b, &v, 4); return b; }
static inline float fdiv(float a, float b) { return nudge_ulp(a / b, g_arms.div_ulp, fbits(a) ^ (fbits(b) * 0x9E3779B9u)); }
static inline float fsqrt(float a) { return nudge_ulp(sqrtf(a), g_arms.div_ulp, fbits(a)); }
static inline float fpow_hook(float r, float x) { return nudge_ulp(r, g_arms.pow_ulp, fbits(x)); }
#else
static inline float fdiv(float a, float b) { return a / b; }
static inline float fsqrt(float a) { return sqrtf(a); }
static inline float fpow_hook(float r, float) { return r; }
#endif

struct f3 { float x, y, z; };
struct Cov3 { float xx, xy, xz, yy, yz, zz; };          // matrix_types.h:26-31
struct Mat33 { f3 r[3]; };                               // matrix_types.h:33-36 (rows)

static inline f3 mk3(float x, float y, float z) { f3 v{x, y, z}; return v; }
static inline f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
static inline f3 neg(f3 a) { return mk3(-a.x, -a.y, -a.z); }
// vector_math.cuh:235-238
static inline float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// vector_math.cuh:117-120
static inline f3 cross(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// vector_math.cuh:241-244
static inline float length(f3 v) { return fsqrt(dot(v, v)); }
// vector_math.cuh:247-252 (rsqrtf -> exact 1/sqrt, see header)
// TEST HOOK (oracle only; tests/test_oracle.py::test_a_2_ulp_reciprocal_square_root_flips_no_integer_decision): CUDA's rsqrtf is an
// approximation within 2 ulp that cannot be observed here; g_oracle_rsqrt_ulp != 0 moves every reciprocal square root of a
// normalize() by that many ulp (-2 .. 2), 3 = by a pseudo-random number of ulp in [-2, 2] keyed by the operand's bits -- so that
// the integer decisions downstream (the 0.8 normal gates of ICP / association, counts, iteration numbers) can be shown not to
// hang on the stand-in.  0 (the default) is this build's specification: the correctly rounded 1 / sqrt.
extern int g_oracle_rsqrt_ulp;
static inline float oracle_rsqrt(float x) {
    float inv = fdiv(1.0f, fsqrt(x));
    const int mode = g_oracle_rsqrt_ulp;
    if (mode != 0 && inv > 0.f && inv < 3.0e38f) {
        uint32_t bits; memcpy(&bits, &x, 4);
        const int k = mode == 3 ? (int)((bits * 2654435761u) >> 16) % 5 - 2 : mode;
        uint32_t ib; memcpy(&ib, &inv, 4);
        ib = (uint32_t)((int32_t)ib + k);              // (positive finite floats: consecutive bit patterns are consecutive values)
        memcpy(&inv, &ib, 4);
    }
    return inv;
}
static inline f3 normalize(f3 v) { float inv = oracle_rsqrt(dot(v, v)); return v * inv; }

// ---- Cov3 / Mat33 (matrix_math.cuh) --------------------------------------------------------
static inline Cov3 mkcov(float xx, float xy, float xz, float yy, float yz, float zz) {
    Cov3 c{xx, xy, xz, yy, yz, zz}; return c;
}
static inline Cov3 operator+(const Cov3& a, const Cov3& b) {       // :65-75
    return mkcov(a.xx + b.xx, a.xy + b.xy, a.xz + b.xz, a.yy + b.yy, a.yz + b.yz, a.zz + b.zz);
}
static inline Cov3 operator-(const Cov3& a, const Cov3& b) {       // :77-87
    return mkcov(a.xx - b.xx, a.xy - b.xy, a.xz - b.xz, a.yy - b.yy, a.yz - b.yz, a.zz - b.zz);
}
static inline Cov3 operator*(float b, const Cov3& a) {             // :120-130
    return mkcov(b * a.xx, b * a.xy, b * a.xz, b * a.yy, b * a.yz, b * a.zz);
}
static inline Cov3 operator/(const Cov3& a, float b) {             // :132-142
    return mkcov(fdiv(a.xx, b), fdiv(a.xy, b), fdiv(a.xz, b), fdiv(a.yy, b), fdiv(a.yz, b), fdiv(a.zz, b));
}
static inline f3 operator*(const Cov3& m, f3 b) {                  // :164-169
    return mk3((m.xx * b.x + m.xy * b.y) + m.xz * b.z,
               (m.xy * b.x + m.yy * b.y) + m.yz * b.z,
               (m.xz * b.x + m.yz * b.y) + m.zz * b.z);
}
static inline Cov3 square(const Cov3& a) {                         // :184-194
    return mkcov((a.xx * a.xx + a.xy * a.xy) + a.xz * a.xz,
                 (a.xx * a.xy + a.xy * a.yy) + a.xz * a.yz,
                 (a.xx * a.xz + a.xy * a.yz) + a.xz * a.zz,
                 (a.xy * a.xy + a.yy * a.yy) + a.yz * a.yz,
                 (a.xy * a.xz + a.yy * a.yz) + a.yz * a.zz,
                 (a.xz * a.xz + a.yz * a.yz) + a.zz * a.zz);
}
static inline Cov3 outer(f3 v) {                                   // :212-222
    return mkcov(v.x * v.x, v.x * v.y, v.x * v.z, v.y * v.y, v.y * v.z, v.z * v.z);
}
static inline float trace(const Cov3& a) { return (a.xx + a.yy) + a.zz; }   // :236-239
// :41-63.  det threshold is compared in double (1e-9 is a double literal there).
static inline bool inverse(const Cov3& in, Cov3& out) {
    out.xx = in.zz * in.yy - in.yz * in.yz;
    out.xy = in.xz * in.yz - in.zz * in.xy;
    out.xz = in.xy * in.yz - in.xz * in.yy;
    out.yy = in.zz * in.xx - in.xz * in.xz;
    out.yz = in.xy * in.xz - in.xx * in.yz;
    out.zz = in.xx * in.yy - in.xy * in.xy;
    float det = (in.xx * out.xx + in.xy * out.xy) + in.xz * out.xz;
    if (std::fabs((double)det) > 1e-9) {
        out.xx = fdiv(out.xx, det); out.xy = fdiv(out.xy, det); out.xz = fdiv(out.xz, det); out.yy = fdiv(out.yy, det); out.yz = fdiv(out.yz, det); out.zz = fdiv(out.zz, det);
        return true;
    }
    return false;
}
static inline Mat33 transpose(const Mat33& a) {                    // :498-505
    Mat33 t;
    t.r[0] = mk3(a.r[0].x, a.r[1].x, a.r[2].x);
    t.r[1] = mk3(a.r[0].y, a.r[1].y, a.r[2].y);
    t.r[2] = mk3(a.r[0].z, a.r[1].z, a.r[2].z);
    return t;
}
static inline f3 operator*(const Mat33& a, f3 b) {                 // :484-489
    return mk3(dot(a.r[0], b), dot(a.r[1], b), dot(a.r[2], b));
}
static inline Mat33 operator*(const Mat33& a, const Mat33& b) {    // :381-403
    Mat33 c;
    for (int i = 0; i < 3; i++)
        c.r[i] = mk3((a.r[i].x * b.r[0].x + a.r[i].y * b.r[1].x) + a.r[i].z * b.r[2].x,
                     (a.r[i].x * b.r[0].y + a.r[i].y * b.r[1].y) + a.r[i].z * b.r[2].y,
                     (a.r[i].x * b.r[0].z + a.r[i].y * b.r[1].z) + a.r[i].z * b.r[2].z);
    return c;
}
static inline Cov3 mult_ABAt(const Mat33& A, const Cov3& B) {      // :442-459
    f3 r1 = mk3(B.xx, B.xy, B.xz), r2 = mk3(B.xy, B.yy, B.yz), r3 = mk3(B.xz, B.yz, B.zz);
    f3 t0 = mk3(dot(r1, A.r[0]), dot(r2, A.r[0]), dot(r3, A.r[0]));
    f3 t1 = mk3(dot(r1, A.r[1]), dot(r2, A.r[1]), dot(r3, A.r[1]));
    f3 t2 = mk3(dot(r1, A.r[2]), dot(r2, A.r[2]), dot(r3, A.r[2]));
    return mkcov(dot(A.r[0], t0), dot(A.r[0], t1), dot(A.r[0], t2),
                 dot(A.r[1], t1), dot(A.r[1], t2), dot(A.r[2], t2));
}
static inline Mat33 identity33() {
    Mat33 m; m.r[0] = mk3(1, 0, 0); m.r[1] = mk3(0, 1, 0); m.r[2] = mk3(0, 0, 1); return m;
}

// ---- specified roots (double, IEEE ops only) --------------------------------------------------
// cbrt(a), a > 0: bit-level seed (exponent/3, <3.2% error) + 4 Newton steps y <- (2y + a/y^2)/3
// (quadratic convergence: 3e-2 -> 1e-3 -> 1e-6 -> 1e-12 -> double round-off).
static inline double spec_cbrt(double a) {
    uint64_t b; std::memcpy(&b, &a, 8);
    b = b / 3 + 0x2A9F7893782DA1CEull;          // classic exponent-third seed, <6% error
    double y; std::memcpy(&y, &b, 8);
    for (int i = 0; i < 4; i++) y = (2.0 * y + a / (y * y)) / 3.0;
    return y;
}
// a^(1/5), a > 0: seed (exponent/5, <4.1% error) + 5 Newton steps y <- (4y + a/y^4)/5.
static inline double spec_root5(double a) {
    uint64_t b; std::memcpy(&b, &a, 8);
    b = b / 5 + 0x3325F8C2A7F1C29Aull;          // 4/5 * bits(1.0) biased seed
    double y; std::memcpy(&y, &b, 8);
    for (int i = 0; i < 5; i++) { double y2 = y * y; y = (4.0 * y + a / (y2 * y2)) / 5.0; }
    return y;
}
// x^2.4 = x^2 * (x^(1/5))^2 ; x^(1/2.4) = x^(5/12) = (cbrt(sqrt(sqrt(x))))^5, x > 0.
static inline float spec_pow24(float x) {
    double a = (double)x, t = spec_root5(a);
    return fpow_hook((float)((a * a) * (t * t)), x);
}
static inline float spec_pow_inv24(float x) {
    double t = spec_cbrt(std::sqrt(std::sqrt((double)x)));
    double t2 = t * t;
    return fpow_hook((float)((t2 * t2) * t), x);
}
static inline float spec_cbrtf(float x) { return fpow_hook((float)spec_cbrt((double)x), x); }

// exp(x) for x <= 0 as a specified sequence of IEEE single-precision operations (libm / OCML / CUDA exp are not
// bit-reproducible across platforms, and this one gates nothing: it only weights the taps of the depth pre-filter):
// x = k ln2 + r with a two-constant Cody-Waite reduction, degree-7 Taylor polynomial in r (|r| <= 0.347: truncation
// 5e-9), exact scaling by 2^k.  ~20 float operations, 1-2 ulp; no fused multiply-add (the build forbids contraction).
// (Round 1 evaluated this in double with eleven divisions per call: the pre-filter spent 0.4 ms per 640x480 frame in it.)
static inline float spec_exp_neg(float x) {
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
    float two_k; std::memcpy(&two_k, &bits, 4);
    return p * two_k;
}

// vector_math.cuh:566-585
static inline f3 rgbToLab(f3 c) {
    float r = fdiv(c.x, 255.0f), g = fdiv(c.y, 255.0f), b = fdiv(c.z, 255.0f);
    r = (r > 0.04045f) ? spec_pow24(fdiv(r + 0.055f, 1.055f)) : fdiv(r, 12.92f);
    g = (g > 0.04045f) ? spec_pow24(fdiv(g + 0.055f, 1.055f)) : fdiv(g, 12.92f);
    b = (b > 0.04045f) ? spec_pow24(fdiv(b + 0.055f, 1.055f)) : fdiv(b, 12.92f);
    float x = fdiv((r * 0.4124f + g * 0.3575f) + b * 0.1805f, 0.95047f);
    float y = ((r * 0.2126f + g * 0.7152f) + b * 0.0722f);
    float z = fdiv((r * 0.0193f + g * 0.1192f) + b * 0.9505f, 1.08883f);
    x = (x > 0.008856f) ? spec_cbrtf(x) : 7.787f * x + 16.0f / 116.0f;
    y = (y > 0.008856f) ? spec_cbrtf(y) : 7.787f * y + 16.0f / 116.0f;
    z = (z > 0.008856f) ? spec_cbrtf(z) : 7.787f * z + 16.0f / 116.0f;
    return mk3(116.0f * y - 16.0f, 500.0f * (x - y), 200.0f * (y - z));
}
// vector_math.cuh:543-564.  The two double literals (1.8758, 1.0570) promote those sums to
// double exactly as the C++ expression does.
static inline f3 labToRgb(f3 c) {
    float y = fdiv(c.x + 16.0f, 116.0f);
    float x = fdiv(c.y, 500.0f) + y;
    float z = y - fdiv(c.z, 200.0f);
    float x3 = (x * x) * x, y3 = (y * y) * y, z3 = (z * z) * z;
    x = 0.95047f * ((x3 > 0.008856f) ? x3 : fdiv(x - 16.0f / 116.0f, 7.787f));
    y = 1.0f * ((y3 > 0.008856f) ? y3 : fdiv(y - 16.0f / 116.0f, 7.787f));
    z = 1.08883f * ((z3 > 0.008856f) ? z3 : fdiv(z - 16.0f / 116.0f, 7.787f));
    float r = (x * 3.2406f - y * 1.5372f) - z * 0.4986f;
    float g = (float)(((double)(-x * 0.9689f) + (double)y * 1.8758) + (double)(z * 0.0415f));
    float b = (float)((double)(x * 0.0557f - y * 0.2040f) + (double)z * 1.0570);
    r = (r > 0.0031308f) ? (1.055f * spec_pow_inv24(r) - 0.055f) : 12.92f * r;
    g = (g > 0.0031308f) ? (1.055f * spec_pow_inv24(g) - 0.055f) : 12.92f * g;
    b = (b > 0.0031308f) ? (1.055f * spec_pow_inv24(b) - 0.055f) : 12.92f * b;
    return mk3(fmaxf(0.0f, fminf(1.0f, r)) * 255.0f, fmaxf(0.0f, fminf(1.0f, g)) * 255.0f,
               fmaxf(0.0f, fminf(1.0f, b)) * 255.0f);
}

// supersurfel_fusion_kernels.cu:48-111 (power iteration by repeated squaring; reproduced as is)
static inline f3 pick_axis(const Cov3& M) {
    float vmax = fmaxf(fmaxf(fmaxf(fmaxf(fmaxf(M.xx, M.xy), M.xz), M.yy), M.yz), M.zz);
    if (M.xx == vmax || M.xy == vmax || M.xz == vmax) return normalize(mk3(M.xx, M.xy, M.xz));
    if (M.yy == vmax || M.yz == vmax) return normalize(mk3(M.xy, M.yy, M.yz));
    return normalize(mk3(M.xz, M.yz, M.zz));
}
static inline float rayleigh(const Cov3& A, f3 v) {
    float emax = fmaxf(fmaxf(v.x, v.y), v.z);
    if (v.x == emax) return fdiv((A.xx * v.x + A.xy * v.y) + A.xz * v.z, v.x);
    if (v.y == emax) return fdiv((A.xy * v.x + A.yy * v.y) + A.yz * v.z, v.y);
    return fdiv((A.xz * v.x + A.yz * v.y) + A.zz * v.z, v.z);
}
static inline void eigenDecomposition(const Cov3& A, Mat33& vecs, f3& vals, int n) {
    Cov3 Ai = A / trace(A);
    Cov3 Bi = mkcov(1.f - Ai.xx, -Ai.xy, -Ai.xz, 1.f - Ai.yy, -Ai.yz, 1.f - Ai.zz);
    for (int i = 0; i < n; ++i) {
        Ai = square(Ai); Ai = Ai / trace(Ai);
        Bi = square(Bi); Bi = Bi / trace(Bi);
    }
    vecs.r[0] = pick_axis(Ai);
    vecs.r[2] = pick_axis(Bi);
    vecs.r[1] = cross(vecs.r[2], vecs.r[0]);
    vals.x = rayleigh(A, vecs.r[0]);
    vals.y = rayleigh(A, vecs.r[1]);
    vals.z = rayleigh(A, vecs.r[2]);
}

// TPS_RGBD_kernels.cu:27-59 (guard reproduced: it only rejects -inf)
static inline bool solvePlaneEquations(float& tx, float& ty, float& tz,
                                       float x1, float y1, float z1, float d1,
                                       float x2, float y2, float z2, float d2,
                                       float x3, float y3, float z3, float d3) {
    const float eps = 1e-20f;
    float denA = (x1 * z2 - x2 * z1) * (y2 * z3 - y3 * z2) - (x2 * z3 - x3 * z2) * (y1 * z2 - y2 * z1);
    if (!std::isfinite(denA) && denA < eps) return false;
    tx = fdiv((z2 * d1 - z1 * d2) * (y2 * z3 - y3 * z2) - (z3 * d2 - z2 * d3) * (y1 * z2 - y2 * z1), denA);
    float denB = y1 * z2 - y2 * z1;
    if (denB > eps) {
        ty = fdiv((z2 * d1 - z1 * d2) - tx * (x1 * z2 - x2 * z1), denB);
    } else {
        denB = y2 * z3 - y3 * z2;
        ty = fdiv((z3 * d2 - z2 * d3) - tx * (x2 * z3 - x3 * z2), denB);
    }
    if (z1 > eps)      tz = fdiv((d1 - tx * x1) - ty * y1, z1);
    else if (z2 > eps) tz = fdiv((d2 - tx * x2) - ty * y2, z2);
    else               tz = fdiv((d3 - tx * x3) - ty * y3, z3);
    return true;
}

// lroundf restated with exact ops (round half away from zero); callers guarantee |v| < 2^23.
static inline int round_half_away(float v) {
    float t = truncf(v);
    float d = fabsf(v - t);
    if (d >= 0.5f) t += (v < 0.0f) ? -1.0f : 1.0f;
    return (int)t;
}

// Exact fixed-point accumulation (order-independent replacement for the reference's float
// atomics).  q = rint(v * 2^S) clamped to +-2^lim.
static inline int64_t fx_quant(double v, double scale, double lim) {
    double t = v * scale;
    if (!(t == t)) return 0;                 // NaN -> 0
    if (t > lim) t = lim;
    if (t < -lim) t = -lim;
    return (int64_t)std::llrint(t);          // round-half-even (default rounding mode)
}
// 32-bit terms: rint, then a conversion that saturates like the hardware's v_cvt_i32_f32 (NaN -> 0, >= 2^31 -> INT_MAX,
// <= -2^31 -> INT_MIN)
static inline int32_t fx_quant32r(float v) {
    float t = rintf(v);
    if (!(t == t)) return 0;
    if (t >= 2147483648.0f) return 2147483647;
    if (t <= -2147483648.0f) return (int32_t)0x80000000;
    return (int32_t)t;
}
static inline int32_t fx_quant32(float v, float scale) { return fx_quant32r(v * scale); }

// counter-based RNG (replaces cuRAND XORWOW, whose CUDA sequence is not observable here).
static inline uint32_t rng_u32(uint64_t seed, uint32_t stream, uint32_t& counter) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((((uint64_t)stream) << 32) | (uint64_t)counter);
    counter++;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}
// (0,1] like curand_uniform
static inline float rng_uniform(uint32_t r) { return (float)((r >> 8) + 1u) * 5.9604644775390625e-8f; }

}  // namespace orc
