// ref_decision_vectors.cpp -- TEST INFRASTRUCTURE.  Built ONLY where /root/reference exists, into
// oracle/_ref/decision_vectors (see oracle/Makefile).  Pins the three DECISION helpers that gate integer results of the
// extract / fuse stages to the reference's own text:
//   isUnchangeable       core/include/supersurfel_fusion/TPS_RGBD_kernels.cuh:178-233   (connectivity guard of a relabelling pass)
//   solvePlaneEquations  core/src/TPS_RGBD_kernels.cu:27-59                             (RANSAC / plane fit 3x3 solve)
//   eigenDecomposition   core/src/supersurfel_fusion_kernels.cu:48-111                  (principal frame of a supersurfel)
// They are self-contained __device__ functions of plain C arithmetic inside .cu / .cuh files that cannot be included whole
// (kernels, textures, curand).  oracle/Makefile therefore cuts exactly those line ranges out of the reference's files AT BUILD
// TIME into oracle/_ref/decision_*.inc (git-ignored, deleted again right after this file is compiled: never committed, never
// shipped -- the recipe checks that each range still
// starts at the function's signature) and this file #includes them: what is evaluated below is the reference's text, compiled
// by g++ against NVIDIA's CUDA runtime headers from the image (as ref_math_vectors.cpp) with -ffp-contract=off.
//
// TWO BINARIES from this file (round 6), so that the one stand-in of the parity chain touches only what needs it:
//   _ref/decision_vectors        (default)               isUnchangeable + solvePlaneEquations.  NO stand-in: rsqrtf is only DECLARED
//                                                        (the reference's normalize() mentions it; nothing here calls it), exactly as in
//                                                        ref_math_vectors.cpp -- these two helpers are the reference's text and nothing else.
//   _ref/decision_eigen_vectors  (-DSSF_REF_EIGEN_ONLY)  eigenDecomposition alone, with ONE NAMED STAND-IN: it normalises its vectors
//                                                        with the reference's normalize() = v * rsqrtf(dot(v, v)) (vector_math.cuh:247-252);
//                                                        rsqrtf is a CUDA device intrinsic (approximate, <= 2 ulp) with no host definition and
//                                                        is defined HERE as 1.0f / sqrtf(x) -- the correctly rounded form this build specifies
//                                                        for both oracle and product (oracle_math.h / ssf_math.hpp unit3).  Consequence, stated
//                                                        in tests/test_math.py: the eigen-frame vectors are bit-exact against this generator, and
//                                                        what that does NOT pin is CUDA's own rsqrtf rounding (a relative 2^-22 on every component;
//                                                        bounded by tools/cuda_tolerance_study.py, arm rsqrt~2: no integer decision moves).
// Output: one JSON object, floats printed with %.9g (exact round trip for binary32).
#include <cmath>
#ifdef SSF_REF_EIGEN_ONLY
extern "C" float rsqrtf(float x) { return 1.0f / sqrtf(x); }      // THE stand-in (see above): this binary only
#else
extern "C" float rsqrtf(float x);                                  // declared, never defined, never called: no stand-in in this binary
#endif
#include <supersurfel_fusion/matrix_math.cuh>
#include <cstdint>
#include <cstdio>

namespace ref {
using std::isfinite;
#ifdef SSF_REF_EIGEN_ONLY
#include "_ref/decision_eigenDecomposition.inc"
#else
#include "_ref/decision_isUnchangeable.inc"
#include "_ref/decision_solvePlaneEquations.inc"
#endif
}  // namespace ref

static uint64_t st = 0x9E3779B97F4A7C15ULL;
static double urand() {  // splitmix64 -> [0,1)
    uint64_t z = (st += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31;
    return (double)(z >> 11) / 9007199254740992.0;
}
static double nrand() { double u = urand() + 1e-300, v = urand(); return std::sqrt(-2 * std::log(u)) * std::cos(6.283185307179586 * v); }
static bool first_in_row;
static void open_arr(const char* name, bool first = false) { std::printf("%s\"%s\": [", first ? "" : ",\n", name); first_in_row = true; }
static void put(float v) {
    if (!first_in_row) std::printf(", ");
    first_in_row = false;
    if (std::isnan(v)) std::printf("\"nan\""); else if (std::isinf(v)) std::printf(v > 0 ? "\"inf\"" : "\"-inf\"");
    else if (v == 0.0f && std::signbit(v)) std::printf("-0.0");       // ("-0" would be read back as the integer 0: the sign is part of the bits compared)
    else std::printf("%.9g", (double)v);
}
static void close_arr() { std::printf("]"); }

int main() {
    std::printf("{");
#ifndef SSF_REF_EIGEN_ONLY
    // ---- the connectivity guard over ALL 2^8 ring patterns.  Bit k of the pattern: ring pixel k (NW, N, NE, E, SE, S, SW, W)
    // carries the centre's label.  Two label alphabets (the guard only compares for equality): foreign pixels all alike, and
    // foreign pixels all different.
    {
        open_arr("guard_unchangeable", true);
        for (int alphabet = 0; alphabet < 2; alphabet++)
            for (int pat = 0; pat < 256; pat++) {
                const int ox[8] = {-1, 0, 1, 1, 1, 0, -1, -1}, oy[8] = {-1, -1, -1, 0, 1, 1, 1, 0};
                int img[9];
                img[4] = 7;
                for (int k = 0; k < 8; k++) img[(1 + oy[k]) * 3 + (1 + ox[k])] = ((pat >> k) & 1) ? 7 : (alphabet ? 100 + k : 3);
                put(ref::isUnchangeable(img, 1, 1, 3) ? 1.f : 0.f);
            }
        close_arr();
    }
    // ---- the plane solve: 4096 triples as the extract stage forms them -- rows (x, y, 1, d) of pixels in a superpixel
    // (RANSAC samples, TPS_RGBD_kernels.cu:387) and rows of the 3x3 normal equations (:262) --, plus degenerate ones:
    // collinear pixels, repeated pixels, zero rows, infinities and NaN (the guard `!isfinite(den) && den < eps` only
    // rejects -inf: kept as it is)
    {
        const int N = 4096 + 64;
        static float rows[N][12], th[N][3]; static int ok[N];
        for (int i = 0; i < N; i++) {
            float* r = rows[i];
            if (i < 2048) {                                           // three pixels, z = 1, disparity of a random plane + noise
                const double a = nrand() * 1e-4, b = nrand() * 1e-4, c = 0.2 + urand() * 2.0;
                for (int k = 0; k < 3; k++) {
                    const double x = std::floor(urand() * 640), y = std::floor(urand() * 480);
                    r[4 * k] = (float)x; r[4 * k + 1] = (float)y; r[4 * k + 2] = 1.f; r[4 * k + 3] = (float)(a * x + b * y + c + nrand() * 1e-5);
                }
                if (i % 64 == 0) { r[4] = r[0]; r[5] = r[1]; }        // a repeated pixel
                if (i % 64 == 1) { r[8] = 2 * r[4] - r[0]; r[9] = 2 * r[5] - r[1]; }   // collinear
                if (i % 64 == 2) { r[1] = r[5] = r[9] = 17.f; }       // one image row: denB <= eps on the first branch
            } else if (i < 4096) {                                    // normal equations: sums over n pixels of (xx xy x | xd), (xy yy y | yd), (x y n | d)
                const int n = 20 + (int)(urand() * 400);
                double sxx = 0, sxy = 0, sx = 0, syy = 0, sy = 0, sxd = 0, syd = 0, sd = 0;
                const double a = nrand() * 1e-4, b = nrand() * 1e-4, c = 0.2 + urand() * 2.0, x0 = urand() * 600, y0 = urand() * 440;
                for (int k = 0; k < n; k++) {
                    const double x = std::floor(x0 + urand() * 24), y = std::floor(y0 + urand() * 24), d = a * x + b * y + c + nrand() * 1e-5;
                    sxx += x * x; sxy += x * y; sx += x; syy += y * y; sy += y; sxd += x * d; syd += y * d; sd += d;
                }
                const float v[12] = {(float)sxx, (float)sxy, (float)sx, (float)sxd, (float)sxy, (float)syy, (float)sy, (float)syd, (float)sx, (float)sy, (float)n, (float)sd};
                for (int k = 0; k < 12; k++) r[k] = v[k];
                if (i % 128 == 0) for (int k = 0; k < 12; k++) r[k] = (k == 10) ? 1.f : 0.f;     // an empty superpixel's sums
            } else {                                                  // hostile values
                for (int k = 0; k < 12; k++) r[k] = (float)(nrand() * 10.0);
                const int w = i - 4096;
                if (w % 4 == 0) r[w % 12] = INFINITY;
                if (w % 4 == 1) r[w % 12] = -INFINITY;
                if (w % 4 == 2) r[w % 12] = NAN;
                if (w % 4 == 3) { r[2] = r[6] = r[10] = 0.f; }        // all z zero: every branch divides by zero
            }
            float4 t = make_float4(-7.f, -7.f, -7.f, -7.f);           // (a rejected solve leaves theta untouched)
            ok[i] = ref::solvePlaneEquations(t, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11]) ? 1 : 0;
            th[i][0] = t.x; th[i][1] = t.y; th[i][2] = t.z;
        }
        open_arr("plane_rows"); for (int i = 0; i < N; i++) for (int k = 0; k < 12; k++) put(rows[i][k]); close_arr();
        open_arr("plane_ok"); for (int i = 0; i < N; i++) put((float)ok[i]); close_arr();
        open_arr("plane_theta"); for (int i = 0; i < N; i++) for (int k = 0; k < 3; k++) put(th[i][k]); close_arr();
    }
#else
    // ---- the principal frame of 1024 SPD matrices at supersurfel scales (thin discs: two in-plane axes, a small normal
    // variance), n = 10 squarings as every caller passes (supersurfel_fusion_kernels.cu:209,334,673)
    {
        st = 0xD1B54A32D192ED03ULL;                                   // (its own stream: the two binaries share no state)
        const int N = 1024;
        static Cov3 c[N]; static Mat33 V[N]; static float3 L[N];
        for (int i = 0; i < N; i++) {
            double M[3][3];
            const double s1 = std::pow(10.0, -2.5 + 1.5 * urand()), s2 = s1 * (0.2 + 0.8 * urand()), s3 = s2 * std::pow(10.0, -2.0 + 1.7 * urand());
            // random orthonormal frame (Gram-Schmidt of Gaussian vectors) scaled by (s1, s2, s3)
            double q[3][3];
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) q[a][b] = nrand();
            for (int a = 0; a < 3; a++) {
                for (int p = 0; p < a; p++) { double d = 0; for (int b = 0; b < 3; b++) d += q[a][b] * q[p][b]; for (int b = 0; b < 3; b++) q[a][b] -= d * q[p][b]; }
                double n = 0; for (int b = 0; b < 3; b++) n += q[a][b] * q[a][b];
                n = std::sqrt(n); for (int b = 0; b < 3; b++) q[a][b] /= n;
            }
            const double s[3] = {s1 * s1, s2 * s2, s3 * s3};
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) { M[a][b] = 0; for (int k = 0; k < 3; k++) M[a][b] += s[k] * q[k][a] * q[k][b]; }
            c[i] = make_cov3((float)M[0][0], (float)M[0][1], (float)M[0][2], (float)M[1][1], (float)M[1][2], (float)M[2][2]);
            if (i % 128 == 0) c[i] = make_cov3((float)s[0], 0.f, 0.f, (float)s[1], 0.f, (float)s[2]);       // axis aligned: exact ties in the column choice
            if (i % 128 == 1) c[i] = make_cov3((float)s[0], 0.f, 0.f, (float)s[0], 0.f, (float)s[2]);       // a round disc: degenerate major axis
            V[i] = make_mat33(0, 0, 0, 0, 0, 0, 0, 0, 0); L[i] = make_float3(0, 0, 0);
            ref::eigenDecomposition(c[i], V[i], L[i], 10);
        }
        open_arr("eig_cov", true); for (int i = 0; i < N; i++) { put(c[i].xx); put(c[i].xy); put(c[i].xz); put(c[i].yy); put(c[i].yz); put(c[i].zz); } close_arr();
        open_arr("eig_vecs"); for (int i = 0; i < N; i++) for (int r = 0; r < 3; r++) { put(V[i].rows[r].x); put(V[i].rows[r].y); put(V[i].rows[r].z); } close_arr();
        open_arr("eig_vals"); for (int i = 0; i < N; i++) { put(L[i].x); put(L[i].y); put(L[i].z); } close_arr();
    }
#endif
    std::printf("\n}\n");
    return 0;
}
