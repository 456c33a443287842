// ref_math_vectors.cpp -- TEST INFRASTRUCTURE.  Built ONLY where /root/reference exists, into
// oracle/_ref/math_vectors (see oracle/Makefile).  SURVEY.md Appendix E, fixture G1.
//
// It includes the reference's own host/device math headers where they lie,
//   /root/reference/core/include/supersurfel_fusion/{matrix_types.h, vector_math.cuh, matrix_math.cuh}
// against the NVIDIA CUDA runtime headers that ship in this image (the Triton wheel's
// triton/backends/nvidia/include: cuda_runtime.h, vector_types.h, vector_functions.h -- NVIDIA's files, nothing
// written for this build), compiled by plain g++.  The one name those headers leave undeclared for a host compiler
// is the device intrinsic rsqrtf (used only by the reference's normalize(), which this program never calls): it is
// DECLARED below and never defined -- no implementation is substituted for anything of the reference's.
//
// Every function evaluated here is the reference's code: rgbToLab / labToRgb (vector_math.cuh:543-585), inverse
// (matrix_math.cuh:41-63), square (:184-194), Cov3 * float3 (:164-169), mult_ABAt (:442-459), Mat33 * Mat33
// (:381-403), transpose (:498-505), quatToRotMat incl. its wy = q.w*q.z (:512-527), rotMatToQuat (:529-618).
// Output: one JSON object, floats printed with %.9g (exact round trip for binary32).
extern "C" float rsqrtf(float);
#include <supersurfel_fusion/matrix_math.cuh>
#include <cmath>
#include <cstdint>
#include <cstdio>

static uint64_t st = 0x2545F4914F6CDD1DULL;
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
static void put3(float3 v) { put(v.x); put(v.y); put(v.z); }
static void put6(const Cov3& c) { put(c.xx); put(c.xy); put(c.xz); put(c.yy); put(c.yz); put(c.zz); }
static void put9(const Mat33& m) { for (int i = 0; i < 3; i++) put3(m.rows[i]); }

static Mat33 rot_from(double ax, double ay, double az) {      // Rz * Ry * Rx in float
    const float cx = (float)std::cos(ax), sx = (float)std::sin(ax), cy = (float)std::cos(ay), sy = (float)std::sin(ay),
                cz = (float)std::cos(az), sz = (float)std::sin(az);
    Mat33 Rx = make_mat33(1, 0, 0, 0, cx, -sx, 0, sx, cx), Ry = make_mat33(cy, 0, sy, 0, 1, 0, -sy, 0, cy),
          Rz = make_mat33(cz, -sz, 0, sz, cz, 0, 0, 0, 1);
    return Rz * (Ry * Rx);
}

int main() {
    const int NLAB = 4096, NCOV = 1024, NSING = 64, NROT = 1024;
    std::printf("{");
    // ---- colours: 8-bit triples (what the extract stage converts), arbitrary floats in range, and Lab values back
    {
        static float3 rgb[NLAB], lab[NLAB], back[NLAB];
        for (int i = 0; i < NLAB; i++) {
            if (i < NLAB / 2) rgb[i] = make_float3((float)(int)(urand() * 256), (float)(int)(urand() * 256), (float)(int)(urand() * 256));
            else rgb[i] = make_float3((float)(urand() * 255.0), (float)(urand() * 255.0), (float)(urand() * 255.0));
            if (i % 97 == 0) rgb[i].x = 0.f;                      // the linear segments of the transfer curves
            if (i % 89 == 0) rgb[i] = make_float3(rgb[i].x * 0.04f, rgb[i].y * 0.04f, rgb[i].z * 0.04f);
            lab[i] = rgbToLab(rgb[i]);
            back[i] = labToRgb(lab[i]);
        }
        open_arr("rgb", true); for (int i = 0; i < NLAB; i++) put3(rgb[i]); close_arr();
        open_arr("rgb_to_lab"); for (int i = 0; i < NLAB; i++) put3(lab[i]); close_arr();
        open_arr("lab_to_rgb_of_that"); for (int i = 0; i < NLAB; i++) put3(back[i]); close_arr();
        // free Lab values (confidence-weighted means of the update stage are not on the sRGB gamut's image)
        static float3 lab2[NLAB / 4], rgb2[NLAB / 4];
        for (int i = 0; i < NLAB / 4; i++) {
            lab2[i] = make_float3((float)(urand() * 100.0), (float)(urand() * 200.0 - 100.0), (float)(urand() * 200.0 - 100.0));
            rgb2[i] = labToRgb(lab2[i]);
        }
        open_arr("lab_free"); for (int i = 0; i < NLAB / 4; i++) put3(lab2[i]); close_arr();
        open_arr("lab_to_rgb_free"); for (int i = 0; i < NLAB / 4; i++) put3(rgb2[i]); close_arr();
    }
    // ---- symmetric 3x3: SPD at supersurfel scales + near-singular ones around the |det| > 1e-9 gate
    {
        const int N = NCOV + NSING;
        static Cov3 c[NCOV + NSING], inv[NCOV + NSING], sq[NCOV + NSING], aba[NCOV + NSING];
        static int ok[NCOV + NSING];
        static float tr[NCOV + NSING];
        static float3 v[NCOV + NSING], cv[NCOV + NSING];
        static Mat33 R[NCOV + NSING];
        for (int i = 0; i < N; i++) {
            double M[3][3];
            const double s = i < NCOV ? std::pow(10.0, -3.0 + 3.0 * urand()) : std::pow(10.0, -2.2 + 1.2 * urand());
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) M[a][b] = nrand() * s;
            if (i >= NCOV) for (int b = 0; b < 3; b++) M[2][b] = M[1][b] * (1.0 + 1e-4 * nrand());      // nearly rank deficient
            double C[3][3];
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) { C[a][b] = 0; for (int k = 0; k < 3; k++) C[a][b] += M[a][k] * M[b][k]; }
            c[i] = make_cov3((float)C[0][0], (float)C[0][1], (float)C[0][2], (float)C[1][1], (float)C[1][2], (float)C[2][2]);
            inv[i] = make_cov3(0, 0, 0, 0, 0, 0);
            ok[i] = inverse(c[i], inv[i]) ? 1 : 0;
            sq[i] = square(c[i]);
            tr[i] = trace(c[i]);
            v[i] = make_float3((float)nrand(), (float)nrand(), (float)nrand());
            cv[i] = c[i] * v[i];
            R[i] = rot_from(urand() * 6.28, urand() * 6.28, urand() * 6.28);
            aba[i] = mult_ABAt(R[i], c[i]);
        }
        open_arr("cov"); for (int i = 0; i < N; i++) put6(c[i]); close_arr();
        open_arr("cov_inverse_ok"); for (int i = 0; i < N; i++) put((float)ok[i]); close_arr();
        open_arr("cov_inverse"); for (int i = 0; i < N; i++) put6(inv[i]); close_arr();
        open_arr("cov_square"); for (int i = 0; i < N; i++) put6(sq[i]); close_arr();
        open_arr("cov_trace"); for (int i = 0; i < N; i++) put(tr[i]); close_arr();
        open_arr("vec"); for (int i = 0; i < N; i++) put3(v[i]); close_arr();
        open_arr("cov_times_vec"); for (int i = 0; i < N; i++) put3(cv[i]); close_arr();
        open_arr("rot"); for (int i = 0; i < N; i++) put9(R[i]); close_arr();
        open_arr("mult_ABAt"); for (int i = 0; i < N; i++) put6(aba[i]); close_arr();
    }
    // ---- rotations: products, transpose, matrix * vector, quaternion conversions (all four branches of rotMatToQuat)
    {
        static Mat33 A[NROT], B[NROT], AB[NROT], At[NROT], Q2R[NROT];
        static float3 v[NROT], Av[NROT], vA[NROT];
        static float4 q[NROT], qn[NROT];
        for (int i = 0; i < NROT; i++) {
            const double big = (i % 4 == 0) ? 1.0 : 0.03;                    // every 4th: large angles (trace <= 0 branches)
            A[i] = rot_from((urand() * 2 - 1) * 3.14159 * big + (i % 8 == 4 ? 3.0 : 0.0), (urand() * 2 - 1) * 3.14159 * big,
                            (urand() * 2 - 1) * 3.14159 * big + (i % 8 == 0 ? 3.0 : 0.0));
            B[i] = rot_from(urand() * 6.28, urand() * 6.28, urand() * 6.28);
            AB[i] = A[i] * B[i];
            At[i] = transpose(A[i]);
            v[i] = make_float3((float)nrand(), (float)nrand(), (float)nrand());
            Av[i] = A[i] * v[i];
            vA[i] = v[i] * A[i];
            q[i] = rotMatToQuat(A[i]);
            // a blended, re-normalised quaternion as applyDeformation forms it (deformation_graph_kernels.cu:52-61)
            const float4 q2 = rotMatToQuat(B[i]);
            const float w = (float)urand();
            float4 b = make_float4(w * q[i].x + (1.f - w) * q2.x, w * q[i].y + (1.f - w) * q2.y, w * q[i].z + (1.f - w) * q2.z,
                                   w * q[i].w + (1.f - w) * q2.w);
            const float len = sqrtf(b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w);
            b /= len;
            qn[i] = b;
            Q2R[i] = quatToRotMat(b);
        }
        open_arr("matA"); for (int i = 0; i < NROT; i++) put9(A[i]); close_arr();
        open_arr("matB"); for (int i = 0; i < NROT; i++) put9(B[i]); close_arr();
        open_arr("matA_times_matB"); for (int i = 0; i < NROT; i++) put9(AB[i]); close_arr();
        open_arr("matA_transposed"); for (int i = 0; i < NROT; i++) put9(At[i]); close_arr();
        open_arr("vec2"); for (int i = 0; i < NROT; i++) put3(v[i]); close_arr();
        open_arr("matA_times_vec"); for (int i = 0; i < NROT; i++) put3(Av[i]); close_arr();
        open_arr("vec_times_matA"); for (int i = 0; i < NROT; i++) put3(vA[i]); close_arr();
        open_arr("rotMatToQuat_of_matA"); for (int i = 0; i < NROT; i++) { put(q[i].x); put(q[i].y); put(q[i].z); put(q[i].w); } close_arr();
        open_arr("quat"); for (int i = 0; i < NROT; i++) { put(qn[i].x); put(qn[i].y); put(qn[i].z); put(qn[i].w); } close_arr();
        open_arr("quatToRotMat"); for (int i = 0; i < NROT; i++) put9(Q2R[i]); close_arr();
    }
    std::printf("\n}\n");
    return 0;
}
