// oracle_solvers.cpp -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Dependency-free restatement of the Eigen calls the reference makes on the host inside the ICP
// loop and the pose update:
//   JtJ.ldlt().solve(Jtr)                                   dense_registration.cu:367
//   Eigen::AngleAxisd(angle, axis) -> rotation              dense_registration.cu:377
//   Quaterniond(R).normalized().toRotationMatrix()          dense_registration.cu:384
//   JtJ.lu().inverse()                                      dense_registration.cu:394
//   Quaternionf(R).normalized().toRotationMatrix()          supersurfel_fusion.cu:324
// Eigen is a third-party dependency vendored by the reference (third_party/eigen3, version 3.3.7,
// Eigen/src/Core/util/Macros.h:14-16).  The algorithms below follow its published sources:
//   LDLT  : Eigen/src/Cholesky/LDLT.h  ldlt_inplace<Lower>::unblocked + _solve_impl
//   Quat  : Eigen/src/Geometry/Quaternion.h  quaternionbase_assign_impl<Other,3,3>, toRotationMatrix
//   AxisAng: Eigen/src/Geometry/AngleAxis.h  toRotationMatrix
// and are pinned against the real Eigen by oracle/_ref/eigen_check (tolerance 1e-12 relative:
// Eigen's vectorised reductions may associate differently).
#include <cmath>
#include <limits>
#include <utility>
#include "oracle.h"

namespace orc {

bool ldlt_solve6(const double* A, const double* b, double* x) {
    const int n = 6;
    double m[36];
    for (int i = 0; i < 36; i++) m[i] = A[i];
    int tr[6];
    double temp[6];
    bool zero_matrix = false;
    for (int k = 0; k < n; k++) {
        int big = k; double bigv = std::fabs(m[k * n + k]);
        for (int i = k + 1; i < n; i++) { double v = std::fabs(m[i * n + i]); if (v > bigv) { bigv = v; big = i; } }
        tr[k] = big;
        if (k != big) {
            for (int j = 0; j < k; j++) std::swap(m[k * n + j], m[big * n + j]);
            for (int i = big + 1; i < n; i++) std::swap(m[i * n + k], m[i * n + big]);
            std::swap(m[k * n + k], m[big * n + big]);
            for (int i = k + 1; i < big; i++) std::swap(m[i * n + k], m[big * n + i]);
        }
        if (k > 0) {
            for (int j = 0; j < k; j++) temp[j] = m[j * n + j] * m[k * n + j];
            double acc = 0.0;
            for (int j = 0; j < k; j++) acc += m[k * n + j] * temp[j];
            m[k * n + k] -= acc;
            for (int i = k + 1; i < n; i++) {
                double a2 = 0.0;
                for (int j = 0; j < k; j++) a2 += m[i * n + j] * temp[j];
                m[i * n + k] -= a2;
            }
        }
        double akk = m[k * n + k];
        bool valid = std::fabs(akk) > 0.0;
        if (k == 0 && !valid) { for (int j = 0; j < n; j++) tr[j] = j; zero_matrix = true; break; }
        if (valid) for (int i = k + 1; i < n; i++) m[i * n + k] /= akk;
    }
    double y[6];
    for (int i = 0; i < n; i++) y[i] = b[i];
    for (int k = 0; k < n; k++) std::swap(y[k], y[tr[k]]);
    if (!zero_matrix)
        for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) y[i] -= m[i * n + j] * y[j];
    const double tol = std::numeric_limits<double>::min();
    for (int i = 0; i < n; i++) { double d = m[i * n + i]; y[i] = (std::fabs(d) > tol) ? y[i] / d : 0.0; }
    if (!zero_matrix)
        for (int i = n - 1; i >= 0; i--) for (int j = i + 1; j < n; j++) y[i] -= m[j * n + i] * y[j];
    for (int k = n - 1; k >= 0; k--) std::swap(y[k], y[tr[k]]);
    for (int i = 0; i < n; i++) x[i] = y[i];
    return true;
}

// PartialPivLU + inverse (Eigen/src/LU/PartialPivLU.h): row pivoting on max |a_ik|.
bool lu_inverse6(const double* A, double* Ainv) {
    const int n = 6;
    double lu[36]; int perm[6];
    for (int i = 0; i < 36; i++) lu[i] = A[i];
    for (int i = 0; i < n; i++) perm[i] = i;
    for (int k = 0; k < n; k++) {
        int piv = k; double pv = std::fabs(lu[k * n + k]);
        for (int i = k + 1; i < n; i++) { double v = std::fabs(lu[i * n + k]); if (v > pv) { pv = v; piv = i; } }
        if (piv != k) { for (int j = 0; j < n; j++) std::swap(lu[k * n + j], lu[piv * n + j]); std::swap(perm[k], perm[piv]); }
        for (int i = k + 1; i < n; i++) {
            lu[i * n + k] /= lu[k * n + k];
            for (int j = k + 1; j < n; j++) lu[i * n + j] -= lu[i * n + k] * lu[k * n + j];
        }
    }
    for (int c = 0; c < n; c++) {
        double y[6];
        for (int i = 0; i < n; i++) y[i] = (perm[i] == c) ? 1.0 : 0.0;
        for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) y[i] -= lu[i * n + j] * y[j];
        for (int i = n - 1; i >= 0; i--) {
            for (int j = i + 1; j < n; j++) y[i] -= lu[i * n + j] * y[j];
            y[i] /= lu[i * n + i];
        }
        for (int i = 0; i < n; i++) Ainv[i * n + c] = y[i];
    }
    return true;
}

template <typename T>
static void quat_normalize_rot(T* R) {
    // matrix -> quaternion (Shoemake), Quaternion.h:760-797
    T q[4];  // x, y, z, w
    T t = (R[0] + R[4]) + R[8];
    if (t > T(0)) {
        t = std::sqrt(t + T(1.0));
        q[3] = T(0.5) * t;
        t = T(0.5) / t;
        q[0] = (R[7] - R[5]) * t;
        q[1] = (R[2] - R[6]) * t;
        q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(((R[i * 3 + i] - R[j * 3 + j]) - R[k * 3 + k]) + T(1.0));
        q[i] = T(0.5) * t;
        t = T(0.5) / t;
        q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
        q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
    // normalized(): n / sqrt(squaredNorm) when squaredNorm > 0
    T z = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
    if (z > T(0)) { T nrm = std::sqrt(z); for (int a = 0; a < 4; a++) q[a] = q[a] / nrm; }
    // toRotationMatrix, Quaternion.h:536-569
    const T tx = T(2) * q[0], ty = T(2) * q[1], tz = T(2) * q[2];
    const T twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const T txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const T tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = T(1) - (tyy + tzz); R[1] = txy - twz;          R[2] = txz + twy;
    R[3] = txy + twz;          R[4] = T(1) - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;          R[7] = tyz + twx;          R[8] = T(1) - (txx + tyy);
}
void quat_normalize_rot_d(double* R9) { quat_normalize_rot<double>(R9); }
void quat_normalize_rot_f(float* R9) { quat_normalize_rot<float>(R9); }

// AngleAxis.h:218-243
void angle_axis_to_rot_d(double angle, const double* ax, double* R) {
    const double s = std::sin(angle), c = std::cos(angle);
    const double sx = s * ax[0], sy = s * ax[1], sz = s * ax[2];
    const double cx = (1.0 - c) * ax[0], cy = (1.0 - c) * ax[1], cz = (1.0 - c) * ax[2];
    double tmp;
    tmp = cx * ax[1]; R[1] = tmp - sz; R[3] = tmp + sz;
    tmp = cx * ax[2]; R[2] = tmp + sy; R[6] = tmp - sy;
    tmp = cy * ax[2]; R[5] = tmp - sx; R[7] = tmp + sx;
    R[0] = cx * ax[0] + c; R[4] = cy * ax[1] + c; R[8] = cz * ax[2] + c;
}

}  // namespace orc
