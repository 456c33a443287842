// ref_eigen_vectors.cpp -- TEST INFRASTRUCTURE.  Built ONLY where /root/reference exists, into
// oracle/_ref/eigen_vectors (see oracle/Makefile).  It links nothing but the reference's own
// vendored, header-only Eigen 3.3.7 (/root/reference/third_party/eigen3) and performs exactly
// the Eigen calls the reference makes on the host:
//   JtJ.ldlt().solve(Jtr)                                     core/src/dense_registration.cu:367
//   Eigen::AngleAxisd(angle, axis), iso_rot*Translation*iso_rot  :377-378
//   Eigen::Quaterniond(R).normalized().toRotationMatrix()     :384
//   JtJ.cast<double>().lu().inverse()                         :394
//   Eigen::Quaternionf(R).normalized().toRotationMatrix()     core/src/supersurfel_fusion.cu:324
//   the host step of DenseRegistration::align (loop closure)   core/src/dense_registration.cu:186-205
// on seeded inputs, printing inputs and outputs as JSON lines.  The output is committed as
// tests/golden/eigen_vectors.json and pins the dependency-free solvers of the oracle and of the
// product (tests/test_solvers.py).
#include <Eigen/Dense>
#include <Eigen/Geometry>
#include <cstdint>
#include <cstdio>
#include <cmath>

static uint64_t st = 0x853c49e6748fea9bULL;
static double urand() {  // splitmix64 -> [0,1)
    uint64_t z = (st += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31;
    return (double)(z >> 11) / 9007199254740992.0;
}
static uint64_t st2 = 0x1234567887654321ULL;      // second stream: the align inputs (keeps the older vectors unchanged)
static double urand2() {
    uint64_t z = (st2 += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31;
    return (double)(z >> 11) / 9007199254740992.0;
}
static double nrand() { double u = urand() + 1e-300, v = urand(); return std::sqrt(-2 * std::log(u)) * std::cos(6.283185307179586 * v); }
template <typename M> static void pr(const char* name, const M& m, bool last = false) {
    std::printf("\"%s\": [", name);
    for (int i = 0; i < m.rows(); i++) for (int j = 0; j < m.cols(); j++)
        std::printf("%s%.17g", (i + j) ? ", " : "", (double)m(i, j));
    std::printf("]%s", last ? "" : ", ");
}
int main() {
    const int N = 64;
    for (int c = 0; c < N; c++) {
        // ICP-like normal equations: sum of x x^T over n rows, optionally rank deficient
        int n = (c % 8 == 7) ? 4 : 200 + c * 50;
        Eigen::Matrix<double, 6, 6> JtJ = Eigen::Matrix<double, 6, 6>::Zero();
        Eigen::Matrix<double, 6, 1> Jtr = Eigen::Matrix<double, 6, 1>::Zero();
        for (int k = 0; k < n; k++) {
            Eigen::Matrix<double, 6, 1> x;
            for (int i = 0; i < 6; i++) x(i) = nrand() * (i < 3 ? 2.5 : 0.6);
            double r = nrand() * 0.01;
            JtJ += x * x.transpose(); Jtr += r * x;
        }
        Eigen::Matrix<double, 6, 1> Xp = JtJ.ldlt().solve(Jtr);
        Eigen::MatrixXd cov = JtJ.cast<double>().lu().inverse();
        Eigen::Vector3d tran(Xp(3), Xp(4), Xp(5)), rot_axis(Xp(0), Xp(1), Xp(2));
        if (c % 4 == 1) rot_axis *= 50.0;  // exercise larger angles
        double rot_axis_norm = rot_axis.norm();
        double rot_angle = 0.5f * std::atan(rot_axis_norm);
        rot_axis /= rot_axis_norm;
        tran *= std::cos(rot_angle);
        Eigen::Isometry3d iso_rot(Eigen::AngleAxisd(rot_angle, rot_axis));
        Eigen::Isometry3d iso_iter = iso_rot * Eigen::Isometry3d(Eigen::Translation3d(tran)) * iso_rot;
        Eigen::Matrix4d tf_iter = iso_iter.matrix();
        Eigen::Matrix3d Rn = Eigen::Quaterniond(tf_iter.block<3, 3>(0, 0)).normalized().toRotationMatrix();
        // float pose re-normalisation on a slightly non-orthonormal, arbitrarily oriented rotation
        Eigen::Vector3d ax(nrand(), nrand(), nrand()); ax.normalize();
        Eigen::Matrix3f Rf = Eigen::AngleAxisd(urand() * 6.2, ax).toRotationMatrix().cast<float>();
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rf(i, j) += (float)(nrand() * 1e-4);
        Eigen::Matrix3f Rfn = Eigen::Quaternionf(Rf).normalized().toRotationMatrix();
        Eigen::Matrix<double, 6, 1> covd = cov.diagonal();
        Eigen::Matrix<double, 1, 1> ang; ang(0) = rot_angle;
        Eigen::Matrix<double, 3, 1> axis_in(Xp(0), Xp(1), Xp(2)); if (c % 4 == 1) axis_in *= 50.0;
        Eigen::Matrix<double, 3, 1> tran_in(Xp(3), Xp(4), Xp(5));
        Eigen::Matrix3d Riso = iso_rot.matrix().block<3, 3>(0, 0);
        // DenseRegistration::align, dense_registration.cu:186-205: scale- and centroid-normalised increment
        float a_scale = (float)(0.4 + 2.0 * urand2());
        Eigen::Vector3f a_cs((float)(3.0 * urand2() - 1.5), (float)(3.0 * urand2() - 1.5), (float)(1.0 + 3.0 * urand2()));
        Eigen::Vector3f a_ct((float)(3.0 * urand2() - 1.5), (float)(3.0 * urand2() - 1.5), (float)(1.0 + 3.0 * urand2()));
        Eigen::Matrix4d a_tf;
        {
            Eigen::Matrix<double, 6, 1> Xa = JtJ.ldlt().solve(Jtr);
            Eigen::Vector3d tran(Xa(3), Xa(4), Xa(5));
            Eigen::Vector3d rot_axis(Xa(0), Xa(1), Xa(2));
            double rot_axis_norm = rot_axis.norm();
            double rot_angle = 0.5f * std::atan(rot_axis_norm);
            rot_axis /= rot_axis_norm;
            tran /= a_scale;
            tran *= std::cos(rot_angle);
            Eigen::Isometry3d iso_rot(Eigen::AngleAxisd(rot_angle, rot_axis));
            Eigen::Isometry3d iso_iter = Eigen::Isometry3d(Eigen::Translation3d(double(a_ct.x()), double(a_ct.y()), double(a_ct.z()))) *
                                         iso_rot * Eigen::Isometry3d(Eigen::Translation3d(tran)) * iso_rot *
                                         Eigen::Isometry3d(Eigen::Translation3d(-1.0 * double(a_cs.x()), -1.0 * double(a_cs.y()), -1.0 * double(a_cs.z())));
            a_tf << iso_iter.matrix()(0, 0), iso_iter.matrix()(0, 1), iso_iter.matrix()(0, 2), iso_iter.matrix()(0, 3),
                    iso_iter.matrix()(1, 0), iso_iter.matrix()(1, 1), iso_iter.matrix()(1, 2), iso_iter.matrix()(1, 3),
                    iso_iter.matrix()(2, 0), iso_iter.matrix()(2, 1), iso_iter.matrix()(2, 2), iso_iter.matrix()(2, 3),
                    0.0f, 0.0f, 0.0f, 1.0f;
            a_tf.block<3, 3>(0, 0) = Eigen::Quaterniond(a_tf.block<3, 3>(0, 0)).normalized().toRotationMatrix();
        }
        Eigen::Matrix<double, 1, 1> a_sc; a_sc(0) = (double)a_scale;
        std::printf("{");
        pr("align_scale", a_sc); pr("align_cs", a_cs); pr("align_ct", a_ct); pr("align_tf_iter", a_tf);
        pr("JtJ", JtJ); pr("Jtr", Jtr); pr("ldlt_x", Xp); pr("lu_inv_diag", covd);
        pr("axis_in", axis_in); pr("tran_in", tran_in); pr("angle", ang); pr("R_angleaxis", Riso);
        pr("tf_iter", tf_iter); pr("R_quatd", Rn); pr("Rf_in", Rf); pr("Rf_quatf", Rfn, true);
        std::printf("}\n");
    }
    return 0;
}
