/*
 * ssf_testing.h -- test hooks exported by both libssf_hip.so and libssf_oracle.so next to the
 * ABI of ssf.h.  They expose the dependency-free host solvers that replace the reference's Eigen
 * calls inside the ICP loop (core/src/dense_registration.cu:367,377-378,384,394) and the pose
 * re-normalisation (core/src/supersurfel_fusion.cu:324), so that tests can pin them against
 * tests/golden/eigen_vectors.json (generated from the reference's vendored Eigen 3.3.7) on a box
 * without a GPU.  Not part of the drop-in surface.
 */
#ifndef SSF_TESTING_H
#define SSF_TESTING_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* x = A^-1 b with the diagonally pivoted LDL^T of Eigen::LDLT; A 6x6 row-major symmetric. */
int ssf_dbg_ldlt_solve6(const double* A36, const double* b6, double* x6);
/* Ainv = PartialPivLU(A).inverse() */
int ssf_dbg_lu_inverse6(const double* A36, double* Ainv36);
/* R <- Quaternion(R).normalized().toRotationMatrix(), double / float */
int ssf_dbg_renormalise_d(double* R9);
int ssf_dbg_renormalise_f(float* R9);
/* tf_iter (4x4 row-major) from the solved 6-vector X = (omega, tau), dense_registration.cu:369-384 */
int ssf_dbg_gn_increment(const double* X6, double* tf16);
/* host step of one loop-closure (align) iteration: LDLT solve of (JtJ, Jtr), translation divided by the
 * normalisation scale, T(ct) * Rot * T(tran) * Rot * T(-cs), dense_registration.cu:186-205 */
int ssf_dbg_align_increment(const double* JtJ36, const double* Jtr6, float scale, const float* cs3, const float* ct3, double* tf16);
/* per-element arithmetic of the kernels, evaluated on the host (same inline code as on the device):
 * rgbToLab / labToRgb (vector_math.cuh:543-585), inverse(Cov3) (matrix_math.cuh:41-63),
 * eigenDecomposition (supersurfel_fusion_kernels.cu:48-111), solvePlaneEquations
 * (TPS_RGBD_kernels.cu:27-59).  Each returns what the kernel would compute for one element. */
int ssf_dbg_rgb_to_lab(const float* rgb3, float* lab3);
int ssf_dbg_lab_to_rgb(const float* lab3, float* rgb3);
int ssf_dbg_sym_inverse(const float* cov6, float* inv6);            /* returns 1 when invertible */
int ssf_dbg_principal_frame(const float* cov6, float* vecs9, float* vals3);
int ssf_dbg_plane_solve(const float* rows12, float* theta3);        /* returns 1 when accepted */
/* the connectivity guard of a relabelling pass, isUnchangeable (TPS_RGBD_kernels.cuh:178-233), on a 3 x 3 label patch
 * (row-major): 1 when the centre pixel's label may not change.  The three decision helpers -- this one, the plane solve and
 * the principal frame -- are pinned to the reference's own text by tests/golden/ref_decision_vectors.npz
 * (oracle/ref_decision_vectors.cpp). */
int ssf_dbg_connectivity_guard(const int32_t* labels9);
/* the matrix helpers the fuse / deformation kernels are built from, pinned against the reference's own headers by
 * tests/golden/ref_math_vectors.npz (oracle/ref_math_vectors.cpp): square(Cov3) (matrix_math.cuh:184), Cov3 * float3
 * (:164), mult_ABAt (:442), Mat33 * Mat33 (:381), Mat33 * float3 (:484), float3 * Mat33 (:491), rotMatToQuat (:529),
 * quatToRotMat (:512, with its wy = q.w*q.z).  Matrices are 9 floats row-major, quaternions (x, y, z, w). */
int ssf_dbg_sym_square(const float* cov6, float* out6);
int ssf_dbg_sym_mulv(const float* cov6, const float* v3, float* out3);
int ssf_dbg_mult_abat(const float* R9, const float* cov6, float* out6);
int ssf_dbg_m3_mul(const float* A9, const float* B9, float* out9);
int ssf_dbg_m3_mulv(const float* A9, const float* v3, float* out3);
int ssf_dbg_row_mul(const float* v3, const float* A9, float* out3);
int ssf_dbg_rot_to_quat(const float* R9, float* q4);
int ssf_dbg_quat_to_rot(const float* q4, float* R9);
#ifdef __cplusplus
}
#endif
#endif
