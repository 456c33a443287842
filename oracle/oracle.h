// oracle.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// State of the deterministic single-threaded restatement of the supersurfel_fusion hot path.
// PARITY STATUS: the reference ships no unit tests or golden vectors for this path and its CUDA sources cannot be
// built or run here (no nvcc / OpenCV / ROS / NVIDIA GPU), so a kernel-by-kernel comparison with a reference
// EXECUTION is not obtainable: in that sense the restatement is "parity unpinned".  What IS pinned to material the
// reference holds:
//   * the host solvers (6x6 LDLT, LU inverse, quaternion re-normalisation, angle-axis increment, the align step)
//     against the reference's vendored Eigen 3.3.7 (oracle/_ref/eigen_vectors -> tests/golden/eigen_vectors.json);
//   * the per-element helpers against the reference's own vector_math.cuh / matrix_math.cuh compiled where they lie
//     (oracle/_ref/math_vectors -> tests/golden/ref_math_vectors.npz): inverse, square, Cov3 * v, mult_ABAt, Mat33
//     products, rotMatToQuat, quatToRotMat bit for bit; rgbToLab / labToRgb to 2e-4 Lab / 2e-3 grey levels (libm
//     powf / cbrtf there, a specified IEEE sequence here);
//   * the whole path, loosely, against the only end-to-end artefact the reference commits: over all 790 frames of the
//     rgbd_dataset_freiburg1_xyz sequence it ships, run the way its benchmark node runs it, this oracle's trajectory
//     has ATE 0.0226 m against ground truth (the reference's committed estimated.txt: 0.0195 m) and stays within
//     0.0102 m RMSE of that file (tests/golden/fr1_xyz_*, tests/test_replay.py).
//
// Schedule decisions where the reference is racy / non-deterministic:
//   A1  relabelling pass reads the pre-pass label map only (double buffer)
//   A2  boundary(x,y) == number of 4-neighbours with a different label (out-of-image differs),
//       always derived from the current label map (never stored incrementally)
//   A4  every accumulated sum is an exact integer: int for integer-valued terms, fixed-point
//       int64 for float terms (term-wise rint(v*2^S)); order independent, hence bit-reproducible
//   A9  plane filter sweeps are Jacobi (double buffered); the `x<gridSizeX` typo is kept, the
//       out-of-range read of the very last node is skipped
//   A13 association = exact arg-min of dist, ties -> lowest global id (packed u64 min)
//   A14 insertion order = ascending frame id; overflow drops the highest ids
//   cuRAND -> counter-based splitmix64 stream per (superpixel, sample).  The reference's XORWOW state
//       persists and advances by a data-dependent number of draws per frame; here the draws of
//       frame number e (frames extracted so far by this handle) use counters e*64 .. e*64+63 (a frame
//       consumes at most 20 + 30 draws per stream), i.e. a fixed skip-ahead per frame: every frame
//       still gets fresh draws of the same stream, and frames no longer depend on one another, so
//       their extraction can be batched / pipelined
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../include/ssf.h"
#include "oracle_math.h"

namespace orc {

// exact per-superpixel sums; replaces SuperpixelRGBDCoeffs (TPS_RGBD.hpp:39-43)
struct SpSums {
    int32_t sx, sy, sr, sg, sb, n;          // x, y, r, g, b, n   (integer valued in the reference)
    int32_t dx, dy, dn;                     // inlier-only sums of x, y, 1
    int64_t dxx, dyy, dxy;                  // inlier-only x^2, y^2, xy
    int64_t dxd, dyd, dd;                   // inlier-only x*d, y*d, d : fixed point 2^30
};
static const double DISP_SCALE = 1073741824.0;          // 2^30
static const double DISP_LIM = 4503599627370496.0;      // 2^52 per term
static const double MOM_SCALE = 16777216.0;             // 2^24
static const double MOM_LIM = 1099511627776.0;          // 2^40 per term

// SuperpixelRGBD (TPS_RGBD.hpp:32-37): xy_rg = (cx, cy, r, g), theta_b = (a, b, c, blue), size.x = n
struct Superpixel { float cx, cy, r, g, b, ta, tb, tc, size; };

struct Surfels {                                         // supersurfels.hpp:32-93
    std::vector<f3> pos, col;
    std::vector<int32_t> stamps;                         // 2 per surfel
    std::vector<Mat33> orient;
    std::vector<Cov3> shape;
    std::vector<float> dims;                             // 2 per surfel
    std::vector<float> conf;
    void resize(size_t n) {
        pos.resize(n); col.resize(n); stamps.resize(2 * n); orient.resize(n); shape.resize(n);
        dims.resize(2 * n); conf.resize(n);
    }
    void zero(size_t n);                                  // Supersurfels::memset, supersurfels.hpp:83-92
    void copy_row(size_t dst, const Surfels& s, size_t src);
};

struct Pose { Mat33 R; f3 t; };                           // Transform3, matrix_types.h:38-42

struct IcpState {                                         // locals of featureConstrainedSymmetricICP
    bool active = false, valid = true, done = false;
    int iter = 0;
    double tf_inc[16];
    double prev_error;
    double JtJ[36];
    Mat33 R_init; f3 t_init;                              // R_view, t_view
    Mat33 R_inc, R_corres; f3 t_inc, t_corres;
    Mat33 R_rel; f3 t_rel;
};

struct State {
    ssf_config cfg;
    int W, H, gx, gy, S;
    std::string err;
    // frame-level maps
    std::vector<uint32_t> rgba;        // R | G<<8 | B<<16 | 255<<24   (TPS_RGBD.cu:136)
    std::vector<float> disp, plane_depth;
    std::vector<int32_t> label, label_tmp;
    std::vector<uint8_t> inlier;
    std::vector<SpSums> sums;
    std::vector<Superpixel> sp;
    std::vector<float> samples;        // 4 per (superpixel, sample): a, b, c, score
    uint32_t extract_ordinal = 0;      // frames extracted so far: the RNG epoch of the next frame
    int max_passes = 0;
    Surfels frame, model;
    // rgbToLab(colour) caches: a pure function of the stored colour, refreshed whenever a colour is
    // written, so gates read exactly what the reference recomputes (dense_registration_kernels.cuh:226,
    // supersurfel_fusion_kernels.cu:579-580,631-632)
    std::vector<f3> frame_lab, model_lab;
    int n_model = 0, n_visible = 0, stamp = 0;
    Pose pose;
    IcpState icp;
    int64_t id_offset = 0, global_n_model = -1, global_n_visible = -1;
    bool have_frame = false;
    int last_icp_valid = 0, last_icp_iters = 0;
    // between fuse_begin and fuse_end
    int f_updated = 0, f_inserted = 0, f_old_rows = 0; bool f_first = false;
    std::vector<uint8_t> emigrant;       // per model row: leaves this shard in this frame (multi-GPU migration)
};

// oracle_extract.cpp
void extract(State& s, const uint8_t* rgb, const float* depth, const uint8_t* dynamic_mask);
int  boundary_at(const State& s, const std::vector<int32_t>& lab, int x, int y);
void bilateral_filter(const float* in, float* out, int W, int H, float sigma_color, float sigma_space);
// oracle_track_fuse.cpp
void icp_begin(State& s, const float* prior);
void icp_accumulate(State& s, int64_t* sums);
// loop-closure registration against the current frame (DenseRegistration::align); src_conf may be null (= 1)
bool align(State& s, const f3* src_pos, const f3* src_col, const Mat33* src_orient, const float* src_conf, int n,
           const float* init12, float* rel12, int* iters, int* pairs_last);
// host step of one align iteration (dense_registration.cu:168-210): sums = 29-value record, cs/ct = centroids
void align_increment(const double* JtJ, const double* Jtr, float scale, const float* cs, const float* ct, double* tf_iter16);
void icp_update(State& s, const int64_t* sums, int* again);
void icp_end(State& s, int* valid);
void match(State& s, uint64_t* best, uint8_t* matched);
void fuse(State& s, const uint64_t* best, const uint8_t* matched, ssf_frame_result* out);
// the same in two halves around the exchange of rows between shards (see ssf_stage_fuse_begin in ssf.h)
void fuse_begin(State& s, const uint64_t* best, const uint8_t* matched, int migrate, int32_t* table);
void fuse_end(State& s, const int32_t* table, ssf_frame_result* out);
void apply_deformation(State& s, const float* npos, const float* nrot, const float* ntrans, int m,
                       const float* w4, const int32_t* idx4);
int  shard_owner(const State& s, int f, const Pose& pose);
bool ring_unchangeable(int index, const int ring[8]);        // the connectivity guard on an explicit ring (test hook)
int  rehome_begin(State& s, int32_t* table, int cap);
int  rehome_end(State& s, const int32_t* table, int n);
void rot_to_quat(const Mat33& m, float* q /* x, y, z, w */);    // matrix_math.cuh:529-618
Mat33 quat_to_rot(const float* q);                              // matrix_math.cuh:512-527 (wy quirk kept)
// host solvers (pinned against the reference's vendored Eigen by oracle/_ref)
bool ldlt_solve6(const double* A /*36 row-major, symmetric*/, const double* b, double* x);
bool lu_inverse6(const double* A, double* Ainv);
void quat_normalize_rot_d(double* R9);   // Quaterniond(R).normalized().toRotationMatrix()
void quat_normalize_rot_f(float* R9);    // Quaternionf(R).normalized().toRotationMatrix()
void angle_axis_to_rot_d(double angle, const double* axis, double* R9);

}  // namespace orc
