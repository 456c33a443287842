/* ssf.hpp -- header-only C++ surface over the C ABI of ssf.h, with the method names of the reference's
 * supersurfel_fusion::SupersurfelFusion (core/include/supersurfel_fusion/supersurfel_fusion.hpp:40-143) for the
 * hot path: initialize / processFrame / getPose / getnbSupersurfels / getStamp / getModel / exportModel.
 * A node that owns a `supersurfel_fusion::SupersurfelFusion ssf;` member includes this header instead of the
 * reference's and links libssf_hip.so.  No OpenCV needed: processFrame takes raw pointers; the
 * cv::Mat overloads appear when <opencv2/core.hpp> has been included before this header.
 *
 * Sparse VO, MOD and loop closure stay with the caller; their outputs enter as `vo_pose` and `dynamic`.
 * Errors: the reference exits the process on a CUDA failure (cuda_error_check.h:30-66); this surface throws
 * std::runtime_error with the library's message. */
#ifndef SSF_HPP
#define SSF_HPP
#include <stdexcept>
#include <string>
#include <vector>
#include "ssf.h"

/* The reference's pose / matrix types (core/include/supersurfel_fusion/matrix_types.h:26-42), at GLOBAL scope as there,
 * so that the nodes' lines compile as they stand:
 *     Transform3 pose = ssf.getPose();
 *     tf::Matrix3x3(pose.R.rows[0].x, pose.R.rows[0].y, ... ), tf::Vector3(pose.t.x, pose.t.y, pose.t.z)
 * (node/supersurfel_fusion_node.cpp:87-91, node/supersurfel_fusion_rgbd_benchmark_node.cpp:616-620).  The reference gets
 * float3 from <cuda_runtime.h>; a translation unit that already has HIP's or CUDA's vector types keeps those (same three
 * floats x, y, z), any other gets the plain struct below.  SSF_NO_MATRIX_TYPES: the includer brings its own
 * matrix_types.h. */
#ifndef SSF_NO_MATRIX_TYPES
/* Include order: a translation unit that uses HIP's / CUDA's vector types includes THEIR header before this one (then
 * float3 / float2 / int2 below are theirs).  The other order is a compile error (redefinition of float3 in the vendor
 * header), never a silent mismatch: the three fallbacks have the vendor types' size and field order, asserted below. */
#if !defined(HIP_INCLUDE_HIP_AMD_DETAIL_HIP_VECTOR_TYPES_H) && !defined(__VECTOR_TYPES_H__) && !defined(SSF_HAVE_FLOAT3)
#define SSF_HAVE_FLOAT3
struct float3 { float x, y, z; };
struct float2 { float x, y; };
struct int2 { int x, y; };
#endif
static_assert(sizeof(float3) == 12 && sizeof(float2) == 8 && sizeof(int2) == 8, "ssf.hpp: float3 / float2 / int2 must be the packed vendor layouts");
#ifndef MATRIX_TYPES_HPP            /* the reference header's own guard: both may be included, in either order */
#define MATRIX_TYPES_HPP
struct Cov3 { float xx, xy, xz, yy, yz, zz; };                          /* matrix_types.h:26-31 */
struct Mat33 { float3 rows[3]; };                                       /* matrix_types.h:33-36 */
struct Transform3 { Mat33 R; float3 t; };                               /* matrix_types.h:38-42: camera-to-map */
#endif
#endif

namespace supersurfel_fusion {

struct CamParam { float fx, fy, cx, cy; int height, width; };          /* cam_param.hpp:27-31 */
using ::Transform3; using ::Mat33; using ::Cov3; using ::float3;       /* the reference's are global; both spellings work */

/* Transform3 <-> the C ABI's 12 floats (row-major R, then t) */
inline Transform3 transform3_from_rt(const float v[12]) {
    Transform3 p;
    for (int r = 0; r < 3; r++) { p.R.rows[r].x = v[3 * r]; p.R.rows[r].y = v[3 * r + 1]; p.R.rows[r].z = v[3 * r + 2]; }
    p.t.x = v[9]; p.t.y = v[10]; p.t.z = v[11];
    return p;
}
inline void transform3_to_rt(const Transform3& p, float v[12]) {
    for (int r = 0; r < 3; r++) { v[3 * r] = p.R.rows[r].x; v[3 * r + 1] = p.R.rows[r].y; v[3 * r + 2] = p.R.rows[r].z; }
    v[9] = p.t.x; v[10] = p.t.y; v[11] = p.t.z;
}

/* host copy of a supersurfel set in the reference's SoA layout (supersurfels.hpp:34-40) */
struct HostSupersurfels {
    std::vector<float> positions, colors, orientations, shapes, dims, confidences;
    std::vector<int32_t> stamps;
    int size = 0;
    void resize(int n) {
        size = n; const size_t m = (size_t)(n > 0 ? n : 1);
        positions.resize(3 * m); colors.resize(3 * m); stamps.resize(2 * m); orientations.resize(9 * m);
        shapes.resize(6 * m); dims.resize(2 * m); confidences.resize(m);
    }
    ssf_surfels view() {
        ssf_surfels v; v.positions = positions.data(); v.colors = colors.data(); v.stamps = stamps.data();
        v.orientations = orientations.data(); v.shapes = shapes.data(); v.dims = dims.data(); v.confidences = confidences.data();
        return v;
    }
};

/* ---- getModel() / getFrame() as the reference returns them: device-resident arrays the callers hand to thrust ----------------
 * The reference's Supersurfels (supersurfels.hpp:32-40) holds seven thrust::device_vectors and its nodes copy them out with
 *     thrust::host_vector<float3> positions(ssf.getModel().positions.begin(), ssf.getModel().positions.begin() + ssf.getnbSupersurfels());
 *     thrust::host_vector<Mat33> orientations(ssf.getFrame().orientations);
 * (node/supersurfel_fusion_node.cpp:306-310,423-427,688-690; ...benchmark_node.cpp:189-193,305-306).  A node built for AMD has
 * rocThrust (hipcc, /opt/rocm/include/thrust): when <thrust/...> was included before this header, Supersurfels is a VIEW of the
 * library's device arrays with the same seven member names -- each a DeviceArray<T>: begin() / end() as thrust::device_ptr<T>,
 * size(), and a conversion to thrust::host_vector<T> for the whole-array form -- and getModel() / getFrame() return
 * `const Supersurfels&` exactly as supersurfel_fusion.hpp:86-87: those lines compile as they stand (tests/cpp/node_model_copy.cpp,
 * compiled by hipcc against rocThrust).  The view covers the n valid rows (the reference's vectors have capacity
 * nb_supersurfels_max; its callers stop at getnbSupersurfels()); it is valid until the next call on the object.  Without
 * thrust (plain g++) getModel() / getFrame() hand back host copies (HostSupersurfels), also available as getModelHost() /
 * getFrameHost() in both builds.  SSF_NO_THRUST_VIEW: keep the host-copy form although thrust is there. */
#if defined(THRUST_VERSION) && !defined(SSF_NO_THRUST_VIEW)
#define SSF_THRUST_VIEW 1
}  /* namespace supersurfel_fusion */
#include <thrust/device_ptr.h>
#include <thrust/host_vector.h>
#include <thrust/copy.h>
namespace supersurfel_fusion {
#endif
/* DATA LAYOUT IS UNCONDITIONAL: DeviceArray<T> is a pointer and a count whether or not thrust is there -- only its
 * thrust-typed accessors depend on the include order -- and SupersurfelFusion always holds its two views, so
 * sizeof(SupersurfelFusion) is the same in every translation unit of a node (a .hip file with thrust and a main.cpp
 * without may share one object).  What DOES differ between the two forms is the return type of getModel() / getFrame();
 * the class therefore lives in an inline namespace named after the form (`thrust_view` / `host_copy`): code never spells
 * it, but a function that passes a SupersurfelFusion between translation units of different forms fails to LINK instead
 * of calling the wrong getModel(). */
/* (round 6: DeviceArray and Supersurfels live in the inline namespace too -- their member SETS differ between the forms, and two
 * definitions of one class name in one namespace would be an ODR violation even where the layout agrees) */
#ifdef SSF_THRUST_VIEW
inline namespace thrust_view {
#else
inline namespace host_copy {
#endif

template <typename T> struct DeviceArray {
    T* ptr = nullptr; size_t n = 0;
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T* data() const { return ptr; }
#ifdef SSF_THRUST_VIEW
    typedef thrust::device_ptr<T> iterator;
    typedef thrust::device_ptr<T> const_iterator;
    iterator begin() const { return thrust::device_pointer_cast(ptr); }
    iterator end() const { return thrust::device_pointer_cast(ptr) + n; }
    operator thrust::host_vector<T>() const { thrust::host_vector<T> v(n); thrust::copy(begin(), end(), v.begin()); return v; }
#endif
};
struct Supersurfels {                                                    /* supersurfels.hpp:32-40, as views */
    DeviceArray<float3> positions, colors;
    DeviceArray<int2> stamps;
    DeviceArray<Mat33> orientations;
    DeviceArray<Cov3> shapes;
    DeviceArray<float2> dims;
    DeviceArray<float> confidences;
    void bind(const ssf_surfels& v, size_t n) {
        positions.ptr = reinterpret_cast<float3*>(v.positions); colors.ptr = reinterpret_cast<float3*>(v.colors);
        stamps.ptr = reinterpret_cast<int2*>(v.stamps); orientations.ptr = reinterpret_cast<Mat33*>(v.orientations);
        shapes.ptr = reinterpret_cast<Cov3*>(v.shapes); dims.ptr = reinterpret_cast<float2*>(v.dims); confidences.ptr = v.confidences;
        positions.n = colors.n = stamps.n = orientations.n = shapes.n = dims.n = confidences.n = n;
    }
};
static_assert(sizeof(DeviceArray<float3>) == sizeof(void*) + sizeof(size_t) && sizeof(Supersurfels) == 7 * sizeof(DeviceArray<float>),
              "ssf.hpp: the device views are (pointer, count) pairs in every translation unit");

class SupersurfelFusion {
public:
    SupersurfelFusion() = default;
    SupersurfelFusion(const SupersurfelFusion&) = delete;
    SupersurfelFusion& operator=(const SupersurfelFusion&) = delete;
    ~SupersurfelFusion() { if (h_) ssf_destroy(h_); }

    /* initialize(): supersurfel_fusion.hpp:46-74 -- the reference's complete parameter list, in its order and with its
     * defaults, so that the nodes' calls (node/supersurfel_fusion_node.cpp:256-284,
     * node/supersurfel_fusion_rgbd_benchmark_node.cpp: same 29 positional arguments) compile unchanged.  The 21
     * path-relevant arguments map 1:1 onto ssf_config; the last eight configure the reference's sparse VO (ORB
     * features), loop closure and MOD, which are outside this library (their outputs enter processFrame as `vo_pose`
     * and `dynamic`): accepted and ignored.  What this library adds -- pipelining, batching, the pre-filter switch --
     * is set by name BEFORE initialize (setPipeline / setDepthPrefilter below) or through initialize(const ssf_config&);
     * the defaults are the reference's behaviour: one frame in flight, processFrame filters the depth image first
     * (supersurfel_fusion.cu:180). */
    void initialize(const CamParam& cam, int cell_size = 16, float lambda_pos = 50.f, float lambda_bound = 1000.f,
                    float lambda_size = 10000.f, float lambda_disp = 1e6f, float thresh_disp = 1e-4f,
                    int seg_iter = 10, bool seg_use_ransac = true, int nb_samples = 16, int filter_iter = 4,
                    float filter_alpha = 0.1f, float filter_beta = 1.0f, float filter_threshold = 0.05f,
                    float range_min = 0.2f, float range_max = 5.0f, int delta_t = 20, float conf_thresh = 2500.f,
                    int nb_supersurfels_max = 50000, int icp_iter = 10, double icp_cov_thresh = 0.04,
                    int nb_features = 2000, float features_scale_factor = 1.2f, int features_nb_levels = 8,
                    int ini_th_fast = 20, int min_th_fast = 7, int untracked_threshold = 10,
                    bool enable_loop_closure = true, bool enable_mod = true) {
        (void)nb_features; (void)features_scale_factor; (void)features_nb_levels; (void)ini_th_fast; (void)min_th_fast;
        (void)untracked_threshold; (void)enable_loop_closure; (void)enable_mod;
        ssf_config c; ssf_default_config(&c);
        c.width = cam.width; c.height = cam.height; c.fx = cam.fx; c.fy = cam.fy; c.cx = cam.cx; c.cy = cam.cy;
        c.cell_size = cell_size; c.lambda_pos = lambda_pos; c.lambda_bound = lambda_bound; c.lambda_size = lambda_size;
        c.lambda_disp = lambda_disp; c.thresh_disp = thresh_disp; c.seg_iter = seg_iter; c.seg_use_ransac = seg_use_ransac ? 1 : 0;
        c.nb_samples = nb_samples; c.filter_iter = filter_iter; c.filter_alpha = filter_alpha; c.filter_beta = filter_beta;
        c.filter_threshold = filter_threshold; c.range_min = range_min; c.range_max = range_max; c.delta_t = delta_t;
        c.conf_thresh = conf_thresh; c.nb_supersurfels_max = nb_supersurfels_max; c.icp_iter = icp_iter;
        c.icp_cov_thresh = icp_cov_thresh; c.pipeline_depth = pipeline_depth_; c.extract_batch = extract_batch_;
        c.depth_prefilter = depth_prefilter_ ? 1 : 0;
        initialize(c);
    }
    /* library-specific knobs, by name; they take effect at the next initialize().  pipeline_depth / extract_batch: see
     * ssf_config (0 / 1 = the reference's one-frame-in-flight behaviour; 2 / 8 for replay through processSequence).
     * depth_prefilter false = the caller hands over the depth it wants segmented. */
    void setPipeline(int pipeline_depth, int extract_batch) { pipeline_depth_ = pipeline_depth; extract_batch_ = extract_batch; }
    void setDepthPrefilter(bool on) { depth_prefilter_ = on; }
    void initialize(const ssf_config& c) {
        if (h_) { ssf_destroy(h_); h_ = nullptr; }
        if (ssf_create(&c, &h_) != SSF_OK) { h_ = nullptr; throw std::runtime_error(ssf_last_error(nullptr)); }
        width_ = c.width; height_ = c.height;
    }
    bool isInitialized() const { return h_ != nullptr; }

    /* processFrame(): supersurfel_fusion.hpp:75-76.  rgb: H x W x 3 bytes in RGB order, depth: H x W floats in
     * metres (0 = hole) -- what RGBDCallback / run() build (convertTo(CV_32FC1, depthScale)).  vo_pose: the sparse-VO
     * pose prior (supersurfel_fusion.cu:225-228; row-major R then t; nullptr = previous pose); dynamic: the MOD
     * mask, one byte per superpixel (motion_detection.cu:573-578; nullptr = none). */
    void processFrame(const uint8_t* rgb, const float* depth_m, const float* vo_pose = nullptr, const uint8_t* dynamic = nullptr) {
        check(ssf_process_frame(need(), rgb, depth_m, vo_pose, dynamic, &last_));
    }
    /* replay of a recorded sequence (SupersurfelFusionRGBDBenchmarkNode::run): host images of n frames, results in
     * order; with pipeline_depth / extract_batch > 0 / 1 the extract stage runs ahead (bit-identical results) */
    std::vector<ssf_frame_result> processSequence(const std::vector<const uint8_t*>& rgb, const std::vector<const float*>& depth_m) {
        if (rgb.size() != depth_m.size()) throw std::invalid_argument("processSequence: rgb / depth counts differ");
        std::vector<const void*> r(rgb.begin(), rgb.end()), d(depth_m.begin(), depth_m.end());
        std::vector<ssf_frame_result> out(rgb.size());
        check(ssf_process_sequence(need(), r.data(), d.data(), (int)rgb.size(), 0, out.data()));
        if (!out.empty()) last_ = out.back();
        return out;
    }
#ifdef CV_VERSION
    /* the reference's own signatures (supersurfel_fusion.hpp:75-80); compiled against a cv::Mat test double by
     * tests/test_cpp_wrapper.py (tests/cpp/cv_double.hpp) since OpenCV is not in the build image */
    void processFrame(const cv::Mat& rgb_h, const cv::Mat& depth_h, const float* vo_pose = nullptr, const uint8_t* dynamic = nullptr) {
        const cv::Mat rgb = rgb_h.isContinuous() ? rgb_h : rgb_h.clone(), d = depth_h.isContinuous() ? depth_h : depth_h.clone();
        processFrame(rgb.ptr<uint8_t>(), d.ptr<float>(), vo_pose, dynamic);
    }
    void computeSuperpixelSegIm(cv::Mat& seg_im) {                     /* CV_8UC3, supersurfel_fusion.cu:635-640 */
        seg_im.create(height_, width_, CV_8UC3);
        check(ssf_get_preview_image(need(), seg_im.ptr<uint8_t>()));
    }
    void computeSlantedPlaneIm(cv::Mat& slanted_plane_im) {            /* CV_32FC1, supersurfel_fusion.cu:642-647 */
        slanted_plane_im.create(height_, width_, CV_32FC1);
        check(ssf_get_plane_depth(need(), slanted_plane_im.ptr<float>()));
    }
#endif
    /* the same two images without OpenCV */
    std::vector<uint8_t> getSuperpixelSegIm() {
        std::vector<uint8_t> v((size_t)3 * width_ * height_);
        check(ssf_get_preview_image(need(), v.data()));
        return v;
    }
    std::vector<float> getSlantedPlaneIm() {
        std::vector<float> v((size_t)width_ * height_);
        check(ssf_get_plane_depth(need(), v.data()));
        return v;
    }
    /* getPose(): supersurfel_fusion.hpp:89 -- `const Transform3&`, camera-to-map, valid until the next call on this
     * object (the reference returns a reference to its member; so does this, refreshed from the library) */
    const Transform3& getPose() const {
        float v[12];
        check(ssf_get_pose(need(), v));
        pose_ = transform3_from_rt(v);
        return pose_;
    }
    void setPose(const Transform3& p) {
        float v[12];
        transform3_to_rt(p, v);
        check(ssf_set_pose(need(), v));
    }
    int getnbSupersurfels() const { int n = 0; check(ssf_get_counts(need(), &n, nullptr, nullptr, nullptr)); return n; }
    int getnbVisible() const { int n = 0; check(ssf_get_counts(need(), nullptr, &n, nullptr, nullptr)); return n; }
    int getStamp() const { int s = 0; check(ssf_get_counts(need(), nullptr, nullptr, &s, nullptr)); return s; }
    int getnbSuperpixels() const { int s = 0; check(ssf_get_counts(need(), nullptr, nullptr, nullptr, &s)); return s; }
    /* getModel() / getFrame(): the reference returns device-resident thrust vectors and the node copies
     * [0, nbSupersurfels) to the host (supersurfel_fusion_node.cpp:306-310); here the copy comes back directly */
    HostSupersurfels getModelHost() {
        HostSupersurfels m; m.resize(getnbSupersurfels());
        ssf_surfels v = m.view();
        check(ssf_get_model(need(), 0, m.size, &v));
        return m;
    }
    HostSupersurfels getFrameHost() {
        HostSupersurfels m; m.resize(getnbSuperpixels());
        ssf_surfels v = m.view();
        check(ssf_get_frame(need(), &v));
        return m;
    }
#ifdef SSF_THRUST_VIEW
    /* supersurfel_fusion.hpp:86-87: `const Supersurfels&`, device-resident (views: see Supersurfels above) */
    const Supersurfels& getModel() {
        ssf_surfels v; int n = 0;
        check(ssf_get_model_device(need(), &v, &n));
        model_view_.bind(v, (size_t)n);
        return model_view_;
    }
    const Supersurfels& getFrame() {
        ssf_surfels v; int n = 0;
        check(ssf_get_frame_device(need(), &v, &n));
        frame_view_.bind(v, (size_t)n);
        return frame_view_;
    }
#else
    HostSupersurfels getModel() { return getModelHost(); }
    HostSupersurfels getFrame() { return getFrameHost(); }
#endif
    /* the device views under a name of their own, in both forms (data() / size() only without thrust) */
    const Supersurfels& getModelView() {
        ssf_surfels v; int n = 0;
        check(ssf_get_model_device(need(), &v, &n));
        model_view_.bind(v, (size_t)n);
        return model_view_;
    }
    const Supersurfels& getFrameView() {
        ssf_surfels v; int n = 0;
        check(ssf_get_frame_device(need(), &v, &n));
        frame_view_.bind(v, (size_t)n);
        return frame_view_;
    }
    /* getModel() as the reference returns it: device-resident arrays in the reference's layout (orientations =
     * packed Mat33), n rows, valid until the next call (supersurfel_fusion.hpp:87; the node copies
     * [0, nbSupersurfels) out array by array, supersurfel_fusion_node.cpp:306-310) */
    ssf_surfels getModelDevice(int* n = nullptr) { ssf_surfels v; check(ssf_get_model_device(need(), &v, n)); return v; }
    void exportModel(const std::string& file) { check(ssf_export_model_txt(need(), file.c_str())); }   /* supersurfel_fusion.cu:595-633 */
    /* TPS_RGBD::getIndexImage (TPS_RGBD.hpp:77): the label of every pixel */
    std::vector<int32_t> getIndexImage() {
        std::vector<int32_t> v((size_t)width_ * height_);
        check(ssf_get_index_map(need(), v.data()));
        return v;
    }
    const ssf_frame_result& lastResult() const { return last_; }
    ssf_handle* handle() { return h_; }

private:
    ssf_handle* need() const { if (!h_) throw std::logic_error("SupersurfelFusion: initialize() first"); return h_; }
    void check(int rc) const { if (rc != SSF_OK) throw std::runtime_error(std::string(ssf_last_error(h_))); }
    ssf_handle* h_ = nullptr;
    ssf_frame_result last_{};
    mutable Transform3 pose_{};
    Supersurfels model_view_, frame_view_;        /* always there (layout does not depend on the include order); bound by the thrust form's getModel() / getFrame() and by getModelView() / getFrameView() */
    int width_ = 0, height_ = 0;
    int pipeline_depth_ = 0, extract_batch_ = 1; bool depth_prefilter_ = true;
};

}  /* inline namespace thrust_view / host_copy */
}  /* namespace supersurfel_fusion */
#endif
