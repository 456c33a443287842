// The reference's nodes call SupersurfelFusion::initialize with 29 positional arguments
// (node/supersurfel_fusion_node.cpp:256-284; node/supersurfel_fusion_rgbd_benchmark_node.cpp makes the same call):
// the 21 hot-path parameters followed by the sparse-VO / loop-closure / MOD ones.  This program makes that call against
// include/ssf.hpp with the node's parameter values (node defaults, supersurfel_fusion_node.cpp:224-252) and checks that
// the trailing eight are accepted and IGNORED -- in particular that none of them lands in a library knob (round 2's header
// re-used those positions for pipeline_depth / extract_batch / depth_prefilter).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ssf.hpp"

int main(int argc, char** argv) {
    if (argc < 9) return 2;
    const int W = std::atoi(argv[1]), H = std::atoi(argv[2]), n = std::atoi(argv[3]);
    std::FILE* f = std::fopen(argv[4], "rb");
    if (!f) return 3;
    std::vector<std::vector<uint8_t>> rgb(n, std::vector<uint8_t>((size_t)3 * W * H));
    std::vector<std::vector<float>> depth(n, std::vector<float>((size_t)W * H));
    for (int k = 0; k < n; k++) {
        if (std::fread(rgb[k].data(), 1, rgb[k].size(), f) != rgb[k].size()) return 4;
        if (std::fread(depth[k].data(), 4, depth[k].size(), f) != depth[k].size()) return 4;
    }
    std::fclose(f);
    supersurfel_fusion::CamParam cam_param; cam_param.width = W; cam_param.height = H;
    cam_param.fx = (float)std::atof(argv[5]); cam_param.fy = (float)std::atof(argv[6]);
    cam_param.cx = (float)std::atof(argv[7]); cam_param.cy = (float)std::atof(argv[8]);
    // the node's members, with the node's defaults
    int cell_size = 16, seg_iter = 10, nb_samples = 16, filter_iter = 4, delta_t = 10, nb_supersurfels_max = 50000, icp_iter = 10;
    float lambda_pos = 50.f, lambda_bound = 1000.f, lambda_size = 10000.f, lambda_disp = 1000000.f, thresh_disp = 0.0001f;
    bool seg_use_ransac = true;
    float filter_alpha = 0.1f, filter_beta = 1.0f, filter_threshold = 0.05f, range_min = 0.2f, range_max = 5.0f, conf_thresh_scale = 5.f;
    double icp_cov_thresh = 0.04;
    int nb_features = 2000; float features_scale_factor = 1.2f; int features_nb_levels = 8, ini_th_fast = 20, min_th_fast = 7, untracked_threshold = 10;
    bool enable_loop_closure = true, enable_mod = true;
    float confThresh = cell_size * cell_size * conf_thresh_scale;
    try {
        supersurfel_fusion::SupersurfelFusion ssf;
        ssf.initialize(cam_param,
                       cell_size,
                       lambda_pos,
                       lambda_bound,
                       lambda_size,
                       lambda_disp,
                       thresh_disp,
                       seg_iter,
                       seg_use_ransac,
                       nb_samples,
                       filter_iter,
                       filter_alpha,
                       filter_beta,
                       filter_threshold,
                       range_min,
                       range_max,
                       delta_t,
                       confThresh,
                       nb_supersurfels_max,
                       icp_iter,
                       icp_cov_thresh,
                       nb_features,
                       features_scale_factor,
                       features_nb_levels,
                       ini_th_fast,
                       min_th_fast,
                       untracked_threshold,
                       enable_loop_closure,
                       enable_mod);
        // the same 21 path parameters through the POD configuration: must give the same frames, bit for bit
        ssf_config c; ssf_default_config(&c);
        c.width = W; c.height = H; c.fx = cam_param.fx; c.fy = cam_param.fy; c.cx = cam_param.cx; c.cy = cam_param.cy;
        c.delta_t = delta_t; c.conf_thresh = confThresh;
        const bool defaults_are_the_references = c.cell_size == cell_size && c.lambda_pos == lambda_pos && c.lambda_bound == lambda_bound &&
            c.lambda_size == lambda_size && c.lambda_disp == lambda_disp && c.thresh_disp == thresh_disp && c.seg_iter == seg_iter &&
            c.seg_use_ransac == 1 && c.nb_samples == nb_samples && c.filter_iter == filter_iter && c.filter_alpha == filter_alpha &&
            c.filter_beta == filter_beta && c.filter_threshold == filter_threshold && c.range_min == range_min && c.range_max == range_max &&
            c.nb_supersurfels_max == nb_supersurfels_max && c.icp_iter == icp_iter && c.icp_cov_thresh == icp_cov_thresh &&
            c.pipeline_depth == 0 && c.extract_batch == 1 && c.depth_prefilter == 1;
        std::printf("defaults_are_the_references %d\n", defaults_are_the_references ? 1 : 0);
        supersurfel_fusion::SupersurfelFusion pod;
        pod.initialize(c);
        bool same = true;
        for (int k = 0; k < n; k++) {
            ssf.processFrame(rgb[k].data(), depth[k].data());
            pod.processFrame(rgb[k].data(), depth[k].data());
            const supersurfel_fusion::Transform3 a = ssf.getPose(), b = pod.getPose();
            float va[12], vb[12];
            supersurfel_fusion::transform3_to_rt(a, va); supersurfel_fusion::transform3_to_rt(b, vb);
            for (int i = 0; i < 12; i++) same = same && va[i] == vb[i];
            same = same && ssf.getnbSupersurfels() == pod.getnbSupersurfels() && ssf.getnbVisible() == pod.getnbVisible();
        }
        std::printf("node_call_equals_pod_config %d n=%d\n", same ? 1 : 0, ssf.getnbSupersurfels());
    } catch (const std::exception& e) { std::printf("exception %s\n", e.what()); return 1; }
    return 0;
}
