// Drives include/ssf.hpp like the reference's benchmark node drives SupersurfelFusion: initialize, a few processFrame
// calls on raw frames read from a file, getPose / getnbSupersurfels / getModel, then the same frames through
// processSequence on a second object.  Prints what tests/test_cpp_wrapper.py compares with the Python binding.
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
    using namespace supersurfel_fusion;
    CamParam cam; cam.width = W; cam.height = H;
    cam.fx = (float)std::atof(argv[5]); cam.fy = (float)std::atof(argv[6]); cam.cx = (float)std::atof(argv[7]); cam.cy = (float)std::atof(argv[8]);
    try {
        SupersurfelFusion a;
        if (a.isInitialized()) return 5;
        a.initialize(cam, 16, 10.f, 1000.f, 1000.f, 1e8f);
        for (int k = 0; k < n; k++) {
            a.processFrame(rgb[k].data(), depth[k].data());
            const Transform3 p = a.getPose();
            std::printf("frame %d n=%d vis=%d stamp=%d icp=%d/%d pose", k, a.getnbSupersurfels(), a.getnbVisible(), a.getStamp(),
                        a.lastResult().icp_valid, a.lastResult().icp_iters);
            for (int r = 0; r < 3; r++) std::printf(" %.9g %.9g %.9g", p.R.rows[r].x, p.R.rows[r].y, p.R.rows[r].z);
            std::printf(" %.9g %.9g %.9g", p.t.x, p.t.y, p.t.z);
            std::printf("\n");
        }
        HostSupersurfels m = a.getModel();
        double s = 0; for (float v : m.positions) s += v;
        std::printf("model %d possum %.9g labels %zu\n", m.size, s, a.getIndexImage().size());
        SupersurfelFusion b;
        b.initialize(cam, 16, 10.f, 1000.f, 1000.f, 1e8f);
        std::vector<const uint8_t*> pr; std::vector<const float*> pd;
        for (int k = 0; k < n; k++) { pr.push_back(rgb[k].data()); pd.push_back(depth[k].data()); }
        const std::vector<ssf_frame_result> res = b.processSequence(pr, pd);
        std::printf("sequence n=%d last_n=%d\n", (int)res.size(), res.back().n_model);
        bool same = b.getnbSupersurfels() == a.getnbSupersurfels();
        const Transform3 pa = a.getPose(), pb = b.getPose();
        float va[12], vb[12];
        transform3_to_rt(pa, va); transform3_to_rt(pb, vb);
        for (int i = 0; i < 12; i++) same = same && va[i] == vb[i];
        std::printf("sequence_equals_frames %d\n", same ? 1 : 0);
        SupersurfelFusion c;
        try { c.getPose(); return 6; } catch (const std::logic_error&) { std::printf("uninitialised_throws 1\n"); }
    } catch (const std::exception& e) { std::printf("exception %s\n", e.what()); return 1; }
    return 0;
}
