// The pose block of the reference's two nodes, as written there (the five lines between the markers are the callers' own:
// node/supersurfel_fusion_node.cpp:87-91 and node/supersurfel_fusion_rgbd_benchmark_node.cpp:616-620 are the same text),
// compiled against include/ssf.hpp: `Transform3` must be the reference's type (matrix_types.h:33-42: Mat33 R of three float3
// rows, float3 t), visible unqualified under `using namespace supersurfel_fusion` (node/main.cpp:23), and getPose() must
// hand back what ssf_get_pose holds.  tf:: comes from a test double (tf_double.hpp); frames from a file as in
// wrapper_smoke.cpp.  Prints, per frame, whether the nine + three values the node would broadcast equal the C ABI's.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "ssf.hpp"
#include "tf_double.hpp"

using namespace supersurfel_fusion;

// layout of the reference's types: 12 packed floats, rows then t (what a node memcpy's or hands to a kernel)
static_assert(sizeof(float3) == 12 && sizeof(Mat33) == 36 && sizeof(Transform3) == 48 && sizeof(Cov3) == 24, "matrix_types.h layout");
static_assert(std::is_same<decltype(std::declval<const SupersurfelFusion&>().getPose()), const Transform3&>::value,
              "getPose() returns const Transform3& (supersurfel_fusion.hpp:89)");

int main(int argc, char** argv) {
    if (argc < 9) return 2;
    const int W = std::atoi(argv[1]), H = std::atoi(argv[2]), n = std::atoi(argv[3]);
    std::FILE* f = std::fopen(argv[4], "rb");
    if (!f) return 3;
    std::vector<std::vector<uint8_t>> rgbs(n, std::vector<uint8_t>((size_t)3 * W * H));
    std::vector<std::vector<float>> depths(n, std::vector<float>((size_t)W * H));
    for (int k = 0; k < n; k++) {
        if (std::fread(rgbs[k].data(), 1, rgbs[k].size(), f) != rgbs[k].size()) return 4;
        if (std::fread(depths[k].data(), 4, depths[k].size(), f) != depths[k].size()) return 4;
    }
    std::fclose(f);
    CamParam cam; cam.width = W; cam.height = H;
    cam.fx = (float)std::atof(argv[5]); cam.fy = (float)std::atof(argv[6]); cam.cx = (float)std::atof(argv[7]); cam.cy = (float)std::atof(argv[8]);
    try {
        SupersurfelFusion ssf;
        // the benchmark launch file's column (filter_iter 3, conf_thresh 16*16*10, icp_cov_thresh 0.05), positionally as the nodes pass it
        ssf.initialize(cam, 16, 10.f, 1000.f, 1000.f, 1e8f, 1e-4f, 10, true, 16, 3, 0.1f, 1.0f, 0.05f, 0.2f, 5.0f, 20, 2560.f, 50000, 10, 0.05);
        for (int k = 0; k < n; k++) {
            const uint8_t* rgb = rgbs[k].data(); const float* depth = depths[k].data();
            ssf.processFrame(rgb, depth);

            // ---- the reference nodes' lines, verbatim ----
            Transform3 pose = ssf.getPose();
            tf::Transform opt_to_map(tf::Matrix3x3(pose.R.rows[0].x, pose.R.rows[0].y, pose.R.rows[0].z,
                                                   pose.R.rows[1].x, pose.R.rows[1].y, pose.R.rows[1].z,
                                                   pose.R.rows[2].x, pose.R.rows[2].y, pose.R.rows[2].z),
                                     tf::Vector3(pose.t.x, pose.t.y, pose.t.z));
            // ----------------------------------------------

            float v[12];
            if (ssf_get_pose(ssf.handle(), v) != SSF_OK) return 5;
            bool same = true;
            for (int r = 0; r < 3; r++) {
                const tf::Vector3& row = opt_to_map.getBasis().getRow(r);
                same = same && row.x() == (double)v[3 * r] && row.y() == (double)v[3 * r + 1] && row.z() == (double)v[3 * r + 2];
            }
            const tf::Vector3& o = opt_to_map.getOrigin();
            same = same && o.x() == (double)v[9] && o.y() == (double)v[10] && o.z() == (double)v[11];
            // setPose(getPose()) is the identity on the library's pose
            ssf.setPose(pose);
            float w[12];
            if (ssf_get_pose(ssf.handle(), w) != SSF_OK) return 5;
            for (int i = 0; i < 12; i++) same = same && v[i] == w[i];
            std::printf("frame %d node_pose_equals_abi %d pose", k, same ? 1 : 0);
            for (int i = 0; i < 12; i++) std::printf(" %.9g", v[i]);
            std::printf("\n");
        }
    } catch (const std::exception& e) { std::printf("exception %s\n", e.what()); return 1; }
    return 0;
}
