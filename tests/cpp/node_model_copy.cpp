// The reference nodes' copy-out of the model and of the frame, as written there, compiled by hipcc against rocThrust and
// include/ssf.hpp (thrust included FIRST: getModel() / getFrame() are then `const Supersurfels&` views of device arrays, as
// supersurfel_fusion.hpp:86-87).  The lines between the markers are the callers' own:
//   node/supersurfel_fusion_node.cpp:306-310 (publishModelMarker: range form), :423-427 (publishFrameMarker: whole-array form),
//   :688-690 / node/supersurfel_fusion_rgbd_benchmark_node.cpp:189-193 (whole-array form on the model).
// Checks every copied array against the C ABI's host copy (ssf_get_model / ssf_get_frame), bit for bit.
#include <thrust/host_vector.h>
#include <thrust/device_vector.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ssf.hpp"

using namespace supersurfel_fusion;
#ifndef SSF_THRUST_VIEW
#error "thrust was included first: ssf.hpp must offer the device views"
#endif
static_assert(std::is_same<decltype(std::declval<SupersurfelFusion&>().getModel()), const Supersurfels&>::value, "getModel() -> const Supersurfels&");
static_assert(std::is_same<decltype(std::declval<SupersurfelFusion&>().getFrame()), const Supersurfels&>::value, "getFrame() -> const Supersurfels&");

template <typename T> static bool same(const thrust::host_vector<T>& v, const void* ref, size_t n) {
    return v.size() == n && (n == 0 || std::memcmp(&v[0], ref, n * sizeof(T)) == 0);
}

int main(int argc, char** argv) {
    if (argc < 9) return 2;
    const int W = std::atoi(argv[1]), H = std::atoi(argv[2]), nfr = std::atoi(argv[3]);
    std::FILE* f = std::fopen(argv[4], "rb");
    if (!f) return 3;
    std::vector<std::vector<uint8_t>> rgbs(nfr, std::vector<uint8_t>((size_t)3 * W * H));
    std::vector<std::vector<float>> depths(nfr, std::vector<float>((size_t)W * H));
    for (int k = 0; k < nfr; k++) {
        if (std::fread(rgbs[k].data(), 1, rgbs[k].size(), f) != rgbs[k].size()) return 4;
        if (std::fread(depths[k].data(), 4, depths[k].size(), f) != depths[k].size()) return 4;
    }
    std::fclose(f);
    CamParam cam; cam.width = W; cam.height = H;
    cam.fx = (float)std::atof(argv[5]); cam.fy = (float)std::atof(argv[6]); cam.cx = (float)std::atof(argv[7]); cam.cy = (float)std::atof(argv[8]);
    try {
        SupersurfelFusion ssf;
        ssf.initialize(cam, 16, 10.f, 1000.f, 1000.f, 1e8f, 1e-4f, 10, true, 16, 3, 0.1f, 1.0f, 0.05f, 0.2f, 5.0f, 20, 2560.f, 50000, 10, 0.05);
        for (int k = 0; k < nfr; k++) ssf.processFrame(rgbs[k].data(), depths[k].data());
        bool ok = true;
        {
            // ---- node/supersurfel_fusion_node.cpp:306-310, verbatim ----
            thrust::host_vector<float3> positions(ssf.getModel().positions.begin(), ssf.getModel().positions.begin() + ssf.getnbSupersurfels());
            thrust::host_vector<float3> colors(ssf.getModel().colors.begin(), ssf.getModel().colors.begin() + ssf.getnbSupersurfels());
            thrust::host_vector<Mat33> orientations(ssf.getModel().orientations.begin(), ssf.getModel().orientations.begin() + ssf.getnbSupersurfels());
            thrust::host_vector<float2> dims(ssf.getModel().dims.begin(), ssf.getModel().dims.begin() + ssf.getnbSupersurfels());
            thrust::host_vector<float> confidences(ssf.getModel().confidences.begin(), ssf.getModel().confidences.begin() + ssf.getnbSupersurfels());
            // --------------------------------------------------------------
            const HostSupersurfels ref = ssf.getModelHost();
            const size_t n = (size_t)ssf.getnbSupersurfels();
            ok = ok && n > 0 && same(positions, ref.positions.data(), n) && same(colors, ref.colors.data(), n) &&
                 same(orientations, ref.orientations.data(), n) && same(dims, ref.dims.data(), n) && same(confidences, ref.confidences.data(), n);
            std::printf("model_range_form %d n=%zu\n", ok ? 1 : 0, n);
        }
        {
            // ---- node/supersurfel_fusion_rgbd_benchmark_node.cpp:189-193, verbatim ----
            thrust::host_vector<float3> positions(ssf.getModel().positions);
            thrust::host_vector<float3> colors(ssf.getModel().colors);
            thrust::host_vector<Mat33> orientations(ssf.getModel().orientations);
            thrust::host_vector<float2> dims(ssf.getModel().dims);
            thrust::host_vector<float> confidences(ssf.getModel().confidences);
            // -----------------------------------------------------------------------------
            const HostSupersurfels ref = ssf.getModelHost();
            const size_t n = (size_t)ssf.getnbSupersurfels();
            const bool ok2 = same(positions, ref.positions.data(), n) && same(colors, ref.colors.data(), n) && same(orientations, ref.orientations.data(), n) &&
                             same(dims, ref.dims.data(), n) && same(confidences, ref.confidences.data(), n);
            std::printf("model_whole_array_form %d\n", ok2 ? 1 : 0);
            ok = ok && ok2;
        }
        {
            // ---- node/supersurfel_fusion_node.cpp:423-427, verbatim ----
            thrust::host_vector<float3> positions(ssf.getFrame().positions);
            thrust::host_vector<float3> colors(ssf.getFrame().colors);
            thrust::host_vector<Mat33> orientations(ssf.getFrame().orientations);
            thrust::host_vector<float2> dims(ssf.getFrame().dims);
            thrust::host_vector<float> confidences(ssf.getFrame().confidences);
            // --------------------------------------------------------------
            const HostSupersurfels ref = ssf.getFrameHost();
            const size_t n = (size_t)ssf.getnbSuperpixels();
            // (an invalid frame supersurfel carries confidence -1 and unspecified other fields: compared where valid)
            bool ok3 = positions.size() == n && same(confidences, ref.confidences.data(), n);
            size_t nvalid = 0;
            for (size_t i = 0; i < n && ok3; i++) {
                if (!(ref.confidences[i] > 0.f)) continue;
                nvalid++;
                ok3 = std::memcmp(&positions[i], &ref.positions[3 * i], 12) == 0 && std::memcmp(&colors[i], &ref.colors[3 * i], 12) == 0 &&
                      std::memcmp(&orientations[i], &ref.orientations[9 * i], 36) == 0 && std::memcmp(&dims[i], &ref.dims[2 * i], 8) == 0;
            }
            std::printf("frame_whole_array_form %d valid=%zu of %zu\n", ok3 ? 1 : 0, nvalid, n);
            ok = ok && ok3 && nvalid > 0;
        }
        return ok ? 0 : 1;
    } catch (const std::exception& e) { std::printf("exception %s\n", e.what()); return 1; }
}
