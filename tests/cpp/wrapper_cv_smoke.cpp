// The cv::Mat surface of include/ssf.hpp (processFrame(cv::Mat, cv::Mat), computeSuperpixelSegIm, computeSlantedPlaneIm:
// the reference's signatures, supersurfel_fusion.hpp:75-80) against the cv::Mat test double, plus getModelDevice and the
// setDepthPrefilter switch.  Prints checksums that tests/test_cpp_wrapper.py compares with the Python mirror.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "cv_double.hpp"
#include "ssf.hpp"

int main(int argc, char** argv) {
    if (argc < 9) return 2;
    const int W = std::atoi(argv[1]), H = std::atoi(argv[2]), n = std::atoi(argv[3]);
    std::FILE* f = std::fopen(argv[4], "rb");
    if (!f) return 3;
    using namespace supersurfel_fusion;
    CamParam cam; cam.width = W; cam.height = H;
    cam.fx = (float)std::atof(argv[5]); cam.fy = (float)std::atof(argv[6]); cam.cx = (float)std::atof(argv[7]); cam.cy = (float)std::atof(argv[8]);
    try {
        SupersurfelFusion a;
        // prefilter explicitly OFF (a named setter: initialize() carries the reference's parameter list only), everything
        // else as wrapper_smoke.cpp
        a.setDepthPrefilter(false);
        a.initialize(cam, 16, 10.f, 1000.f, 1000.f, 1e8f);
        for (int k = 0; k < n; k++) {
            cv::Mat rgb(H, W, CV_8UC3), depth(H, W, CV_32FC1);
            if (std::fread(rgb.ptr<uint8_t>(), 1, (size_t)3 * W * H, f) != (size_t)3 * W * H) return 4;
            if (std::fread(depth.ptr<float>(), 4, (size_t)W * H, f) != (size_t)W * H) return 4;
            if (k == 1) rgb.pretendStrided();
            a.processFrame(rgb, depth);
        }
        std::fclose(f);
        cv::Mat seg, plane;
        a.computeSuperpixelSegIm(seg);
        a.computeSlantedPlaneIm(plane);
        unsigned long long s1 = 0; double s2 = 0;
        for (size_t i = 0; i < (size_t)3 * W * H; i++) s1 = s1 * 1315423911ull + seg.ptr<uint8_t>()[i];
        int finite = 0;
        for (size_t i = 0; i < (size_t)W * H; i++) { const float v = plane.ptr<float>()[i]; if (v == v && v > 0.f && v < 100.f) { s2 += v; finite++; } }
        std::printf("seg %dx%d type=%d hash=%llu\n", seg.cols, seg.rows, seg.type(), s1);
        std::printf("plane %dx%d type=%d finite=%d sum=%.9g\n", plane.cols, plane.rows, plane.type(), finite, s2);
        std::printf("same_as_raw %d\n", (a.getSuperpixelSegIm().size() == (size_t)3 * W * H && a.getSlantedPlaneIm()[W + 1] == plane.ptr<float>()[W + 1]) ? 1 : 0);
        int nm = 0;
        const ssf_surfels dev = a.getModelDevice(&nm);
        std::printf("model_device n=%d ptrs=%d\n", nm, (dev.positions && dev.orientations && dev.confidences) ? 1 : 0);
        std::printf("n=%d vis=%d\n", a.getnbSupersurfels(), a.getnbVisible());
    } catch (const std::exception& e) { std::printf("exception %s\n", e.what()); return 1; }
    return 0;
}
