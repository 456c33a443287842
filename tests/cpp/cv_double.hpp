// cv_double.hpp -- a test double of the few cv::Mat members include/ssf.hpp touches, so that its cv::Mat overloads
// (the reference's own signatures, supersurfel_fusion.hpp:75-80) are compiled and run in an image without OpenCV.
// Test infrastructure only: a node includes <opencv2/core.hpp> instead.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>
#define CV_VERSION "test-double"
#define CV_8UC3 16
#define CV_32FC1 5
namespace cv {
class Mat {
public:
    int rows = 0, cols = 0;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    void create(int r, int c, int type) {
        rows = r; cols = c; type_ = type;
        buf_ = std::make_shared<std::vector<unsigned char>>((size_t)r * c * (type == CV_8UC3 ? 3 : 4));
    }
    bool isContinuous() const { return continuous_; }
    Mat clone() const { Mat m; m.rows = rows; m.cols = cols; m.type_ = type_; m.buf_ = std::make_shared<std::vector<unsigned char>>(*buf_); return m; }
    template <typename T> T* ptr() { return reinterpret_cast<T*>(buf_->data()); }
    template <typename T> const T* ptr() const { return reinterpret_cast<const T*>(buf_->data()); }
    void pretendStrided() { continuous_ = false; }      // makes processFrame take its clone() path
    int type() const { return type_; }
private:
    int type_ = 0; bool continuous_ = true;
    std::shared_ptr<std::vector<unsigned char>> buf_;
};
}  // namespace cv
