// cv_compat.h -- the handful of OpenCV value types the uvo_libraries API is written in.
//
// The reference's public functions (uvo_libraries/include/uvo_libraries/VO_utility.h:96-117) take and return
// cv::Mat, cv::KeyPoint, cv::DMatch and cv::Point2f.  When OpenCV headers are available this header simply aliases
// them (`namespace uvocv = cv`), so the replacement functions have exactly the reference's signatures.  When they
// are not (this build image has no OpenCV), it provides layout-compatible stand-ins with the small subset of the
// cv::Mat interface the shim and its callers use.  Nothing here computes anything: the arithmetic lives in
// libuvo_hip.so behind include/uvo_hip.h.
#pragma once

#if !defined(UVO_NO_OPENCV) && defined(__has_include)
#  if __has_include(<opencv2/core.hpp>)
#    define UVO_HAVE_OPENCV 1
#  endif
#endif

#ifdef UVO_HAVE_OPENCV
#include <opencv2/core.hpp>
namespace uvocv = cv;
#else

#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

namespace uvocv {

// depth codes and type packing as in opencv2/core/hal/interface.h
constexpr int CV_8U = 0, CV_32S = 4, CV_32F = 5, CV_64F = 6;
constexpr int make_type(int depth, int cn) { return (depth & 7) + ((cn - 1) << 3); }
constexpr int CV_8UC1 = make_type(CV_8U, 1), CV_8UC3 = make_type(CV_8U, 3), CV_8UC4 = make_type(CV_8U, 4), CV_32SC1 = make_type(CV_32S, 1), CV_32FC1 = make_type(CV_32F, 1),
              CV_32FC2 = make_type(CV_32F, 2), CV_64FC1 = make_type(CV_64F, 1);

struct Point2f { float x = 0, y = 0; Point2f() = default; Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct Point3f { float x = 0, y = 0, z = 0; };

struct KeyPoint {            // memory layout of cv::KeyPoint (28 bytes)
    Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
};
struct DMatch {              // memory layout of cv::DMatch (16 bytes)
    int queryIdx = -1, trainIdx = -1, imgIdx = -1; float distance = 3.402823466e+38f;
    DMatch() = default;
    DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
};

class Mat {                  // dense, continuous, row-major, reference-counted
public:
    int rows = 0, cols = 0;
    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    void create(int r, int c, int type)
    {
        if (r == rows && c == cols && type == type_ && buf_) return;
        rows = r; cols = c; type_ = type;
        buf_ = std::make_shared<std::vector<uint8_t>>((size_t)r * c * elemSize(), 0);
    }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    static Mat eye(int r, int c, int type)
    {
        Mat m(r, c, type);
        for (int i = 0; i < r && i < c; i++) {
            if (m.depth() == CV_64F) m.at<double>(i, i) = 1; else if (m.depth() == CV_32F) m.at<float>(i, i) = 1;
            else if (m.depth() == CV_32S) m.at<int>(i, i) = 1; else m.at<uint8_t>(i, i) = 1;
        }
        return m;
    }
    int type() const { return type_; }
    int depth() const { return type_ & 7; }
    int channels() const { return (type_ >> 3) + 1; }
    size_t elemSize() const { static const int sz[8] = {1, 1, 2, 2, 4, 4, 8, 2}; return (size_t)sz[depth()] * channels(); }
    size_t step() const { return (size_t)cols * elemSize(); }
    bool empty() const { return !buf_ || rows == 0 || cols == 0; }
    bool isContinuous() const { return true; }
    size_t total() const { return (size_t)rows * cols; }
    void release() { rows = cols = 0; buf_.reset(); }
    uint8_t* data() { return buf_ ? buf_->data() : nullptr; }
    const uint8_t* data() const { return buf_ ? buf_->data() : nullptr; }
    template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data() + (size_t)r * step()); }
    template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data() + (size_t)r * step()); }
    template <class T> T& at(int r, int c) { return ptr<T>(r)[c]; }
    template <class T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }
    template <class T> T& at(int i) { return reinterpret_cast<T*>(data())[i]; }              // continuous: linear index
    template <class T> const T& at(int i) const { return reinterpret_cast<const T*>(data())[i]; }
    Mat clone() const
    {
        Mat m; m.rows = rows; m.cols = cols; m.type_ = type_;
        if (buf_) m.buf_ = std::make_shared<std::vector<uint8_t>>(*buf_);
        return m;
    }
    // cv::Mat::push_back(const Mat&): append rows of the same width and type (adopts them when empty)
    void push_back(const Mat& m)
    {
        if (m.empty()) return;
        if (empty()) { *this = m.clone(); return; }
        if (m.cols != cols || m.type_ != type_) throw std::invalid_argument("Mat::push_back: size/type mismatch");
        if (buf_.use_count() > 1) buf_ = std::make_shared<std::vector<uint8_t>>(*buf_);   // other headers keep the old block
        buf_->insert(buf_->end(), m.buf_->begin(), m.buf_->end());
        rows += m.rows;
    }
    void push_back(int v) { Mat m(1, 1, CV_32SC1); m.at<int>(0, 0) = v; push_back(m); }         // Mat::push_back<int>
    void push_back(double v) { Mat m(1, 1, CV_64FC1); m.at<double>(0, 0) = v; push_back(m); }
    Mat row(int r) const                                                      // a copy, not a view
    {
        Mat m(1, cols, type_);
        std::memcpy(m.data(), data() + (size_t)r * step(), step());
        return m;
    }
    Mat t() const
    {
        Mat m(cols, rows, type_);
        const size_t es = elemSize();
        for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++)
            std::memcpy(m.data() + ((size_t)j * rows + i) * es, data() + ((size_t)i * cols + j) * es, es);
        return m;
    }
private:
    int type_ = CV_64FC1;
    std::shared_ptr<std::vector<uint8_t>> buf_;
};

}  // namespace uvocv
#endif  // UVO_HAVE_OPENCV
