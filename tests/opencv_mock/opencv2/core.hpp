// TEST-ONLY stand-in for <opencv2/core.hpp>: the handful of cv:: types the drop-in headers (include/DynaDetect.h, include/ORBextractor.h)
// and the reference's call sites touch -- Mat (ref-counted, ROI views), InputArray / OutputArray proxies, KeyPoint, Size, Rect, Scalar,
// CV_Assert, the type codes -- with OpenCV's names, member names and call shapes, so that the -DSIND_WITH_OPENCV branch of the shims is
// compiled, linked and RUN without OpenCV (which this image does not have).  It is not OpenCV and implements no image processing.
#ifndef SIND_TEST_OPENCV_CORE_MOCK_HPP
#define SIND_TEST_OPENCV_CORE_MOCK_HPP
#include <cstddef>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_16U 2
#define CV_32F 5
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_16UC1 CV_MAKETYPE(CV_16U, 1)
#define CV_Assert(expr) do { if (!(expr)) throw std::runtime_error(std::string("CV_Assert failed: ") + #expr); } while (0)

namespace cv {
typedef unsigned char uchar;
struct Size { int width = 0, height = 0; Size() {} Size(int w, int h) : width(w), height(h) {} };
struct Rect { int x = 0, y = 0, width = 0, height = 0; Rect() {} Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {} };
struct Scalar { double val[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : val{a, b, c, d} {} };
struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };

class Mat {
public:
    int flags = 0, rows = 0, cols = 0; uchar* data = nullptr; size_t step = 0;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(Size s, int type) { create(s.height, s.width, type); }
    Mat(Size s, int type, const Scalar& v) { create(s.height, s.width, type); setTo(v); }
    Mat(int r, int c, int type, void* ext, size_t stp = 0) : flags(type), rows(r), cols(c), data((uchar*)ext), step(stp ? stp : (size_t)c * esz(type)) {}
    void create(int r, int c, int type) {
        if (data && r == rows && c == cols && type == flags) return;
        flags = type; rows = r; cols = c; step = (size_t)c * esz(type);
        buf = std::make_shared<std::vector<uchar>>((size_t)r * step); data = buf->data();
    }
    void release() { buf.reset(); data = nullptr; rows = cols = 0; step = 0; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return flags; }
    int channels() const { return (flags >> CV_CN_SHIFT) + 1; }
    size_t elemSize() const { return esz(flags); }
    uchar* ptr(int y = 0) { return data + (size_t)y * step; }
    const uchar* ptr(int y = 0) const { return data + (size_t)y * step; }
    template <class T> T& at(int y, int x) { return ((T*)ptr(y))[x]; }
    template <class T> const T& at(int y, int x) const { return ((const T*)ptr(y))[x]; }
    Mat operator()(const Rect& r) const { Mat m(*this); m.data = data + (size_t)r.y * step + (size_t)r.x * esz(flags); m.rows = r.height; m.cols = r.width; return m; }
    Mat rowRange(int a, int b) const { return (*this)(Rect(0, a, cols, b - a)); }
    Mat colRange(int a, int b) const { return (*this)(Rect(a, 0, b - a, rows)); }
    void copyTo(Mat& dst) const { dst.create(rows, cols, flags); for (int y = 0; y < rows; y++) std::memcpy(dst.ptr(y), ptr(y), (size_t)cols * esz(flags)); }
    Mat clone() const { Mat m; copyTo(m); return m; }
    bool isContinuous() const { return step == (size_t)cols * esz(flags); }
    void setTo(const Scalar& v) { const int cn = channels();
        for (int y = 0; y < rows; y++) for (int x = 0; x < cols * cn; x++) { if ((flags & 7) == CV_8U) ptr(y)[x] = (uchar)v.val[x % cn]; else if ((flags & 7) == CV_16U) ((unsigned short*)ptr(y))[x] = (unsigned short)v.val[x % cn]; else ((float*)ptr(y))[x] = (float)v.val[x % cn]; } }
private:
    std::shared_ptr<std::vector<uchar>> buf;             // shared by every header that views the same pixels (like OpenCV's reference count)
    static size_t esz(int type) { const int d = type & 7; return (size_t)((type >> CV_CN_SHIFT) + 1) * (d == CV_8U ? 1 : d == CV_16U ? 2 : 4); }
};

// proxy classes in OpenCV's shape: InputArray = const _InputArray&, OutputArray = const _OutputArray&
class _InputArray {
public:
    _InputArray() {}
    _InputArray(const Mat& m) : in(&m) {}
    Mat getMat() const { return in ? *in : Mat(); }
    bool empty() const { return !in || in->empty(); }
protected:
    const Mat* in = nullptr;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray() {}
    _OutputArray(Mat& m) : _InputArray(m), out(&m) {}
    void create(int rows, int cols, int type) const { CV_Assert(out); out->create(rows, cols, type); }
    void release() const { if (out) out->release(); }
    Mat getMat() const { return out ? *out : Mat(); }
private:
    Mat* out = nullptr;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

class KeyPoint {
public:
    Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
    KeyPoint() {}
    KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
        : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};
}  // namespace cv
#endif
