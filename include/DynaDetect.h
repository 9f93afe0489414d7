// Header-only C++ shim with the reference's class name and signature (ORB_SLAM2/include/DynaDetect.h:95-131) on top of
// the C ABI in sind_hip.h.  With OpenCV present (-DSIND_WITH_OPENCV) the arguments are cv::InputArray / cv::OutputArray
// exactly as in the reference, so Examples/RGB-D/rgbd_tum_noros.cc:106-107,135 compiles unchanged; without OpenCV the same
// class takes sind::Image views (pointer + size + stride).  Link with -lsind_hip.
#ifndef SIND_DYNADETECT_SHIM_H
#define SIND_DYNADETECT_SHIM_H
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "sind_hip.h"
#ifdef SIND_WITH_OPENCV
#include <opencv2/core.hpp>
#endif

namespace sind {
struct Image {            // minimal stand-in for a cv::Mat header: caller-owned pixels
    void* data; int width, height, stride /*bytes*/, channels, elem /*bytes per channel*/;
};
}  // namespace sind

namespace ORB_SLAM2 {

class DynaDetect {
public:
#ifdef SIND_WITH_OPENCV
    DynaDetect(const cv::InputArray& imgLast_, const cv::InputArray& imgLastLast_, float fx_, float fy_, float cx_, float cy_, float depthScale_)
    { cv::Mat a = imgLast_.getMat(), b = imgLastLast_.getMat(); init(a.data, b.data, a.cols, a.rows, (int)a.step, fx_, fy_, cx_, cy_, depthScale_); }
    void DetectDynaArea(const cv::InputArray& img_, const cv::InputArray& imgDepth_, cv::OutputArray& imgDyna_, cv::OutputArray& imgLabel_, int nImg_) {
        cv::Mat img = img_.getMat(), dep = imgDepth_.getMat();
        CV_Assert(img.type() == CV_8UC3 && dep.type() == CV_16UC1 && img.cols == w_ && img.rows == h_);
        imgDyna_.create(h_, w_, CV_8UC1); imgLabel_.create(h_, w_, CV_8UC1);
        cv::Mat dy = imgDyna_.getMat(), lb = imgLabel_.getMat();
        std::vector<uint8_t> d((size_t)w_ * h_), l((size_t)w_ * h_);
        check(sind_dyna_detect(h, img.data, (int)img.step, (const uint16_t*)dep.data, (int)dep.step, d.data(), l.data(), nImg_));
        for (int y = 0; y < h_; y++) { std::memcpy(dy.ptr(y), &d[(size_t)y * w_], w_); std::memcpy(lb.ptr(y), &l[(size_t)y * w_], w_); }
    }
#endif
    DynaDetect(const sind::Image& imgLast_, const sind::Image& imgLastLast_, float fx_, float fy_, float cx_, float cy_, float depthScale_)
    { init(imgLast_.data, imgLastLast_.data, imgLast_.width, imgLast_.height, imgLast_.stride, fx_, fy_, cx_, cy_, depthScale_); }
    // imgDyna_ / imgLabel_: caller-allocated width x height u8 views
    void DetectDynaArea(const sind::Image& img_, const sind::Image& imgDepth_, sind::Image& imgDyna_, sind::Image& imgLabel_, int nImg_) {
        if (imgDyna_.stride != w_ || imgLabel_.stride != w_) throw std::runtime_error("DynaDetect: outputs must be dense u8 images");
        check(sind_dyna_detect(h, (const uint8_t*)img_.data, img_.stride, (const uint16_t*)imgDepth_.data, imgDepth_.stride, (uint8_t*)imgDyna_.data,
                               (uint8_t*)imgLabel_.data, nImg_));
    }
    // the caller-side morphologyEx(imDynaMask, MORPH_DILATE, ellipse 15x15) of rgbd_tum_noros.cc:138, for pipelines without OpenCV
    void DilateForTracking(sind::Image& imgDyna_) { check(sind_dyna_dilate15(h, (uint8_t*)imgDyna_.data)); }
    ~DynaDetect() { sind_dyna_destroy(h); }
    DynaDetect(const DynaDetect&) = delete;
    DynaDetect& operator=(const DynaDetect&) = delete;

private:
    sind_dyna* h = nullptr; int w_ = 0, h_ = 0;
    static void check(int rc) { if (rc != SIND_OK) throw std::runtime_error(std::string("sind_dyna: ") + sind_last_error()); }
    void init(const void* last, const void* lastlast, int w, int hh, int stride, float fx, float fy, float cx, float cy, float ds) {
        w_ = w; h_ = hh;
        check(sind_dyna_create(w, hh, fx, fy, cx, cy, ds, 0, &h));
        check(sind_dyna_prime(h, (const uint8_t*)last, (const uint8_t*)lastlast, stride));
    }
};

}  // namespace ORB_SLAM2
#endif
