// Header-only C++ shim with the reference's class name, constructor and call operator
// (ORB_SLAM2/include/ORBextractor.h:54-88) on top of the C ABI in sind_hip.h, so that src/Frame.cc:308
// `(*mpORBextractorLeft)(im, imDynaMask, mvKeys, mDescriptors)` and the getters used by src/Frame.cc:69-75 keep working.
// Build with -DSIND_WITH_OPENCV for the cv:: types (operator() on cv::InputArray / OutputArray, the public mvImagePyramid); link with -lsind_hip.
// tests/cpp/boundary_callsites.cpp compiles and runs that branch against a test-only stand-in for <opencv2/core.hpp>.
#ifndef SIND_ORBEXTRACTOR_SHIM_H
#define SIND_ORBEXTRACTOR_SHIM_H
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "sind_hip.h"
#ifdef SIND_WITH_OPENCV
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#endif

namespace ORB_SLAM2 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
        : nlevels_(nlevels), scaleFactor_(scaleFactor), cap_(2 * nfeatures + 256) {
        check(sind_orb_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, 0, &h));
        scale_.resize(nlevels); inv_scale_.resize(nlevels); sigma2_.resize(nlevels); inv_sigma2_.resize(nlevels);
    }
    ~ORBextractor() { sind_orb_destroy(h); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // plain-pointer form: gray/mask are dense u8 images; returns keypoints and n x 32 descriptors
    void operator()(const uint8_t* gray, int width, int height, int stride, const uint8_t* mask_or_null, int mask_stride,
                    std::vector<sind_keypoint>& keypoints, std::vector<uint8_t>& descriptors) {
        keypoints.resize(cap_); descriptors.resize((size_t)cap_ * 32); int n = 0;
        check(sind_orb_extract(h, gray, width, height, stride, mask_or_null, mask_stride, keypoints.data(), cap_, &n, descriptors.data()));
        keypoints.resize(n); descriptors.resize((size_t)n * 32); tables_ready_ = false;
    }
#ifdef SIND_WITH_OPENCV
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors) {
        if (image.empty()) return;
        cv::Mat im = image.getMat(), mk = mask.empty() ? cv::Mat() : mask.getMat();
        CV_Assert(im.type() == CV_8UC1);
        std::vector<sind_keypoint> k; std::vector<uint8_t> d;
        (*this)(im.data, im.cols, im.rows, (int)im.step, mk.empty() ? nullptr : mk.data, mk.empty() ? 0 : (int)mk.step, k, d);
        if (keepImagePyramid) {
            mvImagePyramid.resize(nlevels_); padded_.resize(nlevels_);
            for (int l = 0; l < nlevels_; l++) {
                int w = 0, hh = 0; check(sind_orb_pyramid(h, 0, l, nullptr, &w, &hh));
                padded_[l].create(hh + 38, w + 38, CV_8UC1);
                check(sind_orb_pyramid(h, 0, l, padded_[l].data, &w, &hh));
                mvImagePyramid[l] = padded_[l](cv::Rect(19, 19, w, hh));
            }
        }
        keypoints.clear(); keypoints.reserve(k.size());
        for (const sind_keypoint& p : k) keypoints.emplace_back(p.x, p.y, p.size, p.angle, p.response, p.octave, p.class_id);
        if (k.empty()) { descriptors.release(); return; }
        descriptors.create((int)k.size(), 32, CV_8U);
        std::memcpy(descriptors.getMat().data, d.data(), d.size());
    }
    // The reference's public pyramid (include/ORBextractor.h:88), read by Frame::ComputeStereoMatches (src/Frame.cc:544, 634-651):
    // level l is the ROI at (19, 19) of a padded image with the 19-px REFLECT_101 border, exactly how ComputePyramid leaves it
    // (src/ORBextractor.cc:1166-1191).  Refreshed after every call while keepImagePyramid is set (one 1.2 MB copy from the GPU per frame;
    // the RGB-D path never reads it and may switch it off).
    std::vector<cv::Mat> mvImagePyramid;
    bool keepImagePyramid = true;
#endif
    int GetLevels() { return nlevels_; }
    float GetScaleFactor() { return scaleFactor_; }
    std::vector<float> GetScaleFactors() { tables(); return scale_; }
    std::vector<float> GetInverseScaleFactors() { tables(); return inv_scale_; }
    std::vector<float> GetScaleSigmaSquares() { tables(); return sigma2_; }
    std::vector<float> GetInverseScaleSigmaSquares() { tables(); return inv_sigma2_; }
    // mvImagePyramid[level] of the last call, with its 19-px border: out must hold (w+38)*(h+38) bytes
    void ImagePyramidLevel(int level, std::vector<uint8_t>& out, int& w, int& hgt) {
        check(sind_orb_pyramid(h, 0, level, nullptr, &w, &hgt)); out.resize((size_t)(w + 38) * (hgt + 38)); check(sind_orb_pyramid(h, 0, level, out.data(), &w, &hgt));
    }

private:
    sind_orb* h = nullptr; int nlevels_; float scaleFactor_; int cap_; bool tables_ready_ = false;
    std::vector<float> scale_, inv_scale_, sigma2_, inv_sigma2_;
#ifdef SIND_WITH_OPENCV
    std::vector<cv::Mat> padded_;
#endif
    static void check(int rc) { if (rc != SIND_OK) throw std::runtime_error(std::string("sind_orb: ") + sind_last_error()); }
    void tables() { if (tables_ready_) return; check(sind_orb_tables(h, scale_.data(), inv_scale_.data(), sigma2_.data(), inv_sigma2_.data(), nullptr, nullptr)); tables_ready_ = true; }
};

}  // namespace ORB_SLAM2
#endif
