// The reference's own call sites of the two drop-in classes, kept in their original shape (same expressions, same argument types), compiled
// with -DSIND_WITH_OPENCV against include/DynaDetect.h + include/ORBextractor.h:
//   Examples/RGB-D/rgbd_tum_noros.cc:100-107, 135, 138     DynaDetect construction through std::make_shared, DetectDynaArea on cv::Mat
//   src/Frame.cc:308-323 (ExtractORB2 / ExtractORB)         (*mpORBextractorLeft)(im, imDynaMask, mvKeys, mDescriptors) and the cv::Mat() form
//   src/Frame.cc:69-75                                      GetLevels / GetScaleFactor / GetScaleFactors / ... getters
//   src/Frame.cc:544, 634-651 (ComputeStereoMatches)        the public mvImagePyramid: [0].rows, [octave].rowRange(..).colRange(..), .cols
// <opencv2/core.hpp> is the test-only stand-in of tests/opencv_mock (this image has no OpenCV); it is functional enough to RUN:
//   boundary_callsites <in.raw> <out.raw> fx fy cx cy depthFactor      (in.raw as in examples/rgbd_tum_noros_shim.cpp)
//   out: per frame  imDynaMask (dilated? no: as returned) u8[h*w], imLabel u8[h*w], int32 nkp, nkp x {x, y, size, angle, response, octave, class_id},
//        nkp x 32 descriptor bytes; after the last frame: int32 nlevels, per level int32 w, h then the w x h pixels of mvImagePyramid[level],
//        then the 11 x 11 patch IL of the stereo matcher's shape taken from level 1.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>
#include <opencv2/core.hpp>
#include "DynaDetect.h"
#include "ORBextractor.h"

using namespace std;

// src/Frame.cc:308-323, verbatim shape
struct FrameLike {
    ORB_SLAM2::ORBextractor *mpORBextractorLeft, *mpORBextractorRight;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight; cv::Mat mDescriptors, mDescriptorsRight;
    void ExtractORB2(int flag, const cv::Mat& im, const cv::Mat& imDynaMask) {
        if (flag == 0) {
            if (!imDynaMask.empty()) (*mpORBextractorLeft)(im, imDynaMask, mvKeys, mDescriptors);
            else (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);
        } else (*mpORBextractorRight)(im, cv::Mat(), mvKeysRight, mDescriptorsRight);
    }
};

int main(int argc, char** argv) {
    if (argc != 8) { fprintf(stderr, "usage: %s in out fx fy cx cy depthFactor\n", argv[0]); return 2; }
    FILE* fi = fopen(argv[1], "rb"); if (!fi) { perror(argv[1]); return 2; }
    int hdr[3]; if (fread(hdr, 4, 3, fi) != 3) return 2;
    const int nImages = hdr[0], w = hdr[1], h = hdr[2]; const size_t np = (size_t)w * h;
    vector<unsigned char> bgr(np * 3 * nImages); vector<unsigned short> depth(np * nImages);
    if (fread(bgr.data(), 1, bgr.size(), fi) != bgr.size() || fread(depth.data(), 2, depth.size(), fi) != depth.size()) return 2;
    fclose(fi);
    const float fx = (float)atof(argv[3]), fy = (float)atof(argv[4]), cx = (float)atof(argv[5]), cy = (float)atof(argv[6]), depthScale = (float)atof(argv[7]);
    FILE* fo = fopen(argv[2], "wb"); if (!fo) { perror(argv[2]); return 2; }
    try {
        // rgbd_tum_noros.cc:100-107
        cv::Mat imDynaMask(cv::Size(w, h), CV_8UC1, cv::Scalar(0));
        cv::Mat imLabel(cv::Size(w, h), CV_8UC1, cv::Scalar(0));
        cv::Mat imgLast(h, w, CV_8UC3, bgr.data());                // cv::imread(..., -1) in the reference
        cv::Mat imgLastLast;
        imgLast.copyTo(imgLastLast);
        std::shared_ptr<ORB_SLAM2::DynaDetect> detertor = std::make_shared<ORB_SLAM2::DynaDetect>
                                                          (imgLast, imgLastLast, fx, fy, cx, cy, depthScale);
        ORB_SLAM2::ORBextractor left(1500, 1.2f, 8, 15, 5), right(1500, 1.2f, 8, 15, 5);
        FrameLike frame{&left, &right};
        // src/Frame.cc:69-75
        const int mnScaleLevels = frame.mpORBextractorLeft->GetLevels();
        const float mfScaleFactor = frame.mpORBextractorLeft->GetScaleFactor();
        const vector<float> mvScaleFactors = frame.mpORBextractorLeft->GetScaleFactors(), mvInvScaleFactors = frame.mpORBextractorLeft->GetInverseScaleFactors();
        const vector<float> mvLevelSigma2 = frame.mpORBextractorLeft->GetScaleSigmaSquares(), mvInvLevelSigma2 = frame.mpORBextractorLeft->GetInverseScaleSigmaSquares();
        if (mnScaleLevels != 8 || mfScaleFactor != 1.2f || mvScaleFactors.size() != 8 || mvInvScaleFactors.size() != 8 || mvLevelSigma2.size() != 8 || mvInvLevelSigma2.size() != 8) return 3;
        vector<unsigned char> gray(np);
        for (int ni = 0; ni < nImages; ni++) {
            cv::Mat imRGB(h, w, CV_8UC3, bgr.data() + np * 3 * ni), imD(h, w, CV_16UC1, depth.data() + np * ni);
            if (ni >= 1) {
                detertor->DetectDynaArea(imRGB, imD, imDynaMask, imLabel, ni);          // rgbd_tum_noros.cc:135 (the 15x15 dilation of :138 is OpenCV's own and stays with the caller)
            }
            for (size_t i = 0; i < np; i++) { const int b = imRGB.data[3 * i], g = imRGB.data[3 * i + 1], r = imRGB.data[3 * i + 2]; gray[i] = (unsigned char)((b * 4899 + g * 9617 + r * 1868 + 8192) >> 14); }
            cv::Mat im(h, w, CV_8UC1, gray.data());
            frame.ExtractORB2(0, im, ni >= 1 ? imDynaMask : cv::Mat());                  // src/Frame.cc:308 / :313
            if (ni == nImages - 1) frame.ExtractORB2(1, im, cv::Mat());                  // the right extractor of the stereo path (:317)
            const int nk = (int)frame.mvKeys.size();
            fwrite(imDynaMask.data, 1, np, fo); fwrite(imLabel.data, 1, np, fo); fwrite(&nk, 4, 1, fo);
            for (const cv::KeyPoint& k : frame.mvKeys) { const float f5[5] = {k.pt.x, k.pt.y, k.size, k.angle, k.response}; const int i2[2] = {k.octave, k.class_id}; fwrite(f5, 4, 5, fo); fwrite(i2, 4, 2, fo); }
            if (nk) fwrite(frame.mDescriptors.data, 1, (size_t)nk * 32, fo);
            if (nk && (frame.mDescriptors.rows != nk || frame.mDescriptors.cols != 32 || frame.mDescriptors.type() != CV_8U)) return 4;
        }
        // src/Frame.cc:544
        const int nRows = frame.mpORBextractorLeft->mvImagePyramid[0].rows;
        if (nRows != h) return 5;
        const int nl = (int)frame.mpORBextractorLeft->mvImagePyramid.size(); fwrite(&nl, 4, 1, fo);
        for (int l = 0; l < nl; l++) {
            const cv::Mat& L = frame.mpORBextractorLeft->mvImagePyramid[l]; const int wh[2] = {L.cols, L.rows}; fwrite(wh, 4, 2, fo);
            for (int y = 0; y < L.rows; y++) fwrite(L.ptr(y), 1, L.cols, fo);
        }
        // src/Frame.cc:634-651 (shape of the sliding-window patch; the matcher's arithmetic on it is OpenCV's)
        { const int octave = 1, scaledvL = 40, scaleduL = 60; const int ww = 5;
          cv::Mat IL = frame.mpORBextractorLeft->mvImagePyramid[octave].rowRange(scaledvL - ww, scaledvL + ww + 1).colRange(scaleduL - ww, scaleduL + ww + 1);
          const float endu = 70.f;
          if (endu >= frame.mpORBextractorRight->mvImagePyramid[octave].cols) return 6;                       // :645
          for (int y = 0; y < IL.rows; y++) fwrite(IL.ptr(y), 1, IL.cols, fo); }
    } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); fclose(fo); return 1; }
    fclose(fo);
    return 0;
}
