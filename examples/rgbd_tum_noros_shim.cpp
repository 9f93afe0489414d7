// The reference's frame loop (Examples/RGB-D/rgbd_tum_noros.cc:94-170) written against the drop-in classes of include/DynaDetect.h and
// include/ORBextractor.h -- ORB_SLAM2::DynaDetect(imgLast, imgLastLast, fx, fy, cx, cy, depthScale), DetectDynaArea(img, depth, imgDyna,
// imgLabel, nImg), the 15x15 dilation of the caller (:108, :138) and (*ORBextractor)(gray, mask, keys, descriptors) (src/Frame.cc:308) --
// without OpenCV: images are sind::Image views over a raw dump.  tests/test_cpp_shim_gpu.py builds it with g++, runs it on the GPU and
// compares every output with the Python mirror of the same C ABI.
//
//   rgbd_tum_noros_shim <in.raw> <out.raw> fx fy cx cy depthFactor nFeatures scaleFactor nLevels iniThFAST minThFAST rgbOrder
//   in : int32 n, w, h ; n x BGR u8 [h][w][3] ; n x depth u16 [h][w]
//   out: per frame  dyna u8[h*w], label u8[h*w], mask u8[h*w], int32 nkp, nkp x sind_keypoint, nkp x 32 descriptor bytes
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "DynaDetect.h"
#include "ORBextractor.h"

int main(int argc, char** argv) {
    if (argc != 14) { std::fprintf(stderr, "usage: %s in out fx fy cx cy depthFactor nFeatures scaleFactor nLevels iniTh minTh rgbOrder\n", argv[0]); return 2; }
    FILE* fi = std::fopen(argv[1], "rb"); if (!fi) { std::perror(argv[1]); return 2; }
    int hdr[3]; if (std::fread(hdr, 4, 3, fi) != 3) return 2;
    const int n = hdr[0], w = hdr[1], h = hdr[2]; const size_t np = (size_t)w * h;
    std::vector<uint8_t> bgr(np * 3 * n); std::vector<uint16_t> depth(np * n);
    if (std::fread(bgr.data(), 1, bgr.size(), fi) != bgr.size() || std::fread(depth.data(), 2, depth.size(), fi) != depth.size()) return 2;
    std::fclose(fi);
    const float fx = (float)std::atof(argv[3]), fy = (float)std::atof(argv[4]), cx = (float)std::atof(argv[5]), cy = (float)std::atof(argv[6]), depthFactor = (float)std::atof(argv[7]);
    const bool rgbOrder = std::atoi(argv[13]) != 0;
    FILE* fo = std::fopen(argv[2], "wb"); if (!fo) { std::perror(argv[2]); return 2; }
    try {
        auto color = [&](int i) { return sind::Image{bgr.data() + np * 3 * i, w, h, w * 3, 3, 1}; };
        auto depthOf = [&](int i) { return sind::Image{depth.data() + np * i, w, h, w * 2, 1, 2}; };
        // rgbd_tum_noros.cc:103-107: the detector starts from the first image twice
        ORB_SLAM2::DynaDetect dynaDetect(color(0), color(0), fx, fy, cx, cy, depthFactor);
        ORB_SLAM2::ORBextractor extractor(std::atoi(argv[8]), (float)std::atof(argv[9]), std::atoi(argv[10]), std::atoi(argv[11]), std::atoi(argv[12]));
        std::vector<uint8_t> dyna(np), label(np), mask(np), gray(np), desc; std::vector<sind_keypoint> keys;
        for (int ni = 0; ni < n; ni++) {
            std::fill(dyna.begin(), dyna.end(), 0); std::fill(label.begin(), label.end(), 0); std::fill(mask.begin(), mask.end(), 0);
            if (ni >= 1) {                                  // :131-139
                sind::Image imDyna{dyna.data(), w, h, w, 1, 1}, imLabel{label.data(), w, h, w, 1, 1};
                dynaDetect.DetectDynaArea(color(ni), depthOf(ni), imDyna, imLabel, ni);
                mask = dyna; sind::Image imMask{mask.data(), w, h, w, 1, 1};
                dynaDetect.DilateForTracking(imMask);
            }
            // Tracking::GrabImageRGBD (src/Tracking.cc:246-259): cvtColor RGB2GRAY / BGR2GRAY (fixed point, 14 bits)
            const uint8_t* p = bgr.data() + np * 3 * ni;
            for (size_t i = 0; i < np; i++) { const int b = p[3 * i], g = p[3 * i + 1], r = p[3 * i + 2]; gray[i] = (uint8_t)((b * (rgbOrder ? 4899 : 1868) + g * 9617 + r * (rgbOrder ? 1868 : 4899) + 8192) >> 14); }
            extractor(gray.data(), w, h, w, mask.data(), w, keys, desc);       // src/Frame.cc:308
            const int nk = (int)keys.size();
            std::fwrite(dyna.data(), 1, np, fo); std::fwrite(label.data(), 1, np, fo); std::fwrite(mask.data(), 1, np, fo);
            std::fwrite(&nk, 4, 1, fo); std::fwrite(keys.data(), sizeof(sind_keypoint), keys.size(), fo); std::fwrite(desc.data(), 1, desc.size(), fo);
        }
        std::printf("Images in the sequence: %d, scale factors %zu\n", n, extractor.GetScaleFactors().size());
    } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); std::fclose(fo); return 1; }
    std::fclose(fo);
    return 0;
}
