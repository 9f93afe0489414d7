// The reference's frame loop (Examples/RGB-D/rgbd_tum_noros.cc:94-170) written against the drop-in classes of include/DynaDetect.h and
// include/ORBextractor.h -- ORB_SLAM2::DynaDetect(imgLast, imgLastLast, fx, fy, cx, cy, depthScale), DetectDynaArea(img, depth, imgDyna,
// imgLabel, nImg), the 15x15 dilation of the caller (:108, :138) and (*ORBextractor)(gray, mask, keys, descriptors) (src/Frame.cc:308) --
// without OpenCV: images are sind::Image views over a raw dump.  tests/test_cpp_shim_gpu.py builds it with g++, runs it on the GPU and
// compares every output with the Python mirror of the same C ABI.
//
//   rgbd_tum_noros_shim <in.raw> <out.raw> fx fy cx cy depthFactor nFeatures scaleFactor nLevels iniThFAST minThFAST rgbOrder [--chunks N [--warmup W] [--rank r --world w --port p]]
//   in : int32 n, w, h ; n x BGR u8 [h][w][3] ; n x depth u16 [h][w]
//   out: per frame  dyna u8[h*w], label u8[h*w], mask u8[h*w], int32 nkp, nkp x sind_keypoint, nkp x 32 descriptor bytes
//
// --chunks N: the SAME results at several times the rate for a sequence that is there as a whole (offline evaluation): the frames go through the library's chunked-sequence
// driver (sind_seq_*, include/sind_hip.h: N verified chunks per rank -- speculate, verify the chunk seams by state fingerprints, repair) instead of the frame loop.  With
// --world w > 1 every rank runs this binary on the same input (rank r, the exchange over TCP ports p .. p + w - 1) and writes the frames it owns:
//   out: per frame  int32 owned ; if owned: the record above
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "DynaDetect.h"
#include "ORBextractor.h"

static int run_chunked(const std::vector<uint8_t>& bgr, const std::vector<uint16_t>& depth, int n, int w, int h, char** argv, int chunks, int warmup, int rank, int world, int port, FILE* fo);

int main(int argc, char** argv) {
    if (argc < 14) { std::fprintf(stderr, "usage: %s in out fx fy cx cy depthFactor nFeatures scaleFactor nLevels iniTh minTh rgbOrder [--chunks N [--warmup W] [--rank r --world w --port p]]\n", argv[0]); return 2; }
    int chunks = 0, warmup = 16, rank = 0, world = 1, port = 0;
    for (int i = 14; i + 1 < argc; i += 2) {
        const std::string k = argv[i]; const int v = std::atoi(argv[i + 1]);
        if (k == "--chunks") chunks = v; else if (k == "--warmup") warmup = v; else if (k == "--rank") rank = v; else if (k == "--world") world = v; else if (k == "--port") port = v;
        else { std::fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    FILE* fi = std::fopen(argv[1], "rb"); if (!fi) { std::perror(argv[1]); return 2; }
    int hdr[3]; if (std::fread(hdr, 4, 3, fi) != 3) return 2;
    const int n = hdr[0], w = hdr[1], h = hdr[2]; const size_t np = (size_t)w * h;
    std::vector<uint8_t> bgr(np * 3 * n); std::vector<uint16_t> depth(np * n);
    if (std::fread(bgr.data(), 1, bgr.size(), fi) != bgr.size() || std::fread(depth.data(), 2, depth.size(), fi) != depth.size()) return 2;
    std::fclose(fi);
    const float fx = (float)std::atof(argv[3]), fy = (float)std::atof(argv[4]), cx = (float)std::atof(argv[5]), cy = (float)std::atof(argv[6]), depthFactor = (float)std::atof(argv[7]);
    const bool rgbOrder = std::atoi(argv[13]) != 0;
    FILE* fo = std::fopen(argv[2], "wb"); if (!fo) { std::perror(argv[2]); return 2; }
    if (chunks > 0) { const int rc = run_chunked(bgr, depth, n, w, h, argv, chunks, warmup, rank, world, port, fo); std::fclose(fo); return rc; }
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

// The whole sequence through sind_seq_*: plain C calls, no Python, no torch.distributed.
static int run_chunked(const std::vector<uint8_t>& bgr, const std::vector<uint16_t>& depth, int n, int w, int h, char** argv, int chunks, int warmup, int rank, int world, int port, FILE* fo) {
    const size_t np = (size_t)w * h;
    auto die = [](const char* what) { std::fprintf(stderr, "error: %s: %s\n", what, sind_last_error()); return 1; };
    sind_seq_config cfg{};
    cfg.pipe.width = w; cfg.pipe.height = h; cfg.pipe.fx = (float)std::atof(argv[3]); cfg.pipe.fy = (float)std::atof(argv[4]); cfg.pipe.cx = (float)std::atof(argv[5]); cfg.pipe.cy = (float)std::atof(argv[6]);
    cfg.pipe.depth_scale = (float)std::atof(argv[7]); cfg.pipe.nfeatures = std::atoi(argv[8]); cfg.pipe.scale_factor = (float)std::atof(argv[9]); cfg.pipe.nlevels = std::atoi(argv[10]);
    cfg.pipe.ini_th_fast = std::atoi(argv[11]); cfg.pipe.min_th_fast = std::atoi(argv[12]); cfg.pipe.orb_gray_rgb_order = std::atoi(argv[13]) != 0; cfg.pipe.streams = chunks; cfg.pipe.frames_per_step = 1;
    cfg.frames = n - 1; cfg.steps = 0; cfg.frames_per_step = 4; cfg.warmup = warmup; cfg.repair_frames_per_step = 3; cfg.retain_frames = -1; cfg.verify = 1;
    sind_seq_net* net = nullptr; sind_seq* seq = nullptr;
    if (world > 1 && sind_seq_net_tcp(rank, world, nullptr, port, &net) != 0) return die("sind_seq_net_tcp");
    if (sind_seq_create(&cfg, net, &seq) != 0) return die("sind_seq_create");
    const int cap = 2 * cfg.pipe.nfeatures + 256;
    std::vector<uint8_t> dyna(np * n, 0), label(np * n, 0), mask(np * n, 0), desc((size_t)n * cap * 32); std::vector<sind_keypoint> kps((size_t)n * cap); std::vector<int> nkp(n, -1);
    if (sind_seq_set_host_source(seq, bgr.data(), depth.data(), n) != 0) return die("sind_seq_set_host_source");
    if (sind_seq_set_outputs(seq, n, dyna.data(), label.data(), mask.data(), kps.data(), cap, nkp.data(), desc.data()) != 0) return die("sind_seq_set_outputs");
    if (sind_seq_run(seq) != 0) return die("sind_seq_run");
    double st[16]; (void)sind_seq_stats(seq, st);
    int T = 0, steps = 0, nchunks = 0; (void)sind_seq_plan(seq, &T, &steps, &nchunks, nullptr);
    if (rank == 0) {                       // frame 0 passes through with an all-zero mask (rgbd_tum_noros.cc:100-116): its keypoints come from the extractor alone
        ORB_SLAM2::ORBextractor extractor(cfg.pipe.nfeatures, cfg.pipe.scale_factor, cfg.pipe.nlevels, cfg.pipe.ini_th_fast, cfg.pipe.min_th_fast);
        std::vector<uint8_t> gray(np), d0; std::vector<sind_keypoint> k0; const bool rgbOrder = cfg.pipe.orb_gray_rgb_order != 0;
        for (size_t i = 0; i < np; i++) { const int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2]; gray[i] = (uint8_t)((b * (rgbOrder ? 4899 : 1868) + g * 9617 + r * (rgbOrder ? 1868 : 4899) + 8192) >> 14); }
        extractor(gray.data(), w, h, w, mask.data(), w, k0, d0);
        nkp[0] = (int)k0.size(); std::copy(k0.begin(), k0.end(), kps.begin()); std::copy(d0.begin(), d0.end(), desc.begin());
    }
    for (int f = 0; f < n; f++) {
        const int owned = nkp[f] >= 0 ? 1 : 0; std::fwrite(&owned, 4, 1, fo);
        if (!owned) continue;
        std::fwrite(dyna.data() + np * f, 1, np, fo); std::fwrite(label.data() + np * f, 1, np, fo); std::fwrite(mask.data() + np * f, 1, np, fo);
        std::fwrite(&nkp[f], 4, 1, fo); std::fwrite(kps.data() + (size_t)f * cap, sizeof(sind_keypoint), (size_t)nkp[f], fo); std::fwrite(desc.data() + (size_t)f * cap * 32, 1, (size_t)nkp[f] * 32, fo);
    }
    std::printf("Images in the sequence: %d; rank %d of %d: %d chunks x %d frames per step x %d steps, %d of %d seams repaired\n", n, rank, world, nchunks, T, steps, (int)st[1], (int)st[0]);
    (void)sind_seq_destroy(seq); if (net) (void)sind_seq_net_destroy(net);
    return 0;
}
