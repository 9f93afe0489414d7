// 128-bit fingerprint of a stream's inter-frame state (reference DynaDetect.h:172-178: imgDynaLast, imgLabelLast, imgMaskHighErrorLast; rolled at
// DynaDetect.cc:1660-1664).  Every output of a frame is a deterministic function of (input frames, state before the frame), so two runs of a sequence
// whose states agree after frame q agree on every later frame: the chunked sequence mode (sindslam_amd/sequence.py) compares the state a chunk rebuilt in its
// warm-up frames with the true state its predecessor ended in, and repairs the chunk where they differ.  The fingerprint stands in for the 1.2 MB blob in
// those comparisons: four independent multiply-rotate lanes over the 64-bit words, folded into two 64-bit values (a false "equal" needs a 128-bit collision).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace sind {

struct StateHash {
    uint64_t a[4] = {0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull}; uint64_t n = 0; int lane = 0;
    static inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
    static inline uint64_t fmix(uint64_t k) { k ^= k >> 33; k *= 0xFF51AFD7ED558CCDull; k ^= k >> 33; k *= 0xC4CEB9FE1A85EC53ull; k ^= k >> 33; return k; }
    inline void word(uint64_t w) {
        static constexpr uint64_t P[4] = {0x9FB21C651E98DF25ull, 0xD6E8FEB86659FD93ull, 0xA0761D6478BD642Full, 0xE7037ED1A0B428DBull};
        a[lane] = (rotl(a[lane], 27) ^ w) * P[lane]; lane = (lane + 1) & 3; n++;
    }
    void words(const uint64_t* w, size_t cnt) { for (size_t i = 0; i < cnt; i++) word(w[i]); }
    void bytes(const void* p, size_t nbytes) {
        const uint8_t* b = (const uint8_t*)p; size_t i = 0;
        for (; i + 8 <= nbytes; i += 8) { uint64_t w; std::memcpy(&w, b + i, 8); word(w); }
        if (i < nbytes) { uint64_t w = 0; std::memcpy(&w, b + i, nbytes - i); word(w); }
        word((uint64_t)nbytes);
    }
    void finish(uint64_t out[2]) const {
        out[0] = fmix(a[0] ^ rotl(a[1], 13) ^ rotl(a[2], 29) ^ rotl(a[3], 47) ^ n);
        out[1] = fmix(a[0] * 0x9E3779B97F4A7C15ull + a[1] * 0xC2B2AE3D27D4EB4Full + a[2] * 0x165667B19E3779F9ull + a[3] * 0x27D4EB2F165667C5ull + rotl(n, 32));
        if (out[0] == 0 && out[1] == 0) out[1] = 1;          // (0, 0) means "no state recorded" to the callers
    }
};

}  // namespace sind
