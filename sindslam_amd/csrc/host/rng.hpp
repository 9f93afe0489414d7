// Host stage support: cv::RNG (core/rand.cpp) — multiply-with-carry generator and the Marsaglia-Tsang ziggurat normal
// sampler, as used by reference DynaDetect.cc:1163, 1187 (cv::RNG rng(12345); rng.gaussian(0.5) per grid sample).
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>

namespace sind {

class CvRng {
public:
    explicit CvRng(uint64_t seed) : state_(seed ? seed : 0xffffffffull) { init_tables(); }
    double gaussian(double sigma) { return (double)randn() * sigma; }
private:
    uint64_t state_;
    static uint64_t step(uint64_t x) { return (uint64_t)(uint32_t)x * 4164903690U + (x >> 32); }
    struct Tables { uint32_t kn[128]; float wn[128], fn[128]; };
    static const Tables& init_tables() {
        static const Tables T = [] {
            Tables t; const double m1 = 2147483648.0; double dn = 3.442619855899, tn = dn; const double vn = 9.91256303526217e-3;
            const double q = vn / std::exp(-.5 * dn * dn);
            t.kn[0] = (uint32_t)((dn / q) * m1); t.kn[1] = 0;
            t.wn[0] = (float)(q / m1); t.wn[127] = (float)(dn / m1);
            t.fn[0] = 1.f; t.fn[127] = (float)std::exp(-.5 * dn * dn);
            for (int i = 126; i >= 1; i--) {
                dn = std::sqrt(-2. * std::log(vn / dn + std::exp(-.5 * dn * dn)));
                t.kn[i + 1] = (uint32_t)((dn / tn) * m1); tn = dn;
                t.fn[i] = (float)std::exp(-.5 * dn * dn); t.wn[i] = (float)(dn / m1);
            }
            return t;
        }();
        return T;
    }
    float randn() {
        const Tables& T = init_tables();
        const float r = 3.442620f, rng_flt = 2.3283064365386962890625e-10f;
        uint64_t temp = state_; float x, y;
        for (;;) {
            const int hz = (int)temp; temp = step(temp);
            const int iz = hz & 127;
            x = hz * T.wn[iz];
            if ((uint32_t)std::abs(hz) < T.kn[iz]) break;
            if (iz == 0) {
                do {
                    x = (uint32_t)temp * rng_flt; temp = step(temp);
                    y = (uint32_t)temp * rng_flt; temp = step(temp);
                    x = (float)(-std::log(x + FLT_MIN) * 0.2904764);
                    y = (float)-std::log(y + FLT_MIN);
                } while (y + y < x * x);
                x = hz > 0 ? r + x : -r - x;
                break;
            }
            y = (uint32_t)temp * rng_flt; temp = step(temp);
            if (T.fn[iz] + y * (T.fn[iz - 1] - T.fn[iz]) < (float)std::exp(-.5 * x * x)) break;
        }
        state_ = temp;
        return x;
    }
};

}  // namespace sind
