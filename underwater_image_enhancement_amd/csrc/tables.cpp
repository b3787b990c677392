// Host-side construction of the constant tables the kernels index (uploaded once per context).
//
//  * LabTables: the integer tables of OpenCV's 8-bit sRGB<->Lab conversion, the arithmetic behind
//    cv2.cvtColor(.., COLOR_RGB2LAB / COLOR_LAB2RGB) at six_stadigy.py:204,207 and
//    enhancement_strategies.py:299,306.  OpenCV is a third-party dependency that is not vendored in the
//    reference; the construction restates its published initLabTabs() recipe.
//  * CastTables: per-binade rounding tables that let k_entry.hip reproduce NumPy's *sequential* float32
//    accumulation of `img.mean(axis=(0,1))` (six_stadigy.py:294) in closed form, chunk by chunk.
#include <cmath>
#include <cstring>

#include "common.h"

namespace uwie {

namespace {

// OpenCV's cubeRoot(): a quartic rational fit on the mantissa, evaluated in double, rounded to float.
float cv_cube_root(float value)
{
    uint32_t bits;
    std::memcpy(&bits, &value, 4);
    const uint32_t mag = bits & 0x7fffffffu, sign = bits & 0x80000000u;
    if (mag == 0) return value;
    int ex = int(mag >> 23) - 127;
    int shx = ex % 3;
    if (shx >= 0) shx -= 3;
    ex = (ex - shx) / 3;
    uint32_t fb = (mag & 0x7fffffu) | (uint32_t(shx + 127) << 23);
    float frf;
    std::memcpy(&frf, &fb, 4);
    const double f = frf;  // 0.125 <= f < 1
    const double num = ((((45.2548339756803022511987494 * f + 192.2798368355061050458134625) * f +
                          119.1654824285581628956914143) * f + 13.43250139086239872172837314) * f +
                        0.1636161226585754240958355063);
    const double den = ((((14.80884093219134573786480845 * f + 151.9714051044435648658557668) * f +
                          168.5254414101568283957668343) * f + 33.9905941350215598754191872) * f + 1.0);
    const float r = float(num / den);
    uint32_t rb;
    std::memcpy(&rb, &r, 4);
    rb = (rb + (uint32_t(ex) << 23)) | sign;
    float out;
    std::memcpy(&out, &rb, 4);
    return out;
}

float srgb_to_linear(float x)
{
    const double xd = x;
    return float(xd <= 809.0 / 20000.0 ? xd / (323.0 / 25.0) : std::pow((xd + 11.0 / 200.0) / (1.0 + 11.0 / 200.0), 12.0 / 5.0));
}

float linear_to_srgb(float x)
{
    const double xd = x;
    return float(xd <= 7827.0 / 2500000.0 ? xd * (323.0 / 25.0)
                                          : std::pow(xd, 1.0 / (12.0 / 5.0)) * (1.0 + 11.0 / 200.0) - 11.0 / 200.0);
}

}  // namespace

void build_lab_tables(LabTables *t)
{
    constexpr int kGammaShift = 3, kLabShift = 12, kLabShift2 = kLabShift + kGammaShift, kBase = 1 << 14;
    constexpr int kMinAB = -8145;
    const double white[3] = {0.950456, 1.0, 1.088754};
    const double rgb2xyz[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
    const double xyz2rgb[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311};

    const float gscale = float(255 * (1 << kGammaShift));
    for (int i = 0; i < 256; ++i) t->gamma[i] = (uint16_t)std::lrintf(gscale * srgb_to_linear(float(i) / 255.0f));
    for (int i = 0; i < 4096; ++i)
        t->invgamma[i] = (uint8_t)std::lrintf(255.0f * linear_to_srgb((1.0f / 4096.0f) * float(i)));

    const float thresh = 216.0f / 24389.0f, slope = 841.0f / 108.0f, bias = 16.0f / 116.0f;
    const float step = 1.0f / (255.0f * float(1 << kGammaShift));
    for (int i = 0; i < 3072; ++i) {
        const float x = step * float(i);
        const float f = x < thresh ? std::fmaf(x, slope, bias) : cv_cube_root(x);
        t->cbrt[i] = (uint16_t)std::lrintf(float(1 << kLabShift2) * f);
    }
    for (int i = 0; i < 256; ++i) {
        int y, ify;
        if (i <= 20) {  // L* <= 8: linear segment
            y = (int)std::lrintf(float(i * kBase * 20 * 9) / float(17 * 29 * 29 * 29));
            ify = (int)std::lrintf(float(kBase) * (16.0f / 116.0f + float(i * 100) / float(4 * 255 * 29)));
        } else {
            const float fy = float(i * 100 * kBase) / float(255 * 116) + float(16 * kBase) / 116.0f;
            ify = (int)std::lrintf(fy);
            y = (int)std::lrintf(fy * fy * fy / float(kBase * kBase));
        }
        t->ltoyf[2 * i] = y;
        t->ltoyf[2 * i + 1] = ify;
    }
    for (int i = kMinAB; i < kBase * 9 / 4 + kMinAB; ++i)
        t->abtoxz[i - kMinAB] = i <= 3390 ? i * 108 / 841 - kBase * 16 / 116 * 108 / 841 : i * i / kBase * i / kBase;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            t->fwd[r * 3 + c] = (int)std::lrint(double(1 << kLabShift) * rgb2xyz[r * 3 + c] / white[r]);
            t->inv[r * 3 + c] = (int)std::lrint(double(1 << kLabShift) * xyz2rgb[r * 3 + c] * white[c]);
        }
}

void build_cast_tables(CastTables *t)
{
    for (int k = 0; k < 256; ++k) {
        const float x = float(k) / 255.0f;  // six_stadigy.py:406
        uint32_t bits;
        std::memcpy(&bits, &x, 4);
        const uint64_t M = k ? ((bits & 0x7fffffu) | 0x800000u) : 0;  // x = M * 2^E
        const int E = int(bits >> 23) - 127 - 23;
        for (int ei = 0; ei < kCastBinades; ++ei) {
            const int shift = (kCastBinadeMin + ei - 23) - E;  // x / ulp_e = M / 2^shift
            uint64_t R = 0;
            uint8_t tie = 0;
            if (M == 0) {
            } else if (shift <= 0) {
                R = M << (-shift);
            } else if (shift < 40) {
                const uint64_t half = 1ull << (shift - 1), rem = M & ((1ull << shift) - 1);
                R = M >> shift;
                if (rem > half) R += 1;
                else if (rem == half) tie = 1;  // ties-to-even depends on the accumulator's parity: resolved sequentially
            }
            t->R[ei][k] = (uint32_t)R;
            t->tie[ei][k] = tie;
            t->RT[k][ei] = (uint32_t)R | (tie ? 0x80000000u : 0u);
        }
    }
    for (int k = 0; k < 256; ++k) {
        t->tiebin[k] = 255;
        for (int ei = 0; ei < kCastBinades; ++ei) {
            const uint32_t R = t->R[ei][k];
            uint32_t &lo = t->RT2[k >> 1][ei].x, &hi = t->RT2[k >> 1][ei].y;
            if (!(k & 1)) lo = hi = 0;
            lo |= (R & 0x1fffu) << (16 * (k & 1));
            hi |= (R >> 13) << (16 * (k & 1));  // R < 2^26
            if (t->tie[ei][k]) t->tiebin[k] = (uint8_t)ei;  // x / ulp = n + 1/2 pins the ulp: at most one binade
        }
    }
}

}  // namespace uwie
