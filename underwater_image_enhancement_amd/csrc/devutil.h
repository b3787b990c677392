// Device-side helpers shared by the kernels (gfx950: wave = 64 lanes).
// The whole library is compiled with -ffp-contract=off: every float/double operation below rounds once,
// in the order written, which is what makes the results reproduce NumPy's / OpenCV's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uwie {

constexpr int kWave = 64;

// x = u8/255 (six_stadigy.py:406, true float32 division), then the cast attenuation of
// color_correction (six_stadigy.py:310,317: channel * 0.85, clip is a no-op for values in [0,1]).
__device__ __forceinline__ float px_norm(uint32_t u) { return (float)u / 255.0f; }
// The same value without the division: q0 = u * fl(1/255), one Newton correction with two fmas.  Equal to the
// correctly rounded u / 255.0f for every byte value (checked exhaustively in tests/test_cabi.py).
__device__ __forceinline__ float px_norm_fast(uint32_t u)
{
    const float x = (float)u, rcp = 1.0f / 255.0f;
    const float q0 = x * rcp;
    return fmaf(fmaf(-q0, 255.0f, x), rcp, q0);
}
__device__ __forceinline__ float px_val(uint32_t u, bool attenuate)
{
    float x = (float)u / 255.0f;
    return attenuate ? x * 0.85f : x;
}
// Four consecutive RGB pixels (12 bytes).  Frames whose pixel count is a multiple of 4 keep every 4-pixel group
// 4-byte aligned (frame bases are multiples of 12 bytes from a 256-byte aligned allocation), so the group comes in
// as three dwords; other frames fall back to byte loads.
struct Px4 {
    uint32_t r[4], g[4], b[4];
};
__device__ __forceinline__ Px4 load_px4(const uint8_t *p, int n, bool aligned)
{
    Px4 o;
    if (aligned && n == 4) {
        const uint32_t *w = reinterpret_cast<const uint32_t *>(p);
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];  // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
        o.r[0] = w0 & 255; o.g[0] = (w0 >> 8) & 255; o.b[0] = (w0 >> 16) & 255;
        o.r[1] = w0 >> 24; o.g[1] = w1 & 255; o.b[1] = (w1 >> 8) & 255;
        o.r[2] = (w1 >> 16) & 255; o.g[2] = w1 >> 24; o.b[2] = w2 & 255;
        o.r[3] = (w2 >> 8) & 255; o.g[3] = (w2 >> 16) & 255; o.b[3] = w2 >> 24;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = i < n;
            o.r[i] = ok ? p[3 * i] : 0;
            o.g[i] = ok ? p[3 * i + 1] : 0;
            o.b[i] = ok ? p[3 * i + 2] : 0;
        }
    }
    return o;
}
// four pixels at any byte address (three unaligned dword loads)
__device__ __forceinline__ Px4 load_px4_any(const uint8_t *p)
{
    typedef uint32_t __attribute__((aligned(1))) u32_a1;
    const u32_a1 *w = reinterpret_cast<const u32_a1 *>(p);
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
    Px4 o;
    o.r[0] = w0 & 255; o.g[0] = (w0 >> 8) & 255; o.b[0] = (w0 >> 16) & 255;
    o.r[1] = w0 >> 24; o.g[1] = w1 & 255; o.b[1] = (w1 >> 8) & 255;
    o.r[2] = (w1 >> 16) & 255; o.g[2] = w1 >> 24; o.b[2] = w2 & 255;
    o.r[3] = (w2 >> 8) & 255; o.g[3] = (w2 >> 16) & 255; o.b[3] = w2 >> 24;
    return o;
}
__device__ __forceinline__ void store_px4(uint8_t *p, const uint32_t *r, const uint32_t *g, const uint32_t *b, int n,
                                          bool aligned)
{
    if (aligned && n == 4) {
        uint32_t *w = reinterpret_cast<uint32_t *>(p);
        w[0] = r[0] | (g[0] << 8) | (b[0] << 16) | (r[1] << 24);
        w[1] = g[1] | (b[1] << 8) | (r[2] << 16) | (g[2] << 24);
        w[2] = b[2] | (r[3] << 8) | (g[3] << 16) | (b[3] << 24);
    } else {
        for (int i = 0; i < n; ++i) {
            p[3 * i] = (uint8_t)r[i];
            p[3 * i + 1] = (uint8_t)g[i];
            p[3 * i + 2] = (uint8_t)b[i];
        }
    }
}

// n / den for many n and one den (the percentile stretch: one denominator per image and channel).  The compiler expands
// a correctly rounded float32 division into v_rcp_f32, one Newton step on the reciprocal y, q0 = n*y and two residual
// corrections q = fma(fma(-den, q, n), y, q), wrapped in v_div_scale / v_div_fixup for operands near the ends of the
// exponent range.  For den in [2^-24, 2^24] and n = 0 or |n| in [2^-60, 2^60] nothing is scaled and nothing is fixed up
// except the sign of a zero, so keeping y and doing the last five operations per value gives the division's bits;
// everything else takes the division (quot), or is known to be tiny either way (quot_unit).
struct StretchDiv {
    float den, y;
    bool fast;
    __device__ __forceinline__ void set(float d)
    {
        den = d;
        fast = d >= 0x1p-24f && d <= 0x1p24f;
        const float r = __builtin_amdgcn_rcpf(d);
        y = fmaf(fmaf(-d, r, 1.0f), r, r);
    }
    __device__ __forceinline__ float seq(float n) const
    {
        float q = n * y;
        q = fmaf(fmaf(-den, q, n), y, q);
        return fmaf(fmaf(-den, q, n), y, q);
    }
    __device__ __forceinline__ float quot(float n) const  // any n
    {
        const float m = fabsf(n);
        return fast && (m == 0.0f || (m >= 0x1p-60f && m <= 0x1p60f)) ? seq(n) : n / den;
    }
    // |n| <= 2 (differences of values in [0, 1]): below 2^-60 the quotient is under 2^-36 by either route, a zero byte
    // after the x255 quantisation that follows in the fused kernels
    __device__ __forceinline__ float quot_unit(float n) const { return fast ? seq(n) : n / den; }
};

// x**g for float32 operands (np.power / torch.pow on float32 data) where it is evaluated per pixel: float64
// exp2(g * log2(x)) with x = m * 2^e, m in [sqrt(1/2), sqrt(2)): log(m) = 2 atanh((m-1)/(m+1)) by its series to t^19
// (truncation 2^-50), exp2 of the fraction by its Taylor polynomial of degree 12 (2^-52), so the value is good to ~2^-46
// and its float32 rounding is the exact power's except about once per million arguments (then the neighbouring float).
// The library pow() this replaces costs about four times as many instructions; the reference's own powf is only
// faithfully rounded, so the contract (<= 1 float32 ulp against it, written in the tests) is unchanged.  Arguments
// outside 2^-126 <= x <= 16, 0 < g <= 64 take pow().
__device__ __forceinline__ float pow_f32_fast(float x, float g)
{
    if (!(x >= 0x1p-126f && x <= 16.0f && g > 0.0f && g <= 64.0f)) return (float)pow((double)x, (double)g);
    const uint32_t bits = __float_as_uint(x);
    int e = (int)(bits >> 23) - 127;
    float mf = __uint_as_float((bits & 0x7fffffu) | 0x3f800000u);  // [1, 2)
    if (mf > 1.41421356f) {
        mf *= 0.5f;
        e += 1;
    }
    const double m = (double)mf, d = m + 1.0;
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    const double t = (m - 1.0) * r, t2 = t * t;
    double p = 1.0 / 19.0;
    p = fma(p, t2, 1.0 / 17.0);
    p = fma(p, t2, 1.0 / 15.0);
    p = fma(p, t2, 1.0 / 13.0);
    p = fma(p, t2, 1.0 / 11.0);
    p = fma(p, t2, 1.0 / 9.0);
    p = fma(p, t2, 1.0 / 7.0);
    p = fma(p, t2, 1.0 / 5.0);
    p = fma(p, t2, 1.0 / 3.0);
    p = fma(p, t2, 1.0);
    const double lg2 = fma(2.8853900817779268 * t, p, (double)e);  // log2(x)
    const double y = (double)g * lg2;
    if (y < -160.0) return 0.0f;  // below the smallest float32 subnormal
    const double n = rint(y), f = y - n;
    double q = 2.5678435993488206e-11;
    q = fma(q, f, 4.4455382718708116e-10);
    q = fma(q, f, 7.054911620801123e-09);
    q = fma(q, f, 1.01780860092397e-07);
    q = fma(q, f, 1.321548679014431e-06);
    q = fma(q, f, 1.5252733804059841e-05);
    q = fma(q, f, 0.0001540353039338161);
    q = fma(q, f, 0.0013333558146428443);
    q = fma(q, f, 0.009618129107628477);
    q = fma(q, f, 0.05550410866482158);
    q = fma(q, f, 0.24022650695910072);
    q = fma(q, f, 0.6931471805599453);
    q = fma(q, f, 1.0);
    const double v = __longlong_as_double(__double_as_longlong(q) + (long long)n * (1ll << 52));  // q * 2^n, |n| <= 256
    return (float)v;
}

// which channel color_correction attenuates for a cast kind (UWIE_CAST_*): greenish -> G, bluish -> B
__device__ __forceinline__ bool px_atten(int kind, int c) { return kind != 0 && c == kind; }
// (img * 255).astype(np.uint8): float32 product, truncation toward zero
__device__ __forceinline__ uint32_t quant_u8(float x) { return (uint32_t)(int)(x * 255.0f) & 0xffu; }

__device__ __forceinline__ uint32_t gray_fixed(uint32_t r, uint32_t g, uint32_t b, int shift)
{
    return shift == 15 ? (r * 9798u + g * 19235u + b * 3735u + 16384u) >> 15
                       : (r * 4899u + g * 9617u + b * 1868u + 8192u) >> 14;
}

// gray_fixed() of four pixels in float32 (round 4: k_gray_strong, k_chunk_hist_quad): R c_r + G c_g + B c_b + 1/2 with
// c = coefficient / 2^shift is exact (integers below 2^24 scaled by a power of two), so its truncation is gray_fixed();
// ATT = 1 / 2: the green / blue byte is the attenuated one, 17 u / 20 = trunc(0.85f u + 0.025) (the fractional part of
// 17 u / 20 is a multiple of 0.05, the float32 error below 3e-5).  d: the twelve bytes R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3.
static inline void gray_f32_coeffs(int shift, float &cr, float &cg, float &cb)
{
    const float sc = shift == 15 ? 1.0f / 32768.0f : 1.0f / 16384.0f;
    cr = (shift == 15 ? 9798.0f : 4899.0f) * sc;
    cg = (shift == 15 ? 19235.0f : 9617.0f) * sc;
    cb = (shift == 15 ? 3735.0f : 1868.0f) * sc;
}
template <int ATT>
__device__ __forceinline__ uint32_t gray1_f32(float r, float g, float b, float cr, float cg, float cb)
{
    if (ATT == 1) g = truncf(fmaf(g, 0.85f, 0.025f));
    if (ATT == 2) b = truncf(fmaf(b, 0.85f, 0.025f));
    return (uint32_t)fmaf(b, cb, fmaf(g, cg, fmaf(r, cr, 0.5f)));
}
template <int ATT>
__device__ __forceinline__ uint32_t gray4_f32(const uint32_t (&d)[3], float cr, float cg, float cb)
{
    // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
    const uint32_t c0 = d[0], c1 = d[1], c2 = d[2];
    auto f = [](uint32_t w, int i) { return (float)((w >> (8 * i)) & 0xffu); };
    const uint32_t g0 = gray1_f32<ATT>(f(c0, 0), f(c0, 1), f(c0, 2), cr, cg, cb);
    const uint32_t g1 = gray1_f32<ATT>(f(c0, 3), f(c1, 0), f(c1, 1), cr, cg, cb);
    const uint32_t g2 = gray1_f32<ATT>(f(c1, 2), f(c1, 3), f(c2, 0), cr, cg, cb);
    const uint32_t g3 = gray1_f32<ATT>(f(c2, 1), f(c2, 2), f(c2, 3), cr, cg, cb);
    return g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
}

// order-preserving integer image of a float32 (radix select, k_select.hip): ascending key == ascending value
__device__ __forceinline__ uint32_t f32_key(float v)
{
    const uint32_t b = __float_as_uint(v);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}

__device__ __forceinline__ uint64_t f64_key(double v)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    return b ^ ((b >> 63) ? 0xffffffffffffffffull : 0x8000000000000000ull);
}

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
__device__ __forceinline__ int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        p = p < 0 ? -p : 2 * (len - 1) - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src)
{
    uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int m)
{
    uint32_t lo = __shfl_xor((uint32_t)v, m), hi = __shfl_xor((uint32_t)(v >> 32), m);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int d)
{
    uint32_t lo = __shfl_up((uint32_t)v, d), hi = __shfl_up((uint32_t)(v >> 32), d);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += shfl_xor_u64(v, o);
    return v;
}
__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint64_t t = shfl_up_u64(v, o);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

}  // namespace uwie
