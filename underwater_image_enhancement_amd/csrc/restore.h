// restore_image (six_stadigy.py:183-188) evaluated on the fly: J = clip((I - A) / t[..., None] + A, 0, 1) with I the
// colour-corrected float32 frame, A float32 and t the float64 transmission, so the quotient and the sum are float64
// and the result is rounded to float32 once (S6:188 astype).  Kernels that need the restored image more than once
// recompute it from the 3 + 8 bytes per pixel it is made of instead of reading a 12-byte float32 copy.
#pragma once
#include "common.h"
#include "devutil.h"

namespace uwie {

typedef uint32_t __attribute__((aligned(1))) u32_any;
typedef double2 __attribute__((aligned(8))) double2_a8;

// TAB: the float32 differences I - A of the 3 x 256 possible bytes come from a table in LDS (3 KB, filled by the
// block in init(): the caller synchronises before the first use) instead of five operations per value.
// TAB == 2: the table holds the differences already widened to float64 (6 KB; `lds_tab` then points at 768 doubles): the
// quotient's numerator is one ds_read_b64, no conversion.
// F32: the UWIE_INTER_F32T flavour (float32 transmission, float32 restore: see one32_raw); a compile-time property so that
// the float64 paths carry no test for it -- kernels instantiate their loop for both and branch once on RestoreSrc::t32.
template <int TAB, bool F32 = false>
struct RestoreImgT {  // per image, in registers
    static constexpr bool t32 = F32;
    const uint8_t *img;
    const double *t;
    const float *tab;
    float a[3];
    bool att[3];
    __device__ __forceinline__ void init(const RestoreSrc &S, int b, size_t npx, void *lds_tab_v = nullptr)
    {
        float *lds_tab = static_cast<float *>(lds_tab_v);
        tab = lds_tab;
        const int k = S.kind ? S.kind[b] : 0;
        img = S.in + (size_t)b * npx * 3;
        t = F32 ? reinterpret_cast<const double *>(reinterpret_cast<const float *>(S.t) + (size_t)b * npx) : S.t + (size_t)b * npx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a[c] = S.A[b * 3 + c];
            att[c] = px_atten(k, c);
        }
        // (Round 3, from the ISA: the selects between a[0..2] / att[0..2] below become selects between their ADDRESSES, which
        // keeps this struct in 40 bytes of scratch memory and reloads a[] from there in the callers' loops.  Building the
        // table from scalars instead keeps it in registers, changes no kernel's time (A/B, 4K x 64) -- and pushes
        // k_stretch_lab_lut<1, 256> over its 102-register budget: its spilled build gave wrong bytes, so this stays.)
        if (TAB) {
            for (int i = threadIdx.x; i < 768; i += blockDim.x) {
                const int c = i >> 8;
                const float x = px_norm_fast((uint32_t)(i & 255));
                const float d = ((c == 0 ? att[0] : c == 1 ? att[1] : att[2]) ? x * 0.85f : x) - (c == 0 ? a[0] : c == 1 ? a[1] : a[2]);
                if (TAB == 2 && !t32) static_cast<double *>(lds_tab_v)[i] = (double)d;
                else lds_tab[i] = d;
            }
        }
    }
    // px_norm_fast(u) == u / 255.0f for every byte (tests/test_cabi.py), so this is px_val() without the division
    __device__ __forceinline__ double diff64(uint32_t u, int c) const  // (double)diff(u, c)
    {
        if (TAB == 2 && !t32) return reinterpret_cast<const double *>(tab)[c * 256 + u];
        return (double)diff(u, c);
    }
    __device__ __forceinline__ float diff(uint32_t u, int c) const
    {
        if (TAB == 2 && !t32) return (float)reinterpret_cast<const double *>(tab)[c * 256 + u];
        if (TAB) return tab[c * 256 + u];
        const float x = px_norm_fast(u);
        return (att[c] ? x * 0.85f : x) - a[c];
    }
    __device__ __forceinline__ float one(uint32_t u, int c, double tv) const
    {
        const float v = (float)((double)diff(u, c) / tv + (double)a[c]);
        return fminf(fmaxf(v, 0.0f), 1.0f);
    }
    // The three quotients of a pixel share the divisor.  The compiler expands a float64 division into v_rcp_f64, two
    // Newton steps on the reciprocal y, q0 = n*y, r = fma(-d, q0, n), q = fma(r, y, q0) (Markstein: correctly rounded),
    // wrapped in v_div_scale / v_div_fixup for operands near the ends of the exponent range.  For a divisor in
    // [2^-100, 2^100] and a numerator that came from a float32 nothing is scaled or fixed up, so evaluating y once and the last
    // three operations per channel gives the same bits as three divisions; other divisors take the division.
    __device__ __forceinline__ static bool recip_ok(double tv) { return tv >= 0x1p-100 && tv <= 0x1p100; }
    __device__ __forceinline__ static double recip(double tv)
    {
        double y = __builtin_amdgcn_rcp(tv);
        y = fma(y, fma(-tv, y, 1.0), y);
        return fma(y, fma(-tv, y, 1.0), y);
    }
    __device__ __forceinline__ float one_fast(uint32_t u, int c, double tv, double y) const
    {
        const double n = diff64(u, c), q0 = n * y;
        const float v = (float)(fma(fma(-tv, q0, n), y, q0) + (double)a[c]);
        return fminf(fmaxf(v, 0.0f), 1.0f);
    }
    // ---- UWIE_INTER_F32T (BASELINE.json configs[4]: reduced-precision intermediates, its own stated tolerance): the
    // transmission is stored as float32 and the restore is float32 throughout -- reciprocal (v_rcp_f32 + one Newton step),
    // one fused multiply-add per value, no division sequence.  Every kernel that needs the restored image evaluates THIS
    // function, so the histogram, the selection and the stretch see the same values and the order statistics stay exact
    // for the image they describe.
    __device__ __forceinline__ const float *tf() const { return reinterpret_cast<const float *>(t); }
    __device__ __forceinline__ static float recip32(float tv)
    {
        const float y = __builtin_amdgcn_rcpf(tv);
        return fmaf(fmaf(-tv, y, 1.0f), y, y);
    }
    __device__ __forceinline__ float one32_raw(uint32_t u, int c, float y) const { return fmaf(diff(u, c), y, a[c]); }
    __device__ __forceinline__ void pixel32(int p, float &r0, float &r1, float &r2) const
    {
        const uint8_t *q = img + (size_t)p * 3;
        const float y = recip32(tf()[p]);
        r0 = fminf(fmaxf(one32_raw(q[0], 0, y), 0.0f), 1.0f);
        r1 = fminf(fmaxf(one32_raw(q[1], 1, y), 0.0f), 1.0f);
        r2 = fminf(fmaxf(one32_raw(q[2], 2, y), 0.0f), 1.0f);
    }
    // The same value before the clip (the caller tells 0 / 1 apart from the interior itself).
    __device__ __forceinline__ float one_fast_raw(uint32_t u, int c, double tv, double y) const
    {
        const double n = diff64(u, c), q0 = n * y;
        return (float)(fma(fma(-tv, q0, n), y, q0) + (double)a[c]);
    }
    // Four whole pixels p .. p+3 whose transmission is known to lie in [0.1, 1] (the guided filter's own clip, S6:180 /
    // ES:232): no range test on the divisor, values NOT clipped.  w: the 12 frame bytes, tv: the four divisors.
    __device__ __forceinline__ void four_raw(const uint32_t (&w)[3], const double (&tv)[4], float (&r)[3][4]) const
    {
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
        const uint32_t ur[4] = {w0 & 255, w0 >> 24, (w1 >> 16) & 255, (w2 >> 8) & 255};
        const uint32_t ug[4] = {(w0 >> 8) & 255, w1 & 255, w1 >> 24, (w2 >> 16) & 255};
        const uint32_t ub[4] = {(w0 >> 16) & 255, (w1 >> 8) & 255, w2 & 255, w2 >> 24};
        if constexpr (F32) {  // tv[i] carries the float32 transmission in its low word: see load_four
            typedef float f2 __attribute__((ext_vector_type(2)));
            float y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = recip32(__int_as_float(__double2loint(tv[i])));
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // two pixels per packed fused multiply-add
                const f2 yy = {y[2 * h], y[2 * h + 1]};
                const f2 r0 = __builtin_elementwise_fma(f2{diff(ur[2 * h], 0), diff(ur[2 * h + 1], 0)}, yy, f2{a[0], a[0]});
                const f2 r1 = __builtin_elementwise_fma(f2{diff(ug[2 * h], 1), diff(ug[2 * h + 1], 1)}, yy, f2{a[1], a[1]});
                const f2 r2 = __builtin_elementwise_fma(f2{diff(ub[2 * h], 2), diff(ub[2 * h + 1], 2)}, yy, f2{a[2], a[2]});
                r[0][2 * h] = r0.x; r[0][2 * h + 1] = r0.y;
                r[1][2 * h] = r1.x; r[1][2 * h + 1] = r1.y;
                r[2][2 * h] = r2.x; r[2][2 * h + 1] = r2.y;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double y = recip(tv[i]);
            r[0][i] = one_fast_raw(ur[i], 0, tv[i], y);
            r[1][i] = one_fast_raw(ug[i], 1, tv[i], y);
            r[2][i] = one_fast_raw(ub[i], 2, tv[i], y);
        }
    }
    __device__ __forceinline__ void load_four(int p, uint32_t (&w)[3], double (&tv)[4]) const
    {
        const u32_any *q = reinterpret_cast<const u32_any *>(img + (size_t)p * 3);
        w[0] = q[0]; w[1] = q[1]; w[2] = q[2];
        if constexpr (F32) {  // one 16-byte load; the four floats travel in the low words of tv (no conversion either way)
            typedef float4 __attribute__((aligned(4))) float4_a4;
            const float4_a4 f = *reinterpret_cast<const float4_a4 *>(tf() + p);
            tv[0] = __hiloint2double(0, __float_as_int(f.x));
            tv[1] = __hiloint2double(0, __float_as_int(f.y));
            tv[2] = __hiloint2double(0, __float_as_int(f.z));
            tv[3] = __hiloint2double(0, __float_as_int(f.w));
            return;
        }
        const double2_a8 ta = *reinterpret_cast<const double2_a8 *>(t + p), tb = *reinterpret_cast<const double2_a8 *>(t + p + 2);
        tv[0] = ta.x; tv[1] = ta.y; tv[2] = tb.x; tv[3] = tb.y;
    }
    // pixels p .. p+3 (n of them exist); any alignment
    __device__ __forceinline__ void four(int p, int n, float (&r)[3][4]) const
    {
        if constexpr (F32) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r[0][i] = r[1][i] = r[2][i] = 0.0f;
                if (i < n) pixel32(p + i, r[0][i], r[1][i], r[2][i]);
            }
            return;
        }
        Px4 v;
        double tv[4];
        if (n == 4) {
            const u32_any *w = reinterpret_cast<const u32_any *>(img + (size_t)p * 3);
            const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
            v.r[0] = w0 & 255; v.g[0] = (w0 >> 8) & 255; v.b[0] = (w0 >> 16) & 255;
            v.r[1] = w0 >> 24; v.g[1] = w1 & 255; v.b[1] = (w1 >> 8) & 255;
            v.r[2] = (w1 >> 16) & 255; v.g[2] = w1 >> 24; v.b[2] = w2 & 255;
            v.r[3] = (w2 >> 8) & 255; v.g[3] = (w2 >> 16) & 255; v.b[3] = w2 >> 24;
            const double2_a8 ta = *reinterpret_cast<const double2_a8 *>(t + p), tb = *reinterpret_cast<const double2_a8 *>(t + p + 2);
            tv[0] = ta.x; tv[1] = ta.y; tv[2] = tb.x; tv[3] = tb.y;
        } else {
            v = load_px4(img + (size_t)p * 3, n, false);
#pragma unroll
            for (int i = 0; i < 4; ++i) tv[i] = i < n ? t[p + i] : 1.0;
        }
        // one branch per group, so that the four pixels' straight-line fast paths can be interleaved
        if (recip_ok(tv[0]) && recip_ok(tv[1]) && recip_ok(tv[2]) && recip_ok(tv[3])) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double y = recip(tv[i]);
                r[0][i] = one_fast(v.r[i], 0, tv[i], y);
                r[1][i] = one_fast(v.g[i], 1, tv[i], y);
                r[2][i] = one_fast(v.b[i], 2, tv[i], y);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r[0][i] = one(v.r[i], 0, tv[i]);
                r[1][i] = one(v.g[i], 1, tv[i]);
                r[2][i] = one(v.b[i], 2, tv[i]);
            }
        }
    }
    __device__ __forceinline__ void pixel(int p, float &r0, float &r1, float &r2) const
    {
        if constexpr (F32) {
            pixel32(p, r0, r1, r2);
            return;
        }
        const uint8_t *q = img + (size_t)p * 3;
        const double tv = t[p];
        if (recip_ok(tv)) {
            const double y = recip(tv);
            r0 = one_fast(q[0], 0, tv, y);
            r1 = one_fast(q[1], 1, tv, y);
            r2 = one_fast(q[2], 2, tv, y);
        } else {
            r0 = one(q[0], 0, tv);
            r1 = one(q[1], 1, tv);
            r2 = one(q[2], 2, tv);
        }
    }
    // ---- ES surface (enhancement_strategies.py:237-249): the same expression kept in float64, clipped there
    __device__ __forceinline__ double one64(uint32_t u, int c, double tv) const
    {
        return fmin(fmax((double)diff(u, c) / tv + (double)a[c], 0.0), 1.0);
    }
    __device__ __forceinline__ double one64_fast(uint32_t u, int c, double tv, double y) const
    {
        const double n = (double)diff(u, c), q0 = n * y;
        return fmin(fmax(fma(fma(-tv, q0, n), y, q0) + (double)a[c], 0.0), 1.0);
    }
    __device__ __forceinline__ void four64(int p, int n, double (&r)[3][4]) const
    {
        Px4 v;
        double tv[4];
        if (n == 4) {
            v = load_px4_any(img + (size_t)p * 3);
            const double2_a8 ta = *reinterpret_cast<const double2_a8 *>(t + p), tb = *reinterpret_cast<const double2_a8 *>(t + p + 2);
            tv[0] = ta.x; tv[1] = ta.y; tv[2] = tb.x; tv[3] = tb.y;
        } else {
            v = load_px4(img + (size_t)p * 3, n, false);
#pragma unroll
            for (int i = 0; i < 4; ++i) tv[i] = i < n ? t[p + i] : 1.0;
        }
        if (recip_ok(tv[0]) && recip_ok(tv[1]) && recip_ok(tv[2]) && recip_ok(tv[3])) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double y = recip(tv[i]);
                r[0][i] = one64_fast(v.r[i], 0, tv[i], y);
                r[1][i] = one64_fast(v.g[i], 1, tv[i], y);
                r[2][i] = one64_fast(v.b[i], 2, tv[i], y);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r[0][i] = one64(v.r[i], 0, tv[i]);
                r[1][i] = one64(v.g[i], 1, tv[i]);
                r[2][i] = one64(v.b[i], 2, tv[i]);
            }
        }
    }
};

using RestoreImg = RestoreImgT<0>;
using RestoreImg32 = RestoreImgT<0, true>;

}  // namespace uwie
