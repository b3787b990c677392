// Per-pixel tail stages: restore_image (six_stadigy.py:183-188), the percentile stretch of
// enhance_contrast / white_balance (six_stadigy.py:198,218), gamma_correction (six_stadigy.py:222-224,
// enhancement_strategies.py:276-285) and the output quantisation (six_stadigy.py:430).
#include "common.h"
#include "devutil.h"
#include "restore.h"

namespace uwie {

namespace {

// result[:,:,c] = (img[:,:,c] - A[c]) / t + A[c]: float32 difference, float64 quotient and sum, float32 store; clip
// (restore.h: the same evaluation the fused tail uses, so the stage tests pin it).
__global__ void __launch_bounds__(256) k_restore(RestoreSrc S, int npx, float *__restrict__ out)
{
    const int b = blockIdx.y;
    RestoreImg R;
    R.init(S, b, (size_t)npx);  // (the stage entry point takes the float64 transmission only)
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npx; p += gridDim.x * 256) {
        float *o = out + ((size_t)b * npx + p) * 3;
        R.pixel(p, o[0], o[1], o[2]);
    }
}

__global__ void __launch_bounds__(256) k_stretch_apply(const float *__restrict__ img, const float *__restrict__ pct,
                                                       int pct_stride, int lo_idx, int hi_idx, float eps, int npx,
                                                       float *__restrict__ out)
{
    const int b = blockIdx.y;
    float lo[3];
    StretchDiv den[3];  // (devutil.h: the division's own operation sequence with the reciprocal kept)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        lo[c] = pct[(b * 3 + c) * pct_stride + lo_idx];
        den[c].set((pct[(b * 3 + c) * pct_stride + hi_idx] - lo[c]) + eps);
    }
    const size_t base = (size_t)b * npx * 3;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npx; p += gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = den[c].quot(img[base + (size_t)p * 3 + c] - lo[c]);
            out[base + (size_t)p * 3 + c] = fminf(fmaxf(v, 0.0f), 1.0f);
        }
    }
}

// float32 power: evaluated in float64 and rounded once (agrees with a correctly rounded powf except on
// astronomically rare double-rounding cases; the exponent is float32(g) like NumPy's weak Python scalar).
__device__ __forceinline__ float pow_f32(float x, float g) { return (float)pow((double)x, (double)g); }

__global__ void __launch_bounds__(256) k_gamma(const float *__restrict__ img, float *__restrict__ out, size_t n, float g,
                                               int clip)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = pow_f32_fast(img[i], g);
        if (clip) v = fminf(fmaxf(v, 0.0f), 1.0f);
        out[i] = v;
    }
}

__global__ void __launch_bounds__(256) k_quantise(const float *__restrict__ img, uint8_t *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = (uint8_t)quant_u8(img[i]);
}


}  // namespace

int launch_restore(const uint8_t *d_in, const int32_t *d_kind, const float *d_A, const double *d_t, Shape s,
                   float *d_out, hipStream_t st)
{
    UWIE_LAUNCH(k_restore, dim3(grid_for(s.npx()), s.B), dim3(256), 0, st, RestoreSrc{d_in, d_kind, d_A, d_t},
                (int)s.npx(), d_out);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

int launch_stretch_apply_f32(const float *d_img, const float *d_pct, int pct_stride, int lo_idx, int hi_idx, float eps,
                             float *d_out, Shape s, hipStream_t st)
{
    UWIE_LAUNCH(k_stretch_apply, dim3(grid_for(s.npx()), s.B), dim3(256), 0, st, d_img, d_pct, pct_stride, lo_idx,
                       hi_idx, eps, (int)s.npx(), d_out);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

int launch_gamma_f32(const float *d_img, float *d_out, size_t n, double g, int mode, hipStream_t st)
{
    UWIE_REQUIRE(mode == 1 || mode == 2, "gamma: mode must be 1 (x**g) or 2 (clip(x**(1/g)))");
    // mode 1: np.power(img, g) with img float32 -> exponent float32(g)      (six_stadigy.py:224)
    // mode 2: np.power(img, 1.0/g): the quotient is a Python float, cast to float32 for a float32 image (ES:284)
    const float e = mode == 1 ? (float)g : (float)(1.0 / g);
    UWIE_LAUNCH(k_gamma, dim3(grid_for(n)), dim3(256), 0, st, d_img, d_out, n, e, mode == 2 ? 1 : 0);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

int launch_quantise_u8(const float *d_img, uint8_t *d_out, size_t n, hipStream_t st)
{
    UWIE_LAUNCH(k_quantise, dim3(grid_for(n)), dim3(256), 0, st, d_img, d_out, n);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

}  // namespace uwie
