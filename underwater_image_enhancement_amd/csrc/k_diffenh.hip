// DifferentiableEnhancement.forward (vgg_16_UIE.py:32-128, the `enhance_image` surface of use_trained_model.py:83-111)
// after the order statistics: colour stretch between two sorted positions per channel -> simplified dark-channel
// dehazing with A = 0.6 -> gamma -> clamp.  float32 throughout, one operation per PyTorch operation (torch evaluates
// `tensor op python_scalar` in float32 with the scalar converted to float32).
//   stretch  (channel - p_low) / (p_high - p_low + 1e-8), clamp 0..1                     vgg_16_UIE.py:88-91
//   dehaze   dark = min_c; t = clamp(1 - omega*dark, 0.1, 1); clamp((img - 0.6)/t + 0.6)  vgg_16_UIE.py:103-117
//   gamma    pow(img + 1e-8, gamma)                                                       vgg_16_UIE.py:127
// pow is evaluated in float64 and rounded once; torch's float32 pow (Sleef, 1 ulp) may differ by one ulp: stated
// tolerance of the gamma stage.
#include "common.h"
#include "devutil.h"

namespace uwie {

namespace {

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// planar: img/out [B][3][n];  interleaved: [B][n][3].  os: [B*3][kSelOsStride] floats, entries 0/1 = p_low, p_high.
__global__ void __launch_bounds__(256) k_diff_enhance(const float *__restrict__ img, int planar, int n,
                                                      const float *__restrict__ params, int flags,
                                                      const float *__restrict__ os, float *__restrict__ out)
{
    const int b = blockIdx.y;
    const float *pr = params + b * 4;
    float lo[3], rng[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float *o = os + (size_t)(b * 3 + c) * kSelOsStride;
        lo[c] = o[0];
        rng[c] = (o[1] - o[0]) + 1e-8f;
    }
    const float omega = pr[2], gamma = pr[3];
    const size_t base = (size_t)b * 3 * n;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = planar ? img[base + (size_t)c * n + p] : img[base + (size_t)p * 3 + c];
            v[c] = clamp01((x - lo[c]) / rng[c]);
        }
        if (flags & 1) {
            const float dark = fminf(fminf(v[0], v[1]), v[2]);
            const float t = fminf(fmaxf(1.0f - omega * dark, 0.1f), 1.0f);
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = clamp01((v[c] - 0.6f) / t + 0.6f);
        }
        if (flags & 2) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = pow_f32_fast(v[c] + 1e-8f, gamma);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float y = clamp01(v[c]);
            if (planar) out[base + (size_t)c * n + p] = y;
            else out[base + (size_t)p * 3 + c] = y;
        }
    }
}

}  // namespace

int launch_diff_enhance(const float *d_img, int planar, Shape s, const float *d_params, int flags, const float *d_os,
                        float *d_out, hipStream_t st)
{
    const int n = (int)s.npx();
    UWIE_LAUNCH(k_diff_enhance, dim3(grid_for(n, 4096), s.B), dim3(256), 0, st, d_img, planar, n, d_params, flags, d_os,
                d_out);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

}  // namespace uwie
