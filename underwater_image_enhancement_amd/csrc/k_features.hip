// vgg_16_UIE.extract_all_features (vgg_16_UIE.py:435-466) for u8 frames: per channel mean, std, min, max, median of
// img[:, :, c]; then mean(img), std(img), mean(img ** 2) of the whole float32 array; zero padding to 79 values.
// The float statistics follow NumPy's float32 arithmetic (pairwise_tree.h):
//   np.mean = float32(float64(np.sum) / n);  np.std = sqrt(sum((x - float32 mean)^2) / float32(n));
//   np.median of an even count = float32(float64(float32(a + b)) / 2) of the two middle order statistics.
// Per-channel sums reuse the quadtree's region kernels on one full-frame region per image; the whole-array sums walk the
// interleaved byte stream (3n elements) in the same 8192-element buffers; min / max / median come from the 256-bin
// histogram of each channel (x = u8/255 is monotone in the byte).
#include "common.h"
#include "devutil.h"
#include "pairwise_tree.h"

namespace uwie {

namespace {

constexpr int kNFeat = 79;
typedef uint4 __attribute__((aligned(1))) u128_unaligned;

// MODE 0: {x, x*x, 0}   MODE 1: {(x - mean)^2, 0, 0}
template <int MODE>
__device__ __forceinline__ void flat_val(uint32_t byte, float mean, float v[3])
{
    const float x = px_norm_fast(byte);
    if (MODE == 0) {
        v[0] = x;
        v[1] = x * x;
        v[2] = 0.0f;
    } else {
        const float d = x - mean;
        v[0] = d * d;
        v[1] = 0.0f;
        v[2] = 0.0f;
    }
}

// NumPy's pairwise_sum on n <= 128 consecutive bytes
template <int MODE>
__device__ void flat_leaf(const uint8_t *p, int n, float mean, float out[3])
{
    float v[3];
    if (n < 8) {
        float a[3] = {0.f, 0.f, 0.f};
        for (int i = 0; i < n; ++i) {
            flat_val<MODE>(p[i], mean, v);
            a[0] += v[0]; a[1] += v[1]; a[2] += v[2];
        }
        out[0] = a[0]; out[1] = a[1]; out[2] = a[2];
        return;
    }
    float acc[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        flat_val<MODE>(p[j], mean, v);
        acc[0][j] = v[0]; acc[1][j] = v[1]; acc[2][j] = v[2];
    }
    int i = 8;
    for (; i < n - (n % 8); i += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            flat_val<MODE>(p[i + j], mean, v);
            acc[0][j] += v[0]; acc[1][j] += v[1]; acc[2][j] += v[2];
        }
    float res[3] = {tree8(acc[0]), tree8(acc[1]), tree8(acc[2])};
    for (; i < n; ++i) {
        flat_val<MODE>(p[i], mean, v);
        res[0] += v[0]; res[1] += v[1]; res[2] += v[2];
    }
    out[0] = res[0]; out[1] = res[1]; out[2] = res[2];
}

// One wavefront per 8192-element buffer of the byte stream of image blockIdx.y.  csum[(b*nch + chunk)*3 + k].
template <int MODE>
__global__ void __launch_bounds__(64) k_flat_chunk_sums(const uint8_t *__restrict__ in, long long N, int nch,
                                                        const float *__restrict__ mean, float *__restrict__ csum)
{
    __shared__ PairwiseTree tree;
    const int b = blockIdx.y, ci = blockIdx.x, lane = threadIdx.x;
    const uint8_t *base = in + (size_t)b * N + (size_t)ci * kNpChunk;
    const int len = (int)min((long long)kNpChunk, N - (long long)ci * kNpChunk);
    const float m = MODE == 1 ? mean[b] : 0.0f;
    float *out = csum + ((size_t)b * nch + ci) * 3;
    if (len == kNpChunk) {  // balanced tree: 64 leaves of 128, lane = leaf, butterfly == recursive halving
        uint32_t raw[32];
        const u128_unaligned *w = reinterpret_cast<const u128_unaligned *>(base + lane * 128);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint4 v = w[q];
            raw[4 * q] = v.x; raw[4 * q + 1] = v.y; raw[4 * q + 2] = v.z; raw[4 * q + 3] = v.w;
        }
        float acc[3][8], v[3];
#pragma unroll
        for (int it = 0; it < 16; ++it)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = 8 * it + j;
                flat_val<MODE>((raw[e >> 2] >> (8 * (e & 3))) & 0xffu, m, v);
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c][j] = it == 0 ? v[c] : acc[c][j] + v[c];
            }
        float s[3] = {tree8(acc[0]), tree8(acc[1]), tree8(acc[2])};
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            s[0] += __shfl_xor(s[0], o);
            s[1] += __shfl_xor(s[1], o);
            s[2] += __shfl_xor(s[2], o);
        }
        if (lane == 0) { out[0] = s[0]; out[1] = s[1]; out[2] = s[2]; }
        return;
    }
    float res[3];
    pairwise_ragged(len, lane, tree, [&](int off, int l, float *s3) { flat_leaf<MODE>(base + off, l, m, s3); }, res);
    if (lane == 0) { out[0] = res[0]; out[1] = res[1]; out[2] = res[2]; }
}

// Sequential accumulation of the buffer sums, one wavefront per image: tot[b*3 + k]; MODE 0 also the flat mean.
template <int MODE>
__global__ void __launch_bounds__(64) k_flat_combine(const float *__restrict__ csum, int nch, long long N,
                                                     float *__restrict__ tot, float *__restrict__ mean)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (lane >= 3) return;
    float acc = 0.0f;
    for (int k = 0; k < nch; ++k) acc = acc + csum[((size_t)b * nch + k) * 3 + lane];
    tot[b * 3 + lane] = acc;
    if (MODE == 0 && lane == 0) mean[b] = (float)((double)acc / (double)N);
}

__global__ void k_full_frame_regions(Region *regs, int B, int H, int W)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) regs[b] = Region{b, 0, 0, H, W};
}

// value of sorted position `r` (0-based) of a channel, from its histogram
__device__ __forceinline__ int code_at_rank(const uint32_t *h, uint32_t r)
{
    uint32_t acc = 0;
    for (int k = 0; k < 256; ++k) {
        acc += h[k];
        if (r < acc) return k;
    }
    return 255;
}

__global__ void k_features_finish(const float *__restrict__ mean_c, const float *__restrict__ vtot_c,
                                  const uint32_t *__restrict__ hist, const float *__restrict__ flat0,
                                  const float *__restrict__ flat1, int npx, float *__restrict__ out)
{
    const int b = blockIdx.x;
    float *f = out + (size_t)b * kNFeat;
    const uint32_t n = (uint32_t)npx;
    for (int c = 0; c < 3; ++c) {
        const uint32_t *h = hist + ((size_t)b * 3 + c) * 256;
        int lo = 0, hi = 255;
        while (lo < 255 && h[lo] == 0) ++lo;
        while (hi > 0 && h[hi] == 0) --hi;
        float med;
        if (n & 1) {
            med = px_norm(code_at_rank(h, n / 2));
        } else {
            const float a = px_norm(code_at_rank(h, n / 2 - 1)), bb = px_norm(code_at_rank(h, n / 2));
            med = (float)((double)(a + bb) / 2.0);
        }
        f[5 * c + 0] = mean_c[b * 3 + c];
        f[5 * c + 1] = sqrtf(vtot_c[b * 3 + c] / (float)npx);
        f[5 * c + 2] = px_norm(lo);
        f[5 * c + 3] = px_norm(hi);
        f[5 * c + 4] = med;
    }
    const double N = 3.0 * (double)npx;
    f[15] = (float)((double)flat0[b * 3 + 0] / N);
    f[16] = sqrtf(flat1[b * 3 + 0] / (float)(3ll * npx));
    f[17] = (float)((double)flat0[b * 3 + 1] / N);
    for (int i = 18; i < kNFeat; ++i) f[i] = 0.0f;
}

struct FeatBufs {
    Region *regs;
    float *csum, *tot, *mean, *vtot, *fcsum, *flat0, *flat1, *fmean;
    uint32_t *hist;
    int maxChunks, fnch;
};

FeatBufs carve_features(Carver &c, Shape s)
{
    FeatBufs f;
    f.maxChunks = cdiv((long long)s.npx(), kNpChunk);
    f.fnch = cdiv(3ll * (long long)s.npx(), kNpChunk);
    f.regs = c.take<Region>(s.B);
    f.csum = c.take<float>((size_t)s.B * f.maxChunks * 3);
    f.tot = c.take<float>((size_t)s.B * 3);
    f.mean = c.take<float>((size_t)s.B * 3);
    f.vtot = c.take<float>((size_t)s.B * 3);
    f.fcsum = c.take<float>((size_t)s.B * f.fnch * 3);
    f.flat0 = c.take<float>((size_t)s.B * 3);
    f.flat1 = c.take<float>((size_t)s.B * 3);
    f.fmean = c.take<float>((size_t)s.B);
    f.hist = c.take<uint32_t>((size_t)s.B * 768);
    return f;
}

}  // namespace

size_t features_ws_bytes(Shape s)
{
    Carver c(nullptr);
    carve_features(c, s);
    return c.total();
}

int launch_features_u8(const uint8_t *d_in, Shape s, float *d_out, void *ws, hipStream_t st)
{
    Carver c(ws);
    FeatBufs f = carve_features(c, s);
    const long long N = 3ll * (long long)s.npx();
    UWIE_LAUNCH(k_full_frame_regions, dim3(cdiv(s.B, 64)), dim3(64), 0, st, f.regs, s.B, s.H, s.W);
    UWIE_LAUNCH_CHECK();
    int rc = launch_region_stats(d_in, nullptr, f.regs, s.B, s.H, s.W, s, f.csum, f.maxChunks, f.tot, f.mean, f.vtot, st);
    if (rc != UWIE_OK) return rc;
    UWIE_HIP_CHECK(hipMemsetAsync(f.hist, 0, sizeof(uint32_t) * (size_t)s.B * 768, st));
    rc = launch_frame_hist(d_in, s, f.hist, st);
    if (rc != UWIE_OK) return rc;
    UWIE_LAUNCH(k_flat_chunk_sums<0>, dim3(f.fnch, s.B), dim3(64), 0, st, d_in, N, f.fnch, f.fmean, f.fcsum);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_flat_combine<0>, dim3(s.B), dim3(64), 0, st, f.fcsum, f.fnch, N, f.flat0, f.fmean);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_flat_chunk_sums<1>, dim3(f.fnch, s.B), dim3(64), 0, st, d_in, N, f.fnch, f.fmean, f.fcsum);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_flat_combine<1>, dim3(s.B), dim3(64), 0, st, f.fcsum, f.fnch, N, f.flat1, f.fmean);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_features_finish, dim3(s.B), dim3(1), 0, st, f.mean, f.vtot, f.hist, f.flat0, f.flat1, (int)s.npx(), d_out);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

}  // namespace uwie
