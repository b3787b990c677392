// estimate_transmission (six_stadigy.py:168-180, enhancement_strategies.py:209-234) and guided_filter
// (six_stadigy.py:26-46): the per-pixel initial transmission and the float64 guided filter built from six
// cv2.boxFilter(.., CV_64F, (r, r)) calls.
//
// cv2.boxFilter on CV_64F data is a running sum along each border-extended row (s += E[x+k] - E[x]) followed by
// a running sum down each column (s0 = SUM + Sp; out = s0 * 1/(k*k); SUM = s0 - Sm), so every output's rounding
// depends on the whole chain from the left / top edge.  This first implementation keeps those chains literally:
// one thread walks one row (k_box_rows) or one column (k_box_cols) in float64.  Anchor k/2 and
// BORDER_REFLECT_101 as in OpenCV; for even k the window is [x - k/2, x + k/2 - 1].
#include "common.h"
#include "devutil.h"

namespace uwie {

namespace {

// ---- sources for the row pass: fill v[NP] with the plane values at (b, y, x)
struct SrcPlanes1 {
    const double *p;
    int H, W;
    static constexpr int NP = 1;
    __device__ __forceinline__ void load(const double *, int b, int y, int x, double *v) const
    {
        v[0] = p[((size_t)b * H + y) * W + x];
    }
};
struct SrcPlanes2 {
    const double *p0, *p1;
    int H, W;
    static constexpr int NP = 2;
    __device__ __forceinline__ void load(const double *, int b, int y, int x, double *v) const
    {
        const size_t i = ((size_t)b * H + y) * W + x;
        v[0] = p0[i];
        v[1] = p1[i];
    }
};
struct SrcGuide {  // I = gray/255 (float64), p = t0 (float32 -> float64); planes I, p, I*p, I*I (six_stadigy.py:28-36)
    const uint8_t *gray;
    const float *t0;
    int H, W;
    static constexpr int NP = 4;
    __device__ __forceinline__ void load(const double *ilut, int b, int y, int x, double *v) const
    {
        const size_t i = ((size_t)b * H + y) * W + x;
        const double I = ilut[gray[i]], p = (double)t0[i];
        v[0] = I;
        v[1] = p;
        v[2] = I * p;
        v[3] = I * I;
    }
};

template <class Src>
__global__ void __launch_bounds__(64) k_box_rows(Src src, double *__restrict__ out, size_t plane_stride, int B, int k)
{
    constexpr int NP = Src::NP;
    __shared__ double ilut[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) ilut[i] = (double)i / 255.0;  // six_stadigy.py:177
    __syncthreads();
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    const int H = src.H, W = src.W;
    if (row >= B * H) return;
    const int b = row / H, y = row % H, a = k / 2;
    double s[NP], lead[NP], trail[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) s[p] = 0.0;
    for (int j = 0; j < k; ++j) {
        src.load(ilut, b, y, reflect101(j - a, W), lead);
#pragma unroll
        for (int p = 0; p < NP; ++p) s[p] += lead[p];
    }
    double *o = out + (size_t)row * W;
#pragma unroll
    for (int p = 0; p < NP; ++p) o[p * plane_stride] = s[p];
    for (int x = 0; x < W - 1; ++x) {
        src.load(ilut, b, y, reflect101(x + k - a, W), lead);
        src.load(ilut, b, y, reflect101(x - a, W), trail);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            s[p] += lead[p] - trail[p];
            o[p * plane_stride + x + 1] = s[p];
        }
    }
}

// ---- epilogues of the column pass
struct EpiStore1 {
    double *dst;
    static constexpr int NP = 1;
    __device__ __forceinline__ void operator()(const double *, size_t i, const double *m) const { dst[i] = m[0]; }
};
struct EpiAB {  // a = cov/(var+eps), b = mean_p - a*mean_I (six_stadigy.py:34-40)
    double *a, *b;
    double eps;
    static constexpr int NP = 4;
    __device__ __forceinline__ void operator()(const double *, size_t i, const double *m) const
    {
        const double cov = m[2] - m[0] * m[1];
        const double var = m[3] - m[0] * m[0];
        const double av = cov / (var + eps);
        a[i] = av;
        b[i] = m[1] - av * m[0];
    }
};
struct EpiQ {  // q = mean_a*I + mean_b, then np.clip(q, 0.1, 1.0) (six_stadigy.py:45,180)
    const uint8_t *gray;
    double *t;
    static constexpr int NP = 2;
    __device__ __forceinline__ void operator()(const double *ilut, size_t i, const double *m) const
    {
        const double q = m[0] * ilut[gray[i]] + m[1];
        t[i] = fmin(fmax(q, 0.1), 1.0);
    }
};

template <class Epi>
__global__ void __launch_bounds__(64) k_box_cols(const double *__restrict__ rs, size_t plane_stride, Epi epi, int B, int H,
                                                 int W, int k)
{
    constexpr int NP = Epi::NP;
    __shared__ double ilut[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) ilut[i] = (double)i / 255.0;
    __syncthreads();
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= B * W) return;
    const int b = col / W, x = col % W, a = k / 2;
    const double scale = 1.0 / ((double)k * (double)k);
    const double *base = rs + (size_t)b * H * W + x;
    double sum[NP], mean[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) sum[p] = 0.0;
    for (int j = 0; j < k - 1; ++j) {
        const size_t r = (size_t)reflect101(j - a, H) * W;
#pragma unroll
        for (int p = 0; p < NP; ++p) sum[p] += base[p * plane_stride + r];
    }
    for (int y = 0; y < H; ++y) {
        const size_t rp = (size_t)reflect101(y + k - 1 - a, H) * W, rm = (size_t)reflect101(y - a, H) * W;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const double s0 = sum[p] + base[p * plane_stride + rp];
            mean[p] = s0 * scale;
            sum[p] = s0 - base[p * plane_stride + rm];
        }
        epi(ilut, ((size_t)b * H + y) * W + x, mean);
    }
}

// first half of estimate_transmission: six_stadigy.py:170-174 / enhancement_strategies.py:221-225
__global__ void __launch_bounds__(256) k_trans_init(const uint8_t *__restrict__ in, const int32_t *__restrict__ kind,
                                                    const float *__restrict__ A, int npx, float omega, float norm_eps,
                                                    int pre_clip, float *__restrict__ t0)
{
    const int b = blockIdx.y;
    const int k = kind ? kind[b] : 0;
    const float d0 = A[b * 3 + 0] + norm_eps, d1 = A[b * 3 + 1] + norm_eps, d2 = A[b * 3 + 2] + norm_eps;
    const uint8_t *img = in + (size_t)b * npx * 3;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npx; p += gridDim.x * 256) {
        const uint8_t *q = img + (size_t)p * 3;
        const float n0 = px_val(q[0], false) / d0;
        const float n1 = px_val(q[1], px_atten(k, 1)) / d1;
        const float n2 = px_val(q[2], px_atten(k, 2)) / d2;
        const float dark = fminf(fminf(n0, n1), n2);
        float t = 1.0f - omega * dark;
        if (pre_clip) t = fminf(fmaxf(t, 0.1f), 1.0f);
        t0[(size_t)b * npx + p] = t;
    }
}

}  // namespace

int launch_trans_init(const uint8_t *d_in, const int32_t *d_kind, const float *d_A, Shape s, float omega, float norm_eps,
                      int pre_clip, float *d_t0, hipStream_t st)
{
    const int blocks = grid_for(s.npx(), 4096);
    UWIE_LAUNCH(k_trans_init, dim3(blocks, s.B), dim3(256), 0, st, d_in, d_kind, d_A, (int)s.npx(), omega,
                       norm_eps, pre_clip, d_t0);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

size_t box_ws_bytes(Shape s)
{
    Carver c(nullptr);
    c.take<double>((size_t)s.B * s.npx());
    return c.total();
}

int launch_box_filter_f64(const double *d_src, double *d_dst, Shape s, int k, void *ws, hipStream_t st)
{
    Carver c(ws);
    const size_t n = (size_t)s.B * s.npx();
    double *rs = c.take<double>(n);
    UWIE_LAUNCH(k_box_rows<SrcPlanes1>, dim3(cdiv((long long)s.B * s.H, 64)), dim3(64), 0, st,
                       SrcPlanes1{d_src, s.H, s.W}, rs, n, s.B, k);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_box_cols<EpiStore1>, dim3(cdiv((long long)s.B * s.W, 64)), dim3(64), 0, st, rs, n,
                       EpiStore1{d_dst}, s.B, s.H, s.W, k);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

size_t guided_ws_bytes(Shape s)
{
    Carver c(nullptr);
    c.take<double>((size_t)s.B * s.npx() * 6);
    return c.total();
}

int launch_guided(const uint8_t *d_gray, const float *d_t0, Shape s, int k, double eps, double *d_t, void *ws,
                  hipStream_t st)
{
    Carver c(ws);
    const size_t n = (size_t)s.B * s.npx();
    double *rs = c.take<double>(n * 6);  // 4 row-sum planes + a + b
    double *pa = rs + 4 * n, *pb = rs + 5 * n;
    const dim3 grows(cdiv((long long)s.B * s.H, 64)), gcols(cdiv((long long)s.B * s.W, 64)), blk(64);
    UWIE_LAUNCH(k_box_rows<SrcGuide>, grows, blk, 0, st, SrcGuide{d_gray, d_t0, s.H, s.W}, rs, n, s.B, k);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_box_cols<EpiAB>, gcols, blk, 0, st, rs, n, EpiAB{pa, pb, eps}, s.B, s.H, s.W, k);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_box_rows<SrcPlanes2>, grows, blk, 0, st, SrcPlanes2{pa, pb, s.H, s.W}, rs, n, s.B, k);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_box_cols<EpiQ>, gcols, blk, 0, st, rs, n, EpiQ{d_gray, d_t}, s.B, s.H, s.W, k);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

}  // namespace uwie
