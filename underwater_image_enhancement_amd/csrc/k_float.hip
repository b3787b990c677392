// Front half of the strategies for GENERAL float images (six_stadigy.py:230-285 on float32 HxWx3, enhancement_strategies.py
// :477-508 on float32 or float64 HxWx3): the reference's signature is "float image in [0, 1]", and its own harnesses feed
// np.random.rand (enhancement_strategies.py:516, example_usage.py:27,44,112).  The fast path of this library starts from
// the u8 frame (every value of a u8-derived image is a function of its byte, which is what its kernels are built on); an
// image that is NOT u8-derived takes the kernels here -- the same arithmetic with the pixel values read from the float
// image itself -- and then the materialised-image stage kernels (k_select / k_tail / k_clahe / k_codes).  Written for
// parity, not speed: nothing here is fused.
//   k_f_prepare      [color_correction (S6:305-323)] and (x * 255).astype(uint8) (S6:149,177,204, ES:180,228,299,339)
//   k_f_mean_seq     img.mean(axis=(0,1)): sequential accumulation per channel in the image's dtype (S6:294)
//   k_fq_chunk_sums  compute_Q's np.sum / np.mean / squared deviations in NumPy's pairwise order (S6:134-147)
//   k_fq_select      compute_Q's score and the greedy step (S6:100-111,152-155)
//   k_f_brightest    get_brightest_pixel (S6:160-165)
//   k_f_trans_init   1 - omega * min_c(img / (A + eps)) [clip] (S6:170-174, ES:221-225)
//   k_f_restore      restore_image (S6:183-188) / recover_image (ES:237-249)
#include "common.h"
#include "devutil.h"
#include "pairwise_tree.h"

namespace uwie {

namespace {

constexpr int kMaxLevelsF = 32;

template <class T>
__global__ void k_f_prepare(const T *__restrict__ x, const int32_t *__restrict__ kind, T *__restrict__ xc, uint8_t *__restrict__ q,
                            size_t npx, int B)
{
    const size_t n = npx * 3 * (size_t)B;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        const int k = kind ? kind[i / (npx * 3)] : 0;
        T v = x[i];
        if (k != 0) {  // color_correction: the dominant channel * 0.85, then np.clip(img, 0, 1) on every channel
            if (c == k) v = v * (T)0.85;
            v = v < (T)0 ? (T)0 : v > (T)1 ? (T)1 : v;
        }
        if (xc) xc[i] = v;
        if (q) q[i] = (uint8_t)((int)(v * (T)255) & 0xff);
    }
}

// one lane per (image, channel): the reduction over axes (0, 1) of an HxWx3 array adds the rows of 3-vectors one after
// the other, i.e. sequentially per channel (measured, NumPy 2.2.6).  The quotient sum / count is taken in
// float64 and cast back: `_mean` divides the float32 sums by an np.intp count, which selects the float64 loop (checked against
// NumPy on 2^24 + 3 pixels, where float32(count) is no longer the count: float64 division reproduces img.mean, float32 does not).
// Contract of the float-image path: finite values (the reference documents [0, 1]); NaN handling is not reproduced (fmin /
// fmax drop a NaN that np.min / np.clip would propagate).  The chain is one dependent addition per pixel and lane -- 8 M at
// 4K -- which is why u8-derived images never come here (api.py routes them to the closed-form cast detection of k_entry.hip).
template <class T>
__global__ void k_f_mean_seq(const T *__restrict__ x, size_t npx, int B, float *__restrict__ mean_out, int32_t *__restrict__ kind)
{
    const int b = blockIdx.x, c = threadIdx.x;
    __shared__ double m[3];
    if (c < 3) {
        const T *p = x + (size_t)b * npx * 3 + c;
        T acc = 0;
        for (size_t i = 0; i < npx; ++i) acc = acc + p[3 * i];
        m[c] = (double)(T)((double)acc / (double)npx);  // _mean: the sum divided in float64, cast back
        if (mean_out) mean_out[b * 3 + c] = (float)m[c];
    }
    __syncthreads();
    if (c == 0 && kind) {
        const T r = (T)m[0], g = (T)m[1], bl = (T)m[2];
        int k = 0;
        if (g > r && g > bl && (g - r) > (T)0.05) k = 1;       // S6:297
        else if (bl > r && bl > g && (bl - r) > (T)0.05) k = 2;  // S6:299
        kind[b] = k;
    }
}

// ---- quadtree statistics on a float image
template <class T, bool VAR>
struct ElemF {
    const T *img;  // one image, HWC
    int W;
    T mean[3];
    __device__ __forceinline__ T get(const T *p, int c) const
    {
        T v = p[c];
        if (VAR) {
            const T d = v - mean[c];
            v = d * d;
        }
        return v;
    }
};

// NumPy's pairwise_sum on a block of n <= 128 elements starting at raster element e0 of region r, three channels at once
template <class T, bool VAR>
__device__ void leaf_sum3f(const ElemF<T, VAR> &el, const Region &r, int e0, int n, T out[3])
{
    int ly = e0 / r.cols, lx = e0 % r.cols;
    const T *p = el.img + ((size_t)(r.y0 + ly) * el.W + r.x0 + lx) * 3;
    auto step = [&]() {
        ++lx;
        p += 3;
        if (lx == r.cols) {
            lx = 0;
            p += (size_t)(el.W - r.cols) * 3;
        }
    };
    if (n < 8) {
        T a0 = 0, a1 = 0, a2 = 0;
        for (int i = 0; i < n; ++i) {
            a0 += el.get(p, 0);
            a1 += el.get(p, 1);
            a2 += el.get(p, 2);
            step();
        }
        out[0] = a0; out[1] = a1; out[2] = a2;
        return;
    }
    T acc[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c][j] = el.get(p, c);
        step();
    }
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c][j] += el.get(p, c);
            step();
        }
    }
    T res[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) res[c] = tree8<T>(acc[c]);
    for (; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < 3; ++c) res[c] += el.get(p, c);
        step();
    }
    out[0] = res[0]; out[1] = res[1]; out[2] = res[2];
}

// one wavefront per (chunk, region); VAR: the means are derived from the sums pass's chunk sums (csum_in)
template <class T, bool VAR>
__global__ void __launch_bounds__(64) k_fq_chunk_sums(const T *__restrict__ in, const Region *__restrict__ regs, int H, int W,
                                                      int maxChunks, T *__restrict__ csum, const T *__restrict__ csum_in)
{
    const int reg = blockIdx.y, ci = blockIdx.x, lane = threadIdx.x;
    const Region r = regs[reg];
    const int n = r.rows * r.cols;
    const int c0 = ci * kNpChunk;
    if (c0 >= n) return;
    const int len = min(kNpChunk, n - c0);
    __shared__ PairwiseTreeT<T> tree;
    __shared__ T s_mean[3];
    ElemF<T, VAR> el;
    el.img = in + (size_t)r.img * H * W * 3;
    el.W = W;
    if (VAR) {
        if (lane < 3) {
            const int nch = (n + kNpChunk - 1) / kNpChunk;
            const T *cs = csum_in + (size_t)reg * maxChunks * 3 + lane;
            T acc = 0;
            for (int k = 0; k < nch; ++k) acc = acc + cs[k * 3];
            s_mean[lane] = (T)((double)acc / (double)n);
        }
        __syncthreads();
        el.mean[0] = s_mean[0]; el.mean[1] = s_mean[1]; el.mean[2] = s_mean[2];
    }
    T *out = csum + ((size_t)reg * maxChunks + ci) * 3;
    T res[3];
    if (len == kNpChunk) {
        leaf_sum3f<T, VAR>(el, r, c0 + lane * 128, 128, res);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if constexpr (sizeof(T) == 4) res[c] += __shfl_xor(res[c], o);
                else res[c] += __longlong_as_double((long long)shfl_xor_u64((uint64_t)__double_as_longlong(res[c]), o));
            }
        }
    } else {
        pairwise_ragged<T>(len, lane, tree, [&](int off, int l, T *s3) { leaf_sum3f<T, VAR>(el, r, c0 + off, l, s3); }, res);
    }
    if (lane == 0) { out[0] = res[0]; out[1] = res[1]; out[2] = res[2]; }
}

struct TraceRecF {
    int32_t y0, x0, rows, cols;
    double score[4];
};

template <class T>
__global__ void __launch_bounds__(64) k_fq_select(Region *__restrict__ blk, Region *__restrict__ regs, const T *__restrict__ csum,
                                                  const T *__restrict__ csum_var, int maxChunks, uint32_t *__restrict__ edges,
                                                  int level, int min_size, TraceRecF *__restrict__ trace)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (regs[b * 4].rows == 0) return;  // leaf reached earlier
    T acc = 0;
    if (lane < 24) {
        const int q = lane / 6, pass = (lane % 6) / 3, c = lane % 3;
        const Region r = regs[b * 4 + q];
        const int n = r.rows * r.cols, nch = (n + kNpChunk - 1) / kNpChunk;
        const T *cs = (pass ? csum_var : csum) + (size_t)(b * 4 + q) * maxChunks * 3 + c;
        for (int k = 0; k < nch; ++k) acc = acc + cs[k * 3];
    }
    __shared__ T tots[24];
    if (lane < 24) tots[lane] = acc;
    __syncthreads();
    if (lane != 0) return;
    double best = 0.0;
    int arg = 0;
    double score[4];
    for (int q = 0; q < 4; ++q) {
        const Region r = regs[b * 4 + q];
        const long long n = (long long)r.rows * r.cols;
        const T *S = tots + q * 6, *V = tots + q * 6 + 3;
        const T t1 = ((S[0] + S[1]) + S[2]) / (T)(3 * n);
        const T t2 = ((S[2] + S[1]) - (T)2 * S[0]) / (T)n;
        const T v0 = V[0] / (T)n, v1 = V[1] / (T)n, v2 = V[2] / (T)n;
        const T t3 = ((v0 + v1) + v2) / (T)3;
        const double t4 = (double)edges[b * 4 + q] / (double)n;
        const double Q = (double)((t1 + t2) - t3) - t4;
        score[q] = Q;
        if (q == 0 || Q > best) { best = Q; arg = q; }
    }
    if (trace) {
        TraceRecF &t = trace[b * kMaxLevelsF + level];
        const Region k = blk[b];
        t.y0 = k.y0; t.x0 = k.x0; t.rows = k.rows; t.cols = k.cols;
        for (int q = 0; q < 4; ++q) t.score[q] = score[q];
    }
    const Region k = regs[b * 4 + arg];
    blk[b] = k;
    const bool leaf = k.rows <= min_size || k.cols <= min_size;
    const int mr = k.rows / 2, mc = k.cols / 2;
    Region q[4] = {{b, k.y0, k.x0, mr, mc},
                   {b, k.y0, k.x0 + mc, mr, k.cols - mc},
                   {b, k.y0 + mr, k.x0, k.rows - mr, mc},
                   {b, k.y0 + mr, k.x0 + mc, k.rows - mr, k.cols - mc}};
    for (int i = 0; i < 4; ++i) {
        if (leaf) q[i].rows = q[i].cols = 0;
        regs[b * 4 + i] = q[i];
        edges[b * 4 + i] = 0;
    }
}

__global__ void k_f_init_blocks(Region *blk, Region *regs, uint32_t *edges, int B, int H, int W, int min_size)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const Region k{b, 0, 0, H, W};
    blk[b] = k;
    const bool leaf = k.rows <= min_size || k.cols <= min_size;
    const int mr = k.rows / 2, mc = k.cols / 2;
    Region q[4] = {{b, k.y0, k.x0, mr, mc},
                   {b, k.y0, k.x0 + mc, mr, k.cols - mc},
                   {b, k.y0 + mr, k.x0, k.rows - mr, mc},
                   {b, k.y0 + mr, k.x0 + mc, k.rows - mr, k.cols - mc}};
    for (int i = 0; i < 4; ++i) {
        if (leaf) q[i].rows = q[i].cols = 0;
        regs[b * 4 + i] = q[i];
        edges[b * 4 + i] = 0;
    }
}

// argmax of (r+g)+b over the leaf (np.sum(block, axis=2) in the image's dtype), first maximum in raster order
template <class T>
__global__ void __launch_bounds__(64) k_f_brightest(const T *__restrict__ in, const Region *__restrict__ blk, int H, int W,
                                                    T *__restrict__ A)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const Region r = blk[b];
    const T *img = in + (size_t)b * H * W * 3;
    const int n = r.rows * r.cols;
    __shared__ T s_best[64];
    __shared__ int s_idx[64];
    T best = 0;
    int bi = 0x7fffffff;
    bool have = false;
    for (int e = lane; e < n; e += 64) {
        const T *p = img + ((size_t)(r.y0 + e / r.cols) * W + r.x0 + e % r.cols) * 3;
        const T v = (p[0] + p[1]) + p[2];
        if (!have || v > best) { best = v; bi = e; have = true; }
    }
    s_best[lane] = best;
    s_idx[lane] = have ? bi : 0x7fffffff;
    __syncthreads();
    if (lane == 0) {
        T bv = 0;
        int bidx = 0x7fffffff;
        for (int l = 0; l < 64; ++l) {
            if (s_idx[l] == 0x7fffffff) continue;
            if (bidx == 0x7fffffff || s_best[l] > bv || (s_best[l] == bv && s_idx[l] < bidx)) { bv = s_best[l]; bidx = s_idx[l]; }
        }
        const T *p = img + ((size_t)(r.y0 + bidx / r.cols) * W + r.x0 + bidx % r.cols) * 3;
        A[b * 3 + 0] = p[0];
        A[b * 3 + 1] = p[1];
        A[b * 3 + 2] = p[2];
    }
}

template <class T>
__global__ void k_f_trans_init(const T *__restrict__ x, const T *__restrict__ A, size_t npx, int B, T omega, T norm_eps, int pre_clip,
                               T *__restrict__ t0)
{
    const size_t n = npx * (size_t)B;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / npx);
        const T *p = x + i * 3, *a = A + b * 3;
        const T n0 = p[0] / (a[0] + norm_eps), n1 = p[1] / (a[1] + norm_eps), n2 = p[2] / (a[2] + norm_eps);
        const T dark = fmin(fmin(n0, n1), n2);  // np.min over the channel axis
        T t = (T)1 - omega * dark;
        if (pre_clip) t = t < (T)0.1 ? (T)0.1 : t > (T)1 ? (T)1 : t;
        t0[i] = t;
    }
}

// S6 (OUT = float, HWC): float64 arithmetic, float32 store, clip.  ES (OUT = double, planar): the difference in the
// image's dtype, everything else float64, clip.
template <class T, class OUT>
__global__ void k_f_restore(const T *__restrict__ x, const T *__restrict__ A, const double *__restrict__ t, size_t npx, int B,
                            OUT *__restrict__ out, int planar)
{
    const size_t n = npx * (size_t)B;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / npx);
        const size_t pi = i - (size_t)b * npx;
        const double tv = t[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const T d = x[i * 3 + c] - A[b * 3 + c];
            OUT v = (OUT)((double)d / tv + (double)A[b * 3 + c]);
            v = v < (OUT)0 ? (OUT)0 : v > (OUT)1 ? (OUT)1 : v;
            if (planar) out[((size_t)b * 3 + c) * npx + pi] = v;
            else out[i * 3 + c] = v;
        }
    }
}

template <class T>
struct FloatLevel {
    Region *blk, *regs;
    T *csum, *csum_var;
    uint32_t *edges;
};

int max_chunks_f(Shape s) { return cdiv((long long)((s.H + 1) / 2) * ((s.W + 1) / 2), kNpChunk); }

template <class T>
FloatLevel<T> carve_float_level(Carver &c, Shape s)
{
    FloatLevel<T> L;
    const size_t nreg = (size_t)s.B * 4;
    L.blk = c.take<Region>(s.B);
    L.regs = c.take<Region>(nreg);
    L.csum = c.take<T>(nreg * max_chunks_f(s) * 3);
    L.csum_var = c.take<T>(nreg * max_chunks_f(s) * 3);
    L.edges = c.take<uint32_t>(nreg);
    return L;
}

}  // namespace

size_t float_airlight_ws_bytes(Shape s)
{
    Carver c(nullptr);
    carve_float_level<double>(c, s);
    return c.total() + canny_ws_bytes(s);
}

template <class T>
int launch_float_prepare(const T *d_x, const int32_t *d_kind, T *d_xc, uint8_t *d_q, Shape s, hipStream_t st)
{
    UWIE_LAUNCH((k_f_prepare<T>), dim3(grid_for((size_t)s.B * s.npx() * 3)), dim3(256), 0, st, d_x, d_kind, d_xc, d_q, s.npx(), s.B);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}
template int launch_float_prepare<float>(const float *, const int32_t *, float *, uint8_t *, Shape, hipStream_t);
template int launch_float_prepare<double>(const double *, const int32_t *, double *, uint8_t *, Shape, hipStream_t);

template <class T>
int launch_float_cast_classify(const T *d_x, Shape s, int32_t *d_kind, float *d_mean, hipStream_t st)
{
    UWIE_LAUNCH((k_f_mean_seq<T>), dim3(s.B), dim3(64), 0, st, d_x, s.npx(), s.B, d_mean, d_kind);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}
template int launch_float_cast_classify<float>(const float *, Shape, int32_t *, float *, hipStream_t);
template int launch_float_cast_classify<double>(const double *, Shape, int32_t *, float *, hipStream_t);

// estimate_atmospheric_light on a float image: d_gray = RGB2GRAY of its quantised frame (the Canny input of every level)
template <class T>
int launch_float_airlight(const T *d_x, const uint8_t *d_gray, Shape s, int min_size, T *d_A, void *ws, hipStream_t st)
{
    Carver c(ws);
    FloatLevel<T> L = carve_float_level<T>(c, s);
    void *canny_ws = c.take<char>(canny_ws_bytes(s));
    const int B = s.B, nreg = 4 * B;
    UWIE_LAUNCH(k_f_init_blocks, dim3(cdiv(B, 64)), dim3(64), 0, st, L.blk, L.regs, L.edges, B, s.H, s.W, min_size);
    UWIE_LAUNCH_CHECK();
    const int maxChunks = max_chunks_f(s);
    int rmax = s.H, cmax = s.W;
    for (int level = 0; level < kMaxLevelsF && rmax > min_size && cmax > min_size; ++level) {
        const int qr = (rmax + 1) / 2, qc = (cmax + 1) / 2;
        const int nch = cdiv((long long)qr * qc, kNpChunk);
        UWIE_LAUNCH((k_fq_chunk_sums<T, false>), dim3(nch, nreg), dim3(64), 0, st, d_x, L.regs, s.H, s.W, maxChunks, L.csum,
                    (const T *)nullptr);
        UWIE_LAUNCH_CHECK();
        UWIE_LAUNCH((k_fq_chunk_sums<T, true>), dim3(nch, nreg), dim3(64), 0, st, d_x, L.regs, s.H, s.W, maxChunks, L.csum_var,
                    (const T *)L.csum);
        UWIE_LAUNCH_CHECK();
        const int rc = launch_canny(d_gray, s, L.regs, nreg, qr, qc, 50, 150, L.edges, nullptr, canny_ws, st, true);
        if (rc != UWIE_OK) return rc;
        UWIE_LAUNCH((k_fq_select<T>), dim3(B), dim3(64), 0, st, L.blk, L.regs, (const T *)L.csum, (const T *)L.csum_var, maxChunks,
                    L.edges, level, min_size, (TraceRecF *)nullptr);
        UWIE_LAUNCH_CHECK();
        rmax = qr;
        cmax = qc;
    }
    UWIE_LAUNCH((k_f_brightest<T>), dim3(B), dim3(64), 0, st, d_x, L.blk, s.H, s.W, d_A);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}
template int launch_float_airlight<float>(const float *, const uint8_t *, Shape, int, float *, void *, hipStream_t);
template int launch_float_airlight<double>(const double *, const uint8_t *, Shape, int, double *, void *, hipStream_t);

template <class T>
int launch_float_trans_init(const T *d_x, const T *d_A, Shape s, double omega, double norm_eps, int pre_clip, T *d_t0, hipStream_t st)
{
    UWIE_LAUNCH((k_f_trans_init<T>), dim3(grid_for((size_t)s.B * s.npx())), dim3(256), 0, st, d_x, d_A, s.npx(), s.B, (T)omega,
                (T)norm_eps, pre_clip, d_t0);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}
template int launch_float_trans_init<float>(const float *, const float *, Shape, double, double, int, float *, hipStream_t);
template int launch_float_trans_init<double>(const double *, const double *, Shape, double, double, int, double *, hipStream_t);

template <class T, class OUT>
int launch_float_restore(const T *d_x, const T *d_A, const double *d_t, Shape s, OUT *d_out, int planar, hipStream_t st)
{
    UWIE_LAUNCH((k_f_restore<T, OUT>), dim3(grid_for((size_t)s.B * s.npx())), dim3(256), 0, st, d_x, d_A, d_t, s.npx(), s.B, d_out, planar);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}
template int launch_float_restore<float, float>(const float *, const float *, const double *, Shape, float *, int, hipStream_t);
template int launch_float_restore<float, double>(const float *, const float *, const double *, Shape, double *, int, hipStream_t);
template int launch_float_restore<double, double>(const double *, const double *, const double *, Shape, double *, int, hipStream_t);

}  // namespace uwie
