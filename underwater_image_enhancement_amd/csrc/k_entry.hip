// Entry stages: cast detection (detect_image_type, six_stadigy.py:292-302), the normalise + colour-correct
// map (six_stadigy.py:406,305-323) and the 8-bit gray plane (six_stadigy.py:149,177).
//
// detect_image_type takes `img.mean(axis=(0,1))` of a float32 HxWx3 array.  NumPy evaluates that reduction
// as a SEQUENTIAL float32 accumulation per channel in raster order (measured, NumPy 2.2.6), whose result
// can be off the true mean by ~1 % at 4K -- enough to flip the 0.05 threshold.  The kernels below reproduce
// that value exactly without an 8-million-step serial chain:
//   * inside one binade [2^e, 2^(e+1)) of the accumulator s, fl(s + x) == s + RN_ulp(x) exactly, so a run
//     of pixels advances s by an INTEGER number of ulps that depends only on the histogram of the run
//     (inputs are u8/255: 256 distinct values);
//   * k_chunk_hist writes one 3x256 histogram per 16384-pixel chunk (one streaming pass over the frame);
//   * k_cast_resolve (one wavefront per image and channel) walks the chunks, advancing s in closed form
//     while the chunk stays inside the binade and has no round-half-even tie, and otherwise drills into
//     the chunk: 64 lanes x 256-pixel runs, a wave prefix scan to find the run that crosses the binade,
//     and plain sequential float adds inside that one run.
#include "common.h"
#include "devutil.h"

namespace uwie {

constexpr int kChunkPx = 16384;  // pixels per histogram chunk
constexpr int kRunPx = 256;      // pixels per lane inside a drilled chunk (64 * 256 = kChunkPx)

__global__ void __launch_bounds__(256) k_chunk_hist(const uint8_t *__restrict__ in, uint32_t *__restrict__ hist,
                                                    int npx, int nchunk)
{
    __shared__ uint32_t h[4][768];
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, w = tid >> 6;
    for (int i = tid; i < 4 * 768; i += 256) (&h[0][0])[i] = 0;
    __syncthreads();
    const uint8_t *img = in + (size_t)b * npx * 3;
    const int p0 = c * kChunkPx, p1 = min(npx, p0 + kChunkPx);
    const bool aligned = (npx & 3) == 0;
    for (int p = p0 + tid * 4; p < p1; p += 1024) {
        const int n = min(4, p1 - p);
        const Px4 v = load_px4(img + (size_t)p * 3, n, aligned);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < n) {
                atomicAdd(&h[w][v.r[i]], 1u);
                atomicAdd(&h[w][256 + v.g[i]], 1u);
                atomicAdd(&h[w][512 + v.b[i]], 1u);
            }
    }
    __syncthreads();
    uint32_t *out = hist + ((size_t)b * nchunk + c) * 768;
    for (int i = tid; i < 768; i += 256) out[i] = h[0][i] + h[1][i] + h[2][i] + h[3][i];
}

__device__ __forceinline__ float from_mantissa(uint32_t S, int e)
{
    return __uint_as_float(((uint32_t)(e + 127) << 23) | (S & 0x7fffffu));
}

// Advance the accumulator `s` over pixels [p0, p0+cnt) of one channel (stride 3 bytes); all 64 lanes call.
__device__ float drill_chunk(const uint8_t *__restrict__ chan, int p0, int cnt, float s, const CastTables *tab,
                             const float *xs)
{
    const int lane = threadIdx.x & 63;
    const int ln = max(0, min(kRunPx, cnt - lane * kRunPx));
    const uint8_t *run = chan + (size_t)(p0 + lane * kRunPx) * 3;
    int first = 0;
    while (first < kWave) {
        int L;  // the run that has to be walked one pixel at a time
        if (s >= 0.25f) {
            const int e = (int)(__float_as_uint(s) >> 23) - 127;
            const int ei = min(e - kCastBinadeMin, kCastBinades - 1);
            uint64_t D = 0;
            uint32_t T = 0;
            if (lane >= first)
                for (int i = 0; i < ln; ++i) {
                    const uint32_t u = run[(size_t)i * 3];
                    D += tab->R[ei][u];
                    T += tab->tie[ei][u];
                }
            const uint64_t incl = wave_incl_scan_u64(D);
            const uint32_t S = (__float_as_uint(s) & 0x7fffffu) | 0x800000u;
            const bool bad = lane >= first && (T != 0 || S + incl >= (1ull << 24));
            const uint64_t mask = __ballot(bad);
            L = mask ? (int)__builtin_ctzll(mask) : kWave;
            const uint64_t adv = L > first ? shfl_u64(incl, L - 1) : 0;
            s = from_mantissa(S + (uint32_t)adv, e);
        } else {
            // tiny accumulator: skip runs that are all zero (adding 0.0f changes nothing), walk the first that is not
            bool nz = false;
            if (lane >= first)
                for (int i = 0; i < ln; ++i) nz |= run[(size_t)i * 3] != 0;
            const uint64_t mask = __ballot(nz);
            L = mask ? (int)__builtin_ctzll(mask) : kWave;
        }
        if (L >= kWave) break;
        if (lane == L)
            for (int i = 0; i < ln; ++i) s = s + xs[run[(size_t)i * 3]];
        s = __shfl(s, L);
        first = L + 1;
    }
    return s;
}

__global__ void __launch_bounds__(64) k_cast_resolve(const uint8_t *__restrict__ in, const uint32_t *__restrict__ hist,
                                                     const CastTables *__restrict__ tab, int npx, int nchunk,
                                                     float *__restrict__ sums)
{
    const int ch = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    __shared__ float xs[256];
    for (int i = lane; i < 256; i += 64) xs[i] = px_norm(i);
    __syncthreads();
    const uint8_t *chan = in + (size_t)b * npx * 3 + ch;
    float s = 0.0f;
    // The only loop-carried value is s.  Everything that does not depend on it is kept off the critical path: the
    // next chunk's histogram slice is prefetched, and the rounding-table row of the current binade stays in registers
    // (it changes ~24 times per channel).
    const uint4 *hp = reinterpret_cast<const uint4 *>(hist + (((size_t)b * nchunk) * 3 + ch) * 256 + lane * 4);
    uint4 hnext = hp[0];
    int ei_cached = -1;
    uint32_t R0 = 0, R1 = 0, R2 = 0, R3 = 0;
    bool t0 = false, t1 = false, t2 = false, t3 = false;
    for (int c = 0; c < nchunk; ++c) {
        const uint4 hv = hnext;
        if (c + 1 < nchunk) hnext = hp[(size_t)(c + 1) * 192];  // 768 words per chunk = 192 uint4
        const uint32_t h0 = hv.x, h1 = hv.y, h2 = hv.z, h3 = hv.w;
        bool done;
        if (s >= 0.25f) {
            const int e = (int)(__float_as_uint(s) >> 23) - 127;
            const int ei = min(e - kCastBinadeMin, kCastBinades - 1);
            if (ei != ei_cached) {
                const uint32_t *R = &tab->R[ei][lane * 4];
                const uint8_t *Tt = &tab->tie[ei][lane * 4];
                R0 = R[0]; R1 = R[1]; R2 = R[2]; R3 = R[3];
                t0 = Tt[0] != 0; t1 = Tt[1] != 0; t2 = Tt[2] != 0; t3 = Tt[3] != 0;
                ei_cached = ei;
            }
            uint64_t D = (uint64_t)h0 * R0 + (uint64_t)h1 * R1 + (uint64_t)h2 * R2 + (uint64_t)h3 * R3;
            uint32_t T = (h0 && t0) + (h1 && t1) + (h2 && t2) + (h3 && t3);
            D = wave_sum_u64(D);
            T = wave_sum_u32(T);
            const uint32_t S = (__float_as_uint(s) & 0x7fffffu) | 0x800000u;
            done = T == 0 && S + D < (1ull << 24);
            if (done) s = from_mantissa(S + (uint32_t)D, e);
        } else {
            const uint32_t nz = wave_sum_u32((lane == 0 ? 0u : h0) + h1 + h2 + h3);
            done = nz == 0;
        }
        if (!done) s = drill_chunk(chan, c * kChunkPx, min(kChunkPx, npx - c * kChunkPx), s, tab, xs);
    }
    if (lane == 0) sums[b * 3 + ch] = s;
}

// mean = float32(sum / count) with the division done in float64 (NumPy's _mean: true_divide of a float32
// array by an intp count selects the float64 loop, numpy/_core/_methods.py), then the three-way test.
__global__ void k_cast_decide(const float *__restrict__ sums, int B, int npx, int32_t *__restrict__ kind,
                              float *__restrict__ mean_out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float r = (float)((double)sums[b * 3 + 0] / (double)npx);
    const float g = (float)((double)sums[b * 3 + 1] / (double)npx);
    const float bl = (float)((double)sums[b * 3 + 2] / (double)npx);
    int k = UWIE_CAST_NORMAL;
    if (g > r && g > bl && (g - r) > 0.05f) k = UWIE_CAST_GREENISH;
    else if (bl > r && bl > g && (bl - r) > 0.05f) k = UWIE_CAST_BLUISH;
    if (kind) kind[b] = k;
    if (mean_out) {
        mean_out[b * 3 + 0] = r;
        mean_out[b * 3 + 1] = g;
        mean_out[b * 3 + 2] = bl;
    }
}

size_t cast_ws_bytes(Shape s)
{
    Carver c(nullptr);
    const int nchunk = cdiv((long long)s.npx(), kChunkPx);
    c.take<uint32_t>((size_t)s.B * nchunk * 768);
    c.take<float>((size_t)s.B * 3);
    return c.total();
}

int launch_cast_classify(uwie_ctx *ctx, const uint8_t *d_in, Shape s, int32_t *d_kind, float *d_mean, void *ws,
                         hipStream_t st)
{
    Carver c(ws);
    const int npx = (int)s.npx();
    const int nchunk = cdiv(npx, kChunkPx);
    uint32_t *hist = c.take<uint32_t>((size_t)s.B * nchunk * 768);
    float *sums = c.take<float>((size_t)s.B * 3);
    UWIE_LAUNCH(k_chunk_hist, dim3(nchunk, s.B), dim3(256), 0, st, d_in, hist, npx, nchunk);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_cast_resolve, dim3(3, s.B), dim3(64), 0, st, d_in, hist, ctx->d_cast, npx, nchunk, sums);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_cast_decide, dim3(cdiv(s.B, 64)), dim3(64), 0, st, sums, s.B, npx, d_kind, d_mean);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

__global__ void k_set_kind(int32_t *kind, int B, int v)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) kind[b] = v;
}

int launch_set_kind(int32_t *d_kind, int B, int kind, hipStream_t st)
{
    UWIE_LAUNCH(k_set_kind, dim3(cdiv(B, 64)), dim3(64), 0, st, d_kind, B, kind);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

__global__ void __launch_bounds__(256) k_normalise_correct(const uint8_t *__restrict__ in, const int32_t *__restrict__ kind,
                                                           float *__restrict__ out, int npx)
{
    const int b = blockIdx.y;
    const int k = kind ? kind[b] : 0;
    const size_t base = (size_t)b * npx * 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)npx * 3; i += (size_t)gridDim.x * 256)
        out[base + i] = px_val(in[base + i], px_atten(k, (int)(i % 3)));
}

int launch_normalise_correct(const uint8_t *d_in, const int32_t *d_kind, float *d_out, Shape s, hipStream_t st)
{
    const int blocks = grid_for(s.npx() * 3, 4096);
    UWIE_LAUNCH(k_normalise_correct, dim3(blocks, s.B), dim3(256), 0, st, d_in, d_kind, d_out, (int)s.npx());
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

// gray = cvtColor((x*255).astype(u8), RGB2GRAY) of the (cast-corrected) frame: six_stadigy.py:149,177.
__global__ void __launch_bounds__(256) k_quant_gray(const uint8_t *__restrict__ in, const int32_t *__restrict__ kind,
                                                    uint8_t *__restrict__ gray, int npx, int shift)
{
    const int b = blockIdx.y;
    const int k = kind ? kind[b] : 0;
    const uint8_t *img = in + (size_t)b * npx * 3;
    uint8_t *g = gray + (size_t)b * npx;
    const bool aligned = (npx & 3) == 0;
    const bool ag = px_atten(k, 1), ab = px_atten(k, 2);
    for (int p = (blockIdx.x * 256 + threadIdx.x) * 4; p < npx; p += gridDim.x * 1024) {
        const int n = min(4, npx - p);
        const Px4 v = load_px4(img + (size_t)p * 3, n, aligned);
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o[i] = gray_fixed(quant_u8(px_val(v.r[i], false)), quant_u8(px_val(v.g[i], ag)), quant_u8(px_val(v.b[i], ab)),
                              shift);
        if (aligned && n == 4) {
            *reinterpret_cast<uint32_t *>(g + p) = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
        } else {
            for (int i = 0; i < n; ++i) g[p + i] = (uint8_t)o[i];
        }
    }
}

int launch_quant_gray(const uint8_t *d_in, const int32_t *d_kind, uint8_t *d_gray, Shape s, int gray_shift,
                      hipStream_t st)
{
    const int blocks = grid_for((s.npx() + 3) / 4, 4096);
    UWIE_LAUNCH(k_quant_gray, dim3(blocks, s.B), dim3(256), 0, st, d_in, d_kind, d_gray, (int)s.npx(), gray_shift);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

}  // namespace uwie
