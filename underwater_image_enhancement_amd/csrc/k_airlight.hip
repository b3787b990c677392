// estimate_atmospheric_light (six_stadigy.py:49-113 == enhancement_strategies.py:77-144): greedy quadtree
// descent.  Per level the four quadrants of the current block are scored with compute_Q
// (six_stadigy.py:116-157) and the walk steps into the first maximum; the leaf's brightest pixel is A.
//
// compute_Q's float32 sums must reproduce NumPy's summation ORDER, because the argmax over the four scores
// is discontinuous: `np.sum` over a (strided or contiguous) 2-D float32 array walks it in raster order in
// buffer chunks of 8192 elements, sums each chunk with its pairwise routine (8 interleaved accumulators on
// blocks of <=128, recursive halving above that) and adds the chunk results sequentially (measured, NumPy
// 2.2.6).  One wavefront evaluates one chunk: for a full chunk lane l owns the l-th 128-element leaf and the
// 64 leaf sums are combined by an xor-butterfly, which is exactly the recursion's tree; a ragged tail chunk
// enumerates the recursion's leaves first and combines them in post-order.
#include "common.h"
#include "devutil.h"
#include "pairwise_tree.h"

#include <cstdlib>

namespace uwie {

namespace {

constexpr int kMaxLevels = 32;

struct TraceRec {  // matches the layout documented in uwie.h
    int32_t y0, x0, rows, cols;
    double score[4];
};

struct LevelBufs {
    Region *blk;       // [B] current block
    Region *regs;      // [4B] its quadrants (inactive when the block is a leaf)
    float *csum;       // [4B][maxChunks][3] chunk sums
    float *csum_var;   // the same for the squared deviations
    float *tot;        // [4B][3] sums
    float *mean;       // [4B][3] means
    float *vtot;       // [4B][3] sums of squared deviations
    uint32_t *edges;   // [4B] Canny edge counts
    uint32_t *hist;    // [4B][3][256] byte histograms of the quadrants (k_q_hist)
    uint8_t *skip;     // [B] 1: the level was decided from the histograms, the exact kernels return at once
    double *chk;       // [B][9] tuning q_hist = 3: the level's four score intervals and the decision taken from them, for k_q_select to verify
};

__global__ void k_init_blocks(Region *blk, int B, int H, int W)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) blk[b] = Region{b, 0, 0, H, W};
}

// The quadtree's set-up in one launch (round 4: k_init_blocks, k_make_quadrants and three memsets were five launches of ~5 us):
// every image's block = its frame, the four quadrants of it, zeroed edge counters, pre-pass flags and level histograms.
__global__ void __launch_bounds__(256) k_q_setup(Region *__restrict__ blk, Region *__restrict__ regs, uint32_t *__restrict__ edges,
                                                 uint32_t *__restrict__ strong, uint32_t *__restrict__ hist, int B, int H, int W,
                                                 int min_size)
{
    const int i = blockIdx.x * 256 + threadIdx.x, n = gridDim.x * 256;
    for (int b = i; b < B; b += n) {
        const Region k{b, 0, 0, H, W};
        blk[b] = k;
        const bool leaf = k.rows <= min_size || k.cols <= min_size;  // six_stadigy.py:76
        const int mr = k.rows / 2, mc = k.cols / 2;                 // six_stadigy.py:85-86
        Region q[4] = {{b, 0, 0, mr, mc}, {b, 0, mc, mr, W - mc}, {b, mr, 0, H - mr, mc}, {b, mr, mc, H - mr, W - mc}};
        for (int j = 0; j < 4; ++j) {
            if (leaf) q[j].rows = q[j].cols = 0;
            regs[b * 4 + j] = q[j];
            edges[b * 4 + j] = 0;
            strong[b * 4 + j] = 0;
        }
    }
    uint4 *h4 = reinterpret_cast<uint4 *>(hist);  // [4 B][768] words: a multiple of four, 256-byte aligned (Carver)
    for (int j = i; j < B * 768; j += n) h4[j] = make_uint4(0, 0, 0, 0);
}

__global__ void k_make_quadrants(const Region *__restrict__ blk, Region *__restrict__ regs, int B, int min_size)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const Region k = blk[b];
    const bool leaf = k.rows <= min_size || k.cols <= min_size;  // six_stadigy.py:76
    const int mr = k.rows / 2, mc = k.cols / 2;                 // six_stadigy.py:85-86
    Region q[4] = {{b, k.y0, k.x0, mr, mc},
                   {b, k.y0, k.x0 + mc, mr, k.cols - mc},
                   {b, k.y0 + mr, k.x0, k.rows - mr, mc},
                   {b, k.y0 + mr, k.x0 + mc, k.rows - mr, k.cols - mc}};
    for (int i = 0; i < 4; ++i) {
        if (leaf) q[i].rows = q[i].cols = 0;
        regs[b * 4 + i] = q[i];
    }
}

// value of element `e` (raster index inside region r) for channel c: x or (x - mean)^2
template <bool VAR>
struct Elem {
    const uint8_t *img;
    int W;
    float att[3];  // color_correction's factor of the channel (six_stadigy.py:305-323): 0.85f or 1.0f (x * 1.0f == x)
    float mean[3];
    __device__ __forceinline__ void set_kind(int kind)
    {
#pragma unroll
        for (int c = 0; c < 3; ++c) att[c] = px_atten(kind, c) ? 0.85f : 1.0f;
    }
    __device__ __forceinline__ float get(const uint8_t *p, int c) const { return get(p, c, c); }
    // value of byte p[i] taken as channel c
    __device__ __forceinline__ float get(const uint8_t *p, int i, int c) const
    {
        float v = px_norm_fast(p[i]) * att[c];  // (a multiply instead of a select per element)
        if (VAR) {
            const float d = v - mean[c];
            v = d * d;
        }
        return v;
    }
};

// Level 0 only (GRAY): the sums pass visits every pixel of the frame exactly once with its normalised, colour-corrected
// value in hand, so it also writes the 8-bit gray plane -- gray = cvtColor((x*255).astype(u8), RGB2GRAY), six_stadigy.py:149,177
// -- instead of a separate sweep over the frame (k_quant_gray: 1.6 GB read again at 4K x 64).
struct GrayOut {
    uint8_t *plane;  // gray plane of the image (nullptr: none)
    int shift;       // RGB2GRAY fixed point (15 or 14)
};
__device__ __forceinline__ uint32_t gray_of(float v0, float v1, float v2, int shift)
{
    return gray_fixed(quant_u8(v0), quant_u8(v1), quant_u8(v2), shift);
}

// NumPy's pairwise_sum on a block of n <= 128 elements starting at raster element e0, three channels at once.
template <bool VAR, bool GRAY = false>
__device__ void leaf_sum3(const Elem<VAR> &el, const Region &r, int e0, int n, float out[3], GrayOut go = GrayOut{nullptr, 15})
{
    static_assert(!(VAR && GRAY), "the gray plane rides with the plain sums");
    int ly = e0 / r.cols, lx = e0 % r.cols;
    const uint8_t *p = el.img + ((size_t)(r.y0 + ly) * el.W + r.x0 + lx) * 3;
    uint8_t *gp = GRAY ? go.plane + (size_t)(r.y0 + ly) * el.W + r.x0 + lx : nullptr;
    auto step = [&]() {
        ++lx;
        p += 3;
        if (GRAY) ++gp;
        if (lx == r.cols) {
            lx = 0;
            p += (size_t)(el.W - r.cols) * 3;
            if (GRAY) gp += el.W - r.cols;
        }
    };
    // the three values of the pixel at p (and its gray byte on the way)
    auto px3 = [&](float &v0, float &v1, float &v2) {
        v0 = el.get(p, 0);
        v1 = el.get(p, 1);
        v2 = el.get(p, 2);
        if constexpr (GRAY) *gp = (uint8_t)gray_of(v0, v1, v2, go.shift);
    };
    if (n < 8) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int i = 0; i < n; ++i) {
            float v0, v1, v2;
            px3(v0, v1, v2);
            a0 += v0;
            a1 += v1;
            a2 += v2;
            step();
        }
        out[0] = a0; out[1] = a1; out[2] = a2;
        return;
    }
    float acc[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        px3(acc[0][j], acc[1][j], acc[2][j]);
        step();
    }
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v0, v1, v2;
            px3(v0, v1, v2);
            acc[0][j] += v0;
            acc[1][j] += v1;
            acc[2][j] += v2;
            step();
        }
    }
    float res[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) res[c] = tree8(acc[c]);
    for (; i < n; ++i) {
        float v0, v1, v2;
        px3(v0, v1, v2);
        res[0] += v0;
        res[1] += v1;
        res[2] += v2;
        step();
    }
    out[0] = res[0]; out[1] = res[1]; out[2] = res[2];
}

// The same for a full 128-element leaf, eight pixels per step: when the eight pixels lie in one row of the region
// their 24 bytes come in as six (unaligned) dword loads instead of 24 byte loads.
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint4 __attribute__((aligned(1))) u128_unaligned;

typedef uint2 __attribute__((aligned(1))) u64x_unaligned;

template <bool VAR, bool GRAY = false>
__device__ void leaf_sum3_full(const Elem<VAR> &el, const Region &r, int e0, float out[3], GrayOut go = GrayOut{nullptr, 15})
{
    static_assert(!(VAR && GRAY), "the gray plane rides with the plain sums");
    int ly = e0 / r.cols, lx = e0 % r.cols;
    const uint8_t *p = el.img + ((size_t)(r.y0 + ly) * el.W + r.x0 + lx) * 3;
    uint8_t *gp = GRAY ? go.plane + (size_t)(r.y0 + ly) * el.W + r.x0 + lx : nullptr;
    float acc[3][8];
    if (lx + 128 <= r.cols) {
        // The leaf lies in one row: its 384 bytes come in as 24 (unaligned) 16-byte loads issued back to back; walking it
        // 24 bytes at a time left one load round trip per step on the critical path (the kernel sat in s_waitcnt).
        uint32_t raw[96];
        const u128_unaligned *w = reinterpret_cast<const u128_unaligned *>(p);
#pragma unroll
        for (int q = 0; q < 24; ++q) {
            const uint4 v = w[q];
            raw[4 * q] = v.x; raw[4 * q + 1] = v.y; raw[4 * q + 2] = v.z; raw[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            uint32_t g8[2] = {0, 0};  // gray bytes of the step's eight pixels
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v3[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int byte = 24 * it + 3 * j + c;
                    const uint8_t u = (uint8_t)(raw[byte >> 2] >> (8 * (byte & 3)));
                    const float v = el.get(&u, 0, c);
                    v3[c] = v;
                    acc[c][j] = it == 0 ? v : acc[c][j] + v;
                }
                if constexpr (GRAY) g8[j >> 2] |= gray_of(v3[0], v3[1], v3[2], go.shift) << (8 * (j & 3));
            }
            if constexpr (GRAY) *reinterpret_cast<u64x_unaligned *>(gp + 8 * it) = make_uint2(g8[0], g8[1]);
        }
        out[0] = tree8(acc[0]);
        out[1] = tree8(acc[1]);
        out[2] = tree8(acc[2]);
        return;
    }
    for (int it = 0; it < 16; ++it) {
        uint8_t px[24];
        uint8_t *gq[8] = {};  // (GRAY) where the step's eight gray bytes go
        if (lx + 8 <= r.cols) {
            const u32_unaligned *w = reinterpret_cast<const u32_unaligned *>(p);
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const uint32_t v = w[q];
                px[4 * q] = (uint8_t)v;
                px[4 * q + 1] = (uint8_t)(v >> 8);
                px[4 * q + 2] = (uint8_t)(v >> 16);
                px[4 * q + 3] = (uint8_t)(v >> 24);
            }
            if (GRAY) gq[0] = gp;
            lx += 8;
            p += 24;
            if (GRAY) gp += 8;
            if (lx == r.cols) {
                lx = 0;
                p += (size_t)(el.W - r.cols) * 3;
                if (GRAY) gp += el.W - r.cols;
            }
            if (GRAY) {
#pragma unroll
                for (int j = 1; j < 8; ++j) gq[j] = gq[0] + j;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                px[3 * j] = p[0];
                px[3 * j + 1] = p[1];
                px[3 * j + 2] = p[2];
                if (GRAY) gq[j] = gp;
                ++lx;
                p += 3;
                if (GRAY) ++gp;
                if (lx == r.cols) {
                    lx = 0;
                    p += (size_t)(el.W - r.cols) * 3;
                    if (GRAY) gp += el.W - r.cols;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v3[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = el.get(px + 3 * j, c);
                v3[c] = v;
                acc[c][j] = it == 0 ? v : acc[c][j] + v;
            }
            if constexpr (GRAY) *gq[j] = (uint8_t)gray_of(v3[0], v3[1], v3[2], go.shift);
        }
    }
    out[0] = tree8(acc[0]);
    out[1] = tree8(acc[1]);
    out[2] = tree8(acc[2]);
}

// Sequential total of a region's chunk sums (NumPy adds each buffer's pairwise result into the running total): lanes 0..2
// (one per channel) add src[k*3 + lane] for k = 0 .. nch-1 in order.  The sums come through a fixed-size LDS piece with
// coalesced loads, so no launch has to size its LDS by the frame (a 24 MP still has 733 chunks per quadrant, 2^30 pixels
// 32768).  One wavefront per workgroup.
constexpr int kSumPiece = 256;
__device__ __forceinline__ float seq_chunk_total(const float *__restrict__ src, int nch, int lane, float *piece)
{
    float acc = 0.0f;
    for (int base = 0; base < nch; base += kSumPiece) {
        const int m = min(kSumPiece, nch - base);
        for (int i = lane; i < m * 3; i += 64) piece[i] = src[(size_t)base * 3 + i];
        __syncthreads();
        if (lane < 3)
            for (int k = 0; k < m; ++k) acc = acc + piece[k * 3 + lane];
        __syncthreads();
    }
    return acc;
}

// One wavefront per (chunk, region).  csum[(reg*maxChunks + chunk)*3 + c] = pairwise sum of that chunk.
// VAR with csum_in != nullptr: the means come from the chunk sums of the previous pass (added sequentially by lanes 0..2,
// as k_q_combine would), so the quadtree needs no combine launch between its two passes.
template <bool VAR, bool GRAY = false>
__global__ void __launch_bounds__(64) k_q_chunk_sums(const uint8_t *__restrict__ in, const int32_t *__restrict__ kind,
                                                     const Region *__restrict__ regs, const float *__restrict__ mean,
                                                     int H, int W, int maxChunks, float *__restrict__ csum,
                                                     const float *__restrict__ csum_in = nullptr,
                                                     uint8_t *__restrict__ gray_out = nullptr, int gray_shift = 15,
                                                     const uint8_t *__restrict__ skip = nullptr)
{
    const int reg = blockIdx.y, ci = blockIdx.x, lane = threadIdx.x;
    const Region r = regs[reg];
    if (skip && skip[r.img]) return;  // decided from the histograms (k_q_decide)
    const int n = r.rows * r.cols;
    const int c0 = ci * kNpChunk;
    if (c0 >= n) return;
    const int len = min(kNpChunk, n - c0);

    __shared__ PairwiseTree tree;

    Elem<VAR> el;
    el.img = in + (size_t)r.img * H * W * 3;
    el.W = W;
    el.set_kind(kind ? kind[r.img] : 0);
    if (VAR) {
        if (csum_in) {
            // the chunk sums come into LDS with coalesced loads; lanes 0..2 then add them in order
            __shared__ float cs_l[kSumPiece * 3];
            const int nch = (n + kNpChunk - 1) / kNpChunk;
            float acc = seq_chunk_total(csum_in + (size_t)reg * maxChunks * 3, nch, lane, cs_l);
            if (lane < 3) acc = (float)((double)acc / (double)n);  // numpy/_core/_methods.py:_mean
            el.mean[0] = __shfl(acc, 0);
            el.mean[1] = __shfl(acc, 1);
            el.mean[2] = __shfl(acc, 2);
        } else {
            el.mean[0] = mean[reg * 3 + 0];
            el.mean[1] = mean[reg * 3 + 1];
            el.mean[2] = mean[reg * 3 + 2];
        }
    }
    float *out = csum + ((size_t)reg * maxChunks + ci) * 3;
    const GrayOut go{GRAY ? gray_out + (size_t)r.img * H * W : nullptr, gray_shift};

    if (len == kNpChunk) {
        // balanced tree: 64 leaves of 128, lane = leaf, butterfly == recursive halving
        float s[3];
        leaf_sum3_full<VAR, GRAY>(el, r, c0 + lane * 128, s, go);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            s[0] += __shfl_xor(s[0], o);
            s[1] += __shfl_xor(s[1], o);
            s[2] += __shfl_xor(s[2], o);
        }
        if (lane == 0) { out[0] = s[0]; out[1] = s[1]; out[2] = s[2]; }
        return;
    }
    // ragged chunk: the recursion tree of NumPy's pairwise sum for this length (pairwise_tree.h)
    float res[3];
    pairwise_ragged(len, lane, tree, [&](int off, int l, float *s3) { leaf_sum3<VAR, GRAY>(el, r, c0 + off, l, s3, go); }, res);
    if (lane == 0) { out[0] = res[0]; out[1] = res[1]; out[2] = res[2]; }
}

// Sequential accumulation of the chunk sums (NumPy adds each buffer's pairwise result into the running total),
// and for the first pass the mean: float32(float64(sum) / n) as numpy/_core/_methods.py:_mean does for a scalar result.
template <bool VAR>
__global__ void __launch_bounds__(64) k_q_combine(const Region *__restrict__ regs, const float *__restrict__ csum, int nreg,
                                                  int maxChunks, float *__restrict__ tot, float *__restrict__ mean)
{
    // one wavefront per region: the chunk sums come into LDS with coalesced loads, lanes 0..2 then add them in order
    __shared__ float cs[kSumPiece * 3];
    const int reg = blockIdx.x, lane = threadIdx.x;
    const Region r = regs[reg];
    const int n = r.rows * r.cols;
    if (n == 0) return;
    const int nch = (n + kNpChunk - 1) / kNpChunk;
    const float acc = seq_chunk_total(csum + (size_t)reg * maxChunks * 3, nch, lane, cs);
    if (lane >= 3) return;
    tot[reg * 3 + lane] = acc;
    if (!VAR) mean[reg * 3 + lane] = (float)((double)acc / (double)n);
}

// The greedy step (six_stadigy.py:100-111): quadrant `arg` becomes the block, its quadrants the next level's regions
// (what k_make_quadrants would write) with zeroed edge counters.  One lane.
__device__ void q_descend(Region *__restrict__ blk, Region *__restrict__ regs, uint32_t *__restrict__ edges, int b, int arg, int min_size)
{
    const Region k = regs[b * 4 + arg];
    blk[b] = k;
    const bool leaf = k.rows <= min_size || k.cols <= min_size;  // six_stadigy.py:76
    const int mr = k.rows / 2, mc = k.cols / 2;                 // six_stadigy.py:85-86
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

// ---- a level decided from byte histograms (round 3) ------------------------------------------------------------------
// compute_Q's score of a quadrant is a function of sums over its pixels, and a pixel's contribution only depends on its
// byte: the EXACT sums (sum x, sum (x - m)^2 per channel) follow from a 3 x 256 histogram of the quadrant -- one pass over
// the RGB bytes where NumPy's order needs two, and a streaming one (k_q_hist: the lane-column histogram words of
// k_chunk_hist; on level 0 it writes the gray plane on the way).  The reference's float32 pairwise sums differ from the
// exact ones by a bounded amount, so k_q_decide evaluates each quadrant's score as an INTERVAL that contains the
// reference's value; when the best quadrant's interval lies above the other three the argmax is known and the level is
// decided (results identical by construction), otherwise -- near ties, or when the caller wants the scores themselves
// (trace) -- the image is left to the exact kernels (k_q_chunk_sums x 2, k_q_select), which return at once for every
// decided image.  Bounds: a chunk's pairwise sum of non-negative terms is within 35 u of exact (15 sequential adds per
// accumulator, 3 + 7 tree levels up to the 8192-element buffer, the unrolled tail), the sequential total over nch chunks
// adds (nch - 1) u, u = 2^-24; the squared deviations add 3 u per element and n (m~ - m)^2 from the rounded mean; every
// float32 operation of the score adds u.  All of them are taken 1.25 x.
constexpr int kQHistCols = 32, kQHistFold = 31;  // fold the 10-bit fields after 31 steps: 31 * 4 pixels * 8 threads per column
static_assert(kQHistFold * 4 * (256 / kQHistCols) < 1024, "10-bit fields");

// NARROW (chosen by the host from the level's largest quadrant): rows of fewer than 256 groups share a step
template <bool GRAY, bool NARROW>
__global__ void __launch_bounds__(256) k_q_hist(const uint8_t *__restrict__ in, const int32_t *__restrict__ kind,
                                                const Region *__restrict__ regs, int H, int W, uint32_t *__restrict__ hist,
                                                uint8_t *__restrict__ gray_out, int gray_shift)
{
    __shared__ __attribute__((aligned(16))) uint32_t h[256 * kQHistCols];
    __shared__ uint32_t cnt[768];
    const int reg = blockIdx.y, tid = threadIdx.x;
    const Region r = regs[reg];
    if (r.rows <= 0 || r.cols <= 0) return;
    const int rpb = (r.rows + (int)gridDim.x - 1) / (int)gridDim.x;
    const int y_lo = blockIdx.x * rpb, y_hi = min(r.rows, y_lo + rpb);
    if (y_lo >= y_hi) return;
    for (int i = tid; i < 256 * kQHistCols / 4; i += 256) reinterpret_cast<uint4 *>(h)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < 768; i += 256) cnt[i] = 0;
    __syncthreads();
    const int knd = kind ? kind[r.img] : 0;
    const uint8_t *img = in + (size_t)r.img * H * W * 3;
    uint8_t *gray = GRAY ? gray_out + (size_t)r.img * H * W : nullptr;
    const uint32_t colb = (uint32_t)(tid & (kQHistCols - 1)) * 4u;
    char *hb = reinterpret_cast<char *>(h);
    auto bump = [&](uint32_t moved, uint32_t inc) { atomicAdd(reinterpret_cast<uint32_t *>(hb + ((moved & 0x7f80u) | colb)), inc); };
    constexpr uint32_t kR = 1u, kG = 1u << 10, kB = 1u << 20;
    auto fold = [&]() {  // all threads; thread v folds the 32 columns of value v into the 32-bit totals and clears them
        __syncthreads();
        uint32_t cr = 0, cg = 0, cb = 0;
#pragma unroll 8
        for (int k = 0; k < kQHistCols; ++k) {
            uint32_t &w = h[tid * kQHistCols + ((tid + k) & (kQHistCols - 1))];
            cr += w & 1023u;
            cg += (w >> 10) & 1023u;
            cb += w >> 20;
            w = 0;
        }
        cnt[tid] += cr; cnt[256 + tid] += cg; cnt[512 + tid] += cb;
        __syncthreads();
    };
    // (x * 255).astype(u8) of the normalised, colour-corrected value without floating point: for every byte u
    //     quant_u8(fl(u / 255) * 255) == u     and     quant_u8(fl(fl(u / 255) * 0.85f) * 255) == 17 u / 20
    // (tests/test_cabi.py checks both against NumPy's float32 arithmetic for all 256 bytes)
    const bool t0 = px_atten(knd, 0), t1 = px_atten(knd, 1), t2 = px_atten(knd, 2);
    auto gray4 = [&](uint32_t rr, uint32_t gg, uint32_t bb) {
        return gray_fixed(t0 ? rr * 17u / 20u : rr, t1 ? gg * 17u / 20u : gg, t2 ? bb * 17u / 20u : bb, gray_shift);
    };
    // groups of four pixels per row: a wide row takes `trips` steps of the block, a narrow one shares a step with RS - 1 more
    // rows (a 270-column quadrant would otherwise use 68 of 256 threads)
    const int G = (r.cols + 3) >> 2, trips = (G + 255) >> 8, RS = NARROW ? 256 / G : 1;
    const int ry = NARROW ? tid / G : 0, gx0 = NARROW ? tid - ry * G : tid;
    const bool lane_on = ry < RS;
    // A step = one group of four pixels per thread.  The 12 bytes of the NEXT step are loaded (unconditionally, position
    // clamped into the row: see k_chunk_hist) into a second register set before this step's atomics.
    struct Cursor {
        int y, t;
    };
    const bool wide = r.cols >= 4;  // (uniform)
    auto advance = [&](Cursor &c) {
        if (++c.t == trips) { c.t = 0; c.y += RS; }
    };
    auto fetch = [&](const Cursor &c, uint32_t (&d)[3]) {
        if (!wide) return;
        const int x = min(4 * (c.t * 256 + gx0), r.cols - 4), y = min(NARROW ? c.y + ry : c.y, y_hi - 1);
        const u32_unaligned *q = reinterpret_cast<const u32_unaligned *>(img + ((size_t)(r.y0 + y) * W + r.x0 + x) * 3);
        d[0] = q[0]; d[1] = q[1]; d[2] = q[2];
    };
    int steps = 0;
    auto body = [&](const Cursor &c, const uint32_t (&d)[3]) {
        if (c.y >= y_hi) return;  // (uniform)
        const int x = 4 * (c.t * 256 + gx0), yy = NARROW ? c.y + ry : c.y, n = (!NARROW || (lane_on && yy < y_hi)) ? min(4, r.cols - x) : 0;
        if (n == 4) {
            const uint32_t c0 = d[0], c1 = d[1], c2 = d[2];
            // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
            bump(c0 << 7, kR); bump(c0 >> 1, kG); bump(c0 >> 9, kB);
            bump(c0 >> 17, kR); bump(c1 << 7, kG); bump(c1 >> 1, kB);
            bump(c1 >> 9, kR); bump(c1 >> 17, kG); bump(c2 << 7, kB);
            bump(c2 >> 1, kR); bump(c2 >> 9, kG); bump(c2 >> 17, kB);
            if constexpr (GRAY) {
                const uint32_t g4 = gray4(c0 & 255, (c0 >> 8) & 255, (c0 >> 16) & 255) |
                                    gray4(c0 >> 24, c1 & 255, (c1 >> 8) & 255) << 8 |
                                    gray4((c1 >> 16) & 255, c1 >> 24, c2 & 255) << 16 |
                                    gray4((c2 >> 8) & 255, (c2 >> 16) & 255, c2 >> 24) << 24;
                *reinterpret_cast<u32_unaligned *>(gray + (size_t)(r.y0 + yy) * W + r.x0 + x) = g4;
            }
        } else {
            const uint8_t *row = img + ((size_t)(r.y0 + yy) * W + r.x0) * 3;
            for (int j = 0; j < n; ++j) {
                const uint8_t *p = row + (size_t)(x + j) * 3;
                bump((uint32_t)p[0] << 7, kR); bump((uint32_t)p[1] << 7, kG); bump((uint32_t)p[2] << 7, kB);
                if constexpr (GRAY) gray[(size_t)(r.y0 + yy) * W + r.x0 + x + j] = (uint8_t)gray4(p[0], p[1], p[2]);
            }
        }
        if (++steps == kQHistFold) {  // (block-uniform)
            fold();
            steps = 0;
        }
    };
    {
        // kAhead steps per batch, two batches of registers taking turns (a register copy would wait for the loads it copies):
        // 2 x 48 bytes per thread in flight
        constexpr int kAhead = 4;
        uint32_t bufa[kAhead][3] = {}, bufb[kAhead][3] = {};
        auto fetchN = [&](Cursor c, uint32_t (&d)[kAhead][3]) {
#pragma unroll
            for (int i = 0; i < kAhead; ++i) {
                fetch(c, d[i]);
                advance(c);
            }
        };
        auto bodyN = [&](Cursor c, const uint32_t (&d)[kAhead][3]) {
#pragma unroll
            for (int i = 0; i < kAhead; ++i) {
                body(c, d[i]);
                advance(c);
            }
        };
        auto skipN = [&](Cursor &c, int n) {
            for (int i = 0; i < n; ++i) advance(c);
        };
        Cursor ca{y_lo, 0}, cb{y_lo, 0};
        skipN(cb, kAhead);
        fetchN(ca, bufa);
        while (ca.y < y_hi) {
            fetchN(cb, bufb);
            bodyN(ca, bufa);
            skipN(ca, 2 * kAhead);
            fetchN(ca, bufa);
            bodyN(cb, bufb);
            skipN(cb, 2 * kAhead);
        }
    }
    fold();
    for (int i = tid; i < 768; i += 256)
        if (cnt[i]) atomicAdd(&hist[(size_t)reg * 768 + i], cnt[i]);
}

// One workgroup of four wavefronts per image, one per quadrant: the quadrant's score interval from its histogram (the three
// channels' sums reduced side by side); thread 0 then decides the level when it can.
// force_exact: leave every image to the exact kernels (the caller records the scores, or tuning q_hist = 2).
__global__ void __launch_bounds__(256) k_q_decide(Region *__restrict__ blk, Region *__restrict__ regs,
                                                  const uint32_t *hist, uint32_t *__restrict__ edges,
                                                  const int32_t *__restrict__ kind, int min_size, int force_exact,
                                                  uint8_t *__restrict__ skip, uint32_t *__restrict__ strong,
                                                  double *__restrict__ chk = nullptr)
{
    __shared__ double s_lo[4], s_hi[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    if (lane == 0) strong[b * 4 + q] = 0;  // the Canny pre-pass flags of this level are consumed: clean for the next one
    if (regs[b * 4].rows == 0) {  // leaf reached earlier: nothing left to decide
        if (threadIdx.x == 0) skip[b] = 1;
        return;
    }
    // every exit below leaves the image's four histograms zeroed for the next level (one memset per call, not per level)
    uint32_t *hq = const_cast<uint32_t *>(hist) + (size_t)(b * 4 + q) * 768;
    auto clear = [&]() {
        for (int i = lane; i < 768; i += 64) hq[i] = 0;
    };
    if (force_exact) {
        if (threadIdx.x == 0) skip[b] = 0;
        clear();
        return;
    }
    const int knd = kind ? kind[b] : 0;
    constexpr double u = 0x1p-24, kSafe = 1.25;
    {
        const Region r = regs[b * 4 + q];
        const double n = (double)r.rows * (double)r.cols;
        const int nch = (int)(((long long)r.rows * r.cols + kNpChunk - 1) / kNpChunk);
        const double eS = kSafe * (nch + 35) * u, eV = kSafe * (nch + 38) * u;
        double xs[3][4], ns[3][4], S[3], V[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t *hc = hist + ((size_t)(b * 4 + q) * 3 + c) * 256;
            const float a = px_atten(knd, c) ? 0.85f : 1.0f;
            S[c] = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int v = lane + 64 * k;
                xs[c][k] = (double)(px_norm_fast((uint32_t)v) * a);  // the element's float32 value, exactly as the sums see it
                ns[c][k] = (double)hc[v];
                S[c] += ns[c][k] * xs[c][k];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int c = 0; c < 3; ++c) S[c] += __shfl_xor(S[c], o);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double m = S[c] / n;
            V[c] = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) V[c] += ns[c][k] * (xs[c][k] - m) * (xs[c][k] - m);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int c = 0; c < 3; ++c) V[c] += __shfl_xor(V[c], o);
        }
        double Slo[3], Shi[3], Vlo[3], Vhi[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double dm = (eS + 2 * u) * fabs(S[c] / n);  // |m~ - m|: the rounded sum, its division, its conversion to float32
            Slo[c] = S[c] * (1.0 - eS); Shi[c] = S[c] * (1.0 + eS);
            Vlo[c] = V[c] * (1.0 - eV); Vhi[c] = (V[c] + n * dm * dm) * (1.0 + eV);
        }
        // six_stadigy.py:134-155 in float32: every operation rounds once (u), conversions of n included
        const double sA = fabs(Shi[0]) + fabs(Shi[1]) + fabs(Shi[2]);
        const double t1lo = (Slo[0] + Slo[1] + Slo[2]) / (3.0 * n), t1hi = (Shi[0] + Shi[1] + Shi[2]) / (3.0 * n);
        const double t2lo = (Slo[2] + Slo[1] - 2.0 * Shi[0]) / n, t2hi = (Shi[2] + Shi[1] - 2.0 * Slo[0]) / n;
        const double t3lo = (Vlo[0] + Vlo[1] + Vlo[2]) / (3.0 * n), t3hi = (Vhi[0] + Vhi[1] + Vhi[2]) / (3.0 * n);
        const double t4 = (double)edges[b * 4 + q] / n;
        // rounding of the float32 expression tree: <= 6 operations on terms bounded by these magnitudes
        const double slack = kSafe * 8.0 * u * (sA / (3.0 * n) + (fabs(Shi[2]) + fabs(Shi[1]) + 2.0 * fabs(Shi[0])) / n + fabs(t3hi)) + 1e-300;
        if (lane == 0) {
            s_lo[q] = (t1lo + t2lo) - t3hi - t4 - slack;
            s_hi[q] = (t1hi + t2hi) - t3lo - t4 + slack;
        }
        clear();  // (this wavefront's own reads of hq are done: the values sit in ns[][])
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    int arg = 0;
    for (int i = 1; i < 4; ++i)
        if (s_lo[i] + s_hi[i] > s_lo[arg] + s_hi[arg]) arg = i;
    bool sure = true;
    for (int i = 0; i < 4; ++i)
        if (i != arg && !(s_lo[arg] > s_hi[i])) sure = false;
    if (chk) {  // tuning q_hist = 3: nothing is decided here -- the exact kernels run and k_q_select checks these against their scores
        for (int i = 0; i < 4; ++i) {
            chk[b * 9 + i] = s_lo[i];
            chk[b * 9 + 4 + i] = s_hi[i];
        }
        chk[b * 9 + 8] = sure ? (double)arg : -1.0;
        skip[b] = 0;
        return;
    }
    skip[b] = sure ? 1 : 0;
    if (sure) q_descend(blk, regs, edges, b, arg, min_size);
}

// compute_Q's final arithmetic (six_stadigy.py:134-155) and the greedy step (six_stadigy.py:100-111).
// It also prepares the next level: the chosen block's quadrants (what k_make_quadrants would write) and zeroed edge
// counters, so a level costs two launches fewer.
// One wavefront per image.  The totals are the sequential sums of the chunk sums (NumPy adds each buffer's pairwise
// result into the running total): lane q*6 + pass*3 + c does that for quadrant q, pass (0 sums, 1 squared deviations)
// and channel c, so no combine launch is needed either.
__global__ void __launch_bounds__(64) k_q_select(Region *__restrict__ blk, Region *__restrict__ regs,
                                                 const float *__restrict__ csum, const float *__restrict__ csum_var, int maxChunks,
                                                 uint32_t *__restrict__ edges, int B, int level, int min_size,
                                                 TraceRec *__restrict__ trace, const uint8_t *__restrict__ skip = nullptr,
                                                 const double *__restrict__ chk = nullptr, uint32_t *__restrict__ status = nullptr)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (regs[b * 4].rows == 0) return;  // leaf reached earlier
    if (skip && skip[b]) return;        // decided from the histograms (k_q_decide)
    // [4 quadrants][2 passes][kSumPiece * 3]: the chunk sums come through fixed-size pieces (coalesced loads), so the launch's
    // LDS does not grow with the frame
    __shared__ float cs_l[8][kSumPiece * 3];
    int nchq[4], nchMax = 0;
    for (int q = 0; q < 4; ++q) {
        const Region r = regs[b * 4 + q];
        nchq[q] = (r.rows * r.cols + kNpChunk - 1) / kNpChunk;
        nchMax = max(nchMax, nchq[q]);
    }
    float acc = 0.0f;
    const int lq = lane < 24 ? lane / 6 : 0, lpass = (lane % 6) / 3, lc = lane % 3;
    for (int base = 0; base < nchMax; base += kSumPiece) {
        for (int q = 0; q < 4; ++q) {
            const int m = min(kSumPiece, nchq[q] - base);
            const size_t off = ((size_t)(b * 4 + q) * maxChunks + base) * 3;
            for (int i = lane; i < m * 3; i += 64) {
                cs_l[q * 2 + 0][i] = csum[off + i];
                cs_l[q * 2 + 1][i] = csum_var[off + i];
            }
        }
        __syncthreads();
        if (lane < 24) {
            const int m = min(kSumPiece, nchq[lq] - base);
            const float *cs = cs_l[lq * 2 + lpass] + lc;
            for (int k = 0; k < m; ++k) acc = acc + cs[k * 3];
        }
        __syncthreads();
    }
    float tots[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) tots[i] = __shfl(acc, i);
    if (lane != 0) return;
    double best = 0.0;
    int arg = 0;
    double score[4];
    for (int q = 0; q < 4; ++q) {
        const Region r = regs[b * 4 + q];
        const long long n = (long long)r.rows * r.cols;
        const float *S = tots + q * 6, *V = tots + q * 6 + 3;
        const float t1 = ((S[0] + S[1]) + S[2]) / (float)(3 * n);
        const float t2 = ((S[2] + S[1]) - 2.0f * S[0]) / (float)n;
        const float v0 = V[0] / (float)n, v1 = V[1] / (float)n, v2 = V[2] / (float)n;
        const float t3 = ((v0 + v1) + v2) / 3.0f;
        const double t4 = (double)edges[b * 4 + q] / (double)n;  // int64 / int -> float64
        const double Q = (double)((t1 + t2) - t3) - t4;
        score[q] = Q;
        if (q == 0 || Q > best) { best = Q; arg = q; }  // np.argmax: first maximum
    }
    if (trace) {
        TraceRec &t = trace[b * kMaxLevels + level];
        const Region k = blk[b];
        t.y0 = k.y0; t.x0 = k.x0; t.rows = k.rows; t.cols = k.cols;
        for (int q = 0; q < 4; ++q) t.score[q] = score[q];
    }
    if (chk && status) {
        // tuning q_hist = 3: the reference-order score of every quadrant has to lie inside the interval k_q_decide
        // derived from the byte histograms, and a decision it would have taken has to be this argmax
        bool bad = false;
        for (int q = 0; q < 4; ++q) bad = bad || !(score[q] >= chk[b * 9 + q] && score[q] <= chk[b * 9 + 4 + q]);
        if (chk[b * 9 + 8] >= 0.0 && (int)chk[b * 9 + 8] != arg) bad = true;
        if (bad) atomicOr(status, (uint32_t)UWIE_STATUS_QTREE_BOUNDS);
    }
    q_descend(blk, regs, edges, b, arg, min_size);
}

// ---- the small levels in one launch ---------------------------------------------------------------------------------
// Once a quadrant holds at most one NumPy buffer (8192 elements) a level is a few microseconds of work behind nine
// launches, and with min_size 1 there are seven such levels at 1080p and at 4K.  k_q_tail walks them all: one workgroup
// of 16 wavefronts per image, four wavefronts per quadrant, everything a level touches staged in LDS.
// Per quadrant (its own LDS slice): the RGB bytes come in with coalesced loads; the pairwise recursion's leaves are
// summed by eight lanes each -- lane j owns NumPy's accumulator r[j], the three-level combination of the eight is an
// xor-butterfly, the tail elements follow sequentially -- and folded back by one wavefront (pairwise_tree.h); the same
// again for the squared deviations; then the slice is reused for the gray bytes and one 16-bit word per pixel
//   [10:0] |dx|+|dy|   [12:11] direction (0 horizontal, 1 vertical, 2 diagonal dx*dy>=0, 3 diagonal dx*dy<0)
//   [14:13] state after suppression (0 none, 1 weak, 2 strong -> edge)
// for cv2.Canny on the quadrant (Sobel with the quadrant's border replicated, magnitudes outside it 0, OpenCV's
// fixed-point tan(22.5 deg)).  Hysteresis, only when the quadrant has a strong pixel at all: one wavefront repeats
// forward / backward sweeps in raster order, 64 consecutive pixels per step; inside a step the horizontal runs are
// closed with an occluded fill on the 64-bit ballots, the rows above and below are read back from LDS.  The sweeps stop
// when one changes nothing, which is the fixed point cv2.Canny's stack reaches.
// The four scores meet in LDS, every thread takes the first maximum and steps into that quadrant; the leaf's brightest
// pixel is taken at the end.
typedef PairwiseTreeT<float, 128, 8> TailTree;
constexpr int kTailSub = 4;                                    // wavefronts per quadrant
constexpr int kTailPixBytes = 3 * kNpChunk;                    // RGB of a quadrant; later its gray bytes + 16-bit words
constexpr int kTailTreeBytes = (int)((sizeof(TailTree) + 15) & ~size_t(15));
constexpr int kTailQuadBytes = kTailPixBytes + kTailTreeBytes;

__device__ __forceinline__ void tail_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// rows x rowbytes bytes (global, row pitch `pitch`) -> dense LDS, by the `nl` lanes of a quadrant (this one is `ql`):
// dword loads; (row, dword) advance without a division per element.
// Round 3: (a) the loads are unconditional -- a row's last, partial dword is fetched as the row's last four bytes and
// shifted, positions past the end re-read the last row: a load under a branch gets its own s_waitcnt vmcnt(0), which made
// every load a round trip of its own; (b) a dword that lands 4-byte aligned in the dense LDS image is stored as one
// instead of four bytes.  Rows shorter than four bytes
// (one RGB pixel, up to three gray ones) keep the byte loads.
struct StageCursor {
    int y, d;
};
template <int N>
__device__ __forceinline__ void tail_stage_n(const uint8_t *__restrict__ src, size_t pitch, int rows, int rowbytes, uint8_t *dst,
                                             int first, int nl, int total, int dpr, int ystep, int dstep, StageCursor &cur)
{
    uint32_t v[N];  // (only the data stays in registers: the store loop walks the positions again)
    StageCursor ld = cur;
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const int bx = 4 * ld.d, cx = min(bx, rowbytes - 4);
        v[u] = *reinterpret_cast<const u32_unaligned *>(src + (size_t)min(ld.y, rows - 1) * pitch + cx) >> (8 * (bx - cx));
        ld.y += ystep;
        ld.d += dstep;
        if (ld.d >= dpr) { ld.d -= dpr; ++ld.y; }
    }
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const int bx = 4 * cur.d;
        const int nb = (first + u * nl < total && cur.y < rows) ? min(4, rowbytes - bx) : 0, off = cur.y * rowbytes + bx;
        if (nb == 4 && !(off & 3)) *reinterpret_cast<uint32_t *>(dst + off) = v[u];
        else
            for (int j = 0; j < nb; ++j) dst[off + j] = (uint8_t)(v[u] >> (8 * j));
        cur.y += ystep;
        cur.d += dstep;
        if (cur.d >= dpr) { cur.d -= dpr; ++cur.y; }
    }
}

__device__ __forceinline__ void tail_stage(const uint8_t *__restrict__ src, size_t pitch, int rows_, int rowbytes_, uint8_t *dst, int ql,
                                           int nl)
{
    // (the quadrant's size sits in vector registers -- it was selected per wavefront -- so the compiler would predicate on
    // it lane by lane: readfirstlane makes the branches below scalar ones)
    const int rows = __builtin_amdgcn_readfirstlane(rows_), rowbytes = __builtin_amdgcn_readfirstlane(rowbytes_);
    const int dpr = (rowbytes + 3) >> 2, total = rows * dpr;
    if (rowbytes >= 4) {
        const int ystep = nl / dpr, dstep = nl - ystep * dpr;
        StageCursor cur;
        cur.y = ql / dpr;
        cur.d = ql - cur.y * dpr;
        int first = ql;
        // eight loads in flight per lane (a 1024-thread block has 128 registers per lane: 24 at once -- the whole of a
        // first-level quadrant in one round trip -- spilled 200 of them)
        for (; first - ql + 4 * nl < total; first += 8 * nl)  // (uniform trip counts)
            tail_stage_n<8>(src, pitch, rows, rowbytes, dst, first, nl, total, dpr, ystep, dstep, cur);
        for (; first - ql < total; first += 4 * nl)
            tail_stage_n<4>(src, pitch, rows, rowbytes, dst, first, nl, total, dpr, ystep, dstep, cur);
        return;
    }
    for (int i = ql; i < total; i += nl) {  // rows of one to three bytes: dpr == 1
        const uint8_t *p = src + (size_t)i * pitch;
        for (int j = 0; j < rowbytes; ++j) dst[i * rowbytes + j] = p[j];
    }
}

// Leaf sums of the pairwise recursion (level nlev of `tree`), eight lanes per leaf; rgb: the quadrant's bytes in raster
// order.  grp: this lane's group among the quadrant's ngrp groups, sub: its accumulator.
template <bool VAR>
__device__ void tail_leaves(const Elem<VAR> &el, const uint8_t *rgb, TailTree &tree, int nlev, int grp, int ngrp, int sub)
{
    const int nLeaf = tree.cnt[nlev];
    for (int i = grp; i < nLeaf; i += ngrp) {
        const int len = tree.len[nlev][i];
        const uint8_t *p = rgb + 3 * (int)tree.off[nlev][i];
        float r[3];
        if (len < 8) {
            r[0] = r[1] = r[2] = 0.0f;
            for (int e = 0; e < len; ++e)
#pragma unroll
                for (int c = 0; c < 3; ++c) r[c] += el.get(p, 3 * e + c, c);
        } else {
            const int n8 = len - (len % 8);
            // the (at most 16) elements of accumulator `sub` are read up front: one LDS round trip per leaf, not per element
            uint8_t raw[16][3];
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int e = min(8 * it + sub, len - 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) raw[it][c] = p[3 * e + c];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) r[c] = el.get(raw[0], c, c);
#pragma unroll
            for (int it = 1; it < 16; ++it)
                if (8 * it + sub < n8) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) r[c] += el.get(raw[it], c, c);
                }
#pragma unroll
            for (int c = 0; c < 3; ++c) {  // ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7))
                r[c] += __shfl_xor(r[c], 1);
                r[c] += __shfl_xor(r[c], 2);
                r[c] += __shfl_xor(r[c], 4);
            }
            for (int e = n8; e < len; ++e)
#pragma unroll
                for (int c = 0; c < 3; ++c) r[c] += el.get(p, 3 * e + c, c);
        }
        if (sub == 0) {
            tree.val[nlev & 1][0][i] = r[0]; tree.val[nlev & 1][1][i] = r[1]; tree.val[nlev & 1][2][i] = r[2];
        }
    }
}

// Sobel gradients of rows [r0, r1) of the quadrant: a lane walks down a column with the separable row terms of the
// three rows it needs in registers.  Writes magnitude | direction, state 0.
__device__ void tail_gradients(const uint8_t *sg, uint16_t *st, int rows, int cols, int r0, int r1, int lane)
{
    for (int x = lane; x < cols; x += 64) {
        const int xm = max(x - 1, 0), xp = min(x + 1, cols - 1);
        auto hrow = [&](int y, int &d, int &s) {
            const int o = min(max(y, 0), rows - 1) * cols;
            const int a = sg[o + xm], b = sg[o + x], c = sg[o + xp];
            d = c - a;
            s = a + 2 * b + c;
        };
        int d0, s0, d1, s1, d2, s2;
        hrow(r0 - 1, d0, s0);
        hrow(r0, d1, s1);
#pragma unroll 4
        for (int y = r0; y < r1; ++y) {
            hrow(y + 1, d2, s2);
            const int dx = d0 + 2 * d1 + d2, dy = s2 - s0;
            const int ax = abs(dx), ay = abs(dy) << 15;
            const int tg22x = ax * 13573;  // (int)(tan(22.5deg) * 2^15 + 0.5)
            const int dir = ay < tg22x ? 0 : ay > tg22x + (ax << 16) ? 1 : (dx ^ dy) < 0 ? 3 : 2;
            st[y * cols + x] = (uint16_t)((ax + abs(dy)) | (dir << 11));
            d0 = d1; s0 = s1; d1 = d2; s1 = s2;
        }
    }
}

// Non-maximum suppression and the double threshold for rows [r0, r1); returns whether this lane kept a strong pixel.
__device__ bool tail_suppress(uint16_t *st, int rows, int cols, int r0, int r1, int lane, int low, int high)
{
    bool strong = false;
    for (int y = r0; y < r1; ++y)
        for (int x = lane; x < cols; x += 64) {
            const int e = y * cols + x;
            const int w = st[e], m = w & 2047, dir = (w >> 11) & 3;
            if (m <= low) continue;
            int o1, o2;  // the two neighbours along the gradient
            bool in1, in2;
            if (dir == 0) { o1 = -1; o2 = 1; in1 = x > 0; in2 = x + 1 < cols; }
            else if (dir == 1) { o1 = -cols; o2 = cols; in1 = y > 0; in2 = y + 1 < rows; }
            else if (dir == 2) { o1 = -cols - 1; o2 = cols + 1; in1 = y > 0 && x > 0; in2 = y + 1 < rows && x + 1 < cols; }
            else { o1 = -cols + 1; o2 = cols - 1; in1 = y > 0 && x + 1 < cols; in2 = y + 1 < rows && x > 0; }
            const int n1 = in1 ? st[e + o1] & 2047 : 0, n2 = in2 ? st[e + o2] & 2047 : 0;
            if (m > n1 && (m > n2 || (dir < 2 && m == n2))) {
                st[e] = (uint16_t)(w | ((m > high ? 2 : 1) << 13));  // the magnitude bits stay: neighbours still read them
                strong = strong || m > high;
            }
        }
    return strong;
}

// Hysteresis on the state words, one wavefront; returns the number of edge pixels.
__device__ uint32_t tail_hysteresis(uint16_t *st, int rows, int cols, int lane)
{
    const int n = rows * cols, nsteps = (n + 63) / 64;
    for (int sweep = 0;; ++sweep) {
        bool changed = false;
        for (int k = 0; k < nsteps; ++k) {
            const int step = (sweep & 1) ? nsteps - 1 - k : k;
            const int e = step * 64 + lane;
            const bool in = e < n;
            const int state = in ? (st[e] >> 13) & 3 : 0;
            const uint64_t weak = __ballot(state == 1);
            if (!weak) continue;
            const int y = in ? e / cols : 0, x = in ? e - y * cols : 0;
            bool hit = false;
            if (state == 1) {
                const bool up = y > 0, dn = y + 1 < rows, lf = x > 0, rt = x + 1 < cols;
                auto S = [&](int o) { return (st[e + o] >> 13) == 2; };
                hit = (up && (S(-cols) || (lf && S(-cols - 1)) || (rt && S(-cols + 1)))) ||
                      (dn && (S(cols) || (lf && S(cols - 1)) || (rt && S(cols + 1)))) || (lf && S(-1)) || (rt && S(1));
            }
            // close the horizontal runs inside the step: a weak pixel next to an edge pixel of the same row is an edge
            uint64_t f = __ballot(state == 2 || hit);
            uint64_t pu = weak & ~__ballot(x == 0);               // may take from the pixel before it
            uint64_t pd = weak & ~__ballot(in && x == cols - 1);  // may take from the pixel after it
            // occluded fill towards higher then lower lanes: every seed reaches both ends of its run
#pragma unroll
            for (int sft = 1; sft < 64; sft <<= 1) {
                f |= pu & (f << sft);
                pu &= pu << sft;
            }
#pragma unroll
            for (int sft = 1; sft < 64; sft <<= 1) {
                f |= pd & (f >> sft);
                pd &= pd >> sft;
            }
            if (state == 1 && ((f >> lane) & 1)) {
                st[e] = (uint16_t)((st[e] & 0x1fff) | (2 << 13));
                changed = true;
            }
            tail_wave_sync();
        }
        if (!__any(changed)) break;
    }
    uint32_t cnt = 0;
    for (int e = lane; e < n; e += 64) cnt += (st[e] >> 13) == 2;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    return cnt;
}

__global__ void __launch_bounds__(256 * kTailSub) k_q_tail(const uint8_t *__restrict__ in, const int32_t *__restrict__ kind,
                                                           const uint8_t *__restrict__ gray, Region *__restrict__ blk, int H,
                                                           int W, int level0, int min_size, TraceRec *__restrict__ trace,
                                                           float *__restrict__ A, int use_hist, uint32_t *__restrict__ status)
{
#ifdef UWIE_TAIL_PROF
    const bool hist_mode = use_hist;  // (profiling build: the trace carries phase times, not scores)
#else
    const bool hist_mode = use_hist && !trace;  // (recorded scores are the exact ones)
    const bool check_mode = hist_mode && use_hist == 2;  // tuning q_hist = 3: intervals AND exact scores, compared below
#endif
    extern __shared__ __attribute__((aligned(16))) uint8_t tail_lds[];
    __shared__ double s_q[4];
    __shared__ float s_sum[4][3];
    __shared__ int s_strong[4], s_nlev[4];
    __shared__ uint32_t s_edges[4];
    __shared__ double s_ilo[4], s_ihi[4], s_sv[4][6];
    __shared__ int s_dec;
    __shared__ double s_clo[4], s_chi[4];  // (check mode) the level's four intervals with the edge term
    const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int qd = wid / kTailSub, sw = wid % kTailSub, ql = sw * 64 + lane;  // quadrant, wavefront and lane inside it
    uint8_t *pix = tail_lds + (size_t)qd * kTailQuadBytes;
    TailTree &tree = *reinterpret_cast<TailTree *>(pix + kTailPixBytes);
    uint8_t *sg = pix;                                                  // after the sums: gray bytes ...
    uint16_t *st = reinterpret_cast<uint16_t *>(pix + kNpChunk);        // ... and the state words
    Region k = blk[b];
    const int knd = kind ? kind[b] : 0;
    const uint8_t *img = in + (size_t)b * H * W * 3;
    const uint8_t *g = gray + (size_t)b * H * W;
    for (int level = level0; level < kMaxLevels; ++level) {
        if (k.rows <= min_size || k.cols <= min_size) break;  // six_stadigy.py:76
        const int mr = k.rows / 2, mc = k.cols / 2;           // six_stadigy.py:85-86
        const Region r = qd == 0   ? Region{b, k.y0, k.x0, mr, mc}
                         : qd == 1 ? Region{b, k.y0, k.x0 + mc, mr, k.cols - mc}
                         : qd == 2 ? Region{b, k.y0 + mr, k.x0, k.rows - mr, mc}
                                   : Region{b, k.y0 + mr, k.x0 + mc, k.rows - mr, k.cols - mc};
        const int n = r.rows * r.cols;
#ifdef UWIE_TAIL_PROF
        const uint64_t c0 = wall_clock64();
        const uint64_t k0 = clock64();
#endif
        // Round 3: the level is decided from the quadrants' byte histograms when the score intervals separate (see
        // k_q_decide: same sums, same bounds, nch = 1); the NumPy-order sums below only run otherwise, or when the scores
        // themselves are recorded.  The histogram borrows the tree's LDS (the tree is only built for the exact sums).
        uint32_t *hq = reinterpret_cast<uint32_t *>(&tree);
        static_assert(sizeof(TailTree) >= 768 * sizeof(uint32_t), "the quadrant's histogram fits the tree's space");
        if (threadIdx.x < 4) s_strong[threadIdx.x] = 0;
        tail_stage(img + ((size_t)r.y0 * W + r.x0) * 3, (size_t)W * 3, r.rows, r.cols * 3, pix, ql, 64 * kTailSub);
        if (hist_mode) {
            for (int i = ql; i < 768; i += 64 * kTailSub) hq[i] = 0;
            __syncthreads();
            {   // a lane counts runs of equal bytes in registers: a flat quadrant would otherwise put every one of its pixels on
                // three LDS addresses (same-address atomics serialise).  Its pixels are 256 apart, so that neighbouring lanes
                // read neighbouring bytes (consecutive pixels per lane: 96-byte lane stride, 16-way bank conflicts)
                uint32_t cur[3] = {0, 0, 0}, run[3] = {0, 0, 0};
                for (int e = ql; e < n; e += 64 * kTailSub) {
                    const uint8_t *p = pix + (size_t)e * 3;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const uint32_t v = p[c];
                        if (v != cur[c]) {
                            if (run[c]) atomicAdd(&hq[c * 256 + cur[c]], run[c]);
                            cur[c] = v;
                            run[c] = 0;
                        }
                        ++run[c];
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (run[c]) atomicAdd(&hq[c * 256 + cur[c]], run[c]);
            }
            __syncthreads();
            // the score interval without the edge term (added once the Canny count is known): wavefronts 0 .. 2 of the
            // quadrant take one channel's sums each, lane 0 of wavefront 0 the bounds
            constexpr double u = 0x1p-24, kSafe = 1.25;
            const double nd = (double)n;
            if (sw < 3) {
                const int c = sw;
                const float a = px_atten(knd, c) ? 0.85f : 1.0f;
                double xs[4], ns[4], acc = 0.0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int v = lane + 64 * j;
                    xs[j] = (double)(px_norm_fast((uint32_t)v) * a);
                    ns[j] = (double)hq[c * 256 + v];
                    acc += ns[j] * xs[j];
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
                const double Sc = acc, m = acc / nd;
                acc = 0.0;
#pragma unroll
                for (int j = 0; j < 4; ++j) acc += ns[j] * (xs[j] - m) * (xs[j] - m);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
                if (lane == 0) {
                    s_sv[qd][c] = Sc;
                    s_sv[qd][3 + c] = acc;
                }
            }
            __syncthreads();
            if (sw == 0 && lane == 0) {
                const double eS = kSafe * 36.0 * u, eV = kSafe * 39.0 * u;
                double Slo[3], Shi[3], Vlo[3], Vhi[3];
                for (int c = 0; c < 3; ++c) {
                    const double Sc = s_sv[qd][c], Vc = s_sv[qd][3 + c];
                    const double dm = (eS + 2 * u) * fabs(Sc / nd);
                    Slo[c] = Sc * (1.0 - eS); Shi[c] = Sc * (1.0 + eS);
                    Vlo[c] = Vc * (1.0 - eV); Vhi[c] = (Vc + nd * dm * dm) * (1.0 + eV);
                }
                const double t1lo = (Slo[0] + Slo[1] + Slo[2]) / (3.0 * nd), t1hi = (Shi[0] + Shi[1] + Shi[2]) / (3.0 * nd);
                const double t2lo = (Slo[2] + Slo[1] - 2.0 * Shi[0]) / nd, t2hi = (Shi[2] + Shi[1] - 2.0 * Slo[0]) / nd;
                const double t3lo = (Vlo[0] + Vlo[1] + Vlo[2]) / (3.0 * nd), t3hi = (Vhi[0] + Vhi[1] + Vhi[2]) / (3.0 * nd);
                const double slack = kSafe * 8.0 * u * ((Shi[0] + Shi[1] + Shi[2]) / (3.0 * nd) + (Shi[2] + Shi[1] + 2.0 * Shi[0]) / nd + t3hi) + 1e-300;
                s_ilo[qd] = (t1lo + t2lo) - t3hi - slack;
                s_ihi[qd] = (t1hi + t2hi) - t3lo + slack;
            }
        }
        __syncthreads();
#ifdef UWIE_TAIL_PROF
        const uint64_t c1 = wall_clock64();
#endif
        // the RGB bytes are done with (for now): gray comes in
        tail_stage(g + (size_t)r.y0 * W + r.x0, (size_t)W, r.rows, r.cols, sg, ql, 64 * kTailSub);
        __syncthreads();
#ifdef UWIE_TAIL_PROF
        const uint64_t c2 = wall_clock64();
#endif
        // cv2.Canny(gray, 50, 150) (six_stadigy.py:150): the quadrant's rows are split over its wavefronts
        const int r0 = sw * r.rows / kTailSub, r1 = (sw + 1) * r.rows / kTailSub;
        tail_gradients(sg, st, r.rows, r.cols, r0, r1, lane);
        __syncthreads();
        if (__any(tail_suppress(st, r.rows, r.cols, r0, r1, lane, 50, 150)) && lane == 0) s_strong[qd] = 1;
        __syncthreads();
        if (sw == 0) {
            const uint32_t edges = s_strong[qd] ? tail_hysteresis(st, r.rows, r.cols, lane) : 0u;  // no strong pixel: no edge
            if (lane == 0) s_edges[qd] = edges;
        }
        __syncthreads();
        int arg = 0;
        bool sure = false;
        if (hist_mode) {
            if (threadIdx.x == 0) {  // one thread decides (its doubles would cost every lane registers), everybody reads
                double lo[4], hi[4];
                const int rows4[4] = {mr, mr, k.rows - mr, k.rows - mr}, cols4[4] = {mc, k.cols - mc, mc, k.cols - mc};
                for (int q = 0; q < 4; ++q) {
                    const double t4 = (double)s_edges[q] / ((double)rows4[q] * (double)cols4[q]);
                    lo[q] = s_ilo[q] - t4;
                    hi[q] = s_ihi[q] - t4;
                    s_clo[q] = lo[q];
                    s_chi[q] = hi[q];
                }
                int best = 0;
                for (int q = 1; q < 4; ++q)
                    if (lo[q] + hi[q] > lo[best] + hi[best]) best = q;
                bool ok = true;
                for (int q = 0; q < 4; ++q)
                    if (q != best && !(lo[best] > hi[q])) ok = false;
                s_dec = best | (ok ? 256 : 0);
            }
            __syncthreads();
            arg = s_dec & 255;
            sure = (s_dec >> 8) != 0 && !check_mode;
        }
        if (!sure) {  // (workgroup-uniform) the exact NumPy-order sums: the RGB bytes come in again
            __syncthreads();
            int nlev = 0;
            if (sw == 0) {
                nlev = pairwise_build<float, 128, 8, true>(n, lane, tree);
                if (lane == 0) s_nlev[qd] = nlev;
            }
            tail_stage(img + ((size_t)r.y0 * W + r.x0) * 3, (size_t)W * 3, r.rows, r.cols * 3, pix, ql, 64 * kTailSub);
            __syncthreads();
            nlev = s_nlev[qd];
            float S[3], V[3];
            {
                Elem<false> el;
                el.img = nullptr; el.W = 0; el.set_kind(knd);
                tail_leaves<false>(el, pix, tree, nlev, ql >> 3, 8 * kTailSub, lane & 7);
            }
            __syncthreads();
            if (sw == 0) {
                pairwise_combine<float, 128, 8, true>(nlev, lane, tree, S);
                if (lane < 3) s_sum[qd][lane] = S[lane];
            }
            __syncthreads();
            {
                Elem<true> el;
                el.img = nullptr; el.W = 0; el.set_kind(knd);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    S[c] = s_sum[qd][c];
                    el.mean[c] = (float)((double)S[c] / (double)n);  // numpy/_core/_methods.py:_mean
                }
                tail_leaves<true>(el, pix, tree, nlev, ql >> 3, 8 * kTailSub, lane & 7);
            }
            __syncthreads();
            if (sw == 0) {
                pairwise_combine<float, 128, 8, true>(nlev, lane, tree, V);
                if (lane == 0) {
                    const float t1 = ((S[0] + S[1]) + S[2]) / (float)(3 * (long long)n);
                    const float t2 = ((S[2] + S[1]) - 2.0f * S[0]) / (float)n;
                    const float v0 = V[0] / (float)n, v1 = V[1] / (float)n, v2 = V[2] / (float)n;
                    const float t3 = ((v0 + v1) + v2) / 3.0f;
                    const double t4 = (double)s_edges[qd] / (double)n;  // int64 / int -> float64
                    s_q[qd] = (double)((t1 + t2) - t3) - t4;
                }
            }
            __syncthreads();
            double score[4];
            for (int q = 0; q < 4; ++q) score[q] = s_q[q];
            arg = 0;
            for (int q = 1; q < 4; ++q)
                if (score[q] > score[arg]) arg = q;  // np.argmax: first maximum
            if (check_mode && threadIdx.x == 0 && status) {  // (see k_q_select)
                bool bad = false;
                for (int q = 0; q < 4; ++q) bad = bad || !(score[q] >= s_clo[q] && score[q] <= s_chi[q]);
                if ((s_dec >> 8) != 0 && (s_dec & 255) != arg) bad = true;
                if (bad) atomicOr(status, (uint32_t)UWIE_STATUS_QTREE_BOUNDS);
            }
            if (trace && threadIdx.x == 0) {
                TraceRec &t = trace[b * kMaxLevels + level];
                t.y0 = k.y0; t.x0 = k.x0; t.rows = k.rows; t.cols = k.cols;
                for (int q = 0; q < 4; ++q) t.score[q] = score[q];
            }
        }
#ifdef UWIE_TAIL_PROF
        if (trace && threadIdx.x == 0) {  // 100 MHz ticks: staging + histogram, gray load, Canny + decision
            TraceRec &t = trace[b * kMaxLevels + level];
            const uint64_t c3 = wall_clock64();
            t.score[0] = (double)(c1 - c0); t.score[1] = (double)(c2 - c1); t.score[2] = (double)(c3 - c2);
            t.score[3] = (double)(clock64() - k0) / (double)(c3 - c0);
        }
#endif
        k = arg == 0   ? Region{b, k.y0, k.x0, mr, mc}
            : arg == 1 ? Region{b, k.y0, k.x0 + mc, mr, k.cols - mc}
            : arg == 2 ? Region{b, k.y0 + mr, k.x0, k.rows - mr, mc}
                       : Region{b, k.y0 + mr, k.x0 + mc, k.rows - mr, k.cols - mc};
        __syncthreads();  // s_q and the slices are rewritten by the next level
    }
    if (wid != 0) return;
    if (lane == 0) blk[b] = k;
    // get_brightest_pixel (six_stadigy.py:160-165): argmax of (r+g)+b over the leaf, first maximum in raster order
    const int n = k.rows * k.cols;
    float best = -1.0f;
    int bi = 0x7fffffff;
    for (int e = lane; e < n; e += 64) {
        const uint8_t *p = img + ((size_t)(k.y0 + e / k.cols) * W + k.x0 + e % k.cols) * 3;
        const float v = (px_val(p[0], false) + px_val(p[1], px_atten(knd, 1))) + px_val(p[2], px_atten(knd, 2));
        if (v > best) { best = v; bi = e; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) {
        const uint8_t *p = img + ((size_t)(k.y0 + bi / k.cols) * W + k.x0 + bi % k.cols) * 3;
        A[b * 3 + 0] = px_val(p[0], false);
        A[b * 3 + 1] = px_val(p[1], px_atten(knd, 1));
        A[b * 3 + 2] = px_val(p[2], px_atten(knd, 2));
    }
}

int max_chunks(Shape s) { return cdiv((long long)((s.H + 1) / 2) * ((s.W + 1) / 2), kNpChunk); }

LevelBufs carve_level(Carver &c, Shape s)
{
    LevelBufs L;
    const size_t nreg = (size_t)s.B * 4;
    L.blk = c.take<Region>(s.B);
    L.regs = c.take<Region>(nreg);
    L.csum = c.take<float>(nreg * max_chunks(s) * 3);
    L.csum_var = c.take<float>(nreg * max_chunks(s) * 3);
    L.tot = c.take<float>(nreg * 3);
    L.mean = c.take<float>(nreg * 3);
    L.vtot = c.take<float>(nreg * 3);
    L.edges = c.take<uint32_t>(nreg);
    L.hist = c.take<uint32_t>(nreg * 768);
    L.skip = c.take<uint8_t>(s.B);
    L.chk = c.take<double>((size_t)s.B * 9);
    return L;
}

}  // namespace

// np.sum / np.mean / np.var arithmetic of block[:, :, c] for a list of regions (see k_q_chunk_sums / k_q_combine):
// tot[reg*3+c] = sum, mean[...] = float32(float64(sum)/n), vtot[...] = sum of (x - mean)^2.  csum: scratch of
// nreg * maxChunks * 3 floats with maxChunks >= ceil(max_rows * max_cols / 8192).
int launch_region_stats(const uint8_t *d_in, const int32_t *d_kind, const Region *d_regs, int nreg, int max_rows,
                        int max_cols, Shape s, float *csum, int maxChunks, float *tot, float *mean, float *vtot,
                        hipStream_t st)
{
    const int nch = cdiv((long long)max_rows * max_cols, kNpChunk);
    UWIE_REQUIRE(nch <= maxChunks, "region_stats: scratch too small");
    UWIE_LAUNCH(k_q_chunk_sums<false>, dim3(nch, nreg), dim3(64), 0, st, d_in, d_kind, d_regs, mean, s.H, s.W, maxChunks, csum,
                (const float *)nullptr);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_q_combine<false>, dim3(nreg), dim3(64), 0, st, d_regs, csum, nreg, maxChunks, tot, mean);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_q_chunk_sums<true>, dim3(nch, nreg), dim3(64), 0, st, d_in, d_kind, d_regs, mean, s.H, s.W, maxChunks, csum,
                (const float *)nullptr);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_q_combine<true>, dim3(nreg), dim3(64), 0, st, d_regs, csum, nreg, maxChunks, vtot, mean);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

size_t airlight_ws_bytes(Shape s)
{
    Carver c(nullptr);
    carve_level(c, s);
    return c.total() + canny_ws_bytes(s);
}

// make_gray_shift != 0: d_gray is not filled yet; the level-0 sums pass writes it (GrayOut) when level 0 is a launched
// level, k_quant_gray otherwise.  UWIE_Q_GRAY_FUSE=0 always takes k_quant_gray.
// Level 0 is a launched level with histograms, the frame has whole groups of four pixels on either side of W / 2 and the
// gray plane is still to be written: cast detection's chunk pass may collect the level-0 histograms (k_chunk_hist_quad).
bool entry_fuse_takes(Shape s, int min_size)
{
    return tune().entry_fuse && tune().q_hist != 0 && gray_strong_takes(s) && s.W <= 16384 && s.H > min_size && s.W > min_size &&
           (long long)((s.H + 1) / 2) * ((s.W + 1) / 2) > kNpChunk;
}

int launch_airlight(uwie_ctx *ctx, const uint8_t *d_in, const int32_t *d_kind, uint8_t *d_gray, Shape s, int min_size,
                    float *d_A, void *d_trace, void *ws, hipStream_t st, int make_gray_shift, const uint32_t *qpart)
{
    Carver c(ws);
    LevelBufs L = carve_level(c, s);
    void *canny_ws = c.take<char>(canny_ws_bytes(s));
    const int B = s.B, nreg = 4 * B;
    const bool one_setup = tune().q_hist != 0 && tune().canny_prepass;  // (the routes that use all of what k_q_setup prepares)
    if (one_setup) {
        UWIE_LAUNCH(k_q_setup, dim3(std::min(1024, cdiv((long long)B * 768, 256))), dim3(256), 0, st, L.blk, L.regs, L.edges,
                    canny_strong_flags(canny_ws, s), L.hist, B, s.H, s.W, min_size);
        UWIE_LAUNCH_CHECK();
    } else {
        UWIE_LAUNCH(k_init_blocks, dim3(cdiv(B, 64)), dim3(64), 0, st, L.blk, B, s.H, s.W);
        UWIE_LAUNCH_CHECK();
    }
    if (d_trace) UWIE_HIP_CHECK(hipMemsetAsync(d_trace, 0, (size_t)B * kMaxLevels * sizeof(TraceRec), st));
    const int maxChunks = max_chunks(s);
    int rmax = s.H, cmax = s.W;  // largest block any image can hold at this level
    if (!one_setup) {
        UWIE_LAUNCH(k_make_quadrants, dim3(cdiv(B, 64)), dim3(64), 0, st, L.blk, L.regs, B, min_size);
        UWIE_LAUNCH_CHECK();
        UWIE_HIP_CHECK(hipMemsetAsync(L.edges, 0, sizeof(uint32_t) * nreg, st));
    }
    constexpr bool use_tail = true, fuse_gray = true;  // (the small levels in one launch; level 0 writes the gray plane)
    bool gray_pending = make_gray_shift != 0;
    {
        const bool level0_launched = s.H > min_size && s.W > min_size &&
                                     !(use_tail && (long long)((s.H + 1) / 2) * ((s.W + 1) / 2) <= kNpChunk);
        if (gray_pending && !(fuse_gray && level0_launched)) {
            const int rc = launch_quant_gray(d_in, d_kind, d_gray, s, make_gray_shift, st);
            if (rc != UWIE_OK) return rc;
            gray_pending = false;
        }
    }
    int level = 0;
    for (; level < kMaxLevels && rmax > min_size && cmax > min_size; ++level) {
        const int qr = (rmax + 1) / 2, qc = (cmax + 1) / 2;  // largest quadrant
        if (use_tail && (long long)qr * qc <= kNpChunk) break;  // k_q_tail walks the rest
        const int nch = cdiv((long long)qr * qc, kNpChunk);
        // Round 3: byte histograms of the quadrants (+ the gray plane on level 0) -> Canny -> decide from the score
        // intervals; the exact kernels below return at once for every decided image.  tuning q_hist = 0: exact only.
        const bool use_hist = tune().q_hist != 0;
        if (use_hist) {
            if (level == 0 && !one_setup)
                UWIE_HIP_CHECK(hipMemsetAsync(L.hist, 0, sizeof(uint32_t) * (size_t)nreg * 768, st));  // (k_q_decide clears after use)
            // ~64 K pixels per block, at least ~2048 blocks when the job has them
            int nblk = std::max(1, cdiv((long long)qr * qc, 65536));
            while (nblk * 2 <= qr && (long long)nblk * nreg < 2048 && (long long)qr * qc / (nblk * 2) >= 8192) nblk *= 2;
            nblk = std::min(nblk, qr);
            // (rows of fewer than 256 four-pixel groups: several rows per step; every quadrant of the level is at most qc wide)
            const bool narrow = cdiv(qc, 4) < 256;
            const auto k_q_hist_gray = k_q_hist<true, false>, k_q_hist_gray_narrow = k_q_hist<true, true>;
            const auto k_q_hist_wide = k_q_hist<false, false>, k_q_hist_narrow = k_q_hist<false, true>;
            // round 4 (tuning entry_fuse): the gray plane comes out of the Canny pre-pass of level 0 (k_gray_strong reads the
            // frame's bytes once for both), not out of this level's histogram pass
            const bool gray_by_prepass = gray_pending && level == 0 && tune().entry_fuse && gray_strong_takes(s);
            bool prepass_done = false;
            // round 4, levels >= 1 (tuning entry_fuse): histograms and the Canny pre-pass of the level in one sweep (k_hist_strong);
            // every block of the level is W >> level columns wide when that divides evenly
            const bool hist_by_prepass = level > 0 && !gray_pending && tune().entry_fuse && tune().canny_prepass &&
                                         (s.W >> level) << level == s.W && (s.W >> level) % 8 == 0 && qc >= 8;
            if (hist_by_prepass) {
                int rc = launch_hist_strong(d_in, d_gray, s, L.regs, nreg, qr, qc, 150, L.hist, canny_ws, st);
                if (rc != UWIE_OK) return rc;
                prepass_done = true;
            } else if (level == 0 && qpart) {  // the histograms were counted by cast detection's chunk pass
                UWIE_REQUIRE(gray_by_prepass || !gray_pending, "airlight: quadrant shares without the fused gray pass");
                int rc = launch_quad_hist_reduce(qpart, s, L.hist, st);
                if (rc != UWIE_OK) return rc;
            } else if (gray_pending && !gray_by_prepass) {
                if (narrow)
                    UWIE_LAUNCH(k_q_hist_gray_narrow, dim3(nblk, nreg), dim3(256), 0, st, d_in, d_kind, L.regs, s.H, s.W, L.hist, d_gray,
                                make_gray_shift);
                else
                    UWIE_LAUNCH(k_q_hist_gray, dim3(nblk, nreg), dim3(256), 0, st, d_in, d_kind, L.regs, s.H, s.W, L.hist, d_gray,
                                make_gray_shift);
                gray_pending = false;
            } else if (narrow) {
                UWIE_LAUNCH(k_q_hist_narrow, dim3(nblk, nreg), dim3(256), 0, st, d_in, d_kind, L.regs, s.H, s.W, L.hist,
                            (uint8_t *)nullptr, 15);
            } else {
                UWIE_LAUNCH(k_q_hist_wide, dim3(nblk, nreg), dim3(256), 0, st, d_in, d_kind, L.regs, s.H, s.W, L.hist,
                            (uint8_t *)nullptr, 15);
            }
            UWIE_LAUNCH_CHECK();
            if (gray_by_prepass) {
                int rc = launch_gray_strong(d_in, d_kind, d_gray, s, L.regs, nreg, qr, qc, 150, make_gray_shift, canny_ws, st);
                if (rc != UWIE_OK) return rc;
                gray_pending = false;
                prepass_done = true;
            }
            int rc = launch_canny(d_gray, s, L.regs, nreg, qr, qc, 50, 150, L.edges, nullptr, canny_ws, st, true, level > 0 || one_setup,
                                  prepass_done);
            if (rc != UWIE_OK) return rc;
            UWIE_LAUNCH(k_q_decide, dim3(B), dim3(256), 0, st, L.blk, L.regs, (const uint32_t *)L.hist, L.edges, d_kind, min_size,
                        (d_trace || tune().q_hist == 2) ? 1 : 0, L.skip, canny_strong_flags(canny_ws, s),
                        tune().q_hist == 3 ? L.chk : (double *)nullptr);
            UWIE_LAUNCH_CHECK();
        }
        const uint8_t *skip = use_hist ? L.skip : nullptr;
        if (gray_pending) {  // level 0: its four quadrants are the whole frame
            UWIE_LAUNCH((k_q_chunk_sums<false, true>), dim3(nch, nreg), dim3(64), 0, st, d_in, d_kind, L.regs, L.mean, s.H, s.W,
                        maxChunks, L.csum, (const float *)nullptr, d_gray, make_gray_shift);
            gray_pending = false;
        } else {
            UWIE_LAUNCH(k_q_chunk_sums<false>, dim3(nch, nreg), dim3(64), 0, st, d_in, d_kind, L.regs, L.mean, s.H, s.W,
                        maxChunks, L.csum, (const float *)nullptr, (uint8_t *)nullptr, 15, skip);
        }
        UWIE_LAUNCH_CHECK();
        UWIE_LAUNCH(k_q_chunk_sums<true>, dim3(nch, nreg), dim3(64), 0, st, d_in, d_kind, L.regs, L.mean, s.H,
                           s.W, maxChunks, L.csum_var, (const float *)L.csum, (uint8_t *)nullptr, 15, skip);
        UWIE_LAUNCH_CHECK();
        if (!use_hist) {
            int rc = launch_canny(d_gray, s, L.regs, nreg, qr, qc, 50, 150, L.edges, nullptr, canny_ws, st, true);
            if (rc != UWIE_OK) return rc;
        }
        UWIE_LAUNCH(k_q_select, dim3(B), dim3(64), 0, st, L.blk, L.regs, (const float *)L.csum, (const float *)L.csum_var,
                           maxChunks, L.edges, B, level, min_size, (TraceRec *)d_trace, skip,
                           use_hist && tune().q_hist == 3 ? (const double *)L.chk : (const double *)nullptr, ctx ? ctx->d_status : nullptr);
        UWIE_LAUNCH_CHECK();
        rmax = qr;
        cmax = qc;
    }
    // the remaining levels (none when the walk above already reached the leaves) and the leaf's brightest pixel
    {
        // more than 64 KB of LDS has to be asked for, once per context (= per device: a process may hold several)
        if (!ctx || !ctx->attr_q_tail) {
            UWIE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_q_tail), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               4 * kTailQuadBytes));
            if (ctx) ctx->attr_q_tail = true;
        }
    }
    UWIE_LAUNCH(k_q_tail, dim3(B), dim3(256 * kTailSub), 4 * kTailQuadBytes, st, d_in, d_kind, d_gray, L.blk, s.H, s.W, level, min_size,
                (TraceRec *)d_trace, d_A, tune().q_hist == 1 ? 1 : tune().q_hist == 3 ? 2 : 0, ctx ? ctx->d_status : nullptr);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

}  // namespace uwie
