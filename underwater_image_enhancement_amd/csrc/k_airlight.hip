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
};

__global__ void k_init_blocks(Region *blk, int B, int H, int W)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) blk[b] = Region{b, 0, 0, H, W};
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
    int W, kind;
    float mean[3];
    __device__ __forceinline__ float get(const uint8_t *p, int c) const { return get(p, c, c); }
    // value of byte p[i] taken as channel c
    __device__ __forceinline__ float get(const uint8_t *p, int i, int c) const
    {
        float v = px_norm_fast(p[i]);
        if (px_atten(kind, c)) v = v * 0.85f;
        if (VAR) {
            const float d = v - mean[c];
            v = d * d;
        }
        return v;
    }
};

// NumPy's pairwise_sum on a block of n <= 128 elements starting at raster element e0, three channels at once.
template <bool VAR>
__device__ void leaf_sum3(const Elem<VAR> &el, const Region &r, int e0, int n, float out[3])
{
    int ly = e0 / r.cols, lx = e0 % r.cols;
    const uint8_t *p = el.img + ((size_t)(r.y0 + ly) * el.W + r.x0 + lx) * 3;
    auto step = [&]() {
        ++lx;
        p += 3;
        if (lx == r.cols) {
            lx = 0;
            p += (size_t)(el.W - r.cols) * 3;
        }
    };
    if (n < 8) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int i = 0; i < n; ++i) {
            a0 += el.get(p, 0);
            a1 += el.get(p, 1);
            a2 += el.get(p, 2);
            step();
        }
        out[0] = a0; out[1] = a1; out[2] = a2;
        return;
    }
    float acc[3][8];
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
    float res[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) res[c] = tree8(acc[c]);
    for (; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < 3; ++c) res[c] += el.get(p, c);
        step();
    }
    out[0] = res[0]; out[1] = res[1]; out[2] = res[2];
}

// The same for a full 128-element leaf, eight pixels per step: when the eight pixels lie in one row of the region
// their 24 bytes come in as six (unaligned) dword loads instead of 24 byte loads.
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint4 __attribute__((aligned(1))) u128_unaligned;

template <bool VAR>
__device__ void leaf_sum3_full(const Elem<VAR> &el, const Region &r, int e0, float out[3])
{
    int ly = e0 / r.cols, lx = e0 % r.cols;
    const uint8_t *p = el.img + ((size_t)(r.y0 + ly) * el.W + r.x0 + lx) * 3;
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
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int byte = 24 * it + 3 * j + c;
                    const uint8_t u = (uint8_t)(raw[byte >> 2] >> (8 * (byte & 3)));
                    const float v = el.get(&u, 0, c);
                    acc[c][j] = it == 0 ? v : acc[c][j] + v;
                }
        }
        out[0] = tree8(acc[0]);
        out[1] = tree8(acc[1]);
        out[2] = tree8(acc[2]);
        return;
    }
    for (int it = 0; it < 16; ++it) {
        uint8_t px[24];
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
            lx += 8;
            p += 24;
            if (lx == r.cols) {
                lx = 0;
                p += (size_t)(el.W - r.cols) * 3;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                px[3 * j] = p[0];
                px[3 * j + 1] = p[1];
                px[3 * j + 2] = p[2];
                ++lx;
                p += 3;
                if (lx == r.cols) {
                    lx = 0;
                    p += (size_t)(el.W - r.cols) * 3;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = el.get(px + 3 * j, c);
                acc[c][j] = it == 0 ? v : acc[c][j] + v;
            }
    }
    out[0] = tree8(acc[0]);
    out[1] = tree8(acc[1]);
    out[2] = tree8(acc[2]);
}

// One wavefront per (chunk, region).  csum[(reg*maxChunks + chunk)*3 + c] = pairwise sum of that chunk.
// VAR with csum_in != nullptr: the means come from the chunk sums of the previous pass (added sequentially by lanes 0..2,
// as k_q_combine would), so the quadtree needs no combine launch between its two passes.
template <bool VAR>
__global__ void __launch_bounds__(64) k_q_chunk_sums(const uint8_t *__restrict__ in, const int32_t *__restrict__ kind,
                                                     const Region *__restrict__ regs, const float *__restrict__ mean,
                                                     int H, int W, int maxChunks, float *__restrict__ csum,
                                                     const float *__restrict__ csum_in = nullptr)
{
    const int reg = blockIdx.y, ci = blockIdx.x, lane = threadIdx.x;
    const Region r = regs[reg];
    const int n = r.rows * r.cols;
    const int c0 = ci * kNpChunk;
    if (c0 >= n) return;
    const int len = min(kNpChunk, n - c0);

    __shared__ PairwiseTree tree;

    Elem<VAR> el;
    el.img = in + (size_t)r.img * H * W * 3;
    el.W = W;
    el.kind = kind ? kind[r.img] : 0;
    if (VAR) {
        if (csum_in) {
            // the chunk sums come into LDS with coalesced loads; lanes 0..2 then add them in order
            extern __shared__ float cs_l[];
            const int nch = (n + kNpChunk - 1) / kNpChunk;
            for (int i = lane; i < nch * 3; i += 64) cs_l[i] = csum_in[(size_t)reg * maxChunks * 3 + i];
            __syncthreads();
            float acc = 0.0f;
            if (lane < 3) {
                for (int k = 0; k < nch; ++k) acc = acc + cs_l[k * 3 + lane];
                acc = (float)((double)acc / (double)n);  // numpy/_core/_methods.py:_mean
            }
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

    if (len == kNpChunk) {
        // balanced tree: 64 leaves of 128, lane = leaf, butterfly == recursive halving
        float s[3];
        leaf_sum3_full<VAR>(el, r, c0 + lane * 128, s);
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
    pairwise_ragged(len, lane, tree, [&](int off, int l, float *s3) { leaf_sum3<VAR>(el, r, c0 + off, l, s3); }, res);
    if (lane == 0) { out[0] = res[0]; out[1] = res[1]; out[2] = res[2]; }
}

// Sequential accumulation of the chunk sums (NumPy adds each buffer's pairwise result into the running total),
// and for the first pass the mean: float32(float64(sum) / n) as numpy/_core/_methods.py:_mean does for a scalar result.
template <bool VAR>
__global__ void __launch_bounds__(64) k_q_combine(const Region *__restrict__ regs, const float *__restrict__ csum, int nreg,
                                                  int maxChunks, float *__restrict__ tot, float *__restrict__ mean)
{
    // one wavefront per region: the chunk sums come into LDS with coalesced loads, lanes 0..2 then add them in order
    extern __shared__ float cs[];
    const int reg = blockIdx.x, lane = threadIdx.x;
    const Region r = regs[reg];
    const int n = r.rows * r.cols;
    if (n == 0) return;
    const int nch = (n + kNpChunk - 1) / kNpChunk;
    for (int i = lane; i < nch * 3; i += 64) cs[i] = csum[(size_t)reg * maxChunks * 3 + i];
    __syncthreads();
    if (lane >= 3) return;
    float acc = 0.0f;
    for (int k = 0; k < nch; ++k) acc = acc + cs[k * 3 + lane];
    tot[reg * 3 + lane] = acc;
    if (!VAR) mean[reg * 3 + lane] = (float)((double)acc / (double)n);
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
                                                 TraceRec *__restrict__ trace)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    if (regs[b * 4].rows == 0) return;  // leaf reached earlier
    extern __shared__ float cs_l[];  // [4 quadrants][2 passes][nchMax * 3]
    int nchMax = 0;
    for (int q = 0; q < 4; ++q) {
        const Region r = regs[b * 4 + q];
        nchMax = max(nchMax, (r.rows * r.cols + kNpChunk - 1) / kNpChunk);
    }
    for (int q = 0; q < 4; ++q) {
        const Region r = regs[b * 4 + q];
        const int nch = (r.rows * r.cols + kNpChunk - 1) / kNpChunk;
        for (int i = lane; i < nch * 3; i += 64) {
            cs_l[(q * 2 + 0) * nchMax * 3 + i] = csum[(size_t)(b * 4 + q) * maxChunks * 3 + i];
            cs_l[(q * 2 + 1) * nchMax * 3 + i] = csum_var[(size_t)(b * 4 + q) * maxChunks * 3 + i];
        }
    }
    __syncthreads();
    float acc = 0.0f;
    if (lane < 24) {
        const int q = lane / 6, pass = (lane % 6) / 3, c = lane % 3;
        const Region r = regs[b * 4 + q];
        const int n = r.rows * r.cols, nch = (n + kNpChunk - 1) / kNpChunk;
        const float *cs = cs_l + (q * 2 + pass) * nchMax * 3 + c;
        for (int k = 0; k < nch; ++k) acc = acc + cs[k * 3];
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

// get_brightest_pixel (six_stadigy.py:160-165): argmax of (r+g)+b over the leaf, first maximum in raster order.
__global__ void __launch_bounds__(64) k_brightest(const uint8_t *__restrict__ in, const int32_t *__restrict__ kind,
                                                  const Region *__restrict__ blk, int H, int W, float *__restrict__ A)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const Region r = blk[b];
    const int k = kind ? kind[b] : 0;
    const uint8_t *img = in + (size_t)b * H * W * 3;
    const int n = r.rows * r.cols;
    float best = -1.0f;
    int bi = 0x7fffffff;
    for (int e = lane; e < n; e += 64) {
        const uint8_t *p = img + ((size_t)(r.y0 + e / r.cols) * W + r.x0 + e % r.cols) * 3;
        const float v = (px_val(p[0], false) + px_val(p[1], px_atten(k, 1))) + px_val(p[2], px_atten(k, 2));
        if (v > best) { best = v; bi = e; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) {
        const uint8_t *p = img + ((size_t)(r.y0 + bi / r.cols) * W + r.x0 + bi % r.cols) * 3;
        A[b * 3 + 0] = px_val(p[0], false);
        A[b * 3 + 1] = px_val(p[1], px_atten(k, 1));
        A[b * 3 + 2] = px_val(p[2], px_atten(k, 2));
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
    UWIE_LAUNCH(k_q_combine<false>, dim3(nreg), dim3(64), sizeof(float) * 3 * nch, st, d_regs, csum, nreg, maxChunks, tot, mean);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_q_chunk_sums<true>, dim3(nch, nreg), dim3(64), 0, st, d_in, d_kind, d_regs, mean, s.H, s.W, maxChunks, csum,
                (const float *)nullptr);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_q_combine<true>, dim3(nreg), dim3(64), sizeof(float) * 3 * nch, st, d_regs, csum, nreg, maxChunks, vtot, mean);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

size_t airlight_ws_bytes(Shape s)
{
    Carver c(nullptr);
    carve_level(c, s);
    return c.total() + canny_ws_bytes(s);
}

int launch_airlight(uwie_ctx *ctx, const uint8_t *d_in, const int32_t *d_kind, const uint8_t *d_gray, Shape s,
                    int min_size, float *d_A, void *d_trace, void *ws, hipStream_t st)
{
    (void)ctx;
    Carver c(ws);
    LevelBufs L = carve_level(c, s);
    void *canny_ws = c.take<char>(canny_ws_bytes(s));
    const int B = s.B, nreg = 4 * B;
    UWIE_LAUNCH(k_init_blocks, dim3(cdiv(B, 64)), dim3(64), 0, st, L.blk, B, s.H, s.W);
    UWIE_LAUNCH_CHECK();
    if (d_trace) UWIE_HIP_CHECK(hipMemsetAsync(d_trace, 0, (size_t)B * kMaxLevels * sizeof(TraceRec), st));
    const int maxChunks = max_chunks(s);
    int rmax = s.H, cmax = s.W;  // largest block any image can hold at this level
    UWIE_LAUNCH(k_make_quadrants, dim3(cdiv(B, 64)), dim3(64), 0, st, L.blk, L.regs, B, min_size);
    UWIE_LAUNCH_CHECK();
    UWIE_HIP_CHECK(hipMemsetAsync(L.edges, 0, sizeof(uint32_t) * nreg, st));
    for (int level = 0; level < kMaxLevels && rmax > min_size && cmax > min_size; ++level) {
        const int qr = (rmax + 1) / 2, qc = (cmax + 1) / 2;  // largest quadrant
        const int nch = cdiv((long long)qr * qc, kNpChunk);
        // sums -> squared deviations (means derived in the kernel) -> Canny -> select (totals derived in the kernel)
        UWIE_LAUNCH(k_q_chunk_sums<false>, dim3(nch, nreg), dim3(64), 0, st, d_in, d_kind, L.regs, L.mean, s.H,
                           s.W, maxChunks, L.csum, (const float *)nullptr);
        UWIE_LAUNCH_CHECK();
        UWIE_LAUNCH(k_q_chunk_sums<true>, dim3(nch, nreg), dim3(64), sizeof(float) * 3 * nch, st, d_in, d_kind, L.regs, L.mean, s.H,
                           s.W, maxChunks, L.csum_var, (const float *)L.csum);
        UWIE_LAUNCH_CHECK();
        int rc = launch_canny(d_gray, s, L.regs, nreg, qr, qc, 50, 150, L.edges, nullptr, canny_ws, st, true);
        if (rc != UWIE_OK) return rc;
        UWIE_LAUNCH(k_q_select, dim3(B), dim3(64), sizeof(float) * 24 * nch, st, L.blk, L.regs, (const float *)L.csum, (const float *)L.csum_var,
                           maxChunks, L.edges, B, level, min_size, (TraceRec *)d_trace);
        UWIE_LAUNCH_CHECK();
        rmax = qr;
        cmax = qc;
    }
    UWIE_LAUNCH(k_brightest, dim3(B), dim3(64), 0, st, d_in, d_kind, L.blk, s.H, s.W, d_A);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

}  // namespace uwie
