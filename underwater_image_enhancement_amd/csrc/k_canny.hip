// cv2.Canny(gray, low, high) with aperture 3 and the L1 gradient norm (six_stadigy.py:150,
// enhancement_strategies.py:181), evaluated on a list of rectangular REGIONS of the gray plane.  compute_Q
// runs Canny on every quadrant of the quadtree block separately (each quadrant is its own image: Sobel
// replicates ITS border and the magnitude outside it is 0), so a region here is one quadrant; the standalone
// entry point uses one region per frame.
//
// Stages (all integer arithmetic, bit-exact by construction):
//   k_canny_gradnms  32x64 tiles: the gray tile (2-pixel halo, replicated at the REGION border like Sobel's
//                  BORDER_REPLICATE on the quadrant) goes through LDS once; a thread then owns 4x2 pixels and keeps its
//                  6x8 gray neighbourhood in registers: separable Sobel on packed 16-bit pairs (v_pk_*), |dx|+|dy| for
//                  the 4x6 magnitudes it needs (0 outside the region, like OpenCV's zero-padded magnitude rows/columns),
//                  direction class and non-maximum suppression with OpenCV's fixed-point tan(22.5 deg) for its 8 pixels.
//                  Writes the map byte {weak, none, strong} of every pixel and the candidates (weak | strong) of each
//                  (tile, wavefront) into that wavefront's own list slot: no atomics, no second barrier.
//   k_canny_gradnms<true> (optional pre-pass, same tiles): gradient magnitudes only -> strong[region] = 1 if any pixel of
//                  the region exceeds the HIGH threshold.  Hysteresis keeps exactly the components that own such a
//                  pixel, so a region without one has no edge at all and its tiles skip everything else (smooth
//                  water: the sensor noise crosses the low threshold everywhere and the high one nowhere).
//   k_canny_union  8-connected components of the candidates: lock-free union-find on pixel indices (links always point
//                  to the smaller index; agent-scope atomics, so XCD placement is irrelevant)
//   k_canny_mark   walks the tile-local ROOTS only (k_canny_gradnms records each with its component's size and number of
//                  strong pixels): the global root of a part that owns a strong pixel is flagged
//   k_canny_emit   per-region edge counts = sizes of the parts whose global root is flagged (= hysteresis)
//   k_canny_paint  the edge map (standalone cv2.Canny only): candidates whose global root is flagged
// The component kernels walk the candidate lists (one wavefront per tile), not the frame: smooth frames have few
// candidates.
// Hysteresis is order independent (an edge pixel is a weak-or-strong pixel whose 8-connected component holds a
// strong one), so the component formulation equals OpenCV's stack-based flood fill.
// WEAKONLY (round 3, the edge COUNTS the quadtree wants): a strong pixel is an edge whatever it is connected to, so only the
// WEAK candidates enter the component machinery -- components of weak pixels, "on" when some member has a strong
// 8-neighbour -- and the count is (strong pixels) + (sizes of the on components).  On dense maps (noise inputs: nearly
// every candidate is strong) that removes almost all labelling work; the standalone edge map keeps every candidate.
#include "common.h"
#include "devutil.h"

#include <cstdlib>

namespace uwie {

namespace {

struct CannyBufs {
    uint8_t *cmap;     // 0 weak, 1 none, 2 strong
    int32_t *label;    // union-find parent (index inside the frame), candidates only
    uint8_t *flag;     // root owns a strong pixel, candidates only
    uint32_t *cand;    // candidate lists: 512 entries per (tile, wavefront), pixel index inside the frame
    uint32_t *ncand;   // their lengths, [tile][4]
    uint2 *roots;      // tile-local component roots, same slots: {pixel index of the root, size | strong pixels << 16}
    uint32_t *nroot;   // their counts, [tile][4]
    uint32_t *nborder; // candidates on the tile's left / right column or bottom row: they lead their list, [tile][4]
    uint32_t *strong;  // [region] some pixel's gradient magnitude exceeds the high threshold (pre-pass), or nullptr
    uint32_t *count;   // [region] edge counts, or nullptr
    uint32_t *nstrong; // (WEAKONLY) strong pixels per tile, [tile]: k_canny_emit adds them to the counts
    int tiles;         // tiles per region in this launch
    uint32_t *status;  // the context's device status word (uwie_device_status) or nullptr: walkers report a bad label there
    int inject;        // tests only (tuning canny_fault_inject): roots do not get their own label -- the round-3 defect
};

typedef short v2s __attribute__((ext_vector_type(2)));
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;

constexpr int kCT_H = 32, kCT_W = 64;      // output tile
constexpr int kSG_H = kCT_H + 4, kSG_W = 72;  // gray tile: 2-pixel halo, rows padded to a multiple of 8 bytes

__device__ __forceinline__ v2s as_v2s(uint32_t u)
{
    union { uint32_t u; v2s v; } c;
    c.u = u;
    return c.v;
}
// bytes i and j of the 64-bit value hi:lo, zero-extended into the two halves of a dword
template <int I, int J>
__device__ __forceinline__ v2s byte_pair(uint32_t lo, uint32_t hi)
{
    return as_v2s(__builtin_amdgcn_perm(hi, lo, 0x0c000c00u | (uint32_t)I | ((uint32_t)J << 16)));
}

// ---- pre-pass: does the region hold a pixel whose gradient magnitude exceeds the high threshold? ---------------------
// The tile kernel above spends its time waiting (a short-lived workgroup per 32 x 64 tile: one load round trip, one
// barrier).  This one streams: a wavefront owns a strip of 256 columns (4 per lane) and walks a band of rows with the
// separable row terms of three rows in registers; each row is one (unaligned) 8-byte load per lane -- the lane's four
// pixels and their two horizontal neighbours -- eight rows in flight.  Sobel on packed 16-bit pairs as above.
// At the region's left / right border the window is assembled from the row's clamped 8-byte load with per-lane byte
// shifts (BORDER_REPLICATE on the quadrant); lanes beyond the region compute garbage that is masked at the end.
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
constexpr int kPreRows = 32;   // rows per band
constexpr int kPreCols = 256;  // columns per strip

struct PreRow { v2s hd0, hd1, hs0, hs1; };

template <bool EDGE>
__device__ __forceinline__ PreRow pre_row(uint64_t w, const uint32_t (&sh)[6])
{
    uint32_t lo, hi;  // bytes 0..3 and 4..5 of the window x-1 .. x+4
    if constexpr (EDGE) {
        uint32_t b[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) b[k] = (uint32_t)(w >> sh[k]) & 0xffu;
        lo = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
        hi = b[4] | (b[5] << 8);
    } else {
        lo = (uint32_t)w;
        hi = (uint32_t)(w >> 32);
    }
    const v2s A0 = byte_pair<0, 1>(lo, hi), A1 = byte_pair<2, 3>(lo, hi), A2 = byte_pair<4, 5>(lo, hi);
    const v2s B0 = byte_pair<1, 2>(lo, hi), B1 = byte_pair<3, 4>(lo, hi);
    PreRow r;
    r.hd0 = A1 - A0; r.hd1 = A2 - A1;
    r.hs0 = A0 + B0 + B0 + A1; r.hs1 = A1 + B1 + B1 + A2;
    return r;
}

template <bool EDGE>
__device__ void pre_band(const uint8_t *__restrict__ g, int W, const Region &r, int x, int y0, int y1, uint32_t &top0, uint32_t &top1)
{
    // the 8 bytes loaded per row start at column xb of the region; window byte k sits sh[k] bits into them
    const int xb = EDGE ? min(max(x - 1, 0), r.cols - 8) : x - 1;
    uint32_t sh[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) sh[k] = EDGE ? 8u * (uint32_t)(min(max(x - 1 + k, 0), r.cols - 1) - xb) : 0u;
    const uint8_t *base = g + (size_t)r.y0 * W + r.x0 + xb;
    auto load = [&](int y) -> uint64_t {
        return *reinterpret_cast<const u64_unaligned *>(base + (size_t)min(max(y, 0), r.rows - 1) * W);
    };
    PreRow a = pre_row<EDGE>(load(y0 - 1), sh), b = pre_row<EDGE>(load(y0), sh);
    v2s m0 = {0, 0}, m1 = {0, 0};
    for (int yb = y0; yb < y1; yb += 8) {
        uint64_t w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = load(min(yb + i, y1 - 1) + 1);  // (past the band: the last row again, harmless)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const PreRow c = pre_row<EDGE>(w[i], sh);
            if (yb + i < y1) {
                const v2s dx0 = a.hd0 + b.hd0 + b.hd0 + c.hd0, dx1 = a.hd1 + b.hd1 + b.hd1 + c.hd1;
                const v2s dy0 = c.hs0 - a.hs0, dy1 = c.hs1 - a.hs1;
                m0 = __builtin_elementwise_max(m0, __builtin_elementwise_abs(dx0) + __builtin_elementwise_abs(dy0));
                m1 = __builtin_elementwise_max(m1, __builtin_elementwise_abs(dx1) + __builtin_elementwise_abs(dy1));
                a = b;
                b = c;
            }
        }
    }
    union { v2s v; uint32_t u; } c0, c1;
    c0.v = m0; c1.v = m1;
    top0 = c0.u; top1 = c1.u;
}

__global__ void __launch_bounds__(256) k_canny_strong(const uint8_t *__restrict__ gray, const Region *__restrict__ regs, int H,
                                                      int W, int strips, int high, uint32_t *__restrict__ strong)
{
    const Region r = regs[blockIdx.y];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (rows are scalar arithmetic)
    const int strip = blockIdx.x % strips, band = (blockIdx.x / strips) * 4 + wv;
    const int x0 = strip * kPreCols, y0 = band * kPreRows;
    if (x0 >= r.cols || y0 >= r.rows) return;
    if (__hip_atomic_load(strong + blockIdx.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // already known
    if (r.cols < 8) {  // narrower than one load: leave it to the full pass
        if (lane == 0) strong[blockIdx.y] = 1;
        return;
    }
    const int y1 = min(y0 + kPreRows, r.rows), x = x0 + 4 * lane;
    const uint8_t *g = gray + (size_t)r.img * H * W;
    uint32_t t0, t1;
    if (x0 >= 1 && x0 + kPreCols + 3 <= r.cols) pre_band<false>  // (the 8-byte loads reach 2 columns past the window)
       (g, W, r, x, y0, y1, t0, t1);
    else pre_band<true>(g, W, r, x, y0, y1, t0, t1);
    // pixel j of the lane is column x + j; columns beyond the region do not count
    const int m[4] = {(int)(t0 & 0xffffu), (int)(t0 >> 16), (int)(t1 & 0xffffu), (int)(t1 >> 16)};
    bool hot = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) hot = hot || (x + j < r.cols && m[j] > high);
    if (__ballot(hot) && lane == 0) strong[blockIdx.y] = 1;
}

// ---- level 0 of the quadtree (round 4): gray plane + pre-pass in ONE sweep over the RGB frame -------------------------
// The four level-0 quadrants are the whole frame, so the pass that answers "any strong pixel?" for them visits every pixel
// once: it computes the gray bytes itself ((x * 255).astype(u8) -> RGB2GRAY of the colour-corrected value, the integer
// identities of k_q_hist / tests/test_cabi.py), stores them, and runs the separable Sobel on them -- the frame's bytes are
// read once here instead of once by the gray-writing histogram pass and the gray plane once more by k_canny_strong.
// A wavefront owns a strip of 64 four-pixel groups of which the first and the last only supply the horizontal neighbours
// (248 output columns: strips overlap by two lanes); a lane gets its left / right gray byte from the neighbouring lane's
// word with one DPP move each.  Requires quadrant widths that are multiples of four (W % 8 == 0).
// The gray value in float32: R c_r + G c_g + B c_b + 1/2 with c = coefficient / 2^shift is exact (integers below 2^24
// scaled by a power of two), so its truncation is gray_fixed(); 17 u / 20 of an attenuated byte is trunc(0.85f u + 0.025)
// (the fractional part of 17 u / 20 is a multiple of 0.05; the float32 error is below 3e-5).
constexpr int kGsCols = 248;

template <int ATT>
__device__ void gs_band(const uint8_t *__restrict__ img, uint8_t *__restrict__ gray, int W, const Region &r, int lane, int x,
                        int y0, int y1, float cr, float cg, float cb, const uint32_t *known, uint32_t &top0, uint32_t &top1)
{
    const int xl = min(max(x, 0), r.cols - 4);
    const bool owner = lane >= 1 && lane <= 62 && x < r.cols;  // (x >= 0 for every lane but lane 0)
    const bool first = x == 0, last = x + 4 == r.cols;
    const uint8_t *src = img + ((size_t)r.y0 * W + r.x0 + xl) * 3;
    uint8_t *dst = gray + (size_t)r.y0 * W + r.x0 + x;
    struct Raw { uint32_t d[3]; };
    auto load = [&](int y) {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(src + (size_t)min(max(y, 0), r.rows - 1) * W * 3);
        Raw v;
        v.d[0] = q[0]; v.d[1] = q[1]; v.d[2] = q[2];
        return v;
    };
    const uint32_t no_sh[6] = {};
    // the row's gray word -> (stored) -> the 6-byte window x-1 .. x+4 -> its separable row terms
    auto row = [&](const Raw &v, int y, bool keep) {
        const uint32_t g4 = gray4_f32<ATT>(v.d, cr, cg, cb);
        if (keep && owner) *reinterpret_cast<uint32_t *>(dst + (size_t)y * W) = g4;
        uint32_t lf = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g4, 0x138, 0xf, 0xf, false) >> 24;  // wave_shr:1: lane - 1
        uint32_t rt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g4, 0x130, 0xf, 0xf, false) & 255u;  // wave_shl:1: lane + 1
        if (first) lf = g4 & 255u;  // BORDER_REPLICATE on the quadrant
        if (last) rt = g4 >> 24;
        return pre_row<false>((uint64_t)lf | ((uint64_t)g4 << 8) | ((uint64_t)rt << 40), no_sh);
    };
    PreRow a = row(load(y0 - 1), y0 - 1, false), b = row(load(y0), y0, true);
    v2s m0 = {0, 0}, m1 = {0, 0};
    for (int yb = y0; yb < y1; yb += 8) {
        Raw w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = load(min(yb + i, y1 - 1) + 1);  // (past the band: the last row again, harmless)
        // a region already known to hold a strong pixel only needs its gray bytes (wavefront-uniform)
        const bool sobel = !__hip_atomic_load(known, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (yb + i < y1) {
                const int yn = yb + i + 1;  // the row below output row yb + i
                const PreRow c = row(w[i], yn, yn < y1);
                if (sobel) {
                    const v2s dx0 = a.hd0 + b.hd0 + b.hd0 + c.hd0, dx1 = a.hd1 + b.hd1 + b.hd1 + c.hd1;
                    const v2s dy0 = c.hs0 - a.hs0, dy1 = c.hs1 - a.hs1;
                    m0 = __builtin_elementwise_max(m0, __builtin_elementwise_abs(dx0) + __builtin_elementwise_abs(dy0));
                    m1 = __builtin_elementwise_max(m1, __builtin_elementwise_abs(dx1) + __builtin_elementwise_abs(dy1));
                }
                a = b;
                b = c;
            }
        }
    }
    union { v2s v; uint32_t u; } c0, c1;
    c0.v = m0; c1.v = m1;
    top0 = owner ? c0.u : 0u;
    top1 = owner ? c1.u : 0u;
}

__global__ void __launch_bounds__(256) k_gray_strong(const uint8_t *__restrict__ in, const int32_t *__restrict__ kind,
                                                     const Region *__restrict__ regs, int H, int W, int strips, int high, float cr,
                                                     float cg, float cb, uint8_t *__restrict__ gray_out, uint32_t *__restrict__ strong)
{
    const Region r = regs[blockIdx.y];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (rows are scalar arithmetic)
    const int strip = blockIdx.x % strips, band = (blockIdx.x / strips) * 4 + wv;
    const int x0 = strip * kGsCols - 4, y0 = band * kPreRows;
    if (x0 + 4 >= r.cols || y0 >= r.rows) return;
    const int y1 = min(y0 + kPreRows, r.rows), x = x0 + 4 * lane;
    const uint8_t *img = in + (size_t)r.img * H * W * 3;
    uint8_t *g = gray_out + (size_t)r.img * H * W;
    const int knd = kind ? kind[r.img] : 0;
    uint32_t t0, t1;
    if (knd == 1) gs_band<1>(img, g, W, r, lane, x, y0, y1, cr, cg, cb, strong + blockIdx.y, t0, t1);
    else if (knd == 2) gs_band<2>(img, g, W, r, lane, x, y0, y1, cr, cg, cb, strong + blockIdx.y, t0, t1);
    else gs_band<0>(img, g, W, r, lane, x, y0, y1, cr, cg, cb, strong + blockIdx.y, t0, t1);
    const int m[4] = {(int)(t0 & 0xffffu), (int)(t0 >> 16), (int)(t1 & 0xffffu), (int)(t1 >> 16)};
    const bool hot = m[0] > high || m[1] > high || m[2] > high || m[3] > high;
    if (__ballot(hot) && lane == 0) strong[blockIdx.y] = 1;
}

// ---- levels >= 1 of the quadtree (round 4): the quadrants' byte histograms AND the pre-pass in one sweep ---------------
// k_q_hist (k_airlight.hip) and k_canny_strong stream over the same four quadrants of the chosen block, one after the
// other: the first is bound by its LDS atomics, the second by its packed arithmetic.  Here one wavefront does both for its
// strip and band -- per row three dwords of RGB for the histogram (the lane-column words of k_chunk_hist: the bank of an
// atomic is the lane's column) next to the 8 gray bytes of the Sobel window -- so the two instruction streams share the
// wavefront's issue slots and a level costs one launch less.  Requires quadrant widths and origins that are multiples of
// four (the host checks (W >> level) % 8 == 0).  A block = four bands of one strip; its histogram is folded after 16 rows
// (10-bit fields: a column takes 2 lanes x 4 wavefronts x 4 pixels per row) and at the end, then added to the region's.
template <bool EDGE>
__device__ void hs_body(const uint8_t *__restrict__ g, const uint8_t *__restrict__ rgb, int W, const Region &r, int x0, int y0,
                        int lane, int tid, int high, uint32_t *__restrict__ strong_flag, uint32_t *__restrict__ hist_reg, uint32_t *h,
                        uint32_t *cnt)
{
    const int x = x0 + 4 * lane;
    const bool rows_on = y0 < r.rows, counted = x + 4 <= r.cols;
    const int y1 = min(y0 + kPreRows, r.rows);
    const bool narrow = r.cols < 8;  // narrower than the window load: left to the full pass (as k_canny_strong)
    // the 8 gray bytes loaded per row start at column xb of the region; window byte k sits sh[k] bits into them
    const int xb = EDGE ? min(max(x - 1, 0), max(r.cols - 8, 0)) : x - 1;
    uint32_t sh[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) sh[k] = EDGE ? 8u * (uint32_t)(min(max(x - 1 + k, 0), r.cols - 1) - xb) : 0u;
    const uint8_t *gbase = g + (size_t)r.y0 * W + r.x0 + xb;
    const uint8_t *cbase = rgb + ((size_t)r.y0 * W + r.x0 + min(x, r.cols - 4)) * 3;
    auto gload = [&](int y) -> uint64_t {
        return *reinterpret_cast<const u64_unaligned *>(gbase + (size_t)min(max(y, 0), r.rows - 1) * W);
    };
    const uint32_t colb = (uint32_t)(tid & 31) * 4u;
    char *hb = reinterpret_cast<char *>(h);
    auto bump = [&](uint32_t moved, uint32_t inc) { atomicAdd(reinterpret_cast<uint32_t *>(hb + ((moved & 0x7f80u) | colb)), inc); };
    constexpr uint32_t kR = 1u, kG = 1u << 10, kB = 1u << 20;
    auto fold = [&]() {  // all threads; thread v folds the 32 columns of value v into the 32-bit totals and clears them
        __syncthreads();
        uint32_t cr = 0, cg = 0, cb = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint4 *wq = reinterpret_cast<uint4 *>(&h[tid * 32 + 4 * ((tid + k) & 7)]);
            const uint4 w = *wq;
            *wq = make_uint4(0, 0, 0, 0);
            cr += (w.x & 1023u) + (w.y & 1023u) + (w.z & 1023u) + (w.w & 1023u);
            cg += ((w.x >> 10) & 1023u) + ((w.y >> 10) & 1023u) + ((w.z >> 10) & 1023u) + ((w.w >> 10) & 1023u);
            cb += (w.x >> 20) + (w.y >> 20) + (w.z >> 20) + (w.w >> 20);
        }
        cnt[tid] += cr; cnt[256 + tid] += cg; cnt[512 + tid] += cb;
        __syncthreads();
    };
    bool sobel = rows_on && !narrow && !__hip_atomic_load(strong_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (wavefront-uniform)
    PreRow a{}, b{};
    if (sobel) {
        a = pre_row<EDGE>(gload(y0 - 1), sh);
        b = pre_row<EDGE>(gload(y0), sh);
    }
    v2s m0 = {0, 0}, m1 = {0, 0};
#pragma unroll 1
    for (int blk = 0; blk < kPreRows / 8; ++blk) {  // the same trip count for the four wavefronts: fold() is a barrier
        const int yb = y0 + 8 * blk;
        if (rows_on && yb < y1) {
            uint64_t w[8];
            uint32_t c[8][3];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int yy = min(yb + i, y1 - 1);  // (past the band: the last row again, harmless)
                w[i] = gload(yy + 1);
                const uint32_t *q = reinterpret_cast<const uint32_t *>(cbase + (size_t)yy * W * 3);
                c[i][0] = q[0]; c[i][1] = q[1]; c[i][2] = q[2];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (yb + i < y1) {
                    if (counted) {
                        const uint32_t c0 = c[i][0], c1 = c[i][1], c2 = c[i][2];
                        // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
                        bump(c0 << 7, kR); bump(c0 >> 1, kG); bump(c0 >> 9, kB);
                        bump(c0 >> 17, kR); bump(c1 << 7, kG); bump(c1 >> 1, kB);
                        bump(c1 >> 9, kR); bump(c1 >> 17, kG); bump(c2 << 7, kB);
                        bump(c2 >> 1, kR); bump(c2 >> 9, kG); bump(c2 >> 17, kB);
                    }
                    if (sobel) {
                        const PreRow cc = pre_row<EDGE>(w[i], sh);
                        const v2s dx0 = a.hd0 + b.hd0 + b.hd0 + cc.hd0, dx1 = a.hd1 + b.hd1 + b.hd1 + cc.hd1;
                        const v2s dy0 = cc.hs0 - a.hs0, dy1 = cc.hs1 - a.hs1;
                        m0 = __builtin_elementwise_max(m0, __builtin_elementwise_abs(dx0) + __builtin_elementwise_abs(dy0));
                        m1 = __builtin_elementwise_max(m1, __builtin_elementwise_abs(dx1) + __builtin_elementwise_abs(dy1));
                        a = b;
                        b = cc;
                    }
                }
            }
        }
        if (blk == 1) fold();
    }
    fold();
    for (int i = tid; i < 768; i += 256)
        if (cnt[i]) atomicAdd(&hist_reg[i], cnt[i]);
    if (rows_on && narrow) {
        if (lane == 0) *strong_flag = 1;
        return;
    }
    union { v2s v; uint32_t u; } u0, u1;
    u0.v = m0; u1.v = m1;
    const int m[4] = {(int)(u0.u & 0xffffu), (int)(u0.u >> 16), (int)(u1.u & 0xffffu), (int)(u1.u >> 16)};
    bool hot = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) hot = hot || (x + j < r.cols && m[j] > high);
    if (__ballot(hot) && lane == 0) *strong_flag = 1;
}

__global__ void __launch_bounds__(256) k_hist_strong(const uint8_t *__restrict__ in, const uint8_t *__restrict__ gray,
                                                     const Region *__restrict__ regs, int H, int W, int strips, int high,
                                                     uint32_t *__restrict__ strong, uint32_t *__restrict__ hist)
{
    __shared__ __attribute__((aligned(16))) uint32_t h[256 * 32];
    __shared__ uint32_t cnt[768];
    const Region r = regs[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int strip = blockIdx.x % strips, band = (blockIdx.x / strips) * 4 + wv;
    const int x0 = strip * kPreCols, y0 = band * kPreRows;
    if (r.rows <= 0 || r.cols < 4 || x0 >= r.cols || (int)(blockIdx.x / strips) * 4 * kPreRows >= r.rows) return;  // (block-uniform)
    for (int i = tid; i < 256 * 32 / 4; i += 256) reinterpret_cast<uint4 *>(h)[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < 768; i += 256) cnt[i] = 0;
    // (the first fold's barrier orders the clears before the first atomics of the OTHER wavefronts; this one for its own)
    __syncthreads();
    const uint8_t *g = gray + (size_t)r.img * H * W, *rgb = in + (size_t)r.img * H * W * 3;
    const bool edge = !(x0 >= 1 && x0 + kPreCols + 3 <= r.cols);  // (block-uniform: the four wavefronts share the strip)
    if (edge) hs_body<true>(g, rgb, W, r, x0, y0, lane, tid, high, strong + blockIdx.y, hist + (size_t)blockIdx.y * 768, h, cnt);
    else hs_body<false>(g, rgb, W, r, x0, y0, lane, tid, high, strong + blockIdx.y, hist + (size_t)blockIdx.y * 768, h, cnt);
}

// lock-free union-find on tile-local indices in LDS (links point to the smaller index)
__device__ __forceinline__ uint32_t lds_ld(const uint32_t *L, int i)
{
    return __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ int lds_find(uint32_t *L, int i)
{
    for (;;) {
        const int p = (int)lds_ld(L, i);
        if (p == i) return i;
        const int gp = (int)lds_ld(L, p);
        if (gp == p) return p;
        atomicMin(L + i, (uint32_t)gp);
        i = gp;
    }
}
__device__ void lds_union(uint32_t *L, int a, int b)
{
    for (;;) {
        a = lds_find(L, a);
        b = lds_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const uint32_t old = atomicCAS(L + a, (uint32_t)a, (uint32_t)b);
        if ((int)old == a) return;
        a = (int)old;
    }
}

template <bool PRE, bool WEAKONLY = false>
__global__ void __launch_bounds__(256) k_canny_gradnms(const uint8_t *__restrict__ gray, const Region *__restrict__ regs,
                                                       int H, int W, int tiles_x, int low, int high, CannyBufs bufs)
{
    __shared__ __attribute__((aligned(8))) uint8_t sg[kSG_H][kSG_W];
    __shared__ uint32_t s_lab[kCT_H * kCT_W], s_info[kCT_H * kCT_W];
    __shared__ uint8_t s_keep[256], s_str[WEAKONLY ? 256 : 1];
    __shared__ uint32_t s_total;  // candidates in the tile
    __shared__ uint32_t s_strong_total;  // (WEAKONLY) strong pixels in the tile
    const Region r = regs[blockIdx.y];
    const int ty0 = (blockIdx.x / tiles_x) * kCT_H, tx0 = (blockIdx.x % tiles_x) * kCT_W;  // tile origin inside the region
    const int tid = threadIdx.x;
    if (tid == 0) s_total = s_strong_total = 0;  // (the barrier after the tile fill orders this before the atomics)
    const size_t sub = ((size_t)blockIdx.y * bufs.tiles + blockIdx.x) * 4 + (tid >> 6);  // this wavefront's list
    if (ty0 >= r.rows || tx0 >= r.cols || (!PRE && bufs.strong && !bufs.strong[blockIdx.y])) {
        if (!PRE && (tid & 63) == 0) bufs.ncand[sub] = bufs.nroot[sub] = bufs.nborder[sub] = 0;
        if (!PRE && WEAKONLY && tid == 0) bufs.nstrong[(size_t)blockIdx.y * bufs.tiles + blockIdx.x] = 0;
        return;
    }
    const size_t base = (size_t)r.img * H * W;
    const uint8_t *g = gray + base;
    if (tx0 >= 2 && tx0 + kCT_W + 2 <= r.cols) {  // no column clamping: (unaligned) dword loads, 17 per tile row
        for (int i = tid; i < kSG_H * 17; i += 256) {
            const int ly = i / 17, lq = i - ly * 17;
            const int ry = min(max(ty0 + ly - 2, 0), r.rows - 1);
            const uint8_t *src = g + (size_t)(r.y0 + ry) * W + r.x0 + tx0 - 2 + 4 * lq;
            *reinterpret_cast<uint32_t *>(&sg[ly][4 * lq]) = *reinterpret_cast<const u32_unaligned *>(src);
        }
    } else {
        for (int i = tid; i < kSG_H * (kCT_W + 4); i += 256) {
            const int ly = i / (kCT_W + 4), lx = i % (kCT_W + 4);
            const int ry = min(max(ty0 + ly - 2, 0), r.rows - 1), rx = min(max(tx0 + lx - 2, 0), r.cols - 1);
            sg[ly][lx] = g[(size_t)(r.y0 + ry) * W + r.x0 + rx];
        }
    }
    __syncthreads();
    const int cg = tid & 15, rp = tid >> 4;            // column group (4 pixels), row pair
    const int ry0 = ty0 + 2 * rp, rx0 = tx0 + 4 * cg;  // region coordinates of this thread's first pixel
    const bool inside = ry0 < r.rows && rx0 < r.cols;  // no early exit: the wavefront scan below needs every lane
    // separable Sobel on column pairs: pair p holds magnitude columns rx0-1+2p, rx0+2p
    v2s hd[6][3], vs[6][3];
#pragma unroll
    for (int gr = 0; gr < 6; ++gr) {
        const uint32_t *row = reinterpret_cast<const uint32_t *>(&sg[2 * rp + gr][4 * cg]);
        const uint32_t w0 = row[0], w1 = row[1];
        const v2s A0 = byte_pair<0, 1>(w0, w1), A1 = byte_pair<2, 3>(w0, w1), A2 = byte_pair<4, 5>(w0, w1),
                  A3 = byte_pair<6, 7>(w0, w1);
        const v2s B0 = byte_pair<1, 2>(w0, w1), B1 = byte_pair<3, 4>(w0, w1), B2 = byte_pair<5, 6>(w0, w1);
        hd[gr][0] = A1 - A0; hd[gr][1] = A2 - A1; hd[gr][2] = A3 - A2;
        vs[gr][0] = A0 + B0 + B0 + A1; vs[gr][1] = A1 + B1 + B1 + A2; vs[gr][2] = A2 + B2 + B2 + A3;
    }
    // magnitudes outside the region are 0: only threads on the region border need the test
    const bool border = ry0 < 1 || ry0 + 2 >= r.rows || rx0 < 1 || rx0 + 4 >= r.cols;
    int mag[4][6], cdx[2][4], cdy[2][4];
#pragma unroll
    for (int mr = 0; mr < 4; ++mr) {
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) {
            const v2s dx = hd[mr][pp] + hd[mr + 1][pp] + hd[mr + 1][pp] + hd[mr + 2][pp];
            const v2s dy = vs[mr + 2][pp] - vs[mr][pp];
            const v2s m = __builtin_elementwise_abs(dx) + __builtin_elementwise_abs(dy);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int mc = 2 * pp + h;
                mag[mr][mc] = (int)(h ? m.y : m.x);
                if ((mr == 1 || mr == 2) && mc >= 1 && mc <= 4) {
                    cdx[mr - 1][mc - 1] = h ? dx.y : dx.x;
                    cdy[mr - 1][mc - 1] = h ? dy.y : dy.x;
                }
            }
        }
    }
    if (border) {
#pragma unroll
        for (int mr = 0; mr < 4; ++mr) {
            const int my = ry0 - 1 + mr;
            const bool rowin = my >= 0 && my < r.rows;
#pragma unroll
            for (int mc = 0; mc < 6; ++mc) {
                const int mx = rx0 - 1 + mc;
                if (!(rowin && mx >= 0 && mx < r.cols)) mag[mr][mc] = 0;
            }
        }
    }
    if constexpr (PRE) {
        int top = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) top = max(top, mag[i + 1][j + 1]);
        if (__ballot(top > high) && (tid & 63) == 0) bufs.strong[blockIdx.y] = 1;
        return;
    }
    // non-maximum suppression, branch-free: the two neighbours along the gradient direction are selected, not branched on
    uint32_t cls[2] = {0x01010101u, 0x01010101u}, keepmask = 0, strongmask = 0;
    // a wavefront whose 8 x 64 pixels all stay at or below the low threshold (smooth water) has nothing to suppress
    int hot = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) hot = max(hot, mag[i + 1][j + 1]);
    if (__ballot(hot > low)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mag[i + 1][j + 1];
            const int dx = cdx[i][j], dy = cdy[i][j];
            const int ax = abs(dx), ay = abs(dy) << 15;
            const int tg22x = ax * 13573;  // (int)(tan(22.5deg) * 2^15 + 0.5)
            const bool isH = ay < tg22x, isV = ay > tg22x + (ax << 16), neg = (dx ^ dy) < 0;
            const int d1 = neg ? mag[i][j + 2] : mag[i][j], d2 = neg ? mag[i + 2][j] : mag[i + 2][j + 2];
            const int n1 = isH ? mag[i + 1][j] : isV ? mag[i][j + 1] : d1;
            const int n2 = isH ? mag[i + 1][j + 2] : isV ? mag[i + 2][j + 1] : d2;
            const bool diag = !isH && !isV;
            // (the three tests written as comparisons combined by scalar logic compiled to more selects, not fewer: 3.21 vs 3.08 ms)
            const bool keep = inside && ry0 + i < r.rows && rx0 + j < r.cols && m > low && m > n1 &&
                              (m > n2 || (!diag && m == n2));
            if (keep) {
                keepmask |= 1u << (4 * i + j);
                if (m > high) strongmask |= 1u << (4 * i + j);
                cls[i] = (cls[i] & ~(0xffu << (8 * j))) | ((m > high ? 2u : 0u) << (8 * j));
            }
        }
    }
    }
    if constexpr (WEAKONLY) {
        // strong pixels are edges: counted here, once per wavefront; only the weak candidates are labelled below
        const uint32_t ns = wave_sum_u32(__popc(strongmask));
        if (ns && (tid & 63) == 0) atomicAdd(&s_strong_total, ns);  // (ordered by the barriers below; added to the region's count at the end)
        keepmask &= ~strongmask;
    }
    const uint32_t ncand = __popc(keepmask);
    // map bytes: one (unaligned) dword per row when the 4 pixels exist, else byte by byte
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (!inside || ry0 + i >= r.rows) continue;
        uint8_t *dst = bufs.cmap + base + (size_t)(r.y0 + ry0 + i) * W + r.x0 + rx0;
        if (rx0 + 3 < r.cols) *reinterpret_cast<u32_unaligned *>(dst) = cls[i];
        else
            for (int j = 0; j < r.cols - rx0; ++j) dst[j] = (uint8_t)(cls[i] >> (8 * j));
    }
    // ---- tile-local components in LDS: on dense maps (noise inputs) nearly every pixel is a candidate and a union-find
    // over global memory alone is contention-bound; merging inside the tile first leaves only the tile-border links to it
    // Links follow the decision tree of sequential two-pass labelling (Wu, Otoo, Suzuki 2009), applied to all pixels at
    // once: with a = up-left, b = up, c = up-right, d = left (inside the tile), a candidate starts as a child of b, else
    // c, else a, else d (a plain store: all four have smaller indices), and only "c without b, but a or d" needs a real
    // union (c with a, else c with d): every other adjacency is implied by the neighbours' own links.  On a dense map
    // that is no union at all, where linking every pixel to its four forward neighbours took eight finds per pixel.
    if (ncand) atomicAdd(&s_total, ncand);
    s_keep[tid] = (uint8_t)keepmask;  // 4x2 candidate bits of this thread's block
    if constexpr (WEAKONLY) s_str[tid] = (uint8_t)strongmask;
    __syncthreads();
    auto is_cand = [&](int ly, int lx) -> bool {  // tile coordinates, inside the tile
        return (s_keep[(ly >> 1) * 16 + (lx >> 2)] >> (((ly & 1) << 2) | (lx & 3))) & 1;
    };
    // WEAKONLY: bit b of onmask = the weak candidate b of this thread has a strong 8-neighbour inside the tile
    uint32_t onmask = 0;
    if constexpr (WEAKONLY) {
        auto is_strong = [&](int ly, int lx) -> bool {
            if ((unsigned)ly >= (unsigned)kCT_H || (unsigned)lx >= (unsigned)kCT_W) return false;
            return (s_str[(ly >> 1) * 16 + (lx >> 2)] >> (((ly & 1) << 2) | (lx & 3))) & 1;
        };
        for (uint32_t km = keepmask; km; km &= km - 1) {
            const int b = __ffs(km) - 1, ly = 2 * rp + (b >> 2), lx = 4 * cg + (b & 3);
            const bool on = is_strong(ly - 1, lx - 1) || is_strong(ly - 1, lx) || is_strong(ly - 1, lx + 1) || is_strong(ly, lx - 1) ||
                            is_strong(ly, lx + 1) || is_strong(ly + 1, lx - 1) || is_strong(ly + 1, lx) || is_strong(ly + 1, lx + 1);
            if (on) onmask |= 1u << b;
        }
    }
    // what a part contributes at its root: its size, and whether it is "strong" (owns a strong pixel; WEAKONLY: touches one)
    auto part_info = [&](int b) -> uint32_t {
        const bool st = WEAKONLY ? ((onmask >> b) & 1) != 0 : ((cls[b >> 2] >> (8 * (b & 3))) & 0xffu) == 2u;
        return 1u | (st ? 0x10000u : 0u);
    };
    uint32_t needmask = 0, leftmask = 0;
    for (uint32_t km = keepmask; km; km &= km - 1) {
        const int b = __ffs(km) - 1, ly = 2 * rp + (b >> 2), lx = 4 * cg + (b & 3), li = ly * kCT_W + lx;
        const bool up = ly > 0, lf = lx > 0, rt = lx + 1 < kCT_W;
        const bool nb = up && is_cand(ly - 1, lx), nc = up && rt && is_cand(ly - 1, lx + 1);
        const bool na = up && lf && is_cand(ly - 1, lx - 1), nd = lf && is_cand(ly, lx - 1);
        s_lab[li] = (uint32_t)(nb ? li - kCT_W : nc ? li - kCT_W + 1 : na ? li - kCT_W - 1 : nd ? li - 1 : li);
        s_info[li] = 0;
        if (!nb && nc && (na || nd)) {
            needmask |= 1u << b;
            if (!na) leftmask |= 1u << b;
        }
    }
    __syncthreads();
    for (uint32_t km = needmask; km; km &= km - 1) {
        const int b = __ffs(km) - 1, li = (2 * rp + (b >> 2)) * kCT_W + 4 * cg + (b & 3);
        lds_union(s_lab, li - kCT_W + 1, (leftmask >> b) & 1 ? li - 1 : li - kCT_W - 1);
    }
    __syncthreads();
    // Dense tiles: size and number of strong pixels of every tile-local component are gathered at its root, and only the
    // roots are recorded for the hysteresis walk.  Sparse tiles skip that (one more barrier, an LDS atomic per
    // candidate): every candidate is recorded as a part of size one.
    // WEAKONLY always gathers (its lists are short, and a gathered component that touches no tile border is finished here)
    const bool dense = WEAKONLY || s_total > 256u;  // block-uniform
    // candidates that can have a neighbour in another tile (left / right column, bottom row) lead the list: the
    // cross-tile union walks only them
    // (WEAKONLY: the top row too -- a strong neighbour in the tile above turns a weak component on)
    const uint32_t bordermask = keepmask & ((cg == 0 ? 0x11u : 0u) | (cg == 15 ? 0x88u : 0u) | (rp == 15 ? 0xf0u : 0u) |
                                            (WEAKONLY && rp == 0 ? 0x0fu : 0u));
    uint32_t myroot[8];
    {
        int k = 0;
        for (uint32_t km = keepmask; km; km &= km - 1, ++k) {
            const int b = __ffs(km) - 1, li = (2 * rp + (b >> 2)) * kCT_W + 4 * cg + (b & 3);
            const uint32_t root = (uint32_t)lds_find(s_lab, li);
            myroot[k] = root;
            if constexpr (WEAKONLY)  // size (12 bits: a tile has 2048 pixels) | members that touch a strong pixel | members on the tile border (< 256)
                atomicAdd(&s_info[root], 1u | (((onmask >> b) & 1u) << 12) | (((bordermask >> b) & 1u) << 24));
            else if (dense) atomicAdd(&s_info[root], part_info(b));
        }
    }
    if (dense) __syncthreads();
    // Lists of this (tile, wavefront), fixed slots (no atomics, nothing to wait for): every candidate, with its global
    // label starting at the root of its tile-local component, and the roots with their component's size / strength.
    // Hysteresis then only has to look at roots: a component is an edge component iff some root of it is strong.
    // WEAKONLY: a component without a member on the tile border cannot grow or be switched on from outside -- it is counted
    // here if it is on and forgotten either way: only border-touching components get a root entry, only border candidates a
    // label and a list slot (on hazy noise most weak components are small and interior: the list walkers' work shrinks to
    // the tile seams).
    uint32_t rootmask = 0;
    {
        uint32_t add = 0;
        int k = 0;
        for (uint32_t km = keepmask; km; km &= km - 1, ++k) {
            const int b = __ffs(km) - 1, li = (2 * rp + (b >> 2)) * kCT_W + 4 * cg + (b & 3);
            if (dense && (int)myroot[k] != li) continue;
            if constexpr (WEAKONLY) {
                const uint32_t info = s_info[li];
                if ((info >> 24) == 0) {
                    if ((info >> 12) & 0xfffu) add += info & 0xfffu;
                    continue;
                }
            }
            rootmask |= 1u << b;
        }
        if constexpr (WEAKONLY) {
            const uint32_t tot = wave_sum_u32(add);
            if (tot && (tid & 63) == 0) atomicAdd(&s_strong_total, tot);
            __syncthreads();
            // (a plain store per tile: atomics on the 4 B region counters, even one per tile, made this kernel wait on 256 hot
            // addresses -- 4.8 ms instead of 2.6 on noise frames at 4K x 64)
            if (tid == 0) bufs.nstrong[(size_t)blockIdx.y * bufs.tiles + blockIdx.x] = s_strong_total;
        }
    }
    const uint32_t nr = __popc(rootmask);
    const uint32_t listmask = WEAKONLY ? bordermask : keepmask;  // candidates that get a label and a list slot
    const uint32_t nl = __popc(listmask), nb = __popc(bordermask);
    const uint32_t all3 = wave_incl_scan_u32(nl | (nr << 10) | (nb << 20));  // one scan for the three offsets (<= 512 each)
    const uint32_t incl = all3 & 0x3ffu, rincl = (all3 >> 10) & 0x3ffu, bincl = all3 >> 20;
    const uint32_t total_b = __shfl(bincl, 63);
    if ((tid & 63) == 63) {
        bufs.ncand[sub] = incl;
        bufs.nroot[sub] = rincl;
        bufs.nborder[sub] = bincl;
    }
    {
        uint32_t *dst_b = bufs.cand + sub * 512 + (bincl - nb);
        uint32_t *dst_i = bufs.cand + sub * 512 + total_b + ((incl - bincl) - (nl - nb));
        int k = 0;
        for (uint32_t km = keepmask; km; km &= km - 1, ++k) {
            const int b = __ffs(km) - 1;
            if (!((listmask >> b) & 1)) continue;
            const int root = (int)myroot[k];
            const int p = (r.y0 + ry0 + (b >> 2)) * W + r.x0 + rx0 + (b & 3);
            const int proot = (r.y0 + ty0 + root / kCT_W) * W + r.x0 + tx0 + root % kCT_W;
            bufs.label[base + p] = proot;
            if ((bordermask >> b) & 1) *dst_b++ = (uint32_t)p;
            else *dst_i++ = (uint32_t)p;
        }
    }
    if (nr) {
        uint2 *dst = bufs.roots + sub * 512 + (rincl - nr);
        for (uint32_t km = rootmask; km; km &= km - 1) {
            const int b = __ffs(km) - 1, li = (2 * rp + (b >> 2)) * kCT_W + 4 * cg + (b & 3);
            const int p = (r.y0 + ry0 + (b >> 2)) * W + r.x0 + rx0 + (b & 3);
            bufs.flag[base + p] = 0;
            // The root itself may be an interior pixel, which gets no list slot and therefore no label above: the links of its
            // border members end here, so it needs its own.  INVARIANT: every label that can be reached
            // from a list entry of this launch is written in this launch -- label[] is never cleared, so anything else found
            // there is a stale index of an earlier level or frame size.  (Round 3's two memory access faults, on the hazy and
            // uniform 4K x 64 benches between 8c71da7 and 338471a, were this store missing: k_canny_mark / k_canny_emit walked
            // from a border member to the root's pixel and on through whatever the previous level had left there.)
            if constexpr (WEAKONLY) {
                if (!bufs.inject) bufs.label[base + p] = p;
            }
            uint32_t info = dense ? s_info[li] : part_info(b);
            if constexpr (WEAKONLY) info = (info & 0xfffu) | (((info >> 12) & 0xfffu) ? 0x10000u : 0u);  // {size, on}
            *dst++ = make_uint2((uint32_t)p, info);
        }
    }
}

// Parent of pixel i (i is a valid index of the frame).  Links only ever point to a SMALLER index or to the pixel itself, so a
// parent above i -- or a negative one, which is above every index as an unsigned number -- cannot have been written by this
// launch: it is reported in the context's status word (uwie_device_status -> UWIE_E_DEVICE) and i is taken as a root.  With
// this every index a walker dereferences is inside the frame and every walk is strictly decreasing, whatever the label
// plane holds: a violated invariant is an error code, never a memory access fault or a hang.
__device__ __forceinline__ int ld_label(const int32_t *L, int i, uint32_t *status)
{
    const int p = __hip_atomic_load(L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)p > (uint32_t)i) {
        if (status) atomicOr(status, (uint32_t)UWIE_STATUS_CANNY_LABEL);
        return i;
    }
    return p;
}

// Root of i with path halving.  Parent links only ever decrease, so a stale read is still an ancestor.
__device__ int uf_find(int32_t *L, int i, uint32_t *status)
{
    for (;;) {
        const int p = ld_label(L, i, status);
        if (p == i) return i;
        const int gp = ld_label(L, p, status);
        if (gp == p) return p;
        atomicMin(L + i, gp);
        i = gp;
    }
}

__device__ void uf_union(int32_t *L, int a, int b, uint32_t *status)
{
    for (;;) {
        a = uf_find(L, a, status);
        b = uf_find(L, b, status);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicCAS(L + a, a, b);  // link the larger root under the smaller one
        if (old == a) return;
        if ((uint32_t)old > (uint32_t)a) {  // not a label of this launch (see ld_label): nothing to continue from
            if (status) atomicOr(status, (uint32_t)UWIE_STATUS_CANNY_LABEL);
            return;
        }
        a = old;  // somebody re-parented `a` first: continue from its new parent
    }
}

// List walkers: one wavefront takes 16 tiles (64 lists) of region blockIdx.y, loads the 64 lengths with one load and
// spreads ALL their entries over its lanes (entry idx -> list by a binary search over the prefix sums, via shuffles),
// so lanes stay busy however unevenly the candidates are spread.
constexpr int kWalkTiles = 16;

template <class F>
__device__ __forceinline__ void for_list(const uint32_t *__restrict__ counts, int tiles, F f)
{
    const int lane = threadIdx.x & 63;
    const int tile0 = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * kWalkTiles;
    if (tile0 >= tiles) return;  // wavefront-uniform
    const size_t sub0 = ((size_t)blockIdx.y * tiles + tile0) * 4;
    const int nsub = min(64, (tiles - tile0) * 4);
    const uint32_t cnt = lane < nsub ? counts[sub0 + lane] : 0;
    const uint32_t incl = wave_incl_scan_u32(cnt), excl = incl - cnt, total = __shfl(incl, 63);
    for (uint32_t b0 = 0; b0 < total; b0 += 64) {
        const uint32_t idx = b0 + lane;
        int sl = 0;  // largest list index whose exclusive prefix is <= idx
#pragma unroll
        for (int step = 32; step; step >>= 1) {
            const uint32_t e = __shfl(excl, sl + step);  // sl + step <= 63
            if (e <= idx) sl += step;
        }
        const uint32_t off = idx - __shfl(excl, sl);
        if (idx < total) f((sub0 + sl) * 512 + off);  // position in the list array
    }
}
template <class F>
__device__ __forceinline__ void for_candidates(const CannyBufs &bufs, F f)
{
    for_list(bufs.ncand, bufs.tiles, [&](size_t pos) { f((int)bufs.cand[pos]); });
}

template <bool WEAKONLY>
__global__ void __launch_bounds__(256) k_canny_union(const Region *__restrict__ regs, int H, int W, CannyBufs bufs)
{
    // On a dense map a tile is a handful of components and its ~200 border candidates ask for the same few unions
    // (tile root, neighbour's tile root) over and over, all of them chasing the same hot roots through L2.  A union is
    // idempotent, so a pair that is already in this block's small LDS table (someone has taken it) is skipped; a
    // collision only costs a repeated union.
    __shared__ unsigned long long s_seen[512];
    for (int i = threadIdx.x; i < 512; i += 256) s_seen[i] = ~0ull;
    __syncthreads();
    const Region r = regs[blockIdx.y];
    const size_t base = (size_t)r.img * H * W;
    const uint8_t *cm = bufs.cmap + base;
    int32_t *L = bufs.label + base;
    auto link = [&](int p, int q) {
        // the labels written by k_canny_gradnms are the tile-local roots: start from them (any ancestor will do)
        int a = ld_label(L, p, bufs.status), b = ld_label(L, q, bufs.status);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const unsigned long long key = ((unsigned long long)(uint32_t)a << 32) | (uint32_t)b;
        const uint32_t slot = ((uint32_t)a * 0x9e3779b1u ^ (uint32_t)b * 0x85ebca6bu) >> 23;
        if (atomicExch(&s_seen[slot], key) == key) return;
        uf_union(L, a, b, bufs.status);
    };
    for_list(bufs.nborder, bufs.tiles, [&](size_t pos) {
        // links inside a tile were made in LDS by k_canny_gradnms: only pairs that straddle a tile border are left
        const int p = (int)bufs.cand[pos];
        const int y = p / W, x = p - y * W;
        const int lx = (x - r.x0) % kCT_W, ly = (y - r.y0) % kCT_H;
        const bool right = x + 1 < r.x0 + r.cols, left = x - 1 >= r.x0, down = y + 1 < r.y0 + r.rows;
        const bool xr = lx == kCT_W - 1, xl = lx == 0, yd = ly == kCT_H - 1;
        if constexpr (WEAKONLY) {
            // the lists hold weak candidates only (map byte 0).  A strong neighbour (2) in another tile turns p's component
            // on: flagged at p's tile-local root, which k_canny_mark reads; weak neighbours in another tile are linked.
            const bool up = y - 1 >= r.y0, yu = ly == 0;
            bool on = false;
            if (left && xl) on = on || cm[p - 1] == 2;
            if (right && xr) on = on || cm[p + 1] == 2;
            if (up) {
                if (left && (yu || xl)) on = on || cm[p - W - 1] == 2;
                if (yu) on = on || cm[p - W] == 2;
                if (right && (yu || xr)) on = on || cm[p - W + 1] == 2;
            }
            if (down) {
                if (left && (yd || xl)) on = on || cm[p + W - 1] == 2;
                if (yd) on = on || cm[p + W] == 2;
                if (right && (yd || xr)) on = on || cm[p + W + 1] == 2;
            }
            if (on) bufs.flag[base + ld_label(L, p, bufs.status)] = 1;
            if (right && xr && cm[p + 1] == 0) link(p, p + 1);
            if (down) {
                if (left && (yd || xl) && cm[p + W - 1] == 0) link(p, p + W - 1);
                if (yd && cm[p + W] == 0) link(p, p + W);
                if (right && (yd || xr) && cm[p + W + 1] == 0) link(p, p + W + 1);
            }
        } else {
            if (right && xr && cm[p + 1] != 1) link(p, p + 1);
            if (down) {
                if (left && (yd || xl) && cm[p + W - 1] != 1) link(p, p + W - 1);
                if (yd && cm[p + W] != 1) link(p, p + W);
                if (right && (yd || xr) && cm[p + W + 1] != 1) link(p, p + W + 1);
            }
        }
    });
}

// a component is an edge component iff one of its tile-local parts owns a strong pixel: flag its global root
template <bool WEAKONLY>
__global__ void __launch_bounds__(256) k_canny_mark(const Region *__restrict__ regs, int H, int W, CannyBufs bufs)
{
    const size_t base = (size_t)regs[blockIdx.y].img * H * W;
    int32_t *L = bufs.label + base;
    for_list(bufs.nroot, bufs.tiles, [&](size_t pos) {
        const uint2 e = bufs.roots[pos];
        const int p = (int)e.x;
        int root = p;
        for (;;) {
            const int q = ld_label(L, root, bufs.status);
            if (q == root) break;
            root = q;
        }
        if (root != p) atomicMin(L + p, root);
        // WEAKONLY: k_canny_union flagged the tile-local roots of parts with a strong neighbour across a tile border (a flag
        // only ever says "this pixel's component is on", whoever set it, so reading it while others write is harmless)
        bool on = (e.y >> 16) != 0;
        if constexpr (WEAKONLY) on = on || __hip_atomic_load(bufs.flag + base + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (on) bufs.flag[base + root] = 1;
        if (root != p) bufs.roots[pos].x = (uint32_t)root;  // k_canny_emit reads the flag there: no second walk up the links
    });
}

// per-region edge counts: the sizes of the tile-local parts whose global root is flagged (= hysteresis)
template <bool WEAKONLY>
__global__ void __launch_bounds__(256) k_canny_emit(const Region *__restrict__ regs, int H, int W, CannyBufs bufs,
                                                    uint32_t *__restrict__ count)
{
    const size_t base = (size_t)regs[blockIdx.y].img * H * W;
    const int32_t *L = bufs.label + base;
    uint32_t mine = 0;
    if constexpr (WEAKONLY) {  // the strong pixels of this wavefront's tiles (for_list: 16 tiles per wavefront)
        const int lane = threadIdx.x & 63;
        const int tile = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * kWalkTiles + lane;
        if (lane < kWalkTiles && tile < bufs.tiles) mine = bufs.nstrong[(size_t)blockIdx.y * bufs.tiles + tile];
    }
    for_list(bufs.nroot, bufs.tiles, [&](size_t pos) {
        const uint2 e = bufs.roots[pos];  // e.x: the part's GLOBAL root (k_canny_mark wrote it back)
        if (bufs.flag[base + e.x]) mine += e.y & 0xffffu;
    });
    (void)L;
    const uint32_t tot = wave_sum_u32(mine);
    if ((threadIdx.x & 63) == 0 && tot) atomicAdd(count + blockIdx.y, tot);
}

// the edge map itself (standalone cv2.Canny): every candidate whose component is flagged
__global__ void __launch_bounds__(256) k_canny_paint(const Region *__restrict__ regs, int H, int W, CannyBufs bufs,
                                                     uint8_t *__restrict__ edges)
{
    const size_t base = (size_t)regs[blockIdx.y].img * H * W;
    const int32_t *L = bufs.label + base;
    for_candidates(bufs, [&](int p) {
        int root = p;
        for (;;) {
            const int q = ld_label(L, root, bufs.status);
            if (q == root) break;
            root = q;
        }
        if (bufs.flag[base + root]) edges[base + p] = 255;
    });
}

__global__ void k_full_regions(Region *regs, int B, int H, int W)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) regs[b] = Region{b, 0, 0, H, W};
}

// Tiles of the largest launches: one region per frame (H x W), or four quadrants per frame (ceil(H/2) x ceil(W/2) each).
size_t canny_list_tiles(Shape s)
{
    const size_t full = (size_t)cdiv(s.H, kCT_H) * cdiv(s.W, kCT_W);
    const size_t quad = (size_t)4 * cdiv((s.H + 1) / 2, kCT_H) * cdiv((s.W + 1) / 2, kCT_W);
    return (size_t)s.B * std::max(full, quad);
}

CannyBufs carve_canny(Carver &c, Shape s)
{
    CannyBufs b;
    const size_t n = (size_t)s.B * s.npx();
    b.cmap = c.take<uint8_t>(n);
    b.label = c.take<int32_t>(n);
    b.flag = c.take<uint8_t>(n);
    b.tiles = 0;  // set per launch
    b.cand = c.take<uint32_t>(canny_list_tiles(s) * 2048);
    b.ncand = c.take<uint32_t>(canny_list_tiles(s) * 4);
    b.roots = c.take<uint2>(canny_list_tiles(s) * 2048);
    b.nroot = c.take<uint32_t>(canny_list_tiles(s) * 4);
    b.nborder = c.take<uint32_t>(canny_list_tiles(s) * 4);
    b.nstrong = c.take<uint32_t>(canny_list_tiles(s));
    b.strong = c.take<uint32_t>((size_t)s.B * 4);
    return b;
}

}  // namespace

size_t canny_ws_bytes(Shape s)
{
    Carver c(nullptr);
    carve_canny(c, s);
    return c.total();
}

int launch_make_full_regions(Region *d_regions, Shape s, hipStream_t st)
{
    UWIE_LAUNCH(k_full_regions, dim3(cdiv(s.B, 64)), dim3(64), 0, st, d_regions, s.B, s.H, s.W);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

uint32_t *canny_strong_flags(void *ws, Shape s)
{
    Carver c(ws);
    return carve_canny(c, s).strong;
}

// The level-0 quadrants of every frame (regions = the quadtree's first four per image, W % 8 == 0): the gray plane of the
// colour-corrected frame and the pre-pass flags in one sweep (k_gray_strong); launch_canny then takes prepass_done = true.
bool gray_strong_takes(Shape s) { return s.W % 8 == 0 && s.W >= 64 && s.H >= 4 && tune().canny_prepass; }

int launch_gray_strong(const uint8_t *d_in, const int32_t *d_kind, uint8_t *d_gray, Shape s, const Region *d_regions, int nreg,
                       int max_rows, int max_cols, int high, int gray_shift, void *ws, hipStream_t st)
{
    Carver c(ws);
    CannyBufs bufs = carve_canny(c, s);
    UWIE_REQUIRE(gray_strong_takes(s) && nreg == s.B * 4 && max_cols % 4 == 0, "gray_strong: frame not taken");
    UWIE_HIP_CHECK(hipMemsetAsync(bufs.strong, 0, sizeof(uint32_t) * nreg, st));
    const int strips = cdiv(max_cols, kGsCols), bandgroups = cdiv(cdiv(max_rows, kPreRows), 4);
    float cr, cg, cb;
    gray_f32_coeffs(gray_shift, cr, cg, cb);
    UWIE_LAUNCH(k_gray_strong, dim3(strips * bandgroups, nreg), dim3(256), 0, st, d_in, d_kind, d_regions, s.H, s.W, strips, high,
                cr, cg, cb, d_gray, bufs.strong);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

// Levels >= 1 (the gray plane exists): quadrant histograms into d_hist ([nreg][768], accumulated) and the pre-pass flags of the
// same regions in one sweep; the flags must be zero on entry (k_q_decide leaves them so), launch_canny then takes prepass_done.
int launch_hist_strong(const uint8_t *d_in, const uint8_t *d_gray, Shape s, const Region *d_regions, int nreg, int max_rows,
                       int max_cols, int high, uint32_t *d_hist, void *ws, hipStream_t st)
{
    Carver c(ws);
    CannyBufs bufs = carve_canny(c, s);
    UWIE_REQUIRE(max_cols % 4 == 0 && max_cols >= 4, "hist_strong: quadrant widths must be multiples of four");
    const int strips = cdiv(max_cols, kPreCols), bandgroups = cdiv(cdiv(max_rows, kPreRows), 4);
    UWIE_LAUNCH(k_hist_strong, dim3(strips * bandgroups, nreg), dim3(256), 0, st, d_in, d_gray, d_regions, s.H, s.W, strips, high,
                bufs.strong, d_hist);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

int launch_canny(const uint8_t *d_gray, Shape s, const Region *d_regions, int nreg, int max_rows, int max_cols, int low,
                 int high, uint32_t *d_count, uint8_t *d_edges, void *ws, hipStream_t st, bool count_is_zeroed,
                 bool strong_is_zeroed, bool prepass_done)
{
    Carver c(ws);
    CannyBufs bufs = carve_canny(c, s);
    const int tiles_x = cdiv(max_cols, kCT_W), tiles_y = cdiv(max_rows, kCT_H);
    const dim3 tgrid(tiles_x * tiles_y, nreg), block(256);
    bufs.tiles = tiles_x * tiles_y;
    if ((size_t)nreg * bufs.tiles > canny_list_tiles(s)) {
        set_error("canny: %d regions of %d x %d exceed the candidate-list workspace", nreg, max_rows, max_cols);
        return UWIE_E_INVALID;
    }
    const dim3 lgrid(cdiv(cdiv(bufs.tiles, kWalkTiles), 4), nreg);  // list walkers: 4 wavefronts per block
    const bool weakonly = d_count && !d_edges;  // counts only: strong pixels need no labelling
    bufs.count = d_count;
    bufs.status = current_ctx() ? current_ctx()->d_status : nullptr;
    bufs.inject = tune().canny_fault_inject;
    if (d_count && !count_is_zeroed) UWIE_HIP_CHECK(hipMemsetAsync(d_count, 0, sizeof(uint32_t) * nreg, st));
    if (d_edges) UWIE_HIP_CHECK(hipMemsetAsync(d_edges, 0, (size_t)s.B * s.npx(), st));
    // Pre-pass (edge counts only: the standalone edge map keeps the single pass): regions without a pixel above the high
    // threshold have no edges.  Tuning canny_prepass = 0 disables it.
    if (d_edges || nreg > s.B * 4 || !tune().canny_prepass) bufs.strong = nullptr;
    if (bufs.strong && !prepass_done) {  // (prepass_done: launch_gray_strong has filled the flags of these regions)
        if (!strong_is_zeroed) UWIE_HIP_CHECK(hipMemsetAsync(bufs.strong, 0, sizeof(uint32_t) * nreg, st));
        if (max_cols >= 8) {  // the streaming pre-pass loads 8 bytes per row
            const int strips = cdiv(max_cols, kPreCols), bandgroups = cdiv(cdiv(max_rows, kPreRows), 4);
            UWIE_LAUNCH(k_canny_strong, dim3(strips * bandgroups, nreg), block, 0, st, d_gray, d_regions, s.H, s.W, strips, high,
                        bufs.strong);
        } else {
            UWIE_LAUNCH((k_canny_gradnms<true, false>), tgrid, block, 0, st, d_gray, d_regions, s.H, s.W, tiles_x, low, high, bufs);
        }
        UWIE_LAUNCH_CHECK();
    }
    const auto k_canny_gradnms_weak = k_canny_gradnms<false, true>;  // (names as the profiler reports them)
    const auto k_canny_gradnms_all = k_canny_gradnms<false, false>;
    if (weakonly) {
        UWIE_LAUNCH(k_canny_gradnms_weak, tgrid, block, 0, st, d_gray, d_regions, s.H, s.W, tiles_x, low, high, bufs);
        UWIE_LAUNCH_CHECK();
        UWIE_LAUNCH(k_canny_union<true>, lgrid, block, 0, st, d_regions, s.H, s.W, bufs);
        UWIE_LAUNCH_CHECK();
        UWIE_LAUNCH(k_canny_mark<true>, lgrid, block, 0, st, d_regions, s.H, s.W, bufs);
        UWIE_LAUNCH_CHECK();
    } else {
        UWIE_LAUNCH(k_canny_gradnms_all, tgrid, block, 0, st, d_gray, d_regions, s.H, s.W, tiles_x, low, high, bufs);
        UWIE_LAUNCH_CHECK();
        UWIE_LAUNCH(k_canny_union<false>, lgrid, block, 0, st, d_regions, s.H, s.W, bufs);
        UWIE_LAUNCH_CHECK();
        UWIE_LAUNCH(k_canny_mark<false>, lgrid, block, 0, st, d_regions, s.H, s.W, bufs);
        UWIE_LAUNCH_CHECK();
    }
    if (d_count) {
        if (weakonly) UWIE_LAUNCH(k_canny_emit<true>, lgrid, block, 0, st, d_regions, s.H, s.W, bufs, d_count);
        else UWIE_LAUNCH(k_canny_emit<false>, lgrid, block, 0, st, d_regions, s.H, s.W, bufs, d_count);
        UWIE_LAUNCH_CHECK();
    }
    if (d_edges) {
        UWIE_LAUNCH(k_canny_paint, lgrid, block, 0, st, d_regions, s.H, s.W, bufs, d_edges);
        UWIE_LAUNCH_CHECK();
    }
    return UWIE_OK;
}

}  // namespace uwie
