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
//   * k_chunk_hist takes one 3x256 histogram per 16384-pixel chunk (one streaming pass over the frame) and k_chunk_ulps
//     turns it into the chunk's ulp advance for every binade of the accumulator;
//   * k_cast_resolve (one workgroup per image and channel) walks the chunks 64 at a time: a prefix scan over their ulp
//     advances finds the first chunk that leaves the binade or holds a round-half-even tie, s jumps there in closed
//     form, and that chunk is drilled by the whole workgroup: 1024 threads x 16 pixels, the same scan on two levels, and
//     plain sequential float adds only for the 16 pixels around the event (and for the frame's first 1024 pixels).
#include "common.h"
#include "devutil.h"

namespace uwie {

constexpr int kChunkPx = 16384;  // pixels per histogram chunk

// Per chunk, channel and binade e of the accumulator: the number of ulps the chunk advances it by,
// D = sum_k hist[k] * RN(x_k / ulp_e), with bit 63 set when some present value ties (x_k / ulp_e = n + 1/2: the result
// then depends on the parity of the running sum and the chunk has to be walked).
constexpr uint64_t kTieBit = 1ull << 63;

// Round 3 layout of the chunk histogram: one 32-bit word per (byte value, lane column), the three channels' counters in
// 10-bit fields of that word.  The LDS bank of an atomic is then the lane's column (lane & 31 -- `ds_add_u32` is serviced in
// the two 32-lane halves, banks (a/4) mod 32), never the pixel's value: no bank conflicts whatever the picture, where the
// value-indexed copies of rounds 1-2 spent 72 % of their LDS cycles on conflicts (neighbouring pixels share their values).
// A column receives 16384 / 32 = 512 pixels of a chunk (+1 for a ragged tail), so a field cannot overflow.
// Output per chunk: the 768 counts as 16-bit pairs (values 2j and 2j+1 of a channel in one word: the operand layout of
// v_dot2_u32_u16 in k_chunk_ulps), and one byte per (channel, binade) telling whether a present value ties there (a byte
// value ties in exactly one binade, CastTables::tiebin).
constexpr int kHistCols = 32;
constexpr int kPairs = 3 * kCastBinades;
static_assert(kChunkPx / kHistCols + 3 < 1024, "10-bit fields");
static_assert(kChunkPx < 65536, "16-bit counts");

__global__ void __launch_bounds__(256) k_chunk_hist(const uint8_t *__restrict__ in, const CastTables *__restrict__ tab,
                                                    uint32_t *__restrict__ hist2, uint8_t *__restrict__ ties, int npx,
                                                    int nchunk)
{
    __shared__ __attribute__((aligned(16))) uint32_t h[256 * kHistCols];
    __shared__ uint32_t s_tie[kPairs];
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const uint32_t tb = tab->tiebin[tid];
    for (int i = tid; i < 256 * kHistCols / 4; i += 256) reinterpret_cast<uint4 *>(h)[i] = make_uint4(0, 0, 0, 0);
    if (tid < kPairs) s_tie[tid] = 0;
    __syncthreads();
    const uint8_t *img = in + (size_t)b * npx * 3;
    const int p0 = c * kChunkPx, p1 = min(npx, p0 + kChunkPx);
    const uint32_t colb = (uint32_t)(tid & (kHistCols - 1)) * 4u;
    char *hb = reinterpret_cast<char *>(h);
    typedef uint32_t __attribute__((aligned(1))) u32_any;
    // byte address of (value, column): value * 128 + column * 4; the caller shifts the value's byte to bit 7 of the word
    auto bump = [&](uint32_t moved, uint32_t inc) { atomicAdd(reinterpret_cast<uint32_t *>(hb + ((moved & 0x7f80u) | colb)), inc); };
    constexpr uint32_t kR = 1u, kG = 1u << 10, kB = 1u << 20;
    // whole groups of four pixels (12 bytes, any alignment: three dword loads); a thread's next kAhead groups are loaded
    // (unconditionally, index clamped: no branch for the compiler to park a vmcnt(0) in) before the atomics of this batch
    const int nfull = (p1 - p0) >> 2;
    constexpr int kAhead = 4;
    if (nfull > 0) {
        uint32_t cur[kAhead][3], nxt[kAhead][3];
        auto fetch = [&](int base, uint32_t (&d)[kAhead][3]) {
#pragma unroll
            for (int i = 0; i < kAhead; ++i) {
                const int g = min(base + i * 256, nfull - 1);
                const u32_any *q = reinterpret_cast<const u32_any *>(img + (size_t)(p0 + 4 * g) * 3);
                d[i][0] = q[0]; d[i][1] = q[1]; d[i][2] = q[2];
            }
        };
        auto consume = [&](int base, const uint32_t (&d)[kAhead][3]) {
#pragma unroll
            for (int i = 0; i < kAhead; ++i) {
                if (base + i * 256 < nfull) {
                    const uint32_t c0 = d[i][0], c1 = d[i][1], c2 = d[i][2];
                    // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
                    bump(c0 << 7, kR); bump(c0 >> 1, kG); bump(c0 >> 9, kB);
                    bump(c0 >> 17, kR); bump(c1 << 7, kG); bump(c1 >> 1, kB);
                    bump(c1 >> 9, kR); bump(c1 >> 17, kG); bump(c2 << 7, kB);
                    bump(c2 >> 1, kR); bump(c2 >> 9, kG); bump(c2 >> 17, kB);
                }
            }
        };
        // two buffers taking turns (a register copy at the end of a trip would wait for the loads it copies)
        constexpr int kStep = kAhead * 256;
        fetch(tid, cur);
        for (int base = tid; base < nfull; base += 2 * kStep) {
            fetch(base + kStep, nxt);
            consume(base, cur);
            fetch(base + 2 * kStep, cur);
            consume(base + kStep, nxt);
        }
    }
    // the chunk's last one to three pixels (frames whose pixel count is no multiple of four)
    for (int p = p0 + 4 * nfull + tid; p < p1; p += 256) {
        bump((uint32_t)img[(size_t)p * 3] << 7, kR);
        bump((uint32_t)img[(size_t)p * 3 + 1] << 7, kG);
        bump((uint32_t)img[(size_t)p * 3 + 2] << 7, kB);
    }
    __syncthreads();
    // thread v folds the 32 columns of value v (rotated start: lane l reads bank l + k)
    uint32_t r = 0, g = 0, bl = 0;
#pragma unroll 8
    for (int k = 0; k < kHistCols; ++k) {
        const uint32_t w = h[tid * kHistCols + ((tid + k) & (kHistCols - 1))];
        r += w & 1023u;
        g += (w >> 10) & 1023u;
        bl += w >> 20;
    }
    if (tb < (uint32_t)kCastBinades) {  // same-value stores: no atomics needed
        if (r) s_tie[tb] = 1;
        if (g) s_tie[kCastBinades + tb] = 1;
        if (bl) s_tie[2 * kCastBinades + tb] = 1;
    }
    const uint32_t r2 = r | (__shfl_down(r, 1) << 16), g2 = g | (__shfl_down(g, 1) << 16), b2 = bl | (__shfl_down(bl, 1) << 16);
    const size_t chunk = (size_t)b * nchunk + c;
    if (!(tid & 1)) {
        uint32_t *out = hist2 + chunk * 384 + (tid >> 1);
        out[0] = r2; out[128] = g2; out[256] = b2;
    }
    __syncthreads();
    if (tid < kPairs) ties[chunk * kPairs + tid] = (uint8_t)s_tie[tid];
}

// ---- round 4 (tuning entry_fuse): the level-0 quadrant histograms of the quadtree out of the same pass ----------------
// compute_Q's score of a level-0 quadrant follows from the quadrant's 3 x 256 byte histogram (k_q_decide), and this pass
// already counts every byte of the frame: a chunk takes its pixels one quadrant at a time -- the groups of four pixels
// left of W / 2, then those right of it, of the rows above H / 2, then of the rows below (a chunk is a raster run, so it
// usually meets two quadrants, the one across the middle row four) -- and folds the lane-column words after each: the
// fold is the quadrant's share of the chunk (stored as 16-bit pairs in `qpart`, summed per quadrant by
// k_quad_hist_reduce), the sum of the folds the chunk's histogram as before.  The frame's bytes are read once for cast
// detection AND the first quadtree level (k_q_hist<gray> read them again: 0.64 ms at 4K x 64).  Requires W % 8 == 0 (whole
// groups on either side of W / 2).
// Phase q (0 TL, 1 TR, 2 BL, 3 BR = the order of the quadtree's regions) of the chunk [p0, p1): its groups are numbered
// row-major over (rows of the half that the chunk touches) x (W / 8 groups of the side); the chunk holds numbers
// [ga, ga + n).  false: no group.
__device__ __forceinline__ bool quad_phase(int p0, int p1, int H, int W, int q, int &ra, int &ga, int &n)
{
    const int mr = H >> 1, mc = W >> 1, Gh = W >> 3;
    const int ya = p0 / W, yb = (p1 - 1) / W;
    const int hf = q >> 1, xs = (q & 1) ? mc : 0;
    ra = max(ya, hf ? mr : 0);
    const int rb = min(yb, hf ? H - 1 : mr - 1);
    ga = 0;
    n = 0;
    if (ra > rb) return false;
    int ge = Gh;
    if (ra == ya) ga = min(max((p0 - ya * W - xs) >> 2, 0), Gh);
    if (rb == yb) ge = min(max((p1 - yb * W - xs) >> 2, 0), Gh);
    n = (rb - ra) * Gh + ge - ga;
    return n > 0;
}

// One phase of a chunk (see above): the atomics of its n groups; ATT >= 0 also writes the groups' gray bytes (the gray plane
// of the colour-corrected frame, six_stadigy.py:149,177, for the cast kind the caller GUESSED: ATT = the attenuated channel,
// 0 none -- launch_cast_classify repairs the frames whose guess turns out wrong).  gside: the gray plane at the side's column.
template <int ATT>
__device__ __forceinline__ void chunk_phase(const uint8_t *__restrict__ side, uint8_t *__restrict__ gside, int W, int Gh, int ra, int ga,
                                            int n, int tid, char *hb, uint32_t colb, float cr, float cg, float cb)
{
    auto bump = [&](uint32_t moved, uint32_t inc) { atomicAdd(reinterpret_cast<uint32_t *>(hb + ((moved & 0x7f80u) | colb)), inc); };
    constexpr uint32_t kR = 1u, kG = 1u << 10, kB = 1u << 20;
    constexpr int kAhead = 4, kStep = kAhead * 256, kWords = ATT >= 0 ? 4 : 3;
    const int dr = 256 / Gh, dg = 256 - dr * Gh;  // a step of 256 groups in (row, group) terms
    const uint32_t px_safe = (uint32_t)(ra * W + 4 * ga);
    // cursor of the loads: group tid + ga of the phase's numbering, then 256 further per load
    int row = ra + (tid + ga) / Gh, gx = (tid + ga) - (row - ra) * Gh;
    auto fetch = [&](int j0, uint32_t (&d)[kAhead][kWords]) {
#pragma unroll
        for (int i = 0; i < kAhead; ++i) {
            uint32_t px = (uint32_t)(row * W + 4 * gx);
            px = j0 + i * 256 < n ? px : px_safe;  // (never under a branch: see k_chunk_hist)
            const uint32_t *a = reinterpret_cast<const uint32_t *>(side + (size_t)px * 3);
            d[i][0] = a[0]; d[i][1] = a[1]; d[i][2] = a[2];
            if constexpr (ATT >= 0) d[i][3] = px;
            gx += dg;
            row += dr;
            if (gx >= Gh) { gx -= Gh; ++row; }
        }
    };
    auto consume = [&](int j0, const uint32_t (&d)[kAhead][kWords]) {
#pragma unroll
        for (int i = 0; i < kAhead; ++i) {
            if (j0 + i * 256 < n) {
                const uint32_t c0 = d[i][0], c1 = d[i][1], c2 = d[i][2];
                bump(c0 << 7, kR); bump(c0 >> 1, kG); bump(c0 >> 9, kB);
                bump(c0 >> 17, kR); bump(c1 << 7, kG); bump(c1 >> 1, kB);
                bump(c1 >> 9, kR); bump(c1 >> 17, kG); bump(c2 << 7, kB);
                bump(c2 >> 1, kR); bump(c2 >> 9, kG); bump(c2 >> 17, kB);
                if constexpr (ATT >= 0) {
                    const uint32_t w3[3] = {c0, c1, c2};
                    *reinterpret_cast<uint32_t *>(gside + d[i][3]) = gray4_f32<ATT>(w3, cr, cg, cb);
                }
            }
        }
    };
    uint32_t cur[kAhead][kWords], nxt[kAhead][kWords];
    fetch(tid, cur);
    for (int base = tid; base < n; base += 2 * kStep) {
        fetch(base + kStep, nxt);
        consume(base, cur);
        fetch(base + 2 * kStep, cur);
        consume(base + kStep, nxt);
    }
}

// gray_out != nullptr: the gray plane is written on the way for the guessed cast kinds (guess[b])
__global__ void __launch_bounds__(256) k_chunk_hist_quad(const uint8_t *__restrict__ in, const CastTables *__restrict__ tab,
                                                         uint32_t *__restrict__ hist2, uint8_t *__restrict__ ties,
                                                         uint32_t *__restrict__ qpart, int npx, int nchunk, int H, int W,
                                                         uint8_t *__restrict__ gray_out, const int32_t *__restrict__ guess, float cr,
                                                         float cg, float cb)
{
    __shared__ __attribute__((aligned(16))) uint32_t h[256 * kHistCols];
    __shared__ uint32_t s_tie[kPairs];
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const uint32_t tb = tab->tiebin[tid];
    for (int i = tid; i < 256 * kHistCols / 4; i += 256) reinterpret_cast<uint4 *>(h)[i] = make_uint4(0, 0, 0, 0);
    if (tid < kPairs) s_tie[tid] = 0;
    __syncthreads();
    const uint8_t *img = in + (size_t)b * npx * 3;
    uint8_t *gimg = gray_out ? gray_out + (size_t)b * npx : nullptr;
    const int att = gray_out ? guess[b] : -1;  // (block-uniform)
    const int p0 = c * kChunkPx, p1 = min(npx, p0 + kChunkPx);
    const uint32_t colb = (uint32_t)(tid & (kHistCols - 1)) * 4u;
    char *hb = reinterpret_cast<char *>(h);
    const int Gh = W >> 3;
    uint32_t tr = 0, tg = 0, tbl = 0;  // the chunk's counts of byte value `tid`
    for (int q = 0; q < 4; ++q) {
        int ra, ga, n;
        if (!quad_phase(p0, p1, H, W, q, ra, ga, n)) continue;  // (block-uniform; the reducer evaluates the same test)
        const int xs = (q & 1) ? (W >> 1) : 0;
        const uint8_t *side = img + (size_t)xs * 3;
        uint8_t *gside = gimg + xs;
        if (att < 0) chunk_phase<-1>(side, gside, W, Gh, ra, ga, n, tid, hb, colb, cr, cg, cb);
        else if (att == 1) chunk_phase<1>(side, gside, W, Gh, ra, ga, n, tid, hb, colb, cr, cg, cb);
        else if (att == 2) chunk_phase<2>(side, gside, W, Gh, ra, ga, n, tid, hb, colb, cr, cg, cb);
        else chunk_phase<0>(side, gside, W, Gh, ra, ga, n, tid, hb, colb, cr, cg, cb);
        __syncthreads();
        // thread v folds (and clears) the 32 columns of value v: this quadrant's share of the chunk.  Four columns per LDS
        // instruction, the quads taken in a rotated order (eight neighbouring lanes cover the 32 banks): the fold is a quarter
        // of the LDS instructions of the b32 version, which made the two folds of a chunk cost as much as a third of its atomics
        uint32_t r = 0, g = 0, bl = 0;
#pragma unroll
        for (int k = 0; k < kHistCols / 4; ++k) {
            uint4 *wq = reinterpret_cast<uint4 *>(&h[tid * kHistCols + 4 * ((tid + k) & (kHistCols / 4 - 1))]);
            const uint4 w = *wq;
            *wq = make_uint4(0, 0, 0, 0);
            r += (w.x & 1023u) + (w.y & 1023u) + (w.z & 1023u) + (w.w & 1023u);
            g += ((w.x >> 10) & 1023u) + ((w.y >> 10) & 1023u) + ((w.z >> 10) & 1023u) + ((w.w >> 10) & 1023u);
            bl += (w.x >> 20) + (w.y >> 20) + (w.z >> 20) + (w.w >> 20);
        }
        tr += r; tg += g; tbl += bl;
        const uint32_t r2 = r | (__shfl_down(r, 1) << 16), g2 = g | (__shfl_down(g, 1) << 16), b2 = bl | (__shfl_down(bl, 1) << 16);
        if (!(tid & 1)) {
            uint32_t *out = qpart + (((size_t)b * nchunk + c) * 4 + q) * 384 + (tid >> 1);
            out[0] = r2; out[128] = g2; out[256] = b2;
        }
        __syncthreads();
    }
    if (tb < (uint32_t)kCastBinades) {  // same-value stores: no atomics needed
        if (tr) s_tie[tb] = 1;
        if (tg) s_tie[kCastBinades + tb] = 1;
        if (tbl) s_tie[2 * kCastBinades + tb] = 1;
    }
    const uint32_t r2 = tr | (__shfl_down(tr, 1) << 16), g2 = tg | (__shfl_down(tg, 1) << 16), b2 = tbl | (__shfl_down(tbl, 1) << 16);
    const size_t chunk = (size_t)b * nchunk + c;
    if (!(tid & 1)) {
        uint32_t *out = hist2 + chunk * 384 + (tid >> 1);
        out[0] = r2; out[128] = g2; out[256] = b2;
    }
    __syncthreads();
    if (tid < kPairs) ties[chunk * kPairs + tid] = (uint8_t)s_tie[tid];
}

// The cast kind of a frame guessed from a strided sample of its pixels (exact integer means of the sample, the reference's
// thresholds on them): only a guess -- k_chunk_hist_quad writes the gray plane for it BEFORE the real decision exists, and
// the frames whose decision differs get their plane again (k_quant_gray with `unless`).
// (2048 pixels: one CU gathers them in ~8 us -- 8192 scattered lines took it 23; a wrong guess only costs the repair)
constexpr int kGuessPx = 2048, kGuessThreads = 256, kGuessPer = kGuessPx / kGuessThreads;
__global__ void __launch_bounds__(kGuessThreads) k_kind_guess(const uint8_t *__restrict__ in, int npx, int32_t *__restrict__ guess)
{
    __shared__ uint32_t acc[3];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < 3) acc[tid] = 0;
    __syncthreads();
    const uint8_t *img = in + (size_t)b * npx * 3;
    const int n = min(npx, kGuessPx);
    const uint32_t stride = (uint32_t)(npx / n);
    typedef uint32_t __attribute__((aligned(1))) u32_any;
    uint32_t w[kGuessPer];
    // a pixel = the low three bytes of one (unaligned) dword; all of a thread's loads in flight.  The last pixel of the last
    // frame is read as the dword that ENDS at its last byte.
#pragma unroll
    for (int k = 0; k < kGuessPer; ++k) {
        const int i = min(tid + k * kGuessThreads, n - 1);
        const size_t px = (size_t)i * stride;
        const bool tail = px + 1 >= (size_t)npx;
        const uint32_t v = *reinterpret_cast<const u32_any *>(img + px * 3 - (tail ? 1 : 0));
        w[k] = tail ? v >> 8 : v;
    }
    uint32_t r = 0, g = 0, bl = 0;
#pragma unroll
    for (int k = 0; k < kGuessPer; ++k) {
        if (tid + k * kGuessThreads < n) {
            r += w[k] & 255u; g += (w[k] >> 8) & 255u; bl += (w[k] >> 16) & 255u;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        r += __shfl_xor(r, o); g += __shfl_xor(g, o); bl += __shfl_xor(bl, o);
    }
    if ((tid & 63) == 0) { atomicAdd(&acc[0], r); atomicAdd(&acc[1], g); atomicAdd(&acc[2], bl); }
    __syncthreads();
    if (tid == 0) {
        const float sc = 1.0f / (255.0f * (float)n);
        const float mr = (float)acc[0] * sc, mg = (float)acc[1] * sc, mb = (float)acc[2] * sc;
        int k = UWIE_CAST_NORMAL;
        if (mg > mr && mg > mb && (mg - mr) > 0.05f) k = UWIE_CAST_GREENISH;
        else if (mb > mr && mb > mg && (mb - mr) > 0.05f) k = UWIE_CAST_BLUISH;
        guess[b] = k;
    }
}

// hist[(b * 4 + q) * 768 + channel * 256 + value] += the quadrant's shares of its chunks.  grid (slices, 4, B), 384 threads:
// thread = one 16-bit pair of qpart; eight loads in flight.
__global__ void __launch_bounds__(384) k_quad_hist_reduce(const uint32_t *__restrict__ qpart, uint32_t *__restrict__ hist, int npx,
                                                          int nchunk, int H, int W)
{
    const int sl = blockIdx.x, q = blockIdx.y, b = blockIdx.z, kp = threadIdx.x;
    const int mr = H >> 1;
    const int pa = (q >> 1) ? mr * W : 0, pb = (q >> 1) ? npx : mr * W;  // the half's pixels
    if (pb <= pa) return;
    const int ca = pa / kChunkPx, cb = (pb - 1) / kChunkPx;  // its chunks (inclusive)
    const int per = (cb - ca + (int)gridDim.x) / (int)gridDim.x;
    const int c0 = ca + sl * per, c1 = min(cb + 1, c0 + per);
    if (c0 >= c1) return;
    const uint32_t *src = qpart + ((size_t)b * nchunk * 4 + q) * 384 + kp;
    uint32_t lo = 0, hi = 0;
    for (int c = c0; c < c1; c += 8) {
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = src[(size_t)min(c + i, c1 - 1) * 4 * 384];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int ra, ga, n;
            const int cc = c + i;
            // A phase the chunk kernel skipped was never written.  Only the first and the last chunk of a half can miss a side
            // (every other chunk lies inside the half and is at least as long as a row: W <= kChunkPx).
            bool on = cc < c1;
            if (on && (cc == ca || cc == cb)) on = quad_phase(cc * kChunkPx, min(npx, (cc + 1) * kChunkPx), H, W, q, ra, ga, n);
            if (on) {
                lo += w[i] & 0xffffu;
                hi += w[i] >> 16;
            }
        }
    }
    uint32_t *out = hist + ((size_t)(b * 4 + q) * 3 + (kp >> 7)) * 256 + 2 * (kp & 127);
    if (lo) atomicAdd(out, lo);
    if (hi) atomicAdd(out + 1, hi);
}

// D for kUlpChunks chunks x 3 channels (48 rows) per block.  D = sum_k n_k R_k with n_k <= 16384 summing to at most 16384
// and R_k < 2^26: R = hi * 2^13 + lo, both sums stay below 2^27, and v_dot2_u32_u16 takes two values per instruction from
// the pair tables (CastTables::RT2) -- one instruction per (row, value, binade) where the 64-bit mad + tie count of rounds
// 1-2 took three, with the table staged in LDS once per 16 chunks instead of fetched from L2 inside the loop.
// A thread owns one binade and eight rows; only the binades a frame of `npx` pixels can reach are computed (nb).
constexpr int kUlpChunks = 16, kUlpRows = 3 * kUlpChunks, kUlpRowStride = kUlpRows + 4, kUlpGroups = kUlpRows / 8;
static_assert(kUlpGroups * kCastBinades <= 256, "one thread per (binade, row group)");
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) k_chunk_ulps(const uint32_t *__restrict__ hist2, const uint8_t *__restrict__ ties,
                                                    const CastTables *__restrict__ tab, int total_chunks, int nb,
                                                    uint64_t *__restrict__ ulps)
{
    __shared__ uint2 t2[128 * kCastBinades];
    __shared__ __attribute__((aligned(16))) uint32_t n2[128 * kUlpRowStride];
    const int tid = threadIdx.x;
    const int g0 = blockIdx.x * kUlpChunks, nc = min(kUlpChunks, total_chunks - g0);
    {
        const uint2 *src = &tab->RT2[0][0];
        static_assert(128 * kCastBinades % 256 == 0, "whole trips");
        // fully unrolled: all loads of a thread in flight together (a rolled loop pays one L2 round trip per element)
#pragma unroll
        for (int i = 0; i < 128 * kCastBinades / 256; ++i) t2[tid + 256 * i] = src[tid + 256 * i];
        const int kp = tid & 127, r0 = (tid >> 7) * (kUlpRows / 2);
        uint32_t w[kUlpRows / 2];
#pragma unroll
        for (int i = 0; i < kUlpRows / 2; ++i) {
            // row = 3 * chunk + channel: consecutive rows are consecutive 128-word runs of hist2.  Clamped, not predicated: a
            // load under a branch gets its own vmcnt(0), 24 round trips instead of one.
            const int row = min(r0 + i, 3 * nc - 1);
            w[i] = hist2[((size_t)g0 * 3 + row) * 128 + kp];
        }
#pragma unroll
        for (int i = 0; i < kUlpRows / 2; ++i) n2[kp * kUlpRowStride + r0 + i] = r0 + i < 3 * nc ? w[i] : 0u;
    }
    __syncthreads();
    const int rg = tid / kCastBinades, ei = tid - rg * kCastBinades;
    if (rg >= kUlpGroups || ei >= nb) return;
    uint32_t lo[8] = {}, hi[8] = {};
    uint8_t tz[8];  // fetched ahead of the loop, clamped rows (see above)
#pragma unroll
    for (int q = 0; q < 8; ++q) tz[q] = ties[((size_t)g0 * 3 + min(rg * 8 + q, 3 * nc - 1)) * kCastBinades + ei];
#pragma unroll 4
    for (int kp = 0; kp < 128; ++kp) {
        const uint2 t = t2[kp * kCastBinades + ei];
        const uint4 na = *reinterpret_cast<const uint4 *>(&n2[kp * kUlpRowStride + rg * 8]);
        const uint4 nb4 = *reinterpret_cast<const uint4 *>(&n2[kp * kUlpRowStride + rg * 8 + 4]);
        const uint32_t n[8] = {na.x, na.y, na.z, na.w, nb4.x, nb4.y, nb4.z, nb4.w};
        const u16x2_t tl = __builtin_bit_cast(u16x2_t, t.x), th = __builtin_bit_cast(u16x2_t, t.y);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const u16x2_t nq = __builtin_bit_cast(u16x2_t, n[q]);
            lo[q] = __builtin_amdgcn_udot2(nq, tl, lo[q], false);
            hi[q] = __builtin_amdgcn_udot2(nq, th, hi[q], false);
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int row = rg * 8 + q;
        if (row >= 3 * nc) break;
        const size_t o = ((size_t)g0 * 3 + row) * kCastBinades + ei;
        ulps[o] = (((uint64_t)hi[q] << 13) + lo[q]) | (tz[q] ? kTieBit : 0);
    }
}

__device__ __forceinline__ float from_mantissa(uint32_t S, int e)
{
    return __uint_as_float(((uint32_t)(e + 127) << 23) | (S & 0x7fffffu));
}

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;

// ---- k_cast_resolve: one block of 16 wavefronts per (image, channel).
// Wavefront 0 walks the chunks 64 at a time: lane j holds the ulp count of chunk c+j for the current binade, a prefix scan
// finds the first chunk that leaves the binade or ties, s jumps to that chunk in closed form.  The whole block then drills
// that chunk: 1024 threads x 16 pixels, every thread sums the table entries of its pixels for binades e and e+1 (the table
// of all binades sits in LDS), a two-level prefix scan finds the first thread whose pixels leave the binade or tie, s jumps
// there in closed form, that thread's 16 pixels are added one by one like NumPy does, and the scan resumes behind it.  A
// binade is left ~24 times per channel at 4K.  (Rounds 1-2: one wavefront per channel drilling 64 x 256 pixels, then 64 x 4,
// with the two table rows in play fetched from global memory at every level and binade change: 142 us of a 1080p frame's
// 800; the 16384 table look-ups of a drill are LDS-rate bound and a single wavefront gets a fifth of that rate.)
constexpr int kResolveThreads = 1024, kResolveWaves = kResolveThreads / kWave, kResolvePer = kChunkPx / kResolveThreads;
// The accumulator leaves a binade every few pixels while it is small (binade e lasts ~2^(e+1) pixels of a mid-gray frame)
// and a pass of the block costs ~3 us whatever it covers: the first kResolveHead pixels of a frame are added one by one
// instead (4 ns each: the floats come from LDS four at a time), which replaces the ~11 passes up to s ~ 512.
constexpr int kResolveHead = 1024;
static_assert(kResolveHead % kResolvePer == 0 && kResolveHead / kResolvePer <= kWave, "the head's pixels belong to wavefront 0");
constexpr int kTblCols = kCastBinades + 1;  // entry [v][e+1] exists for every binade (the last column repeats)
static_assert(kResolvePer == 16, "four packed dwords of pixels per thread");

struct ResolveShared {
    uint64_t wtot[kResolveWaves];   // ulp advance of each wavefront's threads
    uint64_t wadv[kResolveWaves];   // advance in front of the wavefront's first bad thread
    uint32_t wbad[kResolveWaves];   // that thread's lane (64: none)
    uint32_t px[kResolveWaves][4];  // its sixteen pixels
    float head[kResolveHead];       // the first pixels of the frame as floats (see drill_block)
    float s;
    int c;
};

// Advance the accumulator `s` over pixels [p0, p0+cnt) of one channel (stride 3 bytes); all threads of the block call with
// the same arguments and return the same value.
__device__ float drill_block(const uint8_t *__restrict__ chan, int p0, int cnt, float s, const uint64_t *__restrict__ tbl,
                             ResolveShared &sh, bool more_after)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // this thread's sixteen pixels, one byte each (missing pixels: 0, which adds nothing anywhere below)
    uint32_t vals[4] = {0, 0, 0, 0};
    {
        const int ln = max(0, min(kResolvePer, cnt - t * kResolvePer));
        const uint8_t *run = chan + (size_t)(p0 + t * kResolvePer) * 3;
        // 48 bytes = 12 (unaligned) dwords; the last one reaches two bytes past the sixteenth pixel's byte
        if (ln == kResolvePer && (more_after || t * kResolvePer + kResolvePer < cnt)) {
            const u32_unaligned *w = reinterpret_cast<const u32_unaligned *>(run);
            uint32_t d[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) d[q] = w[q];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int byte = 3 * q;
                vals[q >> 2] |= ((d[byte >> 2] >> (8 * (byte & 3))) & 0xffu) << (8 * (q & 3));
            }
        } else {
            for (int q = 0; q < ln; ++q) vals[q >> 2] |= (uint32_t)run[(size_t)q * 3] << (8 * (q & 3));
        }
    }
    auto first_bad = [&](bool bad, uint64_t adv_here, uint64_t &adv) {  // -> global index of the first thread with `bad`
        const uint64_t mask = __ballot(bad);
        const int Lw = mask ? (int)__builtin_ctzll(mask) : kWave;
        const uint64_t a = shfl_u64(adv_here, min(Lw, kWave - 1));
        if (lane == Lw) {
#pragma unroll
            for (int q = 0; q < 4; ++q) sh.px[wave][q] = vals[q];
        }
        if (lane == 0) {
            sh.wbad[wave] = (uint32_t)Lw;
            sh.wadv[wave] = a;
        }
        __syncthreads();
        int L = kResolveThreads;
        adv = 0;
#pragma unroll
        for (int j = kResolveWaves - 1; j >= 0; --j) {
            const uint32_t lw = sh.wbad[j];
            if (lw < (uint32_t)kWave) {
                L = j * kWave + (int)lw;
                adv = sh.wadv[j];
            }
        }
        return L;
    };
    auto add_thread = [&](int L, float acc) {  // the sixteen pixels of thread L, one after the other like NumPy
        const uint32_t *q = sh.px[L >> 6];
        const uint32_t w[4] = {q[0], q[1], q[2], q[3]};
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = acc + px_norm_fast((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
        return acc;
    };
    int first = 0;
    if (s < 0.25f) {  // the frame's first drill (s == 0 unless the frame starts with black pixels)
        if (t < kResolveHead / kResolvePer) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sh.head[t * kResolvePer + i] = px_norm_fast((vals[i >> 2] >> (8 * (i & 3))) & 0xffu);
        }
        __syncthreads();
        for (int i = 0; i < kResolveHead; i += 4) {  // (every thread adds them: s stays uniform without a broadcast)
            const float4 x = *reinterpret_cast<const float4 *>(&sh.head[i]);
            s = (((s + x.x) + x.y) + x.z) + x.w;
        }
        first = kResolveHead / kResolvePer;
    }
    while (first < kResolveThreads) {
        int L;
        if (s >= 0.25f) {
            const int e = (int)(__float_as_uint(s) >> 23) - 127;
            const int ei = min(e - kCastBinadeMin, kCastBinades - 1);
            // one 64-bit entry per (byte value, binade): the ulp count with its tie flag moved up to bit 40, so that one 64-bit
            // add per binade and pixel accumulates both (16 counts below 2^26 stay far below 2^40)
            uint64_t D[2] = {0, 0};
            if (t >= first) {
                const uint64_t *te = tbl + ei;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint64_t *q = te + ((vals[i >> 2] >> (8 * (i & 3))) & 0xffu) * kTblCols;
                    D[0] += q[0];
                    D[1] += q[1];
                }
            }
            const uint32_t T[2] = {(uint32_t)(D[0] >> 40), (uint32_t)(D[1] >> 40)};
            D[0] &= (1ull << 40) - 1;
            D[1] &= (1ull << 40) - 1;
            // events inside this pass: as long as s stays in e or e+1 the sums above remain valid
            for (;;) {
                const int ec = (int)(__float_as_uint(s) >> 23) - 127;
                if (ec != e && !(ec == e + 1 && ei + 1 < kCastBinades)) { L = -1; break; }  // another binade: new pass
                const int w = ec - e;
                const uint64_t incl = wave_incl_scan_u64(t >= first ? D[w] : 0);
                if (lane == kWave - 1) sh.wtot[wave] = incl;
                __syncthreads();
                uint64_t offs = 0, total = 0;
#pragma unroll
                for (int j = 0; j < kResolveWaves; ++j) {
                    const uint64_t v = sh.wtot[j];
                    offs += j < wave ? v : 0;
                    total += v;
                }
                const uint32_t S = (__float_as_uint(s) & 0x7fffffu) | 0x800000u;
                const bool bad = t >= first && (T[w] != 0 || S + offs + incl >= (1ull << 24));
                uint64_t adv;
                L = first_bad(bad, offs + incl - (t >= first ? D[w] : 0), adv);  // (exclusive prefix of the bad thread)
                if (L >= kResolveThreads) adv = total;
                s = from_mantissa(S + (uint32_t)adv, ec);
                if (L >= kResolveThreads) break;
                s = add_thread(L, s);
                first = L + 1;
                if (first >= kResolveThreads) break;
            }
            if (L < 0) continue;
            break;  // no bad thread left, or every thread consumed
        }
        // tiny accumulator: skip threads whose pixels are all zero (adding 0.0f changes nothing)
        uint64_t unused;
        L = first_bad(t >= first && (vals[0] | vals[1] | vals[2] | vals[3]) != 0, 0, unused);
        if (L >= kResolveThreads) break;
        s = add_thread(L, s);
        first = L + 1;
        __syncthreads();  // (sh.px / sh.wbad are written again before the next barrier on this path)
    }
    __syncthreads();
    return s;
}

__global__ void __launch_bounds__(kResolveThreads) k_cast_resolve(const uint8_t *__restrict__ in, const uint64_t *__restrict__ ulps,
                                                                  const CastTables *__restrict__ tab, int npx, int nchunk,
                                                                  float *__restrict__ sums)
{
    extern __shared__ uint64_t tbl[];  // [256][kTblCols]
    __shared__ ResolveShared sh;
    const int ch = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63;
    {
        constexpr int kTrips = (256 * kTblCols + kResolveThreads - 1) / kResolveThreads;
        uint32_t r[kTrips];
#pragma unroll
        for (int k = 0; k < kTrips; ++k) {  // (clamped, all in flight)
            const int i = min(t + k * kResolveThreads, 256 * kTblCols - 1);
            r[k] = tab->RT[i / kTblCols][min(i % kTblCols, kCastBinades - 1)];
        }
#pragma unroll
        for (int k = 0; k < kTrips; ++k) {
            const int i = t + k * kResolveThreads;
            if (i < 256 * kTblCols) tbl[i] = (uint64_t)(r[k] & 0x7fffffffu) | ((uint64_t)(r[k] >> 31) << 40);
        }
    }
    __syncthreads();
    const uint8_t *chan = in + (size_t)b * npx * 3 + ch;
    const uint64_t *u = ulps + (((size_t)b * nchunk) * 3 + ch) * kCastBinades;
    float s = 0.0f;
    int c = 0;
    // Wavefront 0 fetches the ulp counts it will most likely need after a drill -- chunks c+1.. in the NEXT binade (a chunk is
    // drilled because the accumulator leaves its binade there; a tie leaves it where it was) -- before the drill, so the
    // walk that follows does not start with a memory round trip (~1.5 us, ~24 times per channel).
    uint64_t v_pre = 0;
    int pre_c = -1, pre_ei = -1;
    for (;;) {
        if (t < kWave) {  // chunks passed in closed form
            int ei = -1;
            while (c < nchunk && s >= 0.25f) {
                const int e = (int)(__float_as_uint(s) >> 23) - 127;
                ei = min(e - kCastBinadeMin, kCastBinades - 1);
                const bool have = c + lane < nchunk;
                uint64_t v;
                if (c == pre_c && ei == pre_ei) v = v_pre;  // (uniform)
                else v = have ? u[(size_t)(c + lane) * 3 * kCastBinades + ei] : kTieBit;
                pre_c = -1;
                const uint64_t incl = wave_incl_scan_u64(v & ~kTieBit);
                const uint32_t S = (__float_as_uint(s) & 0x7fffffu) | 0x800000u;
                const bool bad = (v & kTieBit) || S + incl >= (1ull << 24);
                const uint64_t mask = __ballot(bad);
                const int L = mask ? (int)__builtin_ctzll(mask) : kWave;
                const uint64_t adv = L > 0 ? shfl_u64(incl, L - 1) : 0;
                s = from_mantissa(S + (uint32_t)adv, e);
                c += L;
                if (L < kWave) break;  // chunk c has to be drilled (or c == nchunk: the padding lanes are "bad")
            }
            if (c < nchunk && ei >= 0) {
                pre_c = c + 1;
                pre_ei = min(ei + 1, kCastBinades - 1);
                v_pre = pre_c + lane < nchunk ? u[(size_t)(pre_c + lane) * 3 * kCastBinades + pre_ei] : kTieBit;
            }
            if (lane == 0) {
                sh.s = s;
                sh.c = c;
            }
        }
        __syncthreads();
        s = sh.s;
        c = sh.c;
        if (c >= nchunk) break;
        const int cpx = min(kChunkPx, npx - c * kChunkPx);
        const bool more_after = b + 1 < (int)gridDim.y || c * kChunkPx + cpx < npx;  // pixels of the buffer behind this chunk
        s = drill_block(chan, c * kChunkPx, cpx, s, tbl, sh, more_after);  // (ends in a barrier: sh.s may be written again)
        ++c;
    }
    if (t == 0) sums[b * 3 + ch] = s;
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
    c.take<uint32_t>((size_t)s.B * nchunk * 384);
    c.take<uint8_t>((size_t)s.B * nchunk * 3 * kCastBinades);
    c.take<uint64_t>((size_t)s.B * nchunk * 3 * kCastBinades);
    c.take<float>((size_t)s.B * 3);
    return c.total();
}

__global__ void k_quant_gray(const uint8_t *__restrict__ in, const int32_t *__restrict__ kind, uint8_t *__restrict__ gray, int npx,
                             int shift, const int32_t *__restrict__ unless);

size_t quad_part_bytes(Shape s) { return (size_t)s.B * cdiv((long long)s.npx(), kChunkPx) * 4 * 384 * sizeof(uint32_t); }

int launch_quad_hist_reduce(const uint32_t *qpart, Shape s, uint32_t *d_hist, hipStream_t st)
{
    const int nchunk = cdiv((long long)s.npx(), kChunkPx);
    const int half_chunks = cdiv(nchunk, 2) + 1;
    const int slices = std::max(1, std::min(32, half_chunks / 8));  // one trip of eight loads per thread where the job has the chunks
    UWIE_LAUNCH(k_quad_hist_reduce, dim3(slices, 4, s.B), dim3(384), 0, st, qpart, d_hist, (int)s.npx(), nchunk, s.H, s.W);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

// ef != nullptr (round 4; the caller checked entry_fuse_takes): the chunk pass also leaves the level-0 quadrants' shares of
// every chunk in ef->qpart ([B][chunks][4][384] 16-bit pairs) for launch_quad_hist_reduce and, with ef->gray, writes the
// gray plane of the colour-corrected frame -- for a guessed cast kind first, again for the frames whose guess was wrong.
int launch_cast_classify(uwie_ctx *ctx, const uint8_t *d_in, Shape s, int32_t *d_kind, float *d_mean, void *ws,
                         hipStream_t st, const EntryFuse *ef)
{
    Carver c(ws);
    const int npx = (int)s.npx();
    const int nchunk = cdiv(npx, kChunkPx);
    uint32_t *hist2 = c.take<uint32_t>((size_t)s.B * nchunk * 384);
    uint8_t *ties = c.take<uint8_t>((size_t)s.B * nchunk * 3 * kCastBinades);
    uint64_t *ulps = c.take<uint64_t>((size_t)s.B * nchunk * 3 * kCastBinades);
    float *sums = c.take<float>((size_t)s.B * 3);
    const bool spec_gray = ef && ef->gray && ef->guess && d_kind;
    if (ef) {
        UWIE_REQUIRE(ef->qpart && s.W % 8 == 0 && s.W >= 64 && s.W <= kChunkPx && s.H >= 2, "cast_classify: quadrant shares need W % 8 == 0");
        float cr = 0, cg = 0, cb = 0;
        if (spec_gray) {
            gray_f32_coeffs(ef->gray_shift, cr, cg, cb);
            UWIE_LAUNCH(k_kind_guess, dim3(s.B), dim3(kGuessThreads), 0, st, d_in, npx, ef->guess);
            UWIE_LAUNCH_CHECK();
        }
        UWIE_LAUNCH(k_chunk_hist_quad, dim3(nchunk, s.B), dim3(256), 0, st, d_in, ctx->d_cast, hist2, ties, ef->qpart, npx, nchunk, s.H,
                    s.W, spec_gray ? ef->gray : (uint8_t *)nullptr, (const int32_t *)ef->guess, cr, cg, cb);
    } else {
        UWIE_LAUNCH(k_chunk_hist, dim3(nchunk, s.B), dim3(256), 0, st, d_in, ctx->d_cast, hist2, ties, npx, nchunk);
    }
    UWIE_LAUNCH_CHECK();
    // the accumulator never exceeds the pixel count: binades above floor(log2(npx)) are never asked for
    const int nb = std::min(kCastBinades, (31 - __builtin_clz((unsigned)std::max(npx, 1))) - kCastBinadeMin + 1);
    UWIE_LAUNCH(k_chunk_ulps, dim3(cdiv(s.B * nchunk, kUlpChunks)), dim3(256), 0, st, hist2, ties, ctx->d_cast, s.B * nchunk, nb,
                ulps);
    UWIE_LAUNCH_CHECK();
    constexpr size_t kResolveLds = (size_t)256 * kTblCols * sizeof(uint64_t);  // 71680 B: asked for once per context
    if (!ctx->attr_cast_resolve) {
        UWIE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_cast_resolve), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)kResolveLds));
        ctx->attr_cast_resolve = true;
    }
    UWIE_LAUNCH(k_cast_resolve, dim3(3, s.B), dim3(kResolveThreads), kResolveLds, st, d_in, ulps, ctx->d_cast, npx, nchunk, sums);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_cast_decide, dim3(cdiv(s.B, 64)), dim3(64), 0, st, sums, s.B, npx, d_kind, d_mean);
    UWIE_LAUNCH_CHECK();
    if (spec_gray) {  // frames whose guess was wrong (none, as a rule: every block returns at once)
        const int blocks = std::min(grid_for((s.npx() + 3) / 4, 4096), 256);
        UWIE_LAUNCH(k_quant_gray, dim3(blocks, s.B), dim3(256), 0, st, d_in, (const int32_t *)d_kind, ef->gray, npx, ef->gray_shift,
                    (const int32_t *)ef->guess);
        UWIE_LAUNCH_CHECK();
    }
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
// unless != nullptr: frames with unless[b] == kind[b] are skipped (their plane was written for that kind already)
__global__ void __launch_bounds__(256) k_quant_gray(const uint8_t *__restrict__ in, const int32_t *__restrict__ kind,
                                                    uint8_t *__restrict__ gray, int npx, int shift,
                                                    const int32_t *__restrict__ unless)
{
    // (x * 255).astype(u8) of the (colour-corrected) value is a map of the byte: one weighted table per channel in LDS
    __shared__ uint32_t wq[3][256];
    const int b = blockIdx.y;
    const int k = kind ? kind[b] : 0;
    if (unless && unless[b] == k) return;
    const uint8_t *img = in + (size_t)b * npx * 3;
    uint8_t *g = gray + (size_t)b * npx;
    const bool aligned = (npx & 3) == 0;
    for (int i = threadIdx.x; i < 768; i += 256) {
        const int c = i >> 8;
        const uint32_t q = quant_u8(px_val(i & 255, px_atten(k, c)));
        const uint32_t w15[3] = {9798u, 19235u, 3735u}, w14[3] = {4899u, 9617u, 1868u};  // gray_fixed's coefficients
        wq[c][i & 255] = q * (shift == 15 ? w15[c] : w14[c]);
    }
    __syncthreads();
    const uint32_t half = shift == 15 ? 16384u : 8192u, sh = shift == 15 ? 15 : 14;
    for (int p = (blockIdx.x * 256 + threadIdx.x) * 4; p < npx; p += gridDim.x * 1024) {
        const int n = min(4, npx - p);
        const Px4 v = load_px4(img + (size_t)p * 3, n, aligned);
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (wq[0][v.r[i]] + wq[1][v.g[i]] + wq[2][v.b[i]] + half) >> sh;
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
    UWIE_LAUNCH(k_quant_gray, dim3(blocks, s.B), dim3(256), 0, st, d_in, d_kind, d_gray, (int)s.npx(), gray_shift,
                (const int32_t *)nullptr);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

}  // namespace uwie
