// Fused tail of the dehazing strategies (six_stadigy.py:230-259): the stages after the guided filter, arranged
// so that each global dependency costs exactly one sweep:
//   k_restore_planar_hist  restore_image (S6:183-188) -> planar float32 + first-digit histogram of the radix select
//   (k_select.hip)         two more histogram sweeps -> order statistics -> percentiles (S6:196-197, 216-217)
//   k_stretch_lab_lut      enhance_contrast (S6:198) [+ white_balance (S6:218)] -> (x*255).astype(u8) -> RGB2LAB
//                          (S6:204) -> LAB bytes + per-tile CLAHE histogram -> clipped, redistributed tile LUT
//   k_clahe_apply_out      CLAHE interpolation (S6:206) -> LAB2RGB -> /255 (S6:207) [-> x**gamma (S6:224)]
//                          -> (y*255).astype(u8) (S6:430); everything after LAB2RGB is a 256-entry LUT per launch
//   k_stretch_out          the CLAHE-free tail (strategy 3): stretch -> white_balance -> output
// Arithmetic is the same as in the unfused stage kernels (k_tail.hip, k_clahe.hip), operation for operation.
#include <cstdlib>

#include "common.h"
#include "devutil.h"
#include "restore.h"

namespace uwie {

namespace {

typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
typedef uint4 __attribute__((aligned(1))) u128_unaligned;

// np.clip(v, 0, 1) for a value that is not NaN: one instruction (the clamp output modifier)
__device__ __forceinline__ float clip01(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); }
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef UWIE_CLAHE_PREFETCH
#define UWIE_CLAHE_PREFETCH 0
#endif
#ifndef UWIE_STRETCH_WAVES
#define UWIE_STRETCH_WAVES 5
#endif
__device__ __forceinline__ uint8_t sat_u8(int v) { return (uint8_t)min(max(v, 0), 255); }
#define UWIE_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

// grid (nblk, B), block 256.  LIN: the histogram is over lin_digit() (select_lin_*), else over the top 11 key bits.
// planar == nullptr: histogram only (the consumers recompute the image from S, restore.h); ghist == nullptr: image
// only; only != nullptr: images none of whose three planes is flagged are skipped.
// COLLECT (with LIN): values whose digit lies in one of the plane's NW predicted windows (LinState::wlo, wspan; two for
// two percentiles, four for strategy 3's four) are filed into the window's list on the way: staged in LDS, moved out in
// batches (a block reserves list space once per batch and window).
// V = double: the ES surface's float64 image (restore.h four64), float64 planes and lists.
template <bool LIN, bool COLLECT, int NW = 2, typename V = float>
__global__ void __launch_bounds__(256) k_restore_planar_hist(RestoreSrc S, int npx, V *__restrict__ planar,
                                                             uint32_t *__restrict__ ghist,
                                                             const uint32_t *__restrict__ only, LinState *__restrict__ lin,
                                                             V *__restrict__ lists, uint32_t cap)
{
    constexpr int NB = LIN ? 2052 : 2048;
    constexpr bool F64 = sizeof(V) == 8;
    // FAST (the float32 linear-digit sweep of strategies 1-3): whole 4-pixel groups take a straight-line path -- numerators
    // from a float64 table in LDS, no range test on the divisor (t is the guided filter's output, clipped to [0.1, 1]), one
    // saturation test per group instead of four operations per value (round 3: 102 -> ~60 VALU instructions per pixel)
    constexpr bool FAST = LIN && !F64;
    // stages: 6 KB (12: float64, NW 4).  Measured at 4K x 64 (round 3, A/B in one run: profiles/ab.sh): 128-entry stages cost
    // 60 % (a full stage sends its candidates to the list one global atomic each); 192-entry stages flushed twice as often
    // with a float32 table (32.4 KB of LDS: five blocks per CU instead of four) are no faster than this
    constexpr int NS = COLLECT ? 3 * NW : 1, SN = COLLECT ? (NW == 2 && !F64 ? 256 : 128) : 1;
    __shared__ uint32_t h[3][NB];
    __shared__ V stg[NS][SN];
    __shared__ uint32_t scount[NS], sbase[NS];
    __shared__ double dtab[FAST ? 768 : 1];
    const int b = blockIdx.y, tid = threadIdx.x;
    if (only && !(only[3 * b] | only[3 * b + 1] | only[3 * b + 2])) return;
    if (ghist) {
        for (int i = tid; i < 3 * NB; i += 256) (&h[0][0])[i] = 0;
        if (tid < NS) scount[tid] = 0;
    }
    // the whole sweep for one flavour of the restore (float64 quotient, or UWIE_INTER_F32T's float32 one)
    auto sweep = [&](auto &R) {
    R.init(S, b, (size_t)npx, dtab);
    __syncthreads();
    V *o0 = planar + (size_t)b * 3 * npx, *o1 = o0 + npx, *o2 = o1 + npx;
    const bool aligned = (npx & 3) == 0;
    // COLLECT: the top bits of a bin's LDS counter say which window (1 .. NW) the bin belongs to, so the histogram
    // atomic's return value tells whether the value is a candidate: no separate window test per value (measured, round 3:
    // atomics without a return value plus the window test in the vector unit: 1.83 instead of 1.49 ms at 4K x 64)
    constexpr int kFlagShift = NW == 2 ? 30 : 29;
    constexpr uint32_t kCntMask = (1u << kFlagShift) - 1u;
    if (COLLECT) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const uint32_t lo = lin[3 * b + c].wlo[w], span = lin[3 * b + c].wspan[w];
                if (lo != kLinNoWin)
                    for (uint32_t i = tid; i <= span; i += 256) h[c][lo + i] = (uint32_t)(w + 1) << kFlagShift;
            }
        __syncthreads();
    }
    // Key digits (sign, exponent, 2 mantissa bits) take a dozen values on a whole frame, so plain LDS atomics would
    // serialise 64 deep: each thread counts runs of equal digits in registers and touches LDS only when the digit
    // changes.  Linear digits spread over thousands of bins and change from pixel to pixel, so they go straight to LDS,
    // except the two saturated bins (clipped 0 and 1 fill whole regions), which are counted in registers.
    uint32_t cur[3] = {0, 0, 0}, run[3] = {0, 0, 0}, sat0[3] = {0, 0, 0}, sat1[3] = {0, 0, 0};
    auto bump = [&](int c, uint32_t d) {
        if (d != cur[c]) {
            if (run[c]) atomicAdd(&h[c][cur[c]], run[c]);
            cur[c] = d;
            run[c] = 0;
        }
        ++run[c];
    };
    auto file = [&](int c, int w, V x) {  // x is a candidate of window w of channel c
        const int j = c * NW + w;
        const uint32_t pos = atomicAdd(&scount[j], 1u);
        if (pos < (uint32_t)SN) {
            stg[j][pos] = x;
        } else {  // a burst (a smooth region at the percentile's level): straight to the list
            const uint32_t idx = atomicAdd(&lin[3 * b + c].gcount[w], 1u);
            if (idx < cap) lists[((size_t)(3 * b + c) * kLinLists + w) * cap + idx] = x;
        }
    };
    auto flush = [&](bool force) {  // all threads of the block call
        __syncthreads();
        bool need = force;
        for (int j = 0; j < NS; ++j) need = need || scount[j] > (uint32_t)(SN * 3 / 4);
        __syncthreads();  // nobody files again before everybody has looked (round 4: the decision has to be block-uniform)
        if (!need) return;
        if (tid < NS) {
            const uint32_t c = min(scount[tid], (uint32_t)SN);
            if (c) sbase[tid] = atomicAdd(&lin[3 * b + tid / NW].gcount[tid % NW], c);
        }
        __syncthreads();
        for (int j = 0; j < NS; ++j) {
            const uint32_t c = min(scount[j], (uint32_t)SN), base = sbase[j];
            V *L = lists + ((size_t)(3 * b + j / NW) * kLinLists + (j % NW)) * cap;
            for (uint32_t i = tid; i < c; i += 256)
                if (base + i < cap) L[base + i] = stg[j][i];
        }
        __syncthreads();
        if (tid < NS) scount[tid] = 0;
        __syncthreads();
    };
    const int step = gridDim.x * 1024, iters = (npx + step - 1) / step;  // block-uniform trip count
    // FAST: the loads of trip it + 1 are issued before the arithmetic of trip it (a trip ends in LDS atomics whose return
    // values it waits for: without this a wavefront's memory latency is only hidden by the other wavefronts of its SIMD)
    uint32_t w_nx[3] = {0, 0, 0};
    double tv_nx[4] = {1.0, 1.0, 1.0, 1.0};
    bool have_nx = false;
    if constexpr (FAST) {
        const int p0 = (blockIdx.x * 256 + tid) * 4;
        have_nx = ghist && p0 + 4 <= npx;
        if (have_nx) R.load_four(p0, w_nx, tv_nx);
    }
    for (int it = 0; it < iters; ++it) {
        const int p = it * step + (blockIdx.x * 256 + tid) * 4;
        const int n = min(4, npx - p);
        bool done = false;
        if constexpr (FAST) {
            const bool have = have_nx;
            const uint32_t w[3] = {w_nx[0], w_nx[1], w_nx[2]};
            const double tv[4] = {tv_nx[0], tv_nx[1], tv_nx[2], tv_nx[3]};
            {
                const int pn = p + step;
                have_nx = ghist && it + 1 < iters && pn + 4 <= npx;
                if (have_nx) R.load_four(pn, w_nx, tv_nx);
            }
            if (have) {
                done = true;
                float v[3][4];  // restored values before the clip
                R.four_raw(w, tv, v);
                // One saturation test per group: clipped zeros and ones fill whole regions (their bins would serialise the
                // LDS atomics 64 deep), so a group that holds any takes the per-value route below; every other group's
                // digit is (uint)(v * 2048) + 1 straight away.
                const float lo = fminf(fminf(__builtin_fminf(v[0][0], __builtin_fminf(v[0][1], v[0][2])), __builtin_fminf(v[0][3], __builtin_fminf(v[1][0], v[1][1]))),
                                       fminf(__builtin_fminf(v[1][2], __builtin_fminf(v[1][3], v[2][0])), __builtin_fminf(v[2][1], __builtin_fminf(v[2][2], v[2][3]))));
                const float hi = fmaxf(fmaxf(__builtin_fmaxf(v[0][0], __builtin_fmaxf(v[0][1], v[0][2])), __builtin_fmaxf(v[0][3], __builtin_fmaxf(v[1][0], v[1][1]))),
                                       fmaxf(__builtin_fmaxf(v[1][2], __builtin_fmaxf(v[1][3], v[2][0])), __builtin_fmaxf(v[2][1], __builtin_fmaxf(v[2][2], v[2][3]))));
                uint32_t old[3][4];
                if (lo > 0.0f && hi < 1.0f) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int c = 0; c < 3; ++c) old[c][i] = atomicAdd(&h[c][1 + (uint32_t)(v[c][i] * 2048.0f)], 1u);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float x = v[c][i];
                            const bool z = !(x > 0.0f), o = x >= 1.0f;
                            sat0[c] += z;
                            sat1[c] += o;
                            old[c][i] = z || o ? 0u : atomicAdd(&h[c][1 + (uint32_t)(x * 2048.0f)], 1u);
                            v[c][i] = z ? 0.0f : o ? 1.0f : x;
                        }
                }
                if (COLLECT) {  // a candidate is never a clipped value: v is what it is
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            if (old[c][i] > kCntMask) file(c, (int)(old[c][i] >> kFlagShift) - 1, v[c][i]);
                }
                if (planar) {
                    if (aligned) {
                        *reinterpret_cast<float4 *>(o0 + p) = make_float4(clip01(v[0][0]), clip01(v[0][1]), clip01(v[0][2]), clip01(v[0][3]));
                        *reinterpret_cast<float4 *>(o1 + p) = make_float4(clip01(v[1][0]), clip01(v[1][1]), clip01(v[1][2]), clip01(v[1][3]));
                        *reinterpret_cast<float4 *>(o2 + p) = make_float4(clip01(v[2][0]), clip01(v[2][1]), clip01(v[2][2]), clip01(v[2][3]));
                    } else {
                        for (int i = 0; i < 4; ++i) {
                            o0[p + i] = clip01(v[0][i]);
                            o1[p + i] = clip01(v[1][i]);
                            o2[p + i] = clip01(v[2][i]);
                        }
                    }
                }
            }
        }
        if (n > 0 && !done) {
            V r[3][4];
            if constexpr (F64) R.four64(p, n, r);
            else R.four(p, n, r);
            if (ghist && LIN) {
                // all twelve histogram atomics first, their return values (window flags) afterwards: one LDS round trip
                uint32_t old[3][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        // r is clipped to [0, 1]: lin_digit(x) == (uint)(x * 2048) + (x > 0)
                        const uint32_t d = (uint32_t)(r[c][i] * (V)2048) + (r[c][i] > (V)0 ? 1u : 0u);
                        const bool live = i < n;
                        sat0[c] += live && d == 0;
                        sat1[c] += live && d == (uint32_t)kLinBins - 1;
                        old[c][i] = live && d - 1 < (uint32_t)kLinBins - 2 ? atomicAdd(&h[c][d], 1u) : 0u;
                    }
                if (COLLECT) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            if (old[c][i] >> kFlagShift) file(c, (int)(old[c][i] >> kFlagShift) - 1, r[c][i]);
                }
            } else if (ghist) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i < n) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            if constexpr (F64) bump(c, (uint32_t)(f64_key(r[c][i]) >> 53));
                            else bump(c, f32_key(r[c][i]) >> 21);
                        }
                    }
                }
            }
            if (planar) {
                if constexpr (F64) {
                    if (aligned && n == 4) {  // (npx a multiple of 4: plane rows of two doubles are 16-byte aligned)
                        V *o[3] = {o0, o1, o2};
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            *reinterpret_cast<double2 *>(o[c] + p) = make_double2(r[c][0], r[c][1]);
                            *reinterpret_cast<double2 *>(o[c] + p + 2) = make_double2(r[c][2], r[c][3]);
                        }
                    } else {
                        for (int i = 0; i < n; ++i) {
                            o0[p + i] = r[0][i];
                            o1[p + i] = r[1][i];
                            o2[p + i] = r[2][i];
                        }
                    }
                } else if (aligned && n == 4) {
                    *reinterpret_cast<float4 *>(o0 + p) = make_float4(r[0][0], r[0][1], r[0][2], r[0][3]);
                    *reinterpret_cast<float4 *>(o1 + p) = make_float4(r[1][0], r[1][1], r[1][2], r[1][3]);
                    *reinterpret_cast<float4 *>(o2 + p) = make_float4(r[2][0], r[2][1], r[2][2], r[2][3]);
                } else {
                    for (int i = 0; i < n; ++i) {
                        o0[p + i] = r[0][i];
                        o1[p + i] = r[1][i];
                        o2[p + i] = r[2][i];
                    }
                }
            }
        }
        // (a barrier per trip would tie the block's wavefronts together every 1024 pixels: the stages hold 128 / 256
        // candidates per window against ~3 arriving per trip, and a full stage overflows straight into the list)
        if (COLLECT && (it & 7) == 7) flush(false);
    }
    if (!ghist) return;
    if (COLLECT) flush(true);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (run[c]) atomicAdd(&h[c][cur[c]], run[c]);
        if (LIN) {
            const uint32_t z = wave_sum_u32(sat0[c]), o = wave_sum_u32(sat1[c]);
            if ((tid & 63) == 0) {
                if (z) atomicAdd(&h[c][0], z);
                if (o) atomicAdd(&h[c][kLinBins - 1], o);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < 3 * NB; i += 256) {
        const uint32_t c = (&h[0][0])[i] & (COLLECT ? kCntMask : 0xffffffffu);
        if (c) atomicAdd(&ghist[(size_t)(b * 3 + i / NB) * kSelGroupStride + (i % NB)], c);
    }
    };  // sweep
    if constexpr (FAST) {
        if (S.t32) {
            RestoreImgT<2, true> R;
            sweep(R);
            return;
        }
    }
    RestoreImgT<FAST ? 2 : 0> R;
    sweep(R);
}

// ---------------------------------------------------------------------------------------------------------------------
// Rank-counting sweep (round 4; six_stadigy.py:183-199): what enhance_contrast needs of the restored image is two order
// statistics per channel, and k_lin_predict has already bracketed each of them by a window of histogram bins.  So instead
// of evaluating the float64 restore for every value and histogramming it (k_restore_planar_hist<true, true>: 52 VALU
// instructions and 9.7 LDS atomics with return per pixel), this sweep
//   * classifies every value with a CHEAP float32 restore (one fused multiply-add on v_rcp_f32 of the float32 transmission)
//     against the window edges widened by kRankMargin -- the float32 value is within 4e-6 of the float64 one (|I - A| / t
//     <= 10: 2^-24 on t, 2^-23 on the reciprocal, half an ulp of a value below 16 on the sum and on the exact value's own
//     rounding), the margin is 1e-5 = 0.02 bins -- and counts the values definitely BELOW each window with wavefront ballots
//     (v_cmp into a scalar pair + s_bcnt1: no LDS traffic);
//   * queues the VALUES that fall inside a widened window (~2 % of them) in LDS and evaluates those, and only those, with the
//     exact float64 sequence, densely (every lane busy), every eight trips: a value whose exact bin lies below the window adds
//     to its count, one inside the window goes to the window's list, as in the histogram sweep.
// k_rank_scan (k_select.hip) then finds each rank in (below, below + list length) and k_lin_finish selects inside the list.
// A rank outside its window (the sample misled the prediction, ~6e-5 per window), or a list that overflowed, flags the
// plane for the generic sweeps exactly as before.  Results are the same order statistics: identical bytes.
// grid (nblk, B), block 256.  float64 transmission only (UWIE_INTER_F32T keeps the histogram sweep: there the float32
// restore IS the value).
constexpr float kRankMargin = 1e-5f;
constexpr int kRankNW = 2, kRankNS = 3 * kRankNW;
// Everything the exact stage touches is PRIVATE to a wavefront (round 4, second version: the first one queued per block --
// an LDS atomic with return per lane and trip, six barriers per flush -- and was slower than the histogram sweep it replaces,
// 2.1 against 1.7 ms, with the same 55 VALU instructions per pixel).  A wavefront appends the values it has to look at exactly
// to its own queue (positions from ballots: no atomics), evaluates 64 of them at a time -- one per lane -- as soon as it has
// 64, and stages the window members it finds per (channel, window) until there are 64 to write to the list in one piece.
// No barrier after the set-up, no LDS atomic in the loop.
constexpr int kRankQW = 64 + 64 * 12;  // queue entries per wavefront: fewer than 64 left over + at most 12 per lane and trip
constexpr int kRankSW = 128;           // stage entries per wavefront and (channel, window): flushed when 64 are there

__device__ __forceinline__ void wave_lds_sync()  // LDS executes a wavefront's instructions in order: a compiler fence is all
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (free functions with everything passed explicitly: as lambdas, the closure of a closure stayed in scratch memory)
struct RankWave {  // what one wavefront's exact stage works with
    const uint8_t *img;
    const double *t;
    const float *ftab;              // LDS: I - A as float32, [3][256]
    float a0, a1, a2;
    const uint32_t *s_wlo, *s_wspan;  // LDS
    const float *s_lo, *s_hi;
    uint32_t *s_below;
    uint32_t *myq;                  // LDS: this wavefront's queue
    float (*mystg)[kRankSW];        // LDS: this wavefront's stages
    uint32_t *gcount0;              // &lin[3 b].gcount[0]; channel c is c * gstride words further
    uint32_t gstride;
    float *lists0;                  // the lists of (3 b, window 0)
    uint32_t cap;
    int lane;
};

// the whole wavefront: the n <= 128 entries of stage j go to their list in one piece
__device__ __forceinline__ void rank_flush_stage(const RankWave &X, int j, uint32_t &scj)
{
    const uint32_t n = scj;
    const int c = j / kRankNW, w = j % kRankNW;
    uint32_t base = 0;
    if (X.lane == 0) base = atomicAdd(X.gcount0 + (size_t)c * X.gstride + w, n);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    float *L = X.lists0 + ((size_t)c * (kLinLists / kRankCapMul) + w) * X.cap;
    wave_lds_sync();
    for (uint32_t i = X.lane; i < n; i += 64)
        if (base + i < X.cap) L[base + i] = X.mystg[j][i];
    wave_lds_sync();
    scj = 0;
}

// lanes < m evaluate queue entries first .. first + m - 1 exactly: restore_image's float64 sequence (restore.h one_fast /
// one, written out with the channel's constants selected by value).  counted: the ballots have seen the value's float32
// image -- "not below" for the window it was queued for (exact: below or inside), definite and right for the other one.
__device__ __forceinline__ void rank_exact_batch(const RankWave &X, uint32_t first, uint32_t m, bool counted, uint32_t (&sc)[kRankNS])
{
    wave_lds_sync();
    int jj = -1;  // the (channel, window) this lane's value turned out to belong to, -1: none
    float x = 0.0f;
    if ((uint32_t)X.lane < m) {
        const uint32_t e = X.myq[first + X.lane];
        const int p = (int)(e >> 2), c = (int)(e & 3u);
        const double tv = X.t[p];
        const uint32_t u = X.img[(size_t)p * 3 + c];
        const float df = X.ftab[c * 256 + (int)u], ac = c == 0 ? X.a0 : c == 1 ? X.a1 : X.a2;
        const double n = (double)df;
        double q;
        if (RestoreImgT<1>::recip_ok(tv)) {
            const double y = RestoreImgT<1>::recip(tv), q0 = n * y;
            q = fma(fma(-tv, q0, n), y, q0);
        } else {
            q = n / tv;
        }
        x = clip01((float)(q + (double)ac));  // S6:186-188
        const uint32_t d = lin_digit(x);
        const float v32 = fmaf(df, __builtin_amdgcn_rcpf((float)tv), ac);
#pragma unroll
        for (int w = 0; w < kRankNW; ++w) {
            const int j = c * kRankNW + w;
            const uint32_t wlo = X.s_wlo[j];
            if (wlo == kLinNoWin) continue;
            if (counted && !(v32 >= X.s_lo[j] && v32 <= X.s_hi[j])) continue;  // not queued for this window
            if (d < wlo) atomicAdd(&X.s_below[j], 1u);  // (rare: the margin's width under the window, or the ragged tail)
            else if (d - wlo <= X.s_wspan[j]) jj = j;
        }
    }
#pragma unroll
    for (int j = 0; j < kRankNS; ++j) {
        const uint64_t mk = __ballot(jj == j);
        if (mk) {  // uniform
            const uint32_t pos = sc[j] + (uint32_t)__popcll(mk & ((1ull << X.lane) - 1));
            if (jj == j) X.mystg[j][pos] = x;
            sc[j] += (uint32_t)__popcll(mk);
            if (sc[j] >= 64u) rank_flush_stage(X, j, sc[j]);
        }
    }
}

__global__ void __launch_bounds__(256) k_restore_rank(RestoreSrc S, int npx, LinState *__restrict__ lin, float *__restrict__ lists,
                                                      uint32_t cap)
{
    constexpr int NW = kRankNW, NS = kRankNS, QW = kRankQW, SW = kRankSW;
    __shared__ uint32_t wq[4][QW];     // (pixel << 2) | channel of the values to evaluate exactly
    __shared__ float wstg[4][NS][SW];
    __shared__ float ftab[768];
    __shared__ uint32_t s_wlo[NS], s_wspan[NS];  // the windows in bins (s_wlo = kLinNoWin: none)
    __shared__ float s_lo[NS], s_hi[NS];         // their edges as values, widened by the margin
    __shared__ uint32_t s_below[NS];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < NS) {
        s_below[tid] = 0;
        const LinState &L = lin[3 * b + tid / NW];
        const uint32_t lo = L.wlo[tid % NW], span = L.wspan[tid % NW];
        s_wlo[tid] = lo;
        s_wspan[tid] = span;
        // bin d holds ((d - 1) / 2048, d / 2048) [d = 1: (0, 1/2048)]; no window: nothing is inside, everything "below"
        s_lo[tid] = lo == kLinNoWin ? __builtin_inff() : (float)(lo - 1) * (1.0f / 2048.0f) - kRankMargin;
        s_hi[tid] = lo == kLinNoWin ? -__builtin_inff() : (float)(lo + span) * (1.0f / 2048.0f) + kRankMargin;
    }
    RestoreImgT<1> R;
    R.init(S, b, (size_t)npx, ftab);
    __syncthreads();
    // (in vector registers on purpose: as scalars the twelve edges, the compare masks of a group and the six counters do not
    // fit the 102 scalar registers, and the first build moved them through v_readlane / v_writelane all the time)
    float elo[NS], ehi[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        elo[j] = s_lo[j];
        ehi[j] = s_hi[j];
    }
    const float a0 = R.a[0], a1 = R.a[1], a2 = R.a[2];
    const uint8_t *img = S.in + (size_t)b * npx * 3;
    const double *tpl = S.t + (size_t)b * npx;
    uint32_t *myq = wq[wv];
    uint32_t cnt[NS] = {0, 0, 0, 0, 0, 0};  // values definitely NOT below each window: wavefront totals
    uint32_t seen = 0;                       // values per channel the ballots have seen (uniform)
    uint32_t qn = 0;                         // entries in this wavefront's queue (uniform)
    uint32_t sc[NS] = {0, 0, 0, 0, 0, 0};    // entries in its stages (uniform)

    const RankWave X{img, tpl, ftab, a0, a1, a2, s_wlo, s_wspan, s_lo, s_hi, s_below, myq, wstg[wv],
                     &lin[3 * b].gcount[0], (uint32_t)(sizeof(LinState) / sizeof(uint32_t)),
                     lists + (size_t)(3 * b) * (kLinLists / kRankCapMul) * cap, cap, lane};

    const int step = gridDim.x * 1024, iters = (npx + step - 1) / step;  // block-uniform trip count
    typedef uint32_t __attribute__((aligned(1))) u32_a1;
#define UWIE_RANK_LOAD4(p_, w_, tv_)                                                                                   \
    do {                                                                                                                \
        const u32_a1 *q_ = reinterpret_cast<const u32_a1 *>(img + (size_t)(p_) * 3);                                    \
        (w_)[0] = q_[0]; (w_)[1] = q_[1]; (w_)[2] = q_[2];                                                              \
        const double2_a8 ta_ = *reinterpret_cast<const double2_a8 *>(tpl + (p_)),                                       \
                         tb_ = *reinterpret_cast<const double2_a8 *>(tpl + (p_) + 2);                                   \
        (tv_)[0] = ta_.x; (tv_)[1] = ta_.y; (tv_)[2] = tb_.x; (tv_)[3] = tb_.y;                                         \
    } while (0)
    // The next trip's group is loaded UNCONDITIONALLY from a clamped position (the frame's last whole group when there is no
    // next group): written as `if (have) load`, the loads sat in a block of their own whose join copies their registers -- the
    // ISA showed s_waitcnt vmcnt right behind them, i.e. every trip waited for the "prefetch" it had just issued (k_chunk_hist
    // had the same defect in round 3).
    // (Two trips of loads in flight instead of one -- 90 registers, still five wavefronts per SIMD -- measured 1.307 / 1.421 ms against
    // 1.306 / 1.411 at 4K x 64: the sweep already moves 8.4 GB at 6.4 TB/s, its loads are not what it waits for.)
    uint32_t w_nx[3] = {0, 0, 0};
    double tv_nx[4] = {1.0, 1.0, 1.0, 1.0};
    bool have_nx = false;
    const int p_last = npx - 4;  // (launch_restore_rank requires npx >= 4)
    {
        const int p0 = (blockIdx.x * 256 + tid) * 4;
        have_nx = p0 + 4 <= npx;
        UWIE_RANK_LOAD4(min(p0, p_last), w_nx, tv_nx);
    }
    for (int it = 0; it < iters; ++it) {
        const int p = it * step + (blockIdx.x * 256 + tid) * 4;
        const bool have = have_nx;
        const uint32_t w0 = w_nx[0], w1 = w_nx[1], w2 = w_nx[2];
        const double tv[4] = {tv_nx[0], tv_nx[1], tv_nx[2], tv_nx[3]};
        {
            const int pn = p + step;
            have_nx = it + 1 < iters && pn + 4 <= npx;
            UWIE_RANK_LOAD4(min(pn, p_last), w_nx, tv_nx);
        }
        uint32_t fm = 0;  // bit 4 c + i: value i of channel c lies in a widened window
        seen += 4u * (uint32_t)__popcll(__ballot(have));
        if (have) {
            float y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = __builtin_amdgcn_rcpf((float)tv[i]);
            const uint32_t ub[3][4] = {{w0 & 255, w0 >> 24, (w1 >> 16) & 255, (w2 >> 8) & 255},
                                       {(w0 >> 8) & 255, w1 & 255, w1 >> 24, (w2 >> 16) & 255},
                                       {(w0 >> 16) & 255, (w1 >> 8) & 255, w2 & 255, w2 >> 24}};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float ac = c == 0 ? a0 : c == 1 ? a1 : a2;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = fmaf(ftab[c * 256 + (int)ub[c][i]], y[i], ac);
                    bool in = false;
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        // counted: the values NOT below the window (v >= edge), so that the same compare serves the membership
                        // test (written as !(v < edge) it cost a second compare: 71 instead of 48 per group)
                        const bool ge = v >= elo[c * NW + w];
                        cnt[c * NW + w] += (uint32_t)__popcll(__ballot(ge));
                        in = in | (ge & (v <= ehi[c * NW + w]));
                    }
                    fm |= (uint32_t)in << (4 * c + i);
                }
                __builtin_amdgcn_sched_barrier(0);  // one channel's masks at a time (scalar register budget)
            }
        }
        // the ragged last group of the frame: exact, NOT in the ballot counts (at most one lane per frame)
        const bool ragged = !have && p < npx;
        // ---- append this trip's values to the wavefront's queue: positions from ballots
        const uint32_t nmine = (uint32_t)__popc(fm);
        if (__ballot(nmine != 0)) {
            uint32_t pre = 0, tot = 0;
#pragma unroll
            for (int bit = 0; bit < 4; ++bit) {  // nmine <= 12
                const uint64_t mk = __ballot((nmine >> bit) & 1u);
                pre += (uint32_t)__popcll(mk & ((1ull << lane) - 1)) << bit;
                tot += (uint32_t)__popcll(mk) << bit;
            }
            uint32_t at = qn + pre;
            for (uint32_t f = fm; f; f &= f - 1, ++at) {
                const int k = __ffs(f) - 1;
                myq[at] = ((uint32_t)(p + (k & 3)) << 2) | (uint32_t)(k >> 2);
            }
            qn += tot;
            while (qn >= 64u) {  // uniform
                qn -= 64u;
                rank_exact_batch(X, qn, 64u, true, sc);
            }
        }
        if (__ballot(ragged)) {  // uniform; the lanes' own queue area above the live entries serves as scratch
            const uint32_t n3 = ragged ? (uint32_t)(npx - p) * 3u : 0u;  // <= 9 values
            wave_lds_sync();
            if (ragged)
                for (uint32_t i = 0; i < n3; ++i) myq[qn + i] = ((uint32_t)(p + (int)(i / 3u)) << 2) | (i % 3u);
            const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)n3, (int)__builtin_ctzll(__ballot(ragged))));
            rank_exact_batch(X, qn, m, false, sc);
        }
    }
    if (qn) rank_exact_batch(X, 0u, qn, true, sc);
#pragma unroll
    for (int j = 0; j < NS; ++j)
        if (sc[j]) rank_flush_stage(X, j, sc[j]);
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        uint32_t k = cnt[j];  // the wavefront's total sits in the lanes that took part in every trip (lane 0 did): the maximum
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) k = max(k, (uint32_t)__shfl_xor((int)k, o));
        if (lane == 0 && seen != k) atomicAdd(&s_below[j], seen - k);  // below = seen - not below
    }
    __syncthreads();
    if (tid < NS && s_below[tid]) atomicAdd(&lin[3 * b + tid / NW].below[tid % NW], s_below[tid]);
}

struct Stretch {  // per image: lo and denominator per channel, for one or two chained stretches
    StretchDiv d1[3], d2[3];
    float lo1[3], lo2[3];
    int two;
    __device__ __forceinline__ void load(const float *pct, int b, int stride, float eps, int two_)
    {
        two = two_;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float *p = pct + (size_t)(b * 3 + c) * stride;
            lo1[c] = p[0];
            d1[c].set((p[1] - p[0]) + eps);
            lo2[c] = two ? p[2] : 0.f;
            d2[c].set(two ? (p[3] - p[2]) + eps : 1.f);
        }
    }
    // for results that are quantised to a byte right away
    __device__ __forceinline__ float apply(float v, int c) const
    {
        v = clip01(d1[c].quot_unit(v - lo1[c]));
        if (two) v = clip01(d2[c].quot_unit(v - lo2[c]));
        return v;
    }
    // Single stretch with a divisor inside StretchDiv's fast range (allfast()): the four values of one channel as two
    // packed float32 pairs (v_pk_add / v_pk_mul / v_pk_fma: the same operations in the same order, two lanes per
    // instruction), then (x * 255).astype(u8) as an index.  v is the clipped restored value.
    __device__ __forceinline__ bool allfast() const { return !two && d1[0].fast && d1[1].fast && d1[2].fast; }
    __device__ __forceinline__ void codes4(const float (&v)[4], int c, uint32_t (&code)[4]) const
    {
        const float den = d1[c].den, y = d1[c].y, lo = lo1[c];
        const f32x2 nd = {-den, -den}, yy = {y, y};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x2 n = f32x2{v[2 * h], v[2 * h + 1]} - lo;
            f32x2 q = n * y;
            q = __builtin_elementwise_fma(__builtin_elementwise_fma(nd, q, n), yy, q);
            q = __builtin_elementwise_fma(__builtin_elementwise_fma(nd, q, n), yy, q);
            const f32x2 s = f32x2{clip01(q.x), clip01(q.y)} * 255.0f;
            code[2 * h] = (uint32_t)(int)s.x;
            code[2 * h + 1] = (uint32_t)(int)s.y;
        }
    }
    // for results that are kept as float32
    __device__ __forceinline__ float apply_exact(float v, int c) const
    {
        v = clip01(d1[c].quot(v - lo1[c]));
        if (two) v = clip01(d2[c].quot(v - lo2[c]));
        return v;
    }
};

struct ClaheGeom {
    int H, W, tx, ty, tw, th, clip;
    float lutScale;
};

__device__ __forceinline__ uint32_t block_incl_scan_256(uint32_t v, uint32_t *wsum)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t incl = wave_incl_scan_u32(v);
    __syncthreads();
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    for (int i = 0; i < w; ++i) incl += wsum[i];
    return incl;
}

// grid (tx*ty, B), block 256.  SRC 0: stored planes; 1: the restored image is recomputed from src (restore.h); 2: the
// code-domain strategies (k_codes.hip): the u8 frame src.in goes through a per-(image, channel) code LUT (codes) and the
// stretch is skipped.
// BS threads per block: 256, or 1024 for small jobs (one workgroup per tile is all the parallelism a single 1080p frame has:
// 64 workgroups on 256 CUs; sixteen wavefronts walk the tile's pixels, the first four finish the histogram: 58 -> ~25 us)
template <int SRC, int BS = 256>
__global__ void __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(BS == 256 ? UWIE_STRETCH_WAVES : 4, 8))) k_stretch_lab_lut(const LabTables *__restrict__ T, const float *__restrict__ planar,
                                                         RestoreSrc src, const float *__restrict__ pct, int pct_stride,
                                                         float eps, int two, const uint8_t *__restrict__ codes, ClaheGeom g,
                                                         uint8_t *__restrict__ lab, uint8_t *__restrict__ lut)
{
    __shared__ uint32_t h[4][256];
    __shared__ uint32_t wsum[4];
    __shared__ uint16_t s_gamma[256], s_cbrt[3072];
    __shared__ int s_fwd[9];
    const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, w = (tid >> 6) & 3, w4 = w;
    for (int i = tid; i < 1024; i += BS) (&h[0][0])[i] = 0;
    for (int i = tid; i < 256; i += BS) s_gamma[i] = T->gamma[i];
    for (int i = tid; i < 3072; i += BS) s_cbrt[i] = T->cbrt[i];
    if (tid < 9) s_fwd[tid] = T->fwd[tid];
    __shared__ uint16_t s_gcode[SRC == 2 ? 768 : 1];  // gamma table after the code LUT, per channel
    Stretch S;
    if (SRC != 2) S.load(pct, b, pct_stride, eps, two);
    if (SRC == 2)
        for (int i = tid; i < 768; i += BS) s_gcode[i] = T->gamma[codes[(size_t)b * 768 + i]];
    __syncthreads();
    const int npx = g.H * g.W;
    const float *r0 = planar + (size_t)b * 3 * npx, *r1 = r0 + npx, *r2 = r1 + npx;
    const uint8_t *img8 = src.in + (size_t)b * npx * 3;
    __shared__ double dtab[SRC == 1 ? 768 : 1];
    // the tile's pixels for one flavour of the restore (float64 quotient, or UWIE_INTER_F32T's float32 one)
    auto pixels = [&](auto &R) {
    if (SRC == 1) {
        R.init(src, b, (size_t)npx, dtab);
        __syncthreads();
    }
    uint8_t *labimg = lab + (size_t)b * npx * 3;
    const int ty = tile / g.tx, txi = tile % g.tx;
    const int area = g.tw * g.th;
    constexpr int Lscale = (116 * 255 + 50) / 100;
    constexpr int Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
    auto to_lab = [&](float v0, float v1, float v2, uint32_t &L, uint32_t &a, uint32_t &bb) {
        // SRC 2: v0..v2 carry the frame's bytes
        const int R = SRC == 2 ? s_gcode[(int)v0] : s_gamma[quant_u8(S.apply(v0, 0))];
        const int G = SRC == 2 ? s_gcode[256 + (int)v1] : s_gamma[quant_u8(S.apply(v1, 1))];
        const int B = SRC == 2 ? s_gcode[512 + (int)v2] : s_gamma[quant_u8(S.apply(v2, 2))];
        const int fX = s_cbrt[UWIE_DESCALE(R * s_fwd[0] + G * s_fwd[1] + B * s_fwd[2], 12)];
        const int fY = s_cbrt[UWIE_DESCALE(R * s_fwd[3] + G * s_fwd[4] + B * s_fwd[5], 12)];
        const int fZ = s_cbrt[UWIE_DESCALE(R * s_fwd[6] + G * s_fwd[7] + B * s_fwd[8], 12)];
        L = sat_u8(UWIE_DESCALE(Lscale * fY + Lshift, 15));
        a = sat_u8(UWIE_DESCALE(500 * (fX - fY) + 128 * (1 << 15), 15));
        bb = sat_u8(UWIE_DESCALE(200 * (fY - fZ) + 128 * (1 << 15), 15));
    };
    // the tile's pixels inside the image: four per thread, 16-byte loads per plane, 12-byte LAB store
    const int x_lo = txi * g.tw, x_hi = min(x_lo + g.tw, g.W), y_lo = ty * g.th, y_hi = min(y_lo + g.th, g.H);
    const int gpr = max((x_hi - x_lo + 3) / 4, 0), total = max(y_hi - y_lo, 0) * gpr;
    const uint32_t gmagic = (uint32_t)(((1ull << 32) + max(gpr, 1) - 1) / max(gpr, 1));
    // RGB2LAB of gamma-table values (OpenCV's RGB2Lab_b): the matrix comes from scalar registers, products of 12-bit
    // values by 12-bit coefficients through the 24-bit multiplier
    const int f0 = T->fwd[0], f1 = T->fwd[1], f2 = T->fwd[2], f3 = T->fwd[3], f4 = T->fwd[4], f5 = T->fwd[5], f6 = T->fwd[6],
              f7 = T->fwd[7], f8 = T->fwd[8];
    auto lab_of = [&](uint32_t Rg, uint32_t Gg, uint32_t Bg, uint32_t &L, uint32_t &a, uint32_t &bb) {
        const uint32_t ix = (__umul24(Rg, f0) + __umul24(Gg, f1) + __umul24(Bg, f2) + 2048u) >> 12;
        const uint32_t iy = (__umul24(Rg, f3) + __umul24(Gg, f4) + __umul24(Bg, f5) + 2048u) >> 12;
        const uint32_t iz = (__umul24(Rg, f6) + __umul24(Gg, f7) + __umul24(Bg, f8) + 2048u) >> 12;
        const int fX = s_cbrt[ix], fY = s_cbrt[iy], fZ = s_cbrt[iz];
        // saturate_cast<uchar>(x >> 15) as clamp-then-shift: written shift-then-clamp, two adjacent values become one
        // v_ashr_pk_u8_i32 (new on gfx950), which writes the low 16 bits of its destination only while this toolchain uses the
        // register as a zero-extended word (profiles/microbench/ashr_pk.hip; round 3: a and b of every fourth pixel came out
        // wrong); the clamp keeps x in [0, 2^23), so a logical shift finishes the job
        auto sat15 = [](int x) { return (uint32_t)min(max(x, 0), (256 << 15) - 1) >> 15; };
        L = sat15(__mul24(Lscale, fY) + (Lshift + (1 << 14)));
        a = sat15(__mul24(500, fX - fY) + (128 * (1 << 15) + (1 << 14)));
        bb = sat15(__mul24(200, fY - fZ) + (128 * (1 << 15) + (1 << 14)));
    };
    const bool fastpath = SRC == 1 && S.allfast();  // block-uniform
    auto group_at = [&](int gi, int &n, int &p) {
        const int row = gpr == 1 ? gi : (int)__umulhi((uint32_t)gi, gmagic), xg = gi - row * gpr;  // gi / gpr
        const int x0 = x_lo + 4 * xg;
        n = min(4, x_hi - x0);
        p = (y_lo + row) * g.W + x0;
    };
    // fast path: the next group's loads are issued before this group's arithmetic
    uint32_t w_nx[3] = {0, 0, 0};
    double tv_nx[4] = {1.0, 1.0, 1.0, 1.0};
    bool have_nx = false;
    if (SRC == 1 && fastpath && tid < total) {
        int n0, p0;
        group_at(tid, n0, p0);
        have_nx = n0 == 4;
        if (have_nx) R.load_four(p0, w_nx, tv_nx);
    }
    for (int gi = tid; gi < total; gi += BS) {
        int n, p;
        group_at(gi, n, p);
        const bool have = have_nx;
        const uint32_t w[3] = {w_nx[0], w_nx[1], w_nx[2]};
        const double tv[4] = {tv_nx[0], tv_nx[1], tv_nx[2], tv_nx[3]};
        have_nx = false;
        if (SRC == 1 && fastpath && gi + BS < total) {
            int nn, pn;
            group_at(gi + BS, nn, pn);
            have_nx = nn == 4;
            if (have_nx) R.load_four(pn, w_nx, tv_nx);
        }
        if (SRC == 1 && have) {
            // whole groups of a recomputed image: straight-line restore (restore.h four_raw), packed stretch
            float v[3][4];
            R.four_raw(w, tv, v);
            uint32_t code[3][4];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float r[4] = {clip01(v[c][0]), clip01(v[c][1]), clip01(v[c][2]), clip01(v[c][3])};
                S.codes4(r, c, code[c]);
            }
            uint32_t L[4], a[4], bb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) lab_of(s_gamma[code[0][i]], s_gamma[code[1][i]], s_gamma[code[2][i]], L[i], a[i], bb[i]);
            uint32_t cl = L[0], cn = 1;  // runs of equal L inside the group: one LDS atomic per run
#pragma unroll
            for (int i = 1; i < 4; ++i) {
                if (L[i] == cl) {
                    ++cn;
                } else {
                    atomicAdd(&h[w4][cl], cn);
                    cl = L[i];
                    cn = 1;
                }
            }
            atomicAdd(&h[w4][cl], cn);
            u32_unaligned *wd = reinterpret_cast<u32_unaligned *>(labimg + (size_t)p * 3);
            wd[0] = L[0] | (a[0] << 8) | (bb[0] << 16) | (L[1] << 24);
            wd[1] = a[1] | (bb[1] << 8) | (L[2] << 16) | (a[2] << 24);
            wd[2] = bb[2] | (L[3] << 8) | (a[3] << 16) | (bb[3] << 24);
            continue;
        }
        float v0[4], v1[4], v2[4];
        if (SRC == 2) {
            const Px4 q = n == 4 ? load_px4_any(img8 + (size_t)p * 3) : load_px4(img8 + (size_t)p * 3, n, false);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v0[i] = (float)q.r[i];
                v1[i] = (float)q.g[i];
                v2[i] = (float)q.b[i];
            }
        } else if (SRC == 1) {
            float r[3][4];
            R.four(p, n, r);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v0[i] = r[0][i];
                v1[i] = r[1][i];
                v2[i] = r[2][i];
            }
        } else if (n == 4) {
            *reinterpret_cast<uint4 *>(v0) = *reinterpret_cast<const u128_unaligned *>(r0 + p);
            *reinterpret_cast<uint4 *>(v1) = *reinterpret_cast<const u128_unaligned *>(r1 + p);
            *reinterpret_cast<uint4 *>(v2) = *reinterpret_cast<const u128_unaligned *>(r2 + p);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v0[i] = i < n ? r0[p + i] : 0.f;
                v1[i] = i < n ? r1[p + i] : 0.f;
                v2[i] = i < n ? r2[p + i] : 0.f;
            }
        }
        uint32_t L[4], a[4], bb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) to_lab(v0[i], v1[i], v2[i], L[i], a[i], bb[i]);
        uint32_t cl = L[0], cn = 1;  // runs of equal L inside the group: one LDS atomic per run
#pragma unroll
        for (int i = 1; i < 4; ++i) {
            if (i >= n) break;
            if (L[i] == cl) {
                ++cn;
            } else {
                atomicAdd(&h[w4][cl], cn);
                cl = L[i];
                cn = 1;
            }
        }
        atomicAdd(&h[w4][cl], cn);
        uint8_t *o = labimg + (size_t)p * 3;
        if (n == 4) {
            u32_unaligned *wd = reinterpret_cast<u32_unaligned *>(o);
            wd[0] = L[0] | (a[0] << 8) | (bb[0] << 16) | (L[1] << 24);
            wd[1] = a[1] | (bb[1] << 8) | (L[2] << 16) | (a[2] << 24);
            wd[2] = bb[2] | (L[3] << 8) | (a[3] << 16) | (bb[3] << 24);
        } else {
            for (int i = 0; i < n; ++i) {
                o[3 * i] = (uint8_t)L[i];
                o[3 * i + 1] = (uint8_t)a[i];
                o[3 * i + 2] = (uint8_t)bb[i];
            }
        }
    }
    // the part of the tile that hangs over the image edge (CLAHE pads by reflection): histogram only
    if (x_lo + g.tw > g.W || y_lo + g.th > g.H) {
        for (int i = tid; i < area; i += BS) {
            const int ey = ty * g.th + i / g.tw, ex = txi * g.tw + i % g.tw;
            if (ey < g.H && ex < g.W) continue;
            const int p = reflect101(ey, g.H) * g.W + reflect101(ex, g.W);
            uint32_t L, a, bb;
            float e0, e1, e2;
            if (SRC == 2) {
                e0 = (float)img8[(size_t)p * 3];
                e1 = (float)img8[(size_t)p * 3 + 1];
                e2 = (float)img8[(size_t)p * 3 + 2];
            } else if (SRC == 1) {
                R.pixel(p, e0, e1, e2);
            } else {
                e0 = r0[p];
                e1 = r1[p];
                e2 = r2[p];
            }
            to_lab(e0, e1, e2, L, a, bb);
            atomicAdd(&h[w][L], 1u);
        }
    }
    };  // pixels
    if (SRC == 1 && src.t32) {
        RestoreImgT<2, true> R;
        pixels(R);
    } else {
        RestoreImgT<2> R;
        pixels(R);
    }
    __syncthreads();
    if (BS > 256 && tid >= 256) return;  // the first four wavefronts finish (ended wavefronts leave the barrier count)
    uint32_t c = h[0][tid] + h[1][tid] + h[2][tid] + h[3][tid];
    if (g.clip > 0) {
        const uint32_t over = c > (uint32_t)g.clip ? c - g.clip : 0;
        if (over) c = g.clip;
        const uint32_t tot = block_incl_scan_256(over, wsum);
        __syncthreads();
        if (tid == 255) wsum[0] = tot;
        __syncthreads();
        const uint32_t clipped = wsum[0];
        __syncthreads();
        const uint32_t batch = clipped / 256, residual = clipped - batch * 256;
        c += batch;
        if (residual) {
            const uint32_t step = max(256u / residual, 1u);
            if (tid % step == 0 && tid / step < residual) c += 1;
        }
    }
    const uint32_t sum = block_incl_scan_256(c, wsum);
    lut[((size_t)b * g.tx * g.ty + tile) * 256 + tid] = sat_u8(__float2int_rn((float)sum * g.lutScale));
}

__device__ __forceinline__ float pow_f32(float x, float e) { return (float)pow((double)x, (double)e); }

// value of the float image for an 8-bit code v after LAB2RGB: v/255 [-> gamma]; and its output quantisation
__device__ __forceinline__ float final_value(int v, int gamma_mode, float gexp)
{
    float y = (float)v / 255.0f;
    if (gamma_mode == 1) y = pow_f32(y, gexp);
    else if (gamma_mode == 2) y = clip01(pow_f32(y, gexp));
    return y;
}

__device__ __forceinline__ int ab_to_xz(int i)
{
    // abToXZ_b of OpenCV's Lab2RGBinteger, evaluated instead of tabulated:
    //     i <= 3390 ? i * 108 / 841 - (1 << 14) * 16 / 116 * 108 / 841 : i * i / (1 << 14) * i / (1 << 14)
    // (C integer division truncates toward 0; the constant is 290).  i = ify +- the a/b term lies in [-8145, 26868].
    // Division-free (round 3; the signed constant divisions were a third of the blend kernel's instructions): truncation =
    // floor after adding 840 to a negative numerator; floor(n / 841) = umulhi(n, ceil(2^32 / 841)) exactly for n < 1.49e6,
    // and n = 108 i + 841 * 1100 + [840] stays in (0, 1.3e6) on the branch that uses it; the cubic branch only sees
    // positive i < 2^15, so both products fit the 24-bit multiplier and the shifts are the divisions.  Checked against the
    // C expression for every i of the domain (tests/test_cabi.py).
    const int t = __mul24(i, 108);
    const uint32_t tp = (uint32_t)(t + ((t >> 31) & 840) + 841 * 1100);
    const int lin = (int)__umulhi(tp, 5106980u) - (1100 + 290);
    const uint32_t sq = __umul24((uint32_t)i, (uint32_t)i) >> 14;
    const int cub = (int)(__umul24(sq, (uint32_t)i) >> 14);
    return i <= 3390 ? lin : cub;
}
// the cubic branch alone (the caller has checked i > 3390)
__device__ __forceinline__ int ab_to_xz_cubic(int i)
{
    const uint32_t sq = __umul24((uint32_t)i, (uint32_t)i) >> 14;
    return (int)(__umul24(sq, (uint32_t)i) >> 14);
}

// One block per (interpolation cell, row chunk) and image.  Between the centres of four neighbouring tiles the four
// tile LUTs a pixel blends are fixed, so the block copies them (a 4x4 window of tiles around the cell, which also
// covers a boundary pixel that float rounding puts into the next cell) into LDS once; the four per-pixel LUT gathers
// then hit LDS instead of global memory, where they made the kernel address-unit bound.
// grid ((tx+1)*(ty+1)*nchunk, B), block 256
// CODES (code-domain strategies, k_codes.hip): the RGB codes go out through per-(image, channel) final LUTs (fin_code /
// fin_val) or are stored raw with their per-channel histogram (codes_out / hist: the rest of the chain needs percentiles).
// F32OUT (not CODES): the float image is wanted too, so the 8-bit code after the inverse gamma is kept; otherwise the
// inverse-gamma table and the final quantisation are composed into one 4096-entry byte table per block.
// Round 3 (97 -> ~70 VALU instructions per pixel): division-free ab_to_xz, the a / b offsets folded into two biased copies
// of the L table (one 16-byte LDS read per pixel), the matrix from scalar registers through the 24-bit multiplier, the
// bilinear blend as packed float32 pairs (same operations, same order), the composed final table.
template <bool CODES, bool F32OUT>
__global__ void __launch_bounds__(256) k_clahe_apply_out(const LabTables *__restrict__ T, const uint8_t *__restrict__ lab,
                                                         const uint8_t *__restrict__ lut, ClaheGeom g, int nchunk,
                                                         int gamma_mode, float gexp, const uint8_t *__restrict__ fin_code,
                                                         const float *__restrict__ fin_val, uint8_t *__restrict__ out_u8,
                                                         float *__restrict__ out_f32, uint8_t *__restrict__ codes_out,
                                                         uint32_t *__restrict__ hist)
{
    constexpr int NF = CODES ? 768 : 256;
    constexpr bool TWOSTEP = CODES || F32OUT;  // inverse gamma -> code -> final value, as two lookups
    constexpr int BASE = 1 << 14;
    // per L: the two abToXZ arguments' L parts and the L terms of the three matrix rows with the rounding constant, so a
    // pixel's Y costs no instruction.  16-byte entries + a separate word array: entry L sits in bank quad L mod 8 (32-byte
    // entries used four of the eight quads)
    __shared__ __attribute__((aligned(16))) int s_l[256][4];   // {ify - 128*BASE/500, ify + 128*BASE/200 - 1, m1*y + 2^13, m4*y + 2^13}
    __shared__ int s_l2[256];                                   // m7*y + 2^13
    __shared__ uint8_t s_invgamma[TWOSTEP ? 4096 : 1];
    __shared__ uint8_t s_fin[TWOSTEP ? 1 : 4096];  // s_fu[invgamma[.]]
    __shared__ float s_ff[NF];
    __shared__ uint8_t s_fu[NF];
    __shared__ uint32_t s_h[CODES ? 4 * 768 : 1];
    // the 4 x 4 window of tile LUTs, the four columns of a window row interleaved: word [wy][v] = {lut(wy,0)[v], .., lut(wy,3)[v]},
    // so a pixel reads one word per window row (two, not four byte gathers) and a byte permute picks its two columns
    __shared__ uint32_t s_lut4[4][256];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int cell = blockIdx.x / nchunk, chunk = blockIdx.x - cell * nchunk;
    const int cyi = cell / (g.tx + 1), cxi = cell - cyi * (g.tx + 1);
    // nominal pixel rectangle of the cell: tile centres are at (i + 1/2) * tile size
    const int cx0 = max(cxi * g.tw - g.tw / 2, 0), cx1 = min((cxi + 1) * g.tw - g.tw / 2, g.W);
    const int ry0 = max(cyi * g.th - g.th / 2, 0), ry1 = min((cyi + 1) * g.th - g.th / 2, g.H);
    const int rows_per = (ry1 - ry0 + nchunk - 1) / nchunk;
    const int cy0 = ry0 + chunk * rows_per, cy1 = min(cy0 + rows_per, ry1);
    if (cx0 >= cx1 || cy0 >= cy1) return;
    const int wx0 = min(max(cxi - 2, 0), max(g.tx - 4, 0)), wy0 = min(max(cyi - 2, 0), max(g.ty - 4, 0));
    const int m0 = T->inv[0], m1 = T->inv[1], m2 = T->inv[2], m3 = T->inv[3], m4 = T->inv[4], m5 = T->inv[5], m6 = T->inv[6],
              m7 = T->inv[7], m8 = T->inv[8];  // scalar registers
    {
        const int yy = T->ltoyf[2 * tid], ify = T->ltoyf[2 * tid + 1];
        *reinterpret_cast<int4 *>(&s_l[tid][0]) = make_int4(ify - 128 * BASE / 500, ify + (128 * BASE / 200 - 1), m1 * yy + (1 << 13), m4 * yy + (1 << 13));
        s_l2[tid] = m7 * yy + (1 << 13);
    }
    if (TWOSTEP)
        for (int i = tid; i < 4096; i += 256) s_invgamma[i] = T->invgamma[i];
    if (CODES) {
        for (int i = tid; i < 768; i += 256) {
            s_fu[i] = fin_code ? fin_code[(size_t)b * 768 + i] : (uint8_t)0;
            s_ff[i] = fin_val ? fin_val[(size_t)b * 768 + i] : 0.f;
        }
        if (hist)
            for (int i = tid; i < 4 * 768; i += 256) s_h[i] = 0;
    } else {
        const float y = final_value(tid, gamma_mode, gexp);
        s_ff[tid] = y;
        s_fu[tid] = (uint8_t)quant_u8(y);
    }
    const uint8_t *Lt = lut + (size_t)b * g.tx * g.ty * 256;
    {  // thread (wy, q): values 4q .. 4q+3 of the four tile LUTs of window row wy, transposed into four words
        const int wy = tid >> 6, q = tid & 63;
        const int tyy = min(wy0 + wy, g.ty - 1);
        uint32_t d[4];
#pragma unroll
        for (int wx = 0; wx < 4; ++wx)
            d[wx] = reinterpret_cast<const uint32_t *>(Lt + (size_t)(tyy * g.tx + min(wx0 + wx, g.tx - 1)) * 256)[q];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            s_lut4[wy][4 * q + k] = ((d[0] >> (8 * k)) & 255u) | (((d[1] >> (8 * k)) & 255u) << 8) | (((d[2] >> (8 * k)) & 255u) << 16) |
                                    (((d[3] >> (8 * k)) & 255u) << 24);
    }
    __syncthreads();
    if (!TWOSTEP) {
        for (int i = tid; i < 1024; i += 256) {  // four entries per step: dword in, dword out
            const uint32_t q = reinterpret_cast<const uint32_t *>(T->invgamma)[i];
            reinterpret_cast<uint32_t *>(s_fin)[i] = (uint32_t)s_fu[q & 255] | ((uint32_t)s_fu[(q >> 8) & 255] << 8) |
                                                     ((uint32_t)s_fu[(q >> 16) & 255] << 16) | ((uint32_t)s_fu[q >> 24] << 24);
        }
        __syncthreads();
    }
    const float inv_tw = 1.0f / (float)g.tw, inv_th = 1.0f / (float)g.th;
    const int gpr = (cx1 - cx0 + 3) / 4, total = (cy1 - cy0) * gpr;  // 4-pixel groups per row, in the block
    const uint32_t gmagic = (uint32_t)(((1ull << 32) + gpr - 1) / gpr);
    struct Cols {  // per 4-pixel group: LUT window columns and blend weights of its pixels (they depend on x only)
        uint32_t sel[4];   // v_perm_b32 selector: {column c1 of row 1, c1 of row 2, c2 of row 1, c2 of row 2}
        float xa[4], xa1[4];
    };
    auto cols_of = [&](int x0) {
        Cols C;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x = min(x0 + i, g.W - 1);
            const float txf = (float)x * inv_tw - 0.5f;
            int tx1 = (int)floorf(txf);
            int tx2 = tx1 + 1;
            C.xa[i] = txf - (float)tx1;
            C.xa1[i] = 1.0f - C.xa[i];
            tx1 = max(tx1, 0);
            tx2 = min(tx2, g.tx - 1);
            const uint32_t c1 = (uint32_t)min(max(tx1 - wx0, 0), 3), c2 = (uint32_t)min(max(tx2 - wx0, 0), 3);
            C.sel[i] = c1 | ((4u + c1) << 8) | (c2 << 16) | ((4u + c2) << 24);  // bytes 0..3: second source, 4..7: first source
        }
        return C;
    };
    // the three LAB words of a whole group (issued one row ahead by the row-walking loop below)
    auto fetch = [&](int row, int xg, uint32_t (&w)[3]) {
        const int y = cy0 + row, x0 = cx0 + 4 * xg;
        if (row < cy1 - cy0 && x0 + 4 <= cx1) {
            const u32_unaligned *q = reinterpret_cast<const u32_unaligned *>(lab + (((size_t)b * g.H + y) * g.W + x0) * 3);
            w[0] = q[0]; w[1] = q[1]; w[2] = q[2];
        }
    };
    auto group = [&](int row, int xg, const Cols &C, const uint32_t (&pre)[3], bool has_pre) {
        const int y = cy0 + row, x0 = cx0 + 4 * xg, n = min(4, cx1 - x0);
        const size_t pix = ((size_t)b * g.H + y) * g.W + x0;
        Px4 in4;  // r,g,b fields hold L,a,b
        if (n == 4) {
            uint32_t w0 = pre[0], w1 = pre[1], w2 = pre[2];
            if (!has_pre) {
                const u32_unaligned *w = reinterpret_cast<const u32_unaligned *>(lab + pix * 3);
                w0 = w[0]; w1 = w[1]; w2 = w[2];
            }
            in4.r[0] = w0 & 255; in4.g[0] = (w0 >> 8) & 255; in4.b[0] = (w0 >> 16) & 255;
            in4.r[1] = w0 >> 24; in4.g[1] = w1 & 255; in4.b[1] = (w1 >> 8) & 255;
            in4.r[2] = (w1 >> 16) & 255; in4.g[2] = w1 >> 24; in4.b[2] = w2 & 255;
            in4.r[3] = (w2 >> 8) & 255; in4.g[3] = (w2 >> 16) & 255; in4.b[3] = w2 >> 24;
        } else {
            in4 = load_px4(lab + pix * 3, n, false);
        }
        const float tyf = (float)y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf);
        int ty2 = ty1 + 1;
        const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        ty1 = max(ty1, 0);
        ty2 = min(ty2, g.ty - 1);
        const uint32_t *row1 = s_lut4[min(max(ty1 - wy0, 0), 3)], *row2 = s_lut4[min(max(ty2 - wy0, 0), 3)];
        const f32x2 yw = {ya1, ya};
        uint32_t o0[4], o1[4], o2[4];
        int ix[4], iz[4], yt0[4], yt1[4], yt2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = in4.r[i], aa = in4.g[i], bb = in4.b[i];
            // res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya, the two rows as one packed pair
            const uint32_t pk = __builtin_amdgcn_perm(row2[v], row1[v], C.sel[i]);  // {l11, l21, l12, l22}
            const f32x2 l1 = {(float)(pk & 255u), (float)((pk >> 8) & 255u)};
            const f32x2 l2 = {(float)((pk >> 16) & 255u), (float)(pk >> 24)};
            const f32x2 rw = (l1 * C.xa1[i] + l2 * C.xa[i]) * yw;
            const float res = rw.x + rw.y;
            const int LL = min(max(__float2int_rn(res), 0), 255);
            // LAB2RGB (Lab2RGBinteger): arguments of abToXZ for X and Z
            const int4 ly = *reinterpret_cast<const int4 *>(&s_l[LL][0]);
            yt0[i] = ly.z;
            yt1[i] = ly.w;
            yt2[i] = s_l2[LL];
            ix[i] = ly.x + (int)(((uint32_t)__umul24(aa, 5 * 53687) + (1u << 7)) >> 13);
            iz[i] = ly.y - (int)(((uint32_t)__umul24(bb, 41943) + (1u << 4)) >> 9);
        }
        // Almost every pixel has both arguments on the cubic branch (the linear one is for fX, fZ below 0.207: very dark or
        // strongly saturated colours): one test per group, the general form only for groups that need it
        int xx[4], zz[4];
        if (min(min(min(ix[0], ix[1]), min(ix[2], ix[3])), min(min(iz[0], iz[1]), min(iz[2], iz[3]))) > 3390) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xx[i] = ab_to_xz_cubic(ix[i]);
                zz[i] = ab_to_xz_cubic(iz[i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xx[i] = ab_to_xz(ix[i]);
                zz[i] = ab_to_xz(iz[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ro = min(max((__mul24(m0, xx[i]) + __mul24(m2, zz[i]) + yt0[i]) >> 14, 0), 4095);
            const int go = min(max((__mul24(m3, xx[i]) + __mul24(m5, zz[i]) + yt1[i]) >> 14, 0), 4095);
            const int bo = min(max((__mul24(m6, xx[i]) + __mul24(m8, zz[i]) + yt2[i]) >> 14, 0), 4095);
            if (TWOSTEP) {
                o0[i] = s_invgamma[ro];
                o1[i] = s_invgamma[go];
                o2[i] = s_invgamma[bo];
            } else {
                o0[i] = s_fin[ro];
                o1[i] = s_fin[go];
                o2[i] = s_fin[bo];
            }
        }
        const size_t o = pix * 3;
        if (CODES) {
            if (codes_out) {
                if (n == 4) {
                    u32_unaligned *w = reinterpret_cast<u32_unaligned *>(codes_out + o);
                    w[0] = o0[0] | (o1[0] << 8) | (o2[0] << 16) | (o0[1] << 24);
                    w[1] = o1[1] | (o2[1] << 8) | (o0[2] << 16) | (o1[2] << 24);
                    w[2] = o2[2] | (o0[3] << 8) | (o1[3] << 16) | (o2[3] << 24);
                } else {
                    store_px4(codes_out + o, o0, o1, o2, n, false);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o1[i] += 256;
                o2[i] += 512;
            }
            if (hist) {
                uint32_t *hw = s_h + (tid >> 6) * 768;
                for (int i = 0; i < n; ++i) {
                    atomicAdd(&hw[o0[i]], 1u);
                    atomicAdd(&hw[o1[i]], 1u);
                    atomicAdd(&hw[o2[i]], 1u);
                }
            }
        }
        if (TWOSTEP && out_f32)
            for (int i = 0; i < n; ++i) {
                out_f32[o + 3 * i] = s_ff[o0[i]];
                out_f32[o + 3 * i + 1] = s_ff[o1[i]];
                out_f32[o + 3 * i + 2] = s_ff[o2[i]];
            }
        if (out_u8) {
            if (TWOSTEP) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    o0[i] = s_fu[o0[i]];
                    o1[i] = s_fu[o1[i]];
                    o2[i] = s_fu[o2[i]];
                }
            }
            if (n == 4) {
                u32_unaligned *w = reinterpret_cast<u32_unaligned *>(out_u8 + o);
                w[0] = o0[0] | (o1[0] << 8) | (o2[0] << 16) | (o0[1] << 24);
                w[1] = o1[1] | (o2[1] << 8) | (o0[2] << 16) | (o1[2] << 24);
                w[2] = o2[2] | (o0[3] << 8) | (o1[3] << 16) | (o2[3] << 24);
            } else {
                store_px4(out_u8 + o, o0, o1, o2, n, false);
            }
        }
    };
    const int rows = cy1 - cy0, rstep = 256 / max(gpr, 1);
    if (gpr <= 256 && rstep * gpr * 8 >= 256 * 7) {
        // a thread keeps its 4-pixel column group and walks down the rows: the column terms are computed once
        if (tid < rstep * gpr) {
            const int xg = tid % gpr;
            const Cols C = cols_of(cx0 + 4 * xg);
#if UWIE_CLAHE_PREFETCH  // measured at 4K x 64: 1.34 ms against 1.27 without (three more live registers cost a wavefront per SIMD)
            uint32_t nx[3] = {0, 0, 0};
            fetch(tid / gpr, xg, nx);
            for (int row = tid / gpr; row < rows; row += rstep) {
                const uint32_t cur[3] = {nx[0], nx[1], nx[2]};
                fetch(row + rstep, xg, nx);  // the next row's words travel while this row is blended
                group(row, xg, C, cur, true);
            }
#else
            const uint32_t none[3] = {0, 0, 0};
            for (int row = tid / gpr; row < rows; row += rstep) group(row, xg, C, none, false);
#endif
        }
    } else {
        const uint32_t none[3] = {0, 0, 0};
        for (int gi = tid; gi < total; gi += 256) {
            const int row = gpr == 1 ? gi : (int)__umulhi((uint32_t)gi, gmagic), xg = gi - row * gpr;  // gi / gpr (gi < 2^32 / gpr)
            group(row, xg, cols_of(cx0 + 4 * xg), none, false);
        }
    }
    if (CODES && hist) {
        __syncthreads();
        for (int i = tid; i < 768; i += 256) {
            const uint32_t c = s_h[i] + s_h[768 + i] + s_h[2 * 768 + i] + s_h[3 * 768 + i];
            if (c) atomicAdd(&hist[(size_t)b * 768 + i], c);
        }
    }
}

// CLAHE-free tail: planar restored -> stretch [-> stretch] [-> gamma] -> HWC outputs.  grid (n, B)
__global__ void __launch_bounds__(256) k_stretch_out(const float *__restrict__ planar, const float *__restrict__ pct,
                                                     int pct_stride, float eps, int two, int npx, int gamma_mode, float gexp,
                                                     uint8_t *__restrict__ out_u8, float *__restrict__ out_f32)
{
    const int b = blockIdx.y;
    Stretch S;
    S.load(pct, b, pct_stride, eps, two);
    const float *r = planar + (size_t)b * 3 * npx;
    const bool aligned = (npx & 3) == 0;  // plane rows of four floats are 16-byte aligned, output groups dword aligned
    for (int p = (blockIdx.x * 256 + threadIdx.x) * 4; p < npx; p += gridDim.x * 1024) {  // four pixels per thread
        const int n = min(4, npx - p);
        float v[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float *src = r + (size_t)c * npx + p;
            if (aligned && n == 4) {
                const float4 q = *reinterpret_cast<const float4 *>(src);
                v[c][0] = q.x; v[c][1] = q.y; v[c][2] = q.z; v[c][3] = q.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[c][i] = i < n ? src[i] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float y = S.apply_exact(v[c][i], c);
                if (gamma_mode == 1) y = pow_f32(y, gexp);
                else if (gamma_mode == 2) y = clip01(pow_f32(y, gexp));
                v[c][i] = y;
            }
        }
        const size_t o = ((size_t)b * npx + p) * 3;
        if (out_f32) {
            float *dst = out_f32 + o;
            if (aligned && n == 4) {
                float4 *d4 = reinterpret_cast<float4 *>(dst);
                d4[0] = make_float4(v[0][0], v[1][0], v[2][0], v[0][1]);
                d4[1] = make_float4(v[1][1], v[2][1], v[0][2], v[1][2]);
                d4[2] = make_float4(v[2][2], v[0][3], v[1][3], v[2][3]);
            } else {
                for (int i = 0; i < n; ++i) {
                    dst[3 * i] = v[0][i];
                    dst[3 * i + 1] = v[1][i];
                    dst[3 * i + 2] = v[2][i];
                }
            }
        }
        if (out_u8) {
            uint32_t q0[4], q1[4], q2[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                q0[i] = quant_u8(v[0][i]);
                q1[i] = quant_u8(v[1][i]);
                q2[i] = quant_u8(v[2][i]);
            }
            store_px4(out_u8 + o, q0, q1, q2, n, aligned);
        }
    }
}

// ---- ES surface, float64
// recovered = clip((img - A) / t + A, 0, 1): float32 difference, float64 quotient/sum/result (ES:247-248)
// (recovered = clip((img - A) / t + A, 0, 1) in float64, ES:247-248: k_restore_planar_hist<..., double>)
// pct: [B][3][2] float64 = lo, hi.  grid (n, B), four pixels per thread.  SRC: the recovered image is recomputed from S
// instead of read from planar.  The three stretch quotients of a channel share the denominator: the float64 division's
// own sequence with the reciprocal kept (as restore.h does for the divisor t), true division outside its safe range.
struct StretchDiv64 {
    double den, y;
    bool fast;
    __device__ __forceinline__ void set(double d)
    {
        den = d;
        fast = d >= 0x1p-100 && d <= 0x1p100;
        double r = __builtin_amdgcn_rcp(d);
        r = fma(r, fma(-d, r, 1.0), r);
        y = fma(r, fma(-d, r, 1.0), r);
    }
    __device__ __forceinline__ double quot(double n) const
    {
        const double m = fabs(n);
        if (!(fast && (m == 0.0 || (m >= 0x1p-500 && m <= 0x1p500)))) return n / den;
        const double q0 = n * y;
        return fma(fma(-den, q0, n), y, q0);
    }
};
template <bool SRC>
__global__ void __launch_bounds__(256) k_stretch64_out(const double *__restrict__ planar, RestoreSrc S,
                                                       const double *__restrict__ pct, int npx, int apply_gamma, double gexp,
                                                       uint8_t *__restrict__ out_u8, float *__restrict__ out_f32,
                                                       double *__restrict__ out_f64)
{
    const int b = blockIdx.y;
    double lo[3];
    StretchDiv64 den[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        lo[c] = pct[(b * 3 + c) * 2];
        den[c].set((pct[(b * 3 + c) * 2 + 1] - lo[c]) + 1e-10);
    }
    RestoreImg R;
    if (SRC) R.init(S, b, (size_t)npx);
    const double *r = planar + (size_t)b * 3 * npx;
    const bool aligned = (npx & 3) == 0;
    for (int p = (blockIdx.x * 256 + threadIdx.x) * 4; p < npx; p += gridDim.x * 1024) {
        const int n = min(4, npx - p);
        double v[3][4];
        if (SRC) {
            R.four64(p, n, v);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[c][i] = i < n ? r[(size_t)c * npx + p + i] : 0.0;
        }
        uint32_t q[3][4];
        double f[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double y = fmin(fmax(den[c].quot(v[c][i] - lo[c]), 0.0), 1.0);
                if (apply_gamma) y = fmin(fmax(pow(y, gexp), 0.0), 1.0);
                q[c][i] = (uint32_t)((int)(y * 255.0) & 0xff);
                f[c][i] = y;
            }
        const size_t o = ((size_t)b * npx + p) * 3;
        if (out_u8) store_px4(out_u8 + o, q[0], q[1], q[2], n, aligned);
        if (out_f32)
            for (int i = 0; i < n; ++i) {
                out_f32[o + 3 * i] = (float)f[0][i];
                out_f32[o + 3 * i + 1] = (float)f[1][i];
                out_f32[o + 3 * i + 2] = (float)f[2][i];
            }
        if (out_f64)  // the reference's own result (enhancement_strategies.py:247,269-270,284-285: float64)
            for (int i = 0; i < n; ++i) {
                out_f64[o + 3 * i] = f[0][i];
                out_f64[o + 3 * i + 1] = f[1][i];
                out_f64[o + 3 * i + 2] = f[2][i];
            }
    }
}

ClaheGeom make_geom(Shape s, double clip, int tx, int ty)
{
    ClaheGeom g;
    g.H = s.H; g.W = s.W; g.tx = tx; g.ty = ty;
    int We = s.W, He = s.H;
    if (s.W % tx != 0 || s.H % ty != 0) {
        We = s.W + (tx - s.W % tx);
        He = s.H + (ty - s.H % ty);
    }
    g.tw = We / tx;
    g.th = He / ty;
    const int area = g.tw * g.th;
    g.lutScale = (float)(256 - 1) / (float)area;
    g.clip = 0;
    if (clip > 0.0) {
        g.clip = (int)(clip * area / 256);
        if (g.clip < 1) g.clip = 1;
    }
    return g;
}

float gamma_exponent(int mode, double g) { return mode == 1 ? (float)g : mode == 2 ? (float)(1.0 / g) : 1.0f; }

}  // namespace

int launch_restore_planar_hist(const uint8_t *d_in, const int32_t *d_kind, const float *d_A, const double *d_t, Shape s,
                               float *d_planar, uint32_t *d_ghist, hipStream_t st, bool linear, const uint32_t *d_only,
                               const SelectPlan *plan, int t32)
{
    // 24.6 KB of LDS per block: six blocks per CU, 1536 resident on the chip.  Enough blocks for several full rounds
    // (2048 blocks were 1.33 rounds: a third of the chip idle for half the kernel).
    int nblk = cdiv(12288, s.B);  // 4K x 64: 32 per frame 2.79 ms, 96: 2.52, 192: 2.43, 384: 2.47 (round 1)
    // (a block clears and flushes 6 K counters: small batches get fewer, longer blocks; 4K x 16: 768 per frame 0.62 ms, 192: 0.53)
    nblk = nblk < 16 ? 16 : nblk > 384 ? 384 : nblk;
    const int need = cdiv((long long)s.npx(), 1024);
    if (nblk > need) nblk = need;
    const RestoreSrc S{d_in, d_kind, d_A, d_t, t32};
    const dim3 grid(nblk, s.B);
    const auto k_restore_hist_collect = k_restore_planar_hist<true, true, 2>;  // (names as the profiler reports them)
    const auto k_restore_hist_collect4 = k_restore_planar_hist<true, true, 4>;
    const auto k_restore_hist_lin = k_restore_planar_hist<true, false>;
    const auto k_restore_hist_key = k_restore_planar_hist<false, false>;
    if (plan) {
        UWIE_REQUIRE(linear && d_ghist == plan->ghist, "restore: a selection plan goes with its own linear histogram");
        if (plan->nq <= 2)
            UWIE_LAUNCH(k_restore_hist_collect, grid, dim3(256), 0, st, S, (int)s.npx(), d_planar, d_ghist, d_only,
                        (LinState *)plan->lin, plan->lists, plan->cap);
        else
            UWIE_LAUNCH(k_restore_hist_collect4, grid, dim3(256), 0, st, S, (int)s.npx(), d_planar, d_ghist, d_only,
                        (LinState *)plan->lin, plan->lists, plan->cap);
    } else if (linear) {
        UWIE_LAUNCH(k_restore_hist_lin, grid, dim3(256), 0, st, S, (int)s.npx(), d_planar, d_ghist, d_only,
                    (LinState *)nullptr, (float *)nullptr, 0u);
    } else {
        UWIE_LAUNCH(k_restore_hist_key, grid, dim3(256), 0, st, S, (int)s.npx(), d_planar, d_ghist, d_only,
                    (LinState *)nullptr, (float *)nullptr, 0u);
    }
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

int launch_restore_rank(const RestoreSrc &src, Shape s, const SelectPlan &plan, hipStream_t st)
{
    UWIE_REQUIRE(!src.t32 && plan.nq <= 2 && plan.predicted && s.npx() >= 4, "restore_rank: float64 transmission, two predicted percentiles");
    // 28.8 KB of LDS per block: five blocks per CU, 1280 on the chip.  Four whole rounds of them per call (the histogram
    // sweep's 12288 blocks were 9.6 rounds, the last one 60 % full: 1.43 -> 1.37 ms at 4K x 64, A/B/C with 2560 / 3840 / 5120).
    int nblk = cdiv(5120, s.B);
    nblk = nblk < 16 ? 16 : nblk > 384 ? 384 : nblk;
    const int need = cdiv((long long)s.npx(), 1024);
    if (nblk > need) nblk = need;
    UWIE_LAUNCH(k_restore_rank, dim3(nblk, s.B), dim3(256), 0, st, src, (int)s.npx(), (LinState *)plan.lin, plan.lists,
                (uint32_t)kRankCapMul * plan.cap);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

int launch_recover64_planar_hist(const uint8_t *d_in, const float *d_A, const double *d_t, Shape s, double *d_planar,
                                 uint32_t *d_ghist, hipStream_t st, bool linear, const uint32_t *d_only, const SelectPlan *plan)
{
    int nblk = cdiv(12288, s.B);
    nblk = nblk < 16 ? 16 : nblk > 384 ? 384 : nblk;
    const int need = cdiv((long long)s.npx(), 1024);
    if (nblk > need) nblk = need;
    const RestoreSrc S{d_in, nullptr, d_A, d_t};
    const dim3 grid(nblk, s.B);
    const auto k_recover64_hist_collect = k_restore_planar_hist<true, true, 2, double>;  // (names as the profiler reports them)
    const auto k_recover64_hist_lin = k_restore_planar_hist<true, false, 2, double>;
    const auto k_recover64_hist_key = k_restore_planar_hist<false, false, 2, double>;
    if (plan) {
        UWIE_REQUIRE(linear && d_ghist == plan->ghist && plan->nq <= 2, "recover: a selection plan goes with its own linear histogram");
        UWIE_LAUNCH(k_recover64_hist_collect, grid, dim3(256), 0, st, S, (int)s.npx(), d_planar, d_ghist, d_only,
                    (LinState *)plan->lin, reinterpret_cast<double *>(plan->lists), plan->cap);
    } else if (linear) {
        UWIE_LAUNCH(k_recover64_hist_lin, grid, dim3(256), 0, st, S, (int)s.npx(), d_planar, d_ghist, d_only, (LinState *)nullptr,
                    (double *)nullptr, 0u);
    } else {
        UWIE_LAUNCH(k_recover64_hist_key, grid, dim3(256), 0, st, S, (int)s.npx(), d_planar, d_ghist, d_only, (LinState *)nullptr,
                    (double *)nullptr, 0u);
    }
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

int launch_tail_plain64(const double *d_planar, const double *d_pct, Shape s, int apply_gamma, double gamma,
                        uint8_t *d_out_u8, float *d_out_f32, hipStream_t st, const RestoreSrc *src, double *d_out_f64)
{
    const dim3 grid(grid_for(s.npx(), 4096), s.B);
    if (src)
        UWIE_LAUNCH(k_stretch64_out<true>, grid, dim3(256), 0, st, d_planar, *src, d_pct, (int)s.npx(), apply_gamma, 1.0 / gamma,
                    d_out_u8, d_out_f32, d_out_f64);
    else
        UWIE_LAUNCH(k_stretch64_out<false>, grid, dim3(256), 0, st, d_planar, RestoreSrc{}, d_pct, (int)s.npx(), apply_gamma,
                    1.0 / gamma, d_out_u8, d_out_f32, d_out_f64);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

size_t tail_ws_bytes(Shape s, int tx, int ty)
{
    Carver c(nullptr);
    c.take<uint8_t>((size_t)s.B * tx * ty * 256);
    c.take<uint8_t>((size_t)s.B * s.npx() * 3);
    return c.total();
}

int launch_tail_clahe(uwie_ctx *ctx, const float *d_planar, const float *d_pct, int pct_stride, float eps, int two,
                      Shape s, double clip, int tx, int ty, int gamma_mode, double gamma, uint8_t *d_out_u8,
                      float *d_out_f32, void *ws, hipStream_t st, const RestoreSrc *src)
{
    Carver c(ws);
    uint8_t *lut = c.take<uint8_t>((size_t)s.B * tx * ty * 256);
    uint8_t *lab = c.take<uint8_t>((size_t)s.B * s.npx() * 3);
    const ClaheGeom g = make_geom(s, clip, tx, ty);
    const auto k_stretch_lab_lut_wide = k_stretch_lab_lut<1, 1024>;  // (name as the profiler reports it)
    if (src && (long)tx * ty * s.B < 512 && (long)g.tw * g.th >= 8192)
        UWIE_LAUNCH(k_stretch_lab_lut_wide, dim3(tx * ty, s.B), dim3(1024), 0, st, ctx->d_lab, d_planar, *src, d_pct, pct_stride,
                    eps, two, (const uint8_t *)nullptr, g, lab, lut);
    else if (src)
        UWIE_LAUNCH(k_stretch_lab_lut<1>, dim3(tx * ty, s.B), dim3(256), 0, st, ctx->d_lab, d_planar, *src, d_pct, pct_stride,
                    eps, two, (const uint8_t *)nullptr, g, lab, lut);
    else
        UWIE_LAUNCH(k_stretch_lab_lut<0>, dim3(tx * ty, s.B), dim3(256), 0, st, ctx->d_lab, d_planar, RestoreSrc{}, d_pct,
                    pct_stride, eps, two, (const uint8_t *)nullptr, g, lab, lut);
    UWIE_LAUNCH_CHECK();
    // row chunks per interpolation cell: enough blocks to fill the chip, at least ~16 rows each
    const int cells = (tx + 1) * (ty + 1);
#ifndef UWIE_APPLY_TOTAL
#define UWIE_APPLY_TOTAL 25920
#endif
    int nchunk = cdiv(UWIE_APPLY_TOTAL, cells * s.B);  // ~14 rounds of the 1792 resident blocks (7 per CU by registers)
    nchunk = std::max(1, std::min(nchunk, std::max(1, g.th / 16)));
    const auto k_clahe_apply_u8 = k_clahe_apply_out<false, false>;  // (names as the profiler reports them)
    const auto k_clahe_apply_f32 = k_clahe_apply_out<false, true>;
    if (d_out_f32)
        UWIE_LAUNCH(k_clahe_apply_f32, dim3(cells * nchunk, s.B), dim3(256), 0, st, ctx->d_lab, lab, lut, g, nchunk, gamma_mode,
                    gamma_exponent(gamma_mode, gamma), (const uint8_t *)nullptr, (const float *)nullptr, d_out_u8, d_out_f32,
                    (uint8_t *)nullptr, (uint32_t *)nullptr);
    else
        UWIE_LAUNCH(k_clahe_apply_u8, dim3(cells * nchunk, s.B), dim3(256), 0, st, ctx->d_lab, lab, lut, g, nchunk, gamma_mode,
                    gamma_exponent(gamma_mode, gamma), (const uint8_t *)nullptr, (const float *)nullptr, d_out_u8, d_out_f32,
                    (uint8_t *)nullptr, (uint32_t *)nullptr);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

// code-domain strategies (k_codes.hip): CLAHE blend -> LAB2RGB -> RGB codes -> final per-channel LUTs, or raw codes + histogram
int launch_clahe_apply_codes(uwie_ctx *ctx, const uint8_t *d_lab, const uint8_t *d_tile_lut, Shape s, double clip, int tx,
                             int ty, const uint8_t *d_fin_code, const float *d_fin_val, uint8_t *d_out_u8, float *d_out_f32,
                             uint8_t *d_codes_out, uint32_t *d_hist, hipStream_t st)
{
    const ClaheGeom g = make_geom(s, clip, tx, ty);
    const int cells = (tx + 1) * (ty + 1);
    int nchunk = cdiv(25920, cells * s.B);
    nchunk = std::max(1, std::min(nchunk, std::max(1, g.th / 16)));
    const auto k_clahe_apply_codes = k_clahe_apply_out<true, false>;
    UWIE_LAUNCH(k_clahe_apply_codes, dim3(cells * nchunk, s.B), dim3(256), 0, st, ctx->d_lab, d_lab, d_tile_lut, g, nchunk, 0, 1.0f,
                d_fin_code, d_fin_val, d_out_u8, d_out_f32, d_codes_out, d_hist);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

// code-domain strategies (k_codes.hip): u8 frame -> per-(image, channel) code LUT -> RGB2LAB -> LAB bytes + tile LUTs
int launch_codes_lab_lut(uwie_ctx *ctx, const uint8_t *d_in, const uint8_t *d_code_lut, Shape s, double clip, int tx, int ty,
                         uint8_t *d_lab, uint8_t *d_tile_lut, hipStream_t st)
{
    const ClaheGeom g = make_geom(s, clip, tx, ty);
    RestoreSrc src{};
    src.in = d_in;
    UWIE_LAUNCH(k_stretch_lab_lut<2>, dim3(tx * ty, s.B), dim3(256), 0, st, ctx->d_lab, (const float *)nullptr, src,
                (const float *)nullptr, 0, 0.0f, 0, d_code_lut, g, d_lab, d_tile_lut);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

int launch_tail_plain(const float *d_planar, const float *d_pct, int pct_stride, float eps, int two, Shape s,
                      int gamma_mode, double gamma, uint8_t *d_out_u8, float *d_out_f32, hipStream_t st)
{
    UWIE_LAUNCH(k_stretch_out, dim3(grid_for(s.npx(), 4096), s.B), dim3(256), 0, st, d_planar, d_pct, pct_stride, eps, two,
                (int)s.npx(), gamma_mode, gamma_exponent(gamma_mode, gamma), d_out_u8, d_out_f32);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

}  // namespace uwie
