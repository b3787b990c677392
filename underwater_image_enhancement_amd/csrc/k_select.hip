// np.percentile(channel, q) (six_stadigy.py:196-197,216-217; enhancement_strategies.py:265-266) for float32
// channels (S6 surface) and float64 channels (ES surface).
//
// NumPy (2.2.6, method "linear") computes, in the dtype of the data:
//     qd = q / dtype(100);  vi = (n - 1) * qd;  prev = floor(vi);  t = vi - prev
//     a = sorted[prev], b = sorted[prev + 1];  r = a + (b - a) * t;  if t >= 0.5: r = b - (b - a) * (1 - t)
// The index arithmetic depends only on (n, q) and is done on the host in that dtype; the two order statistics per
// percentile are found EXACTLY on the device by an MSD radix select on the order-preserving integer image of the
// float bits (float32: 11+11+10 bits, float64: 5x11+9): per digit one LDS-privatised histogram sweep over the channel
// plane, then a tiny scan kernel that narrows every query to the bucket holding its rank.  Up to 8 ranks
// (4 percentiles) per channel are resolved in the same sweeps; queries sharing a prefix share a histogram ("group").
// A producer kernel may accumulate the first digit's histogram itself (k_fused.hip), saving one sweep.
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "devutil.h"
#include "restore.h"

namespace uwie {

namespace {

constexpr int kMaxRanks = 2 * kMaxPct;
constexpr int kBins = 2048;

template <typename V>
struct Traits;
template <>
struct Traits<float> {
    using K = uint32_t;
    static constexpr int NPASS = 3;
    __host__ __device__ static int shift(int p) { return p == 0 ? 21 : p == 1 ? 10 : 0; }
    __host__ __device__ static int bits(int p) { return p == 2 ? 10 : 11; }
    __device__ static K key(float v) { return f32_key(v); }
    __device__ static float value(K k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)); }
};
template <>
struct Traits<double> {
    using K = uint64_t;
    static constexpr int NPASS = 6;
    __host__ __device__ static int shift(int p) { return p == 5 ? 0 : 53 - 11 * p; }
    __host__ __device__ static int bits(int p) { return p == 5 ? 9 : 11; }
    __device__ static K key(double v) { return f64_key(v); }
    __device__ static double value(K k)
    {
        return __longlong_as_double((long long)(k ^ ((k >> 63) ? 0x8000000000000000ull : 0xffffffffffffffffull)));
    }
};

template <typename K>
struct SelState {                 // one per (image, channel)
    K prefix[kMaxRanks];          // key bits resolved so far (right aligned)
    K gprefix[kMaxRanks];         // distinct prefixes
    uint32_t rank[kMaxRanks];     // remaining rank inside the prefix bucket
    uint32_t gid[kMaxRanks];      // histogram group of the query
    uint32_t ngroups;
};

struct RankList {
    uint32_t r[kMaxRanks];
    int n;
};
template <typename V>
struct FracList {
    V t[kMaxPct];
    int n;
};

template <typename K>
__global__ void k_sel_init(SelState<K> *st, int nbc, RankList ranks)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbc) return;
    SelState<K> s;
    for (int q = 0; q < kMaxRanks; ++q) {
        s.prefix[q] = 0;
        s.gprefix[q] = 0;
        s.rank[q] = q < ranks.n ? ranks.r[q] : 0;
        s.gid[q] = 0;
    }
    s.ngroups = 1;
    st[i] = s;
}

template <typename V>
__device__ __forceinline__ void sel_count(uint32_t *h, V v, int shift, int bits, uint32_t mask, int first_pass, int ng,
                                          int nbins, const typename Traits<V>::K *gp)
{
    using K = typename Traits<V>::K;
    const K key = Traits<V>::key(v);
    const uint32_t d = (uint32_t)(key >> shift) & mask;
    if (first_pass) {
        atomicAdd(&h[d], 1u);
    } else {
        // prefixes of distinct groups differ, so at most one matches: select its index, then one conditional atomic
        // (gp[g] for g >= ng holds a value no prefix can take, see k_sel_hist)
        const K pre = key >> (shift + bits);
        uint32_t idx = 0xffffffffu;
#pragma unroll
        for (int g = 0; g < 4; ++g) idx = pre == gp[g] ? (uint32_t)g : idx;
        if (ng > 4) {
#pragma unroll
            for (int g = 4; g < kMaxRanks; ++g) idx = pre == gp[g] ? (uint32_t)g : idx;
        }
        if (idx != 0xffffffffu) atomicAdd(&h[idx * nbins + d], 1u);
    }
}

// grid (blocks, B*3); dynamic LDS = ng_cap * nbins * 4 bytes.  Each block sweeps a contiguous slab of the plane with
// 16-byte loads where the layout allows; only elements whose resolved prefix matches a query group are counted.
template <typename V>
__global__ void __launch_bounds__(256) k_sel_hist(const V *__restrict__ vals, size_t img_stride, size_t chan_stride,
                                                  int elem_stride, int n,
                                                  const SelState<typename Traits<V>::K> *__restrict__ st, int shift,
                                                  int bits, int first_pass, int ng_cap, uint32_t *__restrict__ ghist,
                                                  const uint32_t *__restrict__ only)
{
    using K = typename Traits<V>::K;
    constexpr int VEC = 16 / sizeof(V);
    extern __shared__ uint32_t h[];
    const int bc = blockIdx.y, nbins = 1 << bits;
    if (only && !only[bc]) return;  // fallback sweeps of the linear path: only planes whose candidate list overflowed
    const SelState<K> *s = st + bc;
    const int ng = first_pass ? 1 : min((int)s->ngroups, ng_cap);
    K gp[kMaxRanks];
#pragma unroll
    for (int g = 0; g < kMaxRanks; ++g) gp[g] = g < ng ? s->gprefix[g] : ~K(0);  // ~0 >> (shift + bits) is not a prefix
    for (int i = threadIdx.x; i < ng * nbins; i += 256) h[i] = 0;
    __syncthreads();
    const V *v = vals + (size_t)(bc / 3) * img_stride + (size_t)(bc % 3) * chan_stride;
    const uint32_t mask = (uint32_t)nbins - 1;
    const int per = (((n + 3) / 4 + gridDim.x - 1) / gridDim.x) * 4;                // slab, multiple of 4 elements
    const int lo = min(n, blockIdx.x * per), hi = min(n, lo + per);                 // lo == hi for surplus blocks
    if (elem_stride == 1 && ((size_t)v & 15) == 0) {
        const int hiv = lo + ((hi - lo) / VEC) * VEC;
        constexpr int U = 4;  // 16-byte loads in flight per thread: the sweep is latency-bound, not compute-bound
        int i = lo + threadIdx.x * VEC;
        for (; i + (U - 1) * 256 * VEC < hiv; i += U * 256 * VEC) {
            uint4 raw[U];
#pragma unroll
            for (int u = 0; u < U; ++u) raw[u] = *reinterpret_cast<const uint4 *>(v + i + u * 256 * VEC);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                V q[VEC];
                *reinterpret_cast<uint4 *>(q) = raw[u];
#pragma unroll
                for (int j = 0; j < VEC; ++j) sel_count<V>(h, q[j], shift, bits, mask, first_pass, ng, nbins, gp);
            }
        }
        for (; i < hiv; i += 256 * VEC) {
            V q[VEC];
            *reinterpret_cast<uint4 *>(q) = *reinterpret_cast<const uint4 *>(v + i);
#pragma unroll
            for (int j = 0; j < VEC; ++j) sel_count<V>(h, q[j], shift, bits, mask, first_pass, ng, nbins, gp);
        }
        for (int i = hiv + threadIdx.x; i < hi; i += 256) sel_count<V>(h, v[i], shift, bits, mask, first_pass, ng, nbins, gp);
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += 256)
            sel_count<V>(h, v[(size_t)i * elem_stride], shift, bits, mask, first_pass, ng, nbins, gp);
    }
    __syncthreads();
    uint32_t *gh = ghist + (size_t)bc * kMaxRanks * kBins;
    for (int i = threadIdx.x; i < ng * nbins; i += 256) {
        const uint32_t c = h[i];
        if (c) atomicAdd(&gh[(i / nbins) * kBins + (i % nbins)], c);
    }
}

// one block per (image, channel): narrow each query by one digit, then regroup the prefixes
template <typename V>
__global__ void __launch_bounds__(256) k_sel_scan(SelState<typename Traits<V>::K> *__restrict__ st,
                                                  const uint32_t *__restrict__ ghist, int bits, int nq, int last_pass,
                                                  V *__restrict__ os, const uint32_t *__restrict__ only)
{
    using K = typename Traits<V>::K;
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t found_digit, found_rank;
    const int bc = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (only && !only[bc]) return;
    SelState<K> *s = st + bc;
    const int nbins = 1 << bits, per = nbins / 256;
    const uint32_t *gh = ghist + (size_t)bc * kMaxRanks * kBins;
    for (int q = 0; q < nq; ++q) {
        const uint32_t *hq = gh + s->gid[q] * kBins;
        const uint32_t rank = s->rank[q];
        uint32_t loc = 0;
        for (int i = 0; i < per; ++i) loc += hq[tid * per + i];
        uint32_t incl = wave_incl_scan_u32(loc);
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t off = 0;
        for (int i = 0; i < w; ++i) off += wsum[i];
        incl += off;
        const uint32_t excl = incl - loc;
        if (excl <= rank && rank < incl) {
            uint32_t acc = excl;
            for (int i = 0; i < per; ++i) {
                const uint32_t c = hq[tid * per + i];
                if (rank < acc + c) {
                    found_digit = tid * per + i;
                    found_rank = rank - acc;
                    break;
                }
                acc += c;
            }
        }
        __syncthreads();
        if (tid == 0) {
            s->prefix[q] = (s->prefix[q] << bits) | (K)found_digit;
            s->rank[q] = found_rank;
        }
        __syncthreads();
    }
    if (tid == 0) {
        uint32_t ng = 0;
        for (int q = 0; q < nq; ++q) {
            uint32_t g = 0;
            for (; g < ng; ++g)
                if (s->gprefix[g] == s->prefix[q]) break;
            if (g == ng) s->gprefix[ng++] = s->prefix[q];
            s->gid[q] = g;
        }
        s->ngroups = ng;
        if (last_pass)
            for (int q = 0; q < nq; ++q) os[bc * kMaxRanks + q] = Traits<V>::value(s->prefix[q]);
    }
}

// NumPy's _lerp (numpy/lib/_function_base_impl.py), in the data's dtype
template <typename V>
__device__ __forceinline__ V np_lerp(V a, V b, V t)
{
    const V diff = b - a;
    V r = a + diff * t;
    if (t >= (V)0.5) r = b - diff * ((V)1 - t);
    return r;
}

template <typename V>
__global__ void k_pct_finish(const V *__restrict__ os, int nbc, FracList<V> fr, V *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbc * fr.n) return;
    const int bc = i / fr.n, j = i % fr.n;
    out[i] = np_lerp<V>(os[bc * kMaxRanks + 2 * j], os[bc * kMaxRanks + 2 * j + 1], fr.t[j]);
}

// Percentiles of a stretch of a stretch from ONE selection: the first stretch f1(x) = clip((x - lo1)/(hi1 - lo1 + eps))
// is monotone non-decreasing in floating point (rounding, division by a positive constant and clip all are), so the
// k-th order statistic of f1(img) is f1 of the k-th order statistic of img.  Percentiles 0,1 are the first
// stretch's (six_stadigy.py:196-197); 2,3 are the second's (white_balance, six_stadigy.py:216-217), taken on
// f1(img).  out[bc][4] = lo1, hi1, lo2, hi2.
__global__ void k_pct_finish_chain(const float *__restrict__ os, int nbc, FracList<float> fr, float eps,
                                   float *__restrict__ out)
{
    const int bc = blockIdx.x * blockDim.x + threadIdx.x;
    if (bc >= nbc) return;
    const float *o = os + bc * kMaxRanks;
    const float lo1 = np_lerp<float>(o[0], o[1], fr.t[0]), hi1 = np_lerp<float>(o[2], o[3], fr.t[1]);
    const float den = (hi1 - lo1) + eps;
    float m[4];
    for (int i = 0; i < 4; ++i) m[i] = fminf(fmaxf((o[4 + i] - lo1) / den, 0.0f), 1.0f);
    out[bc * 4 + 0] = lo1;
    out[bc * 4 + 1] = hi1;
    out[bc * 4 + 2] = np_lerp<float>(m[0], m[1], fr.t[2]);
    out[bc * 4 + 3] = np_lerp<float>(m[2], m[3], fr.t[3]);
}

// Host side of np.percentile's index arithmetic ("linear" method), in the data's dtype V.
template <typename V>
void percentile_indices(long long n, double q_percent, uint32_t *prev, uint32_t *next, double *t)
{
    const V q = (V)q_percent / (V)100;
    const V nm1 = (V)(n - 1);
    const V vi = nm1 * q;
    const V p = std::floor(vi);
    if (vi >= nm1) {  // above bounds: both neighbours are the maximum
        *prev = *next = (uint32_t)(n - 1);
        *t = 0.0;
        return;
    }
    *prev = (uint32_t)p;
    *next = *prev + 1;
    *t = (double)(vi - p);
}

template <typename V>
int begin_t(Shape s, const double *q_percent, int nq, void *ws, hipStream_t st, SelectPlan *plan)
{
    using K = typename Traits<V>::K;
    UWIE_REQUIRE(nq >= 1 && nq <= kMaxPct, "percentiles: 1..4 percentiles per call");
    const long long n = (long long)s.npx();
    UWIE_REQUIRE(n >= 1 && n < (1ll << 31), "percentiles: plane size out of range");
    Carver c(ws);
    const int nbc = s.B * 3;
    plan->state = c.take<SelState<uint64_t>>(nbc);  // sized for the wider key
    plan->ghist = c.take<uint32_t>((size_t)nbc * kMaxRanks * kBins);
    plan->os = c.take<double>((size_t)nbc * kMaxRanks);
    plan->nq = nq;
    plan->is64 = sizeof(V) == 8;
    RankList ranks;
    ranks.n = 2 * nq;
    for (int j = 0; j < nq; ++j)
        percentile_indices<V>(n, q_percent[j], &ranks.r[2 * j], &ranks.r[2 * j + 1], &plan->t[j]);
    UWIE_LAUNCH(k_sel_init<K>, dim3(cdiv(nbc, 64)), dim3(64), 0, st, (SelState<K> *)plan->state, nbc, ranks);
    UWIE_LAUNCH_CHECK();
    UWIE_HIP_CHECK(hipMemsetAsync(plan->ghist, 0, sizeof(uint32_t) * (size_t)nbc * kMaxRanks * kBins, st));
    return UWIE_OK;
}

// DifferentiableEnhancement.color_stretch_batch (vgg_16_UIE.py:78-82): per image, the sorted positions
// int((L/100.0) * n) clamped to [0, n-1] of L_low and L_high, evaluated in float64 like the Python expression
// (L is the float32 parameter's value).  params[b*stride + 0/1] = L_low, L_high.
__global__ void k_sel_init_stretch_ranks(SelState<uint32_t> *st, int nbc, const float *__restrict__ params, int stride,
                                         int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbc) return;
    const float *pr = params + (size_t)(i / 3) * stride;
    SelState<uint32_t> s;
    for (int q = 0; q < kMaxRanks; ++q) {
        s.prefix[q] = 0;
        s.gprefix[q] = 0;
        s.rank[q] = 0;
        s.gid[q] = 0;
    }
    for (int q = 0; q < 2; ++q) {
        const double pos = ((double)pr[q] / 100.0) * (double)n;
        long long idx = (long long)pos;  // int(): truncation toward zero
        idx = idx < 0 ? 0 : idx > n - 1 ? n - 1 : idx;
        s.rank[q] = (uint32_t)idx;
    }
    s.ngroups = 1;
    st[i] = s;
}

template <typename V>
int run_t(const SelectPlan &plan, const V *d_vals, int planar, Shape s, bool pass1_done, hipStream_t st,
          const uint32_t *only = nullptr)
{
    using K = typename Traits<V>::K;
    const long long n = (long long)s.npx();
    const int nbc = s.B * 3;
    SelState<K> *state = (SelState<K> *)plan.state;
    // few, fat blocks: the LDS histogram is zeroed and flushed once per block
    int blocks = (int)((n + 262143) / 262144);
    if (blocks * nbc < 1024) blocks = cdiv(1024, nbc);
    blocks = blocks < 1 ? 1 : blocks > 256 ? 256 : blocks;
    const int ng_cap = 2 * plan.nq;  // at most one group per rank
    for (int p = 0; p < Traits<V>::NPASS; ++p) {
        const int shift = Traits<V>::shift(p), bits = Traits<V>::bits(p);
        if (p > 0) UWIE_HIP_CHECK(hipMemsetAsync(plan.ghist, 0, sizeof(uint32_t) * (size_t)nbc * kMaxRanks * kBins, st));
        if (p > 0 || !pass1_done) {
            const size_t lds = (size_t)(p == 0 ? 1 : ng_cap) * (1u << bits) * sizeof(uint32_t);
            UWIE_LAUNCH(k_sel_hist<V>, dim3(blocks, nbc), dim3(256), lds, st, d_vals, (size_t)n * 3,
                        planar ? (size_t)n : (size_t)1, planar ? 1 : 3, (int)n, state, shift, bits, p == 0 ? 1 : 0, ng_cap,
                        plan.ghist, only);
            UWIE_LAUNCH_CHECK();
        }
        UWIE_LAUNCH(k_sel_scan<V>, dim3(nbc), dim3(256), 0, st, state, plan.ghist, bits, 2 * plan.nq,
                    p == Traits<V>::NPASS - 1 ? 1 : 0, (V *)plan.os, only);
        UWIE_LAUNCH_CHECK();
    }
    return UWIE_OK;
}

template <typename V>
int lerp_t(const SelectPlan &plan, Shape s, V *d_out, hipStream_t st)
{
    FracList<V> fr;
    fr.n = plan.nq;
    for (int j = 0; j < plan.nq; ++j) fr.t[j] = (V)plan.t[j];
    const int nbc = s.B * 3;
    UWIE_LAUNCH(k_pct_finish<V>, dim3(cdiv(nbc * plan.nq, 64)), dim3(64), 0, st, (const V *)plan.os, nbc, fr, d_out);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

}  // namespace

// ===================================================================================== linear-digit selection
namespace {

constexpr uint32_t kLinDone = 0xffffffffu;
constexpr int kLinStage = 512;  // candidates a block stages in LDS per group before it reserves list space

// list capacity per (image, channel, group); UWIE_LIN_CAP overrides it (tests force the overflow / fallback path)
uint32_t lin_cap(Shape s)
{
    const uint32_t dflt = (uint32_t)std::max<size_t>(65536, s.npx() / 64);
    const long v = tune().lin_cap;  // (0 outside an entry point: the workspace is sized for the default)
    return v > 0 && v < (long)dflt ? (uint32_t)v : dflt;
}

// digit d (0 <= d < nbins, nbins <= 2304) with excl(d) <= rank < incl(d) over the LDS counts h[]; all 256 threads call
__device__ void block_find_digit(const uint32_t *h, int nbins, uint32_t rank, uint32_t *wsum, uint32_t *found,
                                 uint32_t &digit, uint32_t &rrank)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, per = (nbins + 255) / 256;
    uint32_t loc = 0;
    for (int i = 0; i < per; ++i) {
        const int k = tid * per + i;
        if (k < nbins) loc += h[k];
    }
    uint32_t incl = wave_incl_scan_u32(loc);
    __syncthreads();
    if (lane == 63) wsum[w] = incl;
    if (tid == 0) { found[0] = (uint32_t)nbins - 1; found[1] = 0; }
    __syncthreads();
    for (int i = 0; i < w; ++i) incl += wsum[i];
    const uint32_t excl = incl - loc;
    if (excl <= rank && rank < incl) {
        uint32_t acc = excl;
        for (int i = 0; i < per; ++i) {
            const int k = tid * per + i;
            const uint32_t c = k < nbins ? h[k] : 0;
            if (rank < acc + c) {
                found[0] = (uint32_t)k;
                found[1] = rank - acc;
                break;
            }
            acc += c;
        }
    }
    __syncthreads();
    digit = found[0];
    rrank = found[1];
    __syncthreads();
}

constexpr int kLinSampleOff = 4096;  // the sample histogram of a plane lives behind the producer's in its ghist group

// grid (blocks, B): linear-digit histogram of the restored image on every stride-th group of four pixels (ngs groups;
// the frame as one flat array, so the sample is spread over rows and columns alike)
template <typename V>
__global__ void __launch_bounds__(256) k_lin_sample(RestoreSrc S, int npx, int stride, int ngs, uint32_t *__restrict__ ghist)
{
    __shared__ uint32_t h[3][kLinBins];
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < 3 * kLinBins; i += 256) (&h[0][0])[i] = 0;
    __syncthreads();
    RestoreImg R;
    RestoreImg32 R32;  // UWIE_INTER_F32T (float32 planes only)
    if (S.t32) R32.init(S, b, (size_t)npx);
    else R.init(S, b, (size_t)npx);
    for (int i = blockIdx.x * 256 + tid; i < ngs; i += gridDim.x * 256) {
        V r[3][4];
        if constexpr (sizeof(V) == 8) R.four64((i * stride + stride / 2) * 4, 4, r);  // (ES surface: the float64 image)
        else if (S.t32) R32.four((i * stride + stride / 2) * 4, 4, r);
        else R.four((i * stride + stride / 2) * 4, 4, r);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(&h[c][lin_digit(r[c][j])], 1u);
    }
    __syncthreads();
    for (int i = tid; i < 3 * kLinBins; i += 256) {
        const uint32_t c = (&h[0][0])[i];
        if (c) atomicAdd(&ghist[(size_t)(b * 3 + i / kLinBins) * kSelGroupStride + kLinSampleOff + (i % kLinBins)], c);
    }
}

// one block per (image, channel): clears the state; with a sample (ns > 0) the window of percentile j spans the bins
// that hold the sample ranks r_j*ns/n -+ delta, delta = 4 binomial standard deviations of the sample rank (+2); windows
// that touch are merged.  shift: test knob, moves the windows up by that many bins.
__global__ void __launch_bounds__(256) k_lin_predict(LinState *__restrict__ st, const uint32_t *__restrict__ ghist,
                                                     RankList ranks, uint32_t n, uint32_t ns, int shift)
{
    __shared__ uint32_t h[kLinBins], wsum[4], found[2], wl[kMaxPct], wh[kMaxPct];
    const int bc = blockIdx.x, tid = threadIdx.x;
    if (tid < kMaxPct) { wl[tid] = kLinNoWin; wh[tid] = 0; }
    if (ns > 0) {
        const uint32_t *gh = ghist + (size_t)bc * kSelGroupStride + kLinSampleOff;
        for (int i = tid; i < kLinBins; i += 256) h[i] = gh[i];
        __syncthreads();
        for (int j = 0; j < ranks.n / 2 && j < kMaxPct; ++j) {
            const double p = (double)ranks.r[2 * j] / (double)n, sd = sqrt((double)ns * p * (1.0 - p));
            const double c0 = (double)ranks.r[2 * j] * ns / n, c1 = (double)ranks.r[2 * j + 1] * ns / n, delta = 4.0 * sd + 2.0;
            const uint32_t rlo = (uint32_t)fmax(c0 - delta, 0.0), rhi = (uint32_t)fmin(c1 + delta, (double)ns - 1.0);
            uint32_t dlo, dhi, rr;
            block_find_digit(h, kLinBins, rlo, wsum, found, dlo, rr);
            block_find_digit(h, kLinBins, rhi, wsum, found, dhi, rr);
            if (tid == 0) {  // interior bins only: the scan answers the bins of exact 0 and exact 1 by itself
                wl[j] = (uint32_t)min(max((int)dlo + shift, 1), kLinBins - 2);
                wh[j] = (uint32_t)min(max((int)dhi + shift, (int)wl[j]), kLinBins - 2);
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        LinState s;
        for (int q = 0; q < kMaxRanks; ++q) { s.rr[q] = 0; s.qbin[q] = 0; s.gid[q] = kLinDone; s.gbin[q] = kLinDone; }
        for (int g = 0; g < kLinLists; ++g) s.gcount[g] = 0;
        s.ngroups = 0;
        // windows that touch or overlap become one (the producer files a value under one window only)
        for (bool again = true; again;) {
            again = false;
            for (int i = 0; i < kMaxPct; ++i)
                for (int j = i + 1; j < kMaxPct; ++j)
                    if (wl[i] != kLinNoWin && wl[j] != kLinNoWin && wl[j] <= wh[i] + 1 && wl[i] <= wh[j] + 1) {
                        wl[i] = min(wl[i], wl[j]);
                        wh[i] = max(wh[i], wh[j]);
                        wl[j] = kLinNoWin;
                        again = true;
                    }
        }
        for (int w = 0; w < kMaxPct; ++w) {
            s.wlo[w] = wl[w];
            s.wspan[w] = wl[w] == kLinNoWin ? 0 : wh[w] - wl[w];
            s.below[w] = 0;
        }
        st[bc] = s;
    }
}

// one block per (image, channel): the bin of every rank from the producer's histogram; ranks in the bins of exact
// 0 / exact 1 are answered here.  A plane all of whose other queries fall into the predicted windows is done with
// collecting (ngroups = 0, gid = the window's list); otherwise its queries are grouped by bin for the collecting sweep.
template <typename V>
__global__ void __launch_bounds__(256) k_lin_scan(LinState *__restrict__ st, const uint32_t *__restrict__ ghist,
                                                  RankList ranks, V *__restrict__ os, uint32_t *__restrict__ flags,
                                                  uint32_t cap, uint32_t *__restrict__ bar)
{
    __shared__ uint32_t h[kLinBins], wsum[4], found[2], qbin[kMaxRanks], qrr[kMaxRanks];
    if (bar && threadIdx.x == 0) bar[blockIdx.x] = 0;  // the one-launch fallback's per-plane barrier counter (k_rank_fallback)
    const int bc = blockIdx.x, tid = threadIdx.x;
    const uint32_t *gh = ghist + (size_t)bc * kSelGroupStride;
    for (int i = tid; i < kLinBins; i += 256) h[i] = gh[i];
    __syncthreads();
    for (int q = 0; q < ranks.n; ++q) {
        uint32_t d, rr;
        block_find_digit(h, kLinBins, ranks.r[q], wsum, found, d, rr);
        if (tid == 0) { qbin[q] = d; qrr[q] = rr; }
    }
    __syncthreads();
    if (tid == 0) {
        LinState s = st[bc];
        bool covered = true;
        for (int q = 0; q < kMaxRanks; ++q) { s.rr[q] = 0; s.qbin[q] = 0; s.gid[q] = kLinDone; s.gbin[q] = kLinDone; }
        for (int q = 0; q < ranks.n; ++q) {
            s.rr[q] = qrr[q];
            s.qbin[q] = qbin[q];
            if (qbin[q] == 0 || qbin[q] == kLinBins - 1) {  // every element of these bins is 0 resp. 1
                os[bc * kMaxRanks + q] = qbin[q] == 0 ? (V)0 : (V)1;
                continue;
            }
            int w = -1;
            for (int i = kMaxPct - 1; i >= 0; --i)
                if (qbin[q] - s.wlo[i] <= s.wspan[i]) w = i;
            if (w >= 0 && s.gcount[w] > cap) w = -1;  // the window met a heavy bin: the target bin alone may still fit
            if (w >= 0) s.gid[q] = (uint32_t)w;
            else covered = false;
        }
        uint32_t ng = 0;
        if (!covered) {
            for (int g = 0; g < kLinLists; ++g) s.gcount[g] = 0;
            for (int q = 0; q < ranks.n; ++q) {
                s.gid[q] = kLinDone;
                if (qbin[q] == 0 || qbin[q] == kLinBins - 1) continue;
                uint32_t g = 0;
                for (; g < ng; ++g)
                    if (s.gbin[g] == qbin[q]) break;
                if (g == ng) s.gbin[ng++] = qbin[q];
                s.gid[q] = g;
            }
        }
        s.ngroups = ng;
        st[bc] = s;
        flags[bc] = 0;
    }
}

// After the rank-counting sweep (k_restore_rank): one thread per (image, channel).  Window w holds the elements of bins
// wlo .. wlo + wspan in its list (gcount of them) and `below` elements lie in the bins under it, so rank r belongs to the
// window with below <= r < below + gcount, at position r - below of its list.  Bins 0 and kLinBins - 1 hold exact zeros and
// exact ones only: a rank under a window that starts at bin 1 is 0, one above a window that ends at bin 2048 is 1.  Anything
// else -- the prediction missed, or the list overflowed -- flags the plane for the generic sweeps.
__global__ void k_rank_scan(LinState *__restrict__ st, RankList ranks, float *__restrict__ os, uint32_t *__restrict__ flags,
                            uint32_t cap, int nbc, uint32_t *__restrict__ bar)
{
    const int bc = blockIdx.x * blockDim.x + threadIdx.x;
    if (bc >= nbc) return;
    bar[bc] = 0;  // the fallback's per-plane barrier counter (k_rank_fallback)
    LinState s = st[bc];
    bool miss = false;
    for (int q = 0; q < kMaxRanks; ++q) { s.rr[q] = 0; s.qbin[q] = 0; s.gid[q] = kLinDone; s.gbin[q] = kLinDone; }
    for (int q = 0; q < ranks.n; ++q) {
        const uint32_t r = ranks.r[q];
        bool done = false;
        for (int w = 0; w < kMaxPct && !done; ++w) {
            if (s.wlo[w] == kLinNoWin) continue;
            const uint32_t below = s.below[w], cnt = s.gcount[w];
            if (r >= below && r - below < cnt) {
                if (cnt <= cap) {
                    s.gid[q] = (uint32_t)w;
                    s.rr[q] = r - below;
                    s.qbin[q] = kLinAnyBin;
                } else {
                    miss = true;
                }
                done = true;
            } else if (r < below && s.wlo[w] == 1u) {
                os[bc * kMaxRanks + q] = 0.0f;
                done = true;
            } else if (r >= below && s.wlo[w] + s.wspan[w] == (uint32_t)kLinBins - 2u) {
                os[bc * kMaxRanks + q] = 1.0f;
                done = true;
            }
        }
        miss = miss || !done;
    }
    s.ngroups = 0;
    st[bc] = s;
    flags[bc] = miss ? 1u : 0u;
}

// grid (blocks, B*3): one sweep over the plane; elements whose bin is a target bin go to that group's list.  A block
// stages its candidates in LDS (they are ~0.5 % of the elements) and reserves list space for a batch at a time.
template <typename V>
__global__ void __launch_bounds__(256) k_lin_collect(const V *__restrict__ vals, int n, LinState *__restrict__ st,
                                                     V *__restrict__ lists, uint32_t cap)
{
    constexpr int kStage = sizeof(V) == 8 ? 256 : 512;  // 16 KB of LDS either way
    __shared__ V stg[kMaxRanks][kStage];
    __shared__ uint32_t scount[kMaxRanks], sbase[kMaxRanks];
    const int bc = blockIdx.y, tid = threadIdx.x;
    LinState *s = st + bc;
    const int ng = (int)s->ngroups;
    if (ng == 0) return;
    // Target bins are interior (1..2048: the scan answers bins 0 and 2049 itself), so with d = (uint32)(x * 2048):
    // x is in bin tb  <=>  d == tb - 1, except that x = 0 also gives d = 0: zeros are mapped to a d nobody asks for.
    uint32_t gd[kMaxRanks];
#pragma unroll
    for (int g = 0; g < kMaxRanks; ++g) gd[g] = g < ng ? s->gbin[g] - 1 : kLinDone;
    if (tid < kMaxRanks) scount[tid] = 0;
    __syncthreads();
    V *L = lists + (size_t)bc * kLinLists * cap;
    const V *v = vals + (size_t)bc * n;
    const int per = (((n + 3) / 4 + gridDim.x - 1) / gridDim.x) * 4;
    const int lo = min(n, blockIdx.x * per), hi = min(n, lo + per);
    auto take = [&](V x, bool live) {
        const uint32_t d = (live && x > (V)0) ? (uint32_t)(x * (V)2048) : 0xfffffffeu;
        uint32_t idx = kLinDone;
#pragma unroll
        for (int g = 0; g < 4; ++g) idx = d == gd[g] ? (uint32_t)g : idx;
        if (ng > 4) {
#pragma unroll
            for (int g = 4; g < kMaxRanks; ++g) idx = d == gd[g] ? (uint32_t)g : idx;
        }
        if (idx != kLinDone) {
            const uint32_t pos = atomicAdd(&scount[idx], 1u);
            if (pos < (uint32_t)kStage) {
                stg[idx][pos] = x;
            } else {  // a burst (a smooth region at the percentile's level): straight to the list
                const uint32_t at = atomicAdd(&s->gcount[idx], 1u);
                if (at < cap) L[(size_t)idx * cap + at] = x;
            }
        }
    };
    // all threads of the block call: move the staged candidates to the lists when a stage is three quarters full
    auto flush = [&](bool force) {
        __syncthreads();
        bool need = force;
        for (int g = 0; g < ng; ++g) need = need || scount[g] > (uint32_t)(kStage * 3 / 4);
        if (!need) return;  // block-uniform
        if (tid < ng) sbase[tid] = atomicAdd(&s->gcount[tid], min(scount[tid], (uint32_t)kStage));
        __syncthreads();
        for (int g = 0; g < ng; ++g) {
            const uint32_t c = min(scount[g], (uint32_t)kStage), base = sbase[g];
            for (uint32_t i = tid; i < c; i += 256)
                if (base + i < cap) L[(size_t)g * cap + base + i] = stg[g][i];
        }
        __syncthreads();
        if (tid < kMaxRanks) scount[tid] = 0;
        __syncthreads();
    };
    const bool vec = ((size_t)v & 15) == 0;
    const int hiv = vec ? lo + ((hi - lo) / 4) * 4 : lo;
    constexpr int U = sizeof(V) == 8 ? 2 : 4;  // 16-byte loads: four floats, two doubles (two per group of four)
    for (int base = lo; base < hiv; base += U * 1024) {  // block-uniform trip count
        const int i = base + tid * 4;
        V raw[U][4];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ok[u] = i + u * 1024 < hiv;
            if constexpr (sizeof(V) == 8) {
                const double2 lo2 = ok[u] ? *reinterpret_cast<const double2 *>(v + i + u * 1024) : make_double2(0, 0);
                const double2 hi2 = ok[u] ? *reinterpret_cast<const double2 *>(v + i + u * 1024 + 2) : make_double2(0, 0);
                raw[u][0] = lo2.x; raw[u][1] = lo2.y; raw[u][2] = hi2.x; raw[u][3] = hi2.y;
            } else {
                const float4 q = ok[u] ? *reinterpret_cast<const float4 *>(v + i + u * 1024) : make_float4(0, 0, 0, 0);
                raw[u][0] = q.x; raw[u][1] = q.y; raw[u][2] = q.z; raw[u][3] = q.w;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) take(raw[u][e], ok[u]);
        }
        flush(false);
    }
    for (int t0 = hiv; t0 < hi; t0 += 256) {
        const int j = t0 + tid;
        take(j < hi ? v[j] : (V)0, j < hi);
        flush(false);
    }
    flush(true);
}

// The same sweep when the planes are not stored: grid (blocks, B), the block recomputes its pixels' three restored
// values from the frame and the transmission (restore.h, 11 instead of 12 bytes per pixel and no stored copy) and
// files each under its own channel's groups.  NG: most groups per channel (ranks of the call).
template <typename V, int NG>
__global__ void __launch_bounds__(256) k_lin_collect_src(RestoreSrc S, int n, LinState *__restrict__ st,
                                                         V *__restrict__ lists, uint32_t cap)
{
    constexpr int kStage = sizeof(V) == 8 ? 256 : 512;  // 24 KB of LDS either way (V = double: the ES surface's float64 image)
    __shared__ V stg[3 * NG][kStage];
    __shared__ uint32_t scount[3 * NG], sbase[3 * NG];
    const int b = blockIdx.y, tid = threadIdx.x;
    LinState *s = st + 3 * b;
    int ngc[3];
    uint32_t gd[3][NG];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ngc[c] = min((int)s[c].ngroups, NG);
#pragma unroll
        for (int g = 0; g < NG; ++g) gd[c][g] = g < ngc[c] ? s[c].gbin[g] - 1 : kLinDone;
    }
    if (ngc[0] + ngc[1] + ngc[2] == 0) return;
    if (tid < 3 * NG) scount[tid] = 0;
    __syncthreads();
    RestoreImg R;
    RestoreImg32 R32;  // UWIE_INTER_F32T (float32 planes only)
    if (S.t32) R32.init(S, b, (size_t)n);
    else R.init(S, b, (size_t)n);
    const int per = (((n + 3) / 4 + gridDim.x - 1) / gridDim.x) * 4;
    const int lo = min(n, blockIdx.x * per), hi = min(n, lo + per);
    auto take = [&](int c, V x, bool live) {
        const uint32_t d = (live && x > (V)0) ? (uint32_t)(x * (V)2048) : 0xfffffffeu;  // see k_lin_collect
        uint32_t idx = kLinDone;
#pragma unroll
        for (int g = 0; g < NG; ++g) idx = d == gd[c][g] ? (uint32_t)g : idx;
        if (idx != kLinDone) {
            const uint32_t pos = atomicAdd(&scount[c * NG + idx], 1u);
            if (pos < (uint32_t)kStage) {
                stg[c * NG + idx][pos] = x;
            } else {  // a burst: straight to the list
                const uint32_t at = atomicAdd(&s[c].gcount[idx], 1u);
                if (at < cap) lists[((size_t)(3 * b + c) * kLinLists + idx) * cap + at] = x;
            }
        }
    };
    auto flush = [&](bool force) {  // all threads of the block call
        __syncthreads();
        bool need = force;
        for (int j = 0; j < 3 * NG; ++j) need = need || scount[j] > (uint32_t)(kStage * 3 / 4);
        if (!need) return;  // block-uniform
        if (tid < 3 * NG) {
            const uint32_t c = min(scount[tid], (uint32_t)kStage);
            if (c) sbase[tid] = atomicAdd(&s[tid / NG].gcount[tid % NG], c);
        }
        __syncthreads();
        for (int j = 0; j < 3 * NG; ++j) {
            const uint32_t c = min(scount[j], (uint32_t)kStage), base = sbase[j];
            V *L = lists + ((size_t)(3 * b + j / NG) * kLinLists + (j % NG)) * cap;
            for (uint32_t i = tid; i < c; i += 256)
                if (base + i < cap) L[base + i] = stg[j][i];
        }
        __syncthreads();
        if (tid < 3 * NG) scount[tid] = 0;
        __syncthreads();
    };
    constexpr int U = 2;
    for (int base = lo; base < hi; base += U * 1024) {  // block-uniform trip count
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = base + u * 1024 + tid * 4, m = min(4, hi - p);
            if (m > 0) {
                V r[3][4];
                if constexpr (sizeof(V) == 8) R.four64(p, m, r);
                else if (S.t32) R32.four(p, m, r);
                else R.four(p, m, r);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    take(0, r[0][i], i < m);
                    take(1, r[1][i], i < m);
                    take(2, r[2][i], i < m);
                }
            }
        }
        flush(false);
    }
    flush(true);
}

// One block of 1024 per (image, channel, query): the query is finished on its list by a radix select among the list's
// elements of the query's bin (a window's list holds neighbouring bins too).
// The elements of one linear bin span 1/2048 of [0, 1], so their keys share most of their high bits: fixed key digits
// would put thousands of increments on a handful of LDS counters (same-address atomics serialise; that was 0.28 ms at
// 4K x 64).  The first sweep therefore only copies the bin's elements into LDS (if they fit) and takes the minimum and
// maximum key; the select then runs over the bits in which those two differ, eleven at a time from the top, where the
// elements spread evenly.
template <typename V>
__global__ void __launch_bounds__(1024) k_lin_finish(const LinState *__restrict__ st, const V *__restrict__ lists,
                                                     uint32_t cap, V *__restrict__ os, uint32_t *__restrict__ flags, int pair,
                                                     int nlists)
{
    using K = typename Traits<V>::K;
    constexpr int kBuf = 49152 / (int)sizeof(V);
    __shared__ uint32_t h[2048], wsum[16], found[2], s_nloc;
    __shared__ unsigned long long s_kmin, s_kmax;
    __shared__ V s_buf[kBuf];
    const int bc = blockIdx.x, tid = threadIdx.x, lane = tid & 63;  // grid (B*3, ranks)
    const LinState *s = st + bc;
    // pair: blocks 0 .. ranks/2 - 1 take the even queries (and their successor when it is the next element of the same list),
    // blocks ranks/2 .. the odd queries that are not such successors
    int q = blockIdx.y;
    if (pair) {
        const int half = (int)gridDim.y / 2;
        if (q < half) {
            q = 2 * q;
        } else {
            q = 2 * (q - half) + 1;
            if (s->gid[q] != kLinDone && s->gid[q] == s->gid[q - 1] && s->qbin[q] == s->qbin[q - 1] && s->rr[q] - s->rr[q - 1] <= 1u)
                return;  // its predecessor's block answers it
        }
    }
    const bool with_next = pair && !(q & 1);
    const uint32_t g = s->gid[q];
    if (g == kLinDone) return;  // block-uniform: answered by the scan
    const uint32_t cnt = s->gcount[g];
    if (cnt > cap) {
        if (tid == 0) flags[bc] = 1;
        return;
    }
    const V *L = lists + ((size_t)bc * nlists + g) * cap;  // (nlists lists of `cap` elements per plane: kLinLists, or 2 four times as long)
    uint32_t r = s->rr[q];
    const uint32_t tb = s->qbin[q];
    // A whole window's list (tb == kLinAnyBin, the rank-counting sweep: several bins, ~1 % of the plane, 100 K elements at 4K)
    // does not fit the LDS buffer, and every further pass over it in global memory is a dozen dependent round trips of a
    // lone block (0.27 ms at 4K x 64).  So ONE pass bins its elements 2048 ways across the window's value range and finds the
    // rank's sub-bin; from then on "mine" is that sub-bin's few dozen elements and everything else runs in LDS.
    bool sub = false;
    float sub_lo = 0.0f, sub_scale = 0.0f;
    uint32_t sub_d = 0;
    auto subbin = [&](V x) -> uint32_t {
        const float f = ((float)x - sub_lo) * sub_scale;  // monotone in x: the same expression in every pass
        return (uint32_t)fminf(fmaxf(f, 0.0f), 2047.0f);
    };
    if (tb == kLinAnyBin && cnt > (uint32_t)kBuf) {  // block-uniform
        sub = true;
        sub_lo = (float)(s->wlo[g] - 1u) * (1.0f / 2048.0f);
        sub_scale = 2048.0f * 2048.0f / (float)(s->wspan[g] + 1u);
        for (int i = tid; i < 2048; i += 1024) h[i] = 0;
        __syncthreads();
        constexpr int kFly = sizeof(V) == 4 ? 16 : 8;  // loads in flight per thread (a lone block: latency is all there is)
        for (uint32_t base = 0; base < cnt; base += kFly * 1024) {
            V x[kFly];
#pragma unroll
            for (int u = 0; u < kFly; ++u) {
                const uint32_t i = min(base + u * 1024 + tid, cnt - 1);  // (clamped, not predicated: no wait per load)
                x[u] = L[i];
            }
#pragma unroll
            for (int u = 0; u < kFly; ++u)
                if (base + u * 1024 + tid < cnt) atomicAdd(&h[subbin(x[u])], 1u);
        }
        __syncthreads();
        uint32_t d, rr;
        block_find_digit(h, 2048, r, wsum, found, d, rr);
        sub_d = d;
        r = rr;
    }
    const uint32_t r_mine = r;  // the query's rank among "mine"
    auto is_mine = [&](V x, uint32_t i) -> bool {
        if (tb != kLinAnyBin) return lin_digit(x) == tb;
        return i < cnt && (!sub || subbin(x) == sub_d);
    };
    if (tid == 0) {
        s_nloc = 0;
        s_kmin = ~0ull;
        s_kmax = 0ull;
    }
    __syncthreads();
    {
        unsigned long long kmin = ~0ull, kmax = 0ull;
        constexpr int kFly = sizeof(V) == 4 ? 16 : 8;
        for (uint32_t base = 0; base < cnt; base += kFly * 1024) {  // kFly loads in flight per thread
            V x[kFly];
#pragma unroll
            for (int u = 0; u < kFly; ++u) {
                const uint32_t i = base + u * 1024 + tid;
                const V v = L[min(i, cnt - 1)];  // (clamped load, then the select: no wait per load)
                x[u] = i < cnt ? v : (V)-1;  // (bin 0, never a list's target)
            }
#pragma unroll
            for (int u = 0; u < kFly; ++u) {
                const bool mine = is_mine(x[u], base + u * 1024 + tid);
                const uint64_t m = __ballot(mine);
                if (m) {  // wavefront-aggregated append
                    const int leader = (int)__builtin_ctzll(m);
                    uint32_t at = 0;
                    if (lane == leader) at = atomicAdd(&s_nloc, (uint32_t)__popcll(m));
                    at = __shfl(at, leader) + (uint32_t)__popcll(m & ((1ull << lane) - 1));
                    if (mine) {
                        const unsigned long long key = (unsigned long long)Traits<V>::key(x[u]);
                        kmin = key < kmin ? key : kmin;
                        kmax = key > kmax ? key : kmax;
                        if (at < (uint32_t)kBuf) s_buf[at] = x[u];
                    }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long a = shfl_u64(kmin, lane ^ o), b = shfl_u64(kmax, lane ^ o);
            kmin = a < kmin ? a : kmin;
            kmax = b > kmax ? b : kmax;
        }
        if (lane == 0) {
            atomicMin(&s_kmin, kmin);
            atomicMax(&s_kmax, kmax);
        }
    }
    __syncthreads();
    const uint32_t nloc = s_nloc;
    const bool local = nloc <= (uint32_t)kBuf;
    const unsigned long long kmin = s_kmin, kmax = s_kmax;
    int lo = kmin == kmax ? 0 : 64 - __clzll((long long)(kmin ^ kmax));  // the keys differ in their low `lo` bits only
    unsigned long long prefix = lo >= 64 ? 0ull : kmin >> lo;
    while (lo > 0) {  // block-uniform
        const int bits = min(11, lo), shift = lo - bits, nbins = 1 << bits;
        for (int i = tid; i < nbins; i += 1024) h[i] = 0;
        __syncthreads();
        auto take = [&](V x, bool ok) {
            const unsigned long long key = (unsigned long long)Traits<V>::key(x);
            if (ok && (lo >= 64 || (key >> lo) == prefix)) atomicAdd(&h[(uint32_t)(key >> shift) & (uint32_t)(nbins - 1)], 1u);
        };
        if (local) {
            for (uint32_t i = tid; i < nloc; i += 1024) take(s_buf[i], true);
        } else {
            for (uint32_t base = 0; base < cnt; base += 8192) {
                V x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t i = base + u * 1024 + tid;
                    x[u] = i < cnt ? L[i] : (V)-1;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) take(x[u], is_mine(x[u], base + u * 1024 + tid));
            }
        }
        __syncthreads();
        uint32_t d, rr;
        block_find_digit(h, nbins, r, wsum, found, d, rr);  // (threads 256.. hold no bins there)
        prefix = (prefix << bits) | d;
        r = rr;
        lo = shift;
    }
    if (tid == 0) os[bc * kMaxRanks + q] = Traits<V>::value((K)prefix);
    // pair != 0 (grid.y = queries / 2): the block also answers query q + 1 when it is the NEXT order statistic of the same
    // list (np.percentile's two neighbours, S6:196-197): one more pass -- how many elements are <= x_r, and the smallest one
    // above it -- instead of a second block repeating the whole select.
    if (!with_next) return;
    const int q1 = q + 1;
    if (s->gid[q1] != g || s->qbin[q1] != tb || s->rr[q1] - s->rr[q] > 1u) return;  // (answered by its own block: see the launch)
    if (s->rr[q1] == s->rr[q]) {
        if (tid == 0) os[bc * kMaxRanks + q1] = Traits<V>::value((K)prefix);
        return;
    }
    __syncthreads();
    if (tid == 0) {
        s_nloc = 0;      // elements <= x_r
        s_kmin = ~0ull;  // smallest key above x_r's
    }
    __syncthreads();
    {
        uint32_t cle = 0;
        unsigned long long kgt = ~0ull;
        auto look = [&](V x, bool ok) {
            const unsigned long long key = (unsigned long long)Traits<V>::key(x);
            if (ok) {
                if (key <= prefix) ++cle;
                else kgt = key < kgt ? key : kgt;
            }
        };
        if (local) {
            for (uint32_t i = tid; i < nloc; i += 1024) look(s_buf[i], true);
        } else {
            for (uint32_t base = 0; base < cnt; base += 1024) {
                const uint32_t i = base + tid;
                const V x = i < cnt ? L[i] : (V)-1;
                look(x, is_mine(x, i));
            }
        }
        cle = wave_sum_u32(cle);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long a = shfl_u64(kgt, lane ^ o);
            kgt = a < kgt ? a : kgt;
        }
        if (lane == 0) {
            atomicAdd(&s_nloc, cle);
            atomicMin(&s_kmin, kgt);
        }
    }
    __syncthreads();
    // x_r has rank r_mine among "mine": the next order statistic is x_r again if more than r_mine + 1 of them are <= x_r,
    // else the smallest element above it -- of the sub-bin, or (none left there) of the rest of the list: one more pass
    if (sub && s_nloc <= r_mine + 1u && s_kmin == ~0ull) {  // block-uniform
        unsigned long long kgt = ~0ull;
        for (uint32_t base = 0; base < cnt; base += 1024) {
            const uint32_t i = base + tid;
            if (i < cnt) {
                const unsigned long long key = (unsigned long long)Traits<V>::key(L[i]);
                if (key > prefix && key < kgt) kgt = key;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long a = shfl_u64(kgt, lane ^ o);
            kgt = a < kgt ? a : kgt;
        }
        if (lane == 0) atomicMin(&s_kmin, kgt);
        __syncthreads();
    }
    if (tid == 0) os[bc * kMaxRanks + q1] = Traits<V>::value((K)(s_nloc > r_mine + 1u ? prefix : s_kmin));
}

// The rank route's fallback in ONE launch (round 4): a plane the scan flagged -- a rank outside its predicted window, a list
// that overflowed -- gets its four order statistics from three key-digit sweeps (11 + 11 + 10 bits, as k_sel_hist / k_sel_scan)
// over values RECOMPUTED from the frame and the transmission: no stored float32 planes (12 B/px of workspace that a strategy-2
// call no longer reserves), and one launch that returns at once for every unflagged plane instead of eleven that each do.
// grid (G, B*3), block 256: the G blocks of a flagged plane meet at a counter in global memory between the sweeps.  Workgroups
// are dispatched in the order of their linear index (x fastest), so the G blocks of a plane are dispatched together and the
// blocks that wait can only be waiting for blocks that are resident or about to be: no deadlock while G fits the chip.  The
// wait is bounded all the same (UWIE_STATUS_FALLBACK_SYNC).
constexpr int kFallbackBlocks = 96;  // (24 took 2.8 ms for two flagged 4K planes: a block then sweeps 346 K pixels three times)
__global__ void __launch_bounds__(256) k_rank_fallback(RestoreSrc S, int npx, RankList ranks, const uint32_t *__restrict__ flags,
                                                       uint32_t *__restrict__ ghist, uint32_t *__restrict__ bar, float *__restrict__ os,
                                                       uint32_t *__restrict__ status)
{
    const int bc = blockIdx.y;
    if (!flags[bc]) return;  // block-uniform: the common case
    __shared__ uint32_t h[4][2048], wsum[4], found[2];
    __shared__ uint32_t s_pre[4], s_rank[4];
    const int b = bc / 3, c = bc % 3, tid = threadIdx.x, G = gridDim.x, nq = ranks.n;
    uint32_t *gh = ghist + (size_t)bc * kMaxRanks * kBins;  // 16384 counters: sweep 0 [0, 2048), sweep 1 [2048, 10240), sweep 2 [10240, 14336)
    uint32_t phase = 0;
    auto plane_barrier = [&]() {
        ++phase;
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            atomicAdd(&bar[bc], 1u);
            uint32_t spins = 0;
            while (__hip_atomic_load(&bar[bc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase * (uint32_t)G) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 22)) {  // seconds: a defect, not a wait
                    atomicOr(status, (uint32_t)UWIE_STATUS_FALLBACK_SYNC);
                    break;
                }
            }
            __threadfence();
        }
        __syncthreads();
    };
    for (int i = blockIdx.x * 256 + tid; i < kMaxRanks * kBins; i += G * 256) gh[i] = 0;
    plane_barrier();
    RestoreImg R;
    RestoreImg32 R32;  // UWIE_INTER_F32T: the float32 restore IS the value (restore.h pixel32)
    if (S.t32) R32.init(S, b, (size_t)npx);
    else R.init(S, b, (size_t)npx);
    const int per = (npx + G - 1) / G, lo = min(npx, (int)blockIdx.x * per), hi = min(npx, lo + per);
    uint32_t pre[4] = {0, 0, 0, 0}, rk[4] = {0, 0, 0, 0}, gpre[4] = {0, 0, 0, 0};
    for (int q = 0; q < 4; ++q) rk[q] = q < nq ? ranks.r[q] : ranks.r[0];
    int ng = 1;
    uint32_t goff = 0;
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = Traits<float>::shift(pass), bits = Traits<float>::bits(pass), nbins = 1 << bits;
        for (int i = tid; i < ng * nbins; i += 256) (&h[0][0])[(i / nbins) * 2048 + (i % nbins)] = 0;
        __syncthreads();
        for (int p = lo + tid; p < hi; p += 256) {
            float v;
            if (S.t32) {
                const uint32_t u = R32.img[(size_t)p * 3 + c];
                v = fminf(fmaxf(R32.one32_raw(u, c, R32.recip32(R32.tf()[p])), 0.0f), 1.0f);
            } else {
                const uint32_t u = R.img[(size_t)p * 3 + c];
                const double tv = R.t[p];
                v = R.recip_ok(tv) ? R.one_fast(u, c, tv, R.recip(tv)) : R.one(u, c, tv);
            }
            const uint32_t key = f32_key(v), d = (key >> shift) & (uint32_t)(nbins - 1), hp = pass ? key >> (shift + bits) : 0u;
            for (int g = 0; g < ng; ++g)
                if (hp == gpre[g]) atomicAdd(&h[g][d], 1u);
        }
        __syncthreads();
        for (int i = tid; i < ng * nbins; i += 256) {
            const uint32_t cnt = h[i / nbins][i % nbins];
            if (cnt) atomicAdd(&gh[goff + (uint32_t)i], cnt);
        }
        plane_barrier();
        for (int i = tid; i < ng * nbins; i += 256) h[i / nbins][i % nbins] = __hip_atomic_load(&gh[goff + (uint32_t)i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        // every block narrows the four queries by this digit (redundantly: no second barrier), then regroups the prefixes
        for (int q = 0; q < 4; ++q) {
            int g = 0;
            for (int j = 0; j < ng; ++j)
                if (pre[q] == gpre[j]) g = j;
            uint32_t d, rr;
            block_find_digit(h[g], nbins, rk[q], wsum, found, d, rr);
            if (tid == 0) {
                s_pre[q] = (pre[q] << bits) | d;
                s_rank[q] = rr;
            }
            __syncthreads();
        }
        for (int q = 0; q < 4; ++q) {
            pre[q] = s_pre[q];
            rk[q] = s_rank[q];
        }
        __syncthreads();
        goff += (uint32_t)(ng * nbins);
        ng = 0;
        for (int q = 0; q < 4; ++q) {
            bool seen = false;
            for (int j = 0; j < ng; ++j) seen = seen || gpre[j] == pre[q];
            if (!seen) gpre[ng++] = pre[q];
        }
    }
    if (blockIdx.x == 0 && tid == 0)
        for (int q = 0; q < nq; ++q) os[bc * kMaxRanks + q] = Traits<float>::value(pre[q]);
}

struct LinBufs {
    LinState *lin;
    float *lists;
    uint32_t *flags;
};
LinBufs carve_lin(Carver &c, Shape s)
{
    LinBufs b;
    const size_t nbc = (size_t)s.B * 3;
    b.lin = c.take<LinState>(nbc);
    b.flags = c.take<uint32_t>(nbc);
    b.lists = reinterpret_cast<float *>(c.take<double>(nbc * kLinLists * lin_cap(s)));  // (sized for the float64 surface)
    return b;
}

}  // namespace

int select_lin_begin(Shape s, const double *q_percent, int nq, void *ws, hipStream_t st, SelectPlan *plan,
                     const RestoreSrc *predict)
{
    UWIE_REQUIRE(nq >= 1 && nq <= kMaxPct, "percentiles: 1..4 percentiles per call");
    const long long n = (long long)s.npx();
    UWIE_REQUIRE(n >= 1 && n < (1ll << 31), "percentiles: plane size out of range");
    Carver c(ws);
    const int nbc = s.B * 3;
    plan->state = c.take<SelState<uint64_t>>(nbc);
    plan->ghist = c.take<uint32_t>((size_t)nbc * kMaxRanks * kBins);
    plan->os = c.take<double>((size_t)nbc * kMaxRanks);
    const LinBufs lb = carve_lin(c, s);
    plan->lin = lb.lin;
    plan->lists = lb.lists;
    plan->flags = lb.flags;
    plan->cap = lin_cap(s);
    plan->nq = nq;
    plan->is64 = false;
    for (int j = 0; j < nq; ++j)
        percentile_indices<float>(n, q_percent[j], &plan->ranks[2 * j], &plan->ranks[2 * j + 1], &plan->t[j]);
    // only the first kLinBins counters of every (image, channel) are used by the producer (and kLinBins more by the sample)
    UWIE_HIP_CHECK(hipMemsetAsync(plan->ghist, 0, sizeof(uint32_t) * (size_t)nbc * kMaxRanks * kBins, st));
    RankList ranks;
    ranks.n = 2 * nq;
    for (int j = 0; j < ranks.n; ++j) ranks.r[j] = plan->ranks[j];
    uint32_t ns = 0;
    // tuning (tests): lin_no_predict switches the prediction off, lin_predict_shift = k moves the predicted windows k bins
    // up (a large shift makes every prediction miss: collecting-sweep fallback)
    const int shift = tune().lin_predict_shift;
    plan->predicted = false;
    if (predict && !tune().lin_no_predict) {
        plan->predicted = true;
        // ~128 K sample pixels per frame in evenly spaced groups of four (odd stride: no column is favoured): the
        // sampling error of a 1 % rank is ~0.03 % of the frame, half a bin where the 2048 bins are equally full
        const int ngroups = (int)(n / 4);
        if (ngroups > 0) {
            int stride = std::max(1, ngroups / 32768);
            if (stride > 1) stride |= 1;
            const int ngs = ngroups / stride;
            ns = 4u * (uint32_t)ngs;
            UWIE_LAUNCH(k_lin_sample<float>, dim3(std::max(1, std::min(8, cdiv(ngs, 512))), s.B), dim3(256), 0, st, *predict, (int)n,
                        stride, ngs, plan->ghist);
            UWIE_LAUNCH_CHECK();
        }
    }
    UWIE_LAUNCH(k_lin_predict, dim3(nbc), dim3(256), 0, st, (LinState *)plan->lin, plan->ghist, ranks, (uint32_t)n, ns,
                shift);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

// After the producer has filled the linear-digit histogram: scan -> one collecting sweep -> finish on the lists;
// planes whose list overflowed (a heavy bin, e.g. a constant image) take the generic three sweeps.
int select_lin_run(const SelectPlan &plan, float *d_planar, Shape s, hipStream_t st, const RestoreSrc *src)
{
    const int n = (int)s.npx(), nbc = s.B * 3;
    RankList ranks;
    ranks.n = 2 * plan.nq;
    for (int j = 0; j < ranks.n; ++j) ranks.r[j] = plan.ranks[j];
    UWIE_REQUIRE(!src || ranks.n <= 4, "select_lin_run: the recomputing sweep handles at most two percentiles");
    LinState *lin = (LinState *)plan.lin;
    UWIE_LAUNCH(k_lin_scan<float>, dim3(nbc), dim3(256), 0, st, lin, plan.ghist, ranks, (float *)plan.os, plan.flags, plan.cap,
                src ? reinterpret_cast<uint32_t *>(plan.state) : (uint32_t *)nullptr);
    UWIE_LAUNCH_CHECK();
    int blocks = (int)(((long long)n + 131071) / 131072);
    if (blocks * nbc < 1024) blocks = cdiv(1024, nbc);
    blocks = blocks < 1 ? 1 : blocks > 256 ? 256 : blocks;
    // after a prediction the sweep only serves the few planes it missed (a heavy bin next to the target made the
    // window's list overflow, or the sample was off); its blocks return at once for the others, and the ones that work
    // are alone on the chip: more, smaller blocks (4K x 64 with 4 such planes: 32 per plane 0.29 ms, 128: 0.12 ms)
    if (plan.predicted) blocks = std::max(1, std::min(2 * blocks, cdiv(24576, nbc)));
    if (src) {
        const int per_image = std::min(3 * blocks, std::max(1, cdiv(n, 2048)));
        UWIE_LAUNCH((k_lin_collect_src<float, 4>), dim3(per_image, s.B), dim3(256), 0, st, *src, n, lin, plan.lists, plan.cap);
    } else {
        UWIE_LAUNCH(k_lin_collect<float>, dim3(blocks, nbc), dim3(256), 0, st, (const float *)d_planar, n, lin, plan.lists, plan.cap);
    }
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_lin_finish<float>, dim3(nbc, ranks.n), dim3(1024), 0, st, lin, (const float *)plan.lists, plan.cap, (float *)plan.os,
                plan.flags, 0, kLinLists);
    UWIE_LAUNCH_CHECK();
    // the flagged planes (their kernels return at once for the others).  Recomputed values: the one-launch fallback, no stored
    // planes (round 4; rounds 1-3 wrote the flagged images out and ran the generic chain: eleven launches and 12 B/px of
    // workspace).  Stored planes: the generic three-digit sweeps on them.
    if (src) {
        uwie_ctx *ctx = current_ctx();
        UWIE_REQUIRE(ctx != nullptr, "select_lin_run: needs a context");
        UWIE_LAUNCH(k_rank_fallback, dim3(kFallbackBlocks, nbc), dim3(256), 0, st, *src, n, ranks, plan.flags, plan.ghist,
                    reinterpret_cast<uint32_t *>(plan.state), (float *)plan.os, ctx->d_status);
        UWIE_LAUNCH_CHECK();
        return UWIE_OK;
    }
    UWIE_LAUNCH(k_sel_init<uint32_t>, dim3(cdiv(nbc, 64)), dim3(64), 0, st, (SelState<uint32_t> *)plan.state, nbc, ranks);
    UWIE_LAUNCH_CHECK();
    UWIE_HIP_CHECK(hipMemsetAsync(plan.ghist, 0, sizeof(uint32_t) * (size_t)nbc * kMaxRanks * kBins, st));
    return run_t<float>(plan, d_planar, 1, s, false, st, plan.flags);
}

// After launch_restore_rank: scan -> finish on the window lists; flagged planes take the fallback above.
int select_rank_run(const SelectPlan &plan, float *d_planar, Shape s, hipStream_t st, const RestoreSrc &src)
{
    const int nbc = s.B * 3;
    RankList ranks;
    ranks.n = 2 * plan.nq;
    for (int j = 0; j < ranks.n; ++j) ranks.r[j] = plan.ranks[j];
    LinState *lin = (LinState *)plan.lin;
    UWIE_LAUNCH(k_rank_scan, dim3(cdiv(nbc, 64)), dim3(64), 0, st, lin, ranks, (float *)plan.os, plan.flags, kRankCapMul * plan.cap, nbc,
                reinterpret_cast<uint32_t *>(plan.state));
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_lin_finish<float>, dim3(nbc, ranks.n), dim3(1024), 0, st, lin, (const float *)plan.lists, kRankCapMul * plan.cap,
                (float *)plan.os, plan.flags, 1, kLinLists / kRankCapMul);
    UWIE_LAUNCH_CHECK();
    (void)d_planar;  // (no stored planes on this route)
    uwie_ctx *ctx = current_ctx();
    UWIE_REQUIRE(ctx != nullptr && ranks.n <= 4, "select_rank_run: needs a context and at most two percentiles");
    // the barrier counters: one word per plane in the selection state's memory, cleared by k_rank_scan
    UWIE_LAUNCH(k_rank_fallback, dim3(kFallbackBlocks, nbc), dim3(256), 0, st, src, (int)s.npx(), ranks, plan.flags, plan.ghist,
                reinterpret_cast<uint32_t *>(plan.state), (float *)plan.os, ctx->d_status);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

// float64 planes (ES surface): the same selection without the prediction; the lists hold doubles
int select_lin_begin64(Shape s, const double *q_percent, int nq, void *ws, hipStream_t st, SelectPlan *plan,
                       const RestoreSrc *predict)
{
    UWIE_REQUIRE(nq >= 1 && nq <= kMaxPct, "percentiles: 1..4 percentiles per call");
    const long long n = (long long)s.npx();
    UWIE_REQUIRE(n >= 1 && n < (1ll << 31), "percentiles: plane size out of range");
    Carver c(ws);
    const int nbc = s.B * 3;
    plan->state = c.take<SelState<uint64_t>>(nbc);
    plan->ghist = c.take<uint32_t>((size_t)nbc * kMaxRanks * kBins);
    plan->os = c.take<double>((size_t)nbc * kMaxRanks);
    const LinBufs lb = carve_lin(c, s);
    plan->lin = lb.lin;
    plan->lists = lb.lists;
    plan->flags = lb.flags;
    plan->cap = lin_cap(s);
    plan->nq = nq;
    plan->is64 = true;
    for (int j = 0; j < nq; ++j)
        percentile_indices<double>(n, q_percent[j], &plan->ranks[2 * j], &plan->ranks[2 * j + 1], &plan->t[j]);
    UWIE_HIP_CHECK(hipMemsetAsync(plan->ghist, 0, sizeof(uint32_t) * (size_t)nbc * kMaxRanks * kBins, st));
    RankList ranks;
    ranks.n = 2 * nq;
    for (int j = 0; j < ranks.n; ++j) ranks.r[j] = plan->ranks[j];
    // prediction as in select_lin_begin (same tuning selectors)
    uint32_t ns = 0;
    const int shift = tune().lin_predict_shift;
    plan->predicted = false;
    const int ngroups = (int)(n / 4);
    if (predict && nq <= 2 && ngroups > 0 && !tune().lin_no_predict) {
        plan->predicted = true;
        int stride = std::max(1, ngroups / 32768);
        if (stride > 1) stride |= 1;
        const int ngs = ngroups / stride;
        ns = 4u * (uint32_t)ngs;
        UWIE_LAUNCH(k_lin_sample<double>, dim3(std::max(1, std::min(8, cdiv(ngs, 512))), s.B), dim3(256), 0, st, *predict, (int)n,
                    stride, ngs, plan->ghist);
        UWIE_LAUNCH_CHECK();
    }
    UWIE_LAUNCH(k_lin_predict, dim3(nbc), dim3(256), 0, st, (LinState *)plan->lin, plan->ghist, ranks, (uint32_t)n, ns, shift);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

int select_lin_run64(const SelectPlan &plan, double *d_planar, Shape s, hipStream_t st, const RestoreSrc *src)
{
    const int n = (int)s.npx(), nbc = s.B * 3;
    RankList ranks;
    ranks.n = 2 * plan.nq;
    for (int j = 0; j < ranks.n; ++j) ranks.r[j] = plan.ranks[j];
    UWIE_REQUIRE(!src || ranks.n <= 4, "select_lin_run64: the recomputing sweep handles at most two percentiles");
    LinState *lin = (LinState *)plan.lin;
    double *lists = reinterpret_cast<double *>(plan.lists);
    UWIE_LAUNCH(k_lin_scan<double>, dim3(nbc), dim3(256), 0, st, lin, plan.ghist, ranks, (double *)plan.os, plan.flags, plan.cap,
                (uint32_t *)nullptr);
    UWIE_LAUNCH_CHECK();
    int blocks = (int)(((long long)n + 131071) / 131072);
    if (blocks * nbc < 1024) blocks = cdiv(1024, nbc);
    blocks = blocks < 1 ? 1 : blocks > 256 ? 256 : blocks;
    if (plan.predicted) blocks = std::max(1, std::min(2 * blocks, cdiv(24576, nbc)));  // (see select_lin_run)
    if (src) {
        const int per_image = std::min(3 * blocks, std::max(1, cdiv(n, 2048)));
        const auto k_lin_collect_src64 = k_lin_collect_src<double, 4>;
        UWIE_LAUNCH(k_lin_collect_src64, dim3(per_image, s.B), dim3(256), 0, st, *src, n, lin, lists, plan.cap);
    } else {
        UWIE_LAUNCH(k_lin_collect<double>, dim3(blocks, nbc), dim3(256), 0, st, (const double *)d_planar, n, lin, lists, plan.cap);
    }
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_lin_finish<double>, dim3(nbc, ranks.n), dim3(1024), 0, st, lin, (const double *)lists, plan.cap, (double *)plan.os,
                plan.flags, 0, kLinLists);
    UWIE_LAUNCH_CHECK();
    // generic path for the flagged planes (its kernels return at once for the others); without stored planes the
    // flagged images are written out first
    if (src) {
        const int rc = launch_recover64_planar_hist(src->in, src->A, src->t, s, d_planar, nullptr, st, true, plan.flags);
        if (rc != UWIE_OK) return rc;
    }
    UWIE_LAUNCH(k_sel_init<uint64_t>, dim3(cdiv(nbc, 64)), dim3(64), 0, st, (SelState<uint64_t> *)plan.state, nbc, ranks);
    UWIE_LAUNCH_CHECK();
    UWIE_HIP_CHECK(hipMemsetAsync(plan.ghist, 0, sizeof(uint32_t) * (size_t)nbc * kMaxRanks * kBins, st));
    return run_t<double>(plan, d_planar, 1, s, false, st, plan.flags);
}

size_t select_ws_bytes(Shape s)
{
    Carver c(nullptr);
    const size_t nbc = (size_t)s.B * 3;
    c.take<SelState<uint64_t>>(nbc);
    c.take<uint32_t>(nbc * kMaxRanks * kBins);
    c.take<double>(nbc * kMaxRanks);
    carve_lin(c, s);
    return c.total();
}

int select_begin_stretch_ranks(Shape s, const float *d_params, int stride, void *ws, hipStream_t st, SelectPlan *plan)
{
    const long long n = (long long)s.npx();
    UWIE_REQUIRE(n >= 1 && n < (1ll << 31), "stretch ranks: plane size out of range");
    Carver c(ws);
    const int nbc = s.B * 3;
    plan->state = c.take<SelState<uint64_t>>(nbc);
    plan->ghist = c.take<uint32_t>((size_t)nbc * kMaxRanks * kBins);
    plan->os = c.take<double>((size_t)nbc * kMaxRanks);
    plan->nq = 1;  // two ranks
    plan->is64 = false;
    plan->t[0] = 0.0;
    UWIE_LAUNCH(k_sel_init_stretch_ranks, dim3(cdiv(nbc, 64)), dim3(64), 0, st, (SelState<uint32_t> *)plan->state, nbc,
                d_params, stride, (int)n);
    UWIE_LAUNCH_CHECK();
    UWIE_HIP_CHECK(hipMemsetAsync(plan->ghist, 0, sizeof(uint32_t) * (size_t)nbc * kMaxRanks * kBins, st));
    return UWIE_OK;
}

int select_begin(Shape s, const double *q, int nq, void *ws, hipStream_t st, SelectPlan *plan)
{
    return begin_t<float>(s, q, nq, ws, st, plan);
}
int select_begin64(Shape s, const double *q, int nq, void *ws, hipStream_t st, SelectPlan *plan)
{
    return begin_t<double>(s, q, nq, ws, st, plan);
}
int select_run(const SelectPlan &plan, const float *d_vals, int planar, Shape s, bool pass1_done, hipStream_t st)
{
    return run_t<float>(plan, d_vals, planar, s, pass1_done, st);
}
int select_run64(const SelectPlan &plan, const double *d_vals, int planar, Shape s, bool pass1_done, hipStream_t st)
{
    return run_t<double>(plan, d_vals, planar, s, pass1_done, st);
}
int select_lerp(const SelectPlan &plan, Shape s, float *d_out, hipStream_t st) { return lerp_t<float>(plan, s, d_out, st); }
int select_lerp64(const SelectPlan &plan, Shape s, double *d_out, hipStream_t st)
{
    return lerp_t<double>(plan, s, d_out, st);
}

int select_lerp_chain(const SelectPlan &plan, Shape s, float eps, float *d_pct4, hipStream_t st)
{
    UWIE_REQUIRE(plan.nq == 4 && !plan.is64, "chained stretch needs 4 float32 percentiles");
    FracList<float> fr;
    fr.n = 4;
    for (int j = 0; j < 4; ++j) fr.t[j] = (float)plan.t[j];
    const int nbc = s.B * 3;
    UWIE_LAUNCH(k_pct_finish_chain, dim3(cdiv(nbc, 64)), dim3(64), 0, st, (const float *)plan.os, nbc, fr, eps, d_pct4);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

int launch_percentiles_f32(const float *d_vals, int planar, Shape s, const double *q_percent, int nq, float *d_out,
                           void *ws, hipStream_t st)
{
    SelectPlan plan;
    int rc = select_begin(s, q_percent, nq, ws, st, &plan);
    if (rc != UWIE_OK) return rc;
    rc = select_run(plan, d_vals, planar, s, false, st);
    if (rc != UWIE_OK) return rc;
    return select_lerp(plan, s, d_out, st);
}

}  // namespace uwie
