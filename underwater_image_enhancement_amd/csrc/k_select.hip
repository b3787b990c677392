// np.percentile(channel, q) for float32 channels (six_stadigy.py:196-197,216-217; enhancement_strategies.py:265-266).
//
// NumPy (2.2.6, method "linear", float32 input) computes, all in float32:
//     q32 = float32(q) / float32(100);  vi = float32(n - 1) * q32;  prev = floor(vi);  t = vi - prev
//     a = sorted[prev], b = sorted[prev + 1];  r = a + (b - a) * t;  if t >= 0.5: r = b - (b - a) * (1 - t)
// The index arithmetic depends only on (n, q) and is done on the host in float32; the two order statistics per
// percentile are found EXACTLY on the device by a 3-pass MSD radix select on the order-preserving integer image
// of the float bits (11 + 11 + 10 bits): per pass one LDS-privatised histogram sweep over the channel plane, then a
// tiny scan kernel that narrows every query to the digit holding its rank.  Up to 8 ranks (4 percentiles) per
// channel are resolved in the same three sweeps; queries sharing a prefix share a histogram ("group").
#include "common.h"
#include "devutil.h"

namespace uwie {

namespace {

constexpr int kMaxRanks = 2 * kMaxPct;
constexpr int kBins = 2048;

struct SelState {                 // one per (image, channel)
    uint32_t prefix[kMaxRanks];   // key bits resolved so far (right aligned)
    uint32_t rank[kMaxRanks];     // remaining rank inside the prefix bucket
    uint32_t gid[kMaxRanks];      // histogram group of the query
    uint32_t gprefix[kMaxRanks];  // distinct prefixes
    uint32_t ngroups;
};

struct RankList {
    uint32_t r[kMaxRanks];
    int n;
};
struct FracList {
    float t[kMaxPct];
    int n;
};

__device__ __forceinline__ float key_f32(uint32_t k)
{
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}

__global__ void k_sel_init(SelState *st, int nbc, RankList ranks)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbc) return;
    SelState s;
    for (int q = 0; q < kMaxRanks; ++q) {
        s.prefix[q] = 0;
        s.rank[q] = q < ranks.n ? ranks.r[q] : 0;
        s.gid[q] = 0;
        s.gprefix[q] = 0;
    }
    s.ngroups = 1;
    st[i] = s;
}

// grid (blocks, B*3); dynamic LDS = ng_cap * nbins * 4 bytes.  Each block sweeps a contiguous slab of the plane with
// 16-byte loads where the layout allows; only elements whose resolved prefix matches a query group are counted.
__device__ __forceinline__ void sel_count(uint32_t *h, float v, int shift, int bits, uint32_t mask, int first_pass, int ng,
                                          int nbins, const uint32_t *gp)
{
    const uint32_t key = f32_key(v);
    const uint32_t d = (key >> shift) & mask;
    if (first_pass) {
        atomicAdd(&h[d], 1u);
    } else {
        const uint32_t pre = key >> (shift + bits);
#pragma unroll
        for (int g = 0; g < kMaxRanks; ++g)
            if (g < ng && pre == gp[g]) atomicAdd(&h[g * nbins + d], 1u);
    }
}

__global__ void __launch_bounds__(256) k_sel_hist(const float *__restrict__ vals, size_t img_stride, size_t chan_stride,
                                                  int elem_stride, int n, const SelState *__restrict__ st, int shift,
                                                  int bits, int first_pass, int ng_cap, uint32_t *__restrict__ ghist)
{
    extern __shared__ uint32_t h[];
    const int bc = blockIdx.y, nbins = 1 << bits;
    const SelState *s = st + bc;
    const int ng = first_pass ? 1 : min((int)s->ngroups, ng_cap);
    uint32_t gp[kMaxRanks];
#pragma unroll
    for (int g = 0; g < kMaxRanks; ++g) gp[g] = s->gprefix[g];
    for (int i = threadIdx.x; i < ng * nbins; i += 256) h[i] = 0;
    __syncthreads();
    const float *v = vals + (size_t)(bc / 3) * img_stride + (size_t)(bc % 3) * chan_stride;
    const uint32_t mask = (uint32_t)nbins - 1;
    // slab of this block, in units of 4 elements
    const int per = (((n + 3) / 4 + gridDim.x - 1) / gridDim.x) * 4;
    const int lo = min(n, blockIdx.x * per), hi = min(n, lo + per);  // lo == hi for surplus blocks
    if (elem_stride == 1 && ((size_t)v & 15) == 0) {
        const int hi4 = lo + ((hi - lo) & ~3);
        for (int i = lo + threadIdx.x * 4; i < hi4; i += 1024) {
            const float4 q = *reinterpret_cast<const float4 *>(v + i);
            sel_count(h, q.x, shift, bits, mask, first_pass, ng, nbins, gp);
            sel_count(h, q.y, shift, bits, mask, first_pass, ng, nbins, gp);
            sel_count(h, q.z, shift, bits, mask, first_pass, ng, nbins, gp);
            sel_count(h, q.w, shift, bits, mask, first_pass, ng, nbins, gp);
        }
        for (int i = hi4 + threadIdx.x; i < hi; i += 256) sel_count(h, v[i], shift, bits, mask, first_pass, ng, nbins, gp);
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += 256)
            sel_count(h, v[(size_t)i * elem_stride], shift, bits, mask, first_pass, ng, nbins, gp);
    }
    __syncthreads();
    uint32_t *gh = ghist + (size_t)bc * kMaxRanks * kBins;
    for (int i = threadIdx.x; i < ng * nbins; i += 256) {
        const uint32_t c = h[i];
        if (c) atomicAdd(&gh[(i / nbins) * kBins + (i % nbins)], c);
    }
}

// one block per (image, channel): narrow each query by one digit, then regroup the prefixes
__global__ void __launch_bounds__(256) k_sel_scan(SelState *__restrict__ st, const uint32_t *__restrict__ ghist, int bits,
                                                  int nq, int last_pass, float *__restrict__ os)
{
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t found_digit, found_rank;
    const int bc = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    SelState *s = st + bc;
    const int nbins = 1 << bits, per = nbins / 256;  // 8 or 4 bins per thread
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
            s->prefix[q] = (s->prefix[q] << bits) | found_digit;
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
            for (int q = 0; q < nq; ++q) os[bc * kMaxRanks + q] = key_f32(s->prefix[q]);
    }
}

// NumPy's _lerp (numpy/lib/_function_base_impl.py) in float32
__global__ void k_pct_finish(const float *__restrict__ os, int nbc, FracList fr, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbc * fr.n) return;
    const int bc = i / fr.n, j = i % fr.n;
    const float a = os[bc * kMaxRanks + 2 * j], b = os[bc * kMaxRanks + 2 * j + 1], t = fr.t[j];
    const float diff = b - a;
    float r = a + diff * t;
    if (t >= 0.5f) r = b - diff * (1.0f - t);
    out[i] = r;
}

// Percentiles of a stretch of a stretch from ONE selection: the first stretch f1(x) = clip((x - lo1)/(hi1 - lo1 + eps))
// is monotone non-decreasing in floating point (rounding, division by a positive constant and clip all are), so the
// k-th order statistic of f1(img) is f1 of the k-th order statistic of img.  Percentiles 0,1 are the first
// stretch's (six_stadigy.py:196-197); 2,3 are the second's (white_balance, six_stadigy.py:216-217), taken on
// f1(img).  out[bc][4] = lo1, hi1, lo2, hi2.
__device__ __forceinline__ float np_lerp(float a, float b, float t)
{
    const float diff = b - a;
    float r = a + diff * t;
    if (t >= 0.5f) r = b - diff * (1.0f - t);
    return r;
}

__global__ void k_pct_finish_chain(const float *__restrict__ os, int nbc, FracList fr, float eps, float *__restrict__ out)
{
    const int bc = blockIdx.x * blockDim.x + threadIdx.x;
    if (bc >= nbc) return;
    const float *o = os + bc * kMaxRanks;
    const float lo1 = np_lerp(o[0], o[1], fr.t[0]), hi1 = np_lerp(o[2], o[3], fr.t[1]);
    const float den = (hi1 - lo1) + eps;
    float m[4];
    for (int i = 0; i < 4; ++i) m[i] = fminf(fmaxf((o[4 + i] - lo1) / den, 0.0f), 1.0f);
    out[bc * 4 + 0] = lo1;
    out[bc * 4 + 1] = hi1;
    out[bc * 4 + 2] = np_lerp(m[0], m[1], fr.t[2]);
    out[bc * 4 + 3] = np_lerp(m[2], m[3], fr.t[3]);
}

}  // namespace

size_t select_ws_bytes(Shape s)
{
    Carver c(nullptr);
    const size_t nbc = (size_t)s.B * 3;
    c.take<SelState>(nbc);
    c.take<uint32_t>(nbc * kMaxRanks * kBins);
    c.take<float>(nbc * kMaxRanks);
    return c.total();
}

// Host side of np.percentile's index arithmetic (float32, NumPy 2.2.6 "linear" method).
static void percentile_indices(long long n, double q_percent, uint32_t *prev, uint32_t *next, float *t)
{
    const float q32 = (float)q_percent / 100.0f;
    const float nm1 = (float)(n - 1);
    const float vi = nm1 * q32;
    float p = floorf(vi);
    if (vi >= nm1) {  // above bounds: both neighbours are the maximum
        *prev = *next = (uint32_t)(n - 1);
        *t = 0.0f;
        return;
    }
    *prev = (uint32_t)p;
    *next = *prev + 1;
    *t = vi - p;
}

int select_begin(Shape s, const double *q_percent, int nq, void *ws, hipStream_t st, SelectPlan *plan)
{
    UWIE_REQUIRE(nq >= 1 && nq <= kMaxPct, "percentiles: 1..4 percentiles per call");
    const long long n = (long long)s.npx();
    UWIE_REQUIRE(n >= 1 && n < (1ll << 31), "percentiles: plane size out of range");
    Carver c(ws);
    const int nbc = s.B * 3;
    plan->state = c.take<SelState>(nbc);
    plan->ghist = c.take<uint32_t>((size_t)nbc * kMaxRanks * kBins);
    plan->os = c.take<float>((size_t)nbc * kMaxRanks);
    plan->nq = nq;
    RankList ranks;
    ranks.n = 2 * nq;
    for (int j = 0; j < nq; ++j) percentile_indices(n, q_percent[j], &ranks.r[2 * j], &ranks.r[2 * j + 1], &plan->t[j]);
    UWIE_LAUNCH(k_sel_init, dim3(cdiv(nbc, 64)), dim3(64), 0, st, (SelState *)plan->state, nbc, ranks);
    UWIE_LAUNCH_CHECK();
    UWIE_HIP_CHECK(hipMemsetAsync(plan->ghist, 0, sizeof(uint32_t) * (size_t)nbc * kMaxRanks * kBins, st));
    return UWIE_OK;
}

// Runs the histogram sweeps (all three, or the last two when the producer of the values already accumulated the
// first-digit histogram into plan.ghist) and leaves the 2*nq order statistics per (image, channel) in plan.os.
int select_run(const SelectPlan &plan, const float *d_vals, int planar, Shape s, bool pass1_done, hipStream_t st)
{
    const long long n = (long long)s.npx();
    const int nbc = s.B * 3;
    SelState *state = (SelState *)plan.state;
    const int shifts[3] = {21, 10, 0}, bitsv[3] = {11, 11, 10};
    // few, fat blocks: the LDS histogram is zeroed and flushed once per block
    int blocks = (int)((n + 262143) / 262144);
    if (blocks * nbc < 1024) blocks = cdiv(1024, nbc);
    blocks = blocks < 1 ? 1 : blocks > 256 ? 256 : blocks;
    const int ng_cap = 2 * plan.nq;  // at most one group per rank
    for (int p = 0; p < 3; ++p) {
        if (p > 0) UWIE_HIP_CHECK(hipMemsetAsync(plan.ghist, 0, sizeof(uint32_t) * (size_t)nbc * kMaxRanks * kBins, st));
        if (p > 0 || !pass1_done) {
            const size_t lds = (size_t)(p == 0 ? 1 : ng_cap) * (1u << bitsv[p]) * sizeof(uint32_t);
            UWIE_LAUNCH(k_sel_hist, dim3(blocks, nbc), dim3(256), lds, st, d_vals, (size_t)n * 3,
                        planar ? (size_t)n : (size_t)1, planar ? 1 : 3, (int)n, state, shifts[p], bitsv[p], p == 0 ? 1 : 0,
                        ng_cap, plan.ghist);
            UWIE_LAUNCH_CHECK();
        }
        UWIE_LAUNCH(k_sel_scan, dim3(nbc), dim3(256), 0, st, state, plan.ghist, bitsv[p], 2 * plan.nq, p == 2 ? 1 : 0,
                    plan.os);
        UWIE_LAUNCH_CHECK();
    }
    return UWIE_OK;
}

int select_lerp(const SelectPlan &plan, Shape s, float *d_out, hipStream_t st)
{
    FracList fr;
    fr.n = plan.nq;
    for (int j = 0; j < plan.nq; ++j) fr.t[j] = plan.t[j];
    const int nbc = s.B * 3;
    UWIE_LAUNCH(k_pct_finish, dim3(cdiv(nbc * plan.nq, 64)), dim3(64), 0, st, plan.os, nbc, fr, d_out);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

int select_lerp_chain(const SelectPlan &plan, Shape s, float eps, float *d_pct4, hipStream_t st)
{
    UWIE_REQUIRE(plan.nq == 4, "chained stretch needs 4 percentiles");
    FracList fr;
    fr.n = 4;
    for (int j = 0; j < 4; ++j) fr.t[j] = plan.t[j];
    const int nbc = s.B * 3;
    UWIE_LAUNCH(k_pct_finish_chain, dim3(cdiv(nbc, 64)), dim3(64), 0, st, plan.os, nbc, fr, eps, d_pct4);
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
