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

__device__ __forceinline__ uint32_t f32_key(float v)
{
    const uint32_t b = __float_as_uint(v);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
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

// grid (blocks, B*3); dynamic LDS = ngroups_cap * nbins * 4 bytes
__global__ void __launch_bounds__(256) k_sel_hist(const float *__restrict__ vals, size_t img_stride, size_t chan_stride,
                                                  int elem_stride, int n, const SelState *__restrict__ st, int shift,
                                                  int bits, int first_pass, uint32_t *__restrict__ ghist)
{
    extern __shared__ uint32_t h[];
    const int bc = blockIdx.y, nbins = 1 << bits;
    const SelState *s = st + bc;
    const int ng = first_pass ? 1 : (int)s->ngroups;
    uint32_t gp[kMaxRanks];
#pragma unroll
    for (int g = 0; g < kMaxRanks; ++g) gp[g] = s->gprefix[g];
    for (int i = threadIdx.x; i < ng * nbins; i += 256) h[i] = 0;
    __syncthreads();
    const float *v = vals + (size_t)(bc / 3) * img_stride + (size_t)(bc % 3) * chan_stride;
    const uint32_t mask = (uint32_t)nbins - 1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t key = f32_key(v[(size_t)i * elem_stride]);
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

int launch_percentiles_f32(const float *d_vals, int planar, Shape s, const double *q_percent, int nq, float *d_out,
                           void *ws, hipStream_t st)
{
    UWIE_REQUIRE(nq >= 1 && nq <= kMaxPct, "percentiles: 1..4 percentiles per call");
    const long long n = (long long)s.npx();
    UWIE_REQUIRE(n >= 1 && n < (1ll << 31), "percentiles: plane size out of range");
    Carver c(ws);
    const int nbc = s.B * 3;
    SelState *state = c.take<SelState>(nbc);
    uint32_t *ghist = c.take<uint32_t>((size_t)nbc * kMaxRanks * kBins);
    float *os = c.take<float>((size_t)nbc * kMaxRanks);
    RankList ranks;
    FracList fr;
    ranks.n = 2 * nq;
    fr.n = nq;
    for (int j = 0; j < nq; ++j) percentile_indices(n, q_percent[j], &ranks.r[2 * j], &ranks.r[2 * j + 1], &fr.t[j]);
    UWIE_LAUNCH(k_sel_init, dim3(cdiv(nbc, 64)), dim3(64), 0, st, state, nbc, ranks);
    UWIE_LAUNCH_CHECK();
    const int shifts[3] = {21, 10, 0}, bitsv[3] = {11, 11, 10};
    const int blocks = grid_for((size_t)n / 64 + 1, 128);
    for (int p = 0; p < 3; ++p) {
        UWIE_HIP_CHECK(hipMemsetAsync(ghist, 0, sizeof(uint32_t) * (size_t)nbc * kMaxRanks * kBins, st));
        const size_t lds = (size_t)(p == 0 ? 1 : kMaxRanks) * (1u << bitsv[p]) * sizeof(uint32_t);
        UWIE_LAUNCH(k_sel_hist, dim3(blocks, nbc), dim3(256), lds, st, d_vals, (size_t)n * 3,
                           planar ? (size_t)n : (size_t)1, planar ? 1 : 3, (int)n, state, shifts[p], bitsv[p],
                           p == 0 ? 1 : 0, ghist);
        UWIE_LAUNCH_CHECK();
        UWIE_LAUNCH(k_sel_scan, dim3(nbc), dim3(256), 0, st, state, ghist, bitsv[p], 2 * nq, p == 2 ? 1 : 0, os);
        UWIE_LAUNCH_CHECK();
    }
    UWIE_LAUNCH(k_pct_finish, dim3(cdiv(nbc * nq, 64)), dim3(64), 0, st, os, nbc, fr, d_out);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

}  // namespace uwie
