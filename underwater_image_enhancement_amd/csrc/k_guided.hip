// estimate_transmission (six_stadigy.py:168-180, enhancement_strategies.py:209-234) and guided_filter
// (six_stadigy.py:26-46): the per-pixel initial transmission and the float64 guided filter built from six
// cv2.boxFilter(.., CV_64F, (r, r)) calls.
//
// cv2.boxFilter on CV_64F data is a running sum along each border-extended row (s += E[x+k] - E[x]) followed by
// a running sum down each column (s0 = SUM + Sp; out = s0 * 1/(k*k); SUM = s0 - Sm), so every output's rounding
// depends on the whole chain from the left / top edge.  The kernels of this file keep those chains literally:
// one lane walks one row (k_box_rows_lane / k_box_rows_ring) or one column (k_box_cols) in float64.  Anchor k/2 and
// BORDER_REFLECT_101 as in OpenCV; for even k the window is [x - k/2, x + k/2 - 1].
#include "common.h"
#include "devutil.h"
#include "guided_wave.h"

namespace uwie {

namespace {

// ---- sources for the row pass.  Raw = what a lane holds per pixel; plane(raw, w) = the value of plane w.
typedef double2 __attribute__((aligned(8))) gdouble2_a8;
typedef float4 __attribute__((aligned(4))) gfloat4_a4;
typedef uint32_t __attribute__((aligned(1))) gu32_a1;
typedef double gd2v __attribute__((ext_vector_type(2), aligned(8)));
// The float64 planes are written once and read once, tens of GB behind: nontemporal stores everywhere and nontemporal loads
// in the column pass (a lane's loads there are whole 512-byte runs of a row) took the four passes from 16.9 to 16.25 ms at
// 4K x 64 (A/B inside one run).  The row passes' loads stay plain: their lines are shared between trips.
__device__ __forceinline__ void st_d2(double *p, double v0, double v1) { __builtin_nontemporal_store(gd2v{v0, v1}, reinterpret_cast<gd2v *>(p)); }
__device__ __forceinline__ void st_d1(double *p, double v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ double ld_d1(const double *p) { return __builtin_nontemporal_load(p); }

struct SrcPlanes1 {
    const double *p;
    int H, W;
    static constexpr const char *kName = "k_box_rows_lane<SrcPlanes1>";
    static constexpr int NP = 1;
    static constexpr int CH = 16;  // columns per lane and trip in k_box_rows_lane
    struct Raw {
        double v;
    };
    __device__ __forceinline__ Raw load(int b, int y, int x) const { return Raw{p[((size_t)b * H + y) * W + x]}; }
    __device__ static __forceinline__ double plane(const Raw &r, int, const double *) { return r.v; }
    template <int N>
    __device__ __forceinline__ void load_run(int b, int y, int x, Raw (&r)[N]) const  // N consecutive columns from x
    {
        const double *q = p + ((size_t)b * H + y) * W + x;
#pragma unroll
        for (int i = 0; i < N; i += 2) {
            const gdouble2_a8 v = *reinterpret_cast<const gdouble2_a8 *>(q + i);
            r[i].v = v.x; r[i + 1].v = v.y;
        }
    }
};
struct SrcPlanes2 {
    const double *p0, *p1;
    int H, W;
    static constexpr const char *kName = "k_box_rows_lane<SrcPlanes2>";
    static constexpr int NP = 2;
    struct Raw {
        double a, b;
    };
    __device__ __forceinline__ Raw load(int b, int y, int x) const
    {
        const size_t i = ((size_t)b * H + y) * W + x;
        return Raw{p0[i], p1[i]};
    }
    __device__ static __forceinline__ double plane(const Raw &r, int w, const double *) { return w == 0 ? r.a : r.b; }
    static constexpr int CH = 8;
    template <int N>
    __device__ __forceinline__ void load_run(int b, int y, int x, Raw (&r)[N]) const
    {
        const size_t i0 = ((size_t)b * H + y) * W + x;
#pragma unroll
        for (int i = 0; i < N; i += 2) {
            const gdouble2_a8 va = *reinterpret_cast<const gdouble2_a8 *>(p0 + i0 + i), vb = *reinterpret_cast<const gdouble2_a8 *>(p1 + i0 + i);
            r[i].a = va.x; r[i + 1].a = va.y;
            r[i].b = vb.x; r[i + 1].b = vb.y;
        }
    }
};
// I = gray/255 (float64), p = t0 (.astype(float64): float32 from the u8-derived frame, float64 for a float64 image on the
// dict surface); planes I, p, I*p, I*I (six_stadigy.py:28-36)
template <class TP>
struct SrcGuideT {
    const uint8_t *gray;
    const TP *t0;
    int H, W;
    static constexpr int NP = 4;
    struct Raw {
        TP t0;
        uint32_t g;
    };
    __device__ __forceinline__ Raw load(int b, int y, int x) const
    {
        const size_t i = ((size_t)b * H + y) * W + x;
        return Raw{t0[i], gray[i]};
    }
    __device__ static __forceinline__ double plane(const Raw &r, int w, const double *ilut)
    {
        const double I = ilut[r.g], p = (double)r.t0;
        return w == 0 ? I : w == 1 ? p : w == 2 ? I * p : I * I;
    }
    static constexpr int CH = sizeof(TP) == 4 ? 16 : 8;
    static constexpr const char *kName = "k_box_rows_lane<SrcGuideT>";
    template <int N>
    __device__ __forceinline__ void load_run(int b, int y, int x, Raw (&r)[N]) const
    {
        const size_t i0 = ((size_t)b * H + y) * W + x;
#pragma unroll
        for (int i = 0; i < N; i += 4) {
            const uint32_t g4 = *reinterpret_cast<const gu32_a1 *>(gray + i0 + i);
            r[i].g = g4 & 255u; r[i + 1].g = (g4 >> 8) & 255u; r[i + 2].g = (g4 >> 16) & 255u; r[i + 3].g = g4 >> 24;
            if constexpr (sizeof(TP) == 4) {
                const gfloat4_a4 v = *reinterpret_cast<const gfloat4_a4 *>(t0 + i0 + i);
                r[i].t0 = v.x; r[i + 1].t0 = v.y; r[i + 2].t0 = v.z; r[i + 3].t0 = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const gdouble2_a8 v = *reinterpret_cast<const gdouble2_a8 *>(t0 + i0 + i + j);
                    r[i + j].t0 = v.x; r[i + j + 1].t0 = v.y;
                }
            }
        }
    }
};
using SrcGuide = SrcGuideT<float>;

// Row pass, round 4: the same literal chain, one LANE per row and every plane of the row in that lane.
// Rounds 1-3 staged 64 x 16 tiles through LDS for a workgroup of NP wavefronts: 240 barrier-separated tiles of a 4K row,
// each a global round trip (4.0 + 2.6 ms of the exact-order filter's 10.0 at 4K x 16).  Here a workgroup is ONE wavefront that
// owns 64 rows: a lane reads its row's entering and leaving elements CH columns at a time (16-byte loads, the next trip's
// issued before this trip's arithmetic), keeps the NP running sums in registers (NP independent dependency chains per lane)
// and hands the trip's NP x 64 x 16 results through an 8.5 KB LDS tile so that they leave as whole 128-byte lines (eight rows
// per store instruction; stored lane by lane they were 64 partial lines per instruction and the pass ran at 1.7 TB/s).  The
// tile is wavefront-private: no barrier anywhere.  grid (ceil(H / 64), B), block 64; trips are aligned to 16 columns.
__device__ __forceinline__ void gwave_lds_sync()  // LDS executes one wavefront's instructions in order: a compiler fence is all
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kLaneOT = 17;  // tile row stride in doubles: lane = row writes hit 64 different bank pairs

// K > 0: the window is known at compile time and no leaving element is loaded at all -- E[col - 1] entered the window K
// columns earlier, so the lane keeps its last ceil(K / CH) runs of entering elements in registers.  K = 0: any window, the
// leaving run is loaded like the entering one (it hits in L2).
// BOUND: nothing but the running sums in front of every 16th column is stored -- out[((b * nstrips + j) * NP + p) * H + y] =
// the sum of row y before column 16 j (j = 0: the initial sum, which is column 0's result) -- for k_box_fused_ab below, whose
// workgroups restart the chains there; plane_stride then carries nstrips.
template <class Src, int K, bool BOUND = false>
__global__ void __launch_bounds__(64) k_box_rows_lane(Src src, double *__restrict__ out, size_t plane_stride, int k_rt)
{
    constexpr int NP = Src::NP, CH = Src::CH;  // CH = 8 or 16 columns per trip
    constexpr int D = K > 0 ? (K + CH - 1) / CH : 1;  // runs of history in front of the current one
    using Raw = typename Src::Raw;
    __shared__ double ilut[256];
    __shared__ double otile[BOUND ? 1 : NP][BOUND ? 1 : 64 * kLaneOT];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) ilut[i] = (double)i / 255.0;  // six_stadigy.py:177
    __syncthreads();
    const int k = K > 0 ? K : k_rt;
    const int b = blockIdx.y, H = src.H, W = src.W, a = k / 2, y0 = (int)blockIdx.x * 64;
    const int y = min(y0 + lane, H - 1);  // (rows past the end repeat the last one and store nothing)
    const bool live = y0 + lane < H;
    double s[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) s[p] = 0.0;
    for (int j = 0; j < k; ++j) {  // E[0] + ... + E[k-1], left to right (BORDER_REFLECT_101, anchor a)
        const Raw r = src.load(b, y, reflect101(j - a, W));
#pragma unroll
        for (int p = 0; p < NP; ++p) s[p] += Src::plane(r, p, ilut);
    }
    double *o[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        o[p] = out + (size_t)p * plane_stride + ((size_t)b * H + y) * W;
        if (!BOUND && live) o[p][0] = s[p];
    }
    // columns col0 .. col0 + CH - 1: out[col] = (s += E[col - 1 + k] - E[col - 1]); E[j] = src[reflect101(j - a)], so the
    // entering run starts at source column xe(col0) = col0 - 1 + k - a and the leaving one at col0 - 1 - a.
    // hist[D * CH + c] is the entering element of column col0 + c; hist[i] is source column xe(col0) + i - D * CH.
    // interior trips: nothing the trip (or its history) reads is in the border extension
    const int xe0 = k - 1 - a;
    auto interior = [&](int col0) {
        const int first = K > 0 ? col0 + xe0 - D * CH : col0 - 1 - a;
        return first >= 0 && col0 + xe0 + CH <= W && col0 + CH <= W;
    };
    Raw hist[(D + 1) * CH], nx[CH];
    bool primed = false, have_nx = false;
    // the transposed side of the tile: lane -> row i * (64 / LPR) + lane / LPR, column pair (lane % LPR) * 2
    constexpr int LPR = CH / 2, RPI = 64 / LPR;
    const int t_row = lane / LPR, t_col = (lane % LPR) * 2;
    for (int col0 = 0; col0 < W; col0 += CH) {
        if (BOUND && (col0 & 15) == 0 && live) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
                out[(((size_t)b * plane_stride + (size_t)(col0 >> 4)) * NP + p) * H + y] = s[p];
        }
        if (!interior(col0)) {  // a trip that touches the left / right border, or the ragged end of the row: element by element
            for (int c = 0; c < CH && col0 + c < W; ++c) {
                const int col = col0 + c;
                if (col == 0) continue;  // (stored above)
                const Raw le = src.load(b, y, reflect101(col - 1 + k - a, W)), tr = src.load(b, y, reflect101(col - 1 - a, W));
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    s[p] += Src::plane(le, p, ilut) - Src::plane(tr, p, ilut);
                    if (!BOUND && live) o[p][col] = s[p];
                }
            }
            primed = false;
            have_nx = false;
            continue;
        }
        if (K > 0 && !primed) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                Raw tmp[CH];
                src.template load_run<CH>(b, y, col0 + xe0 - (D - d) * CH, tmp);
#pragma unroll
                for (int c = 0; c < CH; ++c) hist[d * CH + c] = tmp[c];
            }
            primed = true;
        }
        if (!have_nx) src.template load_run<CH>(b, y, col0 + xe0, nx);
#pragma unroll
        for (int c = 0; c < CH; ++c) hist[D * CH + c] = nx[c];
        if (K == 0) {
            Raw tmp[CH];
            src.template load_run<CH>(b, y, col0 - 1 - a, tmp);
#pragma unroll
            for (int c = 0; c < CH; ++c) hist[c] = tmp[c];  // D = 1: hist[0 .. CH) is the leaving run
        }
        have_nx = interior(col0 + CH);
        if (have_nx) src.template load_run<CH>(b, y, col0 + CH + xe0, nx);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const Raw &le = hist[D * CH + c], &tr = hist[K > 0 ? D * CH + c - K : c];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                s[p] += Src::plane(le, p, ilut) - Src::plane(tr, p, ilut);
                if (!BOUND) otile[p][lane * kLaneOT + c] = s[p];
            }
        }
        if (K > 0) {
#pragma unroll
            for (int i = 0; i < D * CH; ++i) hist[i] = hist[i + CH];
        }
        if (BOUND) continue;
        gwave_lds_sync();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            double *op = out + (size_t)p * plane_stride + ((size_t)b * H + y0) * W + col0 + t_col;
#pragma unroll
            for (int i = 0; i < LPR; ++i) {
                const int row = i * RPI + t_row;
                const double v0 = otile[p][row * kLaneOT + t_col], v1 = otile[p][row * kLaneOT + t_col + 1];
                if (y0 + row < H) st_d2(op + (size_t)row * W, v0, v1);
            }
        }
        gwave_lds_sync();
    }
}

template <class Src>
static void launch_rows_lane(Src src, double *rs, size_t n, int k, int B, hipStream_t st)
{
    const dim3 grid(cdiv(src.H, 64), B), blk(64);
    UWIE_PROF(Src::kName, st);
    if (k == 15) hipLaunchKernelGGL((k_box_rows_lane<Src, 15>), grid, blk, 0, st, src, rs, n, k);
    else hipLaunchKernelGGL((k_box_rows_lane<Src, 0>), grid, blk, 0, st, src, rs, n, k);
}

// Row pass over float64 PLANES (the second box filter's a and b, and uwie_box_filter_f64), windows of at most 16 columns: the
// lane kernel above reads such a plane in 16-byte pieces, 64 lines per instruction, and every line comes back for the next
// trip and again when its elements leave (6.7 ms at 4K x 64 against 5.4 for the four-plane pass that moves more).  Here one
// wavefront owns 64 rows of ONE plane (blockIdx.z) and a trip's entering run arrives as whole lines -- eight rows per load
// instruction, the next trip's issued before this trip's arithmetic -- into a 32-column LDS ring per row; the ring still holds
// the previous run, so the leaving elements (k <= 16 columns back) are read from it and never loaded.  Results leave through
// the same kind of tile as above.  25.6 KB of LDS per wavefront; wavefront-private, no barrier.
constexpr int kRingW = 32, kRingS = kRingW + 1;  // ring columns per row and its row stride in doubles

__global__ void __launch_bounds__(64) k_box_rows_ring(const double *__restrict__ p0, const double *__restrict__ p1, int H, int W,
                                                      double *__restrict__ out, size_t plane_stride, int k)
{
    constexpr int CH = 16, LPR = CH / 2, RPI = 64 / LPR;
    __shared__ double ring[64 * kRingS];
    __shared__ double otile[64 * kLaneOT];
    const int lane = threadIdx.x, b = blockIdx.y, a = k / 2, y0 = (int)blockIdx.x * 64;
    const double *src = (blockIdx.z ? p1 : p0) + (size_t)b * H * W;
    out += (size_t)blockIdx.z * plane_stride + (size_t)b * H * W;
    const int y = min(y0 + lane, H - 1);  // (rows past the end repeat the last one and store nothing)
    const bool live = y0 + lane < H;
    const double *srow = src + (size_t)y * W;
    double *orow = out + (size_t)y * W;
    double s = 0.0;
    for (int j = 0; j < k; ++j) s += srow[reflect101(j - a, W)];  // E[0] + ... + E[k-1], left to right
    if (live) orow[0] = s;
    const int xe0 = k - 1 - a;  // the entering run of columns col0 .. col0 + 15 starts at source column col0 + xe0
    auto interior = [&](int col0) { return col0 + xe0 - CH >= 0 && col0 + xe0 + CH <= W && col0 + CH <= W; };
    const int t_row = lane / LPR, t_col = (lane % LPR) * 2;
    // the coalesced side: instruction i covers rows i * 8 + lane / 8 (clamped into the frame), two columns per lane
    auto load_run = [&](int x, gdouble2_a8 (&r)[LPR]) {
#pragma unroll
        for (int i = 0; i < LPR; ++i)  // (plain loads: a row's run shares its lines with the next trip's; nontemporal, 3.7 -> 5.0 ms)
            r[i] = *reinterpret_cast<const gdouble2_a8 *>(src + (size_t)min(y0 + i * RPI + t_row, H - 1) * W + x + t_col);
    };
    auto put_run = [&](int slot, const gdouble2_a8 (&r)[LPR]) {
#pragma unroll
        for (int i = 0; i < LPR; ++i) {
            double *q = ring + (i * RPI + t_row) * kRingS + slot * CH + t_col;
            q[0] = r[i].x;
            q[1] = r[i].y;
        }
    };
    gdouble2_a8 nx[LPR];
    bool primed = false, have_nx = false;
    int slot = 0;
    for (int col0 = 0; col0 < W; col0 += CH) {
        if (!interior(col0)) {  // a trip that touches the left / right border, or the ragged end of the row: element by element
            for (int c = 0; c < CH && col0 + c < W; ++c) {
                const int col = col0 + c;
                if (col == 0) continue;  // (stored above)
                s += srow[reflect101(col - 1 + k - a, W)] - srow[reflect101(col - 1 - a, W)];
                if (live) orow[col] = s;
            }
            primed = false;
            have_nx = false;
            continue;
        }
        slot ^= 1;
        if (!primed) {  // the run before this trip's: its elements leave during this trip
            gdouble2_a8 pv[LPR];
            load_run(col0 + xe0 - CH, pv);
            put_run(slot ^ 1, pv);
            primed = true;
        }
        if (!have_nx) load_run(col0 + xe0, nx);
        put_run(slot, nx);
        gwave_lds_sync();
        have_nx = interior(col0 + CH);
        if (have_nx) load_run(col0 + CH + xe0, nx);
        const double *rrow = ring + lane * kRingS;
        const int le0 = slot * CH, tr0 = slot * CH - k + kRingW;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            s += rrow[le0 + c] - rrow[(tr0 + c) & (kRingW - 1)];
            otile[lane * kLaneOT + c] = s;
        }
        gwave_lds_sync();
        double *op = out + (size_t)y0 * W + col0 + t_col;
#pragma unroll
        for (int i = 0; i < LPR; ++i) {
            const int row = i * RPI + t_row;
            const double v0 = otile[row * kLaneOT + t_col], v1 = otile[row * kLaneOT + t_col + 1];
            if (y0 + row < H) st_d2(op + (size_t)row * W, v0, v1);
        }
        gwave_lds_sync();
    }
}

// rows of one or two float64 planes: the ring kernel when the window fits it, the lane kernel otherwise
static void launch_rows_planes(const double *p0, const double *p1, Shape s, double *rs, size_t n, int k, hipStream_t st)
{
    if (k <= 16 && s.W >= 2) {
        UWIE_PROF(p1 ? "k_box_rows_ring x2" : "k_box_rows_ring", st);
        hipLaunchKernelGGL(k_box_rows_ring, dim3(cdiv(s.H, 64), s.B, p1 ? 2 : 1), dim3(64), 0, st, p0, p1, s.H, s.W, rs, n, k);
    } else if (p1) {
        launch_rows_lane(SrcPlanes2{p0, p1, s.H, s.W}, rs, n, k, s.B, st);
    } else {
        launch_rows_lane(SrcPlanes1{p0, s.H, s.W}, rs, n, k, s.B, st);
    }
}

// ---- epilogues of the column pass
struct EpiStore1 {
    double *dst;
    static constexpr const char *kName = "k_box_cols<EpiStore1>";
    static constexpr int NP = 1, U = 7;  // U: rows per load batch of the ring walk in k_box_cols
    __device__ __forceinline__ void operator()(const double *, size_t i, const double *m) const { st_d1(dst + i, m[0]); }
};
struct EpiAB {  // a = cov/(var+eps), b = mean_p - a*mean_I (six_stadigy.py:34-40)
    double *a, *b;
    double eps;
    static constexpr const char *kName = "k_box_cols<EpiAB>";
    static constexpr int NP = 4, U = 2;
    __device__ __forceinline__ void operator()(const double *, size_t i, const double *m) const
    {
        const double cov = m[2] - m[0] * m[1];
        const double var = m[3] - m[0] * m[0];
        const double av = cov / (var + eps);
        st_d1(a + i, av);
        st_d1(b + i, m[1] - av * m[0]);
    }
};
struct EpiQ {  // q = mean_a*I + mean_b, then np.clip(q, 0.1, 1.0) (six_stadigy.py:45,180)
    const uint8_t *gray;
    double *t;
    static constexpr const char *kName = "k_box_cols<EpiQ>";
    static constexpr int NP = 2, U = 7;
    __device__ __forceinline__ void operator()(const double *ilut, size_t i, const double *m) const
    {
        const double q = m[0] * ilut[gray[i]] + m[1];
        st_d1(t + i, fmin(fmax(q, 0.1), 1.0));
    }
};

// K > 0: the window is known at compile time and the leaving row is never loaded: what enters at row step y leaves at step
// y + K - 1, so the lane keeps its last K - 1 entering values of every plane in a register ring (the ring index is the unrolled
// step).  With loads, the leaving rows of a 4K x 64 batch are 110 MB behind the entering ones: out of L2 again.
template <class Epi, int K>
__global__ void __launch_bounds__(64) k_box_cols(const double *__restrict__ rs, size_t plane_stride, Epi epi, int B, int H,
                                                 int W, int k_rt)
{
    constexpr int NP = Epi::NP;
    __shared__ double ilut[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) ilut[i] = (double)i / 255.0;
    __syncthreads();
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= B * W) return;
    const int k = K > 0 ? K : k_rt;
    const int b = col / W, x = col % W, a = k / 2;
    const double scale = 1.0 / ((double)k * (double)k);
    const double *base = rs + (size_t)b * H * W + x;
    double sum[NP], mean[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) sum[p] = 0.0;
    for (int j = 0; j < k - 1; ++j) {
        const size_t r = (size_t)reflect101(j - a, H) * W;
#pragma unroll
        for (int p = 0; p < NP; ++p) sum[p] += base[p * plane_stride + r];
    }
    // rows [y_lo, y_hi): neither the entering nor the leaving row is in the border extension
    const int y_lo = min(a, H), y_hi = max(y_lo, H - (k - 1 - a));
    int y = 0;
    auto one_row = [&](int yy) {
        const size_t rp = (size_t)reflect101(yy + k - 1 - a, H) * W, rm = (size_t)reflect101(yy - a, H) * W;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const double s0 = sum[p] + base[p * plane_stride + rp];
            mean[p] = s0 * scale;
            sum[p] = s0 - base[p * plane_stride + rm];
        }
        epi(ilut, ((size_t)b * H + yy) * W + x, mean);
    };
    for (; y < y_lo; ++y) one_row(y);
    if constexpr (K > 1) {
        constexpr int R = K - 1, U = Epi::U, NB = R / U;  // ring entries; entering rows are loaded U at a time, one batch ahead
        static_assert(R % U == 0, "the unrolled period is whole batches");
        if (y + R <= y_hi) {
            double ring[R][NP], in[U][NP];
#pragma unroll
            for (int u = 0; u < R; ++u)  // the rows that leave at steps y .. y + R - 1: y - a + u
#pragma unroll
                for (int p = 0; p < NP; ++p) ring[u][p] = ld_d1(base + p * plane_stride + (size_t)(y - a + u) * W);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int p = 0; p < NP; ++p) in[u][p] = ld_d1(base + p * plane_stride + (size_t)(y + K - 1 - a + u) * W);
            for (; y + R <= y_hi; y += R) {
#pragma unroll
                for (int h = 0; h < NB; ++h) {
                    double nx[U][NP];
#pragma unroll
                    for (int u = 0; u < U; ++u) {  // (past the last period: any valid row, the values are dropped)
                        const size_t r = (size_t)min(y + (h + 1) * U + K - 1 - a + u, H - 1) * W;
#pragma unroll
                        for (int p = 0; p < NP; ++p) nx[u][p] = ld_d1(base + p * plane_stride + r);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            const double s0 = sum[p] + in[u][p];
                            mean[p] = s0 * scale;
                            sum[p] = s0 - ring[h * U + u][p];
                            ring[h * U + u][p] = in[u][p];
                        }
                        epi(ilut, ((size_t)b * H + y + h * U + u) * W + x, mean);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int p = 0; p < NP; ++p) in[u][p] = nx[u][p];
                    __builtin_amdgcn_sched_barrier(0);  // (or the scheduler lifts every batch's loads to the top of the period)
                }
            }
        }
    } else {
        constexpr int kColU = 4;  // rows per trip, every load of the trip issued before its arithmetic
        for (; y + kColU <= y_hi; y += kColU) {
            double in[kColU][NP], ou[kColU][NP];
            const double *pin = base + (size_t)(y + k - 1 - a) * W, *pou = base + (size_t)(y - a) * W;
#pragma unroll
            for (int u = 0; u < kColU; ++u)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    in[u][p] = pin[p * plane_stride + (size_t)u * W];
                    ou[u][p] = pou[p * plane_stride + (size_t)u * W];
                }
#pragma unroll
            for (int u = 0; u < kColU; ++u) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const double s0 = sum[p] + in[u][p];
                    mean[p] = s0 * scale;
                    sum[p] = s0 - ou[u][p];
                }
                epi(ilut, ((size_t)b * H + y + u) * W + x, mean);
            }
        }
    }
    for (; y < H; ++y) one_row(y);
}

// ---- First box filter of the exact-order mode, rows and columns in one kernel (k = 15): the four row-sum planes never
// leave the chip.  The column chains are sequential from row 0 and the row chains from column 0, so a workgroup can own all
// rows of a 16-column STRIP only if something hands it the row chains' values at the strip's left edge: k_box_rows_lane<BOUND>
// walks the rows once for those (2 B/px written instead of 32).  One wavefront per strip then goes down the frame 16 rows at
// a time, lane = 4 * i + plane throughout:
//   stage    16 rows x 32 raw columns (x0 - 8 .. x0 + 23: whole 128-byte runs, four lanes per row) -> {I, p} as doubles in LDS
//   rows     lane (row i, plane): the 16 chain steps s += E[col + 7] - E[col - 8] from the boundary value, literally as
//            k_box_rows_lane does them; results into a 32-row LDS ring [row & 31][4 * col + plane]
//   columns  lane (column i, plane): the literal column chain of k_box_cols over the 16 new rows (output row = row - 7,
//            entering row from the ring, leaving row 14 rows back in the ring, reflected rows at the top / bottom too)
//   a, b     lane (output row i, 4 columns): the four means of a pixel from LDS, EpiAB's arithmetic, 32-byte stores
// Frames of at least 16 rows; the edge strips stage their raw columns element by element (BORDER_REFLECT_101).
constexpr int kFuseRS = 68;  // ring / means row stride in doubles: 64 + 4 (row-phase writes of 32 lanes hit 32 bank pairs)

template <class TP>
__global__ void __launch_bounds__(64) k_box_fused_ab(const uint8_t *__restrict__ gray, const TP *__restrict__ t0,
                                                     const double *__restrict__ bound, int H, int W, int nstrips, double eps,
                                                     double *__restrict__ pa, double *__restrict__ pb)
{
    constexpr int K = 15, A = 7;
    __shared__ double ilut[256];
    __shared__ double ring[32 * kFuseRS];
    __shared__ __attribute__((aligned(16))) double2 stage[16 * 34];  // {I, p}; the column pass's means live here too
    double *means = reinterpret_cast<double *>(stage);
    static_assert(sizeof(stage) >= 16 * kFuseRS * sizeof(double), "means alias the stage");
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) ilut[i] = (double)i / 255.0;  // six_stadigy.py:177
    __syncthreads();
    const int3 bid = xcd_folded_block();  // neighbouring strips (they share their raw lines) run behind the same L2
    const int strip = bid.x, b = bid.z, x0 = strip * 16;
    const int li = lane >> 2, pl = lane & 3;
    const uint8_t *g = gray + (size_t)b * H * W;
    const TP *t = t0 + (size_t)b * H * W;
    const double *bnd = bound + (((size_t)b * nstrips + strip) * 4 + pl) * H;
    const bool edge = x0 - 8 < 0 || x0 + 24 > W;  // uniform
    const double scale = 1.0 / ((double)K * (double)K);

    // raw data of one 16-row block: lane (row li, quarter pl) holds raw columns x0 - 8 + 8 pl .. + 7 of row r0 + li, exactly as
    // loaded (a conversion at the load would make the wavefront wait for it at once; the loads run two blocks ahead)
    struct RawRun {
        TP tv[8];
        uint32_t g0, g1;  // the eight guide bytes
        double s0;        // the chain's value in front of the strip, for (row li, plane pl)
    };
    auto fetch = [&](int r0, RawRun &R) {
        const int y = min(r0 + li, H - 1);
        R.s0 = bnd[y];
        const int xs = x0 - 8 + 8 * pl;
        if (!edge) {
            const gu32_a1 *gw = reinterpret_cast<const gu32_a1 *>(g + (size_t)y * W + xs);
            R.g0 = gw[0];
            R.g1 = gw[1];
            const TP *tp = t + (size_t)y * W + xs;
            if constexpr (sizeof(TP) == 4) {
                const gfloat4_a4 v0 = *reinterpret_cast<const gfloat4_a4 *>(tp), v1 = *reinterpret_cast<const gfloat4_a4 *>(tp + 4);
                R.tv[0] = v0.x; R.tv[1] = v0.y; R.tv[2] = v0.z; R.tv[3] = v0.w;
                R.tv[4] = v1.x; R.tv[5] = v1.y; R.tv[6] = v1.z; R.tv[7] = v1.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    const gdouble2_a8 v = *reinterpret_cast<const gdouble2_a8 *>(tp + i);
                    R.tv[i] = v.x; R.tv[i + 1] = v.y;
                }
            }
        } else {
            R.g0 = R.g1 = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const size_t q = (size_t)y * W + reflect101(xs + i, W);
                const uint32_t gq = g[q];
                if (i < 4) R.g0 |= gq << (8 * i);
                else R.g1 |= gq << (8 * (i - 4));
                R.tv[i] = t[q];
            }
        }
    };
    auto put = [&](const RawRun &R) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t gq = ((i < 4 ? R.g0 : R.g1) >> (8 * (i & 3))) & 255u;
            stage[li * 33 + 8 * pl + i] = make_double2(ilut[gq], (double)R.tv[i]);
        }
    };
    // plane pl of a staged element {I, p}: I, p, I * p, I * I -- the two factors are read from the element at lane-constant
    // offsets (selecting them from a 16-byte read took six v_cndmask per value)
    const int off_a = pl == 1 ? 1 : 0, off_b = pl == 2 ? 1 : 0;
    const bool has_b = pl >= 2;
    auto value = [&](const double2 *e) {
        const double *d = reinterpret_cast<const double *>(e);
        const double fa = d[off_a], fb = d[off_b];
        return has_b ? fa * fb : fa;
    };

    double sum = 0.0;  // the column chain of (column li, plane pl)
    // column steps for ring rows r_first .. r_last (r_last < H): output row r - 7 each, its mean into means[r - r_first]
    auto col_steps = [&](int r_first, int r_last) {
        for (int r = r_first; r <= r_last; ++r) {
            if (r < A) continue;
            if (r == A) {  // SUM of the first K - 1 rows of the extended column, top to bottom (k_box_cols)
                sum = 0.0;
                for (int j = 0; j < K - 1; ++j) sum += ring[(reflect101(j - A, H) & 31) * kFuseRS + lane];
            }
            const int y = r - A;
            const double e = ring[(r & 31) * kFuseRS + lane], l = ring[(reflect101(y - A, H) & 31) * kFuseRS + lane];
            const double s0 = sum + e;
            means[(r - r_first) * kFuseRS + lane] = s0 * scale;
            sum = s0 - l;
        }
    };
    // a, b of output rows y_first + (0 .. n_rows - 1) from means: lane (row li, columns 4 pl .. 4 pl + 3)
    auto epilogue = [&](int y_first, int n_rows) {
        const int y = y_first + li;
        if (li >= n_rows || y < 0) return;
        double av[4], bv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double2 m01 = *reinterpret_cast<const double2 *>(means + li * kFuseRS + (4 * pl + c) * 4);
            const double2 m23 = *reinterpret_cast<const double2 *>(means + li * kFuseRS + (4 * pl + c) * 4 + 2);
            const double cov = m23.x - m01.x * m01.y;   // six_stadigy.py:34-40
            const double var = m23.y - m01.x * m01.x;
            av[c] = cov / (var + eps);
            bv[c] = m01.y - av[c] * m01.x;
        }
        const size_t o = ((size_t)b * H + y) * W + x0 + 4 * pl;
        if (x0 + 4 * pl + 4 <= W) {
            st_d2(pa + o, av[0], av[1]); st_d2(pa + o + 2, av[2], av[3]);
            st_d2(pb + o, bv[0], bv[1]); st_d2(pb + o + 2, bv[2], bv[3]);
        } else {
            for (int c = 0; c < 4 && x0 + 4 * pl + c < W; ++c) {
                pa[o + c] = av[c];
                pb[o + c] = bv[c];
            }
        }
    };

    // One block of 16 rows: `use` holds its raw data, `fill` receives the loads of the block after the next (two blocks ahead:
    // a block is shorter than a load under traffic).  The three buffers rotate through the unrolled loop below -- as register
    // copies (cur = nxt) the rotation made the wavefront wait for the youngest loads at every block.
    auto block = [&](int r0, const RawRun &use, RawRun &fill) {
        put(use);
        const double s_in = use.s0;
        gwave_lds_sync();
        if (r0 + 32 < H) fetch(r0 + 32, fill);
        // ---- rows: lane (row li, plane pl)
        if (r0 + li < H) {
            double s = s_in;
            const double2 *srow = stage + li * 33;
            double *rrow = ring + ((r0 + li) & 31) * kFuseRS + pl;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (c > 0 || strip > 0) s += value(srow + c + K) - value(srow + c);  // column 0 of the frame is the initial sum itself
                rrow[4 * c] = s;
            }
        }
        gwave_lds_sync();
        // ---- columns: lane (column li, plane pl)
        const int r_last = min(r0 + 15, H - 1);
        if (r0 >= 16 && r0 + 15 < H) {  // whole block away from the top: the 32 ring reads first, then the chain
            double e[16], l[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                e[i] = ring[((r0 + i) & 31) * kFuseRS + lane];
                l[i] = ring[((r0 + i - 2 * A) & 31) * kFuseRS + lane];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const double s0 = sum + e[i];
                means[i * kFuseRS + lane] = s0 * scale;
                sum = s0 - l[i];
            }
        } else {
            col_steps(r0, r_last);
        }
        gwave_lds_sync();
        epilogue(r0 - A, r_last - r0 + 1);
        gwave_lds_sync();
    };
    RawRun b0, b1, b2;
    fetch(0, b0);
    fetch(16, b1);  // (rows past the end are clamped: H >= 16)
    for (int r0 = 0; r0 < H; r0 += 48) {
        block(r0, b0, b2);
        if (r0 + 16 < H) block(r0 + 16, b1, b0);
        if (r0 + 32 < H) block(r0 + 32, b2, b1);
    }
    // ---- the last A output rows: their entering rows are reflections of rows the ring still holds
    {
        const int y_first = H - A;
        for (int y = y_first; y < H; ++y) {
            const double e = ring[(reflect101(y + K - 1 - A, H) & 31) * kFuseRS + lane], l = ring[(reflect101(y - A, H) & 31) * kFuseRS + lane];
            const double s0 = sum + e;
            means[(y - y_first) * kFuseRS + lane] = s0 * scale;
            sum = s0 - l;
        }
        gwave_lds_sync();
        epilogue(y_first, A);
    }
}

template <class Epi>
static void launch_cols(const double *rs, size_t n, Epi epi, Shape s, int k, hipStream_t st)
{
    const dim3 grid(cdiv((long long)s.B * s.W, 64)), blk(64);
    UWIE_PROF(Epi::kName, st);
    if (k == 15) hipLaunchKernelGGL((k_box_cols<Epi, 15>), grid, blk, 0, st, rs, n, epi, s.B, s.H, s.W, k);
    else hipLaunchKernelGGL((k_box_cols<Epi, 0>), grid, blk, 0, st, rs, n, epi, s.B, s.H, s.W, k);
}

// first half of estimate_transmission: six_stadigy.py:170-174 / enhancement_strategies.py:221-225
__global__ void __launch_bounds__(256) k_trans_init(const uint8_t *__restrict__ in, const int32_t *__restrict__ kind,
                                                    const float *__restrict__ A, int npx, float omega, float norm_eps,
                                                    int pre_clip, float *__restrict__ t0)
{
    // img / (A + eps) takes 256 values per channel: the IEEE divisions are done once per block into an LDS table
    // (three per thread) instead of three per pixel, which made this kernel instruction-bound.
    __shared__ float nrm[3][256];
    const int b = blockIdx.y;
    const int k = kind ? kind[b] : 0;
    const uint8_t *img = in + (size_t)b * npx * 3;
    const bool aligned = (npx & 3) == 0;
    for (int i = threadIdx.x; i < 768; i += 256) {
        const int c = i >> 8;
        nrm[c][i & 255] = px_val(i & 255, px_atten(k, c)) / (A[b * 3 + c] + norm_eps);
    }
    __syncthreads();
    float *trow = t0 + (size_t)b * npx;
    auto t_of = [&](const Px4 &v, int i) {
        const float dark = fminf(fminf(nrm[0][v.r[i]], nrm[1][v.g[i]]), nrm[2][v.b[i]]);
        float t = 1.0f - omega * dark;
        if (pre_clip) t = fminf(fmaxf(t, 0.1f), 1.0f);
        return t;
    };
#ifndef UWIE_TI_PREFETCH
#define UWIE_TI_PREFETCH 1
#endif
    if (UWIE_TI_PREFETCH && aligned && npx >= 4) {
        // whole groups only; the next trip's group is loaded unconditionally from a clamped position (a load under a branch is
        // followed by its own wait: DESIGN section 7 item 4)
        const int stride = gridDim.x * 1024, plast = npx - 4;
        int p = (blockIdx.x * 256 + threadIdx.x) * 4;
        // (the three words stay as loaded until their trip: an unpack next to the load would wait for it at once)
        auto ld3 = [&](int q) { return *reinterpret_cast<const uint3 *>(img + (size_t)q * 3); };  // 12-byte groups, 4-byte aligned
        uint3 nx = ld3(min(p, plast));
        for (; p < npx; p += stride) {
            const uint3 w = nx;
            nx = ld3(min(p + stride, plast));
            Px4 v;
            v.r[0] = w.x & 255; v.g[0] = (w.x >> 8) & 255; v.b[0] = (w.x >> 16) & 255;
            v.r[1] = w.x >> 24; v.g[1] = w.y & 255; v.b[1] = (w.y >> 8) & 255;
            v.r[2] = (w.y >> 16) & 255; v.g[2] = w.y >> 24; v.b[2] = w.z & 255;
            v.r[3] = (w.z >> 8) & 255; v.g[3] = (w.z >> 16) & 255; v.b[3] = w.z >> 24;
            // (a nontemporal store measured the same: 0.657 / 0.662 ms against 0.673 / 0.663)
            *reinterpret_cast<float4 *>(trow + p) = make_float4(t_of(v, 0), t_of(v, 1), t_of(v, 2), t_of(v, 3));
        }
        return;
    }
    for (int p = (blockIdx.x * 256 + threadIdx.x) * 4; p < npx; p += gridDim.x * 1024) {
        const int n = min(4, npx - p);
        const Px4 v = load_px4(img + (size_t)p * 3, n, aligned);
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dark = fminf(fminf(nrm[0][v.r[i]], nrm[1][v.g[i]]), nrm[2][v.b[i]]);
            float t = 1.0f - omega * dark;
            if (pre_clip) t = fminf(fmaxf(t, 0.1f), 1.0f);
            o[i] = t;
        }
        if (aligned && n == 4) {
            *reinterpret_cast<float4 *>(trow + p) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            for (int i = 0; i < n; ++i) trow[p + i] = o[i];
        }
    }
}

}  // namespace

int launch_trans_init(const uint8_t *d_in, const int32_t *d_kind, const float *d_A, Shape s, float omega, float norm_eps,
                      int pre_clip, float *d_t0, hipStream_t st)
{
    // Every block builds the 768-entry quotient table first (three IEEE divisions per thread, a barrier): blocks that then handle
    // only a couple of groups per thread pay for it with every eighth pixel.  ~16 K blocks per call (eight rounds of the chip).
#ifndef UWIE_TI_BLOCKS
#define UWIE_TI_BLOCKS 16384
#endif
    int blocks = grid_for((s.npx() + 3) / 4, 4096);
    {
        int want = cdiv(UWIE_TI_BLOCKS, s.B);
        want = want < 32 ? 32 : want;
        if (blocks > want) blocks = want;
    }
    UWIE_LAUNCH(k_trans_init, dim3(blocks, s.B), dim3(256), 0, st, d_in, d_kind, d_A, (int)s.npx(), omega,
                       norm_eps, pre_clip, d_t0);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

size_t box_ws_bytes(Shape s)
{
    Carver c(nullptr);
    c.take<double>((size_t)s.B * s.npx());
    return c.total();
}

int launch_box_filter_f64(const double *d_src, double *d_dst, Shape s, int k, void *ws, hipStream_t st)
{
    Carver c(ws);
    const size_t n = (size_t)s.B * s.npx();
    double *rs = c.take<double>(n);
    launch_rows_planes(d_src, nullptr, s, rs, n, k, st);
    UWIE_LAUNCH_CHECK();
    launch_cols(rs, n, EpiStore1{d_dst}, s, k, st);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

size_t guided_ws_bytes(Shape s)
{
    Carver c(nullptr);
    c.take<double>((size_t)s.B * s.npx() * 6);
    return c.total();
}

template <class TP>
static int launch_guided_t(const uint8_t *d_gray, const TP *d_t0, Shape s, int k, double eps, double *d_t, void *ws, hipStream_t st)
{
    Carver c(ws);
    const size_t n = (size_t)s.B * s.npx();
    double *rs = c.take<double>(n * 6);  // 4 row-sum planes + a + b
    double *pa = rs + 4 * n, *pb = rs + 5 * n;
    if (k == 15 && s.H >= 16 && s.W >= 32 && tune().exact_fused) {
        // rows and columns of the first filter in one kernel; the row chains' values at the strips' left edges first
        const int nstrips = cdiv(s.W, 16);
        {
            UWIE_PROF("k_box_rows_lane<SrcGuideT, bounds>", st);
            hipLaunchKernelGGL((k_box_rows_lane<SrcGuideT<TP>, 15, true>), dim3(cdiv(s.H, 64), s.B), dim3(64), 0, st,
                               SrcGuideT<TP>{d_gray, d_t0, s.H, s.W}, rs, (size_t)nstrips, k);
        }
        UWIE_LAUNCH_CHECK();
        UWIE_LAUNCH(k_box_fused_ab<TP>, dim3(nstrips, 1, s.B), dim3(64), 0, st, d_gray, d_t0, rs, s.H, s.W, nstrips, eps, pa, pb);
        UWIE_LAUNCH_CHECK();
    } else {
        launch_rows_lane(SrcGuideT<TP>{d_gray, d_t0, s.H, s.W}, rs, n, k, s.B, st);
        UWIE_LAUNCH_CHECK();
        launch_cols(rs, n, EpiAB{pa, pb, eps}, s, k, st);
        UWIE_LAUNCH_CHECK();
    }
    launch_rows_planes(pa, pb, s, rs, n, k, st);
    UWIE_LAUNCH_CHECK();
    launch_cols(rs, n, EpiQ{d_gray, d_t}, s, k, st);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

int launch_guided(const uint8_t *d_gray, const float *d_t0, Shape s, int k, double eps, double *d_t, void *ws, hipStream_t st)
{
    return launch_guided_t<float>(d_gray, d_t0, s, k, eps, d_t, ws, st);
}
// float64 p (float64 images on the dict surface): the exact-order kernels only
int launch_guided_p64(const uint8_t *d_gray, const double *d_t0, Shape s, int k, double eps, double *d_t, void *ws, hipStream_t st)
{
    return launch_guided_t<double>(d_gray, d_t0, s, k, eps, d_t, ws, st);
}

}  // namespace uwie
