// guided_filter + clip (six_stadigy.py:26-46,178-180) as ONE kernel, float64, for when bit-identical OpenCV
// summation order is not required (uwie_params.gf_exact == 0, the default).
//
// The exact-order path (k_guided.hip) must materialise the running-sum planes: 6 float64 planes, ~140 B/px of HBM
// traffic.  Here a workgroup (1024 threads) owns a 64-column strip of one image and streams down its rows once,
// TH rows per step; everything between the 5 B/px input (gray u8 + t0 f32) and the 8 B/px output (t, float64) lives
// in LDS and registers.  Both box filters are evaluated vertical-first (the box is separable):
//   step m (input rows t = m*TH .. m*TH+TH-1):
//     A1  finish the previous step: a = cov/(var+eps), b = mean_p - a*mean_I -> ab ring;  q = mean_a*I + mean_b,
//         clip -> HBM                                                   (consumes the means of the previous step)
//     A2  store the prefetched input rows into the raw ring {t0, gray}; issue the prefetch of the next TH rows
//     --- barrier ---
//     A3a per-row differences derived(entering row) - derived(leaving row) of I, p, I*p, I*I, and of a, b,
//         for all TH rows at once (one task per row and column)
//     --- barrier ---
//     A3b vertical running sums: one thread per (plane, column) adds the TH differences in sequence -> vrow, v2row
//     --- barrier ---
//     B   k-tap horizontal sums of all TH rows (3520 independent tasks for k = 15)          -> mrow[TH], m2row[TH]
//     --- barrier ---
// Four barriers per TH rows, and the wide phases (A1, B) have thousands of independent tasks, so the long LDS and
// float64 latencies overlap instead of adding up row by row.  Lags: a/b row r1 = t - L, output row
// y2 = t - 2L - 1 - TH (L = k - k/2; the extra TH keeps every a/b row a step's vertical update touches already
// finished by A1).
// Window = [i - k/2, i - k/2 + k - 1] with BORDER_REFLECT_101 in both directions, exactly OpenCV's box; only the ORDER
// of the float64 additions differs from cv2.boxFilter's running sums (both add the same <= k*k terms), which moves t
// by ~1e-15.  Stated tolerance: |t - t_oracle| <= 1e-11 (tests/test_gpu_stages.py); the pipeline's u8 output stays
// within the 1-LSB bar (observed: identical).
#include "common.h"
#include "devutil.h"

#include <cstdlib>

namespace uwie {

namespace {

constexpr int kStripW = 64;
constexpr int kFastThreads = 1024;

struct FastGeom {
    int H, W, k, a, L, TH, RCraw, RCab, RCg, NC1, NCM;
    uint32_t Mraw, Mab, Mg;  // ceil(2^32 / RC): row % RC without an integer division in the row loop
    size_t lds_bytes;
};

FastGeom make_fast_geom(Shape s, int k, int TH)
{
    FastGeom g;
    g.H = s.H; g.W = s.W; g.k = k; g.TH = TH;
    g.a = k / 2;
    g.L = k - g.a;
    g.RCraw = k + TH + 2;
    g.RCab = 2 * TH + k + 3;
    g.RCg = 2 * g.L + 3 * TH + 3;  // A1 reads row y2 while A2 (same barrier interval) stores rows up to tb+TH-1
    g.NC1 = kStripW + k - 1;
    g.NCM = kStripW + 4 * (k - 1);
    const size_t doubles = (size_t)g.RCraw * g.NCM            // raw ring: float2 {t0, gray}
                           + (size_t)TH * 4 * g.NCM           // vrow
                           + (size_t)TH * 4 * g.NC1           // mrow
                           + (size_t)g.RCab * 2 * g.NC1       // ab ring
                           + (size_t)TH * 2 * g.NC1           // v2row
                           + (size_t)TH * 2 * kStripW         // m2row
                           + 256                              // ilut
                           + ((size_t)g.RCg * kStripW + 7) / 8;  // gray ring (bytes)
    g.lds_bytes = doubles * sizeof(double);
    g.Mraw = (uint32_t)(((1ull << 32) + g.RCraw - 1) / g.RCraw);
    g.Mab = (uint32_t)(((1ull << 32) + g.RCab - 1) / g.RCab);
    g.Mg = (uint32_t)(((1ull << 32) + g.RCg - 1) / g.RCg);
    return g;
}

// row % rc for 0 <= row < 2^20 (m = ceil(2^32 / rc))
__device__ __forceinline__ int fast_mod(int row, int rc, uint32_t m)
{
    int r = row - (int)__umulhi((uint32_t)row, m) * rc;
    return r < 0 ? r + rc : r;
}

// k-tap sum.  For the window widths the reference uses (six_stadigy.py:234,245,255 and config.py: 10, 15, 20) the loop
// is fully unrolled: all loads issue back to back and the adds form a short tree, one LDS round trip instead of k/3.
template <int K>
__device__ __forceinline__ double ksum_fixed(const double *q)
{
    double v[K];
#pragma unroll
    for (int j = 0; j < K; ++j) v[j] = q[j];
#pragma unroll
    for (int w = 1; w < K; w *= 2)
#pragma unroll
        for (int j = 0; j + w < K; j += 2 * w) v[j] += v[j + w];
    return v[0];
}

__device__ __forceinline__ double ksum(const double *q, int k)
{
    if (k == 15) return ksum_fixed<15>(q);
    if (k == 20) return ksum_fixed<20>(q);
    if (k == 10) return ksum_fixed<10>(q);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    int j = 0;
    for (; j + 2 < k; j += 3) {
        s0 += q[j];
        s1 += q[j + 1];
        s2 += q[j + 2];
    }
    for (; j < k; ++j) s0 += q[j];
    return (s0 + s1) + s2;
}

template <int TH>
__global__ void __launch_bounds__(kFastThreads) k_guided_fast(const uint8_t *__restrict__ gray,
                                                              const float *__restrict__ t0, double *__restrict__ tout,
                                                              FastGeom g, double eps)
{
    constexpr int NT = kFastThreads, TW = kStripW;
    constexpr int MAXP = 4;  // horizontal stage-1 tasks per thread: TH * 4 * NC1 <= MAXP * NT
    extern __shared__ double sm[];
    const int tid = threadIdx.x, b = blockIdx.y, x0 = blockIdx.x * TW;
    const int H = g.H, W = g.W, k = g.k, a = g.a, L = g.L, NC1 = g.NC1, NCM = g.NCM;
    float2 *raw = reinterpret_cast<float2 *>(sm);       // [RCraw][NCM]  {t0, gray}
    double *vrow = sm + (size_t)g.RCraw * NCM;          // [TH][4][NCM]  vertical sums of I, p, I*p, I*I
    double *mrow = vrow + (size_t)TH * 4 * NCM;         // [TH][4][NC1]  their window means
    double *abr = mrow + (size_t)TH * 4 * NC1;          // [RCab][2][NC1]
    double *v2row = abr + (size_t)g.RCab * 2 * NC1;     // [TH][2][NC1]  vertical sums of a, b
    double *m2row = v2row + (size_t)TH * 2 * NC1;       // [TH][2][TW]
    double *ilut = m2row + (size_t)TH * 2 * TW;         // [256]
    uint8_t *gring = reinterpret_cast<uint8_t *>(ilut + 256);  // [RCg][TW] gray of the output columns
    if (tid < 256) ilut[tid] = (double)tid / 255.0;     // six_stadigy.py:177

    const int cmin = max(0, x0 - 2 * (k - 1)), cmax = min(W, x0 + TW + 2 * (k - 1)), ncin = cmax - cmin;
    const double scale = 1.0 / ((double)k * (double)k);
    const size_t img = (size_t)b * H * W;
    const int nrow1 = 4 * NC1;  // horizontal stage-1 tasks per row: (plane, a/b column)

    // ---- static task descriptors
    // ingest: thread -> (row-in-step, input column)
    const int in_i = tid / ncin, in_c = tid - in_i * ncin;
    const bool in_ok = in_i < TH;
    // horizontal stage 1: task = i * nrow1 + pl * NC1 + col
    int h1_i[MAXP], h1_src[MAXP], h1_dst[MAXP], h1_base[MAXP];
    bool h1_fast[MAXP], h1_need[MAXP], h1_ok[MAXP];
#pragma unroll
    for (int j = 0; j < MAXP; ++j) {
        const int task = tid + NT * j;
        h1_ok[j] = task < TH * nrow1;
        const int i = h1_ok[j] ? task / nrow1 : 0, rem = h1_ok[j] ? task - i * nrow1 : 0;
        const int pl = rem / NC1, col = rem - pl * NC1;
        const int xe = x0 - a + col, xr = reflect101(xe, W);
        h1_i[j] = i;
        h1_src[j] = (i * 4 + pl) * NCM;
        h1_dst[j] = (i * 4 + pl) * NC1 + col;
        h1_base[j] = xr - a;
        h1_need[j] = h1_ok[j] && xe <= W - 1 + (k - 1 - a);  // columns past that feed no real output pixel
        h1_fast[j] = xr - a >= 0 && xr - a + k - 1 < W;
    }
    // horizontal stage 2: task = i * 2*TW + pl * TW + x  (TH * 128 <= NT)
    const bool h2_ok = tid < TH * 2 * TW;
    const int h2_i = tid / (2 * TW), h2_pl = (tid / TW) & 1, h2_x = tid & (TW - 1);
    // finish (A1): a/b task = i * NC1 + col (threads 0 .. TH*NC1-1); output task = i * TW + x (threads from NT-TH*TW)
    const int ab_i = tid / NC1, ab_c = tid - ab_i * NC1;
    const bool ab_ok = ab_i < TH;
    const int out_t = tid - (NT - TH * TW);
    const bool out_ok = out_t >= 0;
    const int out_i = out_ok ? out_t / TW : 0, out_x = out_ok ? out_t & (TW - 1) : 0;

    auto derive = [&](const float2 r, double *d) {
        const double I = ilut[(int)r.y], p = (double)r.x;
        d[0] = I;
        d[1] = p;
        d[2] = I * p;
        d[3] = I * I;
    };

    double V1 = 0.0, V2 = 0.0;  // running vertical sums of this thread's (plane, column)
    float pre_t = 0.f;
    uint32_t pre_g = 0;
    if (in_ok && in_i < H) {
        const size_t idx = img + (size_t)in_i * W + cmin + in_c;
        pre_t = t0[idx];
        pre_g = gray[idx];
    }
    __syncthreads();

    const int t_end = H + 2 * L + TH;  // last tick that still produces an output row
    for (int tb = 0; tb <= t_end + TH; tb += TH) {
        // ================= A1: finish the previous step =================
        if (tb > 0) {
            const int tp = tb - TH;
            if (ab_ok) {
                const int r1 = tp + ab_i - L;
                if (r1 >= 0 && r1 < H) {
                    const double *m = mrow + (size_t)ab_i * 4 * NC1 + ab_c;
                    const double mI = m[0], mp = m[NC1], mIp = m[2 * NC1], mII = m[3 * NC1];
                    const double cov = mIp - mI * mp, var = mII - mI * mI;
                    const double av = cov / (var + eps);
                    double *dst = abr + (size_t)fast_mod(r1, g.RCab, g.Mab) * 2 * NC1;
                    dst[ab_c] = av;
                    dst[NC1 + ab_c] = mp - av * mI;
                }
            }
            if (out_ok) {
                const int y2 = tp + out_i - 2 * L - 1 - TH;
                if (y2 >= 0 && y2 < H && x0 + out_x < W) {
                    const double *m = m2row + (size_t)out_i * 2 * TW;
                    const double I = ilut[gring[fast_mod(y2, g.RCg, g.Mg) * TW + out_x]];
                    const double q = m[out_x] * I + m[TW + out_x];
                    tout[img + (size_t)y2 * W + x0 + out_x] = fmin(fmax(q, 0.1), 1.0);
                }
            }
        }
        // ================= A2: ring store + prefetch =================
        if (in_ok) {
            const int t = tb + in_i;
            if (t < H) {
                raw[(size_t)fast_mod(t, g.RCraw, g.Mraw) * NCM + in_c] = make_float2(pre_t, (float)pre_g);
                const int xo = cmin + in_c - x0;
                if (xo >= 0 && xo < TW) gring[fast_mod(t, g.RCg, g.Mg) * TW + xo] = (uint8_t)pre_g;
            }
            const int tn = t + TH;
            if (tn < H) {
                const size_t idx = img + (size_t)tn * W + cmin + in_c;
                pre_t = t0[idx];
                pre_g = gray[idx];
            }
        }
        __syncthreads();
        // ================= A3a: per-row differences (entering - leaving), all rows of the step in parallel =================
        if (in_ok) {  // stage 1: task (row-in-step in_i, input column in_c)
            const int r1 = tb + in_i - L;
            if (r1 > 0 && r1 < H) {
                double e[4], l[4];
                derive(raw[(size_t)fast_mod(reflect101(r1 - a + k - 1, H), g.RCraw, g.Mraw) * NCM + in_c], e);
                derive(raw[(size_t)fast_mod(reflect101(r1 - 1 - a, H), g.RCraw, g.Mraw) * NCM + in_c], l);
                double *v = vrow + (size_t)in_i * 4 * NCM + in_c;
                v[0] = e[0] - l[0];
                v[NCM] = e[1] - l[1];
                v[2 * NCM] = e[2] - l[2];
                v[3 * NCM] = e[3] - l[3];
            }
        }
        if (ab_ok) {  // stage 2: task (row-in-step ab_i, a/b column ab_c)
            const int y2 = tb + ab_i - 2 * L - 1 - TH;
            if (y2 > 0 && y2 < H) {
                const double *e = abr + (size_t)fast_mod(reflect101(y2 - a + k - 1, H), g.RCab, g.Mab) * 2 * NC1;
                const double *l = abr + (size_t)fast_mod(reflect101(y2 - 1 - a, H), g.RCab, g.Mab) * 2 * NC1;
                double *v = v2row + (size_t)ab_i * 2 * NC1 + ab_c;
                v[0] = e[ab_c] - l[ab_c];
                v[NC1] = e[NC1 + ab_c] - l[NC1 + ab_c];
            }
        }
        __syncthreads();
        // ================= A3b: running sums down the TH rows, one thread per (plane, column) =================
        if (tid < 4 * ncin) {
            const int pl = tid / ncin, c = tid - pl * ncin;
#pragma unroll
            for (int i = 0; i < TH; ++i) {
                const int r1 = tb + i - L;
                if (r1 < 0 || r1 >= H) continue;
                double *v = vrow + (size_t)(i * 4 + pl) * NCM + c;
                if (r1 == 0) {
                    double acc = 0.0, d[4];
                    for (int j = 0; j < k; ++j) {
                        derive(raw[(size_t)fast_mod(reflect101(j - a, H), g.RCraw, g.Mraw) * NCM + c], d);
                        acc += pl == 0 ? d[0] : pl == 1 ? d[1] : pl == 2 ? d[2] : d[3];
                    }
                    V1 = acc;
                } else {
                    V1 += *v;
                }
                *v = V1;
            }
        } else if (tid >= 512 && tid < 512 + 2 * NC1) {
            const int u = tid - 512, pl = u / NC1, c = u - pl * NC1;
#pragma unroll
            for (int i = 0; i < TH; ++i) {
                const int y2 = tb + i - 2 * L - 1 - TH;
                if (y2 < 0 || y2 >= H) continue;
                double *v = v2row + (size_t)(i * 2 + pl) * NC1 + c;
                if (y2 == 0) {
                    double acc = 0.0;
                    for (int j = 0; j < k; ++j)
                        acc += abr[(size_t)fast_mod(reflect101(j - a, H), g.RCab, g.Mab) * 2 * NC1 + pl * NC1 + c];
                    V2 = acc;
                } else {
                    V2 += *v;
                }
                *v = V2;
            }
        }
        __syncthreads();
        // ================= B: horizontal k-tap sums of all TH rows =================
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            if (!h1_ok[j]) continue;
            const int r1 = tb + h1_i[j] - L;
            if (r1 < 0 || r1 >= H) continue;
            const double *src = vrow + h1_src[j];
            double s = 0.0;
            if (!h1_need[j]) {
            } else if (h1_fast[j]) {
                s = ksum(src + (h1_base[j] - cmin), k);
            } else {
                for (int q = 0; q < k; ++q) s += src[reflect101(h1_base[j] + q, W) - cmin];
            }
            mrow[h1_dst[j]] = s * scale;
        }
        if (h2_ok) {
            const int y2 = tb + h2_i - 2 * L - 1 - TH;
            if (y2 >= 0 && y2 < H)
                m2row[(size_t)h2_i * 2 * TW + h2_pl * TW + h2_x] =
                    ksum(v2row + (size_t)h2_i * 2 * NC1 + h2_pl * NC1 + h2_x, k) * scale;
        }
        __syncthreads();
    }
}

template <int TH>
int launch_th(const uint8_t *d_gray, const float *d_t0, Shape s, const FastGeom &g, double eps, double *d_t, hipStream_t st)
{
    // more than 64 KB of LDS has to be asked for: once per context (= device) and kernel, for the most a CU has
    uwie_ctx *ctx = current_ctx();
    if (g.lds_bytes > 64 * 1024 && !(ctx && (ctx->attr_gf_fast & TH))) {
        UWIE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_guided_fast<TH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if (ctx) ctx->attr_gf_fast |= TH;
    }
    UWIE_LAUNCH(k_guided_fast<TH>, dim3(cdiv(s.W, kStripW), s.B), dim3(kFastThreads), g.lds_bytes, st, d_gray, d_t0, d_t, g,
                eps);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

bool fits(const FastGeom &g, int TH)
{
    return g.lds_bytes <= 160 * 1024 && TH * g.NCM <= kFastThreads && TH * 4 * g.NC1 <= 4 * kFastThreads &&
           TH * g.NC1 <= kFastThreads && TH * 2 * kStripW <= kFastThreads && 512 + 2 * g.NC1 <= kFastThreads &&
           4 * g.NCM <= 512 &&
           g.H < (1 << 20);
}

}  // namespace

// Whether launch_guided_fast will take this (shape, window) with one of the k_guided_fast<TH> kernels (the wavefront kernels
// of k_guided_pipe.hip are tried first, but every shape they take is also taken here); if not, the exact-order path and its
// six float64 planes of workspace are needed.
bool guided_fast_handles(Shape s, int k)
{
    return fits(make_fast_geom(s, k, 8), 8) || fits(make_fast_geom(s, k, 4), 4) || fits(make_fast_geom(s, k, 2), 2);
}

// Returns UWIE_OK and sets *handled = 0 when the window is too wide for the LDS-resident formulation (the caller
// then uses the exact-order path).
int launch_guided_fast(const uint8_t *d_gray, const float *d_t0, Shape s, int k, double eps, double *d_t, int *handled,
                       hipStream_t st, bool ring_fx)
{
    *handled = 0;
    // software-pipelined wavefront kernels (k_guided_pipe.hip) first; tuning gf_pipe = 0 keeps the LDS-tiled strip kernel
    if (tune().gf_pipe) {
        const int rcp = launch_guided_pipe(d_gray, d_t0, s, k, eps, ring_fx ? 1 : 0, d_t, handled, st);
        if (rcp != UWIE_OK || *handled) return rcp;
    }
    const FastGeom g8 = make_fast_geom(s, k, 8), g4 = make_fast_geom(s, k, 4), g2 = make_fast_geom(s, k, 2);
    int rc = UWIE_OK;
    if (fits(g8, 8)) rc = launch_th<8>(d_gray, d_t0, s, g8, eps, d_t, st);
    else if (fits(g4, 4)) rc = launch_th<4>(d_gray, d_t0, s, g4, eps, d_t, st);
    else if (fits(g2, 2)) rc = launch_th<2>(d_gray, d_t0, s, g2, eps, d_t, st);
    else return UWIE_OK;
    if (rc == UWIE_OK) *handled = 1;
    return rc;
}

}  // namespace uwie
