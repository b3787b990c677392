// guided_filter + clip (six_stadigy.py:26-46,178-180) as ONE kernel, float64, for when bit-identical OpenCV
// summation order is not required (uwie_params.gf_exact == 0, the default).
//
// The exact-order path (k_guided.hip) must materialise the running-sum planes: 6 float64 planes, ~140 B/px of HBM
// traffic.  Here a workgroup owns a 64-column strip of one image and streams down its rows once; everything between
// the 5 B/px input (gray u8 + t0 f32) and the 8 B/px output (t, float64) lives in LDS and registers.  Both box
// filters are evaluated vertical-first (the box is separable, the order of the two passes is free here):
//   tick t, interval 1:  store the prefetched input row t into the raw ring (k+2 rows of {t0, gray})
//                        V1[4 planes][column] += derived(row t-1) - derived(row t-k-1)     (registers, running)
//                        V2[a,b][column]      += ab(row r1-2) - ab(row r1-k-2)             (registers, running)
//           interval 2:  k-tap horizontal sums of V1 -> means of I, p, I*p, I*I for a/b row r1 = t - L
//                        k-tap horizontal sums of V2 -> mean_a, mean_b for output row y2 = t - 2L - 1
//           interval 3:  a = cov/(var+eps), b = mean_p - a*mean_I -> ab ring (k+2 rows);
//                        q = mean_a*I + mean_b, clip -> HBM
// (L = k - k/2; interval 3 shares a barrier interval with the next tick's interval 1: two barriers per row; input rows are prefetched into registers 8-16 rows ahead.)
// Window = [i - k/2, i - k/2 + k - 1] with BORDER_REFLECT_101 in both directions, exactly OpenCV's box; only the ORDER
// of the float64 additions differs from cv2.boxFilter's running sums (both add the same <= k*k terms), which moves t
// by ~1e-15.  Stated tolerance: |t - t_oracle| <= 1e-11 (tests/test_gpu_stages.py); the pipeline's u8 output stays
// within the 1-LSB bar (observed: identical).
#include "common.h"
#include "devutil.h"

namespace uwie {

namespace {

constexpr int kStripW = 64;

struct FastGeom {
    int H, W, k, a, L, RCraw, RCab, NC1, NCINMAX;
    uint32_t Mraw, Mab;  // ceil(2^32 / RC): row % RC without an integer division in the row loop
    size_t lds_bytes;
};

FastGeom make_fast_geom(Shape s, int k)
{
    FastGeom g;
    g.H = s.H; g.W = s.W; g.k = k;
    g.a = k / 2;
    g.L = k - g.a;
    g.RCab = k + 3;
    g.RCraw = (k + 2 > 2 * g.L + 3) ? k + 2 : 2 * g.L + 3;
    g.NC1 = kStripW + k - 1;
    g.NCINMAX = kStripW + 4 * (k - 1);
    const size_t doubles = (size_t)g.RCraw * g.NCINMAX          // raw ring: float2 {t0, gray}
                           + 4 * (size_t)g.NCINMAX              // vrow
                           + 4 * (size_t)g.NC1                  // mrow
                           + (size_t)g.RCab * 2 * g.NC1         // ab ring
                           + 2 * (size_t)g.NC1                  // v2row
                           + 2 * (size_t)kStripW                // m2row
                           + 256;                               // ilut
    g.lds_bytes = doubles * sizeof(double);
    g.Mraw = (uint32_t)(((1ull << 32) + g.RCraw - 1) / g.RCraw);
    g.Mab = (uint32_t)(((1ull << 32) + g.RCab - 1) / g.RCab);
    return g;
}

// row % rc for 0 <= row < 2^20 (m = ceil(2^32 / rc))
__device__ __forceinline__ int fast_mod(int row, int rc, uint32_t m)
{
    int r = row - (int)__umulhi((uint32_t)row, m) * rc;
    return r < 0 ? r + rc : r;
}

__device__ __forceinline__ double ksum(const double *q, int k)
{
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

__global__ void __launch_bounds__(256) k_guided_fast(const uint8_t *__restrict__ gray, const float *__restrict__ t0,
                                                     double *__restrict__ tout, FastGeom g, double eps)
{
    extern __shared__ double sm[];
    const int tid = threadIdx.x, b = blockIdx.y, x0 = blockIdx.x * kStripW;
    const int H = g.H, W = g.W, k = g.k, a = g.a, L = g.L, NC1 = g.NC1, NCM = g.NCINMAX, TW = kStripW;
    float2 *raw = reinterpret_cast<float2 *>(sm);      // [RCraw][NCM]  {t0, gray}
    double *vrow = sm + (size_t)g.RCraw * NCM;         // [4][NCM]  vertical sums of I, p, I*p, I*I
    double *mrow = vrow + 4 * (size_t)NCM;             // [4][NC1]  their window means
    double *abr = mrow + 4 * (size_t)NC1;              // [RCab][2][NC1]
    double *v2row = abr + (size_t)g.RCab * 2 * NC1;    // [2][NC1]  vertical sums of a, b
    double *m2row = v2row + 2 * (size_t)NC1;           // [2][TW]
    double *ilut = m2row + 2 * (size_t)TW;             // [256]
    ilut[tid] = (double)tid / 255.0;                   // six_stadigy.py:177

    const int cmin = max(0, x0 - 2 * (k - 1)), cmax = min(W, x0 + TW + 2 * (k - 1)), ncin = cmax - cmin;
    const double scale = 1.0 / ((double)k * (double)k);
    const size_t img = (size_t)b * H * W;
    const int ntask1 = 4 * NC1;

    // horizontal stage-1 task geometry: task -> (plane, a/b column); its window starts at actual column xr - a
    int t1_off[2], t1_base[2];
    bool t1_fast[2], t1_need[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int task = tid + 256 * i;
        const int pl = task < ntask1 ? task / NC1 : 0, col = task < ntask1 ? task % NC1 : 0;
        const int xe = x0 - a + col;
        const int xr = reflect101(xe, W);
        t1_need[i] = task < ntask1 && xe <= W - 1 + (k - 1 - a);  // columns past that feed no real output pixel
        t1_base[i] = xr - a;
        t1_fast[i] = xr - a >= 0 && xr - a + k - 1 < W;
        t1_off[i] = pl;  // plane; column recomputed from task
    }
    auto derive = [&](const float2 r, double *d) {
        const double I = ilut[(int)r.y], p = (double)r.x;
        d[0] = I;
        d[1] = p;
        d[2] = I * p;
        d[3] = I * I;
    };

    double V1[4] = {0.0, 0.0, 0.0, 0.0}, V2[2] = {0.0, 0.0};
    // Input rows are prefetched a whole group of kGroup rows ahead into registers: a strip walks down the frame one
    // row per tick, every row is a fresh set of cache lines (and mostly a fresh page), and with three barriers per
    // tick nothing else hides that latency.
    constexpr int kGroup = 8;
    float cur_t[kGroup], nxt_t[kGroup];
    uint32_t cur_g[kGroup], nxt_g[kGroup];
#pragma unroll
    for (int i = 0; i < kGroup; ++i) {
        cur_t[i] = nxt_t[i] = 0.f;
        cur_g[i] = nxt_g[i] = 0;
        if (i < H && tid < ncin) {
            const size_t idx = img + (size_t)i * W + cmin + tid;
            cur_t[i] = t0[idx];
            cur_g[i] = gray[idx];
        }
    }
    __syncthreads();

    // a/b row and output row of tick tt (consumes the means written in tick tt's interval 2); executed at the start
    // of tick tt+1, in the same barrier interval as that tick's vertical updates (they touch disjoint LDS rows)
    auto finish_tick = [&](int tt) {
        const int r1 = tt - L, y2 = tt - 2 * L - 1;
        const bool do1 = r1 >= 0 && r1 < H, do2 = y2 >= 0 && y2 < H;
        if (do1 && tid < NC1) {
            const double mI = mrow[tid], mp = mrow[NC1 + tid], mIp = mrow[2 * NC1 + tid], mII = mrow[3 * NC1 + tid];
            const double cov = mIp - mI * mp, var = mII - mI * mI;
            const double av = cov / (var + eps);
            double *dst = abr + (size_t)fast_mod(r1, g.RCab, g.Mab) * 2 * NC1;
            dst[tid] = av;
            dst[NC1 + tid] = mp - av * mI;
        }
        if (do2 && tid >= 128 && tid < 128 + TW) {
            const int x = tid - 128;
            if (x0 + x < W) {
                const float2 r = raw[(size_t)fast_mod(y2, g.RCraw, g.Mraw) * NCM + (x0 + x - cmin)];
                const double q = m2row[x] * ilut[(int)r.y] + m2row[TW + x];
                tout[img + (size_t)y2 * W + x0 + x] = fmin(fmax(q, 0.1), 1.0);
            }
        }
    };

    // a/b row r1 lags the input by L rows; the output row lags r1 by L + 1 (one more than the window needs, so that
    // every a/b row an output window touches -- including the reflected row k/2 of an even k -- is already stored)
    for (int tg = 0; tg <= H + 2 * L; tg += kGroup) {
#pragma unroll
        for (int i = 0; i < kGroup; ++i) {
            const int row = tg + kGroup + i;
            if (row < H && tid < ncin) {
                const size_t idx = img + (size_t)row * W + cmin + tid;
                nxt_t[i] = t0[idx];
                nxt_g[i] = gray[idx];
            }
        }
#pragma unroll
      for (int gi = 0; gi < kGroup; ++gi) {
        const int t = tg + gi;
        if (t > H + 2 * L) break;
        const int r1 = t - L, y2 = t - 2 * L - 1;
        const bool do1 = r1 >= 0 && r1 < H, do2 = y2 >= 0 && y2 < H;
        // ================= interval 1: finish the previous tick; ring store, vertical running sums =================
        if (t > 0) finish_tick(t - 1);
        if (t < H && tid < ncin)
            raw[(size_t)fast_mod(t, g.RCraw, g.Mraw) * NCM + tid] = make_float2(cur_t[gi], (float)cur_g[gi]);
        if (do1 && tid < ncin) {
            // rows entering/leaving the window of a/b row r1 are <= t-1: stored in earlier ticks
            if (r1 == 0) {
                double acc[4] = {0.0, 0.0, 0.0, 0.0}, d[4];
                for (int j = 0; j < k; ++j) {
                    const int row = reflect101(j - a, H);
                    // row t (== L for even k) is being stored by this very thread in this tick: use the registers
                    derive(raw[(size_t)fast_mod(row, g.RCraw, g.Mraw) * NCM + tid], d);
                    acc[0] += d[0]; acc[1] += d[1]; acc[2] += d[2]; acc[3] += d[3];
                }
                V1[0] = acc[0]; V1[1] = acc[1]; V1[2] = acc[2]; V1[3] = acc[3];
            } else {
                double e[4], l[4];
                derive(raw[(size_t)fast_mod(reflect101(r1 - a + k - 1, H), g.RCraw, g.Mraw) * NCM + tid], e);
                derive(raw[(size_t)fast_mod(reflect101(r1 - 1 - a, H), g.RCraw, g.Mraw) * NCM + tid], l);
                V1[0] += e[0] - l[0]; V1[1] += e[1] - l[1]; V1[2] += e[2] - l[2]; V1[3] += e[3] - l[3];
            }
            vrow[tid] = V1[0];
            vrow[NCM + tid] = V1[1];
            vrow[2 * NCM + tid] = V1[2];
            vrow[3 * NCM + tid] = V1[3];
        }
        if (do2 && tid < NC1) {
            if (y2 == 0) {
                double s0 = 0.0, s1 = 0.0;
                for (int j = 0; j < k; ++j) {
                    const double *q = abr + (size_t)fast_mod(reflect101(j - a, H), g.RCab, g.Mab) * 2 * NC1;
                    s0 += q[tid];
                    s1 += q[NC1 + tid];
                }
                V2[0] = s0;
                V2[1] = s1;
            } else {
                const double *e = abr + (size_t)fast_mod(reflect101(y2 - a + k - 1, H), g.RCab, g.Mab) * 2 * NC1;
                const double *l = abr + (size_t)fast_mod(reflect101(y2 - 1 - a, H), g.RCab, g.Mab) * 2 * NC1;
                V2[0] += e[tid] - l[tid];
                V2[1] += e[NC1 + tid] - l[NC1 + tid];
            }
            v2row[tid] = V2[0];
            v2row[NC1 + tid] = V2[1];
        }
        __syncthreads();
        // ================= interval 2: horizontal k-tap sums =================
        // 4*NC1 stage-1 tasks (thread tid takes task tid; the surplus goes to threads 0..127) and 2*TW stage-2 tasks
        // (threads 128..255), so every wave runs at most two k-tap sums
        if (do1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int task = tid + 256 * i;
                if (task < ntask1 && (i == 0 || tid < 128)) {
                    const double *src = vrow + (size_t)t1_off[i] * NCM;
                    double s = 0.0;
                    if (!t1_need[i]) {
                    } else if (t1_fast[i]) {
                        s = ksum(src + (t1_base[i] - cmin), k);
                    } else {
                        for (int j = 0; j < k; ++j) s += src[reflect101(t1_base[i] + j, W) - cmin];
                    }
                    mrow[task] = s * scale;  // task == plane * NC1 + column
                }
            }
        }
        if (do2 && tid >= 128) {
            const int u = tid - 128, pl = u / TW, x = u % TW;
            m2row[u] = ksum(v2row + pl * NC1 + x, k) * scale;
        }
        __syncthreads();
      }
#pragma unroll
        for (int i = 0; i < kGroup; ++i) {
            cur_t[i] = nxt_t[i];
            cur_g[i] = nxt_g[i];
        }
    }
    finish_tick(H + 2 * L);
}

}  // namespace

// Returns UWIE_OK and sets *handled = 0 when the window is too wide for the LDS-resident formulation (the caller
// then uses the exact-order path).
int launch_guided_fast(const uint8_t *d_gray, const float *d_t0, Shape s, int k, double eps, double *d_t, int *handled,
                       hipStream_t st)
{
    const FastGeom g = make_fast_geom(s, k);
    *handled = 0;
    if (g.lds_bytes > 150 * 1024 || 4 * g.NC1 > 384 || g.NCINMAX > 256 || s.H >= (1 << 20)) return UWIE_OK;
    static thread_local size_t attr_set = 0;
    if (g.lds_bytes > 64 * 1024 && g.lds_bytes > attr_set) {
        UWIE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_guided_fast),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes));
        attr_set = g.lds_bytes;
    }
    UWIE_LAUNCH(k_guided_fast, dim3(cdiv(s.W, kStripW), s.B), dim3(256), g.lds_bytes, st, d_gray, d_t0, d_t, g, eps);
    UWIE_LAUNCH_CHECK();
    *handled = 1;
    return UWIE_OK;
}

}  // namespace uwie
