// quality_assessment.QualityAssessment (quality_assessment.py:15-286): the eight no-reference scores and their weighted
// sum, as device reductions, so that picking the best of N strategies (main.py:130-146) does not leave the GPU.
//   k_qa_maps    one pass over the quantised frame: gray plane (RGB2GRAY), histograms of gray, HSV saturation and LAB
//                lightness, and the four colourfulness sums (from the float image when given, else from u8/255)
//   k_qa_lap     3x3 Laplacian of the gray plane (BORDER_REFLECT_101) as exact integer sums of l and l*l
//   Canny        the quadtree's kernels on one full-frame region per image -> edge count
//   k_qa_finish  scores in float64 from the histograms / sums
// Integer-derived quantities (gray, S, L, Canny, the threshold counts) are exact.  The floating statistics are
// evaluated in float64 from histograms and exact integer sums instead of NumPy's float32 pairwise sums, and the
// Laplacian uses l/255 for the sum of float32(g/255) terms: stated tolerance 2e-3 on the 0..100 scores
// (tests/test_gpu_stages.py).
#include "common.h"
#include "devutil.h"

namespace uwie {

namespace {

constexpr int kQaScores = 9;  // contrast, sharpness, entropy, saturation, brightness, edge_density, colorfulness, naturalness, total

struct QaWeights {
    double w[8];
};

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint64_t u = shfl_xor_u64((uint64_t)__double_as_longlong(v), o);
        v += __longlong_as_double((long long)u);
    }
    return v;
}

// grid (nblk, B), block 256.  hist: [B][3][256] (gray, S, L); csum: [B][4] doubles (sum rg, yb, rg^2, yb^2).
__global__ void __launch_bounds__(256) k_qa_maps(const LabTables *__restrict__ T, const uint8_t *__restrict__ in,
                                                 const float *__restrict__ fimg, int npx, int shift,
                                                 uint8_t *__restrict__ gray, uint32_t *__restrict__ hist,
                                                 double *__restrict__ csum)
{
    __shared__ uint32_t h[3][256];
    __shared__ int s_sdiv[256];
    __shared__ uint16_t s_gamma[256], s_cbrt[3072];
    __shared__ int s_fwd[3];
    __shared__ double s_red[4][4];
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < 768; i += 256) (&h[0][0])[i] = 0;
    s_sdiv[tid] = tid ? __double2int_rn((double)(255 << 12) / (double)tid) : 0;  // RGB2HSV_b's sdiv_table
    s_gamma[tid] = T->gamma[tid];
    for (int i = tid; i < 3072; i += 256) s_cbrt[i] = T->cbrt[i];
    if (tid < 3) s_fwd[tid] = T->fwd[3 + tid];
    __syncthreads();
    const uint8_t *img = in + (size_t)b * npx * 3;
    const float *fi = fimg ? fimg + (size_t)b * npx * 3 : nullptr;
    uint8_t *g = gray + (size_t)b * npx;
    constexpr int Lscale = (116 * 255 + 50) / 100;
    constexpr int Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int p = blockIdx.x * 256 + tid; p < npx; p += gridDim.x * 256) {
        const uint32_t r = img[(size_t)p * 3], gg = img[(size_t)p * 3 + 1], bl = img[(size_t)p * 3 + 2];
        const uint32_t gv = gray_fixed(r, gg, bl, shift);
        g[p] = (uint8_t)gv;
        const int v = max(max(r, gg), bl), vmin = min(min(r, gg), bl);
        const int sat = ((v - vmin) * s_sdiv[v] + (1 << 11)) >> 12;
        const int R = s_gamma[r], G = s_gamma[gg], Bq = s_gamma[bl];
        const int fY = s_cbrt[(R * s_fwd[0] + G * s_fwd[1] + Bq * s_fwd[2] + (1 << 11)) >> 12];
        const int L = min(max((Lscale * fY + Lshift + (1 << 14)) >> 15, 0), 255);
        atomicAdd(&h[0][gv], 1u);
        atomicAdd(&h[1][sat], 1u);
        atomicAdd(&h[2][L], 1u);
        float fr, fg, fb;
        if (fi) {
            fr = fi[(size_t)p * 3]; fg = fi[(size_t)p * 3 + 1]; fb = fi[(size_t)p * 3 + 2];
        } else {
            fr = px_norm(r); fg = px_norm(gg); fb = px_norm(bl);
        }
        const float rg = fr - fg, yb = 0.5f * (fr + fg) - fb;  // quality_assessment.py:158-159
        a0 += (double)rg; a1 += (double)yb; a2 += (double)rg * (double)rg; a3 += (double)yb * (double)yb;
    }
    a0 = wave_sum_f64(a0); a1 = wave_sum_f64(a1); a2 = wave_sum_f64(a2); a3 = wave_sum_f64(a3);
    if ((tid & 63) == 0) {
        s_red[tid >> 6][0] = a0; s_red[tid >> 6][1] = a1; s_red[tid >> 6][2] = a2; s_red[tid >> 6][3] = a3;
    }
    __syncthreads();
    if (tid < 4) atomicAdd(&csum[b * 4 + tid], (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]));
    for (int i = tid; i < 768; i += 256) {
        const uint32_t c = (&h[0][0])[i];
        if (c) atomicAdd(&hist[(size_t)b * 768 + i], c);
    }
}

// cv2.Laplacian(gray, CV_64F), ksize 1: l = up + down + left + right - 4*centre, BORDER_REFLECT_101.  lsum: [B][2] int64.
__global__ void __launch_bounds__(256) k_qa_lap(const uint8_t *__restrict__ gray, int H, int W,
                                                unsigned long long *__restrict__ lsum)
{
    __shared__ long long s_red[4][2];
    const int b = blockIdx.y, tid = threadIdx.x, npx = H * W;
    const uint8_t *g = gray + (size_t)b * npx;
    long long s1 = 0, s2 = 0;
    for (int p = blockIdx.x * 256 + tid; p < npx; p += gridDim.x * 256) {
        const int y = p / W, x = p - y * W;
        const int yu = y > 0 ? y - 1 : (H > 1 ? 1 : 0), yd = y + 1 < H ? y + 1 : (H > 1 ? H - 2 : 0);
        const int xl = x > 0 ? x - 1 : (W > 1 ? 1 : 0), xr = x + 1 < W ? x + 1 : (W > 1 ? W - 2 : 0);
        const int l = (int)g[yu * W + x] + g[yd * W + x] + g[y * W + xl] + g[y * W + xr] - 4 * (int)g[p];
        s1 += l;
        s2 += (long long)l * l;
    }
    s1 = (long long)wave_sum_u64((uint64_t)s1);
    s2 = (long long)wave_sum_u64((uint64_t)s2);
    if ((tid & 63) == 0) { s_red[tid >> 6][0] = s1; s_red[tid >> 6][1] = s2; }
    __syncthreads();
    if (tid < 2)
        atomicAdd(&lsum[b * 2 + tid], (unsigned long long)(s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid]));
}

__device__ __forceinline__ double clip100(double v) { return fmin(fmax(v, 0.0), 100.0); }

__global__ void k_qa_finish(const uint32_t *__restrict__ hist, const double *__restrict__ csum,
                            const unsigned long long *__restrict__ lsum, const uint32_t *__restrict__ edges, int B,
                            int npx, QaWeights wt, double *__restrict__ out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t *hg = hist + (size_t)b * 768, *hs = hg + 256, *hl = hg + 512;
    const double n = (double)npx;
    double mg = 0.0, ms = 0.0, mL = 0.0, ent = 0.0, dark = 0.0, bright = 0.0, oversat = 0.0;
    for (int k = 0; k < 256; ++k) {
        const float xf = px_norm(k);  // the float32 value NumPy holds for this byte
        mg += (double)hg[k] * (double)xf;
        ms += (double)hs[k] * (double)xf;
        mL += (double)hl[k] * (double)k;
        if (hg[k]) {
            const double pk = (double)hg[k] / n;
            ent -= pk * log(pk);
        }
        if (xf < 0.1f) dark += hg[k];
        if (xf > 0.9f) bright += hg[k];
        if (xf > 0.9f) oversat += hs[k];
    }
    mg /= n; ms /= n; mL /= n;
    double vg = 0.0;
    for (int k = 0; k < 256; ++k) {
        const double d = (double)px_norm(k) - mg;
        vg += (double)hg[k] * d * d;
    }
    vg /= n;
    const double l1 = (double)(long long)lsum[b * 2], l2 = (double)(long long)lsum[b * 2 + 1];
    const double vlap = (l2 - l1 * l1 / n) / n / (255.0 * 255.0);
    const double *cs = csum + b * 4;
    const double mrg = cs[0] / n, myb = cs[1] / n;
    const double vrg = fmax(cs[2] / n - mrg * mrg, 0.0), vyb = fmax(cs[3] / n - myb * myb, 0.0);
    double sc[8];
    sc[0] = clip100(sqrt(vg) / 0.5 * 100.0);                                        // quality_assessment.py:25-31
    sc[1] = clip100(vlap / 0.5 * 100.0);                                            // :45-52
    sc[2] = clip100((ent / log(2.0) - 4.0) / 4.0 * 100.0);                          // :66-73
    sc[3] = clip100(ms * 100.0);                                                    // :87-94
    sc[4] = 100.0 - clip100(fabs(mL - 128.0) / 128.0 * 100.0);                      // :108-117
    sc[5] = clip100((double)edges[b] / n / 0.2 * 100.0);                            // :131-140
    sc[6] = clip100((sqrt(vrg + vyb) + 0.3 * sqrt(mrg * mrg + myb * myb)) / 0.5 * 100.0);  // :154-175
    sc[7] = 100.0 - clip100((oversat / n + dark / n + bright / n) * 200.0);         // :189-205
    double total = 0.0;
    for (int i = 0; i < 8; ++i) {
        out[(size_t)b * kQaScores + i] = sc[i];
        total += sc[i] * wt.w[i];  // quality_assessment.py:283
    }
    out[(size_t)b * kQaScores + 8] = total;
}

struct QaBufs {
    uint8_t *gray;
    uint32_t *hist, *edges;
    double *csum;
    unsigned long long *lsum;
    Region *regs;
    void *canny;
};

QaBufs carve_qa(Carver &c, Shape s)
{
    QaBufs q;
    q.gray = c.take<uint8_t>((size_t)s.B * s.npx());
    q.hist = c.take<uint32_t>((size_t)s.B * 768);
    q.edges = c.take<uint32_t>(s.B);
    q.csum = c.take<double>((size_t)s.B * 4);
    q.lsum = c.take<unsigned long long>((size_t)s.B * 2);
    q.regs = c.take<Region>(s.B);
    q.canny = c.take<char>(canny_ws_bytes(s));
    return q;
}

// main.py:145: best_strategy = max(strategy_scores, key=strategy_scores.get) -- the first maximum in the strategies' order.
// scores: [n][B][9] (the weighted total is entry 8)
__global__ void k_pick_best(const double *__restrict__ scores, int n, int B, int32_t *__restrict__ best)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int k = 0;
    double top = scores[(size_t)b * 9 + 8];
    for (int i = 1; i < n; ++i) {
        const double v = scores[((size_t)i * B + b) * 9 + 8];
        if (v > top) { top = v; k = i; }
    }
    best[b] = k;
}

// out[b] = all[best[b]][b]: 16 bytes per thread (frames are 3 * npx bytes; the tail goes byte by byte)
__global__ void __launch_bounds__(256) k_gather_best(const uint8_t *__restrict__ all, const int32_t *__restrict__ best, size_t frame_bytes,
                                                     int B, uint8_t *__restrict__ out)
{
    const int b = blockIdx.y;
    const uint8_t *src = all + ((size_t)best[b] * B + b) * frame_bytes;
    uint8_t *dst = out + (size_t)b * frame_bytes;
    const size_t n16 = (((size_t)(src - (const uint8_t *)nullptr) | (size_t)(dst - (uint8_t *)nullptr)) & 15) ? 0 : frame_bytes / 16;
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        reinterpret_cast<uint4 *>(dst)[i] = reinterpret_cast<const uint4 *>(src)[i];
    for (size_t i = n16 * 16 + blockIdx.x * 256 + threadIdx.x; i < frame_bytes; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace

int launch_pick_best(const double *d_scores, int n, Shape s, const uint8_t *d_all, int32_t *d_best, uint8_t *d_out, hipStream_t st)
{
    UWIE_LAUNCH(k_pick_best, dim3(cdiv(s.B, 64)), dim3(64), 0, st, d_scores, n, s.B, d_best);
    UWIE_LAUNCH_CHECK();
    if (d_out) {
        UWIE_LAUNCH(k_gather_best, dim3(grid_for(s.npx() * 3 / 16 + 1, 1024), s.B), dim3(256), 0, st, d_all, d_best, s.npx() * 3, s.B, d_out);
        UWIE_LAUNCH_CHECK();
    }
    return UWIE_OK;
}

size_t quality_ws_bytes(Shape s)
{
    Carver c(nullptr);
    carve_qa(c, s);
    return c.total();
}

int launch_quality_scores(uwie_ctx *ctx, const uint8_t *d_u8, const float *d_f32, Shape s, int gray_shift,
                          const double *weights8, double *d_scores, void *ws, hipStream_t st)
{
    Carver c(ws);
    QaBufs q = carve_qa(c, s);
    const int npx = (int)s.npx();
    QaWeights wt;
    for (int i = 0; i < 8; ++i) wt.w[i] = weights8[i];
    UWIE_HIP_CHECK(hipMemsetAsync(q.hist, 0, sizeof(uint32_t) * (size_t)s.B * 768, st));
    UWIE_HIP_CHECK(hipMemsetAsync(q.csum, 0, sizeof(double) * (size_t)s.B * 4, st));
    UWIE_HIP_CHECK(hipMemsetAsync(q.lsum, 0, sizeof(unsigned long long) * (size_t)s.B * 2, st));
    const dim3 grid(grid_for(s.npx(), 1024), s.B);
    UWIE_LAUNCH(k_qa_maps, grid, dim3(256), 0, st, ctx->d_lab, d_u8, d_f32, npx, gray_shift, q.gray, q.hist, q.csum);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_qa_lap, grid, dim3(256), 0, st, q.gray, s.H, s.W, q.lsum);
    UWIE_LAUNCH_CHECK();
    int rc = launch_make_full_regions(q.regs, s, st);
    if (rc != UWIE_OK) return rc;
    rc = launch_canny(q.gray, s, q.regs, s.B, s.H, s.W, 50, 150, q.edges, nullptr, q.canny, st);
    if (rc != UWIE_OK) return rc;
    UWIE_LAUNCH(k_qa_finish, dim3(cdiv(s.B, 64)), dim3(64), 0, st, q.hist, q.csum, q.lsum, q.edges, s.B, npx, wt, d_scores);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

}  // namespace uwie
