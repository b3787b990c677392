// The 8-bit colour / histogram primitives of the hot path, as integer kernels:
//   cv2.cvtColor RGB2GRAY (six_stadigy.py:149,177), RGB2LAB / LAB2RGB (six_stadigy.py:204,207),
//   cv2.createCLAHE(clip, tiles).apply (six_stadigy.py:205-206), cv2.equalizeHist (enhancement_strategies.py:343).
// CLAHE: one workgroup per (tile, image) builds the 256-bin tile histogram in LDS (wave-private copies, reflected
// right/bottom padding when the frame is not a multiple of the grid), clips, redistributes, prefix-sums and
// writes the tile's 256-entry LUT; a second kernel blends the four neighbouring LUTs per pixel in float32 in
// OpenCV's operation order.
#include "common.h"
#include "devutil.h"

namespace uwie {

namespace {

__device__ __forceinline__ uint8_t sat_u8(int v) { return (uint8_t)min(max(v, 0), 255); }
#define UWIE_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

__device__ __forceinline__ void rgb2lab_px(const LabTables *__restrict__ T, uint32_t r8, uint32_t g8, uint32_t b8,
                                           uint8_t *o)
{
    const int R = T->gamma[r8], G = T->gamma[g8], B = T->gamma[b8];
    const int *C = T->fwd;
    const int fX = T->cbrt[UWIE_DESCALE(R * C[0] + G * C[1] + B * C[2], 12)];
    const int fY = T->cbrt[UWIE_DESCALE(R * C[3] + G * C[4] + B * C[5], 12)];
    const int fZ = T->cbrt[UWIE_DESCALE(R * C[6] + G * C[7] + B * C[8], 12)];
    constexpr int Lscale = (116 * 255 + 50) / 100;
    constexpr int Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
    o[0] = sat_u8(UWIE_DESCALE(Lscale * fY + Lshift, 15));
    o[1] = sat_u8(UWIE_DESCALE(500 * (fX - fY) + 128 * (1 << 15), 15));
    o[2] = sat_u8(UWIE_DESCALE(200 * (fY - fZ) + 128 * (1 << 15), 15));
}

__device__ __forceinline__ void lab2rgb_px(const LabTables *__restrict__ T, int LL, int aa, int bb, uint8_t *o)
{
    constexpr int BASE = 1 << 14, kMinAB = -8145;
    const int y = T->ltoyf[LL * 2], ify = T->ltoyf[LL * 2 + 1];
    const int adiv = ((5 * aa * 53687 + (1 << 7)) >> 13) - 128 * BASE / 500;
    const int bdiv = ((bb * 41943 + (1 << 4)) >> 9) - 128 * BASE / 200 + 1;
    const int x = T->abtoxz[ify + adiv - kMinAB], z = T->abtoxz[ify - bdiv - kMinAB];
    const int *C = T->inv;
    const int ro = min(max(UWIE_DESCALE(C[0] * x + C[1] * y + C[2] * z, 14), 0), 4095);
    const int go = min(max(UWIE_DESCALE(C[3] * x + C[4] * y + C[5] * z, 14), 0), 4095);
    const int bo = min(max(UWIE_DESCALE(C[6] * x + C[7] * y + C[8] * z, 14), 0), 4095);
    o[0] = T->invgamma[ro];
    o[1] = T->invgamma[go];
    o[2] = T->invgamma[bo];
}

__global__ void __launch_bounds__(256) k_rgb2gray(const uint8_t *__restrict__ rgb, uint8_t *__restrict__ gray, size_t n,
                                                  int shift)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        gray[i] = (uint8_t)gray_fixed(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], shift);
}

__global__ void __launch_bounds__(256) k_rgb2lab(const LabTables *__restrict__ T, const uint8_t *__restrict__ rgb,
                                                 uint8_t *__restrict__ lab, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint8_t o[3];
        rgb2lab_px(T, rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], o);
        lab[3 * i] = o[0]; lab[3 * i + 1] = o[1]; lab[3 * i + 2] = o[2];
    }
}

__global__ void __launch_bounds__(256) k_lab2rgb(const LabTables *__restrict__ T, const uint8_t *__restrict__ lab,
                                                 uint8_t *__restrict__ rgb, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint8_t o[3];
        lab2rgb_px(T, lab[3 * i], lab[3 * i + 1], lab[3 * i + 2], o);
        rgb[3 * i] = o[0]; rgb[3 * i + 1] = o[1]; rgb[3 * i + 2] = o[2];
    }
}

// (img*255).astype(u8) -> RGB2LAB, float32 HWC in, u8 LAB HWC out (six_stadigy.py:204)
__global__ void __launch_bounds__(256) k_quant_rgb2lab(const LabTables *__restrict__ T, const float *__restrict__ img,
                                                       uint8_t *__restrict__ lab, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint8_t o[3];
        rgb2lab_px(T, quant_u8(img[3 * i]), quant_u8(img[3 * i + 1]), quant_u8(img[3 * i + 2]), o);
        lab[3 * i] = o[0]; lab[3 * i + 1] = o[1]; lab[3 * i + 2] = o[2];
    }
}

// LAB2RGB -> .astype(float32) / 255.0 (six_stadigy.py:207)
__global__ void __launch_bounds__(256) k_lab2rgb_f32(const LabTables *__restrict__ T, const uint8_t *__restrict__ lab,
                                                     float *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint8_t o[3];
        lab2rgb_px(T, lab[3 * i], lab[3 * i + 1], lab[3 * i + 2], o);
        out[3 * i] = (float)o[0] / 255.0f;
        out[3 * i + 1] = (float)o[1] / 255.0f;
        out[3 * i + 2] = (float)o[2] / 255.0f;
    }
}

struct ClaheGeom {
    int H, W, tx, ty, tw, th;  // tw/th: tile size on the (possibly padded) frame
    int clip;                  // integer clip limit, 0 = no clipping
    float lutScale;            // 255 / tile area
};

// block-wide exclusive/inclusive helpers over 256 threads
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

// grid (tx*ty, B), block 256.  src: u8 plane with pixel stride `ps` bytes.
__global__ void __launch_bounds__(256) k_clahe_lut(const uint8_t *__restrict__ src, int ps, ClaheGeom g,
                                                   uint8_t *__restrict__ lut)
{
    __shared__ uint32_t h[4][256];
    __shared__ uint32_t wsum[4];
    const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, w = tid >> 6;
    for (int i = tid; i < 1024; i += 256) (&h[0][0])[i] = 0;
    __syncthreads();
    const int ty = tile / g.tx, txi = tile % g.tx;
    const uint8_t *plane = src + (size_t)b * g.H * g.W * ps;
    const int area = g.tw * g.th;
    for (int i = tid; i < area; i += 256) {
        const int ey = ty * g.th + i / g.tw, ex = txi * g.tw + i % g.tw;  // coordinates on the padded frame
        const int sy = reflect101(ey, g.H), sx = reflect101(ex, g.W);     // copyMakeBorder(.., BORDER_REFLECT_101)
        atomicAdd(&h[w][plane[((size_t)sy * g.W + sx) * ps]], 1u);
    }
    __syncthreads();
    uint32_t c = h[0][tid] + h[1][tid] + h[2][tid] + h[3][tid];
    if (g.clip > 0) {
        uint32_t over = c > (uint32_t)g.clip ? c - g.clip : 0;
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
    const int v = __float2int_rn((float)sum * g.lutScale);  // cvRound: half to even
    lut[((size_t)b * g.tx * g.ty + tile) * 256 + tid] = sat_u8(v);
}

// bilinear blend of the four surrounding tile LUTs; src/dst u8 planes with pixel strides
__global__ void __launch_bounds__(256) k_clahe_apply(const uint8_t *__restrict__ src, int ps,
                                                     const uint8_t *__restrict__ lut, ClaheGeom g,
                                                     uint8_t *__restrict__ dst, int pd)
{
    const int b = blockIdx.y;
    const int npx = g.H * g.W;
    const float inv_tw = 1.0f / (float)g.tw, inv_th = 1.0f / (float)g.th;
    const uint8_t *L = lut + (size_t)b * g.tx * g.ty * 256;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npx; p += gridDim.x * 256) {
        const int y = p / g.W, x = p % g.W;
        const float tyf = (float)y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf);
        int ty2 = ty1 + 1;
        const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        ty1 = max(ty1, 0);
        ty2 = min(ty2, g.ty - 1);
        const float txf = (float)x * inv_tw - 0.5f;
        int tx1 = (int)floorf(txf);
        int tx2 = tx1 + 1;
        const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
        tx1 = max(tx1, 0);
        tx2 = min(tx2, g.tx - 1);
        const int v = src[((size_t)b * npx + p) * ps];
        const float l11 = (float)L[(ty1 * g.tx + tx1) * 256 + v], l12 = (float)L[(ty1 * g.tx + tx2) * 256 + v];
        const float l21 = (float)L[(ty2 * g.tx + tx1) * 256 + v], l22 = (float)L[(ty2 * g.tx + tx2) * 256 + v];
        const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
        dst[((size_t)b * npx + p) * pd] = sat_u8(__float2int_rn(res));
    }
}

// cv2.equalizeHist: grid (B), block 256
__global__ void __launch_bounds__(256) k_eqhist_lut(const uint8_t *__restrict__ src, int npx, uint8_t *__restrict__ lut)
{
    __shared__ uint32_t h[4][256];
    __shared__ uint32_t wsum[4];
    __shared__ int first;
    const int b = blockIdx.x, tid = threadIdx.x, w = tid >> 6;
    for (int i = tid; i < 1024; i += 256) (&h[0][0])[i] = 0;
    if (tid == 0) first = 256;
    __syncthreads();
    const uint8_t *p = src + (size_t)b * npx;
    for (int i = tid; i < npx; i += 256) atomicAdd(&h[w][p[i]], 1u);
    __syncthreads();
    const uint32_t c = h[0][tid] + h[1][tid] + h[2][tid] + h[3][tid];
    if (c) atomicMin(&first, tid);
    __syncthreads();
    const int i0 = first;
    const uint32_t c0 = h[0][i0] + h[1][i0] + h[2][i0] + h[3][i0];
    const uint32_t incl = block_incl_scan_256(tid > i0 ? c : 0, wsum);
    uint8_t out;
    if (c0 == (uint32_t)npx) {
        out = (uint8_t)i0;  // constant plane: dst.setTo(i0)
    } else {
        const float scale = (256 - 1.f) / (float)(npx - (int)c0);
        out = tid <= i0 ? 0 : sat_u8(__float2int_rn((float)incl * scale));
    }
    lut[b * 256 + tid] = out;
}

__global__ void __launch_bounds__(256) k_apply_lut_u8(const uint8_t *__restrict__ src, const uint8_t *__restrict__ lut,
                                                      uint8_t *__restrict__ dst, int npx)
{
    const int b = blockIdx.y;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npx; p += gridDim.x * 256)
        dst[(size_t)b * npx + p] = lut[b * 256 + src[(size_t)b * npx + p]];
}


ClaheGeom make_geom(Shape s, double clip, int tx, int ty)
{
    ClaheGeom g;
    g.H = s.H; g.W = s.W; g.tx = tx; g.ty = ty;
    int We = s.W, He = s.H;
    if (s.W % tx != 0 || s.H % ty != 0) {  // OpenCV pads BOTH directions as soon as either is ragged
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

}  // namespace

int launch_rgb2gray_u8(const uint8_t *d_rgb, uint8_t *d_gray, size_t n, int shift, hipStream_t st)
{
    UWIE_LAUNCH(k_rgb2gray, dim3(grid_for(n)), dim3(256), 0, st, d_rgb, d_gray, n, shift);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}
int launch_rgb2lab_u8(uwie_ctx *ctx, const uint8_t *d_rgb, uint8_t *d_lab, size_t n, hipStream_t st)
{
    UWIE_LAUNCH(k_rgb2lab, dim3(grid_for(n)), dim3(256), 0, st, ctx->d_lab, d_rgb, d_lab, n);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}
int launch_lab2rgb_u8(uwie_ctx *ctx, const uint8_t *d_lab, uint8_t *d_rgb, size_t n, hipStream_t st)
{
    UWIE_LAUNCH(k_lab2rgb, dim3(grid_for(n)), dim3(256), 0, st, ctx->d_lab, d_lab, d_rgb, n);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

size_t clahe_ws_bytes(Shape s, int tx, int ty)
{
    Carver c(nullptr);
    c.take<uint8_t>((size_t)s.B * tx * ty * 256);
    c.take<uint8_t>((size_t)s.B * s.npx() * 3);
    return c.total();
}

static int clahe_plane(const uint8_t *src, int ps, uint8_t *dst, int pd, Shape s, double clip, int tx, int ty,
                       uint8_t *lut, hipStream_t st)
{
    UWIE_REQUIRE(tx >= 1 && ty >= 1 && tx * ty <= 4096, "clahe: bad tile grid");
    const ClaheGeom g = make_geom(s, clip, tx, ty);
    UWIE_LAUNCH(k_clahe_lut, dim3(tx * ty, s.B), dim3(256), 0, st, src, ps, g, lut);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_clahe_apply, dim3(grid_for(s.npx()), s.B), dim3(256), 0, st, src, ps, lut, g, dst, pd);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

int launch_clahe_plane_u8(const uint8_t *d_plane, uint8_t *d_out, Shape s, double clip, int tx, int ty, void *ws,
                          hipStream_t st)
{
    Carver c(ws);
    uint8_t *lut = c.take<uint8_t>((size_t)s.B * tx * ty * 256);
    return clahe_plane(d_plane, 1, d_out, 1, s, clip, tx, ty, lut, st);
}

// apply_clahe on a float32 HWC image (six_stadigy.py:202-208)
int launch_clahe_f32(uwie_ctx *ctx, const float *d_img, float *d_out, Shape s, double clip, int tx, int ty, void *ws,
                     hipStream_t st)
{
    Carver c(ws);
    uint8_t *lut = c.take<uint8_t>((size_t)s.B * tx * ty * 256);
    uint8_t *lab = c.take<uint8_t>((size_t)s.B * s.npx() * 3);
    const size_t n = (size_t)s.B * s.npx();
    UWIE_LAUNCH(k_quant_rgb2lab, dim3(grid_for(n)), dim3(256), 0, st, ctx->d_lab, d_img, lab, n);
    UWIE_LAUNCH_CHECK();
    int rc = clahe_plane(lab, 3, lab, 3, s, clip, tx, ty, lut, st);  // in place on the L byte of every pixel
    if (rc != UWIE_OK) return rc;
    UWIE_LAUNCH(k_lab2rgb_f32, dim3(grid_for(n)), dim3(256), 0, st, ctx->d_lab, lab, d_out, n);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

int launch_equalize_hist_u8(const uint8_t *d_plane, uint8_t *d_out, Shape s, void *ws, hipStream_t st)
{
    Carver c(ws);
    uint8_t *lut = c.take<uint8_t>((size_t)s.B * 256);
    UWIE_LAUNCH(k_eqhist_lut, dim3(s.B), dim3(256), 0, st, d_plane, (int)s.npx(), lut);
    UWIE_LAUNCH_CHECK();
    UWIE_LAUNCH(k_apply_lut_u8, dim3(grid_for(s.npx()), s.B), dim3(256), 0, st, d_plane, lut, d_out, (int)s.npx());
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

}  // namespace uwie
