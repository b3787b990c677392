/*
 * oracle/cvref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the seven OpenCV primitives the reference's hot path
 * calls (reference call sites: six_stadigy.py:31-43,149-150,177,204-207;
 * enhancement_strategies.py:31-43,180-181,228,299-306,343).
 *
 * OpenCV (opencv-python>=4.5.0, requirements.txt:2) is a third-party
 * dependency that is NOT vendored in /root/reference and NOT installed in this
 * image, so these functions restate OpenCV 4.x's published algorithms from
 * knowledge of its imgproc sources (box_filter.simd.hpp, color_rgb.simd.hpp,
 * color_lab.cpp, clahe.cpp, canny.cpp, histogram.cpp).  PARITY UNPINNED: no
 * golden output of cv2 exists in the reference or can be produced here; the
 * functions are pinned only by hand-derived known-answer tests
 * (tests/test_oracle_cv.py) and by the colour KATs listed in SURVEY.md A4.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  Build: `make -C oracle` (gcc -O2 -ffp-contract=off: every
 * floating-point operation below is meant to round exactly once, like the
 * SSE2 baseline build of OpenCV does).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* border handling: cv::borderInterpolate(p, len, BORDER_REFLECT_101)  */
/* ------------------------------------------------------------------ */
static int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* ------------------------------------------------------------------ */
/* cv2.boxFilter(src, CV_64F, (k,k)) on a CV_64F plane.                */
/* anchor = k/2, normalize = true, BORDER_REFLECT_101.                 */
/* RowSum<double,double>: running sum along the border-extended row:   */
/*     s = sum of first k (left to right); then s += E[i+k] - E[i].    */
/* ColumnSum<double,double>: SUM = sum of first k-1 row-sums (top to   */
/*     bottom, starting from 0); per output row: s0 = SUM + Sp;        */
/*     out = s0*scale; SUM = s0 - Sm.                                  */
/* The rounding of every output therefore depends on the whole chain   */
/* from the left edge / the top edge: it is restated literally.        */
/* ------------------------------------------------------------------ */
void cvref_box_filter_f64(const double *src, double *dst, int H, int W, int k)
{
    const int a = k / 2;
    const double scale = 1.0 / ((double)k * (double)k);
    double *rs = (double *)malloc(sizeof(double) * (size_t)H * W);
    double *ext = (double *)malloc(sizeof(double) * (size_t)(W + k));
    for (int y = 0; y < H; ++y) {
        const double *S = src + (size_t)y * W;
        for (int j = 0; j < W + k - 1; ++j) ext[j] = S[reflect101(j - a, W)];
        double s = 0;
        for (int j = 0; j < k; ++j) s += ext[j];
        double *D = rs + (size_t)y * W;
        D[0] = s;
        for (int x = 0; x < W - 1; ++x) {
            s += ext[x + k] - ext[x];
            D[x + 1] = s;
        }
    }
    double *SUM = (double *)calloc((size_t)W, sizeof(double));
    for (int j = 0; j < k - 1; ++j) {
        const double *Sp = rs + (size_t)reflect101(j - a, H) * W;
        for (int x = 0; x < W; ++x) SUM[x] += Sp[x];
    }
    for (int y = 0; y < H; ++y) {
        const double *Sp = rs + (size_t)reflect101(y + k - 1 - a, H) * W;
        const double *Sm = rs + (size_t)reflect101(y - a, H) * W;
        double *D = dst + (size_t)y * W;
        for (int x = 0; x < W; ++x) {
            double s0 = SUM[x] + Sp[x];
            D[x] = s0 * scale;
            SUM[x] = s0 - Sm[x];
        }
    }
    free(SUM);
    free(ext);
    free(rs);
}

/* ------------------------------------------------------------------ */
/* cv2.cvtColor(u8 RGB, COLOR_RGB2GRAY)                                */
/* shift == 15: OpenCV 4.x RGB2Gray<uchar> (RY15/GY15/BY15, gray_shift) */
/* shift == 14: the older yuv_shift coefficients (R2Y/G2Y/B2Y).        */
/* ------------------------------------------------------------------ */
void cvref_rgb2gray_u8(const uint8_t *src, uint8_t *dst, size_t n, int shift)
{
    int cr, cg, cb;
    if (shift == 15) { cr = 9798; cg = 19235; cb = 3735; }
    else { shift = 14; cr = 4899; cg = 9617; cb = 1868; }
    const int delta = 1 << (shift - 1);
    for (size_t i = 0; i < n; ++i) {
        int r = src[3 * i], g = src[3 * i + 1], b = src[3 * i + 2];
        dst[i] = (uint8_t)((r * cr + g * cg + b * cb + delta) >> shift);
    }
}

/* ------------------------------------------------------------------ */
/* 8-bit sRGB <-> CIE Lab, OpenCV's integer paths                       */
/* (RGB2Lab_b and Lab2RGBinteger of color_lab.cpp).                    */
/* ------------------------------------------------------------------ */
#define LAB_SHIFT 12
#define GAMMA_SHIFT 3
#define LAB_SHIFT2 (LAB_SHIFT + GAMMA_SHIFT)
#define LAB_CBRT_TAB_SIZE_B (256 * 3 / 2 * (1 << GAMMA_SHIFT))
#define INV_GAMMA_TAB_SIZE 4096
#define LAB_BASE (1 << 14)
#define MIN_AB_VALUE (-8145)
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

static uint16_t sRGBGammaTab_b[256];
static uint16_t sRGBInvGammaTab_b[INV_GAMMA_TAB_SIZE];
static uint16_t LabCbrtTab_b[LAB_CBRT_TAB_SIZE_B];
static int LabToYF_b[256 * 2];
static int abToXZ_b[LAB_BASE * 9 / 4];
static int fwdCoeffs[9], invCoeffs[9];
static int labTabsReady = 0;

/* cv::cubeRoot / softfloat cbrt: Turkowski's quartic rational approximation,
 * evaluated in double, rounded to float, exponent patched in afterwards. */
static float cv_cbrt_f32(float value)
{
    union { float f; int32_t i; } v, m;
    v.f = value;
    int ix = v.i & 0x7fffffff;
    int s = v.i & 0x80000000;
    int ex = (ix >> 23) - 127;
    int shx = ex % 3;
    shx -= shx >= 0 ? 3 : 0;
    ex = (ex - shx) / 3; /* exponent of cube root */
    v.i = (ix & ((1 << 23) - 1)) | ((shx + 127) << 23);
    double fr = v.f; /* 0.125 <= fr < 1.0 */
    fr = ((((45.2548339756803022511987494 * fr +
             192.2798368355061050458134625) * fr +
            119.1654824285581628956914143) * fr +
           13.43250139086239872172837314) * fr +
          0.1636161226585754240958355063) /
         ((((14.80884093219134573786480845 * fr +
             151.9714051044435648658557668) * fr +
            168.5254414101568283957668343) * fr +
           33.9905941350215598754191872) * fr +
          1.0);
    m.f = value;
    v.f = (float)fr;
    v.i = (v.i + (ex << 23) + s) & (m.i * 2 != 0 ? -1 : 0);
    return v.f;
}

static float apply_gamma(float x)
{
    double xd = x;
    return (float)(xd <= 809.0 / 20000.0 ? xd / (323.0 / 25.0)
                                         : pow((xd + 11.0 / 200.0) / (1.0 + 11.0 / 200.0), 12.0 / 5.0));
}
static float apply_inv_gamma(float x)
{
    double xd = x;
    return (float)(xd <= 7827.0 / 2500000.0 ? xd * (323.0 / 25.0)
                                            : pow(xd, 1.0 / (12.0 / 5.0)) * (1.0 + 11.0 / 200.0) - 11.0 / 200.0);
}
static int cv_round_f(float v) { return (int)lrintf(v); }
static int cv_round_d(double v) { return (int)lrint(v); }

static void init_lab_tabs(void)
{
    if (labTabsReady) return;
    static const double D65[3] = {0.950456, 1., 1.088754};
    static const double sRGB2XYZ_D65[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160,
                                           0.072169, 0.019334, 0.119193, 0.950227};
    static const double XYZ2sRGB_D65[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991,
                                           0.041556, 0.055648, -0.204043, 1.057311};
    const float f255 = 255.0f;
    const float intScale = (float)(255 * (1 << GAMMA_SHIFT));
    for (int i = 0; i < 256; ++i) {
        float x = (float)i / f255;
        sRGBGammaTab_b[i] = (uint16_t)cv_round_f(intScale * apply_gamma(x));
    }
    const float invScale = 1.0f / (float)INV_GAMMA_TAB_SIZE;
    for (int i = 0; i < INV_GAMMA_TAB_SIZE; ++i) {
        float x = invScale * (float)i;
        sRGBInvGammaTab_b[i] = (uint16_t)cv_round_f(f255 * apply_inv_gamma(x));
    }
    const float lthresh = 216.0f / 24389.0f; /* (6/29)^3 */
    const float lscale = 841.0f / 108.0f;    /* 7.787 */
    const float lbias = 16.0f / 116.0f;
    const float cbTabScale = 1.0f / (f255 * (float)(1 << GAMMA_SHIFT));
    const float lshift2 = (float)(1 << LAB_SHIFT2);
    for (int i = 0; i < LAB_CBRT_TAB_SIZE_B; ++i) {
        float x = cbTabScale * (float)i;
        float f = x < lthresh ? fmaf(x, lscale, lbias) : cv_cbrt_f32(x);
        LabCbrtTab_b[i] = (uint16_t)cv_round_f(lshift2 * f);
    }
    const int BASE = LAB_BASE;
    for (int i = 0; i < 256; ++i) {
        int y, ify;
        if (i <= 20) { /* 8*255/100 == 20.4 */
            y = cv_round_f((float)(i * BASE * 20 * 9) / (float)(17 * 29 * 29 * 29));
            /* fy = 7.787*yy + 16/116 with 7.787/903.3 == 1/(4*29): fy = 16/116 + L*100/(255*116).
             * (My recollection of OpenCV's source has the divisor written as 3*255*29 here, which would
             * make L<=20 discontinuous with L>=21; unverifiable without OpenCV -- the consistent form is used.) */
            ify = cv_round_f((float)BASE * (16.0f / 116.0f + (float)(i * 100) / (float)(4 * 255 * 29)));
        } else {
            float fy = (float)(i * 100 * BASE) / (float)(255 * 116) + (float)(16 * BASE) / 116.0f;
            ify = cv_round_f(fy);
            y = cv_round_f(fy * fy * fy / (float)(BASE * BASE));
        }
        LabToYF_b[i * 2] = y;
        LabToYF_b[i * 2 + 1] = ify;
    }
    for (int i = MIN_AB_VALUE; i < LAB_BASE * 9 / 4 + MIN_AB_VALUE; ++i) {
        int v;
        if (i <= 3390) /* 6/29*BASE = 3389.73 */
            v = i * 108 / 841 - BASE * 16 / 116 * 108 / 841;
        else
            v = i * i / BASE * i / BASE;
        abToXZ_b[i - MIN_AB_VALUE] = v;
    }
    const double lshift = (double)(1 << LAB_SHIFT);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            fwdCoeffs[i * 3 + j] = cv_round_d(lshift * sRGB2XYZ_D65[i * 3 + j] / D65[i]);
            /* out channel i (R,G,B) from normalised x,y,z (column j) */
            invCoeffs[i * 3 + j] = cv_round_d(lshift * XYZ2sRGB_D65[i * 3 + j] * D65[j]);
        }
    labTabsReady = 1;
}

void cvref_rgb2lab_u8(const uint8_t *src, uint8_t *dst, size_t n)
{
    init_lab_tabs();
    const int Lscale = (116 * 255 + 50) / 100;
    const int Lshift = -((16 * 255 * (1 << LAB_SHIFT2) + 50) / 100);
    const int *C = fwdCoeffs;
    for (size_t i = 0; i < n; ++i) {
        int R = sRGBGammaTab_b[src[3 * i]], G = sRGBGammaTab_b[src[3 * i + 1]], B = sRGBGammaTab_b[src[3 * i + 2]];
        int fX = LabCbrtTab_b[DESCALE(R * C[0] + G * C[1] + B * C[2], LAB_SHIFT)];
        int fY = LabCbrtTab_b[DESCALE(R * C[3] + G * C[4] + B * C[5], LAB_SHIFT)];
        int fZ = LabCbrtTab_b[DESCALE(R * C[6] + G * C[7] + B * C[8], LAB_SHIFT)];
        int L = DESCALE(Lscale * fY + Lshift, LAB_SHIFT2);
        int a = DESCALE(500 * (fX - fY) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2);
        int b = DESCALE(200 * (fY - fZ) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2);
        dst[3 * i] = (uint8_t)(L < 0 ? 0 : L > 255 ? 255 : L);
        dst[3 * i + 1] = (uint8_t)(a < 0 ? 0 : a > 255 ? 255 : a);
        dst[3 * i + 2] = (uint8_t)(b < 0 ? 0 : b > 255 ? 255 : b);
    }
}

void cvref_lab2rgb_u8(const uint8_t *src, uint8_t *dst, size_t n)
{
    init_lab_tabs();
    const int BASE = LAB_BASE;
    const int shift = LAB_SHIFT + (14 - 12); /* lab_shift + (base_shift - inv_gamma_shift) */
    const int *C = invCoeffs;
    for (size_t i = 0; i < n; ++i) {
        int LL = src[3 * i], aa = src[3 * i + 1], bb = src[3 * i + 2];
        int y = LabToYF_b[LL * 2];
        int ify = LabToYF_b[LL * 2 + 1];
        int adiv = ((5 * aa * 53687 + (1 << 7)) >> 13) - 128 * BASE / 500;
        int bdiv = ((bb * 41943 + (1 << 4)) >> 9) - 128 * BASE / 200 + 1;
        int x = abToXZ_b[ify + adiv - MIN_AB_VALUE];
        int z = abToXZ_b[ify - bdiv - MIN_AB_VALUE];
        int ro = DESCALE(C[0] * x + C[1] * y + C[2] * z, shift);
        int go = DESCALE(C[3] * x + C[4] * y + C[5] * z, shift);
        int bo = DESCALE(C[6] * x + C[7] * y + C[8] * z, shift);
        ro = ro < 0 ? 0 : ro > INV_GAMMA_TAB_SIZE - 1 ? INV_GAMMA_TAB_SIZE - 1 : ro;
        go = go < 0 ? 0 : go > INV_GAMMA_TAB_SIZE - 1 ? INV_GAMMA_TAB_SIZE - 1 : go;
        bo = bo < 0 ? 0 : bo > INV_GAMMA_TAB_SIZE - 1 ? INV_GAMMA_TAB_SIZE - 1 : bo;
        dst[3 * i] = (uint8_t)sRGBInvGammaTab_b[ro];
        dst[3 * i + 1] = (uint8_t)sRGBInvGammaTab_b[go];
        dst[3 * i + 2] = (uint8_t)sRGBInvGammaTab_b[bo];
    }
}

/* table export so tests can inspect / cross-check the product's own tables */
void cvref_lab_tables(uint16_t *gamma256, uint16_t *invgamma4096, uint16_t *cbrt3072, int *ltoyf512,
                      int *abtoxz36864, int *fwd9, int *inv9)
{
    init_lab_tabs();
    if (gamma256) memcpy(gamma256, sRGBGammaTab_b, sizeof sRGBGammaTab_b);
    if (invgamma4096) memcpy(invgamma4096, sRGBInvGammaTab_b, sizeof sRGBInvGammaTab_b);
    if (cbrt3072) memcpy(cbrt3072, LabCbrtTab_b, sizeof LabCbrtTab_b);
    if (ltoyf512) memcpy(ltoyf512, LabToYF_b, sizeof LabToYF_b);
    if (abtoxz36864) memcpy(abtoxz36864, abToXZ_b, sizeof abToXZ_b);
    if (fwd9) memcpy(fwd9, fwdCoeffs, sizeof fwdCoeffs);
    if (inv9) memcpy(inv9, invCoeffs, sizeof invCoeffs);
}

/* ------------------------------------------------------------------ */
/* cv2.createCLAHE(clipLimit, (tilesX, tilesY)).apply(u8 plane)        */
/* ------------------------------------------------------------------ */
static uint8_t sat_u8_round(float v)
{
    long r = lrintf(v); /* round half to even, like cvRound */
    return (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
}

void cvref_clahe_u8(const uint8_t *src, uint8_t *dst, int H, int W, double clipLimitD, int tilesX, int tilesY)
{
    int We = W, He = H;
    const uint8_t *lutSrc = src;
    uint8_t *ext = NULL;
    if (W % tilesX != 0 || H % tilesY != 0) {
        /* copyMakeBorder(.., 0, tilesY-(H%tilesY), 0, tilesX-(W%tilesX), BORDER_REFLECT_101) */
        We = W + (tilesX - (W % tilesX));
        He = H + (tilesY - (H % tilesY));
        ext = (uint8_t *)malloc((size_t)We * He);
        for (int y = 0; y < He; ++y) {
            int sy = reflect101(y, H);
            for (int x = 0; x < We; ++x) ext[(size_t)y * We + x] = src[(size_t)sy * W + reflect101(x, W)];
        }
        lutSrc = ext;
    }
    const int tw = We / tilesX, th = He / tilesY;
    const int tileSizeTotal = tw * th;
    const float lutScale = (float)(256 - 1) / (float)tileSizeTotal;
    int clipLimit = 0;
    if (clipLimitD > 0.0) {
        clipLimit = (int)(clipLimitD * tileSizeTotal / 256);
        if (clipLimit < 1) clipLimit = 1;
    }
    uint8_t *lut = (uint8_t *)malloc((size_t)tilesX * tilesY * 256);
    for (int k = 0; k < tilesX * tilesY; ++k) {
        const int ty = k / tilesX, tx = k % tilesX;
        int hist[256];
        memset(hist, 0, sizeof hist);
        for (int y = 0; y < th; ++y) {
            const uint8_t *row = lutSrc + (size_t)(ty * th + y) * We + tx * tw;
            for (int x = 0; x < tw; ++x) hist[row[x]]++;
        }
        if (clipLimit > 0) {
            int clipped = 0;
            for (int i = 0; i < 256; ++i)
                if (hist[i] > clipLimit) {
                    clipped += hist[i] - clipLimit;
                    hist[i] = clipLimit;
                }
            int redistBatch = clipped / 256;
            int residual = clipped - redistBatch * 256;
            for (int i = 0; i < 256; ++i) hist[i] += redistBatch;
            if (residual != 0) {
                int residualStep = 256 / residual;
                if (residualStep < 1) residualStep = 1;
                for (int i = 0; i < 256 && residual > 0; i += residualStep, residual--) hist[i]++;
            }
        }
        int sum = 0;
        uint8_t *tl = lut + (size_t)k * 256;
        for (int i = 0; i < 256; ++i) {
            sum += hist[i];
            tl[i] = sat_u8_round((float)sum * lutScale);
        }
    }
    const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
    for (int y = 0; y < H; ++y) {
        float tyf = (float)y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf);
        int ty2 = ty1 + 1;
        float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tilesY - 1) ty2 = tilesY - 1;
        const uint8_t *p1 = lut + (size_t)ty1 * tilesX * 256;
        const uint8_t *p2 = lut + (size_t)ty2 * tilesX * 256;
        for (int x = 0; x < W; ++x) {
            float txf = (float)x * inv_tw - 0.5f;
            int tx1 = (int)floorf(txf);
            int tx2 = tx1 + 1;
            float xa = txf - (float)tx1, xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > tilesX - 1) tx2 = tilesX - 1;
            int v = src[(size_t)y * W + x];
            int ind1 = tx1 * 256 + v, ind2 = tx2 * 256 + v;
            float res = ((float)p1[ind1] * xa1 + (float)p1[ind2] * xa) * ya1 +
                        ((float)p2[ind1] * xa1 + (float)p2[ind2] * xa) * ya;
            dst[(size_t)y * W + x] = sat_u8_round(res);
        }
    }
    free(lut);
    free(ext);
}

/* ------------------------------------------------------------------ */
/* cv2.Canny(u8, low, high): aperture 3, L1 gradient.                  */
/* Sobel with BORDER_REPLICATE; magnitude rows/cols outside the image  */
/* are 0; NMS with the fixed-point tan(22.5deg); 8-connected hysteresis. */
/* ------------------------------------------------------------------ */
void cvref_canny_u8(const uint8_t *src, uint8_t *dst, int H, int W, double low_thresh, double high_thresh)
{
    if (low_thresh > high_thresh) { double t = low_thresh; low_thresh = high_thresh; high_thresh = t; }
    const int low = (int)floor(low_thresh), high = (int)floor(high_thresh);
    const int CANNY_SHIFT = 15;
    const int TG22 = (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5);
    short *dx = (short *)malloc(sizeof(short) * (size_t)H * W);
    short *dy = (short *)malloc(sizeof(short) * (size_t)H * W);
    int *mag = (int *)calloc((size_t)(H + 2) * (W + 2), sizeof(int));
    uint8_t *map = (uint8_t *)malloc((size_t)(H + 2) * (W + 2));
    const size_t ms = (size_t)W + 2;
#define PX(yy, xx) ((int)src[(size_t)((yy) < 0 ? 0 : (yy) >= H ? H - 1 : (yy)) * W + ((xx) < 0 ? 0 : (xx) >= W ? W - 1 : (xx))])
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int gx = (PX(y - 1, x + 1) - PX(y - 1, x - 1)) + 2 * (PX(y, x + 1) - PX(y, x - 1)) +
                     (PX(y + 1, x + 1) - PX(y + 1, x - 1));
            int gy = (PX(y + 1, x - 1) - PX(y - 1, x - 1)) + 2 * (PX(y + 1, x) - PX(y - 1, x)) +
                     (PX(y + 1, x + 1) - PX(y - 1, x + 1));
            dx[(size_t)y * W + x] = (short)gx;
            dy[(size_t)y * W + x] = (short)gy;
            mag[(size_t)(y + 1) * ms + x + 1] = abs(gx) + abs(gy);
        }
#undef PX
    memset(map, 1, (size_t)(H + 2) * (W + 2));
    int *stack = (int *)malloc(sizeof(int) * (size_t)H * W + 16);
    size_t sp = 0;
    for (int y = 0; y < H; ++y) {
        const int *mp = mag + (size_t)y * ms + 1, *ma = mp + ms, *mn = ma + ms;
        uint8_t *pm = map + (size_t)(y + 1) * ms + 1;
        for (int j = 0; j < W; ++j) {
            int m = ma[j];
            int keep = 0;
            if (m > low) {
                int xs = dx[(size_t)y * W + j], ys = dy[(size_t)y * W + j];
                int x = abs(xs), yv = abs(ys) << CANNY_SHIFT;
                int tg22x = x * TG22;
                if (yv < tg22x) {
                    keep = (m > ma[j - 1] && m >= ma[j + 1]);
                } else {
                    int tg67x = tg22x + (x << (CANNY_SHIFT + 1));
                    if (yv > tg67x) {
                        keep = (m > mp[j] && m >= mn[j]);
                    } else {
                        int s = (xs ^ ys) < 0 ? -1 : 1;
                        keep = (m > mp[j - s] && m > mn[j + s]);
                    }
                }
            }
            if (keep) {
                if (m > high) { pm[j] = 2; stack[sp++] = (int)((size_t)(y + 1) * ms + j + 1); }
                else pm[j] = 0;
            } else pm[j] = 1;
        }
    }
    while (sp) {
        int p = stack[--sp];
        static const int dxy[8][2] = {{-1, -1}, {-1, 0}, {-1, 1}, {0, -1}, {0, 1}, {1, -1}, {1, 0}, {1, 1}};
        for (int d = 0; d < 8; ++d) {
            int q = p + dxy[d][0] * (int)ms + dxy[d][1];
            if (map[q] == 0) { map[q] = 2; stack[sp++] = q; }
        }
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) dst[(size_t)y * W + x] = map[(size_t)(y + 1) * ms + x + 1] == 2 ? 255 : 0;
    free(stack); free(map); free(mag); free(dy); free(dx);
}

/* ------------------------------------------------------------------ */
/* cv2.equalizeHist(u8 plane)                                          */
/* ------------------------------------------------------------------ */
void cvref_equalize_hist_u8(const uint8_t *src, uint8_t *dst, size_t n)
{
    int hist[256];
    memset(hist, 0, sizeof hist);
    for (size_t i = 0; i < n; ++i) hist[src[i]]++;
    int i = 0;
    while (!hist[i]) ++i;
    const int total = (int)n;
    if (hist[i] == total) { memset(dst, i, n); return; }
    const float scale = (256 - 1.f) / (float)(total - hist[i]);
    int sum = 0;
    uint8_t lut[256];
    memset(lut, 0, sizeof lut);
    for (lut[i++] = 0; i < 256; ++i) {
        sum += hist[i];
        lut[i] = sat_u8_round((float)sum * scale);
    }
    for (size_t k = 0; k < n; ++k) dst[k] = lut[src[k]];
}


/* ---------------------------------------------------------------------------------------------------------------------
 * cvtColor(u8, COLOR_RGB2HSV), 8-bit, hue range 180 (OpenCV 4.x RGB2HSV_b): v = max, s = (diff * sdiv[v] + 2^11) >> 12
 * with sdiv[i] = saturate_cast<int>((255 << 12) / (1. * i)) (cvRound of the double quotient), h in [0, 180).
 * Used by quality_assessment.py:79,192 (saturation channel only).  PARITY UNPINNED (no OpenCV here), KAT-tested. */
void cvref_rgb2hsv_u8(const uint8_t *src, uint8_t *dst, size_t n)
{
    static int sdiv[256], hdiv[256], init = 0;
    const int hsv_shift = 12, hr = 180;
    if (!init) {
        sdiv[0] = hdiv[0] = 0;
        for (int i = 1; i < 256; ++i) {
            sdiv[i] = (int)lrint((255 << hsv_shift) / (1. * i));
            hdiv[i] = (int)lrint((hr << hsv_shift) / (6. * i));
        }
        init = 1;
    }
    for (size_t p = 0; p < n; ++p) {
        const int r = src[3 * p], g = src[3 * p + 1], b = src[3 * p + 2];
        int v = r > g ? r : g; v = v > b ? v : b;
        int vmin = r < g ? r : g; vmin = vmin < b ? vmin : b;
        const int diff = v - vmin;
        const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
        const int s = (diff * sdiv[v] + (1 << (hsv_shift - 1))) >> hsv_shift;
        int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
        h = (h * hdiv[diff] + (1 << (hsv_shift - 1))) >> hsv_shift;
        h += h < 0 ? hr : 0;
        dst[3 * p] = (uint8_t)h;
        dst[3 * p + 1] = (uint8_t)s;
        dst[3 * p + 2] = (uint8_t)v;
    }
}
