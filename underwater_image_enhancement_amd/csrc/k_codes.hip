// "Code-domain" evaluation of the strategies that never leave 8-bit data for long:
//   six_stadigy.py strategies 4, 5, 6 (S6:262-285) and enhancement_strategies.py's clahe_enhancement /
//   histogram_equalization strategies (ES:400-420, 461-474).
// Between the u8 input frame and CLAHE, and between LAB2RGB's u8 output and the final quantisation, every stage
// (u8/255, colour correction, percentile stretch, white balance, equalizeHist, gamma, quantisation) is a monotone
// per-channel map of an 8-bit code.  So the float image never has to exist: k_code_chain evaluates the chain on the
// 256 possible codes of one (image, channel) -- in float32 for the S6 surface, float64 for the ES surface, in the
// reference's operation order -- and np.percentile's order statistics come from the 256-bin histogram (a monotone map
// keeps the order, so the k-th smallest value is the map of the k-th smallest code).  The frame is then touched once
// per stage that mixes channels or neighbours (CLAHE) plus one LUT application.
#include <cmath>

#include "common.h"
#include "devutil.h"

namespace uwie {

namespace {

__device__ __forceinline__ uint8_t sat_u8(int v) { return (uint8_t)min(max(v, 0), 255); }
#define UWIE_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

// ------------------------------------------------------------------ histograms of u8 HWC frames
// grid (nblk, B): hist[b][c][256] += counts
__global__ void __launch_bounds__(256) k_frame_hist(const uint8_t *__restrict__ in, int npx, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[4][768];
    const int b = blockIdx.y, tid = threadIdx.x, w = tid >> 6;
    for (int i = tid; i < 4 * 768; i += 256) (&h[0][0])[i] = 0;
    __syncthreads();
    const uint8_t *img = in + (size_t)b * npx * 3;
    const bool aligned = (npx & 3) == 0;
    for (int p = (blockIdx.x * 256 + tid) * 4; p < npx; p += gridDim.x * 1024) {  // four pixels = three dwords per thread
        const int n = min(4, npx - p);
        const Px4 v = load_px4(img + (size_t)p * 3, n, aligned);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < n) {
                atomicAdd(&h[w][v.r[i]], 1u);
                atomicAdd(&h[w][256 + v.g[i]], 1u);
                atomicAdd(&h[w][512 + v.b[i]], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < 768; i += 256) {
        const uint32_t c = h[0][i] + h[1][i] + h[2][i] + h[3][i];
        if (c) atomicAdd(&hist[(size_t)b * 768 + i], c);
    }
}

// ------------------------------------------------------------------ the chain
enum { OP_INIT = 1, OP_INITQUANT, OP_FROMCODE, OP_QUANT, OP_EQUALIZE, OP_STRETCH, OP_GAMMA, OP_INITCODE };
struct ChainOp {
    int op, mode;
    uint32_t rank[4];  // STRETCH: prev/next of the low percentile, prev/next of the high percentile
    double t[2];       // STRETCH: lerp weights; GAMMA: t[0] = exponent
};
struct ChainProg {
    int nops;
    ChainOp ops[6];
    double eps;
};

template <typename T>
__device__ __forceinline__ T np_lerp_t(T a, T b, T t)
{
    const T diff = b - a;
    T r = a + diff * t;
    if (t >= (T)0.5) r = b - diff * ((T)1 - t);
    return r;
}

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

// grid (3, B), block 256: thread u evaluates the chain for input code u of channel blockIdx.x
template <typename T>
__global__ void __launch_bounds__(256) k_code_chain(const uint32_t *__restrict__ hist, const int32_t *__restrict__ kind,
                                                    ChainProg prog, uint8_t *__restrict__ lut_code,
                                                    float *__restrict__ lut_val, double *__restrict__ lut_val64 = nullptr)
{
    __shared__ uint32_t wsum[4], eqh[256], total_px;
    __shared__ T os[4];
    __shared__ int first;
    const int c = blockIdx.x, b = blockIdx.y, u = threadIdx.x;
    const bool atten = px_atten(kind ? kind[b] : 0, c);
    const uint32_t h = hist[(size_t)(b * 3 + c) * 256 + u];
    const uint32_t incl = block_incl_scan_256(h, wsum);
    const uint32_t excl = incl - h;
    if (u == 255) total_px = incl;
    __syncthreads();
    T val = 0;
    int code = u;
    for (int k = 0; k < prog.nops; ++k) {
        const ChainOp &op = prog.ops[k];
        switch (op.op) {
        case OP_INIT:  // x = u8/255 [* 0.85]: float32 like the reference frame
            val = (T)px_val(u, atten);
            break;
        case OP_INITQUANT:  // (x * 255).astype(u8) of the float32 frame
            code = quant_u8(px_val(u, atten));
            break;
        case OP_INITCODE:  // the frame IS (img * 255).astype(u8) of a general float image (k_float.hip)
            code = u;
            break;
        case OP_FROMCODE:  // .astype(T) / 255.0
            val = (T)code / (T)255;
            break;
        case OP_QUANT:  // (val * 255).astype(u8), in val's precision
            code = (int)(val * (T)255) & 0xff;
            break;
        case OP_EQUALIZE: {  // cv2.equalizeHist on the current codes (ES:343)
            __syncthreads();
            eqh[u] = 0;
            if (u == 0) first = 256;
            __syncthreads();
            if (h) atomicAdd(&eqh[code], h);
            __syncthreads();
            const uint32_t cnt = eqh[u];
            if (cnt) atomicMin(&first, u);
            __syncthreads();
            const int i0 = first;
            const uint32_t c0 = eqh[i0];
            const uint32_t cum = block_incl_scan_256(u > i0 ? cnt : 0, wsum);
            int mapped;
            if (c0 == total_px) {
                mapped = i0;  // constant plane: dst.setTo(i0)
            } else {
                const float scale = (256 - 1.f) / (float)(int)(total_px - c0);
                mapped = u <= i0 ? 0 : sat_u8(__float2int_rn((float)cum * scale));
            }
            __syncthreads();
            eqh[u] = (uint32_t)mapped;  // reuse as the equalisation LUT
            __syncthreads();
            code = (int)eqh[code];
            __syncthreads();
            break;
        }
        case OP_STRETCH: {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (h && excl <= op.rank[j] && op.rank[j] < incl) os[j] = val;
            __syncthreads();
            const T lo = np_lerp_t<T>(os[0], os[1], (T)op.t[0]), hi = np_lerp_t<T>(os[2], os[3], (T)op.t[1]);
            const T v = (val - lo) / ((hi - lo) + (T)prog.eps);
            val = v < (T)0 ? (T)0 : v > (T)1 ? (T)1 : v;
            break;
        }
        case OP_GAMMA: {
            T v;
            if (sizeof(T) == 4) v = (T)(float)pow((double)val, (double)(float)op.t[0]);  // float32 power, rounded once
            else v = (T)pow((double)val, op.t[0]);
            if (op.mode == 2) v = v < (T)0 ? (T)0 : v > (T)1 ? (T)1 : v;
            val = v;
            break;
        }
        default:
            break;
        }
    }
    if (lut_code) lut_code[(size_t)(b * 3 + c) * 256 + u] = (uint8_t)code;
    if (lut_val) lut_val[(size_t)(b * 3 + c) * 256 + u] = (float)val;
    if (lut_val64) lut_val64[(size_t)(b * 3 + c) * 256 + u] = (double)val;
}

// ------------------------------------------------------------------ frame kernels
// (the LAB / CLAHE stages are k_fused.hip's tile and cell kernels in their code-domain modes: launch_codes_lab_lut,
// launch_clahe_apply_codes)

// u8 HWC codes -> outputs through per-(image, channel) LUTs.  grid (n, B)
__global__ void __launch_bounds__(256) k_apply_lut3(const uint8_t *__restrict__ codes, const uint8_t *__restrict__ fin_code,
                                                    const float *__restrict__ fin_val, int npx,
                                                    uint8_t *__restrict__ out_u8, float *__restrict__ out_f32,
                                                    const double *__restrict__ fin_val64 = nullptr,
                                                    double *__restrict__ out_f64 = nullptr)
{
    __shared__ float s_ff[768];
    __shared__ uint8_t s_fu[768];
    const int tid = threadIdx.x, b = blockIdx.y;
    for (int i = tid; i < 768; i += 256) {
        s_fu[i] = fin_code[(size_t)b * 768 + i];
        s_ff[i] = fin_val[(size_t)b * 768 + i];
    }
    __syncthreads();
    const size_t base = (size_t)b * npx * 3;
    const bool aligned = (npx & 3) == 0;  // 4-pixel groups are then dword aligned (devutil.h load_px4)
    for (int p = (blockIdx.x * 256 + tid) * 4; p < npx; p += gridDim.x * 1024) {
        const int n = min(4, npx - p);
        const Px4 v = load_px4(codes + base + (size_t)p * 3, n, aligned);
        uint32_t r[4], g[4], bl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r[i] = v.r[i];
            g[i] = 256 + v.g[i];
            bl[i] = 512 + v.b[i];
        }
        if (out_f32) {
            float *o = out_f32 + base + (size_t)p * 3;
            if (aligned && n == 4) {
                float4 *o4 = reinterpret_cast<float4 *>(o);
                o4[0] = make_float4(s_ff[r[0]], s_ff[g[0]], s_ff[bl[0]], s_ff[r[1]]);
                o4[1] = make_float4(s_ff[g[1]], s_ff[bl[1]], s_ff[r[2]], s_ff[g[2]]);
                o4[2] = make_float4(s_ff[bl[2]], s_ff[r[3]], s_ff[g[3]], s_ff[bl[3]]);
            } else {
                for (int i = 0; i < n; ++i) {
                    o[3 * i] = s_ff[r[i]];
                    o[3 * i + 1] = s_ff[g[i]];
                    o[3 * i + 2] = s_ff[bl[i]];
                }
            }
        }
        if (out_f64) {  // the dict surface's float64 image (enhancement_strategies.py:307,345): straight from the table
            double *o = out_f64 + base + (size_t)p * 3;
            const double *tb = fin_val64 + (size_t)b * 768;
            for (int i = 0; i < n; ++i) {
                o[3 * i] = tb[r[i]];
                o[3 * i + 1] = tb[g[i]];
                o[3 * i + 2] = tb[bl[i]];
            }
        }
        if (out_u8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r[i] = s_fu[r[i]];
                g[i] = s_fu[g[i]];
                bl[i] = s_fu[bl[i]];
            }
            store_px4(out_u8 + base + (size_t)p * 3, r, g, bl, n, aligned);
        }
    }
}


// np.percentile's index arithmetic: float32 for float32 data (S6 surface), float64 for float64 data (ES surface)
void stretch_op(ChainOp *op, long long n, double lo_pct, double hi_pct, bool f64)
{
    op->op = OP_STRETCH;
    op->mode = 0;
    const double q[2] = {lo_pct, hi_pct};
    for (int j = 0; j < 2; ++j) {
        if (f64) {
            const double qq = q[j] / 100.0, nm1 = (double)(n - 1), vi = nm1 * qq, p = std::floor(vi);
            if (vi >= nm1) {
                op->rank[2 * j] = op->rank[2 * j + 1] = (uint32_t)(n - 1);
                op->t[j] = 0.0;
            } else {
                op->rank[2 * j] = (uint32_t)p;
                op->rank[2 * j + 1] = (uint32_t)p + 1;
                op->t[j] = vi - p;
            }
        } else {
            const float qq = (float)q[j] / 100.0f, nm1 = (float)(n - 1), vi = nm1 * qq, p = std::floor(vi);
            if (vi >= nm1) {
                op->rank[2 * j] = op->rank[2 * j + 1] = (uint32_t)(n - 1);
                op->t[j] = 0.0;
            } else {
                op->rank[2 * j] = (uint32_t)p;
                op->rank[2 * j + 1] = (uint32_t)p + 1;
                op->t[j] = (double)(vi - p);
            }
        }
    }
}

ChainOp simple_op(int op, int mode = 0, double v = 0.0)
{
    ChainOp o{};
    o.op = op;
    o.mode = mode;
    o.t[0] = v;
    return o;
}

struct CodeBufs {
    uint32_t *hist_in, *hist_mid;  // [B][3][256]
    uint8_t *lut_a_code;           // code LUT feeding CLAHE
    uint8_t *fin_code;
    float *fin_val;
    double *fin_val64;
    uint8_t *tile_lut, *lab, *codes;
};

CodeBufs carve_codes(Carver &c, Shape s, int tx, int ty)
{
    CodeBufs b;
    const size_t n3 = (size_t)s.B * s.npx() * 3;
    b.hist_in = c.take<uint32_t>((size_t)s.B * 768);
    b.hist_mid = c.take<uint32_t>((size_t)s.B * 768);
    b.lut_a_code = c.take<uint8_t>((size_t)s.B * 768);
    b.fin_code = c.take<uint8_t>((size_t)s.B * 768);
    b.fin_val = c.take<float>((size_t)s.B * 768);
    b.fin_val64 = c.take<double>((size_t)s.B * 768);
    b.tile_lut = c.take<uint8_t>((size_t)s.B * tx * ty * 256);
    b.lab = c.take<uint8_t>(n3);
    b.codes = c.take<uint8_t>(n3);
    return b;
}

}  // namespace

size_t codes_ws_bytes(Shape s, int tx, int ty)
{
    Carver c(nullptr);
    carve_codes(c, s, tx, ty);
    return c.total();
}

// The five strategies that live in the code domain.  `f64` selects the ES arithmetic (float64 after the first
// a few thousand blocks in all, many pixels each: a block pays 3072 LDS clears and up to 768 global atomics
static dim3 hist_grid(Shape s)
{
    const int need = cdiv((long long)s.npx(), 1024);
    return dim3(std::max(1, std::min(need, std::max(16, 6144 / std::max(s.B, 1)))), s.B);
}

int launch_frame_hist(const uint8_t *d_in, Shape s, uint32_t *d_hist, hipStream_t st)
{
    UWIE_LAUNCH(k_frame_hist, hist_grid(s), dim3(256), 0, st, d_in, (int)s.npx(), d_hist);
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

// u8 quantisation, eps 1e-10, gamma = clip(x**(1/g))); otherwise S6 arithmetic (float32, eps 1e-6, x**g).
int launch_code_strategy(uwie_ctx *ctx, const uint8_t *d_in, const int32_t *d_kind, Shape s, const uwie_params *p,
                         uint8_t *d_out_u8, float *d_out_f32, void *ws, hipStream_t st, double *d_out_f64, bool quantised)
{
    Carver c(ws);
    const int tx = p->tiles_x, ty = p->tiles_y;
    CodeBufs B = carve_codes(c, s, tx, ty);
    const long long n = (long long)s.npx();
    const bool es = p->surface == UWIE_SURFACE_DICT;
    const int k = p->strategy;
    const dim3 gpx(grid_for(s.npx(), 1024), s.B), blk(256);
    const double wb = p->wb_percentile;

    UWIE_HIP_CHECK(hipMemsetAsync(B.hist_in, 0, sizeof(uint32_t) * (size_t)s.B * 768, st));
    UWIE_HIP_CHECK(hipMemsetAsync(B.hist_mid, 0, sizeof(uint32_t) * (size_t)s.B * 768, st));
    UWIE_LAUNCH(k_frame_hist, hist_grid(s), blk, 0, st, d_in, (int)n, B.hist_in);
    UWIE_LAUNCH_CHECK();

    ChainProg pre{}, post{};
    pre.eps = post.eps = es ? 1e-10 : (double)1e-6f;
    bool clahe = true, post_needs_hist = false;
    ChainOp gam = simple_op(OP_GAMMA, es ? 2 : 1, es ? 1.0 / p->gamma : (double)(float)p->gamma);
    const int init_q = quantised ? OP_INITCODE : OP_INITQUANT;  // d_in = the u8 frame, or the quantised float image
    if (quantised && !(es || k == 4)) {
        set_error("a pre-quantised frame only fits the strategies that start by quantising (S6 strategy 4, dict clahe / hist-eq)");
        return UWIE_E_INVALID;
    }
    if (!es && k == 4) {  // S6:262-268  clahe -> stretch -> white_balance -> gamma
        pre.ops[pre.nops++] = simple_op(init_q);
        post.ops[post.nops++] = simple_op(OP_FROMCODE);
        stretch_op(&post.ops[post.nops++], n, p->L_low, p->L_high, false);
        stretch_op(&post.ops[post.nops++], n, wb, 100 - wb, false);
        post.ops[post.nops++] = gam;
        post_needs_hist = true;
    } else if (!es && k == 5) {  // S6:270-277  white_balance -> stretch -> clahe -> gamma
        pre.ops[pre.nops++] = simple_op(OP_INIT);
        stretch_op(&pre.ops[pre.nops++], n, wb, 100 - wb, false);
        stretch_op(&pre.ops[pre.nops++], n, p->L_low, p->L_high, false);
        pre.ops[pre.nops++] = simple_op(OP_QUANT);
        post.ops[post.nops++] = simple_op(OP_FROMCODE);
        post.ops[post.nops++] = gam;
    } else if (!es && k == 6) {  // S6:279-285  stretch -> clahe -> gamma
        pre.ops[pre.nops++] = simple_op(OP_INIT);
        stretch_op(&pre.ops[pre.nops++], n, p->L_low, p->L_high, false);
        pre.ops[pre.nops++] = simple_op(OP_QUANT);
        post.ops[post.nops++] = simple_op(OP_FROMCODE);
        post.ops[post.nops++] = gam;
    } else if (es && k == UWIE_DICT_CLAHE_ENHANCEMENT) {  // ES:400-420
        pre.ops[pre.nops++] = simple_op(init_q);
        post.ops[post.nops++] = simple_op(OP_FROMCODE);
        stretch_op(&post.ops[post.nops++], n, p->L_low, p->L_high, true);
        if (p->apply_gamma) post.ops[post.nops++] = gam;
        post_needs_hist = true;
    } else if (es && k == UWIE_DICT_HISTOGRAM_EQUALIZATION) {  // ES:461-474: everything is one LUT of the input frame
        clahe = false;
        post.ops[post.nops++] = simple_op(init_q);
        post.ops[post.nops++] = simple_op(OP_EQUALIZE);
        post.ops[post.nops++] = simple_op(OP_FROMCODE);
        stretch_op(&post.ops[post.nops++], n, p->L_low, p->L_high, true);
        if (p->apply_gamma) post.ops[post.nops++] = gam;
    } else {
        set_error("strategy %d of surface %d is not a code-domain strategy", k, p->surface);
        return UWIE_E_INVALID;
    }
    post.ops[post.nops++] = simple_op(OP_QUANT);

    if (d_out_f64 && !es) {
        set_error("float64 output is the dict surface's (enhancement_strategies.py returns float64)");
        return UWIE_E_INVALID;
    }
    auto run_chain = [&](const ChainProg &prog, const uint32_t *hist, const int32_t *kind, uint8_t *lc, float *lv) -> int {
        if (es) UWIE_LAUNCH(k_code_chain<double>, dim3(3, s.B), blk, 0, st, hist, kind, prog, lc, lv, lv ? B.fin_val64 : (double *)nullptr);
        else UWIE_LAUNCH(k_code_chain<float>, dim3(3, s.B), blk, 0, st, hist, kind, prog, lc, lv, (double *)nullptr);
        UWIE_LAUNCH_CHECK();
        return UWIE_OK;
    };

    if (!clahe) {
        int rc = run_chain(post, B.hist_in, d_kind, B.fin_code, B.fin_val);
        if (rc != UWIE_OK) return rc;
        UWIE_LAUNCH(k_apply_lut3, gpx, blk, 0, st, d_in, B.fin_code, B.fin_val, (int)n, d_out_u8, d_out_f32,
                    (const double *)B.fin_val64, d_out_f64);
        UWIE_LAUNCH_CHECK();
        return UWIE_OK;
    }
    int rc = run_chain(pre, B.hist_in, d_kind, B.lut_a_code, nullptr);
    if (rc != UWIE_OK) return rc;
    rc = launch_codes_lab_lut(ctx, d_in, B.lut_a_code, s, p->clip_limit, tx, ty, B.lab, B.tile_lut, st);
    if (rc != UWIE_OK) return rc;
    if (post_needs_hist) {
        rc = launch_clahe_apply_codes(ctx, B.lab, B.tile_lut, s, p->clip_limit, tx, ty, nullptr, nullptr, nullptr, nullptr, B.codes,
                                      B.hist_mid, st);
        if (rc != UWIE_OK) return rc;
        rc = run_chain(post, B.hist_mid, nullptr, B.fin_code, B.fin_val);
        if (rc != UWIE_OK) return rc;
        UWIE_LAUNCH(k_apply_lut3, gpx, blk, 0, st, B.codes, B.fin_code, B.fin_val, (int)n, d_out_u8, d_out_f32,
                    (const double *)B.fin_val64, d_out_f64);
        UWIE_LAUNCH_CHECK();
        return UWIE_OK;
    }
    // the post chain has no percentile: its LUT does not depend on the image, evaluate it on a flat histogram
    rc = run_chain(post, B.hist_in, nullptr, B.fin_code, B.fin_val);
    if (rc != UWIE_OK) return rc;
    return launch_clahe_apply_codes(ctx, B.lab, B.tile_lut, s, p->clip_limit, tx, ty, B.fin_code, B.fin_val, d_out_u8, d_out_f32,
                                    nullptr, nullptr, st);
}

}  // namespace uwie
