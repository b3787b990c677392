// guided_filter + clip (six_stadigy.py:26-46,178-180), float64, one autonomous wavefront per strip: no workgroup
// barriers, no shared intermediates.  Default path when uwie_params.gf_exact == 0 and the window is one of the
// reference's (10, 15, 20: six_stadigy.py:234,245,255, config.py); other widths fall back to k_guided_fast.hip.
//
// A wavefront owns 128 adjacent "slots" (lane l holds slots 2l and 2l+1 in registers) of one band of rows and walks
// down the band one row per tick.  Both box filters are evaluated vertical-first:
//   tick r1:  V1 += D(row r1+Lb) - D(row r1-1-a)          vertical running sums of I, p, I*p, I*I per slot; the two raw
//                                                          rows come from global memory (the leaving row is an L2 hit)
//             m  = window sums of V1 over slots s..s+k-1    ACROSS LANES through a wave-private LDS staging line: the lane
//                                                          writes its pair sum P and slot value, reads P of lanes
//                                                          l+1..l+k/2 (ds_read2_b64, immediate offsets): 5 LDS
//                                                          instructions per plane for k = 15, no workgroup barrier
//                                                          (LDS executes a wave's instructions in order)
//             a = cov/(var+eps), b = mean_p - a*mean_I      -> ring[r1 % (2a+1)] in LDS, addressed only by the owning lane
//             V2 += ab(row y2+Lb) - ab(row y2-1-a)          vertical running sums of a, b for output row y2 = r1 - a
//             q  = window sums of V2, * I + ..., clip       -> HBM
// Slot s holds raw column x_lo - 2a + s (reflected at the image border), a/b column x_lo - a + s and output column
// x_lo + s; a strip produces NV = 128 - 2(k-1) output columns.  cv2.boxFilter pads its SOURCE by reflection, so the
// second box filter needs a/b at reflect101(column); for even k that is not what the slot arithmetic gives at a
// virtual column, so strips that touch the left/right border replace those slots by the a/b of the mirrored real column
// (one ds_bpermute shuffle).  Rows: a/b rows exist only for real rows and the ring is indexed by reflect101(row).
// Same windows, borders and float64 products as cv2.boxFilter; only the ORDER of the additions differs.  Stated
// tolerance as for k_guided_fast.hip: |t - t_oracle| <= 1e-11 (tests/test_gpu_stages.py).
#include "common.h"
#include "devutil.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace uwie {

namespace {

constexpr int kWaveSlots = 128;

struct WaveGeom {
    int H, W, band, nbands;
};

template <int K, int NREG_ = 1>
struct WaveCfg {
    static constexpr int a = K / 2, Lb = K - 1 - a, M = K / 2;
    static constexpr int NV = kWaveSlots - 2 * (K - 1);  // output columns per strip
    static constexpr int RC = 2 * a + 1;  // ring rows: output row y2 = r1 - a needs a/b rows y2-a .. y2+Lb and, leaving, y2-1-a
    static constexpr int NL = (kWaveSlots - K) / 2 + 1;  // lanes that own a valid a/b slot; the rest share one dummy entry
    static constexpr int SW = (64 + M + 2) & ~1;          // staging line: 64 lanes + the look-ahead of the last lane
    static constexpr int region_doubles = 6 * SW;         // up to 3 planes x {P, slot-0 value}
    // The lanes without a valid a/b slot share one dummy ring entry, unless leaving it out (and masking those lanes'
    // ring accesses) lets one more wavefront fit on the CU: k = 20, 41.3 KB -> 40.6 KB, 3 -> 4 wavefronts.
    static constexpr int lds_with(int nlp) { return RC * 2 * nlp * 16 + NREG_ * region_doubles * 8; }
    static constexpr bool kDummy = (160 * 1024) / lds_with(NL) == (160 * 1024) / lds_with(NL + 1);
    static constexpr int NLp = kDummy ? NL + 1 : NL;
    static constexpr int ring_bytes = RC * 2 * NLp * 16;
    // two staging regions (alternating) unless that costs a resident wavefront per CU
    // staging regions: 1 measured faster than 2 alternating ones (6.6 vs 7.4 ms at 4K x 64, k = 15): LDS bytes per
    // wavefront decide how many strips a CU holds (five up to 31.5 KB each, four beyond: measured), and that matters
    // more than overlapping two staging round trips
    static constexpr int NREG = NREG_;
    static constexpr int lds_bytes = ring_bytes + NREG * region_doubles * 8;
};

__device__ __forceinline__ double bperm64(int addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Orders this wavefront's LDS accesses for the compiler (lane l reads what lane l+d wrote).  No instruction is emitted:
// the LDS unit executes one wavefront's instructions in order.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Window sums of two planes.  v[p][0], v[p][1] = this lane's slots 2l, 2l+1 of plane p;
// o[p][0] = sum of slots 2l .. 2l+K-1, o[p][1] = sum of slots 2l+1 .. 2l+K.
template <int K, int NREG>
__device__ __forceinline__ void window_sums2(double *reg, int lane, const double (&v)[2][2], double (&o)[2][2])
{
    using C = WaveCfg<K, NREG>;
    constexpr int M = C::M, SW = C::SW;
    double P[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        P[p] = v[p][0] + v[p][1];
        reg[(2 * p) * SW + lane] = P[p];
        reg[(2 * p + 1) * SW + lane] = v[p][0];
    }
    wave_sync();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const double *ps = reg + (2 * p) * SW + lane;
        double mid0 = ps[1], mid1 = ps[2];  // sum of P over lanes l+1 .. l+M-1, two chains
#pragma unroll
        for (int d = 3; d < M; d += 2) mid0 += ps[d];
#pragma unroll
        for (int d = 4; d < M; d += 2) mid1 += ps[d];
        const double mid = mid0 + mid1;
        const double f0 = ps[SW + M];  // slot 2(l+M)
        if constexpr (K & 1) {  // K = 2M+1: o0 = P[l..l+M-1] + v0[l+M];  o1 = v1 + P[l+1..l+M]
            o[p][0] = (P[p] + mid) + f0;
            o[p][1] = (v[p][1] + mid) + ps[M];
        } else {                // K = 2M:   o0 = P[l..l+M-1];            o1 = v1 + P[l+1..l+M-1] + v0[l+M]
            o[p][0] = P[p] + mid;
            o[p][1] = (v[p][1] + mid) + f0;
        }
    }
    if constexpr (C::NREG == 1) wave_sync();
}

// The first box filter's three planes in ONE staging round trip: the guide is a byte, so the window sums of g and g*g
// are exact integers (<= 225 * 255^2 < 2^24) and travel packed in one 64-bit word (g in the high half, g*g in the low
// half: the halves never carry into each other); p and g*p are float64.  Same outputs as window_sums2.
template <int K, int NREG>
__device__ __forceinline__ void window_sums3(double *reg, int lane, const unsigned long long (&vi)[2], const double (&v)[2][2],
                                             unsigned long long (&oi)[2], double (&o)[2][2])
{
    using C = WaveCfg<K, NREG>;
    constexpr int M = C::M, SW = C::SW;
    unsigned long long *regi = reinterpret_cast<unsigned long long *>(reg) + 4 * SW;
    double P[2];
    const unsigned long long Pi = vi[0] + vi[1];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        P[p] = v[p][0] + v[p][1];
        reg[(2 * p) * SW + lane] = P[p];
        reg[(2 * p + 1) * SW + lane] = v[p][0];
    }
    regi[lane] = Pi;
    regi[SW + lane] = vi[0];
    wave_sync();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const double *ps = reg + (2 * p) * SW + lane;
        double mid0 = ps[1], mid1 = ps[2];  // sum of P over lanes l+1 .. l+M-1, two chains
#pragma unroll
        for (int d = 3; d < M; d += 2) mid0 += ps[d];
#pragma unroll
        for (int d = 4; d < M; d += 2) mid1 += ps[d];
        const double mid = mid0 + mid1;
        const double f0 = ps[SW + M];  // slot 2(l+M)
        if constexpr (K & 1) {
            o[p][0] = (P[p] + mid) + f0;
            o[p][1] = (v[p][1] + mid) + ps[M];
        } else {
            o[p][0] = P[p] + mid;
            o[p][1] = (v[p][1] + mid) + f0;
        }
    }
    {
        const unsigned long long *ps = regi + lane;
        unsigned long long mid0 = ps[1], mid1 = ps[2];
#pragma unroll
        for (int d = 3; d < M; d += 2) mid0 += ps[d];
#pragma unroll
        for (int d = 4; d < M; d += 2) mid1 += ps[d];
        const unsigned long long mid = mid0 + mid1, f0 = ps[SW + M];
        if constexpr (K & 1) {
            oi[0] = (Pi + mid) + f0;
            oi[1] = (vi[1] + mid) + ps[M];
        } else {
            oi[0] = Pi + mid;
            oi[1] = (vi[1] + mid) + f0;
        }
    }
    if constexpr (C::NREG == 1) wave_sync();
}

// (double)g / 255.0, correctly rounded for every g in 0..255 (the identity is checked exhaustively in tests/test_cabi.py)
__device__ __forceinline__ double u8_over_255(uint32_t g)
{
    const double x = (double)g, rcp = 1.0 / 255.0;
    const double q0 = x * rcp;
    return fma(fma(-q0, 255.0, x), rcp, q0);
}

// 1/x by the hardware seed and two Newton steps: within ~2 ulp, which the kernel's 1e-11 tolerance on t covers with
// four orders of magnitude to spare; the IEEE division sequence is more than twice as long and sits on the tick's
// critical path (var + eps >= eps > 0, no special cases).
__device__ __forceinline__ double recip_nr(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    y = fma(fma(-x, y, 1.0), y, y);
    return y;
}

__device__ __forceinline__ int reflect_clamp(int p, int len)
{
    if (p < 0) p = -p;
    if (p > len - 1) p = 2 * (len - 1) - p;
    return min(max(p, 0), len - 1);
}

struct TickIn {
    float te[2], tl[2];
    uint32_t ge[2], gl[2], go[2];
};

template <int K, int R, int NREG, int WPE>
__global__ void __launch_bounds__(64, WPE) k_guided_wave(const uint8_t *__restrict__ gray, const float *__restrict__ t0,
                                                    double *__restrict__ tout, WaveGeom g, double eps)
{
    using C = WaveCfg<K, NREG>;
    constexpr int a = C::a, Lb = C::Lb, NV = C::NV, RC = C::RC, NLp = C::NLp;
    extern __shared__ double2 ring[];  // [RC][2][NLp]: {a, b} of slot 2l (half 0) and of slot 2l+1 (half 1)
    double *stage = reinterpret_cast<double *>(ring) + C::ring_bytes / 8;
    const int lane = threadIdx.x;
    const int rl = C::kDummy ? min(lane, C::NL) : lane;
    const bool own = C::kDummy || lane < C::NL;  // (values of the other lanes never reach a valid output)
    const int H = g.H, W = g.W;
    const int x_lo = blockIdx.x * NV;
    const int y_lo = blockIdx.y * g.band, y_hi = min(H, y_lo + g.band);
    const int r_lo = max(0, y_lo - a), r_hi = min(H - 1, y_hi - 1 + Lb), r_end = y_hi - 1 + a;
    const size_t img = (size_t)blockIdx.z * H * W;
    const uint8_t *gimg = gray + img;
    const float *timg = t0 + img;
    double *oimg = tout + img;
    const double scale = 1.0 / ((double)K * (double)K);

    int craw[2], cout[2], fix_addr[2];
    bool ook[2], fix_need[2], fix_odd[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int s = 2 * lane + j;
        craw[j] = reflect_clamp(x_lo - 2 * a + s, W);
        const int xo = x_lo + s;
        ook[j] = s < NV && xo < W;
        cout[j] = min(xo, W - 1);
        const int c = x_lo - a + s, cr = reflect_clamp(c, W);
        const int sp = min(max(cr - (x_lo - a), 0), kWaveSlots - 1);
        fix_need[j] = cr != c;
        fix_addr[j] = (sp >> 1) << 2;
        fix_odd[j] = sp & 1;
    }
    const bool edge = x_lo - a < 0 || x_lo - a + (kWaveSlots - K) > W - 1;  // wave-uniform
    const bool pair_store = ook[0] && ook[1];
    const uint32_t ofs_q = (uint32_t)(x_lo + 2 * lane) * 8u;

    // byte offsets as 32-bit unsigned: global loads take the row base from SGPRs and a 32-bit VGPR offset
    const uint32_t ofs_t[2] = {(uint32_t)craw[0] * 4u, (uint32_t)craw[1] * 4u};
    const uint32_t ofs_g[2] = {(uint32_t)craw[0], (uint32_t)craw[1]}, ofs_o[2] = {(uint32_t)cout[0], (uint32_t)cout[1]};
    auto load_rows = [&](const char *te, const char *tl, const uint8_t *ge, const uint8_t *gl, const uint8_t *go, TickIn &in) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            in.te[j] = *reinterpret_cast<const float *>(te + ofs_t[j]);
            in.ge[j] = ge[ofs_g[j]];
            in.tl[j] = *reinterpret_cast<const float *>(tl + ofs_t[j]);
            in.gl[j] = gl[ofs_g[j]];
            in.go[j] = go[ofs_o[j]];
        }
    };
    // inputs of ticks rb .. rb+R-1: entering raw row r1+Lb, leaving raw row r1-1-a, guide of the output row r1-a
    auto load_group = [&](int rb, TickIn *in) {
        if (rb - 1 - a >= 0 && rb + R - 1 + Lb <= H - 1) {  // no reflection: advance the row pointers
            const char *te = reinterpret_cast<const char *>(timg + (size_t)(rb + Lb) * W);
            const char *tl = reinterpret_cast<const char *>(timg + (size_t)(rb - 1 - a) * W);
            const uint8_t *ge = gimg + (size_t)(rb + Lb) * W, *gl = gimg + (size_t)(rb - 1 - a) * W,
                          *go = gimg + (size_t)(rb - a) * W;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                load_rows(te, tl, ge, gl, go, in[i]);
                te += (size_t)W * 4; tl += (size_t)W * 4;
                ge += W; gl += W; go += W;
            }
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int r1 = rb + i;
                const size_t re = (size_t)reflect_clamp(r1 + Lb, H) * W, rlv = (size_t)reflect_clamp(r1 - 1 - a, H) * W,
                             ro = (size_t)min(max(r1 - a, 0), H - 1) * W;
                load_rows(reinterpret_cast<const char *>(timg + re), reinterpret_cast<const char *>(timg + rlv), gimg + re,
                          gimg + rlv, gimg + ro, in[i]);
            }
        }
    };

    // ---- prologue: vertical sums of the band's first a/b row by direct summation
    double V1[2][2], V2[2][2];   // V1: vertical sums of p and g*p (g = the guide byte; I = g/255 is applied to the window sums)
    uint32_t Sg[2] = {0, 0}, Sgg[2] = {0, 0};  // vertical sums of g and g*g: exact integers
    V1[0][0] = V1[0][1] = V1[1][0] = V1[1][1] = 0.0;
    V2[0][0] = V2[0][1] = V2[1][0] = V2[1][1] = 0.0;
    for (int j = 0; j < K; ++j) {
        const size_t row = (size_t)reflect_clamp(r_lo - a + j, H) * W;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const uint32_t gq = gimg[row + craw[c]];
            const double p = (double)timg[row + craw[c]];
            Sg[c] += gq;
            Sgg[c] += gq * gq;
            V1[0][c] += p;
            V1[1][c] += (double)gq * p;
        }
    }
    // mean_I = sum(g) / (255 K^2), corr_I = sum(g*g) / (255^2 K^2), corr_Ip = sum(g*p) / (255 K^2)
    const double scale_g = scale * (1.0 / 255.0), scale_gg = scale * (1.0 / (255.0 * 255.0));

    int wslot = r_lo % RC;                        // ring slot of a/b row r1
    int pslot = wslot == 0 ? RC - 1 : wslot - 1;  // ring slot of a/b row r1 - 1
    int batch = 0;                                // staging region toggle

    auto store_row = [&](int y2, const double *q) {
        double *orow = oimg + (size_t)y2 * W;
        char *ob = reinterpret_cast<char *>(orow);
        if (pair_store && ((reinterpret_cast<uintptr_t>(orow + x_lo) & 15) == 0)) {
            *reinterpret_cast<double2 *>(ob + ofs_q) = make_double2(q[0], q[1]);
        } else {
            if (ook[0]) *reinterpret_cast<double *>(ob + ofs_q) = q[0];
            if (ook[1]) *reinterpret_cast<double *>(ob + ofs_q + 8u) = q[1];
        }
    };
    auto region = [&]() {
        double *r = stage + (C::NREG == 2 ? (batch & 1) * C::region_doubles : 0);
        ++batch;
        return r;
    };

    // One tick.  STEADY: a/b row and output row both incremental, in range and free of reflection: straight-line code
    // (rows are stored by the caller after the R ticks).  Otherwise the general form with wave-uniform branches.
    auto tick = [&](auto steady_tag, auto edge_tag, int r1, const TickIn &in, double *q) {
        constexpr bool STEADY = decltype(steady_tag)::value, EDGE = decltype(edge_tag)::value;
        const bool do1 = STEADY || r1 <= r_hi;
        double av[2] = {0.0, 0.0}, bv[2] = {0.0, 0.0};
        if (do1) {
            if (STEADY || r1 != r_lo) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double pe = (double)in.te[c], pl = (double)in.tl[c];
                    Sg[c] += in.ge[c] - in.gl[c];
                    Sgg[c] += in.ge[c] * in.ge[c] - in.gl[c] * in.gl[c];
                    V1[0][c] += pe - pl;
                    V1[1][c] += (double)in.ge[c] * pe - (double)in.gl[c] * pl;
                }
            }
            double m[4][2];
            {
                const unsigned long long vi[2] = {((unsigned long long)Sg[0] << 32) | Sgg[0], ((unsigned long long)Sg[1] << 32) | Sgg[1]};
                const double vin[2][2] = {{V1[0][0], V1[0][1]}, {V1[1][0], V1[1][1]}};
                unsigned long long oi[2];
                double o[2][2];
                window_sums3<K, NREG>(region(), lane, vi, vin, oi, o);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    m[0][c] = (double)(uint32_t)(oi[c] >> 32) * scale_g;
                    m[1][c] = o[0][c] * scale;
                    m[2][c] = o[1][c] * scale_g;
                    m[3][c] = (double)(uint32_t)oi[c] * scale_gg;
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double mI = m[0][c], mp = m[1][c], mIp = m[2][c], mII = m[3][c];
                const double cov = mIp - mI * mp, var = mII - mI * mI;
                av[c] = cov * recip_nr(var + eps);
                bv[c] = mp - av[c] * mI;
            }
            if (EDGE) {
                double fa[2], fb[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double a0 = bperm64(fix_addr[c], av[0]), a1 = bperm64(fix_addr[c], av[1]);
                    const double b0 = bperm64(fix_addr[c], bv[0]), b1 = bperm64(fix_addr[c], bv[1]);
                    fa[c] = fix_need[c] ? (fix_odd[c] ? a1 : a0) : av[c];
                    fb[c] = fix_need[c] ? (fix_odd[c] ? b1 : b0) : bv[c];
                }
                av[0] = fa[0]; av[1] = fa[1];
                bv[0] = fb[0]; bv[1] = fb[1];
            }
        }
        const int y2 = r1 - a;
        const bool do2 = STEADY || y2 >= y_lo, init2 = !STEADY && y2 == y_lo;
        double2 lv0 = make_double2(0.0, 0.0), lv1 = lv0;
        if (do2 && !init2) {  // leaving a/b row y2-1-a = r1-RC: read before its slot is overwritten by row r1
            const int ls = STEADY ? wslot : reflect_clamp(y2 - 1 - a, H) % RC;
            if (own) {
                lv0 = ring[(ls * 2 + 0) * NLp + rl];
                lv1 = ring[(ls * 2 + 1) * NLp + rl];
            }
        }
        if (do1) {
            if (own) {
                ring[(wslot * 2 + 0) * NLp + rl] = make_double2(av[0], bv[0]);
                ring[(wslot * 2 + 1) * NLp + rl] = make_double2(av[1], bv[1]);
            }
        }
        if (do2) {
            if (init2) {
                for (int j = 0; j < K; ++j) {
                    const int rs = reflect_clamp(y_lo - a + j, H) % RC;
                    double2 e0 = make_double2(0.0, 0.0), e1 = e0;
                    if (own) {
                        e0 = ring[(rs * 2 + 0) * NLp + rl];
                        e1 = ring[(rs * 2 + 1) * NLp + rl];
                    }
                    V2[0][0] += e0.x; V2[1][0] += e0.y;
                    V2[0][1] += e1.x; V2[1][1] += e1.y;
                }
            } else {
                double2 e0, e1;
                const int er = STEADY ? r1 - a + Lb : reflect_clamp(y2 + Lb, H);
                if ((STEADY && (K & 1)) || (do1 && er == r1)) {  // odd K: the row just computed
                    e0 = make_double2(av[0], bv[0]);
                    e1 = make_double2(av[1], bv[1]);
                } else {
                    const int es = STEADY ? pslot : er % RC;
                    e0 = e1 = make_double2(0.0, 0.0);
                    if (own) {
                        e0 = ring[(es * 2 + 0) * NLp + rl];
                        e1 = ring[(es * 2 + 1) * NLp + rl];
                    }
                }
                V2[0][0] += e0.x - lv0.x; V2[1][0] += e0.y - lv0.y;
                V2[0][1] += e1.x - lv1.x; V2[1][1] += e1.y - lv1.y;
            }
            const double vin[2][2] = {{V2[0][0], V2[0][1]}, {V2[1][0], V2[1][1]}};
            double o[2][2];
            window_sums2<K, NREG>(region(), lane, vin, o);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double I = u8_over_255(in.go[c]);
                q[c] = fmin(fmax((o[0][c] * scale) * I + o[1][c] * scale, 0.1), 1.0);
            }
            if (!STEADY) store_row(y2, q);
        }
        if (do1) {
            pslot = wslot;
            wslot = wslot + 1 == RC ? 0 : wslot + 1;
        }
    };

    TickIn nxt[R];
    load_group(r_lo, nxt);

    using T = std::true_type;
    using F = std::false_type;
    for (int rb = r_lo; rb <= r_end; rb += R) {
        TickIn cur[R];
#pragma unroll
        for (int i = 0; i < R; ++i) cur[i] = nxt[i];
        load_group(rb + R, nxt);
        const bool steady = rb - RC >= r_lo && rb + R - 1 <= r_hi;
        double q[R][2];
        if (steady && !edge) {  // one straight-line block for the R ticks, rows stored afterwards
#pragma unroll
            for (int i = 0; i < R; ++i) tick(T{}, F{}, rb + i, cur[i], q[i]);
#pragma unroll
            for (int i = 0; i < R; ++i) store_row(rb + i - a, q[i]);
        } else if (steady) {
#pragma unroll
            for (int i = 0; i < R; ++i) tick(T{}, T{}, rb + i, cur[i], q[i]);
#pragma unroll
            for (int i = 0; i < R; ++i) store_row(rb + i - a, q[i]);
        } else {
#pragma unroll
            for (int i = 0; i < R; ++i)
                if (rb + i <= r_end) tick(F{}, T{}, rb + i, cur[i], q[i]);
        }
    }
}

template <int K, int R, int NREG = 1, int WPE = 2>
int launch_wave(const uint8_t *d_gray, const float *d_t0, Shape s, double eps, double *d_t, hipStream_t st)
{
    using C = WaveCfg<K, NREG>;
    const int nstrips = cdiv(s.W, C::NV);
    // enough wavefronts to fill the chip a few times over: 256 CUs x (160 KB / LDS per wavefront) resident
    const int resident = 256 * max(1, (160 * 1024) / C::lds_bytes);
    int nbands = 1;
    static const char *env = getenv("UWIE_GF_BANDS");
    if (env) nbands = atoi(env);
    else {
        const long strips = (long)nstrips * s.B;
        // ~12 wavefronts per resident slot evens out the tail (4K x 64, k = 15: 1 band 6.44 ms, 2: 6.01, 6: 5.89, 16: 6.22;
        // every band pays 2(k-1) extra rows and a start-up sum)
        if (strips < 12L * resident) nbands = (int)cdiv((size_t)(12L * resident), (size_t)strips);
        // ... but a band should be at least eight times its 2(k-1) overlap rows long: small batches would otherwise be
        // cut into many short bands (4K x 16, k = 15: 25 bands 1.59 ms, 8 bands 1.47 ms)
        // -- as long as that still leaves three wavefronts per resident slot (a single 1080p frame needs all the bands
        // it can get)
        const int cap = std::max(1, s.H / (16 * (K - 1)));
        nbands = std::min(nbands, std::max(cap, (int)cdiv((size_t)(3L * resident), (size_t)strips)));
    }
    nbands = max(1, min(nbands, s.H / max(64, 4 * K)));
    WaveGeom g;
    g.H = s.H; g.W = s.W;
    g.band = cdiv(s.H, nbands);
    g.nbands = cdiv(s.H, g.band);
    {
        UWIE_PROF("k_guided_wave", st);
        hipLaunchKernelGGL((k_guided_wave<K, R, NREG, WPE>), dim3(nstrips, g.nbands, s.B), dim3(64), (size_t)C::lds_bytes, st, d_gray,
                           d_t0, d_t, g, eps);
    }
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

}  // namespace

int launch_guided_wave(const uint8_t *d_gray, const float *d_t0, Shape s, int k, double eps, double *d_t, int *handled,
                       hipStream_t st)
{
    *handled = 0;
    if (s.W < 2 * k || s.H < 4 * k || s.B > 65535) return UWIE_OK;
    int rc;
    static const char *env_r = getenv("UWIE_GF_R");
    const int R = env_r ? atoi(env_r) : 2;
#define UWIE_WAVE_CASE(KK)                                                                    \
    case KK:                                                                                  \
        rc = R == 4 ? launch_wave<KK, 4, 1, 1>(d_gray, d_t0, s, eps, d_t, st)                 \
                    : launch_wave<KK, 2, 1, 2>(d_gray, d_t0, s, eps, d_t, st);                \
        break;
    switch (k) {
        UWIE_WAVE_CASE(10)
        UWIE_WAVE_CASE(15)
        UWIE_WAVE_CASE(20)
    default: return UWIE_OK;
    }
#undef UWIE_WAVE_CASE
    if (rc == UWIE_OK) *handled = 1;
    return rc;
}

}  // namespace uwie
