// guided_filter + clip (six_stadigy.py:26-46,178-180) as a software-pipelined wavefront kernel: the default path for the
// reference's window widths (10, 15, 20: six_stadigy.py:234,245,255, config.py:29-53).
//
// One autonomous wavefront owns 128 adjacent "slots" (two per lane) of a band of rows and walks down one row per step,
// both box filters vertical-first with the horizontal window sums taken across lanes through a wave-private LDS staging
// line, no workgroup barrier; the row's dependent chain
//     raw rows -> vertical sums -> [LDS] -> window sums -> a, b -> ring -> vertical sums -> [LDS] -> window sums -> q
// is cut at its two LDS round trips into three phases that run on three DIFFERENT rows in one step:
//     step i:   C(i-2): window sums of V2 (staged by step i-1) -> q = mean_a*I + mean_b, clip -> HBM
//               B(i-1): window sums of V1 (staged by step i-1) -> a, b -> V2 += ab(entering) - ab(leaving)
//               A(i)  : V1 += raw(entering row) - raw(leaving row)
//               -- all LDS reads of the step are issued first, all LDS writes (stagings of V1(i), V2(i-1), ring row i-1)
//                  last, one compiler-only fence in between (LDS executes a wave's instructions in order) --
// so a step's critical path is one LDS read plus the a/b arithmetic instead of the whole chain, and the three phases
// give the scheduler independent instruction streams (the kernel is instruction-issue bound: profiles/microbench).
//
// Ring of a/b rows (k rows per slot, the only large per-strip state), two formats:
//   RING_F64  float64 pairs, 16 B per slot and row (27 KB per strip at k = 15: five wavefronts per CU)
//   RING_FX32 fixed point int32 pairs, 8 B per slot and row (14 KB: eight wavefronts per CU, two per SIMD).  a and b - b0
//             are rounded ONCE to multiples of 2^-Sa / 2^-Sb when they enter the ring; every sum after that is exact
//             (integers below 2^53 in float64), so the error of t is the mean of k*k independent roundings:
//             |a| <= amax and |b - b0| <= 0.45 + amax follow from 0.1 <= p <= 1 (pre-clipped transmission, S6:174) and
//             Cauchy-Schwarz, Sa/Sb are chosen so that |fixed| < 2^30: resolution 2^-31 .. 2^-32, error of t ~ 1e-11 rms.
//             Only offered when the transmission is pre-clipped (six_stadigy surface).
// Guide sums (sum g, sum g*g) are exact integers, sum p and sum g*p exact in float64 (multiples of 2^-27 below 2^26);
// a = cov/(var+eps) from the integer forms  var*(255 K^2)^2 = K^2*sum(gg) - sum(g)^2,
// cov*255*K^4 = K^2*sum(gp) - sum(g)*sum(p), one Newton step on v_rcp_f64 (2^-23 -> 2^-46).
// Same windows and borders as cv2.boxFilter (BORDER_REFLECT_101, anchor k/2); stated tolerance on t: 1e-11 (RING_F64),
// 5e-10 (RING_FX32) -- tests/test_gpu_stages.py.
#include "common.h"
#include "devutil.h"
#include "guided_wave.h"

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <utility>

namespace uwie {

namespace {

#define UWIE_TRY_RC(call)               \
    do {                                \
        const int _rc = (call);         \
        if (_rc != UWIE_OK) return _rc; \
    } while (0)

#ifndef UWIE_GF_MINBAND
#define UWIE_GF_MINBAND 32  // rows: 1080p x 1 (20 strips): 64-row bands 75 us, 32-row bands 56 us
#endif
constexpr int kPipeSlots = 128;
struct PipeGeom {
    int H, W, band;
    // Border launch around an interior block that k_guided_split covers (iy1 > iy0): blockIdx.y 0 = rows [0, iy0),
    // 1 = rows [iy1, H), 2.. = the interior's own bands (iband rows each) for the strips outside [is0, is1) only.
    int iy0, iy1, iband, is0, is1;
};

template <int K>
struct PipeCfg {
    static constexpr int K_ = K, a = K / 2, Lb = K - 1 - a, M = K / 2;
    static constexpr int NV = kPipeSlots - 2 * (K - 1);   // output columns per strip
    static constexpr int RC = 2 * a + 1;                  // ring rows
    static constexpr int NL = (kPipeSlots - K) / 2 + 1;   // lanes that own a valid a/b slot
    // Staging lines are 64 entries: a lane reads up to M entries past its own, and a read past the end of a line lands
    // in the next line (or the pad).  Only lanes >= NL read there, and their sums feed no stored output.
    static constexpr int SW = 64;
    static constexpr int s1_doubles = 6 * SW;             // P, v0 of {sum p, sum g*p, packed guide sums}
    static constexpr int s2_doubles = 4 * SW + 16;        // P, v0 of {a, b} + pad (the look-ahead of the last lanes)
    // lanes >= NL own no a/b slot: the float64 ring masks them (a dummy entry would cost the fifth wavefront per CU),
    // the fixed-point ring gives them one shared dummy entry (no exec masking in the row loop)
    static constexpr int NLp(bool fx) { return fx ? NL + 1 : NL; }
    static constexpr int ring_bytes(bool fx) { return RC * NLp(fx) * (fx ? 16 : 32); }
    static constexpr int lds_bytes(bool fx) { return ring_bytes(fx) + (s1_doubles + s2_doubles) * 8; }
};

// What the loads of one step return, untouched: the phases unpack at the point of use (an unpack right after the load
// would make the wave wait for it at once -- the loads are issued a step ahead).  PAIR: a lane's two slots are adjacent
// columns and come as one 8-byte / 2-byte load; otherwise one load per slot.
struct PipeIn {
    uint32_t te[2], tl[2];  // float bits
    uint32_t ge[2], gl[2], go[2];  // PAIR: [0] holds both bytes
};
template <bool PAIR>
__device__ __forceinline__ uint32_t pipe_byte(const uint32_t (&v)[2], int c)
{
    if constexpr (PAIR) return c == 0 ? (v[0] & 255u) : (v[0] >> 8);
    else return v[c];
}

// one a/b ring entry of a lane (its two slots), in the ring's own number format
template <bool FX>
struct RingEntry;
template <>
struct RingEntry<false> {
    double a[2], b[2];
};
template <>
struct RingEntry<true> {
    int32_t a[2], b[2];
};

template <int K, bool FX, typename TOut>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) k_guided_pipe(const uint8_t *__restrict__ gray, const float *__restrict__ t0,
                                                       TOut *__restrict__ tout, PipeGeom g, PipeConsts cs)
{
    using C = PipeCfg<K>;
    using Entry = RingEntry<FX>;
    constexpr int a = C::a, Lb = C::Lb, NV = C::NV, RC = C::RC, NL = C::NL, M = C::M, SW = C::SW;
    constexpr int NLp = C::NLp(FX), EB = FX ? 16 : 32;
    constexpr double K2 = (double)(K * K);
    extern __shared__ double2 lds_raw[];
    char *lds = reinterpret_cast<char *>(lds_raw);

    const int lane = threadIdx.x;
    const bool own = lane < NLp;
    const int H = g.H, W = g.W;
    const int x_lo = blockIdx.x * NV;
    int y_lo = blockIdx.y * g.band, y_hi = min(H, y_lo + g.band);
    if (g.iy1 > g.iy0) {  // border launch (wave-uniform)
        if (blockIdx.y == 0) { y_lo = 0; y_hi = g.iy0; }
        else if (blockIdx.y == 1) { y_lo = g.iy1; y_hi = H; }
        else {
            if ((int)blockIdx.x >= g.is0 && (int)blockIdx.x < g.is1) return;
            y_lo = g.iy0 + ((int)blockIdx.y - 2) * g.iband;
            y_hi = y_lo + g.iband;
        }
        if (y_lo >= y_hi) return;
    }
    const int r_lo = max(0, y_lo - a), r_hi = min(H - 1, y_hi - 1 + Lb), r_end = y_hi - 1 + a;
    const size_t img = (size_t)blockIdx.z * H * W;
    const uint32_t npx = (uint32_t)H * (uint32_t)W;
    const __amdgpu_buffer_rsrc_t rT = pipe_rsrc(t0 + img, npx * 4u), rG = pipe_rsrc(gray + img, npx),
                                 rO = pipe_rsrc(tout + img, npx * (uint32_t)sizeof(TOut));
    const uint32_t pitch_t = (uint32_t)W * 4u, pitch_g = (uint32_t)W, pitch_o = (uint32_t)W * (uint32_t)sizeof(TOut);

    int craw[2], fix_addr[2];
    bool ook[2], fix_need[2], fix_odd[2];
    uint32_t ofs_o[2], ofs_q[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int s = 2 * lane + j;
        craw[j] = pipe_reflect(x_lo - 2 * a + s, W);
        const int xo = x_lo + s;
        ook[j] = s < NV && xo < W;
        ofs_o[j] = (uint32_t)min(xo, W - 1);
        ofs_q[j] = ook[j] ? (uint32_t)xo * (uint32_t)sizeof(TOut) : kNoStore;
        const int c = x_lo - a + s, cr = pipe_reflect(c, W);
        const int sp = min(max(cr - (x_lo - a), 0), kPipeSlots - 1);
        fix_need[j] = cr != c;
        fix_addr[j] = (sp >> 1) << 2;
        fix_odd[j] = sp & 1;
    }
    // strips away from the left/right border (and an even W): a lane's two slots are adjacent, 8/2-byte aligned columns
    const bool edge = x_lo - 2 * a < 0 || x_lo - 2 * a + kPipeSlots > W || (W & 1);  // wave-uniform
    const uint32_t ofs_t[2] = {(uint32_t)craw[0] * 4u, (uint32_t)craw[1] * 4u};
    const uint32_t ofs_g[2] = {(uint32_t)craw[0], (uint32_t)craw[1]};

    // loop-invariant LDS addresses of this lane
    const uint32_t a_s1 = pipe_opaque((uint32_t)C::ring_bytes(FX) + (uint32_t)lane * 8u);
    const double *s1 = reinterpret_cast<const double *>(lds + a_s1);      // staging of V1: [P_p | p0 | P_gp | gp0 | P_i | i0]
    const double *s2 = s1 + C::s1_doubles;                                // staging of V2: [P_a | a0 | P_b | b0]
    const uint32_t a_ring = pipe_opaque((uint32_t)min(lane, NLp - 1) * (uint32_t)EB);

    // rows of step i: entering raw row i+Lb and leaving raw row i-1-a for A(i), guide of output row i-2-a for C(i-2)
    auto load_rows = [&](auto pair_tag, uint32_t oe_t, uint32_t ol_t, uint32_t oe_g, uint32_t ol_g, uint32_t oo_g, PipeIn &in) {
        if constexpr (decltype(pair_tag)::value) {
            const u32x2 te = __builtin_amdgcn_raw_buffer_load_b64(rT, ofs_t[0], oe_t, 0);
            const u32x2 tl = __builtin_amdgcn_raw_buffer_load_b64(rT, ofs_t[0], ol_t, 0);
            in.ge[0] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rG, ofs_g[0], oe_g, 0);
            in.gl[0] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rG, ofs_g[0], ol_g, 0);
            in.go[0] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rG, ofs_o[0], oo_g, 0);
            in.te[0] = te.x; in.te[1] = te.y;
            in.tl[0] = tl.x; in.tl[1] = tl.y;
            in.ge[1] = in.gl[1] = in.go[1] = 0;
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                in.te[j] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rT, ofs_t[j], oe_t, 0);
                in.tl[j] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rT, ofs_t[j], ol_t, 0);
                in.ge[j] = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rG, ofs_g[j], oe_g, 0);
                in.gl[j] = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rG, ofs_g[j], ol_g, 0);
                in.go[j] = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rG, ofs_o[j], oo_g, 0);
            }
        }
    };
    auto load_step = [&](int i, PipeIn &in) {  // any step: reflected rows
        const uint32_t re = (uint32_t)pipe_reflect(i + Lb, H), rl = (uint32_t)pipe_reflect(i - 1 - a, H),
                       ro = (uint32_t)min(max(i - 2 - a, 0), H - 1);
        load_rows(std::false_type{}, re * pitch_t, rl * pitch_t, re * pitch_g, rl * pitch_g, ro * pitch_g, in);
    };

    // ---- prologue: vertical sums of the band's first a/b row by direct summation
    double V1p[2] = {0.0, 0.0}, V1gp[2] = {0.0, 0.0};  // vertical sums of p and g*p (g = guide byte)
    uint32_t Sg[2] = {0, 0}, Sgg[2] = {0, 0};          // vertical sums of g and g*g: exact integers
    double V2a[2] = {0.0, 0.0}, V2b[2] = {0.0, 0.0};   // vertical sums of the ring's a and b (ring units)
    for (int j = 0; j < K; ++j) {
        const uint32_t row = (uint32_t)pipe_reflect(r_lo - a + j, H);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const uint32_t gq = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rG, ofs_g[c], row * pitch_g, 0);
            const double p = (double)__uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rT, ofs_t[c], row * pitch_t, 0));
            Sg[c] += gq;
            Sgg[c] += gq * gq;
            V1p[c] += p;
            V1gp[c] += (double)gq * p;
        }
    }

    int wslot = r_lo % RC;                        // ring slot of the a/b row phase B works on
    int pslot = wslot == 0 ? RC - 1 : wslot - 1;  // ring slot of the row before it

    // lanes >= NLp own no a/b slot: F64 masks them, FX32 points them at a shared dummy entry (NLp = NL + 1, no masking)
    auto ring_load = [&](int slot) {
        Entry e;
        const char *p = lds + (a_ring + (uint32_t)(slot * NLp * EB));
        if constexpr (FX) {
            const int4 v = *reinterpret_cast<const int4 *>(p);
            e.a[0] = v.x; e.b[0] = v.y; e.a[1] = v.z; e.b[1] = v.w;
        } else {
            double2 v0 = make_double2(0.0, 0.0), v1 = v0;
            if (own) {
                v0 = reinterpret_cast<const double2 *>(p)[0];
                v1 = reinterpret_cast<const double2 *>(p)[1];
            }
            e.a[0] = v0.x; e.b[0] = v0.y; e.a[1] = v1.x; e.b[1] = v1.y;
        }
        return e;
    };
    auto ring_store = [&](int slot, const Entry &e) {
        char *p = lds + (a_ring + (uint32_t)(slot * NLp * EB));
        if constexpr (FX) {
            *reinterpret_cast<int4 *>(p) = make_int4(e.a[0], e.b[0], e.a[1], e.b[1]);
        } else {
            if (own) {
                reinterpret_cast<double2 *>(p)[0] = make_double2(e.a[0], e.b[0]);
                reinterpret_cast<double2 *>(p)[1] = make_double2(e.a[1], e.b[1]);
            }
        }
    };
    auto v2_add = [&](const Entry &e) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            V2a[c] += (double)e.a[c];
            V2b[c] += (double)e.b[c];
        }
    };
    auto v2_slide = [&](const Entry &e, const Entry &l) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if constexpr (FX) {  // |fixed| < 2^30: the difference fits
                V2a[c] += (double)(e.a[c] - l.a[c]);
                V2b[c] += (double)(e.b[c] - l.b[c]);
            } else {
                V2a[c] += e.a[c] - l.a[c];
                V2b[c] += e.b[c] - l.b[c];
            }
        }
    };

    auto store_row = [&](auto pair_tag, uint32_t orow, const double *q) {  // orow: byte offset of the output row
        if constexpr (std::is_same<TOut, double>::value) {
            // (always two 8-byte stores: a 16-byte buffer store with an SGPR row offset was seen to pick up the NEXT
            // instruction's write to its data registers on gfx950 -- the compiler only guards that hazard for stores
            // without a register soffset)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, q[c]), rO, ofs_q[c], orow, 0);
        } else {
            if constexpr (decltype(pair_tag)::value) {
                const float2 v = make_float2((float)q[0], (float)q[1]);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v),
                                                      rO, ofs_q[0], orow, 0);
            } else {
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)q[c]), rO, ofs_q[c], orow, 0);
            }
        }
    };

    // window sums of one float64 plane from its staging lines: o0 = slots 2l .. 2l+K-1, o1 = slots 2l+1 .. 2l+K
    auto window = [&](const double *ps, double P, double v1, double &o0, double &o1) {
        double mid0 = ps[1], mid1 = ps[2];
#pragma unroll
        for (int d = 3; d < M; d += 2) mid0 += ps[d];
#pragma unroll
        for (int d = 4; d < M; d += 2) mid1 += ps[d];
        const double mid = mid0 + mid1, f0 = ps[SW + M];
        if constexpr (K & 1) {
            o0 = (P + mid) + f0;
            o1 = (v1 + mid) + ps[M];
        } else {
            o0 = P + mid;
            o1 = (v1 + mid) + f0;
        }
    };

    // One step.  FULL: all three phases in their steady state (no band start/end, no reflected rows): straight-line code.
    // orow_c: byte offset of output row i-2-a (FULL steps; the others compute it)
    auto step = [&](auto full_tag, auto edge_tag, int i, const PipeIn &in, uint32_t orow_c) {
        constexpr bool FULL = decltype(full_tag)::value, EDGE = decltype(edge_tag)::value, PAIR = FULL && !EDGE;
        const int rB = i - 1, rC = i - 2;
        const bool doB = FULL || (rB >= r_lo && rB <= r_end), doC = FULL || (rC >= r_lo && rC <= r_end);
        const bool doB1 = doB && (FULL || rB <= r_hi);          // a/b row rB exists in this band
        const bool doB2 = doB && (FULL || rB - a >= y_lo);      // output row rB - a belongs to this band
        const bool doC2 = doC && (FULL || rC - a >= y_lo);
        const bool doA1 = FULL || i <= r_hi;
        double *w1 = const_cast<double *>(s1), *w2 = const_cast<double *>(s2);
        const uint2 *s1i = reinterpret_cast<const uint2 *>(s1 + 4 * SW);

        // ================= read phase
        // C(i-2): window sums of the staged V2
        double oa[2] = {0.0, 0.0}, ob[2] = {0.0, 0.0};
        if (doC2) {
            window(s2, V2a[0] + V2a[1], V2a[1], oa[0], oa[1]);
            window(s2 + 2 * SW, V2b[0] + V2b[1], V2b[1], ob[0], ob[1]);
        }
        // B(i-1): window sums of the staged V1, the leaving (and, for even K, entering) ring rows
        double oP[2] = {0.0, 0.0}, oGP[2] = {0.0, 0.0};
        uint32_t oG[2] = {0, 0}, oGG[2] = {0, 0};
        Entry lv{}, ev{};
        if (doB1) {
            window(s1, V1p[0] + V1p[1], V1p[1], oP[0], oP[1]);
            window(s1 + 2 * SW, V1gp[0] + V1gp[1], V1gp[1], oGP[0], oGP[1]);
            const uint2 *ps = s1i;
            uint2 m0 = ps[1], m1 = ps[2];
#pragma unroll
            for (int d = 3; d < M; d += 2) { m0.x += ps[d].x; m0.y += ps[d].y; }
#pragma unroll
            for (int d = 4; d < M; d += 2) { m1.x += ps[d].x; m1.y += ps[d].y; }
            const uint32_t midg = m0.x + m1.x, midgg = m0.y + m1.y;
            const uint2 f0 = ps[SW + M];
            if constexpr (K & 1) {
                const uint2 pm = ps[M];
                oG[0] = Sg[0] + Sg[1] + midg + f0.x;   oGG[0] = Sgg[0] + Sgg[1] + midgg + f0.y;
                oG[1] = Sg[1] + midg + pm.x;           oGG[1] = Sgg[1] + midgg + pm.y;
            } else {
                oG[0] = Sg[0] + Sg[1] + midg;          oGG[0] = Sgg[0] + Sgg[1] + midgg;
                oG[1] = Sg[1] + midg + f0.x;           oGG[1] = Sgg[1] + midgg + f0.y;
            }
        }
        if (FULL) {
            lv = ring_load(wslot);                          // row rB - RC, about to be overwritten by row rB
            if constexpr (!(K & 1)) ev = ring_load(pslot);  // even K: entering row rB - 1
        }

        // ================= compute
        // C: q = mean_a * I + mean_b, clip (six_stadigy.py:45,180)
        if (doC2) {
            double q[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double gd = (double)pipe_byte<PAIR>(in.go, c);
                q[c] = fmin(fmax(fma(oa[c] * cs.kaI, gd, fma(ob[c], cs.kb, cs.b0)), 0.1), 1.0);
            }
#ifdef PIPE_KO_STORE
            if (q[0] + q[1] == 12345.0)
#endif
            if constexpr (FULL && !EDGE) store_row(std::true_type{}, orow_c, q);
            else store_row(std::false_type{}, FULL ? orow_c : (uint32_t)(rC - a) * pitch_o, q);
        }
        // B: a = cov / (var + eps), b = mean_p - a * mean_I (six_stadigy.py:39-40) in the ring's format
        Entry nv{};
        if (doB1) {
            double av[2], bt[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint32_t nvar = oGG[c] * (uint32_t)(K * K) - oG[c] * oG[c];  // (255 K^2)^2 var: exact (mod 2^32, < 2^32)
                const double D = fma((double)nvar, 1.0 / 255.0, cs.Ek);            // 255 K^4 (var + eps)
                const double gd = (double)oG[c];
                const double ncov = fma(K2, oGP[c], -(gd * oP[c]));                // 255 K^4 cov
                av[c] = ncov * pipe_rcp(D);
                bt[c] = fma(av[c] * (-1.0 / 255.0), gd, oP[c]);                    // K^2 b
            }
            if constexpr (FX) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    nv.a[c] = __double2loint(fma(av[c], cs.fxa, kMagic));
                    nv.b[c] = __double2loint(fma(bt[c], cs.fxb, cs.magic_b));
                }
            } else {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    nv.a[c] = av[c];
                    nv.b[c] = bt[c] * (1.0 / K2);
                }
            }
            if (EDGE) {  // a/b of virtual columns = a/b of the mirrored real column (cv2.boxFilter pads its source)
                Entry fx = nv;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if constexpr (FX) {
                        const int a0 = __builtin_amdgcn_ds_bpermute(fix_addr[c], nv.a[0]), a1 = __builtin_amdgcn_ds_bpermute(fix_addr[c], nv.a[1]);
                        const int b0 = __builtin_amdgcn_ds_bpermute(fix_addr[c], nv.b[0]), b1 = __builtin_amdgcn_ds_bpermute(fix_addr[c], nv.b[1]);
                        if (fix_need[c]) {
                            fx.a[c] = fix_odd[c] ? a1 : a0;
                            fx.b[c] = fix_odd[c] ? b1 : b0;
                        }
                    } else {
                        auto bp = [&](double v) {
                            const int lo = __builtin_amdgcn_ds_bpermute(fix_addr[c], __double2loint(v));
                            const int hi = __builtin_amdgcn_ds_bpermute(fix_addr[c], __double2hiint(v));
                            return __hiloint2double(hi, lo);
                        };
                        const double a0 = bp(nv.a[0]), a1 = bp(nv.a[1]), b0 = bp(nv.b[0]), b1 = bp(nv.b[1]);
                        if (fix_need[c]) {
                            fx.a[c] = fix_odd[c] ? a1 : a0;
                            fx.b[c] = fix_odd[c] ? b1 : b0;
                        }
                    }
                }
                nv = fx;
            }
        }
        bool ring_written = false;
        if (doB2) {
            if (FULL) {
                if constexpr (K & 1) v2_slide(nv, lv);   // odd K: the entering row is the row just computed
                else v2_slide(ev, lv);
            } else {
                // band start / end and reflected rows: the ring is addressed by reflect101(row) % RC.  The leaving row
                // shares its slot with the new row, so it is read first; then the new row goes in, then the rest.
                const int y2 = rB - a;
                const bool slide = y2 != y_lo;
                Entry l{};
                if (slide) l = ring_load(pipe_reflect(y2 - 1 - a, H) % RC);
                if (doB1) {
                    pipe_sync();
                    ring_store(wslot, nv);
                    pipe_sync();
                    ring_written = true;
                }
                if (!slide) {
                    V2a[0] = V2a[1] = V2b[0] = V2b[1] = 0.0;
                    for (int j = 0; j < K; ++j) v2_add(ring_load(pipe_reflect(y_lo - a + j, H) % RC));
                } else {
                    v2_slide(ring_load(pipe_reflect(y2 + Lb, H) % RC), l);
                }
            }
        }
        // A: V1 += raw(entering) - raw(leaving)
        if (doA1 && (FULL || i != r_lo)) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double pe = (double)__uint_as_float(in.te[c]), pl = (double)__uint_as_float(in.tl[c]);
                const uint32_t ge = pipe_byte<PAIR>(in.ge, c), gl = pipe_byte<PAIR>(in.gl, c);
                Sg[c] += ge - gl;
                Sgg[c] += ge * ge - gl * gl;
                V1p[c] += pe - pl;
                V1gp[c] += (double)ge * pe - (double)gl * pl;
            }
        }

        // ================= write phase
        pipe_sync();
        if (doA1) {
            w1[0] = V1p[0] + V1p[1];
            w1[SW] = V1p[0];
            w1[2 * SW] = V1gp[0] + V1gp[1];
            w1[3 * SW] = V1gp[0];
            uint2 *wi = reinterpret_cast<uint2 *>(w1 + 4 * SW);
            wi[0] = make_uint2(Sg[0] + Sg[1], Sgg[0] + Sgg[1]);
            wi[SW] = make_uint2(Sg[0], Sgg[0]);
        }
        if (doB2) {
            w2[0] = V2a[0] + V2a[1];
            w2[SW] = V2a[0];
            w2[2 * SW] = V2b[0] + V2b[1];
            w2[3 * SW] = V2b[0];
        }
        if (doB1) {
            if (!ring_written) ring_store(wslot, nv);
            pslot = wslot;
            wslot = wslot + 1 == RC ? 0 : wslot + 1;
        }
        pipe_sync();
    };

    using T = std::true_type;
    using F = std::false_type;
    // FULL steps i in [f_lo, f_hi]: phases C(i-2), B(i-1), A(i) all steady, and neither this step's stores nor the next
    // step's loads touch a reflected row, so every row offset advances by one pitch per step
    const int f_lo = max(r_lo + 2 + RC, a + 2), f_hi = min(r_hi, H - 2 - Lb);
    PipeIn nxt;
    load_step(r_lo, nxt);
    int i = r_lo;
    while (i <= r_end + 2) {
        if (i >= f_lo && i <= f_hi) {
            // row offsets of the loads for step i+1 and of the store of step i
            uint32_t oe_t = (uint32_t)(i + 1 + Lb) * pitch_t, ol_t = (uint32_t)(i - a) * pitch_t;
            uint32_t oe_g = (uint32_t)(i + 1 + Lb) * pitch_g, ol_g = (uint32_t)(i - a) * pitch_g, oo_g = (uint32_t)(i - 1 - a) * pitch_g;
            uint32_t orow = (uint32_t)(i - 2 - a) * pitch_o;
            // two steps per trip, the row buffers alternating roles (no register copies: a copy would wait for its load)
            auto advance = [&]() {
                oe_t += pitch_t; ol_t += pitch_t; oe_g += pitch_g; ol_g += pitch_g; oo_g += pitch_g; orow += pitch_o;
            };
            if (!edge) {
                PipeIn alt;
                // the steps outside this loop keep one byte per slot: repack on the way in, unpack on the way out
                nxt.ge[0] |= nxt.ge[1] << 8; nxt.gl[0] |= nxt.gl[1] << 8; nxt.go[0] |= nxt.go[1] << 8;
                for (; i + 1 <= f_hi; i += 2) {
#ifndef PIPE_KO_LOADS
                    load_rows(T{}, oe_t, ol_t, oe_g, ol_g, oo_g, alt);
#endif
                    step(T{}, F{}, i, nxt, orow);
                    advance();
#ifdef PIPE_SCHED_BARRIER
                    __builtin_amdgcn_sched_barrier(0);
#endif
#ifndef PIPE_KO_LOADS
                    load_rows(T{}, oe_t, ol_t, oe_g, ol_g, oo_g, nxt);
#endif
                    step(T{}, F{}, i + 1, alt, orow);
                    advance();
#ifdef PIPE_SCHED_BARRIER
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
                if (i <= f_hi) {
                    load_rows(T{}, oe_t, ol_t, oe_g, ol_g, oo_g, alt);
                    step(T{}, F{}, i, nxt, orow);
                    nxt = alt;
                    ++i;
                }
                nxt.ge[1] = nxt.ge[0] >> 8; nxt.gl[1] = nxt.gl[0] >> 8; nxt.go[1] = nxt.go[0] >> 8;
                nxt.ge[0] &= 255u; nxt.gl[0] &= 255u; nxt.go[0] &= 255u;
            } else {
                for (; i <= f_hi; ++i) {
                    const PipeIn cur = nxt;
                    load_rows(F{}, oe_t, ol_t, oe_g, ol_g, oo_g, nxt);
                    step(T{}, T{}, i, cur, orow);
                    advance();
                }
            }
        } else {
            const PipeIn cur = nxt;
            load_step(i + 1, nxt);
            step(F{}, T{}, i, cur, 0u);
            ++i;
        }
    }
}


// ---------------------------------------------------------------- split ring: a in LDS, b in registers (float64, k = 15)
// The b half of the ring: N rows x 2 slots of float64 that must live in registers: a recursive struct of scalars with
// compile-time access (an array would have to be proven constant-indexed after unrolling and was seen to stay in
// scratch memory).
template <int N>
struct RegRing {
    double x0, x1;
    RegRing<N - 1> rest;
    template <int S>
    __device__ __forceinline__ void swap_at(double &b0, double &b1)  // returns the old entry, stores the new one
    {
        if constexpr (S == 0) {
            const double o0 = x0, o1 = x1;
            x0 = b0; x1 = b1;
            b0 = o0; b1 = o1;
        } else rest.template swap_at<S - 1>(b0, b1);
    }
    __device__ __forceinline__ void clear() { x0 = x1 = 0.0; rest.clear(); }
};
template <>
struct RegRing<0> {
    template <int S> __device__ __forceinline__ void swap_at(double &, double &) {}
    __device__ __forceinline__ void clear() {}
};

// The last RC raw rows of a lane's two slots (transmission as float32 bits, the two guide bytes packed), in registers: the
// row that leaves the first box filter's window is the row that entered RC steps earlier, so it is never re-read from
// memory (those re-reads missed L2 and doubled the kernel's HBM traffic).
template <int N, bool G>  // G: the guide bytes ride along
struct RawRing {
    uint32_t t0, t1, g;
    RawRing<N - 1, G> rest;
    template <int S>
    __device__ __forceinline__ void swap_at(uint32_t &a0, uint32_t &a1, uint32_t &gg)
    {
        if constexpr (S == 0) {
            const uint32_t o0 = t0, o1 = t1;
            t0 = a0; t1 = a1;
            a0 = o0; a1 = o1;
            if constexpr (G) {
                const uint32_t og = g;
                g = gg;
                gg = og;
            }
        } else rest.template swap_at<S - 1>(a0, a1, gg);
    }
    template <int S>
    __device__ __forceinline__ void set_at(uint32_t a0, uint32_t a1, uint32_t gg)
    {
        if constexpr (S == 0) {
            t0 = a0; t1 = a1;
            if constexpr (G) g = gg;
        } else rest.template set_at<S - 1>(a0, a1, gg);
    }
};
template <bool G>
struct RawRing<0, G> {
    template <int S> __device__ __forceinline__ void swap_at(uint32_t &, uint32_t &, uint32_t &) {}
    template <int S> __device__ __forceinline__ void set_at(uint32_t, uint32_t, uint32_t) {}
};

// Round 4, SPLIT_QUAD: the staging lines hold QUAD sums instead of pair sums.  A lane's pair sum P[l] = s[2l] + s[2l+1] meets its
// right neighbour's through one DPP wavefront shift (v_mov_b32_dpp wave_shl:1, no LDS); what is staged is R[l] = P[l] + P[l+1]
// (slots 2l .. 2l+3) and one of the lane's own slots, and a window is a few quad sums two lanes apart -- k = 15: four 16-byte LDS
// reads per plane pair where the pair sums took eight, k = 20: five for ten, k = 10: three for five; the same two writes, the same
// 32 bytes per lane (split_body: stage_q / window_q).  SPLIT_QUAD = 0 keeps the pair sums (A/B builds).
#ifndef SPLIT_QUAD
#define SPLIT_QUAD 1
#endif
#ifndef SPLIT_MUL24
#define SPLIT_MUL24 1  // 0: v_mul_lo_u32 for the variance numerator (A/B builds)
#endif
__device__ __forceinline__ double dpp_next(double v)  // the value lane + 1 holds (0 in lane 63: only unstored slots read it)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t dpp_next(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);
}
__device__ __forceinline__ double2 q_add(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 q_sub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 q_next(double2 v) { return make_double2(dpp_next(v.x), dpp_next(v.y)); }
__device__ __forceinline__ uint2 q_add(uint2 a, uint2 b) { return make_uint2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ uint2 q_sub(uint2 a, uint2 b) { return make_uint2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ uint2 q_next(uint2 v) { return make_uint2(dpp_next(v.x), dpp_next(v.y)); }

// Per window width: row buffers in flight and whether the raw rows of the window ride in registers.
//   K = 15: b ring 60 + raw ring 30 registers, three buffers (the round-2 kernel);  K = 10: 40 + 20, five buffers;
//   K = 20: the b ring alone takes 80 registers, so the leaving raw row is re-read (t0 row 4 B + guide 1 B per slot), four buffers
template <int K>
struct SplitCfg {
    static constexpr bool RAWREG = K <= 16;
    static constexpr int NB = K % 3 == 0 ? 3 : K > 16 ? 2 : K % 5 == 0 ? 5 : 2;
    // AR: rows of the a ring that live in registers next to b (ring slots 0 .. AR-1; compile-time slots, as for b).  The LDS
    // of a strip decides how many wavefronts a CU holds: k = 20 with all 20 rows of a in LDS needs 23.1 KB (6 per CU), with
    // three of them in registers (12 VGPRs) and the staging lines put FIRST -- so that the last lanes' look-ahead past the
    // end of the last line lands in the ring instead of a pad -- 19.9 KB: eight per CU, two per SIMD, like k = 15.
    static constexpr int AR = K > 16 ? 3 : 0;
    static constexpr bool STAGE_FIRST = AR > 0;
    // a ring + staged V1 (3072 B) + staged V2 (2048 B) [+ the look-ahead of the last lanes past the end of the last line]
    static constexpr int pad_bytes = 16 * (PipeCfg<K>::M + 1) > 128 ? 16 * (PipeCfg<K>::M + 1) : 128;
    static constexpr int s2_bytes = STAGE_FIRST ? 2048 : 2048 + pad_bytes;
    static constexpr int ring_lds_bytes = (K - AR) * (PipeCfg<K>::NL + 1) * 16;
    static_assert(!STAGE_FIRST || ring_lds_bytes >= pad_bytes, "the look-ahead must stay inside the allocation");
    static constexpr int lds_bytes = ring_lds_bytes + PipeCfg<K>::s1_doubles * 8 + s2_bytes;
};

// Interior blocks only: every raw row/column the band touches exists (no reflection), W even, band % RC == 0.  Then a
// band is: V1 of its first a/b row by direct summation, one WARM ring period (phases A and B only: the ring and V2
// fill up from zero, which IS the direct sum of the first window because the leaving rows are the zeros the ring was
// cleared to), and band/RC periods of identical steps.  Every step of a period has its ring slot as a compile-time
// constant: b lives in registers, the LDS address of a is an immediate, and there is no slot arithmetic at all.
// t0 of one pixel from its bytes (FUSE: first half of estimate_transmission, six_stadigy.py:170-174 / enhancement_strategies.py:
// 221-225, exactly k_trans_init's operations): tab = img / (A + eps) of the 3 x 256 possible bytes (IEEE divisions, made once per
// workgroup).  This file is compiled with -ffp-contract=fast: the product and the difference are NumPy's two roundings, so they
// are spelled with the intrinsics that are never contracted.
__device__ __forceinline__ uint32_t fuse_t0_bits(const float *tab, uint32_t r, uint32_t g, uint32_t b, float omega, int pre_clip)
{
    const float dark = fminf(fminf(tab[r], tab[256 + g]), tab[512 + b]);
    float t = __fsub_rn(1.0f, __fmul_rn(omega, dark));
    if (pre_clip) t = fminf(fmaxf(t, 0.1f), 1.0f);
    return __float_as_uint(t);
}

// FUSE: the raw transmission plane is not read -- t0 is computed from the frame's bytes on the way in (`t0` then points at the
// u8 RGB frame, tab at the workgroup's LDS table); lane: the wavefront's lane (the 8-wavefront workgroups of k_guided_split8).
// (Measured and dropped, round 4: the same with the one-wavefront launch shape and the table in GLOBAL memory, six gathers
// through a buffer resource at the top of a step -- 3.70 ms against 3.46 for the LDS table and 2.83 + 0.71 unfused.)
template <int K, bool EDGE, typename TOut, bool FUSE = false>
__device__ __forceinline__ void split_body(const uint8_t *__restrict__ gray, const float *__restrict__ t0, TOut *__restrict__ tout,
                                           const SplitGeom &g, const PipeConsts &cs, char *lds, int3 bid, int lane_in = -1,
                                           const float *tab = nullptr, float omega = 0.0f, int pre_clip = 0)
{
    using C = PipeCfg<K>;
    // Any window width (round 3: even ones too).  With the output row of an a/b row r taken as r - Lb (Lb = K - 1 - a rows
    // below the anchor; = a for odd K) the a/b row that ENTERS the second box filter's window is always the row just
    // computed and the one that leaves it is K rows older, so both rings have period K for odd and even K alike:
    //     a/b row r   <- raw rows r - a .. r + Lb           (cv2.boxFilter's anchor k/2)
    //     output row y <- a/b rows y - a .. y + Lb   =>   the window that ends at a/b row r belongs to output row r - Lb
    constexpr int a = C::a, Lb = C::Lb, NV = C::NV, RC = K, NL = C::NL, M = C::M, SW = C::SW;
    // row buffers in flight (loads are issued NB - 1 steps ahead); the ring period is a multiple of it
    constexpr int NB = SplitCfg<K>::NB;
    static_assert(RC % NB == 0, "the ring period must be a multiple of the row buffers");
    // RAWREG: the last K raw transmission rows of a lane's slots ride in registers (no second read of t0); wide windows do
    // not have the registers for it next to the b ring and re-read the leaving row
    constexpr bool RAWREG = SplitCfg<K>::RAWREG;
    constexpr int NLp = NL + 1, EB = 16;
    constexpr int AR = SplitCfg<K>::AR;                              // a rows in registers (ring slots 0 .. AR-1)
    constexpr int ring_bytes = SplitCfg<K>::ring_lds_bytes;          // a rows in LDS (ring slots AR .. RC-1)
    constexpr uint32_t stage_base = SplitCfg<K>::STAGE_FIRST ? 0u : (uint32_t)ring_bytes;
    constexpr uint32_t ring_base = SplitCfg<K>::STAGE_FIRST ? 3072u + 2048u : 0u;
    constexpr double K2 = (double)(K * K);
    const int lane = lane_in >= 0 ? lane_in : (int)threadIdx.x;
    const int H = g.H, W = g.W;
    const int x_lo = bid.x * NV;
    const int y_lo = g.y0 + bid.y * g.band;
    const int r_lo = y_lo - a;
    const size_t img = (size_t)bid.z * H * W;
    const uint32_t npx = (uint32_t)H * (uint32_t)W;
    // FUSE: rT is the RGB frame (3 bytes per column) instead of the float32 t0 plane
    constexpr uint32_t kTB = FUSE ? 3u : 4u;
    const __amdgpu_buffer_rsrc_t rT = FUSE ? pipe_rsrc(reinterpret_cast<const uint8_t *>(t0) + img * 3, npx * 3u) : pipe_rsrc(t0 + img, npx * 4u),
                                 rG = pipe_rsrc(gray + img, npx), rO = pipe_rsrc(tout + img, npx * (uint32_t)sizeof(TOut));
    const uint32_t pitch_t = (uint32_t)W * kTB, pitch_g = (uint32_t)W, pitch_o = (uint32_t)W * (uint32_t)sizeof(TOut);

    // Interior strips (!EDGE): a lane's two slots are adjacent raw columns inside the image: one 8-byte / 2-byte load for
    // both.  EDGE strips (their raw columns run over the left/right image border): reflected columns, one load per slot,
    // and the a/b of virtual columns replaced by those of the mirrored real column (cv2.boxFilter pads its source).
    const int xo0 = x_lo + 2 * lane;
    int craw[2], fix_addr[2];
    bool fix_need[2], fix_odd[2];
    uint32_t ofs_t[2], ofs_g[2], ofs_o[2], ofs_q[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sl = 2 * lane + j;
        craw[j] = EDGE ? pipe_reflect(x_lo - 2 * a + sl, W) : x_lo - 2 * a + sl;
        ofs_t[j] = (uint32_t)craw[j] * kTB;
        ofs_g[j] = (uint32_t)craw[j];
        ofs_o[j] = EDGE ? (uint32_t)min(xo0 + j, W - 1) : (uint32_t)min(xo0, W - 2);
        ofs_q[j] = sl < NV && xo0 + j < W ? (uint32_t)(xo0 + j) * (uint32_t)sizeof(TOut) : kNoStore;
        const int c = x_lo - a + sl, cr = pipe_reflect(c, W);
        const int sp = min(max(cr - (x_lo - a), 0), kPipeSlots - 1);
        fix_need[j] = EDGE && cr != c;
        fix_addr[j] = (sp >> 1) << 2;
        fix_odd[j] = sp & 1;
    }

    const uint32_t a_ring = pipe_opaque(ring_base + (uint32_t)min(lane, NLp - 1) * (uint32_t)EB);

    // what the loads return, untouched (unpacked at the point of use: an early unpack would wait for the load at once)
#ifndef SPLIT_RAW_G
// 1: the guide bytes of the last RC rows ride in the register ring too; 0: the leaving row's two bytes are re-read.  Rounds 2 - 3:
// 3.13 ms against 3.03 at 4K x 64 (register pressure).  Round 4, with the quad-sum staging (209 registers without the bytes):
// 2.45 / 2.50 ms against 2.49 / 2.55 -- one load and its row arithmetic less per step.
#define SPLIT_RAW_G 1
#endif
    // (k = 15 without the fused transmission only: k = 10 with its five row buffers and the fused kernel spill with them)
    constexpr bool RAWG = SPLIT_RAW_G && RAWREG && K == 15 && !FUSE;
    struct In {
        uint32_t te[2];         // entering raw row: float bits of the two slots
        uint32_t ge[2], go[2];  // its guide bytes / the guide bytes of the output row; !EDGE: [0] holds both bytes
        uint32_t gl[2];         // (!RAWG) guide bytes of the leaving row
        uint32_t tl[2];         // (!RAWREG) leaving raw row
    };
    auto byte_of = [](const uint32_t (&v)[2], int c) { return EDGE ? v[c] : (c == 0 ? (v[0] & 255u) : (v[0] >> 8)); };
    static_assert(!FUSE || RAWREG, "the fused transmission keeps its rows in the register ring (no leaving-row reload)");
    auto load_rows = [&](uint32_t oe_t, uint32_t oe_g, uint32_t ol_t, uint32_t ol_g, uint32_t oo_g, In &in) {
        if constexpr (!RAWREG) {
            if constexpr (!EDGE) {
                const u32x2 tl = __builtin_amdgcn_raw_buffer_load_b64(rT, ofs_t[0], ol_t, 0);
                in.tl[0] = tl.x; in.tl[1] = tl.y;
            } else {
                in.tl[0] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rT, ofs_t[0], ol_t, 0);
                in.tl[1] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rT, ofs_t[1], ol_t, 0);
            }
        }
        if constexpr (!RAWG) {
            if constexpr (!EDGE) in.gl[0] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rG, ofs_g[0], ol_g, 0);
            else {
                in.gl[0] = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rG, ofs_g[0], ol_g, 0);
                in.gl[1] = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rG, ofs_g[1], ol_g, 0);
            }
        }
        if constexpr (!EDGE) {
            if constexpr (FUSE) {  // the two pixels' six bytes: a dword (2-byte aligned: the column is even) and a short
                in.te[0] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rT, ofs_t[0], oe_t, 0);
                in.te[1] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rT, ofs_t[0] + 4u, oe_t, 0);
            } else {
                const u32x2 te = __builtin_amdgcn_raw_buffer_load_b64(rT, ofs_t[0], oe_t, 0);
                in.te[0] = te.x; in.te[1] = te.y;
            }
            in.ge[0] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rG, ofs_g[0], oe_g, 0);
            in.go[0] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rG, ofs_o[0], oo_g, 0);
            in.ge[1] = in.go[1] = 0;
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (FUSE) {  // one pixel per load set: R | G << 8 | B << 16
                    const uint32_t r = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rT, ofs_t[j], oe_t, 0);
                    const uint32_t gg = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rT, ofs_t[j] + 1u, oe_t, 0);
                    const uint32_t bb = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rT, ofs_t[j] + 2u, oe_t, 0);
                    in.te[j] = r | (gg << 8) | (bb << 16);
                } else {
                    in.te[j] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rT, ofs_t[j], oe_t, 0);
                }
                in.ge[j] = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rG, ofs_g[j], oe_g, 0);
                in.go[j] = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rG, ofs_o[j], oo_g, 0);
            }
        }
    };
    // float bits of the entering row's two transmissions (FUSE: from the bytes the loads returned)
    auto te_bits = [&](const In &in, uint32_t (&tb)[2]) {
        if constexpr (!FUSE) {
            tb[0] = in.te[0]; tb[1] = in.te[1];
        } else if constexpr (!EDGE) {
            const uint32_t w0 = in.te[0], w1 = in.te[1];
            tb[0] = fuse_t0_bits(tab, w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u, omega, pre_clip);
            tb[1] = fuse_t0_bits(tab, w0 >> 24, w1 & 255u, (w1 >> 8) & 255u, omega, pre_clip);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                tb[j] = fuse_t0_bits(tab, in.te[j] & 255u, (in.te[j] >> 8) & 255u, (in.te[j] >> 16) & 255u, omega, pre_clip);
        }
    };

    // ---- prologue: vertical sums of a/b row r_lo by direct summation; ring and V2 start from zero
    double V1p[2] = {0.0, 0.0}, V1gp[2] = {0.0, 0.0};
    uint32_t Sg[2] = {0, 0}, Sgg[2] = {0, 0};
    double V2a[2] = {0.0, 0.0}, V2b[2] = {0.0, 0.0};
    RawRing<RAWREG ? RC : 0, RAWG> raw;  // slot j <-> raw row r_lo - a + j (mod RC): the slot of the row that leaves at a step is the step's slot
    auto prologue_row = [&](auto j_tag) {
        constexpr int J = decltype(j_tag)::value;
        const uint32_t row = (uint32_t)pipe_reflect(r_lo - a + J, H);  // rows above / below the image: BORDER_REFLECT_101
        uint32_t tb[2], gq[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            gq[c] = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rG, ofs_g[c], row * pitch_g, 0);
            if constexpr (FUSE) {
                const uint32_t r = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rT, ofs_t[c], row * pitch_t, 0);
                const uint32_t gg = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rT, ofs_t[c] + 1u, row * pitch_t, 0);
                const uint32_t bb = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rT, ofs_t[c] + 2u, row * pitch_t, 0);
                tb[c] = fuse_t0_bits(tab, r, gg, bb, omega, pre_clip);
            } else {
                tb[c] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rT, ofs_t[c], row * pitch_t, 0);
            }
            const double p = (double)__uint_as_float(tb[c]);
            Sg[c] += gq[c];
            Sgg[c] += gq[c] * gq[c];
            V1p[c] += p;
            V1gp[c] += (double)gq[c] * p;
        }
        if constexpr (RAWREG) raw.template set_at<J>(tb[0], tb[1], gq[0] | (gq[1] << 8));
    };
    [&]<int... J>(std::integer_sequence<int, J...>) { (prologue_row(IC<J>{}), ...); }(std::make_integer_sequence<int, RC>{});
    RegRing<RC> rb;
    rb.clear();
#pragma unroll
    for (int s = 0; s < RC - AR; ++s) *reinterpret_cast<double2 *>(lds + a_ring + (uint32_t)(s * NLp * EB)) = make_double2(0.0, 0.0);
    RegRing<AR> ra;  // the first AR rows of the a ring
    ra.clear();

    // Staging, interleaved: the pair sums of TWO planes share a 16-byte entry, so a lane's look-ahead (entries l+1 .. l+M)
    // comes in as M ds_read_b128 that serve both planes (256 B/clk in the LDS, where the ds_read2_b64 pairs of separate
    // 8-byte lines get 128 B/clk: the kernel is LDS-time bound as much as issue bound).  Lines of 64 entries; a read past
    // the end of a line lands in the next one and only feeds lanes whose sums are never stored.
    //   PP1[i] = {P_p, P_gp}   VV1[i] = {p0, gp0}    PI[i] = {P_g, P_gg}  VI[i] = {g0, gg0} (uint2)   -- staged V1
    //   PP2[i] = {P_a, P_b}    VV2[i] = {a0, b0}                                                       -- staged V2
    const uint32_t a16 = pipe_opaque(stage_base + (uint32_t)lane * 16u);
    const uint32_t a8 = pipe_opaque(stage_base + 2048u + (uint32_t)lane * 8u);
    const double2 *pp1 = reinterpret_cast<const double2 *>(lds + a16), *vv1 = pp1 + SW;
    const uint2 *pi = reinterpret_cast<const uint2 *>(lds + a8), *vi = pi + SW;
    const double2 *pp2 = reinterpret_cast<const double2 *>(lds + a16 + 3072u), *vv2 = pp2 + SW;
    static_assert(C::s1_doubles * 8 == 3072 && (SplitCfg<K>::STAGE_FIRST || SplitCfg<K>::s2_bytes >= 2048 + 16 * (C::M + 1)), "staging layout");
    // window sums of a plane pair: o[0] = slots 2l .. 2l+K-1, o[1] = slots 2l+1 .. 2l+K
    //   odd K  (M = (K-1)/2): o[0] = P[l] + P[l+1..l+M-1] + v0[l+M],   o[1] = v1[l] + P[l+1..l+M-1] + P[l+M]
    //   even K (M = K/2):     o[0] = P[l] + P[l+1..l+M-1],             o[1] = v1[l] + P[l+1..l+M-1] + v0[l+M]
    auto window2 = [&](const double2 *pp, const double2 *vv, double2 P, double2 v1, double2 (&o)[2]) {
        double2 m0 = pp[1], m1 = pp[2];
#pragma unroll
        for (int d = 3; d < M; d += 2) { const double2 t = pp[d]; m0.x += t.x; m0.y += t.y; }
#pragma unroll
        for (int d = 4; d < M; d += 2) { const double2 t = pp[d]; m1.x += t.x; m1.y += t.y; }
        const double2 f0 = vv[M];
        const double midx = m0.x + m1.x, midy = m0.y + m1.y;
        if constexpr (K & 1) {
            const double2 pm = pp[M];
            o[0] = make_double2((P.x + midx) + f0.x, (P.y + midy) + f0.y);
            o[1] = make_double2((v1.x + midx) + pm.x, (v1.y + midy) + pm.y);
        } else {
            o[0] = make_double2(P.x + midx, P.y + midy);
            o[1] = make_double2((v1.x + midx) + f0.x, (v1.y + midy) + f0.y);
        }
    };
    // QUAD: lines of quad sums (see SPLIT_QUAD above).  s0 / s1: a lane's even / odd slot, P = s0 + s1, R[l] = P[l] + P[l+1]; the second
    // line holds the slot that the window one column further on gains (odd K: the odd slot, even K: the even one) as it is.
    //   k = 15 (M = 7):  o1 = (s1 + P[l+1]) + R[l+2] + R[l+4] + R[l+6],   o0 = o1 - s1[l+7] + s0       keep = s1 + P[l+1]
    //   k = 20 (M = 10): o0 = R[l] + R[l+2] + R[l+4] + R[l+6] + R[l+8],   o1 = o0 - s0 + s0[l+10]      keep = R[l]
    //   k = 10 (M = 5):  o0 = P[l] + R[l+1] + R[l+3],                     o1 = s1 + R[l+1] + R[l+3] + s0[l+5]
    // (the sums of p and g p are exact in float64, so the order does not matter there; for a and b one subtraction rounds like one
    // addition: both sums have the window's magnitude).  keep rides in registers from the staging to the next step's read phase:
    // the vertical sums do not change in between.
    // (k = 10 keeps its pair sums: three reads for five measured 0.61 / 0.62 ms against 0.60 / 0.60 at 4K x 16; k = 20: 0.90 -> 0.83)
    constexpr bool QUAD = SPLIT_QUAD != 0 && ((K & 1) || !(M & 1));
    static_assert(!QUAD || !(K & 1) || (M & 1), "odd windows: an odd number of pair sums on either side of the lane's own");
    double2 keep1 = make_double2(0.0, 0.0), keep2 = keep1;
    uint2 keepI = make_uint2(0u, 0u);
    auto stage_q = [&](auto *line, auto s0, auto s1, auto &keep) {
        const auto P = q_add(s0, s1);
        const auto Pn = q_next(P);
        const auto R = q_add(P, Pn);
        line[0] = R;
        if constexpr (K & 1) {
            line[SW] = s1;
            keep = q_add(s1, Pn);
        } else {
            line[SW] = s0;
            keep = R;
        }
    };
    auto window_q = [&](const auto *rr, auto keep, auto s0, auto s1, auto *o) {
        using T = decltype(keep);
        const T w = rr[SW + M];
        if constexpr (K & 1) {
            T mid = rr[2];
#pragma unroll
            for (int d = 4; d <= M - 3; d += 2) mid = q_add(mid, rr[d]);
            o[1] = q_add(q_add(keep, mid), rr[M - 1]);
            o[0] = q_add(q_sub(o[1], w), s0);
        } else if constexpr (!(M & 1)) {
            T mid = rr[2];
#pragma unroll
            for (int d = 4; d <= M - 2; d += 2) mid = q_add(mid, rr[d]);
            o[0] = q_add(keep, mid);
            o[1] = q_add(q_sub(o[0], s0), w);
        } else {
            T mid = rr[1];
#pragma unroll
            for (int d = 3; d <= M - 2; d += 2) mid = q_add(mid, rr[d]);
            o[0] = q_add(q_add(s0, s1), mid);
            o[1] = q_add(q_add(s1, mid), w);
        }
    };
    auto stage_v1 = [&]() {
        if constexpr (QUAD) {
            stage_q(const_cast<double2 *>(pp1), make_double2(V1p[0], V1gp[0]), make_double2(V1p[1], V1gp[1]), keep1);
            stage_q(const_cast<uint2 *>(pi), make_uint2(Sg[0], Sgg[0]), make_uint2(Sg[1], Sgg[1]), keepI);
        } else {
            const_cast<double2 *>(pp1)[0] = make_double2(V1p[0] + V1p[1], V1gp[0] + V1gp[1]);
            const_cast<double2 *>(vv1)[0] = make_double2(V1p[0], V1gp[0]);
            const_cast<uint2 *>(pi)[0] = make_uint2(Sg[0] + Sg[1], Sgg[0] + Sgg[1]);
            const_cast<uint2 *>(vi)[0] = make_uint2(Sg[0], Sgg[0]);
        }
    };

    // One step i: C(i-2) [not in the warm period], B(i-1) on ring slot S, A(i).
    auto step = [&](auto warm_tag, auto slot_tag, const In &in, uint32_t orow, bool row_ok) {
        constexpr bool WARM = decltype(warm_tag)::value;
        constexpr int S = decltype(slot_tag)::value;
        char *ring_p = lds + a_ring + (uint32_t)((S >= AR ? S - AR : 0) * NLp * EB);
        // ================= read phase
        double2 oab[2] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0)};  // {sum a, sum b} of the two slots
        if constexpr (!WARM) {
            if constexpr (QUAD) window_q(pp2, keep2, make_double2(V2a[0], V2b[0]), make_double2(V2a[1], V2b[1]), oab);
            else window2(pp2, vv2, make_double2(V2a[0] + V2a[1], V2b[0] + V2b[1]), make_double2(V2a[1], V2b[1]), oab);
        }
        double2 opg[2];  // {sum p, sum g*p}
        if constexpr (QUAD) window_q(pp1, keep1, make_double2(V1p[0], V1gp[0]), make_double2(V1p[1], V1gp[1]), opg);
        else window2(pp1, vv1, make_double2(V1p[0] + V1p[1], V1gp[0] + V1gp[1]), make_double2(V1p[1], V1gp[1]), opg);
        uint32_t oG[2], oGG[2];
        if constexpr (QUAD) {
            uint2 oi[2];
            window_q(pi, keepI, make_uint2(Sg[0], Sgg[0]), make_uint2(Sg[1], Sgg[1]), oi);
            oG[0] = oi[0].x; oGG[0] = oi[0].y;
            oG[1] = oi[1].x; oGG[1] = oi[1].y;
        } else {
            const uint2 *ps = pi;
            uint2 m0 = ps[1], m1 = ps[2];
#pragma unroll
            for (int d = 3; d < M; d += 2) { m0.x += ps[d].x; m0.y += ps[d].y; }
#pragma unroll
            for (int d = 4; d < M; d += 2) { m1.x += ps[d].x; m1.y += ps[d].y; }
            const uint32_t midg = m0.x + m1.x, midgg = m0.y + m1.y;
            const uint2 f0 = vi[M];
            if constexpr (K & 1) {
                const uint2 pm = ps[M];
                oG[0] = Sg[0] + Sg[1] + midg + f0.x;   oGG[0] = Sgg[0] + Sgg[1] + midgg + f0.y;
                oG[1] = Sg[1] + midg + pm.x;           oGG[1] = Sgg[1] + midgg + pm.y;
            } else {
                oG[0] = Sg[0] + Sg[1] + midg;          oGG[0] = Sgg[0] + Sgg[1] + midgg;
                oG[1] = Sg[1] + midg + f0.x;           oGG[1] = Sgg[1] + midgg + f0.y;
            }
        }
        double2 la = make_double2(0.0, 0.0);  // a of the leaving row (same slot)
        if constexpr (S >= AR) la = *reinterpret_cast<const double2 *>(ring_p);

        // ================= compute
        if constexpr (!WARM) {  // C: q = mean_a * I + mean_b, clip (six_stadigy.py:45,180)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double gd = (double)byte_of(in.go, c);
                const double q = fmin(fmax(fma(oab[c].x * cs.kaI, gd, fma(oab[c].y, cs.kb, cs.b0)), 0.1), 1.0);
                const uint32_t oq = row_ok ? ofs_q[c] : kNoStore;  // (a band's last period may run past the image)
                if constexpr (std::is_same<TOut, double>::value)
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, q), rO, oq, orow, 0);
                else
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)q), rO, oq, orow, 0);
            }
        }
        // B: a = cov / (var + eps), b = mean_p - a * mean_I (six_stadigy.py:39-40)
        double av[2], bv[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            // (255 K^2)^2 var, exact.  k <= 16: both factors stay below 2^24 (sum g <= 255 K^2, sum g^2 <= 255^2 K^2) and the products
            // below 2^32, so the full-rate 24-bit multiplier serves (v_mul_lo_u32 is a quarter-rate instruction)
            constexpr bool MUL24 = SPLIT_MUL24 && 255 * 255 * K * K < (1 << 24);  // (k = 20: sum g^2 needs 25 bits; the difference is exact mod 2^32)
            const uint32_t nvar = MUL24 ? __umul24(oGG[c], (uint32_t)(K * K)) - __umul24(oG[c], oG[c])
                                        : oGG[c] * (uint32_t)(K * K) - oG[c] * oG[c];
            const double D = fma((double)nvar, 1.0 / 255.0, cs.Ek);            // 255 K^4 (var + eps)
            const double gd = (double)oG[c];
            const double ncov = fma(K2, opg[c].y, -(gd * opg[c].x));           // 255 K^4 cov
            av[c] = ncov * pipe_rcp(D);
            bv[c] = fma(av[c] * (-1.0 / 255.0), gd, opg[c].x) * (1.0 / K2);
        }
        if constexpr (EDGE) {
            double fa[2] = {av[0], av[1]}, fb[2] = {bv[0], bv[1]};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                auto bp = [&](double v) {
                    const int lo = __builtin_amdgcn_ds_bpermute(fix_addr[c], __double2loint(v));
                    const int hi = __builtin_amdgcn_ds_bpermute(fix_addr[c], __double2hiint(v));
                    return __hiloint2double(hi, lo);
                };
                const double a0 = bp(av[0]), a1 = bp(av[1]), b0 = bp(bv[0]), b1 = bp(bv[1]);
                if (fix_need[c]) {
                    fa[c] = fix_odd[c] ? a1 : a0;
                    fb[c] = fix_odd[c] ? b1 : b0;
                }
            }
            av[0] = fa[0]; av[1] = fa[1];
            bv[0] = fb[0]; bv[1] = fb[1];
        }
        double lb0 = bv[0], lb1 = bv[1];
        rb.template swap_at<S>(lb0, lb1);  // registers: new b in, b of the leaving row out
        if constexpr (S < AR) {
            la = make_double2(av[0], av[1]);
            ra.template swap_at<S>(la.x, la.y);
        }
        V2a[0] += av[0] - la.x; V2a[1] += av[1] - la.y;
        V2b[0] += bv[0] - lb0;  V2b[1] += bv[1] - lb1;
        // A: V1 += raw(entering) - raw(leaving); the entering row takes the leaving row's place in the register ring
        uint32_t tbe[2];
        te_bits(in, tbe);
        uint32_t lt[2] = {tbe[0], tbe[1]}, lg = EDGE ? (in.ge[0] | (in.ge[1] << 8)) : in.ge[0];
        if constexpr (RAWREG) raw.template swap_at<S>(lt[0], lt[1], lg);
        else { lt[0] = in.tl[0]; lt[1] = in.tl[1]; }
        if constexpr (!RAWG) lg = EDGE ? (in.gl[0] | (in.gl[1] << 8)) : in.gl[0];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const double pe = (double)__uint_as_float(tbe[c]), pl = (double)__uint_as_float(lt[c]);
            const uint32_t ge = byte_of(in.ge, c), gl = c == 0 ? (lg & 255u) : (lg >> 8);
            Sg[c] += ge - gl;
            Sgg[c] += ge * ge - gl * gl;
            V1p[c] += pe - pl;
            V1gp[c] += (double)ge * pe - (double)gl * pl;
        }

        // ================= write phase
        pipe_sync();
        stage_v1();
        if constexpr (QUAD) {
            stage_q(const_cast<double2 *>(pp2), make_double2(V2a[0], V2b[0]), make_double2(V2a[1], V2b[1]), keep2);
        } else {
            const_cast<double2 *>(pp2)[0] = make_double2(V2a[0] + V2a[1], V2b[0] + V2b[1]);
            const_cast<double2 *>(vv2)[0] = make_double2(V2a[0], V2b[0]);
        }
        if constexpr (S >= AR) *reinterpret_cast<double2 *>(ring_p) = make_double2(av[0], av[1]);
        pipe_sync();
    };

    // fill: stage V1(r_lo); loads for steps r_lo+1 and r_lo+2 (step i: rows i+Lb entering, i-1-a leaving, i-2-a guide)
    pipe_sync();
    stage_v1();
    pipe_sync();
    // Row counters (wavefront-uniform, scalar): every row index is reflected into the image on use, which is all the top
    // and bottom borders need -- a/b "rows" above row 0 computed from reflected raw rows ARE the a/b rows cv2.boxFilter's
    // BORDER_REFLECT_101 mirrors in (the window of virtual row -j reflects onto the window of row j), and likewise below.
    const int i0 = r_lo + 1;
    int re = i0 + Lb;       // entering raw row of the next load (the leaving row is re - RC)
    int gr = i0 - Lb - 2;   // guide row that rides with the next load: the load issued at step i (NB - 1 steps ahead) serves
                            // step i + NB - 1, whose C phase stores row i + NB - 3 - Lb (unused before the first normal step)
    int yo = y_lo;          // row the next normal step stores
    auto issue = [&](In &fill) {
        // (the row counters pass through an empty asm: without it instruction selection may emit the row arithmetic of a whole
        // ring period ahead of its first step -- 45 live scalars, spilled to vector lanes, then to scratch)
        asm volatile("" : "+s"(re), "+s"(gr), "+s"(yo));
        const uint32_t er = (uint32_t)pipe_reflect(re, H), lr = (uint32_t)pipe_reflect(re - RC, H);
        const uint32_t og = (uint32_t)min(max(gr, 0), H - 1);
        load_rows(er * pitch_t, er * pitch_g, lr * pitch_t, lr * pitch_g, og * pitch_g, fill);
        ++re;
        ++gr;
    };
    In buf[NB];  // (indexed by compile-time constants only)
    [&]<int... Q>(std::integer_sequence<int, Q...>) { (issue(buf[Q]), ...); }(std::make_integer_sequence<int, NB - 1>{});
    auto one = [&](auto warm_tag, auto slot_tag, In &fill, const In &use) {
        issue(fill);
        step(warm_tag, slot_tag, use, (uint32_t)yo * pitch_o, yo < H);
        if constexpr (!decltype(warm_tag)::value) ++yo;
        __builtin_amdgcn_sched_barrier(0);  // one step's LDS reads are not stretched over its neighbours (VGPR budget)
    };
    auto period = [&](auto warm_tag) {
        // step with ring slot S uses buffer S % NB and refills the buffer the previous step used
        auto at = [&](auto slot_tag) {
            constexpr int S = decltype(slot_tag)::value;
            one(warm_tag, slot_tag, buf[(S + NB - 1) % NB], buf[S % NB]);
        };
        [&]<int... Q>(std::integer_sequence<int, Q...>) { (at(IC<Q>{}), ...); }(std::make_integer_sequence<int, RC>{});
    };
    // warm period: steps r_lo+1 .. r_lo+RC (slots 0 .. RC-1)
    period(std::true_type{});
    // normal periods: the band's rows (the last band stops at the image's last row)
    const int nper = (((int)bid.y == g.nbands - 1 ? g.y_end - y_lo : g.band) + RC - 1) / RC;
    for (int p = 0; p < nper; ++p) period(std::false_type{});
}


template <int K, typename TOut>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_guided_split(const uint8_t *__restrict__ gray, const float *__restrict__ t0, TOut *__restrict__ tout, SplitGeom g, PipeConsts cs)
{
    extern __shared__ double2 lds_raw[];
    char *lds = reinterpret_cast<char *>(lds_raw);
    // Neighbouring strips share the cache lines their halos overlap in (a strip advances 100 columns, its rows start at
    // arbitrary bytes).  Workgroups are dealt to the 8 XCDs round-robin in launch order, so launch order is folded here:
    // the workgroups one XCD receives form one contiguous run of (strip, band, image) triples, strips fastest, and
    // neighbours run at the same time behind the same L2.
    int3 bid = {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    if (g.xcd_fold) {
        const uint32_t gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
        const uint32_t lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const uint32_t xcd = lin & 7u, idx = lin >> 3, q = n >> 3, r = n & 7u;
        const uint32_t log = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        bid.x = (int)(log % gx);
        bid.y = (int)((log / gx) % gy);
        bid.z = (int)(log / (gx * gy));
    }
    const int x_lo = bid.x * PipeCfg<K>::NV;
    const bool edge = x_lo - 2 * PipeCfg<K>::a < 0 || x_lo - 2 * PipeCfg<K>::a + kPipeSlots > g.W;  // wave-uniform
    if (edge) split_body<K, true, TOut>(gray, t0, tout, g, cs, lds, bid);
    else split_body<K, false, TOut>(gray, t0, tout, g, cs, lds, bid);
}

// ---------------------------------------------------------------- measured and dropped (round 4): eight ADJACENT strips per workgroup
// VERDICT r03 asked for the strips' 28-slot halo to be shared: eight wavefronts of one workgroup on adjacent 128-slot strips of a
// 1024-slot group (996 columns out instead of 8 x 100), the staging lines of the eight end to end in LDS so that a lane's
// look-ahead runs on into its right neighbour's entries, two LDS counters per wavefront (stagings done / read phases done,
// polled with s_sleep, bounded) ordering "w reads what w + 1 staged" -- LDS filled to the byte (40 KB of lines + 8 x 15 KB of
// rings = 160 KB, the counters in ring entries of lanes that own no slot).  Built, bit-exact against the tolerance tests on the
// first run -- and SLOWER at 4K x 64 (A/B in one run): 3.32 ms against 2.82 for the lone strips.  With the waits and signals
// compiled out (wrong results, timing only) it took 2.73 ms: the 20 % fewer slots buy 3 %, because a workgroup of eight
// wavefronts that start, warm up and end together keeps a CU less busy than eight staggered lone wavefronts (k_guided_split8
// below pays the same ~10 %), and two counter round trips per step (+0.6 ms) cost more than the halo.  The kernel is bound
// by the latency chain of ONE wavefront's step at two wavefronts per SIMD; anything that adds to that chain loses.
//
// ---------------------------------------------------------------- the same strips, eight to a workgroup, t0 computed on the way in
// Round 4: the raw transmission plane (k_trans_init: 3 B/px read, 4 written, then 4 read here; 0.72 ms at 4K x 64) is not
// materialised.  t0 of a pixel is a function of its three bytes -- min_c(img_c / (A_c + eps)) through a 3 x 256 table of IEEE
// quotients -- so a strip's wavefront reads the frame's bytes (6 per lane and row instead of 8) and looks the quotients up.
// The table is 3 KB: one per wavefront would cost the eighth wavefront of a CU (8 x 19.2 KB of ring and staging lines leave
// 10 KB), so eight strips share a workgroup and ONE table.  Nothing else is shared: every wavefront keeps its own LDS
// region and its own pace (one barrier, after the table is filled).  grid (ceil(strips * bands / 8), 1, B), block 512.
struct FuseT0Args {
    const uint8_t *rgb;    // [B][H][W][3]
    const int32_t *kind;   // [B] cast kinds or nullptr
    const float *A;        // [B][3]
    float omega, norm_eps;
    int pre_clip;
};

template <int K, typename TOut>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_guided_split8(const uint8_t *__restrict__ gray, FuseT0Args fz, TOut *__restrict__ tout, SplitGeom g, PipeConsts cs, int nstrips)
{
    extern __shared__ double2 lds_raw[];
    char *lds = reinterpret_cast<char *>(lds_raw);
    constexpr int per_wave = SplitCfg<K>::lds_bytes;
    float *tab = reinterpret_cast<float *>(lds + 8 * per_wave);
    // (readfirstlane: the compiler has to KNOW that the wavefront index is uniform -- strip, band and every row offset derive
    // from it, and a row offset it takes for divergent turns each buffer access into a waterfall loop: 803 loops in the first build)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // Launch order folded per XCD as in k_guided_split: the workgroups one XCD receives (every eighth in launch order) are
    // neighbours -- consecutive groups of strips, band after band, image after image -- behind the same L2.
    int grp = (int)blockIdx.x, b = (int)blockIdx.z;
    if (g.xcd_fold) {
        const uint32_t gx = gridDim.x, n = gx * gridDim.z, lin = blockIdx.x + gx * blockIdx.z;
        const uint32_t xcd = lin & 7u, idx = lin >> 3, q = n >> 3, r = n & 7u;
        const uint32_t log = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        grp = (int)(log % gx);
        b = (int)(log / gx);
    }
    {
        const int kd = fz.kind ? fz.kind[b] : 0;
        for (int i = threadIdx.x; i < 768; i += 512) {
            const int c = i >> 8;
            tab[i] = px_val(i & 255, px_atten(kd, c)) / (fz.A[b * 3 + c] + fz.norm_eps);  // S6:170 / ES:221, as k_trans_init
        }
    }
    __syncthreads();
    const int item = grp * 8 + wave;
    if (item >= nstrips * g.nbands) return;  // (wavefront-uniform; no barrier follows)
    const int3 bid = {item % nstrips, item / nstrips, b};
    const int x_lo = bid.x * PipeCfg<K>::NV;
    const bool edge = x_lo - 2 * PipeCfg<K>::a < 0 || x_lo - 2 * PipeCfg<K>::a + kPipeSlots > g.W;  // wave-uniform
    const float *t0 = reinterpret_cast<const float *>(fz.rgb);
    if (edge) split_body<K, true, TOut, true>(gray, t0, tout, g, cs, lds + wave * per_wave, bid, lane, tab, fz.omega, fz.pre_clip);
    else split_body<K, false, TOut, true>(gray, t0, tout, g, cs, lds + wave * per_wave, bid, lane, tab, fz.omega, fz.pre_clip);
}

template <int K, typename TOut>
int launch_split8(const uint8_t *d_gray, const FuseT0Args &fz, Shape s, const PipeConsts &cs, TOut *d_t, int y0, int band, int nbands,
                  int y_end, hipStream_t st)
{
    using C = PipeCfg<K>;
    constexpr int lds = 8 * SplitCfg<K>::lds_bytes + 768 * 4;
    static_assert(lds <= 160 * 1024, "eight strips and the table have to fit one CU");
#ifndef UWIE_SPLIT8_FOLD
#define UWIE_SPLIT8_FOLD 1
#endif
    SplitGeom g{s.H, s.W, y0, band, y_end, UWIE_SPLIT8_FOLD, nbands};
    const int nstrips = cdiv(s.W, C::NV);
    uwie_ctx *ctx = current_ctx();
    const int bit = std::is_same<TOut, double>::value ? 1 : 2;
    if (ctx && !(ctx->attr_gf_split8 & bit)) {  // more than 64 KB of dynamic LDS: once per context and instantiation
        UWIE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_guided_split8<K, TOut>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        ctx->attr_gf_split8 |= bit;
    }
    {
        UWIE_PROF("k_guided_split8", st);
        hipLaunchKernelGGL((k_guided_split8<K, TOut>), dim3(cdiv((long long)nstrips * nbands, 8), 1, s.B), dim3(512), (size_t)lds, st, d_gray,
                           fz, d_t, g, cs, nstrips);
    }
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

template <int K, typename TOut>
int launch_split(const uint8_t *d_gray, const float *d_t0, Shape s, const PipeConsts &cs, TOut *d_t, int y0, int band, int nbands,
                 int y_end, hipStream_t st)
{
    using C = PipeCfg<K>;
    constexpr int lds = SplitCfg<K>::lds_bytes;
#ifndef UWIE_SPLIT_FOLD
#define UWIE_SPLIT_FOLD 1
#endif
    SplitGeom g{s.H, s.W, y0, band, y_end, UWIE_SPLIT_FOLD, nbands};  // launch order folded per XCD (measured: reads 6.50 -> 3.6 GB per launch)
    const int nstrips = cdiv(s.W, C::NV);
    {
        UWIE_PROF("k_guided_split", st);
        hipLaunchKernelGGL((k_guided_split<K, TOut>), dim3(nstrips, nbands, s.B), dim3(64), (size_t)lds, st, d_gray, d_t0, d_t, g, cs);
    }
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

// border != nullptr: {iy0, iy1, iband, nibands, is0, is1}: the launch covers everything around the interior block
template <int K, bool FX, typename TOut>
int launch_pipe(const uint8_t *d_gray, const float *d_t0, Shape s, const PipeConsts &cs, TOut *d_t, hipStream_t st,
                const int *border = nullptr)
{
    using C = PipeCfg<K>;
    constexpr int lds = C::lds_bytes(FX);
    const int nstrips = cdiv(s.W, C::NV);
    const int resident = 256 * std::max(1, std::min(8, (160 * 1024) / lds));
    int nbands = 1;
    if (tune().gf_bands > 0) nbands = tune().gf_bands;
    else {
        const long strips = (long)nstrips * s.B;
        if (strips < 12L * resident) nbands = (int)cdiv((size_t)(12L * resident), (size_t)strips);
        const int cap = std::max(1, s.H / (16 * (K - 1)));
        nbands = std::min(nbands, std::max(cap, (int)cdiv((size_t)(3L * resident), (size_t)strips)));
    }
    nbands = std::max(1, std::min(nbands, s.H / std::max(UWIE_GF_MINBAND, 2 * K)));
    PipeGeom g{};
    g.H = s.H; g.W = s.W;
    g.band = cdiv(s.H, nbands);
    int gy = cdiv(s.H, g.band);
    if (border) {
        g.iy0 = border[0]; g.iy1 = border[1]; g.iband = border[2];
        gy = 2 + border[3];
        g.is0 = border[4]; g.is1 = border[5];
    }
    {
        UWIE_PROF("k_guided_pipe", st);
        hipLaunchKernelGGL((k_guided_pipe<K, FX, TOut>), dim3(nstrips, gy, s.B), dim3(64), (size_t)lds, st, d_gray, d_t0, d_t, g, cs);
    }
    UWIE_LAUNCH_CHECK();
    return UWIE_OK;
}

}  // namespace

// Rows [*iy0, min(H, *iy0 + *nb * *band)) go to k_guided_split (float64 ring; k = 10, 15 or 20, an even W, a job large enough
// to fill the chip with long bands); false = the general kernel alone.
//   odd k:  every row -- the kernel reflects row indices itself, which is exact for a symmetric window;
//   even k: the window [y - k/2, y + k/2 - 1] is not symmetric, so a virtual a/b row above row 0 built from reflected raw rows
//           is NOT the a/b row cv2.boxFilter mirrors in: the split kernel takes the whole bands between row k and row
//           H - (k - 2) (no reflection anywhere in them), the general kernel the rows above and below (launch_pipe, border).
bool guided_split_plan(Shape s, int k, int *iy0, int *band, int *nb, int *rows)
{
    if (rows) *rows = 0;
    if ((k != 10 && k != 15 && k != 20) || (s.W & 1) || !tune().gf_split) return false;
    if (s.W < 2 * k || s.H < 4 * k || s.B > 65535 || s.npx() >= ((size_t)1 << 27)) return false;
    const int RC = k, a = k / 2, Lb = k - 1 - a, NV = kPipeSlots - 2 * (k - 1);
    const bool odd = k & 1;
    *iy0 = odd ? 0 : 2 * a;
    const int periods = odd ? cdiv(s.H, RC) : (s.H - 2 * a - 2 * Lb) / RC;
    if (periods < 4) return false;
    const long strips = (long)cdiv(s.W, NV) * s.B;
    const bool forced = tune().gf_bands > 0;  // (tests: the split kernel on small frames)
    int n;
    if (forced) n = tune().gf_bands;
    else {
        // ~8 wavefronts per resident slot (256 CUs x 8) even out the tail; a band costs one extra ring period
        n = (int)cdiv((size_t)(8L * 2048), (size_t)strips);
        n = std::min(n, std::max(1, periods * RC / 180));  // ... but at least ~180 rows long
    }
    n = std::max(1, std::min(n, periods));
    if (odd) {
        const int per_band = cdiv(periods, n);
        *band = RC * per_band;
        *nb = cdiv(periods, per_band);  // the last band may be shorter (it stops at the image's last row)
        if (rows) *rows = s.H;
    } else {
        const int per_band = periods / n;  // whole periods only: the last band takes the periods that are left over
        *band = RC * per_band;
        *nb = periods / per_band;
        if (rows) *rows = periods * RC;
    }
    // small jobs (fewer long bands than half the chip holds): the general kernel cuts shorter bands
    return forced || strips * n >= 1024;
}

// The guided filter with the transmission's first half fused in (k_guided_split8): window 15, a frame the split kernel takes
// whole, tuning gf_pipe / gf_split / gf_fuse on.  false = the caller materialises t0 (launch_trans_init) as before.
bool guided_fused_takes(Shape s, int k)
{
    if (k != 15 || !tune().gf_pipe || !tune().gf_fuse) return false;
    int iy0, band, nb, rows;
    return guided_split_plan(s, k, &iy0, &band, &nb, &rows) && rows == s.H;
}

int launch_guided_fused(const uint8_t *d_gray, const uint8_t *d_rgb, const int32_t *d_kind, const float *d_A, float omega,
                        float norm_eps, int pre_clip, Shape s, int k, double eps, double *d_t, hipStream_t st, bool out_f32)
{
    UWIE_REQUIRE(guided_fused_takes(s, k) && eps > 0.0, "guided_fused: not a job for the fused kernel");
    const double K2 = (double)k * k, scale = 1.0 / K2;
    PipeConsts cs{};
    cs.Ek = 255.0 * K2 * K2 * eps;
    cs.kaI = scale / 255.0;
    cs.kb = scale;
    cs.b0 = 0.0;
    int iy0, band, nb, rows;
    guided_split_plan(s, k, &iy0, &band, &nb, &rows);
    const FuseT0Args fz{d_rgb, d_kind, d_A, omega, norm_eps, pre_clip};
    if (out_f32) return launch_split8<15, float>(d_gray, fz, s, cs, reinterpret_cast<float *>(d_t), iy0, band, nb, s.H, st);
    return launch_split8<15, double>(d_gray, fz, s, cs, d_t, iy0, band, nb, s.H, st);
}

// ring: 0 = float64 (split ring where guided_split_plan takes the job), 1 = fixed-point int32 (requires 0.1 <= t0 <= 1:
// the caller's pre-clip, six_stadigy.py:174)
int launch_guided_pipe(const uint8_t *d_gray, const float *d_t0, Shape s, int k, double eps, int ring, double *d_t,
                       int *handled, hipStream_t st, bool out_f32)
{
    *handled = 0;
    if (out_f32 && (ring != 0 || (s.W & 1))) return UWIE_OK;  // float32 output: float64 ring, paired stores
    if (s.W < 2 * k || s.H < 4 * k || s.B > 65535 || !(eps > 0.0)) return UWIE_OK;
    if (k != 10 && k != 15 && k != 20) return UWIE_OK;
    const double K2 = (double)k * k, scale = 1.0 / K2;
    PipeConsts cs{};
    cs.Ek = 255.0 * K2 * K2 * eps;
    if (ring == 1) {
        // |cov| <= sigma_p * sigma_I, sigma_p <= 0.45, sigma_I <= 0.5:  |a| <= 0.45 s / (s^2 + eps), s = sigma_I
        const double se = std::sqrt(eps);
        const double amax = (se <= 0.5 ? 0.45 / (2.0 * se) : 0.225 / (0.25 + eps)) * 1.02;
        const double hb = (0.45 + amax) * 1.02;   // |b - 0.55|
        const int Sa = (int)std::floor(std::log2(1073741824.0 / amax)), Sb = (int)std::floor(std::log2(1073741824.0 / hb));
        if (Sa < 28 || Sb < 28) ring = 0;  // wide a/b range (tiny eps): keep float64
        else {
            const double fa = std::ldexp(1.0, Sa), fb = std::ldexp(1.0, Sb);
            const double b0i = std::nearbyint(0.55 * fb);
            cs.fxa = fa;
            cs.fxb = scale * fb;
            cs.magic_b = kMagic - b0i;
            cs.kaI = scale / 255.0 / fa;
            cs.kb = scale / fb;
            cs.b0 = b0i / fb;
        }
    }
    if (ring == 0) {
        cs.kaI = scale / 255.0;
        cs.kb = scale;
        cs.b0 = 0.0;
    }
    // float64: the split-ring kernels.  Tuning gf_split = 0 keeps the general kernel.
    if (ring == 0) {
        int iy0, band, nb, rows;
        if (out_f32) {
            // UWIE_INTER_F32T: the same kernels with a float32 store (q is rounded once, after the clip)
            float *d_tf = reinterpret_cast<float *>(d_t);
            const bool split = guided_split_plan(s, k, &iy0, &band, &nb, &rows);
            const int border[6] = {iy0, iy0 + rows, band, 0, 0, 0};
            int rc = UWIE_OK;
            switch (k) {
            case 15:
                rc = split ? launch_split<15, float>(d_gray, d_t0, s, cs, d_tf, iy0, band, nb, s.H, st)
                           : launch_pipe<15, false, float>(d_gray, d_t0, s, cs, d_tf, st);
                break;
            case 20:
                rc = split ? launch_split<20, float>(d_gray, d_t0, s, cs, d_tf, iy0, band, nb, iy0 + rows, st)
                           : launch_pipe<20, false, float>(d_gray, d_t0, s, cs, d_tf, st);
                if (rc == UWIE_OK && split) rc = launch_pipe<20, false, float>(d_gray, d_t0, s, cs, d_tf, st, border);
                break;
            default:
                rc = split ? launch_split<10, float>(d_gray, d_t0, s, cs, d_tf, iy0, band, nb, iy0 + rows, st)
                           : launch_pipe<10, false, float>(d_gray, d_t0, s, cs, d_tf, st);
                if (rc == UWIE_OK && split) rc = launch_pipe<10, false, float>(d_gray, d_t0, s, cs, d_tf, st, border);
                break;
            }
            if (rc == UWIE_OK) *handled = 1;
            return rc;
        }
        if (guided_split_plan(s, k, &iy0, &band, &nb, &rows)) {
            if (k == 15) {
                UWIE_TRY_RC((launch_split<15, double>(d_gray, d_t0, s, cs, d_t, iy0, band, nb, s.H, st)));
            } else {
                // even window: rows [0, iy0) and [iy0 + rows, H) around the split kernel's whole periods
                const int border[6] = {iy0, iy0 + rows, band, 0, 0, 0};
                if (k == 20) {
                    UWIE_TRY_RC((launch_split<20, double>(d_gray, d_t0, s, cs, d_t, iy0, band, nb, iy0 + rows, st)));
                    UWIE_TRY_RC((launch_pipe<20, false, double>(d_gray, d_t0, s, cs, d_t, st, border)));
                } else {
                    UWIE_TRY_RC((launch_split<10, double>(d_gray, d_t0, s, cs, d_t, iy0, band, nb, iy0 + rows, st)));
                    UWIE_TRY_RC((launch_pipe<10, false, double>(d_gray, d_t0, s, cs, d_t, st, border)));
                }
            }
            *handled = 1;
            return UWIE_OK;
        }
    }
    int rc;
#define UWIE_PIPE_CASE(KK)                                                                                   \
    case KK:                                                                                                 \
        rc = ring == 1 ? launch_pipe<KK, true, double>(d_gray, d_t0, s, cs, d_t, st)                         \
                       : launch_pipe<KK, false, double>(d_gray, d_t0, s, cs, d_t, st);                       \
        break;
    switch (k) {
        UWIE_PIPE_CASE(10)
        UWIE_PIPE_CASE(15)
        UWIE_PIPE_CASE(20)
    default: return UWIE_OK;
    }
#undef UWIE_PIPE_CASE
    if (rc == UWIE_OK) *handled = 1;
    return rc;
}

}  // namespace uwie
