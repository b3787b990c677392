// Shared declarations of libuwie.so (gfx950 only; no CPU fallback anywhere in this library).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/uwie.h"

namespace uwie {

// ---------------------------------------------------------------- errors
void set_error(const char *fmt, ...);
#define UWIE_HIP_CHECK(expr)                                                            \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            ::uwie::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return UWIE_E_HIP;                                                          \
        }                                                                               \
    } while (0)
#define UWIE_LAUNCH_CHECK() UWIE_HIP_CHECK(hipGetLastError())
#define UWIE_REQUIRE(cond, msg)                 \
    do {                                        \
        if (!(cond)) {                          \
            ::uwie::set_error("%s", msg);       \
            return UWIE_E_INVALID;              \
        }                                       \
    } while (0)

// ---------------------------------------------------------------- constant tables (device copies owned by the context)
constexpr int kCastBinadeMin = -2;  // sequential-mean emulation handles accumulators >= 2^-2 in closed form
constexpr int kCastBinades = 34;    // e = -2 .. 31

struct LabTables {           // OpenCV's integer sRGB<->Lab tables (see tables.cpp)
    uint16_t gamma[256];     // sRGBGammaTab_b
    uint16_t cbrt[3072];     // LabCbrtTab_b
    int32_t fwd[9];          // RGB->XYZ/white, 12-bit
    uint8_t invgamma[4096];  // sRGBInvGammaTab_b (values <= 255)
    int32_t ltoyf[512];      // LabToYF_b
    int32_t abtoxz[36864];   // abToXZ_b
    int32_t inv[9];          // XYZ*white->RGB, 12-bit
};

struct CastTables {                         // rounding tables for the sequential float32 mean (k_entry.hip)
    uint32_t R[kCastBinades][256];          // RN(x_k / ulp_e)
    uint8_t tie[kCastBinades][256];         // 1 if x_k / ulp_e has fractional part exactly 1/2
    uint32_t RT[256][kCastBinades];         // R transposed, tie in bit 31 (R < 2^26): one coalesced row per value
    uint2 RT2[128][kCastBinades];           // values 2j, 2j+1: x = low 13 bits of R as a 16-bit pair, y = R >> 13 likewise
    uint8_t tiebin[256];                    // the one binade index in which value k ties (255: none in range)
};

void build_lab_tables(LabTables *t);
void build_cast_tables(CastTables *t);

// ---------------------------------------------------------------- optional per-kernel timing (prof.hip)
struct Profiler;
Profiler *prof_create();
void prof_destroy(Profiler *p);
void prof_enable(Profiler *p, bool on);
void prof_filter(Profiler *p, const char *name);
void prof_bind(Profiler *p);  // makes p the recorder of this host thread (nullptr / disabled = no recording)
int prof_collect(Profiler *p);
int prof_row(Profiler *p, int i, const char **name, double *ms, int *calls);
class ProfScope {  // records a start event now and a stop event when it goes out of scope
public:
    ProfScope(const char *name, hipStream_t st);
    ~ProfScope();
    ProfScope(const ProfScope &) = delete;

private:
    int rec_;
    hipStream_t st_;
};
#define UWIE_PROF_CAT2(a, b) a##b
#define UWIE_PROF_CAT(a, b) UWIE_PROF_CAT2(a, b)
#define UWIE_PROF(name, st) ::uwie::ProfScope UWIE_PROF_CAT(_uwie_prof_, __LINE__)(name, st)
// every kernel launch goes through this: named timing scope + launch
#define UWIE_LAUNCH(kernel, grid, block, lds, st, ...)                      \
    do {                                                                    \
        UWIE_PROF(#kernel, st);                                             \
        hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);      \
    } while (0)

}  // namespace uwie

namespace uwie {
// Route selectors of one context.  The defaults are what ships and what uwie_params cannot express: which of several
// equivalent routes a stage takes (tests force the fallback routes through uwie_set_tuning; profiles/ scripts compare
// them).  Environment variables UWIE_<NAME> are read ONCE, in uwie_create -- never per call.
struct Tuning {
    int gf_pipe = 1;            // guided filter: wavefront kernels of k_guided_pipe.hip (0: the LDS-tiled strip kernel)
    int gf_split = 1;           // ... their split-ring form for large jobs (0: the general kernel alone)
    int gf_bands = 0;           // ... bands per strip (0: chosen from the job's size)
    int select_generic = 0;     // percentile selection: the three-digit key sweeps alone
    int restore_store = 0;      // strategies 1-2 / dict dehazing: keep the restored image in planes instead of recomputing it
    int lin_predict3 = 0;       // strategy 3: predicted windows as for strategies 1-2
    int lin_cap = 0;            // candidate list capacity (0: default; small values force the overflow fallback)
    int lin_no_predict = 0;     // no predicted windows: always the collecting sweep
    int q_hist = 1;             // quadtree levels decided from byte histograms when the interval test allows (2: never allows)
    int lin_predict_shift = 0;  // predicted windows moved by this many bins (large: every prediction misses)
    int streams = 1;            // uwie_enhance_u8: sub-batches on this many internal streams (1 .. 4)
    int canny_prepass = 1;      // quadtree: the streaming "any strong pixel?" pass before Canny
    int gf_fuse = 0;            // ... with t0 computed from the frame's bytes inside it (k_guided_split8: -0.08 ms per 4K x 64 step, opt-in; 0: k_trans_init + t0 plane)
    int rank_sweep = 1;         // strategies 1-2: the rank-counting restore sweep on jobs of >= 16 MP (0: the histogram sweep; 2: any size)
    int canny_fault_inject = 0; // tests only: k_canny_gradnms leaves out the root labels (the round-3 defect): uwie_device_status must report it
    int exact_fused = 1;         // gf_exact = 1, k = 15: rows and columns of the first box filter in one kernel (0: separate passes)
    int entry_fuse = 1;         // frames with W % 8 == 0: the level-0 quadrant histograms come out of cast detection's chunk pass and the
                                // gray plane out of the Canny pre-pass (0: k_q_hist<gray> + k_canny_strong as in rounds 3 - 4)
};
const Tuning &tune();  // tuning of the context whose entry point is running on this host thread (defaults outside one)
uwie_ctx *current_ctx();
}  // namespace uwie

struct uwie_ctx {
    int device;
    uwie::Tuning tune;
    bool attr_q_tail = false;   // > 64 KB LDS attributes set on this context's device: k_q_tail,
    bool attr_cast_resolve = false;  // k_cast_resolve (the rounding table of every binade: 70 KB),
    int attr_gf_fast = 0;       // k_guided_fast<TH> (bit TH)
    int attr_gf_split8 = 0;     // k_guided_split8<15, double / float> (bits 1 / 2)
    uwie::LabTables *d_lab;
    uwie::CastTables *d_cast;
    uwie::Profiler *prof;
    uint32_t *d_status = nullptr;  // device status word: UWIE_STATUS_* bits set by kernels that find an invariant violated
    // two-way batch pipelining (uwie_enhance_u8): helper streams and the fork / join events, created on first use
    hipStream_t aux[4];
    hipEvent_t fork, join[4];
    bool aux_ready;
};

namespace uwie {

// ---------------------------------------------------------------- workspace carving
struct Carver {
    char *base;
    size_t off;
    explicit Carver(void *b) : base(static_cast<char *>(b)), off(0) {}
    template <typename T>
    T *take(size_t count)
    {
        off = (off + 255) & ~size_t(255);
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
    size_t total() const { return (off + 255) & ~size_t(255); }
};

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
// blocks of 256 threads for a grid-stride loop over n items
static inline int grid_for(size_t n, size_t cap = 8192)
{
    size_t g = (n + 255) / 256;
    if (g > cap) g = cap;
    return g < 1 ? 1 : (int)g;
}

// ---------------------------------------------------------------- region descriptor used by the Canny / quadtree kernels
struct Region {  // a rectangle of image `img`; rows == 0 marks an inactive region
    int32_t img, y0, x0, rows, cols;
};

// ---------------------------------------------------------------- stage launchers (each enqueues on `stream`, returns UWIE_*)
struct Shape {
    int B, H, W;
    size_t npx() const { return (size_t)H * W; }
};

// k_entry.hip
size_t cast_ws_bytes(Shape s);
struct EntryFuse {       // what cast detection's chunk pass leaves for the quadtree (tuning entry_fuse)
    uint32_t *qpart;     // the level-0 quadrants' shares of every chunk
    uint8_t *gray;       // the gray plane, or nullptr (then the quadtree's level 0 writes it: k_gray_strong)
    int32_t *guess;      // [B] the cast kinds the plane was first written for
    int gray_shift;
};
int launch_cast_classify(uwie_ctx *ctx, const uint8_t *d_in, Shape s, int32_t *d_kind, float *d_mean, void *ws,
                         hipStream_t st, const EntryFuse *ef = nullptr);
size_t quad_part_bytes(Shape s);  // the level-0 quadrants' shares of every chunk (tuning entry_fuse)
int launch_quad_hist_reduce(const uint32_t *qpart, Shape s, uint32_t *d_hist, hipStream_t st);
int launch_set_kind(int32_t *d_kind, int B, int kind, hipStream_t st);
int launch_normalise_correct(const uint8_t *d_in, const int32_t *d_kind, float *d_out, Shape s, hipStream_t st);
int launch_quant_gray(const uint8_t *d_in, const int32_t *d_kind, uint8_t *d_gray, Shape s, int gray_shift,
                      hipStream_t st);

// k_airlight.hip
size_t airlight_ws_bytes(Shape s);
// make_gray_shift != 0: d_gray is written on the way (the level-0 sums pass or k_quant_gray) instead of read
// qpart: the level-0 quadrant shares that launch_cast_classify left for these frames (entry_fuse_takes), or nullptr
int launch_airlight(uwie_ctx *ctx, const uint8_t *d_in, const int32_t *d_kind, uint8_t *d_gray, Shape s, int min_size,
                    float *d_A, void *d_trace, void *ws, hipStream_t st, int make_gray_shift = 0, const uint32_t *qpart = nullptr);
bool entry_fuse_takes(Shape s, int min_size);

// k_canny.hip  (regions: device array of nreg Region; max_rows/max_cols bound every region)
size_t canny_ws_bytes(Shape s);
int launch_canny(const uint8_t *d_gray, Shape s, const Region *d_regions, int nreg, int max_rows, int max_cols, int low,
                 int high, uint32_t *d_count, uint8_t *d_edges, void *ws, hipStream_t st, bool count_is_zeroed = false,
                 bool strong_is_zeroed = false, bool prepass_done = false);
bool gray_strong_takes(Shape s);
int launch_gray_strong(const uint8_t *d_in, const int32_t *d_kind, uint8_t *d_gray, Shape s, const Region *d_regions, int nreg,
                       int max_rows, int max_cols, int high, int gray_shift, void *ws, hipStream_t st);
int launch_hist_strong(const uint8_t *d_in, const uint8_t *d_gray, Shape s, const Region *d_regions, int nreg, int max_rows,
                       int max_cols, int high, uint32_t *d_hist, void *ws, hipStream_t st);
uint32_t *canny_strong_flags(void *ws, Shape s);  // the pre-pass flags inside a Canny workspace ([regions])
int launch_make_full_regions(Region *d_regions, Shape s, hipStream_t st);

// k_airlight.hip: NumPy-order sum / mean / sum of squared deviations per region and channel
int launch_region_stats(const uint8_t *d_in, const int32_t *d_kind, const Region *d_regs, int nreg, int max_rows,
                        int max_cols, Shape s, float *csum, int maxChunks, float *tot, float *mean, float *vtot,
                        hipStream_t st);
// k_codes.hip: hist[b][c][256] += counts of the u8 frame (caller zeroes)
int launch_frame_hist(const uint8_t *d_in, Shape s, uint32_t *d_hist, hipStream_t st);
// k_features.hip: vgg_16_UIE.extract_all_features (79 floats per image)
size_t features_ws_bytes(Shape s);
int launch_features_u8(const uint8_t *d_in, Shape s, float *d_out, void *ws, hipStream_t st);

// k_quality.hip: quality_assessment.QualityAssessment scores, [B][9] float64 (8 scores + weighted total)
size_t quality_ws_bytes(Shape s);
int launch_quality_scores(uwie_ctx *ctx, const uint8_t *d_u8, const float *d_f32, Shape s, int gray_shift,
                          const double *weights8, double *d_scores, void *ws, hipStream_t st);

// best[b] = first argmax over the n strategies of scores[k][b][8]; d_out (optional) [B][H][W][3] = d_all[best[b]][b]
int launch_pick_best(const double *d_scores, int n, Shape s, const uint8_t *d_all, int32_t *d_best, uint8_t *d_out, hipStream_t st);

// k_guided.hip
size_t guided_ws_bytes(Shape s);
size_t box_ws_bytes(Shape s);
int launch_trans_init(const uint8_t *d_in, const int32_t *d_kind, const float *d_A, Shape s, float omega, float norm_eps,
                      int pre_clip, float *d_t0, hipStream_t st);
int launch_box_filter_f64(const double *d_src, double *d_dst, Shape s, int k, void *ws, hipStream_t st);
int launch_guided(const uint8_t *d_gray, const float *d_t0, Shape s, int k, double eps, double *d_t, void *ws,
                  hipStream_t st);

bool guided_fast_handles(Shape s, int k);
// k_guided_fast.hip: fused float64 guided filter (free summation order); *handled = 0 -> use launch_guided
// ring_fx: the caller guarantees 0.1 <= t0 <= 1 (pre-clipped transmission) and accepts the fixed-point a/b ring
int launch_guided_fast(const uint8_t *d_gray, const float *d_t0, Shape s, int k, double eps, double *d_t, int *handled,
                       hipStream_t st, bool ring_fx = false);
// k_guided_pipe.hip: software-pipelined wavefront kernel for k in {10, 15, 20}; ring 0 = float64, 1 = fixed-point int32
bool guided_split_plan(Shape s, int k, int *iy0, int *band, int *nb, int *rows = nullptr);
// out_f32: d_t is written as float32 (UWIE_INTER_F32T); only the pipe / split kernels do that, so *handled = 0 then means
// "not taken, nothing written"
int launch_guided_pipe(const uint8_t *d_gray, const float *d_t0, Shape s, int k, double eps, int ring, double *d_t,
                       int *handled, hipStream_t st, bool out_f32 = false);
// the same filter with t0 = 1 - omega * min_c(img / (A + eps)) [clipped] computed inside it from the u8 frame (round 4)
bool guided_fused_takes(Shape s, int k);
int launch_guided_fused(const uint8_t *d_gray, const uint8_t *d_rgb, const int32_t *d_kind, const float *d_A, float omega,
                        float norm_eps, int pre_clip, Shape s, int k, double eps, double *d_t, hipStream_t st, bool out_f32);

// k_select.hip
constexpr int kMaxPct = 4;  // percentiles per call
size_t select_ws_bytes(Shape s);
// d_vals: float32 image, HWC ([B][H][W][3], planar = 0) or planar ([B][3][H][W], planar = 1); d_out [B][3][nq]
int launch_percentiles_f32(const float *d_vals, int planar, Shape s, const double *q_percent, int nq, float *d_out,
                           void *ws, hipStream_t st);

// selection in pieces, for producers that fuse the first histogram sweep (digit = f32_key(v) >> 21, 2048 bins,
// accumulated into plan.ghist[(b*3 + c) * kSelGroupStride + digit])
constexpr int kSelGroupStride = 2 * kMaxPct * 2048;
constexpr int kSelOsStride = 2 * kMaxPct;  // order statistics per (image, channel) in SelectPlan::os
struct SelectPlan {
    void *state;
    uint32_t *ghist;
    void *os;              // order statistics, float or double [B*3][8]
    double t[kMaxPct];     // lerp weights (exactly representable values of the data's dtype)
    int nq;
    bool is64;
    // linear-digit path (select_lin_*): per (image, channel) state, candidate lists and fallback flags
    void *lin;
    float *lists;          // [B*3][kLinLists][cap]
    uint32_t *flags;       // [B*3] 1 = a candidate list overflowed, use the generic sweeps
    uint32_t cap;
    bool predicted;        // the producer files the predicted windows: the collecting sweep is the rare fallback
    uint32_t ranks[2 * kMaxPct];
};
// Selection on clipped [0, 1] float32 planes with ONE sweep after the producer's histogram: the first digit is a linear
// 2050-way split of [0, 1] (exact 0 and exact 1 get their own bins), which a whole frame spreads over thousands of bins,
// so the elements of the few target bins are collected during the sweep and the rest of the selection runs on those
// short lists.  lin_digit() is the split; the producer adds its counts to plan.ghist[(b*3+c)*kSelGroupStride + digit].
constexpr int kLinBins = 2050;
__host__ __device__ inline uint32_t lin_digit(float x)
{
    if (!(x > 0.0f)) return 0;          // zeros (and anything not above zero)
    if (x >= 1.0f) return kLinBins - 1;
    const uint32_t d = (uint32_t)(x * 2048.0f);  // exact product, truncation: monotone in x
    return 1 + (d > 2047u ? 2047u : d);
}
__host__ __device__ inline uint32_t lin_digit(double x)  // the same split for the float64 planes of the ES surface
{
    if (!(x > 0.0)) return 0;
    if (x >= 1.0) return kLinBins - 1;
    const uint32_t d = (uint32_t)(x * 2048.0);
    return 1 + (d > 2047u ? 2047u : d);
}
// Per (image, channel) state of the linear-digit selection.  With a prediction (select_lin_begin(..., predict)) the
// producer already files the elements of two windows of bins, each around the predicted position of a percentile's
// ranks, into one list per window during its own sweep; the scan then checks the prediction against the exact
// histogram, and only planes it missed need the collecting sweep (whose groups reuse the lists).
constexpr int kLinLists = 2 * kMaxPct;   // candidate lists per (image, channel)
constexpr uint32_t kLinNoWin = 0x80000000u;
struct LinState {
    uint32_t rr[2 * kMaxPct];    // rank of the query inside its bin
    uint32_t qbin[2 * kMaxPct];  // bin of the query
    uint32_t gid[2 * kMaxPct];   // list of the query; kLinDone: answered by the scan
    uint32_t gbin[2 * kMaxPct];  // bin of the collecting sweep's group g (list g)
    uint32_t gcount[kLinLists];  // elements filed in each list
    uint32_t ngroups;            // groups the collecting sweep has to fill (0: the prediction covered every query)
    uint32_t wlo[kMaxPct], wspan[kMaxPct];  // predicted windows (one per percentile, merged when they touch): bins
                                            // wlo .. wlo + wspan (wlo = kLinNoWin: none), list = window
    uint32_t below[kMaxPct];     // (rank-counting sweep, k_restore_rank) elements in the bins below window w
};
// The rank route has two windows per plane, not kLinLists lists: the same memory as 2 lists of kRankCapMul x cap elements (a window
// of whole histogram bins holds 1 - 2 % of a plane whose values cluster on a few levels -- hazy noise frames: 139 K of 8.3 M -- and
// the histogram route's cap of 1/64 overflowed there on two planes in 192)
constexpr int kRankCapMul = 4;
constexpr uint32_t kLinAnyBin = 0xfffffffeu;  // LinState::qbin: the query's list is its whole window, rr its rank inside it
struct RestoreSrc;
// predict != nullptr: the target bins are predicted from a subsample of the restored image (k_lin_sample)
int select_lin_begin(Shape s, const double *q_percent, int nq, void *ws, hipStream_t st, SelectPlan *plan,
                     const RestoreSrc *predict = nullptr);
// src != nullptr: the values are recomputed from *src; d_planar is then only written (and read back) for planes that
// fall back to the generic sweeps
int select_lin_run(const SelectPlan &plan, float *d_planar, Shape s, hipStream_t st, const RestoreSrc *src = nullptr);
// float64 planes (ES surface): linear first digit by the producer, one collecting sweep, finish on the lists
int select_lin_begin64(Shape s, const double *q_percent, int nq, void *ws, hipStream_t st, SelectPlan *plan,
                       const RestoreSrc *predict = nullptr);
int select_lin_run64(const SelectPlan &plan, double *d_planar, Shape s, hipStream_t st, const RestoreSrc *src = nullptr);
int select_begin(Shape s, const double *q_percent, int nq, void *ws, hipStream_t st, SelectPlan *plan);
int select_run(const SelectPlan &plan, const float *d_vals, int planar, Shape s, bool pass1_done, hipStream_t st);
int select_lerp(const SelectPlan &plan, Shape s, float *d_out, hipStream_t st);                     // [B][3][nq]
int select_lerp_chain(const SelectPlan &plan, Shape s, float eps, float *d_pct4, hipStream_t st);   // [B][3][4]
// explicit sorted positions per image (vgg_16_UIE.py:78-82): os[bc*8 + 0/1] = sorted[int(L_low/100*n)], sorted[int(L_high/100*n)]
int select_begin_stretch_ranks(Shape s, const float *d_params, int stride, void *ws, hipStream_t st, SelectPlan *plan);
// k_diffenh.hip: DifferentiableEnhancement.forward (vgg_16_UIE.py:32-128) after the selection
int launch_diff_enhance(const float *d_img, int planar, Shape s, const float *d_params, int flags, const float *d_os,
                        float *d_out, hipStream_t st);
// float64 data (ES surface): first digit = f64_key(v) >> 53
int select_begin64(Shape s, const double *q_percent, int nq, void *ws, hipStream_t st, SelectPlan *plan);
int select_run64(const SelectPlan &plan, const double *d_vals, int planar, Shape s, bool pass1_done, hipStream_t st);
int select_lerp64(const SelectPlan &plan, Shape s, double *d_out, hipStream_t st);

// k_fused.hip: the fused tail of the dehazing strategies
// what the restored image (six_stadigy.py:183-188) is made of; consumers may recompute it from here (restore.h)
struct RestoreSrc {
    const uint8_t *in;    // [B][H][W][3]
    const int32_t *kind;  // [B] cast kinds or nullptr
    const float *A;       // [B][3]
    const double *t;      // [B][H][W]; float32 data when t32 is set (uwie_params.inter_dtype = UWIE_INTER_F32T)
    int t32 = 0;
};
// plan != nullptr: the linear-digit histogram goes to plan->ghist and the elements of the predicted windows to the lists
// t32: d_t holds float32 data (RestoreSrc::t32)
int launch_restore_planar_hist(const uint8_t *d_in, const int32_t *d_kind, const float *d_A, const double *d_t, Shape s,
                               float *d_planar, uint32_t *d_ghist, hipStream_t st, bool linear = false,
                               const uint32_t *d_only = nullptr, const SelectPlan *plan = nullptr, int t32 = 0);
// the rank-counting sweep (k_restore_rank): counts below each predicted window + the windows' members, no histogram
int launch_restore_rank(const RestoreSrc &src, Shape s, const SelectPlan &plan, hipStream_t st);
int select_rank_run(const SelectPlan &plan, float *d_planar, Shape s, hipStream_t st, const RestoreSrc &src);
size_t tail_ws_bytes(Shape s, int tx, int ty);
// d_pct: [B][3][pct_stride] = lo1, hi1 [, lo2, hi2]; two = second stretch present; gamma_mode 0/1/2
int launch_tail_clahe(uwie_ctx *ctx, const float *d_planar, const float *d_pct, int pct_stride, float eps, int two,
                      Shape s, double clip, int tx, int ty, int gamma_mode, double gamma, uint8_t *d_out_u8,
                      float *d_out_f32, void *ws, hipStream_t st, const RestoreSrc *src = nullptr);
int launch_codes_lab_lut(uwie_ctx *ctx, const uint8_t *d_in, const uint8_t *d_code_lut, Shape s, double clip, int tx, int ty,
                         uint8_t *d_lab, uint8_t *d_tile_lut, hipStream_t st);
int launch_clahe_apply_codes(uwie_ctx *ctx, const uint8_t *d_lab, const uint8_t *d_tile_lut, Shape s, double clip, int tx,
                             int ty, const uint8_t *d_fin_code, const float *d_fin_val, uint8_t *d_out_u8, float *d_out_f32,
                             uint8_t *d_codes_out, uint32_t *d_hist, hipStream_t st);
int launch_tail_plain(const float *d_planar, const float *d_pct, int pct_stride, float eps, int two, Shape s,
                      int gamma_mode, double gamma, uint8_t *d_out_u8, float *d_out_f32, hipStream_t st);
// ES surface (float64): recover_image (ES:237-249) -> planar float64 + first select digit; color_enhancement
// (ES:269-270, eps 1e-10) [-> gamma_correction (ES:284-285)] -> (y*255).astype(u8) (main.py:155) / float32 copy
int launch_recover64_planar_hist(const uint8_t *d_in, const float *d_A, const double *d_t, Shape s, double *d_planar,
                                 uint32_t *d_ghist, hipStream_t st, bool linear = false, const uint32_t *d_only = nullptr,
                                 const SelectPlan *plan = nullptr);
int launch_tail_plain64(const double *d_planar, const double *d_pct, Shape s, int apply_gamma, double gamma,
                        uint8_t *d_out_u8, float *d_out_f32, hipStream_t st, const RestoreSrc *src = nullptr,
                        double *d_out_f64 = nullptr);

// k_codes.hip: strategies 4-6 of six_stadigy.py and the clahe / histogram-equalisation strategies of
// enhancement_strategies.py, evaluated on 8-bit codes (per image and channel LUT chains + histograms)
size_t codes_ws_bytes(Shape s, int tx, int ty);
// quantised: d_in is (img * 255).astype(u8) of a general float image instead of the u8 frame the float image came from
int launch_code_strategy(uwie_ctx *ctx, const uint8_t *d_in, const int32_t *d_kind, Shape s, const uwie_params *p,
                         uint8_t *d_out_u8, float *d_out_f32, void *ws, hipStream_t st, double *d_out_f64 = nullptr,
                         bool quantised = false);

// k_float.hip: the front half of the strategies on general (not u8-derived) float images, T = float or double
size_t float_airlight_ws_bytes(Shape s);
template <class T> int launch_float_prepare(const T *d_x, const int32_t *d_kind, T *d_xc, uint8_t *d_q, Shape s, hipStream_t st);
template <class T> int launch_float_cast_classify(const T *d_x, Shape s, int32_t *d_kind, float *d_mean, hipStream_t st);
template <class T> int launch_float_airlight(const T *d_x, const uint8_t *d_gray, Shape s, int min_size, T *d_A, void *ws, hipStream_t st);
template <class T> int launch_float_trans_init(const T *d_x, const T *d_A, Shape s, double omega, double norm_eps, int pre_clip, T *d_t0,
                                               hipStream_t st);
template <class T, class OUT> int launch_float_restore(const T *d_x, const T *d_A, const double *d_t, Shape s, OUT *d_out, int planar,
                                                       hipStream_t st);
int launch_guided_p64(const uint8_t *d_gray, const double *d_t0, Shape s, int k, double eps, double *d_t, void *ws, hipStream_t st);

// k_tail.hip
int launch_restore(const uint8_t *d_in, const int32_t *d_kind, const float *d_A, const double *d_t, Shape s,
                   float *d_out, hipStream_t st);
// out = clip((x - lo) / (hi - lo + eps), 0, 1) with lo = d_pct[(b*3+c)*pct_stride + lo_idx], hi likewise
int launch_stretch_apply_f32(const float *d_img, const float *d_pct, int pct_stride, int lo_idx, int hi_idx, float eps,
                             float *d_out, Shape s, hipStream_t st);
int launch_gamma_f32(const float *d_img, float *d_out, size_t n, double g, int mode, hipStream_t st);
int launch_quantise_u8(const float *d_img, uint8_t *d_out, size_t n, hipStream_t st);

// k_clahe.hip
size_t clahe_ws_bytes(Shape s, int tx, int ty);
int launch_rgb2gray_u8(const uint8_t *d_rgb, uint8_t *d_gray, size_t n, int shift, hipStream_t st);
int launch_rgb2lab_u8(uwie_ctx *ctx, const uint8_t *d_rgb, uint8_t *d_lab, size_t n, hipStream_t st);
int launch_lab2rgb_u8(uwie_ctx *ctx, const uint8_t *d_lab, uint8_t *d_rgb, size_t n, hipStream_t st);
int launch_clahe_plane_u8(const uint8_t *d_plane, uint8_t *d_out, Shape s, double clip, int tx, int ty, void *ws,
                          hipStream_t st);
int launch_clahe_f32(uwie_ctx *ctx, const float *d_img, float *d_out, Shape s, double clip, int tx, int ty, void *ws,
                     hipStream_t st);
int launch_equalize_hist_u8(const uint8_t *d_plane, uint8_t *d_out, Shape s, void *ws, hipStream_t st);

}  // namespace uwie
