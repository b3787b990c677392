// C ABI of libuwie.so (include/uwie.h): context, parameter defaults, workspace sizing, the per-stage entry points
// and the strategy pipelines.  Everything here only enqueues kernels on the caller's stream.
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "common.h"

#include <cstdlib>

namespace uwie {

static thread_local char g_err[512] = "";
static thread_local uwie_ctx *g_ctx = nullptr;  // context of the entry point running on this thread
static const Tuning g_default_tuning{};

const Tuning &tune() { return g_ctx ? g_ctx->tune : g_default_tuning; }
uwie_ctx *current_ctx() { return g_ctx; }

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

namespace {

// Every exported function that launches work runs inside one of these: the context's device becomes current (kernels,
// events and helper streams belong to it) and is put back afterwards, and the context's tuning is what tune() answers.
class CallScope {
public:
    explicit CallScope(uwie_ctx *ctx) : prev_ctx_(g_ctx)
    {
        g_ctx = ctx;
        if (ctx && hipGetDevice(&prev_dev_) == hipSuccess && prev_dev_ != ctx->device) {
            if (hipSetDevice(ctx->device) == hipSuccess) switched_ = true;
            else ok_ = false;
        }
    }
    ~CallScope()
    {
        if (switched_) (void)hipSetDevice(prev_dev_);
        g_ctx = prev_ctx_;
    }
    CallScope(const CallScope &) = delete;
    bool ok() const { return ok_; }

private:
    uwie_ctx *prev_ctx_;
    int prev_dev_ = -1;
    bool switched_ = false, ok_ = true;
};
#define UWIE_SCOPE(ctx)                                                       \
    CallScope _uwie_scope(ctx);                                               \
    if (!_uwie_scope.ok()) {                                                  \
        set_error("cannot make device %d current", (ctx)->device);           \
        return UWIE_E_HIP;                                                    \
    }

struct KnobName { const char *name; int Tuning::*field; };
const KnobName kKnobs[] = {{"gf_pipe", &Tuning::gf_pipe}, {"gf_split", &Tuning::gf_split}, {"gf_bands", &Tuning::gf_bands},
                           {"select_generic", &Tuning::select_generic}, {"restore_store", &Tuning::restore_store},
                           {"lin_predict3", &Tuning::lin_predict3}, {"lin_cap", &Tuning::lin_cap},
                           {"lin_no_predict", &Tuning::lin_no_predict}, {"q_hist", &Tuning::q_hist}, {"lin_predict_shift", &Tuning::lin_predict_shift},
                           {"streams", &Tuning::streams}, {"canny_prepass", &Tuning::canny_prepass},
                           {"canny_fault_inject", &Tuning::canny_fault_inject}, {"rank_sweep", &Tuning::rank_sweep},
                           {"gf_fuse", &Tuning::gf_fuse}, {"exact_fused", &Tuning::exact_fused}, {"entry_fuse", &Tuning::entry_fuse}};

void tuning_from_env(Tuning *t)  // uwie_create only
{
    for (const KnobName &k : kKnobs) {
        char var[64] = "UWIE_";
        size_t n = 5;
        for (const char *c = k.name; *c && n + 1 < sizeof var; ++c) var[n++] = (char)(*c >= 'a' && *c <= 'z' ? *c - 32 : *c);
        var[n] = 0;
        if (const char *e = getenv(var)) t->*(k.field) = atoi(e);
    }
}

bool shape_ok(int B, int H, int W)
{
    if (B < 1 || H < 1 || W < 1) return false;
    const long long npx = (long long)H * W;
    return npx < (1ll << 30) && (long long)B * npx < (1ll << 40);
}

#define UWIE_CHECK_SHAPE(B, H, W) UWIE_REQUIRE(shape_ok(B, H, W), "batch/H/W out of range")
#define UWIE_CHECK_WS(need)                                                                           \
    do {                                                                                              \
        if ((need) > 0 && (!d_workspace || workspace_bytes < (need))) {                               \
            set_error("workspace too small: need %zu bytes, got %zu", (size_t)(need), workspace_bytes); \
            return UWIE_E_WORKSPACE;                                                                  \
        }                                                                                             \
    } while (0)
#define UWIE_TRY(call)            \
    do {                          \
        int _rc = (call);         \
        if (_rc != UWIE_OK) return _rc; \
    } while (0)

// Buffers of one enhance() call, carved from the caller's workspace.
struct Pipe {
    int32_t *kind;
    float *A;
    float *pct;
    float *F;       // float32 working image (planar on the dehazing paths)
    double *F64;    // float64 planar image (dict-surface dehazing only)
    double *pct64;
    uint8_t *gray;
    float *t0;
    double *t;
    void *scratch;
    size_t scratch_bytes;
    uint32_t *qpart;  // level-0 quadrant shares of cast detection's chunks (tuning entry_fuse; inside the scratch, behind what cast
                      // detection and the quadtree use), or nullptr
    int32_t *guess;   // ... and the cast kinds its gray plane was first written for
};

size_t max5(size_t a, size_t b, size_t c, size_t d, size_t e)
{
    size_t m = a;
    if (b > m) m = b;
    if (c > m) m = c;
    if (d > m) m = d;
    if (e > m) m = e;
    return m;
}

bool dehazes(const uwie_params *p)
{
    if (p->surface == UWIE_SURFACE_SIX) return p->strategy >= 1 && p->strategy <= 3;
    return p->strategy >= UWIE_DICT_STRONG_DEHAZING && p->strategy <= UWIE_DICT_LIGHT_ENHANCEMENT;
}

Pipe carve_pipe(Carver &c, Shape s, const uwie_params *p)
{
    Pipe P{};
    const size_t n = (size_t)s.B * s.npx();
    P.kind = c.take<int32_t>(s.B);
    P.A = c.take<float>((size_t)s.B * 3);
    P.pct = c.take<float>((size_t)s.B * 3 * kMaxPct);
    P.guess = c.take<int32_t>(s.B);
    const bool dz = !p || dehazes(p);
    const bool dict_dz = p && p->surface == UWIE_SURFACE_DICT && dz;
    if (dict_dz) {
        P.F64 = c.take<double>(n * 3);
        P.pct64 = c.take<double>((size_t)s.B * 3 * kMaxPct);
    } else if (!p || (p->surface == UWIE_SURFACE_SIX && dz && (p->strategy == 3 || tune().restore_store || tune().select_generic))) {
        // The float32 planes of the restored image (12 B/px): strategy 3's tail reads them, the stored-plane and key-sweep
        // routes (tuning) work on them.  Strategies 1-2 recompute the image in every sweep and -- round 4 -- in the
        // selection's fallback too, and the code-domain strategies never had a use for them: nothing reserved
        // (a 4K x 64 strategy-2 call: 43 -> 31 B/px).  tune() is the calling context's tuning inside an entry point and the
        // default tuning in uwie_workspace_bytes, which has no context: uwie_workspace_bytes_ctx answers for a context.
        P.F = c.take<float>(n * 3);
    }
    const int tx = p ? p->tiles_x : 8, ty = p ? p->tiles_y : 8;
    // the exact-order guided filter materialises six float64 planes; the default kernels keep everything on chip
    const bool gf_planes = dz && (!p || p->gf_exact || !guided_fast_handles(s, p->gf_ksize));
    if (dz) {
        P.gray = c.take<uint8_t>(n);
        // The raw transmission lives from k_trans_init to the end of the guided filter.  The scratch region is idle exactly then
        // (the quadtree is done with it, the selection has not started), so t0 borrows its first 4 B/px -- unless the exact-order
        // filter is on, whose six planes are IN the scratch while it reads t0.
        if (gf_planes) P.t0 = c.take<float>(n);
        P.t = c.take<double>(n);
    }
    P.scratch_bytes = max5(cast_ws_bytes(s), dz ? airlight_ws_bytes(s) : 0, gf_planes ? guided_ws_bytes(s) : 0,
                           select_ws_bytes(s), clahe_ws_bytes(s, tx > 0 ? tx : 8, ty > 0 ? ty : 8));
    const size_t cw = codes_ws_bytes(s, tx > 0 ? tx : 8, ty > 0 ? ty : 8);
    if (cw > P.scratch_bytes) P.scratch_bytes = cw;
    if (dz && !gf_planes && P.scratch_bytes < n * sizeof(float)) P.scratch_bytes = n * sizeof(float);
    // round 4: cast detection's chunk pass leaves the level-0 quadrant histograms of the quadtree behind (six_stadigy surface with
    // cast detection on; frames entry_fuse_takes).  They are written before the quadtree starts and read by its first launches.
    size_t qoff = 0;
    if (dz && p && p->surface == UWIE_SURFACE_SIX && p->cast_correct && p->forced_cast < 0 && entry_fuse_takes(s, p->min_size)) {
        qoff = (std::max(cast_ws_bytes(s), airlight_ws_bytes(s)) + 255) & ~size_t(255);
        if (P.scratch_bytes < qoff + quad_part_bytes(s)) P.scratch_bytes = qoff + quad_part_bytes(s);
    }
    P.scratch = c.take<char>(P.scratch_bytes);
    if (dz && !gf_planes) P.t0 = static_cast<float *>(P.scratch);
    if (qoff && P.scratch) P.qpart = reinterpret_cast<uint32_t *>(static_cast<char *>(P.scratch) + qoff);
    return P;
}

// A parameter set whose carve_pipe layout holds every one of `n` six_stadigy sets (uwie_enhance_all_u8 runs them from
// one workspace): dehazing layout, the exact-order guided filter's planes if any dehazing set needs them (gf_exact, or
// a window the fused kernels do not take), the largest CLAHE tile grid.
uwie_params merged_params(const uwie_params *ps, int n, Shape s)
{
    uwie_params m = ps[0];
    m.surface = UWIE_SURFACE_SIX;
    m.strategy = 2;
    for (int i = 0; i < n; ++i) {
        if (ps[i].surface == UWIE_SURFACE_SIX && ps[i].strategy == 3) m.strategy = 3;  // (its tail reads the stored planes)
        if (ps[i].tiles_x > m.tiles_x) m.tiles_x = ps[i].tiles_x;
        if (ps[i].tiles_y > m.tiles_y) m.tiles_y = ps[i].tiles_y;
        if (ps[i].surface == UWIE_SURFACE_SIX && ps[i].strategy <= 3 &&
            (ps[i].gf_exact || !guided_fast_handles(s, ps[i].gf_ksize)))
            m.gf_exact = 1;
    }
    return m;
}

// *t_is_f32 (optional): set when the transmission was written as float32 (UWIE_INTER_F32T and a window / frame the
// wavefront kernels take); everything downstream reads it through RestoreSrc::t32
int stage_guided(uwie_ctx *ctx, const Pipe &P, Shape s, const uwie_params *p, hipStream_t st, int *t_is_f32 = nullptr)
{
    int handled = 0;
    if (t_is_f32) *t_is_f32 = 0;
    if (t_is_f32 && p->inter_dtype == UWIE_INTER_F32T && !p->gf_exact && p->surface == UWIE_SURFACE_SIX && tune().gf_pipe) {
        UWIE_TRY(launch_guided_pipe(P.gray, P.t0, s, p->gf_ksize, p->gf_eps, 0, P.t, &handled, st, true));
        if (handled) {
            *t_is_f32 = 1;
            return UWIE_OK;
        }
    }
    // fixed-point a/b ring: only with the pre-clipped transmission of the six_stadigy surface (0.1 <= t0 <= 1 bounds a, b)
    const bool fx = p->surface == UWIE_SURFACE_SIX && p->inter_dtype == UWIE_INTER_FX32;
    if (!p->gf_exact) UWIE_TRY(launch_guided_fast(P.gray, P.t0, s, p->gf_ksize, p->gf_eps, P.t, &handled, st, fx));
    if (!handled) UWIE_TRY(launch_guided(P.gray, P.t0, s, p->gf_ksize, p->gf_eps, P.t, P.scratch, st));
    return UWIE_OK;
}

int stage_stretch(const Pipe &P, Shape s, double lo, double hi, float eps, hipStream_t st)
{
    const double q[2] = {lo, hi};
    UWIE_TRY(launch_percentiles_f32(P.F, 0, s, q, 2, P.pct, P.scratch, st));
    return launch_stretch_apply_f32(P.F, P.pct, 2, 0, 1, eps, P.F, s, st);
}

// Cast detection (or the forced kind): what every six_stadigy strategy starts from.
// then_airlight: six_airlight follows on the same Pipe -- the chunk pass may collect the level-0 quadrant histograms for it
bool quad_from_cast(const uwie_params *p, const Pipe &P) { return P.qpart && p->forced_cast < 0 && p->cast_correct; }

int six_cast(uwie_ctx *ctx, const uint8_t *d_in, Shape s, const uwie_params *p, const Pipe &P, const int32_t **kind,
             hipStream_t st, bool then_airlight = false)
{
    *kind = nullptr;
    if (p->forced_cast >= 0) {
        UWIE_TRY(launch_set_kind(P.kind, s.B, p->forced_cast, st));
        *kind = P.kind;
    } else if (p->cast_correct) {
        // tuning entry_fuse = 2: histograms only, the gray plane comes out of the quadtree's pre-pass (k_gray_strong)
        const EntryFuse ef{P.qpart, tune().entry_fuse == 2 ? nullptr : P.gray, P.guess, p->gray_shift};
        UWIE_TRY(launch_cast_classify(ctx, d_in, s, P.kind, nullptr, P.scratch, st, then_airlight && quad_from_cast(p, P) ? &ef : nullptr));
        *kind = P.kind;
    }
    return UWIE_OK;
}

// Gray plane and atmospheric light: they depend on the (colour-corrected) frame only, so strategies 1-3 share them.
int six_airlight(uwie_ctx *ctx, const uint8_t *d_in, const int32_t *kind, Shape s, const uwie_params *p, const Pipe &P,
                 hipStream_t st, bool after_cast = false)
{
    const bool fused = after_cast && quad_from_cast(p, P);
    // (gray_shift 0: the plane is there already -- cast detection's chunk pass wrote it)
    return launch_airlight(ctx, d_in, kind, P.gray, s, p->min_size, P.A, nullptr, P.scratch, st,
                           fused && tune().entry_fuse != 2 ? 0 : p->gray_shift, fused ? P.qpart : nullptr);
}

// Strategies 1-3 from (kind, gray, A) on: transmission -> guided filter -> restore -> stretch -> CLAHE / white balance.
int six_dehaze_tail(uwie_ctx *ctx, const uint8_t *d_in, const int32_t *kind, Shape s, const uwie_params *p, const Pipe &P,
                    uint8_t *d_out_u8, float *d_out_f32, hipStream_t st)
{
    const float eps = 1e-6f;  // six_stadigy.py:198,218
    const int k = p->strategy;
    int t_is_f32 = 0;
    // (the three-digit key sweeps of tuning select_generic instantiate the restore for the float64 plane only: no float32 t there)
    const bool want_f32 = p->inter_dtype == UWIE_INTER_F32T && !tune().select_generic && !(s.W & 1);
    if (!p->gf_exact && p->gf_eps > 0.0 && (p->inter_dtype != UWIE_INTER_FX32) && guided_fused_takes(s, p->gf_ksize)) {
        // round 4: the transmission's first half (S6:170-174) is evaluated inside the guided filter: no t0 plane
        UWIE_TRY(launch_guided_fused(P.gray, d_in, kind, P.A, (float)p->omega, 1e-6f, 1, s, p->gf_ksize, p->gf_eps, P.t, st, want_f32));
        t_is_f32 = want_f32;
    } else {
        UWIE_TRY(launch_trans_init(d_in, kind, P.A, s, (float)p->omega, 1e-6f, 1, P.t0, st));
        UWIE_TRY(stage_guided(ctx, P, s, p, st, tune().select_generic ? nullptr : &t_is_f32));
    }
    // fused tail: restore writes the planar image into P.F and feeds the selection's first histogram sweep
    SelectPlan plan;
    const double q[4] = {p->L_low, p->L_high, p->wb_percentile, 100 - p->wb_percentile};
    // the restored image is clipped to [0, 1]: linear first digit, one collecting sweep (k_select.hip, select_lin_*);
    // tuning select_generic keeps the three-digit key sweeps
    const RestoreSrc src{d_in, kind, P.A, P.t, t_is_f32};
    bool recompute = false;
    if (tune().select_generic) {
        UWIE_TRY(select_begin(s, q, k == 3 ? 4 : 2, P.scratch, st, &plan));
        UWIE_TRY(launch_restore_planar_hist(d_in, kind, P.A, P.t, s, P.F, plan.ghist, st, false, nullptr, nullptr, t_is_f32));
        UWIE_TRY(select_run(plan, P.F, 1, s, true, st));
    } else {
        // Strategies 1 and 2 never store the restored image: the histogram sweep (which also files the elements of the
        // predicted target bins, select_lin_begin) and the stretch each recompute it from the frame and t (restore.h:
        // 11 + 14 bytes per pixel instead of 23 + 15, and no collecting sweep).  P.F is only written for images whose
        // selection falls back to the generic sweeps.  Tuning restore_store keeps the stored planes (4K x 64: 17.9 vs
        // 17.5 ms per step, round 2); the tests compare both modes.
        recompute = k != 3 && !tune().restore_store;
        // Strategy 3 keeps the planes (its tail reads them), so the exact target bins can be collected from them by one
        // streaming sweep: no predicted windows there -- its percentiles (20 / 85 / 2 / 98) sit where the bins are full and
        // four windows made the restore sweep 2.9 ms at 4K x 16 against 0.65 + 0.35 for sweep + collection.
        // Tuning lin_predict3 brings the windows back.
        const bool predict = k != 3 || tune().lin_predict3;
        UWIE_TRY(select_lin_begin(s, q, k == 3 ? 4 : 2, P.scratch, st, &plan, predict ? &src : nullptr));
        // (small jobs keep the histogram sweep: at 1080p x 1 the rank-counting kernels' fixed costs -- wavefront-private queues,
        // a window-wide list for the finish -- make them 54 + 20 us against 34 + 14; tuning rank_sweep = 2 forces them anyway)
        const bool big = (size_t)s.B * s.npx() >= ((size_t)1 << 24) || tune().rank_sweep >= 2;
        if (recompute && plan.predicted && !t_is_f32 && tune().rank_sweep && big && s.npx() >= 4) {
            // round 4: no histogram at all -- counts below the predicted windows + the windows' members (k_restore_rank)
            UWIE_TRY(launch_restore_rank(src, s, plan, st));
            UWIE_TRY(select_rank_run(plan, P.F, s, st, src));
        } else {
            UWIE_TRY(launch_restore_planar_hist(d_in, kind, P.A, P.t, s, recompute ? nullptr : P.F, plan.ghist, st, true, nullptr,
                                                &plan, t_is_f32));
            UWIE_TRY(select_lin_run(plan, P.F, s, st, recompute ? &src : nullptr));
        }
    }
    if (k == 3) {
        UWIE_TRY(select_lerp_chain(plan, s, eps, P.pct, st));
        return launch_tail_plain(P.F, P.pct, 4, eps, 1, s, 0, 1.0, d_out_u8, d_out_f32, st);
    }
    UWIE_TRY(select_lerp(plan, s, P.pct, st));
    return launch_tail_clahe(ctx, P.F, P.pct, 2, eps, 0, s, p->clip_limit, p->tiles_x, p->tiles_y, k == 1 ? 1 : 0,
                             p->gamma, d_out_u8, d_out_f32, P.scratch, st, recompute ? &src : nullptr);
}

// Writes the strategy's float image to d_out_f32 and/or its (y*255).astype(u8) image to d_out_u8.
int run_six(uwie_ctx *ctx, const uint8_t *d_in, Shape s, const uwie_params *p, const Pipe &P, uint8_t *d_out_u8,
            float *d_out_f32, hipStream_t st)
{
    const int32_t *kind = nullptr;
    const bool dz = p->strategy >= 1 && p->strategy <= 3;
    UWIE_TRY(six_cast(ctx, d_in, s, p, P, &kind, st, dz));
    if (dz) {
        UWIE_TRY(six_airlight(ctx, d_in, kind, s, p, P, st, true));
        return six_dehaze_tail(ctx, d_in, kind, s, p, P, d_out_u8, d_out_f32, st);
    }
    // strategies 4-6 never leave 8-bit data for long: evaluated as per-image LUT chains (k_codes.hip)
    return launch_code_strategy(ctx, d_in, kind, s, p, d_out_u8, d_out_f32, P.scratch, st);
}

// enhancement_strategies.py apply_strong/medium/light (ES:350-444) on the uncorrected frame
// have_airlight: P.gray and P.A already hold the gray plane and the atmospheric light of these frames (they depend on the
// frame, min_size and gray_shift only: ES:353,379,425 all call estimate_atmospheric_light(img) -- uwie_select_best_u8 shares them)
int run_dict_dehaze(uwie_ctx *ctx, const uint8_t *d_in, Shape s, const uwie_params *p, const Pipe &P, uint8_t *d_out_u8,
                    float *d_out_f32, hipStream_t st, double *d_out_f64 = nullptr, bool have_airlight = false)
{
    if (!have_airlight)
        UWIE_TRY(launch_airlight(ctx, d_in, nullptr, P.gray, s, p->min_size, P.A, nullptr, P.scratch, st, p->gray_shift));
    if (!p->gf_exact && p->gf_eps > 0.0 && guided_fused_takes(s, p->gf_ksize)) {
        UWIE_TRY(launch_guided_fused(P.gray, d_in, nullptr, P.A, (float)p->omega, 1e-10f, 0, s, p->gf_ksize, p->gf_eps, P.t, st, false));
    } else {
        UWIE_TRY(launch_trans_init(d_in, nullptr, P.A, s, (float)p->omega, 1e-10f, 0, P.t0, st));  // ES:221-225
        UWIE_TRY(stage_guided(ctx, P, s, p, st));
    }
    SelectPlan plan;
    const double q[2] = {p->L_low, p->L_high};
    // the recovered image is clipped to [0, 1]: linear first digit, one collecting sweep (select_lin_*64);
    // tuning select_generic keeps the six-digit key sweeps
    const RestoreSrc src{d_in, nullptr, P.A, P.t};
    bool recompute = false;
    if (tune().select_generic) {
        UWIE_TRY(select_begin64(s, q, 2, P.scratch, st, &plan));
        UWIE_TRY(launch_recover64_planar_hist(d_in, P.A, P.t, s, P.F64, plan.ghist, st));
        UWIE_TRY(select_run64(plan, P.F64, 1, s, true, st));
    } else {
        // the float64 image (24 B/px) is not stored either: histogram sweep, collecting sweep and stretch recompute it
        // from the frame and t (11 B/px each); P.F64 only serves images that fall back to the generic sweeps.
        // Tuning restore_store keeps the stored planes.
        recompute = !tune().restore_store;
        UWIE_TRY(select_lin_begin64(s, q, 2, P.scratch, st, &plan, &src));
        UWIE_TRY(launch_recover64_planar_hist(d_in, P.A, P.t, s, recompute ? nullptr : P.F64, plan.ghist, st, true, nullptr,
                                              &plan));
        UWIE_TRY(select_lin_run64(plan, P.F64, s, st, recompute ? &src : nullptr));
    }
    UWIE_TRY(select_lerp64(plan, s, P.pct64, st));
    return launch_tail_plain64(P.F64, P.pct64, s, p->apply_gamma, p->gamma, d_out_u8, d_out_f32, st, recompute ? &src : nullptr,
                               d_out_f64);
}

int check_params(const uwie_params *p)
{
    UWIE_REQUIRE(p != nullptr, "params is NULL");
    UWIE_REQUIRE(p->surface == UWIE_SURFACE_SIX || p->surface == UWIE_SURFACE_DICT, "unknown surface");
    if (p->surface == UWIE_SURFACE_SIX) UWIE_REQUIRE(p->strategy >= 1 && p->strategy <= 6, "unknown strategy");
    else UWIE_REQUIRE(p->strategy >= 0 && p->strategy <= 4, "unknown strategy");
    UWIE_REQUIRE(p->gray_shift == 14 || p->gray_shift == 15, "gray_shift must be 14 or 15");
    UWIE_REQUIRE(p->forced_cast >= -1 && p->forced_cast <= 2, "forced_cast must be -1 or a UWIE_CAST_* kind");
    UWIE_REQUIRE(p->min_size >= 1, "min_size must be >= 1");
    UWIE_REQUIRE(p->tiles_x >= 1 && p->tiles_y >= 1 && p->tiles_x * p->tiles_y <= 4096, "bad CLAHE tile grid");
    if (dehazes(p)) UWIE_REQUIRE(p->gf_ksize >= 1 && p->gf_ksize <= 1024, "guided-filter width out of range");
    // np.percentile raises ValueError("Percentiles must be in the range [0, 100]") for these (also for NaN)
    UWIE_REQUIRE(p->L_low >= 0.0 && p->L_low <= 100.0 && p->L_high >= 0.0 && p->L_high <= 100.0,
                 "L_low / L_high must be percentiles in [0, 100]");
    if (p->surface == UWIE_SURFACE_SIX && (p->strategy >= 3 && p->strategy <= 5))
        UWIE_REQUIRE(p->wb_percentile >= 0.0 && p->wb_percentile <= 100.0, "wb_percentile must be in [0, 100]");
    UWIE_REQUIRE(p->inter_dtype == UWIE_INTER_F64 || p->inter_dtype == UWIE_INTER_FX32 || p->inter_dtype == UWIE_INTER_F32T,
                 "unknown inter_dtype");
    return UWIE_OK;
}


// ---------------------------------------------------------------- general float images (k_float.hip)
// Buffers of one float-image call.  T = the image's dtype (float: either surface; double: dict surface only).
template <class T>
struct FloatPipe {
    int32_t *kind;
    T *A;
    float *pct;
    double *pct64;
    T *xc;          // colour-corrected copy (SIX surface with cast correction)
    uint8_t *q;     // (x * 255).astype(u8)
    uint8_t *gray;
    T *t0;
    double *t;
    float *F, *F2;  // SIX: working images, HWC
    double *F64;    // DICT: recovered image, planar
    void *scratch;
    size_t scratch_bytes;
    uint32_t *qpart;  // level-0 quadrant shares of cast detection's chunks (tuning entry_fuse; inside the scratch, behind what cast
                      // detection and the quadtree use), or nullptr
    int32_t *guess;   // ... and the cast kinds its gray plane was first written for
};

template <class T>
FloatPipe<T> carve_float(Carver &c, Shape s, const uwie_params *p)
{
    FloatPipe<T> P{};
    const size_t n = (size_t)s.B * s.npx();
    const bool six = p->surface == UWIE_SURFACE_SIX, dz = dehazes(p);
    P.kind = c.take<int32_t>(s.B);
    P.A = c.take<T>((size_t)s.B * 3);
    P.pct = c.take<float>((size_t)s.B * 3 * kMaxPct);
    P.pct64 = c.take<double>((size_t)s.B * 3 * kMaxPct);
    if (six) P.xc = c.take<T>(n * 3);
    P.q = c.take<uint8_t>(n * 3);
    if (dz) {
        P.gray = c.take<uint8_t>(n);
        P.t0 = c.take<T>(n);
        P.t = c.take<double>(n);
    }
    if (six) {
        P.F = c.take<float>(n * 3);
        P.F2 = c.take<float>(n * 3);
    } else if (dz) {
        P.F64 = c.take<double>(n * 3);
    }
    const int tx = p->tiles_x > 0 ? p->tiles_x : 8, ty = p->tiles_y > 0 ? p->tiles_y : 8;
    P.scratch_bytes = max5(dz ? float_airlight_ws_bytes(s) : 0, dz ? guided_ws_bytes(s) : 0, select_ws_bytes(s), clahe_ws_bytes(s, tx, ty),
                           codes_ws_bytes(s, tx, ty));
    P.scratch = c.take<char>(P.scratch_bytes);
    return P;
}

// percentile stretch of a float32 HWC image in place (S6:191-199 / :211-219)
static int float_stretch(const FloatPipe<float> &P, float *img, Shape s, double lo, double hi, hipStream_t st)
{
    const double q[2] = {lo, hi};
    UWIE_TRY(launch_percentiles_f32(img, 0, s, q, 2, P.pct, P.scratch, st));
    return launch_stretch_apply_f32(img, P.pct, 2, 0, 1, 1e-6f, img, s, st);
}

// six_stadigy.py strategies 1-6 on a float32 image (S6:230-285); the result image goes to out_f32 and/or out_u8
static int run_float_six(uwie_ctx *ctx, const float *d_img, Shape s, const uwie_params *p, const FloatPipe<float> &P, uint8_t *out_u8,
                         float *out_f32, hipStream_t st)
{
    const size_t n3 = (size_t)s.B * s.npx() * 3;
    const int32_t *kind = nullptr;
    if (p->forced_cast >= 0) {
        UWIE_TRY(launch_set_kind(P.kind, s.B, p->forced_cast, st));
        kind = P.kind;
    } else if (p->cast_correct) {
        UWIE_TRY(launch_float_cast_classify<float>(d_img, s, P.kind, nullptr, st));
        kind = P.kind;
    }
    const int k = p->strategy;
    const bool dz = k <= 3;
    const float *x = d_img;
    UWIE_TRY(launch_float_prepare<float>(d_img, kind, kind ? P.xc : nullptr, dz ? P.q : nullptr, s, st));
    if (kind) x = P.xc;
    float *y = P.F;
    if (dz) {
        UWIE_TRY(launch_rgb2gray_u8(P.q, P.gray, (size_t)s.B * s.npx(), p->gray_shift, st));
        UWIE_TRY(launch_float_airlight<float>(x, P.gray, s, p->min_size, P.A, P.scratch, st));
        UWIE_TRY(launch_float_trans_init<float>(x, P.A, s, p->omega, (double)1e-6f, 1, P.t0, st));
        int handled = 0;
        if (!p->gf_exact) UWIE_TRY(launch_guided_fast(P.gray, P.t0, s, p->gf_ksize, p->gf_eps, P.t, &handled, st, false));
        if (!handled) UWIE_TRY(launch_guided(P.gray, P.t0, s, p->gf_ksize, p->gf_eps, P.t, P.scratch, st));
        UWIE_TRY((launch_float_restore<float, float>(x, P.A, P.t, s, y, 0, st)));
        UWIE_TRY(float_stretch(P, y, s, p->L_low, p->L_high, st));
        if (k == 3) {
            UWIE_TRY(float_stretch(P, y, s, p->wb_percentile, 100 - p->wb_percentile, st));
        } else {
            UWIE_TRY(launch_clahe_f32(ctx, y, P.F2, s, p->clip_limit, p->tiles_x, p->tiles_y, P.scratch, st));
            y = P.F2;
            if (k == 1) UWIE_TRY(launch_gamma_f32(y, y, n3, p->gamma, 1, st));
        }
    } else if (k == 4) {  // S6:262-268  clahe -> stretch -> white_balance -> gamma
        UWIE_TRY(launch_clahe_f32(ctx, x, y, s, p->clip_limit, p->tiles_x, p->tiles_y, P.scratch, st));
        UWIE_TRY(float_stretch(P, y, s, p->L_low, p->L_high, st));
        UWIE_TRY(float_stretch(P, y, s, p->wb_percentile, 100 - p->wb_percentile, st));
        UWIE_TRY(launch_gamma_f32(y, y, n3, p->gamma, 1, st));
    } else {  // S6:270-285  [white_balance ->] stretch -> clahe -> gamma
        UWIE_HIP_CHECK(hipMemcpyAsync(y, x, n3 * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (k == 5) UWIE_TRY(float_stretch(P, y, s, p->wb_percentile, 100 - p->wb_percentile, st));
        UWIE_TRY(float_stretch(P, y, s, p->L_low, p->L_high, st));
        UWIE_TRY(launch_clahe_f32(ctx, y, P.F2, s, p->clip_limit, p->tiles_x, p->tiles_y, P.scratch, st));
        y = P.F2;
        UWIE_TRY(launch_gamma_f32(y, y, n3, p->gamma, 1, st));
    }
    if (out_f32) UWIE_HIP_CHECK(hipMemcpyAsync(out_f32, y, n3 * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (out_u8) UWIE_TRY(launch_quantise_u8(y, out_u8, n3, st));
    return UWIE_OK;
}

// enhancement_strategies.py apply_strategy bodies (ES:350-474) on a float32 or float64 image
template <class T>
static int run_float_dict(uwie_ctx *ctx, const T *d_img, Shape s, const uwie_params *p, const FloatPipe<T> &P, uint8_t *out_u8,
                          float *out_f32, double *out_f64, hipStream_t st)
{
    UWIE_TRY(launch_float_prepare<T>(d_img, nullptr, (T *)nullptr, P.q, s, st));
    if (!dehazes(p))  // clahe_enhancement / histogram_equalization start by quantising: code domain from here on
        return launch_code_strategy(ctx, P.q, nullptr, s, p, out_u8, out_f32, P.scratch, st, out_f64, true);
    UWIE_TRY(launch_rgb2gray_u8(P.q, P.gray, (size_t)s.B * s.npx(), p->gray_shift, st));
    UWIE_TRY(launch_float_airlight<T>(d_img, P.gray, s, p->min_size, P.A, P.scratch, st));
    UWIE_TRY(launch_float_trans_init<T>(d_img, P.A, s, p->omega, 1e-10, 0, P.t0, st));  // ES:221-225: no clip
    if constexpr (sizeof(T) == 4) {
        int handled = 0;
        if (!p->gf_exact) UWIE_TRY(launch_guided_fast(P.gray, P.t0, s, p->gf_ksize, p->gf_eps, P.t, &handled, st, false));
        if (!handled) UWIE_TRY(launch_guided(P.gray, P.t0, s, p->gf_ksize, p->gf_eps, P.t, P.scratch, st));
    } else {
        UWIE_TRY(launch_guided_p64(P.gray, P.t0, s, p->gf_ksize, p->gf_eps, P.t, P.scratch, st));
    }
    UWIE_TRY((launch_float_restore<T, double>(d_img, P.A, P.t, s, P.F64, 1, st)));
    SelectPlan plan;
    const double q[2] = {p->L_low, p->L_high};
    UWIE_TRY(select_begin64(s, q, 2, P.scratch, st, &plan));
    UWIE_TRY(select_run64(plan, P.F64, 1, s, false, st));
    UWIE_TRY(select_lerp64(plan, s, P.pct64, st));
    return launch_tail_plain64(P.F64, P.pct64, s, p->apply_gamma, p->gamma, out_u8, out_f32, st, nullptr, out_f64);
}

template <class T>
static size_t float_ws_bytes(int batch, int H, int W, const uwie_params *p)
{
    Carver c(nullptr);
    carve_float<T>(c, Shape{batch, H, W}, p);
    return c.total();
}

}  // namespace
}  // namespace uwie

using namespace uwie;

extern "C" {

const char *uwie_last_error(void) { return g_err; }
const char *uwie_version(void) { return "uwie 0.1 (gfx950)"; }

int uwie_create(int device, uwie_ctx **out_ctx)
{
    UWIE_REQUIRE(out_ctx != nullptr, "out_ctx is NULL");
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        set_error("no HIP device available (libuwie has no CPU path)");
        return UWIE_E_NODEVICE;
    }
    UWIE_REQUIRE(device >= 0 && device < count, "device index out of range");
    UWIE_HIP_CHECK(hipSetDevice(device));
    uwie_ctx *ctx = new uwie_ctx();
    ctx->aux_ready = false;
    ctx->prof = prof_create();
    ctx->device = device;
    tuning_from_env(&ctx->tune);
    LabTables *lab = new LabTables();
    CastTables *cast = new CastTables();
    build_lab_tables(lab);
    build_cast_tables(cast);
    hipError_t e = hipMalloc((void **)&ctx->d_lab, sizeof(LabTables));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_cast, sizeof(CastTables));
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_status, 64);
    if (e == hipSuccess) e = hipMemset(ctx->d_status, 0, 64);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_lab, lab, sizeof(LabTables), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ctx->d_cast, cast, sizeof(CastTables), hipMemcpyHostToDevice);
    delete lab;
    delete cast;
    if (e != hipSuccess) {
        set_error("context table upload failed: %s", hipGetErrorString(e));
        uwie_destroy(ctx);
        return UWIE_E_HIP;
    }
    *out_ctx = ctx;
    return UWIE_OK;
}

void uwie_destroy(uwie_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->d_lab) (void)hipFree(ctx->d_lab);
    if (ctx->d_cast) (void)hipFree(ctx->d_cast);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->aux_ready) {
        for (int i = 0; i < 4; ++i) {
            (void)hipStreamDestroy(ctx->aux[i]);
            (void)hipEventDestroy(ctx->join[i]);
        }
        (void)hipEventDestroy(ctx->fork);
    }
    prof_bind(nullptr);
    prof_destroy(ctx->prof);
    delete ctx;
}

int uwie_device_status(uwie_ctx *ctx, void *stream, uint32_t *bits)
{
    UWIE_REQUIRE(ctx != nullptr, "device_status: NULL context");
    UWIE_SCOPE(ctx);
    uint32_t v = 0;
    hipStream_t st = (hipStream_t)stream;
    UWIE_HIP_CHECK(hipMemcpyAsync(&v, ctx->d_status, sizeof v, hipMemcpyDeviceToHost, st));
    UWIE_HIP_CHECK(hipStreamSynchronize(st));
    if (v) UWIE_HIP_CHECK(hipMemsetAsync(ctx->d_status, 0, sizeof v, st));
    if (bits) *bits = v;
    if (v) {
        set_error("device status 0x%x:%s the results of the calls since the last check are not valid", v,
                  (v & UWIE_STATUS_CANNY_LABEL) ? " Canny hysteresis met a component label that this launch did not write (k_canny.hip);"
                  : (v & UWIE_STATUS_FALLBACK_SYNC) ? " the percentile fallback's blocks gave up waiting for each other (k_select.hip);"
                  : (v & UWIE_STATUS_QTREE_BOUNDS) ? " a quadtree score fell outside its histogram interval (k_airlight.hip, tuning q_hist = 3);" : "");
        return UWIE_E_DEVICE;
    }
    return UWIE_OK;
}

int uwie_profile_enable(uwie_ctx *ctx, int on)
{
    UWIE_REQUIRE(ctx != nullptr, "profile_enable: NULL context");
    prof_enable(ctx->prof, on != 0);
    prof_bind(on ? ctx->prof : nullptr);
    return UWIE_OK;
}

int uwie_profile_filter(uwie_ctx *ctx, const char *kernel_name)
{
    UWIE_REQUIRE(ctx != nullptr, "ctx is NULL");
    prof_filter(ctx->prof, kernel_name);
    return UWIE_OK;
}

int uwie_profile_collect(uwie_ctx *ctx)
{
    if (!ctx) {
        set_error("profile_collect: NULL context");
        return UWIE_E_INVALID;
    }
    return prof_collect(ctx->prof);
}

int uwie_profile_row(uwie_ctx *ctx, int i, const char **name, double *total_ms, int *calls)
{
    UWIE_REQUIRE(ctx && name && total_ms && calls, "profile_row: NULL pointer");
    return prof_row(ctx->prof, i, name, total_ms, calls);
}

int uwie_set_tuning(uwie_ctx *ctx, const char *name, int value)
{
    UWIE_REQUIRE(ctx && name, "set_tuning: NULL pointer");
    for (const KnobName &k : kKnobs)
        if (std::strcmp(k.name, name) == 0) {
            ctx->tune.*(k.field) = value;
            return UWIE_OK;
        }
    set_error("set_tuning: unknown selector '%s'", name);
    return UWIE_E_INVALID;
}

int uwie_get_tuning(uwie_ctx *ctx, const char *name, int *value)
{
    UWIE_REQUIRE(ctx && name && value, "get_tuning: NULL pointer");
    for (const KnobName &k : kKnobs)
        if (std::strcmp(k.name, name) == 0) {
            *value = ctx->tune.*(k.field);
            return UWIE_OK;
        }
    set_error("get_tuning: unknown selector '%s'", name);
    return UWIE_E_INVALID;
}

int uwie_params_init(uwie_params *p, int surface, int strategy)
{
    UWIE_REQUIRE(p != nullptr, "params is NULL");
    std::memset(p, 0, sizeof *p);
    p->surface = surface;
    p->strategy = strategy;
    p->gray_shift = 15;
    p->min_size = 1;
    p->tiles_x = p->tiles_y = 8;
    p->wb_percentile = -1.0;
    p->gamma = 1.2;
    p->forced_cast = -1;
    if (surface == UWIE_SURFACE_SIX) {
        p->cast_correct = 1;
        switch (strategy) {  // six_stadigy.py:230-285
        case 1: p->omega = 0.3; p->gf_ksize = 20; p->gf_eps = 5e-1; p->L_low = 5; p->L_high = 98; p->clip_limit = 3.0; p->gamma = 1.5; p->apply_gamma = 1; break;
        case 2: p->omega = 0.5; p->gf_ksize = 15; p->gf_eps = 5e-1; p->L_low = 15; p->L_high = 95; p->clip_limit = 2.0; break;
        case 3: p->omega = 0.7; p->gf_ksize = 10; p->gf_eps = 1e-1; p->L_low = 20; p->L_high = 85; p->wb_percentile = 2; break;
        case 4: p->clip_limit = 4.0; p->L_low = 10; p->L_high = 95; p->wb_percentile = 3; p->gamma = 1.3; p->apply_gamma = 1; break;
        case 5: p->wb_percentile = 2; p->L_low = 15; p->L_high = 90; p->clip_limit = 1.5; p->gamma = 1.2; p->apply_gamma = 1; break;
        case 6: p->L_low = 5; p->L_high = 98; p->clip_limit = 3.5; p->gamma = 1.4; p->apply_gamma = 1; break;
        default: set_error("unknown strategy %d", strategy); return UWIE_E_INVALID;
        }
        return UWIE_OK;
    }
    if (surface == UWIE_SURFACE_DICT) {
        p->gf_eps = 0.001;  // estimate_transmission's default; callers never pass it (ES:209,354-358)
        switch (strategy) {  // in-code defaults of ES:350-474
        case UWIE_DICT_STRONG_DEHAZING: p->omega = 0.5; p->gf_ksize = 15; p->L_low = 10; p->L_high = 95; break;
        case UWIE_DICT_MEDIUM_DEHAZING: p->omega = 0.6; p->gf_ksize = 20; p->L_low = 15; p->L_high = 92; break;
        case UWIE_DICT_LIGHT_ENHANCEMENT: p->omega = 0.4; p->gf_ksize = 10; p->L_low = 15; p->L_high = 95; break;
        case UWIE_DICT_CLAHE_ENHANCEMENT: p->clip_limit = 2.0; p->L_low = 20; p->L_high = 85; break;
        case UWIE_DICT_HISTOGRAM_EQUALIZATION: p->L_low = 10; p->L_high = 95; break;
        default: set_error("unknown strategy %d", strategy); return UWIE_E_INVALID;
        }
        return UWIE_OK;
    }
    set_error("unknown surface %d", surface);
    return UWIE_E_INVALID;
}

// How many sub-batches uwie_enhance_u8 runs on separate streams: tuning `streams` = 2..4, default 1.
// Two streams shorten a 4K x 64 step by ~3 % because issue-bound and memory-bound stages of different sub-batches overlap;
// it is opt-in because overlapped kernels no longer have a per-kernel duration that means anything (bench.py's roofline
// line and rocprofv3's averages both read 1.8x for the guided filter).
static int enhance_split(int batch)
{
    const int n = std::min(std::max(tune().streams, 1), 4);
    return batch >= n ? n : 1;
}

size_t uwie_workspace_bytes(int batch, int H, int W, const uwie_params *p)
{
    if (!shape_ok(batch, H, W)) return 0;
    const Shape s{batch, H, W};
    Carver c(nullptr);
    carve_pipe(c, s, p);
    size_t pipe = c.total();
    // uwie_enhance_u8 may run the batch as up to four sub-batches on separate streams (tuning `streams`), each with its own
    // slice: sized for whichever split is the largest, so the answer does not depend on a context
    for (int nsplit = 2; nsplit <= 4 && nsplit <= batch; ++nsplit) {
        size_t off = 0;
        for (int i = 0; i < nsplit; ++i) {
            Carver ci(nullptr);
            carve_pipe(ci, Shape{batch / nsplit + (i < batch % nsplit ? 1 : 0), H, W}, p);
            off = (off + ci.total() + 255) & ~(size_t)255;
        }
        if (off > pipe) pipe = off;
    }
    if (p) return pipe;  // a pipeline call with these parameters
    // p == NULL: any stage entry point; they carve their own (smaller) layouts from the same buffer
    size_t stage = canny_ws_bytes(s) + (size_t)batch * sizeof(Region) + 256;
    const size_t gf = guided_ws_bytes(s);
    if (gf > stage) stage = gf;
    const size_t ft = features_ws_bytes(s);
    if (ft > stage) stage = ft;
    const size_t qa = quality_ws_bytes(s);
    if (qa > stage) stage = qa;
    const size_t cl = clahe_ws_bytes(s, 8, 8);
    if (cl > stage) stage = cl;
    const size_t al = airlight_ws_bytes(s) + (size_t)batch * s.npx() + 256;
    if (al > stage) stage = al;
    return pipe > stage ? pipe : stage;
}

size_t uwie_workspace_bytes_ctx(uwie_ctx *ctx, int batch, int H, int W, const uwie_params *p)
{
    if (!ctx) return uwie_workspace_bytes(batch, H, W, p);
    CallScope scope(ctx);  // tune() answers for this context
    return uwie_workspace_bytes(batch, H, W, p);
}

size_t uwie_workspace_bytes_all(int batch, int H, int W, const uwie_params *p6)
{
    if (!shape_ok(batch, H, W)) return 0;
    uwie_params P6[6];
    for (int k = 0; k < 6; ++k) {
        if (p6) P6[k] = p6[k];
        else if (uwie_params_init(&P6[k], UWIE_SURFACE_SIX, k + 1) != UWIE_OK) return 0;
    }
    const Shape s{batch, H, W};
    const uwie_params Pm = merged_params(P6, 6, s);
    Carver c(nullptr);
    carve_pipe(c, s, &Pm);
    return c.total();
}

int uwie_enhance_u8(uwie_ctx *ctx, const uint8_t *d_in, uint8_t *d_out_u8, float *d_out_f32, int batch, int H, int W,
                    const uwie_params *p, void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_in && (d_out_u8 || d_out_f32), "enhance: NULL context or image pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    UWIE_TRY(check_params(p));
    const Shape s{batch, H, W};
    Carver c(d_workspace);
    Pipe P = carve_pipe(c, s, p);
    UWIE_CHECK_WS(c.total());
    hipStream_t st = (hipStream_t)stream;
    // Frames are independent, so the batch runs as sub-batches on separate streams: while one sits in an issue-bound
    // stage (guided filter) another can be in a memory-bound one.  The caller's stream waits for all of them.
    const int nsplit = enhance_split(batch);
    if (nsplit >= 2 && batch >= nsplit) {
        int cnt[4], start[4];
        Pipe Ps[4];
        size_t off = 0;
        bool fits_ws = true;
        for (int i = 0, b0 = 0; i < nsplit; ++i) {
            cnt[i] = batch / nsplit + (i < batch % nsplit ? 1 : 0);
            start[i] = b0;
            b0 += cnt[i];
            Carver ci(static_cast<char *>(d_workspace) + off);
            Ps[i] = carve_pipe(ci, Shape{cnt[i], H, W}, p);
            if (off + ci.total() > workspace_bytes) fits_ws = false;  // (a caller that sized for another parameter set)
            off = (off + ci.total() + 255) & ~(size_t)255;
        }
        if (fits_ws) {
            if (!ctx->aux_ready) {
                for (int i = 0; i < 4; ++i) {
                    UWIE_HIP_CHECK(hipStreamCreateWithFlags(&ctx->aux[i], hipStreamNonBlocking));
                    UWIE_HIP_CHECK(hipEventCreateWithFlags(&ctx->join[i], hipEventDisableTiming));
                }
                UWIE_HIP_CHECK(hipEventCreateWithFlags(&ctx->fork, hipEventDisableTiming));
                ctx->aux_ready = true;
            }
            UWIE_HIP_CHECK(hipEventRecord(ctx->fork, st));
            const size_t px = (size_t)H * W;
            for (int i = 0; i < nsplit; ++i) {
                UWIE_HIP_CHECK(hipStreamWaitEvent(ctx->aux[i], ctx->fork, 0));
                const size_t b0 = (size_t)start[i];
                const Shape si{cnt[i], H, W};
                const uint8_t *in_i = d_in + b0 * px * 3;
                uint8_t *o8 = d_out_u8 ? d_out_u8 + b0 * px * 3 : nullptr;
                float *of = d_out_f32 ? d_out_f32 + b0 * px * 3 : nullptr;
                int rc;
                if (p->surface == UWIE_SURFACE_SIX) rc = run_six(ctx, in_i, si, p, Ps[i], o8, of, ctx->aux[i]);
                else if (!dehazes(p)) rc = launch_code_strategy(ctx, in_i, nullptr, si, p, o8, of, Ps[i].scratch, ctx->aux[i]);
                else rc = run_dict_dehaze(ctx, in_i, si, p, Ps[i], o8, of, ctx->aux[i]);
                if (rc != UWIE_OK) return rc;
                UWIE_HIP_CHECK(hipEventRecord(ctx->join[i], ctx->aux[i]));
            }
            for (int i = 0; i < nsplit; ++i) UWIE_HIP_CHECK(hipStreamWaitEvent(st, ctx->join[i], 0));
            return UWIE_OK;
        }
    }
    if (p->surface == UWIE_SURFACE_SIX) {
        return run_six(ctx, d_in, s, p, P, d_out_u8, d_out_f32, st);
    }
    if (!dehazes(p)) return launch_code_strategy(ctx, d_in, nullptr, s, p, d_out_u8, d_out_f32, P.scratch, st);
    return run_dict_dehaze(ctx, d_in, s, p, P, d_out_u8, d_out_f32, st);
}

int uwie_enhance_u8_f64(uwie_ctx *ctx, const uint8_t *d_in, uint8_t *d_out_u8, double *d_out_f64, int batch, int H, int W,
                        const uwie_params *p, void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_in && d_out_f64, "enhance_f64: NULL context or image pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    UWIE_TRY(check_params(p));
    UWIE_REQUIRE(p->surface == UWIE_SURFACE_DICT, "enhance_f64: float64 images are the dict surface's (ES:247,307,345)");
    const Shape s{batch, H, W};
    Carver c(d_workspace);
    Pipe P = carve_pipe(c, s, p);
    UWIE_CHECK_WS(c.total());
    hipStream_t st = (hipStream_t)stream;
    if (!dehazes(p)) return launch_code_strategy(ctx, d_in, nullptr, s, p, d_out_u8, nullptr, P.scratch, st, d_out_f64);
    return run_dict_dehaze(ctx, d_in, s, p, P, d_out_u8, nullptr, st, d_out_f64);
}


size_t uwie_workspace_bytes_float(int batch, int H, int W, const uwie_params *p, int elem_bytes)
{
    if (!shape_ok(batch, H, W) || !p || (elem_bytes != 4 && elem_bytes != 8)) return 0;
    return elem_bytes == 4 ? float_ws_bytes<float>(batch, H, W, p) : float_ws_bytes<double>(batch, H, W, p);
}

int uwie_enhance_f32(uwie_ctx *ctx, const float *d_img, uint8_t *d_out_u8, float *d_out_f32, double *d_out_f64, int batch, int H, int W,
                     const uwie_params *p, void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_img && (d_out_u8 || d_out_f32 || d_out_f64), "enhance_f32: NULL context or image pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    UWIE_TRY(check_params(p));
    const Shape s{batch, H, W};
    Carver c(d_workspace);
    FloatPipe<float> P = carve_float<float>(c, s, p);
    UWIE_CHECK_WS(c.total());
    if (p->surface == UWIE_SURFACE_SIX) {
        UWIE_REQUIRE(!d_out_f64, "enhance_f32: the six_stadigy strategies return float32 images");
        return run_float_six(ctx, d_img, s, p, P, d_out_u8, d_out_f32, (hipStream_t)stream);
    }
    return run_float_dict<float>(ctx, d_img, s, p, P, d_out_u8, d_out_f32, d_out_f64, (hipStream_t)stream);
}

int uwie_enhance_f64(uwie_ctx *ctx, const double *d_img, uint8_t *d_out_u8, double *d_out_f64, int batch, int H, int W,
                     const uwie_params *p, void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_img && (d_out_u8 || d_out_f64), "enhance_f64: NULL context or image pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    UWIE_TRY(check_params(p));
    UWIE_REQUIRE(p->surface == UWIE_SURFACE_DICT, "enhance_f64: float64 images are the dict surface's (six_stadigy.py works on float32, S6:406)");
    const Shape s{batch, H, W};
    Carver c(d_workspace);
    FloatPipe<double> P = carve_float<double>(c, s, p);
    UWIE_CHECK_WS(c.total());
    return run_float_dict<double>(ctx, d_img, s, p, P, d_out_u8, nullptr, d_out_f64, (hipStream_t)stream);
}

int uwie_color_correct_f32(uwie_ctx *ctx, const float *d_img, const int32_t *d_kind, float *d_out, int batch, int H, int W, void *stream)
{
    UWIE_REQUIRE(ctx && d_img && d_kind && d_out, "color_correct_f32: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    return launch_float_prepare<float>(d_img, d_kind, d_out, nullptr, Shape{batch, H, W}, (hipStream_t)stream);
}

int uwie_cast_classify_f32(uwie_ctx *ctx, const float *d_img, int batch, int H, int W, int32_t *d_kind, float *d_mean_rgb, void *stream)
{
    UWIE_REQUIRE(ctx && d_img && (d_kind || d_mean_rgb), "cast_classify_f32: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    return launch_float_cast_classify<float>(d_img, Shape{batch, H, W}, d_kind, d_mean_rgb, (hipStream_t)stream);
}

int uwie_enhance_all_u8(uwie_ctx *ctx, const uint8_t *d_in, uint8_t *d_out_u8, int32_t *d_kind, int batch, int H, int W,
                        const uwie_params *p6, void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_in && d_out_u8, "enhance_all: NULL context or image pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    uwie_params P6[6];
    for (int k = 0; k < 6; ++k) {
        if (p6) P6[k] = p6[k];
        else UWIE_TRY(uwie_params_init(&P6[k], UWIE_SURFACE_SIX, k + 1));
        UWIE_TRY(check_params(&P6[k]));
        UWIE_REQUIRE(P6[k].surface == UWIE_SURFACE_SIX && P6[k].strategy == k + 1, "enhance_all: params must be strategies 1..6 in order");
        UWIE_REQUIRE(P6[k].cast_correct == P6[0].cast_correct && P6[k].forced_cast == P6[0].forced_cast &&
                         P6[k].gray_shift == P6[0].gray_shift && P6[k].min_size == P6[0].min_size,
                     "enhance_all: the six parameter sets must agree on the shared stages (cast, gray, quadtree)");
    }
    const Shape s{batch, H, W};
    Carver c(d_workspace);
    const uwie_params Pm = merged_params(P6, 6, s);
    Pipe P = carve_pipe(c, s, &Pm);  // a dehazing layout sized for the most demanding of the six sets
    UWIE_CHECK_WS(c.total());
    hipStream_t st = (hipStream_t)stream;
    const int32_t *kind = nullptr;
    UWIE_TRY(six_cast(ctx, d_in, s, &P6[0], P, &kind, st, true));
    if (d_kind) {
        if (kind) UWIE_HIP_CHECK(hipMemcpyAsync(d_kind, kind, sizeof(int32_t) * batch, hipMemcpyDeviceToDevice, st));
        else UWIE_HIP_CHECK(hipMemsetAsync(d_kind, 0, sizeof(int32_t) * batch, st));
    }
    UWIE_TRY(six_airlight(ctx, d_in, kind, s, &P6[0], P, st, true));
    const size_t out_stride = (size_t)batch * H * W * 3;
    for (int k = 0; k < 6; ++k) {
        uint8_t *out = d_out_u8 + (size_t)k * out_stride;
        if (k < 3) UWIE_TRY(six_dehaze_tail(ctx, d_in, kind, s, &P6[k], P, out, nullptr, st));
        else UWIE_TRY(launch_code_strategy(ctx, d_in, kind, s, &P6[k], out, nullptr, P.scratch, st));
    }
    return UWIE_OK;
}

// ---- main.py:118-146: every strategy of Config.STRATEGIES on the frame, comprehensive_assessment of each result, the best one
namespace {
struct SelectBufs {
    void *pipe;
    size_t pipe_bytes;
    uint8_t *all;
    float *f32;
    void *qa;
};
SelectBufs carve_select(Carver &c, Shape s, const uwie_params *ps, int n, bool own_all)
{
    SelectBufs b{};
    for (int k = 0; k < n; ++k) {
        Carver ck(nullptr);
        carve_pipe(ck, s, &ps[k]);
        if (ck.total() > b.pipe_bytes) b.pipe_bytes = ck.total();
    }
    b.pipe = c.take<char>(b.pipe_bytes);
    b.all = own_all ? c.take<uint8_t>((size_t)n * s.B * s.npx() * 3) : nullptr;
    b.f32 = c.take<float>((size_t)s.B * s.npx() * 3);
    b.qa = c.take<char>(quality_ws_bytes(s));
    return b;
}
}  // namespace

size_t uwie_workspace_bytes_select(int batch, int H, int W, const uwie_params *ps, int n, int with_outputs)
{
    if (!shape_ok(batch, H, W) || !ps || n < 1 || n > 16) return 0;
    Carver c(nullptr);
    carve_select(c, Shape{batch, H, W}, ps, n, !with_outputs);
    return c.total();
}

int uwie_select_best_u8(uwie_ctx *ctx, const uint8_t *d_in, int batch, int H, int W, const uwie_params *ps, int n,
                        const double *weights8, uint8_t *d_best_u8, int32_t *d_best, double *d_scores, uint8_t *d_all_u8,
                        void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_in && ps && d_best && d_scores, "select_best: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    UWIE_REQUIRE(n >= 1 && n <= 16, "select_best: 1 .. 16 strategies");
    for (int k = 0; k < n; ++k) UWIE_TRY(check_params(&ps[k]));
    static const double kDefault[8] = {0.20, 0.20, 0.15, 0.15, 0.10, 0.10, 0.05, 0.05};  // quality_assessment.py:229-238
    const Shape s{batch, H, W};
    Carver c(d_workspace);
    SelectBufs sb = carve_select(c, s, ps, n, d_all_u8 == nullptr);
    UWIE_CHECK_WS(c.total());
    hipStream_t st = (hipStream_t)stream;
    uint8_t *all = d_all_u8 ? d_all_u8 : sb.all;
    const size_t frame_set = (size_t)batch * H * W * 3;
    int shared_with = -1;  // the dict dehazing set whose gray plane / atmospheric light are still in the workspace
    for (int k = 0; k < n; ++k) {
        const uwie_params *p = &ps[k];
        Carver ck(sb.pipe);
        Pipe P = carve_pipe(ck, s, p);
        uint8_t *out = all + (size_t)k * frame_set;
        if (p->surface == UWIE_SURFACE_SIX) {
            UWIE_TRY(run_six(ctx, d_in, s, p, P, out, sb.f32, st));
            shared_with = -1;
        } else if (!dehazes(p)) {
            UWIE_TRY(launch_code_strategy(ctx, d_in, nullptr, s, p, out, sb.f32, P.scratch, st));
            shared_with = -1;
        } else {
            // one quadtree for all the dehazing sets that follow each other with the same leaf size and gray coefficients
            const bool have = shared_with >= 0 && ps[shared_with].min_size == p->min_size && ps[shared_with].gray_shift == p->gray_shift &&
                              ps[shared_with].gf_exact == p->gf_exact;
            UWIE_TRY(run_dict_dehaze(ctx, d_in, s, p, P, out, sb.f32, st, nullptr, have));
            if (!have) shared_with = k;
        }
        // (img * 255).astype(uint8) is what every score but colourfulness starts from; that one takes the float image
        UWIE_TRY(launch_quality_scores(ctx, out, sb.f32, s, p->gray_shift, weights8 ? weights8 : kDefault,
                                       d_scores + (size_t)k * batch * 9, sb.qa, st));
    }
    return launch_pick_best(d_scores, n, s, all, d_best, d_best_u8, st);
}

int uwie_diff_enhance_f32(uwie_ctx *ctx, const float *d_img, float *d_out, int batch, int H, int W, int planar,
                          const float *d_params, int flags, void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_img && d_out && d_params, "diff_enhance: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    UWIE_REQUIRE((flags & ~3) == 0, "diff_enhance: flags are UWIE_DIFF_OMEGA | UWIE_DIFF_GAMMA");
    const Shape s{batch, H, W};
    UWIE_CHECK_WS(select_ws_bytes(s));
    hipStream_t st = (hipStream_t)stream;
    SelectPlan plan;
    UWIE_TRY(select_begin_stretch_ranks(s, d_params, 4, d_workspace, st, &plan));
    UWIE_TRY(select_run(plan, d_img, planar ? 1 : 0, s, false, st));
    return launch_diff_enhance(d_img, planar ? 1 : 0, s, d_params, flags, (const float *)plan.os, d_out, st);
}

int uwie_extract_features_u8(uwie_ctx *ctx, const uint8_t *d_in, float *d_features, int batch, int H, int W,
                             void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_in && d_features, "extract_features: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    const Shape s{batch, H, W};
    UWIE_CHECK_WS(features_ws_bytes(s));
    return launch_features_u8(d_in, s, d_features, d_workspace, (hipStream_t)stream);
}

int uwie_quality_scores(uwie_ctx *ctx, const uint8_t *d_u8, const float *d_f32, int batch, int H, int W, int gray_shift,
                        const double *weights8, double *d_scores, void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_u8 && d_scores, "quality_scores: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    UWIE_REQUIRE(gray_shift == 14 || gray_shift == 15, "gray_shift must be 14 or 15");
    static const double kDefault[8] = {0.20, 0.20, 0.15, 0.15, 0.10, 0.10, 0.05, 0.05};  // quality_assessment.py:229-238
    const Shape s{batch, H, W};
    UWIE_CHECK_WS(quality_ws_bytes(s));
    return launch_quality_scores(ctx, d_u8, d_f32, s, gray_shift, weights8 ? weights8 : kDefault, d_scores, d_workspace,
                                 (hipStream_t)stream);
}

/* ---------------------------------------------------------------- stage entry points */

int uwie_cast_classify(uwie_ctx *ctx, const uint8_t *d_in, int batch, int H, int W, int32_t *d_kind, float *d_mean_rgb,
                       void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_in && (d_kind || d_mean_rgb), "cast_classify: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    const Shape s{batch, H, W};
    UWIE_CHECK_WS(cast_ws_bytes(s));
    return launch_cast_classify(ctx, d_in, s, d_kind, d_mean_rgb, d_workspace, (hipStream_t)stream);
}

int uwie_normalise_correct(uwie_ctx *ctx, const uint8_t *d_in, const int32_t *d_kind, float *d_out_f32, int batch, int H,
                           int W, void *stream)
{
    UWIE_REQUIRE(ctx && d_in && d_out_f32, "normalise_correct: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    return launch_normalise_correct(d_in, d_kind, d_out_f32, Shape{batch, H, W}, (hipStream_t)stream);
}

int uwie_atmospheric_light(uwie_ctx *ctx, const uint8_t *d_in, const int32_t *d_kind, int batch, int H, int W,
                           const uwie_params *p, float *d_A, void *d_trace, void *d_workspace, size_t workspace_bytes,
                           void *stream)
{
    UWIE_REQUIRE(ctx && d_in && d_A && p, "atmospheric_light: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    UWIE_REQUIRE(p->min_size >= 1 && (p->gray_shift == 14 || p->gray_shift == 15), "atmospheric_light: bad params");
    const Shape s{batch, H, W};
    Carver c(d_workspace);
    uint8_t *gray = c.take<uint8_t>((size_t)batch * s.npx());
    void *ws = c.take<char>(airlight_ws_bytes(s));
    UWIE_CHECK_WS(c.total());
    hipStream_t st = (hipStream_t)stream;
    return launch_airlight(ctx, d_in, d_kind, gray, s, p->min_size, d_A, d_trace, ws, st, p->gray_shift);
}

int uwie_transmission_init(uwie_ctx *ctx, const uint8_t *d_in, const int32_t *d_kind, const float *d_A, int batch, int H,
                           int W, const uwie_params *p, float *d_t0, uint8_t *d_gray, void *stream)
{
    UWIE_REQUIRE(ctx && d_in && d_A && p && (d_t0 || d_gray), "transmission_init: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    const Shape s{batch, H, W};
    hipStream_t st = (hipStream_t)stream;
    const bool six = p->surface == UWIE_SURFACE_SIX;
    if (d_t0) UWIE_TRY(launch_trans_init(d_in, d_kind, d_A, s, (float)p->omega, six ? 1e-6f : 1e-10f, six ? 1 : 0, d_t0, st));
    if (d_gray) UWIE_TRY(launch_quant_gray(d_in, d_kind, d_gray, s, p->gray_shift, st));
    return UWIE_OK;
}

int uwie_box_filter_f64(uwie_ctx *ctx, const double *d_src, double *d_dst, int batch, int H, int W, int ksize,
                        void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_src && d_dst, "box_filter: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    UWIE_REQUIRE(ksize >= 1 && ksize <= 1024, "box_filter: ksize out of range");
    const Shape s{batch, H, W};
    UWIE_CHECK_WS(box_ws_bytes(s));
    return launch_box_filter_f64(d_src, d_dst, s, ksize, d_workspace, (hipStream_t)stream);
}

int uwie_guided_filter(uwie_ctx *ctx, const uint8_t *d_gray, const float *d_t0, int batch, int H, int W, int ksize,
                       double eps, int exact, double *d_t, void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_gray && d_t0 && d_t, "guided_filter: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    UWIE_REQUIRE(ksize >= 1 && ksize <= 1024, "guided_filter: ksize out of range");
    const Shape s{batch, H, W};
    UWIE_CHECK_WS(guided_ws_bytes(s));
    UWIE_REQUIRE(exact >= 0 && exact <= 2, "guided_filter: mode is 0 (fused float64), 1 (exact order) or 2 (fixed-point ring)");
    int handled = 0;
    if (exact != 1) UWIE_TRY(launch_guided_fast(d_gray, d_t0, s, ksize, eps, d_t, &handled, (hipStream_t)stream, exact == 2));
    if (handled) return UWIE_OK;
    return launch_guided(d_gray, d_t0, s, ksize, eps, d_t, d_workspace, (hipStream_t)stream);
}

int uwie_guided_plan(int batch, int H, int W, int ksize, int *split_row0, int *split_rows)
{
    UWIE_REQUIRE(split_row0 && split_rows, "guided_plan: NULL pointer");
    UWIE_CHECK_SHAPE(batch, H, W);
    int iy0 = 0, band = 0, nb = 0, rows = 0;
    const bool split = guided_split_plan(Shape{batch, H, W}, ksize, &iy0, &band, &nb, &rows);
    *split_row0 = split ? iy0 : 0;
    *split_rows = split ? rows : 0;
    return UWIE_OK;
}

int uwie_restore(uwie_ctx *ctx, const uint8_t *d_in, const int32_t *d_kind, const float *d_A, const double *d_t,
                 int batch, int H, int W, float *d_out_f32, void *stream)
{
    UWIE_REQUIRE(ctx && d_in && d_A && d_t && d_out_f32, "restore: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    return launch_restore(d_in, d_kind, d_A, d_t, Shape{batch, H, W}, d_out_f32, (hipStream_t)stream);
}

int uwie_percentiles_f32(uwie_ctx *ctx, const float *d_img, int batch, int H, int W, const double *q_percent, int nq,
                         float *d_out, void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_img && q_percent && d_out, "percentiles: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    const Shape s{batch, H, W};
    UWIE_CHECK_WS(select_ws_bytes(s));
    return launch_percentiles_f32(d_img, 0, s, q_percent, nq, d_out, d_workspace, (hipStream_t)stream);
}

int uwie_stretch_f32(uwie_ctx *ctx, const float *d_img, float *d_out, int batch, int H, int W, double lo_percent,
                     double hi_percent, void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_img && d_out, "stretch: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    const Shape s{batch, H, W};
    Carver c(d_workspace);
    float *pct = c.take<float>((size_t)batch * 3 * 2);
    void *ws = c.take<char>(select_ws_bytes(s));
    UWIE_CHECK_WS(c.total());
    const double q[2] = {lo_percent, hi_percent};
    hipStream_t st = (hipStream_t)stream;
    UWIE_TRY(launch_percentiles_f32(d_img, 0, s, q, 2, pct, ws, st));
    return launch_stretch_apply_f32(d_img, pct, 2, 0, 1, 1e-6f, d_out, s, st);
}

int uwie_gamma_f32(uwie_ctx *ctx, const float *d_img, float *d_out, size_t n, double g, int mode, void *stream)
{
    UWIE_REQUIRE(ctx && d_img && d_out, "gamma: NULL pointer");
    UWIE_SCOPE(ctx);
    return launch_gamma_f32(d_img, d_out, n, g, mode, (hipStream_t)stream);
}

int uwie_clahe_f32(uwie_ctx *ctx, const float *d_img, float *d_out, int batch, int H, int W, double clip_limit,
                   int tiles_x, int tiles_y, void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_img && d_out, "clahe: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    UWIE_REQUIRE(tiles_x >= 1 && tiles_y >= 1 && tiles_x * tiles_y <= 4096, "clahe: bad tile grid");
    const Shape s{batch, H, W};
    UWIE_CHECK_WS(clahe_ws_bytes(s, tiles_x, tiles_y));
    return launch_clahe_f32(ctx, d_img, d_out, s, clip_limit, tiles_x, tiles_y, d_workspace, (hipStream_t)stream);
}

int uwie_rgb2gray_u8(uwie_ctx *ctx, const uint8_t *d_rgb, uint8_t *d_gray, size_t npixels, int gray_shift, void *stream)
{
    UWIE_REQUIRE(ctx && d_rgb && d_gray, "rgb2gray: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_REQUIRE(gray_shift == 14 || gray_shift == 15, "gray_shift must be 14 or 15");
    return launch_rgb2gray_u8(d_rgb, d_gray, npixels, gray_shift, (hipStream_t)stream);
}

int uwie_rgb2lab_u8(uwie_ctx *ctx, const uint8_t *d_rgb, uint8_t *d_lab, size_t npixels, void *stream)
{
    UWIE_REQUIRE(ctx && d_rgb && d_lab, "rgb2lab: NULL pointer");
    UWIE_SCOPE(ctx);
    return launch_rgb2lab_u8(ctx, d_rgb, d_lab, npixels, (hipStream_t)stream);
}

int uwie_lab2rgb_u8(uwie_ctx *ctx, const uint8_t *d_lab, uint8_t *d_rgb, size_t npixels, void *stream)
{
    UWIE_REQUIRE(ctx && d_lab && d_rgb, "lab2rgb: NULL pointer");
    UWIE_SCOPE(ctx);
    return launch_lab2rgb_u8(ctx, d_lab, d_rgb, npixels, (hipStream_t)stream);
}

int uwie_clahe_u8(uwie_ctx *ctx, const uint8_t *d_plane, uint8_t *d_out, int batch, int H, int W, double clip_limit,
                  int tiles_x, int tiles_y, void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_plane && d_out, "clahe_u8: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    UWIE_REQUIRE(tiles_x >= 1 && tiles_y >= 1 && tiles_x * tiles_y <= 4096, "clahe: bad tile grid");
    const Shape s{batch, H, W};
    UWIE_CHECK_WS(clahe_ws_bytes(s, tiles_x, tiles_y));
    return launch_clahe_plane_u8(d_plane, d_out, s, clip_limit, tiles_x, tiles_y, d_workspace, (hipStream_t)stream);
}

int uwie_canny_u8(uwie_ctx *ctx, const uint8_t *d_gray, uint8_t *d_edges, int batch, int H, int W, int low, int high,
                  void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_gray && d_edges, "canny: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    if (low > high) { const int t = low; low = high; high = t; }
    const Shape s{batch, H, W};
    Carver c(d_workspace);
    Region *regs = c.take<Region>(batch);
    void *ws = c.take<char>(canny_ws_bytes(s));
    UWIE_CHECK_WS(c.total());
    hipStream_t st = (hipStream_t)stream;
    UWIE_TRY(launch_make_full_regions(regs, s, st));
    return launch_canny(d_gray, s, regs, batch, H, W, low, high, nullptr, d_edges, ws, st);
}

int uwie_equalize_hist_u8(uwie_ctx *ctx, const uint8_t *d_plane, uint8_t *d_out, int batch, int H, int W,
                          void *d_workspace, size_t workspace_bytes, void *stream)
{
    UWIE_REQUIRE(ctx && d_plane && d_out, "equalize_hist: NULL pointer");
    UWIE_SCOPE(ctx);
    UWIE_CHECK_SHAPE(batch, H, W);
    const Shape s{batch, H, W};
    UWIE_CHECK_WS((size_t)batch * 256 + 256);
    return launch_equalize_hist_u8(d_plane, d_out, s, d_workspace, (hipStream_t)stream);
}

}  // extern "C"
