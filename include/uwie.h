/*
 * uwie.h -- C ABI of libuwie.so: the MI355X (gfx950) implementation of the
 * underwater-image-enhancement hot path.
 *
 * The reference (submarine0418/underwater_image_enhancement) is pure Python and
 * has NO FFI/plugin interface for this path; the entry points below are what a
 * ctypes binding for it would call.  Each one names the reference interface it
 * replaces (S6 = six_stadigy.py, ES = enhancement_strategies.py).
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer (HBM); images are HWC,
 *     RGB-interleaved, contiguous, batch-major: [batch][H][W][3];
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*) and
 *     returns without synchronising; scalars that the reference reads back on
 *     the host (cast kind, atmospheric light, percentiles) stay on the device;
 *   - the library never allocates or frees caller memory: scratch comes from
 *     the caller (`d_workspace`, sized by uwie_workspace_bytes);
 *   - return value: 0 on success, a negative UWIE_E_* code on failure; the
 *     message is available from uwie_last_error() (thread-local);
 *   - a context belongs to one device and one host thread/stream at a time.
 */
#ifndef UWIE_H_
#define UWIE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UWIE_OK 0
#define UWIE_E_INVALID (-1)   /* bad argument (null pointer, size, unknown strategy: ES:500-501 ValueError) */
#define UWIE_E_WORKSPACE (-2) /* workspace too small */
#define UWIE_E_HIP (-3)       /* a HIP runtime call failed */
#define UWIE_E_NODEVICE (-4)  /* no gfx950 device / code object cannot load */
#define UWIE_E_DEVICE (-5)    /* a kernel found one of its own invariants violated (uwie_device_status) */

/* which reference surface's arithmetic to follow */
#define UWIE_SURFACE_SIX 0  /* six_stadigy.EnhancementStrategies.strategy1..6 (S6:230-285)        */
#define UWIE_SURFACE_DICT 1 /* enhancement_strategies.EnhancementStrategies.apply_strategy (ES:477) */

/* strategy ids.  SIX: 1..6 as in S6:230-285.  DICT: the keys of ES:489-498. */
#define UWIE_DICT_STRONG_DEHAZING 0
#define UWIE_DICT_MEDIUM_DEHAZING 1
#define UWIE_DICT_LIGHT_ENHANCEMENT 2
#define UWIE_DICT_CLAHE_ENHANCEMENT 3
#define UWIE_DICT_HISTOGRAM_EQUALIZATION 4

#define UWIE_CAST_NORMAL 0
#define UWIE_CAST_GREENISH 1
#define UWIE_CAST_BLUISH 2

typedef struct uwie_ctx uwie_ctx;

/* Numeric parameters of one strategy.  uwie_params_init fills the reference's
 * hard-coded (SIX, S6:230-285) or in-code default (DICT, ES:356-372,382-395,
 * 405-419,428-441,466-473) values; the host wrapper overrides fields from the
 * caller's params dict (config.py:28-75 value sets). */
typedef struct uwie_params {
    int32_t surface;       /* UWIE_SURFACE_*                                                     */
    int32_t strategy;      /* see above                                                          */
    int32_t cast_correct;  /* 1: detect_image_type + color_correction first (S6:409,413)         */
    int32_t forced_cast;   /* -1: detect (if cast_correct).  UWIE_CAST_*: skip detection and apply
                              color_correction for this kind (caller ran detect_image_type itself)  */
    int32_t gray_shift;    /* RGB2GRAY fixed point: 15 (OpenCV 4.x, default) or 14               */
    int32_t min_size;      /* quadtree leaf size (S6:49, ES:77): 1                               */
    double omega;          /* S6:173 / ES:225 (a Python float there: float32 arithmetic rounds it, float64 keeps it) */
    int32_t gf_ksize;      /* guided-filter box WIDTH `r` (S6:31, ES:31)                         */
    double gf_eps;         /* S6: 0.5/0.5/0.1 (:234,245,255); ES: always 0.001 (:209)            */
    double L_low, L_high;  /* percentile stretch bounds in percent (S6:191, ES:252)              */
    double wb_percentile;  /* S6:211 white_balance percentile; < 0 = stage absent                */
    double clip_limit;     /* CLAHE clip limit (S6:202, ES:288); <= 0 = stage absent             */
    int32_t tiles_x, tiles_y; /* CLAHE tile grid (S6: 8x8 fixed; ES: tile_grid_size)             */
    double gamma;          /* S6:222 exponent g (x**g); ES:276 g (clip(x**(1/g)))                */
    int32_t apply_gamma;   /* 0/1 (ES `apply_gamma`; S6 strategies 1,4,5,6 always 1)             */
    int32_t gf_exact;      /* guided filter: 1 = reproduce cv2.boxFilter's float64 running-sum order bit for
                              bit (6 materialised planes); 0 (default) = fused single-kernel float64 filter,
                              same window/border, free summation order: |t - t_exact| <= 1e-11 (observed 7e-15).
                              What that means for the u8 OUTPUT of gf_exact = 0: the same bytes as gf_exact = 1 except
                              where the exact value of a pixel sits on a truncation boundary ahead of CLAHE, so that the
                              last bit of t decides its byte -- about one pixel in 1e9 bytes (two in a 25 440-case soak,
                              profiles/r03_soak.txt; tests/test_gpu_fuzz.py keeps one such frame) -- and that pixel then
                              differs by up to CLAHE's local slope (<= clip_limit) times gamma's slope: 2 - 3 LSB seen,
                              1 LSB without CLAHE.  gf_exact = 1 has no such pixel; it costs ~6x the filter's time.
                              The DICT surface's float64 image differs in its last bits (<= 1e-11) under gf_exact = 0. */
    int32_t inter_dtype;   /* number format of the guided filter's a/b intermediates (S6:39-43) when gf_exact = 0:
                              UWIE_INTER_F64 (default) float64 like the reference; UWIE_INTER_FX32 32-bit fixed
                              point (BASELINE.json configs[4] "reduced-precision intermediates": a and b are rounded
                              once to 2^-31 / 2^-30, every sum stays exact; |t - t_exact| <= 5e-10; u8 output: about
                              one byte in 1e6-1e7 differs, by 1 LSB before CLAHE and up to CLAHE's local slope
                              after it -- tests/test_gpu_fuzz.py).  SIX surface only (needs the pre-clipped transmission, S6:174); other
                              cases silently keep float64.
                              UWIE_INTER_F32T (round 3; BASELINE.json configs[4] "fp16 intermediates ... stated tolerance"):
                              the refined transmission (S6:180) is stored as float32 and restore_image (S6:183-188) runs in
                              float32 with a reciprocal instead of the float64 division: half the bytes of the t plane, a
                              quarter of the restore arithmetic.  NOT the <= 1 LSB mode -- its own contract against the float64
                              path: >= 99.98 % of the u8 bytes identical, <= 5e-5 of them off by more than 1 LSB, none by more
                              than 10, PSNR >= 80 dB (tests/test_gpu_fuzz.py::test_f32t_transmission_stated_tolerance).  SIX
                              surface, strategies 1-3, windows 10 / 15 / 20 on frames the wavefront kernels take (even W,
                              H >= 4k, W >= 2k); everything else silently keeps float64.                         */
} uwie_params;

#define UWIE_INTER_F64 0
#define UWIE_INTER_FX32 1
#define UWIE_INTER_F32T 2

const char *uwie_last_error(void);
const char *uwie_version(void);

int uwie_create(int device, uwie_ctx **out_ctx);
void uwie_destroy(uwie_ctx *ctx);

/* Device-side self checks (no reference counterpart).  Kernels that chase indices through workspace memory validate every
 * index before dereferencing it; one that is out of range is not followed -- the kernel sets a bit in the context's
 * device status word and carries on with a safe substitute, so a defect in the library is an error code, never a memory
 * access fault.  Entry points do not synchronise, so they cannot report it themselves: uwie_device_status waits for
 * `stream`, returns UWIE_OK when no bit is set and UWIE_E_DEVICE otherwise (*bits, optional, receives the word; the word
 * is cleared).  A caller that reads results back synchronises anyway and calls this right after (the Python wrapper does).
 *   UWIE_STATUS_CANNY_LABEL  cv2.Canny's hysteresis (S6:150; k_canny.hip union / mark / emit / paint) met a component
 *                            label that the current launch did not write. */
#define UWIE_STATUS_CANNY_LABEL 1u
/*   UWIE_STATUS_FALLBACK_SYNC  the blocks of a plane in the percentile selection's one-launch fallback (k_rank_fallback) gave up
 *                              waiting for each other (seconds): the kernel terminated, that plane's percentiles are not valid. */
#define UWIE_STATUS_FALLBACK_SYNC 2u
/*   UWIE_STATUS_QTREE_BOUNDS   (tuning q_hist = 3 only: a checking route) a quadrant's reference-order score (compute_Q, S6:116-157)
 *                              fell outside the interval derived from its byte histogram, or the histogram route's decision
 *                              differs from the reference-order argmax: the rounding bounds of k_q_decide / k_q_tail are too tight. */
#define UWIE_STATUS_QTREE_BOUNDS 4u
int uwie_device_status(uwie_ctx *ctx, void *stream, uint32_t *bits);

/* Per-kernel timing for benchmarks (no reference counterpart; the reference only has a per-image wall clock,
 * S6:393-500).  enable(1) starts recording one HIP-event pair per kernel launch on the launch stream;
 * collect() synchronises the device, folds the intervals by kernel name and returns the number of rows
 * (negative on error); row(i) reads one row.  Recording is per host thread. */
int uwie_profile_enable(uwie_ctx *ctx, int on);
/* Restrict recording to launches of one kernel (name as uwie_profile_row reports it); NULL or "" records all.  An event
 * pair per launch costs ~9 us of stream time, 6 % of a 4K x 64 step with every launch recorded: a benchmark that reports
 * whole-job time records the one kernel it needs. */
int uwie_profile_filter(uwie_ctx *ctx, const char *kernel_name);
int uwie_profile_collect(uwie_ctx *ctx);
int uwie_profile_row(uwie_ctx *ctx, int i, const char **name, double *total_ms, int *calls);

/* Fill `p` with the reference defaults for (surface, strategy). */
int uwie_params_init(uwie_params *p, int surface, int strategy);

/* Route selectors of a context: which of several equivalent routes a stage takes where uwie_params has no say (the parity
 * tests force the fallback routes this way, profiles/ scripts compare them).  Two classes:
 *   SAME BYTES on every setting -- the selection, storage and quadtree routes: select_generic (0), restore_store (0),
 *     lin_predict3 (0), lin_cap (0 = default), lin_no_predict (0), lin_predict_shift (0), rank_sweep (1: strategies 1-2 count
 *     ranks against the predicted windows; 0: the histogram sweep), q_hist (1: quadtree levels decided from byte histograms where
 *     the score intervals allow; 0: NumPy-order kernels only; 2: histograms taken, never used; 3: both, and every reference-order
 *     score is checked against its interval -- UWIE_STATUS_QTREE_BOUNDS), canny_prepass (1),
 *     streams (1; 2 .. 4 = sub-batches on internal streams), gf_fuse (1: the transmission's first half is evaluated inside the
 *     guided filter for window 15, same float32 operations; 0: k_trans_init writes a t0 plane first),
 *     entry_fuse (1, round 4; SIX surface with cast detection, strategies 1-3, frames whose width is a multiple of 8: the
 *     level-0 quadrant histograms of estimate_atmospheric_light (S6:49-157) are counted by detect_image_type's own pass over
 *     the frame (S6:292) and that pass writes the gray plane (S6:149,177) for a cast kind guessed from 2048 pixels -- frames
 *     whose decision differs get their plane again; 2: histograms from that pass, the gray plane from the quadtree's level-0
 *     Canny pre-pass; 0: one pass over the frame per stage, as in rounds 1-3);
 *   SAME TRANSMISSION TO 1e-11, hence the u8 contract of uwie_params.gf_exact = 0 -- which fused guided-filter kernel runs:
 *     gf_pipe (1), gf_split (1), gf_bands (0 = chosen from the job).  They sum the same windows in different orders.
 *     (UWIE_INTER_F32T needs the wavefront kernels: with gf_pipe = 0 it keeps float64.)
 *   canny_fault_inject (0) is for tests/test_gpu_robustness.py only (it breaks an invariant on purpose; see uwie_device_status).
 * An environment variable UWIE_<NAME> sets the initial value; it is read once, in uwie_create -- no entry point reads the
 * environment.  Unknown names are an error. */
int uwie_set_tuning(uwie_ctx *ctx, const char *name, int value);
int uwie_get_tuning(uwie_ctx *ctx, const char *name, int *value);

/* Scratch bytes needed by uwie_enhance_u8 / any stage entry point for this shape, for a context with the DEFAULT route
 * selectors (p = NULL: any call).  uwie_workspace_bytes_ctx answers for a given context: the tuning selectors restore_store and
 * select_generic make the dehazing strategies keep the restored image in float32 planes (12 B/px more). */
size_t uwie_workspace_bytes(int batch, int H, int W, const uwie_params *p);
size_t uwie_workspace_bytes_ctx(uwie_ctx *ctx, int batch, int H, int W, const uwie_params *p);

/*
 * enhance(u8 RGB) -> u8 RGB for a whole batch.
 *   SIX : x = u8/255 (S6:406) -> [detect_image_type, color_correction (S6:409,413)]
 *         -> strategyN (S6:230-285) -> (y*255).astype(uint8) (S6:430)
 *   DICT: x = u8/255 (main.py:108) -> apply_strategy body (ES:350-474) -> (y*255).astype(uint8) (main.py:155)
 * d_out_f32 (optional, may be NULL) receives the float image the reference's
 * strategy function returns ([batch][H][W][3] float32; float64 values of the
 * DICT surface are rounded to float32).
 */
int uwie_enhance_u8(uwie_ctx *ctx, const uint8_t *d_in, uint8_t *d_out_u8, float *d_out_f32, int batch, int H, int W,
                    const uwie_params *p, void *d_workspace, size_t workspace_bytes, void *stream);

/* The dict surface with the reference's own result type: recover_image / clahe_enhancement / histogram_equalization /
 * color_enhancement / gamma_correction return float64 (ES:247,307,345,269-270,284-285), and the caller quantises THAT
 * ((enhanced * 255).astype(np.uint8), main.py:155).  d_out_f64: [batch][H][W][3] float64; d_out_u8 (optional) the device's
 * own quantisation of the same values.  UWIE_SURFACE_DICT only; same workspace as uwie_enhance_u8. */
int uwie_enhance_u8_f64(uwie_ctx *ctx, const uint8_t *d_in, uint8_t *d_out_u8, double *d_out_f64, int batch, int H, int W,
                        const uwie_params *p, void *d_workspace, size_t workspace_bytes, void *stream);

/* The same strategies on GENERAL float images: the reference's functions take "float HxWx3 in [0, 1]" (S6:230-285,
 * ES:477-508) and its own harnesses feed np.random.rand (ES:516, example_usage.py:27,44,112).  uwie_enhance_u8 starts from
 * the u8 frame a float image was made of (the fast path: every kernel exploits that a pixel value is a function of its
 * byte); an image that is NOT u8-derived goes through these entry points: the same arithmetic with the pixel values read
 * from the float image, then the materialised-image stage kernels.  Written for parity, not speed.
 *   uwie_enhance_f32: float32 [batch][H][W][3] in.  SIX surface: strategy1..6(img) [with detect_image_type /
 *     color_correction first when cast_correct or forced_cast say so]; float32 image to d_out_f32, its (y*255).astype(u8)
 *     to d_out_u8.  DICT surface: apply_strategy body; the reference's float64 image to d_out_f64 (and/or float32 / u8).
 *   uwie_enhance_f64: float64 in, DICT surface only (six_stadigy.py works on float32 frames, S6:406).
 *   uwie_workspace_bytes_float: scratch bytes of either (elem_bytes = 4 or 8).
 *   uwie_cast_classify_f32: detect_image_type (S6:292-302) on a general float32 image. */
size_t uwie_workspace_bytes_float(int batch, int H, int W, const uwie_params *p, int elem_bytes);
int uwie_enhance_f32(uwie_ctx *ctx, const float *d_img, uint8_t *d_out_u8, float *d_out_f32, double *d_out_f64, int batch, int H, int W,
                     const uwie_params *p, void *d_workspace, size_t workspace_bytes, void *stream);
int uwie_enhance_f64(uwie_ctx *ctx, const double *d_img, uint8_t *d_out_u8, double *d_out_f64, int batch, int H, int W,
                     const uwie_params *p, void *d_workspace, size_t workspace_bytes, void *stream);
int uwie_cast_classify_f32(uwie_ctx *ctx, const float *d_img, int batch, int H, int W, int32_t *d_kind, float *d_mean_rgb, void *stream);
/* color_correction (S6:305-323) of a general float32 image: d_kind[b] = UWIE_CAST_* per image (NORMAL copies). */
int uwie_color_correct_f32(uwie_ctx *ctx, const float *d_img, const int32_t *d_kind, float *d_out, int batch, int H, int W, void *stream);

/* Scratch bytes uwie_enhance_all_u8 needs for these six parameter sets (NULL = the defaults): one layout sized for the
 * most demanding of them (exact-order guided filter planes, widest CLAHE tile grid). */
size_t uwie_workspace_bytes_all(int batch, int H, int W, const uwie_params *p6);

/*
 * The batch driver's inner loop (six_stadigy.py:398-431): one cast detection / correction per image, then all six
 * strategies on the corrected image.  The gray plane and the atmospheric light depend on the corrected image only, so
 * strategies 1-3 share one quadtree instead of running three (the reference recomputes it, with the same result).
 * d_out_u8 is [6][batch][H][W][3]: plane k holds strategy k+1.  d_kind (optional) receives UWIE_CAST_* per image
 * (the driver's image_type column).  p6: six parameter sets, strategies 1..6 in order, or NULL for the reference defaults.
 * Workspace: uwie_workspace_bytes(batch, H, W, <a strategy 1-3 parameter set>).
 */
int uwie_enhance_all_u8(uwie_ctx *ctx, const uint8_t *d_in, uint8_t *d_out_u8, int32_t *d_kind, int batch, int H, int W,
                        const uwie_params *p6, void *d_workspace, size_t workspace_bytes, void *stream);

/*
 * DifferentiableEnhancement.forward (vgg_16_UIE.py:32-128), the arithmetic behind
 * EnhancementPredictor.enhance_image (use_trained_model.py:83-111): per channel stretch between the sorted
 * positions int(L_low/100*n) and int(L_high/100*n) (torch.sort) -> [UWIE_DIFF_OMEGA] dark-channel dehazing with
 * A = 0.6 -> [UWIE_DIFF_GAMMA] pow(x + 1e-8, gamma) -> clamp(0, 1).  float32 in, float32 out, same layout:
 * planar != 0: [batch][3][H][W] (the module's NCHW), else [batch][H][W][3].
 * d_params: [batch][4] float32 = {L_low, L_high, omega, gamma} (the module's (B,1) parameter tensors).
 * Bit-exact against torch on the CPU except pow (<= 1 float32 ulp).  Workspace: uwie_workspace_bytes.
 */
#define UWIE_DIFF_OMEGA 1
#define UWIE_DIFF_GAMMA 2
int uwie_diff_enhance_f32(uwie_ctx *ctx, const float *d_img, float *d_out, int batch, int H, int W, int planar,
                          const float *d_params, int flags, void *d_workspace, size_t workspace_bytes, void *stream);

/*
 * vgg_16_UIE.extract_all_features (vgg_16_UIE.py:435-466) for uint8 frames: d_features [batch][79] float32 =
 * {mean, std, min, max, median} of each channel of img = u8/255, then mean(img), std(img), mean(img**2), zeros.
 * NumPy float32 arithmetic (pairwise sums over 8192-element buffers) reproduced bit for bit.
 */
int uwie_extract_features_u8(uwie_ctx *ctx, const uint8_t *d_in, float *d_features, int batch, int H, int W,
                             void *d_workspace, size_t workspace_bytes, void *stream);

/*
 * QualityAssessment.comprehensive_assessment (quality_assessment.py:215-286; called on every strategy output by
 * main.py:130-146 to pick the best one).  d_u8: the quantised frame (img*255).astype(uint8), [batch][H][W][3];
 * d_f32 (optional): the float image itself, used by the colourfulness score only (NULL: u8/255 is used).
 * weights8 (host pointer, optional): weights in the order contrast, sharpness, entropy, saturation, brightness,
 * edge_density, colorfulness, naturalness (NULL: the reference defaults).  d_scores: [batch][9] float64 = the eight
 * scores in that order, then the weighted total.  Integer-derived parts are exact; float statistics are evaluated in
 * float64 (NumPy: float32 pairwise): 2e-3 on the 0..100 scores.
 */
int uwie_quality_scores(uwie_ctx *ctx, const uint8_t *d_u8, const float *d_f32, int batch, int H, int W, int gray_shift,
                        const double *weights8, double *d_scores, void *d_workspace, size_t workspace_bytes, void *stream);

/*
 * The labelling loop of main.py:118-146 for a batch: every parameter set ps[0 .. n-1] (Config.STRATEGIES, config.py:28-75:
 * five DICT sets; SIX sets are accepted too) is applied to the frames, QualityAssessment.comprehensive_assessment
 * (quality_assessment.py:215-286) scores each result, and the best one per frame -- the FIRST maximum of the weighted total in
 * the sets' order, like max(strategy_scores, key=strategy_scores.get) (main.py:145) -- is selected, all on the device.
 * Consecutive DICT dehazing sets share one atmospheric-light quadtree (ES:353,379,425 evaluate the same function of the frame).
 *   d_scores [n][batch][9] float64: eight scores + weighted total per (set, frame); weights8 as in uwie_quality_scores;
 *   d_best   [batch] int32: index of the winning set;  d_best_u8 (optional) [batch][H][W][3]: its (y * 255).astype(uint8);
 *   d_all_u8 (optional) [n][batch][H][W][3]: every set's output (NULL: kept in the workspace only).
 * Workspace: uwie_workspace_bytes_select(batch, H, W, ps, n, d_all_u8 != NULL).
 */
size_t uwie_workspace_bytes_select(int batch, int H, int W, const uwie_params *ps, int n, int with_outputs);
int uwie_select_best_u8(uwie_ctx *ctx, const uint8_t *d_in, int batch, int H, int W, const uwie_params *ps, int n,
                        const double *weights8, uint8_t *d_best_u8, int32_t *d_best, double *d_scores, uint8_t *d_all_u8,
                        void *d_workspace, size_t workspace_bytes, void *stream);

/* ---------------- per-stage entry points (parity tests, composition) ---------------- */

/* detect_image_type (S6:292-302): NumPy's sequential float32 channel means and the 3-way kind. */
int uwie_cast_classify(uwie_ctx *ctx, const uint8_t *d_in, int batch, int H, int W, int32_t *d_kind,
                       float *d_mean_rgb, void *d_workspace, size_t workspace_bytes, void *stream);

/* color_correction (S6:305-323) on x = u8/255: float32 [batch][H][W][3].  d_kind NULL = "normal". */
int uwie_normalise_correct(uwie_ctx *ctx, const uint8_t *d_in, const int32_t *d_kind, float *d_out_f32, int batch,
                           int H, int W, void *stream);

/* estimate_atmospheric_light (S6:49-113, ES:77-144): greedy quadtree descent, A = brightest pixel of
 * the leaf.  d_kind NULL = no cast correction.  d_trace (optional) receives per level, per image:
 * {y0,x0,rows,cols} as int32[4] followed by the four float64 scores, i.e. 4 int32 + 4 double = 48 bytes,
 * for at most 32 levels: [batch][32][48 bytes]. */
int uwie_atmospheric_light(uwie_ctx *ctx, const uint8_t *d_in, const int32_t *d_kind, int batch, int H, int W,
                           const uwie_params *p, float *d_A, void *d_trace, void *d_workspace,
                           size_t workspace_bytes, void *stream);

/* first half of estimate_transmission (S6:170-177 / ES:221-228): t0 = 1 - omega*min_c(x/(A+eps)) [clipped
 * on the SIX surface] as float32 [batch][H][W], and the 8-bit gray guide [batch][H][W]. */
int uwie_transmission_init(uwie_ctx *ctx, const uint8_t *d_in, const int32_t *d_kind, const float *d_A, int batch,
                           int H, int W, const uwie_params *p, float *d_t0, uint8_t *d_gray, void *stream);

/* cv2.boxFilter(src, CV_64F, (k,k)) on float64 planes [batch][H][W] (S6:31-43). */
int uwie_box_filter_f64(uwie_ctx *ctx, const double *d_src, double *d_dst, int batch, int H, int W, int ksize,
                        void *d_workspace, size_t workspace_bytes, void *stream);

/* guided_filter(gray/255, t0, r, eps) followed by clip(.,0.1,1) (S6:178-180, ES:229-232): float64 [batch][H][W].
 * `exact`: 0 = fused float64 kernels (uwie_params.gf_exact = 0), 1 = cv2.boxFilter's running-sum order, 2 = fused with
 * the fixed-point a/b ring (uwie_params.inter_dtype = UWIE_INTER_FX32; the caller guarantees 0.1 <= t0 <= 1). */
int uwie_guided_filter(uwie_ctx *ctx, const uint8_t *d_gray, const float *d_t0, int batch, int H, int W, int ksize,
                       double eps, int exact, double *d_t, void *d_workspace, size_t workspace_bytes, void *stream);

/* Which rows of a frame the default float64 guided filter (gf_exact = 0, inter_dtype = F64) hands to its main kernel
 * (k_guided_split: rows [*split_row0, *split_row0 + *split_rows)).  ksize 15: every row of the frame, one launch (the
 * kernel reflects row indices at the top and bottom borders itself).  ksize 10 / 20 (non-symmetric window): the whole
 * ring periods between row ksize and row H - (ksize - 2); the rows above and below go to the general kernel
 * (k_guided_pipe) in a second launch.  0 rows = the general kernel alone (small jobs, odd widths, other windows).
 * For benchmarks that price each kernel by the pixels it covers.  The function has no context: it answers for the DEFAULT
 * tuning (gf_pipe = gf_split = 1, gf_bands = 0); a context with other settings runs a different plan. */
int uwie_guided_plan(int batch, int H, int W, int ksize, int *split_row0, int *split_rows);

/* restore_image (S6:183-188): float32 [batch][H][W][3]. */
int uwie_restore(uwie_ctx *ctx, const uint8_t *d_in, const int32_t *d_kind, const float *d_A, const double *d_t,
                 int batch, int H, int W, float *d_out_f32, void *stream);

/* np.percentile(img[:,:,c], q) for every image, channel and q (NumPy 2.2.6 float32 arithmetic,
 * S6:196-197,216-217): d_out [batch][3][nq] float32.  q in percent. */
int uwie_percentiles_f32(uwie_ctx *ctx, const float *d_img, int batch, int H, int W, const double *q_percent, int nq,
                         float *d_out, void *d_workspace, size_t workspace_bytes, void *stream);

/* enhance_contrast / white_balance (S6:191-199, 211-219): clip((x-lo)/(hi-lo+1e-6),0,1). */
int uwie_stretch_f32(uwie_ctx *ctx, const float *d_img, float *d_out, int batch, int H, int W, double lo_percent,
                     double hi_percent, void *d_workspace, size_t workspace_bytes, void *stream);

/* gamma_correction: mode 1 = x**g (S6:222-224), mode 2 = clip(x**(1/g),0,1) (ES:276-285). */
int uwie_gamma_f32(uwie_ctx *ctx, const float *d_img, float *d_out, size_t n, double g, int mode, void *stream);

/* apply_clahe (S6:202-208): (x*255).astype(u8) -> RGB2LAB -> CLAHE(L) -> LAB2RGB -> /255 float32. */
int uwie_clahe_f32(uwie_ctx *ctx, const float *d_img, float *d_out, int batch, int H, int W, double clip_limit,
                   int tiles_x, int tiles_y, void *d_workspace, size_t workspace_bytes, void *stream);

/* the OpenCV primitives on their own (u8 in, u8 out) */
int uwie_rgb2gray_u8(uwie_ctx *ctx, const uint8_t *d_rgb, uint8_t *d_gray, size_t npixels, int gray_shift,
                     void *stream);
int uwie_rgb2lab_u8(uwie_ctx *ctx, const uint8_t *d_rgb, uint8_t *d_lab, size_t npixels, void *stream);
int uwie_lab2rgb_u8(uwie_ctx *ctx, const uint8_t *d_lab, uint8_t *d_rgb, size_t npixels, void *stream);
int uwie_clahe_u8(uwie_ctx *ctx, const uint8_t *d_plane, uint8_t *d_out, int batch, int H, int W, double clip_limit,
                  int tiles_x, int tiles_y, void *d_workspace, size_t workspace_bytes, void *stream);
/* cv2.Canny(gray, low, high) (S6:150): edge map (0/255) [batch][H][W]. */
int uwie_canny_u8(uwie_ctx *ctx, const uint8_t *d_gray, uint8_t *d_edges, int batch, int H, int W, int low, int high,
                  void *d_workspace, size_t workspace_bytes, void *stream);
/* cv2.equalizeHist per plane (ES:343): [batch][H][W]. */
int uwie_equalize_hist_u8(uwie_ctx *ctx, const uint8_t *d_plane, uint8_t *d_out, int batch, int H, int W,
                          void *d_workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* UWIE_H_ */
