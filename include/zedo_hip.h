/*
 * zedo_hip.h - C ABI of libzedo_hip.so: the MI355X (gfx950) implementation of ZeDO's
 * optimisation-in-the-loop diffusion sampling path.
 *
 * The reference (ipl-uw/ZeDO-Release) is pure Python/PyTorch and has no FFI; its boundary for
 * this path is a set of Python callables.  Each entry point below is the
 * native body of one of those callables and is bound with ctypes by
 * zedo-release_amd/zedo_hip/__init__.py; INTEGRATION.md shows the binding a reference maintainer
 * would add.  Citations are file:line in the reference tree.
 *
 * Conventions
 *  - every pointer named d_* is a DEVICE pointer to fp32 (unless stated), h_* is a HOST pointer;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued on
 *    it and nothing synchronises unless stated;
 *  - functions write only caller-allocated outputs and the caller-provided workspace; the only
 *    library-owned device memory lives inside zedo_weights_t / zedo_schedule_t handles;
 *  - calls on distinct streams may run concurrently, from one host thread or several, provided each call has its own
 *    workspace and outputs (one process per GPU is the intended use; a second device in the same process works: launch
 *    attributes are cached per device; a call runs on the device that is current when it is made, which must be the
 *    device of its pointers and of `stream`).  Handles are read-only to the row-batched entry points and may be shared
 *    by concurrent calls; zedo_weights_set_math is the exception (switch modes between runs, not during them).
 *    Process-wide state exists only outside the data path: the zedo_profile_* diagnostic (one session at a time; thread
 *    safe; samples launches of every stream) and the ZEDO_CHUNK_ROWS environment value, read once.
 *    tests/test_reentrancy_gpu.py runs two host threads x two streams through this contract;
 *  - rows are hypothesis-major: global row g = h*N + n (h = hypothesis, n = pose) - the order in
 *    which the reference's hypothesis loop produces them (run/opt_main.py:166-222).  A call may
 *    hold any contiguous shard of the global rows: local row b is global row row_offset + b, so
 *    pose = (row_offset+b) % N and hypothesis = (row_offset+b) / N (row_offset = 0 on one GPU);
 *  - return value: 0 on success, a negative ZEDO_E_* code or a positive hipError_t otherwise.
 */
#ifndef ZEDO_HIP_H
#define ZEDO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZEDO_ABI_VERSION 5   /* 3: + zedo_reproj_degenerate, zedo_pose_min, zedo_weights_set_math / zedo_weights_get_math;
                              * 4: + zedo_profile_bracket_ms;  5: + zedo_probe_mfma_peak_f16, workspace rows rounded to 64 again */

#define ZEDO_OK 0
#define ZEDO_E_BADARG (-1)      /* NULL pointer, non-positive size, unsupported dimension */
#define ZEDO_E_NOGPU (-2)       /* no gfx950 device visible */
#define ZEDO_E_WORKSPACE (-3)   /* workspace smaller than zedo_workspace_bytes() */

typedef struct zedo_weights zedo_weights_t;     /* packed ScoreModelFC_Adv parameters on device */
typedef struct zedo_schedule zedo_schedule_t;   /* per-step tables for one timestamp vector */

int zedo_abi_version(void);
const char *zedo_error_string(int code);

/* ---- score network parameters -------------------------------------------------------------
 * Replaces ScoreModelFC_Adv.__init__/load_state_dict (lib/algorithms/advanced/model.py:101-152,
 * run/opt_main.py:120-137).  h_params: the 34 parameter tensors of the state dict, fp32,
 * concatenated in state-dict order WITHOUT the float64 `sigmas` buffer:
 *   pre_dense.{weight[H,J3],bias[H]}, pre_dense_t.{weight[H,E],bias[H]}, pre_gnorm.{weight,bias}[H],
 *   shared_time_embed.0.{weight[E,E],bias[E]},
 *   for b in 1..n_blocks, k in 1..2: b{b}_dense{k}.{weight[H,H],bias[H]},
 *       b{b}_dense{k}_t.{weight[H,E],bias[H]}, b{b}_gnorm{k}.{weight,bias}[H],
 *   post_dense.{weight[J3,H],bias[J3]}
 * Supported: hidden H = 1024 (GroupNorm(32): groups of 32 channels), embed E = 512,
 * J3 = n_joints*joint_dim <= 64, n_blocks = 2.  n_floats must equal the exact total.
 * Synchronises `stream` before returning (h_params may be freed by the caller).
 */
int zedo_weights_create(const float *h_params, size_t n_floats, int n_joints, int joint_dim, int hidden,
                        int embed, int n_blocks, void *stream, zedo_weights_t **out);
void zedo_weights_destroy(zedo_weights_t *w);

/* ---- arithmetic of the dense layers (opt-in) ----------------------------------------------------------------------
 * ZEDO_MATH_F32 (default): exact fp32 MFMA (v_mfma_f32_32x32x2_f32), bitwise an fma chain per output.
 * ZEDO_MATH_F16X3: all six dense layers on the fp16 matrix pipe at fp32-level accuracy: every operand travels as two fp16
 *   pieces (a = ah + al + O(2^-24 |a|)), a 16-deep k block costs three fp16 MFMAs (al.bh + ah.bl + ah.bh, fp32
 *   accumulation); W carries a per-layer power-of-two scale undone exactly in the epilogue; activations between the
 *   hidden layers live in the same 4 bytes per element as two fp16 planes.  Per-product error = one fp32 rounding;
 *   measured max |y - y_fp64| of a layer 1.4e-6 (exact-fp32 kernel: 2.3e-6).  Results differ from ZEDO_MATH_F32 in the
 *   last bits, like any two fp32 implementations of the network do; the geometry kernels (reprojection, IPO, metric) and the
 *   fp32 pose state are unchanged.  Activations are stored as unscaled fp16 pieces: the call returns ZEDO_E_BADARG for a
 *   network whose GroupNorm parameters could produce |activation| >= 32768 (bound: sum over the residual path of
 *   max|gamma| sqrt(31) + max|beta|; trained checkpoints: O(10)), for non-finite weights, and for a weight matrix with a non-zero
 *   row whose largest entry is below 2^-8 of the matrix maximum (each matrix carries ONE scale: such a row would lose bits; DESIGN.md
 *   section 3 has the bound).
 *   zedo_weights_set_math builds the split copy of the six weight matrices on first use and
 *   synchronises `stream`; the mode is a property of the handle and applies to every later call that takes it.
 */
#define ZEDO_MATH_F32 0
#define ZEDO_MATH_F16X3 1
int zedo_weights_set_math(zedo_weights_t *w, int mode, void *stream);
int zedo_weights_get_math(const zedo_weights_t *w);

/* ---- per-step tables ----------------------------------------------------------------------
 * For the timestamp vector h_t[S] (torch.linspace(sde.T, eps, S), run/opt_main.py:198) builds on
 * the device, once:
 *   tbias[s][l][:] = W_l_t . SiLU(W_s . posemb(999 t_s) + b_s) + b_l_t + b_l     (model.py:251-281;
 *                    the time branch is identical for every row because vec_t = ones(B)*t,
 *                    advanced/sampling.py:497)
 *   a[s], c[s]     with  x' = a x + c eps_theta(x, 999 t)  == EulerMaruyamaPredictor.update_fn on
 *                    the probability-flow reverse sub-VP SDE with dt = -1/n_sde
 *                    (advanced/sampling.py:185-191, sde_lib.py:93-100,187-198, utils.py:751-777).
 * label_scale: the network is evaluated at labels = h_t * label_scale (fp32 product).  Pass 999 with
 * h_t = SDE times t (utils.py:762), or 1 with h_t = labels when calling the model surface directly
 * (then a, c are computed at t = label/999).
 * Synchronises `stream` before returning.
 */
int zedo_schedule_create(const zedo_weights_t *w, const float *h_t, int S, float label_scale, float beta_min,
                         float beta_max, int n_sde, void *stream, zedo_schedule_t **out);
void zedo_schedule_destroy(zedo_schedule_t *s);
/* debug/parity accessors: copy tables to host (synchronise). tbias: [S][1+2*n_blocks][H]. */
int zedo_schedule_read(const zedo_schedule_t *s, float *h_tbias, float *h_a, float *h_c);

/* Bytes of device workspace needed by the row-batched entry points below for B rows: 8448 bytes per row
 * (rows rounded up to 64) up to 2^20 rows; larger batches are walked in chunks of that many rows, so the
 * workspace never exceeds 8.9 GB (BASELINE configs 3/4: 3.5 M rows per GPU).  Environment (read once):
 * ZEDO_CHUNK_ROWS=<rows> overrides the chunk size (tests). */
size_t zedo_workspace_bytes(int B);

/* ---- reprojection geometry ------------------------------------------------------------------
 * Step-invariant part of gradient_field_gen (simple_zeroshot_opt.py:61-71,99 and the conf clamp
 * :64-66): geom[n][j] = (r_x, r_y, clamp(conf,1e-4,1)^4, 0, rhat_x, rhat_y, rhat_z, 0) with
 * r = Kinv [u v 1]^T / z and rhat = r/|r|.  d_conf may be NULL (weight 1).  d_uv [N,J,2], d_K [N,3,3],
 * d_conf [N,J] -> d_geom [N,J,8].  If d_conf_clamped != NULL the clamped confidences are also
 * written there (the reference clamps the caller's tensor in place).
 */
int zedo_reproj_prepare(const float *d_uv, const float *d_K, const float *d_conf, int N, int J,
                        float *d_geom, float *d_conf_clamped, void *stream);

/* How many of the N poses have a SINGULAR least-squares system for T (simple_zeroshot_opt.py:73-92): with the centred
 * closed form the 3x3 normal matrix is singular exactly when sum_j W_j |r_j - rbar|^2 == 0 - every ray of the pose
 * coincides.  The reference's torch.inverse(AtA) raises there (:89-92); the solve kernels would divide by zero
 * and produce a NaN T.  The quantity depends only on d_geom (not on x or the step), so it is checked ONCE per
 * problem: the host mirror raises like the reference does when a solve is requested and *h_count > 0.
 * Same fp32 arithmetic as the kernels' denominator.  Synchronises `stream`. */
int zedo_reproj_degenerate(const float *d_geom, int N, int J, int *h_count, void *stream);

/* gradient_field_gen (simple_zeroshot_opt.py:46-125, noise_type None):
 *   solve_T != 0 : T = weighted least-squares translation (:73-93, sign fix :93), written to d_T
 *   solve_T == 0 : T = d_T as given (:95-96)
 *   g = ((x+T).r^) r^ - (x+T)  (:99,109) written to d_g.  d_x [B,J,3], d_T [B,3], d_g [B,J,3].
 */
int zedo_reproj_grad(const float *d_x, const float *d_geom, float *d_T, int solve_T, float *d_g, int B,
                     int N, int J, long long row_offset, void *stream);

/* ---- score network / predictor ---------------------------------------------------------------
 * eps = ScoreModelFC_Adv.forward(x, labels = 999 t_step) (model.py:215-298), d_eps [B,J,3]. */
int zedo_score_eps(const zedo_weights_t *w, const zedo_schedule_t *s, int step, const float *d_x,
                   float *d_eps, int B, void *d_workspace, size_t workspace_bytes, void *stream);
/* one pc_sampler call (advanced/sampling.py:450-527): x <- a[step] x + c[step] eps(x). In place. */
int zedo_sde_step(const zedo_weights_t *w, const zedo_schedule_t *s, int step, float *d_x, int B,
                  void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- the fused OIL loop: run/opt_main.py:202-220 -------------------------------------------------
 * for i in [step_begin, step_end): g,T = gradient_field_gen(x, T if i < switch_step else None);
 *                                   x += g;  x = a_i x + c_i eps(x, t_i)
 * State stays on the device; no host round trip.  d_x [B,J,3] in/out, d_T [B,3] in/out,
 * d_geom [N,J,8].  switch_step = S//5 in the reference.
 */
int zedo_oil_run(const zedo_weights_t *w, const zedo_schedule_t *s, float *d_x, const float *d_geom,
                 float *d_T, int step_begin, int step_end, int switch_step, int B, int N, long long row_offset,
                 void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- IPO: run/opt_main.py:177-195 + RotOpt (simple_zeroshot_opt.py:8-31) -----------------------------
 * Per row b=(h,n): T0 = ipo_T * normalise(Kinv [u0 v0 1]); `iters` Adam(lr 0.1) iterations on
 * (rot_vect, rot_vect_<axes>, scale) minimising mean |proj(R x0_h[kl] + T0 clamp(scale)) - uv[kl]|
 * where the mean's divisor is `normaliser` (= N*k*2 of the reference batch; pass the GLOBAL value
 * when rows are sharded).  d_x0 [H,J,3] (centred cluster poses), d_uv [N,J,2], d_K [N,3,3],
 * h_keylist[k] joint indices, axes_mask bit0=x bit1=y bit2=z.
 * Outputs: d_R [B,3,3], d_T [B,3] = T0*clamp(scale), optional d_q [B,4], d_scale [B] (may be NULL).
 * h_keylist is consumed before the call returns (it travels as a kernel argument); nothing synchronises and nothing is
 * copied (Adam's bias-correction terms are constants of the code object): every call, the first one included, is a plain
 * kernel launch on `stream` - legal under stream capture and safe beside fits on other streams or host threads.
 * The ten gradient sums over the key joints are formed in ONE fixed pairing order by both kernels behind this entry
 * point (one row per half-wave for small batches, one lane per row for large ones): a row's result does not depend on B,
 * row_offset, the shard it is in or the kernel that ran it.
 * H = number of hypotheses in d_x0: row_offset + B > H*N is rejected (ZEDO_E_BADARG).
 */
int zedo_ipo_fit(const float *d_x0, const float *d_uv, const float *d_K, const int *h_keylist, int k,
                 int axes_mask, float ipo_T, float min_scale, float max_scale, int iters, double normaliser,
                 float *d_R, float *d_T, float *d_q, float *d_scale, int B, int H, int N, int J, long long row_offset,
                 void *stream);

/* The same fit, resumable (parity instrument: one Adam iteration from a captured optimiser state).
 * d_state [B,15] fp32 = (param[5], exp_avg[5], exp_avg_sq[5]) in the order (rot_vect, rot_vect_x, _y, _z, scale),
 * i.e. torch.optim.Adam's per-parameter state (opt_main.py:183).  it_begin = iterations already applied to
 * d_state; with it_begin == 0 the input content is ignored and the fit starts from RotOpt's initial values
 * (simple_zeroshot_opt.py:11-16).  Runs iterations [it_begin, it_begin + iters) and writes the state back.
 */
int zedo_ipo_fit_resume(const float *d_x0, const float *d_uv, const float *d_K, const int *h_keylist, int k,
                        int axes_mask, float ipo_T, float min_scale, float max_scale, int iters, double normaliser,
                        float *d_R, float *d_T, float *d_q, float *d_scale, float *d_state, int it_begin, int B, int H,
                        int N, int J, long long row_offset, void *stream);

/* x[b] = R[b] . x0[h(b)]   (run/opt_main.py:201).  d_x [B,J,3]; d_x0 [H,J,3]. */
int zedo_rotate_init(const float *d_x0, const float *d_R, float *d_x, int B, int H, int N, int J, long long row_offset,
                     void *stream);

/* ---- hypothesis selection: eval_multi inner loops (lib/dataset/h36m.py:394-417, pw3d.py:302-331) ----
 * err[b] = mean_j || pred[b,j] - gt[n(b),j] ||, after similarity (Procrustes, scaling, reflection
 * 'best': lib/utils/transforms.py:42-127) alignment when procrustes != 0; then per pose n the
 * minimum over the hypotheses present in [0,B) and its hypothesis index.
 * d_pred [B,J,3] fp32 rows (h,n); d_gt [N,J,3] float64, root-centred, metres.
 * d_err [B] float64 (output, required); d_best [N] float64; d_best_h [N] int32 (first minimum, the
 * np.argmin rule).  Poses with no local row get +inf / -1.
 */
int zedo_min_mpjpe(const float *d_pred, const double *d_gt, int B, int N, int J, long long row_offset,
                   int procrustes, double *d_err, double *d_best, int *d_best_h, void *stream);

/* The second half of zedo_min_mpjpe on its own: per pose n the minimum of d_err over the hypotheses present in
 * [0,B) and the first hypothesis index that attains it (np.amin / np.argmin, NaN wins: h36m.py:411-412).  For
 * callers that edit the per-row errors first - eval_multi's `valid_ind` (h36m.py:396-397: hypotheses not listed for
 * a pose are skipped) sets them to +inf - including on a row shard (row_offset). */
int zedo_pose_min(const double *d_err, int B, int N, long long row_offset, double *d_best, int *d_best_h, void *stream);

/* ---- diagnostics: sampled per-kernel timing ---------------------------------------------------------
 * Between zedo_profile_start and zedo_profile_stop every `sample_every`-th launch of each kernel class
 * issued by zedo_oil_run / zedo_sde_step / zedo_score_eps is bracketed by two hipEvents recorded on the
 * launch stream (at most max_samples pairs).  zedo_profile_stop synchronises those events and fills,
 * per class, the summed elapsed milliseconds, the number of sampled launches and the number of launches
 * seen.  Process-wide diagnostic state; not part of the data path.  Arrays have ZEDO_PROF_CLASSES entries.
 */
#define ZEDO_PROF_HIDDEN 0  /* the four 1024x1024 dense layers (+GroupNorm+SiLU[+residual]) */
#define ZEDO_PROF_PRE 1     /* pre_dense (+GroupNorm+SiLU) */
#define ZEDO_PROF_POST 2    /* post_dense + SDE update */
#define ZEDO_PROF_REPROJ 3  /* reprojection correction (stand-alone launch: first iteration of a zedo_oil_run call) */
#define ZEDO_PROF_CLASSES 4
int zedo_profile_start(int sample_every, int max_samples);
/* What this box's matrix pipe sustains right now: `iters` x 16 back-to-back v_mfma_f32_32x32x2_f32 per wave on every
 * SIMD (two waves each) -> TFLOP/s, and the shader clock seen over that run.  Boxes of one pool differ by a few per
 * cent in clock / power state; bench.py reports the dominant kernel against this number next to the datasheet peak.
 * Synchronises `stream`.  Diagnostic, not part of the data path. */
int zedo_probe_mfma_peak(int iters, double *h_tflops, double *h_shader_ghz, void *stream);
/* The same for the fp16 matrix pipe the opt-in split-fp16 mode runs on: `iters` x 16 back-to-back v_mfma_f32_32x32x16_f16 per wave,
 * two waves per SIMD, operands that differ per lane and per instruction, no LDS / memory traffic.  Under a dense fp16 MFMA stream power
 * management grants well below the 2.4 GHz the 2.5 PFLOP/s datasheet peak is quoted at: this is the ceiling attainable on this box. */
int zedo_probe_mfma_peak_f16(int iters, double *h_tflops, double *h_shader_ghz, void *stream);
int zedo_profile_stop(double *h_total_ms, long long *h_samples, long long *h_launches);
/* Shader clock (GHz) the sampled hidden-layer launches of the last profiling session really ran at: shader cycles over
 * 100 MHz wall ticks, taken by workgroup 0 of each sampled launch around its tile.  0 if nothing was sampled. */
double zedo_profile_shader_ghz(void);
/* What the sampling bracket itself measures (milliseconds): the median of 15 EMPTY event pairs recorded by the last
 * zedo_profile_stop on the stream its samples came from.  Every sampled duration contains it once; the sums returned by
 * zedo_profile_stop are raw - subtract samples x this value for the kernels' own time (bench.py does, and checks that
 * the per-class times of one OIL iteration then add up to no more than the wall time of an iteration).  0 if nothing
 * was sampled. */
double zedo_profile_bracket_ms(void);

#ifdef __cplusplus
}
#endif
#endif /* ZEDO_HIP_H */
