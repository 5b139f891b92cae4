/*
 * cindm_hip.h -- C ABI of libcindm_hip.so: MI355X (gfx950) implementation of CinDM's
 * compositional diffusion SAMPLING hot path.
 *
 * The upstream reference (AI4Science-WestlakeU/cindm) has no native/FFI layer: its callers
 * construct Python classes and call methods.  Each entry point below therefore cites the
 * reference *Python* interface (file:line relative to the reference root) whose arithmetic
 * it replaces; the Python face in cindm_amd/ keeps those class/method signatures and binds
 * to this library through ctypes (see INTEGRATION.md for the binding stub).
 *
 * Conventions
 *   - every function returns int: 0 = OK, <0 = error; text via cindm_last_error() (thread-local)
 *   - no C++ types / exceptions cross the boundary; no torch types in any signature
 *   - the CALLER owns every tensor buffer and the workspace and passes raw DEVICE pointers plus
 *     explicit sizes; the library owns only opaque handles and its packed copies of the weights
 *   - all launches are asynchronous on the hipStream_t passed in (as void*); nothing
 *     synchronises the device except *_finalize
 *   - a handle is bound to the device current at *_create and is not thread-safe
 *   - all tensors are fp32, contiguous, layout [batch, time, feature] (the reference's API
 *     layout, model/diffusion_1d.py:610-614), which is also the channel-last layout the kernels
 *     use internally
 */
#ifndef CINDM_HIP_H
#define CINDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an entry point's argument list changes or an entry point is added.  History: 1 = rounds 1-2; 2 = round 3's
 * positional sum_boundary argument of cindm_airfoil_design_grad / cindm_ddpm2d_sample_force, and round 4's additions
 * (cindm_unet1d_poll, cindm_ddpm2d_predict, cindm_unet1d_phase_prof_*, option "no_exchange", cindm_unet1d_recovered).
 * A caller compiled against another version must not bind: compare cindm_abi_version() with this constant. */
#define CINDM_ABI_VERSION 4

typedef struct cindm_unet1d cindm_unet1d;
typedef struct cindm_ddpm1d cindm_ddpm1d;

/* ------------------------------------------------------------------ misc */
int cindm_abi_version(void);
const char* cindm_last_error(void);
/* sha256 (hex) of the sources this library was compiled from (everything under cindm_amd/csrc plus this header), embedded by
 * cindm_amd/build.py: the Python face refuses a library whose hash differs from the sources next to it. */
const char* cindm_source_hash(void);

/* In-kernel phase clocks of the persistent 2-D convolution kernel (cindm_amd/csrc/kernels2d_v2.h), profiling build only
 * (libcindm_hip_prof.so): copies [8 categories][2 roles][8 phases] sums of 10 ns ticks (128 words) to dst and clears them.
 * Returns 1 in the profiling build, 0 in the production build (zeros), -1 on error.  No reference counterpart (a tool). */
int cindm_ws_prof_read(unsigned long long* dst);

/* ------------------------------------------------------------------ TemporalUnet1D
 * Replaces TemporalUnet1D.__init__/forward, model/diffusion_1d.py:517-646, and the blocks it
 * is built from: SinusoidalPosEmb :146, Conv1dBlock :197, ResidualTemporalBlock :483,
 * Residual/PreNorm/LayerNorm/LinearAttentionTemporal :75/:134/:123/:272, Downsample1d :92,
 * Upsample1d :100. */
typedef struct {
    int32_t horizon;          /* TemporalUnet1D(horizon=...)           :521 */
    int32_t transition_dim;   /* n_bodies * 4 features                 :522 */
    int32_t dim;              /* base width, 64                        :524 */
    int32_t n_mults;          /* len(dim_mults), <= 8                  :525 */
    int32_t dim_mults[8];
    int32_t attention;        /* 0/1                                   :526 */
    int32_t timesteps;        /* size of the per-timestep bias table (diffusion T, 1000) */
} cindm_unet1d_desc;

int  cindm_unet1d_create(const cindm_unet1d_desc* desc, cindm_unet1d** out);
void cindm_unet1d_destroy(cindm_unet1d* h);

/* State-dict manifest (the reference's nn.Module.state_dict() key names, registration order,
 * SURVEY.md Appendix A.3): number of tensors; name/shape of tensor idx. */
int  cindm_unet1d_num_params(const cindm_unet1d* h);
int  cindm_unet1d_param_info(const cindm_unet1d* h, int idx, char* name, int name_cap,
                             int64_t shape[4], int* ndim);
/* Copy one state-dict tensor (PyTorch layout, fp32) into the handle; src may be a host
 * (on_device=0) or device (on_device=1) pointer.  Replaces load_state_dict for this module. */
int  cindm_unet1d_set_param(cindm_unet1d* h, const char* key, const float* src, int64_t numel,
                            int on_device);
/* Optional: sinusoidal embedding table [timesteps, dim] (host fp32), row t = SinusoidalPosEmb(t)
 * (:151-158).  If never called, *_finalize computes it with libm expf/sinf/cosf. */
int  cindm_unet1d_set_sinusoid_table(cindm_unet1d* h, const float* table_host, int64_t numel);
/* Repack weights to the kernels' [tap][Cin][Cout] layout and precompute, for every timestep,
 * the time path time_mlp -> per-block Mish->Linear biases (:537-542, :493-497, :509) with the
 * GEMM kernels.  Synchronises `stream`. */
int  cindm_unet1d_finalize(cindm_unet1d* h, void* stream);

/* Kernel-path selection for this handle (before *_finalize; changing an option un-finalizes the handle, except the run-time options
 * "no_exchange", "recover", "tune").  Every alternative path computes the same function (the parity suite runs all of them); defaults
 * are the fast path.  Keys (round 6: 24; DESIGN.md section 4.6 lists what was removed and why):
 *   "mfma_f32" (1 = exact fp32 MFMA kernels instead of the split-fp16 ones), "local_gn", "attn_site", "attn_head" (0 / 1 / 2),
 *   "level0" (master switch of the level kernels), "level1" (0 / 1 / 2 samples per workgroup), "ups_last", "ups_tail", "dconv", "dconv2",
 *   "dresample" (0 / 1 / 2: general kernel / 32 / 16-or-32 columns per workgroup), "l2_prefetch" (launches touch their successor's weights), "ws_alias", "pingpong" (the sample loops keep t / step index /
 *   epochs in two slots advanced by the step's update), "fuse_update" (plain single-model steps apply the update inside the last U-Net
 *   kernel), "fuse_gather" (time composition of two-body states: the first U-Net kernel reads the state's windows in place),
 *   "taps" (1 = block outputs that live only inside a level kernel are also stored for cindm_unet1d_tap; off on the sampling path),
 *   "auto_range" (1 = the range rule: a checkpoint outside the split-fp16 window runs on the fp32 kernels), "range_fallback" (read-only:
 *   1 a weight left the window 2^-12 <= max|w| <= 2^15, 2 the calibration batch overflowed, 3 the caller's own batch did --
 *   cindm_unet1d_range_escalate), "no_exchange", "recover", "stress", "tune" (same-box A/B word: bit 0 = round 5's L2 warm-up placement,
 *   regions and issuers, bit 1 = round 5's plain output stores, bit 2 + i = launch i of the forward issues no warm-up), "dbg" (timing ablations / forced time-outs: wrong results).
 * No reference counterpart (PyTorch picks its own kernels). */
int  cindm_unet1d_set_option(cindm_unet1d* h, const char* key, int32_t value);
int  cindm_unet1d_get_option(const cindm_unet1d* h, const char* key, int32_t* value);

/* Synchronises `stream` and reports a device-side fault of earlier forwards (the bounded in-kernel exchange between
 * workgroup pairs of the C = 512 GroupNorms timing out).  0 = healthy. */
int  cindm_unet1d_status(cindm_unet1d* h, void* stream);
/* The same check for callers that recover: 0 = healthy, 1 = an exchange of an earlier forward on `stream` timed out (its results
 * are invalid; the flag is cleared), < 0 = error.  Recovery = set_option("no_exchange", 1), re-issue the work, set it back: with
 * "no_exchange" the forward runs on the kernels that exchange nothing between workgroups (per-layer conv_gemm_h3 launches with
 * consumer-side GroupNorm at C = 512, the one-workgroup attention site kernel), which cannot time out.  "no_exchange" is a
 * RUN-TIME option: it does not un-finalize the handle.  The sample loops (cindm_ddpm1d_sample / _sample_ddim / _sample_guided)
 * do this by themselves: a chain whose exchange timed out (foreign load on the device kept a partner workgroup from becoming
 * resident) is re-run ONCE from its initial state on the exchange-free kernels; cindm_unet1d_recovered() counts those re-runs. */
int  cindm_unet1d_poll(cindm_unet1d* h, void* stream);
int  cindm_unet1d_recovered(const cindm_unet1d* h);
/* The range rule on the caller's own data (ABI 4): on = 1 repacks a handle that runs the split-fp16 kernels for the exact fp32-MFMA
 * kernels ("range_fallback" then reads 3) -- called by the Python face when the FIRST forward / chain after a weight synchronisation
 * returns inf / nan; on = 0 undoes exactly that.  Synchronises the stream and repacks the weights as cindm_unet1d_finalize does (a
 * finalisation step, not part of a chain's steady state: at most once per weight synchronisation). */
int  cindm_unet1d_range_escalate(cindm_unet1d* h, int32_t on, void* stream);
/* Profiling builds only (cindm_amd/build.py --prof: -DCINDM_PHASE_PROF -> libcindm_hip_prof.so): per-launch, per-workgroup,
 * per-wave phase clocks (s_memrealtime, 100 MHz) of the level kernels, dconv2_kernel, dresample_kernel and attn1d_head_kernel.
 * enable: (re)allocates the record buffer and arms it for every following forward of this handle (inside graph replays too);
 * read: copies the records of the LAST forward/replayed step, [launch slot][workgroup < 1024][wave < 8][16 stamps] uint64,
 * into host memory (dst_cap in uint64 words) and returns the number of launch slots used (names via cindm_unet1d_phase_prof_name).
 * In a production build enable returns -1 ("not a profiling build"). */
int  cindm_unet1d_phase_prof_enable(cindm_unet1d* h, int32_t on);
int  cindm_unet1d_phase_prof_read(cindm_unet1d* h, unsigned long long* dst, int64_t dst_cap, void* stream);
const char* cindm_unet1d_phase_prof_name(const cindm_unet1d* h, int32_t slot);

size_t cindm_unet1d_workspace_bytes(const cindm_unet1d* h, int64_t rows);
/* eps[rows,horizon,F] = TemporalUnet1D.forward(x[rows,horizon,F], time=t)   (:610-646).
 * All rows share timestep t (as every caller on the sampling path does).  If t_dev != NULL the
 * timestep is read from that device int32 at kernel run time (graph-replayable) and t is ignored. */
int  cindm_unet1d_forward(cindm_unet1d* h, const float* x, int32_t t, const int32_t* t_dev,
                          float* eps, int64_t rows, void* ws, size_t ws_bytes, void* stream);
/* Debug/parity aid: copy a named intermediate activation of the LAST forward (same rows/ws)
 * to dst (device, capacity dst_cap floats) in channel-last layout [rows, L, C].
 * Names: reference module paths, e.g. "downs.0.0", "mid_attn", "ups.2.3", "final_conv.0". */
int  cindm_unet1d_tap(cindm_unet1d* h, const char* name, int64_t rows, void* ws, float* dst,
                      int64_t dst_cap, int64_t shape[3], void* stream);
/* Instrumented forward for bench.py's roofline leg: as cindm_unet1d_forward, but every launch is
 * issued 8 times back-to-back (launches are idempotent) inside one pair of HIP events recorded on `stream`,
 * the bracket time divided by 8 being the launch's duration (this amortises the ~6 us cost of an event
 * bracket and includes the dependent-launch gap a kernel also pays inside the sampling graph); returns per kernel kind k
 * (0..4 = conv_gemm_kernel<T> for T = 0,1,3,4,5; 5 = linattn_core_kernel) the launch count, the summed
 * duration in ms and the summed ALGORITHMIC FLOPs (2*M*N*K of the layer, no padding).
 * Synchronises `stream`. */
int  cindm_unet1d_profile(cindm_unet1d* h, const float* x, int32_t t, float* eps, int64_t rows,
                          void* ws, size_t ws_bytes, void* stream,
                          int32_t counts[6], float ms[6], double flops[6]);
/* Same, one record per launch in issue order (at most cap): kernel kind, duration ms, algorithmic FLOPs,
 * and (grid.x, grid.y, pipeline stages) triples. */
int  cindm_unet1d_profile_detail(cindm_unet1d* h, const float* x, int32_t t, float* eps, int64_t rows,
                                 void* ws, size_t ws_bytes, void* stream, int32_t cap, int32_t* n_out,
                                 int32_t* kind, float* ms, double* flops, int32_t* grid_xy_stages);
/* Number of kernel launches one forward issues (for DESIGN/bench bookkeeping). */
int  cindm_unet1d_launches_per_forward(const cindm_unet1d* h);

/* ------------------------------------------------------------------ GaussianDiffusion1D (sampling half)
 * Replaces GaussianDiffusion1D.model_predictions :951, gradient :1857, p_mean_variance :1033,
 * q_posterior :938, predict_start_from_noise :914, p_sample :1047, p_sample_compose_inside :1190,
 * p_sample_compose_outside :1380, p_sample_loop :1656, sample_compose_multibodies :1986,
 * q_sample :2399 of model/diffusion_1d.py. */
typedef struct {
    int32_t timesteps;
    /* the 13 fp32 buffers registered at :873-910, host pointers, each [timesteps] */
    const float* betas;
    const float* alphas_cumprod;
    const float* alphas_cumprod_prev;
    const float* sqrt_alphas_cumprod;
    const float* sqrt_one_minus_alphas_cumprod;
    const float* log_one_minus_alphas_cumprod;
    const float* sqrt_recip_alphas_cumprod;
    const float* sqrt_recipm1_alphas_cumprod;
    const float* posterior_variance;
    const float* posterior_log_variance_clipped;
    const float* posterior_mean_coef1;
    const float* posterior_mean_coef2;
    const float* loss_weight;
} cindm_sched_desc;

int  cindm_ddpm1d_create(const cindm_sched_desc* desc, cindm_ddpm1d** out);
void cindm_ddpm1d_destroy(cindm_ddpm1d* h);

enum {
    CINDM_COMPOSE_PLAIN = 0,          /* self.model(x) on the whole state                  :1006 */
    CINDM_COMPOSE_MEAN_INSIDE = 1,    /* compose_mode "mean-inside"                        :994-996 */
    CINDM_COMPOSE_SUM_INSIDE = 2,     /* compose_mode "sum-inside"                         :997-999 */
    CINDM_COMPOSE_MEAN_OUTSIDE = 3,   /* p_sample_compose_outside compose_mode "mean"      :1447-1452 */
    CINDM_COMPOSE_NOISESUM_OUTSIDE = 4, /* compose_mode "noise_sum"                        :1453-1463 */
    CINDM_COMPOSE_MULTIBODY = 5       /* gradient(): pair model + unconditioned model      :1865-1926 */
};
enum { CINDM_OBJ_PRED_NOISE = 0, CINDM_OBJ_PRED_X0 = 1, CINDM_OBJ_PRED_V = 2 };

typedef struct {
    int32_t mode;               /* CINDM_COMPOSE_* */
    int32_t n_windows;          /* n_composed + 1                                          :977 */
    int32_t compose_start_step; /* window stride                                           :978 */
    int32_t window;             /* single_model_step = image_size                          :963 */
    int32_t n_bodies;           /* compose_n_bodies (state feature dim = 4*n_bodies)       :964 */
    int32_t cond_steps;         /* conditioned_steps: rows of `cond` prepended to x        :956-957 */
    int32_t objective;          /* CINDM_OBJ_*                                             :1010-1027 */
    int32_t clip_denoised;      /* x_start.clamp_(-1,1)                                    :1038-1039 */
    float   uncond_coef;        /* coefficient_unconditioned_grad = 1.4 (MULTIBODY)        :1900 */
} cindm_compose_desc;

/* Bytes of the step workspace: the U-Net inputs / predictions of a composed step, the U-Net workspace(s), the guided step's
 * staging, AND (ABI 4) the chain-level buffers of the sample loops -- the x_T snapshot the exchange-time-out recovery re-runs
 * from (B * L_tot * F floats) and the DDIM loop's per-step tables (5 words per U-Net timestep).  Nothing is allocated after
 * *_create / *_finalize: no entry point below calls hipMalloc / hipFree (asserted by a CPU test over the sources). */
size_t cindm_ddpm1d_workspace_bytes(const cindm_ddpm1d* h, const cindm_unet1d* pair,
                                    const cindm_unet1d* uncond, const cindm_compose_desc* c, int64_t B);

/* p_mean_variance (:1033-1044) for one timestep: evaluate the composed U-Nets on x
 * [B, L_tot, 4*n_bodies] (with cond [B, cond_steps, F] prepended when cond_steps > 0), aggregate
 * eps over windows / body pairs, x0 = predict_start_from_noise, clamp, posterior mean.
 * Writes mean_out, x0_out, eps_out (each [B, L_tot, F], any may be NULL). */
int  cindm_ddpm1d_predict(cindm_ddpm1d* h, cindm_unet1d* pair, cindm_unet1d* uncond,
                          const cindm_compose_desc* c, const float* x, const float* cond,
                          int32_t t, const int32_t* t_dev, int64_t B,
                          float* mean_out, float* x0_out, float* eps_out,
                          void* ws, size_t ws_bytes, void* stream);

/* One full reverse step without design guidance (p_sample :1061-1120 / p_sample_compose_inside
 * :1209-1283 / p_sample_compose_outside :1406-1523 with design_fn=None):
 *   x <- mean + exp(0.5*logvar_t) * z   (z = 0 at t == 0), in place.
 * z = noise[B,L_tot,F] if noise != NULL, else the library's counter-based Gaussian
 * keyed by (seed, sample_offset + b, t, element).
 * Inpainting (:1715-1718): if inpaint_cond != NULL, afterwards
 *   x[:, :inpaint_steps] = q_sample(inpaint_cond, t, inpaint_noise or counter-based noise). */
int  cindm_ddpm1d_step(cindm_ddpm1d* h, cindm_unet1d* pair, cindm_unet1d* uncond,
                       const cindm_compose_desc* c, float* x, const float* cond,
                       const float* noise, uint64_t seed, int64_t sample_offset,
                       const float* inpaint_cond, int32_t inpaint_steps, const float* inpaint_noise,
                       int32_t t, const int32_t* t_dev, int64_t B, float* x0_out,
                       void* ws, size_t ws_bytes, void* stream);

/* The reverse loop (p_sample_loop :1682-1720 / sample_compose_multibodies :2000-2033 without
 * design guidance): for t = t_start, t_start-1, ..., t_end: cindm_ddpm1d_step.  One step is
 * captured into a hipGraph (timestep held in a device counter) and replayed.
 * noise_steps: NULL (counter-based noise) or device tensor [timesteps, B, L_tot, F] indexed by t;
 * inpaint_noise_steps likewise [timesteps, B, inpaint_steps, F]. */
int  cindm_ddpm1d_sample(cindm_ddpm1d* h, cindm_unet1d* pair, cindm_unet1d* uncond,
                         const cindm_compose_desc* c, float* x, const float* cond,
                         const float* noise_steps, uint64_t seed, int64_t sample_offset,
                         const float* inpaint_cond, int32_t inpaint_steps,
                         const float* inpaint_noise_steps,
                         int32_t t_start, int32_t t_end, int64_t B,
                         void* ws, size_t ws_bytes, void* stream, int32_t use_graph);

/* Built-in design objective: the paper's point objective (inference/inverse_design_diffusion_1d.py:211-229),
 *   "L2":       coef * sum_bodies sum_b mean_{last n steps} || pos - target ||_2
 *   "L2square": coef * sum_bodies sum_b mean_{last n steps} || pos - target ||_2^2
 *   (+ time_consistency_coef * sum_b mean_l || pos[l+1] - pos[l] ||^2 over the position channels),
 * whose gradient with respect to the state is evaluated in closed form inside the update kernel. */
typedef struct {
    int32_t mode;               /* 1 = "L2", 2 = "L2square" */
    int32_t alpha;              /* 0: "standard" (gradient as is), 1: "standard-alpha" (x eta_t)   :1243-1248 */
    int32_t recurrence;         /* 0: non-recurrence branch; N >= 1: "-recurrence-N"                :1284-1370 */
    int32_t last_n_step;
    float   coef;
    float   time_consistency_coef;
    float   pos_target[2];
} cindm_design_desc;

/* The reverse loop WITH design guidance by the built-in objective ("standard" / "standard-alpha", optionally
 * "-recurrence-N"; p_sample :1061-1186, p_sample_compose_inside :1209-1370, p_sample_compose_outside :1406-1652):
 * per step, max(N, 1) x [ p_mean_variance; pred = mean - [eta_t] grad objective(x); overwrite the first
 * overwrite_steps rows with initial_state_overwrite [B, overwrite_steps, F] if given; relaxation
 * x <- sqrt(abar_t/abar_{t-1}) pred + sqrt(1 - abar_t/abar_{t-1}) z' ] with the last iteration's
 * pred + sigma_t z as the result -- all inside the captured step, no host code in the loop.
 * recur_noise_steps: NULL (counter-based) or [timesteps, max(N,1), B, L_tot, F] indexed by t. */
int  cindm_ddpm1d_sample_guided(cindm_ddpm1d* h, cindm_unet1d* pair, cindm_unet1d* uncond,
                                const cindm_compose_desc* c, const cindm_design_desc* dz, float* x,
                                const float* cond, const float* noise_steps,
                                const float* recur_noise_steps, uint64_t seed, int64_t sample_offset,
                                const float* inpaint_cond, int32_t inpaint_steps,
                                const float* inpaint_noise_steps,
                                const float* initial_state_overwrite, int32_t overwrite_steps,
                                int32_t t_start, int32_t t_end, int64_t B,
                                void* ws, size_t ws_bytes, void* stream, int32_t use_graph);

/* DDIM loop (ddim_sample, model/diffusion_1d.py:1724-1804, design_fn == None): n_steps updates
 * x <- x0 * sqrt(alpha_next) + c * eps + sigma * z at times[0] > times[1] > ... > times[n_steps]
 * (times[n_steps] == -1: the last update returns x0); times [n_steps + 1] and coefs [n_steps][3] =
 * (sqrt(alpha_next), c, sigma) are HOST arrays computed by the caller in the reference's tensor
 * arithmetic (:1743-1777).  x0 is clamped when c->clip_denoised, eps is the model's (composed)
 * prediction, not re-derived (:1755).  noise_steps / inpaint_noise_steps, when given, are indexed
 * by the STEP index: [n_steps, B, L, F] / [n_steps, B, inpaint_steps, F]. */
int  cindm_ddpm1d_sample_ddim(cindm_ddpm1d* h, cindm_unet1d* pair, cindm_unet1d* uncond,
                              const cindm_compose_desc* c, float* x, const float* cond,
                              int32_t n_steps, const int32_t* times, const float* coefs,
                              const float* noise_steps, uint64_t seed, int64_t sample_offset,
                              const float* inpaint_cond, int32_t inpaint_steps,
                              const float* inpaint_noise_steps, int64_t B,
                              void* ws, size_t ws_bytes, void* stream, int32_t use_graph);

/* out[n] ~ N(0,1): the library's counter-based Gaussian (Philox4x32-10 + Box-Muller) for the
 * initial state x_T (:1673), keyed by (seed, sample_offset + b, step_tag, element);
 * out is [B, per_sample]. */
int  cindm_fill_normal(float* out, int64_t B, int64_t per_sample, uint64_t seed,
                       int64_t sample_offset, int32_t step_tag, void* stream);

/* Timing aid for bench.py: elapsed ms of `n_replays` steps measured with HIP events on `stream`
 * is done by the caller; this returns the number of kernel launches in one step. */
int  cindm_ddpm1d_launches_per_step(const cindm_ddpm1d* h, const cindm_unet1d* pair,
                                    const cindm_unet1d* uncond, const cindm_compose_desc* c);

/* What the reverse step emitted LAST on this handle consisted of (by cindm_ddpm1d_step / _predict or by the
 * capture of a sample loop): `launches` = kernels launched (gather + U-Net(s) + update + step counter), as
 * launched, not modelled; `fused_update` = 1 when the reverse-step update (:1033-1044, :1281) ran inside the
 * U-Net's last kernel instead of compose_update_kernel.  Tests assert the fused path through this. */
int  cindm_ddpm1d_last_step_info(const cindm_ddpm1d* h, int32_t* launches, int32_t* fused_update);
/* What the last sampling chain of this handle did: info[0] = 1 when it was re-run once on the exchange-free plan after an in-kernel
 * exchange timed out; info[1] = 1 when it ran on the exchange-free plan from the start because another chain was already in flight
 * on the device in this process (one chain per device is the rule of the exchange kernels; the registry is per process);
 * info[2] = chains in flight on the device when it started, itself included.  No reference counterpart. */
int  cindm_ddpm1d_last_chain_info(const cindm_ddpm1d* h, int32_t info[4]);

/* ===================================================================== 2-D airfoil path
 * Replaces Unet.forward (model/diffusion_2d.py:369-408) and GaussianDiffusion.p_sample /
 * p_sample_loop (model/diffusion_2d.py:788-907) of the reference for the sampling path
 * (no self-conditioning, no learned variance, objective pred_noise).
 *
 * Device layout of states and model outputs: channel-last fp32 [images, H*W, CP] with
 * CP = cindm_unet2d_padded_channels() (channels rounded up to a multiple of 4; padding
 * channels are zero).  An "image" is one boundary of one design: images = B * num_boundaries,
 * boundary-major inside a design, exactly the reference's [B, nb*C, H, W] -> [B*nb, C, H, W]
 * view (:901). */
typedef struct cindm_unet2d cindm_unet2d;

typedef struct {
    int32_t dim;              /* Unet(dim=...)                              :283 */
    int32_t n_mults;
    int32_t dim_mults[4];     /* entries 1 or 2                             :287 */
    int32_t channels;         /* 21 for the airfoil states (6 frames*3 + 3) :288 */
    int32_t image_size;       /* square, power of two <= 64                 :660 */
    int32_t timesteps;        /* rows of the per-timestep scale/shift table */
} cindm_unet2d_desc;

int  cindm_unet2d_create(const cindm_unet2d_desc* desc, cindm_unet2d** out);
void cindm_unet2d_destroy(cindm_unet2d* h);
int  cindm_unet2d_num_params(const cindm_unet2d* h);
int  cindm_unet2d_param_info(const cindm_unet2d* h, int idx, char* name, int name_cap,
                             int64_t shape[4], int* ndim);
int  cindm_unet2d_set_param(cindm_unet2d* h, const char* key, const float* src, int64_t numel,
                            int on_device);
int  cindm_unet2d_set_sinusoid_table(cindm_unet2d* h, const float* table, int64_t numel);
/* Weight standardisation (WeightStandardizedConv2d :116-124) folded, MFMA-fragment repack,
 * time path (SinusoidalPosEmb -> Linear -> GELU -> Linear -> per block SiLU -> Linear, :320-326,
 * :205-208) evaluated for every timestep into a device table. */
/* Kernel-path selection, as cindm_unet1d_set_option.  Keys: "mfma_f32", "la_site", "conv_ws" (1 = persistent
 * wave-specialised 3x3 kernel, 0 = per-tile kernel, 2 / 3 = only the plain-source / GroupNorm-on-load convolutions on
 * it), "ws_nosplit" (0 / 1 / 2: where that kernel splits K over its matrix waves), "tail_h3" (ResnetBlock tails with a res_conv GEMM
 * on the split-fp16 products), "ws_alias" (block-internal temporaries share workspace), "la_wpi" / "la_nsplit" (workgroups per image of
 * the LinearAttention kernels), "stress", "auto_range", "range_fallback" (read-only), "dbg2". */
int  cindm_unet2d_set_option(cindm_unet2d* h, const char* key, int32_t value);
int  cindm_unet2d_get_option(const cindm_unet2d* h, const char* key, int32_t* value);
int  cindm_unet2d_finalize(cindm_unet2d* h, void* stream);
int  cindm_unet2d_padded_channels(const cindm_unet2d* h);
size_t cindm_unet2d_workspace_bytes(const cindm_unet2d* h, int64_t images);
int  cindm_unet2d_launches_per_forward(const cindm_unet2d* h);
/* eps[images, H*W, CP] = Unet(x[images, H*W, CP], t); t_dev (device int32) wins over t. */
int  cindm_unet2d_forward(cindm_unet2d* h, const float* x, int32_t t, const int32_t* t_dev,
                          float* eps, int64_t images, void* ws, size_t ws_bytes, void* stream);
/* Measurement hook for bench.py: one forward with every launch bracketed by HIP events on
 * `stream`; per kernel kind (0 conv3x3, 1 conv1x1, 2 stem 7x7, 3 qkv 1x1, 4 linear attention,
 * 5 full attention, 6 statistics / LayerNorm-residual): launches, total ms, algorithmic FLOPs. */
int  cindm_unet2d_profile(cindm_unet2d* h, const float* x, int32_t t, float* eps, int64_t images,
                          void* ws, size_t ws_bytes, void* stream, int32_t counts[7], float ms[7],
                          double flops[7]);
/* Test hook: copy the intermediate activation `name` of the last forward into dst as
 * [images, H*W, C]; shape receives (images, H*W, C). */
int  cindm_unet2d_tap(cindm_unet2d* h, const char* name, int64_t images, void* ws, float* dst,
                      int64_t dst_cap, int64_t shape[3], void* stream);

size_t cindm_ddpm2d_workspace_bytes(const cindm_unet2d* u, int64_t images);
/* One reverse step x_t -> x_{t-1} of GaussianDiffusion.p_sample (:788-808) for B designs of
 * nb boundaries: Unet on all B*nb images; the model output's state channels (all but the last
 * 3) are averaged (use_average_share=1) or summed over the nb boundaries of a design
 * (share_states_over_boundaries :712-725); x0 from eps, clamp; posterior mean; + sigma_t * z
 * (use_average_share bit 1 set = the constructor's share_noise False, p_mean_variance :757-773: the
 * prediction is left alone and the clamped x_start, then the posterior mean, are shared instead)
 * with z shared over the boundaries for the state channels (sample_noise :775-785).  z comes
 * from noise_state [B, H*W, C-3] / noise_boundary [B*nb, H*W, 3] when given, else from the
 * counter-based generator (seed, sample_offset + b, t).  In place on x.  Optional outputs:
 * x0_out (clamped x_start), mean_out (posterior mean), both [B*nb, H*W, CP].
 * Bits 4-5 of use_average_share carry the diffusion's objective (CINDM_OBJ_*; :743-753): pred_x0 takes the model output as
 * x_start, pred_v takes x_start = sqrt(abar_t) x - sqrt(1 - abar_t) v; for both the reference shares nothing inside
 * model_predictions (share_noise False still shares x_start and the mean afterwards) and re-derives the noise from x_start.
 * The same word is used by cindm_ddpm2d_predict, cindm_ddpm2d_sample and cindm_ddpm2d_sample_force. */
int  cindm_ddpm2d_step(cindm_ddpm1d* sched, cindm_unet2d* u, float* x, int64_t B, int32_t nb,
                       int32_t use_average_share, int32_t clip_denoised,
                       const float* noise_state, const float* noise_boundary, uint64_t seed,
                       int64_t sample_offset, int32_t t, const int32_t* t_dev, float* x0_out,
                       float* mean_out, void* ws, size_t ws_bytes, void* stream);
/* GaussianDiffusion.model_predictions (:727-754; objective in bits 4-5 of use_average_share, see above) for B designs of nb boundaries: Unet on all
 * B*nb images; share_noise != 0: the model output's state channels are averaged (use_average_share = 1) or summed (0)
 * over the boundaries of a design (:732-733); x_start = predict_start_from_noise(x, t, pred_noise), clamped to [-1, 1]
 * when clip_x_start; rederive_pred_noise (with clip_x_start, :738-739): pred_noise = predict_noise_from_start(x, t,
 * x_start) = (sqrt_recip_t x - x_start) / sqrt_recipm1_t.  Writes pred_noise_out and x_start_out, both [B*nb, H*W, CP]
 * (either may be NULL); x is not modified. */
int  cindm_ddpm2d_predict(cindm_ddpm1d* sched, cindm_unet2d* u, const float* x, int64_t B, int32_t nb,
                          int32_t use_average_share, int32_t share_noise, int32_t clip_x_start,
                          int32_t rederive_pred_noise, int32_t t, const int32_t* t_dev,
                          float* pred_noise_out, float* x_start_out, void* ws, size_t ws_bytes, void* stream);
/* p_sample_loop (:893-907) without design guidance: steps t_start .. t_end in place on x, one
 * captured HIP graph replayed per step when use_graph.  noise_*_steps, when given, are indexed
 * [timesteps, ...] by t. */
int  cindm_ddpm2d_sample(cindm_ddpm1d* sched, cindm_unet2d* u, float* x, int64_t B, int32_t nb,
                         int32_t use_average_share, const float* noise_state_steps,
                         const float* noise_boundary_steps, uint64_t seed, int64_t sample_offset,
                         int32_t t_start, int32_t t_end, void* ws, size_t ws_bytes, void* stream,
                         int32_t use_graph);
/* x_T for the 2-D path (sample_noise :775-785, :895): state channels shared over the boundaries of a design. */
int  cindm_fill_noise2d(float* x, int64_t B, int32_t nb, int32_t hw, int32_t channels,
                        int32_t padded_channels, uint64_t seed, int64_t sample_offset,
                        int32_t step_tag, void* stream);


/* ------------------------------------------------------------------ ForceUnet and the airfoil design gradient
 * Replaces ForceUnet.__init__/forward, model/diffusion_2d.py:411-486 (the surrogate of the lift / drag forces), and --
 * instead of torch.autograd.grad through it -- the design_fn of inference/inverse_design_2d.py:208-214 built from
 * force_fn :98-132 (sum_boundary = True) and overlap_fn :134-143.  Only input gradients exist (frozen weights).
 * Tensors: channel-last fp32; network input [images, H*W, channels]; sampler state [B*nb, H*W, CP] as cindm_ddpm2d_*. */
typedef struct cindm_forceunet cindm_forceunet;
typedef struct {
    int32_t dim;              /* 64                                         :414 */
    int32_t n_mults;          /* len(dim_mults) <= 4                        :417 */
    int32_t dim_mults[4];     /* (1, 2, 4, 8): the bottleneck must be 512 wide (final = Linear(512, 2), :458) */
    int32_t channels;         /* 4 = (pressure, boundary mask, 2 offsets)   :418 */
    int32_t image_size;       /* 64 = 8 * 2^(n_mults - 1): the coarsest level is 8 x 8 (other sizes: fewer levels) */
} cindm_forceunet_desc;

int  cindm_forceunet_create(const cindm_forceunet_desc* desc, cindm_forceunet** out);
void cindm_forceunet_destroy(cindm_forceunet* h);
int  cindm_forceunet_num_params(const cindm_forceunet* h);
int  cindm_forceunet_param_info(const cindm_forceunet* h, int idx, char* name, int name_cap, int64_t shape[4], int* ndim);
int  cindm_forceunet_set_param(cindm_forceunet* h, const char* key, const float* src, int64_t numel, int on_device);
/* Kernel-path options of a handle (as cindm_unet1d_set_option; take effect at the next finalize): "h3" = 0 keeps the
 * forward 3x3 convolutions, "h3_bwd" = 0 the input-gradient ones, on the exact fp32 MFMA kernel instead of the
 * split-fp16 one (both paths meet the 2e-5 parity bound); "auto_range" = 0 skips the calibration forward of the range
 * rule; "range_fallback" (read-only) = 1 when that forward switched the handle to the fp32 convolutions; "stress" > 0:
 * pseudo-random delays before the producer -> consumer hand-overs of the persistent convolution kernel (race tests);
 * "la_fused" (0 = LinearAttention sites layer by layer), "gn_bwd_fused" (0 / 1 / 2: the GroupNorm + SiLU derivative), "ws_nosplit",
 * and the run-time options "no_exchange", "recover", "dbg". */
int  cindm_forceunet_set_option(cindm_forceunet* h, const char* key, int32_t value);
int  cindm_forceunet_get_option(const cindm_forceunet* h, const char* key, int32_t* value);
/* folds weight standardisation, packs forward and backward-data fragments, runs the range rule's calibration forward */
int  cindm_forceunet_finalize(cindm_forceunet* h, void* stream);
size_t cindm_forceunet_workspace_bytes(const cindm_forceunet* h, int64_t images, int32_t with_grad);
/* out[images, 2] = ForceUnet.forward(x)   (:460-486) */
int  cindm_forceunet_forward(cindm_forceunet* h, const float* x, float* out, int64_t images, void* ws, size_t ws_bytes, void* stream);
/* the same forward, and dx = d( sum_images lambda_force * |out[:,0]| + out[:,1] ) / dx  (what force_fn differentiates, :113-117) */
int  cindm_forceunet_grad(cindm_forceunet* h, const float* x, float lambda_force, float* out, float* dx, int64_t images,
                          void* ws, size_t ws_bytes, void* stream);
/* the same forward, and dx = d( sum_images dout[img] . out[img] ) / dx for a caller-given dout [images, 2]: the backward of
 * ForceUnet.forward under torch.autograd (cindm_amd.ForceUnet registers it), so force_fn of inverse_design_2d.py:98-132
 * runs against this model as written */
int  cindm_forceunet_vjp(cindm_forceunet* h, const float* x, const float* dout, float* out, float* dx, int64_t images,
                         void* ws, size_t ws_bytes, void* stream);
/* 1 when an in-kernel exchange between workgroups of this handle's kernels (the GroupNorm derivative on whole pixel rows, option
 * gn_bwd_fused = 2) timed out since the last call -- the affected gradients are NaN, never finite and wrong --, 0 otherwise; clears
 * the flag; synchronises `stream`.  cindm_ddpm2d_sample_force reads it at the end of the chain and re-runs ONCE from the kept x_T
 * on the exchange-free derivative (option recover = 0: returns an error instead); after cindm_forceunet_grad / _vjp /
 * cindm_airfoil_design_grad the caller polls it (the Python face does, and re-runs the call with option no_exchange = 1).
 * cindm_forceunet_recovered: how many chains / calls were re-run that way.  No reference counterpart. */
int  cindm_forceunet_status(cindm_forceunet* h, void* stream);
int  cindm_forceunet_recovered(const cindm_forceunet* h);
/* grad[B*nb, H*W, CP] = design_fn(x) = grad_force + lambda_overlap * grad_overlap (inverse_design_2d.py:208-214), the
 * tensor GaussianDiffusion.p_sample subtracts under "standard" / "standard-alpha" guidance (model/diffusion_2d.py:813-817).
 * frames = (real channels - 3) / 3; p_min / p_max: the pressure normalisation of the data set (:85-87). */
/* frames_per_pass: how many of the `frames` surrogate passes of one gradient run as one batch (a divisor of frames;
 * cindm_airfoil_design_grad uses the largest one whose workspace fits what it is given) */
size_t cindm_airfoil_design_workspace_bytes(const cindm_forceunet* h, int64_t B, int32_t nb, int32_t frames_per_pass);
/* sum_boundary: force_fn's switch (:98-132): 1 = the surrogate sees the clamped sum of a design's boundaries and the pressure
 * in simulator units (:100-120, the script's default); 0 = each copy's own boundary channels and the normalised pressure,
 * forces summed over the boundaries of a design (:122-130). */
int  cindm_airfoil_design_grad(cindm_forceunet* h, const float* x, int64_t B, int32_t nb, int32_t frames, int32_t CP,
                               float p_min, float p_max, float lambda_force, float lambda_overlap, int32_t downsampling_factor,
                               int32_t sum_boundary, float* grad, void* ws, size_t ws_bytes, void* stream);
/* Design-guided 2-D sampling with the airfoil objective inside the captured step (inference/inverse_design_2d.py:236-244
 * with design_guidance = "standard-alpha"; p_sample's guided tail model/diffusion_2d.py:806-845): per reverse step
 * g = cindm_airfoil_design_grad(x_t); x_{t-1} = p_sample(x_t) (as cindm_ddpm2d_sample); x_{t-1} -= eta[t] * g, with eta a
 * device table of `timesteps` floats (coeff_ratio * betas.flip(0)).  grad: a device buffer shaped like x; ws as
 * cindm_ddpm2d_sample, ws_force as cindm_airfoil_design_grad.  One hipGraph per call, replayed once per timestep.
 * SYNCHRONISES `stream` before it returns, with use_graph = 0 as well (since round 5): the chain's exchange flag is read at its end and a
 * timed-out chain is re-run once from the x_T kept in `ws` (round 6: a slice of the caller's workspace -- nothing is allocated); an eager
 * chain therefore cannot be enqueued asynchronously or captured by the caller.  The error word is cleared when it is read. */
int  cindm_ddpm2d_sample_force(cindm_ddpm1d* sched, cindm_unet2d* u, cindm_forceunet* f, float* x, int64_t B, int32_t nb,
                               int32_t use_average_share, const float* noise_state_steps,
                               const float* noise_boundary_steps, uint64_t seed, int64_t sample_offset,
                               int32_t t_start, int32_t t_end, int32_t frames, float p_min, float p_max,
                               float lambda_force, float lambda_overlap, int32_t down_factor, int32_t sum_boundary,
                               const float* eta, float* grad, void* ws, size_t ws_bytes, void* ws_force,
                               size_t ws_force_bytes, void* stream, int32_t use_graph);

/* ------------------------------------------------------------------ multi-GPU: the one collective of the path
 * the design batch is sharded over the ranks with no communication inside the reverse loop;
 * the final designs are all-gathered ONCE.  The reference has no counterpart (inference/inverse_design_diffusion_1d.py:161 is
 * single-device).  These entry points call RCCL (librccl.so, resolved with dlopen at first use: the library itself has no link
 * dependency on it) over xGMI:
 *   cindm_comm_unique_id   ncclGetUniqueId: 128 opaque bytes made by rank 0 and handed to every rank by the caller's own
 *                          rendezvous (the Python face broadcasts them over the torch.distributed group it already has);
 *   cindm_comm_init        ncclCommInitRank on the CURRENT device -> an opaque communicator;
 *   cindm_all_gather_designs   ncclAllGather(float): every rank contributes per_rank_elems floats at `local`, `out` receives
 *                          world * per_rank_elems floats in rank order; asynchronous on `stream`;
 *   cindm_comm_destroy     ncclCommDestroy.
 * Ragged shards are padded to the largest one by the caller (cindm_amd/dist.py). */
typedef struct cindm_comm cindm_comm;
int  cindm_comm_unique_id(unsigned char id[128]);
int  cindm_comm_init(const unsigned char id[128], int32_t world, int32_t rank, cindm_comm** out);
int  cindm_comm_world(const cindm_comm* c);
int  cindm_all_gather_designs(const float* local, float* out, int64_t per_rank_elems, cindm_comm* comm, void* stream);
void cindm_comm_destroy(cindm_comm* c);

#ifdef __cplusplus
}
#endif
#endif /* CINDM_HIP_H */
