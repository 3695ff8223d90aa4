/* libv2a_hip.so -- C ABI of the MI355X-native (gfx950) hot path of video-to-action.
 *
 * The reference (video-to-action/video-to-action-release) is pure Python on PyTorch and has NO FFI; its plugin surface for this
 * path is Python classes (GoalGaussianDiffusion / Unet_Libero, DiffusionUnetImagePolicy, Global_EnvReplayBuffer_Img).  The Python
 * shells under video-to-action-release_amd/{flowdiffusion,diffuser}/ keep those surfaces and bind THIS library with ctypes
 * (video-to-action-release_amd/v2a_hip/_lib.py; INTEGRATION.md shows the stub).  Each entry point below names the reference code it
 * replaces (paths relative to the reference root).
 *
 * Conventions: plain pointers + sizes, no torch types; every pointer is DEVICE memory unless marked HOST; fp32; activations
 * channels-last ([N,H,W,C]; sequences [N,1,T,C]; video [B,F,H,W,C]); asynchronous on `stream`, no device synchronisation, no
 * allocation (scratch is passed in; query with the *_workspace_bytes functions); returns 0 or a negative error:
 *   -1 bad argument, -2 kernel launch failed, -3 workspace too small, -4 replay episode shorter than act_len+1.
 * Thread-safe for distinct streams.
 */
#ifndef V2A_H
#define V2A_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* v2a_stream_t; /* hipStream_t */

/* activation ids (norm / act kernels) */
enum { V2A_ACT_NONE = 0, V2A_ACT_SILU = 1, V2A_ACT_RELU = 2, V2A_ACT_MISH = 3, V2A_ACT_GELU = 4 };

/* ---------------------------------------------------------------------------------------------- contractions (csrc/igemm.hip)
 * One implicit-GEMM kernel family (exact-f32 MFMA 32x32x2) replaces every torch conv / linear on the path:
 *   Conv3d spatial + temporal parts   flowdiffusion/flowdiffusion/guided_diffusion/guided_diffusion/nn.py:30-87
 *   nearest-x2 Upsample + conv        .../guided_diffusion/unet.py:86-115   (ups = 1: folded into the loader)
 *   decoder skip concat               .../guided_diffusion/unet.py:681      (x2: second source, no cat tensor)
 *   ResNet-18 convs                   diffuser/diffusion_policy/common/vision_nets.py:29-39 (torchvision resnet18)
 *   Conv1d k5/k3/k1, ConvTranspose1d  diffuser/diffusion_policy/model/conv1d_components.py:7-40
 *   nn.Linear                         conditional_unet1d.py:35-39,88-93; vision_nets.py:140; unet.py:204-210,482-486
 * and their data gradients (same kernel: weight pack mode 1, idil = forward stride, pad = k-1-pad).
 * w_packed: [Cout][KH][KW][C1+C2] (mode 0) -- a [Cout][Cin] Linear / 1x1 weight is already in that form.
 * y (+y2 when csplit > 0: channels [csplit,Cout) go to y2) [N,OH,OW,Cout]; rowvec [batches][Cout] is added per
 * (m / rows_per_batch, channel); residual [N,OH,OW,Cout].  bmode = 1: data gradient computed straight from the FORWARD pack
 * [Cred][KH][KW][Cout] of the layer (no second packed copy; needs Cred % 16 == 0 and Cout % 4 == 0). */
/* process-wide MFMA precision of the contraction kernels: 0 = exact f32 (parity configuration, default), 1 = bf16 inputs with f32
 * accumulation and fp32 HBM storage (performance configuration; the reference's GPU path is fp16 autocast). Returns the old mode. */
int v2a_set_precision(int mode);
int v2a_get_precision(void);
int v2a_set_policy_half(int f16);   /* 16-bit format of the policy's MFMA mode (v2a_set_precision(1)): 0 bf16 (default), 1 IEEE fp16; returns the old value */
int v2a_get_policy_half(void);
int v2a_debug_force_tile(int bm, int bn);   /* tuning aid: force the forward tile (128x128 | 128x64 | 64x64), 0,0 = heuristic */
int v2a_debug_force_wgrad_plan(int bm, int bn, int split);   /* tuning aid: force the weight-gradient tile / split; 0,0,0 = heuristic */
size_t v2a_conv2d_workspace_bytes(int M, int Cout, int K);
int v2a_conv2d_fwd(const float* x, const float* x2, const float* w_packed, const float* bias, const float* rowvec,
                   const float* residual, float* y, float* y2, int csplit, int N, int H, int W, int C1, int C2, int OH, int OW,
                   int Cout, int KH, int KW, int sh, int sw, int ph, int pw, int idil, int ups, int rows_per_batch, int bmode, void* workspace,
                   size_t workspace_bytes, v2a_stream_t stream);
/* weight gradient, written in the TORCH layout [Cout][Cin][KH][KW] (replaces autograd's conv backward-weight);
 * dbias != NULL: the bias gradient sum_rows dy is produced by the same launch */
size_t v2a_conv2d_wgrad_workspace_bytes(int M, int Cout, int K);
int v2a_conv2d_wgrad(const float* x, const float* x2, const float* dy, float* dw, float* dbias, int N, int H, int W, int C1, int C2, int OH,
                     int OW, int Cout, int KH, int KW, int sh, int sw, int ph, int pw, int idil, int ups, int accumulate,
                     void* workspace, size_t workspace_bytes, v2a_stream_t stream);
/* the same weight gradient from bf16 twins of x [N,H,W,C] and dy [N,OH,OW,Cout] (the rounded copies the bf16 forward / data-gradient
 * convs of the bf16-MFMA mode already made): half the operand traffic of the converting kernel.  Cout >= 64, K > 64, C % 8 == 0. */
size_t v2a_conv2d_wgrad_h_workspace_bytes(int M, int Cout, int K);
int v2a_conv2d_wgrad_h(const void* x_h, const void* x2_h /* twin of the second input of a channel concat, or NULL */, const void* dy_h,
                       float* dw, float* dbias, int N, int H, int W, int C, int C2, int OH, int OW, int Cout, int KH, int KW, int sh, int sw,
                       int ph, int pw, int idil, int ups, int accumulate, void* workspace, size_t workspace_bytes, v2a_stream_t stream);
/* tile / split-K plan the two launchers above will use for a problem size (benchmark labelling) */
int v2a_conv2d_plan(int M, int Cout, int K, int* bm, int* bn, int* split);
int v2a_debug_wgrad_dma(int on);   /* tuning aid: fp32 weight gradients on the LDS-DMA kernel (default on); returns the old value */
int v2a_conv2d_wgrad_plan(int M, int Cout, int K, int* bm, int* bn, int* split);
/* torch weight [Cout][Cin][KH][KW] -> mode 0: [Cout][KH][KW][Cin]; mode 1: [Cin][KH'][KW'][Cout] flipped (dgrad / transposed) */
int v2a_pack_weight(const float* src, float* dst, int Cout, int Cin, int KH, int KW, int mode, v2a_stream_t stream);
/* every pack of a model in two launches: table_dev int64 [n][7] = {src, dst fp32 or 0, Cout, Cin, taps, mode, dst bf16 or 0};
 * transposed 0: mode-0 rows, chunks_dev int32 {operand, start} of v2a_pack_chunk_elems() elements; transposed 1: mode-1 rows
 * (data-gradient operand = tap-reversed transpose), chunks_dev {operand, 64x64 tile index over [Cout x Cin*taps]}.
 * mode | 256: the 16-bit destination is the hi plane of THREE bf16 planes (hi, mid, lo of the fp32 value; Cout * Cin * taps elements apart)
 * instead of one twin: the pre-split weight operand of v2a_conv2d_fwd_p3 */
int v2a_pack_chunk_elems(void);
int v2a_pack_weights_multi(const int64_t* table_dev, const int* chunks_dev, int nchunks, int transposed, v2a_stream_t stream);

/* ---------------------------------------------------------------------------------------------- GroupNorm (csrc/norm.hip)
 * y = film(act(gn(x) + residual)) on [N,S,C]; x2 != NULL: x is channels [0,C1) and x2 channels [C1,C) of a concat.
 * replaces GroupNorm32+SiLU (.../guided_diffusion/nn.py:26-28, unet.py:187-190,211-216,289,628-631), GroupNorm(C/16)+ReLU(+identity)
 * (diffuser/diffusion_policy/model/multi_image_obs_encoder.py:66-74), GroupNorm(8)+Mish+FiLM (conv1d_components.py:23-40,
 * conditional_unet1d.py:46-66).  film [N][2][C] = (scale, shift).  mean / rstd [N*G] are outputs (saved for backward). */
size_t v2a_groupnorm_workspace_bytes(int N, int S, int C, int G);
int v2a_groupnorm_fwd(const float* x, const float* x2, int C1, const float* gamma, const float* beta, const float* residual,
                      const float* film, int film_ld, float* y, float* mean, float* rstd, int N, int S, int C, int G, float eps, int act,
                      void* workspace, size_t workspace_bytes, v2a_stream_t stream);   /* film_ld: floats between samples' FiLM rows (0 = 2*C) */
int v2a_groupnorm_bwd(const float* x, const float* gamma, const float* beta, const float* residual, const float* film, int film_ld,
                      const float* dout, const float* mean, const float* rstd, float* dx, float* dres, float* dfilm, float* colsum,
                      float* dgamma, float* dbeta, int accumulate_params, int N, int S, int C, int G, int act, void* workspace,
                      size_t workspace_bytes, v2a_stream_t stream);
/* the same two operations additionally emitting the bf16 twin of their output (y_h / dx_h, may be NULL): in the bf16-MFMA mode the conv
 * that consumes the GroupNorm output (forward) or its input gradient (backward) reads the twin, which saves the cast launch */
int v2a_groupnorm_fwd_t(const float* x, const float* x2, int C1, const float* gamma, const float* beta, const float* residual,
                        const float* film, int film_ld, float* y, void* y_h, float* mean, float* rstd, int N, int S, int C, int G, float eps,
                        int act, void* workspace, size_t workspace_bytes, v2a_stream_t stream);
int v2a_groupnorm_bwd_t(const float* x, const float* gamma, const float* beta, const float* residual, const float* film, int film_ld,
                        const float* dout, const float* mean, const float* rstd, float* dx, void* dx_h, float* dres, float* dfilm,
                        float* colsum, float* dgamma, float* dbeta, int accumulate_params, int N, int S, int C, int G, int act,
                        void* workspace, size_t workspace_bytes, v2a_stream_t stream);
/* The same two operations taking the normalised tensor (forward: x; backward: dout) as the still-unreduced split-K slabs of the conv
   that produces it: element = sum_s slabs[s][idx] + cbias[c] (forward) / + sresid[idx] (backward).  The conv's reduce launch disappears;
   `x` (forward) / `dout_sum` (backward, optional) receive the finished tensor.  Only for shapes v2a_groupnorm_takes_slabs() accepts
   (one wave per (sample, group): S * C / G <= 1024, C / G in {16, 32, 64, 128}); nslab = 0 behaves like the _t functions. */
/* v2a_groupnorm_fwd whose statistics pass is replaced by the producing convs' epilogue sums (per-64-row blocks, as
   v2a_conv2d_fwd_dma_f32(..., stats, ...) writes them when v2a_conv2d_dma_f32_can_emit_stats says it can): the normalised tensor is
   read once instead of twice (GroupNorm32 of the fp32 sampler, guided_diffusion/nn.py:95-97 after Conv3d nn.py:53-87) */
int v2a_conv2d_dma_f32_can_emit_stats(int M, int Cout, int K);
int v2a_groupnorm_fwd_st(const float* x, const float* x2, int C1, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                         int N, int S, int C, int G, float eps, int act, const float* stats1, const float* stats2, void* workspace,
                         size_t workspace_bytes, v2a_stream_t stream);
int v2a_groupnorm_takes_slabs(int S, int C, int G);
/* v2a_groupnorm_fwd_s also takes a post-activation addend as explicit operands: out = act(gn(x) ...) + post, with `post` dense, or
   `post_slabs` = the still unreduced split-K slabs (post_nslab of them, post_stride floats apart, + post_bias) of the conv that produces
   it -- the residual branch of ConditionalResidualBlock1D (conditional_unet1d.py:46-66: `out = self.blocks[1](out); out = out +
   self.residual_conv(x)`) without an add / reduce launch of its own.  Only where v2a_groupnorm_takes_post says 1 (slabs of 256 / 512 /
   1024 elements, 16 ... 128 channels per group); V2A_ERR_ARG otherwise.  All null / 0: no addend. */
int v2a_groupnorm_takes_post(int S, int C, int G);
/* yh_plane_stride (both functions): 0 = y_h / dx_h is ONE 16-bit twin of the output in the policy's 16-bit format; > 0 = it is the hi plane of
   THREE bf16 planes (hi, mid, lo of the fp32 value, that many elements apart): the pre-split operand of v2a_conv2d_fwd_p3 for the conv that
   consumes this output.  Planes only on the float4 wave path (v2a_groupnorm_takes_post's shapes); V2A_ERR_ARG otherwise. */
int v2a_groupnorm_fwd_s(const float* x, const float* x2, int C1, const float* gamma, const float* beta, const float* residual,
                        const float* film, int film_ld, float* y, void* y_h, size_t yh_plane_stride, float* mean, float* rstd, int N, int S, int C,
                        int G, float eps, int act, const float* slabs, int nslab, size_t slab_stride, const float* cbias, const float* post,
                        const float* post_slabs, int post_nslab, size_t post_stride, const float* post_bias,
                        void* workspace, size_t workspace_bytes, v2a_stream_t s);
int v2a_groupnorm_bwd_s(const float* x, const float* gamma, const float* beta, const float* residual, const float* film, int film_ld,
                        const float* dout, const float* mean, const float* rstd, float* dx, void* dx_h, size_t yh_plane_stride, float* dres, float* dfilm,
                        float* colsum, float* dgamma, float* dbeta, int accumulate_params, int N, int S, int C, int G, int act,
                        const float* slabs, int nslab, size_t slab_stride, const float* sresid, float* dout_sum,
                        void* workspace, size_t workspace_bytes, v2a_stream_t s);
/* dgamma / dbeta of many GroupNorm layers in ONE launch (pass dgamma = dbeta = NULL to v2a_groupnorm_bwd and keep its colsum):
   table [nrows][5] int64 rows {colsum ptr, dgamma ptr, dbeta ptr, N, C}; work [nwork][2] int32 = (row, 64-channel block) */
int v2a_gn_param_grads_multi(const void* table, const void* work, int nwork, v2a_stream_t s);

/* ---------------------------------------------------------------------------------------------- elementwise (csrc/elementwise.hip) */
/* backward helpers of the video UNet: 2x2 sum pooling (gradient of the folded nearest upsample, unet.py:86-115) and per-sample column
 * sums (gradient of the embedding row vector broadcast over a sample's rows, unet.py:239-260) */
int v2a_sumpool2x2(const float* du, float* dx, int N, int H, int W, int C, v2a_stream_t s);
size_t v2a_colsum_batched_workspace_bytes(int B, int rows, int C);
int v2a_colsum_batched(const float* x, float* out, int B, int rows, int C, int accumulate, void* workspace, size_t workspace_bytes, v2a_stream_t s);
int v2a_act_fwd(const float* x, float* y, size_t n, int act, v2a_stream_t s);                 /* nn.Mish / nn.SiLU / nn.GELU */
int v2a_act_bwd(const float* x, const float* dy, float* dx, size_t n, int act, v2a_stream_t s);
int v2a_axpy(const float* a, const float* b, float* out, float alpha, size_t n, v2a_stream_t s);   /* out = a + alpha b */
int v2a_scale_by_device_scalar(float* x, size_t n, const float* scalar, v2a_stream_t s);        /* x *= *scalar (loss scale) */
int v2a_copy2d(const float* src, float* dst, int rows, int cols, int ld_src, int ld_dst, int accumulate, v2a_stream_t s); /* torch.cat / slicing */
int v2a_colsum(const float* x, float* out, int rows, int cols, int accumulate, v2a_stream_t s);   /* bias gradients */
/* kind 0: SinusoidalPosEmb (diffuser/diffusion_policy/model/positional_embedding.py:10-17); kind 1: timestep_embedding (.../nn.py:171-189) */
int v2a_sincos_embed(const int64_t* t, float* out, int B, int dim, int kind, v2a_stream_t s);
/* normalise actions + DDPMScheduler.add_noise (diffusion_unet_image_policy.py:255; normalizer.py:139-146) */
/* act_min / act_max [act_dim]: the action limits of shape_meta (lb_train_diffusion_unet_image_orn10.yaml:27); NULL = -1 / +1 */
int v2a_add_noise(const float* act, const float* noise, const int64_t* t, const float* alphas_cumprod, float* out, int B, int per,
                  const float* act_min, const float* act_max, int act_dim, v2a_stream_t s);
/* F.mse_loss(...).mean() and its gradient (diffusion_unet_image_policy.py:273-276) */
int v2a_mse_loss(const float* pred, const float* target, float* loss, float* dpred, int n, v2a_stream_t s);
/* the same with the loss GRADIENT multiplied by *grad_scale_dev (device float, may be null): the dynamic loss scale of the fp16 policy mode
 * (GradScaler's scale; lb_online_trainer_v7.py:72-76,604) -- the returned loss is unscaled */
int v2a_mse_loss_scaled(const float* pred, const float* target, float* loss, float* dpred, int n, const float* grad_scale_dev, v2a_stream_t s);
/* DDPMScheduler.step (mode 0) / DDIMScheduler.step (mode 1) on the action trajectory (diffusion_unet_image_policy.py:121-128) */
int v2a_policy_sched_step(const float* eps, const float* sample, const float* noise, float* out, int n, float c_sb, float c_sa,
                          float c0, float c1, float sigma, int mode, v2a_stream_t s);
/* LimitsConstNormalizer.unnormalize with the policy's action limits (normalizer.py:148-161); NULL limits = -1 / +1 */
int v2a_unnormalize_action(const float* x, float* out, int n, const float* act_min, const float* act_max, int act_dim, v2a_stream_t s);
/* NCHW -> NHWC (+ 2x-1 image normalisation, normalizer.py:139-146; uint8 / 255, diffuser/datasets/img_utils.py:27-37) */
int v2a_nchw_to_nhwc_f32(const float* src, float* dst, int N, int C, int HW, int normalize, v2a_stream_t s);
int v2a_nchw_to_nhwc_u8(const uint8_t* src, float* dst, int N, int C, int HW, int normalize, v2a_stream_t s);
int v2a_nhwc_to_nchw_f32(const float* src, float* dst, int N, int C, int HW, v2a_stream_t s);
/* NCHW RGB (float or uint8) -> the interior of a zero-bordered [N][H + 2 pad][W + 2 pad][4] image (channel 3 = 0), same normalisation:
   the input layout of v2a_conv2d_fwd_window_f32 and of the stem's weight gradient.  The caller zeroes dst once. */
int v2a_nchw_to_nhwc4p(const void* src, int is_u8, float* dst, int N, int H, int W, int pad, int normalize, v2a_stream_t s);
/* Unet_Libero input pack 'b (f c) h w' + repeated cond image -> [B,f,H,W,6] (flowdiffusion/flowdiffusion/unet.py:217-220) */
int v2a_video_pack(const float* img, const float* cond, float* xin, int B, int f, int HW, size_t img_bstride, size_t cond_bstride,
                   int frame_ch /* 3 RGB (Unet_Libero/MW/Thor/Bridge), 2 flow (UnetMWFlow) */, v2a_stream_t s);
/* one fused sampler step: v-pred -> x0 (clamp) -> posterior mean + sigma*noise (mode 0, goal_diffusion.py:561-580), DDIM update (mode 1,
 * :617-634), last DDIM pair (mode 2, :619-622); gw > 0 = classifier-free guidance mix (:536-547); final = unnormalize + clamp (:640,:650) */
int v2a_video_denoise_step(const float* v, const float* v_uncond, const float* img, const float* noise, float* out, int B, int f, int HW,
                           float sa, float s1, float ra, float rm, float c1, float c2, float sigma, float gw, int mode, int final,
                           int frame_ch, v2a_stream_t s);
/* Table-driven sampler step for ALL three objectives (goal_diffusion.py:499-559 model_predictions; objective 0 pred_noise, 1 pred_x0,
 * 2 pred_v) with the step index read from device memory, so one captured hipGraph {UNet forward, this step, v2a_video_sampler_advance}
 * is replayed for every step of p_sample_loop / ddim_sample (:582-641).  table_dev: rows of v2a_video_denoise_row_bytes() bytes =
 * {sa, s1, ra, rm, c1, c2, sigma, gw: float; mode, final, t, pad: int32} (coefficient meaning as v2a_video_denoise_step).  state_dev:
 * uint64[3] = {current row, Philox seed, Philox counter of the initial image} or NULL (row step_imm).  noise: explicit tensor, or NULL
 * with use_philox bit 0 set: drawn in the kernel (the values v2a_philox_normal(seed, state[2] + (row + 1) * ceil(total / 4)) would write),
 * not drawn at all where sigma = 0.  use_philox bit 1: the table holds guided rows (gw > 0) -- v_uncond is then mandatory (V2A_ERR_ARG
 * when null; without the bit the kernel never reads v_uncond).  `out` may alias `img`. */
int v2a_video_denoise_row_bytes(void);
int v2a_video_denoise_step2(const float* v, const float* v_uncond, const float* img, const float* noise, float* out, int B, int f, int HW,
                            int frame_ch, int objective, const void* table_dev, const uint64_t* state_dev, int step_imm, int use_philox,
                            const uint64_t* row_seeds_dev, v2a_stream_t s);
/* row_seeds_dev (optional, uint64 [B]): one Philox seed per sample instead of state[1] -- row b then draws exactly what a one-row call
   with seed row_seeds_dev[b] draws, so a batched GoalGaussianDiffusion.sample reproduces its rows sampled one at a time (the
   exploration round of lb_online_trainer_v7.py:866-891 as ONE call); v2a_philox_normal_rows draws the initial image the same way. */
int v2a_philox_normal_rows(float* out, int rows, size_t row_elems, const uint64_t* seeds_dev, uint64_t offset_imm, v2a_stream_t s);
/* state_dev[0] += 1; tt[0..B) = t of the new row (the time step the next UNet forward embeds) */
int v2a_video_sampler_advance(uint64_t* state_dev, const void* table_dev, int64_t* tt, int B, int nrows, v2a_stream_t s);
/* out_i = x W_i^T + bias_i for n <= v2a_emb_linear_multi_max() weight matrices sharing x [B <= 16][K]: the per-ResBlock `emb_layers`
 * Linears of the video UNet (guided_diffusion/guided_diffusion/unet.py:204-210,248-257) in one launch.  w / bias / out / couts: HOST arrays. */
int v2a_emb_linear_multi_max(void);
int v2a_emb_linear_multi(const float* x, int B, int K, const float* const* w, const float* const* bias, float* const* out, const int* couts,
                         int n, v2a_stream_t s);
/* Transformer policy backbone (flowdiffusion/flowdiffusion/diffusion_policy_baseline/transformer_for_diffusion.py:75-110, the
 * torch.nn.MultiheadAttention inside its encoder / decoder layers): softmax(q k^T / sqrt(D) + mask) v per (batch, head); mask = additive
 * float [Tq][Tk] (-inf blocks) or NULL.  q / k / v are column blocks of packed projections: row r of batch b at ptr + (b*T + r)*ld + h*D.
 * The backward writes dq / dk / dv with the same leading dimensions. */
int v2a_mha_fwd(const float* q, const float* k, const float* v, const float* mask, float* out, int B, int Tq, int Tk, int H, int D, int ldq,
                int ldk, int ldv, float p_drop, uint64_t seed, uint64_t stream_id, v2a_stream_t s);
int v2a_mha_bwd(const float* q, const float* k, const float* v, const float* mask, const float* dout, float* dq, float* dk, float* dv, int B,
                int Tq, int Tk, int H, int D, int ldq, int ldk, int ldv, float p_drop, uint64_t seed, uint64_t stream_id, v2a_stream_t s);
/* nn.Dropout of the same backbone (p_drop_emb / p_drop_attn, transformer_for_diffusion.py:61,80,98): y = x * keep / (1 - p) with a stateless
 * mask -- element i of random stream `stream_id` under `seed` is kept iff hash(seed, stream_id, i) >= p; the same call on dy is the backward.
 * The attention kernels above apply the same rule to the probabilities (element index ((b*H + h)*Tq + i)*Tk + j). */
int v2a_dropout(const float* x, float* y, size_t n, float p, uint64_t seed, uint64_t stream_id, v2a_stream_t s);
/* video-model training (GoalGaussianDiffusion.forward / p_losses, flowdiffusion/flowdiffusion/goal_diffusion.py:674-724):
 * q_sample (:674-680) with the [0,1] -> [-1,1] normalisation of forward (:722) folded in; per-sample mean of l2 / l1 (objective 0 =
 * pred_noise, 1 = pred_x0, 2 = pred_v, :699-707) times loss_weight[t] then the batch mean (:709-713); and its gradient with respect to
 * the model output, scaled by the device scalar `gscale` (NULL = 1).  out_cl / dout are channels-last [B,f,HW,ci]; img / noise keep the
 * reference's 'b (f c) h w' layout. */
int v2a_video_qsample(const float* img, const float* noise, const int64_t* t, const float* sqrt_acp, const float* sqrt_1m_acp, float* out, int B,
                      size_t per, int normalize, v2a_stream_t s);
size_t v2a_video_loss_workspace_bytes(int B);
int v2a_video_loss_fwd(const float* out_cl, const float* img, const float* noise, const int64_t* t, const float* sqrt_acp,
                       const float* sqrt_1m_acp, const float* loss_weight, float* loss, int B, int f, int HW, int ci, int objective, int l1,
                       int normalize, void* ws, size_t ws_bytes, v2a_stream_t s);
int v2a_video_loss_bwd(const float* out_cl, const float* img, const float* noise, const int64_t* t, const float* sqrt_acp,
                       const float* sqrt_1m_acp, const float* loss_weight, const float* gscale, float* dout, int B, int f, int HW, int ci,
                       int objective, int l1, int normalize, v2a_stream_t s);
/* counter-based Philox4x32-10 generators (replace torch.randn / torch.randint of compute_loss :246-252 in perf runs) */
int v2a_philox_normal(float* out, size_t n, uint64_t seed, const uint64_t* offset_dev, uint64_t offset_imm, v2a_stream_t s);
int v2a_philox_randint(int64_t* out, int n, int high, uint64_t seed, const uint64_t* offset_dev, uint64_t offset_imm, v2a_stream_t s);
int v2a_advance_counter(uint64_t* ctr, uint64_t inc, v2a_stream_t s);
/* fp32 convs over PRE-SPLIT operands (round 5): x3 / x2_3 / w3 point at the hi plane of three bf16 planes (hi = bf16(v), mid = bf16(v - hi),
 * lo = bf16(v - hi - mid); `*_plane_stride` elements apart) of x [N,H,W,C1] / x2 [N,H,W,C2] / the packed weight [Cout][KH][KW][C1+C2].  The
 * arithmetic, split plan, workspace (v2a_conv2d_dma_f32_workspace_bytes) and results -- bit for bit -- of v2a_conv2d_fwd_dma_f32_d in its
 * three-plane mode, without the per-tile re-splitting: the ConditionalUnet1D's small-M / deep-K Conv1d GEMMs
 * (diffuser/diffusion_policy/model/conditional_unet1d.py:46-66, conv1d_components.py:23-40).  Planes are written by v2a_groupnorm_fwd_s /
 * _bwd_s (yh_plane_stride), v2a_opt_step_packed (twin format 2), v2a_pack_weights_multi (mode | 256) and v2a_split3_f32.  nslab_out as
 * v2a_conv2d_fwd_dma_f32_d (null: the reduce launch runs inside).  Stateless; thread-safe for distinct streams and workspaces. */
int v2a_conv2d_p3_eligible(int M, int Cout, int K, int C1, int C2);
int v2a_conv2d_fwd_p3(const void* x3, size_t x_plane_stride, const void* x2_3, size_t x2_plane_stride, const void* w3, size_t w_plane_stride,
                      const float* bias, const float* residual, float* y, const void* zeros, int N, int H, int W, int C1, int C2, int Cout,
                      int KH, int KW, int sh, int sw, int ph, int pw, int OH, int OW, int* nslab_out, void* workspace, size_t workspace_bytes,
                      v2a_stream_t stream);
int v2a_split3_f32(const float* x, void* y3, size_t n, size_t plane_stride, v2a_stream_t stream);   /* fp32 [n] -> three bf16 planes (n % 4 == 0) */
/* measurement / test hook: 0 = a zero-interleaved input (idil = 2: data gradient of a stride-2 conv, transposed conv) multiplies every
   filter tap as in round 5; 1 (default) = class-major tile rows whose K loop walks the live taps only (2.25 of 9 for a 3 x 3 filter).
   Returns the old value.  Process-wide. */
int v2a_debug_set_parity_classes(int on);
/* 3 x 3 / stride 1 / pad 1 convs over small square maps (32 / 16 / 8 / 4: the policy's ResNet-18 encoders, torchvision BasicBlock as built by
   diffusion_policy/model/vision/model_getter.py + multi_image_obs_encoder.py): 1 when v2a_conv2d_fwd_dma_f32 / _d run the problem on
   conv_maps_x3 (256-row tiles, phases of 36 MFMAs; csrc/igemm_x3m.hip).  v2a_debug_set_maps_kernel(0) keeps them on conv_halo_x3 (the
   round-5 form; measurement / test hook, returns the old value, process-wide). */
/* v2a_conv2d_fwd runs reductions of at most 64 values that its vector loader cannot take (ConditionalUnet1D's layers over the 7 action
   channels: model/conditional_unet1d.py:137-160,198-201) on a direct kernel (conv_smallk, csrc/igemm.hip).  v2a_debug_set_smallk(0) sends
   them back to the tile kernels (measurement / test hook; returns the old value, a negative argument only queries; process-wide). */
int v2a_debug_set_smallk(int on);
int v2a_conv2d_x3m_eligible(int N, int S, int C, int Cout);
int v2a_debug_set_maps_kernel(int on);
int v2a_set_f32_conv_mode(int x3);   /* fp32 convs: 1 = three-bf16-plane products (fp32-equivalent accuracy, default), 0 = exact-f32 MFMA; returns the old value */
int v2a_get_f32_conv_mode(void);
int v2a_debug_timestamp(uint64_t* dst, v2a_stream_t s);   /* measurement aid: *dst = constant-rate wall clock (100 MHz) when the stream gets here */

/* ---------------------------------------------------------------------------------------------- attention (csrc/attention.hip) */
/* QKVAttentionLegacy.forward (.../guided_diffusion/unet.py:341-358): qkv [n_frames*L][heads*3*ch] -> out [n_frames*L][heads*ch] */
int v2a_attention_fwd(const float* qkv, float* out, int n_frames, int L, int heads, int head_ch, v2a_stream_t s);
/* its backward (video-model training: the reference differentiates AttentionBlock through autograd, unet.py:303-358): d(qkv) from qkv,
 * the forward output and d(output); fp32, deterministic */
int v2a_attention_bwd(const float* qkv, const float* out, const float* dout, float* dqkv, int n_frames, int L, int heads, int head_ch,
                      v2a_stream_t s);
/* PerceiverAttention core (.../guided_diffusion/imagen.py:295-319): l2norm(q,k) * scales, sim * 8, softmax, @ v */
int v2a_perceiver_attention(const float* q, const float* kv, const float* q_scale, const float* k_scale, float* out, int B, int Lq,
                            int Lk, int H, int D, float sim_scale, v2a_stream_t s);
int v2a_layernorm(const float* x, const float* g, const float* b, float* y, int rows, int D, float eps, v2a_stream_t s); /* imagen.py:198-211 */
/* backward passes of the text branch (video-model training; the reference differentiates PerceiverResampler with autograd,
 * imagen.py:254-372): perceiver attention (dq, dkv, per-(b,head) scale gradients dscale [B*H][2][D]), LayerNorm (dx + per-row
 * [rows][2][D] contributions to dg / db), and the broadcast that is the gradient of a mean over rows */
int v2a_perceiver_attention_bwd(const float* q, const float* kv, const float* q_scale, const float* k_scale, const float* out,
                                const float* dout, float* dq, float* dkv, float* dscale, int B, int Lq, int Lk, int H, int D,
                                float sim_scale, v2a_stream_t s);
int v2a_layernorm_bwd(const float* x, const float* g, const float* dy, float* dx, float* contrib, int rows, int D, float eps, v2a_stream_t s);
int v2a_bcast_rows(const float* dout, float* dx, int B, int R, int D, float scale, v2a_stream_t s);
int v2a_mean_rows(const float* x, float* out, int B, int R, int D, v2a_stream_t s);

/* ---------------------------------------------------------------------------------------------- pooling (csrc/pool.hip) */
int v2a_maxpool3x3s2_fwd(const float* x, float* y, int8_t* idx, int N, int H, int W, int C, v2a_stream_t s);   /* resnet18.maxpool */
int v2a_maxpool3x3s2_bwd(const float* dy, const int8_t* idx, float* dx, int N, int H, int W, int C, v2a_stream_t s);
/* SpatialSoftmax.forward (diffuser/diffusion_policy/common/base_nets.py:234-285): feat [B,H,W,K] -> kp [B,K,2]; att saved */
int v2a_spatial_softmax_fwd(const float* feat, float* kp, float* att, int B, int H, int W, int K, v2a_stream_t s);
int v2a_spatial_softmax_bwd(const float* att, const float* kp, const float* dkp, float* dfeat, int B, int H, int W, int K, v2a_stream_t s);

/* 16-bit format of every `_h` entry point (csrc/igemm_h.hip, igemm_h2.hip, igemm_h3.hip, norm_h.hip, v2a_attention_fwd_h, the f32 <-> 16-bit
 * casts and packs): 0 = bf16 (default), 1 = IEEE fp16 -- the reference's own 16-bit type (fp16 autocast: lb_online_trainer_v7.py:72-76,889).
 * Same kernels, instantiated with v_mfma_f32_32x32x16_f16 / v_cvt_f16_f32.  PER HOST THREAD (thread_local): the flag is the format of the
 * 16-bit tensors the calling thread hands to its next `_h` launches (the Python wrappers set it from the tensor dtype before every call), so
 * two threads driving bf16 and fp16 tensors do not disturb each other.  Returns the thread's previous value. */
int v2a_set_half_format(int f16);
int v2a_get_half_format(void);

/* ------------------------------------------------------------------------------------- bf16-storage convolution (csrc/igemm_h.hip)
 * The reference's GPU path runs the video UNet under fp16 autocast (diffuser/libero/lb_online_trainer_v7.py:889,
 * guided_diffusion/guided_diffusion/nn.py:53-87 Conv3d); this is that configuration on gfx950: activations and packed weights are
 * bf16 in HBM, accumulation fp32, epilogue (bias + embedding row vector + residual) in fp32 registers.
 * x / x2 / residual / y: bf16 channels-last; w_packed: bf16 [Cout][KH][KW][C1+C2] (v2a_pack_weight_h); bias / rowvec: fp32;
 * exactly one of y (bf16) / y_f32 non-null; zeros: >= 128 zero bytes (padding taps read it).  C1 % 64 == C2 % 64 == 0. */
size_t v2a_conv2d_h_workspace_bytes(int M, int Cout, int K);
int v2a_conv2d_fwd_h(const void* x, const void* x2, const void* w_packed, const float* bias, const float* rowvec, const void* residual,
                     const float* residual_f32, void* y, float* y_f32, const void* zeros, int N, int H, int W, int C1, int C2, int Cout,
                     int KH, int KW, int sh, int sw, int ph, int pw, int ups, int idil, int OH, int OW, int rows_per_batch,
                     float* stats, void* workspace, size_t workspace_bytes, v2a_stream_t s);
/* the same LDS-DMA kernel over fp32 tensors with the exact-f32 MFMA: parity-configuration conv for channel counts that are multiples of
 * 32 (replaces v2a_conv2d_fwd for those layers; w_packed = the fp32 forward pack [Cout][KH][KW][C1+C2]) */
size_t v2a_conv2d_dma_f32_workspace_bytes(int M, int Cout, int K);
int v2a_conv2d_h_can_emit_stats(int M, int Cout, int K);   /* 1: v2a_conv2d_fwd_h takes a stats buffer for this problem size */
int v2a_conv2d_fwd_dma_f32(const float* x, const float* x2, const float* w_packed, const float* bias, const float* rowvec,
                           const float* residual, float* y, const void* zeros, int N, int H, int W, int C1, int C2, int Cout, int KH, int KW,
                           int sh, int sw, int ph, int pw, int ups, int idil, int OH, int OW, int rows_per_batch, float* stats,
                           void* workspace, size_t workspace_bytes, v2a_stream_t s);
/* 1 when v2a_conv2d_fwd_dma_f32 runs this temporal (3 x 1, stride 1, pad (1, 0)) conv over x [B, F, HW, C] -- the `temporal_conv` of the
   factorised Conv3d, guided_diffusion/nn.py:53-87 -- on the frame-stack three-plane kernel (csrc/igemm_x3t.hip conv_frames_x3: all F = 7
   frames of 64 pixels x 128 output channels per workgroup, the operands split into bf16 planes once for the three taps; three-plane mode
   only).  A pure function of the shape. */
int v2a_conv2d_x3t_eligible(int B, int F, int HW, int C, int Cout, int rows_per_batch, int has_rowvec);
/* 1 when v2a_conv2d_fwd_dma_f32 runs this 3 x 3 / stride 1 / pad 1 conv over an H x W map (H, W: the conv's map, i.e. twice the source's
   behind an upsample) -- the `spatial_conv` of the factorised Conv3d -- on the phase-structured patch kernel (csrc/igemm_x3p.hip
   conv_patch_x3: a 16 x 16 pixel patch x 128 output channels per persistent workgroup; three-plane mode, no row vector / statistics).
   Depends on the shape and the device's CU count only. */
int v2a_conv2d_x3p_eligible(int N, int H, int W, int C, int Cout);
/* Upsample (nearest x2, guided_diffusion/unet.py:105-115) + 3 x 3 conv in the fp32 configuration as FOUR 2 x 2 convs over the source map, one
   per parity class of the output pixel: two of the three filter rows / columns read the same source pixel, so their weights are summed
   once (v2a_pack_weight_ups4: forward pack [Cout][3][3][C] -> [4][Cout][2][2][C]) and every output needs 4 of the 9 products
   (conv_patch_x3<.., 2>, csrc/igemm_x3p.hip).  v2a_conv2d_fwd_x3p_ups4: x = the SOURCE [N, H/2, W/2, C], y [N, H, W, Cout], bias only;
   where v2a_conv2d_x3p_ups4_eligible(N, H, W, C, Cout) (H, W the OUTPUT map, multiples of 32). */
int v2a_conv2d_x3p_ups4_eligible(int N, int H, int W, int C, int Cout);
/* the same decomposition for the 16-bit video UNet (csrc/igemm_hp.hip conv_patch_h_ups4: 32 x 16 class pixels x 128 channels per persistent
   workgroup, register-staged operands; format = v2a_set_half_format of the calling thread): x = the SOURCE [N, H/2, W/2, C] 16-bit,
   w_ups4 = [4][Cout][2][2][C] 16-bit (v2a_pack_weight_ups4 of the fp32 pack, then v2a_cast_f32_h), y [N, H, W, Cout] 16-bit, bias fp32;
   where v2a_conv2d_hp_ups4_eligible(N, H, W, C, Cout) (H % 64 == 0, W % 32 == 0: the OUTPUT map). */
int v2a_conv2d_hp_ups4_eligible(int N, int H, int W, int C, int Cout);
int v2a_conv2d_fwd_hp_ups4(const void* x, const void* w_ups4, const float* bias, void* y, const void* zeros, int N, int H, int W, int C,
                           int Cout, v2a_stream_t stream);
int v2a_pack_weight_ups4(const float* w_packed, float* out, int Cout, int C, v2a_stream_t stream);
int v2a_conv2d_fwd_x3p_ups4(const float* x, const float* w_ups4, const float* bias, float* y, const void* zeros, int N, int H, int W, int C,
                            int Cout, v2a_stream_t stream);
/* GroupNorm32 + SiLU in front of a ResBlock conv (guided_diffusion/unet.py:181-197 in_layers / out_layers, nn.py:95-97) folded into that
   conv in the fp32 configuration: v2a_groupnorm_stats_f32 turns the producing conv's per-64-row statistics blocks `stats`
   [N * S/64][2][C] into mean / rstd [N][G] (the reduce + finalise launches of v2a_groupnorm_fwd_st, S % 64 == 0, workspace >= N*64*2*C*8
   bytes); v2a_conv2d_fwd_x3p_gn is conv_patch_x3 reading act((x - mean) * rstd * gamma + beta) -- the arithmetic of the apply pass it
   replaces, done on the halo in registers; x [N images, H, W, C], gn_frames images per GroupNorm sample, act 0 / 1 (SiLU).  3 x 3 /
   stride 1 / pad 1, bias only; where v2a_conv2d_x3p_eligible(N, H, W, C, Cout) and (C / G) % 4 == 0. */
int v2a_groupnorm_stats_f32(const float* stats, float* mean, float* rstd, int N, int S, int C, int G, float eps, void* workspace,
                            size_t workspace_bytes, v2a_stream_t stream);
int v2a_conv2d_fwd_x3p_gn(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int G, int gn_frames,
                          int act, const float* w_packed, const float* bias, float* y, const void* zeros, int N, int H, int W, int C, int Cout,
                          v2a_stream_t stream);
/* "channel window" form for few-channel inputs (the RGB stem of the policy's ResNet-18 encoders: torchvision resnet18.conv1 7x7 / 2 behind
   diffuser/diffusion_policy/common/vision_nets.py:29-39): pixel (ih, iw) = the C floats at x + ((n * H + ih) * W + iw) * xpitch, xpitch <= C
   (overlapping windows); no padding (the buffer carries its zero border, v2a_nchw_to_nhwc4p); w_packed [Cout][KH][KW][C] (pack mode 2 of
   v2a_pack_weights_multi for the 7x7x3 filter -> [Cout][7][8][4]).  fp32 three-plane conv mode only (V2A_ERR_ARG otherwise). */
int v2a_conv2d_fwd_window_f32(const float* x, const float* w_packed, const float* bias, float* y, const void* zeros, int N, int H, int W,
                              int xpitch, int C, int Cout, int KH, int KW, int sh, int sw, int OH, int OW, void* workspace,
                              size_t workspace_bytes, v2a_stream_t stream);
/* same, leaving the split-K reduce to the consuming GroupNorm launch (v2a_groupnorm_fwd_s / _bwd_s): *nslab_out (HOST) = number of
   fp32 slabs [M][Cout] left in `workspace` (bias / residual not applied, y untouched), or 0 when the conv finished y itself */
int v2a_conv2d_fwd_dma_f32_d(const float* x, const float* x2, const float* w_packed, const float* bias, const float* rowvec,
                             const float* residual, float* y, const void* zeros, int N, int H, int W, int C1, int C2, int Cout, int KH, int KW,
                             int sh, int sw, int ph, int pw, int ups, int idil, int OH, int OW, int rows_per_batch, int* nslab_out,
                             void* workspace, size_t workspace_bytes, v2a_stream_t stream);
/* bf16-operand variant (fp32 output): only for problems whose plan splits K (v2a_conv2d_h_splits() > 1); slabs hold neither bias nor
   the fp32 residual -- the consumer adds both */
int v2a_conv2d_h_splits(int M, int Cout, int K);
int v2a_conv2d_fwd_h_d(const void* x, const void* x2, const void* w_packed, const float* bias, const void* reserved,
                       const float* residual_f32, float* y_f32, const void* zeros, int N, int H, int W, int C1, int C2, int Cout, int KH,
                       int KW, int sh, int sw, int ph, int pw, int ups, int idil, int OH, int OW, int rows_per_batch, int* nslab_out,
                       void* workspace, size_t workspace_bytes, v2a_stream_t stream);
/* residual (bf16) xor residual_f32; idil 1 | 2; stats (optional, only when v2a_conv2d_h_workspace_bytes() == 0 and y is bf16):
 * [ceil(M/64)][2][Cout] per-64-row sum / sum of squares of the rounded outputs, consumed by v2a_groupnorm_fwd_h */
/* multi-stage 256-row variant of v2a_conv2d_fwd_h for the large layers (csrc/igemm_h2.hip: 8 waves, 4-5 LDS stages, counted vmcnt):
 * bf16 in / bf16 out, optional bf16 residual and statistics, no split-K; v2a_conv2d_h2_eligible says whether a problem qualifies */
/* Halo-tile 3x3 / stride 1 / pad 1 convolution of the bf16-storage video UNet (csrc/igemm_h3.hip): the 18 x 18 input halo of a
   16 x 16 output patch is DMA-ed once per 32-channel chunk and shared by the nine taps (Conv3d spatial part, nn.py:45,64-69) */
int v2a_conv2d_h3_eligible(int N, int H, int W, int C, int Cout, int KH, int KW, int sh, int sw, int ph, int pw, int ups, int C2);
int v2a_conv2d_fwd_h3(const void* x, const void* w_packed, const float* bias, const float* rowvec, const void* residual, void* y,
                      const void* zeros, int N, int H, int W, int C, int Cout, int ups, int rows_per_batch, float* stats, v2a_stream_t stream);
/* GroupNorm folded into the consuming 3x3 conv (bf16-storage sampler: GroupNorm32 + SiLU in front of every Conv3d of a ResBlock,
   guided_diffusion/unet.py:180-260): v2a_groupnorm_prep_h turns the statistics into a per-(sample, channel) scale / shift table
   ab [N][2][C]; v2a_conv2d_fwd_h3_gn applies y = act(x * a + b) to its input halo while it sits in LDS (x [N,H,W,C1] | x2 [N,H,W,C-C1]
   along channels; gn_frames images per GroupNorm sample; act 0 = none, 1 = SiLU), so the normalised tensor is never written;
   v2a_groupnorm_apply_h is the stand-alone apply pass for consumers that cannot (same arithmetic) */
int v2a_groupnorm_prep_h(const void* x, const void* x2, int C1, const float* gamma, const float* beta, float* mean, float* rstd,
                         const float* stats1, const float* stats2, int N, int S, int C, int G, float eps, float* ab_out, void* workspace,
                         size_t workspace_bytes, v2a_stream_t stream);
int v2a_groupnorm_apply_h(const void* x, const void* x2, int C1, const float* ab, void* y, int N, int S, int C, int act, v2a_stream_t stream);
int v2a_conv2d_fwd_h3_gn(const void* x, const void* x2, int C1, const float* gn_ab, int gn_frames, int act, const void* w_packed,
                         const float* bias, const float* rowvec, const void* residual, void* y, const void* zeros, int N, int H, int W, int C,
                         int Cout, int rows_per_batch, float* stats, v2a_stream_t stream);
/* Temporal (3 x 1 x 1) part of the factorised Conv3d (nn.py:45-69: temporal_conv) on the frame-stack tile (csrc/igemm_h3.hip,
   conv_frames_h3): all F = 7 frames of 64 pixels in one workgroup, the input DMA-ed once per 32-channel chunk for the three taps.
   Same argument meaning as v2a_conv2d_fwd_h3 with x viewed as [B, F, HW, C] */
int v2a_conv2d_t3_eligible(int N, int H, int W, int C, int Cout, int KH, int KW, int sh, int sw, int ph, int pw, int ups, int C2);
int v2a_conv2d_fwd_t3(const void* x, const void* w_packed, const float* bias, const float* rowvec, const void* residual, void* y,
                      const void* zeros, int B, int F, int HW, int C, int Cout, int rows_per_batch, float* stats, v2a_stream_t stream);
int v2a_conv2d_h2_eligible(int M, int Cout, int K, int C1, int C2);
int v2a_conv2d_fwd_h2(const void* x, const void* x2, const void* w_packed, const float* bias, const float* rowvec, const void* residual,
                      void* y, const void* zeros, int N, int H, int W, int C1, int C2, int Cout, int KH, int KW, int sh, int sw, int ph,
                      int pw, int ups, int OH, int OW, int rows_per_batch, float* stats, v2a_stream_t s);
/* GroupNorm + activation and QKV attention over bf16 tensors (csrc/norm_h.hip, csrc/attention.hip): same math as v2a_groupnorm_fwd /
 * v2a_attention_fwd (reference nn.py:26-28 GroupNorm32 computes in fp32 and returns the input dtype; unet.py:341-358), bf16 I/O. */
size_t v2a_groupnorm_h_workspace_bytes(int N, int S, int C);
int v2a_groupnorm_fwd_h(const void* x, const void* x2, int C1, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                        const float* stats1, const float* stats2, int N, int S, int C, int G, float eps, int act, void* workspace,
                        size_t workspace_bytes, v2a_stream_t s);
int v2a_attention_fwd_h(const void* qkv, void* out, int n_frames, int L, int heads, int head_ch, v2a_stream_t s);
int v2a_pack_weight_h(const float* w, void* out, int Cout, int Cin, int taps, v2a_stream_t s);   /* [Cout][Cin][taps] f32 -> [Cout][taps][Cin] bf16 */
int v2a_cast_f32_bf16(const float* x, void* y, size_t n, v2a_stream_t s);   /* always bf16 (independent of v2a_set_half_format) */
int v2a_cast_f32_h(const float* x, void* y, size_t n, int f16, v2a_stream_t s);   /* explicit 16-bit format: 0 bf16, 1 IEEE fp16 */
/* fp32 [M][Cin] -> bf16 [M][Cpad] with zero channels behind Cin: the 6-channel stem input of Unet_Libero (unet.py:195-222) padded to
   one 32-channel chunk of the bf16 conv kernels */
int v2a_pad_cast_f32_bf16(const float* x, void* y, size_t M, int Cin, int Cpad, v2a_stream_t s);
int v2a_cast_bf16_f32(const void* x, float* y, size_t n, v2a_stream_t s);   /* always bf16 */
int v2a_cast_h_f32(const void* x, float* y, size_t n, int f16, v2a_stream_t s);

/* ---------------------------------------------------------------------------------------------- optimiser (csrc/optim.hip)
 * clip_grad_norm_(1.0) -> AdamW.step -> zero_grad -> EMA.update  (diffuser/libero/lb_online_trainer_v7.py:604-624;
 * hyper-parameters config/libero/lb_tk8_65to72.py:138-153).  table_dev: int64 [n_tensors][6] = {p, g, m, v, ema, numel};
 * chunks_dev: int32 [nchunks][2] = {tensor, start} with chunks of v2a_opt_chunk_elems(); state_dev: v2a_opt_state_bytes() bytes
 * initialised from the HOST image filled by v2a_opt_state_init; partial_dev: nchunks doubles. */
int v2a_opt_chunk_elems(void);
size_t v2a_opt_state_bytes(void);
int v2a_opt_state_init(void* host_state, double lr, double b1, double b2, double eps, double wd, double max_norm, double ema_inv_gamma,
                       double ema_power, double ema_min, double ema_beta, int ema_update_after, int ema_update_every);
int v2a_opt_state_peek(const void* host_state, float* grad_norm, float* clip_coef, long long* step, float* ema_decay);
/* step counters of a HOST copy of the state, for `trainer.save/load` (lb_online_trainer_v7.py:367-408); lr <= 0 keeps the rate */
int v2a_opt_state_counters(const void* host_state, long long* step, long long* ema_step, int* ema_initted);
int v2a_opt_state_set_counters(void* host_state, long long step, long long ema_step, int ema_initted, double lr);
/* dynamic loss scaling = torch.cuda.amp.GradScaler's contract inside the fused tail (reference: accelerate fp16 mixed precision,
 * lb_online_trainer_v7.py:72-76,604-612): the gradients arrive multiplied by loss_scale; v2a_opt_step unscales inside the clip factor,
 * skips the parameter / moment update (EMA and zero_grad still run) when their norm is inf / nan, halves the scale then, and grows it by
 * growth_factor after growth_interval clean steps.  init_scale <= 0: off (default). */
int v2a_opt_state_set_scaler(void* host_state, double init_scale, double growth_factor, double backoff_factor, int growth_interval,
                             int growth_tracker);   /* growth_tracker: 0 for a fresh scaler; a resumed checkpoint's `_growth_tracker` */
int v2a_opt_state_scaler(const void* host_state, float* loss_scale, int* growth_tracker, int* skipped_last, long long* skipped_steps,
                         float* growth_factor, float* backoff_factor, int* growth_interval, int* scaler_on);   /* every out pointer optional */
size_t v2a_opt_state_loss_scale_offset(void);
int v2a_opt_step(const int64_t* table_dev, const int* chunks_dev, int nchunks, void* state_dev, double* partial_dev, int zero_grad, v2a_stream_t s);
/* same, and the update kernel also writes the conv operand packs of the updated parameters (packs_dev: [tensors][6] int64 = {forward pack
   [Cout][taps][Cin] or 0, Cin, taps, channel-window pack (v2a_conv2d_fwd_window_f32) or 0, 16-bit side copy of the forward pack or 0, its
   format: 0 bf16 twin, 1 IEEE fp16 twin, 2 = three bf16 planes hi / mid / lo of the fp32 value, the tensor's element count apart
   (v2a_conv2d_fwd_p3's weight operand)}; the launch of v2a_pack_weights_multi that would read every parameter again is not needed for
   those operands) */
   presum_first / presum_count: chunks [first, first + count) of partial_dev already hold this step's sums of squares (written by
   v2a_opt_presum with the same range on a stream ordered before this call); 0 / 0: every chunk is summed here.  Explicit operands: nothing
   is remembered between the two calls. */
int v2a_opt_step_packed(const int64_t* table_dev, const int* chunks_dev, int nchunks, void* state_dev, double* partial_dev, int zero_grad,
                        const int64_t* packs_dev, int presum_first, int presum_count, v2a_stream_t s);
/* gradient-norm partial sums of chunks [first, first + count) ahead of the step (those gradients are final earlier: the ConditionalUnet1D
   slice, while the encoder backward runs).  Stateless: the caller hands the same range to v2a_opt_step_packed. */
int v2a_opt_presum(const int64_t* table_dev, const int* chunks_dev, int first, int count, double* partial_dev, v2a_stream_t s);
int v2a_opt_scale_grads(const int64_t* table_dev, const int* chunks_dev, int nchunks, float scale, v2a_stream_t s);  /* 1/world after the RCCL sum */

/* ---------------------------------------------------------------------------------------------- replay (csrc/replay.hip)
 * Global_EnvReplayBuffer_Img.sample_random_batch_seq + EnvImg_UnitBuffer.sample_seq (diffuser/datasets/env_img_replay_buffer.py:
 * 68-116, 278-302) and the 'rand_prob' split of sample_from_bufs (lb_online_trainer_v7.py:826-830).
 * np_state / py_state: HOST uint32[625] = MT19937 key + position of numpy's legacy global RandomState / CPython's `random`;
 * both are advanced in place exactly as np.random.randint / random.randint would have advanced them. */
int v2a_replay_sample_indices(uint32_t* np_state, uint32_t* py_state, const int32_t* episode_len, int32_t n_episodes, int32_t batch,
                              int32_t act_len, int64_t* out_episode, int64_t* out_start);          /* all HOST pointers */
int v2a_replay_count_uniform_below(uint32_t* np_state, int32_t batch, double prob);              /* HOST; returns the count */
int v2a_mt_seed_numpy(uint32_t* state, uint32_t seed);                                           /* HOST; np.random.seed(int) */
int v2a_mt_seed_python(uint32_t* state, const uint32_t* key, int key_len);                       /* HOST; random.seed(int) */
/* frames [n][H][W][3] (uint8 or fp32), acts [n][act_dim]; start/goal images + action chunk of every row in one gather */
int v2a_replay_gather(const void* frames, int dtype_u8, const float* acts, const int64_t* frame_start, float* out_start, float* out_goal,
                      float* out_acts, int B, int H, int W, int act_len, int act_dim, int normalize, int chw_out, v2a_stream_t s);

/* ---- deferred weight-gradient reduces: one launch sums the split-K slabs of MANY layers (same summation order as the per-layer
 * reduce kernels).  v2a_conv2d_wgrad_deferred / _h_deferred = v2a_conv2d_wgrad / _h with the reduce left out: `slabs` is the layer's own
 * scratch (it must stay untouched until v2a_wgrad_reduce_multi ran); item_out (HOST, v2a_wgrad_item_bytes() bytes), *blocks_out and
 * *form_out describe the pending reduce (blocks_out = 0: dw is already final).  The caller copies the items to the device and lists the
 * work as [nwork][4] int32 rows {item, block, blocks of the item, form}. */
int v2a_wgrad_item_bytes(void);
int v2a_conv2d_wgrad_deferred(const float* x, const float* x2, const float* dy, float* dw, float* dbias, int N, int H, int W, int C1, int C2,
                              int OH, int OW, int Cout, int KH, int KW, int sh, int sw, int ph, int pw, int idil, int ups, int accumulate,
                              void* slabs, size_t slab_bytes, void* item_out, int* blocks_out, int* form_out, v2a_stream_t stream);
int v2a_conv2d_wgrad_h_deferred(const void* x_h, const void* x2_h, const void* dy_h, float* dw, float* dbias, int N, int H, int W, int C1,
                                int C2, int OH, int OW, int Cout, int KH, int KW, int sh, int sw, int ph, int pw, int idil, int ups,
                                int accumulate, void* slabs, size_t slab_bytes, void* item_out, int* blocks_out, int* form_out,
                                v2a_stream_t stream);
int v2a_wgrad_reduce_multi(const void* items_dev, const void* work_dev, int nwork, v2a_stream_t stream);

/* ---- grouped weight gradients: up to v2a_wgrad_multi_max() gradients of different layers in ONE launch (descriptors travel in the
 * kernel arguments; replaces the per-layer torch autograd weight-gradient calls of ResNet18 / ConditionalUnet1D in the policy step,
 * diffuser/diffusion_policy/common/vision_nets.py:29-39, model/conditional_unet1d.py:46-66).
 * v2a_conv2d_wgrad_describe launches nothing.  want_splits = 0 (planning): writes *variant_out (0 = exact-f32 LDS-DMA body on x / dy,
 * 1 / 2 = bf16 twin-fed 128- / 64-row body on x_h / dy_h, -1 = not eligible: use v2a_conv2d_wgrad), *tiles_out, *rtiles_out.
 * want_splits >= 1: also item_out (HOST, v2a_wgrad_item_bytes()), *splits_out (<= want_splits) and the reduce item (ritem_out,
 * *rblocks_out, *rform_out) for v2a_wgrad_reduce_multi; `slabs` = the layer's own scratch, (splits * Cout * K + splits * Cout) * 4 B.
 * v2a_conv2d_wgrad_multi: items / variants / tiles are HOST arrays of n entries. */
int v2a_conv2d_wgrad_describe(const float* x, const float* x2, const float* dy, const void* x_h, const void* x2_h, const void* dy_h, float* dw,
                              float* dbias, int N, int H, int W, int C1, int C2, int OH, int OW, int Cout, int KH, int KW, int sh, int sw, int ph,
                              int pw, int idil, int ups, int accumulate, int want_splits, void* slabs, size_t slab_bytes, void* item_out,
                              int* variant_out, int* tiles_out, int* rtiles_out, int* splits_out, void* ritem_out, int* rblocks_out,
                              int* rform_out);
int v2a_wgrad_multi_max(void);
int v2a_wgrad_family(int variant);   /* kernel family of a described gradient (one family per v2a_conv2d_wgrad_multi launch): 0 exact 64x64 / twin-fed, 1 halo, 2 / 3 three-bf16-plane 64x64 / 128x128 */
int v2a_conv2d_wgrad_multi(const void* items, const int* variants, const int* tiles, int n, v2a_stream_t stream);

/* ---------------------------------------------------------------------------------------------- persistent predict_action (csrc/policy_persist.hip)
 * Every scheduler step of DiffusionUnetImagePolicy.conditional_sample (diffusion_policy/diffusion_unet_image_policy.py:88-133) over
 * ConditionalUnet1D.forward (model/conditional_unet1d.py:186-246), the scheduler updates and the action un-normalisation (:139-201) as ONE
 * persistent launch for batch <= 2 (the rollout loop calls predict_action at batch 1: diffuser/libero/lb_online_trainer_v7.py:1060-1122).
 * The caller describes the network as an op list in device memory (struct PPOp / PPArgs in the .hip file; v2a_hip/policy_persist.py builds
 * it from the live fp32 parameters, nothing is packed). */
size_t v2a_policy_persist_op_bytes(void);          /* sizeof(PPOp) / sizeof(PPArgs): the host mirror checks itself against these */
size_t v2a_policy_persist_args_bytes(void);
int v2a_policy_persist_waves_per_wg(void);         /* the host picks how many waves share an output channel from this and nwg */
size_t v2a_policy_persist_lds_bytes(int B, int Tin, int Tout, int Cin, int K, int stride, int pad, int type);   /* 0: the op cannot run */
int v2a_policy_persist_launch(const void* args_host, int nwg, size_t lds_bytes, v2a_stream_t stream);   /* nwg workgroups, all resident (<= #CUs) */

/* ---------------------------------------------------------------------------------------------- direct gradient exchange (csrc/dp.hip)
 * The second algorithm behind the data-parallel gradient all-reduce of the policy step (reference: torch DDP through accelerator.prepare,
 * diffuser/libero/lb_online_trainer_v7.py:153-154; its hooks reduce inside accelerator.backward :604, before clip_grad_norm_ :608): a
 * reduce-scatter + all-gather over hipIpc peer pointers in one stream-ordered launch per arena slice, for a node where RCCL would run the
 * 349 MB message over a ring.  Sums are taken in rank order on the chunk's owner, so every rank ends with bit-identical gradients.
 * Host side: v2a_hip/dp.py GradReducer(algo="direct"). */
int v2a_dp_max_world(void);                        /* ranks one launch can connect (8: one xGMI node) */
int v2a_dp_slots(void);                            /* independent flag sets = arena slices that may be in flight together */
size_t v2a_dp_signal_bytes(void);
int v2a_dp_arena_alloc(void** arena_out, size_t bytes);   /* a gradient arena as an allocation of its own (zeroed): the unit the peers map */
int v2a_dp_arena_free(void* arena);
int v2a_dp_signal_alloc(void** sig_out);           /* this rank's flag block: uncached device memory, zeroed */
int v2a_dp_signal_free(void* sig);
int v2a_dp_errword_alloc(int** word_out);          /* 16 pinned host ints the kernel raises: [0] 0 fine / 1 + p: peer p never arrived; [1] slot,
                                                      [2] flag value awaited (3 * epoch + barrier), [3] workgroup, [4] flag value seen */
int v2a_dp_errword_free(int* word);
int v2a_dp_ipc_export(const void* ptr, void* handle64_out, uint64_t* offset_out, uint64_t* alloc_bytes_out);   /* handle of the allocation
                                                      holding ptr, ptr's offset in it, the allocation's size */
int v2a_dp_ipc_open(const void* handle64, void** base_out);                          /* map a peer's allocation (once per handle and process) */
int v2a_dp_ipc_close(void* base);
/* arenas / signals: HOST arrays of `world` device pointers valid in this process, same peer order on every rank; slot < v2a_dp_slots();
   epoch = 1, 2, 3 ... per slot, equal on all ranks; blocks: workgroups of the launch (0 = default 64, <= 256), equal on all ranks.
   Stream-ordered, no host wait. */
int v2a_dp_allreduce_direct(float* const* arenas, uint32_t* const* signals, int world, int rank, size_t lo, size_t hi, int slot, uint32_t epoch,
                            int* errword, int timeout_ms, int blocks, v2a_stream_t stream);

/* ---------------------------------------------------------------------------------------------- random-action episode file (csrc/h5read.hip)
 * Native reader for the HDF5 file of the reference's generator (environment/libero/lb_data/lb_randsam.py:84-104: groups
 * `{task}/{episode}` holding `agentview_image` uint8 [T+1,128,128,3], `action` float [T,7], `ee_poses` float [T+1,3]), replacing the
 * h5py calls of the loader (diffuser/libero/lb_online_trainer_v7.py:718-780).  HOST functions, HOST pointers, no stream; every
 * function returns 0 / a count, or -1 with the reason in v2a_h5_last_error().  Supported: the "earliest" on-disk format h5py writes by
 * default (superblock 0/1, symbol-table groups, version-1 object headers, contiguous / compact / unfiltered chunked datasets of
 * little-endian integers and floats); everything else is refused, not guessed. */
int v2a_h5_open(const char* path, void** handle_out);
void v2a_h5_close(void* handle);
const char* v2a_h5_last_error(void* handle);
int v2a_h5_exists(void* handle, const char* path);                                   /* 1 / 0 / -1 */
long v2a_h5_list(void* handle, const char* group_path, char* buf, size_t cap);       /* member names, '\n'-separated; -2: buf too small */
int v2a_h5_dataset_info(void* handle, const char* path, int* type_class, int* elem_size, int* is_signed, int* ndim, long long* dims,
                        long long* nbytes);                                        /* type_class 0 integer, 1 float; dims[8] */
int v2a_h5_read(void* handle, const char* path, void* dst_host, size_t dst_bytes);   /* whole dataset, row-major */

#ifdef __cplusplus
}
#endif
#endif /* V2A_H */
