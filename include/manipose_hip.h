/* manipose_hip.h -- C ABI of libmanipose_hip.so: the MI355X (gfx950) implementation of ManiPose's
 * data-parallel 2D->3D lifting hot path.
 *
 * The reference (cedricrommel/manipose @ 2025-01-17) is pure Python/PyTorch and has NO native interface;
 * its "plugin API" for this path is the nn.Module contract of hpe/mh_so3_hpe/architectures.
 * This header is the boundary a binding for that contract attaches to (ctypes stub: INTEGRATION.md; the
 * in-tree binding is manipose_amd/_lib.py).  Each entry point names the reference code it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  All data pointers are DEVICE pointers to
 *     contiguous fp32 buffers unless stated otherwise; `stream` is a hipStream_t (NULL = default stream).
 *   - every function returns 0 on success, non-zero on error (1 bad argument, 2 HIP runtime error,
 *     3 bad state); mp_last_error() returns the message of the last failure on the calling thread.
 *   - calls only ENQUEUE work on `stream`; nothing synchronises the host except mp_prof_collect().
 *   - token layout everywhere: row m = (b*T + t)*J + j of a (B, T, J, C) activation.
 *   - pose layout: (B, K, T, 17, 3); scores: (B, K, T[, 1]); targets: (B, T, 17, 3)  (reference layouts).
 */
#ifndef MANIPOSE_HIP_H
#define MANIPOSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MP_ABI_VERSION 8

int mp_abi_version(void);
const char* mp_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Stand-alone operators (the HBM-bound kernels of the path; also used by the roofline harness)
 * ---------------------------------------------------------------------------------------------- */

/* PoseDecoder.forward: architectures/pose_decoder.py:32-55 = _compute_rotation_mats (:57-83,
 * utils/rotation_tools.py:35-57) + build_t_pose_from_bone_lengths (:98-120) + forward_kinematics
 * (utils/forward_kinematics.py:6-48), root at the origin (rmcl_manifold_mix_ste.py:92).
 *   rot6d  : (K, B*T*17, rot_stride) head output, channels 0..rot_dim-1 are the rotation representation (rot_stride >= rot_dim)
 *   rot_dim: 6 = two 3-vectors, Gram-Schmidt (rotation_tools.py:35-57); 4 = two 2-vectors -> R_theta R_phi
 *            (rotation_tools.py:60-116; conf/config.yaml:47 model.rot_dim)
 *   lengths: (B, 16) segment lengths;  poses: (B, K, T, 17, 3) */
int mp_fk_decode_fwd(const float* rot6d, int rot_stride, int rot_dim, const float* lengths, float* poses, int B, int K, int T,
                     void* stream);
/* gradient of the above: d_rot6d has rot6d's layout (channels >= rot_dim untouched); d_len_pose: (B*K*T, 16)
 * per-pose segment-length gradients (summed over a window's poses by mp_bones_mean_bwd). */
int mp_fk_decode_bwd(const float* rot6d, int rot_stride, int rot_dim, const float* lengths, const float* d_poses, float* d_rot6d,
                     float* d_len_pose, int B, int K, int T, void* stream);

/* Multi-hypothesis training loss with its gradient, metrics/losses.py:104-170 + :75-101 +
 * metrics/regularizations.py:160-174 assembled like make_loss/compute_and_acc_loss
 * (hpe/main_h36m_lifting.py:101-209).  terms (device, 4 floats) = wloss, score_reg, vloss, sreg (already
 * weighted; total = their sum).  argmin (device int32 (B,T)) / d_poses / d_scores may be NULL.
 * scratch: >= 4*ceil(B*T/256) floats; with >= 4*ceil(B*T/48) the kernel runs on more, shorter-lived workgroups (same results up to the order of the
 * final sum). */
typedef struct mp_loss_config {
  float rmcl_score_reg; /* beta, conf/config.yaml:36 (0.1) */
  float vel_loss;       /* conf/config.yaml:33 (2.0) */
  float smooth_reg;     /* conf/config.yaml:34 (0.5) */
  int w_loss;           /* conf/config.yaml:32: 1 = weight joints by STANDARD_H36M_WEIGHTS (losses.py:6-8); 0 = unweighted; 2 = joint_weights below */
  int sq_loss;          /* conf/config.yaml:31: squared distances in the WTA and velocity terms (losses.py:46-72,96-97,110-116) */
  float joint_weights[17]; /* w_loss == 2: the caller's per-joint weights (the `weights` argument of losses.py:14-43,104-138, regularizations.py:160-174) */
} mp_loss_config;
int mp_wta_loss(const float* poses, const float* scores, const float* target, const mp_loss_config* cfg, float* terms,
                int32_t* argmin, float* d_poses, float* d_scores, int B, int K, int T, float* scratch,
                int64_t scratch_floats, void* stream);
/* single-hypothesis variant (ManifoldMixSTE, make_loss :113-127): terms (3 floats) = wloss, vloss, sreg */
int mp_single_loss(const float* poses, const float* target, const mp_loss_config* cfg, float* terms, float* d_poses,
                   int B, int T, float* scratch, int64_t scratch_floats, void* stream);

/* The rigid_seg_reg term of make_loss (hpe/main_h36m_lifting.py:170-177): weight * segments_time_consistency(pred.permute(0,3,2,1),
 * skeleton, mode="sum") (metrics/regularizations.py:8-45, metrics/utils.py:4-20) = weight * sum over windows and the 16 bones of the
 * unbiased variance over time of the bone length, for (B, T, 17, 3) predictions of the single-hypothesis models.  term: 1 device float;
 * d_poses (may be NULL): the gradient is ADDED to it (the other loss terms' gradient is already there).  scratch >= B floats. */
int mp_rigid_segments_loss(const float* poses, float weight, float* term, float* d_poses, int B, int T, float* scratch,
                           int64_t scratch_floats, void* stream);

/* RMCLManifoldMixSTE.aggregate (rmcl_manifold_mix_ste.py:141-185): mode 0 "weighted_ave", 1 "best_score",
 * 2 "oracle" (needs target).  out: (B, T, 17, 3). */
int mp_aggregate(const float* poses, const float* scores, const float* target, int mode, float* out, int B, int K, int T,
                 void* stream);
/* mpjpe_error(mode="sum") (metrics/mean_joint_errors.py:31-36): out_sum (device, 1 float) = sum of the
 * per-joint L2 errors of n_joints joints.  scratch >= 4*min(ceil(n/256),1024)+4 floats. */
int mp_mpjpe_sum(const float* pred, const float* target, int64_t n_joints, float* out_sum, float* scratch,
                 int64_t scratch_floats, void* stream);

/* torch.optim.Adam(lr, weight_decay) update (hpe/main_h36m_lifting.py:234-238), fused over a flat buffer.
 * grad_scale multiplies the gradient first (1/world_size after a sum all-reduce). step counts from 1. */
int mp_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int step, float lr,
                 float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream);

/* The same with per-element multipliers of the learning rate and of the weight decay (flat buffers in the parameter layout):
 * mup.optim.MuAdam as the reference builds it under model.mup (hpe/main_h36m_lifting.py:227-232) - parameters with two width
 * dimensions train with lr / width_mult and weight_decay * width_mult. */
int mp_adam_step_scaled(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int step, float lr, float beta1,
                        float beta2, float eps, float weight_decay, float grad_scale, const float* lr_mult, const float* wd_mult, void* stream);

/* Building blocks exposed for unit parity tests (same kernels the model engine launches). */
int mp_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, float* y, float* stats, int M, int C,
                     void* stream);
/* scratch: >= 1024 * 2 * C floats (per-workgroup dgamma/dbeta partials, reduced deterministically) */
int mp_layernorm_bwd(const float* dy, const float* x, const float* stats, const float* gamma, const float* dskip, float* dx,
                     float* dgamma, float* dbeta, int M, int C, float* scratch, int64_t scratch_floats, void* stream);
/* y = x W^T + b (nn.Linear); epilogue 0 none, 1 GELU (z receives gelu'(x W^T + b), which is all the backward needs), 2 residual: y = r + y */
int mp_linear_fwd(const float* x, const float* W, const float* b, float* y, float* z, const float* r, int M, int N, int K,
                  int epilogue, void* stream);
/* dx = dy W ; dW += dy^T x ; db += colsum(dy).  slab >= workspace reported by mp_linear_bwd_slab_floats */
int64_t mp_linear_bwd_slab_floats(int N, int K);
int mp_linear_bwd(const float* dy, const float* x, const float* W, float* dx, float* dW, float* db, int M, int N, int K,
                  float* slab, int64_t slab_floats, void* stream);
/* bf16 matrix-core variants (precision 1 of the engine).  x, W, dy are bf16 unless *_f32 says fp32; y/z bf16, dx fp32
 * or bf16 (dx_f32); r and the residual output y of epilogue 2 are fp32; dW/db fp32 (accumulated). */
int mp_linear_fwd_bf16(const void* x, const void* W, const float* b, void* y, void* z, const float* r, int M, int N, int K,
                       int epilogue, void* stream);
int mp_linear_bwd_bf16(const void* dy, int dy_f32, const void* x, const void* W, void* dx, int dx_f32, float* dW, float* db,
                       int M, int N, int K, float* slab, int64_t slab_floats, void* stream);
/* Attention core of Attention.forward (architectures/mix_ste.py:271-279) on a fused qkv buffer (M, 3C).
 * temporal = 0: attends over the J tokens of a frame; 1: over the T frames of a joint. */
int mp_attention_fwd(const float* qkv, float* out, float* lse, int temporal, int B, int T, int J, int C, int H, void* stream);
int mp_attention_bwd(const float* qkv, const float* out, const float* d_out, const float* lse, float* delta, float* d_qkv,
                     int temporal, int B, int T, int J, int C, int H, void* stream);

/* bf16 storage variants of the two calls above (temporal: MFMA kernels when T <= 256 and head dim in {64, 16}) */
int mp_attention_fwd_bf16(const void* qkv, void* out, float* lse, int temporal, int B, int T, int J, int C, int H, void* stream);
int mp_attention_bwd_bf16(const void* qkv, const void* out, const void* d_out, const float* lse, float* delta, void* d_qkv,
                          int temporal, int B, int T, int J, int C, int H, void* stream);

/* Split precision "bf16x3" (precision 2 of the engine): a value x is carried as two bf16 numbers hi = bf16(x), lo = bf16(x - hi)
 * in two planes of the same shape, and every product a*b of a Linear layer (architectures/mix_ste.py:216-222,257-261,280-281) or of
 * the attention core (:271-279) is evaluated as a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on the bf16 matrix cores with fp32 accumulation:
 * 16 significand bits per operand, which keeps the model within the 1e-4 m MPJPE bound of the fp32 reference at matrix-core speed.
 * mp_split_bf16: fp32 -> (hi, lo) planes (n a multiple of 4). */
int mp_split_bf16(const float* src, void* hi, void* lo, int64_t n, void* stream);
/* y = x W^T + b on planar operands.  epilogue 0: y planar (y = hi plane, y_lo); 1: GELU, y planar and z = gelu'(pre-activation) as
 * plain bf16; 2: y = r + (x W^T + b) in fp32 (y_lo unused). */
int mp_linear_fwd_bf16x3(const void* x_hi, const void* x_lo, const void* W_hi, const void* W_lo, const float* b, void* y, void* y_lo,
                         void* z, const float* r, int M, int N, int K, int epilogue, void* stream);
/* The residual Linear of a Block from the third block on (architectures/mix_ste.py:352-368 inside ST_foward :157-173): the block input is
 * the shared post-norm of the previous block's output, x = LayerNorm(r_in) (Spatial_norm / Temporal_norm, eps folded into rstats), which
 * the engine never materialises - the epilogue recomputes it from r_in and its row statistics while adding the branch:
 *   y[m][n] = (r_in[m][n] - mean[m]) * rstd[m] * rgamma[n] + rbeta[n] + mask(m) * (x W^T + b)[m][n]        (fp32, N columns)
 * rstats: (M, 2) = (mean, rstd) per row; mask: DropPath multipliers per sample or NULL (mask_mode 1: sample = m / J, a spatial block's
 * (b, t); 2: sample = b * J + m % J, a temporal block's (b, j), with b = m / (T J)); r_in and y may not alias.  This is the same
 * kernel path mp_model_forward takes in precision 2 (persistent split-precision GEMM when the problem has >= 2 tiles per CU or
 * "gemm_persist_min_tiles" says so). */
int mp_linear_fwd_bf16x3_lnres(const void* x_hi, const void* x_lo, const void* W_hi, const void* W_lo, const float* b, float* y,
                               const float* r_in, const float* rstats, const float* rgamma, const float* rbeta, const float* mask, int mask_mode,
                               int T, int J, int M, int N, int K, void* stream);
/* The same Linear (architectures/mix_ste.py:216-222, 257-261) on the "f16f8" operand format - one fp16 product plus ONE block-scaled fp8
 * product per 64 reduction indices instead of three bf16 products.  A value v is carried as hi = fp16(v) (x16 / W16, row-major [rows][K])
 * and a correction plane of the same byte geometry (x8 / W8: 2 K bytes per row): for every four reduction indices 4 q .. 4 q + 3 the 8 bytes
 *   activation row:  4 x e4m3(2^11 (v - hi))  |  4 x e4m3(hi)
 *   weight row:      4 x e4m3(2^4 hi)         |  4 x e4m3(2^15 (v - hi))
 * so that the fp8 dot product of an activation row and a weight row is 2^15 (x_lo w_hi + x_hi w_lo).  y = x W^T + b in fp32.
 * N must be a multiple of 256, K of 64 (>= 128).  The engine runs its qkv and fc1 forward GEMMs in this form (precision 2, "f16f8_inputs"). */
/* fp32 -> the two planes of that format for a matrix whose rows are multiples of 64 elements long (n = rows * K elements; hi16: n fp16
 * values, corr8: 2 n bytes); weight != 0 selects the weight form of the correction rows. */
int mp_split_f16f8(const float* src, void* hi16, void* corr8, int64_t n, int weight, void* stream);
int mp_linear_fwd_f16f8(const void* x16, const void* x8, const void* W16, const void* W8, const float* b, float* y, int M, int N, int K,
                        void* stream);
/* attention core on a planar fused qkv buffer, planar output.  scratch: 4*M*C floats, needed only where no MFMA kernel covers the
 * shape (spatial: 16 <= J <= 32 tokens, head dim 64 or 16, <= 8 heads; temporal: T <= 256, head dim 64 or 16); NULL otherwise. */
int mp_attention_fwd_bf16x3(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, float* lse, float* scratch, int temporal,
                            int B, int T, int J, int C, int H, void* stream);

/* K output heads, head k = LayerNorm(C, eps 1e-5) -> Linear(C, O)  (MCLHead stack, rmcl_manifold_mix_ste.py:291-298; MixSTE.head,
 * mix_ste.py:123-126), all fp32.  Packed parameters: gamma, beta [K][C]; W [K][O][C]; b [K][O].  out [K][M][O]; stats [M][2] (mean, rstd
 * of x, shared by the heads) and fold (mp_heads_fold_floats(C) floats: the LayerNorm affine folded into the weights) are written by the
 * forward and, with out, read by the backward.  impl 0 = the engine's choice (fp32 matrix cores when C is 128, 256 or 512 and K*O <= 48, row kernels
 * otherwise), 1 = row kernels, 2 = matrix cores (MP_ERR_ARG when the shape is not covered).
 * Backward: dx [M][C] is overwritten; dgamma, dbeta, dW, db are ACCUMULATED into.  scratch: mp_heads_bwd_scratch_floats(K, O, C). */
int64_t mp_heads_fold_floats(int C);
int64_t mp_heads_bwd_scratch_floats(int K, int O, int C);
int mp_heads_fwd(const float* x, const float* gamma, const float* beta, const float* W, const float* b, int K, int O, float* out, float* stats,
                 float* fold, int M, int C, int impl, void* stream);
int mp_heads_bwd(const float* x, const float* stats, const float* fold, const float* out, const float* gamma, const float* beta, const float* W,
                 const float* b, const float* d_out, float* dx, float* dgamma, float* dbeta, float* dW, float* db, int K, int O, int M, int C, int impl,
                 float* scratch, int64_t scratch_floats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Model engine: RMCLManifoldMixSTE / ManifoldMixSTE forward + backward as one native launch sequence
 * (replaces RMCLManifoldMixSTE.forward, architectures/rmcl_manifold_mix_ste.py:83-106 and everything it
 * calls: mix_ste.py:128-173,216-222,255-282,352-368; manifold_mix_ste.py:75-88,139-154; pose_decoder.py)
 * ---------------------------------------------------------------------------------------------- */
typedef struct mp_model mp_model;

typedef struct mp_model_config {
  int arch;            /* 0 = "rmcl_manifold" (K heads + scores), 1 = "manifold" (single hypothesis), 2 = "mixste": the bare MixSTE
                        * regressor of main_h36m_lifting.py:617-628 (mix_ste.py:175-191; out_dim 3, no bones net, no decoder;
                        * state-dict keys without the "rotations_module." prefix) */
  int num_frame;       /* T  (cfg.data.seq_len) */
  int num_joints;      /* 17 */
  int num_bones;       /* 16 */
  int embed_dim_rot, depth_rot, num_heads_rot;
  int embed_dim_seg, depth_seg, num_heads_seg;
  int n_hyp;           /* K (ignored for arch 1) */
  float drop_path_rate;/* stochastic depth: linspace(0, rate, depth) per module (mix_ste.py:70) */
  int max_batch;       /* workspace is sized for this many windows; 0 = layout-only handle (no device memory) */
  int precision;       /* 0 = fp32 matrix cores; 1 = bf16 matrix cores / fp32 accumulate; 2 = "bf16x3": split-precision forward (planar hi/lo
                        * bf16 operands, three matrix-core products each: within the 1e-4 m MPJPE bound); the backward runs in bf16 on the hi
                        * planes unless f16_backward (below) moves the qkv / fc1 (/ fc2) layers of an f16f8 model to fp16 operands */
  int rot_rep_dim;     /* 6 (default when 0) or 4: rotation representation the heads emit (pose_decoder.py:22-31) */
  /* mu-parametrisation (model.mup, conf/config.yaml:52) and explicit attention scales; per backbone (rotations / segments), 0 = default:
   *   qk_scale      softmax scale of Attention (mix_ste.py:243-244): default head_dim^-0.5; muP 1 / head_dim
   *   resid_scale   Block.residual_scale (mix_ste.py:330,353-358): x = x * resid_scale + branch; default 1; muP 1 / sqrt(depth)
   *   readout_mult  multiplier on the input of the head's last Linear, mup.MuReadout (mix_ste.py:118-121, rmcl_manifold_mix_ste.py:278-285):
   *                 y = W (readout_mult * x) + b with readout_mult = output_mult / width_mult; default 1 */
  float qk_scale_rot, resid_scale_rot, readout_mult_rot;
  float qk_scale_seg, resid_scale_seg, readout_mult_seg;
  /* ABI v7: the numerics- and scheduling-affecting knobs of ONE model (they were process-wide options / environment variables up to v6;
   * the library reads no environment variable).  All 0 = the defaults.
   *   f16f8         precision 2 only, rotations net of a width that is a multiple of 256: 0 (default) = every Linear product as three bf16
   *                 products of bf16 hi / lo planes; 1 = the qkv and fc1 Linear layers read "f16f8" operands (mp_linear_fwd_f16f8: one fp16
   *                 product + one block-scaled fp8 correction product per 64 reduction indices); 2 = the fc2 layer as well (needs f16_backward);
   *                 3 (round 6; head dim 64, residual scale 1) = ALL FOUR Linear layers of a block: the attention kernels and the fc1 epilogue
   *                 write their outputs as f16f8 planes too, no bf16 copy of those activations exists, and the bf16 backward rounds the fp16
   *                 planes to bf16 where it reads them (weight-gradient GEMM fragments, the temporal attention backward's O); f16_backward
   *                 must be 0; a model that does not qualify (other head dims / widths, muP residual scale) runs as f16f8 = 0
   *   f16_backward  with f16f8 >= 1: 1 = the backward GEMMs of those layers run on fp16 operands - gradients carried as fp16 of S x value,
   *                 S a power of two chosen per backward on the device from the residual gradient at the top of the backbone, stores saturate at
   *                 +-65504 and are counted (mp_model_grad_health);
   *                 0 (default) = bf16 backward on a bf16 copy of those activations
   *   streams       bit 0 set: the segments net is enqueued on the caller's stream instead of the engine's side stream; bit 1 set: the
   *                 weight-gradient GEMMs likewise instead of the engine's third stream (debugging / single-queue profiles; results are
   *                 bit-identical either way) */
  int f16f8;
  int f16_backward;
  int streams;
  /* ABI v8.  debug: bit 0 = the stream-hazard check (below): every launch of mp_model_forward / mp_model_backward is declared to a host-side
   * tracker together with the events the engine records and waits on; costs host time only, changes no launch.  0 = off. */
  int debug;
} mp_model_config;

int mp_model_create(const mp_model_config* cfg, mp_model** out);
void mp_model_destroy(mp_model* m);
int64_t mp_model_workspace_bytes(const mp_model* m);

/* Flat parameter layout: parameter i (state-dict key `name`, reference naming) occupies
 * floats [offset, offset + numel) of the flat parameter buffer (offsets are 16-byte aligned; gaps are
 * padding).  The gradient buffer and the Adam moment buffers use the same layout. */
int mp_model_num_params(const mp_model* m);
int64_t mp_model_flat_size(const mp_model* m);
int mp_model_param_info(const mp_model* m, int index, char* name, int name_cap, int64_t* offset, int64_t* numel);

/* DropPath mask layout for a batch of B windows: branch i (name like "rotations_module.STEblocks.3.attn")
 * uses floats [offset, offset+count) of the mask buffer; values are 0 or 1/keep. */
int mp_model_num_mask_branches(const mp_model* m);
int mp_model_mask_info(const mp_model* m, int B, int index, char* name, int name_cap, int64_t* offset, int64_t* count,
                       float* keep_prob);
int64_t mp_model_mask_floats(const mp_model* m, int B);

/* forward: x (B, T, 17, 2) -> poses (B, K, T, 17, 3) [, scores (B, K, T, 1) for arch 0].
 * train is a bit set: bit 0 applies DropPath: masks = masks_override (device, layout above) if non-NULL, else drawn from
 * (seed, step); bit 1 (value 2) announces that NO mp_model_backward will follow this forward (torch.no_grad() evaluation): tensors that
 * only the backward reads are then not written, and mp_model_backward returns MP_ERR_STATE.  Otherwise the activations needed by
 * mp_model_backward are kept inside the model until the next forward. */
int mp_model_forward(mp_model* m, const float* flat_params, const float* x, int B, float* poses, float* scores, int train,
                     const float* masks_override, uint64_t seed, uint64_t step, void* stream);
/* backward of the LAST forward: d_poses (B,K,T,17,3), d_scores (B,K,T,1) or NULL; parameter gradients are
 * ACCUMULATED into flat_grads (zero it first for a fresh gradient). */
int mp_model_backward(mp_model* m, const float* flat_params, float* flat_grads, const float* d_poses, const float* d_scores,
                      void* stream);
/* Gradient buckets for overlapping the data-parallel exchange with the backward (the reference has no counterpart: it wraps the
 * model in nn.DataParallel, hpe/main_h36m_lifting.py:749-751).  Bucket i = the parameter gradients of layer i of the rotations net
 * (STEblocks.i and TTEblocks.i: one contiguous range [offset, offset + numel) of the flat gradient buffer, ~25 MB at full width).  The
 * backward finishes them from the last layer down; mp_model_grad_bucket_wait makes `stream` wait (device side, the host does not block)
 * until bucket `index` of the LAST mp_model_backward is final, so a collective enqueued on that stream afterwards runs while the rest of
 * the backward is still computing.  Everything outside the buckets (embeddings, shared norms, heads, the segments net) is final when the
 * stream mp_model_backward was given has passed the call. */
int mp_model_grad_bucket_count(const mp_model* m);
int mp_model_grad_bucket_info(const mp_model* m, int index, int64_t* offset, int64_t* numel);
int mp_model_grad_bucket_wait(mp_model* m, int index, void* stream);
/* Health of the scaled-fp16 gradient operands of the LAST mp_model_backward (f16_backward models; zeros otherwise): copies 4 floats to the
 * HOST after synchronising `stream`: out[0] = S, the power-of-two scale of that backward; out[1] = number of fp16 gradient elements that
 * hit the +-65504 clamp (stores saturate, they never write inf); out[2] = number of non-finite gradient elements met at those stores
 * (written as 0); out[3] = 1 / S (what consumers of the scaled operands multiply by).  A trainer that sees out[1] + out[2] > 0 should redo
 * the step with f16_backward off or skip it (manipose_amd/training.py sums the counters on the device, all-reduces the sum over the ranks and
 * raises / warns every health_interval steps - the steps since the last check were applied; health_interval = 1 stops on the first).  MP_ERR_STATE unless a completed mp_model_backward is
 * the last engine call (a forward, or a backward that failed midway, has overwritten what the counters described).
 * mp_model_grad_health_async: the same four values as floats into a DEVICE (or pinned host) buffer by an asynchronous copy on `stream`, no
 * host synchronisation - for a trainer that looks at the counters every N steps.  Cost, accepted: a backward of an f16_backward model reads the
 * residual-gradient stream of the heads once more on its main stream to choose S (one M x C fp32 pass, ~0.25 ms at the benchmark's batch) and
 * resets the counters there; the synchronous form blocks the host until `stream` is idle. */
int mp_model_grad_health(mp_model* m, float* out4_host, void* stream);
int mp_model_grad_health_async(mp_model* m, float* out4_device, void* stream);
/* Which of the engine's two extra streams the NEXT forward / backward calls use: the `streams` bit set of mp_model_config, changed on a live
 * model (bench.py times the kernel classes of the same model with every kernel on one queue).  Synchronises the engine's streams first. */
int mp_model_set_streams(mp_model* m, int streams);
/* Stream-hazard check (mp_model_config::debug bit 0; csrc/hazard.h): the engine runs its rotations net, its segments net and its weight-gradient
 * GEMMs on three streams ordered by events.  With the check on, every launch declares the byte ranges of the workspace / gradient buffers it
 * reads and writes and its stream, every event record / wait is mirrored, and a vector clock per stream decides for each pair of launches on
 * different streams that touch the same bytes (at least one writing) whether an event path orders them.  out4 = {launches declared, conflicting
 * cross-stream pairs found ORDERED, pairs found UNORDERED (violations), events recorded}; the first violations are written to `msg` as lines
 * of text (may be NULL).  Returns MP_ERR_STATE when the model was created without the debug bit. */
int mp_model_hazard_report(const mp_model* m, int64_t* out4, char* msg, int msg_cap);
/* The tracker by itself (no device involved; used by the CPU tests): streams and events are small integers. */
typedef struct mp_hazard mp_hazard;
mp_hazard* mp_hazard_create(void);
void mp_hazard_destroy(mp_hazard* h);
int mp_hazard_launch(mp_hazard* h, int stream, const char* name, int n, const int64_t* addr, const int64_t* bytes, const int* is_write);
int mp_hazard_record(mp_hazard* h, int event, int stream);
int mp_hazard_wait(mp_hazard* h, int stream, int event);
int mp_hazard_report(const mp_hazard* h, int64_t* out4, char* msg, int msg_cap);
/* intermediate outputs of the last forward (device pointers owned by the model): 0 = head output
 * (K, B*T*17, O), 1 = segment lengths (B, 16), 2 = the DropPath multipliers of the last train-mode forward (layout: mp_model_mask_info);
 * the fp32 residual stream block by block (blocks in execution order STE0, TTE0, STE1, ...; (B*T*N, C) each): 100 + 2 l = after the
 * attention branch of block l of the rotations net, 101 + 2 l = after its MLP branch, 99 = its embedding output; 300 + 2 l, 301 + 2 l,
 * 299 the same for the segments net; 500 / 501 / 502 = the 2-byte gradient operands dz (M x 2C) / dqkv (M x 3C) / residual-gradient copy (M x C) of the
 * LAST block the last backward differentiated (bf16, or scaled fp16 for dz / dqkv of an f16_backward model; numel counts floats = element pairs) */
int mp_model_peek(const mp_model* m, int which, const float** ptr, int64_t* numel);
/* copy `numel` floats of intermediate `which` into dst (device) on `stream` */
int mp_model_peek_copy(const mp_model* m, int which, float* dst, int64_t numel, void* stream);

/* per-kernel-class device timing (HIP events on the stream each kernel is launched on): classes 0 gemm_fwd, 1 gemm_dgrad,
 * 2 gemm_wgrad, 3 attention, 4 layernorm, 5 other partition the launches; class 6 gemm_persist is the SUBSET of classes 0/1 that
 * ran gemm_bf16_persist_kernel (the kernel with the largest share of a training step: its average launch time is what
 * rocprofv3 reports for that kernel name).  collect() synchronises the events, adds up elapsed ms / launch counts /
 * algorithmic FLOPs (and, for the forward / dgrad GEMM classes, algorithmic bytes: every operand read once, every output written
 * once) per class since the last reset and resets. */
#define MP_PROF_CLASSES 7
/* The Linear GEMM launches of the same interval by KIND = module * 12 + direction * 4 + layer (module 0 rotations / 1 segments net;
 * direction 0 forward, 1 dgrad, 2 weight gradient; layer 0 qkv, 1 proj, 2 fc1, 3 fc2: architectures/mix_ste.py:216-222,257-261,280-281):
 * what mp_prof_collect added up at its last call - elapsed ms, launches, how many of them ran gemm_bf16_persist_kernel, issued matrix-core
 * FLOPs, algorithmic bytes (forward / dgrad kinds) and 2 M N K - so that a forward instantiation running at 0.14 of the matrix peak is not
 * averaged with a dgrad running at 0.38.  Arrays of MP_PROF_KINDS entries; all but ms / launches may be NULL. */
#define MP_PROF_KINDS 24
int mp_prof_enable(mp_model* m, int on);
int mp_prof_collect(mp_model* m, double* ms, int64_t* launches, double* flops, double* bytes /* nullable: algorithmic bytes, GEMM classes */,
                    double* model_flops /* nullable: 2 M N K of the mathematical products; `flops` counts the matrix-core work issued, 3x that for the
                                         * split-precision forward */);
int mp_prof_kinds(const mp_model* m, double* ms, int64_t* launches, int64_t* persist_launches, double* flops, double* bytes, double* model_flops);

/* GPU-resident PoseSequenceGenerator (hpe/mh_so3_hpe/data/generators.py:44-219) + PoseFlip
 * (hpe/mh_so3_hpe/augmentations/transforms.py:7-28, functional.py:7-31): cuts B windows of T frames out of pose sequences stored
 * back to back in device memory.  poses_2d (Ntot,J,2), poses_3d (Ntot,J,3); seq_offset (S+1) device int64: first frame of each
 * sequence; win_seq / win_start (B) device int32: sequence and first frame (within it) of every window - the reference's
 * _map_index_to_pose / _map_index_to_frame entries or its random start; frames past the end of the sequence replicate its last
 * frame (the drop_last=False padding); win_flip (B) device bytes or null: windows to mirror (u / x negated, joint j read from
 * mirror[j]); mirror (J) HOST int32.  mask2d (B,T,J) device floats or null: multipliers of the 2-D input (the occlusion patterns
 * of generators.py:171-216); noise2d (B,T,J,2) or null: noise added to the 2-D input before the mask (miss_type "noisy").
 * Outputs X (B,T,J,2), y (B,T,J,3). */
int mp_gather_windows(const float* poses_2d, const float* poses_3d, const int64_t* seq_offset, int S, const int32_t* win_seq,
                      const int32_t* win_start, const uint8_t* win_flip, const int32_t* mirror, const float* mask2d, const float* noise2d,
                      int B, int T, int J, float* X, float* y, void* stream);

/* Dataset ingest: the raw arrays of the reference's on-disk formats -> the resident sequences mp_gather_windows reads.
 * mp_ingest_pose3d: raw (frames_raw, raw_joints, 3) device floats; frames (N) device int32 or null (null: the first N raw frames;
 * otherwise the raw frame of every output frame - temporal stride, valid-frame selection); joint_map (J <= 32) HOST int32 or null:
 * raw joint of every output joint.  out[n][j] = (T(raw[map[j]] - raw[root_raw]) - T(raw[map[root_out]])) / divisor, where a negative
 * root index drops its term and T is the world-to-camera transform qrot(qinverse(orientation), . - translation) when
 * orientation (4, w-first, HOST) / translation (3, HOST) are given, identity when both are null.
 *   Human3.6M (hpe/mh_so3_hpe/data/h36m_lifting.py:620-660 joint selection; data/utils.py:29-58 read_3d_data;
 *   data/camera.py:24-28; data/quaternion.py:6-31): map = the 17 kept joints, camera given, root_raw -1, root_out 0, divisor 1.
 *   MPI-INF-3DHP (data/dataset_3dhp.py:153-176,185-203): map = MAP_H36M_TO_MPI_JOINTS, no camera, root_raw 14, root_out -1,
 *   divisor 1000, frames = the valid test frames.
 * mp_ingest_pose2d: raw (frames_raw, raw_joints, raw_channels >= 2) pixel keypoints -> out (N, J, 2) = X / w * 2 - [1, h / w]
 * (data/camera.py:9-14 as called by data/utils.py:9-26 and dataset_3dhp.py:170-175,212-224; the subtraction runs in double as in
 * the reference, where a float64 list is subtracted from the float32 array). */
int mp_ingest_pose3d(const float* raw, int raw_joints, const int32_t* frames, int64_t N, const int32_t* joint_map, int J,
                     const float* orientation, const float* translation, int root_raw, int root_out, float divisor, float* out,
                     void* stream);
int mp_ingest_pose2d(const float* raw, int raw_joints, int raw_channels, const int32_t* frames, int64_t N, const int32_t* joint_map,
                     int J, float res_w, float res_h, float* out, void* stream);

/* Evaluation analytics of pose sequences in one pass over the frames (17-joint H36M / 3DHP tree compiled in): the running sums
 * behind mpjpe_error / mse_error / jointwise_error / segments_len_err (hpe/mh_so3_hpe/metrics/mean_joint_errors.py:31-130),
 * sagittal_symmetry(_per_bone) and segments_time_consistency(_per_bone) (metrics/regularizations.py:8-157), the evaluation form of
 * mean_velocity_error (metrics/losses.py:75-101) and keypoint_3d_pck / keypoint_3d_auc (metrics/pck.py:92-199; alignment 'none',
 * or 'scale' with scale_align = 1).  pred / gt are addressed through ELEMENT strides of (b, t, j, c), so the reference's
 * (B,3,J,L) permutations need no copy; gt (and its strides) may be null for the prediction-only metrics; mask: (B,L,J) bytes or null.
 * out: (B, mp_pose_metrics_row_floats()) sums per batch item, laid out as
 *   [0] sum ||e||  [1] sum ||e||^2  [2] sum_pairs |l-r|  [3] sum_pairs (l-r)^2  [4] sum_bones |gt-pred|  [5] sum_bones (gt-pred)
 *   [6] #(||e|| < pck_threshold)  [7] sum_j #(AUC thresholds i*auc_max/(auc_steps-1) above ||e||)  [8] #visible joints
 *   [9] sum ||d_t pred - d_t gt||  [10] the same squared  [11] frames;
 *   then per bone k (16): sum (len-len0), sum (len-len0)^2, sum |gt-pred|, sum (gt-pred);  per left/right pair (6): sum |l-r|,
 *   sum (l-r)^2;  per joint (17): sum ||e||, sum ||e||^2.   len0: (B,16) bone lengths of frame 0 (the shift of the variance sums).
 * scratch: >= B * ceil(L/128) * row_floats floats. */
int mp_pose_metrics_row_floats(void);
/* segments_len_err(..., mode="no_agg") (metrics/mean_joint_errors.py:83-130): out (B*L, 16) = ground-truth minus predicted bone length per
 * frame and bone (absolute value unless signed_diff); strides as in mp_pose_metrics. */
int mp_bone_length_table(const float* pred, const int64_t* pred_strides, const float* gt, const int64_t* gt_strides, int B, int L, int signed_diff,
                         float* out, void* stream);
int mp_pose_metrics(const float* pred, const int64_t* pred_strides, const float* gt, const int64_t* gt_strides, const uint8_t* mask, int B,
                    int L, int J, float pred_scale, float gt_scale, float pck_threshold, float auc_max, int auc_steps, int scale_align,
                    float* out, float* len0, float* scratch, int64_t scratch_floats, void* stream);

/* Procrustes-aligned errors: per frame the similarity transform (scale, proper rotation, translation) taking the predicted joints onto
 * the target ones in the least-squares sense - p_mpjpe (hpe/mh_so3_hpe/metrics/mean_joint_errors.py:148-189, batched numpy SVD on the
 * host in the reference) and the 'procrustes' alignment of keypoint_3d_pck / keypoint_3d_auc (metrics/pck.py:5-60,127-131) - solved on
 * the device with Horn's quaternion closed form (thread per frame, 4x4 Jacobi).  pred, gt: (N,17,3) contiguous; mask (N,17) bytes or
 * null.  out[5]: sum of aligned per-joint errors, #(error < pck_threshold), sum of AUC threshold counts, #visible joints, frames.
 * scratch: >= 5 * ceil(N/256) floats. */
int mp_procrustes_errors(const float* pred, const float* gt, const uint8_t* mask, int64_t N, int J, float pred_scale, float gt_scale,
                         float pck_threshold, float auc_max, int auc_steps, float* out, float* scratch, int64_t scratch_floats, void* stream);

/* test / tuning hooks (no reference counterpart; process-wide, they select between kernels that the GPU tests hold to the same results):
 * "gemm_small_tile" (1 = 128x128 GEMM tiles everywhere), "gemm_persist_min_tiles" (output-tile count from which the persistent GEMM
 * kernels are used; 0 = default), "gemm_persist_mode" (0 tiled kernels only, 1 = default: persistent kernel where it applies),
 * "gemm_persist_wgs" (workgroups = CUs the persistent GEMMs occupy; 0 = default: all), "attn_two_phase" (0 = the one-strip-at-a-time
 * split-precision temporal attention forward for every shape; 1 = default: the two-phase kernel for head dim 64 and T > 128),
 * "heads_mfma" (0 = row kernels for the output heads, 1 = default: matrix cores wherever covered, 2 = only from 16 outputs up).
 * Everything that changes a model's arithmetic or stream use is a field of mp_model_config. */
int mp_set_option(const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif /* MANIPOSE_HIP_H */
