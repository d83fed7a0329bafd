// extern "C" shims of the stand-alone operators declared in include/manipose_hip.h
#include <string.h>
#include "common.h"
#include "kernels.h"
#include "../../include/manipose_hip.h"

namespace mp {
const char* last_error();
long wgrad_f32_slab_floats(int Mtok, int Nout, int Kin);
}
using namespace mp;

static LossCfg to_cfg(const mp_loss_config* c) {
  LossCfg l;
  l.beta = c->rmcl_score_reg; l.vel_w = c->vel_loss; l.smooth_w = c->smooth_reg; l.use_joint_weights = c->w_loss; l.squared = c->sq_loss;
  for (int j = 0; j < 17; ++j) l.joint_weights[j] = c->joint_weights[j];
  return l;
}

extern "C" {

int mp_abi_version(void) { return MP_ABI_VERSION; }
const char* mp_last_error(void) { return mp::last_error(); }

int mp_fk_decode_fwd(const float* rot6d, int rot_stride, int rot_dim, const float* lengths, float* poses, int B, int K, int T,
                     void* stream) {
  MP_CHECK(rot6d && lengths && poses, MP_ERR_ARG, "mp_fk_decode_fwd: null pointer");
  return fk_decode_fwd(rot6d, rot_stride, rot_dim, lengths, poses, B, K, T, (hipStream_t)stream);
}
int mp_fk_decode_bwd(const float* rot6d, int rot_stride, int rot_dim, const float* lengths, const float* d_poses, float* d_rot6d,
                     float* d_len_pose, int B, int K, int T, void* stream) {
  MP_CHECK(rot6d && lengths && d_poses && d_rot6d && d_len_pose, MP_ERR_ARG, "mp_fk_decode_bwd: null pointer");
  return fk_decode_bwd(rot6d, rot_stride, rot_dim, lengths, d_poses, d_rot6d, d_len_pose, B, K, T, (hipStream_t)stream);
}

int mp_wta_loss(const float* poses, const float* scores, const float* target, const mp_loss_config* cfg, float* terms,
                int32_t* argmin, float* d_poses, float* d_scores, int B, int K, int T, float* scratch, int64_t scratch_floats,
                void* stream) {
  MP_CHECK(poses && scores && target && cfg && terms && scratch, MP_ERR_ARG, "mp_wta_loss: null pointer");
  return wta_loss(poses, scores, target, to_cfg(cfg), terms, argmin, d_poses, d_scores, B, K, T, scratch, scratch_floats,
                  (hipStream_t)stream);
}
int mp_single_loss(const float* poses, const float* target, const mp_loss_config* cfg, float* terms, float* d_poses, int B, int T,
                   float* scratch, int64_t scratch_floats, void* stream) {
  MP_CHECK(poses && target && cfg && terms && scratch, MP_ERR_ARG, "mp_single_loss: null pointer");
  return single_loss(poses, target, to_cfg(cfg), terms, d_poses, B, T, scratch, scratch_floats, (hipStream_t)stream);
}
int mp_rigid_segments_loss(const float* poses, float weight, float* term, float* d_poses, int B, int T, float* scratch, int64_t scratch_floats,
                           void* stream) {
  return rigid_segments_loss(poses, weight, term, d_poses, B, T, scratch, (long)scratch_floats, (hipStream_t)stream);
}

int mp_aggregate(const float* poses, const float* scores, const float* target, int mode, float* out, int B, int K, int T,
                 void* stream) {
  MP_CHECK(poses && out, MP_ERR_ARG, "mp_aggregate: null pointer");
  return aggregate_poses(poses, scores, target, mode, out, B, K, T, (hipStream_t)stream);
}
int mp_mpjpe_sum(const float* pred, const float* target, int64_t n_joints, float* out_sum, float* scratch, int64_t scratch_floats,
                 void* stream) {
  MP_CHECK(pred && target && out_sum && scratch && n_joints > 0, MP_ERR_ARG, "mp_mpjpe_sum: bad argument");
  return mpjpe_sum(pred, target, n_joints, out_sum, scratch, scratch_floats, (hipStream_t)stream);
}
int mp_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int step, float lr, float beta1,
                 float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
  MP_CHECK(params && grads && exp_avg && exp_avg_sq && n > 0, MP_ERR_ARG, "mp_adam_step: bad argument");
  return adam_step(params, grads, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, weight_decay, grad_scale,
                   (hipStream_t)stream);
}

int mp_adam_step_scaled(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int step, float lr, float beta1,
                        float beta2, float eps, float weight_decay, float grad_scale, const float* lr_mult, const float* wd_mult, void* stream) {
  MP_CHECK(params && grads && exp_avg && exp_avg_sq && lr_mult && wd_mult && n > 0, MP_ERR_ARG, "mp_adam_step_scaled: bad argument");
  return adam_step(params, grads, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, weight_decay, grad_scale, (hipStream_t)stream, lr_mult,
                   wd_mult);
}

int mp_layernorm_fwd(const float* x, const float* gamma, const float* beta, float eps, float* y, float* stats, int M, int C,
                     void* stream) {
  MP_CHECK(x && gamma && beta && y && stats, MP_ERR_ARG, "mp_layernorm_fwd: null pointer");
  LnFwdArgs a = {};
  a.x = x; a.M = M; a.C = C; a.g2 = gamma; a.b2 = beta; a.eps2 = eps; a.y2 = y; a.stats2 = stats;
  return ln_fwd(a, 0, (hipStream_t)stream);
}
int mp_layernorm_bwd(const float* dy, const float* x, const float* stats, const float* gamma, const float* dskip, float* dx,
                     float* dgamma, float* dbeta, int M, int C, float* scratch, int64_t scratch_floats, void* stream) {
  MP_CHECK(dy && x && stats && gamma && dx && dgamma && dbeta && scratch, MP_ERR_ARG, "mp_layernorm_bwd: null pointer");
  return ln_bwd(dy, 0, x, stats, gamma, dskip, dx, nullptr, nullptr, 0, 1, 1, dgamma, dbeta, M, C, scratch, scratch_floats, (hipStream_t)stream);
}

int mp_linear_fwd(const float* x, const float* W, const float* b, float* y, float* z, const float* r, int M, int N, int K,
                  int epilogue, void* stream) {
  MP_CHECK(x && W && y, MP_ERR_ARG, "mp_linear_fwd: null pointer");
  MP_CHECK(epilogue >= 0 && epilogue <= 2, MP_ERR_ARG, "mp_linear_fwd: epilogue %d", epilogue);
  MP_CHECK(epilogue != 1 || z, MP_ERR_ARG, "mp_linear_fwd: GELU epilogue needs z");
  MP_CHECK(epilogue != 2 || r, MP_ERR_ARG, "mp_linear_fwd: residual epilogue needs r");
  GemmF32Args g = {};
  g.A = x; g.lda = K; g.B = W; g.ldb = K; g.C = y; g.ldc = N; g.M = M; g.N = N; g.K = K; g.bias = b; g.Z = z; g.R = r;
  return gemm_f32(0, 0, epilogue == 0 ? EPI_BIAS : (epilogue == 1 ? EPI_BIAS_GELU : EPI_BIAS_RESID), g, (hipStream_t)stream);
}
int64_t mp_linear_bwd_slab_floats(int N, int K) { return wgrad_f32_slab_floats(0, N, K) + (int64_t)N * K + N; }
int mp_linear_bwd(const float* dy, const float* x, const float* W, float* dx, float* dW, float* db, int M, int N, int K,
                  float* slab, int64_t slab_floats, void* stream) {
  MP_CHECK(dy && x && W && dW && slab, MP_ERR_ARG, "mp_linear_bwd: null pointer");
  if (dx) {
    GemmF32Args g = {};
    g.A = dy; g.lda = N; g.B = W; g.ldb = K; g.C = dx; g.ldc = K; g.M = M; g.N = K; g.K = N;
    int rc = gemm_f32(0, 1, EPI_BIAS, g, (hipStream_t)stream);
    if (rc) return rc;
  }
  return wgrad_f32(dy, N, x, K, M, N, K, dW, db, slab, slab_floats, (hipStream_t)stream);
}

int mp_linear_fwd_bf16(const void* x, const void* W, const float* b, void* y, void* z, const float* r, int M, int N, int K,
                       int epilogue, void* stream) {
  MP_CHECK(x && W && y && epilogue >= 0 && epilogue <= 2, MP_ERR_ARG, "mp_linear_fwd_bf16: bad argument");
  MP_CHECK((epilogue != 1 || z) && (epilogue != 2 || r), MP_ERR_ARG, "mp_linear_fwd_bf16: epilogue operand missing");
  GemmB16Args g = {};
  g.A = x; g.lda = K; g.B = W; g.ldb = K; g.C = y; g.ldc = N; g.M = M; g.N = N; g.K = K; g.bias = b; g.Z = z; g.R = r;
  const int epi = epilogue == 0 ? EPI_BIAS : (epilogue == 1 ? EPI_BIAS_GELU : EPI_BIAS_RESID);
  return gemm_bf16(g, 0, 0, 0, epilogue == 2, epi, (hipStream_t)stream);
}
int mp_linear_bwd_bf16(const void* dy, int dy_f32, const void* x, const void* W, void* dx, int dx_f32, float* dW, float* db,
                       int M, int N, int K, float* slab, int64_t slab_floats, void* stream) {
  MP_CHECK(dy && x && W && dW && slab, MP_ERR_ARG, "mp_linear_bwd_bf16: null pointer");
  if (dx) {
    GemmB16Args g = {};
    g.A = dy; g.lda = N; g.B = W; g.ldb = K; g.C = dx; g.ldc = K; g.M = M; g.N = K; g.K = N;
    int rc = gemm_bf16(g, dy_f32, 0, 1, dx_f32, EPI_BIAS, (hipStream_t)stream);
    if (rc) return rc;
  }
  return wgrad_bf16(dy, dy_f32, N, (const bf16*)x, K, M, N, K, dW, db, slab, slab_floats, (hipStream_t)stream);
}

int mp_attention_fwd(const float* qkv, float* out, float* lse, int temporal, int B, int T, int J, int C, int H, void* stream) {
  MP_CHECK(qkv && out && (!temporal || lse), MP_ERR_ARG, "mp_attention_fwd: null pointer");
  return temporal ? attn_temporal_fwd(qkv, out, lse, 0, B, T, J, C, H, (hipStream_t)stream)
                  : attn_spatial_fwd(qkv, out, 0, B, T, J, C, H, (hipStream_t)stream);
}
int mp_attention_bwd(const float* qkv, const float* out, const float* d_out, const float* lse, float* delta, float* d_qkv,
                     int temporal, int B, int T, int J, int C, int H, void* stream) {
  MP_CHECK(qkv && d_out && d_qkv && (!temporal || (out && lse && delta)), MP_ERR_ARG, "mp_attention_bwd: null pointer");
  return temporal ? attn_temporal_bwd(qkv, out, d_out, lse, delta, d_qkv, 0, B, T, J, C, H, (hipStream_t)stream)
                  : attn_spatial_bwd(qkv, d_out, d_qkv, 0, B, T, J, C, H, (hipStream_t)stream);
}

int mp_attention_fwd_bf16(const void* qkv, void* out, float* lse, int temporal, int B, int T, int J, int C, int H, void* stream) {
  MP_CHECK(qkv && out && (!temporal || lse), MP_ERR_ARG, "mp_attention_fwd_bf16: null pointer");
  return temporal ? attn_temporal_fwd(qkv, out, lse, 1, B, T, J, C, H, (hipStream_t)stream)
                  : attn_spatial_fwd(qkv, out, 1, B, T, J, C, H, (hipStream_t)stream);
}
int mp_attention_bwd_bf16(const void* qkv, const void* out, const void* d_out, const float* lse, float* delta, void* d_qkv,
                          int temporal, int B, int T, int J, int C, int H, void* stream) {
  MP_CHECK(qkv && d_out && d_qkv && (!temporal || (out && lse && delta)), MP_ERR_ARG, "mp_attention_bwd_bf16: null pointer");
  return temporal ? attn_temporal_bwd(qkv, out, d_out, lse, delta, d_qkv, 1, B, T, J, C, H, (hipStream_t)stream)
                  : attn_spatial_bwd(qkv, d_out, d_qkv, 1, B, T, J, C, H, (hipStream_t)stream);
}

/* split precision ("bf16x3"): planar hi/lo bf16 operands, three matrix-core products per k-tile */
int mp_split_bf16(const float* src, void* hi, void* lo, int64_t n, void* stream) {
  MP_CHECK(src && hi && lo && n > 0, MP_ERR_ARG, "mp_split_bf16: bad argument");
  return split_planes(src, (bf16*)hi, (bf16*)lo, (long)n, (hipStream_t)stream);
}
int mp_linear_fwd_bf16x3(const void* x_hi, const void* x_lo, const void* W_hi, const void* W_lo, const float* b, void* y, void* y_lo,
                         void* z, const float* r, int M, int N, int K, int epilogue, void* stream) {
  MP_CHECK(x_hi && x_lo && W_hi && W_lo && y && epilogue >= 0 && epilogue <= 2, MP_ERR_ARG, "mp_linear_fwd_bf16x3: bad argument");
  MP_CHECK((epilogue == 2 || y_lo) && (epilogue != 1 || z) && (epilogue != 2 || r), MP_ERR_ARG, "mp_linear_fwd_bf16x3: epilogue operand missing");
  GemmB16Args g = {};
  g.A = x_hi; g.A_lo = x_lo; g.lda = K; g.B = W_hi; g.B_lo = W_lo; g.ldb = K; g.C = y; g.C_lo = y_lo; g.ldc = N; g.M = M; g.N = N; g.K = K;
  g.bias = b; g.Z = z; g.R = r;
  const int epi = epilogue == 0 ? EPI_BIAS : (epilogue == 1 ? EPI_BIAS_GELU : EPI_BIAS_RESID);
  return gemm_bf16x3(g, epilogue == 2, epi, (hipStream_t)stream);
}
int mp_linear_fwd_bf16x3_lnres(const void* x_hi, const void* x_lo, const void* W_hi, const void* W_lo, const float* b, float* y,
                               const float* r_in, const float* rstats, const float* rgamma, const float* rbeta, const float* mask, int mask_mode,
                               int T, int J, int M, int N, int K, void* stream) {
  MP_CHECK(x_hi && x_lo && W_hi && W_lo && y && r_in && rstats && rgamma && rbeta, MP_ERR_ARG, "mp_linear_fwd_bf16x3_lnres: null argument");
  MP_CHECK(mask == nullptr || ((mask_mode == 1 || mask_mode == 2) && T > 0 && J > 0 && M % (T * J) == 0), MP_ERR_ARG,
           "mp_linear_fwd_bf16x3_lnres: a DropPath mask needs mask_mode 1 or 2 and M a multiple of T*J");
  MP_CHECK(y != r_in, MP_ERR_ARG, "mp_linear_fwd_bf16x3_lnres: y and r_in may not alias");
  GemmB16Args g = {};
  g.A = x_hi; g.A_lo = x_lo; g.lda = K; g.B = W_hi; g.B_lo = W_lo; g.ldb = K; g.C = y; g.ldc = N; g.M = M; g.N = N; g.K = K;
  g.bias = b; g.R = r_in; g.rstats = rstats; g.rgamma = rgamma; g.rbeta = rbeta;
  g.mask = mask; g.mask_mode = mask ? mask_mode : 0; g.T = T; g.J = J;
  return gemm_bf16x3(g, 1, EPI_BIAS_RESID, (hipStream_t)stream);
}
int mp_split_f16f8(const float* src, void* hi16, void* corr8, int64_t n, int weight, void* stream) {
  MP_CHECK(src && hi16 && corr8 && n > 0 && n % 64 == 0, MP_ERR_ARG, "mp_split_f16f8: bad argument (n must be a multiple of 64)");
  return cast_to_f16f8(src, hi16, corr8, (long)n, weight, (hipStream_t)stream);
}
int mp_linear_fwd_f16f8(const void* x16, const void* x8, const void* W16, const void* W8, const float* b, float* y, int M, int N, int K,
                        void* stream) {
  MP_CHECK(x16 && x8 && W16 && W8 && y, MP_ERR_ARG, "mp_linear_fwd_f16f8: null argument");
  GemmB16Args g = {};
  g.A = x16; g.A_lo = x8; g.lda = K; g.B = W16; g.B_lo = W8; g.ldb = K; g.C = y; g.ldc = N; g.M = M; g.N = N; g.K = K; g.bias = b;
  return gemm_f16f8(g, 1, EPI_BIAS, (hipStream_t)stream);
}
int mp_attention_fwd_bf16x3(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, float* lse, float* scratch, int temporal,
                            int B, int T, int J, int C, int H, void* stream) {
  MP_CHECK(qkv_hi && qkv_lo && out_hi && out_lo && (!temporal || lse), MP_ERR_ARG, "mp_attention_fwd_bf16x3: null pointer");
  return temporal ? attn_temporal_fwd_x3((const bf16*)qkv_hi, (const bf16*)qkv_lo, (bf16*)out_hi, (bf16*)out_lo, lse, scratch, B, T, J, C, H,
                                         (hipStream_t)stream)
                  : attn_spatial_fwd_x3((const bf16*)qkv_hi, (const bf16*)qkv_lo, (bf16*)out_hi, (bf16*)out_lo, scratch, B, T, J, C, H,
                                        (hipStream_t)stream);
}

static int heads_pick(int impl, int K, int O, int C, const char* who, bool* mfma) {
  MP_CHECK(impl >= 0 && impl <= 2, MP_ERR_ARG, "%s: impl %d", who, impl);
  MP_CHECK(K >= 1 && K <= 8 && O >= 1, MP_ERR_ARG, "%s: K=%d O=%d unsupported", who, K, O);
  *mfma = impl == 2 || (impl == 0 && heads_use_mfma(K, O, C));
  MP_CHECK(!*mfma || heads_mfma_supported(K, O, C), MP_ERR_ARG, "%s: K=%d O=%d C=%d not covered by the matrix-core heads", who, K, O, C);
  return MP_OK;
}
int64_t mp_heads_fold_floats(int C) { return heads_fold_floats(C); }
int64_t mp_heads_bwd_scratch_floats(int K, int O, int C) { return 512L * K * ((long)O * C + O + 2 * C) + 256L * heads_fold_floats(C); }
int mp_heads_fwd(const float* x, const float* gamma, const float* beta, const float* W, const float* b, int K, int O, float* out, float* stats,
                 float* fold, int M, int C, int impl, void* stream) {
  MP_CHECK(x && gamma && beta && W && b && out && stats && fold && M > 0, MP_ERR_ARG, "mp_heads_fwd: bad argument");
  bool mfma = false;
  if (int rc = heads_pick(impl, K, O, C, "mp_heads_fwd", &mfma)) return rc;
  HeadParams p = {};
  for (int k = 0; k < K; ++k) { p.gamma[k] = gamma + (long)k * C; p.beta[k] = beta + (long)k * C; p.W[k] = W + (long)k * O * C; p.b[k] = b + (long)k * O; }
  return mfma ? heads_fwd_mfma(x, p, K, O, out, stats, M, C, fold, (hipStream_t)stream) : heads_fwd(x, p, K, O, out, stats, M, C, (hipStream_t)stream);
}
int mp_heads_bwd(const float* x, const float* stats, const float* fold, const float* out, const float* gamma, const float* beta, const float* W,
                 const float* b, const float* d_out, float* dx, float* dgamma, float* dbeta, float* dW, float* db, int K, int O, int M, int C, int impl,
                 float* scratch, int64_t scratch_floats, void* stream) {
  MP_CHECK(x && stats && fold && out && gamma && beta && W && b && d_out && dx && dgamma && dbeta && dW && db && scratch && M > 0, MP_ERR_ARG,
           "mp_heads_bwd: bad argument");
  bool mfma = false;
  if (int rc = heads_pick(impl, K, O, C, "mp_heads_bwd", &mfma)) return rc;
  HeadParams p = {};
  HeadGrads g = {};
  for (int k = 0; k < K; ++k) {
    p.gamma[k] = gamma + (long)k * C; p.beta[k] = beta + (long)k * C; p.W[k] = W + (long)k * O * C; p.b[k] = b + (long)k * O;
    g.gamma[k] = dgamma + (long)k * C; g.beta[k] = dbeta + (long)k * C; g.W[k] = dW + (long)k * O * C; g.b[k] = db + (long)k * O;
  }
  return mfma ? heads_bwd_mfma(x, stats, fold, out, p, g, K, O, d_out, dx, M, C, scratch, scratch_floats, (hipStream_t)stream, nullptr)
              : heads_bwd(x, stats, p, g, K, O, d_out, dx, M, C, scratch, scratch_floats, (hipStream_t)stream);
}

int mp_gather_windows(const float* poses_2d, const float* poses_3d, const int64_t* seq_offset, int S, const int32_t* win_seq,
                      const int32_t* win_start, const uint8_t* win_flip, const int32_t* mirror, const float* mask2d, const float* noise2d,
                      int B, int T, int J, float* X, float* y, void* stream) {
  static_assert(sizeof(long) == sizeof(int64_t), "LP64");
  return gather_windows(poses_2d, poses_3d, (const long*)seq_offset, S, win_seq, win_start, win_flip, mirror, mask2d, noise2d, B, T, J,
                        X, y, (hipStream_t)stream);
}

int mp_ingest_pose3d(const float* raw, int raw_joints, const int32_t* frames, int64_t N, const int32_t* joint_map, int J,
                     const float* orientation, const float* translation, int root_raw, int root_out, float divisor, float* out,
                     void* stream) {
  return ingest_pose3d(raw, raw_joints, frames, (long)N, joint_map, J, orientation, translation, root_raw, root_out, divisor, out,
                       (hipStream_t)stream);
}

int mp_ingest_pose2d(const float* raw, int raw_joints, int raw_channels, const int32_t* frames, int64_t N, const int32_t* joint_map,
                     int J, float res_w, float res_h, float* out, void* stream) {
  return ingest_pose2d(raw, raw_joints, raw_channels, frames, (long)N, joint_map, J, res_w, res_h, out, (hipStream_t)stream);
}

int mp_procrustes_errors(const float* pred, const float* gt, const uint8_t* mask, int64_t N, int J, float pred_scale, float gt_scale,
                         float pck_threshold, float auc_max, int auc_steps, float* out, float* scratch, int64_t scratch_floats, void* stream) {
  return procrustes_errors(pred, gt, mask, (long)N, J, pred_scale, gt_scale, pck_threshold, auc_max, auc_steps, 1, out, scratch,
                           (long)scratch_floats, (hipStream_t)stream);
}

int mp_pose_metrics_row_floats(void) { return pose_metrics_row_floats(); }
int mp_bone_length_table(const float* pred, const int64_t* pred_strides, const float* gt, const int64_t* gt_strides, int B, int L, int signed_diff,
                         float* out, void* stream) {
  MP_CHECK(pred && pred_strides && gt && gt_strides && out, MP_ERR_ARG, "mp_bone_length_table: null pointer");
  long ps[4], gs[4];
  for (int i = 0; i < 4; ++i) { ps[i] = (long)pred_strides[i]; gs[i] = (long)gt_strides[i]; }
  return bone_length_table(pred, ps, gt, gs, B, L, signed_diff, out, (hipStream_t)stream);
}
int mp_pose_metrics(const float* pred, const int64_t* pred_strides, const float* gt, const int64_t* gt_strides, const uint8_t* mask, int B,
                    int L, int J, float pred_scale, float gt_scale, float pck_threshold, float auc_max, int auc_steps, int scale_align,
                    float* out, float* len0, float* scratch, int64_t scratch_floats, void* stream) {
  MP_CHECK(pred && pred_strides && out && len0 && scratch, MP_ERR_ARG, "mp_pose_metrics: null pointer");
  long ps[4], gs[4] = {0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) { ps[i] = (long)pred_strides[i]; if (gt_strides) gs[i] = (long)gt_strides[i]; }
  return pose_metrics(pred, ps, gt, gt_strides ? gs : nullptr, mask, B, L, J, pred_scale, gt_scale, pck_threshold, auc_max, auc_steps,
                      scale_align, out, len0, scratch, (long)scratch_floats, (hipStream_t)stream);
}

/* test / tuning hooks (include/manipose_hip.h): process-wide selectors between kernels that are tested to agree; everything that changes a
 * model's arithmetic or its stream use is a field of mp_model_config */
int mp_set_option(const char* name, int value) {
  MP_CHECK(name, MP_ERR_ARG, "mp_set_option: null name");
  if (!strcmp(name, "gemm_small_tile")) { gemm_bf16_force_small_tile(value != 0); return MP_OK; }
  if (!strcmp(name, "gemm_persist_min_tiles")) { gemm_bf16_persist_min_tiles(value); return MP_OK; }
  if (!strcmp(name, "gemm_persist_mode")) { gemm_bf16_persist_mode(value); return MP_OK; }
  if (!strcmp(name, "gemm_persist_wgs")) { gemm_bf16_persist_wgs(value); return MP_OK; }
  if (!strcmp(name, "attn_two_phase")) { attn_two_phase(value); return MP_OK; }
  if (!strcmp(name, "heads_mfma")) { heads_mfma_mode(value); return MP_OK; }
  MP_CHECK(false, MP_ERR_ARG, "mp_set_option: unknown option '%s' (side_streams / f16f8_inputs became mp_model_config::streams / f16f8 in ABI v7)", name);
}

}  // extern "C"
