// Internal launcher declarations shared by the engine and the C-ABI shims.  All pointers are device
// pointers; all launchers enqueue on `st` and return MP_OK or an error code with mp::set_error() filled in.
#pragma once
#include "common.h"

namespace mp {

// ---------------------------------------------------------------- GEMM (gemm_f32.hip)
enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RESID = 2, EPI_DGELU = 3, EPI_SLAB = 4 };
struct GemmF32Args {
  const float* A;
  const float* B;
  float* C;
  long lda, ldb, ldc;
  int M, N, K;
  const float* bias;   // per output column, may be null
  float* Z;            // EPI_BIAS_GELU: gelu'(pre-activation) (written); EPI_DGELU: the same (read)
  const float* R;      // EPI_BIAS_RESID: residual stream (read)
  const float* mask;   // EPI_BIAS_RESID: DropPath multipliers per sample, may be null
  float rscale;        // EPI_BIAS_RESID: C = rscale * R + mask * (A B + bias); 0 means 1 (muP residual scale 1 / sqrt(depth), mix_ste.py:330)
  int mask_mode, T, J; // see droppath_scale()
  float* bias_slab;    // EPI_SLAB
  int k_per_split;     // EPI_SLAB
};
int gemm_f32(int AL, int BL, int EPI, GemmF32Args g, hipStream_t st);
int wgrad_f32(const float* dY, long lddy, const float* X, long ldx, int Mtok, int Nout, int Kin, float* dW, float* db,
              float* slab, long slab_floats, hipStream_t st);

// ---------------------------------------------------------------- gemm_bf16.hip
struct GemmB16Args {
  const void* A;
  const void* B;
  void* C;
  long lda, ldb, ldc;
  int M, N, K;
  const float* bias;
  void* Z;             // gelu'(pre-activation) kept for the backward, element type of C
  const float* R;      // fp32 residual stream
  // residual epilogue with the residual recomputed instead of materialised: R holds the INPUT of a LayerNorm, the residual added is
  // (R - mean) * rstd * rgamma + rbeta with (mean, rstd) = rstats[2 row .. 2 row + 1]; all three null: R is added as it is
  const float* rstats;
  const float* rgamma;
  const float* rbeta;
  const float* mask;
  int mask_mode, T, J;
  float* bias_slab;
  int bias_parts;      // weight-gradient kernel, 256-wide tiles: the column tiles 0 .. bias_parts-1 of a row tile share its bias-gradient sums by k-tile (0 = 1)
  int k_per_split;
  int debug;           // timing-ablation bits (MANIPOSE_GEMM_DEBUG), 0 in production
  int stagger;         // diagnostics (MANIPOSE_GEMM_STAGGER): start delay per phase group of the persistent kernels, clock ticks
  long long* stamps;   // diagnostics (MANIPOSE_GEMM_STAMPS=<device address>): persistent kernels record [workgroup][tile < 64][epilogue start, end] in 10 ns ticks
  // split precision (gemm_bf16x3): lo planes of A, B and (planar outputs) C; A / B / C are the hi planes (common.h: bf16p)
  const void* A_lo;
  const void* B_lo;
  void* C_lo;
  float rscale;        // residual epilogue: C = rscale * R + mask * (A B + bias); 0 means 1 (muP, mix_ste.py:330)
  // fp16 operands (the dgrad / weight-gradient GEMMs of a layer whose gradient operand is carried as fp16 times a power of two): the same
  // kernels with v_mfma_f32_16x16x32_f16.  The dgrad's bf16 output keeps that scale (its consumer, a LayerNorm backward, divides it out: dy_scale);
  // the weight gradient's slab reduction multiplies by wgrad_bf16's oscale.
  int f16;
  int out_f16f8;       // gemm_f16f8 with EPI_BIAS_GELU: C / C_lo are the fp16 / correction planes of the "f16f8" format instead of bf16 hi / lo planes
  const float* gout;   // EPI_DGELU with a 2-byte output: non-null writes dz as fp16 of *gout x value instead of bf16 (device address; the operand format above)
  unsigned* gsat;      // with gout: the two counters of its saturating stores (clamped / non-finite elements; common.h sat_f16x4), may be null
};
int gemm_bf16(GemmB16Args g, int a_f32, int a_tr, int b_tr, int c_f32, int epi, hipStream_t st);
// C = A B^T on planar hi/lo operands ("N","N" layouts: the forward Linear), three bf16 MFMA products per k-tile, fp32 accumulate.
// epi EPI_BIAS / EPI_BIAS_GELU: C planar (C, C_lo), Z = gelu' as plain bf16; EPI_BIAS_RESID: C fp32 (c_f32 must be 1)
int gemm_bf16x3(GemmB16Args g, int c_f32, int epi, hipStream_t st);
// y = x W^T + b on fp16 hi planes + 8-bit correction planes (gemm_bf16.hip, mma_stage_f8): A / B the fp16 planes, A_lo / B_lo the corrections, C fp32
int gemm_f16f8(GemmB16Args g, int c_f32, int epi, hipStream_t st);
int cast_to_f16f8(const float* src, void* hi16, void* cat8, long n, int weight, hipStream_t st);     // n % 64 == 0; see common.h "f16f8"
int cast_to_bf16x2(const float* src, bf16* hi, bf16* lo, long n, hipStream_t st);
int wgrad_bf16(const void* dY, int dy_f32, long lddy, const bf16* X, long ldx, int Mtok, int Nout, int Kin, float* dW, float* db,
               float* slab, long slab_floats, hipStream_t st, int f16 = 0, const float* oscale = nullptr, int x_f16 = 0);      // f16: dY and X are fp16, dW += *oscale x dY^T X (device address); x_f16: X alone is fp16 (the fp16 plane of an f16f8 activation), rounded to bf16 per fragment
int cast_to_bf16(const float* src, bf16* dst, long n, hipStream_t st);
void gemm_bf16_force_small_tile(bool on);          // test hooks (mp_set_option)
void gemm_bf16_persist_min_tiles(int n);
void gemm_bf16_persist_mode(int mode);
void gemm_bf16_persist_wgs(int n);                 // workgroups (= CUs) of the persistent kernels; 0 = all
int gemm_bf16_take_last_persist();                  // 1 if this thread's last gemm_bf16() ran the persistent kernel (and clears it)

// ---------------------------------------------------------------- elementwise.hip
struct LnFwdArgs {
  const float* x;        // [M][C] input
  int M, C;
  // stage 1 (optional, g1 != null): x1 = LN(x; g1,b1,eps1) (+ pos[t(m)]) written fp32
  const float* g1; const float* b1; float eps1;
  const float* pos; int T, J;          // pos: [T][C] temporal positional table or null
  float* x1; float* stats1;            // stats: [M][2] = (mean, rstd)
  // stage 2 (optional, g2 != null): y2 = LN(stage-1 output or x; g2,b2,eps2)
  const float* g2; const float* b2; float eps2;
  void* y2; float* stats2;
  void* y2_lo;           // out mode 2 (planar hi/lo bf16, common.h): the lo plane; out mode 3: the 8-bit correction plane (2 C bytes per row)
  void* y2_b16;          // out mode 3 ("f16f8", common.h: y2 = the fp16 plane): a plain bf16 copy of the output as well (the backward's operand), or null
};
int ln_fwd(const LnFwdArgs& a, int out_mode /* 0 fp32, 1 bf16, 2 planar bf16 hi/lo, 3 f16f8 planes + bf16 copy (C % 64 == 0) */, hipStream_t st);
// dx = [dskip +] LN'(dy); partial param grads are reduced and ADDED into dgamma/dbeta.
// dx = [rs * dskip +] LN'(dy) (rs: the block's residual scale, 1 unless muP)
int ln_bwd(const void* dy, int dy_bf16, const float* x, const float* stats, const float* gamma, const float* dskip, float* dx, void* dx_b16,
           const float* mask, int mask_mode, int T, int J, float* dgamma, float* dbeta, int M, int C, float* scratch,
           long scratch_floats, hipStream_t st, hipStream_t st_param = nullptr, hipEvent_t ev = nullptr, float rs = 1.0f,
           const float* dy_scale = nullptr, const float* b16_gs = nullptr);      // device addresses: *dy_scale multiplies dy on load (null = 1); b16_gs non-null:
                                                                                 // dx_b16 is written as fp16 of *b16_gs x value instead of bf16
int ln_bwd2(const void* dy1, int dy_bf16, const float* x1, const float* stats1, const float* gamma1, const float* dskip,
            const float* x0, const float* stats0, const float* gamma0, const float* beta0 /* non-null: x1 == LN0(x0) is recomputed */,
            float* dx, void* dx_b16, const float* mask, int mask_mode, int T, int J, float* dgamma1, float* dbeta1, float* dgamma0, float* dbeta0, int M, int C, float* scratch, long scratch_floats,
            hipStream_t st, hipStream_t st_param = nullptr, hipEvent_t ev = nullptr, float rs = 1.0f, const float* dy_scale = nullptr,
            const float* b16_gs = nullptr);      // as in ln_bwd (dy_scale applies to dy1)
// gsc (8 floats on the device) <- {S, 1 / S, scratch, 1, 0u, 0u (saturation / non-finite counters of this backward's fp16 stores)} with S the power of two that brings the largest |element| of the two tensors (the second may be null) into [2^11, 2^12) (elementwise.hip)
int grad_scale(const float* d_poses, long n_poses, const float* d_scores, long n_scores, float* gsc, hipStream_t st);
int grad_health_pack(const float* gsc, float* out4, hipStream_t st);      // mp_model_grad_health_async: the four health values as floats (device to device)
// dst = s * src ; dst += s * src  (muP readout multiplier on the head weights / their gradients)
int scale_copy(float* dst, const float* src, float s, long n, hipStream_t st);
int axpy_scaled(float* dst, const float* src, float s, long n, hipStream_t st);

int embed_fwd(const float* xin, const float* W, const float* b, const float* spos, float* out, int M, int C, int J,
              hipStream_t st);
long embed_bwd_scratch_floats(int C, int J);
int embed_bwd(const float* g, const float* xin, float* dW, float* db, float* dspos, int M, int C, int J, float* scratch,
              long scratch_floats, hipStream_t st);
int bones_embed_fwd(const float* xin, const float* W, const float* b, const float* spos, float* out, int BT, int IN, int O,
                    hipStream_t st);
int bones_embed_bwd(const float* g, const float* xin, float* dW, float* db, float* dspos, int BT, int IN, int O,
                    float* scratch, long scratch_floats, hipStream_t st);
int tpos_grad(const float* g, float* dtpos, int B, int T, int J, int C, hipStream_t st);
int scale_rows(const float* g, const float* mask, int mask_mode, void* out, int out_bf16, int M, int C, int T, int J, hipStream_t st);
int adam_step(float* p, const float* g, float* m, float* v, long n, int step, float lr, float beta1, float beta2, float eps,
              float weight_decay, float grad_scale, hipStream_t st, const float* lr_mult = nullptr, const float* wd_mult = nullptr);
struct MaskDesc { int offset, count; float keep; };
int droppath_masks(float* masks, const MaskDesc* descs, int ndesc, unsigned long long seed, unsigned long long step,
                   hipStream_t st);

// ---------------------------------------------------------------- attention.hip
void attn_grad_f16_override(const float* gout);   // attention_mfma.hip: the MFMA backward launches issued next on this thread write dQ / dK / dV as fp16(*gout x value) (device address; null = bf16)
void attn_out_f16_override(int on);                // attention_mfma.hip: the temporal MFMA backward launches issued next on this thread read `out` as an fp16 plane (f16f8 = 3)
void attn_scale_override(float s);   // softmax scale of the attention launches issued next on this thread (0 = head_dim ** -0.5)
// qkv: [M][3C] (q | k | v, head-major inside each), out: [M][C]; token layout m = (b*T + t)*J + j
int attn_spatial_fwd(const void* qkv, void* out, int is_bf16, int B, int T, int J, int C, int H, hipStream_t st);
int attn_spatial_bwd(const void* qkv, const void* dout, void* dqkv, int is_bf16, int B, int T, int J, int C, int H, hipStream_t st);
int attn_temporal_fwd(const void* qkv, void* out, float* lse, int is_bf16, int B, int T, int J, int C, int H, hipStream_t st);
int attn_temporal_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int is_bf16,
                      int B, int T, int J, int C, int H, hipStream_t st);
// split-precision forward (planar hi/lo qkv and output, common.h).  scratch (4 M C floats) is only used for shapes the MFMA kernels
// do not cover (the planes are joined to fp32, the fp32 kernels run, the result is split again); may be null otherwise.
bool attn_x3_needs_scratch(int temporal, int T, int J, int C, int H);
// out_f16f8: the output leaves as "f16f8" planes (out_hi = fp16 plane, out_lo = 8-bit correction plane; head dim 64, MFMA kernels only)
int attn_spatial_fwd_x3(const bf16* qkv_hi, const bf16* qkv_lo, bf16* out_hi, bf16* out_lo, float* scratch, int B, int T, int J, int C, int H,
                        hipStream_t st, int out_f16f8 = 0);
int attn_temporal_fwd_x3(const bf16* qkv_hi, const bf16* qkv_lo, bf16* out_hi, bf16* out_lo, float* lse, float* scratch, int B, int T, int J,
                         int C, int H, hipStream_t st, int out_f16f8 = 0);
void attn_two_phase(int on);        // 0 = one-strip-at-a-time split-precision temporal forward for every shape (mp_set_option)
int join_planes(const bf16* hi, const bf16* lo, float* out, long n, hipStream_t st);
int split_planes(const float* in, bf16* hi, bf16* lo, long n, hipStream_t st);

// ---------------------------------------------------------------- heads.hip
// K heads of LayerNorm(C, eps 1e-5) + Linear(C, O): out[k][m][o]
struct HeadParams { const float* gamma[8]; const float* beta[8]; const float* W[8]; const float* b[8]; };
struct HeadGrads { float* gamma[8]; float* beta[8]; float* W[8]; float* b[8]; };
int heads_fwd(const float* x, const HeadParams& p, int K, int O, float* out, float* stats, int M, int C, hipStream_t st);
bool heads_mfma_supported(int K, int O, int C);
// heads on the fp32 matrix cores (heads_mfma.hip), same results as heads_fwd / heads_bwd.  fold: heads_fold_floats(C) floats owned by the
// module, written by the forward and read by the same step's backward, which also reads the forward's head outputs (`out`).
inline long heads_fold_floats(int C) { return 48L * C + 48; }
bool heads_use_mfma(int K, int O, int C);   // the engine's choice: a covered width (mp_set_option("heads_mfma"): 0 row kernels everywhere, 2 only from 16 outputs up)
void heads_mfma_mode(int mode);
int heads_fwd_mfma(const float* x, const HeadParams& p, int K, int O, float* out, float* stats, int M, int C, float* fold, hipStream_t st);
int heads_bwd_mfma(const float* x, const float* stats, const float* fold, const float* out, const HeadParams& p, const HeadGrads& gp, int K, int O,
                   const float* dout, float* dx, int M, int C, float* scratch, long scratch_floats, hipStream_t st, hipStream_t st_param);
int heads_bwd(const float* x, const float* stats, const HeadParams& p, const HeadGrads& gp, int K, int O, const float* dout,
              float* dx, int M, int C, float* scratch, long scratch_floats, hipStream_t st, hipStream_t st_param = nullptr);
// score head: logit[b,k,t] = sum_j ws_k[j] * headout[k][(b,t,j)][O-1] + bs_k ; scores = softmax_k
struct ScoreParams { const float* w[8]; const float* b[8]; };
struct ScoreGrads { float* w[8]; float* b[8]; };
int scores_fwd(const float* headout, const ScoreParams& p, int K, int O, float* scores, int B, int T, int J, hipStream_t st);
long scores_bwd_scratch_floats(int K, int B, int T);   // dlogit [K][B T] + the parameter-gradient partials
int scores_bwd(const float* headout, const float* scores, const float* dscores, const ScoreParams& p, const ScoreGrads& gp,
               int K, int O, float* dheadout, int B, int T, int J, float* scratch, long scratch_floats, hipStream_t st, hipStream_t st_param = nullptr, hipEvent_t ev = nullptr);
int bones_mean_fwd(const float* headout, float* lengths, int B, int T, int S, hipStream_t st);
int bones_mean_bwd(const float* dlen_pose, int KT, float* dlengths, float* dheadout, int B, int T, int S, hipStream_t st);

// ---------------------------------------------------------------- fk_decode.hip
// rot: element (k, m, c) at rot[(k*M + m)*rot_stride + c], m = (b*T+t)*J + j, c < 6; lengths [B][16]; poses [B][K][T][17][3]
int fk_decode_fwd(const float* rot, int rot_stride, int rot_dim, const float* lengths, float* poses, int B, int K, int T, hipStream_t st);
int fk_decode_bwd(const float* rot, int rot_stride, int rot_dim, const float* lengths, const float* dposes, float* drot,
                  float* dlen_pose, int B, int K, int T, hipStream_t st);

// ---------------------------------------------------------------- wta_loss.hip
struct LossCfg { float beta, vel_w, smooth_w; int use_joint_weights /* 0 none, 1 STANDARD_H36M_WEIGHTS, 2 joint_weights[] */, squared; float joint_weights[17]; };
// terms[4] = (wloss, score_reg, vloss, sreg); total = sum. dposes/dscores (may be null) receive d total / d input.
int wta_loss(const float* poses, const float* scores, const float* y, const LossCfg& cfg, float* terms, int* argmin,
             float* dposes, float* dscores, int B, int K, int T, float* scratch, long scratch_floats, hipStream_t st);
// single-hypothesis loss (ManifoldMixSTE): terms[3] = (wloss, vloss, sreg)
int single_loss(const float* poses, const float* y, const LossCfg& cfg, float* terms, float* dposes, int B, int T,
                float* scratch, long scratch_floats, hipStream_t st);
// rigid_seg_reg (main_h36m_lifting.py:170-177) of (B,T,17,3) poses: term[0] = weight * sum Var_t(bone length); dposes += gradient (may be null)
int rigid_segments_loss(const float* poses, float weight, float* term, float* dposes, int B, int T, float* scratch, long scratch_floats,
                        hipStream_t st);
// eval: aggregate + MPJPE sums. mode 0 weighted_ave, 1 best_score, 2 oracle
int aggregate_poses(const float* poses, const float* scores, const float* y, int mode, float* out, int B, int K, int T,
                    hipStream_t st);
int mpjpe_sum(const float* pred, const float* gt, long njoints, float* out_sum, float* scratch, long scratch_floats,
              hipStream_t st);

// ---------------------------------------------------------------- pose_metrics.hip
int pose_metrics_row_floats();
// segments_len_err(mode="no_agg") (mean_joint_errors.py:83-130): out[(b*L + t)*16 + k] = gt bone length - predicted bone length (|.| if !signed_)
int bone_length_table(const float* pred, const long* ps, const float* gt, const long* gs, int B, int L, int signed_, float* out, hipStream_t st);
int pose_metrics(const float* pred, const long* ps, const float* gt, const long* gs, const unsigned char* mask, int B, int L, int J,
                 float pred_scale, float gt_scale, float pck_thr, float auc_max, int auc_n, int scale_align, float* out, float* len0,
                 float* scratch, long scratch_floats, hipStream_t st);

// ---------------------------------------------------------------- windows.hip
int gather_windows(const float* p2, const float* p3, const long* seq_offset, int S, const int* win_seq, const int* win_start,
                   const unsigned char* win_flip, const int* mirror, const float* mask2d, const float* noise2d, int B, int T, int J,
                   float* X, float* y, hipStream_t st);

// ---------------------------------------------------------------- ingest.hip
int ingest_pose3d(const float* raw, int Jraw, const int* frames, long N, const int* joint_map, int J, const float* quat,
                  const float* trans, int root_raw, int root_out, float div, float* out, hipStream_t st);
int ingest_pose2d(const float* raw, int Jraw, int Craw, const int* frames, long N, const int* joint_map, int J, float w, float h,
                  float* out, hipStream_t st);

// ---------------------------------------------------------------- procrustes.hip
int procrustes_errors(const float* pred, const float* gt, const unsigned char* mask, long N, int J, float pred_scale, float gt_scale,
                      float pck_thr, float auc_max, int auc_n, int scaled, float* out, float* scratch, long scratch_floats, hipStream_t st);

}  // namespace mp
