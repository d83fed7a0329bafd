// Model engine: owns the activation workspace and issues the whole forward / backward launch sequence of
// RMCLManifoldMixSTE (rmcl_manifold_mix_ste.py:83-106) / ManifoldMixSTE (manifold_mix_ste.py:75-88) natively,
// one kernel stream, no host synchronisation, no tensor transposes.  The Python nn.Module mirror
// (manipose_amd/architectures) only hands over device pointers.
#include <stdlib.h>
#include <string.h>
#include <memory>
#include <string>
#include <vector>
#include "common.h"
#include "kernels.h"
#include "hazard.h"
#include "../../include/manipose_hip.h"

namespace mp {

bool attn_tmfma_supported(int T, int D);            // attention_mfma.hip
bool attn_smfma_supported(int N, int D, int H);
// Gradient operands of a layer whose backward GEMMs run on fp16 operands (mp_model_config::f16_backward) are carried as fp16 of S x value,
// S a power of two chosen per backward on the device (grad_scale, elementwise.hip: the largest element of the residual gradient at the top
// of the backbone lands in [2^11, 2^12), which leaves fp16 a factor 2^4 of headroom above it and 2^25 of normal range below - measured, see
// there and mp_model_backward); producers and consumers read S and 1 / S from
// mp_model::gsc, every store saturates and is counted there (common.h sat_f16x4, mp_model_grad_health).

const char* last_error();
long wgrad_f32_slab_floats(int Mtok, int Nout, int Kin);

// the attention backward launches issued inside the guard's scope write dQ / dK / dV as scaled fp16 (null: bf16); reset on every exit path
struct AttnGradF16Scope {
  explicit AttnGradF16Scope(const float* gsc) { attn_grad_f16_override(gsc); }
  ~AttnGradF16Scope() { attn_grad_f16_override(nullptr); }
  AttnGradF16Scope(const AttnGradF16Scope&) = delete;
  AttnGradF16Scope& operator=(const AttnGradF16Scope&) = delete;
};

struct ParamDesc { std::string name; long offset, numel; };
struct BlockP { int n1w, n1b, qkvw, qkvb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b; };
struct BlockWS {
  float *x_in, *st1, *lse, *x_mid, *st2, *x_out, *stp;
  void *a1, *qkv, *ao, *a2, *z, *f;        // void*: fp32 or bf16 by precision
  void *a1l, *qkvl, *aol, *a2l, *fl;       // precision 2 (split, common.h): the lo planes; a1 ... f are then the hi planes (= the bf16 tensors the backward reads)
  void *a1h, *a2h;                         // Module::f8: fp16 planes of a1 / a2 ("f16f8", common.h); a1l / a2l then hold their 8-bit correction planes, a1 / a2 the bf16 copies
};
struct MaskBranch { std::string name; float keep; int spatial; };

struct Module {
  std::string prefix;
  bool is_rot;
  int N, C, H, depth;     // tokens per frame (17 joints / 16 bones), width, heads, depth
  int K, O;               // heads, out features per head
  bool f8 = false;        // precision 2: the qkv and fc1 Linear layers read "f16f8" operands (one fp16 + one fp8 matrix-core step per k-tile instead of three bf16 ones)
  bool f8g = false;       // f8 and the backward of the qkv and fc1 layers on fp16 operands (dqkv / dz written as scaled fp16 by the attention backward kernels / the
                          // fc2 dgrad, the fp16 planes of a1 / a2 as the weight-gradient operands, the f16f8 shadow's fp16 plane as the dgrads' weights): a1, a2 need no bf16 copy
  bool f8m = false;       // f8g and the fc2 layer too: f (the GELU output) is written as f16f8 planes by the fc1 epilogue and read so by the fc2 forward GEMM; its
                          // backward runs on fp16 operands (the gradient copy of the LayerNorm backward behind the block as scaled fp16)
  bool f8a = false;       // f8 (without f8g) on ALL FOUR Linear layers (mp_model_config::f16f8 = 3, round 6): the attention kernels and the fc1 epilogue write f16f8
                          // planes too (ao, f), NO bf16 copy of a1 / ao / a2 / f exists - the bf16 backward's weight-gradient GEMMs round the fp16 planes to
                          // bf16 per fragment (wgrad_bf16 x_f16) and the temporal attention backward reads O from its fp16 plane
  float qk_scale, rs, readout;   // attention softmax scale (0 = head_dim^-0.5), residual scale, MuReadout input multiplier (1 unless muP)
  float *hw_eff, *hdw;    // readout != 1: the heads' weights times readout (forward / dx) and the scratch their gradient lands in
  int emb_w, emb_b, spos, tpos, sn_w, sn_b, tn_w, tn_b;
  std::vector<BlockP> bp; // order: STE0, TTE0, STE1, TTE1, ...
  int hg[8], hb[8], hw[8], hbias[8], sw[8], sb[8];
  std::vector<BlockWS> ws;
  float *x_final, *hstats, *hfold, *headout, *dheadout;
  std::vector<MaskBranch> masks;   // 2 per block: attn, mlp
  int mask_base;                   // index of this module's first branch in the global list
};

}  // namespace mp

using namespace mp;

struct mp_model {
  mp_model_config cfg;
  std::vector<ParamDesc> params;
  long flat_size = 0;
  Module rot, seg;
  // workspace
  char* arena = nullptr;
  size_t arena_bytes = 0;
  float *g = nullptr, *tmpC = nullptr, *tmpMask = nullptr, *delta = nullptr;
  bf16* g_b16 = nullptr;                     // bf16 copy of the gradient stream (A operand of dgrad/wgrad in precision 1)
  void *tmp2C = nullptr, *tmp3C = nullptr;   // dz / dqkv: fp32 or bf16 by precision
  bf16* wbf = nullptr;                       // bf16 shadow of the flat parameter buffer (precision 1; precision 2: its hi plane)
  bf16* wbf_lo = nullptr;                    // precision 2: lo plane of the shadow
  void* w16 = nullptr;                       // a module with f8: "f16f8" shadow of the flat parameter buffer (fp16 plane, 8-bit correction plane in the weight form)
  char* w8 = nullptr;
  float* gsc = nullptr;                      // {S, 1 / S, scratch, 1}: the gradient scale of the current backward (modules with f8g)
  long xattn_half = 0;
  float* xattn = nullptr;                    // precision 2, attention shapes without an MFMA kernel: fp32 scratch (4 M C floats) of the join -> fp32 kernel -> split route
  float *slab = nullptr, *small = nullptr, *lengths = nullptr, *dlen_pose = nullptr, *maskbuf = nullptr, *dscore_zero = nullptr;
  long slab_floats = 0, small_floats = 0;
  // The bones net runs concurrently with the rotations net on a second stream with its own scratch set; the host code stays
  // sequential and simply swaps which set the fields above point at while it enqueues that module's kernels.
  struct ScratchSet { float *g, *tmpC, *tmpMask, *delta, *slab, *small; bf16* g_b16; void *tmp2C, *tmp3C; long slab_floats, small_floats; };
  ScratchSet sets[2];
  hipStream_t st2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_heads = nullptr, ev_sync = nullptr;
  std::vector<hipEvent_t> ev_bucket;        // one per MixSTE layer of the rotations net: its (STE_i, TTE_i) parameter gradients of the last backward are final
  bool buckets_recorded = false;
  float *hsmall = nullptr, *sc_dlogit = nullptr;   // scratch of the head / score parameter-gradient kernels when they run on the wgrad stream
  long hsmall_floats = 0;
  float* lnpart = nullptr;                   // one partial-sum slice per LayerNorm backward of the rotations net (reduced on the wgrad stream)
  long lnpart_slice = 0;
  int lnpart_n = 0;
  // weight-gradient GEMMs of the rotations net do not feed the backward chain: they run on a third stream, ordered against the
  // chain's scratch-buffer reuse by events (E = "operand ready", W = "wgrad done reading")
  hipStream_t st3 = nullptr;
  hipEvent_t evE[2][4] = {}, evW[2][4] = {};
  bool wgrad_async = false;
  bool has_seg = true;                       // false for arch 2 (bare MixSTE): no segments module, no decoder
  float cur_rs = 1.0f;                       // residual scale of the module whose kernels are being enqueued (Module::rs)
  // state of the last forward
  int B = 0;
  bool train = false;
  bool infer = false;                        // last forward: the caller announced that no backward follows (mp_model_forward, train bit 1)
  const float* x_in = nullptr;
  // profiling
  bool prof = false;
  std::vector<hipEvent_t> ev;
  std::vector<int> ev_cls;
  std::vector<double> ev_flops, ev_bytes, ev_mflops;   // issued matrix-core flops, algorithmic bytes, flops of the mathematical product (2 M N K)
  std::vector<char> ev_tag;              // 1: the launch ran gemm_bf16_persist_kernel
  std::vector<signed char> ev_kind;      // Linear GEMM launches: module * 12 + direction * 4 + layer (mp_prof_kinds), -1 otherwise
  std::unique_ptr<mp::HazardTracker> hz; // mp_model_config::debug bit 0: the stream-hazard check (hazard.h); null otherwise
  int cur_kind = -1;                     // kind of the GEMM being enqueued (set by the backbone code around linear_fwd / dgrad / wgrad)
  double kind_ms[MP_PROF_KINDS] = {}, kind_flops[MP_PROF_KINDS] = {}, kind_bytes[MP_PROF_KINDS] = {}, kind_mflops[MP_PROF_KINDS] = {};
  int64_t kind_launches[MP_PROF_KINDS] = {}, kind_persist[MP_PROF_KINDS] = {};
  size_t ev_used = 0;
};

namespace mp {

static int add_param(mp_model* m, const std::string& name, long numel) {
  ParamDesc d{name, m->flat_size, numel};
  m->params.push_back(d);
  m->flat_size += (numel + 63) / 64 * 64;   // 256-byte aligned slots
  return (int)m->params.size() - 1;
}

static void build_module_params(mp_model* m, Module& md) {
  const std::string& p = md.prefix;
  const int C = md.C;
  if (md.is_rot) {
    md.emb_w = add_param(m, p + "Spatial_patch_to_embedding.weight", (long)C * 2);
    md.emb_b = add_param(m, p + "Spatial_patch_to_embedding.bias", C);
  } else {
    md.emb_w = add_param(m, p + "joints_to_segments_proj.weight", (long)md.N * C * m->cfg.num_joints * 2);
    md.emb_b = add_param(m, p + "joints_to_segments_proj.bias", (long)md.N * C);
  }
  md.spos = add_param(m, p + "Spatial_pos_embed", (long)md.N * C);
  md.tpos = add_param(m, p + "Temporal_pos_embed", (long)m->cfg.num_frame * C);
  md.bp.resize(2 * md.depth);
  for (int i = 0; i < md.depth; ++i) {
    for (int kind = 0; kind < 2; ++kind) {
      const std::string b = p + (kind == 0 ? "STEblocks." : "TTEblocks.") + std::to_string(i) + ".";
      BlockP& q = md.bp[2 * i + kind];
      q.n1w = add_param(m, b + "norm1.weight", C);
      q.n1b = add_param(m, b + "norm1.bias", C);
      q.qkvw = add_param(m, b + "attn.qkv.weight", 3L * C * C);
      q.qkvb = add_param(m, b + "attn.qkv.bias", 3L * C);
      q.pw = add_param(m, b + "attn.proj.weight", (long)C * C);
      q.pb = add_param(m, b + "attn.proj.bias", C);
      q.n2w = add_param(m, b + "norm2.weight", C);
      q.n2b = add_param(m, b + "norm2.bias", C);
      q.f1w = add_param(m, b + "mlp.fc1.weight", 2L * C * C);
      q.f1b = add_param(m, b + "mlp.fc1.bias", 2L * C);
      q.f2w = add_param(m, b + "mlp.fc2.weight", 2L * C * C);
      q.f2b = add_param(m, b + "mlp.fc2.bias", C);
      const float rate = (md.depth > 1) ? m->cfg.drop_path_rate * (float)i / (float)(md.depth - 1) : 0.f;   // linspace(0, rate, depth)
      md.masks.push_back({b + "attn", 1.0f - rate, kind == 0});
      md.masks.push_back({b + "mlp", 1.0f - rate, kind == 0});
    }
  }
  md.sn_w = add_param(m, p + "Spatial_norm.weight", C);
  md.sn_b = add_param(m, p + "Spatial_norm.bias", C);
  md.tn_w = add_param(m, p + "Temporal_norm.weight", C);
  md.tn_b = add_param(m, p + "Temporal_norm.bias", C);
  if (md.is_rot && m->cfg.arch == 0) {
    for (int k = 0; k < md.K; ++k) {
      const std::string h = p + "head." + std::to_string(k) + ".";
      md.hg[k] = add_param(m, h + "norm.weight", C);
      md.hb[k] = add_param(m, h + "norm.bias", C);
      md.hw[k] = add_param(m, h + "prediction_head.weight", (long)md.O * C);
      md.hbias[k] = add_param(m, h + "prediction_head.bias", md.O);
      md.sw[k] = add_param(m, h + "score_head.weight", md.N);
      md.sb[k] = add_param(m, h + "score_head.bias", 1);
    }
  } else {
    md.hg[0] = add_param(m, p + "head.0.weight", C);
    md.hb[0] = add_param(m, p + "head.0.bias", C);
    md.hw[0] = add_param(m, p + "head.1.weight", (long)md.O * C);
    md.hbias[0] = add_param(m, p + "head.1.bias", md.O);
  }
}

struct Bump {
  size_t off = 0;
  char* base = nullptr;
  float* take(long floats) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += ((size_t)floats * 4 + 255) / 256 * 256;
    return p;
  }
};

static void carve_module(Module& md, Bump& bp, long M, int T, int Bmax, int precision) {
  const long C = md.C;
  const int half = precision >= 1;
  auto act = [&](long n) -> void* { return bp.take(half ? (n + 1) / 2 : n); };   // bf16 activations take half the floats
  auto lo = [&](long n) -> void* { return precision == 2 ? bp.take((n + 1) / 2) : nullptr; };   // lo plane of a planar activation
  md.ws.resize(2 * md.depth);
  // The lo planes (and, with f8, the 8-bit correction planes that take the place of a1l / a2l) are read by the forward consumer of their own block
  // only - the backward runs on the hi planes - and a module's kernels are enqueued on one stream in block order: ONE set per module, shared
  // by all its blocks (16 C bytes per token instead of 16 C x 2 depth: 124 -> 84 GiB at the bench's B = 79).
  void* const a1l = lo(M * C); void* const qkvl = lo(M * 3 * C); void* const aol = lo(M * C); void* const a2l = lo(M * C); void* const fl = lo(M * 2 * C);
  for (size_t l = 0; l < md.ws.size(); ++l) {
    BlockWS& w = md.ws[l];
    const bool lazy_in = half && l >= 2 && C <= 512;         // lazy_block_input(): recomputed where it is used, never stored
    w.x_in = lazy_in ? nullptr : bp.take(M * C);
    w.st1 = bp.take(M * 2);     w.a1 = act(M * C);   w.qkv = act(M * 3 * C);
    w.lse = bp.take((long)Bmax * md.N * md.H * T);
    w.ao = act(M * C);         w.x_mid = bp.take(M * C);   w.st2 = bp.take(M * 2);  w.a2 = act(M * C);
    w.z = act(M * 2 * C);      w.f = act(M * 2 * C);       w.x_out = bp.take(M * C); w.stp = bp.take(M * 2);
    w.a1l = a1l; w.qkvl = qkvl; w.aol = aol; w.a2l = a2l; w.fl = fl;
    // f8g: nobody reads a bf16 a1 / a2 (the backward runs on the fp16 planes): the fp16 planes take their place
    w.a1h = md.f8 ? ((md.f8g || md.f8a) ? w.a1 : bp.take((M * C + 1) / 2)) : nullptr;      // (f8a: no bf16 copy either)
    w.a2h = md.f8 ? ((md.f8g || md.f8a) ? w.a2 : bp.take((M * C + 1) / 2)) : nullptr;
  }
  md.x_final = bp.take(M * C);
  md.hw_eff = md.readout != 1.0f ? bp.take((long)md.K * md.O * C) : nullptr;
  md.hdw = md.readout != 1.0f ? bp.take((long)md.K * md.O * C) : nullptr;
  md.hstats = bp.take(M * 2);
  md.hfold = bp.take(heads_fold_floats(C));
  md.headout = bp.take((long)md.K * M * md.O);
  md.dheadout = bp.take((long)md.K * M * md.O);
}

static long small_scratch_floats(const mp_model* m) {
  long s = 1024L * 4 * 1024;                                                       // ln_bwd / ln_bwd2 partials (LNB_GRID rows)
  const Module* mods[2] = {&m->rot, &m->seg};
  for (const Module* md : mods) s = max(s, 512L * md->K * ((long)md->O * md->C + md->O + 2 * md->C));   // heads_bwd
  s = max(s, embed_bwd_scratch_floats(m->rot.C, m->rot.N));                       // embed_bwd
  s = max(s, 33L * m->seg.N * m->seg.C * 35);                                     // bones_embed_bwd
  s = max(s, scores_bwd_scratch_floats(8, m->cfg.max_batch, m->cfg.num_frame) + 64);  // scores_bwd dlogit + partials / loss partials
  return s + 1024;
}

static void carve_all(mp_model* m, Bump& bp) {
  const int Bm = m->cfg.max_batch, T = m->cfg.num_frame;
  const long Mr = (long)Bm * T * m->rot.N, Ms = (long)Bm * T * m->seg.N;
  const int half = m->cfg.precision >= 1;
  carve_module(m->rot, bp, Mr, T, Bm, m->cfg.precision);
  if (m->has_seg) carve_module(m->seg, bp, Ms, T, Bm, m->cfg.precision);
  if (half) m->wbf = reinterpret_cast<bf16*>(bp.take((m->flat_size + 1) / 2));
  if (m->cfg.precision == 2) {
    m->wbf_lo = reinterpret_cast<bf16*>(bp.take((m->flat_size + 1) / 2));
    if (m->rot.f8 || m->seg.f8) {
      m->w16 = bp.take((m->flat_size + 1) / 2);
      m->w8 = reinterpret_cast<char*>(bp.take((m->flat_size + 1) / 2));
      m->gsc = bp.take(64);
    }
    long need = 0;
    if (attn_x3_needs_scratch(0, T, m->rot.N, m->rot.C, m->rot.H) || attn_x3_needs_scratch(1, T, m->rot.N, m->rot.C, m->rot.H)) need = max(need, 4 * Mr * m->rot.C);
    if (m->has_seg && (attn_x3_needs_scratch(0, T, m->seg.N, m->seg.C, m->seg.H) || attn_x3_needs_scratch(1, T, m->seg.N, m->seg.C, m->seg.H)))
      need = max(need, 4 * Ms * m->seg.C);
    // the two nets run concurrently on two streams: one scratch each
    m->xattn = need ? bp.take(2 * need) : nullptr;
    m->xattn_half = need;
  }
  const int halfp = m->cfg.precision >= 1;
  const Module* mods[2] = {&m->rot, &m->seg};
  for (int i = 0; i < (m->has_seg ? 2 : 1); ++i) {
    const Module* md = mods[i];
    const long M = (i == 0) ? Mr : Ms, MC = M * md->C;
    mp_model::ScratchSet& sc = m->sets[i];
    sc.g = bp.take(MC);
    sc.tmpC = bp.take(MC);
    sc.tmpMask = bp.take(MC);
    sc.g_b16 = halfp ? reinterpret_cast<bf16*>(bp.take((MC + 1) / 2)) : nullptr;
    sc.tmp2C = bp.take(halfp ? MC : 2 * MC);
    sc.tmp3C = bp.take(halfp ? (3 * MC + 1) / 2 : 3 * MC);
    sc.delta = bp.take((long)Bm * md->N * md->H * T);
    const int C = md->C;
    long slab = wgrad_f32_slab_floats((int)M, 3 * C, C);   // same tile-derived bound for the bf16 kernels
    slab = max(slab, wgrad_f32_slab_floats((int)M, C, C));
    slab = max(slab, wgrad_f32_slab_floats((int)M, 2 * C, C));
    slab = max(slab, wgrad_f32_slab_floats((int)M, C, 2 * C));
    sc.slab_floats = slab;
    sc.slab = bp.take(slab);
    sc.small_floats = small_scratch_floats(m);
    sc.small = bp.take(sc.small_floats);
  }
  m->lengths = bp.take((long)Bm * m->seg.N);
  m->dlen_pose = bp.take((long)Bm * m->rot.K * T * m->seg.N);
  long nm = 0;
  for (const Module* md : mods)
    for (const auto& b : md->masks) nm += b.spatial ? (long)Bm * T : (long)Bm * md->N;
  m->maskbuf = bp.take(nm + 64);
  m->dscore_zero = bp.take((long)Bm * m->rot.K * T);
  m->hsmall_floats = 512L * m->rot.K * ((long)m->rot.O * m->rot.C + m->rot.O + 2 * m->rot.C) + 1024;
  m->hsmall = bp.take(m->hsmall_floats);
  m->sc_dlogit = bp.take(scores_bwd_scratch_floats(m->rot.K, Bm, T) + 64);
  m->lnpart_slice = 1024L * 4 * m->rot.C;                  // LNB_GRID rows x up to 4 C partial sums (ln_bwd2)
  m->lnpart_n = 3 * 2 * m->rot.depth + 2;                  // at most three LayerNorm backwards per block
  m->lnpart = bp.take(m->lnpart_slice * m->lnpart_n);
}

static void use_scratch(mp_model* m, int i) {
  const mp_model::ScratchSet& sc = m->sets[i];
  m->g = sc.g; m->tmpC = sc.tmpC; m->tmpMask = sc.tmpMask; m->delta = sc.delta; m->slab = sc.slab; m->small = sc.small;
  m->g_b16 = sc.g_b16; m->tmp2C = sc.tmp2C; m->tmp3C = sc.tmp3C; m->slab_floats = sc.slab_floats; m->small_floats = sc.small_floats;
}

// ---- profiling helpers -------------------------------------------------------------------------
enum { PC_GEMM_FWD = 0, PC_GEMM_DGRAD = 1, PC_GEMM_WGRAD = 2, PC_ATTN = 3, PC_LN = 4, PC_OTHER = 5 };
enum { LK_QKV = 0, LK_PROJ = 1, LK_FC1 = 2, LK_FC2 = 3 };
struct ProfScope {
  mp_model* m;
  hipStream_t st;
  bool on;
  ProfScope(mp_model* mm, hipStream_t s, int cls, double flops, double bytes = 0.0, double mflops = -1.0) : m(mm), st(s), on(false) {
    if (m->prof && m->ev_used + 2 <= m->ev.size()) {
      on = true;
      m->ev_cls.push_back(cls);
      m->ev_flops.push_back(flops);
      m->ev_bytes.push_back(bytes);
      m->ev_mflops.push_back(mflops >= 0.0 ? mflops : flops);
      m->ev_kind.push_back((signed char)((cls <= PC_GEMM_WGRAD) ? m->cur_kind : -1));
      (void)gemm_bf16_take_last_persist();
      (void)hipEventRecord(m->ev[m->ev_used], st);
    }
  }
  ~ProfScope() {
    if (on) {
      (void)hipEventRecord(m->ev[m->ev_used + 1], st);
      m->ev_used += 2;
      m->ev_tag.push_back((char)gemm_bf16_take_last_persist());
    }
  }
};
#define RUN(cls, flops, call)                  \
  do {                                         \
    ProfScope ps__(m, st, (cls), (flops));     \
    int rc__ = (call);                         \
    if (rc__) return rc__;                     \
  } while (0)
// the same with the launch's ALGORITHMIC bytes (operands read once + outputs written once): GEMM classes only
#define RUNB(cls, flops, bytes, call)                  \
  do {                                                 \
    ProfScope ps__(m, st, (cls), (flops), (bytes));    \
    int rc__ = (call);                                 \
    if (rc__) return rc__;                             \
  } while (0)

// ---- stream-hazard check (hazard.h): every launch declares its stream and the byte ranges it reads (HR) / writes (HW); the engine's event
// records and waits go through ev_record / ev_wait so that the tracker sees the same ordering edges HIP does ----
#define HR(p, bytes) mp::hz_r((p), (double)(bytes))
#define HW(p, bytes) mp::hz_w((p), (double)(bytes))
#define HZ(st_, name_, ...)                                                                                                  \
  do {                                                                                                                       \
    if (m->hz) {                                                                                                             \
      const mp::HzAccess hz_a__[] = {__VA_ARGS__};                                                                           \
      m->hz->launch(m->hz->stream_id((const void*)(st_)), (name_), hz_a__, (int)(sizeof(hz_a__) / sizeof(hz_a__[0])));       \
    }                                                                                                                        \
  } while (0)
static hipError_t ev_record(mp_model* m, hipEvent_t ev, hipStream_t st) {
  if (m->hz) m->hz->record((const void*)ev, m->hz->stream_id((const void*)st));
  return hipEventRecord(ev, st);
}
static hipError_t ev_wait(mp_model* m, hipStream_t st, hipEvent_t ev) {
  if (m->hz) m->hz->wait(m->hz->stream_id((const void*)st), (const void*)ev);
  return hipStreamWaitEvent(st, ev, 0);
}
// a kernel wrapper that moves its parameter-gradient tail to (st_param, ev) records ev on st and makes st_param wait for it (param_stream(),
// elementwise.hip; scores_bwd, heads.hip): the same edge for the tracker; returns the stream the tail runs on
static hipStream_t hz_param_edge(mp_model* m, hipStream_t st, hipStream_t st_param, hipEvent_t ev) {
  if (st_param == nullptr || ev == nullptr) return st;
  if (m->hz) { m->hz->record((const void*)ev, m->hz->stream_id((const void*)st)); m->hz->wait(m->hz->stream_id((const void*)st_param), (const void*)ev); }
  return st_param;
}

static const float* P(const mp_model* m, const float* flat, int idx) { return flat + m->params[idx].offset; }
static float* G(const mp_model* m, float* flat, int idx) { return flat + m->params[idx].offset; }

// Linear layers: precision 0 -> fp32 matrix cores on fp32 buffers; precision 1 -> bf16 matrix cores, bf16 activations /
// shadow weights, fp32 residual stream and fp32 gradient stream (converted to bf16 while staging).
// rstats / rgamma / rbeta (bf16 mode, residual epilogue): the residual is LayerNorm(R) recomputed in the epilogue (kernels.h)
static int linear_fwd(mp_model* m, hipStream_t st, const float* fp, const void* A, int widx, int bidx, void* Cc, long M, int N, int K,
                      int epi, void* Z, const float* R, const float* mask, int mask_mode, int T, int J, const float* rstats = nullptr,
                      const float* rgamma = nullptr, const float* rbeta = nullptr, const void* A_lo = nullptr, void* C_lo = nullptr,
                      bool f8in = false, bool f8out = false) {
  {      // stream-hazard check: operands of this launch (element sizes by precision; the weight shadow, not the fp32 parameters, is the B operand from precision 1 on)
    const int p = m->cfg.precision;
    const double ea = p == 0 ? 4.0 : 2.0, ec = (p == 0 || epi == EPI_BIAS_RESID) ? 4.0 : 2.0;
    const long woff = m->params[widx].offset;
    HZ(st, "linear_fwd", HR(A, ea * M * K), HR(A_lo, ea * M * K), HR(p == 0 ? (const void*)P(m, fp, widx) : (const void*)(m->wbf + woff), ea * N * K),
       HR(p == 2 ? (const void*)(m->wbf_lo + woff) : nullptr, 2.0 * N * K), HR(f8in ? (const char*)m->w16 + woff * 2 : nullptr, 2.0 * N * K),
       HR(f8in ? m->w8 + woff * 2 : nullptr, 2.0 * N * K), HW(Cc, ec * M * N), HW(C_lo, ec * M * N), HW(Z, ec * M * N), HR(R, 4.0 * M * N), HR(rstats, 8.0 * M),
       HR(mask, 4.0 * (mask_mode == 1 ? M / J : (M / ((long)T * J)) * J)));
  }
  if (m->cfg.precision == 0) {
    GemmF32Args g = {};
    g.A = (const float*)A; g.lda = K; g.B = P(m, fp, widx); g.ldb = K; g.C = (float*)Cc; g.ldc = N; g.M = (int)M; g.N = N; g.K = K;
    g.bias = P(m, fp, bidx); g.Z = (float*)Z; g.R = R; g.mask = mask; g.mask_mode = mask ? mask_mode : 0; g.T = T; g.J = J; g.rscale = m->cur_rs;
    RUNB(PC_GEMM_FWD, 2.0 * M * N * K, 4.0 * (M * K + (double)N * K + M * N * (1 + (epi == EPI_BIAS_RESID) + (epi == EPI_BIAS_GELU))),
         gemm_f32(0, 0, epi, g, st));
    return MP_OK;
  }
  GemmB16Args g = {};
  g.A = A; g.lda = K; g.B = m->wbf + m->params[widx].offset; g.ldb = K; g.C = Cc; g.ldc = N; g.M = (int)M; g.N = N; g.K = K;
  g.bias = P(m, fp, bidx); g.Z = Z; g.R = R; g.mask = mask; g.mask_mode = mask ? mask_mode : 0; g.T = T; g.J = J;
  g.rstats = rstats; g.rgamma = rgamma; g.rbeta = rbeta; g.rscale = m->cur_rs;
  if (m->cfg.precision == 2 && f8in) {
    // "f16f8" operands: A = the fp16 plane, A_lo = its correction plane, weights from the f16f8 shadow; planar bf16 outputs as below
    g.A_lo = A_lo; g.B = reinterpret_cast<const char*>(m->w16) + m->params[widx].offset * 2; g.B_lo = m->w8 + m->params[widx].offset * 2; g.C_lo = C_lo;
    g.out_f16f8 = f8out ? 1 : 0;
    const double ob = epi == EPI_BIAS_RESID ? 8.0 * M * N : 4.0 * M * N + (epi == EPI_BIAS_GELU ? 2.0 * M * N : 0.0);
    {
      ProfScope ps__(m, st, PC_GEMM_FWD, 4.0 * M * N * K, 4.0 * (M * K + (double)N * K) + ob, 2.0 * M * N * K);   // two matrix-core steps per k-tile (one fp16, one fp8 at twice the depth)
      int rc__ = gemm_f16f8(g, epi == EPI_BIAS_RESID ? 1 : 0, epi, st);
      if (rc__) return rc__;
    }
    return MP_OK;
  }
  if (m->cfg.precision == 2) {
    // split precision: planar hi/lo A and weights (both planes read), planar outputs (+ the plain-bf16 gelu'), fp32 residual in + out
    g.A_lo = A_lo; g.B_lo = m->wbf_lo + m->params[widx].offset; g.C_lo = C_lo;
    const double ob = epi == EPI_BIAS_RESID ? 8.0 * M * N : 4.0 * M * N + (epi == EPI_BIAS_GELU ? 2.0 * M * N : 0.0);
    {
      ProfScope ps__(m, st, PC_GEMM_FWD, 6.0 * M * N * K, 4.0 * (M * K + (double)N * K) + ob, 2.0 * M * N * K);   // three products issued per product of the model
      int rc__ = gemm_bf16x3(g, epi == EPI_BIAS_RESID ? 1 : 0, epi, st);
      if (rc__) return rc__;
    }
    return MP_OK;
  }
  // bf16 A and weights; output bf16 (+ a second bf16 output gelu') or, for the residual epilogue, fp32 in + fp32 out
  const double obytes = epi == EPI_BIAS_RESID ? 8.0 * M * N : 2.0 * M * N * (epi == EPI_BIAS_GELU ? 2 : 1);
  RUNB(PC_GEMM_FWD, 2.0 * M * N * K, 2.0 * (M * K + (double)N * K) + obytes, gemm_bf16(g, 0, 0, 0, epi == EPI_BIAS_RESID ? 1 : 0, epi, st));
  return MP_OK;
}
// dX[M,K] = dY[M,N] W[N,K]  (optionally * gelu'(Z)).  dy_f32 / dx_f32: storage of dY / dX in bf16 mode.
static int linear_dgrad(mp_model* m, hipStream_t st, const float* fp, const void* dY, int dy_f32, int widx, void* dX, int dx_f32,
                        long M, int N, int K, void* Z, bool f16 = false, const float* gout = nullptr) {
  {
    const int p = m->cfg.precision;
    const long woff = m->params[widx].offset;
    HZ(st, "linear_dgrad", HR(dY, (p == 0 || dy_f32 ? 4.0 : 2.0) * M * N), HR(p == 0 ? (const void*)P(m, fp, widx) : (f16 ? (const void*)((const char*)m->w16 + woff * 2) : (const void*)(m->wbf + woff)), (p == 0 ? 4.0 : 2.0) * N * K),
       HW(dX, (p == 0 || dx_f32 ? 4.0 : 2.0) * M * K), HR(Z, (p == 0 ? 4.0 : 2.0) * M * K), HR(gout, 4));
  }
  if (m->cfg.precision == 0) {
    GemmF32Args g = {};
    g.A = (const float*)dY; g.lda = N; g.B = P(m, fp, widx); g.ldb = K; g.C = (float*)dX; g.ldc = K; g.M = (int)M; g.N = K; g.K = N;
    g.Z = (float*)Z;
    RUNB(PC_GEMM_DGRAD, 2.0 * M * N * K, 4.0 * (M * N + (double)N * K + M * K * (Z ? 2 : 1)), gemm_f32(0, 1, Z ? EPI_DGELU : EPI_BIAS, g, st));
    return MP_OK;
  }
  GemmB16Args g = {};
  g.A = dY; g.lda = N; g.B = m->wbf + m->params[widx].offset; g.ldb = K; g.C = dX; g.ldc = K; g.M = (int)M; g.N = K; g.K = N; g.Z = Z;
  g.gout = gout;
  g.gsat = gout != nullptr ? reinterpret_cast<unsigned*>(m->gsc + 4) : nullptr;
  if (f16) {      // dY = fp16(S x gradient), weights from the fp16 plane of the f16f8 shadow
    g.f16 = 1;      // (the bf16 output keeps dY's scale: the LayerNorm backward that reads it gets dy_scale = 1 / S)
    g.B = reinterpret_cast<const char*>(m->w16) + m->params[widx].offset * 2;
  }
  RUNB(PC_GEMM_DGRAD, 2.0 * M * N * K, (dy_f32 ? 4.0 : 2.0) * M * N + 2.0 * N * K + (dx_f32 ? 4.0 : 2.0) * M * K + (Z ? 2.0 * M * K : 0.0),
       gemm_bf16(g, dy_f32, 0, 1, dx_f32, Z ? EPI_DGELU : EPI_BIAS, st));
  return MP_OK;
}
static int linear_wgrad(mp_model* m, hipStream_t st, const void* dY, int dy_f32, const void* X, float* dW, float* db, long M, int N,
                        int K, bool f16 = false, bool x_f16 = false) {
  HZ(st, "linear_wgrad", HR(dY, (m->cfg.precision == 0 || dy_f32 ? 4.0 : 2.0) * M * N), HR(X, (m->cfg.precision == 0 ? 4.0 : 2.0) * M * K), HW(dW, 4.0 * N * K), HW(db, 4.0 * N),
     HW(m->slab, 4.0 * m->slab_floats));
  if (m->cfg.precision == 0)
    RUN(PC_GEMM_WGRAD, 2.0 * M * N * K, wgrad_f32((const float*)dY, N, (const float*)X, K, (int)M, N, K, dW, db, m->slab, m->slab_floats, st));
  else
    RUN(PC_GEMM_WGRAD, 2.0 * M * N * K, wgrad_bf16(dY, dy_f32, N, (const bf16*)X, K, (int)M, N, K, dW, db, m->slab, m->slab_floats, st, f16 ? 1 : 0,
                                                   f16 ? m->gsc + 1 : nullptr, x_f16 ? 1 : 0));
  return MP_OK;
}

static const float* branch_mask(const mp_model* m, const Module& md, int l, int br, int B, bool train) {
  if (!train) return nullptr;
  const MaskBranch& mb = md.masks[2 * l + br];
  if (mb.keep >= 1.0f) return nullptr;
  // offset of this branch inside maskbuf for batch B
  long off = 0;
  const Module* mods[2] = {&m->rot, &m->seg};
  for (const Module* q : mods) {
    for (size_t i = 0; i < q->masks.size(); ++i) {
      if (q == &md && (int)i == 2 * l + br) return m->maskbuf + off;
      off += q->masks[i].spatial ? (long)B * m->cfg.num_frame : (long)B * q->N;
    }
  }
  return nullptr;
}

// Is the input of block l (the shared post-norm of block l-1's output) recomputed where it is used instead of stored?  bf16 mode,
// blocks >= 2 (no embedding / positional table behind them), C <= 512 (their backward then runs through ln_bwd2, which
// recomputes it as well).
static bool lazy_block_input(const mp_model* m, const Module& md, int l) {
  return m->cfg.precision >= 1 && l >= 2 && md.C <= 512;
}

static int backbone_fwd_impl(mp_model* m, Module& md, const float* fp, int B, hipStream_t st);
static int backbone_bwd_impl(mp_model* m, Module& md, const float* fp, float* fg, int B, hipStream_t st);
// the module's attention scale / residual scale apply to every launch of its backbone (attention launchers read the thread's override)
static int backbone_fwd(mp_model* m, Module& md, const float* fp, int B, hipStream_t st) {
  attn_scale_override(md.qk_scale);
  m->cur_rs = md.rs;
  const int rc = backbone_fwd_impl(m, md, fp, B, st);
  m->cur_kind = -1;
  attn_scale_override(0.f);
  m->cur_rs = 1.0f;
  return rc;
}
static int backbone_bwd(mp_model* m, Module& md, const float* fp, float* fg, int B, hipStream_t st) {
  attn_scale_override(md.qk_scale);
  m->cur_rs = md.rs;
  const int rc = backbone_bwd_impl(m, md, fp, fg, B, st);
  m->cur_kind = -1;
  attn_scale_override(0.f);
  m->cur_rs = 1.0f;
  return rc;
}

static void hz_ln_fwd(mp_model* m, hipStream_t st, const LnFwdArgs& a, int out_mode) {
  const double MC = (double)a.M * a.C, eo = out_mode == 0 ? 4.0 : 2.0;
  HZ(st, "ln_fwd", HR(a.x, 4.0 * MC), HW(a.x1, 4.0 * MC), HW(a.stats1, 8.0 * a.M), HW(a.y2, eo * MC), HW(a.y2_lo, 2.0 * MC), HW(a.y2_b16, 2.0 * MC),
     HW(a.stats2, 8.0 * a.M));
}
static int backbone_fwd_impl(mp_model* m, Module& md, const float* fp, int B, hipStream_t st) {
  const int T = m->cfg.num_frame, N = md.N, C = md.C, H = md.H, L = 2 * md.depth;
  const int half = md.f8 ? 3 : m->cfg.precision;          // LayerNorm output mode: 0 fp32, 1 bf16, 2 planar hi/lo bf16, 3 f16f8 planes + bf16 copy
  const bool x3 = m->cfg.precision == 2;
  float* const xattn = (x3 && m->xattn) ? m->xattn + (st == m->st2 ? m->xattn_half : 0) : nullptr;
  const long M = (long)B * T * N;
  // norm1 of block 0 (the input embedding has already been written to ws[0].x_in)
  {
    LnFwdArgs a = {};
    a.x = md.ws[0].x_in; a.M = (int)M; a.C = C;
    a.g2 = P(m, fp, md.bp[0].n1w); a.b2 = P(m, fp, md.bp[0].n1b); a.eps2 = 1e-6f; a.y2 = md.ws[0].a1; a.y2_lo = md.ws[0].a1l; a.stats2 = md.ws[0].st1;
    if (md.f8) { a.y2 = md.ws[0].a1h; a.y2_b16 = (m->infer || md.f8g || md.f8a) ? nullptr : md.ws[0].a1; }
    hz_ln_fwd(m, st, a, half);
    RUN(PC_LN, 0, ln_fwd(a, half, st));
  }
  for (int l = 0; l < L; ++l) {
    const BlockP& q = md.bp[l];
    BlockWS& w = md.ws[l];
    const bool spatial = (l % 2 == 0);
    const int mode = spatial ? 1 : 2;
    m->cur_kind = (md.is_rot ? 0 : 12) + 0 * 4 + LK_QKV;
    int rc = linear_fwd(m, st, fp, md.f8 ? w.a1h : w.a1, q.qkvw, q.qkvb, w.qkv, M, 3 * C, C, EPI_BIAS, nullptr, nullptr, nullptr, 0, T, N, nullptr, nullptr, nullptr,
                        w.a1l, w.qkvl, md.f8);
    if (rc) return rc;
    {
      const double eb = m->cfg.precision == 0 ? 4.0 : 2.0;
      HZ(st, "attention_fwd", HR(w.qkv, eb * M * 3 * C), HR(x3 ? w.qkvl : nullptr, eb * M * 3 * C), HW(w.ao, eb * M * C), HW(x3 ? w.aol : nullptr, eb * M * C),
         HW(spatial ? nullptr : w.lse, 4.0 * B * N * H * T), HW(xattn, xattn ? 16.0 * M * C : 0.0));
    }
    if (x3) {
      if (spatial) RUN(PC_ATTN, 12.0 * B * T * N * N * C, attn_spatial_fwd_x3((const bf16*)w.qkv, (const bf16*)w.qkvl, (bf16*)w.ao, (bf16*)w.aol, xattn, B, T, N, C, H, st, md.f8a));
      else RUN(PC_ATTN, 12.0 * B * N * (double)T * T * C, attn_temporal_fwd_x3((const bf16*)w.qkv, (const bf16*)w.qkvl, (bf16*)w.ao, (bf16*)w.aol, w.lse, xattn, B, T, N, C, H, st, md.f8a));
    } else if (spatial) RUN(PC_ATTN, 4.0 * B * T * N * N * C, attn_spatial_fwd(w.qkv, w.ao, half, B, T, N, C, H, st));
    else RUN(PC_ATTN, 4.0 * B * N * (double)T * T * C, attn_temporal_fwd(w.qkv, w.ao, w.lse, half, B, T, N, C, H, st));
    // block input: materialised (blocks 0 and 1: embedding / positional table involved), or - bf16 mode - recomputed in the
    // residual epilogue as the shared post-norm of the previous block's output (ln_fwd below does not store it then)
    const bool lazy_in = lazy_block_input(m, md, l);
    if (lazy_in) {
      const bool pspatial = ((l - 1) % 2 == 0);
      m->cur_kind = (md.is_rot ? 0 : 12) + 0 * 4 + LK_PROJ;
      rc = linear_fwd(m, st, fp, w.ao, q.pw, q.pb, w.x_mid, M, C, C, EPI_BIAS_RESID, nullptr, md.ws[l - 1].x_out,
                      branch_mask(m, md, l, 0, B, m->train), mode, T, N, md.ws[l - 1].stp, P(m, fp, pspatial ? md.sn_w : md.tn_w),
                      P(m, fp, pspatial ? md.sn_b : md.tn_b), w.aol, nullptr, md.f8a);
    } else {
      m->cur_kind = (md.is_rot ? 0 : 12) + 0 * 4 + LK_PROJ;
      rc = linear_fwd(m, st, fp, w.ao, q.pw, q.pb, w.x_mid, M, C, C, EPI_BIAS_RESID, nullptr, w.x_in,
                      branch_mask(m, md, l, 0, B, m->train), mode, T, N, nullptr, nullptr, nullptr, w.aol, nullptr, md.f8a);
    }
    if (rc) return rc;
    {
      LnFwdArgs a = {};
      a.x = w.x_mid; a.M = (int)M; a.C = C;
      a.g2 = P(m, fp, q.n2w); a.b2 = P(m, fp, q.n2b); a.eps2 = 1e-6f; a.y2 = w.a2; a.y2_lo = w.a2l; a.stats2 = w.st2;
      if (md.f8) { a.y2 = w.a2h; a.y2_b16 = (m->infer || md.f8g || md.f8a) ? nullptr : w.a2; }
      hz_ln_fwd(m, st, a, half);
    RUN(PC_LN, 0, ln_fwd(a, half, st));
    }
    // (inference, precision >= 1: gelu' - read by the fc2 dgrad only - is not written)
    m->cur_kind = (md.is_rot ? 0 : 12) + 0 * 4 + LK_FC1;
    rc = linear_fwd(m, st, fp, md.f8 ? w.a2h : w.a2, q.f1w, q.f1b, w.f, M, 2 * C, C, EPI_BIAS_GELU, (m->infer && m->cfg.precision >= 1) ? nullptr : w.z, nullptr, nullptr, 0, T, N, nullptr, nullptr, nullptr, w.a2l, w.fl,
                    md.f8, md.f8m || md.f8a);
    if (rc) return rc;
    m->cur_kind = (md.is_rot ? 0 : 12) + 0 * 4 + LK_FC2;
    rc = linear_fwd(m, st, fp, w.f, q.f2w, q.f2b, w.x_out, M, C, 2 * C, EPI_BIAS_RESID, nullptr, w.x_mid,
                    branch_mask(m, md, l, 1, B, m->train), mode, T, N, nullptr, nullptr, nullptr, w.fl, nullptr, md.f8m || md.f8a);
    if (rc) return rc;
    // shared post-norm (mix_ste.py:143,154,166,170), Temporal_pos_embed after the first spatial block (:149),
    // fused with the next block's norm1
    LnFwdArgs a = {};
    a.x = w.x_out; a.M = (int)M; a.C = C;
    a.g1 = P(m, fp, spatial ? md.sn_w : md.tn_w); a.b1 = P(m, fp, spatial ? md.sn_b : md.tn_b); a.eps1 = 1e-6f;
    a.pos = (l == 0) ? P(m, fp, md.tpos) : nullptr; a.T = T; a.J = N;
    a.x1 = (l + 1 < L) ? (lazy_block_input(m, md, l + 1) ? nullptr : md.ws[l + 1].x_in) : md.x_final;
    a.stats1 = w.stp;
    if (l + 1 < L) {
      a.g2 = P(m, fp, md.bp[l + 1].n1w); a.b2 = P(m, fp, md.bp[l + 1].n1b); a.eps2 = 1e-6f;
      a.y2 = md.ws[l + 1].a1; a.y2_lo = md.ws[l + 1].a1l; a.stats2 = md.ws[l + 1].st1;
      if (md.f8) { a.y2 = md.ws[l + 1].a1h; a.y2_b16 = (m->infer || md.f8g || md.f8a) ? nullptr : md.ws[l + 1].a1; }
    }
    hz_ln_fwd(m, st, a, half);
    RUN(PC_LN, 0, ln_fwd(a, half, st));
  }
  return MP_OK;
}

// LayerNorm backward (ln_bwd / ln_bwd2): the row kernel on st reads dy, x (or, recomputed input: x0 + stats0), the statistics, the skip gradient g
// (read-modify-write: dx = dskip + ...) and writes dx, the 2-byte copy and the per-workgroup partial sums; the reduction of the partials into the
// parameter gradients runs on st_param behind `ev` when given (param_stream(), elementwise.hip), else on st
static void hz_ln_bwd(mp_model* m, hipStream_t st, hipStream_t st_param, hipEvent_t ev, const char* name, const void* dy, double dy_elem, const float* x,
                      const float* stats, float* g, void* g_b16, const float* mask, const float* x0, const float* stats0, long M, int C, float* scratch,
                      long scratch_floats, float* dg1, float* db1, float* dg0, float* db0) {
  if (!m->hz) return;
  const double MC = (double)M * C;
  HZ(st, name, HR(dy, dy_elem * MC), HR(x, 4.0 * MC), HR(stats, 8.0 * M), HR(x0, 4.0 * MC), HR(stats0, 8.0 * M), HW(g, 4.0 * MC), HW(g_b16, 2.0 * MC),
     HR(mask, 4), HW(scratch, 4.0 * scratch_floats));
  const hipStream_t sp = hz_param_edge(m, st, st_param, ev);
  HZ(sp, "ln_bwd.param_reduce", HR(scratch, 4.0 * scratch_floats), HW(dg1, 4.0 * C), HW(db1, 4.0 * C), HW(dg0, 4.0 * C), HW(db0, 4.0 * C));
}
// on entry m->g holds dL/d x_final; on exit m->g holds dL/d (embedding output)
static int backbone_bwd_impl(mp_model* m, Module& md, const float* fp, float* fg, int B, hipStream_t st) {
  const int T = m->cfg.num_frame, N = md.N, C = md.C, H = md.H, L = 2 * md.depth;
  const long M = (long)B * T * N;
  const int half = m->cfg.precision >= 1;       // precision 2: the backward runs in bf16 on the hi planes
  float* g = m->g;
  bool post_done = false;
  // asynchronous weight gradients (rotations net on the main stream only)
  const bool wasync = m->wgrad_async && md.is_rot && st != m->st2;
  hipStream_t sw = wasync ? m->st3 : st;
  bool have_prev = false;       // a previous block's W events exist
  int par = 0;
  // LayerNorm parameter gradients: the reduction of each call's partial sums runs on the weight-gradient stream, from its own slice
  int ln_call = 0;
  hipStream_t lst = wasync ? sw : nullptr;
  hipEvent_t lev = wasync ? m->ev_heads : nullptr;
  auto ln_scratch = [&]() -> float* { return (wasync && ln_call < m->lnpart_n) ? m->lnpart + (long)(ln_call++) * m->lnpart_slice : m->small; };
  auto ln_floats = [&](float* p) -> long { return p == m->small ? m->small_floats : m->lnpart_slice; };
#define E_READY(i) do { if (wasync) { MP_HIP(ev_record(m, m->evE[par][i], st)); MP_HIP(ev_wait(m, sw, m->evE[par][i])); } } while (0)
#define W_DONE(i) do { if (wasync) MP_HIP(ev_record(m, m->evW[par][i], sw)); } while (0)
#define WAIT_W(p, i) do { if (wasync) MP_HIP(ev_wait(m, st, m->evW[p][i])); } while (0)
  for (int l = L - 1; l >= 0; --l, par ^= 1) {
    const BlockP& q = md.bp[l];
    BlockWS& w = md.ws[l];
    const bool spatial = (l % 2 == 0);
    const int mode = spatial ? 1 : 2;
    // (a) shared post-norm (+ Temporal_pos_embed gradient behind block 0); skipped when the previous iteration's fused
    // norm1/post-norm backward (h) has already produced dL/d x_out[l]
    const float* mk2 = branch_mask(m, md, l, 1, B, m->train);
    if (!post_done) {
      if (l == 0) {
        HZ(st, "tpos_grad", HR(g, 4.0 * M * C), HW(G(m, fg, md.tpos), 4.0 * T * C));
        RUN(PC_OTHER, 0, tpos_grad(g, G(m, fg, md.tpos), B, T, N, C, st));
      }
      float* lsc = ln_scratch();
      hz_ln_bwd(m, st, lsc == m->small ? nullptr : lst, lev, "ln_bwd.postnorm", g, 4.0, w.x_out, w.stp, g, m->g_b16, mk2, nullptr, nullptr, M, C, lsc, ln_floats(lsc),
                G(m, fg, spatial ? md.sn_w : md.tn_w), G(m, fg, spatial ? md.sn_b : md.tn_b), nullptr, nullptr);
      RUN(PC_LN, 0, ln_bwd(g, 0, w.x_out, w.stp, P(m, fp, spatial ? md.sn_w : md.tn_w), nullptr, g, m->g_b16, mk2, mode, T, N,
                           G(m, fg, spatial ? md.sn_w : md.tn_w), G(m, fg, spatial ? md.sn_b : md.tn_b), (int)M, C, lsc,
                           ln_floats(lsc), st, lsc == m->small ? nullptr : lst, lev, 1.0f, nullptr, md.f8m ? m->gsc : nullptr));
    }
    post_done = false;
    // (b) mlp branch: fc2 (gb = DropPath-scaled branch gradient; a bf16 copy emitted by the LN backward in precision 1)
    const void* gb = half ? (const void*)m->g_b16 : (const void*)g;
    if (mk2 && !half) {
      HZ(st, "scale_rows", HR(g, 4.0 * M * C), HW(m->tmpMask, 4.0 * M * C));
      RUN(PC_OTHER, 0, scale_rows(g, mk2, mode, m->tmpMask, 0, (int)M, C, T, N, st));
      gb = m->tmpMask;
    }
    E_READY(0);                                                    // gb is ready
    m->cur_kind = (md.is_rot ? 0 : 12) + 2 * 4 + LK_FC2;
    int rc = linear_wgrad(m, sw, gb, 0, w.f, G(m, fg, q.f2w), G(m, fg, q.f2b), M, C, 2 * C, md.f8m, md.f8a);      // f8m: gb and f are fp16; f8a: f alone is (rounded to bf16 per fragment)
    if (rc) return rc;
    W_DONE(0);
    if (have_prev) WAIT_W(par ^ 1, 1);                             // previous block's fc1 wgrad still reads tmp2C
    // f8g: dz leaves as scaled fp16; f8m: gb already carries the scale (fp16 operands), so the epilogue adds none
    m->cur_kind = (md.is_rot ? 0 : 12) + 1 * 4 + LK_FC2;
    rc = linear_dgrad(m, st, fp, gb, 0, q.f2w, m->tmp2C, 0, M, C, 2 * C, w.z, md.f8m, md.f8m ? m->gsc + 3 : (md.f8g ? m->gsc : nullptr));
    if (rc) return rc;
    // (c) fc1
    E_READY(1);                                                    // dz (tmp2C) is ready
    m->cur_kind = (md.is_rot ? 0 : 12) + 2 * 4 + LK_FC1;
    rc = linear_wgrad(m, sw, m->tmp2C, 0, (md.f8g || md.f8a) ? w.a2h : w.a2, G(m, fg, q.f1w), G(m, fg, q.f1b), M, 2 * C, C, md.f8g, md.f8a);
    if (rc) return rc;
    W_DONE(1);
    m->cur_kind = (md.is_rot ? 0 : 12) + 1 * 4 + LK_FC1;
    rc = linear_dgrad(m, st, fp, m->tmp2C, 0, q.f1w, m->tmpC, 0, M, 2 * C, C, nullptr, md.f8g);     // d(norm2 out): bf16 in precision 1
    if (rc) return rc;
    WAIT_W(par, 0);                                                // the fc2 wgrad must be done with gb before (d) rewrites it
    // (d) norm2 + skip
    const float* mk1 = branch_mask(m, md, l, 0, B, m->train);
    float* lsc2 = ln_scratch();
    hz_ln_bwd(m, st, lsc2 == m->small ? nullptr : lst, lev, "ln_bwd.norm2", m->tmpC, half ? 2.0 : 4.0, w.x_mid, w.st2, g, m->g_b16, mk1, nullptr, nullptr, M, C, lsc2,
              ln_floats(lsc2), G(m, fg, q.n2w), G(m, fg, q.n2b), nullptr, nullptr);
    RUN(PC_LN, 0, ln_bwd(m->tmpC, half, w.x_mid, w.st2, P(m, fp, q.n2w), g, g, m->g_b16, mk1, mode, T, N, G(m, fg, q.n2w),
                         G(m, fg, q.n2b), (int)M, C, lsc2, ln_floats(lsc2), st, lsc2 == m->small ? nullptr : lst, lev, md.rs,
                         md.f8g ? m->gsc + 1 : nullptr));
    // (e) attention branch: proj
    gb = half ? (const void*)m->g_b16 : (const void*)g;
    if (mk1 && !half) {
      HZ(st, "scale_rows", HR(g, 4.0 * M * C), HW(m->tmpMask, 4.0 * M * C));
      RUN(PC_OTHER, 0, scale_rows(g, mk1, mode, m->tmpMask, 0, (int)M, C, T, N, st));
      gb = m->tmpMask;
    }
    E_READY(2);                                                    // gb (second half of the block) is ready
    m->cur_kind = (md.is_rot ? 0 : 12) + 2 * 4 + LK_PROJ;
    rc = linear_wgrad(m, sw, gb, 0, w.ao, G(m, fg, q.pw), G(m, fg, q.pb), M, C, C, false, md.f8a);
    if (rc) return rc;
    W_DONE(2);
    m->cur_kind = (md.is_rot ? 0 : 12) + 1 * 4 + LK_PROJ;
    rc = linear_dgrad(m, st, fp, gb, 0, q.pw, m->tmpC, 0, M, C, C, nullptr);          // d(attention out): bf16 in bf16 mode
    if (rc) return rc;
    if (have_prev) WAIT_W(par ^ 1, 3);                             // previous block's qkv wgrad still reads tmp3C
    // (f) attention core
    {
      const double eb = half ? 2.0 : 4.0;
      HZ(st, "attention_bwd", HR(w.qkv, eb * M * 3 * C), HR(spatial ? nullptr : w.ao, eb * M * C), HR(m->tmpC, eb * M * C), HR(spatial ? nullptr : w.lse, 4.0 * B * N * H * T),
         HW(spatial ? nullptr : m->delta, 4.0 * B * N * H * T), HW(m->tmp3C, eb * M * 3 * C));
      AttnGradF16Scope f16_out(md.f8g ? m->gsc : nullptr);      // f8g: dqkv leaves as scaled fp16 (its two consumers below run on fp16 operands)
      if (spatial) RUN(PC_ATTN, 10.0 * B * T * N * N * C, attn_spatial_bwd(w.qkv, m->tmpC, m->tmp3C, half, B, T, N, C, H, st));
      else {
        attn_out_f16_override(md.f8a ? 1 : 0);      // f8a: O exists as an fp16 plane
        ProfScope ps__(m, st, PC_ATTN, 10.0 * B * N * (double)T * T * C);
        const int rc__ = attn_temporal_bwd(w.qkv, w.ao, m->tmpC, w.lse, m->delta, m->tmp3C, half, B, T, N, C, H, st);
        attn_out_f16_override(0);
        if (rc__) return rc__;
      }
    }
    // (g) qkv
    E_READY(3);                                                    // dqkv (tmp3C) is ready
    m->cur_kind = (md.is_rot ? 0 : 12) + 2 * 4 + LK_QKV;
    rc = linear_wgrad(m, sw, m->tmp3C, 0, (md.f8g || md.f8a) ? w.a1h : w.a1, G(m, fg, q.qkvw), G(m, fg, q.qkvb), M, 3 * C, C, md.f8g, md.f8a);
    if (rc) return rc;
    W_DONE(3);
    m->cur_kind = (md.is_rot ? 0 : 12) + 1 * 4 + LK_QKV;
    rc = linear_dgrad(m, st, fp, m->tmp3C, 0, q.qkvw, m->tmpC, 0, M, 3 * C, C, nullptr, md.f8g);   // d(norm1 out): bf16 in precision 1
    if (rc) return rc;
    WAIT_W(par, 2);                                                // the proj wgrad must be done with gb before (h) rewrites it
    have_prev = true;
    // (h) norm1 + skip; for l >= 2 fused with the shared post-norm backward of block l-1 (its (a) step)
    if (l >= 2 && C <= 512) {
      const BlockP& qp = md.bp[l - 1];
      (void)qp;
      BlockWS& wp = md.ws[l - 1];
      const bool pspatial = ((l - 1) % 2 == 0);
      const float* mkp = branch_mask(m, md, l - 1, 1, B, m->train);
      float* lsc3 = ln_scratch();
      hz_ln_bwd(m, st, lsc3 == m->small ? nullptr : lst, lev, "ln_bwd2.norm1+postnorm", m->tmpC, half ? 2.0 : 4.0, w.x_in, w.st1, g, m->g_b16, mkp, wp.x_out, wp.stp, M, C,
                lsc3, ln_floats(lsc3), G(m, fg, q.n1w), G(m, fg, q.n1b), G(m, fg, pspatial ? md.sn_w : md.tn_w), G(m, fg, pspatial ? md.sn_b : md.tn_b));
      RUN(PC_LN, 0, ln_bwd2(m->tmpC, half, w.x_in, w.st1, P(m, fp, q.n1w), g, wp.x_out, wp.stp, P(m, fp, pspatial ? md.sn_w : md.tn_w),
                            P(m, fp, pspatial ? md.sn_b : md.tn_b), g, m->g_b16, mkp, pspatial ? 1 : 2, T, N, G(m, fg, q.n1w), G(m, fg, q.n1b),
                            G(m, fg, pspatial ? md.sn_w : md.tn_w), G(m, fg, pspatial ? md.sn_b : md.tn_b), (int)M, C, lsc3,
                            ln_floats(lsc3), st, lsc3 == m->small ? nullptr : lst, lev, md.rs, md.f8g ? m->gsc + 1 : nullptr,
                            md.f8m ? m->gsc : nullptr));
      post_done = true;
    } else {
      float* lsc4 = ln_scratch();
      hz_ln_bwd(m, st, lsc4 == m->small ? nullptr : lst, lev, "ln_bwd.norm1", m->tmpC, half ? 2.0 : 4.0, w.x_in, w.st1, g, nullptr, nullptr, nullptr, nullptr, M, C, lsc4,
                ln_floats(lsc4), G(m, fg, q.n1w), G(m, fg, q.n1b), nullptr, nullptr);
      RUN(PC_LN, 0, ln_bwd(m->tmpC, half, w.x_in, w.st1, P(m, fp, q.n1w), g, g, nullptr, nullptr, 0, T, N, G(m, fg, q.n1w),
                           G(m, fg, q.n1b), (int)M, C, lsc4, ln_floats(lsc4), st, lsc4 == m->small ? nullptr : lst, lev, md.rs,
                           md.f8g ? m->gsc + 1 : nullptr));
    }
    // gradient bucket of layer i = l / 2 (STE_i and TTE_i: one contiguous range of the flat buffer): everything that writes it has been
    // enqueued - the weight-gradient stream first waits for the main stream's position, then carries the event
    if (md.is_rot && (l % 2 == 0) && (size_t)(l / 2) < m->ev_bucket.size()) {
      if (sw != st) {
        MP_HIP(ev_record(m, m->ev_sync, st));
        MP_HIP(ev_wait(m, sw, m->ev_sync));
      }
      MP_HIP(ev_record(m, m->ev_bucket[l / 2], sw));
    }
  }
  if (wasync && have_prev) {    // the last block's fc1 / qkv weight gradients
    WAIT_W(par ^ 1, 1);
    WAIT_W(par ^ 1, 3);
  }
#undef E_READY
#undef W_DONE
#undef WAIT_W
  return MP_OK;
}

// MuReadout (readout != 1): y = W (readout * x) + b is evaluated with the scaled copy W_eff = readout * W (refresh_head_weights, once per
// forward); the kernels' weight gradient d W_eff lands in a zeroed scratch and readout * d W_eff is added to the gradient buffer afterwards
static void head_params(const mp_model* m, const Module& md, const float* fp, HeadParams& hp) {
  const long WC = (long)md.O * md.C;
  for (int k = 0; k < md.K; ++k) {
    hp.gamma[k] = P(m, fp, md.hg[k]); hp.beta[k] = P(m, fp, md.hb[k]); hp.b[k] = P(m, fp, md.hbias[k]);
    hp.W[k] = md.hw_eff != nullptr ? md.hw_eff + k * WC : P(m, fp, md.hw[k]);
  }
}
static void head_grads(const mp_model* m, const Module& md, float* fg, HeadGrads& hg) {
  const long WC = (long)md.O * md.C;
  for (int k = 0; k < md.K; ++k) {
    hg.gamma[k] = G(m, fg, md.hg[k]); hg.beta[k] = G(m, fg, md.hb[k]); hg.b[k] = G(m, fg, md.hbias[k]);
    hg.W[k] = md.hdw != nullptr ? md.hdw + k * WC : G(m, fg, md.hw[k]);
  }
}
// floats of the flat buffer from the first head parameter of a module to the end of its last one (the heads' slots are contiguous)
static long head_param_floats(const mp_model* m, const Module& md) {
  const bool rmcl = md.is_rot && m->cfg.arch == 0;
  const ParamDesc& first = m->params[md.hg[0]];
  const ParamDesc& last = m->params[rmcl ? md.sb[md.K - 1] : md.hbias[0]];
  return last.offset + (last.numel + 63) / 64 * 64 - first.offset;
}
static int refresh_head_weights(mp_model* m, const Module& md, const float* fp, hipStream_t st) {
  if (md.hw_eff == nullptr) return MP_OK;
  const long WC = (long)md.O * md.C;
  for (int k = 0; k < md.K; ++k) {
    int rc = scale_copy(md.hw_eff + k * WC, P(m, fp, md.hw[k]), md.readout, WC, st);
    if (rc) return rc;
  }
  return MP_OK;
}
static int zero_head_grad_scratch(mp_model* m, const Module& md, hipStream_t st) {
  if (md.hdw == nullptr) return MP_OK;
  HZ(st, "heads.zero_dw_scratch", HW(md.hdw, 4.0 * md.K * md.O * md.C));
  MP_HIP(hipMemsetAsync(md.hdw, 0, sizeof(float) * (size_t)md.K * md.O * md.C, st));
  return MP_OK;
}
static int flush_head_grads(mp_model* m, const Module& md, float* fg, hipStream_t st) {
  if (md.hdw == nullptr) return MP_OK;
  const long WC = (long)md.O * md.C;
  for (int k = 0; k < md.K; ++k) {
    HZ(st, "heads.flush_dw", HR(md.hdw + k * WC, 4.0 * WC), HW(G(m, fg, md.hw[k]), 4.0 * WC));
    int rc = axpy_scaled(G(m, fg, md.hw[k]), md.hdw + k * WC, md.readout, WC, st);
    if (rc) return rc;
  }
  return MP_OK;
}

}  // namespace mp

// =================================================================================================
// C ABI: model engine
// =================================================================================================
extern "C" {

int mp_model_create(const mp_model_config* cfg, mp_model** out) {
  MP_CHECK(cfg != nullptr && out != nullptr, MP_ERR_ARG, "mp_model_create: null argument");
  MP_CHECK(cfg->precision >= 0 && cfg->precision <= 2, MP_ERR_ARG, "mp_model_create: precision %d (0 = fp32, 1 = bf16, 2 = bf16x3)", cfg->precision);
  MP_CHECK(cfg->precision == 0 || (cfg->embed_dim_rot % 8 == 0 && cfg->embed_dim_seg % 8 == 0), MP_ERR_ARG,
           "mp_model_create: bf16 precision needs embedding widths that are multiples of 8");
  MP_CHECK(cfg->num_joints == 17 && cfg->num_bones == 16, MP_ERR_ARG, "mp_model_create: the decoder is built for the 17-joint H36M tree");
  MP_CHECK(cfg->num_frame >= 2 && cfg->max_batch >= 0, MP_ERR_ARG, "mp_model_create: num_frame >= 2, max_batch >= 0");
  MP_CHECK(cfg->embed_dim_rot % cfg->num_heads_rot == 0 && cfg->embed_dim_seg % cfg->num_heads_seg == 0, MP_ERR_ARG,
           "mp_model_create: embed dim must be divisible by heads");
  MP_CHECK(cfg->arch >= 0 && cfg->arch <= 2, MP_ERR_ARG, "mp_model_create: arch %d (0 rmcl_manifold, 1 manifold, 2 mixste)", cfg->arch);
  MP_CHECK(cfg->arch != 0 || (cfg->n_hyp >= 1 && cfg->n_hyp <= 8), MP_ERR_ARG, "mp_model_create: n_hyp in 1..8");
  MP_CHECK(cfg->rot_rep_dim == 0 || cfg->rot_rep_dim == 4 || cfg->rot_rep_dim == 6, MP_ERR_ARG,
           "mp_model_create: rot_rep_dim %d (4 or 6; 0 = 6)", cfg->rot_rep_dim);
  MP_CHECK(cfg->f16f8 >= 0 && cfg->f16f8 <= 3 && (cfg->f16_backward == 0 || cfg->f16_backward == 1) && cfg->streams >= 0 && cfg->streams <= 3, MP_ERR_ARG,
           "mp_model_create: f16f8 %d (0..3), f16_backward %d (0/1), streams %d (bit set 0..3)", cfg->f16f8, cfg->f16_backward, cfg->streams);
  MP_CHECK(cfg->f16f8 != 3 || cfg->f16_backward == 0, MP_ERR_ARG, "mp_model_create: f16f8 = 3 (all four Linear layers) keeps the bf16 backward: f16_backward must be 0");
  MP_CHECK(cfg->debug >= 0 && cfg->debug <= 1, MP_ERR_ARG, "mp_model_create: debug %d (bit 0 = stream-hazard check)", cfg->debug);
  MP_CHECK(cfg->f16f8 == 0 || cfg->precision == 2, MP_ERR_ARG, "mp_model_create: f16f8 operands belong to precision 2 (bf16x3)");
  MP_CHECK(cfg->f16_backward == 0 || cfg->f16f8 >= 1, MP_ERR_ARG, "mp_model_create: f16_backward needs f16f8 >= 1");
  MP_CHECK(cfg->f16f8 != 2 || cfg->f16_backward == 1, MP_ERR_ARG, "mp_model_create: f16f8 = 2 (the fc2 layer) needs f16_backward");
  mp_model* m = new mp_model();
  m->cfg = *cfg;
  if (cfg->rot_rep_dim == 0) m->cfg.rot_rep_dim = 6;
  if (cfg->arch != 0) m->cfg.n_hyp = 1;
  m->has_seg = cfg->arch != 2;          // arch 2: the bare MixSTE regressor (main_h36m_lifting.py:617-628): no bones net, no decoder
  m->rot.prefix = m->has_seg ? "rotations_module." : ""; m->rot.is_rot = true;
  m->rot.N = cfg->num_joints; m->rot.C = cfg->embed_dim_rot; m->rot.H = cfg->num_heads_rot; m->rot.depth = cfg->depth_rot;
  m->rot.K = m->cfg.n_hyp; m->rot.O = (cfg->arch == 2) ? 3 : m->cfg.rot_rep_dim + ((cfg->arch == 0) ? 1 : 0);   // + the score-embedding channel; mixste: xyz
  m->seg.prefix = "segments_module."; m->seg.is_rot = false;
  m->seg.N = cfg->num_bones; m->seg.C = cfg->embed_dim_seg; m->seg.H = cfg->num_heads_seg; m->seg.depth = cfg->depth_seg;
  m->seg.K = 1; m->seg.O = 1;
  m->rot.qk_scale = cfg->qk_scale_rot; m->rot.rs = cfg->resid_scale_rot != 0.f ? cfg->resid_scale_rot : 1.0f;
  m->rot.readout = cfg->readout_mult_rot != 0.f ? cfg->readout_mult_rot : 1.0f;
  m->seg.qk_scale = cfg->qk_scale_seg; m->seg.rs = cfg->resid_scale_seg != 0.f ? cfg->resid_scale_seg : 1.0f;
  m->seg.readout = cfg->readout_mult_seg != 0.f ? cfg->readout_mult_seg : 1.0f;
  MP_CHECK(m->rot.qk_scale >= 0.f && m->seg.qk_scale >= 0.f && m->rot.rs > 0.f && m->seg.rs > 0.f, MP_ERR_ARG,
           "mp_model_create: attention / residual scales must be positive (0 = default)");
  build_module_params(m, m->rot);
  if (m->has_seg) build_module_params(m, m->seg);
  m->rot.mask_base = 0;
  m->seg.mask_base = (int)m->rot.masks.size();
  if (cfg->max_batch == 0) {   // layout-only handle (parameter / mask layout queries, no device needed)
    *out = m;
    return MP_OK;
  }
  // f16f8 inputs: the rotations net when its width lets every such GEMM run the persistent 256 x 256 kernel (N = 3 C, 2 C multiples of 256, K = C of 64)
  m->rot.f8 = cfg->precision == 2 && cfg->f16f8 >= 1 && m->rot.C % 256 == 0 && m->rot.C >= 256;
  m->seg.f8 = false;
  m->rot.f8m = false;
  m->rot.f8g = m->rot.f8 && cfg->f16_backward != 0 && attn_tmfma_supported(cfg->num_frame, m->rot.C / m->rot.H) &&
               attn_smfma_supported(m->rot.N, m->rot.C / m->rot.H, m->rot.H);
  m->rot.f8m = m->rot.f8g && (m->rot.rs == 0.f || m->rot.rs == 1.0f) && cfg->f16f8 == 2;      // (a residual scale other than 1 - muP - keeps the tiled bf16x3 fc2)
  // f16f8 = 3: all four layers, bf16 backward.  Needs the MFMA attention kernels with head dim 64 (their f16f8 output form) and the persistent residual
  // epilogue (residual scale 1); a model that does not qualify runs plain bf16x3 (three bf16 products everywhere) - so 3 is a safe default of callers
  m->rot.f8a = m->rot.f8 && cfg->f16f8 == 3 && (m->rot.rs == 0.f || m->rot.rs == 1.0f) && m->rot.C / m->rot.H == 64 && m->rot.C % m->rot.H == 0 &&
               !attn_x3_needs_scratch(0, cfg->num_frame, m->rot.N, m->rot.C, m->rot.H) && !attn_x3_needs_scratch(1, cfg->num_frame, m->rot.N, m->rot.C, m->rot.H);
  if (cfg->f16f8 == 3 && !m->rot.f8a) m->rot.f8 = false;
  Bump dry;
  carve_all(m, dry);
  m->arena_bytes = dry.off;
  hipError_t e = hipMalloc((void**)&m->arena, m->arena_bytes);
  if (e != hipSuccess) {
    set_error("mp_model_create: hipMalloc(%zu bytes) failed: %s", m->arena_bytes, hipGetErrorString(e));
    (void)hipGetLastError();            // the refusal is reported through the return code: do not leave it as the thread's sticky HIP error
    delete m;
    return MP_ERR_HIP;
  }
  Bump real;
  real.base = m->arena;
  carve_all(m, real);
  use_scratch(m, 0);
  auto side_stream = [](hipStream_t* st, int) { return hipStreamCreateWithFlags(st, hipStreamNonBlocking); };
  if (side_stream(&m->st2, 0) != hipSuccess || hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming) != hipSuccess) {
    set_error("mp_model_create: could not create the side stream / events");
    (void)hipFree(m->arena);
    delete m;
    return MP_ERR_HIP;
  }
  {
    m->wgrad_async = (cfg->streams & 2) == 0;
    bool ok = side_stream(&m->st3, 1) == hipSuccess;
    for (int a = 0; a < 2 && ok; ++a)
      for (int b = 0; b < 4 && ok; ++b)
        ok = hipEventCreateWithFlags(&m->evE[a][b], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&m->evW[a][b], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&m->ev_heads, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&m->ev_sync, hipEventDisableTiming) == hipSuccess;
    m->ev_bucket.assign((size_t)m->rot.depth, nullptr);
    for (auto& e : m->ev_bucket) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
      set_error("mp_model_create: could not create the wgrad stream / events");
      (void)hipFree(m->arena);
      delete m;
      return MP_ERR_HIP;
    }
  }
  if (cfg->debug & 1) m->hz.reset(new HazardTracker());
  e = hipMemset(m->dscore_zero, 0, sizeof(float) * (size_t)cfg->max_batch * m->rot.K * cfg->num_frame);
  if (e != hipSuccess) {
    set_error("mp_model_create: hipMemset failed: %s", hipGetErrorString(e));
    (void)hipFree(m->arena);
    delete m;
    return MP_ERR_HIP;
  }
  *out = m;
  return MP_OK;
}

void mp_model_destroy(mp_model* m) {
  if (!m) return;
  for (auto& e : m->ev) (void)hipEventDestroy(e);
  if (m->st2) { (void)hipStreamSynchronize(m->st2); (void)hipStreamDestroy(m->st2); }
  if (m->st3) { (void)hipStreamSynchronize(m->st3); (void)hipStreamDestroy(m->st3); }
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 4; ++b) {
      if (m->evE[a][b]) (void)hipEventDestroy(m->evE[a][b]);
      if (m->evW[a][b]) (void)hipEventDestroy(m->evW[a][b]);
    }
  if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
  if (m->ev_join) (void)hipEventDestroy(m->ev_join);
  if (m->ev_heads) (void)hipEventDestroy(m->ev_heads);
  if (m->ev_sync) (void)hipEventDestroy(m->ev_sync);
  for (auto& e : m->ev_bucket) if (e) (void)hipEventDestroy(e);
  if (m->arena) (void)hipFree(m->arena);
  delete m;
}

int64_t mp_model_workspace_bytes(const mp_model* m) { return m ? (int64_t)m->arena_bytes : 0; }
int mp_model_num_params(const mp_model* m) { return m ? (int)m->params.size() : 0; }
int64_t mp_model_flat_size(const mp_model* m) { return m ? m->flat_size : 0; }

int mp_model_param_info(const mp_model* m, int index, char* name, int name_cap, int64_t* offset, int64_t* numel) {
  MP_CHECK(m && index >= 0 && index < (int)m->params.size(), MP_ERR_ARG, "mp_model_param_info: bad index %d", index);
  const ParamDesc& d = m->params[index];
  if (name && name_cap > 0) snprintf(name, name_cap, "%s", d.name.c_str());
  if (offset) *offset = d.offset;
  if (numel) *numel = d.numel;
  return MP_OK;
}

int mp_model_num_mask_branches(const mp_model* m) { return m ? (int)(m->rot.masks.size() + m->seg.masks.size()) : 0; }

int mp_model_mask_info(const mp_model* m, int B, int index, char* name, int name_cap, int64_t* offset, int64_t* count,
                       float* keep_prob) {
  MP_CHECK(m && index >= 0 && index < mp_model_num_mask_branches(m), MP_ERR_ARG, "mp_model_mask_info: bad index %d", index);
  long off = 0;
  int i = 0;
  const Module* mods[2] = {&m->rot, &m->seg};
  for (const Module* q : mods) {
    for (const auto& b : q->masks) {
      const long cnt = b.spatial ? (long)B * m->cfg.num_frame : (long)B * q->N;
      if (i == index) {
        if (name && name_cap > 0) snprintf(name, name_cap, "%s", b.name.c_str());
        if (offset) *offset = off;
        if (count) *count = cnt;
        if (keep_prob) *keep_prob = b.keep;
        return MP_OK;
      }
      off += cnt;
      ++i;
    }
  }
  return MP_ERR_ARG;
}

int64_t mp_model_mask_floats(const mp_model* m, int B) {
  if (!m) return 0;
  long off = 0;
  const Module* mods[2] = {&m->rot, &m->seg};
  for (const Module* q : mods)
    for (const auto& b : q->masks) off += b.spatial ? (long)B * m->cfg.num_frame : (long)B * q->N;
  return off;
}

int mp_model_forward(mp_model* m, const float* fp, const float* x, int B, float* poses, float* scores, int train,
                     const float* masks_override, uint64_t seed, uint64_t step, void* stream) {
  MP_CHECK(m && fp && x && poses, MP_ERR_ARG, "mp_model_forward: null argument");
  MP_CHECK(B >= 1 && B <= m->cfg.max_batch, MP_ERR_ARG, "mp_model_forward: batch %d outside 1..max_batch=%d", B, m->cfg.max_batch);
  MP_CHECK(m->cfg.arch != 0 || scores != nullptr, MP_ERR_ARG, "mp_model_forward: scores buffer required for rmcl_manifold");
  hipStream_t st = (hipStream_t)stream;
  const int T = m->cfg.num_frame, J = m->cfg.num_joints, S = m->cfg.num_bones, K = m->rot.K;
  m->B = B;
  m->buckets_recorded = false;         // the bucket events belong to a backward of an earlier forward
  m->train = (train & 1) != 0 && m->cfg.drop_path_rate > 0.f;
  m->infer = (train & 2) != 0;         // no backward will follow: tensors only the backward reads are not written
  m->x_in = x;
  if (m->train) {
    if (masks_override) {
      HZ(st, "masks.copy", HW(m->maskbuf, 4.0 * mp_model_mask_floats(m, B)));
      MP_HIP(hipMemcpyAsync(m->maskbuf, masks_override, sizeof(float) * mp_model_mask_floats(m, B), hipMemcpyDeviceToDevice, st));
    } else {
      std::vector<MaskDesc> ds;
      long off = 0;
      const Module* mods[2] = {&m->rot, &m->seg};
      for (const Module* q : mods)
        for (const auto& b : q->masks) {
          const int cnt = b.spatial ? B * T : B * q->N;
          ds.push_back({(int)off, cnt, b.keep});
          off += cnt;
        }
      HZ(st, "droppath_masks", HW(m->maskbuf, 4.0 * off));
      RUN(PC_OTHER, 0, droppath_masks(m->maskbuf, ds.data(), (int)ds.size(), seed, step, st));
    }
  }
  if (m->cfg.precision >= 1) HZ(st, "weight_shadow", HW(m->wbf, 2.0 * m->flat_size), HW(m->wbf_lo, 2.0 * m->flat_size), HW(m->w16, 2.0 * m->flat_size), HW(m->w8, 2.0 * m->flat_size));
  if (m->cfg.precision == 1) RUN(PC_OTHER, 0, cast_to_bf16(fp, m->wbf, m->flat_size, st));
  if (m->cfg.precision == 2) RUN(PC_OTHER, 0, cast_to_bf16x2(fp, m->wbf, m->wbf_lo, m->flat_size, st));
  if (m->w16 != nullptr) RUN(PC_OTHER, 0, cast_to_f16f8(fp, m->w16, m->w8, m->flat_size, 1, st));
  // fork: the side stream may start once the masks / bf16 weights above are in place
  const hipStream_t side = (m->cfg.streams & 1) ? st : m->st2;
  MP_HIP(ev_record(m, m->ev_fork, st));
  MP_HIP(ev_wait(m, side, m->ev_fork));
  use_scratch(m, 0);
  // rotations backbone (mix_ste.py:128-173)
  const long Mr = (long)B * T * J, Ms = (long)B * T * S;
  if (m->hz) m->hz->prune();      // drop the records every stream has synchronised past (the previous steps)
  HZ(st, "embed_fwd", HW(m->rot.ws[0].x_in, 4.0 * Mr * m->rot.C));
  RUN(PC_OTHER, 0, embed_fwd(x, P(m, fp, m->rot.emb_w), P(m, fp, m->rot.emb_b), P(m, fp, m->rot.spos), m->rot.ws[0].x_in, (int)Mr,
                             m->rot.C, J, st));
  int rc = backbone_fwd(m, m->rot, fp, B, st);
  if (rc) return rc;
  HeadParams hp;
  head_params(m, m->rot, fp, hp);
  rc = refresh_head_weights(m, m->rot, fp, st);
  if (rc) return rc;
  HZ(st, "heads_fwd", HR(m->rot.x_final, 4.0 * Mr * m->rot.C), HW(m->rot.headout, 4.0 * K * Mr * m->rot.O), HW(m->rot.hstats, 8.0 * Mr), HW(m->rot.hfold, 4.0 * heads_fold_floats(m->rot.C)));
  if (heads_use_mfma(K, m->rot.O, m->rot.C))
    RUN(PC_OTHER, 0, heads_fwd_mfma(m->rot.x_final, hp, K, m->rot.O, m->rot.headout, m->rot.hstats, (int)Mr, m->rot.C, m->rot.hfold, st));
  else
    RUN(PC_OTHER, 0, heads_fwd(m->rot.x_final, hp, K, m->rot.O, m->rot.headout, m->rot.hstats, (int)Mr, m->rot.C, st));
  if (m->cfg.arch == 0) {
    ScoreParams sp;
    for (int k = 0; k < K; ++k) { sp.w[k] = P(m, fp, m->rot.sw[k]); sp.b[k] = P(m, fp, m->rot.sb[k]); }
    HZ(st, "scores_fwd", HR(m->rot.headout, 4.0 * K * Mr * m->rot.O), HW(scores, 4.0 * B * K * T));
    RUN(PC_OTHER, 0, scores_fwd(m->rot.headout, sp, K, m->rot.O, scores, B, T, J, st));
  }
  if (!m->has_seg) {   // MixSTE.forward (mix_ste.py:175-191): the head output IS the pose, rows (b, t, j) = the (B, 1, T, J, 3) layout
    MP_HIP(hipMemcpyAsync(poses, m->rot.headout, sizeof(float) * Mr * 3, hipMemcpyDeviceToDevice, st));
    return MP_OK;
  }
  // bones net (manifold_mix_ste.py:139-154) on the side stream, concurrently with the rotations net enqueued above
  {
    hipStream_t main_st = st;
    st = side;
    use_scratch(m, 1);
    HZ(st, "bones_embed_fwd", HW(m->seg.ws[0].x_in, 4.0 * Ms * m->seg.C));
    RUN(PC_OTHER, 0, bones_embed_fwd(x, P(m, fp, m->seg.emb_w), P(m, fp, m->seg.emb_b), P(m, fp, m->seg.spos), m->seg.ws[0].x_in,
                                     B * T, J * 2, S * m->seg.C, st));
    rc = backbone_fwd(m, m->seg, fp, B, st);
    if (rc) return rc;
    HeadParams hs;
    head_params(m, m->seg, fp, hs);
    rc = refresh_head_weights(m, m->seg, fp, st);
    if (rc) return rc;
    HZ(st, "heads_fwd.seg", HR(m->seg.x_final, 4.0 * Ms * m->seg.C), HW(m->seg.headout, 4.0 * Ms), HW(m->seg.hstats, 8.0 * Ms), HW(m->seg.hfold, 4.0 * heads_fold_floats(m->seg.C)));
    if (heads_use_mfma(1, 1, m->seg.C))
      RUN(PC_OTHER, 0, heads_fwd_mfma(m->seg.x_final, hs, 1, 1, m->seg.headout, m->seg.hstats, (int)Ms, m->seg.C, m->seg.hfold, st));
    else
      RUN(PC_OTHER, 0, heads_fwd(m->seg.x_final, hs, 1, 1, m->seg.headout, m->seg.hstats, (int)Ms, m->seg.C, st));
    HZ(st, "bones_mean_fwd", HR(m->seg.headout, 4.0 * Ms), HW(m->lengths, 4.0 * B * S));
    RUN(PC_OTHER, 0, bones_mean_fwd(m->seg.headout, m->lengths, B, T, S, st));
    MP_HIP(ev_record(m, m->ev_join, side));
    st = main_st;
    use_scratch(m, 0);
    MP_HIP(ev_wait(m, st, m->ev_join));
  }
  // manifold decoder (pose_decoder.py:32-55)
  HZ(st, "fk_decode_fwd", HR(m->rot.headout, 4.0 * K * Mr * m->rot.O), HR(m->lengths, 4.0 * B * S), HW(poses, 12.0 * B * K * T * J));
  RUN(PC_OTHER, 0, fk_decode_fwd(m->rot.headout, m->rot.O, m->cfg.rot_rep_dim, m->lengths, poses, B, K, T, st));
  return MP_OK;
}

int mp_model_backward(mp_model* m, const float* fp, float* fg, const float* d_poses, const float* d_scores, void* stream) {
  MP_CHECK(m && fp && fg && d_poses, MP_ERR_ARG, "mp_model_backward: null argument");
  MP_CHECK(m->B >= 1, MP_ERR_STATE, "mp_model_backward: no forward has been run");
  MP_CHECK(!m->infer, MP_ERR_STATE, "mp_model_backward: the last forward was run in inference mode (train bit 1 set)");
  hipStream_t st = (hipStream_t)stream;
  const int B = m->B, T = m->cfg.num_frame, J = m->cfg.num_joints, S = m->cfg.num_bones, K = m->rot.K;
  const long Mr = (long)B * T * J, Ms = (long)B * T * S;
  m->buckets_recorded = false;         // set again below, once every bucket event of THIS backward has been enqueued
  use_scratch(m, 0);
  // parameter gradients of the heads / score heads are needed by nobody downstream: with the weight-gradient stream on they run there,
  // at the start of the backward where that stream is idle (own scratch, ordered behind the last writer of dheadout)
  hipStream_t pst = m->wgrad_async ? m->st3 : nullptr;
  { int rz = zero_head_grad_scratch(m, m->rot, st); if (rz) return rz; }
  // decoder
  if (m->has_seg)
    HZ(st, "fk_decode_bwd", HR(m->rot.headout, 4.0 * K * Mr * m->rot.O), HR(m->lengths, 4.0 * B * S), HR(d_poses, 12.0 * B * K * T * J), HW(m->rot.dheadout, 4.0 * K * Mr * m->rot.O),
       HW(m->dlen_pose, 4.0 * B * K * T * S));
  else HZ(st, "dposes_copy (no decoder)", HR(d_poses, 12.0 * Mr), HW(m->rot.dheadout, 12.0 * Mr));
  if (m->has_seg) RUN(PC_OTHER, 0, fk_decode_bwd(m->rot.headout, m->rot.O, m->cfg.rot_rep_dim, m->lengths, d_poses, m->rot.dheadout, m->dlen_pose, B, K, T, st));
  else MP_HIP(hipMemcpyAsync(m->rot.dheadout, d_poses, sizeof(float) * Mr * 3, hipMemcpyDeviceToDevice, st));
  if (m->cfg.arch == 0) {
    ScoreParams sp;
    ScoreGrads sg;
    for (int k = 0; k < K; ++k) {
      sp.w[k] = P(m, fp, m->rot.sw[k]); sp.b[k] = P(m, fp, m->rot.sb[k]);
      sg.w[k] = G(m, fg, m->rot.sw[k]); sg.b[k] = G(m, fg, m->rot.sb[k]);
    }
    // scores are recomputed into tmpC-sized scratch? no: softmax outputs are cheap to recompute from the head output
    float* sc = m->tmpC;    // (B,K,T) scratch, free at this point of the backward
    HZ(st, "scores_fwd.recompute", HR(m->rot.headout, 4.0 * K * Mr * m->rot.O), HW(sc, 4.0 * B * K * T));
    RUN(PC_OTHER, 0, scores_fwd(m->rot.headout, sp, K, m->rot.O, sc, B, T, J, st));
    {
      float* const ssc = pst ? m->sc_dlogit : m->small;
      const double sscb = 4.0 * (pst ? scores_bwd_scratch_floats(K, B, T) : m->small_floats);
      HZ(st, "scores_bwd", HR(m->rot.headout, 4.0 * K * Mr * m->rot.O), HR(sc, 4.0 * B * K * T), HW(m->rot.dheadout, 4.0 * K * Mr * m->rot.O), HW(ssc, sscb));
      const hipStream_t sp_st = hz_param_edge(m, st, pst, pst ? m->ev_heads : nullptr);
      HZ(sp_st, "scores_bwd.params", HR(m->rot.headout, 4.0 * K * Mr * m->rot.O), HW(ssc, sscb), HW(G(m, fg, m->rot.hg[0]), 4.0 * head_param_floats(m, m->rot)));
    }
    RUN(PC_OTHER, 0, scores_bwd(m->rot.headout, sc, d_scores ? d_scores : m->dscore_zero, sp, sg, K, m->rot.O, m->rot.dheadout, B, T, J,
                                pst ? m->sc_dlogit : m->small, pst ? scores_bwd_scratch_floats(K, B, T) : m->small_floats, st, pst, pst ? m->ev_heads : nullptr));
  } else if (pst) {
    MP_HIP(ev_record(m, m->ev_heads, st));             // dheadout is final: the head parameter gradients may start on the other stream
    MP_HIP(ev_wait(m, pst, m->ev_heads));
  }
  // fork: the bones-net backward only needs the per-pose length gradients of the decoder backward
  const hipStream_t side = (m->cfg.streams & 1) ? st : m->st2;
  MP_HIP(ev_record(m, m->ev_fork, st));
  MP_HIP(ev_wait(m, side, m->ev_fork));
  use_scratch(m, 0);
  // rotations module
  HeadParams hp;
  HeadGrads hg;
  head_params(m, m->rot, fp, hp);
  head_grads(m, m->rot, fg, hg);
  {
    float* const hsc = pst ? m->hsmall : m->small;
    const double hscb = 4.0 * (pst ? m->hsmall_floats : m->small_floats);
    HZ(st, "heads_bwd.dx", HR(m->rot.x_final, 4.0 * Mr * m->rot.C), HR(m->rot.hstats, 8.0 * Mr), HR(m->rot.hfold, 4.0 * heads_fold_floats(m->rot.C)), HR(m->rot.headout, 4.0 * K * Mr * m->rot.O),
       HR(m->rot.dheadout, 4.0 * K * Mr * m->rot.O), HW(m->g, 4.0 * Mr * m->rot.C), HW(pst ? nullptr : hsc, hscb));
    // (the parameter kernels go to pst WITHOUT an event of their own: they read what the dx kernel reads, all final since ev_heads above)
    HZ(pst ? pst : st, "heads_bwd.params", HR(m->rot.x_final, 4.0 * Mr * m->rot.C), HR(m->rot.hstats, 8.0 * Mr), HR(m->rot.headout, 4.0 * K * Mr * m->rot.O),
       HR(m->rot.dheadout, 4.0 * K * Mr * m->rot.O), HW(hsc, hscb), HW(G(m, fg, m->rot.hg[0]), 4.0 * head_param_floats(m, m->rot)),
       HW(m->rot.hdw, m->rot.hdw ? 4.0 * K * m->rot.O * m->rot.C : 0.0));      // (MuReadout: the weight gradients land in the zeroed scratch, flushed below)
  }
  if (heads_use_mfma(K, m->rot.O, m->rot.C))
    RUN(PC_OTHER, 0, heads_bwd_mfma(m->rot.x_final, m->rot.hstats, m->rot.hfold, m->rot.headout, hp, hg, K, m->rot.O, m->rot.dheadout, m->g, (int)Mr, m->rot.C,
                                    pst ? m->hsmall : m->small, pst ? m->hsmall_floats : m->small_floats, st, pst));
  else
    RUN(PC_OTHER, 0, heads_bwd(m->rot.x_final, m->rot.hstats, hp, hg, K, m->rot.O, m->rot.dheadout, m->g, (int)Mr, m->rot.C,
                               pst ? m->hsmall : m->small, pst ? m->hsmall_floats : m->small_floats, st, pst));
  { int rf = flush_head_grads(m, m->rot, fg, pst ? pst : st); if (rf) return rf; }
  // f16_backward: this backward's gradient scale S, chosen on the device (no host round trip) from the residual gradient the backbone
  // starts from - NOT from d_poses: the WTA loss gradient is a unit vector per joint whatever the error, so its maximum is a constant of the
  // batch shape while the interior gradients shrank by four orders of magnitude over 200 optimisation steps (measured, round 4: 38 % of
  // dz's true values would have been fp16-subnormal at an S taken from d_poses).  One read of g (M x C floats, ~0.13 ms at full size).
  if (m->rot.f8g) HZ(st, "grad_scale", HR(m->g, 4.0 * Mr * m->rot.C), HW(m->gsc, 32));
  if (m->rot.f8g) RUN(PC_OTHER, 0, grad_scale(m->g, Mr * m->rot.C, nullptr, 0, m->gsc, st));
  int rc = backbone_bwd(m, m->rot, fp, fg, B, st);
  if (rc) return rc;
  HZ(st, "embed_bwd", HR(m->g, 4.0 * Mr * m->rot.C), HW(m->small, 4.0 * m->small_floats), HW(G(m, fg, m->rot.emb_w), 4.0 * m->rot.C * 2), HW(G(m, fg, m->rot.spos), 4.0 * J * m->rot.C));
  RUN(PC_OTHER, 0, embed_bwd(m->g, m->x_in, G(m, fg, m->rot.emb_w), G(m, fg, m->rot.emb_b), G(m, fg, m->rot.spos), (int)Mr, m->rot.C, J,
                             m->small, m->small_floats, st));
  // segments module, on the side stream with its own scratch set
  if (m->has_seg) {
    hipStream_t main_st = st;
    st = side;
    use_scratch(m, 1);
    HZ(st, "bones_mean_bwd", HR(m->dlen_pose, 4.0 * B * K * T * S), HW(m->seg.dheadout, 4.0 * Ms));
    RUN(PC_OTHER, 0, bones_mean_bwd(m->dlen_pose, K * T, nullptr, m->seg.dheadout, B, T, S, st));
    HeadParams hs;
    HeadGrads hgs;
    head_params(m, m->seg, fp, hs);
    head_grads(m, m->seg, fg, hgs);
    { int rz = zero_head_grad_scratch(m, m->seg, st); if (rz) return rz; }
    HZ(st, "heads_bwd.seg", HR(m->seg.x_final, 4.0 * Ms * m->seg.C), HR(m->seg.hstats, 8.0 * Ms), HR(m->seg.headout, 4.0 * Ms), HR(m->seg.dheadout, 4.0 * Ms), HW(m->g, 4.0 * Ms * m->seg.C),
       HW(m->small, 4.0 * m->small_floats), HW(hgs.W[0], 4.0 * m->seg.C), HW(hgs.gamma[0], 4.0 * m->seg.C), HW(hgs.beta[0], 4.0 * m->seg.C));
    if (heads_use_mfma(1, 1, m->seg.C))
      RUN(PC_OTHER, 0, heads_bwd_mfma(m->seg.x_final, m->seg.hstats, m->seg.hfold, m->seg.headout, hs, hgs, 1, 1, m->seg.dheadout, m->g, (int)Ms,
                                      m->seg.C, m->small, m->small_floats, st, nullptr));
    else
      RUN(PC_OTHER, 0, heads_bwd(m->seg.x_final, m->seg.hstats, hs, hgs, 1, 1, m->seg.dheadout, m->g, (int)Ms, m->seg.C, m->small,
                                 m->small_floats, st));
    { int rf = flush_head_grads(m, m->seg, fg, st); if (rf) return rf; }
    rc = backbone_bwd(m, m->seg, fp, fg, B, st);
    if (rc) return rc;
    HZ(st, "bones_embed_bwd", HR(m->g, 4.0 * Ms * m->seg.C), HW(m->small, 4.0 * m->small_floats), HW(G(m, fg, m->seg.emb_w), 4.0 * S * m->seg.C * J * 2), HW(G(m, fg, m->seg.spos), 4.0 * S * m->seg.C));
    RUN(PC_OTHER, 0, bones_embed_bwd(m->g, m->x_in, G(m, fg, m->seg.emb_w), G(m, fg, m->seg.emb_b), G(m, fg, m->seg.spos), B * T, J * 2,
                                     S * m->seg.C, m->small, m->small_floats, st));
    MP_HIP(ev_record(m, m->ev_join, side));
    st = main_st;
    use_scratch(m, 0);
    MP_HIP(ev_wait(m, st, m->ev_join));
  }
  if (pst) {                                              // everything queued on the weight-gradient stream belongs to this backward
    MP_HIP(ev_record(m, m->ev_heads, pst));
    MP_HIP(ev_wait(m, st, m->ev_heads));
  }
  m->buckets_recorded = true;
  return MP_OK;
}

static int hazard_report(const HazardTracker* hz, int64_t* out4, char* msg, int msg_cap) {
  if (out4) { out4[0] = hz->launches(); out4[1] = hz->ordered_pairs(); out4[2] = hz->violations(); out4[3] = hz->events(); }
  if (msg && msg_cap > 0) {
    std::string t;
    for (const std::string& l : hz->messages()) { t += l; t += '\n'; }
    snprintf(msg, (size_t)msg_cap, "%s", t.c_str());
  }
  return MP_OK;
}
int mp_model_hazard_report(const mp_model* m, int64_t* out4, char* msg, int msg_cap) {
  MP_CHECK(m != nullptr, MP_ERR_ARG, "mp_model_hazard_report: null model");
  MP_CHECK(m->hz != nullptr, MP_ERR_STATE, "mp_model_hazard_report: the model was created without mp_model_config::debug bit 0");
  return hazard_report(m->hz.get(), out4, msg, msg_cap);
}
struct mp_hazard { HazardTracker t; };
mp_hazard* mp_hazard_create(void) { return new mp_hazard(); }
void mp_hazard_destroy(mp_hazard* h) { delete h; }
int mp_hazard_launch(mp_hazard* h, int stream, const char* name, int n, const int64_t* addr, const int64_t* bytes, const int* is_write) {
  MP_CHECK(h && stream >= 0 && stream < HazardTracker::MAXS && n >= 0 && (n == 0 || (addr && bytes && is_write)), MP_ERR_ARG, "mp_hazard_launch: bad argument");
  std::vector<HzAccess> acc;
  for (int i = 0; i < n; ++i) acc.push_back(HzAccess{(const void*)(uintptr_t)addr[i], (size_t)bytes[i], is_write[i] != 0});
  // (streams are named by small integers here: id i is the i-th distinct handle, so register them in order)
  for (int i = 0; i <= stream; ++i) (void)h->t.stream_id((const void*)(uintptr_t)(i + 1));
  h->t.launch(stream, name, acc.data(), n);
  return MP_OK;
}
int mp_hazard_record(mp_hazard* h, int event, int stream) {
  MP_CHECK(h && stream >= 0 && stream < HazardTracker::MAXS, MP_ERR_ARG, "mp_hazard_record: bad argument");
  for (int i = 0; i <= stream; ++i) (void)h->t.stream_id((const void*)(uintptr_t)(i + 1));
  h->t.record((const void*)(uintptr_t)(event + 1), stream);
  return MP_OK;
}
int mp_hazard_wait(mp_hazard* h, int stream, int event) {
  MP_CHECK(h && stream >= 0 && stream < HazardTracker::MAXS, MP_ERR_ARG, "mp_hazard_wait: bad argument");
  for (int i = 0; i <= stream; ++i) (void)h->t.stream_id((const void*)(uintptr_t)(i + 1));
  h->t.wait(stream, (const void*)(uintptr_t)(event + 1));
  return MP_OK;
}
int mp_hazard_report(const mp_hazard* h, int64_t* out4, char* msg, int msg_cap) {
  MP_CHECK(h != nullptr, MP_ERR_ARG, "mp_hazard_report: null tracker");
  return hazard_report(&h->t, out4, msg, msg_cap);
}

int mp_model_set_streams(mp_model* m, int streams) {
  MP_CHECK(m != nullptr && streams >= 0 && streams <= 3, MP_ERR_ARG, "mp_model_set_streams: streams %d (bit set 0..3)", streams);
  MP_CHECK(m->cfg.max_batch > 0, MP_ERR_STATE, "mp_model_set_streams: a layout-only handle has no streams");
  MP_HIP(hipStreamSynchronize(m->st2));
  MP_HIP(hipStreamSynchronize(m->st3));
  m->cfg.streams = streams;
  m->wgrad_async = (streams & 2) == 0;
  return MP_OK;
}

// {S, 1 / S, scratch, 1, clamped (u32), non-finite (u32)} of mp_model::gsc -> {S, clamped, non-finite, 1 / S} as floats
static void grad_health_values(const float* h, float* out) {
  unsigned c[2];
  memcpy(c, h + 4, sizeof(c));
  out[0] = h[0]; out[1] = (float)c[0]; out[2] = (float)c[1]; out[3] = h[1];
}
int mp_model_grad_health(mp_model* m, float* out, void* stream) {
  MP_CHECK(m && out, MP_ERR_ARG, "mp_model_grad_health: null argument");
  out[0] = out[1] = out[2] = out[3] = 0.f;
  if (m->gsc == nullptr || !m->rot.f8g) return MP_OK;
  MP_CHECK(m->buckets_recorded, MP_ERR_STATE, "mp_model_grad_health: no completed mp_model_backward since the last mp_model_forward");
  float h[8];
  MP_HIP(hipMemcpyAsync(h, m->gsc, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
  MP_HIP(hipStreamSynchronize((hipStream_t)stream));
  grad_health_values(h, out);
  return MP_OK;
}
int mp_model_grad_health_async(mp_model* m, float* out, void* stream) {
  MP_CHECK(m && out, MP_ERR_ARG, "mp_model_grad_health_async: null argument");
  if (m->gsc == nullptr || !m->rot.f8g) {
    MP_HIP(hipMemsetAsync(out, 0, 4 * sizeof(float), (hipStream_t)stream));
    return MP_OK;
  }
  MP_CHECK(m->buckets_recorded, MP_ERR_STATE, "mp_model_grad_health_async: no completed mp_model_backward since the last mp_model_forward");
  return grad_health_pack(m->gsc, out, (hipStream_t)stream);
}

// (the bucket LAYOUT is a property of the parameter layout and also answered by a layout-only handle; waiting needs a device model, whose
// ev_bucket holds exactly rot.depth events - mp_model_grad_bucket_wait checks the index against it)
int mp_model_grad_bucket_count(const mp_model* m) { return m ? m->rot.depth : 0; }

int mp_model_grad_bucket_info(const mp_model* m, int index, int64_t* offset, int64_t* numel) {
  MP_CHECK(m && index >= 0 && index < m->rot.depth, MP_ERR_ARG, "mp_model_grad_bucket_info: bad index %d", index);
  // layer i of the rotations net = blocks 2 i (STE_i) and 2 i + 1 (TTE_i): norm1.weight of the first ... fc2.bias of the second
  const BlockP& a = m->rot.bp[2 * index];
  const BlockP& b = m->rot.bp[2 * index + 1];
  const ParamDesc& first = m->params[a.n1w];
  const ParamDesc& last = m->params[b.f2b];
  if (offset) *offset = first.offset;
  if (numel) *numel = last.offset + (last.numel + 63) / 64 * 64 - first.offset;
  return MP_OK;
}

int mp_model_grad_bucket_wait(mp_model* m, int index, void* stream) {
  MP_CHECK(m && index >= 0 && index < (int)m->ev_bucket.size(), MP_ERR_ARG, "mp_model_grad_bucket_wait: bad index %d (a layout-only handle has no events)", index);
  MP_CHECK(m->buckets_recorded, MP_ERR_STATE, "mp_model_grad_bucket_wait: no completed mp_model_backward since the last mp_model_forward");
  MP_HIP(ev_wait(m, (hipStream_t)stream, m->ev_bucket[index]));
  return MP_OK;
}

int mp_model_peek(const mp_model* m, int which, const float** ptr, int64_t* numel) {
  MP_CHECK(m && ptr && numel && m->B >= 1, MP_ERR_ARG, "mp_model_peek: bad argument or no forward yet");
  const long Mr = (long)m->B * m->cfg.num_frame * m->cfg.num_joints;
  if (which == 0) { *ptr = m->rot.headout; *numel = (long)m->rot.K * Mr * m->rot.O; return MP_OK; }
  if (which == 1 && m->has_seg) { *ptr = m->lengths; *numel = (long)m->B * m->cfg.num_bones; return MP_OK; }
  if (which == 2 && m->train) { *ptr = m->maskbuf; *numel = mp_model_mask_floats(m, m->B); return MP_OK; }   // DropPath multipliers of the last train-mode forward
  // fp32 residual stream of the backbones, block by block (bisecting a discrepancy to a layer): 100 + 2 l = after the attention branch of
  // block l of the rotations net (x_mid), 101 + 2 l = after its MLP branch (x_out); 300 + ... the same for the segments net;
  // 99 / 299 = the embedding output when it is materialised.  Blocks in execution order STE0, TTE0, STE1, ...
  if (which >= 99 && which < 500) {
    const bool seg = which >= 299;
    MP_CHECK(!seg || m->has_seg, MP_ERR_ARG, "mp_model_peek: no segments net");
    const Module& md = seg ? m->seg : m->rot;
    const long M = (long)m->B * m->cfg.num_frame * md.N;
    const int k = which - (seg ? 300 : 100);
    if (k == -1) { *ptr = md.ws[0].x_in; *numel = M * md.C; return MP_OK; }
    MP_CHECK(k >= 0 && k / 2 < (int)md.ws.size(), MP_ERR_ARG, "mp_model_peek: block %d of %d", k / 2, (int)md.ws.size());
    *ptr = (k & 1) ? md.ws[k / 2].x_out : md.ws[k / 2].x_mid;
    *numel = M * md.C;
    return MP_OK;
  }
  // 2-byte gradient operands of the LAST block the backward differentiated (STE0 of the rotations net), as the backward left them in its scratch:
  // 500 = dz (M x 2C), 501 = dqkv (M x 3C) - bf16, or scaled fp16 in an f16_backward model; 502 = the 2-byte copy of the residual gradient
  // (M x C) as its LAST writer left it: the attention-branch copy, which feeds the proj layer and is always bf16.  numel counts FLOATS (two
  // elements each).  For tests that look at where these values sit in the fp16 range.
  if (which >= 500 && which <= 502) {
    const long MC = Mr * m->rot.C;
    const mp_model::ScratchSet& sc = m->sets[0];
    *ptr = which == 500 ? (const float*)sc.tmp2C : (which == 501 ? (const float*)sc.tmp3C : (const float*)sc.g_b16);
    MP_CHECK(m->cfg.precision >= 1 && *ptr != nullptr, MP_ERR_ARG, "mp_model_peek: %d needs a bf16 / bf16x3 model", which);
    *numel = (which == 500 ? 2 : (which == 501 ? 3 : 1)) * MC / 2;
    return MP_OK;
  }
  MP_CHECK(false, MP_ERR_ARG, "mp_model_peek: which=%d", which);
}

int mp_model_peek_copy(const mp_model* m, int which, float* dst, int64_t numel, void* stream) {
  const float* src = nullptr;
  int64_t n = 0;
  int rc = mp_model_peek(m, which, &src, &n);
  if (rc) return rc;
  MP_CHECK(dst && numel == n, MP_ERR_ARG, "mp_model_peek_copy: expected %ld floats", (long)n);
  MP_HIP(hipMemcpyAsync(dst, src, sizeof(float) * n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MP_OK;
}

int mp_prof_enable(mp_model* m, int on) {
  MP_CHECK(m, MP_ERR_ARG, "mp_prof_enable: null model");
  if (on && m->ev.empty()) {
    // two events per profiled launch, ~480 launches per training step: room for ~130 steps between two collects (16384 events - 17 steps - silently
    // truncated the per-class sums of a 30-step timed region until round 6: averages per launch were right, sums per step were not)
    m->ev.resize(131072);
    for (auto& e : m->ev) MP_HIP(hipEventCreate(&e));
  }
  m->prof = on != 0;
  m->ev_used = 0;
  m->ev_cls.clear();
  m->ev_flops.clear();
  m->ev_bytes.clear();
  m->ev_mflops.clear();
  m->ev_tag.clear();
  m->ev_kind.clear();
  return MP_OK;
}

int mp_prof_collect(mp_model* m, double* ms, int64_t* launches, double* flops, double* bytes, double* model_flops) {
  MP_CHECK(m && ms && launches && flops, MP_ERR_ARG, "mp_prof_collect: null argument");
  for (int c = 0; c < MP_PROF_CLASSES; ++c) { ms[c] = 0; launches[c] = 0; flops[c] = 0; if (bytes) bytes[c] = 0; if (model_flops) model_flops[c] = 0; }
  for (int k = 0; k < MP_PROF_KINDS; ++k) { m->kind_ms[k] = m->kind_flops[k] = m->kind_bytes[k] = m->kind_mflops[k] = 0.0; m->kind_launches[k] = m->kind_persist[k] = 0; }
  for (size_t i = 0; i < m->ev_cls.size() && 2 * i + 1 < m->ev_used; ++i) {
    MP_HIP(hipEventSynchronize(m->ev[2 * i + 1]));
    float t = 0.f;
    MP_HIP(hipEventElapsedTime(&t, m->ev[2 * i], m->ev[2 * i + 1]));
    const int c = m->ev_cls[i];
    ms[c] += t;
    launches[c] += 1;
    flops[c] += m->ev_flops[i];
    if (bytes) bytes[c] += m->ev_bytes[i];
    if (model_flops) model_flops[c] += m->ev_mflops[i];
    const int kd = i < m->ev_kind.size() ? m->ev_kind[i] : -1;
    if (kd >= 0 && kd < MP_PROF_KINDS) {
      m->kind_ms[kd] += t; m->kind_launches[kd] += 1; m->kind_flops[kd] += m->ev_flops[i]; m->kind_bytes[kd] += m->ev_bytes[i];
      m->kind_mflops[kd] += m->ev_mflops[i];
      if (i < m->ev_tag.size() && m->ev_tag[i]) m->kind_persist[kd] += 1;
    }
    if (i < m->ev_tag.size() && m->ev_tag[i]) {         // class 6: the launches of classes 0/1 that ran gemm_bf16_persist_kernel
      ms[6] += t;
      launches[6] += 1;
      flops[6] += m->ev_flops[i];
      if (bytes) bytes[6] += m->ev_bytes[i];
      if (model_flops) model_flops[6] += m->ev_mflops[i];
    }
  }
  m->ev_used = 0;
  m->ev_cls.clear();
  m->ev_flops.clear();
  m->ev_bytes.clear();
  m->ev_mflops.clear();
  m->ev_tag.clear();
  m->ev_kind.clear();
  return MP_OK;
}

int mp_prof_kinds(const mp_model* m, double* ms, int64_t* launches, int64_t* persist_launches, double* flops, double* bytes, double* model_flops) {
  MP_CHECK(m && ms && launches, MP_ERR_ARG, "mp_prof_kinds: null argument");
  for (int k = 0; k < MP_PROF_KINDS; ++k) {
    ms[k] = m->kind_ms[k]; launches[k] = m->kind_launches[k];
    if (persist_launches) persist_launches[k] = m->kind_persist[k];
    if (flops) flops[k] = m->kind_flops[k];
    if (bytes) bytes[k] = m->kind_bytes[k];
    if (model_flops) model_flops[k] = m->kind_mflops[k];
  }
  return MP_OK;
}

}  // extern "C"
