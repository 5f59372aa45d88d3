// Host side of libamber_mpn.so: parameter layout, workspace carving, launch sequencing, C ABI.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "mpn_internal.h"

namespace amber {
namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}
int fail_hip(hipError_t e, const char* what) {
  return fail(AMBER_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}
#define HIP_TRY(expr, what)                          \
  do {                                               \
    hipError_t _e = (expr);                          \
    if (_e != hipSuccess) return fail_hip(_e, what); \
  } while (0)

inline size_t align256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------------
// parameter layout (offsets in floats into `params` and `packed`), see include/amber_mpn.h
// ---------------------------------------------------------------------------------------------------
struct MlpLayout {
  int64_t W[kMaxLayers];   // params offset of layer weight
  int64_t B[kMaxLayers];   // params offset of layer bias
  int64_t P[kMaxLayers];   // params offset of post-norm parameter 0 (decoder) or -1
  int64_t Q[kMaxLayers];
  int64_t PW[kMaxLayers][3];  // packed offsets: layer 0 may have up to 3 column blocks, layers >= 1 use [0]
  int32_t layers;
  int32_t in_cols;            // columns of layer 0
};
struct Layout {
  MlpLayout node_embed, edge_embed;
  std::vector<MlpLayout> edge, node;
  std::vector<int64_t> eg, eb, ng, nb;  // LayerNorm params per step (or -1)
  MlpLayout dec;
  int64_t dec_f2[kMaxLayers];  // packed offsets of the decoder matrices in the split-precision format, or -1
  int64_t node_fold;           // packed offset of the folded node tables [3][F_n + 1][L] (embed_proj_kernel), or -1
  int64_t readout_w, readout_b;
  int64_t n_params, n_packed;
};

int validate(const AmberMpnDesc* d) {
  (void)hipGetLastError();  // drop a stale error left by an earlier, unrelated runtime call of this thread
  if (!d) return fail(AMBER_ERR_INVALID_ARGUMENT, "desc is NULL");
  if (d->latent != 32 && d->latent != 64 && d->latent != 128 && d->latent != 256)
    return fail(AMBER_ERR_UNSUPPORTED, "latent_dimension %d not supported (32, 64, 128, 256)", d->latent);
  if (d->num_steps < 0 || d->num_step_repeats < 1) return fail(AMBER_ERR_INVALID_ARGUMENT, "bad num_steps / num_step_repeats");
  if (d->mlp_layers < 1 || d->mlp_layers > kMaxLayers)
    return fail(AMBER_ERR_UNSUPPORTED, "stack.mlp.num_layers %d not in 1..%d", d->mlp_layers, kMaxLayers);
  if (d->activation < AMBER_ACT_RELU || d->activation > AMBER_ACT_TANH)
    return fail(AMBER_ERR_UNSUPPORTED, "LatentMLP activation id %d (relu, leakyrelu, tanh only)", d->activation);
  if (d->residual < 0 || d->residual > 2 || d->layer_norm < 0 || d->layer_norm > 2)
    return fail(AMBER_ERR_INVALID_ARGUMENT, "bad residual / layer_norm mode");
  if (d->in_node_features < 0 || d->in_node_features > 64 || d->in_edge_features < 0 || d->in_edge_features > 64)
    return fail(AMBER_ERR_UNSUPPORTED, "raw feature widths must be in 0..64");
  if (d->embed_layers < 0 || d->embed_layers > kMaxLayers) return fail(AMBER_ERR_UNSUPPORTED, "embed_layers out of range");
  if (d->embed_layers > 0 && (d->embed_activation < AMBER_ACT_RELU || d->embed_activation > AMBER_ACT_TANH))
    return fail(AMBER_ERR_UNSUPPORTED, "embedding MLP activation id %d", d->embed_activation);
  if (d->has_head) {
    if (d->decoder_layers < 0 || d->decoder_layers > kMaxLayers) return fail(AMBER_ERR_UNSUPPORTED, "decoder_layers out of range");
    if (d->decoder_activation < AMBER_ACT_RELU || d->decoder_activation > AMBER_ACT_GELU)
      return fail(AMBER_ERR_UNSUPPORTED, "decoder activation id %d", d->decoder_activation);
    if (d->decoder_norm < 0 || d->decoder_norm > 2) return fail(AMBER_ERR_INVALID_ARGUMENT, "bad decoder_norm");
  }
  if (d->precision != AMBER_PREC_FP32 && d->precision != AMBER_PREC_F16X2 && d->precision != AMBER_PREC_F16)
    return fail(AMBER_ERR_UNSUPPORTED, "precision id %d", d->precision);
  if (!(d->ln_eps > 0.f)) return fail(AMBER_ERR_INVALID_ARGUMENT, "ln_eps must be > 0");
  return AMBER_OK;
}

bool use_fast_head(const AmberMpnDesc& d);
bool use_node_fold(const AmberMpnDesc& d);
bool use_b3_forward(const AmberMpnDesc& d);

Layout build_layout(const AmberMpnDesc& d) {
  Layout lay;
  const int64_t L = d.latent;
  int64_t po = 0, pk = 0;
  // one packed matrix: L*L floats (exact fp32 tiles, hi/lo fp16 pairs) or 3 L*L bf16 pieces, + header
  const int64_t mat_floats = (use_b3_forward(d) ? L * L * 3 / 2 : L * L) + kPackedHeader;
  auto mlp = [&](MlpLayout& m, int layers, int in_cols, int col_blocks, bool norm, bool pack_first) {
    m.layers = layers;
    m.in_cols = in_cols;
    for (int l = 0; l < kMaxLayers; ++l) {
      m.W[l] = m.B[l] = m.P[l] = m.Q[l] = -1;
      m.PW[l][0] = m.PW[l][1] = m.PW[l][2] = -1;
    }
    for (int l = 0; l < layers; ++l) {
      const int64_t cols = l == 0 ? in_cols : L;
      m.W[l] = po;
      po += L * cols;
      m.B[l] = po;
      po += L;
      if (norm) {
        m.P[l] = po;
        po += L;
        m.Q[l] = po;
        po += L;
      }
      if (l == 0) {
        if (pack_first)
          for (int cb = 0; cb < col_blocks; ++cb) {
            m.PW[0][cb] = pk;
            pk += mat_floats;
          }
      } else {
        m.PW[l][0] = pk;
        pk += mat_floats;
      }
    }
  };
  auto embed = [&](MlpLayout& m, int F) {
    if (F == 0) {  // EmptyLinear: bias only
      m.layers = 0;
      m.in_cols = 0;
      for (int l = 0; l < kMaxLayers; ++l) m.W[l] = m.B[l] = m.P[l] = m.Q[l] = m.PW[l][0] = m.PW[l][1] = m.PW[l][2] = -1;
      m.B[0] = po;
      po += L;
    } else {
      mlp(m, d.embed_layers > 0 ? d.embed_layers : 1, F, 0, false, false);
    }
  };
  embed(lay.node_embed, d.in_node_features);
  embed(lay.edge_embed, d.in_edge_features);
  lay.edge.resize(d.num_steps);
  lay.node.resize(d.num_steps);
  lay.eg.assign(d.num_steps, -1);
  lay.eb.assign(d.num_steps, -1);
  lay.ng.assign(d.num_steps, -1);
  lay.nb.assign(d.num_steps, -1);
  for (int s = 0; s < d.num_steps; ++s) {
    mlp(lay.edge[s], d.mlp_layers, 3 * d.latent, 3, false, true);
    mlp(lay.node[s], d.mlp_layers, 2 * d.latent, 2, false, true);
    if (d.layer_norm != AMBER_MODE_NONE) {
      lay.eg[s] = po;
      po += L;
      lay.eb[s] = po;
      po += L;
      lay.ng[s] = po;
      po += L;
      lay.nb[s] = po;
      po += L;
    }
  }
  lay.dec.layers = 0;
  for (int l = 0; l < kMaxLayers; ++l) lay.dec_f2[l] = -1;
  lay.readout_w = lay.readout_b = -1;
  if (d.has_head) {
    mlp(lay.dec, d.decoder_layers, d.latent, 1, d.decoder_norm != AMBER_NORM_NONE, true);
    if (use_fast_head(d))
      for (int l = 0; l < d.decoder_layers; ++l) {
        lay.dec_f2[l] = pk;
        pk += L * L + kPackedHeader;
      }
    lay.readout_w = po;
    po += L;
    lay.readout_b = po;
    po += 1;
  }
  lay.node_fold = -1;
  if (use_node_fold(d)) {
    lay.node_fold = pk;
    pk += (3 * (d.in_node_features + 1) * L + 63) / 64 * 64;
  }
  lay.n_params = po;
  lay.n_packed = pk;
  return lay;
}

// ---------------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------------
struct Workspace {
  float *x, *e, *Ps, *Pd, *agg, *head, *tail;
  size_t bytes;
};
Workspace carve(const AmberMpnDesc& d, int64_t N, int64_t E, void* base) {
  const size_t L = static_cast<size_t>(d.latent);
  const size_t T = static_cast<size_t>(cdiv(E, 16));  // finest reduction tile (edge_kernel_h); edge_kernel uses every other slot
  const size_t nb = align256(sizeof(float) * static_cast<size_t>(N) * L);
  const size_t ebytes = align256(sizeof(float) * static_cast<size_t>(E) * L);
  const size_t tb = align256(sizeof(float) * T * L);
  char* p = static_cast<char*>(base);
  Workspace w;
  size_t off = 0;
  auto take = [&](size_t n) {
    float* r = reinterpret_cast<float*>(p + off);
    off += n;
    return r;
  };
  w.x = take(nb);
  w.Ps = take(nb);
  w.Pd = take(nb);
  w.agg = take(nb);
  w.head = take(tb);
  w.tail = take(tb);
  w.e = take(ebytes);
  w.bytes = off;
  return w;
}

// training tape: block inputs of every step, x_t[N,L] and e_t[E,L] (CSR order), t = 0 .. iters, and the mean aggregate
// m_t[N,L] of every step as the forward's node kernel formed it (with an "outer" residual / LayerNorm the node module
// aggregates the edge module's output BEFORE them, which is not the e_{t+1} of the tape)
struct Tape {
  float *x, *e, *m;
  size_t bytes;
};
bool has_outer_mode(const AmberMpnDesc& d) { return d.residual == AMBER_MODE_OUTER || d.layer_norm == AMBER_MODE_OUTER; }
Tape carve_tape(const AmberMpnDesc& d, int64_t N, int64_t E, float* base) {
  const size_t L = static_cast<size_t>(d.latent), T1 = static_cast<size_t>(d.num_steps) * d.num_step_repeats + 1;
  Tape t;
  t.x = base;
  const size_t xb = align256(sizeof(float) * T1 * static_cast<size_t>(N) * L);
  t.e = reinterpret_cast<float*>(reinterpret_cast<char*>(base) + xb);
  t.bytes = xb + align256(sizeof(float) * T1 * static_cast<size_t>(E) * L);
  // the mean aggregate of every step: with an "outer" mode it is not a function of e_{t+1} at all, and otherwise keeping what
  // the forward's node kernel used (N rows a step) is cheaper than summing the E rows of e_{t+1} again in the backward
  t.m = reinterpret_cast<float*>(reinterpret_cast<char*>(base) + t.bytes);
  t.bytes += align256(sizeof(float) * (T1 - 1) * static_cast<size_t>(N) * L);
  return t;
}

// ---------------------------------------------------------------------------------------------------
// optional per-kernel timing (bench hook)
// ---------------------------------------------------------------------------------------------------
struct Timer {
  bool on = false;
  hipStream_t s = nullptr;
  struct Rec {
    hipEvent_t a, b;
    int cls, cls2;
  };
  std::vector<Rec> recs;
  void begin(int cls, int cls2 = -1) {  // cls2: a second class the same launch is also booked under
    if (!on) return;
    Rec r;
    r.cls = cls;
    r.cls2 = cls2;
    (void)hipEventCreate(&r.a);
    (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.a, s);
    recs.push_back(r);
  }
  void end() {
    if (!on) return;
    (void)hipEventRecord(recs.back().b, s);
  }
  void collect(float* ms, int32_t* launches) {
    for (int i = 0; i < AMBER_MPN_NUM_KERNEL_CLASSES; ++i) {
      ms[i] = 0.f;
      if (launches) launches[i] = 0;
    }
    (void)hipStreamSynchronize(s);
    for (auto& r : recs) {
      float t = 0.f;
      (void)hipEventElapsedTime(&t, r.a, r.b);
      ms[r.cls] += t;
      if (launches) launches[r.cls] += 1;
      if (r.cls2 >= 0) {
        ms[r.cls2] += t;
        if (launches) launches[r.cls2] += 1;
      }
      (void)hipEventDestroy(r.a);
      (void)hipEventDestroy(r.b);
    }
    recs.clear();
  }
};

// Debug switches of the environment, read ONCE per process: the packed-buffer layout depends on them (use_fast_head), so
// a value that changed between amber_mpn_pack_weights and a forward would make the kernels read the wrong matrices.
struct EnvSwitches {
  bool no_fused_head, no_fused_embed, exact_backward, no_fused_backward;
  EnvSwitches()
      : no_fused_head(getenv("AMBER_MPN_NO_FUSED_HEAD") != nullptr),
        no_fused_embed(getenv("AMBER_MPN_NO_FUSED_EMBED") != nullptr),
        exact_backward(getenv("AMBER_MPN_EXACT_BACKWARD") != nullptr),
        no_fused_backward(getenv("AMBER_MPN_NO_FUSED_BACKWARD") != nullptr) {}
};
const EnvSwitches& env_switches() {
  static const EnvSwitches e;
  return e;
}
// kernel-variant timing switches (tools/exp_dbg.py): only in builds that compile them in, and then read per call
int dbg_switches() {
#ifdef AMBER_DBG_SWITCHES
  const char* dbg = getenv("AMBER_MPN_DBG");
  return dbg ? atoi(dbg) : 0;
#else
  return 0;
#endif
}

// the backward's contractions on bf16x3 operands (mpn_bwd.hip) unless the descriptor asks for exact fp32 arithmetic
bool use_split_backward(const AmberMpnDesc& d) {
  return d.precision != AMBER_PREC_FP32 && bwd_split_supported(d.latent) && !env_switches().exact_backward;
}
// latent 256 outside the exact mode: the row-per-lane kernels on bf16x3 operands (every packed matrix in that format)
bool use_b3_forward(const AmberMpnDesc& d) { return d.precision != AMBER_PREC_FP32 && fwd_b3_supported(d.latent); }
bool use_fast_edge(const AmberMpnDesc& d) {
  return d.precision != AMBER_PREC_FP32 && edge_f2_supported(d.latent, d.mlp_layers, d.activation, d.residual, d.layer_norm);
}
bool use_fast_node(const AmberMpnDesc& d) {
  return d.precision != AMBER_PREC_FP32 && node_f2_supported(d.latent, d.activation, d.residual, d.layer_norm);
}
// the head rides on the last split-precision node launch (decoder layers as further phases, readout in the epilogue)
bool use_fast_head(const AmberMpnDesc& d) {
  return d.has_head && use_fast_node(d) && d.num_steps * d.num_step_repeats > 0 &&
         head_f2_supported(d.decoder_layers, d.decoder_activation, d.decoder_norm) && !env_switches().no_fused_head;
}
// The node embedding and the first step's projections run as ONE pass over the raw node features (embed_proj_kernel) when
// the embedding is linear or constant; tables in the packed buffer.  (Switched off together with the fused edge embedding
// by AMBER_MPN_NO_FUSED_EMBED.)
bool use_node_fold(const AmberMpnDesc& d) {
  return d.num_steps * d.num_step_repeats > 0 && (d.in_node_features == 0 || d.embed_layers == 0) &&
         d.in_node_features <= kMaxFoldNodeFeatures && !env_switches().no_fused_embed;
}
int device_cus() {
  constexpr int kMaxDevices = 64;
  static int cached[kMaxDevices] = {};  // 0 = not asked yet (benign race: every thread stores the same value)
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev >= 0 && dev < kMaxDevices && cached[dev] > 0) return cached[dev];
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
  if (dev >= 0 && dev < kMaxDevices) cached[dev] = cus;
  return cus;
}
hipError_t run_edge(const AmberMpnDesc& d, EdgeArgs& ea, hipStream_t s) {
  if (use_fast_edge(d)) {
    ea.red_shift = 4;  // the split-precision kernel sums per half of a tile: reduction tiles of 16 edges
    ea.single = d.precision == AMBER_PREC_F16 ? 1 : 0;
    return launch_edge_f2(d.latent, ea, device_cus(), s);
  }
  ea.red_shift = 5;
  return launch_edge(d.latent, use_b3_forward(d) ? 1 : 0, ea, s);
}

// Node launch: the split-precision kernel does the update and the projections; a decoder / readout that rides on the
// same launch (the exact dense chain) follows as a launch of its own on the updated rows.
hipError_t run_node(const AmberMpnDesc& d, const NodeArgs& na, hipStream_t s) {
  if (!na.fast) return launch_node(d.latent, use_b3_forward(d) ? 1 : 0, na, s);
  NodeArgs u = na;
  u.dbg = dbg_switches();
  u.single = d.precision == AMBER_PREC_F16 ? 1 : 0;
  // the head joins this launch when its matrices exist in the split-precision format and the update runs here too
  const bool head_here = na.do_readout && na.do_update && !na.dense_out && na.n_dense > 0 && na.dense_f2[0] != nullptr;
  if (!head_here) {
    u.n_dense = 0;
    u.do_readout = 0;
    u.dense_out = nullptr;
  }
  hipError_t e = launch_node_f2(d.latent, u, device_cus(), s);
  if (e != hipSuccess || head_here || (na.n_dense == 0 && !na.do_readout)) return e;
  NodeArgs hd = na;
  hd.fast = 0;
  hd.do_update = 0;
  hd.do_proj = 0;
  hd.x_in = na.do_update ? na.x_out : na.x_in;
  hd.x_out = nullptr;
  return launch_node(d.latent, use_b3_forward(d) ? 1 : 0, hd, s);
}

// ---------------------------------------------------------------------------------------------------
// argument builders
// ---------------------------------------------------------------------------------------------------
void fill_edge_step(EdgeArgs& a, const AmberMpnDesc& d, const Layout& lay, int step, const float* params, const float* packed) {
  const MlpLayout& m = lay.edge[step];
  a.w0e = packed + m.PW[0][2];
  a.nl = d.mlp_layers;
  a.act = d.activation;
  for (int l = 1; l < d.mlp_layers; ++l) a.layers[l] = DenseLayer{packed + m.PW[l][0], params + m.B[l], nullptr, nullptr};
  a.residual = d.residual;
  a.layer_norm = d.layer_norm;
  a.gamma = d.layer_norm != AMBER_MODE_NONE ? params + lay.eg[step] : nullptr;
  a.beta = d.layer_norm != AMBER_MODE_NONE ? params + lay.eb[step] : nullptr;
  a.eps = d.ln_eps;
  a.dbg = dbg_switches();
}
void fill_node_update(NodeArgs& a, const AmberMpnDesc& d, const Layout& lay, int step, const float* params, const float* packed) {
  const MlpLayout& m = lay.node[step];
  a.do_update = 1;
  a.fast = use_fast_node(d) ? 1 : 0;
  a.w0x = packed + m.PW[0][0];
  a.w0a = packed + m.PW[0][1];
  a.b0 = params + m.B[0];
  a.nl = d.mlp_layers;
  a.act = d.activation;
  for (int l = 1; l < d.mlp_layers; ++l) a.layers[l] = DenseLayer{packed + m.PW[l][0], params + m.B[l], nullptr, nullptr};
  a.red_shift = use_fast_edge(d) ? 4 : 5;  // must match the edge kernel that wrote head / tail (run_edge)
  a.residual = d.residual;
  a.layer_norm = d.layer_norm;
  a.gamma = d.layer_norm != AMBER_MODE_NONE ? params + lay.ng[step] : nullptr;
  a.beta = d.layer_norm != AMBER_MODE_NONE ? params + lay.nb[step] : nullptr;
  a.eps = d.ln_eps;
}
void fill_node_proj(NodeArgs& a, const AmberMpnDesc& d, const Layout& lay, int step, const float* params, const float* packed,
                    float* Ps, float* Pd) {
  const MlpLayout& m = lay.edge[step];
  a.do_proj = 1;
  a.fast = use_fast_node(d) ? 1 : 0;
  a.w0s = packed + m.PW[0][0];
  a.w0d = packed + m.PW[0][1];
  a.b0e = params + m.B[0];
  a.Ps = Ps;
  a.Pd = Pd;
}
void fill_node_head(NodeArgs& a, const AmberMpnDesc& d, const Layout& lay, const float* params, const float* packed,
                    const int32_t* out_index, float* out) {
  a.n_dense = d.decoder_layers;
  a.dense_act = d.decoder_activation;
  a.dense_norm = d.decoder_norm;
  for (int l = 0; l < d.decoder_layers; ++l)
  {
    a.dense[l] = DenseLayer{packed + lay.dec.PW[l][0], params + lay.dec.B[l], lay.dec.P[l] >= 0 ? params + lay.dec.P[l] : nullptr,
                            lay.dec.Q[l] >= 0 ? params + lay.dec.Q[l] : nullptr};
    a.dense_f2[l] = lay.dec_f2[l] >= 0 ? packed + lay.dec_f2[l] : nullptr;
  }
  a.do_readout = 1;
  a.wr = params + lay.readout_w;
  a.br = params + lay.readout_b;
  a.out_index = out_index;
  a.out = out;
}

// one Embedding module: linear / constant / MLP, rows optionally gathered through perm; result in `out`
int run_embedding(const AmberMpnDesc& d, const MlpLayout& m, int F, const float* params, const float* packed, const float* in,
                  int64_t rows, const int32_t* perm, float* out, hipStream_t s, Timer& tm) {
  const int L = d.latent;
  tm.begin(AMBER_KC_EMBED);
  hipError_t e;
  if (F == 0)
    e = launch_embed(L, nullptr, rows, 0, nullptr, params + m.B[0], nullptr, -1, out, s);
  else
    e = launch_embed(L, in, rows, F, params + m.W[0], params + m.B[0], perm, d.embed_layers > 0 ? d.embed_activation : -1, out, s);
  tm.end();
  HIP_TRY(e, "embed kernel");
  if (F > 0 && d.embed_layers > 1) {
    NodeArgs a{};
    a.N = rows;
    a.x_in = out;
    a.n_dense = d.embed_layers - 1;
    a.dense_act = d.embed_activation;
    a.dense_norm = AMBER_NORM_NONE;
    for (int l = 1; l < d.embed_layers; ++l) a.dense[l - 1] = DenseLayer{packed + m.PW[l][0], params + m.B[l], nullptr, nullptr};
    a.dense_out = out;  // in place: a wave reads its 32 rows before it writes them
    tm.begin(AMBER_KC_EMBED);
    e = launch_node(L, use_b3_forward(d) ? 1 : 0, a, s);
    tm.end();
    HIP_TRY(e, "embedding MLP kernel");
  }
  return AMBER_OK;
}

int forward_impl(const AmberMpnDesc* desc, const float* params, const float* packed, int64_t N, int64_t E, const float* x,
                 const float* edge_attr, const int32_t* rowptr, const int32_t* src_sorted, const int32_t* dst_sorted,
                 const int32_t* perm, const int32_t* out_index, float* out_node, float* out_edge, float* out_sizing,
                 void* workspace, size_t workspace_bytes, hipStream_t s, Timer& tm, float* tape = nullptr,
                 const PostArgs* post = nullptr) {
  int rc = validate(desc);
  if (rc) return rc;
  const AmberMpnDesc& d = *desc;
  if (N < 0 || E < 0 || N >= (1ll << 31) || E >= (1ll << 31)) return fail(AMBER_ERR_INVALID_ARGUMENT, "N / E out of int32 range");
  if (!params || !packed || !workspace) return fail(AMBER_ERR_INVALID_ARGUMENT, "params / packed / workspace is NULL");
  if (N > 0 && d.in_node_features > 0 && !x) return fail(AMBER_ERR_INVALID_ARGUMENT, "x is NULL");
  if (E > 0 && d.in_edge_features > 0 && !edge_attr) return fail(AMBER_ERR_INVALID_ARGUMENT, "edge_attr is NULL");
  if (E > 0 && (!rowptr || !src_sorted || !dst_sorted || !perm)) return fail(AMBER_ERR_INVALID_ARGUMENT, "CSR arrays are NULL");
  if (N > 0 && !rowptr) return fail(AMBER_ERR_INVALID_ARGUMENT, "rowptr is NULL");
  if (out_sizing && !d.has_head) return fail(AMBER_ERR_INVALID_ARGUMENT, "out_sizing requested but desc->has_head == 0");
  if (workspace_bytes < amber_mpn_workspace_bytes(desc, N, E))
    return fail(AMBER_ERR_WORKSPACE_TOO_SMALL, "workspace %zu < %zu bytes", workspace_bytes, amber_mpn_workspace_bytes(desc, N, E));
  if (N == 0) return AMBER_OK;  // an edge needs nodes: nothing to compute
  const Layout lay = build_layout(d);
  const Workspace w = carve(d, N, E, workspace);
  const int L = d.latent;
  const int iters = d.num_steps * d.num_step_repeats;
  // Training: the block inputs of every step are kept (x_t, e_t, t = 0 .. iters) instead of being updated in place.
  const Tape tp = carve_tape(d, N, E, tape);
  auto xbuf = [&](int t) { return tape ? tp.x + static_cast<size_t>(t) * N * L : w.x; };
  auto ebuf = [&](int t) { return tape ? tp.e + static_cast<size_t>(t) * E * L : w.e; };

  const bool fold_node = lay.node_fold >= 0;  // x_0, P_s, P_d of step 0 in one pass over the raw features
  if (fold_node) {
    tm.begin(AMBER_KC_EMBED);
    hipError_t e = launch_embed_proj(L, x, N, d.in_node_features, packed + lay.node_fold, xbuf(0), w.Ps, w.Pd, s);
    tm.end();
    HIP_TRY(e, "node embedding + projection kernel");
  } else {
    rc = run_embedding(d, lay.node_embed, d.in_node_features, params, packed, x, N, nullptr, xbuf(0), s, tm);
    if (rc) return rc;
  }
  // The linear (or constant) edge embedding of few raw features is fused into the first edge step: its rows are
  // computed there on the fly and e is first written by that step.
  const bool fuse_edge_embed = !tape && iters > 0 && use_fast_edge(d) && d.in_edge_features <= kMaxFusedEmbed &&
                               (d.in_edge_features == 0 || d.embed_layers == 0) && !env_switches().no_fused_embed;
  if (!fuse_edge_embed) {
    rc = run_embedding(d, lay.edge_embed, d.in_edge_features, params, packed, edge_attr, E, perm, ebuf(0), s, tm);
    if (rc) return rc;
  }

  const bool want_head = d.has_head && out_sizing;
  if (iters == 0) {
    NodeArgs a{};
    a.N = N;
    a.x_in = xbuf(0);
    a.x_out = out_node;
    if (want_head) {
      fill_node_head(a, d, lay, params, packed, out_index, out_sizing);
      if (post) a.post = *post;
    }
    if (out_node || want_head) {
      tm.begin(AMBER_KC_NODE);
      hipError_t e = launch_node(L, use_b3_forward(d) ? 1 : 0, a, s);
      tm.end();
      HIP_TRY(e, "node kernel");
    }
    if (out_edge) {
      tm.begin(AMBER_KC_OTHER);
      hipError_t e = launch_permute_rows(ebuf(0), perm, E, L, 0, out_edge, s);
      tm.end();
      HIP_TRY(e, "permute kernel");
    }
    return AMBER_OK;
  }
  if (!fold_node) {  // first-layer projections of step 0
    NodeArgs a{};
    a.N = N;
    a.x_in = xbuf(0);
    fill_node_proj(a, d, lay, 0, params, packed, w.Ps, w.Pd);
    tm.begin(AMBER_KC_NODE);
    hipError_t e = run_node(d, a, s);
    tm.end();
    HIP_TRY(e, "node projection kernel");
  }
  for (int it = 0; it < iters; ++it) {
    const int step = it / d.num_step_repeats;
    const bool last = it == iters - 1;
    EdgeArgs ea{};
    ea.E = E;
    ea.N = N;
    ea.src = src_sorted;
    ea.dst = dst_sorted;
    ea.rowptr = rowptr;
    ea.perm = perm;
    ea.e = ebuf(it);
    ea.e_out = tape ? ebuf(it + 1) : nullptr;
    ea.Ps = w.Ps;
    ea.Pd = w.Pd;
    fill_edge_step(ea, d, lay, step, params, packed);
    ea.agg = w.agg;
    ea.head = w.head;
    ea.tail = w.tail;
    if (it == 0 && fuse_edge_embed) {
      ea.emb_on = 1;
      ea.emb_f = d.in_edge_features;
      ea.emb_attr = edge_attr;
      ea.emb_w = d.in_edge_features > 0 ? params + lay.edge_embed.W[0] : nullptr;
      ea.emb_b = params + lay.edge_embed.B[0];
      ea.emb_w0 = params + lay.edge[step].W[0];
    }
    ea.write_e = (last && !tape) ? 0 : 1;  // after the last step only the aggregate (and out_edge, if asked for) is needed
    ea.out_edge = last ? out_edge : nullptr;
    tm.begin(AMBER_KC_EDGE, it == 0 ? AMBER_KC_EDGE_FIRST : -1);
    hipError_t e = run_edge(d, ea, s);
    tm.end();
    HIP_TRY(e, "edge kernel");
    if (tape && tp.m)
      HIP_TRY(launch_partials_mean(w.agg, w.head, w.tail, rowptr, use_fast_edge(d) ? 4 : 5, N, L,
                                   tp.m + static_cast<size_t>(it) * N * L, s),
              "mean aggregate for the tape");

    NodeArgs na{};
    na.N = N;
    na.x_in = xbuf(it);
    na.rowptr = rowptr;
    na.agg = w.agg;
    na.head = w.head;
    na.tail = w.tail;
    fill_node_update(na, d, lay, step, params, packed);
    if (!last) {
      na.x_out = xbuf(it + 1);
      fill_node_proj(na, d, lay, (it + 1) / d.num_step_repeats, params, packed, w.Ps, w.Pd);
    } else {
      na.x_out = tape ? xbuf(it + 1) : out_node;
      if (want_head) {
        fill_node_head(na, d, lay, params, packed, out_index, out_sizing);
        if (post) na.post = *post;
        if (na.fast && !na.x_out && !na.dense_f2[0]) na.x_out = w.x;  // the head runs as a launch of its own on the updated rows
      }
    }
    tm.begin(AMBER_KC_NODE, last ? AMBER_KC_NODE_LAST : -1);
    e = run_node(d, na, s);
    tm.end();
    HIP_TRY(e, "node kernel");
    if (last && tape && out_node)
      HIP_TRY(hipMemcpyAsync(out_node, xbuf(it + 1), sizeof(float) * static_cast<size_t>(N) * L, hipMemcpyDeviceToDevice, s),
              "copy of the final node latents");
  }
  return AMBER_OK;
}

// ---------------------------------------------------------------------------------------------------
// training: configurations the backward kernels take, workspace, and the launch sequence
// ---------------------------------------------------------------------------------------------------
int validate_train(const AmberMpnDesc& d) {
  if (d.has_head) {
    if (d.decoder_norm != AMBER_NORM_NONE && d.decoder_norm != AMBER_NORM_LAYER)
      return fail(AMBER_ERR_UNSUPPORTED, "training: decoder BatchNorm is not built (latent_normalization none / layer are)");
  }
  return AMBER_OK;
}

// decoder activations whose derivative the backward takes from the stored pre-activation (the others: from the output)
bool decoder_needs_pre(const AmberMpnDesc& d) {
  return d.decoder_activation == AMBER_ACT_SILU || d.decoder_activation == AMBER_ACT_MISH || d.decoder_activation == AMBER_ACT_GELU;
}
struct BwdWorkspace {
  float *dX, *dE, *M, *DM, *Ps, *Pd, *dPs, *dPd;
  float* AN[kMaxLayers];
  float* DHN[kMaxLayers];
  float* AE[kMaxLayers];
  float* DHE[kMaxLayers];
  float* APN[kMaxLayers];  // decoder LayerNorm: activations before the norm
  float* PREN[kMaxLayers];  // decoder silu / mish / gelu: pre-activations (their derivative is not a function of the output)
  float *gpart, *cpart, *lnpart, *pnpart, *epart, *mats;
  size_t mat_floats;
  size_t bytes;
};
constexpr int kBwdMats = 22;  // node 0..11 (2 + 2 per layer, plain / transposed), edge 12..19, projections^T 20, 21
BwdWorkspace carve_bwd(const AmberMpnDesc& d, int64_t N, int64_t E, void* base) {
  const size_t L = static_cast<size_t>(d.latent);
  const size_t nb = align256(sizeof(float) * static_cast<size_t>(N) * L);
  const size_t eb = align256(sizeof(float) * static_cast<size_t>(E) * L);
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t n) {
    float* r = reinterpret_cast<float*>(p + off);
    off += n;
    return r;
  };
  BwdWorkspace w;
  w.dX = take(nb);
  w.M = take(nb);
  w.DM = take(nb);
  w.Ps = take(nb);
  w.Pd = take(nb);
  w.dPs = take(nb);
  w.dPd = take(nb);
  // row buffers per layer: the stack's MLPs, the decoder (node rows) and the embedding MLPs' tails (their a_0 sits in the
  // last slot, see backward_impl)
  const int ne = d.mlp_layers > d.embed_layers ? d.mlp_layers : d.embed_layers;
  const int nn = ne > d.decoder_layers ? ne : d.decoder_layers;
  const bool dnorm = d.has_head && d.decoder_norm == AMBER_NORM_LAYER;
  const bool dpre = d.has_head && decoder_needs_pre(d);
  for (int l = 0; l < kMaxLayers; ++l) {
    w.AN[l] = l < nn ? take(nb) : nullptr;
    w.DHN[l] = l < nn ? take(nb) : nullptr;
    w.APN[l] = dnorm && l < d.decoder_layers ? take(nb) : nullptr;
    w.PREN[l] = dpre && l < d.decoder_layers ? take(nb) : nullptr;
  }
  w.dE = take(eb);
  for (int l = 0; l < kMaxLayers; ++l) {
    w.AE[l] = l < ne ? take(eb) : nullptr;
    w.DHE[l] = l < ne ? take(eb) : nullptr;
  }
  // split-K partial products, or the fused backward kernels' per-workgroup partials
  const size_t nparts = kWgradMaxChunks > kFusedMaxProducts * kFusedBwdBlocks ? kWgradMaxChunks : kFusedMaxProducts * kFusedBwdBlocks;
  w.gpart = take(align256(sizeof(float) * nparts * L * L));
  w.cpart = take(align256(sizeof(float) * nparts * L));
  w.lnpart = take(align256(sizeof(float) * kBwdMaxBlocks * 2 * L));
  w.pnpart = dnorm ? take(align256(sizeof(float) * kBwdMaxBlocks * kBwdWavesPerBlock * d.decoder_layers * 2 * L)) : nullptr;
  w.epart = take(align256(sizeof(float) * kWgradMaxChunks * L * 65));
  w.mat_floats = bwd_mat_floats(d.latent);
  w.mats = take(align256(sizeof(float) * kBwdMats * w.mat_floats));
  w.bytes = off;
  return w;
}

int backward_impl(const AmberMpnDesc* desc, const float* params, const float* packed, int64_t N, int64_t E, const float* x,
                  const float* edge_attr, const int32_t* rowptr, const int32_t* src_sorted, const int32_t* dst_sorted,
                  const int32_t* perm, const int32_t* rowptr_src, const int32_t* perm_src, const int32_t* out_index,
                  const float* grad_node, const float* grad_edge, const float* grad_sizing, float* tape, float* grad_params,
                  void* workspace, hipStream_t s) {
  const AmberMpnDesc& d = *desc;
  const Layout lay = build_layout(d);
  const Tape tp = carve_tape(d, N, E, tape);
  const BwdWorkspace w = carve_bwd(d, N, E, workspace);
  const int L = d.latent, nl = d.mlp_layers;
  const int iters = d.num_steps * d.num_step_repeats;
  const size_t NL = static_cast<size_t>(N) * L, EL = static_cast<size_t>(E) * L;
  auto xbuf = [&](int t) { return tp.x + static_cast<size_t>(t) * NL; };
  auto ebuf = [&](int t) { return tp.e + static_cast<size_t>(t) * EL; };
  float* const g = grad_params;
  auto mat = [&](int i) { return w.mats + static_cast<size_t>(i) * w.mat_floats; };
  const int split = use_split_backward(d) ? 1 : 0;
  const int fmt = split ? AMBER_PACK_B3 : AMBER_PACK_F32;

  HIP_TRY(hipMemsetAsync(g, 0, sizeof(float) * static_cast<size_t>(lay.n_params), s), "clear gradients");
  if (N == 0) return AMBER_OK;
  if (grad_node)
    HIP_TRY(hipMemcpyAsync(w.dX, grad_node, sizeof(float) * NL, hipMemcpyDeviceToDevice, s), "grad_node");
  else
    HIP_TRY(hipMemsetAsync(w.dX, 0, sizeof(float) * NL, s), "clear dX");
  if (E > 0) {
    if (grad_edge)
      HIP_TRY(launch_permute_rows(grad_edge, perm, E, L, 1, w.dE, s), "grad_edge to CSR order");
    else
      HIP_TRY(hipMemsetAsync(w.dE, 0, sizeof(float) * EL, s), "clear dE");
  }
  const int gn = chain_bwd_grid(N), ge = chain_bwd_grid(E);  // partial column sums: one per workgroup

  // ---- SupervisedMPN head: readout(decoder(x_T[mask])) ----
  if (d.has_head && grad_sizing) {
    const int nd = d.decoder_layers;
    ChainBwdArgs a{};
    a.mode = AMBER_BWD_HEAD;
    a.split = split;
    a.rows = N;
    a.rin = xbuf(iters);
    a.nl = nd;
    a.act = d.decoder_activation;
    a.residual = AMBER_MODE_NONE;
    a.layer_norm = AMBER_MODE_NONE;
    a.eps = d.ln_eps;
    {
      PackBatch pb{};
      for (int l = 0; l < nd; ++l) {
        pb.job[pb.n++] = PackJob{params + lay.dec.W[l], mat(2 * l), L, 0, 0, fmt};
        pb.job[pb.n++] = PackJob{params + lay.dec.W[l], mat(2 * l + 1), L, 0, 1, fmt};
      }
      HIP_TRY(launch_pack_batch(L, pb, s), "pack decoder matrices");
    }
    for (int l = 0; l < nd; ++l) {
      if (l == 0) {
        a.w_in = mat(0);
        a.wT_in = mat(1);
        a.b0 = params + lay.dec.B[0];
      } else {
        a.layers[l] = DenseLayer{mat(2 * l), params + lay.dec.B[l], nullptr, nullptr};
        a.wT[l] = mat(2 * l + 1);
      }
      a.A[l] = w.AN[l];
      a.DH[l] = w.DHN[l];
      a.PRE[l] = w.PREN[l];  // nullptr unless the activation needs it
    }
    const bool dnorm = d.decoder_norm == AMBER_NORM_LAYER && nd > 0;
    const int gwaves = gn * kBwdWavesPerBlock;
    if (dnorm) {  // mlp.py:86-93: a LayerNorm behind every layer's activation (torch's default eps, as the forward)
      a.eps = 1e-5f;
      for (int l = 0; l < nd; ++l) {
        a.pn_gamma[l] = params + lay.dec.P[l];
        a.pn_beta[l] = params + lay.dec.Q[l];
        a.AP[l] = w.APN[l];
      }
      a.pn_part = w.pnpart;
      HIP_TRY(hipMemsetAsync(w.pnpart, 0, sizeof(float) * static_cast<size_t>(gwaves) * nd * 2 * L, s), "clear decoder LayerNorm partials");
    }
    a.dio = w.dX;
    a.dout = grad_sizing;
    a.out_index = out_index;
    a.wr = params + lay.readout_w;
    a.part = w.lnpart;
    HIP_TRY(launch_chain_bwd(L, a, s), "head backward kernel");
    HIP_TRY(launch_colpart_reduce(w.lnpart, gn, L, g + lay.readout_w, L, g + lay.readout_b, 1, s), "readout gradient");
    if (dnorm)
      for (int l = 0; l < nd; ++l)
        HIP_TRY(launch_colpart_reduce_strided(w.pnpart + static_cast<size_t>(l) * 2 * L, gwaves, static_cast<int64_t>(nd) * 2 * L, L,
                                              g + lay.dec.P[l], g + lay.dec.Q[l], s),
                "decoder LayerNorm gradient");
    for (int l = 0; l < nd; ++l)
      HIP_TRY(launch_wgrad(L, split, w.DHN[l], l == 0 ? xbuf(iters) : w.AN[l - 1], N, g + lay.dec.W[l], L, 0, g + lay.dec.B[l], w.gpart,
                           w.cpart, s),
              "decoder weight gradient");
  }

  // weight gradients of one module: one batched launch (+ one ordered reduction) over the module's matrices
  auto wjob = [](WgradBatch& b, const float* A, const float* B, float* dW, int ld, int col0, float* db) {
    b.job[b.n++] = WgradJob{A, B, dW, db, ld, col0};
  };
  for (int it = iters - 1; it >= 0; --it) {
    const int step = it / d.num_step_repeats;
    const MlpLayout& mn = lay.node[step];
    const MlpLayout& me = lay.edge[step];
    const float* x_in = xbuf(it);
    const float* e_in = ebuf(it);
    const bool ln = d.layer_norm != AMBER_MODE_NONE;
    // ---- the step's matrices in the exact operand order, plain and transposed: one launch ----
    // slots: node W0x 0/1, W0a 2/3, W_l 2+2l / 3+2l;  edge W0e 12/13, W_l 12+2l / 13+2l;  W0s^T 20, W0d^T 21
    {
      PackBatch pb{};
      auto pj = [&](const float* W, int ld, int col0, int slot, int transpose) {
        pb.job[pb.n++] = PackJob{W, mat(slot), ld, col0, transpose, fmt};
      };
      pj(params + mn.W[0], 2 * L, 0, 0, 0);
      pj(params + mn.W[0], 2 * L, 0, 1, 1);
      pj(params + mn.W[0], 2 * L, L, 2, 0);
      pj(params + mn.W[0], 2 * L, L, 3, 1);
      for (int l = 1; l < nl; ++l) {
        pj(params + mn.W[l], L, 0, 2 + 2 * l, 0);
        pj(params + mn.W[l], L, 0, 3 + 2 * l, 1);
      }
      HIP_TRY(launch_pack_batch(L, pb, s), "pack node matrices");
      if (E > 0) {
        pb.n = 0;
        pj(params + me.W[0], 3 * L, 2 * L, 12, 0);
        pj(params + me.W[0], 3 * L, 2 * L, 13, 1);
        for (int l = 1; l < nl; ++l) {
          pj(params + me.W[l], L, 0, 12 + 2 * l, 0);
          pj(params + me.W[l], L, 0, 13 + 2 * l, 1);
        }
        pj(params + me.W[0], 3 * L, 0, 20, 1);
        pj(params + me.W[0], 3 * L, L, 21, 1);
        HIP_TRY(launch_pack_batch(L, pb, s), "pack edge matrices");
      }
    }
    // ---- node module: x' = LN(MLP([x, mean(e')]) + x) ----
    const float* mean_in = w.M;
    if (E > 0)  // the aggregate the forward's node kernel used, kept on the tape
      mean_in = tp.m + static_cast<size_t>(it) * NL;
    else
      HIP_TRY(hipMemsetAsync(w.M, 0, sizeof(float) * NL, s), "clear M");
    {
      ChainBwdArgs a{};
      a.mode = AMBER_BWD_NODE;
      a.split = split;
      a.rows = N;
      a.rin = x_in;
      a.m = mean_in;
      a.nl = nl;
      a.act = d.activation;
      // node rows have one consumer, so "outer" only differs from "inner" for residual outer + LayerNorm inner
      // (y = LN(m) + x); every other combination is one of the inner forms (message_passing_block.py:124-128)
      a.residual = d.residual == AMBER_MODE_OUTER ? (d.layer_norm == AMBER_MODE_INNER ? AMBER_MODE_OUTER : AMBER_MODE_INNER) : d.residual;
      a.layer_norm = ln ? AMBER_MODE_INNER : AMBER_MODE_NONE;
      a.gamma = ln ? params + lay.ng[step] : nullptr;
      a.eps = d.ln_eps;
      a.w_in = mat(0);
      a.wT_in = mat(1);
      a.w_m = mat(2);
      a.wT_m = mat(3);
      a.b0 = params + mn.B[0];
      for (int l = 1; l < nl; ++l) {
        a.layers[l] = DenseLayer{mat(2 + 2 * l), params + mn.B[l], nullptr, nullptr};
        a.wT[l] = mat(3 + 2 * l);
      }
      for (int l = 0; l < nl; ++l) {
        a.A[l] = w.AN[l];
        a.DH[l] = w.DHN[l];
      }
      a.dio = w.dX;
      a.dm_out = w.DM;
      a.part = ln ? w.lnpart : nullptr;
      WgradBatch wb{};
      wb.split = split;
      wb.M = N;
      wjob(wb, w.DHN[0], x_in, g + mn.W[0], 2 * L, 0, g + mn.B[0]);  // dW0x, db0
      wjob(wb, w.DHN[0], mean_in, g + mn.W[0], 2 * L, L, nullptr);   // dW0a
      for (int l = 1; l < nl; ++l) wjob(wb, w.DHN[l], w.AN[l - 1], g + mn.W[l], L, 0, g + mn.B[l]);
      if (node_bwd_fused_supported(L, a) && !env_switches().no_fused_backward) {
        const int gf = edge_bwd_fused_grid(N);
        HIP_TRY(launch_node_bwd_fused(L, a, w.gpart, w.cpart, s), "fused node backward kernel");
        if (ln) HIP_TRY(launch_colpart_reduce(w.lnpart, gf, L, g + lay.ng[step], L, g + lay.nb[step], L, s), "node LayerNorm gradient");
        HIP_TRY(launch_wgrad_reduce(L, wb, w.gpart, w.cpart, gf, s), "node weight gradients (reduction)");
      } else {
        HIP_TRY(launch_chain_bwd(L, a, s), "node backward kernel");
        if (ln) HIP_TRY(launch_colpart_reduce(w.lnpart, gn, L, g + lay.ng[step], L, g + lay.nb[step], L, s), "node LayerNorm gradient");
        HIP_TRY(launch_wgrad_batch(L, wb, w.gpart, w.cpart, s), "node weight gradients");
      }
    }
    if (E == 0) continue;
    // ---- edge module: e' = LN(MLP([x_s, x_d, e]) + e), first layer split into P_s[src] + P_d[dst] + W0e e ----
    {
      NodeArgs pa{};
      pa.N = N;
      pa.x_in = x_in;
      fill_node_proj(pa, d, lay, step, params, packed, w.Ps, w.Pd);
      HIP_TRY(run_node(d, pa, s), "projection recompute");
      ChainBwdArgs a{};
      a.mode = AMBER_BWD_EDGE;
      a.split = split;
      a.rows = E;
      a.rin = e_in;
      a.Ps = w.Ps;
      a.Pd = w.Pd;
      a.src = src_sorted;
      a.dst = dst_sorted;
      a.rowptr = rowptr;
      a.nl = nl;
      a.act = d.activation;
      a.residual = d.residual;
      a.layer_norm = d.layer_norm;
      a.gamma = ln ? params + lay.eg[step] : nullptr;
      a.eps = d.ln_eps;
      a.w_in = mat(12);
      a.wT_in = mat(13);
      for (int l = 1; l < nl; ++l) {
        a.layers[l] = DenseLayer{mat(12 + 2 * l), params + me.B[l], nullptr, nullptr};
        a.wT[l] = mat(13 + 2 * l);
      }
      for (int l = 0; l < nl; ++l) {
        a.A[l] = w.AE[l];
        a.DH[l] = w.DHE[l];
      }
      a.dio = w.dE;
      a.dm_in = w.DM;
      a.part = ln ? w.lnpart : nullptr;
      WgradBatch we{};
      we.split = split;
      we.M = E;
      wjob(we, w.DHE[0], e_in, g + me.W[0], 3 * L, 2 * L, nullptr);  // dW0e
      for (int l = 1; l < nl; ++l) wjob(we, w.DHE[l], w.AE[l - 1], g + me.W[l], L, 0, g + me.B[l]);
      if (edge_bwd_fused_supported(L, a) && !env_switches().no_fused_backward) {
        // chain and weight gradients in one pass over the edges; what is left of the latter is the ordered reduction
        const int gf = edge_bwd_fused_grid(E);
        HIP_TRY(launch_edge_bwd_fused(L, a, w.gpart, w.cpart, s), "fused edge backward kernel");
        if (ln) HIP_TRY(launch_colpart_reduce(w.lnpart, gf, L, g + lay.eg[step], L, g + lay.eb[step], L, s), "edge LayerNorm gradient");
        HIP_TRY(launch_wgrad_reduce(L, we, w.gpart, w.cpart, gf, s), "edge weight gradients (reduction)");
      } else {
        HIP_TRY(launch_chain_bwd(L, a, s), "edge backward kernel");
        if (ln) HIP_TRY(launch_colpart_reduce(w.lnpart, ge, L, g + lay.eg[step], L, g + lay.eb[step], L, s), "edge LayerNorm gradient");
        HIP_TRY(launch_wgrad_batch(L, we, w.gpart, w.cpart, s), "edge weight gradients");
      }
      // gathered first-layer terms: gradients summed per destination (CSR rows) and per source (source-sorted order)
      HIP_TRY(launch_segment_reduce(w.DHE[0], rowptr, nullptr, N, L, 0, w.dPd, s), "dP_d");
      HIP_TRY(launch_segment_reduce(w.DHE[0], rowptr_src, perm_src, N, L, 0, w.dPs, s), "dP_s");
      WgradBatch wp{};
      wp.split = split;
      wp.M = N;
      wjob(wp, w.dPs, x_in, g + me.W[0], 3 * L, 0, nullptr);       // dW0s
      wjob(wp, w.dPd, x_in, g + me.W[0], 3 * L, L, g + me.B[0]);   // dW0d, db0
      HIP_TRY(launch_wgrad_batch(L, wp, w.gpart, w.cpart, s), "projection weight gradients");
      HIP_TRY(launch_rows_gemm_add(L, split, w.dX, w.dPs, mat(20), w.dPd, mat(21), N, s), "dx from the projections");
    }
  }
  // ---- input embeddings (src/mpn/input_embedding.py:41-42; Embedding.forward, src/mpn/common/embedding.py:44-54) ----
  // linear / constant: d W = d^T raw, d b = column sums.  LatentMLP (embed_layers = K >= 1): a_0 = act(W_0 raw + b_0),
  // a_l = act(W_l a_{l-1} + b_l); the tail l >= 1 is a dense chain (chain kernel, node mode without an aggregate) on a
  // recomputed a_0, then act' of layer 0 and the raw-feature product.
  auto embedding_bwd = [&](const MlpLayout& m, int F, const float* raw, const int32_t* rperm, int64_t rows, float* dio,
                           float* const* A, float* const* DH, const float* out0) -> int {
    const int K = d.embed_layers;
    if (F == 0 || K == 0) {
      HIP_TRY(launch_embed_bwd(L, dio, raw, rperm, rows, F, F > 0 ? g + m.W[0] : nullptr, g + m.B[0], w.epart, s), "embedding gradient");
      return AMBER_OK;
    }
    const float* a0 = out0;  // K == 1: the embedding's output is a_0
    if (K > 1) {
      float* a0buf = A[K - 1];
      HIP_TRY(launch_embed(L, raw, rows, F, params + m.W[0], params + m.B[0], rperm, d.embed_activation, a0buf, s), "embedding recompute");
      a0 = a0buf;
      PackBatch pb{};
      for (int l = 1; l < K; ++l) {
        pb.job[pb.n++] = PackJob{params + m.W[l], mat(2 * (l - 1)), L, 0, 0, fmt};
        pb.job[pb.n++] = PackJob{params + m.W[l], mat(2 * (l - 1) + 1), L, 0, 1, fmt};
      }
      HIP_TRY(launch_pack_batch(L, pb, s), "pack embedding matrices");
      ChainBwdArgs a{};
      a.mode = AMBER_BWD_NODE;
      a.split = split;
      a.rows = rows;
      a.rin = a0;
      a.m = nullptr;  // no aggregate: a plain dense chain
      a.nl = K - 1;
      a.act = d.embed_activation;
      a.residual = AMBER_MODE_NONE;
      a.layer_norm = AMBER_MODE_NONE;
      a.eps = d.ln_eps;
      a.w_in = mat(0);
      a.wT_in = mat(1);
      a.b0 = params + m.B[1];
      for (int l = 1; l < K - 1; ++l) {
        a.layers[l] = DenseLayer{mat(2 * l), params + m.B[l + 1], nullptr, nullptr};
        a.wT[l] = mat(2 * l + 1);
      }
      for (int l = 0; l < K - 1; ++l) {
        a.A[l] = A[l];
        a.DH[l] = DH[l];
      }
      a.dio = dio;
      HIP_TRY(launch_chain_bwd(L, a, s), "embedding MLP backward kernel");
      WgradBatch wb{};
      wb.split = split;
      wb.M = rows;
      wjob(wb, DH[0], a0, g + m.W[1], L, 0, g + m.B[1]);
      for (int l = 1; l < K - 1; ++l) wjob(wb, DH[l], A[l - 1], g + m.W[l + 1], L, 0, g + m.B[l + 1]);
      HIP_TRY(launch_wgrad_batch(L, wb, w.gpart, w.cpart, s), "embedding MLP weight gradients");
    }
    HIP_TRY(launch_mul_dact(dio, a0, rows * L, d.embed_activation, s), "embedding activation gradient");
    HIP_TRY(launch_embed_bwd(L, dio, raw, rperm, rows, F, g + m.W[0], g + m.B[0], w.epart, s), "embedding gradient");
    return AMBER_OK;
  };
  int rc = embedding_bwd(lay.node_embed, d.in_node_features, x, nullptr, N, w.dX, w.AN, w.DHN, xbuf(0));
  if (rc) return rc;
  if (E > 0) {
    rc = embedding_bwd(lay.edge_embed, d.in_edge_features, edge_attr, perm, E, w.dE, w.AE, w.DHE, ebuf(0));
    if (rc) return rc;
  }
  return AMBER_OK;
}

}  // namespace
}  // namespace amber

using namespace amber;

extern "C" {

int amber_mpn_abi_version(void) { return AMBER_MPN_ABI_VERSION; }
const char* amber_mpn_last_error(void) { return g_last_error.c_str(); }

int64_t amber_mpn_param_floats(const AmberMpnDesc* desc) {
  if (validate(desc)) return -1;
  return build_layout(*desc).n_params;
}
int64_t amber_mpn_packed_floats(const AmberMpnDesc* desc) {
  if (validate(desc)) return -1;
  const int64_t n = build_layout(*desc).n_packed;
  return n > 0 ? n : 4;  // never hand out a zero-sized buffer
}

int amber_mpn_pack_weights(const AmberMpnDesc* desc, const float* params, float* packed, void* stream) {
  int rc = validate(desc);
  if (rc) return rc;
  if (!params || !packed) return fail(AMBER_ERR_INVALID_ARGUMENT, "params / packed is NULL");
  const Layout lay = build_layout(*desc);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int L = desc->latent;
  // Per matrix the operand format follows its consumer: FMT_F32 = exact fp32 tiles (A operand of the row-per-lane
  // chain: exact kernels, decoder); FMT_F2 = hi/lo fp16 as the B operand of the feature-per-lane chain (split-precision
  // edge and node kernels).  `f0[cb]` is the format of column block cb of layer 0, `share0` makes its FMT_F2 blocks
  // share one power-of-two scale (they feed one accumulator), `frest` is the format of layers >= 1.
  // At latent 256 outside the exact mode the FMT_F32 matrices are bf16x3 pieces instead (use_b3_forward).
  enum { FMT_F32 = 0, FMT_F2 = 2 };
  const bool b3fwd = use_b3_forward(*desc);
  // the matrices are queued and re-tiled in batched launches (a training step packs ~150 matrices after every
  // optimizer step: one launch each would be ~450 tiny kernels)
  F2PackBatch q2{};
  PackBatch q32{};
  auto flush = [&]() -> hipError_t {
    hipError_t e = launch_pack_f2_batch(L, q2, s);
    q2.n = 0;
    if (e != hipSuccess) return e;
    e = launch_pack_batch(L, q32, s);
    q32.n = 0;
    return e;
  };
  // kperm: the matrix is applied to a tile held in accumulator registers (hidden layers, projections of x', decoder)
  auto pack_one = [&](int fmt, const float* W, int ld, int col0, int scol0, int scols, int kperm, float* out) -> hipError_t {
    if (fmt == FMT_F2) {
      q2.job[q2.n++] = F2PackJob{W, out, ld, col0, scol0, scols, kperm};
      if (q2.n == kF2PackMaxJobs) return flush();
    } else {
      q32.job[q32.n++] = PackJob{W, out, ld, col0, 0, b3fwd ? AMBER_PACK_B3 : AMBER_PACK_F32};
      if (q32.n == kPackMaxJobs) return flush();
    }
    return hipSuccess;
  };
  auto pack_mlp = [&](const MlpLayout& m, int blocks0, int f00 = FMT_F32, int f01 = FMT_F32, int f02 = FMT_F32, bool share0 = false,
                      int frest = FMT_F32, int kperm0 = 0) -> hipError_t {  // kperm0: bit cb = column block cb of layer 0
    const int f0[3] = {f00, f01, f02};
    for (int l = 0; l < m.layers; ++l) {
      if (l == 0) {
        for (int cb = 0; cb < blocks0; ++cb)
          if (m.PW[0][cb] >= 0) {
            hipError_t e = pack_one(f0[cb], params + m.W[0], m.in_cols, cb * L, share0 ? 0 : cb * L, share0 ? blocks0 * L : L,
                                    (kperm0 >> cb) & 1, packed + m.PW[0][cb]);
            if (e != hipSuccess) return e;
          }
      } else if (m.PW[l][0] >= 0) {
        hipError_t e = pack_one(frest, params + m.W[l], L, 0, 0, L, 1, packed + m.PW[l][0]);
        if (e != hipSuccess) return e;
      }
    }
    return hipSuccess;
  };
  HIP_TRY(pack_mlp(lay.node_embed, 0), "pack node embedding");
  HIP_TRY(pack_mlp(lay.edge_embed, 0), "pack edge embedding");
  for (int st = 0; st < desc->num_steps; ++st) {
    const int fe = use_fast_edge(*desc) ? FMT_F2 : FMT_F32, fn = use_fast_node(*desc) ? FMT_F2 : FMT_F32;
    // W0s / W0d are applied by the node kernel to x' in accumulator registers, W0e by the edge kernel to the rows of e,
    // which it also holds in accumulator layout: the k order of all three follows the permuted columns
    HIP_TRY(pack_mlp(lay.edge[st], 3, fn, fn, fe, false, fe, 0b111), "pack edge MLP");
    HIP_TRY(pack_mlp(lay.node[st], 2, fn, fn, FMT_F32, true, fn), "pack node MLP");
  }
  if (desc->has_head) {
    HIP_TRY(pack_mlp(lay.dec, 1), "pack decoder");
    for (int l = 0; l < desc->decoder_layers; ++l)
      if (lay.dec_f2[l] >= 0)
        HIP_TRY(pack_one(FMT_F2, params + lay.dec.W[l], L, 0, 0, L, 1, packed + lay.dec_f2[l]), "pack decoder (split precision)");
  }
  HIP_TRY(flush(), "pack weights");
  if (lay.node_fold >= 0) {
    const int F = desc->in_node_features;
    HIP_TRY(launch_fold_node_tables(L, F, F > 0 ? params + lay.node_embed.W[0] : nullptr, params + lay.node_embed.B[0],
                                    params + lay.edge[0].W[0], params + lay.edge[0].B[0], packed + lay.node_fold, s),
            "fold node tables");
  }
  return AMBER_OK;
}

size_t amber_mpn_csr_workspace_bytes(int64_t num_nodes, int64_t num_edges) {
  if (num_nodes < 0 || num_edges < 0) return 0;
  return csr_workspace_bytes(num_nodes, num_edges) + 256;
}

int amber_mpn_csr_build(const int64_t* edge_index, int64_t num_edges, int64_t num_nodes, int32_t* rowptr, int32_t* src_sorted,
                        int32_t* dst_sorted, int32_t* perm, int32_t* status, void* workspace, size_t workspace_bytes,
                        void* stream) {
  if (num_nodes < 0 || num_edges < 0 || num_nodes >= (1ll << 31) || num_edges >= (1ll << 31))
    return fail(AMBER_ERR_INVALID_ARGUMENT, "num_nodes / num_edges out of int32 range");
  if (!rowptr || !status || !workspace) return fail(AMBER_ERR_INVALID_ARGUMENT, "rowptr / status / workspace is NULL");
  if (num_edges > 0 && (!edge_index || !src_sorted || !dst_sorted || !perm))
    return fail(AMBER_ERR_INVALID_ARGUMENT, "edge arrays are NULL");
  if (num_edges > 0 && num_nodes == 0) return fail(AMBER_ERR_INVALID_ARGUMENT, "edges without nodes");
  if (workspace_bytes < amber_mpn_csr_workspace_bytes(num_nodes, num_edges))
    return fail(AMBER_ERR_WORKSPACE_TOO_SMALL, "csr workspace %zu < %zu bytes", workspace_bytes,
                amber_mpn_csr_workspace_bytes(num_nodes, num_edges));
  (void)hipGetLastError();
  // carve from a 256-byte aligned base
  uintptr_t base = (reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255);
  HIP_TRY(csr_build(edge_index, num_edges, num_nodes, rowptr, src_sorted, dst_sorted, perm, status, reinterpret_cast<void*>(base),
                    static_cast<hipStream_t>(stream)),
          "csr_build");
  return AMBER_OK;
}

int amber_mpn_csr_build_static(const int64_t* edge_index, int64_t num_edges, int64_t num_nodes, int32_t* rowptr,
                               int32_t* src_sorted, int32_t* dst_sorted, int32_t* perm, int32_t* status, void* stream) {
  if (num_nodes < 0 || num_edges < 0 || num_nodes >= (1ll << 31) || num_edges >= (1ll << 31))
    return fail(AMBER_ERR_INVALID_ARGUMENT, "num_nodes / num_edges out of int32 range");
  if (!rowptr || !status) return fail(AMBER_ERR_INVALID_ARGUMENT, "rowptr / status is NULL");
  if (num_edges > 0 && (!edge_index || !src_sorted || !dst_sorted || !perm))
    return fail(AMBER_ERR_INVALID_ARGUMENT, "edge arrays are NULL");
  if (num_edges > 0 && num_nodes == 0) return fail(AMBER_ERR_INVALID_ARGUMENT, "edges without nodes");
  const size_t need = csr_workspace_bytes(num_nodes, num_edges);
  if (need > static_cast<size_t>(AMBER_MPN_CSR_STATIC_SCRATCH_BYTES))
    return fail(AMBER_ERR_WORKSPACE_TOO_SMALL,
                "graph needs %zu bytes of CSR scratch, the static array holds %zu (use amber_mpn_csr_build with a workspace)", need,
                static_cast<size_t>(AMBER_MPN_CSR_STATIC_SCRATCH_BYTES));
  (void)hipGetLastError();
  void* ws = nullptr;
  HIP_TRY(csr_static_scratch(&ws), "address of the static CSR scratch");
  HIP_TRY(csr_build(edge_index, num_edges, num_nodes, rowptr, src_sorted, dst_sorted, perm, status, ws,
                    static_cast<hipStream_t>(stream)),
          "csr_build (static scratch)");
  return AMBER_OK;
}

int amber_mpn_segment_mean(const float* values_sorted, const int32_t* rowptr, int64_t num_nodes, int64_t num_edges, int32_t latent,
                           float* out, void* stream) {
  if (num_nodes < 0 || num_edges < 0) return fail(AMBER_ERR_INVALID_ARGUMENT, "negative size");
  if (latent <= 0 || latent % 4 != 0 || latent > 1024) return fail(AMBER_ERR_UNSUPPORTED, "latent %d: need a multiple of 4, <= 1024", latent);
  if (num_nodes > 0 && (!rowptr || !out)) return fail(AMBER_ERR_INVALID_ARGUMENT, "rowptr / out is NULL");
  if (num_edges > 0 && !values_sorted) return fail(AMBER_ERR_INVALID_ARGUMENT, "values is NULL");
  (void)hipGetLastError();
  HIP_TRY(launch_segment_mean(values_sorted, rowptr, num_nodes, latent, out, static_cast<hipStream_t>(stream)), "segment_mean");
  return AMBER_OK;
}

int amber_mpn_permute_rows(const float* in, const int32_t* perm, int64_t rows, int32_t width, int32_t gather, float* out,
                           void* stream) {
  if (rows < 0 || width <= 0 || width % 4 != 0) return fail(AMBER_ERR_INVALID_ARGUMENT, "bad rows / width");
  if (rows > 0 && (!in || !perm || !out)) return fail(AMBER_ERR_INVALID_ARGUMENT, "NULL pointer");
  (void)hipGetLastError();
  HIP_TRY(launch_permute_rows(in, perm, rows, width, gather, out, static_cast<hipStream_t>(stream)), "permute_rows");
  return AMBER_OK;
}

size_t amber_mpn_workspace_bytes(const AmberMpnDesc* desc, int64_t num_nodes, int64_t num_edges) {
  if (validate(desc) || num_nodes < 0 || num_edges < 0) return 0;
  return carve(*desc, num_nodes, num_edges, nullptr).bytes + 256;
}

static void* aligned_ws(void* workspace) {
  return reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
}

int amber_mpn_embed(const AmberMpnDesc* desc, const float* params, const float* packed, int64_t num_nodes, int64_t num_edges,
                    const float* x_raw, const float* edge_attr_raw, const int32_t* perm, float* x_latent, float* e_latent_sorted,
                    void* workspace, size_t workspace_bytes, void* stream) {
  (void)workspace;
  (void)workspace_bytes;
  int rc = validate(desc);
  if (rc) return rc;
  if (!params || !packed) return fail(AMBER_ERR_INVALID_ARGUMENT, "params / packed is NULL");
  if (num_nodes > 0 && (!x_latent || (desc->in_node_features > 0 && !x_raw))) return fail(AMBER_ERR_INVALID_ARGUMENT, "node buffers NULL");
  if (num_edges > 0 && (!e_latent_sorted || !perm || (desc->in_edge_features > 0 && !edge_attr_raw)))
    return fail(AMBER_ERR_INVALID_ARGUMENT, "edge buffers NULL");
  const Layout lay = build_layout(*desc);
  Timer tm;
  hipStream_t s = static_cast<hipStream_t>(stream);
  rc = run_embedding(*desc, lay.node_embed, desc->in_node_features, params, packed, x_raw, num_nodes, nullptr, x_latent, s, tm);
  if (rc) return rc;
  return run_embedding(*desc, lay.edge_embed, desc->in_edge_features, params, packed, edge_attr_raw, num_edges, perm,
                       e_latent_sorted, s, tm);
}

int amber_mpn_block(const AmberMpnDesc* desc, const float* params, const float* packed, int32_t step, int64_t num_nodes,
                    int64_t num_edges, const int32_t* rowptr, const int32_t* src_sorted, const int32_t* dst_sorted, float* x_latent,
                    float* e_latent_sorted, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = validate(desc);
  if (rc) return rc;
  if (step < 0 || step >= desc->num_steps) return fail(AMBER_ERR_INVALID_ARGUMENT, "step %d out of range", step);
  if (!params || !packed || !workspace) return fail(AMBER_ERR_INVALID_ARGUMENT, "params / packed / workspace is NULL");
  if (workspace_bytes < amber_mpn_workspace_bytes(desc, num_nodes, num_edges))
    return fail(AMBER_ERR_WORKSPACE_TOO_SMALL, "workspace too small");
  if (num_nodes == 0) return AMBER_OK;
  if (!rowptr || !x_latent) return fail(AMBER_ERR_INVALID_ARGUMENT, "rowptr / x_latent is NULL");
  if (num_edges > 0 && (!src_sorted || !dst_sorted || !e_latent_sorted)) return fail(AMBER_ERR_INVALID_ARGUMENT, "edge buffers NULL");
  const Layout lay = build_layout(*desc);
  const Workspace w = carve(*desc, num_nodes, num_edges, aligned_ws(workspace));
  hipStream_t s = static_cast<hipStream_t>(stream);
  NodeArgs pa{};
  pa.N = num_nodes;
  pa.x_in = x_latent;
  fill_node_proj(pa, *desc, lay, step, params, packed, w.Ps, w.Pd);
  HIP_TRY(run_node(*desc, pa, s), "node projection kernel");
  EdgeArgs ea{};
  ea.E = num_edges;
  ea.N = num_nodes;
  ea.src = src_sorted;
  ea.dst = dst_sorted;
  ea.rowptr = rowptr;
  ea.e = e_latent_sorted;
  ea.Ps = w.Ps;
  ea.Pd = w.Pd;
  fill_edge_step(ea, *desc, lay, step, params, packed);
  ea.agg = w.agg;
  ea.head = w.head;
  ea.tail = w.tail;
  ea.write_e = 1;
  HIP_TRY(run_edge(*desc, ea, s), "edge kernel");
  NodeArgs na{};
  na.N = num_nodes;
  na.x_in = x_latent;
  na.x_out = x_latent;
  na.rowptr = rowptr;
  na.agg = w.agg;
  na.head = w.head;
  na.tail = w.tail;
  fill_node_update(na, *desc, lay, step, params, packed);
  HIP_TRY(run_node(*desc, na, s), "node kernel");
  return AMBER_OK;
}

int amber_mpn_project(const AmberMpnDesc* desc, const float* params, const float* packed, int32_t step, int64_t num_nodes,
                      const float* x_latent, float* p_src, float* p_dst, void* stream) {
  int rc = validate(desc);
  if (rc) return rc;
  if (step < 0 || step >= desc->num_steps) return fail(AMBER_ERR_INVALID_ARGUMENT, "step %d out of range", step);
  if (!params || !packed) return fail(AMBER_ERR_INVALID_ARGUMENT, "params / packed is NULL");
  if (num_nodes == 0) return AMBER_OK;
  if (!x_latent || !p_src || !p_dst) return fail(AMBER_ERR_INVALID_ARGUMENT, "x_latent / p_src / p_dst is NULL");
  const Layout lay = build_layout(*desc);
  NodeArgs pa{};
  pa.N = num_nodes;
  pa.x_in = x_latent;
  fill_node_proj(pa, *desc, lay, step, params, packed, p_src, p_dst);
  HIP_TRY(run_node(*desc, pa, static_cast<hipStream_t>(stream)), "node projection kernel");
  return AMBER_OK;
}

int amber_mpn_head(const AmberMpnDesc* desc, const float* params, const float* packed, int64_t num_nodes, const float* x_latent,
                   const int32_t* out_index, float* out_sizing, void* stream) {
  int rc = validate(desc);
  if (rc) return rc;
  if (!desc->has_head) return fail(AMBER_ERR_INVALID_ARGUMENT, "desc->has_head == 0");
  if (!params || !packed) return fail(AMBER_ERR_INVALID_ARGUMENT, "params / packed is NULL");
  if (num_nodes == 0) return AMBER_OK;
  if (!x_latent || !out_sizing) return fail(AMBER_ERR_INVALID_ARGUMENT, "x_latent / out_sizing is NULL");
  const Layout lay = build_layout(*desc);
  NodeArgs a{};
  a.N = num_nodes;
  a.x_in = x_latent;
  fill_node_head(a, *desc, lay, params, packed, out_index, out_sizing);
  HIP_TRY(launch_node(desc->latent, use_b3_forward(*desc) ? 1 : 0, a, static_cast<hipStream_t>(stream)), "head kernel");
  return AMBER_OK;
}

int amber_mpn_forward(const AmberMpnDesc* desc, const float* params, const float* packed, int64_t num_nodes, int64_t num_edges,
                      const float* x, const float* edge_attr, const int32_t* rowptr, const int32_t* src_sorted,
                      const int32_t* dst_sorted, const int32_t* perm, const int32_t* out_index, float* out_node, float* out_edge,
                      float* out_sizing, void* workspace, size_t workspace_bytes, void* stream) {
  Timer tm;
  return forward_impl(desc, params, packed, num_nodes, num_edges, x, edge_attr, rowptr, src_sorted, dst_sorted, perm, out_index,
                      out_node, out_edge, out_sizing, workspace ? aligned_ws(workspace) : nullptr, workspace_bytes,
                      static_cast<hipStream_t>(stream), tm);
}

int amber_mpn_forward_timed(const AmberMpnDesc* desc, const float* params, const float* packed, int64_t num_nodes,
                            int64_t num_edges, const float* x, const float* edge_attr, const int32_t* rowptr,
                            const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* perm, const int32_t* out_index,
                            float* out_node, float* out_edge, float* out_sizing, void* workspace, size_t workspace_bytes,
                            void* stream, float* kernel_ms, int32_t* kernel_launches) {
  if (!kernel_ms) return fail(AMBER_ERR_INVALID_ARGUMENT, "kernel_ms is NULL");
  Timer tm;
  tm.on = true;
  tm.s = static_cast<hipStream_t>(stream);
  int rc = forward_impl(desc, params, packed, num_nodes, num_edges, x, edge_attr, rowptr, src_sorted, dst_sorted, perm, out_index,
                        out_node, out_edge, out_sizing, workspace ? aligned_ws(workspace) : nullptr, workspace_bytes, tm.s, tm);
  tm.collect(kernel_ms, kernel_launches);
  return rc;
}

size_t amber_mpn_tape_bytes(const AmberMpnDesc* desc, int64_t num_nodes, int64_t num_edges) {
  if (validate(desc) || num_nodes < 0 || num_edges < 0) return 0;
  return carve_tape(*desc, num_nodes, num_edges, nullptr).bytes + 256;
}

int amber_mpn_forward_train(const AmberMpnDesc* desc, const float* params, const float* packed, int64_t num_nodes, int64_t num_edges,
                            const float* x, const float* edge_attr, const int32_t* rowptr, const int32_t* src_sorted,
                            const int32_t* dst_sorted, const int32_t* perm, const int32_t* out_index, float* out_node, float* out_edge,
                            float* out_sizing, void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = validate(desc);
  if (rc) return rc;
  rc = validate_train(*desc);
  if (rc) return rc;
  if (!tape) return fail(AMBER_ERR_INVALID_ARGUMENT, "tape is NULL");
  if (tape_bytes < amber_mpn_tape_bytes(desc, num_nodes, num_edges))
    return fail(AMBER_ERR_WORKSPACE_TOO_SMALL, "tape %zu < %zu bytes", tape_bytes, amber_mpn_tape_bytes(desc, num_nodes, num_edges));
  Timer tm;
  return forward_impl(desc, params, packed, num_nodes, num_edges, x, edge_attr, rowptr, src_sorted, dst_sorted, perm, out_index,
                      out_node, out_edge, out_sizing, workspace ? aligned_ws(workspace) : nullptr, workspace_bytes,
                      static_cast<hipStream_t>(stream), tm, static_cast<float*>(aligned_ws(tape)));
}

size_t amber_mpn_backward_workspace_bytes(const AmberMpnDesc* desc, int64_t num_nodes, int64_t num_edges) {
  if (validate(desc) || num_nodes < 0 || num_edges < 0) return 0;
  const size_t fw = amber_mpn_workspace_bytes(desc, num_nodes, num_edges);
  const size_t bw = carve_bwd(*desc, num_nodes, num_edges, nullptr).bytes + 256;
  return fw > bw ? fw : bw;
}

int amber_mpn_backward(const AmberMpnDesc* desc, const float* params, const float* packed, int64_t num_nodes, int64_t num_edges,
                       const float* x, const float* edge_attr, const int32_t* rowptr, const int32_t* src_sorted,
                       const int32_t* dst_sorted, const int32_t* perm, const int32_t* rowptr_src, const int32_t* perm_src,
                       const int32_t* out_index, const float* grad_node, const float* grad_edge, const float* grad_sizing,
                       void* tape, size_t tape_bytes, float* grad_params, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = validate(desc);
  if (rc) return rc;
  rc = validate_train(*desc);
  if (rc) return rc;
  const int64_t N = num_nodes, E = num_edges;
  if (N < 0 || E < 0 || N >= (1ll << 31) || E >= (1ll << 31)) return fail(AMBER_ERR_INVALID_ARGUMENT, "N / E out of int32 range");
  if (!params || !packed || !tape || !grad_params || !workspace)
    return fail(AMBER_ERR_INVALID_ARGUMENT, "params / packed / tape / grad_params / workspace is NULL");
  if (N > 0 && desc->in_node_features > 0 && !x) return fail(AMBER_ERR_INVALID_ARGUMENT, "x is NULL");
  if (E > 0 && desc->in_edge_features > 0 && !edge_attr) return fail(AMBER_ERR_INVALID_ARGUMENT, "edge_attr is NULL");
  if (E > 0 && (!rowptr || !src_sorted || !dst_sorted || !perm || !rowptr_src || !perm_src))
    return fail(AMBER_ERR_INVALID_ARGUMENT, "CSR arrays are NULL");
  if (N > 0 && !rowptr) return fail(AMBER_ERR_INVALID_ARGUMENT, "rowptr is NULL");
  if (grad_sizing && !desc->has_head) return fail(AMBER_ERR_INVALID_ARGUMENT, "grad_sizing given but desc->has_head == 0");
  if (tape_bytes < amber_mpn_tape_bytes(desc, N, E)) return fail(AMBER_ERR_WORKSPACE_TOO_SMALL, "tape too small");
  if (workspace_bytes < amber_mpn_backward_workspace_bytes(desc, N, E))
    return fail(AMBER_ERR_WORKSPACE_TOO_SMALL, "workspace %zu < %zu bytes", workspace_bytes,
                amber_mpn_backward_workspace_bytes(desc, N, E));
  return backward_impl(desc, params, packed, N, E, x, edge_attr, rowptr, src_sorted, dst_sorted, perm, rowptr_src, perm_src, out_index,
                       grad_node, grad_edge, grad_sizing, static_cast<float*>(aligned_ws(tape)), grad_params, aligned_ws(workspace),
                       static_cast<hipStream_t>(stream));
}

// ---- pre / post glue of _predict ----
static int make_post_args(const AmberGlueDesc* g, const float* baseline, const double* pred_mean, const double* pred_var,
                          bool differentiable, PostArgs& p) {
  if (!g) return fail(AMBER_ERR_INVALID_ARGUMENT, "glue is NULL");
  if (g->transform < AMBER_TRANSFORM_NONE || g->transform > AMBER_TRANSFORM_EXP)
    return fail(AMBER_ERR_UNSUPPORTED, "Unsupported transform type: id %d", g->transform);
  if (g->denormalize_predictions && (!pred_mean || !pred_var))
    return fail(AMBER_ERR_INVALID_ARGUMENT, "denormalize_predictions needs pred_mean / pred_var");
  if (g->apply_transform && g->predict_residual && !baseline)
    return fail(AMBER_ERR_INVALID_ARGUMENT, "Baseline must be provided for residual prediction.");
  if (differentiable && (g->clip || g->scale0 != 1.f || g->scale1 != 1.f))
    return fail(AMBER_ERR_INVALID_ARGUMENT, "clip / damping belong to the detached inference path and have no gradient");
  p = PostArgs{};
  p.on = 1;
  p.denorm = g->denormalize_predictions ? 1 : 0;
  p.transform = g->transform;
  p.residual = g->predict_residual ? 1 : 0;
  p.apply_transform = g->apply_transform ? 1 : 0;
  p.clip = g->clip & 3;
  p.mean = pred_mean;
  p.var = pred_var;
  p.eps = g->epsilon;
  p.baseline = baseline;
  p.clip_min = g->clip_min;
  p.clip_max = g->clip_max;
  p.scale0 = g->scale0;
  p.scale1 = g->scale1;
  return AMBER_OK;
}

int amber_mpn_normalize_rows(const float* in, int64_t rows, int32_t width, const double* mean, const double* var, double epsilon,
                             double clip, float* out, void* stream) {
  if (rows < 0 || width < 0) return fail(AMBER_ERR_INVALID_ARGUMENT, "negative size");
  if (rows * width == 0) return AMBER_OK;
  if (!in || !out || !mean || !var) return fail(AMBER_ERR_INVALID_ARGUMENT, "NULL pointer");
  (void)hipGetLastError();
  HIP_TRY(launch_normalize_rows(in, rows, width, mean, var, epsilon, clip, out, static_cast<hipStream_t>(stream)), "normalize_rows");
  return AMBER_OK;
}

int amber_mpn_postprocess(const AmberGlueDesc* glue, const float* pred, const float* baseline, const double* pred_mean,
                          const double* pred_var, int64_t n, float* out, void* stream) {
  PostArgs p;
  int rc = make_post_args(glue, baseline, pred_mean, pred_var, false, p);
  if (rc) return rc;
  if (n < 0) return fail(AMBER_ERR_INVALID_ARGUMENT, "negative size");
  if (n == 0) return AMBER_OK;
  if (!pred || !out) return fail(AMBER_ERR_INVALID_ARGUMENT, "pred / out is NULL");
  (void)hipGetLastError();
  HIP_TRY(launch_postprocess(p, pred, n, out, static_cast<hipStream_t>(stream)), "postprocess");
  return AMBER_OK;
}

int amber_mpn_postprocess_backward(const AmberGlueDesc* glue, const float* pred, const float* baseline, const double* pred_mean,
                                   const double* pred_var, const float* grad_out, int64_t n, float* grad_pred, void* stream) {
  PostArgs p;
  int rc = make_post_args(glue, baseline, pred_mean, pred_var, true, p);
  if (rc) return rc;
  if (n < 0) return fail(AMBER_ERR_INVALID_ARGUMENT, "negative size");
  if (n == 0) return AMBER_OK;
  if (!pred || !grad_out || !grad_pred) return fail(AMBER_ERR_INVALID_ARGUMENT, "pred / grad_out / grad_pred is NULL");
  (void)hipGetLastError();
  HIP_TRY(launch_postprocess_bwd(p, pred, grad_out, n, grad_pred, static_cast<hipStream_t>(stream)), "postprocess backward");
  return AMBER_OK;
}

int amber_mpn_inverse_transform(int32_t transform, const float* in, const float* baseline, int64_t n, float* out, void* stream) {
  if (transform < AMBER_TRANSFORM_NONE || transform > AMBER_TRANSFORM_EXP)
    return fail(AMBER_ERR_UNSUPPORTED, "Unsupported transform type: id %d", transform);
  if (n < 0) return fail(AMBER_ERR_INVALID_ARGUMENT, "negative size");
  if (n == 0) return AMBER_OK;
  if (!in || !out) return fail(AMBER_ERR_INVALID_ARGUMENT, "in / out is NULL");
  (void)hipGetLastError();
  HIP_TRY(launch_inverse_transform(transform, in, baseline, n, out, static_cast<hipStream_t>(stream)), "inverse_transform");
  return AMBER_OK;
}

// workspace of amber_mpn_predict: [forward workspace | normalised x | normalised edge_attr]
static size_t predict_norm_bytes(const AmberMpnDesc* d, int64_t N, int64_t E, size_t* xb_out) {
  const size_t xb = align256(sizeof(float) * static_cast<size_t>(N) * d->in_node_features);
  const size_t eb = align256(sizeof(float) * static_cast<size_t>(E) * d->in_edge_features);
  if (xb_out) *xb_out = xb;
  return xb + eb;
}
size_t amber_mpn_predict_workspace_bytes(const AmberMpnDesc* desc, int64_t num_nodes, int64_t num_edges) {
  const size_t fw = amber_mpn_workspace_bytes(desc, num_nodes, num_edges);
  if (fw == 0) return 0;
  return fw + predict_norm_bytes(desc, num_nodes, num_edges, nullptr);
}

int amber_mpn_predict(const AmberMpnDesc* desc, const AmberGlueDesc* glue, const float* params, const float* packed,
                      int64_t num_nodes, int64_t num_edges, const float* x, const float* edge_attr, const int32_t* rowptr,
                      const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* perm, const int32_t* out_index,
                      const double* node_mean, const double* node_var, const double* edge_mean, const double* edge_var,
                      const double* pred_mean, const double* pred_var, const float* baseline, float* out_sizing, void* workspace,
                      size_t workspace_bytes, void* stream) {
  int rc = validate(desc);
  if (rc) return rc;
  if (!desc->has_head) return fail(AMBER_ERR_INVALID_ARGUMENT, "amber_mpn_predict needs desc->has_head == 1");
  if (!out_sizing) return fail(AMBER_ERR_INVALID_ARGUMENT, "out_sizing is NULL");
  PostArgs p;
  rc = make_post_args(glue, baseline, pred_mean, pred_var, false, p);
  if (rc) return rc;
  if (num_nodes < 0 || num_edges < 0) return fail(AMBER_ERR_INVALID_ARGUMENT, "negative size");
  if (!workspace) return fail(AMBER_ERR_INVALID_ARGUMENT, "workspace is NULL");
  const size_t need = amber_mpn_predict_workspace_bytes(desc, num_nodes, num_edges);
  if (workspace_bytes < need) return fail(AMBER_ERR_WORKSPACE_TOO_SMALL, "workspace %zu < %zu bytes", workspace_bytes, need);
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* base = static_cast<char*>(aligned_ws(workspace));
  const size_t fw = amber_mpn_workspace_bytes(desc, num_nodes, num_edges) - 256;  // carved size, without the alignment slack
  const float* xin = x;
  const float* ein = edge_attr;
  if (glue->normalize_inputs) {
    size_t xb = 0;
    predict_norm_bytes(desc, num_nodes, num_edges, &xb);
    float* xn = reinterpret_cast<float*>(base + align256(fw));
    float* en = reinterpret_cast<float*>(base + align256(fw) + xb);
    if (num_nodes > 0 && desc->in_node_features > 0) {
      if (!x || !node_mean || !node_var) return fail(AMBER_ERR_INVALID_ARGUMENT, "x / node statistics are NULL");
      HIP_TRY(launch_normalize_rows(x, num_nodes, desc->in_node_features, node_mean, node_var, glue->epsilon, glue->input_clip, xn, s),
              "normalize x");
      xin = xn;
    }
    if (num_edges > 0 && desc->in_edge_features > 0) {
      if (!edge_attr || !edge_mean || !edge_var) return fail(AMBER_ERR_INVALID_ARGUMENT, "edge_attr / edge statistics are NULL");
      HIP_TRY(launch_normalize_rows(edge_attr, num_edges, desc->in_edge_features, edge_mean, edge_var, glue->epsilon,
                                    glue->input_clip, en, s),
              "normalize edge_attr");
      ein = en;
    }
  }
  Timer tm;
  return forward_impl(desc, params, packed, num_nodes, num_edges, xin, ein, rowptr, src_sorted, dst_sorted, perm, out_index, nullptr,
                      nullptr, out_sizing, base, workspace_bytes, s, tm, nullptr, &p);
}

// ---- graph construction ----
int32_t amber_mpn_graph_edge_features(const AmberGraphDesc* g) {
  if (!g || (g->dim != 2 && g->dim != 3)) return -1;
  return (g->distance_vector ? g->dim : 0) + (g->euclidean_distance ? 1 : 0) + (g->edge_curvature ? 1 : 0);
}

int amber_mpn_build_graph(const AmberGraphDesc* gdesc, int32_t num_graphs, const int64_t* node_ptr, const int64_t* edge_ptr,
                          const int64_t* nbr_a, const int64_t* nbr_b, const float* positions, const float* curvature,
                          int64_t num_nodes, int64_t num_pairs, int64_t* edge_index, float* edge_attr, int64_t* batch,
                          float* degree, int32_t* status, void* stream) {
  if (!gdesc) return fail(AMBER_ERR_INVALID_ARGUMENT, "gdesc is NULL");
  if (gdesc->dim != 2 && gdesc->dim != 3) return fail(AMBER_ERR_UNSUPPORTED, "positions must have 2 or 3 columns, given %d", gdesc->dim);
  if (num_graphs < 0 || num_nodes < 0 || num_pairs < 0) return fail(AMBER_ERR_INVALID_ARGUMENT, "negative size");
  if (!status) return fail(AMBER_ERR_INVALID_ARGUMENT, "status is NULL");
  if ((num_nodes > 0 || num_pairs > 0) && (num_graphs == 0 || !node_ptr || !edge_ptr))
    return fail(AMBER_ERR_INVALID_ARGUMENT, "node_ptr / edge_ptr missing");
  if (num_pairs > 0 && num_nodes == 0) return fail(AMBER_ERR_INVALID_ARGUMENT, "pairs without nodes");
  const int F = amber_mpn_graph_edge_features(gdesc);
  const int64_t E = 2 * num_pairs + (gdesc->add_self_edges ? num_nodes : 0);
  if (gdesc->edge_capacity < 0 || gdesc->edge_offset < 0 ||
      (gdesc->edge_capacity > 0 && gdesc->edge_offset + E > gdesc->edge_capacity) || (gdesc->edge_capacity == 0 && gdesc->edge_offset != 0))
    return fail(AMBER_ERR_INVALID_ARGUMENT, "edge_offset + %lld edges do not fit edge_capacity %lld", static_cast<long long>(E),
                static_cast<long long>(gdesc->edge_capacity));
  if (num_pairs > 0 && (!nbr_a || !nbr_b)) return fail(AMBER_ERR_INVALID_ARGUMENT, "nbr_a / nbr_b is NULL");
  if (E > 0 && !edge_index) return fail(AMBER_ERR_INVALID_ARGUMENT, "edge_index is NULL");
  if (E > 0 && F > 0 && !edge_attr) return fail(AMBER_ERR_INVALID_ARGUMENT, "edge_attr is NULL");
  if (num_pairs > 0 && (gdesc->distance_vector || gdesc->euclidean_distance) && !positions)
    return fail(AMBER_ERR_INVALID_ARGUMENT, "positions is NULL");
  (void)hipGetLastError();
  GraphArgs a{};
  a.G = num_graphs;
  a.dim = gdesc->dim;
  a.add_self_edges = gdesc->add_self_edges ? 1 : 0;
  a.distance_vector = gdesc->distance_vector ? 1 : 0;
  a.euclidean_distance = gdesc->euclidean_distance ? 1 : 0;
  a.edge_curvature = gdesc->edge_curvature ? 1 : 0;
  a.F = F;
  a.N = num_nodes;
  a.M = num_pairs;
  a.E = gdesc->edge_capacity > 0 ? gdesc->edge_capacity : E;
  a.eoff = gdesc->edge_offset;
  a.node_ptr = node_ptr;
  a.edge_ptr = edge_ptr;
  a.nbr_a = nbr_a;
  a.nbr_b = nbr_b;
  a.positions = positions;
  a.curvature = gdesc->edge_curvature ? curvature : nullptr;
  a.edge_index = edge_index;
  a.edge_attr = edge_attr;
  a.batch = batch;
  a.degree_i = reinterpret_cast<int32_t*>(degree);
  a.status = status;
  HIP_TRY(launch_build_graph(a, static_cast<hipStream_t>(stream)), "build_graph");
  return AMBER_OK;
}

int amber_mpn_build_inter_edges(const AmberGraphDesc* gdesc, int64_t num_src, int64_t num_dst, int64_t index_offset,
                                const int64_t* nearest, const double* pos_src, const double* pos_dst, int64_t* edge_index,
                                float* edge_attr, int32_t* status, void* stream) {
  if (!gdesc) return fail(AMBER_ERR_INVALID_ARGUMENT, "gdesc is NULL");
  if (gdesc->dim != 2 && gdesc->dim != 3) return fail(AMBER_ERR_UNSUPPORTED, "positions must have 2 or 3 columns, given %d", gdesc->dim);
  if (num_src < 0 || num_dst < 0 || index_offset < 0) return fail(AMBER_ERR_INVALID_ARGUMENT, "negative size");
  if (!status) return fail(AMBER_ERR_INVALID_ARGUMENT, "status is NULL");
  if (num_src > 0 && num_dst == 0) return fail(AMBER_ERR_INVALID_ARGUMENT, "no destination nodes");
  const int F = amber_mpn_graph_edge_features(gdesc);
  const int64_t E = 2 * num_src;
  if (gdesc->edge_capacity < 0 || gdesc->edge_offset < 0 ||
      (gdesc->edge_capacity > 0 && gdesc->edge_offset + E > gdesc->edge_capacity) || (gdesc->edge_capacity == 0 && gdesc->edge_offset != 0))
    return fail(AMBER_ERR_INVALID_ARGUMENT, "edge_offset + %lld edges do not fit edge_capacity %lld", static_cast<long long>(E),
                static_cast<long long>(gdesc->edge_capacity));
  if (num_src > 0 && (!nearest || !edge_index)) return fail(AMBER_ERR_INVALID_ARGUMENT, "nearest / edge_index is NULL");
  if (num_src > 0 && F > 0 && !edge_attr) return fail(AMBER_ERR_INVALID_ARGUMENT, "edge_attr is NULL");
  if (num_src > 0 && (gdesc->distance_vector || gdesc->euclidean_distance) && (!pos_src || !pos_dst))
    return fail(AMBER_ERR_INVALID_ARGUMENT, "positions are NULL");
  (void)hipGetLastError();
  GraphArgs a{};
  a.dim = gdesc->dim;
  a.distance_vector = gdesc->distance_vector ? 1 : 0;
  a.euclidean_distance = gdesc->euclidean_distance ? 1 : 0;
  a.edge_curvature = gdesc->edge_curvature ? 1 : 0;
  a.F = F;
  a.E = gdesc->edge_capacity > 0 ? gdesc->edge_capacity : E;
  a.eoff = gdesc->edge_offset;
  a.edge_index = edge_index;
  a.edge_attr = edge_attr;
  a.status = status;
  HIP_TRY(launch_inter_edges(a, nearest, pos_src, pos_dst, num_src, num_dst, index_offset, static_cast<hipStream_t>(stream)),
          "build_inter_edges");
  return AMBER_OK;
}

// ---- the step after the model ----
size_t amber_mpn_project_workspace_bytes(int32_t k, int64_t M, int64_t V) {
  if (k < 1 || k > 8 || M < 0 || V < 0) return 0;
  return project_workspace_bytes(k, M, V) + 256;
}

int amber_mpn_project_elements_to_vertices(const int64_t* simplices, int32_t k, int64_t M, int64_t V, const double* element_values,
                                           const double* volumes, double* out, int32_t* status, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  if (k < 1 || k > 8) return fail(AMBER_ERR_UNSUPPORTED, "verts_per_element %d not in 1..8", k);
  if (M < 0 || V < 0 || static_cast<int64_t>(k) * M >= (1ll << 31) || V >= (1ll << 31))
    return fail(AMBER_ERR_INVALID_ARGUMENT, "sizes out of int32 range");
  if (!status || !workspace) return fail(AMBER_ERR_INVALID_ARGUMENT, "status / workspace is NULL");
  if (M > 0 && (!simplices || !element_values || !volumes)) return fail(AMBER_ERR_INVALID_ARGUMENT, "element arrays are NULL");
  if (M > 0 && V == 0) return fail(AMBER_ERR_INVALID_ARGUMENT, "elements without vertices");
  if (V > 0 && !out) return fail(AMBER_ERR_INVALID_ARGUMENT, "out is NULL");
  if (workspace_bytes < amber_mpn_project_workspace_bytes(k, M, V))
    return fail(AMBER_ERR_WORKSPACE_TOO_SMALL, "workspace %zu < %zu bytes", workspace_bytes, amber_mpn_project_workspace_bytes(k, M, V));
  (void)hipGetLastError();
  HIP_TRY(launch_project_elements(simplices, k, M, V, element_values, volumes, out, status, aligned_ws(workspace),
                                  static_cast<hipStream_t>(stream)),
          "project_elements_to_vertices");
  return AMBER_OK;
}

size_t amber_mpn_estimate_workspace_bytes(int64_t V, int64_t Me) {
  if (V < 0 || Me < 0) return 0;
  return estimate_workspace_bytes(V, Me) + 256;
}

int amber_mpn_estimate_num_elements(const int64_t* simplices, int32_t k, int64_t M, int64_t V, const double* volumes,
                                    const double* sizing_field, int32_t vertex_field, int32_t dim, const int64_t* edges, int64_t Me,
                                    const double* positions, double gradation, double* out, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  if (dim != 2 && dim != 3) return fail(AMBER_ERR_UNSUPPORTED, "Dimension %d not supported", dim);
  if (k < 1 || k > 8 || M < 0 || V < 0 || Me < 0) return fail(AMBER_ERR_INVALID_ARGUMENT, "bad sizes");
  if (!out || !workspace) return fail(AMBER_ERR_INVALID_ARGUMENT, "out / workspace is NULL");
  if (M > 0 && (!volumes || !sizing_field)) return fail(AMBER_ERR_INVALID_ARGUMENT, "volumes / sizing_field is NULL");
  if (vertex_field) {
    if (M > 0 && !simplices) return fail(AMBER_ERR_INVALID_ARGUMENT, "simplices is NULL");
    if (V > 0 && !sizing_field) return fail(AMBER_ERR_INVALID_ARGUMENT, "sizing_field is NULL");
    if (Me > 0 && (!edges || !positions)) return fail(AMBER_ERR_INVALID_ARGUMENT, "edges / positions is NULL");
  }
  if (workspace_bytes < amber_mpn_estimate_workspace_bytes(V, Me)) return fail(AMBER_ERR_WORKSPACE_TOO_SMALL, "workspace too small");
  (void)hipGetLastError();
  HIP_TRY(launch_estimate_elements(simplices, k, M, V, volumes, sizing_field, vertex_field ? 1 : 0, dim, edges, vertex_field ? Me : 0,
                                   positions, gradation, out, aligned_ws(workspace), static_cast<hipStream_t>(stream)),
          "estimate_num_elements");
  return AMBER_OK;
}

}  // extern "C"
