// Internal C++ interface between the host driver (api.hip) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/amber_mpn.h"

namespace amber {

constexpr int kTile = AMBER_MPN_TILE;  // sorted edges (or nodes) per wave tile
constexpr int kMaxLayers = AMBER_MPN_MAX_MLP_LAYERS;

// One dense layer of a chain: packed L x L weight (MFMA operand order), bias, optional post-activation norm.
struct DenseLayer {
  const float* w;     // packed, L*L floats
  const float* b;     // [L]
  const float* p;     // norm parameter 0 (gamma / scale) or nullptr
  const float* q;     // norm parameter 1 (beta / shift) or nullptr
};

// Arguments of the fused edge kernel (K2-K6): one MessagePassingEdgeModule + residual + LayerNorm +
// segmented sum over destinations.
struct EdgeArgs {
  int64_t E;
  int64_t N;
  const int32_t* src;     // [E] sorted
  const int32_t* dst;     // [E] sorted
  const int32_t* rowptr;  // [N+1]
  const int32_t* perm;    // [E] or nullptr (only for out_edge)
  float* e;               // [E,L] latent edge features, CSR order, updated in place
  float* e_out;           // [E,L] where the updated rows go instead (training tape), or nullptr: in place
  const float* Ps;        // [N,L]  W0[:, 0:L] x          (source part of the first layer)
  const float* Pd;        // [N,L]  W0[:, L:2L] x + b0    (destination part + bias)
  const float* w0e;       // packed W0[:, 2L:3L]
  DenseLayer layers[kMaxLayers];  // layers[1..nl-1]
  int32_t nl;
  int32_t act;
  int32_t residual;       // AMBER_MODE_*
  int32_t layer_norm;     // AMBER_MODE_*
  const float* gamma;
  const float* beta;
  float eps;
  float* agg;             // [N,L] sums of rows that lie inside one tile
  float* head;            // [T,L] partial sum of a tile's first row when that row began in an earlier tile
  float* tail;            // [T,L] partial sum of a tile's last row when it began here and continues
  int32_t red_shift;      // log2 of the reduction tile the partial sums are indexed by (5 = 32 edges)
  int32_t write_e;        // store the updated e (CSR order)
  float* out_edge;        // [E,L] edge_index order, or nullptr
  int32_t dbg;            // experiment switches (AMBER_MPN_DBG), 0 in production
  // first step with the linear edge embedding fused in (K1): e is not read, its rows are computed on the fly as
  // emb_b + emb_w * edge_attr[perm[p]]; emb_attr == nullptr: e holds the embedded rows as usual
  const float* emb_attr;  // [E, emb_f] raw edge features, edge_index order
  const float* emb_w;     // [L, emb_f]
  const float* emb_b;     // [L]
  const float* emb_w0;    // [L, 3L] fp32 first-layer weight of this step's edge MLP: the fused step folds its e block with the
                          // embedding, W0e (emb_w attr + emb_b) = (W0e emb_w) attr + W0e emb_b, so layer 0 needs no contraction
  int32_t emb_f;          // 0 .. kMaxFusedEmbed
  int32_t emb_on;
  int32_t single;         // AMBER_PREC_F16: plain fp16 operands (one MFMA a product) instead of the hi/lo split
};
constexpr int kMaxFusedEmbed = 4;

// Post-processing of the head's output inside the readout epilogue (or stand-alone): RunningNormalizer
// ._denormalize, PredictionTransform.forward, _clip_detach, damping (see mpn_post.h).  on == 0: the raw network output.
struct PostArgs {
  int32_t on;
  int32_t denorm;           // prediction normaliser present
  int32_t transform;        // AMBER_TRANSFORM_*
  int32_t residual;         // predict_residual: add transform^-1(baseline) before the transform
  int32_t apply_transform;  // 0: training with inverse_transform_in_loss (the loss sees the denormalised raw output)
  int32_t clip;             // bit 0: lower bound, bit 1: upper bound
  const double* mean;       // [1] device
  const double* var;        // [1] device
  double eps;
  const float* baseline;    // [N_m] current sizing field, indexed like the output
  float clip_min, clip_max, scale0, scale1;
};

// Arguments of the fused node kernel (K6 finish, K7, K8, next step's first-layer projections, K9, K10).
struct NodeArgs {
  int64_t N;
  const float* x_in;      // [N,L]
  float* x_out;           // [N,L] or nullptr (may alias x_in)
  // update phase (MessagePassingNodeModule + residual + LayerNorm)
  int32_t do_update;
  const int32_t* rowptr;
  const float* agg;
  const float* head;
  const float* tail;
  int32_t red_shift;      // log2 of the reduction tile of head / tail (must match the edge kernel that wrote them)
  int32_t fast;           // update / projection weights are in the split-precision B-operand format (node_kernel_f2)
  const float* w0x;       // packed W0[:, 0:L]
  const float* w0a;       // packed W0[:, L:2L]
  const float* b0;
  DenseLayer layers[kMaxLayers];  // layers[1..nl-1]
  int32_t nl;
  int32_t act;
  int32_t residual;       // AMBER_MODE_*: add x_in before (inner) or after (outer) the inner LayerNorm
  int32_t layer_norm;     // AMBER_MODE_*
  const float* gamma;
  const float* beta;
  float eps;
  // projections for the next edge step
  int32_t do_proj;
  const float* w0s;       // packed next-step edge W0[:, 0:L]
  const float* w0d;       // packed next-step edge W0[:, L:2L]
  const float* b0e;       // next-step edge b0
  float* Ps;
  float* Pd;
  // dense chain (decoder MLP or embedding MLP tail)
  int32_t n_dense;
  DenseLayer dense[kMaxLayers];
  const float* dense_f2[kMaxLayers];  // the same matrices in the split-precision B-operand format, or nullptr (node_kernel_f2)
  int32_t dense_act;
  int32_t dense_norm;     // AMBER_NORM_*
  float* dense_out;       // [N,L] or nullptr
  // readout Linear(L,1)
  int32_t do_readout;
  const float* wr;        // [L]
  const float* br;        // [1]
  const int32_t* out_index;  // [N] or nullptr
  float* out;
  PostArgs post;          // applied to the readout value before it is stored
  int32_t dbg;            // experiment switches (AMBER_MPN_DBG), 0 in production
  int32_t single;         // AMBER_PREC_F16, as in EdgeArgs
};

// b3 = 1: the packed matrices are bf16x3 pieces (AMBER_PACK_B3) and the contractions run on the fp16-rate pipe; built for
// latent 256, which the split-precision feature-per-lane kernels do not take, and for the configurations they do not take at 128
constexpr bool fwd_b3_supported(int L) { return L == 128 || L == 256; }
hipError_t launch_edge(int L, int b3, const EdgeArgs& a, hipStream_t s);
hipError_t launch_node(int L, int b3, const NodeArgs& a, hipStream_t s);
// out[r] = act(W in[idx(r)] + b): small-K linear (K = F >= 0), idx = perm gather or identity
hipError_t launch_embed(int L, const float* in, int64_t rows, int F, const float* W, const float* b, const int32_t* perm,
                        int act_or_neg1, float* out, hipStream_t s);
// x = A raw + b, P_s = W0s x, P_d = W0d x + b0 from the raw node features through tables [3][F + 1][L] (see embed_proj_kernel)
constexpr int kMaxFoldNodeFeatures = 16;
hipError_t launch_embed_proj(int L, const float* in, int64_t rows, int F, const float* tables, float* out_x, float* out_ps,
                             float* out_pd, hipStream_t s);
hipError_t launch_fold_node_tables(int L, int F, const float* A, const float* b, const float* W0, const float* b0, float* tables,
                                   hipStream_t s);
// re-tile W[L, ld] columns [col0, col0+L) into MFMA operand order
hipError_t launch_pack(int L, const float* W, int ld, int col0, float* out, hipStream_t s);
// A packed matrix is L*L floats of body + kPackedHeader floats (split-precision formats keep their scales there).
constexpr int kPackedHeader = 8;
// feature-per-lane split-precision edge kernel, mpn_f2.hip (weights as B operand: launch_pack_f2)
constexpr int kF2PackMaxJobs = 24;
struct F2PackJob {
  const float* W;   // [L, ld]
  float* out;       // packed matrix + header
  int32_t ld, col0, scale_col0, scale_cols;
  int32_t kperm;    // 1: the contraction's input arrives from accumulator registers (k order permuted like the columns)
};
struct F2PackBatch {
  F2PackJob job[kF2PackMaxJobs];
  int32_t n;
};
hipError_t launch_pack_f2_batch(int L, const F2PackBatch& b, hipStream_t s);
hipError_t launch_pack_f2(int L, const float* W, int ld, int col0, int scale_col0, int scale_cols, int kperm, float* out, hipStream_t s);
hipError_t launch_edge_f2(int L, const EdgeArgs& a, int num_cus, hipStream_t s);
bool edge_f2_supported(int L, int nl, int act, int residual, int layer_norm);
// feature-per-lane node kernel (update + next-step projections; weights as B operand, W0x / W0a sharing one scale)
hipError_t launch_node_f2(int L, const NodeArgs& a, int num_cus, hipStream_t s);
bool node_f2_supported(int L, int act, int residual, int layer_norm);
// the split-precision node kernel can run the SupervisedMPN head (decoder layers as further phases + readout)
bool head_f2_supported(int n_dense, int dense_act, int dense_norm);
hipError_t launch_segment_mean(const float* vals, const int32_t* rowptr, int64_t N, int L, float* out, hipStream_t s);
hipError_t launch_permute_rows(const float* in, const int32_t* perm, int64_t rows, int width, int gather, float* out,
                               hipStream_t s);

// ---- backward (training), mpn_bwd.hip ----
enum { AMBER_BWD_EDGE = 0, AMBER_BWD_NODE = 1, AMBER_BWD_HEAD = 2 };
constexpr int kBwdMaxBlocks = 512;     // persistent grid of the chain kernel (4 waves a block, one partial per block)
constexpr int kBwdWavesPerBlock = 4;   // = kWavesPerBlock of mpn_tile.h (checked in mpn_bwd.hip)
constexpr int kWgradMaxChunks = 512;   // split-K chunks of a weight-gradient product
// Recompute + chain rule of one module on 32-row tiles (see chain_bwd_kernel).  All matrices in the exact fp32 operand
// order (launch_pack / launch_pack_t).
// Operand format of the backward's contractions: exact fp32 (v_mfma_f32_32x32x2_f32) or every float as three bf16 pieces
// on the fp16-rate pipe (bf16x3, see split3 in mpn_bwd.hip: fp32-grade, 2.7 times fewer matrix-pipe cycles).
constexpr int AMBER_PACK_F32 = 0, AMBER_PACK_B3 = 1;
constexpr bool bwd_split_supported(int L) { return L == 32 || L == 64 || L == 128; }
// floats of one packed matrix of the backward in either format (bf16x3: three 2-byte pieces a value)
constexpr size_t bwd_mat_floats(int L) { return static_cast<size_t>(L) * L * 3 / 2 + kPackedHeader; }
struct ChainBwdArgs {
  int32_t mode;           // AMBER_BWD_*
  int32_t split;          // 0: matrices packed AMBER_PACK_F32; 1: AMBER_PACK_B3
  int64_t rows;           // E (edge) or N (node, head)
  const float* rin;       // [rows,L] streamed input of the first layer and residual term: e_in / x_in
  const float* Ps;        // edge: [N,L] source part of the first layer
  const float* Pd;        // edge: [N,L] destination part + bias
  const int32_t* src;     // edge: [E] sorted
  const int32_t* dst;     // edge: [E] sorted
  const int32_t* rowptr;  // edge: [N+1]
  const float* m;         // node: [N,L] mean of the incoming (updated) edge features
  const float* w_in;      // first-layer matrix applied to rin (W0e / W0x / decoder W0)
  const float* w_m;       // node: W0a
  const float* b0;        // node, head: first-layer bias (edge: inside Pd)
  DenseLayer layers[kMaxLayers];  // layers[1..nl-1]
  const float* wT_in;     // transposes of the above
  const float* wT_m;
  const float* wT[kMaxLayers];
  int32_t nl;             // layers of the chain (head: may be 0)
  int32_t act;
  int32_t residual;       // AMBER_MODE_*; node rows: OUTER only together with layer_norm INNER (the host folds the rest)
  int32_t layer_norm;     // AMBER_MODE_*; node rows: NONE / INNER
  const float* gamma;
  float eps;
  // head with a LayerNorm behind every decoder layer (latent_normalization "layer"), else pn_gamma[0] == nullptr
  const float* pn_gamma[kMaxLayers];
  const float* pn_beta[kMaxLayers];
  float* AP[kMaxLayers];  // [rows,L] activation of layer l before its LayerNorm
  // head with an activation whose derivative needs the pre-activation (silu, mish, gelu): [rows,L] W_l in_l + b_l, else nullptr
  float* PRE[kMaxLayers];
  float* pn_part;         // [grid waves][nl][2][L] (d gamma_l, d beta_l) sums, zeroed by the caller
  float* dio;             // [rows,L] in: gradient at the module's output; out: gradient at rin (head: added to)
  const float* dm_in;     // edge: [N,L] gradient at the mean aggregate (divided by the in-degree here)
  const float* dout;      // head: [N_m] gradient at the readout
  const int32_t* out_index;  // head: [N] or nullptr
  const float* wr;        // head: readout weight [L]
  float* dm_out;          // node: [N,L] gradient at the mean aggregate
  float* A[kMaxLayers];   // [rows,L] output of layer l (l < nl-1)
  float* DH[kMaxLayers];  // [rows,L] gradient at the pre-activation of layer l
  float* part;            // [grid waves][2][L] partial column sums: LayerNorm (d gamma, d beta) or readout (d wr, d br)
};
int chain_bwd_grid(int64_t rows);
// chain + weight gradients of the edge module in one kernel (edge_bwd_fused_kernel): one workgroup per CU
constexpr int kFusedBwdBlocks = 256;
bool edge_bwd_fused_supported(int L, const ChainBwdArgs& a);
int edge_bwd_fused_grid(int64_t rows);
// wpart: [2][grid][L*L] partial products (dW0e, dW1), cpart: [2][grid][L] (slot 1: db1), a.part: [grid][2][L] LayerNorm sums
hipError_t launch_edge_bwd_fused(int L, const ChainBwdArgs& a, float* wpart, float* cpart, hipStream_t s);
// the same for the node module: wpart [3][grid][L*L] (dW0x, dW0a, dW1), cpart [3][grid][L] (slot 0: db0, slot 2: db1)
constexpr int kFusedMaxProducts = 3;
bool node_bwd_fused_supported(int L, const ChainBwdArgs& a);
hipError_t launch_node_bwd_fused(int L, const ChainBwdArgs& a, float* wpart, float* cpart, hipStream_t s);
hipError_t launch_chain_bwd(int L, const ChainBwdArgs& a, hipStream_t s);
hipError_t launch_colpart_reduce(const float* part, int G, int L, float* out0, int n0, float* out1, int n1, hipStream_t s);
// the same over partials that lie gstride floats apart (per-layer slots of pn_part)
hipError_t launch_colpart_reduce_strided(const float* part, int G, int64_t gstride, int L, float* out0, float* out1, hipStream_t s);
// d[i] *= act'(.) expressed through the activation's output a[i] (first layer of an embedding MLP)
hipError_t launch_mul_dact(float* d, const float* a, int64_t n, int act, hipStream_t s);
// out[n] = mean of the incoming edge rows from the edge kernel's partial sums (agg / head / tail), as the node kernel forms it
hipError_t launch_partials_mean(const float* agg, const float* head, const float* tail, const int32_t* rowptr, int red_shift,
                                int64_t N, int L, float* out, hipStream_t s);
// weight-gradient products over the same rows, one launch (the matrices of one module)
constexpr int kWgradMaxJobs = 6;
struct WgradJob {
  const float* A;   // [M,L] gradient at the layer's pre-activation
  const float* B;   // [M,L] the layer's input
  float* dW;        // dW[:, col0:col0+L] += A^T B, leading dimension ld
  float* db;        // db += column sums of A, or nullptr
  int32_t ld, col0;
};
struct WgradBatch {
  WgradJob job[kWgradMaxJobs];
  int32_t n;
  int32_t split;  // 1: bf16x3 operands
  int64_t M;
};
int wgrad_chunks(int64_t M, int njobs, int64_t* chunk_out);
// the ordered reduction of G partial products per job alone (partials laid out [job][G][L*L], [job][G][L])
hipError_t launch_wgrad_reduce(int L, const WgradBatch& b, const float* part, const float* cpart, int G, hipStream_t s);
hipError_t launch_wgrad_batch(int L, const WgradBatch& b, float* part, float* cpart, hipStream_t s);
// dW[:, col0:col0+L] += A^T B (A, B: [M,L]); db += column sums of A (db may be nullptr)
hipError_t launch_wgrad(int L, int split, const float* A, const float* B, int64_t M, float* dW, int ld, int col0, float* db,
                        float* part, float* cpart, hipStream_t s);
constexpr int kPackMaxJobs = 16;
struct PackJob {
  const float* W;
  float* out;
  int32_t ld, col0, transpose;
  int32_t fmt;  // AMBER_PACK_F32 / AMBER_PACK_B3
};
struct PackBatch {
  PackJob job[kPackMaxJobs];
  int32_t n;
};
hipError_t launch_pack_batch(int L, const PackBatch& b, hipStream_t s);
int embed_bwd_chunks(int64_t rows, int64_t* chunk_out);
hipError_t launch_embed_bwd(int L, const float* d, const float* raw, const int32_t* perm, int64_t rows, int F, float* dW, float* db,
                            float* part, hipStream_t s);
hipError_t launch_segment_reduce(const float* vals, const int32_t* rowptr, const int32_t* idx, int64_t N, int L, int mean, float* out,
                                 hipStream_t s);
hipError_t launch_rows_gemm_add(int L, int split, float* io, const float* in0, const float* w0, const float* in1, const float* w1,
                                int64_t rows, hipStream_t s);
hipError_t launch_pack_t(int L, const float* W, int ld, int col0, float* out, hipStream_t s);

// ---- pre / post glue of _predict, glue.hip ----
hipError_t launch_normalize_rows(const float* in, int64_t rows, int width, const double* mean, const double* var, double eps,
                                 double clip, float* out, hipStream_t s);
hipError_t launch_inverse_transform(int transform, const float* in, const float* baseline, int64_t n, float* out, hipStream_t s);
hipError_t launch_postprocess(const PostArgs& p, const float* pred, int64_t n, float* out, hipStream_t s);
hipError_t launch_postprocess_bwd(const PostArgs& p, const float* pred, const float* grad_out, int64_t n, float* grad_pred,
                                  hipStream_t s);

// ---- graph construction, graph.hip ----
struct GraphArgs {
  int32_t G, dim, add_self_edges, distance_vector, euclidean_distance, edge_curvature, F;
  int64_t N, M, E;             // nodes, undirected pairs, row stride of edge_index (= edges of the whole output)
  int64_t eoff;                // column of the output at which this call's edges begin
  const int64_t* node_ptr;     // [G+1]
  const int64_t* edge_ptr;     // [G+1]
  const int64_t* nbr_a;        // [M] local ids
  const int64_t* nbr_b;        // [M]
  const float* positions;      // [N, dim] or nullptr
  const float* curvature;      // [M] or nullptr
  int64_t* edge_index;         // [2, E]
  float* edge_attr;            // [E, F]
  int64_t* batch;              // [N] or nullptr
  int32_t* degree_i;           // [N] or nullptr: int32 counts, converted to float32 in place
  int32_t* status;
};
hipError_t launch_build_graph(const GraphArgs& a, hipStream_t s);
hipError_t launch_inter_edges(const GraphArgs& a, const int64_t* nearest, const double* pos_src, const double* pos_dst, int64_t n_src,
                              int64_t n_dst, int64_t index_offset, hipStream_t s);

// ---- the step after the model: element -> vertex projection, element-count estimate (mesh.hip) ----
size_t project_workspace_bytes(int k, int64_t M, int64_t V);
hipError_t launch_project_elements(const int64_t* t, int k, int64_t M, int64_t V, const double* val, const double* vol, double* out,
                                   int32_t* status, void* ws, hipStream_t s);
size_t estimate_workspace_bytes(int64_t V, int64_t Me);
hipError_t launch_estimate_elements(const int64_t* t, int k, int64_t M, int64_t V, const double* vol, const double* field,
                                    int vertex_field, int dim, const int64_t* edges, int64_t Me, const double* pos,
                                    double gradation, double* out, void* ws, hipStream_t s);

size_t csr_workspace_bytes(int64_t N, int64_t E);
hipError_t csr_build(const int64_t* edge_index, int64_t E, int64_t N, int32_t* rowptr, int32_t* src_sorted,
                     int32_t* dst_sorted, int32_t* perm, int32_t* status, void* ws, hipStream_t s);
// this GPU's copy of the file-scope scratch array of csr.hip (AMBER_MPN_CSR_STATIC_SCRATCH_BYTES), 256-byte aligned
hipError_t csr_static_scratch(void** out);

}  // namespace amber
