/*
 * amber_mpn.h -- C ABI of libamber_mpn.so: AMBER's message-passing-network forward path on MI355X (gfx950).
 *
 * The reference (NiklasFreymuth/AMBER) is 100 % Python and has no FFI of its own; each entry point below
 * replaces the torch / torch_scatter code of the reference function it cites (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer unless a comment says "host".  The caller owns every buffer
 *     (inputs, outputs, workspace); the library allocates nothing and keeps no state between calls.
 *   - All floating point data is fp32, row-major; indices handed in are int64 (torch edge_index), the
 *     CSR the library builds and consumes is int32.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).  All work is enqueued on it; no
 *     call synchronises the device.
 *   - Return value: 0 on success, negative AMBER_ERR_* otherwise; amber_mpn_last_error() returns a
 *     thread-local message for the last failing call.
 */
#ifndef AMBER_MPN_H
#define AMBER_MPN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMBER_MPN_ABI_VERSION 2
#define AMBER_MPN_MAX_MLP_LAYERS 4
#define AMBER_MPN_TILE 32 /* sorted edges (or nodes) per wave tile */

enum {
  AMBER_OK = 0,
  AMBER_ERR_INVALID_ARGUMENT = -1,
  AMBER_ERR_UNSUPPORTED = -2,
  AMBER_ERR_WORKSPACE_TOO_SMALL = -3,
  AMBER_ERR_HIP = -4
};

/* activation functions: LatentMLP (src/mpn/common/latent_mlp.py:76-83) uses the first three,
 * the decoder MLP (src/algorithm/architecture/mlp.py:68-83) all of them; torch default arguments. */
enum {
  AMBER_ACT_RELU = 0,
  AMBER_ACT_LEAKYRELU = 1, /* negative slope 0.01 */
  AMBER_ACT_TANH = 2,
  AMBER_ACT_ELU = 3,
  AMBER_ACT_SILU = 4,
  AMBER_ACT_MISH = 5,
  AMBER_ACT_GELU = 6 /* exact erf form */
};

/* residual_connections / layer_norm placement (src/mpn/message_passing_block.py:34-38,62-74) */
enum { AMBER_MODE_NONE = 0, AMBER_MODE_INNER = 1, AMBER_MODE_OUTER = 2 };

/* decoder latent_normalization (src/algorithm/architecture/mlp.py:86-89), eval mode */
enum { AMBER_NORM_NONE = 0, AMBER_NORM_LAYER = 1, AMBER_NORM_AFFINE = 2 /* BatchNorm1d folded to scale, shift */ };

/* precision of the MLP contractions */
enum {
  AMBER_PREC_FP32 = 0, /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32): a k-ordered fmaf chain */
  AMBER_PREC_F16X2 = 1, /* fp32-grade split precision, meets the fp32 tolerance (1e-5).  The edge and node kernels of the
                           stack (relu / leakyrelu / tanh, every residual / LayerNorm mode, 1..2 weight matrices that fit
                           LDS at the latent width; latent <= 128) and a relu / leakyrelu / tanh decoder without latent
                           normalisation run feature-per-lane on hi + lo fp16 operand pairs (22 bits, exact power-of-two
                           tile / matrix scaling, 3 fp16 MFMAs per product, fp32 accumulate).  Everything those kernels
                           do not take - latent 256, MLP embeddings, decoders with other activations or a latent
                           normalisation, the stand-alone amber_mpn_head, more hidden matrices than fit LDS - runs on the
                           row-per-lane kernels with bf16x3 operands (three bf16 pieces a float, six piece products,
                           <= 2^-22 a product) at latent 128 and 256, and as exact fp32 MFMA chains at latent 32 / 64. */
  AMBER_PREC_F16 = 2    /* reduced precision: the same kernels with plain fp16 operands (one MFMA per product, fp32
                           accumulate, fp32 storage of all features); tolerance 1e-2, the bound BASELINE's north star
                           gives for bf16-class MLPs.  Never the default. */
};

/* Static description of the network: the hyper-parameters the reference reads from
 * config/algorithm/architecture/mpn.yaml ("Config / flags"). */
typedef struct AmberMpnDesc {
  int32_t latent;             /* latent_dimension: 32, 64, 128, or 256 (256: row-per-lane kernels - exact fp32 for AMBER_PREC_FP32,
                                 bf16x3 operands otherwise, see AMBER_PREC_F16X2; backward at 256 exact fp32) */
  int32_t num_steps;          /* stack.num_steps: blocks with their own weights */
  int32_t num_step_repeats;   /* stack.num_step_repeats: each block applied this often */
  int32_t mlp_layers;         /* stack.mlp.num_layers, 1..AMBER_MPN_MAX_MLP_LAYERS */
  int32_t activation;         /* stack.mlp.activation_function */
  int32_t residual;           /* AMBER_MODE_* */
  int32_t layer_norm;         /* AMBER_MODE_* */
  int32_t in_node_features;   /* F_n >= 0 (0 = learned constant embedding, empty_linear_layer.py:44-45) */
  int32_t in_edge_features;   /* F_e >= 0 */
  int32_t embed_layers;       /* 0 = nn.Linear embedding; k > 0 = LatentMLP embedding with k layers (embedding.py:32-38) */
  int32_t embed_activation;   /* activation of the embedding MLP */
  int32_t has_head;           /* 1 = SupervisedMPN decoder MLP + readout Linear(L,1) follow (supervised_mpn.py:51-54) */
  int32_t decoder_layers;     /* decoder.num_layers, 0..AMBER_MPN_MAX_MLP_LAYERS */
  int32_t decoder_activation; /* decoder.activation_function */
  int32_t decoder_norm;       /* AMBER_NORM_* applied after each decoder activation */
  int32_t precision;          /* AMBER_PREC_* */
  float ln_eps;               /* LayerNorm eps (torch default 1e-5) */
} AmberMpnDesc;

/* ---------------------------------------------------------------------------------------------------
 * Parameters.  The caller concatenates the module's parameters, each in its nn.Linear / nn.LayerNorm
 * layout ([out, in] row-major weights), into ONE fp32 device buffer in this order:
 *
 *   node embedding, then edge embedding; each is
 *       F == 0            : bias[L]
 *       embed_layers == 0 : W[L,F], b[L]
 *       embed_layers  > 0 : W0[L,F], b0[L], then (embed_layers-1) x { W[L,L], b[L] }
 *   for every step s < num_steps:
 *       edge MLP : W0[L,3L], b0[L], then (mlp_layers-1) x { W[L,L], b[L] }      (columns: x[src] | x[dst] | e)
 *       node MLP : W0[L,2L], b0[L], then (mlp_layers-1) x { W[L,L], b[L] }      (columns: x | mean of incoming e)
 *       if layer_norm != NONE: edge LN gamma[L], beta[L], node LN gamma[L], beta[L]
 *   if has_head:
 *       decoder_layers x { W[L,L], b[L], and if decoder_norm != NONE: p[L], q[L] }
 *           (LAYER: p = gamma, q = beta;  AFFINE: p = scale, q = shift of the folded eval BatchNorm)
 *       readout w[L], b[1]
 *
 * amber_mpn_pack_weights re-tiles the L x L matrices into the MFMA operand order the kernels read
 * (`packed`, amber_mpn_packed_floats(desc) floats).  Both buffers are passed to every compute call.
 * ------------------------------------------------------------------------------------------------- */
int64_t amber_mpn_param_floats(const AmberMpnDesc* desc);
int64_t amber_mpn_packed_floats(const AmberMpnDesc* desc);
int amber_mpn_pack_weights(const AmberMpnDesc* desc, const float* params, float* packed, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * CSR construction: destination-sorted, stable (inside a destination's row the edges keep their
 * edge_index order, i.e. the order torch_scatter's scatter_add visits them).  Built once per mesh.
 * Replaces the implicit per-call index handling of scatter_mean(edge_attr, dest_indices, ...)
 * (src/mpn/message_passing_modules.py:46-48,67-68).  Bit-exact vs torch.sort(dst, stable=True).
 *   edge_index : int64 [2, E], row 0 = source, row 1 = destination
 *   rowptr[N+1], src_sorted[E], dst_sorted[E], perm[E] (perm[p] = original edge id at sorted position p)
 *   status     : int32[1], set to 0, or 1 if an index is outside [0, N) (offending edges are clamped)
 * ------------------------------------------------------------------------------------------------- */
size_t amber_mpn_csr_workspace_bytes(int64_t num_nodes, int64_t num_edges);
int amber_mpn_csr_build(const int64_t* edge_index, int64_t num_edges, int64_t num_nodes, int32_t* rowptr,
                        int32_t* src_sorted, int32_t* dst_sorted, int32_t* perm, int32_t* status, void* workspace,
                        size_t workspace_bytes, void* stream);

/* amber_mpn_csr_build without a workspace argument and without any allocation.  No counterpart in the reference: it is
 * for callers that cannot allocate device memory around the call (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  The count / cursor / scan / slot areas live in a file-scope device array of
 * AMBER_MPN_CSR_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends; graphs with
 * amber_mpn_csr_workspace_bytes(num_nodes, num_edges) - 256 above that are rejected (AMBER_ERR_WORKSPACE_TOO_SMALL).
 * Same outputs as amber_mpn_csr_build, bit for bit.  All calls on one GPU share that GPU's array: calls that overlap
 * on one GPU (different streams or threads) race; calls on one stream are ordered and each clears what it needs. */
#define AMBER_MPN_CSR_STATIC_SCRATCH_BYTES (16u << 20)
int amber_mpn_csr_build_static(const int64_t* edge_index, int64_t num_edges, int64_t num_nodes, int32_t* rowptr,
                               int32_t* src_sorted, int32_t* dst_sorted, int32_t* perm, int32_t* status, void* stream);

/* Deterministic segmented mean over CSR rows, no float atomics: out[n] = sum(values[rowptr[n]:rowptr[n+1]]) /
 * max(deg, 1), summed in row order.  Replaces torch_scatter.scatter_mean at
 * src/mpn/message_passing_modules.py:68 (values already destination-sorted).  L % 4 == 0, L <= 1024. */
int amber_mpn_segment_mean(const float* values_sorted, const int32_t* rowptr, int64_t num_nodes, int64_t num_edges,
                           int32_t latent, float* out, void* stream);

/* Row gather / scatter between edge_index order and CSR order: out[p] = in[perm[p]] (gather = 1) or
 * out[perm[p]] = in[p] (gather = 0); rows of `width` floats (width % 4 == 0). */
int amber_mpn_permute_rows(const float* in, const int32_t* perm, int64_t rows, int32_t width, int32_t gather,
                           float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Stage entry points (each mirrors one reference module; used for per-stage parity tests and by hosts
 * that drive the stack themselves).  `workspace` must hold amber_mpn_workspace_bytes(desc, N, E) bytes.
 * ------------------------------------------------------------------------------------------------- */
size_t amber_mpn_workspace_bytes(const AmberMpnDesc* desc, int64_t num_nodes, int64_t num_edges);

/* MessagePassingInputEmbedding.forward (src/mpn/input_embedding.py:33-42; Embedding, EmptyLinear):
 * x_latent[N,L] = embed(x_raw[N,F_n]);  e_latent_sorted[E,L] = embed(edge_attr_raw[perm][E,F_e]). */
int amber_mpn_embed(const AmberMpnDesc* desc, const float* params, const float* packed, int64_t num_nodes,
                    int64_t num_edges, const float* x_raw, const float* edge_attr_raw, const int32_t* perm,
                    float* x_latent, float* e_latent_sorted, void* workspace, size_t workspace_bytes, void* stream);

/* MessagePassingBlock.forward for block `step` (src/mpn/message_passing_block.py:107-128; edge module
 * message_passing_modules.py:38-52, node module :61-72), in place on latent features:
 * x_latent[N,L] and e_latent_sorted[E,L] (CSR order). */
int amber_mpn_block(const AmberMpnDesc* desc, const float* params, const float* packed, int32_t step,
                    int64_t num_nodes, int64_t num_edges, const int32_t* rowptr, const int32_t* src_sorted,
                    const int32_t* dst_sorted, float* x_latent, float* e_latent_sorted, void* workspace,
                    size_t workspace_bytes, void* stream);

/* First-layer projections of the edge MLP of block `step` (the weight split of
 * MessagePassingEdgeModule.forward, src/mpn/message_passing_modules.py:46-52: W0 [x_s | x_d | e] = W0s x_s + W0d x_d + W0e e):
 * p_src[N,L] = W0[:, 0:L] x,  p_dst[N,L] = W0[:, L:2L] x + b0.  The block and forward entry points compute them
 * internally; exposed for per-stage parity tests. */
int amber_mpn_project(const AmberMpnDesc* desc, const float* params, const float* packed, int32_t step, int64_t num_nodes,
                      const float* x_latent, float* p_src, float* p_dst, void* stream);

/* SupervisedMPN head (src/algorithm/architecture/supervised_mpn.py:51-54; MLP mlp.py:60-95):
 * out_sizing[out_index[n]] = readout(decoder(x_latent[n])) for every n with out_index[n] >= 0
 * (out_index == NULL: identity).  out_index encodes x[mask_output]: the rank of n among kept nodes. */
int amber_mpn_head(const AmberMpnDesc* desc, const float* params, const float* packed, int64_t num_nodes,
                   const float* x_latent, const int32_t* out_index, float* out_sizing, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * The whole path in one call: MessagePassingBase.forward (src/mpn/message_passing_base.py:127-146, eval mode:
 * edge dropout is the identity, src/mpn/common/edge_dropout.py:24) followed, if desc->has_head, by the
 * SupervisedMPN head.  Inputs are never written (the create_graph_copy guarantee, mpn_util.py:82-83).
 *   x[N,F_n], edge_attr[E,F_e] (edge_index order), CSR from amber_mpn_csr_build
 *   out_node  [N,L]  or NULL : final node latents  ({"node": x} of unpack_features, mpn_util.py:22-46)
 *   out_edge  [E,L]  or NULL : final edge latents in edge_index order
 *   out_sizing[N_m]  or NULL : head output (requires has_head); out_index as in amber_mpn_head
 * ------------------------------------------------------------------------------------------------- */
int amber_mpn_forward(const AmberMpnDesc* desc, const float* params, const float* packed, int64_t num_nodes,
                      int64_t num_edges, const float* x, const float* edge_attr, const int32_t* rowptr,
                      const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* perm,
                      const int32_t* out_index, float* out_node, float* out_edge, float* out_sizing,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Timing hook for benchmarks: like amber_mpn_forward, and additionally records HIP events around every
 * kernel launch ON `stream` and, after synchronising it, writes per-kernel-class device milliseconds to
 * the HOST array kernel_ms[AMBER_MPN_NUM_KERNEL_CLASSES] (accumulated over the call).  Synchronises. */
#define AMBER_MPN_NUM_KERNEL_CLASSES 6
/* EDGE / NODE hold every launch of those kernels; EDGE_FIRST (the first step's launch, which computes the edge embedding
 * on the fly when it is fused) and NODE_LAST (the last launch, which carries the decoder + readout when the head is fused)
 * repeat the two launches that differ from the steady-state variant, so a caller can separate them. */
enum { AMBER_KC_EMBED = 0, AMBER_KC_EDGE = 1, AMBER_KC_NODE = 2, AMBER_KC_OTHER = 3, AMBER_KC_EDGE_FIRST = 4, AMBER_KC_NODE_LAST = 5 };
int amber_mpn_forward_timed(const AmberMpnDesc* desc, const float* params, const float* packed, int64_t num_nodes,
                            int64_t num_edges, const float* x, const float* edge_attr, const int32_t* rowptr,
                            const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* perm,
                            const int32_t* out_index, float* out_node, float* out_edge, float* out_sizing,
                            void* workspace, size_t workspace_bytes, void* stream, float* kernel_ms,
                            int32_t* kernel_launches);

/* ---------------------------------------------------------------------------------------------------
 * Training (autograd through the path; the reference's training step: src/algorithm/core/amber.py:62-82 calls
 * loss.backward() on the output of SupervisedMPN.forward).
 *
 * amber_mpn_forward_train = amber_mpn_forward that additionally keeps the block inputs of every step on a
 * caller-owned TAPE (x_t[N,L], e_t[E,L] in CSR order, t = 0 .. num_steps * num_step_repeats, and the mean aggregate
 * m_t[N,L] of every step as the forward formed it; amber_mpn_tape_bytes).  amber_mpn_backward recomputes each module from its tape entry (activation checkpointing
 * per block) and writes the gradient of every parameter, in the layout of `params`, to grad_params
 * (amber_mpn_param_floats floats, overwritten).  Gradients arrive as any of
 *   grad_sizing[N_m] (needs has_head), grad_node[N,L], grad_edge[E,L] (edge_index order); NULL = zero.
 * Inputs (x, edge_attr) get no gradient: they are data, not parameters (nothing upstream of the MPN trains).
 * rowptr_src / perm_src: the SOURCE-sorted order of the CSR positions, i.e. amber_mpn_csr_build applied to the
 * edge list [dst_sorted ; src_sorted] (its rowptr and perm outputs) -- used for the deterministic per-source sums.
 * Edge dropout (src/mpn/common/edge_dropout.py:24-25) happens before the call: the caller passes the kept edges.
 * Arithmetic of the backward's contractions: exact fp32 MFMA for AMBER_PREC_FP32 and at latent 256 (correct and slow there);
 * otherwise every float as three bf16 pieces on the fp16-rate pipe (six piece products, relative error <= 2^-22).
 * Built for: every latent width of the forward; residual / layer_norm none, inner or outer (message_passing_block.py:62-74,91-97);
 * constant, linear or LatentMLP embeddings (embedding.py:32-38); every decoder activation of the forward (silu / mish /
 * gelu: the head's chain keeps their pre-activations in the workspace) with latent_normalization none or layer
 * (mlp.py:86-93).  Decoder BatchNorm returns AMBER_ERR_UNSUPPORTED.
 * ------------------------------------------------------------------------------------------------- */
size_t amber_mpn_tape_bytes(const AmberMpnDesc* desc, int64_t num_nodes, int64_t num_edges);
int amber_mpn_forward_train(const AmberMpnDesc* desc, const float* params, const float* packed, int64_t num_nodes,
                            int64_t num_edges, const float* x, const float* edge_attr, const int32_t* rowptr,
                            const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* perm,
                            const int32_t* out_index, float* out_node, float* out_edge, float* out_sizing, void* tape,
                            size_t tape_bytes, void* workspace, size_t workspace_bytes, void* stream);
size_t amber_mpn_backward_workspace_bytes(const AmberMpnDesc* desc, int64_t num_nodes, int64_t num_edges);
int amber_mpn_backward(const AmberMpnDesc* desc, const float* params, const float* packed, int64_t num_nodes,
                       int64_t num_edges, const float* x, const float* edge_attr, const int32_t* rowptr,
                       const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* perm,
                       const int32_t* rowptr_src, const int32_t* perm_src, const int32_t* out_index,
                       const float* grad_node, const float* grad_edge, const float* grad_sizing, void* tape,
                       size_t tape_bytes, float* grad_params, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Pre / post numeric glue of the model call: what MeshGenerationAlgorithm._predict
 * (src/algorithm/core/mesh_generation_algorithm.py:343-367) does either side of self.model(batch), and the clip + damping
 * of the predicted sizing field that follows it in the inference step (_clip_detach :369-371, Amber._clip_detach_dampen
 * src/algorithm/core/amber.py:213-246).
 *
 * Running statistics are the float64 buffers of TorchRunningMeanStd (src/algorithm/normalizer/torch_running_mean_std.py:
 * 23-25), passed as DEVICE pointers (no host read of them is needed).  Integer/bit behaviour: the normalised inputs are
 * bit-identical to the reference's (same IEEE float64 operations, one rounding to float32); the post-processing uses
 * float32 exp / log / expm1 / log1p, equal to the reference's within 1e-6 relative.
 * ------------------------------------------------------------------------------------------------- */
enum { AMBER_TRANSFORM_NONE = 0, AMBER_TRANSFORM_SOFTPLUS = 1, AMBER_TRANSFORM_EXP = 2 };

typedef struct AmberGlueDesc {
  int32_t normalize_inputs;        /* node / edge normalisers present (graph_running_normalizer.py:60-67) */
  int32_t denormalize_predictions; /* prediction normaliser present (running_normalizer.py:35-38, :82-85) */
  int32_t transform;               /* AMBER_TRANSFORM_* (src/algorithm/prediction_transform/__init__.py:6-27) */
  int32_t predict_residual;        /* prediction + transform^-1(baseline) before the transform (prediction_transform.py:78-86) */
  int32_t apply_transform;         /* 0 = is_train with inverse_transform_in_loss: raw output (prediction_transform.py:75-76) */
  int32_t clip;                    /* bit 0: clip_min set, bit 1: clip_max set (np.clip with None bounds) */
  float clip_min, clip_max;        /* gmsh_kwargs min_sizing_field / max_sizing_field */
  float scale0;                    /* damping: 1 / damping_factor ** (inference_steps - depth - 1), or 1 */
  float scale1;                    /* last_step_damping, or 1 */
  double input_clip;               /* maximum |normalised input| (normalizer yaml: 20) */
  double epsilon;                  /* added to the variances (running_normalizer.py:18: 1e-6) */
} AmberGlueDesc;

/* RunningNormalizer._normalize (running_normalizer.py:87-106): out = float32(clip((in - mean) / sqrt(var + epsilon), +-clip)),
 * rows x width values, mean / var float64 [width] on the device.  Bit-identical to the reference. */
int amber_mpn_normalize_rows(const float* in, int64_t rows, int32_t width, const double* mean, const double* var,
                             double epsilon, double clip, float* out, void* stream);

/* RunningNormalizer.denormalize_predictions (running_normalizer.py:82-85,108-119) + PredictionTransform.forward
 * (prediction_transform.py:58-88) + clip + damping on n network outputs.  pred_mean / pred_var: float64 [1] (ignored
 * unless glue->denormalize_predictions); baseline: current sizing field [n] (needed iff predict_residual and the
 * transform is applied). */
int amber_mpn_postprocess(const AmberGlueDesc* glue, const float* pred, const float* baseline, const double* pred_mean,
                          const double* pred_var, int64_t n, float* out, void* stream);
/* Gradient of amber_mpn_postprocess with respect to pred (training: the loss sees the post-processed value).  Clip and
 * damping are part of the detached inference path and must be off. */
int amber_mpn_postprocess_backward(const AmberGlueDesc* glue, const float* pred, const float* baseline,
                                   const double* pred_mean, const double* pred_var, const float* grad_out, int64_t n,
                                   float* grad_pred, void* stream);

/* Labels into network space for the loss and the label statistics (PredictionTransform.inverse,
 * prediction_transform.py:90-128): out = transform^-1(in) - (baseline ? transform^-1(baseline) : 0), with
 * transform^-1 = softplus: v + log(-expm1(-v)); exp: log(v); none: v.  baseline (residual prediction) may be NULL. */
int amber_mpn_inverse_transform(int32_t transform, const float* in, const float* baseline, int64_t n, float* out, void* stream);

/* _predict in one call (inference): normalise x and edge_attr into the workspace, run amber_mpn_forward with the
 * SupervisedMPN head, and apply the post-processing inside the head kernel's readout epilogue (no pass of its own).
 * Replaces mesh_generation_algorithm.py:353-366 (+ the clip / damping of amber.py:213-246 when the glue asks for them).
 * desc->has_head must be 1.  workspace: amber_mpn_predict_workspace_bytes.  node_* / edge_* statistics: float64
 * [F_n] / [F_e] (ignored unless glue->normalize_inputs; a width of 0 needs none). */
size_t amber_mpn_predict_workspace_bytes(const AmberMpnDesc* desc, int64_t num_nodes, int64_t num_edges);
int amber_mpn_predict(const AmberMpnDesc* desc, const AmberGlueDesc* glue, const float* params, const float* packed,
                      int64_t num_nodes, int64_t num_edges, const float* x, const float* edge_attr, const int32_t* rowptr,
                      const int32_t* src_sorted, const int32_t* dst_sorted, const int32_t* perm, const int32_t* out_index,
                      const double* node_mean, const double* node_var, const double* edge_mean, const double* edge_var,
                      const double* pred_mean, const double* pred_var, const float* baseline, float* out_sizing,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Graph construction on the device: the step before the path.  For a batch of G meshes,
 * each given by its undirected neighbour pairs (local ids; MeshWrapper.mesh_edges for vertex graphs,
 * .element_neighbors for element graphs) and node positions (.vertex_positions / .element_midpoints), writes
 *   edge_index [2, E] int64 : per graph [a -> b | b -> a | v -> v], ids offset by the nodes of the graphs before it
 *   edge_attr  [E, F] fp32  : columns in the reference's order: distance_vector (dim), euclidean_distance, edge_curvature
 *   batch [N] int64 (graph id of every node), degree [N] fp32 (occurrences of the node among the pairs)
 * E = 2 M + (add_self_edges ? N : 0), F = amber_mpn_graph_edge_features(gdesc).
 * Replaces get_mesh_vertex_edges / get_mesh_element_edges (src/mesh_util/transforms/mesh_to_graph.py:58-145), the
 * "degree" feature (:224-225) and the collation of Batch.from_data_list reached through make_batch
 * (src/helpers/torch_util.py:34-55).  edge_index / batch / degree are bit-exact; so are the float features (IEEE
 * subtract, multiply, add, sqrt in the reference's order).  node_ptr / edge_ptr: int64 [G+1] DEVICE arrays (prefix sums
 * of the per-graph node and pair counts).  curvature: one value per undirected pair (0 off the boundary) or NULL = 0.
 * status: int32[1], 0, or 1 if a local id is outside its graph (clamped).  batch / degree may be NULL.
 * ------------------------------------------------------------------------------------------------- */
typedef struct AmberGraphDesc {
  int32_t dim;                /* columns of positions: 2 or 3 */
  int32_t add_self_edges;
  int32_t distance_vector;    /* edge feature flags ("distance_vector", "euclidean_distance", "edge_curvature") */
  int32_t euclidean_distance;
  int32_t edge_curvature;
  int32_t reserved;
  int64_t edge_capacity;      /* columns of the caller's edge_index / rows of edge_attr when this call fills only a part of
                                 them (a hierarchical graph is assembled by several calls); 0 = exactly this call's edges */
  int64_t edge_offset;        /* first column this call writes */
} AmberGraphDesc;
int32_t amber_mpn_graph_edge_features(const AmberGraphDesc* gdesc);
int amber_mpn_build_graph(const AmberGraphDesc* gdesc, int32_t num_graphs, const int64_t* node_ptr, const int64_t* edge_ptr,
                          const int64_t* nbr_a, const int64_t* nbr_b, const float* positions, const float* curvature,
                          int64_t num_nodes, int64_t num_pairs, int64_t* edge_index, float* edge_attr, int64_t* batch,
                          float* degree, int32_t* status, void* stream);

/* Inter-level edges of the hierarchical observation graph (get_inter_graph_edges, src/mesh_util/transforms/
 * mesh_to_graph.py:239-301; used by AmberData._extend_to_hierarchical_graph, src/algorithm/dataloader/amber_data.py:68-115):
 * source node k <-> nearest[k] + index_offset, both directions, 2 * num_src edges written at gdesc->edge_offset:
 *   columns [0, num_src): k -> partner with features of pos_src[k] - pos_dst[nearest[k]]; columns [num_src, 2 num_src): back.
 * nearest[k] is the partner in the destination mesh (closest vertex / containing element: the host's KD-tree or element
 * finder); positions are FLOAT64 [n, dim] device arrays and the features are computed in float64 and cast, as the
 * reference does in numpy: bit-exact.  edge_curvature, if selected, is 0 for these edges. */
int amber_mpn_build_inter_edges(const AmberGraphDesc* gdesc, int64_t num_src, int64_t num_dst, int64_t index_offset,
                                const int64_t* nearest, const double* pos_src, const double* pos_dst, int64_t* edge_index,
                                float* edge_attr, int32_t* status, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * The step after the model, in float64 as the reference computes it.  A simplicial mesh is
 * given as skfem holds it: simplices t [k, M] int64 (k = dim + 1 vertex ids per element, row-major: mesh.t), per-element
 * volumes [M] (mesh.simplex_volumes).
 *
 * amber_mpn_project_elements_to_vertices: volume-weighted mean of the values of the elements around every vertex
 * (project_elements_to_vertices, src/mesh_util/sizing_field_util.py:43-70: two torch_scatter.scatter_add over
 * mesh.t.T.flatten()).  Every vertex adds its entries in that list's order (the stable CSR sort of amber_mpn_csr_build),
 * so sums and quotient are BIT-IDENTICAL to the reference's; a vertex without elements gives nan (0 / 0), as there.
 * status: int32[1], 1 if a vertex id lies outside [0, num_vertices) (clamped).
 *
 * amber_mpn_estimate_num_elements: the heuristic element count of the mesh a sizing field would produce
 * (_mesh_sizing_field_to_num_elements, sizing_field_util.py:128-200).  vertex_field = 1: sizing_field [V] on the vertices,
 * smoothed by two gradation passes over the mesh edges (edges [2, Me] int64 = mesh.mesh_edges, positions [V, dim] float64;
 * np.maximum.at becomes an order-independent atomic maximum), averaged per element, times 0.618 in 3D; vertex_field = 0:
 * sizing_field [M] on the elements (edges / positions unused).  out: float64[2] on the device - out[0] the estimate, out[1]
 * 1.0 if a vertex id of `simplices` / `edges` lay outside [0, num_vertices) (it was clamped; the reference's numpy indexing
 * raises IndexError, and so does the Python wrapper), else 0.0 (ABI version 2; version 1 wrote out[0] only and did
 * not check the ids).  A nan in the sizing field propagates through the gradation maximum as in np.maximum.at.  Equal to
 * the reference within 1e-12 relative (pow and the order of the final sum differ in the last bits).
 * ------------------------------------------------------------------------------------------------- */
size_t amber_mpn_project_workspace_bytes(int32_t verts_per_element, int64_t num_elements, int64_t num_vertices);
int amber_mpn_project_elements_to_vertices(const int64_t* simplices, int32_t verts_per_element, int64_t num_elements,
                                           int64_t num_vertices, const double* element_values, const double* volumes, double* out,
                                           int32_t* status, void* workspace, size_t workspace_bytes, void* stream);
size_t amber_mpn_estimate_workspace_bytes(int64_t num_vertices, int64_t num_mesh_edges);
int amber_mpn_estimate_num_elements(const int64_t* simplices, int32_t verts_per_element, int64_t num_elements, int64_t num_vertices,
                                    const double* volumes, const double* sizing_field, int32_t vertex_field, int32_t dim,
                                    const int64_t* edges, int64_t num_mesh_edges, const double* positions, double gradation,
                                    double* out, void* workspace, size_t workspace_bytes, void* stream);

int amber_mpn_abi_version(void);
const char* amber_mpn_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* AMBER_MPN_H */
