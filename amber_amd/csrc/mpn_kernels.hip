// Hand-written gfx950 (CDNA4, wave64) kernels for AMBER's message-passing network forward path.
//
// Data layout of a wave tile
// --------------------------
// Every wave owns a tile of 32 rows (edges in CSR order, or nodes) x L latent features, held TRANSPOSED
// in MFMA accumulator layout: with c = lane & 31 (row of the tile) and h = lane >> 5, lane (h, c) holds
// the 16-byte chunks  X[c][8j + 4h .. 8j + 4h + 3],  j = 0 .. L/8-1  (so the two half-waves split each row).
// This is at once
//   * what a lane gets by loading 16-byte pieces of its own row from memory,
//   * the C/D layout of v_mfma_f32_32x32x2_f32 when the weights are the A operand (output feature on the
//     MFMA row) and the tile is the B operand (tile row on the MFMA column):
//         acc[fb][i]  <->  feature 32 fb + (i & 3) + 8 (i >> 2) + 4 h  =  chunk j = 4 fb + (i >> 2), element i & 3,
//   * a valid B operand of the NEXT layer's MFMAs with no data movement: k-step (j, t) contracts over
//     k = 8j + 4h + t in half-wave h, and the weights are pre-tiled (pack_kernel) to match.
// So a whole MLP (edge: W0e e + P_s[src] + P_d[dst] -> act -> W1 -> act -> +residual -> LayerNorm) runs
// inside one wave's registers; LayerNorm needs one cross-half shuffle per row.
//
// The first layer of the edge MLP is split by columns (W0 [x_s | x_d | e] = W0s x_s + W0d x_d + W0e e): the
// node kernel computes P_s = W0s x and P_d = W0d x + b0 once per node and the edge kernel gathers them.
//
// Aggregation (scatter_mean) is fused into the edge kernel: the tile is staged in LDS and summed
// sequentially per destination row in CSR order (deterministic, no atomics).  Rows cut by a tile
// boundary leave partial sums in head/tail buffers that the node kernel adds in tile order.
#include "mpn_b3.h"
#include "mpn_post.h"
#include "mpn_tile.h"

namespace amber {

// ---------------------------------------------------------------------------------------------------
// edge kernel: K2 (gather), K3 (edge MLP), K4 (residual), K5 (LayerNorm), K6 (segmented sum)
// reference: MessagePassingEdgeModule.forward (src/mpn/message_passing_modules.py:38-52),
//            MessagePassingBlock.forward (src/mpn/message_passing_block.py:117-121,127-128),
//            scatter_mean call (message_passing_modules.py:68)
// ---------------------------------------------------------------------------------------------------
// B3 (latent 256 outside the exact mode): the matrices arrive as bf16x3 pieces (mpn_b3.h), every contraction runs on the
// fp16-rate pipe and streams its matrix ONCE per workgroup through an LDS window (tile_gemm_b3_window) - the split-precision
// form of these kernels where one hi/lo fp16 matrix no longer fits the LDS of the feature-per-lane kernels.  Its barriers
// need every wave of the workgroup at every contraction: these kernels keep all their waves to the end (rows past the end
// are clamped) and every contraction sits in workgroup-uniform control flow.
#define AMBER_GEMM(ACC, W, IN, LANE)                      \
  do {                                                    \
    if constexpr (B3)                                     \
      tile_gemm_b3_window<L>(ACC, W, IN, LANE, wwin);     \
    else                                                  \
      tile_gemm<L>(ACC, W, IN, LANE);                     \
  } while (0)

template <int L, bool B3>
__global__ __launch_bounds__(kBlock) void edge_kernel(const EdgeArgs a, const int64_t nblk) {
  __shared__ __attribute__((aligned(16))) float stage_all[kWavesPerBlock * stage_floats<L>()];
  __shared__ __attribute__((aligned(16))) u32x4 wwin[B3 ? (L / 32) * 3 * 64 : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  float* stage = stage_all + wave * stage_floats<L>();

  const int64_t tile = xcd_remap(blockIdx.x, nblk) * kWavesPerBlock + wave;
  const int64_t p0 = tile * kTile;
  int nvalid = 0;
  if (p0 < a.E) nvalid = (a.E - p0) < kTile ? static_cast<int>(a.E - p0) : kTile;
  int64_t p = p0 + c;
  if (p > a.E - 1) p = a.E - 1;  // clamp: out-of-range lanes recompute the last edge and store nothing
  const int s = a.src[p], d = a.dst[p];

  Tile<L> ein;
  tile_load_row<L>(ein, a.e, p, h);
  Tile<L> acc;
  tile_load_row<L>(acc, a.Ps, s, h);
  tile_add_row<L>(acc, a.Pd, d, h);
  AMBER_GEMM(acc, a.w0e, ein, lane);
  tile_act<L>(acc, a.act, stage, lane);
#pragma unroll 1
  for (int l = 1; l < a.nl; ++l) {
    Tile<L> nxt;
    tile_load_vec<L>(nxt, a.layers[l].b, h);
    AMBER_GEMM(nxt, a.layers[l].w, acc, lane);
    tile_act<L>(nxt, a.act, stage, lane);
    acc = nxt;
  }
  if (a.residual == AMBER_MODE_INNER) tile_add<L>(acc, ein);
  if (a.layer_norm == AMBER_MODE_INNER) tile_layernorm<L>(acc, a.gamma, a.beta, a.eps, h);

  // ---- the node module aggregates THESE values (message_passing_block.py:119-123) ----
  __syncthreads();
  tile_stage<L>(stage, acc, c, h);
  __syncthreads();
  if (nvalid > 0) {
    constexpr int FPL = (L + 63) / 64;  // features per lane in the reduction
    const bool active = lane * FPL < L;
    const int fo = active ? lane * FPL : 0;
    int cur = __builtin_amdgcn_readlane(d, 0);
    bool started_before = p0 > static_cast<int64_t>(a.rowptr[cur]);
    float sum[FPL];
#pragma unroll
    for (int qf = 0; qf < FPL; ++qf) sum[qf] = 0.f;
    for (int cc = 0; cc <= nvalid; ++cc) {
      const int dc = cc < nvalid ? __builtin_amdgcn_readlane(d, cc) : -1;
      if (dc != cur) {
        // row `cur` ends at sorted position p0 + cc (exclusive) within this tile, or the tile ends
        const bool ends_here = cc < nvalid || (p0 + nvalid) == static_cast<int64_t>(a.rowptr[cur + 1]);
        float* dstp;
        if (!started_before && ends_here)
          dstp = a.agg + static_cast<int64_t>(cur) * L;
        else if (started_before)
          dstp = a.head + tile * L;
        else
          dstp = a.tail + tile * L;
        if (active) {
#pragma unroll
          for (int qf = 0; qf < FPL; ++qf) dstp[fo + qf] = sum[qf];
        }
        cur = dc;
        started_before = false;
#pragma unroll
        for (int qf = 0; qf < FPL; ++qf) sum[qf] = 0.f;
      }
      if (cc < nvalid) {
#pragma unroll
        for (int qf = 0; qf < FPL; ++qf) sum[qf] += stage[cc * stage_stride<L>() + fo + qf];
      }
    }
  }
  const bool outer = a.residual == AMBER_MODE_OUTER || a.layer_norm == AMBER_MODE_OUTER;
  if (outer) {
    if (a.residual == AMBER_MODE_OUTER) tile_add<L>(acc, ein);
    if (a.layer_norm == AMBER_MODE_OUTER) tile_layernorm<L>(acc, a.gamma, a.beta, a.eps, h);
    __syncthreads();
    tile_stage<L>(stage, acc, c, h);
    __syncthreads();
  }
  if (a.write_e) stage_store_rows<L>(stage, a.e_out ? a.e_out : a.e, p0, nvalid, nullptr, lane);
  if (a.out_edge) stage_store_rows<L>(stage, a.out_edge, p0, nvalid, a.perm, lane);
}

// ---------------------------------------------------------------------------------------------------
// node kernel: finish K6 (combine partial sums, divide by in-degree), K7 (node MLP), K8 (residual + LN),
// the next edge step's first-layer projections P_s / P_d, and optionally K9/K10 (decoder + readout).
// reference: MessagePassingNodeModule.forward (src/mpn/message_passing_modules.py:61-72),
//            MessagePassingBlock.forward (message_passing_block.py:123-128),
//            SupervisedMPN.forward (src/algorithm/architecture/supervised_mpn.py:51-54), MLP (mlp.py:60-95)
// Also used as a generic row-wise dense chain (embedding MLP tail).
// ---------------------------------------------------------------------------------------------------
template <int L, bool B3>
__global__ __launch_bounds__(kBlock) void node_kernel(const NodeArgs a, const int64_t nblk) {
  __shared__ __attribute__((aligned(16))) float stage_all[kWavesPerBlock * stage_floats<L>()];
  __shared__ __attribute__((aligned(16))) u32x4 wwin[B3 ? (L / 32) * 3 * 64 : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  float* stage = stage_all + wave * stage_floats<L>();

  const int64_t tile = xcd_remap(blockIdx.x, nblk) * kWavesPerBlock + wave;
  const int64_t n0 = tile * kTile;
  int nvalid = 0;
  if (n0 < a.N) nvalid = (a.N - n0) < kTile ? static_cast<int>(a.N - n0) : kTile;
  int64_t n = n0 + c;
  if (n > a.N - 1) n = a.N - 1;

  Tile<L> v;
  tile_load_row<L>(v, a.x_in, n, h);

  if (a.do_update) {
    // mean of the incoming (updated) edge features; zero in-degree -> 0 (scatter_mean semantics)
    Tile<L> g;
    const int r0 = a.rowptr[n], r1 = a.rowptr[n + 1];
    const int deg = r1 - r0;
    if (deg == 0) {
      tile_zero<L>(g);
    } else {
      const int t0 = r0 >> a.red_shift, t1 = (r1 - 1) >> a.red_shift;
      if (t0 == t1) {
        tile_load_row<L>(g, a.agg, n, h);
      } else {
        // the row was cut by tile boundaries: tail of its first tile, then heads, in tile order
        tile_load_row<L>(g, a.tail, t0, h);
        for (int t = t0 + 1; t <= t1; ++t) tile_add_row<L>(g, a.head, t, h);
      }
      const float degf = static_cast<float>(deg);
#pragma unroll
      for (int fb = 0; fb < L / 32; ++fb) g.a[fb] = g.a[fb] / degf;
    }
    Tile<L> acc;
    tile_load_vec<L>(acc, a.b0, h);
    AMBER_GEMM(acc, a.w0x, v, lane);
    AMBER_GEMM(acc, a.w0a, g, lane);
    tile_act<L>(acc, a.act, stage, lane);
#pragma unroll 1
    for (int l = 1; l < a.nl; ++l) {
      Tile<L> nxt;
      tile_load_vec<L>(nxt, a.layers[l].b, h);
      AMBER_GEMM(nxt, a.layers[l].w, acc, lane);
      tile_act<L>(nxt, a.act, stage, lane);
      acc = nxt;
    }
    // order of message_passing_block.py:124-128: inner residual, inner LN, outer residual, outer LN
    if (a.residual == AMBER_MODE_INNER) tile_add<L>(acc, v);
    if (a.layer_norm == AMBER_MODE_INNER) tile_layernorm<L>(acc, a.gamma, a.beta, a.eps, h);
    if (a.residual == AMBER_MODE_OUTER) tile_add<L>(acc, v);
    if (a.layer_norm == AMBER_MODE_OUTER) tile_layernorm<L>(acc, a.gamma, a.beta, a.eps, h);
    v = acc;
  }
  if (a.x_out) {
    __syncthreads();
    tile_stage<L>(stage, v, c, h);
    __syncthreads();
    stage_store_rows<L>(stage, a.x_out, n0, nvalid, nullptr, lane);
  }
  if (a.do_proj) {
    Tile<L> acc;
    tile_zero<L>(acc);
    AMBER_GEMM(acc, a.w0s, v, lane);
    __syncthreads();
    tile_stage<L>(stage, acc, c, h);
    __syncthreads();
    stage_store_rows<L>(stage, a.Ps, n0, nvalid, nullptr, lane);
    tile_load_vec<L>(acc, a.b0e, h);
    AMBER_GEMM(acc, a.w0d, v, lane);
    __syncthreads();
    tile_stage<L>(stage, acc, c, h);
    __syncthreads();
    stage_store_rows<L>(stage, a.Pd, n0, nvalid, nullptr, lane);
  }
  if (a.n_dense > 0) {
#pragma unroll 1
    for (int l = 0; l < a.n_dense; ++l) {
      Tile<L> nxt;
      tile_load_vec<L>(nxt, a.dense[l].b, h);
      AMBER_GEMM(nxt, a.dense[l].w, v, lane);
      tile_act<L>(nxt, a.dense_act, stage, lane);
      if (a.dense_norm == AMBER_NORM_LAYER)
        tile_layernorm<L>(nxt, a.dense[l].p, a.dense[l].q, 1e-5f, h);
      else if (a.dense_norm == AMBER_NORM_AFFINE)
        tile_affine<L>(nxt, a.dense[l].p, a.dense[l].q, h);
      v = nxt;
    }
    if (a.dense_out) {
      __syncthreads();
      tile_stage<L>(stage, v, c, h);
      __syncthreads();
      stage_store_rows<L>(stage, a.dense_out, n0, nvalid, nullptr, lane);
    }
  }
  if (a.do_readout) {
    const f32x4* w = reinterpret_cast<const f32x4*>(a.wr) + h;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int j = 0; j < L / 8; ++j) {
      const f32x4 x = tile_get_chunk<L>(v, j);
      const f32x4 ww = w[2 * j];
      s0 += x[0] * ww[0];
      s1 += x[1] * ww[1];
      s2 += x[2] * ww[2];
      s3 += x[3] * ww[3];
    }
    float s = (s0 + s1) + (s2 + s3);
    s += __shfl_xor(s, 32);
    s += a.br[0];
    if (h == 0 && c < nvalid) {
      const int64_t oi = a.out_index ? static_cast<int64_t>(a.out_index[n]) : n;
      if (oi >= 0) a.out[oi] = a.post.on ? post_one(a.post, s, oi) : s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// K0 / K1: input embeddings with a small contraction (F raw features): out[r] = act(W in[idx(r)] + b).
// reference: Embedding.forward (src/mpn/common/embedding.py:44-54), EmptyLinear (empty_linear_layer.py:44-45)
// One thread per (row, 4 output features); W^T staged in LDS.  HBM-write bound.
// ---------------------------------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(256) void embed_kernel(const float* __restrict__ in, int64_t rows, int F,
                                                    const float* __restrict__ W, const float* __restrict__ b,
                                                    const int32_t* __restrict__ perm, int act, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float wt[];  // [F][L]
  for (int i = threadIdx.x; i < F * L; i += blockDim.x) {
    const int f = i / F, k = i % F;  // W is [L, F] row-major
    wt[k * L + f] = W[i];
  }
  __syncthreads();
  constexpr int LPR = L / 4;
  constexpr int RPB = 256 / LPR;
  const int q = threadIdx.x % LPR;
  const f32x4 bias = reinterpret_cast<const f32x4*>(b)[q];
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * RPB + threadIdx.x / LPR; r < rows;
       r += static_cast<int64_t>(gridDim.x) * RPB) {
    const int64_t sr = perm ? static_cast<int64_t>(perm[r]) : r;
    f32x4 acc = bias;
    for (int k = 0; k < F; ++k) {
      const float xv = in[sr * F + k];
      acc += xv * reinterpret_cast<const f32x4*>(wt + k * L)[q];
    }
    if (act >= 0) {
      acc[0] = act_one(acc[0], act);
      acc[1] = act_one(acc[1], act);
      acc[2] = act_one(acc[2], act);
      acc[3] = act_one(acc[3], act);
    }
    reinterpret_cast<f32x4*>(out + r * L)[q] = acc;
  }
}

// ---------------------------------------------------------------------------------------------------
// K0 and the first step's projections in one pass over the raw node features.  The node embedding is linear (or a
// learned constant), so the projections of the embedded rows are linear in the raw features as well:
//   x = A raw + b,   P_s = W0s x = (W0s A) raw + W0s b,   P_d = W0d x + b0 = (W0d A) raw + (W0d b + b0)
// with tables [3][F + 1][L] (row F: the constant) formed once per weight update (fold_node_tables_kernel, called by
// amber_mpn_pack_weights).  One write-bound kernel replaces the embedding kernel and a projection launch of the node
// kernel (which re-read x for two contractions).  x is computed with the embedding kernel's own operations (bit-equal).
// reference: Embedding.forward (src/mpn/common/embedding.py:44-54), first layer of MessagePassingEdgeModule's MLP on
// [x_s, x_d, e] (src/mpn/message_passing_modules.py:46-52)
// ---------------------------------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(256) void embed_proj_kernel(const float* __restrict__ in, int64_t rows, int F,
                                                         const float* __restrict__ tables, float* __restrict__ out_x,
                                                         float* __restrict__ out_ps, float* __restrict__ out_pd) {
  extern __shared__ __attribute__((aligned(16))) float tb[];  // [3][F + 1][L]
  const int TL = (F + 1) * L;
  for (int i = threadIdx.x; i < 3 * TL; i += blockDim.x) tb[i] = tables[i];
  __syncthreads();
  constexpr int LPR = L / 4;
  constexpr int RPB = 256 / LPR;
  const int q = threadIdx.x % LPR;
  f32x4 c[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) c[t] = reinterpret_cast<const f32x4*>(tb + t * TL + F * L)[q];
  for (int64_t r = static_cast<int64_t>(blockIdx.x) * RPB + threadIdx.x / LPR; r < rows;
       r += static_cast<int64_t>(gridDim.x) * RPB) {
    f32x4 acc[3] = {c[0], c[1], c[2]};
    for (int k = 0; k < F; ++k) {
      const float xv = in[r * F + k];
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] += xv * reinterpret_cast<const f32x4*>(tb + t * TL + k * L)[q];
    }
    reinterpret_cast<f32x4*>(out_x + r * L)[q] = acc[0];
    reinterpret_cast<f32x4*>(out_ps + r * L)[q] = acc[1];
    reinterpret_cast<f32x4*>(out_pd + r * L)[q] = acc[2];
  }
}
// tables[0][k][j] = A[j][k], [0][F][j] = b[j];  tables[1] = W0s (A | b);  tables[2] = W0d (A | b) (+ b0 in row F);
// W0 = first layer of the step-0 edge MLP, [L, 3 L] row-major; A = node embedding weight [L, F] (nullptr: F = 0)
__global__ void fold_node_tables_kernel(const float* __restrict__ A, const float* __restrict__ b, const float* __restrict__ W0,
                                        const float* __restrict__ b0, int L, int F, float* __restrict__ tables) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (t, k, j)
  if (i >= 3 * (F + 1) * L) return;
  const int j = i % L, k = (i / L) % (F + 1), t = i / (L * (F + 1));
  float v;
  if (t == 0) {
    v = k < F ? A[j * F + k] : b[j];
  } else {
    const float* wr = W0 + static_cast<int64_t>(j) * (3 * L) + (t - 1) * L;
    float sum = 0.f;
    for (int m = 0; m < L; ++m) sum = fmaf(wr[m], k < F ? A[m * F + k] : b[m], sum);
    v = (t == 2 && k == F) ? sum + b0[j] : sum;
  }
  tables[i] = v;
}

// Re-tile columns [col0, col0 + L) of W[L, ld] into MFMA operand order:
// out[((j * L/32 + fb) * 64 + lane) * 4 + t] = W[32 fb + (lane & 31)][col0 + 8 j + 4 (lane >> 5) + t]
template <int L>
__global__ void pack_kernel(const float* __restrict__ W, int ld, int col0, float* __restrict__ out) {
  constexpr int NFB = L / 32;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // float4 index
  if (i >= L * L / 4) return;
  const int lane = i & 63, fb = (i >> 6) % NFB, j = (i >> 6) / NFB;
  const float* src = W + static_cast<int64_t>(32 * fb + (lane & 31)) * ld + col0 + 8 * j + 4 * (lane >> 5);
  f32x4 v;
  v[0] = src[0];
  v[1] = src[1];
  v[2] = src[2];
  v[3] = src[3];
  reinterpret_cast<f32x4*>(out)[i] = v;
}

// Stand-alone K6: deterministic CSR segmented mean (thread per (row, 16-byte feature chunk), summed in row order).
__global__ void segment_mean_kernel(const float* __restrict__ vals, const int32_t* __restrict__ rowptr, int64_t N,
                                    int L, float* __restrict__ out) {
  const int lpr = L / 4;
  const int64_t total = N * lpr;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t n = i / lpr;
    const int q = static_cast<int>(i % lpr);
    const int r0 = rowptr[n], r1 = rowptr[n + 1];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int p = r0; p < r1; ++p) s += reinterpret_cast<const f32x4*>(vals + static_cast<int64_t>(p) * L)[q];
    const float cnt = static_cast<float>(r1 - r0 > 1 ? r1 - r0 : 1);
    reinterpret_cast<f32x4*>(out + n * L)[q] = s / cnt;
  }
}

__global__ void permute_rows_kernel(const float* __restrict__ in, const int32_t* __restrict__ perm, int64_t rows,
                                    int width, int gather, float* __restrict__ out) {
  const int lpr = width / 4;
  const int64_t total = rows * lpr;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t r = i / lpr;
    const int q = static_cast<int>(i % lpr);
    const int64_t o = perm[r];
    const int64_t sr = gather ? o : r, dr = gather ? r : o;
    reinterpret_cast<f32x4*>(out + dr * width)[q] = reinterpret_cast<const f32x4*>(in + sr * width)[q];
  }
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

template <int L, bool B3 = false>
static hipError_t launch_edge_t(const EdgeArgs& a, hipStream_t s) {
  const int64_t nblk = cdiv(cdiv(a.E, kTile), kWavesPerBlock);
  if (nblk == 0) return hipSuccess;
  hipLaunchKernelGGL((edge_kernel<L, B3>), dim3(static_cast<unsigned>(nblk)), dim3(kBlock), 0, s, a, nblk);
  return hipGetLastError();
}
template <int L, bool B3 = false>
static hipError_t launch_node_t(const NodeArgs& a, hipStream_t s) {
  const int64_t nblk = cdiv(cdiv(a.N, kTile), kWavesPerBlock);
  if (nblk == 0) return hipSuccess;
  hipLaunchKernelGGL((node_kernel<L, B3>), dim3(static_cast<unsigned>(nblk)), dim3(kBlock), 0, s, a, nblk);
  return hipGetLastError();
}
template <int L>
static hipError_t launch_embed_t(const float* in, int64_t rows, int F, const float* W, const float* b,
                                 const int32_t* perm, int act, float* out, hipStream_t s) {
  if (rows == 0) return hipSuccess;
  constexpr int RPB = 256 / (L / 4);
  int64_t grid = cdiv(rows, RPB);
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(embed_kernel<L>, dim3(static_cast<unsigned>(grid)), dim3(256), static_cast<size_t>(F) * L * sizeof(float),
                     s, in, rows, F, W, b, perm, act, out);
  return hipGetLastError();
}
template <int L>
static hipError_t launch_embed_proj_t(const float* in, int64_t rows, int F, const float* tables, float* out_x, float* out_ps,
                                      float* out_pd, hipStream_t s) {
  if (rows == 0) return hipSuccess;
  constexpr int RPB = 256 / (L / 4);
  int64_t grid = cdiv(rows, RPB);
  if (grid > 256 * 16) grid = 256 * 16;
  const size_t lds = static_cast<size_t>(3) * (F + 1) * L * sizeof(float);
  static thread_local size_t allowed = 0;  // dynamic LDS beyond 64 KB must be allowed per function
  if (lds > 64 * 1024 && lds > allowed) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(embed_proj_kernel<L>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       static_cast<int>(lds));
    if (e != hipSuccess) return e;
    allowed = lds;
  }
  hipLaunchKernelGGL(embed_proj_kernel<L>, dim3(static_cast<unsigned>(grid)), dim3(256), lds, s, in, rows, F, tables, out_x, out_ps,
                     out_pd);
  return hipGetLastError();
}
template <int L>
static hipError_t launch_pack_t(const float* W, int ld, int col0, float* out, hipStream_t s) {
  hipLaunchKernelGGL(pack_kernel<L>, dim3(cdiv(L * L / 4, 256)), dim3(256), 0, s, W, ld, col0, out);
  return hipGetLastError();
}

#define AMBER_DISPATCH_L(L, CALL)       \
  switch (L) {                          \
    case 32:                            \
      return CALL<32>;                  \
    case 64:                            \
      return CALL<64>;                  \
    case 128:                           \
      return CALL<128>;                 \
    default:                            \
      return hipErrorInvalidValue;      \
  }

hipError_t launch_edge(int L, int b3, const EdgeArgs& a, hipStream_t s) {
  if (b3) return L == 256 ? launch_edge_t<256, true>(a, s) : (L == 128 ? launch_edge_t<128, true>(a, s) : hipErrorInvalidValue);
  switch (L) {
    case 32: return launch_edge_t<32>(a, s);
    case 64: return launch_edge_t<64>(a, s);
    case 128: return launch_edge_t<128>(a, s);
    case 256: return launch_edge_t<256>(a, s);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_node(int L, int b3, const NodeArgs& a, hipStream_t s) {
  if (b3) return L == 256 ? launch_node_t<256, true>(a, s) : (L == 128 ? launch_node_t<128, true>(a, s) : hipErrorInvalidValue);
  switch (L) {
    case 32: return launch_node_t<32>(a, s);
    case 64: return launch_node_t<64>(a, s);
    case 128: return launch_node_t<128>(a, s);
    case 256: return launch_node_t<256>(a, s);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_embed_proj(int L, const float* in, int64_t rows, int F, const float* tables, float* out_x, float* out_ps,
                             float* out_pd, hipStream_t s) {
  switch (L) {
    case 32: return launch_embed_proj_t<32>(in, rows, F, tables, out_x, out_ps, out_pd, s);
    case 64: return launch_embed_proj_t<64>(in, rows, F, tables, out_x, out_ps, out_pd, s);
    case 128: return launch_embed_proj_t<128>(in, rows, F, tables, out_x, out_ps, out_pd, s);
    case 256: return launch_embed_proj_t<256>(in, rows, F, tables, out_x, out_ps, out_pd, s);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_fold_node_tables(int L, int F, const float* A, const float* b, const float* W0, const float* b0, float* tables,
                                   hipStream_t s) {
  const int total = 3 * (F + 1) * L;
  hipLaunchKernelGGL(fold_node_tables_kernel, dim3((total + 255) / 256), dim3(256), 0, s, A, b, W0, b0, L, F, tables);
  return hipGetLastError();
}
hipError_t launch_embed(int L, const float* in, int64_t rows, int F, const float* W, const float* b, const int32_t* perm,
                        int act, float* out, hipStream_t s) {
  switch (L) {
    case 32: return launch_embed_t<32>(in, rows, F, W, b, perm, act, out, s);
    case 64: return launch_embed_t<64>(in, rows, F, W, b, perm, act, out, s);
    case 128: return launch_embed_t<128>(in, rows, F, W, b, perm, act, out, s);
    case 256: return launch_embed_t<256>(in, rows, F, W, b, perm, act, out, s);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_pack(int L, const float* W, int ld, int col0, float* out, hipStream_t s) {
  switch (L) {
    case 32: return launch_pack_t<32>(W, ld, col0, out, s);
    case 64: return launch_pack_t<64>(W, ld, col0, out, s);
    case 128: return launch_pack_t<128>(W, ld, col0, out, s);
    case 256: return launch_pack_t<256>(W, ld, col0, out, s);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_segment_mean(const float* vals, const int32_t* rowptr, int64_t N, int L, float* out, hipStream_t s) {
  if (N == 0) return hipSuccess;
  int64_t grid = cdiv(N * (L / 4), 256);
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(segment_mean_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, s, vals, rowptr, N, L, out);
  return hipGetLastError();
}
hipError_t launch_permute_rows(const float* in, const int32_t* perm, int64_t rows, int width, int gather, float* out,
                               hipStream_t s) {
  if (rows == 0) return hipSuccess;
  int64_t grid = cdiv(rows * (width / 4), 256);
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(permute_rows_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, s, in, perm, rows, width, gather, out);
  return hipGetLastError();
}

}  // namespace amber
