// Backward (training) kernels of the message-passing network, exact fp32 (v_mfma_f32_32x32x2_f32), gfx950.
//
// Design: activation checkpointing at block granularity.  The training forward keeps only the block INPUTS of every
// step (x_t, e_t: the "tape"); each backward kernel recomputes its module's forward for a 32-row tile in registers
// (same row-per-lane layout and MFMA chain as the exact forward kernels, mpn_kernels.hip) and runs the chain rule on
// the spot.  What leaves a kernel is only what a reduction over ALL rows needs afterwards:
//   * the pre-activation gradients DH[l] and layer inputs A[l-1] of every layer -> weight gradients
//     dW_l = DH[l]^T A[l-1] by a split-K MFMA kernel (rows are the contraction dimension) + an ordered reduction,
//   * per-wave partial column sums for the LayerNorm gradients,
//   * the gradient with respect to the module's inputs (in place over the incoming gradient).
// The weight split of the forward carries over: dP_s[n] = sum over edges with source n of DH_e[0], dP_d[n] likewise
// over destinations (deterministic segmented sums over the two CSR orders, no atomics), and the node-level products
// dW0s = dP_s^T x, dW0d = dP_d^T x, dx += W0s^T dP_s + W0d^T dP_d replace their E-sized counterparts.
//
// reference: autograd through MessagePassingBlock.forward (src/mpn/message_passing_block.py:107-146),
// MessagePassingEdgeModule / NodeModule (src/mpn/message_passing_modules.py:38-72), LatentMLP
// (src/mpn/common/latent_mlp.py:116-119), SupervisedMPN.forward (src/algorithm/architecture/supervised_mpn.py:39-56);
// the training step that needs it: src/algorithm/core/amber.py:62-82.
#include "mpn_tile.h"

namespace amber {

// ---------------------------------------------------------------------------------------------------
// wave-local tile helpers (no workgroup barrier: every wave walks its own tiles)
// ---------------------------------------------------------------------------------------------------
template <int L>
DEVINL void tile_store_w(float* stage, const Tile<L>& t, float* __restrict__ out, int64_t row0, int nvalid, int lane) {
  wave_sync();
  tile_stage<L>(stage, t, lane & 31, lane >> 5);
  wave_sync();
  stage_store_rows<L>(stage, out, row0, nvalid, nullptr, lane);
}
template <int L>
DEVINL void tile_act_w(Tile<L>& t, int act, float* stage, int lane) {
  if (act == AMBER_ACT_RELU || act == AMBER_ACT_LEAKYRELU) {
    tile_act_fast<L>(t, act);
  } else {
    wave_sync();
    tile_stage<L>(stage, t, lane & 31, lane >> 5);
    wave_sync();
    for (int i = lane; i < stage_floats<L>(); i += 64) stage[i] = act_one(stage[i], act);
    wave_sync();
    tile_unstage<L>(stage, t, lane & 31, lane >> 5);
  }
}
// d *= act'(pre-activation), expressed through the activation's OUTPUT a (relu / leakyrelu: the sign survives;
// tanh: 1 - a^2).  At a == 0 torch's backward takes the negative branch (x > 0 ? 1 : slope).
template <int L>
DEVINL void tile_mul_dact(Tile<L>& d, const Tile<L>& a, int act) {
  if (act == AMBER_ACT_TANH) {
#pragma unroll
    for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
      for (int i = 0; i < 16; ++i) d.a[fb][i] *= 1.f - a.a[fb][i] * a.a[fb][i];
  } else {
    const float slope = act == AMBER_ACT_LEAKYRELU ? 0.01f : 0.f;
#pragma unroll
    for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
      for (int i = 0; i < 16; ++i) d.a[fb][i] = a.a[fb][i] > 0.f ? d.a[fb][i] : slope * d.a[fb][i];
  }
}
// sum over the L features of this lane's row (the row lives in lanes (0, c) and (1, c))
template <int L>
DEVINL float tile_row_sum(const Tile<L>& t) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
      s0 += t.a[fb][i + 0];
      s1 += t.a[fb][i + 1];
      s2 += t.a[fb][i + 2];
      s3 += t.a[fb][i + 3];
    }
  float s = (s0 + s1) + (s2 + s3);
  s += __shfl_xor(s, 32);
  return s;
}
template <int L>
DEVINL float tile_row_dot(const Tile<L>& x, const Tile<L>& y) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
      s0 += x.a[fb][i + 0] * y.a[fb][i + 0];
      s1 += x.a[fb][i + 1] * y.a[fb][i + 1];
      s2 += x.a[fb][i + 2] * y.a[fb][i + 2];
      s3 += x.a[fb][i + 3] * y.a[fb][i + 3];
    }
  float s = (s0 + s1) + (s2 + s3);
  s += __shfl_xor(s, 32);
  return s;
}
// t -> (t - mean) * rstd per row; returns rstd (the statistics of torch.nn.functional.layer_norm: biased variance)
template <int L>
DEVINL float tile_ln_xhat(Tile<L>& t, float eps) {
  const float mean = tile_row_sum<L>(t) * (1.0f / L);
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb) t.a[fb] = t.a[fb] - mean;
  const float var = tile_row_dot<L>(t, t) * (1.0f / L);
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb) t.a[fb] = t.a[fb] * rstd;
  return rstd;
}
// psum[q] += sum over the tile's 32 rows of column fo + q (through the wave's LDS stage; rows that do not exist must
// hold zeros)
template <int L, int FPL>
DEVINL void tile_colsum_acc(float* stage, const Tile<L>& t, float (&psum)[FPL], bool active, int fo, int lane) {
  wave_sync();
  tile_stage<L>(stage, t, lane & 31, lane >> 5);
  wave_sync();
  if (active) {
    for (int r = 0; r < kTile; ++r)
#pragma unroll
      for (int q = 0; q < FPL; ++q) psum[q] += stage[r * stage_stride<L>() + fo + q];
  }
}
// make this wave's global stores visible to its own later loads issued from other lanes
DEVINL void wave_global_handover() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// ---------------------------------------------------------------------------------------------------
// chain backward: recompute + chain rule of one module for a 32-row tile.
//   mode EDGE: rows = edges (CSR order); h0 = P_s[src] + P_d[dst] + W0e e;  dy = dE + DM[dst] / deg(dst)
//   mode NODE: rows = nodes; h0 = b0 + W0x x + W0a m;                       dy = dX;  also DM = W0a^T dh0
//   mode HEAD: rows = nodes; decoder chain on x, out = wr . a_last + br;    dy = dout[out_index] * wr
// then a_l = act(W_l a_{l-1} + b_l), r = a_last (+ rin), y = LayerNorm(r) (inner modes).
// ---------------------------------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(kBlock) void chain_bwd_kernel(const ChainBwdArgs a, const int64_t ntiles) {
  __shared__ __attribute__((aligned(16))) float stage_all[kWavesPerBlock * stage_floats<L>()];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  float* stage = stage_all + wave * stage_floats<L>();
  constexpr int FPL = (L + 63) / 64;
  const bool active = lane * FPL < L;
  const int fo = active ? lane * FPL : 0;
  float ps0[FPL], ps1[FPL];
#pragma unroll
  for (int q = 0; q < FPL; ++q) ps0[q] = ps1[q] = 0.f;

  const int64_t gw = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  const int64_t G = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  for (int64_t tile = gw; tile < ntiles; tile += G) {
    const int64_t p0 = tile * kTile;
    const int nvalid = (a.rows - p0) < kTile ? static_cast<int>(a.rows - p0) : kTile;
    int64_t p = p0 + c;
    if (p > a.rows - 1) p = a.rows - 1;  // rows beyond the end repeat the last row; their gradient is zeroed below
    const bool rowok = c < nvalid;

    // ---- recompute the forward chain ----
    Tile<L> rin;
    tile_load_row<L>(rin, a.rin, p, h);
    Tile<L> cur;
    int d = 0;
    if (a.nl == 0) {
      cur = rin;
    } else {
      if (a.mode == AMBER_BWD_EDGE) {
        const int s = a.src[p];
        d = a.dst[p];
        tile_load_row<L>(cur, a.Ps, s, h);
        tile_add_row<L>(cur, a.Pd, d, h);
        tile_gemm<L>(cur, a.w_in, rin, lane);
      } else {
        tile_load_vec<L>(cur, a.b0, h);
        tile_gemm<L>(cur, a.w_in, rin, lane);
        if (a.mode == AMBER_BWD_NODE) {
          Tile<L> mt;
          tile_load_row<L>(mt, a.m, p, h);
          tile_gemm<L>(cur, a.w_m, mt, lane);
        }
      }
      tile_act_w<L>(cur, a.act, stage, lane);
#pragma unroll 1
      for (int l = 1; l < a.nl; ++l) {
        tile_store_w<L>(stage, cur, a.A[l - 1], p0, nvalid, lane);  // input of layer l: needed by dW_l and by act' below
        Tile<L> nxt;
        tile_load_vec<L>(nxt, a.layers[l].b, h);
        tile_gemm<L>(nxt, a.layers[l].w, cur, lane);
        tile_act_w<L>(nxt, a.act, stage, lane);
        cur = nxt;
      }
    }
    // ---- gradient arriving at the module's output ----
    Tile<L> dr;
    if (a.mode == AMBER_BWD_EDGE) {
      if (a.nl == 0) d = a.dst[p];
      tile_load_row<L>(dr, a.dio, p, h);
      const float inv = 1.0f / static_cast<float>(max(a.rowptr[d + 1] - a.rowptr[d], 1));  // scatter_mean's divisor
      const f32x4* pm = reinterpret_cast<const f32x4*>(a.dm_in + static_cast<int64_t>(d) * L) + h;
#pragma unroll
      for (int j = 0; j < L / 8; ++j) tile_set_chunk<L>(dr, j, tile_get_chunk<L>(dr, j) + inv * pm[2 * j]);
    } else if (a.mode == AMBER_BWD_NODE) {
      tile_load_row<L>(dr, a.dio, p, h);
    } else {
      const int64_t oi = a.out_index ? static_cast<int64_t>(a.out_index[p]) : p;
      const float dscal = (rowok && oi >= 0) ? a.dout[oi] : 0.f;
      // readout gradients: d wr = sum_rows dout * a_last, d br = sum_rows dout
      Tile<L> t;
#pragma unroll
      for (int fb = 0; fb < L / 32; ++fb) t.a[fb] = cur.a[fb] * dscal;
      tile_colsum_acc<L, FPL>(stage, t, ps0, active, fo, lane);
#pragma unroll
      for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
        for (int i = 0; i < 16; ++i) t.a[fb][i] = dscal;
      tile_colsum_acc<L, FPL>(stage, t, ps1, active, fo, lane);
      tile_load_vec<L>(dr, a.wr, h);
#pragma unroll
      for (int fb = 0; fb < L / 32; ++fb) dr.a[fb] = dr.a[fb] * dscal;
    }
    if (!rowok) tile_zero<L>(dr);
    // ---- LayerNorm backward (statistics recomputed) ----
    if (a.layer_norm == AMBER_MODE_INNER) {
      Tile<L> xh = cur;
      if (a.residual == AMBER_MODE_INNER) tile_add<L>(xh, rin);
      const float rstd = tile_ln_xhat<L>(xh, a.eps);
      {
        Tile<L> t;
#pragma unroll
        for (int fb = 0; fb < L / 32; ++fb) t.a[fb] = dr.a[fb] * xh.a[fb];
        tile_colsum_acc<L, FPL>(stage, t, ps0, active, fo, lane);  // d gamma
        tile_colsum_acc<L, FPL>(stage, dr, ps1, active, fo, lane); // d beta
      }
      Tile<L> g;
      tile_load_vec<L>(g, a.gamma, h);
#pragma unroll
      for (int fb = 0; fb < L / 32; ++fb) g.a[fb] = g.a[fb] * dr.a[fb];
      const float m1 = tile_row_sum<L>(g) * (1.0f / L);
      const float m2 = tile_row_dot<L>(g, xh) * (1.0f / L);
#pragma unroll
      for (int fb = 0; fb < L / 32; ++fb) dr.a[fb] = (g.a[fb] - m1 - xh.a[fb] * m2) * rstd;
    }
    // ---- chain rule through the layers, last to first ----
    Tile<L> din;
    if (a.nl == 0) {
      din = dr;
    } else {
      Tile<L> dcur = dr;
      if (a.nl > 1) wave_global_handover();  // the A[l] rows stored above are re-read by other lanes
#pragma unroll 1
      for (int l = a.nl - 1; l >= 1; --l) {
        if (l != a.nl - 1) tile_load_row<L>(cur, a.A[l], p, h);
        tile_mul_dact<L>(dcur, cur, a.act);
        tile_store_w<L>(stage, dcur, a.DH[l], p0, nvalid, lane);
        Tile<L> nd;
        tile_zero<L>(nd);
        tile_gemm<L>(nd, a.wT[l], dcur, lane);
        dcur = nd;
      }
      if (a.nl > 1) tile_load_row<L>(cur, a.A[0], p, h);
      tile_mul_dact<L>(dcur, cur, a.act);
      tile_store_w<L>(stage, dcur, a.DH[0], p0, nvalid, lane);
      if (a.residual == AMBER_MODE_INNER)
        din = dr;
      else
        tile_zero<L>(din);
      tile_gemm<L>(din, a.wT_in, dcur, lane);
      if (a.mode == AMBER_BWD_NODE) {
        Tile<L> dm;
        tile_zero<L>(dm);
        tile_gemm<L>(dm, a.wT_m, dcur, lane);
        tile_store_w<L>(stage, dm, a.dm_out, p0, nvalid, lane);
      }
    }
    if (a.mode == AMBER_BWD_HEAD) tile_add_row<L>(din, a.dio, p, h);  // joins a gradient that arrived at the node latents
    tile_store_w<L>(stage, din, a.dio, p0, nvalid, lane);
  }
  if (a.part && active) {
    float* pp = a.part + gw * 2 * L;
#pragma unroll
    for (int q = 0; q < FPL; ++q) {
      pp[fo + q] = ps0[q];
      pp[L + fo + q] = ps1[q];
    }
  }
}

// out0[f] += sum_g part[g][0][f] (f < n0), out1[f] += sum_g part[g][1][f] (f < n1), in g order
__global__ void colpart_reduce_kernel(const float* __restrict__ part, int G, int L, float* __restrict__ out0, int n0,
                                      float* __restrict__ out1, int n1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * L) return;
  const int k = i / L, f = i % L;
  float* out = k == 0 ? out0 : out1;
  if (!out || f >= (k == 0 ? n0 : n1)) return;
  float s = 0.f;
  for (int g = 0; g < G; ++g) s += part[(static_cast<int64_t>(g) * 2 + k) * L + f];
  out[f] += s;
}

// ---------------------------------------------------------------------------------------------------
// weight gradient: P[g] = A[rows of chunk g]^T B[rows of chunk g]  (A, B: [M, L] row-major), the rows are the
// contraction dimension: v_mfma_f32_32x32x2_f32 with two rows per instruction, both operands loaded coalesced
// straight from memory (lane (h, c): row r + h, column 32 blk + c).  One wave per block of 32 output features.
// ---------------------------------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(64 * (L / 32)) void wgrad_kernel(const float* __restrict__ A, const float* __restrict__ B, int64_t M,
                                                              int64_t chunk, float* __restrict__ part, float* __restrict__ cpart) {
  constexpr int NFB = L / 32, U = 8;
  const int lane = threadIdx.x & 63, ob = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * chunk;
  const int64_t r1 = r0 + chunk < M ? r0 + chunk : M;
  f32x16 acc[NFB];
#pragma unroll
  for (int j = 0; j < NFB; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  float cs = 0.f;
  for (int64_t r = r0; r < r1; r += 2 * U) {
    float av[U], bv[U][NFB];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t rr = r + 2 * u + h;
      const bool ok = rr < r1;
      const int64_t rc = ok ? rr : r0;
      const float x = A[rc * L + 32 * ob + c];
      av[u] = ok ? x : 0.f;
#pragma unroll
      for (int j = 0; j < NFB; ++j) {
        const float y = B[rc * L + 32 * j + c];
        bv[u][j] = ok ? y : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      cs += av[u];
#pragma unroll
      for (int j = 0; j < NFB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][j], acc[j], 0, 0, 0);
    }
  }
  float* P = part + static_cast<int64_t>(blockIdx.x) * L * L;
#pragma unroll
  for (int j = 0; j < NFB; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = (i & 3) + 8 * (i >> 2) + 4 * h;
      P[(32 * ob + m) * L + 32 * j + c] = acc[j][i];
    }
  if (cpart) {
    cs += __shfl_xor(cs, 32);
    if (h == 0) cpart[static_cast<int64_t>(blockIdx.x) * L + 32 * ob + c] = cs;
  }
}
// dW[o][col0 + k] += sum_g part[g][o][k];  db[o] += sum_g cpart[g][o]   (g ascending: deterministic)
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ cpart, int G, int L,
                                    float* __restrict__ dW, int ld, int col0, float* __restrict__ db) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < L * L) {
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += part[static_cast<int64_t>(g) * L * L + i];
    dW[static_cast<int64_t>(i / L) * ld + col0 + (i % L)] += s;
  } else if (i < L * L + L && db && cpart) {
    const int o = i - L * L;
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += cpart[static_cast<int64_t>(g) * L + o];
    db[o] += s;
  }
}

// ---------------------------------------------------------------------------------------------------
// embedding gradient (few raw features F): part[g][k][o] = sum over the chunk's rows of d[row][o] * raw[idx(row)][k],
// k = F is the bias (raw = 1).  reference: Embedding.forward (src/mpn/common/embedding.py:44-54)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ d, const float* __restrict__ raw,
                                                        const int32_t* __restrict__ perm, int64_t rows, int64_t chunk, int L,
                                                        int F, float* __restrict__ part) {
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * chunk;
  const int64_t r1 = r0 + chunk < rows ? r0 + chunk : rows;
  for (int idx = threadIdx.x; idx < L * (F + 1); idx += blockDim.x) {
    const int o = idx % L, k = idx / L;
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      float w = 1.f;
      if (k < F) w = raw[(perm ? static_cast<int64_t>(perm[r]) : r) * F + k];
      s += d[r * L + o] * w;
    }
    part[static_cast<int64_t>(blockIdx.x) * L * (F + 1) + idx] = s;
  }
}
__global__ void embed_bwd_reduce_kernel(const float* __restrict__ part, int G, int L, int F, float* __restrict__ dW,
                                        float* __restrict__ db) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= L * (F + 1)) return;
  const int o = idx % L, k = idx / L;
  float s = 0.f;
  for (int g = 0; g < G; ++g) s += part[static_cast<int64_t>(g) * L * (F + 1) + idx];
  if (k < F)
    dW[o * F + k] += s;
  else
    db[o] += s;
}

// ---------------------------------------------------------------------------------------------------
// deterministic segmented sum / mean over CSR rows, optionally through a row index (the source-sorted order addresses
// rows of the destination-sorted buffers): out[n] = sum_{q in [rowptr[n], rowptr[n+1])} vals[idx ? idx[q] : q]
// ---------------------------------------------------------------------------------------------------
__global__ void segment_reduce_kernel(const float* __restrict__ vals, const int32_t* __restrict__ rowptr,
                                      const int32_t* __restrict__ idx, int64_t N, int L, int mean, float* __restrict__ out) {
  const int lpr = L / 4;
  const int64_t total = N * lpr;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t n = i / lpr;
    const int q = static_cast<int>(i % lpr);
    const int r0 = rowptr[n], r1 = rowptr[n + 1];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int p = r0; p < r1; ++p) {
      const int64_t row = idx ? static_cast<int64_t>(idx[p]) : p;
      s += reinterpret_cast<const f32x4*>(vals + row * L)[q];
    }
    if (mean) s = s / static_cast<float>(r1 - r0 > 1 ? r1 - r0 : 1);
    reinterpret_cast<f32x4*>(out + n * L)[q] = s;
  }
}

// io[r] += W0 in0[r] + W1 in1[r] on row tiles (W packed): dx += W0s^T dP_s + W0d^T dP_d
template <int L>
__global__ __launch_bounds__(kBlock) void rows_gemm_add_kernel(float* __restrict__ io, const float* __restrict__ in0,
                                                               const float* __restrict__ w0, const float* __restrict__ in1,
                                                               const float* __restrict__ w1, int64_t rows, int64_t ntiles) {
  __shared__ __attribute__((aligned(16))) float stage_all[kWavesPerBlock * stage_floats<L>()];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c = lane & 31;
  float* stage = stage_all + wave * stage_floats<L>();
  const int64_t tile = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  if (tile >= ntiles) return;
  const int64_t p0 = tile * kTile;
  const int nvalid = (rows - p0) < kTile ? static_cast<int>(rows - p0) : kTile;
  int64_t p = p0 + c;
  if (p > rows - 1) p = rows - 1;
  Tile<L> acc, t;
  tile_load_row<L>(acc, io, p, h);
  tile_load_row<L>(t, in0, p, h);
  tile_gemm<L>(acc, w0, t, lane);
  if (in1) {
    tile_load_row<L>(t, in1, p, h);
    tile_gemm<L>(acc, w1, t, lane);
  }
  tile_store_w<L>(stage, acc, io, p0, nvalid, lane);
}

// Re-tile the TRANSPOSE of columns [col0, col0 + L) of W[L, ld] into MFMA operand order (see pack_kernel):
// element (row r, column k) of the packed matrix is W[k][col0 + r]
template <int L>
__global__ void pack_t_kernel(const float* __restrict__ W, int ld, int col0, float* __restrict__ out) {
  constexpr int NFB = L / 32;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // float4 index
  if (i >= L * L / 4) return;
  const int lane = i & 63, fb = (i >> 6) % NFB, j = (i >> 6) / NFB;
  const int r = 32 * fb + (lane & 31), k0 = 8 * j + 4 * (lane >> 5);
  f32x4 v;
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = W[static_cast<int64_t>(k0 + t) * ld + col0 + r];
  reinterpret_cast<f32x4*>(out)[i] = v;
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

int chain_bwd_grid(int64_t rows) {
  int64_t nblk = cdiv(cdiv(rows, kTile), kWavesPerBlock);
  if (nblk > kBwdMaxBlocks) nblk = kBwdMaxBlocks;
  if (nblk < 1) nblk = 1;
  return static_cast<int>(nblk);
}
template <int L>
static hipError_t launch_chain_bwd_t(const ChainBwdArgs& a, hipStream_t s) {
  const int64_t ntiles = cdiv(a.rows, kTile);
  hipLaunchKernelGGL(chain_bwd_kernel<L>, dim3(chain_bwd_grid(a.rows)), dim3(kBlock), 0, s, a, ntiles);
  return hipGetLastError();
}
hipError_t launch_chain_bwd(int L, const ChainBwdArgs& a, hipStream_t s) {
  if (a.rows == 0) return hipSuccess;
  switch (L) {
    case 32: return launch_chain_bwd_t<32>(a, s);
    case 64: return launch_chain_bwd_t<64>(a, s);
    case 128: return launch_chain_bwd_t<128>(a, s);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_colpart_reduce(const float* part, int G, int L, float* out0, int n0, float* out1, int n1, hipStream_t s) {
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3(cdiv(2 * L, 256)), dim3(256), 0, s, part, G, L, out0, n0, out1, n1);
  return hipGetLastError();
}

int wgrad_chunks(int64_t M, int64_t* chunk_out) {
  int64_t chunk = 512;
  int64_t G = cdiv(M, chunk);
  if (G > kWgradMaxChunks) {
    chunk = cdiv(cdiv(M, kWgradMaxChunks), 16) * 16;
    G = cdiv(M, chunk);
  }
  if (G < 1) G = 1;
  if (chunk_out) *chunk_out = chunk;
  return static_cast<int>(G);
}
template <int L>
static hipError_t launch_wgrad_t(const float* A, const float* B, int64_t M, float* dW, int ld, int col0, float* db, float* part,
                                 float* cpart, hipStream_t s) {
  int64_t chunk = 0;
  const int G = wgrad_chunks(M, &chunk);
  hipLaunchKernelGGL(wgrad_kernel<L>, dim3(G), dim3(64 * (L / 32)), 0, s, A, B, M, chunk, part, db ? cpart : nullptr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(L * L + L, 256)), dim3(256), 0, s, part, db ? cpart : nullptr, G, L, dW, ld, col0, db);
  return hipGetLastError();
}
hipError_t launch_wgrad(int L, const float* A, const float* B, int64_t M, float* dW, int ld, int col0, float* db, float* part,
                        float* cpart, hipStream_t s) {
  if (M == 0) return hipSuccess;
  switch (L) {
    case 32: return launch_wgrad_t<32>(A, B, M, dW, ld, col0, db, part, cpart, s);
    case 64: return launch_wgrad_t<64>(A, B, M, dW, ld, col0, db, part, cpart, s);
    case 128: return launch_wgrad_t<128>(A, B, M, dW, ld, col0, db, part, cpart, s);
    default: return hipErrorInvalidValue;
  }
}

int embed_bwd_chunks(int64_t rows, int64_t* chunk_out) {
  int64_t chunk = 256;
  int64_t G = cdiv(rows, chunk);
  if (G > kWgradMaxChunks) {
    chunk = cdiv(rows, kWgradMaxChunks);
    G = cdiv(rows, chunk);
  }
  if (G < 1) G = 1;
  if (chunk_out) *chunk_out = chunk;
  return static_cast<int>(G);
}
hipError_t launch_embed_bwd(int L, const float* d, const float* raw, const int32_t* perm, int64_t rows, int F, float* dW, float* db,
                            float* part, hipStream_t s) {
  if (rows == 0) return hipSuccess;
  int64_t chunk = 0;
  const int G = embed_bwd_chunks(rows, &chunk);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(G), dim3(256), 0, s, d, raw, perm, rows, chunk, L, F, part);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(embed_bwd_reduce_kernel, dim3(cdiv(L * (F + 1), 256)), dim3(256), 0, s, part, G, L, F, dW, db);
  return hipGetLastError();
}

hipError_t launch_segment_reduce(const float* vals, const int32_t* rowptr, const int32_t* idx, int64_t N, int L, int mean, float* out,
                                 hipStream_t s) {
  if (N == 0) return hipSuccess;
  int64_t grid = cdiv(N * (L / 4), 256);
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(segment_reduce_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, s, vals, rowptr, idx, N, L, mean, out);
  return hipGetLastError();
}

template <int L>
static hipError_t launch_rows_gemm_add_t(float* io, const float* in0, const float* w0, const float* in1, const float* w1, int64_t rows,
                                         hipStream_t s) {
  const int64_t ntiles = cdiv(rows, kTile);
  hipLaunchKernelGGL(rows_gemm_add_kernel<L>, dim3(static_cast<unsigned>(cdiv(ntiles, kWavesPerBlock))), dim3(kBlock), 0, s, io, in0, w0,
                     in1, w1, rows, ntiles);
  return hipGetLastError();
}
hipError_t launch_rows_gemm_add(int L, float* io, const float* in0, const float* w0, const float* in1, const float* w1, int64_t rows,
                                hipStream_t s) {
  if (rows == 0) return hipSuccess;
  switch (L) {
    case 32: return launch_rows_gemm_add_t<32>(io, in0, w0, in1, w1, rows, s);
    case 64: return launch_rows_gemm_add_t<64>(io, in0, w0, in1, w1, rows, s);
    case 128: return launch_rows_gemm_add_t<128>(io, in0, w0, in1, w1, rows, s);
    default: return hipErrorInvalidValue;
  }
}

template <int L>
static hipError_t launch_pack_t_t(const float* W, int ld, int col0, float* out, hipStream_t s) {
  hipLaunchKernelGGL(pack_t_kernel<L>, dim3(cdiv(L * L / 4, 256)), dim3(256), 0, s, W, ld, col0, out);
  return hipGetLastError();
}
hipError_t launch_pack_t(int L, const float* W, int ld, int col0, float* out, hipStream_t s) {
  switch (L) {
    case 32: return launch_pack_t_t<32>(W, ld, col0, out, s);
    case 64: return launch_pack_t_t<64>(W, ld, col0, out, s);
    case 128: return launch_pack_t_t<128>(W, ld, col0, out, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace amber
