// Backward (training) kernels of the message-passing network, gfx950: exact fp32 (v_mfma_f32_32x32x2_f32) or, for
// split-precision models, bf16x3 operands on the fp16-rate pipe (mpn_b3.h; template parameter B3).
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
#include "mpn_b3.h"

namespace amber {

static_assert(kBwdWavesPerBlock == kWavesPerBlock, "pn_part is sized by kBwdWavesPerBlock");

// ---------------------------------------------------------------------------------------------------
// wave-local tile helpers (no workgroup barrier: every wave walks its own tiles)
// ---------------------------------------------------------------------------------------------------
// tile_gemm with a bounded weight prefetch: left alone, the scheduler hoists the loads of ALL k-steps' weight tiles
// (L = 128: 256 registers) above the first MFMA; a scheduling barrier every other k-step keeps two steps in flight.
template <int L>
DEVINL void tile_gemm_lean(Tile<L>& acc, const float* __restrict__ wpacked, const Tile<L>& in, int lane) {
  constexpr int NFB = L / 32, NJ = L / 8;
  const f32x4* __restrict__ w = reinterpret_cast<const f32x4*>(wpacked) + lane;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    f32x4 wf[NFB];
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) wf[fb] = w[(j * NFB + fb) * 64];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float b = in.a[j >> 2][(j & 3) * 4 + t];
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) acc.a[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[fb][t], b, acc.a[fb], 0, 0, 0);
    }
#ifdef AMBER_BWD_SCHED_BARRIER
    if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
#endif
  }
}
template <int L, bool B3>
DEVINL void tile_gemm_sel(Tile<L>& acc, const float* __restrict__ wpacked, const Tile<L>& in, int lane) {
  if constexpr (B3)
    tile_gemm_b3<L>(acc, wpacked, in, lane);
  else
    tile_gemm_lean<L>(acc, wpacked, in, lane);
}
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
// tanh: 1 - a^2; elu: a + 1 = exp(x) on the negative side).  At a == 0 torch's backward takes the negative branch
// (x > 0 ? 1 : slope).
template <int L>
DEVINL void tile_mul_dact(Tile<L>& d, const Tile<L>& a, int act) {
  if (act == AMBER_ACT_TANH) {
#pragma unroll
    for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
      for (int i = 0; i < 16; ++i) d.a[fb][i] *= 1.f - a.a[fb][i] * a.a[fb][i];
  } else if (act == AMBER_ACT_ELU) {
#pragma unroll
    for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
      for (int i = 0; i < 16; ++i) d.a[fb][i] = a.a[fb][i] > 0.f ? d.a[fb][i] : d.a[fb][i] * (a.a[fb][i] + 1.f);
  } else {
    const float slope = act == AMBER_ACT_LEAKYRELU ? 0.01f : 0.f;
#pragma unroll
    for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
      for (int i = 0; i < 16; ++i) d.a[fb][i] = a.a[fb][i] > 0.f ? d.a[fb][i] : slope * d.a[fb][i];
  }
}
// act'(x) from the pre-activation x (torch's formulas: silu, mish with softplus threshold 20, exact gelu)
DEVINL float dact_one(float x, int act) {
  switch (act) {
    case AMBER_ACT_SILU: {
      const float sg = 1.f / (1.f + expf(-x));
      return sg * (1.f + x * (1.f - sg));
    }
    case AMBER_ACT_MISH: {
      const float sp = x > 20.f ? x : log1pf(expf(x));
      const float t = tanhf(sp);
      const float sg = 1.f / (1.f + expf(-x));  // d softplus / dx (1 beyond the threshold to fp32)
      return t + x * (1.f - t * t) * sg;
    }
    case AMBER_ACT_GELU:
      return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
    default:
      return 1.f;
  }
}
// d *= act'(pre) through the wave's LDS stage (one instance of each math routine)
template <int L>
DEVINL void tile_mul_dact_pre(Tile<L>& d, const Tile<L>& pre, int act, float* stage, int lane) {
  wave_sync();
  tile_stage<L>(stage, pre, lane & 31, lane >> 5);
  wave_sync();
  for (int i = lane; i < stage_floats<L>(); i += 64) stage[i] = dact_one(stage[i], act);
  wave_sync();
  Tile<L> f;
  tile_unstage<L>(stage, f, lane & 31, lane >> 5);
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb) d.a[fb] = d.a[fb] * f.a[fb];
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
// a tile set aside in LDS, lane-private (every lane writes and reads its own 16-byte chunks: no synchronisation needed)
template <int L>
DEVINL void tile_park(float* pk, const Tile<L>& t, int lane) {
  f32x4* p = reinterpret_cast<f32x4*>(pk) + lane;
#pragma unroll
  for (int j = 0; j < L / 8; ++j) p[j * 64] = tile_get_chunk<L>(t, j);
}
template <int L>
DEVINL void tile_unpark(Tile<L>& t, const float* pk, int lane) {
  const f32x4* p = reinterpret_cast<const f32x4*>(pk) + lane;
#pragma unroll
  for (int j = 0; j < L / 8; ++j) tile_set_chunk<L>(t, j, p[j * 64]);
}
// make this wave's global stores visible to its own later loads issued from other lanes.  Workgroup scope: the wave's
// stores and loads go through the same vector L1, so ordering them (s_waitcnt) is enough.  Agent scope would be wrong
// for the purpose and ruinous in cost: MI355X's L2s are per XCD, so an agent-scope release writes the whole XCD's dirty
// L2 lines back and the acquire invalidates it - once per tile, that was 80 % of the backward's time.
DEVINL void wave_global_handover() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---------------------------------------------------------------------------------------------------
// chain backward: recompute + chain rule of one module for a 32-row tile.
//   mode EDGE: rows = edges (CSR order); h0 = P_s[src] + P_d[dst] + W0e e;  dy = dE + DM[dst] / deg(dst)
//   mode NODE: rows = nodes; h0 = b0 + W0x x + W0a m;                       dy = dX;  also DM = W0a^T dh0
//   mode HEAD: rows = nodes; decoder chain on x, out = wr . a_last + br;    dy = dout[out_index] * wr
// then a_l = act(W_l a_{l-1} + b_l), r = a_last (+ rin), y = LayerNorm(r) (inner modes).
// EXT = true is a second instantiation for the configurations beside AMBER's default (the default's code generation stays
// what it was):
//   * "outer" residual / LayerNorm (message_passing_block.py:91-97): the module's output that the node module aggregates
//     (e_mid) is not the block's output e_out = [LN](e_mid [+ e]); the gradient from the aggregate joins at e_mid, after
//     the outer LayerNorm's backward.  Node rows have no second consumer, so the host folds their outer modes into the
//     inner forms except residual outer + LayerNorm inner: y = LN(m) + x.
//   * mode NODE without an aggregate (a.m == nullptr): a plain dense chain - the tail of a LatentMLP input embedding
//     (src/mpn/common/embedding.py:35-38).
//   * mode HEAD with a LayerNorm behind every decoder layer (mlp.py:86-93, latent_normalization "layer"): a_l is kept in
//     AP[l], the next layer reads n_l = LN_l(a_l); the (d gamma_l, d beta_l) column sums are added per wave into
//     pn_part[wave][l][2][L] (every wave owns its slots: no atomics, fixed order).
// ---------------------------------------------------------------------------------------------------
template <int L, bool EXT, bool B3>
__global__ __launch_bounds__(kBlock, L <= 64 ? 2 : 1) void chain_bwd_kernel(const ChainBwdArgs a, const int64_t ntiles) {
  __shared__ __attribute__((aligned(16))) float stage_all[kWavesPerBlock * stage_floats<L>()];
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* stage = stage_all + wave * stage_floats<L>();
  constexpr int FPL = (L + 63) / 64;
  const bool active = lane0 * FPL < L;
  const int fo = active ? lane0 * FPL : 0;
  float ps0[FPL], ps1[FPL];
#pragma unroll
  for (int q = 0; q < FPL; ++q) ps0[q] = ps1[q] = 0.f;
  // latent 128 on bf16x3 operands: the matrices are streamed once per workgroup through an LDS window (tile_gemm_b3_shared);
  // its barriers need every wave of the workgroup at every contraction, so the tile loop below is workgroup-uniform (a
  // wave past the last tile works on clamped rows with nvalid <= 0: nothing of it is stored or summed)
  constexpr bool SHARED_W = B3 && L == 128;
  __shared__ __attribute__((aligned(16))) u32x4 wwin[SHARED_W ? 2 * (L / 32) * 3 * 64 : 1];
  // the inner residual's share of the input gradient waits in LDS (PARK) instead of in the gradient buffer: a row-sized
  // store, its hand-over and the re-read less per tile (the bf16x3 instantiations have the LDS for it: 156 KB at latent 128)
  constexpr bool PARK = B3;
  __shared__ __attribute__((aligned(16))) float park_all[PARK ? kWavesPerBlock * L * 32 : 4];
  float* const pk = park_all + (PARK ? wave * L * 32 : 0);
#define GEMM(ACC, W, IN, LANE)                                 \
  do {                                                         \
    if constexpr (SHARED_W)                                    \
      tile_gemm_b3_shared<L>(ACC, W, IN, LANE, wwin);          \
    else                                                       \
      tile_gemm_sel<L, B3>(ACC, W, IN, LANE);                  \
  } while (0)

  const int64_t gw = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  const int64_t G = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  for (int64_t tile = gw; tile - wave < ntiles; tile += G) {
    // The lane id is re-derived from an opaque copy in every iteration: with a loop-invariant lane id the weight tiles,
    // biases and LayerNorm parameters are loop-invariant loads, and hoisted out of the tile loop they occupy (and spill)
    // hundreds of registers.
    int lane = lane0;
#ifndef AMBER_BWD_ALLOW_HOIST
    asm volatile("" : "+v"(lane));
#endif
    const int c = lane & 31, h = lane >> 5;
    const int64_t p0 = tile * kTile;
    const int nvalid = (a.rows - p0) < kTile ? static_cast<int>(a.rows - p0 > 0 ? a.rows - p0 : 0) : kTile;
    int64_t p = p0 + c;
    if (p > a.rows - 1) p = a.rows - 1;  // rows beyond the end repeat the last row; their gradient is zeroed below
    const bool rowok = c < nvalid;
    bool ln_outer = false, has_m = a.mode == AMBER_BWD_NODE, dnorm = false, use_pre = false;
    if constexpr (EXT) {
      ln_outer = a.layer_norm == AMBER_MODE_OUTER;
      has_m = has_m && a.m != nullptr;
      dnorm = a.mode == AMBER_BWD_HEAD && a.pn_gamma[0] != nullptr;
      use_pre = a.mode == AMBER_BWD_HEAD && a.PRE[0] != nullptr;
    }
    // is the residual term inside the LayerNorm's input?  (outer LayerNorm: whenever there is a residual at all)
    const bool res_in_ln = ln_outer ? a.residual != AMBER_MODE_NONE : a.residual == AMBER_MODE_INNER;

    // ---- recompute the forward chain (tiles that are needed again later are re-read, not kept: registers) ----
    Tile<L> cur;
    int d = 0;
    if (a.mode == AMBER_BWD_EDGE) d = a.dst[p];
    {
      Tile<L> rin;
      tile_load_row<L>(rin, a.rin, p, h);
      if (a.nl == 0) {
        cur = rin;
      } else {
        if (a.mode == AMBER_BWD_EDGE) {
          const int s = a.src[p];
          tile_load_row<L>(cur, a.Ps, s, h);
          tile_add_row<L>(cur, a.Pd, d, h);
          GEMM(cur, a.w_in, rin, lane);
        } else {
          tile_load_vec<L>(cur, a.b0, h);
          GEMM(cur, a.w_in, rin, lane);
          if (has_m) {
            tile_load_row<L>(rin, a.m, p, h);
            GEMM(cur, a.w_m, rin, lane);
          }
        }
      }
    }
    if (a.nl > 0) {
      if constexpr (EXT) {
        if (use_pre) tile_store_w<L>(stage, cur, a.PRE[0], p0, nvalid, lane);
      }
      tile_act_w<L>(cur, a.act, stage, lane);
      if constexpr (EXT) {
        if (dnorm) {
          tile_store_w<L>(stage, cur, a.AP[0], p0, nvalid, lane);
          tile_layernorm<L>(cur, a.pn_gamma[0], a.pn_beta[0], a.eps, h);
        }
      }
#pragma unroll 1
      for (int l = 1; l < a.nl; ++l) {
        tile_store_w<L>(stage, cur, a.A[l - 1], p0, nvalid, lane);  // input of layer l: needed by dW_l and by act' below
        Tile<L> nxt;
        tile_load_vec<L>(nxt, a.layers[l].b, h);
        GEMM(nxt, a.layers[l].w, cur, lane);
        if constexpr (EXT) {
          if (use_pre) tile_store_w<L>(stage, nxt, a.PRE[l], p0, nvalid, lane);
        }
        tile_act_w<L>(nxt, a.act, stage, lane);
        cur = nxt;
        if constexpr (EXT) {
          if (dnorm) {
            tile_store_w<L>(stage, cur, a.AP[l], p0, nvalid, lane);
            tile_layernorm<L>(cur, a.pn_gamma[l], a.pn_beta[l], a.eps, h);
          }
        }
      }
    }
    // ---- gradient arriving at the module's output ----
    Tile<L> dr;
    if (a.mode == AMBER_BWD_EDGE) {
      tile_load_row<L>(dr, a.dio, p, h);
      if (!ln_outer) {
        const float inv = 1.0f / static_cast<float>(max(a.rowptr[d + 1] - a.rowptr[d], 1));  // scatter_mean's divisor
        const f32x4* pm = reinterpret_cast<const f32x4*>(a.dm_in + static_cast<int64_t>(d) * L) + h;
#pragma unroll
        for (int j = 0; j < L / 8; ++j) tile_set_chunk<L>(dr, j, tile_get_chunk<L>(dr, j) + inv * pm[2 * j]);
      }
    } else if (a.mode == AMBER_BWD_NODE) {
      tile_load_row<L>(dr, a.dio, p, h);
    } else {
      const int64_t oi = a.out_index ? static_cast<int64_t>(a.out_index[p]) : p;
      const float dscal = (rowok && oi >= 0) ? a.dout[oi] : 0.f;
      // readout gradients: d wr = sum_rows dout * a_last, d br = sum_rows dout
#pragma unroll
      for (int fb = 0; fb < L / 32; ++fb) dr.a[fb] = cur.a[fb] * dscal;
      tile_colsum_acc<L, FPL>(stage, dr, ps0, active, fo, lane);
#pragma unroll
      for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
        for (int i = 0; i < 16; ++i) dr.a[fb][i] = dscal;
      tile_colsum_acc<L, FPL>(stage, dr, ps1, active, fo, lane);
      tile_load_vec<L>(dr, a.wr, h);
#pragma unroll
      for (int fb = 0; fb < L / 32; ++fb) dr.a[fb] = dr.a[fb] * dscal;
    }
    if (!rowok) tile_zero<L>(dr);
    // ---- LayerNorm backward (statistics recomputed) ----
    if (a.layer_norm == AMBER_MODE_INNER || ln_outer) {
      Tile<L> xh = cur;
      if (res_in_ln) tile_add_row<L>(xh, a.rin, p, h);
      const float rstd = tile_ln_xhat<L>(xh, a.eps);
      tile_colsum_acc<L, FPL>(stage, dr, ps1, active, fo, lane);  // d beta
      // g = gamma * dy (in place), row means of g and g * xhat
      {
        const f32x4* gp = reinterpret_cast<const f32x4*>(a.gamma) + h;
        // d gamma: column sums of dy * xhat, staged without a tile of its own
        wave_sync();
        {
          f32x4* sp = reinterpret_cast<f32x4*>(stage + c * stage_stride<L>()) + h;
#pragma unroll
          for (int j = 0; j < L / 8; ++j) sp[2 * j] = tile_get_chunk<L>(dr, j) * tile_get_chunk<L>(xh, j);
        }
        wave_sync();
        if (active) {
          for (int r = 0; r < kTile; ++r)
#pragma unroll
            for (int q = 0; q < FPL; ++q) ps0[q] += stage[r * stage_stride<L>() + fo + q];
        }
#pragma unroll
        for (int j = 0; j < L / 8; ++j) tile_set_chunk<L>(dr, j, tile_get_chunk<L>(dr, j) * gp[2 * j]);
      }
      const float m1 = tile_row_sum<L>(dr) * (1.0f / L);
      const float m2 = tile_row_dot<L>(dr, xh) * (1.0f / L);
#pragma unroll
      for (int fb = 0; fb < L / 32; ++fb) dr.a[fb] = (dr.a[fb] - m1 - xh.a[fb] * m2) * rstd;
    }
    // what joins the input gradient at the end besides W_in^T dh0: the residual term.  Inner residual: the gradient as it
    // stands here, parked in dio.  Outer residual: the gradient that arrived at the block's output (after the outer
    // LayerNorm's backward, if there is one) - dio still holds it unless the LayerNorm changed it.
    const bool res_any = a.residual != AMBER_MODE_NONE;
    if constexpr (EXT) {
      if (ln_outer && a.mode == AMBER_BWD_EDGE) {
        if (a.residual == AMBER_MODE_OUTER) tile_store_w<L>(stage, dr, a.dio, p0, nvalid, lane);
        // the aggregate's gradient joins at e_mid, below the outer LayerNorm
        const float inv = rowok ? 1.0f / static_cast<float>(max(a.rowptr[d + 1] - a.rowptr[d], 1)) : 0.f;
        const f32x4* pm = reinterpret_cast<const f32x4*>(a.dm_in + static_cast<int64_t>(d) * L) + h;
#pragma unroll
        for (int j = 0; j < L / 8; ++j) tile_set_chunk<L>(dr, j, tile_get_chunk<L>(dr, j) + inv * pm[2 * j]);
      }
    }
    // ---- chain rule through the layers, last to first ----
    if (a.nl > 0) {
      const bool keep = a.residual == AMBER_MODE_INNER;
      if (keep) {  // set aside until it joins the input gradient
        if constexpr (PARK)
          tile_park<L>(pk, dr, lane);
        else
          tile_store_w<L>(stage, dr, a.dio, p0, nvalid, lane);
      }
      // rows stored above are re-read by other lanes of this wave
      if (a.nl > 1 || dnorm || use_pre || (res_any && !(PARK && keep))) wave_global_handover();
      // decoder LayerNorm behind layer l: dr arrives at n_l = LN_l(a_l) and leaves at a_l
      auto post_norm_bwd = [&](int l) {
        if constexpr (EXT) {
          tile_load_row<L>(cur, a.AP[l], p, h);
          const float rstd = tile_ln_xhat<L>(cur, a.eps);
          const int64_t gw_l = (gw * a.nl + l) * 2 * L;
          float pg[FPL], pb[FPL];
#pragma unroll
          for (int q = 0; q < FPL; ++q) pg[q] = pb[q] = 0.f;
          tile_colsum_acc<L, FPL>(stage, dr, pb, active, fo, lane);  // d beta_l
          wave_sync();
          {
            f32x4* sp = reinterpret_cast<f32x4*>(stage + c * stage_stride<L>()) + h;
#pragma unroll
            for (int j = 0; j < L / 8; ++j) sp[2 * j] = tile_get_chunk<L>(dr, j) * tile_get_chunk<L>(cur, j);
          }
          wave_sync();
          if (active) {
            for (int r = 0; r < kTile; ++r)
#pragma unroll
              for (int q = 0; q < FPL; ++q) pg[q] += stage[r * stage_stride<L>() + fo + q];
#pragma unroll
            for (int q = 0; q < FPL; ++q) {
              a.pn_part[gw_l + fo + q] += pg[q];
              a.pn_part[gw_l + L + fo + q] += pb[q];
            }
          }
          const f32x4* gp = reinterpret_cast<const f32x4*>(a.pn_gamma[l]) + h;
#pragma unroll
          for (int j = 0; j < L / 8; ++j) tile_set_chunk<L>(dr, j, tile_get_chunk<L>(dr, j) * gp[2 * j]);
          const float m1 = tile_row_sum<L>(dr) * (1.0f / L);
          const float m2 = tile_row_dot<L>(dr, cur) * (1.0f / L);
#pragma unroll
          for (int fb = 0; fb < L / 32; ++fb) dr.a[fb] = (dr.a[fb] - m1 - cur.a[fb] * m2) * rstd;
          tile_load_row<L>(cur, a.AP[l], p, h);  // a_l again, for act'
        }
      };
#pragma unroll 1
      for (int l = a.nl - 1; l >= 1; --l) {
        if (dnorm)
          post_norm_bwd(l);
        else if (l != a.nl - 1)
          tile_load_row<L>(cur, a.A[l], p, h);
        bool from_pre = false;
        if constexpr (EXT) {
          if (use_pre) {
            tile_load_row<L>(cur, a.PRE[l], p, h);
            tile_mul_dact_pre<L>(dr, cur, a.act, stage, lane);
            from_pre = true;
          }
        }
        if (!from_pre) tile_mul_dact<L>(dr, cur, a.act);
        tile_store_w<L>(stage, dr, a.DH[l], p0, nvalid, lane);
        Tile<L> nd;
        tile_zero<L>(nd);
        GEMM(nd, a.wT[l], dr, lane);
        dr = nd;
      }
      if (dnorm)
        post_norm_bwd(0);
      else if (a.nl > 1)
        tile_load_row<L>(cur, a.A[0], p, h);
      {
        bool from_pre = false;
        if constexpr (EXT) {
          if (use_pre) {
            tile_load_row<L>(cur, a.PRE[0], p, h);
            tile_mul_dact_pre<L>(dr, cur, a.act, stage, lane);
            from_pre = true;
          }
        }
        if (!from_pre) tile_mul_dact<L>(dr, cur, a.act);
      }
      tile_store_w<L>(stage, dr, a.DH[0], p0, nvalid, lane);
      if (has_m) {
        tile_zero<L>(cur);
        GEMM(cur, a.wT_m, dr, lane);
        tile_store_w<L>(stage, cur, a.dm_out, p0, nvalid, lane);
      }
      // gradient at the first layer's streamed input (+ the parked residual term; head: + what arrived at the latents)
      if (PARK && keep)
        tile_unpark<L>(cur, pk, lane);
      else if (res_any || a.mode == AMBER_BWD_HEAD)
        tile_load_row<L>(cur, a.dio, p, h);
      else
        tile_zero<L>(cur);
      GEMM(cur, a.wT_in, dr, lane);
      tile_store_w<L>(stage, cur, a.dio, p0, nvalid, lane);
    } else {
      if (a.mode == AMBER_BWD_HEAD) tile_add_row<L>(dr, a.dio, p, h);
      tile_store_w<L>(stage, dr, a.dio, p0, nvalid, lane);
    }
  }
  // partial column sums: the block's four waves are added in wave order through LDS, one partial per block
  if (a.part) {
    __syncthreads();  // every wave is past its last tile: the stages are free
    float* red = stage_all;  // [wave][2][L]
    if (active) {
#pragma unroll
      for (int q = 0; q < FPL; ++q) {
        red[(wave * 2 + 0) * L + fo + q] = ps0[q];
        red[(wave * 2 + 1) * L + fo + q] = ps1[q];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * L; i += kBlock) {
      float sum = red[i];
#pragma unroll
      for (int w = 1; w < kWavesPerBlock; ++w) sum += red[w * 2 * L + i];
      a.part[static_cast<int64_t>(blockIdx.x) * 2 * L + i] = sum;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// fused edge backward (latent <= 64, two layers, inner / no residual and LayerNorm, bf16x3 operands): the chain of
// chain_bwd_kernel in mode EDGE AND the module's weight gradients in one pass over the edges.  The chain kernel's
// intermediates (A_0, DH_1, the parked residual term, and DH_0 / e read again by the weight-gradient kernel) are 11 of
// the 15 edge-sized tensors one step's backward moves through HBM; here A_0, the residual term and the input rows stay
// in registers and every wave keeps its own dW_0e = DH_0^T e and dW_1 = DH_1^T A_0 in 2 (L/32)^2 accumulator tiles - one
// wave per SIMD (512 registers a lane).  The rows are the contraction dimension of those products: both tiles go through
// the wave's two LDS stages and come back transposed (lane (h, m): rows 16 s + 8 h .. + 7 of feature column m), as
// bf16x3 pieces.  What still leaves the kernel per edge: DH_0 (the per-source and per-destination sums need it) and the
// gradient at e.  Per-workgroup partial products go to wgrad_reduce_kernel like the split-K partials of wgrad_kernel.
// ---------------------------------------------------------------------------------------------------
template <int L>
DEVINL void tile_wgrad_b3(f32x16 (&acc)[L / 32][L / 32], const float* sx, const float* sy, int c, int h) {
  constexpr int NFB = L / 32, ST = stage_stride<L>();
#pragma unroll
  for (int s = 0; s < kTile / 16; ++s) {
    u32x4 xp[NFB][3], yp[NFB][3];
    const int r0 = 16 * s + 8 * h;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
      float v[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) v[jj] = sx[(r0 + jj) * ST + 32 * fb + c];
      split3(v, xp[fb]);
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) v[jj] = sy[(r0 + jj) * ST + 32 * fb + c];
      split3(v, yp[fb]);
    }
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int fo = 0; fo < NFB; ++fo)
#pragma unroll
        for (int fi = 0; fi < NFB; ++fi) acc[fo][fi] = mfma_b(xp[fo][kB3W[q]], yp[fi][kB3X[q]], acc[fo][fi]);
  }
}

#undef GEMM

// one per-wave product (accumulator tiles g) of the workgroup's waves added in wave order through LDS and stored as the
// workgroup's partial P[out feature][in feature]; every wave of the workgroup calls it
template <int L>
DEVINL void block_product_store(const f32x16 (&g)[L / 32][L / 32], float* lds, int wave, int lane, float* __restrict__ P) {
  constexpr int NFB = L / 32, TF = NFB * NFB * 16 * 64;
  const int c = lane & 31, h = lane >> 5;
  if (wave > 0) {
    float* mine = lds + static_cast<size_t>(wave - 1) * TF;
#pragma unroll
    for (int i = 0; i < NFB; ++i)
#pragma unroll
      for (int j = 0; j < NFB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[((i * NFB + j) * 16 + r) * 64 + lane] = g[i][j][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < NFB; ++i)
#pragma unroll
      for (int j = 0; j < NFB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = g[i][j][r];
#pragma unroll
          for (int w = 1; w < kWavesPerBlock; ++w) v += lds[static_cast<size_t>(w - 1) * TF + ((i * NFB + j) * 16 + r) * 64 + lane];
          const int m = (r & 3) + 8 * (r >> 2) + 4 * h;  // output feature within block i; input feature 32 j + c
          P[(32 * i + m) * L + 32 * j + c] = v;
        }
  }
  __syncthreads();
}

template <int L>
__global__ __launch_bounds__(kBlock, 1) void edge_bwd_fused_kernel(const ChainBwdArgs a, const int64_t ntiles, float* __restrict__ wpart,
                                                                    float* __restrict__ cpart) {
  extern __shared__ __attribute__((aligned(16))) float lds_fused[];
  constexpr int NFB = L / 32, SF = stage_floats<L>(), ST = stage_stride<L>();
  const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* sx = lds_fused + (2 * wave) * SF;
  float* sy = sx + SF;
  constexpr int FPL = (L + 63) / 64;
  const bool active = lane0 * FPL < L;
  const int fo = active ? lane0 * FPL : 0;
  float ps0[FPL], ps1[FPL], ps2[FPL];
#pragma unroll
  for (int q = 0; q < FPL; ++q) ps0[q] = ps1[q] = ps2[q] = 0.f;
  f32x16 g0[NFB][NFB], g1[NFB][NFB];
#pragma unroll
  for (int i = 0; i < NFB; ++i)
#pragma unroll
    for (int j = 0; j < NFB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) g0[i][j][r] = g1[i][j][r] = 0.f;
  const bool ln = a.layer_norm == AMBER_MODE_INNER, res = a.residual == AMBER_MODE_INNER;
  // W1, W1^T and W0e^T (bf16x3 operand order) live in LDS behind the stages: read per tile from L2 they cost a tenth of
  // the training step, and their loads queue behind the tile's stores
  constexpr int WF = L * L * 3 / 2;
  float* wl = lds_fused + 2 * kWavesPerBlock * SF;
  {
    const float* srcs[3] = {a.layers[1].w, a.wT[1], a.wT_in};
#pragma unroll
    for (int mtx = 0; mtx < 3; ++mtx)
      for (int i = threadIdx.x; i < WF / 4; i += kBlock)
        reinterpret_cast<f32x4*>(wl + mtx * WF)[i] = reinterpret_cast<const f32x4*>(srcs[mtx])[i];
    __syncthreads();
  }
  Tile<L> keep;
  tile_zero<L>(keep);
  bool have_prev = false;
  int64_t prev_p0 = 0;
  int prev_nv = 0;

  const int64_t gw = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  const int64_t G = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  // a tile's endpoints and in-degree are fetched a tile ahead: vector-memory results return in order, so a tile's first
  // loads wait for the previous tile's stores - with the indices already here, the row gathers go out at once
  auto row_of = [&](int64_t t, int c_) {
    const int64_t q = t * kTile + c_;
    return q > a.rows - 1 ? a.rows - 1 : q;
  };
  int dn = 0, sn = 0, degn = 1;
  if (gw < ntiles) {
    const int64_t q = row_of(gw, lane0 & 31);
    dn = a.dst[q];
    sn = a.src[q];
    degn = a.rowptr[dn + 1] - a.rowptr[dn];
  }
  for (int64_t tile = gw; tile < ntiles; tile += G) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));  // see chain_bwd_kernel: keeps the weight tiles from being hoisted out of the loop
    const int c = lane & 31, h = lane >> 5;
    const int64_t p0 = tile * kTile;
    const int nvalid = (a.rows - p0) < kTile ? static_cast<int>(a.rows - p0) : kTile;
    const int64_t p = row_of(tile, c);
    const bool rowok = c < nvalid;
    const int d = dn, sidx = sn, deg = degn;
    // ---- recompute: a0 = act(P_s[src] + P_d[dst] + W0e e), a1 = act(W1 a0 + b1) ----
    Tile<L> rin, a0, a1, dr;
    tile_load_row<L>(rin, a.rin, p, h);
    tile_load_row<L>(a0, a.Ps, sidx, h);
    tile_add_row<L>(a0, a.Pd, d, h);
    tile_load_row<L>(dr, a.dio, p, h);  // the incoming gradient, asked for early
    {
      const int64_t q = row_of(tile + G < ntiles ? tile + G : tile, c);  // branch-free: the last tile asks for its own again
      dn = a.dst[q];
      sn = a.src[q];
    }
    // W0e comes from L2 (its loads go out here, ahead of the previous tile's stores); the other three matrices are in LDS
    tile_gemm_b3<L>(a0, a.w_in, rin, lane);
    if (have_prev) {
      stage_store_rows<L>(sx, a.DH[0], prev_p0, prev_nv, nullptr, lane);
      tile_store_w<L>(sx, keep, a.dio, prev_p0, prev_nv, lane);
    }
    tile_act_w<L>(a0, a.act, sx, lane);
    degn = a.rowptr[dn + 1] - a.rowptr[dn];
    tile_load_vec<L>(a1, a.layers[1].b, h);
    tile_gemm_b3<L>(a1, wl, a0, lane);
    tile_act_w<L>(a1, a.act, sx, lane);
    // ---- gradient at the module's output: dE + DM[dst] / deg(dst) ----
    {
      const float inv = 1.0f / static_cast<float>(max(deg, 1));
      const f32x4* pm = reinterpret_cast<const f32x4*>(a.dm_in + static_cast<int64_t>(d) * L) + h;
#pragma unroll
      for (int j = 0; j < L / 8; ++j) tile_set_chunk<L>(dr, j, tile_get_chunk<L>(dr, j) + inv * pm[2 * j]);
    }
    if (!rowok) tile_zero<L>(dr);
    // ---- LayerNorm backward (as in chain_bwd_kernel) ----
    if (ln) {
      Tile<L> xh = a1;
      if (res) tile_add<L>(xh, rin);
      const float rstd = tile_ln_xhat<L>(xh, a.eps);
      tile_colsum_acc<L, FPL>(sx, dr, ps1, active, fo, lane);  // d beta
      const f32x4* gp = reinterpret_cast<const f32x4*>(a.gamma) + h;
      wave_sync();
      {
        f32x4* sp = reinterpret_cast<f32x4*>(sx + c * ST) + h;
#pragma unroll
        for (int j = 0; j < L / 8; ++j) sp[2 * j] = tile_get_chunk<L>(dr, j) * tile_get_chunk<L>(xh, j);
      }
      wave_sync();
      if (active) {
        for (int r = 0; r < kTile; ++r)
#pragma unroll
          for (int q = 0; q < FPL; ++q) ps0[q] += sx[r * ST + fo + q];  // d gamma
      }
#pragma unroll
      for (int j = 0; j < L / 8; ++j) tile_set_chunk<L>(dr, j, tile_get_chunk<L>(dr, j) * gp[2 * j]);
      const float m1 = tile_row_sum<L>(dr) * (1.0f / L);
      const float m2 = tile_row_dot<L>(dr, xh) * (1.0f / L);
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) dr.a[fb] = (dr.a[fb] - m1 - xh.a[fb] * m2) * rstd;
    }
    Tile<L> keep_now = dr;  // the residual term of the input gradient
    // ---- layer 1: DH1 = dr * act'(a1);  dW1 += DH1^T a0, db1 += column sums;  dr = W1^T DH1 ----
    tile_mul_dact<L>(dr, a1, a.act);
    wave_sync();
    tile_stage<L>(sx, dr, c, h);
    tile_stage<L>(sy, a0, c, h);
    wave_sync();
    if (active) {
      for (int r = 0; r < kTile; ++r)
#pragma unroll
        for (int q = 0; q < FPL; ++q) ps2[q] += sx[r * ST + fo + q];
    }
    tile_wgrad_b3<L>(g1, sx, sy, c, h);
    Tile<L> nd;
    tile_zero<L>(nd);
    tile_gemm_b3<L>(nd, wl + WF, dr, lane);
    // ---- layer 0: DH0 = nd * act'(a0);  dW0e += DH0^T e;  gradient at e = W0e^T DH0 (+ the residual term) ----
    tile_mul_dact<L>(nd, a0, a.act);
    wave_sync();
    tile_stage<L>(sx, nd, c, h);
    tile_stage<L>(sy, rin, c, h);
    wave_sync();
    tile_wgrad_b3<L>(g0, sx, sy, c, h);
    if (!res) tile_zero<L>(keep_now);
    tile_gemm_b3<L>(keep_now, wl + 2 * WF, nd, lane);
    keep = keep_now;
    // DH0 (still staged in sx) and the input gradient (keep) are stored at the top of the wave's next tile, behind that
    // tile's loads: results return in order, so loads queued behind 16 KB of stores would wait for their acknowledgement
    have_prev = true;
    prev_p0 = p0;
    prev_nv = nvalid;
  }
  if (have_prev) {
    stage_store_rows<L>(sx, a.DH[0], prev_p0, prev_nv, nullptr, lane0);
    tile_store_w<L>(sx, keep, a.dio, prev_p0, prev_nv, lane0);
  }
  // ---- per-workgroup partials: column sums first, then the two products, waves added in wave order through LDS ----
  __syncthreads();
  {
    float* red = lds_fused;  // [wave][3][L]
    if (active) {
#pragma unroll
      for (int q = 0; q < FPL; ++q) {
        red[(wave * 3 + 0) * L + fo + q] = ps0[q];
        red[(wave * 3 + 1) * L + fo + q] = ps1[q];
        red[(wave * 3 + 2) * L + fo + q] = ps2[q];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * L; i += kBlock) {
      float sum = red[i];
#pragma unroll
      for (int w = 1; w < kWavesPerBlock; ++w) sum += red[w * 3 * L + i];
      if (i < 2 * L) {
        if (a.part) a.part[static_cast<int64_t>(blockIdx.x) * 2 * L + i] = sum;
      } else {
        cpart[(static_cast<int64_t>(gridDim.x) + blockIdx.x) * L + (i - 2 * L)] = sum;  // job 1 (dW1)'s bias partial
      }
    }
    __syncthreads();
  }
  block_product_store<L>(g0, lds_fused, wave, lane0, wpart + (static_cast<int64_t>(0) * gridDim.x + blockIdx.x) * L * L);
  block_product_store<L>(g1, lds_fused, wave, lane0, wpart + (static_cast<int64_t>(1) * gridDim.x + blockIdx.x) * L * L);
}

// ---------------------------------------------------------------------------------------------------
// fused node backward (same conditions as the fused edge backward): the node-mode chain and its three weight-gradient
// products dW0x = DH0^T x, dW0a = DH0^T m, dW1 = DH1^T A0 (192 accumulator registers at latent 64), one wave per SIMD.
// An airfoil-sized batch has two node tiles a wave: the launch is two tiles' latency instead of the chain kernel's tile
// (store, hand over, re-read) plus a split-K launch.  Matrices come from L2 (with so few tiles a workgroup would load an
// LDS copy for nothing).  cpart slots: job 0 (dW0x) carries db0, job 2 (dW1) db1.
// ---------------------------------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(kBlock, 1) void node_bwd_fused_kernel(const ChainBwdArgs a, const int64_t ntiles, float* __restrict__ wpart,
                                                                    float* __restrict__ cpart) {
  extern __shared__ __attribute__((aligned(16))) float lds_fused[];
  constexpr int NFB = L / 32, SF = stage_floats<L>(), ST = stage_stride<L>();
  const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* sx = lds_fused + (2 * wave) * SF;
  float* sy = sx + SF;
  constexpr int FPL = (L + 63) / 64;
  const bool active = lane0 * FPL < L;
  const int fo = active ? lane0 * FPL : 0;
  float ps0[FPL], ps1[FPL], ps2[FPL], ps3[FPL];
#pragma unroll
  for (int q = 0; q < FPL; ++q) ps0[q] = ps1[q] = ps2[q] = ps3[q] = 0.f;
  f32x16 gx[NFB][NFB], gm[NFB][NFB], g1[NFB][NFB];
#pragma unroll
  for (int i = 0; i < NFB; ++i)
#pragma unroll
    for (int j = 0; j < NFB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) gx[i][j][r] = gm[i][j][r] = g1[i][j][r] = 0.f;
  const bool ln = a.layer_norm == AMBER_MODE_INNER, res = a.residual == AMBER_MODE_INNER;
  const int64_t gw = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + wave;
  const int64_t G = static_cast<int64_t>(gridDim.x) * kWavesPerBlock;
  for (int64_t tile = gw; tile < ntiles; tile += G) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int c = lane & 31, h = lane >> 5;
    const int64_t p0 = tile * kTile;
    const int nvalid = (a.rows - p0) < kTile ? static_cast<int>(a.rows - p0) : kTile;
    int64_t p = p0 + c;
    if (p > a.rows - 1) p = a.rows - 1;
    const bool rowok = c < nvalid;
    // ---- recompute: a0 = act(b0 + W0x x + W0a m), a1 = act(W1 a0 + b1) ----
    // x and m are read again where they are needed later (L2): cheaper than 64 registers held across the tile
    Tile<L> a0, a1, dr;
    tile_load_vec<L>(a0, a.b0, h);
    {
      Tile<L> x, m;
      tile_load_row<L>(x, a.rin, p, h);
      tile_load_row<L>(m, a.m, p, h);
      tile_load_row<L>(dr, a.dio, p, h);  // the incoming gradient, asked for early
      tile_gemm_b3<L>(a0, a.w_in, x, lane);
      tile_gemm_b3<L>(a0, a.w_m, m, lane);
    }
    tile_act_w<L>(a0, a.act, sx, lane);
    tile_load_vec<L>(a1, a.layers[1].b, h);
    tile_gemm_b3<L>(a1, a.layers[1].w, a0, lane);
    tile_act_w<L>(a1, a.act, sx, lane);
    if (!rowok) tile_zero<L>(dr);
    // ---- LayerNorm backward (as in chain_bwd_kernel) ----
    if (ln) {
      Tile<L> xh = a1;
      if (res) tile_add_row<L>(xh, a.rin, p, h);
      const float rstd = tile_ln_xhat<L>(xh, a.eps);
      tile_colsum_acc<L, FPL>(sx, dr, ps1, active, fo, lane);  // d beta
      const f32x4* gp = reinterpret_cast<const f32x4*>(a.gamma) + h;
      wave_sync();
      {
        f32x4* sp = reinterpret_cast<f32x4*>(sx + c * ST) + h;
#pragma unroll
        for (int j = 0; j < L / 8; ++j) sp[2 * j] = tile_get_chunk<L>(dr, j) * tile_get_chunk<L>(xh, j);
      }
      wave_sync();
      if (active) {
        for (int r = 0; r < kTile; ++r)
#pragma unroll
          for (int q = 0; q < FPL; ++q) ps0[q] += sx[r * ST + fo + q];  // d gamma
      }
#pragma unroll
      for (int j = 0; j < L / 8; ++j) tile_set_chunk<L>(dr, j, tile_get_chunk<L>(dr, j) * gp[2 * j]);
      const float m1 = tile_row_sum<L>(dr) * (1.0f / L);
      const float m2 = tile_row_dot<L>(dr, xh) * (1.0f / L);
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) dr.a[fb] = (dr.a[fb] - m1 - xh.a[fb] * m2) * rstd;
    }
    Tile<L> keep = dr;  // the residual term of the input gradient
    // ---- layer 1: DH1 = dr * act'(a1);  dW1 += DH1^T a0, db1 += column sums;  nd = W1^T DH1 ----
    tile_mul_dact<L>(dr, a1, a.act);
    wave_sync();
    tile_stage<L>(sx, dr, c, h);
    tile_stage<L>(sy, a0, c, h);
    wave_sync();
    if (active) {
      for (int r = 0; r < kTile; ++r)
#pragma unroll
        for (int q = 0; q < FPL; ++q) ps2[q] += sx[r * ST + fo + q];
    }
    tile_wgrad_b3<L>(g1, sx, sy, c, h);
    Tile<L> nd;
    tile_zero<L>(nd);
    tile_gemm_b3<L>(nd, a.wT[1], dr, lane);
    // ---- layer 0: DH0 = nd * act'(a0);  dW0x += DH0^T x, dW0a += DH0^T m, db0 += column sums ----
    tile_mul_dact<L>(nd, a0, a.act);
    {
      Tile<L> x, m;
      tile_load_row<L>(x, a.rin, p, h);
      tile_load_row<L>(m, a.m, p, h);
      wave_sync();
      tile_stage<L>(sx, nd, c, h);
      tile_stage<L>(sy, x, c, h);
      wave_sync();
      if (active) {
        for (int r = 0; r < kTile; ++r)
#pragma unroll
          for (int q = 0; q < FPL; ++q) ps3[q] += sx[r * ST + fo + q];
      }
      tile_wgrad_b3<L>(gx, sx, sy, c, h);
      wave_sync();
      tile_stage<L>(sy, m, c, h);
      wave_sync();
      tile_wgrad_b3<L>(gm, sx, sy, c, h);
    }
    // ---- gradients at the aggregate and at x (+ the residual term) ----
    {
      Tile<L> dm;
      tile_zero<L>(dm);
      tile_gemm_b3<L>(dm, a.wT_m, nd, lane);
      tile_store_w<L>(sy, dm, a.dm_out, p0, nvalid, lane);
    }
    if (!res) tile_zero<L>(keep);
    tile_gemm_b3<L>(keep, a.wT_in, nd, lane);
    tile_store_w<L>(sx, keep, a.dio, p0, nvalid, lane);
  }
  // ---- per-workgroup partials ----
  __syncthreads();
  {
    float* red = lds_fused;  // [wave][4][L]
    if (active) {
#pragma unroll
      for (int q = 0; q < FPL; ++q) {
        red[(wave * 4 + 0) * L + fo + q] = ps0[q];
        red[(wave * 4 + 1) * L + fo + q] = ps1[q];
        red[(wave * 4 + 2) * L + fo + q] = ps2[q];
        red[(wave * 4 + 3) * L + fo + q] = ps3[q];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * L; i += kBlock) {
      float sum = red[i];
#pragma unroll
      for (int w = 1; w < kWavesPerBlock; ++w) sum += red[w * 4 * L + i];
      if (i < 2 * L) {
        if (a.part) a.part[static_cast<int64_t>(blockIdx.x) * 2 * L + i] = sum;
      } else if (i < 3 * L) {
        cpart[(static_cast<int64_t>(2) * gridDim.x + blockIdx.x) * L + (i - 2 * L)] = sum;  // job 2 (dW1): db1
      } else {
        cpart[(static_cast<int64_t>(0) * gridDim.x + blockIdx.x) * L + (i - 3 * L)] = sum;  // job 0 (dW0x): db0
      }
    }
    __syncthreads();
  }
  block_product_store<L>(gx, lds_fused, wave, lane0, wpart + (static_cast<int64_t>(0) * gridDim.x + blockIdx.x) * L * L);
  block_product_store<L>(gm, lds_fused, wave, lane0, wpart + (static_cast<int64_t>(1) * gridDim.x + blockIdx.x) * L * L);
  block_product_store<L>(g1, lds_fused, wave, lane0, wpart + (static_cast<int64_t>(2) * gridDim.x + blockIdx.x) * L * L);
}

// out0[f] += sum_g part[g][0][f] (f < n0), out1[f] += sum_g part[g][1][f] (f < n1): one wave per output, lane j adds
// the partials g = j, j + 64, ... in order and the lanes are combined by a fixed butterfly (deterministic)
__global__ __launch_bounds__(64) void colpart_reduce_kernel(const float* __restrict__ part, int G, int64_t gstride, int L,
                                                            float* __restrict__ out0, int n0, float* __restrict__ out1, int n1) {
  const int i = blockIdx.x;  // output (k, f)
  const int k = i / L, f = i % L;
  float* out = k == 0 ? out0 : out1;
  if (!out || f >= (k == 0 ? n0 : n1)) return;
  float s = 0.f;
  for (int g = threadIdx.x; g < G; g += 64) s += part[static_cast<int64_t>(g) * gstride + k * L + f];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if (threadIdx.x == 0) out[f] += s;
}

// ---------------------------------------------------------------------------------------------------
// weight gradient: P[g] = A[rows of chunk g]^T B[rows of chunk g]  (A, B: [M, L] row-major).  The rows are the
// contraction dimension: v_mfma_f32_32x32x2_f32 takes two rows per instruction, and both operands come straight from
// memory, coalesced: lane (h, c) loads row r + h, features VEC c .. VEC c + VEC - 1 (VEC = L / 32 floats, one vector
// load of B per row pair), and MFMA (ta, tb) pairs element ta of the A vector with element tb of the B vector, i.e. the
// 32 x 32 tile of output features {VEC m + ta} x input features {VEC n + tb} - a relabelling of the tiles, undone by
// the store.  A workgroup is 8 waves: GROUPS = VEC / TA groups split the A elements (a wave keeps TA x VEC tiles, 64
// registers at L = 128 / 64) and S = 8 / GROUPS row slices split the chunk's rows; the slices are added in slice order
// through LDS, one partial product per workgroup.
// ---------------------------------------------------------------------------------------------------
template <int N>
struct VecOf {
  typedef float type __attribute__((ext_vector_type(N)));
};
template <>
struct VecOf<1> {
  typedef float type;
};
template <int L>
struct WgradCfg {
  static constexpr int VEC = L / 32;
  static constexpr int TA = L == 64 ? 2 : 1;
  static constexpr int GROUPS = VEC / TA;
  static constexpr int S = 8 / GROUPS;
  static constexpr int TILE_FLOATS = TA * VEC * 16 * 64;  // one wave's accumulators
  static constexpr size_t lds_bytes() { return sizeof(float) * (static_cast<size_t>(S - 1) * GROUPS * (TILE_FLOATS + TA * 32)); }
};

// blockIdx.y selects the job of a batch (products over the same rows M: the weight matrices of one module)
template <int L, bool B3>
__global__ __launch_bounds__(512) void wgrad_kernel(const WgradBatch jobs, int64_t chunk, float* __restrict__ part_all,
                                                    float* __restrict__ cpart_all) {
  const WgradJob jb = jobs.job[blockIdx.y];
  const float* __restrict__ A = jb.A;
  const float* __restrict__ B = jb.B;
  const int64_t M = jobs.M;
  float* __restrict__ part = part_all + static_cast<int64_t>(blockIdx.y) * gridDim.x * L * L;
  float* __restrict__ cpart = jb.db ? cpart_all + static_cast<int64_t>(blockIdx.y) * gridDim.x * L : nullptr;
  using C = WgradCfg<L>;
  constexpr int VEC = C::VEC, TA = C::TA, GROUPS = C::GROUPS, S = C::S, U = 8;
  typedef typename VecOf<VEC>::type vecf;
  typedef typename VecOf<TA>::type veca;
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int grp = wave % GROUPS, slice = wave / GROUPS;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * chunk;
  const int64_t r1 = r0 + chunk < M ? r0 + chunk : M;
  const int64_t sub = ((chunk + S - 1) / S + 1) & ~static_cast<int64_t>(1);  // rows per slice, even
  const int64_t rs0 = r0 + slice * sub;
  const int64_t rs1 = rs0 + sub < r1 ? rs0 + sub : r1;
  f32x16 acc[TA][VEC];
#pragma unroll
  for (int ta = 0; ta < TA; ++ta)
#pragma unroll
    for (int tb = 0; tb < VEC; ++tb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[ta][tb][i] = 0.f;
  float cs[TA];
#pragma unroll
  for (int ta = 0; ta < TA; ++ta) cs[ta] = 0.f;
  if constexpr (B3) {
    // bf16x3 (see split3): 16 rows per instruction; lane (h, c) holds rows r + 8 h .. + 7 of its features
    for (int64_t r = rs0; r < rs1; r += 16) {
      veca av[8];
      vecf bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t rr = r + 8 * h + u;
        const bool ok = rr < rs1;
        const int64_t rc = ok ? rr : rs0;
        veca x = *reinterpret_cast<const veca*>(A + rc * L + VEC * c + TA * grp);
        vecf y = *reinterpret_cast<const vecf*>(B + rc * L + VEC * c);
        if (!ok) {
          x = x * 0.f;
          y = y * 0.f;
        }
        av[u] = x;
        bv[u] = y;
      }
      u32x4 ap[TA][3];
#pragma unroll
      for (int ta = 0; ta < TA; ++ta) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if constexpr (TA == 1) v[u] = av[u]; else v[u] = av[u][ta];
          cs[ta] += v[u];
        }
        split3(v, ap[ta]);
      }
#pragma unroll
      for (int tb = 0; tb < VEC; ++tb) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if constexpr (VEC == 1) v[u] = bv[u]; else v[u] = bv[u][tb];
        }
        u32x4 bp[3];
        split3(v, bp);
#pragma unroll
        for (int ta = 0; ta < TA; ++ta) {
          AMBER_B3_PRODUCTS(acc[ta][tb], ap[ta], bp);
        }
      }
    }
  } else
  for (int64_t r = rs0; r < rs1; r += 2 * U) {
    veca av[U];
    vecf bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t rr = r + 2 * u + h;
      const bool ok = rr < rs1;
      const int64_t rc = ok ? rr : rs0;
      veca x = *reinterpret_cast<const veca*>(A + rc * L + VEC * c + TA * grp);
      vecf y = *reinterpret_cast<const vecf*>(B + rc * L + VEC * c);
      if (!ok) {
        x = x * 0.f;
        y = y * 0.f;
      }
      av[u] = x;
      bv[u] = y;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int ta = 0; ta < TA; ++ta) {
        float xa;
        if constexpr (TA == 1) xa = av[u]; else xa = av[u][ta];
        cs[ta] += xa;
#pragma unroll
        for (int tb = 0; tb < VEC; ++tb) {
          float yb;
          if constexpr (VEC == 1) yb = bv[u]; else yb = bv[u][tb];
          acc[ta][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, yb, acc[ta][tb], 0, 0, 0);
        }
      }
  }
#pragma unroll
  for (int ta = 0; ta < TA; ++ta) cs[ta] += __shfl_xor(cs[ta], 32);
  // ---- add the row slices in slice order ----
  float* cred = red + static_cast<size_t>(S - 1) * GROUPS * C::TILE_FLOATS;
  if (slice > 0) {
    float* mine = red + static_cast<size_t>((slice - 1) * GROUPS + grp) * C::TILE_FLOATS;
#pragma unroll
    for (int ta = 0; ta < TA; ++ta)
#pragma unroll
      for (int tb = 0; tb < VEC; ++tb)
#pragma unroll
        for (int i = 0; i < 16; ++i) mine[((ta * VEC + tb) * 16 + i) * 64 + lane] = acc[ta][tb][i];
    if (h == 0) {
#pragma unroll
      for (int ta = 0; ta < TA; ++ta) cred[((slice - 1) * GROUPS + grp) * TA * 32 + ta * 32 + c] = cs[ta];
    }
  }
  __syncthreads();
  if (slice > 0) return;
  for (int sl = 1; sl < S; ++sl) {
    const float* other = red + static_cast<size_t>((sl - 1) * GROUPS + grp) * C::TILE_FLOATS;
#pragma unroll
    for (int ta = 0; ta < TA; ++ta)
#pragma unroll
      for (int tb = 0; tb < VEC; ++tb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[ta][tb][i] += other[((ta * VEC + tb) * 16 + i) * 64 + lane];
#pragma unroll
    for (int ta = 0; ta < TA; ++ta) cs[ta] += cred[((sl - 1) * GROUPS + grp) * TA * 32 + ta * 32 + c];
  }
  // ---- store: register i of tile (ta, tb) is output feature VEC m + TA grp + ta, input feature VEC c + tb ----
  float* P = part + static_cast<int64_t>(blockIdx.x) * L * L;
#pragma unroll
  for (int ta = 0; ta < TA; ++ta)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = (i & 3) + 8 * (i >> 2) + 4 * h;
      vecf v;
#pragma unroll
      for (int tb = 0; tb < VEC; ++tb) {
        if constexpr (VEC == 1) v = acc[ta][tb][i]; else v[tb] = acc[ta][tb][i];
      }
      *reinterpret_cast<vecf*>(P + (VEC * m + TA * grp + ta) * L + VEC * c) = v;
    }
  if (cpart && h == 0) {
#pragma unroll
    for (int ta = 0; ta < TA; ++ta) cpart[static_cast<int64_t>(blockIdx.x) * L + VEC * c + TA * grp + ta] = cs[ta];
  }
}
// dW[o][col0 + k] += sum_g part[g][o][k];  db[o] += sum_g cpart[g][o], per job (blockIdx.y).  A workgroup takes 64
// consecutive outputs and kRedSlices interleaved subsequences of g (a handful of independent loads a thread instead of
// one long latency-bound loop), combined in a fixed order (deterministic).
constexpr int kRedSlices = 16;
__global__ __launch_bounds__(64 * kRedSlices) void wgrad_reduce_kernel(const WgradBatch jobs, const float* __restrict__ part_all,
                                                                       const float* __restrict__ cpart_all, int G, int L) {
  __shared__ float sm[kRedSlices][64];
  const WgradJob jb = jobs.job[blockIdx.y];
  const int LL = L * L;
  const float* part = part_all + static_cast<int64_t>(blockIdx.y) * G * LL;
  const float* cpart = cpart_all + static_cast<int64_t>(blockIdx.y) * G * L;
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
  const bool is_w = i < LL, is_b = !is_w && i < LL + L && jb.db;
  float s = 0.f;
  if (is_w) {
#pragma unroll 4
    for (int g = sl; g < G; g += kRedSlices) s += part[static_cast<int64_t>(g) * LL + i];
  } else if (is_b) {
#pragma unroll 4
    for (int g = sl; g < G; g += kRedSlices) s += cpart[static_cast<int64_t>(g) * L + (i - LL)];
  }
  sm[sl][threadIdx.x & 63] = s;
  __syncthreads();
  if (sl == 0) {
    const int t = threadIdx.x;
    float tot = sm[0][t];
#pragma unroll
    for (int k = 1; k < kRedSlices; ++k) tot += sm[k][t];
    if (is_w)
      jb.dW[static_cast<int64_t>(i / L) * jb.ld + jb.col0 + (i % L)] += tot;
    else if (is_b)
      jb.db[i - LL] += tot;
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
// The same for few raw features (F <= kEmbFastF, every mesh graph of the reference): a thread owns one output feature o
// and one of 256 / L interleaved row slices and keeps all F + 1 sums in registers, so that d is read ONCE (coalesced
// over o) instead of once per raw feature; the slices are added in slice order through LDS (deterministic).
constexpr int kEmbFastF = 7;
template <int L>
__global__ __launch_bounds__(256) void embed_bwd_fast_kernel(const float* __restrict__ d, const float* __restrict__ raw,
                                                             const int32_t* __restrict__ perm, int64_t rows, int64_t chunk, int F,
                                                             float* __restrict__ part) {
  constexpr int S = 256 / L;  // row slices
  __shared__ float red[S][kEmbFastF + 1][L];
  const int o = threadIdx.x % L, sl = threadIdx.x / L;
  const int64_t r0 = static_cast<int64_t>(blockIdx.x) * chunk;
  const int64_t r1 = r0 + chunk < rows ? r0 + chunk : rows;
  float acc[kEmbFastF + 1];
#pragma unroll
  for (int k = 0; k <= kEmbFastF; ++k) acc[k] = 0.f;
  for (int64_t r = r0 + sl; r < r1; r += S) {
    const float dv = d[r * L + o];
    const float* pr = raw + (perm ? static_cast<int64_t>(perm[r]) : r) * F;
#pragma unroll
    for (int k = 0; k < kEmbFastF; ++k)
      if (k < F) acc[k] += dv * pr[k];
    acc[kEmbFastF] += dv;  // bias column (raw = 1)
  }
#pragma unroll
  for (int k = 0; k <= kEmbFastF; ++k) red[sl][k][o] = acc[k];
  __syncthreads();
  if (sl == 0) {
    float* P = part + static_cast<int64_t>(blockIdx.x) * L * (F + 1);
#pragma unroll
    for (int k = 0; k <= kEmbFastF; ++k) {
      if (k < F || k == kEmbFastF) {
        float sum = red[0][k][o];
#pragma unroll
        for (int q = 1; q < S; ++q) sum += red[q][k][o];
        P[(k == kEmbFastF ? F : k) * L + o] = sum;
      }
    }
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
template <int L, bool B3>
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
  tile_gemm_sel<L, B3>(acc, w0, t, lane);
  if (in1) {
    tile_load_row<L>(t, in1, p, h);
    tile_gemm_sel<L, B3>(acc, w1, t, lane);
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

// The matrices one backward step needs, re-tiled in one launch: blockIdx.y selects the job; plain (pack_kernel order) or
// transposed (pack_t_kernel order) tiles of columns [col0, col0 + L) of W[L, ld]
template <int L>
__global__ void pack_batch_kernel(const PackBatch jobs) {
  constexpr int NFB = L / 32;
  const PackJob jb = jobs.job[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // float4 index
  if (jb.fmt == AMBER_PACK_B3) {
    // bf16x3 operand order (tile_gemm_b3): one thread per (k-step s, output block fb, lane); the lane's eight k-values of
    // the step as three 16-byte piece vectors, 64 lanes of a piece contiguous
    if (i >= (L / 16) * NFB * 64) return;
    const int lane = i & 63, fb = (i >> 6) % NFB, s = (i >> 6) / NFB;
    const int r = 32 * fb + (lane & 31), h = lane >> 5;
    float v[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int k = 16 * s + 8 * (jj >> 2) + 4 * h + (jj & 3);
      v[jj] = jb.transpose ? jb.W[static_cast<int64_t>(k) * jb.ld + jb.col0 + r] : jb.W[static_cast<int64_t>(r) * jb.ld + jb.col0 + k];
    }
    u32x4 pc[3];
    split3(v, pc);
    u32x4* out = reinterpret_cast<u32x4*>(jb.out) + static_cast<int64_t>((s * NFB + fb) * 3) * 64 + lane;
    out[0] = pc[0];
    out[64] = pc[1];
    out[128] = pc[2];
    return;
  }
  if (i >= L * L / 4) return;
  const int lane = i & 63, fb = (i >> 6) % NFB, j = (i >> 6) / NFB;
  const int r = 32 * fb + (lane & 31), k0 = 8 * j + 4 * (lane >> 5);
  f32x4 v;
  if (jb.transpose) {
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = jb.W[static_cast<int64_t>(k0 + t) * jb.ld + jb.col0 + r];
  } else {
    const float* src = jb.W + static_cast<int64_t>(r) * jb.ld + jb.col0 + k0;
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = src[t];
  }
  reinterpret_cast<f32x4*>(jb.out)[i] = v;
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
  const bool ext = a.residual == AMBER_MODE_OUTER || a.layer_norm == AMBER_MODE_OUTER ||
                   (a.mode == AMBER_BWD_NODE && a.m == nullptr) ||
                   (a.mode == AMBER_BWD_HEAD && (a.pn_gamma[0] != nullptr || a.PRE[0] != nullptr));
  const dim3 grid(chain_bwd_grid(a.rows));
  if constexpr (L <= 128) {
    if (a.split) {
      if (ext)
        hipLaunchKernelGGL((chain_bwd_kernel<L, true, true>), grid, dim3(kBlock), 0, s, a, ntiles);
      else
        hipLaunchKernelGGL((chain_bwd_kernel<L, false, true>), grid, dim3(kBlock), 0, s, a, ntiles);
      return hipGetLastError();
    }
  }
  if (a.split) return hipErrorInvalidValue;  // bwd_split_supported(L) is the caller's check
  if (ext)
    hipLaunchKernelGGL((chain_bwd_kernel<L, true, false>), grid, dim3(kBlock), 0, s, a, ntiles);
  else
    hipLaunchKernelGGL((chain_bwd_kernel<L, false, false>), grid, dim3(kBlock), 0, s, a, ntiles);
  return hipGetLastError();
}
hipError_t launch_chain_bwd(int L, const ChainBwdArgs& a, hipStream_t s) {
  if (a.rows == 0) return hipSuccess;
  switch (L) {
    case 32: return launch_chain_bwd_t<32>(a, s);
    case 64: return launch_chain_bwd_t<64>(a, s);
    case 128: return launch_chain_bwd_t<128>(a, s);
    case 256: return launch_chain_bwd_t<256>(a, s);
    default: return hipErrorInvalidValue;
  }
}
bool edge_bwd_fused_supported(int L, const ChainBwdArgs& a) {
  return (L == 32 || L == 64) && a.split && a.mode == AMBER_BWD_EDGE && a.nl == 2 && a.residual != AMBER_MODE_OUTER &&
         a.layer_norm != AMBER_MODE_OUTER;
}
int edge_bwd_fused_grid(int64_t rows) {
  int64_t nblk = cdiv(cdiv(rows, kTile), kWavesPerBlock);
  if (nblk > kFusedBwdBlocks) nblk = kFusedBwdBlocks;
  if (nblk < 1) nblk = 1;
  return static_cast<int>(nblk);
}
template <int L>
static hipError_t launch_edge_bwd_fused_t(const ChainBwdArgs& a, float* wpart, float* cpart, hipStream_t s) {
  const size_t lds = sizeof(float) * (2 * kWavesPerBlock * stage_floats<L>() + 3 * (L * L * 3 / 2));  // stages + three matrices
  static_assert(sizeof(float) * 2 * kWavesPerBlock * stage_floats<L>() >= sizeof(float) * 3 * (L / 32) * (L / 32) * 16 * 64,
                "the end-of-kernel reduction reuses the stages");
  static_assert(sizeof(float) * (2 * kWavesPerBlock * stage_floats<L>() + 3 * (L * L * 3 / 2)) <= 160 * 1024, "LDS of a CU");
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(edge_bwd_fused_kernel<L>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(edge_bwd_fused_kernel<L>, dim3(edge_bwd_fused_grid(a.rows)), dim3(kBlock), lds, s, a, cdiv(a.rows, kTile), wpart, cpart);
  return hipGetLastError();
}
hipError_t launch_edge_bwd_fused(int L, const ChainBwdArgs& a, float* wpart, float* cpart, hipStream_t s) {
  if (a.rows == 0) return hipSuccess;
  if (!edge_bwd_fused_supported(L, a)) return hipErrorInvalidValue;
  return L == 32 ? launch_edge_bwd_fused_t<32>(a, wpart, cpart, s) : launch_edge_bwd_fused_t<64>(a, wpart, cpart, s);
}
bool node_bwd_fused_supported(int L, const ChainBwdArgs& a) {
  return (L == 32 || L == 64) && a.split && a.mode == AMBER_BWD_NODE && a.m != nullptr && a.nl == 2 && a.residual != AMBER_MODE_OUTER &&
         a.layer_norm != AMBER_MODE_OUTER;
}
template <int L>
static hipError_t launch_node_bwd_fused_t(const ChainBwdArgs& a, float* wpart, float* cpart, hipStream_t s) {
  const size_t lds = sizeof(float) * 2 * kWavesPerBlock * stage_floats<L>();
  static_assert(sizeof(float) * 2 * kWavesPerBlock * stage_floats<L>() >= sizeof(float) * 3 * (L / 32) * (L / 32) * 16 * 64,
                "the end-of-kernel reduction reuses the stages");
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(node_bwd_fused_kernel<L>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(node_bwd_fused_kernel<L>, dim3(edge_bwd_fused_grid(a.rows)), dim3(kBlock), lds, s, a, cdiv(a.rows, kTile), wpart, cpart);
  return hipGetLastError();
}
hipError_t launch_node_bwd_fused(int L, const ChainBwdArgs& a, float* wpart, float* cpart, hipStream_t s) {
  if (a.rows == 0) return hipSuccess;
  if (!node_bwd_fused_supported(L, a)) return hipErrorInvalidValue;
  return L == 32 ? launch_node_bwd_fused_t<32>(a, wpart, cpart, s) : launch_node_bwd_fused_t<64>(a, wpart, cpart, s);
}
hipError_t launch_wgrad_reduce(int L, const WgradBatch& b, const float* part, const float* cpart, int G, hipStream_t s) {
  if (b.n == 0) return hipSuccess;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(L * L + L, 64), b.n), dim3(64 * kRedSlices), 0, s, b, part, cpart, G, L);
  return hipGetLastError();
}
hipError_t launch_colpart_reduce(const float* part, int G, int L, float* out0, int n0, float* out1, int n1, hipStream_t s) {
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3(2 * L), dim3(64), 0, s, part, G, static_cast<int64_t>(2 * L), L, out0, n0, out1, n1);
  return hipGetLastError();
}
hipError_t launch_colpart_reduce_strided(const float* part, int G, int64_t gstride, int L, float* out0, float* out1, hipStream_t s) {
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3(2 * L), dim3(64), 0, s, part, G, gstride, L, out0, L, out1, L);
  return hipGetLastError();
}

__global__ void mul_dact_kernel(float* __restrict__ d, const float* __restrict__ a, int64_t n, int act) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const float av = a[i], dv = d[i];
    if (act == AMBER_ACT_TANH)
      d[i] = dv * (1.f - av * av);
    else
      d[i] = av > 0.f ? dv : (act == AMBER_ACT_LEAKYRELU ? 0.01f : 0.f) * dv;
  }
}
hipError_t launch_mul_dact(float* d, const float* a, int64_t n, int act, hipStream_t s) {
  if (n == 0) return hipSuccess;
  int64_t grid = cdiv(n, 256);
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(mul_dact_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, s, d, a, n, act);
  return hipGetLastError();
}

// the mean aggregate exactly as node_kernel forms it from the edge kernel's partial sums (same operations, same order)
__global__ void partials_mean_kernel(const float* __restrict__ agg, const float* __restrict__ head, const float* __restrict__ tail,
                                     const int32_t* __restrict__ rowptr, int red_shift, int64_t N, int L, float* __restrict__ out) {
  const int lpr = L / 4;
  const int64_t total = N * lpr;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t n = i / lpr;
    const int q = static_cast<int>(i % lpr);
    const int r0 = rowptr[n], r1 = rowptr[n + 1];
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (r1 > r0) {
      const int t0 = r0 >> red_shift, t1 = (r1 - 1) >> red_shift;
      if (t0 == t1) {
        g = reinterpret_cast<const f32x4*>(agg + n * L)[q];
      } else {
        g = reinterpret_cast<const f32x4*>(tail + static_cast<int64_t>(t0) * L)[q];
        for (int t = t0 + 1; t <= t1; ++t) g += reinterpret_cast<const f32x4*>(head + static_cast<int64_t>(t) * L)[q];
      }
      g = g / static_cast<float>(r1 - r0);
    }
    reinterpret_cast<f32x4*>(out + n * L)[q] = g;
  }
}
hipError_t launch_partials_mean(const float* agg, const float* head, const float* tail, const int32_t* rowptr, int red_shift,
                                int64_t N, int L, float* out, hipStream_t s) {
  if (N == 0) return hipSuccess;
  int64_t grid = cdiv(N * (L / 4), 256);
  if (grid > 256 * 32) grid = 256 * 32;
  hipLaunchKernelGGL(partials_mean_kernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, s, agg, head, tail, rowptr, red_shift, N, L,
                     out);
  return hipGetLastError();
}

int wgrad_chunks(int64_t M, int njobs, int64_t* chunk_out) {
  const int maxg = kWgradMaxChunks / (njobs < 1 ? 1 : njobs);  // the partial buffers hold kWgradMaxChunks products in all
  int64_t chunk = cdiv(cdiv(M, maxg), 16) * 16;
  if (chunk < 256) chunk = 256;  // larger minimum chunks (fewer partials for N-sized products) measured no faster
  int64_t G = cdiv(M, chunk);
  if (G < 1) G = 1;
  if (chunk_out) *chunk_out = chunk;
  return static_cast<int>(G);
}
template <int L, bool B3>
static hipError_t launch_wgrad_t(const WgradBatch& b, float* part, float* cpart, hipStream_t s) {
  int64_t chunk = 0;
  const int G = wgrad_chunks(b.M, b.n, &chunk);
  const size_t lds = WgradCfg<L>::lds_bytes();
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<L, B3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((wgrad_kernel<L, B3>), dim3(G, b.n), dim3(512), lds, s, b, chunk, part, cpart);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(L * L + L, 64), b.n), dim3(64 * kRedSlices), 0, s, b, part, cpart, G, L);
  return hipGetLastError();
}
hipError_t launch_wgrad_batch(int L, const WgradBatch& b, float* part, float* cpart, hipStream_t s) {
  if (b.M == 0 || b.n == 0) return hipSuccess;
  if (b.n > kWgradMaxJobs) return hipErrorInvalidValue;
  if (b.split && !bwd_split_supported(L)) return hipErrorInvalidValue;
  switch (L) {
    case 32: return b.split ? launch_wgrad_t<32, true>(b, part, cpart, s) : launch_wgrad_t<32, false>(b, part, cpart, s);
    case 64: return b.split ? launch_wgrad_t<64, true>(b, part, cpart, s) : launch_wgrad_t<64, false>(b, part, cpart, s);
    case 128: return b.split ? launch_wgrad_t<128, true>(b, part, cpart, s) : launch_wgrad_t<128, false>(b, part, cpart, s);
    case 256: return launch_wgrad_t<256, false>(b, part, cpart, s);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_wgrad(int L, int split, const float* A, const float* B, int64_t M, float* dW, int ld, int col0, float* db,
                        float* part, float* cpart, hipStream_t s) {
  WgradBatch b{};
  b.n = 1;
  b.M = M;
  b.split = split;
  b.job[0] = WgradJob{A, B, dW, db, ld, col0};
  return launch_wgrad_batch(L, b, part, cpart, s);
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
  if (F <= kEmbFastF && L == 32)
    hipLaunchKernelGGL(embed_bwd_fast_kernel<32>, dim3(G), dim3(256), 0, s, d, raw, perm, rows, chunk, F, part);
  else if (F <= kEmbFastF && L == 64)
    hipLaunchKernelGGL(embed_bwd_fast_kernel<64>, dim3(G), dim3(256), 0, s, d, raw, perm, rows, chunk, F, part);
  else if (F <= kEmbFastF && L == 128)
    hipLaunchKernelGGL(embed_bwd_fast_kernel<128>, dim3(G), dim3(256), 0, s, d, raw, perm, rows, chunk, F, part);
  else
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

template <int L, bool B3>
static hipError_t launch_rows_gemm_add_t(float* io, const float* in0, const float* w0, const float* in1, const float* w1, int64_t rows,
                                         hipStream_t s) {
  const int64_t ntiles = cdiv(rows, kTile);
  hipLaunchKernelGGL((rows_gemm_add_kernel<L, B3>), dim3(static_cast<unsigned>(cdiv(ntiles, kWavesPerBlock))), dim3(kBlock), 0, s, io,
                     in0, w0, in1, w1, rows, ntiles);
  return hipGetLastError();
}
hipError_t launch_rows_gemm_add(int L, int split, float* io, const float* in0, const float* w0, const float* in1, const float* w1,
                                int64_t rows, hipStream_t s) {
  if (rows == 0) return hipSuccess;
  if (split && !bwd_split_supported(L)) return hipErrorInvalidValue;
  switch (L) {
    case 32: return split ? launch_rows_gemm_add_t<32, true>(io, in0, w0, in1, w1, rows, s) : launch_rows_gemm_add_t<32, false>(io, in0, w0, in1, w1, rows, s);
    case 64: return split ? launch_rows_gemm_add_t<64, true>(io, in0, w0, in1, w1, rows, s) : launch_rows_gemm_add_t<64, false>(io, in0, w0, in1, w1, rows, s);
    case 128: return split ? launch_rows_gemm_add_t<128, true>(io, in0, w0, in1, w1, rows, s) : launch_rows_gemm_add_t<128, false>(io, in0, w0, in1, w1, rows, s);
    case 256: return launch_rows_gemm_add_t<256, false>(io, in0, w0, in1, w1, rows, s);
    default: return hipErrorInvalidValue;
  }
}

template <int L>
static hipError_t launch_pack_batch_t(const PackBatch& b, hipStream_t s) {
  hipLaunchKernelGGL(pack_batch_kernel<L>, dim3(cdiv(L * L / 4, 256), b.n), dim3(256), 0, s, b);
  return hipGetLastError();
}
hipError_t launch_pack_batch(int L, const PackBatch& b, hipStream_t s) {
  if (b.n == 0) return hipSuccess;
  if (b.n > kPackMaxJobs) return hipErrorInvalidValue;
  switch (L) {
    case 32: return launch_pack_batch_t<32>(b, s);
    case 64: return launch_pack_batch_t<64>(b, s);
    case 128: return launch_pack_batch_t<128>(b, s);
    case 256: return launch_pack_batch_t<256>(b, s);  // forward only (amber_mpn_pack_weights): latent 256 has no backward
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
    case 256: return launch_pack_t_t<256>(W, ld, col0, out, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace amber
