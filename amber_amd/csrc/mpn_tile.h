// Device-side tile helpers shared by the kernel translation units (see the layout note in mpn_kernels.hip).
#pragma once
#include "mpn_internal.h"

namespace amber {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define DEVINL __device__ __forceinline__

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = 64 * kWavesPerBlock;

template <int L>
struct Tile {
  f32x16 a[L / 32];
};

template <int L>
constexpr int stage_stride() {
  return L + 4;  // floats; +16 B per row keeps 16-byte alignment and spreads rows over LDS banks
}
template <int L>
constexpr int stage_floats() {
  return kTile * stage_stride<L>();
}

// ---------------------------------------------------------------------------------------------------
// activations (torch defaults)
// ---------------------------------------------------------------------------------------------------
DEVINL float act_one(float v, int act) {
  switch (act) {
    case AMBER_ACT_RELU:
      return v > 0.f ? v : 0.f;
    case AMBER_ACT_LEAKYRELU:
      return v > 0.f ? v : 0.01f * v;
    case AMBER_ACT_TANH:
      return tanhf(v);
    case AMBER_ACT_ELU:
      return v > 0.f ? v : expm1f(v);
    case AMBER_ACT_SILU:
      return v / (1.f + expf(-v));
    case AMBER_ACT_MISH: {
      const float sp = v > 20.f ? v : log1pf(expf(v));  // softplus, threshold 20
      return v * tanhf(sp);
    }
    case AMBER_ACT_GELU:
      return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    default:
      return v;
  }
}

// ---------------------------------------------------------------------------------------------------
// tile helpers (all loops fully unrolled: register indices must be compile-time constants)
// ---------------------------------------------------------------------------------------------------
template <int L>
DEVINL void tile_set_chunk(Tile<L>& t, int j, f32x4 v) {
  t.a[j >> 2][(j & 3) * 4 + 0] = v[0];
  t.a[j >> 2][(j & 3) * 4 + 1] = v[1];
  t.a[j >> 2][(j & 3) * 4 + 2] = v[2];
  t.a[j >> 2][(j & 3) * 4 + 3] = v[3];
}
template <int L>
DEVINL f32x4 tile_get_chunk(const Tile<L>& t, int j) {
  f32x4 v;
  v[0] = t.a[j >> 2][(j & 3) * 4 + 0];
  v[1] = t.a[j >> 2][(j & 3) * 4 + 1];
  v[2] = t.a[j >> 2][(j & 3) * 4 + 2];
  v[3] = t.a[j >> 2][(j & 3) * 4 + 3];
  return v;
}

template <int L>
DEVINL void tile_zero(Tile<L>& t) {
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
    for (int i = 0; i < 16; ++i) t.a[fb][i] = 0.f;
}

// t = rows[row][:]  (this lane's chunks of one row-major row of L floats)
template <int L>
DEVINL void tile_load_row(Tile<L>& t, const float* __restrict__ base, int64_t row, int h) {
  const f32x4* p = reinterpret_cast<const f32x4*>(base + row * L) + h;
#pragma unroll
  for (int j = 0; j < L / 8; ++j) tile_set_chunk<L>(t, j, p[2 * j]);
}
template <int L>
DEVINL void tile_add_row(Tile<L>& t, const float* __restrict__ base, int64_t row, int h) {
  const f32x4* p = reinterpret_cast<const f32x4*>(base + row * L) + h;
#pragma unroll
  for (int j = 0; j < L / 8; ++j) tile_set_chunk<L>(t, j, tile_get_chunk<L>(t, j) + p[2 * j]);
}
// t[c][f] = vec[f] for every row (bias as initial accumulator)
template <int L>
DEVINL void tile_load_vec(Tile<L>& t, const float* __restrict__ vec, int h) {
  const f32x4* p = reinterpret_cast<const f32x4*>(vec) + h;
#pragma unroll
  for (int j = 0; j < L / 8; ++j) tile_set_chunk<L>(t, j, p[2 * j]);
}
template <int L>
DEVINL void tile_add(Tile<L>& t, const Tile<L>& o) {
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb) t.a[fb] += o.a[fb];
}

// acc += W * in, W an L x L matrix pre-tiled by pack_kernel; exact fp32 MFMA (a k-ordered fmaf chain).
template <int L>
DEVINL void tile_gemm(Tile<L>& acc, const float* __restrict__ wpacked, const Tile<L>& in, int lane) {
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
  }
}

// LayerNorm over the L features of every row: the row lives in lanes (0, c) and (1, c).
template <int L>
DEVINL void tile_layernorm(Tile<L>& t, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int h) {
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
  const float mean = s * (1.0f / L);
  s0 = s1 = s2 = s3 = 0.f;
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
      const float d0 = t.a[fb][i + 0] - mean, d1 = t.a[fb][i + 1] - mean;
      const float d2 = t.a[fb][i + 2] - mean, d3 = t.a[fb][i + 3] - mean;
      s0 += d0 * d0;
      s1 += d1 * d1;
      s2 += d2 * d2;
      s3 += d3 * d3;
    }
  float q = (s0 + s1) + (s2 + s3);
  q += __shfl_xor(q, 32);
  const float rstd = 1.0f / sqrtf(q * (1.0f / L) + eps);
  const f32x4* g = reinterpret_cast<const f32x4*>(gamma) + h;
  const f32x4* b = reinterpret_cast<const f32x4*>(beta) + h;
#pragma unroll
  for (int j = 0; j < L / 8; ++j) {
    const f32x4 v = tile_get_chunk<L>(t, j);
    tile_set_chunk<L>(t, j, (v - mean) * rstd * g[2 * j] + b[2 * j]);
  }
}
// per-feature affine v * p + q (eval-mode BatchNorm folded on the host)
template <int L>
DEVINL void tile_affine(Tile<L>& t, const float* __restrict__ p, const float* __restrict__ q, int h) {
  const f32x4* g = reinterpret_cast<const f32x4*>(p) + h;
  const f32x4* b = reinterpret_cast<const f32x4*>(q) + h;
#pragma unroll
  for (int j = 0; j < L / 8; ++j) tile_set_chunk<L>(t, j, tile_get_chunk<L>(t, j) * g[2 * j] + b[2 * j]);
}

// tile <-> LDS staging area of the wave, row-major [32][L + 4]
template <int L>
DEVINL void tile_stage(float* stage, const Tile<L>& t, int c, int h) {
  f32x4* p = reinterpret_cast<f32x4*>(stage + c * stage_stride<L>()) + h;
#pragma unroll
  for (int j = 0; j < L / 8; ++j) p[2 * j] = tile_get_chunk<L>(t, j);
}
template <int L>
DEVINL void tile_unstage(const float* stage, Tile<L>& t, int c, int h) {
  const f32x4* p = reinterpret_cast<const f32x4*>(stage + c * stage_stride<L>()) + h;
#pragma unroll
  for (int j = 0; j < L / 8; ++j) tile_set_chunk<L>(t, j, p[2 * j]);
}

// Activation on a tile.  relu / leakyrelu stay in registers; the transcendental ones take a round trip
// through the wave's LDS staging area so that each math routine is instantiated once (block-uniform branch).
template <int L>
DEVINL void tile_act(Tile<L>& t, int act, float* stage, int lane) {
  if (act == AMBER_ACT_RELU) {
#pragma unroll
    for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
      for (int i = 0; i < 16; ++i) t.a[fb][i] = fmaxf(t.a[fb][i], 0.f);
  } else if (act == AMBER_ACT_LEAKYRELU) {
#pragma unroll
    for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float v = t.a[fb][i];
        t.a[fb][i] = v > 0.f ? v : 0.01f * v;
      }
  } else {
    const int c = lane & 31, h = lane >> 5;
    __syncthreads();
    tile_stage<L>(stage, t, c, h);
    __syncthreads();
    for (int i = lane; i < stage_floats<L>(); i += 64) stage[i] = act_one(stage[i], act);
    __syncthreads();
    tile_unstage<L>(stage, t, c, h);
  }
}

// Store the staged tile as whole rows: out[row(r)][:] = stage[r][:], r < nvalid.  rowmap == nullptr: rows
// row0 + r; otherwise rows rowmap[row0 + r] (un-permute to edge_index order).
template <int L>
DEVINL void stage_store_rows(const float* stage, float* __restrict__ out, int64_t row0, int nvalid,
                             const int32_t* __restrict__ rowmap, int lane) {
  constexpr int LPR = L / 4;        // lanes per row (16 B each)
  constexpr int RPI = 64 / LPR;     // rows per wave instruction
  const int q = lane % LPR;
#pragma unroll
  for (int it = 0; it < kTile / RPI; ++it) {
    const int r = it * RPI + lane / LPR;
    if (r < nvalid) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * stage_stride<L>() + 4 * q);
      const int64_t dr = rowmap ? static_cast<int64_t>(rowmap[row0 + r]) : row0 + r;
      *reinterpret_cast<f32x4*>(out + dr * L + 4 * q) = v;
    }
  }
}

// XCD-aware block remap: blocks b and b + 8 share an XCD (round-robin dispatch), so give each XCD a
// contiguous range of tiles; neighbouring tiles gather overlapping P_s rows and then share an L2.
DEVINL int64_t xcd_remap(int64_t bid, int64_t nblk) {
  const int64_t xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}


// ---- split-precision (f16x2) helpers, see the note at the top of mpn_f2.hip ----
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

DEVINL void wave_sync() {
  // LDS instructions of one wave execute in order; this only stops the compiler from moving LDS accesses of
  // different lanes' data across the hand-over point.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kMinScaleExp = 127 - 24;

// t *= factor (a power of two: exact)
template <int L>
DEVINL void tile_scale(Tile<L>& t, float factor) {
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb) t.a[fb] = t.a[fb] * factor;
}

// The same two operations element by element, for tiles that arrived as row pieces (a lane's L/32 consecutive features of
// one row per load): their registers sit in [row][feature block] order, and the vector forms above - packed fp32
// instructions on register pairs [feature block][row, row + 1] - make the compiler gather every pair with two moves first
// (3 instructions for 2 values; measured in the ISA).  Scalar: one instruction a value, the "transpose" is free.
template <int L>
DEVINL void tile_scale_rows(Tile<L>& t, float factor) {
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float r = t.a[fb][i] * factor;
      asm("" : "+v"(r));  // opaque to the SLP vectoriser, which would pair the rows again (and gather the pairs with moves)
      t.a[fb][i] = r;
    }
}
template <int L>
DEVINL void tile_add_rows(Tile<L>& t, const Tile<L>& o) {
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float r = t.a[fb][i] + o.a[fb][i];
      asm("" : "+v"(r));
      t.a[fb][i] = r;
    }
}

template <int L>
DEVINL void tile_act_fast(Tile<L>& t, int act) {
  const float slope = act == AMBER_ACT_LEAKYRELU ? 0.01f : 0.f;
#ifdef AMBER_ACT_PK
  // experiment of round 2 (reverted there: 1e-3 off on 10-20-step networks), kept behind a switch to find the cause
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb) {
    const f32x16 sv = t.a[fb] * slope;
#pragma unroll
    for (int i = 0; i < 16; ++i) t.a[fb][i] = fmaxf(t.a[fb][i], sv[i]);
  }
#else
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float v = t.a[fb][i];
      t.a[fb][i] = fmaxf(v, slope * v);  // slope in [0, 1): max picks v for v > 0 and slope * v otherwise
    }
#endif
}

// t = act(t * post + bias) with relu / leakyrelu, the multiply-adds and the slope multiply as packed fp32 operations on
// the accumulator's register pairs (rows 2j, 2j + 1 of one feature block): half the instructions of the scalar form, the
// same roundings (fused multiply-add, one multiply).  AMBER_EPI_PK selects it in the kernels' layer epilogues.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int L>
DEVINL void tile_bias_act_pk(Tile<L>& t, float post, const float (&bv)[L / 32], int act) {
  const float slope = act == AMBER_ACT_LEAKYRELU ? 0.01f : 0.f;
  const f32x2 post2 = {post, post}, slope2 = {slope, slope};
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb) {
    const f32x2 b2 = {bv[fb], bv[fb]};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f32x2 p = {t.a[fb][2 * j], t.a[fb][2 * j + 1]};
      p = __builtin_elementwise_fma(p, post2, b2);
      const f32x2 q = p * slope2;
      t.a[fb][2 * j] = fmaxf(p.x, q.x);
      t.a[fb][2 * j + 1] = fmaxf(p.y, q.y);
    }
  }
}

// tanh on accumulator registers for the feature-per-lane kernels (no LDS trip, no library call), two ranges that share one
// v_rcp_f32:  |x| < 0.8: the Pade [5/4] form as a CORRECTION to x,  tanh x = x - x x^2 (315 + 14 x^2) / (945 + 420 x^2 + 15 x^4)
// (truncation < 5e-9 there; the roundings only touch the correction, so tiny arguments keep their relative accuracy);
// otherwise sign(x) (1 - 2 / (exp(2 |x|) + 1)) with v_exp_f32 - exp overflows to +inf for |x| > 44, the quotient becomes 0 and
// the result saturates at 1.  Relative error <= ~2e-7 everywhere (checked against float64 over [-6, 6] in steps of 6e-6 and
// around 0).  reference: torch.nn.Tanh in LatentMLP (src/mpn/common/latent_mlp.py:76-83)
DEVINL float tanh_fast(float x) {
  const float ax = fabsf(x), x2 = x * x;
  const float n = fmaf(14.0f, x2, 315.0f);
  const float d = fmaf(fmaf(15.0f, x2, 420.0f), x2, 945.0f);
  const float t = __builtin_amdgcn_exp2f(2.885390081777927f * ax);  // exp(2 |x|)
  const bool small = ax < 0.8f;
  const float r = __builtin_amdgcn_rcpf(small ? d : t + 1.0f);
  const float ra = fmaf(-x, x2 * (n * r), x);
  const float rb = copysignf(fmaf(-2.0f, r, 1.0f), x);
  return small ? ra : rb;
}
template <int L>
DEVINL void tile_tanh(Tile<L>& t) {
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
    for (int i = 0; i < 16; ++i) t.a[fb][i] = tanh_fast(t.a[fb][i]);
}
// relu / leakyrelu / tanh (the "general" instantiations of the feature-per-lane kernels)
template <int L>
DEVINL void tile_act_var(Tile<L>& t, int act) {
  if (act == AMBER_ACT_TANH) {
    tile_tanh<L>(t);
  } else {
    tile_act_fast<L>(t, act);
  }
}

}  // namespace amber
