// bf16x3 operands for the fp16-rate matrix pipe (the backward kernels, mpn_bwd.hip, and the row-per-lane forward kernels at
// latent 256, mpn_kernels.hip).
#pragma once
#include "mpn_tile.h"

namespace amber {

// ---- split operands for the fp16-rate matrix pipe: a float as THREE bf16 pieces (bf16x3) ----
// x = p0 + p1 + p2 exactly: p0 = x truncated to its upper 16 bits, p1 = (x - p0) truncated, p2 = the rest (at most 8
// significant bits are left, so it is a bf16 number).  bf16 keeps fp32's exponent, so - unlike the forward's fp16 hi/lo
// pairs - no scale has to be carried, whatever the magnitude of a gradient row.  A product takes the six piece products
// whose weight is 2^-16 or more (p1 p2, p2 p1, p2 p2 <= 2^-24 are dropped): 6 v_mfma_f32_32x32x16_bf16 of 8 passes for
// 16 k-values against 8 v_mfma_f32_32x32x2_f32 of 16 passes - 2.7 times fewer matrix-pipe cycles, fp32-grade results
// (relative error of a product <= 2^-22).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

DEVINL unsigned pack_upper_halves(float first, float second) {  // (upper 16 bits of second) << 16 | upper 16 bits of first
  return __builtin_amdgcn_perm(__float_as_uint(second), __float_as_uint(first), 0x07060302u);
}
DEVINL void split3(const float (&v)[8], u32x4 (&p)[3]) {
  float r1[8], r2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    r1[i] = v[i] - __uint_as_float(__float_as_uint(v[i]) & 0xffff0000u);
    r2[i] = r1[i] - __uint_as_float(__float_as_uint(r1[i]) & 0xffff0000u);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    p[0][q] = pack_upper_halves(v[2 * q], v[2 * q + 1]);
    p[1][q] = pack_upper_halves(r1[2 * q], r1[2 * q + 1]);
    p[2][q] = pack_upper_halves(r2[2 * q], r2[2 * q + 1]);
  }
}
DEVINL f32x16 mfma_b(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the six piece products of one 32 x 32 x 16 step, small terms first: (piece of the first operand, piece of the second)
__device__ constexpr int kB3W[6] = {2, 0, 1, 1, 0, 0};
__device__ constexpr int kB3X[6] = {0, 2, 1, 0, 1, 0};
#define AMBER_B3_PRODUCTS(ACC, W, X) \
  ACC = mfma_b(W[2], X[0], ACC);     \
  ACC = mfma_b(W[0], X[2], ACC);     \
  ACC = mfma_b(W[1], X[1], ACC);     \
  ACC = mfma_b(W[1], X[0], ACC);     \
  ACC = mfma_b(W[0], X[1], ACC);     \
  ACC = mfma_b(W[0], X[0], ACC)

// acc += W * in with W in bf16x3 operand order (pack_batch_kernel, fmt 1): k-step s takes this lane's registers
// 8 (s & 1) .. + 7 of block s / 2, i.e. features 16 s + 8 (jj / 4) + 4 h + jj % 4 - the accumulator layout is the operand
// layout, nothing moves between lanes.
template <int L>
DEVINL void tile_gemm_b3(Tile<L>& acc, const float* __restrict__ wpacked, const Tile<L>& in, int lane) {
  constexpr int NFB = L / 32, NS = L / 16, GF = NFB < 4 ? NFB : 4;  // output blocks whose weight pieces are held at once
  const u32x4* __restrict__ w = reinterpret_cast<const u32x4*>(wpacked) + lane;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    float v[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) v[jj] = in.a[s >> 1][8 * (s & 1) + jj];
    u32x4 x[3];
    split3(v, x);
#pragma unroll
    for (int g = 0; g < NFB; g += GF) {
      u32x4 wf[GF][3];
#pragma unroll
      for (int fb = 0; fb < GF; ++fb)
#pragma unroll
        for (int q = 0; q < 3; ++q) wf[fb][q] = w[((s * NFB + g + fb) * 3 + q) * 64];
      // piece product outer, output block inner: consecutive instructions are independent chains
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int fb = 0; fb < GF; ++fb) acc.a[g + fb] = mfma_b(wf[fb][kB3W[q]], x[kB3X[q]], acc.a[g + fb]);
    }
#ifdef AMBER_BWD_SCHED_BARRIER
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
}

// The same contraction with the matrix streamed ONCE per workgroup: a k-step's weight pieces (L/32 * 3 * 1 KB) go from
// L2 into a double-buffered LDS window by all threads of the workgroup and every wave takes its operands from there.  At
// latent 128 a wave that reads its matrices itself pulls 96 KB per contraction and 32-row tile through L2 - the chain
// kernel's bound at config 5's size (120 GB a launch).  EVERY wave of the workgroup must make the call (one barrier per
// k-step): the caller's control flow around it has to be workgroup-uniform.
template <int L>
DEVINL void tile_gemm_b3_shared(Tile<L>& acc, const float* __restrict__ wpacked, const Tile<L>& in, int lane, u32x4* wbuf) {
  constexpr int NFB = L / 32, NS = L / 16, STEP = NFB * 3 * 64, PER = STEP / kBlock;
  static_assert(STEP % kBlock == 0, "a k-step's pieces divide over the workgroup's threads");
  const u32x4* __restrict__ wg = reinterpret_cast<const u32x4*>(wpacked);
  // from the caller's lane id, which is opaque per tile: with threadIdx.x itself the weight loads are invariant of the tile loop
  // and the compiler hoists all of them out of it (hundreds of registers)
  const int tid = (static_cast<int>(threadIdx.x) & ~63) | lane;
  u32x4 r[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) r[k] = wg[k * kBlock + tid];
#pragma unroll
  for (int k = 0; k < PER; ++k) wbuf[k * kBlock + tid] = r[k];  // window 0 is free: the previous call ended with a barrier
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (s + 1 < NS) {
#pragma unroll
      for (int k = 0; k < PER; ++k) r[k] = wg[(s + 1) * STEP + k * kBlock + tid];
    }
    float v[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) v[jj] = in.a[s >> 1][8 * (s & 1) + jj];
    u32x4 x[3];
    split3(v, x);
    const u32x4* wl = wbuf + (s & 1) * STEP + lane;
    u32x4 wf[NFB][3];
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int q = 0; q < 3; ++q) wf[fb][q] = wl[(fb * 3 + q) * 64];
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) acc.a[fb] = mfma_b(wf[fb][kB3W[q]], x[kB3X[q]], acc.a[fb]);
    if (s + 1 < NS) {
#pragma unroll
      for (int k = 0; k < PER; ++k) wbuf[((s + 1) & 1) * STEP + k * kBlock + tid] = r[k];
    }
    __syncthreads();  // window (s + 1) & 1 is complete, and window s & 1 has been read by every wave
  }
}

// The same with ONE window (latent 256: a k-step's pieces are 24 KB and the row stages already take 133 KB of the LDS): the
// window is refilled between two barriers, the next k-step's pieces are in flight in registers meanwhile.
template <int L>
DEVINL void tile_gemm_b3_window(Tile<L>& acc, const float* __restrict__ wpacked, const Tile<L>& in, int lane, u32x4* wwin) {
  constexpr int NFB = L / 32, NS = L / 16, STEP = NFB * 3 * 64, PER = STEP / kBlock, GF = 2;
  static_assert(STEP % kBlock == 0 && NFB % GF == 0, "a k-step's pieces divide over the workgroup's threads");
  const u32x4* __restrict__ wg = reinterpret_cast<const u32x4*>(wpacked);
  const int tid = (static_cast<int>(threadIdx.x) & ~63) | lane;
  u32x4 r[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) r[k] = wg[k * kBlock + tid];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    __syncthreads();  // every wave has read the previous k-step's window
#pragma unroll
    for (int k = 0; k < PER; ++k) wwin[k * kBlock + tid] = r[k];
    __syncthreads();
    if (s + 1 < NS) {
#pragma unroll
      for (int k = 0; k < PER; ++k) r[k] = wg[(s + 1) * STEP + k * kBlock + tid];
    }
    float v[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) v[jj] = in.a[s >> 1][8 * (s & 1) + jj];
    u32x4 x[3];
    split3(v, x);
    const u32x4* wl = wwin + lane;
#pragma unroll
    for (int g = 0; g < NFB; g += GF) {
      u32x4 wf[GF][3];
#pragma unroll
      for (int fb = 0; fb < GF; ++fb)
#pragma unroll
        for (int q = 0; q < 3; ++q) wf[fb][q] = wl[((g + fb) * 3 + q) * 64];
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int fb = 0; fb < GF; ++fb) acc.a[g + fb] = mfma_b(wf[fb][kB3W[q]], x[kB3X[q]], acc.a[g + fb]);
    }
  }
}

}  // namespace amber
