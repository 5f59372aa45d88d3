// Split-precision ("f16x2") edge kernel, second generation: feature-per-lane accumulators.
//
// Every contraction runs on the fp16 matrix pipe at fp32-grade accuracy: both operands are split into two fp16
// terms, x = x_hi + x_lo (22 significant bits together), and three v_mfma_f32_32x32x16_f16 (hi*hi, hi*lo, lo*hi,
// fp32 accumulate) make one 16-deep k-step; the dropped lo*lo term is 2^-22 relative.  fp16's narrow exponent range
// is handled by exact power-of-two scaling: every weight matrix is scaled at pack time so that max|W| lies in
// [2^14, 2^15), every tile is scaled by a power of two derived from its own max|x|, and the accumulator is multiplied
// by the (power-of-two) inverse afterwards.
//
// Orientation.  The TILE is the MFMA's A operand (row = edge, k = input feature) and the transposed weight matrix its
// B operand (column = output feature), so an accumulator holds  acc[fb][i] = Y[edge 8 (i >> 2) + 4 h + (i & 3)]
// [feature 32 fb + n]  in lane (h = lane >> 5, n = lane & 31): the FEATURE runs across the lanes and the 32 edges of
// the tile across registers and lane halves.  In that layout
//   * one register of a wave is two whole 128-byte row pieces, so the gathered first-layer terms P_s[src] + P_d[dst],
//     the residual rows and the stores of e' go straight between registers and global memory, coalesced, with no
//     trip through LDS;
//   * the per-destination sums of the aggregation are sums over REGISTERS (plus one exchange between the lane
//     halves), with wave-uniform control flow from a bit mask of the row boundaries - no LDS, no atomics;
//   * LayerNorm is a reduction across the 32 lanes of a half (DPP).
// Only the operand that feeds a contraction has to be "edge per lane": the streamed e tile is loaded as whole row
// pieces, split into hi/lo fp16 and written to the wave's 4 KB LDS window as [edge][feature] planes (read back with
// ds_read_b128), and a hidden layer is handed to the next contraction through the same window as a
// [feature][edge] image read with the transposing ds_read_b64_tr_b16.  32 features pass through the window at a time.
//
// Structure: persistent workgroups of 8 waves (two per SIMD) keep all layer weights of the edge MLP in LDS (L = 128:
// 2 x 64 KB) and walk the CSR-ordered edge tiles; each wave owns a tile end to end (no workgroup barrier in the
// loop).  reference: MessagePassingEdgeModule.forward (src/mpn/message_passing_modules.py:38-52),
// MessagePassingBlock.forward (src/mpn/message_passing_block.py:117-121), scatter_mean (message_passing_modules.py:68).
#include "mpn_tile.h"

// experiment switches for tools/exp_dbg.py (timing only, results are wrong with a switch on): build with
// make EXTRA=-DAMBER_DBG_SWITCHES; compiled out otherwise
#ifdef AMBER_DBG_SWITCHES
#define DBG(bit) (a.dbg & (bit))
#else
#define DBG(bit) 0
#endif

namespace amber {

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));

constexpr int kF2Waves = 8;
constexpr int kF2Block = 64 * kF2Waves;
constexpr int kWinBytes = 4096;  // per-wave LDS window: hi plane [0, 2048), lo plane [2048, 4096)
constexpr int kLoPlane = 2048;

DEVINL float wave_max(float m) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  return m;
}
// power-of-two scale that brings m into [2^14, 2^15), and its inverse
DEVINL void pow2_scale(float m, float& scale, float& inv) {
  int ex = static_cast<int>((__float_as_uint(m) >> 23) & 0xffu);
  ex = ex < kMinScaleExp ? kMinScaleExp : ex;
  ex = ex > 254 ? 254 : ex;
  scale = __uint_as_float(static_cast<unsigned>(268 - ex) << 23);
  inv = __uint_as_float(static_cast<unsigned>(ex - 14) << 23);
}
template <int L>
DEVINL float tile_absmax(const Tile<L>& t) {
  float m0 = 0.f, m1 = 0.f;
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb)
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      m0 = fmaxf(m0, fabsf(t.a[fb][i]));
      m1 = fmaxf(m1, fabsf(t.a[fb][i + 1]));
    }
  return fmaxf(m0, m1);
}
DEVINL void split4(const f32x4 x, float scale, half4& hi, half4& lo) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float xs = x[t] * scale;
    const _Float16 hh = static_cast<_Float16>(xs);
    hi[t] = hh;
    lo[t] = static_cast<_Float16>(xs - static_cast<float>(hh));
  }
}
// sum over the 32 lanes of a half-wave, result in every lane of the half
DEVINL float half_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));                   // lane ^ 16
  return v;
}

// transposing LDS read (ds_read_b64_tr_b16): per group of 16 lanes a block of 4 rows x 16 columns of 16-bit elements,
// delivered column-major; lane 4q + p of the group supplies the address of row q, columns 4p .. 4p+3, and lane i
// receives column i (element q = row q).  EXEC must be all ones.
DEVINL half4 lds_read_tr(const unsigned char* p) {
  return __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p)));
}

// streamed rows of the e tile as whole row pieces: v[pass][it] = e[row it*8 + lane/8][32 pass + 4 (lane % 8) .. +3]
template <int L>
struct ELanding {
  f32x4 v[L / 32][4];
};

// one 16-deep k-step of all feature blocks: acc[fb] += A(hi, lo) x W^T(hi, lo), weights in LDS as
// [k-step s][fb][hi | lo][lane] x 16 bytes
template <int L>
DEVINL void kstep(Tile<L>& acc, const half8 ah, const half8 al, const unsigned char* wl, int s, int lane) {
  constexpr int NFB = L / 32;
  const half8* w = reinterpret_cast<const half8*>(wl) + lane;
  half8 bh[NFB], bl[NFB];
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) {
    bh[fb] = w[((s * NFB + fb) * 2 + 0) * 64];
    bl[fb] = w[((s * NFB + fb) * 2 + 1) * 64];
  }
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) acc.a[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[fb], acc.a[fb], 0, 0, 0);
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) acc.a[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[fb], acc.a[fb], 0, 0, 0);
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) acc.a[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[fb], acc.a[fb], 0, 0, 0);
}

template <int L>
__global__ __launch_bounds__(kF2Block) void edge_kernel_f2(const EdgeArgs a, const int64_t ntiles) {
  constexpr int NFB = L / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MAT_BYTES = L * L * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;

  // all layer weights (hi/lo fp16, B-operand order) into LDS, once per workgroup
  for (int m = 0; m < a.nl; ++m) {
    const f32x4* src = reinterpret_cast<const f32x4*>(m == 0 ? a.w0e : a.layers[m].w);
    f32x4* dst = reinterpret_cast<f32x4*>(smem + m * MAT_BYTES);
    for (int i = tid; i < MAT_BYTES / 16; i += kF2Block) dst[i] = src[i];
  }
  __syncthreads();
  unsigned char* win = smem + a.nl * MAT_BYTES + wave * kWinBytes;

  // lane-constant LDS offsets
  // e planes ([edge][32 features] fp16, 64-byte rows, 16-byte slots XOR-swizzled by (row >> 2) & 3)
  int ew_off[4];  // writer: row it*8 + lane/8, 8-byte piece lane % 8
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int rr = it * 8 + (lane >> 3), q = lane & 7;
    ew_off[it] = rr * 64 + ((((q >> 1) ^ ((rr >> 2) & 3))) << 4) + ((q & 1) << 3);
  }
  int er_off[2];  // reader (A operand of k-step sp): row n, slot 2 sp + h
#pragma unroll
  for (int sp = 0; sp < 2; ++sp) er_off[sp] = n * 64 + ((((2 * sp + h) ^ ((n >> 2) & 3))) << 4);
  // hidden-layer image ([feature][32 edges] fp16, 64-byte rows, 8-byte units XOR-swizzled by (row >> 1) & 7)
  int hw_off[4];  // writer: row n, unit 2 g + h
#pragma unroll
  for (int g = 0; g < 4; ++g) hw_off[g] = n * 64 + ((((2 * g + h) ^ ((n >> 1) & 7))) << 3);
  int hr_off[2][2];  // transposing reader: k-step sp, chunk t; this lane supplies row 16 sp + 8 h + 4 t + q, unit 4 colblk + p
  {
    const int gl = lane & 15, q = gl >> 2, p = gl & 3, colblk = (lane >> 4) & 1;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = 16 * sp + 8 * h + 4 * t + q;
        hr_off[sp][t] = row * 64 + ((((4 * colblk + p) ^ ((row >> 1) & 7))) << 3);
      }
  }
  const float wsinv0 = a.w0e[L * L], wscale0 = a.w0e[L * L + 1];  // pack-time power-of-two weight scale (matrix header)
  const float mh0 = h == 0 ? 1.f : 0.f, mh1 = 1.f - mh0;

  // XCD-aware persistent schedule: blocks b, b+8, ... share an XCD (round-robin dispatch); each XCD walks one
  // contiguous eighth of the tiles with its waves side by side, so gathered P_s rows are shared in that L2.
  const int64_t xcd = blockIdx.x & 7, local = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  const int64_t tiles_per_xcd = (ntiles + 7) >> 3;
  const int64_t t_begin = xcd * tiles_per_xcd;
  const int64_t t_end = t_begin + tiles_per_xcd < ntiles ? t_begin + tiles_per_xcd : ntiles;
  const int64_t stride = bpx * kF2Waves;
  int64_t tile = t_begin + local * kF2Waves + wave;

  // lane c (both halves) holds source / destination of tile row c; the next tile's are loaded ahead
  int s0 = 0, d0 = 0, s1 = 0, d1 = 0;
  auto load_idx = [&](int64_t t, int& s, int& d) {
    int64_t p = t * kTile + n;
    if (p > a.E - 1) p = a.E - 1;  // rows beyond the end repeat the last edge and are never stored
    s = a.src[p];
    d = a.dst[p];
  };
  if (tile < t_end) {
    load_idx(tile, s0, d0);
    if (tile + stride < t_end) load_idx(tile + stride, s1, d1);
  }
  for (; tile < t_end; tile += stride) {
    const int64_t p0 = tile * kTile;
    const int nvalid = (a.E - p0) < kTile ? static_cast<int>(a.E - p0) : kTile;  // >= 1: tile < ntiles
    float* const et = a.e + p0 * L;     // this tile's rows of e (wave-uniform base, 32-bit lane offsets)
    const int rlim = nvalid - 1;
    // Lane offsets of the direct row accesses are recomputed per tile from these opaque copies: hoisted out of the tile
    // loop they would occupy dozens of registers for its whole length.
    int hq = 4 * h * L + n, nq = n;
    asm volatile("" : "+v"(hq), "+v"(nq));
    // ---- streamed e rows (whole row pieces) ----
    ELanding<L> la;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = it * 8 + (lane >> 3);
      const f32x4* p = reinterpret_cast<const f32x4*>(et + static_cast<unsigned>((r < rlim ? r : rlim) * L)) + (lane & 7);
#pragma unroll
      for (int ps = 0; ps < NFB; ++ps) la.v[ps][it] = p[ps * 8];
    }
    // ---- gathered first-layer terms, straight into the accumulator layout ----
    Tile<L> acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = 8 * (i >> 2) + (i & 3) + 4 * h;
      const int si = __shfl(s0, r);
      const float* ps = a.Ps + static_cast<int64_t>(DBG(32) ? 0 : si) * L + n;
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) acc.a[fb][i] = ps[32 * fb];
    }
    int s2 = 0, d2 = 0;
    if (tile + 2 * stride < t_end) load_idx(tile + 2 * stride, s2, d2);
    const int firstd = __builtin_amdgcn_readlane(d0, 0);
    const int lastd = __builtin_amdgcn_readlane(d0, nvalid - 1);
    const int rp_first = a.rowptr[firstd], rp_last = a.rowptr[lastd + 1];
    // bit r: row r of the tile begins a new destination (wave-uniform); rows beyond the end never do
    const int dprev = __shfl_up(d0, 1);  // every lane takes part: the shuffle must not sit behind the n > 0 test
    unsigned newseg = static_cast<unsigned>(__ballot((d0 != dprev) & (n > 0)));
    if (nvalid < kTile) newseg &= (1u << nvalid) - 1u;

    // ---- layer 0: acc = (P_s + P_d) + W0e e, in the scaled domain ----
    float scale, inv;
    {
      float m = 0.f;
#pragma unroll
      for (int ps = 0; ps < NFB; ++ps)
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
          for (int t = 0; t < 4; ++t) m = fmaxf(m, fabsf(la.v[ps][it][t]));
      pow2_scale(wave_max(m), scale, inv);
    }
    tile_scale<L>(acc, scale * wscale0);
#pragma unroll
    for (int ps = 0; ps < NFB; ++ps) {
      wave_sync();  // the previous readers of the window are done
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        half4 hi, lo;
        split4(la.v[ps][it], scale, hi, lo);
        *reinterpret_cast<half4*>(win + ew_off[it]) = hi;
        *reinterpret_cast<half4*>(win + kLoPlane + ew_off[it]) = lo;
      }
      wave_sync();
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        const half8 ah = *reinterpret_cast<const half8*>(win + er_off[sp]);
        const half8 al = *reinterpret_cast<const half8*>(win + kLoPlane + er_off[sp]);
        if (!DBG(8)) kstep<L>(acc, ah, al, smem, 2 * ps + sp, lane);
      }
    }
    // destination terms (rows repeat inside a tile: L1 / L2 hits) join after the contraction, when the landing
    // registers of e are free again
    {
      const float us = inv * wsinv0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int di = __shfl(d0, 8 * (i >> 2) + (i & 3) + 4 * h);
        const float* pd = a.Pd + static_cast<int64_t>(DBG(2) ? 0 : di) * L + n;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) acc.a[fb][i] = fmaf(acc.a[fb][i], us, pd[32 * fb]);
      }
    }
    tile_act_fast<L>(acc, a.act);
    // ---- hidden layers: the activations go through the window as a [feature][edge] image ----
#pragma unroll 1
    for (int l = 1; l < a.nl; ++l) {
      const float* wg = a.layers[l].w;
      pow2_scale(wave_max(tile_absmax<L>(acc)), scale, inv);
      Tile<L> nxt;
      {
        const float bs = scale * wg[L * L + 1];
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
          const float b = a.layers[l].b[32 * fb + n] * bs;
#pragma unroll
          for (int i = 0; i < 16; ++i) nxt.a[fb][i] = b;
        }
      }
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) {
        wave_sync();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 x;
          x[0] = acc.a[fb][4 * g + 0];
          x[1] = acc.a[fb][4 * g + 1];
          x[2] = acc.a[fb][4 * g + 2];
          x[3] = acc.a[fb][4 * g + 3];
          half4 hi, lo;
          split4(x, scale, hi, lo);
          *reinterpret_cast<half4*>(win + hw_off[g]) = hi;
          *reinterpret_cast<half4*>(win + kLoPlane + hw_off[g]) = lo;
        }
        wave_sync();
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          const half4 a0 = lds_read_tr(win + hr_off[sp][0]), a1 = lds_read_tr(win + hr_off[sp][1]);
          const half4 b0 = lds_read_tr(win + kLoPlane + hr_off[sp][0]), b1 = lds_read_tr(win + kLoPlane + hr_off[sp][1]);
          half8 ah, al;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            ah[t] = a0[t];
            ah[4 + t] = a1[t];
            al[t] = b0[t];
            al[4 + t] = b1[t];
          }
          if (!DBG(8)) kstep<L>(nxt, ah, al, smem + l * MAT_BYTES, 2 * fb + sp, lane);
        }
      }
      tile_scale<L>(nxt, inv * wg[L * L]);
      tile_act_fast<L>(nxt, a.act);
      acc = nxt;
    }
    // ---- residual: the rows of e again (this wave's own loads of a moment ago), now feature per lane ----
    if (a.residual == AMBER_MODE_INNER && !DBG(1)) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        // rows beyond the end repeat the last edge
        const unsigned off = nvalid == kTile ? static_cast<unsigned>(hq + (8 * (i >> 2) + (i & 3)) * L)
                                             : static_cast<unsigned>(min(8 * (i >> 2) + (i & 3) + 4 * h, rlim) * L + n);
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) acc.a[fb][i] += et[off + 32 * fb];
      }
    }
    // ---- LayerNorm over the features of every edge: across the lanes of a half, one register per edge ----
    if (a.layer_norm == AMBER_MODE_INNER && !DBG(64)) {
      float g[NFB], bt[NFB];
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) {
        g[fb] = a.gamma[32 * fb + n];
        bt[fb] = a.beta[32 * fb + n];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float s = acc.a[0][i];
#pragma unroll
        for (int fb = 1; fb < NFB; ++fb) s += acc.a[fb][i];
        const float mean = half_sum(s) * (1.0f / L);
        float q = 0.f;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
          const float dlt = acc.a[fb][i] - mean;
          acc.a[fb][i] = dlt;
          q += dlt * dlt;
        }
        const float rstd = __builtin_amdgcn_rsqf(half_sum(q) * (1.0f / L) + a.eps);  // v_rsq_f32: 1 ulp
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) acc.a[fb][i] = acc.a[fb][i] * rstd * g[fb] + bt[fb];
      }
    }
    // ---- stores of e' (two whole 128-byte row pieces per register) ----
    if (a.write_e || a.out_edge) {
      int pm = 0;
      if (a.out_edge) {
        int64_t p = p0 + n;
        if (p > a.E - 1) p = a.E - 1;
        pm = a.perm[p];
      }
      if (a.write_e && !DBG(16)) {
        if (nvalid == kTile) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const unsigned off = static_cast<unsigned>(hq + (8 * (i >> 2) + (i & 3)) * L);
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) et[off + 32 * fb] = acc.a[fb][i];
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int r = 8 * (i >> 2) + (i & 3) + 4 * h;
            if (r < nvalid) {
#pragma unroll
              for (int fb = 0; fb < NFB; ++fb) et[static_cast<unsigned>(hq + (8 * (i >> 2) + (i & 3)) * L) + 32 * fb] = acc.a[fb][i];
            }
          }
        }
      }
      if (a.out_edge) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int r = 8 * (i >> 2) + (i & 3) + 4 * h;
          const int orow = __shfl(pm, r);
          if (r < nvalid) {
            float* po = a.out_edge + static_cast<int64_t>(orow) * L + n;
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) po[32 * fb] = acc.a[fb][i];
          }
        }
      }
    }
    // ---- per-destination sums in CSR order: sums over registers, wave-uniform control flow, then the two lane
    // halves are added; rows that lie inside the tile go to agg, cut rows to head / tail (see mpn_kernels.hip) ----
    {
      if (nvalid < kTile) {  // rows beyond the end do not count
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (8 * (i >> 2) + (i & 3) + 4 * h >= nvalid) {
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) acc.a[fb][i] = 0.f;
          }
      }
      const bool ends_last = (p0 + nvalid) == static_cast<int64_t>(rp_last);
      if (!DBG(4)) {
      bool sb = p0 > static_cast<int64_t>(rp_first);
      float sum[NFB];
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) sum[fb] = 0.f;
      auto flush = [&](float* dstp) {
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
          // both halves' partial sums: swap the upper half of one copy with the lower half of the other
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum[fb]), __float_as_uint(sum[fb]), false, false);
          const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
          if (h == 0) dstp[static_cast<unsigned>(nq) + 32 * fb] = tot;
          sum[fb] = 0.f;
        }
      };
#pragma unroll
      for (int r = 0; r < kTile; ++r) {
        if (r > 0 && ((newseg >> r) & 1u)) {
          flush(sb ? a.head + tile * L : a.agg + static_cast<int64_t>(__builtin_amdgcn_readlane(d0, r - 1)) * L);
          sb = false;
        }
        const int i = 4 * (r >> 3) + (r & 3);
        const float mk = ((r >> 2) & 1) ? mh1 : mh0;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) sum[fb] = fmaf(acc.a[fb][i], mk, sum[fb]);
      }
      flush((!sb && ends_last) ? a.agg + static_cast<int64_t>(lastd) * L : (sb ? a.head + tile * L : a.tail + tile * L));
      }
    }
    s0 = s1; d0 = d1;
    s1 = s2; d1 = d2;
  }
}

// ---------------------------------------------------------------------------------------------------
// weight packing: absmax -> power-of-two scale -> hi/lo fp16 in B-operand order (+ header)
// ---------------------------------------------------------------------------------------------------
__global__ void absmax_f2_kernel(const float* __restrict__ W, int ld, int col0, int ncols, int L, unsigned* __restrict__ out_bits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L * ncols) return;
  const float v = fabsf(W[static_cast<int64_t>(i / ncols) * ld + col0 + (i % ncols)]);
  atomicMax(out_bits, __float_as_uint(v));  // non-negative floats order like their bit patterns; integer atomic
}

// out: [s][fb][hi|lo][lane][8 halves]; element j of lane (h, r) is W[32 fb + r][col0 + 16 s + 8 h + j];
// header (8 floats after the L*L*4 body bytes): [0] = inverse scale, [1] = scale
template <int L>
__global__ void pack_f2_kernel(const float* __restrict__ W, int ld, int col0, float* __restrict__ out,
                               const unsigned* __restrict__ absmax_bits) {
  constexpr int NFB = L / 32, NS = L / 16;
  int ex = static_cast<int>((*absmax_bits >> 23) & 0xffu);
  ex = ex < kMinScaleExp ? kMinScaleExp : ex;
  const float scale = __uint_as_float(static_cast<unsigned>(268 - ex) << 23);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (s, fb, lane)
  if (i == 0) {
    out[L * L + 0] = __uint_as_float(static_cast<unsigned>(ex - 14) << 23);
    out[L * L + 1] = scale;
  }
  if (i >= NS * NFB * 64) return;
  const int lane = i & 63, fb = (i >> 6) % NFB, s = (i >> 6) / NFB;
  const float* src = W + static_cast<int64_t>(32 * fb + (lane & 31)) * ld + col0 + 16 * s + 8 * (lane >> 5);
  half8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float ws = src[j] * scale;
    const _Float16 hh = static_cast<_Float16>(ws);
    hi[j] = hh;
    lo[j] = static_cast<_Float16>(ws - static_cast<float>(hh));
  }
  half8* o = reinterpret_cast<half8*>(out);
  o[((s * NFB + fb) * 2 + 0) * 64 + lane] = hi;
  o[((s * NFB + fb) * 2 + 1) * 64 + lane] = lo;
}

template <int L>
static hipError_t launch_pack_f2_t(const float* W, int ld, int col0, int scale_col0, int scale_cols, float* out, hipStream_t s) {
  unsigned* bits = reinterpret_cast<unsigned*>(out + L * L + 4);  // scratch word inside the header
  hipError_t e = hipMemsetAsync(bits, 0, sizeof(unsigned), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(absmax_f2_kernel, dim3((L * scale_cols + 255) / 256), dim3(256), 0, s, W, ld, scale_col0, scale_cols, L, bits);
  hipLaunchKernelGGL(pack_f2_kernel<L>, dim3((L / 16 * (L / 32) * 64 + 255) / 256), dim3(256), 0, s, W, ld, col0, out, bits);
  return hipGetLastError();
}

hipError_t launch_pack_f2(int L, const float* W, int ld, int col0, int scale_col0, int scale_cols, float* out, hipStream_t s) {
  switch (L) {
    case 32: return launch_pack_f2_t<32>(W, ld, col0, scale_col0, scale_cols, out, s);
    case 64: return launch_pack_f2_t<64>(W, ld, col0, scale_col0, scale_cols, out, s);
    case 128: return launch_pack_f2_t<128>(W, ld, col0, scale_col0, scale_cols, out, s);
    default: return hipErrorInvalidValue;
  }
}

size_t edge_f2_lds_bytes(int L, int nl) {
  return static_cast<size_t>(nl) * L * L * 4 + static_cast<size_t>(kF2Waves) * kWinBytes;
}

bool edge_f2_supported(int L, int nl, int act, int residual, int layer_norm) {
  return (act == AMBER_ACT_RELU || act == AMBER_ACT_LEAKYRELU) && residual != AMBER_MODE_OUTER &&
         layer_norm != AMBER_MODE_OUTER && edge_f2_lds_bytes(L, nl) <= 160 * 1024;
}

template <int L>
static hipError_t launch_edge_f2_t(const EdgeArgs& a, int num_cus, hipStream_t s) {
  const int64_t ntiles = (a.E + kTile - 1) / kTile;
  if (ntiles == 0) return hipSuccess;
  const size_t lds = edge_f2_lds_bytes(L, a.nl);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(edge_kernel_f2<L>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds));
  if (e != hipSuccess) return e;
  int64_t grid = (num_cus / 8) * 8;  // one persistent workgroup per CU, a multiple of the 8 XCDs
  if (grid < 8) grid = 8;
  hipLaunchKernelGGL(edge_kernel_f2<L>, dim3(static_cast<unsigned>(grid)), dim3(kF2Block), lds, s, a, ntiles);
  return hipGetLastError();
}

hipError_t launch_edge_f2(int L, const EdgeArgs& a, int num_cus, hipStream_t s) {
  switch (L) {
    case 32: return launch_edge_f2_t<32>(a, num_cus, s);
    case 64: return launch_edge_f2_t<64>(a, num_cus, s);
    case 128: return launch_edge_f2_t<128>(a, num_cus, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace amber
