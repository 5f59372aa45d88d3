// Split-precision ("f16x2") edge kernel: the same wave-tile chain as edge_kernel<L> (mpn_kernels.hip), with every
// fp32 contraction carried out on the fp16 matrix pipe at fp32-grade accuracy.
//
// Both operands are split into two fp16 terms, x = x_hi + x_lo (22 significant bits together), and three MFMAs
// (v_mfma_f32_32x32x16_f16: hi*hi, hi*lo, lo*hi, fp32 accumulate) replace the eight v_mfma_f32_32x32x2_f32 of the
// exact path for the same 16-deep k-step; the dropped lo*lo term is 2^-22 relative.  fp16's narrow exponent range is
// handled by exact power-of-two scaling: every weight matrix is scaled at pack time so that max|W| lies in
// [2^14, 2^15), every tile row is scaled per MFMA column (= per lane, so it is free in this layout) by its own
// max|x|, and the accumulator column is multiplied by the inverse (also a power of two) afterwards.
//
// Structure: persistent workgroups of 8 waves (two per SIMD, <= 256 registers each) keep all layer weights of the
// edge MLP in LDS (L = 128: 2 x 64 KB) and walk the CSR-ordered edge tiles; each wave owns a tile end to end, so
// there is no workgroup barrier in the loop and the two waves of a SIMD overlap each other's memory and MFMA phases.
// All row traffic is whole 128-byte row pieces, transposed to / from the accumulator layout through a 4 KB
// per-wave LDS window (XOR-swizzled 16-byte slots); the epilogue sums per destination in CSR order with
// wave-uniform control flow, 32 features a pass.  (A one-wave-per-SIMD variant with the next tile prefetched into
// registers was tried: hipcc spilled the landing areas to scratch and it ran 1.6x slower.)
#include "mpn_tile.h"

namespace amber {

constexpr int kFastWaves = 8;  // two waves per SIMD (<= 256 registers each) overlap each other's memory and MFMA phases
constexpr int kFastBlock = 64 * kFastWaves;

// Coalesced row traffic.  A lane-per-row 16-byte load touches 32 different rows per wave instruction; instead every
// wave instruction reads whole row pieces of PF features (PF * 4 bytes, whole 128-byte lines) into a register
// "landing" area, and the pieces are turned into the accumulator layout PF features at a time through the wave's
// LDS window (XOR-swizzled 16-byte slots, conflict-free both ways).  Landing areas double as prefetch buffers: the
// next tile's rows are in flight while the current tile computes.
template <int L>
struct Geo {
  static constexpr int PF = 32;                // features per pass through the LDS window (4 KB per wave)
  static constexpr int NPASS = L / PF;
  static constexpr int LPP = PF / 4;           // lanes per row piece
  static constexpr int RPI = 64 / LPP;         // rows per wave instruction
  static constexpr int NIT = kTile / RPI;
  static constexpr int STAGE_FLOATS = kTile * PF;
};
template <int L>
struct RowLanding {
  f32x4 v[Geo<L>::NPASS][Geo<L>::NIT];
};
// lane c (both half-waves) holds the global row index of tile row c
template <int L>
DEVINL void rows_issue(RowLanding<L>& ld, const float* __restrict__ base, int rowidx_by_c, int lane) {
  using G = Geo<L>;
  const int slot = lane % G::LPP;
#pragma unroll
  for (int it = 0; it < G::NIT; ++it) {
    const int r = it * G::RPI + lane / G::LPP;
    const int row = __shfl(rowidx_by_c, r);
    const f32x4* p = reinterpret_cast<const f32x4*>(base + static_cast<int64_t>(row) * L) + slot;
#pragma unroll
    for (int ps = 0; ps < G::NPASS; ++ps) ld.v[ps][it] = p[ps * G::LPP];
  }
}
template <int L, bool ADD>
DEVINL void rows_land(Tile<L>& t, const RowLanding<L>& ld, float* stage, int lane) {
  using G = Geo<L>;
  const int c = lane & 31, h = lane >> 5, slot = lane % G::LPP;
#pragma unroll
  for (int ps = 0; ps < G::NPASS; ++ps) {
    wave_sync();
#pragma unroll
    for (int it = 0; it < G::NIT; ++it) {
      const int r = it * G::RPI + lane / G::LPP;
      *reinterpret_cast<f32x4*>(stage + r * G::PF + ((slot ^ (r & (G::LPP - 1))) << 2)) = ld.v[ps][it];
    }
    wave_sync();
#pragma unroll
    for (int jj = 0; jj < G::PF / 8; ++jj) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(stage + c * G::PF + (((2 * jj + h) ^ (c & (G::LPP - 1))) << 2));
      const int j = ps * (G::PF / 8) + jj;
      tile_set_chunk<L>(t, j, ADD ? tile_get_chunk<L>(t, j) + v : v);
    }
  }
}

template <int L>
__global__ __launch_bounds__(kFastBlock) void edge_kernel_h(const EdgeArgs a, const int64_t ntiles) {
  using G = Geo<L>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MAT_BYTES = L * L * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;

  // all layer weights (hi/lo fp16, MFMA operand order) into LDS, once per workgroup
  for (int m = 0; m < a.nl; ++m) {
    const f32x4* src = reinterpret_cast<const f32x4*>(m == 0 ? a.w0e : a.layers[m].w);
    f32x4* dst = reinterpret_cast<f32x4*>(smem + m * MAT_BYTES);
    for (int i = tid; i < MAT_BYTES / 16; i += kFastBlock) dst[i] = src[i];
  }
  __syncthreads();
  float* stage = reinterpret_cast<float*>(smem + a.nl * MAT_BYTES) + wave * G::STAGE_FLOATS;
  const float wsinv0 = a.w0e[L * L], wscale0 = a.w0e[L * L + 1];  // pack-time power-of-two weight scale (matrix header)

  // XCD-aware persistent schedule: blocks b, b+8, ... share an XCD (round-robin dispatch); each XCD walks one
  // contiguous eighth of the tiles with its waves side by side, so gathered P_s rows are shared in that L2.
  const int64_t xcd = blockIdx.x & 7, local = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  const int64_t tiles_per_xcd = (ntiles + 7) >> 3;
  const int64_t t_begin = xcd * tiles_per_xcd;
  const int64_t t_end = t_begin + tiles_per_xcd < ntiles ? t_begin + tiles_per_xcd : ntiles;
  const int64_t stride = bpx * kFastWaves;
  int64_t tile = t_begin + local * kFastWaves + wave;

  // the indices of a wave's next two tiles are loaded ahead, so a tile starts with its row loads, not with a
  // dependent index fetch
  int s0 = 0, d0 = 0, pc0 = 0, s1 = 0, d1 = 0, pc1 = 0;
  auto load_idx = [&](int64_t t, int& s, int& d, int& pc) {
    int64_t p = t * kTile + c;
    if (p > a.E - 1) p = a.E - 1;  // rows beyond the end repeat the last edge and are never stored
    pc = static_cast<int>(p);
    s = a.src[p];
    d = a.dst[p];
  };
  if (tile < t_end) {
    load_idx(tile, s0, d0, pc0);
    if (tile + stride < t_end) load_idx(tile + stride, s1, d1, pc1);
  }
  for (; tile < t_end; tile += stride) {
    const int64_t p0 = tile * kTile;
    const int nvalid = (a.E - p0) < kTile ? static_cast<int>(a.E - p0) : kTile;  // >= 1: tile < ntiles
    RowLanding<L> la, lb;
    rows_issue<L>(la, a.e, pc0, lane);
    rows_issue<L>(lb, (a.dbg & 32) ? a.e : a.Ps, (a.dbg & 32) ? pc0 : s0, lane);
    int s2 = 0, d2 = 0, pc2 = 0;
    if (tile + 2 * stride < t_end) load_idx(tile + 2 * stride, s2, d2, pc2);
    const int firstd = __builtin_amdgcn_readlane(d0, 0);
    const int lastd = __builtin_amdgcn_readlane(d0, nvalid - 1);
    const int rp_first = a.rowptr[firstd], rp_last = a.rowptr[lastd + 1];

    // The accumulator starts from the gathered first-layer terms (resp. the bias) moved into the scaled domain
    // (x 2^k, exact), the MFMAs add the scaled products, one multiply by 2^-k brings the sum back.
    Tile<L> acc;
    float scale, inv;
    {
      Tile<L> ein;
      rows_land<L, false>(ein, la, stage, lane);
      if (!(a.dbg & 2)) rows_issue<L>(la, a.Pd, d0, lane);  // destination rows repeat inside a tile: L1 / L2 hits
      rows_land<L, false>(acc, lb, stage, lane);
      if (!(a.dbg & 2)) rows_land<L, true>(acc, la, stage, lane);
      tile_row_scale<L>(ein, scale, inv);
      tile_scale<L>(acc, scale * wscale0);
      if (!(a.dbg & 8)) tile_gemm_h<L>(acc, smem, ein, scale, lane);
    }
    // residual rows: read again (this wave's own loads of a few microseconds ago, L2 / MALL resident) instead of
    // being held in 64 registers across the remaining layers
    tile_scale<L>(acc, inv * wsinv0);
    tile_act_fast<L>(acc, a.act);
#pragma unroll 1
    for (int l = 1; l < a.nl; ++l) {
      Tile<L> nxt;
      tile_row_scale<L>(acc, scale, inv);
      tile_load_vec<L>(nxt, a.layers[l].b, h);
      tile_scale<L>(nxt, scale * a.layers[l].w[L * L + 1]);
      if (!(a.dbg & 8)) tile_gemm_h<L>(nxt, smem + l * MAT_BYTES, acc, scale, lane);
      tile_scale<L>(nxt, inv * a.layers[l].w[L * L]);
      tile_act_fast<L>(nxt, a.act);
      acc = nxt;
    }
    if (a.residual == AMBER_MODE_INNER && !(a.dbg & 1)) {
      RowLanding<L> lr;
      rows_issue<L>(lr, a.e, pc0, lane);
      rows_land<L, true>(acc, lr, stage, lane);
    }
    if (a.layer_norm == AMBER_MODE_INNER) tile_layernorm<L>(acc, a.gamma, a.beta, a.eps, h);

    // ---- epilogue, PF features a pass: per-destination sums in CSR order (lanes across features, all control
    // flow wave-uniform), then whole-row-piece stores ----
    const bool sb0 = p0 > static_cast<int64_t>(rp_first);
    const bool ends_last = (p0 + nvalid) == static_cast<int64_t>(rp_last);
#pragma unroll
    for (int ps = 0; ps < G::NPASS; ++ps) {
      wave_sync();  // the previous pass's readers are done
#pragma unroll
      for (int jj = 0; jj < G::PF / 8; ++jj)
        *reinterpret_cast<f32x4*>(stage + c * G::PF + (((2 * jj + h) ^ (c & (G::LPP - 1))) << 2)) =
            tile_get_chunk<L>(acc, ps * (G::PF / 8) + jj);
      wave_sync();
      if (lane < G::PF && !(a.dbg & 4)) {
        const int f = lane;
        float sum = 0.f;
        int cur = firstd;
        bool sb = sb0;
#pragma unroll
        for (int i = 0; i < kTile; ++i) {
          if (i < nvalid) {
            const int dc = __builtin_amdgcn_readlane(d0, i);
            if (dc != cur) {
              float* dstp = sb ? a.head + tile * L : a.agg + static_cast<int64_t>(cur) * L;
              dstp[ps * G::PF + f] = sum;
              cur = dc;
              sb = false;
              sum = 0.f;
            }
            sum += stage[i * G::PF + ((((f >> 2) ^ (i & (G::LPP - 1))) << 2) | (f & 3))];
          }
        }
        float* dstp = (!sb && ends_last) ? a.agg + static_cast<int64_t>(cur) * L : (sb ? a.head + tile * L : a.tail + tile * L);
        dstp[ps * G::PF + f] = sum;
      }
      if (a.write_e || a.out_edge) {
        const int slot = lane % G::LPP;
#pragma unroll
        for (int it = 0; it < G::NIT; ++it) {
          const int r = it * G::RPI + lane / G::LPP;
          const f32x4 v = *reinterpret_cast<const f32x4*>(stage + r * G::PF + ((slot ^ (r & (G::LPP - 1))) << 2));
          if (r < nvalid) {
            if (a.write_e && !(a.dbg & 16)) *reinterpret_cast<f32x4*>(a.e + (p0 + r) * L + ps * G::PF + 4 * slot) = v;
            if (a.out_edge)
              *reinterpret_cast<f32x4*>(a.out_edge + static_cast<int64_t>(a.perm[p0 + r]) * L + ps * G::PF + 4 * slot) = v;
          }
        }
      }
    }
    s0 = s1; d0 = d1; pc0 = pc1;
    s1 = s2; d1 = d2; pc1 = pc2;
  }
}

// ---------------------------------------------------------------------------------------------------
// weight packing: absmax -> power-of-two scale -> hi/lo fp16 in MFMA operand order (+ header)
// ---------------------------------------------------------------------------------------------------
__global__ void absmax_kernel(const float* __restrict__ W, int ld, int col0, int ncols, int L, unsigned* __restrict__ out_bits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L * ncols) return;
  const float v = fabsf(W[static_cast<int64_t>(i / ncols) * ld + col0 + (i % ncols)]);
  atomicMax(out_bits, __float_as_uint(v));  // non-negative floats order like their bit patterns; integer atomic
}

// out: [s][fb][hi|lo][lane][8 halves]; element jj of lane (h, r) is W[32 fb + r][16 s + 8 (jj >> 2) + 4 h + (jj & 3)];
// header (8 floats after the L*L*4 body bytes): [0] = inverse scale (read by the kernel), [1] = scale
template <int L>
__global__ void pack_h_kernel(const float* __restrict__ W, int ld, int col0, float* __restrict__ out,
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
  const float* src = W + static_cast<int64_t>(32 * fb + (lane & 31)) * ld + col0 + 16 * s + 4 * (lane >> 5);
  half8 hi, lo;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    const float ws = src[8 * (jj >> 2) + (jj & 3)] * scale;
    const _Float16 hh = static_cast<_Float16>(ws);
    hi[jj] = hh;
    lo[jj] = static_cast<_Float16>(ws - static_cast<float>(hh));
  }
  half8* o = reinterpret_cast<half8*>(out);
  o[((s * NFB + fb) * 2 + 0) * 64 + lane] = hi;
  o[((s * NFB + fb) * 2 + 1) * 64 + lane] = lo;
}

template <int L>
static hipError_t launch_pack_h_t(const float* W, int ld, int col0, int scale_col0, int scale_cols, float* out, hipStream_t s) {
  unsigned* bits = reinterpret_cast<unsigned*>(out + L * L + 4);  // scratch word inside the header
  hipError_t e = hipMemsetAsync(bits, 0, sizeof(unsigned), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(absmax_kernel, dim3((L * scale_cols + 255) / 256), dim3(256), 0, s, W, ld, scale_col0, scale_cols, L, bits);
  hipLaunchKernelGGL(pack_h_kernel<L>, dim3((L / 16 * (L / 32) * 64 + 255) / 256), dim3(256), 0, s, W, ld, col0, out, bits);
  return hipGetLastError();
}

hipError_t launch_pack_h(int L, const float* W, int ld, int col0, int scale_col0, int scale_cols, float* out, hipStream_t s) {
  switch (L) {
    case 32: return launch_pack_h_t<32>(W, ld, col0, scale_col0, scale_cols, out, s);
    case 64: return launch_pack_h_t<64>(W, ld, col0, scale_col0, scale_cols, out, s);
    case 128: return launch_pack_h_t<128>(W, ld, col0, scale_col0, scale_cols, out, s);
    default: return hipErrorInvalidValue;
  }
}

size_t edge_fast_lds_bytes(int L, int nl) {
  return static_cast<size_t>(nl) * L * L * 4 + static_cast<size_t>(kFastWaves) * kTile * 32 * 4;
}

bool edge_fast_supported(int L, int nl, int act, int residual, int layer_norm) {
  return (act == AMBER_ACT_RELU || act == AMBER_ACT_LEAKYRELU) && residual != AMBER_MODE_OUTER &&
         layer_norm != AMBER_MODE_OUTER && edge_fast_lds_bytes(L, nl) <= 160 * 1024;
}

template <int L>
static hipError_t launch_edge_h_t(const EdgeArgs& a, int num_cus, hipStream_t s) {
  const int64_t ntiles = (a.E + kTile - 1) / kTile;
  if (ntiles == 0) return hipSuccess;
  const size_t lds = edge_fast_lds_bytes(L, a.nl);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(edge_kernel_h<L>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(lds));
  if (e != hipSuccess) return e;
  int64_t grid = (num_cus / 8) * 8;  // one persistent workgroup per CU, a multiple of the 8 XCDs
  if (grid < 8) grid = 8;
  hipLaunchKernelGGL(edge_kernel_h<L>, dim3(static_cast<unsigned>(grid)), dim3(kFastBlock), lds, s, a, ntiles);
  return hipGetLastError();
}

hipError_t launch_edge_h(int L, const EdgeArgs& a, int num_cus, hipStream_t s) {
  switch (L) {
    case 32: return launch_edge_h_t<32>(a, num_cus, s);
    case 64: return launch_edge_h_t<64>(a, num_cus, s);
    case 128: return launch_edge_h_t<128>(a, num_cus, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace amber
