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
#include "mpn_post.h"
#include "mpn_tile.h"

// experiment switches for tools/exp_dbg.py (timing only, results are wrong with a switch on): build with
// make EXTRA=-DAMBER_DBG_SWITCHES; compiled out otherwise
#ifdef AMBER_DBG_SWITCHES
#define DBG(bit) (a.dbg & (bit))
#else
#define DBG(bit) 0
#endif

// In-kernel cycle stamps of the edge kernel's tile loop (diagnostic build only: make EXTRA=-DAMBER_STAMPS; the stamps go
// to a buffer of their own that nothing else reads, and launch_edge_f2 prints the per-section averages).
#ifdef AMBER_STAMPS
#define STAMP(k)                                                    \
  do {                                                              \
    __builtin_amdgcn_sched_barrier(0);                              \
    const unsigned long long _t = __builtin_amdgcn_s_memtime();     \
    st_acc[k] += static_cast<unsigned>(_t - st_prev);               \
    st_prev = _t;                                                   \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)
constexpr int kStamps = 12;
#else
#define STAMP(k)
#endif

#ifndef AMBER_EDGE_PD_EARLY
#define AMBER_EDGE_PD_EARLY 1
#endif
#ifndef AMBER_EDGE_PREFETCH_E
#define AMBER_EDGE_PREFETCH_E 1
#endif
#ifndef AMBER_KSTEP_GROUP_NODE
#define AMBER_KSTEP_GROUP_NODE 2
#endif
#ifndef AMBER_KSTEP_GROUP
#define AMBER_KSTEP_GROUP 2
#endif
#ifndef AMBER_NODE_RELOAD_V
#define AMBER_NODE_RELOAD_V 0
#endif

namespace amber {

typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));

constexpr int kF2Waves = 8;
#ifndef AMBER_EDGE_WAVES_64
#define AMBER_EDGE_WAVES_64 12
#endif
#ifndef AMBER_EDGE_WAVES_32
#define AMBER_EDGE_WAVES_32 16
#endif
// Waves per workgroup (= per CU) of the edge kernel.  L = 64: 12 (three per SIMD; the kernel fits 168 registers without
// the tile pipeline): -8 % per launch against 8 waves.  L = 32: 16 (four per SIMD): -22 %.  L = 128: 8 - registers (acc 64 + landing 64 + operands 48) and
// LDS (2 x 64 KB weights + 4 KB window a wave) allow no third wave.
#ifndef AMBER_NODE_WAVES_64
#define AMBER_NODE_WAVES_64 12
#endif
#ifndef AMBER_NODE_WAVES_32
#define AMBER_NODE_WAVES_32 16
#endif
template <int L>
constexpr int node_waves() {
  return L == 64 ? AMBER_NODE_WAVES_64 : (L == 32 ? AMBER_NODE_WAVES_32 : kF2Waves);
}
template <int L>
constexpr int edge_waves() {
  return L == 64 ? AMBER_EDGE_WAVES_64 : (L == 32 ? AMBER_EDGE_WAVES_32 : kF2Waves);
}
constexpr int kWinBytes = 4096;  // per-wave LDS window: hi plane [0, 2048), lo plane [2048, 4096)
constexpr int kLoPlane = 2048;

// max over the 64 lanes, result in every lane: DPP inside the rows of 16 lanes, then the row and half swaps of gfx950
// (v_permlane16_swap / v_permlane32_swap) - no LDS round trips (ds_bpermute) and no lane-address registers, which the
// compiler hoists out of the tile loop and spills
#define AMBER_DPP_F(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
DEVINL float wave_max(float m) {
  m = fmaxf(m, AMBER_DPP_F(m, 0xB1));   // quad_perm [1,0,3,2]
  m = fmaxf(m, AMBER_DPP_F(m, 0x4E));   // quad_perm [2,3,0,1]
  m = fmaxf(m, AMBER_DPP_F(m, 0x141));  // row_half_mirror
  m = fmaxf(m, AMBER_DPP_F(m, 0x140));  // row_mirror: every lane of a row of 16 holds the row's max
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);  // rows (0,0,2,2) | (1,1,3,3)
  m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);  // halves (lo,lo) | (hi,hi)
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
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
// x * scale -> hi + lo fp16, four values.  v_fma_mixlo/mixhi_f16 compute the f32 fma and round once to fp16 into one
// half of the destination: hi = fp16(x * scale), lo = fp16(x * scale - hi) (the difference is exact in f32), two
// instructions a value.  (Left to the compiler the same arithmetic took about 3.5.)
DEVINL void split4(const f32x4 x, float scale, half4& hi, half4& lo) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  u32x2 hh, ll;
  // (Round 3 checked these four statements as a suspect for the rare wrong rows described at launch_edge_f2_general: they
  // are not - a build with the split written in plain C showed the same errors, and the compiler does keep the wait state
  // between a half-register write and its reader here: it sees the operands of an asm statement.)
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    unsigned hv, lv;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hv) : "v"(x[2 * p]), "v"(scale));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hv) : "v"(x[2 * p + 1]), "v"(scale));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lv) : "v"(x[2 * p]), "v"(scale), "v"(hv));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lv) : "v"(x[2 * p + 1]), "v"(scale), "v"(hv));
    hh[p] = hv;
    ll[p] = lv;
  }
  hi = __builtin_bit_cast(half4, hh);
  lo = __builtin_bit_cast(half4, ll);
}
// single-term variant (AMBER_PREC_F16): x * scale -> fp16, no correction term
DEVINL void round4(const f32x4 x, float scale, half4& hi) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  u32x2 hh;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    unsigned hv;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hv) : "v"(x[2 * p]), "v"(scale));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hv) : "v"(x[2 * p + 1]), "v"(scale));
    hh[p] = hv;
  }
  hi = __builtin_bit_cast(half4, hh);
}
// Row indices for the direct row accesses.  Register i of lane (h, n) holds one tile row, the same row for all 32 lanes
// of a half, so its index (source / destination node, output row) is needed in every lane of the half.  The lane that
// holds it: lane i of each row of 16 lanes, when that lane loaded the index of row tile_xrow(lane) - then ONE DPP move
// (row_newbcast:i, gfx90a+) delivers it, instead of two v_readlane, two moves and a select.
// Edge tiles: register i of half h is CSR POSITION 16 h + i of the tile (edge_pos); the MFMA's row 8 (i >> 2) + (i & 3) +
// 4 h is only the slot the position occupies in the operand images (load_e places it there).  Each half of a wave then
// owns sixteen consecutive edges: the direct row accesses of a half are one address and immediate offsets, and the
// per-destination sums run over a lane's own registers with no exchange between the halves.
DEVINL int tile_xrow(int lane) { return 16 * (lane >> 5) + (lane & 15); }
template <int I>
DEVINL int row_bcast_c(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x150 + I, 0xf, 0xf, true);  // row_newbcast:I
}
DEVINL int row_bcast(int v, int i) {  // i: a constant after unrolling
  switch (i) {
    case 0: return row_bcast_c<0>(v);
    case 1: return row_bcast_c<1>(v);
    case 2: return row_bcast_c<2>(v);
    case 3: return row_bcast_c<3>(v);
    case 4: return row_bcast_c<4>(v);
    case 5: return row_bcast_c<5>(v);
    case 6: return row_bcast_c<6>(v);
    case 7: return row_bcast_c<7>(v);
    case 8: return row_bcast_c<8>(v);
    case 9: return row_bcast_c<9>(v);
    case 10: return row_bcast_c<10>(v);
    case 11: return row_bcast_c<11>(v);
    case 12: return row_bcast_c<12>(v);
    case 13: return row_bcast_c<13>(v);
    case 14: return row_bcast_c<14>(v);
    default: return row_bcast_c<15>(v);
  }
}
// sum over the 32 lanes of a half-wave, result in every lane of the half
DEVINL float half_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  // the other row of the half (lane ^ 16): v_permlane16_swap gives rows (0,0,2,2) and (1,1,3,3); the same two addends as
  // a lane ^ 16 exchange, without the LDS round trip of ds_swizzle
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// transposing LDS read (ds_read_b64_tr_b16): per group of 16 lanes a block of 4 rows x 16 columns of 16-bit elements,
// delivered column-major; lane 4q + p of the group supplies the address of row q, columns 4p .. 4p+3, and lane i
// receives column i (element q = row q).  EXEC must be all ones.
DEVINL half4 lds_read_tr(const unsigned char* p) {
  return __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p)));
}

// consecutive rows of L floats whose byte offsets fit the 12-bit immediate of a global access
template <int L>
constexpr int kRowGroup = L >= 128 ? 8 : 16;

// Which feature an accumulator column is.  The MFMA puts output column (fb, n) - feature block fb, lane n of a half - in
// register fb of lane n; WHICH feature that column computes is the packer's choice (the column order of the B operand).
// Column (fb, n) = feature NFB n + fb: a lane's NFB registers of one row are then NFB CONSECUTIVE floats of the row in
// memory, so every direct row access (gathered P_s / P_d rows, residual rows, stores of e' / x' / P_s / P_d, partial
// sums) is one 16-byte (L = 128) access a lane instead of four 4-byte ones: a quarter of the vector-memory instructions,
// whose issue - not the bytes - was what the row phases of the tile loop waited for.  A contraction whose INPUT comes
// from accumulator registers sees its k index permuted the same way (pack_f2: kperm).
template <int L>
DEVINL constexpr int fcol(int fb, int n) {
  return (L / 32) * n + fb;
}
// the NFB consecutive floats at p (16 / 8 / 4 bytes), as one access
template <int L>
DEVINL void piece_load(const float* __restrict__ p, float (&v)[L / 32]) {
  if constexpr (L == 128) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0], v[1] = t[1], v[2] = t[2], v[3] = t[3];
  } else if constexpr (L == 64) {
    const f32x2 t = *reinterpret_cast<const f32x2*>(p);
    v[0] = t[0], v[1] = t[1];
  } else {
    v[0] = p[0];
  }
}
template <int L>
DEVINL void piece_store(float* __restrict__ p, const float (&v)[L / 32]) {
  if constexpr (L == 128) {
    f32x4 t;
    t[0] = v[0], t[1] = v[1], t[2] = v[2], t[3] = v[3];
    *reinterpret_cast<f32x4*>(p) = t;
  } else if constexpr (L == 64) {
    f32x2 t;
    t[0] = v[0], t[1] = v[1];
    *reinterpret_cast<f32x2*>(p) = t;
  } else {
    p[0] = v[0];
  }
}
// register fb of row register i of a tile <- / -> piece
template <int L>
DEVINL void tile_set_piece(Tile<L>& t, int i, const float (&v)[L / 32]) {
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb) t.a[fb][i] = v[fb];
}
template <int L>
DEVINL void tile_get_piece(const Tile<L>& t, int i, float (&v)[L / 32]) {
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb) v[fb] = t.a[fb][i];
}

// streamed rows of the e tile as whole row pieces: v[pass][it] = e[row it*8 + lane/8][32 pass + 4 (lane % 8) .. +3]
template <int L>
struct ELanding {
  f32x4 v[L / 32][4];
};

// one 16-deep k-step of all feature blocks: acc[fb] += A(hi, lo) x W^T(hi, lo), weights in LDS as
// [k-step s][fb][hi | lo][lane] x 16 bytes
// SPLIT = false (AMBER_PREC_F16): plain fp16 operands, the hi terms only - one MFMA a product (tolerance 1e-2)
// GROUP feature blocks at a time: GROUP = 2 keeps half the B-operand registers in flight (16 instead of 32 at L = 128), which is
// what lets the edge kernel hold the rows of e through the hidden layer without spilling; the node kernel keeps 4
template <int L, bool SPLIT, int GROUP>
DEVINL void kstep(Tile<L>& acc, const half8 ah, const half8 al, const unsigned char* wl, int s, int lane) {
  constexpr int NFB = L / 32;
  constexpr int G = NFB < GROUP ? NFB : GROUP;
  const half8* w = reinterpret_cast<const half8*>(wl) + lane;
#pragma unroll
  for (int f0 = 0; f0 < NFB; f0 += G) {
    half8 bh[G], bl[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      bh[g] = w[((s * NFB + f0 + g) * 2 + 0) * 64];
      if constexpr (SPLIT) bl[g] = w[((s * NFB + f0 + g) * 2 + 1) * 64];
    }
#pragma unroll
    for (int g = 0; g < G; ++g) acc.a[f0 + g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[g], acc.a[f0 + g], 0, 0, 0);
    if constexpr (SPLIT) {
#pragma unroll
      for (int g = 0; g < G; ++g) acc.a[f0 + g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[g], acc.a[f0 + g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < G; ++g) acc.a[f0 + g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[g], acc.a[f0 + g], 0, 0, 0);
    }
  }
}

// lane-constant byte offsets into the wave's LDS window.  Four registers; the other ten offsets the two images need
// differ from these by XOR constants and multiples of 1024 (the swizzles are XORs and the row strides powers of two).
// The node kernel, which is out of registers, forms them at the point of use (PIN: node launch -3 %, it reloaded ten
// hoisted offsets from scratch in every tile); the edge kernel has room and lets the compiler hoist them (PIN there:
// +0.6 %).
struct WinOff {
  int ew0;  // row-piece image ([row][32 features] fp16, 64-byte rows, 16-byte slots XOR-swizzled by (row >> 2) & 3):
            // writer, row it*8 + lane/8, 8-byte piece lane % 8; this is it = 0
  int er0;  // ... reader (A operand of k-step sp): row n, slot 2 sp + h; this is sp = 0
  int hw0;  // transposed image ([feature][32 rows] fp16, 64-byte rows, 8-byte units XOR-swizzled by (row >> 1) & 7):
            // writer, image row n, unit 2 g + h; this is g = 0
  int hr0;  // ... transposing reader: k-step sp, chunk t; this lane supplies image row 16 sp + 8 h + 4 t + q,
            // unit 4 colblk + p; this is sp = 0, t = 0
  // it -> it + 1: row + 8 = 512 bytes further and the swizzle's bit 1 flips (slot ^ 2 = 32 bytes); it -> it + 2: + 1024
  DEVINL int ew(int it) const { return (ew0 ^ ((it & 1) ? 544 : 0)) + 1024 * (it >> 1); }
  DEVINL int er(int sp) const { return er0 ^ (sp << 5); }  // slot ^ 2
  DEVINL int hw(int g) const { return hw0 ^ (g << 4); }    // unit ^ 2 g
  // t -> 1: row + 4 = 256 bytes further and the swizzle's bit 1 flips (unit ^ 2 = 16 bytes); sp -> 1: row + 16 = + 1024
  DEVINL int hr(int sp, int t) const { return (hr0 ^ (t ? 272 : 0)) + 1024 * sp; }
  // keeps the derived offsets from being computed once and for all before the tile loop
  DEVINL void pin() { asm volatile("" : "+v"(ew0), "+v"(er0), "+v"(hw0), "+v"(hr0)); }
};
DEVINL WinOff make_win_off(int lane) {
  WinOff o;
  const int n = lane & 31, h = lane >> 5;
  {
    const int rr = lane >> 3, q = lane & 7;
    o.ew0 = rr * 64 + ((((q >> 1) ^ ((rr >> 2) & 3))) << 4) + ((q & 1) << 3);
  }
  o.er0 = n * 64 + (((h ^ ((n >> 2) & 3))) << 4);
  o.hw0 = n * 64 + (((h ^ ((n >> 1) & 7))) << 3);
  const int gl = lane & 15, q = gl >> 2, p = gl & 3, colblk = (lane >> 4) & 1;
  const int row = 8 * h + q;
  o.hr0 = row * 64 + ((((4 * colblk + p) ^ ((row >> 1) & 7))) << 3);
  return o;
}

// acc += (tile * scale) x W^T for a tile held as whole row pieces (ELanding): 32 features a pass are split into hi/lo
// fp16, written to the window as a [row][feature] image and read back as the A operand with ds_read_b128
struct NoPassHook {
  DEVINL void operator()(int) const {}
};
// `after(ps)` runs when pass ps has left its source registers (la.v[ps] / in.a[ps]): the place to issue loads that may
// land in them, under the pass's MFMAs
template <int L, bool SPLIT, bool PIN = false, int GROUP = AMBER_KSTEP_GROUP_NODE, typename F = NoPassHook>
DEVINL void gemm_rows(Tile<L>& acc, const ELanding<L>& la, float scale, unsigned char* win, const WinOff& o_in,
                      const unsigned char* wl, int lane, F&& after = NoPassHook()) {
  WinOff o = o_in;
  if constexpr (PIN) o.pin();
#pragma unroll
  for (int ps = 0; ps < L / 32; ++ps) {
    wave_sync();  // the previous readers of the window are done
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      half4 hi, lo;
      if constexpr (SPLIT) {
        split4(la.v[ps][it], scale, hi, lo);
        *reinterpret_cast<half4*>(win + kLoPlane + o.ew(it)) = lo;
      } else {
        round4(la.v[ps][it], scale, hi);
      }
      *reinterpret_cast<half4*>(win + o.ew(it)) = hi;
    }
    after(ps);
    wave_sync();
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      const half8 ah = *reinterpret_cast<const half8*>(win + o.er(sp));
      half8 al = ah;
      if constexpr (SPLIT) al = *reinterpret_cast<const half8*>(win + kLoPlane + o.er(sp));
      kstep<L, SPLIT, GROUP>(acc, ah, al, wl, 2 * ps + sp, lane);
    }
  }
}

// acc += (tile * scale) x W^T for a tile in accumulator layout (feature per lane): 32 features a pass go through the
// window as a [feature][row] image and come back as the A operand through the transposing read
template <int L, bool SPLIT, bool PIN = false, int GROUP = AMBER_KSTEP_GROUP_NODE, typename F = NoPassHook>
DEVINL void gemm_acc(Tile<L>& acc, const Tile<L>& in, float scale, unsigned char* win, const WinOff& o_in,
                     const unsigned char* wl, int lane, F&& after = NoPassHook()) {
  WinOff o = o_in;
  if constexpr (PIN) o.pin();
#pragma unroll
  for (int fb = 0; fb < L / 32; ++fb) {
    wave_sync();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 x;
      x[0] = in.a[fb][4 * g + 0];
      x[1] = in.a[fb][4 * g + 1];
      x[2] = in.a[fb][4 * g + 2];
      x[3] = in.a[fb][4 * g + 3];
      half4 hi, lo;
      if constexpr (SPLIT) {
        split4(x, scale, hi, lo);
        *reinterpret_cast<half4*>(win + kLoPlane + o.hw(g)) = lo;
      } else {
        round4(x, scale, hi);
      }
      *reinterpret_cast<half4*>(win + o.hw(g)) = hi;
    }
    after(fb);
    wave_sync();
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      const half4 a0 = lds_read_tr(win + o.hr(sp, 0)), a1 = lds_read_tr(win + o.hr(sp, 1));
      half8 ah, al;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        ah[t] = a0[t];
        ah[4 + t] = a1[t];
      }
      al = ah;
      if constexpr (SPLIT) {
        const half4 b0 = lds_read_tr(win + kLoPlane + o.hr(sp, 0)), b1 = lds_read_tr(win + kLoPlane + o.hr(sp, 1));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          al[t] = b0[t];
          al[4 + t] = b1[t];
        }
      }
      kstep<L, SPLIT, GROUP>(acc, ah, al, wl, 2 * fb + sp, lane);
    }
  }
}

template <int L>
DEVINL float landing_absmax(const ELanding<L>& la) {
  float m = 0.f;
#pragma unroll
  for (int ps = 0; ps < L / 32; ++ps)
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int t = 0; t < 4; ++t) m = fmaxf(m, fabsf(la.v[ps][it][t]));
  return m;
}
// half_sum of sixteen values at once, step by step over all of them: a DPP operand needs two wait states after the
// instruction that wrote it, and one value's five steps are a dependent chain - taken one value at a time the compiler
// filled the hazards with s_nop (173 of them in one LayerNorm)
DEVINL void half_sum16(float (&v)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] += AMBER_DPP_F(v[i], 0xB1);   // quad_perm [1,0,3,2]
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] += AMBER_DPP_F(v[i], 0x4E);   // quad_perm [2,3,0,1]
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] += AMBER_DPP_F(v[i], 0x141);  // row_half_mirror
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] += AMBER_DPP_F(v[i], 0x140);  // row_mirror
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i]), false, false);
    v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
}
// The same sums as a reduce-scatter (round 3): every step halves the number of registers instead of repeating all sixteen -
// lanes whose bit k is 0 keep the even register of a pair and hand the odd one to their partner l ^ 2^k, the others the other
// way round - so that after the four steps inside a row of 16 lanes and the exchange between the two rows of a half, lane l
// holds the half's total of value l & 15: 8 + 4 + 2 + 1 additions (each with two selects and the partner exchange) instead of
// 5 x 16, and whatever follows per value (the mean's scale, the inverse deviation: an FMA and a quarter-rate v_rsq_f32) is done
// ONCE, in that lane, before row_bcast hands every lane the sixteen results.  The partners and the order of the additions are
// those of half_sum16 (a + b = b + a): the totals are bit-identical.
template <int CTRL, int BANKS>
DEVINL float dpp_into(float old, float src) {  // lanes of the banks in BANKS take src through CTRL, the others keep old
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, 0xf, BANKS, false));
}
#ifndef AMBER_LN_BANKED
#define AMBER_LN_BANKED 1
#endif
#if AMBER_LN_BANKED
// out = a + a[partner] in the lanes of the banks BA, b + b[partner] in the lanes of the banks BB (a bank = 4 lanes of a row of
// 16); the partner through the DPP control.  Two v_add_f32_dpp under bank masks: no select.  (s_nop 1: a DPP operand needs two
// wait states after the instruction that wrote it, and the compiler does not see the DPP inside the asm.)
#define AMBER_BANKED_ADD(out, a, b, CA, BA, CB, BB)                                                                       \
  asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 " CA " row_mask:0xf bank_mask:" BA "\n\tv_add_f32_dpp %0, %2, %2 " CB       \
      " row_mask:0xf bank_mask:" BB                                                                                       \
      : "=&v"(out)                                                                                                        \
      : "v"(a), "v"(b))
DEVINL float half_reduce_scatter16(const float (&v)[16]) {
  const int l = threadIdx.x;  // bits 0..3: the lane's place in its row of 16
  const bool b0 = (l & 1) != 0, b1 = (l & 2) != 0;
  float w[8], x[4], y[2];
  // the two steps whose partners differ in a whole bank come first (8 and 4 results, two instructions each)
#pragma unroll
  for (int k = 0; k < 8; ++k) AMBER_BANKED_ADD(w[k], v[2 * k], v[2 * k + 1], "row_ror:8", "0x3", "row_ror:8", "0xc");  // l ^ 8
#pragma unroll
  for (int k = 0; k < 4; ++k) AMBER_BANKED_ADD(x[k], w[2 * k], w[2 * k + 1], "row_shl:4", "0x5", "row_shr:4", "0xa");  // l ^ 4
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float send = b1 ? x[2 * k] : x[2 * k + 1], keep = b1 ? x[2 * k + 1] : x[2 * k];
    y[k] = keep + AMBER_DPP_F(send, 0x4E);  // quad_perm [2,3,0,1]: l ^ 2
  }
  const float send = b0 ? y[0] : y[1], keep = b0 ? y[1] : y[0];
  const float z = keep + AMBER_DPP_F(send, 0xB1);  // quad_perm [1,0,3,2]: l ^ 1
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(z), __float_as_uint(z), false, false);  // the half's other row
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// which lane of a row ends with value i in that order: bit 3 of the lane chose between v[2k] / v[2k + 1] (bit 0 of i), bit 2
// between w[2k] / w[2k + 1] (bit 1 of i), bit 1 bit 2 of i, bit 0 bit 3 of i
constexpr int ln_lane_of(int i) { return ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3); }
#else
DEVINL float half_reduce_scatter16(const float (&v)[16]) {
  const int l = threadIdx.x;  // bits 0..3: the lane's place in its row of 16
  const bool b0 = (l & 1) != 0, b1 = (l & 2) != 0, b2 = (l & 4) != 0, b3 = (l & 8) != 0;
  float w[8], x[4], y[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float send = b0 ? v[2 * k] : v[2 * k + 1], keep = b0 ? v[2 * k + 1] : v[2 * k];
    w[k] = keep + AMBER_DPP_F(send, 0xB1);  // quad_perm [1,0,3,2]: l ^ 1
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float send = b1 ? w[2 * k] : w[2 * k + 1], keep = b1 ? w[2 * k + 1] : w[2 * k];
    x[k] = keep + AMBER_DPP_F(send, 0x4E);  // quad_perm [2,3,0,1]: l ^ 2
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float send = b2 ? x[2 * k] : x[2 * k + 1], keep = b2 ? x[2 * k + 1] : x[2 * k];
    float t = dpp_into<0x104, 0x5>(send, send);  // row_shl:4 into the lanes with bit 2 clear (banks 0, 2): from l + 4
    t = dpp_into<0x114, 0xA>(t, send);           // row_shr:4 into the lanes with bit 2 set (banks 1, 3): from l - 4
    y[k] = keep + t;
  }
  const float send = b3 ? y[0] : y[1], keep = b3 ? y[1] : y[0];
  const float z = keep + AMBER_DPP_F(send, 0x128);  // row_ror:8: l ^ 8
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(z), __float_as_uint(z), false, false);  // the half's other row
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
constexpr int ln_lane_of(int i) { return i; }
#endif
#ifndef AMBER_LN_SCATTER
#define AMBER_LN_SCATTER 1
#endif
// LayerNorm over the features of every row of a tile in accumulator layout: across the lanes of a half, one register a row
template <int L>
DEVINL void tile_layernorm_acc_pre(Tile<L>& acc, const float (&g)[L / 32], const float (&bt)[L / 32], float eps);
template <int L>
DEVINL void tile_layernorm_acc(Tile<L>& acc, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int n) {
  constexpr int NFB = L / 32;
  float g[NFB], bt[NFB];
  piece_load<L>(gamma + NFB * n, g);
  piece_load<L>(beta + NFB * n, bt);
  tile_layernorm_acc_pre<L>(acc, g, bt, eps);
}
// ... with this lane's gamma / beta already in registers (fetched at the tile's top where registers allow: on a one-mesh
// graph a launch is one tile's dependent chain, and every small load inside it is an exposed round trip to L2)
template <int L>
DEVINL void tile_layernorm_acc_pre(Tile<L>& acc, const float (&g)[L / 32], const float (&bt)[L / 32], float eps) {
  constexpr int NFB = L / 32;
  float s[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    s[i] = acc.a[0][i];
#pragma unroll
    for (int fb = 1; fb < NFB; ++fb) s[i] += acc.a[fb][i];
  }
#if AMBER_LN_SCATTER
  const float mean_l = half_reduce_scatter16(s) * (1.0f / L);  // lane l: the mean of row l & 15
#else
  half_sum16(s);
#endif
#pragma unroll
  for (int i = 0; i < 16; ++i) {
#if AMBER_LN_SCATTER
    const float mean = __int_as_float(row_bcast(__float_as_int(mean_l), ln_lane_of(i)));
#else
    const float mean = s[i] * (1.0f / L);
#endif
    float q = 0.f;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
      const float dlt = acc.a[fb][i] - mean;
      acc.a[fb][i] = dlt;
      q += dlt * dlt;
    }
    s[i] = q;
  }
#if AMBER_LN_SCATTER
  const float rstd_l = __builtin_amdgcn_rsqf(half_reduce_scatter16(s) * (1.0f / L) + eps);  // v_rsq_f32: 1 ulp; lane l: row l & 15
#else
  half_sum16(s);
#endif
#pragma unroll
  for (int i = 0; i < 16; ++i) {
#if AMBER_LN_SCATTER
    const float rstd = __int_as_float(row_bcast(__float_as_int(rstd_l), ln_lane_of(i)));
#else
    const float rstd = __builtin_amdgcn_rsqf(s[i] * (1.0f / L) + eps);  // v_rsq_f32: 1 ulp
#endif
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
      // (scalar here - so that the result lands in the store's register quad without a move - was tried in round 3: the edge
      // kernel at L = 128 then spills 66 registers)
      acc.a[fb][i] = acc.a[fb][i] * rstd * g[fb] + bt[fb];
    }
  }
}

#ifdef AMBER_STAMPS
__device__ unsigned long long g_stamps[kStamps + 1];
#endif
// EMB: first step with the linear edge embedding fused in (e is computed on the fly instead of read)
// VAR = 0: AMBER's configuration space of the stack (relu / leakyrelu, residual and LayerNorm none or inner) - the kernel of
// the headline, whose code generation nothing else may disturb.  VAR = 1 (round 3): the general instantiation - tanh as well,
// and the "outer" residual / LayerNorm (message_passing_block.py:62-74, 127-128), where the per-destination sums are taken
// from the value BEFORE the outer operations (what the node module aggregates) and the stored e' is the value after them;
// it reads its rows of e at the tile's top instead of a tile ahead (with an outer residual they are needed to the end).
// VAR = 2: the default structure with tanh as the activation (no outer mode).
template <int L, bool EMB, bool SPLIT, int VAR = 0>
__global__ __launch_bounds__(64 * edge_waves<L>()) void edge_kernel_f2(const EdgeArgs a, const int64_t ntiles) {
  constexpr int NFB = L / 32;
  constexpr bool GEN = (VAR & 1) != 0;  // general epilogue (outer modes); VAR = 2: the default structure with tanh
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MAT_BYTES = L * L * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave id as a scalar
  const int n = lane & 31, h = lane >> 5;

  unsigned char* win = smem + a.nl * MAT_BYTES + wave * kWinBytes;

  const WinOff wo = make_win_off(lane);
  const float wsinv0 = a.w0e[L * L], wscale0 = a.w0e[L * L + 1];  // pack-time power-of-two weight scale (matrix header)

  // XCD-aware persistent schedule: blocks b, b+8, ... share an XCD (round-robin dispatch); each XCD walks one
  // contiguous eighth of the tiles with its waves side by side, so gathered P_s rows are shared in that L2.
  // (tile numbers fit 32 bits: E < 2^31; 32-bit scalars keep the loop's bookkeeping out of the spill lanes)
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  const int tiles_per_xcd = static_cast<int>((ntiles + 7) >> 3);
  const int t_begin = xcd * tiles_per_xcd;
  const int t_end = t_begin + tiles_per_xcd < static_cast<int>(ntiles) ? t_begin + tiles_per_xcd : static_cast<int>(ntiles);
  const int stride = bpx * edge_waves<L>();
  // Round r of this wave.  Full rounds (every wave of the XCD has a tile) keep the contiguous pattern - the waves of a
  // workgroup side by side.  The LAST, partial round is dealt across the workgroups (tile j of it to workgroup j % bpx,
  // wave j / bpx) instead of filling the first workgroups' waves: on a one-mesh graph (1.5 tiles a wave) the second
  // round then loads every CU with half its waves, not half the CUs with all of theirs.
  const int full_rounds = t_end > t_begin ? (t_end - t_begin) / stride : 0;
#ifndef AMBER_EDGE_NO_DEAL
  const int tile_last = t_begin + full_rounds * stride + local + bpx * wave;  // may lie beyond t_end: then the wave has none
#else
  const int tile_last = t_begin + full_rounds * stride + local * edge_waves<L>() + wave;
#endif
  const int tile_first = t_begin + local * edge_waves<L>() + wave;
  const int my_tiles = full_rounds + (tile_last < t_end ? 1 : 0);
  auto tile_of = [&](int r) -> int { return r < full_rounds ? tile_first + r * stride : tile_last; };
  int rnd = 0;
  int tile = tile_of(0);

  // d: lane c (both halves) holds the destination of tile row c (row boundaries of the per-destination sums); sx / dx:
  // source / destination of row tile_xrow(lane), the form the row broadcasts read (see row_bcast).  The next tile's are
  // loaded a tile ahead.
  const int xr = tile_xrow(lane);
  int sx0 = 0, d0 = 0, dx0 = 0, sx1 = 0, d1 = 0, dx1 = 0;
  auto load_idx = [&](int64_t t, int& sx, int& d, int& dx) {
    const int64_t q = t * kTile, last = a.E - 1;
    const int64_t p = q + n < last ? q + n : last;  // rows beyond the end repeat the last edge and are never stored
    const int64_t px = q + xr < last ? q + xr : last;
    d = a.dst[p];
    sx = a.src[px];
    dx = a.dst[px];
  };
  if (my_tiles > 0) load_idx(tile, sx0, d0, dx0);
  // Software pipeline over the tiles of a wave (PIPE): the rows of e that open a tile are loaded during the PREVIOUS
  // tile's epilogue (LayerNorm, stores, per-destination sums), when the landing registers of the hidden layer are free,
  // so their latency runs under that work.  Measured at L = 64: -2.4 % per launch (L = 32: +30 %, off).  At L = 128 the 64 loop-carried
  // registers do not fit beside the accumulators (the kernel spills, and every scratch reload drains the in-order
  // memory counter, prefetch included: +50 %), so the wide kernel keeps the loads at the top of the tile.
  constexpr bool PIPE = false;
  // The rows of e arrive ONCE, in accumulator layout (register i of half h = position 16 h + i, a lane's NFB features of it
  // as one piece): they feed the first contraction through the [feature][edge] window image like a hidden activation
  // (W0e is packed with the permuted k order for that) and stay in their registers for the residual e + MLP(...), which
  // round 1 / 2 re-read feature per lane - by then the rows had left the L2: 5 GB of a launch's 18.7 GB of traffic.
  Tile<L> et;      // e of the tile at hand
  Tile<L> pn;      // P_s[src] of the tile at hand, accumulator layout (becomes the accumulator)
  auto gather_ps = [&](int sidx, int nq) {
    const float* const psn = a.Ps + NFB * nq;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const unsigned si = static_cast<unsigned>(row_bcast(sidx, i));
      float pc[NFB];
      piece_load<L>(psn + static_cast<size_t>(DBG(32) ? 0u : si) * L, pc);
      tile_set_piece<L>(pn, i, pc);
    }
  };
  // rows of e of tile t into `et` (accumulator layout).  With AMBER_EDGE_PREFETCH_E the NEXT tile's rows are fetched as soon
  // as the residual has taken the current ones - et's registers are free from there to the tile's end - ahead of this
  // tile's stores, so that the tile's top issues half as many loads and never waits for them behind the stores.
  auto load_et = [&](int64_t t, int hq_, int nq_) {
    const int64_t q0 = t * kTile;
    const int nv = (a.E - q0) < kTile ? static_cast<int>(a.E - q0) : kTile;
    const float* const rows = a.e + q0 * L;
    // (two loops under a real, wave-uniform branch: with the test inside one loop the compiler computed BOTH address forms for
    // every row and selected between them - 57 vector instructions to issue 16 loads; the empty asm keeps the arms apart)
    if (nv == kTile) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {  // one address per group of rows, rows inside it are immediate offsets
        float pc[NFB];
        piece_load<L>(rows + static_cast<unsigned>(hq_) + kRowGroup<L> * (i / kRowGroup<L>) * L + (i % kRowGroup<L>) * L, pc);
        tile_set_piece<L>(et, i, pc);
      }
    } else {
      asm volatile("; partial tile");
#pragma unroll
      for (int i = 0; i < 16; ++i) {  // rows beyond the end repeat the last edge and are never stored
        float pc[NFB];
        piece_load<L>(rows + static_cast<unsigned>(min(i + 16 * h, nv - 1) * L + NFB * nq_), pc);
        tile_set_piece<L>(et, i, pc);
      }
    }
  };
  constexpr bool PREFETCH_E = !EMB && AMBER_EDGE_PREFETCH_E != 0 && !GEN;
  if constexpr (PREFETCH_E) {
    if (my_tiles > 0) load_et(tile, 16 * h * L + NFB * n, n);
  }
  // (the first tile's indices and rows of e are in flight: the weight prologue below runs under their latency - on a
  // one-mesh graph a launch is only one or two tiles a wave, so the prologue is a visible share of it)
  // all layer weights (hi/lo fp16, B-operand order) into LDS, once per workgroup
  for (int m = EMB ? 1 : 0; m < a.nl; ++m) {
    const f32x4* src = reinterpret_cast<const f32x4*>(m == 0 ? a.w0e : a.layers[m].w);
    f32x4* dst = reinterpret_cast<f32x4*>(smem + m * MAT_BYTES);
    for (int i = tid; i < MAT_BYTES / 16; i += 64 * edge_waves<L>()) dst[i] = src[i];
  }
  if constexpr (EMB) {
    // The fused first step needs no first-layer contraction: e = emb_w attr + emb_b is linear in the raw features, so
    // W0e e = (W0e emb_w) attr + W0e emb_b.  The folded table [kMaxFusedEmbed + 1][L] (last row: the constant) takes the
    // place of the first matrix in LDS; every workgroup forms it once (L (F_e + 1) dot products of length L, fp32).
    float* fold = reinterpret_cast<float*>(smem);
    for (int o = tid; o < (kMaxFusedEmbed + 1) * L; o += 64 * edge_waves<L>()) {
      const int k = o / L, j = o - k * L;
      float sum = 0.f;
      if (k < a.emb_f || k == kMaxFusedEmbed) {
        const float* wr = a.emb_w0 + static_cast<int64_t>(j) * (3 * L) + 2 * L;
        for (int m = 0; m < L; ++m) sum = fmaf(wr[m], k == kMaxFusedEmbed ? a.emb_b[m] : a.emb_w[m * a.emb_f + k], sum);
      }
      fold[o] = sum;
    }
  }
  __syncthreads();
#ifdef AMBER_STAMPS
  unsigned st_acc[kStamps] = {};
  unsigned st_tiles = 0;
  unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#endif
  for (; rnd < my_tiles; tile = tile_of(++rnd)) {
    STAMP(0);
#ifdef AMBER_STAMPS
    ++st_tiles;
#endif
    const bool has_next = rnd + 1 < my_tiles;
    const int tnext = tile_of(rnd + 1);
    const int64_t p0 = static_cast<int64_t>(tile) * kTile;
    const int nvalid = (a.E - p0) < kTile ? static_cast<int>(a.E - p0) : kTile;  // >= 1: tile < ntiles
    const float* const et_rows = a.e + p0 * L;  // this tile's rows of e (wave-uniform base, 32-bit lane offsets)
    float* const eo = (a.e_out ? a.e_out : a.e) + p0 * L;  // where the updated rows go (training tape: not in place)
    const int rlim = nvalid - 1;
    // Lane offsets of the direct row accesses are recomputed per tile from these opaque copies: hoisted out of the tile
    // loop they would occupy dozens of registers for its whole length.
    int hq = 16 * h * L + NFB * n, nq = n;  // first row of this lane's half at its NFB features; its lane of the half
    asm volatile("" : "+v"(hq), "+v"(nq));
    // ---- first step with the edge embedding fused in: the rows of e are computed, not read (K1: Embedding.forward,
    // src/mpn/common/embedding.py:44-54, linear form) - directly in accumulator layout ----
    float eav[kMaxFusedEmbed] = {0.f, 0.f, 0.f, 0.f};  // raw features of tile row tile_xrow(lane) (see row_bcast)
    if (EMB && a.emb_f > 0) {
      int64_t p = p0 + xr;
      if (p > a.E - 1) p = a.E - 1;
      const float* pa = a.emb_attr + static_cast<int64_t>(a.perm[p]) * a.emb_f;
#pragma unroll
      for (int k = 0; k < kMaxFusedEmbed; ++k)
        if (k < a.emb_f) eav[k] = pa[k];
    }
    auto embed_tile = [&](Tile<L>& t) {
      float wq[NFB][kMaxFusedEmbed], bq[NFB];
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) {
        bq[fb] = a.emb_b[fcol<L>(fb, nq)];
#pragma unroll
        for (int k = 0; k < kMaxFusedEmbed; ++k) wq[fb][k] = k < a.emb_f ? a.emb_w[fcol<L>(fb, nq) * a.emb_f + k] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float ev[kMaxFusedEmbed];
#pragma unroll
        for (int k = 0; k < kMaxFusedEmbed; ++k) ev[k] = __int_as_float(row_bcast(__float_as_int(eav[k]), i));
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
          float v = bq[fb];
#pragma unroll
          for (int k = 0; k < kMaxFusedEmbed; ++k) v = fmaf(ev[k], wq[fb][k], v);  // k ascending, as the embed kernel
          t.a[fb][i] = v;
        }
      }
    };
    // t += (W0e emb_w) attr + W0e emb_b from the folded table (see the prologue)
    auto fold_tile_add = [&](Tile<L>& t) {
      const float* fold = reinterpret_cast<const float*>(smem);
      float wq[NFB][kMaxFusedEmbed], bq[NFB];
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) {
        bq[fb] = fold[kMaxFusedEmbed * L + fcol<L>(fb, nq)];
#pragma unroll
        for (int k = 0; k < kMaxFusedEmbed; ++k) wq[fb][k] = fold[k * L + fcol<L>(fb, nq)];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float ev[kMaxFusedEmbed];
#pragma unroll
        for (int k = 0; k < kMaxFusedEmbed; ++k) ev[k] = __int_as_float(row_bcast(__float_as_int(eav[k]), i));
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
          float v = bq[fb];
#pragma unroll
          for (int k = 0; k < kMaxFusedEmbed; ++k) v = fmaf(ev[k], wq[fb][k], v);
          t.a[fb][i] += v;
        }
      }
    };
    // ---- streamed e rows (whole row pieces; PIPE: issued during the previous tile's epilogue) and gathered
    // first-layer terms, straight into the accumulator layout ----
    // (the output rows of the last step's edge latents: loaded HERE, ahead of the tile's other loads, so that the waits
    // for those cover it; a load left pending at the stores makes the compiler drain the memory counter after them)
    int pm = 0;
    if (a.out_edge) {
      int64_t p = p0 + xr;
      if (p > a.E - 1) p = a.E - 1;
      pm = a.perm[p];
    }
    if constexpr (!EMB && !PREFETCH_E) load_et(tile, hq, nq);
    gather_ps(sx0, nq);
    // L <= 64 (registers allow it): the tile-invariant vectors - the last hidden layer's bias, LayerNorm gamma / beta - are
    // fetched with the tile's first loads instead of where they are used (each an exposed round trip to L2 on a one-mesh graph)
#ifndef AMBER_EDGE_EARLY_MAXL
#define AMBER_EDGE_EARLY_MAXL 64
#endif
    constexpr bool EARLY = L <= AMBER_EDGE_EARLY_MAXL;
    float e_b1[NFB], e_g[NFB], e_bt[NFB];
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) e_b1[fb] = e_g[fb] = e_bt[fb] = 0.f;
    if constexpr (EARLY) {
      if (a.nl > 1) piece_load<L>(a.layers[a.nl - 1].b + NFB * nq, e_b1);
      if (a.layer_norm != AMBER_MODE_NONE) {
        piece_load<L>(a.gamma + NFB * nq, e_g);
        piece_load<L>(a.beta + NFB * nq, e_bt);
      }
    }
    Tile<L> acc = pn;
    // The per-destination sums run per HALF over its own sixteen positions (reduction tiles of 16 edges, red_shift 4):
    // first / last destination of this half's positions and where their rows begin / end
    const int nvh = nvalid - 16 * h < 0 ? 0 : (nvalid - 16 * h > 16 ? 16 : nvalid - 16 * h);  // positions of this half that exist
    const int firstd = row_bcast(dx0, 0);
    int lastd;
    if (nvalid == kTile)
      lastd = row_bcast(dx0, 15);
    else
      lastd = __shfl(d0, nvh > 0 ? 16 * h + nvh - 1 : 0);
    const int rp_first = a.rowptr[firstd], rp_last = a.rowptr[lastd + 1];
    // bit r: position r of the tile begins a new destination (wave-uniform); positions beyond the end never do
    const int dprev = __shfl_up(d0, 1);  // every lane takes part: the shuffle must not sit behind the n > 0 test
    unsigned newseg = static_cast<unsigned>(__ballot((d0 != dprev) & (n > 0)));
    if (nvalid < kTile) newseg &= (1u << nvalid) - 1u;
    STAMP(1);  // loads issued, bookkeeping

    // ---- layer 0: acc = (P_s + P_d) + W0e e, in the scaled domain ----
    float scale = 1.f, inv = 1.f, us = 1.f;
    if constexpr (EMB) {
      fold_tile_add(acc);  // exact fp32, no scaled domain
    } else {
      pow2_scale(wave_max(tile_absmax<L>(et)), scale, inv);
      STAMP(2);  // e rows arrived, tile maximum
      tile_scale_rows<L>(acc, scale * wscale0);  // (the gathered rows sit in row-piece order: see tile_scale_rows)
      STAMP(3);  // P_s rows arrived
    }
    // destination terms P_d[dst] (rows repeat inside a tile: L1 / L2 hits) join after the contraction, when the landing
    // registers of e are free again.  (AMBER_PREFETCH_PD fetches a quarter of them after every pass of the contraction
    // into the landing registers that pass has left: no spill, and 1 % slower - the other wave of the SIMD already covers
    // these loads.)
    float pdp[16][NFB];
    auto load_pd = [&](int ps) {
      const float* const pdn = a.Pd + NFB * nq;
#pragma unroll
      for (int j = 0; j < 16 / NFB; ++j) {
        const int i = (16 / NFB) * ps + j;
        const unsigned di = static_cast<unsigned>(row_bcast(dx0, i));
        piece_load<L>(pdn + static_cast<size_t>(DBG(2) ? 0u : di) * L, pdp[i]);
      }
    };
    if constexpr (EMB) {
#pragma unroll
      for (int ps = 0; ps < NFB; ++ps) load_pd(ps);
    } else {
#if AMBER_EDGE_PD_EARLY
#pragma unroll
      for (int ps = 0; ps < NFB; ++ps) load_pd(ps);
      if (!DBG(8)) gemm_acc<L, SPLIT, true, AMBER_KSTEP_GROUP>(acc, et, scale, win, wo, smem, lane);
#else
      if (!DBG(8)) gemm_acc<L, SPLIT, true, AMBER_KSTEP_GROUP>(acc, et, scale, win, wo, smem, lane);
#pragma unroll
      for (int ps = 0; ps < NFB; ++ps) load_pd(ps);
#endif
      us = inv * wsinv0;
    }
    STAMP(4);  // layer 0 contraction
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) acc.a[fb][i] = fmaf(acc.a[fb][i], us, pdp[i][fb]);
    if constexpr (VAR == 0) tile_act_fast<L>(acc, a.act); else if constexpr (VAR == 2) tile_tanh<L>(acc); else tile_act_var<L>(acc, a.act);
    // Where this half's first row began and its last row ends (rowptr words loaded at the top of the tile).  Taken here,
    // where the in-order memory counter has just been waited down for the P_d rows anyway: read after the stores of e'
    // they would make the wave wait until those stores have drained.
    const int64_t q0 = p0 + 16 * h;
    int sb_i = q0 > static_cast<int64_t>(rp_first) ? 1 : 0;          // the half's first row began before it
    int ends_last_i = (q0 + nvh) == static_cast<int64_t>(rp_last) ? 1 : 0;  // its last row ends with it
    asm volatile("" : "+v"(sb_i), "+v"(ends_last_i));  // evaluated HERE, not sunk to the point of use
    STAMP(5);  // P_d rows, activation
    // ---- hidden layers: the activations go through the window as a [feature][edge] image ----
    const bool res_rows = !EMB && a.residual == AMBER_MODE_INNER && !DBG(1);
    // one hidden layer: acc <- act(W_l acc + b_l); `hook(ps)` as in gemm_acc
    auto hidden_layer = [&](int l, auto&& hook) {
      const float* wg = a.layers[l].w;
      float bv[NFB];
      if (EARLY && l == a.nl - 1) {
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) bv[fb] = e_b1[fb];
      } else {
        piece_load<L>(a.layers[l].b + NFB * nq, bv);
      }
      pow2_scale(wave_max(tile_absmax<L>(acc)), scale, inv);
      // accumulate from zero (the first MFMA of a block takes the constant 0 as its C operand: no 64 moves to seed the
      // accumulators) and add the bias with the multiply that takes the result back out of the scaled domain
      Tile<L> nxt;
      tile_zero<L>(nxt);
      if (!DBG(8)) {
        gemm_acc<L, SPLIT, true, AMBER_KSTEP_GROUP>(nxt, acc, scale, win, wo, smem + l * MAT_BYTES, lane, hook);
      } else {
#pragma unroll
        for (int ps = 0; ps < NFB; ++ps) hook(ps);
      }
      const float post = inv * wg[L * L];
#ifdef AMBER_EPI_PK
      if constexpr (VAR == 0) {
        tile_bias_act_pk<L>(nxt, post, bv, a.act);
      } else
#endif
      {
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
#pragma unroll
          for (int i = 0; i < 16; ++i) nxt.a[fb][i] = fmaf(nxt.a[fb][i], post, bv[fb]);
        }
        if constexpr (VAR == 0) tile_act_fast<L>(nxt, a.act); else if constexpr (VAR == 2) tile_tanh<L>(nxt); else tile_act_var<L>(nxt, a.act);
      }
      acc = nxt;
    };
#pragma unroll 1
    for (int l = 1; l < a.nl - 1; ++l) hidden_layer(l, NoPassHook());
    if (a.nl > 1) hidden_layer(a.nl - 1, NoPassHook());
    // the next tile's indices: fetched here, ahead of this tile's stores (a load behind them would wait for them to drain)
    // and after the contractions, whose registers they would otherwise share
    if (has_next) load_idx(tnext, sx1, d1, dx1);
    STAMP(6);  // hidden layers
    // ---- residual: the rows of e again (this wave's own loads of a moment ago), now feature per lane ----
    auto add_residual = [&]() __attribute__((always_inline)) {
      if constexpr (EMB) {
        Tile<L> e0;
        embed_tile(e0);
        tile_add<L>(acc, e0);
      } else {
        tile_add_rows<L>(acc, et);  // the rows of e, still in their registers (row-piece order: see tile_add_rows)
      }
    };
    if (EMB && a.residual == AMBER_MODE_INNER) {
      add_residual();
    } else if (res_rows) {
      add_residual();
    }
    if constexpr (PREFETCH_E)  // (no branch: the last tile fetches its own rows again, unread)
      load_et(has_next ? tnext : tile, hq, nq);
    STAMP(7);  // residual
    // ---- LayerNorm over the features of every edge: across the lanes of a half, one register per edge ----
    if (a.layer_norm == AMBER_MODE_INNER && !DBG(64)) {
      if constexpr (EARLY) tile_layernorm_acc_pre<L>(acc, e_g, e_bt, a.eps);
      else tile_layernorm_acc<L>(acc, a.gamma, a.beta, a.eps, nq);
    }
    STAMP(8);  // LayerNorm
    // "outer" residual / LayerNorm (VAR = 1 only): what the node module aggregates is the value at THIS point
    // (message_passing_block.py:119-123), so the per-destination sums run first and the outer operations, then the stores
    // of e', follow them
    const bool outer_ops = GEN && (a.residual == AMBER_MODE_OUTER || a.layer_norm == AMBER_MODE_OUTER);
    // (the addresses of the stores and partial sums are formed from here on: derived at the tile's top from the same lane
    // offsets they are live - and spilled - across both contractions)
    asm volatile("" : "+v"(hq), "+v"(nq));
    // ---- stores of e' (two whole 128-byte row pieces per register) ----
    // (a two-pass loop instead of a lambda for the stores: called through a lambda - even force-inlined - the default
    // instantiation's tile went through scratch.  VAR = 0: one pass, stores then sums, the straight-line code it always was;
    // VAR = 1: the sums and the outer operations in pass 0, the stores in pass 0 (no outer operation) or pass 1)
#pragma unroll
    for (int pass = 0; pass < (GEN ? 2 : 1); ++pass) {
    if (pass == (outer_ops ? 1 : 0)) {
    if (a.write_e || a.out_edge) {
      if (a.write_e && !DBG(16)) {
        if (nvalid == kTile) {
#pragma unroll
          for (int g = 0; g < 16 / kRowGroup<L>; ++g) {
            float* const pg = eo + static_cast<unsigned>(hq) + kRowGroup<L> * g * L;  // one address per group, see the residual
#pragma unroll
            for (int j = 0; j < kRowGroup<L>; ++j) {
              float pc[NFB];
              tile_get_piece<L>(acc, kRowGroup<L> * g + j, pc);
              piece_store<L>(pg + j * L, pc);
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            if (i + 16 * h < nvalid) {
              float pc[NFB];
              tile_get_piece<L>(acc, i, pc);
              piece_store<L>(eo + static_cast<unsigned>(hq) + i * L, pc);
            }
          }
        }
      }
      if (a.out_edge) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int r = i + 16 * h;
          const unsigned orow = static_cast<unsigned>(row_bcast(pm, i));
          if (r < nvalid) {
            float pc[NFB];
            tile_get_piece<L>(acc, i, pc);
            piece_store<L>(a.out_edge + static_cast<size_t>(orow) * L + NFB * nq, pc);
          }
        }
      }
    }
    }
    STAMP(9);  // stores of e'
    if (pass == 0) {
    // ---- per-destination sums in CSR order: every lane adds up its own registers (its half's sixteen consecutive
    // edges); a row that ends inside the half goes to agg, rows cut by the half's borders to head / tail of the
    // reduction tile 2 tile + h (see mpn_kernels.hip); the two halves differ only in WHERE their rows end, which the
    // wave handles by predication ----
    {
      // positions beyond the end do not count: the default instantiation zeroes them in the tile (its rows are stored by
      // now); the general one, whose tile still goes through the outer operations, skips them in the sums
      if (!GEN && nvalid < kTile) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (i + 16 * h >= nvalid) {
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) acc.a[fb][i] = 0.f;
          }
      }
      if (!DBG(4)) {
        bool sb = sb_i != 0;
        const bool ends_last = ends_last_i != 0;
        const unsigned mybits = h ? (newseg >> 16) : (newseg & 0xffffu);
        const unsigned anybits = (newseg | (newseg >> 16)) & 0xfffeu;  // wave-uniform: some half ends a row before position i
        float* const slot_head = a.head + (2 * static_cast<int64_t>(tile) + h) * L + NFB * nq;
        float* const slot_tail = a.tail + (2 * static_cast<int64_t>(tile) + h) * L + NFB * nq;
        float* const aggn = a.agg + NFB * nq;
        float sum[NFB];
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) sum[fb] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i > 0 && ((anybits >> i) & 1u)) {
            const unsigned dp = static_cast<unsigned>(row_bcast(dx0, i - 1));  // outside the predicated part: all lanes take part
            if ((mybits >> i) & 1u) {
              piece_store<L>(sb ? slot_head : aggn + static_cast<size_t>(dp) * L, sum);
#pragma unroll
              for (int fb = 0; fb < NFB; ++fb) sum[fb] = 0.f;
              sb = false;
            }
          }
#pragma unroll
          for (int fb = 0; fb < NFB; ++fb) {
            float r = sum[fb] + ((!GEN || i + 16 * h < nvalid) ? acc.a[fb][i] : 0.f);
            asm("" : "+v"(r));  // scalar: paired over feature blocks the addends would be gathered with two moves a pair
            sum[fb] = r;
          }
        }
        if (nvh > 0) {
          float* const dstp = (!sb && ends_last) ? aggn + static_cast<size_t>(static_cast<unsigned>(lastd)) * L : (sb ? slot_head : slot_tail);
          piece_store<L>(dstp, sum);
        }
      }
    }
    if constexpr (GEN) {
      if (outer_ops) {
        if (a.residual == AMBER_MODE_OUTER) add_residual();
        if (a.layer_norm == AMBER_MODE_OUTER) {
          if constexpr (EARLY) tile_layernorm_acc_pre<L>(acc, e_g, e_bt, a.eps);
          else tile_layernorm_acc<L>(acc, a.gamma, a.beta, a.eps, nq);
        }
      }
    }
    }  // pass == 0
    }  // passes
    sx0 = sx1;
    d0 = d1;
    dx0 = dx1;
    STAMP(10);  // per-destination sums
  }
#ifdef AMBER_STAMPS
  if (lane == 0) {
    for (int k = 0; k < kStamps; ++k) atomicAdd(&g_stamps[k], static_cast<unsigned long long>(st_acc[k]));
    atomicAdd(&g_stamps[kStamps], static_cast<unsigned long long>(st_tiles));
  }
#endif
}

// ---------------------------------------------------------------------------------------------------
// node kernel, same orientation: finish K6 (combine partial sums, divide by the in-degree), K7 (node MLP), K8
// (residual + LayerNorm) and the next edge step's first-layer projections P_s / P_d.
// reference: MessagePassingNodeModule.forward (src/mpn/message_passing_modules.py:61-72),
//            MessagePassingBlock.forward (src/mpn/message_passing_block.py:123-128)
//
// A node tile is 32 consecutive rows; x and the aggregate arrive as whole row pieces, x', P_s and P_d leave straight
// from the accumulator layout.  The update needs 2 + (nl - 1) weight matrices and the projections 2 more: at L = 128
// only two of them (64 KB each) fit beside the waves' windows, so the workgroup's 8 waves walk the matrices together,
// one PHASE per matrix, each wave on its own tile: while a phase contracts with the matrix in one LDS slot the next
// matrix streams into the other slot by LDS-DMA (global_load_lds), and one workgroup barrier separates the phases.
// When all matrices fit (L <= 64) they are loaded once and the phases run without barriers.
// ---------------------------------------------------------------------------------------------------
// HEAD: the instantiation of the LAST node launch of a SupervisedMPN forward - no projections follow, the decoder's
// layers and the readout do (kept apart so that neither variant pays the other's registers).
#ifdef AMBER_STAMPS
__device__ unsigned long long g_nstamps[kStamps + 1];
#endif
// VAR: as in edge_kernel_f2 (1 = tanh as well, and the residual-outer + LayerNorm-inner order y = LN(m) + x; the other outer
// combinations are the inner forms for node rows, which have no second consumer)
template <int L, bool SPLIT, bool HEAD, int VAR = 0>
__global__ __launch_bounds__(64 * node_waves<L>()) void node_kernel_f2(const NodeArgs a, const int64_t ntiles, const int resident) {
  constexpr int NFB = L / 32;
  constexpr int MAT_BYTES = L * L * 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave id as a scalar
  const int n = lane & 31, h = lane >> 5;
  const int nu = a.do_update ? a.nl + 1 : 0;   // phases of the update: W0x, W0a, hidden layers
  const int np = (!HEAD && a.do_proj) ? 2 : 0;  // phases of the projections: W0s, W0d
  const int nd = HEAD ? a.n_dense : 0;          // phases of the decoder
  const int P = nu + np + nd;                   // phases per batch of tiles
  // matrix of phase k (wave-uniform)
  auto mat = [&](int k) -> const float* {
    if (k < nu) return k == 0 ? a.w0x : (k == 1 ? a.w0a : a.layers[k - 1].w);
    if (k < nu + np) return k == nu ? a.w0s : a.w0d;
    return a.dense_f2[k - nu - np];
  };
  // this wave's share of one matrix, global -> LDS slot, no registers (1 KB per instruction)
  auto stream_in = [&](const float* src, int slot) {
    int lo16 = lane * 16;  // opaque: hoisted out of the batch loop, the per-matrix source addresses would be spilled
    asm volatile("" : "+v"(lo16));
    for (int j = wave; j < MAT_BYTES / 1024; j += node_waves<L>())
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(reinterpret_cast<const unsigned char*>(src) + j * 1024 + static_cast<unsigned>(lo16)),
          (__attribute__((address_space(3))) void*)(smem + slot * MAT_BYTES + j * 1024), 16, 0, 0);
  };
  const int nslots = resident ? P : 2;
  unsigned char* win = smem + nslots * MAT_BYTES + wave * kWinBytes;
  const WinOff wo = make_win_off(lane);

  const int64_t xcd = blockIdx.x & 7, local = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  const int64_t tiles_per_xcd = (ntiles + 7) >> 3;
  const int64_t t_begin = xcd * tiles_per_xcd;
  const int64_t t_end = t_begin + tiles_per_xcd < ntiles ? t_begin + tiles_per_xcd : ntiles;
  const int64_t stride = bpx * node_waves<L>();
  const int64_t first = t_begin + local * node_waves<L>();
  const int64_t nbatch = first < t_end ? (t_end - first + stride - 1) / stride : 0;  // the same for every wave of the workgroup
  if (nbatch == 0) return;

  // Ring mode hand-over without draining the stores: vector memory operations complete in issue order, so each wave
  // issues one ordinary "marker" load right behind its share of the LDS-DMA and waits for that load (a counted
  // s_waitcnt that leaves the younger stores in flight) before the barrier that opens the phase reading the slot.
  float marker = 0.f;
  auto marker_load = [&](const float* src) { marker = src[lane]; };
  if (resident) {
    // all matrices stream in by LDS-DMA; the wait for them sits in the FIRST phase's begin_phase, behind the first tile's
    // row loads, so that on a one-mesh graph (one or two tiles a wave) the prologue runs under their latency
    for (int k = 0; k < P; ++k) stream_in(mat(k), k);
  } else {
    stream_in(mat(0), 0);
    marker_load(mat(0));
  }
  int gphase = 0;  // phases begun so far (ring mode: slot = gphase & 1)
  const int64_t total_phases = nbatch * P;
  // begin a phase: returns the LDS address of its matrix.  Ring mode: the barrier tells that every wave has left the
  // previous phase (whose slot is refilled now) and that every wave's share of this phase's matrix has landed.
  auto begin_phase = [&](int k) -> const unsigned char* {
    int slot = k;
    if (resident) {
      if (gphase == 0) {  // every wave's share of the matrices has landed (vector memory completes in order)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    } else {
      asm volatile("" ::"v"(marker));  // the compiler waits for the marker load here: this wave's DMA share is in LDS
      if (!DBG(512)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      slot = gphase & 1;
      if (gphase + 1 < total_phases) {
        const float* nm = mat(k + 1 < P ? k + 1 : 0);
        if (!DBG(1024)) {
          stream_in(nm, slot ^ 1);
          marker_load(nm);
        }
      }
    }
    ++gphase;
    return smem + slot * MAT_BYTES;
  };
  // Staggered wave groups (ring mode, AMBER_NODE_STAGGER): the two waves of a SIMD (w and w + waves / 2) pass each phase
  // barrier at DIFFERENT points of their program - the first group where it always was (after the row work that follows
  // a contraction, right before the next contraction), the second group right after its contraction, BEFORE that row
  // work.  Between two barriers the first group then contracts and does row work, the second does (the previous
  // phase's) row work and contracts: the SIMD's matrix pipe serves one wave at a time and one wave's loads / stores run
  // under the other's MFMAs, instead of both contracting and then both waiting on memory.  The ring's hand-over rules
  // are untouched (a wave arrives at barrier k after its contraction k - 1 and after its share of matrix k has landed).
#ifndef AMBER_NODE_STAGGER
#define AMBER_NODE_STAGGER 0  // measured in round 3: node launch +1 % (1.251 vs 1.238 ms), see DESIGN.md section 5
#endif
  const bool early = AMBER_NODE_STAGGER && !resident && wave >= node_waves<L>() / 2;  // wave-uniform
  bool pend = false;                  // this wave has already passed the barrier of its next phase
  const unsigned char* wl_pend = nullptr;
  auto phase_begin = [&](int kk) -> const unsigned char* {
    if (!pend) wl_pend = begin_phase(kk);
    pend = false;
    return wl_pend;
  };
  // after the contraction of a phase; kn = the phase that follows (0 = the next batch's first)
  auto phase_done = [&](int kn) {
    if (early && gphase < total_phases) {
      wl_pend = begin_phase(kn);
      pend = true;
    }
  };

#ifdef AMBER_STAMPS
  unsigned st_acc[kStamps] = {};
  unsigned st_tiles = 0;
  unsigned long long st_prev = __builtin_amdgcn_s_memtime();
#endif
  for (int64_t b = 0; b < nbatch; ++b) {
    STAMP(0);
#ifdef AMBER_STAMPS
    ++st_tiles;
#endif
    // A wave without a tile of its own keeps the workgroup's barriers: it recomputes the last tile of the range and
    // stores nothing (nstore = 0) - cheaper for the register allocator than guarding every phase.
    const int64_t mytile = first + b * stride + wave;
    const int64_t tile = mytile < t_end ? mytile : t_end - 1;
    const int64_t n0 = tile * kTile;
    const int nvalid = (a.N - n0) < kTile ? static_cast<int>(a.N - n0) : kTile;  // rows of the tile that exist (>= 1)
    const int nstore = mytile < t_end ? nvalid : 0;                              // rows this wave stores
    const int rlim = nvalid - 1;
    const float* const xt = a.x_in + n0 * L;
    // (h4: the row numbers 8 (i >> 2) + (i & 3) + 4 h of the 16 registers are cheaper to recompute than to keep -
    // hoisted out of the loop they are spilled, and every scratch reload drains the in-order memory counter)
    int hq = 4 * h * L + NFB * n, nq = n, lq = lane, h4 = 4 * h;  // hq: first row of the half at this lane's NFB features
    asm volatile("" : "+v"(hq), "+v"(nq), "+v"(lq), "+v"(h4));
    // the sixteen row numbers are formed again in every phase that uses them (one add each, partial tiles only): shared
    // between the phases they stay live across the contractions and are spilled inside the tile loop
#ifndef AMBER_NODE_NO_REPIN
#define AMBER_REPIN_H4 asm volatile("" : "+v"(h4), "+v"(hq))
#else
#define AMBER_REPIN_H4
#endif
    Tile<L> v;  // the tile's node features in accumulator layout (after the update: x')
    int k = 0;
    // L <= 64 (registers allow it): the tile-invariant vectors of the update - first- and second-layer bias, LayerNorm gamma /
    // beta - are fetched HERE, with the tile's first loads, instead of where they are used
#ifndef AMBER_NODE_EARLY_MAXL
#define AMBER_NODE_EARLY_MAXL 64
#endif
    constexpr bool EARLY = L <= AMBER_NODE_EARLY_MAXL;
    constexpr bool EARLY_HD = L <= 32 && EARLY;  // (at 64 the extra landing registers spill 15-50)
    float e_b0[NFB], e_b1[NFB], e_g[NFB], e_bt[NFB];
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) e_b0[fb] = e_b1[fb] = e_g[fb] = e_bt[fb] = 0.f;
    if constexpr (EARLY) {
      if (a.do_update) {
        piece_load<L>(a.b0 + NFB * nq, e_b0);
        if (a.nl > 1) piece_load<L>(a.layers[1].b + NFB * nq, e_b1);
        if (a.layer_norm != AMBER_MODE_NONE) {
          piece_load<L>(a.gamma + NFB * nq, e_g);
          piece_load<L>(a.beta + NFB * nq, e_bt);
        }
      }
    }
    if (a.do_update) {
      // in-degrees and reduction slots of the tile's rows (used by the aggregate phase, loaded a phase ahead)
      int dg[4], tf[4], tl[4];
      int64_t nd[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = it * 8 + (lq >> 3);
        nd[it] = n0 + (r < rlim ? r : rlim);
        const int r0 = a.rowptr[nd[it]], r1 = a.rowptr[nd[it] + 1];
        dg[it] = r1 - r0;
        tf[it] = r0 >> a.red_shift;
        tl[it] = dg[it] > 0 && !DBG(2048) ? (r1 - 1) >> a.red_shift : tf[it];
      }
      Tile<L> acc;
      float scale, inv;
      ELanding<L> lg;  // whole row pieces of the aggregate
      ELanding<L> hd;  // ... and of the first head slot of rows a reduction tile's border cut
      {
        // ---- whole row pieces of x ----
        ELanding<L> lx;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int r = it * 8 + (lq >> 3);
          const int rc = r < rlim ? r : rlim;  // rows beyond the end repeat the last node and are never stored
          const f32x4* px = reinterpret_cast<const f32x4*>(xt + static_cast<unsigned>(rc * L)) + (lq & 7);
#pragma unroll
          for (int ps = 0; ps < NFB; ++ps) lx.v[ps][it] = px[ps * 8];
        }
        pow2_scale(wave_max(landing_absmax<L>(lx)), scale, inv);
        STAMP(1);  // x rows arrived
        tile_zero<L>(acc);  // the first MFMAs take the constant 0 as C; the bias joins the final rescale
        const unsigned char* wl = phase_begin(k++);
        STAMP(2);  // barrier
#ifndef AMBER_NODE_NO_PREFETCH_AGG
        // The rows of sums (agg, or the tail slot when a reduction tile's border cut the row) are fetched under THIS
        // contraction: every pass leaves a quarter of x's landing registers, and that quarter of the aggregate's pieces is
        // loaded into them at once - the waves of a workgroup walk the phases together, so nothing else would cover the
        // latency of these loads (10 k of a tile's 102 k cycles when they were issued after the phase).
        if (!DBG(256))
          gemm_rows<L, SPLIT, true>(acc, lx, scale, win, wo, wl, lane, [&](int ps) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const float* src = tl[it] != tf[it] ? a.tail + static_cast<int64_t>(tf[it]) * L : a.agg + nd[it] * L;
              lg.v[ps][it] = (reinterpret_cast<const f32x4*>(src) + (lq & 7))[ps * 8];
              if constexpr (EARLY_HD) {  // the first head slot of a cut row in the same breath (L = 32: registers allow it)
                hd.v[ps][it] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (tl[it] != tf[it])
                  hd.v[ps][it] = (reinterpret_cast<const f32x4*>(a.head + static_cast<int64_t>(tf[it] + 1) * L) + (lq & 7))[ps * 8];
              }
            }
          });
        else
#endif
        {
          if (!DBG(256)) gemm_rows<L, SPLIT, true>(acc, lx, scale, win, wo, wl, lane);
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const float* src = tl[it] != tf[it] ? a.tail + static_cast<int64_t>(tf[it]) * L : a.agg + nd[it] * L;
            const f32x4* pg = reinterpret_cast<const f32x4*>(src) + (lq & 7);
#pragma unroll
            for (int ps = 0; ps < NFB; ++ps) lg.v[ps][it] = pg[ps * 8];
          }
        }
        phase_done(k < P ? k : 0);
        STAMP(3);  // W0x
      }
      {
        // ---- whole row pieces of the aggregate: sums of the edge kernel -> mean.  After the row of sums: the first
        // head slot of a cut row; rows cut more than once (longer than a reduction tile) add their further heads in a loop.
        {
#if !defined(AMBER_NODE_NO_PREFETCH_AGG)
          if (!EARLY_HD || DBG(256))
#endif
          {
#pragma unroll
          for (int it = 0; it < 4; ++it) {
#pragma unroll
            for (int ps = 0; ps < NFB; ++ps) hd.v[ps][it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (tl[it] != tf[it]) {
              const f32x4* ph = reinterpret_cast<const f32x4*>(a.head + static_cast<int64_t>(tf[it] + 1) * L) + (lq & 7);
#pragma unroll
              for (int ps = 0; ps < NFB; ++ps) hd.v[ps][it] = ph[ps * 8];
            }
          }
          }
#pragma unroll
          for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int ps = 0; ps < NFB; ++ps) lg.v[ps][it] += hd.v[ps][it];
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          for (int t = tf[it] + 2; t <= tl[it]; ++t) {
            const f32x4* ph = reinterpret_cast<const f32x4*>(a.head + static_cast<int64_t>(t) * L) + (lq & 7);
#pragma unroll
            for (int ps = 0; ps < NFB; ++ps) lg.v[ps][it] += ph[ps * 8];
          }
          // mean; zero in-degree -> 0 (scatter_mean semantics; that row of agg was never written)
          const float rd = 1.0f / static_cast<float>(dg[it] > 1 ? dg[it] : 1);
#pragma unroll
          for (int ps = 0; ps < NFB; ++ps)
#pragma unroll
            for (int t = 0; t < 4; ++t) lg.v[ps][it][t] = dg[it] > 0 ? lg.v[ps][it][t] * rd : 0.f;
        }
        // x and the aggregate feed one accumulator (W0x and W0a share their pack-time scale): move the accumulator
        // from x's power-of-two scale to the aggregate's (exact)
        float sg, ig;
        pow2_scale(wave_max(landing_absmax<L>(lg)), sg, ig);
        tile_scale<L>(acc, sg * inv);
        STAMP(4);  // aggregate rows arrived, mean
        const unsigned char* wl = phase_begin(k++);
        STAMP(2);
        if (!DBG(256)) gemm_rows<L, SPLIT, true>(acc, lg, sg, win, wo, wl, lane);
        phase_done(k < P ? k : 0);
        STAMP(5);  // W0a
        {
          const float post = ig * a.w0x[L * L];
          float bv[NFB];
          if constexpr (EARLY) {
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) bv[fb] = e_b0[fb];
          } else {
            piece_load<L>(a.b0 + NFB * nq, bv);
          }
#pragma unroll
          for (int fb = 0; fb < NFB; ++fb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc.a[fb][i] = fmaf(acc.a[fb][i], post, bv[fb]);
          }
        }
        if constexpr (VAR == 0) tile_act_fast<L>(acc, a.act); else if constexpr (VAR == 2) tile_tanh<L>(acc); else tile_act_var<L>(acc, a.act);
      }
#pragma unroll 1
      for (int l = 1; l < a.nl; ++l) {
        STAMP(6);  // bias, activation
        const unsigned char* wl = phase_begin(k++);
        STAMP(2);
        const float* wg = a.layers[l].w;
        pow2_scale(wave_max(tile_absmax<L>(acc)), scale, inv);
        Tile<L> nxt;
        tile_zero<L>(nxt);
        if (!DBG(256)) gemm_acc<L, SPLIT, true>(nxt, acc, scale, win, wo, wl, lane);
        phase_done(k < P ? k : 0);
        STAMP(7);  // hidden layer
        {
          const float post = inv * wg[L * L];
          float bv[NFB];
          if (EARLY && l == 1) {
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) bv[fb] = e_b1[fb];
          } else {
            piece_load<L>(a.layers[l].b + NFB * nq, bv);
          }
#pragma unroll
          for (int fb = 0; fb < NFB; ++fb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) nxt.a[fb][i] = fmaf(nxt.a[fb][i], post, bv[fb]);
          }
        }
        if constexpr (VAR == 0) tile_act_fast<L>(nxt, a.act); else if constexpr (VAR == 2) tile_tanh<L>(nxt); else tile_act_var<L>(nxt, a.act);
        acc = nxt;
      }
      auto node_ln = [&](Tile<L>& t) __attribute__((always_inline)) {
        if constexpr (EARLY) tile_layernorm_acc_pre<L>(t, e_g, e_bt, a.eps);
        else tile_layernorm_acc<L>(t, a.gamma, a.beta, a.eps, nq);
      };
      auto add_x_residual = [&]() __attribute__((always_inline)) {
        AMBER_REPIN_H4;
        if (nvalid == kTile) {
          // one 64-bit address per group of four registers; rows and feature blocks inside it are immediate offsets
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float* const pg = xt + static_cast<unsigned>(hq) + 8 * g * L;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float pc[NFB];
              piece_load<L>(pg + j * L, pc);
#pragma unroll
              for (int fb = 0; fb < NFB; ++fb) acc.a[fb][4 * g + j] += pc[fb];
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            float pc[NFB];
            piece_load<L>(xt + static_cast<unsigned>(min(8 * (i >> 2) + (i & 3) + h4, rlim) * L + NFB * n), pc);
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) acc.a[fb][i] += pc[fb];
          }
        }
      };
      if constexpr ((VAR & 1) == 0) {
        // order of message_passing_block.py:124-125: inner residual, inner LayerNorm
        if (a.residual == AMBER_MODE_INNER) add_x_residual();
        if (a.layer_norm == AMBER_MODE_INNER && !DBG(4096)) node_ln(acc);
      } else {
        // message_passing_block.py:124-128: inner residual, inner LayerNorm, outer residual, outer LayerNorm.  Node rows have
        // no consumer in between, so only residual-outer + LayerNorm-inner differs from "residual, then LayerNorm"
        if (a.residual == AMBER_MODE_OUTER && a.layer_norm == AMBER_MODE_INNER) {
          node_ln(acc);
          add_x_residual();
        } else {
          if (a.residual != AMBER_MODE_NONE) add_x_residual();
          if (a.layer_norm != AMBER_MODE_NONE) node_ln(acc);
        }
      }
      v = acc;
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float pc[NFB];
        piece_load<L>(xt + static_cast<unsigned>(min(8 * (i >> 2) + (i & 3) + h4, rlim) * L + NFB * n), pc);
        tile_set_piece<L>(v, i, pc);
      }
    }
    // store a tile in accumulator layout as rows n0 .. n0 + nvalid - 1 of `out`
    auto store_tile = [&](float* out, const Tile<L>& t) {
      float* const ot = out + n0 * L;
      AMBER_REPIN_H4;
      if (nstore == kTile) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float* const pg = ot + static_cast<unsigned>(hq) + 8 * g * L;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float pc[NFB];
            tile_get_piece<L>(t, 4 * g + j, pc);
            piece_store<L>(pg + j * L, pc);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (8 * (i >> 2) + (i & 3) + h4 < nstore) {
            float pc[NFB];
            tile_get_piece<L>(t, i, pc);
            piece_store<L>(ot + static_cast<unsigned>(hq + (8 * (i >> 2) + (i & 3)) * L), pc);
          }
        }
      }
    };
    STAMP(8);  // bias, act, residual, LayerNorm
    // the bias of the second projection, fetched before the tile's stores: vector memory completes in order, so a load
    // issued behind the stores of x' and P_s would make its user wait until those have drained
    float bv_d[NFB];
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) bv_d[fb] = 0.f;
    if (!HEAD && a.do_proj) piece_load<L>(a.b0e + NFB * nq, bv_d);
    if (a.x_out && a.do_update) store_tile(a.x_out, v);
    if (!HEAD && a.do_proj) {
      float scale, inv;
      pow2_scale(wave_max(tile_absmax<L>(v)), scale, inv);
      {
        STAMP(9);  // store x'
        const unsigned char* wl = phase_begin(k++);
        STAMP(2);
        Tile<L> acc;
        tile_zero<L>(acc);
        if (!DBG(256)) gemm_acc<L, SPLIT, true>(acc, v, scale, win, wo, wl, lane);
        phase_done(k < P ? k : 0);
        tile_scale<L>(acc, inv * a.w0s[L * L]);
        store_tile(a.Ps, acc);
      }
      {
        STAMP(10);  // W0s, store P_s
        const unsigned char* wl = phase_begin(k++);
        STAMP(2);
        // x' stays in registers across the first projection.  (Round 1 re-read it here at L = 128, the wave's own stores of
        // a moment ago, for want of registers: with the in-order memory counter that load waited for the stores of x' and
        // P_s to drain - 14 k of a tile's 113 k cycles.  AMBER_NODE_RELOAD_V brings the re-read back for comparison.)
        if constexpr (AMBER_NODE_RELOAD_V) {
          const float* const vt = (a.do_update ? a.x_out : a.x_in) + n0 * L;
          AMBER_REPIN_H4;
          if (nvalid == kTile) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float* const pg = vt + static_cast<unsigned>(hq) + 8 * g * L;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                float pc[NFB];
                piece_load<L>(pg + j * L, pc);
                tile_set_piece<L>(v, 4 * g + j, pc);
              }
            }
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              float pc[NFB];
              piece_load<L>(vt + static_cast<unsigned>(min(8 * (i >> 2) + (i & 3) + h4, rlim) * L + NFB * nq), pc);
              tile_set_piece<L>(v, i, pc);
            }
          }
        }
        Tile<L> acc;
        tile_zero<L>(acc);
        if (!DBG(256)) gemm_acc<L, SPLIT, true>(acc, v, scale, win, wo, wl, lane);
        phase_done(k < P ? k : 0);
        {
          const float post = inv * a.w0d[L * L];
#pragma unroll
          for (int fb = 0; fb < NFB; ++fb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc.a[fb][i] = fmaf(acc.a[fb][i], post, bv_d[fb]);
          }
        }
        store_tile(a.Pd, acc);
        STAMP(11);  // reload x', W0d, store P_d
      }
    }
    // ---- SupervisedMPN head on the updated rows (K9, K10: src/algorithm/architecture/supervised_mpn.py:51-54, MLP
    // mlp.py:60-95 with relu / leakyrelu and no latent normalisation): the decoder's layers as further phases ----
    if constexpr (HEAD) {
#pragma unroll 1
    for (int l = 0; l < a.n_dense; ++l) {
      const unsigned char* wl = phase_begin(k++);
      const float* wg = a.dense_f2[l];
      float scale, inv;
      pow2_scale(wave_max(tile_absmax<L>(v)), scale, inv);
      Tile<L> nxt;
      tile_zero<L>(nxt);
      gemm_acc<L, SPLIT, true>(nxt, v, scale, win, wo, wl, lane);
      phase_done(k < P ? k : 0);
      {
        const float post = inv * wg[L * L];
        float bv[NFB];
        piece_load<L>(a.dense[l].b + NFB * nq, bv);
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
#pragma unroll
          for (int i = 0; i < 16; ++i) nxt.a[fb][i] = fmaf(nxt.a[fb][i], post, bv[fb]);
        }
      }
      if constexpr (VAR == 0) tile_act_fast<L>(nxt, a.dense_act); else tile_act_var<L>(nxt, a.dense_act);
      v = nxt;
    }
    if (a.do_readout) {
      // readout Linear(L, 1): a dot product over the features = over the lanes of a half, one register per row
      float wq[NFB];
      piece_load<L>(a.wr + NFB * nq, wq);
      const float br = a.br[0];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float sacc = v.a[0][i] * wq[0];
#pragma unroll
        for (int fb = 1; fb < NFB; ++fb) sacc = fmaf(v.a[fb][i], wq[fb], sacc);
        const float tot = half_sum(sacc) + br;
        const int r = 8 * (i >> 2) + (i & 3) + h4;
        if (n == 0 && r < nstore) {
          const int64_t row = n0 + r;
          const int64_t oi = a.out_index ? static_cast<int64_t>(a.out_index[row]) : row;
          if (oi >= 0) a.out[oi] = a.post.on ? post_one(a.post, tot, oi) : tot;
        }
      }
    }
    }
  }
#ifdef AMBER_STAMPS
  if (lane == 0) {
    for (int k = 0; k < kStamps; ++k) atomicAdd(&g_nstamps[k], static_cast<unsigned long long>(st_acc[k]));
    atomicAdd(&g_nstamps[kStamps], static_cast<unsigned long long>(st_tiles));
  }
#endif
}

#ifndef AMBER_F2_GENERAL_TU
// ---------------------------------------------------------------------------------------------------
// weight packing: absmax -> power-of-two scale -> hi/lo fp16 in B-operand order (+ header)
// ---------------------------------------------------------------------------------------------------
// out: [s][fb][hi|lo][lane][8 halves]; element j of lane (h, r) is W[fcol(fb, r)][col0 + k], the output feature of
// accumulator column (fb, r) (see fcol), with k = 16 s + 8 h + j when the contraction's input comes from memory rows in
// their own order (kperm = 0) and k = fcol(s / 2, 16 (s % 2) + 8 h + j) when it comes from accumulator registers through
// the window ([feature][row] image of pass s / 2: image row 16 (s % 2) + 8 h + j is accumulator column (s / 2, that row));
// header (8 floats after the L*L*4 body bytes): [0] = inverse scale, [1] = scale
// Batched form (one launch pair for many matrices; a training step re-tiles every matrix after each optimizer step):
// a workgroup per job finds max|W| over the job's scale columns (max is order-independent: no atomics needed),
template <int L>
__global__ __launch_bounds__(256) void absmax_f2_batch_kernel(const F2PackBatch jobs) {
  __shared__ unsigned red[256];
  const F2PackJob jb = jobs.job[blockIdx.x];
  unsigned m = 0u;
  for (int i = threadIdx.x; i < L * jb.scale_cols; i += 256) {
    const unsigned v = __float_as_uint(fabsf(jb.W[static_cast<int64_t>(i / jb.scale_cols) * jb.ld + jb.scale_col0 + (i % jb.scale_cols)]));
    m = v > m ? v : m;  // non-negative floats order like their bit patterns
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + o] ? red[threadIdx.x] : red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) reinterpret_cast<unsigned*>(jb.out + L * L + 4)[0] = red[0];  // scratch word inside the header
}
// and blockIdx.y selects the job of the re-tiling kernel proper
template <int L>
__global__ void pack_f2_batch_kernel(const F2PackBatch jobs) {
  const F2PackJob jb = jobs.job[blockIdx.y];
  constexpr int NFB = L / 32, NS = L / 16;
  int ex = static_cast<int>((reinterpret_cast<const unsigned*>(jb.out + L * L + 4)[0] >> 23) & 0xffu);
  ex = ex < kMinScaleExp ? kMinScaleExp : ex;
  const float scale = __uint_as_float(static_cast<unsigned>(268 - ex) << 23);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (s, fb, lane)
  if (i == 0) {
    jb.out[L * L + 0] = __uint_as_float(static_cast<unsigned>(ex - 14) << 23);
    jb.out[L * L + 1] = scale;
  }
  if (i >= NS * NFB * 64) return;
  const int lane = i & 63, fb = (i >> 6) % NFB, s = (i >> 6) / NFB;
  const float* src = jb.W + static_cast<int64_t>(fcol<L>(fb, lane & 31)) * jb.ld + jb.col0;
  half8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int kk = 8 * (lane >> 5) + j;
    const int k = jb.kperm ? fcol<L>(s >> 1, 16 * (s & 1) + kk) : 16 * s + kk;
    const float ws = src[k] * scale;
    const _Float16 hh = static_cast<_Float16>(ws);
    hi[j] = hh;
    lo[j] = static_cast<_Float16>(ws - static_cast<float>(hh));
  }
  half8* o = reinterpret_cast<half8*>(jb.out);
  o[((s * NFB + fb) * 2 + 0) * 64 + lane] = hi;
  o[((s * NFB + fb) * 2 + 1) * 64 + lane] = lo;
}

template <int L>
static hipError_t launch_pack_f2_batch_t(const F2PackBatch& b, hipStream_t s) {
  hipLaunchKernelGGL(absmax_f2_batch_kernel<L>, dim3(b.n), dim3(256), 0, s, b);
  hipLaunchKernelGGL(pack_f2_batch_kernel<L>, dim3((L / 16 * (L / 32) * 64 + 255) / 256, b.n), dim3(256), 0, s, b);
  return hipGetLastError();
}
hipError_t launch_pack_f2_batch(int L, const F2PackBatch& b, hipStream_t s) {
  if (b.n == 0) return hipSuccess;
  if (b.n > kF2PackMaxJobs) return hipErrorInvalidValue;
  switch (L) {
    case 32: return launch_pack_f2_batch_t<32>(b, s);
    case 64: return launch_pack_f2_batch_t<64>(b, s);
    case 128: return launch_pack_f2_batch_t<128>(b, s);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_pack_f2(int L, const float* W, int ld, int col0, int scale_col0, int scale_cols, int kperm, float* out, hipStream_t s) {
  F2PackBatch b{};
  b.n = 1;
  b.job[0] = F2PackJob{W, out, ld, col0, scale_col0, scale_cols, kperm};
  return launch_pack_f2_batch(L, b, s);
}

#endif  // !AMBER_F2_GENERAL_TU

// hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device, size): a runtime call per launch is what a
// forward on a small graph (42 launches of ~30 us) can do without
static hipError_t allow_dynamic_lds(const void* fn, size_t lds) {
  struct Seen {
    const void* fn;
    int dev;
    size_t lds;
  };
  static thread_local Seen seen[64];
  static thread_local int nseen = 0;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  for (int i = 0; i < nseen; ++i)
    if (seen[i].fn == fn && seen[i].dev == dev && seen[i].lds >= lds) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  if (e != hipSuccess) return e;
  if (nseen < 64) seen[nseen++] = Seen{fn, dev, lds};
  return hipSuccess;
}
// Persistent grid: one workgroup per CU, a multiple of the 8 XCDs - but no more workgroups than there are tiles to hand
// every wave one (small graphs: a workgroup without a tile would still load the layer weights into its LDS).
static int64_t persistent_grid(int num_cus, int64_t ntiles, int waves) {
  int64_t grid = (num_cus / 8) * 8;
  if (grid < 8) grid = 8;
  const int64_t need = ((ntiles + waves - 1) / waves + 7) / 8 * 8;
  return need < grid ? (need < 8 ? 8 : need) : grid;
}

#ifndef AMBER_F2_GENERAL_TU
size_t edge_f2_lds_bytes(int L, int nl) {
  const int waves = L == 64 ? edge_waves<64>() : (L == 32 ? edge_waves<32>() : kF2Waves);
  return static_cast<size_t>(nl) * L * L * 4 + static_cast<size_t>(waves) * kWinBytes;
}

bool edge_f2_supported(int L, int nl, int act, int residual, int layer_norm) {
  (void)residual;
  (void)layer_norm;  // every residual / LayerNorm mode: the outer ones and tanh run on the general instantiations (VAR = 1, 2)
  return (act == AMBER_ACT_RELU || act == AMBER_ACT_LEAKYRELU || act == AMBER_ACT_TANH) && edge_f2_lds_bytes(L, nl) <= 160 * 1024;
}
#else
size_t edge_f2_lds_bytes(int L, int nl);
#endif
// The general instantiations (VAR != 0) live in translation units of their own (this file compiled with
// -DAMBER_F2_GENERAL_TU=1 / =2: build time, and the default instantiations' code generation stays untouched by them).
// Round 3 met rare, run-dependent wrong rows in these instantiations (an even edge row of a tile in lanes 32-63 with the
// fused embedding; the last node row of a tile in the outer-residual node kernel) and traced them to ONE instruction form
// the SLP vectoriser emits: a packed fp32 operation whose low result selects the high register of src1
// (v_pk_add_f32 / v_pk_fma_f32 ... op_sel:[0,1,..]).  Every translation unit of the library now goes through
// csrc/pkfix.py, which exchanges src0 and src1 of such instructions between compiler and assembler (the same selection on
// src0 is sound); tools/check_isa.py verifies the shipped library.  The experiments are in DESIGN.md section 5.3.
hipError_t launch_edge_f2_general(int L, int var, const EdgeArgs& a, int num_cus, hipStream_t s);
hipError_t launch_edge_f2_general_emb(int L, int var, const EdgeArgs& a, int num_cus, hipStream_t s);
hipError_t launch_node_f2_general(int L, const NodeArgs& a, int num_cus, hipStream_t s);
// which instantiation: 1 = general (an "outer" residual / LayerNorm; takes tanh too), 2 = the default structure with tanh
static int f2_variant(int act, int residual, int layer_norm) {
  if (residual == AMBER_MODE_OUTER || layer_norm == AMBER_MODE_OUTER) return 1;
  return act == AMBER_ACT_TANH ? 2 : 0;
}

template <int L, bool EMB, bool SPLIT, int VAR>
static hipError_t launch_edge_f2_t(const EdgeArgs& a, int num_cus, hipStream_t s) {
  const int64_t ntiles = (a.E + kTile - 1) / kTile;
  if (ntiles == 0) return hipSuccess;
  const size_t lds = edge_f2_lds_bytes(L, a.nl);
  hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(edge_kernel_f2<L, EMB, SPLIT, VAR>), lds);
  if (e != hipSuccess) return e;
  const int64_t grid = persistent_grid(num_cus, ntiles, edge_waves<L>());
#ifdef AMBER_STAMPS
  unsigned long long zero[kStamps + 1] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero, sizeof(zero));
#endif
  hipLaunchKernelGGL((edge_kernel_f2<L, EMB, SPLIT, VAR>), dim3(static_cast<unsigned>(grid)), dim3(64 * edge_waves<L>()), lds, s, a, ntiles);
#ifdef AMBER_STAMPS
  {
    (void)hipStreamSynchronize(s);
    unsigned long long got[kStamps + 1] = {};
    (void)hipMemcpyFromSymbol(got, HIP_SYMBOL(g_stamps), sizeof(got));
    static const char* names[kStamps] = {"loop-top", "issue-loads", "wait-e+max", "wait-Ps+scale", "gemm0", "Pd+act", "hidden", "residual", "LN", "stores", "segsum", ""};
    double tot = 0;
    for (int k = 0; k < kStamps; ++k) tot += static_cast<double>(got[k]);
    fprintf(stderr, "[stamps L=%d EMB=%d] tiles %llu, cycles/tile %.0f:", L, int(EMB), got[kStamps], tot / static_cast<double>(got[kStamps] ? got[kStamps] : 1));
    for (int k = 0; k < 11; ++k) fprintf(stderr, " %s %.0f", names[k], static_cast<double>(got[k]) / static_cast<double>(got[kStamps] ? got[kStamps] : 1));
    fprintf(stderr, "\n");
  }
#endif
  return hipGetLastError();
}
// AMBER_F2_GENERAL_TU: 1 = the general instantiations without the fused edge embedding (and the node kernel's), 2 = the general
// instantiations WITH the fused edge embedding (mpn_f2_general_emb.hip)
template <int L, int VAR>
static hipError_t launch_edge_f2_v(const EdgeArgs& a, int num_cus, hipStream_t s) {
#if !defined(AMBER_F2_GENERAL_TU)
  if (a.single) return a.emb_on ? launch_edge_f2_t<L, true, false, VAR>(a, num_cus, s) : launch_edge_f2_t<L, false, false, VAR>(a, num_cus, s);
  return a.emb_on ? launch_edge_f2_t<L, true, true, VAR>(a, num_cus, s) : launch_edge_f2_t<L, false, true, VAR>(a, num_cus, s);
#elif AMBER_F2_GENERAL_TU == 1
  return a.single ? launch_edge_f2_t<L, false, false, VAR>(a, num_cus, s) : launch_edge_f2_t<L, false, true, VAR>(a, num_cus, s);
#else
  return a.single ? launch_edge_f2_t<L, true, false, VAR>(a, num_cus, s) : launch_edge_f2_t<L, true, true, VAR>(a, num_cus, s);
#endif
}
#ifdef AMBER_F2_GENERAL_TU
#if AMBER_F2_GENERAL_TU == 1
hipError_t launch_edge_f2_general(int L, int var, const EdgeArgs& a, int num_cus, hipStream_t s) {
  if (a.emb_on) return launch_edge_f2_general_emb(L, var, a, num_cus, s);
#else
hipError_t launch_edge_f2_general_emb(int L, int var, const EdgeArgs& a, int num_cus, hipStream_t s) {
#endif
  switch (L) {
    case 32: return var == 1 ? launch_edge_f2_v<32, 1>(a, num_cus, s) : launch_edge_f2_v<32, 2>(a, num_cus, s);
    case 64: return var == 1 ? launch_edge_f2_v<64, 1>(a, num_cus, s) : launch_edge_f2_v<64, 2>(a, num_cus, s);
    case 128: return var == 1 ? launch_edge_f2_v<128, 1>(a, num_cus, s) : launch_edge_f2_v<128, 2>(a, num_cus, s);
    default: return hipErrorInvalidValue;
  }
}
#else
hipError_t launch_edge_f2(int L, const EdgeArgs& a, int num_cus, hipStream_t s) {
  if (a.E == 0) return hipSuccess;
  if (a.emb_on && (a.emb_f < 0 || a.emb_f > kMaxFusedEmbed || !a.emb_b || !a.emb_w0 || !a.perm)) return hipErrorInvalidValue;
  const int var = f2_variant(a.act, a.residual, a.layer_norm);
  if (var != 0) return launch_edge_f2_general(L, var, a, num_cus, s);
  switch (L) {
    case 32: return launch_edge_f2_v<32, 0>(a, num_cus, s);
    case 64: return launch_edge_f2_v<64, 0>(a, num_cus, s);
    case 128: return launch_edge_f2_v<128, 0>(a, num_cus, s);
    default: return hipErrorInvalidValue;
  }
}
#endif

#ifndef AMBER_F2_GENERAL_TU
bool node_f2_supported(int L, int act, int residual, int layer_norm) {
  (void)residual;
  (void)layer_norm;  // every mode: see f2_general
  return L <= 128 && (act == AMBER_ACT_RELU || act == AMBER_ACT_LEAKYRELU || act == AMBER_ACT_TANH);
}

bool head_f2_supported(int n_dense, int dense_act, int dense_norm) {
  return n_dense >= 1 && (dense_act == AMBER_ACT_RELU || dense_act == AMBER_ACT_LEAKYRELU || dense_act == AMBER_ACT_TANH) &&
         dense_norm == AMBER_NORM_NONE;
}
#endif  // !AMBER_F2_GENERAL_TU

template <int L, bool SPLIT, bool HEAD, int VAR>
static hipError_t launch_node_f2_t(const NodeArgs& a, int num_cus, hipStream_t s) {
  const int64_t ntiles = (a.N + kTile - 1) / kTile;
  if (ntiles == 0 || (!a.do_update && !a.do_proj && !HEAD)) return hipSuccess;
  if (HEAD && (a.do_proj || a.n_dense < 1 || !a.do_readout)) return hipErrorInvalidValue;
  const int P = (a.do_update ? a.nl + 1 : 0) + (a.do_proj ? 2 : 0) + (HEAD ? a.n_dense : 0);
  const size_t mat = static_cast<size_t>(L) * L * 4, wins = static_cast<size_t>(node_waves<L>()) * kWinBytes;
  const int resident = P * mat + wins <= 160 * 1024 ? 1 : 0;
  const size_t lds = (resident ? P : 2) * mat + wins;
  hipError_t e = allow_dynamic_lds(reinterpret_cast<const void*>(node_kernel_f2<L, SPLIT, HEAD, VAR>), lds);
  if (e != hipSuccess) return e;
  const int64_t grid = persistent_grid(num_cus, ntiles, node_waves<L>());
#ifdef AMBER_STAMPS
  unsigned long long zero[kStamps + 1] = {};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_nstamps), zero, sizeof(zero));
#endif
  hipLaunchKernelGGL((node_kernel_f2<L, SPLIT, HEAD, VAR>), dim3(static_cast<unsigned>(grid)), dim3(64 * node_waves<L>()), lds, s, a, ntiles, resident);
#ifdef AMBER_STAMPS
  {
    (void)hipStreamSynchronize(s);
    unsigned long long got[kStamps + 1] = {};
    (void)hipMemcpyFromSymbol(got, HIP_SYMBOL(g_nstamps), sizeof(got));
    static const char* names[kStamps] = {"loop-top", "wait-x", "barriers", "W0x", "wait-agg", "W0a", "bias-act", "hidden", "res-LN", "store-x", "W0s+store", "W0d+store"};
    double tot = 0;
    const double nt = static_cast<double>(got[kStamps] ? got[kStamps] : 1);
    for (int k = 0; k < kStamps; ++k) tot += static_cast<double>(got[k]);
    fprintf(stderr, "[node stamps L=%d HEAD=%d upd=%d] wave-batches %llu, cycles/batch %.0f:", L, int(HEAD), a.do_update, got[kStamps], tot / nt);
    for (int k = 0; k < kStamps; ++k) fprintf(stderr, " %s %.0f", names[k], static_cast<double>(got[k]) / nt);
    fprintf(stderr, "\n");
  }
#endif
  return hipGetLastError();
}

template <int L, int VAR>
static hipError_t launch_node_f2_v(const NodeArgs& a, int num_cus, hipStream_t s) {
  const bool head = a.do_readout != 0;  // the caller (run_node) clears the head fields unless it rides on this launch
  if (a.single) return head ? launch_node_f2_t<L, false, true, VAR>(a, num_cus, s) : launch_node_f2_t<L, false, false, VAR>(a, num_cus, s);
  return head ? launch_node_f2_t<L, true, true, VAR>(a, num_cus, s) : launch_node_f2_t<L, true, false, VAR>(a, num_cus, s);
}
#if defined(AMBER_F2_GENERAL_TU) && AMBER_F2_GENERAL_TU == 2
// (the node kernels are in the other general translation unit)
#elif defined(AMBER_F2_GENERAL_TU)
hipError_t launch_node_f2_general(int L, const NodeArgs& a, int num_cus, hipStream_t s) {
  switch (L) {
    case 32: return launch_node_f2_v<32, 1>(a, num_cus, s);
    case 64: return launch_node_f2_v<64, 1>(a, num_cus, s);
    case 128: return launch_node_f2_v<128, 1>(a, num_cus, s);
    default: return hipErrorInvalidValue;
  }
}
#else
hipError_t launch_node_f2(int L, const NodeArgs& a, int num_cus, hipStream_t s) {
  // (the node kernel's general instantiation costs nothing measurable: one variant serves tanh and the outer order)
  const bool general = (a.do_update && f2_variant(a.act, a.residual, a.layer_norm) != 0) || (a.do_readout && a.dense_act == AMBER_ACT_TANH);
  if (general) return launch_node_f2_general(L, a, num_cus, s);
  switch (L) {
    case 32: return launch_node_f2_v<32, 0>(a, num_cus, s);
    case 64: return launch_node_f2_v<64, 0>(a, num_cus, s);
    case 128: return launch_node_f2_v<128, 0>(a, num_cus, s);
    default: return hipErrorInvalidValue;
  }
}
#endif

}  // namespace amber
