// Split-precision ("f16x2") weight packing for the row-per-lane chain of the fast node kernel (mpn_kernels.hip):
// hi/lo fp16 in A-operand order with the k permutation of the accumulator layout.  The numerics of the scheme are
// described at the top of mpn_f2.hip.
#include "mpn_tile.h"

namespace amber {

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

}  // namespace amber
