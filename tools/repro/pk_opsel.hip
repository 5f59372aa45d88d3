// Stand-alone look at the instruction form csrc/pkfix.py rewrites:
//   v_pk_add_f32 D, A, B op_sel:[0,1] op_sel_hi:[1,0]      D.lo = A.lo + B.hi,  D.hi = A.hi + B.lo
// against the same sums with the selection on src0 (operands exchanged) and with scalar adds, on freshly loaded operands,
// many waves per SIMD.  Prints how many low / high results differ from the scalar ones per form.
// build: hipcc -O2 --offload-arch=gfx950 tools/repro/pk_opsel.hip -o tools/repro/pk_opsel ; run: tools/repro/pk_opsel [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void probe(const f2* __restrict__ a, const f2* __restrict__ b, int n, int iters, unsigned* bad, int mfma_neighbours) {
  unsigned bad_src1_lo = 0, bad_src1_hi = 0, bad_src0_lo = 0, bad_src0_hi = 0;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  if (mfma_neighbours && ((threadIdx.x >> 6) & 1)) {
    // every other wave keeps the matrix pipe and the LDS busy instead (the kernels' waves share their SIMD with such waves)
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef float f16v __attribute__((ext_vector_type(16)));
    __shared__ float lds[4096];
    f16v c = {};
    h8 x8, y8;
    for (int j = 0; j < 8; ++j) { x8[j] = static_cast<_Float16>(threadIdx.x * 0.001f + j); y8[j] = static_cast<_Float16>(0.5f - j * 0.01f); }
    for (int it = 0; it < iters * 64; ++it) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(x8, y8, c, 0, 0, 0);
      lds[(threadIdx.x * 5 + it) & 4095] = c[it & 15];
      x8[0] = static_cast<_Float16>(lds[(threadIdx.x * 3 + it) & 4095] * 1e-3f);
    }
    if (c[0] == 12345.678f) bad[7] = 1;
    return;
  }
  f2 acc = {0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    for (int i = tid; i < n; i += stride) {
      const f2 x = a[i];
      f2 y = b[(i + it) % n];
      y += acc;  // a VALU-produced operand, as in the kernels
      f2 r1, r0;
      asm volatile("s_waitcnt vmcnt(0)\n\tv_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r1) : "v"(x), "v"(y));
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r0) : "v"(y), "v"(x));
      const float lo = x.x + y.y, hi = x.y + y.x;
      bad_src1_lo += r1.x != lo;
      bad_src1_hi += r1.y != hi;
      bad_src0_lo += r0.x != lo;
      bad_src0_hi += r0.y != hi;
      acc.x = 0.25f * lo;
      acc.y = 0.25f * hi;
    }
  }
  if (bad_src1_lo) atomicAdd(&bad[0], bad_src1_lo);
  if (bad_src1_hi) atomicAdd(&bad[1], bad_src1_hi);
  if (bad_src0_lo) atomicAdd(&bad[2], bad_src0_lo);
  if (bad_src0_hi) atomicAdd(&bad[3], bad_src0_hi);
  if (bad_src1_lo && (threadIdx.x & 63) >= 32) atomicAdd(&bad[4], bad_src1_lo);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  const int n = 1 << 22;
  std::vector<f2> ha(n), hb(n);
  for (int i = 0; i < n; ++i) {
    ha[i] = f2{static_cast<float>((i * 2654435761u) >> 8) * 1e-6f, static_cast<float>((i * 40503u) & 0xffff) * 1e-3f};
    hb[i] = f2{static_cast<float>((i * 2246822519u) >> 9) * 1e-6f, static_cast<float>((i * 3266489917u) >> 10) * 1e-5f};
  }
  f2 *a, *b;
  unsigned* bad;
  if (hipMalloc(&a, n * sizeof(f2)) != hipSuccess || hipMalloc(&b, n * sizeof(f2)) != hipSuccess || hipMalloc(&bad, 8 * sizeof(unsigned)) != hipSuccess) return 2;
  hipMemcpy(a, ha.data(), n * sizeof(f2), hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), n * sizeof(f2), hipMemcpyHostToDevice);
  hipMemset(bad, 0, 8 * sizeof(unsigned));
  for (int neighbours = 0; neighbours < 2; ++neighbours) {
    hipMemset(bad, 0, 8 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(256 * 2), dim3(512), 0, 0, a, b, n, iters, bad, neighbours);
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    unsigned h[8];
    hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
    printf("%s: %d x %d packed adds per form; wrong results: selection on src1 low %u (lanes 32-63: %u) high %u | selection on src0 low %u high %u\n",
           neighbours ? "beside MFMA / LDS waves" : "alone", iters, n, h[0], h[4], h[1], h[2], h[3]);
  }
  return 0;
}
