// The step AFTER the model in AMBER's inference step: element -> vertex projection of a
// sizing field and the heuristic element-count estimate of the new mesh, in float64 as the reference computes them.
// reference: /root/reference/src/mesh_util/sizing_field_util.py:43-70 (project_elements_to_vertices), :128-200
// (_mesh_sizing_field_to_num_elements).  Both are HBM / latency bound integer-and-float64 row work; the projection
// reuses the destination-sorted CSR builder (csr.hip) on the element-vertex incidence list, so that every vertex adds
// its entries in the order torch_scatter's scatter_add visits them: bit-identical sums.
#include "mpn_internal.h"

// every float64 product and sum here rounds on its own, as numpy / torch on the CPU do: hipcc contracts a * b + c into a
// fused multiply-add by default (the HIP __dmul_rn / __dadd_rn included: plain operators that carry the contract flag of
// the header they were compiled in), so this file is built with -ffp-contract=off (Makefile) and says so here as well
#pragma clang fp contract(off)

namespace amber {
namespace {

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline unsigned grid_for(int64_t items, int block) {
  int64_t g = cdiv(items, block);
  if (g > 256 * 32) g = 256 * 32;
  if (g < 1) g = 1;
  return static_cast<unsigned>(g);
}

// incidence list in the form csr_build takes: entry i = m k + j (mesh.t.T.flatten()[i]) is vertex t[j][m]
__global__ void incidence_kernel(const int64_t* __restrict__ t, int k, int64_t M, int64_t* __restrict__ ei) {
  const int64_t total = M * k;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t m = i / k;
    const int j = static_cast<int>(i - m * k);
    const int64_t v = t[static_cast<int64_t>(j) * M + m];
    ei[i] = v;          // "source" row: unused, kept inside [0, V) like the destination
    ei[total + i] = v;  // destination row: the vertex
  }
}

// out[v] = sum_i val[m(i)] * vol[m(i)] / sum_i vol[m(i)] over the entries of vertex v in entry order; the product is
// rounded before it is added (no fused multiply-add: the reference multiplies the arrays first, then scatter-adds)
__global__ void project_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ perm, int k, int64_t V,
                               const double* __restrict__ val, const double* __restrict__ vol, double* __restrict__ out) {
  for (int64_t v = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; v < V;
       v += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    double s = 0.0, w = 0.0;
    for (int p = rowptr[v]; p < rowptr[v + 1]; ++p) {
      const int64_t m = perm[p] / k;
      const double wm = vol[m];
      const double prod = val[m] * wm;
      s = s + prod;
      w = w + wm;
    }
    out[v] = s / w;  // a vertex without elements: 0 / 0 = nan, as in the reference
  }
}

__device__ inline void atomic_max_f64(double* addr, double v) {
  unsigned long long* a = reinterpret_cast<unsigned long long*>(addr);
  if (v != v) {  // np.maximum propagates a nan candidate (and keeps a nan that is already there: `nan < v` is false below)
    atomicExch(a, static_cast<unsigned long long>(__double_as_longlong(v)));
    return;
  }
  unsigned long long old = *a;
  while (__longlong_as_double(static_cast<long long>(old)) < v) {  // np.maximum: keeps the larger; order-independent
    const unsigned long long seen = atomicCAS(a, old, static_cast<unsigned long long>(__double_as_longlong(v)));
    if (seen == old) break;
    old = seen;
  }
}

// one gradation pass, first half: the lower bounds both ends of every edge impose on each other, from the field as it
// stands BEFORE the pass (sizing_field_util.py:172-181)
// a vertex id outside [0, V) is clamped (memory-safe) and reported through `flag` (the reference's numpy indexing raises)
__device__ inline int64_t checked_vertex(int64_t v, int64_t V, int32_t* flag) {
  if (v < 0 || v >= V) {
    *flag = 1;  // every writer stores the same value
    return v < 0 ? 0 : V - 1;
  }
  return v;
}
__global__ void gradation_bounds_kernel(const int64_t* __restrict__ edges, int64_t Me, int64_t V, const double* __restrict__ pos, int dim,
                                        const double* __restrict__ field, double gradation, double* __restrict__ m0,
                                        double* __restrict__ m1, int32_t* __restrict__ flag) {
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < Me;
       e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t a = checked_vertex(edges[e], V, flag), b = checked_vertex(edges[Me + e], V, flag);
    double d2 = 0.0;
    for (int c = 0; c < dim; ++c) {
      const double d = pos[a * dim + c] - pos[b * dim + c];
      const double sq = d * d;
      d2 = d2 + sq;
    }
    const double dist = sqrt(d2);
    const double s0 = field[a], s1 = field[b];
    m1[e] = s0 * pow(gradation, -(dist / s0));
    m0[e] = s1 * pow(gradation, -(dist / s1));
  }
}
// second half: np.maximum.at on both ends (the maximum does not depend on the order of the updates)
__global__ void gradation_apply_kernel(const int64_t* __restrict__ edges, int64_t Me, int64_t V, const double* __restrict__ m0,
                                       const double* __restrict__ m1, double* __restrict__ field, int32_t* __restrict__ flag) {
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < Me;
       e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    atomic_max_f64(field + checked_vertex(edges[e], V, flag), m0[e]);
    atomic_max_f64(field + checked_vertex(edges[Me + e], V, flag), m1[e]);
  }
}

__device__ inline double edge_length_to_volume(double s, int dim) {  // geometry_util.py:41-44
  return dim == 2 ? s * s * sqrt(3.0) / 4.0 : s * s * s * sqrt(2.0) / 12.0;
}

// per-block partial sums of volume / predicted volume, in element order inside a block; blocks are added in order by
// density_final_kernel (deterministic; numpy's pairwise sum differs from it in the last bits only)
constexpr int kDensityBlock = 256;
__global__ __launch_bounds__(kDensityBlock) void density_partial_kernel(const int64_t* __restrict__ t, int k, int64_t M, int64_t V,
                                                                        const double* __restrict__ vol,
                                                                        const double* __restrict__ field, int vertex_field, int dim,
                                                                        double* __restrict__ part, int32_t* __restrict__ flag) {
  __shared__ double red[kDensityBlock];
  double s = 0.0;
  for (int64_t m = static_cast<int64_t>(blockIdx.x) * kDensityBlock + threadIdx.x; m < M;
       m += static_cast<int64_t>(gridDim.x) * kDensityBlock) {
    double sz;
    if (vertex_field) {  // mean of the simplex' vertex values (:187)
      double acc = 0.0;
      for (int j = 0; j < k; ++j) acc += field[checked_vertex(t[static_cast<int64_t>(j) * M + m], V, flag)];
      sz = acc / k;
    } else {
      sz = field[m];
    }
    s += vol[m] / edge_length_to_volume(sz, dim);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = kDensityBlock / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void density_final_kernel(const double* __restrict__ part, int n, double factor, const int32_t* __restrict__ flag,
                                     double* __restrict__ out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += part[i];
    out[0] = s * factor;
    out[1] = flag[0] ? 1.0 : 0.0;  // a vertex id was out of range (clamped): the caller raises, as numpy's indexing would
  }
}

inline size_t align256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }
constexpr int kDensityGrid = 1024;

}  // namespace

size_t project_workspace_bytes(int k, int64_t M, int64_t V) {
  const size_t n = static_cast<size_t>(k) * M;
  return align256(16 * n) + align256(4 * (static_cast<size_t>(V) + 1)) + 3 * align256(4 * n) + 256 + csr_workspace_bytes(V, k * M);
}

hipError_t launch_project_elements(const int64_t* t, int k, int64_t M, int64_t V, const double* val, const double* vol, double* out,
                                   int32_t* status, void* ws, hipStream_t s) {
  const size_t n = static_cast<size_t>(k) * M;
  char* p = static_cast<char*>(ws);
  int64_t* ei = reinterpret_cast<int64_t*>(p);
  p += align256(16 * n);
  int32_t* rowptr = reinterpret_cast<int32_t*>(p);
  p += align256(4 * (static_cast<size_t>(V) + 1));
  int32_t* src_sorted = reinterpret_cast<int32_t*>(p);
  p += align256(4 * n);
  int32_t* dst_sorted = reinterpret_cast<int32_t*>(p);
  p += align256(4 * n);
  int32_t* perm = reinterpret_cast<int32_t*>(p);
  p += align256(4 * n);
  if (n > 0) hipLaunchKernelGGL(incidence_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, t, k, M, ei);
  hipError_t e = csr_build(ei, static_cast<int64_t>(n), V, rowptr, src_sorted, dst_sorted, perm, status, p, s);
  if (e != hipSuccess) return e;
  if (V > 0) hipLaunchKernelGGL(project_kernel, dim3(grid_for(V, 256)), dim3(256), 0, s, rowptr, perm, k, V, val, vol, out);
  return hipGetLastError();
}

size_t estimate_workspace_bytes(int64_t V, int64_t Me) {
  return align256(8 * static_cast<size_t>(V)) + 2 * align256(8 * static_cast<size_t>(Me)) + align256(8 * kDensityGrid) + 256;
}

hipError_t launch_estimate_elements(const int64_t* t, int k, int64_t M, int64_t V, const double* vol, const double* field,
                                    int vertex_field, int dim, const int64_t* edges, int64_t Me, const double* pos,
                                    double gradation, double* out, void* ws, hipStream_t s) {
  char* p = static_cast<char*>(ws);
  double* smooth = reinterpret_cast<double*>(p);
  p += align256(8 * static_cast<size_t>(V));
  double* m0 = reinterpret_cast<double*>(p);
  p += align256(8 * static_cast<size_t>(Me));
  double* m1 = reinterpret_cast<double*>(p);
  p += align256(8 * static_cast<size_t>(Me));
  double* part = reinterpret_cast<double*>(p);
  p += align256(8 * kDensityGrid);
  int32_t* flag = reinterpret_cast<int32_t*>(p);
  hipError_t e0 = hipMemsetAsync(flag, 0, sizeof(int32_t), s);
  if (e0 != hipSuccess) return e0;
  const double* use = field;
  if (vertex_field) {
    hipError_t e = hipMemcpyAsync(smooth, field, 8 * static_cast<size_t>(V), hipMemcpyDeviceToDevice, s);  // sizing_field.copy()
    if (e != hipSuccess) return e;
    for (int pass = 0; pass < 2 && Me > 0; ++pass) {
      hipLaunchKernelGGL(gradation_bounds_kernel, dim3(grid_for(Me, 256)), dim3(256), 0, s, edges, Me, V, pos, dim, smooth, gradation, m0, m1, flag);
      hipLaunchKernelGGL(gradation_apply_kernel, dim3(grid_for(Me, 256)), dim3(256), 0, s, edges, Me, V, m0, m1, smooth, flag);
    }
    use = smooth;
  }
  int grid = static_cast<int>(cdiv(M > 0 ? M : 1, kDensityBlock));
  if (grid > kDensityGrid) grid = kDensityGrid;
  hipLaunchKernelGGL(density_partial_kernel, dim3(grid), dim3(kDensityBlock), 0, s, t, k, M, V, vol, use, vertex_field, dim, part, flag);
  const double factor = (vertex_field && dim == 3) ? 0.618 : 1.0;  // :192-194
  hipLaunchKernelGGL(density_final_kernel, dim3(1), dim3(64), 0, s, part, grid, factor, flag, out);
  return hipGetLastError();
}

}  // namespace amber
