// Graph construction on the device: the step BEFORE the message-passing path.
// For a batch of G meshes given as undirected neighbour pairs (local ids) + positions, writes in one pass
//   edge_index [2, E]  : per graph [a -> b | b -> a | v -> v], node ids offset per graph
//   edge_attr  [E, F]  : distance_vector (dim columns), euclidean_distance, edge_curvature, in that column order
//   batch [N], degree [N]
// reference: get_mesh_vertex_edges / get_mesh_element_edges (src/mesh_util/transforms/mesh_to_graph.py:58-145), the
// "degree" vertex feature (:224-225, np.unique(mesh_edges, return_counts=True)[1]), and the collation of
// torch_geometric's Batch.from_data_list reached through make_batch (src/helpers/torch_util.py:34-55): graphs
// concatenated in list order, edge_index offset by the number of nodes before the graph, batch = graph id.
// Integer / byte work, HBM-bound: 16 B read + (32 + 8 F) B written per undirected pair; (16 + 4 F) + 12 B per node.
#include "mpn_internal.h"

namespace amber {
namespace {

// graph that owns global item i: the last g with ptr[g] <= i  (ptr ascending, ptr[0] = 0, ptr[G] = total)
__device__ __forceinline__ int owner(const int64_t* __restrict__ ptr, int G, int64_t i) {
  int lo = 0, hi = G;  // invariant: ptr[lo] <= i < ptr[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ptr[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

// one edge row of features; d = position(destination) - position(source)
__device__ __forceinline__ void write_features(const GraphArgs& a, float* __restrict__ row, float dx, float dy, float dz,
                                               float norm, float curv) {
  int c = 0;
  if (a.distance_vector) {
    row[c++] = dx;
    row[c++] = dy;
    if (a.dim == 3) row[c++] = dz;
  }
  if (a.euclidean_distance) row[c++] = norm;
  if (a.edge_curvature) row[c++] = curv;
}

__global__ __launch_bounds__(256) void graph_pairs_kernel(const GraphArgs a) {
  for (int64_t j = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; j < a.M;
       j += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int g = owner(a.edge_ptr, a.G, j);
    const int64_t e0 = a.edge_ptr[g], Mg = a.edge_ptr[g + 1] - e0;
    const int64_t n0 = a.node_ptr[g], ng = a.node_ptr[g + 1] - n0;
    const int64_t base = 2 * e0 + (a.add_self_edges ? n0 : 0);
    int64_t va = a.nbr_a[j], vb = a.nbr_b[j];
    if (va < 0 || va >= ng || vb < 0 || vb >= ng) {  // report, clamp, go on (no out-of-bounds access)
      *a.status = 1;
      va = va < 0 ? 0 : (va >= ng ? ng - 1 : va);
      vb = vb < 0 ? 0 : (vb >= ng ? ng - 1 : vb);
    }
    const int64_t ga = va + n0, gb = vb + n0;
    const int64_t fwd = a.eoff + base + (j - e0), rev = fwd + Mg;
    a.edge_index[fwd] = ga;
    a.edge_index[a.E + fwd] = gb;
    a.edge_index[rev] = gb;
    a.edge_index[a.E + rev] = ga;
    if (a.degree_i) {
      atomicAdd(a.degree_i + ga, 1);  // integer atomics: the result does not depend on the order
      atomicAdd(a.degree_i + gb, 1);
    }
    if (a.F > 0) {
      float dx = 0.f, dy = 0.f, dz = 0.f, norm = 0.f;
      if (a.distance_vector || a.euclidean_distance) {
        const float* pa = a.positions + ga * a.dim;
        const float* pb = a.positions + gb * a.dim;
        dx = pb[0] - pa[0];
        dy = pb[1] - pa[1];
        float ss = __fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy));  // separately rounded, in column order
        if (a.dim == 3) {
          dz = pb[2] - pa[2];
          ss = __fadd_rn(ss, __fmul_rn(dz, dz));
        }
        norm = sqrtf(ss);
      }
      const float curv = a.curvature ? a.curvature[j] : 0.f;
      write_features(a, a.edge_attr + fwd * a.F, dx, dy, dz, norm, curv);
      write_features(a, a.edge_attr + rev * a.F, -dx, -dy, -dz, norm, curv);  // a - b == -(b - a) exactly
    }
  }
}

__global__ __launch_bounds__(256) void graph_nodes_kernel(const GraphArgs a) {
  for (int64_t v = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; v < a.N;
       v += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int g = owner(a.node_ptr, a.G, v);
    if (a.batch) a.batch[v] = g;
    if (a.add_self_edges) {
      const int64_t e1 = a.edge_ptr[g + 1], n0 = a.node_ptr[g];
      const int64_t p = a.eoff + 2 * e1 + n0 + (v - n0);  // behind both directions of this graph's pairs
      a.edge_index[p] = v;
      a.edge_index[a.E + p] = v;
      if (a.F > 0) write_features(a, a.edge_attr + p * a.F, 0.f, 0.f, 0.f, 0.f, 0.f);
    }
  }
}

// Inter-level edges of AMBER's hierarchical observation graph (get_inter_graph_edges, mesh_to_graph.py:239-301): node k
// of the source (current) mesh <-> its partner nearest[k] of the destination (initial) mesh, both directions:
//   edge k          : k -> nearest[k] + offset,   features from  pos_src[k] - pos_dst[nearest[k]]
//   edge n_src + k  : nearest[k] + offset -> k,   the negated vector, the same distance, curvature 0
// The reference computes these features in float64 numpy (positions as scikit-fem holds them) and casts to float32 at the
// end; so does this kernel (separately rounded multiply / add, as numpy's sqrt(add.reduce(x * x))).
__global__ __launch_bounds__(256) void graph_inter_kernel(const GraphArgs a, const int64_t* __restrict__ nearest,
                                                          const double* __restrict__ pos_src, const double* __restrict__ pos_dst,
                                                          int64_t n_src, int64_t n_dst, int64_t index_offset) {
  for (int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; k < n_src;
       k += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    int64_t q = nearest[k];
    if (q < 0 || q >= n_dst) {
      *a.status = 1;
      q = q < 0 ? 0 : n_dst - 1;
    }
    const int64_t fwd = a.eoff + k, rev = fwd + n_src;
    a.edge_index[fwd] = k;
    a.edge_index[a.E + fwd] = q + index_offset;
    a.edge_index[rev] = q + index_offset;
    a.edge_index[a.E + rev] = k;
    if (a.F > 0) {
      const double* ps = pos_src + k * a.dim;
      const double* pd = pos_dst + q * a.dim;
      const double dx = ps[0] - pd[0], dy = ps[1] - pd[1], dz = a.dim == 3 ? ps[2] - pd[2] : 0.0;
      double ss = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
      if (a.dim == 3) ss = __dadd_rn(ss, __dmul_rn(dz, dz));
      const float norm = static_cast<float>(sqrt(ss));
      write_features(a, a.edge_attr + fwd * a.F, static_cast<float>(dx), static_cast<float>(dy), static_cast<float>(dz), norm, 0.f);
      write_features(a, a.edge_attr + rev * a.F, static_cast<float>(-dx), static_cast<float>(-dy), static_cast<float>(-dz), norm, 0.f);
    }
  }
}

// occurrence counts (int32, accumulated in place) -> float32 feature column
__global__ __launch_bounds__(256) void graph_degree_kernel(int32_t* __restrict__ deg, int64_t N) {
  for (int64_t v = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; v < N;
       v += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const float f = static_cast<float>(deg[v]);
    reinterpret_cast<float*>(deg)[v] = f;
  }
}

unsigned grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 256 * 32) g = 256 * 32;
  if (g < 1) g = 1;
  return static_cast<unsigned>(g);
}

}  // namespace

hipError_t launch_inter_edges(const GraphArgs& a, const int64_t* nearest, const double* pos_src, const double* pos_dst, int64_t n_src,
                              int64_t n_dst, int64_t index_offset, hipStream_t s) {
  hipError_t e = hipMemsetAsync(a.status, 0, sizeof(int32_t), s);
  if (e != hipSuccess) return e;
  if (n_src > 0) hipLaunchKernelGGL(graph_inter_kernel, dim3(grid_for(n_src)), dim3(256), 0, s, a, nearest, pos_src, pos_dst, n_src, n_dst, index_offset);
  return hipGetLastError();
}

hipError_t launch_build_graph(const GraphArgs& a, hipStream_t s) {
  hipError_t e = hipMemsetAsync(a.status, 0, sizeof(int32_t), s);
  if (e != hipSuccess) return e;
  if (a.degree_i && a.N > 0) {
    e = hipMemsetAsync(a.degree_i, 0, sizeof(int32_t) * static_cast<size_t>(a.N), s);
    if (e != hipSuccess) return e;
  }
  if (a.M > 0) hipLaunchKernelGGL(graph_pairs_kernel, dim3(grid_for(a.M)), dim3(256), 0, s, a);
  if (a.N > 0 && (a.batch || a.add_self_edges)) hipLaunchKernelGGL(graph_nodes_kernel, dim3(grid_for(a.N)), dim3(256), 0, s, a);
  if (a.degree_i && a.N > 0) hipLaunchKernelGGL(graph_degree_kernel, dim3(grid_for(a.N)), dim3(256), 0, s, a.degree_i, a.N);
  return hipGetLastError();
}

}  // namespace amber
