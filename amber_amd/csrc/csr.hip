// Destination-sorted CSR of a torch edge_index[2, E] (int64), built on the device with integer work only.
//
//   1. count   : in-degree histogram with integer atomics (integer adds commute: result is deterministic)
//   2. scan    : exclusive prefix sum -> rowptr (three-kernel block scan)
//   3. fill    : every edge takes a slot in its destination's row through an atomic cursor (slot order is
//                arbitrary at this point)
//   4. sort    : every row's edge ids are sorted ascending, which makes the result EQUAL to a stable
//                counting sort by destination (bit-exact vs torch.sort(dst, stable=True)), i.e. inside a
//                row the edges keep the order in which the reference's scatter_add visits them
//                (src/mpn/message_passing_modules.py:67-68).  Short rows: one thread, insertion sort;
//                long rows (hub nodes of hierarchical graphs): one wave, rank counting.
//   5. gather  : src_sorted / dst_sorted
#include <mutex>

#include "mpn_internal.h"

// Scratch of the allocation-free entry point amber_mpn_csr_build_static (include/amber_mpn.h): the count | cursor |
// chunk_sum | slots areas that amber_mpn_csr_build takes from the caller's workspace.  One copy per GPU the library runs
// on, held until the process ends.  Host code never uses this name as an address: csr_static_scratch() asks the runtime
// for each device's copy once and hands that pointer on.
static __device__ __align__(256) unsigned char scratch[AMBER_MPN_CSR_STATIC_SCRATCH_BYTES];

namespace amber {

namespace {

constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;                       // per thread
constexpr int kScanChunk = kScanBlock * kScanItems; // per block
constexpr int kShortRow = 32;

__global__ void csr_count_kernel(const int64_t* __restrict__ ei, int64_t E, int64_t N, int32_t* __restrict__ count,
                                 int32_t* __restrict__ status) {
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < E;
       e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    int64_t s = ei[e], d = ei[E + e];
    if (s < 0 || s >= N || d < 0 || d >= N) {
      *status = 1;  // benign race: every writer stores the same value
      d = d < 0 ? 0 : (d >= N ? N - 1 : d);
    }
    atomicAdd(&count[d], 1);
  }
}

// block-level exclusive scan of `count` chunks; writes per-chunk totals
__global__ void scan_reduce_kernel(const int32_t* __restrict__ count, int64_t N, int32_t* __restrict__ chunk_sum) {
  __shared__ int32_t red[kScanBlock];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanChunk + static_cast<int64_t>(threadIdx.x) * kScanItems;
  int32_t s = 0;
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < N) s += count[base + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = kScanBlock / 2; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) chunk_sum[blockIdx.x] = red[0];
}

// single block: exclusive scan of the chunk totals in place (serial over groups of kScanBlock)
__global__ void scan_chunks_kernel(int32_t* __restrict__ chunk_sum, int64_t nchunks) {
  __shared__ int32_t buf[kScanBlock];
  __shared__ int32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t g = 0; g < nchunks; g += kScanBlock) {
    const int64_t i = g + threadIdx.x;
    const int32_t v = i < nchunks ? chunk_sum[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {  // Hillis-Steele inclusive scan
      const int32_t t = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nchunks) chunk_sum[i] = carry + buf[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 0) carry += buf[kScanBlock - 1];
    __syncthreads();
  }
}

__global__ void scan_apply_kernel(const int32_t* __restrict__ count, int64_t N, const int32_t* __restrict__ chunk_off,
                                  int32_t* __restrict__ rowptr, int32_t total_edges) {
  __shared__ int32_t buf[kScanBlock];
  const int64_t base = static_cast<int64_t>(blockIdx.x) * kScanChunk + static_cast<int64_t>(threadIdx.x) * kScanItems;
  int32_t local[kScanItems];
  int32_t s = 0;
  for (int i = 0; i < kScanItems; ++i) {
    local[i] = base + i < N ? count[base + i] : 0;
    s += local[i];
  }
  buf[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < kScanBlock; off <<= 1) {
    const int32_t t = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
    __syncthreads();
    buf[threadIdx.x] += t;
    __syncthreads();
  }
  int32_t run = chunk_off[blockIdx.x] + buf[threadIdx.x] - s;
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < N) rowptr[base + i] = run;
    run += local[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) rowptr[N] = total_edges;
}

__global__ void csr_fill_kernel(const int64_t* __restrict__ ei, int64_t E, int64_t N, const int32_t* __restrict__ rowptr,
                                int32_t* __restrict__ cursor, int32_t* __restrict__ slots) {
  for (int64_t e = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < E;
       e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    int64_t d = ei[E + e];
    d = d < 0 ? 0 : (d >= N ? N - 1 : d);
    const int32_t k = atomicAdd(&cursor[d], 1);
    slots[rowptr[d] + k] = static_cast<int32_t>(e);
  }
}

// rows with at most kShortRow entries: one thread sorts its row (insertion sort on a private copy)
__global__ void csr_sort_short_kernel(const int32_t* __restrict__ rowptr, int64_t N, const int32_t* __restrict__ slots,
                                      int32_t* __restrict__ perm) {
  for (int64_t n = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; n < N;
       n += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int r0 = rowptr[n], m = rowptr[n + 1] - r0;
    if (m > kShortRow) continue;
    int32_t v[kShortRow];
    for (int i = 0; i < m; ++i) v[i] = slots[r0 + i];
    for (int i = 1; i < m; ++i) {
      const int32_t key = v[i];
      int j = i - 1;
      while (j >= 0 && v[j] > key) {
        v[j + 1] = v[j];
        --j;
      }
      v[j + 1] = key;
    }
    for (int i = 0; i < m; ++i) perm[r0 + i] = v[i];
  }
}

// rows longer than kShortRow: one wave per row, rank of an id = number of smaller ids in the row (ids are unique)
__global__ void csr_sort_long_kernel(const int32_t* __restrict__ rowptr, int64_t N, const int32_t* __restrict__ slots,
                                     int32_t* __restrict__ perm) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  for (int64_t n = wave; n < N; n += nwaves) {
    const int r0 = rowptr[n], m = rowptr[n + 1] - r0;
    if (m <= kShortRow) continue;
    for (int i = lane; i < m; i += 64) {
      const int32_t key = slots[r0 + i];
      int rank = 0;
      for (int k = 0; k < m; ++k) rank += slots[r0 + k] < key ? 1 : 0;
      perm[r0 + rank] = key;
    }
  }
}

__global__ void csr_gather_kernel(const int64_t* __restrict__ ei, int64_t E, int64_t N, const int32_t* __restrict__ rowptr,
                                  const int32_t* __restrict__ perm, int32_t* __restrict__ src_sorted,
                                  int32_t* __restrict__ dst_sorted) {
  for (int64_t p = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; p < E;
       p += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int64_t e = perm[p];
    int64_t s = ei[e], d = ei[E + e];
    s = s < 0 ? 0 : (s >= N ? N - 1 : s);
    d = d < 0 ? 0 : (d >= N ? N - 1 : d);
    src_sorted[p] = static_cast<int32_t>(s);
    dst_sorted[p] = static_cast<int32_t>(d);
  }
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline size_t align256(size_t x) { return (x + 255) & ~static_cast<size_t>(255); }
inline unsigned grid_for(int64_t items, int block) {
  int64_t g = cdiv(items, block);
  if (g > 256 * 32) g = 256 * 32;
  if (g < 1) g = 1;
  return static_cast<unsigned>(g);
}

}  // namespace

// workspace: count[N] | cursor[N] | chunk_sum[nchunks] | slots[E]
size_t csr_workspace_bytes(int64_t N, int64_t E) {
  const int64_t nchunks = cdiv(N > 0 ? N : 1, kScanChunk);
  return align256(sizeof(int32_t) * static_cast<size_t>(N)) * 2 + align256(sizeof(int32_t) * static_cast<size_t>(nchunks)) +
         align256(sizeof(int32_t) * static_cast<size_t>(E));
}

// Device address of this GPU's copy of `scratch` (looked up once per device, with that device current).
hipError_t csr_static_scratch(void** out) {
  constexpr int kMaxDevices = 64;
  static std::mutex mu;
  static void* ptr_of[kMaxDevices] = {};
  *out = nullptr;
  int dev = 0;
  hipError_t err = hipGetDevice(&dev);
  if (err != hipSuccess) return err;
  if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(mu);
  if (!ptr_of[dev]) {
    void* p = nullptr;
    err = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (err != hipSuccess) return err;
    if (!p) return hipErrorInvalidSymbol;
    ptr_of[dev] = p;
  }
  *out = ptr_of[dev];
  return hipSuccess;
}

hipError_t csr_build(const int64_t* edge_index, int64_t E, int64_t N, int32_t* rowptr, int32_t* src_sorted,
                     int32_t* dst_sorted, int32_t* perm, int32_t* status, void* ws, hipStream_t s) {
  char* base = static_cast<char*>(ws);
  const int64_t nchunks = cdiv(N > 0 ? N : 1, kScanChunk);
  int32_t* count = reinterpret_cast<int32_t*>(base);
  int32_t* cursor = reinterpret_cast<int32_t*>(base + align256(sizeof(int32_t) * static_cast<size_t>(N)));
  int32_t* chunk_sum = reinterpret_cast<int32_t*>(base + 2 * align256(sizeof(int32_t) * static_cast<size_t>(N)));
  int32_t* slots = reinterpret_cast<int32_t*>(base + 2 * align256(sizeof(int32_t) * static_cast<size_t>(N)) +
                                              align256(sizeof(int32_t) * static_cast<size_t>(nchunks)));
  hipError_t err;
  if ((err = hipMemsetAsync(status, 0, sizeof(int32_t), s)) != hipSuccess) return err;
  if (N > 0) {
    // count and cursor are adjacent (each padded to 256 B): one memset clears both
    if ((err = hipMemsetAsync(count, 0, 2 * align256(sizeof(int32_t) * static_cast<size_t>(N)), s)) != hipSuccess) return err;
  }
  if (E > 0 && N > 0)
    hipLaunchKernelGGL(csr_count_kernel, dim3(grid_for(E, 256)), dim3(256), 0, s, edge_index, E, N, count, status);
  if (N > 0) {
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(static_cast<unsigned>(nchunks)), dim3(kScanBlock), 0, s, count, N, chunk_sum);
    hipLaunchKernelGGL(scan_chunks_kernel, dim3(1), dim3(kScanBlock), 0, s, chunk_sum, nchunks);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(static_cast<unsigned>(nchunks)), dim3(kScanBlock), 0, s, count, N, chunk_sum,
                       rowptr, static_cast<int32_t>(E));
  } else {
    if ((err = hipMemsetAsync(rowptr, 0, sizeof(int32_t), s)) != hipSuccess) return err;
  }
  if (E > 0 && N > 0) {
    hipLaunchKernelGGL(csr_fill_kernel, dim3(grid_for(E, 256)), dim3(256), 0, s, edge_index, E, N, rowptr, cursor, slots);
    hipLaunchKernelGGL(csr_sort_short_kernel, dim3(grid_for(N, 256)), dim3(256), 0, s, rowptr, N, slots, perm);
    hipLaunchKernelGGL(csr_sort_long_kernel, dim3(grid_for(N * 64, 256)), dim3(256), 0, s, rowptr, N, slots, perm);
    hipLaunchKernelGGL(csr_gather_kernel, dim3(grid_for(E, 256)), dim3(256), 0, s, edge_index, E, N, rowptr, perm, src_sorted,
                       dst_sorted);
  }
  return hipGetLastError();
}

}  // namespace amber
