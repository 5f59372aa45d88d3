// Pre / post numeric glue of MeshGenerationAlgorithm._predict as stand-alone kernels.
// The forward entry point amber_mpn_predict applies the post-processing inside the head kernel's readout epilogue
// (mpn_kernels.hip); these kernels serve hosts that drive the stages themselves, training (the backward of the
// post-processing) and the per-stage parity tests.  All of it is elementwise and HBM-bound: 4 bytes in, 4 bytes out a value.
#include "mpn_post.h"

namespace amber {

// RunningNormalizer._normalize (src/algorithm/normalizer/running_normalizer.py:87-106): one thread per value; the
// per-feature statistics (float64) are read through the scalar / L1 path
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ in, int64_t total, int width,
                                                             const double* __restrict__ mean, const double* __restrict__ var,
                                                             double eps, double clip, float* __restrict__ out) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    const int k = static_cast<int>(i % width);
    out[i] = normalize_one(in[i], mean[k], var[k], eps, clip);
  }
}

__global__ __launch_bounds__(256) void postprocess_kernel(const PostArgs p, const float* __restrict__ pred, int64_t n,
                                                          float* __restrict__ out) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    out[i] = post_one(p, pred[i], i);
}

// d out / d pred of post_one, times grad_out: std * transform'(u) (clip and damping belong to the detached inference
// path, mesh_generation_algorithm.py:369-371, and are not differentiated: requesting them here is rejected by the host)
__global__ __launch_bounds__(256) void postprocess_bwd_kernel(const PostArgs p, const float* __restrict__ pred,
                                                              const float* __restrict__ grad_out, int64_t n,
                                                              float* __restrict__ grad_pred) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    float u = 0.f;
    const float y = post_one(p, pred[i], i, &u);
    float g = grad_out[i];
    if (p.apply_transform) {
      if (p.transform == AMBER_TRANSFORM_SOFTPLUS) {
        const float z = expf(u);
        g = u > 20.f ? g : g * z / (z + 1.f);  // torch's softplus backward
      } else if (p.transform == AMBER_TRANSFORM_EXP)
        g = g * y;
    }
    if (p.denorm) g = static_cast<float>(static_cast<double>(g) * sqrt(p.var[0] + p.eps));
    grad_pred[i] = g;
  }
}

__device__ __forceinline__ float inverse_one(int transform, float v) {
  return transform == AMBER_TRANSFORM_SOFTPLUS ? softplus_inverse(v) : (transform == AMBER_TRANSFORM_EXP ? logf(v) : v);
}
__global__ __launch_bounds__(256) void inverse_transform_kernel(int transform, const float* __restrict__ in,
                                                                const float* __restrict__ baseline, int64_t n,
                                                                float* __restrict__ out) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    float v = inverse_one(transform, in[i]);
    if (baseline) v = v - inverse_one(transform, baseline[i]);  // residual prediction: prediction_transform.py:121-124
    out[i] = v;
  }
}

static inline unsigned grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 256 * 32) g = 256 * 32;
  if (g < 1) g = 1;
  return static_cast<unsigned>(g);
}

hipError_t launch_normalize_rows(const float* in, int64_t rows, int width, const double* mean, const double* var, double eps,
                                 double clip, float* out, hipStream_t s) {
  const int64_t total = rows * width;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(normalize_rows_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, total, width, mean, var, eps, clip, out);
  return hipGetLastError();
}
hipError_t launch_inverse_transform(int transform, const float* in, const float* baseline, int64_t n, float* out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(inverse_transform_kernel, dim3(grid_for(n)), dim3(256), 0, s, transform, in, baseline, n, out);
  return hipGetLastError();
}
hipError_t launch_postprocess(const PostArgs& p, const float* pred, int64_t n, float* out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(postprocess_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, pred, n, out);
  return hipGetLastError();
}
hipError_t launch_postprocess_bwd(const PostArgs& p, const float* pred, const float* grad_out, int64_t n, float* grad_pred,
                                  hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(postprocess_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, pred, grad_out, n, grad_pred);
  return hipGetLastError();
}

}  // namespace amber
