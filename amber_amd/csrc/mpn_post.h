// Pre / post numeric glue of MeshGenerationAlgorithm._predict, device side.
// reference: RunningNormalizer._normalize / _denormalize (src/algorithm/normalizer/running_normalizer.py:87-119),
//            PredictionTransform.forward (src/algorithm/prediction_transform/prediction_transform.py:58-88),
//            SoftplusTransform (softplus_transform.py:6-11), ExponentialTransform (exponential_transform.py:21-47),
//            _clip_detach (src/algorithm/core/mesh_generation_algorithm.py:369-371),
//            _clip_detach_dampen (src/algorithm/core/amber.py:213-246).
#pragma once
#include "mpn_internal.h"

namespace amber {

// RunningNormalizer._normalize for one value: the statistics are float64 buffers, so torch promotes the whole
// expression to float64 and casts the clipped result to float32 (`.float()`).  Every operation below is a single
// correctly rounded IEEE operation, as on the CPU: the result is bit-identical.
__device__ __forceinline__ float normalize_one(float x, double mean, double var, double eps, double clip) {
  double v = (static_cast<double>(x) - mean) / sqrt(var + eps);
  v = v < -clip ? -clip : v;  // torch.clip: NaN stays NaN (both comparisons false)
  v = v > clip ? clip : v;
  return static_cast<float>(v);
}

// softplus^-1(b) = b + log(-expm1(-b))   (softplus_transform.py:10-11), float32 like the reference's baseline tensor
__device__ __forceinline__ float softplus_inverse(float b) { return b + logf(-expm1f(-b)); }
// torch.nn.functional.softplus, beta = 1, threshold = 20
__device__ __forceinline__ float softplus_fwd(float u) { return u > 20.f ? u : log1pf(expf(u)); }

// value the network output `pred` turns into: denormalise (float64, then back to float32), add the inverse-transformed
// baseline (residual prediction), transform, clip, dampen.  Returns through `u_out` the argument of the transform (for
// the backward).
__device__ __forceinline__ float post_one(const PostArgs& p, float pred, int64_t oi, float* u_out = nullptr) {
  float v = pred;
  if (p.denorm) {
    const double sd = sqrt(p.var[0] + p.eps);
    // two separately rounded operations, as torch computes `input * sqrt(var + eps) + mean` (no fused multiply-add)
    v = static_cast<float>(__dadd_rn(__dmul_rn(static_cast<double>(v), sd), p.mean[0]));
  }
  if (p.apply_transform) {
    if (p.residual) {
      const float b = p.baseline[oi];
      float ib = b;
      if (p.transform == AMBER_TRANSFORM_SOFTPLUS) ib = softplus_inverse(b);
      else if (p.transform == AMBER_TRANSFORM_EXP) ib = logf(b);
      v = v + ib;
    }
    if (u_out) *u_out = v;
    if (p.transform == AMBER_TRANSFORM_SOFTPLUS) {
      v = softplus_fwd(v);
    } else if (p.transform == AMBER_TRANSFORM_EXP) {
      v = expf(v);
      if (v != v) v = 100.f;  // exponential_transform.py:31-35: NaN predictions are replaced by 100
    }
  } else if (u_out) {
    *u_out = v;
  }
  if (p.clip & 1) v = v < p.clip_min ? p.clip_min : v;  // np.clip: max(min(v, hi), lo); NaN propagates
  if (p.clip & 2) v = v > p.clip_max ? p.clip_max : v;
  if (p.scale0 != 1.f) v = v * p.scale0;
  if (p.scale1 != 1.f) v = v * p.scale1;
  return v;
}

}  // namespace amber
