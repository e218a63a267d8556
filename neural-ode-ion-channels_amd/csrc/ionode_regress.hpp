// ionode_regress.hpp -- the reference's MLP state-space regression step on the fp32 MFMA.
//
//   reference (train-s1.py:891-909, train-d2.py:901-915):   p = net(x_av.float()) / netscale  [+ model_dadt]
//                                                           loss = MSELoss(reduction='sum')(p, y_dadt.float())
//                                                           loss.backward(); Adam(lr 1e-3).step(); StepLR.step()
// full batch (~132 k rows of (V/100, a) -> da/dt), 4000-8000 iterations: the reference's actual training cost.
//
// One iteration = three kernels, no host synchronisation in between:
//   ionode_regress_kernel   16 rows per tile (one per MFMA column), forward + backward of the whole net for the tile with every
//                           layer's activations in LDS -- the SAME GradMlp::vjp as the ODE backward sweep, seeded with
//                           d loss / d net = 2 (p - y) / netscale; activations never leave the CU except as the (d_l, h_l)
//                           record of the tile (ionode_grad.hpp)
//   ionode_grad_reduce      split-K fp32 MFMA GEMM of the records -> per-slab partial dW, db (ionode_grad_reduce.hpp)
//   ionode_adam_kernel      slab sum + torch.optim.Adam's update on the flat state dict + refresh of the MFMA fragment image
#pragma once

#include "ionode_grad_reduce.hpp"

namespace ionode {

struct RArgs {
  const float *img;       // grad image of the CURRENT weights
  const float *x;         // [M][2] fp32 rows (V / vrange, a) -- `x_av.float()`
  const float *y;         // [M] fp32 targets -- `y_dadt.float()`
  const float *offset;    // [M] fp32 closed-form term added to net/netscale (NN-d: model_dadt, train-d2.py:903) or NULL
  float *records;         // [n_tiles][record_floats]
  double *loss_part;      // [gridDim.x] partial sums of squared residuals
  int32_t M, L, N, NT;
  int64_t record_floats;
  float netscale;
};

// N <= 200: TWO workgroups per compute unit (two wavefronts per SIMD, <= 256 registers, 70 KB of LDS each): one tile's layer boundaries,
// barriers and record stores run beside the other's MFMAs.  N = 500 fills the register file and the LDS with one.
#ifndef IONODE_REGRESS_WG_PER_CU
#define IONODE_REGRESS_WG_PER_CU 2
#endif
template <int NT>
__global__ void __launch_bounds__(256, (NT <= 13 ? IONODE_REGRESS_WG_PER_CU : 1)) ionode_regress_kernel(const RArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  GArgs g;  // GradMlp::init reads the image pointer and the MLP shape only
  g.img = a.img; g.k.L = a.L;
  GradMlp<NT> mlp;
  mlp.init(g, smem, wave, lane);
  const int n_tiles = (a.M + 15) / 16;
  double acc = 0.0;
  // a tile's inputs are fetched ONE TILE AHEAD (round 5): loaded at the top of their own tile, their wait -- vmcnt(0), which also covers the
  // weight ring's refills in flight -- stood in front of the tile's first instruction: one exposed memory round trip per tile
  struct In { float x0, x1, off, yt; };
  auto fetch = [&](int tile) -> In {
    const int row = tile * 16 + j;
    const int r = row < a.M ? row : a.M - 1;
    const f32x2 xx = *reinterpret_cast<const f32x2 *>(a.x + 2 * (size_t)r);
    return In{xx[0], xx[1], a.offset ? a.offset[r] : 0.0f, a.y[r]};
  };
  In nxt = fetch((int)blockIdx.x < n_tiles ? (int)blockIdx.x : 0);
  asm volatile("" : "+v"(nxt.x0), "+v"(nxt.x1), "+v"(nxt.off), "+v"(nxt.yt));   // (waited for here, not behind the loop header: see ionode_grad_reduce.hpp)
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const bool valid = tile * 16 + j < a.M;
    const In cur = nxt;
    if (tile + (int)gridDim.x < n_tiles) nxt = fetch(tile + (int)gridDim.x);
    const float x0 = cur.x0, x1 = cur.x1, off = cur.off, yt = cur.yt;
    const float ns = a.netscale;
    float resid = 0.0f;
    mlp.vjp_from_output(x0, x1, a.records + (size_t)tile * a.record_floats, [&](float net) -> float {
      // p = net / netscale (+ model_dadt), then MSELoss(sum): d loss / d net = 2 (p - y) / netscale   (all fp32, as torch)
      float p = net / ns;
      if (a.offset) p = p + off;
      resid = valid ? p - yt : 0.0f;
      return valid ? (2.0f * resid) / ns : 0.0f;
    });
    if (wave == 0 && lane < 16) acc += (double)resid * (double)resid;
  }
  // one partial per workgroup (deterministic; the host / Adam kernel sums them)
  if (wave == 0) {
    double t = (lane < 16) ? acc : 0.0;
#pragma unroll
    for (int s = 1; s < 16; s <<= 1) t += __shfl_xor(t, s);
    if (lane == 0) a.loss_part[blockIdx.x] = t;
  }
}

#ifndef IONODE_GRAD_TEMPLATES_ONLY  // (inst_grad32.hip instantiates the N = 500 templates only)
// torch.optim.Adam (no amsgrad, no weight decay) on the flat state dict, fp32, element-wise as torch computes it:
//   m = m + (g - m) * (1 - beta1);  v = v * beta2 + g * g * (1 - beta2)
//   denom = sqrt(v) / sqrt(1 - beta2^t) + eps;  w = w - (lr / (1 - beta1^t)) * m / denom
// g = sum over the reduce kernel's slabs (fp32, slab order).  grad_out (optional) receives g.
__global__ void ionode_adam_kernel(int n, int n_slabs, size_t partf, const float *__restrict__ partials,
                                   const int32_t *__restrict__ padmap, float *__restrict__ w, float *__restrict__ m,
                                   float *__restrict__ v, float lr, float beta1, float beta2, float eps, float bc1, float bc2_sqrt,
                                   float *__restrict__ grad_out, int apply) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t k = (size_t)padmap[i];
  float g = 0.0f;
  for (int s = 0; s < n_slabs; ++s) g += partials[(size_t)s * partf + k];
  if (grad_out) grad_out[i] = g;
  if (!apply) return;
  float mi = m[i], vi = v[i];
  mi = mi + (g - mi) * (1.0f - beta1);
  vi = vi * beta2 + (g * g) * (1.0f - beta2);
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  w[i] = w[i] - (lr / bc1) * (mi / denom);
  m[i] = mi; v[i] = vi;
}

// image[k] = flat[imgmap[k] - 1] (imgmap[k] == 0: padding) -- the grad image of the updated weights, rebuilt on the device
__global__ void ionode_image_refresh_kernel(size_t n_img, const int32_t *__restrict__ imgmap, const float *__restrict__ w,
                                            float *__restrict__ img) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_img) return;
  const int32_t s = imgmap[k];
  img[k] = s > 0 ? w[s - 1] : 0.0f;
}

#endif  // IONODE_GRAD_TEMPLATES_ONLY

template <int NT> inline void launch_regress(const RArgs &a, unsigned grid, hipStream_t s) {
  const size_t lds = grad_lds_bytes(a.L, NT);
  auto kern = ionode_regress_kernel<NT>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
}

}  // namespace ionode
