// Training-mode dropout of the ChainedDiffuser transformer (p = 0.1 in the reference's shipped configuration):
//   nn.Dropout after the attention / FFN residual branches       layers.py:34,58,82-84,146,181
//   nn.Dropout(0.1) inside traj_encoder / pos_regressor / rot_regressor   diffusion_head.py:46,183,193
// (the dropout on the attention WEIGHTS, multihead_custom_attention.py:413, is generated inside the attention kernels.)
// Counter-based: y = x * keep(index, site, {seed, offset}) / (1 - p) with the Philox block function of a3d_common.h, so
// the backward pass regenerates the mask instead of storing it (the same entry point serves both directions) and the
// training step stays capturable in a hipGraph -- the state is a device-resident snapshot, no host RNG is involved.
#include "a3d_common.h"
#include "../../include/act3d_hip.h"

namespace a3d {

// one thread per 8-element block (one Philox call); n need not be a multiple of 8
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n,
                                                      const unsigned long long* __restrict__ state, uint32_t site,
                                                      uint32_t thr16, float scale) {
  const DropKey key = drop_key(state);
  const size_t nblk = (n + 7) >> 3;
  for (size_t blk = (size_t)blockIdx.x * blockDim.x + threadIdx.x; blk < nblk; blk += (size_t)gridDim.x * blockDim.x) {
    const uint32_t keep = drop_keep8(key, (uint32_t)blk, (uint32_t)(blk >> 32), 0xFFFFFFFFu, site, thr16);
    const size_t base = blk << 3;
    if (base + 8 <= n && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0) {
      float4 a = *reinterpret_cast<const float4*>(x + base), b = *reinterpret_cast<const float4*>(x + base + 4);
      a.x = (keep & 1u) ? a.x * scale : 0.f;
      a.y = (keep & 2u) ? a.y * scale : 0.f;
      a.z = (keep & 4u) ? a.z * scale : 0.f;
      a.w = (keep & 8u) ? a.w * scale : 0.f;
      b.x = (keep & 16u) ? b.x * scale : 0.f;
      b.y = (keep & 32u) ? b.y * scale : 0.f;
      b.z = (keep & 64u) ? b.z * scale : 0.f;
      b.w = (keep & 128u) ? b.w * scale : 0.f;
      *reinterpret_cast<float4*>(y + base) = a;
      *reinterpret_cast<float4*>(y + base + 4) = b;
    } else {
      for (int j = 0; j < 8 && base + j < n; ++j) y[base + j] = ((keep >> j) & 1u) ? x[base + j] * scale : 0.f;
    }
  }
}

// keep flags as bytes (tests / CPU twin comparison): out[i] = 1 if element i is kept
__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned char* __restrict__ out, size_t n,
                                                           const unsigned long long* __restrict__ state, uint32_t c2,
                                                           uint32_t c1, uint32_t site, uint32_t thr16) {
  const DropKey key = drop_key(state);
  const size_t nblk = (n + 7) >> 3;
  for (size_t blk = (size_t)blockIdx.x * blockDim.x + threadIdx.x; blk < nblk; blk += (size_t)gridDim.x * blockDim.x) {
    const uint32_t keep = (c2 == 0xFFFFFFFFu) ? drop_keep8(key, (uint32_t)blk, (uint32_t)(blk >> 32), c2, site, thr16)
                                              : drop_keep8(key, (uint32_t)blk, c1, c2, site, thr16);
    for (int j = 0; j < 8 && (blk << 3) + j < n; ++j) out[(blk << 3) + j] = (unsigned char)((keep >> j) & 1u);
  }
}

}  // namespace a3d

using namespace a3d;

static int drop_params(const char* fn, float p, uint32_t* thr16, float* scale) {
  if (!(p >= 0.f) || !(p < 1.f)) {
    set_error("%s: dropout probability %g outside [0, 1)", fn, (double)p);
    return A3D_ERR_ARG;
  }
  *thr16 = (uint32_t)lrintf(p * 65536.0f);
  *scale = 1.0f / (1.0f - p);
  return A3D_OK;
}

extern "C" int a3d_dropout(const float* x, float* y, size_t n, const unsigned long long* state, unsigned int site,
                           float p, void* stream) {
  if (!x || !y || !state) { set_error("a3d_dropout: null pointer"); return A3D_ERR_ARG; }
  uint32_t thr; float scale;
  int rc = drop_params("a3d_dropout", p, &thr, &scale);
  if (rc) return rc;
  if (!n) return A3D_OK;
  const size_t nblk = (n + 7) >> 3;
  hipLaunchKernelGGL(dropout_kernel, dim3((int)std::min<size_t>((nblk + 255) / 256, 4096)), dim3(256), 0,
                     (hipStream_t)stream, x, y, n, state, site, thr, scale);
  return check_launch("a3d_dropout");
}

extern "C" int a3d_dropout_mask(unsigned char* out, size_t n, const unsigned long long* state, unsigned int c2,
                                unsigned int c1, unsigned int site, float p, void* stream) {
  if (!out || !state) { set_error("a3d_dropout_mask: null pointer"); return A3D_ERR_ARG; }
  uint32_t thr; float scale;
  int rc = drop_params("a3d_dropout_mask", p, &thr, &scale);
  if (rc) return rc;
  if (!n) return A3D_OK;
  const size_t nblk = (n + 7) >> 3;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((int)std::min<size_t>((nblk + 255) / 256, 4096)), dim3(256), 0,
                     (hipStream_t)stream, out, n, state, c2, c1, site, thr);
  return check_launch("a3d_dropout_mask");
}
