// Decoding heads, losses, ghost-point sampler and optimizer of the keypose path (all HBM/latency-bound, tiny).
//
//   a3d_mask_logits_{fwd,bwd}   einsum("bt c, npts bt c -> bt npts")            act3d.py:493-494
//   a3d_argmax_gather           torch.max(mask).indices (first max) + coordinate gather   act3d.py:312-314, 512-513
//   a3d_soft_ce_loss            label = softmax(-||ghost-gt||/spread); CE(logits, label)  main_keypose.py:382-405
//   a3d_quat_sigmoid_{fwd,bwd}  normalise_quat + sigmoid of the 5-vector prediction        act3d.py:526-533, utils.py:51-52
//   a3d_ortho6d_sigmoid_{fwd,bwd}  6D -> rotation matrix + sigmoid (the 6D_* heads)        act3d.py:529-533, utils.py:93-130
//   a3d_select_row_{fwd,bwd}    row of the top-scoring ghost point (offset / feature)     act3d.py:513-522
//   a3d_mse_loss / a3d_l1_loss  F.mse_loss / F.l1_loss (mean) with gradient               main_keypose.py:362-380, diffusion_model.py:315-323
//   a3d_sample_ghost_points     Philox4x32-10 uniform cube / ball-rejection sampler        act3d.py:394-440, utils.py:68-84
//   a3d_adamw_step              torch.optim.AdamW update on the flat parameter buffer      engine.py:89-102
#include "a3d_common.h"
#include "../../include/act3d_hip.h"

namespace a3d {

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = -INFINITY;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s = fmaxf(s, red[w]);
  return s;
}

__global__ __launch_bounds__(256) void mask_logits_fwd_kernel(
    const float* __restrict__ q, const float* __restrict__ F, float* __restrict__ out, int B, int Ng, int E) {
  const size_t total = (size_t)B * Ng;
  for (size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x; row < total;
       row += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(row / Ng);
    const float* f = F + row * E;
    const float* qq = q + (size_t)b * E;
    float s = 0.f;
    for (int c = 0; c < E; ++c) s += qq[c] * f[c];
    out[row] = s;
  }
}

// dF[b][n][c] (+)= dlog[b][n] q[b][c];  dq[b][c] = sum_n dlog[b][n] F[b][n][c]
// one 1024-thread workgroup per sample: 16 row-groups x 64 channels (x2 for E > 64), rows summed in a fixed order
__global__ __launch_bounds__(1024) void mask_logits_bwd_kernel(
    const float* __restrict__ q, const float* __restrict__ F, const float* __restrict__ dlog,
    float* __restrict__ dF, float* __restrict__ dq, int B, int Ng, int E, int accumulate_dF) {
  __shared__ float part[16][128];
  const int b = blockIdx.x, t = threadIdx.x;
  const int c = t & 63, pr = t >> 6;
  const float q0 = (c < E) ? q[(size_t)b * E + c] : 0.f;
  const float q1 = (c + 64 < E) ? q[(size_t)b * E + c + 64] : 0.f;
  float a0 = 0.f, a1 = 0.f;
  for (int n = pr; n < Ng; n += 16) {
    const float d = dlog[(size_t)b * Ng + n];
    const size_t base = ((size_t)b * Ng + n) * E;
    if (c < E) {
      a0 += d * F[base + c];
      const float v = d * q0;
      if (accumulate_dF) dF[base + c] += v; else dF[base + c] = v;
    }
    if (c + 64 < E) {
      a1 += d * F[base + c + 64];
      const float v = d * q1;
      if (accumulate_dF) dF[base + c + 64] += v; else dF[base + c + 64] = v;
    }
  }
  part[pr][c] = a0;
  part[pr][c + 64] = a1;
  __syncthreads();
  if (t < E) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += part[u][t];
    dq[(size_t)b * E + t] = s;
  }
}

__global__ __launch_bounds__(256) void argmax_gather_kernel(
    const float* __restrict__ logits, const float* __restrict__ ghost, long long* __restrict__ top_idx,
    float* __restrict__ pos, int Ng) {
  __shared__ float bv[256];
  __shared__ int bi[256];
  const int b = blockIdx.x, t = threadIdx.x;
  float best = -INFINITY;
  int besti = 0x7FFFFFFF;
  for (int n = t; n < Ng; n += 256) {
    const float v = logits[(size_t)b * Ng + n];
    if (v > best || (v == best && n < besti) || besti == 0x7FFFFFFF) { best = v; besti = n; }
  }
  bv[t] = best; bi[t] = besti;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) {
      const float v2 = bv[t + s];
      const int i2 = bi[t + s];
      if (i2 != 0x7FFFFFFF && (bi[t] == 0x7FFFFFFF || v2 > bv[t] || (v2 == bv[t] && i2 < bi[t]))) { bv[t] = v2; bi[t] = i2; }
    }
    __syncthreads();
  }
  if (t == 0) top_idx[b] = bi[0];
  if (t < 3 && pos) pos[b * 3 + t] = ghost[((size_t)b * Ng + bi[0]) * 3 + t];
}

// per sample: loss_b = -sum_n y_n log_softmax(z)_n,  y = (1-ls) softmax(-l2/spread) + ls/Ng
// dz = gscale * (softmax(z) - y)     (sum_n y_n = 1)
__global__ __launch_bounds__(256) void soft_ce_kernel(
    const float* __restrict__ ghost, const float* __restrict__ gt, const float* __restrict__ logits,
    float* __restrict__ loss_b, float* __restrict__ dlogits, int Ng, float spread, float label_smoothing,
    float gscale) {
  __shared__ float red[4];
  const int b = blockIdx.x, t = threadIdx.x;
  const float gx = gt[b * 3 + 0], gy = gt[b * 3 + 1], gz = gt[b * 3 + 2];
  const float* z = logits + (size_t)b * Ng;
  // pass 1: maxima
  float mz = -INFINITY, ma = -INFINITY;
  for (int n = t; n < Ng; n += 256) {
    const float* p = ghost + ((size_t)b * Ng + n) * 3;
    const float dx = p[0] - gx, dy = p[1] - gy, dz = p[2] - gz;
    const float a = -sqrtf(dx * dx + dy * dy + dz * dz) / spread;
    ma = fmaxf(ma, a);
    mz = fmaxf(mz, z[n]);
  }
  ma = block_max(ma, red);
  mz = block_max(mz, red);
  float sa = 0.f, sz = 0.f;
  for (int n = t; n < Ng; n += 256) {
    const float* p = ghost + ((size_t)b * Ng + n) * 3;
    const float dx = p[0] - gx, dy = p[1] - gy, dz = p[2] - gz;
    const float a = -sqrtf(dx * dx + dy * dy + dz * dz) / spread;
    sa += expf(a - ma);
    sz += expf(z[n] - mz);
  }
  sa = block_sum(sa, red);
  sz = block_sum(sz, red);
  const float lse = mz + logf(sz);
  float acc = 0.f;
  for (int n = t; n < Ng; n += 256) {
    const float* p = ghost + ((size_t)b * Ng + n) * 3;
    const float dx = p[0] - gx, dy = p[1] - gy, dz = p[2] - gz;
    const float a = -sqrtf(dx * dx + dy * dy + dz * dz) / spread;
    const float y = (1.f - label_smoothing) * (expf(a - ma) / sa) + label_smoothing / (float)Ng;
    const float lsm = z[n] - lse;
    acc -= y * lsm;
    if (dlogits) dlogits[(size_t)b * Ng + n] = gscale * (expf(lsm) - y);
  }
  acc = block_sum(acc, red);
  if (t == 0) loss_b[b] = acc;
}

// out[0] = scale * sum(in[0..n))   (single workgroup; n is small)
__global__ __launch_bounds__(256) void reduce_sum_kernel(const float* __restrict__ in, int n, float scale,
                                                         float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += in[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s * scale;
}

// kind 0: mse, 1: l1.  loss = coeff * mean(f(p - t)); grad = d loss / d p
__global__ __launch_bounds__(256) void elem_loss_kernel(const float* __restrict__ p, const float* __restrict__ tg,
                                                        int n, int kind, float coeff, float* __restrict__ loss,
                                                        float* __restrict__ grad) {
  __shared__ float red[4];
  float s = 0.f;
  const float inv = coeff / (float)n;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float d = p[i] - tg[i];
    if (kind == 0) {
      s += d * d;
      if (grad) grad[i] = 2.f * d * inv;
    } else {
      s += fabsf(d);
      if (grad) grad[i] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * inv;
    }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] = s * inv;
}

__global__ void scale_by_scalar_kernel(const float* __restrict__ x, const float* __restrict__ sc, float* __restrict__ y,
                                       size_t n) {
  const float s = sc[0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i] * s;
}

// pred [B][5] -> rot [B][4] = x / max(|x|, 1e-10), grip [B][1] = sigmoid
__global__ void quat_sigmoid_fwd_kernel(const float* __restrict__ pred, float* __restrict__ rot,
                                        float* __restrict__ grip, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* p = pred + (size_t)b * 5;
  const float nrm = fmaxf(sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3]), 1e-10f);
  for (int i = 0; i < 4; ++i) rot[b * 4 + i] = p[i] / nrm;
  grip[b] = 1.f / (1.f + expf(-p[4]));
}
__global__ void quat_sigmoid_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ drot,
                                        const float* __restrict__ dgrip, float* __restrict__ dpred, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* p = pred + (size_t)b * 5;
  const float n2 = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3]);
  const float nrm = fmaxf(n2, 1e-10f);
  float dot = 0.f;
  for (int i = 0; i < 4; ++i) dot += (drot ? drot[b * 4 + i] : 0.f) * p[i];
  for (int i = 0; i < 4; ++i) {
    const float dy = drot ? drot[b * 4 + i] : 0.f;
    // y = x / n: dx = dy / n - x (x.dy) / n^3   (the clamp branch has zero norm-gradient)
    dpred[b * 5 + i] = (n2 > 1e-10f) ? (dy / nrm - p[i] * dot / (nrm * nrm * nrm)) : dy / nrm;
  }
  const float sg = 1.f / (1.f + expf(-p[4]));
  dpred[b * 5 + 4] = (dgrip ? dgrip[b] : 0.f) * sg * (1.f - sg);
}

// 6D head (act3d.py:529-531, model/utils/utils.py:93-130): pred [B][7] -> rot [B][3][3] with COLUMNS
// x = a / |a|, y = z x x, z = (x x b) / |x x b|  (a = pred[0:3], b = pred[3:6]; |.| clamped at 1e-8), grip = sigmoid(pred[6]).
__device__ __forceinline__ void cross3(const float* u, const float* v, float* o) {
  o[0] = u[1] * v[2] - u[2] * v[1];
  o[1] = u[2] * v[0] - u[0] * v[2];
  o[2] = u[0] * v[1] - u[1] * v[0];
}
__global__ void ortho6d_sigmoid_fwd_kernel(const float* __restrict__ pred, float* __restrict__ rot,
                                           float* __restrict__ grip, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* p = pred + (size_t)b * 7;
  float x[3], y[3], z[3];
  const float na = fmaxf(sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]), 1e-8f);
  for (int i = 0; i < 3; ++i) x[i] = p[i] / na;
  cross3(x, p + 3, z);
  const float nz = fmaxf(sqrtf(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]), 1e-8f);
  for (int i = 0; i < 3; ++i) z[i] /= nz;
  cross3(z, x, y);
  float* r = rot + (size_t)b * 9;
  for (int i = 0; i < 3; ++i) { r[i * 3 + 0] = x[i]; r[i * 3 + 1] = y[i]; r[i * 3 + 2] = z[i]; }
  grip[b] = 1.f / (1.f + expf(-p[6]));
}
// Reverse of the chain above.  u = v / max(|v|, eps): dv = (du - u (u.du)) / |v| on the unclamped branch, du / eps else.
__global__ void ortho6d_sigmoid_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ drot,
                                           const float* __restrict__ dgrip, float* __restrict__ dpred, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* p = pred + (size_t)b * 7;
  float x[3], z0[3], z[3], gx[3] = {0.f, 0.f, 0.f}, gy[3] = {0.f, 0.f, 0.f}, gz[3] = {0.f, 0.f, 0.f}, t[3];
  const float ma = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  const float na = fmaxf(ma, 1e-8f);
  for (int i = 0; i < 3; ++i) x[i] = p[i] / na;
  cross3(x, p + 3, z0);
  const float mz = sqrtf(z0[0] * z0[0] + z0[1] * z0[1] + z0[2] * z0[2]);
  const float nz = fmaxf(mz, 1e-8f);
  for (int i = 0; i < 3; ++i) z[i] = z0[i] / nz;
  if (drot) {
    const float* g = drot + (size_t)b * 9;
    for (int i = 0; i < 3; ++i) { gx[i] = g[i * 3 + 0]; gy[i] = g[i * 3 + 1]; gz[i] = g[i * 3 + 2]; }
  }
  // y = z x x
  cross3(x, gy, t);
  for (int i = 0; i < 3; ++i) gz[i] += t[i];
  cross3(gy, z, t);
  for (int i = 0; i < 3; ++i) gx[i] += t[i];
  // z = z0 / max(|z0|, eps)
  float gz0[3];
  {
    const float d = z[0] * gz[0] + z[1] * gz[1] + z[2] * gz[2];
    for (int i = 0; i < 3; ++i) gz0[i] = (mz > 1e-8f) ? (gz[i] - z[i] * d) / nz : gz[i] / nz;
  }
  // z0 = x x b
  cross3(p + 3, gz0, t);
  for (int i = 0; i < 3; ++i) gx[i] += t[i];
  cross3(gz0, x, t);
  for (int i = 0; i < 3; ++i) dpred[(size_t)b * 7 + 3 + i] = t[i];
  // x = a / max(|a|, eps)
  {
    const float d = x[0] * gx[0] + x[1] * gx[1] + x[2] * gx[2];
    for (int i = 0; i < 3; ++i) dpred[(size_t)b * 7 + i] = (ma > 1e-8f) ? (gx[i] - x[i] * d) / na : gx[i] / na;
  }
  const float sg = 1.f / (1.f + expf(-p[6]));
  dpred[(size_t)b * 7 + 6] = (dgrip ? dgrip[b] : 0.f) * sg * (1.f - sg);
}

// y[b][:] = x[b][idx[b]][:]   (act3d.py:513-522: the top ghost point's offset / feature row); the backward writes the whole
// gradient map in one pass (zero except the selected row), so no memset precedes it.
__global__ void select_row_fwd_kernel(const float* __restrict__ x, const long long* __restrict__ idx, float* __restrict__ y,
                                      int B, int N, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * W) return;
  const int b = i / W, c = i - b * W;
  y[i] = x[((size_t)b * N + idx[b]) * W + c];
}
__global__ void select_row_bwd_kernel(const float* __restrict__ dy, const long long* __restrict__ idx, float* __restrict__ dx,
                                      int B, int N, int W) {
  const size_t total = (size_t)B * N * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % W);
    const size_t bn = i / W;
    const int b = (int)(bn / N);
    const long long n = (long long)(bn - (size_t)b * N);
    dx[i] = (n == idx[b]) ? dy[(size_t)b * W + c] : 0.f;
  }
}

// Philox4x32-10 lives in a3d_common.h (shared with the dropout masks of attention.hip / dropout.hip)
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// state[0] = seed, state[1] = call offset (uint64 each).  anchor == null: uniform in [lo, hi].
// else: uniform in ball(anchor, radius) intersected with the box clip(anchor -+ radius, lo, hi) by rejection
// (at most max_attempts draws per point, then the clipped anchor itself).
__global__ __launch_bounds__(256) void sample_ghost_kernel(
    const unsigned long long* __restrict__ state, const float* __restrict__ bounds, const float* __restrict__ anchor,
    float radius, float* __restrict__ out, int B, int Ng, int level, int max_attempts) {
  const size_t total = (size_t)B * Ng;
  const unsigned long long seed = state[0], offs = state[1];
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(idx / Ng), i = (int)(idx - (size_t)b * Ng);
    float lo[3], hi[3], ctr[3];
    for (int a = 0; a < 3; ++a) {
      lo[a] = bounds[a]; hi[a] = bounds[3 + a];
      if (anchor) {
        ctr[a] = anchor[b * 3 + a];
        lo[a] = fminf(fmaxf(sub_rn(ctr[a], radius), bounds[a]), bounds[3 + a]);
        hi[a] = fminf(fmaxf(add_rn(ctr[a], radius), bounds[a]), bounds[3 + a]);
      }
    }
    float p[3] = {0.f, 0.f, 0.f};
    bool ok = false;
    const int tries = anchor ? max_attempts : 1;
    for (int a = 0; a < tries && !ok; ++a) {
      uint32_t r[4];
      philox4x32_10((uint32_t)i, (uint32_t)b, (uint32_t)level | ((uint32_t)a << 8), (uint32_t)offs,
                    (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(offs >> 32), r);
      // explicit round-to-nearest ops (no fma contraction): the CPU twin in oracle/sampling.py must match bit for bit
      for (int c = 0; c < 3; ++c) p[c] = add_rn(lo[c], mul_rn(u01(r[c]), sub_rn(hi[c], lo[c])));
      if (!anchor) { ok = true; break; }
      const float dx = sub_rn(p[0], ctr[0]), dy = sub_rn(p[1], ctr[1]), dz = sub_rn(p[2], ctr[2]);
      ok = sqrt_rn(add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz))) < radius;
    }
    if (!ok) for (int c = 0; c < 3; ++c) p[c] = fminf(fmaxf(ctr[c], bounds[c]), bounds[3 + c]);
    for (int c = 0; c < 3; ++c) out[idx * 3 + c] = p[c];
  }
}
__global__ void rng_advance_kernel(unsigned long long* state, unsigned long long n) { state[1] += n; }

// ---------------------------------------------------------------- AdamW on the flat buffer
// torch.optim.AdamW decides PER PARAMETER: a parameter whose .grad is None is skipped (no decay, no moment update, its own
// `step` does not advance -- engine.py:121-124 find_unused_parameters: the FPN blocks / embeddings a configuration never
// reads), every other parameter is updated in full, zero-gradient elements (dead ReLU rows, pad channels) included, with the
// bias correction of ITS OWN step count.  The flat buffer has no None: seg_off [nseg + 1] delimits the parameters, and
// seg_state [nseg][4] = {step, active, lr / bc1, sqrt(bc2)} carries the per-parameter state.  A parameter is "in the graph"
// from the first step in which any element of its gradient segment is non-zero, and stays in (which tensors a model's
// forward reaches is structural); adamw_prepare_kernel (one workgroup per parameter; the scan only runs until the parameter
// is in) advances the step counts and leaves the coefficients for adamw_kernel, whose threads find their element's parameter
// by bisection of the LDS-staged offsets.  Elements [0, n_nodecay) use weight decay wd0, the rest wd1.
constexpr int ADAMW_LDS_SEGS = 4096;
__global__ __launch_bounds__(256) void adamw_prepare_kernel(const float* __restrict__ g, const long long* __restrict__ seg_off,
                                                            float* __restrict__ seg_state, float* __restrict__ step, float lr,
                                                            float beta1, float beta2) {
  const int seg = blockIdx.x;
  float* st = seg_state + (size_t)seg * 4;
  const long long a = seg_off[seg], b = seg_off[seg + 1];
  int in_graph = st[0] > 0.0f;
  if (!in_graph) {
    int nz = 0;
    for (long long i = a + threadIdx.x; i < b; i += blockDim.x) nz |= (g[i] != 0.0f);
    in_graph = __syncthreads_or(nz);
  }
  if (threadIdx.x == 0) {
    if (in_graph) {
      const float t = st[0] + 1.0f;
      st[0] = t;
      st[1] = 1.0f;
      st[2] = lr / (1.0f - powf(beta1, t));
      st[3] = sqrtf(1.0f - powf(beta2, t));
    } else {
      st[1] = 0.0f;
    }
    if (seg == 0) step[0] += 1.0f;            // completed optimizer steps (the value torch keeps for every updated parameter)
  }
}
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    const long long* __restrict__ seg_off, const float* __restrict__ seg_state,
                                                    int nseg, size_t n, size_t n_nodecay, float lr, float beta1, float beta2,
                                                    float eps, float wd0, float wd1, float gscale) {
  __shared__ long long offS[ADAMW_LDS_SEGS + 1];
  const bool staged = nseg <= ADAMW_LDS_SEGS;
  if (staged) {
    for (int i = threadIdx.x; i <= nseg; i += blockDim.x) offS[i] = seg_off[i];
    __syncthreads();
  }
  const long long* off = staged ? offS : seg_off;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    int lo = 0, hi = nseg - 1;                 // largest seg with off[seg] <= i
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if ((size_t)off[mid] <= i) lo = mid; else hi = mid - 1;
    }
    const float* st = seg_state + (size_t)lo * 4;
    if (st[1] == 0.0f) continue;               // parameter not in the graph: left alone, as torch skips .grad is None
    const float step_size = st[2], bc2s = st[3];
    const float gi = g[i] * gscale;
    const float wd = (i < n_nodecay) ? wd0 : wd1;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2s + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// ---------------------------------------------------------------- keypose evaluation metrics, per-sample columns
// LossAndMetrics.compute_metrics (main_keypose.py:431-482) as a table: one thread per sample writes
//   cols[b] = { e_final, [e_final < 0.01], e_level_0 .. e_level_{nlev-1}, rot_l1, [rot_l1 < 0.05], [rot_l1 < 0.025], grip_ok }
// with e = |position - gt_xyz|_2, rot_l1 = |quat - gt_quat|_1 (symmetric: min over +-gt), grip_ok = (open > 0.5) == gt_open;
// the per-task / overall means are one small matrix product with the group-indicator matrix on the host side of the API.
__global__ void keypose_errors_kernel(const float* __restrict__ pos, const float* __restrict__ rot,
                                      const float* __restrict__ grip, const float* __restrict__ gt, int ldgt,
                                      float* __restrict__ cols, int B, int nlev, int symmetric) {
  const int K = 6 + nlev;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
    const float* g = gt + (size_t)b * ldgt;
    float* o = cols + (size_t)b * K;
    for (int s = 0; s <= nlev; ++s) {
      const float* p = pos + ((size_t)s * B + b) * 3;
      const float dx = p[0] - g[0], dy = p[1] - g[1], dz = p[2] - g[2];
      const float e = sqrtf(dx * dx + dy * dy + dz * dz);
      if (s == 0) { o[0] = e; o[1] = e < 0.01f ? 1.f : 0.f; }
      else o[1 + s] = e;
    }
    float a = 0.f, a_neg = 0.f;
    for (int c = 0; c < 4; ++c) { a += fabsf(rot[b * 4 + c] - g[3 + c]); a_neg += fabsf(rot[b * 4 + c] + g[3 + c]); }
    const float l1 = symmetric ? fminf(a, a_neg) : a;
    o[2 + nlev] = l1;
    o[3 + nlev] = l1 < 0.05f ? 1.f : 0.f;
    o[4 + nlev] = l1 < 0.025f ? 1.f : 0.f;
    o[5 + nlev] = ((grip[b] > 0.5f) == (g[7] != 0.f)) ? 1.f : 0.f;
  }
}

// symmetric quaternion regression loss (main_keypose.py:370-376): coeff * mean_b min(mse(q, g), mse(q, -g)); grad optional
__global__ __launch_bounds__(256) void sym_quat_loss_kernel(const float* __restrict__ q, const float* __restrict__ gt, int ldgt,
                                                            float coeff, float* __restrict__ loss, float* __restrict__ grad, int B) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float a = 0.f, an = 0.f;
    for (int c = 0; c < 4; ++c) {
      const float d = q[b * 4 + c] - gt[(size_t)b * ldgt + c], dn = q[b * 4 + c] + gt[(size_t)b * ldgt + c];
      a += d * d; an += dn * dn;
    }
    const bool pos = a < an;                                    // select_mask = (quat_loss < quat_loss_)
    acc += (pos ? a : an) * 0.25f;
    if (grad)
      for (int c = 0; c < 4; ++c)
        grad[b * 4 + c] = coeff * 0.5f / (float)B * (q[b * 4 + c] + (pos ? -1.f : 1.f) * gt[(size_t)b * ldgt + c]);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = coeff * (red[0] + red[1] + red[2] + red[3]) / (float)B;
}

}  // namespace a3d

using namespace a3d;

static inline int grid_for(size_t n, int cap = 4096) { return (int)std::min<size_t>((n + 255) / 256, (size_t)cap); }

extern "C" int a3d_mask_logits_fwd(const float* q, const float* F, float* out, int B, int Ng, int E, void* stream) {
  if (!q || !F || !out || B <= 0 || Ng <= 0 || E <= 0) { set_error("a3d_mask_logits_fwd: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(mask_logits_fwd_kernel, dim3(grid_for((size_t)B * Ng)), dim3(256), 0, (hipStream_t)stream, q, F, out, B, Ng, E);
  return check_launch("a3d_mask_logits_fwd");
}
extern "C" int a3d_mask_logits_bwd(const float* q, const float* F, const float* dlog, float* dF, float* dq, int B,
                                   int Ng, int E, int accumulate_dF, void* stream) {
  if (!q || !F || !dlog || !dF || !dq || B <= 0 || Ng <= 0 || E <= 0 || E > 128) { set_error("a3d_mask_logits_bwd: bad argument (E=%d)", E); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(mask_logits_bwd_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, q, F, dlog, dF, dq, B, Ng, E, accumulate_dF);
  return check_launch("a3d_mask_logits_bwd");
}
extern "C" int a3d_argmax_gather(const float* logits, const float* ghost, long long* top_idx, float* pos, int B, int Ng,
                                 void* stream) {
  if (!logits || !top_idx || B <= 0 || Ng <= 0 || (pos && !ghost)) { set_error("a3d_argmax_gather: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(argmax_gather_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ghost, top_idx, pos, Ng);
  return check_launch("a3d_argmax_gather");
}
extern "C" int a3d_soft_ce_loss(const float* ghost, const float* gt, const float* logits, float* loss_b, float* loss,
                                float* dlogits, int B, int Ng, float spread, float label_smoothing, float coeff,
                                void* stream) {
  if (!ghost || !gt || !logits || !loss_b || !loss || B <= 0 || Ng <= 0 || spread <= 0.f) { set_error("a3d_soft_ce_loss: bad argument"); return A3D_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(soft_ce_kernel, dim3(B), dim3(256), 0, s, ghost, gt, logits, loss_b, dlogits, Ng, spread, label_smoothing, coeff / (float)B);
  int rc = check_launch("a3d_soft_ce_loss");
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, s, loss_b, B, coeff / (float)B, loss);
  return check_launch("a3d_soft_ce_loss(reduce)");
}
extern "C" int a3d_elem_loss(const float* pred, const float* target, int n, int kind, float coeff, float* loss,
                             float* grad, void* stream) {
  if (!pred || !target || !loss || n <= 0 || (kind != 0 && kind != 1)) { set_error("a3d_elem_loss: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(elem_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, target, n, kind, coeff, loss, grad);
  return check_launch("a3d_elem_loss");
}
extern "C" int a3d_scale_by_scalar(const float* x, const float* scalar, float* y, size_t n, void* stream) {
  if (!x || !scalar || !y) { set_error("a3d_scale_by_scalar: null pointer"); return A3D_ERR_ARG; }
  if (n == 0) return A3D_OK;
  hipLaunchKernelGGL(scale_by_scalar_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, scalar, y, n);
  return check_launch("a3d_scale_by_scalar");
}
extern "C" int a3d_quat_sigmoid_fwd(const float* pred, float* rot, float* grip, int B, void* stream) {
  if (!pred || !rot || !grip || B <= 0) { set_error("a3d_quat_sigmoid_fwd: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(quat_sigmoid_fwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, pred, rot, grip, B);
  return check_launch("a3d_quat_sigmoid_fwd");
}
extern "C" int a3d_quat_sigmoid_bwd(const float* pred, const float* drot, const float* dgrip, float* dpred, int B,
                                    void* stream) {
  if (!pred || !dpred || B <= 0) { set_error("a3d_quat_sigmoid_bwd: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(quat_sigmoid_bwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, pred, drot, dgrip, dpred, B);
  return check_launch("a3d_quat_sigmoid_bwd");
}
extern "C" int a3d_ortho6d_sigmoid_fwd(const float* pred, float* rot, float* grip, int B, void* stream) {
  if (!pred || !rot || !grip || B <= 0) { set_error("a3d_ortho6d_sigmoid_fwd: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(ortho6d_sigmoid_fwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, pred, rot, grip, B);
  return check_launch("a3d_ortho6d_sigmoid_fwd");
}
extern "C" int a3d_ortho6d_sigmoid_bwd(const float* pred, const float* drot, const float* dgrip, float* dpred, int B,
                                       void* stream) {
  if (!pred || !dpred || B <= 0) { set_error("a3d_ortho6d_sigmoid_bwd: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(ortho6d_sigmoid_bwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, pred, drot, dgrip, dpred, B);
  return check_launch("a3d_ortho6d_sigmoid_bwd");
}
extern "C" int a3d_select_row_fwd(const float* x, const long long* idx, float* y, int B, int N, int W, void* stream) {
  if (!x || !idx || !y || B <= 0 || N <= 0 || W <= 0) { set_error("a3d_select_row_fwd: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(select_row_fwd_kernel, dim3(cdiv(B * W, 256)), dim3(256), 0, (hipStream_t)stream, x, idx, y, B, N, W);
  return check_launch("a3d_select_row_fwd");
}
extern "C" int a3d_select_row_bwd(const float* dy, const long long* idx, float* dx, int B, int N, int W, void* stream) {
  if (!dy || !idx || !dx || B <= 0 || N <= 0 || W <= 0) { set_error("a3d_select_row_bwd: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(select_row_bwd_kernel, dim3(grid_for((size_t)B * N * W)), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, B, N, W);
  return check_launch("a3d_select_row_bwd");
}
extern "C" int a3d_sample_ghost_points(const unsigned long long* state, const float* bounds, const float* anchor,
                                       float radius, float* out, int B, int Ng, int level, int max_attempts,
                                       void* stream) {
  if (!state || !bounds || !out || B <= 0 || Ng <= 0 || level < 0 || level > 255 || max_attempts < 1 || max_attempts > 65535 || (anchor && radius <= 0.f)) {
    set_error("a3d_sample_ghost_points: bad argument");
    return A3D_ERR_ARG;
  }
  hipLaunchKernelGGL(sample_ghost_kernel, dim3(grid_for((size_t)B * Ng)), dim3(256), 0, (hipStream_t)stream, state, bounds, anchor, radius, out, B, Ng, level, max_attempts);
  return check_launch("a3d_sample_ghost_points");
}
extern "C" int a3d_rng_advance(unsigned long long* state, unsigned long long n, void* stream) {
  if (!state) { set_error("a3d_rng_advance: null state"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, n);
  return check_launch("a3d_rng_advance");
}
extern "C" void a3d_philox4x32_10_host(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}
extern "C" void a3d_sincos_host(const float* x, float* sn, float* cs, size_t n) {
  for (size_t i = 0; i < n; ++i) fast_sincos(x[i], sn + i, cs + i);
}
extern "C" int a3d_adamw_step(float* p, const float* g, float* m, float* v, float* step, const long long* seg_off,
                              float* seg_state, int nseg, size_t n, size_t n_nodecay, float lr, float beta1, float beta2, float eps,
                              float wd_nodecay, float wd_decay, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || !step || !seg_off || !seg_state) { set_error("a3d_adamw_step: null pointer"); return A3D_ERR_ARG; }
  if (n == 0) return A3D_OK;
  if (nseg <= 0) { set_error("a3d_adamw_step: the flat buffer needs at least one parameter segment (nseg = %d)", nseg); return A3D_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(adamw_prepare_kernel, dim3(nseg), dim3(256), 0, s, g, seg_off, seg_state, step, lr, beta1, beta2);
  int rc = check_launch("a3d_adamw_step(prepare)");
  if (rc) return rc;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, s, p, g, m, v, seg_off, seg_state, nseg, n, n_nodecay, lr,
                     beta1, beta2, eps, wd_nodecay, wd_decay, grad_scale);
  return check_launch("a3d_adamw_step");
}

extern "C" int a3d_keypose_errors(const float* pos, const float* rot, const float* grip, const float* gt, int ldgt, float* cols,
                                  int B, int nlev, int symmetric, void* stream) {
  if (!pos || !rot || !grip || !gt || !cols || B <= 0 || nlev < 0 || nlev > 8 || ldgt < 8) {
    set_error("a3d_keypose_errors: bad argument (B=%d nlev=%d ldgt=%d)", B, nlev, ldgt);
    return A3D_ERR_ARG;
  }
  hipLaunchKernelGGL(keypose_errors_kernel, dim3(cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, pos, rot, grip, gt, ldgt, cols, B,
                     nlev, symmetric);
  return check_launch("a3d_keypose_errors");
}
extern "C" int a3d_sym_quat_loss(const float* q, const float* gt, int ldgt, float coeff, float* loss, float* grad, int B,
                                 void* stream) {
  if (!q || !gt || !loss || B <= 0 || ldgt < 4) { set_error("a3d_sym_quat_loss: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(sym_quat_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, q, gt, ldgt, coeff, loss, grad, B);
  return check_launch("a3d_sym_quat_loss");
}
