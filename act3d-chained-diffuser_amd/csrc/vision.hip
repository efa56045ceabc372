// Frozen-backbone BatchNorm (train-mode statistics, as the reference runs it) fused with ReLU and the bottleneck
// residual add, for bf16 NHWC activations.  SURVEY §8f-1: after the hot path is fused, the frozen CLIP-RN50
// dominates the step, and more than half of its time is not convolution but BatchNorm + ReLU + add + dtype copies
// (profiles/r01_bench_eager_B64_*.txt).  The backbone is frozen (act3d.py:72-73) but never put in eval() and the
// trainer calls model.train() (engine.py:147), so its BatchNorm layers normalise with per-batch statistics and keep
// updating their running statistics: that behaviour is reproduced here.
//
//   a3d_bn_stats      per-channel partial (sum, sum of squares) over slabs of rows          1 read of x
//   a3d_bn_finalize   mean / biased var -> scale, shift; running-stat update (momentum, unbiased var)
//   a3d_bn_apply      y = relu?(x * scale[c] + shift[c] (+ residual))                        1 read (+1) + 1 write
// HBM-bound: 2 reads + 1 write of the activation instead of the 8-10 passes of MIOpen BN (3 kernels) + add + relu +
// casts.  x, residual, y: bf16, rows = N*H*W contiguous rows of C channels (torch channels_last), C % 8 == 0.
#include "a3d_common.h"
#include "../../include/act3d_hip.h"

namespace a3d {

// grid (nslab); each thread owns 8 consecutive channels (one 16-byte load per row); C / 8 threads span a row and the
// 256 / (C / 8) row-groups of the workgroup stride over the slab; an LDS tree adds the row-groups.  C % 8 == 0, C <= 2048.
__global__ __launch_bounds__(256) void bn_stats_kernel(const uint4* __restrict__ x, float* __restrict__ partial,
                                                       size_t rows, int C, int nslab) {
  __shared__ float red[256 * 16];
  const int tpr = C >> 3;                  // threads per row
  const int nsub = 256 / tpr;              // row-groups per workgroup
  const int cp = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
  const size_t per = (rows + nslab - 1) / nslab;
  const size_t r0 = (size_t)blockIdx.x * per;
  const size_t r1 = r0 + per < rows ? r0 + per : rows;
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
  if (rsub < nsub) {
    auto accum = [&](const uint4& v) {
      const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = __uint_as_float(w[j] << 16), b = __uint_as_float(w[j] & 0xFFFF0000u);
        s[2 * j] += a; q[2 * j] += a * a;
        s[2 * j + 1] += b; q[2 * j + 1] += b * b;
      }
    };
    size_t r = r0 + rsub;
    const size_t step = (size_t)nsub;
    // eight independent 16-byte loads in flight per thread (the loop is latency-bound otherwise: 3.1 TB/s with four)
    for (; r + 7 * step < r1; r += 8 * step) {
      uint4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x[(r + u * step) * tpr + cp];
#pragma unroll
      for (int u = 0; u < 8; ++u) accum(v[u]);
    }
    for (; r + 3 * step < r1; r += 4 * step) {
      const uint4 v0 = x[r * tpr + cp];
      const uint4 v1 = x[(r + step) * tpr + cp];
      const uint4 v2 = x[(r + 2 * step) * tpr + cp];
      const uint4 v3 = x[(r + 3 * step) * tpr + cp];
      accum(v0); accum(v1); accum(v2); accum(v3);
    }
    for (; r < r1; r += step) accum(x[r * tpr + cp]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[threadIdx.x * 16 + j] = s[j]; red[threadIdx.x * 16 + 8 + j] = q[j]; }
  __syncthreads();
  if (rsub == 0) {
    float* p = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float ss = 0.f, qq = 0.f;
      for (int u = 0; u < nsub; ++u) {
        ss += red[(u * tpr + cp) * 16 + j];
        qq += red[(u * tpr + cp) * 16 + 8 + j];
      }
      p[cp * 8 + j] = ss;
      p[C + cp * 8 + j] = qq;
    }
  }
}

// grid ceil(C / 16); 16 channels x 64 slab-groups per 1024-thread workgroup: the reduction over <= 1024 slabs is
// <= 16 dependent-latency rounds (it was 128 with 4 slab-groups, 35 us per BatchNorm layer)
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partial, int nslab, double rows, int C,
                                                           float eps, float momentum, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float* __restrict__ scale,
                                                           float* __restrict__ shift, int train) {
  __shared__ double rs[16][16], rq[16][16];
  const int cx = threadIdx.x & 15, sg = threadIdx.x >> 4, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 16 + cx;
  double s = 0.0, q = 0.0;
  if (train && c < C) {
    int i = sg;
    double s1 = 0.0, q1 = 0.0;
    for (; i + 64 < nslab; i += 128) {      // two independent load pairs in flight
      s += (double)partial[(size_t)i * 2 * C + c];
      q += (double)partial[(size_t)i * 2 * C + C + c];
      s1 += (double)partial[(size_t)(i + 64) * 2 * C + c];
      q1 += (double)partial[(size_t)(i + 64) * 2 * C + C + c];
    }
    if (i < nslab) {
      s += (double)partial[(size_t)i * 2 * C + c];
      q += (double)partial[(size_t)i * 2 * C + C + c];
    }
    s += s1;
    q += q1;
  }
  // the wave's four slab-groups live in lanes cx, cx + 16, cx + 32, cx + 48: row-swap sums, then 16 waves through LDS
  s = colsum4(s);
  q = colsum4(q);
  if ((threadIdx.x & 63) < 16) { rs[wave][cx] = s; rq[wave][cx] = q; }
  __syncthreads();
  if (sg != 0 || c >= C) return;
  double mean, var;
  if (train) {
    s = 0.0; q = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) { s += rs[u][cx]; q += rq[u][cx]; }
    mean = s / rows;
    var = q / rows - mean * mean;
    if (var < 0.0) var = 0.0;
    if (running_mean) {
      const double unb = rows > 1.0 ? var * rows / (rows - 1.0) : var;
      running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
  } else {
    mean = running_mean[c];
    var = running_var[c];
  }
  const float inv = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = (gamma ? gamma[c] : 1.f) * inv;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - (float)mean * sc;
}

__device__ __forceinline__ unsigned int pack2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  return __builtin_bit_cast(unsigned int, __builtin_convertvector((f2){a, b}, b2));
}

// CLIP input normalisation (model/utils/clip.py:19, act3d.py:364) + NCHW fp32 -> NHWC bf16 in one pass: 4 pixels per thread.
// y[n][h][w][c] = bf16((x[n][c][h][w] - mean[c]) / std[c])   (replaces sub, div, channels-last copy and cast: 4 passes)
__global__ __launch_bounds__(256) void rgb_normalize_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ stdv, unsigned short* __restrict__ y,
                                                            size_t N, size_t HW) {
  const size_t quads = N * HW / 4;
  const float m0 = mean[0], m1 = mean[1], m2 = mean[2], s0 = stdv[0], s1 = stdv[1], s2 = stdv[2];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = (i * 4) / HW, p = (i * 4) - n * HW;          // HW % 4 == 0: a quad never straddles images
    const float* base = x + n * 3 * HW + p;
    const float4 r = *reinterpret_cast<const float4*>(base);
    const float4 g = *reinterpret_cast<const float4*>(base + HW);
    const float4 b = *reinterpret_cast<const float4*>(base + 2 * HW);
    const float rr[4] = {r.x, r.y, r.z, r.w}, gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {b.x, b.y, b.z, b.w};
    unsigned short o[12];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[3 * j + 0] = f2bf((rr[j] - m0) / s0);
      o[3 * j + 1] = f2bf((gg[j] - m1) / s1);
      o[3 * j + 2] = f2bf((bb[j] - m2) / s2);
    }
    uint2* dst = reinterpret_cast<uint2*>(y + (n * HW + p) * 3);      // 24 bytes, 8-byte aligned (p % 4 == 0)
    dst[0] = make_uint2(o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16));
    dst[1] = make_uint2(o[4] | ((unsigned)o[5] << 16), o[6] | ((unsigned)o[7] << 16));
    dst[2] = make_uint2(o[8] | ((unsigned)o[9] << 16), o[10] | ((unsigned)o[11] << 16));
  }
}

// 8 channels (16 B) per thread
__global__ __launch_bounds__(256) void bn_apply_kernel(const uint4* __restrict__ x, const uint4* __restrict__ res,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       uint4* __restrict__ y, size_t nvec, int C8, int relu) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C8) * 8;
    const uint4 v = x[i];
    uint4 rv = make_uint4(0, 0, 0, 0);
    if (res) rv = res[i];
    const unsigned int vw[4] = {v.x, v.y, v.z, v.w};
    const unsigned int rw[4] = {rv.x, rv.y, rv.z, rv.w};
    unsigned int ow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = __uint_as_float(vw[j] << 16) * scale[c + 2 * j] + shift[c + 2 * j];
      float b = __uint_as_float(vw[j] & 0xFFFF0000u) * scale[c + 2 * j + 1] + shift[c + 2 * j + 1];
      if (res) {
        a += __uint_as_float(rw[j] << 16);
        b += __uint_as_float(rw[j] & 0xFFFF0000u);
      }
      if (relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
      ow[j] = pack2(a, b);
    }
    y[i] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
  }
}

// y = relu?(x * scale + shift (+ residual)) as bn_apply_kernel, plus the 2x2 average pool of y that follows the
// activation in the CLIP bottlenecks (AvgPool2d(stride) after bn2 / in the downsample branch) and the stem: each thread
// owns 8 channels of one POOLED pixel, evaluates its four source pixels, optionally writes them (y_full) and writes their
// mean.  scale == NULL means identity (plain average pool of x).  Saves the pool kernel's re-read of the full activation.
__global__ __launch_bounds__(256) void bn_apply_pool2_kernel(const uint4* __restrict__ x, const uint4* __restrict__ res,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             uint4* __restrict__ y_full, uint4* __restrict__ y_pool, int N,
                                                             int H, int W, int C8, int relu) {
  const int Ho = H >> 1, Wo = W >> 1;
  const size_t nout = (size_t)N * Ho * Wo * C8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nout; i += (size_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    size_t pix = i / C8;
    const int wo = (int)(pix % Wo);
    pix /= Wo;
    const int ho = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    const int c = c8 * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale ? scale[c + j] : 1.f; sh[j] = scale ? shift[c + j] : 0.f; }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const size_t src = (((size_t)n * H + 2 * ho + dy) * W + 2 * wo + dx) * C8 + c8;
        const uint4 v = x[src];
        uint4 rv = make_uint4(0, 0, 0, 0);
        if (res) rv = res[src];
        const unsigned int vw[4] = {v.x, v.y, v.z, v.w};
        const unsigned int rw[4] = {rv.x, rv.y, rv.z, rv.w};
        unsigned int ow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float a = __uint_as_float(vw[j] << 16) * sc[2 * j] + sh[2 * j];
          float b = __uint_as_float(vw[j] & 0xFFFF0000u) * sc[2 * j + 1] + sh[2 * j + 1];
          if (res) {
            a += __uint_as_float(rw[j] << 16);
            b += __uint_as_float(rw[j] & 0xFFFF0000u);
          }
          if (relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
          ow[j] = pack2(a, b);
          // pool what the next layer would have read: the bf16-rounded activation
          acc[2 * j] += __uint_as_float(ow[j] << 16);
          acc[2 * j + 1] += __uint_as_float(ow[j] & 0xFFFF0000u);
        }
        if (y_full) y_full[src] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
      }
    }
    y_pool[i] = make_uint4(pack2(acc[0] * 0.25f, acc[1] * 0.25f), pack2(acc[2] * 0.25f, acc[3] * 0.25f),
                           pack2(acc[4] * 0.25f, acc[5] * 0.25f), pack2(acc[6] * 0.25f, acc[7] * 0.25f));
  }
}

// FPN top-down step (torchvision FeaturePyramidNetwork: inner_lateral + F.interpolate(last_inner, nearest)) for an exact
// 2x upsampling, bf16 NHWC, C % 4 == 0 (8-byte vectors; the policy's C = 60 is not a multiple of 8):
//   fwd  y[n][h][w][:]  = lat[n][h][w][:] + top[n][h/2][w/2][:]
//   bwd  dtop[n][i][j][:] = sum of the 2x2 dy block (the lateral branch's gradient is dy itself)
__device__ __forceinline__ uint2 add_bf16x4(uint2 a, uint2 b) {
  return make_uint2(pack2(__uint_as_float(a.x << 16) + __uint_as_float(b.x << 16),
                          __uint_as_float(a.x & 0xFFFF0000u) + __uint_as_float(b.x & 0xFFFF0000u)),
                    pack2(__uint_as_float(a.y << 16) + __uint_as_float(b.y << 16),
                          __uint_as_float(a.y & 0xFFFF0000u) + __uint_as_float(b.y & 0xFFFF0000u)));
}
__global__ __launch_bounds__(256) void upsample2_add_fwd_kernel(const uint2* __restrict__ lat, const uint2* __restrict__ top,
                                                                uint2* __restrict__ y, int N, int H, int W, int C4) {
  const size_t total = (size_t)N * H * W * C4;
  const int Ht = H >> 1, Wt = W >> 1;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    size_t pix = i / C4;
    const int w = (int)(pix % W);
    pix /= W;
    const int h = (int)(pix % H);
    const int n = (int)(pix / H);
    y[i] = add_bf16x4(lat[i], top[(((size_t)n * Ht + (h >> 1)) * Wt + (w >> 1)) * C4 + c]);
  }
}
__global__ __launch_bounds__(256) void upsample2_add_bwd_kernel(const uint2* __restrict__ dy, uint2* __restrict__ dtop, int N,
                                                                int H, int W, int C4) {
  const int Ht = H >> 1, Wt = W >> 1;
  const size_t total = (size_t)N * Ht * Wt * C4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    size_t pix = i / C4;
    const int wt = (int)(pix % Wt);
    pix /= Wt;
    const int ht = (int)(pix % Ht);
    const int n = (int)(pix / Ht);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const uint2 v = dy[(((size_t)n * H + 2 * ht + dyy) * W + 2 * wt + dx) * C4 + c];
        acc[0] += __uint_as_float(v.x << 16);
        acc[1] += __uint_as_float(v.x & 0xFFFF0000u);
        acc[2] += __uint_as_float(v.y << 16);
        acc[3] += __uint_as_float(v.y & 0xFFFF0000u);
      }
    dtop[i] = make_uint2(pack2(acc[0], acc[1]), pack2(acc[2], acc[3]));
  }
}

}  // namespace a3d

using namespace a3d;

extern "C" int a3d_bn_nslab(size_t rows, int C) {
  // enough slabs to fill the chip with 4 workgroups per CU, at least 64 rows per slab
  size_t n = 1024;
  if (n > rows / 64) n = rows / 64;
  if (n < 1) n = 1;
  return (int)n;
}

extern "C" int a3d_bn_stats(const void* x, float* partial, size_t rows, int C, int nslab, void* stream) {
  if (!x || !partial || rows == 0 || C < 8 || (C % 8) != 0 || C > 2048 || (256 % (C / 8)) != 0 || nslab < 1 ||
      (((uintptr_t)x) & 15)) {
    set_error("a3d_bn_stats: bad argument (C=%d must be 8 * a divisor of 256, nslab=%d)", C, nslab);
    return A3D_ERR_ARG;
  }
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nslab), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, partial, rows, C,
                     nslab);
  return check_launch("a3d_bn_stats");
}

extern "C" int a3d_bn_finalize(const float* partial, int nslab, size_t rows, int C, float eps, float momentum,
                               const float* gamma, const float* beta, float* running_mean, float* running_var,
                               float* scale, float* shift, int train, void* stream) {
  if (C <= 0 || !scale || !shift || (train && !partial) || (!train && (!running_mean || !running_var))) {
    set_error("a3d_bn_finalize: bad argument");
    return A3D_ERR_ARG;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 16)), dim3(1024), 0, (hipStream_t)stream, partial, nslab, (double)rows,
                     C, eps, momentum, gamma, beta, running_mean, running_var, scale, shift, train);
  return check_launch("a3d_bn_finalize");
}

extern "C" int a3d_bn_apply(const void* x, const void* residual, const float* scale, const float* shift, void* y,
                            size_t rows, int C, int relu, void* stream) {
  if (!x || !scale || !shift || !y || rows == 0 || C <= 0 || (C % 8) != 0 ||
      ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)residual)) & 15)) {
    set_error("a3d_bn_apply: bad argument (C=%d must be a multiple of 8, pointers 16-byte aligned)", C);
    return A3D_ERR_ARG;
  }
  const size_t nvec = rows * (size_t)(C / 8);
  const int grid = (int)std::min<size_t>((nvec + 255) / 256, 16384);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (const uint4*)residual,
                     scale, shift, (uint4*)y, nvec, C / 8, relu);
  return check_launch("a3d_bn_apply");
}

extern "C" int a3d_bn_apply_pool2(const void* x, const void* residual, const float* scale, const float* shift, void* y_full,
                                  void* y_pool, int N, int H, int W, int C, int relu, void* stream) {
  if (!x || !y_pool || (scale && !shift) || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C % 8) != 0 ||
      ((((uintptr_t)x) | ((uintptr_t)y_pool) | ((uintptr_t)y_full) | ((uintptr_t)residual)) & 15)) {
    set_error("a3d_bn_apply_pool2: bad argument (H=%d W=%d must be even, C=%d a multiple of 8, pointers 16-byte aligned)", H,
              W, C);
    return A3D_ERR_ARG;
  }
  const size_t nout = (size_t)N * (H / 2) * (W / 2) * (C / 8);
  const int grid = (int)std::min<size_t>((nout + 255) / 256, 16384);
  hipLaunchKernelGGL(bn_apply_pool2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)x,
                     (const uint4*)residual, scale, shift, (uint4*)y_full, (uint4*)y_pool, N, H, W, C / 8, relu);
  return check_launch("a3d_bn_apply_pool2");
}

static int check_up2(const char* fn, const void* a, const void* b, const void* c, int N, int H, int W, int C) {
  if (!a || !b || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C % 4) != 0 ||
      ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 7)) {
    set_error("%s: bad argument (H=%d W=%d must be even, C=%d a multiple of 4, pointers 8-byte aligned)", fn, H, W, C);
    return A3D_ERR_ARG;
  }
  return A3D_OK;
}

extern "C" int a3d_upsample2_add_fwd(const void* lat, const void* top, void* y, int N, int H, int W, int C, void* stream) {
  int rc = check_up2("a3d_upsample2_add_fwd", lat, top, y, N, H, W, C);
  if (rc || !y) { if (!rc) set_error("a3d_upsample2_add_fwd: null output"); return A3D_ERR_ARG; }
  const size_t total = (size_t)N * H * W * (C / 4);
  const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(upsample2_add_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint2*)lat,
                     (const uint2*)top, (uint2*)y, N, H, W, C / 4);
  return check_launch("a3d_upsample2_add_fwd");
}

extern "C" int a3d_upsample2_add_bwd(const void* dy, void* dtop, int N, int H, int W, int C, void* stream) {
  int rc = check_up2("a3d_upsample2_add_bwd", dy, dtop, nullptr, N, H, W, C);
  if (rc) return rc;
  const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
  const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(upsample2_add_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint2*)dy, (uint2*)dtop,
                     N, H, W, C / 4);
  return check_launch("a3d_upsample2_add_bwd");
}

extern "C" int a3d_rgb_normalize_nhwc_bf16(const float* x, const float* mean, const float* stdv, void* y, size_t N, int H, int W,
                                           void* stream) {
  const size_t HW = (size_t)H * W;
  if (!x || !mean || !stdv || !y || N == 0 || H <= 0 || W <= 0 || (HW % 4) != 0 || ((((uintptr_t)x) & 15) != 0) ||
      ((((uintptr_t)y) & 7) != 0)) {
    set_error("a3d_rgb_normalize_nhwc_bf16: bad argument (H * W must be a multiple of 4, x 16-byte aligned)");
    return A3D_ERR_ARG;
  }
  const size_t quads = N * HW / 4;
  hipLaunchKernelGGL(rgb_normalize_kernel, dim3((int)std::min<size_t>((quads + 255) / 256, 16384)), dim3(256), 0,
                     (hipStream_t)stream, x, mean, stdv, (unsigned short*)y, N, HW);
  return check_launch("a3d_rgb_normalize_nhwc_bf16");
}
