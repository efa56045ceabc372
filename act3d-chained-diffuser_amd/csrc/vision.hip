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
    // four independent 16-byte loads in flight per thread (the loop is latency-bound otherwise)
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
  __shared__ double rs[64][17], rq[64][17];
  const int cx = threadIdx.x & 15, sg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cx;
  double s = 0.0, q = 0.0;
  if (train && c < C) {
    for (int i = sg; i < nslab; i += 64) {
      s += (double)partial[(size_t)i * 2 * C + c];
      q += (double)partial[(size_t)i * 2 * C + C + c];
    }
  }
  rs[sg][cx] = s;
  rq[sg][cx] = q;
  __syncthreads();
  if (sg != 0 || c >= C) return;
  double mean, var;
  if (train) {
    s = 0.0; q = 0.0;
    for (int u = 0; u < 64; ++u) { s += rs[u][cx]; q += rq[u][cx]; }
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
