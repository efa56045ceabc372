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

// grid (nslab, C / CG): a workgroup owns a group of CG = min(C, 256) channels; each thread owns 8 consecutive channels (one
// 16-byte load per row), tpr = CG / 8 threads span a row segment and the nsub = 256 / tpr row-groups of the workgroup cover
// a CHUNK of 8 * nsub rows per iteration (eight independent 16-byte loads per thread; the next chunk's loads are issued before
// the current one is accumulated).  Chunks are dealt to the slabs round-robin (slab i takes chunks i, i + nslab, ...): at any
// moment the workgroups of the grid read one contiguous stretch of the activation.  Round 3's contiguous slabs started
// every workgroup at a multiple of rows / nslab * C * 2 bytes (512 KB for the layer-1 maps) and walked in lockstep -- the
// same HBM channels for the whole grid, 3.1 TB/s against bn_apply's 6.1 on the same tensors.  The assignment is fixed, so
// the sums are reproducible run to run.  An LDS tree adds the row-groups.  C % 8 == 0, C / 8 divides 256 or C % 256 == 0.
__global__ __launch_bounds__(256) void bn_stats_kernel(const uint4* __restrict__ x, float* __restrict__ partial,
                                                       size_t rows, int C, int CG, int nslab) {
  __shared__ float red[256 * 16];
  const int tpr = CG >> 3;                 // threads per row segment
  const int nsub = 256 / tpr;              // row-groups per workgroup
  const int cp = threadIdx.x % tpr, rsub = threadIdx.x / tpr;
  const int C8 = C >> 3;
  const int cvec = blockIdx.y * tpr + cp;  // this thread's 16-byte column of a row
  const size_t chunk_rows = (size_t)nsub * 8;
  const size_t nchunk = (rows + chunk_rows - 1) / chunk_rows;
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
  auto accum = [&](const uint4& v) {
    const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = __uint_as_float(w[j] << 16), b = __uint_as_float(w[j] & 0xFFFF0000u);
      s[2 * j] += a; q[2 * j] += a * a;
      s[2 * j + 1] += b; q[2 * j + 1] += b * b;
    }
  };
  auto load = [&](size_t ch, uint4 (&v)[8]) {
    const size_t r = ch * chunk_rows + rsub;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t ru = r + (size_t)u * nsub;
      const uint4 t = x[(ru < rows ? ru : rows - 1) * C8 + cvec];        // clamped address + select: no branch around the load
      v[u] = ru < rows ? t : make_uint4(0, 0, 0, 0);
    }
  };
  uint4 cur[8], nxt[8];
  size_t ch = blockIdx.x;
  if (ch < nchunk) load(ch, cur);
  while (ch < nchunk) {
    const size_t nch = ch + nslab;
    if (nch < nchunk) load(nch, nxt);
#pragma unroll
    for (int u = 0; u < 8; ++u) accum(cur[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
    ch = nch;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[threadIdx.x * 16 + j] = s[j]; red[threadIdx.x * 16 + 8 + j] = q[j]; }
  __syncthreads();
  // 2 * CG sums per workgroup: thread t < 2 * CG adds the nsub row-groups of (statistic t / CG, channel t % CG)
  for (int o = threadIdx.x; o < 2 * CG; o += 256) {
    const int which = o / CG, c = o - which * CG;
    const int tp = c >> 3, j = c & 7;
    float acc = 0.f;
    for (int u = 0; u < nsub; ++u) acc += red[(u * tpr + tp) * 16 + which * 8 + j];
    partial[(size_t)blockIdx.x * 2 * C + (size_t)which * C + blockIdx.y * CG + c] = acc;
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
// RBN: the residual is itself a raw convolution output with its own BatchNorm (the bottleneck's downsample branch,
// clip.py:28-43: identity = downsample(x) = bn(conv(avgpool(x)))): y = relu?(x * scale + shift + (res * rscale + rshift)) -- the
// branch's normalised map is never materialised (one read + one write of a 4 * planes map less per layer)
template <bool RBN>
__global__ __launch_bounds__(256) void bn_apply_kernel(const uint4* __restrict__ x, const uint4* __restrict__ res,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                       uint4* __restrict__ y, size_t nvec, int C8, int relu) {
  // C8 = C / 8 is a power of two for every BatchNorm of the ResNet (C = 32 .. 2048): mask instead of a 64-bit modulo per element
  const bool pow2 = (C8 & (C8 - 1)) == 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (pow2 ? (int)((unsigned int)i & (unsigned int)(C8 - 1)) : (int)(i % C8)) * 8;
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
        float ra = __uint_as_float(rw[j] << 16), rb = __uint_as_float(rw[j] & 0xFFFF0000u);
        if (RBN) {
          ra = ra * rscale[c + 2 * j] + rshift[c + 2 * j];
          rb = rb * rscale[c + 2 * j + 1] + rshift[c + 2 * j + 1];
        }
        a += ra;
        b += rb;
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
  // 32-bit index arithmetic (nout < 2^31: checked by the host), shifts / masks when C8 and Wo are powers of two (always, for the
  // ResNet's maps): the five 64-bit divisions per element this replaces were ~500 instructions for 4 loads and <= 5 stores
  const bool p2 = (C8 & (C8 - 1)) == 0 && (Wo & (Wo - 1)) == 0;
  const int c8_sh = 31 - __clz(C8), wo_sh = 31 - __clz(Wo);
  for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned int)nout; i += gridDim.x * blockDim.x) {
    int c8, wo;
    unsigned int pix;
    if (p2) {
      c8 = (int)(i & (unsigned int)(C8 - 1));
      pix = i >> c8_sh;
      wo = (int)(pix & (unsigned int)(Wo - 1));
      pix >>= wo_sh;
    } else {
      c8 = (int)(i % (unsigned int)C8);
      pix = i / (unsigned int)C8;
      wo = (int)(pix % (unsigned int)Wo);
      pix /= (unsigned int)Wo;
    }
    const int ho = (int)(pix % (unsigned int)Ho);
    const int n = (int)(pix / (unsigned int)Ho);
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
// 2x upsampling, bf16 NHWC, C % 4 == 0 (8-byte vectors; the policy's C = 60 is not a multiple of 8), with the lateral 1x1
// convolution's BIAS folded in (the convolution itself then runs bias-free: no separate bias-add pass over the map forward,
// no separate reduction pass over its gradient backward -- 0.5 + 0.7 ms of torch kernels per step at B = 64):
//   fwd  y[n][h][w][:]  = lat[n][h][w][:] + bias[:] + top[n][h/2][w/2][:]           (one rounding; bias / top may be null)
//   bwd  dtop[n][i][j][:] = sum of the 2x2 dy block (the lateral branch's gradient is dy itself);
//        dbias partial[blk][:] = the workgroup's column sums of dy (fixed order; a3d_colsum_reduce adds them)
__device__ __forceinline__ void bf16x4_to_f32(uint2 a, float (&o)[4]) {
  o[0] = __uint_as_float(a.x << 16); o[1] = __uint_as_float(a.x & 0xFFFF0000u);
  o[2] = __uint_as_float(a.y << 16); o[3] = __uint_as_float(a.y & 0xFFFF0000u);
}
__global__ __launch_bounds__(256) void upsample2_add_fwd_kernel(const uint2* __restrict__ lat, const uint2* __restrict__ top,
                                                                const float* __restrict__ bias, int nbias,
                                                                uint2* __restrict__ y, int N, int H, int W, int C4) {
  const size_t total = (size_t)N * H * W * C4;
  const int Ht = H >> 1, Wt = W >> 1;
  // shifts / masks when C4, W and H are powers of two (the policy's maps: 64 padded channels, 128 .. 8 pixels a side): the
  // generic path's five 64-bit divisions per 8-byte element made this pass VALU-bound (~200 instructions per element)
  const bool p2 = (C4 & (C4 - 1)) == 0 && (W & (W - 1)) == 0 && (H & (H - 1)) == 0;
  const int c_sh = 31 - __clz(C4), w_sh = 31 - __clz(W), h_sh = 31 - __clz(H);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c, w, h, n;
    if (p2) {
      c = (int)((unsigned int)i & (unsigned int)(C4 - 1));
      const size_t pix = i >> c_sh;
      w = (int)((unsigned int)pix & (unsigned int)(W - 1));
      h = (int)((unsigned int)(pix >> w_sh) & (unsigned int)(H - 1));
      n = (int)(pix >> (w_sh + h_sh));
    } else {
      c = (int)(i % C4);
      size_t pix = i / C4;
      w = (int)(pix % W);
      pix /= W;
      h = (int)(pix % H);
      n = (int)(pix / H);
    }
    float a[4], t[4] = {0.f, 0.f, 0.f, 0.f};
    bf16x4_to_f32(lat[i], a);
    if (top) bf16x4_to_f32(top[(((size_t)n * Ht + (h >> 1)) * Wt + (w >> 1)) * C4 + c], t);
    if (bias) {
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] += (c * 4 + j < nbias) ? bias[c * 4 + j] : 0.f;      // pad channels carry no bias
    }
    y[i] = make_uint2(pack2(a[0] + t[0], a[1] + t[1]), pack2(a[2] + t[2], a[3] + t[3]));
  }
}
// grid-stride over POOLED pixels x channel quads; with 256 % C4 == 0 a thread keeps its channel quad for the whole loop, so
// the column sums of dy are per-thread registers + one LDS tree per workgroup (bias_partial [gridDim.x][4 * C4], or null)
__global__ __launch_bounds__(256) void upsample2_add_bwd_kernel(const uint2* __restrict__ dy, uint2* __restrict__ dtop,
                                                                float* __restrict__ bias_partial, int N, int H, int W, int C4) {
  __shared__ float red[256 * 4];
  const int Ht = H >> 1, Wt = W >> 1;
  const size_t total = (size_t)N * Ht * Wt * C4;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  const bool p2 = (C4 & (C4 - 1)) == 0 && (Wt & (Wt - 1)) == 0 && (Ht & (Ht - 1)) == 0;      // as in the forward kernel
  const int c_sh = 31 - __clz(C4), w_sh = 31 - __clz(Wt), h_sh = 31 - __clz(Ht);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    int c, wt, ht, n;
    if (p2) {
      c = (int)((unsigned int)i & (unsigned int)(C4 - 1));
      const size_t pix = i >> c_sh;
      wt = (int)((unsigned int)pix & (unsigned int)(Wt - 1));
      ht = (int)((unsigned int)(pix >> w_sh) & (unsigned int)(Ht - 1));
      n = (int)(pix >> (w_sh + h_sh));
    } else {
      c = (int)(i % C4);
      size_t pix = i / C4;
      wt = (int)(pix % Wt);
      pix /= Wt;
      ht = (int)(pix % Ht);
      n = (int)(pix / Ht);
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float v[4];
        bf16x4_to_f32(dy[(((size_t)n * H + 2 * ht + dyy) * W + 2 * wt + dx) * C4 + c], v);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += v[j];
      }
    if (dtop) dtop[i] = make_uint2(pack2(acc[0], acc[1]), pack2(acc[2], acc[3]));
#pragma unroll
    for (int j = 0; j < 4; ++j) cs[j] += acc[j];
  }
  if (bias_partial) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[threadIdx.x * 4 + j] = cs[j];
    __syncthreads();
    for (int o = threadIdx.x; o < 4 * C4; o += 256) {          // C4 up to 256 (C = 1024): more outputs than threads
      const int c = o >> 2, j = o & 3;
      float a = 0.f;
      for (int u = c; u < 256; u += C4) a += red[u * 4 + j];
      bias_partial[(size_t)blockIdx.x * 4 * C4 + o] = a;
    }
  }
}

// out[c] += sum_i partial[i][c] in a fixed order: 64 columns x 16 row groups per 1024-thread workgroup, eight independent
// loads in flight per thread, LDS tree over the row groups (rows <= a few thousand; one thread per column took 145 us for
// 4096 rows -- a chain of exposed L2 round trips)
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* __restrict__ partial, int nrow, int C, float* __restrict__ out,
                                                            int nout) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = 0.f;
  if (c < C) {
    int i = rg;
    for (; i + 7 * 16 < nrow; i += 8 * 16) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += partial[(size_t)(i + u * 16) * C + c];
    }
    for (; i < nrow; i += 16) a[0] += partial[(size_t)i * C + c];
  }
  red[rg][cl] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (rg == 0 && c < nout) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) s += red[u][cl];
    out[c] += s;
  }
}

// column sums of the fp32 rows a gather's backward sees: rows (b, s), s < k, of src [B][S][ld] -> partial [gridDim.x][C];
// blockIdx.y walks 64-column blocks (any C)
__global__ __launch_bounds__(256) void colsum_rows_kernel(const float* __restrict__ src, int B, int S, int k, int ld, int C,
                                                          float* __restrict__ partial) {
  __shared__ float red[256];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;           // 64 columns x 4 row groups
  const int c = blockIdx.y * 64 + cl;
  const size_t total = (size_t)B * k;
  float a = 0.f;
  if (c < C) {
    for (size_t r = (size_t)blockIdx.x * 4 + rg; r < total; r += (size_t)gridDim.x * 4) {
      const size_t b = r / k, s = r - b * k;
      a += src[(b * S + s) * ld + c];
    }
  }
  red[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x < 64 && c < C)
    partial[(size_t)blockIdx.x * C + c] = (red[cl] + red[64 + cl]) + (red[128 + cl] + red[192 + cl]);
}

}  // namespace a3d

using namespace a3d;

static int bn_channel_group(int C) { return C < 256 ? C : 256; }

extern "C" int a3d_bn_nslab(size_t rows, int C) {
  // ~2048 workgroups over (slab, 256-channel group) -- 8 per CU -- with at least one 8-row-per-thread chunk per slab, and at
  // most 1024 slabs (a3d_bn_finalize reads nslab x 2 x C partial sums)
  if (C < 8) return 1;
  const int CG = bn_channel_group(C);
  const size_t chunk_rows = (size_t)(256 / (CG / 8)) * 8;
  size_t n = 2048 / (size_t)(C / CG);
  const size_t nchunk = (rows + chunk_rows - 1) / chunk_rows;
  if (n > nchunk) n = nchunk;
  if (n > 1024) n = 1024;
  if (n < 1) n = 1;
  return (int)n;
}

extern "C" int a3d_bn_stats(const void* x, float* partial, size_t rows, int C, int nslab, void* stream) {
  const int CG = bn_channel_group(C);
  if (!x || !partial || rows == 0 || C < 8 || (C % 8) != 0 || C > 2048 || (256 % (CG / 8)) != 0 || (C % CG) != 0 || nslab < 1 ||
      (((uintptr_t)x) & 15)) {
    set_error("a3d_bn_stats: bad argument (C=%d must be 8 * a divisor of 256 or a multiple of 256, nslab=%d)", C, nslab);
    return A3D_ERR_ARG;
  }
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nslab, C / CG), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, partial, rows, C, CG,
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

// Workgroups of the BatchNorm-apply passes (grid-stride loops): 16384 by default.  A smaller grid leaves wave slots to the kernels of
// a concurrent stream: the backbone that engine.GraphedStep(prefetch=...) runs next to the hot path is captured with 256 (one
// workgroup per CU) -- 18.87 -> 18.45 ms per keypose step, while the same grid costs the stand-alone backbone ~0.1 ms
// (profiles/r06_prefetch_ab.json).  A3D_BN_GRID sets the process default; a3d_bn_grid_cap(cap) sets it (cap > 0) and returns the previous value.
static int g_bn_grid_cap = getenv("A3D_BN_GRID") ? std::max(64, atoi(getenv("A3D_BN_GRID"))) : 16384;
static int bn_grid_cap() { return g_bn_grid_cap; }
extern "C" int a3d_bn_grid_cap(int cap) {
  const int prev = g_bn_grid_cap;
  if (cap > 0) g_bn_grid_cap = std::max(64, cap);
  return prev;
}

extern "C" int a3d_bn_apply(const void* x, const void* residual, const float* res_scale, const float* res_shift, const float* scale,
                            const float* shift, void* y, size_t rows, int C, int relu, void* stream) {
  if (!x || !scale || !shift || !y || rows == 0 || C <= 0 || (C % 8) != 0 || (res_scale && (!res_shift || !residual)) ||
      ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)residual)) & 15)) {
    set_error("a3d_bn_apply: bad argument (C=%d must be a multiple of 8, pointers 16-byte aligned, res_scale needs res_shift and a residual)", C);
    return A3D_ERR_ARG;
  }
  const size_t nvec = rows * (size_t)(C / 8);
  const int grid = (int)std::min<size_t>((nvec + 255) / 256, (size_t)bn_grid_cap());
  if (res_scale)
    hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (const uint4*)residual,
                       scale, shift, res_scale, res_shift, (uint4*)y, nvec, C / 8, relu);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (const uint4*)residual,
                       scale, shift, (const float*)nullptr, (const float*)nullptr, (uint4*)y, nvec, C / 8, relu);
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
  if (nout >= ((size_t)1 << 31)) { set_error("a3d_bn_apply_pool2: map too large (%zu 16-byte outputs)", nout); return A3D_ERR_ARG; }
  const int grid = (int)std::min<size_t>((nout + 255) / 256, (size_t)bn_grid_cap());
  hipLaunchKernelGGL(bn_apply_pool2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)x,
                     (const uint4*)residual, scale, shift, (uint4*)y_full, (uint4*)y_pool, N, H, W, C / 8, relu);
  return check_launch("a3d_bn_apply_pool2");
}

static int check_up2(const char* fn, const void* a, const void* b, const void* c /* may be null */, int N, int H, int W, int C) {
  if (!a || !b || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C % 4) != 0 ||
      ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 7)) {
    set_error("%s: bad argument (H=%d W=%d must be even, C=%d a multiple of 4, pointers 8-byte aligned)", fn, H, W, C);
    return A3D_ERR_ARG;
  }
  return A3D_OK;
}

extern "C" int a3d_upsample2_add_fwd(const void* lat, const void* top, const float* bias, int nbias, void* y, int N, int H, int W,
                                     int C, void* stream) {
  int rc = check_up2("a3d_upsample2_add_fwd", lat, y, top, N, H, W, C);
  if (rc) return rc;
  if (bias && (nbias <= 0 || nbias > C)) { set_error("a3d_upsample2_add_fwd: %d bias entries for %d channels", nbias, C); return A3D_ERR_ARG; }
  const size_t total = (size_t)N * H * W * (C / 4);
  const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(upsample2_add_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint2*)lat,
                     (const uint2*)top, bias, nbias, (uint2*)y, N, H, W, C / 4);
  return check_launch("a3d_upsample2_add_fwd");
}

static int up2_bwd_grid(int N, int H, int W, int C) {
  const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
  return (int)std::min<size_t>((total + 255) / 256, 2048);
}
extern "C" size_t a3d_upsample2_add_bwd_ws_floats(int N, int H, int W, int C) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
  return (size_t)up2_bwd_grid(N, H, W, C) * C;
}

extern "C" int a3d_upsample2_add_bwd(const void* dy, void* dtop, float* dbias, int nbias, float* ws, int N, int H, int W, int C,
                                     void* stream) {
  int rc = check_up2("a3d_upsample2_add_bwd", dy, dy, dtop, N, H, W, C);
  if (rc) return rc;
  if (!dtop && !dbias) { set_error("a3d_upsample2_add_bwd: nothing to compute (dtop and dbias are both null)"); return A3D_ERR_ARG; }
  if (dbias && (!ws || (256 % (C / 4)) != 0 || nbias <= 0 || nbias > C)) {
    set_error("a3d_upsample2_add_bwd: the bias gradient needs a workspace, C / 4 = %d dividing 256 and 0 < nbias <= C", C / 4);
    return A3D_ERR_ARG;
  }
  const int grid = up2_bwd_grid(N, H, W, C);
  hipLaunchKernelGGL(upsample2_add_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint2*)dy, (uint2*)dtop,
                     dbias ? ws : nullptr, N, H, W, C / 4);
  rc = check_launch("a3d_upsample2_add_bwd");
  if (rc || !dbias) return rc;
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, (hipStream_t)stream, ws, grid, C, dbias, nbias);
  return check_launch("a3d_upsample2_add_bwd(bias)");
}

extern "C" size_t a3d_colsum_rows_ws_floats(int B, int k, int C) {
  if (B <= 0 || k <= 0 || C <= 0) return 0;
  const size_t nblk = std::min<size_t>(((size_t)B * k + 63) / 64, 1024);
  return nblk * C;
}

extern "C" int a3d_colsum_rows(const float* src, int B, int S, int k, int ld, int C, float* out, int nout, float* ws, void* stream) {
  if (!src || !out || !ws || B <= 0 || S <= 0 || k <= 0 || k > S || C <= 0 || ld < C || nout <= 0 || nout > C) {
    set_error("a3d_colsum_rows: bad argument (B=%d S=%d k=%d ld=%d C=%d nout=%d; k <= S, nout <= C <= ld)", B, S, k, ld, C, nout);
    return A3D_ERR_ARG;
  }
  const int nblk = (int)(a3d_colsum_rows_ws_floats(B, k, C) / C);
  hipLaunchKernelGGL(colsum_rows_kernel, dim3(nblk, cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, src, B, S, k, ld, C, ws);
  int rc = check_launch("a3d_colsum_rows");
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, (hipStream_t)stream, ws, nblk, C, out, nout);
  return check_launch("a3d_colsum_rows(reduce)");
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
