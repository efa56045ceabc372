// Scene-token construction for the Act3D coarse-to-fine loop: HBM-bound gather/select kernels.
//
//   a3d_pcd_downsample   reference: F.interpolate(pcd, 1/f, 'bilinear') + rearrange (act3d.py:379-383,
//                        encoder.py:147-158).  For even f the bilinear sample is the mean of the 2x2 block at
//                        (f*y + f/2 - 1, f*x + f/2 - 1); the additions are ordered as ATen's CPU kernels order them
//                        (see the kernel) so the result is bit-identical to the reference run on CPU.
//   a3d_knn_topk         reference: l2 = ((pos - pcd)**2).sum(-1).sqrt(); topk(k, largest=False).indices
//                        (act3d.py:244-245).  Radix-select on the fp32 bit pattern + bitonic sort of the k
//                        survivors in LDS; order = ascending (distance, index), i.e. torch's sorted order with a
//                        defined tie-break.
//   a3d_build_context    reference: per-sample python gathers + torch.cat with the gripper token
//                        (act3d.py:247-260).  One pass: ctx[b] = [feat[b][idx[b]] | extra[b]].
//   a3d_build_context_bwd scatter of d(ctx) back to the (zero-initialised) feature-map gradient.
#include "a3d_common.h"
#include "../../include/act3d_hip.h"

namespace a3d {

__global__ __launch_bounds__(256) void pcd_downsample_kernel(
    const float* __restrict__ pcd, float* __restrict__ out, int BC, int C, int Hin, int Win, int f) {
  const int h = Hin / f, w = Win / f;
  const size_t total = (size_t)BC * h * w;
  const int off = f / 2 - 1;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % w);
    const int y = (int)((idx / w) % h);
    const int bc = (int)(idx / ((size_t)w * h));
    const int y0 = f * y + off, x0 = f * x + off;
    const int y1 = min(y0 + 1, Hin - 1), x1 = min(x0 + 1, Win - 1);
    const int b = bc / C, cam = bc - b * C;
    float* o = out + (((size_t)b * C + cam) * h * w + (size_t)y * w + x) * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float* p = pcd + ((size_t)bc * 3 + ch) * Hin * Win;
      const float p00 = p[(size_t)y0 * Win + x0], p01 = p[(size_t)y0 * Win + x1];
      const float p10 = p[(size_t)y1 * Win + x0], p11 = p[(size_t)y1 * Win + x1];
      // Bit-compatibility with the reference's CPU run: ATen evaluates w00*a + w01*b + w10*c + w11*d left to
      // right when out_h + out_w <= 128 (its "vectorized" kernel) and separably, rows first, otherwise
      // (aten/src/ATen/native/cpu/UpSampleKernel.cpp, _use_vectorized_kernel_cond_2d).  All weights are powers of
      // two, so every product is exact and only the order of the additions matters.
      if (h + w <= 128) {
        o[ch] = add_rn(add_rn(add_rn(0.25f * p00, 0.25f * p01), 0.25f * p10), 0.25f * p11);
      } else {
        const float t0 = add_rn(0.5f * p00, 0.5f * p01);
        const float t1 = add_rn(0.5f * p10, 0.5f * p11);
        o[ch] = add_rn(0.5f * t0, 0.5f * t1);
      }
    }
  }
}

// d = sqrt((px-x)^2 + (py-y)^2 + (pz-z)^2) in the reference's operation order, no fma contraction
__device__ __forceinline__ float l2_dist(float px, float py, float pz, const float* q) {
  const float dx = sub_rn(px, q[0]), dy = sub_rn(py, q[1]), dz = sub_rn(pz, q[2]);
  const float s = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
  return sqrt_rn(s);
}

// distance of a scene point to the nearest of `npos` query positions (npos = 1: the Act3D k-NN centre, act3d.py:247-248;
// npos = L: find_traj_nn of the multi-scale diffusion head, model/utils/utils.py:39-48, which ranks SQUARED distances)
__device__ __forceinline__ float nn_dist(const float* __restrict__ pos, int npos, int squared, const float* q) {
  float best = 0.f;
  for (int l = 0; l < npos; ++l) {
    const float dx = sub_rn(pos[l * 3 + 0], q[0]), dy = sub_rn(pos[l * 3 + 1], q[1]), dz = sub_rn(pos[l * 3 + 2], q[2]);
    const float s = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
    const float d = squared ? s : sqrt_rn(s);
    best = (l == 0) ? d : fminf(best, d);
  }
  return best;
}

constexpr int TK_THREADS = 1024;

// One workgroup per sample.  dist_ws: [B][N] uint32 scratch.  keys: dynamic LDS, kpad uint64.
__global__ __launch_bounds__(TK_THREADS) void knn_topk_kernel(
    const float* __restrict__ pos, const float* __restrict__ xyz, unsigned int* __restrict__ dist_ws,
    long long* __restrict__ idx_out, float* __restrict__ dist_out, int N, int k, int kpad, int npos, int squared) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  __shared__ unsigned int hist[256];
  __shared__ unsigned int sh_prefix, sh_krem, sh_count, sh_neq;
  const int b = blockIdx.x, t = threadIdx.x;
  const float* qp = pos + (size_t)b * npos * 3;
  const float* pts = xyz + (size_t)b * N * 3;
  unsigned int* dw = dist_ws + (size_t)b * N;

  for (int i = t; i < N; i += TK_THREADS) dw[i] = __float_as_uint(nn_dist(qp, npos, squared, pts + (size_t)i * 3));
  if (t == 0) { sh_prefix = 0; sh_krem = (unsigned int)k; }
  __syncthreads();

  // ---- 4 radix passes over the value bits (distances are >= 0: bit pattern is monotonic)
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (t < 256) hist[t] = 0;
    __syncthreads();
    const unsigned int prefix = sh_prefix;
    const unsigned int himask = (pass == 0) ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = t; i < N; i += TK_THREADS) {
      const unsigned int v = dw[i];
      if ((v & himask) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (t == 0) {
      unsigned int krem = sh_krem, cum = 0;
      int bin = 0;
      for (; bin < 256; ++bin) {
        if (cum + hist[bin] >= krem) break;
        cum += hist[bin];
      }
      if (bin > 255) bin = 255;
      sh_krem = krem - cum;
      sh_prefix = prefix | ((unsigned int)bin << shift);
      sh_neq = hist[bin];
    }
    __syncthreads();
  }
  const unsigned int T = sh_prefix;       // k-th smallest value
  unsigned int need = sh_krem;            // how many elements == T are wanted
  unsigned int idx_thr = 0xFFFFFFFFu;     // elements == T with idx <= idx_thr are taken
  if (sh_neq != need) {
    // ties at the threshold: select the `need` lowest indices among bits == T (3 radix passes on idx)
    __syncthreads();
    if (t == 0) { sh_prefix = 0; sh_krem = need; }
    __syncthreads();
    for (int pass = 0; pass < 3; ++pass) {
      const int shift = 16 - 8 * pass;
      if (t < 256) hist[t] = 0;
      __syncthreads();
      const unsigned int prefix = sh_prefix;
      const unsigned int himask = (pass == 0) ? 0u : (0xFFFFFFFFu << (shift + 8));
      for (int i = t; i < N; i += TK_THREADS) {
        if (dw[i] == T && (((unsigned int)i) & himask) == prefix) atomicAdd(&hist[(((unsigned int)i) >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (t == 0) {
        unsigned int krem = sh_krem, cum = 0;
        int bin = 0;
        for (; bin < 256; ++bin) {
          if (cum + hist[bin] >= krem) break;
          cum += hist[bin];
        }
        if (bin > 255) bin = 255;
        sh_krem = krem - cum;
        sh_prefix = prefix | ((unsigned int)bin << shift);
      }
      __syncthreads();
    }
    idx_thr = sh_prefix;
  }
  // ---- collect survivors
  if (t == 0) sh_count = 0;
  for (int i = t; i < kpad; i += TK_THREADS) keys[i] = 0xFFFFFFFFFFFFFFFFull;
  __syncthreads();
  for (int i = t; i < N; i += TK_THREADS) {
    const unsigned int v = dw[i];
    if (v < T || (v == T && (unsigned int)i <= idx_thr)) {
      const unsigned int slot = atomicAdd(&sh_count, 1u);
      if (slot < (unsigned int)kpad) keys[slot] = ((unsigned long long)v << 32) | (unsigned int)i;
    }
  }
  __syncthreads();
  // ---- bitonic sort of kpad 64-bit keys (value major, index minor)
  for (int size = 2; size <= kpad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = t; i < (kpad >> 1); i += TK_THREADS) {
        const int lo = (i / stride) * (stride << 1) + (i % stride);
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const unsigned long long a = keys[lo], c = keys[hi];
        if ((a > c) == up) { keys[lo] = c; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = t; i < k; i += TK_THREADS) {
    const unsigned long long kk = keys[i];
    idx_out[(size_t)b * k + i] = (long long)(kk & 0xFFFFFFFFull);
    if (dist_out) dist_out[(size_t)b * k + i] = __uint_as_float((unsigned int)(kk >> 32));
  }
}

// hist[bin] += 1 for every lane with pred, with the wave's most common case -- many lanes in the same bin (distances
// share their exponent bits) -- folded into one LDS atomic: up to two rounds of "leader's bin" peeling, then plain atomics.
__device__ __forceinline__ void hist_add_wave(unsigned int* hist, unsigned int bin, bool pred, int lane) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const unsigned long long act = __ballot(pred);
    if (act == 0ull) return;
    const int leader = __ffsll((long long)act) - 1;
    const unsigned int lb = __builtin_amdgcn_readlane(bin, leader);
    const unsigned long long same = __ballot(pred && bin == lb);
    if (lane == leader) atomicAdd(&hist[lb], (unsigned int)__popcll(same));
    pred = pred && (bin != lb);
  }
  if (pred) atomicAdd(&hist[bin], 1u);
}

// Same selection as knn_topk_kernel with each thread's ITEMS distances held in registers (N <= 1024 * ITEMS): no scratch
// round trips through L2 in the seven passes, wave-aggregated histogram atomics.
template <int ITEMS>
__global__ __launch_bounds__(TK_THREADS) void knn_topk_reg_kernel(
    const float* __restrict__ pos, const float* __restrict__ xyz, long long* __restrict__ idx_out,
    float* __restrict__ dist_out, int N, int k, int kpad, int npos, int squared) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  __shared__ unsigned int hist[256];
  __shared__ unsigned int sh_prefix, sh_krem, sh_count, sh_neq;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63;
  const float* qp = pos + (size_t)b * npos * 3;
  const float* pts = xyz + (size_t)b * N * 3;

  unsigned int dv[ITEMS];     // element i of this thread is point t + i * 1024 (0xFFFFFFFF beyond N: never selected, k <= N)
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int n = t + i * TK_THREADS;
    dv[i] = (n < N) ? __float_as_uint(nn_dist(qp, npos, squared, pts + (size_t)n * 3)) : 0xFFFFFFFFu;
  }
  if (t == 0) { sh_prefix = 0; sh_krem = (unsigned int)k; }
  __syncthreads();

  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (t < 256) hist[t] = 0;
    __syncthreads();
    const unsigned int prefix = sh_prefix;
    const unsigned int himask = (pass == 0) ? 0u : (0xFFFFFFFFu << (shift + 8));
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const bool in = (t + i * TK_THREADS < N) && ((dv[i] & himask) == prefix);
      hist_add_wave(hist, (dv[i] >> shift) & 255u, in, lane);
    }
    __syncthreads();
    if (t == 0) {
      unsigned int krem = sh_krem, cum = 0;
      int bin = 0;
      for (; bin < 256; ++bin) {
        if (cum + hist[bin] >= krem) break;
        cum += hist[bin];
      }
      if (bin > 255) bin = 255;
      sh_krem = krem - cum;
      sh_prefix = prefix | ((unsigned int)bin << shift);
      sh_neq = hist[bin];
    }
    __syncthreads();
  }
  const unsigned int T = sh_prefix;       // k-th smallest value
  unsigned int need = sh_krem;            // how many elements == T are wanted
  unsigned int idx_thr = 0xFFFFFFFFu;     // elements == T with idx <= idx_thr are taken
  if (sh_neq != need) {
    __syncthreads();
    if (t == 0) { sh_prefix = 0; sh_krem = need; }
    __syncthreads();
    for (int pass = 0; pass < 3; ++pass) {
      const int shift = 16 - 8 * pass;
      if (t < 256) hist[t] = 0;
      __syncthreads();
      const unsigned int prefix = sh_prefix;
      const unsigned int himask = (pass == 0) ? 0u : (0xFFFFFFFFu << (shift + 8));
#pragma unroll
      for (int i = 0; i < ITEMS; ++i) {
        const unsigned int n = (unsigned int)(t + i * TK_THREADS);
        const bool in = (n < (unsigned int)N) && dv[i] == T && ((n & himask) == prefix);
        hist_add_wave(hist, (n >> shift) & 255u, in, lane);
      }
      __syncthreads();
      if (t == 0) {
        unsigned int krem = sh_krem, cum = 0;
        int bin = 0;
        for (; bin < 256; ++bin) {
          if (cum + hist[bin] >= krem) break;
          cum += hist[bin];
        }
        if (bin > 255) bin = 255;
        sh_krem = krem - cum;
        sh_prefix = prefix | ((unsigned int)bin << shift);
      }
      __syncthreads();
    }
    idx_thr = sh_prefix;
  }
  // ---- collect survivors (one LDS atomic per wave-instruction: lanes take consecutive slots)
  if (t == 0) sh_count = 0;
  for (int i = t; i < kpad; i += TK_THREADS) keys[i] = 0xFFFFFFFFFFFFFFFFull;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const unsigned int n = (unsigned int)(t + i * TK_THREADS);
    const unsigned int v = dv[i];
    const bool take = (n < (unsigned int)N) && (v < T || (v == T && n <= idx_thr));
    const unsigned long long m = __ballot(take);
    if (m != 0ull) {
      const int leader = __ffsll((long long)m) - 1;
      unsigned int base = 0;
      if (lane == leader) base = atomicAdd(&sh_count, (unsigned int)__popcll(m));
      base = __builtin_amdgcn_readlane(base, leader);
      if (take) {
        const unsigned int slot = base + (unsigned int)__popcll(m & ((1ull << lane) - 1ull));
        if (slot < (unsigned int)kpad) keys[slot] = ((unsigned long long)v << 32) | n;
      }
    }
  }
  __syncthreads();
  // ---- bitonic sort of kpad 64-bit keys (value major, index minor)
  for (int size = 2; size <= kpad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = t; i < (kpad >> 1); i += TK_THREADS) {
        const int lo = (i / stride) * (stride << 1) + (i % stride);
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const unsigned long long a = keys[lo], c = keys[hi];
        if ((a > c) == up) { keys[lo] = c; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = t; i < k; i += TK_THREADS) {
    const unsigned long long kk = keys[i];
    idx_out[(size_t)b * k + i] = (long long)(kk & 0xFFFFFFFFull);
    if (dist_out) dist_out[(size_t)b * k + i] = __uint_as_float((unsigned int)(kk >> 32));
  }
}

// ctx[b][s][:] = s < k ? feat[b][idx ? idx[b][s] : s][:] : extra[b][s-k][:]   (rows of E floats, E % 4 == 0)
__global__ __launch_bounds__(256) void build_context_kernel(
    const float* __restrict__ feat, const long long* __restrict__ idx, const float* __restrict__ extra,
    float* __restrict__ ctx, int B, int Npts, int k, int X, int E4) {
  const int S = k + X;
  const size_t total = (size_t)B * S * E4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % E4);
    const size_t row = i / E4;
    const int s = (int)(row % S), b = (int)(row / S);
    float4 v;
    if (s < k) {
      const long long src = idx ? idx[(size_t)b * k + s] : (long long)s;
      v = reinterpret_cast<const float4*>(feat)[((size_t)b * Npts + src) * E4 + e];
    } else {
      v = reinterpret_cast<const float4*>(extra)[((size_t)b * X + (s - k)) * E4 + e];
    }
    reinterpret_cast<float4*>(ctx)[i] = v;
  }
}

// bf16 token rows (the FPN's channels-last bf16 output read in place: [B][Npts][4 * E4] bf16) -> fp32 context rows
// `bias` (fp32 [4 * E4] or null) is added to the gathered rows: the bias of the FPN's 3x3 output convolution, applied to the
// few rows a level reads instead of to the whole map (its gradient = the column sums of d(ctx), a3d_colsum_rows)
__global__ __launch_bounds__(256) void build_context_bf16_kernel(
    const unsigned short* __restrict__ feat, const long long* __restrict__ idx, const float* __restrict__ extra,
    const float* __restrict__ bias, float* __restrict__ ctx, int B, int Npts, int k, int X, int E4, int F4) {
  const int S = k + X;
  const size_t total = (size_t)B * S * E4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % E4);
    const size_t row = i / E4;
    const int s = (int)(row % S), b = (int)(row / S);
    float4 v;
    if (s < k) {
      const long long src = idx ? idx[(size_t)b * k + s] : (long long)s;
      const s16x4 h = reinterpret_cast<const s16x4*>(feat)[((size_t)b * Npts + src) * F4 + e];
      v = make_float4(bf2f((unsigned short)h[0]), bf2f((unsigned short)h[1]), bf2f((unsigned short)h[2]), bf2f((unsigned short)h[3]));
      if (bias) { v.x += bias[4 * e]; v.y += bias[4 * e + 1]; v.z += bias[4 * e + 2]; v.w += bias[4 * e + 3]; }
    } else {
      v = reinterpret_cast<const float4*>(extra)[((size_t)b * X + (s - k)) * E4 + e];
    }
    reinterpret_cast<float4*>(ctx)[i] = v;
  }
}

// d(ctx) rows -> the bf16 gradient map of the token tensor (zero-initialised by the caller, shared by every level that
// gathered from the map: accumulate = read-add-store, indices unique within one call); extra rows -> dextra (fp32)
__global__ __launch_bounds__(256) void build_context_bwd_bf16_kernel(
    const float* __restrict__ dctx, const long long* __restrict__ idx, unsigned short* __restrict__ dfeat,
    float* __restrict__ dextra, int B, int Npts, int k, int X, int E4, int F4, int accumulate) {
  const int S = k + X;
  const size_t total = (size_t)B * S * E4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % E4);
    const size_t row = i / E4;
    const int s = (int)(row % S), b = (int)(row / S);
    float4 v = reinterpret_cast<const float4*>(dctx)[i];
    if (s < k) {
      if (!dfeat) continue;
      const long long dst = idx ? idx[(size_t)b * k + s] : (long long)s;
      s16x4* p = reinterpret_cast<s16x4*>(dfeat) + ((size_t)b * Npts + dst) * F4 + e;
      if (accumulate) {
        const s16x4 o = *p;
        v.x += bf2f((unsigned short)o[0]); v.y += bf2f((unsigned short)o[1]);
        v.z += bf2f((unsigned short)o[2]); v.w += bf2f((unsigned short)o[3]);
      }
      s16x4 r;
      r[0] = (short)f2bf(v.x); r[1] = (short)f2bf(v.y); r[2] = (short)f2bf(v.z); r[3] = (short)f2bf(v.w);
      *p = r;
    } else if (dextra) {
      reinterpret_cast<float4*>(dextra)[((size_t)b * X + (s - k)) * E4 + e] = v;
    }
  }
}

// scalar-width variant for xyz rows (W floats per row)
__global__ __launch_bounds__(256) void build_context_rows_kernel(
    const float* __restrict__ feat, const long long* __restrict__ idx, const float* __restrict__ extra,
    float* __restrict__ ctx, int B, int Npts, int k, int X, int W) {
  const int S = k + X;
  const size_t total = (size_t)B * S * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % W);
    const size_t row = i / W;
    const int s = (int)(row % S), b = (int)(row / S);
    float v;
    if (s < k) {
      const long long src = idx ? idx[(size_t)b * k + s] : (long long)s;
      v = feat[((size_t)b * Npts + src) * W + e];
    } else {
      v = extra[((size_t)b * X + (s - k)) * W + e];
    }
    ctx[i] = v;
  }
}

// dfeat[b][idx[b][s]][:] (+)= dctx[b][s][:]  (indices unique per sample); dextra[b][x][:] = dctx[b][k+x][:]
__global__ __launch_bounds__(256) void build_context_bwd_kernel(
    const float* __restrict__ dctx, const long long* __restrict__ idx, float* __restrict__ dfeat,
    float* __restrict__ dextra, int B, int Npts, int k, int X, int E4, int accumulate) {
  const int S = k + X;
  const size_t total = (size_t)B * S * E4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % E4);
    const size_t row = i / E4;
    const int s = (int)(row % S), b = (int)(row / S);
    const float4 v = reinterpret_cast<const float4*>(dctx)[i];
    if (s < k) {
      if (!dfeat) continue;
      const long long dst = idx ? idx[(size_t)b * k + s] : (long long)s;
      float4* p = reinterpret_cast<float4*>(dfeat) + ((size_t)b * Npts + dst) * E4 + e;
      if (accumulate) {
        float4 o = *p;
        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
        *p = o;
      } else {
        *p = v;
      }
    } else if (dextra) {
      reinterpret_cast<float4*>(dextra)[((size_t)b * X + (s - k)) * E4 + e] = v;
    }
  }
}

}  // namespace a3d

using namespace a3d;

extern "C" int a3d_pcd_downsample(const float* pcd, float* out_xyz, int B, int C, int Hin, int Win, int factor,
                                  void* stream) {
  if (!pcd || !out_xyz || B <= 0 || C <= 0 || factor < 2 || (factor & 1) || Hin % factor || Win % factor) {
    set_error("a3d_pcd_downsample: bad argument (B=%d C=%d H=%d W=%d f=%d)", B, C, Hin, Win, factor);
    return A3D_ERR_ARG;
  }
  const size_t total = (size_t)B * C * (Hin / factor) * (Win / factor);
  const int grid = (int)std::min<size_t>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(pcd_downsample_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pcd, out_xyz, B * C, C,
                     Hin, Win, factor);
  return check_launch("a3d_pcd_downsample");
}

extern "C" size_t a3d_knn_topk_ws_bytes(int B, int N) { return (size_t)B * N * sizeof(unsigned int); }

static int nn_topk_launch(const char* fn, const float* pos, int npos, int squared, const float* xyz, void* ws,
                          long long* idx_out, float* dist_out, int B, int N, int k, void* stream) {
  if (!pos || !xyz || !ws || !idx_out || B <= 0 || N <= 0 || k <= 0 || k > N || k > 16384 || npos <= 0) {
    set_error("%s: bad argument (B=%d N=%d k=%d npos=%d; need 0 < k <= min(N, 16384))", fn, B, N, k, npos);
    return A3D_ERR_ARG;
  }
  int kpad = 2;
  while (kpad < k) kpad <<= 1;
  const size_t lds = (size_t)kpad * sizeof(unsigned long long);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)knn_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    (void)hipFuncSetAttribute((const void*)knn_topk_reg_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    (void)hipFuncSetAttribute((const void*)knn_topk_reg_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    attr_set = true;
  }
  static const bool use_ws = getenv("A3D_KNN_WS") && atoi(getenv("A3D_KNN_WS")) != 0;   // A/B switch: scratch-based kernel
  if (!use_ws && N <= 16 * TK_THREADS)
    hipLaunchKernelGGL(knn_topk_reg_kernel<16>, dim3(B), dim3(TK_THREADS), lds, (hipStream_t)stream, pos, xyz, idx_out,
                       dist_out, N, k, kpad, npos, squared);
  else if (!use_ws && N <= 64 * TK_THREADS)
    hipLaunchKernelGGL(knn_topk_reg_kernel<64>, dim3(B), dim3(TK_THREADS), lds, (hipStream_t)stream, pos, xyz, idx_out,
                       dist_out, N, k, kpad, npos, squared);
  else
    hipLaunchKernelGGL(knn_topk_kernel, dim3(B), dim3(TK_THREADS), lds, (hipStream_t)stream, pos, xyz,
                       (unsigned int*)ws, idx_out, dist_out, N, k, kpad, npos, squared);
  return check_launch(fn);
}

extern "C" int a3d_knn_topk(const float* pos, const float* xyz, void* ws, long long* idx_out, float* dist_out,
                            int B, int N, int k, void* stream) {
  return nn_topk_launch("a3d_knn_topk", pos, 1, 0, xyz, ws, idx_out, dist_out, B, N, k, stream);
}

extern "C" int a3d_traj_nn_topk(const float* traj_xyz, int L, const float* xyz, void* ws, long long* idx_out, float* dist_out,
                                int B, int N, int k, void* stream) {
  return nn_topk_launch("a3d_traj_nn_topk", traj_xyz, L, 1, xyz, ws, idx_out, dist_out, B, N, k, stream);
}

extern "C" int a3d_build_context(const float* feat, const long long* idx, const float* extra, float* ctx, int B,
                                 int Npts, int k, int X, int W, void* stream) {
  if (!feat || !ctx || B <= 0 || k <= 0 || X < 0 || W <= 0 || (X > 0 && !extra) || (!idx && k != Npts)) {
    set_error("a3d_build_context: bad argument (B=%d Npts=%d k=%d X=%d W=%d)", B, Npts, k, X, W);
    return A3D_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (W % 4 == 0) && ((((uintptr_t)feat | (uintptr_t)ctx | (uintptr_t)extra) & 15) == 0);
  if (vec) {
    const size_t total = (size_t)B * (k + X) * (W / 4);
    const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(build_context_kernel, dim3(grid), dim3(256), 0, s, feat, idx, extra, ctx, B, Npts, k, X, W / 4);
  } else {
    const size_t total = (size_t)B * (k + X) * W;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(build_context_rows_kernel, dim3(grid), dim3(256), 0, s, feat, idx, extra, ctx, B, Npts, k, X, W);
  }
  return check_launch("a3d_build_context");
}

extern "C" int a3d_build_context_bwd(const float* dctx, const long long* idx, float* dfeat, float* dextra, int B,
                                     int Npts, int k, int X, int W, int accumulate, void* stream) {
  if (!dctx || B <= 0 || k <= 0 || X < 0 || W <= 0 || (W % 4) != 0 || (!idx && k != Npts)) {
    set_error("a3d_build_context_bwd: bad argument (B=%d Npts=%d k=%d X=%d W=%d)", B, Npts, k, X, W);
    return A3D_ERR_ARG;
  }
  const size_t total = (size_t)B * (k + X) * (W / 4);
  const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(build_context_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dctx, idx, dfeat,
                     dextra, B, Npts, k, X, W / 4, accumulate);
  return check_launch("a3d_build_context_bwd");
}

extern "C" int a3d_build_context_bf16(const void* feat, int ldf, const long long* idx, const float* extra, const float* bias,
                                      float* ctx, int B, int Npts, int k, int X, int W, void* stream) {
  if (!feat || !ctx || B <= 0 || k <= 0 || X < 0 || W <= 0 || (W % 4) != 0 || ldf < W || (ldf % 4) != 0 || (X > 0 && !extra) || (!idx && k != Npts) ||
      ((((uintptr_t)feat) & 7) != 0) || ((((uintptr_t)ctx | (uintptr_t)extra) & 15) != 0)) {
    set_error("a3d_build_context_bf16: bad argument (B=%d Npts=%d k=%d X=%d W=%d; W %% 4 == 0, 8 / 16-byte aligned)", B, Npts, k, X, W);
    return A3D_ERR_ARG;
  }
  const size_t total = (size_t)B * (k + X) * (W / 4);
  const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(build_context_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)feat, idx,
                     extra, bias, ctx, B, Npts, k, X, W / 4, ldf / 4);
  return check_launch("a3d_build_context_bf16");
}

extern "C" int a3d_build_context_bwd_bf16(const float* dctx, const long long* idx, void* dfeat, int ldf, float* dextra, int B,
                                          int Npts, int k, int X, int W, int accumulate, void* stream) {
  if (!dctx || B <= 0 || k <= 0 || X < 0 || W <= 0 || (W % 4) != 0 || ldf < W || (ldf % 4) != 0 || (!idx && k != Npts) || ((((uintptr_t)dfeat) & 7) != 0)) {
    set_error("a3d_build_context_bwd_bf16: bad argument (B=%d Npts=%d k=%d X=%d W=%d)", B, Npts, k, X, W);
    return A3D_ERR_ARG;
  }
  const size_t total = (size_t)B * (k + X) * (W / 4);
  const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(build_context_bwd_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dctx, idx,
                     (unsigned short*)dfeat, dextra, B, Npts, k, X, W / 4, ldf / 4, accumulate);
  return check_launch("a3d_build_context_bwd_bf16");
}
