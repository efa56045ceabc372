// Small-E dense layers of the Act3D / ChainedDiffuser hot path on f32 MFMA (v_mfma_f32_16x16x4_f32).
//
// Every Linear on the path has K,N <= 512 (E = 60/120, FFN 4E = 480, instruction 512) while the row
// count M = B*tokens is large (up to B*4098), so these are streaming GEMMs: bound by HBM traffic of
// X and Y, not by MFMA.  They use the exact-f32 matrix instruction (same numerics as an fmaf chain,
// MI355X_MICROARCH.md "FP32-input MFMA") so the projections match the fp32 reference to rounding.
//
//   a3d_linear_fwd    Y = act(X W^T + b)            reference: F.linear in multihead_custom_attention.py:246-303,
//                                                   layers.py:313-332 (FFN), diffusion_head.py:41-49,177-199
//   a3d_linear_wgrad  dW += dY^T X, db += sum dY    (autograd of the same)
//   a3d_add_layernorm_{fwd,bwd}  y = LN(a + r)      reference: layers.py:308-309, 329-331 (post-norm residual)
#include "a3d_common.h"
#include "../../include/act3d_hip.h"
#include <stdlib.h>

namespace a3d {

// linear_split.hip
bool linear_split_applicable(const float* X, int ldx, const float* W, int ldw, const float* bias, const float* Y, int ldy,
                             const float* mask, int ldm, int M, int N, int K, int act);
int linear_split_launch(const float* X, int ldx, const float* W, int ldw, const float* bias, float* Y, int ldy, const float* mask,
                        int ldm, int M, int N, int K, int act, int w_transposed, hipStream_t s);
bool linear_wgrad_split_applicable(const float* dY, int lddy, const float* X, int ldx, int M);
int linear_wgrad_split_launch(const float* dY, int lddy, const float* X, int ldx, int has_bias, int M, int N, int K, int nsplit,
                              int rows_per_split, float* partial, hipStream_t s);

constexpr int LT_BM = 64;   // rows per workgroup
constexpr int LT_BN = 64;   // cols per workgroup
constexpr int LT_KC = 64;   // contraction chunk staged in LDS (K = 60 is ONE stage, K = 120 two)
constexpr int LT_LD = 68;   // padded LDS row stride in floats (272 B: conflict-free b128 column reads)

__device__ __forceinline__ float4 ld4_guard(const float* p, int k, int K, bool vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (vec && k + 3 < K) return *reinterpret_cast<const float4*>(p);
  if (k + 0 < K) v.x = p[0];
  if (k + 1 < K) v.y = p[1];
  if (k + 2 < K) v.z = p[2];
  if (k + 3 < K) v.w = p[3];
  return v;
}

// act: 0 none, 1 relu, 2 multiply by (mask > 0) [relu backward fused into dgrad], 3 Y += result [a gradient accumulated in place]
// One 64x64 output tile per workgroup; the next K-chunk is prefetched into registers while the current one is
// multiplied, so a chunk costs one HBM round trip, not two barriers + a dependent load.
// DROP: the nn.Dropout that follows the layer (dropout.hip's mask: element index m * N + n of the contiguous output, N % 8 == 0)
// applied in the epilogue -- bit-identical to a3d_linear_fwd followed by a3d_dropout in place, one launch instead of two.
template <bool WT, bool DROP>
__global__ __launch_bounds__(256) void linear_fwd_kernel(
    const float* __restrict__ X, int ldx, const float* __restrict__ W, int ldw,
    const float* __restrict__ bias, float* __restrict__ Y, int ldy,
    const float* __restrict__ mask, int ldm, int M, int N, int K, int act,
    const unsigned long long* __restrict__ drop_state, uint32_t drop_site, uint32_t drop_thr16, float drop_scale) {
  __shared__ __attribute__((aligned(16))) float Xs[LT_BM * LT_LD];
  __shared__ __attribute__((aligned(16))) float Ws[LT_BN * LT_LD];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * LT_BM, n0 = blockIdx.y * LT_BN;

  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool x_vec = ((ldx & 3) == 0) && ((((uintptr_t)X) & 15) == 0);
  const bool w_vec = ((ldw & 3) == 0) && ((((uintptr_t)W) & 15) == 0);
  // staging roles: thread -> row sr (0..63), four float4 at columns sk + 16*i
  const int sr = t >> 2, sk = (t & 3) * 4;
  // transposed-W staging roles: thread -> k row (t >> 4) + 16*i, four consecutive n at (t & 15) * 4
  const int tk = t >> 4, tn = (t & 15) * 4;

  float4 xr[4], wr[4];
  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + sk + 16 * i;
      const int m = m0 + sr;
      xr[i] = (m < M && k < K) ? ld4_guard(X + (size_t)m * ldx + k, k, K, x_vec) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (!WT) {
        const int n = n0 + sr;
        wr[i] = (n < N && k < K) ? ld4_guard(W + (size_t)n * ldw + k, k, K, w_vec) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const int kk = k0 + tk + 16 * i, n = n0 + tn;
        wr[i] = (kk < K && n < N) ? ld4_guard(W + (size_t)kk * ldw + n, n, N, w_vec) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(&Xs[sr * LT_LD + sk + 16 * i]) = xr[i];
      if (!WT) {
        *reinterpret_cast<float4*>(&Ws[sr * LT_LD + sk + 16 * i]) = wr[i];
      } else {
        const int kk = tk + 16 * i;
        Ws[(tn + 0) * LT_LD + kk] = wr[i].x;
        Ws[(tn + 1) * LT_LD + kk] = wr[i].y;
        Ws[(tn + 2) * LT_LD + kk] = wr[i].z;
        Ws[(tn + 3) * LT_LD + kk] = wr[i].w;
      }
    }
  };

  load_chunk(0);
  for (int k0 = 0; k0 < K; k0 += LT_KC) {
    store_chunk();
    __syncthreads();
    if (k0 + LT_KC < K) load_chunk(k0 + LT_KC);
    const int ksteps = min(4, (K - k0 + 15) >> 4);
    for (int s4 = 0; s4 < ksteps; ++s4) {
      // contraction index of MFMA step j in lane group g is k = s4*16 + g*4 + j on both operands
      const float4 a = *reinterpret_cast<const float4*>(&Xs[(wave * 16 + li) * LT_LD + s4 * 16 + g * 4]);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float4 b = *reinterpret_cast<const float4*>(&Ws[(nt * 16 + li) * LT_LD + s4 * 16 + g * 4]);
        acc[nt] = mfma_f32_16x16x4(a.x, b.x, acc[nt]);
        acc[nt] = mfma_f32_16x16x4(a.y, b.y, acc[nt]);
        acc[nt] = mfma_f32_16x16x4(a.z, b.z, acc[nt]);
        acc[nt] = mfma_f32_16x16x4(a.w, b.w, acc[nt]);
      }
    }
    __syncthreads();
  }
  // C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg
  DropKey dkey{0u, 0u};
  if (DROP) dkey = drop_key(drop_state);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int n = n0 + nt * 16 + li;
    // A 16 x 16 tile holds 32 eight-element mask blocks (16 rows x 2 halves): lane (g, li) generates the block of row
    // g*4 + (li & 3), half (li >> 2) & 1 (every block twice), and the four rows of a lane's column fetch theirs by shuffle --
    // one Philox call per lane and tile instead of one per element.  All lanes take part (before the bounds guards).
    uint32_t kbits[4] = {0u, 0u, 0u, 0u};
    if (DROP) {
      const size_t blk = ((size_t)(m0 + wave * 16 + g * 4 + (li & 3)) * N + (n0 + nt * 16 + ((li >> 2) & 1) * 8)) >> 3;
      const uint32_t mine = drop_keep8(dkey, (uint32_t)blk, (uint32_t)(blk >> 32), 0xFFFFFFFFu, drop_site, drop_thr16);
#pragma unroll
      for (int r = 0; r < 4; ++r) kbits[r] = (uint32_t)__shfl((int)mine, (lane & 48) + r + 4 * (li >> 3), 64);
    }
    if (n >= N) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wave * 16 + g * 4 + r;
      if (m >= M) continue;
      float v = acc[nt][r] + bv;
      if (act == 1) v = fmaxf(v, 0.f);
      else if (act == 2) v = (mask[(size_t)m * ldm + n] > 0.f) ? v : 0.f;
      else if (act == 3) v += Y[(size_t)m * ldy + n];
      if (DROP) v = ((kbits[r] >> (li & 7)) & 1u) ? v * drop_scale : 0.f;
      Y[(size_t)m * ldy + n] = v;
    }
  }
}

// dW[n][k] += sum_m dY[m][n] X[m][k];  column k == K of the virtual X is all ones -> db[n].
// 64 (n) x 64 (k) outputs per workgroup, the m reduction split over blockIdx.z; 64 rows of m per LDS stage with the
// next stage prefetched into registers.
constexpr int WG_MC = 64;    // rows of m staged per step
constexpr int WG_LD = 68;    // padded LDS row stride (floats)
__global__ __launch_bounds__(256) void linear_wgrad_kernel(
    const float* __restrict__ dY, int lddy, const float* __restrict__ X, int ldx,
    float* __restrict__ dW, int lddw, float* __restrict__ db, int M, int N, int K, int rows_per_split,
    float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float Ys[WG_MC * WG_LD];
  __shared__ __attribute__((aligned(16))) float Xs[WG_MC * WG_LD];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int mbeg = blockIdx.z * rows_per_split;
  const int mend = min(M, mbeg + rows_per_split);
  const int KE = db ? K + 1 : K;   // virtual ones column

  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool y_vec = ((lddy & 3) == 0) && ((((uintptr_t)dY) & 15) == 0) && ((n0 & 3) == 0);
  const bool x_vec = ((ldx & 3) == 0) && ((((uintptr_t)X) & 15) == 0);
  const int sm = t >> 4, sc = (t & 15) * 4;   // thread -> m rows sm + 16*i, 4 consecutive cols at sc
  float4 yr[4], xr[4];
  auto load_stage = [&](int mb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mb + sm + 16 * i;
      float4 vy = make_float4(0.f, 0.f, 0.f, 0.f), vx = vy;
      if (m < mend) {
        vy = ld4_guard(dY + (size_t)m * lddy + n0 + sc, n0 + sc, N, y_vec);
        const int k = k0 + sc;
        vx = ld4_guard(X + (size_t)m * ldx + k, k, K, x_vec);
        if (db) {
          if (k + 0 == K) vx.x = 1.f;
          if (k + 1 == K) vx.y = 1.f;
          if (k + 2 == K) vx.z = 1.f;
          if (k + 3 == K) vx.w = 1.f;
        }
      }
      yr[i] = vy;
      xr[i] = vx;
    }
  };
  load_stage(mbeg);
  for (int mb = mbeg; mb < mend; mb += WG_MC) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<float4*>(&Ys[(sm + 16 * i) * WG_LD + sc]) = yr[i];
      *reinterpret_cast<float4*>(&Xs[(sm + 16 * i) * WG_LD + sc]) = xr[i];
    }
    __syncthreads();
    if (mb + WG_MC < mend) load_stage(mb + WG_MC);
    const int msteps = min(4, (mend - mb + 15) >> 4);
    for (int mm = 0; mm < msteps; ++mm) {
      // wave -> n tile `wave`; contraction index of step j in group g is m = mm*16 + g*4 + j
      float a[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = Ys[(mm * 16 + g * 4 + j) * WG_LD + wave * 16 + li];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float b = Xs[(mm * 16 + g * 4 + j) * WG_LD + kt * 16 + li];
          acc[kt] = mfma_f32_16x16x4(a[j], b, acc[kt]);
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int k = k0 + kt * 16 + li;
    if (k >= KE) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wave * 16 + g * 4 + r;
      if (n >= N) continue;
      const float v = acc[kt][r];
      if (partial) partial[((size_t)blockIdx.z * N + n) * KE + k] = v;   // two-stage: plain store, reduced below
      else if (k < K) atomicAdd(&dW[(size_t)n * lddw + k], v);
      else atomicAdd(&db[n], v);
    }
  }
}

// second stage of the split-M weight gradient: dW[n][k] (+ db[n]) += sum_z partial[z][n][k], z in fixed order
// (deterministic, and no memory-side float atomics: those serialise on the N*K addresses, ~10 G atomics/s)
// 1024 threads = 64 outputs x 16 split-groups: <= nsplit/16 dependent loads per thread.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit,
                                                            float* __restrict__ dW, int lddw, float* __restrict__ db,
                                                            int N, int K, int KE) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, zg = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + lane;
  const int total = N * KE;
  float s0 = 0.f, s1 = 0.f;
  if (idx < total) {
    int z = zg;
    for (; z + 16 < nsplit; z += 32) {
      s0 += partial[(size_t)z * total + idx];
      s1 += partial[(size_t)(z + 16) * total + idx];
    }
    if (z < nsplit) s0 += partial[(size_t)z * total + idx];
  }
  red[zg][lane] = s0 + s1;
  __syncthreads();
  if (zg != 0 || idx >= total) return;
  float v = 0.f;
#pragma unroll
  for (int u = 0; u < 16; ++u) v += red[u][lane];
  const int n = idx / KE, k = idx - n * KE;
  if (k < K) dW[(size_t)n * lddw + k] += v;
  else db[n] += v;
}

// ---------------------------------------------------------------- residual + LayerNorm
// one wave per row, E <= 512
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(
    const float* __restrict__ A, const float* __restrict__ R, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ Y, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, int M, int E, float eps) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  for (int m = wave_global; m < M; m += nwaves) {
    float v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = lane + i * 64;
      float x = 0.f;
      if (e < E) {
        x = A[(size_t)m * E + e];
        if (R) x += R[(size_t)m * E + e];
      }
      v[i] = x;
      s += x;
    }
    const float mean = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = lane + i * 64;
      if (e < E) { const float d = v[i] - mean; q += d * d; }
    }
    const float var = wave_sum(q) / (float)E;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = lane + i * 64;
      if (e < E) Y[(size_t)m * E + e] = (v[i] - mean) * rstd * gamma[e] + beta[e];
    }
    if (lane == 0) { mean_out[m] = mean; rstd_out[m] = rstd; }
  }
}

__global__ __launch_bounds__(256) void add_ln_bwd_kernel(
    const float* __restrict__ A, const float* __restrict__ R, const float* __restrict__ gamma,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    const float* __restrict__ dY, float* __restrict__ dS, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int M, int E) {
  __shared__ float red_g[4][512];
  __shared__ float red_b[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_global = blockIdx.x * 4 + wave;
  const int nwaves = gridDim.x * 4;
  float pg[8], pb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { pg[i] = 0.f; pb[i] = 0.f; }
  for (int m = wave_global; m < M; m += nwaves) {
    const float mean = mean_in[m], rstd = rstd_in[m];
    float xh[8], gy[8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = lane + i * 64;
      xh[i] = 0.f; gy[i] = 0.f;
      if (e < E) {
        float x = A[(size_t)m * E + e];
        if (R) x += R[(size_t)m * E + e];
        const float dy = dY[(size_t)m * E + e];
        xh[i] = (x - mean) * rstd;
        gy[i] = dy * gamma[e];
        pg[i] += dy * xh[i];
        pb[i] += dy;
        s1 += gy[i];
        s2 += gy[i] * xh[i];
      }
    }
    s1 = wave_sum(s1) / (float)E;
    s2 = wave_sum(s2) / (float)E;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = lane + i * 64;
      if (e < E) dS[(size_t)m * E + e] = rstd * (gy[i] - s1 - xh[i] * s2);
    }
  }
  if (dgamma) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = lane + i * 64;
      if (e < 512) { red_g[wave][e] = pg[i]; red_b[wave][e] = pb[i]; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += 256) {
      const float sg = red_g[0][e] + red_g[1][e] + red_g[2][e] + red_g[3][e];
      const float sb = red_b[0][e] + red_b[1][e] + red_b[2][e] + red_b[3][e];
      atomicAdd(&dgamma[e], sg);
      atomicAdd(&dbeta[e], sb);
    }
  }
}

// The same for E <= 128 (both models: 60 / 120) with LPR = 16 / 32 lanes per row and a float4 per lane: 4 / 2 rows per wave and
// pass instead of one row per wave with 60 of 512 element slots used, 4- / 5-step reductions, a workgroup walks `rows_per_wg`
// consecutive rows (one pass for the small maps) and issues its 2 E atomics once.  The one-row-per-wave kernel above took 33 us for
// the 21 312 x 60 ghost-token rows (5 MB): a chain of dependent load -> 6-step reduce -> store rounds, ~5 rows deep per wave.
// DROP: second output dSd = dropout(dS) (the gradient through the nn.Dropout on the residual branch, dropout.hip's mask over the
// flat index m * E + e, E % 8 == 0) -- bit-identical to a3d_dropout(dS -> dSd) afterwards, without the launch and the re-read.
template <int LPR, bool DROP>
__global__ __launch_bounds__(256) void add_ln_bwd_rows_kernel(
    const float* __restrict__ A, const float* __restrict__ R, const float* __restrict__ gamma,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in, const float* __restrict__ dY,
    float* __restrict__ dS, float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int E, int rows_per_wg,
    float* __restrict__ dSd, const unsigned long long* __restrict__ drop_state, uint32_t drop_site, uint32_t drop_thr16,
    float drop_scale) {
  constexpr int RPW = 64 / LPR, SLOTS = 256 / LPR;
  __shared__ float red_g[SLOTS][LPR * 4];
  __shared__ float red_b[SLOTS][LPR * 4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int sub = lane / LPR, l = lane % LPR;
  const int e0 = l * 4;
  const bool act = e0 < E;                                       // E % 4 == 0: a lane's four channels are all valid or all padding
  float gm[4] = {0.f, 0.f, 0.f, 0.f};
  if (act) {
#pragma unroll
    for (int j = 0; j < 4; ++j) gm[j] = gamma[e0 + j];           // parameters in a flat buffer are only 4-byte aligned
  }
  float pg[4] = {0.f, 0.f, 0.f, 0.f}, pb[4] = {0.f, 0.f, 0.f, 0.f};
  const int m_beg = blockIdx.x * rows_per_wg, m_end = min(M, m_beg + rows_per_wg);
  const float inv_e = 1.0f / (float)E;
  for (int m0 = m_beg + wave * RPW; m0 < m_end; m0 += 4 * RPW) {
    const int m = m0 + sub;
    const bool ok = act && m < m_end;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f), dy = x;
    float mean = 0.f, rstd = 0.f;
    if (ok) {
      x = *reinterpret_cast<const float4*>(A + (size_t)m * E + e0);
      if (R) {
        const float4 r = *reinterpret_cast<const float4*>(R + (size_t)m * E + e0);
        x.x += r.x; x.y += r.y; x.z += r.z; x.w += r.w;
      }
      dy = *reinterpret_cast<const float4*>(dY + (size_t)m * E + e0);
      mean = mean_in[m];
      rstd = rstd_in[m];
    }
    const float xv[4] = {x.x, x.y, x.z, x.w}, dv[4] = {dy.x, dy.y, dy.z, dy.w};
    float xh[4], gy[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xh[j] = ok ? (xv[j] - mean) * rstd : 0.f;
      gy[j] = dv[j] * gm[j];
      pg[j] += dv[j] * xh[j];
      pb[j] += dv[j];
      s1 += gy[j];
      s2 += gy[j] * xh[j];
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    s1 *= inv_e;
    s2 *= inv_e;
    if (ok) {
      const float4 o = make_float4(rstd * (gy[0] - s1 - xh[0] * s2), rstd * (gy[1] - s1 - xh[1] * s2),
                                   rstd * (gy[2] - s1 - xh[2] * s2), rstd * (gy[3] - s1 - xh[3] * s2));
      *reinterpret_cast<float4*>(dS + (size_t)m * E + e0) = o;
      if (DROP) {
        const size_t idx = (size_t)m * E + e0;                     // a lane's four channels are one half of a mask block
        const uint32_t keep = drop_keep8(drop_key(drop_state), (uint32_t)(idx >> 3), (uint32_t)(idx >> 35), 0xFFFFFFFFu, drop_site,
                                         drop_thr16) >> (idx & 4);
        *reinterpret_cast<float4*>(dSd + idx) = make_float4((keep & 1u) ? o.x * drop_scale : 0.f, (keep & 2u) ? o.y * drop_scale : 0.f,
                                                            (keep & 4u) ? o.z * drop_scale : 0.f, (keep & 8u) ? o.w * drop_scale : 0.f);
      }
    }
  }
  if (dgamma) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { red_g[wave * RPW + sub][e0 + j] = pg[j]; red_b[wave * RPW + sub][e0 + j] = pb[j]; }
    __syncthreads();
    for (int e = t; e < E; e += 256) {
      float sg = 0.f, sb = 0.f;
#pragma unroll
      for (int u = 0; u < SLOTS; ++u) { sg += red_g[u][e]; sb += red_b[u][e]; }
      atomicAdd(&dgamma[e], sg);
      atomicAdd(&dbeta[e], sb);
    }
  }
}

}  // namespace a3d

using namespace a3d;

extern "C" int a3d_linear_fwd(const float* X, int ldx, const float* W, int ldw, const float* bias,
                              float* Y, int ldy, const float* mask, int ldm, int M, int N, int K,
                              int act, int w_transposed, void* stream) {
  if (!X || !W || !Y || M < 0 || N <= 0 || K <= 0 || ldx < K || ldy < N) {
    set_error("a3d_linear_fwd: bad argument (M=%d N=%d K=%d ldx=%d ldy=%d)", M, N, K, ldx, ldy);
    return A3D_ERR_ARG;
  }
  if (act == 2 && !mask) { set_error("a3d_linear_fwd: act=2 needs a mask"); return A3D_ERR_ARG; }
  if (M == 0) return A3D_OK;
  dim3 grid(cdiv(M, LT_BM), cdiv(N, LT_BN));
  hipStream_t s = (hipStream_t)stream;
  // large row counts: the bf16x3 kernel of linear_split.hip (fp32-accurate, 2.7x the matrix rate of the f32 MFMA)
  if (linear_split_applicable(X, ldx, W, ldw, bias, Y, ldy, mask, ldm, M, N, K, act))
    return linear_split_launch(X, ldx, W, ldw, bias, Y, ldy, mask, ldm, M, N, K, act, w_transposed, s);
  if (w_transposed)
    hipLaunchKernelGGL((linear_fwd_kernel<true, false>), grid, dim3(256), 0, s, X, ldx, W, ldw, bias, Y, ldy, mask, ldm, M, N, K, act,
                       (const unsigned long long*)nullptr, 0u, 0u, 0.f);
  else
    hipLaunchKernelGGL((linear_fwd_kernel<false, false>), grid, dim3(256), 0, s, X, ldx, W, ldw, bias, Y, ldy, mask, ldm, M, N, K, act,
                       (const unsigned long long*)nullptr, 0u, 0u, 0.f);
  return check_launch("a3d_linear_fwd");
}

static int linear_drop_params(const char* fn, float p, uint32_t* thr16, float* scale) {     // as dropout.hip's drop_params
  if (!(p >= 0.f) || !(p < 1.f)) {
    set_error("%s: dropout probability %g outside [0, 1)", fn, (double)p);
    return A3D_ERR_ARG;
  }
  *thr16 = (uint32_t)lrintf(p * 65536.0f);
  *scale = 1.0f / (1.0f - p);
  return A3D_OK;
}

// Y = dropout(act(X W^T + b)): the layer and the nn.Dropout behind it (layers.py:82-84,146,181, diffusion_head.py:46,183,193) in
// one launch.  Same bits as a3d_linear_fwd + a3d_dropout(Y -> Y) -- which is also what runs when the epilogue does not apply
// (N % 8 != 0, or a row count that the bf16x3 kernel of linear_split.hip serves).
extern "C" int a3d_linear_fwd_drop(const float* X, int ldx, const float* W, int ldw, const float* bias, float* Y, int ldy,
                                   const float* mask, int ldm, int M, int N, int K, int act, int w_transposed,
                                   const unsigned long long* state, unsigned int site, float p, void* stream) {
  if (!X || !W || !Y || !state || M < 0 || N <= 0 || K <= 0 || ldx < K || ldy != N || act == 3) {
    set_error("a3d_linear_fwd_drop: bad argument (M=%d N=%d K=%d ldx=%d ldy=%d act=%d; the output must be contiguous, ldy == N, and "
              "act 3 has no dropout form)", M, N, K, ldx, ldy, act);
    return A3D_ERR_ARG;
  }
  if (act == 2 && !mask) { set_error("a3d_linear_fwd_drop: act=2 needs a mask"); return A3D_ERR_ARG; }
  uint32_t thr; float scale;
  int rc = linear_drop_params("a3d_linear_fwd_drop", p, &thr, &scale);
  if (rc) return rc;
  if (M == 0) return A3D_OK;
  if ((N & 7) || linear_split_applicable(X, ldx, W, ldw, bias, Y, ldy, mask, ldm, M, N, K, act)) {
    rc = a3d_linear_fwd(X, ldx, W, ldw, bias, Y, ldy, mask, ldm, M, N, K, act, w_transposed, stream);
    return rc ? rc : a3d_dropout(Y, Y, (size_t)M * N, state, site, p, stream);
  }
  dim3 grid(cdiv(M, LT_BM), cdiv(N, LT_BN));
  hipStream_t s = (hipStream_t)stream;
  if (w_transposed)
    hipLaunchKernelGGL((linear_fwd_kernel<true, true>), grid, dim3(256), 0, s, X, ldx, W, ldw, bias, Y, ldy, mask, ldm, M, N, K, act,
                       state, (uint32_t)site, thr, scale);
  else
    hipLaunchKernelGGL((linear_fwd_kernel<false, true>), grid, dim3(256), 0, s, X, ldx, W, ldw, bias, Y, ldy, mask, ldm, M, N, K, act,
                       state, (uint32_t)site, thr, scale);
  return check_launch("a3d_linear_fwd_drop");
}

// With a workspace the M reduction is split into single-stage (64-row) .. 256-row chunks whose partial tiles a second
// kernel adds in order; without one (or for M < 1024) the one-stage kernel accumulates with float atomics.
// (A3D_WGRAD_TWO_STAGE_MIN_ROWS to A/B the threshold: the trajectory stream's M = B * L = 1100 rows sit just above it.)
static int wg_two_stage_min_rows() {
  static const int v = getenv("A3D_WGRAD_TWO_STAGE_MIN_ROWS") ? atoi(getenv("A3D_WGRAD_TWO_STAGE_MIN_ROWS")) : 1024;
  return v;
}

static void wgrad_plan(int M, int N, int KE, bool have_ws, int* nsplit_out, int* rows_out, bool* two_stage) {
  const int tiles = cdiv(N, 64) * cdiv(KE, 64);
  int nsplit;
  *two_stage = have_ws && M >= wg_two_stage_min_rows();
  if (*two_stage) {
    // no atomics: oversubscribe (up to ~4 workgroups per CU hide the stage latency); short chunks for mid-size M
    static int target2 = getenv("A3D_WGRAD_WGS2") ? atoi(getenv("A3D_WGRAD_WGS2")) : 1024;
    nsplit = max(1, min(cdiv(M, WG_MC), cdiv(target2, tiles)));
    if (cdiv(M, nsplit) > 256) nsplit = max(nsplit, min(cdiv(M, 256), 4 * cdiv(target2, tiles)));
  } else {
    // one-stage: ~256 workgroups in flight, at least 256 rows (4 stages) per workgroup
    static int target_wgs = getenv("A3D_WGRAD_WGS") ? atoi(getenv("A3D_WGRAD_WGS")) : 256;
    nsplit = max(1, min(cdiv(M, 256), cdiv(target_wgs, tiles)));
  }
  int rows = cdiv(cdiv(M, nsplit), WG_MC) * WG_MC;
  *nsplit_out = cdiv(M, rows);
  *rows_out = rows;
}

extern "C" size_t a3d_linear_wgrad_ws_bytes(int M, int N, int K, int has_bias) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int KE = has_bias ? K + 1 : K;
  int nsplit, rows;
  bool two;
  wgrad_plan(M, N, KE, true, &nsplit, &rows, &two);
  return two ? (size_t)nsplit * N * KE * sizeof(float) : 0;
}

extern "C" int a3d_linear_wgrad_ws(const float* dY, int lddy, const float* X, int ldx, float* dW, int lddw,
                                   float* db, int M, int N, int K, float* ws, size_t ws_bytes, void* stream) {
  if (!dY || !X || !dW || M < 0 || N <= 0 || K <= 0) {
    set_error("a3d_linear_wgrad: bad argument (M=%d N=%d K=%d)", M, N, K);
    return A3D_ERR_ARG;
  }
  if (M == 0) return A3D_OK;
  const int KE = db ? K + 1 : K;
  int nsplit, rows;
  bool two_stage;
  wgrad_plan(M, N, KE, ws != nullptr, &nsplit, &rows, &two_stage);
  if (two_stage && ws_bytes < (size_t)nsplit * N * KE * sizeof(float)) {
    set_error("a3d_linear_wgrad_ws: workspace too small (%zu bytes, need %zu)", ws_bytes,
              (size_t)nsplit * N * KE * sizeof(float));
    return A3D_ERR_ARG;
  }
  dim3 grid(cdiv(N, 64), cdiv(KE, 64), nsplit);
  if (two_stage && linear_wgrad_split_applicable(dY, lddy, X, ldx, M)) {
    // large row counts: first stage on the bf16 pipe with three-part operands (linear_split.hip), same partial layout
    const int rc = linear_wgrad_split_launch(dY, lddy, X, ldx, db != nullptr, M, N, K, nsplit, rows, ws, (hipStream_t)stream);
    if (rc) return rc;
  } else
  hipLaunchKernelGGL(linear_wgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, dY, lddy, X, ldx, dW,
                     lddw, db, M, N, K, rows, two_stage ? ws : nullptr);
  if (two_stage)
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(N * KE, 64)), dim3(1024), 0, (hipStream_t)stream, ws, nsplit, dW,
                       lddw, db, N, K, KE);
  return check_launch("a3d_linear_wgrad");
}

extern "C" int a3d_sq_wgrad_reduce(const float* partial, int nsplit, float* dW, int lddw, float* db, int E, void* stream) {
  if (!partial || !dW || !db || nsplit < 1 || E <= 0) { set_error("a3d_sq_wgrad_reduce: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(E * (E + 1), 64)), dim3(1024), 0, (hipStream_t)stream, partial, nsplit, dW,
                     lddw, db, E, E, E + 1);
  return check_launch("a3d_sq_wgrad_reduce");
}

extern "C" int a3d_linear_wgrad(const float* dY, int lddy, const float* X, int ldx, float* dW, int lddw,
                                float* db, int M, int N, int K, void* stream) {
  return a3d_linear_wgrad_ws(dY, lddy, X, ldx, dW, lddw, db, M, N, K, nullptr, 0, stream);
}

extern "C" int a3d_add_layernorm_fwd(const float* A, const float* R, const float* gamma, const float* beta,
                                     float* Y, float* mean, float* rstd, int M, int E, float eps,
                                     void* stream) {
  if (!A || !gamma || !beta || !Y || !mean || !rstd || E <= 0 || E > 512 || M < 0) {
    set_error("a3d_add_layernorm_fwd: bad argument (M=%d E=%d)", M, E);
    return A3D_ERR_ARG;
  }
  if (M == 0) return A3D_OK;
  const int grid = min(cdiv(M, 4), 4096);
  hipLaunchKernelGGL(add_ln_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, A, R, gamma, beta, Y,
                     mean, rstd, M, E, eps);
  return check_launch("a3d_add_layernorm_fwd");
}

extern "C" int a3d_add_layernorm_bwd(const float* A, const float* R, const float* gamma, const float* mean,
                                     const float* rstd, const float* dY, float* dS, float* dgamma,
                                     float* dbeta, int M, int E, void* stream) {
  if (!A || !gamma || !mean || !rstd || !dY || !dS || E <= 0 || E > 512 || M < 0 || (!dgamma != !dbeta)) {
    set_error("a3d_add_layernorm_bwd: bad argument (M=%d E=%d)", M, E);
    return A3D_ERR_ARG;
  }
  if (M == 0) return A3D_OK;
  // Measured (round 5, gpurun r05a): E = 120 22.1 -> 14.8 us per call (the trajectory model), E = 60 33.0 -> 35.1 us (Act3D's
  // 21 312 ghost rows: no gain) -- the rows kernel serves 64 < E <= 128 only.
  if (E > 64 && E <= 128 && (E & 3) == 0 && ((((uintptr_t)A) | ((uintptr_t)R) | ((uintptr_t)dY) | ((uintptr_t)dS)) & 15) == 0) {
    // rows-per-workgroup: one pass (16 / 8 rows) while that still fills the chip, more rows per workgroup (fewer atomics) beyond
    const int per_pass = E <= 64 ? 16 : 8;
    int rpw = per_pass;
    while (cdiv(M, rpw) > 2048 && rpw < 16 * per_pass) rpw *= 2;
    const int grid = cdiv(M, rpw);
    if (E <= 64)
      hipLaunchKernelGGL((add_ln_bwd_rows_kernel<16, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, A, R, gamma, mean, rstd, dY, dS,
                         dgamma, dbeta, M, E, rpw, (float*)nullptr, (const unsigned long long*)nullptr, 0u, 0u, 0.f);
    else
      hipLaunchKernelGGL((add_ln_bwd_rows_kernel<32, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, A, R, gamma, mean, rstd, dY, dS,
                         dgamma, dbeta, M, E, rpw, (float*)nullptr, (const unsigned long long*)nullptr, 0u, 0u, 0.f);
    return check_launch("a3d_add_layernorm_bwd");
  }
  const int grid = min(cdiv(M, 16), 1024);
  hipLaunchKernelGGL(add_ln_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, A, R, gamma, mean,
                     rstd, dY, dS, dgamma, dbeta, M, E);
  return check_launch("a3d_add_layernorm_bwd");
}

// a3d_add_layernorm_bwd with a second output dS_drop = dropout(dS): the gradient of the residual branch behind its nn.Dropout
// (layers.py:146,181: x + dropout(branch) -> LayerNorm).  Same bits as a3d_add_layernorm_bwd + a3d_dropout(dS -> dS_drop), which is
// what runs outside the rows kernel's range (64 < E <= 128, E % 8 == 0).
extern "C" int a3d_add_layernorm_bwd_drop(const float* A, const float* R, const float* gamma, const float* mean, const float* rstd,
                                          const float* dY, float* dS, float* dS_drop, float* dgamma, float* dbeta, int M, int E,
                                          const unsigned long long* state, unsigned int site, float p, void* stream) {
  if (!A || !gamma || !mean || !rstd || !dY || !dS || !dS_drop || !state || dS_drop == dS || E <= 0 || E > 512 || M < 0 ||
      (!dgamma != !dbeta)) {
    set_error("a3d_add_layernorm_bwd_drop: bad argument (M=%d E=%d; dS_drop must be a second buffer)", M, E);
    return A3D_ERR_ARG;
  }
  uint32_t thr; float scale;
  int rc = linear_drop_params("a3d_add_layernorm_bwd_drop", p, &thr, &scale);
  if (rc) return rc;
  if (M == 0) return A3D_OK;
  if (E > 64 && E <= 128 && (E & 7) == 0 &&
      ((((uintptr_t)A) | ((uintptr_t)R) | ((uintptr_t)dY) | ((uintptr_t)dS) | ((uintptr_t)dS_drop)) & 15) == 0) {
    int rpw = 8;                                                   // as a3d_add_layernorm_bwd for this width
    while (cdiv(M, rpw) > 2048 && rpw < 128) rpw *= 2;
    hipLaunchKernelGGL((add_ln_bwd_rows_kernel<32, true>), dim3(cdiv(M, rpw)), dim3(256), 0, (hipStream_t)stream, A, R, gamma, mean, rstd,
                       dY, dS, dgamma, dbeta, M, E, rpw, dS_drop, state, (uint32_t)site, thr, scale);
    return check_launch("a3d_add_layernorm_bwd_drop");
  }
  rc = a3d_add_layernorm_bwd(A, R, gamma, mean, rstd, dY, dS, dgamma, dbeta, M, E, stream);
  return rc ? rc : a3d_dropout(dS, dS_drop, (size_t)M * E, state, site, p, stream);
}
