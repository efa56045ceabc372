// Act3D's query stream -- ONE learned query token per sample through num_query_cross_attn_layers x
// [RelativeCrossAttentionLayer -> FeedforwardLayer] per pyramid level (act3d.py:467-480, layers.py:293-351) -- as four fused
// launches per layer and direction around the key-streaming kernels of single_query.hip.
//
// The stream has M = B rows (64 at the bench shape) of E = 60 channels: every dense layer on it is a 64 x 60 x 60 product.
// Run op by op (round 3) a layer was 10 forward and 16 backward launches of one workgroup each -- q projection, RoPE, combine,
// value projection, out projection, LayerNorm, two FFN linears, LayerNorm; their dgrads, wgrads, reductions and the adds
// between them -- ~7 us apiece on the step's critical path: 52 single-workgroup a3d_linear_fwd + 28 a3d_linear_wgrad + 48
// add_ln launches per step, ~0.9 ms of 27.  Here one workgroup keeps the rows in LDS and walks the chain:
//   a3d_qs_pre_fwd    x -> q = (W_q x + b_q) * d^-1/2 -> RoPE(q, xyz) -> qrot [B][H][16]      (feeds a3d_sq_attn_fwd)
//   a3d_qs_post_fwd   xbar (a3d_sq_attn_fwd's per-head weighted context mean) -> value projection -> out projection ->
//                     LayerNorm(x + .) -> FFN -> LayerNorm(. + FFN) = the layer's output; intermediates saved for the backward
//   a3d_qs_post_bwd   d(output) -> both LayerNorms, the FFN, the out / value projections backwards, every weight and bias
//                     gradient accumulated in place -> dxbar, cD (feed a3d_sq_attn_bwd) and the residual branch's d(x)
//   a3d_qs_pre_bwd    dq partials of a3d_sq_attn_bwd -> RoPE^T, scale -> dW_q, db_q; d(x) += dq W_q
// All products use the exact-f32 MFMA (v_mfma_f32_16x16x4_f32, an fmaf chain in k order, as linear.hip); weight gradients are
// accumulated in registers by the single workgroup and flushed once with returnless float atomics (one add per address per
// launch: the result does not depend on any ordering, unlike the multi-workgroup atomics of the small-M a3d_linear_wgrad).
// Restricted like single_query.hip: E <= 60 (one 64-wide tile), E % 12 == 0, H <= 4, FFN hidden = E.
// Measured (MI355X, B = 64): 12 + 38 us forward, 82 + 31 us backward per layer against ~135 us for the 26
// launches replaced -- what is left is one workgroup's dependency chain (~10 dense stages, each one L2 round trip + ~1 us of
// MFMA + barriers), which is why the win is the launch count (-106 per step) and only 0.1 ms of time.
#include "a3d_common.h"
#include "../../include/act3d_hip.h"

namespace a3d {

constexpr int QS_R = 64;        // rows per block (the workgroup loops over blocks of 64 samples)
constexpr int QS_LD = 68;       // LDS row stride (floats)
constexpr int QS_TILE = QS_R * QS_LD;

// per-row record the forward leaves for the backward: o | Y | y1 | h | o2 (E each) | mean1 rstd1 mean2 rstd2
__host__ __device__ constexpr int qs_save_width(int E) { return 5 * E + 4; }

// Tile loads are split into FETCH (global -> registers: every load of the tile issued back to back, branch-free clamped
// addresses, so a tile costs ONE memory round trip; the first version's guarded scalar loop exposed sixteen -- 80-160 us per
// kernel) and COMMIT (registers -> LDS), so that the next stage's tiles travel while the current stage computes.
struct QsRows { float4 v[4]; };       // rows of a 64 x 64 activation tile: thread -> row t >> 2 (+ 64 rows / ... ), 16-byte columns
struct QsWeight { float v[16]; };     // a 64 x 64 weight tile: only 4-byte aligned inside the flat parameter buffer

// rows r0 .. r0 + 63 of src (row stride ld floats, `cols` valid columns, cols % 4 == 0, 16-byte aligned rows); zero elsewhere
__device__ __forceinline__ QsRows qs_fetch_rows(const float* __restrict__ src, int ld, int r0, int nrows, int cols) {
  QsRows q;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + i * 256;
    const int r = idx >> 4, c = (idx & 15) * 4;
    const bool ok = r0 + r < nrows && c < cols;
    const float4 t = *reinterpret_cast<const float4*>(src + (size_t)(ok ? r0 + r : 0) * ld + (ok ? c : 0));
    q.v[i] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  return q;
}
__device__ __forceinline__ void qs_commit_rows(float* T, const QsRows& q) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + i * 256;
    *reinterpret_cast<float4*>(&T[(idx >> 4) * QS_LD + (idx & 15) * 4]) = q.v[i];
  }
}
__device__ __forceinline__ void qs_load_rows(float* T, const float* __restrict__ src, int ld, int r0, int nrows, int cols) {
  qs_commit_rows(T, qs_fetch_rows(src, ld, r0, nrows, cols));
}
// W[N][K] (row stride ldw) zero padded to 64 x 64
__device__ __forceinline__ QsWeight qs_fetch_weight(const float* __restrict__ W, int ldw, int N, int K) {
  QsWeight w;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = threadIdx.x + i * 256;
    const int n = idx >> 6, k = idx & 63;
    const bool ok = n < N && k < K;
    const float t = W[ok ? (size_t)n * ldw + k : 0];
    w.v[i] = ok ? t : 0.f;
  }
  return w;
}
__device__ __forceinline__ void qs_commit_weight(float* T, const QsWeight& w) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = threadIdx.x + i * 256;
    T[(idx >> 6) * QS_LD + (idx & 63)] = w.v[i];
  }
}
__device__ __forceinline__ void qs_load_weight(float* T, const float* __restrict__ W, int ldw, int N, int K) {
  qs_commit_weight(T, qs_fetch_weight(W, ldw, N, K));
}
// acc[nt] (rows wave*16 + g*4 + r, column nt*16 + li) = sum_k X[row][k] W[col][k]     (y = x W^T)
__device__ __forceinline__ void qs_gemm_nt(const float* Xs, const float* Ws, f32x4 (&acc)[4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const float4 a = *reinterpret_cast<const float4*>(&Xs[(wave * 16 + li) * QS_LD + s4 * 16 + g * 4]);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const float4 b = *reinterpret_cast<const float4*>(&Ws[(nt * 16 + li) * QS_LD + s4 * 16 + g * 4]);
      acc[nt] = mfma_f32_16x16x4(a.x, b.x, acc[nt]);
      acc[nt] = mfma_f32_16x16x4(a.y, b.y, acc[nt]);
      acc[nt] = mfma_f32_16x16x4(a.z, b.z, acc[nt]);
      acc[nt] = mfma_f32_16x16x4(a.w, b.w, acc[nt]);
    }
  }
}
// acc[ct] (rows wave*16 + g*4 + r, column ct*16 + li) = sum_{n in [nlo, nhi)} D[row][n] W[n][col]     (dx = dy W)
__device__ __forceinline__ void qs_gemm_nn(const float* Ds, const float* Ws, f32x4 (&acc)[4], int nlo = 0, int nhi = 64) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int kk = 0; kk < 16; ++kk) {
    const int n = kk * 4 + g;
    const float a = (n >= nlo && n < nhi) ? Ds[(wave * 16 + li) * QS_LD + n] : 0.f;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = mfma_f32_16x16x4(a, Ws[n * QS_LD + ct * 16 + li], acc[ct]);
  }
}
// wacc[ct] (rows n = wave*16 + g*4 + r, column ct*16 + li) += sum_rows D[row][n] X[row][col]     (dW += dY^T X)
__device__ __forceinline__ void qs_wgrad(const float* Ds, const float* Xs, f32x4 (&wacc)[4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
#pragma unroll 4
  for (int kk = 0; kk < 16; ++kk) {
    const int row = kk * 4 + g;
    const float a = Ds[row * QS_LD + wave * 16 + li];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) wacc[ct] = mfma_f32_16x16x4(a, Xs[row * QS_LD + ct * 16 + li], wacc[ct]);
  }
}
// T[row][col] = act(acc + bias[col]) for col < N (0 beyond); act 1 = ReLU
__device__ __forceinline__ void qs_store_tile(float* T, const f32x4 (&acc)[4], const float* __restrict__ bias, int N, int act) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int c = nt * 16 + li;
    const float bv = (bias && c < N) ? bias[c] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = (c < N) ? acc[nt][r] + bv : 0.f;
      if (act == 1) v = fmaxf(v, 0.f);
      T[(wave * 16 + g * 4 + r) * QS_LD + c] = v;
    }
  }
}
// rows of a tile -> global (row stride ld)
__device__ __forceinline__ void qs_write_rows(float* __restrict__ dst, int ld, const float* T, int r0, int nrows, int cols) {
  for (int i = threadIdx.x; i < QS_R * 64; i += blockDim.x) {
    const int r = i >> 6, c = i & 63;
    if (r0 + r < nrows && c < cols) dst[(size_t)(r0 + r) * ld + c] = T[r * QS_LD + c];
  }
}
__device__ __forceinline__ float quad_sum(float v) {      // the four threads of a row (consecutive lanes)
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}
// Y = LayerNorm(A + R) gamma + beta per row (4 threads per row, two-pass); stats -> st[row][0..1]; Y may alias A or R
__device__ __forceinline__ void qs_add_layernorm(const float* A, const float* R, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float* Y, float* st, int E) {
  const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = q + 4 * i;
    v[i] = (c < E) ? A[r * QS_LD + c] + R[r * QS_LD + c] : 0.f;
    s += v[i];
  }
  const float mean = quad_sum(s) / (float)E;
  float qq = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float d = (q + 4 * i < E) ? v[i] - mean : 0.f;
    qq += d * d;
  }
  const float rstd = 1.0f / sqrtf(quad_sum(qq) / (float)E + 1e-5f);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = q + 4 * i;
    if (c < E) Y[r * QS_LD + c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
  }
  if (q == 0) { st[r * 2] = mean; st[r * 2 + 1] = rstd; }
}
// dS = d(A + R) of Y = LayerNorm(A + R) for upstream dY; per-column sums of dY * xhat and dY are ADDED to pg / pb (registers of
// thread (column = t & 63, row group = t >> 6), reduced by the caller at the end).  dS may alias dY.
__device__ __forceinline__ void qs_layernorm_bwd(const float* A, const float* R, const float* __restrict__ gamma, const float* st,
                                                 const float* dY, float* dS, float* rowst, float& pg, float& pb, int E) {
  {
    const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
    const float mean = st[r * 2], rstd = st[r * 2 + 1];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = q + 4 * i;
      if (c < E) {
        const float xh = (A[r * QS_LD + c] + R[r * QS_LD + c] - mean) * rstd;
        const float gy = dY[r * QS_LD + c] * gamma[c];
        s1 += gy;
        s2 += gy * xh;
      }
    }
    s1 = quad_sum(s1) / (float)E;
    s2 = quad_sum(s2) / (float)E;
    if (q == 0) { rowst[r * 2] = s1; rowst[r * 2 + 1] = s2; }
  }
  __syncthreads();
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  if (c < E) {
    const float gm = gamma[c];
    for (int r = rg * 16; r < rg * 16 + 16; ++r) {
      const float mean = st[r * 2], rstd = st[r * 2 + 1];
      const float xh = (A[r * QS_LD + c] + R[r * QS_LD + c] - mean) * rstd;
      const float dy = dY[r * QS_LD + c];
      pg += dy * xh;
      pb += dy;
      dS[r * QS_LD + c] = rstd * (dy * gm - rowst[r * 2] - xh * rowst[r * 2 + 1]);
    }
  }
  __syncthreads();
}
// column sums of a tile added to pb (thread (column, row group) as above)
__device__ __forceinline__ void qs_colsum(const float* T, float& pb) {
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  for (int r = rg * 16; r < rg * 16 + 16; ++r) pb += T[r * QS_LD + c];
}
// The gradient buffers are ADDED to (tied modules accumulate over the pyramid levels).  A read-add-store exposes one memory
// round trip per flush -- twelve of them were a third of the first version's backward kernel; a returnless
// global_atomic_add_f32 is fire-and-forget, and with ONE workgroup per launch and one add per address per launch the result
// does not depend on any ordering (stream order separates the launches): deterministic.
// vectors: NV per-thread partials p[0..NV) (thread = (column t & 63, row group t >> 6)) -> dst[v][c] += sum over the row groups
template <int NV>
__device__ __forceinline__ void qs_flush_vecs(float* const (&dst)[NV], const float (&p)[NV], float* red, int n) {
  __syncthreads();
#pragma unroll
  for (int v = 0; v < NV; ++v) red[v * 256 + threadIdx.x] = p[v];
  __syncthreads();
  for (int i = threadIdx.x; i < NV * 64; i += blockDim.x) {
    const int v = i >> 6, c = i & 63;
    if (c < n) atomicAdd(&dst[v][c], (red[v * 256 + c] + red[v * 256 + 64 + c]) + (red[v * 256 + 128 + c] + red[v * 256 + 192 + c]));
  }
}
// dW[n][k] += wacc (rows n < N, columns k < K)
__device__ __forceinline__ void qs_flush_wgrad(float* __restrict__ dW, int ldw, const f32x4 (&wacc)[4], int N, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int k = ct * 16 + li;
    if (k >= K) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = wave * 16 + g * 4 + r;
      if (n < N) atomicAdd(&dW[(size_t)n * ldw + k], wacc[ct][r]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ pre: q projection + RoPE
__global__ __launch_bounds__(256) void qs_pre_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wq,
                                                         const float* __restrict__ bq, const float* __restrict__ xyz,
                                                         const float* __restrict__ freq, float scale, float* __restrict__ qrot,
                                                         int B, int E, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;
  float* Xs = Ws + QS_TILE;
  float* T = Xs + QS_TILE;
  qs_load_weight(Ws, wq, E, E, E);
  const int half = E >> 1, third = E / 3;
  for (int r0 = 0; r0 < B; r0 += QS_R) {
    qs_load_rows(Xs, x, E, r0, B, E);
    __syncthreads();
    f32x4 acc[4];
    qs_gemm_nt(Xs, Ws, acc);
    qs_store_tile(T, acc, bq, E, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < QS_R * half; i += blockDim.x) {
      const int r = i / half, c = 2 * (i - r * half);
      const float y0 = T[r * QS_LD + c] * scale, y1 = T[r * QS_LD + c + 1] * scale;
      float sn = 0.f, cs = 1.f;
      if (xyz && r0 + r < B) {
        const int axis = c / third;
        fast_sincos(xyz[(size_t)(r0 + r) * 3 + axis] * freq[(c - axis * third) >> 1], &sn, &cs);
      }
      T[r * QS_LD + c] = y0 * cs - y1 * sn;
      T[r * QS_LD + c + 1] = y1 * cs + y0 * sn;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < QS_R * H * 16; i += blockDim.x) {
      const int d = i & 15, h = (i >> 4) % H, r = i / (16 * H);
      if (r0 + r < B) qrot[((size_t)(r0 + r) * H + h) * 16 + d] = d < HD ? T[r * QS_LD + h * HD + d] : 0.f;
    }
    __syncthreads();
  }
}

// dq_pre = scale * R^T sum_s dqp; dW_q += dq_pre^T x, db_q += sum dq_pre; dx += dq_pre W_q (dx: in / out, holds the residual grad)
__global__ __launch_bounds__(256) void qs_pre_bwd_kernel(const float* __restrict__ dqp, int nsplit, const float* __restrict__ xyz,
                                                         const float* __restrict__ freq, float scale, const float* __restrict__ x,
                                                         const float* __restrict__ wq, float* __restrict__ dwq,
                                                         float* __restrict__ dbq, float* __restrict__ dx, int B, int E, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;
  float* Xs = Ws + QS_TILE;
  float* D = Xs + QS_TILE;
  float* Q = D + QS_TILE;
  float* red = Q + QS_TILE;
  qs_load_weight(Ws, wq, E, E, E);
  const int half = E >> 1, third = E / 3;
  f32x4 wacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) wacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float pb = 0.f;
  for (int r0 = 0; r0 < B; r0 += QS_R) {
    const QsRows xr = qs_fetch_rows(x, E, r0, B, E);
    // sum of the key-split partials of the rotated-query gradient: Q[row][h * 16 + d] (H * 16 = 64 columns), every thread's
    // 16 elements x nsplit loads issued in batches of eight
    for (int j = 0; j < 16; ++j) {
      const int i = threadIdx.x + j * 256;
      const int r = i >> 6, c = i & 63;
      const bool ok = r0 + r < B && c < H * 16;
      const float* src = dqp + ((size_t)(ok ? r0 + r : 0) * H) * 16 + (ok ? c : 0);
      float a = 0.f;
      int sp = 0;
      for (; sp + 7 < nsplit; sp += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(sp + u) * B * H * 16];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += t[u];
      }
      for (; sp < nsplit; ++sp) a += src[(size_t)sp * B * H * 16];
      Q[r * QS_LD + c] = ok ? a : 0.f;
    }
    qs_commit_rows(Xs, xr);
    __syncthreads();
    for (int i = threadIdx.x; i < QS_R * 32; i += blockDim.x) {
      const int r = i >> 5, p = i & 31;
      const int c0 = 2 * p, c1 = c0 + 1;
      float y0 = 0.f, y1 = 0.f;
      if (p < half && r0 + r < B) {
        const int h0 = c0 / HD, h1 = c1 / HD;
        const float g0 = Q[r * QS_LD + h0 * 16 + (c0 - h0 * HD)], g1 = Q[r * QS_LD + h1 * 16 + (c1 - h1 * HD)];
        y0 = g0; y1 = g1;
        if (xyz) {
          const int axis = c0 / third;
          float sn, cs;
          fast_sincos(xyz[(size_t)(r0 + r) * 3 + axis] * freq[(c0 - axis * third) >> 1], &sn, &cs);
          y0 = cs * g0 + sn * g1;
          y1 = cs * g1 - sn * g0;
        }
        y0 *= scale; y1 *= scale;
      }
      D[r * QS_LD + c0] = y0;
      D[r * QS_LD + c1] = y1;
    }
    __syncthreads();
    qs_wgrad(D, Xs, wacc);
    qs_colsum(D, pb);
    f32x4 acc[4];
    qs_gemm_nn(D, Ws, acc);
    {
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int c = ct * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = r0 + wave * 16 + g * 4 + r;
          if (row < B && c < E) dx[(size_t)row * E + c] += acc[ct][r];
        }
      }
    }
    __syncthreads();
  }
  qs_flush_wgrad(dwq, E, wacc, E, E);
  float* const dst[1] = {dbq};
  const float pv[1] = {pb};
  qs_flush_vecs<1>(dst, pv, red, E);
}

// ------------------------------------------------------------------------------------------------ post: forward
__global__ __launch_bounds__(256) void qs_post_fwd_kernel(const float* __restrict__ xbar, const float* __restrict__ resid,
                                                          a3d_qs_params p, float* __restrict__ save, float* __restrict__ y_out, int B,
                                                          int E, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;
  float* T0 = Ws + QS_TILE;
  float* T1 = T0 + QS_TILE;
  float* T2 = T1 + QS_TILE;
  float* st = T2 + QS_TILE;           // [64][2]
  const int SW = qs_save_width(E);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  for (int r0 = 0; r0 < B; r0 += QS_R) {
    // ---- o[row][h*15 + d] = W_v[h*15 + d] . xbar[row][h] + b_v: per head, the column tiles that hold its 15 channels -> T1
    qs_load_weight(Ws, p.wv, E, E, E);
    for (int h = 0; h < H; ++h) {
      qs_load_rows(T0, xbar + (size_t)h * E, H * E, r0, B, E);
      __syncthreads();
      f32x4 acc[4];
      qs_gemm_nt(T0, Ws, acc);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int c = nt * 16 + li;
        if (c >= h * HD && c < (h + 1) * HD) {
          const float bv = p.bv ? p.bv[c] : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) T1[(wave * 16 + g * 4 + r) * QS_LD + c] = acc[nt][r] + bv;
        }
      }
      __syncthreads();
    }
    for (int i = threadIdx.x; i < QS_R * (64 - E); i += blockDim.x) T1[(i / (64 - E)) * QS_LD + E + i % (64 - E)] = 0.f;
    __syncthreads();
    qs_write_rows(save, SW, T1, r0, B, E);                                           // o
    // ---- Y = o W_o^T + b_o -> T2;  x1 = LayerNorm(resid + Y) -> T0
    qs_load_weight(Ws, p.wo, E, E, E);
    qs_load_rows(T0, resid, E, r0, B, E);
    __syncthreads();
    {
      f32x4 acc[4];
      qs_gemm_nt(T1, Ws, acc);
      qs_store_tile(T2, acc, p.bo, E, 0);
    }
    __syncthreads();
    qs_write_rows(save + E, SW, T2, r0, B, E);                                       // Y
    qs_add_layernorm(T0, T2, p.g1, p.b1, T0, st, E);
    __syncthreads();
    qs_write_rows(save + 2 * E, SW, T0, r0, B, E);                                   // y1
    if ((int)threadIdx.x < QS_R && r0 + (int)threadIdx.x < B) {
      save[(size_t)(r0 + threadIdx.x) * SW + 5 * E] = st[threadIdx.x * 2];
      save[(size_t)(r0 + threadIdx.x) * SW + 5 * E + 1] = st[threadIdx.x * 2 + 1];
    }
    // ---- h = relu(y1 W_1^T + c_1) -> T1;  o2 = h W_2^T + c_2 -> T2;  y2 = LayerNorm(y1 + o2)
    qs_load_weight(Ws, p.w1, E, E, E);
    __syncthreads();
    {
      f32x4 acc[4];
      qs_gemm_nt(T0, Ws, acc);
      qs_store_tile(T1, acc, p.c1, E, 1);
    }
    __syncthreads();
    qs_write_rows(save + 3 * E, SW, T1, r0, B, E);                                   // h
    qs_load_weight(Ws, p.w2, E, E, E);
    __syncthreads();
    {
      f32x4 acc[4];
      qs_gemm_nt(T1, Ws, acc);
      qs_store_tile(T2, acc, p.c2, E, 0);
    }
    __syncthreads();
    qs_write_rows(save + 4 * E, SW, T2, r0, B, E);                                   // o2
    qs_add_layernorm(T0, T2, p.g2, p.b2, T0, st, E);
    __syncthreads();
    qs_write_rows(y_out, E, T0, r0, B, E);
    if ((int)threadIdx.x < QS_R && r0 + (int)threadIdx.x < B) {
      save[(size_t)(r0 + threadIdx.x) * SW + 5 * E + 2] = st[threadIdx.x * 2];
      save[(size_t)(r0 + threadIdx.x) * SW + 5 * E + 3] = st[threadIdx.x * 2 + 1];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ post: backward
__global__ __launch_bounds__(256) void qs_post_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ resid,
                                                          const float* __restrict__ xbar, const float* __restrict__ save,
                                                          a3d_qs_params p, a3d_qs_grads gr, float* __restrict__ dxbar,
                                                          float* __restrict__ cD, float* __restrict__ dresid, int B, int E, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                   // the weight of the current product
  float* XA = Ws + QS_TILE;           // a saved activation (operand of a weight gradient / LayerNorm input)
  float* XB = XA + QS_TILE;           // second LayerNorm input
  float* D = XB + QS_TILE;            // upstream gradient of the current stage
  float* G = D + QS_TILE;             // produced gradient
  float* st = G + QS_TILE;            // [64][2] saved LayerNorm statistics
  float* rowst = st + 2 * QS_R;       // [64][2]
  float* red = rowst + 2 * QS_R;      // [8][256]
  const int SW = qs_save_width(E);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
  f32x4 w2a[4], w1a[4], woa[4], wva[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { w2a[i] = f32x4{0.f, 0.f, 0.f, 0.f}; w1a[i] = w2a[i]; woa[i] = w2a[i]; wva[i] = w2a[i]; }
  float pg2 = 0.f, pb2 = 0.f, pc2 = 0.f, pc1 = 0.f, pg1 = 0.f, pb1 = 0.f, pbo = 0.f, pbv = 0.f;
  auto load_stats = [&](int r0, int off) {
    if ((int)threadIdx.x < QS_R) {
      const bool ok = r0 + (int)threadIdx.x < B;
      st[threadIdx.x * 2] = ok ? save[(size_t)(r0 + threadIdx.x) * SW + 5 * E + off] : 0.f;
      st[threadIdx.x * 2 + 1] = ok ? save[(size_t)(r0 + threadIdx.x) * SW + 5 * E + off + 1] : 0.f;
    }
  };
  for (int r0 = 0; r0 < B; r0 += QS_R) {
    // ---- y2 = LayerNorm(y1 + o2): ds2 -> D
    qs_load_rows(XA, save + 2 * E, SW, r0, B, E);      // y1
    qs_load_rows(XB, save + 4 * E, SW, r0, B, E);      // o2
    qs_load_rows(D, dy, E, r0, B, E);
    load_stats(r0, 2);
    __syncthreads();
    qs_layernorm_bwd(XA, XB, p.g2, st, D, D, rowst, pg2, pb2, E);
    // ---- o2 = h W_2^T + c_2:  dW_2 += ds2^T h, dc_2 += sum ds2, dh = (ds2 W_2) * (h > 0) -> G
    qs_load_rows(XB, save + 3 * E, SW, r0, B, E);      // h
    qs_load_weight(Ws, p.w2, E, E, E);
    __syncthreads();
    qs_wgrad(D, XB, w2a);
    qs_colsum(D, pc2);
    {
      f32x4 acc[4];
      qs_gemm_nn(D, Ws, acc);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int c = ct * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wave * 16 + g * 4 + r;
          G[row * QS_LD + c] = (c < E && XB[row * QS_LD + c] > 0.f) ? acc[ct][r] : 0.f;
        }
      }
    }
    __syncthreads();
    // ---- h = relu(y1 W_1^T + c_1):  dW_1 += dh^T y1, dc_1 += sum dh, dy1 = dh W_1 + ds2 -> D
    qs_load_weight(Ws, p.w1, E, E, E);
    __syncthreads();
    qs_wgrad(G, XA, w1a);
    qs_colsum(G, pc1);
    {
      f32x4 acc[4];
      qs_gemm_nn(G, Ws, acc);
      __syncthreads();                                  // every wave is done reading D's old rows through qs_wgrad / colsum above
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int c = ct * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wave * 16 + g * 4 + r;
          D[row * QS_LD + c] = (c < E) ? D[row * QS_LD + c] + acc[ct][r] : 0.f;
        }
      }
    }
    // ---- y1 = LayerNorm(resid + Y): ds1 -> D (= the residual branch's d(x))
    qs_load_rows(XA, resid, E, r0, B, E);
    qs_load_rows(XB, save + E, SW, r0, B, E);          // Y
    load_stats(r0, 0);
    __syncthreads();
    qs_layernorm_bwd(XA, XB, p.g1, st, D, D, rowst, pg1, pb1, E);
    qs_write_rows(dresid, E, D, r0, B, E);
    // ---- Y = o W_o^T + b_o:  dW_o += ds1^T o, db_o += sum ds1, dO = ds1 W_o -> G
    qs_load_rows(XA, save, SW, r0, B, E);              // o
    qs_load_weight(Ws, p.wo, E, E, E);
    __syncthreads();
    qs_wgrad(D, XA, woa);
    qs_colsum(D, pbo);
    {
      f32x4 acc[4];
      qs_gemm_nn(D, Ws, acc);
      qs_store_tile(G, acc, nullptr, E, 0);
    }
    __syncthreads();
    qs_colsum(G, pbv);
    // ---- o_h = W_v,h xbar_h + b_v,h per head:  dxbar_h = dO_h W_v,h,  cD_h = dxbar_h . xbar_h,  dW_v,h += dO_h^T xbar_h
    qs_load_weight(Ws, p.wv, E, E, E);
    for (int h = 0; h < H; ++h) {
      qs_load_rows(XA, xbar + (size_t)h * E, H * E, r0, B, E);
      __syncthreads();
      {
        // weight gradient of the head's 15 rows: the A operand is dO restricted to them (other rows of the tile add zero)
        const int nrow = wave * 16 + li;
        const bool mine = nrow >= h * HD && nrow < (h + 1) * HD;
#pragma unroll 4
        for (int kk = 0; kk < 16; ++kk) {
          const int row = kk * 4 + g;
          const float a = mine ? G[row * QS_LD + nrow] : 0.f;
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) wva[ct] = mfma_f32_16x16x4(a, XA[row * QS_LD + ct * 16 + li], wva[ct]);
        }
        f32x4 acc[4];
        qs_gemm_nn(G, Ws, acc, h * HD, (h + 1) * HD);
        qs_store_tile(XB, acc, nullptr, E, 0);
      }
      __syncthreads();
      {
        const int r = threadIdx.x >> 2, q = threadIdx.x & 3;
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int c = q + 4 * i;
          if (c < E) dot += XB[r * QS_LD + c] * XA[r * QS_LD + c];
        }
        dot = quad_sum(dot);
        if (q == 0 && r0 + r < B) cD[(size_t)(r0 + r) * H + h] = dot;
      }
      qs_write_rows(dxbar + (size_t)h * E, H * E, XB, r0, B, E);
      __syncthreads();
    }
  }
  qs_flush_wgrad(gr.dw2, E, w2a, E, E);
  qs_flush_wgrad(gr.dw1, E, w1a, E, E);
  qs_flush_wgrad(gr.dwo, E, woa, E, E);
  qs_flush_wgrad(gr.dwv, E, wva, E, E);
  float* const dst[8] = {gr.dg2, gr.db2, gr.dc2, gr.dc1, gr.dg1, gr.db1, gr.dbo, gr.dbv};
  const float pv[8] = {pg2, pb2, pc2, pc1, pg1, pb1, pbo, pbv};
  qs_flush_vecs<8>(dst, pv, red, E);
}

}  // namespace a3d

using namespace a3d;

static int qs_check(const char* fn, int B, int E, int H) {
  if (B <= 0 || E <= 0 || E > 60 || (E % 12) != 0 || H <= 0 || H > 4 || H * HD != E) {
    set_error("%s: bad shape (B=%d E=%d H=%d; E = 15 H <= 60, E %% 12 == 0)", fn, B, E, H);
    return A3D_ERR_ARG;
  }
  return A3D_OK;
}
template <typename K>
static void qs_allow_lds(K kernel) { (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024); }

extern "C" size_t a3d_qs_save_floats(int B, int E) { return (B <= 0 || E <= 0) ? 0 : (size_t)B * qs_save_width(E); }

extern "C" int a3d_qs_pre_fwd(const float* x, const float* wq, const float* bq, const float* xyz, const float* freq, float scale,
                              float* qrot, int B, int E, int H, void* stream) {
  int rc = qs_check("a3d_qs_pre_fwd", B, E, H);
  if (rc) return rc;
  if (!x || !wq || !qrot || (xyz && !freq) || (((uintptr_t)x) & 15)) { set_error("a3d_qs_pre_fwd: null / misaligned pointer (x: 16 bytes)"); return A3D_ERR_ARG; }
  static bool once = false;
  if (!once) { qs_allow_lds(qs_pre_fwd_kernel); qs_allow_lds(qs_pre_bwd_kernel); qs_allow_lds(qs_post_fwd_kernel); qs_allow_lds(qs_post_bwd_kernel); once = true; }
  hipLaunchKernelGGL(qs_pre_fwd_kernel, dim3(1), dim3(256), 3 * QS_TILE * sizeof(float), (hipStream_t)stream, x, wq, bq, xyz, freq, scale,
                     qrot, B, E, H);
  return check_launch("a3d_qs_pre_fwd");
}

extern "C" int a3d_qs_pre_bwd(const float* dqp, int nsplit, const float* xyz, const float* freq, float scale, const float* x,
                              const float* wq, float* dwq, float* dbq, float* dx, int B, int E, int H, void* stream) {
  int rc = qs_check("a3d_qs_pre_bwd", B, E, H);
  if (rc) return rc;
  if (!dqp || nsplit < 1 || !x || !wq || !dwq || !dbq || !dx || (xyz && !freq) || (((uintptr_t)x) & 15)) {
    set_error("a3d_qs_pre_bwd: bad argument (x 16-byte aligned)");
    return A3D_ERR_ARG;
  }
  static bool once = false;
  if (!once) { qs_allow_lds(qs_pre_bwd_kernel); once = true; }
  hipLaunchKernelGGL(qs_pre_bwd_kernel, dim3(1), dim3(256), (4 * QS_TILE + 256) * sizeof(float), (hipStream_t)stream, dqp, nsplit, xyz,
                     freq, scale, x, wq, dwq, dbq, dx, B, E, H);
  return check_launch("a3d_qs_pre_bwd");
}

static bool qs_params_ok(const a3d_qs_params* p) {
  return p && p->wv && p->wo && p->g1 && p->b1 && p->w1 && p->w2 && p->g2 && p->b2;
}

extern "C" int a3d_qs_post_fwd(const float* xbar, const float* resid, const a3d_qs_params* p, float* save, float* y, int B, int E,
                               int H, void* stream) {
  int rc = qs_check("a3d_qs_post_fwd", B, E, H);
  if (rc) return rc;
  if (!xbar || !resid || !qs_params_ok(p) || !save || !y || ((((uintptr_t)xbar) | ((uintptr_t)resid) | ((uintptr_t)save)) & 15)) {
    set_error("a3d_qs_post_fwd: null / misaligned pointer (xbar, resid, save: 16 bytes)");
    return A3D_ERR_ARG;
  }
  static bool once = false;
  if (!once) { qs_allow_lds(qs_post_fwd_kernel); once = true; }
  hipLaunchKernelGGL(qs_post_fwd_kernel, dim3(1), dim3(256), (4 * QS_TILE + 2 * QS_R) * sizeof(float), (hipStream_t)stream, xbar, resid, *p,
                     save, y, B, E, H);
  return check_launch("a3d_qs_post_fwd");
}

extern "C" int a3d_qs_post_bwd(const float* dy, const float* resid, const float* xbar, const float* save, const a3d_qs_params* p,
                               const a3d_qs_grads* gr, float* dxbar, float* cD, float* dresid, int B, int E, int H, void* stream) {
  int rc = qs_check("a3d_qs_post_bwd", B, E, H);
  if (rc) return rc;
  if (!dy || !resid || !xbar || !save || !qs_params_ok(p) || !gr || !gr->dwv || !gr->dbv || !gr->dwo || !gr->dbo || !gr->dg1 || !gr->db1 ||
      !gr->dw1 || !gr->dc1 || !gr->dw2 || !gr->dc2 || !gr->dg2 || !gr->db2 || !dxbar || !cD || !dresid ||
      ((((uintptr_t)dy) | ((uintptr_t)resid) | ((uintptr_t)xbar) | ((uintptr_t)save)) & 15)) {
    set_error("a3d_qs_post_bwd: null / misaligned pointer (dy, resid, xbar, save: 16 bytes)");
    return A3D_ERR_ARG;
  }
  static bool once = false;
  if (!once) { qs_allow_lds(qs_post_bwd_kernel); once = true; }
  hipLaunchKernelGGL(qs_post_bwd_kernel, dim3(1), dim3(256), (5 * QS_TILE + 4 * QS_R + 8 * 256) * sizeof(float), (hipStream_t)stream, dy, resid,
                     xbar, save, *p, *gr, dxbar, cD, dresid, B, E, H);
  return check_launch("a3d_qs_post_bwd");
}
