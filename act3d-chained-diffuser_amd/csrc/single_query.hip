// Attention of ONE query per sample over the scene context -- Act3D's query stream (act3d.py:467-480: the learned query
// token cross-attends to the level's context, RelativeCrossAttentionLayer layers.py:293-310) -- without ever materialising
// the projected keys / values.
//
// With a single query the value projection commutes with the softmax-weighted sum,
//     o_h = sum_k p_k (W_v,h x_k + b_v,h) = W_v,h (sum_k p_k x_k) + b_v,h          (sum_k p_k = 1),
// so the forward needs, per head, only the p-weighted mean xbar_h of the RAW context rows; the key projection
// (W_k x_k + b_k, rotated by the key's xyz) is recomputed per 64-key tile in LDS and consumed on the spot.  The general
// path (rope.hip + attention*.hip) writes four operand formats of K and V (1152 B per key) for the forward and reads /
// writes as much again in the backward, to serve 1 query: 2 x 6 blocks x ~0.33 ms per B = 64 step.  Here a block is one
// pass over the context (240 B per key) forward, and one pass backward that recomputes the keys, forms
//     ds_k,h = p_k,h (dxbar_h . x_k - dxbar_h . xbar_h),   dxbar_h = W_v,h^T dO_h,
//     dX_k   = sum_h p_k,h dxbar_h  +  (R_k^T (ds_k (x) q)) W_k,
//     dW_k  += (R_k^T (ds_k (x) q))^T X      (accumulated over the workgroup's tiles on the MFMA pipe, two-stage reduce),
// and the rotated-query gradient dq_h = sum_k ds_k,h k_k,h.  All contractions use the exact-f32 MFMA (as linear.hip).
// Restricted to E <= 64 channels, H <= 4 heads (Act3D: E = 60, H = 4), no key-padding mask (the query stream has none).
#include "a3d_common.h"
#include "../../include/act3d_hip.h"

namespace a3d {

constexpr int SQ_T = 64;       // keys per tile
constexpr int SQ_LD = 68;      // LDS row stride (floats) of the [64][<=64] tiles
constexpr int SQ_NT = 4;       // 16-column tiles of a row (E <= 64)

// round-5 candidates for the backward key pass, each an A/B switch at build time (A3D_HIPCC_FLAGS="-DSQ_DX_LDS=0 ..."):
//   SQ_DX_LDS  the dX tile leaves through an LDS tile as coalesced float4 rows (a tile is ONE contiguous 64 E-float block of dX),
//              with sum_h p_h dxbar_h as a fifth k-step of its GEMM instead of 128 LDS reads + 64 fma per lane; before: 16 guarded
//              4-byte stores per lane
//   SQ_DQ_PAR  the rotated-query gradient on all 64 lanes of the head's wave (key quarters) instead of a 64-step chain on 15 lanes
#ifndef SQ_DX_LDS
#define SQ_DX_LDS 1
#endif
#ifndef SQ_DQ_PAR
#define SQ_DQ_PAR 1
#endif

// context rows n0 .. n0+63 of sample b (zero beyond S / E; column E := 1 for valid rows if `ones`): global -> registers
// (issued one tile ahead, so the HBM round trip hides behind the previous tile's arithmetic) -> Xs[64][SQ_LD]
struct SqRows { float4 v[4]; };
__device__ __forceinline__ SqRows sq_load_rows(const float* __restrict__ X, int b, int n0, int S, int E, bool ones) {
  SqRows s;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + i * 256;
    const int r = idx >> 4, c = (idx & 15) * 4;
    const int n = n0 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < S && c < E) v = *reinterpret_cast<const float4*>(X + ((size_t)b * S + n) * E + c);      // E % 4 == 0
    if (ones && n < S) {
      if (c + 0 == E) v.x = 1.f;
      if (c + 1 == E) v.y = 1.f;
      if (c + 2 == E) v.z = 1.f;
      if (c + 3 == E) v.w = 1.f;
    }
    s.v[i] = v;
  }
  return s;
}
__device__ __forceinline__ void sq_store_rows(float* Xs, const SqRows& s) {
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + i * 256;
    *reinterpret_cast<float4*>(&Xs[(idx >> 4) * SQ_LD + (idx & 15) * 4]) = s.v[i];
  }
}

// Ws[64][SQ_LD] <- W[E][E] (zero padded); W may be only 4-byte aligned (flat parameter buffer)
__device__ __forceinline__ void sq_stage_weight(float* Ws, const float* __restrict__ W, int ldw, int E) {
  for (int idx = threadIdx.x; idx < SQ_T * 64; idx += blockDim.x) {
    const int j = idx >> 6, c = idx & 63;
    Ws[j * SQ_LD + c] = (j < E && c < E) ? W[(size_t)j * ldw + c] : 0.f;
  }
}

// T[64][SQ_LD] = rope(Xs W^T + bias) for the tile's rows (rows >= S zero); Ws holds W (rows = output channels).
// KEEP: the (cos, sin) of the up to 8 (row, pair) items this thread rotates are left in `keep` (registers, constant
// indexing) for the inverse rotation of the backward, which visits the same items in the same order.
constexpr int SQ_ROT = 8;      // (row, channel pair) items per thread: 64 rows x E / 2 <= 30 pairs over 256 threads
template <bool KEEP>
__device__ __forceinline__ void sq_project_rope(float* T, const float* Xs, const float* Ws, const float* __restrict__ bias,
                                                const float* __restrict__ xyz, const float* __restrict__ freq, int b, int n0,
                                                int S, int E, float (&keep)[SQ_ROT][2]) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  f32x4 acc[SQ_NT];
#pragma unroll
  for (int i = 0; i < SQ_NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // 16 k-steps over the 64 staged columns (columns >= E are zero on at least one operand): a constant trip count lets the
  // compiler hoist the LDS operand reads over the MFMAs
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const float a = Xs[(wave * 16 + li) * SQ_LD + kk * 4 + g];
#pragma unroll
    for (int nt = 0; nt < SQ_NT; ++nt) acc[nt] = mfma_f32_16x16x4(a, Ws[(nt * 16 + li) * SQ_LD + kk * 4 + g], acc[nt]);
  }
  // the angles of this thread's items while the MFMAs drain (registers only: no hazard with T)
  const int half = E >> 1, third = E / 3;
  float cs_[SQ_ROT], sn_[SQ_ROT];
  if (xyz) {
#pragma unroll
    for (int it = 0; it < SQ_ROT; ++it) {
      const int idx = t + it * 256;
      const int r = idx / half, p = idx - r * half;
      const int n = n0 + r;
      cs_[it] = 1.f; sn_[it] = 0.f;
      if (idx < SQ_T * half && n < S) {
        const int c = 2 * p;
        const int axis = c / third;
        const int kf = (c - axis * third) >> 1;
        fast_sincos(xyz[((size_t)b * S + n) * 3 + axis] * freq[kf], &sn_[it], &cs_[it]);
      }
      if (KEEP) { keep[it][0] = cs_[it]; keep[it][1] = sn_[it]; }
    }
  }
#pragma unroll
  for (int nt = 0; nt < SQ_NT; ++nt) {
    const int c = nt * 16 + li;
    const float bv = (c < E && bias) ? bias[c] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 16 + g * 4 + r;
      T[row * SQ_LD + c] = (c < E && n0 + row < S) ? acc[nt][r] + bv : 0.f;
    }
  }
  __syncthreads();
  if (xyz) {
#pragma unroll
    for (int it = 0; it < SQ_ROT; ++it) {
      const int idx = t + it * 256;
      if (idx >= SQ_T * half) break;
      const int r = idx / half, c = 2 * (idx - r * half);
      const float y0 = T[r * SQ_LD + c], y1 = T[r * SQ_LD + c + 1];
      T[r * SQ_LD + c] = y0 * cs_[it] - y1 * sn_[it];
      T[r * SQ_LD + c + 1] = y1 * cs_[it] + y0 * sn_[it];
    }
    __syncthreads();
  }
}

// out[h][key] = sum_c A[key][c] M[h][c] for h < 4 via one 16-column MFMA tile; M in LDS as [16][SQ_LD] (rows >= H zero)
__device__ __forceinline__ void sq_rows_times_heads(const float* A, const float* M, float* out /*[4][64]*/, int E) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < 16; kk += 2) {          // M's columns >= E are zero
    acc = mfma_f32_16x16x4(A[(wave * 16 + li) * SQ_LD + kk * 4 + g], M[li * SQ_LD + kk * 4 + g], acc);
    acc1 = mfma_f32_16x16x4(A[(wave * 16 + li) * SQ_LD + kk * 4 + 4 + g], M[li * SQ_LD + kk * 4 + 4 + g], acc1);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] += acc1[r];
  if (li < 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) out[li * SQ_T + wave * 16 + g * 4 + r] = acc[r];
  }
}

// ------------------------------------------------------------------------------------------------ forward
// grid (nsplit, B); partial [B][nsplit][H][E + 2] = {m, l, xbar[E]} per head
// EC: E as a compile-time constant (60 = Act3D; 0 = run-time E): the rotation loops divide by E / 2 and E / 3 per item
// (rope.hip's proj_rope_split_kernel has the numbers)
template <int EC>
__global__ __launch_bounds__(256) void sq_fwd_kernel(const float* __restrict__ X, const float* __restrict__ xyz,
                                                     const float* __restrict__ Wk, int ldw, const float* __restrict__ bk,
                                                     const float* __restrict__ qrot, const float* __restrict__ freq,
                                                     float* __restrict__ part, int B, int S, int E_rt, int H_rt, int nsplit) {
  const int E = EC > 0 ? EC : E_rt, H = EC > 0 ? EC / HD : H_rt;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;
  float* Ws = Xs + SQ_T * SQ_LD;
  float* T = Ws + SQ_T * SQ_LD;
  float* Qm = T + SQ_T * SQ_LD;            // [16][SQ_LD]: row h = the head's rotated query in its channel range, else 0
  float* sS = Qm + 16 * SQ_LD;             // [4][64]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int b = blockIdx.y, sp = blockIdx.x;
  sq_stage_weight(Ws, Wk, ldw, E);
  for (int idx = t; idx < 16 * SQ_LD; idx += 256) {
    const int h = idx / SQ_LD, c = idx - h * SQ_LD;
    const int d = c - h * HD;
    Qm[idx] = (h < H && c < E && d >= 0 && d < HD) ? qrot[((size_t)b * H + h) * 16 + d] : 0.f;
  }
  const int ntile = (S + SQ_T - 1) / SQ_T;
  const int t_beg = (int)((long long)ntile * sp / nsplit), t_end = (int)((long long)ntile * (sp + 1) / nsplit);
  float m_run = -INFINITY, l_run = 0.f, xb = 0.f;      // wave = head; lane = channel of xbar
  SqRows rows;
  if (t_beg < t_end) rows = sq_load_rows(X, b, t_beg * SQ_T, S, E, false);
  __syncthreads();
  for (int tile = t_beg; tile < t_end; ++tile) {
    const int n0 = tile * SQ_T;
    sq_store_rows(Xs, rows);
    if (tile + 1 < t_end) rows = sq_load_rows(X, b, n0 + SQ_T, S, E, false);
    __syncthreads();
    float unused[SQ_ROT][2];
    sq_project_rope<false>(T, Xs, Ws, bk, xyz, freq, b, n0, S, E, unused);
    sq_rows_times_heads(T, Qm, sS, E);
    __syncthreads();
    float alpha = 1.f;
    if (wave < H) {
      const float s = (n0 + lane < S) ? sS[wave * SQ_T + lane] : -INFINITY;
      const float m_new = fmaxf(m_run, wave_max(s));
      const float p = (s == -INFINITY) ? 0.f : __expf(s - m_new);
      alpha = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
      l_run = l_run * alpha + wave_sum(p);
      m_run = m_new;
      sS[wave * SQ_T + lane] = p;            // own row, own element: no hazard with the other waves
    }
    __syncthreads();
    if (wave < H && lane < E) {
      float a = 0.f;
#pragma unroll 8
      for (int k = 0; k < SQ_T; ++k) a += sS[wave * SQ_T + k] * Xs[k * SQ_LD + lane];
      xb = xb * alpha + a;
    }
    __syncthreads();
  }
  if (wave < H) {
    float* o = part + (((size_t)b * nsplit + sp) * H + wave) * (E + 2);
    if (lane == 0) { o[0] = m_run; o[1] = l_run; }
    if (lane < E) o[2 + lane] = xb;
  }
}

// xbar [B][H][E], lse [B][H] from the key-split partials
__global__ __launch_bounds__(256) void sq_combine_kernel(const float* __restrict__ part, float* __restrict__ xbar,
                                                         float* __restrict__ lse, int B, int H, int E, int nsplit) {
  const int bh = blockIdx.x;                 // one workgroup per (b, h)
  const int b = bh / H, h = bh - b * H;
  float m = -INFINITY;
  for (int s = 0; s < nsplit; ++s) m = fmaxf(m, part[(((size_t)b * nsplit + s) * H + h) * (E + 2)]);
  const float m_use = (m == -INFINITY) ? 0.f : m;
  float l = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float* p = part + (((size_t)b * nsplit + s) * H + h) * (E + 2);
    l += p[1] * __expf(p[0] - m_use);
  }
  for (int c = threadIdx.x; c < E; c += blockDim.x) {
    float a = 0.f;
    for (int s = 0; s < nsplit; ++s) {
      const float* p = part + (((size_t)b * nsplit + s) * H + h) * (E + 2);
      a += p[2 + c] * __expf(p[0] - m_use);
    }
    xbar[(size_t)bh * E + c] = l > 0.f ? a / l : 0.f;
  }
  if (threadIdx.x == 0) lse[bh] = l > 0.f ? m + logf(l) : -INFINITY;
}

// o[b][hd] = W_v[hd] . xbar[b][head(hd)] + b_v[hd]
__global__ void sq_vproj_kernel(const float* __restrict__ xbar, const float* __restrict__ Wv, int ldw, const float* __restrict__ bv,
                                float* __restrict__ o, int B, int H, int E) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * E) return;
  const int b = i / E, hd = i - b * E, h = hd / HD;
  const float* x = xbar + ((size_t)b * H + h) * E;
  const float* w = Wv + (size_t)hd * ldw;
  float a = bv ? bv[hd] : 0.f;
  for (int c = 0; c < E; ++c) a += w[c] * x[c];
  o[i] = a;
}

// dxbar[b][h][c] = sum_d dO[b][h*15+d] W_v[h*15+d][c];  cD[b][h] = dxbar . xbar
// dW_v[hd][c] += sum_b dO[b][hd] xbar[b][h][c];  db_v[hd] += sum_b dO[b][hd]          (grid: B*H blocks, then E blocks)
__global__ __launch_bounds__(64) void sq_vproj_bwd_kernel(const float* __restrict__ dO, const float* __restrict__ xbar,
                                                          const float* __restrict__ Wv, int ldw, float* __restrict__ dxbar,
                                                          float* __restrict__ cD, float* __restrict__ dWv, int lddw,
                                                          float* __restrict__ dbv, int B, int H, int E) {
  const int lane = threadIdx.x;
  if ((int)blockIdx.x < B * H) {
    const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
    float a = 0.f;
    if (lane < E) {
      for (int d = 0; d < HD; ++d) a += dO[(size_t)b * E + h * HD + d] * Wv[(size_t)(h * HD + d) * ldw + lane];
      dxbar[(size_t)bh * E + lane] = a;
    }
    const float dot = wave_sum(lane < E ? a * xbar[(size_t)bh * E + lane] : 0.f);
    if (lane == 0) cD[bh] = dot;
  } else {
    const int hd = blockIdx.x - B * H, h = hd / HD;
    float a = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
      const float g = dO[(size_t)b * E + hd];
      sb += g;
      if (lane < E) a += g * xbar[((size_t)b * H + h) * E + lane];
    }
    if (lane < E) dWv[(size_t)hd * lddw + lane] += a;
    if (lane == 0 && dbv) dbv[hd] += sb;
  }
}

// Phase timestamps (wall_clock64, 100 MHz) of workgroup (0, 0) of the last sq_bwd launch while a3d_dbg_sq_prof(1, ..) is armed:
// development aid (profiles/sq_bwd_phases.py), no effect on results.  Marks: 0 entry, 1 weights / query / dxbar staged, then for
// the workgroup's LAST tile 2 rows in LDS, 3 keys projected + rotated, 4 scores and dp, 5 p and ds, 6 rotated-query gradient,
// 7 inverse rotation, 8 dX tile stored, 9 dW accumulated; 10 exit; 11 = tiles this workgroup walked.
__device__ long long g_sq_prof[16];
__device__ int g_sq_prof_on;
#define SQ_MARK(i) do { if (prof_on && threadIdx.x == 0) g_sq_prof[i] = wall_clock64(); } while (0)

// ------------------------------------------------------------------------------------------------ backward
// grid (nsplit, B).  dX [B][S][E] (written, every row once); wpart [B * nsplit][E][E + 1] (dW_k | db_k partials);
// dqp [nsplit][B][H][1][16] (rotated-query gradient partials, the layout a3d_rope_merge_bwd reads with Npad = 1)
template <int EC>
__global__ __launch_bounds__(256, 2) void sq_bwd_kernel(const float* __restrict__ X, const float* __restrict__ xyz,
                                                     const float* __restrict__ Wk, int ldw, const float* __restrict__ bk,
                                                     const float* __restrict__ qrot, const float* __restrict__ freq,
                                                     const float* __restrict__ lse, const float* __restrict__ dxbar,
                                                     const float* __restrict__ cD, float* __restrict__ dX,
                                                     float* __restrict__ wpart, float* __restrict__ dqp, int B, int S, int E_rt,
                                                     int H_rt, int nsplit, int acc_dx) {
  const int E = EC > 0 ? EC : E_rt, H = EC > 0 ? EC / HD : H_rt;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;
  float* Ws = Xs + SQ_T * SQ_LD;
  float* T = Ws + SQ_T * SQ_LD;            // rotated keys, then (in place) the gradient w.r.t. the un-rotated projection
  float* Qm = T + SQ_T * SQ_LD;            // [16][SQ_LD]
  float* Dm = Qm + 16 * SQ_LD;             // [16][SQ_LD]: row h = dxbar[b][h]
  float* sS = Dm + 16 * SQ_LD;             // [4][64] scores -> p
  float* dS = sS + 4 * SQ_T;               // [4][64] dp -> ds
  float* Ys = dS + 4 * SQ_T;               // [64][SQ_LD] the dX tile on its way out (SQ_DX_LDS)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int b = blockIdx.y, sp = blockIdx.x;
  const bool prof_on = g_sq_prof_on != 0 && blockIdx.x == 0 && blockIdx.y == 0;
  SQ_MARK(0);
  sq_stage_weight(Ws, Wk, ldw, E);
  for (int idx = t; idx < 16 * SQ_LD; idx += 256) {
    const int h = idx / SQ_LD, c = idx - h * SQ_LD;
    const int d = c - h * HD;
    Qm[idx] = (h < H && c < E && d >= 0 && d < HD) ? qrot[((size_t)b * H + h) * 16 + d] : 0.f;
    Dm[idx] = (h < H && c < E) ? dxbar[((size_t)b * H + h) * E + c] : 0.f;
  }
  const float lse_h = wave < H ? lse[(size_t)b * H + wave] : 0.f;
  const float cd_h = wave < H ? cD[(size_t)b * H + wave] : 0.f;
  const int ntile = (S + SQ_T - 1) / SQ_T;
  const int t_beg = (int)((long long)ntile * sp / nsplit), t_end = (int)((long long)ntile * (sp + 1) / nsplit);
  f32x4 wacc[SQ_NT];                        // dW_k rows n = wave*16 + g*4 + r, columns kt*16 + li (column E = db_k)
#pragma unroll
  for (int i = 0; i < SQ_NT; ++i) wacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dqa = 0.f;                          // wave = head, lane = channel d < 15
  const int half = E >> 1;
  SqRows rows;
  if (t_beg < t_end) rows = sq_load_rows(X, b, t_beg * SQ_T, S, E, true);     // column E = 1: the bias gradient rides in dW
  __syncthreads();
  SQ_MARK(1);
  int ys_n0 = -1;                            // first row of the dX tile waiting in Ys (-1: none)
  auto flush_dx = [&]() {                    // Ys -> dX rows ys_n0 .. +63: one contiguous block of dX, float4 per thread and pass
#if SQ_DX_LDS
    if (ys_n0 >= 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = t + i * 256;
        const int r = idx >> 4, c = (idx & 15) * 4;
        if (ys_n0 + r < S && c < E) {
          float4* dst = reinterpret_cast<float4*>(dX + ((size_t)b * S + ys_n0 + r) * E + c);
          float4 v = *reinterpret_cast<const float4*>(&Ys[r * SQ_LD + c]);
          if (acc_dx) {                        // the context's gradient summed in place (several consumers, one buffer)
            const float4 o = *dst;
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
          *dst = v;
        }
      }
    }
#endif
  };
  for (int tile = t_beg; tile < t_end; ++tile) {
    const int n0 = tile * SQ_T;
    flush_dx();                              // the previous tile's dX (its closing barrier made Ys complete)
    sq_store_rows(Xs, rows);
    if (tile + 1 < t_end) rows = sq_load_rows(X, b, n0 + SQ_T, S, E, true);
    __syncthreads();
    SQ_MARK(2);
    float rot[SQ_ROT][2];                                    // (cos, sin) of this thread's items, reused by the inverse rotation
    sq_project_rope<true>(T, Xs, Ws, bk, xyz, freq, b, n0, S, E, rot);
    SQ_MARK(3);
    sq_rows_times_heads(T, Qm, sS, E);                       // scores
    sq_rows_times_heads(Xs, Dm, dS, E);                      // dp = dxbar . x_k
    __syncthreads();
    SQ_MARK(4);
    if (wave < H) {
      const bool ok = n0 + lane < S && lse_h != -INFINITY;
      const float p = ok ? __expf(sS[wave * SQ_T + lane] - lse_h) : 0.f;
      sS[wave * SQ_T + lane] = p;
      dS[wave * SQ_T + lane] = p * (dS[wave * SQ_T + lane] - cd_h);
    }
    __syncthreads();
    SQ_MARK(5);
    // rotated-query gradient: dq_h[d] += sum_k ds_k,h k_k[h*15 + d]   (T still holds the rotated keys)
#if SQ_DQ_PAR
    if (wave < H && li < HD) {                  // lane = (key quarter g, channel li): partial sums, reduced over g after the loop
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < SQ_T / 4; ++k) a += dS[wave * SQ_T + g * (SQ_T / 4) + k] * T[(g * (SQ_T / 4) + k) * SQ_LD + wave * HD + li];
      dqa += a;
    }
#else
    if (wave < H && lane < HD) {
      float a = 0.f;
#pragma unroll 8
      for (int k = 0; k < SQ_T; ++k) a += dS[wave * SQ_T + k] * T[k * SQ_LD + wave * HD + lane];
      dqa += a;
    }
#endif
    __syncthreads();
    SQ_MARK(6);
    // T <- R_k^T (ds_k (x) q): gradient w.r.t. the projected (un-rotated) key rows
#pragma unroll
    for (int it = 0; it < SQ_ROT; ++it) {
      const int idx = t + it * 256;
      if (idx >= SQ_T * half) break;
      const int r = idx / half, p = idx - r * half;
      const int c0 = 2 * p, c1 = c0 + 1;
      const int h0 = c0 / HD, h1 = c1 / HD;
      const float g0 = dS[h0 * SQ_T + r] * Qm[h0 * SQ_LD + c0];
      const float g1 = dS[h1 * SQ_T + r] * Qm[h1 * SQ_LD + c1];
      float y0 = g0, y1 = g1;
      if (xyz && n0 + r < S) {
        const float cs = rot[it][0], sn = rot[it][1];
        y0 = cs * g0 + sn * g1;
        y1 = cs * g1 - sn * g0;
      }
      T[r * SQ_LD + c0] = y0;
      T[r * SQ_LD + c1] = y1;
    }
    __syncthreads();
    SQ_MARK(7);
    // dX tile = T W_k (dgrad, contraction over the projection's output channels) + sum_h p_h dxbar_h
    {
      f32x4 acc[SQ_NT];
#pragma unroll
      for (int i = 0; i < SQ_NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {          // T's columns >= E and Ws's rows >= E are zero
        const float a = T[(wave * 16 + li) * SQ_LD + kk * 4 + g];
#pragma unroll
        for (int ct = 0; ct < SQ_NT; ++ct) acc[ct] = mfma_f32_16x16x4(a, Ws[(kk * 4 + g) * SQ_LD + ct * 16 + li], acc[ct]);
      }
#if SQ_DX_LDS
      {
        // + sum_h p_h dxbar_h as one more k-step: A[row][k = head g] = p, B[k = head g][col] = dxbar (rows >= H of both are zero)
        const float pa = sS[g * SQ_T + wave * 16 + li];
#pragma unroll
        for (int ct = 0; ct < SQ_NT; ++ct) acc[ct] = mfma_f32_16x16x4(pa, Dm[g * SQ_LD + ct * 16 + li], acc[ct]);
      }
#pragma unroll
      for (int ct = 0; ct < SQ_NT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ys[(wave * 16 + g * 4 + r) * SQ_LD + ct * 16 + li] = acc[ct][r];
      ys_n0 = n0;                               // stored after the tile's closing barrier (top of the next iteration / after the loop)
#else
#pragma unroll
      for (int ct = 0; ct < SQ_NT; ++ct) {
        const int c = ct * 16 + li;
        if (c >= E) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wave * 16 + g * 4 + r;
          const int n = n0 + row;
          if (n >= S) continue;
          float v = acc[ct][r];
          for (int h = 0; h < H; ++h) v += sS[h * SQ_T + row] * Dm[h * SQ_LD + c];
          if (acc_dx) v += dX[((size_t)b * S + n) * E + c];
          dX[((size_t)b * S + n) * E + c] = v;
        }
      }
#endif
    }
    SQ_MARK(8);
    // dW_k | db_k += T^T [Xs | 1]   (contraction over the tile's keys; wave -> output rows n = wave*16 .. +15)
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) {
      float a[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = T[(mm * 16 + g * 4 + j) * SQ_LD + wave * 16 + li];
#pragma unroll
      for (int kt = 0; kt < SQ_NT; ++kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wacc[kt] = mfma_f32_16x16x4(a[j], Xs[(mm * 16 + g * 4 + j) * SQ_LD + kt * 16 + li], wacc[kt]);
      }
    }
    __syncthreads();
    SQ_MARK(9);
  }
  flush_dx();                                // the last tile
#if SQ_DQ_PAR
  dqa += __shfl_xor(dqa, 16, 64);            // the four key quarters of a channel
  dqa += __shfl_xor(dqa, 32, 64);
#endif
  const int KE = E + 1;
  float* wp = wpart + ((size_t)b * nsplit + sp) * E * KE;
#pragma unroll
  for (int kt = 0; kt < SQ_NT; ++kt) {
    const int k = kt * 16 + li;
    if (k >= KE) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = wave * 16 + g * 4 + r;
      if (n < E) wp[(size_t)n * KE + k] = wacc[kt][r];
    }
  }
  if (wave < H && lane < 16) dqp[(((size_t)sp * B + b) * H + wave) * 16 + lane] = lane < HD ? dqa : 0.f;
  SQ_MARK(10);
  if (prof_on && threadIdx.x == 0) g_sq_prof[11] = t_end - t_beg;
}

// wave-local kernels of single_query_wave.hip (round 5): the default; they return false when switched off (A3D_SQ_WAVE=0)
bool sqw_launch_fwd(const float* X, const float* xyz, const float* Wk, int ldw, const float* bk, const float* qrot, const float* freq,
                    float* part, int B, int S, int E, int H, int nsplit, hipStream_t s);
bool sqw_launch_bwd(const float* X, const float* xyz, const float* Wk, int ldw, const float* bk, const float* qrot, const float* freq,
                    const float* lse, const float* dxbar, const float* cD, float* dX, float* wpart, float* dqp, int B, int S, int E,
                    int H, int nsplit, int acc_dx, hipStream_t s);

}  // namespace a3d

using namespace a3d;

static int sq_check(const char* fn, int B, int S, int E, int H, int nsplit) {
  if (B <= 0 || S <= 0 || E <= 0 || E > 60 || (E % 6) != 0 || (E % 4) != 0 || H <= 0 || H > 4 || H * HD != E || nsplit < 1 ||
      nsplit > 256) {
    set_error("%s: bad shape (B=%d S=%d E=%d H=%d nsplit=%d; E = 15 H <= 60, E %% 12 == 0)", fn, B, S, E, H, nsplit);
    return A3D_ERR_ARG;
  }
  return A3D_OK;
}

extern "C" size_t a3d_sq_fwd_ws_floats(int B, int H, int E, int nsplit) { return (size_t)B * nsplit * H * (E + 2); }

extern "C" int a3d_sq_attn_fwd(const float* X, const float* xyz, const float* Wk, int ldw, const float* bk, const float* Wv,
                               int ldwv, const float* bv, const float* qrot, const float* freq, float* ws, float* xbar,
                               float* lse, float* o, int B, int S, int E, int H, int nsplit, void* stream) {
  int rc = sq_check("a3d_sq_attn_fwd", B, S, E, H, nsplit);
  if (rc) return rc;
  // o == NULL: stop after xbar / lse (the value projection runs in the caller's fused layer kernel, a3d_qs_post_fwd)
  if (!X || !Wk || (o && !Wv) || !qrot || !ws || !xbar || !lse || (xyz && !freq) || ((((uintptr_t)X) & 15) != 0)) {
    set_error("a3d_sq_attn_fwd: null / misaligned pointer");
    return A3D_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)(3 * SQ_T * SQ_LD + 16 * SQ_LD + 4 * SQ_T) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)sq_fwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    (void)hipFuncSetAttribute((const void*)sq_fwd_kernel<60>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    attr_set = true;
  }
  if (sqw_launch_fwd(X, xyz, Wk, ldw, bk, qrot, freq, ws, B, S, E, H, nsplit, s)) {
  } else if (E == 60 && H == 4)
    hipLaunchKernelGGL(sq_fwd_kernel<60>, dim3(nsplit, B), dim3(256), lds, s, X, xyz, Wk, ldw, bk, qrot, freq, ws, B, S, E, H, nsplit);
  else
    hipLaunchKernelGGL(sq_fwd_kernel<0>, dim3(nsplit, B), dim3(256), lds, s, X, xyz, Wk, ldw, bk, qrot, freq, ws, B, S, E, H, nsplit);
  rc = check_launch("a3d_sq_attn_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(sq_combine_kernel, dim3(B * H), dim3(64), 0, s, ws, xbar, lse, B, H, E, nsplit);
  rc = check_launch("a3d_sq_attn_fwd(combine)");
  if (rc || !o) return rc;
  hipLaunchKernelGGL(sq_vproj_kernel, dim3(cdiv(B * E, 256)), dim3(256), 0, s, xbar, Wv, ldwv, bv, o, B, H, E);
  return check_launch("a3d_sq_attn_fwd(vproj)");
}

extern "C" size_t a3d_sq_bwd_ws_floats(int B, int H, int E, int nsplit) {
  return (size_t)B * H * E + (size_t)B * H + (size_t)B * nsplit * E * (E + 1);      // dxbar | cD | weight-gradient partials
}

static int sq_attn_bwd_impl(const float* X, const float* xyz, const float* Wk, int ldw, const float* bk, const float* Wv,
                            int ldwv, const float* qrot, const float* freq, const float* xbar, const float* lse,
                            const float* dO, float* ws, float* dX, float* dqp, float* dWk, int lddwk, float* dbk, float* dWv,
                            int lddwv, float* dbv, int B, int S, int E, int H, int nsplit, int acc_dx, void* stream) {
  int rc = sq_check("a3d_sq_attn_bwd", B, S, E, H, nsplit);
  if (rc) return rc;
  // dO == NULL: ws already holds dxbar | cD (written by a3d_qs_post_bwd, which also owns the value projection's gradients)
  if (!X || !Wk || (dO && (!Wv || !dWv)) || !qrot || !xbar || !lse || !ws || !dX || !dqp || !dWk || !dbk || (xyz && !freq) ||
      ((((uintptr_t)X | (uintptr_t)dX) & 15) != 0)) {
    set_error("a3d_sq_attn_bwd: null / misaligned pointer");
    return A3D_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  float* dxbar = ws;
  float* cD = dxbar + (size_t)B * H * E;
  float* wpart = cD + (size_t)B * H;
  if (dO) {
    hipLaunchKernelGGL(sq_vproj_bwd_kernel, dim3(B * H + E), dim3(64), 0, s, dO, xbar, Wv, ldwv, dxbar, cD, dWv, lddwv, dbv, B, H, E);
    rc = check_launch("a3d_sq_attn_bwd(vproj)");
    if (rc) return rc;
  }
  const size_t lds = (size_t)((3 + SQ_DX_LDS) * SQ_T * SQ_LD + 32 * SQ_LD + 8 * SQ_T) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)sq_bwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    (void)hipFuncSetAttribute((const void*)sq_bwd_kernel<60>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    attr_set = true;
  }
  if (sqw_launch_bwd(X, xyz, Wk, ldw, bk, qrot, freq, lse, dxbar, cD, dX, wpart, dqp, B, S, E, H, nsplit, acc_dx, s)) {
  } else if (E == 60 && H == 4)
    hipLaunchKernelGGL(sq_bwd_kernel<60>, dim3(nsplit, B), dim3(256), lds, s, X, xyz, Wk, ldw, bk, qrot, freq, lse, dxbar, cD, dX,
                       wpart, dqp, B, S, E, H, nsplit, acc_dx);
  else
    hipLaunchKernelGGL(sq_bwd_kernel<0>, dim3(nsplit, B), dim3(256), lds, s, X, xyz, Wk, ldw, bk, qrot, freq, lse, dxbar, cD, dX,
                       wpart, dqp, B, S, E, H, nsplit, acc_dx);
  rc = check_launch("a3d_sq_attn_bwd");
  if (rc) return rc;
  return a3d_sq_wgrad_reduce(wpart, B * nsplit, dWk, lddwk, dbk, E, stream);
}

extern "C" int a3d_sq_attn_bwd(const float* X, const float* xyz, const float* Wk, int ldw, const float* bk, const float* Wv,
                               int ldwv, const float* qrot, const float* freq, const float* xbar, const float* lse,
                               const float* dO, float* ws, float* dX, float* dqp, float* dWk, int lddwk, float* dbk, float* dWv,
                               int lddwv, float* dbv, int B, int S, int E, int H, int nsplit, void* stream) {
  return sq_attn_bwd_impl(X, xyz, Wk, ldw, bk, Wv, ldwv, qrot, freq, xbar, lse, dO, ws, dX, dqp, dWk, lddwk, dbk, dWv, lddwv, dbv, B, S,
                          E, H, nsplit, 0, stream);
}

// the same with dX += (accumulate_dX != 0): the context's gradient summed in place by its consumers
extern "C" int a3d_sq_attn_bwd_acc(const float* X, const float* xyz, const float* Wk, int ldw, const float* bk, const float* Wv,
                                   int ldwv, const float* qrot, const float* freq, const float* xbar, const float* lse,
                                   const float* dO, float* ws, float* dX, float* dqp, float* dWk, int lddwk, float* dbk,
                                   float* dWv, int lddwv, float* dbv, int B, int S, int E, int H, int nsplit, int accumulate_dX,
                                   void* stream) {
  return sq_attn_bwd_impl(X, xyz, Wk, ldw, bk, Wv, ldwv, qrot, freq, xbar, lse, dO, ws, dX, dqp, dWk, lddwk, dbk, dWv, lddwv, dbv, B, S,
                          E, H, nsplit, accumulate_dX ? 1 : 0, stream);
}

// development aid: arm (on != 0) / disarm the phase timestamps of sq_bwd_kernel's workgroup (0, 0) and read the 12 values of
// the last armed launch back (out12 may be NULL when only arming)
extern "C" int a3d_dbg_sq_prof(int on, long long* out12) {
  const int v = on ? 1 : 0;
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_sq_prof_on), &v, sizeof(int));
  if (e == hipSuccess && out12) e = hipMemcpyFromSymbol(out12, HIP_SYMBOL(g_sq_prof), 12 * sizeof(long long));
  if (e != hipSuccess) { set_error("a3d_dbg_sq_prof: %s", hipGetErrorString(e)); return A3D_ERR_LAUNCH; }
  return A3D_OK;
}
