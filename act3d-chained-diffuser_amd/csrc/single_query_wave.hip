// Wave-local key passes of the single-query attention (round 5): the same arithmetic as single_query.hip's sq_fwd_kernel /
// sq_bwd_kernel (Act3D's query stream, act3d.py:467-480 -> RelativeCrossAttentionLayer, layers.py:293-310, on
// MultiheadCustomAttention, multihead_custom_attention.py:246-359), restructured so that a wave owns 16 keys END TO END and the
// tile loop has NO workgroup barrier.
//
// Why: the round-4 kernels walk a 64-key tile in 9 barrier-separated phases (rows -> LDS | projection | RoPE in LDS | scores |
// p, ds | dq | inverse rotation | dX | dW); the phase probe (profiles/sq_bwd_phases.py, MI355X) showed 13.5 us per tile for ~3 us of
// f32 MFMA issue, i.e. a latency chain (MfmaUtil 24 %, VALUBusy 23 %).  Here every product is formed TRANSPOSED,
//     T^T[c][key] = sum_cin W_k[c][cin] X[key][cin]       (A = W_k rows from LDS, B = X rows straight from global memory),
// so that an MFMA lane (li = lane & 15, g = lane >> 4) holds, for ITS key li, the 16 projected channels ct * 16 + g * 4 + r:
//   * both channels of every RoPE pair sit in one lane: the rotation is register arithmetic on the lane's own key's xyz (no LDS
//     round trip, no shuffle, 8 sincos per lane);
//   * the rotated keys are directly the B operand of the score product  s^T[h][key] = sum_c Q[h][c] T[key][c]  (the contraction
//     index is enumerated as (ct, r) <-> channel ct * 16 + g * 4 + r on both operands);
//   * the gradient w.r.t. the projected keys, the inverse rotation and the rotated-query gradient are lane-local;
//   * dX^T[cin][key] = sum_c W_k[c][cin] G[key][c] leaves as one float4 per lane and 16-channel block, no staging tile.
// Only the weight gradient contracts over KEYS and needs G with channels on the lane index: one wave-private LDS round trip
// (16 x 64 floats, wavefront-scope fence, no s_barrier).  Softmax statistics (forward) are per LANE (each lane owns one key
// per step) and merged once after the loop; dW / dq partials are per wave and merged once after the loop.
// Exact-f32 MFMA throughout (v_mfma_f32_16x16x4_f32), as before.  E <= 64, H <= 4, E % 4 == 0.
#include "a3d_common.h"
#include "../../include/act3d_hip.h"
#include <stdlib.h>

namespace a3d {

// workgroups per CU the forward kernel is compiled for: 2 (<= 256 VGPRs, 175 used) or 3 (<= 170: one spill); A/B build flag
#ifndef SQW_FWD_WGS_PER_CU
#define SQW_FWD_WGS_PER_CU 2
#endif
constexpr int SQW_LD = 68;      // row stride (floats) of the LDS matrices W_k, W_k^T, Q, dxbar
constexpr int SQW_XLD = 72;     // row stride of the wave-private key tiles: 72 = 8 (mod 32): the (4 s + g, li) scalar reads hit every
                                // bank exactly twice, and the backward's LDS stays under 80 KB (two workgroups per CU)

__device__ __forceinline__ float sqw_pick(int ax, float x, float y, float z) { return ax == 0 ? x : (ax == 1 ? y : z); }
__device__ __forceinline__ float sqw_pick4(int i, float a, float b, float c, float d) { return i == 0 ? a : (i == 1 ? b : (i == 2 ? c : d)); }
// an integer the compiler must treat as new in every loop iteration: LDS operand addresses derived from it are not loop-invariant,
// so the (invariant) W_k / Q fragments are re-read from LDS per step instead of being hoisted into 64+ VGPRs (spills at 256)
__device__ __forceinline__ int sqw_opaque_zero() {
  int z = 0;
  asm volatile("" : "+v"(z));
  return z;
}
__device__ __forceinline__ void sqw_wave_sync() {
  // orders this wave's LDS writes before its later LDS reads of OTHER lanes' data (one wave executes in lockstep and its LDS
  // operations complete in order; the fences keep the compiler from moving them across)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float sqw_row_sum(float v) {      // sum over the 16 lanes that share g (li = 0..15)
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// W (E x E, row stride ldw, 4-byte aligned) -> Ws[64][SQW_LD] (and its transpose WsT when non-null), zero padded
__device__ __forceinline__ void sqw_stage_weight(float* Ws, float* WsT, const float* __restrict__ W, int ldw, int E) {
  for (int idx = threadIdx.x; idx < 64 * 64; idx += blockDim.x) {
    const int j = idx >> 6, c = idx & 63;
    const float v = (j < E && c < E) ? W[(size_t)j * ldw + c] : 0.f;
    Ws[j * SQW_LD + c] = v;
    if (WsT) WsT[c * SQW_LD + j] = v;
  }
}

struct SqwKey { float4 x[4]; float px, py, pz; };
// the lane's key row in B-operand order: x[ct] = X[n][ct * 16 + g * 4 .. + 3] (zero beyond S / E; channel E := 1 when `ones`)
__device__ __forceinline__ SqwKey sqw_load_key(const float* __restrict__ X, const float* __restrict__ xyz, int b, int n, int S, int E,
                                               int g, bool ones) {
  SqwKey k;
  const bool valid = n < S;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int c0 = ct * 16 + g * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid && c0 < E) v = *reinterpret_cast<const float4*>(X + ((size_t)b * S + n) * E + c0);      // E % 4 == 0
    if (ones && valid && c0 == E) v.x = 1.f;
    k.x[ct] = v;
  }
  k.px = k.py = k.pz = 0.f;
  if (xyz && valid) {
    const float* p = xyz + ((size_t)b * S + n) * 3;
    k.px = p[0]; k.py = p[1]; k.pz = p[2];
  }
  return k;
}

// acc[ct][r] = rope(W_k x + b_k)[ct * 16 + g * 4 + r] of the lane's key; cs / sn: the 8 pairs' rotation (kept for the backward)
__device__ __forceinline__ void sqw_project_rope(const float* Ws, const float* Bs, const SqwKey& k, const float (&fq)[4][2],
                                                 unsigned int ax, bool rotate, int li, int g, f32x4 (&acc)[4], float (&cs)[4][2],
                                                 float (&sn)[4][2]) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const float4 bv = *reinterpret_cast<const float4*>(&Bs[ct * 16 + g * 4]);
    acc[ct] = f32x4{bv.x, bv.y, bv.z, bv.w};
  }
  // W_k fragments double-buffered by hand: the next 16-channel block's four float4 are in flight while the current block's 16
  // MFMAs issue; the scheduling barriers keep the compiler from hoisting ALL sixteen float4 (64 VGPRs) in front of the MFMAs
  float4 a[4], an[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) a[ct] = *reinterpret_cast<const float4*>(&Ws[(ct * 16 + li) * SQW_LD + g * 4]);
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    if (jt < 3) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) an[ct] = *reinterpret_cast<const float4*>(&Ws[(ct * 16 + li) * SQW_LD + (jt + 1) * 16 + g * 4]);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = mfma_f32_16x16x4(a[ct].x, k.x[jt].x, acc[ct]);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = mfma_f32_16x16x4(a[ct].y, k.x[jt].y, acc[ct]);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = mfma_f32_16x16x4(a[ct].z, k.x[jt].z, acc[ct]);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = mfma_f32_16x16x4(a[ct].w, k.x[jt].w, acc[ct]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) a[ct] = an[ct];
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      cs[ct][j] = 1.f;
      sn[ct][j] = 0.f;
    }
  if (rotate) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        fast_sincos(sqw_pick((int)((ax >> (2 * (ct * 2 + j))) & 3u), k.px, k.py, k.pz) * fq[ct][j], &sn[ct][j], &cs[ct][j]);
        const float y0 = acc[ct][2 * j], y1 = acc[ct][2 * j + 1];
        acc[ct][2 * j] = y0 * cs[ct][j] - y1 * sn[ct][j];
        acc[ct][2 * j + 1] = y1 * cs[ct][j] + y0 * sn[ct][j];
      }
  }
}

// out[r] (lanes with g == 0): sum_c M[r][c] V[key li][c] for the 4 rows r of M ([16][SQW_LD], rows >= 4 unused by the caller);
// V given per lane as v[ct][e] = V[key][ct * 16 + g * 4 + e]
__device__ __forceinline__ f32x4 sqw_heads_dot(const float* M, int li, int g, const f32x4 (&v)[4]) {
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const float4 a = *reinterpret_cast<const float4*>(&M[li * SQW_LD + ct * 16 + g * 4]);
    s0 = mfma_f32_16x16x4(a.x, v[ct][0], s0);
    s1 = mfma_f32_16x16x4(a.y, v[ct][1], s1);
    s0 = mfma_f32_16x16x4(a.z, v[ct][2], s0);
    s1 = mfma_f32_16x16x4(a.w, v[ct][3], s1);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) s0[r] += s1[r];
  return s0;
}

// the lane's RoPE constants: pair (ct, j) = channels c, c + 1 with c = ct * 16 + g * 4 + 2 j -> axis c / (E / 3), frequency index
// (the eight axis indices packed two bits each into one register)
__device__ __forceinline__ unsigned int sqw_rope_consts(const float* __restrict__ freq, int E, int g, bool rotate, float (&fq)[4][2]) {
  const int third = E / 3;
  unsigned int ax = 0;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = ct * 16 + g * 4 + 2 * j;
      fq[ct][j] = 0.f;
      if (rotate && c < E) {
        const int axis = c / third;
        ax |= (unsigned int)axis << (2 * (ct * 2 + j));
        fq[ct][j] = freq[(c - axis * third) >> 1];
      }
    }
  return ax;
}

// ------------------------------------------------------------------------------------------------ forward
// grid (nsplit, B); partial [B][nsplit][H][E + 2] = {m, l, xbar[E]} per head (as sq_fwd_kernel: sq_combine_kernel reads it)
template <int EC>
__global__ __launch_bounds__(256, SQW_FWD_WGS_PER_CU) void sqw_fwd_kernel(const float* __restrict__ X, const float* __restrict__ xyz,
                                                         const float* __restrict__ Wk, int ldw, const float* __restrict__ bk,
                                                         const float* __restrict__ qrot, const float* __restrict__ freq,
                                                         float* __restrict__ part, int B, int S, int E_rt, int H_rt, int nsplit) {
  const int E = EC > 0 ? EC : E_rt, H = EC > 0 ? EC / HD : H_rt;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                         // [64][SQW_LD]
  float* Qm = Ws + 64 * SQW_LD;             // [16][SQW_LD]: row h = the head's rotated query in its channel range, else 0
  float* Bs = Qm + 16 * SQW_LD;             // [64] bias
  float* Mx = Bs + 64;                      // [4 waves][4 heads] maxima
  float* Red = Mx + 16;                     // [4 waves][4 heads][SQW_LD]: xbar partial at [0, 64), l at [64]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int b = blockIdx.y, sp = blockIdx.x;
  const bool rotate = xyz != nullptr;
  sqw_stage_weight(Ws, nullptr, Wk, ldw, E);
  for (int idx = t; idx < 16 * SQW_LD; idx += 256) {
    const int h = idx / SQW_LD, c = idx - h * SQW_LD;
    const int d = c - h * HD;
    Qm[idx] = (h < H && c < E && d >= 0 && d < HD) ? qrot[((size_t)b * H + h) * 16 + d] : 0.f;
  }
  if (t < 64) Bs[t] = (t < E && bk) ? bk[t] : 0.f;
  float fq[4][2];
  const unsigned int ax = sqw_rope_consts(freq, E, g, rotate, fq);
  const int ntile = (S + 63) >> 6;
  const int t_beg = (int)((long long)ntile * sp / nsplit), t_end = (int)((long long)ntile * (sp + 1) / nsplit);
  float m[4], l[4], xa[4][16];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    m[h] = -INFINITY;
    l[h] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) xa[h][i] = 0.f;
  }
  SqwKey cur;
  if (t_beg < t_end) cur = sqw_load_key(X, xyz, b, t_beg * 64 + wave * 16 + li, S, E, g, false);
  __syncthreads();
  for (int tile = t_beg; tile < t_end; ++tile) {
    const int n = tile * 64 + wave * 16 + li;
    SqwKey nxt = cur;
    if (tile + 1 < t_end) nxt = sqw_load_key(X, xyz, b, n + 64, S, E, g, false);      // one step ahead: hides the HBM round trip
    f32x4 acc[4];
    float cs[4][2], sn[4][2];
    const int oz = sqw_opaque_zero();
    sqw_project_rope(Ws + oz, Bs + oz, cur, fq, ax, rotate, li, g, acc, cs, sn);
    const f32x4 sc = sqw_heads_dot(Qm + oz, li, g, acc);
    const bool valid = n < S;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      if (h >= H) break;
      const float sv = __shfl(sc[h], li, 64);                   // lane li (g == 0) holds the four heads' scores of key li
      const float s = valid ? sv : -INFINITY;
      const float m_new = fmaxf(m[h], s);
      const float p = (s == -INFINITY) ? 0.f : __expf(s - m_new);
      const float alpha = (m[h] == -INFINITY) ? 0.f : __expf(m[h] - m_new);
      l[h] = l[h] * alpha + p;
      m[h] = m_new;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        xa[h][ct * 4 + 0] = fmaf(xa[h][ct * 4 + 0], alpha, p * cur.x[ct].x);
        xa[h][ct * 4 + 1] = fmaf(xa[h][ct * 4 + 1], alpha, p * cur.x[ct].y);
        xa[h][ct * 4 + 2] = fmaf(xa[h][ct * 4 + 2], alpha, p * cur.x[ct].z);
        xa[h][ct * 4 + 3] = fmaf(xa[h][ct * 4 + 3], alpha, p * cur.x[ct].w);
      }
    }
    cur = nxt;
  }
  // ---- merge the per-lane softmax states: workgroup maximum per head, rescale, sum over the key lanes and the waves
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    float mw = m[h];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) mw = fmaxf(mw, __shfl_xor(mw, o, 64));
    if (lane == 0) Mx[wave * 4 + h] = mw;
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    if (h >= H) break;
    const float mall = fmaxf(fmaxf(Mx[h], Mx[4 + h]), fmaxf(Mx[8 + h], Mx[12 + h]));
    const float f = (m[h] == -INFINITY) ? 0.f : __expf(m[h] - mall);
    const float ls = sqw_row_sum(l[h] * f);
    if (lane == 0) Red[(wave * 4 + h) * SQW_LD + 64] = ls;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float v = sqw_row_sum(xa[h][i] * f);
      if (li == 0) Red[(wave * 4 + h) * SQW_LD + (i >> 2) * 16 + g * 4 + (i & 3)] = v;
    }
  }
  __syncthreads();
  for (int idx = t; idx < H * (E + 2); idx += 256) {
    const int h = idx / (E + 2), c = idx - h * (E + 2);
    float* o = part + (((size_t)b * nsplit + sp) * H + h) * (E + 2);
    if (c == 0) {
      o[0] = fmaxf(fmaxf(Mx[h], Mx[4 + h]), fmaxf(Mx[8 + h], Mx[12 + h]));
    } else {
      const int col = c == 1 ? 64 : c - 2;
      o[c] = (Red[(0 * 4 + h) * SQW_LD + col] + Red[(1 * 4 + h) * SQW_LD + col]) + (Red[(2 * 4 + h) * SQW_LD + col] + Red[(3 * 4 + h) * SQW_LD + col]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
// grid (nsplit, B).  dX [B][S][E] (every row written once, or += when acc_dx); wpart [B * nsplit][E][E + 1] (dW_k | db_k);
// dqp [nsplit][B][H][1][16] -- the layouts of sq_bwd_kernel.
template <int EC>
__global__ __launch_bounds__(256, 2) void sqw_bwd_kernel(const float* __restrict__ X, const float* __restrict__ xyz,
                                                         const float* __restrict__ Wk, int ldw, const float* __restrict__ bk,
                                                         const float* __restrict__ qrot, const float* __restrict__ freq,
                                                         const float* __restrict__ lse, const float* __restrict__ dxbar,
                                                         const float* __restrict__ cD, float* __restrict__ dX,
                                                         float* __restrict__ wpart, float* __restrict__ dqp, int B, int S, int E_rt,
                                                         int H_rt, int nsplit, int acc_dx) {
  const int E = EC > 0 ? EC : E_rt, H = EC > 0 ? EC / HD : H_rt;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                         // [64][SQW_LD]   W_k[c][cin]
  float* WsT = Ws + 64 * SQW_LD;            // [64][SQW_LD]   W_k^T[cin][c]
  float* Qm = WsT + 64 * SQW_LD;            // [16][SQW_LD]   row h: the head's rotated query in its channel range
  float* Dm = Qm + 16 * SQW_LD;             // [16][SQW_LD]   row h: dxbar[b][h]
  float* Bs = Dm + 16 * SQW_LD;             // [64] bias | [64] q by channel (column sums of Qm)
  float* Qc = Bs + 64;
  float* Cst = Qc + 64;                     // [8]: lse[b][h] | cD[b][h]
  float* Xw = Cst + 16 + 128;               // (Cst[16 ..): [4 waves][4 g][8] pair frequencies)  [4 waves][16][SQW_XLD]  the wave's key rows (+ ones column at E)
  float* Gw = Xw + 4 * 16 * SQW_XLD;        // [4 waves][16][SQW_XLD]  gradient w.r.t. the projected keys
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int b = blockIdx.y, sp = blockIdx.x;
  const bool rotate = xyz != nullptr;
  sqw_stage_weight(Ws, WsT, Wk, ldw, E);
  for (int idx = t; idx < 16 * SQW_LD; idx += 256) {
    const int h = idx / SQW_LD, c = idx - h * SQW_LD;
    const int d = c - h * HD;
    Qm[idx] = (h < H && c < E && d >= 0 && d < HD) ? qrot[((size_t)b * H + h) * 16 + d] : 0.f;
    Dm[idx] = (h < H && c < E) ? dxbar[((size_t)b * H + h) * E + c] : 0.f;
  }
  if (t < 64) {
    Bs[t] = (t < E && bk) ? bk[t] : 0.f;
    const int h = t / HD;
    Qc[t] = (t < E && h < H) ? qrot[((size_t)b * H + h) * 16 + (t - h * HD)] : 0.f;
  }
  unsigned int ax;
  {
    float fq0[4][2];
    ax = sqw_rope_consts(freq, E, g, rotate, fq0);
    if (li == 0) {                           // the lane group's eight pair frequencies -> LDS (re-read per step, as the per-head constants)
#pragma unroll
      for (int i = 0; i < 8; ++i) Cst[16 + (wave * 4 + g) * 8 + i] = fq0[i >> 1][i & 1];
    }
  }
  if (t < 4) {                               // per-head constants live in LDS (re-read per step: 8 VGPRs that the 256-register budget lacks)
    Cst[t] = t < H ? lse[(size_t)b * H + t] : -INFINITY;
    Cst[4 + t] = t < H ? cD[(size_t)b * H + t] : 0.f;
  }
  unsigned int hd = 0;                       // head of channel ct * 16 + g * 4 + r, two bits each (3 = also the pad channels: their q is zero)
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) hd |= (unsigned int)min((ct * 16 + g * 4 + r) / HD, 3) << (2 * (ct * 4 + r));
  const int ntile = (S + 63) >> 6;
  const int t_beg = (int)((long long)ntile * sp / nsplit), t_end = (int)((long long)ntile * (sp + 1) / nsplit);
  f32x4 wacc[4][4];                          // dW_k tile (ct, kt): rows c = ct * 16 + g * 4 + r, column cin = kt * 16 + li (cin = E: db_k)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) wacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dqa[4][4];                           // rotated-query gradient by channel, this lane's keys
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) dqa[i][j] = 0.f;
  float* xw = Xw + wave * 16 * SQW_XLD;
  float* gw = Gw + wave * 16 * SQW_XLD;
  SqwKey cur;
  if (t_beg < t_end) cur = sqw_load_key(X, xyz, b, t_beg * 64 + wave * 16 + li, S, E, g, true);
  __syncthreads();
  for (int tile = t_beg; tile < t_end; ++tile) {
    const int n = tile * 64 + wave * 16 + li;
    const bool valid = n < S;
    SqwKey nxt = cur;
    if (tile + 1 < t_end) nxt = sqw_load_key(X, xyz, b, n + 64, S, E, g, true);
    // the key rows in the wave's LDS tile for the weight gradient's B operand (lane = input channel there)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<float4*>(&xw[li * SQW_XLD + ct * 16 + g * 4]) = cur.x[ct];
    f32x4 acc[4];
    float cs[4][2], sn[4][2];
    const int oz = sqw_opaque_zero();
    const float4 f0 = *reinterpret_cast<const float4*>(&Cst[oz + 16 + (wave * 4 + g) * 8]), f1 = *reinterpret_cast<const float4*>(&Cst[oz + 16 + (wave * 4 + g) * 8 + 4]);
    const float fq[4][2] = {{f0.x, f0.y}, {f0.z, f0.w}, {f1.x, f1.y}, {f1.z, f1.w}};
    sqw_project_rope(Ws + oz, Bs + oz, cur, fq, ax, rotate, li, g, acc, cs, sn);
    const f32x4 sc = sqw_heads_dot(Qm + oz, li, g, acc);                   // scores       (g == 0 lanes: head r, key li)
    f32x4 xv[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) xv[ct] = f32x4{cur.x[ct].x, cur.x[ct].y, cur.x[ct].z, cur.x[ct].w};
    const f32x4 dpv = sqw_heads_dot(Dm + oz, li, g, xv);                        // dp = dxbar . x_k (Dm's column E is zero: the ones column drops out)
    float p[4], ds[4];
    const float4 lse4 = *reinterpret_cast<const float4*>(&Cst[oz]), cd4 = *reinterpret_cast<const float4*>(&Cst[oz + 4]);
    const float lse_h[4] = {lse4.x, lse4.y, lse4.z, lse4.w}, cd_h[4] = {cd4.x, cd4.y, cd4.z, cd4.w};
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float sv = __shfl(sc[h], li, 64), dv = __shfl(dpv[h], li, 64);
      const bool ok = valid && h < H && lse_h[h] != -INFINITY;
      p[h] = ok ? __expf(sv - lse_h[h]) : 0.f;
      ds[h] = p[h] * (dv - cd_h[h]);
    }
    // rotated-query gradient (lane-local partial), gradient w.r.t. the rotated keys, inverse rotation -> acc = G[key][c]
    const float* Qco = Qc + oz;
    const float4 qc[4] = {*reinterpret_cast<const float4*>(&Qco[0 * 16 + g * 4]), *reinterpret_cast<const float4*>(&Qco[1 * 16 + g * 4]),
                          *reinterpret_cast<const float4*>(&Qco[2 * 16 + g * 4]), *reinterpret_cast<const float4*>(&Qco[3 * 16 + g * 4])};
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const float qv[4] = {qc[ct].x, qc[ct].y, qc[ct].z, qc[ct].w};
      float gr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dsel = sqw_pick4((int)((hd >> (2 * (ct * 4 + r))) & 3u), ds[0], ds[1], ds[2], ds[3]);
        dqa[ct][r] = fmaf(dsel, acc[ct][r], dqa[ct][r]);
        gr[r] = dsel * qv[r];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float g0 = gr[2 * j], g1 = gr[2 * j + 1];
        acc[ct][2 * j] = cs[ct][j] * g0 + sn[ct][j] * g1;
        acc[ct][2 * j + 1] = cs[ct][j] * g1 - sn[ct][j] * g0;
      }
      *reinterpret_cast<float4*>(&gw[li * SQW_XLD + ct * 16 + g * 4]) = make_float4(acc[ct][0], acc[ct][1], acc[ct][2], acc[ct][3]);
    }
    // dX^T[cin][key] = sum_c W_k[c][cin] G[key][c] + sum_h dxbar[h][cin] p[h][key]
    {
      f32x4 dxa[4];
      const float pg = sqw_pick4(g, p[0], p[1], p[2], p[3]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) dxa[mt] = mfma_f32_16x16x4(Dm[oz + g * SQW_LD + mt * 16 + li], pg, f32x4{0.f, 0.f, 0.f, 0.f});
      float4 a[4], an[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) a[mt] = *reinterpret_cast<const float4*>(&WsT[oz + (mt * 16 + li) * SQW_LD + g * 4]);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        if (ct < 3) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) an[mt] = *reinterpret_cast<const float4*>(&WsT[oz + (mt * 16 + li) * SQW_LD + (ct + 1) * 16 + g * 4]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) dxa[mt] = mfma_f32_16x16x4(a[mt].x, acc[ct][0], dxa[mt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) dxa[mt] = mfma_f32_16x16x4(a[mt].y, acc[ct][1], dxa[mt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) dxa[mt] = mfma_f32_16x16x4(a[mt].z, acc[ct][2], dxa[mt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) dxa[mt] = mfma_f32_16x16x4(a[mt].w, acc[ct][3], dxa[mt]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a[mt] = an[mt];
      }
      if (valid) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int c0 = mt * 16 + g * 4;
          if (c0 < E) {
            float4* dst = reinterpret_cast<float4*>(dX + ((size_t)b * S + n) * E + c0);
            float4 v = make_float4(dxa[mt][0], dxa[mt][1], dxa[mt][2], dxa[mt][3]);
            if (acc_dx) {                      // the context's gradient summed in place (several consumers, one buffer)
              const float4 o = *dst;
              v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *dst = v;
          }
        }
      }
    }
    // dW_k | db_k += G^T [X | 1] over the wave's 16 keys: the one product that contracts over keys -> operands from the wave's tiles
    sqw_wave_sync();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float ga[4], xb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ga[q] = gw[(4 * s + g) * SQW_XLD + q * 16 + li];
        xb[q] = xw[(4 * s + g) * SQW_XLD + q * 16 + li];
      }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) wacc[ct][kt] = mfma_f32_16x16x4(ga[ct], xb[kt], wacc[ct][kt]);
    }
    sqw_wave_sync();                           // the next step's tile writes come after these reads
    cur = nxt;
  }
  // ---- merge: dW_k partials of the four waves (fixed order), rotated-query gradient over key lanes and waves
  __syncthreads();
  float* Wred = Xw;                            // [64][SQW_LD] over the key tiles (no longer needed); Dq [4][64] behind it
  float* Dq = Wred + 64 * SQW_LD;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* d = &Wred[(ct * 16 + g * 4 + r) * SQW_LD + kt * 16 + li];
            *d = (w == 0) ? wacc[ct][kt][r] : *d + wacc[ct][kt][r];
          }
    }
    __syncthreads();
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = sqw_row_sum(dqa[ct][r]);
      if (li == 0) Dq[wave * 64 + ct * 16 + g * 4 + r] = v;
    }
  __syncthreads();
  const int KE = E + 1;
  float* wp = wpart + ((size_t)b * nsplit + sp) * E * KE;
  for (int idx = t; idx < E * KE; idx += 256) {
    const int c = idx / KE, k = idx - c * KE;
    wp[idx] = Wred[c * SQW_LD + k];
  }
  if (t < H * 16) {
    const int h = t >> 4, d = t & 15;
    const int c = h * HD + d;
    dqp[(((size_t)sp * B + b) * H + h) * 16 + d] = d < HD ? (Dq[c] + Dq[64 + c]) + (Dq[128 + c] + Dq[192 + c]) : 0.f;
  }
}

// launchers used by single_query.hip's entry points
void sqw_launch_fwd(const float* X, const float* xyz, const float* Wk, int ldw, const float* bk, const float* qrot, const float* freq,
                    float* part, int B, int S, int E, int H, int nsplit, hipStream_t s) {
  const size_t lds = (size_t)(64 * SQW_LD + 16 * SQW_LD + 64 + 16 + 16 * SQW_LD) * sizeof(float);
  if (E == 60 && H == 4)
    hipLaunchKernelGGL(sqw_fwd_kernel<60>, dim3(nsplit, B), dim3(256), lds, s, X, xyz, Wk, ldw, bk, qrot, freq, part, B, S, E, H, nsplit);
  else
    hipLaunchKernelGGL(sqw_fwd_kernel<0>, dim3(nsplit, B), dim3(256), lds, s, X, xyz, Wk, ldw, bk, qrot, freq, part, B, S, E, H, nsplit);
}

void sqw_launch_bwd(const float* X, const float* xyz, const float* Wk, int ldw, const float* bk, const float* qrot, const float* freq,
                    const float* lse, const float* dxbar, const float* cD, float* dX, float* wpart, float* dqp, int B, int S, int E,
                    int H, int nsplit, int acc_dx, hipStream_t s) {
  const size_t lds = (size_t)(2 * 64 * SQW_LD + 2 * 16 * SQW_LD + 128 + 16 + 128 + 2 * 4 * 16 * SQW_XLD) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)sqw_bwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    (void)hipFuncSetAttribute((const void*)sqw_bwd_kernel<60>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    attr_set = true;
  }
  if (E == 60 && H == 4)
    hipLaunchKernelGGL(sqw_bwd_kernel<60>, dim3(nsplit, B), dim3(256), lds, s, X, xyz, Wk, ldw, bk, qrot, freq, lse, dxbar, cD, dX, wpart,
                       dqp, B, S, E, H, nsplit, acc_dx);
  else
    hipLaunchKernelGGL(sqw_bwd_kernel<0>, dim3(nsplit, B), dim3(256), lds, s, X, xyz, Wk, ldw, bk, qrot, freq, lse, dxbar, cD, dX, wpart,
                       dqp, B, S, E, H, nsplit, acc_dx);
}

}  // namespace a3d
