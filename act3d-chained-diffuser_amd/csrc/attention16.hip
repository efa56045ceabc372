// Split-fp16 attention for gfx950: the ghost-point <-> scene cross-attention (and the diffusion transformer's attention
// core) on v_mfma_f32_16x16x32_f16 with HALF the matrix work and a third of the vector work of the split-bf16 kernels in
// attention.hip / attention_bwd.hip, at the same parity class.
//
// Reference semantics (multihead_custom_attention.py:355-447): per head h (d = 15), A = softmax(q_h k_h^T + mask),
// o_h = A v_h.  What bounds this op on MI355X is NOT the matrix pipe: with d = 15 a score costs 60 algorithmic FLOPs
// but one v_exp_f32 (quarter rate: 6.5 cycles per wave64 op, profiles/r03_inst_rate.txt) plus its share of
// conversions, so the inner loops are written to minimise VECTOR instructions per score:
//   * q, k are TWO-part fp16 (x = hi + lo, 22 mantissa bits; fp16 subnormals are honoured by the MFMA, same file):
//     logits are fp32-grade from two K = 32 MFMAs per 16x16 tile, [k_hi|k_lo].[q_hi|q_hi] + [k_hi|k_lo].[q_lo|q_lo]
//     (three with three-part bf16).  exp() turns an ABSOLUTE logit error into a relative weight error, which is why the
//     logit operands keep two parts while everything downstream of the softmax is single fp16:
//   * P (and in the backward dS) and dO are single fp16: their rounding (2^-12 relative) is not amplified; the softmax
//     denominator is accumulated on the MFMA from the SAME rounded P (ones-channel of V), V keeps two parts (a rounded V
//     would make D = dO . O inconsistent with dP = dO . V, and dP - D is a difference of nearly equal numbers when the
//     softmax is sharp), and the backward differentiates exactly the function of the rounded dO (D from the rounded dO).
//   * log2(e) is folded into q by the projection kernel, -m (forward) / -lse (backward) / -D ride in as MFMA accumulator
//     inits: exp2 is applied DIRECTLY to MFMA results -- no per-score argument arithmetic at all.
//   * lazy rescaling: the running max is only revised when a score exceeds it by 2^8 (a wave-uniform, rarely taken
//     branch), so the common path has no cross-lane traffic and no accumulator rescale.
//   * dO rows are normalised by a power of two per (b, h, q) row (exact), so that dS fits fp16 whatever the loss scale.
// Per 64 keys x 16 queries a wave issues 10 (fwd) / 14 (dQ) / 16 (dK,dV) MFMAs against 18 / 26 / 32 before.
//
// Operand formats ("16" formats, written by a3d_proj_rope_split16 / attn16_bwd_prep):
//   rows16   [B][H][Npad][32] fp16 : hi(16) | lo(16) of the 16-padded head row        (q, k, v)
//   planes16 [B][H][parts][16][Npad] fp16 : hi (and lo) planes, transposed (8 consecutive rows of one channel = one MFMA A
//            fragment); v always has both parts, q / k carry `plane_parts` (backward only)
//   dO rows  [B][H][Lqp][16]  fp16, dO planes [B][H][16][Lqp] fp16 (both of the row-normalised dO * ln 2)
// Scores live in log2 units (q carries log2 e): LSE2 = log2 sum_k 2^s2.
#include "a3d_common.h"
#include "../../include/act3d_hip.h"
#include <stdlib.h>

namespace a3d {

typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2_;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;

constexpr int C16 = 64;              // keys (fwd, dQ) or queries (dK/dV) per staged chunk
constexpr float P_OFF = 4.0f;        // p = 2^(s - m + P_OFF): keeps the small weights of a row out of fp16's subnormals
constexpr float P_THR = 8.0f;        // lazy rescale: revise the running max when a score exceeds it by 2^P_THR
constexpr float LN2_F = 0.6931471805599453f;

__device__ __forceinline__ f32x4 mfma_f16(s16x8 a, s16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}
// two floats -> packed fp16 (round to nearest even; v_cvt_pk_f16_f32)
__device__ __forceinline__ unsigned int pk_f16(float a, float b) {
  return __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2_){a, b}, h16x2));
}
__device__ __forceinline__ float max16(const f32x4& a, const f32x4& b, const f32x4& c, const f32x4& d) {
  const float m0 = fmaxf(fmaxf(a[0], a[1]), a[2]);
  const float m1 = fmaxf(fmaxf(a[3], b[0]), b[1]);
  const float m2 = fmaxf(fmaxf(b[2], b[3]), c[0]);
  const float m3 = fmaxf(fmaxf(c[1], c[2]), c[3]);
  const float m4 = fmaxf(fmaxf(d[0], d[1]), d[2]);
  return fmaxf(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)), fmaxf(m4, d[3]));
}

// ------------------------------------------------------------------------------------------------ forward
struct Fwd16Stage {
  s16x8 k, v;
  float bias;
};

// QT 16-query tiles per wave (128 queries per workgroup at QT = 2); 64-key chunks of the K rows and the V plane are
// staged through LDS once per workgroup (double buffered, one barrier per chunk).  Scores are computed transposed
// (S^T = K Q^T) with the key rows of the two 16x16 tiles of a 32-key half interleaved (tile T row i <-> key
// (i >> 2) * 8 + (i & 3) + 4 T): after exp2 a lane holds, in order, the 8 consecutive keys the P operand of the PV MFMA
// wants, so P never touches LDS.
template <bool DROP, int QT>
__global__ __launch_bounds__(256, 2) void attn16_fwd_kernel(
    const unsigned short* __restrict__ Qr, const unsigned short* __restrict__ Kr, const unsigned short* __restrict__ Vp,
    const unsigned char* __restrict__ kmask, float* __restrict__ O, float* __restrict__ LSE2, float* __restrict__ Op,
    float* __restrict__ Mp, float* __restrict__ Lp, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit,
    const unsigned long long* __restrict__ drop_state, unsigned int drop_site, unsigned int drop_thr, float drop_scale) {
  __shared__ __attribute__((aligned(16))) unsigned short Ksm[2][C16 * 32];      // [k_hi | k_lo] rows tile
  __shared__ __attribute__((aligned(16))) unsigned short Vsm[2][4 * 16 * 32];   // [plane hi/lo][32-key half][16 ch][32 keys]
  __shared__ __attribute__((aligned(16))) float biasS[2][C16];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  constexpr int QW = 64 * QT;
  const int tiles_x = (Lqp + QW - 1) / QW;
  int group, within;
  if (!xcd_decode(tiles_x * nsplit, B * H, group, within)) return;
  const int b = group / H, h = group - b * H;
  const int sp = within / tiles_x;
  const int E = H * HD;
  const int qbase = (within - sp * tiles_x) * QW + wave * (16 * QT);
  const size_t bh = (size_t)b * H + h;

  s16x8 qhi[QT], qlo[QT];
  bool active[QT];
  bool any_active = false;
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const int q0 = qbase + u * 16;
    active[u] = q0 < Lqp;
    any_active = any_active || active[u];
    qhi[u] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    qlo[u] = qhi[u];
    if (active[u]) {
      const unsigned short* qp = Qr + (bh * Lqp + q0 + li) * 32;
      qhi[u] = *reinterpret_cast<const s16x8*>(qp + (g & 1) * 8);
      qlo[u] = *reinterpret_cast<const s16x8*>(qp + 16 + (g & 1) * 8);
    }
  }
  const int nch = Sp / C16;
  const int cps = (nch + nsplit - 1) / nsplit;
  const int c_beg = sp * cps;
  const int c_end = min(nch, c_beg + cps);

  const int krow = t >> 2, kseg = t & 3;
  const int vplane = t >> 7, vd = (t >> 3) & 15, vseg = t & 7;
  auto stage_load = [&](int c) {
    Fwd16Stage st;
    st.k = *reinterpret_cast<const s16x8*>(Kr + (bh * Sp + (size_t)c * C16 + krow) * 32 + kseg * 8);
    st.v = *reinterpret_cast<const s16x8*>(Vp + ((bh * 2 + vplane) * 16 + vd) * Sp + (size_t)c * C16 + vseg * 8);
    st.bias = 0.f;
    if (t < C16) {
      const int key = c * C16 + t;
      bool valid = key < S;
      if (valid && kmask) valid = kmask[(size_t)b * S + key] == 0;
      st.bias = valid ? 0.f : -INFINITY;
    }
    return st;
  };
  auto stage_store = [&](const Fwd16Stage& st, int buf) {
    *reinterpret_cast<s16x8*>(&Ksm[buf][tile_off(krow, kseg)]) = st.k;
    // padded channel 15 of V := 1.0: acc[d = 15] accumulates the softmax denominator on the MFMA pipe from exactly the
    // rounded P the numerator uses
    const s16x8 ones = {0x3C00, 0x3C00, 0x3C00, 0x3C00, 0x3C00, 0x3C00, 0x3C00, 0x3C00};
    *reinterpret_cast<s16x8*>(&Vsm[buf][((vplane * 2 + (vseg >> 2)) * 16) * 32 + plane_off(vd, vseg & 3)]) =
        (vplane == 0 && vd == 15) ? ones : st.v;
    if (t < C16) biasS[buf][t] = st.bias;
  };

  int koff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) koff[j] = tile_off((j >> 1) * 32 + (li >> 2) * 8 + (li & 3) + (j & 1) * 4, g);
  const int voff = plane_off(li, g);

  float m_run[QT], l_run[QT];            // running max (log2 units, exact per query column); l_run: DROP only
  f32x4 cin[QT];                         // MFMA accumulator init of the score tiles: P_OFF - m_run
  f32x4 acc0[QT], acc1[QT];
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    m_run[u] = 0.f;
    l_run[u] = 0.f;
    cin[u] = f32x4{P_OFF, P_OFF, P_OFF, P_OFF};
    acc0[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1[u] = acc0[u];
  }
  DropKey dkey = {0u, 0u};
  if (DROP) dkey = drop_key(drop_state);

  if (c_beg < c_end) {
    stage_store(stage_load(c_beg), 0);
    __syncthreads();
  }
  for (int c = c_beg; c < c_end; ++c) {
    const int buf = (c - c_beg) & 1;
    Fwd16Stage nxt;
    const bool has_next = (c + 1 < c_end);
    if (has_next) nxt = stage_load(c + 1);
    const bool first = (c == c_beg);
    const bool masked = (kmask != nullptr) || ((c + 1) * C16 > S);     // wave-uniform: the chunk may hold invalid keys

    if (any_active) {
      s16x8 kf[4], vh[2], vl[2];
#pragma unroll
      for (int j = 0; j < 4; ++j) kf[j] = *reinterpret_cast<const s16x8*>(&Ksm[buf][koff[j]]);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        vh[hf] = *reinterpret_cast<const s16x8*>(&Vsm[buf][((0 * 2 + hf) * 16) * 32 + voff]);
        vl[hf] = *reinterpret_cast<const s16x8*>(&Vsm[buf][((1 * 2 + hf) * 16) * 32 + voff]);
      }
      f32x4 s[QT][4];
      if (masked) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(&biasS[buf][(j >> 1) * 32 + g * 8 + (j & 1) * 4]);
#pragma unroll
          for (int u = 0; u < QT; ++u) s[u][j] = mfma_f16(kf[j], qhi[u], cin[u] + b4);
        }
      } else {
#pragma unroll
        for (int u = 0; u < QT; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) s[u][j] = mfma_f16(kf[j], qhi[u], cin[u]);
      }
#pragma unroll
      for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[u][j] = mfma_f16(kf[j], qlo[u], s[u][j]);

#pragma unroll
      for (int u = 0; u < QT; ++u) {
        // s = s2 - m_run + P_OFF.  Common path: nothing exceeds 2^(P_OFF + P_THR) -> exponentiate as is.
        const float mx = max16(s[u][0], s[u][1], s[u][2], s[u][3]);
        if (first || __builtin_amdgcn_ballot_w64(mx > P_OFF + P_THR) != 0ull) {
          const float cm = colmax4(mx);                                       // exact chunk max of the lane's query
          float shift = first ? (cm - P_OFF) : fmaxf(cm - P_OFF, 0.f);
          if (cm == -INFINITY) shift = 0.f;                                   // every key so far masked
          m_run[u] += shift;
          const float alpha = __builtin_amdgcn_exp2f(-shift);
#pragma unroll
          for (int r = 0; r < 4; ++r) { acc0[u][r] *= alpha; acc1[u][r] *= alpha; cin[u][r] -= shift; }
          if (DROP) l_run[u] *= alpha;
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[u][j][r] -= shift;
        }
        s16x8 pf[2];
        float l_tile = 0.f;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          unsigned int w[4];
          unsigned int keep = 0xFFu;
          if (DROP) keep = drop_keep8(dkey, (uint32_t)(c * (C16 / 8) + hf * 4 + g), (uint32_t)(qbase + u * 16 + li), (uint32_t)bh, drop_site, drop_thr);
#pragma unroll
          for (int T = 0; T < 2; ++T) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              const f32x4& sj = s[u][hf * 2 + T];
              float p0 = __builtin_amdgcn_exp2f(sj[2 * pr]), p1 = __builtin_amdgcn_exp2f(sj[2 * pr + 1]);
              if (DROP) {
                l_tile += p0 + p1;
                const int j = T * 4 + 2 * pr;
                p0 = ((keep >> j) & 1u) ? p0 * drop_scale : 0.f;
                p1 = ((keep >> (j + 1)) & 1u) ? p1 * drop_scale : 0.f;
              }
              w[T * 2 + pr] = pk_f16(p0, p1);
            }
          }
          pf[hf] = __builtin_bit_cast(s16x8, (u32x4_){w[0], w[1], w[2], w[3]});
        }
        if (DROP) l_run[u] += l_tile;
        // V keeps both parts: O = sum_k p~_k v_k / sum_k p~_k is an exactly normalised average of the 22-bit v rows, so
        // what is left of the P rounding is proportional to the spread of v under the weights, not to |v|
        acc0[u] = mfma_f16(vh[0], pf[0], acc0[u]);
        acc1[u] = mfma_f16(vh[1], pf[1], acc1[u]);
        acc0[u] = mfma_f16(vl[0], pf[0], acc0[u]);
        acc1[u] = mfma_f16(vl[1], pf[1], acc1[u]);
      }
    }
    if (has_next) stage_store(nxt, buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int u = 0; u < QT; ++u) {
    if (!active[u]) continue;
    f32x4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = acc0[u][r] + acc1[u][r];
    float l_tot;
    if (DROP) {
      l_tot = l_run[u] + __shfl_xor(l_run[u], 16, 64);
      l_tot += __shfl_xor(l_tot, 32, 64);
    } else {
      l_tot = __shfl(acc[3], 48 + li, 64);             // channel 15 (lane group g = 3, register 3) holds sum_k p
    }
    const int q = qbase + u * 16 + li;
    const float mq = m_run[u] - P_OFF;                 // p = 2^(s2 - mq)
    if (nsplit == 1) {
      const float inv = (l_tot > 0.f) ? 1.0f / l_tot : 0.f;
      if (q < Lq) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = g * 4 + r;
          if (d < HD) O[((size_t)b * Lq + q) * E + h * HD + d] = acc[r] * inv;
        }
      }
      if (g == 0) LSE2[bh * Lqp + q] = (l_tot > 0.f) ? (mq + __builtin_amdgcn_logf(l_tot)) : -INFINITY;
    } else {
      const size_t row = (((size_t)sp * B + b) * H + h) * Lqp + q;
      *reinterpret_cast<f32x4*>(&Op[row * HDP + g * 4]) = acc;
      if (g == 0) { Mp[row] = (l_tot > 0.f) ? mq : -INFINITY; Lp[row] = l_tot; }
    }
  }
}

__global__ __launch_bounds__(256) void attn16_combine_kernel(
    const float* __restrict__ Op, const float* __restrict__ Mp, const float* __restrict__ Lp,
    float* __restrict__ O, float* __restrict__ LSE2, int B, int H, int Lq, int Lqp, int nsplit) {
  const size_t rows = (size_t)B * H * Lqp;
  const int E = H * HD;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < rows * HDP;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(idx & 15);
    const size_t row = idx >> 4;
    const int q = (int)(row % Lqp);
    const size_t bh = row / Lqp;
    const int h = (int)(bh % H), b = (int)(bh / H);
    float m = -INFINITY;
    for (int s = 0; s < nsplit; ++s) m = fmaxf(m, Mp[s * rows + row]);
    const float m_use = (m == -INFINITY) ? 0.f : m;
    float l = 0.f, o = 0.f;
    for (int s = 0; s < nsplit; ++s) {
      const float w = exp2f(Mp[s * rows + row] - m_use);
      l += Lp[s * rows + row] * w;
      o += Op[(s * rows + row) * HDP + d] * w;
    }
    if (d < HD && q < Lq) O[((size_t)b * Lq + q) * E + h * HD + d] = (l > 0.f) ? o / l : 0.f;
    if (d == 0) LSE2[row] = (l > 0.f) ? (m + log2f(l)) : -INFINITY;
  }
}

// ------------------------------------------------------------------------------------------------ backward: prep
// Per (b, h, q): e = exponent with max_d |dO ln2| / 2^e in [0.5, 1); dOn = fp16(dO ln2 2^-e) (rows + planes formats);
// D = sum_d dOn * O (from the ROUNDED dOn: the backward is the exact derivative for the upstream gradient dOn 2^e / ln2);
// rexp = e (int; -100 for an all-zero row).  grid (Lqp / 64, B)
__global__ __launch_bounds__(256) void attn16_bwd_prep_kernel(
    const float* __restrict__ dO, const float* __restrict__ O, unsigned short* __restrict__ dOr,
    unsigned short* __restrict__ dOp, float* __restrict__ D, int* __restrict__ rexp, int B, int H, int Lq, int Lqp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int E = H * HD;
  const int ldt = E + 1;
  float* Td = smem;
  float* To = smem + 64 * ldt;
  const int b = blockIdx.y, q0 = blockIdx.x * 64;
  for (int idx = threadIdx.x; idx < 64 * E; idx += blockDim.x) {
    const int r = idx / E, c = idx - r * E;
    const int q = q0 + r;
    float a = 0.f, o = 0.f;
    if (q < Lq) {
      a = dO[((size_t)b * Lq + q) * E + c] * LN2_F;
      o = O[((size_t)b * Lq + q) * E + c];
    }
    Td[r * ldt + c] = a;
    To[r * ldt + c] = o;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 64 * H; idx += blockDim.x) {
    const int r = idx & 63, h = idx >> 6;
    float* td = Td + r * ldt + h * HD;
    const float* to = To + r * ldt + h * HD;
    float mx = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) mx = fmaxf(mx, fabsf(td[d]));
    int e = -100;
    if (mx > 0.f && mx < INFINITY) (void)frexpf(mx, &e);
    if (e < -100) e = -100;
    const float inv = ldexpf(1.0f, -e);
    float dsum = 0.f;
    unsigned int w[8];
    float v[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) v[d] = (d < HD) ? td[d] * inv : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = pk_f16(v[2 * i], v[2 * i + 1]);
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      const unsigned short hb = (unsigned short)((d & 1) ? (w[d >> 1] >> 16) : (w[d >> 1] & 0xFFFFu));
      const float rv = (float)__builtin_bit_cast(_Float16, hb);
      td[d] = rv;                                           // the planes pass re-reads the rounded values
      dsum += rv * to[d];
    }
    const size_t row = ((size_t)b * H + h) * Lqp + q0 + r;
    *reinterpret_cast<u32x4_*>(dOr + row * 16) = (u32x4_){w[0], w[1], w[2], w[3]};
    *reinterpret_cast<u32x4_*>(dOr + row * 16 + 8) = (u32x4_){w[4], w[5], w[6], w[7]};
    D[row] = dsum;
    rexp[row] = e;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < H * 16 * 8; idx += blockDim.x) {
    const int seg = idx & 7;
    const int d = (idx >> 3) & 15;
    const int h = idx >> 7;
    unsigned int w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = seg * 8 + 2 * j;
      const float v0 = (d < HD) ? Td[r * ldt + h * HD + d] : 0.f;
      const float v1 = (d < HD) ? Td[(r + 1) * ldt + h * HD + d] : 0.f;
      w[j] = pk_f16(v0, v1);
    }
    *reinterpret_cast<u32x4_*>(dOp + (((size_t)b * H + h) * 16 + d) * Lqp + q0 + seg * 8) = (u32x4_){w[0], w[1], w[2], w[3]};
  }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
struct Dq16Stage {
  s16x8 k, v, kp;
  float bias;
};

// dQ2[q] = 2^e_q * sum_k G[q, k] K[k],  G = P o (dPn - Dn)  (dPn = V dOn: the ln 2 of d/ds2 rides in dOn)
// DROP: O = (M o P) V  ->  dPn = M o (V dOn), G = P o (dPn - Dn).
// PL: parts of the K planes contracted with G (2: K_hi G + K_lo G -- sum_k G = 0 makes dQ a function of key DIFFERENCES, so
// a 2^-12 rounding of K is amplified by |K| / |K - K'| between the keys that share a query's weight; 1: single fp16)
template <bool DROP, int QT, int PL>
__global__ __launch_bounds__(256, 2) void attn16_bwd_dq_kernel(
    const unsigned short* __restrict__ Qr, const unsigned short* __restrict__ Kr, const unsigned short* __restrict__ Kp,
    const unsigned short* __restrict__ Vr, const unsigned char* __restrict__ kmask, const unsigned short* __restrict__ dOr,
    const float* __restrict__ LSE2, const float* __restrict__ D, const int* __restrict__ rexp, float* __restrict__ dQp,
    int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit, const unsigned long long* __restrict__ drop_state,
    unsigned int drop_site, unsigned int drop_thr, float drop_scale) {
  __shared__ __attribute__((aligned(16))) unsigned short Ksm[2][C16 * 32];      // [k_hi | k_lo] rows tile
  __shared__ __attribute__((aligned(16))) unsigned short Vsm[2][C16 * 32];      // [v_hi | v_lo] rows tile
  __shared__ __attribute__((aligned(16))) unsigned short Kpm[2][PL * 2 * 16 * 32];   // K planes [part][32-key half][16][32]
  __shared__ __attribute__((aligned(16))) float biasS[2][C16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  constexpr int QW = 64 * QT;
  const int tiles_x = (Lqp + QW - 1) / QW;
  int group, within;
  if (!xcd_decode(tiles_x * nsplit, B * H, group, within)) return;
  const int b = group / H, h = group - b * H;
  const int sp = within / tiles_x;
  const size_t bh = (size_t)b * H + h;
  const int qbase = (within - sp * tiles_x) * QW + wave * (16 * QT);

  s16x8 qhi[QT], qlo[QT], dod[QT];
  f32x4 cS[QT], cD[QT];                  // accumulator inits: -lse2 (score tiles), -Dn (dP tiles)
  float nd[QT], rs[QT];
  bool active[QT];
  bool any_active = false;
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const int q = qbase + u * 16 + li;
    active[u] = (qbase + u * 16) < Lqp;
    any_active = any_active || active[u];
    qhi[u] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    qlo[u] = qhi[u]; dod[u] = qhi[u];
    float lse_q = INFINITY, d_q = 0.f;
    rs[u] = 0.f;
    if (active[u]) {
      const unsigned short* qp = Qr + (bh * Lqp + q) * 32;
      qhi[u] = *reinterpret_cast<const s16x8*>(qp + (g & 1) * 8);
      qlo[u] = *reinterpret_cast<const s16x8*>(qp + 16 + (g & 1) * 8);
      dod[u] = *reinterpret_cast<const s16x8*>(dOr + (bh * Lqp + q) * 16 + (g & 1) * 8);     // [dOn | dOn]
      if (q < Lq) {
        lse_q = LSE2[bh * Lqp + q];
        if (lse_q == -INFINITY) lse_q = INFINITY;
        d_q = D[bh * Lqp + q];
        rs[u] = ldexpf(1.0f, rexp[bh * Lqp + q]);
      }
    }
    cS[u] = f32x4{-lse_q, -lse_q, -lse_q, -lse_q};
    nd[u] = -d_q;
    cD[u] = DROP ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{-d_q, -d_q, -d_q, -d_q};
  }
  const int nch = Sp / C16;
  const int cps = (nch + nsplit - 1) / nsplit;
  const int c_beg = sp * cps, c_end = min(nch, c_beg + cps);
  const int krow = t >> 2, kseg = t & 3;
  const int vplane = t >> 7, vd = (t >> 3) & 15, vseg = t & 7;
  auto stage_load = [&](int c) {
    Dq16Stage st;
    st.k = *reinterpret_cast<const s16x8*>(Kr + (bh * Sp + (size_t)c * C16 + krow) * 32 + kseg * 8);
    st.v = *reinterpret_cast<const s16x8*>(Vr + (bh * Sp + (size_t)c * C16 + krow) * 32 + kseg * 8);
    if (vplane < PL) st.kp = *reinterpret_cast<const s16x8*>(Kp + ((bh * PL + vplane) * 16 + vd) * Sp + (size_t)c * C16 + vseg * 8);
    st.bias = 0.f;
    if (t < C16) {
      const int key = c * C16 + t;
      bool valid = key < S;
      if (valid && kmask) valid = kmask[(size_t)b * S + key] == 0;
      st.bias = valid ? 0.f : -INFINITY;
    }
    return st;
  };
  auto stage_store = [&](const Dq16Stage& st, int buf) {
    *reinterpret_cast<s16x8*>(&Ksm[buf][tile_off(krow, kseg)]) = st.k;
    *reinterpret_cast<s16x8*>(&Vsm[buf][tile_off(krow, kseg)]) = st.v;
    if (vplane < PL) *reinterpret_cast<s16x8*>(&Kpm[buf][((vplane * 2 + (vseg >> 2)) * 16) * 32 + plane_off(vd, vseg & 3)]) = st.kp;
    if (t < C16) biasS[buf][t] = st.bias;
  };

  int koff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) koff[j] = tile_off((j >> 1) * 32 + (li >> 2) * 8 + (li & 3) + (j & 1) * 4, g);
  const int poff = plane_off(li, g);

  f32x4 acc0[QT], acc1[QT];
#pragma unroll
  for (int u = 0; u < QT; ++u) { acc0[u] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[u] = acc0[u]; }
  DropKey dkey = {0u, 0u};
  if (DROP) dkey = drop_key(drop_state);
  if (c_beg < c_end) {
    stage_store(stage_load(c_beg), 0);
    __syncthreads();
  }
  for (int c = c_beg; c < c_end; ++c) {
    const int buf = (c - c_beg) & 1;
    Dq16Stage nxt;
    const bool has_next = (c + 1 < c_end);
    if (has_next) nxt = stage_load(c + 1);
    const bool masked = (kmask != nullptr) || ((c + 1) * C16 > S);
    if (any_active) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        s16x8 kf[2], vf[2];
#pragma unroll
        for (int T = 0; T < 2; ++T) {
          kf[T] = *reinterpret_cast<const s16x8*>(&Ksm[buf][koff[hf * 2 + T]]);
          vf[T] = *reinterpret_cast<const s16x8*>(&Vsm[buf][koff[hf * 2 + T]]);
        }
        s16x8 kp[PL];
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) kp[pl] = *reinterpret_cast<const s16x8*>(&Kpm[buf][((pl * 2 + hf) * 16) * 32 + poff]);
        f32x4 sT[QT][2], dpT[QT][2];
        if (masked) {
#pragma unroll
          for (int T = 0; T < 2; ++T) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(&biasS[buf][hf * 32 + g * 8 + T * 4]);
#pragma unroll
            for (int u = 0; u < QT; ++u) sT[u][T] = mfma_f16(kf[T], qhi[u], cS[u] + b4);
          }
        } else {
#pragma unroll
          for (int u = 0; u < QT; ++u)
#pragma unroll
            for (int T = 0; T < 2; ++T) sT[u][T] = mfma_f16(kf[T], qhi[u], cS[u]);
        }
#pragma unroll
        for (int u = 0; u < QT; ++u)
#pragma unroll
          for (int T = 0; T < 2; ++T) {
            dpT[u][T] = mfma_f16(vf[T], dod[u], cD[u]);
            sT[u][T] = mfma_f16(kf[T], qlo[u], sT[u][T]);
          }
#pragma unroll
        for (int u = 0; u < QT; ++u) {
          unsigned int keep = 0xFFu;
          if (DROP) keep = drop_keep8(dkey, (uint32_t)(c * (C16 / 8) + hf * 4 + g), (uint32_t)(qbase + u * 16 + li), (uint32_t)bh, drop_site, drop_thr);
          unsigned int w[4];
#pragma unroll
          for (int T = 0; T < 2; ++T) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              const float p0 = __builtin_amdgcn_exp2f(sT[u][T][2 * pr]), p1 = __builtin_amdgcn_exp2f(sT[u][T][2 * pr + 1]);
              float g0, g1;
              if (DROP) {
                const int j = T * 4 + 2 * pr;
                const float m0 = ((keep >> j) & 1u) ? drop_scale : 0.f, m1 = ((keep >> (j + 1)) & 1u) ? drop_scale : 0.f;
                g0 = p0 * __builtin_fmaf(m0, dpT[u][T][2 * pr], nd[u]);
                g1 = p1 * __builtin_fmaf(m1, dpT[u][T][2 * pr + 1], nd[u]);
              } else {
                g0 = p0 * dpT[u][T][2 * pr];
                g1 = p1 * dpT[u][T][2 * pr + 1];
              }
              w[T * 2 + pr] = pk_f16(g0, g1);
            }
          }
          const s16x8 gf = __builtin_bit_cast(s16x8, (u32x4_){w[0], w[1], w[2], w[3]});
          f32x4& acc = hf ? acc1[u] : acc0[u];
#pragma unroll
          for (int pl = 0; pl < PL; ++pl) acc = mfma_f16(kp[pl], gf, acc);
        }
      }
    }
    if (has_next) stage_store(nxt, buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    if (!active[u]) continue;
    const size_t row = (((size_t)sp * B + b) * H + h) * Lqp + qbase + u * 16 + li;
    f32x4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (acc0[u][r] + acc1[u][r]) * rs[u];
    *reinterpret_cast<f32x4*>(&dQp[row * HDP + g * 4]) = acc;
  }
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
struct Dkv16Stage {
  s16x8 q, o, qp, op;
  float nl, nd;
};

// A workgroup owns 64 KT keys of one (b, h) and walks the queries in 64-row chunks.  With E_bh = max_q e_q:
//   P'[q, k] = 2^(s2 - lse2 + e_q - E_bh)  (<= 1),  G' = P' o (dPn - Dn)
//   dV = log2(e) 2^E_bh sum_q P'[q, k] dOn[q],   dK = 2^E_bh sum_q G'[q, k] Q2[q]
// PL: parts of the Q planes contracted with G' (as for K in the dQ kernel)
template <int KT, int PL>
__global__ __launch_bounds__(256, 2) void attn16_bwd_dkv_kernel(
    const unsigned short* __restrict__ Qr, const unsigned short* __restrict__ Qp, const unsigned short* __restrict__ Kr,
    const unsigned short* __restrict__ Vr, const unsigned char* __restrict__ kmask, const unsigned short* __restrict__ dOr,
    const unsigned short* __restrict__ dOp, const float* __restrict__ LSE2, const float* __restrict__ D,
    const int* __restrict__ rexp, float* __restrict__ dK, float* __restrict__ dV, int B, int H, int Lq, int Lqp, int S,
    int Sp) {
  __shared__ __attribute__((aligned(16))) unsigned short Qsm[2][C16 * 32];      // [q_hi | q_lo] rows tile
  __shared__ __attribute__((aligned(16))) unsigned short Osm[2][C16 * 32];      // [dOn | dOn]   rows tile
  __shared__ __attribute__((aligned(16))) unsigned short Qpm[2][PL * 2 * 16 * 32];   // Q planes [part][32-query half][16][32]
  __shared__ __attribute__((aligned(16))) unsigned short Opm[2][2 * 16 * 32];   // dOn plane
  __shared__ __attribute__((aligned(16))) float nlS[2][C16];
  __shared__ __attribute__((aligned(16))) float ndS[2][C16];
  __shared__ int emaxS[4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  constexpr int KW = 64 * KT;
  int group, within;
  if (!xcd_decode((Sp + KW - 1) / KW, B * H, group, within)) return;
  const int b = group / H, h = group - b * H;
  const size_t bh = (size_t)b * H + h;

  // E_bh = max_q e_q over the (b, h)'s query rows
  {
    int e = -100;
    for (int q = t; q < Lq; q += 256) e = max(e, rexp[bh * Lqp + q]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e = max(e, __shfl_xor(e, o, 64));
    if (lane == 0) emaxS[wave] = e;
  }
  __syncthreads();
  const int emax = max(max(emaxS[0], emaxS[1]), max(emaxS[2], emaxS[3]));

  s16x8 khh[KT], kll[KT], vB[KT];
  f32x4 bias4[KT];
  int key[KT];
#pragma unroll
  for (int u = 0; u < KT; ++u) {
    key[u] = within * KW + (wave * KT + u) * 16 + li;
    khh[u] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    kll[u] = khh[u]; vB[u] = khh[u];
    bool valid = false;
    if (key[u] < Sp) {
      const unsigned short* kp = Kr + (bh * Sp + key[u]) * 32;
      const unsigned short* vp = Vr + (bh * Sp + key[u]) * 32;
      khh[u] = *reinterpret_cast<const s16x8*>(kp + (g & 1) * 8);               // B = [k_hi | k_hi]
      kll[u] = *reinterpret_cast<const s16x8*>(kp + 16 + (g & 1) * 8);          // B = [k_lo | k_lo]
      vB[u] = *reinterpret_cast<const s16x8*>(vp + g * 8);                      // B = [v_hi | v_lo] against A = [dOn | dOn]
      valid = key[u] < S;
      if (valid && kmask) valid = kmask[(size_t)b * S + key[u]] == 0;
    }
    const float bias_k = valid ? 0.f : -INFINITY;
    bias4[u] = f32x4{bias_k, bias_k, bias_k, bias_k};
  }
  const bool masked = (kmask != nullptr) || (within * KW + KW > S);           // workgroup-uniform

  const int qrow = t >> 2, qseg = t & 3;
  const int pplane = t >> 7, pd = (t >> 3) & 15, pseg = t & 7;
  auto stage_load = [&](int c) {
    Dkv16Stage st;
    st.q = *reinterpret_cast<const s16x8*>(Qr + (bh * Lqp + (size_t)c * C16 + qrow) * 32 + qseg * 8);
    st.o = *reinterpret_cast<const s16x8*>(dOr + (bh * Lqp + (size_t)c * C16 + qrow) * 16 + (qseg & 1) * 8);
    if (pplane < PL) st.qp = *reinterpret_cast<const s16x8*>(Qp + ((bh * PL + pplane) * 16 + pd) * Lqp + (size_t)c * C16 + pseg * 8);
    // the dOn plane is staged by the upper half of the workgroup (PL = 1: the half that has no Q plane to fetch)
    if (t >= 128) st.op = *reinterpret_cast<const s16x8*>(dOp + (bh * 16 + pd) * Lqp + (size_t)c * C16 + pseg * 8);
    st.nl = -INFINITY;
    st.nd = 0.f;
    if (t < C16) {
      const int qq = c * C16 + t;
      if (qq < Lq) {
        const float l = LSE2[bh * Lqp + qq];
        if (l != -INFINITY) st.nl = (float)(rexp[bh * Lqp + qq] - emax) - l;
        st.nd = -D[bh * Lqp + qq];
      }
    }
    return st;
  };
  auto stage_store = [&](const Dkv16Stage& st, int buf) {
    *reinterpret_cast<s16x8*>(&Qsm[buf][tile_off(qrow, qseg)]) = st.q;
    *reinterpret_cast<s16x8*>(&Osm[buf][tile_off(qrow, qseg)]) = st.o;
    const int po = (pseg >> 2) * 16 * 32 + plane_off(pd, pseg & 3);
    if (pplane < PL) *reinterpret_cast<s16x8*>(&Qpm[buf][pplane * 2 * 16 * 32 + po]) = st.qp;
    if (t >= 128) *reinterpret_cast<s16x8*>(&Opm[buf][po]) = st.op;
    if (t < C16) { nlS[buf][t] = st.nl; ndS[buf][t] = st.nd; }
  };

  int qoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) qoff[j] = tile_off((j >> 1) * 32 + (li >> 2) * 8 + (li & 3) + (j & 1) * 4, g);
  const int poff = plane_off(li, g);

  f32x4 dk0[KT], dk1[KT], dv0[KT], dv1[KT];
#pragma unroll
  for (int u = 0; u < KT; ++u) { dk0[u] = f32x4{0.f, 0.f, 0.f, 0.f}; dk1[u] = dk0[u]; dv0[u] = dk0[u]; dv1[u] = dk0[u]; }
  const int nch = Lqp / C16;
  stage_store(stage_load(0), 0);
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const int buf = c & 1;
    Dkv16Stage nxt;
    const bool has_next = (c + 1 < nch);
    if (has_next) nxt = stage_load(c + 1);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      s16x8 qf[2], of[2];
      f32x4 nl4[2], nd4[2];
#pragma unroll
      for (int T = 0; T < 2; ++T) {
        qf[T] = *reinterpret_cast<const s16x8*>(&Qsm[buf][qoff[hf * 2 + T]]);
        of[T] = *reinterpret_cast<const s16x8*>(&Osm[buf][qoff[hf * 2 + T]]);
        nl4[T] = *reinterpret_cast<const f32x4*>(&nlS[buf][hf * 32 + g * 8 + T * 4]);
        nd4[T] = *reinterpret_cast<const f32x4*>(&ndS[buf][hf * 32 + g * 8 + T * 4]);
      }
      const s16x8 otp = *reinterpret_cast<const s16x8*>(&Opm[buf][hf * 16 * 32 + poff]);
      s16x8 qtp[PL];
#pragma unroll
      for (int pl = 0; pl < PL; ++pl) qtp[pl] = *reinterpret_cast<const s16x8*>(&Qpm[buf][((pl * 2 + hf) * 16) * 32 + poff]);
      f32x4 s[KT][2], dp[KT][2];
      if (masked) {
#pragma unroll
        for (int u = 0; u < KT; ++u)
#pragma unroll
          for (int T = 0; T < 2; ++T) s[u][T] = mfma_f16(qf[T], khh[u], nl4[T] + bias4[u]);
      } else {
#pragma unroll
        for (int u = 0; u < KT; ++u)
#pragma unroll
          for (int T = 0; T < 2; ++T) s[u][T] = mfma_f16(qf[T], khh[u], nl4[T]);
      }
#pragma unroll
      for (int u = 0; u < KT; ++u)
#pragma unroll
        for (int T = 0; T < 2; ++T) {
          dp[u][T] = mfma_f16(of[T], vB[u], nd4[T]);
          s[u][T] = mfma_f16(qf[T], kll[u], s[u][T]);
        }
#pragma unroll
      for (int u = 0; u < KT; ++u) {
        unsigned int pw[4], gw[4];
#pragma unroll
        for (int T = 0; T < 2; ++T) {
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const float p0 = __builtin_amdgcn_exp2f(s[u][T][2 * pr]), p1 = __builtin_amdgcn_exp2f(s[u][T][2 * pr + 1]);
            pw[T * 2 + pr] = pk_f16(p0, p1);
            gw[T * 2 + pr] = pk_f16(p0 * dp[u][T][2 * pr], p1 * dp[u][T][2 * pr + 1]);
          }
        }
        const s16x8 pf = __builtin_bit_cast(s16x8, (u32x4_){pw[0], pw[1], pw[2], pw[3]});
        const s16x8 gf = __builtin_bit_cast(s16x8, (u32x4_){gw[0], gw[1], gw[2], gw[3]});
        f32x4& dv = hf ? dv1[u] : dv0[u];
        f32x4& dk = hf ? dk1[u] : dk0[u];
        dv = mfma_f16(otp, pf, dv);
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) dk = mfma_f16(qtp[pl], gf, dk);
      }
    }
    if (has_next) stage_store(nxt, buf ^ 1);
    __syncthreads();
  }
  const float sk = ldexpf(1.0f, emax), sv = sk * LOG2E_F;
#pragma unroll
  for (int u = 0; u < KT; ++u) {
    if (key[u] >= Sp) continue;
    f32x4 dk, dv;
#pragma unroll
    for (int r = 0; r < 4; ++r) { dk[r] = (dk0[u][r] + dk1[u][r]) * sk; dv[r] = (dv0[u][r] + dv1[u][r]) * sv; }
    *reinterpret_cast<f32x4*>(&dK[(bh * Sp + key[u]) * HDP + g * 4]) = dk;
    *reinterpret_cast<f32x4*>(&dV[(bh * Sp + key[u]) * HDP + g * 4]) = dv;
  }
}

}  // namespace a3d

using namespace a3d;

static int check16(const char* fn, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit, int qmod) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lqp < Lq || (Lqp % qmod) != 0 || S <= 0 || Sp < S || (Sp % C16) != 0 || nsplit < 1 ||
      nsplit > 64) {
    set_error("%s: bad argument (B=%d H=%d Lq=%d Lqp=%d S=%d Sp=%d nsplit=%d; need Lqp %% %d == 0, Sp %% 64 == 0)", fn, B, H,
              Lq, Lqp, S, Sp, nsplit, qmod);
    return A3D_ERR_ARG;
  }
  return A3D_OK;
}

static int drop_params(const char* fn, const unsigned long long* drop_state, float drop_p, bool& drop, unsigned int& thr,
                       float& dscale) {
  drop = drop_state != nullptr && drop_p > 0.f;
  if (drop_state && !(drop_p >= 0.f && drop_p < 1.f)) {
    set_error("%s: dropout probability %g outside [0, 1)", fn, (double)drop_p);
    return A3D_ERR_ARG;
  }
  thr = drop ? (unsigned int)lrintf(drop_p * 65536.0f) : 0u;
  dscale = drop ? 1.0f / (1.0f - drop_p) : 1.0f;
  return A3D_OK;
}

extern "C" int a3d_attn16_fwd(const void* Qr, const void* Kr, const void* Vp, const unsigned char* kmask, float* O,
                              float* LSE2, float* ws, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit,
                              const unsigned long long* drop_state, unsigned int drop_site, float drop_p, void* stream) {
  int rc = check16("a3d_attn16_fwd", B, H, Lq, Lqp, S, Sp, nsplit, 16);
  if (rc) return rc;
  if (!Qr || !Kr || !Vp || !O || !LSE2 || (nsplit > 1 && !ws)) { set_error("a3d_attn16_fwd: null pointer"); return A3D_ERR_ARG; }
  bool drop; unsigned int thr; float dscale;
  rc = drop_params("a3d_attn16_fwd", drop_state, drop_p, drop, thr, dscale);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const size_t rows = (size_t)B * H * Lqp;
  float* Op = ws;
  float* Mp = ws ? ws + (size_t)nsplit * rows * HDP : nullptr;
  float* Lp = ws ? Mp + (size_t)nsplit * rows : nullptr;
  static const int qt_env = getenv("A3D_ATTN_QT") ? atoi(getenv("A3D_ATTN_QT")) : 0;
  const int QT = qt_env ? qt_env : (Lq > 64 ? 2 : 1);
  dim3 grid(xcd_grid(B * H, cdiv(Lqp, 64 * QT) * nsplit));
  const unsigned long long* nostate = nullptr;
#define A3D_L16F(DROPV, QTV, ST, SITE, THR, SC)                                                                          \
  hipLaunchKernelGGL((attn16_fwd_kernel<DROPV, QTV>), grid, dim3(256), 0, s, (const unsigned short*)Qr,                 \
                     (const unsigned short*)Kr, (const unsigned short*)Vp, kmask, O, LSE2, Op, Mp, Lp, B, H, Lq, Lqp, S, \
                     Sp, nsplit, ST, SITE, THR, SC)
  if (drop) {
    if (QT == 2) A3D_L16F(true, 2, drop_state, drop_site, thr, dscale);
    else A3D_L16F(true, 1, drop_state, drop_site, thr, dscale);
  } else {
    if (QT == 2) A3D_L16F(false, 2, nostate, 0u, 0u, 1.0f);
    else A3D_L16F(false, 1, nostate, 0u, 0u, 1.0f);
  }
#undef A3D_L16F
  rc = check_launch("a3d_attn16_fwd");
  if (rc) return rc;
  if (nsplit > 1) {
    const int cg = (int)std::min<size_t>((rows * HDP + 255) / 256, 4096);
    hipLaunchKernelGGL(attn16_combine_kernel, dim3(cg), dim3(256), 0, s, Op, Mp, Lp, O, LSE2, B, H, Lq, Lqp, nsplit);
    rc = check_launch("a3d_attn16_fwd(combine)");
  }
  return rc;
}

extern "C" int a3d_attn16_bwd(const void* Qr, const void* Qp, const void* Kr, const void* Kp, int plane_parts, const void* Vr,
                              const unsigned char* kmask, const float* O, const float* dO, const float* LSE2, void* dOr,
                              void* dOp, float* D, int* rexp, float* dQp, float* dK, float* dV, int B, int H, int Lq,
                              int Lqp, int S, int Sp, int nsplit, const unsigned long long* drop_state,
                              unsigned int drop_site, float drop_p, void* stream) {
  int rc = check16("a3d_attn16_bwd", B, H, Lq, Lqp, S, Sp, nsplit, 64);
  if (rc) return rc;
  if (!Qr || !Qp || !Kr || !Kp || !Vr || !O || !dO || !LSE2 || !dOr || !dOp || !D || !rexp || !dQp || !dK || !dV) {
    set_error("a3d_attn16_bwd: null pointer");
    return A3D_ERR_ARG;
  }
  bool drop; unsigned int thr; float dscale;
  rc = drop_params("a3d_attn16_bwd", drop_state, drop_p, drop, thr, dscale);
  if (rc) return rc;
  if (drop) { set_error("a3d_attn16_bwd: attention-weight dropout is not implemented in the fp16 backward"); return A3D_ERR_ARG; }
  if (plane_parts != 1 && plane_parts != 2) { set_error("a3d_attn16_bwd: plane_parts must be 1 or 2, got %d", plane_parts); return A3D_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const int E = H * HD;
  const size_t lds = (size_t)2 * 64 * (E + 1) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn16_bwd_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(attn16_bwd_prep_kernel, dim3(Lqp / 64, B), dim3(256), lds, s, dO, O, (unsigned short*)dOr,
                     (unsigned short*)dOp, D, rexp, B, H, Lq, Lqp);
  rc = check_launch("a3d_attn16_bwd(prep)");
  if (rc) return rc;
  static const int qt_env = getenv("A3D_ATTN_QT") ? atoi(getenv("A3D_ATTN_QT")) : 0;
  const int QT = qt_env ? qt_env : (Lq > 64 ? 2 : 1);
  const int KT = qt_env ? qt_env : (((size_t)B * H * (Sp / 128) >= 1024 && Lq > 16) ? 2 : 1);
  const dim3 gq(xcd_grid(B * H, cdiv(Lqp, 64 * QT) * nsplit)), gk(xcd_grid(B * H, cdiv(Sp, 64 * KT)));
  const unsigned long long* nostate = nullptr;
#define A3D_L16Q(QTV, PLV)                                                                                                \
  hipLaunchKernelGGL((attn16_bwd_dq_kernel<false, QTV, PLV>), gq, dim3(256), 0, s, (const unsigned short*)Qr,            \
                     (const unsigned short*)Kr, (const unsigned short*)Kp, (const unsigned short*)Vr, kmask,              \
                     (const unsigned short*)dOr, LSE2, D, rexp, dQp, B, H, Lq, Lqp, S, Sp, nsplit, nostate, 0u, 0u, 1.0f)
#define A3D_L16K(KTV, PLV)                                                                                                \
  hipLaunchKernelGGL((attn16_bwd_dkv_kernel<KTV, PLV>), gk, dim3(256), 0, s, (const unsigned short*)Qr,                  \
                     (const unsigned short*)Qp, (const unsigned short*)Kr, (const unsigned short*)Vr, kmask,              \
                     (const unsigned short*)dOr, (const unsigned short*)dOp, LSE2, D, rexp, dK, dV, B, H, Lq, Lqp, S, Sp)
  if (plane_parts == 2) { if (QT == 2) A3D_L16Q(2, 2); else A3D_L16Q(1, 2); }
  else { if (QT == 2) A3D_L16Q(2, 1); else A3D_L16Q(1, 1); }
  rc = check_launch("a3d_attn16_bwd(dq)");
  if (rc) return rc;
  if (plane_parts == 2) { if (KT == 2) A3D_L16K(2, 2); else A3D_L16K(1, 2); }
  else { if (KT == 2) A3D_L16K(2, 1); else A3D_L16K(1, 1); }
#undef A3D_L16Q
#undef A3D_L16K
  return check_launch("a3d_attn16_bwd(dkv)");
}
