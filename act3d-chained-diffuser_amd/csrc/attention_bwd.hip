// Attention backward on split-bf16 MFMA (v_mfma_f32_16x16x32_bf16) -- same operand discipline as the forward.
//
// dQ, dK, dV of  O = softmax(Q K^T + mask) V  by recomputation from the saved log-sum-exp.  Every contraction has
// both operands as bf16 hi+lo pairs (2^-17 operand precision, fp32 accumulate):
//     S   = Q K^T          3 MFMAs / 16x16 tile     [x_hi | x_lo] . [y_hi | y_hi]  +  [x_hi | x_lo] . [y_lo | y_lo]
//                                                   + [x_hi | x_lo2] . [y_lo2 | y_hi]   (three-part q, k: fp32-grade logits,
//                                                   the same scores the forward saw -> P_bwd == P_fwd)
//     dP  = dO V^T         2 MFMAs / 16x16 tile
//     dQ += dS K, dK += dS^T Q, dV += P^T dO        3 MFMAs / (16 x 32-deep) block:  A_hi B_hi + A_hi B_lo + A_lo B_hi
// against 4 f32 MFMAs (32 cycles each) per 16x16x16 block in the exact-f32 version it replaces (attention.hip,
// kept as the A/B reference behind A3D_BWD_F32=1): 11-14 bf16 MFMAs x 16 cycles per 16x32 block instead of 24-32 x 32.
// As in the forward, the score tiles are produced in the orientation (S^T for dQ, S for dK/dV) and with the row
// interleave that leaves P / dS in B-operand order in registers: no score tile is staged through LDS.
// Operand tensors (rope.hip / attn_bwd_prep): rows format [..][N][hi16|lo16] and planes format [..][2][16][N]
// of q, k, v and dO.
#include "a3d_common.h"
#include "../../include/act3d_hip.h"
#include <stdlib.h>

namespace a3d {

constexpr int BKC = 64;         // rows (keys or queries) per staged chunk

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// 8 floats -> hi and lo bf16x8 (round-to-nearest-even, x ~= hi + lo)
__device__ __forceinline__ void split8(const float* x, s16x8& hi, s16x8& lo) {
  unsigned int hw[4], lw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bf16x2_t hb = __builtin_convertvector((f32x2_t){x[2 * i], x[2 * i + 1]}, bf16x2_t);
    const unsigned int h2 = __builtin_bit_cast(unsigned int, hb);
    const float r0 = x[2 * i] - __uint_as_float(h2 << 16);
    const float r1 = x[2 * i + 1] - __uint_as_float(h2 & 0xFFFF0000u);
    const bf16x2_t lb = __builtin_convertvector((f32x2_t){r0, r1}, bf16x2_t);
    hw[i] = h2;
    lw[i] = __builtin_bit_cast(unsigned int, lb);
  }
  hi = __builtin_bit_cast(s16x8, (u32x4_t){hw[0], hw[1], hw[2], hw[3]});
  lo = __builtin_bit_cast(s16x8, (u32x4_t){lw[0], lw[1], lw[2], lw[3]});
}

// ------------------------------------------------------------------------------------------------ prep
// dOs (rows format), dOt (planes format) of dO, and D[b][h][q] = sum_d dO * O.  grid (Lqp/64, B)
__global__ __launch_bounds__(256) void attn_bwd_prep_bf16_kernel(
    const float* __restrict__ dO, const float* __restrict__ O, unsigned short* __restrict__ dOs,
    unsigned short* __restrict__ dOt, float* __restrict__ D, int B, int H, int Lq, int Lqp) {
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
      a = dO[((size_t)b * Lq + q) * E + c];
      o = O[((size_t)b * Lq + q) * E + c];
    }
    Td[r * ldt + c] = a;
    To[r * ldt + c] = o;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 64 * H; idx += blockDim.x) {
    const int r = idx & 63, h = idx >> 6;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s += Td[r * ldt + h * HD + d] * To[r * ldt + h * HD + d];
    D[((size_t)b * H + h) * Lqp + q0 + r] = s;
  }
  for (int idx = threadIdx.x; idx < 64 * H * 4; idx += blockDim.x) {
    const int seg = idx & 3;
    const int r = (idx >> 2) & 63;
    const int h = idx >> 8;
    const int dbase = (seg & 1) * 8;
    const bool want_lo = (seg >> 1) != 0;
    s16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = dbase + j;
      const float v = (d < HD) ? Td[r * ldt + h * HD + d] : 0.f;
      unsigned short hi, lo;
      split_bf16(v, hi, lo);
      out[j] = (short)(want_lo ? lo : hi);
    }
    *reinterpret_cast<s16x8*>(dOs + (((size_t)b * H + h) * Lqp + q0 + r) * 32 + seg * 8) = out;
  }
  for (int idx = threadIdx.x; idx < H * 2 * 16 * 8; idx += blockDim.x) {
    const int seg = idx & 7;
    const int d = (idx >> 3) & 15;
    const int plane = (idx >> 7) & 1;
    const int h = idx >> 8;
    s16x8 out;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = seg * 8 + j;
      const float v = (d < HD) ? Td[r * ldt + h * HD + d] : 0.f;
      unsigned short hi, lo;
      split_bf16(v, hi, lo);
      out[j] = (short)(plane ? lo : hi);
    }
    *reinterpret_cast<s16x8*>(dOt + ((((size_t)b * H + h) * 2 + plane) * 16 + d) * Lqp + q0 + seg * 8) = out;
  }
}

// ------------------------------------------------------------------------------------------------ dQ
struct DqStage {
  s16x8 k, v, kt, k2;
  float bias;
};

// DROP (both kernels): O = (M o P) V with M = keep / (1 - p) regenerated from the Philox counters the forward used, so
// dV += (M o P)^T dO, dP = M o (dO V^T), dS = P o (dP - D) with D = rowsum(dO o O) as without dropout.
// QT: 16-query tiles per wave (as in the forward: every staged chunk and every fragment read serves QT tiles).
template <bool DROP, int QT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_bf16_kernel(
    const unsigned short* __restrict__ Qs, const unsigned short* __restrict__ Ks,
    const unsigned short* __restrict__ Kt, const unsigned short* __restrict__ Vs,
    const unsigned char* __restrict__ kmask, const unsigned short* __restrict__ dOs,
    const float* __restrict__ LSE, const float* __restrict__ D, float* __restrict__ dQp, int B, int H, int Lq,
    int Lqp, int S, int Sp, int nsplit, const unsigned long long* __restrict__ drop_state, unsigned int drop_site,
    unsigned int drop_thr, float drop_scale) {
  __shared__ __attribute__((aligned(16))) unsigned short Ksm[2][BKC * 32];    // [k_hi | k_lo]  rows tile
  __shared__ __attribute__((aligned(16))) unsigned short K3sm[2][BKC * 32];   // [k_hi | k_lo2] rows tile
  __shared__ __attribute__((aligned(16))) unsigned short Vsm[2][BKC * 32];    // [v_hi | v_lo]  rows tile
  __shared__ __attribute__((aligned(16))) unsigned short Ktm[2][4 * 16 * 32]; // K planes [plane][32-key half][16][32]
  __shared__ __attribute__((aligned(16))) float biasS[2][BKC];
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

  s16x8 qhi[QT], qlo[QT], q3[QT], dohi[QT], dolo[QT];
  float nl[QT], nd[QT];                               // -lse log2 e, -D of this lane's query (column li)
  bool active[QT];
  bool any_active = false;
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const int q = qbase + u * 16 + li;
    active[u] = (qbase + u * 16) < Lqp;
    any_active = any_active || active[u];
    qhi[u] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    qlo[u] = qhi[u]; q3[u] = qhi[u]; dohi[u] = qhi[u]; dolo[u] = qhi[u];
    float lse_q = INFINITY, d_q = 0.f;
    if (active[u]) {
      const unsigned short* qp = Qs + (bh * Lqp + q) * QKW;
      qhi[u] = *reinterpret_cast<const s16x8*>(qp + (g & 1) * 8);
      qlo[u] = *reinterpret_cast<const s16x8*>(qp + 16 + (g & 1) * 8);
      q3[u] = *reinterpret_cast<const s16x8*>(qp + ((g < 2) ? 32 + g * 8 : (g & 1) * 8));   // [q_lo2 | q_hi], as in the forward
      const unsigned short* op = dOs + (bh * Lqp + q) * VRW;
      dohi[u] = *reinterpret_cast<const s16x8*>(op + (g & 1) * 8);
      dolo[u] = *reinterpret_cast<const s16x8*>(op + 16 + (g & 1) * 8);
      if (q < Lq) {
        lse_q = LSE[bh * Lqp + q];
        if (lse_q == -INFINITY) lse_q = INFINITY;
        d_q = D[bh * Lqp + q];
      }
    }
    nl[u] = -lse_q * LOG2E_F;
    nd[u] = -d_q;
  }
  const int nch = Sp / BKC;
  const int cps = (nch + nsplit - 1) / nsplit;
  const int c_beg = sp * cps, c_end = min(nch, c_beg + cps);
  const int krow = t >> 2, kseg = t & 3;
  const int vplane = t >> 7, vd = (t >> 3) & 15, vseg = t & 7;
  auto stage_load = [&](int c) {
    DqStage st;
    st.k = *reinterpret_cast<const s16x8*>(Ks + (bh * Sp + (size_t)c * BKC + krow) * QKW + kseg * 8);
    if (t < 2 * BKC) st.k2 = *reinterpret_cast<const s16x8*>(Ks + (bh * Sp + (size_t)c * BKC + (t >> 1)) * QKW + 32 + (t & 1) * 8);
    st.v = *reinterpret_cast<const s16x8*>(Vs + (bh * Sp + (size_t)c * BKC + krow) * VRW + kseg * 8);
    st.kt = *reinterpret_cast<const s16x8*>(Kt + ((bh * 2 + vplane) * 16 + vd) * Sp + (size_t)c * BKC + vseg * 8);
    st.bias = 0.f;
    if (t < BKC) {
      const int key = c * BKC + t;
      bool valid = key < S;
      if (valid && kmask) valid = kmask[(size_t)b * S + key] == 0;
      st.bias = valid ? 0.f : -INFINITY;
    }
    return st;
  };
  auto stage_store = [&](const DqStage& st, int buf) {
    *reinterpret_cast<s16x8*>(&Ksm[buf][tile_off(krow, kseg)]) = st.k;
    if (kseg < 2) *reinterpret_cast<s16x8*>(&K3sm[buf][tile_off(krow, kseg)]) = st.k;
    if (t < 2 * BKC) *reinterpret_cast<s16x8*>(&K3sm[buf][tile_off(t >> 1, 2 + (t & 1))]) = st.k2;
    *reinterpret_cast<s16x8*>(&Vsm[buf][tile_off(krow, kseg)]) = st.v;
    *reinterpret_cast<s16x8*>(&Ktm[buf][((vplane * 2 + (vseg >> 2)) * 16) * 32 + plane_off(vd, vseg & 3)]) = st.kt;
    if (t < BKC) biasS[buf][t] = st.bias;
  };

  int koff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) koff[j] = tile_off((j >> 1) * 32 + (li >> 2) * 8 + (li & 3) + (j & 1) * 4, g);
  const int poff = plane_off(li, g);

  f32x4 acc0[QT], acc1[QT];
#pragma unroll
  for (int u = 0; u < QT; ++u) { acc0[u] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[u] = acc0[u]; }
  // P = exp(s - lse) = exp2(fma(s, log2 e, -lse log2 e)) (one packed FMA per two scores); -D rides in as the MFMA
  // accumulator init of dP, so dS = P * (dP - D) is one packed multiply
  const f32x2_t c2 = {LOG2E_F, LOG2E_F};
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  DropKey dkey = {0u, 0u};
  if (DROP) dkey = drop_key(drop_state);
  if (c_beg < c_end) {
    stage_store(stage_load(c_beg), 0);
    __syncthreads();
  }
  for (int c = c_beg; c < c_end; ++c) {
    const int buf = (c - c_beg) & 1;
    DqStage nxt;
    const bool has_next = (c + 1 < c_end);
    if (has_next) nxt = stage_load(c + 1);
    if (any_active) {
      // per 32-key half: the fragment reads first (shared by the QT query tiles), then per tile four independent MFMA
      // chains (two score tiles, two dP tiles)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        s16x8 kf[2], k3[2], vf[2];
        f32x4 b4[2];
#pragma unroll
        for (int T = 0; T < 2; ++T) {
          kf[T] = *reinterpret_cast<const s16x8*>(&Ksm[buf][koff[hf * 2 + T]]);
          k3[T] = *reinterpret_cast<const s16x8*>(&K3sm[buf][koff[hf * 2 + T]]);
          vf[T] = *reinterpret_cast<const s16x8*>(&Vsm[buf][koff[hf * 2 + T]]);
          b4[T] = *reinterpret_cast<const f32x4*>(&biasS[buf][hf * 32 + g * 8 + T * 4]);
        }
        const s16x8 kth = *reinterpret_cast<const s16x8*>(&Ktm[buf][((0 * 2 + hf) * 16) * 32 + poff]);
        const s16x8 ktl = *reinterpret_cast<const s16x8*>(&Ktm[buf][((1 * 2 + hf) * 16) * 32 + poff]);
        f32x4 sT[QT][2], dpT[QT][2];
#pragma unroll
        for (int u = 0; u < QT; ++u) {
          const f32x4 nd4 = {nd[u], nd[u], nd[u], nd[u]};
#pragma unroll
          for (int T = 0; T < 2; ++T) {
            sT[u][T] = mfma_bf16_16x16x32(kf[T], qhi[u], b4[T]);
            dpT[u][T] = mfma_bf16_16x16x32(vf[T], dohi[u], DROP ? zero4 : nd4);
          }
        }
#pragma unroll
        for (int u = 0; u < QT; ++u)
#pragma unroll
          for (int T = 0; T < 2; ++T) {
            sT[u][T] = mfma_bf16_16x16x32(kf[T], qlo[u], sT[u][T]);
            dpT[u][T] = mfma_bf16_16x16x32(vf[T], dolo[u], dpT[u][T]);
          }
#pragma unroll
        for (int u = 0; u < QT; ++u)
#pragma unroll
          for (int T = 0; T < 2; ++T) sT[u][T] = mfma_bf16_16x16x32(k3[T], q3[u], sT[u][T]);
#pragma unroll
        for (int u = 0; u < QT; ++u) {
          unsigned int keep = 0xFFu;
          if (DROP) keep = drop_keep8(dkey, (uint32_t)(c * (BKC / 8) + hf * 4 + g), (uint32_t)(qbase + u * 16 + li), (uint32_t)bh, drop_site, drop_thr);
          const f32x2_t nl2 = {nl[u], nl[u]};
          float ds[8];
#pragma unroll
          for (int T = 0; T < 2; ++T) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              const f32x2_t arg = __builtin_elementwise_fma((f32x2_t){sT[u][T][2 * pr], sT[u][T][2 * pr + 1]}, c2, nl2);
              const f32x2_t p2 = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
              f32x2_t d2;
              if (DROP) {
                const int j = T * 4 + 2 * pr;
                const float m0 = ((keep >> j) & 1u) ? drop_scale : 0.f, m1 = ((keep >> (j + 1)) & 1u) ? drop_scale : 0.f;
                d2 = p2 * (f32x2_t){__builtin_fmaf(m0, dpT[u][T][2 * pr], nd[u]), __builtin_fmaf(m1, dpT[u][T][2 * pr + 1], nd[u])};
              } else {
                d2 = p2 * (f32x2_t){dpT[u][T][2 * pr], dpT[u][T][2 * pr + 1]};
              }
              ds[T * 4 + 2 * pr] = d2.x;
              ds[T * 4 + 2 * pr + 1] = d2.y;
            }
          }
          s16x8 dhi, dlo;
          split8(ds, dhi, dlo);
          f32x4& acc = hf ? acc1[u] : acc0[u];
          acc = mfma_bf16_16x16x32(kth, dhi, acc);
          acc = mfma_bf16_16x16x32(kth, dlo, acc);
          acc = mfma_bf16_16x16x32(ktl, dhi, acc);
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
    for (int r = 0; r < 4; ++r) acc[r] = acc0[u][r] + acc1[u][r];
    *reinterpret_cast<f32x4*>(&dQp[row * HDP + g * 4]) = acc;
  }
}

// ------------------------------------------------------------------------------------------------ dK, dV
template <int KT>
struct DkvStage {
  s16x8 q, o, qt, ot, q2;
  float lse, d;
  unsigned char mb[2 * KT];
};

// KT: 16-key tiles per wave; a workgroup owns 64 * KT keys of one (sample, head) and walks the queries in 64-row chunks
// staged through LDS, so every staged chunk (22 KB of stores) and every fragment read serves KT tiles (see the forward).
// DROP: a lane of this kernel holds ONE key and 8 consecutive queries, the transpose of the (query, 8-key block) unit
// the Philox counters are defined on; the workgroup therefore generates the 64 query x (64 KT) key keep tile of a chunk
// cooperatively (512 KT calls, 2 KT per thread, during staging) into LDS as maskS[key block][query] bytes and every lane
// reads its 8 queries' bytes with one ds_read_b64.
template <bool DROP, int KT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_bf16_kernel(
    const unsigned short* __restrict__ Qs, const unsigned short* __restrict__ Qt,
    const unsigned short* __restrict__ Ks, const unsigned short* __restrict__ Vs,
    const unsigned char* __restrict__ kmask, const unsigned short* __restrict__ dOs,
    const unsigned short* __restrict__ dOt, const float* __restrict__ LSE, const float* __restrict__ D,
    float* __restrict__ dK, float* __restrict__ dV, int B, int H, int Lq, int Lqp, int S, int Sp,
    const unsigned long long* __restrict__ drop_state, unsigned int drop_site, unsigned int drop_thr, float drop_scale) {
  __shared__ __attribute__((aligned(16))) unsigned char maskS[2][DROP ? 8 * KT * BKC : 16];
  __shared__ __attribute__((aligned(16))) unsigned short Qsm[2][BKC * 32];    // [q_hi | q_lo]  rows tile
  __shared__ __attribute__((aligned(16))) unsigned short Q3sm[2][BKC * 32];   // [q_hi | q_lo2] rows tile
  __shared__ __attribute__((aligned(16))) unsigned short Osm[2][BKC * 32];    // [dO_hi | dO_lo] rows tile
  __shared__ __attribute__((aligned(16))) unsigned short Qtm[2][4 * 16 * 32]; // planes [plane][32-query half][16][32]
  __shared__ __attribute__((aligned(16))) unsigned short Otm[2][4 * 16 * 32];
  __shared__ __attribute__((aligned(16))) float lseS[2][BKC];
  __shared__ __attribute__((aligned(16))) float dS_[2][BKC];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  constexpr int KW = 64 * KT;                          // keys per workgroup
  int group, within;
  if (!xcd_decode((Sp + KW - 1) / KW, B * H, group, within)) return;
  const int b = group / H, h = group - b * H;
  const size_t bh = (size_t)b * H + h;

  s16x8 khi[KT], klo[KT], k3[KT], vhi[KT], vlo[KT];
  f32x4 bias4[KT];
  int key[KT];
#pragma unroll
  for (int u = 0; u < KT; ++u) {
    key[u] = within * KW + (wave * KT + u) * 16 + li;
    khi[u] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    klo[u] = khi[u]; k3[u] = khi[u]; vhi[u] = khi[u]; vlo[u] = khi[u];
    bool valid = false;
    if (key[u] < Sp) {
      const unsigned short* kp = Ks + (bh * Sp + key[u]) * QKW;
      const unsigned short* vp = Vs + (bh * Sp + key[u]) * VRW;
      khi[u] = *reinterpret_cast<const s16x8*>(kp + (g & 1) * 8);
      klo[u] = *reinterpret_cast<const s16x8*>(kp + 16 + (g & 1) * 8);
      k3[u] = *reinterpret_cast<const s16x8*>(kp + ((g < 2) ? 32 + g * 8 : (g & 1) * 8));   // [k_lo2 | k_hi] vs A = [q_hi | q_lo2]
      vhi[u] = *reinterpret_cast<const s16x8*>(vp + (g & 1) * 8);
      vlo[u] = *reinterpret_cast<const s16x8*>(vp + 16 + (g & 1) * 8);
      valid = key[u] < S;
      if (valid && kmask) valid = kmask[(size_t)b * S + key[u]] == 0;
    }
    const float bias_k = valid ? 0.f : -INFINITY;
    bias4[u] = f32x4{bias_k, bias_k, bias_k, bias_k};
  }
  const f32x2_t c2 = {LOG2E_F, LOG2E_F};

  const int qrow = t >> 2, qseg = t & 3;
  const int pplane = t >> 7, pd = (t >> 3) & 15, pseg = t & 7;
  DropKey dkey = {0u, 0u};
  if (DROP) dkey = drop_key(drop_state);
  auto stage_load = [&](int c) {
    DkvStage<KT> st;
    if (DROP) {
#pragma unroll
      for (int i = 0; i < 2 * KT; ++i)
        st.mb[i] = (unsigned char)drop_keep8(dkey, (uint32_t)(within * (8 * KT) + (t >> 6) + 4 * i), (uint32_t)(c * BKC + (t & 63)),
                                             (uint32_t)bh, drop_site, drop_thr);
    }
    st.q = *reinterpret_cast<const s16x8*>(Qs + (bh * Lqp + (size_t)c * BKC + qrow) * QKW + qseg * 8);
    if (t < 2 * BKC) st.q2 = *reinterpret_cast<const s16x8*>(Qs + (bh * Lqp + (size_t)c * BKC + (t >> 1)) * QKW + 32 + (t & 1) * 8);
    st.o = *reinterpret_cast<const s16x8*>(dOs + (bh * Lqp + (size_t)c * BKC + qrow) * VRW + qseg * 8);
    const size_t pb = ((bh * 2 + pplane) * 16 + pd) * Lqp + (size_t)c * BKC + pseg * 8;
    st.qt = *reinterpret_cast<const s16x8*>(Qt + pb);
    st.ot = *reinterpret_cast<const s16x8*>(dOt + pb);
    st.lse = INFINITY;
    st.d = 0.f;
    if (t < BKC) {
      const int qq = c * BKC + t;
      if (qq < Lq) {
        float l = LSE[bh * Lqp + qq];
        st.lse = (l == -INFINITY) ? INFINITY : l;
        st.d = D[bh * Lqp + qq];
      }
    }
    return st;
  };
  auto stage_store = [&](const DkvStage<KT>& st, int buf) {
    *reinterpret_cast<s16x8*>(&Qsm[buf][tile_off(qrow, qseg)]) = st.q;
    if (qseg < 2) *reinterpret_cast<s16x8*>(&Q3sm[buf][tile_off(qrow, qseg)]) = st.q;
    if (t < 2 * BKC) *reinterpret_cast<s16x8*>(&Q3sm[buf][tile_off(t >> 1, 2 + (t & 1))]) = st.q2;
    *reinterpret_cast<s16x8*>(&Osm[buf][tile_off(qrow, qseg)]) = st.o;
    const int po = ((pplane * 2 + (pseg >> 2)) * 16) * 32 + plane_off(pd, pseg & 3);
    *reinterpret_cast<s16x8*>(&Qtm[buf][po]) = st.qt;
    *reinterpret_cast<s16x8*>(&Otm[buf][po]) = st.ot;
    if (t < BKC) { lseS[buf][t] = -st.lse * LOG2E_F; dS_[buf][t] = -st.d; }   // -lse log2 e (masked row: -inf), -D
    if (DROP) {
#pragma unroll
      for (int i = 0; i < 2 * KT; ++i) maskS[buf][((t >> 6) + 4 * i) * BKC + (t & 63)] = st.mb[i];
    }
  };

  int qoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) qoff[j] = tile_off((j >> 1) * 32 + (li >> 2) * 8 + (li & 3) + (j & 1) * 4, g);
  const int poff = plane_off(li, g);

  f32x4 dk0[KT], dk1[KT], dv0[KT], dv1[KT];
#pragma unroll
  for (int u = 0; u < KT; ++u) { dk0[u] = f32x4{0.f, 0.f, 0.f, 0.f}; dk1[u] = dk0[u]; dv0[u] = dk0[u]; dv1[u] = dk0[u]; }
  const int nch = Lqp / BKC;
  stage_store(stage_load(0), 0);
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    const int buf = c & 1;
    DkvStage<KT> nxt;
    const bool has_next = (c + 1 < nch);
    if (has_next) nxt = stage_load(c + 1);
    // per 32-query half: fragment reads first (shared by the KT key tiles); the key mask (0 / -inf) and -D enter as MFMA
    // accumulator inits
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      s16x8 qf[2], q3[2], of[2];
      f32x4 nl4[2], nd[2];
#pragma unroll
      for (int T = 0; T < 2; ++T) {
        qf[T] = *reinterpret_cast<const s16x8*>(&Qsm[buf][qoff[hf * 2 + T]]);
        q3[T] = *reinterpret_cast<const s16x8*>(&Q3sm[buf][qoff[hf * 2 + T]]);
        of[T] = *reinterpret_cast<const s16x8*>(&Osm[buf][qoff[hf * 2 + T]]);
        nl4[T] = *reinterpret_cast<const f32x4*>(&lseS[buf][hf * 32 + g * 8 + T * 4]);
        nd[T] = *reinterpret_cast<const f32x4*>(&dS_[buf][hf * 32 + g * 8 + T * 4]);
      }
      const s16x8 oth = *reinterpret_cast<const s16x8*>(&Otm[buf][((0 * 2 + hf) * 16) * 32 + poff]);
      const s16x8 otl = *reinterpret_cast<const s16x8*>(&Otm[buf][((1 * 2 + hf) * 16) * 32 + poff]);
      const s16x8 qth = *reinterpret_cast<const s16x8*>(&Qtm[buf][((0 * 2 + hf) * 16) * 32 + poff]);
      const s16x8 qtl = *reinterpret_cast<const s16x8*>(&Qtm[buf][((1 * 2 + hf) * 16) * 32 + poff]);
      f32x4 s[KT][2], dp[KT][2];
#pragma unroll
      for (int u = 0; u < KT; ++u)
#pragma unroll
        for (int T = 0; T < 2; ++T) {
          s[u][T] = mfma_bf16_16x16x32(qf[T], khi[u], bias4[u]);
          dp[u][T] = mfma_bf16_16x16x32(of[T], vhi[u], DROP ? f32x4{0.f, 0.f, 0.f, 0.f} : nd[T]);
        }
#pragma unroll
      for (int u = 0; u < KT; ++u)
#pragma unroll
        for (int T = 0; T < 2; ++T) {
          s[u][T] = mfma_bf16_16x16x32(qf[T], klo[u], s[u][T]);
          dp[u][T] = mfma_bf16_16x16x32(of[T], vlo[u], dp[u][T]);
        }
#pragma unroll
      for (int u = 0; u < KT; ++u)
#pragma unroll
        for (int T = 0; T < 2; ++T) s[u][T] = mfma_bf16_16x16x32(q3[T], k3[u], s[u][T]);
#pragma unroll
      for (int u = 0; u < KT; ++u) {
        unsigned long long keep8 = 0;      // byte j: keep flags of query hf * 32 + g * 8 + j for this tile's key block
        if (DROP) keep8 = *reinterpret_cast<const unsigned long long*>(&maskS[buf][((wave * KT + u) * 2 + (li >> 3)) * BKC + hf * 32 + g * 8]);
        float p8[8], ds8[8];
#pragma unroll
        for (int T = 0; T < 2; ++T) {
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            // P = exp2(fma(s, log2 e, -lse log2 e)); dS = P * (dP - D)
            const f32x2_t arg = __builtin_elementwise_fma((f32x2_t){s[u][T][2 * pr], s[u][T][2 * pr + 1]}, c2,
                                                          (f32x2_t){nl4[T][2 * pr], nl4[T][2 * pr + 1]});
            f32x2_t p2 = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
            f32x2_t d2;
            if (DROP) {
              const int j = T * 4 + 2 * pr;
              const float m0 = ((keep8 >> (8 * j + (li & 7))) & 1ull) ? drop_scale : 0.f;
              const float m1 = ((keep8 >> (8 * (j + 1) + (li & 7))) & 1ull) ? drop_scale : 0.f;
              d2 = p2 * (f32x2_t){__builtin_fmaf(m0, dp[u][T][2 * pr], nd[T][2 * pr]), __builtin_fmaf(m1, dp[u][T][2 * pr + 1], nd[T][2 * pr + 1])};
              p2 = p2 * (f32x2_t){m0, m1};             // dV sees the dropped weights
            } else {
              d2 = p2 * (f32x2_t){dp[u][T][2 * pr], dp[u][T][2 * pr + 1]};
            }
            p8[T * 4 + 2 * pr] = p2.x;
            p8[T * 4 + 2 * pr + 1] = p2.y;
            ds8[T * 4 + 2 * pr] = d2.x;
            ds8[T * 4 + 2 * pr + 1] = d2.y;
          }
        }
        s16x8 phi, plo, dhi, dlo;
        split8(p8, phi, plo);
        split8(ds8, dhi, dlo);
        f32x4& dv = hf ? dv1[u] : dv0[u];
        f32x4& dk = hf ? dk1[u] : dk0[u];
        dv = mfma_bf16_16x16x32(oth, phi, dv);
        dk = mfma_bf16_16x16x32(qth, dhi, dk);
        dv = mfma_bf16_16x16x32(oth, plo, dv);
        dk = mfma_bf16_16x16x32(qth, dlo, dk);
        dv = mfma_bf16_16x16x32(otl, phi, dv);
        dk = mfma_bf16_16x16x32(qtl, dhi, dk);
      }
    }
    if (has_next) stage_store(nxt, buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < KT; ++u) {
    if (key[u] >= Sp) continue;
    f32x4 dk, dv;
#pragma unroll
    for (int r = 0; r < 4; ++r) { dk[r] = dk0[u][r] + dk1[u][r]; dv[r] = dv0[u][r] + dv1[u][r]; }
    *reinterpret_cast<f32x4*>(&dK[(bh * Sp + key[u]) * HDP + g * 4]) = dk;
    *reinterpret_cast<f32x4*>(&dV[(bh * Sp + key[u]) * HDP + g * 4]) = dv;
  }
}

}  // namespace a3d

using namespace a3d;

static int attn_bwd_bf16_launch(const void* Qs, const void* Qt, const void* Ks, const void* Kt, const void* Vs,
                                const unsigned char* kmask, const float* O, const float* dO, const float* LSE,
                                void* dOs, void* dOt, float* D, float* dQp, float* dK, float* dV, int B, int H,
                                int Lq, int Lqp, int S, int Sp, int nsplit, const unsigned long long* drop_state,
                                unsigned int drop_site, float drop_p, void* stream) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lqp < Lq || (Lqp % 64) != 0 || S <= 0 || Sp < S || (Sp % 64) != 0 || nsplit < 1 ||
      nsplit > 64) {
    set_error("a3d_attn_bwd_bf16: bad argument (B=%d H=%d Lq=%d Lqp=%d S=%d Sp=%d nsplit=%d; Lqp, Sp %% 64 == 0)", B, H, Lq,
              Lqp, S, Sp, nsplit);
    return A3D_ERR_ARG;
  }
  if (!Qs || !Qt || !Ks || !Kt || !Vs || !O || !dO || !LSE || !dOs || !dOt || !D || !dQp || !dK || !dV) {
    set_error("a3d_attn_bwd_bf16: null pointer");
    return A3D_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  const int E = H * HD;
  const size_t lds = (size_t)2 * 64 * (E + 1) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_prep_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_bwd_prep_bf16_kernel, dim3(Lqp / 64, B), dim3(256), lds, s, dO, O, (unsigned short*)dOs,
                     (unsigned short*)dOt, D, B, H, Lq, Lqp);
  int rc = check_launch("a3d_attn_bwd_bf16(prep)");
  if (rc) return rc;
  const bool drop = drop_state != nullptr && drop_p > 0.f;
  if (drop_state && !(drop_p >= 0.f && drop_p < 1.f)) {
    set_error("a3d_attn_bwd_bf16_dropout: dropout probability %g outside [0, 1)", (double)drop_p);
    return A3D_ERR_ARG;
  }
  const unsigned int thr = drop ? (unsigned int)lrintf(drop_p * 65536.0f) : 0u;
  const float dscale = drop ? 1.0f / (1.0f - drop_p) : 1.0f;
  // two query tiles per wave for dQ once the query set fills them; two key tiles per wave for dK / dV when there are
  // enough 128-key workgroups left to fill the chip (A3D_ATTN_QT = 1 / 2 forces either form of both)
  static const int qt_env = getenv("A3D_ATTN_QT") ? atoi(getenv("A3D_ATTN_QT")) : 0;
  const int QT = qt_env ? qt_env : (Lq > 64 ? 2 : 1);
  const int KT = qt_env ? qt_env : (((size_t)B * H * (Sp / 128) >= 1024 && Lq > 16) ? 2 : 1);
  const dim3 gq(xcd_grid(B * H, cdiv(Lqp, 64 * QT) * nsplit)), gk(xcd_grid(B * H, cdiv(Sp, 64 * KT)));
  const unsigned long long* nostate = nullptr;
#define A3D_LAUNCH_DQ(DROPV, QTV, ST, SITE, THR, SC)                                                                        \
  hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<DROPV, QTV>), gq, dim3(256), 0, s, (const unsigned short*)Qs,                \
                     (const unsigned short*)Ks, (const unsigned short*)Kt, (const unsigned short*)Vs, kmask,               \
                     (const unsigned short*)dOs, LSE, D, dQp, B, H, Lq, Lqp, S, Sp, nsplit, ST, SITE, THR, SC)
#define A3D_LAUNCH_DKV(DROPV, KTV, ST, SITE, THR, SC)                                                                       \
  hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<DROPV, KTV>), gk, dim3(256), 0, s, (const unsigned short*)Qs,               \
                     (const unsigned short*)Qt, (const unsigned short*)Ks, (const unsigned short*)Vs, kmask,               \
                     (const unsigned short*)dOs, (const unsigned short*)dOt, LSE, D, dK, dV, B, H, Lq, Lqp, S, Sp, ST,     \
                     SITE, THR, SC)
  if (drop) {
    if (QT == 2) A3D_LAUNCH_DQ(true, 2, drop_state, drop_site, thr, dscale);
    else A3D_LAUNCH_DQ(true, 1, drop_state, drop_site, thr, dscale);
  } else {
    if (QT == 2) A3D_LAUNCH_DQ(false, 2, nostate, 0u, 0u, 1.0f);
    else A3D_LAUNCH_DQ(false, 1, nostate, 0u, 0u, 1.0f);
  }
  rc = check_launch("a3d_attn_bwd_bf16(dq)");
  if (rc) return rc;
  if (drop) {
    if (KT == 2) A3D_LAUNCH_DKV(true, 2, drop_state, drop_site, thr, dscale);
    else A3D_LAUNCH_DKV(true, 1, drop_state, drop_site, thr, dscale);
  } else {
    if (KT == 2) A3D_LAUNCH_DKV(false, 2, nostate, 0u, 0u, 1.0f);
    else A3D_LAUNCH_DKV(false, 1, nostate, 0u, 0u, 1.0f);
  }
#undef A3D_LAUNCH_DQ
#undef A3D_LAUNCH_DKV
  return check_launch("a3d_attn_bwd_bf16(dkv)");
}

extern "C" int a3d_attn_bwd_bf16(const void* Qs, const void* Qt, const void* Ks, const void* Kt, const void* Vs,
                                 const unsigned char* kmask, const float* O, const float* dO, const float* LSE,
                                 void* dOs, void* dOt, float* D, float* dQp, float* dK, float* dV, int B, int H,
                                 int Lq, int Lqp, int S, int Sp, int nsplit, void* stream) {
  return attn_bwd_bf16_launch(Qs, Qt, Ks, Kt, Vs, kmask, O, dO, LSE, dOs, dOt, D, dQp, dK, dV, B, H, Lq, Lqp, S, Sp, nsplit,
                              nullptr, 0u, 0.f, stream);
}

extern "C" int a3d_attn_bwd_bf16_dropout(const void* Qs, const void* Qt, const void* Ks, const void* Kt, const void* Vs,
                                         const unsigned char* kmask, const float* O, const float* dO, const float* LSE,
                                         void* dOs, void* dOt, float* D, float* dQp, float* dK, float* dV, int B, int H,
                                         int Lq, int Lqp, int S, int Sp, int nsplit, const unsigned long long* drop_state,
                                         unsigned int drop_site, float drop_p, void* stream) {
  if (!drop_state) { set_error("a3d_attn_bwd_bf16_dropout: null dropout state"); return A3D_ERR_ARG; }
  return attn_bwd_bf16_launch(Qs, Qt, Ks, Kt, Vs, kmask, O, dO, LSE, dOs, dOt, D, dQp, dK, dV, B, H, Lq, Lqp, S, Sp, nsplit,
                              drop_state, drop_site, drop_p, stream);
}
