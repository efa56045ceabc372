// Ghost-point <-> scene cross-attention and the diffusion transformer's attention core for gfx950.
//
// Reference semantics (multihead_custom_attention.py:355-447): per head h (d = 15), A = softmax(q_h k_h^T
// + key_padding_mask), o_h = A v_h, heads concatenated.  q arrives already scaled by d^-1/2 and rotated
// (rope.hip).  The reference materialises the (B*H, Lq, S) score tensor three times; this file never does.
//
// Forward  (attn_fwd_kernel): flash-style online softmax on bf16 MFMA tiles (v_mfma_f32_16x16x32_bf16).
//   Operands are split bf16 x = hi + lo (+ lo2), so the K = 32 contraction of one MFMA carries two 16-wide parts
//   of the 16-padded head dim.  Scores use three parts per operand and three MFMAs ([k_hi|k_lo].[q_hi|q_hi] +
//   [k_hi|k_lo].[q_lo|q_lo] + [k_hi|k_lo2].[q_lo2|q_hi]): fp32-grade logits, because exp() turns an absolute
//   score error into a relative weight error and 2^-17 |q||k| (two parts) is ~1e-3 at |s| ~ 100 -- the parity
//   bar against the fp32 reference.  A single plain-bf16 MFMA is off by 1e-1 there.
//   The score tile is computed transposed (S^T = K Q^T) with the key rows of the two 16x16 tiles of a
//   32-key half interleaved (tile T row i <-> key (i>>2)*8 + (i&3) + 4T) so that after exp() every lane
//   already holds, in order, the 8 consecutive keys the P operand of the PV MFMA wants: P never touches LDS.
//   PV = V_hi P_hi + V_hi P_lo + V_lo P_hi (3 MFMAs per 32 keys).
// Backward: attention_bwd.hip (split-bf16 MFMA on the same operand formats; the exact-f32 MFMA backward that preceded it was
//   deleted in round 4 -- one A/B reference per kernel).
#include "a3d_common.h"
#include "../../include/act3d_hip.h"
#include <stdlib.h>

namespace a3d {

constexpr int KC = 64;          // keys per staged chunk
constexpr int FLD = 20;         // padded fp32 row stride (80 B) for the backward's [64][16] tiles

struct FwdStage {
  s16x8 k, v, k2;
  float bias;
};

// __launch_bounds__(256, 2): <= 256 VGPRs, which lets the compiler keep MFMA results in VGPRs (no v_accvgpr_read copies)
// QT: 16-query tiles per wave.  A workgroup covers 64 * QT queries of one (sample, head); every K / V fragment a wave
// reads from LDS (16 ds_read_b128 per 64-key chunk) and every chunk the workgroup stages (14.5 KB of ds_write) feeds QT
// times as many MFMAs.  With QT = 1 the kernel is LDS-bound: per 64-key chunk and CU (8 waves) ~510 LDS-array cycles of
// fragment reads + ~400 of staging stores against 18 MFMAs x 17 cycles x 2 waves = 610 per SIMD; QT = 2 halves the LDS
// side and doubles the independent MFMA chains per wave.  QT = 1 stays for short query sets (Lq <= 64: the query stream
// of Act3D, the 16-step trajectories of the denoiser), where a second tile would be padding.
// DROP: training-mode dropout of the attention weights (multihead_custom_attention.py:413): A = softmax(..) is
// normalised by the UN-dropped row sum, then every weight is kept with probability 1 - p and scaled by 1 / (1 - p).  The
// keep flags of a lane's 8 consecutive keys come from one Philox call (a3d_common.h); the denominator is a per-lane f32
// sum (the ones-channel of V would see the dropped weights).
template <bool DROP, int QT>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(
    const unsigned short* __restrict__ Qs, const unsigned short* __restrict__ Ks,
    const unsigned short* __restrict__ Vt, const unsigned char* __restrict__ kmask,
    float* __restrict__ O, float* __restrict__ LSE, float* __restrict__ Op, float* __restrict__ Mp,
    float* __restrict__ Lp, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit,
    const unsigned long long* __restrict__ drop_state, unsigned int drop_site, unsigned int drop_thr, float drop_scale) {
  __shared__ __attribute__((aligned(16))) unsigned short Ksm[2][KC * 32];    // [k_hi | k_lo]  rows tile
  __shared__ __attribute__((aligned(16))) unsigned short K3sm[2][KC * 32];   // [k_hi | k_lo2] rows tile
  __shared__ __attribute__((aligned(16))) unsigned short Vsm[2][4 * 16 * 32];   // [plane][32-key half][16 ch][32 keys]
  __shared__ __attribute__((aligned(16))) float biasS[2][KC];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  constexpr int QW = 64 * QT;                       // queries per workgroup
  const int tiles_x = (Lqp + QW - 1) / QW;
  int group, within;
  if (!xcd_decode(tiles_x * nsplit, B * H, group, within)) return;
  const int b = group / H, h = group - b * H;
  const int sp = within / tiles_x;
  const int E = H * HD;
  const int qbase = (within - sp * tiles_x) * QW + wave * (16 * QT);
  const size_t bh = (size_t)b * H + h;

  // B operands of the three score MFMAs: [q_hi | q_hi], [q_lo | q_lo], [q_lo2 | q_hi]  (against A = [k_hi | k_lo],
  // [k_hi | k_lo], [k_hi | k_lo2]): every product of (k_hi + k_lo + k_lo2)(q_hi + q_lo + q_lo2) above 2^-24
  s16x8 qhi[QT], qlo[QT], q3[QT];
  bool active[QT];
  bool any_active = false;
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const int q0 = qbase + u * 16;
    active[u] = q0 < Lqp;
    any_active = any_active || active[u];
    qhi[u] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    qlo[u] = qhi[u];
    q3[u] = qhi[u];
    if (active[u]) {
      const unsigned short* qp = Qs + (bh * Lqp + q0 + li) * QKW;
      qhi[u] = *reinterpret_cast<const s16x8*>(qp + (g & 1) * 8);
      qlo[u] = *reinterpret_cast<const s16x8*>(qp + 16 + (g & 1) * 8);
      q3[u] = *reinterpret_cast<const s16x8*>(qp + ((g < 2) ? 32 + g * 8 : (g & 1) * 8));   // g < 2: q_lo2, else q_hi
    }
  }

  const int nch = Sp / KC;
  const int cps = (nch + nsplit - 1) / nsplit;
  const int c_beg = sp * cps;
  const int c_end = min(nch, c_beg + cps);

  // staging roles
  const int krow = t >> 2, kseg = t & 3;
  const int vplane = t >> 7, vd = (t >> 3) & 15, vseg = t & 7;
  auto stage_load = [&](int c) {
    FwdStage st;
    st.k = *reinterpret_cast<const s16x8*>(Ks + (bh * Sp + (size_t)c * KC + krow) * QKW + kseg * 8);
    if (t < 2 * KC) st.k2 = *reinterpret_cast<const s16x8*>(Ks + (bh * Sp + (size_t)c * KC + (t >> 1)) * QKW + 32 + (t & 1) * 8);
    st.v = *reinterpret_cast<const s16x8*>(Vt + ((bh * 2 + vplane) * 16 + vd) * Sp + (size_t)c * KC + vseg * 8);
    st.bias = 0.f;
    if (t < KC) {
      const int key = c * KC + t;
      bool valid = key < S;
      if (valid && kmask) valid = kmask[(size_t)b * S + key] == 0;
      st.bias = valid ? 0.f : -INFINITY;
    }
    return st;
  };
  auto stage_store = [&](const FwdStage& st, int buf) {
    *reinterpret_cast<s16x8*>(&Ksm[buf][tile_off(krow, kseg)]) = st.k;
    if (kseg < 2) *reinterpret_cast<s16x8*>(&K3sm[buf][tile_off(krow, kseg)]) = st.k;           // k_hi
    if (t < 2 * KC) *reinterpret_cast<s16x8*>(&K3sm[buf][tile_off(t >> 1, 2 + (t & 1))]) = st.k2;   // k_lo2
    // padded channel 15 of V_hi := 1.0, so that acc[d = 15] accumulates the softmax denominator sum_k (p_hi + p_lo)
    // on the MFMA pipe, from exactly the rounded P the numerator uses (no VALU row sums, self-consistent weights)
    const s16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    *reinterpret_cast<s16x8*>(&Vsm[buf][((vplane * 2 + (vseg >> 2)) * 16) * 32 + plane_off(vd, vseg & 3)]) =
        (vplane == 0 && vd == 15) ? ones : st.v;
    if (t < KC) biasS[buf][t] = st.bias;
  };

  // per-lane fragment offsets (constant over the key loop): score tile j = hf * 2 + T covers keys
  // hf * 32 + (i >> 2) * 8 + (i & 3) + 4 T, i = 0..15 (the interleave that leaves P in B-operand order)
  int koff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) koff[j] = tile_off((j >> 1) * 32 + (li >> 2) * 8 + (li & 3) + (j & 1) * 4, g);
  const int voff = plane_off(li, g);

  float m_run[QT], l_run[QT];                       // l_run (DROP): this lane's share of sum_k p (un-dropped)
  f32x4 acc0[QT], acc1[QT];                         // one accumulator per 32-key half: two PV chains per query tile
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    m_run[u] = -INFINITY;
    l_run[u] = 0.f;
    acc0[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1[u] = acc0[u];
  }
  DropKey dkey = {0u, 0u};
  if (DROP) dkey = drop_key(drop_state);
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

  if (c_beg < c_end) {
    stage_store(stage_load(c_beg), 0);
    __syncthreads();
  }
  for (int c = c_beg; c < c_end; ++c) {
    const int buf = (c - c_beg) & 1;
    FwdStage nxt;
    const bool has_next = (c + 1 < c_end);
    if (has_next) nxt = stage_load(c + 1);

    if (any_active) {
      // ---- all LDS fragment reads of the chunk up front (shared by the wave's QT query tiles)
      s16x8 kf[4], k3[4], vh[2], vl[2];
      f32x4 bias4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        kf[j] = *reinterpret_cast<const s16x8*>(&Ksm[buf][koff[j]]);
        k3[j] = *reinterpret_cast<const s16x8*>(&K3sm[buf][koff[j]]);
        bias4[j] = *reinterpret_cast<const f32x4*>(&biasS[buf][(j >> 1) * 32 + g * 8 + (j & 1) * 4]);
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        vh[hf] = *reinterpret_cast<const s16x8*>(&Vsm[buf][((0 * 2 + hf) * 16) * 32 + voff]);
        vl[hf] = *reinterpret_cast<const s16x8*>(&Vsm[buf][((1 * 2 + hf) * 16) * 32 + voff]);
      }
      // ---- scores of every tile: 4 * QT independent MFMA chains
      f32x4 s[QT][4];
#pragma unroll
      for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[u][j] = mfma_bf16_16x16x32(kf[j], qhi[u], bias4[j]);
#pragma unroll
      for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[u][j] = mfma_bf16_16x16x32(kf[j], qlo[u], s[u][j]);
#pragma unroll
      for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[u][j] = mfma_bf16_16x16x32(k3[j], q3[u], s[u][j]);

#pragma unroll
      for (int u = 0; u < QT; ++u) {
        // running max: a depth-3 tree over the lane's 16 scores, then across the column's four lanes
        float mt[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) mt[j] = fmaxf(fmaxf(s[u][j][0], s[u][j][1]), fmaxf(s[u][j][2], s[u][j][3]));
        const float mx = colmax4(fmaxf(fmaxf(mt[0], mt[1]), fmaxf(mt[2], mt[3])));
        const float m_new = fmaxf(m_run[u], mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        // p = exp(s - m) as exp2(fma(s, log2 e, -m log2 e)): one packed FMA per two scores + v_exp_f32
        const float nm = -m_use * LOG2E_F;
        const float alpha = __builtin_amdgcn_exp2f(__builtin_fmaf(m_run[u], LOG2E_F, nm));   // m_run = -inf -> 0
        const f32x2 c2 = {LOG2E_F, LOG2E_F}, nm2 = {nm, nm};
        s16x8 phi[2], plo[2];
        float l_tile = 0.f;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          unsigned int hw[4], lw[4];
          unsigned int keep = 0xFFu;
          if (DROP) keep = drop_keep8(dkey, (uint32_t)(c * (KC / 8) + hf * 4 + g), (uint32_t)(qbase + u * 16 + li), (uint32_t)bh, drop_site, drop_thr);
#pragma unroll
          for (int T = 0; T < 2; ++T) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              const f32x4& sj = s[u][hf * 2 + T];
              const f32x2 arg = __builtin_elementwise_fma((f32x2){sj[2 * pr], sj[2 * pr + 1]}, c2, nm2);
              f32x2 p2 = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
              if (DROP) {
                l_tile += p2.x + p2.y;
                const int j = T * 4 + 2 * pr;          // this lane's keys of the half: g * 8 + j, g * 8 + j + 1
                p2.x = ((keep >> j) & 1u) ? p2.x * drop_scale : 0.f;
                p2.y = ((keep >> (j + 1)) & 1u) ? p2.y * drop_scale : 0.f;
              }
              // x = hi + lo, both halves rounded to nearest-even; the compiler lowers the conversion to
              // v_cvt_pk_bf16_f32 and tracks its hazards (a hand-written asm statement is opaque to it)
              const unsigned int h2 = __builtin_bit_cast(unsigned int, __builtin_convertvector(p2, bf16x2));
              const f32x2 r2 = p2 - (f32x2){__uint_as_float(h2 << 16), __uint_as_float(h2 & 0xFFFF0000u)};
              hw[T * 2 + pr] = h2;
              lw[T * 2 + pr] = __builtin_bit_cast(unsigned int, __builtin_convertvector(r2, bf16x2));
            }
          }
          phi[hf] = __builtin_bit_cast(s16x8, (u32x4){hw[0], hw[1], hw[2], hw[3]});
          plo[hf] = __builtin_bit_cast(s16x8, (u32x4){lw[0], lw[1], lw[2], lw[3]});
        }
        m_run[u] = m_new;
        if (DROP) l_run[u] = l_run[u] * alpha + l_tile;
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc0[u][r] *= alpha; acc1[u][r] *= alpha; }
        acc0[u] = mfma_bf16_16x16x32(vh[0], phi[0], acc0[u]);
        acc1[u] = mfma_bf16_16x16x32(vh[1], phi[1], acc1[u]);
        acc0[u] = mfma_bf16_16x16x32(vh[0], plo[0], acc0[u]);
        acc1[u] = mfma_bf16_16x16x32(vh[1], plo[1], acc1[u]);
        acc0[u] = mfma_bf16_16x16x32(vl[0], phi[0], acc0[u]);
        acc1[u] = mfma_bf16_16x16x32(vl[1], phi[1], acc1[u]);
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
      l_tot += __shfl_xor(l_tot, 32, 64);              // the four lane groups hold disjoint key subsets of column li
    } else {
      l_tot = __shfl(acc[3], 48 + li, 64);             // channel 15 (lane group g = 3, register 3) holds sum_k p
    }
    const int q = qbase + u * 16 + li;
    if (nsplit == 1) {
      const float inv = (l_tot > 0.f) ? 1.0f / l_tot : 0.f;
      if (q < Lq) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = g * 4 + r;
          if (d < HD) O[((size_t)b * Lq + q) * E + h * HD + d] = acc[r] * inv;
        }
      }
      if (g == 0) LSE[bh * Lqp + q] = (l_tot > 0.f) ? (m_run[u] + logf(l_tot)) : -INFINITY;
    } else {
      const size_t row = (((size_t)sp * B + b) * H + h) * Lqp + q;
      *reinterpret_cast<f32x4*>(&Op[row * HDP + g * 4]) = acc;
      if (g == 0) { Mp[row] = m_run[u]; Lp[row] = l_tot; }
    }
  }
}

__global__ __launch_bounds__(256) void attn_combine_kernel(
    const float* __restrict__ Op, const float* __restrict__ Mp, const float* __restrict__ Lp,
    float* __restrict__ O, float* __restrict__ LSE, int B, int H, int Lq, int Lqp, int nsplit) {
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
      const float w = __expf(Mp[s * rows + row] - m_use);
      l += Lp[s * rows + row] * w;
      o += Op[(s * rows + row) * HDP + d] * w;
    }
    if (d < HD && q < Lq) O[((size_t)b * Lq + q) * E + h * HD + d] = (l > 0.f) ? o / l : 0.f;
    if (d == 0) LSE[row] = (l > 0.f) ? (m + logf(l)) : -INFINITY;
  }
}

}  // namespace a3d

using namespace a3d;

static int check_attn_args(const char* fn, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lqp < Lq || (Lqp % 16) != 0 || S <= 0 || Sp < S || (Sp % KC) != 0 ||
      nsplit < 1 || nsplit > 64) {
    set_error("%s: bad argument (B=%d H=%d Lq=%d Lqp=%d S=%d Sp=%d nsplit=%d)", fn, B, H, Lq, Lqp, S, Sp, nsplit);
    return A3D_ERR_ARG;
  }
  return A3D_OK;
}

static int attn_fwd_launch(const void* Qs, const void* Ks, const void* Vt, const unsigned char* kmask,
                           float* O, float* LSE, float* ws, int B, int H, int Lq, int Lqp, int S, int Sp,
                           int nsplit, const unsigned long long* drop_state, unsigned int drop_site, float drop_p,
                           void* stream) {
  int rc = check_attn_args("a3d_attn_fwd", B, H, Lq, Lqp, S, Sp, nsplit);
  if (rc) return rc;
  if (!Qs || !Ks || !Vt || !O || !LSE || (nsplit > 1 && !ws)) {
    set_error("a3d_attn_fwd: null pointer");
    return A3D_ERR_ARG;
  }
  const bool drop = drop_state != nullptr && drop_p > 0.f;
  if (drop_state && !(drop_p >= 0.f && drop_p < 1.f)) {
    set_error("a3d_attn_fwd_dropout: dropout probability %g outside [0, 1)", (double)drop_p);
    return A3D_ERR_ARG;
  }
  const unsigned int thr = drop ? (unsigned int)lrintf(drop_p * 65536.0f) : 0u;
  const float dscale = drop ? 1.0f / (1.0f - drop_p) : 1.0f;
  hipStream_t s = (hipStream_t)stream;
  const size_t rows = (size_t)B * H * Lqp;
  float* Op = ws;
  float* Mp = ws ? ws + (size_t)nsplit * rows * HDP : nullptr;
  float* Lp = ws ? Mp + (size_t)nsplit * rows : nullptr;
  // two query tiles per wave (128 queries per workgroup) once the query set fills them; A3D_ATTN_QT=1 forces the narrow form
  static const int qt_env = getenv("A3D_ATTN_QT") ? atoi(getenv("A3D_ATTN_QT")) : 0;
  const int QT = qt_env ? qt_env : (Lq > 64 ? 2 : 1);
  dim3 grid(xcd_grid(B * H, cdiv(Lqp, 64 * QT) * nsplit));
  const unsigned long long* nostate = nullptr;
#define A3D_LAUNCH_FWD(DROPV, QTV, ST, SITE, THR, SC)                                                                    \
  hipLaunchKernelGGL((attn_fwd_kernel<DROPV, QTV>), grid, dim3(256), 0, s, (const unsigned short*)Qs,                   \
                     (const unsigned short*)Ks, (const unsigned short*)Vt, kmask, O, LSE, Op, Mp, Lp, B, H, Lq, Lqp, S, \
                     Sp, nsplit, ST, SITE, THR, SC)
  if (drop) {
    if (QT == 2) A3D_LAUNCH_FWD(true, 2, drop_state, drop_site, thr, dscale);
    else A3D_LAUNCH_FWD(true, 1, drop_state, drop_site, thr, dscale);
  } else {
    if (QT == 2) A3D_LAUNCH_FWD(false, 2, nostate, 0u, 0u, 1.0f);
    else A3D_LAUNCH_FWD(false, 1, nostate, 0u, 0u, 1.0f);
  }
#undef A3D_LAUNCH_FWD
  rc = check_launch("a3d_attn_fwd");
  if (rc) return rc;
  if (nsplit > 1) {
    const int cg = (int)std::min<size_t>((rows * HDP + 255) / 256, 4096);
    hipLaunchKernelGGL(attn_combine_kernel, dim3(cg), dim3(256), 0, s, Op, Mp, Lp, O, LSE, B, H, Lq, Lqp, nsplit);
    rc = check_launch("a3d_attn_fwd(combine)");
  }
  return rc;
}

extern "C" int a3d_attn_fwd(const void* Qs, const void* Ks, const void* Vt, const unsigned char* kmask,
                            float* O, float* LSE, float* ws, int B, int H, int Lq, int Lqp, int S, int Sp,
                            int nsplit, void* stream) {
  return attn_fwd_launch(Qs, Ks, Vt, kmask, O, LSE, ws, B, H, Lq, Lqp, S, Sp, nsplit, nullptr, 0u, 0.f, stream);
}

extern "C" int a3d_attn_fwd_dropout(const void* Qs, const void* Ks, const void* Vt, const unsigned char* kmask,
                                    float* O, float* LSE, float* ws, int B, int H, int Lq, int Lqp, int S, int Sp,
                                    int nsplit, const unsigned long long* drop_state, unsigned int drop_site,
                                    float drop_p, void* stream) {
  if (!drop_state) { set_error("a3d_attn_fwd_dropout: null dropout state"); return A3D_ERR_ARG; }
  return attn_fwd_launch(Qs, Ks, Vt, kmask, O, LSE, ws, B, H, Lq, Lqp, S, Sp, nsplit, drop_state, drop_site, drop_p, stream);
}

extern "C" size_t a3d_attn_fwd_ws_floats(int B, int H, int Lqp, int nsplit) {
  if (nsplit <= 1) return 0;
  return (size_t)nsplit * B * H * Lqp * (HDP + 2);
}
