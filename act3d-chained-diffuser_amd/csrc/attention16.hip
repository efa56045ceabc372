// Split-fp16 attention for gfx950: the ghost-point <-> scene cross-attention (and the diffusion transformer's attention
// core) on v_mfma_f32_16x16x32_f16 / _bf16.  Default family of the package (ops.ATTN_MODE = "f16"); the three-part bf16
// kernels in attention.hip / attention_bwd.hip stay as the A/B reference (A3D_ATTN_MODE=bf16x3).
//
// Reference semantics (multihead_custom_attention.py:355-447): per head h (d = 15), A = softmax(q_h k_h^T + mask),
// o_h = A v_h.  What bounds this op on MI355X is NOT the matrix pipe: with d = 15 a score costs 60 algorithmic FLOPs
// but one v_exp_f32 (6.5 cycles per wave64 op; a packed 16-bit conversion costs the same, profiles/r03_inst_rate.txt), so
// the inner loops are written to minimise VECTOR instructions per score:
//   * q, k are TWO-part fp16 (x = hi + lo, 22 mantissa bits; fp16 subnormals are honoured by the MFMA, same file):
//     logits are fp32-grade from two K = 32 MFMAs per 16x16 tile, [k_hi|k_lo].[q_hi|q_hi] + [k_hi|k_lo].[q_lo|q_lo]
//     (three with three-part bf16).  exp() turns an ABSOLUTE logit error into a relative weight error.
//   * log2(e) is folded into q by the projection kernel, -m (forward) / -lse (backward) / -D ride in as MFMA accumulator
//     inits: exp2 is applied DIRECTLY to MFMA results -- no per-score argument arithmetic at all.
//   * lazy rescaling: the running max is only revised when a score exceeds it by 2^8 (a wave-uniform, rarely taken
//     branch), so the common path has no cross-lane traffic and no accumulator rescale.
//   * the softmax denominator is accumulated on the MFMA from the SAME rounded P (ones-channel of V, written by the
//     projection kernel).
//   * dO rows are normalised by a power of two per (b, h, q) row (exact), so that G = P (dP - D) fits the 16-bit formats
//     whatever the loss scale.
// What is NOT single 16-bit, and why (every item was first built single-part, measured, and widened; DESIGN.md section 4
// holds the numbers): P in the forward is two-part fp16 (single: 1.27e-3 of scale on the gain-3 Act3D fixture, bar 1e-3);
// G in dQ is two-part (single breaks sum_k G = 0: key-bias gradients off by 3-7e-3); V is two-part (a rounded V makes
// D = dO . O inconsistent with dP = dO . V, a difference of nearly equal numbers when the softmax is sharp); dO is two-part;
// and the contractions over the QUERY axis (dK = G^T Q, dV = P^T dO) run in split bf16, not fp16: keys with tiny weights
// (2^-30 and below, but thousands of them feed one context row's gradient) fall out of fp16's range whatever the offset.
// A3D_ATTN_FAST=1 selects the single-part P / G variants (fwd 0.104 vs 0.127 ms) for users who accept 2.5e-4-class errors.
// Per 64 keys x 16 queries a wave issues 14 (fwd) / 22 (dQ) / 28 (dK,dV) MFMAs against 18 / 26 / 32 in the bf16 family,
// and roughly half its vector instructions.
//
// Staging.  Tiles arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass) into a ring of NB
// buffers with NB - 1 chunks in flight, ONE raw s_barrier per chunk and counted vmcnt waits.  An LDS-DMA lands
// lane-linearly (wave base + lane * 16 B), so the bank swizzle of the tiles (a3d_common.h tile_off / plane_off) is applied
// on the SOURCE address: lane l fetches the 16-byte segment that belongs at position l.  (Measured: the ring removed the
// staging VGPRs and ds_writes but did not change the kernel time -- the kernels are issue-bound on VALU + MFMA, not
// latency-bound; it is kept because the staging registers and the ds_write pass are gone.)
//
// Backward of the query axis ("packs").  attn16_bwd_prep sorts the queries of one (b, h) by the exponent of their dO row
// (descending), and writes per 64 sorted queries one 20 KB LDS image: fp16 two-part Q rows and dOn rows for the score /
// dP MFMAs, bf16 hi/lo Q^T and dOn^T planes for the dK / dV MFMAs, -LSE, -D, the permutation and the chunk exponent.
// attn16_bwd_dkv carries its accumulators in units of 2^E_chunk (a power-of-two rescale per chunk, monotone because of the
// sort; rows 2^60 below the largest are dropped) -- block floating point over the query axis, so that dO of any dynamic
// range (padded queries, masked losses) keeps 16 significant bits per row.
//
// Operand formats ("16" formats, written by a3d_proj_rope_split16 / a3d_rope_split16 / attn16_bwd_prep):
//   rows16   [B][H][Npad][32] fp16 : hi(16) | lo(16) of the 16-padded head row        (q, k, v)
//   planes16 [B][H][2][16][Npad] fp16 : hi and lo planes, transposed (8 consecutive rows of one channel = one MFMA A
//            fragment); the value planes carry 1.0 in the padded channel 15 of the hi plane (softmax denominator)
//   dOr      [B][H][Lqp][32] fp16 : hi | lo of the row-normalised dO * ln 2;  pack: see PK_* below
// Scores live in log2 units (q carries log2 e): LSE2 = log2 sum_k 2^s2.
#include "attn_ring.h"
#include "../../include/act3d_hip.h"
#include <stdlib.h>

namespace a3d {

// ------------------------------------------------------------------------------------------------ forward
// QT 16-query tiles per wave (128 queries per workgroup at QT = 2).  Scores are computed transposed (S^T = K Q^T) with
// the key rows of the two 16x16 tiles of a 32-key half interleaved (tile T row i <-> key (i >> 2) * 8 + (i & 3) + 4 T):
// after exp2 a lane holds, in order, the 8 consecutive keys the P operand of the PV MFMA wants, so P never touches LDS.
// DROP: training-mode dropout of the attention weights (multihead_custom_attention.py:413), Philox keep flags as in
// attention.hip; the denominator is then a per-lane f32 sum of the un-dropped weights.
// PP: parts of P in the PV product.  2 (default): p = hi + lo, both fp16 -- the output error of a single-fp16 P is
// 2^-12 |v_k - o| per dominant key, which the gain-3 reference fixtures (|logit| ~ 100, one or two keys per query) turn into
// 1.3e-3 of the mask-logit scale two layers later: over the 1e-3 parity bar.  1: single fp16 P (A3D_ATTN_FAST=1), 2-4e-4 per
// attention output, 25 % fewer vector instructions.
// PP = 2 (round 6, gradient-free passes only -- see the launch code for why a pass with a backward keeps PP = 3) is ADAPTIVE: the low part of P (8 v_fma_mix + 2 MFMAs per 16 queries x 64 keys, a quarter of the loop's
// vector work) is formed only for chunks that hold a DOMINANT key of some query of the tile -- a weight above 2^-LO_SPAN of the
// query's running denominator (wave-uniform test on the chunk maximum the lazy rescale computes anyway; the denominator is
// re-read from the accumulators every 8th chunk, in between it is stale = smaller = conservative).  Why that is enough: the
// output error of a single-fp16 P is sum_k w_k eps_k (v_k - o) with |eps_k| <= 2^-11, w_k the normalised weights; keys with
// w_k <= tau contribute at most 2^-11.8 sqrt(tau) |v - o|_max in the root-mean-square sense (3.5e-5 at tau = 2^-6) however many
// there are, and what the round-3 measurement attributed the 1.27e-3 model-level error to are the one or two keys per query that
// carry most of the weight on the gain-3 fixtures -- those keep both parts.  PP = 3: both parts everywhere (the round-5 kernel).
constexpr float LO_SPAN = 6.0f;
constexpr int A3D_ATTN16_V_ROWS = 1, A3D_ATTN16_NOGRAD = 2;
constexpr int FWD_NB = 4;
// VR: the values arrive as ROWS ([B][H][Sp][32], channel 15 of the hi part = 1.0) and the V^T fragments come from transposed LDS reads
// (attn_ring.h tr_frag) -- the rows-only operand set of round 6: the projection kernel writes ONE layout of K and V; !VR: value planes.
template <bool DROP, int QT, int PP, bool VR>
__global__ __launch_bounds__(256, 2) void attn16_fwd_kernel(
    const unsigned short* __restrict__ Qr, const unsigned short* __restrict__ Kr, const unsigned short* __restrict__ Vp,
    const unsigned char* __restrict__ kmask, float* __restrict__ O, float* __restrict__ LSE2, float* __restrict__ Op,
    float* __restrict__ Mp, float* __restrict__ Lp, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit,
    const unsigned long long* __restrict__ drop_state, unsigned int drop_site, unsigned int drop_thr, float drop_scale) {
  __shared__ __attribute__((aligned(16))) unsigned short Ksm[FWD_NB][C16 * 32];      // [k_hi | k_lo] rows tile
  __shared__ __attribute__((aligned(16))) unsigned short Vsm[FWD_NB][4 * 16 * 32];   // [plane hi/lo][32-key half][16 ch][32 keys]
  __shared__ unsigned int maskW[MASKW];

  const int t = threadIdx.x, lane = t & 63;
  // the wave index is uniform per wave, but threadIdx-derived values are formally divergent: without the readfirstlane every
  // branch on it (padding-only waves, active tiles) and the chunk loop itself were compiled as exec-masked vector control flow
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 15, g = lane >> 4;
  constexpr int QW = 64 * QT;
  const int tiles_x = (Lqp + QW - 1) / QW;
  int group, within;
  if (!xcd_decode(tiles_x * nsplit, B * H, group, within)) return;
  group = __builtin_amdgcn_readfirstlane(group);             // (integer divisions run on the vector unit: back to scalars)
  within = __builtin_amdgcn_readfirstlane(within);
  const int b = __builtin_amdgcn_readfirstlane(group / H), h = group - b * H;
  const int sp = __builtin_amdgcn_readfirstlane(within / tiles_x);
  const int E = H * HD;
  const int qbase = (within - sp * tiles_x) * QW + wave * (16 * QT);
  const size_t bh = (size_t)b * H + h;
  const bool any_masked = (kmask != nullptr) || (Sp != S);
  if (any_masked) {
    build_key_mask(maskW, kmask, b, S, Sp);
    __syncthreads();
  }

  s16x8 qhi[QT], qlo[QT];
  bool active[QT];
  bool any_active = false;
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const int q0 = qbase + u * 16;
    active[u] = q0 < Lq;                                     // the tile holds at least one real query
    any_active = any_active || active[u];
    qhi[u] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    qlo[u] = qhi[u];
    if (active[u]) {
      const unsigned short* qp = Qr + (bh * Lqp + q0 + li) * 32;
      qhi[u] = *reinterpret_cast<const s16x8*>(qp + (g & 1) * 8);
      qlo[u] = *reinterpret_cast<const s16x8*>(qp + 16 + (g & 1) * 8);
    }
  }
#pragma unroll
  for (int u = 0; u < QT; ++u) { A3D_PIN(qhi[u]); A3D_PIN(qlo[u]); }
  const int nch = Sp / C16;
  const int cps = __builtin_amdgcn_readfirstlane((nch + nsplit - 1) / nsplit);
  const int c_beg = sp * cps;
  const int c_end = min(nch, c_beg + cps);

  // 2 LDS-DMA instructions per wave and chunk: its 16 rows of the K tile, its (plane, half) sub-tile of V
  const unsigned short* Kbase = Kr + bh * Sp * 32;
  const unsigned short* Vbase = VR ? Vp + bh * Sp * 32 : Vp + ((bh * 2 + (wave >> 1)) * 16) * Sp + (wave & 1) * 32;
  auto issue = [&](int c, int slot) {
    const int cc = min(c, c_end - 1);                        // past the end: a harmless re-fetch keeps the vmcnt count fixed
    dma_rows_tile(Kbase + (size_t)cc * C16 * 32, 32, Ksm[slot], wave, lane);
    if (VR) dma_rows_tile(Vbase + (size_t)cc * C16 * 32, 32, Vsm[slot], wave, lane);
    else dma_plane_subtile(Vbase, Sp, (size_t)cc * C16, &Vsm[slot][wave * 512], lane);
  };

  // LDS fragment addresses: ONE base per tile and slot, everything else an immediate of the ds_read.  Score tile j of a chunk reads
  // row (j >> 1) * 32 + (li >> 2) * 8 + (li & 3) + (j & 1) * 4: the swizzle key (0 - (row >> 3)) & 3 does not depend on j, so
  // koff[j] = koff0 + FWD_KJ(j); likewise tr_off(.., hf, rr) = tr_off(.., 0, 0) + (hf * 32 + rr * 4) * 32 (row >> 3 = hf * 4 + g).
  // (Round 5 kept 12 offset registers and paid one v_lshl_or per read and chunk to add the runtime slot base: 12 of ~130 VALU.)
#define FWD_KJ(j) ((((j) >> 1) * 32 + ((j) & 1) * 4) * 32)
  const int koff0 = tile_off((li >> 2) * 8 + (li & 3), g);
  const int voff = plane_off(li, g);
  const int vtr0 = tr_off(li, g, 0, 0, 0), vtr1 = tr_off(li, g, 1, 0, 0);       // VR: hi / lo part of the transposed value reads

  float m_run[QT], l_run[QT];            // running max (log2 units, exact per query column); l_run: DROP only
  float thr[QT];                         // PP == 2: a chunk maximum above it (same units as the score tiles) asks for the low part of P
  f32x4 cin[QT];                         // MFMA accumulator init of the score tiles: P_OFF - m_run
  f32x4 acc0[QT], acc1[QT];
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    m_run[u] = 0.f;
    l_run[u] = 0.f;
    thr[u] = -INFINITY;
    cin[u] = f32x4{P_OFF, P_OFF, P_OFF, P_OFF};
    acc0[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1[u] = acc0[u];
  }
  DropKey dkey = {0u, 0u};
  if (DROP) { dkey = drop_key(drop_state); A3D_PIN(dkey.k0); A3D_PIN(dkey.k1); }

  if (c_beg < c_end) {
#pragma unroll
    for (int i = 0; i < FWD_NB - 1; ++i) issue(c_beg + i, i);
  }
  for (int c = c_beg; c < c_end; ++c) {
    const int slot = (c - c_beg) % FWD_NB;
    wait_vm<2 * (FWD_NB - 2)>();                              // this wave's pieces of chunk c have landed ...
    ring_barrier();                                           // ... everybody's have, and everybody is done with chunk c - 1
    issue(c + FWD_NB - 1, (slot + FWD_NB - 1) % FWD_NB);      // into the buffer chunk c - 1 just vacated
    const bool first = (c == c_beg);
    const bool masked = any_masked && ((kmask != nullptr) || ((c + 1) * C16 > S));     // wave-uniform
    if (!any_active) continue;                                // a wave of padding queries only feeds the ring

    s16x8 kf[4], vh[2], vl[2];
    {
      const unsigned short* kb = &Ksm[slot][koff0];
#pragma unroll
      for (int j = 0; j < 4; ++j) kf[j] = *reinterpret_cast<const s16x8*>(kb + FWD_KJ(j));
      if (VR) {
        const unsigned short *vb0 = &Vsm[slot][vtr0], *vb1 = &Vsm[slot][vtr1];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          vh[hf] = tr_frag(vb0 + hf * 1024, 0, 128);
          vl[hf] = tr_frag(vb1 + hf * 1024, 0, 128);
        }
      } else {
        const unsigned short* vb = &Vsm[slot][voff];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          vh[hf] = *reinterpret_cast<const s16x8*>(vb + ((0 * 2 + hf) * 16) * 32);
          vl[hf] = *reinterpret_cast<const s16x8*>(vb + ((1 * 2 + hf) * 16) * 32);
        }
      }
    }
    // Order of one chunk: score MFMAs of all tiles, then per tile { rescale? | exp + pack | PV }.  Round 6 also built the tile-pipelined
    // order (scores(0) | rescale(0)? | scores(1) beside exp(0) | rescale(1)? | PV(0) beside exp(1) | PV(1); -DA3D_ATTN_FWD_TILE_PIPE):
    // 128.1 vs 125.7 us per launch at the bench shape (profiles/r06_attn_ab.txt) -- on this SIMD vector work next to an MFMA is close
    // to ADDITIVE (profiles/r06_coissue.txt: 1 MFMA + 8 v_exp = 31.2 ns against 27.2 + 8.1), so re-pairing matrix and vector work of
    // different tiles buys nothing and the longer live ranges cost.  A sched_group_barrier interleave on top of it changed the
    // results of this kernel (different digests of O on identical inputs) and was dropped.
    f32x4 s[QT][4];
    s16x8 pf[QT][2];
    float pv[QT][2][8];                    // the fp32 weights, kept for the low part
    bool lo_part[QT];
    auto scores = [&](const int u) {
      if (masked) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s[u][j] = mfma_f16(kf[j], qhi[u], cin[u] + bias_of(maskW[c * 2 + (j >> 1)], g, j & 1));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) s[u][j] = mfma_f16(kf[j], qhi[u], cin[u]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) s[u][j] = mfma_f16(kf[j], qlo[u], s[u][j]);
    };
    auto rescale = [&](const int u) {
      // s = s2 - m_run + P_OFF.  Common path: nothing exceeds 2^(P_OFF + P_THR) -> exponentiate as is; PP == 2: and nothing is
      // within 2^-LO_SPAN of the running denominator (thr <= P_OFF + P_THR: ONE wave-uniform test guards both rare paths)
      const float mx = max16(s[u][0], s[u][1], s[u][2], s[u][3]);
      lo_part[u] = (PP == 3);
      if (first || __builtin_amdgcn_ballot_w64(mx > (PP == 2 ? thr[u] : P_OFF + P_THR)) != 0ull) {
        if (PP == 2) lo_part[u] = true;
        if (PP != 2 || first || __builtin_amdgcn_ballot_w64(mx > P_OFF + P_THR) != 0ull) {
          const float cm = colmax4(mx);                                       // exact chunk max of the lane's query
          float shift = first ? (cm - P_OFF) : fmaxf(cm - P_OFF, 0.f);
          if (cm == -INFINITY) shift = 0.f;                                   // every key so far masked
          m_run[u] += shift;
          if (PP == 2) thr[u] -= shift;
          const float alpha = __builtin_amdgcn_exp2f(-shift);
#pragma unroll
          for (int r = 0; r < 4; ++r) { acc0[u][r] *= alpha; acc1[u][r] *= alpha; cin[u][r] -= shift; }
          if (DROP) l_run[u] *= alpha;
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[u][j][r] -= shift;
        }
      }
    };
    auto weights = [&](const int u) {
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
            if (PP >= 2) { pv[u][hf][T * 4 + 2 * pr] = p0; pv[u][hf][T * 4 + 2 * pr + 1] = p1; }
          }
        }
        pf[u][hf] = __builtin_bit_cast(s16x8, (u32x4_){w[0], w[1], w[2], w[3]});
      }
      if (DROP) l_run[u] += l_tile;
    };
    auto values = [&](const int u) {
      // V keeps both parts: O = sum_k p~_k v_k / sum_k p~_k is an exactly normalised average of the 22-bit v rows, so
      // what is left of the P rounding is proportional to the spread of v under the weights, not to |v|
      acc0[u] = mfma_f16(vh[0], pf[u][0], acc0[u]);
      acc1[u] = mfma_f16(vh[1], pf[u][1], acc1[u]);
      acc0[u] = mfma_f16(vl[0], pf[u][0], acc0[u]);
      acc1[u] = mfma_f16(vl[1], pf[u][1], acc1[u]);
      if (PP >= 2 && lo_part[u]) {                               // wave-uniform
        s16x8 pl[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          unsigned int wl[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) wl[i] = lo_f16(pv[u][hf][2 * i], pv[u][hf][2 * i + 1], (unsigned int)__builtin_bit_cast(u32x4_, pf[u][hf])[i]);
          pl[hf] = __builtin_bit_cast(s16x8, (u32x4_){wl[0], wl[1], wl[2], wl[3]});
        }
        acc0[u] = mfma_f16(vh[0], pl[0], acc0[u]);
        acc1[u] = mfma_f16(vh[1], pl[1], acc1[u]);
      }
      if (PP == 2 && ((c - c_beg) & 7) == 0) {
        // refresh the dominance threshold from the running denominator: sum_k p sits in channel 15 of the accumulators (lane
        // group 3, register 3); with dropout the per-lane partial sums (a lower bound: conservative).  log2 units like the scores.
        const float lsum = DROP ? l_run[u] : ((g == 3) ? acc0[u][3] + acc1[u][3] : 0.f);
        const float lcol = colmax4(lsum);
        thr[u] = fminf(__builtin_amdgcn_logf(lcol) - LO_SPAN, P_OFF + P_THR);
      }
    };
#ifndef A3D_ATTN_FWD_TILE_PIPE
#pragma unroll
    for (int u = 0; u < QT; ++u) scores(u);
#pragma unroll
    for (int u = 0; u < QT; ++u) { rescale(u); weights(u); values(u); }
#else
    scores(0);
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      rescale(u);
      if (u + 1 < QT) scores(u + 1);
      if (u > 0) values(u - 1);
      weights(u);
      A3D_PIN(pf[u][0]); A3D_PIN(pf[u][1]);                  // (otherwise hipcc sinks the exponentials past the next rescale branch, next to their PV)
    }
    values(QT - 1);
#endif
  }
#undef FWD_KJ
  wait_vm<0>();                                               // the trailing dummy fetches must not outlive the workgroup

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
// One workgroup per (b, h).  Per query row q:  e_q = exponent with max_d |dO ln2| / 2^e in [0.5, 1);  dOn = dO ln2 2^-e as
// two-part fp16 (hi | lo rows, for the dQ kernel);  D = sum_d dOn * O from the ROUNDED dOn (the backward is the exact
// derivative for the upstream gradient dOn 2^e / ln2);  rexp = e (-100 for an all-zero row).
//
// The dK / dV kernel contracts over QUERIES, i.e. it mixes rows of different scale inside one MFMA, and fp16 has 5 exponent
// bits: with every row normalised against the largest row of the (b, h), rows 2^-20 below it vanish -- and they can be the
// only rows a parameter's gradient lives on (measured: 20 % error on the vision-language attention of a reference fixture
// whose offset loss puts 1e6-scale gradients on the ghost rows).  So the rows are SORTED by e_q (descending) and the dK / dV
// kernel consumes them in that order, 64 per chunk, each chunk normalised against its own largest row (E_c), its
// accumulators carried in units of 2^E_c: block floating point over the query axis.  Rows more than 2^60 below the largest
// row of the (b, h) are dropped.  The same contraction also needs the RANGE of the weights themselves: a key that a sharp
// softmax gives 1e-10 still receives a gradient (1e-10 of the others'), and it can be all a parameter ever sees (the
// vision-language attention above); fp16 ends at 6e-8.  P' and G' therefore enter the dV / dK MFMAs as split bf16 (hi + lo,
// 16 mantissa bits, fp32's exponent range) against bf16 hi / lo planes of dOn and Q, while scores and dP stay fp16.
// This kernel writes, per chunk, the LDS images the dK / dV kernel needs ("pack", 20 KB):
//   [0, 4 KB) Q rows tile (fp16 hi | lo)  [4, 8) dOn rows tile (fp16 hi | lo)  [8, 12) Q planes bf16 hi / lo
//   [12, 16) dOn planes bf16 hi / lo  [16 KB ..) nl[64] = B_OFF - lse2 + e_q - E_c, nd[64] = -Dn, perm[64] = original row
//   (dropout counters), E_c
constexpr int PK_HALFS = 10240;                // 20 KB per chunk
constexpr int PK_QROWS = 0, PK_OROWS = 2048, PK_QPL = 4096, PK_OPL = 6144, PK_NL = 8192, PK_ND = 8320, PK_PERM = 8448, PK_HDR = 8576;
constexpr int DROP_SPAN = 60;                  // rows more than 2^DROP_SPAN below the largest row are dropped

constexpr int PREP_GROUP = 8;                 // chunks (of 64 sorted rows) staged per barrier round

__global__ __launch_bounds__(256) void attn16_bwd_prep_kernel(
    const float* __restrict__ dO, const float* __restrict__ O, const float* __restrict__ LSE2,
    const unsigned short* __restrict__ Qr, unsigned short* __restrict__ dOr, float* __restrict__ D,
    int* __restrict__ rexp, unsigned short* __restrict__ pack, int B, int H, int Lq, int Lqp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned int* keyS = reinterpret_cast<unsigned int*>(smem_raw);                             // [Lqp] (300 - e) << 16 | row
  unsigned int* sortS = keyS + Lqp;                                                           // [Lqp] the same, ascending
  float* dS = reinterpret_cast<float*>(sortS + Lqp);                                          // [Lqp] -Dn
  unsigned short* qrowS = reinterpret_cast<unsigned short*>(dS + Lqp);                        // [<= 512][32] gathered Q rows
  unsigned short* orowS = qrowS + PREP_GROUP * 64 * 32;                                       // [<= 512][32] gathered dOn rows
  const int t = threadIdx.x;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int E = H * HD;
  const size_t bh = (size_t)blockIdx.x;

  // ---- A: per-row exponent, normalised two-part fp16 row, D
  for (int q = t; q < Lqp; q += 256) {
    int e = -100;
    unsigned int w[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, wl[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    float dsum = 0.f;
    bool live = q < Lq;
    if (live) {
      const float* gp = dO + ((size_t)b * Lq + q) * E + h * HD;
      const float* op = O + ((size_t)b * Lq + q) * E + h * HD;
      float v[16];
      float mx = 0.f;
#pragma unroll
      for (int d = 0; d < 16; ++d) {
        v[d] = (d < HD) ? gp[d] * LN2_F : 0.f;
        mx = fmaxf(mx, fabsf(v[d]));
      }
      if (mx > 0.f && mx < INFINITY) (void)frexpf(mx, &e);
      if (e < -100) e = -100;
      const float inv = ldexpf(1.0f, -e);
#pragma unroll
      for (int i = 0; i < 8; ++i) pk_f16_2(v[2 * i] * inv, v[2 * i + 1] * inv, w[i], wl[i]);
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        const int sh = (d & 1) * 16;
        const float rv = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[d >> 1] >> sh)) +
                         (float)__builtin_bit_cast(_Float16, (unsigned short)(wl[d >> 1] >> sh));
        dsum += rv * op[d];
      }
      live = LSE2[bh * Lqp + q] != -INFINITY;
    }
    const size_t row = bh * Lqp + q;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<u32x4_*>(dOr + row * 32 + i * 8) = (u32x4_){w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]};
      *reinterpret_cast<u32x4_*>(dOr + row * 32 + 16 + i * 8) = (u32x4_){wl[4 * i], wl[4 * i + 1], wl[4 * i + 2], wl[4 * i + 3]};
    }
    D[row] = dsum;
    rexp[row] = e;
    dS[q] = -dsum;
    // sort key: descending exponent, then ascending row (deterministic); padding rows and rows without a finite lse last
    keyS[q] = ((live ? (unsigned int)(300 - e) : 0xFFFFu) << 16) | (unsigned int)q;          // e in [-100, 128] -> 172 .. 400
  }
  __syncthreads();
  // ---- B: ascending key order = a STABLE counting sort by the key's high half (230 possible values: 300 - e of a live row, then the
  // dead rows), rows ascending inside a value because the placement walks the rows in order -- the same order as ranking every key
  // by counting the smaller ones, which this replaces: that is Lqp^2 / 4 vector compares on ONE CU (Lqp = 3136, the trajectory
  // model's context rows: 2.5 M per workgroup, ~200 us of the 54 us AVERAGE this kernel shows in the diffusion training trace).
  {
    constexpr int NB = 230;                                      // live: 172 .. 400 -> 0 .. 228; dead (0xFFFF) -> 229
    int* base = reinterpret_cast<int*>(qrowS);                   // [NB] next free position of each value (qrowS is idle until pass C)
    int* wcnt = base + NB;                                       // [4][NB] rows of the current 256-row round, per wave
    auto bucket_of = [](unsigned int key) { return (int)min(key >> 16, 401u) - 172; };
    const int lane = t & 63, wave = t >> 6;
    for (int i = t; i < 5 * NB; i += 256) base[i] = 0;
    __syncthreads();
    for (int q = t; q < Lqp; q += 256) atomicAdd(&base[bucket_of(keyS[q])], 1);
    __syncthreads();
    if (t < 64) {                                                // exclusive scan of the 230 counts: 4 per lane + a wave scan
      int v[4], sum = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = (4 * t + j < NB) ? base[4 * t + j] : 0; sum += v[j]; }
      int incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int nb = __shfl_up(incl, o, 64);
        if (lane >= o) incl += nb;
      }
      int run = incl - sum;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * t + j < NB) { base[4 * t + j] = run; run += v[j]; }
    }
    __syncthreads();
    for (int q0 = 0; q0 < Lqp; q0 += 256) {
      const int q = q0 + t;
      const bool valid = q < Lqp;
      const unsigned int key = valid ? keyS[q] : 0u;
      const int bk = valid ? bucket_of(key) : 255;               // 8 bits; 255 never matches a real value
      unsigned long long same = ~0ull;                            // lanes of this wave holding the same value: 8 ballots
#pragma unroll
      for (int bit = 0; bit < 8; ++bit) {
        const bool on = (bk >> bit) & 1;
        const unsigned long long bl = __ballot(on);
        same &= on ? bl : ~bl;
      }
      const int before = __popcll(same & ((1ull << lane) - 1ull));
      if (valid && before == 0) wcnt[wave * NB + bk] = __popcll(same);
      __syncthreads();
      if (valid) {
        int pos = base[bk] + before;
        for (int w = 0; w < wave; ++w) pos += wcnt[w * NB + bk];
        sortS[pos] = key;
      }
      __syncthreads();
      for (int i = t; i < NB; i += 256) {
        base[i] += wcnt[i] + wcnt[NB + i] + wcnt[2 * NB + i] + wcnt[3 * NB + i];
        wcnt[i] = 0; wcnt[NB + i] = 0; wcnt[2 * NB + i] = 0; wcnt[3 * NB + i] = 0;
      }
      __syncthreads();
    }
  }
  const unsigned int rank0 = sortS[0] >> 16;                                                   // the largest row's 300 - e
  // ---- C: one pack per 64 sorted rows, PREP_GROUP chunks per staging round
  const int nch = Lqp / 64;
  for (int c0 = 0; c0 < nch; c0 += PREP_GROUP) {
    const int ng = min(PREP_GROUP, nch - c0);
    // gathered Q rows of the group, natural layout; the dOn rows are recomputed from dO with the stored exponent (the same
    // operations as pass A, so the same bits) rather than re-read from dOr: reading back another lane's global stores would
    // need a device-scope fence, and a release fence writes back every dirty line of the XCD's L2 (measured: 50 us here)
    for (int idx = t; idx < ng * 256; idx += 256) {
      const int r = idx >> 2, seg = idx & 3;
      const int q = (int)(sortS[c0 * 64 + r] & 0xFFFFu);
      *reinterpret_cast<u32x4_*>(qrowS + r * 32 + seg * 8) = *reinterpret_cast<const u32x4_*>(Qr + (bh * Lqp + q) * 32 + seg * 8);
    }
    for (int r = t; r < ng * 64; r += 256) {
      const unsigned int key = sortS[c0 * 64 + r];
      const int q = (int)(key & 0xFFFFu);
      unsigned int w[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, wl[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
      if ((key >> 16) != 0xFFFFu) {                          // a live row: its exponent is in the sort key (rows without one are dead)
        const float* gp = dO + ((size_t)b * Lq + q) * E + h * HD;
        const float inv = ldexpf(1.0f, (int)(key >> 16) - 300);
        float v[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) v[d] = (d < HD) ? gp[d] * LN2_F : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) pk_f16_2(v[2 * i] * inv, v[2 * i + 1] * inv, w[i], wl[i]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        *reinterpret_cast<u32x4_*>(orowS + r * 32 + i * 8) = (u32x4_){w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]};
        *reinterpret_cast<u32x4_*>(orowS + r * 32 + 16 + i * 8) = (u32x4_){wl[4 * i], wl[4 * i + 1], wl[4 * i + 2], wl[4 * i + 3]};
      }
    }
    __syncthreads();
    for (int cg = 0; cg < ng; ++cg) {
      const int c = c0 + cg;
      unsigned short* pk = pack + (bh * nch + c) * (size_t)PK_HALFS;
      const unsigned short* qg = qrowS + cg * 64 * 32;
      const unsigned short* og = orowS + cg * 64 * 32;
      const unsigned int rank_c = sortS[c * 64] >> 16;
      {
        // Q rows tile and dOn rows tile (LDS images: unit t = row t >> 2, position t & 3 holds segment pos ^ f(row))
        const int r = t >> 2, pos = t & 3;
        const int seg = pos ^ ((0 - (r >> 3)) & 3);
        *reinterpret_cast<u32x4_*>(pk + PK_QROWS + t * 8) = *reinterpret_cast<const u32x4_*>(qg + r * 32 + seg * 8);
        *reinterpret_cast<u32x4_*>(pk + PK_OROWS + t * 8) = *reinterpret_cast<const u32x4_*>(og + r * 32 + seg * 8);
      }
      {
        // Q and dOn planes, bf16 hi / lo of x = hi16 + lo16: sub-tile (part, half) = t >> 6; unit t & 63: channel (t & 63) >> 2,
        // position t & 3
        const int sub = t >> 6, u = t & 63;
        const int part = sub >> 1, half = sub & 1;
        const int ch = u >> 2, pos = u & 3;
        const int seg = pos ^ ((0 - (ch >> 2)) & 3);
        s16x8 oq, oo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int rr = (half * 32 + seg * 8 + j) * 32;
          const float vq = (float)__builtin_bit_cast(_Float16, qg[rr + ch]) + (float)__builtin_bit_cast(_Float16, qg[rr + 16 + ch]);
          const float vo = (float)__builtin_bit_cast(_Float16, og[rr + ch]) + (float)__builtin_bit_cast(_Float16, og[rr + 16 + ch]);
          unsigned short qh, ql, oh, ol;
          split_bf16(vq, qh, ql);
          split_bf16(vo, oh, ol);
          oq[j] = (short)(part ? ql : qh);
          oo[j] = (short)(part ? ol : oh);
        }
        *reinterpret_cast<s16x8*>(pk + PK_QPL + t * 8) = oq;
        *reinterpret_cast<s16x8*>(pk + PK_OPL + t * 8) = oo;
      }
      if (t < 64) {
        const unsigned int key = sortS[c * 64 + t];
        const int q = (int)(key & 0xFFFFu);
        const unsigned int rank = key >> 16;
        float nl = -INFINITY, ndv = 0.f;
        if (rank != 0xFFFFu && rank <= rank0 + DROP_SPAN) {
          nl = B_OFF - LSE2[bh * Lqp + q] - (float)(int)(rank - rank_c);            // e_q - E_c = rank_c - rank
          ndv = dS[q];
        }
        reinterpret_cast<float*>(pk + PK_NL)[t] = nl;
        reinterpret_cast<float*>(pk + PK_ND)[t] = ndv;
        reinterpret_cast<int*>(pk + PK_PERM)[t] = q;
      } else if (t == 64) {
        // E_c; a chunk that starts beyond the drop span (or in the padding) is dead, and so is everything after it
        reinterpret_cast<int*>(pk + PK_HDR)[0] = (rank_c == 0xFFFFu || rank_c > rank0 + DROP_SPAN) ? -1000 : 300 - (int)rank_c;
      } else if (t == 65) {
        // live 32-row halves of the chunk (the rows are sorted: if row 32 is dead, so are rows 33 .. 63)
        const unsigned int rank_h = sortS[c * 64 + 32] >> 16;
        reinterpret_cast<int*>(pk + PK_HDR)[1] = (rank_h == 0xFFFFu || rank_h > rank0 + DROP_SPAN) ? 1 : 2;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
// dQ2[q] = 2^e_q * sum_k G[q, k] K[k],  G = P o (dPn - Dn)  (dPn = V dOn: the ln 2 of d/ds2 rides in dOn)
// DROP: O = (M o P) V  ->  dPn = M o (V dOn), G = P o (dPn - Dn).
// GP: parts of G in dQ = G K.  2 (default): g = hi + lo -- sum_k G[q, k] = 0, so a 2^-12 rounding of G leaves a common-mode
// error |K| sum_k dG that the aggregated gradients (key bias: a sum over 65k rows that is ~0 in exact arithmetic) show as
// 3-7e-3 of their scale; 1: single fp16 G (A3D_ATTN_FAST=1).
constexpr int DQ_NB = 3;
template <bool DROP, int QT, int GP>
__global__ __launch_bounds__(256, 2) void attn16_bwd_dq_kernel(
    const unsigned short* __restrict__ Qr, const unsigned short* __restrict__ Kr,
    const unsigned short* __restrict__ Vr, const unsigned char* __restrict__ kmask, const unsigned short* __restrict__ dOr,
    const float* __restrict__ LSE2, const float* __restrict__ D, const int* __restrict__ rexp, float* __restrict__ dQp,
    int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit, const unsigned long long* __restrict__ drop_state,
    unsigned int drop_site, unsigned int drop_thr, float drop_scale) {
  // (round 6: no K planes -- the K^T fragments of dQ^T = K^T G^T are transposed reads of the K rows tile, attn_ring.h tr_frag)
  __shared__ __attribute__((aligned(16))) unsigned short KVsm[DQ_NB][2][C16 * 32];  // [k_hi | k_lo] rows tile, [v_hi | v_lo] rows tile
  __shared__ unsigned int maskW[MASKW];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // scalar control flow (see the forward kernel)
  const int li = lane & 15, g = lane >> 4;
  constexpr int QW = 64 * QT;
  const int tiles_x = (Lqp + QW - 1) / QW;
  int group, within;
  if (!xcd_decode(tiles_x * nsplit, B * H, group, within)) return;
  group = __builtin_amdgcn_readfirstlane(group);
  within = __builtin_amdgcn_readfirstlane(within);
  const int b = __builtin_amdgcn_readfirstlane(group / H), h = group - b * H;
  const int sp = __builtin_amdgcn_readfirstlane(within / tiles_x);
  const size_t bh = (size_t)b * H + h;
  const int qbase = (within - sp * tiles_x) * QW + wave * (16 * QT);
  const bool any_masked = (kmask != nullptr) || (Sp != S);
  if (any_masked) {
    build_key_mask(maskW, kmask, b, S, Sp);
    __syncthreads();
  }

  s16x8 qhi[QT], qlo[QT], dohi[QT], dolo[QT];
  f32x4 cS[QT], cD[QT];                  // accumulator inits: -lse2 (score tiles), -Dn (dP tiles)
  float nd[QT], rs[QT];
  bool active[QT];
  bool any_active = false;
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const int q = qbase + u * 16 + li;
    active[u] = (qbase + u * 16) < Lq;
    any_active = any_active || active[u];
    qhi[u] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    qlo[u] = qhi[u]; dohi[u] = qhi[u]; dolo[u] = qhi[u];
    float lse_q = INFINITY, d_q = 0.f;
    rs[u] = 0.f;
    if (active[u]) {
      const unsigned short* qp = Qr + (bh * Lqp + q) * 32;
      qhi[u] = *reinterpret_cast<const s16x8*>(qp + (g & 1) * 8);
      qlo[u] = *reinterpret_cast<const s16x8*>(qp + 16 + (g & 1) * 8);
      dohi[u] = *reinterpret_cast<const s16x8*>(dOr + (bh * Lqp + q) * 32 + (g & 1) * 8);          // [dOn_hi | dOn_hi]
      dolo[u] = *reinterpret_cast<const s16x8*>(dOr + (bh * Lqp + q) * 32 + 16 + (g & 1) * 8);     // [dOn_lo | dOn_lo]
      if (q < Lq) {
        lse_q = LSE2[bh * Lqp + q];
        if (lse_q == -INFINITY) lse_q = INFINITY;
        d_q = D[bh * Lqp + q];
        rs[u] = ldexpf(1.0f, rexp[bh * Lqp + q]);
      }
    }
    cS[u] = f32x4{B_OFF - lse_q, B_OFF - lse_q, B_OFF - lse_q, B_OFF - lse_q};
    rs[u] *= 0.015625f;                                     // 2^-B_OFF
    nd[u] = -d_q;
    cD[u] = DROP ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{-d_q, -d_q, -d_q, -d_q};
  }
#pragma unroll
  for (int u = 0; u < QT; ++u) { A3D_PIN(qhi[u]); A3D_PIN(qlo[u]); A3D_PIN(dohi[u]); A3D_PIN(dolo[u]); A3D_PIN(cS[u]); A3D_PIN(cD[u]); A3D_PIN(rs[u]); }
  const int nch = Sp / C16;
  const int cps = __builtin_amdgcn_readfirstlane((nch + nsplit - 1) / nsplit);
  const int c_beg = sp * cps, c_end = min(nch, c_beg + cps);

  // 2 LDS-DMA instructions per wave and chunk: 16 rows of K, 16 rows of V
  const unsigned short* Kbase = Kr + bh * Sp * 32;
  const unsigned short* Vbase = Vr + bh * Sp * 32;
  auto issue = [&](int c, int slot) {
    const int cc = min(c, c_end - 1);
    dma_rows_tile(Kbase + (size_t)cc * C16 * 32, 32, KVsm[slot][0], wave, lane);
    dma_rows_tile(Vbase + (size_t)cc * C16 * 32, 32, KVsm[slot][1], wave, lane);
  };

  // one LDS base per tile and slot, immediates for the rest (see the forward kernel): score tile (hf, T) = koff0 + DQ_KJ(hf * 2 + T),
  // V = K + one tile, transposed K reads = ktr{0,1} + (hf * 32 + rr * 4) * 32
#define DQ_KJ(j) ((((j) >> 1) * 32 + ((j) & 1) * 4) * 32)
  const int koff0 = tile_off((li >> 2) * 8 + (li & 3), g);
  const int ktr0 = tr_off(li, g, 0, 0, 0), ktr1 = tr_off(li, g, 1, 0, 0);

  f32x4 acc0[QT], acc1[QT];
#pragma unroll
  for (int u = 0; u < QT; ++u) { acc0[u] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[u] = acc0[u]; }
  DropKey dkey = {0u, 0u};
  if (DROP) { dkey = drop_key(drop_state); A3D_PIN(dkey.k0); A3D_PIN(dkey.k1); }
  if (c_beg < c_end) {
#pragma unroll
    for (int i = 0; i < DQ_NB - 1; ++i) issue(c_beg + i, i);
  }
  for (int c = c_beg; c < c_end; ++c) {
    const int slot = (c - c_beg) % DQ_NB;
    wait_vm<2 * (DQ_NB - 2)>();
    ring_barrier();
    issue(c + DQ_NB - 1, (slot + DQ_NB - 1) % DQ_NB);
    const bool masked = any_masked && ((kmask != nullptr) || ((c + 1) * C16 > S));
    if (!any_active) continue;
    const unsigned short* kb = &KVsm[slot][0][koff0];
    const unsigned short *tb0 = &KVsm[slot][0][ktr0], *tb1 = &KVsm[slot][0][ktr1];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      // S = 64 n + 1 (4096 scene tokens + the gripper token): the last chunk holds one real key -- its second half is all padding
      if (hf == 1 && c * C16 + 32 >= S) continue;                  // wave-uniform; masked keys contribute exact zeros
      s16x8 kf[2], vf[2], kp[2];
#pragma unroll
      for (int T = 0; T < 2; ++T) {
        kf[T] = *reinterpret_cast<const s16x8*>(kb + DQ_KJ(hf * 2 + T));
        vf[T] = *reinterpret_cast<const s16x8*>(kb + C16 * 32 + DQ_KJ(hf * 2 + T));
      }
      kp[0] = tr_frag(tb0 + hf * 1024, 0, 128);
      kp[1] = tr_frag(tb1 + hf * 1024, 0, 128);
      f32x4 sT[QT][2], dpT[QT][2];
      if (masked) {
        const unsigned int word = maskW[c * 2 + hf];
#pragma unroll
        for (int T = 0; T < 2; ++T) {
          const f32x4 b4 = bias_of(word, g, T);
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
          dpT[u][T] = mfma_f16(vf[T], dohi[u], cD[u]);
          sT[u][T] = mfma_f16(kf[T], qlo[u], sT[u][T]);
        }
#pragma unroll
      for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int T = 0; T < 2; ++T) dpT[u][T] = mfma_f16(vf[T], dolo[u], dpT[u][T]);
#pragma unroll
      for (int u = 0; u < QT; ++u) {
        unsigned int keep = 0xFFu;
        if (DROP) keep = drop_keep8(dkey, (uint32_t)(c * (C16 / 8) + hf * 4 + g), (uint32_t)(qbase + u * 16 + li), (uint32_t)bh, drop_site, drop_thr);
        unsigned int w[4], wl[4];
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
            if (GP == 2) pk_f16_2(g0, g1, w[T * 2 + pr], wl[T * 2 + pr]);
            else w[T * 2 + pr] = pk_f16(g0, g1);
          }
        }
        const s16x8 gf = __builtin_bit_cast(s16x8, (u32x4_){w[0], w[1], w[2], w[3]});
        f32x4& acc = hf ? acc1[u] : acc0[u];
        acc = mfma_f16(kp[0], gf, acc);
        acc = mfma_f16(kp[1], gf, acc);
        if (GP == 2) acc = mfma_f16(kp[0], __builtin_bit_cast(s16x8, (u32x4_){wl[0], wl[1], wl[2], wl[3]}), acc);
      }
    }
  }
  wait_vm<0>();
#undef DQ_KJ
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
// A workgroup owns 64 KT keys of one (b, h) and walks the query rows in the SORTED order of the prep kernel's packs, 64 per
// chunk.  With E_c the largest row exponent of chunk c:
//   P'[q, k] = 2^(B_OFF + s2 - lse2 + e_q - E_c),  G' = P' o (dPn - Dn)
//   dV += 2^E_c sum_{q in c} P'[q, k] dOn[q],   dK += 2^E_c sum_{q in c} G'[q, k] Q2[q]
// the accumulators are kept in units of 2^E_c and rescaled (exactly, by a power of two) when E_c drops.  P' and G' go into
// the dV / dK MFMAs as split bf16 (range: see the prep kernel) in both modes (GP only selects the dQ kernel's variant).
// DROP: a lane of this kernel holds ONE key and 8 consecutive rows, the transpose of the (query, 8-key block) unit the Philox
// counters are defined on; the keep bits of a chunk are generated cooperatively into LDS from the ORIGINAL row indices.
constexpr int DKV_NB = 2;
template <bool DROP, int KT, int GP>
__global__ __launch_bounds__(256, 2) void attn16_bwd_dkv_kernel(
    const unsigned short* __restrict__ pack, const unsigned short* __restrict__ Kr, const unsigned short* __restrict__ Vr,
    const unsigned char* __restrict__ kmask, float* __restrict__ dK, float* __restrict__ dV, int B, int H, int Lqp, int S,
    int Sp, const unsigned long long* __restrict__ drop_state, unsigned int drop_site, unsigned int drop_thr,
    float drop_scale) {
  __shared__ __attribute__((aligned(16))) unsigned short pk[DKV_NB][PK_HALFS];
  __shared__ __attribute__((aligned(16))) unsigned char maskS[2][DROP ? 8 * KT * C16 : 16];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // scalar control flow (see the forward kernel)
  const int li = lane & 15, g = lane >> 4;
  constexpr int KW = 64 * KT;
  int group, within;
  if (!xcd_decode((Sp + KW - 1) / KW, B * H, group, within)) return;
  group = __builtin_amdgcn_readfirstlane(group);
  within = __builtin_amdgcn_readfirstlane(within);
  const int b = __builtin_amdgcn_readfirstlane(group / H), h = group - b * H;
  const size_t bh = (size_t)b * H + h;

  s16x8 khh[KT], kll[KT], vhh[KT], vll[KT];
  f32x4 bias4[KT];
  int key[KT];
#pragma unroll
  for (int u = 0; u < KT; ++u) {
    key[u] = within * KW + (wave * KT + u) * 16 + li;
    khh[u] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
    kll[u] = khh[u]; vhh[u] = khh[u]; vll[u] = khh[u];
    bool valid = false;
    if (key[u] < Sp) {
      const unsigned short* kp = Kr + (bh * Sp + key[u]) * 32;
      const unsigned short* vp = Vr + (bh * Sp + key[u]) * 32;
      khh[u] = *reinterpret_cast<const s16x8*>(kp + (g & 1) * 8);               // B = [k_hi | k_hi]
      kll[u] = *reinterpret_cast<const s16x8*>(kp + 16 + (g & 1) * 8);          // B = [k_lo | k_lo]
      vhh[u] = *reinterpret_cast<const s16x8*>(vp + (g & 1) * 8);               // B = [v_hi | v_hi] against A = [dOn_hi | dOn_lo]
      vll[u] = *reinterpret_cast<const s16x8*>(vp + 16 + (g & 1) * 8);          // B = [v_lo | v_lo]
      valid = key[u] < S;
      if (valid && kmask) valid = kmask[(size_t)b * S + key[u]] == 0;
    }
    const float bias_k = valid ? 0.f : -INFINITY;
    bias4[u] = f32x4{bias_k, bias_k, bias_k, bias_k};
  }
  const bool masked = (kmask != nullptr) || (within * KW + KW > S);           // workgroup-uniform
  // Key tiles that lie entirely beyond S: with S = 64 n + 1 (4096 scene tokens + the gripper token) the last workgroup of every
  // (b, h) holds ONE real key in 128 -- 3 % of this kernel's workgroups used to do full work on -inf scores.  Such tiles (and waves
  // that hold nothing else) keep feeding the ring and the barriers but skip the MFMA / softmax section; their outputs are zeros.
  bool live[KT];
  bool any_live = false;
#pragma unroll
  for (int u = 0; u < KT; ++u) {
    live[u] = within * KW + (wave * KT + u) * 16 < S;
    any_live = any_live || live[u];
  }
  DropKey dkey = {0u, 0u};
  if (DROP) dkey = drop_key(drop_state);
#pragma unroll
  for (int u = 0; u < KT; ++u) { A3D_PIN(khh[u]); A3D_PIN(kll[u]); A3D_PIN(vhh[u]); A3D_PIN(vll[u]); A3D_PIN(bias4[u]); }
  if (DROP) { A3D_PIN(dkey.k0); A3D_PIN(dkey.k1); }

  // 5 LDS-DMA instructions per wave and chunk: the pack is the LDS image, wave w copies its 5 KB quarter
  const int nch = Lqp / C16;
  const unsigned short* pbase = pack + bh * nch * (size_t)PK_HALFS + wave * 2560 + lane * 8;
  auto issue = [&](int c, int slot) {
    const unsigned short* src = pbase + (size_t)min(c, nch - 1) * PK_HALFS;
#pragma unroll
    for (int i = 0; i < 5; ++i) glds16(src + i * 512, &pk[slot][wave * 2560 + i * 512]);
  };

  // one LDS base per tile and slot, immediates for the rest (see the forward kernel)
#define DKV_QJ(j) ((((j) >> 1) * 32 + ((j) & 1) * 4) * 32)
  const int qoff0 = tile_off((li >> 2) * 8 + (li & 3), g);
  const int poff = plane_off(li, g);

  f32x4 dk0[KT], dk1[KT], dv0[KT], dv1[KT];
#pragma unroll
  for (int u = 0; u < KT; ++u) { dk0[u] = f32x4{0.f, 0.f, 0.f, 0.f}; dk1[u] = dk0[u]; dv0[u] = dk0[u]; dv1[u] = dk0[u]; }
  int e_ref = -1000;                                  // exponent unit of the accumulators
#pragma unroll
  for (int i = 0; i < DKV_NB - 1; ++i) issue(i, i);
  for (int c = 0; c < nch; ++c) {
    const int slot = c % DKV_NB;
    wait_vm<5 * (DKV_NB - 2)>();
    ring_barrier();
    issue(c + DKV_NB - 1, (slot + DKV_NB - 1) % DKV_NB);
    const unsigned short* P = pk[slot];
    const int e_c = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(P + PK_HDR));
    if (e_c <= -1000) continue;                       // nothing but padding / dropped rows from here on (sorted)
    // rows are sorted, dead ones (padding, all-zero dO, beyond the drop span) last: a chunk whose row 32 is dead has a dead second half
    // (Lq = 333 in 6 chunks of 64: the last one holds 13 live rows -- one twelfth of this kernel's matrix and softmax work was on zeros)
    const int halves = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(P + PK_HDR)[1]);
    if (e_c != e_ref) {
      if (e_ref > -1000) {
        const float f = ldexpf(1.0f, min(e_ref - e_c, 126));
#pragma unroll
        for (int u = 0; u < KT; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) { dk0[u][r] *= f; dk1[u][r] *= f; dv0[u][r] *= f; dv1[u][r] *= f; }
      }
      e_ref = e_c;
    }
    if (DROP) {
      // keep bits of chunk c: maskS[c & 1][key block][row] bytes (bit j = key j of the block), 2 KT Philox calls per thread
      const unsigned int q_orig = (unsigned int)reinterpret_cast<const int*>(P + PK_PERM)[t & 63];
#pragma unroll
      for (int i = 0; i < 2 * KT; ++i)
        maskS[c & 1][((t >> 6) + 4 * i) * C16 + (t & 63)] =
            (unsigned char)drop_keep8(dkey, (uint32_t)(within * (8 * KT) + (t >> 6) + 4 * i), q_orig, (uint32_t)bh, drop_site, drop_thr);
      ring_barrier();
    }
    if (!any_live) continue;                            // (after the cooperative mask generation and its barrier)
    const unsigned short *Pq = P + qoff0, *Pp = P + poff;
    const float* Pn = reinterpret_cast<const float*>(P + PK_NL) + g * 8;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      if (hf == 1 && halves < 2) continue;             // wave-uniform; dead rows have P' = 2^-inf = 0: identical bits
      s16x8 qf[2], of[2], qtp[2], otp[2];
      f32x4 nl4[2], nd4[2];
#pragma unroll
      for (int T = 0; T < 2; ++T) {
        qf[T] = *reinterpret_cast<const s16x8*>(Pq + PK_QROWS + DKV_QJ(hf * 2 + T));
        of[T] = *reinterpret_cast<const s16x8*>(Pq + PK_OROWS + DKV_QJ(hf * 2 + T));
        nl4[T] = *reinterpret_cast<const f32x4*>(Pn + hf * 32 + T * 4);
        nd4[T] = *reinterpret_cast<const f32x4*>(Pn + (PK_ND - PK_NL) / 2 + hf * 32 + T * 4);
      }
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        qtp[pl] = *reinterpret_cast<const s16x8*>(Pp + PK_QPL + ((pl * 2 + hf) * 16) * 32);
        otp[pl] = *reinterpret_cast<const s16x8*>(Pp + PK_OPL + ((pl * 2 + hf) * 16) * 32);
      }
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
          dp[u][T] = mfma_f16(of[T], vhh[u], DROP ? f32x4{0.f, 0.f, 0.f, 0.f} : nd4[T]);
          s[u][T] = mfma_f16(qf[T], kll[u], s[u][T]);
        }
#pragma unroll
      for (int u = 0; u < KT; ++u)
#pragma unroll
        for (int T = 0; T < 2; ++T) dp[u][T] = mfma_f16(of[T], vll[u], dp[u][T]);
#pragma unroll
      for (int u = 0; u < KT; ++u) {
        if (!live[u]) continue;            // a key tile beyond S (wave-uniform)
        unsigned long long keep8 = 0;      // byte j: keep flags of row hf * 32 + g * 8 + j for this tile's key block
        if (DROP) keep8 = *reinterpret_cast<const unsigned long long*>(&maskS[c & 1][((wave * KT + u) * 2 + (li >> 3)) * C16 + hf * 32 + g * 8]);
        unsigned int pw[4], pl_[4], gw[4], gl[4];
#pragma unroll
        for (int T = 0; T < 2; ++T) {
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            float p0 = __builtin_amdgcn_exp2f(s[u][T][2 * pr]), p1 = __builtin_amdgcn_exp2f(s[u][T][2 * pr + 1]);
            float g0, g1;
            if (DROP) {
              const int j = T * 4 + 2 * pr;
              const float m0 = ((keep8 >> (8 * j + (li & 7))) & 1ull) ? drop_scale : 0.f;
              const float m1 = ((keep8 >> (8 * (j + 1) + (li & 7))) & 1ull) ? drop_scale : 0.f;
              g0 = p0 * __builtin_fmaf(m0, dp[u][T][2 * pr], nd4[T][2 * pr]);
              g1 = p1 * __builtin_fmaf(m1, dp[u][T][2 * pr + 1], nd4[T][2 * pr + 1]);
              p0 *= m0;                                  // dV sees the dropped weights
              p1 *= m1;
            } else {
              g0 = p0 * dp[u][T][2 * pr];
              g1 = p1 * dp[u][T][2 * pr + 1];
            }
#ifdef A3D_BF16_SPLIT_RNE
            pk_bf16_2(p0, p1, pw[T * 2 + pr], pl_[T * 2 + pr]);
            pk_bf16_2(g0, g1, gw[T * 2 + pr], gl[T * 2 + pr]);
#else
            pk_bf16_2t(p0, p1, pw[T * 2 + pr], pl_[T * 2 + pr]);
            pk_bf16_2t(g0, g1, gw[T * 2 + pr], gl[T * 2 + pr]);
#endif
          }
        }
        const s16x8 pf = __builtin_bit_cast(s16x8, (u32x4_){pw[0], pw[1], pw[2], pw[3]});
        const s16x8 gf = __builtin_bit_cast(s16x8, (u32x4_){gw[0], gw[1], gw[2], gw[3]});
        f32x4& dv = hf ? dv1[u] : dv0[u];
        f32x4& dk = hf ? dk1[u] : dk0[u];
        dv = mfma_bf16_16x16x32(otp[0], pf, dv);
        dk = mfma_bf16_16x16x32(qtp[0], gf, dk);
        dv = mfma_bf16_16x16x32(otp[1], pf, dv);
        dk = mfma_bf16_16x16x32(qtp[1], gf, dk);
        dv = mfma_bf16_16x16x32(otp[0], __builtin_bit_cast(s16x8, (u32x4_){pl_[0], pl_[1], pl_[2], pl_[3]}), dv);
        dk = mfma_bf16_16x16x32(qtp[0], __builtin_bit_cast(s16x8, (u32x4_){gl[0], gl[1], gl[2], gl[3]}), dk);
      }
    }
  }
  wait_vm<0>();
#undef DKV_QJ
  const float sk = (e_ref > -1000) ? ldexpf(1.0f, e_ref - (int)B_OFF) : 0.f, sv = sk * LOG2E_F;
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

int a3d::attn16_check_shapes(const char* fn, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit, int qmod) {
  if (B <= 0 || H <= 0 || Lq <= 0 || Lqp < Lq || (Lqp % qmod) != 0 || S <= 0 || Sp < S || (Sp % C16) != 0 || nsplit < 1 ||
      nsplit > 64 || Sp > MASKW * 32) {
    set_error("%s: bad argument (B=%d H=%d Lq=%d Lqp=%d S=%d Sp=%d nsplit=%d; need Lqp %% %d == 0, Sp %% 64 == 0, Sp <= %d)", fn,
              B, H, Lq, Lqp, S, Sp, nsplit, qmod, MASKW * 32);
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

// flags: A3D_ATTN16_V_ROWS -- Vp holds value ROWS (ones in channel 15 of the hi part; a3d_proj_rope_split16 parts | 8);
//        A3D_ATTN16_NOGRAD -- no backward will follow: the low part of P is formed only where a key dominates (PP = 2, see the kernel)
static int attn16_fwd_launch(const char* fn, const void* Qr, const void* Kr, const void* Vp, const unsigned char* kmask, float* O,
                             float* LSE2, float* ws, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit,
                             const unsigned long long* drop_state, unsigned int drop_site, float drop_p, int flags, void* stream);
extern "C" int a3d_attn16_fwd(const void* Qr, const void* Kr, const void* Vp, const unsigned char* kmask, float* O,
                              float* LSE2, float* ws, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit,
                              const unsigned long long* drop_state, unsigned int drop_site, float drop_p, void* stream) {
  return attn16_fwd_launch("a3d_attn16_fwd", Qr, Kr, Vp, kmask, O, LSE2, ws, B, H, Lq, Lqp, S, Sp, nsplit, drop_state, drop_site,
                           drop_p, 0, stream);
}
extern "C" int a3d_attn16_fwd_rows(const void* Qr, const void* Kr, const void* Vr, const unsigned char* kmask, float* O,
                                   float* LSE2, float* ws, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit,
                                   const unsigned long long* drop_state, unsigned int drop_site, float drop_p, int nograd,
                                   void* stream) {
  return attn16_fwd_launch("a3d_attn16_fwd_rows", Qr, Kr, Vr, kmask, O, LSE2, ws, B, H, Lq, Lqp, S, Sp, nsplit, drop_state,
                           drop_site, drop_p, A3D_ATTN16_V_ROWS | (nograd ? A3D_ATTN16_NOGRAD : 0), stream);
}
static int attn16_fwd_launch(const char* fn, const void* Qr, const void* Kr, const void* Vp, const unsigned char* kmask, float* O,
                             float* LSE2, float* ws, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit,
                             const unsigned long long* drop_state, unsigned int drop_site, float drop_p, int flags, void* stream) {
  const bool v_rows = (flags & A3D_ATTN16_V_ROWS) != 0, nograd = (flags & A3D_ATTN16_NOGRAD) != 0;
  int rc = attn16_check_shapes(fn, B, H, Lq, Lqp, S, Sp, nsplit, 16);
  if (rc) return rc;
  if (!Qr || !Kr || !Vp || !O || !LSE2 || (nsplit > 1 && !ws)) { set_error("%s: null pointer", fn); return A3D_ERR_ARG; }
  bool drop; unsigned int thr; float dscale;
  rc = drop_params(fn, drop_state, drop_p, drop, thr, dscale);
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
  static const bool fast = getenv("A3D_ATTN_FAST") && atoi(getenv("A3D_ATTN_FAST")) != 0;     // single-fp16 P in the forward
#define A3D_L16F(DROPV, QTV, PPV, ST, SITE, THR, SC)                                                                     \
  do { if (v_rows) hipLaunchKernelGGL((attn16_fwd_kernel<DROPV, QTV, PPV, true>), grid, dim3(256), 0, s, (const unsigned short*)Qr, \
                     (const unsigned short*)Kr, (const unsigned short*)Vp, kmask, O, LSE2, Op, Mp, Lp, B, H, Lq, Lqp, S, \
                     Sp, nsplit, ST, SITE, THR, SC);                                                                     \
       else hipLaunchKernelGGL((attn16_fwd_kernel<DROPV, QTV, PPV, false>), grid, dim3(256), 0, s, (const unsigned short*)Qr, \
                     (const unsigned short*)Kr, (const unsigned short*)Vp, kmask, O, LSE2, Op, Mp, Lp, B, H, Lq, Lqp, S, \
                     Sp, nsplit, ST, SITE, THR, SC); } while (0)
  // P parts: both everywhere when a backward follows (its D = dO . O and its recomputed weights must agree with this pass to ~2^-20:
  // near-uniform attention has gradients that are a small covariance on a large common mode -- measured in round 6: the adaptive low
  // part leaves the forward inside 5e-4 but puts the level-0 ghost-attention gradients of the cfg-4 shape 1-3 % off); adaptive for
  // gradient-free passes (evaluation, sampling); A3D_ATTN_LO_ADAPT=1 forces adaptive, A3D_ATTN_FAST=1 the single-part kernels
  static const bool lo_adapt_env = getenv("A3D_ATTN_LO_ADAPT") && atoi(getenv("A3D_ATTN_LO_ADAPT")) != 0;
  const bool lo_adapt = nograd || lo_adapt_env;
#define A3D_L16F_PP(DROPV, QTV, ST, SITE, THR, SC)                                                                       \
  do { if (fast) A3D_L16F(DROPV, QTV, 1, ST, SITE, THR, SC); else if (lo_adapt) A3D_L16F(DROPV, QTV, 2, ST, SITE, THR, SC);   \
       else A3D_L16F(DROPV, QTV, 3, ST, SITE, THR, SC); } while (0)
  if (drop) {
    if (QT == 2) A3D_L16F_PP(true, 2, drop_state, drop_site, thr, dscale);
    else A3D_L16F_PP(true, 1, drop_state, drop_site, thr, dscale);
  } else {
    if (QT == 2) A3D_L16F_PP(false, 2, nostate, 0u, 0u, 1.0f);
    else A3D_L16F_PP(false, 1, nostate, 0u, 0u, 1.0f);
  }
#undef A3D_L16F_PP
#undef A3D_L16F
  rc = check_launch(fn);
  if (rc) return rc;
  if (nsplit > 1) rc = attn16_launch_combine(Op, Mp, Lp, O, LSE2, B, H, Lq, Lqp, nsplit, s);
  return rc;
}

int a3d::attn16_launch_combine(const float* Op, const float* Mp, const float* Lp, float* O, float* LSE2, int B, int H, int Lq,
                               int Lqp, int nsplit, hipStream_t s) {
  const size_t rows = (size_t)B * H * Lqp;
  const int cg = (int)std::min<size_t>((rows * HDP + 255) / 256, 4096);
  hipLaunchKernelGGL(attn16_combine_kernel, dim3(cg), dim3(256), 0, s, Op, Mp, Lp, O, LSE2, B, H, Lq, Lqp, nsplit);
  return check_launch("a3d_attn16_fwd(combine)");
}

extern "C" size_t a3d_attn16_bwd_pack_bytes(int B, int H, int Lqp) {
  return (size_t)B * H * (Lqp / C16) * PK_HALFS * sizeof(unsigned short);
}

extern "C" int a3d_attn16_bwd(const void* Qr, const void* Qp, const void* Kr, const void* Kp, const void* Vr,
                              const unsigned char* kmask, const float* O, const float* dO, const float* LSE2, void* dOr,
                              void* pack, float* D, int* rexp, float* dQp, float* dK, float* dV, int B, int H, int Lq,
                              int Lqp, int S, int Sp, int nsplit, const unsigned long long* drop_state,
                              unsigned int drop_site, float drop_p, void* stream) {
  int rc = attn16_check_shapes("a3d_attn16_bwd", B, H, Lq, Lqp, S, Sp, nsplit, 64);
  if (rc) return rc;
  (void)Qp; (void)Kp;      // round 6: the q / k planes are no longer read (K^T comes from transposed LDS reads of the rows); may be NULL
  if (!Qr || !Kr || !Vr || !O || !dO || !LSE2 || !dOr || !pack || !D || !rexp || !dQp || !dK || !dV) {
    set_error("a3d_attn16_bwd: null pointer");
    return A3D_ERR_ARG;
  }
  bool drop; unsigned int thr; float dscale;
  rc = drop_params("a3d_attn16_bwd", drop_state, drop_p, drop, thr, dscale);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)Lqp * 12 + (size_t)2 * PREP_GROUP * 64 * 64;
  constexpr size_t PREP_LDS_MAX = 150 * 1024;          // the dynamic-LDS attribute set below (160 KB per CU on gfx950)
  if (lds > PREP_LDS_MAX) {
    const size_t max_lqp = (PREP_LDS_MAX - (size_t)2 * PREP_GROUP * 64 * 64) / 12 / 64 * 64;
    set_error("a3d_attn16_bwd: Lqp = %d exceeds the per-(b, h) row sort of the prep kernel (12 B of LDS per query: Lqp <= %zu); "
              "use the bf16x3 backward (a3d_attn_bwd_bf16) for longer query sets", Lqp, max_lqp);
    return A3D_ERR_ARG;
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn16_bwd_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PREP_LDS_MAX);
    attr_set = true;
  }
  hipLaunchKernelGGL(attn16_bwd_prep_kernel, dim3(B * H), dim3(256), lds, s, dO, O, LSE2, (const unsigned short*)Qr,
                     (unsigned short*)dOr, D, rexp, (unsigned short*)pack, B, H, Lq, Lqp);
  rc = check_launch("a3d_attn16_bwd(prep)");
  if (rc) return rc;
  static const int qt_env = getenv("A3D_ATTN_QT") ? atoi(getenv("A3D_ATTN_QT")) : 0;
  const int QT = qt_env ? qt_env : (Lq > 64 ? 2 : 1);
  const int KT = qt_env ? qt_env : (((size_t)B * H * (Sp / 128) >= 1024 && Lq > 16) ? 2 : 1);
  const dim3 gq(xcd_grid(B * H, cdiv(Lqp, 64 * QT) * nsplit)), gk(xcd_grid(B * H, cdiv(Sp, 64 * KT)));
  const unsigned long long* nostate = nullptr;
  static const bool fast = getenv("A3D_ATTN_FAST") && atoi(getenv("A3D_ATTN_FAST")) != 0;     // single-fp16 G
#define A3D_L16Q_(DROPV, QTV, GPV, ST, SITE, THR, SC)                                                                     \
  hipLaunchKernelGGL((attn16_bwd_dq_kernel<DROPV, QTV, GPV>), gq, dim3(256), 0, s, (const unsigned short*)Qr,            \
                     (const unsigned short*)Kr, (const unsigned short*)Vr, kmask,                                         \
                     (const unsigned short*)dOr, LSE2, D, rexp, dQp, B, H, Lq, Lqp, S, Sp, nsplit, ST, SITE, THR, SC)
#define A3D_L16K_(DROPV, KTV, GPV, ST, SITE, THR, SC)                                                                     \
  hipLaunchKernelGGL((attn16_bwd_dkv_kernel<DROPV, KTV, GPV>), gk, dim3(256), 0, s, (const unsigned short*)pack,         \
                     (const unsigned short*)Kr, (const unsigned short*)Vr, kmask, dK, dV, B, H, Lqp, S, Sp, ST, SITE,    \
                     THR, SC)
#define A3D_L16Q(DROPV, QTV, ST, SITE, THR, SC) \
  do { if (fast) A3D_L16Q_(DROPV, QTV, 1, ST, SITE, THR, SC); else A3D_L16Q_(DROPV, QTV, 2, ST, SITE, THR, SC); } while (0)
#define A3D_L16K(DROPV, KTV, ST, SITE, THR, SC) \
  do { if (fast) A3D_L16K_(DROPV, KTV, 1, ST, SITE, THR, SC); else A3D_L16K_(DROPV, KTV, 2, ST, SITE, THR, SC); } while (0)
  if (drop) {
    if (QT == 2) A3D_L16Q(true, 2, drop_state, drop_site, thr, dscale);
    else A3D_L16Q(true, 1, drop_state, drop_site, thr, dscale);
  } else {
    if (QT == 2) A3D_L16Q(false, 2, nostate, 0u, 0u, 1.0f);
    else A3D_L16Q(false, 1, nostate, 0u, 0u, 1.0f);
  }
  rc = check_launch("a3d_attn16_bwd(dq)");
  if (rc) return rc;
  if (drop) {
    if (KT == 2) A3D_L16K(true, 2, drop_state, drop_site, thr, dscale);
    else A3D_L16K(true, 1, drop_state, drop_site, thr, dscale);
  } else {
    if (KT == 2) A3D_L16K(false, 2, nostate, 0u, 0u, 1.0f);
    else A3D_L16K(false, 1, nostate, 0u, 0u, 1.0f);
  }
#undef A3D_L16Q
#undef A3D_L16K
#undef A3D_L16Q_
#undef A3D_L16K_
  return check_launch("a3d_attn16_bwd(dkv)");
}
