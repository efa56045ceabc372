// One evaluation of the ChainedDiffuser denoising network, fused for the 100-step sampling loop (inference only).
//
// Reference: DiffusionHead.forward / _one_attention_round (model/trajectory_optimization/diffusion_head.py:200-363) called
// once per step by DiffusionPlanner.conditional_sample (diffusion_model.py:86-119), on top of ParallelAttentionLayer
// (model/utils/layers.py:115-218) and MultiheadCustomAttention (multihead_custom_attention.py:157-462).
//
// The unfused path issues ~200 launches per step for B * L = 1024 trajectory rows (25 per layer: AdaLN, projection,
// RoPE + operand split, attention, combine, out-projection, LayerNorm, ...), each a few microseconds of latency.  Here a
// step is 18 launches:
//   a3d_dn_head   per sample:  traj_encoder MLP -> [+ index embedding -> q-proj -> attention over the 53 instruction
//                 tokens -> out-proj -> LayerNorm]                                     (diffusion_head.py:214-219,330-335)
//   per layer (8): a3d_dn_cross  per (sample, head, key split): AdaLN(x + index embedding) -> this head's q-projection
//                 -> RoPE -> flash attention of the L <= 16 queries against the cached context K / V, streamed from
//                 HBM straight into MFMA operands (no LDS staging: a 16-query tile has no reuse to stage for)
//                 a3d_dn_rest   per sample: combine the key splits -> out-proj -> LayerNorm -> self-attention block
//                 (AdaLN, q/k/v projection, RoPE, 16 x 16 softmax, out-proj, LayerNorm) -> AdaLN -> FFN -> LayerNorm
//   a3d_dn_tail   per sample: position / rotation regressors -> trajectory update -> DDPM step (inpainting, clipping,
//                 posterior mean + noise)                      (diffusion_head.py:268-272, diffusion_model.py:106-117)
// Context cache (round 6): the operand formats of the split-fp16 attention (attention16.hip) -- K as rows16 [B][H][Sp][32] fp16 =
// hi(16) | lo(16), V as planes16 [B][H][2][16][Sp] fp16 hi / lo with 1.0 in the padded channel 15 of the hi plane (the softmax
// denominator rides on the PV MFMA) -- 128 B per key and head, written by ONE a3d_proj_rope_split16 launch per layer.  Per 32-key
// half a wave issues 4 + 2 v_mfma_f32_16x16x32_f16 (two-part q and k: 22-bit logits; P single fp16 with the low part added only
// around dominant keys: sampling keeps no gradient) instead of the 8 v_mfma_f32_16x16x4_f32 + 3 bf16 MFMAs of rounds 2 - 5
// (fp32 K rows, two-part bf16 P and V): the round-5 phase probe showed the streaming loop MFMA / VALU-issue bound, not HBM bound
// (8 f32 MFMAs = 256 issue cycles + ~150 vector instructions per half), so the exact-f32 logits were what the loop spent its time on.
// All dense layers use the exact-f32 MFMA of linear.hip (an fmaf chain in k order).
#include "attn_ring.h"
#include "../../include/act3d_hip.h"
#include <string.h>

namespace a3d {

constexpr int DR = 16;            // trajectory rows (steps) per sample: one MFMA tile
constexpr int LDX = 132;          // LDS row stride (floats) of the [16][<=128] tiles
constexpr int LDQK = 260;         // [16][<=256]
constexpr int LDH = 516;          // [16][<=512]

// Y[16][N] = act(X[16][K] W[N][K]^T + bias), X / Y in LDS (X zero-padded to a multiple of 16 columns), W / bias in
// global memory (L2-resident: every sample's workgroup reads the same weights).  The workgroup's waves (4 or 8) take the
// 16-column output tiles round-robin: the more waves, the more weight fragments are in flight per CU -- the per-sample
// kernels are bound by that (PMC: MfmaUtil 2.5 %, VALUBusy 3.3 %, 8.6 MB of HBM fetch per 103 us launch).  Contraction index permutation: MFMA j of a 16-channel block contracts lane group g with channel
// 4 g + j, so that A and B are one float4 each per block.  The weight fragments of the NEXT (tile, 128-channel chunk) are
// fetched into registers while the current chunk's 32 MFMAs issue (a load inside the MFMA loop would expose one L2 round
// trip per 16 channels: measured 136 us for the layer remainder against ~20 us of MFMA issue); two accumulators per tile
// keep the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32 off the 32-cycle issue rate.
constexpr int WCH = 8;            // 16-channel blocks per register chunk

// (The fragment loads are branch-free -- clamped addresses + selects: a per-lane guarded load compiles to one basic block
// and one s_waitcnt vmcnt(0) per load, i.e. one exposed L2 round trip per 16 channels.)
template <int ACT, bool VEC>
__device__ __forceinline__ void wg_linear_impl(const float* Xs, int ldx, int K, const float* __restrict__ W, int ldw,
                                               const float* __restrict__ bias, int N, float* Ys, int ldy) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int ntile = (N + 15) >> 4, nblk = (K + 15) >> 4;
  const int nchunk = (nblk + WCH - 1) / WCH;
  auto loadw = [&](int ct, int c, float4 (&w)[WCH]) {
    const int n = ct * 16 + li;
    const float* wrow = W + (size_t)min(n, N - 1) * ldw;
    const bool row_ok = n < N;
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int k0 = (c * WCH + i) * 16 + 4 * g;
      float4 wv;
      if (VEC) {                                             // K % 4 == 0, rows 16-byte aligned
        wv = *reinterpret_cast<const float4*>(wrow + min(k0, K - 4));
        if (!(row_ok && k0 < K)) wv = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        wv.x = wrow[min(k0 + 0, K - 1)];
        wv.y = wrow[min(k0 + 1, K - 1)];
        wv.z = wrow[min(k0 + 2, K - 1)];
        wv.w = wrow[min(k0 + 3, K - 1)];
        wv.x = (row_ok && k0 + 0 < K) ? wv.x : 0.f;
        wv.y = (row_ok && k0 + 1 < K) ? wv.y : 0.f;
        wv.z = (row_ok && k0 + 2 < K) ? wv.z : 0.f;
        wv.w = (row_ok && k0 + 3 < K) ? wv.w : 0.f;
      }
      w[i] = wv;
    }
  };
  float4 wc[WCH], wn[WCH];
#pragma unroll
  for (int i = 0; i < WCH; ++i) wn[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  int ct = wave, c = 0;
  if (ct < ntile) loadw(ct, 0, wc);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  while (ct < ntile) {
    int nct = ct, nc = c + 1;
    if (nc == nchunk) { nc = 0; nct = ct + nwave; }
    if (nct < ntile) loadw(nct, nc, wn);                     // wave-uniform condition
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int blk = min(c * WCH + i, nblk - 1);            // blocks past the end multiply zero weights
      const float4 a = *reinterpret_cast<const float4*>(&Xs[li * ldx + blk * 16 + 4 * g]);
      acc0 = mfma_f32_16x16x4(a.x, wc[i].x, acc0);
      acc1 = mfma_f32_16x16x4(a.y, wc[i].y, acc1);
      acc0 = mfma_f32_16x16x4(a.z, wc[i].z, acc0);
      acc1 = mfma_f32_16x16x4(a.w, wc[i].w, acc1);
    }
    if (c == nchunk - 1) {
      const int n = ct * 16 + li;
      if (n < N) {
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc0[r] + acc1[r] + bv;
          if (ACT == 1) v = fmaxf(v, 0.f);
          Ys[(g * 4 + r) * ldy + n] = v;
        }
      }
      acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
      acc1 = acc0;
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) wc[i] = wn[i];
    ct = nct;
    c = nc;
  }
  __syncthreads();
}

template <int ACT>
__device__ __forceinline__ void wg_linear(const float* Xs, int ldx, int K, const float* __restrict__ W, int ldw,
                                          const float* __restrict__ bias, int N, float* Ys, int ldy) {
  const bool vec = ((ldw & 3) == 0) && ((K & 3) == 0) && ((((uintptr_t)W) & 15) == 0);
  if (vec) wg_linear_impl<ACT, true>(Xs, ldx, K, W, ldw, bias, N, Ys, ldy);
  else wg_linear_impl<ACT, false>(Xs, ldx, K, W, ldw, bias, N, Ys, ldy);
}

// zero the pad columns [E, Epad) of a [16][ld] tile
__device__ __forceinline__ void wg_zero_pad(float* T, int ld, int E, int Epad) {
  for (int i = threadIdx.x; i < DR * (Epad - E); i += blockDim.x) {
    const int r = i / (Epad - E), c = E + i - r * (Epad - E);
    T[r * ld + c] = 0.f;
  }
}

// Y = LayerNorm(A + R) * gamma + beta over E <= 128 columns, 16 threads per row (two-pass, like add_ln_fwd_kernel).  A thread
// reads and writes the same (row, column) set and the row reductions are register shuffles, so Y may alias A or R and only
// the trailing barrier is needed; gamma / beta are fetched up front (one L2 round trip, not one per element).
__device__ __forceinline__ void wg_add_layernorm(const float* A, int lda, const float* R, int ldr, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float* Y, int ldy, int E) {
  const int r = (threadIdx.x >> 4) & 15, s = threadIdx.x & 15;
  const bool worker = threadIdx.x < 256;                  // 16 rows x 16 threads; further waves only join the barrier
  float v[8], gm[8], bt[8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = s + 16 * i;
    const bool ok = worker && c < E;
    gm[i] = ok ? gamma[c] : 0.f;
    bt[i] = ok ? beta[c] : 0.f;
    v[i] = ok ? A[r * lda + c] + R[r * ldr + c] : 0.f;
    sum += v[i];
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)E;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float d = (s + 16 * i < E) ? v[i] - mean : 0.f;
    q += d * d;
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.0f / sqrtf(q / (float)E + 1e-5f);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = s + 16 * i;
    if (worker && c < E) Y[r * ldy + c] = (v[i] - mean) * rstd * gm[i] + bt[i];
  }
  __syncthreads();
}

// Y = (X [+ sem]) * (1 + mod[c]) + mod[E + c]   (AdaLN, layers.py:273-290; mod == nullptr: plain X [+ sem]); optionally
// Y2 = the same modulation of X without the index embedding (the value stream of the self-attention).  E <= 128.
__device__ __forceinline__ void wg_adaln(const float* X, int ldx, const float* __restrict__ sem, const float* __restrict__ mod,
                                         float* Y, int ldy, int L, int E, float* Y2 = nullptr, int ldy2 = 0) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {                            // DR * E <= 2048 = 8 x 256: predicated, all loads issued up front
    const int i = threadIdx.x + j * blockDim.x;
    if (i < DR * E) {
      const int r = i / E, c = i - r * E;
      const float x0 = X[r * ldx + c];
      const float sc = mod ? 1.0f + mod[c] : 1.0f, sh = mod ? mod[E + c] : 0.f;
      const float se = (sem && r < L) ? sem[r * E + c] : 0.f;
      Y[r * ldy + c] = (x0 + se) * sc + sh;
      if (Y2) Y2[r * ldy2 + c] = x0 * sc + sh;
    }
  }
  __syncthreads();
}

// in-place RoPE-3D of the E-wide blocks starting at columns col0, col0 + E, ... (nblk blocks) of T, by the rows' xyz
// (block 0 is first multiplied by `scale0`: q = q * d^-1/2 before the rotation, multihead_custom_attention.py:325);
// freq == nullptr: scaling only
__device__ __forceinline__ void wg_rope(float* T, int ld, int col0, int nblk, const float* __restrict__ xyz, int ldxyz,
                                        const float* __restrict__ freq, int L, int E, float scale0) {
  const int half = E >> 1, third = E / 3;
  for (int i = threadIdx.x; i < DR * half; i += blockDim.x) {
    const int r = i / half, p = i - r * half;
    const int c = 2 * p;
    float sn = 0.f, cs = 1.f;
    if (freq && r < L) {
      const int axis = c / third;
      const int k = (c - axis * third) >> 1;
      fast_sincos(xyz[r * ldxyz + axis] * freq[k], &sn, &cs);
    }
    for (int bk = 0; bk < nblk; ++bk) {
      float* q = T + r * ld + col0 + bk * E + c;
      const float sc = bk == 0 ? scale0 : 1.0f;
      const float y0 = q[0] * sc, y1 = q[1] * sc;
      q[0] = y0 * cs - y1 * sn;
      q[1] = y1 * cs + y0 * sn;
    }
  }
  __syncthreads();
}

// O[l][h * 15 + d] = softmax_s(q_h[l] . k_h[s] + mask) v_h[s]; one thread per (row, head), keys in a loop (S is small: the
// 53 instruction tokens or the <= 16 trajectory steps).  Q in LDS; K / V rows at stride ldk / ldv (global or LDS).
__device__ __forceinline__ void wg_small_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv,
                                                   const unsigned char* __restrict__ kmask, int S, int H, float* O, int ldo) {
  const int r = (threadIdx.x >> 4) & 15, h = threadIdx.x & 15;
  if (threadIdx.x < 256 && h < H) {
    float q[HD], acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { q[d] = Q[r * ldq + h * HD + d]; acc[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    for (int s = 0; s < S; ++s) {
      if (kmask && kmask[s]) continue;
      float sc = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) sc += q[d] * K[(size_t)s * ldk + h * HD + d];
      const float mn = fmaxf(m, sc);
      const float a = __expf(m - mn), p = __expf(sc - mn);
      l = l * a + p;
#pragma unroll
      for (int d = 0; d < HD; ++d) acc[d] = acc[d] * a + p * V[(size_t)s * ldv + h * HD + d];
      m = mn;
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) O[r * ldo + h * HD + d] = acc[d] * inv;
  }
  __syncthreads();
}

// The same for H <= 8 heads on all 512 threads: four threads per (row, head) take every fourth key and merge their softmax states
// with two xor-shuffles (the 53 instruction tokens are a serial 53-step chain per thread above: ~30 us of the persistent sampler's
// 50 us head).  Requires blockDim.x == 512.
__device__ __forceinline__ void wg_small_attention4(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, int S, int H,
                                                    float* O, int ldo) {
  const int t = threadIdx.x;
  const int part = t & 3, h = (t >> 2) & 7, r = t >> 5;
  const bool on = h < H;
  float q[HD], acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) { q[d] = on ? Q[r * ldq + h * HD + d] : 0.f; acc[d] = 0.f; }
  float m = -INFINITY, l = 0.f;
  if (on) {
    for (int s = part; s < S; s += 4) {
      float sc = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) sc += q[d] * K[(size_t)s * ldk + h * HD + d];
      const float mn = fmaxf(m, sc);
      const float a = __expf(m - mn), p = __expf(sc - mn);
      l = l * a + p;
#pragma unroll
      for (int d = 0; d < HD; ++d) acc[d] = acc[d] * a + p * V[(size_t)s * ldv + h * HD + d];
      m = mn;
    }
  }
  float mg = fmaxf(m, __shfl_xor(m, 1, 64));
  mg = fmaxf(mg, __shfl_xor(mg, 2, 64));
  const float f = (m == -INFINITY) ? 0.f : __expf(m - mg);
  l *= f;
  l += __shfl_xor(l, 1, 64);
  l += __shfl_xor(l, 2, 64);
  const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    float v = acc[d] * f;
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    if (on && part == 0) O[r * ldo + h * HD + d] = v * inv;
  }
  __syncthreads();
}

__device__ __forceinline__ void wg_load_rows(const float* __restrict__ src, int E, int L, float* T, int ld, int Epad) {
  for (int i = threadIdx.x; i < DR * Epad; i += blockDim.x) {
    const int r = i / Epad, c = i - r * Epad;
    T[r * ld + c] = (r < L && c < E) ? src[r * E + c] : 0.f;
  }
  __syncthreads();
}

// L2 warm-up of the weights a per-sample kernel is about to stream.  Between two layers' per-sample kernels the cross
// attention streams ~200 MB of K / V cache through the 4 MiB L2 of every XCD, so each dn_rest launch finds its 0.75 MB of
// weights evicted and would fetch them miss by miss on its dependent path (25 phases, one exposed HBM round trip per weight
// chunk: PMC showed 8.6 MB of HBM fetch per launch = 8 XCDs x the layer's weights, MfmaUtil 2.5 %).  Here every workgroup
// touches, up front and all at once, one dword of each 128-byte line of its share of the matrices (the workgroups of one
// XCD -- blockIdx % 8 -- split the lines among themselves), so that the misses overlap each other instead of the compute.
struct WarmList { const float* p[6]; int n[6]; };
__device__ __forceinline__ void wg_warm_l2(const WarmList& wl, int nblocks) {
  const int per_xcd = max(1, min(8, nblocks >> 3));
  const int part = (blockIdx.x >> 3) % per_xcd;
  float keep[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) keep[i] = 0.f;
#pragma unroll
  for (int m = 0; m < 6; ++m) {
    if (wl.p[m] == nullptr) continue;
    const int nline = (wl.n[m] + 31) >> 5;
#pragma unroll
    for (int i = 0; i < 8; ++i) {                               // <= 8 x blockDim lines per matrix and workgroup share
      const int line = (i * (int)blockDim.x + (int)threadIdx.x) * per_xcd + part;
      if (line < nline) keep[i] += wl.p[m][min(line * 32, wl.n[m] - 1)];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(keep[i]));
}

// ------------------------------------------------------------------------------------------------ head
__global__ __launch_bounds__(512) void dn_head_kernel(const float* __restrict__ traj, int D, a3d_dn_head_params p,
                                                      float* __restrict__ x_out, int L, int E, int H, int warm) {
  __shared__ __attribute__((aligned(16))) float Xs[DR * LDX], As[DR * LDX], Ts[DR * LDX], Qs[DR * LDX];
  extern __shared__ __attribute__((aligned(16))) float kvS[];          // [S_lang][2E]: the instruction tokens' k | v rows
  const int b = blockIdx.x;
  const int Epad = (E + 15) & ~15;
  if (warm & 1) {
    const WarmList wl = {{p.enc_w1, p.lang_kv ? p.q_w : nullptr, p.lang_kv ? p.out_w : nullptr, nullptr, nullptr, nullptr},
                         {E * E, E * E, E * E, 0, 0, 0}};
    wg_warm_l2(wl, gridDim.x);
  }
  // trajectory rows (D = 9 channels) padded to one 16-channel block
  for (int i = threadIdx.x; i < DR * 16; i += blockDim.x) {
    const int r = i >> 4, c = i & 15;
    As[r * LDX + c] = (r < L && c < D) ? traj[((size_t)b * L + r) * D + c] : 0.f;
  }
  wg_zero_pad(Ts, LDX, E, Epad);
  wg_zero_pad(Xs, LDX, E, Epad);
  wg_zero_pad(Qs, LDX, E, Epad);
  __syncthreads();
  wg_linear<1>(As, LDX, D, p.enc_w0, D, p.enc_b0, E, Ts, LDX);          // Linear(9, E) + ReLU
  wg_linear<0>(Ts, LDX, E, p.enc_w1, E, p.enc_b1, E, Xs, LDX);          // Linear(E, E)         -> trajectory features
  if (p.lang_kv) {
    // traj_lang_attention: q = x + index embedding, keys = values = the instruction tokens, no positions, no AdaLN, no FFN
    wg_zero_pad(As, LDX, E, Epad);
    wg_adaln(Xs, LDX, p.sem, nullptr, As, LDX, L, E);
    wg_linear<0>(As, LDX, E, p.q_w, E, p.q_b, E, Qs, LDX);
    const float scale = 1.0f / sqrtf((float)HD);
    for (int i = threadIdx.x; i < DR * E; i += blockDim.x) Qs[(i / E) * LDX + i % E] *= scale;
    __syncthreads();
    const float* kv = p.lang_kv + (size_t)b * p.S_lang * 2 * E;
    for (int i = threadIdx.x; i < p.S_lang * 2 * E; i += blockDim.x) kvS[i] = kv[i];    // coalesced; the key loop reads LDS
    __syncthreads();
    wg_small_attention(Qs, LDX, kvS, 2 * E, kvS + E, 2 * E, nullptr, p.S_lang, H, As, LDX);
    wg_linear<0>(As, LDX, E, p.out_w, E, p.out_b, E, Ts, LDX);
    wg_add_layernorm(Xs, LDX, Ts, LDX, p.ln_g, p.ln_b, Xs, LDX, E);
  }
  for (int i = threadIdx.x; i < L * E; i += blockDim.x) x_out[(size_t)b * L * E + i] = Xs[(i / E) * LDX + i % E];
}

// ------------------------------------------------------------------------------------------------ streaming attention core
// One wave, 16 queries of one head against a key range of the cached context, 32 keys per step, fragments straight from global
// memory (no LDS staging: a 16-query tile has no reuse to stage for).  Shared by dn_cross_kernel (per-phase launches) and the
// persistent sampler's stream role: the same arithmetic, so the two paths agree to the summation order of their combines.
//   scores : S^T = K Q^T on two 16-key tiles (rows interleaved so that a lane ends up with 8 consecutive keys, as attention16.hip):
//            [k_hi | k_lo] . [q_hi | q_hi] + [k_hi | k_lo] . [q_lo | q_lo]; q carries log2(e) / sqrt(d): log2 units, and the MFMA
//            accumulator is initialised with P_OFF - m: exp2 applies directly to the MFMA result
//   weights: lazy running maximum (revised when a score exceeds it by 2^P_THR: a wave-uniform, rarely taken branch); P = fp16(p),
//            plus its low part fp16(p - P) only for halves that hold a key within 2^-LO_SPAN of the running denominator (the
//            gradient-free rule of attention16.hip: sampling never back-propagates)
//   values : O^T += V^T P with two-part V; channel 15 of the hi plane is 1.0: the denominator accumulates on the MFMA
struct DnKv16 { s16x8 k0, k1, vh, vl; };
struct DnStream {
  s16x8 qhh, qll;                 // B operands [q_hi | q_hi], [q_lo | q_lo] of the lane's query (li), channels (g & 1) * 8 .. + 7
  f32x4 cin, acc;                 // P_OFF - m (score accumulator init), O^T accumulator (rows = channels 4 g + r, column = query li)
  float m_run, thr;               // running maximum (log2 units), dominance threshold (units of the score tiles)
  int n;                          // halves consumed so far
};
constexpr float DN_LO_SPAN = 6.0f;
// K / V fragments in flight per wave of the persistent stream role.  3; measured in round 6 against 6 (A3D_HIPCC_FLAGS=-DDN_DEPTH=6):
// 0.761 vs 0.835 ms per denoise step at cfg-3 -- the key loop takes ~1 us per 32-key half per wave at EITHER depth and with 2.4x
// less issue work than in round 5: 8 waves x 4 KB per us = ~30 GB/s per CU, the per-CU share of the chip's HBM bandwidth
// (MI355X_MICROARCH.md: ~10 B / cycle / CU for an HBM-missing stream).  The streaming role is bound by the NUMBER OF CUs it owns.
#ifndef DN_DEPTH
#define DN_DEPTH 3
#endif
// q8: the 8 channels (g & 1) * 8 .. + 7 of the lane's query row, already scaled by 1 / sqrt(d) (natural-log logits)
__device__ __forceinline__ void dn_stream_init(DnStream& st, const float (&q8)[8]) {
  unsigned int h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) pk_f16_2(q8[2 * i] * LOG2E_F, q8[2 * i + 1] * LOG2E_F, h[i], l[i]);
  st.qhh = __builtin_bit_cast(s16x8, (u32x4_){h[0], h[1], h[2], h[3]});
  st.qll = __builtin_bit_cast(s16x8, (u32x4_){l[0], l[1], l[2], l[3]});
  st.cin = f32x4{P_OFF, P_OFF, P_OFF, P_OFF};
  st.acc = f32x4{0.f, 0.f, 0.f, 0.f};
  st.m_run = 0.f;
  st.thr = -INFINITY;
  st.n = 0;
}
// Kb: this (sample, head)'s key rows [Sp][32]; Vhi / Vlo: the lane's channel row (li) of the hi / lo value plane [Sp]
template <bool NONTEMPORAL>
__device__ __forceinline__ DnKv16 dn_stream_load(const unsigned short* Kb, const unsigned short* Vhi, const unsigned short* Vlo, int hf,
                                                 int li, int g) {
  typedef const __attribute__((address_space(1))) s16x8* g_s16x8_p;
  const int key0 = hf * 32;
  const int r0 = key0 + (li >> 2) * 8 + (li & 3);
  DnKv16 f;
  if (NONTEMPORAL) {
    // the K / V stream is read once per item: it must not push the layer weights the sample role re-reads out of the 4 MB L2s
    // (global address space stated explicitly: the pointers come out of a device-memory table -> generic pointers, FLAT loads)
    f.k0 = __builtin_nontemporal_load((g_s16x8_p)(Kb + (size_t)r0 * 32 + g * 8));
    f.k1 = __builtin_nontemporal_load((g_s16x8_p)(Kb + (size_t)(r0 + 4) * 32 + g * 8));
    f.vh = __builtin_nontemporal_load((g_s16x8_p)(Vhi + key0 + g * 8));
    f.vl = __builtin_nontemporal_load((g_s16x8_p)(Vlo + key0 + g * 8));
  } else {
    f.k0 = *reinterpret_cast<const s16x8*>(Kb + (size_t)r0 * 32 + g * 8);
    f.k1 = *reinterpret_cast<const s16x8*>(Kb + (size_t)(r0 + 4) * 32 + g * 8);
    f.vh = *reinterpret_cast<const s16x8*>(Vhi + key0 + g * 8);
    f.vl = *reinterpret_cast<const s16x8*>(Vlo + key0 + g * 8);
  }
  return f;
}
// s_lim: number of valid keys (keys >= s_lim are masked; 0 masks the whole half)
__device__ __forceinline__ void dn_stream_consume(DnStream& st, const DnKv16& f, int hf, int s_lim, int g) {
  const int key0 = hf * 32;
  f32x4 s[2];
  if (key0 + 32 > s_lim) {                                          // wave-uniform: only the last half of the context (or a dead one)
#pragma unroll
    for (int T = 0; T < 2; ++T) {
      f32x4 c = st.cin;
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = (key0 + g * 8 + T * 4 + r < s_lim) ? c[r] : -INFINITY;
      s[T] = mfma_f16(T ? f.k1 : f.k0, st.qhh, c);
    }
  } else {
    s[0] = mfma_f16(f.k0, st.qhh, st.cin);
    s[1] = mfma_f16(f.k1, st.qhh, st.cin);
  }
  s[0] = mfma_f16(f.k0, st.qll, s[0]);
  s[1] = mfma_f16(f.k1, st.qll, s[1]);
  const float mx = fmaxf(max3f(s[0][0], s[0][1], s[0][2]), fmaxf(max3f(s[0][3], s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
  const bool first = st.n == 0;
  bool lo_part = false;
  if (first || __builtin_amdgcn_ballot_w64(mx > st.thr) != 0ull) {
    lo_part = true;
    if (first || __builtin_amdgcn_ballot_w64(mx > P_OFF + P_THR) != 0ull) {
      const float cm = colmax4(mx);                                 // exact maximum of the lane's query over the half
      float shift = first ? (cm - P_OFF) : fmaxf(cm - P_OFF, 0.f);
      if (cm == -INFINITY) shift = 0.f;                             // every key so far masked
      st.m_run += shift;
      st.thr -= shift;
      const float alpha = __builtin_amdgcn_exp2f(-shift);
#pragma unroll
      for (int r = 0; r < 4; ++r) { st.acc[r] *= alpha; st.cin[r] -= shift; s[0][r] -= shift; s[1][r] -= shift; }
    }
  }
  float p[8];
  unsigned int w[4];
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      p[T * 4 + 2 * pr] = __builtin_amdgcn_exp2f(s[T][2 * pr]);
      p[T * 4 + 2 * pr + 1] = __builtin_amdgcn_exp2f(s[T][2 * pr + 1]);
      w[T * 2 + pr] = pk_f16(p[T * 4 + 2 * pr], p[T * 4 + 2 * pr + 1]);
    }
  const s16x8 pf = __builtin_bit_cast(s16x8, (u32x4_){w[0], w[1], w[2], w[3]});
  st.acc = mfma_f16(f.vh, pf, st.acc);
  st.acc = mfma_f16(f.vl, pf, st.acc);
  if (lo_part) {                                                    // wave-uniform
    unsigned int wl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wl[i] = lo_f16(p[2 * i], p[2 * i + 1], w[i]);
    st.acc = mfma_f16(f.vh, __builtin_bit_cast(s16x8, (u32x4_){wl[0], wl[1], wl[2], wl[3]}), st.acc);
  }
  if ((st.n & 7) == 0) {
    // refresh the dominance threshold from the running denominator (channel 15 = lane group 3, register 3); stale in between
    // = smaller = conservative
    const float lcol = colmax4((g == 3) ? st.acc[3] : 0.f);
    st.thr = fminf(__builtin_amdgcn_logf(lcol) - DN_LO_SPAN, P_OFF + P_THR);
  }
  ++st.n;
}
// the running maximum in the units the combines use (natural log: they weight partials with exp(m_s - m)); any_valid == false
// (the wave's range held no valid key): -inf, so that the split drops out of the maximum
__device__ __forceinline__ float dn_stream_max_nat(const DnStream& st, bool any_valid) {
  return any_valid ? (st.m_run - P_OFF) * LN2_F : -INFINITY;
}

// ------------------------------------------------------------------------------------------------ cross attention
// grid: (sample, head, key split) flattened XCD-aware; workspace Op [nsplit][B][H][16][16] (column 15 = sum_k p), Mp [..][16]
__global__ __launch_bounds__(256) void dn_cross_kernel(const float* __restrict__ x, const float* __restrict__ traj, int D,
                                                       a3d_dn_cross_params p, float* __restrict__ Op, float* __restrict__ Mp,
                                                       int B, int L, int E, int H, int S, int Sp, int nsplit) {
  __shared__ __attribute__((aligned(16))) float As[DR * LDX];
  __shared__ __attribute__((aligned(16))) float Pre[DR * 16];
  __shared__ __attribute__((aligned(16))) float Qh[DR * 16];
  __shared__ __attribute__((aligned(16))) float Cacc[4][16][16];
  __shared__ float Cm[4][16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  int group, sp;
  if (!xcd_decode(nsplit, B * H, group, sp)) return;
  const int b = group / H, h = group - b * H;
  const size_t bh = (size_t)b * H + h;
  // ---- q rows of this head: AdaLN(x + index embedding) W_q^T + b_q, scaled, rotated
  for (int i = t; i < DR * E; i += blockDim.x) {
    const int r = i / E, c = i - r * E;
    float v = 0.f;
    if (r < L) {
      v = x[((size_t)b * L + r) * E + c] + (p.sem ? p.sem[r * E + c] : 0.f);
      if (p.mod) v = v * (1.0f + p.mod[c]) + p.mod[E + c];
    }
    As[r * LDX + c] = v;
  }
  __syncthreads();
  const int c_lo = (h * HD) & ~1;                     // 16-channel window holding the head's 15 channels and their RoPE partners
  {
    const int r = t >> 4, j = t & 15, c = c_lo + j;
    float acc = 0.f;
    if (c < E) {
      const float* wrow = p.q_w + (size_t)c * E;
      for (int k = 0; k < E; ++k) acc += As[r * LDX + k] * wrow[k];
      acc = (acc + p.q_b[c]) * (1.0f / sqrtf((float)HD));
    }
    Pre[r * 16 + j] = acc;
  }
  __syncthreads();
  if (p.freq && (t & 1) == 0) {
    const int r = t >> 4, j = t & 15, c = c_lo + j;       // c even: pair (c, c + 1)
    if (c + 1 < E && r < L) {
      const int third = E / 3;
      const int axis = c / third;
      const int k = (c - axis * third) >> 1;
      const float th = traj[((size_t)b * L + r) * D + axis] * p.freq[k];
      float sn, cs;
      fast_sincos(th, &sn, &cs);
      const float y0 = Pre[r * 16 + j], y1 = Pre[r * 16 + j + 1];
      Pre[r * 16 + j] = y0 * cs - y1 * sn;
      Pre[r * 16 + j + 1] = y1 * cs + y0 * sn;
    }
  }
  __syncthreads();
  {
    const int r = t >> 4, d = t & 15;
    Qh[r * 16 + d] = (d < HD && r < L) ? Pre[r * 16 + (h * HD - c_lo) + d] : 0.f;
  }
  __syncthreads();
  float q8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) q8[e] = Qh[li * 16 + (g & 1) * 8 + e];           // (channel 15 is the zero pad)
  DnStream st;
  dn_stream_init(st, q8);

  // ---- this wave's range of 32-key halves
  const int NH = Sp >> 5;
  const int parts = nsplit * 4, part = sp * 4 + wave;
  const int h_beg = (int)((long long)NH * part / parts), h_end = (int)((long long)NH * (part + 1) / parts);
  const unsigned short* Kb = reinterpret_cast<const unsigned short*>(p.Kf) + bh * (size_t)Sp * 32;
  const unsigned short* Vhi = p.Vt + ((bh * 2 + 0) * 16 + li) * (size_t)Sp;
  const unsigned short* Vlo = p.Vt + ((bh * 2 + 1) * 16 + li) * (size_t)Sp;
  // Three fragments in flight per wave in three FIXED register sets, the loop unrolled by three, unconditional clamped refills and
  // masked consumption (see the persistent sampler's stream role below for the why: with rotating sets -- cur = nxt -- the register
  // moves wait for loads in flight and the effective depth was one: 2 us per 32-key half, the bare memory latency)
  const int NHc = Sp >> 5;
  DnKv16 f0, f1, f2;
  __builtin_amdgcn_sched_barrier(0);
  f0 = dn_stream_load<false>(Kb, Vhi, Vlo, min(h_beg, NHc - 1), li, g);
  __builtin_amdgcn_sched_barrier(0);
  f1 = dn_stream_load<false>(Kb, Vhi, Vlo, min(h_beg + 1, NHc - 1), li, g);
  __builtin_amdgcn_sched_barrier(0);
  f2 = dn_stream_load<false>(Kb, Vhi, Vlo, min(h_beg + 2, NHc - 1), li, g);
  __builtin_amdgcn_sched_barrier(0);
  int hf = h_beg;
  for (; hf + 3 <= h_end; hf += 3) {
    dn_stream_consume(st, f0, hf, S, g);
    __builtin_amdgcn_sched_barrier(0);
    f0 = dn_stream_load<false>(Kb, Vhi, Vlo, min(hf + 3, NHc - 1), li, g);
    __builtin_amdgcn_sched_barrier(0);
    dn_stream_consume(st, f1, hf + 1, S, g);
    __builtin_amdgcn_sched_barrier(0);
    f1 = dn_stream_load<false>(Kb, Vhi, Vlo, min(hf + 4, NHc - 1), li, g);
    __builtin_amdgcn_sched_barrier(0);
    dn_stream_consume(st, f2, hf + 2, S, g);
    __builtin_amdgcn_sched_barrier(0);
    f2 = dn_stream_load<false>(Kb, Vhi, Vlo, min(hf + 5, NHc - 1), li, g);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (hf < h_end) dn_stream_consume(st, f0, hf, S, g);              // the last one or two halves of the range (wave-uniform branches)
  if (hf + 1 < h_end) dn_stream_consume(st, f1, hf + 1, S, g);
  const f32x4 acc = st.acc;
  const float m_run = dn_stream_max_nat(st, h_beg < h_end && h_beg * 32 < S);
  // ---- combine the four waves (disjoint key ranges) and write this split's partial
#pragma unroll
  for (int r = 0; r < 4; ++r) Cacc[wave][li][g * 4 + r] = acc[r];
  if (g == 0) Cm[wave][li] = m_run;
  __syncthreads();
  {
    const int q = t >> 4, d = t & 15;
    const float m = fmaxf(fmaxf(Cm[0][q], Cm[1][q]), fmaxf(Cm[2][q], Cm[3][q]));
    const float m_use = (m == -INFINITY) ? 0.f : m;
    float o = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) o += __expf(Cm[w][q] - m_use) * Cacc[w][q][d];
    const size_t row = (((size_t)sp * B + b) * H + h) * 16 + q;
    Op[row * 16 + d] = o;
    if (d == 0) Mp[row] = m;
  }
}

// Small per-layer vectors (biases, LayerNorm / AdaLN parameters, the index embedding) -> LDS in ONE batch of loads at the
// head of the kernel.  Each of them lives on its own page of the flat parameter buffer / modulation tables, and a load issued
// where it is needed exposes a full miss (translation + HBM, 2-8 us measured per phase with A3D_DN_PROF) on the workgroup's
// dependent path; issued together the misses overlap.  Vectors of <= 512 elements; `big` (the index embedding) <= 2048.
struct VecList { const float* p[12]; int n[12]; };
__device__ __forceinline__ void wg_stage_vectors(const VecList& vl, float* const (&dst)[12], const float* big, int nbig, float* bigdst) {
  float r[12][2], rb[8];
  const int t = threadIdx.x, nt = blockDim.x;                  // 256 or 512 threads: two passes cover 512 elements
#pragma unroll
  for (int j = 0; j < 12; ++j)
#pragma unroll
    for (int u = 0; u < 2; ++u) r[j][u] = (vl.p[j] && t + u * nt < vl.n[j]) ? vl.p[j][t + u * nt] : 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) rb[i] = (big && t + i * nt < nbig) ? big[t + i * nt] : 0.f;
#pragma unroll
  for (int j = 0; j < 12; ++j)
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (vl.p[j] && t + u * nt < vl.n[j]) dst[j][t + u * nt] = r[j][u];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (big && t + i * nt < nbig) bigdst[t + i * nt] = rb[i];
}

// Phase timestamps of workgroup 0 of the last dn_rest launch (A3D_DN_PROF=1; wall_clock64 = 100 MHz): development aid read
// back by a3d_dbg_dn_prof, no effect on results.
__device__ long long g_dn_prof[32];
#define DN_MARK(i) do { if ((warm & 2) && blockIdx.x == 0 && threadIdx.x == 0) g_dn_prof[i] = wall_clock64(); } while (0)

// ------------------------------------------------------------------------------------------------ rest of a layer, looped
// One workgroup per sample runs everything between two cross-attentions.  Round 3 measured the first (straight-line) version
// of this kernel, 82 KB of code that a workgroup executes exactly once, with A3D_DN_PROF: phases with almost no arithmetic
// (AdaLN, RoPE, the ReLU instantiation of the dense layer) took 4-12 us each in proportion to their CODE SIZE (instruction
// fetch: 64 KB instruction cache per CU pair, every line a miss).  This version runs the same 13 operations as a LOOP over an
// operation table (built by the host, passed by value, indexed with scalar loads) around ONE copy of each operation's code:
// five of the thirteen are the dense layer, three the residual LayerNorm, two the AdaLN, so after the first visit an
// operation's instructions are cache hits (32 KB of code; AdaLN 4.4 -> 1.0 us, RoPE 5.8 -> 1.1 us; the straight-line kernel was
// deleted in round 4).
enum { DN_OP_LINEAR = 0, DN_OP_ADDLN = 1, DN_OP_ADALN = 2, DN_OP_ROPE = 3, DN_OP_ATTN = 4 };
struct DnOp {
  int type;
  int a, b, c, d, e, f, g;     // LDS offsets (floats from the start of the dynamic LDS; -1 = none) / sizes, per type (see dn_build_ops)
  const float* W;              // LINEAR: weights (global)
};
struct DnOpTable { int n; DnOp op[13]; };

// operation table of one layer remainder (dn_rest_loop_kernel / the persistent sampler): LDS offsets in floats from the start
// of the dynamic LDS.  One definition for the host (a3d_dn_rest passes the table by value) and the device (the persistent
// sampler builds it per layer in LDS).
__host__ __device__ inline void dn_build_ops(DnOpTable& tab, const a3d_dn_rest_params& p, int E) {
  const int oX = 0, oA = DR * LDX, oB = 2 * DR * LDX, oT = 3 * DR * LDX, oQK = 4 * DR * LDX, oH = oQK + DR * LDQK, oP = oH + DR * LDH;
  const int oMisc = oP + 2560 + DR * 128;
  int n = 0;
  auto lin = [&](int x, int ldx, int K, const float* W, int bias, int N, int y, int ldy, int act) {
    tab.op[n++] = DnOp{DN_OP_LINEAR, x, ldx, K, bias, N, y, ldy | (act << 16), W};
  };
  auto other = [&](int type, int a, int b, int c, int d, int e) { tab.op[n++] = DnOp{type, a, b, c, d, e, 0, 0, nullptr}; };
  lin(oA, LDX, E, p.c_out_w, p.c_out_b ? oP : -1, E, oT, LDX, 0);
  other(DN_OP_ADDLN, oX, oT, oP + 128, oP + 256, oX);
  if (p.s_in_w) {
    other(DN_OP_ADALN, oX, p.sem ? oP + 2560 : -1, p.s_mod ? oP + 384 : -1, oA, oB);
    lin(oA, LDX, E, p.s_in_w, p.s_in_b ? oP + 640 : -1, 2 * E, oQK, LDQK, 0);
    lin(oB, LDX, E, p.s_in_w + (size_t)2 * E * E, p.s_in_b ? oP + 640 + 2 * E : -1, E, oH, LDH, 0);
    other(DN_OP_ROPE, oQK, oMisc, p.freq ? oMisc + 160 : -1, 0, 0);
    other(DN_OP_ATTN, oQK, oQK + E, oH, oMisc + 192, oA);
    lin(oA, LDX, E, p.s_out_w, p.s_out_b ? oP + 1024 : -1, E, oT, LDX, 0);
    other(DN_OP_ADDLN, oX, oT, oP + 1152, oP + 1280, oX);
  }
  if (p.f_w1) {
    other(DN_OP_ADALN, oX, -1, p.f_mod ? oP + 1408 : -1, oA, -1);
    lin(oA, LDX, E, p.f_w1, p.f_b1 ? oP + 1664 : -1, p.F, oH, LDH, 1);
    lin(oH, LDH, p.F, p.f_w2, p.f_b2 ? oP + 2176 : -1, E, oT, LDX, 0);
    other(DN_OP_ADDLN, oA, oT, oP + 2304, oP + 2432, oX);
  }
  tab.n = n;
}

// Round 4 measured a deeper weight fetch here -- a wave's rounds fetched four at a time (32 float4 per lane in flight) before
// their MFMAs, on the hypothesis that every round exposes an L2 round trip: 0.913 ms per denoise step with guarded loads, 1.10 ms
// with unconditional ones (redundant re-fetches for waves with fewer than four rounds), against 0.86 ms for the one-round-ahead
// pipeline below (gpurun r04e / r04f, cfg-3).  The hypothesis is refuted; the pipeline stays.
template <bool VEC>
__device__ __forceinline__ void wg_linear_rt(const float* Xs, int ldx, int K, const float* __restrict__ W, int ldw, const float* bias,
                                             int N, float* Ys, int ldy, int act) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int ntile = (N + 15) >> 4, nblk = (K + 15) >> 4;
  const int nchunk = (nblk + WCH - 1) / WCH;
  auto loadw = [&](int ct, int c, float4 (&w)[WCH]) {
    const int n = ct * 16 + li;
    const float* wrow = W + (size_t)min(n, N - 1) * ldw;
    const bool row_ok = n < N;
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int k0 = (c * WCH + i) * 16 + 4 * g;
      float4 wv;
      if (VEC) {
        wv = *reinterpret_cast<const float4*>(wrow + min(k0, K - 4));
        if (!(row_ok && k0 < K)) wv = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        wv.x = wrow[min(k0 + 0, K - 1)];
        wv.y = wrow[min(k0 + 1, K - 1)];
        wv.z = wrow[min(k0 + 2, K - 1)];
        wv.w = wrow[min(k0 + 3, K - 1)];
        wv.x = (row_ok && k0 + 0 < K) ? wv.x : 0.f;
        wv.y = (row_ok && k0 + 1 < K) ? wv.y : 0.f;
        wv.z = (row_ok && k0 + 2 < K) ? wv.z : 0.f;
        wv.w = (row_ok && k0 + 3 < K) ? wv.w : 0.f;
      }
      w[i] = wv;
    }
  };
  float4 wc[WCH], wn[WCH];
#pragma unroll
  for (int i = 0; i < WCH; ++i) wn[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  int ct = wave, c = 0;
  if (ct < ntile) loadw(ct, 0, wc);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  while (ct < ntile) {
    int nct = ct, nc = c + 1;
    if (nc == nchunk) { nc = 0; nct = ct + nwave; }
    if (nct < ntile) loadw(nct, nc, wn);
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int blk = min(c * WCH + i, nblk - 1);
      const float4 a = *reinterpret_cast<const float4*>(&Xs[li * ldx + blk * 16 + 4 * g]);
      acc0 = mfma_f32_16x16x4(a.x, wc[i].x, acc0);
      acc1 = mfma_f32_16x16x4(a.y, wc[i].y, acc1);
      acc0 = mfma_f32_16x16x4(a.z, wc[i].z, acc0);
      acc1 = mfma_f32_16x16x4(a.w, wc[i].w, acc1);
    }
    if (c == nchunk - 1) {
      const int n = ct * 16 + li;
      if (n < N) {
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc0[r] + acc1[r] + bv;
          if (act == 1) v = fmaxf(v, 0.f);
          Ys[(g * 4 + r) * ldy + n] = v;
        }
      }
      acc0 = f32x4{0.f, 0.f, 0.f, 0.f};
      acc1 = acc0;
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) wc[i] = wn[i];
    ct = nct;
    c = nc;
  }
  __syncthreads();
}

constexpr int DN_MISC_XYZ = 160;                 // floats of the misc area that hold the sample's trajectory rows (L * D)
constexpr int DN_PS = 2560 + DR * 128 + 256;     // floats of staged parameters: vectors | index embedding | xyz rows, freq, mask
__global__ __launch_bounds__(512) void dn_rest_loop_kernel(const float* __restrict__ x_in, const float* __restrict__ traj, int D,
                                                           const float* __restrict__ Op, const float* __restrict__ Mp,
                                                           a3d_dn_rest_params p, DnOpTable tab, float* __restrict__ x_out, int B,
                                                           int L, int E, int H, int nsplit, int warm) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;
  float* As = Xs + DR * LDX;
  float* Hs = smem + 4 * DR * LDX + DR * LDQK;
  float* Ps = Hs + DR * LDH;
  const int b = blockIdx.x;
  const int Epad = (E + 15) & ~15;
  DN_MARK(0);
  {
    float* const vd[12] = {Ps, Ps + 128, Ps + 256, Ps + 384, Ps + 640, Ps + 1024, Ps + 1152, Ps + 1280, Ps + 1408, Ps + 1664, Ps + 2176,
                           Ps + 2304};
    const VecList vl = {{p.c_out_b, p.c_ln_g, p.c_ln_b, p.s_in_w ? p.s_mod : nullptr, p.s_in_w ? p.s_in_b : nullptr,
                         p.s_in_w ? p.s_out_b : nullptr, p.s_in_w ? p.s_ln_g : nullptr, p.s_in_w ? p.s_ln_b : nullptr,
                         p.f_w1 ? p.f_mod : nullptr, p.f_w1 ? p.f_b1 : nullptr, p.f_w1 ? p.f_b2 : nullptr, p.f_w1 ? p.f_ln_g : nullptr},
                        {E, E, E, 2 * E, 3 * E, E, E, E, 2 * E, p.F, E, E}};
    wg_stage_vectors(vl, vd, (p.sem && p.s_in_w) ? p.sem : nullptr, L * E, Ps + 2560);
    const int t = threadIdx.x;
    float* const misc = Ps + 2560 + DR * 128;                    // [0, 144): xyz rows (ld = D); [160, 192): freq; [192, 208): key mask
    if (p.f_w1 && t < E) Ps[2432 + t] = p.f_ln_b[t];
    if (p.s_in_w) {
      if (t < L * D) misc[t] = traj[(size_t)b * L * D + t];
      if (p.freq && t < E / 6) misc[160 + t] = p.freq[t];
      if (t < DR) misc[192 + t] = (p.kmask && t < L && p.kmask[(size_t)b * L + t]) ? 1.f : 0.f;
    }
  }
  if (warm & 1) {
    const WarmList wl = {{p.c_out_w, p.s_in_w, p.s_in_w ? p.s_out_w : nullptr, p.f_w1, p.f_w1 ? p.f_w2 : nullptr, nullptr},
                         {E * E, 3 * E * E, E * E, p.F * E, p.F * E, 0}};
    wg_warm_l2(wl, gridDim.x);
  }
  DN_MARK(1);
  wg_load_rows(x_in + (size_t)b * L * E, E, L, Xs, LDX, Epad);
  // pad columns of every tile a dense layer reads: As, Bs, Ts ([E, Epad)) and the FFN hidden tile ([F, Fpad))
#pragma nounroll
  for (int k = 1; k < 4; ++k) wg_zero_pad(smem + k * DR * LDX, LDX, E, Epad);
  if (p.f_w1) {
    const int padw = ((p.F + 15) & ~15) - p.F;
    for (int i = threadIdx.x; i < DR * padw; i += blockDim.x) Hs[(i / padw) * LDH + p.F + i % padw] = 0.f;
  }
  // ---- cross-attention output: combine the key splits
#pragma unroll 4
  for (int i = threadIdx.x; i < DR * E; i += blockDim.x) {
    const int r = i / E, c = i - r * E;
    const int h = c / HD, d = c - h * HD;
    float m = -INFINITY;
    for (int s = 0; s < nsplit; ++s) m = fmaxf(m, Mp[(((size_t)s * B + b) * H + h) * 16 + r]);
    const float m_use = (m == -INFINITY) ? 0.f : m;
    float num = 0.f, den = 0.f;
    for (int s = 0; s < nsplit; ++s) {
      const size_t row = (((size_t)s * B + b) * H + h) * 16 + r;
      const float w = __expf(Mp[row] - m_use);
      num += w * Op[row * 16 + d];
      den += w * Op[row * 16 + 15];
    }
    As[r * LDX + c] = den > 0.f ? num / den : 0.f;
  }
  __syncthreads();
  DN_MARK(2);
#pragma nounroll
  for (int i = 0; i < tab.n; ++i) {
    const DnOp& op = tab.op[i];
    switch (op.type) {
      case DN_OP_LINEAR: {
        // a: X, b: ldx, c: K, d: bias, e: N, f: Y, g: ldy | act in the sign bit of ldw-free field: K < 0 never; act = op.type >> 8 unused
        const bool vec = ((op.c & 3) == 0) && ((((uintptr_t)op.W) & 15) == 0);
        const int act = op.g >> 16, ldy = op.g & 0xFFFF;
        if (vec) wg_linear_rt<true>(smem + op.a, op.b, op.c, op.W, op.c, op.d >= 0 ? smem + op.d : nullptr, op.e, smem + op.f, ldy, act);
        else wg_linear_rt<false>(smem + op.a, op.b, op.c, op.W, op.c, op.d >= 0 ? smem + op.d : nullptr, op.e, smem + op.f, ldy, act);
        break;
      }
      case DN_OP_ADDLN:        // a: A, b: R, c: gamma, d: beta, e: Y
        wg_add_layernorm(smem + op.a, LDX, smem + op.b, LDX, smem + op.c, smem + op.d, smem + op.e, LDX, E);
        break;
      case DN_OP_ADALN:        // a: X, b: sem, c: mod, d: Y, e: Y2
        wg_adaln(smem + op.a, LDX, op.b >= 0 ? smem + op.b : nullptr, op.c >= 0 ? smem + op.c : nullptr, smem + op.d, LDX, L, E,
                 op.e >= 0 ? smem + op.e : nullptr, LDX);
        break;
      case DN_OP_ROPE:         // a: T (q | k blocks), b: xyz rows, c: freq
        wg_rope(smem + op.a, LDQK, 0, 2, smem + op.b, D, op.c >= 0 ? smem + op.c : nullptr, L, E, 1.0f / sqrtf((float)HD));
        break;
      default: {               // DN_OP_ATTN   a: Q, b: K, c: V, d: mask (floats), e: O
        const float* mk = smem + op.d;
        const float* Q = smem + op.a;
        const float* Kk = smem + op.b;
        const float* V = smem + op.c;
        float* O = smem + op.e;
        const int r = (threadIdx.x >> 4) & 15, h = threadIdx.x & 15;
        if (threadIdx.x < 256 && h < H) {
          float q[HD], acc[HD];
#pragma unroll
          for (int d = 0; d < HD; ++d) { q[d] = Q[r * LDQK + h * HD + d]; acc[d] = 0.f; }
          float m = -INFINITY, l = 0.f;
          for (int s2 = 0; s2 < L; ++s2) {
            if (mk[s2] != 0.f) continue;
            float sc = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) sc += q[d] * Kk[s2 * LDQK + h * HD + d];
            const float mn = fmaxf(m, sc);
            const float a = __expf(m - mn), pw = __expf(sc - mn);
            l = l * a + pw;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = acc[d] * a + pw * V[s2 * LDH + h * HD + d];
            m = mn;
          }
          const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
          for (int d = 0; d < HD; ++d) O[r * LDX + h * HD + d] = acc[d] * inv;
        }
        __syncthreads();
        break;
      }
    }
    DN_MARK(3 + i);
  }
  for (int i = threadIdx.x; i < L * E; i += blockDim.x) x_out[(size_t)b * L * E + i] = Xs[(i / E) * LDX + i % E];
  DN_MARK(17);
}

// ------------------------------------------------------------------------------------------------ tail
__global__ __launch_bounds__(512) void dn_tail_kernel(const float* __restrict__ pos_feats, const float* __restrict__ rot_feats,
                                                      const float* __restrict__ traj, int D, a3d_dn_tail_params p,
                                                      float* __restrict__ traj_out, int L, int E, int t_step, int warm) {
  __shared__ __attribute__((aligned(16))) float Xs[DR * LDX], Ts[DR * LDX], Us[DR * 16];
  const int b = blockIdx.x;
  const int Epad = (E + 15) & ~15;
  if (warm & 1) {
    const WarmList wl = {{p.pos_w0, p.rot_w0, nullptr, nullptr, nullptr, nullptr}, {E * E, E * E, 0, 0, 0, 0}};
    wg_warm_l2(wl, gridDim.x);
  }
  wg_zero_pad(Ts, LDX, E, Epad);
  wg_load_rows(pos_feats + (size_t)b * L * E, E, L, Xs, LDX, Epad);
  wg_linear<1>(Xs, LDX, E, p.pos_w0, E, p.pos_b0, E, Ts, LDX);
  wg_linear<0>(Ts, LDX, E, p.pos_w1, E, p.pos_b1, 3, Us, 16);
  wg_load_rows(rot_feats + (size_t)b * L * E, E, L, Xs, LDX, Epad);
  wg_linear<1>(Xs, LDX, E, p.rot_w0, E, p.rot_b0, E, Ts, LDX);
  wg_linear<0>(Ts, LDX, E, p.rot_w1, E, p.rot_b1, D - 3, Us + 3, 16);
  // out = cat(traj_xyz + d_pos, rot); then the reverse step of diffusion_model.py:106-117 (see a3d_ddpm_step)
  for (int i = threadIdx.x; i < L * D; i += blockDim.x) {
    const int r = i / D, c = i - r * D;
    const size_t gi = ((size_t)b * L + r) * D + c;
    float mo = Us[r * 16 + c] + (c < 3 ? traj[gi] : 0.f);
    if (p.cond_mask && p.cond_mask[gi]) mo = p.cond_data[gi];
    float out = mo;
    if (t_step > 0) {
      const float* cf = ((c < 3) ? p.coef_pos : p.coef_rot) + (size_t)t_step * 3;
      const float x0 = fminf(fmaxf(mo, -1.0f), 1.0f);
      out = cf[0] * x0 + cf[1] * traj[gi];
      if (p.noise) out += cf[2] * p.noise[gi];
    }
    traj_out[gi] = out;
  }
}

// rotated K rows in fp32: out[b][h][n][d] = rope(Y[b, n, :E] * scale)[h * 15 + d] (d < 15; column 15 and rows >= N zero)
__global__ __launch_bounds__(256) void rope_rows_f32_kernel(const float* __restrict__ Y, int ldy, const float* __restrict__ xyz,
                                                            const float* __restrict__ freq, float scale, float* __restrict__ out,
                                                            int B, int N, int Npad, int E, int H) {
  const size_t total = (size_t)B * H * Npad * 16;
  const int third = E / 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i & 15);
    const size_t row = i >> 4;
    const int n = (int)(row % Npad);
    const size_t bh = row / Npad;
    const int h = (int)(bh % H), b = (int)(bh / H);
    float v = 0.f;
    if (d < HD && n < N) {
      const int c = h * HD + d;
      const size_t m = (size_t)b * N + n;
      const float y = Y[m * ldy + c] * scale;
      if (xyz) {
        const int ce = c & ~1;
        const int axis = ce / third;
        const int k = (ce - axis * third) >> 1;
        float sn, cs;
        fast_sincos(xyz[m * 3 + axis] * freq[k], &sn, &cs);
        const float yp = Y[m * ldy + (c ^ 1)] * scale;
        v = (c & 1) ? (y * cs + yp * sn) : (y * cs - yp * sn);
      } else {
        v = y;
      }
    }
    out[i] = v;
  }
}


// ================================================================================================ persistent sampler (round 5)
// The whole 100-step sampling loop of one trajectory batch as ONE launch with two workgroup roles (north_star: "trajectory-noise
// add + eps-prediction loop fused per denoise step"; diffusion_model.py:86-119 over diffusion_head.py:200-363):
//   sample role  (workgroups 0 .. B-1, one per trajectory): the per-sample chain of a step -- traj_encoder [+ instruction
//                attention] -> per layer {q = rope(W_q AdaLN(x + index embedding)) published for the streamers | wait for the
//                layer's cross-attention partials | combine -> out-proj -> LayerNorm -> self-attention block -> FFN block} ->
//                regressors -> DDPM reverse step -- looping over layers AND denoise steps without leaving the kernel;
//   stream role  (the remaining CUs): serve (sample, layer, key split) items from a ready queue: every wave streams one head's
//                slice of the cached context K / V against the sample's 16 published queries (the arithmetic of dn_cross_kernel)
//                and writes the partial (o, m); a counter per sample tells the sample workgroup when its layer is complete.
// Why: with one launch per phase the 64 sample workgroups of a3d_dn_rest sit on 64 of 256 CUs for 61 us per layer (a latency chain:
// MfmaUtil 2.5 %) while the HBM-bound cross-attention launch waits behind them, and vice versa; the samples never interact, so
// nothing but the launch boundaries forces them into lockstep.  Here each sample advances at its own pace, the streamers stay busy
// with whichever samples are ready, and the step-invariant staging of a layer's vectors / weights overlaps the sample's own wait.
// Synchronisation: agent-scope release / acquire on a ready queue (sample -> streamers) and a per-sample completion counter
// (streamers -> sample); all workgroups are co-resident (grid <= CU count, one workgroup per CU by LDS size), every spin loop is
// bounded and raises an abort flag instead of hanging.
struct DnLayerDev { a3d_dn_cross_params c; a3d_dn_rest_params r; };      // c.mod, r.s_mod, r.f_mod: BASES of the [T][2E] tables
struct DnPersist {
  const DnLayerDev* layers;       // device array [n_traj + n_pos + n_rot]
  a3d_dn_head_params head;
  a3d_dn_tail_params tail;        // tail.noise: BASE of the [T][B][L][D] step noise (row t is used at step t > 0), or NULL
  float* traj;                    // [B][L][D] in / out
  float* qbuf;                    // [B][16][128] published queries of the sample's current layer
  float* part;                    // Op [nse][B][H][16][16] | Mp [nse][B][H][16]   (nse = nsplit * nsub)
  int* sync;                      // [0] ticket  [1] queue tail  [2] abort  | [16 + 16 u] completion counter of unit u |
                                  // [.. + 16 b] self-attention exchange counter of sample b | ready queue
  float* kvx;                     // [2 roles][B][2][NT * 16][2E] rotated keys | values of the self-attention, exchanged between a
                                  // sample's row tiles (NT > 1 only; two buffers alternate from layer to layer)
  float* xbuf;                    // [U][2][16][128]: x after the trajectory stack (primary -> helper) | rotation features (helper -> primary)
  int B, L, NT, D, E, H, S, Sp, nsplit, nsub, n_traj, n_pos, n_rot, t_first, nsteps, spin_limit;
  long long* prof;                // development aid (A3D_DN_PROF=1): phase timestamps, see a3d_dn_persist_prof; else NULL
};
constexpr int DNP_PROF_WORDS = 256;      // long longs: [0, 96) 32 items x {ticket, ready, done} of streamer 0; [96, 96 + 7 * 16) layer marks of sample 0; [250..] head / tail
// A trajectory of L <= 64 steps is NT = ceil(L / 16) row tiles; a UNIT = (sample b, tile) = one sample-role workgroup, u = b NT + tile.
// Every row-wise operation (encoder, cross-attention queries / partials, projections, LayerNorm, FFN, regressors, DDPM step) is the
// unit's own; only the self-attention couples the tiles of a sample: they exchange their rotated keys / values through kvx.
// Two sample-role workgroups per unit: the PRIMARY runs head, trajectory stack, position stack and tail; the HELPER runs the
// rotation stack (diffusion_head.py:343-357: the position and the rotation stack both start from the trajectory stack's output and
// are independent), so a step's chain is n_traj + max(n_pos, n_rot) layers deep instead of n_traj + n_pos + n_rot.  They hand
// x over through xbuf (primary -> helper after the trajectory stack, helper -> primary after the rotation stack) with one flag each.
// For the streamers a helper is just another unit: streaming unit id v = role * U + u (qbuf / partial / completion-counter slot).
constexpr int DNP_XDONE0 = 16;                      // sync words: [16 + 16 v] completion counter of streaming unit v < 2U
__host__ __device__ __forceinline__ int dnp_kvdone0(int B, int NT) { return DNP_XDONE0 + 16 * 2 * B * NT; }          // [+ 16 (role B + b)]
__host__ __device__ __forceinline__ int dnp_xtready0(int B, int NT) { return dnp_kvdone0(B, NT) + 16 * 2 * B; }      // [+ 16 u]
__host__ __device__ __forceinline__ int dnp_rdone0(int B, int NT) { return dnp_xtready0(B, NT) + 16 * B * NT; }       // [+ 16 u]
__host__ __device__ __forceinline__ int dnp_grp0(int B, int NT) { return dnp_rdone0(B, NT) + 16 * B * NT; }           // [+ 16 (role B + b)] publishes of the group's tiles
__host__ __device__ __forceinline__ int dnp_queue0(int B, int NT) { return dnp_grp0(B, NT) + 16 * 2 * B; }

// thread 0 of the workgroup spins until *flag >= target (acquire, agent scope); returns false when the kernel is aborting
// (the polls are RELAXED agent-scope loads; no acquire fence follows: see dnp_ld)
__device__ __forceinline__ bool dnp_wait_ge(int* flag, int target, int* abort_flag, int spin_limit) {
  int spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(16);                      // ~0.4 us between polls: 200 pollers must not hammer one memory channel
    if ((++spins & 127) == 0) {
      if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
      if (spins > spin_limit) {
        __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
  return true;
}
// Exchange data (published queries, attention partials) is READ with agent-scope relaxed atomic word loads: they are served from the
// coherence point whatever stale copy the reader's XCD L2 may hold (the flag polls above are the same kind of load and do observe
// the other role's stores), so the reader needs NO acquire fence.  An agent-scope acquire invalidates the XCD's L2: with one per
// queue item (512 per layer round) the layer weights the sample role streams from L2 were evicted continuously and the layer
// remainder ran at Infinity-Cache latency (measured: 1.01 ms per denoise step with the fences vs 0.86 ms for the per-phase launches).
// Writers publish with plain stores + workgroup barrier + agent-scope RELEASE (L2 write-back of a few KB) + the flag store.
__device__ __forceinline__ float dnp_ld(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// ... and WRITTEN with agent-scope relaxed atomic word stores (write-through to the coherence point), followed by the workgroup
// barrier (which waits for every wave's outstanding stores) and a relaxed flag update by thread 0.  No agent-scope RELEASE fence
// either: a release writes back the XCD's whole L2, and with one per queue item the phase probe showed every memory-latency-bound
// phase of the sample role slowing down as the items got finer (layer remainder 53 -> 84 us from 4 to 16 key splits).
__device__ __forceinline__ void dnp_st(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned int*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (Round 6 measured 16-byte write-through stores -- inline-asm global_store_dwordx4 sc1 + an explicit vmcnt drain -- for the partials
// and the published queries: neither phase got shorter (partials + completion 5 us, publish 1.7 us), so the scalar forms stay.)
// The argument block is re-read through an opaque copy of its pointer at every phase: otherwise the compiler hoists ALL its (loop-
// invariant) pointer loads to the top of the role function and keeps ~150 SGPRs live across the step / layer loops (755 SGPR spills).
__device__ __forceinline__ const DnPersist& dnp_args(const DnPersist* p) {
  asm volatile("" : "+s"(p));
  return *p;
}

// the same for the shape scalars: every per-thread index expression (i / E, i % E, row / column offsets ...) of the inlined phase
// helpers is invariant across the step / layer loops, and hoisted to the role's prologue they are hundreds of live VGPRs (398 spills)
__device__ __forceinline__ int dnp_opaque(int v) {
  asm volatile("" : "+v"(v));
  return __builtin_amdgcn_readfirstlane(v);          // uniform again (a scalar register) as far as the compiler is concerned
}

// One queue item: (role, sample bs, layer, key split) against ALL NTL row tiles of the trajectory (round 6; rounds 5's items were per
// tile, so that at the reference's horizon L = 50 every one of the 4 tiles streamed the sample's K / V slice separately: 4x the
// algorithmic reads, 0.022 of the HBM roofline).  A wave (one head, one key sub-range) loads each 32-key fragment ONCE and runs the
// NTL tiles' score / softmax / PV sequences against it -- independent instruction streams the scheduler interleaves.
template <int NTL>
__device__ __forceinline__ void dnp_stream_item(const DnPersist& a, const a3d_dn_cross_params& c, int bs, int v0, int sp, float* Op,
                                                float* Mp, int wave, int li, int g, long long* pmark) {
  const int U = 2 * a.B * a.NT;
  const int nse = a.nsplit * a.nsub;
  const int h = wave % a.H, sub = wave / a.H;
  const int se = sp * a.nsub + sub;
  const size_t bh = (size_t)bs * a.H + h;
  DnStream st[NTL];
#pragma unroll
  for (int tl = 0; tl < NTL; ++tl) {
    const float* qrow = a.qbuf + ((size_t)(v0 + tl) * 16 + li) * 128 + h * HD;     // rows >= L and columns >= E are published as zeros
    float q8[8];                                       // channels (g & 1) * 8 .. + 7 of query li (channel 15 = pad)
#pragma unroll
    for (int e = 0; e < 8; ++e) q8[e] = ((g & 1) * 8 + e < HD) ? dnp_ld(qrow + (g & 1) * 8 + e) : 0.f;
    dn_stream_init(st[tl], q8);
  }
  if (pmark && wave == 0 && (li | g) == 0) pmark[0] = wall_clock64();           // (development aid: the queries have arrived)
  const int NH = a.Sp >> 5;
  const int h_beg = (int)((long long)NH * se / nse), h_end = (int)((long long)NH * (se + 1) / nse);
  const unsigned short* Kb = reinterpret_cast<const unsigned short*>(c.Kf) + bh * (size_t)a.Sp * 32;
  const unsigned short* Vhi = c.Vt + ((bh * 2 + 0) * 16 + li) * (size_t)a.Sp;
  const unsigned short* Vlo = c.Vt + ((bh * 2 + 1) * 16 + li) * (size_t)a.Sp;
  // Three fragments in flight per wave, in THREE FIXED register sets: the loop is unrolled by three and each set is refilled in
  // place right after its use.  (Rotating the sets -- cur = f0; f0 = f1; f1 = f2; f2 = load() -- compiles to register moves of
  // fragments whose loads are still in flight, i.e. a full s_waitcnt per iteration and an effective depth of ONE: the phase
  // probe showed 1.75 us per 32-key half, the bare memory latency, in this loop and in dn_cross_kernel's two-set version.)
  // (Measured and reverted in round 5: THREE halves per softmax update with ping-pong fragment groups made an item SLOWER,
  // 20 -> 23.5 us for 12 halves: gpurun r05q.  The single-half body stays; round 6 changed what it issues, see the file header.)
  const int S_keys = a.S;               // read ONCE: a load of the argument block inside the loop is the newest load there and forces vmcnt(0)
  auto consume = [&](const DnKv16& f, int hf) {
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) dn_stream_consume(st[tl], f, hf, S_keys, g);
  };
  // DN_DEPTH fragments (4 loads of 1 KB per wave each) in flight per wave, in FIXED register sets refilled in place right after their
  // use.  (Rotating the sets -- cur = f0; f0 = f1; ... -- compiles to register moves of fragments whose loads are still in flight,
  // i.e. a full s_waitcnt per iteration and an effective depth of ONE: round 5.)  Round 6's phase marks
  // (profiles/r06_dn_persist_phases.json) put an item at ~20 us: queries 2, first fragments 2, key loop 11 - 14 (12 halves),
  // partials + completion 5.
  DnKv16 f[DN_DEPTH];
#pragma unroll
  for (int i = 0; i < DN_DEPTH; ++i) {
    __builtin_amdgcn_sched_barrier(0);
    f[i] = dn_stream_load<true>(Kb, Vhi, Vlo, min(h_beg + i, NH - 1), li, g);      // (past the range: clamped, loaded, never used)
  }
  __builtin_amdgcn_sched_barrier(0);
  // straight-line body: UNCONDITIONAL refills (clamped addresses) and masked consumption, so that the number of loads in
  // flight at every use is a compile-time constant and the compiler waits with a counted vmcnt, not vmcnt(0)
  // (the scheduling barriers pin the order consume | refill | consume ...: left alone, the machine scheduler sinks all the
  // loads to the end of the body to save registers and the next iteration opens with vmcnt(0) again)
  int hf = h_beg;
  if (pmark && wave == 0 && (li | g) == 0) pmark[1] = wall_clock64();           // (the loads are issued; the loop's first use waits for them)
  for (; hf + DN_DEPTH <= h_end; hf += DN_DEPTH) {
#pragma unroll
    for (int i = 0; i < DN_DEPTH; ++i) {
      consume(f[i], hf + i);
      __builtin_amdgcn_sched_barrier(0);
      f[i] = dn_stream_load<true>(Kb, Vhi, Vlo, min(hf + DN_DEPTH + i, NH - 1), li, g);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < DN_DEPTH - 1; ++i)
    if (hf + i < h_end) consume(f[i], hf + i);                  // the last halves of the range (wave-uniform branches)
  if (pmark && wave == 0 && (li | g) == 0) pmark[2] = wall_clock64();           // (the key range is consumed)
  const bool any_valid = h_beg < h_end && h_beg * 32 < S_keys;
  // this wave's partials: acc[r] = o[query li][d = 4 g + r] (d = 15: sum_k p), running maximum per query, one block per tile's unit
#pragma unroll
  for (int tl = 0; tl < NTL; ++tl) {
    const size_t row0 = (((size_t)se * U + (v0 + tl)) * a.H + h) * 16;
    float* od = &Op[(row0 + li) * 16 + 4 * g];
    dnp_st(od, st[tl].acc[0]); dnp_st(od + 1, st[tl].acc[1]); dnp_st(od + 2, st[tl].acc[2]); dnp_st(od + 3, st[tl].acc[3]);
    if (g == 0) dnp_st(&Mp[row0 + li], dn_stream_max_nat(st[tl], any_valid));
  }
}

// ---- stream role
__device__ __forceinline__ void dnp_stream_role(const DnPersist* ap, float* smem) {
  const DnPersist& a = dnp_args(ap);
  int* sh = reinterpret_cast<int*>(smem);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int NL = a.n_traj + a.n_pos + a.n_rot;
  const int U0 = a.B * a.NT, U = 2 * U0;            // U: streaming units (primaries + helpers)
  const int G = 2 * a.B;                            // groups: (role, sample) = the NT row tiles that share one K / V pass
  const long long total = (long long)a.nsteps * NL * a.B * a.nsplit;
  const int nse = a.nsplit * a.nsub;
  float* Op = a.part;
  float* Mp = a.part + (size_t)nse * U * a.H * 256;
  int* abort_flag = a.sync + 2;
  int nprof = 0;
  for (;;) {
    __syncthreads();                                     // everybody has read the previous item's sh[]
    long long tk0 = 0;
    if (t == 0) {
      if (a.prof) tk0 = wall_clock64();
      const long long i = (long long)atomicAdd(&a.sync[0], 1);
      int code = 0;
      if (i < total) {
        int* slot = a.sync + dnp_queue0(a.B, a.NT) + (int)(i / a.nsplit);
        if (dnp_wait_ge(slot, 1, abort_flag, a.spin_limit)) code = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      sh[0] = i < total ? (int)(i % a.nsplit) : -1;
      sh[1] = code;
      if (a.prof && (int)blockIdx.x == U && nprof < 32) { a.prof[nprof * 3] = tk0; a.prof[nprof * 3 + 1] = wall_clock64(); }
    }
    __syncthreads();
    const int sp = sh[0], code = sh[1];
    if (sp < 0 || code == 0) break;
    const int gl = (code - 1) / G, grp = (code - 1) - gl * G;        // gl: the layer; grp = role B + sample
    const int role = grp / a.B, bs = grp - role * a.B;
    const int v0 = role * U0 + bs * a.NT;                             // first streaming unit (row tile 0) of the group
    const a3d_dn_cross_params& c = dnp_args(ap).layers[gl].c;
    if (wave < a.H * a.nsub) {
      long long* pmark = (a.prof && (int)blockIdx.x == U && nprof < 14) ? a.prof + 208 + 3 * nprof : nullptr;     // prof words [208, 250)
      switch (a.NT) {
        case 1: dnp_stream_item<1>(a, c, bs, v0, sp, Op, Mp, wave, li, g, pmark); break;
        case 2: dnp_stream_item<2>(a, c, bs, v0, sp, Op, Mp, wave, li, g, pmark); break;
        case 3: dnp_stream_item<3>(a, c, bs, v0, sp, Op, Mp, wave, li, g, pmark); break;
        default: dnp_stream_item<4>(a, c, bs, v0, sp, Op, Mp, wave, li, g, pmark); break;
      }
    }
    __syncthreads();
    if (t < a.NT) __hip_atomic_fetch_add(&a.sync[DNP_XDONE0 + 16 * (v0 + t)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == 0 && a.prof && (int)blockIdx.x == U && nprof < 32) { a.prof[nprof * 3 + 2] = wall_clock64(); ++nprof; }
  }
}

// what the self-attention of a multi-tile trajectory needs beyond the unit's own LDS tiles
struct DnpTiles {
  float* kvx;                    // this sample's [2][NT * 16][2E] exchange buffers
  int* kvdone;                   // this sample's counter
  int* abort_flag;
  const unsigned char* kmask;    // this sample's [Lfull] key-padding mask or NULL
  int NT, tile, Lfull, target, parity, spin_limit;
};

// the layer remainder's operation loop (the body of dn_rest_loop_kernel's table walk, without the phase probe)
__device__ __forceinline__ bool dnp_run_ops(float* smem, const DnOpTable& tab, int L, int D, int E, int H, const DnpTiles& tl) {
#pragma nounroll
  for (int i = 0; i < tab.n; ++i) {
    const DnOp& op = tab.op[i];
    switch (op.type) {
      case DN_OP_LINEAR: {
        const bool vec = ((op.c & 3) == 0) && ((((uintptr_t)op.W) & 15) == 0);
        const int act = op.g >> 16, ldy = op.g & 0xFFFF;
        if (vec) wg_linear_rt<true>(smem + op.a, op.b, op.c, op.W, op.c, op.d >= 0 ? smem + op.d : nullptr, op.e, smem + op.f, ldy, act);
        else wg_linear_rt<false>(smem + op.a, op.b, op.c, op.W, op.c, op.d >= 0 ? smem + op.d : nullptr, op.e, smem + op.f, ldy, act);
        break;
      }
      case DN_OP_ADDLN:
        wg_add_layernorm(smem + op.a, LDX, smem + op.b, LDX, smem + op.c, smem + op.d, smem + op.e, LDX, E);
        break;
      case DN_OP_ADALN:
        wg_adaln(smem + op.a, LDX, op.b >= 0 ? smem + op.b : nullptr, op.c >= 0 ? smem + op.c : nullptr, smem + op.d, LDX, L, E,
                 op.e >= 0 ? smem + op.e : nullptr, LDX);
        break;
      case DN_OP_ROPE:
        wg_rope(smem + op.a, LDQK, 0, 2, smem + op.b, D, op.c >= 0 ? smem + op.c : nullptr, L, E, 1.0f / sqrtf((float)HD));
        break;
      default: {               // DN_OP_ATTN   a: Q, b: K, c: V, d: mask (floats), e: O
        const float* mk = smem + op.d;
        const float* Q = smem + op.a;
        const float* Kk = smem + op.b;
        const float* V = smem + op.c;
        float* O = smem + op.e;
        const int r = (threadIdx.x >> 4) & 15, h = threadIdx.x & 15;
        const bool worker = threadIdx.x < 256 && h < H;
        float q[HD], acc[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) { q[d] = worker ? Q[r * LDQK + h * HD + d] : 0.f; acc[d] = 0.f; }
        float m = -INFINITY, l = 0.f;
        if (tl.NT == 1) {
          if (worker) {
            for (int s2 = 0; s2 < L; ++s2) {
              if (mk[s2] != 0.f) continue;
              float sc = 0.f;
#pragma unroll
              for (int d = 0; d < HD; ++d) sc += q[d] * Kk[s2 * LDQK + h * HD + d];
              const float mn = fmaxf(m, sc);
              const float al = __expf(m - mn), pw = __expf(sc - mn);
              l = l * al + pw;
#pragma unroll
              for (int d = 0; d < HD; ++d) acc[d] = acc[d] * al + pw * V[s2 * LDH + h * HD + d];
              m = mn;
            }
          }
        } else {
          // ---- multi-tile trajectory: the keys are all L rows of the SAMPLE.  Publish this tile's rotated keys / values, wait for
          // the sample's other tiles, then walk the tiles (staged through two idle row tiles) with one running softmax state
          float* Kc = smem + 2 * DR * LDX;                 // Bs: the value stream's AdaLN output, consumed by the v projection
          float* Vc = smem + 3 * DR * LDX;                 // Ts: written next by the out projection
          const int ldkv = 2 * E;
          float* mine = tl.kvx + ((size_t)tl.parity * tl.NT * 16 + (size_t)tl.tile * 16) * ldkv;
          for (int i = threadIdx.x; i < DR * E; i += blockDim.x) {
            const int rr = i / E, c = i - rr * E;
            dnp_st(&mine[rr * ldkv + c], Kk[rr * LDQK + c]);
            dnp_st(&mine[rr * ldkv + E + c], V[rr * LDH + c]);
          }
          __syncthreads();
          int* ok = reinterpret_cast<int*>(smem + 4 * DR * LDX);       // q | k tile's first word: q is in registers, k is published
          if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(tl.kvdone, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *ok = dnp_wait_ge(tl.kvdone, tl.target, tl.abort_flag, tl.spin_limit) ? 1 : 0;
          }
          __syncthreads();
          if (*ok == 0) return false;
          __syncthreads();
          for (int tt = 0; tt < tl.NT; ++tt) {
            const float* src = tl.kvx + ((size_t)tl.parity * tl.NT * 16 + (size_t)tt * 16) * ldkv;
            for (int i = threadIdx.x; i < DR * E; i += blockDim.x) {
              const int rr = i / E, c = i - rr * E;
              Kc[rr * LDX + c] = dnp_ld(src + rr * ldkv + c);
              Vc[rr * LDX + c] = dnp_ld(src + rr * ldkv + E + c);
            }
            __syncthreads();
            const int nk = min(16, tl.Lfull - tt * 16);
            if (worker) {
              for (int s2 = 0; s2 < nk; ++s2) {
                if (tl.kmask && tl.kmask[tt * 16 + s2]) continue;
                float sc = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) sc += q[d] * Kc[s2 * LDX + h * HD + d];
                const float mn = fmaxf(m, sc);
                const float al = __expf(m - mn), pw = __expf(sc - mn);
                l = l * al + pw;
#pragma unroll
                for (int d = 0; d < HD; ++d) acc[d] = acc[d] * al + pw * Vc[s2 * LDX + h * HD + d];
                m = mn;
              }
            }
            __syncthreads();
          }
        }
        if (worker) {
          const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
          for (int d = 0; d < HD; ++d) O[r * LDX + h * HD + d] = acc[d] * inv;
        }
        __syncthreads();
        break;
      }
    }
  }
  return true;
}

constexpr int DNP_LDS_FLOATS = DR * (4 * LDX + LDQK + LDH) + DN_PS + 2 * DR * LDX + 256 + 16 + 160;

// ---- sample role
__device__ __forceinline__ void dnp_sample_role(const DnPersist* ap, float* smem) {
  const DnPersist& a = dnp_args(ap);
  float* smem0 = smem;
  float *Xs, *As, *Bs, *Ts, *QK, *Hs, *Ps, *Xt, *Pf, *Tr;
  int* shi;
  DnOpTable* tab;
  const int b = blockIdx.x, t = threadIdx.x;                  // b: the STREAMING UNIT id v = role * U0 + u
  const int NT = a.NT, U0 = a.B * NT, U = 2 * U0;
  const int role = b >= U0 ? 1 : 0, u = b - role * U0;        // role 0: primary (head, trajectory + position stack, tail); 1: helper (rotation stack)
  const int bs = u / NT, tile = u - bs * NT, r0 = tile * 16, Lf = a.L;
  const int L0 = min(16, Lf - r0), D0 = a.D, E0 = a.E, H0 = a.H;      // L (below): the unit's own rows
  int L, D, E, H, Epad;
  // LDS map + shape scalars, re-derived from opaque values at every phase (see dnp_opaque)
#define DNP_REFRESH()                                                                                                        \
  do {                                                                                                                       \
    smem = smem0 + dnp_opaque(0);                                                                                            \
    Xs = smem; As = Xs + DR * LDX; Bs = As + DR * LDX; Ts = Bs + DR * LDX;                                                   \
    QK = smem + 4 * DR * LDX; Hs = QK + DR * LDQK; Ps = Hs + DR * LDH;                                                       \
    Xt = Ps + DN_PS;              /* x after the trajectory stack (start of the position and the rotation stack) */        \
    Pf = Xt + DR * LDX;           /* position features */                                                                   \
    Tr = Pf + DR * LDX;           /* [16][16] the sample's trajectory rows (D <= 16 channels) */                            \
    shi = reinterpret_cast<int*>(Tr + 256);                                                                                  \
    tab = reinterpret_cast<DnOpTable*>(Tr + 256 + 16);                                                                       \
    L = dnp_opaque(L0); D = dnp_opaque(D0); E = dnp_opaque(E0); H = dnp_opaque(H0); Epad = (E + 15) & ~15;                   \
  } while (0)
  DNP_REFRESH();
  const int NL = a.n_traj + a.n_pos + a.n_rot;
  const int nse = a.nsplit * a.nsub;
  const float* Op = a.part;
  const float* Mp = a.part + (size_t)nse * U * a.H * 256;
  int* abort_flag = a.sync + 2;
  int sa_count = 0;                                           // self-attention blocks this workgroup has been through (all steps)
  int my_layers = 0;                                          // layers this workgroup has published (all steps)
  const int l_beg = role ? a.n_traj + a.n_pos : 0, l_end = role ? NL : a.n_traj + a.n_pos;
  float* xb_traj = a.xbuf + ((size_t)u * 2 + 0) * 16 * 128;   // x after the trajectory stack
  float* xb_rot = a.xbuf + ((size_t)u * 2 + 1) * 16 * 128;    // rotation features
  int* xt_flag = a.sync + dnp_xtready0(a.B, NT) + 16 * u;
  int* rd_flag = a.sync + dnp_rdone0(a.B, NT) + 16 * u;
  long long* const prof = a.prof;
#define DNP_MARK(slot) do { if (prof && u == 0 && t == 0 && step == 1) prof[slot] = wall_clock64(); } while (0)
  for (int i = t; i < 256; i += blockDim.x) {
    const int r = i >> 4, c = i & 15;
    Tr[i] = (r < L && c < D) ? a.traj[((size_t)bs * Lf + r0 + r) * D + c] : 0.f;
  }
  for (int i = t; i < 4 * DR * LDX; i += blockDim.x) smem[i] = 0.f;          // pads of the four row tiles
  __syncthreads();
  for (int step = 0; step < a.nsteps; ++step) {
    const int t_step = a.t_first - step;
    DNP_REFRESH();
    if (role == 0) DNP_MARK(250);
    if (role == 1) {
      // ================= helper: take x (after the trajectory stack) and the step's trajectory rows over from the primary
      if (t == 0) shi[0] = dnp_wait_ge(xt_flag, step + 1, abort_flag, a.spin_limit) ? 1 : 0;
      __syncthreads();
      if (shi[0] == 0) return;
      for (int i = t; i < DR * 128; i += blockDim.x) {
        const int r = i >> 7, c = i & 127;
        Xs[r * LDX + c] = dnp_ld(xb_traj + i);
      }
      for (int i = t; i < 256; i += blockDim.x) {
        const int r = i >> 4, c = i & 15;
        Tr[i] = (r < L && c < D) ? dnp_ld(&a.traj[((size_t)bs * Lf + r0 + r) * D + c]) : 0.f;
      }
      __syncthreads();
    } else
    // ================= head: trajectory encoder [+ attention over the instruction tokens]   (dn_head_kernel)
    {
      const a3d_dn_head_params& p = dnp_args(ap).head;
      float* Qs = Bs;
      float* kvS = QK;                       // [S_lang][2E] over the idle QK | Hs | Ps area
      for (int i = t; i < DR * 16; i += blockDim.x) As[(i >> 4) * LDX + (i & 15)] = Tr[i];
      wg_zero_pad(Ts, LDX, E, Epad);
      wg_zero_pad(Xs, LDX, E, Epad);
      wg_zero_pad(Qs, LDX, E, Epad);
      __syncthreads();
      wg_linear<1>(As, LDX, D, p.enc_w0, D, p.enc_b0, E, Ts, LDX);
      wg_linear<0>(Ts, LDX, E, p.enc_w1, E, p.enc_b1, E, Xs, LDX);
      if (p.lang_kv) {
        wg_zero_pad(As, LDX, E, Epad);
        wg_adaln(Xs, LDX, p.sem + (size_t)r0 * E, nullptr, As, LDX, L, E);
        wg_linear<0>(As, LDX, E, p.q_w, E, p.q_b, E, Qs, LDX);
        const float scale = 1.0f / sqrtf((float)HD);
        for (int i = t; i < DR * E; i += blockDim.x) Qs[(i / E) * LDX + i % E] *= scale;
        __syncthreads();
        // instruction k | v rows -> LDS: float4 loads, eight in flight per thread (the plain element loop issued one dependent
        // global load per iteration: 25 round trips, most of the 56 us the phase probe showed for the head)
        {
          const float4* kv4 = reinterpret_cast<const float4*>(p.lang_kv + (size_t)bs * p.S_lang * 2 * E);      // rows of 2E floats: 16-byte aligned
          const int n4 = (p.S_lang * 2 * E) >> 2;
          for (int i0 = 0; i0 < n4; i0 += 8 * (int)blockDim.x) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int i = i0 + u * (int)blockDim.x + t;
              v[u] = i < n4 ? kv4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int i = i0 + u * (int)blockDim.x + t;
              if (i < n4) reinterpret_cast<float4*>(kvS)[i] = v[u];
            }
          }
        }
        __syncthreads();
        wg_small_attention4(Qs, LDX, kvS, 2 * E, kvS + E, 2 * E, p.S_lang, H, As, LDX);
        wg_linear<0>(As, LDX, E, p.out_w, E, p.out_b, E, Ts, LDX);
        wg_add_layernorm(Xs, LDX, Ts, LDX, p.ln_g, p.ln_b, Xs, LDX, E);
      }
    }
    // ================= layers
    for (int l = l_beg; l < l_end; ++l) {
      DNP_REFRESH();
      DNP_MARK(96 + 7 * l + 0);
      if (role == 0 && l == a.n_traj) {                           // end of the trajectory stack: hand x to the helper (rotation stack)
        for (int i = t; i < DR * 128; i += blockDim.x) {
          const int r = i >> 7, c = i & 127;
          dnp_st(&xb_traj[i], c < E ? Xs[r * LDX + c] : 0.f);
        }
        __syncthreads();
        if (t == 0) __hip_atomic_store(xt_flag, step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // ---- (1) publish the layer's queries: rope(W_q AdaLN(x + index embedding) + b_q) * d^-1/2, all heads
      {
        const a3d_dn_cross_params& c = dnp_args(ap).layers[l].c;
        wg_zero_pad(As, LDX, E, Epad);
        wg_zero_pad(Ts, LDX, E, Epad);
        wg_adaln(Xs, LDX, c.sem ? c.sem + (size_t)r0 * E : nullptr, c.mod ? c.mod + (size_t)t_step * 2 * E : nullptr, As, LDX, L, E);
        wg_linear<0>(As, LDX, E, c.q_w, E, c.q_b, E, Ts, LDX);
        wg_rope(Ts, LDX, 0, 1, Tr, 16, c.freq, L, E, 1.0f / sqrtf((float)HD));
        DNP_MARK(96 + 7 * l + 1);
        float* qrow = a.qbuf + (size_t)b * 16 * 128;
        for (int i = t; i < 16 * 128; i += blockDim.x) {
          const int r = i >> 7, cc = i & 127;
          dnp_st(&qrow[i], (r < L && cc < E) ? Ts[r * LDX + cc] : 0.f);
        }
        __syncthreads();
        if (t == 0) {
          // the NT row tiles of a (role, sample) share one K / V pass: whoever publishes last queues the group's item.  (The tiles
          // of a sample advance layer by layer together -- each needs the group's item of layer l to finish layer l -- so the
          // k-th multiple of NT is the k-th layer of this role's stacks.)
          const int grp = role * a.B + bs;
          const int cnt = __hip_atomic_fetch_add(&a.sync[dnp_grp0(a.B, NT) + 16 * grp], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
          if (cnt % NT == 0) {
            const int slot = atomicAdd(&a.sync[1], 1);
            __hip_atomic_store(&a.sync[dnp_queue0(a.B, NT) + slot], 1 + l * (2 * a.B) + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      DNP_MARK(96 + 7 * l + 2);
      DNP_REFRESH();
      // ---- (2) while the streamers work: this layer's vectors -> LDS, weights touched in L2, pads, operation table
      {
        const a3d_dn_rest_params& p = dnp_args(ap).layers[l].r;
        const float* s_mod = p.s_mod ? p.s_mod + (size_t)t_step * 2 * E : nullptr;
        const float* f_mod = p.f_mod ? p.f_mod + (size_t)t_step * 2 * E : nullptr;
        float* const vd[12] = {Ps, Ps + 128, Ps + 256, Ps + 384, Ps + 640, Ps + 1024, Ps + 1152, Ps + 1280, Ps + 1408, Ps + 1664, Ps + 2176,
                               Ps + 2304};
        const VecList vl = {{p.c_out_b, p.c_ln_g, p.c_ln_b, p.s_in_w ? s_mod : nullptr, p.s_in_w ? p.s_in_b : nullptr,
                             p.s_in_w ? p.s_out_b : nullptr, p.s_in_w ? p.s_ln_g : nullptr, p.s_in_w ? p.s_ln_b : nullptr,
                             p.f_w1 ? f_mod : nullptr, p.f_w1 ? p.f_b1 : nullptr, p.f_w1 ? p.f_b2 : nullptr, p.f_w1 ? p.f_ln_g : nullptr},
                            {E, E, E, 2 * E, 3 * E, E, E, E, 2 * E, p.F, E, E}};
        wg_stage_vectors(vl, vd, (p.sem && p.s_in_w) ? p.sem + (size_t)r0 * E : nullptr, L * E, Ps + 2560);
        float* const misc = Ps + 2560 + DR * 128;
        if (p.f_w1 && t < E) Ps[2432 + t] = p.f_ln_b[t];
        if (p.s_in_w) {
          if (t < L * D) misc[t] = Tr[(t / D) * 16 + t % D];
          if (p.freq && t < E / 6) misc[160 + t] = p.freq[t];
          if (t < DR) misc[192 + t] = (p.kmask && t < L && p.kmask[(size_t)bs * Lf + r0 + t]) ? 1.f : 0.f;
        }
        const WarmList wl = {{p.c_out_w, p.s_in_w, p.s_in_w ? p.s_out_w : nullptr, p.f_w1, p.f_w1 ? p.f_w2 : nullptr, nullptr},
                             {E * E, 3 * E * E, E * E, p.F * E, p.F * E, 0}};
        wg_warm_l2(wl, U0);
        if (t == 0) {
          a3d_dn_rest_params pl = p;
          pl.s_mod = s_mod;
          pl.f_mod = f_mod;
          dn_build_ops(*tab, pl, E);
        }
        wg_zero_pad(Bs, LDX, E, Epad);
        if (p.f_w1) {
          const int padw = ((p.F + 15) & ~15) - p.F;
          for (int i = t; i < DR * padw; i += blockDim.x) Hs[(i / padw) * LDH + p.F + i % padw] = 0.f;
        }
      }
      DNP_MARK(96 + 7 * l + 3);
      DNP_REFRESH();
      // ---- (3) wait for the layer's nsplit items of this sample
      if (t == 0) shi[0] = dnp_wait_ge(&a.sync[DNP_XDONE0 + 16 * b], (my_layers + 1) * a.nsplit, abort_flag, a.spin_limit) ? 1 : 0;
      __syncthreads();
      if (shi[0] == 0) return;
      DNP_MARK(96 + 7 * l + 4);
      // ---- (4) combine the key splits -> As
      wg_zero_pad(As, LDX, E, Epad);
      wg_zero_pad(Ts, LDX, E, Epad);
      // one thread per (query row, head, channel quad): 16 x 8 x 4 = the workgroup's 512 threads.  The partials of all key splits
      // are fetched with INDEPENDENT loads (fixed trip count, predicated, eight splits per batch) -- a loop over a run-time split
      // count issues them one L2 round trip after the other, and this sits on the sample's critical path (measured: 8 splits cost
      // 0.15 ms per denoise step more than 4 with the serial loop)
      {
        constexpr int NSE_MAX = 16;
        const int r = t >> 5, h = (t >> 2) & 7, q = t & 3;
        const bool on = h < H;
        const size_t stride = (size_t)U * H * 16;                   // rows per split
        const size_t row0 = ((size_t)b * H + (on ? h : 0)) * 16 + r;
        float ms[NSE_MAX];
#pragma unroll
        for (int s = 0; s < NSE_MAX; ++s) ms[s] = (on && s < nse) ? dnp_ld(&Mp[row0 + s * stride]) : -INFINITY;
        float m = -INFINITY;
#pragma unroll
        for (int s = 0; s < NSE_MAX; ++s) m = fmaxf(m, ms[s]);
        const float m_use = (m == -INFINITY) ? 0.f : m;
        float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s0 = 0; s0 < NSE_MAX; s0 += 8) {
          float4 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (on && s0 + u < nse) {
              const float* src = &Op[(row0 + (s0 + u) * stride) * 16 + 4 * q];
              v[u] = make_float4(dnp_ld(src), dnp_ld(src + 1), dnp_ld(src + 2), dnp_ld(src + 3));
            } else {
              v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float w = (on && s0 + u < nse) ? __expf(ms[s0 + u] - m_use) : 0.f;
            acc4.x += w * v[u].x; acc4.y += w * v[u].y; acc4.z += w * v[u].z; acc4.w += w * v[u].w;
          }
        }
        const float den = __shfl(acc4.w, (t & 63) | 3, 64);          // channel 15 = sum_k p lives in the quad's last thread
        const float inv = den > 0.f ? 1.0f / den : 0.f;
        if (on) {
          float* dst = &As[r * LDX + h * HD + 4 * q];
          dst[0] = acc4.x * inv;
          dst[1] = acc4.y * inv;
          dst[2] = acc4.z * inv;
          if (q < 3) dst[3] = acc4.w * inv;
        }
      }
      __syncthreads();
      DNP_MARK(96 + 7 * l + 5);
      DNP_REFRESH();
      // ---- (5) out-proj + LayerNorm, self-attention block, FFN block
      {
        const a3d_dn_rest_params& p = dnp_args(ap).layers[l].r;
        const bool has_sa = p.s_in_w != nullptr;
        DnpTiles tl;
        tl.kvx = a.kvx ? a.kvx + ((size_t)role * a.B + bs) * 2 * NT * 16 * 2 * E : nullptr;
        tl.kvdone = a.sync + dnp_kvdone0(a.B, NT) + 16 * (role * a.B + bs);
        tl.abort_flag = abort_flag;
        tl.kmask = p.kmask ? p.kmask + (size_t)bs * Lf : nullptr;
        tl.NT = NT; tl.tile = tile; tl.Lfull = Lf; tl.target = NT * (sa_count + 1); tl.parity = sa_count & 1;
        tl.spin_limit = a.spin_limit;
        if (!dnp_run_ops(smem, *tab, L, D, E, H, tl)) return;
        if (has_sa) ++sa_count;
        ++my_layers;
      }
      DNP_MARK(96 + 7 * l + 6);
    }
    DNP_REFRESH();
    if (role == 1) {
      // ================= helper: the rotation features back to the primary
      for (int i = t; i < DR * 128; i += blockDim.x) {
        const int r = i >> 7, c = i & 127;
        dnp_st(&xb_rot[i], c < E ? Xs[r * LDX + c] : 0.f);
      }
      __syncthreads();
      if (t == 0) __hip_atomic_store(rd_flag, step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    DNP_MARK(251);
    // the rotation features (helper) -> Xt; the position features stay in Xs
    if (t == 0) shi[0] = dnp_wait_ge(rd_flag, step + 1, abort_flag, a.spin_limit) ? 1 : 0;
    __syncthreads();
    if (shi[0] == 0) return;
    for (int i = t; i < DR * 128; i += blockDim.x) {
      const int r = i >> 7, c = i & 127;
      Xt[r * LDX + c] = dnp_ld(xb_rot + i);
    }
    __syncthreads();
    DNP_MARK(253);
    // ================= tail: regressors, trajectory update, DDPM reverse step   (dn_tail_kernel; Xs = position, Xt = rotation features)
    {
      const a3d_dn_tail_params& p = dnp_args(ap).tail;
      float* Us = QK;                        // [16][16]
      wg_zero_pad(Ts, LDX, E, Epad);
      __syncthreads();
      wg_linear<1>(Xs, LDX, E, p.pos_w0, E, p.pos_b0, E, Ts, LDX);
      wg_linear<0>(Ts, LDX, E, p.pos_w1, E, p.pos_b1, 3, Us, 16);
      wg_linear<1>(Xt, LDX, E, p.rot_w0, E, p.rot_b0, E, Ts, LDX);
      wg_linear<0>(Ts, LDX, E, p.rot_w1, E, p.rot_b1, D - 3, Us + 3, 16);
      for (int i = t; i < L * D; i += blockDim.x) {
        const int r = i / D, c = i - r * D;
        const size_t gi = ((size_t)bs * Lf + r0 + r) * D + c;
        const float old = Tr[r * 16 + c];
        float mo = Us[r * 16 + c] + (c < 3 ? old : 0.f);
        if (p.cond_mask && p.cond_mask[gi]) mo = p.cond_data[gi];
        float out = mo;
        if (t_step > 0) {
          const float* cf = ((c < 3) ? p.coef_pos : p.coef_rot) + (size_t)t_step * 3;
          const float x0 = fminf(fmaxf(mo, -1.0f), 1.0f);
          out = cf[0] * x0 + cf[1] * old;
          if (p.noise) out += cf[2] * p.noise[(size_t)t_step * a.B * Lf * D + gi];
        }
        Tr[r * 16 + c] = out;
        dnp_st(&a.traj[gi], out);                    // the helper workgroup reads the rows next step
      }
      __syncthreads();
      DNP_MARK(252);
    }
  }
}

// The argument block lives in device memory (behind the sync words) and is read field by field with scalar loads: passed by value
// its ~120 SGPRs of pointers stay live across both roles and the register allocator spills hundreds of them through VGPR lanes
// (measured at compile time: 884 SGPR + 974 VGPR spills, 3.5 KB of scratch per lane).  A one-wave kernel writes the block first.
// It also zeroes the synchronisation words (ticket, queue tail, abort word, completion counters, ready queue).  A captured
// hipMemsetAsync did that first: eager launches and the FIRST replay of a captured graph were right, every later replay ran on
// stale counters (wrong trajectories, then a memory fault at 100 steps; profiles/r05_persist_graph_probe.txt) -- a kernel node
// replays like every other launch of this library.
__global__ __launch_bounds__(256) void dn_persist_args_kernel(DnPersist a, DnPersist* dst, int* sync, int words) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) sync[i] = 0;
  if (blockIdx.x == 0) {
    const int n = (int)(sizeof(DnPersist) / sizeof(int));
    const int* src = reinterpret_cast<const int*>(&a);
    for (int i = threadIdx.x; i < n; i += blockDim.x) reinterpret_cast<int*>(dst)[i] = src[i];
  }
}
// An aborted launch (a co-resident workgroup never arrived: fewer free CUs than the grid assumes -- CU masking, another process on
// the device) must not hand back a plausible-looking trajectory: this node follows the persistent launch and turns the whole batch
// into NaN when the abort word is set.  In-band, no host synchronisation, replays with the graph.
__global__ __launch_bounds__(256) void dn_persist_poison_kernel(float* __restrict__ traj, int n, const int* __restrict__ sync) {
  if (sync[2] == 0) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) traj[i] = __builtin_nanf("");
}
__global__ __launch_bounds__(512) void dn_persist_kernel(const DnPersist* __restrict__ ap, int U) {      // U: sample-role workgroups (2 per unit)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < U) dnp_sample_role(ap, smem);
  else dnp_stream_role(ap, smem);
}

}  // namespace a3d

using namespace a3d;

// threads per sample-workgroup of the per-sample kernels: 512 (8 waves) by default, A3D_DN_THREADS=256 for the A/B run
// (measured on MI355X, cfg-3: 1.125 -> 0.955 ms per denoise step; 16 waves gave nothing more: 0.949)
static int dn_threads() {
  static const int n = (getenv("A3D_DN_THREADS") && atoi(getenv("A3D_DN_THREADS")) == 256) ? 256 : 512;
  return n;
}

// L2 warm-up of the weights at the head of the per-sample kernels (wg_warm_l2); A3D_DN_WARM=0 for the A/B run
static int dn_warm() {
  static const int w = ((getenv("A3D_DN_WARM") && atoi(getenv("A3D_DN_WARM")) == 0) ? 0 : 1) |
                       ((getenv("A3D_DN_PROF") && atoi(getenv("A3D_DN_PROF")) != 0) ? 2 : 0);
  return w;
}

// 18 phase timestamps (100 MHz ticks) of workgroup 0 of the last a3d_dn_rest launch made with A3D_DN_PROF=1
extern "C" int a3d_dbg_dn_prof(long long* out18) {
  if (!out18) { set_error("a3d_dbg_dn_prof: null pointer"); return A3D_ERR_ARG; }
  hipError_t e = hipMemcpyFromSymbol(out18, HIP_SYMBOL(g_dn_prof), 18 * sizeof(long long));
  if (e != hipSuccess) { set_error("a3d_dbg_dn_prof: %s", hipGetErrorString(e)); return A3D_ERR_LAUNCH; }
  return A3D_OK;
}

static int dn_check(const char* fn, int B, int L, int E, int H) {
  if (B <= 0 || L <= 0 || L > DR || E <= 0 || E > 128 || (E % 6) != 0 || H * HD != E) {
    set_error("%s: bad shape (B=%d L=%d E=%d H=%d; L <= 16, E = 15 H <= 128)", fn, B, L, E, H);
    return A3D_ERR_ARG;
  }
  return A3D_OK;
}

extern "C" int a3d_dn_head(const float* traj, int D, const a3d_dn_head_params* p, float* x_out, int B, int L, int E, int H,
                           void* stream) {
  int rc = dn_check("a3d_dn_head", B, L, E, H);
  if (rc) return rc;
  if (!traj || !p || !x_out || D <= 0 || D > 16 || !p->enc_w0 || !p->enc_w1 ||
      (p->lang_kv && (!p->q_w || !p->out_w || !p->ln_g || !p->sem || p->S_lang <= 0))) {
    set_error("a3d_dn_head: bad argument");
    return A3D_ERR_ARG;
  }
  const size_t lds = p->lang_kv ? (size_t)p->S_lang * 2 * E * sizeof(float) : 0;
  if (lds + 4 * DR * LDX * sizeof(float) > 150 * 1024) {
    set_error("a3d_dn_head: %d instruction tokens do not fit the LDS staging", p->S_lang);
    return A3D_ERR_ARG;
  }
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)dn_head_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(dn_head_kernel, dim3(B), dim3(dn_threads()), lds, (hipStream_t)stream, traj, D, *p, x_out, L, E, H, dn_warm());
  return check_launch("a3d_dn_head");
}

extern "C" size_t a3d_dn_cross_ws_floats(int B, int H, int nsplit) { return (size_t)nsplit * B * H * 16 * 17; }

extern "C" int a3d_dn_cross(const float* x, const float* traj, int D, const a3d_dn_cross_params* p, float* ws, int B, int L,
                            int E, int H, int S, int Sp, int nsplit, void* stream) {
  int rc = dn_check("a3d_dn_cross", B, L, E, H);
  if (rc) return rc;
  if (!x || !traj || !p || !ws || !p->q_w || !p->q_b || !p->Kf || !p->Vt || S <= 0 || Sp < S || (Sp % 64) != 0 || nsplit < 1 ||
      nsplit > 64 || D < 3) {
    set_error("a3d_dn_cross: bad argument (S=%d Sp=%d nsplit=%d)", S, Sp, nsplit);
    return A3D_ERR_ARG;
  }
  float* Op = ws;
  float* Mp = ws + (size_t)nsplit * B * H * 16 * 16;
  hipLaunchKernelGGL(dn_cross_kernel, dim3(xcd_grid(B * H, nsplit)), dim3(256), 0, (hipStream_t)stream, x, traj, D, *p, Op, Mp,
                     B, L, E, H, S, Sp, nsplit);
  return check_launch("a3d_dn_cross");
}

extern "C" int a3d_dn_rest(const float* x_in, const float* traj, int D, const float* ws, const a3d_dn_rest_params* p,
                           float* x_out, int B, int L, int E, int H, int nsplit, void* stream) {
  int rc = dn_check("a3d_dn_rest", B, L, E, H);
  if (rc) return rc;
  if (!x_in || !traj || !ws || !p || !x_out || !p->c_out_w || !p->c_ln_g || nsplit < 1 || (p->f_w1 && (p->F <= 0 || p->F > 512)) ||
      (p->s_in_w && (!p->s_out_w || !p->s_ln_g))) {
    set_error("a3d_dn_rest: bad argument");
    return A3D_ERR_ARG;
  }
  if (D < 3 || L * D > DN_MISC_XYZ) {      // the staged xyz rows share a 256-float LDS area with freq (at 160) and the key mask (at 192)
    set_error("a3d_dn_rest: L * D = %d trajectory values do not fit the %d-float staging area (D >= 3)", L * D, DN_MISC_XYZ);
    return A3D_ERR_ARG;
  }
  const float* Op = ws;
  const float* Mp = ws + (size_t)nsplit * B * H * 16 * 16;
  DnOpTable tab;
  dn_build_ops(tab, *p, E);
  const size_t lds2 = ((size_t)DR * (4 * LDX + LDQK + LDH) + DN_PS) * sizeof(float);
  static bool attr2 = false;
  if (!attr2) {
    (void)hipFuncSetAttribute((const void*)dn_rest_loop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr2 = true;
  }
  hipLaunchKernelGGL(dn_rest_loop_kernel, dim3(B), dim3(dn_threads()), lds2, (hipStream_t)stream, x_in, traj, D, Op, Mp, *p, tab, x_out, B,
                     L, E, H, nsplit, dn_warm());
  return check_launch("a3d_dn_rest");
}

// ---- persistent sampler: host side
static int dnp_nsub(int H) { return (H <= 8 && 8 % H == 0) ? 8 / H : 1; }

extern "C" int a3d_dn_persist_splits(int H, int nsplit) { return nsplit * dnp_nsub(H); }

static size_t dnp_words(int B, int L, int n_layers, int nsteps) {
  const int NT = (L + DR - 1) / DR;
  return (((size_t)dnp_queue0(B, NT) + (size_t)nsteps * n_layers * B * NT) + 3) & ~(size_t)3;
}

extern "C" size_t a3d_dn_persist_kvx_floats(int B, int L, int E) {
  const int NT = (L + DR - 1) / DR;
  return NT > 1 ? (size_t)2 * B * 2 * NT * 16 * 2 * E : 0;            // two roles (primary / helper workgroups) x two alternating buffers
}

extern "C" size_t a3d_dn_persist_xbuf_floats(int B, int L) { return (size_t)B * ((L + DR - 1) / DR) * 2 * 16 * 128; }

extern "C" size_t a3d_dn_persist_sync_ints(int B, int L, int n_layers, int nsteps) {
  if (B <= 0 || L <= 0 || n_layers <= 0 || nsteps <= 0) return 0;
  const size_t words = dnp_words(B, L, n_layers, nsteps);
  return ((words + 3) & ~(size_t)3) + ((sizeof(DnPersist) + 15) / 16) * 4 + 2 * DNP_PROF_WORDS;      // + the argument block + phase probe
}

// development aid: the DNP_PROF_WORDS phase timestamps (100 MHz ticks) of the last a3d_dn_persist launch made with A3D_DN_PROF=1
// (sync / B / n_layers / nsteps as passed to that launch); host buffer
extern "C" int a3d_dn_persist_prof(const int* sync, int B, int L, int n_layers, int nsteps, long long* out256) {
  if (!sync || !out256) { set_error("a3d_dn_persist_prof: null pointer"); return A3D_ERR_ARG; }
  const size_t words = dnp_words(B, L, n_layers, nsteps);
  const int* src = sync + words + ((sizeof(DnPersist) + 15) / 16) * 4;
  hipError_t e = hipMemcpy(out256, src, DNP_PROF_WORDS * sizeof(long long), hipMemcpyDeviceToHost);
  if (e != hipSuccess) { set_error("a3d_dn_persist_prof: %s", hipGetErrorString(e)); return A3D_ERR_LAUNCH; }
  return A3D_OK;
}

extern "C" int a3d_dn_persist(const a3d_dn_layer_params* layers_dev, int n_traj, int n_pos, int n_rot, const a3d_dn_head_params* head,
                              const a3d_dn_tail_params* tail, float* traj, float* qbuf, float* part, float* kvx, float* xbuf, int* sync,
                              int B, int L, int D, int E, int H, int S, int Sp, int nsplit, int t_first, int nsteps, void* stream) {
  int rc = A3D_OK;
  if (B <= 0 || L <= 0 || L > 4 * DR || E <= 0 || E > 128 || (E % 6) != 0 || H * HD != E) {
    set_error("a3d_dn_persist: bad shape (B=%d L=%d E=%d H=%d; L <= 64, E = 15 H <= 128)", B, L, E, H);
    return A3D_ERR_ARG;
  }
  const int NT = (L + DR - 1) / DR, U = B * NT;
  static_assert(sizeof(a3d_dn_layer_params) == sizeof(DnLayerDev), "layer table layout");
  if (!layers_dev || !head || !tail || !traj || !qbuf || !part || !xbuf || !sync || (NT > 1 && !kvx) || n_traj < 0 || n_pos < 1 || n_rot < 1 || D < 4 ||
      D > 16 || std::min(L, DR) * D > DN_MISC_XYZ || S <= 0 || Sp < S || (Sp % 64) != 0 || nsplit < 1 || nsplit * dnp_nsub(H) > 16 || H > 8 || nsteps < 1 || t_first < nsteps - 1 ||
      !head->enc_w0 || !head->enc_w1 || (head->lang_kv && (!head->q_w || !head->out_w || !head->ln_g || !head->sem || head->S_lang <= 0)) ||
      !tail->pos_w0 || !tail->rot_w0 || !tail->coef_pos || !tail->coef_rot || (tail->cond_mask && !tail->cond_data)) {
    set_error("a3d_dn_persist: bad argument (B=%d L=%d D=%d E=%d H=%d S=%d Sp=%d nsplit=%d stacks %d/%d/%d steps %d from t=%d)", B, L, D, E, H,
              S, Sp, nsplit, n_traj, n_pos, n_rot, nsteps, t_first);
    return A3D_ERR_ARG;
  }
  if (head->lang_kv && (size_t)head->S_lang * 2 * E > (size_t)DR * (LDQK + LDH) + DN_PS) {
    set_error("a3d_dn_persist: %d instruction tokens do not fit the LDS staging", head->S_lang);
    return A3D_ERR_ARG;
  }
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = -1;
  }
  // every workgroup must be resident at once (the roles wait for each other): one workgroup per CU (121 KB of LDS each)
  const int nworkers = n_cu - 2 * U;                  // two sample-role workgroups per unit (primary + rotation-stack helper)
  if (n_cu <= 0 || nworkers < 16) {
    set_error("a3d_dn_persist: %d trajectories x %d row tiles x 2 roles leave %d of %d CUs for the streaming role (>= 16 needed)", B, NT,
              nworkers, n_cu);
    return A3D_ERR_ARG;
  }
  const size_t words = dnp_words(B, L, n_traj + n_pos + n_rot, nsteps);
  DnPersist a;
  a.layers = reinterpret_cast<const DnLayerDev*>(layers_dev);
  a.head = *head;
  a.tail = *tail;
  a.traj = traj; a.qbuf = qbuf; a.part = part; a.sync = sync; a.kvx = kvx; a.xbuf = xbuf;
  a.B = B; a.L = L; a.NT = NT; a.D = D; a.E = E; a.H = H; a.S = S; a.Sp = Sp; a.nsplit = nsplit; a.nsub = dnp_nsub(H);
  a.n_traj = n_traj; a.n_pos = n_pos; a.n_rot = n_rot; a.t_first = t_first; a.nsteps = nsteps;
  a.prof = (dn_warm() & 2) ? reinterpret_cast<long long*>(sync + words + ((sizeof(DnPersist) + 15) / 16) * 4) : nullptr;
  a.spin_limit = 1 << 21;                     // ~2 s of polling: a wait is at most a few milliseconds; beyond it the launch aborts
  if (const char* sl = getenv("A3D_DN_SPIN_LIMIT")) a.spin_limit = atoi(sl);      // test hook: 0 forces the abort path (tests/test_diffusion_gpu.py)
  hipStream_t s = (hipStream_t)stream;
  DnPersist* a_dev = reinterpret_cast<DnPersist*>(sync + words);
  hipLaunchKernelGGL(dn_persist_args_kernel, dim3((unsigned)std::min<size_t>((words + 255) / 256, 256)), dim3(256), 0, s, a, a_dev, sync,
                     (int)words);
  rc = check_launch("a3d_dn_persist(args)");
  if (rc) return rc;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)dn_persist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(dn_persist_kernel, dim3(2 * U + nworkers), dim3(512), (size_t)DNP_LDS_FLOATS * sizeof(float), s, a_dev, 2 * U);
  rc = check_launch("a3d_dn_persist");
  if (rc) return rc;
  hipLaunchKernelGGL(dn_persist_poison_kernel, dim3(std::min(cdiv(B * L * D, 256), 64)), dim3(256), 0, s, traj, B * L * D, sync);
  return check_launch("a3d_dn_persist(poison)");
}

extern "C" int a3d_dn_tail(const float* pos_feats, const float* rot_feats, const float* traj, int D,
                           const a3d_dn_tail_params* p, float* traj_out, int B, int L, int E, int t_step, void* stream) {
  if (!pos_feats || !rot_feats || !traj || !p || !traj_out || B <= 0 || L <= 0 || L > DR || E <= 0 || E > 128 || D < 4 || D > 16 ||
      t_step < 0 || !p->pos_w0 || !p->rot_w0 || !p->coef_pos || !p->coef_rot || (p->cond_mask && !p->cond_data)) {
    set_error("a3d_dn_tail: bad argument");
    return A3D_ERR_ARG;
  }
  hipLaunchKernelGGL(dn_tail_kernel, dim3(B), dim3(dn_threads()), 0, (hipStream_t)stream, pos_feats, rot_feats, traj, D, *p, traj_out, L,
                     E, t_step, dn_warm());
  return check_launch("a3d_dn_tail");
}

extern "C" int a3d_rope_rows_f32(const float* Y, int ldy, const float* xyz, const float* freq, float scale, float* out, int B,
                                 int N, int Npad, int E, int H, void* stream) {
  if (!Y || !out || (xyz && !freq) || B <= 0 || N <= 0 || Npad < N || E <= 0 || H * HD != E || ldy < E) {
    set_error("a3d_rope_rows_f32: bad argument");
    return A3D_ERR_ARG;
  }
  const size_t total = (size_t)B * H * Npad * 16;
  hipLaunchKernelGGL(rope_rows_f32_kernel, dim3((int)std::min<size_t>((total + 255) / 256, 16384)), dim3(256), 0,
                     (hipStream_t)stream, Y, ldy, xyz, freq, scale, out, B, N, Npad, E, H);
  return check_launch("a3d_rope_rows_f32");
}
