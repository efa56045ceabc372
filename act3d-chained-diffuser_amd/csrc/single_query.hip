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

// (The key passes themselves are the wave-local kernels of single_query_wave.hip, round 5.  Their predecessors -- a 64-key tile
// walked in nine barrier-separated phases through LDS, rounds 3 - 4 -- lived in this file as the A3D_SQ_WAVE=0 arm until round 6;
// the A/B is decided (sq_bwd 191 -> 105 us, sq_fwd 90 -> 55 us in the step, profiles/r05_bench_B64.json) and they are gone.  What
// stays here: the key-split combine, the value projection on the weighted mean, and the C entry points.)

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

// the key passes (single_query_wave.hip)
void sqw_launch_fwd(const float* X, const float* xyz, const float* Wk, int ldw, const float* bk, const float* qrot, const float* freq,
                    float* part, int B, int S, int E, int H, int nsplit, hipStream_t s);
void sqw_launch_bwd(const float* X, const float* xyz, const float* Wk, int ldw, const float* bk, const float* qrot, const float* freq,
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
  sqw_launch_fwd(X, xyz, Wk, ldw, bk, qrot, freq, ws, B, S, E, H, nsplit, s);
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
  sqw_launch_bwd(X, xyz, Wk, ldw, bk, qrot, freq, lse, dxbar, cD, dX, wpart, dqp, B, S, E, H, nsplit, acc_dx, s);
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
