// Large-M dense layers on the bf16 matrix pipe at fp32 accuracy ("bf16x3"): Y = act(X W^T + b) for M >= 4096 rows.
//
// Reference: F.linear at multihead_custom_attention.py:246-303,447, layers.py:88-94,313-332, diffusion_head.py:41-49 -- the
// same calls a3d_linear_fwd serves (linear.hip), whose exact-f32 MFMA (v_mfma_f32_16x16x4_f32, 1/16 of the bf16 rate) is what
// bounds the 67 584-row context layers of the diffusion training step (E = 120, F = 480: 25 TF/s, 40 % of that step) while
// their HBM traffic would allow 7x more.  Here every fp32 operand is split into THREE bf16 parts on its way into LDS,
//     x = x_h + x_m + x_l   (8 + 8 + 8 significant bits, exact up to 2^-25 |x|),
// and a 32-deep K step of one 16x16 tile is six v_mfma_f32_16x16x32_bf16 (h.h, h.m, m.h, h.l, l.h, m.m): every product
// term down to 2^-24 of the largest, fp32 accumulation -- the result differs from the fmaf chain by fp32 rounding only
// (tests/test_kernels_gpu.py holds both paths to the same 2e-5 + 1e-5 |ref| bound), at 16 / 6 = 2.7x the matrix rate, which puts
// these layers at their HBM time.  Two parts (three MFMAs) would be 5x but leave 1e-5-class errors; not taken.
//
// Tiling: 256 threads = 4 waves, each 32 rows x 64 columns (2 x 4 MFMA tiles, computed TRANSPOSED -- A = W rows, B = X rows --
// so that a lane owns 4 consecutive output columns of one row: 16-byte stores); workgroup tile 128 x 64 (N <= 64) or 64 x 128.
// One LDS stage of [rows][32] bf16 tiles per part (plane_off swizzle, a3d_common.h), the next K step's fp32 rows prefetched
// into registers while the current one is multiplied.  Workgroups that share X rows (the column blocks of one row block) run
// back to back on one XCD (xcd_decode), so X comes from HBM once.
// w_transposed (dgrad: dX = dY W) reads W[k][n] rows and transposes them while staging (the W tile is small).
// The weight gradient of the same layers (first stage of the two-stage reduction) is at the end of this file.
#include "a3d_common.h"
#include "../../include/act3d_hip.h"
#include <stdlib.h>

namespace a3d {

typedef __attribute__((ext_vector_type(2))) float ls_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 ls_bf16x2;

// (a, b) -> three packed bf16 pairs (low half = a's part, high half = b's)
__device__ __forceinline__ void split3(float a, float b, unsigned int& h, unsigned int& m, unsigned int& l) {
  h = __builtin_bit_cast(unsigned int, __builtin_convertvector((ls_f32x2){a, b}, ls_bf16x2));
  float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xFFFF0000u);
  m = __builtin_bit_cast(unsigned int, __builtin_convertvector((ls_f32x2){ra, rb}, ls_bf16x2));
  ra -= __uint_as_float(m << 16);
  rb -= __uint_as_float(m & 0xFFFF0000u);
  l = __builtin_bit_cast(unsigned int, __builtin_convertvector((ls_f32x2){ra, rb}, ls_bf16x2));
}

template <int WN, bool WT>     // WN waves along N (1: 128 x 64 workgroup tile, 2: 64 x 128)
__global__ __launch_bounds__(256, 2) void linear_split_kernel(
    const float* __restrict__ X, int ldx, const float* __restrict__ W, int ldw, const float* __restrict__ bias,
    float* __restrict__ Y, int ldy, const float* __restrict__ mask, int ldm, int M, int N, int K, int act) {
  constexpr int WM = 4 / WN, BM = 32 * WM, BN = 64 * WN;
  constexpr int XL = BM * 8 / 256, WL = BN * 8 / 256;          // float4 per thread and K step
  __shared__ __attribute__((aligned(16))) unsigned short Xs[3][BM * 32];
  __shared__ __attribute__((aligned(16))) unsigned short Ws[3][BN * 32];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int ncb = (N + BN - 1) / BN, nrb = (M + BM - 1) / BM;
  int rb, cb;
  if (!xcd_decode(ncb, nrb, rb, cb)) return;
  const int m0 = rb * BM, n0 = cb * BN;
  const int ksteps = (K + 31) >> 5;

  // parameters live in one flat buffer at 4-byte granularity (engine.FlatParams): W rows / bias are read with scalar loads
  // unless they happen to be 16-byte aligned (the W tile is the small operand: BN x 32 floats per step)
  const bool w_vec = ((ldw & 3) == 0) && ((((uintptr_t)W) & 15) == 0);
  auto ldw4 = [&](const float* p) {
    if (w_vec) return *reinterpret_cast<const float4*>(p);
    return make_float4(p[0], p[1], p[2], p[3]);
  };
  float4 xr[XL], wr[WL];
  auto load = [&](int ks) {
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int idx = t + i * 256, row = idx >> 3, k = ks * 32 + (idx & 7) * 4;
      const int m = min(m0 + row, M - 1);                        // tail rows are masked at the store
      xr[i] = (k < K) ? *reinterpret_cast<const float4*>(X + (size_t)m * ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int idx = t + i * 256;
      if (!WT) {
        const int row = idx >> 3, k = ks * 32 + (idx & 7) * 4;
        const int n = min(n0 + row, N - 1);
        wr[i] = (k < K) ? ldw4(W + (size_t)n * ldw + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const int kk = idx / (BN / 4), j = idx - kk * (BN / 4);  // W[k][n .. n + 3]
        const int k = ks * 32 + kk, n = min(n0 + j * 4, N - 4);
        wr[i] = (k < K) ? ldw4(W + (size_t)k * ldw + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int idx = t + i * 256, row = idx >> 3, s4 = idx & 7;
      unsigned int h0, m0_, l0, h1, m1, l1;
      split3(xr[i].x, xr[i].y, h0, m0_, l0);
      split3(xr[i].z, xr[i].w, h1, m1, l1);
      const int off = plane_off(row, s4 >> 1) + (s4 & 1) * 4;
      *reinterpret_cast<uint2*>(&Xs[0][off]) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(&Xs[1][off]) = make_uint2(m0_, m1);
      *reinterpret_cast<uint2*>(&Xs[2][off]) = make_uint2(l0, l1);
    }
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int idx = t + i * 256;
      unsigned int h0, m0_, l0, h1, m1, l1;
      split3(wr[i].x, wr[i].y, h0, m0_, l0);
      split3(wr[i].z, wr[i].w, h1, m1, l1);
      if (!WT) {
        const int row = idx >> 3, s4 = idx & 7;
        const int off = plane_off(row, s4 >> 1) + (s4 & 1) * 4;
        *reinterpret_cast<uint2*>(&Ws[0][off]) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(&Ws[1][off]) = make_uint2(m0_, m1);
        *reinterpret_cast<uint2*>(&Ws[2][off]) = make_uint2(l0, l1);
      } else {
        const int kk = idx / (BN / 4), j = idx - kk * (BN / 4);
        const unsigned int hs[4] = {h0 & 0xFFFFu, h0 >> 16, h1 & 0xFFFFu, h1 >> 16};
        const unsigned int ms[4] = {m0_ & 0xFFFFu, m0_ >> 16, m1 & 0xFFFFu, m1 >> 16};
        const unsigned int ls[4] = {l0 & 0xFFFFu, l0 >> 16, l1 & 0xFFFFu, l1 >> 16};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int off = plane_off(j * 4 + e, kk >> 3) + (kk & 7);
          Ws[0][off] = (unsigned short)hs[e];
          Ws[1][off] = (unsigned short)ms[e];
          Ws[2][off] = (unsigned short)ls[e];
        }
      }
    }
  };

  f32x4 acc[4][2];                                               // [tn][tm]: D[n = g * 4 + r][m = li]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  load(0);
  for (int ks = 0; ks < ksteps; ++ks) {
    if (ks > 0) __syncthreads();                                 // the previous step's fragment reads are done
    stage();
    __syncthreads();
    if (ks + 1 < ksteps) load(ks + 1);
    s16x8 xa[3][2], wb[3][4];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) xa[p][tm] = *reinterpret_cast<const s16x8*>(&Xs[p][plane_off(wm * 32 + tm * 16 + li, g)]);
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) wb[p][tn] = *reinterpret_cast<const s16x8*>(&Ws[p][plane_off(wn * 64 + tn * 16 + li, g)]);
    }
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        f32x4 c = acc[tn][tm];                                   // smallest terms first
        c = mfma_bf16_16x16x32(wb[1][tn], xa[1][tm], c);
        c = mfma_bf16_16x16x32(wb[0][tn], xa[2][tm], c);
        c = mfma_bf16_16x16x32(wb[2][tn], xa[0][tm], c);
        c = mfma_bf16_16x16x32(wb[0][tn], xa[1][tm], c);
        c = mfma_bf16_16x16x32(wb[1][tn], xa[0][tm], c);
        c = mfma_bf16_16x16x32(wb[0][tn], xa[0][tm], c);
        acc[tn][tm] = c;
      }
  }

#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
    const int m = m0 + wm * 32 + tm * 16 + li;
    if (m >= M) continue;
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const int n = n0 + wn * 64 + tn * 16 + g * 4;
      if (n >= N) continue;                                      // N % 4 == 0: a lane's four columns are in or out together
      float4 v = make_float4(acc[tn][tm][0], acc[tn][tm][1], acc[tn][tm][2], acc[tn][tm][3]);
      if (bias) { v.x += bias[n]; v.y += bias[n + 1]; v.z += bias[n + 2]; v.w += bias[n + 3]; }
      if (act == 1) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      } else if (act == 2) {
        const float4 mk = *reinterpret_cast<const float4*>(mask + (size_t)m * ldm + n);
        v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
      } else if (act == 3) {                                     // accumulate into Y (a gradient summed in place)
        const float4 old = *reinterpret_cast<const float4*>(Y + (size_t)m * ldy + n);
        v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
      }
      *reinterpret_cast<float4*>(Y + (size_t)m * ldy + n) = v;
    }
  }
}

// true if a3d_linear_fwd should take this path (linear.hip asks); A3D_LINEAR_SPLIT=0 keeps the exact-f32 MFMA kernel (A/B)
bool linear_split_applicable(const float* X, int ldx, const float* W, int ldw, const float* bias, const float* Y, int ldy,
                             const float* mask, int ldm, int M, int N, int K, int act) {
  static const bool on = !(getenv("A3D_LINEAR_SPLIT") && atoi(getenv("A3D_LINEAR_SPLIT")) == 0);
  static const int min_m = getenv("A3D_LINEAR_SPLIT_MIN_M") ? atoi(getenv("A3D_LINEAR_SPLIT_MIN_M")) : 4096;
  if (!on || M < min_m || K < 32 || N < 32 || (K & 3) || (N & 3) || ((ldx | ldy) & 3)) return false;
  if ((((uintptr_t)X | (uintptr_t)Y) & 15) != 0) return false;          // activations: torch allocations, always aligned
  if (act == 2 && (!mask || (ldm & 3) || (((uintptr_t)mask) & 15) != 0)) return false;
  return true;
}

int linear_split_launch(const float* X, int ldx, const float* W, int ldw, const float* bias, float* Y, int ldy, const float* mask,
                        int ldm, int M, int N, int K, int act, int w_transposed, hipStream_t s) {
  const bool wide = N > 64;
  const int BM = wide ? 64 : 128, BN = wide ? 128 : 64;
  dim3 grid(xcd_grid(cdiv(M, BM), cdiv(N, BN)));
#define A3D_LS(WNV, WTV) \
  hipLaunchKernelGGL((linear_split_kernel<WNV, WTV>), grid, dim3(256), 0, s, X, ldx, W, ldw, bias, Y, ldy, mask, ldm, M, N, K, act)
  if (wide) { if (w_transposed) A3D_LS(2, true); else A3D_LS(2, false); }
  else { if (w_transposed) A3D_LS(1, true); else A3D_LS(1, false); }
#undef A3D_LS
  return check_launch("a3d_linear_fwd(split)");
}

// ------------------------------------------------------------------------------------------------ weight gradient
// partial[z][n][k] = sum over the rows m of slab z of dY[m][n] X[m][k]  (k == K: the virtual ones column -> db), the first
// stage of linear.hip's two-stage weight gradient (wgrad_reduce_kernel adds the slabs in a fixed order), on the bf16 pipe with
// three-part operands.  The contraction index is the ROW index, which is the slow index of both operands in memory, so both
// tiles are transposed while they are staged: a thread holds the same four columns of two consecutive rows and writes, per
// column and part, one packed pair (m, m + 1) into the [column][32 rows] tile.  64 (n) x 64 (k) outputs per workgroup as in
// linear_wgrad_kernel (same grid, same partial layout); wave w owns n-tile w and all four k-tiles.
__global__ __launch_bounds__(256, 2) void linear_wgrad_split_kernel(
    const float* __restrict__ dY, int lddy, const float* __restrict__ X, int ldx, int has_bias, int M, int N, int K,
    int rows_per_split, float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) unsigned short Ys[3][64 * 32];
  __shared__ __attribute__((aligned(16))) unsigned short Xt[3][64 * 32];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const int mbeg = blockIdx.z * rows_per_split;
  const int mend = min(M, mbeg + rows_per_split);
  const int KE = has_bias ? K + 1 : K;
  const int mp = t >> 4, c4 = (t & 15) * 4;                       // rows 2 mp, 2 mp + 1 of the step; columns c4 .. c4 + 3

  auto ld4 = [&](const float* base, int ld, int m, int c, int C, bool ones) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m >= mend) return v;
    const float* p = base + (size_t)m * ld + c;
    if (c + 3 < C) return *reinterpret_cast<const float4*>(p);
    if (c + 0 < C) v.x = p[0]; else if (ones && c + 0 == C) v.x = 1.f;
    if (c + 1 < C) v.y = p[1]; else if (ones && c + 1 == C) v.y = 1.f;
    if (c + 2 < C) v.z = p[2]; else if (ones && c + 2 == C) v.z = 1.f;
    if (c + 3 < C) v.w = p[3]; else if (ones && c + 3 == C) v.w = 1.f;
    return v;
  };
  float4 yr[2], xr[2];
  auto load = [&](int mb) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      yr[u] = ld4(dY, lddy, mb + 2 * mp + u, n0 + c4, N, false);
      xr[u] = ld4(X, ldx, mb + 2 * mp + u, k0 + c4, K, has_bias != 0);
    }
  };
  auto stage_one = [&](unsigned short (*T)[64 * 32], const float4& r0, const float4& r1) {
    const float a[4] = {r0.x, r0.y, r0.z, r0.w}, b[4] = {r1.x, r1.y, r1.z, r1.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned int h, m, l;
      split3(a[e], b[e], h, m, l);                               // low half: row 2 mp, high half: row 2 mp + 1
      const int off = plane_off(c4 + e, mp >> 2) + (mp & 3) * 2;
      *reinterpret_cast<unsigned int*>(&T[0][off]) = h;
      *reinterpret_cast<unsigned int*>(&T[1][off]) = m;
      *reinterpret_cast<unsigned int*>(&T[2][off]) = l;
    }
  };

  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  load(mbeg);
  for (int mb = mbeg; mb < mend; mb += 32) {
    if (mb > mbeg) __syncthreads();
    stage_one(Ys, yr[0], yr[1]);
    stage_one(Xt, xr[0], xr[1]);
    __syncthreads();
    if (mb + 32 < mend) load(mb + 32);
    s16x8 ya[3], xb[3][4];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      ya[p] = *reinterpret_cast<const s16x8*>(&Ys[p][plane_off(wave * 16 + li, g)]);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) xb[p][kt] = *reinterpret_cast<const s16x8*>(&Xt[p][plane_off(kt * 16 + li, g)]);
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x4 c = acc[kt];
      c = mfma_bf16_16x16x32(ya[1], xb[1][kt], c);
      c = mfma_bf16_16x16x32(ya[0], xb[2][kt], c);
      c = mfma_bf16_16x16x32(ya[2], xb[0][kt], c);
      c = mfma_bf16_16x16x32(ya[0], xb[1][kt], c);
      c = mfma_bf16_16x16x32(ya[1], xb[0][kt], c);
      c = mfma_bf16_16x16x32(ya[0], xb[0][kt], c);
      acc[kt] = c;
    }
  }
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int k = k0 + kt * 16 + li;
    if (k >= KE) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wave * 16 + g * 4 + r;
      if (n < N) partial[((size_t)blockIdx.z * N + n) * KE + k] = acc[kt][r];
    }
  }
}

bool linear_wgrad_split_applicable(const float* dY, int lddy, const float* X, int ldx, int M) {
  static const bool on = !(getenv("A3D_LINEAR_SPLIT") && atoi(getenv("A3D_LINEAR_SPLIT")) == 0) &&
                         !(getenv("A3D_WGRAD_SPLIT") && atoi(getenv("A3D_WGRAD_SPLIT")) == 0);
  static const int min_m = getenv("A3D_LINEAR_SPLIT_MIN_M") ? atoi(getenv("A3D_LINEAR_SPLIT_MIN_M")) : 4096;
  return on && M >= min_m && ((lddy | ldx) & 3) == 0 && ((((uintptr_t)dY | (uintptr_t)X) & 15) == 0);
}

int linear_wgrad_split_launch(const float* dY, int lddy, const float* X, int ldx, int has_bias, int M, int N, int K, int nsplit,
                              int rows_per_split, float* partial, hipStream_t s) {
  const int KE = has_bias ? K + 1 : K;
  dim3 grid(cdiv(N, 64), cdiv(KE, 64), nsplit);
  hipLaunchKernelGGL(linear_wgrad_split_kernel, grid, dim3(256), 0, s, dY, lddy, X, ldx, has_bias, M, N, K, rows_per_split, partial);
  return check_launch("a3d_linear_wgrad(split)");
}

}  // namespace a3d
