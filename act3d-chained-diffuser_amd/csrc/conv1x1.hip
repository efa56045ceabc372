// 1x1 convolutions of the frozen backbone as a bf16 MFMA GEMM with the neighbouring BatchNorm work folded in
// (CLIP ModifiedResNet bottleneck, model/utils/clip.py:28-43: conv1 / conv3 / downsample are 1x1).
//   y[m][n] = sum_k f(x[m][k]) * w[n][k],   f(x) = relu?(x * in_scale[k] + in_shift[k])  (identity when in_scale == NULL)
//   x [M][K] bf16 (NHWC rows, M = images * H * W), w [N][K] bf16, y [M][N] bf16 (fp32 accumulation, one rounding);
//   optional epilogue: per-workgroup partial (sum, sum of squares) of the ROUNDED outputs per channel, in the layout
//   a3d_bn_finalize reduces ([slab][2][N]) -- the statistics pass of the BatchNorm that follows the convolution.
// Folding BatchNorm-apply + ReLU of the PRODUCER into the A-operand load and the statistics of the CONSUMER's BatchNorm
// into the epilogue removes one read + one write and one read of the activation per fused layer.  It pays where the GEMM is
// HBM-bound -- layers 1 and 2 (M = 2^18 .. 2^20 rows, K <= 256): there MIOpen's convolution already runs at 4.8 - 5.5 TB/s
// (profiles/r04_conv_layers.json), so the only thing to win is the BatchNorm traffic, and the kernel has to match MIOpen's
// bandwidth to keep it.  On the deep layers (K, N up to 2048 at M = 2^14 .. 2^16) the convolution is compute-bound and MIOpen /
// CK reach 0.6 - 1.0 PFLOP/s; the round-2/3 re-staging kernel that served them lost by 3x (profiles/r04_conv1x1_layers.json) and
// was deleted in round 4: those shapes stay on MIOpen.
// Tiling: 256 threads = 4 waves in a WM x WN grid, 64 x 64 outputs per wave (16 MFMA 16x16x32 tiles, computed TRANSPOSED --
// A operand = w rows, B operand = x rows -- so that a lane owns consecutive channels of one row).
#include "a3d_common.h"
#include "../../include/act3d_hip.h"

namespace a3d {

constexpr int C1_BK = 32;

// weight rows in LDS: 64 bytes per row and K step, 16-byte segment g stored at g ^ (bit 1 | bit 4 << 1 of the row): conflict-free
// ds_read_b128 for the PERMUTED row order the fragments use (plane_off, which the activation tile keeps, is 2-way conflicted for it;
// tests/test_conv3x3_layout_cpu.py enumerates both)
__device__ __forceinline__ int c1_woff(int row, int seg) { return row * 32 + ((seg ^ (((row >> 1) & 1) | (((row >> 4) & 1) << 1))) << 3); }

// The workgroup's weight block W[BN][K] is staged ONCE and stays in LDS while the workgroup walks its M tiles -- the first kernel
// re-staged it for every tile and K step, which for K = 64 -> N = 256 is four times the bytes of the activation tile it is
// multiplied with (LDS writes and L2 reads, not HBM, bounded it: 15.5 vs 11.9 ms for the whole backbone in round 3) -- and the
// (tile, K step) pairs of a workgroup form one flat sequence of steps whose activation chunks are fetched D steps ahead into
// registers (~32 KB in flight per workgroup, unconditional clamped loads), one barrier per step, two LDS buffers.
//   LDS: W KS x BN x 64 B (8 - 64 KB) | X 2 x BM x 64 B | the producer's BatchNorm scale / shift (K floats each)
// EP (round 6, the FPN's lateral convolutions -- torchvision FeaturePyramidNetwork.inner_blocks + the top-down add): the epilogue adds
// the convolution's bias (fp32, the first nbias channels) and the nearest-2x-upsampled coarser map top [images][H/2][W/2][N] bf16 to
// the fp32 accumulators before the ONE rounding, y = x w^T + bias + up2(top): the lateral map is never written and re-read by a
// top-down pass (a3d_upsample2_add_fwd: 1.2 GB of traffic for the 128 x 128 level of 256 images).
struct C1Epilogue { const float* bias; int nbias; const unsigned short* top; int H, W; };
template <int WN, int KS, bool EP = false>
__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                             const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                             int in_relu, unsigned short* __restrict__ y, float* __restrict__ partial,
                                                             long long M, int N, C1Epilogue ep) {
  constexpr int WM = 4 / WN, BM = 64 * WM, BN = 64 * WN, K = KS * 32;
  constexpr int XL = BM * 4 / 256;                                // 16-byte segments each thread stages per step: 1, 2 or 4
  constexpr int D = 8 / XL;                                       // steps in flight: 8 x 16 B per thread = 32 KB per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned short smem_c1[];
  unsigned short* Ws = smem_c1;                                   // [KS][BN * 32]
  unsigned short* Xs = Ws + KS * BN * 32;                         // [2][BM * 32]
  float* scS = reinterpret_cast<float*>(Xs + 2 * BM * 32);        // [K] scale | [K] shift
  __shared__ float redS[4][64], redQ[4][64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  const int n0 = blockIdx.y * BN;
  const long long mtiles = (M + BM - 1) / BM;
  for (int i = t; i < KS * BN * 4; i += 256) {
    const int ks = i / (BN * 4), rem = i - ks * (BN * 4), row = rem >> 2, seg = rem & 3;
    *reinterpret_cast<uint4*>(&Ws[ks * BN * 32 + c1_woff(row, seg)]) =
        *reinterpret_cast<const uint4*>(w + (size_t)(n0 + row) * K + ks * C1_BK + seg * 8);
  }
  if (in_scale)
    for (int i = t; i < K; i += 256) { scS[i] = in_scale[i]; scS[K + i] = in_shift[i]; }
  float ssum[4][4], ssq[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[a][r] = 0.f; ssq[a][r] = 0.f; }
  float epb[16];                                                  // EP: the bias of the lane's 16 consecutive channels
  if (EP) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = n0 + wn * 64 + g * 16 + j;
      epb[j] = (ep.bias && c < ep.nbias) ? ep.bias[c] : 0.f;       // pad channels carry no bias
    }
  }
  const long long my_tiles = blockIdx.x < mtiles ? (mtiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const long long total = my_tiles * KS;                          // flat steps: s -> (tile blockIdx.x + (s / KS) gridDim.x, K step s % KS)
  auto load = [&](long long s, uint4 (&r)[XL]) {
    const long long m0 = (blockIdx.x + (s / KS) * gridDim.x) * BM;
    const int ks = (int)(s % KS);
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int idx = t + i * 256, row = idx >> 2, seg = idx & 3;
      const long long m = m0 + row < M ? m0 + row : M - 1;        // clamped: tail rows are masked at the store
      r[i] = *reinterpret_cast<const uint4*>(x + (size_t)m * K + ks * C1_BK + seg * 8);
    }
  };
  const float relu_lo = in_relu ? 0.f : -INFINITY;
  auto stage = [&](int buf, int ks, const uint4 (&r)[XL]) {
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int idx = t + i * 256, row = idx >> 2, seg = idx & 3;
      uint4 v = r[i];
      if (in_scale) {
        const int k0 = ks * C1_BK + seg * 8;
        unsigned int u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float a = __uint_as_float(u[j] << 16) * scS[k0 + 2 * j] + scS[K + k0 + 2 * j];
          float b = __uint_as_float(u[j] & 0xFFFF0000u) * scS[k0 + 2 * j + 1] + scS[K + k0 + 2 * j + 1];
          a = fmaxf(a, relu_lo);
          b = fmaxf(b, relu_lo);
          u[j] = pk_bf16(a, b);                                   // v_cvt_pk_bf16_f32: RNE, the bits of f2bf
        }
        v = make_uint4(u[0], u[1], u[2], u[3]);
      }
      *reinterpret_cast<uint4*>(&Xs[buf * BM * 32 + plane_off(row, seg)]) = v;
    }
  };
  if (total > 0) {
    uint4 xr[D][XL];
#pragma unroll
    for (int j = 0; j < D; ++j) load(j < total ? j : total - 1, xr[j]);
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                              // W, scale / shift staged
    for (long long s0 = 0; s0 < total; s0 += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const long long s = s0 + j;
        if (s >= total) break;                                    // workgroup-uniform
        const int ks = (int)(s % KS), buf = (int)(s & 1);
        stage(buf, ks, xr[j]);
        load(s + D < total ? s + D : total - 1, xr[j]);           // unconditional: D steps ahead (the tail re-fetches the last chunk)
        __syncthreads();                                          // stage(s) visible; every wave is past its reads of step s - 1's other buffer
        s16x8 xa[4], wb[4];
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
          xa[tm] = *reinterpret_cast<const s16x8*>(&Xs[buf * BM * 32 + plane_off(wm * 64 + tm * 16 + li, g)]);
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
          wb[tn] = *reinterpret_cast<const s16x8*>(&Ws[ks * BN * 32 + c1_woff(wn * 64 + (li >> 2) * 16 + tn * 4 + (li & 3), g)]);
        uint4 tv[4][2];                                           // EP: the top map's 16 channels under each of the lane's 4 rows
        if (EP && ks == KS - 1 && ep.top) {                       // issued ahead of the MFMAs whose epilogue consumes them
          const long long m0 = (blockIdx.x + (s / KS) * gridDim.x) * BM;
#pragma unroll
          for (int tm = 0; tm < 4; ++tm) {
            const long long m = m0 + wm * 64 + tm * 16 + li;
            const unsigned int mm = (unsigned int)(m < M ? m : M - 1);
            const unsigned int hq = mm / (unsigned int)ep.W, w_ = mm - hq * (unsigned int)ep.W;
            const unsigned int n_ = hq / (unsigned int)ep.H, h_ = hq - n_ * (unsigned int)ep.H;
            const size_t tr = ((size_t)n_ * (ep.H >> 1) + (h_ >> 1)) * (ep.W >> 1) + (w_ >> 1);
            const uint4* src = reinterpret_cast<const uint4*>(ep.top + tr * N + n0 + wn * 64 + g * 16);
            tv[tm][0] = src[0];
            tv[tm][1] = src[1];
          }
        }
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
          for (int tm = 0; tm < 4; ++tm) acc[tn][tm] = mfma_bf16_16x16x32(wb[tn], xa[tm], acc[tn][tm]);
        if (ks == KS - 1) {
          // tile done: round once; the weight rows were fed to the MFMA permuted (tile tn row i <-> channel (i >> 2) * 16 + tn * 4 +
          // (i & 3) of the wave's 64), so a lane holds 16 CONSECUTIVE channels of its row: 32-byte stores, and the four lane groups
          // of a row write one full 128-byte line (8-byte stores of 4 channels reached 3.25 TB/s on the 64 -> 256 layers against
          // MIOpen's 5.2); statistics of the rounded values
          const long long m0 = (blockIdx.x + (s / KS) * gridDim.x) * BM;
#pragma unroll
          for (int tm = 0; tm < 4; ++tm) {
            const long long m = m0 + wm * 64 + tm * 16 + li;
            const bool ok = m < M;
            unsigned int pk[8];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
              // the software rounding (5 VALU per value + packing) made this epilogue ~770 VALU instructions per 32 MFMA; the hardware
              // pair conversion is RNE with the same bits
#pragma unroll
              for (int r2 = 0; r2 < 2; ++r2) {
                float a0 = acc[tn][tm][2 * r2], a1 = acc[tn][tm][2 * r2 + 1];
                if (EP) {                                        // word 2 tn + r2 of the lane's 16 channels = channels 4 tn + 2 r2, + 1
                  a0 += epb[4 * tn + 2 * r2];
                  a1 += epb[4 * tn + 2 * r2 + 1];
                  if (ep.top) {
                    const uint4 q = tv[tm][tn >> 1];
                    const unsigned int tw = (tn & 1) ? (r2 ? q.w : q.z) : (r2 ? q.y : q.x);
                    a0 += __uint_as_float(tw << 16);
                    a1 += __uint_as_float(tw & 0xFFFF0000u);
                  }
                }
                const unsigned int u = pk_bf16(a0, a1);
                if (!EP && ok) {                                 // (the EP instances carry no statistics: partial is null there)
                  const float v0 = __uint_as_float(u << 16), v1 = __uint_as_float(u & 0xFFFF0000u);
                  ssum[tn][2 * r2] += v0;
                  ssq[tn][2 * r2] += v0 * v0;
                  ssum[tn][2 * r2 + 1] += v1;
                  ssq[tn][2 * r2 + 1] += v1 * v1;
                }
                pk[2 * tn + r2] = u;
              }
              acc[tn][tm] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (ok) {
              uint4* dst = reinterpret_cast<uint4*>(y + (size_t)m * N + n0 + wn * 64 + g * 16);
              dst[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
              dst[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
            }
          }
        }
      }
    }
  }
  if (EP || !partial) return;
#pragma unroll
  for (int tn = 0; tn < 4; ++tn)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sv = ssum[tn][r], q = ssq[tn][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { sv += __shfl_xor(sv, o, 64); q += __shfl_xor(q, o, 64); }
      if (li == 0) { redS[wave][g * 16 + tn * 4 + r] = sv; redQ[wave][g * 16 + tn * 4 + r] = q; }      // the permuted channel of (tn, g, r)
    }
  __syncthreads();
  for (int i = t; i < BN; i += 256) {
    const int wn_i = i >> 6, c = i & 63;
    float sv = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < WM; ++j) { sv += redS[j * WN + wn_i][c]; q += redQ[j * WN + wn_i][c]; }
    float* p = partial + (size_t)blockIdx.x * 2 * N;
    p[n0 + i] = sv;
    p[N + n0 + i] = q;
  }
}

}  // namespace a3d

using namespace a3d;

static int c1_wn(int N) { return N >= 256 ? 4 : (N >= 128 ? 2 : 1); }
// dynamic LDS of the streaming kernel: W block + two X buffers + scale / shift
static size_t c1_stream_lds(int K, int N) {
  const int wn = c1_wn(N), bn = 64 * wn, bm = 64 * (4 / wn);
  return (size_t)(K / 32) * bn * 64 + (size_t)2 * bm * 64 + (size_t)2 * K * sizeof(float);
}
// the shapes the resident-weight kernel serves: K in {64, 128, 256}, its LDS block within 96 KB (two workgroups per CU up to 78 KB)
static bool c1_streams(int K, int N) {
  return (K == 64 || K == 128 || K == 256) && (N % 64) == 0 && (N < 256 || (N % 256) == 0) && c1_stream_lds(K, N) <= 96 * 1024;
}
static int c1_slabs(size_t M, int K, int N) {
  const int wn = c1_wn(N);
  const int bm = 64 * (4 / wn);
  const size_t mtiles = (M + bm - 1) / bm;
  const int ntiles = N / (64 * wn);
  // persistent: as many workgroups as are resident at once (LDS-limited), each walking its share of the M tiles
  const size_t lds = c1_stream_lds(K, N) + 2048;
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, (160 * 1024) / lds));      // 180 registers per lane: two workgroups per CU
  return (int)std::min<size_t>(mtiles, (size_t)std::max(1, 256 * per_cu / ntiles));
}

// the deep-layer GEMM (conv1x1_deep.hip) takes the shapes the resident-weight kernel refuses; A3D_CONV1X1_DEEP=0: those stay with the library (A/B)
// Measured (round 6, 256 images): in isolation (profiles/r06_conv1x1_layers.json: the GEMM with its folded passes against MIOpen's convolution +
// the BatchNorm passes it needs) it wins on five of the ten deep shapes and loses up to 16 % on the others -- its 64 x 64 wave tiles read
// 0.5 KB of LDS per MFMA, the LDS port's rate, so it runs at 0.43 - 0.72 PFLOP/s where CK reaches 0.6 - 0.86 -- but in the captured
// training step serving EVERY deep shape is fastest (profiles/r06_conv1x1_deep_ab.json: 21.01 ms per step against 21.24 with the five
// isolated winners only and 21.34 with none).  A3D_CONV1X1_DEEP=0 / a3d_conv1x1_deep_mode(0): those shapes stay with the library (A/B).
static int c1_deep_mode = getenv("A3D_CONV1X1_DEEP") ? atoi(getenv("A3D_CONV1X1_DEEP")) : 1;
extern "C" int a3d_conv1x1_deep_mode(int mode) {                // sets the mode (0 / 1) and returns the previous one; mode < 0: query only
  const int prev = c1_deep_mode;
  if (mode >= 0) c1_deep_mode = mode > 0 ? 1 : 0;
  return prev;
}
static bool c1_deep(int K, int N) { return c1_deep_mode != 0 && !c1_streams(K, N) && conv1x1_deep_serves(K, N); }

extern "C" int a3d_conv1x1_streams(int K, int N) { return (K > 0 && N > 0 && (c1_streams(K, N) || c1_deep(K, N))) ? 1 : 0; }

extern "C" int a3d_conv1x1_nslab(size_t M, int K, int N) {
  if (M == 0 || N <= 0 || K <= 0) return 0;
  if (c1_streams(K, N)) return c1_slabs(M, K, N);
  return c1_deep(K, N) ? conv1x1_deep_slabs(M, K, N) : 0;
}

extern "C" int a3d_conv1x1_bn_fwd(const void* x, const void* w, const float* in_scale, const float* in_shift, int in_relu,
                                  void* y, float* partial, size_t M, int K, int N, void* stream) {
  if (!x || !w || !y || M == 0 || K <= 0 || N <= 0 || !(c1_streams(K, N) || c1_deep(K, N)) || (in_scale && !in_shift) ||
      ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) != 0)) {
    set_error("a3d_conv1x1_bn_fwd: bad argument (M=%zu K=%d N=%d; served shapes: K in {64, 128, 256}, N in {64, 128, 256 j}, weight block "
              "+ buffers within 96 KB of LDS (the resident-weight kernel), or K = 64 j in 128 .. 2048 and N = 128 j up to 2048 (the deep-layer "
              "GEMM) -- a3d_conv1x1_streams; 16-byte aligned operands)", M, K, N);
    return A3D_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  if (!c1_streams(K, N)) return conv1x1_deep_launch(x, w, in_scale, in_shift, in_relu, y, partial, M, K, N, s);
  const unsigned short* xs = (const unsigned short*)x;
  const unsigned short* ws = (const unsigned short*)w;
  unsigned short* ys = (unsigned short*)y;
  const int slabs = c1_slabs(M, K, N);
  const size_t lds = c1_stream_lds(K, N);
  const dim3 grid(slabs, N >= 256 ? N / 256 : 1);
#define A3D_C1S(WNV, KSV)                                                                                                         \
  do {                                                                                                                           \
    static bool once = false;                                                                                                    \
    if (!once) { (void)hipFuncSetAttribute((const void*)conv1x1_stream_kernel<WNV, KSV>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); once = true; } \
    hipLaunchKernelGGL((conv1x1_stream_kernel<WNV, KSV>), grid, dim3(256), lds, s, xs, ws, in_scale, in_shift, in_relu, ys, partial, (long long)M, N, C1Epilogue{nullptr, 0, nullptr, 0, 0}); \
  } while (0)
  const int wn = c1_wn(N), ks = K / 32;
  if (wn == 4 && ks == 2) A3D_C1S(4, 2);
  else if (wn == 4 && ks == 4) A3D_C1S(4, 4);
  else if (wn == 2 && ks == 2) A3D_C1S(2, 2);
  else if (wn == 2 && ks == 4) A3D_C1S(2, 4);
  else if (wn == 2 && ks == 8) A3D_C1S(2, 8);
  else if (wn == 1 && ks == 2) A3D_C1S(1, 2);
  else if (wn == 1 && ks == 4) A3D_C1S(1, 4);
  else if (wn == 1 && ks == 8) A3D_C1S(1, 8);
  else { set_error("a3d_conv1x1_bn_fwd: no streaming instance for K=%d N=%d", K, N); return A3D_ERR_ARG; }
#undef A3D_C1S
  return check_launch("a3d_conv1x1_bn_fwd");
}

extern "C" int a3d_conv1x1_topdown_serves(int K, int N) { return (c1_streams(K, N) && N <= 128) ? 1 : 0; }

extern "C" int a3d_conv1x1_topdown_fwd(const void* x, const void* w, const float* bias, int nbias, const void* top, void* y,
                                       size_t images, int H, int W, int K, int N, void* stream) {
  const size_t M = images * (size_t)H * (size_t)W;
  if (!x || !w || !y || images == 0 || H <= 0 || W <= 0 || K <= 0 || N <= 0 || !a3d_conv1x1_topdown_serves(K, N) || M > 0xFFFFFFFFull ||
      (top && ((H & 1) || (W & 1))) || (bias && (nbias <= 0 || nbias > N)) ||
      ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)top) & 15) != 0)) {
    set_error("a3d_conv1x1_topdown_fwd: bad argument (images=%zu H=%d W=%d K=%d N=%d nbias=%d; served: K in {64, 128, 256}, N in {64, 128}, "
              "H and W even with a top map, at most 2^32 rows, 0 < nbias <= N with a bias, 16-byte aligned operands)", images, H, W, K, N, nbias);
    return A3D_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  const int slabs = c1_slabs(M, K, N);
  const size_t lds = c1_stream_lds(K, N);
  const dim3 grid(slabs, 1);
  const C1Epilogue ep{bias, bias ? nbias : 0, (const unsigned short*)top, H, W};
#define A3D_C1E(WNV, KSV)                                                                                                         \
  do {                                                                                                                           \
    static bool once = false;                                                                                                    \
    if (!once) { (void)hipFuncSetAttribute((const void*)conv1x1_stream_kernel<WNV, KSV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); once = true; } \
    hipLaunchKernelGGL((conv1x1_stream_kernel<WNV, KSV, true>), grid, dim3(256), lds, s, (const unsigned short*)x, (const unsigned short*)w, \
                       (const float*)nullptr, (const float*)nullptr, 0, (unsigned short*)y, (float*)nullptr, (long long)M, N, ep);  \
  } while (0)
  const int wn = c1_wn(N), ks = K / 32;
  if (wn == 2 && ks == 2) A3D_C1E(2, 2);
  else if (wn == 2 && ks == 4) A3D_C1E(2, 4);
  else if (wn == 2 && ks == 8) A3D_C1E(2, 8);
  else if (wn == 1 && ks == 2) A3D_C1E(1, 2);
  else if (wn == 1 && ks == 4) A3D_C1E(1, 4);
  else if (wn == 1 && ks == 8) A3D_C1E(1, 8);
  else { set_error("a3d_conv1x1_topdown_fwd: no streaming instance for K=%d N=%d", K, N); return A3D_ERR_ARG; }
#undef A3D_C1E
  return check_launch("a3d_conv1x1_topdown_fwd");
}
