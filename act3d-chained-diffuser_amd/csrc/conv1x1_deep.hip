// 1x1 convolutions of the frozen backbone's DEEP layers (CLIP ModifiedResNet layers 2 - 4, model/utils/clip.py:28-43: conv1 / conv3 /
// downsample with K, N in 128 .. 2048 at M = 2^14 .. 2^18 rows) as a bf16 MFMA GEMM with the neighbouring BatchNorm work folded in
// (round 6).  Same contract as conv1x1.hip's streaming kernel (a3d_conv1x1_bn_fwd dispatches here for the shapes
// that kernel refuses):
//   y[m][n] = sum_k f(x[m][k]) * w[n][k],   f(x) = relu?(x * in_scale[k] + in_shift[k]) rounded to bf16 (identity without in_scale)
//   x [M][K] bf16 (NHWC rows), w [N][K] bf16, y [M][N] bf16 (fp32 accumulation, one rounding); optional epilogue: per-workgroup partial
//   (sum, sum of squares) of the ROUNDED outputs per channel, [slab][2][N] -- the statistics pass of the BatchNorm that follows.
// Why: on these layers the library convolution (CK) runs at 0.5 - 0.86 PFLOP/s, partly HBM-bound already (profiles/r04_conv_layers.json),
// and every one of them is followed by a statistics pass over its output (bn_stats: one more read of up to 268 MB) and, for conv3
// of the un-strided blocks, preceded by a BatchNorm-apply + ReLU pass over its input (read + write).  This kernel has to match the
// library's GEMM rate to keep what folding those passes wins.
// Structure: 128 x 128 output tile per workgroup (4 waves as 2 x 2, 64 x 64 per wave = 16 accumulator tiles of v_mfma_f32_16x16x32_bf16),
// K steps of 64; both operands go global -> registers one step ahead -> (normalise) -> LDS, two LDS buffers, ONE barrier per step; a
// persistent workgroup keeps its block of N and walks its share of the M tiles as one flat sequence of (tile, K step) pairs, so the
// load pipeline never drains at a tile boundary; the workgroups of all N blocks walk the same M tiles at the same time (x is fetched
// from HBM once and shared through L2).  The products are computed TRANSPOSED (A operand = weight rows, B operand = x rows) with the
// weight rows of a wave permuted (MFMA tile tn row i <-> channel (i >> 2) 16 + tn 4 + (i & 3)), so that a lane ends up with 16
// consecutive channels of its row: 32-byte stores, full 128-byte lines per row and wave -- conv1x1.hip's epilogue.
// LDS rows are 128 bytes (64 bf16) in eight 16-byte segments; segment s of row r is stored at s ^ key(r): x rows (read as 16
// consecutive rows per instruction) key = (r >> 1) & 7, weight rows (read in the permuted order) key = ((r >> 4) & 3) << 1 | ((r >> 1) & 1)
// -- the 16 rows of a ds_read_b128 then fall on 16 distinct (bank half, segment) pairs.
#include "a3d_common.h"
#include "../../include/act3d_hip.h"

namespace a3d {

constexpr int CD_BM = 128, CD_BN = 128, CD_BK = 64;

__device__ __forceinline__ int cd_xoff(int row, int seg) { return row * 64 + ((seg ^ ((row >> 1) & 7)) << 3); }
__device__ __forceinline__ int cd_woff(int row, int seg) { return row * 64 + ((seg ^ ((((row >> 4) & 3) << 1) | ((row >> 1) & 1))) << 3); }

// BatchNorm-apply (+ ReLU) of the producer on one bf16 pair: v_cvt_pk_bf16_f32 rounds to nearest even, the bits the unfused a3d_bn_apply writes
__device__ __forceinline__ unsigned int cd_norm2(unsigned int u, const float* scS, int K, int k, float relu_lo) {
  float a = __uint_as_float(u << 16) * scS[k] + scS[K + k];
  float b = __uint_as_float(u & 0xFFFF0000u) * scS[k + 1] + scS[K + k + 1];
  return pk_bf16(fmaxf(a, relu_lo), fmaxf(b, relu_lo));
}

// one K step of a wave: 2 K halves x (4 x-row fragments + 4 permuted weight fragments -> 16 MFMAs)
__device__ __forceinline__ void cd_compute(const unsigned short* Xb, const unsigned short* Wb, const int (&xo)[2][4], const int (&wo)[2][4],
                                           f32x4 (&acc)[4][4]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    s16x8 xa[4], wb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xa[i] = *reinterpret_cast<const s16x8*>(&Xb[xo[h][i]]);
      wb[i] = *reinterpret_cast<const s16x8*>(&Wb[wo[h][i]]);
    }
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) acc[tn][tm] = mfma_bf16_16x16x32(wb[tn], xa[tm], acc[tn][tm]);
  }
}

// tile done: round once, statistics of the rounded values, 16 consecutive channels per lane and row (conv1x1.hip's epilogue)
__device__ __forceinline__ void cd_epilogue(f32x4 (&acc)[4][4], float (&ssum)[4][4], float (&ssq)[4][4], unsigned short* __restrict__ y,
                                            long long m0, long long M, int N, int n0, int wm, int wn, int li, int g) {
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
    const long long m = m0 + wm * 64 + tm * 16 + li;
    const bool ok = m < M;
    unsigned int pk[8];
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        const unsigned int u = pk_bf16(acc[tn][tm][2 * r2], acc[tn][tm][2 * r2 + 1]);
        const float v0 = ok ? __uint_as_float(u << 16) : 0.f, v1 = ok ? __uint_as_float(u & 0xFFFF0000u) : 0.f;
        ssum[tn][2 * r2] += v0;
        ssq[tn][2 * r2] += v0 * v0;
        ssum[tn][2 * r2 + 1] += v1;
        ssq[tn][2 * r2 + 1] += v1 * v1;
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

template <bool XFORM>
__global__ __launch_bounds__(256, 2) void conv1x1_deep_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w,
                                                            const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                            int in_relu, unsigned short* __restrict__ y, float* __restrict__ partial,
                                                            long long M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem_cd[];
  unsigned short* Xs = smem_cd;                                   // [2][128 rows x 64]
  unsigned short* Ws = Xs + 2 * CD_BM * CD_BK;                    // [2][128 rows x 64]
  float* scS = reinterpret_cast<float*>(Ws + 2 * CD_BN * CD_BK);  // [K] scale | [K] shift (only with in_scale)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.y * CD_BN;
  const int KS = K / CD_BK;
  const long long mtiles = (M + CD_BM - 1) / CD_BM;
  if (XFORM)
    for (int i = t; i < K; i += 256) { scS[i] = in_scale[i]; scS[K + i] = in_shift[i]; }
  float ssum[4][4], ssq[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[a][r] = 0.f; ssq[a][r] = 0.f; }
  const long long my_tiles = blockIdx.x < mtiles ? (mtiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const long long total = my_tiles * KS;                          // flat steps: s -> (tile blockIdx.x + (s / KS) gridDim.x, K step s % KS)
  // this thread's four 16-byte segments of a 128 x 64 operand tile: rows (t >> 3) + 32 i, segment t & 7 (8 threads = one 128-byte row piece)
  const int srow = t >> 3, sseg = t & 7;
  // the load stream's position (one step ahead of the compute stream's), advanced without divisions
  long long l_m0 = (long long)blockIdx.x * CD_BM;
  int l_ks = 0;
  const float relu_lo = in_relu ? 0.f : -INFINITY;
  // (the loads and the staging are written out in the loop body on plain local arrays: passed by reference into lambdas, hipcc kept
  // one of the two register sets in scratch memory across the loop's back edge -- a store + vmcnt wait right behind every prefetch)
#define CD_LOAD1(P, i)                                                                                                \
  {                                                                                                                   \
    const long long m_ = l_m0 + srow + 32 * i < M ? l_m0 + srow + 32 * i : M - 1; /* clamped: masked at the store */   \
    P##x##i = *reinterpret_cast<const uint4*>(x + (size_t)m_ * K + l_ks * CD_BK + sseg * 8);                          \
    P##w##i = *reinterpret_cast<const uint4*>(w + (size_t)(n0 + srow + 32 * i) * K + l_ks * CD_BK + sseg * 8);        \
  }
  // fetch the load stream's current step into register set P, then advance the stream (past the end: the last chunk again, never staged)
#define CD_LOAD(P)                                                                                                    \
  do {                                                                                                                \
    CD_LOAD1(P, 0) CD_LOAD1(P, 1) CD_LOAD1(P, 2) CD_LOAD1(P, 3)                                                       \
    if (l_left > 1) { --l_left; if (++l_ks == KS) { l_ks = 0; l_m0 += (long long)gridDim.x * CD_BM; } }               \
  } while (0)
#define CD_STAGE1(P, i)                                                                                               \
  {                                                                                                                   \
    uint4 v = P##x##i;                                                                                                \
    if (XFORM) {                                                                                                      \
      const int k0 = ks * CD_BK + sseg * 8;                                                                           \
      unsigned int u0 = v.x, u1 = v.y, u2 = v.z, u3 = v.w;                                                            \
      u0 = cd_norm2(u0, scS, K, k0 + 0, relu_lo); u1 = cd_norm2(u1, scS, K, k0 + 2, relu_lo);                         \
      u2 = cd_norm2(u2, scS, K, k0 + 4, relu_lo); u3 = cd_norm2(u3, scS, K, k0 + 6, relu_lo);                         \
      v = make_uint4(u0, u1, u2, u3);                                                                                 \
    }                                                                                                                 \
    *reinterpret_cast<uint4*>(&Xs[buf * CD_BM * CD_BK + cd_xoff(srow + 32 * i, sseg)]) = v;                           \
    *reinterpret_cast<uint4*>(&Ws[buf * CD_BN * CD_BK + cd_woff(srow + 32 * i, sseg)]) = P##w##i;                     \
  }
#define CD_STAGE(P) CD_STAGE1(P, 0) CD_STAGE1(P, 1) CD_STAGE1(P, 2) CD_STAGE1(P, 3)
  // fragment offsets (halfs) of the first K half of a step; the second half is segment + 4, i.e. offset ^ 32 halfs (the keys only
  // touch the segment's low bits... they touch all three: add 4 to the segment BEFORE the xor -> precompute both)
  int xo[2][4], wo[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xo[h][i] = cd_xoff(wm * 64 + i * 16 + li, h * 4 + g);
      wo[h][i] = cd_woff(wn * 64 + (li >> 2) * 16 + i * 4 + (li & 3), h * 4 + g);
    }
  if (total > 0) {
    // TWO register sets, two steps in flight (round 6, first measurement: with one step ahead a step took ~3 us -- the bare memory latency
    // -- against 0.26 us of MFMA issue; the library's kernel was 1.7x faster on the 16 x 16 maps)
    uint4 ax0, ax1, ax2, ax3, aw0, aw1, aw2, aw3, bx0, bx1, bx2, bx3, bw0, bw1, bw2, bw3;
    long long l_left = total;                                     // steps the load stream still has to fetch (it parks on the last one)
    CD_LOAD(a);
    CD_LOAD(b);
    long long c_m0 = (long long)blockIdx.x * CD_BM;               // the compute stream's tile
    int ks = 0;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                              // scale / shift staged
#define CD_STEP(P, BUF)                                                                                               \
    {                                                                                                                 \
      constexpr int buf = BUF;                                                                                        \
      CD_STAGE(P)                                                                                                     \
      CD_LOAD(P);                                                 /* two steps ahead */                               \
      __syncthreads();                                            /* stage(s) visible; every wave is past its reads of the other buffer */ \
      cd_compute(Xs + buf * CD_BM * CD_BK, Ws + buf * CD_BN * CD_BK, xo, wo, acc);                                    \
      if (ks == KS - 1) cd_epilogue(acc, ssum, ssq, y, c_m0, M, N, n0, wm, wn, li, g);                                \
      if (++ks == KS) { ks = 0; c_m0 += (long long)gridDim.x * CD_BM; }                                               \
    }
    for (long long s = 0; s < total; s += 2) {
      CD_STEP(a, 0)
      if (s + 1 < total) CD_STEP(b, 1)                            // workgroup-uniform
    }
#undef CD_STEP
  }
  if (!partial) return;
  __syncthreads();                                                // every wave is past its last fragment reads: Xs is free
  float* redS = reinterpret_cast<float*>(Xs);                     // [4 waves][64] sums | [4][64] sums of squares
  float* redQ = redS + 4 * 64;
#pragma unroll
  for (int tn = 0; tn < 4; ++tn)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sv = ssum[tn][r], q = ssq[tn][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { sv += __shfl_xor(sv, o, 64); q += __shfl_xor(q, o, 64); }
      if (li == 0) { redS[wave * 64 + g * 16 + tn * 4 + r] = sv; redQ[wave * 64 + g * 16 + tn * 4 + r] = q; }      // the permuted channel of (tn, g, r)
    }
  __syncthreads();
  if (t < CD_BN) {
    const int wn_i = t >> 6, c = t & 63;
    const float sv = redS[(0 * 2 + wn_i) * 64 + c] + redS[(1 * 2 + wn_i) * 64 + c];       // the two M halves (wm = 0, 1) of column block wn_i
    const float q = redQ[(0 * 2 + wn_i) * 64 + c] + redQ[(1 * 2 + wn_i) * 64 + c];
    float* p = partial + (size_t)blockIdx.x * 2 * N;
    p[n0 + t] = sv;
    p[N + n0 + t] = q;
  }
}

#undef CD_LOAD
#undef CD_LOAD1
#undef CD_STAGE1
#undef CD_STAGE

}  // namespace a3d

using namespace a3d;

// served: K a multiple of 64 from 128 to 2048, N a multiple of 128 up to 2048; the folded BatchNorm-apply for K <= 1024
bool a3d::conv1x1_deep_serves(int K, int N) { return K >= 128 && K <= 2048 && (K % CD_BK) == 0 && N >= 128 && N <= 2048 && (N % CD_BN) == 0; }
static size_t cd_lds(int K, bool xform) { return (size_t)2 * (CD_BM + CD_BN) * CD_BK * 2 + (xform ? (size_t)2 * K * sizeof(float) : 0); }

int a3d::conv1x1_deep_slabs(size_t M, int K, int N) {
  (void)K;
  const size_t mtiles = (M + CD_BM - 1) / CD_BM;
  const int nblocks = N / CD_BN;
  return (int)std::min<size_t>(mtiles, (size_t)std::max(1, 512 / nblocks));       // two resident workgroups per CU
}

int a3d::conv1x1_deep_launch(const void* x, const void* w, const float* in_scale, const float* in_shift, int in_relu, void* y, float* partial,
                             size_t M, int K, int N, hipStream_t s) {
  if (in_scale && K > 1024) { set_error("a3d_conv1x1_bn_fwd: the folded BatchNorm-apply of the deep-layer kernel serves K <= 1024 (K=%d)", K); return A3D_ERR_ARG; }
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute((const void*)conv1x1_deep_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute((const void*)conv1x1_deep_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    once = true;
  }
  const dim3 grid(conv1x1_deep_slabs(M, K, N), N / CD_BN);
  if (in_scale)
    hipLaunchKernelGGL(conv1x1_deep_kernel<true>, grid, dim3(256), cd_lds(K, true), s, (const unsigned short*)x, (const unsigned short*)w,
                       in_scale, in_shift, in_relu, (unsigned short*)y, partial, (long long)M, N, K);
  else
    hipLaunchKernelGGL(conv1x1_deep_kernel<false>, grid, dim3(256), cd_lds(K, false), s, (const unsigned short*)x, (const unsigned short*)w,
                       in_scale, in_shift, in_relu, (unsigned short*)y, partial, (long long)M, N, K);
  return check_launch("a3d_conv1x1_bn_fwd(deep)");
}
