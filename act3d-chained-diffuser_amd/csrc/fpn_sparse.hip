// Sparse backward of the FPN's 3x3 output convolution (torchvision FeaturePyramidNetwork.layer_blocks, act3d.py:76-77), round 6.
//
// Act3D reads the fine FPN map only through the k-NN gathers of its pyramid levels (act3d.py:244-260: 4096 of the 65 536 pixels of
// a sample per level, two levels on the 128 x 128 map), so the gradient that reaches the convolution's OUTPUT is non-zero on
// 6 - 12 % of the pixels -- but the library's backward (MIOpen igemm_wrw, 0.72 ms per step at the bench shape: the largest single
// launch of the step, and igemm_bwd, 0.34 ms) runs dense over a map that build_context_bwd first has to zero-fill and scatter into.
//
//   weight gradient  dW[tap][co][ci] = sum over gathered tokens j of  G_j[co] * X[pixel(j) + offset(tap)][ci]
//
// is computed here straight from what the gather's backward already holds -- the token indices and the fp32 gradient rows of the
// context -- without looking at the dense map: per 64 tokens the nine shifted input rows are GATHERED (128 B each, bf16 NHWC), and
// the contraction over tokens runs on the bf16 MFMA with both operands read TRANSPOSED out of LDS (ds_read_b64_tr_b16: tokens are
// the K dimension, but the tiles are stored token-major as they arrive).  12 % of the dense work, no zero-filled map in the loop.
// (The input gradient stays with the library for now: it needs the 3x3-dilated pixel set, i.e. a tile bitmap -- DESIGN.md.)
#include "attn_ring.h"
#include "../../include/act3d_hip.h"

namespace a3d {

constexpr int FS_C = 64;                 // channels of the FPN maps on the bf16 path (60 padded to 64)
constexpr int FS_T = 64;                 // tokens per block
constexpr int FS_LD = 72;                // LDS row stride in halfs (144 B: the 16 rows a fragment read touches fall on distinct banks)
constexpr int FS_WGS = 256;              // persistent workgroups (one partial tile set each)
constexpr int FS_DW = 9 * FS_C * FS_C;   // floats of one weight-gradient set [tap][co][ci]

// lane (li, g) <- column `col0 + li` of rows row0 + g * 8 .. + 7 of a token-major [rows][FS_LD] tile: an MFMA A / B fragment whose K
// index is the token (two transposed reads of 4 rows x 16 columns; lane i of a 16-lane group passes the address of row (i >> 2),
// columns 4 (i & 3) .. + 3 and receives column i -- attn_ring.h lds_tr16)
__device__ __forceinline__ s16x8 fs_tr_frag(const unsigned short* tile, int row0, int col0, int li, int g) {
  const unsigned short* p = tile + (row0 + g * 8 + (li >> 2)) * FS_LD + col0 + (li & 3) * 4;
  const s16x4_ a = lds_tr16(p), b = lds_tr16(p + 4 * FS_LD);
  return s16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_tokens_kernel(
    const unsigned short* __restrict__ X, const long long* __restrict__ idx, const float* __restrict__ G, int g_rows, int E,
    float* __restrict__ partial, int B, int k, int ncam, int H, int W) {
  __shared__ __attribute__((aligned(16))) unsigned short Gs[FS_T * FS_LD];
  __shared__ __attribute__((aligned(16))) unsigned short Xs[3][FS_T * FS_LD];
  __shared__ int tokN[FS_T], tokH[FS_T], tokW[FS_T];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int bps = (k + FS_T - 1) / FS_T;                      // blocks per sample
  const int nblocks = B * bps;
  f32x4 acc[9][4];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const int b = blk / bps, j0 = (blk - b * bps) * FS_T;
    __syncthreads();                                          // the previous block's fragment reads are done
    if (t < FS_T) {
      const int j = j0 + t;
      int n = -1, h = 0, w = 0;
      if (j < k) {
        const long long p = idx[(size_t)b * k + j];
        const int cam = (int)(p / ((long long)H * W));
        const int pix = (int)(p - (long long)cam * H * W);
        n = b * ncam + cam; h = pix / W; w = pix - h * W;
      }
      tokN[t] = n; tokH[t] = h; tokW[t] = w;
    }
    {
      // gradient rows -> bf16, token-major: thread = (token, 16-channel block)
      const int tk = t >> 2, c0 = (t & 3) * 16;
      const int j = j0 + tk;
      unsigned int w16[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = c0 + 2 * i;
        float a = 0.f, bb = 0.f;
        if (j < k) {
          const float* row = G + ((size_t)b * g_rows + j) * E;
          if (c < E) a = row[c];
          if (c + 1 < E) bb = row[c + 1];
        }
        w16[i] = pk_bf16(a, bb);
      }
      u32x4_* dst = reinterpret_cast<u32x4_*>(&Gs[tk * FS_LD + c0]);
      dst[0] = u32x4_{w16[0], w16[1], w16[2], w16[3]};
      dst[1] = u32x4_{w16[4], w16[5], w16[6], w16[7]};
    }
    __syncthreads();
#pragma unroll
    for (int dyi = 0; dyi < 3; ++dyi) {
      {
        // the three shifted input rows (dx = -1, 0, 1) of every token: thread = (token, 32-byte segment)
        const int tk = t >> 2, seg = t & 3;
        const int n = tokN[tk], hh = tokH[tk] + dyi - 1;
        u32x4_ v[3][2];
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) {
          const int ww = tokW[tk] + dxi - 1;
          v[dxi][0] = u32x4_{0u, 0u, 0u, 0u};
          v[dxi][1] = v[dxi][0];
          if (n >= 0 && hh >= 0 && hh < H && ww >= 0 && ww < W) {
            const u32x4_* src = reinterpret_cast<const u32x4_*>(X + (((size_t)n * H + hh) * W + ww) * FS_C + seg * 16);
            v[dxi][0] = src[0];
            v[dxi][1] = src[1];
          }
        }
        if (dyi) __syncthreads();                             // the previous tap row's fragment reads are done
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) {
          u32x4_* dst = reinterpret_cast<u32x4_*>(&Xs[dxi][tk * FS_LD + seg * 16]);
          dst[0] = v[dxi][0];
          dst[1] = v[dxi][1];
        }
      }
      __syncthreads();
      // wave w owns output channels co = 16 w .. + 15: acc[tap][ct] (co x ci tile ct) += G^T (co x tokens) . X_tap (tokens x ci)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const s16x8 ga = fs_tr_frag(Gs, ks * 32, wave * 16, li, g);
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct)
            acc[dyi * 3 + dxi][ct] = mfma_bf16_16x16x32(ga, fs_tr_frag(Xs[dxi], ks * 32, ct * 16, li, g), acc[dyi * 3 + dxi][ct]);
      }
    }
  }
  // this workgroup's partial set: [tap][co][ci]; lane (li = ci, g) register r holds co = 16 w + 4 g + r
  float* out = partial + (size_t)blockIdx.x * FS_DW;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(tap * FS_C + wave * 16 + g * 4 + r) * FS_C + ct * 16 + li] = acc[tap][ct][r];
}

// dW[i] (+)= sum over the workgroups' partial sets, in a fixed order
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW, int nsets,
                                                                   int accumulate) {
  // thread = (element i, quarter q of the sets): four threads per element so that 4 x the loads are in flight (the first version
  // -- one thread walking all 256 sets -- took 60 us for 37 MB: a latency chain), combined through LDS in a fixed order
  __shared__ float red[4][64];
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < FS_DW) {
    const int per = (nsets + 3) / 4;
    const int p0 = q * per, p1 = min(nsets, p0 + per);
    int p = p0;
    for (; p + 4 <= p1; p += 4) {
      s0 += partial[(size_t)p * FS_DW + i];
      s1 += partial[(size_t)(p + 1) * FS_DW + i];
      s2 += partial[(size_t)(p + 2) * FS_DW + i];
      s3 += partial[(size_t)(p + 3) * FS_DW + i];
    }
    for (; p < p1; ++p) s0 += partial[(size_t)p * FS_DW + i];
  }
  red[q][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q == 0 && i < FS_DW) {
    const float s = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    dW[i] = accumulate ? dW[i] + s : s;
  }
}

}  // namespace a3d

using namespace a3d;

extern "C" size_t a3d_conv3x3_wgrad_tokens_ws_floats(void) { return (size_t)FS_WGS * FS_DW; }

extern "C" int a3d_conv3x3_wgrad_tokens(const void* X, const long long* idx, const float* G, int g_rows, int E, float* ws, float* dW,
                                        int accumulate, int B, int k, int ncam, int H, int W, void* stream) {
  if (!X || !idx || !G || !ws || !dW || B <= 0 || k <= 0 || ncam <= 0 || H <= 0 || W <= 0 || E <= 0 || E > FS_C || g_rows < k ||
      (((uintptr_t)X) & 15) != 0) {
    set_error("a3d_conv3x3_wgrad_tokens: bad argument (B=%d k=%d ncam=%d H=%d W=%d E=%d g_rows=%d; E <= 64, X 16-byte aligned)", B, k,
              ncam, H, W, E, g_rows);
    return A3D_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  const int nblocks = B * ((k + FS_T - 1) / FS_T);
  const int grid = nblocks < FS_WGS ? nblocks : FS_WGS;
  hipLaunchKernelGGL(conv3x3_wgrad_tokens_kernel, dim3(grid), dim3(256), 0, s, (const unsigned short*)X, idx, G, g_rows, E, ws, B, k,
                     ncam, H, W);
  int rc = check_launch("a3d_conv3x3_wgrad_tokens");
  if (rc) return rc;
  hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(cdiv(FS_DW, 64)), dim3(256), 0, s, ws, dW, grid, accumulate ? 1 : 0);
  return check_launch("a3d_conv3x3_wgrad_tokens(reduce)");
}
