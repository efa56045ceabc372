// The stem's first convolution of the frozen backbone (CLIP ModifiedResNet conv1: 3 -> 32 channels, 3x3, stride 2, padding 1, no bias;
// model/utils/clip.py:22-43) with the image normalisation in front of it and the BatchNorm statistics behind it folded in
// (round 6):
//   y[n][oy][ox][co] = sum_{ci, kh, kw} bf16((rgb[n][ci][2 oy + kh - 1][2 ox + kw - 1] - mean[ci]) / std[ci]) * w[co][ci][kh][kw]
//   (0 outside the image: the padding of the NORMALISED map, as F.conv2d pads the map a3d_rgb_normalize_nhwc_bf16 writes);
//   rgb fp32 planar [N][3][H][W] (the reference's `rgbs`, act3d.py:365 normalises them), w bf16 [32][27], y bf16 NHWC [N][H/2][W/2][32],
//   fp32 accumulation, one rounding; epilogue: per-workgroup (sum, sum of squares) of the rounded outputs, [slab][2][32].
// Why: the library spent 147 - 214 us on this 7 GFLOP layer (0.37 GB of traffic: 46 us of HBM time), after a 50 us normalisation
// pass that writes the bf16 NHWC image and before a 45 us statistics pass that re-reads the 268 MB output.  Here the raw image is read
// once and the output written once.
// Structure: persistent workgroups walk 8 x 32-pixel output tiles; the 17 x 65 x 3 input window is normalised into LDS as bf16; K =
// 27 (padded to 32) is ONE MFMA step: per wave 2 output rows x 32 pixels = 4 pixel tiles x 2 channel tiles of v_mfma_f32_16x16x32_bf16,
// computed transposed (A = weight rows, B = pixel patches) with the weight rows permuted (tile tn row i <-> channel (i >> 2) 8 + tn 4 +
// (i & 3)) so that a lane holds 8 consecutive channels of its pixel: 16-byte stores, one full 64-byte row per pixel and wave.
#include "a3d_common.h"
#include "../../include/act3d_hip.h"

namespace a3d {

constexpr int ST_TH = 8, ST_TW = 32;                   // output tile
constexpr int ST_IH = 2 * ST_TH + 1, ST_IW = 2 * ST_TW + 1, ST_LD = ST_IW + 1;      // input window 17 x 65, LDS row stride 66

__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ rgb, const float* __restrict__ mean,
                                                        const float* __restrict__ stdv, const unsigned short* __restrict__ w,
                                                        unsigned short* __restrict__ y, float* __restrict__ partial, int nimg, int H,
                                                        int W) {
  __shared__ unsigned short Xs[3 * ST_IH * ST_LD];
  __shared__ float redS[4][32], redQ[4][32];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int OH = H / 2, OW = W / 2;
  const int tiles_x = OW / ST_TW, tiles_y = OH / ST_TH, tpi = tiles_x * tiles_y;
  const int ntiles = nimg * tpi;
  // weight fragments: A operand row i of channel tile tn = channel (i >> 2) 8 + tn 4 + (i & 3); lane (li, g) holds k = 8 g .. 8 g + 7
  s16x8 wf0, wf1;
  {
    const int c0 = (li >> 2) * 8 + (li & 3), c1 = c0 + 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = g * 8 + j;
      wf0[j] = (short)(k < 27 ? w[c0 * 27 + k] : 0);
      wf1[j] = (short)(k < 27 ? w[c1 * 27 + k] : 0);
    }
  }
  // this lane's patch offsets in the LDS window, relative to its pixel's window origin: k = (ci 3 + kh) 3 + kw (k >= 27: any finite
  // window value -- its weight is zero)
  auto koff_of = [&](int j) {
    const int k = g * 8 + j, ci = k / 9, kh = (k - ci * 9) / 3, kw = k - ci * 9 - kh * 3;
    return k < 27 ? (ci * ST_IH + kh) * ST_LD + kw : 0;
  };
  const int koff0 = koff_of(0), koff1 = koff_of(1), koff2 = koff_of(2), koff3 = koff_of(3), koff4 = koff_of(4), koff5 = koff_of(5),
            koff6 = koff_of(6), koff7 = koff_of(7);
  __shared__ float msS[6];                                      // mean | std (read by row: a select chain over six registers became a scratch table)
  if (t < 3) { msS[t] = mean[t]; msS[3 + t] = stdv[t]; }
  // statistics of this lane's 8 channels (tile 0 rows r = 0..3 -> index r, tile 1 -> 4 + r)
  float sm0 = 0.f, sm1 = 0.f, sm2 = 0.f, sm3 = 0.f, sm4 = 0.f, sm5 = 0.f, sm6 = 0.f, sm7 = 0.f;
  float sq0 = 0.f, sq1 = 0.f, sq2 = 0.f, sq3 = 0.f, sq4 = 0.f, sq5 = 0.f, sq6 = 0.f, sq7 = 0.f;
#define ST_STAT_(i, v) { sm##i += (v); sq##i += (v) * (v); }
#define ST_STAT2_(i, j, u) ST_STAT_(i, __uint_as_float((u) << 16)) ST_STAT_(j, __uint_as_float((u) & 0xFFFF0000u))
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int img = tile / tpi, rem = tile - img * tpi, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int iy0 = 2 * ty * ST_TH - 1, ix0 = 2 * tx * ST_TW - 1;
    __syncthreads();                                              // the previous tile's patch reads are done
    // window rows (ci, r) = 51 rows of 65 columns: a wave takes every fourth row, a lane one column (+ column 64 by the first 51 threads).
    // ALL loads first, unconditional on clamped addresses (with the bounds test around the load every element was its own round trip to
    // memory: 258 us for the layer), then normalise (the same subtraction and division as the separate normalisation pass: the same
    // bf16 value), round with v_cvt_pk_bf16_f32, zero-pad
    const float* img_base = rgb + (size_t)img * 3 * H * W;
    float xv[14];
#pragma unroll
    for (int i = 0; i < 13; ++i) {
      const int rr = min(wave + 4 * i, 3 * ST_IH - 1);
      const int ci = (rr >= ST_IH) + (rr >= 2 * ST_IH), r = rr - ci * ST_IH;
      const int iy = min(max(iy0 + r, 0), H - 1), ix = min(max(ix0 + lane, 0), W - 1);
      xv[i] = img_base[((size_t)ci * H + iy) * W + ix];
    }
    {
      const int rr = min(t, 3 * ST_IH - 1);
      const int ci = (rr >= ST_IH) + (rr >= 2 * ST_IH), r = rr - ci * ST_IH;
      const int iy = min(max(iy0 + r, 0), H - 1), ix = min(ix0 + ST_IW - 1, W - 1);
      xv[13] = img_base[((size_t)ci * H + iy) * W + ix];
    }
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      const int rr = i < 13 ? wave + 4 * i : t, c = i < 13 ? lane : ST_IW - 1;
      if (rr < 3 * ST_IH) {
        const int ci = (rr >= ST_IH) + (rr >= 2 * ST_IH), r = rr - ci * ST_IH;
        const int iy = iy0 + r, ix = ix0 + c;
        const bool inside = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const unsigned int u = pk_bf16((xv[i] - msS[ci]) / msS[3 + ci], 0.f);      // the division a3d_rgb_normalize_nhwc_bf16 performs: the same bf16 value
        Xs[rr * ST_LD + c] = inside ? (unsigned short)(u & 0xFFFFu) : (unsigned short)0;      // the zero padding of the normalised map
      }
    }
    __syncthreads();
    const int oy0 = ty * ST_TH, ox0 = tx * ST_TW;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
      // pixel tile tm of this wave: output row 2 wave + (tm >> 1), columns (tm & 1) 16 + li; its window origin in the LDS window
      const int orow = 2 * wave + (tm >> 1), ocol = (tm & 1) * 16 + li;
      const unsigned short* xp = Xs + (2 * orow) * ST_LD + 2 * ocol;
      const s16x8 xa = {(short)xp[koff0], (short)xp[koff1], (short)xp[koff2], (short)xp[koff3], (short)xp[koff4], (short)xp[koff5],
                        (short)xp[koff6], (short)xp[koff7]};
      const f32x4 a0 = mfma_bf16_16x16x32(wf0, xa, f32x4{0.f, 0.f, 0.f, 0.f}), a1 = mfma_bf16_16x16x32(wf1, xa, f32x4{0.f, 0.f, 0.f, 0.f});
      const unsigned int p0 = pk_bf16(a0[0], a0[1]), p1 = pk_bf16(a0[2], a0[3]), p2 = pk_bf16(a1[0], a1[1]), p3 = pk_bf16(a1[2], a1[3]);
      ST_STAT2_(0, 1, p0) ST_STAT2_(2, 3, p1) ST_STAT2_(4, 5, p2) ST_STAT2_(6, 7, p3)
      // lane group g: channels 8 g .. 8 g + 7 (tile 0: + 0..3, tile 1: + 4..7) of pixel (oy, ox)
      const int oy = oy0 + orow, ox = ox0 + ocol;
      *reinterpret_cast<uint4*>(y + (((size_t)img * OH + oy) * OW + ox) * 32 + g * 8) = make_uint4(p0, p1, p2, p3);
    }
  }
  if (!partial) return;
#define ST_RED(i)                                                                                         \
  {                                                                                                       \
    float sv = sm##i, q = sq##i;                                                                          \
    _Pragma("unroll") for (int o = 1; o < 16; o <<= 1) { sv += __shfl_xor(sv, o, 64); q += __shfl_xor(q, o, 64); } \
    if (li == 0) { redS[wave][g * 8 + i] = sv; redQ[wave][g * 8 + i] = q; }                               \
  }
  ST_RED(0) ST_RED(1) ST_RED(2) ST_RED(3) ST_RED(4) ST_RED(5) ST_RED(6) ST_RED(7)
#undef ST_RED
#undef ST_STAT_
#undef ST_STAT2_
  __syncthreads();
  if (t < 32) {
    float* p = partial + (size_t)blockIdx.x * 64;
    p[t] = redS[0][t] + redS[1][t] + redS[2][t] + redS[3][t];
    p[32 + t] = redQ[0][t] + redQ[1][t] + redQ[2][t] + redQ[3][t];
  }
}

}  // namespace a3d

using namespace a3d;

static bool stem_serves(size_t N, int H, int W) {
  return N > 0 && N < 65536 && H > 0 && W > 0 && (H % (2 * ST_TH)) == 0 && (W % (2 * ST_TW)) == 0 && (size_t)H * W < ((size_t)1 << 30);
}
extern "C" int a3d_stem_conv_nslab(size_t N, int H, int W) {
  if (!stem_serves(N, H, W)) return 0;
  const size_t ntiles = N * (size_t)(H / (2 * ST_TH)) * (size_t)(W / (2 * ST_TW));
  return (int)std::min<size_t>(ntiles, 2048);                   // persistent: eight workgroups per CU
}
extern "C" int a3d_stem_conv_bn_fwd(const float* rgb, const float* mean, const float* stdv, const void* w, void* y, float* partial, size_t N,
                                    int H, int W, void* stream) {
  if (!rgb || !mean || !stdv || !w || !y || !stem_serves(N, H, W) || ((((uintptr_t)y) & 15) != 0)) {
    set_error("a3d_stem_conv_bn_fwd: bad argument (images=%zu H=%d W=%d; H a multiple of 16, W of 64; y 16-byte aligned)", N, H, W);
    return A3D_ERR_ARG;
  }
  hipLaunchKernelGGL(stem_conv_kernel, dim3(a3d_stem_conv_nslab(N, H, W)), dim3(256), 0, (hipStream_t)stream, rgb, mean, stdv,
                     (const unsigned short*)w, (unsigned short*)y, partial, (int)N, H, W);
  return check_launch("a3d_stem_conv_bn_fwd");
}
