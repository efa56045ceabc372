// RoPE-3D applied in-register to projected q/k rows + conversion to the attention kernel's operand formats.
//
// The reference materialises a (B,N,E,2) cos/sin code tensor per token set (position_encodings.py:64-97)
// and rotates the *projected, already scaled* q and k over the full E vector before the head split
// (multihead_custom_attention.py:325,348-359; embed_rotary position_encodings.py:31-34).  Here the angles
// are recomputed from xyz in-kernel (12 B/token instead of 8E B/token) and the result is written straight
// into the MFMA operand layouts used by attention.hip:
//
//   rows format [B][H][Npad][W] bf16 : per row  hi(16) | lo(16) [| lo2(16)]  (head dim 15 padded to 16); W = 48
//              (x = hi + lo + lo2) for the q / k score operands, W = 32 (x = hi + lo) for v / dO rows
//   VT format  [B][H][2][16][Npad] bf16 : plane 0 = hi, plane 1 = lo, transposed so that 8 consecutive
//              keys of one channel are one 16-byte MFMA A-fragment.
// Rows n >= N and slot d = 15 are written as zeros (finite padding is required by 0 * x in PV).
#include "a3d_common.h"
#include "../../include/act3d_hip.h"

namespace a3d {

constexpr int RT_ROWS = 64;

// fills T[r][c] (r < 64, c < E) with the rotated, scaled row; rows >= N are zero
__device__ __forceinline__ void rope_tile_to_lds(float* T, int ldt, const float* __restrict__ Y, int ldy,
                                                 const float* __restrict__ xyz,
                                                 const float* __restrict__ freq, float scale, int b,
                                                 int n0, int N, int E) {
  const int half = E >> 1;
  const int third = E / 3;
  for (int idx = threadIdx.x; idx < RT_ROWS * half; idx += blockDim.x) {
    const int r = idx / half, p = idx - r * half;
    const int n = n0 + r;
    float o0 = 0.f, o1 = 0.f;
    if (n < N) {
      const size_t m = (size_t)b * N + n;
      const float y0 = Y[m * ldy + 2 * p] * scale;
      const float y1 = Y[m * ldy + 2 * p + 1] * scale;
      if (xyz) {
        const int c = 2 * p;
        const int axis = c / third;
        const int k = (c - axis * third) >> 1;
        const float th = xyz[m * 3 + axis] * freq[k];
        float sn, cs;
        sincosf(th, &sn, &cs);
        o0 = y0 * cs - y1 * sn;
        o1 = y1 * cs + y0 * sn;
      } else {
        o0 = y0;
        o1 = y1;
      }
    }
    T[r * ldt + 2 * p] = o0;
    T[r * ldt + 2 * p + 1] = o1;
  }
}

// Writes the rotated, scaled rows in QK format (rows_out) and / or VT format (planes_out); either may be null.
__global__ __launch_bounds__(256) void rope_split_kernel(
    const float* __restrict__ Y, int ldy, const float* __restrict__ xyz, const float* __restrict__ freq,
    float scale, unsigned short* __restrict__ rows_out, int rows_width, unsigned short* __restrict__ planes_out, int B,
    int N, int Npad, int E, int H) {
  extern __shared__ __attribute__((aligned(16))) float T[];
  const int ldt = E + 1;
  const int b = blockIdx.y, n0 = blockIdx.x * RT_ROWS;
  rope_tile_to_lds(T, ldt, Y, ldy, xyz, freq, scale, b, n0, N, E);
  __syncthreads();
  if (rows_out) {
    const int nseg = rows_width >> 3;   // 4 (hi, lo) or 6 (hi, lo, lo2) 16-byte segments per row
    for (int idx = threadIdx.x; idx < RT_ROWS * H * nseg; idx += blockDim.x) {
      const int seg = idx % nseg;
      const int r = (idx / nseg) % RT_ROWS;
      const int h = (idx / nseg) / RT_ROWS;
      const int n = n0 + r;
      if (n >= Npad) continue;
      const int dbase = (seg & 1) * 8;
      const int part = seg >> 1;
      s16x8 out;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int d = dbase + j;
        const float v = (d < HD) ? T[r * ldt + h * HD + d] : 0.f;
        unsigned short hi, lo, lo2;
        split_bf16_3(v, hi, lo, lo2);
        out[j] = (short)(part == 0 ? hi : (part == 1 ? lo : lo2));
      }
      *reinterpret_cast<s16x8*>(rows_out + (((size_t)b * H + h) * Npad + n) * rows_width + seg * 8) = out;
    }
  }
  if (planes_out) {
    for (int idx = threadIdx.x; idx < H * 2 * 16 * 8; idx += blockDim.x) {
      const int seg = idx & 7;
      const int d = (idx >> 3) & 15;
      const int plane = (idx >> 7) & 1;
      const int h = idx >> 8;
      s16x8 out;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = seg * 8 + j;
        const float v = (d < HD) ? T[r * ldt + h * HD + d] : 0.f;
        unsigned short hi, lo;
        split_bf16(v, hi, lo);
        out[j] = (short)(plane ? lo : hi);
      }
      *reinterpret_cast<s16x8*>(planes_out + ((((size_t)b * H + h) * 2 + plane) * 16 + d) * Npad + n0 + seg * 8) = out;
    }
  }
}

// dY[m][c] = scale * R(xyz)^T * sum_s dR[s][b][h][n][d]   (R^T = inverse rotation; identity when xyz == null)
__global__ __launch_bounds__(256) void rope_merge_bwd_kernel(
    const float* __restrict__ dR, int nsplit, const float* __restrict__ xyz, const float* __restrict__ freq,
    float scale, float* __restrict__ dY, int ldy, int B, int N, int Npad, int E, int H) {
  const int half = E >> 1;
  const int third = E / 3;
  const size_t total = (size_t)B * N * half;
  const size_t split_stride = (size_t)B * H * Npad * HDP;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int p = (int)(idx % half);
    const size_t m = idx / half;
    const int n = (int)(m % N), b = (int)(m / N);
    const int c0 = 2 * p, c1 = 2 * p + 1;
    const int h0 = c0 / HD, d0 = c0 - h0 * HD;
    const int h1 = c1 / HD, d1 = c1 - h1 * HD;
    const size_t o0 = (((size_t)b * H + h0) * Npad + n) * HDP + d0;
    const size_t o1 = (((size_t)b * H + h1) * Npad + n) * HDP + d1;
    float g0 = 0.f, g1 = 0.f;
    for (int s = 0; s < nsplit; ++s) {
      g0 += dR[s * split_stride + o0];
      g1 += dR[s * split_stride + o1];
    }
    float y0 = g0, y1 = g1;
    if (xyz) {
      const int axis = c0 / third;
      const int k = (c0 - axis * third) >> 1;
      const float th = xyz[m * 3 + axis] * freq[k];
      float sn, cs;
      sincosf(th, &sn, &cs);
      y0 = cs * g0 + sn * g1;
      y1 = cs * g1 - sn * g0;
    }
    dY[m * ldy + c0] = y0 * scale;
    dY[m * ldy + c1] = y1 * scale;
  }
}

}  // namespace a3d

using namespace a3d;

static int check_rope_args(const char* fn, int B, int N, int Npad, int E, int H) {
  if (B <= 0 || N <= 0 || Npad < N || (Npad % 64) != 0 || E <= 0 || (E % 6) != 0 || H <= 0 || E != H * HD) {
    set_error("%s: bad argument (B=%d N=%d Npad=%d E=%d H=%d; need E == 15*H, E %% 6 == 0, Npad %% 64 == 0)",
              fn, B, N, Npad, E, H);
    return A3D_ERR_ARG;
  }
  return A3D_OK;
}

extern "C" int a3d_rope_split(const float* Y, int ldy, const float* xyz, const float* freq, float scale,
                              void* rows_out, int rows_width, void* planes_out, int B, int N, int Npad, int E, int H,
                              void* stream) {
  int rc = check_rope_args("a3d_rope_split", B, N, Npad, E, H);
  if (rc) return rc;
  if (rows_out && rows_width != VRW && rows_width != QKW) {
    set_error("a3d_rope_split: rows_width must be 32 (hi|lo) or 48 (hi|lo|lo2), got %d", rows_width);
    return A3D_ERR_ARG;
  }
  if (!Y || (!rows_out && !planes_out) || (xyz && !freq)) { set_error("a3d_rope_split: null pointer"); return A3D_ERR_ARG; }
  dim3 grid(Npad / RT_ROWS, B);
  const size_t lds = (size_t)RT_ROWS * (E + 1) * sizeof(float);
  hipLaunchKernelGGL(rope_split_kernel, grid, dim3(256), lds, (hipStream_t)stream, Y, ldy, xyz, freq, scale,
                     (unsigned short*)rows_out, rows_width, (unsigned short*)planes_out, B, N, Npad, E, H);
  return check_launch("a3d_rope_split");
}

extern "C" int a3d_rope_split_qk(const float* Y, int ldy, const float* xyz, const float* freq, float scale,
                                 void* dst, int B, int N, int Npad, int E, int H, void* stream) {
  return a3d_rope_split(Y, ldy, xyz, freq, scale, dst, QKW, nullptr, B, N, Npad, E, H, stream);
}

extern "C" int a3d_split_vt(const float* Y, int ldy, void* dst, int B, int N, int Npad, int E, int H,
                            void* stream) {
  return a3d_rope_split(Y, ldy, nullptr, nullptr, 1.0f, nullptr, 0, dst, B, N, Npad, E, H, stream);
}

extern "C" int a3d_rope_merge_bwd(const float* dR, int nsplit, const float* xyz, const float* freq,
                                  float scale, float* dY, int ldy, int B, int N, int Npad, int E, int H,
                                  void* stream) {
  if (B <= 0 || N <= 0 || Npad < N || E != H * HD || nsplit < 1 || !dR || !dY || (xyz && !freq)) {
    set_error("a3d_rope_merge_bwd: bad argument (B=%d N=%d Npad=%d E=%d H=%d nsplit=%d)", B, N, Npad, E, H,
              nsplit);
    return A3D_ERR_ARG;
  }
  const size_t total = (size_t)B * N * (E / 2);
  const int grid = (int)std::min<size_t>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(rope_merge_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dR, nsplit, xyz,
                     freq, scale, dY, ldy, B, N, Npad, E, H);
  return check_launch("a3d_rope_merge_bwd");
}
