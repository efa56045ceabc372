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
#include <stdlib.h>

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
        fast_sincos(th, &sn, &cs);
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

// T[r][c] (64 rows x E, rotated, scaled, rows >= N zero) -> rows format (width 32 / 48) and / or planes format.
// Each work item splits its 8 values once and stores every part it yields (2-3 row segments, 2 plane segments).
__device__ __forceinline__ void write_operand_formats(const float* T, int ldt, unsigned short* __restrict__ rows_out,
                                                      int rows_width, unsigned short* __restrict__ planes_out, int b, int n0,
                                                      int Npad, int H) {
  if (rows_out) {
    const bool three = rows_width == QKW;
    for (int idx = threadIdx.x; idx < RT_ROWS * H * 2; idx += blockDim.x) {
      const int half = idx & 1;                 // head-dim elements 0-7 / 8-15
      const int r = (idx >> 1) % RT_ROWS;
      const int h = (idx >> 1) / RT_ROWS;
      const int n = n0 + r;
      if (n >= Npad) continue;
      s16x8 ohi, olo, olo2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int d = half * 8 + j;
        const float v = (d < HD) ? T[r * ldt + h * HD + d] : 0.f;
        unsigned short hi, lo, lo2;
        split_bf16_3(v, hi, lo, lo2);
        ohi[j] = (short)hi;
        olo[j] = (short)lo;
        olo2[j] = (short)lo2;
      }
      unsigned short* dst = rows_out + (((size_t)b * H + h) * Npad + n) * rows_width + half * 8;
      *reinterpret_cast<s16x8*>(dst) = ohi;
      *reinterpret_cast<s16x8*>(dst + 16) = olo;
      if (three) *reinterpret_cast<s16x8*>(dst + 32) = olo2;
    }
  }
  if (planes_out) {
    for (int idx = threadIdx.x; idx < H * 16 * 8; idx += blockDim.x) {
      const int seg = idx & 7;                  // 8 consecutive rows (keys)
      const int d = (idx >> 3) & 15;
      const int h = idx >> 7;
      s16x8 ohi, olo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = seg * 8 + j;
        const float v = (d < HD) ? T[r * ldt + h * HD + d] : 0.f;
        unsigned short hi, lo;
        split_bf16(v, hi, lo);
        ohi[j] = (short)hi;
        olo[j] = (short)lo;
      }
      unsigned short* dst = planes_out + ((((size_t)b * H + h) * 2 + 0) * 16 + d) * Npad + n0 + seg * 8;
      *reinterpret_cast<s16x8*>(dst) = ohi;
      *reinterpret_cast<s16x8*>(dst + (size_t)16 * Npad) = olo;
    }
  }
}

// The "16" formats of attention16.hip: rows16 [B][H][Npad][32] fp16 = hi(16) | lo(16) (x = hi + lo, 22 mantissa bits; the
// lo part may be an fp16 subnormal, which the MFMA honours), planes16 [B][H][parts][16][Npad] = fp16 hi (and lo when parts == 2) planes, transposed.
typedef __attribute__((ext_vector_type(2))) _Float16 rh16x2;
typedef __attribute__((ext_vector_type(2))) float rf32x2;
__device__ __forceinline__ unsigned short f2h(float x) { return __builtin_bit_cast(unsigned short, (_Float16)x); }
__device__ __forceinline__ float h2f(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }

// two floats -> (hi, lo) packed fp16 pairs, x = hi + lo: hi = v_cvt_pk_f16_f32 (RNE), lo = fp16(x - hi) by v_fma_mixlo / mixhi_f16
// (the fp16 half enters the fma directly; x * 1.0 - hi is exact in fp32, so the single rounding gives the bits of the
// convert - subtract - convert sequence this replaces: 3 instructions per PAIR instead of ~6 per element; attn_ring.h's lo_f16)
__device__ __forceinline__ void rp_split_f16(float a, float b, unsigned int& hi, unsigned int& lo) {
  typedef __attribute__((ext_vector_type(2))) float rp_f32x2;
  typedef __attribute__((ext_vector_type(2))) _Float16 rp_h16x2;
  hi = __builtin_bit_cast(unsigned int, __builtin_convertvector((rp_f32x2){a, b}, rp_h16x2));
#ifdef A3D_NO_FMA_MIX
  const rp_h16x2 hh = __builtin_bit_cast(rp_h16x2, hi);
  lo = __builtin_bit_cast(unsigned int, __builtin_convertvector((rp_f32x2){a - (float)hh[0], b - (float)hh[1]}, rp_h16x2));
#else
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(lo)
      : "v"(a), "v"(b), "v"(hi));
#endif
}

__device__ __forceinline__ void write_operand_formats16(const float* T, int ldt, unsigned short* __restrict__ rows_out,
                                                        unsigned short* __restrict__ planes_out, int plane_parts, int b,
                                                        int n0, int Npad, int H) {
  typedef __attribute__((ext_vector_type(4))) unsigned int rp_u32x4;
  if (rows_out) {
    // plane_parts & 8: the padded channel 15 of the hi part of every ROW is written as 1.0 -- the value rows of the rows-only operand
    // set (round 6): the forward forms its V^T fragments from these rows with transposed LDS reads, and its PV MFMA then accumulates
    // the softmax denominator in output channel 15 exactly as with the value planes; the backward kernels contract that channel
    // against the zero-padded channel 15 of dO
    const float row_pad = (plane_parts & 8) ? 1.0f : 0.f;
    for (int idx = threadIdx.x; idx < RT_ROWS * H * 2; idx += blockDim.x) {
      const int half = idx & 1;
      const int r = (idx >> 1) % RT_ROWS;
      const int h = (idx >> 1) / RT_ROWS;
      const int n = n0 + r;
      if (n >= Npad) continue;
      unsigned int ohi[4], olo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int d0 = half * 8 + 2 * j, d1 = d0 + 1;
        const float v0 = (d0 < HD) ? T[r * ldt + h * HD + d0] : 0.f;
        const float v1 = (d1 < HD) ? T[r * ldt + h * HD + d1] : row_pad;       // d1 = 15 is the only padded channel
        rp_split_f16(v0, v1, ohi[j], olo[j]);
      }
      unsigned short* dst = rows_out + (((size_t)b * H + h) * Npad + n) * 32 + half * 8;
      *reinterpret_cast<rp_u32x4*>(dst) = rp_u32x4{ohi[0], ohi[1], ohi[2], ohi[3]};
      *reinterpret_cast<rp_u32x4*>(dst + 16) = rp_u32x4{olo[0], olo[1], olo[2], olo[3]};
    }
  }
  if (planes_out) {
    // plane_parts: 1 or 2 planes (hi [, lo]); + 4: the padded channel 15 of the hi plane is written as 1.0 -- the value planes
    // of the forward, whose PV MFMA then accumulates the softmax denominator in output channel 15 (attention16.hip)
    const int parts = plane_parts & 3;
    const bool ones = (plane_parts & 4) != 0;
    for (int idx = threadIdx.x; idx < H * 16 * 8; idx += blockDim.x) {
      const int seg = idx & 7;
      const int d = (idx >> 3) & 15;
      const int h = idx >> 7;
      const float pad = (ones && d == HD) ? 1.0f : 0.f;
      unsigned int o[4], o2[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v0 = (d < HD) ? T[(seg * 8 + 2 * j) * ldt + h * HD + d] : pad;
        const float v1 = (d < HD) ? T[(seg * 8 + 2 * j + 1) * ldt + h * HD + d] : pad;
        rp_split_f16(v0, v1, o[j], o2[j]);
      }
      unsigned short* dst = planes_out + ((((size_t)b * H + h) * parts) * 16 + d) * Npad + n0 + seg * 8;
      *reinterpret_cast<rp_u32x4*>(dst) = rp_u32x4{o[0], o[1], o[2], o[3]};
      if (parts == 2) *reinterpret_cast<rp_u32x4*>(dst + (size_t)16 * Npad) = rp_u32x4{o2[0], o2[1], o2[2], o2[3]};
    }
  }
}

// Writes the rotated, scaled rows in rows format (rows_out) and / or planes format (planes_out); either may be null.
__global__ __launch_bounds__(256) void rope_split_kernel(
    const float* __restrict__ Y, int ldy, const float* __restrict__ xyz, const float* __restrict__ freq,
    float scale, unsigned short* __restrict__ rows_out, int rows_width, unsigned short* __restrict__ planes_out, int B,
    int N, int Npad, int E, int H, int fmt16) {
  extern __shared__ __attribute__((aligned(16))) float T[];
  const int ldt = E + 1;
  const int b = blockIdx.y, n0 = blockIdx.x * RT_ROWS;
  rope_tile_to_lds(T, ldt, Y, ldy, xyz, freq, scale, b, n0, N, E);
  __syncthreads();
  if (fmt16) write_operand_formats16(T, ldt, rows_out, planes_out, rows_width, b, n0, Npad, H);   // rows_width = plane parts
  else write_operand_formats(T, ldt, rows_out, rows_width, planes_out, b, n0, Npad, H);
}

// ---------------------------------------------------------------------------------------------------------------
// Projection + RoPE + operand formatting in one kernel: the projected rows never go to HBM.
//   Y[64 rows][E] = X[64][K] W_blk[E][K]^T + bias_blk  (exact fp32 MFMA 16x16x4, as linear.hip)
//   -> scale, rotate by xyz (in LDS) -> rows / planes operand formats.
// blockIdx.y selects the output block (0: W rows [0, E), 1: W rows [E, 2E)) -- q | k of a packed q,k projection or
// k | v of a packed k,v projection (multihead_custom_attention.py:251-303), each with its own rotation / scale / formats.
struct ProjBlock {
  const float* xyz;        // [B][N][3] or null (no rotation)
  float scale;
  unsigned short* rows;    // rows format or null
  int rows_width;          // 32 or 48
  unsigned short* planes;  // planes format or null
};
constexpr int PR_KC = 64;    // K chunk
constexpr int PR_LD = 68;    // padded LDS row stride (floats)

// EC: the channel count as a compile-time constant (60 / 120: the two models' widths; 0 = the run-time E).  The rotation loop
// divides its item index by E / 2 and its channel by E / 3 -- with a run-time E that is two ~25-instruction integer divisions per
// (row, pair) item, about half of this kernel's dynamic VALU instructions, and the kernel is issue-bound (four workgroups per CU,
// ~1300 instructions per wave and tile): the specialisations turn them into multiply-shifts and give the loops constant bounds.
// KEQ: the input width K equals E (every in-projection of the two models): one K chunk for E = 60, constant k-step counts.
template <int NT, int EC, bool KEQ>   // 16-column output tiles: 4 (E <= 64) or 8 (E <= 128)
__global__ __launch_bounds__(256) void proj_rope_split_kernel(
    const float* __restrict__ X, int ldx, const float* __restrict__ W, int ldw, const float* __restrict__ bias, int K,
    ProjBlock blk0, ProjBlock blk1, const float* __restrict__ freq, int B, int N, int Npad, int E_rt, int H_rt, int fmt16) {
  const int E = EC > 0 ? EC : E_rt, H = EC > 0 ? EC / HD : H_rt;
  if (KEQ) K = EC;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                          // [64][PR_LD]
  float* Ws = smem + RT_ROWS * PR_LD;        // [NT * 16][PR_LD]
  float* T = smem;                           // [64][E + 1], aliases Xs / Ws after the contraction
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, n0 = blockIdx.x * RT_ROWS;
  const ProjBlock blk = blockIdx.y ? blk1 : blk0;
  const float* Wb = W + (size_t)blockIdx.y * E * ldw;
  const float* bb = bias ? bias + blockIdx.y * E : nullptr;
  const bool w_vec = ((((uintptr_t)Wb) & 15) == 0) && ((ldw & 3) == 0);

  f32x4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += PR_KC) {
    // stage X[64][64] and W_blk[NT*16][64] (zero beyond N / E / K)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = t + i * 256;
      const int r = idx >> 4, c = (idx & 15) * 4;
      const int n = n0 + r, k = k0 + c;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N && k < K) v = *reinterpret_cast<const float4*>(X + ((size_t)b * N + n) * ldx + k);
      *reinterpret_cast<float4*>(&Xs[r * PR_LD + c]) = v;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int idx = t + i * 256;
      const int j = idx >> 4, c = (idx & 15) * 4;
      const int k = k0 + c;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < E && k < K) {
        const float* wp = Wb + (size_t)j * ldw + k;
        // parameters living in a flat optimizer buffer are only 4-byte aligned
        v = w_vec ? *reinterpret_cast<const float4*>(wp) : make_float4(wp[0], wp[1], wp[2], wp[3]);
      }
      *reinterpret_cast<float4*>(&Ws[j * PR_LD + c]) = v;
    }
    __syncthreads();
    const int ksteps = min(16, (K - k0 + 3) >> 2);
    for (int kk = 0; kk < ksteps; ++kk) {
      const float a = Xs[(wave * 16 + li) * PR_LD + kk * 4 + g];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma_f32_16x16x4(a, Ws[(nt * 16 + li) * PR_LD + kk * 4 + g], acc[nt]);
    }
    __syncthreads();
  }
  // Y tile (+ bias, * scale) -> T; rows n >= N stay zero (finite padding)
  const int ldt = E + 1;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = nt * 16 + li;
    if (c >= E) continue;
    const float bv = bb ? bb[c] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 16 + g * 4 + r;
      T[row * ldt + c] = (n0 + row < N) ? (acc[nt][r] + bv) * blk.scale : 0.f;
    }
  }
  __syncthreads();
  if (blk.xyz) {
    const int half = E >> 1, third = E / 3;
    for (int idx = t; idx < RT_ROWS * half; idx += 256) {
      const int r = idx / half, p = idx - r * half;
      const int n = n0 + r;
      if (n >= N) continue;
      const int c = 2 * p;
      const int axis = c / third;
      const int kf = (c - axis * third) >> 1;
      const float th = blk.xyz[((size_t)b * N + n) * 3 + axis] * freq[kf];
      float sn, cs;
      fast_sincos(th, &sn, &cs);
      const float y0 = T[r * ldt + c], y1 = T[r * ldt + c + 1];
      T[r * ldt + c] = y0 * cs - y1 * sn;
      T[r * ldt + c + 1] = y1 * cs + y0 * sn;
    }
    __syncthreads();
  }
  if (fmt16) write_operand_formats16(T, ldt, blk.rows, blk.planes, blk.rows_width, b, n0, Npad, H);   // rows_width = plane parts
  else write_operand_formats(T, ldt, blk.rows, blk.rows_width, blk.planes, b, n0, Npad, H);
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
      fast_sincos(th, &sn, &cs);
      y0 = cs * g0 + sn * g1;
      y1 = cs * g1 - sn * g0;
    }
    dY[m * ldy + c0] = y0 * scale;
    dY[m * ldy + c1] = y1 * scale;
  }
}

// The same per (row, pair) item with 32-bit indices and the channel count as a compile-time constant (grid.y = sample): the generic
// kernel above spends most of its instructions on four 64-bit integer divisions per item (idx -> pair / row / sample) and two
// run-time divisions by E / 3 and 15; here idx % (EC / 2), / 15 and / (EC / 3) are multiply-shifts.  Same loads, same arithmetic.
template <int EC>
__global__ __launch_bounds__(256) void rope_merge_bwd_pairs_kernel(
    const float* __restrict__ dR, int nsplit, const float* __restrict__ xyz, const float* __restrict__ freq,
    float scale, float* __restrict__ dY, int ldy, int B, int N, int Npad) {
  constexpr int H = EC / HD, half = EC / 2, third = EC / 3;
  const int b = blockIdx.y;
  const unsigned int idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= (unsigned int)N * half) return;
  const int n = (int)(idx / half), p = (int)(idx - (unsigned int)n * half);
  const int c0 = 2 * p, c1 = c0 + 1;
  const int h0 = c0 / HD, d0 = c0 - h0 * HD;
  const int h1 = c1 / HD, d1 = c1 - h1 * HD;
  const size_t split_stride = (size_t)B * H * Npad * HDP;
  const size_t o0 = (((size_t)b * H + h0) * Npad + n) * HDP + d0;
  const size_t o1 = (((size_t)b * H + h1) * Npad + n) * HDP + d1;
  float g0 = 0.f, g1 = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    g0 += dR[s * split_stride + o0];
    g1 += dR[s * split_stride + o1];
  }
  const size_t m = (size_t)b * N + n;
  float y0 = g0, y1 = g1;
  if (xyz) {
    const int axis = c0 / third;
    const int k = (c0 - axis * third) >> 1;
    float sn, cs;
    fast_sincos(xyz[m * 3 + axis] * freq[k], &sn, &cs);
    y0 = cs * g0 + sn * g1;
    y1 = cs * g1 - sn * g0;
  }
  dY[m * ldy + c0] = y0 * scale;
  dY[m * ldy + c1] = y1 * scale;
}

// The same, one THREAD per row for the two models' widths (EC = 60 / 120 channels, H = EC / 15 heads).  The kernel above spends
// its time on four 64-bit integer divisions per channel pair (item -> row / pair / sample) and reads each head's 15 channels of a
// row as a separate 60-byte piece; here a thread sums its row's H x 16-float records over the splits with float4 loads
// (consecutive threads = consecutive rows = consecutive 64-byte records of a head plane), rotates the EC / 2 pairs in registers
// (no index arithmetic left: every channel index is a constant after unrolling) and writes the row as EC / 4 float4.
template <int EC>
__global__ __launch_bounds__(256) void rope_merge_bwd_rows_kernel(
    const float* __restrict__ dR, int nsplit, const float* __restrict__ xyz, const float* __restrict__ freq,
    float scale, float* __restrict__ dY, int ldy, int B, int N, int Npad) {
  constexpr int H = EC / HD, third = EC / 3, NF = third / 2;
  const int b = blockIdx.y;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const size_t split_stride = (size_t)B * H * Npad * HDP;
  float g[EC];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const float* src = dR + (((size_t)b * H + h) * Npad + n) * HDP;
    float4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(src + q * 4);
    for (int s = 1; s < nsplit; ++s) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 w = *reinterpret_cast<const float4*>(src + s * split_stride + q * 4);
        v[q].x += w.x; v[q].y += w.y; v[q].z += w.z; v[q].w += w.w;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float e[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (q * 4 + j < HD) g[h * HD + q * 4 + j] = e[j];
    }
  }
  const size_t m = (size_t)b * N + n;
  if (xyz) {
    float fr[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) fr[k] = freq[k];
#pragma unroll
    for (int axis = 0; axis < 3; ++axis) {
      const float x = xyz[m * 3 + axis];
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        const int c0 = axis * third + 2 * k;
        float sn, cs;
        fast_sincos(x * fr[k], &sn, &cs);
        const float g0 = g[c0], g1 = g[c0 + 1];
        g[c0] = cs * g0 + sn * g1;
        g[c0 + 1] = cs * g1 - sn * g0;
      }
    }
  }
  float* dst = dY + m * ldy;
#pragma unroll
  for (int q = 0; q < EC / 4; ++q)
    *reinterpret_cast<float4*>(dst + q * 4) = make_float4(g[q * 4] * scale, g[q * 4 + 1] * scale, g[q * 4 + 2] * scale, g[q * 4 + 3] * scale);
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

static int rope_split_launch(const char* fn, const float* Y, int ldy, const float* xyz, const float* freq, float scale,
                             void* rows_out, int rows_width, void* planes_out, int B, int N, int Npad, int E, int H,
                             int fmt16, void* stream) {
  int rc = check_rope_args(fn, B, N, Npad, E, H);
  if (rc) return rc;
  if (fmt16 ? ((rows_width & 3) != 1 && (rows_width & 3) != 2) || (rows_width & ~15) : (rows_out && rows_width != VRW && rows_width != QKW)) {
    set_error(fmt16 ? "%s: plane parts must be 1 or 2 (+ 4: ones channel in the hi plane, + 8: in the rows), got %d" : "%s: rows_width must be 32 (hi|lo) or 48 (hi|lo|lo2), got %d", fn,
              rows_width);
    return A3D_ERR_ARG;
  }
  if (!Y || (!rows_out && !planes_out) || (xyz && !freq)) { set_error("%s: null pointer", fn); return A3D_ERR_ARG; }
  dim3 grid(Npad / RT_ROWS, B);
  const size_t lds = (size_t)RT_ROWS * (E + 1) * sizeof(float);
  hipLaunchKernelGGL(rope_split_kernel, grid, dim3(256), lds, (hipStream_t)stream, Y, ldy, xyz, freq, scale,
                     (unsigned short*)rows_out, rows_width, (unsigned short*)planes_out, B, N, Npad, E, H, fmt16);
  return check_launch(fn);
}

extern "C" int a3d_rope_split(const float* Y, int ldy, const float* xyz, const float* freq, float scale,
                              void* rows_out, int rows_width, void* planes_out, int B, int N, int Npad, int E, int H,
                              void* stream) {
  return rope_split_launch("a3d_rope_split", Y, ldy, xyz, freq, scale, rows_out, rows_width, planes_out, B, N, Npad, E, H, 0,
                           stream);
}

extern "C" int a3d_rope_split16(const float* Y, int ldy, const float* xyz, const float* freq, float scale, void* rows_out,
                                void* planes_out, int plane_parts, int B, int N, int Npad, int E, int H, void* stream) {
  return rope_split_launch("a3d_rope_split16", Y, ldy, xyz, freq, scale, rows_out, plane_parts, planes_out, B, N, Npad, E, H,
                           1, stream);
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
  // the row-per-thread kernel serves the two models' widths when rows of dY can be written as float4 (dR is HDP = 16 floats per
  // record, 16-byte aligned by construction of the attention backward's partial buffers)
  // Measured on MI355X (round 5, gpurun r05a): 31.9 us vs 27.1 us per call for E = 60 and 16.4 vs 9.1 us for E = 120 -- the
  // row-per-thread kernel keeps EC floats per thread live and runs at a fraction of the pair-per-thread kernel's occupancy.
  // Kept behind A3D_ROPE_MERGE_ROWS=1 for the record; the pair kernel is the default again.
  static const bool rows_on = getenv("A3D_ROPE_MERGE_ROWS") && atoi(getenv("A3D_ROPE_MERGE_ROWS")) != 0;
  const bool rows_ok = rows_on && ((E == 60 && H == 4) || (E == 120 && H == 8)) && (ldy & 3) == 0 && ((((uintptr_t)dY) | ((uintptr_t)dR)) & 15) == 0 &&
                       (Npad * HDP) % 4 == 0;
  if (rows_ok) {
    const dim3 grid(cdiv(N, 256), B);
    if (E == 60)
      hipLaunchKernelGGL(rope_merge_bwd_rows_kernel<60>, grid, dim3(256), 0, (hipStream_t)stream, dR, nsplit, xyz, freq, scale, dY, ldy, B, N, Npad);
    else
      hipLaunchKernelGGL(rope_merge_bwd_rows_kernel<120>, grid, dim3(256), 0, (hipStream_t)stream, dR, nsplit, xyz, freq, scale, dY, ldy, B, N, Npad);
    return check_launch("a3d_rope_merge_bwd");
  }
  // opt-in (A3D_ROPE_MERGE_PAIRS=1).  Measured in the eager keypose step (gpurun r05r, same box, kernel trace): 26.2 vs 27.1 us on
  // average over the step's 18 calls, 36.3 vs 37.3 us for the 12 context-row calls -- the integer divisions were NOT this kernel's
  // cost; it moves 131 MB per context-row call in 37 us = 3.5 TB/s (64-byte head records in, 240-byte rows out).  Not worth a default.
  static const bool pairs_on = getenv("A3D_ROPE_MERGE_PAIRS") && atoi(getenv("A3D_ROPE_MERGE_PAIRS")) != 0;
  if (pairs_on && ((E == 60 && H == 4) || (E == 120 && H == 8)) && (size_t)N * (E / 2) < (1u << 31) && B <= 65535) {
    const dim3 grid2((unsigned)(((size_t)N * (E / 2) + 255) / 256), B);
    if (E == 60)
      hipLaunchKernelGGL(rope_merge_bwd_pairs_kernel<60>, grid2, dim3(256), 0, (hipStream_t)stream, dR, nsplit, xyz, freq, scale, dY, ldy, B, N, Npad);
    else
      hipLaunchKernelGGL(rope_merge_bwd_pairs_kernel<120>, grid2, dim3(256), 0, (hipStream_t)stream, dR, nsplit, xyz, freq, scale, dY, ldy, B, N, Npad);
    return check_launch("a3d_rope_merge_bwd");
  }
  const size_t total = (size_t)B * N * (E / 2);
  const int grid = (int)std::min<size_t>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(rope_merge_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dR, nsplit, xyz,
                     freq, scale, dY, ldy, B, N, Npad, E, H);
  return check_launch("a3d_rope_merge_bwd");
}

static int proj_rope_split_launch(const char* fn, const float* X, int ldx, const float* W, int ldw, const float* bias, int K,
                                  const float* xyz0, float scale0, void* rows0, int rows0_width, void* planes0,
                                  const float* xyz1, float scale1, void* rows1, int rows1_width, void* planes1,
                                  const float* freq, int B, int N, int Npad, int E, int H, int fmt16, void* stream) {
  int rc = check_rope_args(fn, B, N, Npad, E, H);
  if (rc) return rc;
  const bool two = rows1 || planes1;
  auto parts_ok = [](int p) { return ((p & 3) == 1 || (p & 3) == 2) && !(p & ~15); };      // | 4: ones channel in the hi plane, | 8: in the rows
  const bool w0_ok = fmt16 ? parts_ok(rows0_width) : (!rows0 || rows0_width == VRW || rows0_width == QKW);
  const bool w1_ok = fmt16 ? (!two || parts_ok(rows1_width)) : (!rows1 || rows1_width == VRW || rows1_width == QKW);
  if (!X || !W || K <= 0 || (K & 3) || (ldx & 3) || (((uintptr_t)X) & 15) || E > 128 ||
      (!rows0 && !planes0) || ((xyz0 || xyz1) && !freq) || !w0_ok || !w1_ok) {
    set_error("%s: bad argument (K=%d ldx=%d must be multiples of 4, X 16-byte aligned, E=%d <= 128, %s)", fn, K, ldx, E,
              fmt16 ? "plane parts 1 or 2" : "rows widths 32 or 48");
    return A3D_ERR_ARG;
  }
  ProjBlock b0{xyz0, scale0, (unsigned short*)rows0, rows0_width, (unsigned short*)planes0};
  ProjBlock b1{xyz1, scale1, (unsigned short*)rows1, rows1_width, (unsigned short*)planes1};
  const int NT = E <= 64 ? 4 : 8;
  // (round 4 measured a persistent, weight-resident variant of this kernel for K <= 64 -- weights kept in LDS, next tile's rows
  // prefetched in registers: 0.149 ms against 0.138 ms for the k|v block of the bench shape, gpurun r04h; deleted.)
  dim3 grid(Npad / RT_ROWS, two ? 2 : 1, B);
  const size_t lds = std::max((size_t)(RT_ROWS + NT * 16) * PR_LD, (size_t)RT_ROWS * (E + 1)) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)proj_rope_split_kernel<8, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void*)proj_rope_split_kernel<8, 120, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    (void)hipFuncSetAttribute((const void*)proj_rope_split_kernel<8, 120, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    attr_set = true;
  }
#define A3D_PRS(NTV, ECV, KQ)                                                                                                   \
  hipLaunchKernelGGL((proj_rope_split_kernel<NTV, ECV, KQ>), grid, dim3(256), lds, (hipStream_t)stream, X, ldx, W, ldw, bias, K, b0, b1, \
                     freq, B, N, Npad, E, H, fmt16)
  if (E == 60 && H == 4 && K == 60) A3D_PRS(4, 60, true);             // Act3D
  else if (E == 60 && H == 4) A3D_PRS(4, 60, false);
  else if (E == 120 && H == 8 && K == 120) A3D_PRS(8, 120, true);      // the trajectory diffusion model
  else if (E == 120 && H == 8) A3D_PRS(8, 120, false);
  else if (NT == 4) A3D_PRS(4, 0, false);
  else A3D_PRS(8, 0, false);
#undef A3D_PRS
  return check_launch(fn);
}

extern "C" int a3d_proj_rope_split(const float* X, int ldx, const float* W, int ldw, const float* bias, int K,
                                   const float* xyz0, float scale0, void* rows0, int rows0_width, void* planes0,
                                   const float* xyz1, float scale1, void* rows1, int rows1_width, void* planes1,
                                   const float* freq, int B, int N, int Npad, int E, int H, void* stream) {
  return proj_rope_split_launch("a3d_proj_rope_split", X, ldx, W, ldw, bias, K, xyz0, scale0, rows0, rows0_width, planes0,
                                xyz1, scale1, rows1, rows1_width, planes1, freq, B, N, Npad, E, H, 0, stream);
}

extern "C" int a3d_proj_rope_split16(const float* X, int ldx, const float* W, int ldw, const float* bias, int K,
                                     const float* xyz0, float scale0, void* rows0, void* planes0, int parts0,
                                     const float* xyz1, float scale1, void* rows1, void* planes1, int parts1,
                                     const float* freq, int B, int N, int Npad, int E, int H, void* stream) {
  return proj_rope_split_launch("a3d_proj_rope_split16", X, ldx, W, ldw, bias, K, xyz0, scale0, rows0, parts0, planes0, xyz1,
                                scale1, rows1, parts1, planes1, freq, B, N, Npad, E, H, 1, stream);
}
