// fp8 (OCP e4m3fn) attention forward for gfx950 -- the OPT-IN mode BASELINE.json configs[4] names ("fp8 MFMA attention, 4
// ghost-point levels at 10k points"); ops.ATTN_MODE = "fp8" (A3D_ATTN_MODE=fp8).  Not the default and not a parity path:
// its stated tolerance is e4m3's (3 mantissa bits, relative rounding 2^-4), see DESIGN.md section 4 and
// tests/test_attn8_gpu.py; everything the 1e-3 bar applies to runs on the split-fp16 family (attention16.hip).
//
// Reference semantics as attention16.hip (multihead_custom_attention.py:355-447); the backward of a forward computed here is
// the split-fp16 backward evaluated with this forward's O and LSE (forward quantised, backward in 16-bit operands).
//
// Per 64 keys x 16 queries a wave issues 6 MFMAs (v_mfma_f32_16x16x32_fp8_fp8; 14 in the fp16 family): 4 score tiles and 2
// PV products.  With head dim 16 only half of the K = 32 contraction of a score MFMA is needed; the other half carries
// the second fp8 part of q for free:  [k | k] . [q_hi | q_lo] = k (q_hi + q_lo)  -- q enters with 7 bits, k with 4.
//
// Scaling ("per-tile amax", the tile being one (sample, head)): attn8_amax_kernel reduces max |k|, max |q|, max |v| per
// (b, h); all scales are POWERS OF TWO (exact), and q and k get opposite ones, k8 = k 2^ek, q8 = q 2^-ek with ek balancing
// the two maxima, so the MFMA result is the logit itself and needs no per-score rescale.  e4m3 is a floating-point format:
// its relative precision is the same over 2^-6 .. 448, so a finer (per 64-key) scale would buy nothing for K and V (it is
// what integer or block formats need); what the scale has to guarantee is only that nothing overflows 448 and that the
// operands sit in the normal range.  v8 = v 2^ev with max |v8| in [128, 256); the value planes carry 1.0 in the padded
// channel 15, so the softmax denominator is accumulated on the MFMA from the same rounded weights as the numerator.
// Weights are formed as p = 2^(s - m + 5) <= 2^8 with the lazy running max of the fp16 kernels (revised when a score
// exceeds it by 2^3): weights down to 2^-11 of the running maximum stay in e4m3's normal range, 2^-14 in its subnormals.
//
// Operand formats ("8" formats, written by attn8_pack_kernel from the "16" formats of the projection kernels):
//   K8 [B][H][Sp][16] bytes   rows,   V8 [B][H][16][Sp] bytes   planes (channel 15 = 1.0),   amax [B][H][4] u32 (float bits)
// Q stays in the rows16 format: each workgroup converts its own queries in the prologue.
// Staging: ONE LDS-DMA instruction per wave and 64-key chunk (lanes 0-15: 16 key rows, lanes 16-31: 4 value channels x 64
// keys), ring of 4 buffers as in attention16.hip.
#include "attn_ring.h"
#include "../../include/act3d_hip.h"
#include <stdlib.h>

namespace a3d {

constexpr float P8_OFF = 5.0f;       // p = 2^(s - m + P8_OFF)
constexpr float P8_THR = 3.0f;       // lazy rescale threshold: p <= 2^(P8_OFF + P8_THR) = 256 < 448
constexpr int F8_NB = 4;
constexpr int F8_REGION = 768;       // bytes of LDS per wave and ring slot (512 used; see region_base)
constexpr int F8_SLOT = 4 * F8_REGION + 64;

__device__ __forceinline__ f32x4 mfma_f8(unsigned long long a, unsigned long long b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)a, (long)b, c, 0, 0, 0);
}
// four floats -> four e4m3 bytes (round to nearest even; the callers guarantee |x| <= 448)
__device__ __forceinline__ unsigned int pk_f8x4(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned int)w;
}
__device__ __forceinline__ float f8_byte(unsigned int w, int i) {
  return i == 0 ? __builtin_amdgcn_cvt_f32_fp8((int)w, 0) : i == 1 ? __builtin_amdgcn_cvt_f32_fp8((int)w, 1)
       : i == 2 ? __builtin_amdgcn_cvt_f32_fp8((int)w, 2) : __builtin_amdgcn_cvt_f32_fp8((int)w, 3);
}
__device__ __forceinline__ float clamp448(float x) { return __builtin_amdgcn_fmed3f(x, -448.f, 448.f); }
__device__ __forceinline__ float half_bits_to_float(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }

// exponent e with amax = f 2^e, f in [0.5, 1)  (amax = 0 -> 0)
__device__ __forceinline__ int exponent_of(unsigned int amax_bits) {
  const float a = __uint_as_float(amax_bits);
  if (!(a > 0.f)) return 0;
  int e;
  frexpf(a, &e);
  return e;
}
// ek: k8 = k 2^ek, q8 = q 2^-ek, both maxima below 2^ceil((e_k + e_q) / 2);  ev: v8 = v 2^ev with max |v8| in [128, 256)
__device__ __forceinline__ void scales_of(const unsigned int* __restrict__ amax, size_t bh, int& ek, int& ev) {
  const int e_k = exponent_of(amax[bh * 4 + 0]), e_q = exponent_of(amax[bh * 4 + 1]), e_v = exponent_of(amax[bh * 4 + 2]);
  const int d = e_q - e_k;
  ek = (d >= 0) ? (d >> 1) : -((-d + 1) >> 1);             // floor(d / 2)
  ev = 8 - e_v;
}
// LDS byte offset of wave w's region inside a ring slot: odd regions are shifted by 64 B so that the key rows of the two
// 16-row regions a score-tile fragment read touches fall on complementary banks
__device__ __forceinline__ int region_base(int w) { return w * F8_REGION + (w & 1) * 64; }

// ------------------------------------------------------------------------------------------------ amax, pack
// amax over the hi parts of the "16" operands (|x| <= |hi| (1 + 2^-11): the scales keep a factor >= 1.75 of headroom)
__global__ __launch_bounds__(256) void attn8_amax_kernel(
    const unsigned short* __restrict__ Qr, const unsigned short* __restrict__ Kr, const unsigned short* __restrict__ Vp,
    unsigned int* __restrict__ amax, int B, int H, int Lq, int Lqp, int S, int Sp) {
  const size_t bh = blockIdx.y;
  const int t = threadIdx.x;
  float mk = 0.f, mq = 0.f, mv = 0.f;
  // rows: thread = (row, 8-half segment of the hi part)
  for (int idx = blockIdx.x * 256 + t; idx < S * 2; idx += gridDim.x * 256) {
    const int row = idx >> 1, seg = idx & 1;
    const s16x8 v = *reinterpret_cast<const s16x8*>(Kr + (bh * Sp + row) * 32 + seg * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) mk = fmaxf(mk, fabsf(half_bits_to_float((unsigned short)v[i])));
  }
  for (int idx = blockIdx.x * 256 + t; idx < Lq * 2; idx += gridDim.x * 256) {
    const int row = idx >> 1, seg = idx & 1;
    const s16x8 v = *reinterpret_cast<const s16x8*>(Qr + (bh * Lqp + row) * 32 + seg * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) mq = fmaxf(mq, fabsf(half_bits_to_float((unsigned short)v[i])));
  }
  // value hi plane [15 real channels][Sp]; 8 keys per thread (pads are zero)
  const int per_ch = Sp / 8;
  for (int idx = blockIdx.x * 256 + t; idx < HD * per_ch; idx += gridDim.x * 256) {
    const int ch = idx / per_ch, k8 = idx - ch * per_ch;
    const s16x8 v = *reinterpret_cast<const s16x8*>(Vp + ((bh * 2 + 0) * 16 + ch) * Sp + k8 * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) mv = fmaxf(mv, fabsf(half_bits_to_float((unsigned short)v[i])));
  }
  __shared__ float red[3][4];
  const int lane = t & 63, wave = t >> 6;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    mk = fmaxf(mk, __shfl_xor(mk, o, 64));
    mq = fmaxf(mq, __shfl_xor(mq, o, 64));
    mv = fmaxf(mv, __shfl_xor(mv, o, 64));
  }
  if (lane == 0) { red[0][wave] = mk; red[1][wave] = mq; red[2][wave] = mv; }
  __syncthreads();
  if (t < 3) {
    const float m = fmaxf(fmaxf(red[t][0], red[t][1]), fmaxf(red[t][2], red[t][3]));
    atomicMax(&amax[bh * 4 + t], __float_as_uint(m));        // non-negative floats order like their bit patterns
  }
}

__global__ __launch_bounds__(256) void attn8_pack_kernel(
    const unsigned short* __restrict__ Kr, const unsigned short* __restrict__ Vp, const unsigned int* __restrict__ amax,
    unsigned char* __restrict__ K8, unsigned char* __restrict__ V8, int B, int H, int S, int Sp) {
  const size_t bh = blockIdx.y;
  const int t = threadIdx.x;
  int ek, ev;
  scales_of(amax, bh, ek, ev);
  const float sk = ldexpf(1.0f, ek), sv = ldexpf(1.0f, ev);
  // K rows: one thread per key row (64 B in, 16 B out)
  for (int row = blockIdx.x * 256 + t; row < Sp; row += gridDim.x * 256) {
    const unsigned short* src = Kr + (bh * Sp + row) * 32;
    const s16x8 h0 = *reinterpret_cast<const s16x8*>(src), h1 = *reinterpret_cast<const s16x8*>(src + 8);
    const s16x8 l0 = *reinterpret_cast<const s16x8*>(src + 16), l1 = *reinterpret_cast<const s16x8*>(src + 24);
    float x[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      x[i] = clamp448((half_bits_to_float((unsigned short)h0[i]) + half_bits_to_float((unsigned short)l0[i])) * sk);
      x[8 + i] = clamp448((half_bits_to_float((unsigned short)h1[i]) + half_bits_to_float((unsigned short)l1[i])) * sk);
    }
    u32x4_ o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pk_f8x4(x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]);
    *reinterpret_cast<u32x4_*>(K8 + (bh * Sp + row) * 16) = o;
  }
  // V planes: thread = (channel, 8 consecutive keys): 2 x 16 B in, 8 B out
  const int per_ch = Sp / 8;
  for (int idx = blockIdx.x * 256 + t; idx < 16 * per_ch; idx += gridDim.x * 256) {
    const int ch = idx / per_ch, k8 = idx - ch * per_ch;
    const s16x8 hi = *reinterpret_cast<const s16x8*>(Vp + ((bh * 2 + 0) * 16 + ch) * Sp + k8 * 8);
    const s16x8 lo = *reinterpret_cast<const s16x8*>(Vp + ((bh * 2 + 1) * 16 + ch) * Sp + k8 * 8);
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float v = half_bits_to_float((unsigned short)hi[i]) + half_bits_to_float((unsigned short)lo[i]);
      x[i] = (ch == HD) ? v : clamp448(v * sv);              // channel 15: the ones channel (1.0 for real keys, 0 for pads)
    }
    uint2 o;
    o.x = pk_f8x4(x[0], x[1], x[2], x[3]);
    o.y = pk_f8x4(x[4], x[5], x[6], x[7]);
    *reinterpret_cast<uint2*>(V8 + (bh * 16 + ch) * Sp + k8 * 8) = o;
  }
}

// ------------------------------------------------------------------------------------------------ forward
// Same tiling as attn16_fwd_kernel: one workgroup = 4 waves = 64 QT queries of one (b, h); scores transposed (S^T = K Q^T)
// with the key rows of the two 16x16 tiles of a 32-key half interleaved, so that after exp2 a lane holds the 8 consecutive
// keys the P operand of the PV MFMA wants.
template <int QT>
__global__ __launch_bounds__(256, 2) void attn8_fwd_kernel(
    const unsigned short* __restrict__ Qr, const unsigned char* __restrict__ K8, const unsigned char* __restrict__ V8,
    const unsigned int* __restrict__ amax, const unsigned char* __restrict__ kmask, float* __restrict__ O,
    float* __restrict__ LSE2, float* __restrict__ Op, float* __restrict__ Mp, float* __restrict__ Lp, int B, int H, int Lq,
    int Lqp, int S, int Sp, int nsplit) {
  __shared__ __attribute__((aligned(16))) unsigned char ring[F8_NB][F8_SLOT];
  __shared__ unsigned int maskW[MASKW];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  constexpr int QW = 64 * QT;
  const int tiles_x = (Lqp + QW - 1) / QW;
  int group, within;
  if (!xcd_decode(tiles_x * nsplit, B * H, group, within)) return;
  const int b = group / H, h = group - b * H;
  const int sp = within / tiles_x;
  const int E = H * HD;
  const int qbase = (within - sp * tiles_x) * QW + wave * (16 * QT);
  const size_t bh = (size_t)b * H + h;
  const bool any_masked = (kmask != nullptr) || (Sp != S);
  if (any_masked) {
    build_key_mask(maskW, kmask, b, S, Sp);
    __syncthreads();
  }
  int ek, ev;
  scales_of(amax, bh, ek, ev);

  // queries: rows16 (fp16 hi | lo) -> q 2^-ek as two fp8 parts; lane group g carries [hi ch 0-7 | hi 8-15 | lo 0-7 | lo 8-15]
  unsigned long long qf[QT];
  bool active[QT];
  bool any_active = false;
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    const int q0 = qbase + u * 16;
    active[u] = q0 < Lq;
    any_active = any_active || active[u];
    qf[u] = 0ull;
    if (active[u]) {
      const unsigned short* qp = Qr + (bh * Lqp + q0 + li) * 32;
      const s16x8 hi = *reinterpret_cast<const s16x8*>(qp + (g & 1) * 8);
      const s16x8 lo = *reinterpret_cast<const s16x8*>(qp + 16 + (g & 1) * 8);
      const float sq = ldexpf(1.0f, -ek);
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        x[i] = clamp448((half_bits_to_float((unsigned short)hi[i]) + half_bits_to_float((unsigned short)lo[i])) * sq);
      const unsigned int h0 = pk_f8x4(x[0], x[1], x[2], x[3]), h1 = pk_f8x4(x[4], x[5], x[6], x[7]);
      const unsigned int l0 = pk_f8x4(x[0] - f8_byte(h0, 0), x[1] - f8_byte(h0, 1), x[2] - f8_byte(h0, 2), x[3] - f8_byte(h0, 3));
      const unsigned int l1 = pk_f8x4(x[4] - f8_byte(h1, 0), x[5] - f8_byte(h1, 1), x[6] - f8_byte(h1, 2), x[7] - f8_byte(h1, 3));
      const unsigned int w0 = (g < 2) ? h0 : l0, w1 = (g < 2) ? h1 : l1;
      qf[u] = ((unsigned long long)w1 << 32) | w0;
    }
  }
#pragma unroll
  for (int u = 0; u < QT; ++u) A3D_PIN(qf[u]);
  const int nch = Sp / C16;
  const int cps = (nch + nsplit - 1) / nsplit;
  const int c_beg = sp * cps;
  const int c_end = min(nch, c_beg + cps);

  // one LDS-DMA per wave and chunk: lanes 0-15 -> key rows 16 wave + lane; lanes 16-31 -> value channel 4 wave + (l >> 2),
  // 16-key segment (l & 3) ^ wave (source-side swizzle: the four regions then read conflict-free); lanes 32-63 idle
  const unsigned char* Kbase = K8 + bh * Sp * 16;
  const unsigned char* Vbase = V8 + bh * 16 * Sp;
  const int l16 = lane & 15;
  auto issue = [&](int c, int slot) {
    const int cc = min(c, c_end - 1);                        // past the end: a harmless re-fetch keeps the vmcnt count fixed
    const unsigned char* src = (lane < 16)
        ? Kbase + ((size_t)cc * C16 + wave * 16 + l16) * 16
        : Vbase + (size_t)(wave * 4 + (l16 >> 2)) * Sp + (size_t)cc * C16 + (((l16 & 3) ^ wave) * 16);
    if (lane < 32) glds16(src, &ring[slot][region_base(wave)]);
  };

  // fragment offsets (bytes inside a slot).  score tile j of the chunk: key row (j >> 1) * 32 + (li >> 2) * 8 + (li & 3) + (j & 1) * 4,
  // channels (g & 1) * 8 .. + 7 (lane groups 2, 3 re-read what 0, 1 read: A = [k | k])
  int koff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (j >> 1) * 32 + (li >> 2) * 8 + (li & 3) + (j & 1) * 4;
    koff[j] = region_base(row >> 4) + (row & 15) * 16 + (g & 1) * 8;
  }
  // value fragment of 32-key half hf: channel li, keys hf * 32 + g * 8 .. + 7 -> region li >> 2, segment hf * 2 + (g >> 1)
  int voff[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
    voff[hf] = region_base(li >> 2) + 256 + ((li & 3) * 4 + ((hf * 2 + (g >> 1)) ^ (li >> 2))) * 16 + (g & 1) * 8;

  float m_run[QT];
  f32x4 cin[QT], acc0[QT], acc1[QT];
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    m_run[u] = 0.f;
    cin[u] = f32x4{P8_OFF, P8_OFF, P8_OFF, P8_OFF};
    acc0[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc1[u] = acc0[u];
  }

  if (c_beg < c_end) {
#pragma unroll
    for (int i = 0; i < F8_NB - 1; ++i) issue(c_beg + i, i);
  }
  for (int c = c_beg; c < c_end; ++c) {
    const int slot = (c - c_beg) % F8_NB;
    wait_vm<F8_NB - 2>();
    ring_barrier();
    issue(c + F8_NB - 1, (slot + F8_NB - 1) % F8_NB);
    const bool first = (c == c_beg);
    const bool masked = any_masked && ((kmask != nullptr) || ((c + 1) * C16 > S));
    if (!any_active) continue;

    unsigned long long kf[4], vf[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) kf[j] = *reinterpret_cast<const unsigned long long*>(&ring[slot][koff[j]]);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) vf[hf] = *reinterpret_cast<const unsigned long long*>(&ring[slot][voff[hf]]);

    f32x4 s[QT][4];
    if (masked) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 b4 = bias_of(maskW[c * 2 + (j >> 1)], g, j & 1);
#pragma unroll
        for (int u = 0; u < QT; ++u) s[u][j] = mfma_f8(kf[j], qf[u], cin[u] + b4);
      }
    } else {
#pragma unroll
      for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[u][j] = mfma_f8(kf[j], qf[u], cin[u]);
    }

#pragma unroll
    for (int u = 0; u < QT; ++u) {
      const float mx = max16(s[u][0], s[u][1], s[u][2], s[u][3]);
      if (first || __builtin_amdgcn_ballot_w64(mx > P8_OFF + P8_THR) != 0ull) {
        const float cm = colmax4(mx);
        float shift = first ? (cm - P8_OFF) : fmaxf(cm - P8_OFF, 0.f);
        if (cm == -INFINITY) shift = 0.f;
        m_run[u] += shift;
        const float alpha = __builtin_amdgcn_exp2f(-shift);
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc0[u][r] *= alpha; acc1[u][r] *= alpha; cin[u][r] -= shift; }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[u][j][r] -= shift;
      }
      unsigned long long pf[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const f32x4& s0 = s[u][hf * 2], &s1 = s[u][hf * 2 + 1];
        const unsigned int w0 = pk_f8x4(__builtin_amdgcn_exp2f(s0[0]), __builtin_amdgcn_exp2f(s0[1]),
                                        __builtin_amdgcn_exp2f(s0[2]), __builtin_amdgcn_exp2f(s0[3]));
        const unsigned int w1 = pk_f8x4(__builtin_amdgcn_exp2f(s1[0]), __builtin_amdgcn_exp2f(s1[1]),
                                        __builtin_amdgcn_exp2f(s1[2]), __builtin_amdgcn_exp2f(s1[3]));
        pf[hf] = ((unsigned long long)w1 << 32) | w0;
      }
      acc0[u] = mfma_f8(vf[0], pf[0], acc0[u]);
      acc1[u] = mfma_f8(vf[1], pf[1], acc1[u]);
    }
  }
  wait_vm<0>();

  const float ov = ldexpf(1.0f, -ev);                  // undo the value scale
#pragma unroll
  for (int u = 0; u < QT; ++u) {
    if (!active[u]) continue;
    f32x4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = acc0[u][r] + acc1[u][r];
    const float l_tot = __shfl(acc[3], 48 + li, 64);   // channel 15 (lane group 3, register 3) holds sum_k p
    const int q = qbase + u * 16 + li;
    const float mq = m_run[u] - P8_OFF;                // p = 2^(s2 - mq)
    if (nsplit == 1) {
      const float inv = (l_tot > 0.f) ? ov / l_tot : 0.f;
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
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] *= ov;        // the combine kernel divides by Lp, never reads channel 15
      *reinterpret_cast<f32x4*>(&Op[row * HDP + g * 4]) = acc;
      if (g == 0) { Mp[row] = (l_tot > 0.f) ? mq : -INFINITY; Lp[row] = l_tot; }
    }
  }
}

}  // namespace a3d

using namespace a3d;

extern "C" size_t a3d_attn8_operand_bytes(int B, int H, int Sp) {
  // K8 rows + V8 planes + the amax words, each 256-byte aligned
  const size_t kv = (size_t)B * H * Sp * 16;
  return 2 * ((kv + 255) / 256 * 256) + (((size_t)B * H * 16 + 255) / 256 * 256);
}

extern "C" int a3d_attn8_fwd(const void* Qr, const void* Kr, const void* Vp, void* ops8, const unsigned char* kmask, float* O,
                             float* LSE2, float* ws, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit, void* stream) {
  int rc = attn16_check_shapes("a3d_attn8_fwd", B, H, Lq, Lqp, S, Sp, nsplit, 16);
  if (rc) return rc;
  if (!Qr || !Kr || !Vp || !ops8 || !O || !LSE2 || (nsplit > 1 && !ws)) { set_error("a3d_attn8_fwd: null pointer"); return A3D_ERR_ARG; }
  if (((size_t)ops8 & 255) != 0) { set_error("a3d_attn8_fwd: ops8 must be 256-byte aligned"); return A3D_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  const size_t kv = ((size_t)B * H * Sp * 16 + 255) / 256 * 256;
  unsigned char* K8 = (unsigned char*)ops8;
  unsigned char* V8 = K8 + kv;
  unsigned int* amax = (unsigned int*)(V8 + kv);
  rc = zero_words(amax, (size_t)B * H * 16, s, "a3d_attn8_fwd(zero)");
  if (rc) return rc;
  const int gx = std::max(1, std::min(cdiv(Sp, 256), 16));
  hipLaunchKernelGGL(attn8_amax_kernel, dim3(gx, B * H), dim3(256), 0, s, (const unsigned short*)Qr, (const unsigned short*)Kr,
                     (const unsigned short*)Vp, amax, B, H, Lq, Lqp, S, Sp);
  rc = check_launch("a3d_attn8_fwd(amax)");
  if (rc) return rc;
  hipLaunchKernelGGL(attn8_pack_kernel, dim3(gx, B * H), dim3(256), 0, s, (const unsigned short*)Kr, (const unsigned short*)Vp, amax,
                     K8, V8, B, H, S, Sp);
  rc = check_launch("a3d_attn8_fwd(pack)");
  if (rc) return rc;
  const size_t rows = (size_t)B * H * Lqp;
  float* Op = ws;
  float* Mp = ws ? ws + (size_t)nsplit * rows * HDP : nullptr;
  float* Lp = ws ? Mp + (size_t)nsplit * rows : nullptr;
  static const int qt_env = getenv("A3D_ATTN_QT") ? atoi(getenv("A3D_ATTN_QT")) : 0;
  const int QT = qt_env ? qt_env : (Lq > 64 ? 2 : 1);
  dim3 grid(xcd_grid(B * H, cdiv(Lqp, 64 * QT) * nsplit));
  if (QT == 2)
    hipLaunchKernelGGL((attn8_fwd_kernel<2>), grid, dim3(256), 0, s, (const unsigned short*)Qr, K8, V8, amax, kmask, O, LSE2, Op, Mp,
                       Lp, B, H, Lq, Lqp, S, Sp, nsplit);
  else
    hipLaunchKernelGGL((attn8_fwd_kernel<1>), grid, dim3(256), 0, s, (const unsigned short*)Qr, K8, V8, amax, kmask, O, LSE2, Op, Mp,
                       Lp, B, H, Lq, Lqp, S, Sp, nsplit);
  rc = check_launch("a3d_attn8_fwd");
  if (rc) return rc;
  if (nsplit > 1) rc = attn16_launch_combine(Op, Mp, Lp, O, LSE2, B, H, Lq, Lqp, nsplit, s);
  return rc;
}
