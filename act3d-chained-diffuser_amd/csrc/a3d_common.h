// Shared device/host helpers for the act3d HIP library (gfx950 / CDNA4 only).
// Everything here is internal; the public C-ABI lives in include/act3d_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#define A3D_OK 0
#define A3D_ERR_ARG (-22)      // EINVAL
#define A3D_ERR_LAUNCH (-5)    // EIO
#define A3D_ERR_NOMEM (-12)    // ENOMEM

namespace a3d {

void set_error(const char* fmt, ...);
int check_launch(const char* what);
// Zeroes `bytes` (a multiple of 4, 4-byte aligned) with a KERNEL node.  hipMemsetAsync is not used anywhere in this library: a
// captured memset node replayed wrong on this stack (right on eager launches and on the first replay, stale on every later
// one -- profiles/r05_persist_graph_probe.txt), and every entry point here must be capturable.
int zero_words(void* p, size_t bytes, hipStream_t s, const char* what);

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int HD = 15;    // head dim of both models (60/4, 120/8)
constexpr float LOG2E_F = 1.4426950408889634f;
constexpr int HDP = 16;   // padded head dim

// round-to-nearest-even fp32 -> bf16 bits (finite inputs)
__device__ __forceinline__ unsigned short f2bf(float x) {
  unsigned int u = __float_as_uint(x);
  unsigned int r = u + 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(r >> 16);
}
// two fp32 -> packed bf16 pair (a in the low half) with the gfx950 hardware conversion v_cvt_pk_bf16_f32: round-to-nearest-even,
// the same bits as f2bf for finite inputs
__device__ __forceinline__ unsigned int pk_bf16(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
__device__ __forceinline__ float bf2f(unsigned short h) {
  return __uint_as_float(((unsigned int)h) << 16);
}
// x ~= hi + lo with hi, lo bf16 (16 mantissa bits kept in total)
__device__ __forceinline__ void split_bf16(float x, unsigned short& hi, unsigned short& lo) {
  hi = f2bf(x);
  lo = f2bf(x - bf2f(hi));
}

// x ~= hi + lo + lo2 (24 mantissa bits: fp32-exact up to the last rounding).  The q / k score operands carry all three
// parts: exp() turns an absolute score error into a relative weight error, and 2^-17 * |q||k| is ~1e-3 at |s| ~ 100.
__device__ __forceinline__ void split_bf16_3(float x, unsigned short& hi, unsigned short& lo, unsigned short& lo2) {
  hi = f2bf(x);
  const float r1 = x - bf2f(hi);
  lo = f2bf(r1);
  lo2 = f2bf(r1 - bf2f(lo));
}
// row widths (bf16 elements) of the "rows" operand format: q / k rows are [hi16 | lo16 | lo2 16], v / dO rows [hi16 | lo16]
constexpr int QKW = 48;
constexpr int VRW = 32;
// LDS tiles feeding MFMA A operands with ds_read_b128.  A wave's b128 read is serviced in four NON-contiguous
// 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, {32-35,44-47,52-59}, {36-43,48-51,60-63};
// MI355X_MICROARCH.md, LDS), i.e. with lane = g*16 + li each group holds li-blocks {0,3} of one g and {1,2} of the next.
// A tile is [rows][32] bf16 (64-byte rows, four 16-byte segments); data segment `seg` of a row whose li-block is `blk`
// sits at position seg ^ ((-blk) & 3): within every group the 16 lanes then hit 16 distinct 16-byte slots of the
// 256-byte bank row (conflict-free), where the naive seg ^ blk gives 2-way conflicts on every read.
//   rows tile  : [64 keys/queries][32]; lane (li, g) reads row base + (li >> 2) * 8 + (li & 3) -> blk = (row >> 3) & 3
//   plane tile : [16 channels][32 keys];  lane (li, g) reads row li                             -> blk = row >> 2
__device__ __forceinline__ int tile_off(int row, int seg) { return row * 32 + ((seg ^ ((0 - (row >> 3)) & 3)) << 3); }
__device__ __forceinline__ int plane_off(int row, int seg) { return row * 32 + ((seg ^ ((0 - (row >> 2)) & 3)) << 3); }

// max over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (one MFMA result column) with the gfx950 row-swap VALU ops
// (v_permlane16_swap / v_permlane32_swap: no LDS round trip, unlike ds_bpermute-based __shfl_xor)
__device__ __forceinline__ float colmax4(float v) {
  typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
  u32x2_t r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// sum over the four lanes l, l ^ 16, l ^ 32, l ^ 48 of a double (two 32-bit row swaps per step)
__device__ __forceinline__ double colsum4(double v) {
  typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
  unsigned int lo = (unsigned int)(__double_as_longlong(v) & 0xFFFFFFFFll), hi = (unsigned int)(__double_as_longlong(v) >> 32);
  u32x2_t a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  u32x2_t b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  v = __longlong_as_double(((long long)b[0] << 32) | a[0]) + __longlong_as_double(((long long)b[1] << 32) | a[1]);
  lo = (unsigned int)(__double_as_longlong(v) & 0xFFFFFFFFll);
  hi = (unsigned int)(__double_as_longlong(v) >> 32);
  a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __longlong_as_double(((long long)b[0] << 32) | a[0]) + __longlong_as_double(((long long)b[1] << 32) | a[1]);
}

__device__ __forceinline__ f32x4 mfma_bf16_16x16x32(s16x8 a, s16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_f32_16x16x4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// RoPE-3D frequency of pair index k inside one axis third (reference
// model/utils/position_encodings.py:72-75): exp(-ln(1e4) * 2k / (E/3)).
__device__ __forceinline__ float rope_freq(int k, int E) {
  return expf((float)(2 * k) * (-9.210340371976184f / (float)(E / 3)));
}

// sin and cos of a RoPE angle (xyz coordinate x frequency <= 1: a few radians).  libm's sincosf costs ~100 VALU instructions
// per call (Payne-Hanek branch, two separate polynomials behind calls) and was the largest single item of the rotating
// kernels (7.5 calls per thread and 64-key tile in proj_rope_split / sq_fwd, twice that in sq_bwd).  For |x| < 200:
// k = rint(x * 2/pi), r = x - k * pi/2 by two FMAs against a two-part pi/2 (the first product is exact inside the fma, so
// r carries one rounding: |error| < 4e-8), then the Cephes single-precision minimax polynomials on [-pi/4, pi/4] (1 ulp)
// and a quadrant swap -- 22 instructions, both results within 2 ulp of the correctly rounded values (tests/test_host_cpu.py
// checks the host mirror a3d_sincos_host against float64 over the range).  Larger arguments: see fast_sincos below.
__host__ __device__ __forceinline__ void sincos_poly(float x, float* sn, float* cs) {
  const float kf = rintf(x * 0.636619772367581343f);
  float r = fmaf(-kf, 1.57079637050628662109375f, x);
  r = fmaf(-kf, -4.37113900018624283e-8f, r);
  const float z = r * r;
  const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                        fmaf(-0.5f, z, 1.0f));
  const int q = (int)kf;
  const float s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
  *sn = (q & 2) ? -s0 : s0;
  *cs = ((q + 1) & 2) ? -c0 : c0;
}
// Large arguments (|x| >= 200: never reached by real RoPE angles -- coordinates in metres times a frequency <= 1) are reduced
// modulo 2 pi in DOUBLE precision (two-part 2 pi, |error| < 1e-13 for |x| < 1e8) and take the same polynomials: eight inline
// instructions on a path that is never hot, NO function call in any rotating kernel.  History, kept because both earlier forms
// cost a GPU lease each: round 4 called libm's sincosf out of line through pointer results (the callers' (cos, sin) arrays then
// lived in scratch memory: 80 B per lane in sq_fwd / sq_bwd); round 5 first returned a float2 from a __noinline__ static
// function instead -- with that build ONE fused denoise step was no longer run-to-run deterministic on MI355X (3e-4 .. 1e-3
// differences between identical launches of dn_cross / dn_rest, profiles/r05_cfg3_probe_ab.txt: the round-4 tree and this tree
// with round 4's helper are bit-stable, the float2-returning call is not), although the call is never taken.
__host__ __device__ __forceinline__ void fast_sincos(float x, float* sn, float* cs) {
#if defined(A3D_LIBM_SINCOS) && defined(__HIP_DEVICE_COMPILE__)        // A/B build (A3D_HIPCC_FLAGS=-DA3D_LIBM_SINCOS): libm everywhere
  sincosf(x, sn, cs);
#else
  if (fabsf(x) >= 200.0f) {
    const double t = (double)x;
    const double k = rint(t * 0.15915494309189535);
    x = (float)fma(-k, 2.4492935982947064e-16, fma(-k, 6.283185307179586, t));      // |x| <= pi now (inf / nan stay nan)
  }
  sincos_poly(x, sn, cs);
#endif
}

// IEEE round-to-nearest single operations that the compiler may NOT contract into an fma.  (HIP's __fadd_rn /
// __fmul_rn are plain operators and __fsqrt_rn is the approximate native sqrt unless OCML_BASIC_ROUNDED_OPERATIONS
// is defined, so they cannot be used where bit-exactness with the CPU oracle is required.)
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float sqrt_rn(float a) { return __builtin_sqrtf(a); }   // correctly rounded (HIP default)

// ---------------------------------------------------------------- Philox4x32-10 (Salmon et al. 2011)
__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                       uint32_t k0, uint32_t k1, uint32_t out[4]) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ---------------------------------------------------------------- dropout masks (training-mode nn.Dropout / F.dropout)
// One Philox call yields the keep flags of a block of 8 elements: element j of the block is KEPT iff the j-th 16-bit
// field of the 128 output bits is >= thr16 = round(p * 65536); kept elements are scaled by 1 / (1 - p).
//   counter = (c0, c1, c2, site), key = the two halves of  seed + offset * 0x9E3779B97F4A7C15  (mod 2^64): the four counter
//   words are taken, so the generator offset has to live in the key; the odd multiplier keeps (seed, offset) pairs that a
//   plain XOR would alias -- (1, 0) and (0, 1) -- on different streams
//   attention weights (multihead_custom_attention.py:413): c0 = key >> 3, c1 = query, c2 = b * H + h
//   elementwise sites  (layers.py:82-84,146,181; diffusion_head.py:46,183,193): c0, c1 = lo / hi of (index >> 3),
//                      c2 = 0xFFFFFFFF
// `site` identifies the nn.Dropout module instance; {seed, offset} is a device-resident uint64[2] snapshot taken at the
// start of the forward pass, so forward and backward of one step regenerate the same masks and the step stays capturable.
struct DropKey { uint32_t k0, k1; };
__device__ __forceinline__ DropKey drop_key(const unsigned long long* state) {
  const unsigned long long seed = state[0], offs = state[1];
  const unsigned long long key = seed + offs * 0x9E3779B97F4A7C15ull;
  DropKey k;
  k.k0 = (uint32_t)key;
  k.k1 = (uint32_t)(key >> 32);
  return k;
}
// bit j of the result = keep flag of element j of the 8-element block
__device__ __forceinline__ uint32_t drop_keep8(DropKey k, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t site,
                                               uint32_t thr16) {
  uint32_t r[4];
  philox4x32_10(c0, c1, c2, site, k.k0, k.k1, r);
  uint32_t bits = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    bits |= ((r[w] & 0xFFFFu) >= thr16 ? 1u : 0u) << (2 * w);
    bits |= ((r[w] >> 16) >= thr16 ? 1u : 0u) << (2 * w + 1);
  }
  return bits;
}

// XCD-aware workgroup decoding (MI355X: 8 XCDs with private 4 MiB L2s; workgroup i is dispatched to XCD i % 8).
// All `per_group` workgroups that share one (sample, head)'s K/V or Q/dO tensors are placed on ONE XCD, so that
// those tensors are fetched from HBM once and re-read from that XCD's L2.  Grid = xcd_grid(ngroups, per_group).
// Placement only affects speed, never results.
__device__ __forceinline__ bool xcd_decode(int per_group, int ngroups, int& group, int& within) {
  const int id = blockIdx.x;
  const int xcd = id & 7, slot = id >> 3;
  group = (slot / per_group) * 8 + xcd;
  within = slot % per_group;
  return group < ngroups;
}
static inline int xcd_grid(int ngroups, int per_group) { return ((ngroups + 7) / 8) * 8 * per_group; }

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// deep-layer 1x1 convolution GEMM (conv1x1_deep.hip), dispatched to by a3d_conv1x1_bn_fwd (conv1x1.hip)
bool conv1x1_deep_serves(int K, int N);
int conv1x1_deep_slabs(size_t M, int K, int N);
int conv1x1_deep_launch(const void* x, const void* w, const float* in_scale, const float* in_shift, int in_relu, void* y, float* partial,
                        size_t M, int K, int N, hipStream_t s);

}  // namespace a3d
