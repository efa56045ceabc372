// Pieces shared by the split-fp16 attention kernels (attention16.hip) and the fp8 forward (attention8.hip): vector types,
// 16-bit packing helpers, the LDS-DMA ring primitives, the key-validity bitmask.  See attention16.hip for the design notes.
#pragma once
#include "a3d_common.h"

namespace a3d {

typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 h16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2_;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

constexpr int C16 = 64;              // keys (fwd, dQ) or queries (dK/dV) per staged chunk
constexpr float P_OFF = 4.0f;        // p = 2^(s - m + P_OFF): keeps the small weights of a row out of fp16's subnormals
constexpr float P_THR = 8.0f;        // lazy rescale: revise the running max when a score exceeds it by 2^P_THR
constexpr float LN2_F = 0.6931471805599453f;
// backward: P and G = P (dP - D) are formed as 2^B_OFF times their value (folded into the -lse accumulator init, undone in
// the output scale).  Attention over 4097 keys has weights ~2^-12 and G two or three orders below; without the offset they
// sit in fp16's subnormals (absolute precision 2^-25) -- measured as a 1.6 % error of the gripper-token key's gradient.
constexpr float B_OFF = 6.0f;
constexpr int MASKW = 512;           // key-validity bitmask words in LDS: Sp <= 16384

__device__ __forceinline__ f32x4 mfma_f16(s16x8 a, s16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}
// two floats -> packed fp16 (round to nearest even; v_cvt_pk_f16_f32)
__device__ __forceinline__ unsigned int pk_f16(float a, float b) {
  return __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2_){a, b}, h16x2));
}
// lo = fp16(a - hi.x) | fp16(b - hi.y) for hi = pk_f16(a, b): the low parts of the two-part fp16 operands.  Written out in
// C++ this is 2 x v_cvt_f32_f16 + 2 x v_sub_f32 + v_cvt_pk_f16_f32 -- 80 of ~140 VALU instructions of the forward's two-tile
// loop body existed only to form P_lo (round-3 review).  v_fma_mixlo/hi_f16 take the fp16 half straight as an fma operand
// and round the fp32 result into a half: a * 1.0 - hi is exact in fp32 (hi is a's own rounding), so the one rounding to fp16
// gives the same bits as the five-instruction sequence.  hipcc emits no mix instruction on its own; A3D_NO_FMA_MIX builds
// the C++ form for the A/B run.
__device__ __forceinline__ unsigned int lo_f16(float a, float b, unsigned int hi) {
#ifdef A3D_NO_FMA_MIX
  const h16x2 hh = __builtin_bit_cast(h16x2, hi);
  return pk_f16(a - (float)hh[0], b - (float)hh[1]);
#else
  unsigned int lo;
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(lo)
      : "v"(a), "v"(b), "v"(hi));
  return lo;
#endif
}
// x = hi + lo, both fp16 pairs: hi = fp16(x) (round to nearest), lo = fp16(x - hi)
__device__ __forceinline__ void pk_f16_2(float a, float b, unsigned int& hi, unsigned int& lo) {
  hi = pk_f16(a, b);
  lo = lo_f16(a, b, hi);
}
// x = hi + lo, both bf16 pairs (16 mantissa bits, fp32's exponent range): the operands of the query-axis contractions
__device__ __forceinline__ void pk_bf16_2(float a, float b, unsigned int& hi, unsigned int& lo) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
  hi = __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2_){a, b}, bf16x2_));
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xFFFF0000u);
  lo = __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2_){ra, rb}, bf16x2_));
}
// The same split by TRUNCATION (round 6, the dK / dV kernel): hi = the top 16 bits of a and b (ONE v_perm_b32), lo = the top 16 bits of
// x - hi (two v_and, two v_sub, one v_perm_b32) instead of two v_cvt_pk_bf16_f32 + shift + mask + two v_sub.  Priced by
// profiles/ubench/coissue.hip: a v_cvt_pk costs three plain VALU operations next to MFMAs (3.3 ns vs 1.1 ns per wave instruction and
// SIMD), so a pair costs 6.6 ns instead of 11.0 -- the two splits (P' and G') were 88 of the ~124 ns of vector work per (16 keys x 32
// rows).  x - hi is in [0, 2^-7 |x|) and keeps 8 bits: hi + lo carries 15 - 16 significant bits, truncated toward zero (the rounded
// split carried 17).  Measured (profiles/r06_attn_ab.txt, float64 reference): dK / dV errors 7.4e-6 / 8.6e-6 -> 1.6e-5 / 2.0e-5 of
// scale at mild logits, 2.3e-5 / 1.2e-5 -> 2.8e-5 / 1.6e-5 at sharp ones, backward launch 0.372 -> 0.357 ms.  A3D_BF16_SPLIT_RNE
// builds the rounded split; A3D_BF16_SPLIT_TRUNC1 truncates hi only (lo rounded: 8.8 ns per pair, 1.0e-5 / 1.6e-5, 0.367 ms).
__device__ __forceinline__ void pk_bf16_2t(float a, float b, unsigned int& hi, unsigned int& lo) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
  const unsigned int ua = __float_as_uint(a), ub = __float_as_uint(b);
  hi = __builtin_amdgcn_perm(ub, ua, 0x07060302u);                       // (b & 0xFFFF0000) | (a >> 16)
  const float ra = a - __uint_as_float(ua & 0xFFFF0000u), rb = b - __uint_as_float(ub & 0xFFFF0000u);
#ifdef A3D_BF16_SPLIT_TRUNC1
  lo = __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2_){ra, rb}, bf16x2_));
#else
  lo = __builtin_amdgcn_perm(__float_as_uint(rb), __float_as_uint(ra), 0x07060302u);
#endif
}
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float max16(const f32x4& a, const f32x4& b, const f32x4& c, const f32x4& d) {
  const float m0 = max3f(a[0], a[1], a[2]), m1 = max3f(a[3], b[0], b[1]), m2 = max3f(b[2], b[3], c[0]);
  const float m3 = max3f(c[1], c[2], c[3]), m4 = max3f(d[0], d[1], d[2]);
  return fmaxf(max3f(m0, m1, m2), max3f(m3, m4, d[3]));
}

// ---- LDS-DMA pieces: one wave instruction moves 64 lanes x 16 B to `lds` (wave-uniform) + lane * 16.
// Issued through inline asm on purpose: for the builtin form hipcc's waitcnt pass orders EVERY later ds_read behind the
// newest pending LDS-DMA (s_waitcnt vmcnt(0) in front of the first fragment read), which serialises the ring; the asm form
// is invisible to it, and the kernels below place the counted vmcnt waits themselves (their loops issue no other VMEM
// loads, and an uncounted op only ever makes a compiler-placed vmcnt(k) wait longer, never shorter -- returns are in order).
__device__ __forceinline__ void glds16(const void* g, void* lds) {
  const unsigned int dst = __builtin_amdgcn_readfirstlane((unsigned int)(size_t)(lds_void_t*)lds);
  // m0 is declared clobbered instead of saved and restored around the load (round 6: two SALU instructions less per piece; nothing else
  // in these kernels reads m0 -- gfx9 LDS instructions take no bound from it)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(dst) : "memory", "m0");
}
// rows tile [64 rows][32 halfs] (4 KB, tile_off swizzle): wave w brings rows w*16 .. w*16+15.  `row_halfs` = source row
// length in halfs (32: hi | lo rows; 16: single rows duplicated into both halves of the tile row)
__device__ __forceinline__ void dma_rows_tile(const unsigned short* src_row0, int row_halfs, unsigned short* tile, int wave,
                                              int lane) {
  const int row = wave * 16 + (lane >> 2);
  const int seg = (lane & 3) ^ ((0 - (row >> 3)) & 3);
  const int sseg = (row_halfs == 32) ? seg : (seg & 1);
  glds16(src_row0 + (size_t)row * row_halfs + sseg * 8, tile + wave * 512);
}
// plane sub-tile [16 ch][32 rows] (1 KB, plane_off swizzle) from a [16][ld] plane at row offset r0
__device__ __forceinline__ void dma_plane_subtile(const unsigned short* plane, size_t ld, size_t r0, unsigned short* sub, int lane) {
  const int ch = lane >> 2;
  const int seg = (lane & 3) ^ ((0 - (ch >> 2)) & 3);
  glds16(plane + (size_t)ch * ld + r0 + seg * 8, sub);
}

// ---- transposed fragments from a ROWS tile (round 6).  ds_read_b64_tr_b16: within each 16-lane group, lane i passes the address of 4
// consecutive halfs and receives R[i][j] = D[lane 4 j + (i >> 2)][i & 3] (profiles/ubench/tr_read.hip confirms it on gfx950): with
// lane i pointing at (row r0 + (i >> 2), columns 4 (i & 3) .. + 3) of a row-major [4 rows][16 columns] block, lane i receives COLUMN i
// of the four rows.  Two reads (rows + 0..3, + 4..7) give lane (li, g) the MFMA A fragment [channel li][keys g*8 .. g*8+7] of a
// [64 keys][hi16 | lo16] rows tile -- what used to need a second, transposed copy of K (dQ) and V (forward) in HBM and in LDS.
// Bank behaviour with the tile_off swizzle: a 16-lane group touches 4 rows x 32 B at a 64-byte row stride (banks 16 r + 0..7 or
// + 8..15), the two groups of a 32-lane service half differ in (row >> 3) & 3, i.e. in the swizzle -> disjoint banks.
typedef __attribute__((ext_vector_type(4))) short s16x4_;
__device__ __forceinline__ s16x4_ lds_tr16(const unsigned short* p) {
  typedef __attribute__((address_space(3))) s16x4_ lds_s16x4;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
}
// offset (halfs) lane (li, g) passes for: part (0: hi, 1: lo), 32-key half hf, row block rr (keys g*8 + rr*4 ..+3)
__device__ __forceinline__ int tr_off(int li, int g, int part, int hf, int rr) {
  const int row = hf * 32 + g * 8 + rr * 4 + (li >> 2);
  const int c = part * 16 + (li & 3) * 4;
  return tile_off(row, c >> 3) + (c & 7);
}
__device__ __forceinline__ s16x8 tr_frag(const unsigned short* tile, int off0, int off1) {
  const s16x4_ a = lds_tr16(tile + off0), b = lds_tr16(tile + off1);
  return s16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// Pins a register-resident operand loaded before the main loop: the (empty) asm is a use, so hipcc retires the load HERE and
// not at its first use inside the loop, where its vmcnt wait would also drain the LDS-DMA ring.
#define A3D_PIN(x) asm volatile("" ::"v"(x))
#define A3D_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 12, "extend wait_vm");
  if (N == 0) A3D_WAIT_VM(0); else if (N == 1) A3D_WAIT_VM(1); else if (N == 2) A3D_WAIT_VM(2); else if (N == 3) A3D_WAIT_VM(3);
  else if (N == 4) A3D_WAIT_VM(4); else if (N == 5) A3D_WAIT_VM(5); else if (N == 6) A3D_WAIT_VM(6); else if (N == 7) A3D_WAIT_VM(7);
  else if (N == 8) A3D_WAIT_VM(8); else if (N == 9) A3D_WAIT_VM(9); else if (N == 10) A3D_WAIT_VM(10);
  else if (N == 11) A3D_WAIT_VM(11); else A3D_WAIT_VM(12);
}
// raw barrier (a __syncthreads() would drain the LDS-DMA queue with vmcnt(0)); this wave's LDS writes / reads are retired first
__device__ __forceinline__ void ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// key-validity bitmask of sample b (bit k of word k / 32 set = key valid), built once per workgroup
__device__ __forceinline__ void build_key_mask(unsigned int* maskW, const unsigned char* __restrict__ kmask, int b, int S, int Sp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int w0 = wave; w0 < Sp / 64; w0 += 4) {
    const int key = w0 * 64 + lane;
    bool valid = key < S;
    if (valid && kmask) valid = kmask[(size_t)b * S + key] == 0;
    const unsigned long long bits = __builtin_amdgcn_ballot_w64(valid);
    if (lane == 0) { maskW[w0 * 2] = (unsigned int)bits; maskW[w0 * 2 + 1] = (unsigned int)(bits >> 32); }
  }
}
// 0 / -inf biases of the lane's four keys of score tile T of the 32-key half whose validity word is `word`
__device__ __forceinline__ f32x4 bias_of(unsigned int word, int g, int T) {
  const unsigned int bits = word >> (g * 8 + T * 4);
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = ((bits >> i) & 1u) ? 0.f : -INFINITY;
  return r;
}

// launches attn16_combine_kernel (attention16.hip): flash-decoding combine of `nsplit` partial results
int attn16_launch_combine(const float* Op, const float* Mp, const float* Lp, float* O, float* LSE2, int B, int H, int Lq,
                          int Lqp, int nsplit, hipStream_t s);
int attn16_check_shapes(const char* fn, int B, int H, int Lq, int Lqp, int S, int Sp, int nsplit, int qmod);

}  // namespace a3d
