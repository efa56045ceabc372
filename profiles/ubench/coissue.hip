// How much vector work hides behind a v_mfma_f32_16x16x32_f16 on one SIMD of gfx950?  Streams of { 1 MFMA + NV independent VALU ops of one
// kind } at 1 / 2 / 3 waves per SIMD, timed by wall clock; the MFMA-only and VALU-only streams of the same length give the two
// single-pipe times, so every row shows  t(mix)  against  max(t_mfma, t_valu)  (perfect overlap) and  t_mfma + t_valu  (none).
// The attention kernels' instruction budget is priced with this table.
// Build: hipcc --offload-arch=gfx950 -O3 coissue.hip -o coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define ITERS 2000
#define GROUPS 8          // groups per loop iteration (8 independent accumulators)

enum Kind { K_FMA = 0, K_EXP, K_CVT, K_MIX, K_MAX3, K_PKMUL, K_COUNT };
static const char* kind_names[K_COUNT] = {"v_fma_f32", "v_exp_f32", "v_cvt_pk_f16_f32", "v_fma_mixlo_f16", "v_max3_f32", "v_pk_mul_f32"};

template <int KIND>
__device__ __forceinline__ void valu_op(float (&x)[16], int i) {
  float& a = x[i & 15];
  const float b = x[(i + 5) & 15];
  if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(b));
  else if (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
  else if (KIND == K_CVT) { unsigned int r; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); a = __uint_as_float(r | 0x3f000000u); }
  else if (KIND == K_MIX) { unsigned int r = __float_as_uint(a); asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(r) : "v"(b)); a = __uint_as_float(r); }
  else if (KIND == K_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  else if (KIND == K_PKMUL) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 v = {a, b};
    asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(v));
    a = v.x;
  }
}

// MF: MFMAs per group (0 or 1), NV: VALU ops per group
template <int KIND, int MF, int NV>
__global__ __launch_bounds__(256) void stream(float* out, int iters) {
  const int t = threadIdx.x;
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = 0.5f + 0.001f * (float)(t + i);
  f32x4 acc[GROUPS];
#pragma unroll
  for (int i = 0; i < GROUPS; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 ha, hb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.01f * (t & 7) + 0.1f * i); hb[i] = (_Float16)(0.02f * (t & 3) - 0.05f * i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int gi = 0; gi < GROUPS; ++gi) {
      if (MF) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[gi]) : "v"(ha), "v"(hb));
#pragma unroll
      for (int k = 0; k < NV; ++k) valu_op<KIND>(x, gi * NV + k);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < GROUPS; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + t] = s;
}

template <int KIND, int MF, int NV>
static double run(int waves_per_simd, float* out) {
  const int blocks = 256 * waves_per_simd;          // 4 waves per workgroup = 1 per SIMD; waves_per_simd workgroups per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  stream<KIND, MF, NV><<<blocks, 256>>>(out, ITERS);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  stream<KIND, MF, NV><<<blocks, 256>>>(out, ITERS);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return (double)ms * 1e6 / ((double)ITERS * GROUPS * waves_per_simd);      // ns per group and SIMD
}

template <int KIND, int NV>
static void row(int w, float* out, double t_mfma) {
  const double t_mix = run<KIND, 1, NV>(w, out), t_valu = run<KIND, 0, NV>(w, out);
  const double lo = t_mfma > t_valu ? t_mfma : t_valu, hi = t_mfma + t_valu;
  printf("  waves/SIMD %d  1 MFMA + %d %-18s: mix %6.2f ns | valu alone %6.2f | max %6.2f sum %6.2f | hidden %5.1f %% of the smaller | in MFMA units %.2f\n",
         w, NV, kind_names[KIND], t_mix, t_valu, lo, hi, 100.0 * (hi - t_mix) / (hi - lo > 1e-9 ? (t_mfma < t_valu ? t_mfma : t_valu) : 1.0), t_mix / t_mfma);
}

template <int KIND>
static void kind_rows(int w, float* out, double t_mfma) {
  row<KIND, 1>(w, out, t_mfma);
  row<KIND, 2>(w, out, t_mfma);
  row<KIND, 3>(w, out, t_mfma);
  row<KIND, 4>(w, out, t_mfma);
  row<KIND, 6>(w, out, t_mfma);
  row<KIND, 8>(w, out, t_mfma);
}

int main() {
  float* out;
  hipMalloc(&out, 1024 * 1024 * sizeof(float));
  for (int w = 1; w <= 3; ++w) {
    const double t_mfma = run<K_FMA, 1, 0>(w, out);
    printf("waves/SIMD %d: MFMA 16x16x32 f16 alone %.2f ns per instruction and SIMD (16 cycles => %.2f GHz)\n", w, t_mfma, 16.0 / t_mfma);
    kind_rows<K_FMA>(w, out, t_mfma);
    kind_rows<K_EXP>(w, out, t_mfma);
    kind_rows<K_CVT>(w, out, t_mfma);
    kind_rows<K_MIX>(w, out, t_mfma);
    kind_rows<K_MAX3>(w, out, t_mfma);
    kind_rows<K_PKMUL>(w, out, t_mfma);
  }
  return 0;
}
