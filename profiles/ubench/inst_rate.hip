// Instruction issue-rate probe for gfx950 (MI355X): cycles per wave64 instruction on ONE SIMD, measured with s_memtime
// around unrolled, register-independent streams.  Decides the VALU / MFMA budget of the attention inner loops
// ("attention: instruction budget").  Build: hipcc --offload-arch=gfx950 -O3 inst_rate.hip -o inst_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) int i32x8;

#define ITERS 2500
#define REP 8

enum Mode {
  M_FMA = 0, M_PKFMA, M_EXP, M_CVT_BF16, M_CVT_F16, M_MAX3, M_PERM32, M_MFMA_BF16, M_MFMA_F16, M_MFMA_F16_K16,
  M_MFMA_FP8, M_MFMA_32_F16, M_MIX_F16_EXP, M_MIX_FULL, M_LDEXP, M_MFMA_SCALE_FP8, M_MUL, M_PKMUL, M_CVT_RTZ, M_COUNT
};
static const char* mode_names[M_COUNT] = {
    "v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_cvt_pk_bf16_f32", "v_cvt_pk_f16_f32", "v_max3_f32",
    "v_permlane32_swap", "mfma_16x16x32_bf16", "mfma_16x16x32_f16", "mfma_16x16x16_f16", "mfma_16x16x32_fp8",
    "mfma_32x32x16_f16", "mix: 1 mfma_f16 + 2 exp", "mix: 10 mfma + 16 exp + 38 valu", "v_ldexp_f32",
    "mfma_scale_16x16x128_fp8", "v_mul_f32", "v_pk_mul_f32", "v_cvt_pkrtz_f16_f32"};

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, int iters) {
  const int t = threadIdx.x;
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = 0.5f + 0.001f * (float)(t + i);
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x16 acc32[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc32[0][i] = 0.f; acc32[1][i] = 0.f; }
  f16x8 ha, hb;
  bf16x8 ba, bb;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    ha[i] = (_Float16)(0.01f * (t & 7) + 0.1f * i);
    hb[i] = (_Float16)(0.02f * (t & 3) - 0.05f * i);
    ba[i] = (__bf16)(0.01f * (t & 7) + 0.1f * i);
    bb[i] = (__bf16)(0.02f * (t & 3) - 0.05f * i);
  }
  const long fa = 0x3838383838383838L, fb = 0x3030303030303030L;     // e4m3 1.0 / 0.25 patterns
  f16x4 h4a = {ha[0], ha[1], ha[2], ha[3]}, h4b = {hb[0], hb[1], hb[2], hb[3]};
  i32x8 sa, sb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { sa[i] = 0x38383838; sb[i] = 0x30303030; }
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
   for (int rep = 0; rep < REP; ++rep) {
    if (MODE == M_FMA) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(x[(i + 1) & 7]));
    } else if (MODE == M_MUL) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(1.0001f));
    } else if (MODE == M_PKFMA) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x2 v = {x[2 * i], x[2 * i + 1]};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v) : "v"(v));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v) : "v"(v));
        x[2 * i] = v.x; x[2 * i + 1] = v.y;
      }
    } else if (MODE == M_PKMUL) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x2 v = {x[2 * i], x[2 * i + 1]};
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(v));
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v) : "v"(v));
        x[2 * i] = v.x; x[2 * i + 1] = v.y;
      }
    } else if (MODE == M_EXP) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
    } else if (MODE == M_LDEXP) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[i]) : "v"(1));
    } else if (MODE == M_CVT_BF16) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        unsigned int r;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(x[i]), "v"(x[(i + 1) & 7]));
        x[i] = __uint_as_float(r | 0x3f000000u);
      }
    } else if (MODE == M_CVT_F16) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        unsigned int r;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(x[i]), "v"(x[(i + 1) & 7]));
        x[i] = __uint_as_float(r | 0x3f000000u);
      }
    } else if (MODE == M_CVT_RTZ) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        unsigned int r;
        asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(r) : "v"(x[i]), "v"(x[(i + 1) & 7]));
        x[i] = __uint_as_float(r | 0x3f000000u);
      }
    } else if (MODE == M_MAX3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(x[(i + 1) & 7]), "v"(x[(i + 2) & 7]));
    } else if (MODE == M_PERM32) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[2 * i]), "+v"(x[2 * i + 1]));
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[2 * i]), "+v"(x[2 * i + 1]));
    } else if (MODE == M_MFMA_BF16) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(ba), "v"(bb));
    } else if (MODE == M_MFMA_F16) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(ha), "v"(hb));
    } else if (MODE == M_MFMA_F16_K16) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(h4a), "v"(h4b));
    } else if (MODE == M_MFMA_FP8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_fp8_fp8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa), "v"(fb));
    } else if (MODE == M_MFMA_SCALE_FP8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(sa, sb, acc[i], 0, 0, 0, 127, 0, 127);
    } else if (MODE == M_MFMA_32_F16) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc32[i & 1]) : "v"(ha), "v"(hb));
    } else if (MODE == M_MIX_F16_EXP) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(ha), "v"(hb));
        asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(x[(i + 4) & 7]));
      }
    } else if (MODE == M_MIX_FULL) {
      // the instruction mix of one (64 keys x 16 queries) chunk-tile of the fp16 forward: 10 MFMAs, 16 exp, 38 other VALU
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i & 7]) : "v"(ha), "v"(hb));
        if (i < 8) {
          asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
          asm volatile("v_exp_f32 %0, %0" : "+v"(x[(i + 3) & 7]));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (i * 4 + k < 38) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(i + k) & 7]) : "v"(x[(i + k + 1) & 7]));
      }
    }
   }
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i] + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  s += acc32[0][0] + acc32[1][5];
  out[blockIdx.x * blockDim.x + t] = s;
  if ((t & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (t >> 6)] = t1 - t0;
}

// f16 MFMA: are subnormal inputs honoured?  A row 0 = 2^-20 (fp16 subnormal), B col 0 = 1.0 over all 32 k
__global__ void denorm_probe(float* out) {
  const int lane = threadIdx.x;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = __builtin_bit_cast(_Float16, (unsigned short)0x0010);    // 16 * 2^-24 = 2^-20
    b[i] = (_Float16)1.0f;
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (lane == 0) out[0] = c[0];
  // cvt of a value that lands in the fp16 subnormal range
  float v = 3.0e-6f;
  _Float16 h = (_Float16)v;
  if (lane == 0) { out[1] = (float)h; out[2] = (float)(_Float16)(1.0f + 0.00048828125f * 1.5f); }
}

template <int MODE>
static void run(int waves_per_simd, float* out, unsigned long long* cyc, double clk_ghz_hint) {
  const int threads = 256;                      // 4 waves per workgroup = 1 per SIMD; waves_per_simd workgroups per CU
  const int blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<MODE><<<blocks, threads>>>(out, cyc, ITERS);     // warm-up at full length (clock ramp)
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
  hipEventRecord(e0);
  probe<MODE><<<blocks, threads>>>(out, cyc, ITERS);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const int nw = blocks * threads / 64;
  unsigned long long* h = (unsigned long long*)malloc(nw * sizeof(unsigned long long));
  hipMemcpy(h, cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < nw; ++i) mean += (double)h[i];
  mean /= nw;
  free(h);
  int per_iter = 8;
  if (MODE == M_MIX_F16_EXP) per_iter = 8;        // per "group" of 1 mfma + 2 exp
  if (MODE == M_MIX_FULL) per_iter = 1;           // per chunk-tile
  // s_memtime / readcyclecounter ticks at a constant 100 MHz on gfx9 (REFCLK) -- report both the tick-derived figure and the
  // wall-clock figure at the hinted shader clock
  const double wall_cyc = (double)ms * 1e-3 * clk_ghz_hint * 1e9 / ((double)ITERS * REP * per_iter * waves_per_simd);
  printf("%-34s waves/SIMD %d: %8.2f ticks/inst/wave  | wall %.4f ms -> %7.2f cycles/inst/SIMD @ %.2f GHz\n", mode_names[MODE],
         waves_per_simd, mean / ((double)ITERS * REP * per_iter), ms, wall_cyc, clk_ghz_hint);
}

int main(int argc, char** argv) {
  const double clk = argc > 1 ? atof(argv[1]) : 2.4;
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 1024 * 1024 * sizeof(float));
  hipMalloc(&cyc, 65536 * sizeof(unsigned long long));
  float* dout;
  hipMalloc(&dout, 16 * sizeof(float));
  denorm_probe<<<1, 64>>>(dout);
  float h[4];
  hipMemcpy(h, dout, 3 * sizeof(float), hipMemcpyDeviceToHost);
  printf("f16 MFMA subnormal input: 32 * 2^-20 * 1.0 = %.6e (expected %.6e; 0 => flushed)\n", h[0], 32.0 * ldexp(1.0, -20));
  printf("cvt f32->f16 of 3.0e-6 = %.6e (subnormal kept if non-zero); RNE check 1+1.5ulp -> %.8f\n", h[1], h[2]);
  for (int w = 1; w <= 2; ++w) {
    run<M_FMA>(w, out, cyc, clk);
    run<M_MUL>(w, out, cyc, clk);
    run<M_PKFMA>(w, out, cyc, clk);
    run<M_PKMUL>(w, out, cyc, clk);
    run<M_EXP>(w, out, cyc, clk);
    run<M_LDEXP>(w, out, cyc, clk);
    run<M_CVT_BF16>(w, out, cyc, clk);
    run<M_CVT_F16>(w, out, cyc, clk);
    run<M_CVT_RTZ>(w, out, cyc, clk);
    run<M_MAX3>(w, out, cyc, clk);
    run<M_PERM32>(w, out, cyc, clk);
    run<M_MFMA_BF16>(w, out, cyc, clk);
    run<M_MFMA_F16>(w, out, cyc, clk);
    run<M_MFMA_F16_K16>(w, out, cyc, clk);
    run<M_MFMA_FP8>(w, out, cyc, clk);
    run<M_MFMA_SCALE_FP8>(w, out, cyc, clk);
    run<M_MFMA_32_F16>(w, out, cyc, clk);
    run<M_MIX_F16_EXP>(w, out, cyc, clk);
    run<M_MIX_FULL>(w, out, cyc, clk);
  }
  return 0;
}
