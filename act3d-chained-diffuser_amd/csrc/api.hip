// Library-level entry points: version, error reporting, and MFMA layout probes used by the GPU tests to pin
// the operand/result lane mappings every kernel in this library assumes.
#include "a3d_common.h"
#include "../../include/act3d_hip.h"
#include <stdarg.h>
#include <stdio.h>

namespace a3d {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return A3D_ERR_LAUNCH;
  }
  return A3D_OK;
}

__global__ __launch_bounds__(256) void zero_words_kernel(unsigned int* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
int zero_words(void* p, size_t bytes, hipStream_t s, const char* what) {
  if (!bytes) return A3D_OK;
  if ((bytes & 3) || (((size_t)p) & 3)) { set_error("%s: zero_words needs 4-byte granularity", what); return A3D_ERR_ARG; }
  const size_t n = bytes / 4;
  const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(zero_words_kernel, dim3(grid), dim3(256), 0, s, (unsigned int*)p, n);
  return check_launch(what);
}

// D[16][16] = A[16][32] * B[32][16] with the bf16 16x16x32 MFMA; A,B given as bf16 bit patterns, row-major.
__global__ void dbg_mfma_bf16_kernel(const unsigned short* A, const unsigned short* Bm, float* D) {
  const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
  s16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (short)A[li * 32 + g * 8 + e];
    b[e] = (short)Bm[(g * 8 + e) * 16 + li];
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = mfma_bf16_16x16x32(a, b, acc);
  for (int r = 0; r < 4; ++r) D[(g * 4 + r) * 16 + li] = acc[r];
}
// D[16][16] = A[16][4] * B[4][16] with the f32 16x16x4 MFMA
__global__ void dbg_mfma_f32_kernel(const float* A, const float* Bm, float* D) {
  const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = mfma_f32_16x16x4(A[li * 4 + g], Bm[g * 16 + li], acc);
  for (int r = 0; r < 4; ++r) D[(g * 4 + r) * 16 + li] = acc[r];
}

// out[i] = packed {bf16(in[2i]) low, bf16(in[2i+1]) high} via v_cvt_pk_bf16_f32 (rounding-mode probe)
__global__ void dbg_cvt_pk_kernel(const float* in, unsigned int* out, int npairs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npairs) return;
  unsigned int r;
  const float a = in[2 * i], b = in[2 * i + 1];
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  out[i] = r;
}

}  // namespace a3d

using namespace a3d;

extern "C" int a3d_version(void) { return 100; }   // 0.1.0
extern "C" const char* a3d_last_error_string(void) { return g_err; }

extern "C" int a3d_dbg_mfma_bf16(const void* A, const void* B, float* D, void* stream) {
  hipLaunchKernelGGL(dbg_mfma_bf16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned short*)A,
                     (const unsigned short*)B, D);
  return check_launch("a3d_dbg_mfma_bf16");
}
extern "C" int a3d_dbg_mfma_f32(const float* A, const float* B, float* D, void* stream) {
  hipLaunchKernelGGL(dbg_mfma_f32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, D);
  return check_launch("a3d_dbg_mfma_f32");
}
extern "C" int a3d_dbg_cvt_pk_bf16(const float* in, void* out, int npairs, void* stream) {
  hipLaunchKernelGGL(dbg_cvt_pk_kernel, dim3(cdiv(npairs, 256)), dim3(256), 0, (hipStream_t)stream, in, (unsigned int*)out, npairs);
  return check_launch("a3d_dbg_cvt_pk_bf16");
}
