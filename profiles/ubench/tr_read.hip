// Semantics + bank-conflict probe of ds_read_b64_tr_b16 (gfx950) for the attention kernels' "rows only" plan: V^T / K^T MFMA A
// fragments (16 channels x 32 keys, lane (li, g) = channel li, keys g*8 .. g*8+7) read straight from a ROWS tile
// ([64 keys][hi16 | lo16] fp16, 64-byte rows, the tile_off swizzle of a3d_common.h) instead of from a second, transposed copy.
// Build: hipcc --offload-arch=gfx950 -O3 tr_read.hip -o tr_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) _Float16 h16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ s16x4 tr_read(const unsigned short* p) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
}

// (1) raw semantics: LDS holds its own halfword index; lane l passes address 8*l bytes (+ base); out[l][j]
__global__ void probe_linear(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const s16x4 r = tr_read(lds + threadIdx.x * 4);
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)r[j];
}

__device__ __forceinline__ int tile_off(int row, int seg) { return row * 32 + ((seg ^ ((0 - (row >> 3)) & 3)) << 3); }

// (2) the rows-tile gather: element (key, half-column c) of the tile holds key * 32 + c.  Lane (li, g) wants, for part p (0: hi, 1: lo)
// and 32-key half hf, the 8 values [key = hf*32 + g*8 + e][c = p*16 + li], e = 0..7, from two tr reads.
__global__ void probe_rows(unsigned short* out, unsigned long long* cyc, int reps) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[64 * 32];
  for (int i = threadIdx.x; i < 64 * 32; i += 64) {
    const int row = i >> 5, c = i & 31;
    tile[tile_off(row, c >> 3) + (c & 7)] = (unsigned short)(row * 32 + c);
  }
  __syncthreads();
  const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
  int off[2][2][2];
  for (int p = 0; p < 2; ++p)
    for (int hf = 0; hf < 2; ++hf)
      for (int rr = 0; rr < 2; ++rr) {
        const int row = hf * 32 + g * 8 + rr * 4 + (li >> 2);
        const int c = p * 16 + (li & 3) * 4;                     // 4-half column chunk li & 3 of the part
        off[p][hf][rr] = tile_off(row, c >> 3) + (c & 7);
      }
  s16x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const s16x4 r = tr_read(tile + off[p][hf][rr]);
          acc += r;
          if (it == 0)
            for (int j = 0; j < 4; ++j) out[(((p * 2 + hf) * 2 + rr) * 64 + lane) * 4 + j] = (unsigned short)r[j];
        }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[0] = t1 - t0;
  if (acc[0] == 12345) out[0] = 1;
}

int main() {
  unsigned short *d, h[8 * 64 * 4];
  unsigned long long *dc, hc;
  hipMalloc(&d, sizeof(h));
  hipMalloc(&dc, 8);
  hipLaunchKernelGGL(probe_linear, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 64 * 4 * 2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int i = l & 15, grp = l >> 4;
      const int expect = grp * 64 + (4 * j + (i >> 2)) * 4 + (i & 3);     // hypothesis: R[i][j] = D[lane 4j + (i >> 2)][i & 3]
      if (h[l * 4 + j] != expect) ++bad;
    }
  printf("linear addresses: hypothesis R[i][j] = D[4j + (i >> 2)][i & 3] per 16-lane group: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
  if (bad)
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  const int reps = 2000;
  hipLaunchKernelGGL(probe_rows, dim3(1), dim3(64), 0, 0, d, dc, reps);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);
  bad = 0;
  for (int p = 0; p < 2; ++p)
    for (int hf = 0; hf < 2; ++hf)
      for (int rr = 0; rr < 2; ++rr)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 4; ++j) {
            const int li = l & 15, g = l >> 4;
            const int key = hf * 32 + g * 8 + rr * 4 + j, c = p * 16 + li;
            if (h[(((p * 2 + hf) * 2 + rr) * 64 + l) * 4 + j] != key * 32 + c) ++bad;
          }
  printf("rows-tile gather (lane (li, g) <- channel li, keys g*8 + rr*4 + j): %s (%d mismatches); %.1f cycles per tr read (one wave, 8 per iteration)\n",
         bad ? "WRONG" : "confirmed", bad, (double)hc / (reps * 8.0));
  return 0;
}
