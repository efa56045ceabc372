// 3x3 (stride 1, padding 1) convolutions of the frozen backbone's narrow layers as a bf16 MFMA implicit GEMM with the
// neighbouring BatchNorm work folded in (CLIP ModifiedResNet: the stem's conv2 / conv3 and layer1's conv2,
// model/utils/clip.py:22-43 -- 32 -> 32 and 32 -> 64 channels at 128 x 128, 64 -> 64 at 64 x 64 for 256 x 256 images).
//   y[n][oy][ox][co] = sum_{kh, kw, ci} f(x[n][oy + kh - 1][ox + kw - 1][ci]) * w[co][kh][kw][ci]      (0 outside the image)
//   f(v) = relu?(v * in_scale[ci] + in_shift[ci]) rounded to bf16 (identity when in_scale == NULL): BatchNorm-apply + ReLU of
//   the PRODUCER, which the unfused path materialises with a3d_bn_apply;   x, y bf16 NHWC;  w bf16 [Cout][3][3][Cin] (the
//   channels_last layout of the torch weight);  fp32 accumulation, one rounding;  optional epilogue: per-workgroup partial
//   (sum, sum of squares) of the ROUNDED outputs per channel, in the layout a3d_bn_finalize reduces ([slab][2][Cout]).
// Why these layers: at <= 64 channels the library convolutions reach 0.3 - 0.5 PFLOP/s and 1.7 - 2.3 TB/s -- neither roof
// (profiles/r04_conv_layers.json: 238 / 349 / 155 us against 67 / 101 / 34 us of HBM time) -- and each is surrounded by a
// BatchNorm-apply pass over its input (read + write) and a statistics pass over its output that this kernel absorbs.  From 128
// channels on the weights (9 Cin Cout bf16 >= 288 KB) do not fit LDS and CK is compute-bound at 0.7 - 1.0 PFLOP/s: MIOpen.
// Structure (the streaming scheme of conv1x1.hip): a persistent workgroup keeps ALL weights in LDS and walks 8 x 32-pixel output
// tiles; the (tile, 32-channel half) pairs form one flat sequence of steps whose 10 x 34-pixel halo tiles are fetched D steps
// ahead into registers (unconditional clamped loads), normalised and zero-padded on their way into one of two LDS buffers, one
// barrier per step.  Per step and wave: 9 taps x (4 pixel fragments + COUT/16 weight fragments) -> 36 COUT/16 MFMA 16x16x32,
// computed TRANSPOSED (A = weight rows, B = pixels) with the weight rows permuted so that a lane owns 8 / 16 consecutive
// channels of its pixel (16 / 32-byte stores, full lines per pixel).
// The kernel's first version was VALU-bound (172 / 296 / 134 us on the three layers, gpurun r04i: ~1100 VALU instructions per
// wave and step against 72 - 144 MFMA): software bf16 rounding, per-read swizzle arithmetic and three integer divisions per step.
// This one rounds with v_cvt_pk_bf16_f32, normalises with packed fma, reads fragments at precomputed offsets and walks its
// tiles with a carry-propagating cursor.
#include "a3d_common.h"
#include "../../include/act3d_hip.h"

namespace a3d {

constexpr int C3_TH = 8, C3_TW = 32;                              // output tile
constexpr int C3_HW = C3_TW + 2, C3_HP = (C3_TH + 2) * C3_HW;     // halo tile: 10 x 34 = 340 pixels
constexpr int C3_XL = (C3_HP * 4 + 255) / 256;                    // 16-byte segments each thread stages per step: 6

// LDS rows are 64 bytes (32 channels).  Weight row r (aligned blocks of Cout rows): 16-byte segment g stored at g ^ (bit 1 | bit 4 << 1 of r)
// -- the fragments read the PERMUTED rows (i >> 2) 4 NT + tn 4 + (i & 3), for which this keying is conflict-free and (r >> 1) & 3 is 2-way
// conflicted (tests/test_conv3x3_layout_cpu.py).
// Halo pixel hp = hy 34 + hx: segment g stored at g ^ ((hx >> 1) & 3) -- keyed on the COLUMN, so that a tap's row shift (kh 34
// pixels) is a constant byte offset of the read while 16 consecutive pixels from any base stay conflict-free for ds_read_b128
// (checked by enumeration over the instruction's lane groups for every base and row offset).
__device__ __forceinline__ int c3_woff(int row, int seg) { return row * 32 + ((seg ^ (((row >> 1) & 1) | (((row >> 4) & 1) << 1))) << 3); }
__device__ __forceinline__ int c3_xoff(int hp, int hx, int seg) { return hp * 32 + ((seg ^ ((hx >> 1) & 3)) << 3); }

typedef float c3_f32x2 __attribute__((ext_vector_type(2)));
typedef short c3_s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int c3_u32x4 __attribute__((ext_vector_type(4)));

// tile cursor (workgroup-uniform): tile = (img tiles_y + ty) tiles_x + tx, advanced by the grid size with carries -- no division per step
struct C3Cursor {
  int img, ty, tx;
  __device__ __forceinline__ void advance(int dimg, int dty, int dtx, int tiles_y, int tiles_x) {
    tx += dtx;
    if (tx >= tiles_x) { tx -= tiles_x; ++ty; }
    ty += dty;
    if (ty >= tiles_y) { ty -= tiles_y; ++img; }
    img += dimg;
  }
};

// dynamic LDS of one workgroup: weights + two halo buffers + scale / shift
constexpr int c3_lds_bytes(int halves, int nt) { return halves * 9 * 16 * nt * 64 + 2 * C3_HP * 64 + 2 * 32 * halves * 4; }

// COUT = 16 NT output channels per workgroup: blockIdx.y selects the block of COUT channels of the layer's `ctot` (64 -> 64 runs as
// two 32-channel blocks: half the weights per workgroup = two workgroups per CU instead of one, which hides the per-step barrier /
// LDS / global-load latencies a single wave per SIMD exposes: 105 -> see profiles/r04_conv3x3_probe.json)
// LIST (round 6, the token-sparse input gradient of the FPN's output convolution, a3d_conv3x3_dgrad_tiles): the workgroups walk the
// ACTIVE tiles of `tlist` ((image << 16) | (tile row << 8) | tile column, tlist[-2] = their number) instead of every tile of the map,
// and afterwards zero-fill the inactive ones (listed from the END of the same array, tlist[-1] = their number): a tile whose 10 x 34
// halo holds no non-zero input pixel has an all-zero output.
template <int HALVES, int NT, int D, bool LIST = false>
__global__ __launch_bounds__(256, (2 * c3_lds_bytes(HALVES, NT) <= 160 * 1024 ? 2 : 1)) void conv3x3_stream_kernel(
    const unsigned short* __restrict__ x, const unsigned short* __restrict__ w, const float* __restrict__ in_scale,
    const float* __restrict__ in_shift, int in_relu, unsigned short* __restrict__ y, float* __restrict__ partial, int nimg, int H,
    int W, int ctot, const int* __restrict__ tlist = nullptr) {
  constexpr int CIN = 32 * HALVES, COUT = 16 * NT;
  const int co0 = blockIdx.y * COUT;
  static_assert(D % HALVES == 0, "the half of a step must be a compile-time function of its slot");
  extern __shared__ __attribute__((aligned(16))) unsigned short smem_c3[];
  unsigned short* Ws = smem_c3;                                   // [HALVES][9][COUT] rows of 32 channels
  unsigned short* Xs = Ws + HALVES * 9 * COUT * 32;               // [2][C3_HP] rows of 32 channels
  float* scS = reinterpret_cast<float*>(Xs + 2 * C3_HP * 32);     // [CIN] scale | [CIN] shift
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, g = lane >> 4;
  // workgroups of one XCD (blockIdx.x mod 8) take consecutive tiles, so that the halo rows neighbours share meet in one L2
  const int nwg = gridDim.x;
  const int lb = (nwg & 7) == 0 ? (int)(blockIdx.x & 7) * (nwg >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int tiles_x = W / C3_TW, tiles_y = H / C3_TH, tpi = tiles_x * tiles_y;
  const int ntiles = LIST ? __builtin_amdgcn_readfirstlane(tlist[-2]) : nimg * tpi;      // < 2^30: checked by the host
  for (int i = t; i < HALVES * 9 * COUT * 4; i += 256) {
    const int seg = i & 3, r = i >> 2, co = r % COUT, tap = (r / COUT) % 9, h = r / (COUT * 9);
    *reinterpret_cast<uint4*>(&Ws[c3_woff((h * 9 + tap) * COUT + co, seg)]) =
        *reinterpret_cast<const uint4*>(w + ((size_t)(co0 + co) * 9 + tap) * CIN + h * 32 + seg * 8);
  }
  for (int i = t; i < CIN; i += 256) { scS[i] = in_scale ? in_scale[i] : 1.f; scS[CIN + i] = in_scale ? in_shift[i] : 0.f; }
  // this thread's halo segments: pixel hp = (t + 256 i) / 4 of the 10 x 34 tile, 16-byte segment (t & 3)
  const int seg_t = t & 3;
  int hy[C3_XL], hx[C3_XL];
#pragma unroll
  for (int i = 0; i < C3_XL; ++i) {
    const int hp = (t + i * 256) >> 2;
    hy[i] = hp / C3_HW;
    hx[i] = hp - hy[i] * C3_HW;
  }
  c3_f32x2 ssum[NT][2], ssq[NT][2];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int r = 0; r < 2; ++r) { ssum[a][r] = c3_f32x2{0.f, 0.f}; ssq[a][r] = c3_f32x2{0.f, 0.f}; }
  const int my_tiles = lb < ntiles ? (ntiles - lb + nwg - 1) / nwg : 0;
  const int total = my_tiles * HALVES;                            // flat steps: step s = (this workgroup's tile s / HALVES, half s % HALVES)
  // one-time divisions: the first tile and the per-tile advance (grid size) in (image, tile row, tile column) digits
  const int dimg = nwg / tpi, dty = (nwg - dimg * tpi) / tiles_x, dtx = nwg - dimg * tpi - dty * tiles_x;
  C3Cursor cl;                                                    // the LOAD stream's tile (D steps ahead of the compute stream's)
  auto from_list = [&](int ordinal) {                             // LIST: this workgroup's ordinal-th active tile
    const int e = __builtin_amdgcn_readfirstlane(tlist[lb + ordinal * nwg]);
    return C3Cursor{e >> 16, (e >> 8) & 255, e & 255};
  };
  if (LIST) {
    cl = my_tiles > 0 ? from_list(0) : C3Cursor{0, 0, 0};
  } else {
    cl.img = lb / tpi;
    cl.ty = (lb - cl.img * tpi) / tiles_x;
    cl.tx = lb - cl.img * tpi - cl.ty * tiles_x;
  }
  C3Cursor cc = cl;                                               // the compute stream's tile
  int lk = 0, lhalf = 0;                                          // load stream: tile ordinal, half
  int ck = 0;                                                     // compute stream: tile ordinal (LIST)
  auto load_next = [&](uint4 (&r)[C3_XL]) {
    const int y0 = cl.ty * C3_TH, x0 = cl.tx * C3_TW;
    // uniform 64-bit image base + 32-bit byte offset per lane (one image is at most 2^31 bytes: checked by the host)
    const char* base = reinterpret_cast<const char*>(x + (size_t)cl.img * H * W * CIN + lhalf * 32);
#pragma unroll
    for (int i = 0; i < C3_XL; ++i) {
      const int iy = min(max(y0 - 1 + hy[i], 0), H - 1), ix = min(max(x0 - 1 + hx[i], 0), W - 1);   // clamped: masked at the stage
      const unsigned int off = (unsigned int)((iy * W + ix) * CIN + seg_t * 8) * 2u;
      r[i] = *reinterpret_cast<const uint4*>(base + off);
    }
    if (++lhalf == HALVES) {
      lhalf = 0;
      if (lk + 1 < my_tiles) {                                    // past the end: the last tile again (never staged)
        ++lk;
        if (LIST) cl = from_list(lk); else cl.advance(dimg, dty, dtx, tiles_y, tiles_x);
      }
    }
  };
  // BatchNorm-apply + ReLU of the producer on 8 channels: packed fma, hardware bf16 rounding (v_cvt_pk_bf16_f32, RNE), ReLU as a
  // packed signed-integer max on the bf16 pairs (negative floats are negative int16; lower bound -32768 = no ReLU)
  const c3_s16x2 relu_lo = in_relu ? c3_s16x2{0, 0} : c3_s16x2{(short)-32768, (short)-32768};
  auto stage = [&](int buf, int half, const uint4 (&r)[C3_XL]) {
    const int y0 = cc.ty * C3_TH, x0 = cc.tx * C3_TW;
    const int k0 = half * 32 + seg_t * 8;
    c3_f32x2 sc[4], sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sc[j] = c3_f32x2{scS[k0 + 2 * j], scS[k0 + 2 * j + 1]};
      sh[j] = c3_f32x2{scS[CIN + k0 + 2 * j], scS[CIN + k0 + 2 * j + 1]};
    }
#pragma unroll
    for (int i = 0; i < C3_XL; ++i) {
      const int hp = (t + i * 256) >> 2;
      const int iy = y0 - 1 + hy[i], ix = x0 - 1 + hx[i];
      const bool inside = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      unsigned int u[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c3_f32x2 v = {__uint_as_float(u[j] << 16), __uint_as_float(u[j] & 0xFFFF0000u)};
        v = v * sc[j] + sh[j];
        const c3_s16x2 pk = __builtin_elementwise_max(__builtin_bit_cast(c3_s16x2, pk_bf16(v.x, v.y)), relu_lo);
        u[j] = inside ? __builtin_bit_cast(unsigned int, pk) : 0u;      // the convolution's zero padding (of the NORMALISED map)
      }
      if (hp < C3_HP) *reinterpret_cast<c3_u32x4*>(&Xs[buf * C3_HP * 32 + c3_xoff(hp, hx[i], seg_t)]) = c3_u32x4{u[0], u[1], u[2], u[3]};
    }
  };
  // fragment addressing.  Pixels: m-tile tm of this wave = tile row 2 wave + (tm >> 1), columns (tm & 1) 16 + li; tap (kh, kw)
  // reads halo pixel (row + kh) 34 + col + kw: xo[tm][kw] + kh 34 rows of 64 B (the segment swizzle depends on li + kw only).
  // Weights: MFMA tile tn row i <-> channel (i >> 2) 4 NT + tn 4 + (i & 3), so the result rows g 4 + r of tile tn are the channels
  // g 4 NT + tn 4 + r: a lane holds 4 NT consecutive channels of its pixel
  int xo[4][3], woff[NT];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int col = (tm & 1) * 16 + li + kw;
      xo[tm][kw] = c3_xoff((2 * wave + (tm >> 1)) * C3_HW + col, col, g);
    }
#pragma unroll
  for (int tn = 0; tn < NT; ++tn) woff[tn] = c3_woff((li >> 2) * (4 * NT) + tn * 4 + (li & 3), g);
  if (total > 0) {
    uint4 xr[D][C3_XL];
#pragma unroll
    for (int j = 0; j < D; ++j) load_next(xr[j]);
    f32x4 acc[NT][4];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                              // W, scale / shift staged
    for (int s0 = 0; s0 < total; s0 += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int s = s0 + j;
        if (s >= total) break;                                    // workgroup-uniform
        const int half = j % HALVES, buf = s & 1;                 // s0 is a multiple of D, D of HALVES
        stage(buf, half, xr[j]);
        load_next(xr[j]);                                         // unconditional: D steps ahead
        __syncthreads();                                          // stage(s) visible; every wave is past its reads of the other buffer
        const unsigned short* Xb = Xs + buf * C3_HP * 32;
        const unsigned short* Wb = Ws + half * 9 * COUT * 32;
        auto taps_of_row = [&](int kh) {
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            s16x8 xa[4], wb[NT];
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) xa[tm] = *reinterpret_cast<const s16x8*>(&Xb[xo[tm][kw] + kh * (C3_HW * 32)]);
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) wb[tn] = *reinterpret_cast<const s16x8*>(&Wb[(kh * 3 + kw) * COUT * 32 + woff[tn]]);
#pragma unroll
            for (int tn = 0; tn < NT; ++tn)
#pragma unroll
              for (int tm = 0; tm < 4; ++tm) acc[tn][tm] = mfma_bf16_16x16x32(wb[tn], xa[tm], acc[tn][tm]);
          }
        };
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) taps_of_row(kh);
        if (half == HALVES - 1) {
          // tile done: round once (RNE), statistics of the rounded values, 4 NT consecutive channels per lane and pixel
          const int y0 = cc.ty * C3_TH, x0 = cc.tx * C3_TW;
#pragma unroll
          for (int tm = 0; tm < 4; ++tm) {
            const int oy = y0 + 2 * wave + (tm >> 1), ox = x0 + (tm & 1) * 16 + li;
            unsigned int pk[2 * NT];
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
#pragma unroll
              for (int r2 = 0; r2 < 2; ++r2) {
                const unsigned int u = pk_bf16(acc[tn][tm][2 * r2], acc[tn][tm][2 * r2 + 1]);
                const c3_f32x2 v = {__uint_as_float(u << 16), __uint_as_float(u & 0xFFFF0000u)};
                ssum[tn][r2] += v;
                ssq[tn][r2] += v * v;
                pk[2 * tn + r2] = u;
              }
              acc[tn][tm] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            char* ybase = reinterpret_cast<char*>(y + (size_t)cc.img * H * W * ctot + co0);
            c3_u32x4* dst = reinterpret_cast<c3_u32x4*>(ybase + (unsigned int)((oy * W + ox) * ctot + g * (4 * NT)) * 2u);
#pragma unroll
            for (int q = 0; q < NT / 2; ++q) dst[q] = c3_u32x4{pk[4 * q], pk[4 * q + 1], pk[4 * q + 2], pk[4 * q + 3]};
          }
          if (LIST) { if (++ck < my_tiles) cc = from_list(ck); } else cc.advance(dimg, dty, dtx, tiles_y, tiles_x);
        }
      }
    }
  }
  if (LIST) {
    // zero-fill of the inactive tiles: 8 x 32 pixels x COUT channels = one pixel's COUT-channel slice per thread
    const int nzero = __builtin_amdgcn_readfirstlane(tlist[-1]);
    const int last = nimg * tpi - 1;                              // inactive tiles are listed downwards from the end of the array
    for (int k = lb; k < nzero; k += nwg) {
      const int e = __builtin_amdgcn_readfirstlane(tlist[last - k]);
      const int img = e >> 16, oy = ((e >> 8) & 255) * C3_TH + (t >> 5), ox = (e & 255) * C3_TW + (t & 31);
      char* ybase = reinterpret_cast<char*>(y + (size_t)img * H * W * ctot + co0);
      c3_u32x4* dst = reinterpret_cast<c3_u32x4*>(ybase + (unsigned int)((oy * W + ox) * ctot) * 2u);
#pragma unroll
      for (int q = 0; q < COUT / 8; ++q) dst[q] = c3_u32x4{0u, 0u, 0u, 0u};
    }
  }
  if (!partial) return;
  __syncthreads();                                                // every wave is past its last fragment reads: Xs is free
  float* redS = reinterpret_cast<float*>(Xs);                     // [4 waves][COUT] sums | [4][COUT] sums of squares
  float* redQ = redS + 4 * COUT;
#pragma unroll
  for (int tn = 0; tn < NT; ++tn)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sv = ssum[tn][r >> 1][r & 1], q = ssq[tn][r >> 1][r & 1];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { sv += __shfl_xor(sv, o, 64); q += __shfl_xor(q, o, 64); }
      if (li == 0) { redS[wave * COUT + g * (4 * NT) + tn * 4 + r] = sv; redQ[wave * COUT + g * (4 * NT) + tn * 4 + r] = q; }
    }
  __syncthreads();
  if (t < COUT) {
    float sv = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sv += redS[j * COUT + t]; q += redQ[j * COUT + t]; }
    float* p = partial + (size_t)blockIdx.x * 2 * ctot + co0;
    p[t] = sv;
    p[ctot + t] = q;
  }
}

// ---- token-sparse input gradient of the FPN's 3x3 output convolution (round 6) ----------------------------------------------------------
// The gradient map of that convolution's output is non-zero only on the pixels the pyramid levels gathered (act3d.py:244-260), so its
// input gradient is non-zero only on their 3x3 neighbourhoods.  mark: every gathered token marks the (at most four) 8 x 32 output tiles
// its neighbourhood touches; compact: the marks become the list the LIST variant of the stream kernel walks.
__global__ __launch_bounds__(256) void c3_mark_tiles_kernel(const long long* __restrict__ idx, long long ntok, int k, int ncam, int H,
                                                          int W, unsigned char* __restrict__ mask) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= ntok) return;
  const long long p = idx[i];
  const int b = (int)(i / k);
  const int cam = (int)(p / ((long long)H * W));
  const int rem = (int)(p - (long long)cam * H * W);
  const int h = rem / W, w0 = rem - h * W;
  const int tiles_x = W / C3_TW, tiles_y = H / C3_TH;
  const int img = b * ncam + cam;
  const int ty0 = max(h - 1, 0) / C3_TH, ty1 = min(h + 1, H - 1) / C3_TH, tx0 = max(w0 - 1, 0) / C3_TW, tx1 = min(w0 + 1, W - 1) / C3_TW;
  for (int ty = ty0; ty <= ty1; ++ty)
    for (int tx = tx0; tx <= tx1; ++tx) mask[((size_t)img * tiles_y + ty) * tiles_x + tx] = 1;       // (racing stores of the same value)
}

// one workgroup of 1024 threads: active tiles in ascending order from the front of tlist, inactive ones from the back; the two
// counts in front of it (tlist[-2], tlist[-1]).  Deterministic (an exclusive scan, no atomics).
__global__ __launch_bounds__(1024) void c3_compact_tiles_kernel(const unsigned char* __restrict__ mask, int ntiles, int tiles_x,
                                                              int tiles_y, int* __restrict__ tlist) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int tpi = tiles_x * tiles_y;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < ntiles; base += 1024) {
    const int id = base + t;
    const int on = (id < ntiles && mask[id]) ? 1 : 0;
    const unsigned long long bal = __ballot(on);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = carry;
    for (int w2 = 0; w2 < wave; ++w2) off += wsum[w2];
    if (id < ntiles) {
      const int img = id / tpi, r = id - img * tpi, ty = r / tiles_x, tx = r - ty * tiles_x;
      const int e = (img << 16) | (ty << 8) | tx;
      const int pos_on = off + before;                            // active tiles before this one
      if (on) tlist[pos_on] = e; else tlist[ntiles - 1 - (id - pos_on)] = e;
    }
    __syncthreads();
    if (t == 0) { int tot = 0; for (int w2 = 0; w2 < 16; ++w2) tot += wsum[w2]; carry += tot; }
    __syncthreads();
  }
  if (t == 0) { tlist[-2] = carry; tlist[-1] = ntiles - carry; }
}

}  // namespace a3d

using namespace a3d;

// output channels per workgroup: 64 -> 64 splits into two blocks of 32 (two workgroups per CU)
static int c3_cblock(int Cin, int Cout) {
  static const bool split = !(getenv("A3D_C3_SPLIT") && atoi(getenv("A3D_C3_SPLIT")) == 0);      // A/B switch: 0 = one 64-channel block
  return (Cin == 64 && Cout == 64 && split) ? 32 : Cout;
}
static size_t c3_lds(int Cin, int Cout) { return (size_t)c3_lds_bytes(Cin / 32, c3_cblock(Cin, Cout) / 16); }
// served: 32 -> 32, 32 -> 64, 64 -> 64 channels on maps of 8 j x 32 k pixels
static bool c3_serves(int Cin, int Cout, int H, int W) {
  const bool ch = (Cin == 32 && (Cout == 32 || Cout == 64)) || (Cin == 64 && Cout == 64);
  return ch && H > 0 && W > 0 && (H % C3_TH) == 0 && (W % C3_TW) == 0;
}
static int c3_slabs(size_t nimg, int H, int W, int Cin, int Cout) {
  const size_t ntiles = nimg * (size_t)(H / C3_TH) * (size_t)(W / C3_TW);
  const int per_cu = 2 * c3_lds(Cin, Cout) <= 160 * 1024 ? 2 : 1;                    // resident workgroups (LDS-limited)
  const int yblocks = Cout / c3_cblock(Cin, Cout);
  return (int)std::min<size_t>(ntiles, (size_t)std::max(8, 256 * per_cu / yblocks));
}

extern "C" int a3d_conv3x3_serves(int Cin, int Cout, int H, int W) { return c3_serves(Cin, Cout, H, W) ? 1 : 0; }

extern "C" int a3d_conv3x3_nslab(size_t nimg, int H, int W, int Cin, int Cout) {
  if (nimg == 0 || !c3_serves(Cin, Cout, H, W)) return 0;
  return c3_slabs(nimg, H, W, Cin, Cout);
}

extern "C" int a3d_conv3x3_bn_fwd(const void* x, const void* w, const float* in_scale, const float* in_shift, int in_relu, void* y,
                                  float* partial, size_t nimg, int H, int W, int Cin, int Cout, void* stream) {
  if (!x || !w || !y || nimg == 0 || nimg * (size_t)(H > 0 ? H : 1) * (size_t)(W > 0 ? W : 1) / 256 >= (size_t)(1 << 30) || (size_t)(H > 0 ? H : 1) * (size_t)(W > 0 ? W : 1) * 128 >= ((size_t)1 << 31) || !c3_serves(Cin, Cout, H, W) || (in_scale && !in_shift) ||
      ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) != 0)) {
    set_error("a3d_conv3x3_bn_fwd: bad argument (images=%zu H=%d W=%d Cin=%d Cout=%d; served: 32 -> 32, 32 -> 64, 64 -> 64 channels, "
              "H a multiple of 8, W a multiple of 32 -- a3d_conv3x3_serves; 16-byte aligned operands)", nimg, H, W, Cin, Cout);
    return A3D_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  const unsigned short* xs = (const unsigned short*)x;
  const unsigned short* ws = (const unsigned short*)w;
  unsigned short* ys = (unsigned short*)y;
  const int slabs = c3_slabs(nimg, H, W, Cin, Cout);
  const size_t lds = c3_lds(Cin, Cout);
#define A3D_C3S(HV, NTV, DV)                                                                                                        \
  do {                                                                                                                           \
    static bool once = false;                                                                                                    \
    if (!once) { (void)hipFuncSetAttribute((const void*)conv3x3_stream_kernel<HV, NTV, DV>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); once = true; } \
    hipLaunchKernelGGL((conv3x3_stream_kernel<HV, NTV, DV>), dim3(slabs, Cout / c3_cblock(Cin, Cout)), dim3(256), lds, s, xs, ws, in_scale, in_shift, in_relu, ys, partial, (int)nimg, H, W, Cout); \
  } while (0)
  if (Cin == 32 && Cout == 32) A3D_C3S(1, 2, 3);
  else if (Cin == 32 && Cout == 64) A3D_C3S(1, 4, 2);
  else if (c3_cblock(Cin, Cout) == 32) A3D_C3S(2, 2, 2);
  else A3D_C3S(2, 4, 2);
#undef A3D_C3S
  return check_launch("a3d_conv3x3_bn_fwd");
}

// ---- token-sparse input gradient (see the kernels above).  ws: ints = a3d_conv3x3_dgrad_tiles_ws_ints: [2 counts | tile list];
// mask: one byte per 8 x 32 tile of the map, zeroed by the caller before the first a3d_conv3x3_mark_tiles of a backward pass.
extern "C" size_t a3d_conv3x3_tile_count(size_t nimg, int H, int W) {
  return (H > 0 && W > 0 && (H % C3_TH) == 0 && (W % C3_TW) == 0) ? nimg * (size_t)(H / C3_TH) * (size_t)(W / C3_TW) : 0;
}
extern "C" size_t a3d_conv3x3_dgrad_tiles_ws_ints(size_t nimg, int H, int W) { return a3d_conv3x3_tile_count(nimg, H, W) + 4; }

extern "C" int a3d_conv3x3_mark_tiles(const long long* idx, int B, int k, int ncam, int H, int W, unsigned char* mask, void* stream) {
  if (!idx || !mask || B <= 0 || k <= 0 || ncam <= 0 || a3d_conv3x3_tile_count(1, H, W) == 0 || H / C3_TH > 255 || W / C3_TW > 255) {
    set_error("a3d_conv3x3_mark_tiles: bad argument (B=%d k=%d ncam=%d H=%d W=%d; H a multiple of 8, W of 32)", B, k, ncam, H, W);
    return A3D_ERR_ARG;
  }
  const long long ntok = (long long)B * k;
  hipLaunchKernelGGL(c3_mark_tiles_kernel, dim3((unsigned)((ntok + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, ntok, k, ncam, H, W, mask);
  return check_launch("a3d_conv3x3_mark_tiles");
}

// dx = the 3x3 (stride 1, padding 1) convolution of dy with wt on the tiles marked in `mask`, zeros elsewhere.  dy, dx bf16 NHWC
// [nimg][H][W][64]; wt bf16 [64 ci][3][3][64 co] = the forward weight flipped and transposed, wt[ci][kh][kw][co] = w[co][ci][2-kh][2-kw].
extern "C" int a3d_conv3x3_dgrad_tiles(const void* dy, const void* wt, const unsigned char* mask, int* ws, void* dx, size_t nimg, int H,
                                       int W, void* stream) {
  if (!dy || !wt || !mask || !ws || !dx || nimg == 0 || nimg >= 32768 || !c3_serves(64, 64, H, W) || H / C3_TH > 255 || W / C3_TW > 255 ||
      (size_t)H * (size_t)W * 128 >= ((size_t)1 << 31) || ((((uintptr_t)dy | (uintptr_t)wt | (uintptr_t)dx) & 15) != 0)) {
    set_error("a3d_conv3x3_dgrad_tiles: bad argument (images=%zu H=%d W=%d; 64 channels, H a multiple of 8, W of 32, < 32768 images)", nimg, H, W);
    return A3D_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  const int ntiles = (int)a3d_conv3x3_tile_count(nimg, H, W);
  int* tlist = ws + 2;
  hipLaunchKernelGGL(c3_compact_tiles_kernel, dim3(1), dim3(1024), 0, s, mask, ntiles, W / C3_TW, H / C3_TH, tlist);
  int rc = check_launch("a3d_conv3x3_dgrad_tiles(compact)");
  if (rc) return rc;
  const int slabs = c3_slabs(nimg, H, W, 64, 64);
  const size_t lds = c3_lds(64, 64);
  const int cb = c3_cblock(64, 64);
  static bool once = false;
  if (cb == 32) {
    if (!once) { (void)hipFuncSetAttribute((const void*)conv3x3_stream_kernel<2, 2, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); once = true; }
    hipLaunchKernelGGL((conv3x3_stream_kernel<2, 2, 2, true>), dim3(slabs, 2), dim3(256), lds, s, (const unsigned short*)dy, (const unsigned short*)wt,
                       (const float*)nullptr, (const float*)nullptr, 0, (unsigned short*)dx, (float*)nullptr, (int)nimg, H, W, 64, (const int*)tlist);
  } else {
    if (!once) { (void)hipFuncSetAttribute((const void*)conv3x3_stream_kernel<2, 4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); once = true; }
    hipLaunchKernelGGL((conv3x3_stream_kernel<2, 4, 2, true>), dim3(slabs, 1), dim3(256), lds, s, (const unsigned short*)dy, (const unsigned short*)wt,
                       (const float*)nullptr, (const float*)nullptr, 0, (unsigned short*)dx, (float*)nullptr, (int)nimg, H, W, 64, (const int*)tlist);
  }
  return check_launch("a3d_conv3x3_dgrad_tiles");
}
