// Scene-token construction for the Act3D coarse-to-fine loop: HBM-bound gather/select kernels.
//
//   a3d_pcd_downsample   reference: F.interpolate(pcd, 1/f, 'bilinear') + rearrange (act3d.py:379-383,
//                        encoder.py:147-158).  For even f the bilinear sample is the mean of the 2x2 block at
//                        (f*y + f/2 - 1, f*x + f/2 - 1); the additions are ordered as ATen's CPU kernels order them
//                        (see the kernel) so the result is bit-identical to the reference run on CPU.
//   a3d_knn_topk         reference: l2 = ((pos - pcd)**2).sum(-1).sqrt(); topk(k, largest=False).indices
//                        (act3d.py:244-245).  Radix-select on the fp32 bit pattern + bitonic sort of the k
//                        survivors in LDS; order = ascending (distance, index), i.e. torch's sorted order with a
//                        defined tie-break.
//   a3d_build_context    reference: per-sample python gathers + torch.cat with the gripper token
//                        (act3d.py:247-260).  One pass: ctx[b] = [feat[b][idx[b]] | extra[b]].
//   a3d_build_context_bwd scatter of d(ctx) back to the (zero-initialised) feature-map gradient.
#include "a3d_common.h"
#include "../../include/act3d_hip.h"

namespace a3d {

__global__ __launch_bounds__(256) void pcd_downsample_kernel(
    const float* __restrict__ pcd, float* __restrict__ out, int BC, int C, int Hin, int Win, int f) {
  const int h = Hin / f, w = Win / f;
  const size_t total = (size_t)BC * h * w;
  const int off = f / 2 - 1;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % w);
    const int y = (int)((idx / w) % h);
    const int bc = (int)(idx / ((size_t)w * h));
    const int y0 = f * y + off, x0 = f * x + off;
    const int y1 = min(y0 + 1, Hin - 1), x1 = min(x0 + 1, Win - 1);
    const int b = bc / C, cam = bc - b * C;
    float* o = out + (((size_t)b * C + cam) * h * w + (size_t)y * w + x) * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float* p = pcd + ((size_t)bc * 3 + ch) * Hin * Win;
      const float p00 = p[(size_t)y0 * Win + x0], p01 = p[(size_t)y0 * Win + x1];
      const float p10 = p[(size_t)y1 * Win + x0], p11 = p[(size_t)y1 * Win + x1];
      // Bit-compatibility with the reference's CPU run: ATen evaluates w00*a + w01*b + w10*c + w11*d left to
      // right when out_h + out_w <= 128 (its "vectorized" kernel) and separably, rows first, otherwise
      // (aten/src/ATen/native/cpu/UpSampleKernel.cpp, _use_vectorized_kernel_cond_2d).  All weights are powers of
      // two, so every product is exact and only the order of the additions matters.
      if (h + w <= 128) {
        o[ch] = add_rn(add_rn(add_rn(0.25f * p00, 0.25f * p01), 0.25f * p10), 0.25f * p11);
      } else {
        const float t0 = add_rn(0.5f * p00, 0.5f * p01);
        const float t1 = add_rn(0.5f * p10, 0.5f * p11);
        o[ch] = add_rn(0.5f * t0, 0.5f * t1);
      }
    }
  }
}

// d = sqrt((px-x)^2 + (py-y)^2 + (pz-z)^2) in the reference's operation order, no fma contraction
__device__ __forceinline__ float l2_dist(float px, float py, float pz, const float* q) {
  const float dx = sub_rn(px, q[0]), dy = sub_rn(py, q[1]), dz = sub_rn(pz, q[2]);
  const float s = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
  return sqrt_rn(s);
}

// distance of a scene point to the nearest of `npos` query positions (npos = 1: the Act3D k-NN centre, act3d.py:247-248;
// npos = L: find_traj_nn of the multi-scale diffusion head, model/utils/utils.py:39-48, which ranks SQUARED distances)
__device__ __forceinline__ float nn_dist(const float* __restrict__ pos, int npos, int squared, const float* q) {
  float best = 0.f;
  for (int l = 0; l < npos; ++l) {
    const float dx = sub_rn(pos[l * 3 + 0], q[0]), dy = sub_rn(pos[l * 3 + 1], q[1]), dz = sub_rn(pos[l * 3 + 2], q[2]);
    const float s = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
    const float d = squared ? s : sqrt_rn(s);
    best = (l == 0) ? d : fminf(best, d);
  }
  return best;
}

constexpr int TK_THREADS = 1024;

// One workgroup per sample.  dist_ws: [B][N] uint32 scratch.  keys: dynamic LDS, kpad uint64.
__global__ __launch_bounds__(TK_THREADS) void knn_topk_kernel(
    const float* __restrict__ pos, const float* __restrict__ xyz, unsigned int* __restrict__ dist_ws,
    long long* __restrict__ idx_out, float* __restrict__ dist_out, int N, int k, int kpad, int npos, int squared) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  __shared__ unsigned int hist[256];
  __shared__ unsigned int sh_prefix, sh_krem, sh_count, sh_neq;
  const int b = blockIdx.x, t = threadIdx.x;
  const float* qp = pos + (size_t)b * npos * 3;
  const float* pts = xyz + (size_t)b * N * 3;
  unsigned int* dw = dist_ws + (size_t)b * N;

  for (int i = t; i < N; i += TK_THREADS) dw[i] = __float_as_uint(nn_dist(qp, npos, squared, pts + (size_t)i * 3));
  if (t == 0) { sh_prefix = 0; sh_krem = (unsigned int)k; }
  __syncthreads();

  // ---- 4 radix passes over the value bits (distances are >= 0: bit pattern is monotonic)
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (t < 256) hist[t] = 0;
    __syncthreads();
    const unsigned int prefix = sh_prefix;
    const unsigned int himask = (pass == 0) ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = t; i < N; i += TK_THREADS) {
      const unsigned int v = dw[i];
      if ((v & himask) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (t == 0) {
      unsigned int krem = sh_krem, cum = 0;
      int bin = 0;
      for (; bin < 256; ++bin) {
        if (cum + hist[bin] >= krem) break;
        cum += hist[bin];
      }
      if (bin > 255) bin = 255;
      sh_krem = krem - cum;
      sh_prefix = prefix | ((unsigned int)bin << shift);
      sh_neq = hist[bin];
    }
    __syncthreads();
  }
  const unsigned int T = sh_prefix;       // k-th smallest value
  unsigned int need = sh_krem;            // how many elements == T are wanted
  unsigned int idx_thr = 0xFFFFFFFFu;     // elements == T with idx <= idx_thr are taken
  if (sh_neq != need) {
    // ties at the threshold: select the `need` lowest indices among bits == T (3 radix passes on idx)
    __syncthreads();
    if (t == 0) { sh_prefix = 0; sh_krem = need; }
    __syncthreads();
    for (int pass = 0; pass < 3; ++pass) {
      const int shift = 16 - 8 * pass;
      if (t < 256) hist[t] = 0;
      __syncthreads();
      const unsigned int prefix = sh_prefix;
      const unsigned int himask = (pass == 0) ? 0u : (0xFFFFFFFFu << (shift + 8));
      for (int i = t; i < N; i += TK_THREADS) {
        if (dw[i] == T && (((unsigned int)i) & himask) == prefix) atomicAdd(&hist[(((unsigned int)i) >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (t == 0) {
        unsigned int krem = sh_krem, cum = 0;
        int bin = 0;
        for (; bin < 256; ++bin) {
          if (cum + hist[bin] >= krem) break;
          cum += hist[bin];
        }
        if (bin > 255) bin = 255;
        sh_krem = krem - cum;
        sh_prefix = prefix | ((unsigned int)bin << shift);
      }
      __syncthreads();
    }
    idx_thr = sh_prefix;
  }
  // ---- collect survivors
  if (t == 0) sh_count = 0;
  for (int i = t; i < kpad; i += TK_THREADS) keys[i] = 0xFFFFFFFFFFFFFFFFull;
  __syncthreads();
  for (int i = t; i < N; i += TK_THREADS) {
    const unsigned int v = dw[i];
    if (v < T || (v == T && (unsigned int)i <= idx_thr)) {
      const unsigned int slot = atomicAdd(&sh_count, 1u);
      if (slot < (unsigned int)kpad) keys[slot] = ((unsigned long long)v << 32) | (unsigned int)i;
    }
  }
  __syncthreads();
  // ---- bitonic sort of kpad 64-bit keys (value major, index minor)
  for (int size = 2; size <= kpad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = t; i < (kpad >> 1); i += TK_THREADS) {
        const int lo = (i / stride) * (stride << 1) + (i % stride);
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const unsigned long long a = keys[lo], c = keys[hi];
        if ((a > c) == up) { keys[lo] = c; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = t; i < k; i += TK_THREADS) {
    const unsigned long long kk = keys[i];
    idx_out[(size_t)b * k + i] = (long long)(kk & 0xFFFFFFFFull);
    if (dist_out) dist_out[(size_t)b * k + i] = __uint_as_float((unsigned int)(kk >> 32));
  }
}

// hist[bin] += 1 for every lane with pred, with the wave's most common case -- many lanes in the same bin (distances
// share their exponent bits) -- folded into one LDS atomic: up to two rounds of "leader's bin" peeling, then plain atomics.
__device__ __forceinline__ void hist_add_wave(unsigned int* hist, unsigned int bin, bool pred, int lane) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const unsigned long long act = __ballot(pred);
    if (act == 0ull) return;
    const int leader = __ffsll((long long)act) - 1;
    const unsigned int lb = __builtin_amdgcn_readlane(bin, leader);
    const unsigned long long same = __ballot(pred && bin == lb);
    if (lane == leader) atomicAdd(&hist[lb], (unsigned int)__popcll(same));
    pred = pred && (bin != lb);
  }
  if (pred) atomicAdd(&hist[bin], 1u);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Two-launch selection for the shapes of the training step (N = 65 536 fine-map points, k = 4096; one 1024-thread workgroup
// per sample took 206 us on 64 of 256 CUs -- round-3 review): the distance pass and a first, coarse histogram are spread over
// KN_PARTS workgroups per sample; the second launch needs only cheap passes over the stored bit patterns and a sort that
// keeps most of its exchange stages inside the wave.
//   knn_dist_hist_kernel   grid (KN_PARTS, B): distances (IEEE, reference operation order) -> ws_dist [B][N] uint32; histogram
//                          of bits [30:20] (8 exponent + 3 mantissa bits: 2048 bins, wave-aggregated LDS atomics) -> plain
//                          stores into ws_hist [B][KN_PARTS][2048] (no global atomics, nothing to zero)
//   knn_select_sort_kernel grid (B): sums the part histograms, scans them for the bin holding the k-th smallest; one pass over
//                          ws_dist puts every element of a LOWER bin straight into the key array ("definitely in") and the
//                          elements OF that bin into a candidate list in LDS; radix passes over the candidates' remaining 20
//                          value bits (then index bits among ties of the threshold value) pick exactly the missing ones;
//                          bitonic sort of the k 64-bit (distance, index) keys with four keys per thread -- strides 1, 2 in
//                          registers, 4 .. 128 by lane exchange inside the wave, only strides >= 256 through LDS (10 barrier
//                          stages for 4096 keys instead of 78).  A candidate list that overflows its LDS budget (a dense
//                          shell of equidistant points) falls back to radix passes over ws_dist filtered by the bin.
// Same result as knn_topk_kernel: ascending (distance, index), bit for bit.
constexpr int KN_PARTS = 8;
constexpr int KN_BINS = 2048;
constexpr int KN_SHIFT = 20;

__global__ __launch_bounds__(TK_THREADS) void knn_dist_hist_kernel(const float* __restrict__ pos, const float* __restrict__ xyz,
                                                                   unsigned int* __restrict__ ws_dist, unsigned int* __restrict__ ws_hist,
                                                                   int N, int npos, int squared) {
  __shared__ unsigned int hist[KN_BINS];
  const int b = blockIdx.y, part = blockIdx.x, t = threadIdx.x, lane = t & 63;
  const float* qp = pos + (size_t)b * npos * 3;
  const float* pts = xyz + (size_t)b * N * 3;
  for (int i = t; i < KN_BINS; i += TK_THREADS) hist[i] = 0;
  __syncthreads();
  const int per = (N + KN_PARTS - 1) / KN_PARTS;
  const int n0 = part * per, n1 = min(N, n0 + per);
  for (int base = n0; base < n1; base += TK_THREADS) {          // wave-uniform trip count
    const int n = base + t;
    const bool ok = n < n1;
    unsigned int v = 0;
    if (ok) {
      v = __float_as_uint(nn_dist(qp, npos, squared, pts + (size_t)n * 3));
      // a distance is >= +0 unless a NaN / inf point put a sign-bit NaN here: canonicalise so that v >> KN_SHIFT stays
      // below KN_BINS (LDS histogram bound) and such points sort last
      if (v & 0x80000000u) v = 0x7FC00000u;
      ws_dist[(size_t)b * N + n] = v;
    }
    hist_add_wave(hist, v >> KN_SHIFT, ok, lane);
  }
  __syncthreads();
  unsigned int* out = ws_hist + ((size_t)b * KN_PARTS + part) * KN_BINS;
  for (int i = t; i < KN_BINS; i += TK_THREADS) out[i] = hist[i];
}

// wave 0: the bin (of 256) in which the running count reaches krem; res = {bin, count below the bin, count in the bin}
__device__ __forceinline__ void select_bin256(const unsigned int* hist, unsigned int krem, unsigned int* res) {
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  const unsigned int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
  const unsigned int mine = h0 + h1 + h2 + h3;
  unsigned int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned int up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  const unsigned int excl = incl - mine;
  const unsigned int total = __shfl(incl, 63, 64);
  const bool own = (excl < krem && krem <= incl) || (lane == 63 && krem > total);      // krem > total cannot happen for k <= N
  if (own) {
    unsigned int cum = excl;
    int bin = 4 * lane;
    const unsigned int hs[4] = {h0, h1, h2, h3};
    int j = 0;
    for (; j < 3; ++j) {
      if (cum + hs[j] >= krem) break;
      cum += hs[j];
    }
    res[0] = (unsigned int)(bin + j);
    res[1] = cum;
    res[2] = hs[j];
  }
}

// 64-bit key exchange inside the wave
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
  const unsigned int lo = __shfl_xor((unsigned int)v, mask, 64), hi = __shfl_xor((unsigned int)(v >> 32), mask, 64);
  return ((unsigned long long)hi << 32) | lo;
}

// bitonic sort of keys[0, kpad) (kpad a power of two >= 4096 uses the register scheme; smaller arrays the plain LDS network)
__device__ __forceinline__ void wg_bitonic_sort(unsigned long long* keys, int kpad) {
  const int t = threadIdx.x;
  if (kpad < 4 * TK_THREADS) {
    for (int size = 2; size <= kpad; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int i = t; i < (kpad >> 1); i += TK_THREADS) {
          const int lo = (i / stride) * (stride << 1) + (i % stride);
          const int hi = lo + stride;
          const bool up = ((lo & size) == 0);
          const unsigned long long a = keys[lo], c = keys[hi];
          if ((a > c) == up) { keys[lo] = c; keys[hi] = a; }
        }
        __syncthreads();
      }
    }
    return;
  }
  // thread t owns keys [g * 4096 + 4 t, + 4) of every 4096-key group g; all threads walk the same (size, stride) sequence
  const int ngroup = kpad / (4 * TK_THREADS);
  for (int grp = 0; grp < ngroup; ++grp) {
    unsigned long long v[4];
    const int base = grp * 4 * TK_THREADS + 4 * t;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = keys[base + j];
    for (int size = 2; size <= 4 * TK_THREADS; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        if (stride >= 256) {
          // partner in another wave: through LDS
#pragma unroll
          for (int j = 0; j < 4; ++j) keys[base + j] = v[j];
          __syncthreads();
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = base + j;
            const unsigned long long o = keys[i ^ stride];
            const bool up = ((i & size) == 0), lower = ((i & stride) == 0);
            const unsigned long long mn = v[j] < o ? v[j] : o, mx = v[j] < o ? o : v[j];
            v[j] = (lower == up) ? mn : mx;
          }
          __syncthreads();
        } else if (stride >= 4) {
          const int lm = stride >> 2;             // partner lane = lane ^ (stride / 4), same register slot
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int i = base + j;
            const unsigned long long o = shfl_xor_u64(v[j], lm);
            const bool up = ((i & size) == 0), lower = ((i & stride) == 0);
            const unsigned long long mn = v[j] < o ? v[j] : o, mx = v[j] < o ? o : v[j];
            v[j] = (lower == up) ? mn : mx;
          }
        } else {
          // stride 1 or 2: both keys in this thread's registers
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int pj = j ^ stride;
            if (pj > j) {
              const bool up = (((base + j) & size) == 0);
              const unsigned long long a = v[j], c = v[pj];
              if ((a > c) == up) { v[j] = c; v[pj] = a; }
            }
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) keys[base + j] = v[j];
  }
  __syncthreads();
  // merge the sorted 4096-key groups (kpad > 4096 only): remaining bitonic stages over the whole array through LDS
  if (ngroup > 1) {
    // groups were sorted ascending / descending alternately by the (i & size) rule with size = 4096 on absolute indices
    for (int size = 8 * TK_THREADS; size <= kpad; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int i = t; i < (kpad >> 1); i += TK_THREADS) {
          const int lo = (i / stride) * (stride << 1) + (i % stride);
          const int hi = lo + stride;
          const bool up = ((lo & size) == 0);
          const unsigned long long a = keys[lo], c = keys[hi];
          if ((a > c) == up) { keys[lo] = c; keys[hi] = a; }
        }
        __syncthreads();
      }
    }
  }
}

__global__ __launch_bounds__(TK_THREADS) void knn_select_sort_kernel(const unsigned int* __restrict__ ws_dist,
                                                                     const unsigned int* __restrict__ ws_hist,
                                                                     long long* __restrict__ idx_out, float* __restrict__ dist_out,
                                                                     int N, int k, int kpad, int cand_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];        // [kpad] then cand [cand_cap]
  __shared__ unsigned int hist[KN_BINS];
  __shared__ unsigned int wsum[16];
  __shared__ unsigned int sel[3];
  __shared__ unsigned int sh_count, sh_ccount, sh_kbin, sh_cbelow, sh_cbin;
  unsigned long long* cand = keys + kpad;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned int* dw = ws_dist + (size_t)b * N;
  // ---- part histograms -> hist; the bin holding the k-th smallest element
  for (int i = t; i < KN_BINS; i += TK_THREADS) {
    unsigned int s = 0;
#pragma unroll
    for (int p = 0; p < KN_PARTS; ++p) s += ws_hist[((size_t)b * KN_PARTS + p) * KN_BINS + i];
    hist[i] = s;
  }
  if (t == 0) { sh_count = 0; sh_ccount = 0; }
  __syncthreads();
  {
    const unsigned int h0 = hist[2 * t], h1 = hist[2 * t + 1];
    const unsigned int mine = h0 + h1;
    unsigned int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned int off = 0;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    const unsigned int excl = off + incl - mine;
    if (excl < (unsigned int)k && (unsigned int)k <= excl + mine) {           // exactly one thread
      if (excl + h0 >= (unsigned int)k) { sh_kbin = 2 * t; sh_cbelow = excl; sh_cbin = h0; }
      else { sh_kbin = 2 * t + 1; sh_cbelow = excl + h0; sh_cbin = h1; }
    }
  }
  for (int i = t; i < kpad; i += TK_THREADS) keys[i] = 0xFFFFFFFFFFFFFFFFull;
  __syncthreads();
  const unsigned int kbin = sh_kbin, need_total = (unsigned int)k - sh_cbelow, cbin = sh_cbin;
  const bool in_lds = cbin <= (unsigned int)cand_cap;
  // ---- one pass over the distances: lower bins -> keys, the k-th bin -> candidates
  for (int base = 0; base < N; base += TK_THREADS) {
    const int n = base + t;
    const unsigned int v = n < N ? dw[n] : 0xFFFFFFFFu;
    const unsigned int bin = v >> KN_SHIFT;
    const bool take = n < N && bin < kbin, cnd = in_lds && n < N && bin == kbin;
    const unsigned long long mt = __ballot(take), mc = __ballot(cnd);
    if (mt != 0ull) {
      const int leader = __ffsll((long long)mt) - 1;
      unsigned int bs = 0;
      if (lane == leader) bs = atomicAdd(&sh_count, (unsigned int)__popcll(mt));
      bs = __builtin_amdgcn_readlane(bs, leader);
      if (take) keys[bs + (unsigned int)__popcll(mt & ((1ull << lane) - 1ull))] = ((unsigned long long)v << 32) | (unsigned int)n;
    }
    if (mc != 0ull) {
      const int leader = __ffsll((long long)mc) - 1;
      unsigned int bs = 0;
      if (lane == leader) bs = atomicAdd(&sh_ccount, (unsigned int)__popcll(mc));
      bs = __builtin_amdgcn_readlane(bs, leader);
      if (cnd) cand[bs + (unsigned int)__popcll(mc & ((1ull << lane) - 1ull))] = ((unsigned long long)v << 32) | (unsigned int)n;
    }
  }
  __syncthreads();
  // ---- the `need_total` smallest (value, index) pairs of the k-th bin.  Items: the candidate list, or (overflow) ws_dist
  // filtered by the bin.  Radix passes over the low 20 value bits (8 + 8 + 4), then 3 passes over the index among the ties.
  const int nitems = in_lds ? (int)cbin : N;
  auto item = [&](int i, unsigned int& v, unsigned int& n) -> bool {
    if (in_lds) { const unsigned long long c = cand[i]; v = (unsigned int)(c >> 32); n = (unsigned int)c; return true; }
    v = dw[i]; n = (unsigned int)i;
    return (v >> KN_SHIFT) == kbin;
  };
  unsigned int prefix = kbin << KN_SHIFT, krem = need_total, neq = cbin;
  const int shifts[3] = {12, 4, 0};
  const unsigned int widths[3] = {8, 8, 4};
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = shifts[pass];
    const unsigned int himask = 0xFFFFFFFFu << (shift + widths[pass]);
    for (int i = t; i < 256; i += TK_THREADS) hist[i] = 0;
    __syncthreads();
    for (int base = 0; base < nitems; base += TK_THREADS) {
      const int i = base + t;
      unsigned int v = 0, n = 0;
      const bool ok = i < nitems && item(i, v, n) && ((v & himask) == prefix);
      hist_add_wave(hist, (v >> shift) & ((1u << widths[pass]) - 1u), ok, lane);
    }
    __syncthreads();
    select_bin256(hist, krem, sel);
    __syncthreads();
    prefix |= sel[0] << shift;
    krem -= sel[1];
    neq = sel[2];
    __syncthreads();
  }
  const unsigned int T = prefix;            // k-th smallest value
  unsigned int idx_thr = 0xFFFFFFFFu;       // elements == T with index <= idx_thr are taken
  if (neq != krem) {
    unsigned int ipre = 0;
    for (int pass = 0; pass < 3; ++pass) {
      const int shift = 16 - 8 * pass;
      const unsigned int himask = (pass == 0) ? 0u : (0xFFFFFFFFu << (shift + 8));
      for (int i = t; i < 256; i += TK_THREADS) hist[i] = 0;
      __syncthreads();
      for (int base = 0; base < nitems; base += TK_THREADS) {
        const int i = base + t;
        unsigned int v = 0, n = 0;
        const bool ok = i < nitems && item(i, v, n) && v == T && ((n & himask) == ipre);
        hist_add_wave(hist, (n >> shift) & 255u, ok, lane);
      }
      __syncthreads();
      select_bin256(hist, krem, sel);
      __syncthreads();
      ipre |= sel[0] << shift;
      krem -= sel[1];
      __syncthreads();
    }
    idx_thr = ipre;
  }
  for (int base = 0; base < nitems; base += TK_THREADS) {
    const int i = base + t;
    unsigned int v = 0, n = 0;
    const bool take = i < nitems && item(i, v, n) && (v < T || (v == T && n <= idx_thr));
    const unsigned long long mt = __ballot(take);
    if (mt != 0ull) {
      const int leader = __ffsll((long long)mt) - 1;
      unsigned int bs = 0;
      if (lane == leader) bs = atomicAdd(&sh_count, (unsigned int)__popcll(mt));
      bs = __builtin_amdgcn_readlane(bs, leader);
      const unsigned int slot = bs + (unsigned int)__popcll(mt & ((1ull << lane) - 1ull));
      if (take && slot < (unsigned int)kpad) keys[slot] = ((unsigned long long)v << 32) | n;
    }
  }
  __syncthreads();
  wg_bitonic_sort(keys, kpad);
  for (int i = t; i < k; i += TK_THREADS) {
    const unsigned long long kk = keys[i];
    idx_out[(size_t)b * k + i] = (long long)(kk & 0xFFFFFFFFull);
    if (dist_out) dist_out[(size_t)b * k + i] = __uint_as_float((unsigned int)(kk >> 32));
  }
}

// ctx[b][s][:] = s < k ? feat[b][idx ? idx[b][s] : s][:] : extra[b][s-k][:]   (rows of E floats, E % 4 == 0)
__global__ __launch_bounds__(256) void build_context_kernel(
    const float* __restrict__ feat, const long long* __restrict__ idx, const float* __restrict__ extra,
    float* __restrict__ ctx, int B, int Npts, int k, int X, int E4) {
  const int S = k + X;
  const size_t total = (size_t)B * S * E4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % E4);
    const size_t row = i / E4;
    const int s = (int)(row % S), b = (int)(row / S);
    float4 v;
    if (s < k) {
      const long long src = idx ? idx[(size_t)b * k + s] : (long long)s;
      v = reinterpret_cast<const float4*>(feat)[((size_t)b * Npts + src) * E4 + e];
    } else {
      v = reinterpret_cast<const float4*>(extra)[((size_t)b * X + (s - k)) * E4 + e];
    }
    reinterpret_cast<float4*>(ctx)[i] = v;
  }
}

// bf16 token rows (the FPN's channels-last bf16 output read in place: [B][Npts][4 * E4] bf16) -> fp32 context rows
// `bias` (fp32 [4 * E4] or null) is added to the gathered rows: the bias of the FPN's 3x3 output convolution, applied to the
// few rows a level reads instead of to the whole map (its gradient = the column sums of d(ctx), a3d_colsum_rows)
__global__ __launch_bounds__(256) void build_context_bf16_kernel(
    const unsigned short* __restrict__ feat, const long long* __restrict__ idx, const float* __restrict__ extra,
    const float* __restrict__ bias, float* __restrict__ ctx, int B, int Npts, int k, int X, int E4, int F4) {
  const int S = k + X;
  const size_t total = (size_t)B * S * E4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % E4);
    const size_t row = i / E4;
    const int s = (int)(row % S), b = (int)(row / S);
    float4 v;
    if (s < k) {
      const long long src = idx ? idx[(size_t)b * k + s] : (long long)s;
      const s16x4 h = reinterpret_cast<const s16x4*>(feat)[((size_t)b * Npts + src) * F4 + e];
      v = make_float4(bf2f((unsigned short)h[0]), bf2f((unsigned short)h[1]), bf2f((unsigned short)h[2]), bf2f((unsigned short)h[3]));
      if (bias) { v.x += bias[4 * e]; v.y += bias[4 * e + 1]; v.z += bias[4 * e + 2]; v.w += bias[4 * e + 3]; }
    } else {
      v = reinterpret_cast<const float4*>(extra)[((size_t)b * X + (s - k)) * E4 + e];
    }
    reinterpret_cast<float4*>(ctx)[i] = v;
  }
}

// d(ctx) rows -> the bf16 gradient map of the token tensor (zero-initialised by the caller, shared by every level that
// gathered from the map: accumulate = read-add-store, indices unique within one call); extra rows -> dextra (fp32)
__global__ __launch_bounds__(256) void build_context_bwd_bf16_kernel(
    const float* __restrict__ dctx, const long long* __restrict__ idx, unsigned short* __restrict__ dfeat,
    float* __restrict__ dextra, int B, int Npts, int k, int X, int E4, int F4, int accumulate) {
  const int S = k + X;
  const size_t total = (size_t)B * S * E4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % E4);
    const size_t row = i / E4;
    const int s = (int)(row % S), b = (int)(row / S);
    float4 v = reinterpret_cast<const float4*>(dctx)[i];
    if (s < k) {
      if (!dfeat) continue;
      const long long dst = idx ? idx[(size_t)b * k + s] : (long long)s;
      s16x4* p = reinterpret_cast<s16x4*>(dfeat) + ((size_t)b * Npts + dst) * F4 + e;
      if (accumulate) {
        const s16x4 o = *p;
        v.x += bf2f((unsigned short)o[0]); v.y += bf2f((unsigned short)o[1]);
        v.z += bf2f((unsigned short)o[2]); v.w += bf2f((unsigned short)o[3]);
      }
      s16x4 r;
      r[0] = (short)f2bf(v.x); r[1] = (short)f2bf(v.y); r[2] = (short)f2bf(v.z); r[3] = (short)f2bf(v.w);
      *p = r;
    } else if (dextra) {
      reinterpret_cast<float4*>(dextra)[((size_t)b * X + (s - k)) * E4 + e] = v;
    }
  }
}

// scalar-width variant for xyz rows (W floats per row)
__global__ __launch_bounds__(256) void build_context_rows_kernel(
    const float* __restrict__ feat, const long long* __restrict__ idx, const float* __restrict__ extra,
    float* __restrict__ ctx, int B, int Npts, int k, int X, int W) {
  const int S = k + X;
  const size_t total = (size_t)B * S * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % W);
    const size_t row = i / W;
    const int s = (int)(row % S), b = (int)(row / S);
    float v;
    if (s < k) {
      const long long src = idx ? idx[(size_t)b * k + s] : (long long)s;
      v = feat[((size_t)b * Npts + src) * W + e];
    } else {
      v = extra[((size_t)b * X + (s - k)) * W + e];
    }
    ctx[i] = v;
  }
}

// dfeat[b][idx[b][s]][:] (+)= dctx[b][s][:]  (indices unique per sample); dextra[b][x][:] = dctx[b][k+x][:]
__global__ __launch_bounds__(256) void build_context_bwd_kernel(
    const float* __restrict__ dctx, const long long* __restrict__ idx, float* __restrict__ dfeat,
    float* __restrict__ dextra, int B, int Npts, int k, int X, int E4, int accumulate) {
  const int S = k + X;
  const size_t total = (size_t)B * S * E4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = (int)(i % E4);
    const size_t row = i / E4;
    const int s = (int)(row % S), b = (int)(row / S);
    const float4 v = reinterpret_cast<const float4*>(dctx)[i];
    if (s < k) {
      if (!dfeat) continue;
      const long long dst = idx ? idx[(size_t)b * k + s] : (long long)s;
      float4* p = reinterpret_cast<float4*>(dfeat) + ((size_t)b * Npts + dst) * E4 + e;
      if (accumulate) {
        float4 o = *p;
        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
        *p = o;
      } else {
        *p = v;
      }
    } else if (dextra) {
      reinterpret_cast<float4*>(dextra)[((size_t)b * X + (s - k)) * E4 + e] = v;
    }
  }
}

}  // namespace a3d

using namespace a3d;

extern "C" int a3d_pcd_downsample(const float* pcd, float* out_xyz, int B, int C, int Hin, int Win, int factor,
                                  void* stream) {
  if (!pcd || !out_xyz || B <= 0 || C <= 0 || factor < 2 || (factor & 1) || Hin % factor || Win % factor) {
    set_error("a3d_pcd_downsample: bad argument (B=%d C=%d H=%d W=%d f=%d)", B, C, Hin, Win, factor);
    return A3D_ERR_ARG;
  }
  const size_t total = (size_t)B * C * (Hin / factor) * (Win / factor);
  const int grid = (int)std::min<size_t>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(pcd_downsample_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pcd, out_xyz, B * C, C,
                     Hin, Win, factor);
  return check_launch("a3d_pcd_downsample");
}

extern "C" size_t a3d_knn_topk_ws_bytes(int B, int N) {
  // distance bit patterns [B][N] + the part histograms of the two-launch selection [B][KN_PARTS][KN_BINS]
  return ((size_t)B * N + (size_t)B * KN_PARTS * KN_BINS) * sizeof(unsigned int);
}

static int nn_topk_launch(const char* fn, const float* pos, int npos, int squared, const float* xyz, void* ws,
                          long long* idx_out, float* dist_out, int B, int N, int k, void* stream) {
  if (!pos || !xyz || !ws || !idx_out || B <= 0 || N <= 0 || k <= 0 || k > N || k > 16384 || npos <= 0) {
    set_error("%s: bad argument (B=%d N=%d k=%d npos=%d; need 0 < k <= min(N, 16384))", fn, B, N, k, npos);
    return A3D_ERR_ARG;
  }
  int kpad = 2;
  while (kpad < k) kpad <<= 1;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)knn_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    (void)hipFuncSetAttribute((const void*)knn_select_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
    attr_set = true;
  }
  unsigned int* ws_dist = (unsigned int*)ws;
  static const bool one_wg = getenv("A3D_KNN_WS") && atoi(getenv("A3D_KNN_WS")) != 0;   // A/B switch: one workgroup per sample
  // candidate list of the k-th histogram bin: what is left of 136 KB of dynamic LDS after the key array, at most 6144 entries
  const long long cand_room = (136 * 1024 - (long long)kpad * 8) / 8;
  if (one_wg || cand_room < 1024) {
    hipLaunchKernelGGL(knn_topk_kernel, dim3(B), dim3(TK_THREADS), (size_t)kpad * sizeof(unsigned long long), (hipStream_t)stream, pos, xyz,
                       ws_dist, idx_out, dist_out, N, k, kpad, npos, squared);
    return check_launch(fn);
  }
  const int cand_cap = (int)std::min<long long>(cand_room, 6144);
  unsigned int* ws_hist = ws_dist + (size_t)B * N;
  hipLaunchKernelGGL(knn_dist_hist_kernel, dim3(KN_PARTS, B), dim3(TK_THREADS), 0, (hipStream_t)stream, pos, xyz, ws_dist, ws_hist, N, npos,
                     squared);
  int rc = check_launch(fn);
  if (rc) return rc;
  hipLaunchKernelGGL(knn_select_sort_kernel, dim3(B), dim3(TK_THREADS), (size_t)(kpad + cand_cap) * sizeof(unsigned long long),
                     (hipStream_t)stream, ws_dist, ws_hist, idx_out, dist_out, N, k, kpad, cand_cap);
  return check_launch(fn);
}

extern "C" int a3d_knn_topk(const float* pos, const float* xyz, void* ws, long long* idx_out, float* dist_out,
                            int B, int N, int k, void* stream) {
  return nn_topk_launch("a3d_knn_topk", pos, 1, 0, xyz, ws, idx_out, dist_out, B, N, k, stream);
}

extern "C" int a3d_traj_nn_topk(const float* traj_xyz, int L, const float* xyz, void* ws, long long* idx_out, float* dist_out,
                                int B, int N, int k, void* stream) {
  return nn_topk_launch("a3d_traj_nn_topk", traj_xyz, L, 1, xyz, ws, idx_out, dist_out, B, N, k, stream);
}

extern "C" int a3d_build_context(const float* feat, const long long* idx, const float* extra, float* ctx, int B,
                                 int Npts, int k, int X, int W, void* stream) {
  if (!feat || !ctx || B <= 0 || k <= 0 || X < 0 || W <= 0 || (X > 0 && !extra) || (!idx && k != Npts)) {
    set_error("a3d_build_context: bad argument (B=%d Npts=%d k=%d X=%d W=%d)", B, Npts, k, X, W);
    return A3D_ERR_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (W % 4 == 0) && ((((uintptr_t)feat | (uintptr_t)ctx | (uintptr_t)extra) & 15) == 0);
  if (vec) {
    const size_t total = (size_t)B * (k + X) * (W / 4);
    const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(build_context_kernel, dim3(grid), dim3(256), 0, s, feat, idx, extra, ctx, B, Npts, k, X, W / 4);
  } else {
    const size_t total = (size_t)B * (k + X) * W;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(build_context_rows_kernel, dim3(grid), dim3(256), 0, s, feat, idx, extra, ctx, B, Npts, k, X, W);
  }
  return check_launch("a3d_build_context");
}

extern "C" int a3d_build_context_bwd(const float* dctx, const long long* idx, float* dfeat, float* dextra, int B,
                                     int Npts, int k, int X, int W, int accumulate, void* stream) {
  if (!dctx || B <= 0 || k <= 0 || X < 0 || W <= 0 || (W % 4) != 0 || (!idx && k != Npts)) {
    set_error("a3d_build_context_bwd: bad argument (B=%d Npts=%d k=%d X=%d W=%d)", B, Npts, k, X, W);
    return A3D_ERR_ARG;
  }
  const size_t total = (size_t)B * (k + X) * (W / 4);
  const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(build_context_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dctx, idx, dfeat,
                     dextra, B, Npts, k, X, W / 4, accumulate);
  return check_launch("a3d_build_context_bwd");
}

extern "C" int a3d_build_context_bf16(const void* feat, int ldf, const long long* idx, const float* extra, const float* bias,
                                      float* ctx, int B, int Npts, int k, int X, int W, void* stream) {
  if (!feat || !ctx || B <= 0 || k <= 0 || X < 0 || W <= 0 || (W % 4) != 0 || ldf < W || (ldf % 4) != 0 || (X > 0 && !extra) || (!idx && k != Npts) ||
      ((((uintptr_t)feat) & 7) != 0) || ((((uintptr_t)ctx | (uintptr_t)extra) & 15) != 0)) {
    set_error("a3d_build_context_bf16: bad argument (B=%d Npts=%d k=%d X=%d W=%d; W %% 4 == 0, 8 / 16-byte aligned)", B, Npts, k, X, W);
    return A3D_ERR_ARG;
  }
  const size_t total = (size_t)B * (k + X) * (W / 4);
  const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(build_context_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)feat, idx,
                     extra, bias, ctx, B, Npts, k, X, W / 4, ldf / 4);
  return check_launch("a3d_build_context_bf16");
}

extern "C" int a3d_build_context_bwd_bf16(const float* dctx, const long long* idx, void* dfeat, int ldf, float* dextra, int B,
                                          int Npts, int k, int X, int W, int accumulate, void* stream) {
  if (!dctx || B <= 0 || k <= 0 || X < 0 || W <= 0 || (W % 4) != 0 || ldf < W || (ldf % 4) != 0 || (!idx && k != Npts) || ((((uintptr_t)dfeat) & 7) != 0)) {
    set_error("a3d_build_context_bwd_bf16: bad argument (B=%d Npts=%d k=%d X=%d W=%d)", B, Npts, k, X, W);
    return A3D_ERR_ARG;
  }
  const size_t total = (size_t)B * (k + X) * (W / 4);
  const int grid = (int)std::min<size_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(build_context_bwd_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dctx, idx,
                     (unsigned short*)dfeat, dextra, B, Npts, k, X, W / 4, ldf / 4, accumulate);
  return check_launch("a3d_build_context_bwd_bf16");
}
