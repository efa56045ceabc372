// Elementwise pieces of the ChainedDiffuser DDPM trajectory denoiser (all HBM/latency-bound; they exist so
// that one denoise step is a fixed sequence of stream-ordered launches with no host synchronisation and can be
// captured in a hipGraph).
//
//   a3d_ddpm_add_noise   x_t = sqrt(acp[t]) x0 + sqrt(1-acp[t]) eps, position / rotation channel groups with
//                        their own schedules                                   diffusion_model.py:296-305
//   a3d_ddpm_step        inpaint (out[mask] = cond[mask]) + DDPMScheduler.step with prediction_type="sample",
//                        clip_sample, fixed_small variance                      diffusion_model.py:107-117
//   a3d_adaln_{fwd,bwd}  x * (1 + scale) + shift                                layers.py:273-290
//   a3d_sinusoidal_emb   [sin(x f_j) | cos(x f_j)]                              position_encodings.py:7-20
//   a3d_silu_{fwd,bwd}   SiLU in AdaLN's modulation                             layers.py:276-278
//   a3d_add_rows         seq + per-position embedding (seq1_sem_pos)            layers.py:104-105,132-133
//   a3d_traj_update      cat(traj_xyz + delta, rot)                             diffusion_head.py:268-272
#include "a3d_common.h"
#include "../../include/act3d_hip.h"

namespace a3d {

__global__ void ddpm_add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                      const long long* __restrict__ t, const float* __restrict__ acp_pos,
                                      const float* __restrict__ acp_rot, float* __restrict__ out, int B, int L, int D,
                                      int npos) {
  const size_t total = (size_t)B * L * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % D);
    const int b = (int)(i / ((size_t)L * D));
    const float a = (c < npos) ? acp_pos[t[b]] : acp_rot[t[b]];
    out[i] = sqrtf(a) * x0[i] + sqrtf(1.0f - a) * noise[i];
  }
}

// coef tables: [T][3] = (coef_x0, coef_xt, sigma) per schedule
__global__ void ddpm_step_kernel(const float* __restrict__ model_out, const float* __restrict__ sample,
                                 const float* __restrict__ noise, const float* __restrict__ cond_data,
                                 const unsigned char* __restrict__ cond_mask, const float* __restrict__ coef_pos,
                                 const float* __restrict__ coef_rot, float* __restrict__ out, int rows, int D,
                                 int npos, int t) {
  const size_t total = (size_t)rows * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % D);
    float mo = model_out[i];
    if (cond_mask && cond_mask[i]) mo = cond_data[i];
    if (t == 0) { out[i] = mo; continue; }
    const float* cf = ((c < npos) ? coef_pos : coef_rot) + (size_t)t * 3;
    const float x0 = fminf(fmaxf(mo, -1.0f), 1.0f);
    float prev = cf[0] * x0 + cf[1] * sample[i];
    if (noise) prev += cf[2] * noise[i];
    out[i] = prev;
  }
}

__global__ void adaln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mod, float* __restrict__ y,
                                 int B, int L, int E) {
  const size_t total = (size_t)B * L * E;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % E);
    const int b = (int)(i / ((size_t)L * E));
    y[i] = x[i] * (1.0f + mod[(size_t)b * 2 * E + c]) + mod[(size_t)b * 2 * E + E + c];
  }
}
// one workgroup per sample; dmod[b] = [sum_l dy*x | sum_l dy]
__global__ __launch_bounds__(256) void adaln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mod,
                                                        const float* __restrict__ dy, float* __restrict__ dx,
                                                        float* __restrict__ dmod, int L, int E) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < E; c += blockDim.x) {
    const float sc = 1.0f + mod[(size_t)b * 2 * E + c];
    float ds = 0.f, dh = 0.f;
    for (int l = 0; l < L; ++l) {
      const size_t i = ((size_t)b * L + l) * E + c;
      const float g = dy[i];
      ds += g * x[i];
      dh += g;
      dx[i] = g * sc;
    }
    dmod[(size_t)b * 2 * E + c] = ds;
    dmod[(size_t)b * 2 * E + E + c] = dh;
  }
}

__global__ void sinusoidal_kernel(const float* __restrict__ x, float* __restrict__ out, int n, int E) {
  const int half = E / 2;
  const size_t total = (size_t)n * half;
  const float step = 9.210340371976184f / (float)(half - 1);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % half);
    const size_t r = i / half;
    const float f = expf((float)j * -step);
    const float a = x[r] * f;
    out[r * E + j] = sinf(a);
    out[r * E + half + j] = cosf(a);
  }
}

__global__ void silu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    y[i] = v / (1.0f + expf(-v));
  }
}
__global__ void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    const float s = 1.0f / (1.0f + expf(-v));
    dx[i] = dy[i] * s * (1.0f + v * (1.0f - s));
  }
}

__global__ void add_rows_kernel(const float* __restrict__ x, const float* __restrict__ r, float* __restrict__ y, int B,
                                int L, int E) {
  const size_t total = (size_t)B * L * E;
  const size_t le = (size_t)L * E;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    y[i] = x[i] + r[i % le];
}

__global__ void traj_update_kernel(const float* __restrict__ traj, const float* __restrict__ upd, float* __restrict__ out,
                                   int rows, int D, int npos) {
  const size_t total = (size_t)rows * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % D);
    out[i] = (c < npos) ? traj[i] + upd[i] : upd[i];
  }
}

}  // namespace a3d

using namespace a3d;
static inline int gsz(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 4096); }

extern "C" int a3d_ddpm_add_noise(const float* x0, const float* noise, const long long* t, const float* acp_pos,
                                  const float* acp_rot, float* out, int B, int L, int D, int npos, void* stream) {
  if (!x0 || !noise || !t || !acp_pos || !acp_rot || !out || B <= 0 || L <= 0 || D <= 0) { set_error("a3d_ddpm_add_noise: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(ddpm_add_noise_kernel, dim3(gsz((size_t)B * L * D)), dim3(256), 0, (hipStream_t)stream, x0, noise, t, acp_pos, acp_rot, out, B, L, D, npos);
  return check_launch("a3d_ddpm_add_noise");
}
extern "C" int a3d_ddpm_step(const float* model_out, const float* sample, const float* noise, const float* cond_data,
                             const unsigned char* cond_mask, const float* coef_pos, const float* coef_rot, float* out,
                             int rows, int D, int npos, int t, void* stream) {
  if (!model_out || !sample || !coef_pos || !coef_rot || !out || rows <= 0 || D <= 0 || t < 0 || (cond_mask && !cond_data)) { set_error("a3d_ddpm_step: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(ddpm_step_kernel, dim3(gsz((size_t)rows * D)), dim3(256), 0, (hipStream_t)stream, model_out, sample, noise, cond_data, cond_mask, coef_pos, coef_rot, out, rows, D, npos, t);
  return check_launch("a3d_ddpm_step");
}
extern "C" int a3d_adaln_fwd(const float* x, const float* mod, float* y, int B, int L, int E, void* stream) {
  if (!x || !mod || !y || B <= 0 || L <= 0 || E <= 0) { set_error("a3d_adaln_fwd: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(adaln_fwd_kernel, dim3(gsz((size_t)B * L * E)), dim3(256), 0, (hipStream_t)stream, x, mod, y, B, L, E);
  return check_launch("a3d_adaln_fwd");
}
extern "C" int a3d_adaln_bwd(const float* x, const float* mod, const float* dy, float* dx, float* dmod, int B, int L,
                             int E, void* stream) {
  if (!x || !mod || !dy || !dx || !dmod || B <= 0 || L <= 0 || E <= 0) { set_error("a3d_adaln_bwd: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(adaln_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, mod, dy, dx, dmod, L, E);
  return check_launch("a3d_adaln_bwd");
}
extern "C" int a3d_sinusoidal_emb(const float* x, float* out, int n, int E, void* stream) {
  if (!x || !out || n <= 0 || E < 4 || (E & 1)) { set_error("a3d_sinusoidal_emb: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(sinusoidal_kernel, dim3(gsz((size_t)n * E / 2)), dim3(256), 0, (hipStream_t)stream, x, out, n, E);
  return check_launch("a3d_sinusoidal_emb");
}
extern "C" int a3d_silu_fwd(const float* x, float* y, size_t n, void* stream) {
  if (!x || !y) { set_error("a3d_silu_fwd: null pointer"); return A3D_ERR_ARG; }
  if (!n) return A3D_OK;
  hipLaunchKernelGGL(silu_fwd_kernel, dim3(gsz(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  return check_launch("a3d_silu_fwd");
}
extern "C" int a3d_silu_bwd(const float* x, const float* dy, float* dx, size_t n, void* stream) {
  if (!x || !dy || !dx) { set_error("a3d_silu_bwd: null pointer"); return A3D_ERR_ARG; }
  if (!n) return A3D_OK;
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(gsz(n)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n);
  return check_launch("a3d_silu_bwd");
}
extern "C" int a3d_add_rows(const float* x, const float* r, float* y, int B, int L, int E, void* stream) {
  if (!x || !r || !y || B <= 0 || L <= 0 || E <= 0) { set_error("a3d_add_rows: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(add_rows_kernel, dim3(gsz((size_t)B * L * E)), dim3(256), 0, (hipStream_t)stream, x, r, y, B, L, E);
  return check_launch("a3d_add_rows");
}
extern "C" int a3d_traj_update(const float* traj, const float* upd, float* out, int rows, int D, int npos, void* stream) {
  if (!traj || !upd || !out || rows <= 0 || D <= 0) { set_error("a3d_traj_update: bad argument"); return A3D_ERR_ARG; }
  hipLaunchKernelGGL(traj_update_kernel, dim3(gsz((size_t)rows * D)), dim3(256), 0, (hipStream_t)stream, traj, upd, out, rows, D, npos);
  return check_launch("a3d_traj_update");
}
