// Data plane: the training-time image augmentation of datasets/utils.py:40-100 (`Resize`), on the device.
//   reference (per episode chunk, in a DataLoader worker, on (T*N, C, H, W) fp32 tensors, RGB and XYZ with shared draws):
//     nearest resize to (int(H*sc), int(W*sc))  ->  reflect-pad right/bottom back to >= (H, W)  ->  random crop (H, W)
//   here: ONE gather pass over the collated batch after the host->device copy.  The three steps compose into an index map
//     (y, x) -> (y + i, x + j) -> reflect at the resized extent -> nearest source pixel,
//   evaluated per output pixel from four integers per frame (rh, rw, i, j); the RGB rescale [-1, 1] -> [0, 1]
//   (dataset_engine.py:134-137) rides along as an affine on the values.  Nearest source index as ATen's `nearest` mode:
//   min(floorf(dst * (float)in / out), in - 1)  (aten/src/ATen/native/UpSample.h, nearest_neighbor_compute_source_index).
#include "a3d_common.h"
#include "../../include/act3d_hip.h"

namespace a3d {

__global__ __launch_bounds__(256) void resize_crop_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          const int* __restrict__ params, int F, int planes, int H, int W,
                                                          float a, float b) {
  const size_t per_frame = (size_t)planes * H * W;
  const size_t total = (size_t)F * per_frame;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % W);
    const int y = (int)((idx / W) % H);
    const size_t plane = idx / ((size_t)W * H);                 // frame * planes + p
    const int f = (int)(plane / planes);
    const int rh = params[f * 4 + 0], rw = params[f * 4 + 1];
    int yy = y + params[f * 4 + 2], xx = x + params[f * 4 + 3];
    if (yy >= rh) yy = 2 * (rh - 1) - yy;                       // reflect padding (no edge repeat), bottom / right only
    if (xx >= rw) xx = 2 * (rw - 1) - xx;
    const float sy = (float)H / (float)rh, sx = (float)W / (float)rw;
    const int iy = min((int)floorf((float)yy * sy), H - 1);
    const int ix = min((int)floorf((float)xx * sx), W - 1);
    dst[idx] = src[(plane * H + iy) * W + ix] * a + b;
  }
}

}  // namespace a3d

using namespace a3d;

extern "C" int a3d_resize_crop(const float* src, float* dst, const int* params, int frames, int planes, int H, int W,
                               float scale, float shift, void* stream) {
  if (!src || !dst || !params || src == dst || frames <= 0 || planes <= 0 || H <= 0 || W <= 0) {
    set_error("a3d_resize_crop: bad argument (frames=%d planes=%d H=%d W=%d; in-place is not supported)", frames, planes, H, W);
    return A3D_ERR_ARG;
  }
  const size_t total = (size_t)frames * planes * H * W;
  const int grid = (int)std::min<size_t>((total + 255) / 256, 65536);
  hipLaunchKernelGGL(resize_crop_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, params, frames, planes, H, W, scale, shift);
  return check_launch("a3d_resize_crop");
}
