// Per-pedestrian crops of the scene image, cut on the GPU from images kept in HBM.
// Replaces the host loop of the reference's loader (/root/reference/mggan/data_utils/BaseTrajectories.py:254-288
// ImageFeatures_small, called once per pedestrian from trajectories_scene.py:349-356): a (2m+1)x(2m+1) window of
// the "small" scene image around the last observed position, RGB mapped to -1 + v*2/256, plus a one-hot centre
// channel; pixels outside the image read 0 like PIL's crop.  HBM-bound byte work: one lane per output float,
// consecutive lanes walk consecutive x (coalesced 4-byte stores, 3-byte-strided u8 loads served by L1/L2).
#include "common.h"
#include "../../include/mggan_hip.h"

__global__ __launch_bounds__(256) void crop_patches_kernel(const unsigned char* __restrict__ atlas,
                                                           const long long* __restrict__ img_off,
                                                           const int* __restrict__ img_hw, const int* __restrict__ centers,
                                                           int n, int margin, float* __restrict__ out) {
  const int side = 2 * margin + 1, plane = side * side;
  const long long total = (long long)n * 4 * plane;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % side), y = (int)((i / side) % side), c = (int)((i / plane) % 4), p = (int)(i / (4 * plane));
    float v;
    if (c == 3) {
      v = (x == margin && y == margin) ? 1.f : 0.f;
    } else {
      const int H = img_hw[2 * p], W = img_hw[2 * p + 1];
      const int sx = centers[2 * p] - margin + x, sy = centers[2 * p + 1] - margin + y;
      const float px = (sx >= 0 && sx < W && sy >= 0 && sy < H) ? (float)atlas[img_off[p] + ((long long)sy * W + sx) * 3 + c] : 0.f;
      v = (float)(-1.0 + (double)px * 2.0 / 256.0);  // the reference computes this in float64 and rounds once
    }
    out[i] = v;
  }
}

extern "C" {

int mggan_crop_patches(const unsigned char* atlas, const long long* img_off, const int* img_hw, const int* centers, int n,
                       int margin, float* out, hipStream_t stream) {
  MG_CHECK_ARG(n >= 0 && margin >= 0, "crop_patches: bad arguments");
  if (n == 0) return MGGAN_OK;
  MG_CHECK_ARG(atlas && img_off && img_hw && centers && out, "crop_patches: null pointer");
  const long long total = (long long)n * 4 * (2 * margin + 1) * (2 * margin + 1);
  int blocks = cdiv(total, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(crop_patches_kernel, dim3(blocks), dim3(256), 0, stream, atlas, img_off, img_hw, centers, n, margin, out);
  MG_LAUNCH_CHECK("crop_patches");
  return MGGAN_OK;
}

}  // extern "C"
