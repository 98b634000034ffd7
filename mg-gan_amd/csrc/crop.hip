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

// ---- a ragged batch into the static buffers of its shape bucket (train()'s padded batches) -------------------------
// Every batch tensor is (outer, pedestrians, inner) contiguous (in_xy (8, b, 2): outer = 8, inner = 2; the crops
// (b, 4, 33, 33): outer = 1, inner = 4,356).  The real pedestrians are copied in front; the phantom pedestrians behind them
// get their constant content (position tensors: x = slot within a phantom scene of `period` pedestrians, everything else 0).
// ONE launch instead of a copy plus a restore per tensor.
#define MG_PAD_MAX 8
struct PadTensor {
  const float* src;
  float* dst;
  long inner;
  int outer, position;
  long first;  // first work item of this tensor
};
struct PadBatch {
  PadTensor t[MG_PAD_MAX];
  int n, b, b_pad, period;
  long total;
};
__global__ __launch_bounds__(256) void pad_batch_kernel(PadBatch a) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.total; i += (long)gridDim.x * 256) {
    int k = 0;
#pragma unroll 1
    for (int q = 1; q < a.n; ++q)
      if (i >= a.t[q].first) k = q;
    const PadTensor& T = a.t[k];
    const long e = i - T.first, in = e % T.inner, rest = e / T.inner;
    const int ped = (int)(rest % a.b_pad), o = (int)(rest / a.b_pad);
    float v;
    if (ped < a.b) v = T.src[((long)o * a.b + ped) * T.inner + in];
    else v = (T.position && in == 0) ? (float)(ped % a.period) : 0.f;
    T.dst[e] = v;
  }
}

// A handful of small buffers copied in ONE launch (BatchNorm running statistics: 8-64 bytes each): snapshot / roll-back of
// module buffers around work that is issued speculatively (mggan/model/train.py: the next iteration's discriminator context).
#define MG_COPY_MAX 8
struct CopySmall {
  const unsigned char* src[MG_COPY_MAX];
  unsigned char* dst[MG_COPY_MAX];
  int bytes[MG_COPY_MAX];
  int n;
};
__global__ __launch_bounds__(256) void copy_small_kernel(CopySmall a) {
  for (int k = 0; k < a.n; ++k)
    for (int i = threadIdx.x; i < a.bytes[k]; i += 256) a.dst[k][i] = a.src[k][i];
}

extern "C" {

int mggan_copy_small(const void* descs, int n, hipStream_t stream) {
  struct Desc { const void* src; void* dst; long bytes; };
  MG_CHECK_ARG(descs && n > 0 && n <= MG_COPY_MAX, "copy_small: 1..%d buffers", MG_COPY_MAX);
  const Desc* d = (const Desc*)descs;
  CopySmall a;
  a.n = n;
  for (int i = 0; i < MG_COPY_MAX; ++i) {
    const bool on = i < n;
    MG_CHECK_ARG(!on || (d[i].src && d[i].dst && d[i].bytes > 0 && d[i].bytes <= 65536), "copy_small: bad buffer %d", i);
    a.src[i] = on ? (const unsigned char*)d[i].src : nullptr;
    a.dst[i] = on ? (unsigned char*)d[i].dst : nullptr;
    a.bytes[i] = on ? (int)d[i].bytes : 0;
  }
  MG_LAUNCH(copy_small_kernel, dim3(1), dim3(256), 0, stream, a);
  MG_LAUNCH_CHECK("copy_small");
  return MGGAN_OK;
}

int mggan_pad_batch(const void* descs, int n, int b, int b_pad, int period, hipStream_t stream) {
  struct Desc { const float* src; float* dst; long inner; int outer, position; };
  MG_CHECK_ARG(descs && n > 0 && n <= MG_PAD_MAX && b >= 0 && b <= b_pad && period > 0, "pad_batch: bad arguments");
  const Desc* d = (const Desc*)descs;
  PadBatch a;
  a.n = n; a.b = b; a.b_pad = b_pad; a.period = period; a.total = 0;
  for (int i = 0; i < n; ++i) {
    MG_CHECK_ARG(d[i].dst && (d[i].src || b == 0) && d[i].inner > 0 && d[i].outer > 0, "pad_batch: bad tensor %d", i);
    a.t[i].src = d[i].src; a.t[i].dst = d[i].dst; a.t[i].inner = d[i].inner; a.t[i].outer = d[i].outer;
    a.t[i].position = d[i].position; a.t[i].first = a.total;
    a.total += (long)d[i].outer * b_pad * d[i].inner;
  }
  if (a.total == 0) return MGGAN_OK;
  int blocks = cdiv(a.total, 256 * 4);
  if (blocks > 4096) blocks = 4096;
  MG_LAUNCH(pad_batch_kernel, dim3(blocks), dim3(256), 0, stream, a);
  MG_LAUNCH_CHECK("pad_batch");
  return MGGAN_OK;
}

int mggan_crop_patches(const unsigned char* atlas, const long long* img_off, const int* img_hw, const int* centers, int n,
                       int margin, float* out, hipStream_t stream) {
  MG_CHECK_ARG(n >= 0 && margin >= 0, "crop_patches: bad arguments");
  if (n == 0) return MGGAN_OK;
  MG_CHECK_ARG(atlas && img_off && img_hw && centers && out, "crop_patches: null pointer");
  const long long total = (long long)n * 4 * (2 * margin + 1) * (2 * margin + 1);
  int blocks = cdiv(total, 256);
  if (blocks > 8192) blocks = 8192;
  MG_LAUNCH(crop_patches_kernel, dim3(blocks), dim3(256), 0, stream, atlas, img_off, img_hw, centers, n, margin, out);
  MG_LAUNCH_CHECK("crop_patches");
  return MGGAN_OK;
}

}  // extern "C"
