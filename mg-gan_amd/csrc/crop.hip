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

// ---- crops of the AUGMENTED scene image (training: flip -> rotate(expand, nearest) -> Lanczos resize -> crop) ---------------
// The reference transforms the whole scene image per item with Pillow on the host and cuts one 33 x 33 window per pedestrian
// out of the result (/root/reference/mggan/data_utils/trajectories_scene.py:276-357): 7,890 pedestrians/s, 0.8 % of what the
// training step consumes.  Here the un-augmented `scaled_image` of every scene stays in HBM and a workgroup computes ONE
// pedestrian's window directly: only the taps its 33 x 33 output pixels need, with the integers Pillow itself would use
// (mggan/data_utils/aug_geometry.py: the 16.16 inverse affine map of Geometry.c's nearest-neighbour loop, the 22-bit Lanczos
// tables of Resample.c's 8-bit two-pass resize, its clip8 between and behind the passes) -- bit-identical crops.
//   per strip of output rows (as many as keep <= AUG_ROWS source rows in LDS):
//     horizontal pass: a wave stages one row of the rotated canvas (gathered through the affine map from the resident image,
//       flipped on the way) in LDS, lane (X, c) of 99 sums its <= 128 taps -> one u8 of the intermediate image;
//     vertical pass: thread (Y, X, c) sums its taps over the intermediate rows -> u8 -> -1 + v * 2 / 256 (f64, rounded once).
// Integer arithmetic throughout (int32 accumulators like Pillow's); HBM traffic is the gathered source window, read once per strip.
#define AUG_SIDE_MAX 33
#define AUG_KS_MAX 128     // taps per output pixel and pass: Lanczos support 3 x scale, scale <= 21
#define AUG_ROWS 320       // intermediate rows held per strip
#define AUG_SPAN_MAX 800   // source columns of a staged row: 32 x scale + taps
struct AugItem {           // one per batch item (scene instance); mggan/data_utils/device_crops.py packs it as 26 int32
  long long img_off;       // byte offset of the scene's scaled image (h, w, 3) u8 in the atlas
  int w, h, flip, rot;     // flip 0 / 1 (left-right) / 2 (top-bottom); rot 0: alpha == 0 (the canvas is the flipped image)
  int nw, nh, sw, sh;      // rotated canvas, resized ("small") image
  int a[6];                // 16.16 inverse affine map: source x = (a2 + y a1 + x a0) >> 16, source y = (a5 + y a4 + x a3) >> 16
  int ksh, ksv;            // taps per row of the horizontal / vertical table
  long long kh, bh, kv, bv;  // offsets (in int32) of the coefficient rows / (first index, taps) pairs in the table pool
};
__device__ __forceinline__ int aug_clip8(int v) {
  v >>= 22;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}
__global__ __launch_bounds__(256) void crop_patches_aug_kernel(const unsigned char* __restrict__ atlas,
                                                               const AugItem* __restrict__ items, const int* __restrict__ tables,
                                                               const int* __restrict__ ped_item, const int* __restrict__ centers,
                                                               int margin, float* __restrict__ out) {
  __shared__ int khs[AUG_SIDE_MAX * AUG_KS_MAX];
  __shared__ unsigned char tmp[AUG_ROWS * AUG_SIDE_MAX * 3 + 4];
  __shared__ unsigned char rowbuf[4][AUG_SPAN_MAX * 3 + 8];
  const int p = blockIdx.x, side = 2 * margin + 1, plane = side * side;
  const AugItem it = items[ped_item[p]];
  float* o = out + (size_t)p * 4 * plane;
  const int X0 = centers[2 * p] - margin, Y0 = centers[2 * p + 1] - margin;
  // the window's part inside the small image; everything else reads 0 like Image.crop
  const int Xa = max(X0, 0), Xb = min(X0 + side, it.sw), Ya = max(Y0, 0), Yb = min(Y0 + side, it.sh);
  for (int i = threadIdx.x; i < 4 * plane; i += 256) {
    const int c = i / plane, r = i % plane;
    o[i] = c == 3 ? ((r == margin * side + margin) ? 1.f : 0.f) : -1.f;
  }
  if (Xa >= Xb || Ya >= Yb) return;
  const int nX = Xb - Xa, nL = nX * 3;
  const int* kh = tables + it.kh;
  const int* bh = tables + it.bh;
  const int* kv = tables + it.kv;
  const int* bv = tables + it.bv;
  for (int i = threadIdx.x; i < nX * it.ksh; i += 256) khs[i] = kh[(size_t)(Xa + i / it.ksh) * it.ksh + i % it.ksh];
  const int xlo = bh[2 * Xa], xhi = bh[2 * (Xb - 1)] + bh[2 * (Xb - 1) + 1], span = xhi - xlo;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int Xi = lane / 3, ch = lane % 3;
  const int hx0 = lane < nL ? bh[2 * (Xa + Xi)] - xlo : 0, hn = lane < nL ? bh[2 * (Xa + Xi) + 1] : 0;
  const int Xi2 = (lane + 64) / 3, ch2 = (lane + 64) % 3;  // lanes 64..98 of the 99 (X, c) pairs: second pass of the wave
  const int hx1 = lane + 64 < nL ? bh[2 * (Xa + Xi2)] - xlo : 0, hn1 = lane + 64 < nL ? bh[2 * (Xa + Xi2) + 1] : 0;
  const unsigned char* img = atlas + it.img_off;
  __syncthreads();
  int Ys = Ya;
  while (Ys < Yb) {
    // strip [Ys, Ye): as many output rows as keep the source rows within AUG_ROWS
    const int ylo = bv[2 * Ys];
    int Ye = Ys + 1;
    while (Ye < Yb && bv[2 * Ye] + bv[2 * Ye + 1] - ylo <= AUG_ROWS) ++Ye;
    const int yhi = bv[2 * (Ye - 1)] + bv[2 * (Ye - 1) + 1];
    // ---- horizontal pass: source rows ylo .. yhi-1 of the rotated canvas -> tmp ----
    for (int y = ylo + wv; y < yhi; y += 4) {
      unsigned char* rb = rowbuf[wv];
      for (int j = lane; j < span; j += 64) {
        const int xr = xlo + j;
        int xin = xr, yin = y;
        if (it.rot) {
          xin = (int)(((long long)it.a[2] + (long long)y * it.a[1] + (long long)xr * it.a[0]) >> 16);
          yin = (int)(((long long)it.a[5] + (long long)y * it.a[4] + (long long)xr * it.a[3]) >> 16);
        }
        unsigned char r = 0, g = 0, b = 0;
        if (xin >= 0 && xin < it.w && yin >= 0 && yin < it.h) {
          const int xs = it.flip == 1 ? it.w - 1 - xin : xin, ys = it.flip == 2 ? it.h - 1 - yin : yin;
          const unsigned char* q = img + ((size_t)ys * it.w + xs) * 3;
          r = q[0]; g = q[1]; b = q[2];
        }
        rb[3 * j] = r; rb[3 * j + 1] = g; rb[3 * j + 2] = b;
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane < nL) {
        int sacc = 1 << 21;
        const int* k = khs + Xi * it.ksh;
        for (int t = 0; t < hn; ++t) sacc += (int)rb[3 * (hx0 + t) + ch] * k[t];
        tmp[(y - ylo) * (AUG_SIDE_MAX * 3) + lane] = (unsigned char)aug_clip8(sacc);
      }
      if (lane + 64 < nL) {
        int sacc = 1 << 21;
        const int* k = khs + Xi2 * it.ksh;
        for (int t = 0; t < hn1; ++t) sacc += (int)rb[3 * (hx1 + t) + ch2] * k[t];
        tmp[(y - ylo) * (AUG_SIDE_MAX * 3) + lane + 64] = (unsigned char)aug_clip8(sacc);
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- vertical pass: output rows Ys .. Ye-1 ----
    for (int i = threadIdx.x; i < (Ye - Ys) * nL; i += 256) {
      const int Y = Ys + i / nL, l = i % nL;
      const int y0 = bv[2 * Y] - ylo, n = bv[2 * Y + 1];
      const int* k = kv + (size_t)Y * it.ksv;
      int sacc = 1 << 21;
      for (int t = 0; t < n; ++t) sacc += (int)tmp[(y0 + t) * (AUG_SIDE_MAX * 3) + l] * k[t];
      const int v = aug_clip8(sacc);
      o[(l % 3) * plane + (Y - Y0) * side + (Xa + l / 3 - X0)] = (float)(-1.0 + (double)v * 2.0 / 256.0);
    }
    __syncthreads();
    Ys = Ye;
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

int mggan_crop_patches_aug(const unsigned char* atlas, const void* items, const int* tables, const int* ped_item,
                           const int* centers, int n, int margin, float* out, hipStream_t stream) {
  MG_CHECK_ARG(n >= 0 && margin >= 0 && 2 * margin + 1 <= AUG_SIDE_MAX, "crop_patches_aug: window of %d pixels (<= %d)", 2 * margin + 1,
               AUG_SIDE_MAX);
  if (n == 0) return MGGAN_OK;
  MG_CHECK_ARG(atlas && items && tables && ped_item && centers && out, "crop_patches_aug: null pointer");
  static_assert(sizeof(AugItem) == 26 * 4, "AugItem is 26 int32 words");
  MG_LAUNCH(crop_patches_aug_kernel, dim3(n), dim3(256), 0, stream, atlas, (const AugItem*)items, tables, ped_item, centers, margin,
            out);
  MG_LAUNCH_CHECK("crop_patches_aug");
  return MGGAN_OK;
}

}  // extern "C"
