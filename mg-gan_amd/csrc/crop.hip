// Per-pedestrian crops of the scene image, cut on the GPU from images kept in HBM.
// Replaces the host loop of the reference's loader (/root/reference/mggan/data_utils/BaseTrajectories.py:254-288
// ImageFeatures_small, called once per pedestrian from trajectories_scene.py:349-356): a (2m+1)x(2m+1) window of
// the "small" scene image around the last observed position, RGB mapped to -1 + v*2/256, plus a one-hot centre
// channel; pixels outside the image read 0 like PIL's crop.  HBM-bound byte work: one lane per output float,
// consecutive lanes walk consecutive x (coalesced 4-byte stores, 3-byte-strided u8 loads served by L1/L2).
#include <cstring>
#include <cstdlib>
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
#define AUG_L (AUG_SIDE_MAX * 3)  // (X, channel) pairs of a window row
#define AUG_KS_MAX 128     // taps per output pixel and pass: Lanczos support 3 x scale, scale <= 21
#define AUG_ROWS 192       // intermediate rows held per strip
#define AUG_TPAD (AUG_ROWS + AUG_KS_MAX + 12)  // bytes per (X, channel) line of the intermediate image (+ slack: a thread reads whole
                                              // words); 83 words: neighbouring lines start on different LDS banks
#define AUG_SPAN_MAX 800   // source columns of a staged row: 32 x scale + taps
#define AUG_STAGE 9600     // bytes of staged canvas rows per colour plane ...
#define AUG_PLANE (AUG_STAGE + 44)  // ... planes 2,411 words apart: the three channels of a column sit on different banks
#define AUG_BIAS (1 << 22) // coefficients are stored biased (>= 0, < 2^24) as three byte limbs for v_dot4_u32_u8
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
// sum_t byte_t(px) * k_t over four taps, k biased and split into byte limbs: three v_dot4_u32_u8 (full rate; a 32-bit integer
// multiply is quarter rate and would need a bit-field extract per tap) + one for the bias.  All sums are exact modulo 2^32 and
// the true value fits an int32 (Pillow accumulates in int), so the wrap-around cancels.
struct AugAcc {
  unsigned lo, mid, hi, sum;
  __device__ __forceinline__ void clear() { lo = mid = hi = sum = 0u; }
  __device__ __forceinline__ void taps(unsigned px, unsigned k0, unsigned k1, unsigned k2) {
    lo = __builtin_amdgcn_udot4(px, k0, lo, false);
    mid = __builtin_amdgcn_udot4(px, k1, mid, false);
    hi = __builtin_amdgcn_udot4(px, k2, hi, false);
    sum = __builtin_amdgcn_udot4(px, 0x01010101u, sum, false);
  }
  __device__ __forceinline__ int value() const { return (int)((1u << 21) + lo + (mid << 8) + (hi << 16) - (sum << 22)); }
};
__device__ __forceinline__ void aug_pack_limbs(const int k[4], unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = p1 = p2 = 0u;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const unsigned kb = (unsigned)(k[u] + AUG_BIAS);
    p0 |= (kb & 0xffu) << (8 * u);
    p1 |= ((kb >> 8) & 0xffu) << (8 * u);
    p2 |= ((kb >> 16) & 0xffu) << (8 * u);
  }
}
// KS: the taps of a horizontal coefficient row, rounded up to a multiple of 32 -- they live in REGISTERS (3 KS / 4 packed limb
// words) of the thread that owns output column X and channel c for the whole launch; taps beyond an item's own count are zero
// (stored as the bias: they cancel).
template <int KS>
__global__ __launch_bounds__(256) void crop_patches_aug_kernel(const unsigned char* __restrict__ atlas,
                                                               const AugItem* __restrict__ items, const int* __restrict__ tables,
                                                               const int* __restrict__ ped_item, const int* __restrict__ centers,
                                                               int margin, float* __restrict__ out,
                                                               const long long* __restrict__ small_off,
                                                               unsigned char* __restrict__ small, int plane_b, int tpad,
                                                               int rows_cap, const int* __restrict__ bands) {
  // TILE MODE (small != NULL): the workgroup computes one 33 x 33 TILE of an item's whole resized image -- ped_item[p] is the
  // tile's item, centers[2p..] its centre -- as u8 RGB into small + small_off[item]; the crops are then windows of those
  // images (crop_patches_kernel).  Pedestrians of one item share its canvas and their windows overlap (a 33 x 33 window of a
  // 64 x 48 image): from a few pedestrians per item on, the whole image costs less than their windows one by one.
  // dynamic LDS, sized by the launch from the batch's largest tap count (crop_aug_launch): at the usual 2-4x downscale a
  // workgroup needs ~35 KB instead of the 62 KB of the largest case, and a CU holds four of them instead of two -- the staging
  // gather is bound by latency, i.e. by the waves in flight
  extern __shared__ __attribute__((aligned(16))) unsigned char aug_lds[];
  unsigned char* stage = aug_lds;                  // a block of canvas rows, one plane (plane_b bytes) per colour
  unsigned char* tmpT = aug_lds + 3 * plane_b + 16;  // the horizontally resized strip, line (X, c) major, tpad bytes per line
  __shared__ int sbv[2 * AUG_SIDE_MAX];
  const int p = blockIdx.x, side = 2 * margin + 1, plane = side * side, tid = threadIdx.x;
  const AugItem it = items[ped_item[p]];
  float* o = small ? nullptr : out + (size_t)p * 4 * plane;
  unsigned char* so = small ? small + small_off[ped_item[p]] : nullptr;
  const int X0 = centers[2 * p] - margin, Y0 = centers[2 * p + 1] - margin;
  // the window's part inside the small image; everything else reads 0 like Image.crop
  // (bands, tile mode: the workgroup computes rows [bands[2p], bands[2p + 1]) of its tile only -- a batch of few tiles is
  //  split into row bands so that the launch fills the chip)
  const int Xa = max(X0, 0), Xb = min(X0 + side, it.sw);
  const int Ya = max(max(Y0, 0), bands ? bands[2 * p] : 0), Yb = min(min(Y0 + side, it.sh), bands ? bands[2 * p + 1] : it.sh);
  if (o)
    for (int i = tid; i < 4 * plane; i += 256) {
      const int c = i / plane, r = i % plane;
      o[i] = c == 3 ? ((r == margin * side + margin) ? 1.f : 0.f) : -1.f;
    }
  if (Xa >= Xb || Ya >= Yb) return;
  const int nX = Xb - Xa, nL = nX * 3;
  const int* kh = tables + it.kh;
  const int* bh = tables + it.bh;
  const int* kv = tables + it.kv;
  // (first source row, taps) of the window's output rows: read once (the strip loop and the vertical pass walk them)
  if (tid < 2 * (Yb - Ya)) sbv[tid] = tables[it.bv + 2 * Ya + tid];
  for (int i = tid; i < AUG_L * tpad / 4; i += 256) reinterpret_cast<unsigned*>(tmpT)[i] = 0u;  // (slack bytes: finite)
  __syncthreads();
  const int* bv = sbv - 2 * Ya;
  const int xlo = bh[2 * Xa], xhi = bh[2 * (Xb - 1)] + bh[2 * (Xb - 1) + 1], span = xhi - xlo;
  const int span_pad = (span + KS + 7) & ~3;          // a thread reads KS (+4) bytes from its first tap on: slack behind the row
  const int R = min(16, (plane_b - 44) / span_pad);          // canvas rows staged per block
  // horizontal pass: thread (l, half) = output column / channel l of the rows half, half + 2, ... of a staged block
  const int l = tid % AUG_L, half = tid / AUG_L;
  const bool hthread = tid < 2 * AUG_L && l < nL;
  const int Xi = l / 3, ch = l % 3;
  unsigned k0[KS / 4], k1[KS / 4], k2[KS / 4];
  {
    const int* krow = kh + (size_t)(Xa + (hthread ? Xi : 0)) * it.ksh;
#pragma unroll
    for (int q = 0; q < KS / 4; ++q) {
      int k[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) k[u] = (hthread && 4 * q + u < it.ksh) ? krow[4 * q + u] : 0;
      aug_pack_limbs(k, k0[q], k1[q], k2[q]);
    }
  }
  const int hx0 = hthread ? bh[2 * (Xa + Xi)] - xlo : 0;
  const unsigned char* img = atlas + it.img_off;
  // the affine map in 32-bit two's complement like Pillow's `int xx, yy` (Geometry.c checks that every coordinate of the
  // canvas stays below 32,768 pixels = 2^31 in 16.16; partial sums may wrap, the result does not)
  const unsigned ua0 = (unsigned)it.a[0], ua1 = (unsigned)it.a[1], ua2 = (unsigned)it.a[2], ua3 = (unsigned)it.a[3],
                 ua4 = (unsigned)it.a[4], ua5 = (unsigned)it.a[5];
  int Ys = Ya;
  while (Ys < Yb) {
    // strip [Ys, Ye): as many output rows as keep the source rows within rows_cap
    const int ylo = bv[2 * Ys];
    int Ye = Ys + 1;
    while (Ye < Yb && bv[2 * Ye] + bv[2 * Ye + 1] - ylo <= rows_cap) ++Ye;
    const int yhi = bv[2 * (Ye - 1)] + bv[2 * (Ye - 1) + 1];
    for (int yb = ylo; yb < yhi; yb += R) {
      const int nr = min(R, yhi - yb);
      // ---- stage nr canvas rows (gathered through the affine map, flipped on the way).  A thread owns FOUR neighbouring
      // canvas columns of eight rows (threads 0..127: rows 0-7, 128..255: rows 8-15): 32 dword loads in flight before the
      // first LDS store -- the pass is bound by the latency of the gather, not by its bytes -- and one 4-byte LDS store per
      // row and colour plane (byte stores of neighbouring lanes into one LDS word serialise) ----
      {
        const int rh = (tid >> 7) * 8;  // this thread's first row of the block
        const unsigned bx = ua2 + (unsigned)(yb + rh) * ua1, by = ua5 + (unsigned)(yb + rh) * ua4;
        for (int j4 = (tid & 127) * 4; j4 < span; j4 += 4 * 128) {
          unsigned v[8][4];
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4) {
            const int xr = xlo + j4 + c4;
            unsigned cx = bx + (unsigned)xr * ua0, cy = by + (unsigned)xr * ua3;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              v[r][c4] = 0u;
              if (rh + r < nr && j4 + c4 < span) {
                const int xin = it.rot ? ((int)cx >> 16) : xr, yin = it.rot ? ((int)cy >> 16) : yb + rh + r;
                if (xin >= 0 && xin < it.w && yin >= 0 && yin < it.h) {
                  const int xs = it.flip == 1 ? it.w - 1 - xin : xin, ys = it.flip == 2 ? it.h - 1 - yin : yin;
                  const unsigned at = (unsigned)(__mul24(ys, it.w) + xs) * 3u;  // (< 2^15 each; the image holds < 2^31 bytes)
                  __builtin_memcpy(&v[r][c4], img + at, 4);  // one (unaligned) dword; the atlas ends in spare bytes
                }
              }
              cx += ua1;
              cy += ua4;
            }
          }
#pragma unroll
          for (int r = 0; r < 8; ++r)
            if (rh + r < nr) {
              unsigned* at = reinterpret_cast<unsigned*>(stage + (rh + r) * span_pad + j4);  // (span_pad and j4 are multiples of 4)
#pragma unroll
              for (int c = 0; c < 3; ++c)
                at[c * (plane_b / 4)] = ((v[r][0] >> (8 * c)) & 0xffu) | (((v[r][1] >> (8 * c)) & 0xffu) << 8) |
                                          (((v[r][2] >> (8 * c)) & 0xffu) << 16) | (((v[r][3] >> (8 * c)) & 0xffu) << 24);
            }
        }
      }
      __syncthreads();
      // ---- horizontal pass over the staged rows: aligned words, re-aligned to the thread's first tap, four taps per dot4 ----
      if (hthread) {
        for (int r = half; r < nr; r += 2) {
          const unsigned char* row = stage + ch * plane_b + r * span_pad + hx0;
          const int phase = (int)((size_t)row & 3);
          const unsigned* wp = reinterpret_cast<const unsigned*>(row - phase);
          AugAcc acc;
          acc.clear();
          unsigned w0 = wp[0];
#pragma unroll
          for (int q = 0; q < KS / 4; ++q) {
            const unsigned w1 = wp[q + 1];
            acc.taps(__builtin_amdgcn_alignbyte(w1, w0, phase), k0[q], k1[q], k2[q]);
            w0 = w1;
          }
          tmpT[l * tpad + (yb + r - ylo)] = (unsigned char)aug_clip8(acc.value());
        }
      }
      __syncthreads();
    }
    // ---- vertical pass: output rows Ys .. Ye-1; their coefficient rows first go to LDS as packed limbs (the staging area is
    // free now): kvp[(Y - Ys)][q][3] ----
    unsigned* kvp = reinterpret_cast<unsigned*>(stage);
    const int qv = (it.ksv + 3) / 4;
    for (int i = tid; i < (Ye - Ys) * qv; i += 256) {
      const int Yi = i / qv, q = i % qv;
      int k[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) k[u] = 4 * q + u < it.ksv ? kv[(size_t)(Ys + Yi) * it.ksv + 4 * q + u] : 0;
      aug_pack_limbs(k, kvp[3 * i], kvp[3 * i + 1], kvp[3 * i + 2]);
    }
    __syncthreads();
    for (int i = tid; i < (Ye - Ys) * nL; i += 256) {
      const int Y = Ys + i / nL, ll = i % nL;
      const unsigned char* col = tmpT + ll * tpad + (bv[2 * Y] - ylo);
      const int phase = (int)((size_t)col & 3);
      const unsigned* wp = reinterpret_cast<const unsigned*>(col - phase);
      const unsigned* kq = kvp + 3 * (Y - Ys) * qv;
      AugAcc acc;
      acc.clear();
      unsigned w0 = wp[0];
      for (int q = 0; q < qv; ++q) {
        const unsigned w1 = wp[q + 1];
        acc.taps(__builtin_amdgcn_alignbyte(w1, w0, phase), kq[3 * q], kq[3 * q + 1], kq[3 * q + 2]);
        w0 = w1;
      }
      const int v = aug_clip8(acc.value());
      if (o) o[(ll % 3) * plane + (Y - Y0) * side + (Xa + ll / 3 - X0)] = (float)(-1.0 + (double)v * 2.0 / 256.0);
      else so[((size_t)Y * it.sw + Xa + ll / 3) * 3 + ll % 3] = (unsigned char)v;
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
// Nothing to pad (b == b_pad: an exact-shape entry of train()'s graph cache): the tensors are copied as they lie, four floats
// per lane where source and destination are 16-byte aligned -- no index arithmetic per element (22 MB of crops: 22 -> 9 us).
__global__ __launch_bounds__(256) void copy_batch_kernel(PadBatch a) {
  for (int k = 0; k < a.n; ++k) {
    const PadTensor& T = a.t[k];
    const long n = (long)T.outer * a.b_pad * T.inner;
    if ((((size_t)T.src | (size_t)T.dst) & 15) == 0) {
      const long n4 = n >> 2;
      const float4* s4 = reinterpret_cast<const float4*>(T.src);
      float4* d4 = reinterpret_cast<float4*>(T.dst);
      for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) d4[i] = s4[i];
      for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) T.dst[i] = T.src[i];
    } else {
      for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) T.dst[i] = T.src[i];
    }
  }
}
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
  if (b == b_pad) MG_LAUNCH(copy_batch_kernel, dim3(blocks > 2048 ? 2048 : blocks), dim3(256), 0, stream, a);
  else MG_LAUNCH(pad_batch_kernel, dim3(blocks), dim3(256), 0, stream, a);
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

static int crop_aug_launch(const unsigned char* atlas, const void* items, const int* tables, const int* ped_item, const int* centers,
                           int n, int margin, int max_taps, float* out, const long long* small_off, unsigned char* small,
                           const int* bands, hipStream_t stream) {
  const AugItem* it = (const AugItem*)items;
  // LDS of a workgroup from the batch's largest tap count: Lanczos taps = 2 * ceil(3 * scale) + 1, so scale <= taps / 6; a
  // window of 33 output pixels spans <= 33 * scale + taps source pixels in either direction
  const int ks = max_taps <= 32 ? 32 : max_taps <= 64 ? 64 : max_taps <= 96 ? 96 : 128, taps_in = max_taps;
  static const bool lds_max = [] { const char* e = getenv("MGGAN_CROP_LDS"); return e && !strcmp(e, "max"); }();  // (A/B knob)
  if (lds_max) max_taps = AUG_KS_MAX;
  const int reach = (int)(33.0 * (max_taps > 6 ? max_taps / 6.0 : 1.0)) + 1 + max_taps;
  int rows_cap = reach + 8 + 32;
  if (rows_cap > AUG_ROWS) rows_cap = AUG_ROWS;
  int tpad = rows_cap + ks + 12;
  tpad = (tpad + 3) / 4;
  if (!(tpad & 1)) ++tpad;  // (an odd number of words per line: the lines of neighbouring (X, c) start on different banks)
  tpad *= 4;
  long plane_b = 16L * (reach + ks + 16) + 44;   // sixteen staged rows; also >= the packed vertical coefficients of a strip
  if (3 * plane_b < 33L * (ks / 4) * 12 + 64) plane_b = (33L * (ks / 4) * 12 + 64) / 3 + 4;
  if (plane_b > AUG_PLANE) plane_b = AUG_PLANE;
  long words = (plane_b + 3) / 4;
  words += (11 - words % 32 + 32) % 32;          // planes 11 banks apart: the three channels of a column on different banks
  plane_b = words * 4;
  const size_t dyn = (size_t)3 * plane_b + 16 + (size_t)AUG_L * tpad;
  // (the coefficient row of an output column lives in registers: instantiated per tap count, rounded up)
#define AUG_LAUNCH(KS) MG_LAUNCH((crop_patches_aug_kernel<KS>), dim3(n), dim3(256), dyn, stream, atlas, it, tables, ped_item, centers, \
                                 margin, out, small_off, small, (int)plane_b, tpad, rows_cap, bands)
  if (taps_in <= 32) AUG_LAUNCH(32);
  else if (taps_in <= 64) AUG_LAUNCH(64);
  else if (taps_in <= 96) AUG_LAUNCH(96);
  else AUG_LAUNCH(128);
#undef AUG_LAUNCH
  return MGGAN_OK;
}

int mggan_crop_patches_aug(const unsigned char* atlas, const void* items, const int* tables, const int* ped_item,
                           const int* centers, int n, int margin, int max_taps, float* out, hipStream_t stream) {
  MG_CHECK_ARG(n >= 0 && margin >= 0 && 2 * margin + 1 <= AUG_SIDE_MAX, "crop_patches_aug: window of %d pixels (<= %d)", 2 * margin + 1,
               AUG_SIDE_MAX);
  MG_CHECK_ARG(max_taps >= 1 && max_taps <= AUG_KS_MAX, "crop_patches_aug: %d taps per pixel (<= %d)", max_taps, AUG_KS_MAX);
  if (n == 0) return MGGAN_OK;
  MG_CHECK_ARG(atlas && items && tables && ped_item && centers && out, "crop_patches_aug: null pointer");
  static_assert(sizeof(AugItem) == 26 * 4, "AugItem is 26 int32 words");
  crop_aug_launch(atlas, items, tables, ped_item, centers, n, margin, max_taps, out, nullptr, nullptr, nullptr, stream);
  MG_LAUNCH_CHECK("crop_patches_aug");
  return MGGAN_OK;
}

/* The whole resized image of every item, tile by tile: tile p (of n_tiles) belongs to item tile_item[p] and is the 33 x 33 block
   around tile_center[2p..] of that item's resized image; u8 RGB (sh, sw, 3) at small + small_off[item]. */
int mggan_aug_small_images(const unsigned char* atlas, const void* items, const int* tables, const int* tile_item,
                           const int* tile_center, const int* tile_rows, int n_tiles, int max_taps, const long long* small_off,
                           unsigned char* small, hipStream_t stream) {
  MG_CHECK_ARG(n_tiles >= 0 && max_taps >= 1 && max_taps <= AUG_KS_MAX, "aug_small_images: bad arguments");
  if (n_tiles == 0) return MGGAN_OK;
  MG_CHECK_ARG(atlas && items && tables && tile_item && tile_center && small_off && small, "aug_small_images: null pointer");
  const int n = n_tiles;
  crop_aug_launch(atlas, items, tables, tile_item, tile_center, n, (AUG_SIDE_MAX - 1) / 2, max_taps, nullptr, small_off, small,
                  tile_rows, stream);
  MG_LAUNCH_CHECK("aug_small_images");
  return MGGAN_OK;
}

}  // extern "C"
