// Scene CNN + physical (channel-softmax) attention for MG-GAN on gfx950.
//
// Replaces (file:line under /root/reference/mggan/model/modules/cnn.py):
//   Conv_Blocks  :119-160   Conv2d(3x3,p1) -> BatchNorm2d -> ReLU -> MaxPool2d(2)   (x2, C = 16 in G, 8 in D)
//   CNN.forward  :275-282   (B,4,33,33) -> (B,C,16,16) -> (B,C,8,8)
//   AttentionGlobal.forward :109-116  per position: MLP C->32->C (LeakyReLU .01), softmax over CHANNELS, sum_c a_c x_c
// BatchNorm runs in train mode on the hot path (abstract_train.py:111-112): batch statistics
// sit between conv and ReLU, so each block is "conv + per-image partial sums" -> tiny
// reduce/finalize -> the NEXT kernel applies scale/shift + ReLU + 2x2 max-pool in its prologue
// while staging its input tile into LDS (the normalised / pooled activations never touch HBM).
// Direct convolution, register-tiled: one workgroup per image, wave w owns a group of output
// channels (weights become wave-uniform -> scalar loads), lane owns a 1x4 strip of positions
// and reads its 3x6 input patch with one b128 + one b64 LDS read per row.
#include "common.h"
#include "../../include/mggan_hip.h"

#define IH 33
#define IPIX (IH * IH)
#define Y1_LD 36    // row stride of the raw conv1 output (33 + 3 pad): 1x4 strips are aligned 16-byte stores
#define IMG_LD 40     // padded row stride of the 35-row input image in LDS
#define IMG_PLANE (35 * IMG_LD)
#define A1_LD 20      // padded row stride of the 18x18 (16x16 + halo) tiles
#define A1_PLANE (18 * A1_LD)
#define HID 32

__device__ __forceinline__ void load6(const float* p, float r[6]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float2 b = *reinterpret_cast<const float2*>(p + 4);
  r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y;
}

// (4,33,33) image -> zero-haloed LDS planes.  All 4,356 pixels are requested first (contiguous, coalesced,
// 18 loads per thread in flight), the halo is cleared while they fly, then the interior lands.
template <bool ZERO>
__device__ __forceinline__ void stage_image(const float* __restrict__ img, float* imgp) {
  constexpr int NPIX = 4 * IPIX, PER = (NPIX + 255) / 256;
  float v[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int e = threadIdx.x + 256 * u;
    v[u] = img[e < NPIX ? e : 0];
  }
  if (ZERO) {
    for (int i = threadIdx.x; i < 4 * IMG_PLANE; i += 256) imgp[i] = 0.f;
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int e = threadIdx.x + 256 * u;
    if (e < NPIX) {
      const int ci = e / IPIX, rem = e - ci * IPIX, y = rem / IH, x = rem - y * IH;
      imgp[ci * IMG_PLANE + (y + 1) * IMG_LD + x + 1] = v[u];
    }
  }
}

// ---------------- conv1: (4,33,33) -> raw (C,33,33) + per-image (sum, sumsq) ----------------
template <int C>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ img, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ y1,
                                                        float* __restrict__ part) {
  constexpr int COT = C / 4;
  __shared__ __attribute__((aligned(16))) float imgp[4 * IMG_PLANE];
  const int b = blockIdx.x;
  stage_image<true>(img + (size_t)b * 4 * IPIX, imgp);
  __syncthreads();
  const int cg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), pg = threadIdx.x & 63;
  float sum[COT], sq[COT];
#pragma unroll
  for (int co = 0; co < COT; ++co) sum[co] = sq[co] = 0.f;
  for (int s = pg; s < IH * 9; s += 64) {
    const int y = s / 9, x0 = (s % 9) * 4;
    float acc[COT][4];
#pragma unroll
    for (int co = 0; co < COT; ++co) {
      const float bv = bias[cg * COT + co];
#pragma unroll
      for (int px = 0; px < 4; ++px) acc[co][px] = bv;
    }
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        float r[6];
        load6(&imgp[ci * IMG_PLANE + (y + ky) * IMG_LD + x0], r);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int co = 0; co < COT; ++co) {
            const float w = W[((cg * COT + co) * 4 + ci) * 9 + ky * 3 + kx];
#pragma unroll
            for (int px = 0; px < 4; ++px) acc[co][px] = fmaf(w, r[px + kx], acc[co][px]);
          }
      }
#pragma unroll
    for (int co = 0; co < COT; ++co) {
      float* o = y1 + (((size_t)b * C + cg * COT + co) * IH + y) * Y1_LD + x0;
      *reinterpret_cast<float4*>(o) = make_float4(acc[co][0], acc[co][1], acc[co][2], acc[co][3]);  // pad columns: unused
#pragma unroll
      for (int px = 0; px < 4; ++px)
        if (x0 + px < IH) {
          const float v = acc[co][px];
          sum[co] += v;
          sq[co] = fmaf(v, v, sq[co]);
        }
    }
  }
#pragma unroll
  for (int co = 0; co < COT; ++co) {
    const float s = wave_sum(sum[co]), q = wave_sum(sq[co]);
    if (pg == 0) {
      part[(size_t)b * 2 * C + cg * COT + co] = s;
      part[(size_t)b * 2 * C + C + cg * COT + co] = q;
    }
  }
}

// ---------------- conv1 on the matrix cores ----------------
// Implicit GEMM over FLAT output positions p = y*36 + x (the padded row stride of y1 is also the row stride of
// the zero-haloed LDS image, so the patch of position p for tap (ky,kx) sits at p + ky*36 + kx: one linear
// address space, no div/mod in the loop).  Per 16-position tile: D(16 pos x 16 co) += A(16 pos x 4 k) B(4 k x 16 co)
// for the 9 k-steps of K = 36 = (ci,ky,kx) on v_mfma_f32_16x16x4_f32 (exact f32).  B = the weights (9 registers
// per lane for the whole image), A = one ds_read_b32 per k-step (16 consecutive floats per tap; the plane stride
// 1274 == 26 mod 32 keeps the two taps of a half-wave on disjoint banks when the channel changes).  The D
// fragment holds 4 consecutive positions of one channel per lane: one aligned 16-byte store.  The VALU kernel
// above spends 3 of 4 VALU slots on operand shuffling (SQ_INSTS_VALU 5.9k per wave for 1.4k packed FMAs).
#define C1_LD 36
#define C1_PLANE 1274
typedef float c1_f32x4 __attribute__((ext_vector_type(4)));

template <int C>
__global__ __launch_bounds__(256) void conv1_fwd_mfma_kernel(const float* __restrict__ img, const float* __restrict__ W,
                                                             const float* __restrict__ bias, float* __restrict__ y1,
                                                             float* __restrict__ part) {
  constexpr int NPOS = IH * Y1_LD, NT = (NPOS + 15) / 16;  // 1188 flat positions, 75 tiles
  constexpr int NPIX = 4 * IPIX, PER = (NPIX + 255) / 256;
  __shared__ __attribute__((aligned(16))) float imgp[4 * C1_PLANE];
  __shared__ float red[4][2][16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fi = lane & 15, fk = lane >> 4;
  {
    const float* src = img + (size_t)b * NPIX;
    float v[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + 256 * u;
      v[u] = src[e < NPIX ? e : 0];
    }
    for (int i = tid; i < 4 * C1_PLANE; i += 256) imgp[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + 256 * u;
      if (e < NPIX) {
        const int ci = e / IPIX, rem = e - ci * IPIX, y = rem / IH, x = rem - y * IH;
        imgp[ci * C1_PLANE + (y + 1) * C1_LD + x + 1] = v[u];
      }
    }
  }
  float bw[9];
  int offs[9];
#pragma unroll
  for (int s = 0; s < 9; ++s) {
    const int k = 4 * s + fk, ci = k / 9, t = k - ci * 9, ky = t / 3, kx = t - ky * 3;
    bw[s] = fi < C ? W[fi * 36 + k] : 0.f;
    offs[s] = ci * C1_PLANE + ky * C1_LD + kx + fi;
  }
  const float bv = fi < C ? bias[fi] : 0.f;
  float sum = 0.f, sq = 0.f;
  __syncthreads();
  float* yout = y1 + ((size_t)b * C + fi) * NPOS + 4 * fk;
#pragma unroll 1
  for (int t = w; t < NT; t += 8) {  // two independent tiles per pass keep the matrix pipe fed
    const int p0 = 16 * t, p1 = 16 * (t + 4);
    const bool two = t + 4 < NT;
    c1_f32x4 a0 = {bv, bv, bv, bv}, a1 = {bv, bv, bv, bv};
    float x0[9], x1[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      x0[s] = imgp[p0 + offs[s]];
      x1[s] = imgp[(two ? p1 : p0) + offs[s]];
    }
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[s], bw[s], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[s], bw[s], a1, 0, 0, 0);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h == 1 && !two) break;
      const c1_f32x4 acc = h ? a1 : a0;
      const int q = (h ? p1 : p0) + 4 * fk;  // first of this lane's 4 flat positions
      if (fi < C && q < NPOS) {
        *reinterpret_cast<float4*>(yout + (h ? p1 : p0)) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        const int col = q % Y1_LD;  // multiple of 4: only col 32 has pad positions (r >= 1)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (col + r < IH) {
            sum += acc[r];
            sq = fmaf(acc[r], acc[r], sq);
          }
      }
    }
  }
  sum += __shfl_xor(sum, 16, 64); sq += __shfl_xor(sq, 16, 64);
  sum += __shfl_xor(sum, 32, 64); sq += __shfl_xor(sq, 32, 64);
  if (fk == 0) { red[w][0][fi] = sum; red[w][1][fi] = sq; }
  __syncthreads();
  if (tid < 2 * C) {
    const int which = tid / C, c = tid - which * C;
    part[(size_t)b * 2 * C + tid] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
  }
}

// sums[col] = sum_b part[b][col]   (f64 accumulation; one block per column)
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* __restrict__ part, int B, int W, double* sums) {
  __shared__ double red[256];
  const int col = blockIdx.x;
  double acc = 0.0;
  for (int r = threadIdx.x; r < B; r += 256) acc += (double)part[(size_t)r * W + col];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[col] = red[0];
}

// train: batch statistics -> scale/shift (+ running-stat update, cnn.py BN_1 momentum 0.1, eps 1e-5)
// eval : running statistics -> scale/shift.   stat[0..C) = mean, stat[C..2C) = invstd.
__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, int C, int training,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float* run_mean,
                                   float* run_var, long long* nbt, float momentum, float eps, float* scale,
                                   float* shift, float* stat) {
  const int c = threadIdx.x;
  if (c >= C) return;
  float mean, var;
  if (training) {
    const double m = sums[c] / count;
    double v = sums[C + c] / count - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m;
    var = (float)v;
    const double unb = count > 1.0 ? v * count / (count - 1.0) : v;
    float rm = run_mean[c], rv = run_var[c];
    for (int u = 0; u < training; ++u) {  // `training` = how many identical reference forwards this call stands for
      rm = (1.f - momentum) * rm + momentum * mean;
      rv = (1.f - momentum) * rv + momentum * (float)unb;
    }
    run_mean[c] = rm;
    run_var[c] = rv;
    if (c == 0) *nbt += training;
  } else {
    mean = run_mean[c];
    var = run_var[c];
  }
  const float invstd = 1.0f / sqrtf(var + eps);
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - mean * sc;
  stat[c] = mean;
  stat[C + c] = invstd;
}

// Single-GPU fast path: column sums of the per-image partials AND the finalize step in one launch.
// 1024 threads: 32 row-groups x (up to) 32 columns, f64 accumulation, fixed order.
__device__ __forceinline__ void bn_colsum_block(const float* __restrict__ part, int B, int W, double* colsum /*LDS [W]*/,
                                                double* red /*LDS [32][32]*/) {
  const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
  double acc = 0.0;
  if (col < W) {
    // 8 independent loads in flight per lane (a single block must hide the HBM latency by itself)
    int r = rg;
    for (; r + 7 * 32 < B; r += 8 * 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(r + u * 32) * W + col];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (double)v[u];
    }
    for (; r < B; r += 32) acc += (double)part[(size_t)r * W + col];
  }
  red[rg * 32 + col] = acc;
  __syncthreads();
  if (threadIdx.x < W) {
    double t = 0.0;
    for (int i = 0; i < 32; ++i) t += red[i * 32 + threadIdx.x];
    colsum[threadIdx.x] = t;
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void bn_stats_finalize_kernel(const float* __restrict__ part, int B, double count,
                                                                 int C, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float* run_mean,
                                                                 float* run_var, long long* nbt, float momentum,
                                                                 float eps, int updates, float* scale, float* shift,
                                                                 float* stat) {
  __shared__ double red[32 * 32], sums[32];
  bn_colsum_block(part, B, 2 * C, sums, red);
  const int c = threadIdx.x;
  if (c >= C) return;
  const double m = sums[c] / count;
  double v = sums[C + c] / count - m * m;
  if (v < 0.0) v = 0.0;
  const float mean = (float)m, var = (float)v;
  const float unb = (float)(count > 1.0 ? v * count / (count - 1.0) : v);
  float rm = run_mean[c], rv = run_var[c];
  for (int u = 0; u < updates; ++u) {
    rm = (1.f - momentum) * rm + momentum * mean;
    rv = (1.f - momentum) * rv + momentum * unb;
  }
  run_mean[c] = rm;
  run_var[c] = rv;
  if (c == 0) *nbt += updates;
  const float invstd = 1.0f / sqrtf(var + eps);
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - mean * sc;
  stat[c] = mean;
  stat[C + c] = invstd;
}

__global__ __launch_bounds__(1024) void bn_bwd_stats_finalize_kernel(const float* __restrict__ part, int B, double count,
                                                                     int C, const float* __restrict__ gamma,
                                                                     const float* __restrict__ stat, float* coef,
                                                                     float* dgamma, float* dbeta) {
  __shared__ double red[32 * 32], sums[32];
  bn_colsum_block(part, B, 2 * C, sums, red);
  const int c = threadIdx.x;
  if (c >= C) return;
  coef[c] = gamma[c] * stat[C + c];
  coef[C + c] = (float)(sums[c] / count);
  coef[2 * C + c] = (float)(sums[C + c] / count);
  dbeta[c] += (float)sums[c];
  dgamma[c] += (float)sums[C + c];
}

// sums = (sum g, sum g*xhat) -> coef[0..C)=gamma*invstd, [C..2C)=mean(g), [2C..3C)=mean(g*xhat); dgamma/dbeta +=
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ sums, const double* __restrict__ local,
                                       double count, int C,
                                       const float* __restrict__ gamma, const float* __restrict__ stat, float* coef,
                                       float* dgamma, float* dbeta) {
  const int c = threadIdx.x;
  if (c >= C) return;
  coef[c] = gamma[c] * stat[C + c];
  coef[C + c] = (float)(sums[c] / count);
  coef[2 * C + c] = (float)(sums[C + c] / count);
  // parameter grads use this rank's LOCAL sums (the gradient all-reduce adds the ranks up)
  dbeta[c] += (float)local[c];
  dgamma[c] += (float)local[C + c];
}

// a1 = maxpool2(relu(y1*scale+shift)) for pooled position (py,px), channel c; returns argmax code / raw value
__device__ __forceinline__ float pool_bn_relu(const float* __restrict__ base, int ld, float sc, float sh, int& code,
                                              float& raw) {
  const float2 lo = *reinterpret_cast<const float2*>(base), hi = *reinterpret_cast<const float2*>(base + ld);
  const float v[4] = {lo.x, lo.y, hi.x, hi.y};
  float best = fmaxf(fmaf(v[0], sc, sh), 0.f);
  code = 0;
  raw = v[0];
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    const float z = fmaxf(fmaf(v[k], sc, sh), 0.f);
    if (z > best) { best = z; code = k; raw = v[k]; }
  }
  return best;
}

// ---------------- conv2: BN1+ReLU+pool prologue, (C,16,16) -> raw (C,16,16) + partial stats ----------------
template <int C>
__global__ __launch_bounds__(256) void conv2_fwd_kernel(const float* __restrict__ y1, const float* __restrict__ scale1,
                                                        const float* __restrict__ shift1, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ y2,
                                                        float* __restrict__ part) {
  constexpr int COT = C / 4;
  __shared__ __attribute__((aligned(16))) float a1p[C * A1_PLANE];
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < C * A1_PLANE; i += 256) a1p[i] = 0.f;
  __syncthreads();
  {
    const int py = threadIdx.x >> 4, px = threadIdx.x & 15;
#pragma unroll 4
    for (int c = 0; c < C; ++c) {
      int code; float raw;
      const float* base = y1 + (((size_t)b * C + c) * IH + 2 * py) * Y1_LD + 2 * px;
      a1p[c * A1_PLANE + (py + 1) * A1_LD + px + 1] = pool_bn_relu(base, Y1_LD, scale1[c], shift1[c], code, raw);
    }
  }
  __syncthreads();
  const int cg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), pg = threadIdx.x & 63;
  const int py = pg >> 2, x0 = (pg & 3) * 4;
  float acc[COT][4];
#pragma unroll
  for (int co = 0; co < COT; ++co) {
    const float bv = bias[cg * COT + co];
#pragma unroll
    for (int px = 0; px < 4; ++px) acc[co][px] = bv;
  }
#pragma unroll 2
  for (int ci = 0; ci < C; ++ci)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      float r[6];
      load6(&a1p[ci * A1_PLANE + (py + ky) * A1_LD + x0], r);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int co = 0; co < COT; ++co) {
          const float w = W[((cg * COT + co) * C + ci) * 9 + ky * 3 + kx];
#pragma unroll
          for (int px = 0; px < 4; ++px) acc[co][px] = fmaf(w, r[px + kx], acc[co][px]);
        }
    }
#pragma unroll
  for (int co = 0; co < COT; ++co) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int px = 0; px < 4; ++px) { s += acc[co][px]; q = fmaf(acc[co][px], acc[co][px], q); }
    *reinterpret_cast<float4*>(y2 + (((size_t)b * C + cg * COT + co) * 16 + py) * 16 + x0) =
        make_float4(acc[co][0], acc[co][1], acc[co][2], acc[co][3]);
    s = wave_sum(s);
    q = wave_sum(q);
    if (pg == 0) {
      part[(size_t)b * 2 * C + cg * COT + co] = s;
      part[(size_t)b * 2 * C + C + cg * COT + co] = q;
    }
  }
}

// ---------------- attention head: BN2+ReLU+pool prologue, 64 positions per image ----------------
// One lane per position (wave = image).  The 2*32*C MLP weights are staged in LDS and the hidden layer is
// walked with a rolled loop (hidden unit k: recompute h_k, use it, forget it): keeping all of it in
// registers / scalar registers made the compiler spill ~1k SGPRs through v_writelane (12.6k instructions).
template <int C, bool BWD>
__global__ __launch_bounds__(256) void attn_kernel(int B, const float* __restrict__ y2, const float* __restrict__ scale2,
                                                   const float* __restrict__ shift2, const float* __restrict__ Wa,
                                                   const float* __restrict__ ba, const float* __restrict__ Wb,
                                                   const float* __restrict__ bb, float* out, int ld_out,
                                                   // backward only
                                                   const float* __restrict__ dout, int ld_dout,
                                                   const float* __restrict__ stat2, float* ds_s, float* hact,
                                                   float* dz_s, float* vsave, float* G2, float* part) {
  __shared__ __attribute__((aligned(16))) float wa[HID][C];   // Wa[k][c]
  __shared__ __attribute__((aligned(16))) float wbT[HID][C];  // Wb[c][k] transposed
  __shared__ float bas[HID];
  for (int i = threadIdx.x; i < HID * C; i += 256) {
    wa[i / C][i % C] = Wa[i];
    wbT[i % HID][i / HID] = Wb[i];
  }
  if (threadIdx.x < HID) bas[threadIdx.x] = ba[threadIdx.x];
  __syncthreads();
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), pos = threadIdx.x & 63;
  if (b >= B) return;  // whole waves leave together
  const int py = pos >> 3, px = pos & 7;
  float v[C], raw[C];
  int code[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float* base = y2 + (((size_t)b * C + c) * 16 + 2 * py) * 16 + 2 * px;
    v[c] = pool_bn_relu(base, 16, scale2[c], shift2[c], code[c], raw[c]);
  }
  float sc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) sc[c] = bb[c];
#pragma unroll 2
  for (int k = 0; k < HID; ++k) {
    float h0 = bas[k], h1 = 0.f;
#pragma unroll
    for (int c = 0; c < C; c += 2) { h0 = fmaf(wa[k][c], v[c], h0); h1 = fmaf(wa[k][c + 1], v[c + 1], h1); }
    float h = h0 + h1;
    h = h > 0.f ? h : 0.01f * h;  // nn.LeakyReLU() default slope, cnn.py:19-20
#pragma unroll
    for (int c = 0; c < C; ++c) sc[c] = fmaf(wbT[k][c], h, sc[c]);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < C; ++c) mx = fmaxf(mx, sc[c]);
  float den = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { sc[c] = __expf(sc[c] - mx); den += sc[c]; }
  const float inv = 1.f / den;
  float o = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { sc[c] *= inv; o = fmaf(sc[c], v[c], o); }
  if (!BWD) {
    out[(size_t)b * ld_out + pos] = o;
    return;
  }
  // ---- backward: out = sum_c a_c v_c, a = softmax(s) ----
  const float go = dout[(size_t)b * ld_dout + pos];
  const size_t row = (size_t)b * 64 + pos, NR = (size_t)B * 64;  // saved arrays are [feature][NR]
  float dv[C], dsv[C], dot = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { dv[c] = go * sc[c]; dot = fmaf(sc[c], go * v[c], dot); }
#pragma unroll
  for (int c = 0; c < C; ++c) {
    dsv[c] = sc[c] * (go * v[c] - dot);
    ds_s[c * NR + row] = dsv[c];
    vsave[c * NR + row] = v[c];
  }
#pragma unroll 2
  for (int k = 0; k < HID; ++k) {
    float h0 = bas[k], h1 = 0.f, d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int c = 0; c < C; c += 2) {
      h0 = fmaf(wa[k][c], v[c], h0);
      h1 = fmaf(wa[k][c + 1], v[c + 1], h1);
      d0 = fmaf(wbT[k][c], dsv[c], d0);
      d1 = fmaf(wbT[k][c + 1], dsv[c + 1], d1);
    }
    const float hp = h0 + h1;
    const float d = (d0 + d1) * (hp > 0.f ? 1.f : 0.01f);
    dz_s[k * NR + row] = d;
    hact[k * NR + row] = hp > 0.f ? hp : 0.01f * hp;
#pragma unroll
    for (int c = 0; c < C; ++c) dv[c] = fmaf(wa[k][c], d, dv[c]);
  }
  // route through max-pool + ReLU to the raw conv2 output grid; partial sums for the BN backward
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float g = v[c] > 0.f ? dv[c] : 0.f;
    float* gb = G2 + (((size_t)b * C + c) * 16 + 2 * py) * 16 + 2 * px;
    *reinterpret_cast<float2*>(gb) = make_float2(code[c] == 0 ? g : 0.f, code[c] == 1 ? g : 0.f);
    *reinterpret_cast<float2*>(gb + 16) = make_float2(code[c] == 2 ? g : 0.f, code[c] == 3 ? g : 0.f);
    const float xh = (raw[c] - stat2[c]) * stat2[C + c];
    const float s1 = wave_sum(g), s2 = wave_sum(g * xh);
    if (pos == 0) {
      part[(size_t)b * 2 * C + c] = s1;
      part[(size_t)b * 2 * C + C + c] = s2;
    }
  }
}

// ---------------- conv2 backward: BN2 bwd + weight grad + input grad routed through pool1/ReLU ----------------
template <int C>
__global__ __launch_bounds__(256) void conv2_bwd_kernel(int B, const float* __restrict__ y1,
                                                        const float* __restrict__ scale1,
                                                        const float* __restrict__ shift1,
                                                        const float* __restrict__ stat1, const float* __restrict__ y2,
                                                        const float* __restrict__ G2, const float* __restrict__ stat2,
                                                        const float* __restrict__ coef2, const float* __restrict__ W,
                                                        float* G1c, unsigned char* code1, float* part1, float* wpart) {
  constexpr int COT = C / 4, PAIRS = C * C, NQ = 256 / PAIRS, ROWS = 16 / NQ, WLEN = PAIRS * 9 + C;
  __shared__ __attribute__((aligned(16))) float dyp[C * A1_PLANE];
  __shared__ __attribute__((aligned(16))) float a1p[C * A1_PLANE];
  __shared__ float y1r[C * 256];
  for (int i = threadIdx.x; i < C * A1_PLANE; i += 256) { dyp[i] = 0.f; a1p[i] = 0.f; }
  const int pair = threadIdx.x % PAIRS, rq = threadIdx.x / PAIRS;
  const int wco = pair / C, wci = pair % C;
  float wacc[9], bacc = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) wacc[k] = 0.f;
  const int cg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), pg = threadIdx.x & 63;
  const int ty = pg >> 2, x0 = (pg & 3) * 4;

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    lds_barrier();
    {
      const int py = threadIdx.x >> 4, px = threadIdx.x & 15;
#pragma unroll 8
      for (int c = 0; c < C; ++c) {
        int code; float raw;
        const float* base = y1 + (((size_t)b * C + c) * IH + 2 * py) * Y1_LD + 2 * px;
        a1p[c * A1_PLANE + (py + 1) * A1_LD + px + 1] = pool_bn_relu(base, Y1_LD, scale1[c], shift1[c], code, raw);
        y1r[c * 256 + threadIdx.x] = raw;
        const size_t gi = ((size_t)b * C + c) * 256 + threadIdx.x;
        code1[gi] = (unsigned char)code;
        const float xh = (y2[gi] - stat2[c]) * stat2[C + c];
        dyp[c * A1_PLANE + (py + 1) * A1_LD + px + 1] = coef2[c] * (G2[gi] - coef2[C + c] - xh * coef2[2 * C + c]);
      }
    }
    lds_barrier();
    // (1) weight gradient: dW[co][ci][ky][kx] += sum_{y,x} dy[co][y][x] * a1p[ci][y+ky][x+kx]
    for (int y = rq * ROWS; y < (rq + 1) * ROWS; ++y) {
      float dy[16];
#pragma unroll
      for (int x = 0; x < 16; x += 4) {
        // interior starts at column 1 -> unaligned for b128; read scalars
        dy[x] = dyp[wco * A1_PLANE + (y + 1) * A1_LD + x + 1];
        dy[x + 1] = dyp[wco * A1_PLANE + (y + 1) * A1_LD + x + 2];
        dy[x + 2] = dyp[wco * A1_PLANE + (y + 1) * A1_LD + x + 3];
        dy[x + 3] = dyp[wco * A1_PLANE + (y + 1) * A1_LD + x + 4];
      }
      if (wci == 0) {
#pragma unroll
        for (int x = 0; x < 16; ++x) bacc += dy[x];
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        float ar[20];
#pragma unroll
        for (int x = 0; x < 20; x += 4) {
          const float4 t4 = *reinterpret_cast<const float4*>(&a1p[wci * A1_PLANE + (y + ky) * A1_LD + x]);
          ar[x] = t4.x; ar[x + 1] = t4.y; ar[x + 2] = t4.z; ar[x + 3] = t4.w;
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int x = 0; x < 16; ++x) wacc[ky * 3 + kx] = fmaf(dy[x], ar[x + kx], wacc[ky * 3 + kx]);
      }
    }
    // (2) input gradient (flipped kernel), tile = COT input channels x 4 positions
    float acc[COT][4];
#pragma unroll
    for (int ci = 0; ci < COT; ++ci)
#pragma unroll
      for (int px = 0; px < 4; ++px) acc[ci][px] = 0.f;
#pragma unroll 2
    for (int co = 0; co < C; ++co)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        float r[6];
        load6(&dyp[co * A1_PLANE + (ty + ky) * A1_LD + x0], r);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int ci = 0; ci < COT; ++ci) {
            const float w = W[((co * C) + cg * COT + ci) * 9 + (2 - ky) * 3 + (2 - kx)];
#pragma unroll
            for (int px = 0; px < 4; ++px) acc[ci][px] = fmaf(w, r[px + kx], acc[ci][px]);
          }
      }
#pragma unroll
    for (int ci = 0; ci < COT; ++ci) {
      const int c = cg * COT + ci;
      float g[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const bool on = a1p[c * A1_PLANE + (ty + 1) * A1_LD + x0 + px + 1] > 0.f;
        g[px] = on ? acc[ci][px] : 0.f;
        const float xh = (y1r[c * 256 + ty * 16 + x0 + px] - stat1[c]) * stat1[C + c];
        s1 += g[px];
        s2 = fmaf(g[px], xh, s2);
      }
      *reinterpret_cast<float4*>(G1c + (((size_t)b * C + c) * 16 + ty) * 16 + x0) = make_float4(g[0], g[1], g[2], g[3]);
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      if (pg == 0) {
        part1[(size_t)b * 2 * C + c] = s1;
        part1[(size_t)b * 2 * C + C + c] = s2;
      }
    }
  }
  // fold the NQ row-group partials in LDS (fixed order) -> one partial row per workgroup
  lds_barrier();
  float* red = dyp;  // reuse (>= NQ * WLEN floats)
  {
    float* rp = red + rq * WLEN;
#pragma unroll
    for (int k = 0; k < 9; ++k) rp[pair * 9 + k] = wacc[k];
    if (wci == 0) rp[PAIRS * 9 + wco] = bacc;
  }
  lds_barrier();
  float* wp = wpart + (size_t)blockIdx.x * WLEN;
  for (int i = threadIdx.x; i < WLEN; i += 256) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) t += red[q * WLEN + i];
    wp[i] = t;
  }
}

// ---------------- conv2 backward on the matrix cores (C = 16) ----------------
// Same staging as conv2_bwd_kernel; both GEMM-shaped parts run on exact-f32 MFMA (16x16x4):
//   weight grad  dW[co][(tap,ci)] += sum_pos dy[co][pos] * a1[ci][pos+tap]   M=co(16), N=(tap,ci)(9 tiles), K=pos
//   input grad   da1[pos][ci]      = sum_{tap,co} dy[co][pos+tap'] * Wflip   M=pos (16 rows of 16), N=ci, K=(tap,co)
// LDS planes are 361 floats apart (== 9 mod 32): a 16-lane fragment read that walks channels hits 16
// distinct banks.  Wave w owns image rows 4w..4w+3 in both products; the wave-private weight-grad
// accumulators persist over the workgroup's images and meet in LDS once at the end.
#define C2_PLANE 361
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void conv2_bwd_mfma_kernel(int B, const float* __restrict__ y1,
                                                            const float* __restrict__ scale1,
                                                            const float* __restrict__ shift1,
                                                            const float* __restrict__ stat1,
                                                            const float* __restrict__ y2, const float* __restrict__ G2,
                                                            const float* __restrict__ stat2,
                                                            const float* __restrict__ coef2, const float* __restrict__ W,
                                                            float* G1c, unsigned char* code1, float* part1,
                                                            float* wpart) {
  constexpr int C = 16, WLEN = C * C * 9 + C;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dyp = smem;                       // [C][18][20] padded planes, stride C2_PLANE
  float* a1p = dyp + C * C2_PLANE;         // same layout
  float* y1r = a1p + C * C2_PLANE;         // [C][256] raw conv1 value at the pooling argmax
  float* wf = y1r + C * 256;               // [tap'][co][ci] = W[co][ci][8 - tap'] (flipped kernel), 2304
  float* red = wf + 9 * C * C;             // [4][32] cross-wave statistics
  for (int i = threadIdx.x; i < 2 * C * C2_PLANE; i += 256) dyp[i] = 0.f;  // dyp and a1p (halos stay zero)
  for (int i = threadIdx.x; i < 9 * C * C; i += 256) {
    const int tp = i / (C * C), co = (i / C) % C, ci = i % C;
    wf[i] = W[(co * C + ci) * 9 + (8 - tp)];
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  f32x4_t wacc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wacc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bacc = 0.f;

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    lds_barrier();
    {
      const int py = threadIdx.x >> 4, px = threadIdx.x & 15;
#pragma unroll 8
      for (int c = 0; c < C; ++c) {
        int code; float raw;
        const float* base = y1 + (((size_t)b * C + c) * IH + 2 * py) * Y1_LD + 2 * px;
        a1p[c * C2_PLANE + (py + 1) * A1_LD + px + 1] = pool_bn_relu(base, Y1_LD, scale1[c], shift1[c], code, raw);
        y1r[c * 256 + threadIdx.x] = raw;
        const size_t gi = ((size_t)b * C + c) * 256 + threadIdx.x;
        code1[gi] = (unsigned char)code;
        const float xh = (y2[gi] - stat2[c]) * stat2[C + c];
        dyp[c * C2_PLANE + (py + 1) * A1_LD + px + 1] = coef2[c] * (G2[gi] - coef2[C + c] - xh * coef2[2 * C + c]);
      }
    }
    lds_barrier();
    // ---- (1) weight gradient: K runs over this wave's 64 positions (rows 4w..4w+3), 4 positions per MFMA
#pragma unroll 1
    for (int ks = 0; ks < 16; ++ks) {
      const int y = 4 * w + (ks >> 2), x0 = (ks & 3) * 4;
      // A[i = co][k = position x0+fk]
      const float a = dyp[fi * C2_PLANE + (y + 1) * A1_LD + x0 + 1 + fk];
      bacc += a;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        // B[k = position][j = ci] for tap t = (ky,kx): a1 value at (y+ky, x+kx) in padded coordinates
        const float bv = a1p[fi * C2_PLANE + (y + t / 3) * A1_LD + x0 + fk + t % 3];
        wacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, wacc[t], 0, 0, 0);
      }
    }
    // ---- (2) input gradient for rows 4w..4w+3: M = 16 x-positions of one row, N = ci, K = (tap', co)
    float s1 = 0.f, s2 = 0.f;  // BN1-backward partial sums for channel ci = fi
#pragma unroll 1
    for (int ry = 0; ry < 4; ++ry) {
      const int y = 4 * w + ry;
      f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
        for (int c4 = 0; c4 < C; c4 += 4) {
          const int co = c4 + fk;
          // A[i = x][k = (tp, co)] = dy[co] at padded (y + tp/3, x + tp%3);  B[k][j = ci] = wf[tp][co][ci]
          const float a = dyp[co * C2_PLANE + (y + tp / 3) * A1_LD + fi + tp % 3];
          const float bv = wf[(tp * C + co) * C + fi];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc, 0, 0, 0);
        }
      }
      // D fragment: register r <-> x = 4*fk + r, column = ci = fi
      float g[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int x = 4 * fk + r;
        const bool on = a1p[fi * C2_PLANE + (y + 1) * A1_LD + x + 1] > 0.f;
        g[r] = on ? acc[r] : 0.f;
        const float xh = (y1r[fi * 256 + y * 16 + x] - stat1[fi]) * stat1[C + fi];
        s1 += g[r];
        s2 = fmaf(g[r], xh, s2);
      }
      *reinterpret_cast<float4*>(G1c + (((size_t)b * C + fi) * 16 + y) * 16 + 4 * fk) = make_float4(g[0], g[1], g[2], g[3]);
    }
    // per-channel statistics: fold the 4 lane groups, then the 4 waves
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
    if (lane < 16) { red[w * 32 + lane] = s1; red[w * 32 + 16 + lane] = s2; }
    lds_barrier();
    if (threadIdx.x < 32) {
      const float t = (red[threadIdx.x] + red[32 + threadIdx.x]) + (red[64 + threadIdx.x] + red[96 + threadIdx.x]);
      part1[(size_t)b * 2 * C + threadIdx.x] = t;  // [0,16) = sum g, [16,32) = sum g*xhat
    }
  }
  // ---- fold the four waves' weight-gradient fragments; wacc[t][r] of lane l is dW[co = 4*fk + r][ci = fi][tap t]
  lds_barrier();
  float* fold = dyp;  // reuse: [4][WLEN]
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) fold[w * WLEN + ((4 * fk + r) * C + fi) * 9 + t] = wacc[t][r];
  bacc += __shfl_xor(bacc, 16, 64);
  bacc += __shfl_xor(bacc, 32, 64);  // lanes 0..15: sum over this wave's positions of dy[co = lane]
  if (lane < 16) fold[w * WLEN + C * C * 9 + lane] = bacc;
  lds_barrier();
  float* wp = wpart + (size_t)blockIdx.x * WLEN;
  for (int i = threadIdx.x; i < WLEN; i += 256)
    wp[i] = (fold[i] + fold[WLEN + i]) + (fold[2 * WLEN + i] + fold[3 * WLEN + i]);
}

// ---------------- conv1 backward: BN1 bwd + weight grad (the image needs no gradient) ----------------
// Channels are processed 8 at a time so that a workgroup needs 60 KB of LDS (two workgroups per CU);
// the dy1 tile is built cell-wise (one lane per pooled cell, loads of several channels in flight).
template <int C>
__global__ __launch_bounds__(256) void conv1_bwd_kernel(int B, const float* __restrict__ img,
                                                        const float* __restrict__ y1, const float* __restrict__ stat1,
                                                        const float* __restrict__ coef1, const float* __restrict__ G1c,
                                                        const unsigned char* __restrict__ code1, float* wpart) {
  constexpr int CH = 8, NP = C / CH, PAIRS = CH * 4, NQ = 256 / PAIRS, RPQ = (IH + NQ - 1) / NQ, DLD = 36;
  constexpr int WLEN = C * 4 * 9 + C;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* imgp = smem;                  // 4*IMG_PLANE
  float* dy1 = smem + 4 * IMG_PLANE;   // CH*33*36
  for (int i = threadIdx.x; i < CH * IH * DLD; i += 256) dy1[i] = 0.f;
  for (int i = threadIdx.x; i < 4 * IMG_PLANE; i += 256) imgp[i] = 0.f;  // halo stays zero for every image
  const int pair = threadIdx.x % PAIRS, rq = threadIdx.x / PAIRS;
  const int lco = pair / 4, wci = pair % 4;
  float wacc[NP][9], bacc[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    bacc[q] = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) wacc[q][k] = 0.f;
  }
  const int py = threadIdx.x >> 4, px = threadIdx.x & 15;
  // Software pipeline: everything pass (b, q) needs from global memory (the raw conv1 outputs of the 2x2
  // window of this lane's pooled cell, its gradient and argmax code for 8 channels, the 65 border positions,
  // and for q == 0 the image) is requested BEFORE the product phase of the previous pass and lands in LDS
  // after it: the loads fly under ~300 FMAs per lane instead of stalling the two resident workgroups.
  constexpr int NPIX = 4 * IPIX, PER = (NPIX + 255) / 256, NBORD = (CH * 65 + 255) / 256;
  float2 r_lo[CH], r_hi[CH];
  float r_g[CH], r_b[NBORD], r_img[PER];
  int r_code[CH];
  auto fetch = [&](int b, int q, bool with_img) {
#pragma unroll
    for (int lc = 0; lc < CH; ++lc) {
      const int c = q * CH + lc;
      const size_t pi = ((size_t)b * C + c) * 256 + threadIdx.x;
      const float* yb = y1 + (((size_t)b * C + c) * IH + 2 * py) * Y1_LD + 2 * px;
      r_lo[lc] = *reinterpret_cast<const float2*>(yb);
      r_hi[lc] = *reinterpret_cast<const float2*>(yb + Y1_LD);
      r_g[lc] = G1c[pi];
      r_code[lc] = code1[pi];
    }
#pragma unroll
    for (int u = 0; u < NBORD; ++u) {
      const int idx = threadIdx.x + 256 * u, ic = idx < CH * 65 ? idx : 0;
      const int lc = ic / 65, e = ic - lc * 65, c = q * CH + lc;
      const int y = e < 33 ? 32 : e - 33, x = e < 33 ? e : 32;
      r_b[u] = y1[(((size_t)b * C + c) * IH + y) * Y1_LD + x];
    }
    if (with_img) {
      const float* src = img + (size_t)b * NPIX;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int e = threadIdx.x + 256 * u;
        r_img[u] = src[e < NPIX ? e : 0];
      }
    }
  };
  if ((int)blockIdx.x < B) fetch(blockIdx.x, 0, true);
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      lds_barrier();  // the previous product phase is done with dy1 (and, for q == 0, with the image)
      if (q == 0) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
          const int e = threadIdx.x + 256 * u;
          if (e < NPIX) {
            const int ci = e / IPIX, rem = e - ci * IPIX, y = rem / IH, x = rem - y * IH;
            imgp[ci * IMG_PLANE + (y + 1) * IMG_LD + x + 1] = r_img[u];
          }
        }
      }
      // interior 32x32: one lane per pooled cell
#pragma unroll
      for (int lc = 0; lc < CH; ++lc) {
        const int c = q * CH + lc;
        const float v0 = r_lo[lc].x, v1 = r_lo[lc].y, v2 = r_hi[lc].x, v3 = r_hi[lc].y;
        const float g = r_g[lc];
        const int code = r_code[lc];
        const float mean = stat1[c], inv = stat1[C + c], cs = coef1[c], m1 = coef1[C + c], m2 = coef1[2 * C + c];
        float* d = dy1 + (lc * IH + 2 * py) * DLD + 2 * px;
        d[0] = cs * ((code == 0 ? g : 0.f) - m1 - (v0 - mean) * inv * m2);
        d[1] = cs * ((code == 1 ? g : 0.f) - m1 - (v1 - mean) * inv * m2);
        d[DLD] = cs * ((code == 2 ? g : 0.f) - m1 - (v2 - mean) * inv * m2);
        d[DLD + 1] = cs * ((code == 3 ? g : 0.f) - m1 - (v3 - mean) * inv * m2);
      }
      // border row 32 / column 32 (never pooled: g = 0)
#pragma unroll
      for (int u = 0; u < NBORD; ++u) {
        const int idx = threadIdx.x + 256 * u;
        if (idx < CH * 65) {
          const int lc = idx / 65, e = idx - lc * 65, c = q * CH + lc;
          const int y = e < 33 ? 32 : e - 33, x = e < 33 ? e : 32;
          dy1[(lc * IH + y) * DLD + x] = coef1[c] * (-coef1[C + c] - (r_b[u] - stat1[c]) * stat1[C + c] * coef1[2 * C + c]);
        }
      }
      lds_barrier();
      {  // next pass's operands: in flight during this pass's products
        const bool last_q = q == NP - 1;
        const int nb = last_q ? b + (int)gridDim.x : b, nq = last_q ? 0 : q + 1;
        if (nb < B) fetch(nb, nq, last_q);
      }
      const int yend = min(IH, (rq + 1) * RPQ);
      for (int y = rq * RPQ; y < yend; ++y) {
        float dy[DLD];
#pragma unroll
        for (int x = 0; x < DLD; x += 4) {
          const float4 t4 = *reinterpret_cast<const float4*>(&dy1[(lco * IH + y) * DLD + x]);
          dy[x] = t4.x; dy[x + 1] = t4.y; dy[x + 2] = t4.z; dy[x + 3] = t4.w;
        }
        if (wci == 0) {
#pragma unroll
          for (int x = 0; x < IH; ++x) bacc[q] += dy[x];
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          float ar[IMG_LD];
#pragma unroll
          for (int x = 0; x < IMG_LD; x += 4) {
            const float4 t4 = *reinterpret_cast<const float4*>(&imgp[wci * IMG_PLANE + (y + ky) * IMG_LD + x]);
            ar[x] = t4.x; ar[x + 1] = t4.y; ar[x + 2] = t4.z; ar[x + 3] = t4.w;
          }
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int x = 0; x < IH; ++x) wacc[q][ky * 3 + kx] = fmaf(dy[x], ar[x + kx], wacc[q][ky * 3 + kx]);
        }
      }
    }
  }
  lds_barrier();
  float* red = dy1;  // reuse (>= NQ * WLEN floats: 8 * 592 <= 9504)
  {
    float* rp = red + rq * WLEN;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int co = q * CH + lco;
#pragma unroll
      for (int k = 0; k < 9; ++k) rp[(co * 4 + wci) * 9 + k] = wacc[q][k];
      if (wci == 0) rp[C * 36 + co] = bacc[q];
    }
  }
  lds_barrier();
  float* wp = wpart + (size_t)blockIdx.x * WLEN;
  for (int i = threadIdx.x; i < WLEN; i += 256) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) t += red[q * WLEN + i];
    wp[i] = t;
  }
}

// dst[o] += sum_z P[z*stride + o], o < len   (64 outputs x 16 z-lanes per block, fixed order)
__global__ __launch_bounds__(1024) void partial_sum_kernel(const float* __restrict__ P, int nP, int stride, int len,
                                                           float* dst) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, zl = threadIdx.x >> 6;
  const int o = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (o < len) {
    float s0 = 0.f, s1 = 0.f;
    int z = zl;
    for (; z + 16 < nP; z += 32) { s0 += P[(size_t)z * stride + o]; s1 += P[(size_t)(z + 16) * stride + o]; }
    if (z < nP) s0 += P[(size_t)z * stride + o];
    s = s0 + s1;
  }
  red[zl][lane] = s;
  __syncthreads();
  if (zl == 0 && o < len) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][lane];
    dst[o] += t;
  }
}

// two workgroups per CU resident, every workgroup gets the same number of images (+-1)
static int persistent_grid(int B) {
  // 512 workgroups (two per CU) whenever there are more images than that: workgroup i takes images i, i + 512, ...,
  // so with 1,280 images the first 256 workgroups get three and the other 256 two -- five per CU if the dispatcher
  // deals workgroups round-robin -- where ceil(B / 3) = 427 equal workgroups left a third of the CUs with one
  // workgroup and the rest with two (six images)
  return B <= 512 ? B : 512;
}

extern "C" {

int mggan_cnn_bwd_grid(int B) { return persistent_grid(B); }

int mggan_conv1_fwd(const float* img, int B, int C, const float* W, const float* bias, float* y1, float* part,
                    hipStream_t stream) {
  MG_CHECK_ARG(C == 8 || C == 16, "conv1_fwd: channels %d not built (8 or 16)", C);
  if (B == 0) return MGGAN_OK;
  MG_CHECK_ARG(img && W && bias && y1 && part, "conv1_fwd: null pointer");
  if (C == 16) hipLaunchKernelGGL((conv1_fwd_mfma_kernel<16>), dim3(B), dim3(256), 0, stream, img, W, bias, y1, part);
  else hipLaunchKernelGGL((conv1_fwd_mfma_kernel<8>), dim3(B), dim3(256), 0, stream, img, W, bias, y1, part);
  MG_LAUNCH_CHECK("conv1_fwd");
  return MGGAN_OK;
}

int mggan_bn_reduce(const float* part, int B, int W, double* sums, hipStream_t stream) {
  MG_CHECK_ARG(part && sums && W > 0, "bn_reduce: bad arguments");
  hipLaunchKernelGGL(bn_reduce_kernel, dim3(W), dim3(256), 0, stream, part, B, W, sums);
  MG_LAUNCH_CHECK("bn_reduce");
  return MGGAN_OK;
}

int mggan_bn_finalize(const double* sums, double count, int C, int training, const float* gamma, const float* beta,
                      float* run_mean, float* run_var, long long* num_batches_tracked, float momentum, float eps,
                      float* scale, float* shift, float* stat, hipStream_t stream) {
  MG_CHECK_ARG(gamma && beta && run_mean && run_var && scale && shift && stat && C <= 64, "bn_finalize: bad arguments");
  MG_CHECK_ARG(!training || (sums && num_batches_tracked), "bn_finalize: training needs sums and the batch counter");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(64), 0, stream, sums, count, C, training, gamma, beta, run_mean,
                     run_var, num_batches_tracked, momentum, eps, scale, shift, stat);
  MG_LAUNCH_CHECK("bn_finalize");
  return MGGAN_OK;
}

int mggan_bn_stats_finalize(const float* part, int B, double count, int C, const float* gamma, const float* beta,
                            float* run_mean, float* run_var, long long* num_batches_tracked, float momentum, float eps,
                            int updates, float* scale, float* shift, float* stat, hipStream_t stream) {
  MG_CHECK_ARG(part && gamma && beta && run_mean && run_var && num_batches_tracked && scale && shift && stat && C <= 16,
               "bn_stats_finalize: bad arguments");
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(1), dim3(1024), 0, stream, part, B, count, C, gamma, beta, run_mean,
                     run_var, num_batches_tracked, momentum, eps, updates, scale, shift, stat);
  MG_LAUNCH_CHECK("bn_stats_finalize");
  return MGGAN_OK;
}

int mggan_bn_bwd_stats_finalize(const float* part, int B, double count, int C, const float* gamma, const float* stat,
                                float* coef, float* dgamma, float* dbeta, hipStream_t stream) {
  MG_CHECK_ARG(part && gamma && stat && coef && dgamma && dbeta && C <= 16, "bn_bwd_stats_finalize: bad arguments");
  hipLaunchKernelGGL(bn_bwd_stats_finalize_kernel, dim3(1), dim3(1024), 0, stream, part, B, count, C, gamma, stat, coef,
                     dgamma, dbeta);
  MG_LAUNCH_CHECK("bn_bwd_stats_finalize");
  return MGGAN_OK;
}

int mggan_bn_bwd_finalize(const double* sums, const double* local_sums, double count, int C, const float* gamma,
                          const float* stat, float* coef, float* dgamma, float* dbeta, hipStream_t stream) {
  MG_CHECK_ARG(sums && local_sums && gamma && stat && coef && dgamma && dbeta && C <= 64,
               "bn_bwd_finalize: bad arguments");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(64), 0, stream, sums, local_sums, count, C, gamma, stat, coef,
                     dgamma, dbeta);
  MG_LAUNCH_CHECK("bn_bwd_finalize");
  return MGGAN_OK;
}

int mggan_conv2_fwd(const float* y1, int B, int C, const float* scale1, const float* shift1, const float* W,
                    const float* bias, float* y2, float* part, hipStream_t stream) {
  MG_CHECK_ARG(C == 8 || C == 16, "conv2_fwd: channels %d not built (8 or 16)", C);
  if (B == 0) return MGGAN_OK;
  MG_CHECK_ARG(y1 && scale1 && shift1 && W && bias && y2 && part, "conv2_fwd: null pointer");
  // (an implicit-GEMM version on the matrix cores -- 16 position tiles x 36 v_mfma_f32_16x16x4_f32 steps per image,
  //  weights as 36 B-fragment registers, one LDS read per step -- measured 52 us against 38 us for this VALU kernel:
  //  exact-f32 MFMA has the VALU's FLOP rate, and with K = 4 per instruction the operand reads are not amortised the
  //  way the register-blocked VALU loop amortises them)
  if (C == 16)
    hipLaunchKernelGGL((conv2_fwd_kernel<16>), dim3(B), dim3(256), 0, stream, y1, scale1, shift1, W, bias, y2, part);
  else
    hipLaunchKernelGGL((conv2_fwd_kernel<8>), dim3(B), dim3(256), 0, stream, y1, scale1, shift1, W, bias, y2, part);
  MG_LAUNCH_CHECK("conv2_fwd");
  return MGGAN_OK;
}

int mggan_scene_attention_fwd(const float* y2, int B, int C, const float* scale2, const float* shift2, const float* Wa,
                              const float* ba, const float* Wb, const float* bb, float* out, int ld_out,
                              hipStream_t stream) {
  MG_CHECK_ARG(C == 8 || C == 16, "scene_attention_fwd: channels %d not built (8 or 16)", C);
  if (B == 0) return MGGAN_OK;
  MG_CHECK_ARG(y2 && scale2 && shift2 && Wa && ba && Wb && bb && out, "scene_attention_fwd: null pointer");
  if (C == 16)
    hipLaunchKernelGGL((attn_kernel<16, false>), dim3(cdiv(B, 4)), dim3(256), 0, stream, B, y2, scale2, shift2, Wa, ba,
                       Wb, bb, out, ld_out, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  else
    hipLaunchKernelGGL((attn_kernel<8, false>), dim3(cdiv(B, 4)), dim3(256), 0, stream, B, y2, scale2, shift2, Wa, ba,
                       Wb, bb, out, ld_out, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  MG_LAUNCH_CHECK("scene_attention_fwd");
  return MGGAN_OK;
}

int mggan_scene_attention_bwd(const float* y2, int B, int C, const float* scale2, const float* shift2,
                              const float* stat2, const float* Wa, const float* ba, const float* Wb, const float* bb,
                              const float* dout, int ld_dout, float* ds, float* hact, float* dz, float* vsave,
                              float* G2, float* part, hipStream_t stream) {
  MG_CHECK_ARG(C == 8 || C == 16, "scene_attention_bwd: channels %d not built (8 or 16)", C);
  if (B == 0) return MGGAN_OK;
  MG_CHECK_ARG(y2 && scale2 && shift2 && stat2 && Wa && ba && Wb && bb && dout && ds && hact && dz && vsave && G2 && part,
               "scene_attention_bwd: null pointer");
  if (C == 16)
    hipLaunchKernelGGL((attn_kernel<16, true>), dim3(cdiv(B, 4)), dim3(256), 0, stream, B, y2, scale2, shift2, Wa, ba,
                       Wb, bb, nullptr, 0, dout, ld_dout, stat2, ds, hact, dz, vsave, G2, part);
  else
    hipLaunchKernelGGL((attn_kernel<8, true>), dim3(cdiv(B, 4)), dim3(256), 0, stream, B, y2, scale2, shift2, Wa, ba,
                       Wb, bb, nullptr, 0, dout, ld_dout, stat2, ds, hact, dz, vsave, G2, part);
  MG_LAUNCH_CHECK("scene_attention_bwd");
  return MGGAN_OK;
}

/* workspace: mggan_cnn_bwd_grid(B) * (256/(C*C)) * (C*C*9 + C) floats */
int mggan_conv2_bwd(const float* y1, int B, int C, const float* scale1, const float* shift1, const float* stat1,
                    const float* y2, const float* G2, const float* stat2, const float* coef2, const float* W,
                    float* G1c, unsigned char* code1, float* part1, float* dW, float* db, float* workspace,
                    size_t workspace_bytes, hipStream_t stream) {
  MG_CHECK_ARG(C == 8 || C == 16, "conv2_bwd: channels %d not built (8 or 16)", C);
  if (B == 0) return MGGAN_OK;
  MG_CHECK_ARG(y1 && scale1 && shift1 && stat1 && y2 && G2 && stat2 && coef2 && W && G1c && code1 && part1 && workspace,
               "conv2_bwd: null pointer");
  const int grid = persistent_grid(B), NQ = 256 / (C * C), wlen = C * C * 9 + C;
  const size_t need = (size_t)grid * NQ * wlen * sizeof(float);
  if (workspace_bytes < need) {
    mggan_set_error("conv2_bwd: workspace too small (%zu < %zu)", workspace_bytes, need);
    return MGGAN_ERR_WORKSPACE;
  }
  if (C == 16) {
    const size_t lds = (size_t)(2 * 16 * C2_PLANE + 16 * 256 + 9 * 256 + 128) * sizeof(float);
    static bool attr = false;
    if (!attr) {
      hipFuncSetAttribute((const void*)conv2_bwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr = true;
    }
    hipLaunchKernelGGL(conv2_bwd_mfma_kernel, dim3(grid), dim3(256), lds, stream, B, y1, scale1, shift1, stat1, y2, G2,
                       stat2, coef2, W, G1c, code1, part1, workspace);
  } else
    hipLaunchKernelGGL((conv2_bwd_kernel<8>), dim3(grid), dim3(256), 0, stream, B, y1, scale1, shift1, stat1, y2, G2,
                       stat2, coef2, W, G1c, code1, part1, workspace);
  MG_LAUNCH_CHECK("conv2_bwd");
  if (!dW) return MGGAN_OK;  // deferred reduce of the [grid][C*C*9 + C] partial rows
  hipLaunchKernelGGL(partial_sum_kernel, dim3(cdiv(C * C * 9, 64)), dim3(1024), 0, stream, workspace, grid, wlen,
                     C * C * 9, dW);
  hipLaunchKernelGGL(partial_sum_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, stream, workspace + C * C * 9, grid, wlen,
                     C, db);
  MG_LAUNCH_CHECK("conv2_bwd reduce");
  return MGGAN_OK;
}

/* workspace: mggan_cnn_bwd_grid(B) * 8 * (4*C*9 + C) floats */
int mggan_conv1_bwd(const float* img, int B, int C, const float* y1, const float* stat1, const float* coef1,
                    const float* G1c, const unsigned char* code1, float* dW, float* db, float* workspace,
                    size_t workspace_bytes, hipStream_t stream) {
  MG_CHECK_ARG(C == 8 || C == 16, "conv1_bwd: channels %d not built (8 or 16)", C);
  if (B == 0) return MGGAN_OK;
  MG_CHECK_ARG(img && y1 && stat1 && coef1 && G1c && code1 && workspace, "conv1_bwd: null pointer");
  const int grid = persistent_grid(B), NQ = 8, wlen = 4 * C * 9 + C;
  const size_t need = (size_t)grid * NQ * wlen * sizeof(float);
  if (workspace_bytes < need) {
    mggan_set_error("conv1_bwd: workspace too small (%zu < %zu)", workspace_bytes, need);
    return MGGAN_ERR_WORKSPACE;
  }
  const size_t lds = (size_t)(4 * IMG_PLANE + 8 * IH * 36) * sizeof(float);
  if (C == 16) {
    // (MFMA variants of this kernel were measured twice and lost: dW as a 16 x 36 x 1089 implicit GEMM with the
    //  image in two row halves, 131 us; the flat-position form of conv1_fwd_mfma with all 16 channels of dy1 in a
    //  78 KB LDS tile, one workgroup per CU and the next image's operands prefetched under the products, 134 us
    //  (C = 8: 102 us) -- against 96 us (65 us) for this VALU kernel)
    static bool attr16 = false;
    if (!attr16) {
      hipFuncSetAttribute((const void*)conv1_bwd_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr16 = true;
    }
    hipLaunchKernelGGL((conv1_bwd_kernel<16>), dim3(grid), dim3(256), lds, stream, B, img, y1, stat1, coef1, G1c, code1,
                       workspace);
  } else {
    static bool attr8 = false;
    if (!attr8) {
      hipFuncSetAttribute((const void*)conv1_bwd_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr8 = true;
    }
    hipLaunchKernelGGL((conv1_bwd_kernel<8>), dim3(grid), dim3(256), lds, stream, B, img, y1, stat1, coef1, G1c, code1,
                       workspace);
  }
  MG_LAUNCH_CHECK("conv1_bwd");
  if (!dW) return MGGAN_OK;  // deferred reduce of the [grid][4*C*9 + C] partial rows
  hipLaunchKernelGGL(partial_sum_kernel, dim3(cdiv(4 * C * 9, 64)), dim3(1024), 0, stream, workspace, grid, wlen,
                     4 * C * 9, dW);
  hipLaunchKernelGGL(partial_sum_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, stream, workspace + 4 * C * 9, grid, wlen,
                     C, db);
  MG_LAUNCH_CHECK("conv1_bwd reduce");
  return MGGAN_OK;
}

}  // extern "C"
