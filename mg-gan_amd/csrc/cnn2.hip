// Scene CNN, second generation: no full-resolution activation ever touches HBM.
//
// Replaces (file:line under /root/reference/mggan/model/modules/cnn.py) Conv_Blocks :119-160 / CNN.forward :275-282 for
// the first block and the weight gradient of its convolution:
//
//   conv1_pool_kernel   img (B,4,33,33) -> conv1 on the matrix cores, BatchNorm statistics, and the 2x2 pooling
//                       decision on the RAW output: ReLU(scale*x+shift) is monotone in x, so maxpool(ReLU(BN(x))) is
//                       ReLU(BN(x_max)) for scale >= 0 and ReLU(BN(x_min)) for scale < 0, and scale = gamma/sigma has the
//                       sign of the PARAMETER gamma -- known before any statistics exist.  The kernel keeps that raw
//                       extreme of every window (+ its 2-bit position).  What the first-generation kernels wrote and
//                       re-read three times per pass, the raw (B,C,33,36) output (76 KB per image), never exists;
//                       20 KB per image are kept instead.
//   conv2_fwd2_kernel   a1 = ReLU(scale1*x_sel+shift1) -> conv2 -> raw (B,C,16,16) + statistics.
//   image_gram_*        P[s][t] = sum over images and positions of patch[s]*patch[t] (37x37, tap 36 = the constant 1):
//                       the only DENSE quantity the conv1 weight gradient needs -- and it depends on the images alone,
//                       so ONE launch per batch serves every backward pass of both CNNs (three per iteration).
//                       image_gram_ac_kernel (default) takes it from the images' autocorrelation on the vector ALU
//                       (272 k products per image); image_gram_kernel is the tap-by-tap MFMA form (2.1 M).
//   conv1_wgrad_kernel  A[c][t] = sum_pos dy_sparse[c][pos]*patch[pos][t] on the matrix cores (the gradient that
//                       reaches a pooled cell sits on one of the four window positions).
//   conv1_wgrad_finalize  dW = (gamma/sigma) * (A - m1*B - m2*Chat), Chat[c][t] = (sum_s W[c][s] P[s][t] + (bias-mean) B[t]) / sigma
//                       -- the BatchNorm adjoint dx = (gamma/sigma)(dy - mean(dy) - xhat*mean(dy*xhat)) pushed through
//                       the (linear) convolution, in f64: the two coherent sums that cancel in dW (dW is orthogonal
//                       to W) never meet in f32.
// Every kernel is persistent (<= 768 workgroups walk the images), leaves ONE partial row per workgroup, and -- when
// given a ticket word -- lets the last workgroup to finish fold the rows in index order (f64) and do the BatchNorm
// bookkeeping itself: no separate reduce / finalize launches on the single-GPU path.
#include <stdlib.h>
#include "common.h"
#include "comm_dev.h"
#include "../../include/mggan_hip.h"

#define IH 33
#define IPIX (IH * IH)
#define ILD 40               // row stride of the zero-haloed 35x35 image in LDS
#define IPLANE 1424          // plane stride, == 16 (mod 32): the two input channels a half-wave reads hit disjoint banks
#define IZERO (4 * IPLANE)   // 64 zero floats behind the planes (operand of masked lanes)
#define ILDS (4 * IPLANE + 64)
#define A1_LD 20
#define A1_PLANE (18 * A1_LD)
#define NTAP 37              // 36 taps (ci,ky,kx) + the constant 1

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct BnFin {  // fused BatchNorm finalize (ticket == NULL: the caller reduces / finalizes itself, e.g. across ranks)
  unsigned* ticket;
  double count;
  const float* gamma;
  const float* beta;
  float* run_mean;
  float* run_var;
  long long* nbt;
  float momentum, eps;
  int updates;
  float* scale;
  float* shift;
  float* stat;
  const CommArgs* comm;  // sharded training: the last workgroup also exchanges the folded sums (+ its element count) with
                         // the other ranks through the peer-mapped arenas of this channel (device copy of the CommArgs,
                         // mggan_comm_channel_create) before it finalizes -- NULL: this rank's statistics
};

struct BnBwdFin {  // fused BatchNorm-backward finalize
  unsigned* ticket;
  double count;
  const float* gamma;
  const float* stat;   // [mean | invstd] of the forward pass
  float* coef;         // [gamma*invstd | mean(g) | mean(g*xhat)]  (f32, per-element use)
  double* coefd;       // [gamma*invstd | S1 | S2 | mean | invstd] + count  (f64, for conv1_wgrad_finalize); may be NULL
  float* dgamma;
  float* dbeta;
  const CommArgs* comm;  // as in BnFin: coefficients from the GLOBAL sums, dgamma / dbeta take this rank's share
};

// (4,33,33) image -> interior of the zero-haloed planes (the halo is cleared once per workgroup), in two halves: the
// global loads of the NEXT image are issued before the products of the current one and land in LDS after them
#define IMG_PER ((4 * IPIX + 255) / 256)
__device__ __forceinline__ void fetch_image(const float* __restrict__ src, float v[IMG_PER]) {
#pragma unroll
  for (int u = 0; u < IMG_PER; ++u) {
    const int e = threadIdx.x + 256 * u;
    v[u] = src[e < 4 * IPIX ? e : 0];
  }
}
template <int LD, int PLANE>
__device__ __forceinline__ void commit_image(const float v[IMG_PER], float* imgp) {
#pragma unroll
  for (int u = 0; u < IMG_PER; ++u) {
    const int e = threadIdx.x + 256 * u;
    if (e < 4 * IPIX) {
      const int ci = e / IPIX, rem = e - ci * IPIX, y = rem / IH, x = rem - y * IH;
      imgp[ci * PLANE + (y + 1) * LD + x + 1] = v[u];
    }
  }
}
// Second LDS layout, for the kernels whose 16 lanes of an operand walk the 36 TAPS (Gram matrix, conv1 weight
// gradient): row stride 35 (== 3 mod 32) and plane stride 35*35 = 1225 (== 9 mod 32) put tap t = 9*ci + 3*ky + kx
// on bank t (mod 32): 16 consecutive taps -> 16 distinct banks, and a second lane group 16 columns to the right
// takes the other 16 banks.
#define GLD 35
#define GPLANE 1225
#define GZERO (4 * GPLANE)
#define GLDS (4 * GPLANE + 64)

// Partial rows travel between workgroups as agent-scope 8-byte atomics (write-through to the coherence point on the
// producer side, L1-bypassing loads on the consumer side): no release fence, i.e. no write-back of the whole XCD L2
// per workgroup.  -> true for the workgroup that arrives last (every row is then readable through load_part).
__device__ __forceinline__ void store_part(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ double load_part(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// fixed-order f64 column sums of `rows` partial rows of width W (<= 32) by one 256-thread workgroup -> colsum[W] (LDS).
// 8 (W > 16) or 16 row groups; every lane keeps 8 independent loads in flight (a lone workgroup has to hide the
// memory latency by itself) and adds them in a fixed order.
__device__ __forceinline__ void colsum_rows(const double* part, int rows, int W, double* colsum, double* red /*[8*32]*/) {
  const int wp = W > 16 ? 32 : 16, ng = 256 / wp;
  const int col = threadIdx.x % wp, rg = threadIdx.x / wp;
  double acc = 0.0;
  if (col < W) {
    int r = rg;
    for (; r + 7 * ng < rows; r += 8 * ng) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = load_part(part + (size_t)(r + u * ng) * W + col);
      acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; r < rows; r += ng) acc += load_part(part + (size_t)r * W + col);
  }
  red[rg * wp + col] = acc;
  __syncthreads();
  if ((int)threadIdx.x < W) {
    double t = 0.0;
    for (int i = 0; i < ng; ++i) t += red[i * wp + threadIdx.x];
    colsum[threadIdx.x] = t;
  }
  __syncthreads();
}

__device__ __forceinline__ bool last_block(unsigned* ticket, int* flag_lds) {
  __syncthreads();  // every lane's store_part has completed (each waited for its own)
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(ticket, 1u);
    *flag_lds = (t == gridDim.x - 1);
    if (t == gridDim.x - 1) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed
  }
  __syncthreads();
  return *flag_lds != 0;
}

__device__ __forceinline__ void bn_finalize_lane(const BnFin& f, int C, const double* sums) {
  const int c = threadIdx.x;
  if (c >= C) return;
  const double m = sums[c] / f.count;
  double v = sums[C + c] / f.count - m * m;
  if (v < 0.0) v = 0.0;
  const float mean = (float)m, var = (float)v;
  const float unb = (float)(f.count > 1.0 ? v * f.count / (f.count - 1.0) : v);
  float rm = f.run_mean[c], rv = f.run_var[c];
  for (int u = 0; u < f.updates; ++u) {  // this forward may stand for several identical reference forwards (A.8)
    rm = (1.f - f.momentum) * rm + f.momentum * mean;
    rv = (1.f - f.momentum) * rv + f.momentum * unb;
  }
  f.run_mean[c] = rm;
  f.run_var[c] = rv;
  if (c == 0) *f.nbt += f.updates;
  const float invstd = 1.0f / sqrtf(var + f.eps);
  const float sc = f.gamma[c] * invstd;
  f.scale[c] = sc;
  f.shift[c] = f.beta[c] - mean * sc;
  f.stat[c] = mean;
  f.stat[C + c] = invstd;
}

__device__ __forceinline__ void bn_bwd_finalize_lane(const BnBwdFin& f, int C, const double* sums) {
  const int c = threadIdx.x;
  if (c >= C) return;
  const float cs = f.gamma[c] * f.stat[C + c];
  f.coef[c] = cs;
  f.coef[C + c] = (float)(sums[c] / f.count);
  f.coef[2 * C + c] = (float)(sums[C + c] / f.count);
  if (f.coefd) {
    f.coefd[c] = (double)f.gamma[c] * (double)f.stat[C + c];
    f.coefd[C + c] = sums[c];
    f.coefd[2 * C + c] = sums[C + c];
    f.coefd[3 * C + c] = (double)f.stat[c];
    f.coefd[4 * C + c] = (double)f.stat[C + c];
    if (c == 0) f.coefd[5 * C] = f.count;
  }
  f.dbeta[c] += (float)sums[c];
  f.dgamma[c] += (float)sums[C + c];
}

// The fused finalize of a launch's LAST workgroup, with the exchange of sharded training folded in (every thread of the
// workgroup calls it; sums: >= 2C + 1 doubles of LDS holding the folded column sums).  One launch less per BatchNorm
// exchange point than fold -> mggan_bn_sync_finalize -> consumer, and the wait for the peers' flags happens where the
// producer's grid has already drained.
__device__ __forceinline__ void bn_finalize_block(BnFin fin, int C, double* sums) {
  if (fin.comm) {
    if (threadIdx.x == 0) sums[2 * C] = fin.count;
    comm_allreduce_small(*fin.comm, sums, 2 * C + 1);
    fin.count = sums[2 * C];
  }
  bn_finalize_lane(fin, C, sums);
}
__device__ __forceinline__ void bn_bwd_finalize_block(BnBwdFin fin, int C, double* sums, double* local /* 2C doubles of LDS */) {
  if (!fin.comm) {
    bn_bwd_finalize_lane(fin, C, sums);
    return;
  }
  if ((int)threadIdx.x < 2 * C) local[threadIdx.x] = sums[threadIdx.x];
  if (threadIdx.x == 0) sums[2 * C] = fin.count;
  comm_allreduce_small(*fin.comm, sums, 2 * C + 1);
  fin.count = sums[2 * C];
  const int c = threadIdx.x;
  if (c >= C) return;
  const float keep_b = fin.dbeta[c], keep_g = fin.dgamma[c];
  bn_bwd_finalize_lane(fin, C, sums);
  fin.dbeta[c] = keep_b + (float)local[c];
  fin.dgamma[c] = keep_g + (float)local[C + c];
}

// BatchNorm-1 statistics of conv1(W, bias) over the batch whose Gram matrix of image patches is `gram` (sharded training:
// the GLOBAL batch's):  sum x_c = W_c . B + n b_c,  sum x_c^2 = W_c P W_c^T + 2 b_c W_c . B + n b_c^2
// (B = row 36 of P, n = P[36][36]).  f64 throughout: the variance is a difference of two such sums.  Every thread of the
// workgroup calls it; sums: 2C doubles of LDS.
__device__ __forceinline__ void bn1_from_gram_block(const double* __restrict__ gram, int C, const float* __restrict__ W,
                                                    const float* __restrict__ bias, BnFin fin, double* sums,
                                                    double* rs /* C * 36 doubles of LDS */) {
  const int c = threadIdx.x;
  const double n = gram[36 * NTAP + 36];
  // (W_c P)_t for every (channel, tap) on all threads first -- as ONE thread per channel walking its 36 x 36 products this was
  // 25 us at the end of the launch's first workgroup (conv1_pool_kernel<8>: 45 -> 73 us in sharded training); the sums over
  // the taps stay sequential per channel, in the same order: the same bits
  for (int i = threadIdx.x; i < C * 36; i += blockDim.x) {
    const int ch = i / 36, t = i - ch * 36;
    double r = 0.0;
    for (int u = 0; u < 36; ++u) r += (double)W[ch * 36 + u] * gram[u * NTAP + t];
    rs[i] = r;
  }
  __syncthreads();
  if (c < C) {
    double s = 0.0, q = 0.0;
    for (int t = 0; t < 36; ++t) {
      const double wt = (double)W[c * 36 + t];
      s += wt * gram[36 * NTAP + t];
      q += wt * rs[c * 36 + t];
    }
    const double b = (double)bias[c];
    sums[c] = s + n * b;
    sums[C + c] = q + 2.0 * b * s + n * b * b;
  }
  __syncthreads();
  fin.count = n;
  bn_finalize_lane(fin, C, sums);
}

// ------------------------------------------------------------------------------------------------------------------
// conv1 + statistics + pooling decision.  Implicit GEMM on v_mfma_f32_16x16x4_f32: a tile is FOUR pooling windows,
// row i = 4*window + element of the A operand, so that the D fragment of a lane (4 rows of one output channel) is
// exactly one 2x2 window: max / min / argmax are lane-local.  K = 36 is ordered k = 4*tap + ci: the four lane groups
// of a k-step read four different input-channel planes.
template <int C>
__global__ __launch_bounds__(256) void conv1_pool_kernel(int B, const float* __restrict__ img, const float* __restrict__ W,
                                                         const float* __restrict__ bias, const float* __restrict__ gamma,
                                                         float* __restrict__ xsel, unsigned char* __restrict__ code,
                                                         double* part, BnFin fin, const double* __restrict__ gram,
                                                         const int* dims) {
  MG_REAL_IMAGES_COUNT(B, dims, fin)
  __shared__ __attribute__((aligned(16))) float imgp[ILDS];
  __shared__ double redd[4][2][16];
  __shared__ double colsum[40], cred[8 * 32];
  __shared__ int flag;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fi = lane & 15, fk = lane >> 4;
  for (int i = tid; i < ILDS; i += 256) imgp[i] = 0.f;
  float bw[9];
#pragma unroll
  for (int s = 0; s < 9; ++s) bw[s] = fi < C ? W[fi * 36 + fk * 9 + s] : 0.f;
  const float bv = fi < C ? bias[fi] : 0.f;
  // the BatchNorm scale gamma/sigma has the sign of gamma: this lane's channel keeps the window MAXIMUM (gamma >= 0)
  // or MINIMUM (gamma < 0) of the raw output -- sgn * max(sgn * x)
  const float sgn = (fi < C && gamma[fi] < 0.f) ? -1.f : 1.f;
  // A operand: lane (fi, fk) supplies row i = fi (window fi>>2, element fi&3) of input channel fk
  const int aoff = fk * IPLANE + ((fi & 3) >> 1) * ILD + 2 * (fi >> 2) + (fi & 1);
  double dsum = 0.0, dsq = 0.0;
  float pre[IMG_PER];
  if ((int)blockIdx.x < B) fetch_image(img + (size_t)blockIdx.x * 4 * IPIX, pre);
  __syncthreads();
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();  // the previous image's reads are done
    commit_image<ILD, IPLANE>(pre, imgp);
    __syncthreads();
    if (b + (int)gridDim.x < B) fetch_image(img + (size_t)(b + gridDim.x) * 4 * IPIX, pre);  // in flight under the products
    float sum = 0.f, sq = 0.f;
    // Wave w owns the pooled rows py = w, w + 4, w + 8, w + 12; a row is four tiles (px0 = 0, 4, 8, 12), two at a time
    // through the matrix pipe.  A lane then holds window px0 + fk of each tile: a 4 x 4 transposition between the tile
    // index and the lane group (four permlane swaps per value) turns that into px = 4 fk .. 4 fk + 3, i.e. ONE 16-byte
    // store of the selected values and one 4-byte store of the four position codes per lane and row, instead of four
    // 4-byte and four 1-byte stores scattered over 16 channel planes (the stores were 35 of 159 us at 8,192 images).
#pragma unroll 1
    for (int py = w; py < 16; py += 4) {
      float sel[4];
      unsigned cod[4];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int px0 = 8 * half, px1 = 8 * half + 4;
        const float* p0 = imgp + 2 * py * ILD + 2 * px0 + aoff;
        const float* p1 = imgp + 2 * py * ILD + 2 * px1 + aoff;
        f32x4 a0 = {bv, bv, bv, bv}, a1 = {bv, bv, bv, bv};
        float x0[9], x1[9];
#pragma unroll
        for (int s = 0; s < 9; ++s) {
          x0[s] = p0[(s / 3) * ILD + s % 3];
          x1[s] = p1[(s / 3) * ILD + s % 3];
        }
#pragma unroll
        for (int s = 0; s < 9; ++s) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[s], bw[s], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[s], bw[s], a1, 0, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 acc = h ? a1 : a0;
          const float s0 = sgn * acc[0], s1 = sgn * acc[1], s2 = sgn * acc[2], s3 = sgn * acc[3];
          const float mx = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
          const int cx = s0 == mx ? 0 : (s1 == mx ? 1 : (s2 == mx ? 2 : 3));  // the first position that attains it
          sum += (acc[0] + acc[1]) + (acc[2] + acc[3]);
          sq = fmaf(acc[0], acc[0], fmaf(acc[1], acc[1], fmaf(acc[2], acc[2], fmaf(acc[3], acc[3], sq))));
          sel[2 * half + h] = sgn * mx;
          cod[2 * half + h] = (unsigned)cx;
        }
      }
      // tile j, lane group fk  ->  lane group j, element fk
      const mg_u2 sa = __builtin_amdgcn_permlane16_swap(__float_as_uint(sel[0]), __float_as_uint(sel[1]), false, false);
      const mg_u2 sb = __builtin_amdgcn_permlane16_swap(__float_as_uint(sel[2]), __float_as_uint(sel[3]), false, false);
      const mg_u2 s02 = __builtin_amdgcn_permlane32_swap(sa[0], sb[0], false, false);
      const mg_u2 s13 = __builtin_amdgcn_permlane32_swap(sa[1], sb[1], false, false);
      const mg_u2 ca = __builtin_amdgcn_permlane16_swap(cod[0], cod[1], false, false);
      const mg_u2 cb = __builtin_amdgcn_permlane16_swap(cod[2], cod[3], false, false);
      const mg_u2 c02 = __builtin_amdgcn_permlane32_swap(ca[0], cb[0], false, false);
      const mg_u2 c13 = __builtin_amdgcn_permlane32_swap(ca[1], cb[1], false, false);
      if (fi < C) {
        const size_t o = (((size_t)b * C + fi) * 16 + py) * 16 + 4 * fk;
        *reinterpret_cast<f32x4*>(xsel + o) = f32x4{__uint_as_float(s02[0]), __uint_as_float(s13[0]), __uint_as_float(s02[1]),
                                                    __uint_as_float(s13[1])};
        *reinterpret_cast<unsigned*>(code + o) = c02[0] | (c13[0] << 8) | (c02[1] << 16) | (c13[1] << 24);
      }
    }
    // border row 32 / column 32 (never pooled): 65 positions, statistics only
    for (int bt = w; bt < 5; bt += 4) {
      const int i = 16 * bt + fi;
      const int y = i < 33 ? 32 : (i < 65 ? i - 33 : 0), x = i < 33 ? i : (i < 65 ? 32 : 0);
      const float* p = imgp + fk * IPLANE + y * ILD + x;
      f32x4 a = {bv, bv, bv, bv};
#pragma unroll
      for (int s = 0; s < 9; ++s) a = __builtin_amdgcn_mfma_f32_16x16x4f32(p[(s / 3) * ILD + s % 3], bw[s], a, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * bt + 4 * fk + r < 65) {
          sum += a[r];
          sq = fmaf(a[r], a[r], sq);
        }
    }
    dsum += (double)sum;  // f32 within an image, f64 across the images of this workgroup
    dsq += (double)sq;
  }
  // lanes with the same fi (4 lane groups), then the 4 waves
  dsum += __shfl_xor(dsum, 16, 64); dsq += __shfl_xor(dsq, 16, 64);
  dsum += __shfl_xor(dsum, 32, 64); dsq += __shfl_xor(dsq, 32, 64);
  if (fk == 0) { redd[w][0][fi] = dsum; redd[w][1][fi] = dsq; }
  __syncthreads();
  if (tid < 2 * C) {
    const int which = tid / C, c = tid - which * C;
    store_part(part + (size_t)blockIdx.x * 2 * C + tid,
               (redd[0][which][c] + redd[1][which][c]) + (redd[2][which][c] + redd[3][which][c]));
  }
  if (gram) {
    // sharded training: the GLOBAL statistics of this layer follow from the global Gram matrix and the weights alone
    // (nothing this launch computed enters them): workgroup 0 writes scale / shift / running statistics on its way out
    if (blockIdx.x == 0) bn1_from_gram_block(gram, C, W, bias, fin, colsum, reinterpret_cast<double*>(imgp));  // (the image
    //                                                                       tile is free: every wave is behind the barrier above)
    return;
  }
  if (!fin.ticket) return;
  if (!last_block(fin.ticket, &flag)) return;
  colsum_rows(part, gridDim.x, 2 * C, colsum, cred);
  bn_finalize_block(fin, C, colsum);
}

// ------------------------------------------------------------------------------------------------------------------
// conv2 forward: a1 = ReLU(BN1(x_sel)) with x_sel the window maximum (scale >= 0) or minimum (scale < 0)
template <int C>
__global__ __launch_bounds__(256) void conv2_fwd2_kernel(int B, const float* __restrict__ xsel,
                                                         const float* __restrict__ scale1, const float* __restrict__ shift1,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         float* __restrict__ y2, double* part, BnFin fin,
                                                         const int* dims) {
  MG_REAL_IMAGES_COUNT(B, dims, fin)
  constexpr int COT = C / 4;
  __shared__ __attribute__((aligned(16))) float a1p[C * A1_PLANE];
  __shared__ double colsum[40], cred[8 * 32];
  __shared__ int flag;
  for (int i = threadIdx.x; i < C * A1_PLANE; i += 256) a1p[i] = 0.f;
  const int cg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), pg = threadIdx.x & 63;
  // lane -> (row py, column quad x0): the 16 lanes of a ds_read_b128 phase are 16 ROWS of one quad - row stride 20 puts
  // them on sixteen distinct 4-bank groups (20 py mod 64); four rows x four quads per phase (the first version) wrapped the
  // fourth row onto the first: 2-way conflicts on every read, 64 % of the LDS cycles of this kernel at 8,192 images
  const int py = pg & 15, x0 = (pg >> 4) * 4;
  double dsum[COT], dsq[COT];
#pragma unroll
  for (int co = 0; co < COT; ++co) dsum[co] = dsq[co] = 0.0;
  float v[C];
  auto fetch = [&](int b) {
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = xsel[((size_t)b * C + c) * 256 + threadIdx.x];
  };
  if ((int)blockIdx.x < B) fetch(blockIdx.x);
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    {
      const int ppy = threadIdx.x >> 4, ppx = threadIdx.x & 15;
#pragma unroll
      for (int c = 0; c < C; ++c)
        a1p[c * A1_PLANE + (ppy + 1) * A1_LD + ppx + 1] = fmaxf(fmaf(v[c], scale1[c], shift1[c]), 0.f);
    }
    __syncthreads();
    if (b + (int)gridDim.x < B) fetch(b + gridDim.x);  // the next image's inputs fly under this image's products
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
        const float* rp = &a1p[ci * A1_PLANE + (py + ky) * A1_LD + x0];
        const float4 ra = *reinterpret_cast<const float4*>(rp);
        const float2 rb = *reinterpret_cast<const float2*>(rp + 4);
        const float r[6] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int co = 0; co < COT; ++co) {
            const float wv = W[((cg * COT + co) * C + ci) * 9 + ky * 3 + kx];
#pragma unroll
            for (int px = 0; px < 4; ++px) acc[co][px] = fmaf(wv, r[px + kx], acc[co][px]);
          }
      }
#pragma unroll
    for (int co = 0; co < COT; ++co) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int px = 0; px < 4; ++px) { s += acc[co][px]; q = fmaf(acc[co][px], acc[co][px], q); }
      *reinterpret_cast<float4*>(y2 + (((size_t)b * C + cg * COT + co) * 16 + py) * 16 + x0) =
          make_float4(acc[co][0], acc[co][1], acc[co][2], acc[co][3]);
      dsum[co] += (double)wave_sum(s);
      dsq[co] += (double)wave_sum(q);
    }
  }
  if (pg == 0) {
#pragma unroll
    for (int co = 0; co < COT; ++co) {
      store_part(part + (size_t)blockIdx.x * 2 * C + cg * COT + co, dsum[co]);
      store_part(part + (size_t)blockIdx.x * 2 * C + C + cg * COT + co, dsq[co]);
    }
  }
  if (!fin.ticket) return;
  if (!last_block(fin.ticket, &flag)) return;
  colsum_rows(part, gridDim.x, 2 * C, colsum, cred);
  bn_finalize_block(fin, C, colsum);
}

// ------------------------------------------------------------------------------------------------------------------
// conv2 forward on the matrix cores (implicit GEMM, exact-f32 16x16x4 MFMA): per image row y,
//     y2[co][y][x] = b[co] + sum_{tap, ci} a1[ci][y + ky][x + kx] W[co][ci][tap]        M = 16 x, N = co, K = (tap, ci)
// The A operand walks the zero-haloed a1 planes in LDS; its reduction index is ordered so that the two lane groups of a
// 32-lane LDS pass take planes whose distance is 16 banks (C = 16: planes 386 apart, ci = c + 8 (k & 1) + 4 (k >> 1);
// C = 8: planes 388 apart, ci = c + 4 (k & 1) + 2 (k >> 1)) -- the layout rule of conv2_bwd_mfma_kernel (cnn.hip) --; the
// B operand (the weights of a k step) is loop-invariant: 9 C / 4 registers per lane, loaded once per workgroup.  Wave w
// owns image rows 4 w .. 4 w + 3 as four independent accumulator chains; the D fragment (x = 4 k + r, co) goes out as
// one 16-byte store per row.  C = 8 fills half of the N tile.  (The register-tiled VALU kernel above needs one scalar
// weight load per four FMAs and ran at 0.38 of the f32 rate at 8,192 images; it stays selectable: MGGAN_CONV2_VALU=1.)
// (three waves per SIMD: at 172 registers two of the three workgroups a CU is dealt were resident -- 133 -> 102 us at
// 8,192 images with C = 16, six spilled registers included)
template <int C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv2_fwd_mfma_kernel(int B, const float* __restrict__ xsel,
                                                             const float* __restrict__ scale1,
                                                             const float* __restrict__ shift1, const float* __restrict__ W,
                                                             const float* __restrict__ bias, float* __restrict__ y2,
                                                             double* part, BnFin fin, const int* dims) {
  MG_REAL_IMAGES_COUNT(B, dims, fin)
  constexpr int PLANE = C == 16 ? 386 : 388, KS = C / 4;
  __shared__ __attribute__((aligned(16))) float a1p[C * PLANE];
  __shared__ double redd[4][2][16];
  __shared__ double colsum[40], cred[8 * 32];
  __shared__ int flag;
  for (int i = threadIdx.x; i < C * PLANE; i += 256) a1p[i] = 0.f;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  // reduction index of k step (tap, c): input channel of this lane's k group
  int cik[KS];
#pragma unroll
  for (int c = 0; c < KS; ++c) cik[c] = C == 16 ? c + 8 * (fk & 1) + 4 * (fk >> 1) : c + 4 * (fk & 1) + 2 * (fk >> 1);
  float wb[9][KS];  // B[k = (tap, ci)][j = co = fi]
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int c = 0; c < KS; ++c) wb[tp][c] = fi < C ? W[(fi * C + cik[c]) * 9 + tp] : 0.f;
  const float bv = fi < C ? bias[fi] : 0.f;
  double dsum = 0.0, dsq = 0.0;  // channel fi, this wave's rows, every image of the workgroup
  float v[C];
  auto fetch = [&](int b) {
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = xsel[((size_t)b * C + c) * 256 + threadIdx.x];
  };
  if ((int)blockIdx.x < B) fetch(blockIdx.x);
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    {
      const int ppy = threadIdx.x >> 4, ppx = threadIdx.x & 15;
#pragma unroll
      for (int c = 0; c < C; ++c)
        a1p[c * PLANE + (ppy + 1) * A1_LD + ppx + 1] = fmaxf(fmaf(v[c], scale1[c], shift1[c]), 0.f);
    }
    __syncthreads();
    if (b + (int)gridDim.x < B) fetch(b + gridDim.x);  // the next image's inputs fly under this image's products
    f32x4 acc[4];
#pragma unroll
    for (int ry = 0; ry < 4; ++ry) acc[ry] = f32x4{bv, bv, bv, bv};
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        const float* ap = &a1p[cik[c] * PLANE + (4 * w + tp / 3) * A1_LD + fi + tp % 3];
#pragma unroll
        for (int ry = 0; ry < 4; ++ry)
          acc[ry] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[ry * A1_LD], wb[tp][c], acc[ry], 0, 0, 0);
      }
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int ry = 0; ry < 4; ++ry) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sm += acc[ry][r];
        sq = fmaf(acc[ry][r], acc[ry][r], sq);
      }
      if (fi < C)
        *reinterpret_cast<f32x4*>(y2 + (((size_t)b * C + fi) * 16 + 4 * w + ry) * 16 + 4 * fk) = acc[ry];
    }
    dsum += (double)quarters_sum(sm);  // f32 within an image, f64 across images
    dsq += (double)quarters_sum(sq);
  }
  if (lane < 16) {
    redd[w][0][lane] = dsum;
    redd[w][1][lane] = dsq;
  }
  __syncthreads();
  if ((int)threadIdx.x < 2 * C) {
    const int which = threadIdx.x / C, c = threadIdx.x % C;
    store_part(part + (size_t)blockIdx.x * 2 * C + threadIdx.x,
               (redd[0][which][c] + redd[1][which][c]) + (redd[2][which][c] + redd[3][which][c]));
  }
  if (!fin.ticket) return;
  if (!last_block(fin.ticket, &flag)) return;
  colsum_rows(part, gridDim.x, 2 * C, colsum, cred);
  bn_finalize_block(fin, C, colsum);
}

// sums[col] = sum over rows of part[row][col]  (f64; the sharded path all-reduces `sums` before finalizing)
__global__ __launch_bounds__(256) void bn_reduce_rows_kernel(const double* part, int rows, int W, double* sums) {
  __shared__ double colsum[32], cred[8 * 32];
  colsum_rows(part, rows, W, colsum, cred);
  if ((int)threadIdx.x < W) sums[threadIdx.x] = colsum[threadIdx.x];
}

__global__ __launch_bounds__(256) void bn_bwd_rows_finalize_kernel(const double* part, int rows, int C, BnBwdFin fin) {
  __shared__ double colsum[32], cred[8 * 32];
  colsum_rows(part, rows, 2 * C, colsum, cred);
  bn_bwd_finalize_lane(fin, C, colsum);
}

// coefficient block from already reduced (and, sharded, all-reduced) sums; dgamma / dbeta take the LOCAL sums
__global__ void bn_bwd_coef_kernel(const double* sums, const double* local, int C, BnBwdFin fin) {
  __shared__ double s[32];
  if ((int)threadIdx.x < 2 * C) s[threadIdx.x] = sums[threadIdx.x];
  __syncthreads();
  const int c = threadIdx.x;
  if (c >= C) return;
  const float keep_b = fin.dbeta[c], keep_g = fin.dgamma[c];
  bn_bwd_finalize_lane(fin, C, s);
  fin.dbeta[c] = keep_b + (float)local[c];
  fin.dgamma[c] = keep_g + (float)local[C + c];
}

// Sharded training: fold this rank's partial rows, exchange the 2C sums (+ the rank's element count) with the other ranks
// through the peer-mapped arenas, finalize with the GLOBAL statistics -- one launch per BatchNorm exchange point.
__global__ __launch_bounds__(256) void bn_sync_finalize_kernel(CommArgs ca, const double* part, int rows, int C,
                                                               double local_count, BnFin fin) {
  __shared__ double colsum[64], cred[8 * 32];
  colsum_rows(part, rows, 2 * C, colsum, cred);
  if (threadIdx.x == 0) colsum[2 * C] = local_count;
  comm_allreduce_small(ca, colsum, 2 * C + 1);
  fin.count = colsum[2 * C];
  bn_finalize_lane(fin, C, colsum);
}

__global__ __launch_bounds__(256) void bn_bwd_sync_finalize_kernel(CommArgs ca, const double* part, int rows, int C,
                                                                   double local_count, BnBwdFin fin) {
  __shared__ double colsum[64], local[64], cred[8 * 32];
  colsum_rows(part, rows, 2 * C, colsum, cred);
  if ((int)threadIdx.x < 2 * C) local[threadIdx.x] = colsum[threadIdx.x];
  if (threadIdx.x == 0) colsum[2 * C] = local_count;
  comm_allreduce_small(ca, colsum, 2 * C + 1);
  fin.count = colsum[2 * C];
  const int c = threadIdx.x;
  if (c >= C) return;
  // coefficients from the global sums; the parameter gradients take this rank's share (the gradient all-reduce adds)
  const float keep_b = fin.dbeta[c], keep_g = fin.dgamma[c];
  bn_bwd_finalize_lane(fin, C, colsum);
  fin.dbeta[c] = keep_b + (float)local[c];
  fin.dgamma[c] = keep_g + (float)local[C + c];
}

// ... and from the device copy of a channel's CommArgs: what a rank WITHOUT images launches in place of the producer (its
// peers' last workgroups wait for its flags)
__global__ __launch_bounds__(256) void bn_sync_finalize_dev_kernel(const double* part, int rows, int C, BnFin fin) {
  __shared__ double colsum[64], cred[8 * 32];
  colsum_rows(part, rows, 2 * C, colsum, cred);
  bn_finalize_block(fin, C, colsum);
}

// ------------------------------------------------------------------------------------------------------------------
// Gram matrix of the image patches: P[s][t] = sum_{img,pos} patch[pos][s] patch[pos][t], taps ordered t = 9*ci + 3*ky + kx,
// tap 36 = 1.  A and B operand of a 16x16x4 MFMA are the SAME register when both index (tap, position): three LDS
// reads (tap blocks 0-15, 16-31, 32-47) feed the six upper-triangle tile products of a k-step.  The four positions of
// a k-step are (y, x), (y, x+16), (y+1, x), (y+1, x+16): with the tap-per-bank layout the two lane groups of a
// half-wave read disjoint banks (conflict-free ds_read_b32); column 32 is swept by nine extra k-steps per image.
__global__ __launch_bounds__(256) void image_gram_kernel(int B, const float* __restrict__ img, double* part /*[grid][6*256]*/,
                                                         const int* dims) {
  MG_REAL_IMAGES(B, dims)
  __shared__ __attribute__((aligned(16))) float imgp[GLDS];
  __shared__ double fold[4][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fi = lane & 15, fk = lane >> 4;
  for (int i = tid; i < GLDS; i += 256) imgp[i] = 0.f;
  // lane (fi, fk): tap 16*blk + fi at position slot fk
  int toff[3];
  bool real[3], one[3];
#pragma unroll
  for (int blk = 0; blk < 3; ++blk) {
    const int t = 16 * blk + fi;
    real[blk] = t < 36;
    one[blk] = t == 36;
    toff[blk] = t < 36 ? (t / 9) * GPLANE + ((t % 9) / 3) * GLD + (t % 3) : 0;
  }
  const int slot_main = (fk >> 1) * GLD + 16 * (fk & 1);
  double acc[6][4];
#pragma unroll
  for (int q = 0; q < 6; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[q][r] = 0.0;
  float pre[IMG_PER];
  if ((int)blockIdx.x < B) fetch_image(img + (size_t)blockIdx.x * 4 * IPIX, pre);
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    commit_image<GLD, GPLANE>(pre, imgp);
    __syncthreads();
    if (b + (int)gridDim.x < B) fetch_image(img + (size_t)(b + gridDim.x) * 4 * IPIX, pre);
    f32x4 a[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) a[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto kstep = [&](int pos, bool valid) {
      float v[3];
#pragma unroll
      for (int blk = 0; blk < 3; ++blk) {
        v[blk] = imgp[(valid && real[blk]) ? pos + toff[blk] : GZERO];
        if (valid && one[blk]) v[blk] = 1.f;
      }
      a[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[0], v[0], a[0], 0, 0, 0);
      a[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[0], v[1], a[1], 0, 0, 0);
      a[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[0], v[2], a[2], 0, 0, 0);
      a[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[1], v[1], a[3], 0, 0, 0);
      a[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[1], v[2], a[4], 0, 0, 0);
      a[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[2], v[2], a[5], 0, 0, 0);
    };
    // rows (y0, y0+1) x columns (x, x+16): wave w takes x = 4*j + w of every row pair
#pragma unroll 4  // (one row pair per trip left the matrix pipe waiting for its LDS reads: 271 -> 227 us at 8,192 images)
    for (int yp = 0; yp < 17; ++yp) {
      const bool valid = 2 * yp + (fk >> 1) < IH;
      const int rowpos = 2 * yp * GLD + slot_main + w;
#pragma unroll
      for (int j = 0; j < 4; ++j) kstep(rowpos + 4 * j, valid);
    }
    // column 32, rows 4*s + fk
    for (int s9 = w; s9 < 9; s9 += 4) {
      const int y = 4 * s9 + fk;
      kstep(y * GLD + 32, y < IH);
    }
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][r] += (double)a[q][r];  // f32 within an image, f64 across images
  }
  // fold the four waves (fixed order) tile by tile; D fragment: lane holds rows 4*fk + r, column fi
  double* out = part + (size_t)blockIdx.x * 6 * 256;
  for (int q = 0; q < 6; ++q)
    for (int r = 0; r < 4; ++r) {
      __syncthreads();
      fold[w][lane] = acc[q][r];
      __syncthreads();
      if (w == 0) out[q * 256 + (4 * fk + r) * 16 + fi] = (fold[0][lane] + fold[1][lane]) + (fold[2][lane] + fold[3][lane]);
    }
}

// gram[s][t] (37 x 37, symmetric, f64) from the per-workgroup tile partials: 16 outputs x 16 row groups per workgroup,
// 8 loads in flight per lane, fixed order
__global__ __launch_bounds__(256) void image_gram_finalize_kernel(const double* part, int rows, double* gram) {
  __shared__ double red[16][16];
  const int e = blockIdx.x * 16 + (threadIdx.x & 15), rg = threadIdx.x >> 4;
  double t = 0.0;
  int r = rg;
  for (; r + 7 * 16 < rows; r += 8 * 16) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(r + u * 16) * 6 * 256 + e];
    t += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  for (; r < rows; r += 16) t += part[(size_t)r * 6 * 256 + e];
  red[rg][threadIdx.x & 15] = t;
  __syncthreads();
  if (threadIdx.x >= 16) return;
  t = 0.0;
  for (int i = 0; i < 16; ++i) t += red[i][threadIdx.x];
  const int q = e >> 8, i = (e >> 4) & 15, j = e & 15;
  const int bi = q < 3 ? 0 : (q < 5 ? 1 : 2), bj = q < 3 ? q : (q < 5 ? q - 2 : 2);
  const int s = 16 * bi + i, u = 16 * bj + j;
  if (s < NTAP && u < NTAP) {
    gram[s * NTAP + u] = t;
    if (bi != bj) gram[u * NTAP + s] = t;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same Gram matrix from the images' AUTOCORRELATION: patch[pos][(i,k)] = X_i(pos + k) (zero outside the image), so
//   P[(i,k)][(j,l)] = sum over u of X_i(u) X_j(u + l - k), u restricted to the positions pos + k the tap k can reach
// -- a 5x5 table of offsets d = l - k per channel pair, minus the image's first / last row and column where a tap
// looks off the edge.  10 channel pairs (i <= j; the other half by symmetry) x 25 offsets x 1,089 positions = 272 k
// products per image instead of the 37 x 37 x 1,089 = 1.49 M the tap-by-tap product spends (the MFMA kernel above:
// 2.1 M with its padding) -- on the vector ALU, whose f32 rate equals the matrix pipe's.
// A WALKER (one lane) moves along one line of X_i against the five lines around it of X_j with a sliding 5x5 register
// window and keeps 25 sums -- for TWO lines at once, one per half of a packed-f32 FMA: the image sits in LDS as
// float2 (row q, row q + 17), so one ds_read_b64 fetches both halves' operand and the window shift of the fully
// unrolled walk is register renaming (no moves).  Waves 0-2: 10 pairs x 17 double rows walk along x (S per offset and
// row).  Wave 3: 10 pairs x 2 edge columns walk DOWN column 0 / 32 (17 steps of the same float2: rows 0-16 | 17-33):
// the products at an edge column, E0 / E32, summed over the rows.  The four corner products per offset are taken
// directly.  Per workgroup and offset: {all rows, row 0, row 32} of S, E0, E32 -- what the assembly needs to take the
// edge terms out.  f32 within a run of <= 8 images (<= 264 products per sum, the run length of the MFMA kernel's
// accumulators), f64 across runs, workgroups and ranks.
#define AC_LD 37                         // float2 per line (columns -2 .. 34)
#define AC_PLANE (21 * AC_LD)            // float2 per channel: rows q = -2 .. 18 | q + 17
#define AC_LDS2 (4 * AC_PLANE)
#define AC_THREADS 256
#define AC_MAINW 170                     // walkers along x: 10 pairs x 17 double rows (waves 0-2)
#define AC_EDGE0 192                     // walkers down the edge columns: lanes 192 .. 211 (wave 3)
#define AC_RED2 (AC_MAINW * 25)          // reduce buffer (floats; aliases the planes between runs): [main walkers][25]
#define AC_RED3 (AC_RED2 + 20 * 25)      // [edge walkers][25]; then [main walkers][3] (the tap-36 column)
#define AC_MAIN (3 * 10 * 25 * 3)        // [S | E0 | E32][pair][offset][all rows | row 0 | row 32]
#define AC_ROW 2304                      // doubles per workgroup: AC_MAIN + 36 (the tap-36 column) padded to 36 x 64
#define AC_RUN 8
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int ac_pair_i(int p) { return p < 4 ? 0 : (p < 7 ? 1 : (p < 9 ? 2 : 3)); }
__device__ __forceinline__ int ac_pair_j(int p) { return p < 4 ? p : (p < 7 ? p - 3 : (p < 9 ? p - 5 : 3)); }
// X(y, x) of one channel from the double-row planes (y, x in -2 .. 34)
__device__ __forceinline__ float ac_at(const f32x2* plane, int y, int x) {
  return y <= 18 ? plane[(y + 2) * AC_LD + x + 2].x : plane[(y - 15) * AC_LD + x + 2].y;
}
// one walk: STEPS positions LINE float2 apart, the window's other axis PERP apart; acc[perp * 5 + along]
template <int STEPS, int LINE, int PERP>
__device__ __forceinline__ void ac_walk(const f32x2* pa, const f32x2* pw, f32x2 acc[25], f32x2 on[3]) {
  f32x2 w[5][5];
#pragma unroll
  for (int v = 0; v < 5; ++v)
#pragma unroll
    for (int c = 0; c < 5; ++c) w[v][c] = pw[(v - 2) * PERP + (c - 2) * LINE];
#pragma unroll
  for (int x = 0; x < STEPS; ++x) {
    const f32x2 a = pa[x * LINE];
    on[0] += a;
    if (x == 0) on[1] += a;
    if (x == STEPS - 1) on[2] += a;
#pragma unroll
    for (int v = 0; v < 5; ++v)
#pragma unroll
      for (int c = 0; c < 5; ++c) acc[v * 5 + c] = __builtin_elementwise_fma(a, w[v][c], acc[v * 5 + c]);
    if (x < STEPS - 1) {
#pragma unroll
      for (int v = 0; v < 5; ++v) {
#pragma unroll
        for (int c = 0; c < 4; ++c) w[v][c] = w[v][c + 1];
        w[v][4] = pw[(v - 2) * PERP + (x + 3) * LINE];
      }
    }
  }
}

__global__ __launch_bounds__(AC_THREADS, 2) void image_gram_ac_kernel(int B, const float* __restrict__ img,
                                                                      double* part /*[grid][AC_ROW]*/, const int* dims) {
  MG_REAL_IMAGES(B, dims)
  __shared__ f32x2 lds2[AC_LDS2];
  float* lds = (float*)lds2;
  const int tid = threadIdx.x;
  for (int i = tid; i < AC_LDS2; i += AC_THREADS) lds2[i] = f32x2{0.f, 0.f};
  const bool mainw = tid < AC_MAINW, edgew = tid >= AC_EDGE0 && tid < AC_EDGE0 + 20;
  const int p = mainw ? tid / 17 : (edgew ? (tid - AC_EDGE0) >> 1 : 0), q = mainw ? tid - p * 17 : 0;
  const int wbase = mainw ? (q + 2) * AC_LD + 2 : 2 * AC_LD + 2 + (((tid - AC_EDGE0) & 1) ? IH - 1 : 0);
  const f32x2* pa = lds2 + wbase + ac_pair_i(p) * AC_PLANE;
  const f32x2* pw = lds2 + wbase + ac_pair_j(p) * AC_PLANE;
  // loader lane (column lx, line lk of 7): rows q = lk + 7u - 2 (u < 3) | q + 17 of every channel, as float2
  const bool loader = tid < 7 * IH;
  const int lk = loader ? tid / IH : 0, lx = loader ? tid - lk * IH : 0;
  const bool lo_ok = lk >= 2 /* u = 0: rows -2, -1 are halo */, hi_ok = lk <= 3 /* u = 2: rows above 32 are halo */;
  const int src_lo0 = (lk >= 2 ? lk - 2 : 0) * IH + lx;          // half 0, u = 0 (clamped)
  const int src_lo = (lk + 5) * IH + lx;                          // half 0, u = 1 (u = 2: + 7 rows)
  const int src_hi = (lk + 15) * IH + lx;                         // half 1, u = 0 (u = 1: + 7 rows)
  const int src_hi2 = (lk <= 3 ? lk + 29 : IH - 1) * IH + lx;     // half 1, u = 2 (clamped)
  f32x2* dst0 = lds2 + lk * AC_LD + lx + 2;
  // reducer (pair, offset) = lanes 0-249; lanes 0-11 also the tap-36 column (channel, which)
  const bool reducer = tid < 250, reducer1 = tid < 12;
  const int rp = reducer ? tid / 25 : 0, rd = reducer ? tid - rp * 25 : 0, rdT = (rd % 5) * 5 + rd / 5;
  const int o1c = reducer1 ? tid / 3 : 0, o1p = o1c == 0 ? 0 : (o1c == 1 ? 4 : (o1c == 2 ? 7 : 9));
  double* out = part + (size_t)blockIdx.x * AC_ROW;
  bool first = true;
  f32x2 acc[25], on[3];
  float cacc[4];
  auto clear = [&]() {
#pragma unroll
    for (int d = 0; d < 25; ++d) acc[d] = f32x2{0.f, 0.f};
    on[0] = on[1] = on[2] = f32x2{0.f, 0.f};
    cacc[0] = cacc[1] = cacc[2] = cacc[3] = 0.f;
  };
  // the run's f32 sums -> the workgroup's f64 row (its own, in global memory: read-modify-write by the lane that owns the
  // entry), image rows added in index order; the planes are dead here and hold the reduce buffer
  auto flush = [&]() {
    auto add_out = [&](double* o, double v, bool assign) { *o = assign ? v : *o + v; };
    double sall = 0.0, s1all = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // half 0: image rows 0-16; half 1: rows 17-33
      __syncthreads();
      if (mainw || edgew) {
        float* o = lds + (mainw ? tid * 25 : AC_RED2 + (tid - AC_EDGE0) * 25);
#pragma unroll
        for (int d = 0; d < 25; ++d) o[d] = h ? acc[d].y : acc[d].x;
        if (mainw) {
#pragma unroll
          for (int k = 0; k < 3; ++k) lds[AC_RED3 + tid * 3 + k] = h ? on[k].y : on[k].x;
        }
      }
      __syncthreads();
      if (reducer) {
        const float* o = lds + rp * 17 * 25 + rd;
        for (int y = 0; y < (h ? 16 : 17); ++y) sall += (double)o[y * 25];  // (row 33 = walker 16, half 1, is the zero halo)
        double* dst = out + (rp * 25 + rd) * 3;
        if (h == 0) add_out(dst + 1, (double)o[0], first);
        else {
          add_out(dst, sall, first);
          add_out(dst + 2, (double)o[15 * 25], first);
        }
        add_out(dst + 750, (double)lds[AC_RED2 + (rp * 2 + 0) * 25 + rdT], first && h == 0);
        add_out(dst + 1500, (double)lds[AC_RED2 + (rp * 2 + 1) * 25 + rdT], first && h == 0);
      }
      if (reducer1) {
        const float* o = lds + AC_RED3 + o1p * 17 * 3 + tid % 3;
        for (int y = 0; y < (h ? 16 : 17); ++y) s1all += (double)o[y * 3];
        double* dst = out + AC_MAIN + tid * 3;
        if (h == 0) add_out(dst + 1, (double)o[0], first);
        else {
          add_out(dst, s1all, first);
          add_out(dst + 2, (double)o[15 * 3], first);
        }
      }
    }
    if (reducer) {
      double* dst = out + (rp * 25 + rd) * 3;
      add_out(dst + 750 + 1, (double)cacc[0], first);   // E0: (row 0, column 0), (row 32, column 0)
      add_out(dst + 750 + 2, (double)cacc[2], first);
      add_out(dst + 1500 + 1, (double)cacc[1], first);  // E32: (row 0, column 32), (row 32, column 32)
      add_out(dst + 1500 + 2, (double)cacc[3], first);
    }
    first = false;
    __syncthreads();
    for (int i = tid; i < AC_LDS2; i += AC_THREADS) lds2[i] = f32x2{0.f, 0.f};  // (the halo again; the next commit follows a barrier)
    clear();
  };
  clear();
  f32x2 pre[4][3];
  auto fetch = [&](int b) {
    const float* src = img + (size_t)b * 4 * IPIX;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const float* s = src + ch * IPIX;
      const float a0 = s[src_lo0], a1 = s[src_lo], a2 = s[src_lo + 7 * IH];
      const float b0 = s[src_hi], b1 = s[src_hi + 7 * IH], b2 = s[src_hi2];
      pre[ch][0] = f32x2{lo_ok ? a0 : 0.f, b0};
      pre[ch][1] = f32x2{a1, b1};
      pre[ch][2] = f32x2{a2, hi_ok ? b2 : 0.f};
    }
  };
  if ((int)blockIdx.x < B && loader) fetch(blockIdx.x);
  int run = 0;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    if (loader) {
#pragma unroll
      for (int ch = 0; ch < 4; ++ch)
#pragma unroll
        for (int u = 0; u < 3; ++u) dst0[ch * AC_PLANE + 7 * u * AC_LD] = pre[ch][u];
    }
    __syncthreads();
    if (b + (int)gridDim.x < B && loader) fetch(b + gridDim.x);
    if (tid < AC_EDGE0) {
      if (mainw) ac_walk<IH, 1, AC_LD>(pa, pw, acc, on);
    } else if (edgew) {
      ac_walk<17, AC_LD, 1>(pa, pw, acc, on);
    }
    if (reducer) {  // the corner products: (row 0, col 0), (row 0, col 32), (row 32, col 0), (row 32, col 32)
      const f32x2* xi = lds2 + ac_pair_i(rp) * AC_PLANE;
      const f32x2* xj = lds2 + ac_pair_j(rp) * AC_PLANE;
      const int dy = rd / 5 - 2, dx = rd % 5 - 2;
      cacc[0] = fmaf(ac_at(xi, 0, 0), ac_at(xj, dy, dx), cacc[0]);
      cacc[1] = fmaf(ac_at(xi, 0, IH - 1), ac_at(xj, dy, IH - 1 + dx), cacc[1]);
      cacc[2] = fmaf(ac_at(xi, IH - 1, 0), ac_at(xj, IH - 1 + dy, dx), cacc[2]);
      cacc[3] = fmaf(ac_at(xi, IH - 1, IH - 1), ac_at(xj, IH - 1 + dy, IH - 1 + dx), cacc[3]);
    }
    if (++run == AC_RUN) {
      flush();
      run = 0;
    }
  }
  if (run || first) flush();
  if (tid >= 12 && tid < 12 + AC_ROW - AC_MAIN - 36) out[AC_MAIN + 36 + tid - 12] = 0.0;  // (padding columns)
}

// column sums of the per-workgroup rows (f64, fixed order): 16 columns x 16 row groups per workgroup, 8 loads in flight
__global__ __launch_bounds__(256) void image_gram_ac_fold_kernel(const double* part, int rows, double* folded /*[AC_ROW]*/) {
  __shared__ double red[16][16];
  const int col = blockIdx.x * 16 + (threadIdx.x & 15), rg = threadIdx.x >> 4;
  double t = 0.0;
  int r = rg;
  for (; r + 7 * 16 < rows; r += 8 * 16) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(r + u * 16) * AC_ROW + col];
    t += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  for (; r < rows; r += 16) t += part[(size_t)r * AC_ROW + col];
  red[rg][threadIdx.x & 15] = t;
  __syncthreads();
  if (threadIdx.x >= 16) return;
  t = 0.0;
  for (int i = 0; i < 16; ++i) t += red[i][threadIdx.x];
  folded[col] = t;
}

// rows a tap with vertical offset ky reaches, from {all rows, row 0, row 32}
__device__ __forceinline__ double ac_rows(const double* v, int ky) {
  return v[0] - (ky == 1 ? v[1] : 0.0) - (ky == -1 ? v[2] : 0.0);
}
__device__ __forceinline__ double ac_entry(const double* f, int p, int ky, int kx, int dy, int dx) {
  const int d = (dy + 2) * 5 + dx + 2;
  double v = ac_rows(f + ((0 * 10 + p) * 25 + d) * 3, ky);
  if (kx == 1) v -= ac_rows(f + ((1 * 10 + p) * 25 + d) * 3, ky);
  if (kx == -1) v -= ac_rows(f + ((2 * 10 + p) * 25 + d) * 3, ky);
  return v;
}
__global__ __launch_bounds__(256) void image_gram_ac_assemble_kernel(const double* folded, int B, double* gram, const int* dims) {
  MG_REAL_IMAGES(B, dims)
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= NTAP * NTAP) return;
  int s = e / NTAP, t = e - s * NTAP;
  if (s > t) {  // (both triangles from the same expression: the matrix is symmetric to the bit)
    const int u = s;
    s = t;
    t = u;
  }
  double v;
  if (s == 36 && t == 36) {
    v = (double)IPIX * (double)B;
  } else if (s == 36 || t == 36) {
    const int u = s == 36 ? t : s, i = u / 9, ky = (u % 9) / 3 - 1, kx = u % 3 - 1;
    const double* o = folded + AC_MAIN + i * 9;
    v = ac_rows(o, ky);
    if (kx == 1) v -= ac_rows(o + 3, ky);
    if (kx == -1) v -= ac_rows(o + 6, ky);
  } else {
    const int i = s / 9, ky = (s % 9) / 3 - 1, kx = s % 3 - 1;
    const int j = t / 9, ly = (t % 9) / 3 - 1, lx = t % 3 - 1;
    const int lo = i <= j ? i : j, hi = i <= j ? j : i;
    const int p = (lo == 0 ? 0 : (lo == 1 ? 4 : (lo == 2 ? 7 : 9))) + hi - lo;
    v = i <= j ? ac_entry(folded, p, ky, kx, ly - ky, lx - kx) : ac_entry(folded, p, ly, lx, ky - ly, kx - lx);
  }
  gram[e] = v;
}

// ------------------------------------------------------------------------------------------------------------------
// conv1 weight gradient, sparse part: A[c][t] = sum over images and pooled cells of G1c[c][cell] * patch[pos(code)][t].
// MFMA with M = channel, N = tap (three tiles), K = four pooled cells that share ONE window element e: the A operand of
// lane (c = fi, k = fk) is the gradient of cell k if the saved argmax position of (c, cell) equals e, else 0
// (lane-local); B = the patch value of tap j at element e of cell k, from the tap-per-bank LDS image.  Wave w owns
// window element e = w; the cells of a k-step are (py, px), (py, px+8), (py+1, px), (py+1, px+8): the two lane groups
// of a half-wave sit 16 image columns apart, i.e. on disjoint banks.
#define GS_LD 257   // row stride of the staged gradients (floats): channel c on bank c + cell
#define CS_LD 260   // row stride of the staged codes (bytes)
// The finalize (fold the partial rows, the f64 formula, dW +=) rides in the launch (round 5): the LAST C workgroups to finish
// wait until every row is there -- a workgroup only gets one of the last C tickets when at most C - 1 others are still at
// work, and those are running or find free slots: no deadlock -- and take one output channel each.
struct C1Fin {
  unsigned* ticket;  // [arrivals, finished finalizers], zero before the first launch, left zero; NULL: no finalize
  const double* gram;
  const float* W;
  const float* bias;
  const double* coefd;
  float* dW;
};
template <bool ATOMIC>
__device__ __forceinline__ void conv1_wgrad_finalize_channel(const double* part, int rows, int C, int c,
                                                             const double* __restrict__ gram, const float* __restrict__ W,
                                                             const float* __restrict__ bias,
                                                             const double* __restrict__ coefd, float* dW,
                                                             double (*red)[36]);

template <int C>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(int B, const float* __restrict__ img, const float* __restrict__ G1c,
                                                          const unsigned char* __restrict__ code1, double* part /*[grid][C*36]*/,
                                                          C1Fin fin, const int* dims) {
  MG_REAL_IMAGES(B, dims)
  __shared__ __attribute__((aligned(16))) float imgp[GLDS];
  __shared__ float gs[16 * GS_LD];
  __shared__ __attribute__((aligned(16))) unsigned char cs[16 * CS_LD];
  __shared__ double fold[4][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fi = lane & 15, fk = lane >> 4;
  for (int i = tid; i < GLDS; i += 256) imgp[i] = 0.f;
  for (int i = tid; i < 16 * GS_LD; i += 256) gs[i] = 0.f;
  for (int i = tid; i < 16 * CS_LD; i += 256) cs[i] = 255;  // rows c >= C never match an element
  const int e = w;
  // B operand: lane (k = fk, j = fi): tap 16*blk + fi at element e of cell (py0 + (fk>>1), p + 8*(fk&1))
  int boff[3];
#pragma unroll
  for (int blk = 0; blk < 3; ++blk) {
    const int t = 16 * blk + fi;
    boff[blk] = t < 36 ? (t / 9) * GPLANE + ((t % 9) / 3 + 2 * (fk >> 1) + (e >> 1)) * GLD + (t % 3) + 16 * (fk & 1) + (e & 1) : -1;
  }
  const int cell_lane = (fk >> 1) * 16 + 8 * (fk & 1);
  double acc[3][4];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[q][r] = 0.0;
  float pre[IMG_PER], pg_[C];
  unsigned pc_[C / 4];
  auto fetch = [&](int b) {
    fetch_image(img + (size_t)b * 4 * IPIX, pre);
#pragma unroll
    for (int u = 0; u < C; ++u) pg_[u] = G1c[(size_t)b * C * 256 + u * 256 + tid];
#pragma unroll
    for (int u = 0; u < C / 4; ++u) pc_[u] = reinterpret_cast<const unsigned*>(code1 + (size_t)b * C * 256)[u * 256 + tid];
  };
  if ((int)blockIdx.x < B) fetch(blockIdx.x);
  __syncthreads();
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    commit_image<GLD, GPLANE>(pre, imgp);
#pragma unroll
    for (int u = 0; u < C; ++u) gs[u * GS_LD + tid] = pg_[u];
#pragma unroll
    for (int u = 0; u < C / 4; ++u) {  // word u*256 + tid of the (C,256) code bytes: channel = word / 64
      const int word = u * 256 + tid;
      reinterpret_cast<unsigned*>(cs)[(word >> 6) * (CS_LD / 4) + (word & 63)] = pc_[u];
    }
    __syncthreads();
    if (b + (int)gridDim.x < B) fetch(b + gridDim.x);
    f32x4 a[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) a[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (unrolled deep: the operands of a k step are three LDS reads behind a byte compare; with two steps per trip the
    // matrix pipe waited for them -- 181 -> 149 us at 8,192 images with C = 16, 172 -> 140 with C = 8)
#pragma unroll 16
    for (int ks = 0; ks < 64; ++ks) {
      const int py0 = 2 * (ks >> 3), p = ks & 7;
      const int cell = py0 * 16 + p + cell_lane;
      const float g = (int)cs[fi * CS_LD + cell] == e ? gs[fi * GS_LD + cell] : 0.f;
      const int base = 2 * py0 * GLD + 2 * p;
#pragma unroll
      for (int blk = 0; blk < 3; ++blk) {
        const float bv = imgp[boff[blk] >= 0 ? base + boff[blk] : GZERO];
        a[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(g, bv, a[blk], 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][r] += (double)a[q][r];
  }
  // D fragment: lane holds rows c = 4*fk + r, column tap = 16*blk + fi; the four waves (window elements) add up
  double* out = part + (size_t)blockIdx.x * C * 36;
  for (int q = 0; q < 3; ++q)
    for (int r = 0; r < 4; ++r) {
      __syncthreads();
      fold[w][lane] = acc[q][r];
      __syncthreads();
      const int c = 4 * fk + r, t = 16 * q + fi;
      if (w == 0 && c < C && t < 36) {
        const double v = (fold[0][lane] + fold[1][lane]) + (fold[2][lane] + fold[3][lane]);
        if (fin.ticket) store_part(out + c * 36 + t, v);  // (read by another workgroup of this launch)
        else out[c * 36 + t] = v;
      }
    }
  if (!fin.ticket) return;
  __shared__ int my_c;
  __shared__ double red[7][36];
  const int nfin = (int)gridDim.x < C ? (int)gridDim.x : C;
  __syncthreads();  // every lane's store_part has completed (each waited for its own)
  if (tid == 0) {
    const int t = (int)atomicAdd(&fin.ticket[0], 1u);
    my_c = t - ((int)gridDim.x - nfin);
    if (my_c >= 0)
      while (__hip_atomic_load(&fin.ticket[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  if (my_c < 0) return;
  for (int c = my_c; c < C; c += nfin)
    conv1_wgrad_finalize_channel<true>(part, gridDim.x, C, c, fin.gram, fin.W, fin.bias, fin.coefd, fin.dW, red);
  if (tid == 0) {
    const unsigned d = atomicAdd(&fin.ticket[1], 1u);
    if ((int)d == nfin - 1) {  // the last finalizer re-arms both words
      __hip_atomic_store(&fin.ticket[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&fin.ticket[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same sums A[c][t] by GATHER on the vector ALU.  A pooled cell passes its gradient to ONE of its four window
// positions, a different one per channel: the matrix form above multiplies every cell against all four (the A operand is
// zero for three of them) and pads 36 taps to 48 -- 5.3 matrix slots per useful product, 0.10 of the f32 rate.  Here a
// lane owns one channel and walks cells: g * (the 36 patch values around the cell's selected position), 36 FMAs on 36
// ds_read_b32 at a lane-private address -- bound by the LDS (one float per FMA), which is still 2.4x less time than the
// matrix pipe's wasted slots.  The reads are conflict-free by construction: the 32 lanes of an LDS pass are 8
// neighbouring cells of one row x 4 channels; with a row stride of 48 floats (== 16 mod 32) the position (ey, ex) of
// cell px lands on bank 2 px + ex + 16 (ey ^ ky & 1) + const -- 32 different banks for the 32 possible (cell, position)
// pairs, and lanes that picked the same pair read the same address (a broadcast).  Wave w owns four channels (C = 16:
// 4w .. 4w+3, all 16 rows of cells; C = 8: 4 (w & 1) .., every other row).  f32 over a run of four images (64 products
// per sum and lane), then the 16 lanes of a channel meet and the run goes into f64 (LDS); one row per workgroup as above.
#define WG_LD 48
#define WG_PLANE (35 * WG_LD)
#define WG_LDS (4 * WG_PLANE)
#define WG_GLD 264  // channel stride of the staged gradients (floats) and codes (bytes): four channels of a pass 8 banks apart
#define WG_RUN 4
template <int C>
__global__ __launch_bounds__(256, 3) void conv1_wgrad_gather_kernel(int B, const float* __restrict__ img,
                                                                    const float* __restrict__ G1c,
                                                                    const unsigned char* __restrict__ code1,
                                                                    double* part /*[grid][C*36]*/, const int* dims) {
  MG_REAL_IMAGES(B, dims)
  constexpr int NI = C == 16 ? 16 : 8;  // cells per lane and image
  __shared__ __attribute__((aligned(16))) float imgp[WG_LDS];
  __shared__ float gs[C * WG_GLD];
  __shared__ __attribute__((aligned(16))) unsigned char cs[C * WG_GLD];
  __shared__ double dsum[4][4][36];  // [wave][channel of the wave's four][tap]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < WG_LDS; i += 256) imgp[i] = 0.f;
  for (int i = tid; i < 4 * 4 * 36; i += 256) (&dsum[0][0][0])[i] = 0.0;
  const int px = (lane & 7) + 8 * (lane >> 5), cq = (lane >> 3) & 3;
  const int c = 4 * (C == 16 ? w : (w & 1)) + cq;
  const int py0 = C == 16 ? 0 : (w >> 1), pystep = C == 16 ? 1 : 2;
  float acc[36];
#pragma unroll
  for (int t = 0; t < 36; ++t) acc[t] = 0.f;
  float pre[IMG_PER], pg_[C];
  unsigned pc_[C / 4];
  auto fetch = [&](int b) {
    fetch_image(img + (size_t)b * 4 * IPIX, pre);
#pragma unroll
    for (int u = 0; u < C; ++u) pg_[u] = G1c[(size_t)b * C * 256 + u * 256 + tid];
#pragma unroll
    for (int u = 0; u < C / 4; ++u) pc_[u] = reinterpret_cast<const unsigned*>(code1 + (size_t)b * C * 256)[u * 256 + tid];
  };
  // the run's sums: the 16 lanes of a channel (lane bits 0-2 and 5) meet, lane 0 of them adds into the wave's f64 row
  auto flush = [&]() {
#pragma unroll
    for (int t = 0; t < 36; ++t) {
      float v = acc[t];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 32, 64);
      if ((lane & 0x27) == 0) dsum[w][cq][t] += (double)v;
      acc[t] = 0.f;
    }
  };
  if ((int)blockIdx.x < B) fetch(blockIdx.x);
  __syncthreads();
  int run = 0;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    commit_image<WG_LD, WG_PLANE>(pre, imgp);
#pragma unroll
    for (int u = 0; u < C; ++u) gs[u * WG_GLD + tid] = pg_[u];
#pragma unroll
    for (int u = 0; u < C / 4; ++u) {  // word u*256 + tid of the (C,256) code bytes: channel = word / 64
      const int word = u * 256 + tid;
      reinterpret_cast<unsigned*>(cs)[(word >> 6) * (WG_GLD / 4) + (word & 63)] = pc_[u];
    }
    __syncthreads();
    if (b + (int)gridDim.x < B) fetch(b + gridDim.x);
#pragma unroll 2
    for (int i = 0; i < NI; ++i) {
      const int py = py0 + pystep * i, cell = py * 16 + px;
      const float g = gs[c * WG_GLD + cell];
      const int e = cs[c * WG_GLD + cell];
      const float* p = imgp + (2 * py + (e >> 1)) * WG_LD + 2 * px + (e & 1);
      // all 36 reads, THEN the 36 FMAs (scheduling barriers: left alone, the compiler sinks every read to its FMA and waits
      // for each one -- 409 waits per image; 171 -> 147 us at 8,192 images, C = 16).  Three workgroups per CU cover the
      // latency of a cell's reads; a software pipeline in half cells (reads of one half under the FMAs of the other) needs
      // 168 registers + spills and is slower (166 us).
      float v[36];
#pragma unroll
      for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) v[ci * 9 + ky * 3 + kx] = p[ci * WG_PLANE + ky * WG_LD + kx];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 36; ++t) acc[t] = fmaf(g, v[t], acc[t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++run == WG_RUN) {
      flush();
      run = 0;
    }
  }
  if (run) flush();
  __syncthreads();
  double* out = part + (size_t)blockIdx.x * C * 36;
  for (int i = tid; i < C * 36; i += 256) {
    const int ch = i / 36, t = i - ch * 36;
    out[i] = C == 16 ? dsum[ch >> 2][ch & 3][t] : dsum[ch >> 2][ch & 3][t] + dsum[(ch >> 2) + 2][ch & 3][t];
  }
}

// dW[c][t] += cs_c * (A[c][t] - m1_c * Bt[t] - m2_c * Chat[c][t])   (one workgroup per output channel, f64)
// coefd = [cs | S1 | S2 | mean | invstd] (C each) + count, written by the BatchNorm-1 backward finalize
// ATOMIC: the rows were written by OTHER workgroups of the same launch (agent-scope stores): read them past the L1.
template <bool ATOMIC>
__device__ __forceinline__ void conv1_wgrad_finalize_channel(const double* part, int rows, int C, int c,
                                                             const double* __restrict__ gram, const float* __restrict__ W,
                                                             const float* __restrict__ bias,
                                                             const double* __restrict__ coefd, float* dW,
                                                             double (*red)[36]) {
  const int t = threadIdx.x % 36, rg = threadIdx.x / 36;
  if (rg < 7) {
    double s = 0.0;
    const double* p = part + c * 36 + t;
    const size_t ld = (size_t)C * 36;
    int r = rg;
    for (; r + 7 * 7 < rows; r += 8 * 7) {  // 8 loads in flight per lane, fixed order
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ATOMIC ? load_part(p + (size_t)(r + u * 7) * ld) : p[(size_t)(r + u * 7) * ld];
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; r < rows; r += 7) s += ATOMIC ? load_part(p + (size_t)r * ld) : p[(size_t)r * ld];
    red[rg][t] = s;
  }
  __syncthreads();
  if (threadIdx.x < 36) {
    double A = 0.0;
    for (int i = 0; i < 7; ++i) A += red[i][t];
    const double cs = coefd[c], S1 = coefd[C + c], S2 = coefd[2 * C + c], mean = coefd[3 * C + c], inv = coefd[4 * C + c];
    const double n = coefd[5 * C];
    const double Bt = gram[36 * NTAP + t];
    double wp = 0.0;
    for (int s = 0; s < 36; ++s) wp += (double)W[c * 36 + s] * gram[s * NTAP + t];
    const double chat = (wp + ((double)bias[c] - mean) * Bt) * inv;
    dW[c * 36 + t] += (float)(cs * (A - (S1 / n) * Bt - (S2 / n) * chat));
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void conv1_wgrad_finalize_kernel(const double* part, int rows, int C,
                                                                   const double* __restrict__ gram,
                                                                   const float* __restrict__ W, const float* __restrict__ bias,
                                                                   const double* __restrict__ coefd, float* dW) {
  __shared__ double red[7][36];
  conv1_wgrad_finalize_channel<false>(part, rows, C, blockIdx.x, gram, W, bias, coefd, dW, red);
}

// ---- sharded training, layer 1 without an exchange of its own ---------------------------------------
// Forward: the statistics of the conv1 output follow from the batch's Gram matrix of image patches and the weights --
//   sum x_c = W_c . B + n b_c,   sum x_c^2 = W_c P W_c^T + 2 b_c W_c . B + n b_c^2   (B = row 36 of P, n = P[36][36]) --
// so with the GLOBAL Gram matrix (all-reduced once per batch) every conv1 forward pass of the iteration, of either CNN, has
// its global-batch statistics without talking to anybody.  f64 throughout: the variance is a difference of two such sums.
__global__ __launch_bounds__(64) void bn1_from_gram_kernel(const double* __restrict__ gram, int C,
                                                           const float* __restrict__ W, const float* __restrict__ bias,
                                                           BnFin fin) {
  __shared__ double sums[64];
  __shared__ double rs[16 * 36];
  bn1_from_gram_block(gram, C, W, bias, fin, sums, rs);
}

// Backward: what the layer-1 adjoint needs of the other ranks -- the BatchNorm-1 adjoint sums S1 = sum g, S2 = sum g xhat and
// the raw conv1 weight-gradient sums A -- is needed by NOTHING before the weight gradient itself, so this rank's share
// [A (C x 36) | S1 (C) | S2 (C)] is folded into one f64 tail here, travels with the gradient all-reduce of the step
// (csrc/comm.hip: the tail of mggan_comm_allreduce2), and the finalize below runs behind it, identically on every rank.
__global__ __launch_bounds__(256) void conv1_tail_fold_kernel(const double* wrows, int rows, const double* part1, int rows1,
                                                              int C, double* tail, int riders, const double* rider_src) {
  __shared__ double red[7][36];
  __shared__ double colsum[32], cred[8 * 32];
  const int c = blockIdx.x;
  if (c == C) {
    colsum_rows(part1, rows1, 2 * C, colsum, cred);
    if ((int)threadIdx.x < 2 * C) tail[C * 36 + threadIdx.x] = colsum[threadIdx.x];
    // the rider slots behind the CNN's sums: what the caller prepared (rider_src), or zero (whoever rides along writes
    // behind this launch)
    if ((int)threadIdx.x < riders) tail[C * 36 + 2 * C + threadIdx.x] = rider_src ? rider_src[threadIdx.x] : 0.0;
    return;
  }
  const int t = threadIdx.x % 36, rg = threadIdx.x / 36;
  if (rg < 7) {
    double s = 0.0;
    const double* p = wrows + c * 36 + t;
    const size_t ld = (size_t)C * 36;
    int r = rg;
    for (; r + 7 * 7 < rows; r += 8 * 7) {  // 8 loads in flight per lane, fixed order (as conv1_wgrad_finalize_kernel)
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(r + u * 7) * ld];
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; r < rows; r += 7) s += p[(size_t)r * ld];
    red[rg][t] = s;
  }
  __syncthreads();
  if (threadIdx.x >= 36) return;
  double A = 0.0;
  for (int i = 0; i < 7; ++i) A += red[i][t];
  tail[c * 36 + t] = A;
}

// dW1 += (gamma/sigma) (A - (S1/n) B - (S2/n) Chat),  dgamma1 += S2,  dbeta1 += S1  from the GLOBAL tail and Gram matrix
// (n = P[36][36]); one workgroup per output channel.  Every rank computes the same numbers: the slots of these three
// parameters hold zeros during the gradient all-reduce and receive the global-batch gradient here.
__global__ __launch_bounds__(64) void conv1_tail_finalize_kernel(const double* __restrict__ tail,
                                                                 const double* __restrict__ gram, int C,
                                                                 const float* __restrict__ W, const float* __restrict__ bias,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ stat, float* dW, float* dgamma,
                                                                 float* dbeta) {
  const int c = blockIdx.x, t = threadIdx.x;
  const double S1 = tail[C * 36 + c], S2 = tail[C * 36 + C + c], n = gram[36 * NTAP + 36];
  if (t == 36) dbeta[c] += (float)S1;
  if (t == 37) dgamma[c] += (float)S2;
  if (t >= 36) return;
  const double mean = (double)stat[c], inv = (double)stat[C + c], cs = (double)gamma[c] * inv;
  const double Bt = gram[36 * NTAP + t];
  double wp = 0.0;
  for (int s = 0; s < 36; ++s) wp += (double)W[c * 36 + s] * gram[s * NTAP + t];
  const double chat = (wp + ((double)bias[c] - mean) * Bt) * inv;
  dW[c * 36 + t] += (float)(cs * (tail[c * 36 + t] - (S1 / n) * Bt - (S2 / n) * chat));
}

// persistent grid: at most `cap` workgroups (three per CU), and every workgroup walks the same number of images
// (the last one may fall short): 1,280 images -> 512 workgroups x 2-3, 8,192 -> 512 x 16, 12,288 -> 768 x 16
static int grid_for(int B, int cap) {
  static int knob = -1;  // MGGAN_CNN_GRID: measurement knob -- exactly this many workgroups (unequal image counts)
  if (knob < 0) { const char* e = getenv("MGGAN_CNN_GRID"); knob = e ? atoi(e) : 0; }
  if (knob > 0) return B < knob ? B : knob;
  // a few images per workgroup: 512 workgroups (two per CU), workgroup i takes images i, i + 512, ... -- with 1,280 images
  // three or two each, five per CU when the dispatcher deals workgroups round-robin; the equal split (640 x 2) leaves half
  // of the CUs with three workgroups = six images (configs[1]: 1.394-1.400 vs 1.407-1.409 ms, two alternating pairs;
  // from 8,192 images on the three-per-CU rule below wins: 4.74-4.77 vs 4.84-4.87 ms)
  if (B > 512 && B <= 2048) return 512;
  if (B <= cap) return B;
  // every workgroup the same number of images, two or three workgroups per CU -- whichever leaves a CU with fewer images:
  // 8,192 images as 512 x 16 are 32 per CU, as 745 x 11 some CUs walk 33 (configs[2], one box, three alternating pairs:
  // 4.515 -> 4.455 ms; 4,096 images 2.805 -> 2.744; 6,144 a tie either way; 448 or 576 workgroups lose 0.1-0.2 ms)
  const int per3 = (B + cap - 1) / cap, per2 = (B + 511) / 512;
  if (cap == 768 && 2 * per2 < 3 * per3) return (B + per2 - 1) / per2;
  return (B + per3 - 1) / per3;
}

extern "C" {

int mggan_cnn_grid(int B) { return grid_for(B, 768); }

static BnFin make_fin(unsigned* ticket, double count, const float* gamma, const float* beta, float* run_mean, float* run_var,
                      long long* nbt, float momentum, float eps, int updates, float* scale, float* shift, float* stat,
                      const void* comm = nullptr) {
  BnFin f;
  f.comm = (const CommArgs*)comm;
  f.ticket = ticket; f.count = count; f.gamma = gamma; f.beta = beta; f.run_mean = run_mean; f.run_var = run_var;
  f.nbt = nbt; f.momentum = momentum; f.eps = eps; f.updates = updates; f.scale = scale; f.shift = shift; f.stat = stat;
  return f;
}

int mggan_conv1_pool(const float* img, int B, int C, const float* W, const float* bias, float* xsel,
                     unsigned char* code, double* part, unsigned* ticket, double count, const float* gamma,
                     const float* beta, float* run_mean, float* run_var, long long* num_batches_tracked, float momentum,
                     float eps, int updates, float* scale, float* shift, float* stat, const double* gram,
                     const void* comm, const int* dims, hipStream_t stream) {
  MG_CHECK_ARG(C == 8 || C == 16, "conv1_pool: channels %d not built (8 or 16)", C);
  MG_CHECK_ARG(!comm || (ticket && !gram && !dims), "conv1_pool: the in-launch exchange needs the fused finalize (ticket), no Gram "
                                                   "matrix and an unpadded batch");
  const BnFin fin = make_fin(ticket, count, gamma, beta, run_mean, run_var, num_batches_tracked, momentum, eps, updates,
                             scale, shift, stat, comm);
  if (B == 0 && comm) {  // a rank without images still takes part in the exchange its peers' last workgroups run
    MG_CHECK_ARG(part && gamma && beta && run_mean && run_var && num_batches_tracked && scale && shift && stat, "conv1_pool: null pointer");
    MG_LAUNCH(bn_sync_finalize_dev_kernel, dim3(1), dim3(256), 0, stream, part, 0, C, fin);
    MG_LAUNCH_CHECK("conv1_pool");
    return MGGAN_OK;
  }
  if (B == 0) return MGGAN_OK;
  MG_CHECK_ARG(img && W && bias && gamma && xsel && code && part, "conv1_pool: null pointer");
  MG_CHECK_ARG(!(ticket || gram) || (beta && run_mean && run_var && num_batches_tracked && scale && shift && stat),
               "conv1_pool: the fused finalize needs the BatchNorm tensors");
  MG_CHECK_ARG(!(ticket && gram), "conv1_pool: statistics either from this launch (ticket) or from the Gram matrix");
  const int grid = grid_for(B, 768);
  if (C == 16) MG_LAUNCH((conv1_pool_kernel<16>), dim3(grid), dim3(256), 0, stream, B, img, W, bias, gamma, xsel, code, part, fin, gram, dims);
  else MG_LAUNCH((conv1_pool_kernel<8>), dim3(grid), dim3(256), 0, stream, B, img, W, bias, gamma, xsel, code, part, fin, gram, dims);
  MG_LAUNCH_CHECK("conv1_pool");
  return MGGAN_OK;
}

int mggan_conv2_fwd2(const float* xsel, int B, int C, const float* scale1, const float* shift1,
                     const float* W, const float* bias, float* y2, double* part, unsigned* ticket, double count,
                     const float* gamma, const float* beta, float* run_mean, float* run_var,
                     long long* num_batches_tracked, float momentum, float eps, int updates, float* scale, float* shift,
                     float* stat, const void* comm, const int* dims, hipStream_t stream) {
  MG_CHECK_ARG(C == 8 || C == 16, "conv2_fwd2: channels %d not built (8 or 16)", C);
  MG_CHECK_ARG(!comm || (ticket && !dims), "conv2_fwd2: the in-launch exchange needs the fused finalize (ticket) and an unpadded batch");
  const BnFin fin = make_fin(ticket, count, gamma, beta, run_mean, run_var, num_batches_tracked, momentum, eps, updates,
                             scale, shift, stat, comm);
  if (B == 0 && comm) {
    MG_CHECK_ARG(part && gamma && beta && run_mean && run_var && num_batches_tracked && scale && shift && stat, "conv2_fwd2: null pointer");
    MG_LAUNCH(bn_sync_finalize_dev_kernel, dim3(1), dim3(256), 0, stream, part, 0, C, fin);
    MG_LAUNCH_CHECK("conv2_fwd2");
    return MGGAN_OK;
  }
  if (B == 0) return MGGAN_OK;
  MG_CHECK_ARG(xsel && scale1 && shift1 && W && bias && y2 && part, "conv2_fwd2: null pointer");
  MG_CHECK_ARG(!ticket || (gamma && beta && run_mean && run_var && num_batches_tracked && scale && shift && stat),
               "conv2_fwd2: the fused finalize needs the BatchNorm tensors");
  const int grid = grid_for(B, 768);
  static int valu = -1;  // MGGAN_CONV2_VALU=1: the register-tiled VALU kernel (A/B measurements)
  if (valu < 0) { const char* e = getenv("MGGAN_CONV2_VALU"); valu = e && e[0] == '1'; }
  if (valu) {
    if (C == 16) MG_LAUNCH((conv2_fwd2_kernel<16>), dim3(grid), dim3(256), 0, stream, B, xsel, scale1, shift1, W, bias, y2, part, fin, dims);
    else MG_LAUNCH((conv2_fwd2_kernel<8>), dim3(grid), dim3(256), 0, stream, B, xsel, scale1, shift1, W, bias, y2, part, fin, dims);
  } else {
    if (C == 16) MG_LAUNCH((conv2_fwd_mfma_kernel<16>), dim3(grid), dim3(256), 0, stream, B, xsel, scale1, shift1, W, bias, y2, part, fin, dims);
    else MG_LAUNCH((conv2_fwd_mfma_kernel<8>), dim3(grid), dim3(256), 0, stream, B, xsel, scale1, shift1, W, bias, y2, part, fin, dims);
  }
  MG_LAUNCH_CHECK("conv2_fwd2");
  return MGGAN_OK;
}

int mggan_bn_reduce_rows(const double* part, int rows, int W, double* sums, hipStream_t stream) {
  MG_CHECK_ARG(part && sums && W > 0 && W <= 32, "bn_reduce_rows: bad arguments");
  MG_LAUNCH(bn_reduce_rows_kernel, dim3(1), dim3(256), 0, stream, part, rows, W, sums);
  MG_LAUNCH_CHECK("bn_reduce_rows");
  return MGGAN_OK;
}

static BnBwdFin make_bfin(unsigned* ticket, double count, const float* gamma, const float* stat, float* coef, double* coefd,
                          float* dgamma, float* dbeta, const void* comm = nullptr) {
  BnBwdFin f;
  f.comm = (const CommArgs*)comm;
  f.ticket = ticket; f.count = count; f.gamma = gamma; f.stat = stat; f.coef = coef; f.coefd = coefd; f.dgamma = dgamma;
  f.dbeta = dbeta;
  return f;
}

int mggan_bn_bwd_rows_finalize(const double* part, int rows, double count, int C, const float* gamma, const float* stat,
                               float* coef, double* coefd, float* dgamma, float* dbeta, hipStream_t stream) {
  MG_CHECK_ARG(part && gamma && stat && coef && dgamma && dbeta && C <= 16, "bn_bwd_rows_finalize: bad arguments");
  MG_LAUNCH(bn_bwd_rows_finalize_kernel, dim3(1), dim3(256), 0, stream, part, rows, C,
                     make_bfin(nullptr, count, gamma, stat, coef, coefd, dgamma, dbeta));
  MG_LAUNCH_CHECK("bn_bwd_rows_finalize");
  return MGGAN_OK;
}

int mggan_bn_bwd_coef(const double* sums, const double* local_sums, double count, int C, const float* gamma,
                      const float* stat, float* coef, double* coefd, float* dgamma, float* dbeta, hipStream_t stream) {
  MG_CHECK_ARG(sums && local_sums && gamma && stat && coef && dgamma && dbeta && C <= 16, "bn_bwd_coef: bad arguments");
  MG_LAUNCH(bn_bwd_coef_kernel, dim3(1), dim3(64), 0, stream, sums, local_sums, C,
                     make_bfin(nullptr, count, gamma, stat, coef, coefd, dgamma, dbeta));
  MG_LAUNCH_CHECK("bn_bwd_coef");
  return MGGAN_OK;
}

int mggan_bn_sync_finalize(void* const* arenas, int rank, int world, long max_elems, const double* part, int rows,
                           double local_count, int C, const float* gamma, const float* beta, float* run_mean,
                           float* run_var, long long* num_batches_tracked, float momentum, float eps, int updates,
                           float* scale, float* shift, float* stat, hipStream_t stream) {
  MG_CHECK_ARG(arenas && part && gamma && beta && run_mean && run_var && num_batches_tracked && scale && shift && stat &&
                   C <= 16 && world >= 1 && world <= COMM_MAX_RANKS,
               "bn_sync_finalize: bad arguments");
  MG_LAUNCH(bn_sync_finalize_kernel, dim3(1), dim3(256), 0, stream, comm_make_args(arenas, rank, world, max_elems),
                     part, rows, C, local_count,
                     make_fin(nullptr, 0.0, gamma, beta, run_mean, run_var, num_batches_tracked, momentum, eps, updates,
                              scale, shift, stat));
  MG_LAUNCH_CHECK("bn_sync_finalize");
  return MGGAN_OK;
}

int mggan_bn_bwd_sync_finalize(void* const* arenas, int rank, int world, long max_elems, const double* part, int rows,
                               double local_count, int C, const float* gamma, const float* stat, float* coef,
                               double* coefd, float* dgamma, float* dbeta, hipStream_t stream) {
  MG_CHECK_ARG(arenas && part && gamma && stat && coef && dgamma && dbeta && C <= 16 && world >= 1 &&
                   world <= COMM_MAX_RANKS,
               "bn_bwd_sync_finalize: bad arguments");
  MG_LAUNCH(bn_bwd_sync_finalize_kernel, dim3(1), dim3(256), 0, stream,
                     comm_make_args(arenas, rank, world, max_elems), part, rows, C, local_count,
                     make_bfin(nullptr, 0.0, gamma, stat, coef, coefd, dgamma, dbeta));
  MG_LAUNCH_CHECK("bn_bwd_sync_finalize");
  return MGGAN_OK;
}

int mggan_bn1_from_gram(const double* gram, int C, const float* W, const float* bias, const float* gamma,
                        const float* beta, float* run_mean, float* run_var, long long* num_batches_tracked, float momentum,
                        float eps, int updates, float* scale, float* shift, float* stat, hipStream_t stream) {
  MG_CHECK_ARG(gram && W && bias && gamma && beta && run_mean && run_var && num_batches_tracked && scale && shift && stat &&
                   (C == 8 || C == 16),
               "bn1_from_gram: bad arguments");
  MG_LAUNCH(bn1_from_gram_kernel, dim3(1), dim3(64), 0, stream, gram, C, W, bias,
            make_fin(nullptr, 0.0, gamma, beta, run_mean, run_var, num_batches_tracked, momentum, eps, updates, scale, shift,
                     stat));
  MG_LAUNCH_CHECK("bn1_from_gram");
  return MGGAN_OK;
}

int mggan_conv1_tail_floats(int C) { return C * 36 + 2 * C; }

int mggan_conv1_tail_fold(const double* wrows, int rows, const double* part1, int rows1, int C, double* tail, int riders,
                          const double* rider_src, hipStream_t stream) {
  MG_CHECK_ARG(wrows && part1 && tail && rows >= 0 && rows1 >= 0 && (C == 8 || C == 16) && riders >= 0 && riders <= 64,
               "conv1_tail_fold: bad arguments");
  MG_LAUNCH(conv1_tail_fold_kernel, dim3(C + 1), dim3(256), 0, stream, wrows, rows, part1, rows1, C, tail, riders, rider_src);
  MG_LAUNCH_CHECK("conv1_tail_fold");
  return MGGAN_OK;
}

int mggan_conv1_tail_finalize(const double* tail, const double* gram, int C, const float* W, const float* bias,
                              const float* gamma, const float* stat, float* dW, float* dgamma, float* dbeta,
                              hipStream_t stream) {
  MG_CHECK_ARG(tail && gram && W && bias && gamma && stat && dW && dgamma && dbeta && (C == 8 || C == 16),
               "conv1_tail_finalize: bad arguments");
  MG_LAUNCH(conv1_tail_finalize_kernel, dim3(C), dim3(64), 0, stream, tail, gram, C, W, bias, gamma, stat, dW, dgamma, dbeta);
  MG_LAUNCH_CHECK("conv1_tail_finalize");
  return MGGAN_OK;
}

/* workspace: mggan_cnn_grid(B) * 1536 doubles */
// MGGAN_GRAM_KERNEL=mfma selects the tap-by-tap MFMA kernel (the default until round 5); anything else the autocorrelation form
static bool gram_ac() {
  static const bool ac = [] {
    const char* e = getenv("MGGAN_GRAM_KERNEL");
    return !(e && e[0] == 'm');
  }();
  return ac;
}
// two workgroups of the autocorrelation kernel fit a CU (registers): 512 resident, every one with the same number of images
static int gram_ac_grid(int B) {
  static const int cap = [] {
    const char* e = getenv("MGGAN_GRAM_GRID");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 512;
  }();
  return B < cap ? B : cap;
}
size_t mggan_image_gram_workspace(int B) {
  const size_t grid = (size_t)(grid_for(B, 768) > 0 ? grid_for(B, 768) : 1), ac = (size_t)(B > 0 ? gram_ac_grid(B) : 1);
  return (gram_ac() ? (ac + 1) * AC_ROW : grid * 6 * 256) * sizeof(double);
}
int mggan_image_gram(const float* img, int B, double* gram, double* workspace, size_t workspace_bytes,
                     const int* dims, hipStream_t stream) {
  MG_CHECK_ARG(gram && workspace && (img || B == 0), "image_gram: null pointer");
  const int grid = gram_ac() ? gram_ac_grid(B) : grid_for(B, 768);
  MG_CHECK_ARG(workspace_bytes >= mggan_image_gram_workspace(B), "image_gram: workspace too small");
  if (gram_ac()) {
    double* folded = workspace + (size_t)(grid > 0 ? grid : 1) * AC_ROW;
    if (grid > 0) MG_LAUNCH(image_gram_ac_kernel, dim3(grid), dim3(AC_THREADS), 0, stream, B, img, workspace, dims);
    MG_LAUNCH(image_gram_ac_fold_kernel, dim3(AC_ROW / 16), dim3(256), 0, stream, workspace, grid, folded);
    MG_LAUNCH(image_gram_ac_assemble_kernel, dim3((NTAP * NTAP + 255) / 256), dim3(256), 0, stream, folded, B, gram, dims);
  } else {
    if (grid > 0) MG_LAUNCH(image_gram_kernel, dim3(grid), dim3(256), 0, stream, B, img, workspace, dims);
    MG_LAUNCH(image_gram_finalize_kernel, dim3(96), dim3(256), 0, stream, workspace, grid, gram);
  }
  MG_LAUNCH_CHECK("image_gram");
  return MGGAN_OK;
}

/* workspace: mggan_cnn_grid(B) * C * 36 doubles; dW (C,4,3,3) is ACCUMULATED into */
int mggan_conv1_wgrad(const float* img, int B, int C, const float* G1c, const unsigned char* code1, const double* gram,
                      const float* W, const float* bias, const double* coefd, float* dW, double* workspace,
                      size_t workspace_bytes, unsigned* ticket, const int* dims, hipStream_t stream) {
  MG_CHECK_ARG(C == 8 || C == 16, "conv1_wgrad: channels %d not built (8 or 16)", C);
  if (B == 0) return MGGAN_OK;
  MG_CHECK_ARG(img && G1c && code1 && workspace, "conv1_wgrad: null pointer");
  MG_CHECK_ARG(!dW || (gram && W && bias && coefd), "conv1_wgrad: the finalize needs gram / W / bias / coefd");
  const int grid = grid_for(B, 768);
  MG_CHECK_ARG(workspace_bytes >= (size_t)grid * C * 36 * sizeof(double), "conv1_wgrad: workspace too small");
  // ticket (two zeroed words, left zero) && dW: the finalize rides in the launch -- its last C workgroups fold the rows and add
  // dW; without a ticket it is a second launch; dW == NULL: the partial rows stay in `workspace` (sharded training:
  // mggan_conv1_tail_fold / _finalize take over)
  const C1Fin fin = {dW ? ticket : nullptr, gram, W, bias, coefd, dW};
  // MGGAN_C1WGRAD=mfma: the one-hot matrix form (rounds 2-5) instead of the gather on the vector ALU
  static const bool gather = [] {
    const char* e = getenv("MGGAN_C1WGRAD");
    return !(e && e[0] == 'm');
  }();
  if (gather && !fin.ticket) {
    if (C == 16) MG_LAUNCH((conv1_wgrad_gather_kernel<16>), dim3(grid), dim3(256), 0, stream, B, img, G1c, code1, workspace, dims);
    else MG_LAUNCH((conv1_wgrad_gather_kernel<8>), dim3(grid), dim3(256), 0, stream, B, img, G1c, code1, workspace, dims);
  } else if (C == 16) MG_LAUNCH((conv1_wgrad_kernel<16>), dim3(grid), dim3(256), 0, stream, B, img, G1c, code1, workspace, fin, dims);
  else MG_LAUNCH((conv1_wgrad_kernel<8>), dim3(grid), dim3(256), 0, stream, B, img, G1c, code1, workspace, fin, dims);
  if (dW && !ticket)
    MG_LAUNCH(conv1_wgrad_finalize_kernel, dim3(C), dim3(256), 0, stream, workspace, grid, C, gram, W, bias, coefd, dW);
  MG_LAUNCH_CHECK("conv1_wgrad");
  return MGGAN_OK;
}

}  // extern "C"
