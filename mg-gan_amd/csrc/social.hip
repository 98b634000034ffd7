// Block-diagonal social attention for MG-GAN on gfx950.
//
// Replaces (file:line under /root/reference/mggan/model/modules/social.py):
//   SocialFeatures / BearingMTX / DCA_MTX   :67-104  (distance, cos-bearing, DCA of ordered pairs)
//   EmbedSocialFeatures.fc                  :33-48   (MLP 3 -> 32 -> 64 -> F, ReLU)
//   AttentionPooling.forward                :14-30   (sigma_ij = f_ij . (W h_j + b); sigma_ii = -1000;
//                                                     softmax over the scene; S_i = sum_j a_ij h_j; n==1 -> 0)
// The reference builds dense N x N x {3,32,64,F} tensors over the whole batch and loops
// over pedestrians in Python; only same-scene pairs influence results, so the
// kernels here run over the sum_s n_s^2 in-scene ordered pairs only.
//
// Algebra used: the last embedding layer is linear, so
//   sigma_ij = (W3 l2_ij + b3) . Wh_j = l2_ij . (W3^T Wh_j) + b3 . Wh_j = l2_ij . v_j + c_j
// with [v_j | c_j] = Wh_j [W3 | b3] computed once per pedestrian (65 values) by the generic
// GEMM; per pair only the 3->32->64 part of the MLP and a 64-long dot product remain.
// Weights are wave-uniform -> scalar loads; activations stay in VGPRs.
#include "common.h"
#include "../../include/mggan_hip.h"

#define L1 32
#define L2 64

__device__ __forceinline__ void pair_features(const float* __restrict__ xy, const float* __restrict__ dxy, int i, int j,
                                              float f[3]) {
  const float pix = xy[2 * i], piy = xy[2 * i + 1], vix = dxy[2 * i], viy = dxy[2 * i + 1];
  const float dpx = pix - xy[2 * j], dpy = piy - xy[2 * j + 1];
  const float dvx = vix - dxy[2 * j], dvy = viy - dxy[2 * j + 1];
  const float dist = sqrtf(dpx * dpx + dpy * dpy);
  const float bearing = (dpx * vix + dpy * viy) / (dist * sqrtf(vix * vix + viy * viy) + 1e-6f);
  const float ttca = -(dpx * dvx + dpy * dvy) / (dvx * dvx + dvy * dvy + 1e-6f);
  const float cx = dpx + ttca * dvx, cy = dpy + ttca * dvy;
  f[0] = dist;
  f[1] = bearing;
  f[2] = sqrtf(cx * cx + cy * cy);
}

// 64 pairs per workgroup; wave w computes hidden units [16w, 16w+16) of layer 2 for all 64 pairs
// (its weights are wave-uniform -> scalar loads), the four partial scores meet in LDS.
__global__ __launch_bounds__(256) void social_pairs_fwd_kernel(
    int P, const int* __restrict__ pair_i, const int* __restrict__ pair_j, const float* __restrict__ xy,
    const float* __restrict__ dxy, const float* __restrict__ W1, const float* __restrict__ b1,
    const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ vc, float* feat, float* l1s,
    float* l2s, float* sigma) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = blockIdx.x * 64 + lane;
  const bool ok = p < P;
  const int pc = ok ? p : P - 1;
  const int i = pair_i[pc], j = pair_j[pc];
  float f[3];
  pair_features(xy, dxy, i, j, f);
  float l1[L1];
#pragma unroll
  for (int k = 0; k < L1; ++k) {
    float s = b1[k];
    s = fmaf(W1[k * 3 + 0], f[0], s);
    s = fmaf(W1[k * 3 + 1], f[1], s);
    s = fmaf(W1[k * 3 + 2], f[2], s);
    l1[k] = fmaxf(s, 0.f);
  }
  const float* v = vc + (size_t)j * (L2 + 1);
  float sg = 0.f;
  const bool save = l1s != nullptr;
#pragma unroll 4
  for (int mm = 0; mm < L2 / 4; ++mm) {
    const int m = w * (L2 / 4) + mm;
    float s = b2[m];
#pragma unroll
    for (int k = 0; k < L1; ++k) s = fmaf(W2[m * L1 + k], l1[k], s);
    s = fmaxf(s, 0.f);
    if (save && ok) l2s[(size_t)m * P + p] = s;  // feature-major: coalesced across the pair lanes
    sg = fmaf(s, v[m], sg);
  }
  part[w][lane] = sg;
  if (save && ok) {
    if (w == 0) {
      feat[p] = f[0];
      feat[(size_t)P + p] = f[1];
      feat[(size_t)2 * P + p] = f[2];
    }
#pragma unroll
    for (int k = 0; k < L1; ++k)  // static register index; wave w stores rows [8w, 8w+8)
      if ((k >> 3) == w) l1s[(size_t)k * P + p] = l1[k];
  }
  __syncthreads();
  if (w == 0 && ok) {
    const float tot = v[L2] + (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    sigma[p] = (i == j) ? -1000.0f : tot;  // social.py:25
  }
}

// one group of H lanes per pedestrian i: softmax over its scene, S_i = sum_j a_ij h_j
template <int H>
__global__ __launch_bounds__(256) void social_softmax_fwd_kernel(int b, const int* __restrict__ prow,
                                                                 const int* __restrict__ s0a, const int* __restrict__ na,
                                                                 const float* __restrict__ sigma,
                                                                 const float* __restrict__ h, int ld_h, float* att,
                                                                 float* S, int ld_s) {
  const int i = blockIdx.x * (256 / H) + threadIdx.x / H, k = threadIdx.x % H;
  if (i >= b) return;
  const int n = na[i];
  if (n <= 1) {  // social.py:19-20
    S[(size_t)i * ld_s + k] = 0.f;
    return;
  }
  const int s0 = s0a[i], pr = prow[i];
  float mx = -INFINITY;
  for (int j = 0; j < n; ++j) mx = fmaxf(mx, sigma[pr + j]);
  float den = 0.f;
  for (int j = 0; j < n; ++j) den += __expf(sigma[pr + j] - mx);
  const float inv = 1.0f / den;
  float acc = 0.f;
  for (int j = 0; j < n; ++j) {
    const float a = __expf(sigma[pr + j] - mx) * inv;
    if (k == 0 && att) att[pr + j] = a;
    acc = fmaf(a, h[(size_t)(s0 + j) * ld_h + k], acc);
  }
  S[(size_t)i * ld_s + k] = acc;
}

// dsigma_ij = a_ij (da_ij - sum_j a_ij da_ij), da_ij = dS_i . h_j
template <int H>
__global__ __launch_bounds__(256) void social_softmax_bwd_kernel(int b, const int* __restrict__ prow,
                                                                 const int* __restrict__ s0a, const int* __restrict__ na,
                                                                 const float* __restrict__ att,
                                                                 const float* __restrict__ h, int ld_h,
                                                                 const float* __restrict__ dS, int ld_ds,
                                                                 float* dsigma) {
  const int i = blockIdx.x * (256 / H) + threadIdx.x / H, k = threadIdx.x % H;
  if (i >= b) return;  // H divides 64: whole lane groups leave together
  const int n = na[i];
  if (n <= 1) return;
  const int s0 = s0a[i], pr = prow[i];
  const float ds = dS[(size_t)i * ld_ds + k];
  float dot = 0.f;
  for (int j = 0; j < n; ++j) {
    float da = ds * h[(size_t)(s0 + j) * ld_h + k];
#pragma unroll
    for (int o = H / 2; o > 0; o >>= 1) da += __shfl_xor(da, o, 64);
    if (k == 0) dsigma[pr + j] = da;  // stash da
    dot = fmaf(att[pr + j], da, dot);
  }
  if (k == 0) {
    for (int j = 0; j < n; ++j) dsigma[pr + j] = att[pr + j] * (dsigma[pr + j] - dot);
  }
}

// dh_j[k] (+)= sum_i a_ij dS_i[k]
template <int H>
__global__ __launch_bounds__(256) void social_dh_kernel(int b, const int* __restrict__ prow, const int* __restrict__ s0a,
                                                        const int* __restrict__ na, const float* __restrict__ att,
                                                        const float* __restrict__ dS, int ld_ds, float* dh, int ld_dh,
                                                        int accumulate) {
  const int j = blockIdx.x * (256 / H) + threadIdx.x / H, k = threadIdx.x % H;
  if (j >= b) return;
  const int n = na[j];
  float acc = 0.f;
  if (n > 1) {
    const int s0 = s0a[j], lj = j - s0;
    for (int i = 0; i < n; ++i) acc = fmaf(att[prow[s0 + i] + lj], dS[(size_t)(s0 + i) * ld_ds + k], acc);
  }
  float* d = dh + (size_t)j * ld_dh + k;
  *d = accumulate ? (*d + acc) : acc;
}

// per pair: dz2 = dsigma * v_j * relu'(l2), dz1 = (W2^T dz2) * relu'(l1); wave w owns hidden units
// [16w, 16w+16) of layer 2, the four partial W2^T dz2 vectors are summed through LDS.
__global__ __launch_bounds__(256) void social_pairs_bwd_kernel(int P, const int* __restrict__ pair_j,
                                                               const float* __restrict__ dsigma,
                                                               const float* __restrict__ vc,
                                                               const float* __restrict__ l1s,
                                                               const float* __restrict__ l2s,
                                                               const float* __restrict__ W2, float* dz2, float* dz1) {
  __shared__ float part[4][L1][64];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = blockIdx.x * 64 + lane;
  const bool ok = p < P;
  const int pc = ok ? p : P - 1;
  const float dsg = dsigma[pc];
  const float* v = vc + (size_t)pair_j[pc] * (L2 + 1);
  float d1[L1];
#pragma unroll
  for (int k = 0; k < L1; ++k) d1[k] = 0.f;
#pragma unroll 4
  for (int mm = 0; mm < L2 / 4; ++mm) {
    const int m = w * (L2 / 4) + mm;
    const float z = l2s[(size_t)m * P + pc] > 0.f ? dsg * v[m] : 0.f;
    if (ok) dz2[(size_t)m * P + p] = z;
#pragma unroll
    for (int k = 0; k < L1; ++k) d1[k] = fmaf(W2[m * L1 + k], z, d1[k]);
  }
#pragma unroll
  for (int k = 0; k < L1; ++k) part[w][k][lane] = d1[k];
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < L1 / 4; ++kk) {
    const int k = w * (L1 / 4) + kk;
    const float t = (part[0][k][lane] + part[1][k][lane]) + (part[2][k][lane] + part[3][k][lane]);
    if (ok) dz1[(size_t)k * P + p] = l1s[(size_t)k * P + p] > 0.f ? t : 0.f;
  }
}

// dvc[j][m] = sum_i dsigma_ij * (m < 64 ? l2_ij[m] : 1)
__global__ __launch_bounds__(256) void social_dvc_kernel(int b, int P, const int* __restrict__ prow,
                                                         const int* __restrict__ s0a, const int* __restrict__ na,
                                                         const float* __restrict__ dsigma,
                                                         const float* __restrict__ l2s, float* dvc) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)b * (L2 + 1)) return;
  const int m = (int)(t / b), j = (int)(t % b);  // lanes run over pedestrians: l2s[m][pair] reads are coalesced
  const int n = na[j];
  float acc = 0.f;
  if (n > 1) {
    const int s0 = s0a[j], pb = prow[s0] + (j - s0);
    const float* dsg = dsigma + pb;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = 0;
    if (m < L2) {
      const float* l2 = l2s + (size_t)m * P + pb;
      for (; i + 3 < n; i += 4) {
        a0 = fmaf(dsg[i * n], l2[i * n], a0);
        a1 = fmaf(dsg[(i + 1) * n], l2[(i + 1) * n], a1);
        a2 = fmaf(dsg[(i + 2) * n], l2[(i + 2) * n], a2);
        a3 = fmaf(dsg[(i + 3) * n], l2[(i + 3) * n], a3);
      }
      for (; i < n; ++i) a0 = fmaf(dsg[i * n], l2[i * n], a0);
    } else {
      for (; i < n; ++i) a0 += dsg[i * n];
    }
    acc = (a0 + a1) + (a2 + a3);
  }
  dvc[(size_t)j * (L2 + 1) + m] = acc;
}

// ---- fused launches over pedestrian-aligned tiles ------------------------------------------------
// A tile is a run of consecutive pedestrians whose in-scene pairs are contiguous in the pair list and
// number at most 64 (tile = {ped0, ped1, first pair, pair count}, built on the host once per batch).
// Every pair of a pedestrian then sits in the same workgroup, so its softmax row never leaves LDS:
// forward = pair MLP + scores + softmax + pooling in one launch, backward = softmax adjoint + pair MLP
// adjoint in one launch, and the two column reductions (dh, dvc) share a third.

template <int H>
__global__ __launch_bounds__(256) void social_fwd_fused_kernel(
    const int4* __restrict__ tiles, int P, const int* __restrict__ pair_i, const int* __restrict__ pair_j,
    const int* __restrict__ prow, const int* __restrict__ s0a, const int* __restrict__ na,
    const float* __restrict__ xy, const float* __restrict__ dxy, const float* __restrict__ W1,
    const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
    const float* __restrict__ vc, const float* __restrict__ h, int ld_h, float* feat, float* l1s, float* l2s,
    float* att, float* S, int ld_s, int xy_mod) {
  __shared__ float part[4][64];
  __shared__ float sg_s[64], a_s[64];
  const int4 tl = tiles[blockIdx.x];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p0 = tl.z, np = tl.w;
  if (np > 0) {  // workgroup-uniform
    const bool ok = lane < np;
    const int p = p0 + (ok ? lane : 0);
    const int i = pair_i[p], j = pair_j[p];
    float f[3];
    // xy_mod > 0: pedestrian rows repeat with that period (the real and the fake half of a pair pass share the
    // observed positions), so the position tables hold one period only
    pair_features(xy, dxy, xy_mod > 0 ? i % xy_mod : i, xy_mod > 0 ? j % xy_mod : j, f);
    float l1[L1];
#pragma unroll
    for (int k = 0; k < L1; ++k) {
      float s = b1[k];
      s = fmaf(W1[k * 3 + 0], f[0], s);
      s = fmaf(W1[k * 3 + 1], f[1], s);
      s = fmaf(W1[k * 3 + 2], f[2], s);
      l1[k] = fmaxf(s, 0.f);
    }
    const float* v = vc + (size_t)j * (L2 + 1);
    float sg = 0.f;
    const bool save = l1s != nullptr;
#pragma unroll 4
    for (int mm = 0; mm < L2 / 4; ++mm) {
      const int m = w * (L2 / 4) + mm;
      float s = b2[m];
#pragma unroll
      for (int k = 0; k < L1; ++k) s = fmaf(W2[m * L1 + k], l1[k], s);
      s = fmaxf(s, 0.f);
      if (save && ok) l2s[(size_t)m * P + p] = s;
      sg = fmaf(s, v[m], sg);
    }
    part[w][lane] = sg;
    if (save && ok) {
      if (w == 0) {
        feat[p] = f[0];
        feat[(size_t)P + p] = f[1];
        feat[(size_t)2 * P + p] = f[2];
      }
#pragma unroll
      for (int k = 0; k < L1; ++k)
        if ((k >> 3) == w) l1s[(size_t)k * P + p] = l1[k];
    }
    __syncthreads();
    const float tot = v[L2] + (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    const float sc = (i == j) ? -1000.0f : tot;  // social.py:25
    if (w == 0) sg_s[lane] = sc;
    __syncthreads();
    // every wave walks its lanes' softmax rows (same values in all four; wave 0 publishes)
    // (loops over a run-time n are walked four entries at a time with independent partial results: one LDS or memory
    //  round trip per group of four instead of per entry - these walks were most of a workgroup's critical path)
    const int seg0 = prow[i] - p0, n = na[i];
    const float* row = sg_s + seg0;
    float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
    int jj = 0;
    for (; jj + 3 < n; jj += 4) {
      m0 = fmaxf(m0, row[jj]); m1 = fmaxf(m1, row[jj + 1]); m2 = fmaxf(m2, row[jj + 2]); m3 = fmaxf(m3, row[jj + 3]);
    }
    for (; jj < n; ++jj) m0 = fmaxf(m0, row[jj]);
    const float mx = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    for (jj = 0; jj + 3 < n; jj += 4) {
      d0 += __expf(row[jj] - mx); d1 += __expf(row[jj + 1] - mx); d2 += __expf(row[jj + 2] - mx); d3 += __expf(row[jj + 3] - mx);
    }
    for (; jj < n; ++jj) d0 += __expf(row[jj] - mx);
    const float den = (d0 + d1) + (d2 + d3);
    const float a = __expf(sc - mx) * (1.0f / den);
    if (w == 0) {
      a_s[lane] = a;
      if (ok) att[p] = a;
    }
    __syncthreads();
  }
  const int k = threadIdx.x % H;
  for (int q = tl.x + threadIdx.x / H; q < tl.y; q += 256 / H) {
    const int n = na[q];
    float acc = 0.f;
    if (n > 1) {  // social.py:19-20: a lone pedestrian pools nothing
      const float* ar = a_s + (prow[q] - p0);
      const float* hr = h + (size_t)s0a[q] * ld_h + k;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int jj = 0;
      for (; jj + 3 < n; jj += 4) {
        a0 = fmaf(ar[jj], hr[(size_t)jj * ld_h], a0);
        a1 = fmaf(ar[jj + 1], hr[(size_t)(jj + 1) * ld_h], a1);
        a2 = fmaf(ar[jj + 2], hr[(size_t)(jj + 2) * ld_h], a2);
        a3 = fmaf(ar[jj + 3], hr[(size_t)(jj + 3) * ld_h], a3);
      }
      for (; jj < n; ++jj) a0 = fmaf(ar[jj], hr[(size_t)jj * ld_h], a0);
      acc = (a0 + a1) + (a2 + a3);
    }
    S[(size_t)q * ld_s + k] = acc;
  }
}

template <int H>
__global__ __launch_bounds__(256) void social_bwd_fused_kernel(
    const int4* __restrict__ tiles, int P, const int* __restrict__ pair_i, const int* __restrict__ pair_j,
    const int* __restrict__ prow, const int* __restrict__ na, const float* __restrict__ att,
    const float* __restrict__ h, int ld_h, const float* __restrict__ dS, int ld_ds, const float* __restrict__ vc,
    const float* __restrict__ l1s, const float* __restrict__ l2s, const float* __restrict__ W2, float* dsigma,
    float* dz2, float* dz1) {
  // 84 % of this kernel's wave time is spent parked on memory: its duration is (workgroups / resident workgroups) x
  // one workgroup's chain of dependent round trips.  The four partial W2^T dz2 vectors are therefore folded in two halves
  // of 16 layer-1 units through a 16 KB buffer (one more barrier) instead of all 32 through 32 KB: eight resident
  // workgroups per CU instead of four.
  __shared__ float part[4][L1 / 2][64];
  __shared__ float dpart[4][64];
  __shared__ float ad_s[64];
  const int4 tl = tiles[blockIdx.x];
  const int p0 = tl.z, np = tl.w;
  if (np == 0) return;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool ok = lane < np;
  const int p = p0 + (ok ? lane : 0);
  const int i = pair_i[p], j = pair_j[p];
  // da_ij = dS_i . h_j, wave w sums its quarter of the hidden units
  {
    const float* a = dS + (size_t)i * ld_ds + w * (H / 4);
    const float* bq = h + (size_t)j * ld_h + w * (H / 4);
    float d = 0.f;
#pragma unroll
    for (int kk = 0; kk < H / 4; ++kk) d = fmaf(a[kk], bq[kk], d);
    dpart[w][lane] = d;
  }
  const float a_ij = att[p];
  __syncthreads();
  const float da = (dpart[0][lane] + dpart[1][lane]) + (dpart[2][lane] + dpart[3][lane]);
  if (w == 0) ad_s[lane] = a_ij * da;
  __syncthreads();
  const int seg0 = prow[i] - p0, n = na[i];
  const float* adr = ad_s + seg0;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  int jj = 0;
  for (; jj + 3 < n; jj += 4) { t0 += adr[jj]; t1 += adr[jj + 1]; t2 += adr[jj + 2]; t3 += adr[jj + 3]; }
  for (; jj < n; ++jj) t0 += adr[jj];
  const float dot = (t0 + t1) + (t2 + t3);
  const float dsg = a_ij * (da - dot);
  if (w == 0 && ok) dsigma[p] = dsg;
  const float* v = vc + (size_t)j * (L2 + 1);
  float d1[L1];
#pragma unroll
  for (int k = 0; k < L1; ++k) d1[k] = 0.f;
#pragma unroll 4
  for (int mm = 0; mm < L2 / 4; ++mm) {
    const int m = w * (L2 / 4) + mm;
    const float z = l2s[(size_t)m * P + p] > 0.f ? dsg * v[m] : 0.f;
    if (ok) dz2[(size_t)m * P + p] = z;
#pragma unroll
    for (int k = 0; k < L1; ++k) d1[k] = fmaf(W2[m * L1 + k], z, d1[k]);
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();  // every wave has read the first half
#pragma unroll
    for (int k = 0; k < L1 / 2; ++k) part[w][k][lane] = d1[half * (L1 / 2) + k];
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < L1 / 8; ++kk) {
      const int kl = w * (L1 / 8) + kk, k = half * (L1 / 2) + kl;
      const float t = (part[0][kl][lane] + part[1][kl][lane]) + (part[2][kl][lane] + part[3][kl][lane]);
      if (ok) dz1[(size_t)k * P + p] = l1s[(size_t)k * P + p] > 0.f ? t : 0.f;
    }
  }
}

// the two reductions over i for a fixed neighbour j in one launch: workgroups [0, nb_dh) do
// dh_j = sum_i a_ij dS_i, the rest dvc_j = sum_i dsigma_ij [l2_ij | 1]
template <int H>
__global__ __launch_bounds__(256) void social_dh_dvc_kernel(int b, int P, int nb_dh, const int* __restrict__ prow,
                                                            const int* __restrict__ s0a, const int* __restrict__ na,
                                                            const float* __restrict__ att,
                                                            const float* __restrict__ dS, int ld_ds, float* dh,
                                                            int ld_dh, int accumulate,
                                                            const float* __restrict__ dsigma,
                                                            const float* __restrict__ l2s, float* dvc) {
  if ((int)blockIdx.x < nb_dh) {
    const int j = blockIdx.x * (256 / H) + threadIdx.x / H, k = threadIdx.x % H;
    if (j >= b) return;
    const int n = na[j];
    float acc = 0.f;
    if (n > 1) {
      // the pairs of a scene are an n x n block in pedestrian order: pair (s0 + i, j) sits at prow[s0] + i n + (j - s0)
      // (no table lookup per term, and four independent partial sums keep the loads of a column in flight)
      const int s0 = s0a[j], pb = prow[s0] + (j - s0);
      const float* ds = dS + (size_t)s0 * ld_ds + k;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int i = 0;
      for (; i + 3 < n; i += 4) {
        a0 = fmaf(att[pb + i * n], ds[(size_t)i * ld_ds], a0);
        a1 = fmaf(att[pb + (i + 1) * n], ds[(size_t)(i + 1) * ld_ds], a1);
        a2 = fmaf(att[pb + (i + 2) * n], ds[(size_t)(i + 2) * ld_ds], a2);
        a3 = fmaf(att[pb + (i + 3) * n], ds[(size_t)(i + 3) * ld_ds], a3);
      }
      for (; i < n; ++i) a0 = fmaf(att[pb + i * n], ds[(size_t)i * ld_ds], a0);
      acc = (a0 + a1) + (a2 + a3);
    }
    float* d = dh + (size_t)j * ld_dh + k;
    *d = accumulate ? (*d + acc) : acc;
    return;
  }
  const long t = (long)(blockIdx.x - nb_dh) * 256 + threadIdx.x;
  if (t >= (long)b * (L2 + 1)) return;
  const int m = (int)(t / b), j = (int)(t % b);
  const int n = na[j];
  float acc = 0.f;
  if (n > 1) {
    const int s0 = s0a[j], pb = prow[s0] + (j - s0);
    const float* dsg = dsigma + pb;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = 0;
    if (m < L2) {
      const float* l2 = l2s + (size_t)m * P + pb;
      for (; i + 3 < n; i += 4) {
        a0 = fmaf(dsg[i * n], l2[i * n], a0);
        a1 = fmaf(dsg[(i + 1) * n], l2[(i + 1) * n], a1);
        a2 = fmaf(dsg[(i + 2) * n], l2[(i + 2) * n], a2);
        a3 = fmaf(dsg[(i + 3) * n], l2[(i + 3) * n], a3);
      }
      for (; i < n; ++i) a0 = fmaf(dsg[i * n], l2[i * n], a0);
    } else {
      for (; i < n; ++i) a0 += dsg[i * n];
    }
    acc = (a0 + a1) + (a2 + a3);
  }
  dvc[(size_t)j * (L2 + 1) + m] = acc;
}

// W3b[f][0..63] = W3[f][:], W3b[f][64] = b3[f]
__global__ void social_w3b_kernel(const float* __restrict__ W3, const float* __restrict__ b3, float* W3b, int F) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= F * (L2 + 1)) return;
  const int f = t / (L2 + 1), m = t % (L2 + 1);
  W3b[t] = m < L2 ? W3[f * L2 + m] : b3[f];
}

extern "C" {

int mggan_social_w3b(const float* W3, const float* b3, float* W3b, int F, hipStream_t stream) {
  MG_CHECK_ARG(W3 && b3 && W3b && F > 0, "social_w3b: bad arguments");
  hipLaunchKernelGGL(social_w3b_kernel, dim3(cdiv(F * (L2 + 1), 256)), dim3(256), 0, stream, W3, b3, W3b, F);
  MG_LAUNCH_CHECK("social_w3b");
  return MGGAN_OK;
}

int mggan_social_pairs_fwd(int P, const int* pair_i, const int* pair_j, const float* xy_last, const float* dxdy_last,
                           const float* W1, const float* b1, const float* W2, const float* b2, const float* vc,
                           float* feat, float* l1, float* l2, float* sigma, hipStream_t stream) {
  MG_CHECK_ARG(P >= 0, "social_pairs_fwd: negative pair count");
  if (P == 0) return MGGAN_OK;
  MG_CHECK_ARG(pair_i && pair_j && xy_last && dxdy_last && W1 && b1 && W2 && b2 && vc && sigma,
               "social_pairs_fwd: null pointer");
  MG_CHECK_ARG((feat == nullptr) == (l1 == nullptr) && (l1 == nullptr) == (l2 == nullptr),
               "social_pairs_fwd: save buffers must be all set or all NULL");
  hipLaunchKernelGGL(social_pairs_fwd_kernel, dim3(cdiv(P, 64)), dim3(256), 0, stream, P, pair_i, pair_j, xy_last,
                     dxdy_last, W1, b1, W2, b2, vc, feat, l1, l2, sigma);
  MG_LAUNCH_CHECK("social_pairs_fwd");
  return MGGAN_OK;
}

#define SOC_DISPATCH(KERNEL, H, ...)                                                                         \
  do {                                                                                                       \
    if ((H) == 32) hipLaunchKernelGGL((KERNEL<32>), dim3(cdiv(b, 8)), dim3(256), 0, stream, __VA_ARGS__);    \
    else hipLaunchKernelGGL((KERNEL<64>), dim3(cdiv(b, 4)), dim3(256), 0, stream, __VA_ARGS__);              \
  } while (0)

int mggan_social_softmax_fwd(int b, int H, const int* ped_prow, const int* ped_s0, const int* ped_n,
                             const float* sigma, const float* h, int ld_h, float* att, float* S, int ld_s,
                             hipStream_t stream) {
  MG_CHECK_ARG(H == 32 || H == 64, "social_softmax_fwd: hidden size %d not built (32 or 64)", H);
  if (b == 0) return MGGAN_OK;
  MG_CHECK_ARG(ped_prow && ped_s0 && ped_n && h && S, "social_softmax_fwd: null pointer");
  SOC_DISPATCH(social_softmax_fwd_kernel, H, b, ped_prow, ped_s0, ped_n, sigma, h, ld_h, att, S, ld_s);
  MG_LAUNCH_CHECK("social_softmax_fwd");
  return MGGAN_OK;
}

int mggan_social_softmax_bwd(int b, int H, const int* ped_prow, const int* ped_s0, const int* ped_n, const float* att,
                             const float* h, int ld_h, const float* dS, int ld_ds, float* dsigma, float* dh, int ld_dh,
                             int accumulate_dh, hipStream_t stream) {
  MG_CHECK_ARG(H == 32 || H == 64, "social_softmax_bwd: hidden size %d not built (32 or 64)", H);
  if (b == 0) return MGGAN_OK;
  MG_CHECK_ARG(ped_prow && ped_s0 && ped_n && h && dS && dh, "social_softmax_bwd: null pointer");
  SOC_DISPATCH(social_softmax_bwd_kernel, H, b, ped_prow, ped_s0, ped_n, att, h, ld_h, dS, ld_ds, dsigma);
  MG_LAUNCH_CHECK("social_softmax_bwd");
  SOC_DISPATCH(social_dh_kernel, H, b, ped_prow, ped_s0, ped_n, att, dS, ld_ds, dh, ld_dh, accumulate_dh);
  MG_LAUNCH_CHECK("social_dh");
  return MGGAN_OK;
}

int mggan_social_pairs_bwd(int P, int b, const int* pair_j, const int* ped_prow, const int* ped_s0, const int* ped_n,
                           const float* dsigma, const float* vc, const float* l1, const float* l2, const float* W2,
                           float* dz2, float* dz1, float* dvc, hipStream_t stream) {
  MG_CHECK_ARG(ped_prow && ped_s0 && ped_n && dvc, "social_pairs_bwd: null pointer");
  if (P > 0) {
    MG_CHECK_ARG(pair_j && dsigma && vc && l1 && l2 && W2 && dz2 && dz1, "social_pairs_bwd: null pointer");
    hipLaunchKernelGGL(social_pairs_bwd_kernel, dim3(cdiv(P, 64)), dim3(256), 0, stream, P, pair_j, dsigma, vc, l1, l2,
                       W2, dz2, dz1);
    MG_LAUNCH_CHECK("social_pairs_bwd");
  }
  if (b > 0) {
    hipLaunchKernelGGL(social_dvc_kernel, dim3(cdiv((long)b * (L2 + 1), 256)), dim3(256), 0, stream, b, P, ped_prow,
                       ped_s0, ped_n, dsigma, l2, dvc);
    MG_LAUNCH_CHECK("social_dvc");
  }
  return MGGAN_OK;
}

int mggan_social_attention_fwd(int n_tiles, const int* tiles, int P, int H, const int* pair_i, const int* pair_j,
                               const int* ped_prow, const int* ped_s0, const int* ped_n, const float* xy_last,
                               const float* dxdy_last, const float* W1, const float* b1, const float* W2,
                               const float* b2, const float* vc, const float* h, int ld_h, float* feat, float* l1,
                               float* l2, float* att, float* S, int ld_s, int xy_mod, hipStream_t stream) {
  MG_CHECK_ARG(H == 32 || H == 64, "social_attention_fwd: hidden size %d not built (32 or 64)", H);
  MG_CHECK_ARG(n_tiles >= 0 && P >= 0 && xy_mod >= 0, "social_attention_fwd: negative size");
  if (n_tiles == 0) return MGGAN_OK;
  MG_CHECK_ARG(tiles && ped_prow && ped_s0 && ped_n && h && S, "social_attention_fwd: null pointer");
  MG_CHECK_ARG(P == 0 || (pair_i && pair_j && xy_last && dxdy_last && W1 && b1 && W2 && b2 && vc && att),
               "social_attention_fwd: null pointer");
  MG_CHECK_ARG((feat == nullptr) == (l1 == nullptr) && (l1 == nullptr) == (l2 == nullptr),
               "social_attention_fwd: save buffers must be all set or all NULL");
  const int4* t4 = reinterpret_cast<const int4*>(tiles);
  if (H == 32)
    hipLaunchKernelGGL((social_fwd_fused_kernel<32>), dim3(n_tiles), dim3(256), 0, stream, t4, P, pair_i, pair_j,
                       ped_prow, ped_s0, ped_n, xy_last, dxdy_last, W1, b1, W2, b2, vc, h, ld_h, feat, l1, l2, att, S,
                       ld_s, xy_mod);
  else
    hipLaunchKernelGGL((social_fwd_fused_kernel<64>), dim3(n_tiles), dim3(256), 0, stream, t4, P, pair_i, pair_j,
                       ped_prow, ped_s0, ped_n, xy_last, dxdy_last, W1, b1, W2, b2, vc, h, ld_h, feat, l1, l2, att, S,
                       ld_s, xy_mod);
  MG_LAUNCH_CHECK("social_attention_fwd");
  return MGGAN_OK;
}

int mggan_social_attention_bwd(int n_tiles, const int* tiles, int P, int b, int H, const int* pair_i,
                               const int* pair_j, const int* ped_prow, const int* ped_s0, const int* ped_n,
                               const float* att, const float* h, int ld_h, const float* dS, int ld_ds,
                               const float* vc, const float* l1, const float* l2, const float* W2, float* dsigma,
                               float* dz2, float* dz1, float* dvc, float* dh, int ld_dh, int accumulate_dh,
                               hipStream_t stream) {
  MG_CHECK_ARG(H == 32 || H == 64, "social_attention_bwd: hidden size %d not built (32 or 64)", H);
  MG_CHECK_ARG(n_tiles >= 0 && P >= 0 && b >= 0, "social_attention_bwd: negative size");
  if (b == 0) return MGGAN_OK;
  MG_CHECK_ARG(tiles && ped_prow && ped_s0 && ped_n && h && dS && dh && dvc, "social_attention_bwd: null pointer");
  const int4* t4 = reinterpret_cast<const int4*>(tiles);
  if (P > 0) {
    MG_CHECK_ARG(pair_i && pair_j && att && vc && l1 && l2 && W2 && dsigma && dz2 && dz1,
                 "social_attention_bwd: null pointer");
    if (H == 32)
      hipLaunchKernelGGL((social_bwd_fused_kernel<32>), dim3(n_tiles), dim3(256), 0, stream, t4, P, pair_i, pair_j,
                         ped_prow, ped_n, att, h, ld_h, dS, ld_ds, vc, l1, l2, W2, dsigma, dz2, dz1);
    else
      hipLaunchKernelGGL((social_bwd_fused_kernel<64>), dim3(n_tiles), dim3(256), 0, stream, t4, P, pair_i, pair_j,
                         ped_prow, ped_n, att, h, ld_h, dS, ld_ds, vc, l1, l2, W2, dsigma, dz2, dz1);
    MG_LAUNCH_CHECK("social_attention_bwd");
  }
  const int nb_dh = cdiv(b, 256 / H), nb_dvc = cdiv((long)b * (L2 + 1), 256);
  if (H == 32)
    hipLaunchKernelGGL((social_dh_dvc_kernel<32>), dim3(nb_dh + nb_dvc), dim3(256), 0, stream, b, P, nb_dh, ped_prow,
                       ped_s0, ped_n, att, dS, ld_ds, dh, ld_dh, accumulate_dh, dsigma, l2, dvc);
  else
    hipLaunchKernelGGL((social_dh_dvc_kernel<64>), dim3(nb_dh + nb_dvc), dim3(256), 0, stream, b, P, nb_dh, ped_prow,
                       ped_s0, ped_n, att, dS, ld_ds, dh, ld_dh, accumulate_dh, dsigma, l2, dvc);
  MG_LAUNCH_CHECK("social_dh_dvc");
  return MGGAN_OK;
}

}  // extern "C"
