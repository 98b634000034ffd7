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
// These are the UNFUSED per-stage kernels (any scene size); scenes of up to 64 pedestrians -- every batch of the
// benchmark configurations -- take the row-structured MFMA kernels of social_rows.hip instead.
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
  MG_LAUNCH(social_w3b_kernel, dim3(cdiv(F * (L2 + 1), 256)), dim3(256), 0, stream, W3, b3, W3b, F);
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
  MG_LAUNCH(social_pairs_fwd_kernel, dim3(cdiv(P, 64)), dim3(256), 0, stream, P, pair_i, pair_j, xy_last,
                     dxdy_last, W1, b1, W2, b2, vc, feat, l1, l2, sigma);
  MG_LAUNCH_CHECK("social_pairs_fwd");
  return MGGAN_OK;
}

#define SOC_DISPATCH(KERNEL, H, ...)                                                                         \
  do {                                                                                                       \
    if ((H) == 32) MG_LAUNCH((KERNEL<32>), dim3(cdiv(b, 8)), dim3(256), 0, stream, __VA_ARGS__);    \
    else MG_LAUNCH((KERNEL<64>), dim3(cdiv(b, 4)), dim3(256), 0, stream, __VA_ARGS__);              \
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
    MG_LAUNCH(social_pairs_bwd_kernel, dim3(cdiv(P, 64)), dim3(256), 0, stream, P, pair_j, dsigma, vc, l1, l2,
                       W2, dz2, dz1);
    MG_LAUNCH_CHECK("social_pairs_bwd");
  }
  if (b > 0) {
    MG_LAUNCH(social_dvc_kernel, dim3(cdiv((long)b * (L2 + 1), 256)), dim3(256), 0, stream, b, P, ped_prow,
                       ped_s0, ped_n, dsigma, l2, dvc);
    MG_LAUNCH_CHECK("social_dvc");
  }
  return MGGAN_OK;
}

}  // extern "C"
