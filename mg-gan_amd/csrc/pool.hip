// Social-GAN pooling (--pool_type sgan) for MG-GAN on gfx950.
//
// Replaces PoolHiddenNet.forward (/root/reference/mggan/model/modules/social_gan.py:199-229): per scene every
// pedestrian i looks at every j of its scene (itself included): rel = p_j - p_i -> Linear(2,E); [emb | h_j] ->
// Linear -> ReLU -> Linear (the fused MLP chain of mlp.hip) ; max over j.  The reference tiles and concatenates
// dense (n*n) blocks per scene in a Python loop; here the pairs of all scenes are one row list.
//   pair p = (out row pair_o[p], position rows pair_i[p] / pair_j[p], hidden row pair_j[p]); the pairs of an output
//   row are contiguous: [ped_prow[o], ped_prow[o] + ped_n[o]).
#include "common.h"
#include "../../include/mggan_hip.h"

// X[p] = [We (xy[j] - xy[i]) + be | h[j]], rel[p] = xy[j] - xy[i]
__global__ void pool_pairs_fwd_kernel(int P, const int* __restrict__ pair_i, const int* __restrict__ pair_j,
                                      const float* __restrict__ xy, int xy_mod, const float* __restrict__ We,
                                      const float* __restrict__ be, int E, const float* __restrict__ h, int ld_h, int H,
                                      float* __restrict__ X, float* __restrict__ rel) {
  const int W = E + H;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)P * W) return;
  const int p = (int)(t / W), c = (int)(t % W);
  const int j = pair_j[p];
  if (c >= E) {
    X[t] = h[(size_t)j * ld_h + (c - E)];
    return;
  }
  const int i = pair_i[p];
  const int xi = xy_mod > 0 ? i % xy_mod : i, xj = xy_mod > 0 ? j % xy_mod : j;
  const float rx = xy[2 * xj] - xy[2 * xi], ry = xy[2 * xj + 1] - xy[2 * xi + 1];
  X[t] = fmaf(We[2 * c], rx, fmaf(We[2 * c + 1], ry, be[c]));
  if (c == 0) { rel[2 * p] = rx; rel[2 * p + 1] = ry; }
}

// dh[j][k] = sum over the pairs that read h_j of dX[p][E + k]; hid_ptr / hid_pairs: CSR list of those pairs per
// hidden row (fixed order -> deterministic)
__global__ void pool_gather_bwd_kernel(int b, int H, int E, const int* __restrict__ hid_ptr,
                                       const int* __restrict__ hid_pairs, const float* __restrict__ dX, float* dh,
                                       int ld_dh) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)b * H) return;
  const int j = (int)(t / H), k = (int)(t % H);
  float acc = 0.f;
  for (int q = hid_ptr[j]; q < hid_ptr[j + 1]; ++q) acc += dX[(size_t)hid_pairs[q] * (E + H) + E + k];
  dh[(size_t)j * ld_dh + k] = acc;
}

// out[o][c] = max_j Y[prow[o] + j][c] (first maximum), arg[o][c] = that j
__global__ void segment_max_fwd_kernel(int rows, int B, const int* __restrict__ prow, const int* __restrict__ na,
                                       const float* __restrict__ Y, float* out, int* arg) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)rows * B) return;
  const int o = (int)(t / B), c = (int)(t % B);
  const int p0 = prow[o], n = na[o];
  float best = Y[(size_t)p0 * B + c];
  int bj = 0;
  for (int j = 1; j < n; ++j) {
    const float v = Y[(size_t)(p0 + j) * B + c];
    if (v > best) { best = v; bj = j; }
  }
  out[t] = best;
  arg[t] = bj;
}

// dY[p][c] = dOut[o][c] if p is the arg-max row of (o, c), else 0
__global__ void segment_max_bwd_kernel(int P, int B, const int* __restrict__ pair_o, const int* __restrict__ prow,
                                       const float* __restrict__ dOut, int ld_dout, const int* __restrict__ arg,
                                       float* dY) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)P * B) return;
  const int p = (int)(t / B), c = (int)(t % B);
  const int o = pair_o[p];
  dY[t] = (arg[(size_t)o * B + c] == p - prow[o]) ? dOut[(size_t)o * ld_dout + c] : 0.f;
}

extern "C" {

int mggan_pool_pairs_fwd(int P, const int* pair_i, const int* pair_j, const float* xy_last, int xy_mod,
                         const float* We, const float* be, int E, const float* h, int ld_h, int H, float* X, float* rel,
                         hipStream_t stream) {
  MG_CHECK_ARG(P >= 0 && E > 0 && H > 0 && xy_mod >= 0, "pool_pairs_fwd: bad sizes");
  if (P == 0) return MGGAN_OK;
  MG_CHECK_ARG(pair_i && pair_j && xy_last && We && be && h && X && rel, "pool_pairs_fwd: null pointer");
  MG_LAUNCH(pool_pairs_fwd_kernel, dim3(cdiv((long)P * (E + H), 256)), dim3(256), 0, stream, P, pair_i, pair_j,
                     xy_last, xy_mod, We, be, E, h, ld_h, H, X, rel);
  MG_LAUNCH_CHECK("pool_pairs_fwd");
  return MGGAN_OK;
}

int mggan_pool_gather_bwd(int b, int H, int E, const int* hid_ptr, const int* hid_pairs, const float* dX, float* dh,
                          int ld_dh, hipStream_t stream) {
  if ((long)b * H == 0) return MGGAN_OK;
  MG_CHECK_ARG(hid_ptr && hid_pairs && dX && dh, "pool_gather_bwd: null pointer");
  MG_LAUNCH(pool_gather_bwd_kernel, dim3(cdiv((long)b * H, 256)), dim3(256), 0, stream, b, H, E, hid_ptr,
                     hid_pairs, dX, dh, ld_dh);
  MG_LAUNCH_CHECK("pool_gather_bwd");
  return MGGAN_OK;
}

int mggan_segment_max_fwd(int rows, int B, const int* ped_prow, const int* ped_n, const float* Y, float* out, int* arg,
                          hipStream_t stream) {
  if ((long)rows * B == 0) return MGGAN_OK;
  MG_CHECK_ARG(ped_prow && ped_n && Y && out && arg, "segment_max_fwd: null pointer");
  MG_LAUNCH(segment_max_fwd_kernel, dim3(cdiv((long)rows * B, 256)), dim3(256), 0, stream, rows, B, ped_prow,
                     ped_n, Y, out, arg);
  MG_LAUNCH_CHECK("segment_max_fwd");
  return MGGAN_OK;
}

int mggan_segment_max_bwd(int P, int B, const int* pair_o, const int* ped_prow, const float* dOut, int ld_dout,
                          const int* arg, float* dY, hipStream_t stream) {
  if ((long)P * B == 0) return MGGAN_OK;
  MG_CHECK_ARG(pair_o && ped_prow && dOut && arg && dY, "segment_max_bwd: null pointer");
  MG_LAUNCH(segment_max_bwd_kernel, dim3(cdiv((long)P * B, 256)), dim3(256), 0, stream, P, B, pair_o, ped_prow,
                     dOut, ld_dout, arg, dY);
  MG_LAUNCH_CHECK("segment_max_bwd");
  return MGGAN_OK;
}

}  // extern "C"
