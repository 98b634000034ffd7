// Row plumbing of the discriminator's per-(pedestrian, sample) pass (discriminators.py:113-219 of the reference):
// the classifier input X (K*b, 192) = [soc 64 | in_enc 32 | pred_enc 32 | scene 64] is ONE buffer that every
// producer writes into through a column offset / row stride (the MLP chain, the social attention) -- no torch.cat,
// .repeat or slice copies -- and these two kernels do what is left: broadcast the per-pedestrian context into the
// sample blocks, clear the social columns of the blocks that have none, and turn time-major rollout steps
// into rows and back.
#include "common.h"
#include "../../include/mggan_hip.h"

// X[k*b+ped][c_in : c_in+w_in] = in_enc[ped], X[..][c_scene : +w_scene] = scene[ped]; X[k*b+ped][0:w_soc] = 0 for
// k >= soc_blocks.  One lane per (row, 4 floats).
__global__ __launch_bounds__(256) void d_rows_fill_kernel(int b, int K, int soc_blocks, int w_soc, int c_in, int w_in,
                                                          int c_scene, int w_scene, const float* __restrict__ in_enc,
                                                          int ld_in, const float* __restrict__ scene, int ld_scene,
                                                          float* X, int ldx) {
  const int q_soc = w_soc / 4, q_in = w_in / 4, q_sc = w_scene / 4, per = q_soc + q_in + q_sc;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)K * b * per) return;
  const int q = (int)(i % per);
  const long r = i / per;
  const int ped = (int)(r % b), k = (int)(r / b);
  float* row = X + (size_t)r * ldx;
  if (q < q_soc) {
    if (k >= soc_blocks) *reinterpret_cast<float4*>(row + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
  } else if (q < q_soc + q_in) {
    const int c = 4 * (q - q_soc);
    *reinterpret_cast<float4*>(row + c_in + c) = *reinterpret_cast<const float4*>(in_enc + (size_t)ped * ld_in + c);
  } else {
    const int c = 4 * (q - q_soc - q_in);
    *reinterpret_cast<float4*>(row + c_scene + c) = *reinterpret_cast<const float4*>(scene + (size_t)ped * ld_scene + c);
  }
}

// adjoint of the broadcast: din[ped] = sum_k dX[k*b+ped][c_in:], dscene[ped] = sum_k dX[..][c_scene:]  (fixed order)
__global__ __launch_bounds__(256) void d_rows_reduce_kernel(int b, int K, int c_in, int w_in, int c_scene, int w_scene,
                                                            const float* __restrict__ dX, int ldx, float* din, int ld_in,
                                                            float* dscene, int ld_scene) {
  const int per = w_in + w_scene;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)b * per) return;
  const int c = (int)(i % per), ped = (int)(i / per);
  const bool is_in = c < w_in;
  float* dst = is_in ? din : dscene;
  if (!dst) return;
  const int col = is_in ? c_in + c : c_scene + (c - w_in);
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += dX[((size_t)k * b + ped) * ldx + col];
  if (is_in) dst[(size_t)ped * ld_in + c] = s;
  else dst[(size_t)ped * ld_scene + (c - w_in)] = s;
}

// rows (n, 2T) with row stride ld -> time-major steps (T, n, 2): the adjoint of steps_to_rows
// (n rows into a step tensor of n_out >= n rows per step: the first n rows of every step are written)
__global__ __launch_bounds__(256) void rows_to_steps_kernel(const float* __restrict__ rows, int ld, int T, int n, int n_out,
                                                            float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n * T) return;
  const int r = (int)(i / T), t = (int)(i % T);
  *reinterpret_cast<float2*>(out + ((size_t)t * n_out + r) * 2) = *reinterpret_cast<const float2*>(rows + (size_t)r * ld + 2 * t);
}

extern "C" {

int mggan_d_rows_fill(int b, int K, int soc_blocks, int w_soc, int c_in, int w_in, int c_scene, int w_scene,
                      const float* in_enc, int ld_in, const float* scene, int ld_scene, float* X, int ldx,
                      hipStream_t stream) {
  if ((long)b * K == 0) return MGGAN_OK;
  MG_CHECK_ARG(X && (in_enc || w_in == 0) && (scene || w_scene == 0), "d_rows_fill: null pointer");
  MG_CHECK_ARG(w_soc % 4 == 0 && w_in % 4 == 0 && w_scene % 4 == 0 && c_in % 4 == 0 && c_scene % 4 == 0 && ldx % 4 == 0 &&
                   ld_in % 4 == 0 && ld_scene % 4 == 0,
               "d_rows_fill: widths, column offsets and row strides must be multiples of 4 floats");
  const long n = (long)K * b * ((w_soc + w_in + w_scene) / 4);
  if (n == 0) return MGGAN_OK;
  MG_LAUNCH(d_rows_fill_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, b, K, soc_blocks, w_soc, c_in, w_in,
                     c_scene, w_scene, in_enc, ld_in, scene, ld_scene, X, ldx);
  MG_LAUNCH_CHECK("d_rows_fill");
  return MGGAN_OK;
}

int mggan_d_rows_reduce(int b, int K, int c_in, int w_in, int c_scene, int w_scene, const float* dX, int ldx, float* din,
                        int ld_in, float* dscene, int ld_scene, hipStream_t stream) {
  if (b == 0) return MGGAN_OK;
  MG_CHECK_ARG(dX, "d_rows_reduce: null pointer");
  const long n = (long)b * (w_in + w_scene);
  if (n == 0) return MGGAN_OK;
  MG_LAUNCH(d_rows_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, b, K, c_in, w_in, c_scene, w_scene, dX,
                     ldx, din, ld_in, dscene, ld_scene);
  MG_LAUNCH_CHECK("d_rows_reduce");
  return MGGAN_OK;
}

int mggan_rows_to_steps(const float* rows, int ld, int T, int n, float* out, hipStream_t stream) {
  if ((long)n * T == 0) return MGGAN_OK;
  MG_CHECK_ARG(rows && out && ld >= 2 * T && ld % 2 == 0, "rows_to_steps: bad arguments");
  MG_LAUNCH(rows_to_steps_kernel, dim3(cdiv((long)n * T, 256)), dim3(256), 0, stream, rows, ld, T, n, n, out);
  MG_LAUNCH_CHECK("rows_to_steps");
  return MGGAN_OK;
}

/* rows (n, ld >= 2T) -> the first n rows of every step of out (T, n_out, 2) */
int mggan_rows_to_steps_n(const float* rows, int ld, int T, int n, int n_out, float* out, hipStream_t stream) {
  if ((long)n * T == 0) return MGGAN_OK;
  MG_CHECK_ARG(rows && out && ld >= 2 * T && ld % 2 == 0 && n <= n_out, "rows_to_steps_n: bad arguments");
  MG_LAUNCH(rows_to_steps_kernel, dim3(cdiv((long)n * T, 256)), dim3(256), 0, stream, rows, ld, T, n, n_out, out);
  MG_LAUNCH_CHECK("rows_to_steps_n");
  return MGGAN_OK;
}

}  // extern "C"
