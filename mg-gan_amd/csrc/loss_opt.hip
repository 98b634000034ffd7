// GAN losses (+ their gradients), gradient-norm clipping and AdamW for MG-GAN on gfx950.
//
// Replaces (file:line under /root/reference/mggan):
//   BCE with scalar smoothed labels       abstract_train.py:62-67 (phi_1..3, 'NS'), utils.py:18-25
//   D output epsilon-affine               model/modules/discriminators.py:203-204 (p = sigmoid(z)(1-2e-7)+1e-7)
//   1/count(generator) re-weighting, CE   model/train.py:92-113,181-186
//   per-scene min-over-K L2               model/train.py:58-75
//   PM-network 'ml' target                model/train.py:626-639
//   clip_grad_norm_ + AdamW               model/train.py:131-135,209-213,656-658; abstract_train.py:45-50
// All reductions are fixed-order (deterministic); every loss kernel also emits the gradient
// w.r.t. its input so logits never round-trip through autograd.
#include <string.h>
#include "common.h"
#include "comm_dev.h"
#include "../../include/mggan_hip.h"

// ---- BCE on the discriminator output ---------------------------------------------------
// p: D output per row (already sigmoid + eps-affine). loss_r = w_r*scale*BCE(p,y); dp = dloss_r/dp.
__global__ void bce_rows_kernel(int rows, int kind, const float* __restrict__ p_in, float y, const float* __restrict__ y_u,
                                float y_lo, float y_hi, float scale, const int* __restrict__ row_gen,
                                const float* __restrict__ inv_count, float* loss_rows, float* dp) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  if (y_u) y = y_lo + (y_hi - y_lo) * (*y_u);  // smoothed label drawn on the device: U(y_lo, y_hi)
  const float p = p_in[r];
  const float w = (row_gen ? inv_count[row_gen[r]] : 1.f) * scale;
  if (kind == 1) {  // least-squares objective (abstract_train.py:72-75): MSELoss(reduction='none')
    loss_rows[r] = w * (p - y) * (p - y);
    if (dp) dp[r] = w * 2.f * (p - y);
    return;
  }
  const float lp = fmaxf(__logf(p), -100.f), lq = fmaxf(__logf(1.f - p), -100.f);  // BCELoss log clamp
  loss_rows[r] = -w * (y * lp + (1.f - y) * lq);
  if (dp) dp[r] = -w * (y / p - (1.f - y) / (1.f - p));
}

// out[r][2t+c] = (r < n ? a : b)[t][r mod n][c]
__global__ void steps_to_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, int T, int n, int rows,
                                     float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * T) return;
  const int r = (int)(i / T), t = (int)(i % T);
  const float* src = r < n ? a : b;
  const float2 v = *reinterpret_cast<const float2*>(src + ((size_t)t * n + (r < n ? r : r - n)) * 2);
  *reinterpret_cast<float2*>(out + ((size_t)r * T + t) * 2) = v;
}

__global__ void scale_kernel(float* x, long n, const float* __restrict__ s) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= *s;
}

// X[k*b+ped] = [ soc0[ped] if k==0 else 0 | in_enc[ped] | pred_enc[k*b+ped] | scene[ped] ]
__global__ void d_assemble_kernel(int b, int K, int ws, int wi, int wp, int wc, int soc_all,
                                  const float* __restrict__ soc0, const float* __restrict__ in_enc,
                                  const float* __restrict__ pred_enc, const float* __restrict__ scene, float* X) {
  const int W = ws + wi + wp + wc;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)K * b * W) return;
  const int c = (int)(i % W);
  const long r = i / W;
  const int ped = (int)(r % b), k = (int)(r / b);
  float v;
  if (c < ws) v = soc_all == 1 ? soc0[(size_t)r * ws + c] : ((k == 0 || soc_all == 2) ? soc0[(size_t)ped * ws + c] : 0.f);
  else if (c < ws + wi) v = in_enc[(size_t)ped * wi + (c - ws)];
  else if (c < ws + wi + wp) v = pred_enc[(size_t)r * wp + (c - ws - wi)];
  else v = scene[(size_t)ped * wc + (c - ws - wi - wp)];
  X[i] = v;
}

// adjoint: dsoc0[ped] = dX[ped][:ws]; din_enc[ped] = sum_k dX[k*b+ped][ws:ws+wi]; dpred = copy; dscene = sum_k
__global__ void d_assemble_bwd_kernel(int b, int K, int ws, int wi, int wp, int wc, int soc_all,
                                      const float* __restrict__ dX, float* dsoc0, float* din_enc, float* dpred_enc,
                                      float* dscene) {
  const int W = ws + wi + wp + wc;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)b * W) return;
  const int c = (int)(i % W), ped = (int)(i / W);
  if (c < ws) {
    if (dsoc0 && soc_all == 2) {
      float s = 0.f;
      for (int k = 0; k < K; ++k) s += dX[((size_t)k * b + ped) * W + c];
      dsoc0[(size_t)ped * ws + c] = s;
    } else if (dsoc0) {
      for (int k = 0; k < (soc_all ? K : 1); ++k)
        dsoc0[((size_t)k * b + ped) * ws + c] = dX[((size_t)k * b + ped) * W + c];
    }
  } else if (c >= ws + wi && c < ws + wi + wp) {
    if (dpred_enc)
      for (int k = 0; k < K; ++k) dpred_enc[((size_t)k * b + ped) * wp + (c - ws - wi)] = dX[((size_t)k * b + ped) * W + c];
  } else {
    float* dst = c < ws + wi ? din_enc : dscene;
    if (!dst) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += dX[((size_t)k * b + ped) * W + c];
    if (c < ws + wi) dst[(size_t)ped * wi + (c - ws)] = s;
    else dst[(size_t)ped * wc + (c - ws - wi - wp)] = s;
  }
}

// ---- cross entropy per row (generator-id head) ------------------------------------------
__global__ void ce_rows_kernel(int rows, int g, const float* __restrict__ logits, int ld, const int* __restrict__ target,
                               const float* __restrict__ inv_count, float scale, float* loss_rows, float* dlogits,
                               int ldd) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* l = logits + (size_t)r * ld;
  const int t = target[r];
  float mx = -INFINITY;
  for (int c = 0; c < g; ++c) mx = fmaxf(mx, l[c]);
  float den = 0.f;
  for (int c = 0; c < g; ++c) den += __expf(l[c] - mx);
  const float lse = mx + __logf(den);
  const float w = (inv_count ? inv_count[t] : 1.f) * scale;
  loss_rows[r] = w * (lse - l[t]);
  if (dlogits)
    for (int c = 0; c < g; ++c) dlogits[(size_t)r * ldd + c] = w * (__expf(l[c] - lse) - (c == t ? 1.f : 0.f));
}

// ---- all adversarial losses of one optimizer step in ONE launch ---------------------------------
// The discriminator step has three loss terms (real BCE, fake BCE, generator-id CE), the generator step two
// (adversarial BCE, classifier CE with 1/count(generator) row weights); as separate rows + sum launches they are
// 6-8 dependent few-microsecond kernels on the critical chain.  Up to GAN_LOSS_MAX_WG 1024-thread workgroups take
// the rows grid-strided and accumulate the per-term sums in double; the last workgroup to finish adds the
// per-workgroup sums in index order, so the result is deterministic.
#define GAN_LOSS_MAX_WG 256
struct GanLossArgs {
  const float* p;         // (nA + nB) discriminator outputs: rows [0,nA) term A, [nA,nA+nB) term B
  const float* label_u[2];  // device uniform draw per term (label = lo + (hi-lo)*u) or NULL -> label[]
  const int* row_gen;     // generator id per row (terms A and C weighting) or NULL
  const int* seg;         // g+1 segment offsets of the rows sorted by generator (count_q = seg[q+1]-seg[q]) or NULL
  const float* inv_count; // 1/count per generator (used when seg == NULL) or NULL
  const float* logits;    // (nC x g) classifier logits, row stride ld
  const int* target;      // nC class ids
  float* dp;              // (nA + nB) gradient wrt p
  float* dlogits;         // (nC x g) gradient wrt logits (already times grad_c)
  float* out[3];          // where each term's value goes (NULL: term absent)
  float* total;           // A + B + grad_c * C
  double* partial;        // 3 * GAN_LOSS_MAX_WG doubles of scratch
  unsigned* ticket;       // one zero-initialised word; the kernel leaves it at zero again
  const int* dims;        // padded batch (common.h): rows whose pedestrian (row % bmod) is a phantom take no part, the
                          // host-side normalisers (of the padded counts) are corrected; NULL: every row is real
  float label[2], lo[2], hi[2], scale[3], sign_a, grad_c;
  int nA, nB, nC, g, ld, kind, weighted_c, bmod;
};

__device__ __forceinline__ float bce_term(int kind, float p, float y, float w, float* dp) {
  if (kind == 1) {  // least-squares objective (abstract_train.py:72-75)
    *dp = w * 2.f * (p - y);
    return w * (p - y) * (p - y);
  }
  const float lp = fmaxf(__logf(p), -100.f), lq = fmaxf(__logf(1.f - p), -100.f);  // BCELoss log clamp
  *dp = -w * (y / p - (1.f - y) / (1.f - p));
  return -w * (y * lp + (1.f - y) * lq);
}

__global__ __launch_bounds__(1024) void gan_losses_kernel(GanLossArgs a) {
  __shared__ double red[3][16];
  __shared__ float invc[256];
  __shared__ int last;
  const int t = threadIdx.x;
  const bool weighted = a.row_gen != nullptr;
  // Everything this thread reads from memory for its first rows is asked for before the barrier below: labels, the
  // padded-batch record, p, the row's generator.  (Behind the barrier they were a second and a third round trip of a
  // launch that is little else: 12 us for 2,560 rows.)
  float y[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) y[q] = a.label_u[q] ? a.lo[q] + (a.hi[q] - a.lo[q]) * (*a.label_u[q]) : a.label[q];
  const int first = blockIdx.x * 1024 + t, step = gridDim.x * 1024;
  const int bm = a.bmod > 0 ? a.bmod : 1, nr = a.dims ? mg_real_rows(a.dims, bm) : bm;  // row r <-> pedestrian r % bm
  const float corr = mg_pad_corr(a.dims);
  const float pA0 = first < a.nA ? a.p[first] : 0.5f, pB0 = first < a.nB ? a.p[a.nA + first] : 0.5f;
  const int gA0 = weighted && first < a.nA ? a.row_gen[first] : 0;
  if (weighted && t < a.g) {
    if (a.seg) {
      const int c = a.seg[t + 1] - a.seg[t];
      invc[t] = c > 0 ? 1.f / (float)c : 0.f;
    } else {
      invc[t] = a.inv_count[t];
    }
  }
  __syncthreads();
  double acc[3] = {0.0, 0.0, 0.0};
  for (int r = first; r < a.nA; r += step) {
    if (a.dims && r % bm >= nr) { a.dp[r] = 0.f; continue; }
    const float w = (weighted ? invc[r == first ? gA0 : a.row_gen[r]] : 1.f) * (a.scale[0] * corr) * a.sign_a;
    float d;
    acc[0] += (double)bce_term(a.kind, r == first ? pA0 : a.p[r], y[0], w, &d);
    a.dp[r] = d;
  }
  for (int r = first; r < a.nB; r += step) {
    if (a.dims && r % bm >= nr) { a.dp[a.nA + r] = 0.f; continue; }
    float d;
    acc[1] += (double)bce_term(a.kind, r == first ? pB0 : a.p[a.nA + r], y[1], a.scale[1] * corr, &d);
    a.dp[a.nA + r] = d;
  }
  for (int r = first; r < a.nC; r += step) {
    if (a.dims && r % bm >= nr) {
      for (int c = 0; c < a.g; ++c) a.dlogits[(size_t)r * a.g + c] = 0.f;
      continue;
    }
    const float* l = a.logits + (size_t)r * a.ld;
    const int tg = a.target[r];
    float mx = -INFINITY;
    for (int c = 0; c < a.g; ++c) mx = fmaxf(mx, l[c]);
    float den = 0.f;
    for (int c = 0; c < a.g; ++c) den += __expf(l[c] - mx);
    const float lse = mx + __logf(den);
    const float w = (a.weighted_c ? invc[tg] : 1.f) * (a.scale[2] * corr);
    acc[2] += (double)(w * (lse - l[tg]));
    for (int c = 0; c < a.g; ++c)
      a.dlogits[(size_t)r * a.g + c] = a.grad_c * (w * (__expf(l[c] - lse) - (c == tg ? 1.f : 0.f)));
  }
  // workgroup sums: a fixed shuffle tree per wave, the 16 wave sums through LDS, the same tree again (two barriers; the
  // ten-round LDS tree of 1,024 doubles this replaces was a third of the launch)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] += __shfl_xor(acc[q], o, 64);
  if ((t & 63) == 0) {
#pragma unroll
    for (int q = 0; q < 3; ++q) red[q][t >> 6] = acc[q];
  }
  __syncthreads();
  if (t < 64) {
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = t < 16 ? red[q][t] : 0.0;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1)
#pragma unroll
      for (int q = 0; q < 3; ++q) acc[q] += __shfl_xor(acc[q], o, 64);
  }
  if (gridDim.x == 1) {  // one workgroup: nothing to meet
    if (t != 0) return;
    const float A = (float)acc[0], B = (float)acc[1], C = (float)acc[2];
    if (a.out[0]) *a.out[0] = A;
    if (a.out[1]) *a.out[1] = B;
    if (a.out[2]) *a.out[2] = C;
    *a.total = (a.nA ? A : 0.f) + (a.nB ? B : 0.f) + (a.nC ? a.grad_c * C : 0.f);
    return;
  }
  // per-workgroup sums meet in `partial`; the workgroup that takes the last ticket adds them in index order
  // (a fixed order: the result does not depend on which workgroup finishes last) and re-arms the ticket
  if (t == 0) {
#pragma unroll
    for (int q = 0; q < 3; ++q) a.partial[blockIdx.x * 3 + q] = acc[q];
    __threadfence();
    last = atomicAdd(a.ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last || t >= 64) return;
  __threadfence();
  // the first wave of the last workgroup: lane l takes the workgroups l, l + 64, ... in index order, then a fixed
  // shuffle tree (a lone lane walking 3 x gridDim dependent loads was most of this launch at 160k rows)
  double tot[3] = {0.0, 0.0, 0.0};
  for (unsigned b = t; b < gridDim.x; b += 64)
#pragma unroll
    for (int q = 0; q < 3; ++q) tot[q] += __hip_atomic_load(a.partial + b * 3 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int q = 0; q < 3; ++q) tot[q] += __shfl_xor(tot[q], o, 64);
  if (t != 0) return;
  const float A = (float)tot[0], B = (float)tot[1], C = (float)tot[2];
  if (a.out[0]) *a.out[0] = A;
  if (a.out[1]) *a.out[1] = B;
  if (a.out[2]) *a.out[2] = C;
  *a.total = (a.nA ? A : 0.f) + (a.nB ? B : 0.f) + (a.nC ? a.grad_c * C : 0.f);
  *a.ticket = 0u;
}

// ---- per-scene min-over-K L2 (train.py:58-75) -------------------------------------------
// One 256-thread workgroup per scene: eight 32-lane groups take the samples k round-robin, the lanes of a
// group take the scene's pedestrians (contiguous 8-byte reads), each lane sums its |abs - gt| over the T steps,
// a fixed-order shuffle tree adds the pedestrians.  Wave 0 then takes the first minimum over k.
#define L2_MAXK 256
__global__ __launch_bounds__(256) void l2_scene_kernel(int S, int T, int K, int b, const int* __restrict__ scenes,
                                                       const float* __restrict__ gen_abs, const float* __restrict__ gt,
                                                       float* scene_loss, int* scene_arg, const int* dims) {
  __shared__ float ksum[L2_MAXK];
  const int s = blockIdx.x, grp = threadIdx.x >> 5, l = threadIdx.x & 31;
  if (s >= mg_real_scenes(dims, S)) {  // a phantom scene of a padded batch: no loss, no sample carries a gradient
    if (threadIdx.x == 0) { scene_loss[s] = 0.f; scene_arg[s] = -1; }
    return;
  }
  const int p0 = scenes[2 * s], p1 = scenes[2 * s + 1];
  for (int k = grp; k < K; k += 8) {
    float acc = 0.f;
    auto dist = [&](int t, int ped) {
      const float2 a = *reinterpret_cast<const float2*>(gen_abs + (((size_t)t * K + k) * b + ped) * 2);
      const float2 g = *reinterpret_cast<const float2*>(gt + ((size_t)t * b + ped) * 2);
      const float dx = a.x - g.x, dy = a.y - g.y;
      return sqrtf(dx * dx + dy * dy);
    };
    for (int ped = p0 + l; ped < p1; ped += 32) {  // four steps at a time: eight loads in flight, same summation order
      int t = 0;
      for (; t + 3 < T; t += 4) {
        const float d0 = dist(t, ped), d1 = dist(t + 1, ped), d2 = dist(t + 2, ped), d3 = dist(t + 3, ped);
        acc += d0; acc += d1; acc += d2; acc += d3;
      }
      for (; t < T; ++t) acc += dist(t, ped);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (l == 0) ksum[k] = acc;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    float best = INFINITY;
    int bestk = 0;
    for (int k0 = 0; k0 < K; k0 += 64) {
      float v = k0 + lane < K ? ksum[k0 + lane] : INFINITY;
      int vi = k0 + lane;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(vi, o, 64);
        if (ov < v || (ov == v && oi < vi)) { v = ov; vi = oi; }
      }
      if (v < best) { best = v; bestk = vi; }
    }
    if (lane == 0) { scene_loss[s] = best * mg_pad_corr(dims); scene_arg[s] = bestk; }  // (the sum is divided by b_pad)
  }
}

// gabs[t][k][ped] = scale * (abs-gt)/|abs-gt| if k == argmin(scene(ped)) else 0
__global__ void l2_grad_kernel(int T, int K, int b, const int* __restrict__ ped_scene, const int* __restrict__ scene_arg,
                               const float* __restrict__ gen_abs, const float* __restrict__ gt, float scale,
                               float* gabs, const int* dims) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)T * K * b) return;
  scale *= mg_pad_corr(dims);
  const int ped = (int)(i % b), k = (int)((i / b) % K), t = (int)(i / ((long)b * K));
  float gx = 0.f, gy = 0.f;
  if (scene_arg[ped_scene[ped]] == k) {
    const float dx = gen_abs[i * 2] - gt[((size_t)t * b + ped) * 2], dy = gen_abs[i * 2 + 1] - gt[((size_t)t * b + ped) * 2 + 1];
    const float n = sqrtf(dx * dx + dy * dy);
    if (n > 0.f) { gx = scale * dx / n; gy = scale * dy / n; }
  }
  gabs[i * 2] = gx;
  gabs[i * 2 + 1] = gy;
}

// ---- PM-network 'l2' / 'endpoint' targets (train.py:616-624,641-647) ---------------------------------
// target[ped] = argmin_g min_E dist(g, e), dist = mean_t |abs - gt| ('l2', mode 0) or |abs[T-1] - gt[T-1]|
// ('endpoint', mode 1); first minimum wins like torch.argmin.  gen_abs (T,E,g,b,2).
__global__ void pm_target_kernel(int b, int T, int E, int g, int mode, const float* __restrict__ gen_abs,
                                 const float* __restrict__ gt, int* __restrict__ target) {
  const int ped = blockIdx.x * blockDim.x + threadIdx.x;
  if (ped >= b) return;
  float best = INFINITY;
  int arg = 0;
  for (int gi = 0; gi < g; ++gi) {
    float dmin = INFINITY;
    for (int e = 0; e < E; ++e) {
      float d = 0.f;
      for (int t = mode ? T - 1 : 0; t < T; ++t) {
        const float* a = gen_abs + ((((size_t)t * E + e) * g + gi) * b + ped) * 2;
        const float dx = a[0] - gt[((size_t)t * b + ped) * 2], dy = a[1] - gt[((size_t)t * b + ped) * 2 + 1];
        d += sqrtf(dx * dx + dy * dy);
      }
      if (!mode) d /= (float)T;
      dmin = fminf(dmin, d);
    }
    if (dmin < best) { best = dmin; arg = gi; }
  }
  target[ped] = arg;
}

// ---- PM-network 'mgan' target (train.py:606-614) ------------------------------------------------------
// As written in the reference the target softmax runs over the singleton sample axis of branch_out (b,1,g), so
// every target is 1 and the product broadcasts over the batch: loss = -(1/g) sum_{r,j} log p_rj - reg * mean_r H(p_r),
// reg = 0.9^epoch, p = softmax(logits).  Per row: loss_r = -tw sum_j log p_j + reg sum_j p_j log p_j with tw = b/g;
// dlogits_k = scale * [ tw (g p_k - 1) + reg p_k (log p_k - sum_j p_j log p_j) ].
__global__ void pm_mgan_kernel(int b, int g, const float* __restrict__ logits, float tw, float reg_arg,
                               const float* __restrict__ reg_dev, float scale, float* loss_rows, float* dlogits, float* probs) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= b) return;
  const float reg = reg_dev ? *reg_dev : reg_arg;  // device-resident: a captured graph follows the epoch (0.9 ** epoch)
  const float* z = logits + (size_t)r * g;
  float mx = -INFINITY;
  for (int j = 0; j < g; ++j) mx = fmaxf(mx, z[j]);
  float den = 0.f;
  for (int j = 0; j < g; ++j) den += __expf(z[j] - mx);
  const float lden = __logf(den);
  float slog = 0.f, plogp = 0.f;
  for (int j = 0; j < g; ++j) {
    const float lp = z[j] - mx - lden, p = __expf(lp);
    slog += lp;
    plogp = fmaf(p, lp, plogp);
  }
  loss_rows[r] = scale * (-tw * slog + reg * plogp);
  for (int j = 0; j < g; ++j) {
    const float lp = z[j] - mx - lden, p = __expf(lp);
    if (probs) probs[(size_t)r * g + j] = p;
    dlogits[(size_t)r * g + j] = scale * (tw * ((float)g * p - 1.f) + reg * p * (lp - plogp));
  }
}

// ---- PM-network 'ml' loss (train.py:626-639) ---------------------------------------------
// gen_abs (T,E,g,b,2); target = softmax_g( mean_E sum_{t,xy} log N(err; 0, sigma) ); loss_r = -sum target*log_softmax(logits)
// Lane = (pedestrian, generator): the T*E error terms of one generator per lane (the loads that dominate), the
// softmax pair per pedestrian by its first lane.  With `partial` / `ticket` the launch also finishes the job: the mean
// loss (out) and the mean generator probabilities (probs_out[g]) come from per-workgroup f64 sums that the last
// workgroup adds in index order (deterministic); otherwise only the per-row values are written.
#define PM_MAX_WG 64
__global__ __launch_bounds__(256) void pm_ml_kernel(int b, int T, int E, int g, int G2, const float* __restrict__ gen_abs,
                                                    const float* __restrict__ gt, const float* __restrict__ logits,
                                                    float sigma, float scale, float* loss_rows, float* dlogits,
                                                    float* probs, double* partial, unsigned* ticket, float* out,
                                                    float* probs_out, float probs_scale, const int* dims) {
  __shared__ float s_lp[256];
  __shared__ double red[17][256];  // [value][pedestrian slot of this workgroup]
  __shared__ int last;
  const float inv2s = 1.f / (2.f * sigma * sigma);
  const float cst = -__logf(sigma) - 0.91893853320467274178f;  // -log(sigma) - 0.5 log(2 pi)
  const int per = 256 / G2;  // pedestrians per workgroup pass
  const int slot = threadIdx.x / G2, gi = threadIdx.x % G2;
  const int nr = mg_real_rows(dims, b);  // padded batch: the phantom pedestrians [nr, b) get zero rows, the means are over nr
  scale *= mg_pad_corr(dims);
  double wsum[17];
#pragma unroll
  for (int q = 0; q < 17; ++q) wsum[q] = 0.0;
  for (int p0 = blockIdx.x * per; p0 < b; p0 += gridDim.x * per) {
    const int ped = p0 + slot;
    const bool ok = slot < per && ped < nr && gi < g;
    if (slot < per && ped >= nr && ped < b && gi < g) {
      dlogits[(size_t)ped * g + gi] = 0.f;
      if (probs) probs[(size_t)ped * g + gi] = 0.f;
      if (gi == 0) loss_rows[ped] = 0.f;
    }
    float lp = 0.f;
    if (ok) {
      float acc = 0.f;
      auto term = [&](int e, int t) {
        const float2 a = *reinterpret_cast<const float2*>(gen_abs + ((((size_t)t * E + e) * g + gi) * b + ped) * 2);
        const float2 y = *reinterpret_cast<const float2*>(gt + ((size_t)t * b + ped) * 2);
        const float dx = a.x - y.x, dy = a.y - y.y;
        return (-dx * dx * inv2s + cst) + (-dy * dy * inv2s + cst);
      };
      for (int e = 0; e < E; ++e) {  // four steps at a time (eight loads in flight), summed in the same order
        int t = 0;
        for (; t + 3 < T; t += 4) {
          const float v0 = term(e, t), v1 = term(e, t + 1), v2 = term(e, t + 2), v3 = term(e, t + 3);
          acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; t < T; ++t) acc += term(e, t);
      }
      lp = acc / (float)E;
    }
    __syncthreads();
    s_lp[threadIdx.x] = lp;
    __syncthreads();
    if (ok && gi == 0) {
      const float* lps = s_lp + slot * G2;
      float lg[16];
      float mxp = -INFINITY, mxl = -INFINITY;
      for (int q = 0; q < g; ++q) {
        lg[q] = logits[(size_t)ped * g + q];
        mxp = fmaxf(mxp, lps[q]);
        mxl = fmaxf(mxl, lg[q]);
      }
      float dp = 0.f, dl = 0.f;
      for (int q = 0; q < g; ++q) { dp += __expf(lps[q] - mxp); dl += __expf(lg[q] - mxl); }
      const float lsel = mxl + __logf(dl);
      float loss = 0.f;
      for (int q = 0; q < g; ++q) {
        const float tgt = __expf(lps[q] - mxp) / dp;
        const float sm = __expf(lg[q] - lsel);
        loss -= tgt * (lg[q] - lsel);
        dlogits[(size_t)ped * g + q] = scale * (sm - tgt);
        if (probs) probs[(size_t)ped * g + q] = sm;
        wsum[1 + q] += (double)sm;
      }
      loss_rows[ped] = scale * loss;
      wsum[0] += (double)(scale * loss);
    }
  }
  if (!partial) return;
  // workgroup sums in pedestrian-slot order, then the cross-workgroup sums by the last workgroup in index order
  if (gi == 0 && slot < per) {
#pragma unroll
    for (int q = 0; q < 17; ++q) red[q][slot] = wsum[q];
  }
  __syncthreads();
  if (threadIdx.x <= (unsigned)g) {
    double t = 0.0;
    for (int sl = 0; sl < per; ++sl) t += red[threadIdx.x][sl];
    partial[(size_t)blockIdx.x * 17 + threadIdx.x] = t;
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (threadIdx.x <= (unsigned)g) {
    const int q = threadIdx.x;
    double t = 0.0;
    for (unsigned w = 0; w < gridDim.x; ++w) t += ((volatile double*)partial)[(size_t)w * 17 + q];
    if (q == 0) *out = (float)t;
    else if (probs_out) probs_out[q - 1] = probs_scale * (float)(t / (double)(nr > 0 ? nr : 1));
  }
  if (threadIdx.x == 0) *ticket = 0u;
}

// ---- fixed-order sum: out[slot] (+)= alpha * sum_i x[i] -------------------------------------
__global__ __launch_bounds__(1024) void sum_kernel(const float* __restrict__ x, long n, float alpha, float* out,
                                                   int accumulate) {
  __shared__ double red[1024];
  double acc = 0.0;
  for (long i = threadIdx.x; i < n; i += 1024) acc += (double)x[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float v = alpha * (float)red[0];
    *out = accumulate ? (*out + v) : v;
  }
}

// column means of a (rows x g) matrix -> out[g]  (probs/Gen i probability metric, train.py:597-600)
__global__ __launch_bounds__(256) void colmean_kernel(const float* __restrict__ x, int rows, int g, float scale,
                                                      float* out) {
  __shared__ double red[256];
  const int c = blockIdx.x;
  double acc = 0.0;
  for (int r = threadIdx.x; r < rows; r += 256) acc += (double)x[(size_t)r * g + c];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = scale * (float)(red[0] / rows);
}

// ---- (all-reduce +) clip_grad_norm_ + AdamW over a flat parameter buffer with a segment table ---------------
// ONE launch per optimizer step.  A workgroup owns whole 1,024-element chunks of the flat buffer (chunk b, b + grid, ...)
// through every phase, so nothing but the norm couples the workgroups:
//   phase 0 (sharded training over peer-mapped arenas, `comm`): the gradient all-reduce itself -- the chunk protocol of
//            csrc/comm.hip (own chunk into slot r of every rank, flag, wait for the W flags, rank-ordered sum), chunk by chunk
//            by the workgroup that will update those elements; round 5 ran it as a launch of its own in front of this one.
//            The f64 tail that rides with the gradients (this rank's raw conv1 weight-gradient / BatchNorm-1 adjoint sums,
//            csrc/cnn2.hip) is one more chunk, and its finalize -- the global-batch dW1 / dgamma1 / dbeta1 from the
//            summed tail and the global Gram matrix -- happens right there instead of in a third launch: the tail has the
//            launch's last workgroup to itself (its exchange runs beside those of the gradient chunks), the finalized
//            block goes to a side buffer behind the counters, and the workgroups whose chunks hold slots of those three
//            parameters (zero until then, also in the sum) wait for its flag and add it -- the same numbers on every rank.
//            Without `comm` but with a tail (RCCL inside the graph: the exchange ran in front of this launch) the finalize
//            alone happens here.
//   phase 1: the squared norm of the elements a workgroup owns (f64), published; a grid barrier on a per-optimizer counter
//            that only ever grows (the grid is at most CA_MAX_BLOCKS workgroups of 256 threads: all resident); then every
//            workgroup adds the partials in index order (the same total everywhere, bit for bit, run after run).
//   phase 2: clip + AdamW on the same elements.
// Every wait is bounded (the collective's bound, or CA_WAIT_TICKS without one): a launch that gives up leaves NaN weights.
#define CA_MAX_BLOCKS 256
#define CA_CHUNK COMM_CHUNK
#define CA_WAIT_TICKS (20ll * COMM_TICKS_PER_S)
__device__ __forceinline__ void ca_store(double* p, double v) {
  __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ca_load(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}
struct GradTail {          // the f64 tail of a step's gradient exchange and its finalize (mggan_grad_tail_t)
  double* tail;            // [A (C x 36) | S1 (C) | S2 (C) | riders]; NULL: none
  long n2;
  const double* gram;      // 37 x 37, the GLOBAL batch's
  const float* W;          // conv1 weight (C,4,3,3), bias, BatchNorm-1 gamma, stat1 = [mean | invstd]
  const float* bias;
  const float* gamma;
  const float* stat;
  float* dW;               // slots INSIDE the flat gradient buffer: += the global-batch gradient
  float* dgamma;
  float* dbeta;
  int C;
};
// workspace: [CA_MAX_BLOCKS partial sums | arrival counter | finished counter | tail-finalize flag (64-bit words)], zero
// before the first launch; every launch leaves it ready for the next
#define CA_SEG_TABLE 512
#define CA_WS_DOUBLES 264        // partials + counters; the finalize block (38 C <= 608 floats) follows
__device__ __forceinline__ bool ca_wait_ge(unsigned long long* p, unsigned long long target, long long bound) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - t0 > bound) return false;
  }
  return true;
}
__global__ __launch_bounds__(256) void clip_adamw_kernel(float* param, float* grad, float* m, float* v, long n,
                                                         const int* __restrict__ elem_seg,
                                                         const unsigned char* __restrict__ active, int nseg,
                                                         int* seg_step, double* workspace, float max_norm, double lr_arg,
                                                         const double* __restrict__ lr_dev, double beta1, double beta2,
                                                         double eps_d, double wd, int zero_grad, float* norm_out,
                                                         const CommArgs* __restrict__ comm, GradTail gt) {
  __shared__ double red[256];
  __shared__ float s_sq2[CA_SEG_TABLE], s_lr1[CA_SEG_TABLE];
  __shared__ int lost_s, gave_up;
  const double lr = lr_dev ? *lr_dev : lr_arg;  // device-resident: a captured graph follows the schedule
  const int nchunks = (int)((n + CA_CHUNK - 1) / CA_CHUNK);
  // the bias corrections depend on the tensor (its step count), not on the element: two f64 pow() per TENSOR and workgroup
  // instead of per element
  const bool table = nseg <= CA_SEG_TABLE;
  if (table)
    for (int sg = threadIdx.x; sg < nseg; sg += 256) {
      const int t = seg_step[sg] + 1;
      const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
      s_sq2[sg] = (float)sqrt(bc2);
      s_lr1[sg] = (float)(lr / bc1);
    }
  if (threadIdx.x == 0) gave_up = 0;
  unsigned long long* counter = (unsigned long long*)(workspace + CA_MAX_BLOCKS);
  unsigned long long* fin_flag = counter + 2;
  const long long bound = comm ? comm->timeout_ticks : CA_WAIT_TICKS;
  // ---- phase 0: the exchange (and the tail's finalize) ----
  CommHeader* hdr = comm ? (CommHeader*)comm->arena[comm->rank] : nullptr;
  const unsigned seq = comm ? __hip_atomic_load(&hdr->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u : 0u;
  float* fin_out = (float*)(workspace + CA_WS_DOUBLES);  // 38 C floats behind the counters
  // the tail has a workgroup of its own (the last one): nobody exchanges two things in a row
  const int nwork = gridDim.x - (gt.tail ? 1 : 0);
  if (gt.tail && (int)blockIdx.x == nwork) {
    if (comm) {
      const size_t off = ((size_t)n * sizeof(float) + 255) / 256 * 256;
      comm_chunk<double>(*comm, gt.tail, 0, gt.n2, off, nchunks, seq, hdr, &lost_s);
      __syncthreads();
    }
    {
      // dW1 = (gamma/sigma) (A - (S1/n) B - (S2/n) Chat),  dgamma1 = S2,  dbeta1 = S1  (csrc/cnn2.hip:
      // conv1_tail_finalize_kernel, one 64-lane group per output channel) -> fin_out = [dW1 (C x 36) | dbeta1 (C) | dgamma1 (C)]
      const int C = gt.C, t = threadIdx.x & 63;
      const double nn = gt.gram[36 * 37 + 36];
      for (int c = threadIdx.x >> 6; c < C; c += 4) {
        const double S1 = gt.tail[C * 36 + c], S2 = gt.tail[C * 36 + C + c];
        if (t == 36) fin_out[C * 36 + c] = (float)S1;
        if (t == 37) fin_out[C * 37 + c] = (float)S2;
        if (t < 36) {
          const double mean = (double)gt.stat[c], inv = (double)gt.stat[C + c], cs = (double)gt.gamma[c] * inv;
          const double Bt = gt.gram[36 * 37 + t];
          double wp = 0.0;
          for (int s = 0; s < 36; ++s) wp += (double)gt.W[c * 36 + s] * gt.gram[s * 37 + t];
          const double chat = (wp + ((double)gt.bias[c] - mean) * Bt) * inv;
          fin_out[c * 36 + t] = (float)(cs * (gt.tail[c * 36 + t] - (S1 / nn) * Bt - (S2 / nn) * chat));
        }
      }
      __threadfence();
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(fin_flag, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  double acc = 0.0;
  const long dW0 = gt.tail ? (long)(gt.dW - grad) : 0, dG0 = gt.tail ? (long)(gt.dgamma - grad) : 0,
             dB0 = gt.tail ? (long)(gt.dbeta - grad) : 0;
  for (int k = blockIdx.x; k < nchunks && (int)blockIdx.x < nwork; k += nwork) {
    const long e0 = (long)k * CA_CHUNK, cnt = min((long)CA_CHUNK, n - e0), e1 = e0 + cnt;
    if (comm) {
      comm_chunk<float>(*comm, grad, e0, cnt, 0, k, seq, hdr, &lost_s);
      __syncthreads();
    }
    if (gt.tail && ((dW0 < e1 && dW0 + gt.C * 36 > e0) || (dG0 < e1 && dG0 + gt.C > e0) || (dB0 < e1 && dB0 + gt.C > e0))) {
      // this chunk holds slots of dW1 / dgamma1 / dbeta1 (zero so far, also in the sum): the tail workgroup's finalize
      // goes in now -- the same numbers on every rank
      if (threadIdx.x == 0 && !ca_wait_ge(fin_flag, 1ull, bound)) gave_up = 1;
      __syncthreads();
      __threadfence();
      for (long i = e0 + threadIdx.x; i < e1; i += 256) {
        float add = 0.f;
        if (i >= dW0 && i < dW0 + gt.C * 36) add = __builtin_nontemporal_load(fin_out + (i - dW0));
        else if (i >= dB0 && i < dB0 + gt.C) add = __builtin_nontemporal_load(fin_out + gt.C * 36 + (i - dB0));
        else if (i >= dG0 && i < dG0 + gt.C) add = __builtin_nontemporal_load(fin_out + gt.C * 37 + (i - dG0));
        if (add != 0.f) grad[i] += add;
      }
    }
    // ---- phase 1: this chunk's share of the squared norm (f64) ----
    for (long i = e0 + threadIdx.x; i < e1; i += 256) {
      const int sg = elem_seg[i];
      if (sg >= 0 && active[sg]) acc += (double)grad[i] * (double)grad[i];
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    ca_store(workspace + blockIdx.x, red[0]);
    __threadfence();
    const unsigned long long mine = atomicAdd(counter, 1ull);
    // this launch's arrivals are (mine - mine % grid) .. + grid - 1: launches of one optimizer are ordered by their stream
    const unsigned long long target = mine - mine % gridDim.x + gridDim.x;
    if (!ca_wait_ge(counter, target, bound)) gave_up = 1;
    __threadfence();
  }
  __syncthreads();
  // ---- every workgroup: the same fixed-order sum of all partials ----
  red[threadIdx.x] = threadIdx.x < gridDim.x ? ca_load(workspace + threadIdx.x) : 0.0;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float total = (float)sqrt(red[0]);
  float coef = 1.f;
  if (max_norm > 0.f) coef = fminf(max_norm / (total + 1e-6f), 1.f);
  if (gave_up) coef = __builtin_nanf("");  // a wait ran out: visible in the weights, never a half-exchanged update
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = total;
  // ---- phase 2: clip + AdamW on the same elements; the Adam step counter of a touched tensor is its old value + 1 (the
  // counters themselves are advanced by the LAST workgroup to finish its elements: nobody reads them after that)
  for (int k = blockIdx.x; k < nchunks && (int)blockIdx.x < nwork; k += nwork) {
    const long e0 = (long)k * CA_CHUNK, e1 = min(n, e0 + CA_CHUNK);
    for (long i = e0 + threadIdx.x; i < e1; i += 256) {
      const int sg = elem_seg[i];
      if (sg < 0 || !active[sg]) continue;
      const float g = grad[i] * coef;
      grad[i] = zero_grad ? 0.f : g;  // clip_grad_norm_ scales .grad in place; or leave it zeroed for the next step
      // torch.optim.AdamW single-tensor path: scalar factors in double, tensor math in f32
      float p = param[i] * (float)(1.0 - lr * wd);
      const float mi = m[i] + (g - m[i]) * (float)(1.0 - beta1);          // exp_avg.lerp_(grad, 1-beta1)
      const float vi = v[i] * (float)beta2 + g * g * (float)(1.0 - beta2);  // mul_(beta2).addcmul_
      m[i] = mi;
      v[i] = vi;
      float sq2, lr1;
      if (table) {
        sq2 = s_sq2[sg];
        lr1 = s_lr1[sg];
      } else {
        const int t = seg_step[sg] + 1;
        const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
        sq2 = (float)sqrt(bc2);
        lr1 = (float)(lr / bc1);
      }
      const float denom = sqrtf(vi) / sq2 + (float)eps_d;
      p -= lr1 * (mi / denom);
      param[i] = p;
    }
  }
  __shared__ int last;
  __syncthreads();
  unsigned long long* done = counter + 1;
  if (threadIdx.x == 0) {
    const unsigned long long d = atomicAdd(done, 1ull);
    last = d == gridDim.x - 1;
    if (last) {
      __hip_atomic_store(done, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(fin_flag, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
      if (comm) __hip_atomic_store(&hdr->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the collective is closed
    }
  }
  __syncthreads();
  if (last)
    for (int i = threadIdx.x; i < nseg; i += 256)
      if (active[i]) seg_step[i] += 1;
}

__global__ void adam_inc_kernel(int nseg, const unsigned char* __restrict__ active, int* seg_step) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nseg && active[i]) seg_step[i] += 1;
}

// counts of generator ids (int32 atomics: exact, order independent) and their reciprocals
__global__ __launch_bounds__(256) void count_kernel(const int* __restrict__ idx, int n, int g, int* counts,
                                                    const int* dims, int bmod) {
  __shared__ int hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const int bm = bmod > 0 ? bmod : 1, nr = dims ? mg_real_rows(dims, bm) : bm;  // row i <-> pedestrian i % bm
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    if (!dims || i % bm < nr) atomicAdd(&hist[idx[i]], 1);
  __syncthreads();
  if (threadIdx.x < g && hist[threadIdx.x]) atomicAdd(&counts[threadIdx.x], hist[threadIdx.x]);
}
__global__ void inv_count_kernel(const int* __restrict__ counts, int g, float* inv) {
  const int i = threadIdx.x;
  if (i < g) inv[i] = counts[i] > 0 ? 1.f / (float)counts[i] : 0.f;
}

// ---- device-side row bookkeeping for the selected rollouts (replaces get_selection_indices +
// the index gather of standard.py:190-214, utils.py:234-248) ---------------------------------
// idx (b, K) generator ids.  Output rows are stably sorted by generator: row r <-> output position
// pos = k*b + ped; row_slot = number of earlier samples of the same pedestrian with the same generator.
#define BR_MAXG 16
#define BR_BLOCK 1024  // positions per workgroup in the counting / scatter passes

// pass 1: one lane per pedestrian walks its K samples: generator id per output position, occurrence offset
// (noise slot) parked in inv[pos]
__global__ __launch_bounds__(256) void bucket_slots_kernel(const long long* __restrict__ idx, int b, int K,
                                                           int* row_gen_pos, int* inv) {
  const int ped = blockIdx.x * 256 + threadIdx.x;
  if (ped >= b) return;
  int seen[BR_MAXG];
#pragma unroll
  for (int q = 0; q < BR_MAXG; ++q) seen[q] = 0;
#pragma unroll 4
  for (int k = 0; k < K; ++k) {
    const int gi = (int)idx[(size_t)ped * K + k];
    int slot = 0;
#pragma unroll
    for (int q = 0; q < BR_MAXG; ++q) {
      if (q == gi) { slot = seen[q]; seen[q] += 1; }
    }
    row_gen_pos[k * b + ped] = gi;
    inv[k * b + ped] = slot;
  }
}

// pass 2: per block of 1024 consecutive positions, number of rows of every generator
__global__ __launch_bounds__(BR_BLOCK) void bucket_count_kernel(const int* __restrict__ row_gen_pos, int R, int g,
                                                                int* blk_cnt) {
  __shared__ int hist[BR_MAXG];
  if (threadIdx.x < BR_MAXG) hist[threadIdx.x] = 0;
  __syncthreads();
  const int pos = blockIdx.x * BR_BLOCK + threadIdx.x;
  if (pos < R) atomicAdd(&hist[row_gen_pos[pos]], 1);  // integer LDS atomics: exact, order independent
  __syncthreads();
  if (threadIdx.x < g) blk_cnt[blockIdx.x * BR_MAXG + threadIdx.x] = hist[threadIdx.x];
}

// pass 3: exclusive scan over blocks per generator + segment offsets (one small workgroup)
__global__ __launch_bounds__(64) void bucket_scan_kernel(int nblk, int g, int* blk_cnt, int* seg) {
  __shared__ int tot[BR_MAXG];
  const int q = threadIdx.x;
  if (q < g) {
    int run = 0, i = 0;
    for (; i + 7 < nblk; i += 8) {  // eight counts in flight, then their running offsets (160 blocks at 163,840 rows)
      int c[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) c[e] = blk_cnt[(i + e) * BR_MAXG + q];
#pragma unroll
      for (int e = 0; e < 8; ++e) { blk_cnt[(i + e) * BR_MAXG + q] = run; run += c[e]; }
    }
    for (; i < nblk; ++i) { const int c = blk_cnt[i * BR_MAXG + q]; blk_cnt[i * BR_MAXG + q] = run; run += c; }
    tot[q] = run;
  }
  __syncthreads();
  if (q == 0) {
    int run = 0;
    for (int i = 0; i < g; ++i) { seg[i] = run; run += tot[i]; }
    seg[g] = run;
  }
}

// pass 4: stable scatter.  Rank of a position among the same-generator positions of its block: wave ballot +
// prefix over the block's waves.
__global__ __launch_bounds__(BR_BLOCK) void bucket_scatter_kernel(const int* __restrict__ row_gen_pos, int R, int b, int g,
                                                                  int* blk_cnt, const int* __restrict__ seg, int* row_gen,
                                                                  int* row_ped, int* row_slot, int* row_pos, int* inv,
                                                                  int self_reset) {
  __shared__ int wcnt[BR_BLOCK / 64][BR_MAXG];
  const int pos = blockIdx.x * BR_BLOCK + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool ok = pos < R;
  const int gi = ok ? row_gen_pos[pos] : -1;
  int rank = 0;
  for (int q = 0; q < g; ++q) {
    const unsigned long long m = __ballot(gi == q);
    if (gi == q) rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wcnt[wv][q] = __popcll(m);
  }
  __syncthreads();
  if (ok) {
    int before = 0;
    for (int i = 0; i < wv; ++i) before += wcnt[i][gi];
    const int r = seg[gi] + blk_cnt[blockIdx.x * BR_MAXG + gi] + before + rank;
    const int slot = inv[pos];
    row_gen[r] = gi; row_ped[r] = pos % b; row_slot[r] = slot; row_pos[r] = pos; inv[pos] = r;
  }
  if (self_reset) {
    // a workgroup is the only reader of its row of block offsets: it leaves the row zeroed for the next call's counting
    // pass (a caller-owned, persistent buffer: no memset node in front of every call)
    __syncthreads();
    if (threadIdx.x < BR_MAXG) blk_cnt[blockIdx.x * BR_MAXG + threadIdx.x] = 0;
  }
}

// The four passes in ONE workgroup for small row counts (the single-sample rollouts of the discriminator step:
// three dependent launches less on its critical chain).  Same stable order as the multi-pass version.
__global__ __launch_bounds__(BR_BLOCK) void bucket_small_kernel(const long long* __restrict__ idx, int b, int K, int g,
                                                                int* row_gen_pos, int* inv, int* seg, int* row_gen,
                                                                int* row_ped, int* row_slot, int* row_pos) {
  __shared__ int hist[BR_MAXG], base[BR_MAXG], wcnt[BR_BLOCK / 64][BR_MAXG];
  const int R = b * K, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t < BR_MAXG) hist[t] = 0;
  __syncthreads();
  for (int ped = t; ped < b; ped += BR_BLOCK) {
    int seen[BR_MAXG];
#pragma unroll
    for (int q = 0; q < BR_MAXG; ++q) seen[q] = 0;
    for (int k = 0; k < K; ++k) {
      const int gi = (int)idx[(size_t)ped * K + k];
      int slot = 0;
#pragma unroll
      for (int q = 0; q < BR_MAXG; ++q) {
        if (q == gi) { slot = seen[q]; seen[q] += 1; }
      }
      row_gen_pos[k * b + ped] = gi;
      inv[k * b + ped] = slot;
      atomicAdd(&hist[gi], 1);
    }
  }
  __threadfence();
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int i = 0; i < g; ++i) { seg[i] = run; base[i] = run; run += hist[i]; }
    seg[g] = run;
  }
  __syncthreads();
  for (int c0 = 0; c0 < R; c0 += BR_BLOCK) {
    const int pos = c0 + t;
    const bool ok = pos < R;
    const int gi = ok ? row_gen_pos[pos] : -1;
    int rank = 0;
    for (int q = 0; q < g; ++q) {
      const unsigned long long m = __ballot(gi == q);
      if (gi == q) rank = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wcnt[wv][q] = __popcll(m);
    }
    __syncthreads();
    if (ok) {
      int before = 0;
      for (int i = 0; i < wv; ++i) before += wcnt[i][gi];
      const int r = base[gi] + before + rank;
      const int slot = inv[pos];
      row_gen[r] = gi; row_ped[r] = pos % b; row_slot[r] = slot; row_pos[r] = pos; inv[pos] = r;
    }
    __syncthreads();
    if (t < g) {
      int tot = 0;
      for (int i = 0; i < BR_BLOCK / 64; ++i) tot += wcnt[i][t];
      base[t] += tot;
    }
    __syncthreads();
  }
}

// Categorical(logits).sample((K,)).T by inverse CDF from caller-provided uniforms u (b,K) in [0,1)
// (standard.py:217-225 on the device, no host round trip).  sb_build / sb_pick are THE arithmetic of a pick: every
// kernel below that samples goes through them, so that the fused and the stand-alone launches choose alike.
__device__ __forceinline__ void sb_build(const float* __restrict__ l, int g, float (&cdf)[BR_MAXG], float& run) {
  float v[BR_MAXG], mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < BR_MAXG; ++c) {
    v[c] = c < g ? l[c] : -INFINITY;
    mx = fmaxf(mx, v[c]);
  }
  run = 0.f;
#pragma unroll
  for (int c = 0; c < BR_MAXG; ++c) {
    if (c < g) run += __expf(v[c] - mx);
    cdf[c] = run;
  }
}
__device__ __forceinline__ int sb_pick(const float (&cdf)[BR_MAXG], int g, float x) {
  int pick = g - 1;
#pragma unroll
  for (int c = BR_MAXG - 2; c >= 0; --c)
    if (c < g - 1 && x < cdf[c]) pick = c;
  return pick;
}

// One lane per (pedestrian, sample).  Every lane rebuilds its pedestrian's g-entry CDF (the row is an L1 hit) - a lane
// per pedestrian walking its K samples was K dependent strided round trips (61 us at 8,192 x 20).
__global__ void sample_categorical_kernel(int b, int K, int g, const float* __restrict__ logits,
                                          const float* __restrict__ u, long long* idx) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)b * K) return;
  const int ped = (int)(i / K);
  float cdf[BR_MAXG], run;
  sb_build(logits + (size_t)ped * g, g, cdf, run);
  idx[i] = sb_pick(cdf, g, u[i] * run);
}

// How often every generator WILL be picked by a sampling launch (mggan_sample_categorical / mggan_sample_bucket_rows) on
// these logits and uniforms -- the same CDF, the same comparison -- without bucketing anything: sharded training computes
// the generator step's counts ahead of time, so that they travel as f64 riders of an earlier exchange (the discriminator
// step's gradient all-reduce) instead of being a collective of their own.  One workgroup, integer LDS
// atomics (exact, order independent); out[0 .. BR_MAXG) = counts as doubles (zero beyond g).
__global__ __launch_bounds__(256) void sample_counts_kernel(int b, int K, int g, const float* __restrict__ logits,
                                                            const float* __restrict__ u, int* scratch /*[BR_MAXG + 1]*/,
                                                            double* out) {
  __shared__ int hist[BR_MAXG];
  __shared__ int last;
  if (threadIdx.x < BR_MAXG) hist[threadIdx.x] = 0;
  __syncthreads();
  // a lane is ONE draw (pedestrian i / K, sample i % K): coalesced reads of the uniforms and b * K / 256 workgroups -- as a
  // lane per pedestrian walking its K draws (a stride-K read each) this was 21-37 us on the discriminator step's chain at
  // 1,280 pedestrians x 20 samples; the CDF is rebuilt per draw (g exponentials) with the same calls: the same picks
  // (a handful of workgroups walking the draws grid-strided: every workgroup ends with g atomics on the same g words, and
  //  a hundred workgroups queued 20 us of them)
  // ... and the picks of a wave are counted with ballots (64 lanes adding to the same g LDS words one after the other were
  //  most of what was left): lane c < g keeps the wave's count of generator c
  int mine = 0;
  const long total = (long)b * K;
  for (long i0 = (long)blockIdx.x * 256; i0 < total; i0 += (long)gridDim.x * 256) {  // (uniform trip count per workgroup)
    const long i = i0 + threadIdx.x;
    int pick = -1;
    if (i < total) {
      const int ped = (int)(i / K);
      float cdf[BR_MAXG], run;
      sb_build(logits + (size_t)ped * g, g, cdf, run);
      pick = sb_pick(cdf, g, u[i] * run);
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < BR_MAXG; ++c)
      if (c < g) {
        const int n = __popcll(__ballot(pick == c));
        if (lane == c) mine += n;
      }
  }
  if ((threadIdx.x & 63) < g && mine) atomicAdd(&hist[threadIdx.x & 63], mine);
  __syncthreads();
  if (threadIdx.x < BR_MAXG && hist[threadIdx.x]) atomicAdd(&scratch[threadIdx.x], hist[threadIdx.x]);
  // the last workgroup hands the totals over as doubles and re-arms the scratch words (integer sums: order independent)
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(&scratch[BR_MAXG], 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  if (threadIdx.x < BR_MAXG) {
    out[threadIdx.x] = (double)__hip_atomic_load(&scratch[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&scratch[threadIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) __hip_atomic_store(&scratch[BR_MAXG], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void inv_count_f64_kernel(const double* __restrict__ counts, int g, float* inv) {
  const int i = threadIdx.x;
  if (i < g) inv[i] = counts[i] > 0.5 ? 1.f / (float)counts[i] : 0.f;
}

// ---- sampling + bucketing in one launch for small row counts ------------------------------------------------------
// (Beyond a few thousand rows the fusion does not pay: a two-launch form - one workgroup per (sample, 1024-pedestrian
// chunk) re-drawing the earlier samples of its lanes for the occurrence offsets, block counts scanned by the last
// workgroup to finish - took 76 us at 8,192 x 20 against 25 us for the five separate launches: the re-draws and the
// serial, L2-latency-bound scan behind a ticket cost more than the launches they replace.)
// Small row counts (the single-sample rollouts of the discriminator step, small batches): everything in ONE workgroup,
// generator ids and occurrence offsets staged in LDS between the passes.  Same stable order as the multi-pass version.
#define SB_SMALL 2048
__global__ __launch_bounds__(BR_BLOCK) void sample_bucket_small_kernel(int b, int K, int g, const float* __restrict__ logits,
                                                                       const float* __restrict__ u, long long* idx,
                                                                       int* row_gen_pos, int* inv, int* seg, int* row_gen,
                                                                       int* row_ped, int* row_slot, int* row_pos) {
  __shared__ unsigned char s_gen[SB_SMALL];
  __shared__ unsigned short s_slot[SB_SMALL];
  __shared__ int hist[BR_MAXG], base[BR_MAXG], wcnt[BR_BLOCK / 64][BR_MAXG];
  const int R = b * K, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t < BR_MAXG) hist[t] = 0;
  __syncthreads();
  for (int ped = t; ped < b; ped += BR_BLOCK) {
    float cdf[BR_MAXG], run;
    sb_build(logits + (size_t)ped * g, g, cdf, run);
    int seen[BR_MAXG];
#pragma unroll
    for (int q = 0; q < BR_MAXG; ++q) seen[q] = 0;
    for (int k = 0; k < K; ++k) {
      const int gi = sb_pick(cdf, g, u[(size_t)ped * K + k] * run);
      int slot = 0;
#pragma unroll
      for (int q = 0; q < BR_MAXG; ++q) {
        if (q == gi) { slot = seen[q]; seen[q] += 1; }
      }
      idx[(size_t)ped * K + k] = gi;
      row_gen_pos[k * b + ped] = gi;
      s_gen[k * b + ped] = (unsigned char)gi;
      s_slot[k * b + ped] = (unsigned short)slot;
    }
#pragma unroll
    for (int q = 0; q < BR_MAXG; ++q)
      if (seen[q]) atomicAdd(&hist[q], seen[q]);
  }
  __syncthreads();
  if (t == 0) {
    int runc = 0;
    for (int i = 0; i < g; ++i) { seg[i] = runc; base[i] = runc; runc += hist[i]; }
    seg[g] = runc;
  }
  __syncthreads();
  for (int c0 = 0; c0 < R; c0 += BR_BLOCK) {
    const int pos = c0 + t;
    const bool ok = pos < R;
    const int gi = ok ? (int)s_gen[pos] : -1;
    int rank = 0;
    for (int q = 0; q < g; ++q) {
      const unsigned long long m = __ballot(gi == q);
      if (gi == q) rank = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wcnt[wv][q] = __popcll(m);
    }
    __syncthreads();
    if (ok) {
      int before = 0;
      for (int i = 0; i < wv; ++i) before += wcnt[i][gi];
      const int r = base[gi] + before + rank;
      row_gen[r] = gi; row_ped[r] = pos % b; row_slot[r] = (int)s_slot[pos]; row_pos[r] = pos; inv[pos] = r;
    }
    __syncthreads();
    if (t < g) {
      int tot = 0;
      for (int i = 0; i < BR_BLOCK / 64; ++i) tot += wcnt[i][t];
      base[t] += tot;
    }
    __syncthreads();
  }
}

extern "C" {

int mggan_sample_categorical(int b, int K, int g, const float* logits, const float* u, long long* idx,
                             hipStream_t stream) {
  if (b == 0) return MGGAN_OK;
  MG_CHECK_ARG(logits && u && idx && g >= 1 && g <= BR_MAXG, "sample_categorical: bad arguments (num_gens <= 16)");
  MG_LAUNCH(sample_categorical_kernel, dim3(cdiv((long)b * K, 256)), dim3(256), 0, stream, b, K, g, logits, u, idx);
  MG_LAUNCH_CHECK("sample_categorical");
  return MGGAN_OK;
}

int mggan_bucket_rows(const long long* idx, int b, int K, int g, int* row_gen, int* row_ped, int* row_slot, int* row_pos,
                      int* inv, int* seg, int* row_gen_pos, int* blk_cnt, hipStream_t stream);

/* mggan_sample_categorical + mggan_bucket_rows; ONE launch up to 2,048 rows: the same picks, the same tables */
// Above SB_SMALL rows: picks, occurrence slots, per-block counts AND the scan in ONE launch (it was four: sample_categorical,
// bucket_slots, bucket_count, bucket_scan -- 34 us in a row in front of the generator step's rollout at configs[1]).  A lane is
// a pedestrian: its CDF once, its K picks and slots in sample order; the counts of the picks' position blocks are gathered in
// an LDS image of blk_cnt (integer LDS atomics) and its non-zero cells added to the global table (zeroed by the launcher) at the
// end -- exact and order independent.  (One global atomic per pick: 25,600 of them on a hundred addresses took 25 us.)  The workgroup that takes the last ticket
// turns the counts into exclusive offsets per generator and the segment table (the same arithmetic as bucket_scan_kernel)
// and re-arms the ticket.  bucket_scatter_kernel follows.
__global__ __launch_bounds__(256) void sample_slots_scan_kernel(int b, int K, int g, const float* __restrict__ logits,
                                                                const float* __restrict__ u, long long* idx, int* row_gen_pos,
                                                                int* inv, int* blk_cnt, int nblk, int* seg, unsigned* ticket) {
  __shared__ int last;
  __shared__ int tot[BR_MAXG];
  extern __shared__ int hist[];  // [nblk][BR_MAXG]
  for (int i = threadIdx.x; i < nblk * BR_MAXG; i += 256) hist[i] = 0;
  __syncthreads();
  const int ped = blockIdx.x * 256 + threadIdx.x;
  if (ped < b) {
    float cdf[BR_MAXG], run;
    sb_build(logits + (size_t)ped * g, g, cdf, run);
    int seen[BR_MAXG];
#pragma unroll
    for (int q = 0; q < BR_MAXG; ++q) seen[q] = 0;
    for (int k0 = 0; k0 < K; k0 += 4) {  // four draws in flight
      float uu[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) uu[e] = k0 + e < K ? u[(size_t)ped * K + k0 + e] : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = k0 + e;
        if (k >= K) break;
        const int gi = sb_pick(cdf, g, uu[e] * run);
        int slot = 0;
#pragma unroll
        for (int q = 0; q < BR_MAXG; ++q)
          if (q == gi) { slot = seen[q]; seen[q] += 1; }
        idx[(size_t)ped * K + k] = gi;
        const int pos = k * b + ped;
        row_gen_pos[pos] = gi;
        inv[pos] = slot;
        atomicAdd(&hist[(pos / BR_BLOCK) * BR_MAXG + gi], 1);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nblk * BR_MAXG; i += 256)
    if (hist[i]) atomicAdd(&blk_cnt[i], hist[i]);
  __syncthreads();  // (this workgroup's atomics are through; the ticket below is taken behind them)
  if (threadIdx.x == 0) {
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  const int q = threadIdx.x;
  if (q < g) {
    int runc = 0;
    for (int i = 0; i < nblk; ++i) {
      int* cell = &blk_cnt[i * BR_MAXG + q];
      const int c = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(cell, runc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      runc += c;
    }
    tot[q] = runc;
  }
  __syncthreads();
  if (q == 0) {
    int runc = 0;
    for (int i = 0; i < g; ++i) { seg[i] = runc; runc += tot[i]; }
    seg[g] = runc;
    *ticket = 0u;
  }
}

int mggan_sample_bucket_rows(int b, int K, int g, const float* logits, const float* u, long long* idx, int* row_gen,
                             int* row_ped, int* row_slot, int* row_pos, int* inv, int* seg, int* row_gen_pos, int* blk_cnt,
                             int blk_self_reset, unsigned int* ticket, hipStream_t stream) {
  MG_CHECK_ARG(g >= 1 && g <= BR_MAXG && K >= 0 && K < 65536, "sample_bucket_rows: num_gens %d not in 1..%d (samples %d)", g, BR_MAXG, K);
  const long Rl = (long)b * K;
  if (Rl == 0) return MGGAN_OK;
  MG_CHECK_ARG(logits && u && idx && row_gen && row_ped && row_slot && row_pos && inv && seg && row_gen_pos && blk_cnt && ticket,
               "sample_bucket_rows: null pointer");
  const int R = (int)Rl;
  if (R <= SB_SMALL) {
    MG_LAUNCH(sample_bucket_small_kernel, dim3(1), dim3(BR_BLOCK), 0, stream, b, K, g, logits, u, idx, row_gen_pos,
                       inv, seg, row_gen, row_ped, row_slot, row_pos);
    MG_LAUNCH_CHECK("sample_bucket_rows");
    return MGGAN_OK;
  }
  static int merged = -1;  // MGGAN_SB_MERGED=0: the five-launch path (A/B measurements)
  if (merged < 0) { const char* e = getenv("MGGAN_SB_MERGED"); merged = !(e && e[0] == '0'); }
  // (the multi-launch forms leave their block counts behind: a self-resetting buffer is cleared behind them)
  auto multi = [&]() -> int {
    if (int rc = mggan_sample_categorical(b, K, g, logits, u, idx, stream)) return rc;
    if (int rc = mggan_bucket_rows(idx, b, K, g, row_gen, row_ped, row_slot, row_pos, inv, seg, row_gen_pos, blk_cnt, stream)) return rc;
    if (blk_self_reset)
      MG_CHECK_HIP(hipMemsetAsync(blk_cnt, 0, sizeof(int) * (size_t)cdiv(R, BR_BLOCK) * BR_MAXG, stream), "sample_bucket_rows: memset");
    return MGGAN_OK;
  };
  if (!merged) return multi();
  const int nblk = cdiv(R, BR_BLOCK);
  // the scan kernel keeps the block counts in dynamic LDS (nblk * BR_MAXG ints) beside a few static words: above what fits
  // into 64 KB the five-launch path takes over (decided BEFORE anything is written)
  if ((size_t)nblk * BR_MAXG * sizeof(int) > 64 * 1024 - 256) return multi();
  if (!blk_self_reset)
    MG_CHECK_HIP(hipMemsetAsync(blk_cnt, 0, sizeof(int) * (size_t)nblk * BR_MAXG, stream), "sample_bucket_rows: memset");
  MG_LAUNCH(sample_slots_scan_kernel, dim3(cdiv(b, 256)), dim3(256), sizeof(int) * (size_t)nblk * BR_MAXG, stream, b, K, g,
                     logits, u, idx, row_gen_pos, inv, blk_cnt, nblk, seg, ticket);
  MG_LAUNCH(bucket_scatter_kernel, dim3(nblk), dim3(BR_BLOCK), 0, stream, row_gen_pos, R, b, g, blk_cnt, seg, row_gen,
                     row_ped, row_slot, row_pos, inv, blk_self_reset);
  MG_LAUNCH_CHECK("sample_bucket_rows");
  return MGGAN_OK;
}

int mggan_bucket_rows(const long long* idx, int b, int K, int g, int* row_gen, int* row_ped, int* row_slot,
                      int* row_pos, int* inv, int* seg, int* row_gen_pos, int* blk_cnt, hipStream_t stream) {
  MG_CHECK_ARG(idx && row_gen && row_ped && row_slot && row_pos && inv && seg && row_gen_pos && blk_cnt,
               "bucket_rows: null pointer");
  MG_CHECK_ARG(g >= 1 && g <= BR_MAXG, "bucket_rows: num_gens %d exceeds %d", g, BR_MAXG);
  const int R = b * K, nblk = cdiv(R, BR_BLOCK);
  if (R == 0) return MGGAN_OK;
  if (R <= 4 * BR_BLOCK) {
    MG_LAUNCH(bucket_small_kernel, dim3(1), dim3(BR_BLOCK), 0, stream, idx, b, K, g, row_gen_pos, inv, seg,
                       row_gen, row_ped, row_slot, row_pos);
    MG_LAUNCH_CHECK("bucket_rows");
    return MGGAN_OK;
  }
  MG_LAUNCH(bucket_slots_kernel, dim3(cdiv(b, 256)), dim3(256), 0, stream, idx, b, K, row_gen_pos, inv);
  MG_LAUNCH(bucket_count_kernel, dim3(nblk), dim3(BR_BLOCK), 0, stream, row_gen_pos, R, g, blk_cnt);
  MG_LAUNCH(bucket_scan_kernel, dim3(1), dim3(64), 0, stream, nblk, g, blk_cnt, seg);
  MG_LAUNCH(bucket_scatter_kernel, dim3(nblk), dim3(BR_BLOCK), 0, stream, row_gen_pos, R, b, g, blk_cnt, seg,
                     row_gen, row_ped, row_slot, row_pos, inv, 0);
  MG_LAUNCH_CHECK("bucket_rows");
  return MGGAN_OK;
}

int mggan_bce_rows(int rows, int kind, const float* p, float label, const float* label_u, float label_lo,
                   float label_hi, float scale, const int* row_gen, const float* inv_count, float* loss_rows, float* dp,
                   hipStream_t stream) {
  if (rows == 0) return MGGAN_OK;
  MG_CHECK_ARG(p && loss_rows && ((row_gen == nullptr) == (inv_count == nullptr)) && (kind == 0 || kind == 1),
               "bce_rows: bad arguments");
  MG_LAUNCH(bce_rows_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, stream, rows, kind, p, label, label_u, label_lo,
                     label_hi, scale, row_gen, inv_count, loss_rows, dp);
  MG_LAUNCH_CHECK("bce_rows");
  return MGGAN_OK;
}

int mggan_gan_losses(const void* args, hipStream_t stream) {
  MG_CHECK_ARG(args, "gan_losses: null argument block");
  const GanLossArgs a = *static_cast<const GanLossArgs*>(args);
  MG_CHECK_ARG(a.nA >= 0 && a.nB >= 0 && a.nC >= 0 && (a.kind == 0 || a.kind == 1), "gan_losses: bad sizes");
  MG_CHECK_ARG(a.total && a.partial && a.ticket && (a.nA + a.nB == 0 || (a.p && a.dp)), "gan_losses: null pointer");
  MG_CHECK_ARG(a.nC == 0 || (a.logits && a.target && a.dlogits && a.g > 0 && a.g <= 256 && a.ld >= a.g),
               "gan_losses: bad classifier term");
  MG_CHECK_ARG(!a.row_gen || ((a.seg || a.inv_count) && a.g > 0 && a.g <= 256), "gan_losses: row weights need counts");
  MG_CHECK_ARG(!a.weighted_c || a.row_gen, "gan_losses: weighted classifier term needs counts");
  int rows = a.nA > a.nB ? a.nA : a.nB;
  if (a.nC > rows) rows = a.nC;
  // one row of every term per thread up to 262,144 rows; up to 4,096 rows ONE workgroup takes four rows per thread: the
  // ticket round trip of a second workgroup (partial sums out, atomic, partial sums back) costs more than the rows
  int wgs = rows <= 4096 ? 1 : cdiv(rows, 1024);
  wgs = wgs < 1 ? 1 : (wgs > GAN_LOSS_MAX_WG ? GAN_LOSS_MAX_WG : wgs);
  MG_LAUNCH(gan_losses_kernel, dim3(wgs), dim3(1024), 0, stream, a);
  MG_LAUNCH_CHECK("gan_losses");
  return MGGAN_OK;
}

int mggan_steps_to_rows(const float* a, const float* b, int T, int n, float* out, hipStream_t stream) {
  MG_CHECK_ARG(T >= 0 && n >= 0, "steps_to_rows: negative size");
  const int rows = b ? 2 * n : n;
  if ((long)rows * T == 0) return MGGAN_OK;
  MG_CHECK_ARG(a && out, "steps_to_rows: null pointer");
  MG_LAUNCH(steps_to_rows_kernel, dim3(cdiv((long)rows * T, 256)), dim3(256), 0, stream, a, b, T, n, rows, out);
  MG_LAUNCH_CHECK("steps_to_rows");
  return MGGAN_OK;
}

/* the first `rows` rows only of steps laid out (T, n, 2): out (rows, 2T) */
int mggan_steps_to_rows_n(const float* a, int T, int n, int rows, float* out, hipStream_t stream) {
  MG_CHECK_ARG(T >= 0 && n >= 0 && rows >= 0 && rows <= n, "steps_to_rows_n: bad sizes (%d of %d rows)", rows, n);
  if ((long)rows * T == 0) return MGGAN_OK;
  MG_CHECK_ARG(a && out, "steps_to_rows_n: null pointer");
  MG_LAUNCH(steps_to_rows_kernel, dim3(cdiv((long)rows * T, 256)), dim3(256), 0, stream, a, (const float*)nullptr, T, n, rows, out);
  MG_LAUNCH_CHECK("steps_to_rows_n");
  return MGGAN_OK;
}

int mggan_scale(float* x, long n, const float* scalar, hipStream_t stream) {
  if (n == 0) return MGGAN_OK;
  MG_CHECK_ARG(x && scalar, "scale: null pointer");
  MG_LAUNCH(scale_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, x, n, scalar);
  MG_LAUNCH_CHECK("scale");
  return MGGAN_OK;
}

int mggan_d_assemble_fwd(int b, int K, int w_soc, int w_in, int w_pred, int w_scene, int soc_all, const float* soc0,
                         const float* in_enc, const float* pred_enc, const float* scene, float* X,
                         hipStream_t stream) {
  MG_CHECK_ARG(soc0 && in_enc && pred_enc && scene && X, "d_assemble_fwd: null pointer");
  const long n = (long)K * b * (w_soc + w_in + w_pred + w_scene);
  if (n == 0) return MGGAN_OK;
  MG_LAUNCH(d_assemble_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, b, K, w_soc, w_in, w_pred, w_scene,
                     soc_all, soc0, in_enc, pred_enc, scene, X);
  MG_LAUNCH_CHECK("d_assemble_fwd");
  return MGGAN_OK;
}

int mggan_d_assemble_bwd(int b, int K, int w_soc, int w_in, int w_pred, int w_scene, int soc_all, const float* dX,
                         float* dsoc0, float* din_enc, float* dpred_enc, float* dscene, hipStream_t stream) {
  MG_CHECK_ARG(dX, "d_assemble_bwd: null pointer");
  const long n = (long)b * (w_soc + w_in + w_pred + w_scene);
  if (n == 0) return MGGAN_OK;
  MG_LAUNCH(d_assemble_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, b, K, w_soc, w_in, w_pred, w_scene,
                     soc_all, dX, dsoc0, din_enc, dpred_enc, dscene);
  MG_LAUNCH_CHECK("d_assemble_bwd");
  return MGGAN_OK;
}

int mggan_ce_rows(int rows, int g, const float* logits, int ld, const int* target, const float* inv_count, float scale,
                  float* loss_rows, float* dlogits, int ldd, hipStream_t stream) {
  if (rows == 0) return MGGAN_OK;
  MG_CHECK_ARG(logits && target && loss_rows && g > 0, "ce_rows: bad arguments");
  MG_LAUNCH(ce_rows_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, stream, rows, g, logits, ld, target,
                     inv_count, scale, loss_rows, dlogits, ldd);
  MG_LAUNCH_CHECK("ce_rows");
  return MGGAN_OK;
}

int mggan_l2_min_scene(int S, int T, int K, int b, const int* scenes, const int* ped_scene, const float* gen_abs,
                       const float* gt, float grad_scale, float* scene_loss, int* scene_arg, float* gabs, const int* dims,
                       hipStream_t stream) {
  if (S == 0) return MGGAN_OK;
  MG_CHECK_ARG(scenes && ped_scene && gen_abs && gt && scene_loss && scene_arg, "l2_min_scene: null pointer");
  MG_CHECK_ARG(K <= L2_MAXK, "l2_min_scene: %d samples per pedestrian not built (<= %d)", K, L2_MAXK);
  MG_LAUNCH(l2_scene_kernel, dim3(S), dim3(256), 0, stream, S, T, K, b, scenes, gen_abs, gt, scene_loss,
                     scene_arg, dims);
  MG_LAUNCH_CHECK("l2_scene");
  if (gabs) {
    MG_LAUNCH(l2_grad_kernel, dim3(cdiv((long)T * K * b, 256)), dim3(256), 0, stream, T, K, b, ped_scene,
                       scene_arg, gen_abs, gt, grad_scale, gabs, dims);
    MG_LAUNCH_CHECK("l2_grad");
  }
  return MGGAN_OK;
}

int mggan_pm_ml_loss(int b, int T, int E, int g, const float* gen_abs, const float* gt, const float* logits, float sigma,
                     float scale, float* loss_rows, float* dlogits, float* probs, hipStream_t stream) {
  if (b == 0) return MGGAN_OK;
  MG_CHECK_ARG(gen_abs && gt && logits && loss_rows && dlogits && g <= 16, "pm_ml_loss: bad arguments (g <= 16)");
  int G2 = 1;
  while (G2 < g) G2 *= 2;
  const int per = 256 / G2;
  MG_LAUNCH(pm_ml_kernel, dim3(cdiv(b, per)), dim3(256), 0, stream, b, T, E, g, G2, gen_abs, gt, logits, sigma,
                     scale, loss_rows, dlogits, probs, (double*)nullptr, (unsigned*)nullptr, (float*)nullptr,
                     (float*)nullptr, 0.f, (const int*)nullptr);
  MG_LAUNCH_CHECK("pm_ml_loss");
  return MGGAN_OK;
}

int mggan_pm_ml_loss_mean(int b, int T, int E, int g, const float* gen_abs, const float* gt, const float* logits,
                          float sigma, float scale, float* loss_rows, float* dlogits, float* probs, double* partial,
                          unsigned* ticket, float* out, float* probs_out, float probs_scale, const int* dims,
                          hipStream_t stream) {
  MG_CHECK_ARG(out, "pm_ml_loss_mean: null pointer");
  if (b == 0) {
    MG_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(float), stream), "pm_ml_loss_mean: memset");
    return MGGAN_OK;
  }
  MG_CHECK_ARG(gen_abs && gt && logits && loss_rows && dlogits && partial && ticket && g <= 16,
               "pm_ml_loss_mean: bad arguments (g <= 16)");
  int G2 = 1;
  while (G2 < g) G2 *= 2;
  const int per = 256 / G2;
  int wgs = cdiv(b, per);
  if (wgs > PM_MAX_WG) wgs = PM_MAX_WG;
  MG_LAUNCH(pm_ml_kernel, dim3(wgs), dim3(256), 0, stream, b, T, E, g, G2, gen_abs, gt, logits, sigma, scale,
                     loss_rows, dlogits, probs, partial, ticket, out, probs_out, probs_scale, dims);
  MG_LAUNCH_CHECK("pm_ml_loss_mean");
  return MGGAN_OK;
}

int mggan_pm_mgan_loss(int b, int g, const float* logits, float target_weight, float reg, const float* reg_dev, float scale,
                       float* loss_rows, float* dlogits, float* probs, hipStream_t stream) {
  if (b == 0) return MGGAN_OK;
  MG_CHECK_ARG(logits && loss_rows && dlogits && g > 0, "pm_mgan_loss: bad arguments");
  MG_LAUNCH(pm_mgan_kernel, dim3(cdiv(b, 128)), dim3(128), 0, stream, b, g, logits, target_weight, reg, reg_dev, scale,
                     loss_rows, dlogits, probs);
  MG_LAUNCH_CHECK("pm_mgan_loss");
  return MGGAN_OK;
}

int mggan_pm_target(int b, int T, int E, int g, int mode, const float* gen_abs, const float* gt, int* target,
                    hipStream_t stream) {
  if (b == 0) return MGGAN_OK;
  MG_CHECK_ARG(gen_abs && gt && target && (mode == 0 || mode == 1), "pm_target: bad arguments");
  MG_LAUNCH(pm_target_kernel, dim3(cdiv(b, 128)), dim3(128), 0, stream, b, T, E, g, mode, gen_abs, gt, target);
  MG_LAUNCH_CHECK("pm_target");
  return MGGAN_OK;
}

int mggan_sum(const float* x, long n, float alpha, float* out, int accumulate, hipStream_t stream) {
  MG_CHECK_ARG(out && (x || n == 0), "sum: null pointer");
  MG_LAUNCH(sum_kernel, dim3(1), dim3(1024), 0, stream, x, n, alpha, out, accumulate);
  MG_LAUNCH_CHECK("sum");
  return MGGAN_OK;
}

int mggan_colmean(const float* x, int rows, int g, float scale, float* out, hipStream_t stream) {
  MG_CHECK_ARG(x && out && rows > 0 && g > 0, "colmean: bad arguments");
  MG_LAUNCH(colmean_kernel, dim3(g), dim3(256), 0, stream, x, rows, g, scale, out);
  MG_LAUNCH_CHECK("colmean");
  return MGGAN_OK;
}

int mggan_gen_counts(const int* idx, int n, int g, int* counts, float* inv_count, const int* dims, int bmod,
                     hipStream_t stream) {
  MG_CHECK_ARG(idx && counts && inv_count && g <= 256, "gen_counts: bad arguments");
  MG_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int) * g, stream), "gen_counts: memset");
  if (n > 0) {
    int blocks = cdiv(n, 2048);
    MG_LAUNCH(count_kernel, dim3(blocks > 64 ? 64 : blocks), dim3(256), 0, stream, idx, n, g, counts, dims, bmod);
  }
  MG_LAUNCH(inv_count_kernel, dim3(1), dim3(256), 0, stream, counts, g, inv_count);
  MG_LAUNCH_CHECK("gen_counts");
  return MGGAN_OK;
}

/* out: 16 doubles -- how often each generator is picked by a sampling launch on (logits (b, g), u (b*K)); scratch: 17 ints,
   zero before the first call (the launch leaves them zero) */
int mggan_sample_counts(int b, int K, int g, const float* logits, const float* u, int* scratch, double* out,
                        hipStream_t stream) {
  MG_CHECK_ARG(out && scratch && b >= 1 && logits && u && g >= 1 && g <= BR_MAXG && K >= 0 && K <= 255,
               "sample_counts: bad arguments (1..16 generators, up to 255 samples)");
  if (K == 0) {
    MG_CHECK_HIP(hipMemsetAsync(out, 0, BR_MAXG * sizeof(double), stream), "sample_counts: memset");
    return MGGAN_OK;
  }
  long wgs = cdiv((long)b * K, 256);  // one draw per lane up to 256 workgroups (a round trip per draw: walking eight draws
  if (wgs > 256) wgs = 256;           //  per lane in 13 workgroups took as long as the launch this replaced)
  MG_LAUNCH(sample_counts_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, b, K, g, logits, u, scratch, out);
  MG_LAUNCH_CHECK("sample_counts");
  return MGGAN_OK;
}

/* inv_count[i] = 1 / counts[i] (0 for an empty generator) from counts held as doubles (summed over the ranks) */
int mggan_inv_counts_f64(const double* counts, int g, float* inv_count, hipStream_t stream) {
  MG_CHECK_ARG(counts && inv_count && g >= 1 && g <= 256, "inv_counts_f64: bad arguments");
  MG_LAUNCH(inv_count_f64_kernel, dim3(1), dim3(256), 0, stream, counts, g, inv_count);
  MG_LAUNCH_CHECK("inv_counts_f64");
  return MGGAN_OK;
}

int mggan_inv_counts(const int* counts, int g, float* inv_count, hipStream_t stream) {
  MG_CHECK_ARG(counts && inv_count && g <= 256, "inv_counts: bad arguments");
  MG_LAUNCH(inv_count_kernel, dim3(1), dim3(256), 0, stream, counts, g, inv_count);
  MG_LAUNCH_CHECK("inv_counts");
  return MGGAN_OK;
}

/* workspace: 600 doubles, ZERO before the first call (partial sums + the grid barrier's counters + the tail flag + the
   finalized tail block; left ready for the next).  comm (device CommArgs of the calling stream's channel, mggan_comm_channel_create) and tail: see the header. */
int mggan_clip_adamw(float* param, float* grad, float* m, float* v, long n, const int* elem_seg, int nseg,
                     const unsigned char* active, int* seg_step, float max_norm, double lr, const double* lr_dev,
                     double beta1, double beta2, double eps, double weight_decay, int zero_grad, double* workspace,
                     float* norm_out, const void* comm, const mggan_grad_tail_t* tail, hipStream_t stream) {
  MG_CHECK_ARG(param && grad && m && v && elem_seg && active && seg_step && workspace, "clip_adamw: null pointer");
  if (n == 0) return MGGAN_OK;
  GradTail gt;
  memset(&gt, 0, sizeof(gt));
  if (tail && tail->tail) {
    MG_CHECK_ARG(tail->gram && tail->W && tail->bias && tail->gamma && tail->stat && tail->dW && tail->dgamma && tail->dbeta &&
                     (tail->C == 8 || tail->C == 16) && tail->n2 >= (long)tail->C * 38 && tail->n2 <= COMM_CHUNK,
                 "clip_adamw: incomplete gradient tail");
    MG_CHECK_ARG(tail->dW >= grad && tail->dW + tail->C * 36 <= grad + n && tail->dgamma >= grad && tail->dgamma + tail->C <= grad + n &&
                     tail->dbeta >= grad && tail->dbeta + tail->C <= grad + n,
                 "clip_adamw: the tail's gradient slots lie outside the flat gradient buffer");
    gt.tail = tail->tail; gt.n2 = tail->n2; gt.gram = tail->gram; gt.W = tail->W; gt.bias = tail->bias; gt.gamma = tail->gamma;
    gt.stat = tail->stat; gt.dW = tail->dW; gt.dgamma = tail->dgamma; gt.dbeta = tail->dbeta; gt.C = tail->C;
  }
  int blocks = cdiv(n, CA_CHUNK);
  if (blocks > CA_MAX_BLOCKS - 1) blocks = CA_MAX_BLOCKS - 1;
  if (gt.tail) blocks += 1;  // the tail's own workgroup
  MG_LAUNCH(clip_adamw_kernel, dim3(blocks), dim3(256), 0, stream, param, grad, m, v, n, elem_seg, active, nseg, seg_step,
            workspace, max_norm, lr, lr_dev, beta1, beta2, eps, weight_decay, zero_grad, norm_out, (const CommArgs*)comm, gt);
  MG_LAUNCH_CHECK("clip_adamw");
  return MGGAN_OK;
}

}  // extern "C"
