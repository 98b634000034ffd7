// Scene CNN + physical (channel-softmax) attention for MG-GAN on gfx950: attention head and conv2 adjoint.
// (conv1, its pooling decision, conv2 forward, the image Gram matrix and the conv1 weight gradient: csrc/cnn2.hip)
//
// Replaces (file:line under /root/reference/mggan/model/modules/cnn.py):
//   Conv_Blocks  :119-160   Conv2d(3x3,p1) -> BatchNorm2d -> ReLU -> MaxPool2d(2)   (x2, C = 16 in G, 8 in D)
//   CNN.forward  :275-282   (B,4,33,33) -> (B,C,16,16) -> (B,C,8,8)
//   AttentionGlobal.forward :109-116  per position: MLP C->32->C (LeakyReLU .01), softmax over CHANNELS, sum_c a_c x_c
// BatchNorm runs in train mode on the hot path (abstract_train.py:111-112): batch statistics sit between conv and
// ReLU, so each block is "conv + partial sums" -> finalize (fused into the producing kernel's last workgroup) -> the
// NEXT kernel applies scale/shift + ReLU (+ 2x2 max-pool) in its prologue while staging its input tile into LDS.
#include <stdlib.h>
#include "common.h"
#include "comm_dev.h"
#include "../../include/mggan_hip.h"

#define IH 33
#define A1_LD 20      // padded row stride of the 18x18 (16x16 + halo) tiles
#define A1_PLANE (18 * A1_LD)
#define HID 32

struct BnBwdFin {  // fused BatchNorm-backward finalize (same layout as in cnn2.hip)
  unsigned* ticket;
  double count;
  const float* gamma;
  const float* stat;   // [mean | invstd] of the forward pass
  float* coef;         // [gamma*invstd | mean(g) | mean(g*xhat)]
  double* coefd;       // [gamma*invstd | S1 | S2 | mean | invstd] + count (f64; may be NULL)
  float* dgamma;
  float* dbeta;
  const CommArgs* comm;  // sharded training: the last workgroup exchanges the folded sums before it finalizes (cnn2.hip)
};

__device__ __forceinline__ void load6(const float* p, float r[6]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float2 b = *reinterpret_cast<const float2*>(p + 4);
  r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y;
}

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

__device__ __forceinline__ void bn_bwd_finalize_lane(const BnBwdFin& f, int C, const double* sums) {
  const int c = threadIdx.x;
  if (c >= C) return;
  f.coef[c] = f.gamma[c] * f.stat[C + c];
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

// The last workgroup's finalize with the exchange of sharded training folded in (cnn2.hip: bn_bwd_finalize_block): the
// coefficients come from the GLOBAL sums, dgamma / dbeta take this rank's share.  sums: >= 2C + 1 doubles, local: 2C.
__device__ __forceinline__ void bn_bwd_finalize_block(BnBwdFin fin, int C, double* sums, double* local) {
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

// Sharded training, a rank WITHOUT images: it launched no attention adjoint, but its peers' last workgroups wait for its
// share of the exchange.
__global__ __launch_bounds__(256) void bn_bwd_sync_empty_kernel(int C, BnBwdFin fin) {
  __shared__ double sums[64], local[32];
  if ((int)threadIdx.x < 64) sums[threadIdx.x] = 0.0;
  __syncthreads();
  bn_bwd_finalize_block(fin, C, sums, local);
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

// ---------------- attention head: BN2+ReLU+pool prologue, 64 positions per image ----------------
// AttentionGlobal (cnn.py:109-116) per pooled position: MLP C -> 32 -> C (LeakyReLU .01), softmax over CHANNELS,
// out = sum_c a_c v_c -- forward and adjoint, INCLUDING the weight gradients of both layers, on v_mfma_f32_16x16x4_f32.
// A workgroup owns whole images, wave g of it the g-th group of 16 positions of each.  Inside a group lane (pp, kq) =
// (lane & 15, lane >> 4) stands for position 16 g + pp and for channels 4 kq + r, r = 0..3 (C = 8: the upper two quarters
// are padding -- zero weights, zero activations, score -inf):
//   * the prologue (BatchNorm 2 scale/shift, ReLU, 2x2 max-pool with the reference's first-maximum rule) leaves v[r] of
//     the lane's four channels in registers: exactly the B operand of h^T (32 x 16 pos) = [Wa | ba] [v ; 1] when the
//     reduction index is walked as c = 4 k + step (an MFMA's k index may be permuted as long as A agrees);
//   * its D fragment (unit 16 t + 4 kq + r of the lane's position) is the B operand of s^T (C x 16 pos) = [Wb | bb] [h ; 1],
//     whose D fragment is the score of channel 4 kq + r -- the lane's own channels again: softmax = four registers + two
//     cross-row shuffles, out = a . v likewise;
//   * backward: ds -> dh^T = Wb^T ds^T -> dz = dh * leaky'(h) -> dv^T += Wa^T dz^T, every product taking the previous D
//     registers as its B operand; nothing is transposed;
//   * the weight gradients contract over POSITIONS (the N axis of all of the above): dz, h, v, ds of a group go through
//     7 KB of wave-private LDS tiles (row strides 36 / 20 floats: conflict-free both ways) into A / B fragments, dWa / dWb
//     accumulate in registers over every image of the wave, biases as lane-local sums; one partial block per workgroup
//     for the batched fixed-order reduction.
// The fused-tile kernel this replaces wrote ds / v / dz / h (2C + 64 floats per position, 201 MB per launch at 8,192
// images) for a weight-gradient GEMM to read back -- the largest item of that GEMM's batch -- and ran the MLP on the
// VALU with LDS-resident weights.
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define AT_LDU 36  // dz / h tiles [16 pos][32 + 4]
#define AT_LDC 20  // v / ds tiles [16 pos][16 + 4]
#define AT_TILE_FLOATS (2 * 16 * AT_LDU + 2 * 16 * AT_LDC)

__device__ __forceinline__ void at_wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// partial block of a workgroup: [32][C + 1] = dWa | dba, then [C][33] = dWb | dbb
#define AT_WG_FLOATS(C) (32 * ((C) + 1) + (C) * 33)

// Work split (round 5): a WORKGROUP owns whole images, wave g of it the g-th group of 16 positions of every one of them
// (before: a wave walked the four groups of its images one after the other -- four dependent chains of matrix products per
// image and wave; at 1,280 images that serial walk WAS the launch).  What the forward pass decides per pooled cell -- the
// raw conv2 value that wins its 2x2 window and which of the four it is -- is kept (ysel: one float, ycode: one byte per
// channel and cell, both (B, 64 cells, C): a lane's four channels are one 16-byte and one 4-byte access) and is all the
// adjoint reads back: 5 bytes instead of the 16-byte window, and the gradient it routes
// to the raw conv2 grid leaves as ONE float per cell (g2sel; the position is ycode) instead of a 2x2 block with three
// zeros.  Per image and channel: 0.58 KB in + 0.25 KB out instead of 1 KB + 1 KB (268 MB per launch at 8,192 images, C = 16).
template <int C, bool BWD>
__global__ __launch_bounds__(256) void attn_kernel(int B, const float* __restrict__ y2, const float* __restrict__ scale2,
                                                   const float* __restrict__ shift2, const float* __restrict__ Wa,
                                                   const float* __restrict__ ba, const float* __restrict__ Wb,
                                                   const float* __restrict__ bb, float* out, int ld_out,
                                                   float* ysel_out, unsigned char* ycode_out,  // forward: saved selection
                                                   // backward only
                                                   const float* __restrict__ ysel, const unsigned char* __restrict__ ycode,
                                                   const float* __restrict__ dout, int ld_dout,
                                                   const float* __restrict__ stat2, float* g2sel, float* wpart, double* part,
                                                   BnBwdFin fin, const int* dims) {
  MG_REAL_IMAGES_COUNT(B, dims, fin)
  if (!BWD && B < B_padded_) {
    // phantom pedestrians of a padded batch get a zero scene feature (finite: their rows travel through every row-wise
    // kernel downstream and meet zero gradients in the weight-gradient products)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < (B_padded_ - B) * 64; i += gridDim.x * 256)
      out[(size_t)(B + i / 64) * ld_out + (i & 63)] = 0.f;
  }
  constexpr int WGF = AT_WG_FLOATS(C);
  constexpr int SMF = BWD ? (4 * AT_TILE_FLOATS > 4 * WGF ? 4 * AT_TILE_FLOATS : 4 * WGF) : 1;
  __shared__ __attribute__((aligned(16))) float smem[SMF];
  __shared__ double sred[BWD ? 4 * 2 * C : 1];
  __shared__ double colsum[40], cred[BWD ? 8 * 32 : 1];
  __shared__ int flag;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, pp = lane & 15, kq = lane >> 4;
  const bool chq = 4 * kq < C;  // this lane's channel quarter exists
  const int pos = 16 * w + pp, py = pos >> 3, px = pos & 7;  // wave w = group w of every image of the workgroup
  // ---- loop-invariant MFMA A operands ----
  // h^T = Wa v^T: step r <-> channel 4 k + r;  s^T = Wb h^T: step (t, r) <-> unit 16 t + 4 k + r
  float wa_a[2][4], wb_a[2][4];
  // dh^T = Wb^T ds^T (step r <-> channel 4 k + r), dv^T = Wa^T dz^T (step (t, r) <-> unit 16 t + 4 k + r)
  float wbt_a[2][4], wat_a[2][4];
  f32x4 ba_c[2], bb_c;  // the biases in the D layout of their product: accumulator init
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int u = 16 * t + pp;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 4 * kq + r, um = 16 * t + 4 * kq + r;
      wa_a[t][r] = c < C ? Wa[u * C + c] : 0.f;
      wb_a[t][r] = pp < C ? Wb[pp * HID + um] : 0.f;
      if (BWD) {
        wbt_a[t][r] = c < C ? Wb[c * HID + u] : 0.f;
        wat_a[t][r] = pp < C ? Wa[um * C + pp] : 0.f;
      }
      ba_c[t][r] = ba[um];
    }
  }
  float sc2[4], sh2[4], mean2[4], istd2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = chq ? 4 * kq + r : 0;
    bb_c[r] = chq ? bb[c] : 0.f;
    sc2[r] = scale2[c];
    sh2[r] = shift2[c];
    if (BWD) {
      mean2[r] = stat2[c];
      istd2[r] = stat2[C + c];
    }
  }
  // ---- accumulators over every image of this wave (backward) ----
  f32x4 dWa[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // dWa[u = 16 t + 4 kq + r][c = pp]
  f32x4 dWb[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // dWb[c = 4 kq + r][u = 16 t + pp]
  float dba[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dbb[4] = {0.f, 0.f, 0.f, 0.f};  // lane-local (over positions)
  // BatchNorm-2 adjoint sums of channel 4 kq + r: a lane adds ONE term per image (its position), in f64 from the start
  double s1d[4] = {0.0, 0.0, 0.0, 0.0}, s2d[4] = {0.0, 0.0, 0.0, 0.0};
  float* DZs = smem + (BWD ? w * AT_TILE_FLOATS : 0);
  float* Hs = DZs + 16 * AT_LDU;
  float* Vs = Hs + 16 * AT_LDU;
  float* DSs = Vs + 16 * AT_LDC;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

  // what a lane needs of image b: forward -- the 2x2 windows of its four channels at its position (raw conv2 output);
  // backward -- the saved winner of each window, its index, and the gradient of the attended feature at its position
  struct In {
    f32x4 win[4];
    float sel[4];
    int code[4];
    float go;
  };
  auto load_in = [&](int b, In& in) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (!BWD) {
        if (chq) {
          const float* base = y2 + (((size_t)b * C + 4 * kq + r) * 16 + 2 * py) * 16 + 2 * px;
          const float2 lo = *reinterpret_cast<const float2*>(base), hi = *reinterpret_cast<const float2*>(base + 16);
          in.win[r] = f32x4{lo.x, lo.y, hi.x, hi.y};
        } else {
          in.win[r] = zero;
        }
      }
    }
    if (BWD) {  // (cell-major, channel innermost: the lane's four channels are 16 + 4 consecutive bytes)
      const size_t ci = ((size_t)b * 64 + pos) * C + (chq ? 4 * kq : 0);
      const f32x4 sv = *reinterpret_cast<const f32x4*>(ysel + ci);
      const unsigned cw = *reinterpret_cast<const unsigned*>(ycode + ci);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        in.sel[r] = sv[r];
        in.code[r] = (int)((cw >> (8 * r)) & 0xffu);
      }
      in.go = dout[(size_t)b * ld_dout + pos];
    }
  };
  // the inputs of the NEXT image are in flight while this one is computed
  In cur, nxt;
  if ((int)blockIdx.x < B) load_in(blockIdx.x, cur);
#pragma unroll 1
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    if (b + (int)gridDim.x < B) load_in(b + gridDim.x, nxt);
    // prologue: v = maxpool2(relu(bn2(y2))), first maximum wins (pool_bn_relu)
    float v[4], raw[4];
    int code[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (!BWD) {
        float best = fmaxf(fmaf(cur.win[r][0], sc2[r], sh2[r]), 0.f);
        code[r] = 0;
        raw[r] = cur.win[r][0];
#pragma unroll
        for (int k = 1; k < 4; ++k) {
          const float z = fmaxf(fmaf(cur.win[r][k], sc2[r], sh2[r]), 0.f);
          if (z > best) { best = z; code[r] = k; raw[r] = cur.win[r][k]; }
        }
        v[r] = chq ? best : 0.f;
      } else {  // the forward pass kept the winner: the same expression gives the same v
        raw[r] = cur.sel[r];
        code[r] = cur.code[r];
        v[r] = chq ? fmaxf(fmaf(raw[r], sc2[r], sh2[r]), 0.f) : 0.f;
      }
    }
    if (!BWD && ysel_out && chq) {
      const size_t ci = ((size_t)b * 64 + pos) * C + 4 * kq;
      *reinterpret_cast<f32x4*>(ysel_out + ci) = f32x4{raw[0], raw[1], raw[2], raw[3]};
      *reinterpret_cast<unsigned*>(ycode_out + ci) =
          (unsigned)code[0] | ((unsigned)code[1] << 8) | ((unsigned)code[2] << 16) | ((unsigned)code[3] << 24);
    }
    // h = leaky(Wa v + ba): D fragment = unit 16 t + 4 kq + r of position pp
    f32x4 hp[2] = {ba_c[0], ba_c[1]};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      hp[0] = MFMA16(wa_a[0][r], v[r], hp[0]);
      hp[1] = MFMA16(wa_a[1][r], v[r], hp[1]);
    }
    f32x4 h[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[t][r] = hp[t][r] > 0.f ? hp[t][r] : 0.01f * hp[t][r];  // nn.LeakyReLU(), cnn.py:19-20
    // s = Wb h + bb: D fragment = channel 4 kq + r (two accumulator chains)
    f32x4 sa = bb_c, sb = zero;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sa = MFMA16(wb_a[0][r], h[0][r], sa);
      sb = MFMA16(wb_a[1][r], h[1][r], sb);
    }
    const f32x4 sc = sa + sb;
    float mx = chq ? fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])) : -INFINITY;
    mx = quarters_max(mx);
    float a[4], den = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a[r] = chq ? __expf(sc[r] - mx) : 0.f;
      den += a[r];
    }
    const float inv = 1.f / quarters_sum(den);
    float o = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a[r] *= inv;
      o = fmaf(a[r], v[r], o);
    }
    o = quarters_sum(o);
    if (!BWD) {
      if (kq == 0) out[(size_t)b * ld_out + pos] = o;
    } else {
      // ---- backward: out = sum_c a_c v_c, a = softmax(s) ----
      const float go = cur.go;
      const float dot = go * o;  // sum_c a_c (go v_c)
      f32x4 dv, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dv[r] = go * a[r];
        ds[r] = a[r] * (go * v[r] - dot);
        dbb[r] += ds[r];
      }
      // dh^T = Wb^T ds^T; dz = dh * leaky'(h)
      f32x4 dz[2] = {zero, zero};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dz[0] = MFMA16(wbt_a[0][r], ds[r], dz[0]);
        dz[1] = MFMA16(wbt_a[1][r], ds[r], dz[1]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dz[t][r] *= hp[t][r] > 0.f ? 1.f : 0.01f;
          dba[t][r] += dz[t][r];
        }
      // dv^T += Wa^T dz^T (two accumulator chains)
      f32x4 dv2 = zero;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dv = MFMA16(wat_a[0][r], dz[0][r], dv);
        dv2 = MFMA16(wat_a[1][r], dz[1][r], dv2);
      }
      dv += dv2;
      // through max-pool + ReLU: ONE value per pooled cell (it sits on window position ycode of the raw conv2 grid);
      // partial sums of the BatchNorm-2 adjoint
      if (chq) {
        f32x4 gq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gg = v[r] > 0.f ? dv[r] : 0.f;
          gq[r] = gg;
          s1d[r] += (double)gg;
          s2d[r] += (double)(gg * ((raw[r] - mean2[r]) * istd2[r]));
        }
        *reinterpret_cast<f32x4*>(g2sel + ((size_t)b * 64 + pos) * C + 4 * kq) = gq;
      }
      // weight gradients: contraction over the 16 positions through the wave's LDS tiles (pos = step + 4 k)
      *reinterpret_cast<f32x4*>(DZs + pp * AT_LDU + 4 * kq) = dz[0];
      *reinterpret_cast<f32x4*>(DZs + pp * AT_LDU + 16 + 4 * kq) = dz[1];
      *reinterpret_cast<f32x4*>(Hs + pp * AT_LDU + 4 * kq) = h[0];
      *reinterpret_cast<f32x4*>(Hs + pp * AT_LDU + 16 + 4 * kq) = h[1];
      *reinterpret_cast<f32x4*>(Vs + pp * AT_LDC + 4 * kq) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(DSs + pp * AT_LDC + 4 * kq) = ds;
      at_wave_lds_sync();
#pragma unroll
      for (int sp = 0; sp < 4; ++sp) {
        const int pr = sp + 4 * kq;
        const float vb = Vs[pr * AT_LDC + pp], da = DSs[pr * AT_LDC + pp];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          dWa[t] = MFMA16(DZs[pr * AT_LDU + 16 * t + pp], vb, dWa[t]);
          dWb[t] = MFMA16(da, Hs[pr * AT_LDU + 16 * t + pp], dWb[t]);
        }
      }
      at_wave_lds_sync();  // the tiles are rewritten by the next image
    }
    cur = nxt;
  }
  if (!BWD) return;
  // ---- one partial block per workgroup: the four waves meet in LDS (fixed order) ----
  __syncthreads();
  {
    float* mine = smem + w * WGF;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int u = 16 * t + 4 * kq + r;
        if (pp < C) mine[u * (C + 1) + pp] = dWa[t][r];
        const float sb = row_sum16(dba[t][r]);
        if (pp == 0) mine[u * (C + 1) + C] = sb;
        if (chq) mine[32 * (C + 1) + (4 * kq + r) * 33 + 16 * t + pp] = dWb[t][r];
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float sb = row_sum16(dbb[r]);
      if (pp == 0 && chq) mine[32 * (C + 1) + (4 * kq + r) * 33 + 32] = sb;
      // the 16 positions of the wave's group, in lane order (f64 xor-fold over the DPP row)
      double t1 = s1d[r], t2 = s2d[r];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        t1 += __shfl_xor(t1, o, 64);
        t2 += __shfl_xor(t2, o, 64);
      }
      if (pp == 0 && chq) {
        sred[w * 2 * C + 4 * kq + r] = t1;
        sred[w * 2 * C + C + 4 * kq + r] = t2;
      }
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < WGF; e += 256)
    wpart[(size_t)blockIdx.x * WGF + e] = (smem[e] + smem[WGF + e]) + (smem[2 * WGF + e] + smem[3 * WGF + e]);
  if ((int)threadIdx.x < 2 * C)
    store_part(part + (size_t)blockIdx.x * 2 * C + threadIdx.x,
               (sred[threadIdx.x] + sred[2 * C + threadIdx.x]) + (sred[4 * C + threadIdx.x] + sred[6 * C + threadIdx.x]));
  if (!fin.ticket) return;
  if (!last_block(fin.ticket, &flag)) return;
  colsum_rows(part, gridDim.x, 2 * C, colsum, cred);
  bn_bwd_finalize_block(fin, C, colsum, sred);  // (sred: 8C doubles, free by now)
}

// ---------------- conv2 backward: BN2 bwd + weight grad + input grad routed through pool1/ReLU ----------------
template <int C>
__global__ __launch_bounds__(256) void conv2_bwd_kernel(int B, const float* __restrict__ xsel,
                                                        const float* __restrict__ scale1,
                                                        const float* __restrict__ shift1,
                                                        const float* __restrict__ stat1, const float* __restrict__ y2,
                                                        const float* __restrict__ g2sel,
                                                        const unsigned char* __restrict__ ycode,
                                                        const float* __restrict__ stat2,
                                                        const float* __restrict__ coef2, const float* __restrict__ W,
                                                        float* G1c, double* part1, float* wpart, BnBwdFin fin,
                                                        const int* dims) {
  MG_REAL_IMAGES_COUNT(B, dims, fin)
  constexpr int COT = C / 4, PAIRS = C * C, NQ = 256 / PAIRS, ROWS = 16 / NQ, WLEN = PAIRS * 9 + C;
  __shared__ __attribute__((aligned(16))) float dyp[C * A1_PLANE];
  __shared__ __attribute__((aligned(16))) float a1p[C * A1_PLANE];
  __shared__ float y1r[C * 256];
  __shared__ double colsum[32], cred[8 * 32];
  __shared__ int flag;
  double ds1[COT], ds2[COT];
#pragma unroll
  for (int ci = 0; ci < COT; ++ci) ds1[ci] = ds2[ci] = 0.0;
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
        // the pooled cell's raw conv1 value (the window extreme conv1_pool kept)
        const size_t gi = ((size_t)b * C + c) * 256 + threadIdx.x;
        const float raw = xsel[gi];
        a1p[c * A1_PLANE + (py + 1) * A1_LD + px + 1] = fmaxf(fmaf(raw, scale1[c], shift1[c]), 0.f);
        y1r[c * 256 + threadIdx.x] = raw;
        const float xh = (y2[gi] - stat2[c]) * stat2[C + c];
        // the gradient that reached the raw conv2 grid: one value per pooled cell, on window position ycode
        const size_t ce = ((size_t)b * 64 + (py >> 1) * 8 + (px >> 1)) * C + c;
        const float gr = (int)ycode[ce] == (py & 1) * 2 + (px & 1) ? g2sel[ce] : 0.f;
        dyp[c * A1_PLANE + (py + 1) * A1_LD + px + 1] = coef2[c] * (gr - coef2[C + c] - xh * coef2[2 * C + c]);
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
      ds1[ci] += (double)wave_sum(s1);
      ds2[ci] += (double)wave_sum(s2);
    }
  }
  if (pg == 0) {
#pragma unroll
    for (int ci = 0; ci < COT; ++ci) {
      store_part(part1 + (size_t)blockIdx.x * 2 * C + cg * COT + ci, ds1[ci]);
      store_part(part1 + (size_t)blockIdx.x * 2 * C + C + cg * COT + ci, ds2[ci]);
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
  if (!fin.ticket) return;
  if (!last_block(fin.ticket, &flag)) return;
  colsum_rows(part1, gridDim.x, 2 * C, colsum, cred);
  bn_bwd_finalize_lane(fin, C, colsum);
}

// ---------------- conv2 backward on the matrix cores (C = 16) ----------------
// Same staging as conv2_bwd_kernel; both GEMM-shaped parts run on exact-f32 MFMA (16x16x4):
//   weight grad  dW[co][(tap,ci)] += sum_pos dy[co][pos] * a1[ci][pos+tap]   M=co(16), N=(tap,ci)(9 tiles), K=pos
//   input grad   da1[pos][ci]      = sum_{tap,co} dy[co][pos+tap'] * Wflip   M=pos (16 rows of 16), N=ci, K=(tap,co)
// LDS: ds_read_b32 serves 32 lanes per pass over 32 banks, i.e. two values of fk at a time.  Planes are 386 floats
// apart (== 2 mod 32): a fragment read that walks the 16 channels (fi) at two adjacent positions (fk) hits banks
// 2 fi + fk, all 32 distinct; the reads that walk 16 adjacent positions (fi) of two planes take planes EIGHT apart
// (8 * 386 == 16 mod 32) - the reduction index of the input-gradient product is ordered co = c + 8 (fk & 1) +
// 4 (fk >> 1) for that, on both operands - and the flipped weights have row stride 18 (8 * 18 == 16 mod 32).  With
// 361 (== 9 mod 32, chosen for 16-lane passes) half of this kernel's LDS cycles were bank conflicts.  Wave w owns image rows 4w..4w+3 in both products; the wave-private weight-grad
// accumulators persist over the workgroup's images and meet in LDS once at the end.
#define C2_PLANE 386
#define C2_WLD 18  // row stride of the flipped weights [tap'][co][ci]
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// C = 8 (the discriminator's CNN) on the same code: planes 388 apart (4 * 388 == 16 mod 32: the input-gradient product
// orders its reduction index co = c + 4 (fk & 1) + 2 (fk >> 1)); the weight-gradient product packs TWO taps into one N
// tile (columns 0..7: tap t0, columns 8..15: tap t1 -- pairs (0,2) (3,5) (6,8) (1,4) (7,-): the first three sit two
// columns apart, i.e. on disjoint banks), 5 MFMAs per k step instead of 9; the M tile (co) and the N tile of the input
// gradient (ci) are half empty.  152 MFMAs per wave and image against 288 at C = 16.
template <int C>
struct C2Geo {
  static constexpr int PLANE = C == 16 ? 386 : 388;
  static constexpr int NT = C == 16 ? 9 : 5;  // N tiles of the weight-gradient product
  static constexpr int KS = C / 4;            // k steps per tap of the input-gradient product
  static constexpr int LDS_FLOATS = 2 * C * PLANE + C * 256 + 9 * C * C2_WLD + 128;
};

template <int C>
__global__ __launch_bounds__(256) void conv2_bwd_mfma_kernel(int B, const float* __restrict__ xsel,
                                                            const float* __restrict__ scale1,
                                                            const float* __restrict__ shift1,
                                                            const float* __restrict__ stat1,
                                                            const float* __restrict__ y2,
                                                            const float* __restrict__ g2sel,
                                                            const unsigned char* __restrict__ ycode,
                                                            const float* __restrict__ stat2,
                                                            const float* __restrict__ coef2, const float* __restrict__ W,
                                                            float* G1c, double* part1, float* wpart, BnBwdFin fin,
                                                            const int* dims) {
  MG_REAL_IMAGES_COUNT(B, dims, fin)
  constexpr int WLEN = C * C * 9 + C, PLANE = C2Geo<C>::PLANE, NT = C2Geo<C>::NT, KS = C2Geo<C>::KS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dyp = smem;                       // [C][18][20] padded planes, stride PLANE
  float* a1p = dyp + C * PLANE;            // same layout
  float* y1r = a1p + C * PLANE;            // [C][256] raw conv1 value at the pooling argmax
  float* wf = y1r + C * 256;               // [tap'][co][ci] = W[co][ci][8 - tap'] (flipped kernel)
  float* red = wf + 9 * C * C2_WLD;        // [4][32] cross-wave statistics
  __shared__ double colsum[32], cred[8 * 32];
  __shared__ int flag;
  double dstat = 0.0;                      // threads 0..2C-1: this workgroup's sum g (0..C-1) / sum g*xhat (C..2C-1)
  for (int i = threadIdx.x; i < 2 * C * PLANE; i += 256) dyp[i] = 0.f;  // dyp and a1p (halos stay zero)
  for (int i = threadIdx.x; i < 9 * C * C; i += 256) {
    const int tp = i / (C * C), co = (i / C) % C, ci = i % C;
    wf[(tp * C + co) * C2_WLD + ci] = W[(co * C + ci) * 9 + (8 - tp)];
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  // weight-gradient product: A[i = co = fi][k = position], B[k = position][j]: C = 16: j = ci, one tap per tile;
  // C = 8: j = ci + 8 * which tap of the tile's pair.  boff: the lane's a1 plane and tap offset per tile (-1: no tap)
  const int cfi = fi & (C - 1);
  const bool arow = fi < C;
  int boff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int tap;
    if (C == 16) {
      tap = t;
    } else {
      const int t0 = t < 3 ? 3 * t : (t == 3 ? 1 : 7), t1 = t < 3 ? 3 * t + 2 : (t == 3 ? 4 : -1);
      tap = fi < 8 ? t0 : t1;
    }
    boff[t] = tap < 0 ? -1 : cfi * PLANE + (tap / 3) * A1_LD + tap % 3;
  }
  // input-gradient product: reduction index of k step (tap', c): output channel of this lane's k group
  int cok[KS];
#pragma unroll
  for (int c = 0; c < KS; ++c) cok[c] = C == 16 ? c + 8 * (fk & 1) + 4 * (fk >> 1) : c + 4 * (fk & 1) + 2 * (fk >> 1);
  f32x4_t wacc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wacc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bacc = 0.f;

  // the three planes of the NEXT image (raw conv1 value, raw conv2 output, its gradient) are in flight while this one is
  // computed
  // (the gradient on the raw conv2 grid arrives as ONE value per pooled cell + the window position that won the cell in the
  //  forward pass -- attn_kernel: g2sel / ycode -- and is spread onto this thread's position here)
  // Both are (B, 64 cells, C): the C values of this thread's cell are C/4 16-byte loads + one of C bytes, kept RAW until
  // the staging pass (a select on a value just loaded would put the wait for it right behind the prefetch).
  float rx[C], ry[C];
  f32x4_t rg[C / 4];
  unsigned rcode[C / 4];
  const int my_cell = (threadIdx.x >> 5) * 8 + ((threadIdx.x & 15) >> 1);
  const unsigned my_k = ((threadIdx.x >> 4) & 1) * 2 + (threadIdx.x & 1);
  auto fetch = [&](int b) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const size_t gi = ((size_t)b * C + c) * 256 + threadIdx.x;
      rx[c] = xsel[gi];
      ry[c] = y2[gi];
    }
    const size_t ce = ((size_t)b * 64 + my_cell) * C;
#pragma unroll
    for (int q = 0; q < C / 4; ++q) {
      rg[q] = *reinterpret_cast<const f32x4_t*>(g2sel + ce + 4 * q);
      rcode[q] = *reinterpret_cast<const unsigned*>(ycode + ce + 4 * q);
    }
  };
  if ((int)blockIdx.x < B) fetch(blockIdx.x);
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    lds_barrier();
    {
      const int py = threadIdx.x >> 4, px = threadIdx.x & 15;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float raw = rx[c];
        a1p[c * PLANE + (py + 1) * A1_LD + px + 1] = fmaxf(fmaf(raw, scale1[c], shift1[c]), 0.f);
        y1r[c * 256 + threadIdx.x] = raw;
        const float xh = (ry[c] - stat2[c]) * stat2[C + c];
        const float gr = ((rcode[c >> 2] >> (8 * (c & 3))) & 0xffu) == my_k ? rg[c >> 2][c & 3] : 0.f;
        dyp[c * PLANE + (py + 1) * A1_LD + px + 1] = coef2[c] * (gr - coef2[C + c] - xh * coef2[2 * C + c]);
      }
    }
    if (b + (int)gridDim.x < B) fetch(b + gridDim.x);
    lds_barrier();
    // ---- (1) weight gradient: K runs over this wave's 64 positions (rows 4w..4w+3), 4 positions per MFMA
#pragma unroll 8  // (two steps per trip: 209 / 128 us at 8,192 images with C = 16 / 8; eight: 200 / 117)
    for (int ks = 0; ks < 16; ++ks) {
      const int y = 4 * w + (ks >> 2), x0 = (ks & 3) * 4;
      // A[i = co][k = position x0+fk]
      const float av = dyp[cfi * PLANE + (y + 1) * A1_LD + x0 + 1 + fk];
      const float a = arow ? av : 0.f;
      bacc += a;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        // B[k = position][j]: a1 value at (y + ky, x + kx) in padded coordinates of the lane's (channel, tap)
        const float bq = a1p[(boff[t] < 0 ? 0 : boff[t]) + y * A1_LD + x0 + fk];
        const float bv = boff[t] < 0 ? 0.f : bq;
        wacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, wacc[t], 0, 0, 0);
      }
    }
    // ---- (2) input gradient for rows 4w..4w+3: M = 16 x-positions of one row, N = ci, K = (tap', co)
    float s1 = 0.f, s2 = 0.f;  // BN1-backward partial sums for channel ci = fi
    // (the wave's four rows together: four independent accumulator chains -- one chain of 36 dependent MFMAs per row
    //  paid the 40-cycle dependent latency on every one of them -- and one read of the weight fragment serves four rows)
    f32x4_t acc4[4];
#pragma unroll
    for (int ry_ = 0; ry_ < 4; ++ry_) acc4[ry_] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
      for (int c = 0; c < KS; ++c) {
        const int co = cok[c];
        // A[i = x][k = (tp, co)] = dy[co] at padded (y + tp/3, x + tp%3);  B[k][j = ci] = wf[tp][co][ci]
        const float bq = wf[(tp * C + co) * C2_WLD + cfi];
        const float bv = arow ? bq : 0.f;
#pragma unroll
        for (int ry_ = 0; ry_ < 4; ++ry_) {
          const float a = dyp[co * PLANE + (4 * w + ry_ + tp / 3) * A1_LD + fi + tp % 3];
          acc4[ry_] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc4[ry_], 0, 0, 0);
        }
      }
    }
    if (arow) {
#pragma unroll
      for (int ry_ = 0; ry_ < 4; ++ry_) {
        const int y = 4 * w + ry_;
        // D fragment: register r <-> x = 4*fk + r, column = ci = fi
        float g[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int x = 4 * fk + r;
          const bool on = a1p[fi * PLANE + (y + 1) * A1_LD + x + 1] > 0.f;
          g[r] = on ? acc4[ry_][r] : 0.f;
          const float xh = (y1r[fi * 256 + y * 16 + x] - stat1[fi]) * stat1[C + fi];
          s1 += g[r];
          s2 = fmaf(g[r], xh, s2);
        }
        *reinterpret_cast<float4*>(G1c + (((size_t)b * C + fi) * 16 + y) * 16 + 4 * fk) = make_float4(g[0], g[1], g[2], g[3]);
      }
    }
    // per-channel statistics: fold the 4 lane groups, then the 4 waves
    s1 = quarters_sum(s1);
    s2 = quarters_sum(s2);
    if (lane < 16) { red[w * 32 + lane] = s1; red[w * 32 + 16 + lane] = s2; }
    lds_barrier();
    if (threadIdx.x < 2 * C) {  // [0,C) = sum g, [C,2C) = sum g*xhat: f32 within an image, f64 across images
      const int q = threadIdx.x < C ? threadIdx.x : 16 + threadIdx.x - C;
      dstat += (double)((red[q] + red[32 + q]) + (red[64 + q] + red[96 + q]));
    }
  }
  if (threadIdx.x < 2 * C) store_part(part1 + (size_t)blockIdx.x * 2 * C + threadIdx.x, dstat);
  // ---- fold the four waves' weight-gradient fragments; wacc[t][r] of lane l is dW[co = 4*fk + r][ci][tap]
  lds_barrier();
  float* fold = dyp;  // reuse: [4][WLEN]
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    int tap;
    if (C == 16) {
      tap = t;
    } else {
      const int t0 = t < 3 ? 3 * t : (t == 3 ? 1 : 7), t1 = t < 3 ? 3 * t + 2 : (t == 3 ? 4 : -1);
      tap = fi < 8 ? t0 : t1;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * fk + r < C && tap >= 0) fold[w * WLEN + ((4 * fk + r) * C + cfi) * 9 + tap] = wacc[t][r];
  }
  bacc = quarters_sum(bacc);  // lanes 0..C-1: sum over this wave's positions of dy[co = lane]
  if (lane < C) fold[w * WLEN + C * C * 9 + lane] = bacc;
  lds_barrier();
  float* wp = wpart + (size_t)blockIdx.x * WLEN;
  for (int i = threadIdx.x; i < WLEN; i += 256)
    wp[i] = (fold[i] + fold[WLEN + i]) + (fold[2 * WLEN + i] + fold[3 * WLEN + i]);
  if (!fin.ticket) return;
  if (!last_block(fin.ticket, &flag)) return;
  colsum_rows(part1, gridDim.x, 2 * C, colsum, cred);
  bn_bwd_finalize_lane(fin, C, colsum);
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
  static int knob = -1;  // MGGAN_CNN_BWD_GRID: measurement knob -- exactly this many workgroups
  if (knob < 0) { const char* e = getenv("MGGAN_CNN_BWD_GRID"); knob = e ? atoi(e) : 0; }
  if (knob > 0) return B < knob ? B : knob;
  return B <= 512 ? B : 512;
}

static BnBwdFin make_bfin(unsigned* ticket, double count, const float* gamma, const float* stat, float* coef, double* coefd,
                          float* dgamma, float* dbeta, const void* comm = nullptr) {
  BnBwdFin f;
  f.comm = (const CommArgs*)comm;
  f.ticket = ticket; f.count = count; f.gamma = gamma; f.stat = stat; f.coef = coef; f.coefd = coefd; f.dgamma = dgamma;
  f.dbeta = dbeta;
  return f;
}

extern "C" {

int mggan_cnn_bwd_grid(int B) { return persistent_grid(B); }

int mggan_bn_finalize(const double* sums, double count, int C, int training, const float* gamma, const float* beta,
                      float* run_mean, float* run_var, long long* num_batches_tracked, float momentum, float eps,
                      float* scale, float* shift, float* stat, hipStream_t stream) {
  MG_CHECK_ARG(gamma && beta && run_mean && run_var && scale && shift && stat && C <= 64, "bn_finalize: bad arguments");
  MG_CHECK_ARG(!training || (sums && num_batches_tracked), "bn_finalize: training needs sums and the batch counter");
  MG_LAUNCH(bn_finalize_kernel, dim3(1), dim3(64), 0, stream, sums, count, C, training, gamma, beta, run_mean,
                     run_var, num_batches_tracked, momentum, eps, scale, shift, stat);
  MG_LAUNCH_CHECK("bn_finalize");
  return MGGAN_OK;
}

// persistent: a workgroup walks images b, b + grid, ... (its waves' weight fragments are loaded once): FOUR images per
// workgroup up to 2,048 workgroups, beyond that equal shares.  Measured (iteration, one box, MGGAN_ATTN_GRID = exactly that
// many): 1,280 images -- 160: 1.411, 320: 1.405, 428: 1.419, 640: 1.416-1.438, 1,280: 1.482 ms (a workgroup's partial block,
// ticket and weight fragments cost more than its second, third, fourth image); 8,192 images -- 1,024: 4.86, 2,048: 4.74,
// 4,096: 4.91, 8,192: 5.11 ms.
static int attn_grid(int B) {
  static int forced = -1;
  if (forced < 0) { const char* e = getenv("MGGAN_ATTN_GRID"); forced = e ? atoi(e) : 0; }
  if (forced > 0) return forced < B ? forced : B;
  // at least four images per workgroup, at most 2,048 workgroups -- and from 1,024 images on a multiple of the 256 CUs with
  // equal image counts: 1,280 images as 256 x 5 instead of 320 x 4 (64 CUs held two workgroups: 1.367 -> 1.360 ms per
  // configs[1] iteration, three alternating pairs; 512 workgroups: 1.375), 8,192 as 2,048 x 4 as before
  int k = B / 1024;
  k = k > 8 ? 8 : k;
  const int per = k >= 1 ? cdiv(B, 256 * k) : 4;
  return cdiv(B, per);
}

int mggan_scene_attention_grid(int B) { return B > 0 ? attn_grid(B) : 0; }
int mggan_scene_attention_partial_floats(int C) { return AT_WG_FLOATS(C); }

int mggan_scene_attention_fwd(const float* y2, int B, int C, const float* scale2, const float* shift2, const float* Wa,
                              const float* ba, const float* Wb, const float* bb, float* out, int ld_out, float* ysel,
                              unsigned char* ycode, const int* dims, hipStream_t stream) {
  MG_CHECK_ARG(C == 8 || C == 16, "scene_attention_fwd: channels %d not built (8 or 16)", C);
  if (B == 0) return MGGAN_OK;
  MG_CHECK_ARG(y2 && scale2 && shift2 && Wa && ba && Wb && bb && out, "scene_attention_fwd: null pointer");
  MG_CHECK_ARG((ysel == nullptr) == (ycode == nullptr), "scene_attention_fwd: ysel and ycode come together");
  const BnBwdFin none = make_bfin(nullptr, 0.0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (C == 16)
    MG_LAUNCH((attn_kernel<16, false>), dim3(attn_grid(B)), dim3(256), 0, stream, B, y2, scale2, shift2, Wa, ba, Wb, bb,
              out, ld_out, ysel, ycode, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, none, dims);
  else
    MG_LAUNCH((attn_kernel<8, false>), dim3(attn_grid(B)), dim3(256), 0, stream, B, y2, scale2, shift2, Wa, ba, Wb, bb,
              out, ld_out, ysel, ycode, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, none, dims);
  MG_LAUNCH_CHECK("scene_attention_fwd");
  return MGGAN_OK;
}

/* wpart: mggan_scene_attention_grid(B) partial blocks of mggan_scene_attention_partial_floats(C) floats ([32][C+1] = dWa |
 * dba, then [C][33] = dWb | dbb) for mggan_grad_reduce_multi; part: the same number of rows of 2C doubles (sum g | sum
 * g*xhat per workgroup).  ticket != NULL: the launch also finishes the BatchNorm-2 adjoint (coef2 = [gamma*invstd | mean
 * g | mean g*xhat], dgamma2 / dbeta2 accumulated) */
int mggan_scene_attention_bwd(const float* ysel, const unsigned char* ycode, int B, int C, const float* scale2,
                              const float* shift2, const float* stat2, const float* Wa, const float* ba, const float* Wb,
                              const float* bb, const float* dout, int ld_dout, float* g2sel, float* wpart, double* part,
                              unsigned* ticket, double count, const float* gamma2, float* coef2, float* dgamma2,
                              float* dbeta2, const void* comm, const int* dims, hipStream_t stream) {
  MG_CHECK_ARG(C == 8 || C == 16, "scene_attention_bwd: channels %d not built (8 or 16)", C);
  MG_CHECK_ARG(!comm || (ticket && !dims), "scene_attention_bwd: the in-launch exchange needs the fused finalize (ticket) and an "
                                          "unpadded batch");
  if (B == 0 && comm) {
    MG_CHECK_ARG(stat2 && gamma2 && coef2 && dgamma2 && dbeta2, "scene_attention_bwd: null pointer");
    MG_LAUNCH(bn_bwd_sync_empty_kernel, dim3(1), dim3(256), 0, stream, C,
              make_bfin(ticket, count, gamma2, stat2, coef2, nullptr, dgamma2, dbeta2, comm));
    MG_LAUNCH_CHECK("scene_attention_bwd");
    return MGGAN_OK;
  }
  if (B == 0) return MGGAN_OK;
  MG_CHECK_ARG(ysel && ycode && scale2 && shift2 && stat2 && Wa && ba && Wb && bb && dout && g2sel && wpart && part,
               "scene_attention_bwd: null pointer");
  MG_CHECK_ARG(!ticket || (gamma2 && coef2 && dgamma2 && dbeta2), "scene_attention_bwd: the fused finalize needs gamma / coef / grads");
  const BnBwdFin fin = make_bfin(ticket, count, gamma2, stat2, coef2, nullptr, dgamma2, dbeta2, comm);
  if (C == 16)
    MG_LAUNCH((attn_kernel<16, true>), dim3(attn_grid(B)), dim3(256), 0, stream, B, nullptr, scale2, shift2, Wa, ba, Wb,
              bb, nullptr, 0, nullptr, nullptr, ysel, ycode, dout, ld_dout, stat2, g2sel, wpart, part, fin, dims);
  else
    MG_LAUNCH((attn_kernel<8, true>), dim3(attn_grid(B)), dim3(256), 0, stream, B, nullptr, scale2, shift2, Wa, ba, Wb,
              bb, nullptr, 0, nullptr, nullptr, ysel, ycode, dout, ld_dout, stat2, g2sel, wpart, part, fin, dims);
  MG_LAUNCH_CHECK("scene_attention_bwd");
  return MGGAN_OK;
}

/* conv2 adjoint: BatchNorm-2 backward on the fly, weight gradient (per-workgroup partial rows in `workspace`:
 * mggan_cnn_bwd_grid(B) * (256/(C*C)) * (C*C*9 + C) floats, reduced into dW / db here or by the caller's batched
 * reduction when dW == NULL), input gradient routed through ReLU / max-pool of block 1 -> G1c (B,C,16,16) at pooled
 * resolution (it belongs to the window position conv1_pool recorded).  part1: mggan_cnn_bwd_grid(B) rows of 2C doubles; with a
 * ticket the launch also finishes the BatchNorm-1 adjoint (coef1, coefd1 for mggan_conv1_wgrad, dgamma1 / dbeta1). */
int mggan_conv2_bwd(const float* xsel, int B, int C, const float* scale1,
                    const float* shift1, const float* stat1, const float* y2, const float* g2sel,
                    const unsigned char* ycode, const float* stat2,
                    const float* coef2, const float* W, float* G1c, double* part1, float* dW,
                    float* db, float* workspace, size_t workspace_bytes, unsigned* ticket, double count1,
                    const float* gamma1, float* coef1, double* coefd1, float* dgamma1, float* dbeta1, const int* dims,
                    hipStream_t stream) {
  MG_CHECK_ARG(C == 8 || C == 16, "conv2_bwd: channels %d not built (8 or 16)", C);
  if (B == 0) return MGGAN_OK;
  MG_CHECK_ARG(xsel && scale1 && shift1 && stat1 && y2 && g2sel && ycode && stat2 && coef2 && W && G1c && part1 && workspace,
               "conv2_bwd: null pointer");
  MG_CHECK_ARG(!ticket || (gamma1 && coef1 && dgamma1 && dbeta1), "conv2_bwd: the fused finalize needs gamma / coef / grads");
  const int grid = persistent_grid(B), NQ = 256 / (C * C), wlen = C * C * 9 + C;
  const size_t need = (size_t)grid * NQ * wlen * sizeof(float);
  if (workspace_bytes < need) {
    mggan_set_error("conv2_bwd: workspace too small (%zu < %zu)", workspace_bytes, need);
    return MGGAN_ERR_WORKSPACE;
  }
  const BnBwdFin fin = make_bfin(ticket, count1, gamma1, stat1, coef1, coefd1, dgamma1, dbeta1);
  static int valu = -1;  // MGGAN_CONV2_VALU=1: the VALU kernel for C = 8 (A/B measurements)
  if (valu < 0) { const char* e = getenv("MGGAN_CONV2_VALU"); valu = e && e[0] == '1'; }
  if (C == 16 || !valu) {
    const size_t lds = (size_t)(C == 16 ? C2Geo<16>::LDS_FLOATS : C2Geo<8>::LDS_FLOATS) * sizeof(float);
    static bool attr[2] = {false, false};
    const void* fn = C == 16 ? (const void*)conv2_bwd_mfma_kernel<16> : (const void*)conv2_bwd_mfma_kernel<8>;
    if (!attr[C == 16]) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        mggan_set_error("conv2_bwd: cannot raise the dynamic LDS limit to %zu bytes", lds);
        return MGGAN_ERR_LAUNCH;
      }
      attr[C == 16] = true;
    }
    if (C == 16)
      MG_LAUNCH(conv2_bwd_mfma_kernel<16>, dim3(grid), dim3(256), lds, stream, B, xsel, scale1, shift1, stat1, y2, g2sel,
                ycode, stat2, coef2, W, G1c, part1, workspace, fin, dims);
    else
      MG_LAUNCH(conv2_bwd_mfma_kernel<8>, dim3(grid), dim3(256), lds, stream, B, xsel, scale1, shift1, stat1, y2, g2sel,
                ycode, stat2, coef2, W, G1c, part1, workspace, fin, dims);
  } else
    MG_LAUNCH((conv2_bwd_kernel<8>), dim3(grid), dim3(256), 0, stream, B, xsel, scale1, shift1, stat1, y2, g2sel, ycode,
              stat2, coef2, W, G1c, part1, workspace, fin, dims);
  MG_LAUNCH_CHECK("conv2_bwd");
  if (!dW) return MGGAN_OK;  // deferred reduce of the [grid][C*C*9 + C] partial rows
  MG_LAUNCH(partial_sum_kernel, dim3(cdiv(C * C * 9, 64)), dim3(1024), 0, stream, workspace, grid, wlen,
                     C * C * 9, dW);
  MG_LAUNCH(partial_sum_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, stream, workspace + C * C * 9, grid, wlen,
                     C, db);
  MG_LAUNCH_CHECK("conv2_bwd reduce");
  return MGGAN_OK;
}

}  // extern "C"
