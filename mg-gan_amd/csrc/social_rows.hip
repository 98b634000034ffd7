// Social attention of MG-GAN as ROW-structured, recomputing MFMA kernels for gfx950.
//
// Replaces (file:line under /root/reference/mggan/model/modules/social.py):
//   SocialFeatures / BearingMTX / DCA_MTX   :67-104  (distance, cos-bearing, DCA of ordered pairs)
//   EmbedSocialFeatures.fc                  :33-48   (MLP 3 -> 32 -> 64 -> F, ReLU)
//   AttentionPooling.forward                :14-30   (sigma_ij = f_ij . (W h_j + b); sigma_ii = -1000;
//                                                     softmax over the scene; S_i = sum_j a_ij h_j; n==1 -> 0)
// and their adjoints INCLUDING the weight gradients of the 3->32 and 32->64 layers.  Algebra as in social.hip: the last
// embedding layer is linear, sigma_ij = l2_ij . v_j + c_j with [v_j | c_j] = (W_at h_j + b_at) [W3 | b3] (65 values per
// pedestrian, computed by the chain kernel before this one).
//
// Work decomposition.  A workgroup (4 waves) owns whole scenes; wave w owns the attention rows i = w, w+4, ... of the
// scene and walks each row in blocks of 16 neighbours j.  Inside a block lane (pp, kq) = (lane & 15, lane >> 4) stands
// for neighbour j = 16 jb + pp and for the kq-th quarter of every feature axis:
//   * both layers of the pair MLP TRANSPOSED on v_mfma_f32_16x16x4_f32: l1^T (32 x 16 pairs) = [W1 | b1] (32 x 4)
//     [f ; 1] (4 x 16 pairs) is two MFMAs whose D fragment IS the B operand of the next product -- lane (pp, kq) gets
//     units u(s) = 16 (s>>2) + 4 kq + (s&3), s = 0..7, of its pair, and the reduction index of z2^T (64 x 16 pairs) =
//     W2 (64 x 32) l1^T is walked in exactly that order (an MFMA's k index may be permuted freely as long as A agrees);
//     b2 enters as one more k step against a row of ones; the D fragment of lane (pp, kq) is z2[pair pp][m = 16 t + 4 kq
//     + r].  The weights are MFMA operands only (accumulator-file registers), never VALU operands;
//   * the score is a 16-term dot product per lane with the lane's slice of v_j (row-invariant registers) + two
//     cross-row adds; the softmax over the scene is DPP row reductions + a loop over the <= 4 blocks: no LDS, no barrier
//     ("wavefront shuffle reductions for the attention softmax");
//   * backward: the adjoint dz1^T = W2^T dz2^T takes the D registers of the forward product AS ITS B OPERAND (the
//     reduction index of an MFMA may be permuted freely as long as A agrees: k-step (t, r) <-> m = 16 t + 4 kq + r), and
//     its D fragment lands on the units u(s) the lane already holds l1 for;
//   * sums over i for a fixed neighbour (dvc_j = sum_i dsigma_ij [l2_ij | 1], dh_j = sum_i a_ij dS_i) stay in the
//     registers of the lane that stands for j while the wave walks its rows; the four waves' partial sums meet in LDS
//     once per scene (fixed order);
//   * the weight gradient dW2 = sum_pairs dz2 l1^T contracts over PAIRS, the one index that sits on the N axis of the
//     products above: dz2 and l1 of a block are transposed through 6.5 KB of wave-private LDS (conflict-free both ways:
//     row strides 68 / 36 floats, pair = step + 4 k) into A / B fragments; dW2, dW1, db1, db2 accumulate in registers over
//     every scene of the workgroup and leave ONE partial block per workgroup for the batched fixed-order reduction
//     (grad_reduce_multi_kernel) -- nothing per pair is ever written to HBM (the fused tile kernels this file replaces
//     stored 100 floats per pair in the forward pass and 97 more in the backward pass for the weight-gradient GEMMs).
// Scenes of more than 64 pedestrians take the unfused kernels of social.hip.
#include <stdlib.h>
#include "common.h"
#include "../../include/mggan_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#define SR_LDV 68    // staged row of a scene: v_j (64) | c_j | pad
#define SR_LDZ 68    // dz2 transposition tile [16 pairs][64 + 4]
#define SR_LD1 36    // l1 / dz1 transposition tiles [16 pairs][32 + 4]; the dz1 tile's pad columns hold (f0, f1, f2, 1)
#define SR_TILE_FLOATS (16 * SR_LDZ + 2 * 16 * SR_LD1)  // wave-private transposition tiles
#define SR_WG_FLOATS (64 * 33 + 32 * 4)                 // partial block of a workgroup: dW2|db2 [64][33], dW1|db1 [32][4]

struct SocRowsArgs {
  const int* scenes;  // [S][2] = first / past-last pedestrian row
  const float *xy, *dxy;
  const float *W1, *b1, *W2, *b2;
  const float *Wat, *bat, *W3, *b3;  // attention map W_at (F x H) | b_at (F), last embedding layer W3 (F x 64) | b3 (F)
  const float* h;
  const float* dS;
  float *Sout, *dvc, *dh, *partials;
  float *Wh_out, *dWh_out;  // backward: Wh = h W_at^T + b_at and its gradient, (rows, F), for the weight-gradient GEMMs
  int F;                    // social feature width (<= 64)
  float* scratch;     // row_splits > 1: neighbour sums of every (scene, split), [unit][first row of the scene ...][65 + H]
  unsigned* tickets;  // row_splits > 1: one word per scene, zero between launches
  int S, xy_mod, ldv, ld_h, ld_s, ld_ds, ld_dh, accumulate_dh;
  int dvc_rows;       // rows of dvc (= pedestrians): stride of a split's share in `scratch`
  int row_splits;     // the rows of a scene are dealt to this many workgroups (few scenes: more of the chip at work)
};

// (row_sum16 / row_max16: all-lanes reductions over a DPP row, common.h)
// (sum over the four DPP rows of a wave -- the four kq quarters: quarters_sum, common.h)
// LDS traffic between the lanes of ONE wave (the transposition tiles): program order is execution order, the compiler
// only has to keep it
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifdef SR_PROFILE  // measurement build (tools/bench_social.py): segment cycle counts of workgroup 0, wave 0 -> a.scratch
#define SR_T(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); prof[k] += t_ - tlast; tlast = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SR_T(k) do {} while (0)
#endif

// SocialFeatures of the ordered pair (i, j): social.py:67-104 (same operation order as social.hip:pair_features)
__device__ __forceinline__ void sr_features(float pix, float piy, float vix, float viy, float pjx, float pjy, float vjx,
                                            float vjy, float f[3]) {
  const float dpx = pix - pjx, dpy = piy - pjy;
  const float dvx = vix - vjx, dvy = viy - vjy;
  const float dist = sqrtf(dpx * dpx + dpy * dpy);
  const float bearing = (dpx * vix + dpy * viy) / (dist * sqrtf(vix * vix + viy * viy) + 1e-6f);
  const float ttca = -(dpx * dvx + dpy * dvy) / (dvx * dvx + dvy * dvy + 1e-6f);
  const float cx = dpx + ttca * dvx, cy = dpy + ttca * dvy;
  f[0] = dist;
  f[1] = bearing;
  f[2] = sqrtf(cx * cx + cy * cy);
}

__device__ __forceinline__ int sr_unit(int s, int kq) { return 16 * (s >> 2) + 4 * kq + (s & 3); }

// loop-invariant weight fragments of a lane: MFMA A operands only
struct SrWeights {
  float w1a[2];     // [W1 | b1][16 tp + pp][kq]
  float w2a[4][8];  // W2[16 t + pp][u(s)]
  float b2a[4];     // b2[16 t + pp] on the kq == 0 lanes, 0 elsewhere (one k step against a row of ones)
};

__device__ __forceinline__ void sr_load_weights(const SocRowsArgs& a, int pp, int kq, SrWeights& W) {
#pragma unroll
  for (int tp = 0; tp < 2; ++tp) W.w1a[tp] = kq < 3 ? a.W1[(16 * tp + pp) * 3 + kq] : a.b1[16 * tp + pp];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int s = 0; s < 8; ++s) W.w2a[t][s] = a.W2[(16 * t + pp) * 32 + sr_unit(s, kq)];
    W.b2a[t] = kq == 0 ? a.b2[16 * t + pp] : 0.f;
  }
}

// what a lane keeps in registers about "its" neighbour j of a j-block (v_j and h_j are read from the staged scene)
struct SrNeighbour {
  float c, px, py, vx, vy;
  int j;
  bool ok;  // j < n
};

__device__ __forceinline__ void sr_load_neighbour(const SocRowsArgs& a, int s0, int n, int j, const float* vs, SrNeighbour& N) {
  N.ok = j < n;
  N.j = N.ok ? j : 0;
  const int jc = s0 + N.j;
  N.c = vs[N.j * SR_LDV + 64];
  const int jx = a.xy_mod > 0 ? jc % a.xy_mod : jc;
  N.px = a.xy[2 * jx]; N.py = a.xy[2 * jx + 1];
  N.vx = a.dxy[2 * jx]; N.vy = a.dxy[2 * jx + 1];
}

// The per-pedestrian dense stages run inside the same launch, per scene, on the staged rows: Wh_j = W_at h_j + b_at,
// [v_j | c_j] = Wh_j [W3 | b3] -- and their adjoints in the backward kernel -- as 16 x 16 MFMA tiles whose operands are
// read from LDS (the weights sit there for the life of the workgroup).  (A first version walked them on the VALU, two
// LDS reads per FMA: 14 us per 32-pedestrian scene of the discriminator, more than the launches it replaced.)
#define SR_LDW 68  // row stride of the staged [W3 | .] rows and of Wh / dWh in LDS (4 mod 32: see sr_tile)

// acc (+)= A[j0 .. j0+15][0 .. K) x B, one 16 x 16 tile per wave: acc[r] <-> row 4 kq + r, column pp.
// A row-major in LDS (K contiguous, lda % 4 == 0); NT: B given as Bt[col][k] (K contiguous); else B[k][col].
// The reduction index is walked as k = 16 S + 4 kq + r: one 16-byte read per operand and S (NT); strides that are
// 4 mod 32 keep both forms free of bank conflicts.
template <bool NT>
__device__ __forceinline__ f32x4 sr_tile(const float* A, int lda, int j0, const float* B, int ldb, int c0, int K, f32x4 acc,
                                         int pp, int kq) {
  const float* ar = A + (j0 + pp) * lda + 4 * kq;
  for (int S = 0; S < K; S += 16) {
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(ar + S);
    f32x4 b4;
    if (NT) {
      b4 = *reinterpret_cast<const f32x4*>(B + (c0 + pp) * ldb + S + 4 * kq);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) b4[r] = B[(S + 4 * kq + r) * ldb + c0 + pp];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = MFMA16(a4[r], b4[r], acc);
  }
  return acc;
}

struct SrDense {  // LDS of the dense stages
  float* wat;     // W_at [F][H + 4]
  float* w3;      // W3   [F][SR_LDW]
  float* bat;     // [64]
  float* b3;      // [64]
  float* wh;      // Wh / dWh of the scene [rows][SR_LDW]
};

template <int H, int NW>
__device__ __forceinline__ void sr_stage_dense_weights(const SocRowsArgs& a, const SrDense& D) {
  constexpr int NT = 64 * NW;
  const int F = a.F;
  // 16-byte loads (the weight matrices are contiguous and 16-byte aligned: checked by the launchers)
  for (int e = threadIdx.x; e < F * (H / 4); e += NT)
    *reinterpret_cast<f32x4*>(D.wat + (e / (H / 4)) * (H + 4) + 4 * (e % (H / 4))) = *reinterpret_cast<const f32x4*>(a.Wat + 4 * e);
  for (int e = threadIdx.x; e < F * 16; e += NT)
    *reinterpret_cast<f32x4*>(D.w3 + (e >> 4) * SR_LDW + 4 * (e & 15)) = *reinterpret_cast<const f32x4*>(a.W3 + 4 * e);
  for (int f = threadIdx.x; f < F; f += NT) {
    D.bat[f] = a.bat[f];
    // [W3 | b3]: c_j = Wh_j . b3 is column 64 of the product that gives v_j
    *reinterpret_cast<f32x4*>(D.w3 + f * SR_LDW + 64) = f32x4{a.b3[f], 0.f, 0.f, 0.f};
    D.b3[f] = a.b3[f];
  }
}

// h rows of the scene -> hs; Wh -> D.wh (and wh_out); [v | c] -> vs.  Every thread of the workgroup calls it.
template <int H, int NW>
__device__ __forceinline__ void sr_stage_scene(const SocRowsArgs& a, int s0, int n, const SrDense& D, float* vs, float* hs,
                                               float* wh_out) {
  constexpr int LDH = H + 4, NT = 64 * NW;
  const int F = a.F, lane = threadIdx.x & 63, w = threadIdx.x >> 6, pp = lane & 15, kq = lane >> 4;
  const int jt = (n + 15) >> 4;
  for (int e = threadIdx.x; e < n * (H / 4); e += NT) {
    const int row = e / (H / 4), c = e - row * (H / 4);
    *reinterpret_cast<f32x4*>(hs + row * LDH + 4 * c) =
        *reinterpret_cast<const f32x4*>(a.h + (size_t)(s0 + row) * a.ld_h + 4 * c);
  }
  __syncthreads();
  for (int t = w; t < jt * (F >> 4); t += NW) {  // Wh = h W_at^T + b_at
    const int j0 = 16 * (t / (F >> 4)), f0 = 16 * (t % (F >> 4));
    const float bias = D.bat[f0 + pp];
    const f32x4 acc = sr_tile<true>(hs, LDH, j0, D.wat, H + 4, f0, H, f32x4{bias, bias, bias, bias}, pp, kq);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = j0 + 4 * kq + r;
      D.wh[j * SR_LDW + f0 + pp] = acc[r];
      if (wh_out && j < n) wh_out[(size_t)(s0 + j) * F + f0 + pp] = acc[r];
    }
  }
  __syncthreads();
  for (int t = w; t < jt * 5; t += NW) {  // [v | c] = Wh [W3 | b3]  (the fifth column tile: c in its first column)
    const int j0 = 16 * (t / 5), m0 = 16 * (t % 5);
    const f32x4 acc = sr_tile<false>(D.wh, SR_LDW, j0, D.w3, SR_LDW, m0, F, f32x4{0.f, 0.f, 0.f, 0.f}, pp, kq);
    if (m0 + pp <= 64) {
#pragma unroll
      for (int r = 0; r < 4; ++r) vs[(j0 + 4 * kq + r) * SR_LDV + m0 + pp] = acc[r];
    }
  }
  __syncthreads();
}

// the pair MLP for the 16 pairs (row i, neighbours of one block): l1[s] = unit u(s) of the lane's pair,
// z[t][r] = PRE-activation of layer-2 unit 16 t + 4 kq + r
__device__ __forceinline__ void sr_pair_mlp(const SrWeights& W, int kq, const float f[3], float l1[8], f32x4 z[4]) {
  const float fb = kq == 0 ? f[0] : kq == 1 ? f[1] : kq == 2 ? f[2] : 1.0f;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 p0 = MFMA16(W.w1a[0], fb, zero), p1 = MFMA16(W.w1a[1], fb, zero);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    l1[r] = fmaxf(p0[r], 0.f);
    l1[4 + r] = fmaxf(p1[r], 0.f);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) z[t] = MFMA16(W.b2a[t], 1.0f, zero);
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int t = 0; t < 4; ++t) z[t] = MFMA16(W.w2a[t][s], l1[s], z[t]);
}

// sigma_ij = l2_ij . v_j + c_j  (social.py:25: sigma_ii = -1000; padding lanes: -inf)
__device__ __forceinline__ float sr_score(const f32x4 z[4], const float* vrow, const SrNeighbour& N, bool self) {
  float sp0 = 0.f, sp1 = 0.f;
#pragma unroll
  for (int t = 0; t < 4; t += 2) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(vrow + 16 * t), v1 = *reinterpret_cast<const f32x4*>(vrow + 16 * t + 16);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sp0 = fmaf(fmaxf(z[t][r], 0.f), v0[r], sp0);
      sp1 = fmaf(fmaxf(z[t + 1][r], 0.f), v1[r], sp1);
    }
  }
  const float sg = quarters_sum(sp0 + sp1) + N.c;
  return !N.ok ? -INFINITY : (self ? -1000.0f : sg);
}

// ---------------------------------------------------------------------------------------------------------------
// NW waves per workgroup.  A lone wave issues one 16x16x4 product every ~50 cycles (measured with s_memtime around the
// products of these kernels: 47 - 58 cycles each, chains of four accumulators or two alike), the pipe takes one every 32:
// two waves per SIMD fill it.
template <int H, int NJB, int NW>
__global__ __launch_bounds__(64 * NW) void social_rows_fwd_kernel(const SocRowsArgs a) {
  constexpr int HQ = H / 4, LDH = H + 4;
  __shared__ __attribute__((aligned(16))) float vs[16 * NJB * SR_LDV];
  __shared__ __attribute__((aligned(16))) float hs[16 * NJB * LDH];
  __shared__ __attribute__((aligned(16))) float wat_s[64 * (H + 4)], w3_s[64 * SR_LDW + 16], whs[16 * NJB * SR_LDW];
  __shared__ float bat_s[64], b3_s[64];
  const SrDense DN = {wat_s, w3_s, bat_s, b3_s, whs};
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, pp = lane & 15, kq = lane >> 4;
  SrWeights W;
  sr_load_weights(a, pp, kq, W);
  sr_stage_dense_weights<H, NW>(a, DN);
  const int RS = a.row_splits;
  for (int un = blockIdx.x; un < a.S * RS; un += gridDim.x) {
    const int sc = un / RS, rs = un - sc * RS;
    const int s0 = a.scenes[2 * sc], n = a.scenes[2 * sc + 1] - s0;
    if (n <= 1) {  // social.py:19-20: a lone pedestrian pools nothing
      if (n == 1 && rs == 0 && (int)threadIdx.x < H) a.Sout[(size_t)s0 * a.ld_s + threadIdx.x] = 0.f;
      continue;
    }
    __syncthreads();  // every wave is done with the previous scene's rows (and the dense weights are staged)
    sr_stage_scene<H, NW>(a, s0, n, DN, vs, hs, nullptr);
    SrNeighbour N[NJB];
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) sr_load_neighbour(a, s0, n, 16 * jb + pp, vs, N[jb]);
    for (int i = NW * rs + w; i < n; i += NW * RS) {
      const int gi = s0 + i, ix = a.xy_mod > 0 ? gi % a.xy_mod : gi;
      const float pix = a.xy[2 * ix], piy = a.xy[2 * ix + 1], vix = a.dxy[2 * ix], viy = a.dxy[2 * ix + 1];
      float sg[NJB];
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) {
        if (16 * jb >= n) {  // wave-uniform: the scene has no neighbour in this block
          sg[jb] = -INFINITY;
          continue;
        }
        float f[3], l1[8];
        f32x4 z[4];
        sr_features(pix, piy, vix, viy, N[jb].px, N[jb].py, N[jb].vx, N[jb].vy, f);
        sr_pair_mlp(W, kq, f, l1, z);
        sg[jb] = sr_score(z, vs + N[jb].j * SR_LDV + 4 * kq, N[jb], 16 * jb + pp == i);
      }
      float mx = sg[0];
#pragma unroll
      for (int jb = 1; jb < NJB; ++jb) mx = fmaxf(mx, sg[jb]);
      mx = row_max16(mx);
      float e[NJB], den = 0.f;
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) {
        e[jb] = N[jb].ok ? __expf(sg[jb] - mx) : 0.f;
        den += e[jb];
      }
      const float inv = 1.0f / row_sum16(den);
      float out[HQ];
#pragma unroll
      for (int k = 0; k < HQ; ++k) out[k] = 0.f;
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) {
        if (16 * jb >= n) continue;
        const float at = e[jb] * inv;
        const float* hrow = hs + N[jb].j * LDH + kq * HQ;
#pragma unroll
        for (int k = 0; k < HQ; k += 4) {
          const f32x4 h4 = *reinterpret_cast<const f32x4*>(hrow + k);
#pragma unroll
          for (int q = 0; q < 4; ++q) out[k + q] = fmaf(at, h4[q], out[k + q]);
        }
      }
#pragma unroll
      for (int k = 0; k < HQ; ++k) out[k] = row_sum16(out[k]);
      if (pp == 0) {
        float* dst = a.Sout + (size_t)gi * a.ld_s + kq * HQ;
#pragma unroll
        for (int k = 0; k < HQ; k += 4) *reinterpret_cast<f32x4*>(dst + k) = f32x4{out[k], out[k + 1], out[k + 2], out[k + 3]};
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward.  TRAIN: also the weight gradients of the pair MLP (one partial block per workgroup).  KEEP: the forward
// values of a row's blocks stay in registers between the score pass and the adjoint pass (else: recomputed).
//
// Registers decide how many waves a SIMD holds, and a lone wave leaves the matrix pipe a third idle (see the forward
// kernel): the round-3 form of this kernel kept 500 values per lane (one wave per SIMD).  What left the registers:
//   * the A-operand fragments of W2 (forward and transposed) sit in LDS, one row per LANE, shared by every wave (a
//     16-byte read feeds four products);
//   * da_ij = dS_i . h_j and the neighbour part of dh (dh_j = sum_i a_ij dS_i) are two small matrix products per scene
//     on the staged rows (DA = dS h^T before the rows are walked, dh = A^T dS after), not HQ running sums per lane and
//     block: the row walk reads da_ij from LDS and leaves a_ij in the same slot;
//   * the positions of the scene's pedestrians are staged in LDS (the head of a row was a global round trip).
#define SR_LDC 68  // fold row of a wave and neighbour: dv (64) | dc | pad

// The pair MLP's second layer in LDS: W2 row-major, row stride 36 (== 4 mod 32).  Both A-operand fragments come straight
// from it:  forward  W2[16 t + pp][16 sh + 4 kq + 0..3] = units u(4 sh + 0..3, kq): one 16-byte read feeds four products
//           adjoint  W2[16 t + 4 kq + r][16 tp + pp]: 4-byte reads, the four kq groups 16 banks apart
// and the bias the D fragment of the layer-2 product starts from, the same for the 16 lanes of a kq:
//   bzs[16 kq + 4 t + r] = b2[16 t + 4 kq + r]
#define SR_LD2 36
// An LDS address the compiler cannot prove loop-invariant: the reads behind it stay where they are written.  (Left alone it
// hoists the row-invariant fragments -- W2, v_j -- out of the row loop "into registers", finds none free, parks them in
// scratch and reloads them at the head of every row: 52 registers' worth, a global round trip per row.)
__device__ __forceinline__ int sr_here(int off) {  // (the OFFSET: a laundered pointer loses its address space and is read
  asm volatile("" : "+v"(off));                     //  with FLAT loads, which also count on vmcnt)
  return off;
}
// W2T (TRAIN, scenes of up to 32): the transposed copy, row stride 68 -- the adjoint fragments W2[16 t + 4 kq + 0..3][16 tp + pp]
// as ONE 16-byte read (32 four-byte reads per block kept the products waiting: 112 cycles apiece instead of 40)
#define SR_LD2T 68
template <int NW, bool W2T>
__device__ __forceinline__ void sr_stage_w2(const SocRowsArgs& a, float* w2s, float* w2t, float* bzs) {
  for (int e = threadIdx.x; e < 64 * 8; e += 64 * NW) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.W2 + 4 * e);
    *reinterpret_cast<f32x4*>(w2s + (e >> 3) * SR_LD2 + 4 * (e & 7)) = v;
    if (W2T) {
#pragma unroll
      for (int j = 0; j < 4; ++j) w2t[(4 * (e & 7) + j) * SR_LD2T + (e >> 3)] = v[j];
    }
  }
  if (threadIdx.x < 64) bzs[threadIdx.x] = a.b2[16 * ((threadIdx.x >> 2) & 3) + 4 * (threadIdx.x >> 4) + (threadIdx.x & 3)];
}

// sr_pair_mlp with the layer-2 fragments read from LDS (wf = w2s + pp * SR_LD2 + 4 kq, bz = bzs + 16 kq)
__device__ __forceinline__ void sr_pair_mlp_lds(const float w1a[2], const float* wf, const float* bz, int kq, const float f[3],
                                                float l1[8], f32x4 z[4]) {
  const float fb = kq == 0 ? f[0] : kq == 1 ? f[1] : kq == 2 ? f[2] : 1.0f;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 p0 = MFMA16(w1a[0], fb, zero), p1 = MFMA16(w1a[1], fb, zero);
#pragma unroll
  for (int t = 0; t < 4; ++t) z[t] = *reinterpret_cast<const f32x4*>(bz + 4 * t);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    l1[r] = fmaxf(p0[r], 0.f);
    l1[4 + r] = fmaxf(p1[r], 0.f);
  }
#pragma unroll
  for (int sh = 0; sh < 2; ++sh) {
    f32x4 w4[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) w4[t] = *reinterpret_cast<const f32x4*>(wf + 16 * t * SR_LD2 + 16 * sh);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) z[t] = MFMA16(w4[t][j], l1[4 * sh + j], z[t]);
  }
}

// KEEP: 1 = the forward values of every block of a row stay in registers between the score pass and the adjoint pass;
// 2 = those of the row's LAST block only (the adjoint pass starts with it, the others are recomputed); 0 = none.
template <int H, int NJB, int NW, bool TRAIN, int KEEP>
__global__ __launch_bounds__(64 * NW) void social_rows_bwd_kernel(const SocRowsArgs a) {
  constexpr int NT = 64 * NW, LDH = H + 4, NR = 16 * NJB;
  constexpr int DAL = NR + 4;           // row stride of DA / A (4 mod 32 or 20 mod 32: see the reads below)
  constexpr int JR = NJB <= 2 ? NJB : 1;  // neighbour blocks folded per exchange round
  constexpr int RED_FLOATS = NW * 16 * JR * SR_LDC;
  constexpr int TILES = TRAIN ? NW * SR_TILE_FLOATS : 0, EXCH = TRAIN ? NW * SR_WG_FLOATS : 0, WHS = NR * SR_LDW;
  constexpr int A0 = RED_FLOATS > TILES ? RED_FLOATS : TILES, A1 = EXCH > WHS ? EXCH : WHS, A_FLOATS = A0 > A1 ? A0 : A1;
  // region A, one use at a time (barriers between): Wh of the scene while it is staged; the waves' transposition tiles while
  // rows are walked; the neighbour-sum exchange; dWh in the epilogue; the weight-gradient exchange at the end of the workgroup
  __shared__ __attribute__((aligned(16))) float smem[A_FLOATS];
  __shared__ __attribute__((aligned(16))) float vs[NR * SR_LDV];  // [v | c] of the scene; its gradient at the end
  __shared__ __attribute__((aligned(16))) float hs[NR * LDH];     // h of the scene; the neighbour part of dh at the end
  __shared__ __attribute__((aligned(16))) float dss[NR * LDH];    // dS of the scene (pad rows zero)
  __shared__ __attribute__((aligned(16))) float das[NR * DAL];    // da_ij, replaced by a_ij as the rows are walked
  __shared__ __attribute__((aligned(16))) float pvs[NR * 4];      // (x, y, dx, dy) of the scene's pedestrians
  __shared__ __attribute__((aligned(16))) float wat_s[64 * (H + 4)], w3_s[64 * SR_LDW + 16];
  constexpr bool W2T = TRAIN;
  constexpr int NFS = KEEP == 0 ? 1 : NJB;  // feature slots of a wave (no KEEP: the adjoint pass computes them again, one block at a time)
  __shared__ __attribute__((aligned(16))) float w2s[64 * SR_LD2], w2ts[W2T ? 32 * SR_LD2T : 4];
  __shared__ __attribute__((aligned(16))) float bzs[64];
  __shared__ __attribute__((aligned(16))) float fss[NW * NFS * 64];  // the pair features of a row's blocks, per wave
  __shared__ float bat_s[64], b3_s[64];
  float* whs = smem;
  const SrDense DN = {wat_s, w3_s, bat_s, b3_s, whs};
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, pp = lane & 15, kq = lane >> 4;
  float* Zs = smem + w * SR_TILE_FLOATS;  // [16][SR_LDZ]
  float* L1s = Zs + 16 * SR_LDZ;          // [16][SR_LD1]
  float* Gs = L1s + 16 * SR_LD1;          // [16][SR_LD1]: dz1 | (f0, f1, f2, 1)
  float* red = smem;                      // [NW][16 JR][SR_LDC]
  const int wfo = pp * SR_LD2 + 4 * kq;                                          // forward fragments (in w2s)
  const int wto = W2T ? pp * SR_LD2T + 4 * kq : 4 * kq * SR_LD2 + pp;            // adjoint fragments (in w2ts / w2s)
  float w1a[2];
#pragma unroll
  for (int tp = 0; tp < 2; ++tp) w1a[tp] = kq < 3 ? a.W1[(16 * tp + pp) * 3 + kq] : a.b1[16 * tp + pp];
  sr_stage_w2<NW, W2T>(a, w2s, w2ts, bzs);
  sr_stage_dense_weights<H, NW>(a, DN);
  bool wrote = false;  // this workgroup's partial block has been started

  const int RS = a.row_splits;
  __shared__ int last_s;
#ifdef SR_PROFILE
  unsigned long long prof[16], tlast = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < 16; ++k) prof[k] = 0;
#endif
  SR_T(0);  // 0: fragments, dense weights issued
  for (int un = blockIdx.x; un < a.S * RS; un += gridDim.x) {
    const int sc = un / RS, rs = un - sc * RS;
    const int s0 = a.scenes[2 * sc], n = a.scenes[2 * sc + 1] - s0;
    if (n <= 1) {  // no attention, no gradient: dvc = dWh = 0 (Wh: any finite value), dh untouched (or 0)
      int tid_n = threadIdx.x;  // (afresh: see below)
      asm volatile("" : "+v"(tid_n));
      if (n == 1 && rs == 0) {
        if (tid_n < 65) a.dvc[(size_t)s0 * a.ldv + tid_n] = 0.f;
        if (tid_n < a.F) {
          a.Wh_out[(size_t)s0 * a.F + tid_n] = 0.f;
          a.dWh_out[(size_t)s0 * a.F + tid_n] = 0.f;
        }
        if (!a.accumulate_dh && tid_n < H) a.dh[(size_t)s0 * a.ld_dh + tid_n] = 0.f;
      }
      continue;
    }
    const int F = a.F, jt = (n + 15) >> 4;
    __syncthreads();  // (the fragments and dense weights are staged; the previous scene's epilogue is through with everything)
    SR_T(1);  // 1: weights staged (barrier)
    // (thread indices afresh for the staging stage: what the prologue derived from them would otherwise be carried through the row
    // loop, where no register is free -- in scratch)
    int tid_s = threadIdx.x;
    asm volatile("" : "+v"(tid_s));
    const int w_s = tid_s >> 6, pp_s = tid_s & 15, kq_s = (tid_s >> 4) & 3;
    // ---- stage the scene: h, dS (pad rows zero: they enter the two small products as operands), positions
    for (int e = tid_s; e < 16 * jt * (H / 4); e += NT) {
      const int row = e / (H / 4), c = e - row * (H / 4);
      f32x4 h4 = {0.f, 0.f, 0.f, 0.f}, d4 = {0.f, 0.f, 0.f, 0.f};
      if (row < n) {
        h4 = *reinterpret_cast<const f32x4*>(a.h + (size_t)(s0 + row) * a.ld_h + 4 * c);
        d4 = *reinterpret_cast<const f32x4*>(a.dS + (size_t)(s0 + row) * a.ld_ds + 4 * c);
      }
      *reinterpret_cast<f32x4*>(hs + row * LDH + 4 * c) = h4;
      *reinterpret_cast<f32x4*>(dss + row * LDH + 4 * c) = d4;
    }
    for (int j = tid_s; j < 16 * jt; j += NT) {
      const int jc = s0 + (j < n ? j : 0), jx = a.xy_mod > 0 ? jc % a.xy_mod : jc;
      *reinterpret_cast<f32x4*>(pvs + 4 * j) = f32x4{a.xy[2 * jx], a.xy[2 * jx + 1], a.dxy[2 * jx], a.dxy[2 * jx + 1]};
    }
    __syncthreads();
    SR_T(2);  // 2: h, dS, positions staged
    for (int t = w_s; t < jt * (F >> 4); t += NW) {  // Wh = h W_at^T + b_at
      const int j0 = 16 * (t / (F >> 4)), f0 = 16 * (t % (F >> 4));
      const float bias = bat_s[f0 + pp_s];
      const f32x4 acc = sr_tile<true>(hs, LDH, j0, wat_s, H + 4, f0, H, f32x4{bias, bias, bias, bias}, pp_s, kq_s);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + 4 * kq_s + r;
        whs[j * SR_LDW + f0 + pp_s] = acc[r];
        if (rs == 0 && j < n) a.Wh_out[(size_t)(s0 + j) * F + f0 + pp_s] = acc[r];
      }
    }
    // DA = dS h^T (rows of another split: zero -- the slot ends as a_ij, and this workgroup's A^T dS takes its own rows only)
    for (int t = w_s; t < jt * jt; t += NW) {
      const int i0 = 16 * (t / jt), j0 = 16 * (t % jt);
      const f32x4 acc = sr_tile<true>(dss, LDH, i0, hs, LDH, j0, H, f32x4{0.f, 0.f, 0.f, 0.f}, pp_s, kq_s);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 4 * kq_s + r;
        das[i * DAL + j0 + pp_s] = (RS == 1 || (i / NW) % RS == rs) ? acc[r] : 0.f;
      }
    }
    __syncthreads();
    SR_T(3);  // 3: Wh, DA
    for (int t = w_s; t < jt * 5; t += NW) {  // [v | c] = Wh [W3 | b3]  (the fifth column tile: c in its first column)
      const int j0 = 16 * (t / 5), m0 = 16 * (t % 5);
      const f32x4 acc = sr_tile<false>(whs, SR_LDW, j0, w3_s, SR_LDW, m0, F, f32x4{0.f, 0.f, 0.f, 0.f}, pp_s, kq_s);
      if (m0 + pp_s <= 64) {
#pragma unroll
        for (int r = 0; r < 4; ++r) vs[(j0 + 4 * kq_s + r) * SR_LDV + m0 + pp_s] = acc[r];
      }
    }
    __syncthreads();  // (Wh is dead: region A belongs to the waves' tiles now)
    SR_T(4);  // 4: v, c

    // ---- the rows
    // The weight-gradient accumulators live for the rows of ONE scene and are folded into the workgroup's partial block
    // right behind them: kept across scenes they sat in registers through the staging and epilogue tile products, which
    // need the room -- the compiler parked 50 of them in scratch and fetched them back, twice per scene.
    f32x4 dW2[4][2];     // dW2[m = 16 tm + 4 kq + r][u = 16 tu + pp]
    f32x4 dW1[2];        // [dW1 | db1][u = 16 tp + pp][c = 0..3], partial over the lane's pairs (folded over kq below): vector
                         // FMAs -- as products with a 4-column B tile they were 8 of 106 per block for 128 outputs
    float db2[4];        // db2[16 tm + pp], partial over the lane's pairs (folded over kq below)
    if (TRAIN) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        dW2[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        dW2[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        db2[t] = 0.f;
      }
      dW1[0] = dW1[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // (What a lane knows about its neighbour j of a block -- position, c_j -- is read from the staged scene per block, and the
    // pair features wait for the adjoint pass in a wave-private LDS slot: registers held across the row loop are what
    // decides whether eight waves fit.)
    f32x4 dv[NJB][4];
    float dc[NJB];
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) {
#pragma unroll
      for (int t = 0; t < 4; ++t) dv[jb][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      dc[jb] = 0.f;
    }
    float* fsl = fss + w * (NFS * 64);  // [NFS][16 pairs][f0 f1 f2 1]
    for (int i = NW * rs + w; i < n; i += NW * RS) {
      // pass 1 over the row's blocks: the forward values (kept or thrown away) and the score
      // (l1 is not kept across the passes: two products bring it back, 8 registers per block do not fit)
      constexpr int NK = KEEP == 1 ? NJB : 1;
      const int jbl = (n - 1) >> 4;  // the row's last block
      float sg[NJB], da[NJB];
      f32x4 z[NK][4];
      SR_T(5);  // 5: row head
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) {
        if (16 * jb >= n) {
          sg[jb] = -INFINITY;
          da[jb] = 0.f;
          continue;
        }
        const int q = KEEP == 1 ? jb : 0;
        const bool ok = 16 * jb + pp < n;
        const int j = ok ? 16 * jb + pp : 0;
        da[jb] = das[i * DAL + 16 * jb + pp];
        float f[3], l1[8];
        {
          const f32x4 qi = *reinterpret_cast<const f32x4*>((pvs + sr_here(4 * i)));
          const f32x4 qj = *reinterpret_cast<const f32x4*>((pvs + sr_here(4 * j)));
          sr_features(qi[0], qi[1], qi[2], qi[3], qj[0], qj[1], qj[2], qj[3], f);
        }
        if (KEEP != 0 && kq == 0) *reinterpret_cast<f32x4*>(fsl + jb * 64 + 4 * pp) = f32x4{f[0], f[1], f[2], 1.0f};
        sr_pair_mlp_lds(w1a, w2s + sr_here(wfo), bzs + sr_here(16 * kq), kq, f, l1, z[q]);
        const float* vrow = vs + sr_here(j * SR_LDV + 4 * kq);
        float sp0 = 0.f, sp1 = 0.f;
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(vrow + 16 * t), v1 = *reinterpret_cast<const f32x4*>(vrow + 16 * t + 16);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sp0 = fmaf(fmaxf(z[q][t][r], 0.f), v0[r], sp0);
            sp1 = fmaf(fmaxf(z[q][t + 1][r], 0.f), v1[r], sp1);
          }
        }
        const float sgv = quarters_sum(sp0 + sp1) + vrow[64 - 4 * kq];  // sigma_ij = l2_ij . v_j + c_j
        sg[jb] = !ok ? -INFINITY : (16 * jb + pp == i ? -1000.0f : sgv);  // (social.py:25: sigma_ii = -1000)
      }
      SR_T(6);  // 6: pass 1
      float mx = sg[0];
#pragma unroll
      for (int jb = 1; jb < NJB; ++jb) mx = fmaxf(mx, sg[jb]);
      mx = row_max16(mx);
      float at[NJB], den = 0.f;
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) {
        at[jb] = 16 * jb + pp < n ? __expf(sg[jb] - mx) : 0.f;
        den += at[jb];
      }
      const float inv = 1.0f / row_sum16(den);
      float dot = 0.f;
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) {
        at[jb] *= inv;
        dot = fmaf(at[jb], da[jb], dot);
      }
      dot = row_sum16(dot);
      SR_T(7);  // 7: softmax
      // pass 2: the adjoints
#pragma unroll
      for (int jbr = 0; jbr < NJB; ++jbr) {
        const int jb = KEEP == 2 ? NJB - 1 - jbr : jbr;  // (KEEP == 2: the block whose forward values are still there first)
        if (16 * jb >= n) continue;
        const int q = KEEP == 1 ? jb : 0;
        const int j = 16 * jb + pp < n ? 16 * jb + pp : 0;
        if (KEEP == 0 || (KEEP == 2 && jb != jbl)) {
          float f[3], l1[8];
          const f32x4 qi = *reinterpret_cast<const f32x4*>((pvs + sr_here(4 * i)));
          const f32x4 qj = *reinterpret_cast<const f32x4*>((pvs + sr_here(4 * j)));
          sr_features(qi[0], qi[1], qi[2], qi[3], qj[0], qj[1], qj[2], qj[3], f);
          sr_pair_mlp_lds(w1a, w2s + sr_here(wfo), bzs + sr_here(16 * kq), kq, f, l1, z[q]);
          if (TRAIN) {
            if (KEEP == 0 && kq == 0) *reinterpret_cast<f32x4*>(fsl + 4 * pp) = f32x4{f[0], f[1], f[2], 1.0f};
#pragma unroll
            for (int tp = 0; tp < 2; ++tp)
              *reinterpret_cast<f32x4*>(L1s + pp * SR_LD1 + 16 * tp + 4 * kq) =
                  f32x4{l1[4 * tp], l1[4 * tp + 1], l1[4 * tp + 2], l1[4 * tp + 3]};
          }
        } else if (TRAIN) {
          // (l1 again: two products; it goes straight to its tile -- d1's mask reads it back)
          const float fb = fss[sr_here(w * (NFS * 64) + (KEEP == 0 ? 0 : jb) * 64 + 4 * pp) + kq];
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          const f32x4 p0 = MFMA16(w1a[0], fb, zero), p1 = MFMA16(w1a[1], fb, zero);
          *reinterpret_cast<f32x4*>(L1s + pp * SR_LD1 + 4 * kq) =
              f32x4{fmaxf(p0[0], 0.f), fmaxf(p0[1], 0.f), fmaxf(p0[2], 0.f), fmaxf(p0[3], 0.f)};
          *reinterpret_cast<f32x4*>(L1s + pp * SR_LD1 + 16 + 4 * kq) =
              f32x4{fmaxf(p1[0], 0.f), fmaxf(p1[1], 0.f), fmaxf(p1[2], 0.f), fmaxf(p1[3], 0.f)};
        }
        const float daj = das[sr_here(i * DAL + 16 * jb) + pp];
        wave_lds_sync();
        if (kq == 0) das[i * DAL + 16 * jb + pp] = at[jb];  // a_ij for dh_j = sum_i a_ij dS_i (0 on padding lanes)
        // softmax adjoint (0 on padding lanes; sigma_ii is the constant -1000: nothing flows through it);
        // d[v_j | c_j] += dsigma_ij [l2_ij | 1]
        const float dsg = (16 * jb + pp == i) ? 0.f : at[jb] * (daj - dot);
        dc[jb] += dsg;
        const float* vrow = vs + sr_here(j * SR_LDV + 4 * kq);
        f32x4 dz2[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f32x4 v4 = *reinterpret_cast<const f32x4*>(vrow + 16 * t);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dv[jb][t][r] = fmaf(dsg, fmaxf(z[q][t][r], 0.f), dv[jb][t][r]);
            dz2[t][r] = z[q][t][r] > 0.f ? dsg * v4[r] : 0.f;
          }
          if (TRAIN) *reinterpret_cast<f32x4*>(Zs + pp * SR_LDZ + 16 * t + 4 * kq) = dz2[t];
        }
        SR_T(8);  // 8: sums, dz2
        if (TRAIN) {
          // dz1^T (32 units x 16 pairs) = W2^T dz2^T: B operand = registers in the D layout of the forward product
          f32x4 d1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
          const float* wt = W2T ? w2ts + sr_here(wto) : w2s + sr_here(wto);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (W2T) {
              const f32x4 w0 = *reinterpret_cast<const f32x4*>(wt + 16 * t), w1 = *reinterpret_cast<const f32x4*>(wt + 16 * SR_LD2T + 16 * t);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                d1[0] = MFMA16(w0[r], dz2[t][r], d1[0]);
                d1[1] = MFMA16(w1[r], dz2[t][r], d1[1]);
              }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                d1[0] = MFMA16(wt[(16 * t + r) * SR_LD2], dz2[t][r], d1[0]);
                d1[1] = MFMA16(wt[(16 * t + r) * SR_LD2 + 16], dz2[t][r], d1[1]);
              }
            }
          }
          SR_T(9);  // 9: dz1 products
          // d1[tp][r] = unit 16 tp + 4 kq + r = u(s = 4 tp + r): the units this lane wrote l1 for.
          // All three weight gradients contract over the 16 PAIRS: dz2, l1, dz1 and the features go through the wave's
          // LDS tiles into A / B fragments (pair = step + 4 k)
#pragma unroll
          for (int tp = 0; tp < 2; ++tp) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(L1s + pp * SR_LD1 + 16 * tp + 4 * kq);
            f32x4 g;
#pragma unroll
            for (int r = 0; r < 4; ++r) g[r] = l4[r] > 0.f ? d1[tp][r] : 0.f;
            *reinterpret_cast<f32x4*>(Gs + pp * SR_LD1 + 16 * tp + 4 * kq) = g;
          }
          wave_lds_sync();
#pragma unroll
          for (int sp = 0; sp < 4; ++sp) {  // k step sp: MFMA k index kq <-> pair sp + 4 kq
            const int pr = sp + 4 * kq;
            const float* zr = Zs + pr * SR_LDZ + pp;
            const float* lr = L1s + pr * SR_LD1 + pp;
            const float* gr = Gs + pr * SR_LD1 + pp;
            const float b0 = lr[0], b1v = lr[16];
            const f32x4 f4 = *reinterpret_cast<const f32x4*>(fsl + (KEEP == 0 ? 0 : jb) * 64 + 4 * pr);  // (f0, f1, f2, 1) of pair pr
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
              const float av = zr[16 * tm];
              db2[tm] += av;
              dW2[tm][0] = MFMA16(av, b0, dW2[tm][0]);
              dW2[tm][1] = MFMA16(av, b1v, dW2[tm][1]);
            }
            const float g0 = gr[0], g1 = gr[16];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              dW1[0][c] = fmaf(g0, f4[c], dW1[0][c]);
              dW1[1][c] = fmaf(g1, f4[c], dW1[1][c]);
            }
          }
          wave_lds_sync();  // the tiles are rewritten by the next block
          SR_T(10);  // 10: tiles + weight-gradient products
        }
      }
    }
    SR_T(11);  // 11: loop tail
    // (thread indices afresh for the exchanges and the epilogue: what the prologue derived from them would otherwise be carried through the row
    // loop, where no register is free -- in scratch)
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int w_e = tid_e >> 6, pp_e = tid_e & 15, kq_e = (tid_e >> 4) & 3;
    if (TRAIN) {
      // this scene's share of the workgroup's partial block: [64][33] = dW2 | db2, then [32][4] = dW1 | db1
      __syncthreads();  // every wave is done with its tiles (region A)
      float* mine = smem + w_e * SR_WG_FLOATS;
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * tm + 4 * kq_e + r;
          mine[m * 33 + pp_e] = dW2[tm][0][r];
          mine[m * 33 + 16 + pp_e] = dW2[tm][1][r];
        }
        const float sb = quarters_sum(db2[tm]);
        if (kq_e == 0) mine[(16 * tm + pp_e) * 33 + 32] = sb;
      }
#pragma unroll
      for (int tp = 0; tp < 2; ++tp)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float sw = quarters_sum(dW1[tp][c]);
          if (kq_e == 0) mine[64 * 33 + (16 * tp + pp_e) * 4 + c] = sw;
        }
      __syncthreads();
      float* out = a.partials + (size_t)blockIdx.x * SR_WG_FLOATS;
      for (int e = tid_e; e < SR_WG_FLOATS; e += NT) {
        float v = (smem[e] + smem[SR_WG_FLOATS + e]) + (smem[2 * SR_WG_FLOATS + e] + smem[3 * SR_WG_FLOATS + e]);
#pragma unroll
        for (int q = 4; q < NW; q += 4)
          v += (smem[q * SR_WG_FLOATS + e] + smem[(q + 1) * SR_WG_FLOATS + e]) +
               (smem[(q + 2) * SR_WG_FLOATS + e] + smem[(q + 3) * SR_WG_FLOATS + e]);
        out[e] = wrote ? out[e] + v : v;
      }
      wrote = true;
    }
    SR_T(14);  // 14: partial block
    // ---- the waves' neighbour sums meet in LDS (region A: every wave must be done with its tiles), fixed order, JR blocks per
    // round; the folded d[v | c] replace [v | c] in vs.  Beside the first round: the neighbour part of dh = A^T dS -> hs
    // (h is dead: DA was its last reader; every row's a_ij is in place after the barrier)
#pragma unroll
    for (int jb0 = 0; jb0 < NJB; jb0 += JR) {
      if (16 * jb0 >= n) break;
      __syncthreads();
#pragma unroll
      for (int q = 0; q < JR; ++q) {
        float* mine = red + ((w_e * JR + q) * 16 + pp_e) * SR_LDC;
#pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(mine + 16 * t + 4 * kq_e) = dv[jb0 + q][t];
        if (kq_e == 0) mine[64] = dc[jb0 + q];
      }
      if (jb0 == 0)
        for (int t = w_e; t < jt * (H >> 4); t += NW) {
          const int j0 = 16 * (t / (H >> 4)), k0 = 16 * (t % (H >> 4));
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          for (int S = 0; S < 16 * jt; S += 16)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              acc = MFMA16(das[(S + 4 * kq_e + r) * DAL + j0 + pp_e], dss[(S + 4 * kq_e + r) * LDH + k0 + pp_e], acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) hs[(j0 + 4 * kq_e + r) * LDH + k0 + pp_e] = acc[r];
        }
      __syncthreads();
      for (int e = tid_e; e < 16 * JR * 65; e += NT) {
        const int jl = e / 65, c = e - jl * 65, j = 16 * jb0 + jl;
        if (j >= n) break;
        const float* r0 = red + jl * SR_LDC + c;
        float v = (r0[0] + r0[16 * JR * SR_LDC]) + (r0[2 * 16 * JR * SR_LDC] + r0[3 * 16 * JR * SR_LDC]);
#pragma unroll
        for (int q = 4; q < NW; q += 4)
          v += (r0[q * 16 * JR * SR_LDC] + r0[(q + 1) * 16 * JR * SR_LDC]) +
               (r0[(q + 2) * 16 * JR * SR_LDC] + r0[(q + 3) * 16 * JR * SR_LDC]);
        vs[j * SR_LDV + c] = v;
      }
    }
    SR_T(12);  // 12: wave fold, A^T dS
    bool finish = true;
    if (RS > 1) {
      // this workgroup's share -> scratch; the last workgroup of the scene to arrive folds the RS shares in split order.
      // (agent-scope atomic stores: write-through, visible to the folding workgroup without a release fence --
      // MI355X_MICROARCH "valid forms")
      __syncthreads();
      for (int e = tid_e; e < n * (65 + H); e += NT) {
        const int j = e / (65 + H), c = e - j * (65 + H);
        __hip_atomic_store(a.scratch + ((size_t)rs * a.dvc_rows + s0 + j) * (65 + H) + c,
                           c < 65 ? vs[j * SR_LDV + c] : hs[j * LDH + (c - 65)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();  // (drains this workgroup's stores)
      if (tid_e == 0) {
        const unsigned t = __hip_atomic_fetch_add(a.tickets + sc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = t == (unsigned)(RS - 1);
        if (last_s) __hip_atomic_store(a.tickets + sc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      finish = last_s != 0;
      if (finish) {
        for (int e = tid_e; e < n * (65 + H); e += NT) {
          const int j = e / (65 + H), c = e - j * (65 + H);
          const float* src = a.scratch + (size_t)(s0 + j) * (65 + H) + c;
          float v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int q = 1; q < RS; ++q)
            v += __hip_atomic_load(src + (size_t)q * a.dvc_rows * (65 + H), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (c < 65) vs[j * SR_LDV + c] = v;
          else hs[j * LDH + (c - 65)] = v;
        }
      }
    }
    if (finish) {
      // adjoints of the dense stages, on chip (MFMA tiles): dWh_j = d[v_j | c_j] [W3 | b3]^T, dh_j = (neighbour part)
      // + dWh_j W_at
      __syncthreads();
      for (int e = tid_e; e < n * 65; e += NT) {
        const int j = e / 65, m = e - j * 65;
        a.dvc[(size_t)(s0 + j) * a.ldv + m] = vs[j * SR_LDV + m];  // operand of dW3 = Wh^T dv, db3 = Wh^T dc
      }
      for (int t = w_e; t < jt * (F >> 4); t += NW) {
        const int j0 = 16 * (t / (F >> 4)), f0 = 16 * (t % (F >> 4));
        f32x4 acc = sr_tile<true>(vs, SR_LDV, j0, w3_s, SR_LDW, f0, 64, f32x4{0.f, 0.f, 0.f, 0.f}, pp_e, kq_e);
        const float b3f = b3_s[f0 + pp_e];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = j0 + 4 * kq_e + r;
          acc[r] = fmaf(vs[j * SR_LDV + 64], b3f, acc[r]);
          whs[j * SR_LDW + f0 + pp_e] = acc[r];
          if (j < n) a.dWh_out[(size_t)(s0 + j) * F + f0 + pp_e] = acc[r];
        }
      }
      __syncthreads();
      for (int t = w_e; t < jt * (H >> 4); t += NW) {
        const int j0 = 16 * (t / (H >> 4)), k0 = 16 * (t % (H >> 4));
        const f32x4 acc = sr_tile<false>(whs, SR_LDW, j0, wat_s, H + 4, k0, F, f32x4{0.f, 0.f, 0.f, 0.f}, pp_e, kq_e);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = j0 + 4 * kq_e + r;
          if (j < n) {
            const float v = acc[r] + hs[j * LDH + k0 + pp_e];
            float* d = a.dh + (size_t)(s0 + j) * a.ld_dh + k0 + pp_e;
            *d = a.accumulate_dh ? *d + v : v;
          }
        }
      }
    }
  }

  SR_T(13);  // 13: dense adjoints
  if (TRAIN && !wrote)  // (a workgroup without a scene of two or more pedestrians)
    for (int e = threadIdx.x; e < SR_WG_FLOATS; e += NT) a.partials[(size_t)blockIdx.x * SR_WG_FLOATS + e] = 0.f;
#ifdef SR_PROFILE
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.scratch)
    for (int k = 0; k < 16; ++k) a.scratch[k] = (float)prof[k];
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// persistent: one workgroup per CU walks units (scene, row split) un, un + grid, ...
static int sr_grid(int S, int RS) { return S * RS < 256 ? S * RS : 256; }
// few scenes: the rows of a scene are dealt to two workgroups.  (Measured at 64 scenes x 20 pedestrians: the backward launch
// 41 / 35 / 42 us with 1 / 2 / 4 splits -- a workgroup's fixed costs (weight fragments, staging the scene, the exchanges at
// its end) outweigh the rows from four on; the knob MGGAN_SOC_SPLITS forces a value.)
// waves per workgroup: eight (two per SIMD); four where the scene's staged rows leave no room for eight waves' tiles
#define SR_NW(NN) ((NN) == 4 ? 4 : 8)
static int sr_njb(int max_n) { return max_n <= 16 ? 1 : max_n <= 32 ? 2 : 4; }
static int sr_splits(int S, int max_n) {
  const int nw = SR_NW(sr_njb(max_n));
  static const char* force = getenv("MGGAN_SOC_SPLITS");  // measurement knob: 1, 2 or 4
  if (force && atoi(force) >= 1 && atoi(force) <= 4 && nw * atoi(force) <= (max_n > nw ? max_n : nw)) return atoi(force);
  // (inside the iteration graph, beside the branch streams' kernels, the unsplit launch wins at 64 scenes: 1.478 vs
  //  1.489 ms per iteration over three alternating pairs -- no ticket fold, half the workgroups competing for CUs; the split
  //  is kept for really few scenes)
  int rs = 1;
  while (rs < 2 && S * rs * 2 <= 64 && nw * rs * 2 <= max_n) rs *= 2;
  return rs;
}

extern "C" {

int mggan_social_rows_splits(int S, int max_n) { return sr_splits(S, max_n); }
int mggan_social_rows_grid(int S, int max_n) { return sr_grid(S, sr_splits(S, max_n)); }
int mggan_social_rows_partial_floats(void) { return SR_WG_FLOATS; }

#ifndef SR_NWF
#define SR_NWF 8
#endif
#define SR_NWFW(NN) ((NN) == 4 ? 4 : SR_NWF)
#define SR_FWD(HH, NN) \
  MG_LAUNCH((social_rows_fwd_kernel<HH, NN, SR_NWFW(NN)>), dim3(sr_grid(S, a.row_splits)), dim3(64 * SR_NWFW(NN)), 0, stream, a)
#ifndef SR_KEEP2
#define SR_KEEP2 1
#endif
#ifndef SR_NW2
#define SR_NW2 8
#endif
#define SR_NWB(NN, TT) ((TT) && (NN) == 2 ? SR_NW2 : SR_NW(NN))
#define SR_BWD(HH, NN, TT, KK)                                                                                     \
  MG_LAUNCH((social_rows_bwd_kernel<HH, NN, SR_NWB(NN, TT), TT, KK>), dim3(sr_grid(S, a.row_splits)),     \
                     dim3(64 * SR_NWB(NN, TT)), 0, stream, a)

int mggan_social_rows_fwd(int S, const int* scenes, int H, int F, int max_n, const float* xy_last, const float* dxdy_last,
                          int xy_mod, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                          const float* b3, const float* Wat, const float* bat, const float* h, int ld_h, float* Sout,
                          int ld_s, hipStream_t stream) {
  MG_CHECK_ARG(H == 32 || H == 64, "social_rows_fwd: hidden size %d not built (32 or 64)", H);
  MG_CHECK_ARG(S >= 0 && max_n >= 0 && max_n <= 64 && xy_mod >= 0 && F >= 16 && F <= 64 && F % 16 == 0,
               "social_rows_fwd: bad sizes (S %d, max_n %d, F %d: 16, 32, 48 or 64)", S, max_n, F);
  if (S == 0) return MGGAN_OK;
  MG_CHECK_ARG(scenes && xy_last && dxdy_last && W1 && b1 && W2 && b2 && W3 && b3 && Wat && bat && h && Sout,
               "social_rows_fwd: null pointer");
  MG_CHECK_ARG(ld_h % 4 == 0 && ld_s % 4 == 0 && ((size_t)h % 16) == 0 && ((size_t)Sout % 16) == 0,
               "social_rows_fwd: h / S rows must be 16-byte aligned (ld_h %d, ld_s %d)", ld_h, ld_s);
  MG_CHECK_ARG(((size_t)W3 % 16) == 0 && ((size_t)Wat % 16) == 0, "social_rows_fwd: W3 / W_at must be 16-byte aligned");
  SocRowsArgs a = {};
  a.scenes = scenes; a.xy = xy_last; a.dxy = dxdy_last; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.W3 = W3; a.b3 = b3;
  a.Wat = Wat; a.bat = bat; a.F = F; a.h = h; a.Sout = Sout; a.S = S; a.xy_mod = xy_mod; a.ld_h = ld_h; a.ld_s = ld_s;
  a.row_splits = sr_splits(S, max_n);
  const int njb = sr_njb(max_n);
  if (H == 32) {
    if (njb == 1) SR_FWD(32, 1); else if (njb == 2) SR_FWD(32, 2); else SR_FWD(32, 4);
  } else {
    if (njb == 1) SR_FWD(64, 1); else if (njb == 2) SR_FWD(64, 2); else SR_FWD(64, 4);
  }
  MG_LAUNCH_CHECK("social_rows_fwd");
  return MGGAN_OK;
}

int mggan_social_rows_bwd(int S, const int* scenes, int H, int F, int max_n, const float* xy_last, const float* dxdy_last,
                          int xy_mod, const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                          const float* b3, const float* Wat, const float* bat, const float* h, int ld_h, const float* dS,
                          int ld_ds, float* dvc, int ldv, int dvc_rows, float* Wh, float* dWh, float* dh, int ld_dh,
                          int accumulate_dh, float* partials, float* scratch, unsigned* tickets, hipStream_t stream) {
  MG_CHECK_ARG(H == 32 || H == 64, "social_rows_bwd: hidden size %d not built (32 or 64)", H);
  MG_CHECK_ARG(S >= 0 && max_n >= 0 && max_n <= 64 && xy_mod >= 0 && F >= 16 && F <= 64 && F % 16 == 0,
               "social_rows_bwd: bad sizes (S %d, max_n %d, F %d: 16, 32, 48 or 64)", S, max_n, F);
  if (S == 0) return MGGAN_OK;
  MG_CHECK_ARG(scenes && xy_last && dxdy_last && W1 && b1 && W2 && b2 && W3 && b3 && Wat && bat && h && dS && dvc && Wh &&
                   dWh && dh, "social_rows_bwd: null pointer");
  MG_CHECK_ARG(ldv >= 65 && ld_h % 4 == 0 && ld_ds % 4 == 0 && ((size_t)h % 16) == 0 && ((size_t)dS % 16) == 0,
               "social_rows_bwd: h / dS rows must be 16-byte aligned (ld_h %d, ld_ds %d); ldv %d >= 65", ld_h, ld_ds, ldv);
  MG_CHECK_ARG(((size_t)W2 % 16) == 0 && ((size_t)W3 % 16) == 0 && ((size_t)Wat % 16) == 0,
               "social_rows_bwd: W2 / W3 / W_at must be 16-byte aligned");
  SocRowsArgs a = {};
  a.scenes = scenes; a.xy = xy_last; a.dxy = dxdy_last; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.W3 = W3; a.b3 = b3;
  a.Wat = Wat; a.bat = bat; a.F = F; a.h = h; a.Wh_out = Wh; a.dWh_out = dWh;
  a.dS = dS; a.dvc = dvc; a.dh = dh; a.partials = partials; a.S = S; a.xy_mod = xy_mod; a.ldv = ldv; a.ld_h = ld_h;
  a.ld_ds = ld_ds; a.ld_dh = ld_dh; a.accumulate_dh = accumulate_dh;
  a.row_splits = sr_splits(S, max_n); a.dvc_rows = dvc_rows; a.scratch = scratch; a.tickets = tickets;
  MG_CHECK_ARG(a.row_splits == 1 || (scratch && tickets && dvc_rows > 0),
               "social_rows_bwd: %d row splits need scratch (splits x dvc_rows x (65 + H) floats) and tickets (S words, zero)",
               a.row_splits);
  const int njb = sr_njb(max_n);
  const bool train = partials != nullptr;
  // (four blocks per row: the forward values of a row are recomputed in the adjoint pass instead of kept)
#define SR_BWD_N(HH, TT) \
  do { if (njb == 1) SR_BWD(HH, 1, TT, 1); else if (njb == 2) SR_BWD(HH, 2, TT, (TT ? SR_KEEP2 : 1)); else SR_BWD(HH, 4, TT, 0); } while (0)
  if (H == 32) {
    if (train) SR_BWD_N(32, true); else SR_BWD_N(32, false);
  } else {
    if (train) SR_BWD_N(64, true); else SR_BWD_N(64, false);
  }
  MG_LAUNCH_CHECK("social_rows_bwd");
  return MGGAN_OK;
}

}  // extern "C"
