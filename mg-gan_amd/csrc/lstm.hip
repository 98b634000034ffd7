// Time-step-fused LSTM rollouts for MG-GAN on gfx950.
//
// Replaces, on the reference hot path (file:line under /root/reference/mggan):
//   TrajectoryEncoder.forward     model/modules/common_modules.py:48-66  (Linear(2,E) + nn.LSTM over T=7)
//   RelativeDecoder.forward       model/modules/common_modules.py:97-131 (12 x {Linear(2,E); LSTM cell; hidden2pos; xy+=dxdy})
//   enc_h_to_dec_h                model/modules/standard.py:91-94,244-252 (h0 = Linear(136,32)([enc_h|noise]), c0 = 0)
//   selected-rollout gather       model/modules/standard.py:190-214 (folded: one row per selected (ped, sample))
//
// Design (CDNA4): both rollouts are matrix-core kernels (lstm_fwd_mfma_kernel, decoder_fwd/bwd_mfma_kernel): a
// workgroup of four waves owns a tile of 16 rows for all T steps; the gate rows of W_hh are MFMA A-fragments held in
// VGPRs for the whole sequence (read from L2 once per workgroup, never per step -- at these sizes registers beat an
// LDS copy: 64 KB of W_hh at H = 64 is exactly 64 fragment registers per lane), h_t is exchanged through a 2-4 KB LDS
// tile with ONE LDS-only barrier per step, c_t stays in registers, gate rows are permuted so that the cell update is
// lane-local, and (decoder) the hidden2pos head + autoregressive feedback run inside the same launch.
// The input embedding Linear(2,E) is folded algebraically into the gate weights (A = W_ih W_emb, 4H x 2) by a tiny
// prep kernel; its gradient is un-folded by the chain rule in mggan_lstm_unfold_grads.  The encoder's adjoint
// (lstm_bwd_kernel) is a VALU kernel, one lane per (row, unit); the lane-per-unit VALU forward is kept for A/B
// measurements (MGGAN_LSTM_VALU=1).
// Rows are pre-bucketed by generator so that a workgroup's lanes load one generator's weights (L1-resident) and the
// per-generator weight gradients see contiguous segments.
#include <stdlib.h>
#include "common.h"
#include "../../include/mggan_hip.h"

// ---- layout of one prepared (folded/transposed) weight block, in floats -----------
//  A[4H][2] | bias[4H] | WhhT[H][4H] | (decoder only) W1T[H+S][H/2] | b1[H/2] | W2[2][H/2] | b2[2]
__host__ __device__ inline int prep_off_A(int H) { return 0; }
__host__ __device__ inline int prep_off_bias(int H) { return 8 * H; }
__host__ __device__ inline int prep_off_whhT(int H) { return 12 * H; }
__host__ __device__ inline int prep_off_w1T(int H) { return 12 * H + 4 * H * H; }
__host__ __device__ inline int prep_size(int H, int S, int dec) {
  int n = 12 * H + 4 * H * H;
  if (dec) n += (H + S) * (H / 2) + H / 2 + 2 * (H / 2) + 2;
  return (n + 3) / 4 * 4;
}

struct FoldArgs {
  const float *W_emb, *b_emb, *W_ih, *b_ih, *b_hh, *W_hh, *W1, *b1, *W2, *b2;
  long param_stride;  // floats between consecutive generators' parameter blocks
  float* prep;
  int prep_stride, H, E, S, dec;
};

__global__ __launch_bounds__(256) void lstm_fold_kernel(FoldArgs a) {
  const int grp = blockIdx.x;
  const long po = (long)grp * a.param_stride;
  const int H = a.H, E = a.E, G4 = 4 * a.H;
  const int t0 = blockIdx.y * blockDim.x + threadIdx.x, nt = gridDim.y * blockDim.x;
  float* P = a.prep + (size_t)grp * a.prep_stride;
  // A = W_ih W_emb, bias = b_ih + b_hh + W_ih b_emb: 16 lanes per gate row, each takes every 16th embedding
  // column (contiguous reads of the W_ih row) and a shuffle tree adds them -- a lane per row walking E columns
  // alone was a chain of E dependent loads (18 us for the 256 x 64 discriminator encoder)
  {
    const int sub = t0 & 15;
    for (int m = t0 >> 4; m < G4; m += nt >> 4) {
      float s0 = 0.f, s1 = 0.f, sb = 0.f;
      for (int e = sub; e < E; e += 16) {
        const float w = a.W_ih[po + (size_t)m * E + e];
        s0 = fmaf(w, a.W_emb[po + e * 2 + 0], s0);
        s1 = fmaf(w, a.W_emb[po + e * 2 + 1], s1);
        sb = fmaf(w, a.b_emb[po + e], sb);
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        s0 += __shfl_xor(s0, o, 64);
        s1 += __shfl_xor(s1, o, 64);
        sb += __shfl_xor(sb, o, 64);
      }
      if (sub == 0) {
        P[prep_off_A(H) + m * 2 + 0] = s0;
        P[prep_off_A(H) + m * 2 + 1] = s1;
        P[prep_off_bias(H) + m] = sb + (a.b_ih[po + m] + a.b_hh[po + m]);
      }
    }
  }
  for (int i = t0; i < H * G4; i += nt) {
    int k = i / G4, m = i % G4;
    P[prep_off_whhT(H) + i] = a.W_hh[po + (size_t)m * H + k];
  }
  if (a.dec) {
    const int Hh = H / 2, IN = H + a.S;
    float* w1T = P + prep_off_w1T(H);
    for (int i = t0; i < IN * Hh; i += nt) {
      int k = i / Hh, m = i % Hh;
      w1T[i] = a.W1[po + (size_t)m * IN + k];
    }
    float* b1 = w1T + IN * Hh;
    float* w2 = b1 + Hh;
    float* b2 = w2 + 2 * Hh;
    for (int i = t0; i < Hh; i += nt) b1[i] = a.b1[po + i];
    for (int i = t0; i < 2 * Hh; i += nt) w2[i] = a.W2[po + i];
    if (t0 < 2) b2[t0] = a.b2[po + t0];
  }
}

struct UnfoldArgs {
  const float *W_emb, *b_emb, *W_ih;       // parameters (group 0)
  float *dW_emb, *db_emb, *dW_ih, *db_ih, *db_hh;  // gradients (group 0), accumulated
  long param_stride;
  const float* dprep;  // per group: dA[4H][2] | dbias[4H]
  int dprep_stride, H, E;
};

__global__ __launch_bounds__(256) void lstm_unfold_kernel(UnfoldArgs a) {
  const int grp = blockIdx.x;
  const long po = (long)grp * a.param_stride;
  const int E = a.E, G4 = 4 * a.H;
  const int t0 = blockIdx.y * blockDim.x + threadIdx.x, nt = gridDim.y * blockDim.x;
  const float* dA = a.dprep + (size_t)grp * a.dprep_stride;
  const float* dB = dA + 2 * G4;
  for (int i = t0; i < G4 * E; i += nt) {
    int m = i / E, e = i % E;
    float v = dA[m * 2] * a.W_emb[po + e * 2] + dA[m * 2 + 1] * a.W_emb[po + e * 2 + 1] + dB[m] * a.b_emb[po + e];
    a.dW_ih[po + i] += v;
  }
  for (int m = t0; m < G4; m += nt) {
    a.db_ih[po + m] += dB[m];
    a.db_hh[po + m] += dB[m];
  }
  // dW_emb[e][c] = sum_m W_ih[m][e] dA[m][c], db_emb[e] = sum_m W_ih[m][e] dB[m]: one wave per (e, c|bias)
  const int wave = (blockIdx.y * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.y * blockDim.x) >> 6;
  const int lane = threadIdx.x & 63;
  for (int o = wave; o < 3 * E; o += nw) {
    const int e = o / 3, c = o % 3;
    float s = 0.f;
    for (int m = lane; m < G4; m += 64) s = fmaf(a.W_ih[po + (size_t)m * E + e], c < 2 ? dA[m * 2 + c] : dB[m], s);
    s = wave_sum(s);
    if (lane == 0) {
      if (c < 2) a.dW_emb[po + e * 2 + c] += s;
      else a.db_emb[po + e] += s;
    }
  }
}

// ------------------------------------------------------------------------------------
// Trajectory encoder (T = 7 observed steps, row r == pedestrian r).  One lane per (row, hidden unit); the
// four gate rows of W_hh for that unit live in VGPRs for the whole sequence, h_t is exchanged through LDS
// (one row per trajectory, broadcast ds_read_b128), c_t stays in a register.
struct SeqArgs {
  int R, T, b;
  const float* prep;
  const float* x;  // (T,b,2) time-major
  float* hout;     // (R, ld_hout)
  int ld_hout;
  float *Gt, *Cs, *Hp, *Din;  // saved for backward (all NULL in no-grad mode)
};

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// Instruction-issue bound, so the step loop is written for few instructions and <= 256 registers (two waves
// per SIMD): the four gates are two packed pairs (i,f) / (g,o), one v_pk_fma_f32 does two gate MACs per lane.
template <int H>
__global__ __launch_bounds__(256) void lstm_fwd_kernel(SeqArgs p) {
  constexpr int RT = 256 / H, G4 = 4 * H;
  __shared__ __attribute__((aligned(16))) float hbuf[RT][H];
  const int rr = threadIdx.x / H, j = threadIdx.x % H;
  const int r = blockIdx.x * RT + rr;
  const bool valid = r < p.R;
  const int rc = valid ? r : p.R - 1;
  const float* P = p.prep;
  const float* WT = P + prep_off_whhT(H);
  const bool save = p.Gt != nullptr;

  v2f wp[2][H], ap0[2], ap1[2], bp[2];  // pair 0 = gates (i,f), pair 1 = gates (g,o)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int k = 0; k < H; ++k) wp[q][k] = v2f{WT[k * G4 + (2 * q) * H + j], WT[k * G4 + (2 * q + 1) * H + j]};
    ap0[q] = v2f{P[prep_off_A(H) + ((2 * q) * H + j) * 2], P[prep_off_A(H) + ((2 * q + 1) * H + j) * 2]};
    ap1[q] = v2f{P[prep_off_A(H) + ((2 * q) * H + j) * 2 + 1], P[prep_off_A(H) + ((2 * q + 1) * H + j) * 2 + 1]};
    bp[q] = v2f{P[prep_off_bias(H) + (2 * q) * H + j], P[prep_off_bias(H) + (2 * q + 1) * H + j]};
  }
  float hj = 0.f, c = 0.f;
  float hv[H];
#pragma unroll
  for (int k = 0; k < H; ++k) hv[k] = 0.f;

  for (int t = 0; t < p.T; ++t) {
    const size_t rt = (size_t)r * p.T + t;
    const float d0 = p.x[((size_t)t * p.b + rc) * 2], d1 = p.x[((size_t)t * p.b + rc) * 2 + 1];
    if (save && valid) {
      p.Hp[rt * H + j] = hj;
      if (j == 0) { p.Din[rt * 2] = d0; p.Din[rt * 2 + 1] = d1; }
    }
    v2f acc[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      acc[q][0] = pk_fma(ap0[q], v2f{d0, d0}, bp[q]);
      acc[q][1] = ap1[q] * v2f{d1, d1};
    }
#pragma unroll
    for (int k = 0; k < H; k += 2) {
      const v2f ha = v2f{hv[k], hv[k]}, hb = v2f{hv[k + 1], hv[k + 1]};
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        acc[q][0] = pk_fma(wp[q][k], ha, acc[q][0]);
        acc[q][1] = pk_fma(wp[q][k + 1], hb, acc[q][1]);
      }
    }
    const v2f pif = acc[0][0] + acc[0][1], pgo = acc[1][0] + acc[1][1];
    const float gi = mg_sigmoid(pif.x), gf = mg_sigmoid(pif.y), gg = mg_tanh(pgo.x), go = mg_sigmoid(pgo.y);
    c = fmaf(gf, c, gi * gg);
    hj = go * mg_tanh(c);
    if (save && valid) {
      p.Gt[rt * G4 + 0 * H + j] = gi;
      p.Gt[rt * G4 + 1 * H + j] = gf;
      p.Gt[rt * G4 + 2 * H + j] = gg;
      p.Gt[rt * G4 + 3 * H + j] = go;
      p.Cs[rt * H + j] = c;
    }
    lds_barrier();  // every lane of the row has consumed the previous h
    hbuf[rr][j] = hj;
    lds_barrier();
#pragma unroll
    for (int k = 0; k < H; k += 4) {
      float4 t4 = *reinterpret_cast<const float4*>(&hbuf[rr][k]);
      hv[k] = t4.x; hv[k + 1] = t4.y; hv[k + 2] = t4.z; hv[k + 3] = t4.w;
    }
  }
  if (valid) p.hout[(size_t)r * p.ld_hout + j] = hj;
}

// Matrix-core form of the trajectory encoder (same saved-tensor layout as lstm_fwd_kernel, which it replaces on
// the hot path).  A workgroup of four waves owns a tile of 16 trajectories for all T steps; per step ONE matrix phase
//     G = W_hh . h_{t-1}^T     (A = gate rows of W_hh held in registers for the whole sequence, B = the 16 x H tile of
//                               h_{t-1} read from LDS, v_mfma_f32_16x16x4_f32: exact f32)
// Wave w owns hidden units w*H/4 .. (w+1)*H/4 - 1; the rows of its M-tiles are permuted so that D register r of lane
// (fi = lane & 15, fk = lane >> 4) of tile mt is gate r (i, f, g, o) of unit w*H/4 + 4*mt + fk for trajectory fi: the
// cell update is lane-local, the input (dx, dy) enters as two FMAs per gate through the folded A = W_ih W_emb.
// The lane-per-(row, unit) kernel above needs 416 VGPRs at H = 64 (one wave per SIMD, every lane walks all of W_hh's
// column on the VALU: 7 % of the f32 peak); here the 64 x 64 recurrent product of a step is 64 MFMAs per wave.
template <int H>
__global__ __launch_bounds__(256) void lstm_fwd_mfma_kernel(SeqArgs p) {
  constexpr int G4 = 4 * H, U = H / 4, MT = U / 4, KS = H / 4, HLDS = H + 4;
  __shared__ __attribute__((aligned(16))) float hs[2][16 * HLDS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  const int r = blockIdx.x * 16 + fi;
  const bool valid = r < p.R;
  const int rc = valid ? r : p.R - 1;
  const float* P = p.prep;
  const float* WT = P + prep_off_whhT(H);
  const bool sv = p.Gt != nullptr && valid;
  // A operand of k-step ks: lane (fi, fk) supplies M row fi (unit w*U + 4*mt + (fi >> 2), gate fi & 3), K index KS*fk + ks
  float Wg[MT][KS];
  int uj[MT];
  float ca0[MT][4], ca1[MT][4], cb[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) Wg[mt][ks] = WT[(KS * fk + ks) * G4 + (fi & 3) * H + w * U + 4 * mt + (fi >> 2)];
    uj[mt] = w * U + 4 * mt + fk;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = q * H + uj[mt];
      ca0[mt][q] = P[prep_off_A(H) + m * 2];
      ca1[mt][q] = P[prep_off_A(H) + m * 2 + 1];
      cb[mt][q] = P[prep_off_bias(H) + m];
    }
  }
  float c[MT], hprev[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) c[mt] = hprev[mt] = 0.f;
  // (the step's input is asked for one step ahead: fetched at the head of its step it was a global round trip per step,
  // a quarter of the launch at 1,280 rows)
  float2 dn = *reinterpret_cast<const float2*>(p.x + (size_t)rc * 2);
  for (int t = 0; t < p.T; ++t) {
    const size_t rt = (size_t)r * p.T + t;
    const float2 d = dn;
    dn = *reinterpret_cast<const float2*>(p.x + ((size_t)(t + 1 < p.T ? t + 1 : t) * p.b + rc) * 2);
    f32x4 G[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) G[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t > 0) {  // h_0 = 0: nothing to multiply at the first step
      const float* hb = &hs[t & 1][fi * HLDS + KS * fk];
      float hB[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ks += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(hb + ks);
        hB[ks] = v[0]; hB[ks + 1] = v[1]; hB[ks + 2] = v[2]; hB[ks + 3] = v[3];
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) G[mt] = MFMA16(Wg[mt][ks], hB[ks], G[mt]);
    }
    float* hw = hs[(t + 1) & 1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 g;
#pragma unroll
      for (int q = 0; q < 4; ++q) g[q] = fmaf(ca1[mt][q], d.y, fmaf(ca0[mt][q], d.x, G[mt][q] + cb[mt][q]));
      g[0] = mg_sigmoid(g[0]); g[1] = mg_sigmoid(g[1]); g[2] = mg_tanh(g[2]); g[3] = mg_sigmoid(g[3]);
      c[mt] = fmaf(g[1], c[mt], g[0] * g[2]);
      const float hn = g[3] * mg_tanh(c[mt]);
      hw[fi * HLDS + uj[mt]] = hn;
      if (sv) {
        p.Hp[rt * H + uj[mt]] = hprev[mt];
#pragma unroll
        for (int q = 0; q < 4; ++q) p.Gt[rt * G4 + q * H + uj[mt]] = g[q];
        p.Cs[rt * H + uj[mt]] = c[mt];
      }
      hprev[mt] = hn;
    }
    if (sv && w == 0 && fk == 0) *reinterpret_cast<float2*>(p.Din + rt * 2) = d;
    lds_barrier();
  }
  if (valid) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) p.hout[(size_t)r * p.ld_hout + uj[mt]] = hprev[mt];
  }
}

struct SeqBwdArgs {
  int R, T;
  const float* W_hh;
  const float *Gt, *Cs;
  const float* dhT;  // gradient of h_T
  int ld_dhT;
  float* dPre;       // (R,T,4H) gate pre-activation gradients (weight gradients follow as GEMMs)
};

// LW = number of W_hh rows (m) kept in LDS instead of registers.  H = 64 with everything in registers needs 288 VGPRs:
// one wave per SIMD, one workgroup per CU, and the 320 workgroups of 1,280 rows take two rounds on 256 CUs; with a
// quarter of the column in LDS (16 KB, shared by the workgroup's rows: 236 VGPRs) two workgroups fit a CU and one
// round does it: 2.241 vs 2.267 ms/iter (128 rows in LDS: 2.262).  The forward kernel (416 VGPRs) was given the
// same treatment and showed no difference: it is not on the critical chain.
template <int H, int LW>
__global__ __launch_bounds__(256) void lstm_bwd_kernel(SeqBwdArgs p) {
  constexpr int RT = 256 / H, G4 = 4 * H, RW = G4 - LW;
  __shared__ __attribute__((aligned(16))) float dpbuf[RT][G4];
  __shared__ float wl[LW > 0 ? LW : 1][H];
  const int rr = threadIdx.x / H, j = threadIdx.x % H;
  const int r = blockIdx.x * RT + rr;
  const bool valid = r < p.R;
  const int rc = valid ? r : p.R - 1;

  v2f whc[RW / 2];  // column j of W_hh in packed pairs: dh_prev[j] = sum_m W_hh[m][j] dpre[m]
#pragma unroll
  for (int m = 0; m < RW; m += 2) whc[m / 2] = v2f{p.W_hh[(size_t)m * H + j], p.W_hh[(size_t)(m + 1) * H + j]};
  if (LW > 0) {
    for (int i = threadIdx.x; i < LW * H; i += 256) wl[i / H][i % H] = p.W_hh[(size_t)(RW + i / H) * H + (i % H)];
    __syncthreads();
  }
  float dh = p.dhT[(size_t)rc * p.ld_dhT + j], dc = 0.f;

  // software pipeline over time: the saved activations of step t-1 are fetched while step t is computed
  float n_gi, n_gf, n_gg, n_go, n_cc, n_cp;
  auto fetch = [&](int t) {
    const size_t rt = (size_t)rc * p.T + t;
    n_gi = p.Gt[rt * G4 + j]; n_gf = p.Gt[rt * G4 + H + j]; n_gg = p.Gt[rt * G4 + 2 * H + j];
    n_go = p.Gt[rt * G4 + 3 * H + j];
    n_cc = p.Cs[rt * H + j];
    n_cp = t > 0 ? p.Cs[(rt - 1) * H + j] : 0.f;
  };
  fetch(p.T - 1);
  for (int t = p.T - 1; t >= 0; --t) {
    const size_t rt = (size_t)rc * p.T + t;
    const float gi = n_gi, gf = n_gf, gg = n_gg, go = n_go, cc = n_cc, cprev = n_cp;
    if (t > 0) fetch(t - 1);
    const float tc = mg_tanh(cc);
    const float dO = dh * tc;
    dc = fmaf(dh * go, 1.f - tc * tc, dc);
    const float dpi = dc * gg * gi * (1.f - gi);
    const float dpf = dc * cprev * gf * (1.f - gf);
    const float dpg = dc * gi * (1.f - gg * gg);
    const float dpo = dO * go * (1.f - go);
    dc = dc * gf;
    if (valid) {
      p.dPre[rt * G4 + j] = dpi;
      p.dPre[rt * G4 + H + j] = dpf;
      p.dPre[rt * G4 + 2 * H + j] = dpg;
      p.dPre[rt * G4 + 3 * H + j] = dpo;
    }
    dpbuf[rr][j] = dpi;
    dpbuf[rr][H + j] = dpf;
    dpbuf[rr][2 * H + j] = dpg;
    dpbuf[rr][3 * H + j] = dpo;
    lds_barrier();
    v2f nh0 = v2f{0.f, 0.f}, nh1 = v2f{0.f, 0.f};  // packed FMAs: two MACs per instruction
#pragma unroll
    for (int m = 0; m < RW; m += 4) {
      const float4 v = *reinterpret_cast<const float4*>(&dpbuf[rr][m]);
      nh0 = pk_fma(whc[m / 2], v2f{v.x, v.y}, nh0);
      nh1 = pk_fma(whc[m / 2 + 1], v2f{v.z, v.w}, nh1);
    }
#pragma unroll
    for (int m = 0; m < LW; m += 4) {  // the rows of the column that live in LDS (conflict-free: consecutive j)
      const float4 v = *reinterpret_cast<const float4*>(&dpbuf[rr][RW + m]);
      nh0 = pk_fma(v2f{wl[m][j], wl[m + 1][j]}, v2f{v.x, v.y}, nh0);
      nh1 = pk_fma(v2f{wl[m + 2][j], wl[m + 3][j]}, v2f{v.z, v.w}, nh1);
    }
    dh = (nh0.x + nh0.y) + (nh1.x + nh1.y);
    lds_barrier();
  }
}

// Matrix-core form of the encoder's BPTT (same arguments and the same dPre layout as lstm_bwd_kernel, which it replaces
// on the hot path).  A wave owns 16 trajectories x 16 hidden units for all T steps: H = 64 -> four unit blocks of one
// 16-row tile per workgroup, H = 32 -> two unit blocks of two tiles.  Per step
//   1. gate gradients dPre of the lane's own four units (VALU; lane (fi, fk) owns units 16 ub + 4 fk + r of row fi,
//      which is exactly where the D registers of step 2 put dh_{t-1}: the state never leaves the lane)
//      -> global (R,T,4H) for the weight-gradient GEMM and an LDS tile [row][unit*4 + gate], one 16-byte store per unit
//   2. barrier; dh_{t-1}^T [units x rows] = W_hh^T [units x 4H] . dPre^T [4H x rows]: A = this wave's 16 columns of W_hh
//      held in registers for the whole sequence (K index in tile-position order), B = the dPre tile read back as
//      16-byte rows: 4H / 4 MFMAs per wave and step (exact f32).
// The lane-per-(row, unit) kernel walked its W_hh column on the VALU (2.4 % of the f32 peak; 209 us for the
// discriminator's 8,192 x 64 encoder at configs[2], on the discriminator step's critical chain).
template <int H>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void lstm_bwd_mfma_kernel(SeqBwdArgs p) {
  constexpr int G4 = 4 * H, UB = H / 16, RTILES = 4 / UB, KS = H;  // KS = 4H / 4 MFMA k-steps
  constexpr int LD = G4 + 20;  // dPre tile row stride: == 20 mod 64 (H = 64: 276, H = 32: 148)
  __shared__ __attribute__((aligned(16))) float dps[2][RTILES][16 * LD];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  const int ub = w % UB, rtile = w / UB;
  const int r = (blockIdx.x * RTILES + rtile) * 16 + fi;
  const bool valid = r < p.R;
  const int rc = valid ? r : p.R - 1;
  const int u0 = 16 * ub + 4 * fk;  // this lane's four units u0 .. u0 + 3 (as outputs of the matrix phase and in the gates)
  // A operand: M row fi = unit 16 ub + fi; k-step ks = 4 S + i covers tile position pp = 16 S + 4 fk + i
  // (position = unit*4 + gate <-> gate row (pp & 3) * H + (pp >> 2) of W_hh)
  float Ah[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int pp = 16 * (ks >> 2) + 4 * fk + (ks & 3), m = (pp & 3) * H + (pp >> 2);
    Ah[ks] = p.W_hh[(size_t)m * H + 16 * ub + fi];
  }
  f32x4 dh = *reinterpret_cast<const f32x4*>(p.dhT + (size_t)rc * p.ld_dhT + u0);
  f32x4 dc = f32x4{0.f, 0.f, 0.f, 0.f};
  const float vm = valid ? 1.f : 0.f;
  // saved activations of the next step to process, one step ahead (address selected, value masked at use: a select on a
  // loaded value would make the compiler wait for the loads on the spot)
  f32x4 n_g[4], n_c, n_cp;
  auto fetch = [&](int t) {
    const size_t rt = (size_t)rc * p.T + t;
#pragma unroll
    for (int q = 0; q < 4; ++q) n_g[q] = *reinterpret_cast<const f32x4*>(p.Gt + rt * G4 + q * H + u0);
    n_c = *reinterpret_cast<const f32x4*>(p.Cs + rt * H + u0);
    n_cp = *reinterpret_cast<const f32x4*>(p.Cs + (t > 0 ? rt - 1 : rt) * H + u0);
  };
  fetch(p.T - 1);
  for (int t = p.T - 1; t >= 0; --t) {
    const size_t rt = (size_t)rc * p.T + t;
    const f32x4 gi = n_g[0], gf = n_g[1], gg = n_g[2], go = n_g[3], cc = n_c;
    const float cm = t > 0 ? 1.f : 0.f;
    const f32x4 cp = n_cp * cm;
    fetch(t > 0 ? t - 1 : 0);
    float* tile = dps[t & 1][rtile];
    f32x4 dq[4];  // [gate][unit]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float tc = mg_tanh(cc[e]);
      const float dO = dh[e] * tc;
      const float dcv = fmaf(dh[e] * go[e], 1.f - tc * tc, dc[e]);
      dq[0][e] = dcv * gg[e] * gi[e] * (1.f - gi[e]) * vm;
      dq[1][e] = dcv * cp[e] * gf[e] * (1.f - gf[e]) * vm;
      dq[2][e] = dcv * gi[e] * (1.f - gg[e] * gg[e]) * vm;
      dq[3][e] = dO * go[e] * (1.f - go[e]) * vm;
      dc[e] = dcv * gf[e];
      *reinterpret_cast<f32x4*>(&tile[fi * LD + (u0 + e) * 4]) = f32x4{dq[0][e], dq[1][e], dq[2][e], dq[3][e]};
    }
    if (valid) {
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(p.dPre + rt * G4 + q * H + u0) = dq[q];
    }
    lds_barrier();
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int S = 0; S < KS / 4; ++S) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(&tile[fi * LD + 16 * S + 4 * fk]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = MFMA16(Ah[4 * S + i], b4[i], acc[i]);
    }
    dh = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  }
}

// ------------------------------------------------------------------------------------
// Decoder rollout, matrix-core form.  A workgroup of four waves owns a tile of 16 rollout rows of one
// generator for all T steps.  Per step ONE matrix phase with B = h_t (read from a 2 KB LDS tile):
//     [ W_hh ; W1[:, :H] ] . h_t^T   ->   G_{t+1} = W_hh h_t  (gate pre-activations of the NEXT step, without
//                                         the rank-2 input term)  and  u_t = W1[:, :H] h_t + q  (hidden2pos)
// on v_mfma_f32_16x16x4_f32 (exact f32).  Wave w owns hidden units 8w .. 8w+7.  The rows of its two gate
// M-tiles are permuted so that D register r of lane (fi = lane & 15, fk = lane >> 4) of tile mt is gate r
// (i, f, g, o) of unit 8w + 4mt + fk for tile row fi: the cell update is lane-local (two units per lane), the
// autoregressive input enters as two FMAs per gate (A = W_ih W_emb folded), and dxdy = W2 u + b2 is four FMAs
// plus two cross-lane adds.  The u tile is computed by every wave (8 MFMAs) instead of a second exchange.
// Saved for backward in TILE-BLOCKED form: tile tg (= tiles of the generators before it + its index in its own
// generator, see dec_tile_base) owns one contiguous record per step, rows innermost, so that every wave-wide store
// here and load in the backward kernel is one contiguous 0.5-1 KB run (row-major (R,T,H,4) made each of them 16
// separate 64-byte pieces 6 KB apart: 25 % slower backward):
//     Gt (tiles,T,H,16,4)   = gates (i,f,g,o) after activation        Aact (tiles,T,4,16,4) = hidden2pos activations
//     Cs (tiles,T+1,H,16,2) = (c_{t-1}, h_{t-1}) at step index t, i.e. slot 0 = (0, h_0)     Din (tiles,T,16,2)
// with tiles = sum_g ceil(rows_g / 16) <= ceil(R/16) + n_gens.  Rows past a generator's end compute on a clamped
// duplicate and ARE saved (finite values the backward kernel multiplies by zero).
#define DEC_HLD 36  // LDS h tile row stride in floats (16-byte aligned rows, conflict-light)
// Optional padding between the tile records of the saved state (floats, multiples of 4).  A tile's record starts a multiple
// of 96 KB (gates) / 52 KB (cell state) after the first; padding them apart (272 / 144 floats) to spread workgroups that
// walk their tiles in step over the memory channels was measured and changes nothing (576 vs 571-591 us): 0 by default.
#ifndef DEC_GT_PAD
#define DEC_GT_PAD 0
#endif
#ifndef DEC_CS_PAD
#define DEC_CS_PAD 0
#endif

// Q^T [32 x 16 pedestrians] = W_e2d[:, :EIN] enc_h^T + b: one wave per 16-pedestrian tile, both unit tiles
__global__ __launch_bounds__(256) void e2d_shared_kernel(const float* __restrict__ enc_h, int ld_enc, int b, int EIN,
                                                         const float* __restrict__ We2d, int ldw,
                                                         const float* __restrict__ be2d, float* __restrict__ Q) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  const int ped = (blockIdx.x * 4 + w) * 16 + fi;
  if ((blockIdx.x * 4 + w) * 16 >= b) return;
  const bool valid = ped < b;
  const float* xr = enc_h + (size_t)(valid ? ped : b - 1) * ld_enc + 4 * fk;
  const float* w0 = We2d + (size_t)fi * ldw + 4 * fk;
  const float* w1 = We2d + (size_t)(16 + fi) * ldw + 4 * fk;
  f32x4 a0 = *reinterpret_cast<const f32x4*>(be2d + 4 * fk), a1 = *reinterpret_cast<const f32x4*>(be2d + 16 + 4 * fk);
  f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0;
  for (int k = 0; k < EIN; k += 16) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(xr + k);
    const f32x4 u = *reinterpret_cast<const f32x4*>(w0 + k), v = *reinterpret_cast<const f32x4*>(w1 + k);
    a0 = MFMA16(u[0], x[0], a0); a1 = MFMA16(v[0], x[0], a1);
    c0 = MFMA16(u[1], x[1], c0); c1 = MFMA16(v[1], x[1], c1);
    a0 = MFMA16(u[2], x[2], a0); a1 = MFMA16(v[2], x[2], a1);
    c0 = MFMA16(u[3], x[3], c0); c1 = MFMA16(v[3], x[3], c1);
  }
  if (valid) {  // D rows 4 fk + r = units, column fi = pedestrian
    *reinterpret_cast<f32x4*>(Q + (size_t)ped * 32 + 4 * fk) = a0 + c0;
    *reinterpret_cast<f32x4*>(Q + (size_t)ped * 32 + 16 + 4 * fk) = a1 + c1;
  }
}

struct DecFwdArgs {
  int T, b, NW, Rout, EIN, Z, ld_enc, ld_soc;
  const int* seg;
  const float* prep;
  int prep_stride;
  const int *row_ped, *row_slot, *row_pos;
  const float *enc_h, *noise, *soc, *xy0, *dxdy0, *We2d, *be2d;
  float *out_abs, *out_rel;
  float *Gt, *Cs, *Din, *Aact, *E2Din, *SocR;
  const float* Qe;  // (b, H) = b_e2d + W_e2d[:, :EIN] enc_h per pedestrian, or NULL (then the kernel does that product per row)
  float* Nz;        // with Qe: (R, Z) the rows' noise vectors, kept for the weight gradient instead of E2Din
};

// first tile slot of generator gi in the tile-blocked save buffers
__device__ __forceinline__ int dec_tile_base(const int* seg, int gi) {
  int tb = 0;
  for (int g = 0; g < gi; ++g) tb += (seg[g + 1] - seg[g] + 15) / 16;
  return tb;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void decoder_fwd_mfma_kernel(DecFwdArgs p) {
  constexpr int H = 32, G4 = 128, Hh = 16;
  __shared__ __attribute__((aligned(16))) float hs[2][16 * DEC_HLD];
  const int gi = blockIdx.x / p.NW, wi = blockIdx.x % p.NW;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  const float* P = p.prep + (size_t)gi * p.prep_stride;
  const float* WT = P + prep_off_whhT(H);
  const float* w1T = P + prep_off_w1T(H);
  const float* b1 = w1T + 2 * H * Hh;
  const float* w2 = b1 + Hh;
  const float* b2 = w2 + 2 * Hh;
  const int seg0 = p.seg[gi], seg1 = p.seg[gi + 1];
  const int ntiles = (seg1 - seg0 + 15) / 16;
  const bool save = p.Gt != nullptr;
  const int IN = p.EIN + p.Z;
  const int tbase = save ? dec_tile_base(p.seg, gi) : 0;

  for (int tile = wi; tile < ntiles; tile += p.NW) {
    const int r = seg0 + tile * 16 + fi;
    const bool valid = r < seg1;
    const int rc = valid ? r : seg1 - 1;
    const int ped = p.row_ped[rc], slot = p.row_slot[rc], pos = p.row_pos[rc];
    const bool sv = save && valid, sv0 = sv && w == 0;  // row-major saves (E2Din, SocR)
    const size_t tg = (size_t)(tbase + tile);
    // ---- h0 = W_e2d [enc_h | noise] + b_e2d (standard.py:247-252): wave w -> units 8w .. 8w+7 (M rows 0..7) ----
    f32x4 hacc = f32x4{0.f, 0.f, 0.f, 0.f}, hacc2 = f32x4{0.f, 0.f, 0.f, 0.f};
    auto e2d_block = [&](int kb, f32x4& x4, f32x4& a4) {  // loads of one 16-wide K block (this lane's quad)
      const int kq = 16 * kb + 4 * fk;
      const bool kin = kq + 3 < IN;
      x4 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kq + 3 < p.EIN) x4 = *reinterpret_cast<const f32x4*>(p.enc_h + (size_t)ped * p.ld_enc + kq);
      else if (kq >= p.EIN && kin)
        x4 = *reinterpret_cast<const f32x4*>(p.noise + ((size_t)slot * p.b + ped) * p.Z + (kq - p.EIN));
      if (sv0 && kin) {
        if (p.Qe) *reinterpret_cast<f32x4*>(p.Nz + (size_t)r * p.Z + (kq - p.EIN)) = x4;
        else *reinterpret_cast<f32x4*>(p.E2Din + (size_t)r * IN + kq) = x4;
      }
      // A rows = this wave's units: W_e2d[8w + fi][kq .. kq+3] (row-major parameter, one 16-byte load)
      a4 = (kin && fi < 8) ? *reinterpret_cast<const f32x4*>(p.We2d + (size_t)(8 * w + fi) * IN + kq)
                           : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // The enc_h part of the product is the same for the K rows of a pedestrian (and for every generator: W_e2d is
    // shared): with Qe it arrives as 32 floats per row and only the noise columns are multiplied here - 4 MFMAs per
    // wave and tile instead of 36, no 512-byte enc_h gather per row, no (R, EIN) copy kept for the weight gradient.
    if (p.Qe && fk < 2) hacc = *reinterpret_cast<const f32x4*>(p.Qe + (size_t)ped * H + 8 * w + 4 * fk);
    for (int kb0 = p.Qe ? p.EIN / 16 : 0; kb0 * 16 < IN; kb0 += 3) {  // three blocks of loads in flight, then their 12 MFMAs
      f32x4 xs[3], as[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) e2d_block(kb0 + i, xs[i], as[i]);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int q = 0; q < 4; q += 2) {
          hacc = MFMA16(as[i][q], xs[i][q], hacc);
          hacc2 = MFMA16(as[i][q + 1], xs[i][q + 1], hacc2);
        }
    }
    hacc += hacc2;
    // A operands (lane = (M row fi, K index fk) of the 16x4 fragment); K index of step ks is 8 fk + ks.  Loaded per
    // tile, after the h0 phase, so that its loads in flight and these 60 registers are not live together
    float Wg[2][8], Wu[8];
  #pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
  #pragma unroll
      for (int mt = 0; mt < 2; ++mt) Wg[mt][ks] = WT[(8 * fk + ks) * G4 + (fi & 3) * H + 8 * w + 4 * mt + (fi >> 2)];
      Wu[ks] = w1T[(8 * fk + ks) * Hh + fi];
    }
    // per-lane coefficients of its D values: gate r of unit uj[mt]
    int uj[2];
    float ca0[2][4], ca1[2][4], cb[2][4], w2r[2][4], b1r[4];
  #pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      uj[mt] = 8 * w + 4 * mt + fk;
  #pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = q * H + uj[mt];
        ca0[mt][q] = P[prep_off_A(H) + m * 2];
        ca1[mt][q] = P[prep_off_A(H) + m * 2 + 1];
        cb[mt][q] = P[prep_off_bias(H) + m];
      }
    }
  #pragma unroll
    for (int q = 0; q < 4; ++q) {
      w2r[0][q] = w2[4 * fk + q];
      w2r[1][q] = w2[Hh + 4 * fk + q];
      b1r[q] = b1[4 * fk + q];
    }
    const float b20 = b2[0], b21 = b2[1];
    lds_barrier();  // the previous tile's last reads of hs[0] are done
    if (fk < 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) hacc[q] += p.Qe ? 0.f : p.be2d[8 * w + 4 * fk + q];  // (Qe carries the bias)
      *reinterpret_cast<f32x4*>(&hs[0][fi * DEC_HLD + 8 * w + 4 * fk]) = hacc;
      if (save) {
        // (the lane's offset afresh: computed once per workgroup its 64-bit form was the one value too many for three
        // waves per SIMD -- it went to scratch and came back per tile)
        int lo = ((8 * w + 4 * fk) * 16 + fi) * 2;
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float2*>(p.Cs + tg * DEC_CS_PAD + (tg * (p.T + 1) * H + q) * 32 + lo) = float2{0.f, hacc[q]};
      }
    }
    // ---- time-invariant social half of hidden2pos: q = W1[:, H:] soc + b1 (every wave) ----
    f32x4 qv = f32x4{b1r[0], b1r[1], b1r[2], b1r[3]};
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.soc + (size_t)ped * p.ld_soc + 16 * hh + 4 * fk);
      if (sv0) *reinterpret_cast<f32x4*>(p.SocR + (size_t)r * H + 16 * hh + 4 * fk) = s4;
#pragma unroll
      for (int q = 0; q < 4; ++q) qv = MFMA16(w1T[(H + 16 * hh + 4 * fk + q) * Hh + fi], s4[q], qv);
    }
    float d0 = p.dxdy0[ped * 2], d1 = p.dxdy0[ped * 2 + 1];
    float x0 = p.xy0[ped * 2], x1 = p.xy0[ped * 2 + 1];
    float c[2] = {0.f, 0.f};
    lds_barrier();
    f32x4 G[2];
    {
      const f32x4 ha = *reinterpret_cast<const f32x4*>(&hs[0][fi * DEC_HLD + 8 * fk]);
      const f32x4 hb = *reinterpret_cast<const f32x4*>(&hs[0][fi * DEC_HLD + 8 * fk + 4]);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        G[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) G[mt] = MFMA16(Wg[mt][ks], ha[ks], G[mt]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) G[mt] = MFMA16(Wg[mt][4 + ks], hb[ks], G[mt]);
      }
    }

    for (int t = 0; t < p.T; ++t) {
      const size_t tt = tg * p.T + t, tc = tg * (p.T + 1) + t + 1;
      float* hw = hs[(t + 1) & 1];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 g;
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = fmaf(ca1[mt][q], d1, fmaf(ca0[mt][q], d0, G[mt][q] + cb[mt][q]));
        g[0] = mg_sigmoid(g[0]); g[1] = mg_sigmoid(g[1]); g[2] = mg_tanh(g[2]); g[3] = mg_sigmoid(g[3]);
        c[mt] = fmaf(g[1], c[mt], g[0] * g[2]);
        const float hn = g[3] * mg_tanh(c[mt]);
        hw[fi * DEC_HLD + uj[mt]] = hn;
        if (save) {
          *reinterpret_cast<f32x4*>(p.Gt + tg * DEC_GT_PAD + ((tt * H + uj[mt]) * 16 + fi) * 4) = g;
          *reinterpret_cast<float2*>(p.Cs + tg * DEC_CS_PAD + ((tc * H + uj[mt]) * 16 + fi) * 2) = float2{c[mt], hn};
        }
      }
      if (save && w == 0 && fk == 0) *reinterpret_cast<float2*>(p.Din + (tt * 16 + fi) * 2) = float2{d0, d1};
      lds_barrier();
      const f32x4 ha = *reinterpret_cast<const f32x4*>(&hw[fi * DEC_HLD + 8 * fk]);
      const f32x4 hb = *reinterpret_cast<const f32x4*>(&hw[fi * DEC_HLD + 8 * fk + 4]);
      // u = LeakyReLU(W1[:, :H] h + q): lane (fi, fk) holds units m = 4 fk + r of row fi
      f32x4 ua = qv, ub = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        ua = MFMA16(Wu[ks], ha[ks], ua);
        ub = MFMA16(Wu[4 + ks], hb[ks], ub);
      }
      if (t + 1 < p.T) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) G[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) G[mt] = MFMA16(Wg[mt][ks], ha[ks], G[mt]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) G[mt] = MFMA16(Wg[mt][4 + ks], hb[ks], G[mt]);
      }
      f32x4 av;
      float n0 = 0.f, n1 = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float u = ua[q] + ub[q];
        av[q] = u > 0.f ? u : 0.01f * u;  // LeakyReLU(0.01), utils.py:143-144
        n0 = fmaf(w2r[0][q], av[q], n0);
        n1 = fmaf(w2r[1][q], av[q], n1);
      }
      if (save && w == 0) *reinterpret_cast<f32x4*>(p.Aact + ((tt * 4 + fk) * 16 + fi) * 4) = av;
      n0 = quarters_sum(n0);  // v_permlane{16,32}_swap: the four lane groups hold the four unit quarters
      n1 = quarters_sum(n1);
      n0 += b20; n1 += b21;
      d0 = n0; d1 = n1;
      x0 += n0; x1 += n1;
      if (valid && w == 0 && fk == 0) {
        const size_t o = ((size_t)t * p.Rout + pos) * 2;
        *reinterpret_cast<float2*>(p.out_abs + o) = float2{x0, x1};
        *reinterpret_cast<float2*>(p.out_rel + o) = float2{n0, n1};
      }
    }
  }
}

// ---- the rollout forward, ONE WAVE per 16-row tile ------------------------------------------------------------
// The four-wave kernel above splits the 32 hidden units of a tile over four waves: every step ends in an LDS exchange of
// h and a barrier, every wave repeats the position head, and a workgroup's waves wait for each other's transcendentals.
// Here a wave owns ALL 32 units of its 16 rows and nothing leaves its registers:
//   * gates: G^T[128 x 16] = [W_hh | A | b] [h ; d ; 1] as 8 M tiles x 9 k steps of v_mfma_f32_16x16x4_f32.  A row of tile t
//     is (gate fi & 3, unit 4 t + (fi >> 2)), so the D fragment of lane (row n, group fk) is the four gates of unit 4 t + fk:
//     the cell update is lane-local, and the lane ends up with h of the units 4 t + fk, t = 0..7.
//   * the reduction index of an MFMA may be permuted freely as long as both operands agree: k step ks, group fk <-> unit
//     4 ks + fk.  The B operand of the NEXT step's products (h[unit 4 ks + fk][n]) is then exactly the value the lane just
//     computed for tile ks -- the D fragment of one step is the B operand of the next, with no LDS, no barrier, no shuffle.
//   * the folded input embedding (two coefficients per gate row) and the bias ride as a ninth k step against [d0, d1, 1, 0]
//     instead of 96 coefficient registers and three FMAs per gate.
//   * the position head's hidden layer u = W1[:, :H] h + q is one more M tile over the same B operands (8 MFMAs per wave
//     and step instead of 8 per wave in each of four waves).
// 80 MFMAs per wave and step for 16 rows (four-wave kernel: 4 x 24 = 96), eight independent accumulator chains, and the
// waves of a SIMD belong to different tiles: one wave's MFMAs run under another's cell update.  The saved state has the
// layout of the four-wave kernel (the adjoint below reads it unchanged).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void decoder_fwd_wave_kernel(DecFwdArgs p) {
  constexpr int H = 32, G4 = 128, Hh = 16;
  const int gi = blockIdx.x / p.NW, wi = blockIdx.x % p.NW;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  const float* P = p.prep + (size_t)gi * p.prep_stride;
  const float* WT = P + prep_off_whhT(H);
  const float* w1T = P + prep_off_w1T(H);
  const float* b1 = w1T + 2 * H * Hh;
  const float* w2 = b1 + Hh;
  const float* b2 = w2 + 2 * Hh;
  const int seg0 = p.seg[gi], seg1 = p.seg[gi + 1];
  const int ntiles = (seg1 - seg0 + 15) / 16;
  if (wi * 4 + w >= ntiles) return;  // (no barrier anywhere below: a wave may leave on its own)
  const bool save = p.Gt != nullptr;
  const int IN = p.EIN + p.Z;
  const int tbase = save ? dec_tile_base(p.seg, gi) : 0;
  // ---- loop-invariant A operands (this generator's weights; L2-resident) ----
  float Ag[8][9], Au[8];
  {
    const int grow = (fi & 3) * H + (fi >> 2);  // + 4 t: gate row of W_hh / A / bias this lane supplies for tile t
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) Ag[t][ks] = WT[(4 * ks + fk) * G4 + grow + 4 * t];
      const int m = grow + 4 * t;
      Ag[t][8] = fk == 0 ? P[prep_off_A(H) + 2 * m] : fk == 1 ? P[prep_off_A(H) + 2 * m + 1]
                 : fk == 2 ? P[prep_off_bias(H) + m] : 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) Au[ks] = w1T[(4 * ks + fk) * Hh + fi];
  }
  float w2r[2][4], b1r[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    w2r[0][q] = w2[4 * fk + q];
    w2r[1][q] = w2[Hh + 4 * fk + q];
    b1r[q] = b1[4 * fk + q];
  }
  const float b20 = b2[0], b21 = b2[1];

  for (int tile = wi * 4 + w; tile < ntiles; tile += p.NW * 4) {
    const int r = seg0 + tile * 16 + fi;
    const bool valid = r < seg1;
    const int rc = valid ? r : seg1 - 1;
    const int ped = p.row_ped[rc], slot = p.row_slot[rc], pos = p.row_pos[rc];
    const bool sv = save && valid;
    const size_t tg = (size_t)(tbase + tile);
    // ---- h0 = W_e2d [enc_h | noise] + b_e2d (standard.py:247-252): two M tiles whose rows are permuted so that the D
    //      fragment of lane (n, fk) holds the units 4 t + fk (row 4 fk' + r of tile j <-> unit 4 (r + 4 j) + fk') ----
    f32x4 h0a[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const int urow[2] = {4 * (fi & 3) + (fi >> 2), 4 * ((fi & 3) + 4) + (fi >> 2)};  // A row fi of tile j <-> this unit
    if (p.Qe) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) h0a[j][q] = p.Qe[(size_t)ped * H + 4 * (q + 4 * j) + fk];  // (carries the bias)
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) h0a[j][q] = p.be2d[4 * (q + 4 * j) + fk];
    }
    for (int kb = p.Qe ? p.EIN / 4 : 0; 4 * kb < IN; ++kb) {  // k step kb: inputs 4 kb + fk
      const int k = 4 * kb + fk;
      float x = 0.f;
      if (k < p.EIN) x = p.enc_h[(size_t)ped * p.ld_enc + k];
      else if (k < IN) x = p.noise[((size_t)slot * p.b + ped) * p.Z + (k - p.EIN)];
      if (sv && k < IN) {
        if (p.Qe) p.Nz[(size_t)r * p.Z + (k - p.EIN)] = x;
        else p.E2Din[(size_t)r * IN + k] = x;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) h0a[j] = MFMA16(k < IN ? p.We2d[(size_t)urow[j] * IN + k] : 0.f, x, h0a[j]);
    }
    float h[8], c[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      h[t] = h0a[t >> 2][t & 3];
      c[t] = 0.f;
    }
    if (save) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
        *reinterpret_cast<float2*>(p.Cs + tg * DEC_CS_PAD + ((tg * (p.T + 1) * H + 4 * t + fk) * 16 + fi) * 2) = float2{0.f, h[t]};
    }
    // ---- time-invariant social half of hidden2pos: q = W1[:, H:] soc + b1 ----
    f32x4 qv = f32x4{b1r[0], b1r[1], b1r[2], b1r[3]};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const float sx = p.soc[(size_t)ped * p.ld_soc + 4 * ks + fk];
      if (sv) p.SocR[(size_t)r * H + 4 * ks + fk] = sx;
      qv = MFMA16(w1T[(H + 4 * ks + fk) * Hh + fi], sx, qv);
    }
    float d0 = p.dxdy0[ped * 2], d1 = p.dxdy0[ped * 2 + 1];
    float x0 = p.xy0[ped * 2], x1 = p.xy0[ped * 2 + 1];

    f32x4 gS[8];  // the gates of the previous step, stored one step late (see below)
    for (int t = 0; t < p.T; ++t) {
      const size_t tt = tg * p.T + t, tc = tg * (p.T + 1) + t + 1;
      // gates of this step from h_{t-1} (the lane's own values are the B operands) and the last displacement
      const float din = fk == 0 ? d0 : fk == 1 ? d1 : fk == 2 ? 1.f : 0.f;
      f32x4 G[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) G[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
        for (int u = 0; u < 8; ++u) G[u] = MFMA16(Ag[u][ks], h[ks], G[u]);
        // the saved state of step t-1 leaves HERE, one unit per k step, between the matrix instructions of step t
        // (instead of 16 stores in one burst behind the cell update)
        if (save && t > 0) {
          *reinterpret_cast<f32x4*>(p.Gt + tg * DEC_GT_PAD + (((tt - 1) * H + 4 * ks + fk) * 16 + fi) * 4) = gS[ks];
          *reinterpret_cast<float2*>(p.Cs + tg * DEC_CS_PAD + (((tc - 1) * H + 4 * ks + fk) * 16 + fi) * 2) = float2{c[ks], h[ks]};
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) G[u] = MFMA16(Ag[u][8], din, G[u]);
      if (save && fk == 0) *reinterpret_cast<float2*>(p.Din + (tt * 16 + fi) * 2) = float2{d0, d1};
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        f32x4 g = G[u];
        g[0] = mg_sigmoid(g[0]); g[1] = mg_sigmoid(g[1]); g[2] = mg_tanh(g[2]); g[3] = mg_sigmoid(g[3]);
        c[u] = fmaf(g[1], c[u], g[0] * g[2]);
        h[u] = g[3] * mg_tanh(c[u]);
        gS[u] = g;
      }
      if (save && t == p.T - 1) {  // the last step's record has no next step to hide under
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          *reinterpret_cast<f32x4*>(p.Gt + tg * DEC_GT_PAD + ((tt * H + 4 * u + fk) * 16 + fi) * 4) = gS[u];
          *reinterpret_cast<float2*>(p.Cs + tg * DEC_CS_PAD + ((tc * H + 4 * u + fk) * 16 + fi) * 2) = float2{c[u], h[u]};
        }
      }
      // u = LeakyReLU(W1[:, :H] h_t + q): lane (n, fk) holds hidden units 4 fk + r of row n
      f32x4 ua = qv, ub = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ks += 2) {
        ua = MFMA16(Au[ks], h[ks], ua);
        ub = MFMA16(Au[ks + 1], h[ks + 1], ub);
      }
      f32x4 av;
      float n0 = 0.f, n1 = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float uu = ua[q] + ub[q];
        av[q] = uu > 0.f ? uu : 0.01f * uu;  // LeakyReLU(0.01), utils.py:143-144
        n0 = fmaf(w2r[0][q], av[q], n0);
        n1 = fmaf(w2r[1][q], av[q], n1);
      }
      if (save) *reinterpret_cast<f32x4*>(p.Aact + ((tt * 4 + fk) * 16 + fi) * 4) = av;
      n0 = quarters_sum(n0) + b20;  // v_permlane{16,32}_swap: the four lane groups hold the four unit quarters
      n1 = quarters_sum(n1) + b21;
      d0 = n0; d1 = n1;
      x0 += n0; x1 += n1;
      if (valid && fk == 0) {
        const size_t o = ((size_t)t * p.Rout + pos) * 2;
        *reinterpret_cast<float2*>(p.out_abs + o) = float2{x0, x1};
        *reinterpret_cast<float2*>(p.out_rel + o) = float2{n0, n1};
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Fused decoder backward: BPTT over the 12 steps AND the per-generator weight gradients in one launch.
// The un-fused path writes the gate gradients (512 B per row and step, 157 MB at 25,600 rows) and two
// weight-gradient GEMMs read them back: HBM-bound.  Here a workgroup is persistent over the 8-row tiles of
// ONE generator; per step the rank-8 updates
//     dW_hh[128x32] += dPre^T[128x8] h_{t-1}[8x32]      dW1[:, :32][16x32] += dU^T[16x8] h_t[8x32]
// run on the matrix cores (exact-f32 16x16x4 MFMA) straight from the LDS tiles the recurrence already
// needs; dA / dbias / dW2 / db1 / db2 are lane-local FMAs.  Each workgroup leaves ONE partial block
//     [ W_hh 4096 | A 256 | bias 128 | W1h 512 | b1 16 | W2 32 | b2 2 (+2 pad) ]
// which the batched gradient reduce folds per generator.
#define DF_WLEN 5556
#define DF_OFF_W1S 5044  // dW1[:, H:] (16 x 32): the social half of hidden2pos, K = the rows of the workgroup's tiles
#define DF_OFF_A 4096
#define DF_OFF_B 4352
#define DF_OFF_W1 4480
#define DF_OFF_B1 4992
#define DF_OFF_W2 5008
#define DF_OFF_B2 5040
#define DPLD 144  // dpbuf row stride (== 16 mod 32: conflict-free MFMA fragment reads, 16-B aligned rows)
#define HLD 48    // h tile row stride (== 16 mod 32)
struct DecFusedArgs {
  int T, NW, Rout, EIN, Z, e2ld;
  const int* seg;        // generator segment offsets into the sorted rows (g+1)
  const int* row_pos;
  const float *W_hh, *W1, *W2, *We2d;
  long param_stride;
  const float* prep;
  int prep_stride;
  const float *Gt, *Cs, *Din, *Aact, *gabs, *grel;
  float *dH0, *dQ, *dEnc, *dSocR, *wpart;
  const float* SocR;  // (R, S) the rows' social features (saved by the forward), or NULL: no dW1[:, H:] in the block
};

// Matrix-core form of the fused backward, in the lane layout of decoder_fwd_mfma_kernel: four waves per
// 16-row tile, wave w owns hidden units 8w .. 8w+7, lane (fi = lane & 15, fk = lane >> 4) holds the cell
// state gradients of units 8w + fk and 8w + 4 + fk of tile row fi.  Per step, ONE barrier:
//   1. du = W2^T g . LeakyReLU'(u)  (lane-local, units m = 4 fk + r)      dh += W1[:, :H]^T du      (4 MFMAs)
//   2. gate gradients dPre for the lane's two units (VALU) -> LDS tile dps[row][unit*4 + gate]
//   3. barrier; then with B = dPre^T read back as eight 16-byte LDS loads:
//        [ W_hh^T (own 8 units) ; A^T ] . dPre^T  ->  dh_{t-1} of the own units and d(dxdy_t)       (32 MFMAs)
//      (M rows 4 fk + {0,1} = the lane's two units, rows 4 fk + {2,3} = the two input components: every lane
//      gets the complete sums in its D registers, no partial exchange)
//   4. weight gradients with K = the 16 tile rows: dW_hh += dPre^T h_{t-1} (wave w: position tiles 2w, 2w+1 x two
//      column tiles, 16 MFMAs), dW1[:, :H] += du^T h_t (2 MFMAs); dA / dbias = dPre^T [dxdy | 1] are lane-local sums
// The LDS tiles are multi-buffered over t so that the single barrier per step is enough.
#define DB_RS 148  // dPre tile row stride == 20 mod 64: the 16-byte row accesses of a 16-lane group land on banks 20 fi (+0..3),
                   // sixteen distinct 4-bank groups, and the transposed 4-byte reads of the weight-gradient phase (lane =
                   // (position fi, row 4 ks + fk)) start 20 banks apart per fk - 2-way on 12 banks (stride 132: 4-way)
#define DB_HS 36   // h tile row stride, and
#define DB_US 20   // du tile row stride: 4 x stride == 16 mod 32.  The weight-gradient products walk the 16 tile rows as
                   // row = step + 4 k (k = the MFMA's reduction lane group): the two lane groups of a 32-lane LDS pass then
                   // start 4 x stride = 16 banks apart on all three tiles (4 x 148 == 16 mod 32 too).  With row = 4 step + k
                   // the dPre reads of the two groups sat 20 banks apart, 2-way on 12 banks: 45 % of this kernel's LDS
                   // cycles were bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void decoder_bwd_mfma_kernel(DecFusedArgs p) {
  constexpr int H = 32, Hh = 16, S = 32;
  __shared__ __attribute__((aligned(16))) float dps[2][16 * DB_RS];
  __shared__ __attribute__((aligned(16))) float hts[3][16 * DB_HS];
  __shared__ __attribute__((aligned(16))) float dus[2][16 * DB_US];
  __shared__ float red[16 * 52];
  extern __shared__ __attribute__((aligned(16))) float tailw[];  // W_e2d[:, :EIN] (H x e2ld) | W1[:, H:] (Hh x 36)
  const int gi = blockIdx.x / p.NW, wi = blockIdx.x % p.NW;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  const long po = (long)gi * p.param_stride;
  const float* P = p.prep + (size_t)gi * p.prep_stride;
  const float* Whh = p.W_hh + po;
  const float* W1 = p.W1 + po;
  const int seg0 = p.seg[gi], seg1 = p.seg[gi + 1];
  const int ntiles = (seg1 - seg0 + 15) / 16;
  const int IN = p.EIN + p.Z;
  float* wp = p.wpart + (size_t)blockIdx.x * DF_WLEN;
  if (wi >= ntiles) {  // NW leaves room for an uneven split between the generators: no tile, an all-zero partial block
    for (int i = threadIdx.x; i < DF_WLEN; i += 256) wp[i] = 0.f;
    return;
  }
  const int tbase = dec_tile_base(p.seg, gi);
  const int uj[2] = {8 * w + fk, 8 * w + 4 + fk};
  const int sel = fi & 3, us = 8 * w + 4 * (sel & 1) + (fi >> 2);  // A-operand row role of this lane

  // A operands.  Position p = unit*4 + gate in the dPre tile <-> original gate row (p & 3)*H + (p >> 2).
  float Ah[32], Aw1[4], w2c[2][4];
#pragma unroll
  for (int ks = 0; ks < 32; ++ks) {
    const int pp = 16 * (ks >> 2) + 4 * fk + (ks & 3), m = (pp & 3) * H + (pp >> 2);
    Ah[ks] = sel < 2 ? Whh[(size_t)m * H + us] : P[prep_off_A(H) + m * 2 + (sel - 2)];
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) Aw1[ks] = sel < 2 ? W1[(size_t)(4 * fk + ks) * (H + S) + us] : 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    w2c[0][q] = p.W2[po + 4 * fk + q];
    w2c[1][q] = p.W2[po + Hh + 4 * fk + q];
  }
  // The per-tile epilogue (dEnc, dSocR) multiplies by W_e2d[:, :EIN] and W1[:, H:]: from global memory that was three
  // exposed load round trips per tile (10k of a tile's 75k cycles); they are staged once per workgroup instead.
  float* e2s = tailw;
  float* w1s = tailw + H * p.e2ld;
  // (dEnc == NULL: the caller sums dH0 over the rows of a pedestrian first and multiplies once per pedestrian)
  if (p.dEnc)
    for (int i = threadIdx.x; i < H * p.EIN; i += 256) e2s[(i / p.EIN) * p.e2ld + i % p.EIN] = p.We2d[(size_t)(i / p.EIN) * IN + i % p.EIN];
  for (int i = threadIdx.x; i < Hh * S; i += 256) w1s[(i / S) * 36 + i % S] = W1[(size_t)(i / S) * (H + S) + H + i % S];
  f32x4 accW[2][2], accU = f32x4{0.f, 0.f, 0.f, 0.f}, accS = f32x4{0.f, 0.f, 0.f, 0.f};
  // dA (4H x 2) and dbias (4H) = dPre^T [dxdy | 1]: lane-local sums over the lane's own (unit, gate) entries and its tile
  // row, folded over the 16 rows once at the end.  As a third 16-column tile of the matrix product (13 of its 16 columns
  // padding) they cost 8 of the 62 MFMAs of a step.
  float accA0[2][4], accA1[2][4], accBs[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q) accA0[a][q] = accA1[a][q] = accBs[a][q] = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) accW[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float accW2[2][4], accb1[4], accb2[2] = {0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q) accW2[0][q] = accW2[1][q] = accb1[q] = 0.f;

  for (int tile = wi; tile < ntiles; tile += p.NW) {
    const int r = seg0 + tile * 16 + fi;
    const bool valid = r < seg1;
    const int rc = valid ? r : seg1 - 1;
    const float vm = valid ? 1.f : 0.f;  // rows past the segment end contribute nothing
    const int pos = p.row_pos[rc];
    const size_t tg = (size_t)(tbase + tile);
    float dh[2] = {0.f, 0.f}, dc[2] = {0.f, 0.f}, dd0 = 0.f, dd1 = 0.f, s0 = 0.f, s1 = 0.f;
    f32x4 dq = f32x4{0.f, 0.f, 0.f, 0.f};
    // Saved activations of the next step, fetched one whole step ahead into a second register set.  (The 512
    // workgroups run in step and ask for their 13 KB records at the same moment: served as a burst, the last of them
    // waits for most of a step.  Fetching after the step's barrier without the second set cost 15 %; moving the
    // records global -> LDS by DMA, asm-issued so that the compiler does not drain vmcnt early, freed the registers
    // but exposed the same wait in front of the barrier: no gain, not kept.)
    f32x4 n_g[2], n_av;
    float2 n_ch[2], n_din, n_ga, n_gr;
    float cc[2];
    const float gam = p.gabs ? 1.f : 0.f, grm = p.grel ? 1.f : 0.f;
    // A missing output gradient is zero: its ADDRESS is redirected and the value masked where it is consumed - a
    // select on a value just loaded makes the compiler wait for the whole batch of loads on the spot (that is how
    // this loop lost its prefetch once).  Every load below is one contiguous run per wave (tile-blocked saves).
    auto fetch = [&](int t) {
      const size_t tt = tg * p.T + t, tc = tg * (p.T + 1) + t;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        n_g[mt] = *reinterpret_cast<const f32x4*>(p.Gt + tg * DEC_GT_PAD + ((tt * H + uj[mt]) * 16 + fi) * 4);
        n_ch[mt] = *reinterpret_cast<const float2*>(p.Cs + tg * DEC_CS_PAD + ((tc * H + uj[mt]) * 16 + fi) * 2);  // (c_{t-1}, h_{t-1})
      }
      n_av = *reinterpret_cast<const f32x4*>(p.Aact + ((tt * 4 + fk) * 16 + fi) * 4);
      const float* din = p.Din + (tt * 16 + fi) * 2;
      n_din = *reinterpret_cast<const float2*>(din);
      const size_t o = ((size_t)t * p.Rout + pos) * 2;
      n_ga = *reinterpret_cast<const float2*>(p.gabs ? p.gabs + o : din);
      n_gr = *reinterpret_cast<const float2*>(p.grel ? p.grel + o : din);
    };
    lds_barrier();  // the previous tile's last LDS reads are done
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const float2 ch = *reinterpret_cast<const float2*>(p.Cs + tg * DEC_CS_PAD + (((tg * (p.T + 1) + p.T) * H + uj[mt]) * 16 + fi) * 2);
      cc[mt] = ch.x;
      hts[p.T % 3][fi * DB_HS + uj[mt]] = ch.y;  // h_{T-1}
    }
    fetch(p.T - 1);

    for (int t = p.T - 1; t >= 0; --t) {
      float* dpw = dps[t & 1];
      float* duw = dus[t & 1];
      float* htw = hts[t % 3];              // receives h_{t-1}
      const float* htc = hts[(t + 1) % 3];  // holds h_t
      const f32x4 c_g[2] = {n_g[0], n_g[1]}, c_av = n_av;
      const float2 c_ch[2] = {n_ch[0], n_ch[1]}, c_din = n_din, c_ga = n_ga, c_gr = n_gr;
      fetch(t > 0 ? t - 1 : 0);  // (step 0 re-reads its own slots: no branch between the loads)
      s0 = fmaf(c_ga.x, gam, s0); s1 = fmaf(c_ga.y, gam, s1);
      const float g0 = (s0 + dd0 + c_gr.x * grm) * vm, g1 = (s1 + dd1 + c_gr.y * grm) * vm;
      f32x4 du;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        du[q] = fmaf(w2c[0][q], g0, w2c[1][q] * g1) * (c_av[q] > 0.f ? 1.f : 0.01f);
        accW2[0][q] = fmaf(g0, c_av[q], accW2[0][q]);
        accW2[1][q] = fmaf(g1, c_av[q], accW2[1][q]);
      }
      dq += du;
      accb2[0] += g0; accb2[1] += g1;
      if (w == 0) {
        *reinterpret_cast<f32x4*>(&duw[fi * DB_US + 4 * fk]) = du;
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) htw[fi * DB_HS + uj[mt]] = c_ch[mt].y;
      // dh += W1[:, :H]^T du  (rows 4 fk + {0,1} of the M tile are this lane's units)
      f32x4 a1 = f32x4{dh[0], dh[1], 0.f, 0.f}, a2 = f32x4{0.f, 0.f, 0.f, 0.f};
      a1 = MFMA16(Aw1[0], du[0], a1);
      a2 = MFMA16(Aw1[1], du[1], a2);
      a1 = MFMA16(Aw1[2], du[2], a1);
      a2 = MFMA16(Aw1[3], du[3], a2);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const float dhv = a1[mt] + a2[mt];
        const float gi_ = c_g[mt][0], gf = c_g[mt][1], gg = c_g[mt][2], go = c_g[mt][3];
        const float cprev = c_ch[mt].x;  // slot 0 of Cs holds c_{-1} = 0
        const float tc = mg_tanh(cc[mt]);
        const float dO = dhv * tc;
        const float dcv = fmaf(dhv * go, 1.f - tc * tc, dc[mt]);
        f32x4 dp;
        dp[0] = dcv * gg * gi_ * (1.f - gi_) * vm;
        dp[1] = dcv * cprev * gf * (1.f - gf) * vm;
        dp[2] = dcv * gi_ * (1.f - gg * gg) * vm;
        dp[3] = dO * go * (1.f - go) * vm;
        dc[mt] = dcv * gf;
        cc[mt] = cprev;
        *reinterpret_cast<f32x4*>(&dpw[fi * DB_RS + uj[mt] * 4]) = dp;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          accA0[mt][q] = fmaf(dp[q], c_din.x, accA0[mt][q]);
          accA1[mt][q] = fmaf(dp[q], c_din.y, accA1[mt][q]);
          accBs[mt][q] += dp[q];
        }
      }
      lds_barrier();
      // [dh_{t-1} (own units) ; d dxdy_t] = [W_hh^T ; A^T] dPre^T, K = 128 gate rows in tile-position order
      f32x4 acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(&dpw[fi * DB_RS + 16 * j + 4 * fk]);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = MFMA16(Ah[4 * j + q], b4[q], acc[q]);
      }
      // weight gradients, K = the 16 tile rows.  (Run one step late, beside the next step's gate arithmetic, they
      // took exactly as long: the two workgroups of a CU already fill each other's gaps.)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        float a[2], bv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = dpw[(ks + 4 * fk) * DB_RS + 16 * (2 * w + i) + fi];
#pragma unroll
        for (int n = 0; n < 2; ++n) bv[n] = htw[(ks + 4 * fk) * DB_HS + 16 * n + fi];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int n = 0; n < 2; ++n) accW[i][n] = MFMA16(a[i], bv[n], accW[i][n]);
      }
      {  // dW1[:, :H] += du^T h_t : wave w -> column tile (w & 1), K half (w >> 1)
        const int nt = w & 1, kh = w >> 1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int ks = 2 * kh + kk;
          accU = MFMA16(duw[(ks + 4 * fk) * DB_US + fi], htc[(ks + 4 * fk) * DB_HS + 16 * nt + fi], accU);
        }
      }
      const f32x4 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
      dh[0] = sum[0]; dh[1] = sum[1]; dd0 = sum[2]; dd1 = sum[3];
    }
    // ---- per-row outputs of this tile: dH0, dQ, d(enc_h row), d(social row) ----
    float* h0t = hts[2];  // h_{-1} slot of the t = 0 step is hts[0]; hts[2] held h_1 (last read at t = 1)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      if (valid) p.dH0[(size_t)r * H + uj[mt]] = dh[mt];
      h0t[fi * DB_HS + uj[mt]] = dh[mt];
    }
    if (w == 0 && valid) *reinterpret_cast<f32x4*>(p.dQ + (size_t)r * Hh + 4 * fk) = dq;
#pragma unroll
    for (int q = 0; q < 4; ++q) accb1[q] += dq[q];
    if (p.SocR) {
      // dW1[:, H:] += dQ^T SocR over the 16 rows of the tile (once per tile: dQ is the sum over the steps, the social
      // features do not change with the step): dQ goes through the du tile that the last step left free and comes back
      // transposed, wave w -> column tile (w & 1), K half (w >> 1) as for dW1[:, :H].  (As a grouped GEMM over all R rows
      // behind the backward pass this product was a launch of its own: 166 us at 163,840 rows, 26 us at 25,600.)
      float* dqs = dus[1];
      if (w == 0) *reinterpret_cast<f32x4*>(&dqs[fi * DB_US + 4 * fk]) = dq;
      lds_barrier();
      const int nt = w & 1, kh = w >> 1;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int row = 2 * kh + kk + 4 * fk;
        const int rr = min(seg0 + tile * 16 + row, seg1 - 1);  // (rows past the segment end: dQ is zero there)
        accS = MFMA16(dqs[row * DB_US + fi], p.SocR[(size_t)rr * S + 16 * nt + fi], accS);
      }
    }
    if (w < 2) {  // dSocR^T [S x rows] = W1[:, H:]^T dQ^T : wave w -> social columns 16 w .. 16 w + 15
      f32x4 ds = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ds = MFMA16(w1s[(4 * fk + ks) * 36 + 16 * w + fi], dq[ks], ds);
      if (valid) *reinterpret_cast<f32x4*>(p.dSocR + (size_t)r * S + 16 * w + 4 * fk) = ds;
    }
    if (p.dEnc) lds_barrier();
    if (p.dEnc) {  // dEnc^T [EIN x rows] = W_e2d[:, :EIN]^T dH0^T : wave w -> 16-column tiles w, w + 4, ...
      const f32x4 ha = *reinterpret_cast<const f32x4*>(&h0t[fi * DB_HS + 8 * fk]);
      const f32x4 hb = *reinterpret_cast<const f32x4*>(&h0t[fi * DB_HS + 8 * fk + 4]);
      for (int ct = w; ct * 16 < p.EIN; ct += 4) {
        const int col = 16 * ct + fi;
        const bool cin = col < p.EIN;
        f32x4 de = f32x4{0.f, 0.f, 0.f, 0.f}, de2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          de = MFMA16(cin ? e2s[(8 * fk + ks) * p.e2ld + col] : 0.f, ha[ks], de);
          de2 = MFMA16(cin ? e2s[(8 * fk + 4 + ks) * p.e2ld + col] : 0.f, hb[ks], de2);
        }
        de += de2;
        if (valid && 16 * ct + 4 * fk + 3 < p.EIN) *reinterpret_cast<f32x4*>(p.dEnc + (size_t)r * p.EIN + 16 * ct + 4 * fk) = de;
      }
    }
  }

  // ---- this workgroup's partial block ----
  // accW[i][n][q] of lane (fi, fk): tile position pp = 16 (2w + i) + 4 fk + q, column 16 n + fi
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pp = 16 * (2 * w + i) + 4 * fk + q, m = (pp & 3) * H + (pp >> 2);
      wp[m * H + fi] = accW[i][0][q];
      wp[m * H + 16 + fi] = accW[i][1][q];
    }
  // dA / dbias: fold the 16 tile rows (lanes fi of a fixed fk) in a fixed tree; lane fi == 0 stores
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v0 = accA0[mt][q], v1 = accA1[mt][q], v2 = accBs[mt][q];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        v0 += __shfl_xor(v0, o, 64);
        v1 += __shfl_xor(v1, o, 64);
        v2 += __shfl_xor(v2, o, 64);
      }
      if (fi == 0) {
        const int m = q * H + uj[mt];
        wp[DF_OFF_A + m * 2] = v0;
        wp[DF_OFF_A + m * 2 + 1] = v1;
        wp[DF_OFF_B + m] = v2;
      }
    }
  lds_barrier();
  // W1h: column tile (w & 1) is shared by waves w and w ^ 2 (K halves): sum them through LDS
  float* u = dps[0];  // [4][256], and the social half behind it
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    u[w * 256 + (4 * fk + q) * 16 + fi] = accU[q];
    u[1024 + w * 256 + (4 * fk + q) * 16 + fi] = accS[q];
  }
  if (w == 0) {  // lane-local accumulators (W2, b1, b2) -> fold over the 16 tile rows
    float* f = red + fi * 52;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f[4 * fk + q] = accW2[0][q];
      f[Hh + 4 * fk + q] = accW2[1][q];
      f[32 + 4 * fk + q] = accb1[q];
    }
    if (fk == 0) { f[48] = accb2[0]; f[49] = accb2[1]; }
  }
  lds_barrier();
  if (threadIdx.x < 50) {
    const int i = threadIdx.x;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q * 52 + i];
    wp[i < 32 ? DF_OFF_W2 + i : i < 48 ? DF_OFF_B1 + (i - 32) : DF_OFF_B2 + (i - 48)] = t;
  }
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int m = i / 32, k = i % 32, nt = k / 16;
    wp[DF_OFF_W1 + i] = u[nt * 256 + m * 16 + (k & 15)] + u[(nt + 2) * 256 + m * 16 + (k & 15)];
    wp[DF_OFF_W1S + i] = u[1024 + nt * 256 + m * 16 + (k & 15)] + u[1024 + (nt + 2) * 256 + m * 16 + (k & 15)];
  }
}

#ifdef DEC_PROFILE  // measurement build: segment cycle counts of workgroup 0, wave 0, printed at the end
#define DP_T(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); prof[k] += t_ - tlast; tlast = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DP_T(k) do {} while (0)
#endif
// ------------------------------------------------------------------------------------
// The fused backward with TWO waves per 16-row tile (round 4).  In decoder_bwd_mfma_kernel every wave's recurrence
// product has an M tile of 8 units + 8 copies of the two `d dxdy` rows: 128 products per tile and step where 64 carry
// information, and two tiles in flight per CU.  Here wave u of a tile owns units 16 u .. 16 u + 15 (lane (fi, fk): units
// 16 u + 4 fk + 0..3 of tile row fi -- the D rows of its recurrence product, so dh lands where the gate arithmetic wants
// it), the M axis of [W_hh^T] . dPre^T is all units (32 products per wave), the weight gradient dW_hh += dPre^T h is split by
// its M tiles (32 per wave), and d dxdy = A^T dPre (2 x 128 per row) is a lane-local dot product over the lane's own 16
// gate gradients, folded over the four lane groups (v_permlane swaps) and the two waves (LDS, behind the step's one
// barrier).  144 instead of 216 products per tile and step, the vector work per tile unchanged; a workgroup of four waves
// carries two tiles (slots) of one generator side by side, two workgroups per CU: four tiles in flight per CU.
// f32 products and vector instructions share the SIMD's pipe: what counts is their sum per tile.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void decoder_bwd_pair_kernel(DecFusedArgs p) {
  constexpr int H = 32, Hh = 16, S = 32;
  __shared__ __attribute__((aligned(16))) float dps[2][2][16 * DB_RS];   // [slot][t & 1]
  __shared__ __attribute__((aligned(16))) float hts[2][3][16 * DB_HS + 16];  // [slot][t % 3]; columns 32..35 of a row: (dx, dy, 1, 0)
  __shared__ __attribute__((aligned(16))) float dus[2][2][16 * DB_US];   // [slot][t & 1]
  __shared__ __attribute__((aligned(16))) float dds[2][2][2][16 * 2];    // [slot][t & 1][wave of the pair][row][component]
  __shared__ __attribute__((aligned(16))) float cAs[2][4][32];           // [u][fk][(r, gate) x (x, y)]: A of the lane's positions
  // the A fragments of the recurrence product, one row per (u, lane): in registers they were 32 values too many for two
  // waves per SIMD (the compiler parked them in scratch and fetched them back every step)
  __shared__ __attribute__((aligned(16))) float ahs[2][64][36];
  extern __shared__ __attribute__((aligned(16))) float tailw[];  // W_e2d[:, :EIN] (H x e2ld, dEnc only) | W1[:, H:] (Hh x 36)
  // Workgroups b and b + 256 share a CU (512 workgroups dealt round-robin over 256 CUs), i.e. generator gi and gi + n/2.  A
  // generator's tiles beyond its first round go to its FIRST workgroups; the upper half of the generators walks its
  // workgroups backwards, so that the second-round tiles of two generators that share CUs land on different ones (configs[1]:
  // 1,600 tiles on 1,024 slots -- 72 CUs of a generator pair carried four second-round tiles and 56 none).
  const int gi = blockIdx.x / p.NW, n_g = gridDim.x / p.NW;
  const int wi = (2 * gi >= n_g && n_g > 1) ? p.NW - 1 - (int)(blockIdx.x % p.NW) : (int)(blockIdx.x % p.NW);
  // (the wave index in a scalar register: slot, tile and the tile's record bases are then scalars and the per-lane part of
  // every saved-state address is a 32-bit offset -- as 64-bit per-lane pointers they did not fit)
  const int lane = threadIdx.x & 63, w = mg_wave(), fi = lane & 15, fk = lane >> 4;
  const int slot = w >> 1, u = w & 1, ub = 16 * u + 4 * fk;
  const long po = (long)gi * p.param_stride;
  const float* P = p.prep + (size_t)gi * p.prep_stride;
  const float* Whh = p.W_hh + po;
  const float* W1 = p.W1 + po;
  const int seg0 = p.seg[gi], seg1 = p.seg[gi + 1];
  const int ntiles = (seg1 - seg0 + 15) / 16;
  float* wp = p.wpart + (size_t)blockIdx.x * DF_WLEN;
  if (2 * wi >= ntiles) {  // no tile for this workgroup: an all-zero partial block
    for (int i = threadIdx.x; i < DF_WLEN; i += 256) wp[i] = 0.f;
    return;
  }
  const int tbase = dec_tile_base(p.seg, gi);
  // A operands.  Position pp = unit*4 + gate in the dPre tile <-> original gate row (pp & 3)*H + (pp >> 2).
  float Aw1[4], w2c[2][4];
  for (int e = threadIdx.x; e < 2 * 64 * 32; e += 256) {
    const int uu = e >> 11, ll = (e >> 5) & 63, ks = e & 31;
    const int pp = 16 * (ks >> 2) + 4 * (ll >> 4) + (ks & 3), m = (pp & 3) * H + (pp >> 2);
    ahs[uu][ll][ks] = Whh[(size_t)m * H + 16 * uu + (ll & 15)];
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) Aw1[ks] = W1[(size_t)(4 * fk + ks) * (H + S) + 16 * u + fi];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    w2c[0][q] = p.W2[po + 4 * fk + q];
    w2c[1][q] = p.W2[po + Hh + 4 * fk + q];
  }
  if (threadIdx.x < 256) {  // cAs[u'][fk'][(4 r + q) * 2 + c] = A[gate row q*H + 16 u' + 4 fk' + r][c]
    const int e = threadIdx.x, uu = e >> 7, ff = (e >> 5) & 3, k = e & 31, rr = k >> 3, qq = (k >> 1) & 3, c = k & 1;
    cAs[uu][ff][k] = P[prep_off_A(H) + (qq * H + 16 * uu + 4 * ff + rr) * 2 + c];
  }
  float* e2s = tailw;
  float* w1s = p.dEnc ? tailw + H * p.e2ld : tailw;
  const int IN = p.EIN + p.Z;
  if (p.dEnc)
    for (int i = threadIdx.x; i < H * p.EIN; i += 256) e2s[(i / p.EIN) * p.e2ld + i % p.EIN] = p.We2d[(size_t)(i / p.EIN) * IN + i % p.EIN];
  for (int i = threadIdx.x; i < Hh * S; i += 256) w1s[(i / S) * 36 + i % S] = W1[(size_t)(i / S) * (H + S) + H + i % S];
  f32x4 accW[4][2], accU = f32x4{0.f, 0.f, 0.f, 0.f}, accS = f32x4{0.f, 0.f, 0.f, 0.f};
  // dA (4H x 2) and dbias (4H) = dPre^T [dxdy | 1]: lane-local sums over the A fragments of the dW_hh product -- the lane
  // that feeds position 16 (4 u + i) + fi of rows ks + 4 fk into those products multiplies them by (dx, dy, 1) of the same
  // rows ((dx, dy) sit in the pad columns of the h tile's rows), 12 accumulators, folded over fk at the end.  (As sums over the
  // lane's own gate gradients -- 48 accumulators with four units per lane -- they pushed the step loop's operands into
  // scratch; as a third column tile of the product they cost 16 of 88 products per wave and step for 3 useful columns.)
  float accX[4][3];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    accW[a][0] = accW[a][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    accX[a][0] = accX[a][1] = accX[a][2] = 0.f;
  }
  float accW2[2][4], accb1[4], accb2[2] = {0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 4; ++q) accW2[0][q] = accW2[1][q] = accb1[q] = 0.f;
  const float* cAl = cAs[u][fk];
  // (explicitly GLOBAL pointers: behind a select of two pointers the compiler no longer knows the address space and issues
  //  FLAT loads, which count on lgkmcnt as well -- the LDS wait in front of the step's barrier then waits for them)
  typedef __attribute__((address_space(1))) float gfloat;
  const gfloat* ga_base = (const gfloat*)(p.gabs ? p.gabs : p.Din);
  const gfloat* gr_base = (const gfloat*)(p.grel ? p.grel : p.Din);
#ifdef DEC_PROFILE
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif

  for (int base = 2 * wi; base < ntiles; base += 2 * p.NW) {
    const int tile = base + slot;
    const bool tvalid = tile < ntiles;  // (the second slot of the last pair may be empty: it walks the last tile, masked)
    const int tl = tvalid ? tile : ntiles - 1;
    const int r = seg0 + tl * 16 + fi;
    const bool valid = tvalid && r < seg1;
    const int rc = r < seg1 ? r : seg1 - 1;
    const float vm = valid ? 1.f : 0.f;
    const int pos = p.row_pos[rc];
    const size_t tg = (size_t)(tbase + tl);
    f32x4 dh = {0.f, 0.f, 0.f, 0.f}, dc = {0.f, 0.f, 0.f, 0.f}, cc, dq = {0.f, 0.f, 0.f, 0.f};
    float dd0 = 0.f, dd1 = 0.f, s0 = 0.f, s1 = 0.f;
    f32x4 n_g[4], n_av;
    float2 n_ch[4], n_din, n_ga, n_gr;
    const float gam = p.gabs ? 1.f : 0.f, grm = p.grel ? 1.f : 0.f;
    // the tile's records (scalar bases) and this lane's offsets inside a step's record
    const float* gt_t = p.Gt + tg * DEC_GT_PAD + tg * p.T * (H * 64);
    const float* cs_t = p.Cs + tg * DEC_CS_PAD + tg * (p.T + 1) * (H * 32);
    const float* av_t = p.Aact + tg * p.T * 256;
    const float* din_t = p.Din + tg * p.T * 32;
    const int og = (ub * 16 + fi) * 4, oc = (ub * 16 + fi) * 2, oa = (fk * 16 + fi) * 4, od = fi * 2;
    const int oga = p.gabs ? pos * 2 : od, ogr = p.grel ? pos * 2 : od;
    auto fetch = [&](int t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        n_g[q] = *reinterpret_cast<const f32x4*>(gt_t + t * (H * 64) + og + q * 64);
        n_ch[q] = *reinterpret_cast<const float2*>(cs_t + t * (H * 32) + oc + q * 32);  // (c_{t-1}, h_{t-1})
      }
      n_av = *reinterpret_cast<const f32x4*>(av_t + t * 256 + oa);
      const float* din = din_t + t * 32;
      n_din = *reinterpret_cast<const float2*>(din + od);
      // (a missing output gradient: the ADDRESS is redirected to the step's input record -- base chosen once per launch, per-lane
      // offset once per tile -- and the value masked where it is consumed: no branch, no select on a loaded value)
      const gfloat* pa = ga_base + (p.gabs ? (size_t)t * p.Rout * 2 : tg * p.T * 32 + t * 32) + oga;
      const gfloat* pr = gr_base + (p.grel ? (size_t)t * p.Rout * 2 : tg * p.T * 32 + t * 32) + ogr;
      n_ga = float2{pa[0], pa[1]};  // (one 8-byte load each)
      n_gr = float2{pr[0], pr[1]};
    };
    lds_barrier();  // the previous tiles' last LDS reads are done
    {
      f32x4 hT;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float2 ch = *reinterpret_cast<const float2*>(cs_t + p.T * (H * 32) + oc + q * 32);
        cc[q] = ch.x;
        hT[q] = ch.y;
      }
      *reinterpret_cast<f32x4*>(&hts[slot][p.T % 3][fi * DB_HS + ub]) = hT;  // h_{T-1}
    }
    fetch(p.T - 1);

    for (int t = p.T - 1; t >= 0; --t) {
      float* dpw = dps[slot][t & 1];
      float* duw = dus[slot][t & 1];
      float* htw = hts[slot][t % 3];              // receives h_{t-1}
      const float* htc = hts[slot][(t + 1) % 3];  // holds h_t
      float* ddw = dds[slot][t & 1][0];
      // ONE register set for the saved activations: the next step's are asked for behind this step's gate arithmetic (their
      // last reader) and arrive under its products -- a second set (fetched a whole step ahead, as the four-wave kernel does)
      // is 34 registers this kernel does not have
      f32x4(&c_g)[4] = n_g;
      float2(&c_ch)[4] = n_ch;
      const f32x4 c_av = n_av;
      const float2 c_din = n_din, c_ga = n_ga, c_gr = n_gr;
      DP_T(0);  // 0: between steps / tile prologue
#ifdef DEC_PROBE_LOADS  // measurement build: the step's loads and nothing else
      dh += (c_g[0] + c_g[1]) + (c_g[2] + c_g[3]) + c_av;
      dd0 += c_ch[0].x + c_ch[1].y + c_ch[2].x + c_ch[3].y + c_din.x + c_ga.x + c_gr.y;
      continue;
#endif
      s0 = fmaf(c_ga.x, gam, s0); s1 = fmaf(c_ga.y, gam, s1);
      const float g0 = (s0 + dd0 + c_gr.x * grm) * vm, g1 = (s1 + dd1 + c_gr.y * grm) * vm;
      f32x4 du;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        du[q] = fmaf(w2c[0][q], g0, w2c[1][q] * g1) * (c_av[q] > 0.f ? 1.f : 0.01f);
        accW2[0][q] = fmaf(g0, c_av[q], accW2[0][q]);
        accW2[1][q] = fmaf(g1, c_av[q], accW2[1][q]);
      }
      dq += du;
      accb2[0] += g0; accb2[1] += g1;
      if (u == 0) *reinterpret_cast<f32x4*>(&duw[fi * DB_US + 4 * fk]) = du;
      *reinterpret_cast<f32x4*>(&htw[fi * DB_HS + ub]) = f32x4{c_ch[0].y, c_ch[1].y, c_ch[2].y, c_ch[3].y};
      if (u == 0 && fk == 0) *reinterpret_cast<f32x4*>(&htw[fi * DB_HS + 32]) = f32x4{c_din.x, c_din.y, 1.f, 0.f};
      // dh += W1[:, :H]^T du  (rows 4 fk + r of the M tile are this lane's units)
      f32x4 a1 = dh, a2 = f32x4{0.f, 0.f, 0.f, 0.f};
      a1 = MFMA16(Aw1[0], du[0], a1);
      a2 = MFMA16(Aw1[1], du[1], a2);
      a1 = MFMA16(Aw1[2], du[2], a1);
      a2 = MFMA16(Aw1[3], du[3], a2);
      DP_T(1);  // 1: step head (loads issued, du, W1 products)
      float dp0 = 0.f, dp1 = 0.f;  // this lane's share of d dxdy_t = A^T dPre
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float dhv = a1[q] + a2[q];
        const float gi_ = c_g[q][0], gf = c_g[q][1], gg = c_g[q][2], go = c_g[q][3];
        const float cprev = c_ch[q].x;  // slot 0 of Cs holds c_{-1} = 0
        const float tc = mg_tanh(cc[q]);
        const float dO = dhv * tc;
        const float dcv = fmaf(dhv * go, 1.f - tc * tc, dc[q]);
        f32x4 dp;
        dp[0] = dcv * gg * gi_ * (1.f - gi_) * vm;
        dp[1] = dcv * cprev * gf * (1.f - gf) * vm;
        dp[2] = dcv * gi_ * (1.f - gg * gg) * vm;
        dp[3] = dO * go * (1.f - go) * vm;
        dc[q] = dcv * gf;
        cc[q] = cprev;
        *reinterpret_cast<f32x4*>(&dpw[fi * DB_RS + (ub + q) * 4]) = dp;
        const f32x4 ca = *reinterpret_cast<const f32x4*>(cAl + 8 * q), cb = *reinterpret_cast<const f32x4*>(cAl + 8 * q + 4);
        dp0 = fmaf(dp[0], ca[0], fmaf(dp[1], ca[2], fmaf(dp[2], cb[0], fmaf(dp[3], cb[2], dp0))));
        dp1 = fmaf(dp[0], ca[1], fmaf(dp[1], ca[3], fmaf(dp[2], cb[1], fmaf(dp[3], cb[3], dp1))));
      }
      fetch(t > 0 ? t - 1 : 0);  // (step 0 re-reads its own slots: no branch between the loads)
      dp0 = quarters_sum(dp0);
      dp1 = quarters_sum(dp1);
      if (fk == 0) *reinterpret_cast<float2*>(&ddw[u * 32 + fi * 2]) = float2{dp0, dp1};
      DP_T(2);  // 2: gate arithmetic, tile writes
      lds_barrier();
      DP_T(3);  // 3: barrier
      {
        const float2 da = *reinterpret_cast<const float2*>(&ddw[fi * 2]), db = *reinterpret_cast<const float2*>(&ddw[32 + fi * 2]);
        dd0 = da.x + db.x;
        dd1 = da.y + db.y;
      }
      // dh_{t-1} (own 16 units) = W_hh^T dPre^T, K = 128 gate rows in tile-position order
      f32x4 acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      int aho = lane * 36;  // (read here, every step -- not hoisted into registers that are not there; the OFFSET is laundered:
      asm volatile("" : "+v"(aho));  //  a laundered pointer would be read with FLAT loads, which count on vmcnt like the prefetch)
      const float* ahl = &ahs[u][0][0] + aho;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(&dpw[fi * DB_RS + 16 * j + 4 * fk]);
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(ahl + 4 * j);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = MFMA16(a4[q], b4[q], acc[q]);
      }
      DP_T(4);  // 4: recurrence products
      // weight gradients, K = the 16 tile rows: dW_hh position tiles 4 u .. 4 u + 3 x two column tiles
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        float a[4], bv[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = dpw[(ks + 4 * fk) * DB_RS + 16 * (4 * u + i) + fi];
#pragma unroll
        for (int n = 0; n < 2; ++n) bv[n] = htw[(ks + 4 * fk) * DB_HS + 16 * n + fi];
        const float2 dxy = *reinterpret_cast<const float2*>(&htw[(ks + 4 * fk) * DB_HS + 32]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int n = 0; n < 2; ++n) accW[i][n] = MFMA16(a[i], bv[n], accW[i][n]);
          accX[i][0] = fmaf(a[i], dxy.x, accX[i][0]);
          accX[i][1] = fmaf(a[i], dxy.y, accX[i][1]);
          accX[i][2] += a[i];
        }
        // dW1[:, :H] += du^T h_t : wave u -> column tile u
        accU = MFMA16(duw[(ks + 4 * fk) * DB_US + fi], htc[(ks + 4 * fk) * DB_HS + 16 * u + fi], accU);
      }
      dh = (acc[0] + acc[1]) + (acc[2] + acc[3]);
      DP_T(5);  // 5: weight-gradient products
    }
    // ---- per-row outputs of this tile: dH0, dQ, d(enc_h row), d(social row) ----
    const int lane_t = mg_lane(), fi_t = lane_t & 15, fk_t = lane_t >> 4, ub_t = 16 * u + 4 * fk_t;  // (afresh: see the tail)
    const int r_t = seg0 + tl * 16 + fi_t;
    const bool valid_t = tvalid && r_t < seg1;
    float* h0t = hts[slot][2];  // h_{-1} slot of the t = 0 step is hts[0]; hts[2] held h_1 (last read at t = 1)
    if (valid_t) *reinterpret_cast<f32x4*>(p.dH0 + (size_t)r_t * H + ub_t) = dh;
    *reinterpret_cast<f32x4*>(&h0t[fi_t * DB_HS + ub_t]) = dh;
    if (u == 0 && valid_t) *reinterpret_cast<f32x4*>(p.dQ + (size_t)r_t * Hh + 4 * fk_t) = dq;
#pragma unroll
    for (int q = 0; q < 4; ++q) accb1[q] += dq[q];
    if (p.SocR) {
      // dW1[:, H:] += dQ^T SocR over the 16 rows of the tile: wave u -> column tile u
      float* dqs = dus[slot][1];
      if (u == 0) *reinterpret_cast<f32x4*>(&dqs[fi_t * DB_US + 4 * fk_t]) = dq;
      lds_barrier();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int row = ks + 4 * fk_t;
        const int rr = min(seg0 + tl * 16 + row, seg1 - 1);  // (rows past the segment end: dQ is zero there)
        accS = MFMA16(dqs[row * DB_US + fi_t], p.SocR[(size_t)rr * S + 16 * u + fi_t], accS);
      }
    }
    {  // dSocR^T [S x rows] = W1[:, H:]^T dQ^T : wave u -> social columns 16 u .. 16 u + 15
      f32x4 ds = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) ds = MFMA16(w1s[(4 * fk_t + ks) * 36 + 16 * u + fi_t], dq[ks], ds);
      if (valid_t) *reinterpret_cast<f32x4*>(p.dSocR + (size_t)r_t * S + 16 * u + 4 * fk_t) = ds;
    }
    if (p.dEnc) {
      lds_barrier();
      // dEnc^T [EIN x rows] = W_e2d[:, :EIN]^T dH0^T : wave u -> 16-column tiles u, u + 2, ...
      const f32x4 ha = *reinterpret_cast<const f32x4*>(&h0t[fi_t * DB_HS + 8 * fk_t]);
      const f32x4 hb = *reinterpret_cast<const f32x4*>(&h0t[fi_t * DB_HS + 8 * fk_t + 4]);
      for (int ct = u; ct * 16 < p.EIN; ct += 2) {
        const int col = 16 * ct + fi_t;
        const bool cin = col < p.EIN;
        f32x4 de = f32x4{0.f, 0.f, 0.f, 0.f}, de2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          de = MFMA16(cin ? e2s[(8 * fk_t + ks) * p.e2ld + col] : 0.f, ha[ks], de);
          de2 = MFMA16(cin ? e2s[(8 * fk_t + 4 + ks) * p.e2ld + col] : 0.f, hb[ks], de2);
        }
        de += de2;
        if (valid_t && 16 * ct + 4 * fk_t + 3 < p.EIN) *reinterpret_cast<f32x4*>(p.dEnc + (size_t)r_t * p.EIN + 16 * ct + 4 * fk_t) = de;
      }
    }
  }

  // (lane indices afresh -- v_mbcnt -- for the tail: what the prologue derived from them is not carried through the step loop)
  const int lane_e = mg_lane(), fi_e = lane_e & 15, fk_e = lane_e >> 4, ub_e = 16 * u + 4 * fk_e;
  // ---- this workgroup's partial block: the two slots meet in an LDS image of it (slot 0 writes, slot 1 adds) ----
  lds_barrier();
  float* blk = &dps[0][0][0];  // DF_WLEN floats (the dPre tiles are done with)
  static_assert(sizeof(dps) >= DF_WLEN * sizeof(float), "partial block image");
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    if (slot == round) {
      // accW[i][n][q] of lane_e (fi_e, fk_e): tile position pp = 16 (4 u + i) + 4 fk_e + q, column 16 n + fi_e
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int pp = 16 * (4 * u + i) + 4 * fk_e + q, m = (pp & 3) * H + (pp >> 2);
          if (round == 0) {
            blk[m * H + fi_e] = accW[i][0][q];
            blk[m * H + 16 + fi_e] = accW[i][1][q];
          } else {
            blk[m * H + fi_e] += accW[i][0][q];
            blk[m * H + 16 + fi_e] += accW[i][1][q];
          }
        }
      // dA / dbias: the lane's sums for position 16 (4 u + i) + fi, folded over the four row groups fk
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v0 = quarters_sum(accX[i][0]), v1 = quarters_sum(accX[i][1]), v2 = quarters_sum(accX[i][2]);
        if (fk_e == 0) {
          const int pp = 16 * (4 * u + i) + fi_e, m = (pp & 3) * H + (pp >> 2);
          if (round == 0) { blk[DF_OFF_A + m * 2] = v0; blk[DF_OFF_A + m * 2 + 1] = v1; blk[DF_OFF_B + m] = v2; }
          else { blk[DF_OFF_A + m * 2] += v0; blk[DF_OFF_A + m * 2 + 1] += v1; blk[DF_OFF_B + m] += v2; }
        }
      }
      // W1h / W1s: accU / accS [q] of lane_e (fi_e, fk_e): row 4 fk_e + q, column 16 u + fi_e
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int o = (4 * fk_e + q) * 32 + 16 * u + fi_e;
        if (round == 0) { blk[DF_OFF_W1 + o] = accU[q]; blk[DF_OFF_W1S + o] = accS[q]; }
        else { blk[DF_OFF_W1 + o] += accU[q]; blk[DF_OFF_W1S + o] += accS[q]; }
      }
      if (u == 0) {  // lane_e-local accumulators (W2, b1, b2): folded over the 16 tile rows in a fixed tree
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v0 = accW2[0][q], v1 = accW2[1][q], v2 = accb1[q];
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {
            v0 += __shfl_xor(v0, o, 64);
            v1 += __shfl_xor(v1, o, 64);
            v2 += __shfl_xor(v2, o, 64);
          }
          if (fi_e == 0) {
            if (round == 0) { blk[DF_OFF_W2 + 4 * fk_e + q] = v0; blk[DF_OFF_W2 + Hh + 4 * fk_e + q] = v1; blk[DF_OFF_B1 + 4 * fk_e + q] = v2; }
            else { blk[DF_OFF_W2 + 4 * fk_e + q] += v0; blk[DF_OFF_W2 + Hh + 4 * fk_e + q] += v1; blk[DF_OFF_B1 + 4 * fk_e + q] += v2; }
          }
        }
        float b0 = accb2[0], b1v = accb2[1];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          b0 += __shfl_xor(b0, o, 64);
          b1v += __shfl_xor(b1v, o, 64);
        }
        if (lane_e == 0) {
          if (round == 0) { blk[DF_OFF_B2] = b0; blk[DF_OFF_B2 + 1] = b1v; blk[DF_OFF_B2 + 2] = 0.f; blk[DF_OFF_B2 + 3] = 0.f; }
          else { blk[DF_OFF_B2] += b0; blk[DF_OFF_B2 + 1] += b1v; }
        }
      }
    }
    lds_barrier();
  }
  for (int i = threadIdx.x; i < DF_WLEN; i += 256) wp[i] = blk[i];
#ifdef DEC_PROFILE
  DP_T(6);  // 6: tile epilogues and the partial block
  if (blockIdx.x == 0 && threadIdx.x == 0)
    printf("DECPROF between %llu head %llu gates %llu barrier %llu recurrence %llu wgrad %llu rest %llu\n", prof[0], prof[1], prof[2], prof[3], prof[4], prof[5], prof[6]);
#endif
}

// dst[ped][c] (+)= sum_k src[inv[k*b + ped]][c]
__global__ void gather_sum_kernel(const float* __restrict__ src, int lds_, const int* __restrict__ inv, float* dst,
                                  int ldd, int b, int K, int ncols, int accumulate) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)b * ncols) return;
  int ped = (int)(i / ncols), c = (int)(i % ncols);
  // four rows at a time: their indices first, then the four gathers - a one-by-one walk was K times two dependent
  // round trips (index, then row)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 3 < K; k += 4) {
    const int i0 = inv[(size_t)k * b + ped], i1 = inv[(size_t)(k + 1) * b + ped], i2 = inv[(size_t)(k + 2) * b + ped],
              i3 = inv[(size_t)(k + 3) * b + ped];
    s0 += src[(size_t)i0 * lds_ + c];
    s1 += src[(size_t)i1 * lds_ + c];
    s2 += src[(size_t)i2 * lds_ + c];
    s3 += src[(size_t)i3 * lds_ + c];
  }
  for (; k < K; ++k) s0 += src[(size_t)inv[(size_t)k * b + ped] * lds_ + c];
  const float s = (s0 + s1) + (s2 + s3);
  float* d = dst + (size_t)ped * ldd + c;
  *d = accumulate ? (*d + s) : s;
}

// WT[k][n] = W[n][k]
// The per-pedestrian tail of the rollout adjoint in ONE launch (it was gather_sum -> linear_bwd_data -> gather_sum, three
// dependent launches on the critical chain of the generator step):
//   dQe[ped]  = sum_k dH0[inv[k b + ped]]                       (H = 32 columns; also the operand of the e2d weight gradient)
//   dEnc[ped] = dQe[ped] . W_e2d[:, :EIN]                       (EIN columns)
//   dEnc[ped][EIN - S ..] += sum_k dSocR[inv[k b + ped]]        (the social block is the last S columns of enc_h; S = 0: none)
// A workgroup takes 8 pedestrians: thread (p, c) first folds column c of the K rows of pedestrian p (four gathers in flight),
// the 8 x 32 block meets in LDS, then every thread owns columns c, c + 32, ... of its pedestrian's dEnc row.
__global__ __launch_bounds__(256) void rollout_ped_adjoint_kernel(const float* __restrict__ dH0, const float* __restrict__ dSocR,
                                                                  const int* __restrict__ inv, const float* __restrict__ W,
                                                                  int ldw, float* __restrict__ dQe, float* __restrict__ dEnc,
                                                                  int ld_enc, int b, int K, int EIN, int S) {
  constexpr int H = 32;
  __shared__ float q[8][H + 1];
  const int p = threadIdx.x >> 5, c = threadIdx.x & 31, ped = blockIdx.x * 8 + p;
  const bool ok = ped < b;
  const int pc = ok ? ped : b - 1;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  const bool soc = c < S;  // (S <= 32)
  int k = 0;
  for (; k + 3 < K; k += 4) {
    const int i0 = inv[(size_t)k * b + pc], i1 = inv[(size_t)(k + 1) * b + pc], i2 = inv[(size_t)(k + 2) * b + pc],
              i3 = inv[(size_t)(k + 3) * b + pc];
    s0 += dH0[(size_t)i0 * H + c];
    s1 += dH0[(size_t)i1 * H + c];
    s2 += dH0[(size_t)i2 * H + c];
    s3 += dH0[(size_t)i3 * H + c];
    if (soc) {
      t0 += dSocR[(size_t)i0 * S + c];
      t1 += dSocR[(size_t)i1 * S + c];
      t2 += dSocR[(size_t)i2 * S + c];
      t3 += dSocR[(size_t)i3 * S + c];
    }
  }
  for (; k < K; ++k) {
    const int i0 = inv[(size_t)k * b + pc];
    s0 += dH0[(size_t)i0 * H + c];
    if (soc) t0 += dSocR[(size_t)i0 * S + c];
  }
  const float s = (s0 + s1) + (s2 + s3), ts = (t0 + t1) + (t2 + t3);
  q[p][c] = s;
  if (ok) dQe[(size_t)ped * H + c] = s;
  __syncthreads();
  if (!ok) return;  // (a whole 32-lane group at a time: the groups of a wave do not shuffle across)
  for (int c0 = 0; c0 < EIN; c0 += 32) {  // (uniform trip count: the shuffle below wants all 32 lanes of the pedestrian)
    const int col = c0 + c, cc = col < EIN ? col : EIN - 1;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int h = 0; h < H; h += 2) {
      a0 = fmaf(q[p][h], W[(size_t)h * ldw + cc], a0);
      a1 = fmaf(q[p][h + 1], W[(size_t)(h + 1) * ldw + cc], a1);
    }
    // the social block is the last S columns: social column j = col - (EIN - S) was summed by lane j of this pedestrian
    const float tj = __shfl(ts, (col - (EIN - S)) & 31, 32);
    if (col < EIN) dEnc[(size_t)ped * ld_enc + col] = (a0 + a1) + (S > 0 && col >= EIN - S ? tj : 0.f);
  }
}

__global__ void transpose_kernel(const float* __restrict__ W, float* __restrict__ WT, int N, int K) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * K) return;
  int n = i / K, k = i % K;
  WT[(size_t)k * N + n] = W[i];
}

extern "C" {

int mggan_lstm_prep_size(int H, int S, int dec) { return prep_size(H, S, dec); }

int mggan_lstm_fold(const float* W_emb, const float* b_emb, const float* W_ih, const float* b_ih, const float* b_hh,
                    const float* W_hh, const float* W1, const float* b1, const float* W2, const float* b2,
                    long param_stride, int n_groups, int H, int E, int S, int dec, float* prep, int prep_stride,
                    hipStream_t stream) {
  MG_CHECK_ARG(W_emb && b_emb && W_ih && b_ih && b_hh && W_hh && prep, "lstm_fold: null pointer");
  MG_CHECK_ARG(!dec || (W1 && b1 && W2 && b2), "lstm_fold: decoder head pointers missing");
  MG_CHECK_ARG(prep_stride >= prep_size(H, S, dec), "lstm_fold: prep stride too small");
  FoldArgs a = {W_emb, b_emb, W_ih, b_ih, b_hh, W_hh, W1, b1, W2, b2, param_stride, prep, prep_stride, H, E, S, dec};
  MG_LAUNCH(lstm_fold_kernel, dim3(n_groups, 8), dim3(256), 0, stream, a);
  MG_LAUNCH_CHECK("lstm_fold");
  return MGGAN_OK;
}

int mggan_lstm_unfold_grads(const float* W_emb, const float* b_emb, const float* W_ih, float* dW_emb, float* db_emb,
                            float* dW_ih, float* db_ih, float* db_hh, long param_stride, int n_groups, int H, int E,
                            const float* dprep, int dprep_stride, hipStream_t stream) {
  MG_CHECK_ARG(W_emb && b_emb && W_ih && dW_emb && db_emb && dW_ih && db_ih && db_hh && dprep,
               "lstm_unfold_grads: null pointer");
  UnfoldArgs a = {W_emb, b_emb, W_ih, dW_emb, db_emb, dW_ih, db_ih, db_hh, param_stride, dprep, dprep_stride, H, E};
  MG_LAUNCH(lstm_unfold_kernel, dim3(n_groups, 12), dim3(256), 0, stream, a);
  MG_LAUNCH_CHECK("lstm_unfold_grads");
  return MGGAN_OK;
}

int mggan_lstm_encoder_fwd(const float* x, int T, int b, int H, const float* prep, float* hout, int ld_hout, float* Gt,
                           float* Cs, float* Hp, float* Din, hipStream_t stream) {
  MG_CHECK_ARG(x && prep && hout && T > 0 && b >= 0, "lstm_encoder_fwd: bad arguments");
  MG_CHECK_ARG(H == 32 || H == 64, "lstm_encoder_fwd: hidden size %d not built (32 or 64)", H);
  MG_CHECK_ARG((Gt == nullptr) == (Cs == nullptr) && (Gt == nullptr) == (Hp == nullptr) &&
                   (Gt == nullptr) == (Din == nullptr),
               "lstm_encoder_fwd: save buffers must be all set or all NULL");
  if (b == 0) return MGGAN_OK;
  SeqArgs p = {};
  p.R = b; p.T = T; p.b = b; p.prep = prep; p.x = x; p.hout = hout; p.ld_hout = ld_hout;
  p.Gt = Gt; p.Cs = Cs; p.Hp = Hp; p.Din = Din;
  static int valu = -1;  // MGGAN_LSTM_VALU=1: the lane-per-(row, unit) VALU kernel (A/B measurements)
  if (valu < 0) { const char* e = getenv("MGGAN_LSTM_VALU"); valu = e && e[0] == '1'; }
  if (valu) {
    if (H == 32) MG_LAUNCH((lstm_fwd_kernel<32>), dim3(cdiv(b, 8)), dim3(256), 0, stream, p);
    else MG_LAUNCH((lstm_fwd_kernel<64>), dim3(cdiv(b, 4)), dim3(256), 0, stream, p);
  } else {
    if (H == 32) MG_LAUNCH((lstm_fwd_mfma_kernel<32>), dim3(cdiv(b, 16)), dim3(256), 0, stream, p);
    else MG_LAUNCH((lstm_fwd_mfma_kernel<64>), dim3(cdiv(b, 16)), dim3(256), 0, stream, p);
  }
  MG_LAUNCH_CHECK("lstm_encoder_fwd");
  return MGGAN_OK;
}

int mggan_lstm_encoder_bwd(const float* dhT, int ld_dhT, int T, int b, int H, const float* W_hh, const float* prep,
                           const float* Gt, const float* Cs, float* dPre, hipStream_t stream) {
  MG_CHECK_ARG(dhT && W_hh && prep && Gt && Cs && dPre, "lstm_encoder_bwd: null pointer");
  MG_CHECK_ARG(H == 32 || H == 64, "lstm_encoder_bwd: hidden size %d not built (32 or 64)", H);
  if (b == 0) return MGGAN_OK;
  SeqBwdArgs p = {};
  p.R = b; p.T = T; p.W_hh = W_hh; p.Gt = Gt; p.Cs = Cs; p.dhT = dhT; p.ld_dhT = ld_dhT; p.dPre = dPre;
  static int valu = -1;  // MGGAN_LSTM_VALU=1: the lane-per-(row, unit) VALU kernels (A/B measurements)
  if (valu < 0) { const char* e = getenv("MGGAN_LSTM_VALU"); valu = e && e[0] == '1'; }
  const bool vec = (ld_dhT % 4 == 0) && ((size_t)dhT % 16 == 0);  // the matrix-core kernel reads dh_T as 16-byte quads
  // below ~4k trajectories the 16/32-row tiles leave most CUs without a workgroup (1,280 rows: 40-80 workgroups) and
  // the lane-per-(row, unit) kernel with its 4/8-row workgroups is as fast or faster (measured at 1,280: 17-23 us either way)
  // (MGGAN_LSTM_MFMA_MIN_B: measurement knob for the threshold.  Inside the configs[1] graph, three alternating pairs:
  //  1.429-1.433 ms with the lane-per-unit kernel, 1.423-1.432 with the matrix-core one -- within the noise)
  static int min_b = -1;
  if (min_b < 0) { const char* e = getenv("MGGAN_LSTM_MFMA_MIN_B"); min_b = e ? atoi(e) : 4096; }
  if (valu || !vec || b < min_b) {
    if (H == 32) MG_LAUNCH((lstm_bwd_kernel<32, 0>), dim3(cdiv(b, 8)), dim3(256), 0, stream, p);
    else MG_LAUNCH((lstm_bwd_kernel<64, 64>), dim3(cdiv(b, 4)), dim3(256), 0, stream, p);
  } else {
    if (H == 32) MG_LAUNCH((lstm_bwd_mfma_kernel<32>), dim3(cdiv(b, 32)), dim3(256), 0, stream, p);
    else MG_LAUNCH((lstm_bwd_mfma_kernel<64>), dim3(cdiv(b, 16)), dim3(256), 0, stream, p);
  }
  MG_LAUNCH_CHECK("lstm_encoder_bwd");
  return MGGAN_OK;
}

/* Q (b, 32) = be2d + We2d[:, :EIN] enc_h: the part of h0 = enc_h_to_dec_h([enc_h | noise]) (standard.py:247-252) that
 * the K rollout rows of a pedestrian share (W_e2d is one module for all generators).  We2d: (32, ldw) row-major. */
int mggan_decoder_e2d_shared(const float* enc_h, int ld_enc, int b, int EIN, const float* We2d, int ldw, const float* be2d,
                             float* Q, hipStream_t stream) {
  if (b == 0) return MGGAN_OK;
  MG_CHECK_ARG(enc_h && We2d && be2d && Q, "decoder_e2d_shared: null pointer");
  MG_CHECK_ARG(EIN > 0 && EIN % 16 == 0 && ld_enc % 4 == 0 && ldw % 4 == 0 && ldw >= EIN && (((size_t)enc_h) & 15) == 0 &&
                   (((size_t)We2d) & 15) == 0 && (((size_t)Q) & 15) == 0,
               "decoder_e2d_shared: bad layout (EIN %d, ld_enc %d, ldw %d)", EIN, ld_enc, ldw);
  MG_LAUNCH(e2d_shared_kernel, dim3(cdiv(cdiv(b, 16), 4)), dim3(256), 0, stream, enc_h, ld_enc, b, EIN, We2d, ldw, be2d, Q);
  MG_LAUNCH_CHECK("decoder_e2d_shared");
  return MGGAN_OK;
}

int mggan_decoder_rollout_fwd(int R, int T, int b, int H, int EIN, int Z, const float* prep, int prep_stride,
                              const int* seg, int n_gens, const int* row_ped, const int* row_slot, const int* row_pos,
                              const float* enc_h, int ld_enc, const float* noise, const float* soc, int ld_soc,
                              const float* xy0, const float* dxdy0, const float* We2d, const float* be2d,
                              float* out_abs, float* out_rel, int Rout, float* Gt, float* Cs, float* Din,
                              float* Aact, float* E2Din, float* SocR, const float* Qe, float* Nz, hipStream_t stream) {
  MG_CHECK_ARG(prep && seg && row_ped && row_slot && row_pos && enc_h && noise && soc && xy0 && dxdy0 && We2d && be2d &&
                   out_abs && out_rel,
               "decoder_rollout_fwd: null pointer");
  MG_CHECK_ARG(H == 32, "decoder_rollout_fwd: decoder_h_dim %d not built (32)", H);
  MG_CHECK_ARG(EIN % 4 == 0 && Z % 4 == 0 && ld_enc % 4 == 0 && ld_soc % 4 == 0 && n_gens > 0,
               "decoder_rollout_fwd: widths must be multiples of 4 (enc %d, noise %d)", EIN, Z);
  const bool s = Gt != nullptr;
  MG_CHECK_ARG(s == (Cs != nullptr) && s == (Din != nullptr) && s == (Aact != nullptr) &&
                   s == ((Qe ? Nz : E2Din) != nullptr) && s == (SocR != nullptr),
               "decoder_rollout_fwd: save buffers must be all set or all NULL");
  MG_CHECK_ARG(!Qe || EIN % 16 == 0, "decoder_rollout_fwd: the per-pedestrian part needs an encoder width that is a multiple of 16 (%d)", EIN);
  if (R == 0) return MGGAN_OK;
  DecFwdArgs p = {};
  p.T = T; p.b = b; p.Rout = Rout; p.EIN = EIN; p.Z = Z; p.ld_enc = ld_enc; p.ld_soc = ld_soc; p.seg = seg;
  p.prep = prep; p.prep_stride = prep_stride; p.row_ped = row_ped; p.row_slot = row_slot; p.row_pos = row_pos;
  p.enc_h = enc_h; p.noise = noise; p.soc = soc; p.xy0 = xy0; p.dxdy0 = dxdy0; p.We2d = We2d; p.be2d = be2d;
  p.out_abs = out_abs; p.out_rel = out_rel;
  p.Gt = Gt; p.Cs = Cs; p.Din = Din; p.Aact = Aact; p.E2Din = E2Din; p.SocR = SocR; p.Qe = Qe; p.Nz = Nz;
  // one workgroup per 16-row tile, NW workgroups per generator.  The split of R between the generators is drawn on
  // the device (the PM network's categorical samples): below the cap every generator gets room for ALL tiles, so that
  // an uneven draw does not send some workgroups through a second tile; a workgroup without a tile leaves at once.
  const int per_gen = cdiv(R, 16) + 1;
  p.NW = per_gen < 1 ? 1 : (per_gen > 2048 / n_gens ? (2048 / n_gens > 0 ? 2048 / n_gens : 1) : per_gen);
  // Two kernels.  Four waves per tile (a tile's 32 units over four waves, LDS exchange + barrier per step): the shorter
  // dependency chain per tile, best while the launch is a latency chain (8,192 rows: 34 us against 67).  One wave per tile
  // (decoder_fwd_wave_kernel: nothing leaves the registers, 80 instead of 96 MFMAs per 16 rows and step): best once there
  // are several tiles per SIMD -- alone on the GPU at 163,840 rows 0.31 ms against 0.41-0.54 without the saved state, 0.45
  // against 0.51 with it.  MGGAN_DEC_FWD = 4 | 1 forces one of them (A/B measurements).
  static int force = -1;
  if (force < 0) { const char* e = getenv("MGGAN_DEC_FWD"); force = e && (e[0] == '4' || e[0] == '1') ? e[0] - '0' : 0; }
  // (threshold: 16,384 rows -- the 25,600-row launch of the generator step at configs[1] inside the graph, two alternating
  //  pairs: 1.407-1.409 ms with the wave kernel, 1.413-1.416 with the four-wave one; MGGAN_DEC_FWD_MIN moves it)
  static int wave_min = -1;
  if (wave_min < 0) { const char* e = getenv("MGGAN_DEC_FWD_MIN"); wave_min = e ? atoi(e) : 16384; }
  const bool wave = force ? force == 1 : R >= wave_min;
  if (!wave) {
    MG_LAUNCH(decoder_fwd_mfma_kernel, dim3(n_gens * p.NW), dim3(256), 0, stream, p);
  } else {
    // a wave per tile: four tiles per workgroup, the same persistent-grid rule in workgroups of four tiles
    const int per_gen4 = cdiv(cdiv(R, 16) + 1, 4);
    const int cap = 1024 / n_gens > 0 ? 1024 / n_gens : 1;
    p.NW = per_gen4 < 1 ? 1 : (per_gen4 > cap ? cap : per_gen4);
    MG_LAUNCH(decoder_fwd_wave_kernel, dim3(n_gens * p.NW), dim3(256), 0, stream, p);
  }
  MG_LAUNCH_CHECK("decoder_rollout_fwd");
  return MGGAN_OK;
}

int mggan_decoder_save_pads(int* gt_pad, int* cs_pad) {
  *gt_pad = DEC_GT_PAD; *cs_pad = DEC_CS_PAD;
  return MGGAN_OK;
}

int mggan_decoder_bwd_fused_layout(int* wlen, int* off_A, int* off_bias, int* off_W1, int* off_b1, int* off_W2,
                                   int* off_b2, int* off_W1s) {
  *wlen = DF_WLEN; *off_A = DF_OFF_A; *off_bias = DF_OFF_B; *off_W1 = DF_OFF_W1; *off_b1 = DF_OFF_B1;
  *off_W2 = DF_OFF_W2; *off_b2 = DF_OFF_B2; *off_W1s = DF_OFF_W1S;
  return MGGAN_OK;
}

int mggan_decoder_rollout_bwd_fused(int n_gens, int NW, int T, int H, int EIN, int Z, const int* seg, const int* row_pos,
                                    const float* W_hh, const float* W1, const float* W2, long param_stride,
                                    const float* We2d, const float* prep, int prep_stride, const float* Gt,
                                    const float* Cs, const float* Din,
                                    const float* Aact, const float* gabs, const float* grel, int Rout, float* dH0,
                                    float* dQ, float* dEnc, float* dSocR, float* wpart, const float* SocR,
                                    hipStream_t stream) {
  MG_CHECK_ARG(seg && row_pos && W_hh && W1 && W2 && We2d && prep && Gt && Cs && Din && Aact && dH0 && dQ &&
                   dSocR && wpart,
               "decoder_rollout_bwd_fused: null pointer");
  MG_CHECK_ARG(H == 32 && NW > 0 && n_gens > 0, "decoder_rollout_bwd_fused: decoder_h_dim %d not built (32)", H);
  DecFusedArgs p = {};
  p.T = T; p.NW = NW; p.Rout = Rout; p.EIN = EIN; p.Z = Z; p.seg = seg; p.row_pos = row_pos;
  p.W_hh = W_hh; p.W1 = W1; p.W2 = W2; p.We2d = We2d; p.param_stride = param_stride; p.prep = prep;
  p.prep_stride = prep_stride; p.Gt = Gt; p.Cs = Cs; p.Din = Din; p.Aact = Aact;
  p.gabs = gabs; p.grel = grel; p.dH0 = dH0; p.dQ = dQ; p.dEnc = dEnc; p.dSocR = dSocR; p.wpart = wpart;
  p.SocR = SocR;
  p.e2ld = EIN + ((10 - EIN % 8) % 8);  // == 2 mod 8: the A-fragment reads (row 8 fk + ks) hit four 16-bank groups
  const size_t dyn = sizeof(float) * ((size_t)H * p.e2ld + (H / 2) * 36);
  MG_CHECK_ARG(dyn <= 64 * 1024, "decoder_rollout_bwd_fused: encoder width %d too large for the staged epilogue", EIN);
  // (One wave per tile, as in decoder_fwd_wave_kernel, was written and measured for this kernel too -- correct, ~500 registers,
  //  one wave per SIMD: 0.87 vs 0.79 ms at 163,840 rows and 180 vs 143 us at 25,600; with nothing else resident on its
  //  SIMD a wave exposes every dependent latency of the gate arithmetic.  Not kept)
  // Two waves per tile (decoder_bwd_pair_kernel) from 4,096 rollout rows on: 0.64 vs 0.81 ms at 163,840 rows, 151 vs 155 us
  // at 25,600 (1,600 tiles on 1,024 tile slots: two rounds); below that the four-wave kernel's one tile per workgroup
  // spreads a few dozen tiles over more CUs.  MGGAN_DEC_BWD=2 / 4 forces one (A/B).
  static int force = -1;
  if (force < 0) { const char* e = getenv("MGGAN_DEC_BWD"); force = e ? (e[0] == '4' ? 4 : e[0] == '2' ? 2 : 0) : 0; }
  const bool pair = force == 2 || (force == 0 && Rout >= 4096);
  if (pair) {
    const size_t dyn2 = sizeof(float) * ((dEnc ? (size_t)H * p.e2ld : 0) + (H / 2) * 36);
    MG_CHECK_ARG(dyn2 <= 64 * 1024, "decoder_rollout_bwd_fused: encoder width %d too large for the staged epilogue", EIN);
    MG_LAUNCH(decoder_bwd_pair_kernel, dim3(n_gens * NW), dim3(256), dyn2, stream, p);
  } else {
    MG_LAUNCH(decoder_bwd_mfma_kernel, dim3(n_gens * NW), dim3(256), dyn, stream, p);
  }
  MG_LAUNCH_CHECK("decoder_rollout_bwd_fused");
  return MGGAN_OK;
}

int mggan_gather_sum(const float* src, int ld_src, const int* inv, float* dst, int ld_dst, int b, int K, int ncols,
                     int accumulate, hipStream_t stream) {
  MG_CHECK_ARG(src && inv && dst, "gather_sum: null pointer");
  long n = (long)b * ncols;
  if (n == 0) return MGGAN_OK;
  MG_LAUNCH(gather_sum_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, src, ld_src, inv, dst, ld_dst, b, K,
                     ncols, accumulate);
  MG_LAUNCH_CHECK("gather_sum");
  return MGGAN_OK;
}

int mggan_rollout_ped_adjoint(const float* dH0, const float* dSocR, const int* inv, const float* W_e2d, int ldw, float* dQe,
                              float* dEnc, int ld_enc, int b, int K, int EIN, int S, hipStream_t stream) {
  MG_CHECK_ARG(dH0 && inv && W_e2d && dQe && dEnc && (S == 0 || dSocR), "rollout_ped_adjoint: null pointer");
  MG_CHECK_ARG(b >= 0 && K >= 1 && EIN >= 1 && S >= 0 && S <= 32 && S <= EIN && ldw >= EIN && ld_enc >= EIN,
               "rollout_ped_adjoint: bad sizes (b %d, K %d, EIN %d, S %d)", b, K, EIN, S);
  if (b == 0) return MGGAN_OK;
  MG_LAUNCH(rollout_ped_adjoint_kernel, dim3(cdiv(b, 8)), dim3(256), 0, stream, dH0, dSocR, inv, W_e2d, ldw, dQe, dEnc,
                     ld_enc, b, K, EIN, S);
  MG_LAUNCH_CHECK("rollout_ped_adjoint");
  return MGGAN_OK;
}

int mggan_transpose(const float* W, float* WT, int N, int K, hipStream_t stream) {
  MG_CHECK_ARG(W && WT, "transpose: null pointer");
  MG_LAUNCH(transpose_kernel, dim3(cdiv((long)N * K, 256)), dim3(256), 0, stream, W, WT, N, K);
  MG_LAUNCH_CHECK("transpose");
  return MGGAN_OK;
}

}  // extern "C"
