// Generic f32 GEMM family for the small dense layers of MG-GAN (linear forward,
// input-gradient, weight-gradient) plus the elementwise/reduction helpers they
// need.  Replaces the implicit ATen/cuBLAS launches behind nn.Linear on the
// reference hot path (reference call sites
// mggan/utils.py:134-149 make_mlp, model/modules/discriminators.py:46-56,76-108,
// model/modules/standard.py:91-105).
//
// One LDS-tiled, register-blocked kernel: 64x64 output tile per 256-thread
// workgroup (4 waves), 4x4 outputs per lane, K chunked by 16 through LDS in a
// k-major image so that every inner-loop LDS read is a conflict-free
// ds_read_b128.  Operands may be stored k-major or k-minor, which covers
//   forward      Y  = X  W^T      (A k-minor, B k-minor)
//   input grad   dX = dZ W        (A k-minor, B k-major)
//   weight grad  dW = dZ^T X      (A k-major, B k-major; split over rows, two-phase
//                                  deterministic reduction, bias grad via a ones column)
#include "common.h"
#include "../../include/mggan_hip.h"

#define BM 64
#define BN 64
#define BK 32
#define LDT 68  // padded LDS leading dimension (multiple of 4 -> aligned b128 reads)

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K;        // C is MxN, reduction length K
  int lda, ldb, ldc;
  int act;
  float slope;
  int accumulate;     // C += result
  // split-K / grouped (weight-grad) mode
  int splits;         // >0: write partial sums P[z][M*Naug] instead of C
  int ones_col;       // B gets a virtual column N-1 == 1.0 (bias grad); real N-1 columns
  const int* seg;     // optional group row offsets (n_groups+1), scaled by seg_scale
  int seg_scale;
  int n_groups;
  // optional fused activation derivative on the A operand: A(row, n) *= act'(Yact[row*ld_yact + n])
  const float* Yact;
  int ld_yact, act_a;
  float slope_a;
};

// Global -> register fetch of one BMxBK (A) and BNxBK (B) tile; 8 + 8 elements per lane.
template <bool A_KM, bool B_KM, bool ACT>
__device__ __forceinline__ void gemm_fetch(const GemmArgs& g, int m0, int n0, int k0, int k_end, float ra[8],
                                           float rb[8]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = tid + 256 * i;
    {
      int kd, m;
      if (A_KM) { kd = e >> 6; m = e & 63; } else { m = e >> 5; kd = e & 31; }
      const int gk = k0 + kd, gm = m0 + m;
      float v = 0.f;
      if (gk < k_end && gm < g.M) {
        v = A_KM ? g.A[(size_t)gk * g.lda + gm] : g.A[(size_t)gm * g.lda + gk];
        if (ACT) {  // (row, feature) = (gk, gm) for k-major A (weight grad), (gm, gk) for k-minor A (input grad)
          const float y = A_KM ? g.Yact[(size_t)gk * g.ld_yact + gm] : g.Yact[(size_t)gm * g.ld_yact + gk];
          v *= mg_act_grad_from_out(y, g.act_a, g.slope_a);
        }
      }
      ra[i] = v;
    }
    {
      int kd, n;
      if (B_KM) { kd = e >> 6; n = e & 63; } else { n = e >> 5; kd = e & 31; }
      const int gk = k0 + kd, gn = n0 + n;
      float v = 0.f;
      if (gk < k_end && gn < g.N) {
        if (g.ones_col && gn == g.N - 1) v = 1.0f;
        else v = B_KM ? g.B[(size_t)gk * g.ldb + gn] : g.B[(size_t)gn * g.ldb + gk];
      }
      rb[i] = v;
    }
  }
}

// LDS images for the MFMA fragments (v_mfma_f32_16x16x4_f32: lane l holds A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]):
//  * k-major operand (stored [k][m] in memory): LDS [k][LDK], LDK = 80 (== 16 mod 32): the two k rows a
//    32-lane half reads land on disjoint bank halves; stores are row-contiguous (conflict-free);
//  * k-minor operand (stored [m][k] in memory): LDS [m][LDM], LDM = 34: fragment reads hit bank 2i+k
//    (conflict-free per half), stores are row-contiguous.  No transposing stores anywhere.
#define LDK 80
#define LDM 34
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool KM>
__device__ __forceinline__ void lds_put(float* T, int e, float v) {
  if (KM) T[(e >> 6) * LDK + (e & 63)] = v;   // (kd = e>>6, m = e&63)
  else T[(e >> 5) * LDM + (e & 31)] = v;      // (m = e>>5, kd = e&31)
}
template <bool KM>
__device__ __forceinline__ float lds_frag(const float* T, int m, int k) {  // element (m, k) of the tile
  return KM ? T[k * LDK + m] : T[m * LDM + k];
}

template <bool A_KM, bool B_KM, bool ACT>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int bx, const int by, const int bz) {
  __shared__ __attribute__((aligned(16))) float As[A_KM ? BK * LDK : BM * LDM];
  __shared__ __attribute__((aligned(16))) float Bs[B_KM ? BK * LDK : BN * LDM];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = (w >> 1) * 32, wn = (w & 1) * 32;   // this wave's 32x32 quadrant of the 64x64 tile
  const int fi = lane & 15, fk = lane >> 4;
  const int m0 = by * BM, n0 = bx * BN;
  int k_begin = 0, k_end = g.K;
  if (g.splits > 0) {
    int z = bz;
    int grp = z / g.splits, sp = z % g.splits;
    int r0 = 0, r1 = g.K;
    if (g.seg) { r0 = g.seg[grp] * g.seg_scale; r1 = g.seg[grp + 1] * g.seg_scale; }
    int len = r1 - r0;
    int chunk = ((len + g.splits - 1) / g.splits + BK - 1) / BK * BK;
    k_begin = r0 + sp * chunk;
    k_end = min(r1, k_begin + chunk);
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // software pipeline: the global loads of tile k+1 are in flight while tile k is multiplied on the matrix
  // cores (exact f32 MFMA: bitwise a k-ordered fmaf chain)
  float ra[8], rb[8];
  if (k_begin < k_end) gemm_fetch<A_KM, B_KM, ACT>(g, m0, n0, k_begin, k_end, ra, rb);
  for (int k0 = k_begin; k0 < k_end; k0 += BK) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      lds_put<A_KM>(As, tid + 256 * i, ra[i]);
      lds_put<B_KM>(Bs, tid + 256 * i, rb[i]);
    }
    __syncthreads();
    if (k0 + BK < k_end) gemm_fetch<A_KM, B_KM, ACT>(g, m0, n0, k0 + BK, k_end, ra, rb);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = lds_frag<A_KM>(As, wm + 16 * i + fi, kk + fk);
        b[i] = lds_frag<B_KM>(Bs, wn + 16 * i + fi, kk + fk);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // C/D fragment: lane l, register r <-> row (l>>4)*4 + r, column l&15 of each 16x16 tile
  float* P = g.splits > 0 ? g.C + (size_t)bz * g.M * g.N : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = m0 + wm + 16 * i + fk * 4 + r, gn = n0 + wn + 16 * j + fi;
        if (gm >= g.M || gn >= g.N) continue;
        float v = acc[i][j][r];
        if (P) {
          P[(size_t)gm * g.N + gn] = v;
        } else {
          if (g.bias) v += g.bias[gn];
          v = mg_act(v, g.act, g.slope);
          float* c = g.C + (size_t)gm * g.ldc + gn;
          *c = g.accumulate ? (*c + v) : v;
        }
      }
}

template <bool A_KM, bool B_KM, bool ACT = false>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  gemm_tile<A_KM, B_KM, ACT>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ---- many independent weight-gradient GEMMs (partial-sum mode) in ONE launch --------------------
// A training step has ~20 weight gradients, most of them a handful of workgroups and a few microseconds of
// work: launched one by one they are pure launch latency.  The batch kernel maps a flat block index to
// (problem, tile, split); the partial sums are folded by grad_reduce_multi_kernel afterwards.
#define MG_MAX_GEMM_BATCH 16
struct GemmBatch {
  GemmArgs a[MG_MAX_GEMM_BATCH];
  int blk0[MG_MAX_GEMM_BATCH];  // first flat block of each problem
  int gx[MG_MAX_GEMM_BATCH], gy[MG_MAX_GEMM_BATCH];
  int n;
};

template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(256) void gemm_multi_kernel(GemmBatch bt) {
  int di = 0;
#pragma unroll 1
  for (int i = 1; i < bt.n; ++i)
    if ((int)blockIdx.x >= bt.blk0[i]) di = i;
  const int rel = blockIdx.x - bt.blk0[di];
  const int gx = bt.gx[di], gy = bt.gy[di];
  gemm_tile<A_KM, B_KM, false>(bt.a[di], rel % gx, (rel / gx) % gy, rel / (gx * gy));
}

// dW[grp][m*lddw + n] += sum_z P[grp*splits+z][m*Naug+n];  column Naug-1 -> db[grp][m]
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ P, float* dW, float* db, int M,
                                                            int Naug, int has_bias, int lddw, int splits,
                                                            long w_stride, long b_stride) {
  __shared__ float red[16][64];
  const int grp = blockIdx.y;
  const int lane = threadIdx.x & 63, zl = threadIdx.x >> 6;
  const int o = blockIdx.x * 64 + lane;
  const int total = M * Naug;
  float s = 0.f;
  if (o < total) {
    const float* p = P + ((size_t)grp * splits) * total + o;
    float s0 = 0.f, s1 = 0.f;
    int z = zl;
    for (; z + 16 < splits; z += 32) { s0 += p[(size_t)z * total]; s1 += p[(size_t)(z + 16) * total]; }
    if (z < splits) s0 += p[(size_t)z * total];
    s = s0 + s1;
  }
  red[zl][lane] = s;
  __syncthreads();
  if (zl == 0 && o < total) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][lane];
    int m = o / Naug, n = o % Naug;
    if (has_bias && n == Naug - 1) {
      if (db) db[grp * b_stride + m] += t;
    } else {
      dW[grp * w_stride + (size_t)m * lddw + n] += t;
    }
  }
}

// ---- deferred, batched reduction of many weight-gradient partial buffers in ONE launch ----------
// (a backward pass produces ~20 independent partial buffers; reducing each with its own launch costs
// more in launch latency than in work)
#define MG_MAX_REDUCE 48  // 48 x 72-byte descriptors: under the 4 KB kernel-argument limit
struct ReduceDesc {
  const float* P;   // partials: [groups*splits][p_stride]
  float* dW;
  float* db;
  long w_stride, b_stride;
  int M, Naug, has_bias, lddw, splits, groups, p_stride, block0;  // has_bias: bit 0 = last column is the bias
                                                                  // gradient, bit 1 = overwrite; block0: first block
};
struct ReduceBatch {
  ReduceDesc d[MG_MAX_REDUCE];
  int n;
};

// A bandwidth kernel: a workgroup of 8 waves owns a chunk of 256 consecutive outputs of one (buffer, group) and walks
// the buffer's partial blocks (rows of p_stride floats) with wave w on rows w, w + 8, ...; a lane takes four outputs as
// ONE 16-byte load per row when the buffer allows it (base and row pitch 16-byte aligned: a wave then reads 1 KB per row)
// and as four coalesced 4-byte loads otherwise, eight rows in flight per lane.  The eight waves meet in 8 KB of LDS and
// are added in wave order: the sum is a fixed function of (splits, layout), not of timing.  The producers keep `splits`
// at a few hundred at most (stream_slab_rows), so a wave's walk is at most four or five rounds of eight loads.
// (Round 3's form -- 64 outputs per 1,024-thread workgroup, 256-byte wave reads 4 bytes per lane -- took 121 us for the
// 42-58 MB a backward pass of 8,192 pedestrians left behind: 0.35-0.5 TB/s.)
#define MG_RED_WAVES 8
template <bool VEC>
__device__ __forceinline__ void grad_reduce_chunk(const ReduceDesc& D, const int grp, const int c0, const int total,
                                                  float (*red)[256]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // the four outputs of this lane: VEC: c0 + 4 lane + e (one 16-byte load); else c0 + 64 e + lane (four 4-byte loads)
  f32x4 a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* base = D.P + ((size_t)grp * D.splits) * D.p_stride + c0;
  const size_t ps = (size_t)D.p_stride;
  if (VEC) {
    const int c = 4 * lane;
    if (c0 + c < total) {  // (total % 4 == 0 on this path: a lane's four outputs are all inside or all outside)
      const float* p = base + c;
      int z = w;
      for (; z + MG_RED_WAVES * 7 < D.splits; z += MG_RED_WAVES * 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] += *reinterpret_cast<const f32x4*>(p + (size_t)(z + MG_RED_WAVES * u) * ps);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (z + MG_RED_WAVES * u < D.splits) a[u] += *reinterpret_cast<const f32x4*>(p + (size_t)(z + MG_RED_WAVES * u) * ps);
    }
  } else {
    bool in[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) in[e] = c0 + 64 * e + lane < total;
    const float* p = base + lane;
    int z = w;
    for (; z + MG_RED_WAVES * 7 < D.splits; z += MG_RED_WAVES * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float* q = p + (size_t)(z + MG_RED_WAVES * u) * ps;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (in[e]) a[u][e] += q[64 * e];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (z + MG_RED_WAVES * u < D.splits) {
        const float* q = p + (size_t)(z + MG_RED_WAVES * u) * ps;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (in[e]) a[u][e] += q[64 * e];
      }
  }
  const f32x4 s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  // LDS column = offset of the output inside the chunk
  if (VEC) {
    *reinterpret_cast<f32x4*>(&red[w][4 * lane]) = s;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) red[w][64 * e + lane] = s[e];
  }
  __syncthreads();
  if (threadIdx.x < 256) {
    const int o = c0 + threadIdx.x;
    if (o < total) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < MG_RED_WAVES; ++i) t += red[i][threadIdx.x];
      const int m = o / D.Naug, n = o - m * D.Naug;
      const bool overwrite = (D.has_bias & 2) != 0;  // bit 1: '=' instead of '+=' (scratch destinations)
      if ((D.has_bias & 1) && n == D.Naug - 1) {
        if (D.db) {
          float* q = D.db + grp * D.b_stride + m;
          *q = overwrite ? t : *q + t;
        }
      } else {
        float* q = D.dW + grp * D.w_stride + (size_t)m * D.lddw + n;
        *q = overwrite ? t : *q + t;
      }
    }
  }
}

// Tall and narrow buffers (hundreds of partial blocks of a few hundred outputs: the per-workgroup partials of the scene CNN
// and the attention kernels, the slabs of the small products at 1,280 pedestrians): a chunk of 64 outputs, a lane = four
// outputs (cg = lane & 15) of one of FOUR rows a wave reads at once (rg = lane >> 4): 32 rows in flight per workgroup
// pass instead of 8, four times the workgroups.  (With 256-output chunks these buffers made the launch 25-29 us at
// 1,280 pedestrians, where round 3's kernel took 15.)
template <bool VEC>
__device__ __forceinline__ void grad_reduce_chunk_tall(const ReduceDesc& D, const int grp, const int c0, const int total,
                                                       float (*red)[256]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, cg = lane & 15, rg = lane >> 4;
  constexpr int RG = 4 * MG_RED_WAVES;  // row groups per workgroup
  f32x4 a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int c = c0 + 4 * cg;
  const float* p = D.P + ((size_t)grp * D.splits) * D.p_stride + c;
  const size_t ps = (size_t)D.p_stride;
  bool in[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) in[e] = c + e < total;
  auto ld = [&](int z) -> f32x4 {
    const float* q = p + (size_t)z * ps;
    if (VEC) return in[0] ? *reinterpret_cast<const f32x4*>(q) : f32x4{0.f, 0.f, 0.f, 0.f};
    return f32x4{in[0] ? q[0] : 0.f, in[1] ? q[1] : 0.f, in[2] ? q[2] : 0.f, in[3] ? q[3] : 0.f};
  };
  int z = 4 * w + rg;
  for (; z + RG * 7 < D.splits; z += RG * 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += ld(z + RG * u);
  }
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (z + RG * u < D.splits) a[u] += ld(z + RG * u);
  const f32x4 s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  // LDS: red[wave][rg * 64 + 4 cg + e]
  *reinterpret_cast<f32x4*>(&red[w][64 * rg + 4 * cg]) = s;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int o = c0 + threadIdx.x;
    if (o < total) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < MG_RED_WAVES; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) t += red[i][64 * r + threadIdx.x];
      const int m = o / D.Naug, n = o - m * D.Naug;
      const bool overwrite = (D.has_bias & 2) != 0;
      if ((D.has_bias & 1) && n == D.Naug - 1) {
        if (D.db) {
          float* q = D.db + grp * D.b_stride + m;
          *q = overwrite ? t : *q + t;
        }
      } else {
        float* q = D.dW + grp * D.w_stride + (size_t)m * D.lddw + n;
        *q = overwrite ? t : *q + t;
      }
    }
  }
}

__global__ __launch_bounds__(64 * MG_RED_WAVES) void grad_reduce_multi_kernel(ReduceBatch bt) {
  __shared__ __attribute__((aligned(16))) float red[MG_RED_WAVES][256];
  int di = 0;
#pragma unroll 1
  for (int i = 1; i < bt.n; ++i)
    if ((int)blockIdx.x >= bt.d[i].block0) di = i;
  const ReduceDesc& D = bt.d[di];
  const int total = D.M * D.Naug;
  const bool tall = (D.has_bias & 4) != 0;  // (set by the launcher)
  const int cw = tall ? 64 : 256;
  const int cpg = (total + cw - 1) / cw;  // chunks per group
  const int rel = blockIdx.x - D.block0;
  const int grp = rel / cpg, c0 = (rel - grp * cpg) * cw;
  const bool vec = (((size_t)D.P & 15) == 0) && (D.p_stride & 3) == 0 && (total & 3) == 0;
  if (tall) {
    if (vec) grad_reduce_chunk_tall<true>(D, grp, c0, total, red);
    else grad_reduce_chunk_tall<false>(D, grp, c0, total, red);
  } else if (vec) grad_reduce_chunk<true>(D, grp, c0, total, red);
  else grad_reduce_chunk<false>(D, grp, c0, total, red);
}

// dZ = dY * act'(Y)
__global__ void act_bwd_kernel(const float* __restrict__ dY, int lddy, const float* __restrict__ Y, int ldy,
                               float* __restrict__ dZ, int lddz, int rows, int N, int act, float slope) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * N) return;
  int r = (int)(i / N), n = (int)(i % N);
  dZ[(size_t)r * lddz + n] = dY[(size_t)r * lddy + n] * mg_act_grad_from_out(Y[(size_t)r * ldy + n], act, slope);
}

// ---- streaming weight gradients --------------------------------------------------------------------------------
// dW[n][k] = sum_r dZ[r][n] X[r][k], db[n] = sum_r dZ[r][n]: 1,280 - 524,288 rows against outputs of a few dozen
// features, i.e. operand streaming (442 MB of pair activations per discriminator step at 256 x 32 pedestrians).  The
// tiled kernel above stages 64 x 64 x 32 blocks through LDS with 4-byte loads and two barriers per 32 rows and reached a
// third of the HBM rate on them.  Here a wave reads its MFMA fragments straight from memory, no LDS, no barrier:
//  * feature-major operands ([feature][row]: the pair / position kernels of the social and scene attention write them
//    that way, one lane per row): with the reduction index of a 16-row super-step permuted as k = 4 fk + i (the same
//    permutation on both operands: the sum is over the same set), lane (fi, fk) needs rows r + 4 fk .. r + 4 fk + 3 of
//    feature fi - one 16-byte load per operand tile;
//  * row-major operands ([row][feature], everything else): the same rows as four 4-byte loads, 16 lanes side by side.
// A workgroup owns one contiguous slab of rows and one panel of at most 64 x 64 outputs (wider products are cut into
// panels: an operand is then read once per panel of the OTHER operand); its four waves interleave the slab's
// super-steps and fold their accumulators through LDS at the end: ONE partial block per slab, [N][K+1] with column
// K = db (summed on the VALU from the fragments already loaded), in the layout grad_reduce_multi_kernel folds.
struct StreamProb {
  const float *A, *B;  // dZ, X
  float* P;            // partial blocks [slab][M][Kf + 1]
  int rows, M, Kf, lda, ldb, slab, blk0, mode, pm, pk;  // mode 0: feature-major, 16-byte loads; 1: feature-major,
};                                                      // unaligned; 2: row-major
#define MG_MAX_STREAM_BATCH 16
struct StreamBatch {
  StreamProb p[MG_MAX_STREAM_BATCH];
  int n;
};

template <int MT, int NT, int MODE>
__device__ __forceinline__ void wgrad_stream_slab(const StreamProb& q, const int slab, const int m0, const int k0,
                                                  float* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  const int r0 = slab * q.slab, r1 = min(q.rows, r0 + q.slab);
  // features past the matrix read feature 0: their products only reach output rows / columns that are never stored
  const float* ap[MT];
  const float* bp[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int f = m0 + 16 * i + fi < q.M ? m0 + 16 * i + fi : 0;
    ap[i] = MODE == 2 ? q.A + (size_t)(4 * fk) * q.lda + f : q.A + (size_t)f * q.lda + 4 * fk;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int f = k0 + 16 * j + fi < q.Kf ? k0 + 16 * j + fi : 0;
    bp[j] = MODE == 2 ? q.B + (size_t)(4 * fk) * q.ldb + f : q.B + (size_t)f * q.ldb + 4 * fk;
  }
  f32x4 acc[MT][NT];
  float bsum[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    bsum[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto load = [&](const float* p, int r, int ld) -> f32x4 {  // rows r + 4 fk + (0..3) of this lane's feature
    if (MODE == 0) return *reinterpret_cast<const f32x4*>(p + r);
    if (MODE == 1) return f32x4{p[r], p[r + 1], p[r + 2], p[r + 3]};
    const float* pr = p + (size_t)r * ld;
    return f32x4{pr[0], pr[ld], pr[2 * ld], pr[3 * ld]};
  };
  auto mac = [&](const f32x4* a, const f32x4* b) {
#pragma unroll
    for (int i = 0; i < MT; ++i) bsum[i] += (a[i][0] + a[i][1]) + (a[i][2] + a[i][3]);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
  };
  // full super-steps, two in flight per wave (wave w takes super-steps w, w + 4, ...: the workgroup walks its slab
  // 64 rows at a time)
  const int nfull = (r1 - r0) / 16;
  f32x4 a0[MT], b0[NT], a1[MT], b1[NT];
  int s = w;
  if (s < nfull) {
#pragma unroll
    for (int i = 0; i < MT; ++i) a0[i] = load(ap[i], r0 + 16 * s, q.lda);
#pragma unroll
    for (int j = 0; j < NT; ++j) b0[j] = load(bp[j], r0 + 16 * s, q.ldb);
  }
  for (; s < nfull; s += 8) {
    const bool more = s + 4 < nfull;
    const int rn = r0 + 16 * (more ? s + 4 : s);  // (the last odd step re-reads itself: no branch between the loads)
#pragma unroll
    for (int i = 0; i < MT; ++i) a1[i] = load(ap[i], rn, q.lda);
#pragma unroll
    for (int j = 0; j < NT; ++j) b1[j] = load(bp[j], rn, q.ldb);
    mac(a0, b0);
    const int rm = r0 + 16 * (s + 8 < nfull ? s + 8 : s);
#pragma unroll
    for (int i = 0; i < MT; ++i) a0[i] = load(ap[i], rm, q.lda);
#pragma unroll
    for (int j = 0; j < NT; ++j) b0[j] = load(bp[j], rm, q.ldb);
    if (more) mac(a1, b1);
  }
  // the slab's last (partial) super-step: wave 0, element by element
  if (w == 0 && r0 + 16 * nfull < r1) {
    const int r = r0 + 16 * nfull;
    f32x4 a[MT], b[NT];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool in = r + 4 * fk + e < r1;
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i][e] = in ? (MODE == 2 ? ap[i][(size_t)(r + e) * q.lda] : ap[i][r + e]) : 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j][e] = in ? (MODE == 2 ? bp[j][(size_t)(r + e) * q.ldb] : bp[j][r + e]) : 0.f;
    }
    mac(a, b);
  }
  // fold the four waves (fixed order) and store this workgroup's part of the slab's partial block
  const int Naug = q.Kf + 1;
  float* P = q.P + (size_t)slab * q.M * Naug;
  f32x4* red4 = reinterpret_cast<f32x4*>(red);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int t = i * NT + j;
      red4[((t & 1) * 4 + w) * 64 + lane] = acc[i][j];
      __syncthreads();
      if (w == (t & 3)) {  // tile t is summed and stored by wave t mod 4
        const f32x4* src = red4 + (t & 1) * 256 + lane;
        const f32x4 v = (src[0] + src[64]) + (src[128] + src[192]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gm = m0 + 16 * i + 4 * fk + r, gn = k0 + 16 * j + fi;
          if (gm < q.M && gn < q.Kf) P[(size_t)gm * Naug + gn] = v[r];
        }
      }
    }
  }
  if (k0 != 0) return;  // db comes from the first panel of a row of panels
  __syncthreads();
  // lane (fi, fk) holds the sum over its rows of feature m0 + 16 i + fi
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    float v = bsum[i];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (fk == 0) red[(i * 4 + w) * 16 + fi] = v;
  }
  __syncthreads();
  if (threadIdx.x < 16 * MT) {
    const int i = threadIdx.x >> 4, f = threadIdx.x & 15, gm = m0 + 16 * i + f;
    const float* src = red + i * 64 + f;
    if (gm < q.M) P[(size_t)gm * Naug + q.Kf] = (src[0] + src[16]) + (src[32] + src[48]);
  }
}

// one kernel per operand layout: each keeps its own register budget (the row-major form needs more address registers)
// (three waves per SIMD: the row-major form compiled to 176 registers, two waves -- 59 -> 50 us per launch at 256 x 32)
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void wgrad_stream_kernel(StreamBatch bt) {
  __shared__ __attribute__((aligned(16))) float red[2 * 4 * 64 * 4];
  int di = 0;
#pragma unroll 1
  for (int i = 1; i < bt.n; ++i)
    if ((int)blockIdx.x >= bt.p[i].blk0) di = i;
  const StreamProb& q = bt.p[di];
  const int rel = blockIdx.x - q.blk0, np = q.pm * q.pk;
  const int slab = rel / np, panel = rel - slab * np;
  const int m0 = 64 * (panel / q.pk), k0 = 64 * (panel % q.pk);
  const int mt = (min(64, q.M - m0) + 15) >> 4, nt = (min(64, q.Kf - k0) + 15) >> 4;  // 1..4 each
  const int MTc = mt > 2 ? 4 : mt, NTc = nt > 2 ? 4 : nt;
  switch (MTc * 8 + NTc) {
    case 1 * 8 + 1: wgrad_stream_slab<1, 1, MODE>(q, slab, m0, k0, red); break;
    case 1 * 8 + 2: wgrad_stream_slab<1, 2, MODE>(q, slab, m0, k0, red); break;
    case 1 * 8 + 4: wgrad_stream_slab<1, 4, MODE>(q, slab, m0, k0, red); break;
    case 2 * 8 + 1: wgrad_stream_slab<2, 1, MODE>(q, slab, m0, k0, red); break;
    case 2 * 8 + 2: wgrad_stream_slab<2, 2, MODE>(q, slab, m0, k0, red); break;
    case 2 * 8 + 4: wgrad_stream_slab<2, 4, MODE>(q, slab, m0, k0, red); break;
    case 4 * 8 + 1: wgrad_stream_slab<4, 1, MODE>(q, slab, m0, k0, red); break;
    case 4 * 8 + 2: wgrad_stream_slab<4, 2, MODE>(q, slab, m0, k0, red); break;
    default: wgrad_stream_slab<4, 4, MODE>(q, slab, m0, k0, red); break;
  }
}

static void launch_stream(const StreamBatch& sb, int blocks, int mode, hipStream_t stream) {
  if (mode == 0) MG_LAUNCH(wgrad_stream_kernel<0>, dim3(blocks), dim3(256), 0, stream, sb);
  else if (mode == 1) MG_LAUNCH(wgrad_stream_kernel<1>, dim3(blocks), dim3(256), 0, stream, sb);
  else MG_LAUNCH(wgrad_stream_kernel<2>, dim3(blocks), dim3(256), 0, stream, sb);
}

extern "C" {

int mggan_linear_fwd(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int rows, int K,
                     int N, int act, float slope, hipStream_t stream) {
  MG_CHECK_ARG(X && W && Y && rows >= 0 && K > 0 && N > 0, "linear_fwd: bad arguments");
  if (rows == 0) return MGGAN_OK;
  GemmArgs g = {};
  g.A = X; g.B = W; g.C = Y; g.bias = bias;
  g.M = rows; g.N = N; g.K = K; g.lda = ldx; g.ldb = K; g.ldc = ldy;
  g.act = act; g.slope = slope;
  dim3 grid(cdiv(N, BN), cdiv(rows, BM), 1);
  MG_LAUNCH((gemm_kernel<false, false>), grid, dim3(256), 0, stream, g);
  MG_LAUNCH_CHECK("linear_fwd");
  return MGGAN_OK;
}

int mggan_act_bwd(const float* dY, int lddy, const float* Y, int ldy, float* dZ, int lddz, int rows, int N, int act,
                  float slope, hipStream_t stream) {
  MG_CHECK_ARG(dY && Y && dZ, "act_bwd: null pointer");
  long n = (long)rows * N;
  if (n == 0) return MGGAN_OK;
  MG_LAUNCH(act_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, dY, lddy, Y, ldy, dZ, lddz, rows, N, act,
                     slope);
  MG_LAUNCH_CHECK("act_bwd");
  return MGGAN_OK;
}

int mggan_linear_bwd_data(const float* dZ, int lddz, const float* W, int ldw, float* dX, int lddx, int rows, int K,
                          int N, int accumulate, const float* Yact, int ld_yact, int act, float slope,
                          hipStream_t stream) {
  MG_CHECK_ARG(dZ && W && dX && K > 0 && N > 0, "linear_bwd_data: bad arguments");
  if (rows == 0) return MGGAN_OK;
  GemmArgs g = {};
  g.A = dZ; g.B = W; g.C = dX;
  g.M = rows; g.N = K; g.K = N; g.lda = lddz; g.ldb = ldw; g.ldc = lddx;
  g.accumulate = accumulate;
  dim3 grid(cdiv(K, BN), cdiv(rows, BM), 1);
  if (Yact && act != ACT_NONE) {
    g.Yact = Yact; g.ld_yact = ld_yact; g.act_a = act; g.slope_a = slope;
    MG_LAUNCH((gemm_kernel<false, true, true>), grid, dim3(256), 0, stream, g);
  } else {
    MG_LAUNCH((gemm_kernel<false, true>), grid, dim3(256), 0, stream, g);
  }
  MG_LAUNCH_CHECK("linear_bwd_data");
  return MGGAN_OK;
}

size_t mggan_wgrad_workspace_bytes(int rows, int K, int N, int n_groups) {
  int splits = mggan_wgrad_splits(rows, K, N, n_groups);
  return (size_t)splits * (n_groups > 0 ? n_groups : 1) * N * (K + 1) * sizeof(float);
}

// rows per workgroup (slab) of the streaming kernel.  Every slab leaves a partial block of N (K + 1) floats that the
// batched reduction reads back, so a slab must be long enough for that block to be small beside the operands it was
// summed from (<= 1/8: with round 3's fixed 768 slabs the 57,344 x (256 x 64) gate-gradient product wrote 30 MB of
// partials for 73 MB of operands), and there are at most 256 slabs per problem (a wave of the reduction then walks at most
// 32 rows; the problems of a batch fill the chip together, not each on its own); never less than 128 rows.
static int stream_slab_rows(int rows, int K, int N) {
  static int factor = -1;  // MGGAN_WGRAD_SLAB_FACTOR: partial block <= 1/factor of the operands (measurement knob)
  if (factor < 0) { const char* e = getenv("MGGAN_WGRAD_SLAB_FACTOR"); factor = e ? atoi(e) : 8; if (factor < 1) factor = 1; }
  int s = cdiv(cdiv(rows, 256), 64) * 64;
  const int by_output = cdiv(cdiv((long)factor * N * (K + 1), K + N), 64) * 64;
  if (by_output > s) s = by_output;
  static int floor_rows = -1;  // MGGAN_WGRAD_SLAB_MIN
  if (floor_rows < 0) { const char* e = getenv("MGGAN_WGRAD_SLAB_MIN"); floor_rows = e ? atoi(e) : 128; if (floor_rows < 64) floor_rows = 64; }
  return s < floor_rows ? floor_rows : s;
}

static StreamProb stream_problem(const float* dZ, const float* X, float* workspace, int rows, int K, int N, int lddz,
                                 int ldx, int feature_major) {
  StreamProb q = {};
  q.A = dZ; q.B = X; q.P = workspace; q.rows = rows; q.M = N; q.Kf = K; q.lda = lddz; q.ldb = ldx;
  q.slab = stream_slab_rows(rows, K, N);
  q.pm = cdiv(N, 64); q.pk = cdiv(K, 64);
  const bool vec = (lddz % 4 == 0) && (ldx % 4 == 0) && ((size_t)dZ % 16 == 0) && ((size_t)X % 16 == 0);
  q.mode = feature_major ? (vec ? 0 : 1) : 2;
  return q;
}

int mggan_wgrad_splits(int rows, int K, int N, int n_groups) {
  if (n_groups <= 1) return rows > 0 ? cdiv(rows, stream_slab_rows(rows, K, N)) : 1;  // slabs of the streaming kernel
  // short dependent chains: 128-256 rows (4-8 pipelined k-steps) per workgroup, up to ~4 workgroups per CU
  int ng = n_groups > 0 ? n_groups : 1;
  int tiles = cdiv(N, BM) * cdiv(K + 1, BN) * ng;
  int per_group_rows = rows / ng + 1;
  int s = cdiv(per_group_rows, 256);
  if (s * tiles < 512) s = cdiv(per_group_rows, 128);
  int cap = cdiv(2048, tiles);
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  return s;
}

int mggan_grad_reduce_multi(const void* descs, int n, hipStream_t stream) {
  MG_CHECK_ARG(descs || n == 0, "grad_reduce_multi: null descriptor array");
  const ReduceDesc* in = (const ReduceDesc*)descs;
  for (int i0 = 0; i0 < n; i0 += MG_MAX_REDUCE) {
    ReduceBatch bt;
    bt.n = n - i0 < MG_MAX_REDUCE ? n - i0 : MG_MAX_REDUCE;
    int blocks = 0;
    for (int i = 0; i < bt.n; ++i) {
      bt.d[i] = in[i0 + i];
      MG_CHECK_ARG(bt.d[i].P && bt.d[i].dW && bt.d[i].splits > 0 && bt.d[i].groups > 0, "grad_reduce_multi: bad descriptor");
      bt.d[i].block0 = blocks;
      const long total = (long)bt.d[i].M * bt.d[i].Naug;
      const bool tall = bt.d[i].splits >= 128 && total <= 8192;  // many partial blocks of few outputs: 64-output chunks
      bt.d[i].has_bias = (bt.d[i].has_bias & 3) | (tall ? 4 : 0);
      blocks += cdiv(total, tall ? 64 : 256) * bt.d[i].groups;
    }
    MG_LAUNCH(grad_reduce_multi_kernel, dim3(blocks), dim3(64 * MG_RED_WAVES), 0, stream, bt);
    MG_LAUNCH_CHECK("grad_reduce_multi");
  }
  return MGGAN_OK;
}

int mggan_wgrad(const float* dZ, int lddz, const float* X, int ldx, float* dW, int lddw, float* db, int rows, int K,
                int N, const int* seg, int seg_scale, int n_groups, long w_stride, long b_stride, int feature_major,
                const float* Yact, int ld_yact, int act, float slope, void* workspace, size_t workspace_bytes,
                hipStream_t stream) {
  MG_CHECK_ARG(dZ && X && K > 0 && N > 0, "wgrad: bad arguments");
  if (rows == 0) return MGGAN_OK;
  const int ng = n_groups > 0 ? n_groups : 1;
  MG_CHECK_ARG(ng == 1 || seg, "wgrad: grouped mode needs segment offsets");
  const bool streaming = n_groups <= 1 && !(Yact && act != ACT_NONE);  // (grouped / fused-derivative forms: tiled kernel)
  const int splits = mggan_wgrad_splits(rows, K, N, n_groups);
  const int Naug = K + 1;
  size_t need = (size_t)splits * ng * N * Naug * sizeof(float);
  if (workspace_bytes < need || !workspace) {
    mggan_set_error("wgrad: workspace too small (%zu < %zu)", workspace_bytes, need);
    return MGGAN_ERR_WORKSPACE;
  }
  GemmArgs g = {};
  g.A = dZ; g.B = X; g.C = (float*)workspace;
  g.M = N; g.N = Naug; g.K = rows; g.lda = lddz; g.ldb = ldx; g.ldc = Naug;
  g.splits = splits; g.ones_col = 1; g.seg = seg; g.seg_scale = seg_scale > 0 ? seg_scale : 1; g.n_groups = ng;
  if (Yact && act != ACT_NONE) {
    MG_CHECK_ARG(!feature_major, "wgrad: fused activation derivative needs row-major operands");
    g.Yact = Yact; g.ld_yact = ld_yact; g.act_a = act; g.slope_a = slope;
  }
  dim3 grid(cdiv(Naug, BN), cdiv(N, BM), splits * ng);
  // feature_major: operands stored [feature][row] (written coalesced by one-lane-per-row kernels)
  if (streaming) {
    StreamBatch sb;
    sb.n = 1;
    sb.p[0] = stream_problem(dZ, X, (float*)workspace, rows, K, N, lddz, ldx, feature_major);
    sb.p[0].blk0 = 0;
    launch_stream(sb, splits * sb.p[0].pm * sb.p[0].pk, sb.p[0].mode, stream);
  } else if (feature_major) MG_LAUNCH((gemm_kernel<false, false>), grid, dim3(256), 0, stream, g);
  else if (g.Yact) MG_LAUNCH((gemm_kernel<true, true, true>), grid, dim3(256), 0, stream, g);
  else MG_LAUNCH((gemm_kernel<true, true>), grid, dim3(256), 0, stream, g);
  MG_LAUNCH_CHECK("wgrad");
  if (!dW) return MGGAN_OK;  // deferred: the caller reduces the partials later (mggan_grad_reduce_multi)
  dim3 rgrid(cdiv((long)N * Naug, 64), ng, 1);
  MG_LAUNCH(wgrad_reduce_kernel, rgrid, dim3(1024), 0, stream, (const float*)workspace, dW, db, N, Naug, 1,
                     lddw, splits, w_stride, b_stride);
  MG_LAUNCH_CHECK("wgrad_reduce");
  return MGGAN_OK;
}

struct WgradDesc {  // mirrors the ctypes structure in mggan/hip/functions.py
  const float* dZ;
  const float* X;
  float* workspace;
  const int* seg;
  int rows, K, N, lddz, ldx, seg_scale, n_groups, feature_major;
};

int mggan_wgrad_multi(const void* descs, int n, hipStream_t stream) {
  MG_CHECK_ARG(descs || n == 0, "wgrad_multi: null descriptor array");
  const WgradDesc* d = (const WgradDesc*)descs;
  for (int mode = 0; mode < 3; ++mode) {  // ungrouped problems: the streaming kernel, one launch per layout and 16 problems
    StreamBatch sb;
    sb.n = 0;
    int blocks = 0;
    auto launch = [&]() {
      if (sb.n == 0) return;
      launch_stream(sb, blocks, mode, stream);
      sb.n = 0;
      blocks = 0;
    };
    for (int i = 0; i < n; ++i) {
      if (d[i].rows == 0 || d[i].n_groups > 1) continue;
      MG_CHECK_ARG(d[i].dZ && d[i].X && d[i].workspace && d[i].K > 0 && d[i].N > 0, "wgrad_multi: bad descriptor %d", i);
      StreamProb q = stream_problem(d[i].dZ, d[i].X, d[i].workspace, d[i].rows, d[i].K, d[i].N, d[i].lddz, d[i].ldx,
                                    d[i].feature_major);
      if (q.mode != mode) continue;
      q.blk0 = blocks;
      blocks += mggan_wgrad_splits(d[i].rows, d[i].K, d[i].N, 0) * q.pm * q.pk;
      sb.p[sb.n] = q;
      if (++sb.n == MG_MAX_STREAM_BATCH) launch();
    }
    launch();
    MG_LAUNCH_CHECK("wgrad_multi (streaming)");
  }
  for (int fm = 0; fm < 2; ++fm) {
    GemmBatch bt;
    bt.n = 0;
    int blocks = 0;
    auto launch = [&]() {
      if (bt.n == 0) return;
      if (fm) MG_LAUNCH((gemm_multi_kernel<false, false>), dim3(blocks), dim3(256), 0, stream, bt);
      else MG_LAUNCH((gemm_multi_kernel<true, true>), dim3(blocks), dim3(256), 0, stream, bt);
      bt.n = 0;
      blocks = 0;
    };
    for (int i = 0; i < n; ++i) {
      if ((d[i].feature_major != 0) != (fm != 0) || d[i].rows == 0) continue;
      if (d[i].n_groups <= 1) continue;  // went out with the streaming batch
      MG_CHECK_ARG(d[i].dZ && d[i].X && d[i].workspace && d[i].K > 0 && d[i].N > 0, "wgrad_multi: bad descriptor %d", i);
      const int ng = d[i].n_groups > 0 ? d[i].n_groups : 1;
      MG_CHECK_ARG(ng == 1 || d[i].seg, "wgrad_multi: grouped mode needs segment offsets");
      GemmArgs g = {};
      g.A = d[i].dZ; g.B = d[i].X; g.C = d[i].workspace;
      g.M = d[i].N; g.N = d[i].K + 1; g.K = d[i].rows; g.lda = d[i].lddz; g.ldb = d[i].ldx; g.ldc = g.N;
      g.splits = mggan_wgrad_splits(d[i].rows, d[i].K, d[i].N, d[i].n_groups);
      g.ones_col = 1; g.seg = d[i].seg; g.seg_scale = d[i].seg_scale > 0 ? d[i].seg_scale : 1; g.n_groups = ng;
      bt.a[bt.n] = g;
      bt.blk0[bt.n] = blocks;
      bt.gx[bt.n] = cdiv(g.N, BN);
      bt.gy[bt.n] = cdiv(g.M, BM);
      blocks += bt.gx[bt.n] * bt.gy[bt.n] * g.splits * ng;
      if (++bt.n == MG_MAX_GEMM_BATCH) launch();
    }
    launch();
    MG_LAUNCH_CHECK("wgrad_multi");
  }
  return MGGAN_OK;
}

}  // extern "C"
