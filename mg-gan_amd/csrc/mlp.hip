// Fused dense chains for the small MLP stacks of MG-GAN (2-3 Linear layers with activations between
// them): discriminator heads Linear(192,96)-LeakyReLU-Linear(96,1|g), pred_encoder 24-64-32, in_encoder_fc
// 64-32-32 (reference model/modules/discriminators.py:46-56,76-108), the PM-network 128-16-16-g
// (model/modules/standard.py:99-105), all built by utils.make_mlp (utils.py:134-149).
//
// One launch runs the whole chain; a workgroup of four waves owns 16 rows.  The activations of the tile live in
// two 12.5 KB LDS buffers (ping-pong between stages); the WEIGHTS never touch LDS: they are a few
// tens of KB and L2-resident, and every wave fetches the B fragments of its own 16-column tiles straight into
// registers with 16-byte loads along k (the k index inside a 16-wide super-step is permuted so that lane
// (n, fk) needs 4 consecutive floats: k = 16 S + 4 fk + i for step i of super-step S; A uses the same
// permutation, so the sum is over the same set).  Products on v_mfma_f32_16x16x4_f32 (exact f32).  A first
// version kept all weights of the chain as fragment images in LDS (weight-stationary persistent workgroups,
// 80-130 KB): one workgroup per CU, and its launches sat behind the scene-CNN kernels of the other stream until a
// CU had that much LDS free (17-22 us per launch in the training step, up to 83 us); this one needs 25 KB.
// The backward pass is the same kernel: its stages use the weights transposed (dX = dZ W), the activation
// derivative enters as an elementwise factor taken from the saved forward output, and the gate gradients dZ_l
// every stage produces are stored for the weight-gradient GEMMs.
#include "common.h"
#include "../../include/mggan_hip.h"

#define MC_ROWS 16
#define MC_LDX 196     // activation row stride (floats): 16-byte aligned rows, 4 mod 32
#define MC_MAXD 192

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct McStage {  // mirrors mggan/hip/functions.py:_McStage
  const float* W;        // forward layout [N_fwd][K_fwd]
  const float* bias;     // added to the stage output (NULL: none)
  const float* mul_src;  // output *= act'(mul_src[row][col]) through the activation OUTPUT (NULL: none)
  float* out;            // global copy of the stage output (NULL: stays on chip)
  int K, N;              // reduction length / output width of THIS stage
  int ldw;               // row stride of W
  int trans;             // 0: out[r][n] = sum_k in[r][k] W[n][k]   1: out[r][n] = sum_k in[r][k] W[k][n]
  int act, mul_act, ld_mul, ld_out, accumulate;
  float slope, mul_slope;
};
struct McArgs {
  const float* X;        // (rows, K0) input, row stride ldx
  const float* in_mul;   // optional: input *= act'(in_mul[row][k])  (backward through the last activation)
  float* in_store;       // optional: global copy of the transformed input (gate gradient of the last layer)
  int ldx, rows, K0, ld_in_mul, in_mul_act, ld_in_store, n;
  float in_mul_slope;
  McStage s[3];
};

// B fragments of one 16-column tile for the 2 super-steps of chunk c: b[s] holds the W elements
// (k = 16 (2c + s) + 4 fk + i, n = n0 + fi), i = 0..3; zero outside the matrix
__device__ __forceinline__ void mc_load_b(const McStage& S, int n0, int fi, int fk, int c, f32x4* b) {
  const int n = n0 + fi, K = S.K, N = S.N;
  if (!S.trans) {
    const bool vec = (S.ldw & 3) == 0 && (((size_t)S.W & 15) == 0);
    const float* row = S.W + (size_t)(n < N ? n : 0) * S.ldw;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int k0 = 16 * (2 * c + s) + 4 * fk;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n < N && k0 < K) {
        if (vec && k0 + 3 < K) {
          const float4 t = *reinterpret_cast<const float4*>(row + k0);
          v = f32x4{t.x, t.y, t.z, t.w};
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (k0 + i < K) v[i] = row[k0 + i];
        }
      }
      b[s] = v;
    }
  } else {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = 16 * (2 * c + s) + 4 * fk + i;
        if (k < K && n < N) v[i] = S.W[(size_t)k * S.ldw + n];
      }
      b[s] = v;
    }
  }
}

// The kernel is meant to slip in beside the long kernels of the other stream, so it stays small in every
// resource: 25 KB of LDS and ~128 VGPRs (k is walked in chunks of two 16-wide super-steps, the next chunk's
// fragments requested while the current one is multiplied) -- a version that held the whole K of A and B in
// registers (332 VGPRs: one wave per SIMD) waited for a free SIMD behind the scene-CNN kernels like the
// LDS-heavy one before it.
__global__ __launch_bounds__(256) void mlp_chain_kernel(McArgs a) {
  __shared__ __attribute__((aligned(16))) float tiles[2][MC_ROWS * MC_LDX];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fi = lane & 15, fk = lane >> 4;
  const int r0 = blockIdx.x * MC_ROWS;
  float* act = tiles[0];
  float* nxt = tiles[1];

  {  // input tile, zero-padded to a multiple of 16 columns and to 16 rows; optional x *= act'(y) (+ copy out)
    const int K0 = a.K0, KP = (K0 + 31) & ~31;  // products walk k in chunks of 32: pad with zeros
    const bool vec = (a.ldx & 3) == 0 && (K0 & 3) == 0 && (((size_t)a.X & 15) == 0) && !a.in_mul;
    if (vec) {  // <= 3 quads per thread, all requested before the first LDS store
      const int qn = KP >> 2, total = MC_ROWS * qn;
      float4 v[3];
      int at[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int q = tid + 256 * u, qc = q < total ? q : 0;
        const int r = qc / qn, k = (qc - r * qn) << 2;
        const bool ok = q < total && r0 + r < a.rows && k < K0;
        const float4 t = *reinterpret_cast<const float4*>(ok ? a.X + (size_t)(r0 + r) * a.ldx + k : a.X);
        v[u] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        at[u] = q < total ? r * MC_LDX + k : -1;
      }
#pragma unroll
      for (int u = 0; u < 3; ++u)
        if (at[u] >= 0) *reinterpret_cast<float4*>(&act[at[u]]) = v[u];
    } else {
      for (int e = tid; e < MC_ROWS * KP; e += 256) {
        const int r = e / KP, k = e - r * KP, gr = r0 + r;
        float v = 0.f;
        if (gr < a.rows && k < K0) {
          v = a.X[(size_t)gr * a.ldx + k];
          if (a.in_mul) v *= mg_act_grad_from_out(a.in_mul[(size_t)gr * a.ld_in_mul + k], a.in_mul_act, a.in_mul_slope);
          if (a.in_store) a.in_store[(size_t)gr * a.ld_in_store + k] = v;
        }
        act[r * MC_LDX + k] = v;
      }
    }
  }

#pragma unroll 1
  for (int si = 0; si < a.n; ++si) {
    const McStage& S = a.s[si];
    const int N = S.N, NC = (S.K + 31) >> 5, ntt = (N + 15) >> 4;  // chunks of 32 k, 16-column tiles
    const bool last = si == a.n - 1;
    __syncthreads();  // `act` holds this stage's input (and nobody still reads `nxt`)
    const float* arow = &act[fi * MC_LDX + 4 * fk];
#pragma unroll 1
    for (int j = w; j < ntt; j += 4) {
      const int col = 16 * j + fi;
      f32x4 av[2], bv[2], an[2], bn[2];
      mc_load_b(S, 16 * j, fi, fk, 0, bv);
#pragma unroll
      for (int s = 0; s < 2; ++s) av[s] = *reinterpret_cast<const f32x4*>(arow + 16 * s);
      // saved activation for the derivative factor (rows 4 fk + r of this lane's D registers)
      float mulv[4] = {1.f, 1.f, 1.f, 1.f};
      if (S.mul_src) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gr = r0 + 4 * fk + r;
          const bool ok = col < N && gr < a.rows;
          const float y = *(ok ? S.mul_src + (size_t)gr * S.ld_mul + col : S.mul_src);
          mulv[r] = ok ? mg_act_grad_from_out(y, S.mul_act, S.mul_slope) : 0.f;
        }
      }
      const float bias = (S.bias && col < N) ? S.bias[col] : 0.f;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};  // two chains hide the MFMA latency
#pragma unroll 1
      for (int c = 0; c < NC; ++c) {
        if (c + 1 < NC) {  // next chunk's fragments: in flight during this chunk's products
          mc_load_b(S, 16 * j, fi, fk, c + 1, bn);
#pragma unroll
          for (int s = 0; s < 2; ++s) an[s] = *reinterpret_cast<const f32x4*>(arow + 32 * (c + 1) + 16 * s);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {  // (super-steps past K multiply zero-padded columns by zero weights)
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][0], bv[s][0], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][1], bv[s][1], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][2], bv[s][2], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][3], bv[s][3], acc1, 0, 0, 0);
        }
        if (c + 1 < NC) {
#pragma unroll
          for (int s = 0; s < 2; ++s) { av[s] = an[s]; bv[s] = bn[s]; }
        }
      }
      // D fragment: lane l, register r <-> row (l>>4)*4 + r, column l&15 of the 16x16 tile
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * fk + r, gr = r0 + row;
        float v = 0.f;
        if (col < N) {
          v = mg_act((acc0[r] + acc1[r]) + bias, S.act, S.slope) * mulv[r];
          if (gr < a.rows && S.out) {
            float* o = S.out + (size_t)gr * S.ld_out + col;
            *o = S.accumulate ? (*o + v) : v;
          }
        }
        if (!last) nxt[row * MC_LDX + col] = v;
      }
    }
    if (!last) {  // zero the padding of the next stage's K: columns 16*ntt .. next multiple of 32
      const int c0 = 16 * ntt, c1 = (c0 + 31) & ~31;
      for (int e = tid; e < MC_ROWS * (c1 - c0); e += 256) {
        const int r = e / (c1 - c0), c = c0 + e - r * (c1 - c0);
        if (c < MC_LDX) nxt[r * MC_LDX + c] = 0.f;
      }
    }
    float* t = act;
    act = nxt;
    nxt = t;
  }
}

extern "C" {

int mggan_mlp_chain(const void* args, hipStream_t stream) {
  MG_CHECK_ARG(args, "mlp_chain: null arguments");
  McArgs a = *(const McArgs*)args;
  MG_CHECK_ARG(a.X && a.n >= 1 && a.n <= 3 && a.rows >= 0, "mlp_chain: bad arguments (n = %d)", a.n);
  MG_CHECK_ARG(a.K0 >= 1 && a.K0 <= MC_MAXD, "mlp_chain: input width %d not in 1..%d", a.K0, MC_MAXD);
  int k = a.K0;
  for (int i = 0; i < a.n; ++i) {
    MG_CHECK_ARG(a.s[i].W && a.s[i].K == k && a.s[i].N >= 1 && a.s[i].N <= MC_MAXD,
                 "mlp_chain: stage %d has K = %d, N = %d (expected K = %d, N <= %d)", i, a.s[i].K, a.s[i].N, k, MC_MAXD);
    k = a.s[i].N;
  }
  MG_CHECK_ARG(a.s[a.n - 1].out, "mlp_chain: the last stage needs an output pointer");
  if (a.rows == 0) return MGGAN_OK;
  MG_LAUNCH(mlp_chain_kernel, dim3(cdiv(a.rows, MC_ROWS)), dim3(256), 0, stream, a);
  MG_LAUNCH_CHECK("mlp_chain");
  return MGGAN_OK;
}

}  // extern "C"
