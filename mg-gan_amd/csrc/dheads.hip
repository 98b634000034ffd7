// Both heads of the discriminator over a LARGE number of (pedestrian, sample) rows -- the generator step's pass over
// K*b rows (25,600 at 64x20, 163,840 at 256x32) -- as ONE weight-stationary launch per direction.
//
// Replaces (file:line under /root/reference/mggan/model/modules/discriminators.py):
//   discs[0]              :76-85,197-204   Linear(192,96) - LeakyReLU(0.2) - Linear(96,1) [- Sigmoid, eps squeeze]
//   gen_id_reconstructor  :97-108,211-219  Linear(192,96) - LeakyReLU(0.2) - Linear(96,g)
// and, in the generator step, their input gradient (the discriminator is frozen there: no weight gradients).
//
// The generic chain kernel (mlp.hip) streams the weights from L2 for every 16-row tile: right for the many small
// launches, but at 25,600+ rows each of the four head launches re-reads 74 KB of weights 1,600+ times and is latency
// bound per tile (31-51 us per launch, 19 % of the f32 peak).  Here the two first layers are ONE 192 -> 192 product
// (hidden = [head A | head B]); a persistent workgroup of four waves keeps its share of the weights as MFMA B
// fragments in registers for its whole life (wave w owns hidden / output columns 48w .. 48w+47: 3 column tiles x 12
// k-super-steps x 4 = 144 registers), walks the row tiles, and only the 12 KB activation tile moves: global ->
// registers (prefetched under the previous tile's products) -> LDS -> A fragments.  The narrow second layers
// (96 -> 1, 96 -> g) are lane-local partial dot products folded with wave shuffles.
#include "common.h"
#include "../../include/mggan_hip.h"

#define DH_IN 192
#define DH_HID 96
#define DH_LDX 196      // LDS row stride of the activation tile (16-byte aligned rows, 4 mod 32)
#define DH_MAXG 16

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct DHeadsArgs {
  const float* X;      // (rows, ldx)            forward input / (backward: unused)
  int ldx, rows, g, act_a;   // act_a: activation of head A's output (ACT_SIGMOID_EPS or ACT_NONE)
  const float *W1a, *b1a, *W2a, *b2a;   // head A: (96,192), (96), (1,96), (1)
  const float *W1b, *b1b, *W2b, *b2b;   // head B: (96,192), (96), (g,96), (g)
  float *Ha, *Hb;      // (rows, 96) hidden activations (forward: written when not NULL; backward: read)
  float *Ya, *Yb;      // (rows, 1), (rows, g)   forward outputs / (backward: Ya read for the output derivative)
  const float *dYa, *dYb;  // backward: gradients of the outputs
  float* dX;           // backward: (rows, ld_dx)
  int ld_dx;
};

// 16 x 192 tile: global rows [r0, r0+16) -> 12 registers per thread -> LDS
__device__ __forceinline__ void dh_fetch(const float* __restrict__ X, int ldx, int rows, int r0, f32x4 v[3]) {
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int q = threadIdx.x + 256 * u, r = q / 48, k = (q - r * 48) * 4;  // 48 quads per row
    v[u] = r0 + r < rows ? *reinterpret_cast<const f32x4*>(X + (size_t)(r0 + r) * ldx + k) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
__device__ __forceinline__ void dh_commit(const f32x4 v[3], float* tile) {
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int q = threadIdx.x + 256 * u, r = q / 48, k = (q - r * 48) * 4;
    *reinterpret_cast<f32x4*>(&tile[r * DH_LDX + k]) = v[u];
  }
}

// GM: compile-time bound of the second-layer width (g <= GM): sizes the register arrays of the narrow second layer
template <int GM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void dheads_fwd_kernel(DHeadsArgs a) {
  __shared__ __attribute__((aligned(16))) float tile[2][16 * DH_LDX];
  __shared__ float part[4][16][DH_MAXG];  // [wave][row][output] partial second-layer sums
  __shared__ float w2l[2][DH_MAXG][DH_HID];  // second-layer weights of both heads (zero rows beyond the head's width)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  const int head = w >> 1, half = w & 1;          // head 0 = A, 1 = B; this wave's 48 hidden columns: 48*half ..
  for (int i = threadIdx.x; i < 2 * DH_MAXG * DH_HID; i += 256) {
    const int hd = i / (DH_MAXG * DH_HID), o = (i / DH_HID) % DH_MAXG, n = i % DH_HID;
    w2l[hd][o][n] = hd == 0 ? (o == 0 ? a.W2a[n] : 0.f) : (o < a.g ? a.W2b[(size_t)o * DH_HID + n] : 0.f);
  }
  const float* W1 = head ? a.W1b : a.W1a;
  const float* b1 = head ? a.b1b : a.b1a;
  const int n2 = head ? a.g : 1;
  // stationary B fragments: column tile j (hidden unit 48*half + 16 j + fi), super-step ss: k = 16 ss + 4 fk + i
  f32x4 bw[3][12];
  float bias1[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int n = 48 * half + 16 * j + fi;
#pragma unroll
    for (int ss = 0; ss < 12; ++ss) bw[j][ss] = *reinterpret_cast<const f32x4*>(W1 + (size_t)n * DH_IN + 16 * ss + 4 * fk);
    bias1[j] = b1[n];
  }
  const int ntiles = (a.rows + 15) / 16;
  f32x4 pre[3];
  if ((int)blockIdx.x < ntiles) dh_fetch(a.X, a.ldx, a.rows, blockIdx.x * 16, pre);
  int buf = 0;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x, buf ^= 1) {
    const int r0 = t * 16;
    float* tl = tile[buf];
    dh_commit(pre, tl);
    __syncthreads();  // tile ready; the previous tile's partial sums have been consumed
    if (t + (int)gridDim.x < ntiles) dh_fetch(a.X, a.ldx, a.rows, (t + gridDim.x) * 16, pre);
    f32x4 acc[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j][0] = acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* arow = &tl[fi * DH_LDX + 4 * fk];
#pragma unroll
    for (int ss = 0; ss < 12; ++ss) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 16 * ss);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        acc[j][0] = MFMA16(av[0], bw[j][ss][0], acc[j][0]);
        acc[j][1] = MFMA16(av[1], bw[j][ss][1], acc[j][1]);
        acc[j][0] = MFMA16(av[2], bw[j][ss][2], acc[j][0]);
        acc[j][1] = MFMA16(av[3], bw[j][ss][3], acc[j][1]);
      }
    }
    // D fragment: lane (fi, fk), register r <-> row 4 fk + r, hidden column 48 half + 16 j + fi
    float p2[4][GM];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int o = 0; o < GM; ++o) p2[r][o] = 0.f;
    float* Hs = head ? a.Hb : a.Ha;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float h = (acc[j][0][r] + acc[j][1][r]) + bias1[j];
        h = h > 0.f ? h : 0.2f * h;  // LeakyReLU(0.2), discriminators.py:80,101
        const int gr = r0 + 4 * fk + r;
        if (Hs && gr < a.rows) Hs[(size_t)gr * DH_HID + 48 * half + 16 * j + fi] = h;
#pragma unroll
        for (int o = 0; o < GM; ++o) p2[r][o] = fmaf(h, w2l[head][o][48 * half + 16 * j + fi], p2[r][o]);
      }
    // fold the 16 column lanes (fixed butterfly), lane fi == 0 of every row group keeps the sum
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int o = 0; o < GM; ++o)
        if (o < n2) {
          const float v = row_sum16(p2[r][o]);  // DPP row reduction (was four ds_bpermute per value: 128 per wave and tile)
          if (fi == 0) part[w][4 * fk + r][o] = v;
        }
    __syncthreads();
    // second-layer outputs: thread (row, o) adds the two halves of its head
    {
      const int row = threadIdx.x & 15, o = threadIdx.x >> 4, gr = r0 + row;
      if (gr < a.rows) {
        if (o == 0) {
          const float y = (part[0][row][0] + part[1][row][0]) + a.b2a[0];
          a.Ya[gr] = mg_act(y, a.act_a, 0.f);
        } else if (o - 1 < a.g) {
          a.Yb[(size_t)gr * a.g + (o - 1)] = (part[2][row][o - 1] + part[3][row][o - 1]) + a.b2b[o - 1];
        }
      }
    }
  }
}

// Input gradient of both heads (no weight gradients): dH = [dza w2a^T | dYb W2b] .* LeakyReLU'(H)  (16 x 192 tile,
// built on the VALU into LDS), then dX = dH [W1a ; W1b]  (192 x 192 stationary B fragments, wave w -> input columns
// 48 w .. 48 w + 47).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void dheads_bwd_kernel(DHeadsArgs a) {
  __shared__ __attribute__((aligned(16))) float tile[2][16 * DH_LDX];
  __shared__ float w2s[(1 + DH_MAXG) * DH_HID];  // w2a (96) then W2b (g x 96)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  const int g = a.g;
  for (int i = threadIdx.x; i < (1 + g) * DH_HID; i += 256) w2s[i] = i < DH_HID ? a.W2a[i] : a.W2b[i - DH_HID];
  // stationary B fragments of dX[r][n] = sum_k dH[r][k] Wcat[k][n], Wcat = [W1a ; W1b] (192 x 192, row-major rows of
  // 192): column tile j (n = 48 w + 16 j + fi), super-step ss: k = 16 ss + 4 fk + i
  f32x4 bw[3][12];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int n = 48 * w + 16 * j + fi;
#pragma unroll
    for (int ss = 0; ss < 12; ++ss) {
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = 16 * ss + 4 * fk + i;
        v[i] = (k < DH_HID ? a.W1a + (size_t)k * DH_IN : a.W1b + (size_t)(k - DH_HID) * DH_IN)[n];
      }
      bw[j][ss] = v;
    }
  }
  const int ntiles = (a.rows + 15) / 16;
  // per thread: 12 entries of the dH tile = 3 quads (row r, k..k+3), k < 96: head A, else head B
  f32x4 hpre[3];
  auto fetch_h = [&](int r0) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int q = threadIdx.x + 256 * u, r = q / 48, k = (q - r * 48) * 4;
      const float* H = k < DH_HID ? a.Ha + k : a.Hb + (k - DH_HID);
      hpre[u] = r0 + r < a.rows ? *reinterpret_cast<const f32x4*>(H + (size_t)(r0 + r) * DH_HID) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  if ((int)blockIdx.x < ntiles) fetch_h(blockIdx.x * 16);
  __syncthreads();
  int buf = 0;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x, buf ^= 1) {
    const int r0 = t * 16;
    float* tl = tile[buf];
    // ---- dH tile ----
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int q = threadIdx.x + 256 * u, r = q / 48, k = (q - r * 48) * 4, gr = r0 + r;
      f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
      if (gr < a.rows) {
        if (k < DH_HID) {
          const float dza = a.dYa[gr] * mg_act_grad_from_out(a.Ya[gr], a.act_a, 0.f);
#pragma unroll
          for (int i = 0; i < 4; ++i) d[i] = dza * w2s[k + i];
        } else {
          for (int o = 0; o < g; ++o) {
            const float dy = a.dYb[(size_t)gr * g + o];
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = fmaf(dy, w2s[DH_HID + o * DH_HID + (k - DH_HID) + i], d[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] *= hpre[u][i] > 0.f ? 1.f : 0.2f;
      }
      *reinterpret_cast<f32x4*>(&tl[r * DH_LDX + k]) = d;
    }
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) fetch_h((t + gridDim.x) * 16);
    f32x4 acc[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j][0] = acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* arow = &tl[fi * DH_LDX + 4 * fk];
#pragma unroll
    for (int ss = 0; ss < 12; ++ss) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 16 * ss);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        acc[j][0] = MFMA16(av[0], bw[j][ss][0], acc[j][0]);
        acc[j][1] = MFMA16(av[1], bw[j][ss][1], acc[j][1]);
        acc[j][0] = MFMA16(av[2], bw[j][ss][2], acc[j][0]);
        acc[j][1] = MFMA16(av[3], bw[j][ss][3], acc[j][1]);
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gr = r0 + 4 * fk + r;
        if (gr < a.rows) a.dX[(size_t)gr * a.ld_dx + 48 * w + 16 * j + fi] = acc[j][0][r] + acc[j][1][r];
      }
  }
}

extern "C" {

static int dheads_check(const DHeadsArgs& a, const char* what) {
  MG_CHECK_ARG(a.rows >= 0 && a.g >= 1 && a.g <= DH_MAXG - 1, "%s: g = %d not in 1..%d", what, a.g, DH_MAXG - 1);
  MG_CHECK_ARG(a.W1a && a.b1a && a.W2a && a.b2a && a.W1b && a.b1b && a.W2b && a.b2b, "%s: null weight pointer", what);
  MG_CHECK_ARG(a.act_a == ACT_NONE || a.act_a == ACT_SIGMOID_EPS || a.act_a == ACT_SIGMOID, "%s: output activation %d", what, a.act_a);
  return MGGAN_OK;
}

static int dheads_grid(int rows) {
  const int nt = (rows + 15) / 16;
  return nt < 512 ? nt : 512;  // two workgroups per CU (2 waves per SIMD at ~200 VGPRs)
}

/* X (rows, ldx >= 192, 16-byte aligned rows) -> Ya (rows,1) = act_a(head A), Yb (rows,g) = head B logits;
 * Ha / Hb (rows,96): the hidden activations, kept when not NULL (a backward pass follows) */
int mggan_dheads_fwd(const float* X, int ldx, int rows, int g, int act_a, const float* W1a, const float* b1a,
                     const float* W2a, const float* b2a, const float* W1b, const float* b1b, const float* W2b,
                     const float* b2b, float* Ha, float* Hb, float* Ya, float* Yb, hipStream_t stream) {
  DHeadsArgs a = {};
  a.X = X; a.ldx = ldx; a.rows = rows; a.g = g; a.act_a = act_a;
  a.W1a = W1a; a.b1a = b1a; a.W2a = W2a; a.b2a = b2a; a.W1b = W1b; a.b1b = b1b; a.W2b = W2b; a.b2b = b2b;
  a.Ha = Ha; a.Hb = Hb; a.Ya = Ya; a.Yb = Yb;
  if (int rc = dheads_check(a, "dheads_fwd")) return rc;
  if (rows == 0) return MGGAN_OK;
  MG_CHECK_ARG(X && Ya && Yb && ldx >= DH_IN && (ldx & 3) == 0 && (((size_t)X) & 15) == 0, "dheads_fwd: bad input / outputs");
  if (g <= 4) hipLaunchKernelGGL(dheads_fwd_kernel<4>, dim3(dheads_grid(rows)), dim3(256), 0, stream, a);
  else if (g <= 8) hipLaunchKernelGGL(dheads_fwd_kernel<8>, dim3(dheads_grid(rows)), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(dheads_fwd_kernel<16>, dim3(dheads_grid(rows)), dim3(256), 0, stream, a);
  MG_LAUNCH_CHECK("dheads_fwd");
  return MGGAN_OK;
}

/* dX (rows, ld_dx) = d(head A)/dX + d(head B)/dX from the output gradients dYa (rows,1), dYb (rows,g) and the saved
 * Ya, Ha, Hb of mggan_dheads_fwd -- the input gradient only (the generator step freezes the discriminator) */
int mggan_dheads_bwd_data(const float* dYa, const float* dYb, const float* Ya, const float* Ha, const float* Hb, int rows,
                          int g, int act_a, const float* W1a, const float* W2a, const float* W1b, const float* W2b,
                          float* dX, int ld_dx, hipStream_t stream) {
  DHeadsArgs a = {};
  a.rows = rows; a.g = g; a.act_a = act_a;
  a.W1a = W1a; a.W2a = W2a; a.W1b = W1b; a.W2b = W2b;
  a.b1a = a.b2a = a.b1b = a.b2b = W1a;  // unused by the backward kernel
  a.Ha = const_cast<float*>(Ha); a.Hb = const_cast<float*>(Hb); a.Ya = const_cast<float*>(Ya);
  a.dYa = dYa; a.dYb = dYb; a.dX = dX; a.ld_dx = ld_dx;
  if (int rc = dheads_check(a, "dheads_bwd_data")) return rc;
  if (rows == 0) return MGGAN_OK;
  MG_CHECK_ARG(dYa && dYb && Ya && Ha && Hb && dX && ld_dx >= DH_IN, "dheads_bwd_data: null pointer");
  hipLaunchKernelGGL(dheads_bwd_kernel, dim3(dheads_grid(rows)), dim3(256), 0, stream, a);
  MG_LAUNCH_CHECK("dheads_bwd_data");
  return MGGAN_OK;
}

}  // extern "C"
