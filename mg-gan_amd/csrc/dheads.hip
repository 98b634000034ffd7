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
  // backward with trainable heads (the discriminator step's pair pass): head B exists for rows >= row0_b only (Hb / dYb
  // are indexed from there), and the gate gradients the weight-gradient GEMMs need are written out
  int row0_b;
  float* dH_out;       // (rows, 192) = [dZ1 of head A | dZ1 of head B], or NULL
  float* dza_out;      // (rows) = dYa * act'(Ya), or NULL
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
      const bool isb = k >= DH_HID;
      const float* H = isb ? a.Hb + (k - DH_HID) : a.Ha + k;
      const int hr = r0 + r - (isb ? a.row0_b : 0);
      hpre[u] = (r0 + r < a.rows && hr >= 0) ? *reinterpret_cast<const f32x4*>(H + (size_t)hr * DH_HID) : f32x4{0.f, 0.f, 0.f, 0.f};
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
          if (a.dza_out && k == 0) a.dza_out[gr] = dza;
#pragma unroll
          for (int i = 0; i < 4; ++i) d[i] = dza * w2s[k + i];
        } else if (gr >= a.row0_b) {
          for (int o = 0; o < g; ++o) {
            const float dy = a.dYb[(size_t)(gr - a.row0_b) * g + o];
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = fmaf(dy, w2s[DH_HID + o * DH_HID + (k - DH_HID) + i], d[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] *= hpre[u][i] > 0.f ? 1.f : 0.2f;
        if (a.dH_out) *reinterpret_cast<f32x4*>(a.dH_out + (size_t)gr * DH_IN + k) = d;
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


// ---------------------------------------------------------------------------------------------------------------------
// Lean form for the sample blocks k >= 1 of a K-sample pass with a frozen discriminator (the generator step).
// A classifier row (k, ped) is [soc | in_enc | pred_enc | scene] (discriminators.py:179-196): in_enc and scene are
// the pedestrian's, the same in all K blocks; the social block is zero from block 1 on (the list-repeat of
// seq_start_end); only the 32 pred_enc columns differ from row to row.  With
//     P[ped] = b1 + W1[:, in_enc] in_enc(ped) + W1[:, scene] scene(ped)            (once per pedestrian, 96 -> 192)
// the first layers of both heads are   hidden(k, ped) = P[ped] + W1[:, pred_enc] pred_enc(k, ped)   for k >= 1:
// K = 32 instead of 192 in the row product (96 MFMAs per 16-row tile instead of 576), and the rows never have to be
// assembled: X is read in its pred_enc block only.  The backward pass wants d pred_enc alone (in_enc and scene carry no
// gradient when the discriminator is frozen): 192 -> 32, again 96 MFMAs.  LeakyReLU's derivative travels as one bit
// per hidden unit (a 64-bit word per lane and tile) instead of the 768 B per row of saved activations.
// Every wave owns whole tiles (no workgroup-level exchange): products in the transposed orientation
// D[hidden][row] so that the P tile, the input gradient and the stationary weights are all 16-byte accesses.
// Block 0 (rows [0, b), the rows with social features) goes through the full-width kernels above.
struct DLeanArgs {
  const float* X;       // (rows, ldx): only columns c_pe .. c_pe+31 of rows >= row0 are read
  int ldx, c_pe, row0, rows, b, g, act_a;
  const float* P;       // (b, 192)
  const float *W1a, *W1b, *W2a, *b2a, *W2b, *b2b;
  unsigned long long* mask;  // [(tile * 64 + lane)]: bit 4 j + r of lane (fi, fk) <-> hidden unit 16 j + 4 fk + r of row fi
  float *Ya, *Yb;
  const float *dYa, *dYb;
  float* dX;
  int ld_dx;
};
struct DSharedArgs {
  const float* X;       // block-0 rows (b, ldx) with the in_enc (32 wide at c_in) and scene (64 wide at c_sc) blocks filled
  int ldx, b, c_in, c_sc;
  const float *W1a, *b1a, *W1b, *b1b;
  float* P;
};

// P tile of 16 pedestrians: wave w -> hidden units 48 w .. 48 w + 47
__global__ __launch_bounds__(256) void dheads_shared_kernel(DSharedArgs a) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  const int ped = blockIdx.x * 16 + fi;
  const bool valid = ped < a.b;
  f32x4 xv[6];
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const int col = (s < 2 ? a.c_in + 16 * s : a.c_sc + 16 * (s - 2)) + 4 * fk;
    xv[s] = valid ? *reinterpret_cast<const f32x4*>(a.X + (size_t)ped * a.ldx + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int n = 48 * w + 16 * j + fi;
    const float* Wrow = n < DH_HID ? a.W1a + (size_t)n * DH_IN : a.W1b + (size_t)(n - DH_HID) * DH_IN;
    f32x4 wv[6];
#pragma unroll
    for (int s = 0; s < 6; ++s)
      wv[s] = *reinterpret_cast<const f32x4*>(Wrow + (s < 2 ? a.c_in + 16 * s : a.c_sc + 16 * (s - 2)) + 4 * fk);
    const int h0 = 48 * w + 16 * j + 4 * fk;  // this lane's four hidden units (D rows 4 fk + r)
    f32x4 acc0 = *reinterpret_cast<const f32x4*>((h0 < DH_HID ? a.b1a + h0 : a.b1b + (h0 - DH_HID)));
    f32x4 acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      acc0 = MFMA16(wv[s][0], xv[s][0], acc0);
      acc1 = MFMA16(wv[s][1], xv[s][1], acc1);
      acc0 = MFMA16(wv[s][2], xv[s][2], acc0);
      acc1 = MFMA16(wv[s][3], xv[s][3], acc1);
    }
    if (valid) *reinterpret_cast<f32x4*>(a.P + (size_t)ped * DH_IN + h0) = acc0 + acc1;
  }
}

__device__ __forceinline__ void dl_stage_w2(const DLeanArgs& a, float* w2l) {
  for (int i = threadIdx.x; i < (1 + a.g) * DH_HID; i += 256) w2l[i] = i < DH_HID ? a.W2a[i] : a.W2b[i - DH_HID];
  __syncthreads();
}

template <int GM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void dheads_lean_fwd_kernel(DLeanArgs a) {
  __shared__ __attribute__((aligned(16))) float w2l[(1 + DH_MAXG) * DH_HID];  // w2a (96) then W2b (g x 96)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  // A operands: W1cat[:, pred_enc] (192 x 32) from LDS (row stride 36: the 16-byte reads of a 16-lane group start 4 banks
  // apart) -- as 96 stationary registers next to the 48 accumulators and the prefetched P tile they spilled
  __shared__ __attribute__((aligned(16))) float w1s[2 * DH_HID * 36];
  for (int i = threadIdx.x; i < 2 * DH_HID * 8; i += 256) {
    const int n = i >> 3, q = (i & 7) * 4;
    const float* Wrow = (n < DH_HID ? a.W1a + (size_t)n * DH_IN : a.W1b + (size_t)(n - DH_HID) * DH_IN) + a.c_pe + q;
    *reinterpret_cast<f32x4*>(&w1s[n * 36 + q]) = *reinterpret_cast<const f32x4*>(Wrow);
  }
  dl_stage_w2(a, w2l);
  const float* awl = &w1s[fi * 36 + 4 * fk];  // hidden tile j, k block ss: awl[16 j * 36 + 16 ss]
  const float b2a = a.b2a[0];
  const int nt = (a.rows - a.row0 + 15) / 16;
  const int nwv = gridDim.x * 4;
  f32x4 nx[2], nP[12];
  auto fetch = [&](int t) {
    const int gr = a.row0 + 16 * t + fi;
    const bool valid = gr < a.rows;
    const int grc = valid ? gr : a.rows - 1;  // (clamped address, masked where it is consumed)
    const float* xr = a.X + (size_t)grc * a.ldx + a.c_pe + 4 * fk;
    nx[0] = *reinterpret_cast<const f32x4*>(xr);
    nx[1] = *reinterpret_cast<const f32x4*>(xr + 16);
    const float* pr = a.P + (size_t)(grc % a.b) * DH_IN + 4 * fk;
#pragma unroll
    for (int j = 0; j < 12; ++j) nP[j] = *reinterpret_cast<const f32x4*>(pr + 16 * j);
  };
  int t = blockIdx.x * 4 + w;
  if (t < nt) fetch(t);
  for (; t < nt; t += nwv) {
    const int gr = a.row0 + 16 * t + fi;
    const bool valid = gr < a.rows;
    f32x4 acc[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[j] = nP[j];
    const f32x4 x0 = nx[0], x1 = nx[1];
    if (t + nwv < nt) fetch(t + nwv);
    const float* awt = awl;
    asm volatile("" : "+v"(awt));  // (the reads are loop invariant: hoisted, they are the 96 registers again)
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(awt + 16 * j * 36), a1 = *reinterpret_cast<const f32x4*>(awt + 16 * j * 36 + 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j] = MFMA16(a0[i], x0[i], acc[j]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j] = MFMA16(a1[i], x1[i], acc[j]);
    }
    // lane (fi, fk): hidden units 16 j + 4 fk + r of row fi
    unsigned long long m = 0ull;
    float ya = 0.f, yb[GM];
#pragma unroll
    for (int o = 0; o < GM; ++o) yb[o] = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      f32x4 h;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[j][r];
        const bool pos = v > 0.f;
        h[r] = pos ? v : 0.2f * v;  // LeakyReLU(0.2), discriminators.py:80,101
        m |= (unsigned long long)pos << (4 * j + r);
      }
      if (j < 6) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(&w2l[16 * j + 4 * fk]);
        ya = fmaf(h[0], wv[0], fmaf(h[1], wv[1], fmaf(h[2], wv[2], fmaf(h[3], wv[3], ya))));
      } else {
#pragma unroll
        for (int o = 0; o < GM; ++o)
          if (o < a.g) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(&w2l[DH_HID + o * DH_HID + 16 * (j - 6) + 4 * fk]);
            yb[o] = fmaf(h[0], wv[0], fmaf(h[1], wv[1], fmaf(h[2], wv[2], fmaf(h[3], wv[3], yb[o]))));
          }
      }
    }
    ya = quarters_sum(ya);
#pragma unroll
    for (int o = 0; o < GM; ++o)
      if (o < a.g) yb[o] = quarters_sum(yb[o]);
    if (valid && fk == 0) {
      a.Ya[gr] = mg_act(ya + b2a, a.act_a, 0.f);
#pragma unroll
      for (int o = 0; o < GM; ++o)
        if (o < a.g) a.Yb[(size_t)gr * a.g + o] = yb[o] + a.b2b[o];
    }
    if (a.mask) a.mask[(size_t)t * 64 + lane] = m;
  }
}

// d pred_enc^T [32 x rows] = W1cat[:, pred_enc]^T dH^T, dH = [dza w2a^T | dYb W2b] .* LeakyReLU'(hidden) built per lane
template <int GM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void dheads_lean_bwd_kernel(DLeanArgs a) {
  __shared__ __attribute__((aligned(16))) float w2l[(1 + DH_MAXG) * DH_HID];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  dl_stage_w2(a, w2l);
  // stationary A operands: row = pred_enc column 16 ct + fi, k = hidden unit 16 ss + 4 fk + i
  float aw[2][48];
#pragma unroll
  for (int ss = 0; ss < 12; ++ss)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int hid = 16 * ss + 4 * fk + i;
      const float* Wrow = (hid < DH_HID ? a.W1a + (size_t)hid * DH_IN : a.W1b + (size_t)(hid - DH_HID) * DH_IN) + a.c_pe + fi;
      aw[0][4 * ss + i] = Wrow[0];
      aw[1][4 * ss + i] = Wrow[16];
    }
  const int nt = (a.rows - a.row0 + 15) / 16;
  const int nwv = gridDim.x * 4;
  for (int t = blockIdx.x * 4 + w; t < nt; t += nwv) {
    const int gr = a.row0 + 16 * t + fi;
    const bool valid = gr < a.rows;
    const int grc = valid ? gr : a.rows - 1;
    const unsigned long long m = a.mask[(size_t)t * 64 + lane];
    const float vm = valid ? 1.f : 0.f;
    const float dza = a.dYa[grc] * mg_act_grad_from_out(a.Ya[grc], a.act_a, 0.f) * vm;
    float dyb[GM];
#pragma unroll
    for (int o = 0; o < GM; ++o) dyb[o] = o < a.g ? a.dYb[(size_t)grc * a.g + o] * vm : 0.f;
    f32x4 acc[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) acc[ct][0] = acc[ct][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ss = 0; ss < 12; ++ss) {
      f32x4 d;
      if (ss < 6) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(&w2l[16 * ss + 4 * fk]);
        d = wv * dza;
      } else {
        d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < GM; ++o)
          if (o < a.g) d += *reinterpret_cast<const f32x4*>(&w2l[DH_HID + o * DH_HID + 16 * (ss - 6) + 4 * fk]) * dyb[o];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dv = d[i] * (((m >> (4 * ss + i)) & 1ull) ? 1.f : 0.2f);
        acc[0][i & 1] = MFMA16(aw[0][4 * ss + i], dv, acc[0][i & 1]);
        acc[1][i & 1] = MFMA16(aw[1][4 * ss + i], dv, acc[1][i & 1]);
      }
    }
    if (valid) {
      float* o = a.dX + (size_t)gr * a.ld_dx + a.c_pe + 4 * fk;
      *reinterpret_cast<f32x4*>(o) = acc[0][0] + acc[0][1];
      *reinterpret_cast<f32x4*>(o + 16) = acc[1][0] + acc[1][1];
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same lean pass taken one stage further back: the rows of the blocks k >= 1 go from the predicted steps straight
// to the two head outputs in ONE launch, and from the output gradients straight to the gradient of the steps in one.
//   x (24 = 2 T) -> pred_encoder: Linear(24,64) - LeakyReLU(0.2) - Linear(64,32)  (discriminators.py:42-43,129-131)
//                -> hidden = P[ped] + W1cat[:, pred_enc] pred_enc -> LeakyReLU(0.2) -> second layers
// as a chain of transposed products  Y^T = W X^T : the D fragment of one product (lane (fi, fk): units 16 j + 4 fk + r of
// row fi) IS the B operand of the next one when its k-step (j, r) is read as unit 16 j + 4 fk + r, so the activations
// never leave the registers - no (K b, 24) row copy of the steps, no (K b, 32) pred_enc block, no (K b, 64) hidden
// activations, and their three launches (steps_to_rows, the pred_encoder chain, rows_to_steps on the way back) are
// gone.  The frozen pred_encoder keeps nothing but 16 more sign bits in the mask word (bits 48 + 4 j + r).
struct DRowsLeanArgs {
  DLeanArgs h;            // heads part: P, W1*, W2*, mask, Ya / Yb, dYa / dYb, row0, rows, b, g, act_a, c_pe (X / dX unused)
  const float* pred;      // (T = 12, rows, 2) steps, row = k*b + ped
  const float *Wp1, *bp1, *Wp2, *bp2;  // (64,24), (64), (32,64), (32)
  float* dpred;           // (12, rows, 2)
};
#define DR_T 12
#define DR_HP 64
#define DR_LD1 28   // wp1 rows: 24 coefficients in lane order, 8-byte aligned
#define DR_LD2 68

template <int GM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void d_rows_lean_fwd_kernel(DRowsLeanArgs q) {
  const DLeanArgs& a = q.h;
  __shared__ __attribute__((aligned(16))) float w2l[(1 + DH_MAXG) * DH_HID];
  __shared__ __attribute__((aligned(16))) float w1s[2 * DH_HID * 36];
  __shared__ __attribute__((aligned(16))) float wp1[DR_HP * DR_LD1];   // [unit][6 fk + s]: coefficient of step t = fk + 4 (s >> 1), component s & 1
  __shared__ __attribute__((aligned(16))) float wp2[32 * DR_LD2];      // [pred_enc column][unit]
  __shared__ __attribute__((aligned(16))) float bp[DR_HP + 32];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  for (int i = threadIdx.x; i < 2 * DH_HID * 8; i += 256) {
    const int n = i >> 3, c = (i & 7) * 4;
    const float* Wrow = (n < DH_HID ? a.W1a + (size_t)n * DH_IN : a.W1b + (size_t)(n - DH_HID) * DH_IN) + a.c_pe + c;
    *reinterpret_cast<f32x4*>(&w1s[n * 36 + c]) = *reinterpret_cast<const f32x4*>(Wrow);
  }
  for (int i = threadIdx.x; i < DR_HP * 24; i += 256) {
    const int n = i / 24, p = i % 24, k4 = p / 6, s = p % 6;
    wp1[n * DR_LD1 + p] = q.Wp1[n * 24 + 2 * (k4 + 4 * (s >> 1)) + (s & 1)];
  }
  for (int i = threadIdx.x; i < 32 * DR_HP; i += 256) wp2[(i / DR_HP) * DR_LD2 + i % DR_HP] = q.Wp2[i];
  for (int i = threadIdx.x; i < DR_HP + 32; i += 256) bp[i] = i < DR_HP ? q.bp1[i] : q.bp2[i - DR_HP];
  dl_stage_w2(a, w2l);
  const float* awl = &w1s[fi * 36 + 4 * fk];
  const float b2a = a.b2a[0];
  const int nt = (a.rows - a.row0 + 15) / 16;
  const int nwv = gridDim.x * 4;
  float2 nx[3];
  f32x4 nP[12];
  auto fetch = [&](int t) {
    const int gr = a.row0 + 16 * t + fi;
    const int grc = gr < a.rows ? gr : a.rows - 1;  // (clamped address, masked where it is consumed)
#pragma unroll
    for (int u = 0; u < 3; ++u) nx[u] = *reinterpret_cast<const float2*>(q.pred + ((size_t)(fk + 4 * u) * a.rows + grc) * 2);
    const float* pr = a.P + (size_t)(grc % a.b) * DH_IN + 4 * fk;
#pragma unroll
    for (int j = 0; j < 12; ++j) nP[j] = *reinterpret_cast<const f32x4*>(pr + 16 * j);
  };
  int t = blockIdx.x * 4 + w;
  if (t < nt) fetch(t);
  for (; t < nt; t += nwv) {
    const int gr = a.row0 + 16 * t + fi;
    const bool valid = gr < a.rows;
    f32x4 acc[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[j] = nP[j];
    const float xs[6] = {nx[0].x, nx[0].y, nx[1].x, nx[1].y, nx[2].x, nx[2].y};
    if (t + nwv < nt) fetch(t + nwv);
    const float* lw = &w1s[0];
    asm volatile("" : "+v"(lw));  // (every LDS operand below is loop invariant: hoisted, they are 200 registers)
    const long sh = lw - &w1s[0];
    unsigned long long m = 0ull;
    // pred_encoder, first layer: units 16 j + 4 fk + r of row fi
    f32x4 h1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* wr = &wp1[(16 * j + fi) * DR_LD1 + 6 * fk] + sh;
      const float2 c0 = *reinterpret_cast<const float2*>(wr), c1 = *reinterpret_cast<const float2*>(wr + 2),
                   c2 = *reinterpret_cast<const float2*>(wr + 4);
      f32x4 z = *reinterpret_cast<const f32x4*>(&bp[16 * j + 4 * fk] + sh);
      z = MFMA16(c0.x, xs[0], z); z = MFMA16(c0.y, xs[1], z);
      z = MFMA16(c1.x, xs[2], z); z = MFMA16(c1.y, xs[3], z);
      z = MFMA16(c2.x, xs[4], z); z = MFMA16(c2.y, xs[5], z);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool pos = z[r] > 0.f;
        h1[j][r] = pos ? z[r] : 0.2f * z[r];
        m |= (unsigned long long)pos << (48 + 4 * j + r);
      }
    }
    // second layer: pred_enc columns 16 ct + 4 fk + r (two accumulator chains per column tile)
    f32x4 pe[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      f32x4 za = *reinterpret_cast<const f32x4*>(&bp[DR_HP + 16 * ct + 4 * fk] + sh), zb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const f32x4 wa = *reinterpret_cast<const f32x4*>(&wp2[(16 * ct + fi) * DR_LD2 + 16 * j + 4 * fk] + sh);
        const f32x4 wb = *reinterpret_cast<const f32x4*>(&wp2[(16 * ct + fi) * DR_LD2 + 16 * (j + 1) + 4 * fk] + sh);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          za = MFMA16(wa[r], h1[j][r], za);
          zb = MFMA16(wb[r], h1[j + 1][r], zb);
        }
      }
      pe[ct] = za + zb;
    }
    // first layers of both heads on top of the pedestrian's shared part
    const float* awt = awl + sh;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(awt + 16 * j * 36), a1 = *reinterpret_cast<const f32x4*>(awt + 16 * j * 36 + 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j] = MFMA16(a0[i], pe[0][i], acc[j]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j] = MFMA16(a1[i], pe[1][i], acc[j]);
    }
    float ya = 0.f, yb[GM];
#pragma unroll
    for (int o = 0; o < GM; ++o) yb[o] = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      f32x4 h;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[j][r];
        const bool pos = v > 0.f;
        h[r] = pos ? v : 0.2f * v;
        m |= (unsigned long long)pos << (4 * j + r);
      }
      if (j < 6) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(&w2l[16 * j + 4 * fk]);
        ya = fmaf(h[0], wv[0], fmaf(h[1], wv[1], fmaf(h[2], wv[2], fmaf(h[3], wv[3], ya))));
      } else {
#pragma unroll
        for (int o = 0; o < GM; ++o)
          if (o < a.g) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(&w2l[DH_HID + o * DH_HID + 16 * (j - 6) + 4 * fk]);
            yb[o] = fmaf(h[0], wv[0], fmaf(h[1], wv[1], fmaf(h[2], wv[2], fmaf(h[3], wv[3], yb[o]))));
          }
      }
    }
    ya = quarters_sum(ya);
#pragma unroll
    for (int o = 0; o < GM; ++o)
      if (o < a.g) yb[o] = quarters_sum(yb[o]);
    if (valid && fk == 0) {
      a.Ya[gr] = mg_act(ya + b2a, a.act_a, 0.f);
#pragma unroll
      for (int o = 0; o < GM; ++o)
        if (o < a.g) a.Yb[(size_t)gr * a.g + o] = yb[o] + a.b2b[o];
    }
    if (a.mask) a.mask[(size_t)t * 64 + lane] = m;
  }
}

template <int GM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void d_rows_lean_bwd_kernel(DRowsLeanArgs q) {
  const DLeanArgs& a = q.h;
  __shared__ __attribute__((aligned(16))) float w2l[(1 + DH_MAXG) * DH_HID];
  __shared__ __attribute__((aligned(16))) float wp2t[DR_HP * 36];      // [unit][pred_enc column]
  __shared__ __attribute__((aligned(16))) float wp1t[32 * DR_LD2];     // [step coefficient 2 t + c (rows 24 .. 31 zero)][unit]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  for (int i = threadIdx.x; i < 32 * DR_HP; i += 256) wp2t[(i % DR_HP) * 36 + i / DR_HP] = q.Wp2[i];
  for (int i = threadIdx.x; i < 32 * DR_HP; i += 256) {
    const int f = i / DR_HP, n = i % DR_HP;
    wp1t[f * DR_LD2 + n] = f < 24 ? q.Wp1[n * 24 + f] : 0.f;
  }
  dl_stage_w2(a, w2l);
  float aw[2][48];
#pragma unroll
  for (int ss = 0; ss < 12; ++ss)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int hid = 16 * ss + 4 * fk + i;
      const float* Wrow = (hid < DH_HID ? a.W1a + (size_t)hid * DH_IN : a.W1b + (size_t)(hid - DH_HID) * DH_IN) + a.c_pe + fi;
      aw[0][4 * ss + i] = Wrow[0];
      aw[1][4 * ss + i] = Wrow[16];
    }
  const int nt = (a.rows - a.row0 + 15) / 16;
  const int nwv = gridDim.x * 4;
  for (int t = blockIdx.x * 4 + w; t < nt; t += nwv) {
    const int gr = a.row0 + 16 * t + fi;
    const bool valid = gr < a.rows;
    const int grc = valid ? gr : a.rows - 1;
    const unsigned long long m = a.mask[(size_t)t * 64 + lane];
    const float vm = valid ? 1.f : 0.f;
    const float dza = a.dYa[grc] * mg_act_grad_from_out(a.Ya[grc], a.act_a, 0.f) * vm;
    float dyb[GM];
#pragma unroll
    for (int o = 0; o < GM; ++o) dyb[o] = o < a.g ? a.dYb[(size_t)grc * a.g + o] * vm : 0.f;
    f32x4 acc[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) acc[ct][0] = acc[ct][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ss = 0; ss < 12; ++ss) {
      f32x4 d;
      if (ss < 6) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(&w2l[16 * ss + 4 * fk]);
        d = wv * dza;
      } else {
        d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < GM; ++o)
          if (o < a.g) d += *reinterpret_cast<const f32x4*>(&w2l[DH_HID + o * DH_HID + 16 * (ss - 6) + 4 * fk]) * dyb[o];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dv = d[i] * (((m >> (4 * ss + i)) & 1ull) ? 1.f : 0.2f);
        acc[0][i & 1] = MFMA16(aw[0][4 * ss + i], dv, acc[0][i & 1]);
        acc[1][i & 1] = MFMA16(aw[1][4 * ss + i], dv, acc[1][i & 1]);
      }
    }
    const f32x4 dpe[2] = {acc[0][0] + acc[0][1], acc[1][0] + acc[1][1]};  // pred_enc columns 16 ct + 4 fk + r of row fi
    const float* lw = &wp2t[0];
    asm volatile("" : "+v"(lw));
    const long sh = lw - &wp2t[0];
    // pred_encoder adjoint: d h1 = W_p2^T d pred_enc, through LeakyReLU', d x = W_p1^T d h1
    f32x4 dh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(&wp2t[(16 * j + fi) * 36 + 4 * fk] + sh);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(&wp2t[(16 * j + fi) * 36 + 16 + 4 * fk] + sh);
      f32x4 za = f32x4{0.f, 0.f, 0.f, 0.f}, zb = za;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        za = MFMA16(w0[r], dpe[0][r], za);
        zb = MFMA16(w1[r], dpe[1][r], zb);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) dh[j][r] = (za[r] + zb[r]) * (((m >> (48 + 4 * j + r)) & 1ull) ? 1.f : 0.2f);
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      f32x4 za = f32x4{0.f, 0.f, 0.f, 0.f}, zb = za;
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const f32x4 wa = *reinterpret_cast<const f32x4*>(&wp1t[(16 * mt + fi) * DR_LD2 + 16 * j + 4 * fk] + sh);
        const f32x4 wb = *reinterpret_cast<const f32x4*>(&wp1t[(16 * mt + fi) * DR_LD2 + 16 * (j + 1) + 4 * fk] + sh);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          za = MFMA16(wa[r], dh[j][r], za);
          zb = MFMA16(wb[r], dh[j + 1][r], zb);
        }
      }
      const f32x4 dx = za + zb;  // coefficients 16 mt + 4 fk + r = steps 8 mt + 2 fk (+1), both components
      const int t0 = 8 * mt + 2 * fk;
      if (valid && t0 < DR_T) {
        *reinterpret_cast<float2*>(q.dpred + ((size_t)t0 * a.rows + gr) * 2) = float2{dx[0], dx[1]};
        *reinterpret_cast<float2*>(q.dpred + ((size_t)(t0 + 1) * a.rows + gr) * 2) = float2{dx[2], dx[3]};
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// pred_encoder alone (discriminators.py:42-43,129-131), from the time-major steps straight into its column block of the
// classifier input: the first two products of the chain above as their own launch, for the passes that go on through the
// generic kernels (the discriminator step's real/fake pair pass, sample block 0 of the generator step).  Replaces
// steps_to_rows + a 3-stage mlp_chain launch (11 + 15 us in the configs[1] graph); keeps the row copy of the steps and the
// hidden layer when a backward pass follows (operands of the weight gradients).
struct PredEncArgs {
  const float *a, *b2;   // steps (T = 12, n_stride, 2): rows [0, rows_a) from a, rows [rows_a, rows) from b2
  int n_stride, rows_a, rows;
  const float *Wp1, *bp1, *Wp2, *bp2;
  float* X;              // (rows, ldx): columns c_pe .. c_pe+31 receive pred_enc
  int ldx, c_pe;
  float *h1, *xrows;     // (rows, 64), (rows, 24) or NULL
};
__global__ __launch_bounds__(256) void pred_encoder_fwd_kernel(PredEncArgs q) {
  __shared__ __attribute__((aligned(16))) float wp1[DR_HP * DR_LD1];
  __shared__ __attribute__((aligned(16))) float wp2[32 * DR_LD2];
  __shared__ __attribute__((aligned(16))) float bp[DR_HP + 32];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, fk = lane >> 4;
  for (int i = threadIdx.x; i < DR_HP * 24; i += 256) {
    const int n = i / 24, p = i % 24, k4 = p / 6, s = p % 6;
    wp1[n * DR_LD1 + p] = q.Wp1[n * 24 + 2 * (k4 + 4 * (s >> 1)) + (s & 1)];
  }
  for (int i = threadIdx.x; i < 32 * DR_HP; i += 256) wp2[(i / DR_HP) * DR_LD2 + i % DR_HP] = q.Wp2[i];
  for (int i = threadIdx.x; i < DR_HP + 32; i += 256) bp[i] = i < DR_HP ? q.bp1[i] : q.bp2[i - DR_HP];
  __syncthreads();
  const int nt = (q.rows + 15) / 16;
  for (int t = blockIdx.x * 4 + w; t < nt; t += gridDim.x * 4) {
    const int gr = 16 * t + fi;
    const bool valid = gr < q.rows;
    const int grc = valid ? gr : q.rows - 1;
    const float* src = grc < q.rows_a ? q.a + (size_t)grc * 2 : q.b2 + (size_t)(grc - q.rows_a) * 2;
    float2 x[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) x[u] = *reinterpret_cast<const float2*>(src + (size_t)(fk + 4 * u) * q.n_stride * 2);
    if (q.xrows && valid) {
#pragma unroll
      for (int u = 0; u < 3; ++u) *reinterpret_cast<float2*>(q.xrows + (size_t)gr * 24 + 2 * (fk + 4 * u)) = x[u];
    }
    const float xs[6] = {x[0].x, x[0].y, x[1].x, x[1].y, x[2].x, x[2].y};
    f32x4 h1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* wr = &wp1[(16 * j + fi) * DR_LD1 + 6 * fk];
      const float2 c0 = *reinterpret_cast<const float2*>(wr), c1 = *reinterpret_cast<const float2*>(wr + 2),
                   c2 = *reinterpret_cast<const float2*>(wr + 4);
      f32x4 z = *reinterpret_cast<const f32x4*>(&bp[16 * j + 4 * fk]);
      z = MFMA16(c0.x, xs[0], z); z = MFMA16(c0.y, xs[1], z);
      z = MFMA16(c1.x, xs[2], z); z = MFMA16(c1.y, xs[3], z);
      z = MFMA16(c2.x, xs[4], z); z = MFMA16(c2.y, xs[5], z);
#pragma unroll
      for (int r = 0; r < 4; ++r) h1[j][r] = z[r] > 0.f ? z[r] : 0.2f * z[r];
      if (q.h1 && valid) *reinterpret_cast<f32x4*>(q.h1 + (size_t)gr * DR_HP + 16 * j + 4 * fk) = h1[j];
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      f32x4 za = *reinterpret_cast<const f32x4*>(&bp[DR_HP + 16 * ct + 4 * fk]), zb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const f32x4 wa = *reinterpret_cast<const f32x4*>(&wp2[(16 * ct + fi) * DR_LD2 + 16 * j + 4 * fk]);
        const f32x4 wb = *reinterpret_cast<const f32x4*>(&wp2[(16 * ct + fi) * DR_LD2 + 16 * (j + 1) + 4 * fk]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          za = MFMA16(wa[r], h1[j][r], za);
          zb = MFMA16(wb[r], h1[j + 1][r], zb);
        }
      }
      if (valid) *reinterpret_cast<f32x4*>(q.X + (size_t)gr * q.ldx + q.c_pe + 16 * ct + 4 * fk) = za + zb;
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
  if (g <= 4) MG_LAUNCH(dheads_fwd_kernel<4>, dim3(dheads_grid(rows)), dim3(256), 0, stream, a);
  else if (g <= 8) MG_LAUNCH(dheads_fwd_kernel<8>, dim3(dheads_grid(rows)), dim3(256), 0, stream, a);
  else MG_LAUNCH(dheads_fwd_kernel<16>, dim3(dheads_grid(rows)), dim3(256), 0, stream, a);
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
  MG_LAUNCH(dheads_bwd_kernel, dim3(dheads_grid(rows)), dim3(256), 0, stream, a);
  MG_LAUNCH_CHECK("dheads_bwd_data");
  return MGGAN_OK;
}


static int dlean_grid(int rows) {
  const int nw = (rows + 15) / 16;     // one wave per 16-row tile
  const int wg = (nw + 3) / 4;
  return wg < 512 ? wg : 512;
}

/* mggan_dheads_bwd_data for TRAINABLE heads over a pair pass: head B covers rows [row0_b, rows) only (Hb, dYb start
 * there); besides dX the launch leaves dH (rows,192) = the first layers' gate gradients [head A | head B] and dza (rows) =
 * dYa * act'(Ya), the operands of the four weight-gradient products (mggan_wgrad*) */
int mggan_dheads_bwd_train(const float* dYa, const float* dYb, const float* Ya, const float* Ha, const float* Hb, int rows,
                           int row0_b, int g, int act_a, const float* W1a, const float* W2a, const float* W1b,
                           const float* W2b, float* dX, int ld_dx, float* dH, float* dza, hipStream_t stream) {
  DHeadsArgs a = {};
  a.rows = rows; a.g = g; a.act_a = act_a; a.row0_b = row0_b;
  a.W1a = W1a; a.W2a = W2a; a.W1b = W1b; a.W2b = W2b;
  a.b1a = a.b2a = a.b1b = a.b2b = W1a;  // unused by the backward kernel
  a.Ha = const_cast<float*>(Ha); a.Hb = const_cast<float*>(Hb); a.Ya = const_cast<float*>(Ya);
  a.dYa = dYa; a.dYb = dYb; a.dX = dX; a.ld_dx = ld_dx; a.dH_out = dH; a.dza_out = dza;
  if (int rc = dheads_check(a, "dheads_bwd_train")) return rc;
  if (rows == 0) return MGGAN_OK;
  MG_CHECK_ARG(dYa && dYb && Ya && Ha && Hb && dX && dH && dza && ld_dx >= DH_IN && row0_b >= 0 && row0_b <= rows &&
                   (((size_t)dH) & 15) == 0,
               "dheads_bwd_train: bad arguments");
  MG_LAUNCH(dheads_bwd_kernel, dim3(dheads_grid(rows)), dim3(256), 0, stream, a);
  MG_LAUNCH_CHECK("dheads_bwd_train");
  return MGGAN_OK;
}

/* P (b,192) = [b1a ; b1b] + W1cat[:, c_in .. c_in+31] X[:, c_in ..] + W1cat[:, c_sc .. c_sc+63] X[:, c_sc ..] over the block-0
 * rows of X: the part of both heads' first layers that the K sample blocks of a pedestrian share */
int mggan_dheads_shared(const float* X, int ldx, int b, int c_in, int c_sc, const float* W1a, const float* b1a,
                        const float* W1b, const float* b1b, float* P, hipStream_t stream) {
  if (b == 0) return MGGAN_OK;
  MG_CHECK_ARG(X && W1a && b1a && W1b && b1b && P, "dheads_shared: null pointer");
  MG_CHECK_ARG(ldx >= DH_IN && (ldx & 3) == 0 && (((size_t)X) & 15) == 0 && (((size_t)P) & 15) == 0 && c_in % 4 == 0 &&
                   c_sc % 4 == 0 && c_in >= 0 && c_in + 32 <= DH_IN && c_sc >= 0 && c_sc + 64 <= DH_IN,
               "dheads_shared: bad layout (ldx %d, c_in %d, c_sc %d)", ldx, c_in, c_sc);
  DSharedArgs a = {X, ldx, b, c_in, c_sc, W1a, b1a, W1b, b1b, P};
  MG_LAUNCH(dheads_shared_kernel, dim3((b + 15) / 16), dim3(256), 0, stream, a);
  MG_LAUNCH_CHECK("dheads_shared");
  return MGGAN_OK;
}

/* rows [row0, rows) of a K-sample pass (row = k*b + ped): Ya / Yb of both heads from P[row % b] and the pred_enc block
 * (32 columns at c_pe) of X; mask (ceil((rows-row0)/16) * 64 words, or NULL): LeakyReLU sign bits for the backward pass */
int mggan_dheads_lean_fwd(const float* X, int ldx, int c_pe, int row0, int rows, int b, int g, int act_a, const float* P,
                          const float* W1a, const float* W2a, const float* b2a, const float* W1b, const float* W2b,
                          const float* b2b, unsigned long long* mask, float* Ya, float* Yb, hipStream_t stream) {
  MG_CHECK_ARG(g >= 1 && g <= DH_MAXG - 1 && row0 >= 0 && b > 0, "dheads_lean_fwd: g = %d not in 1..%d", g, DH_MAXG - 1);
  if (rows <= row0) return MGGAN_OK;
  MG_CHECK_ARG(X && P && W1a && W2a && b2a && W1b && W2b && b2b && Ya && Yb, "dheads_lean_fwd: null pointer");
  MG_CHECK_ARG(ldx >= DH_IN && (ldx & 3) == 0 && (((size_t)X) & 15) == 0 && (((size_t)P) & 15) == 0 && c_pe % 4 == 0 &&
                   c_pe >= 0 && c_pe + 32 <= DH_IN,
               "dheads_lean_fwd: bad layout (ldx %d, c_pe %d)", ldx, c_pe);
  MG_CHECK_ARG(act_a == ACT_NONE || act_a == ACT_SIGMOID_EPS || act_a == ACT_SIGMOID, "dheads_lean_fwd: output activation %d", act_a);
  DLeanArgs a = {};
  a.X = X; a.ldx = ldx; a.c_pe = c_pe; a.row0 = row0; a.rows = rows; a.b = b; a.g = g; a.act_a = act_a; a.P = P;
  a.W1a = W1a; a.W1b = W1b; a.W2a = W2a; a.b2a = b2a; a.W2b = W2b; a.b2b = b2b; a.mask = mask; a.Ya = Ya; a.Yb = Yb;
  const dim3 grid(dlean_grid(rows - row0));
  if (g <= 4) MG_LAUNCH(dheads_lean_fwd_kernel<4>, grid, dim3(256), 0, stream, a);
  else if (g <= 8) MG_LAUNCH(dheads_lean_fwd_kernel<8>, grid, dim3(256), 0, stream, a);
  else MG_LAUNCH(dheads_lean_fwd_kernel<16>, grid, dim3(256), 0, stream, a);
  MG_LAUNCH_CHECK("dheads_lean_fwd");
  return MGGAN_OK;
}

/* dX[row][c_pe .. c_pe+31] for rows [row0, rows): the pred_enc gradient of both heads (frozen discriminator) */
int mggan_dheads_lean_bwd(const float* dYa, const float* dYb, const float* Ya, const unsigned long long* mask, int row0,
                          int rows, int g, int act_a, const float* W1a, const float* W2a, const float* W1b, const float* W2b,
                          int c_pe, float* dX, int ld_dx, hipStream_t stream) {
  MG_CHECK_ARG(g >= 1 && g <= DH_MAXG - 1 && row0 >= 0, "dheads_lean_bwd: g = %d not in 1..%d", g, DH_MAXG - 1);
  if (rows <= row0) return MGGAN_OK;
  MG_CHECK_ARG(dYa && dYb && Ya && mask && W1a && W2a && W1b && W2b && dX, "dheads_lean_bwd: null pointer");
  MG_CHECK_ARG(ld_dx >= DH_IN && (ld_dx & 3) == 0 && (((size_t)dX) & 15) == 0 && c_pe % 4 == 0 && c_pe >= 0 && c_pe + 32 <= DH_IN,
               "dheads_lean_bwd: bad layout (ld_dx %d, c_pe %d)", ld_dx, c_pe);
  DLeanArgs a = {};
  a.c_pe = c_pe; a.row0 = row0; a.rows = rows; a.g = g; a.act_a = act_a; a.b = 1;
  a.W1a = W1a; a.W1b = W1b; a.W2a = W2a; a.W2b = W2b; a.mask = const_cast<unsigned long long*>(mask);
  a.Ya = const_cast<float*>(Ya); a.dYa = dYa; a.dYb = dYb; a.dX = dX; a.ld_dx = ld_dx;
  const dim3 grid(dlean_grid(rows - row0));
  if (g <= 4) MG_LAUNCH(dheads_lean_bwd_kernel<4>, grid, dim3(256), 0, stream, a);
  else if (g <= 8) MG_LAUNCH(dheads_lean_bwd_kernel<8>, grid, dim3(256), 0, stream, a);
  else MG_LAUNCH(dheads_lean_bwd_kernel<16>, grid, dim3(256), 0, stream, a);
  MG_LAUNCH_CHECK("dheads_lean_bwd");
  return MGGAN_OK;
}


static int d_rows_lean_check(const char* what, int T, int row0, int rows, int b, int g, int act_a) {
  MG_CHECK_ARG(T == DR_T, "%s: pred_len %d not built (12)", what, T);
  MG_CHECK_ARG(g >= 1 && g <= DH_MAXG - 1 && row0 >= 0 && b > 0, "%s: g = %d not in 1..%d", what, g, DH_MAXG - 1);
  MG_CHECK_ARG(act_a == ACT_NONE || act_a == ACT_SIGMOID_EPS || act_a == ACT_SIGMOID, "%s: output activation %d", what, act_a);
  (void)rows;
  return MGGAN_OK;
}

/* rows [row0, rows) of a K-sample pass, from the predicted steps pred (T=12, rows, 2) to the outputs of both heads in one
 * launch: pred_encoder (Wp1 (64,24), bp1, Wp2 (32,64), bp2) -> P[row % b] + W1cat[:, c_pe .. c_pe+31] pred_enc -> heads.
 * mask as mggan_dheads_lean_fwd (bits 48..63: the pred_encoder's hidden units) */
int mggan_d_rows_lean_fwd(const float* pred, int T, int row0, int rows, int b, int g, int act_a, const float* Wp1,
                          const float* bp1, const float* Wp2, const float* bp2, const float* P, int c_pe, const float* W1a,
                          const float* W2a, const float* b2a, const float* W1b, const float* W2b, const float* b2b,
                          unsigned long long* mask, float* Ya, float* Yb, hipStream_t stream) {
  if (int rc = d_rows_lean_check("d_rows_lean_fwd", T, row0, rows, b, g, act_a)) return rc;
  if (rows <= row0) return MGGAN_OK;
  MG_CHECK_ARG(pred && Wp1 && bp1 && Wp2 && bp2 && P && W1a && W2a && b2a && W1b && W2b && b2b && Ya && Yb,
               "d_rows_lean_fwd: null pointer");
  MG_CHECK_ARG((((size_t)P) & 15) == 0 && (((size_t)pred) & 7) == 0 && c_pe % 4 == 0 && c_pe >= 0 && c_pe + 32 <= DH_IN,
               "d_rows_lean_fwd: bad layout (c_pe %d)", c_pe);
  DRowsLeanArgs q = {};
  DLeanArgs& a = q.h;
  a.c_pe = c_pe; a.row0 = row0; a.rows = rows; a.b = b; a.g = g; a.act_a = act_a; a.P = P;
  a.W1a = W1a; a.W1b = W1b; a.W2a = W2a; a.b2a = b2a; a.W2b = W2b; a.b2b = b2b; a.mask = mask; a.Ya = Ya; a.Yb = Yb;
  q.pred = pred; q.Wp1 = Wp1; q.bp1 = bp1; q.Wp2 = Wp2; q.bp2 = bp2;
  const dim3 grid(dlean_grid(rows - row0));
  if (g <= 4) MG_LAUNCH(d_rows_lean_fwd_kernel<4>, grid, dim3(256), 0, stream, q);
  else if (g <= 8) MG_LAUNCH(d_rows_lean_fwd_kernel<8>, grid, dim3(256), 0, stream, q);
  else MG_LAUNCH(d_rows_lean_fwd_kernel<16>, grid, dim3(256), 0, stream, q);
  MG_LAUNCH_CHECK("d_rows_lean_fwd");
  return MGGAN_OK;
}

/* dpred (12, rows, 2), rows [row0, rows): the gradient of the predicted steps from dYa / dYb (frozen discriminator) */
int mggan_d_rows_lean_bwd(const float* dYa, const float* dYb, const float* Ya, const unsigned long long* mask, int T,
                          int row0, int rows, int g, int act_a, const float* Wp1, const float* Wp2, int c_pe,
                          const float* W1a, const float* W2a, const float* W1b, const float* W2b, float* dpred,
                          hipStream_t stream) {
  if (int rc = d_rows_lean_check("d_rows_lean_bwd", T, row0, rows, 1, g, act_a)) return rc;
  if (rows <= row0) return MGGAN_OK;
  MG_CHECK_ARG(dYa && dYb && Ya && mask && Wp1 && Wp2 && W1a && W2a && W1b && W2b && dpred, "d_rows_lean_bwd: null pointer");
  MG_CHECK_ARG((((size_t)dpred) & 7) == 0 && c_pe % 4 == 0 && c_pe >= 0 && c_pe + 32 <= DH_IN, "d_rows_lean_bwd: bad layout (c_pe %d)", c_pe);
  DRowsLeanArgs q = {};
  DLeanArgs& a = q.h;
  a.c_pe = c_pe; a.row0 = row0; a.rows = rows; a.g = g; a.act_a = act_a; a.b = 1;
  a.W1a = W1a; a.W1b = W1b; a.W2a = W2a; a.W2b = W2b; a.mask = const_cast<unsigned long long*>(mask);
  a.Ya = const_cast<float*>(Ya); a.dYa = dYa; a.dYb = dYb;
  q.Wp1 = Wp1; q.Wp2 = Wp2; q.dpred = dpred;
  const dim3 grid(dlean_grid(rows - row0));
  if (g <= 4) MG_LAUNCH(d_rows_lean_bwd_kernel<4>, grid, dim3(256), 0, stream, q);
  else if (g <= 8) MG_LAUNCH(d_rows_lean_bwd_kernel<8>, grid, dim3(256), 0, stream, q);
  else MG_LAUNCH(d_rows_lean_bwd_kernel<16>, grid, dim3(256), 0, stream, q);
  MG_LAUNCH_CHECK("d_rows_lean_bwd");
  return MGGAN_OK;
}


/* pred_encoder (Linear(24,64) - LeakyReLU(0.2) - Linear(64,32)) over `rows` rows taken from time-major steps (T = 12,
 * n_stride, 2): rows [0, rows_a) from a, the rest from b2 (NULL when rows_a == rows); pred_enc -> X[:, c_pe .. c_pe+31]
 * (row stride ldx); h1 (rows,64) and xrows (rows,24): the hidden layer and the row copy of the steps, when not NULL */
int mggan_pred_encoder_fwd(const float* a, const float* b2, int T, int n_stride, int rows_a, int rows, const float* Wp1,
                           const float* bp1, const float* Wp2, const float* bp2, float* X, int ldx, int c_pe, float* h1,
                           float* xrows, hipStream_t stream) {
  MG_CHECK_ARG(T == DR_T, "pred_encoder_fwd: pred_len %d not built (12)", T);
  if (rows <= 0) return MGGAN_OK;
  MG_CHECK_ARG(a && (b2 || rows_a == rows) && Wp1 && bp1 && Wp2 && bp2 && X, "pred_encoder_fwd: null pointer");
  MG_CHECK_ARG(rows_a >= 0 && rows_a <= rows && rows_a <= n_stride && rows - rows_a <= n_stride && (ldx & 3) == 0 &&
                   (c_pe & 3) == 0 && (((size_t)X) & 15) == 0 && (((size_t)a) & 7) == 0 && (!b2 || (((size_t)b2) & 7) == 0) &&
                   (!h1 || (((size_t)h1) & 15) == 0) && (!xrows || (((size_t)xrows) & 7) == 0),
               "pred_encoder_fwd: bad layout (rows %d of which %d from a, stride %d, ldx %d, c_pe %d)", rows, rows_a, n_stride,
               ldx, c_pe);
  PredEncArgs q = {a, b2, n_stride, rows_a, rows, Wp1, bp1, Wp2, bp2, X, ldx, c_pe, h1, xrows};
  const int nt = (rows + 15) / 16, wg = (nt + 3) / 4;
  MG_LAUNCH(pred_encoder_fwd_kernel, dim3(wg < 1024 ? wg : 1024), dim3(256), 0, stream, q);
  MG_LAUNCH_CHECK("pred_encoder_fwd");
  return MGGAN_OK;
}

}  // extern "C"
