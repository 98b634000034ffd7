// Fused dense chains for the small MLP stacks of MG-GAN (2-3 Linear layers with activations between
// them): discriminator heads Linear(192,96)-LeakyReLU-Linear(96,1|g), pred_encoder 24-64-32, in_encoder_fc
// 64-32-32 (reference model/modules/discriminators.py:46-56,76-108), the PM-network 128-16-16-g
// (model/modules/standard.py:99-105), all built by utils.make_mlp (utils.py:134-149).
//
// One launch runs the whole chain, weight-stationary: a persistent workgroup copies ALL weights of the chain
// into LDS once (<= ~90 KB, zero-padded fragment images), then walks its share of the 32- or 64-row tiles; the
// activations of a tile never leave the CU (one LDS buffer, rewritten in place between stages); every product
// runs on v_mfma_f32_16x16x4_f32 (exact f32, k ascending: bit-identical to the one-layer GEMM of gemm.hip).
// The backward pass is the same kernel: its stages use the weights transposed (dX = dZ W), the activation
// derivative enters as an elementwise factor taken from the saved forward output, and the gate gradients dZ_l
// every stage produces are stored for the weight-gradient GEMMs.
//
// Workgroup = 4 waves.  64-row tiles (large batches): wave w owns rows 16w.. and every 16-column tile of the
// stage; 32-row tiles (small batches, more workgroups): wave w owns rows 16*(w&1).. and half (w>>1) of the
// column tiles.  LDS strides: activation rows 194 floats (== 2 mod 32: the 16 rows x 2 k of a half-wave's A
// fragment hit 32 distinct banks); k-minor weight images have row stride == 2 mod 32, k-major ones == 16 mod 32.
#include "common.h"
#include "../../include/mggan_hip.h"

#define MC_LDX 194
#define MC_MAXD 192
#define MC_MAXT 12              // 16-column tiles of the widest stage
#define MC_MAX_WFLOATS 23040    // 90 KB of weight images

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
struct McPlan {  // LDS placement of the weight images (floats)
  int woff[3], wld[3], wfloats;
};

static inline int mc_ld_kminor(int K4) { return ((K4 - 2 + 31) / 32) * 32 + 2; }    // == 2 mod 32, >= K4
static inline int mc_ld_kmajor(int N16) { return ((N16 - 16 + 31) / 32) * 32 + 16; }  // == 16 mod 32, >= N16


// dst[r][c] (row stride dld, even) <- src[r][c] (row stride sld) for r < R_img, c < C_img (multiple of 4), zero
// outside r < R_valid, c < C_valid.  UNR loads per thread are put in flight before the first LDS store (a
// load -> store -> load chain would pay the memory latency once per element); out-of-range lanes read a clamped
// address and select zero, so the loop body is branch-free.  VEC: 16-byte loads (src 16-byte aligned, sld and
// C_valid multiples of 4).
template <int UNR, bool VEC>
__device__ __forceinline__ void mc_copy2d_t(float* dst, int dld, const float* __restrict__ src, size_t sld, int R_img,
                                            int C_img, int R_valid, int C_valid) {
  if (VEC) {
    const int c4n = C_img >> 2, total = R_img * c4n;
    for (int q0 = threadIdx.x; q0 < total; q0 += 256 * UNR) {
      float4 v[UNR];
      int at[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int q = q0 + 256 * u, qc = q < total ? q : 0;
        const int r = qc / c4n, c = (qc - r * c4n) << 2;
        const bool ok = q < total && r < R_valid && c < C_valid;
        const float4 t = *reinterpret_cast<const float4*>(ok ? src + (size_t)r * sld + c : src);
        v[u] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        at[u] = q < total ? r * dld + c : -1;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        if (at[u] >= 0) {
          *reinterpret_cast<float2*>(dst + at[u]) = make_float2(v[u].x, v[u].y);
          *reinterpret_cast<float2*>(dst + at[u] + 2) = make_float2(v[u].z, v[u].w);
        }
    }
  } else {
    const int total = R_img * C_img;
    for (int q0 = threadIdx.x; q0 < total; q0 += 256 * UNR) {
      float v[UNR];
      int at[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int q = q0 + 256 * u, qc = q < total ? q : 0;
        const int r = qc / C_img, c = qc - r * C_img;
        const bool ok = q < total && r < R_valid && c < C_valid;
        const float t = *(ok ? src + (size_t)r * sld + c : src);
        v[u] = ok ? t : 0.f;
        at[u] = q < total ? r * dld + c : -1;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u)
        if (at[u] >= 0) dst[at[u]] = v[u];
    }
  }
}
template <int UNR>
__device__ __forceinline__ void mc_copy2d(float* dst, int dld, const float* __restrict__ src, size_t sld, int R_img,
                                          int C_img, int R_valid, int C_valid) {
  const bool vec = (((size_t)src & 15) == 0) && (sld & 3) == 0 && (C_valid & 3) == 0;
  if (vec) mc_copy2d_t<UNR, true>(dst, dld, src, sld, R_img, C_img, R_valid, C_valid);
  else mc_copy2d_t<UNR, false>(dst, dld, src, sld, R_img, C_img, R_valid, C_valid);
}

// The vectorised input tile in two halves: mc_tile_load puts XU 16-byte loads per thread in flight (the next
// tile's, while the current tile is being multiplied), mc_tile_store lands them in LDS.  One pass covers the
// tile: R_img * C_img / 4 <= 256 * XU.
template <int XU>
struct McTile {
  float4 v[XU];
};
template <int XU>
__device__ __forceinline__ void mc_tile_load(McTile<XU>& t, const float* __restrict__ src, size_t sld, int R_img,
                                             int C_img, int R_valid, int C_valid) {
  const int c4n = C_img >> 2, total = R_img * c4n;
#pragma unroll
  for (int u = 0; u < XU; ++u) {
    const int q = threadIdx.x + 256 * u, qc = q < total ? q : 0;
    const int r = qc / c4n, c = (qc - r * c4n) << 2;
    const bool ok = q < total && r < R_valid && c < C_valid;
    const float4 x = *reinterpret_cast<const float4*>(ok ? src + (size_t)r * sld + c : src);
    t.v[u] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int XU>
__device__ __forceinline__ void mc_tile_store(const McTile<XU>& t, float* dst, int dld, int R_img, int C_img) {
  const int c4n = C_img >> 2, total = R_img * c4n;
#pragma unroll
  for (int u = 0; u < XU; ++u) {
    const int q = threadIdx.x + 256 * u;
    if (q < total) {
      const int r = q / c4n, c = (q - r * c4n) << 2;
      float* d = dst + r * dld + c;
      *reinterpret_cast<float2*>(d) = make_float2(t.v[u].x, t.v[u].y);
      *reinterpret_cast<float2*>(d + 2) = make_float2(t.v[u].z, t.v[u].w);
    }
  }
}

// acc[j] += A(16 x K4) . B_j(K4 x 16), j < NT, on one wave.  One wave per SIMD is resident (the weight images
// fill the LDS), so nothing but this loop hides the LDS latency: the fragments of k-step kk+4 are requested
// before the products of k-step kk are issued.
template <int NT>
__device__ __forceinline__ void mc_products(const float* arow, const float* brow, int bstep, int bj, int K4,
                                            f32x4* acc) {
  float av = arow[0], bv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bv[j] = brow[j * bj];
  for (int kk = 4; kk < K4; kk += 4) {
    const float* bp = brow + (kk >> 2) * bstep;
    const float an = arow[kk];
    float bn[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bn[j] = bp[j * bj];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc[j], 0, 0, 0);
    av = an;
#pragma unroll
    for (int j = 0; j < NT; ++j) bv[j] = bn[j];
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc[j], 0, 0, 0);
}

template <int MT>  // 16-row tiles per workgroup pass: 2 (32 rows) or 4 (64 rows)
__global__ __launch_bounds__(256) void mlp_chain_kernel(McArgs a, McPlan pl) {
  constexpr int ROWS = 16 * MT;
  constexpr int NTW = MT == 4 ? MC_MAXT : MC_MAXT / 2;  // column tiles one wave may own
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wl = smem;
  float* buf = smem + pl.wfloats;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fi = lane & 15, fk = lane >> 4;
  const int mt = MT == 4 ? w : (w & 1), nh = MT == 4 ? 0 : (w >> 1);

  constexpr int XU = MT == 4 ? 12 : 6;
  float* biasL = buf + ROWS * MC_LDX;  // [3][MC_MAXD]
  const int ntiles = (a.rows + ROWS - 1) / ROWS;
  const int K0 = a.K0, K04 = (K0 + 3) & ~3;
  const bool xvec = (((size_t)a.X & 15) == 0) && (a.ldx & 3) == 0 && (K0 & 3) == 0;
  McTile<XU> xt;
  if (xvec && (int)blockIdx.x < ntiles) {  // first input tile: in flight together with the weight images
    const int r0 = blockIdx.x * ROWS, rv = a.rows - r0 < ROWS ? a.rows - r0 : ROWS;
    mc_tile_load<XU>(xt, a.X + (size_t)r0 * a.ldx, a.ldx, ROWS, K04, rv, K0);
  }

  // ---- weight images and biases, once per workgroup ----
#pragma unroll 1
  for (int si = 0; si < a.n; ++si) {
    const McStage& S = a.s[si];
    const int K = S.K, N = S.N, K4 = (K + 3) & ~3, N16 = ((N + 15) >> 4) << 4, ld = pl.wld[si];
    float* T = Wl + pl.woff[si];
    if (!S.trans) mc_copy2d<10>(T, ld, S.W, S.ldw, N16, K4, N, K);   // T[n][k]
    else mc_copy2d<10>(T, ld, S.W, S.ldw, K4, N16, K, N);            // T[k][n]
    if (tid < MC_MAXD) biasL[si * MC_MAXD + tid] = (S.bias && tid < N) ? S.bias[tid] : 0.f;
  }

#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int r0 = tile * ROWS;
    __syncthreads();  // the previous tile's last stage is done with `buf` (and the weight images are complete)
    {  // input tile (zero-padded to a multiple of 4 columns and to ROWS rows)
      const int rv = a.rows - r0 < ROWS ? a.rows - r0 : ROWS;
      if (xvec) {
        mc_tile_store<XU>(xt, buf, MC_LDX, ROWS, K04);
        const int nx = tile + gridDim.x;
        if (nx < ntiles) {  // the next tile's loads fly during this tile's products
          const int r1 = nx * ROWS, rv1 = a.rows - r1 < ROWS ? a.rows - r1 : ROWS;
          mc_tile_load<XU>(xt, a.X + (size_t)r1 * a.ldx, a.ldx, ROWS, K04, rv1, K0);
        }
      } else {
        mc_copy2d_t<XU, false>(buf, MC_LDX, a.X + (size_t)r0 * a.ldx, a.ldx, ROWS, K04, rv, K0);
      }
      if (a.in_mul) {  // backward through the last activation: x *= act'(y); kept for the weight gradient
        __syncthreads();
        for (int e = tid; e < rv * K0; e += 256) {
          const int r = e / K0, k = e - r * K0;
          const float v = buf[r * MC_LDX + k] *
                          mg_act_grad_from_out(a.in_mul[(size_t)(r0 + r) * a.ld_in_mul + k], a.in_mul_act, a.in_mul_slope);
          buf[r * MC_LDX + k] = v;
          if (a.in_store) a.in_store[(size_t)(r0 + r) * a.ld_in_store + k] = v;
        }
      }
    }

#pragma unroll 1
    for (int si = 0; si < a.n; ++si) {
      const McStage& S = a.s[si];
      const int N = S.N, K4 = (S.K + 3) & ~3, ld = pl.wld[si];
      const float* T = Wl + pl.woff[si];
      const int ntt = (N + 15) >> 4;
      const int nth = MT == 4 ? ntt : ((ntt + 1) >> 1), j0 = nh * nth;
      int nt = ntt - j0;
      nt = nt > nth ? nth : (nt < 0 ? 0 : nt);
      const bool trans = S.trans != 0;
      f32x4 acc[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      // saved activations for the derivative factor: fetched now, consumed after the products
      float mulv[NTW][4];
      if (S.mul_src) {
#pragma unroll
        for (int j = 0; j < NTW; ++j)
          if (j < nt) {
            const int col = (j0 + j) * 16 + fi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int gr = r0 + mt * 16 + fk * 4 + r;
              const bool ok = col < N && gr < a.rows;
              const float y = *(ok ? S.mul_src + (size_t)gr * S.ld_mul + col : S.mul_src);
              mulv[j][r] = ok ? mg_act_grad_from_out(y, S.mul_act, S.mul_slope) : 0.f;
            }
          }
      }
      __syncthreads();  // `buf` holds this stage's input
      const float* arow = buf + (mt * 16 + fi) * MC_LDX + fk;
      const float* brow = trans ? T + fk * ld + j0 * 16 + fi : T + (j0 * 16 + fi) * ld + fk;
      const int bstep = trans ? 4 * ld : 4, bj = trans ? 16 : 16 * ld;
      switch (nt) {  // compile-time tile counts: branch-free product loops
        case 1: mc_products<1>(arow, brow, bstep, bj, K4, acc); break;
        case 2: mc_products<2>(arow, brow, bstep, bj, K4, acc); break;
        case 3: mc_products<3>(arow, brow, bstep, bj, K4, acc); break;
        case 4: mc_products<4>(arow, brow, bstep, bj, K4, acc); break;
        case 5: mc_products<5>(arow, brow, bstep, bj, K4, acc); break;
        case 6: mc_products<6>(arow, brow, bstep, bj, K4, acc); break;
        default:
          if (NTW > 6) {
            switch (nt) {
              case 7: mc_products<7 <= NTW ? 7 : 1>(arow, brow, bstep, bj, K4, acc); break;
              case 8: mc_products<8 <= NTW ? 8 : 1>(arow, brow, bstep, bj, K4, acc); break;
              case 9: mc_products<9 <= NTW ? 9 : 1>(arow, brow, bstep, bj, K4, acc); break;
              case 10: mc_products<10 <= NTW ? 10 : 1>(arow, brow, bstep, bj, K4, acc); break;
              case 11: mc_products<11 <= NTW ? 11 : 1>(arow, brow, bstep, bj, K4, acc); break;
              case 12: mc_products<12 <= NTW ? 12 : 1>(arow, brow, bstep, bj, K4, acc); break;
              default: break;
            }
          }
          break;
      }
      __syncthreads();  // every wave has consumed `buf`: the outputs may overwrite it

      // D fragment: lane l, register r <-> row (l>>4)*4 + r, column l&15 of the 16x16 tile
      const bool last = si == a.n - 1;
#pragma unroll
      for (int j = 0; j < NTW; ++j)
        if (j < nt) {
          const int col = (j0 + j) * 16 + fi;
          const float bv = biasL[si * MC_MAXD + (col < MC_MAXD ? col : 0)];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = mt * 16 + fk * 4 + r, gr = r0 + row;
            float v = 0.f;
            if (col < N) {
              v = mg_act(acc[j][r] + bv, S.act, S.slope);
              if (gr < a.rows) {
                if (S.mul_src) v *= mulv[j][r];
                if (S.out) {
                  float* o = S.out + (size_t)gr * S.ld_out + col;
                  *o = S.accumulate ? (*o + v) : v;
                }
              }
            }
            if (!last) buf[row * MC_LDX + col] = v;  // columns N..N16 become the zero padding of the next K
          }
        }
    }
  }
}

extern "C" {

int mggan_mlp_chain(const void* args, hipStream_t stream) {
  MG_CHECK_ARG(args, "mlp_chain: null arguments");
  McArgs a = *(const McArgs*)args;
  MG_CHECK_ARG(a.X && a.n >= 1 && a.n <= 3 && a.rows >= 0, "mlp_chain: bad arguments (n = %d)", a.n);
  MG_CHECK_ARG(a.K0 >= 1 && a.K0 <= MC_MAXD, "mlp_chain: input width %d not in 1..%d", a.K0, MC_MAXD);
  McPlan pl = {};
  int k = a.K0;
  for (int i = 0; i < a.n; ++i) {
    MG_CHECK_ARG(a.s[i].W && a.s[i].K == k && a.s[i].N >= 1 && a.s[i].N <= MC_MAXD,
                 "mlp_chain: stage %d has K = %d, N = %d (expected K = %d, N <= %d)", i, a.s[i].K, a.s[i].N, k, MC_MAXD);
    const int K4 = (k + 3) & ~3, N16 = ((a.s[i].N + 15) >> 4) << 4;
    pl.woff[i] = pl.wfloats;
    if (a.s[i].trans) { pl.wld[i] = mc_ld_kmajor(N16); pl.wfloats += K4 * pl.wld[i]; }
    else { pl.wld[i] = mc_ld_kminor(K4); pl.wfloats += N16 * pl.wld[i]; }
    pl.wfloats = (pl.wfloats + 3) & ~3;
    k = a.s[i].N;
  }
  MG_CHECK_ARG(pl.wfloats <= MC_MAX_WFLOATS, "mlp_chain: %d floats of weights do not fit the LDS image (%d)", pl.wfloats,
               MC_MAX_WFLOATS);
  MG_CHECK_ARG(a.s[a.n - 1].out, "mlp_chain: the last stage needs an output pointer");
  if (a.rows == 0) return MGGAN_OK;
  const bool big = a.rows >= 8192;
  const int rows_per = big ? 64 : 32;
  const int lds = (pl.wfloats + rows_per * MC_LDX + 3 * MC_MAXD) * 4;
  static bool attr_set = false;
  if (!attr_set) {
    const int cap = (MC_MAX_WFLOATS + 64 * MC_LDX + 3 * MC_MAXD) * 4;
    hipError_t e = hipFuncSetAttribute((const void*)mlp_chain_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)mlp_chain_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e != hipSuccess) {
      mggan_set_error("mlp_chain: cannot reserve %d bytes of LDS: %s", cap, hipGetErrorString(e));
      return MGGAN_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int ntiles = cdiv(a.rows, rows_per);
  int per_cu = (160 * 1024) / lds;
  per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
  const int grid = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
  if (big) hipLaunchKernelGGL(mlp_chain_kernel<4>, dim3(grid), dim3(256), lds, stream, a, pl);
  else hipLaunchKernelGGL(mlp_chain_kernel<2>, dim3(grid), dim3(256), lds, stream, a, pl);
  MG_LAUNCH_CHECK("mlp_chain");
  return MGGAN_OK;
}

}  // extern "C"
