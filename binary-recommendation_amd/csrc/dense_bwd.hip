// B1 of one tower layer in ONE launch: dz, dx = dz·W^T and dW = T(x)^T·dz (+ db) on fp32 MFMA, dz never leaves the CU.
//
// (Round 1 ran dx and dW as two kernels with dz written to HBM 112 floats wide and read back: 196 MB of traffic for
//  120 MB of algorithmic bytes, x read twice, and 0.24 of the fp32-MFMA peak.)
//
// One 512-thread workgroup per CU walks 64-row tiles.  Its 8 waves have two ROLES, one wave of each role per SIMD
// (waves w and w+4 share a SIMD), so a SIMD's matrix pipe always has two different instruction streams to draw from:
//   waves 0-3, "dx":  wave r owns rows 16r..16r+15 of the tile.  It loads gy / y straight from global memory in the MFMA
//       A layout (lane (c16,g) <- 16 B of row c16 at n = 16j+4g, one tile ahead), forms dz = act'(y)·BN-backward(gy) in
//       registers, publishes it to the dW waves as the LDS image Zs[buf][n-tile][row][16], and contracts it against the
//       row-major W image in LDS: gx = keep/(1-p) · dz·W^T, transposed through a per-wave LDS patch into 16-B row stores;
//       BatchNorm-backward sums of the producer (sum gx, sum gx·xhat) in the C layout.
//   waves 4-7, "dW":  wave q owns k-tiles 2q, 2q+1 (32 input features).  It loads those columns of x for all 64 rows
//       (full 128-B lines, one tile ahead), applies T() = BatchNorm-affine + keep bits, keeps the result in its own LDS
//       image Xs[k-tile][row][16] and accumulates dW[32 x N] += T(x)^T·dz over ALL tiles of the workgroup in registers
//       (A = Xs, B = Zs, both conflict-free ds_read_b32: the contraction runs over rows); db = column sums of Zs.
//   One barrier per tile (Zs is double buffered).  Each wave issues 224 MFMAs per 64-row tile at the 128 x 100 layer.
// Outputs: gx rows, one [dW | db] slab per workgroup (fixed-order reduction by brDenseFinalize / brReduceSlabs: no
// float atomics), in_sums as double atomics into BR_STAT_REPLICAS replicas.
// Shapes: K, N <= 128 with 16-B aligned rows: every row stride a multiple of 4 floats (so rows of K or N floats are padded
// to 4 and a 16-B access that starts inside a row stays inside its allocation); everything else runs the two-kernel path of mlp.hip.
#include "dense.h"
#include "stamps.h"

namespace br {

using f32x4 = __attribute__((ext_vector_type(4))) float;

#ifdef BR_ABLATE_MFMA      // timing experiment, see dense_fwd.hip
#define BR_ABLATE_KEEP(c) (c)
#else
#define BR_ABLATE_KEEP(c) true
#endif
constexpr int kBwdThreads = 512;
constexpr int kBwdRows = 64;             // rows per tile
constexpr int kBwdRep = BR_STAT_REPLICAS;
constexpr int kBwdPatchLd = 20;          // per-wave 16x16 transposition patch (as dense_fwd.hip)

__device__ __forceinline__ f32x4 mfma16b(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float keep_bit(float x, uint32_t w, uint32_t pos) {      // x & (bit pos of w ? ~0 : 0)
  return __int_as_float(__float_as_int(x) & __builtin_amdgcn_sbfe((int)w, pos, 1u));
}

// Zs / Xs rows are 16 floats (64 B): the 16-B writes of 16 lanes that hold 16 consecutive rows would land on 4 bank groups (4-way
// conflicts: rocprofv3 counted 37 % of the LDS cycles of the 128 x 100 layer as conflict cycles).  The 4-float column group q of row r is
// stored at group q ^ ((r >> 2) & 3): writes become conflict-free, the b32 reads of the MFMA loops (16 columns x 4 rows r = 4s + g of one
// k-step: the same XOR for every lane) stay conflict-free.
__device__ __forceinline__ int bwd_swz(int col16, int row_quad) { return (((col16 >> 2) ^ (row_quad & 3)) << 2) | (col16 & 3); }

template <int NT, int KT, bool IBN>
__global__ __launch_bounds__(kBwdThreads, 2) void dense_bwd_kernel(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int Np = NT * 16, Kp = KT * 16, ldw = Np + 4;
  constexpr int KW = (KT + 1) / 2;                      // keep words per row
  float* Ws = smem;                                     // [Kp][ldw]   W row-major, zero padded
  float* Cs = Ws + Kp * ldw;                            // [4][Np]     out-BN constants
  float* Is = Cs + 4 * Np;                              // [mean Kp | rstd Kp] of the in BN
  float* ssb = Is + 2 * Kp;                             // [scale Kp | shift Kp]
  float* Zs = ssb + 2 * Kp;                             // [2][NT][64][16]  dz
  float* Xs = Zs + 2 * NT * kBwdRows * 16;              // [4 dW waves][2][64][16]  T(x)
  float* patches = Xs + 4 * 2 * kBwdRows * 16;          // [4 dx waves][16][kBwdPatchLd]
  double* red = reinterpret_cast<double*>(patches + 4 * 16 * kBwdPatchLd);    // [2][Kp]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const int K = a.K, N = a.N;
  const int64_t batch = a.batch;
  const int64_t n_tiles = (batch + kBwdRows - 1) / kBwdRows;
  const bool is_dx = wave < 4;                          // wave-uniform role
  BR_STAMP_DECL;
  BR_STAMP_RT(10);
  BR_STAMP(0);
  const int sigmask = a.act == BR_ACT_SIGMOID ? -1 : 0, relumask = a.act == BR_ACT_RELU ? -1 : 0;
  // slabs beyond this grid (the count is sized for the two-kernel path; small batches launch fewer workgroups): zeroed here instead
  // of by a memset launch of their own in front of the kernel (~9 us each at launch-bound sizes)
  for (int64_t z = (int64_t)blockIdx.x * kBwdThreads + threadIdx.x; z < a.zero_n; z += (int64_t)gridDim.x * kBwdThreads) a.zero[z] = 0.f;

  // ---------------- staging: W image, BN constants ----------------
  // (a lambda that both roles call AFTER they have issued their first tile's loads: in program order ahead of them, the staging's
  //  three dependent memory round trips - constants, BatchNorm sums, W - ran before the first tile was even requested)
  auto stage = [&]() {
    for (int k = threadIdx.x; k < Kp; k += kBwdThreads) {
      Is[k] = (IBN && k < K) ? a.i_mean[k] : 0.f;
      Is[Kp + k] = (IBN && k < K) ? a.i_rstd[k] : 0.f;
      ssb[k] = k < K ? (a.scale ? a.scale[k] : 1.f) : 0.f;
      ssb[Kp + k] = (a.scale && k < K) ? a.shift[k] : 0.f;
      red[k] = 0.0; red[Kp + k] = 0.0;
    }
    for (int n = threadIdx.x; n < Np; n += kBwdThreads) {
      float c1 = 1.f, c2 = 0.f, c3 = 0.f, mu = 0.f;
      if (a.o_mean && n < N) {
        double s1 = 0.0, s2 = 0.0;
        for (int r = 0; r < kBwdRep; ++r) { s1 += a.o_sums[(size_t)r * 2 * N + n]; s2 += a.o_sums[(size_t)r * 2 * N + N + n]; }
        const float rs = a.o_rstd[n];
        mu = a.o_mean[n];
        c1 = a.o_gamma[n] * rs;
        c2 = (float)(s1 * (double)a.inv_batch);
        c3 = (float)(s2 * (double)a.inv_batch) * rs;
      }
      // da = c1 * (gy - c2 - (y - mu) * c3)
      Cs[n] = c1; Cs[Np + n] = c2; Cs[2 * Np + n] = c3; Cs[3 * Np + n] = mu;
    }
    {   // W: 4 floats along n per thread (one 16-B load when N % 4 == 0).  Clamped addresses, no lane conditions around the loads and
        // the zero padding applied as a bit mask afterwards: with `if (in range) load` per element hipcc built an exec-masked block
        // with its own s_waitcnt vmcnt(0) around each of the 7 loads - seven memory round trips in a row before the first tile
      constexpr int TOT = Kp * (Np / 4), TR = (TOT + kBwdThreads - 1) / kBwdThreads;
      float4 wv[TR];
      const bool n4 = (N & 3) == 0;
      if (n4) {
  #pragma unroll
        for (int i = 0; i < TR; ++i) {
          const int idx = threadIdx.x + i * kBwdThreads;
          const int idc = idx < TOT ? idx : 0;
          const int k = idc / (Np / 4), n = (idc - k * (Np / 4)) * 4;
          wv[i] = *reinterpret_cast<const float4*>(a.W + (int64_t)(k < K ? k : 0) * N + (n < N ? n : 0));
        }
      } else {
  #pragma unroll
        for (int i = 0; i < TR; ++i) {
          const int idx = threadIdx.x + i * kBwdThreads;
          const int idc = idx < TOT ? idx : 0;
          const int k = idc / (Np / 4), n = (idc - k * (Np / 4)) * 4;
          const float* wr = a.W + (int64_t)(k < K ? k : 0) * N;
          wv[i] = make_float4(wr[n < N ? n : 0], wr[n + 1 < N ? n + 1 : 0], wr[n + 2 < N ? n + 2 : 0], wr[n + 3 < N ? n + 3 : 0]);
        }
      }
  #pragma unroll
      for (int i = 0; i < TR; ++i) {
        const int idx = threadIdx.x + i * kBwdThreads;
        const int k = idx / (Np / 4), n = (idx - k * (Np / 4)) * 4;
        const int kin = k < K ? -1 : 0;
        const int m0 = kin & (n < N ? -1 : 0), m1 = kin & (n + 1 < N ? -1 : 0), m2 = kin & (n + 2 < N ? -1 : 0), m3 = kin & (n + 3 < N ? -1 : 0);
        const float4 w = make_float4(__int_as_float(__float_as_int(wv[i].x) & m0), __int_as_float(__float_as_int(wv[i].y) & m1),
                                     __int_as_float(__float_as_int(wv[i].z) & m2), __int_as_float(__float_as_int(wv[i].w) & m3));
        if (idx < TOT) *reinterpret_cast<float4*>(Ws + k * ldw + n) = w;
      }
    }

  };

  if (is_dx) {
    // =========================================== dx waves ===========================================
    const int rt = wave;                                 // row tile of the 64-row tile
    float* patch = patches + wave * 16 * kBwdPatchLd;
    float4 vg[NT], vy[NT];
    // the A-layout loads touch 16 rows x 64 B per instruction: the CU's address pipe takes them at ~30 cycles each, and a wave that issues
    // a whole tile's 2 NT loads back to back sits ~6 000 cycles at the issue port before its first MFMA (in-kernel stamps).  The next
    // tile's loads are therefore issued two at a time between the MFMA blocks of the first pass (load_gy_j).
    const float *pg_n = a.gy, *py_n = a.y;               // row pointers of the tile being prefetched
    auto gy_rows = [&](int64_t tile) {
      int64_t row = tile * kBwdRows + rt * 16 + c16;
      row = row < batch ? row : batch - 1;
      pg_n = a.gy + row * a.ldgy + 4 * g;
      py_n = a.y + row * a.ldy + 4 * g;
    };
    auto load_gy_j = [&](int j) {                        // A layout: lane (c16,g) <- row c16, columns 16j+4g..+3
      const bool in = 16 * j + 16 <= N || 16 * j + 4 * g < N;          // the 16-B group starts inside N (rows are padded to 4 floats)
      vg[j] = *reinterpret_cast<const float4*>(in ? pg_n + 16 * j : pg_n - 4 * g);
      vy[j] = *reinterpret_cast<const float4*>(in ? py_n + 16 * j : py_n - 4 * g);
    };
    auto load_gy = [&](int64_t tile) {
      gy_rows(tile);
#pragma unroll
      for (int j = 0; j < NT; ++j) load_gy_j(j);
    };
    int64_t tile = blockIdx.x;
    if (tile < n_tiles) load_gy(tile);
    BR_STAMP(1);
    stage();
    BR_STAMP(2);
    float isum[KT], isq[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) { isum[kt] = 0.f; isq[kt] = 0.f; }
    __syncthreads();                                     // staging visible
    BR_STAMP(3);
    int it = 0;
    for (; tile < n_tiles; tile += gridDim.x, ++it) {
      const int64_t rbase = tile * kBwdRows + rt * 16;
      const bool live = rbase + c16 < batch;             // A-layout row of this lane
      float* Zb = Zs + (it & 1) * NT * kBwdRows * 16;
      // ---- dz = act'(y) * BN-backward(gy), published to the dW waves ----
      float4 dz[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = 16 * j + 4 * g;
        const float4 c1 = *reinterpret_cast<const float4*>(Cs + n), c2 = *reinterpret_cast<const float4*>(Cs + Np + n);
        const float4 c3 = *reinterpret_cast<const float4*>(Cs + 2 * Np + n), mu = *reinterpret_cast<const float4*>(Cs + 3 * Np + n);
        const float ge[4] = {vg[j].x, vg[j].y, vg[j].z, vg[j].w}, ye[4] = {vy[j].x, vy[j].y, vy[j].z, vy[j].w};
        const float k1[4] = {c1.x, c1.y, c1.z, c1.w}, k2[4] = {c2.x, c2.y, c2.z, c2.w}, k3[4] = {c3.x, c3.y, c3.z, c3.w}, km[4] = {mu.x, mu.y, mu.z, mu.w};
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int lm = (live && (16 * j + 16 <= N || n + e < N)) ? -1 : 0;      // 0 for rows past the batch / columns past N
          const float da = k1[e] * (ge[e] - k2[e] - (ye[e] - km[e]) * k3[e]);
          // act'(y) from the activation output: sigmoid y(1-y), relu [y > 0], linear 1 - blended with bit masks (no branches)
          const float ds = ye[e] * (1.f - ye[e]);
          const float dr = ye[e] > 0.f ? 1.f : 0.f;
          const float dl = __int_as_float((__float_as_int(ds) & sigmask) | (__float_as_int(dr) & relumask) | (0x3f800000 & ~(sigmask | relumask)));
          v[e] = __int_as_float(__float_as_int(da * dl) & lm);
        }
        dz[j] = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(Zb + (j * kBwdRows + rt * 16 + c16) * 16 + 4 * (g ^ ((c16 >> 2) & 3))) = dz[j];     // swizzled: see bwd_swz
        __builtin_amdgcn_sched_barrier(0);
      }
      BR_STAMP(4);      // dz of the first tile published (includes the wait for gy / y)
      const bool has_next = tile + gridDim.x < n_tiles;                  // wave-uniform
      if (has_next) gy_rows(tile + gridDim.x);
      if (has_next && !a.gx) load_gy(tile + gridDim.x);                  // (no dx product to hide them behind)
      // keep words of this lane's 4 C-layout rows (rows 4g..4g+3 of the row tile) and, with an input BN, the raw x of its outputs
      uint32_t kb[4][KW];
      if (a.keep && a.gx) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int64_t row = rbase + 4 * g + r;
          row = row < batch ? row : batch - 1;
#pragma unroll
          for (int w = 0; w < KW; ++w) kb[r][w] = a.keep[row * a.kw + w];
        }
      }
      BR_STAMP(9);      // next tile's gy / y and this tile's keep words requested
      float xraw[4][KT];                                 // IBN: raw x of this lane's outputs (C layout), requested before the MFMAs
      if (IBN && a.gx) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int64_t row = rbase + 4 * g + r;
          row = row < batch ? row : batch - 1;
          const float* xr_ = a.x + row * a.ldx;
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) { const int k = kt * 16 + c16; xraw[r][kt] = xr_[k < K ? k : 0]; }
        }
      }
      // (no barrier here: the dx product needs only this wave's dz registers and the W image; the tile's barrier sits at the END of the
      //  iteration, so that this wave's MFMA passes run while the dW waves - one per SIMD beside it - transform their next x tile, and
      //  its dz arithmetic for the next tile runs while they multiply: see the dW loop)
      if (a.gx) {
        const int64_t left = batch - rbase;
        const int rows16 = left > 16 ? 16 : (left < 0 ? 0 : (int)left);
        const bool row_ok = c16 < rows16;
        float* gxrow = a.gx + (rbase + (row_ok ? c16 : 0)) * a.ldgx + 4 * g;      // row layout: &gx[rbase + c16][4g]
        // ---- dx = dz · W^T in passes of 4 k-tiles: 4 independent accumulator chains ----
#pragma unroll
        for (int kt0 = 0; kt0 < KT; kt0 += 4) {
          constexpr int WMAX = 4;
          const int Wn = (KT - kt0) < WMAX ? (KT - kt0) : WMAX;
          f32x4 acc[WMAX];
#pragma unroll
          for (int w = 0; w < WMAX; ++w) acc[w] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            float4 b[WMAX];
#pragma unroll
            for (int w = 0; w < WMAX; ++w)
              if (w < Wn) b[w] = *reinterpret_cast<const float4*>(Ws + ((kt0 + w) * 16 + c16) * ldw + 16 * j + 4 * g);
#pragma unroll
            for (int w = 0; w < WMAX; ++w) if (w < Wn) acc[w] = mfma16b(dz[j].x, b[w].x, acc[w]);
#pragma unroll
            for (int w = 0; w < WMAX; ++w) if (w < Wn && BR_ABLATE_KEEP((j & 1) == 0)) acc[w] = mfma16b(dz[j].y, b[w].y, acc[w]);
#pragma unroll
            for (int w = 0; w < WMAX; ++w) if (w < Wn && BR_ABLATE_KEEP(false)) acc[w] = mfma16b(dz[j].z, b[w].z, acc[w]);
#pragma unroll
            for (int w = 0; w < WMAX; ++w) if (w < Wn && BR_ABLATE_KEEP(false)) acc[w] = mfma16b(dz[j].w, b[w].w, acc[w]);
            if (kt0 == 0 && has_next) load_gy_j(j);                    // next tile's gy / y: two loads behind every MFMA block of the first pass
            __builtin_amdgcn_sched_barrier(0);
          }
          BR_STAMP(5 + 2 * (kt0 / 4));      // pass MFMAs issued
          // epilogue of the pass: C layout (rows 4g..4g+3, column k = kt*16+c16): dropout transposed, producer-BN sums,
          // then through the patch into row layout and out as 16-B stores
#pragma unroll
          for (int w = 0; w < WMAX; ++w) {
            if (w < Wn) {
              const int kt = kt0 + w, k = kt * 16 + c16;
              float xm = 0.f, xr = 0.f;
              if (IBN) { xm = Is[k]; xr = Is[Kp + k]; }
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float dh = acc[w][r] * a.inv_keep;
                if (a.keep) dh = keep_bit(dh, kb[r][kt >> 1], 16u * (kt & 1) + c16);
                if (IBN) {                                              // (dh is 0 for rows past the batch: dz is)
                  isum[kt] += dh;
                  isq[kt] = fmaf(dh, (xraw[r][kt] - xm) * xr, isq[kt]);
                }
                patch[(4 * g + r) * kBwdPatchLd + c16] = dh;
              }
              __builtin_amdgcn_wave_barrier();
              const float4 o = *reinterpret_cast<const float4*>(patch + c16 * kBwdPatchLd + 4 * g);
              __builtin_amdgcn_wave_barrier();
              if (row_ok && (kt * 16 + 16 <= K || kt * 16 + 4 * g < K)) *reinterpret_cast<float4*>(gxrow + kt * 16) = o;
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          BR_STAMP(6 + 2 * (kt0 / 4));      // pass epilogue issued
        }
      }
      __syncthreads();                                   // S_it: Zs[it & 1] is published; the dW waves' readers of Zs[(it - 1) & 1] are done
    }
    if (IBN && a.in_sums) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        float sv = isum[kt], q = isq[kt];
        sv += __shfl_xor(sv, 16, 64); sv += __shfl_xor(sv, 32, 64);
        q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
        if (g == 0) { atomicAdd(&red[kt * 16 + c16], (double)sv); atomicAdd(&red[Kp + kt * 16 + c16], (double)q); }
      }
    }
  } else {
    // =========================================== dW waves ===========================================
    const int q = wave - 4;                              // k-tiles 2q, 2q+1
    const bool q_live = 2 * q < KT;                      // wave-uniform
    float* Xw = Xs + q * 2 * kBwdRows * 16;
    const int lr = lane >> 3, lc = lane & 7;             // x loads: row lr + 8i, columns 32q + 4lc..+3
    const int kcol = 32 * q + 4 * lc;
    const bool col_in = kcol < K;
    float4 xv[8];
    uint32_t kx[8];
    auto x_row = [&](int64_t tile, int i) {
      int64_t row = tile * kBwdRows + lr + 8 * i;
      return row < batch ? row : batch - 1;
    };
    auto load_x = [&](int64_t tile) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t row = x_row(tile, i);
        xv[i] = *reinterpret_cast<const float4*>(a.x + row * a.ldx + (col_in ? kcol : 0));
        kx[i] = a.keep ? a.keep[row * a.kw + (col_in ? q : 0)] : 0xFFFFFFFFu;     // word q covers columns 32q..32q+31
      }
    };
    // one of the next tile's 16 loads (8 x rows, 8 keep words): issued one per k-step of the product - back to back they hold the wave
    // at the issue port for thousands of cycles (the CU's address pipe takes ~30 cycles per 16-row access)
    int64_t tile = blockIdx.x;
    if (q_live && tile < n_tiles) load_x(tile);
    BR_STAMP(1);
    stage();
    BR_STAMP(2);
    f32x4 acc[2][NT];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[h][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float dbacc = 0.f;                                   // lane -> column 28q.. of db, half the rows each (see below)
    constexpr int DBC = (Np + 3) / 4;                    // db columns per dW wave
    const int dbc = lane & 31, dbh = lane >> 5;
    const int dbcol = q * DBC + dbc;
    const bool db_live = dbc < DBC && dbcol < Np;
    __syncthreads();                                     // staging visible
    BR_STAMP(3);
    const float4 sc = *reinterpret_cast<const float4*>(ssb + (col_in ? kcol : 0)), sh = *reinterpret_cast<const float4*>(ssb + Kp + (col_in ? kcol : 0));
    // Between two barriers the dx wave of this SIMD first computes dz of tile t (VALU) and then multiplies (MFMA); this wave does the
    // opposite - first dW += T(x(t-1))^T dz(t-1) (MFMA: both operands were published before the last barrier), then T(x(t)) (VALU) -
    // so the SIMD's matrix pipe is fed by one of the two at any time instead of by both and then by neither.
    auto multiply = [&](const float* Zb) {
      if (q_live) {
        // ---- dW[32 x N] += T(x)^T · dz: contraction over the 64 rows, k-step s <-> rows 4s+g ----
        // operands of k-step s+1 are read from LDS while the MFMAs of k-step s issue (left to itself hipcc waits for each pair of
        // ds_reads right before the four MFMAs that use them: 12 100 cycles per tile for 7 200 cycles of MFMA work; with a scheduling
        // fence per k-step and no explicit double buffer: 19 600)
        float a0, a1, bq[NT];
        auto fetch = [&](int s_, float& x0, float& x1, float (&bb)[NT]) {
          const int r = 4 * s_ + g;
          const int cs = bwd_swz(c16, s_);                 // (r >> 2) & 3 == s & 3
          x0 = Xw[r * 16 + cs]; x1 = Xw[(kBwdRows + r) * 16 + cs];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bb[nt] = Zb[(nt * kBwdRows + r) * 16 + cs];
        };
        fetch(0, a0, a1, bq);
#pragma unroll
        for (int s = 0; s < kBwdRows / 4; ++s) {
          float n0 = 0.f, n1 = 0.f, nb[NT];
          if (s + 1 < kBwdRows / 4) fetch(s + 1, n0, n1, nb);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if (!BR_ABLATE_KEEP((s & 7) == 0 || (s & 7) == 3 || (s & 7) == 6)) continue;
            acc[0][nt] = mfma16b(a0, bq[nt], acc[0][nt]);
            if (2 * q + 1 < KT) acc[1][nt] = mfma16b(a1, bq[nt], acc[1][nt]);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (s + 1 < kBwdRows / 4) {
            a0 = n0; a1 = n1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bq[nt] = nb[nt];
          }
        }
      }
      // ---- db: column sums of dz (lane -> one column, half of the rows) ----
      if (db_live) {
        float s0 = 0.f;
        const float* zc = Zb + ((dbcol >> 4) * kBwdRows + 32 * dbh) * 16;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) s0 += zc[r * 16 + bwd_swz(dbcol & 15, r >> 2)];
        dbacc += s0;
      }
    };
    int it = 0;
    for (; tile < n_tiles; tile += gridDim.x, ++it) {
      const bool has_next = tile + gridDim.x < n_tiles;                      // wave-uniform
      if (it == 1) BR_STAMP(6);
      if (it > 0) multiply(Zs + ((it - 1) & 1) * NT * kBwdRows * 16);        // tile it-1: its dz and its T(x) are complete
      if (it == 1) BR_STAMP(7);      // first product done
      if (q_live) {
        // ---- T(x) = BN affine + keep bits -> Xs[k-tile][row][16] (free again: the product above was its last reader) ----
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const uint32_t p0 = 4u * lc;
          float4 t;
          t.x = keep_bit(fmaf(xv[i].x, sc.x, sh.x), kx[i], p0); t.y = keep_bit(fmaf(xv[i].y, sc.y, sh.y), kx[i], p0 + 1);
          t.z = keep_bit(fmaf(xv[i].z, sc.z, sh.z), kx[i], p0 + 2); t.w = keep_bit(fmaf(xv[i].w, sc.w, sh.w), kx[i], p0 + 3);
          if (!col_in) t = make_float4(0.f, 0.f, 0.f, 0.f);
          if (K & 3) {                                     // wave-uniform: a 16-B group may straddle K (padded rows)
            if (kcol + 1 >= K) t.y = 0.f;
            if (kcol + 2 >= K) t.z = 0.f;
            if (kcol + 3 >= K) t.w = 0.f;
          }
          *reinterpret_cast<float4*>(Xw + ((lc >> 2) * kBwdRows + lr + 8 * i) * 16 + 4 * ((lc & 3) ^ (((lr + 8 * i) >> 2) & 3))) = t;
        }
        if (has_next) load_x(tile + gridDim.x);            // next tile's x in flight (issuing these between the product's k-steps was
                                                           // tried: each costs the wave ~600 cycles at the issue port either way)
      }
      BR_STAMP(4);      // first T(x) written
      __syncthreads();                                   // S_it: Zs[it & 1] complete (Xs is this wave's own)
      BR_STAMP(5);      // first barrier passed
    }
    if (it > 0) multiply(Zs + ((it - 1) & 1) * NT * kBwdRows * 16);          // the last tile
    // ---- the workgroup's slab: [dW (K x N) | db (N)], scaled by 1/(1-p) (folded out of T()) ----
    float* slab = a.slabs + (int64_t)blockIdx.x * a.slab_elems;
    if (q_live) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (2 * q + h < KT) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int n = nt * 16 + c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = (2 * q + h) * 16 + 4 * g + r;
              if (k < K && n < N) slab[(int64_t)k * N + n] = acc[h][nt][r] * a.inv_keep;
            }
          }
        }
      }
    }
    if (a.db_off >= 0) {
      const float tot = dbacc + __shfl_xor(dbacc, 32, 64);
      if (db_live && dbh == 0 && dbcol < N) slab[a.db_off + dbcol] = tot;
    }
  }
  BR_STAMP(8);
  if (IBN && a.in_sums) {
    __syncthreads();
    double* rep = a.in_sums + (size_t)(blockIdx.x % kBwdRep) * 2 * K;
    for (int k = threadIdx.x; k < K; k += kBwdThreads) {
      atomicAdd(rep + k, red[k]);
      atomicAdd(rep + K + k, red[Kp + k]);
    }
  }
  BR_STAMP_RT(11);
  BR_STAMP_FLUSH(blockIdx.x * 8 + wave);
}

}  // namespace br

using namespace br;

namespace br {
size_t dense_bwd_fused_lds(int NT, int KT) {
  const int Np = NT * 16, Kp = KT * 16;
  return sizeof(float) * ((size_t)Kp * (Np + 4) + 4 * (size_t)Np + 4 * (size_t)Kp + 2 * (size_t)NT * kBwdRows * 16 + 4 * 2 * (size_t)kBwdRows * 16 +
                          4 * 16 * (size_t)kBwdPatchLd) + sizeof(double) * 2 * (size_t)Kp;
}
}  // namespace br

template <int NT, int KT, bool IBN>
static void launch_bwd_v(unsigned grid, size_t shmem, hipStream_t s, const BwdArgs& a) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)dense_bwd_kernel<NT, KT, IBN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  dense_bwd_kernel<NT, KT, IBN><<<grid, kBwdThreads, shmem, s>>>(a);
}
template <int NT, int KT>
static void launch_bwd(unsigned grid, size_t shmem, hipStream_t s, const BwdArgs& a) {
  if (a.i_mean) launch_bwd_v<NT, KT, true>(grid, shmem, s, a);
  else launch_bwd_v<NT, KT, false>(grid, shmem, s, a);
}

namespace br {
// workgroups (= slabs written) of the fused backward
int dense_bwd_fused_grid(int64_t batch) {
  const int64_t t = ceil_div(batch, kBwdRows);
  return (int)(t < 1 ? 1 : (t > 256 ? 256 : t));
}

// One K-range (<= 128 columns) of a layer's backward in the fused kernel.  Returns BR_ERR_UNSUPPORTED when the LDS image does
// not fit (the caller falls back to the two-kernel path).
int dense_backward_fused(const BwdArgs& a, hipStream_t s) {
  const int NT = (a.N + 15) / 16, KT = (a.K + 15) / 16;
  const size_t shmem = dense_bwd_fused_lds(NT, KT);
  if (shmem > 160 * 1024) return BR_ERR_UNSUPPORTED;
  const unsigned grid = (unsigned)dense_bwd_fused_grid(a.batch);
#define BR_BW_K(NTv, KTv) case KTv: launch_bwd<NTv, KTv>(grid, shmem, s, a); break;
#define BR_BW(NTv) \
  case NTv:        \
    switch (KT) { BR_BW_K(NTv, 1) BR_BW_K(NTv, 2) BR_BW_K(NTv, 3) BR_BW_K(NTv, 4) BR_BW_K(NTv, 5) BR_BW_K(NTv, 6) BR_BW_K(NTv, 7) BR_BW_K(NTv, 8) default: break; } \
    break;
  switch (NT) {
    BR_BW(1) BR_BW(2) BR_BW(3) BR_BW(4) BR_BW(5) BR_BW(6) BR_BW(7) BR_BW(8)
    default: set_error("brDenseBackward: unsupported N"); return BR_ERR_UNSUPPORTED;
  }
  BR_CHECK_LAUNCH("brDenseBackward(fused)");
  return BR_OK;
}
}  // namespace br
