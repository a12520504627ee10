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

// EMU ("bf16x6", dense.h): every fp32 product runs as six bf16 MFMAs on three-piece operands, and every operand is split ONCE:
//   W   -> Wp[n-block J][piece][g][k][8 bf16] in LDS at staging (B fragments of the dx product, 16 B per lane; an odd last n-tile is a
//          half block of 8 B).  1.5 x the fp32 image.
//   dz  -> split by the dx wave that formed it: packed along n in registers (its own A fragments), and published element by element
//          (ds_write_b16) as Zp[n-tile][32-row block b][piece][g][n16][8 bf16]: the dW waves' B fragments, element i <-> row 32b + 4i + g.
//          Single-buffered (the LDS is full): barrier B_t "the dW waves are done with dz(t-1)" in front of the publish, A_t "dz(t) is
//          visible" behind it.
//   T(x)-> the dW wave loads its 32 columns as the fp32 kernel does (full 128-B lines, one tile ahead), transforms, turns one 32-row block at a
//          time through a private 4 KB LDS patch into the A layout (lane (c16, g) <- rows 32b + 4i + g of column c16) and splits in
//          registers.  (Loading x straight in the A layout - 32 dword loads of 4 rows x 64 B per tile and wave - clogged the CU's address
//          pipe: the dx waves' loads behind them took 3.5 x as long, 62 us for the layer instead of 53.)
//   db  -> ones^T dz: three more MFMAs (A = 1.0) per fragment on the dW wave that owns the n-tile (nt mod 4).
// The dz arithmetic, the dx epilogue and the slab layout are the fp32 kernel's, line for line.
template <int NT, int KT>
constexpr size_t bwd_emu_wp_floats() { return (size_t)(NT / 2) * 3 * 4 * (KT * 16) * 4 + (size_t)(NT & 1) * 3 * 4 * (KT * 16) * 2; }
// Zp: per (n-tile, 32-row block, piece) four lane groups g of 16 fragments x 16 B.  The g stride is 17 slots where the LDS has room for it
// (all shapes but N > 112 with K > 96): the dx waves publish element by element, lanes (row & 3) -> g, and with a 256-B stride their four
// groups hit the same banks (rocprofv3: 35 % of the 128 x 100 backward's LDS cycles were conflict cycles); the dW waves' 16-B fragment reads
// are contiguous per group either way.
constexpr int bwd_zp_tile(int zg) { return 2 * 3 * 4 * zg * 4; }       // floats of one n-tile of Zp (6 KB at zg = 16)
constexpr size_t bwd_emu_lds_zg(int NT, int KT, int zg) {
  const size_t Np = NT * 16, Kp = KT * 16;
  return sizeof(float) * ((size_t)(NT / 2) * 3 * 4 * Kp * 4 + (size_t)(NT & 1) * 3 * 4 * Kp * 2 + 4 * Np + 4 * Kp + (size_t)NT * bwd_zp_tile(zg) +
                          4 * 2 * 32 * 16 + 4 * 16 * (size_t)kBwdPatchLd) + sizeof(double) * 2 * Kp;
}
constexpr int bwd_zg(int NT, int KT) { return bwd_emu_lds_zg(NT, KT, 17) <= 160 * 1024 ? 17 : 16; }
constexpr size_t bwd_emu_lds(int NT, int KT) { return bwd_emu_lds_zg(NT, KT, bwd_zg(NT, KT)); }
constexpr bool bwd_emu_fits(int NT, int KT) { return bwd_emu_lds(NT, KT) <= 160 * 1024; }

template <int NT, int KT, bool IBN, bool EMU>
__global__ __launch_bounds__(kBwdThreads, EMU ? 1 : 2) void dense_bwd_kernel(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int Np = NT * 16, Kp = KT * 16, ldw = Np + 4;
  constexpr int KW = (KT + 1) / 2;                      // keep words per row
  constexpr int NBF = NT / 2, NB = (NT + 1) / 2;        // full 32-deep n-blocks of the dx product / all of them (an odd NT ends in a half block)
  constexpr int kWsFloats = EMU ? (int)bwd_emu_wp_floats<NT, KT>() : Kp * ldw;
  constexpr int ZG = bwd_zg(NT, KT), kZpTile = bwd_zp_tile(ZG);
  float* Ws = smem;                                     // fp32: [Kp][ldw] W row-major, zero padded.  EMU: the piece image Wp
  float* Cs = Ws + kWsFloats;                           // [4][Np]     out-BN constants
  float* Is = Cs + 4 * Np;                              // [mean Kp | rstd Kp] of the in BN
  float* ssb = Is + 2 * Kp;                             // [scale Kp | shift Kp]
  float* Zs = ssb + 2 * Kp;                             // [2][NT][64][16]  dz.  EMU: Zp, [NT] tiles of kZpTile floats
  float* Xs = Zs + (EMU ? NT * kZpTile : 2 * NT * kBwdRows * 16);      // [4 dW waves][2][64][16]  T(x).  EMU: [4][2][32][16], a transposition patch
  float* patches = Xs + 4 * 2 * (EMU ? 32 : kBwdRows) * 16;     // [4 dx waves][16][kBwdPatchLd]   (EMU: Xs holds one 32-row block per dW wave)
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
  // EMU: the W loads are REQUESTED before the first tile's loads (stage_w_issue) and consumed behind them (in stage()): the in-order vmcnt
  // then lets the wave wait for W alone - L2 hits that arrive within ~2 k cycles - and split it into the piece image while the tile's
  // 88 KB are still streaming in at the CU's ~10 B/clk of HBM; requested behind the tile they arrived last, ~14 k cycles after entry.
  constexpr int W_ITEMS = NB * KT, W_TRB = EMU ? (W_ITEMS * 64 + kBwdThreads - 1) / kBwdThreads : 1;
  float4 wlo[W_TRB], whi[W_TRB];
  auto stage_w_issue = [&]() {
    if constexpr (EMU) {
      const int k16 = threadIdx.x & 15, gg = (threadIdx.x >> 4) & 3;
      const bool n4 = (N & 3) == 0;
#pragma unroll
      for (int i = 0; i < W_TRB; ++i) {
        const int item = (threadIdx.x >> 6) + i * (kBwdThreads / 64);
        const int itc = item < W_ITEMS ? item : 0;
        const int J = itc / KT, kt = itc - J * KT, k = kt * 16 + k16;
        const float* wr = a.W + (int64_t)(k < K ? k : 0) * N;
        const int n0 = 32 * J + 4 * gg, n1 = n0 + 16;
        if (n4) {      // quads are wholly inside or outside N
          wlo[i] = *reinterpret_cast<const float4*>(wr + (n0 < N ? n0 : 0));
          whi[i] = *reinterpret_cast<const float4*>(wr + (n1 < N ? n1 : 0));
        } else {
          wlo[i] = make_float4(wr[n0 < N ? n0 : 0], wr[n0 + 1 < N ? n0 + 1 : 0], wr[n0 + 2 < N ? n0 + 2 : 0], wr[n0 + 3 < N ? n0 + 3 : 0]);
          whi[i] = make_float4(wr[n1 < N ? n1 : 0], wr[n1 + 1 < N ? n1 + 1 : 0], wr[n1 + 2 < N ? n1 + 2 : 0], wr[n1 + 3 < N ? n1 + 3 : 0]);
        }
      }
    }
  };
  auto stage = [&]() {
    if constexpr (EMU) {
      // W -> Wp straight from global memory: item = (n-block J, k-tile kt) x 64 lanes (k16 = lane & 15, gg = lane >> 4) holds the 8 n-values of
      // ITS B fragment (k = 16 kt + k16; n = 32J + 4gg..+3 and 32J + 16 + 4gg..+3: 16 rows x 64 B per load, the access shape of the gy
      // loads), splits them and writes three 16-B pieces (16 lanes = 256 contiguous bytes per g: conflict-free).  (The first version
      // staged the fp32 rows through the dz region in two chunks - four more barriers and two LDS round trips.)
      const int k16 = threadIdx.x & 15, gg = (threadIdx.x >> 4) & 3;
#pragma unroll
      for (int i = 0; i < W_TRB; ++i) {
        const int item = (threadIdx.x >> 6) + i * (kBwdThreads / 64);
        const int J = item / KT, kt = item - J * KT, k = kt * 16 + k16;
        const int n0 = 32 * J + 4 * gg, n1 = n0 + 16;
        const int kin = k < K ? -1 : 0;
        auto msk = [&](float v, int n) { return __int_as_float(__float_as_int(v) & kin & (n < N ? -1 : 0)); };
        uint32_t ph[4], pm[4], pl[4];
        split3(msk(wlo[i].x, n0), msk(wlo[i].y, n0 + 1), ph[0], pm[0], pl[0]);
        split3(msk(wlo[i].z, n0 + 2), msk(wlo[i].w, n0 + 3), ph[1], pm[1], pl[1]);
        split3(msk(whi[i].x, n1), msk(whi[i].y, n1 + 1), ph[2], pm[2], pl[2]);
        split3(msk(whi[i].z, n1 + 2), msk(whi[i].w, n1 + 3), ph[3], pm[3], pl[3]);
        if (item < W_ITEMS) {
          if (J < NBF) {
            float* d = Ws + (((J * 3 * 4 + gg) * Kp + k) << 2);
            *reinterpret_cast<uint4*>(d) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
            *reinterpret_cast<uint4*>(d + 4 * Kp * 4) = make_uint4(pm[0], pm[1], pm[2], pm[3]);
            *reinterpret_cast<uint4*>(d + 2 * 4 * Kp * 4) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
          } else {
            float* d = Ws + NBF * 3 * 4 * Kp * 4 + ((gg * Kp + k) << 1);
            *reinterpret_cast<uint2*>(d) = make_uint2(ph[0], ph[1]);
            *reinterpret_cast<uint2*>(d + 4 * Kp * 2) = make_uint2(pm[0], pm[1]);
            *reinterpret_cast<uint2*>(d + 2 * 4 * Kp * 2) = make_uint2(pl[0], pl[1]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
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
    if constexpr (!EMU) {   // W: 4 floats along n per thread (one 16-B load when N % 4 == 0).  Clamped addresses, no lane conditions around the loads and
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
      // the 16-B group starts inside N (rows are padded to 4 floats).  Only the last n-tile can straddle N (N > 16 (NT - 1)): written with
      // the runtime test 16j + 16 <= N for every j, hipcc kept one 64-bit select per j alive across the tile loop, spilled them, and
      // reloaded each one - s_waitcnt vmcnt(0) - right behind the previous prefetch load: every load of the next tile waited for the one before
      int g4 = 4 * g;
      if (j == NT - 1) asm volatile("" : "+v"(g4));          // (opaque: the select below is recomputed per tile - two VALU ops - instead of hoisted and spilled)
      const bool in = j < NT - 1 || 16 * j + g4 < N;
      const int off = in ? 16 * j : -g4;
      vg[j] = *reinterpret_cast<const float4*>(pg_n + off);
      vy[j] = *reinterpret_cast<const float4*>(py_n + off);
    };
    auto load_gy_unit = [&](int u) {                     // half of load_gy_j: unit 2j = gy, 2j + 1 = y
      const int j = u >> 1;
      int g4 = 4 * g;
      if (j == NT - 1) asm volatile("" : "+v"(g4));
      const bool in = j < NT - 1 || 16 * j + g4 < N;
      const int off = in ? 16 * j : -g4;
      if (u & 1) vy[j] = *reinterpret_cast<const float4*>(py_n + off);
      else vg[j] = *reinterpret_cast<const float4*>(pg_n + off);
    };
    auto load_gy = [&](int64_t tile) {
      gy_rows(tile);
#pragma unroll
      for (int j = 0; j < NT; ++j) load_gy_j(j);
    };
    int64_t tile = blockIdx.x;
    stage_w_issue();
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
      float* Zb = Zs + (EMU ? 0 : (it & 1)) * NT * kBwdRows * 16;
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
          const int lm = (live && (j < NT - 1 || n + e < N)) ? -1 : 0;            // 0 for rows past the batch / columns past N
          const float da = k1[e] * (ge[e] - k2[e] - (ye[e] - km[e]) * k3[e]);
          // act'(y) from the activation output: sigmoid y(1-y), relu [y > 0], linear 1 - blended with bit masks (no branches)
          const float ds = ye[e] * (1.f - ye[e]);
          const float dr = ye[e] > 0.f ? 1.f : 0.f;
          const float dl = __int_as_float((__float_as_int(ds) & sigmask) | (__float_as_int(dr) & relumask) | (0x3f800000 & ~(sigmask | relumask)));
          v[e] = __int_as_float(__float_as_int(da * dl) & lm);
        }
        dz[j] = make_float4(v[0], v[1], v[2], v[3]);
        if constexpr (!EMU) *reinterpret_cast<float4*>(Zb + (j * kBwdRows + rt * 16 + c16) * 16 + 4 * (g ^ ((c16 >> 2) & 3))) = dz[j];     // swizzled: see bwd_swz
        __builtin_amdgcn_sched_barrier(0);
      }
      // EMU: dz split once per tile into three bf16 pieces, packed along n as the A fragments of the 32-deep n-blocks of the dx product
      // (elements 0-3 from dz[2J], 4-7 from dz[2J+1]; zeros in the half block) ...
      uint32_t ah[EMU ? NB : 1][4], am[EMU ? NB : 1][4], al[EMU ? NB : 1][4];
      if constexpr (EMU) {
#pragma unroll
        for (int J = 0; J < NB; ++J) {
          split3(dz[2 * J].x, dz[2 * J].y, ah[J][0], am[J][0], al[J][0]);
          split3(dz[2 * J].z, dz[2 * J].w, ah[J][1], am[J][1], al[J][1]);
          if (J < NBF) {
            split3(dz[J < NBF ? 2 * J + 1 : 0].x, dz[J < NBF ? 2 * J + 1 : 0].y, ah[J][2], am[J][2], al[J][2]);
            split3(dz[J < NBF ? 2 * J + 1 : 0].z, dz[J < NBF ? 2 * J + 1 : 0].w, ah[J][3], am[J][3], al[J][3]);
          } else {
            ah[J][2] = ah[J][3] = am[J][2] = am[J][3] = al[J][2] = al[J][3] = 0u;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                 // B_it: the dW waves have multiplied tile it-1: the single dz image is free
        // ... and published element by element for the dW waves: element (row r = 16 rt + c16, column n = 16 j + 4 g + e), piece p ->
        // Zp[j][b = r >> 5][p][r & 3][n & 15] half-word (r >> 2) & 7.  Per lane a base + compile-time offsets; the low / high half of a packed
        // pair goes out with ds_write_b16 / ds_write_b16_d16_hi (no VALU).
        uint16_t* zp = reinterpret_cast<uint16_t*>(Zs) + (rt >> 1) * (3 * 4 * ZG * 8) + (((c16 & 3) * ZG + 4 * g) << 3) + 4 * (rt & 1) + (c16 >> 2);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int J = j >> 1, d = 2 * (j & 1) + (e >> 1);
            uint16_t* q = zp + j * (2 * 3 * 4 * ZG * 8) + e * 8;
            const uint32_t vh = ah[J][d], vm = am[J][d], vl = al[J][d];
            q[0] = (uint16_t)((e & 1) ? (vh >> 16) : (vh & 0xffffu));
            q[4 * ZG * 8] = (uint16_t)((e & 1) ? (vm >> 16) : (vm & 0xffffu));
            q[2 * 4 * ZG * 8] = (uint16_t)((e & 1) ? (vl >> 16) : (vl & 0xffffu));
          }
        }
        __syncthreads();                                 // A_it: dz(it) is visible to them
      }
      BR_STAMP(4);      // dz of the first tile published (includes the wait for gy / y)
      const bool has_next = tile + gridDim.x < n_tiles;                  // wave-uniform
      if (has_next) gy_rows(tile + gridDim.x);
      if (has_next && !a.gx) load_gy(tile + gridDim.x);                  // (no dx product to hide them behind)
      // keep words of this lane's 4 C-layout rows (rows 4g..4g+3 of the row tile) and, with an input BN, the raw x of its outputs
      // (EMU without an input BN: the keep bits are applied in ROW layout behind the transposition patch - one row, KW words per lane
      //  instead of four rows: 12 VGPRs that the 256-register budget of the bf16 variant does not have)
      constexpr bool ROWKEEP = EMU && !IBN;
      uint32_t kb[ROWKEEP ? 1 : 4][KW];
      if (ROWKEEP && a.keep && a.gx) {
        int64_t row = rbase + c16;
        row = row < batch ? row : batch - 1;
#pragma unroll
        for (int w = 0; w < KW; ++w) kb[0][w] = a.keep[row * a.kw + w];
      }
      if (!ROWKEEP && a.keep && a.gx) {
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
        // (EMU: lane coordinates recomputed from the thread id behind an opaque barrier - at its 256-VGPR limit hipcc spilled the 64-bit
        //  zero-extensions of c16 / 4g it keeps for this address, and a scratch reload waits for vmcnt(0), i.e. for every prefetch in flight)
        int c16e = c16, g4e = 4 * g;
        if constexpr (EMU) { c16e = threadIdx.x & 15; g4e = (threadIdx.x >> 2) & 12; asm volatile("" : "+v"(c16e), "+v"(g4e)); }
        float* gxrow = a.gx + (rbase + (row_ok ? c16e : 0)) * a.ldgx + g4e;       // row layout: &gx[rbase + c16][4g]
        const float* wpf = Ws + ((g * Kp + c16) << 2);                              // full blocks: + ((J*3 + p) * 4 * Kp + kt * 16) * 4
        const float* wph = Ws + NBF * 3 * 4 * Kp * 4 + ((g * Kp + c16) << 1);       // half block: + (p * 4 * Kp + kt * 16) * 2
        // ---- dx = dz · W^T in passes of 4 k-tiles: 4 independent accumulator chains ----
#pragma unroll
        for (int kt0 = 0; kt0 < KT; kt0 += 4) {
          constexpr int WMAX = 4;
          const int Wn = (KT - kt0) < WMAX ? (KT - kt0) : WMAX;
          f32x4 acc[WMAX];
#pragma unroll
          for (int w = 0; w < WMAX; ++w) acc[w] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if constexpr (EMU) {
            // step (J, w): the three B fragments of step + 1 are requested, then six MFMAs (small products first); a fence per step
            float4 bq[2][3];
            auto fetch_b = [&](int J, int w, float4 (&b)[3]) {
#pragma unroll
              for (int p = 0; p < 3; ++p) {
                if (J < NBF) b[p] = *reinterpret_cast<const float4*>(wpf + (((J * 3 + p) * 4 * Kp + (kt0 + w) * 16) << 2));
                else { const float2 t = *reinterpret_cast<const float2*>(wph + ((p * 4 * Kp + (kt0 + w) * 16) << 1)); b[p] = make_float4(t.x, t.y, 0.f, 0.f); }
              }
            };
            fetch_b(0, 0, bq[0]);
#pragma unroll
            for (int J = 0; J < NB; ++J) {
              const bf16x8 xh = frag8(ah[J]), xm = frag8(am[J]), xl = frag8(al[J]);
#pragma unroll
              for (int w = 0; w < WMAX; ++w) {
                if (w < Wn) {
                  const int st = J * Wn + w, cur = st & 1;
                  const int Jn = (w + 1 < Wn) ? J : J + 1, wn = (w + 1 < Wn) ? w + 1 : 0;
                  if (Jn < NB) fetch_b(Jn, wn, bq[cur ^ 1]);
                  const bf16x8 wh = frag8(bq[cur][0]), wm = frag8(bq[cur][1]), wl = frag8(bq[cur][2]);
                  acc[w] = mfma_bf16(xl, wh, acc[w]);
                  acc[w] = mfma_bf16(xh, wl, acc[w]);
                  acc[w] = mfma_bf16(xm, wm, acc[w]);
                  acc[w] = mfma_bf16(xm, wh, acc[w]);
                  acc[w] = mfma_bf16(xh, wm, acc[w]);
                  acc[w] = mfma_bf16(xh, wh, acc[w]);
                  // next tile's gy / y: its 2 NT loads spread evenly over the steps of ALL passes (every load is 1 KB against a CU share of
                  // HBM of ~9 B/clk: requested in a row they hold the wave at the issue port - behind the first pass alone, +3 000 cycles)
                  constexpr int STEPS = NB * KT, UN = 2 * NT;
                  const int gs = NB * kt0 + st;              // steps of the earlier passes: NB * kt0
                  if (has_next) {
#pragma unroll
                    for (int u = gs * UN / STEPS; u < (gs + 1) * UN / STEPS; ++u) load_gy_unit(u);
                  }
                  __builtin_amdgcn_sched_barrier(0);
                }
              }
            }
          } else
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
                if (!ROWKEEP && a.keep) dh = keep_bit(dh, kb[ROWKEEP ? 0 : r][kt >> 1], 16u * (kt & 1) + c16);
                if (IBN) {                                              // (dh is 0 for rows past the batch: dz is)
                  isum[kt] += dh;
                  isq[kt] = fmaf(dh, (xraw[r][kt] - xm) * xr, isq[kt]);
                }
                patch[(4 * g + r) * kBwdPatchLd + c16] = dh;
              }
              __builtin_amdgcn_wave_barrier();
              float4 o = *reinterpret_cast<const float4*>(patch + c16 * kBwdPatchLd + 4 * g);
              __builtin_amdgcn_wave_barrier();
              if (ROWKEEP && a.keep) {
                const uint32_t w_ = kb[0][kt >> 1], p0 = 16u * (kt & 1) + 4u * g;
                o.x = keep_bit(o.x, w_, p0); o.y = keep_bit(o.y, w_, p0 + 1); o.z = keep_bit(o.z, w_, p0 + 2); o.w = keep_bit(o.w, w_, p0 + 3);
              }
              if (row_ok && (kt * 16 + 16 <= K || kt * 16 + 4 * g < K)) *reinterpret_cast<float4*>(gxrow + kt * 16) = o;
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          BR_STAMP(6 + 2 * (kt0 / 4));      // pass epilogue issued
        }
      }
      if constexpr (!EMU) __syncthreads();               // S_it: Zs[it & 1] is published; the dW waves' readers of Zs[(it - 1) & 1] are done
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
  } else if constexpr (EMU) {
    // =========================================== dW waves, bf16x6 ===========================================
    const int q = wave - 4;                              // k-tiles 2q, 2q+1; db of the n-tiles nt = q (mod 4)
    const bool q_live = 2 * q < KT, h1 = 2 * q + 1 < KT; // wave-uniform
    float* Xw = Xs + q * 2 * 32 * 16;                    // [k-tile h][32 rows][16], swizzled as the fp32 kernel's image
    const int lr = lane >> 3, lc = lane & 7;             // x loads: row lr + 8i, columns 32q + 4lc..+3
    const int kcol = 32 * q + 4 * lc;
    const bool col_in = kcol < K;
    float4 xv[8];
    uint32_t kwl = 0xFFFFFFFFu;                          // keep word q (columns 32q..32q+31) of row `lane` of the tile
    auto load_x = [&](int64_t tile) {
      const int64_t r0 = tile * kBwdRows;
      if (a.keep) { const int64_t row = r0 + lane < batch ? r0 + lane : batch - 1; kwl = a.keep[row * a.kw + (q_live ? q : 0)]; }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int64_t row = r0 + lr + 8 * i;
        row = row < batch ? row : batch - 1;
        xv[i] = *reinterpret_cast<const float4*>(a.x + row * a.ldx + (col_in ? kcol : 0));
      }
    };
    int64_t tile = blockIdx.x;
    stage_w_issue();
    if (q_live && tile < n_tiles) load_x(tile);
    BR_STAMP(1);
    stage();
    BR_STAMP(2);
    f32x4 acc[2][NT];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[h][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NDB = (NT + 3) / 4;
    f32x4 accdb[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d) accdb[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();                                     // staging visible
    BR_STAMP(3);
    const float4 sc = *reinterpret_cast<const float4*>(ssb + (col_in ? kcol : 0)), sh = *reinterpret_cast<const float4*>(ssb + Kp + (col_in ? kcol : 0));
    const uint32_t ones2 = 0x3F803F80u;
    const uint32_t ones4[4] = {ones2, ones2, ones2, ones2};
    const bf16x8 ones = frag8(ones4);
    const float* zl = Zs + ((g * ZG + c16) << 2);        // + (((nt * 2 + b) * 3 + p) * 4 * ZG) * 4
    int it = 0;
    for (; tile < n_tiles; tile += gridDim.x, ++it) {
      const bool has_next = tile + gridDim.x < n_tiles;                      // wave-uniform
      // ---- T(x) = BN affine + keep bits, split into the A fragments of the tile (columns past K: zero) ----
      uint32_t xh[2][2][4], xm[2][2][4], xl[2][2][4];
      if (q_live) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {                    // rows lr + 8i of block b: T() as the fp32 kernel, into the patch
            const float4 xx = xv[4 * b + i];
            const uint32_t kx = __shfl(kwl, 32 * b + lr + 8 * i, 64);
            const uint32_t p0 = 4u * lc;
            float4 t;
            t.x = keep_bit(fmaf(xx.x, sc.x, sh.x), kx, p0); t.y = keep_bit(fmaf(xx.y, sc.y, sh.y), kx, p0 + 1);
            t.z = keep_bit(fmaf(xx.z, sc.z, sh.z), kx, p0 + 2); t.w = keep_bit(fmaf(xx.w, sc.w, sh.w), kx, p0 + 3);
            if (!col_in) t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (K & 3) {                                   // wave-uniform: a 16-B group may straddle K (padded rows)
              if (kcol + 1 >= K) t.y = 0.f;
              if (kcol + 2 >= K) t.z = 0.f;
              if (kcol + 3 >= K) t.w = 0.f;
            }
            *reinterpret_cast<float4*>(Xw + ((lc >> 2) * 32 + lr + 8 * i) * 16 + 4 * ((lc & 3) ^ (((lr + 8 * i) >> 2) & 3))) = t;
          }
          __builtin_amdgcn_wave_barrier();
          float t0[8], t1[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {                    // element i <- row 4i + g of the block, column c16 of k-tile h
            const int cs = bwd_swz(c16, i);
            t0[i] = Xw[(4 * i + g) * 16 + cs];
            t1[i] = Xw[(32 + 4 * i + g) * 16 + cs];
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int p2 = 0; p2 < 4; ++p2) {
            split3(t0[2 * p2], t0[2 * p2 + 1], xh[0][b][p2], xm[0][b][p2], xl[0][b][p2]);
            split3(t1[2 * p2], t1[2 * p2 + 1], xh[1][b][p2], xm[1][b][p2], xl[1][b][p2]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      BR_STAMP(4);      // first T(x) split
      __syncthreads();                                   // B_it: this wave is done with dz(it-1)
      __syncthreads();                                   // A_it: dz(it) is visible
      BR_STAMP(5);
      // ---- dW[32 x N] += T(x)^T · dz: two 32-row blocks x NT n-tiles; the three B fragments of step + 1 are requested before the MFMAs of a step ----
      {
        float4 bq[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[0][p] = *reinterpret_cast<const float4*>(zl + ((p * 4 * ZG) << 2));
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const bf16x8 a0h = frag8(xh[0][b]), a0m = frag8(xm[0][b]), a0l = frag8(xl[0][b]);
          const bf16x8 a1h = frag8(xh[1][b]), a1m = frag8(xm[1][b]), a1l = frag8(xl[1][b]);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int st = b * NT + nt, cur = st & 1;
            const int bn = (nt + 1 < NT) ? b : b + 1, ntn = (nt + 1 < NT) ? nt + 1 : 0;
            if (bn < 2) {
#pragma unroll
              for (int p = 0; p < 3; ++p) bq[cur ^ 1][p] = *reinterpret_cast<const float4*>(zl + ((((ntn * 2 + bn) * 3 + p) * 4 * ZG) << 2));
            }
            const bf16x8 wh = frag8(bq[cur][0]), wm = frag8(bq[cur][1]), wl = frag8(bq[cur][2]);
            if (q_live) {
              acc[0][nt] = mfma_bf16(a0l, wh, acc[0][nt]);
              acc[0][nt] = mfma_bf16(a0h, wl, acc[0][nt]);
              acc[0][nt] = mfma_bf16(a0m, wm, acc[0][nt]);
              acc[0][nt] = mfma_bf16(a0m, wh, acc[0][nt]);
              acc[0][nt] = mfma_bf16(a0h, wm, acc[0][nt]);
              acc[0][nt] = mfma_bf16(a0h, wh, acc[0][nt]);
            }
            if (h1) {
              acc[1][nt] = mfma_bf16(a1l, wh, acc[1][nt]);
              acc[1][nt] = mfma_bf16(a1h, wl, acc[1][nt]);
              acc[1][nt] = mfma_bf16(a1m, wm, acc[1][nt]);
              acc[1][nt] = mfma_bf16(a1m, wh, acc[1][nt]);
              acc[1][nt] = mfma_bf16(a1h, wm, acc[1][nt]);
              acc[1][nt] = mfma_bf16(a1h, wh, acc[1][nt]);
            }
            if ((nt & 3) == q) {                         // db: column sums of dz = ones^T dz (every row of the result tile carries them)
              accdb[nt >> 2] = mfma_bf16(ones, wl, accdb[nt >> 2]);
              accdb[nt >> 2] = mfma_bf16(ones, wm, accdb[nt >> 2]);
              accdb[nt >> 2] = mfma_bf16(ones, wh, accdb[nt >> 2]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if (it == 0) BR_STAMP(7);      // first product done
      // next tile's x: requested HERE, not in front of the barriers - this wave now waits for the dx waves (whose product and epilogue take
      // ~3 x as long as the product above), so the back-pressure of a burst of loads (a CU's share of HBM is ~9 B/clk) stalls an idle wave,
      // and the burst is over before the dx waves request the next gy / y
      if (q_live && has_next) load_x(tile + gridDim.x);
    }
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
    if (a.db_off >= 0 && g == 0) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        if ((nt & 3) == q && nt * 16 + c16 < N) slab[a.db_off + nt * 16 + c16] = accdb[nt >> 2][0];
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
    stage_w_issue();
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

template <int NT, int KT, bool IBN, bool EMU>
static void launch_bwd_v(unsigned grid, size_t shmem, hipStream_t s, const BwdArgs& a) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)dense_bwd_kernel<NT, KT, IBN, EMU>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  dense_bwd_kernel<NT, KT, IBN, EMU><<<grid, kBwdThreads, shmem, s>>>(a);
}
template <int NT, int KT>
static void launch_bwd(unsigned grid, size_t shmem, hipStream_t s, const BwdArgs& a) {
  if constexpr (bwd_emu_fits(NT, KT)) {
    if (mlp_bf16x6()) {
      const size_t sh = bwd_emu_lds(NT, KT);
      if (a.i_mean) launch_bwd_v<NT, KT, true, true>(grid, sh, s, a);
      else launch_bwd_v<NT, KT, false, true>(grid, sh, s, a);
      return;
    }
  }
  if (a.i_mean) launch_bwd_v<NT, KT, true, false>(grid, shmem, s, a);
  else launch_bwd_v<NT, KT, false, false>(grid, shmem, s, a);
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
