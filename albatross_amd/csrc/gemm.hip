// gemm.hip — fp64 MFMA update kernel  C <- C - A * B^T  (K3 trailing update,
// K4/K6 multi-RHS solves and joint covariance).
//
// This is where the N^3/3 flop of the factorisation go (the work Eigen's
// LDLT::compute does at eigen/serializable_ldlt.hpp:27 in the reference).
//
// One workgroup = 4 waves = one 128 x 128 tile of C; each wave owns a 64 x 64
// quadrant = 4 x 4 tiles of v_mfma_f64_16x16x4_f64 (128 accumulator VGPRs).
// The K loop streams 16-deep chunks of both operand panels global -> registers
// -> LDS (double buffered, one barrier per chunk); fragments are one
// ds_read_b64 per lane from a [k][row] image padded to 144 doubles per k so
// the two 16-lane halves of a 32-lane LDS group fall on disjoint banks.
// The MFMA "A" operand carries the C-column panel (pre-negated while staging)
// and the MFMA "B" operand the C-row panel, so that a C/D register holds 16
// CONSECUTIVE ROWS of one C column: epilogue accesses are 128-B segments of
// the column-major matrix.
//
// Roofline: MFMA-bound.  Per tile 2*128*128*K flop against (2*128*K + 2*128*128)
// * 8 B of operand + C traffic (K = 512: 64 flop/B).
#include <cmath>
#include <cstdlib>
#include "common.h"
#include "mfma_f64.h"
#include "gemm_tiles.h"

namespace agp {




// Which C tile a workgroup computes.  Returns false for padding workgroups.
__device__ __forceinline__ bool tile_of_block(const GemmArgs &g, int &bi, int &bj) {
  if (g.remap == 2) {
    // XCD-aware order for the triangular bulk update.  Workgroups b, b + 8, ... run on the same
    // XCD (round-robin dispatch) and share its L2.  The tiles are dealt in SUPER-COLUMNS of 4
    // tile columns; inside one, 8 consecutive ids walk down 8 tile rows (one per XCD) and the
    // next 8 ids take the next column of the SAME rows: an XCD therefore runs the 4 tiles of a
    // tile row back to back (their row strip of the panel is fetched from HBM/MALL once instead
    // of four times) while the 4 column strips stay resident for the whole super-column.
    // Every super-column is padded to a multiple of 8 rows so that (id mod 8) keeps meaning XCD.
    long long l = blockIdx.x;
    int sc = 0;
    while (true) {
      const int rows = g.ntr - 4 * sc;
      const long long cnt = (long long)((rows + 7) / 8) * 32;
      if (l < cnt) break;
      l -= cnt;
      ++sc;
    }
    const int rr = (int)(l & 7), q = (int)(l >> 3);
    bj = 4 * sc + (q & 3);
    bi = 4 * sc + (q >> 2) * 8 + rr;
    return bi < g.ntr && bj < g.ntc && bi >= bj;
  }
  if (g.remap) {
    const unsigned b = blockIdx.x;
    const int xcd = (int)(b & 7), l = (int)(b >> 3);
    const int s = (l >> 6) * 8 + xcd, within = l & 63;
    if (s >= g.nsuper) return false;
    int sj = 0, left = s;
    while (left >= g.nb8 - sj) { left -= g.nb8 - sj; ++sj; }
    const int si = sj + left;
    bi = 8 * si + (within & 7);
    bj = 8 * sj + (within >> 3);
    return bi < g.ntr && bj < g.ntc && bi >= bj;
  }
  if (g.stair) {
    const long long id = blockIdx.x;
    bi = (int)(id % g.ntr);
    bj = (int)(id / g.ntr);
    const long long lb = g.st_lb0 + bi / g.st_tpb;
    const long long gi = lb * g.st_world + ((lb & 1) ? g.st_world - 1 - g.st_rank : g.st_rank);
    return bj <= gi * g.st_tpb - g.st_c0t + (bi % g.st_tpb);
  }
  bj = 0;
  long long id = blockIdx.x + g.tile_first;
  while (true) {
    const int cnt = g.tri ? (g.ntr - bj) : g.ntr;
    if (id < cnt) break;
    id -= cnt;
    ++bj;
  }
  bi = (g.tri ? bj : 0) + (int)id;
  return true;
}

template <bool A_KMAJOR, bool B_KMAJOR>
__device__ __forceinline__ void gemm_nt_sub_body(const GemmArgs &g, double *lds) {
  int bi, bj;
  if (!tile_of_block(g, bi, bj)) return;
  gemm_nt_sub_tile<A_KMAJOR, B_KMAJOR>(g, bi, bj, lds);
}

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_nt_sub_kernel(GemmArgs g) {
  __builtin_amdgcn_s_setprio(AGP_CHAIN_PRIO);  // everything but the bulk update (its own kernel below)
  g.C += (long long)blockIdx.y * g.batch_C;
  g.A += (long long)blockIdx.y * g.batch_A;
  g.B += (long long)blockIdx.y * g.batch_B;
  // one LDS array: [buffer][operand][k][row]
  __shared__ double lds[2 * 2 * GK * GLD];
  gemm_nt_sub_body<A_KMAJOR, B_KMAJOR>(g, lds);
}

// ---------------------------------------------------------------------------
// fp64 VALU variant of the same update: "DPP-broadcast" register tiling.
//
// On gfx950 the fp64 MFMA pipe saturates at ~48 TFLOP/s while plain v_fmac_f64
// issues at full vector rate (profiles/r01/microbench_fp64.txt: 58-69 TFLOP/s,
// clock-limited).  A classic register-tiled FMA kernel is LDS-bound (an 8 x 8
// tile per lane needs 16 LDS doubles per 64 FMAs).  Here the COLUMN operand is
// not replicated per lane: a wave keeps ONE VGPR pair with 64 different column
// values (lane l holds column l of the wave's 64) and every FMA reads it through
// DPP  row_newbcast:c  — each 16-lane row broadcasts ITS lane c — so
//
//   lane (rho = l >> 4, i = l & 15) owns rows {2i, 2i+1, 32+2i, 33+2i} x columns 16 rho + 0..15
//   wave = 64 x 64 of C, workgroup = 2 x 2 waves = the same 128 x 128 tile as the MFMA kernel
//   k-step = 2 ds_read_b128 (row pairs; the four 16-lane rows read the same addresses)
//          + 1 ds_read_b64 (64 consecutive columns) + 64 v_fmac_f64_dpp     (LDS port: ~8 %)
//
// The FMAs and LDS reads are inline asm with explicit s_waitcnt: written in C++
// the compiler reorders them for register pressure and the issue pattern is lost.
// Accumulators start from C (the LDS row image is negated), the epilogue is pure stores.
// ---------------------------------------------------------------------------
typedef double dpp_d2 __attribute__((ext_vector_type(2)));

template <int C>
__device__ __forceinline__ void fmac_bcast(double &acc, double b, double a) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc) : "v"(b), "v"(a), "n"(C));
}

template <int C>
__device__ __forceinline__ void fmac_bcast4(double (&acc)[4][16], double b, const dpp_d2 &a01, const dpp_d2 &a23) {
  fmac_bcast<C>(acc[0][C], b, a01.x);
  fmac_bcast<C>(acc[1][C], b, a01.y);
  fmac_bcast<C>(acc[2][C], b, a23.x);
  fmac_bcast<C>(acc[3][C], b, a23.y);
}

__device__ __forceinline__ void fmac_step(double (&acc)[4][16], double b, const dpp_d2 &a01, const dpp_d2 &a23) {
  fmac_bcast4<0>(acc, b, a01, a23);  fmac_bcast4<1>(acc, b, a01, a23);
  fmac_bcast4<2>(acc, b, a01, a23);  fmac_bcast4<3>(acc, b, a01, a23);
  fmac_bcast4<4>(acc, b, a01, a23);  fmac_bcast4<5>(acc, b, a01, a23);
  fmac_bcast4<6>(acc, b, a01, a23);  fmac_bcast4<7>(acc, b, a01, a23);
  fmac_bcast4<8>(acc, b, a01, a23);  fmac_bcast4<9>(acc, b, a01, a23);
  fmac_bcast4<10>(acc, b, a01, a23); fmac_bcast4<11>(acc, b, a01, a23);
  fmac_bcast4<12>(acc, b, a01, a23); fmac_bcast4<13>(acc, b, a01, a23);
  fmac_bcast4<14>(acc, b, a01, a23); fmac_bcast4<15>(acc, b, a01, a23);
}

// operands of k-row K of the current chunk (byte offsets are literals: no address arithmetic)
template <int K>
__device__ __forceinline__ void dpp_lds_read(dpp_d2 &a01, dpp_d2 &a23, double &b, unsigned a_addr, unsigned b_addr) {
  asm volatile("ds_read_b128 %0, %3 offset:%5\n\tds_read_b128 %1, %3 offset:%6\n\tds_read_b64 %2, %4 offset:%5"
               : "=&v"(a01), "=&v"(a23), "=&v"(b)
               : "v"(a_addr), "v"(b_addr), "n"(K * GLD * 8), "n"(K * GLD * 8 + 256));
}

template <int K>
__device__ __forceinline__ void dpp_k_steps(double (&acc)[4][16], dpp_d2 (&a01)[2], dpp_d2 (&a23)[2], double (&b)[2],
                                            unsigned a_addr, unsigned b_addr) {
  constexpr int cur = K & 1, nxt = cur ^ 1;
  if constexpr (K + 1 < GK) dpp_lds_read<K + 1>(a01[nxt], a23[nxt], b[nxt], a_addr, b_addr);
  fmac_step(acc, b[cur], a01[cur], a23[cur]);
  if constexpr (K + 1 < GK) {
    // LDS returns in order and nothing else is outstanding: the operands of step K + 1 have landed
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a01[nxt]), "+v"(a23[nxt]), "+v"(b[nxt]));
    dpp_k_steps<K + 1>(acc, a01, a23, b, a_addr, b_addr);
  }
}

// debug: summed shader-clock cycles, 100 MHz ticks and workgroup count of the main loops
__device__ unsigned long long g_valu_clock[4];

template <bool A_KMAJOR, bool B_KMAJOR>
__device__ __forceinline__ void gemm_dpp_body(const GemmArgs &g, double *lds) {
  int bi, bj;
  if (!tile_of_block(g, bi, bj)) return;
  const long long i0 = (long long)bi * GT, j0 = (long long)bj * GT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 15, rho = lane >> 4;

  const bool a_vec = (((reinterpret_cast<uintptr_t>(g.A)) & 15) == 0) && ((g.lda & 1) == 0);
  const bool b_vec = (((reinterpret_cast<uintptr_t>(g.B)) & 15) == 0) && ((g.ldb & 1) == 0);
  const bool c_vec = ((g.ldc & 1) == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0);

  double ra[8], rb[8];
  const long long nk = (g.K + GK - 1) / GK;
  load_chunk<A_KMAJOR>(g.A, g.lda, i0, g.M, 0, g.K, a_vec, ra);
  load_chunk<B_KMAJOR>(g.B, g.ldb, j0, g.N, 0, g.K, b_vec, rb);

  // accumulators start from C (entries outside C: zero, never stored)
  const long long row0 = i0 + 64 * wr + 2 * li, col0 = j0 + 64 * wc + 16 * rho;
  double acc[4][16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long long row = row0 + 32 * h, col = col0 + c;
      const double *cp = g.C + row + col * g.ldc;
      dpp_d2 v = {0., 0.};
      if (col < g.N) {
        if (c_vec && row + 1 < g.M) v = *reinterpret_cast<const dpp_d2 *>(cp);
        else {
          if (row < g.M) v.x = cp[0];
          if (row + 1 < g.M) v.y = cp[1];
        }
      }
      acc[2 * h][c] = v.x;
      acc[2 * h + 1][c] = v.y;
    }
  }

  store_chunk<A_KMAJOR, true>(lds, ra);  // rows negated: acc = C + (-a) b
  store_chunk<B_KMAJOR, false>(lds + GK * GLD, rb);
  __syncthreads();

  // LDS byte addresses (the truncated flat address of a __shared__ object is its LDS offset)
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>(lds);
  const unsigned a_lane = lds0 + 8u * (unsigned)(64 * wr + 2 * li);
  const unsigned b_lane = lds0 + 8u * (unsigned)(GK * GLD + 64 * wc + lane);

  const unsigned long long t_c0 = __builtin_amdgcn_s_memtime(), t_r0 = __builtin_amdgcn_s_memrealtime();
  for (long long kc = 0; kc < nk; ++kc) {
    const unsigned cur = (unsigned)(kc & 1) * (2u * GK * GLD * 8u);
    const bool more = kc + 1 < nk;
    if (more) {
      load_chunk<A_KMAJOR>(g.A, g.lda, i0, g.M, (kc + 1) * GK, g.K, a_vec, ra);
      load_chunk<B_KMAJOR>(g.B, g.ldb, j0, g.N, (kc + 1) * GK, g.K, b_vec, rb);
    }
    dpp_d2 a01[2], a23[2];
    double b[2];
    dpp_lds_read<0>(a01[0], a23[0], b[0], a_lane + cur, b_lane + cur);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a01[0]), "+v"(a23[0]), "+v"(b[0]));
    dpp_k_steps<0>(acc, a01, a23, b, a_lane + cur, b_lane + cur);
    if (more) {
      double *An = lds + ((kc + 1) & 1) * (2 * GK * GLD);
      store_chunk<A_KMAJOR, true>(An, ra);
      store_chunk<B_KMAJOR, false>(An + GK * GLD, rb);
    }
    __syncthreads();
  }

  if (threadIdx.x == 0) {
    atomicAdd(&g_valu_clock[0], __builtin_amdgcn_s_memtime() - t_c0);
    atomicAdd(&g_valu_clock[1], __builtin_amdgcn_s_memrealtime() - t_r0);
    atomicAdd(&g_valu_clock[2], 1ull);
  }
  // epilogue: pure stores; per instruction a 16-lane row writes 256 contiguous bytes of one column
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const long long col = col0 + c;
    if (col >= g.N) continue;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long long row = row0 + 32 * h;
      double *cp = g.C + row + col * g.ldc;
      if (c_vec && row + 1 < g.M) {
        dpp_d2 v = {acc[2 * h][c], acc[2 * h + 1][c]};
        *reinterpret_cast<dpp_d2 *>(cp) = v;
      } else {
        if (row < g.M) cp[0] = acc[2 * h][c];
        if (row + 1 < g.M) cp[1] = acc[2 * h + 1][c];
      }
    }
  }
}

__global__ __launch_bounds__(GEMM_THREADS, 2) void trailing_update_valu_kernel(GemmArgs g) {
  __shared__ double lds[2 * 2 * GK * GLD];
  gemm_dpp_body<false, false>(g, lds);
}

// The bulk trailing update of the factorisation (C -= P P^T, lower tiles, K =
// NBO) under its own kernel symbol, so that profiles and bench.py's roofline
// block isolate exactly these launches.
// -DAGP_CLOCK_PROBE (scripts/clock_probe.sh, never the product build): every workgroup adds its shader-clock cycles
// (s_memtime) and 100 MHz ticks (s_memrealtime) to g_mfma_clock - the SCLK the bulk update actually holds
__device__ unsigned long long g_mfma_clock[4];

__global__ __launch_bounds__(GEMM_THREADS, 2) void trailing_update_kernel(GemmArgs g) {
#if AGP_BULK_PRIO > 0
  __builtin_amdgcn_s_setprio(AGP_BULK_PRIO);
#endif
  __shared__ double lds[2 * 2 * GK * GLD];
#ifdef AGP_CLOCK_PROBE
  const unsigned long long t_c0 = __builtin_amdgcn_s_memtime(), t_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  gemm_nt_sub_body<false, false>(g, lds);
#ifdef AGP_CLOCK_PROBE
  if (threadIdx.x == 0) {
    atomicAdd(&g_mfma_clock[0], __builtin_amdgcn_s_memtime() - t_c0);
    atomicAdd(&g_mfma_clock[1], __builtin_amdgcn_s_memrealtime() - t_r0);
    atomicAdd(&g_mfma_clock[2], 1ull);
  }
#endif
}

// ---------------------------------------------------------------------------
// Mixed-precision bulk update (BASELINE config 4): the SAME tile, but the two
// panels are rounded to fp32 while they are staged into LDS and multiplied with
// v_mfma_f32_16x16x4_f32 (4x the fp64 MFMA issue rate); the K <= 512 products of
// one launch accumulate in fp32 registers and are then subtracted from the fp64
// matrix, so the rounding of one outer step never compounds over the next.
// Lane maps as in mfma_f64.h except C/D: register r of lane l holds
// D[m = 4 (l >> 4) + r][n = l & 15].
// ---------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NEGATE>
__device__ __forceinline__ void store_chunk_f32(float *__restrict__ Ls, const double (&r)[8]) {
  const int t = threadIdx.x;
  const int kk = t >> 4, seg = (t & 15) * 8;
  float4 *dst = reinterpret_cast<float4 *>(Ls + kk * GLD + seg);
  const float sg = NEGATE ? -1.f : 1.f;
  dst[0] = make_float4(sg * (float)r[0], sg * (float)r[1], sg * (float)r[2], sg * (float)r[3]);
  dst[1] = make_float4(sg * (float)r[4], sg * (float)r[5], sg * (float)r[6], sg * (float)r[7]);
}

__global__ __launch_bounds__(GEMM_THREADS, 2) void trailing_update_f32_kernel(GemmArgs g) {
  __shared__ float lds[2 * 2 * GK * GLD];
  int bi, bj;
  if (!tile_of_block(g, bi, bj)) return;
  const long long i0 = (long long)bi * GT, j0 = (long long)bj * GT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int ln = lane & 15, lg = lane >> 4;
  const bool a_vec = (((reinterpret_cast<uintptr_t>(g.A)) & 15) == 0) && ((g.lda & 1) == 0);
  const bool b_vec = (((reinterpret_cast<uintptr_t>(g.B)) & 15) == 0) && ((g.ldb & 1) == 0);

  v4f acc[4][4];  // [tj][ti]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};

  double ra[8], rb[8];
  const long long nk = (g.K + GK - 1) / GK;
  load_chunk<false>(g.A, g.lda, i0, g.M, 0, g.K, a_vec, ra);
  load_chunk<false>(g.B, g.ldb, j0, g.N, 0, g.K, b_vec, rb);
  store_chunk_f32<false>(lds, ra);
  store_chunk_f32<true>(lds + GK * GLD, rb);
  __syncthreads();
  for (long long kc = 0; kc < nk; ++kc) {
    const int cur = (int)(kc & 1);
    const float *As = lds + cur * (2 * GK * GLD);
    const float *Bs = As + GK * GLD;
    const bool more = kc + 1 < nk;
    if (more) {
      load_chunk<false>(g.A, g.lda, i0, g.M, (kc + 1) * GK, g.K, a_vec, ra);
      load_chunk<false>(g.B, g.ldb, j0, g.N, (kc + 1) * GK, g.K, b_vec, rb);
    }
#pragma unroll
    for (int s = 0; s < GK / 4; ++s) {
      float fa[4], fb[4];
      const int krow = (4 * s + lg) * GLD;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t] = Bs[krow + 64 * wc + 16 * t + ln];  // MFMA A operand: C-column panel (negated)
        fb[t] = As[krow + 64 * wr + 16 * t + ln];  // MFMA B operand: C-row panel
      }
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
          acc[tj][ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[tj], fb[ti], acc[tj][ti], 0, 0, 0);
    }
    if (more) {
      float *An = lds + (cur ^ 1) * (2 * GK * GLD);
      store_chunk_f32<false>(An, ra);
      store_chunk_f32<true>(An + GK * GLD, rb);
    }
    __syncthreads();
  }
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      const long long row = i0 + 64 * wr + 16 * ti + ln;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long col = j0 + 64 * wc + 16 * tj + 4 * lg + r;
        if (row < g.M && col < g.N) {
          double *c = g.C + row + col * g.ldc;
          *c = *c + (double)acc[tj][ti][r];
        }
      }
    }
}


__global__ __launch_bounds__(GEMM_THREADS, 4) void gemm64_nt_sub_kernel(GemmArgs g) {
  __builtin_amdgcn_s_setprio(AGP_CHAIN_PRIO);  // panel-chain updates: issue ahead of co-resident bulk-update waves
  g.C += (long long)blockIdx.y * g.batch_C;
  g.A += (long long)blockIdx.y * g.batch_A;
  g.B += (long long)blockIdx.y * g.batch_B;
  __shared__ double lds[2 * 2 * GK * SLD];
  int bj = 0;
  long long id = blockIdx.x;
  while (true) {
    // tri: column bj of 64-tiles holds the row tiles bi >= bj
    const int cnt = g.tri ? (g.ntr - bj) : g.ntr;
    if (id < cnt) break;
    id -= cnt;
    ++bj;
  }
  const int bi = (g.tri ? bj : 0) + (int)id;
  gemm64_body(g, (long long)bi * ST, (long long)bj * ST, lds);
}

// 64 x 64 tiles with a TRANSPOSED second operand (B(j, k) at B[k + j * ldb]): the updates of the multi-RHS
// substitutions with few right-hand sides, where a 128 x 128 tile would be mostly padding.  Not triangular.
__global__ __launch_bounds__(GEMM_THREADS, 4) void gemm64_nt_sub_bk_kernel(GemmArgs g) {
  __builtin_amdgcn_s_setprio(AGP_CHAIN_PRIO);
  g.C += (long long)blockIdx.y * g.batch_C;
  g.A += (long long)blockIdx.y * g.batch_A;
  g.B += (long long)blockIdx.y * g.batch_B;
  __shared__ double lds[2 * 2 * GK * SLD];
  const long long id = blockIdx.x;
  const int bj = (int)(id / g.ntr), bi = (int)(id % g.ntr);
  if (g.remap) gemm64_body<true, true>(g, (long long)bi * ST, (long long)bj * ST, lds);  // remap = 1: A transposed too
  else gemm64_body<true, false>(g, (long long)bi * ST, (long long)bj * ST, lds);
}

// The LAST tiles of a bulk update (those that would form a partial round of 128 x 128 workgroups) as
// four 64 x 64 workgroups each: the tail of the launch is a quarter as long.  ntr / ntc / tile_first
// are in 128-tile units like trailing_update_kernel's.
__global__ __launch_bounds__(GEMM_THREADS, 4) void trailing_update_tail_kernel(GemmArgs g) {
  __shared__ double lds[2 * 2 * GK * SLD];
  int bj = 0;
  long long id = (long long)(blockIdx.x >> 2) + g.tile_first;
  while (true) {
    const int cnt = g.ntr - bj;
    if (id < cnt) break;
    id -= cnt;
    ++bj;
  }
  const int bi = bj + (int)id;
  const int q = blockIdx.x & 3, qi = q & 1, qj = q >> 1;
  if (bi == bj && qj > qi) return;  // upper quadrant of a diagonal tile
  gemm64_body(g, (long long)bi * GT + qi * ST, (long long)bj * GT + qj * ST, lds);
}

static long long count_tiles(int ntr, int ntc, int tri) {
  long long total = 0;
  for (int bj = 0; bj < ntc; ++bj) total += tri ? (ntr - bj > 0 ? ntr - bj : 0) : ntr;
  return total;
}

// C(M x N) -= A(M x K) * B(N x K)^T ; tri != 0 keeps only tiles on/below the diagonal.
// count > 1: the same product for `count` independent problems whose operands are batch_* elements apart.
void launch_gemm_nt_sub_batched(hipStream_t s, double *C, long long ldc, long long batch_C, const double *A,
                                long long lda, bool a_kmajor, long long batch_A, const double *B, long long ldb,
                                bool b_kmajor, long long batch_B, long long M, long long N, long long K, bool tri,
                                long long count) {
  if (M <= 0 || N <= 0 || K <= 0 || count <= 0) return;
  GemmArgs g;
  g.C = C; g.ldc = ldc; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb;
  g.M = M; g.N = N; g.K = K; g.tri = tri ? 1 : 0;
  g.remap = 0; g.nsuper = 0; g.nb8 = 0;
  g.batch_C = batch_C; g.batch_A = batch_A; g.batch_B = batch_B;
  g.ntr = (int)((M + GT - 1) / GT);
  g.ntc = (int)((N + GT - 1) / GT);
  if (tri && g.ntc > g.ntr) g.ntc = g.ntr;
  const long long tiles = count_tiles(g.ntr, g.ntc, g.tri);
  if (tiles <= 0) return;
  // fewer 128-tiles than workgroup slots (2 per CU): use 64 x 64 tiles instead
  static int small_limit = -1;
  if (small_limit < 0) {
    const char *e = getenv("AGP_SMALL_TILE_LIMIT");
    small_limit = e ? atoi(e) : 512;
  }
  if (b_kmajor && !tri && tiles * count < small_limit) {
    // launches that cannot fill the chip with 128 x 128 tiles (few right-hand sides, or the inner updates of a
    // substitution with few rows): 64 x 64 tiles with the transposed-operand loader(s).  N = 16384, predict
    // marginal: M = 64: 10.5 -> 5.0 ms, M = 1024: 12.3 -> 8.9 ms, M = 4096: 25.8 -> 24.8 ms.
    GemmArgs h = g;
    h.remap = a_kmajor ? 1 : 0;  // (the XCD remap field is unused by this kernel: it selects the A loader)
    h.ntr = (int)((M + ST - 1) / ST);
    h.ntc = (int)((N + ST - 1) / ST);
    hipLaunchKernelGGL(gemm64_nt_sub_bk_kernel, dim3((unsigned)((long long)h.ntr * h.ntc), (unsigned)count), dim3(GEMM_THREADS), 0, s,
                       h);
    return;
  }
  if (!a_kmajor && !b_kmajor && tiles * count < small_limit) {
    GemmArgs h = g;
    h.ntr = (int)((M + ST - 1) / ST);
    h.ntc = (int)((N + ST - 1) / ST);
    if (tri && h.ntc > h.ntr) h.ntc = h.ntr;
    const long long t64 = count_tiles(h.ntr, h.ntc, h.tri);
    hipLaunchKernelGGL(gemm64_nt_sub_kernel, dim3((unsigned)t64, (unsigned)count), dim3(GEMM_THREADS), 0, s, h);
    return;
  }
  dim3 grid((unsigned)tiles, (unsigned)count), block(GEMM_THREADS);
  if (!a_kmajor && !b_kmajor) hipLaunchKernelGGL((gemm_nt_sub_kernel<false, false>), grid, block, 0, s, g);
  else if (!a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_nt_sub_kernel<false, true>), grid, block, 0, s, g);
  else if (a_kmajor && !b_kmajor) hipLaunchKernelGGL((gemm_nt_sub_kernel<true, false>), grid, block, 0, s, g);
  else hipLaunchKernelGGL((gemm_nt_sub_kernel<true, true>), grid, block, 0, s, g);
}

// C (M x N: the stacked local row blocks lb0.. of rank `rank` of `world`, columns from global column c0) -= A B^T on the
// tiles of the staircase only (see GemmArgs::stair).  block = rows per row block (multiple of 128); c0 a multiple of 128.
void launch_gemm_nt_sub_stair(hipStream_t s, double *C, long long ldc, const double *A, long long lda, const double *B,
                              long long ldb, long long M, long long N, long long K, int world, int rank, long long lb0,
                              long long block, long long c0) {
  if (M <= 0 || N <= 0 || K <= 0) return;
  GemmArgs g;
  g.C = C; g.ldc = ldc; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb;
  g.M = M; g.N = N; g.K = K; g.tri = 0;
  g.remap = 0; g.nsuper = 0; g.nb8 = 0;
  g.ntr = (int)((M + GT - 1) / GT);
  g.ntc = (int)((N + GT - 1) / GT);
  g.stair = 1; g.st_world = world; g.st_rank = rank; g.st_tpb = (int)(block / GT);
  g.st_lb0 = lb0; g.st_c0t = c0 / GT;
  hipLaunchKernelGGL((gemm_nt_sub_kernel<false, false>), dim3((unsigned)((long long)g.ntr * g.ntc)), dim3(GEMM_THREADS), 0, s, g);
}

// C (M x M, lower 64 x 64 tiles) -= P P^T with the 64-tile kernel whatever the size; the tiles of the first done_cols tile
// columns are counted per row tile in done[bi] (GemmArgs::done64): the far trailing update of a step launch
void launch_update64_counted(hipStream_t s, double *C, long long ldc, const double *P, long long ldp, long long M, long long K,
                             unsigned long long *done, int done_cols) {
  if (M <= 0 || K <= 0) return;
  GemmArgs h;
  h.C = C; h.ldc = ldc; h.A = P; h.lda = ldp; h.B = P; h.ldb = ldp;
  h.M = M; h.N = M; h.K = K; h.tri = 1;
  h.remap = 0; h.nsuper = 0; h.nb8 = 0;
  h.ntr = h.ntc = (int)((M + ST - 1) / ST);
  h.done64 = done; h.done64_cols = done_cols;
  const long long t64 = count_tiles(h.ntr, h.ntc, 1);
  hipLaunchKernelGGL(gemm64_nt_sub_kernel, dim3((unsigned)t64, 1), dim3(GEMM_THREADS), 0, s, h);
}

void launch_gemm_nt_sub(hipStream_t s, double *C, long long ldc, const double *A, long long lda,
                        bool a_kmajor, const double *B, long long ldb, bool b_kmajor, long long M,
                        long long N, long long K, bool tri) {
  launch_gemm_nt_sub_batched(s, C, ldc, 0, A, lda, a_kmajor, 0, B, ldb, b_kmajor, 0, M, N, K, tri, 1);
}

// variant 0: MFMA kernel, 2: DPP-broadcast VALU kernel (experiment), 3: fp32-product MFMA kernel (mixed precision)
// entries on or below the diagonal of C covered by the first `count` lower tiles (column-major tile order)
static double lower_entries(long long M, int ntr, long long count) {
  double e = 0.;
  for (int bj = 0; bj < ntr && count > 0; ++bj) {
    const long long w = (M - (long long)bj * GT < GT) ? M - (long long)bj * GT : GT;  // tile column width
    for (int bi = bj; bi < ntr && count > 0; ++bi, --count) {
      const long long h = (M - (long long)bi * GT < GT) ? M - (long long)bi * GT : GT;
      e += (bi == bj) ? 0.5 * (double)w * (double)(w + 1) : (double)h * (double)w;
    }
  }
  return e;
}

void launch_trailing_update_as(int variant, hipStream_t s, double *C, long long ldc, const double *P,
                               const double *Q, long long ldp, long long M, long long K, BulkTiming *timing,
                               unsigned long long *done, int done_cols) {
  if (timing) timing->flops = 0.;
  if (M <= 0 || K <= 0) return;
  GemmArgs g;
  g.done = done; g.done_cols = done_cols;
  g.C = C; g.ldc = ldc; g.A = P; g.lda = ldp; g.B = Q; g.ldb = ldp;
  g.M = M; g.N = M; g.K = K; g.tri = 1;
  g.ntr = (int)((M + GT - 1) / GT);
  g.ntc = g.ntr;
  g.remap = 0; g.nsuper = 0; g.nb8 = 0;
  long long tiles = count_tiles(g.ntr, g.ntc, 1);
  static int use_remap = -1;
  if (use_remap < 0) {
    // AGP_XCD_REMAP=1: 8 x 8 super-tiles per XCD (measured slower: uneven super-tiles on the
    // diagonal); 2: super-columns of 4 tile columns, see tile_of_block
    const char *e = getenv("AGP_XCD_REMAP");
    use_remap = e ? atoi(e) : 0;
  }
  if (use_remap == 2 && g.ntr >= 16) {
    g.remap = 2;
    tiles = 0;
    for (int sc = 0; 4 * sc < g.ntr; ++sc) tiles += (long long)((g.ntr - 4 * sc + 7) / 8) * 32;
  } else if (use_remap == 1 && g.ntr >= 16) {
    g.remap = 1;
    g.nb8 = (g.ntr + 7) / 8;
    g.nsuper = g.nb8 * (g.nb8 + 1) / 2;
    tiles = (long long)((g.nsuper + 7) / 8) * 8 * 64;
  }
  if ((variant == 0 || variant == 4 || variant == 5) && g.remap == 0) {
    // Tail split: a launch of T tiles runs floor(T / slots) full rounds of 128 x 128 workgroups (slots = 2 per
    // CU); the T mod slots tiles left over would occupy a fraction of the chip for a whole further round.
    // They go to trailing_update_tail_kernel as four 64 x 64 workgroups each.
    static int split = -1, slots = 512;
    if (split < 0) {
      // AGP_TAIL_SPLIT=0: one launch of 128 x 128 tiles (the behaviour before the tail split)
      const char *e = getenv("AGP_TAIL_SPLIT");
      split = e ? atoi(e) : 1;
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess &&
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
        slots = 2 * cus;
    }
    const bool do_split = variant == 5 || (variant == 0 && split == 1);
    long long full = tiles, rem = 0;
    if (variant == 4 || (variant == 0 && split == 2)) { full = 0; rem = tiles; }  // AGP_TAIL_SPLIT=2: 64-tiles only
    else if (do_split) {
      if (variant == 0 && tiles < 4LL * slots) {
        // fewer than four rounds of large tiles: 64 x 64 workgroups throughout balance the CUs better
        // (scripts/time_tail_split.py: M = 4096, K = 512: 0.196 ms instead of 0.272)
        full = 0; rem = tiles;
      } else {
        full = (tiles / slots) * slots;
        rem = tiles - full;
        if (rem * 4 >= 3LL * slots) { full = tiles; rem = 0; }  // an almost full round: leave it to the large tiles
      }
    }
    if (full > 0) {
      if (timing && timing->e0) (void)hipEventRecord(timing->e0, s);
      hipLaunchKernelGGL(trailing_update_kernel, dim3((unsigned)full), dim3(GEMM_THREADS), 0, s, g);
      if (timing && timing->e1) {
        (void)hipEventRecord(timing->e1, s);
        timing->flops = 2. * (double)K * lower_entries(M, g.ntr, full);
      }
    }
    if (rem > 0) {
      GemmArgs h = g;
      h.done = nullptr;  // (the counted tiles are the first ones of the launch above)
      h.tile_first = full;
      hipLaunchKernelGGL(trailing_update_tail_kernel, dim3((unsigned)(4 * rem)), dim3(GEMM_THREADS), 0, s, h);
    }
    return;
  }
  if (variant == 2) hipLaunchKernelGGL(trailing_update_valu_kernel, dim3((unsigned)tiles), dim3(GEMM_THREADS), 0, s, g);
  else if (variant == 3) hipLaunchKernelGGL(trailing_update_f32_kernel, dim3((unsigned)tiles), dim3(GEMM_THREADS), 0, s, g);
  else hipLaunchKernelGGL(trailing_update_kernel, dim3((unsigned)tiles), dim3(GEMM_THREADS), 0, s, g);
}

// C (M x N, lower tiles: C(0, 0) sits on the matrix diagonal) -= P Q^T with fp32-rounded panels on the fp32 MFMA path
// (trailing_update_f32_kernel), the result subtracted from the fp64 matrix: the next-block-column update U1 of the
// mixed-precision factorisation (agp_fit_create_mixed)
void launch_update_f32(hipStream_t s, double *C, long long ldc, const double *P, const double *Q, long long ldp, long long M,
                       long long N, long long K) {
  if (M <= 0 || N <= 0 || K <= 0) return;
  GemmArgs g;
  g.C = C; g.ldc = ldc; g.A = P; g.lda = ldp; g.B = Q; g.ldb = ldp;
  g.M = M; g.N = N; g.K = K; g.tri = 1;
  g.ntr = (int)((M + GT - 1) / GT);
  g.ntc = (int)((N + GT - 1) / GT);
  if (g.ntc > g.ntr) g.ntc = g.ntr;
  g.remap = 0; g.nsuper = 0; g.nb8 = 0;
  const long long tiles = count_tiles(g.ntr, g.ntc, 1);
  if (tiles > 0) hipLaunchKernelGGL(trailing_update_f32_kernel, dim3((unsigned)tiles), dim3(GEMM_THREADS), 0, s, g);
}

// number of tiles the first launch (trailing_update_kernel, full rounds) of launch_trailing_update_as(0, ...) covers at size M:
// a merged update may only count tile columns that lie entirely inside it
long long trailing_update_full_tiles(long long M) {
  const int ntr = (int)((M + GT - 1) / GT);
  const long long tiles = count_tiles(ntr, ntr, 1);
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
    cus = 256;
  const long long slots = 2LL * cus;
  if (tiles < 4 * slots) return 0;
  const long long full = (tiles / slots) * slots, rem = tiles - full;
  return (rem * 4 >= 3 * slots) ? tiles : full;
}

void launch_trailing_update(hipStream_t s, double *C, long long ldc, const double *P, const double *Q,
                            long long ldp, long long M, long long K, BulkTiming *timing) {
  static int variant = -1;
  if (variant < 0) {
    // mfma (default) | dpp.  Measured on MI355X (scripts/time_update.py, M = 15872, K = 512): MFMA
    // 48 TFLOP/s at 2.38 GHz; the VALU kernel issues 88 % of its FMA slots but the chip drops to
    // ~1.77 GHz under fp64 vector load (power), which leaves it at 44-45 TFLOP/s.
    const char *e = getenv("AGP_UPDATE_KERNEL");
    variant = (e && e[0] == 'd') ? 2 : 0;
  }
  launch_trailing_update_as(variant, s, C, ldc, P, Q, ldp, M, K, timing, nullptr, 0);
}

void read_mfma_clock(unsigned long long out[4], bool reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mfma_clock), sizeof(unsigned long long) * 4);
  if (reset) {
    unsigned long long z[4] = {0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mfma_clock), z, sizeof(z));
  }
}

void read_valu_clock(unsigned long long out[4], bool reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_valu_clock), sizeof(unsigned long long) * 4);
  if (reset) {
    unsigned long long z[4] = {0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_valu_clock), z, sizeof(z));
  }
}

// ---- bare MFMA issue loop: measured fp64 matrix peak of this device ----------
__global__ __launch_bounds__(256) void mfma_peak_kernel(double *sink, int iters, double a0, double b0) {
  v4d acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = v4zero();
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = mfma16(a, b, acc[i]);
  }
  double s = 0.;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456) sink[0] = s;  // keep the loop live
}

int mfma_f64_peak(hipStream_t s, int iters, double *tflops) {
  double *sink = nullptr;
  if (hipMalloc(&sink, 8) != hipSuccess) return AGP_ERR_HIP;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int blocks = 256 * 2;  // 2 workgroups of 4 waves per CU: 2 waves per SIMD
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, s, sink, 16, 1.0, 2.0);
  (void)hipEventRecord(e0, s);
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, s, sink, iters, 1.0, 2.0);
  (void)hipEventRecord(e1, s);
  hipError_t e = hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 4.0 * (double)iters * 8.0 * 2.0 * 16 * 16 * 4;
  *tflops = flop / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  return e == hipSuccess ? AGP_OK : AGP_ERR_HIP;
}

}  // namespace agp
