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
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include "common.h"
#include "mfma_f64.h"
#include "gemm_tiles.h"

namespace agp {




// Which C tile a workgroup computes.  Returns false for padding workgroups.
__device__ __forceinline__ bool tile_of_block(const GemmArgs &g, int &bi, int &bj) {
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
  if (tile_takes_cpf(g, bi, bj)) gemm_nt_sub_tile_cpf<A_KMAJOR, B_KMAJOR>(g, bi, bj, lds);
  else gemm_nt_sub_tile<A_KMAJOR, B_KMAJOR>(g, bi, bj, lds);
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

// C = Cin - A B^T or C = A B^T (GemmArgs::Cin / assign): the products of the out-of-place substitution against a very
// wide right-hand side (forward_solve_wide, solve.hip).  B always k-major (the right-hand side's own rows).
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_nt_ext_kernel(GemmArgs g) {
  __shared__ double lds[2 * 2 * GK * GLD];
  const int bi = (int)(blockIdx.x % g.ntr), bj = (int)(blockIdx.x / g.ntr);
  if (tile_takes_cpf(g, bi, bj)) gemm_nt_sub_tile_cpf<false, true, true>(g, bi, bj, lds);
  else gemm_nt_sub_tile<false, true, true>(g, bi, bj, lds);
}

// The bulk trailing update of the factorisation (C -= P P^T, lower tiles, K =
// NBO) under its own kernel symbol, so that profiles and bench.py's roofline
// block isolate exactly these launches.
// The tiles that would form a partial LAST round of 128 x 128 workgroups go FIRST, as four 64 x 64 workgroups each
// (small_first / small_count): the launch ends with full rounds, and no second launch (round 3: trailing_update_tail_kernel
// behind this one - 50-100 us per update during which a fraction of the chip worked) is needed for them.
// Merged update (GemmArgs::head_cols): the tiles of the next block column come before everything else and are counted.
// -DAGP_BULK_STAMPS (scripts/build_variant.sh bulk_stamps -DAGP_BULK_STAMPS; scripts/probe_bulk_clock.py): the workgroup in the
// middle of the grid leaves the shader cycles (s_memtime) and the 100 MHz ticks (s_memrealtime) of its whole tile: the clock
// the chip holds under the fp64 bulk update = 100 MHz x cycles / ticks.
#ifdef AGP_BULK_STAMPS
__device__ unsigned long long g_bulk_probe[4];
void read_bulk_probe(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bulk_probe), sizeof(unsigned long long) * 4); }
#else
void read_bulk_probe(unsigned long long *out) { for (int i = 0; i < 4; ++i) out[i] = 0; }
#endif

__global__ __launch_bounds__(GEMM_THREADS, 2) void trailing_update_kernel(GemmArgs g) {
  __shared__ double lds[2 * 2 * GK * GLD];
#ifdef AGP_BULK_STAMPS
  const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  long long wg = blockIdx.x;
  const bool head = wg < g.head_count;
  int bi, bj = 0;
  if (head) {
    const int nth = g.ntr + g.head_cols;  // tile rows of the whole trailing matrix
    long long id = wg;
    while (id >= nth - bj) {
      id -= nth - bj;
      ++bj;
    }
    bi = bj + (int)id;
  } else {
    wg -= g.head_count;
    const bool small = wg < 4LL * g.small_count;
    long long id = small ? (wg >> 2) + g.small_first : wg - 4LL * g.small_count + g.tile_first;
    if (!small && g.order) {
      const int packed = g.order[id];
      if (packed < 0) return;
      bi = packed >> 16;
      bj = packed & 0xffff;
    } else {
      while (true) {
        const int cnt = g.ntr - bj;
        if (id < cnt) break;
        id -= cnt;
        ++bj;
      }
      bi = bj + (int)id;
    }
    // (the sub-triangle right of the head, in the frame of the whole trailing matrix)
    bi += g.head_cols;
    bj += g.head_cols;
    if (small) {
      const int q = (int)(wg & 3), qi = q & 1, qj = q >> 1;
      if (bi == bj && qj > qi) return;  // upper quadrant of a diagonal tile
      gemm64_body(g, (long long)bi * GT + qi * ST, (long long)bj * GT + qj * ST, lds);
      return;
    }
  }
  if (tile_takes_cpf(g, bi, bj)) gemm_nt_sub_tile_cpf<false, false>(g, bi, bj, lds);
  else gemm_nt_sub_tile<false, false>(g, bi, bj, lds);
#ifdef AGP_BULK_STAMPS
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    g_bulk_probe[0] = __builtin_amdgcn_s_memtime() - stamp_c0;
    g_bulk_probe[1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
  }
#endif
  if (head) {
    // every store of the tile acknowledged, then one count: the RELEASE writes this XCD's L2 back, and the kernels that
    // read the tile start behind the gate kernel that saw the count (their start is the matching ACQUIRE)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(g.head_done, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
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
  // (r: row pairs 2 (t & 15) + 32 j of k-row t >> 4, as load_chunk delivers them)
  const int kk = t >> 4, seg = (t & 15) * 2;
  float2 *dst = reinterpret_cast<float2 *>(Ls + kk * GLD + seg);
  const float sg = NEGATE ? -1.f : 1.f;
  dst[0] = make_float2(sg * (float)r[0], sg * (float)r[1]);
  dst[16] = make_float2(sg * (float)r[2], sg * (float)r[3]);
  dst[32] = make_float2(sg * (float)r[4], sg * (float)r[5]);
  dst[48] = make_float2(sg * (float)r[6], sg * (float)r[7]);
}

// chunk (128 rows x 16 k) of an fp32 panel copy: thread t -> k = t >> 4, rows (t & 15) * 8 .. + 7
__device__ __forceinline__ void load_chunk_p32(const float *__restrict__ P, long long ld, long long row0, long long k0, float4 &r0,
                                               float4 &r1) {
  const int t = threadIdx.x;
  const float *p = P + (row0 + (t & 15) * 8) + (k0 + (t >> 4)) * ld;
  r0 = *reinterpret_cast<const float4 *>(p);
  r1 = *reinterpret_cast<const float4 *>(p + 4);
}
template <bool NEGATE>
__device__ __forceinline__ void store_chunk_p32(float *__restrict__ Ls, const float4 r0, const float4 r1) {
  const int t = threadIdx.x;
  float4 *dst = reinterpret_cast<float4 *>(Ls + (t >> 4) * GLD + (t & 15) * 8);
  if (NEGATE) {
    dst[0] = make_float4(-r0.x, -r0.y, -r0.z, -r0.w);
    dst[1] = make_float4(-r1.x, -r1.y, -r1.z, -r1.w);
  } else {
    dst[0] = r0;
    dst[1] = r1;
  }
}

// P32 (rows x K, ld32) = (float) P (rows x K, ldp): the panel of one outer step, once, for all the tiles that read it
__global__ __launch_bounds__(256) void convert_panel_f32_kernel(const double *__restrict__ P, long long ldp, long long rows, float *__restrict__ P32,
                                                                long long ld32) {
  const long long r = ((long long)blockIdx.x * 256 + threadIdx.x) * 2, k = blockIdx.y;
  if (r + 1 < rows) {
    const double2 v = *reinterpret_cast<const double2 *>(P + r + k * ldp);
    *reinterpret_cast<float2 *>(P32 + r + k * ld32) = make_float2((float)v.x, (float)v.y);
  } else if (r < rows) {
    P32[r + k * ld32] = (float)P[r + k * ldp];
  }
}

void launch_convert_panel_f32(hipStream_t s, const double *P, long long ldp, long long rows, long long K, float *P32, long long ld32) {
  if (rows <= 0 || K <= 0) return;
  hipLaunchKernelGGL(convert_panel_f32_kernel, dim3((unsigned)((rows + 511) / 512), (unsigned)K), dim3(256), 0, s, P, ldp, rows, P32, ld32);
}

template <bool P32>
__global__ __launch_bounds__(GEMM_THREADS, 2) void trailing_update_f32_kernel(GemmArgs g) {
  __shared__ float lds[2 * 2 * GK * GLD];
  int bi, bj;
  if (g.order) {  // XCD-aware order (launch_trailing_update_as)
    const int packed = g.order[blockIdx.x];
    if (packed < 0) return;
    bi = packed >> 16;
    bj = packed & 0xffff;
  } else if (!tile_of_block(g, bi, bj)) return;
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
  if (tile_takes_cpf(g, bi, bj)) {
    // interior tile, deep product: C (fp64, 64 values per lane) comes in during the K loop in eight parts and the
    // epilogue only adds and stores (gemm_nt_sub_tile_cpf; here the fp32 accumulators leave room for all of C) - at the
    // fp32 MFMA's rate the K loop of a tile is 4x shorter and the read-modify-write behind it weighed 4x as much
    const long long nq = nk / 8;
    constexpr bool p32 = P32;
    float4 a32_0 = {}, a32_1 = {}, b32_0 = {}, b32_1 = {};
    if constexpr (p32) {
      load_chunk_p32(g.A32, g.ld32, i0, 0, a32_0, a32_1);
      load_chunk_p32(g.B32, g.ld32, j0, 0, b32_0, b32_1);
      store_chunk_p32<false>(lds, a32_0, a32_1);
      store_chunk_p32<true>(lds + GK * GLD, b32_0, b32_1);
    } else {
      load_chunk_interior<false>(g.A, g.lda, i0, 0, ra);
      load_chunk_interior<false>(g.B, g.ldb, j0, 0, rb);
      store_chunk_f32<false>(lds, ra);
      store_chunk_f32<true>(lds + GK * GLD, rb);
    }
    __syncthreads();
    double *const cbase = g.C + (i0 + 64 * wr + ln) + (j0 + 64 * wc + 4 * lg) * g.ldc;
    constexpr int NPRE = 7;  // parts of C fetched during the loop (all eight: 4 more registers than a wave has with fp64 staging; no gain with fp32 staging)
    double cpre[4][4][4];  // [tj][ti][r]
    long long kc = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const long long k_end = (p == 7) ? nk : (p + 1) * nq;
      bool first = true;
      for (; kc < k_end; ++kc) {
        const int cur = (int)(kc & 1);
        const float *As = lds + cur * (2 * GK * GLD);
        const float *Bs = As + GK * GLD;
        const bool more = kc + 1 < nk;
        if (more) {
          if constexpr (p32) {
            load_chunk_p32(g.A32, g.ld32, i0, (kc + 1) * GK, a32_0, a32_1);
            load_chunk_p32(g.B32, g.ld32, j0, (kc + 1) * GK, b32_0, b32_1);
          } else {
            load_chunk_interior<false>(g.A, g.lda, i0, (kc + 1) * GK, ra);
            load_chunk_interior<false>(g.B, g.ldb, j0, (kc + 1) * GK, rb);
          }
        }
        if (first && p < NPRE) {
          first = false;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              cpre[p >> 1][2 * (p & 1) + t][r] = __builtin_nontemporal_load(&cbase[16 * (2 * (p & 1) + t) + (long long)(16 * (p >> 1) + r) * g.ldc]);
        }
#pragma unroll
        for (int s = 0; s < GK / 4; ++s) {
          float fa[4], fb[4];
          const int krow = (4 * s + lg) * GLD;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            fa[t] = Bs[krow + 64 * wc + 16 * t + ln];
            fb[t] = As[krow + 64 * wr + 16 * t + ln];
          }
#pragma unroll
          for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
              acc[tj][ti] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[tj], fb[ti], acc[tj][ti], 0, 0, 0);
        }
        if (more) {
          float *An = lds + (cur ^ 1) * (2 * GK * GLD);
          if constexpr (p32) {
            store_chunk_p32<false>(An, a32_0, a32_1);
            store_chunk_p32<true>(An + GK * GLD, b32_0, b32_1);
          } else {
            store_chunk_f32<false>(An, ra);
            store_chunk_f32<true>(An + GK * GLD, rb);
          }
        }
        __syncthreads();
      }
    }
#pragma unroll
    for (int p = NPRE; p < 8; ++p)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          cpre[p >> 1][2 * (p & 1) + t][r] = cbase[16 * (2 * (p & 1) + t) + (long long)(16 * (p >> 1) + r) * g.ldc];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(cpre[tj][ti][r] + (double)acc[tj][ti][r], &cbase[16 * ti + (long long)(16 * tj + r) * g.ldc]);
    return;
  }
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
  if (g.stair) {  // (see GemmArgs::stair; st_tpb and st_c0t are in 64-tile units here)
    const long long sid = blockIdx.x;
    const int sbi = (int)(sid % g.ntr), sbj = (int)(sid / g.ntr);
    const long long lb = g.st_lb0 + sbi / g.st_tpb;
    const long long gi = lb * g.st_world + ((lb & 1) ? g.st_world - 1 - g.st_rank : g.st_rank);
    if (sbj > gi * g.st_tpb - g.st_c0t + (sbi % g.st_tpb)) return;
    gemm64_body(g, (long long)sbi * ST, (long long)sbj * ST, lds);
    return;
  }
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

__global__ __launch_bounds__(GEMM_THREADS, 6) void gemm32_nt_sub_kernel(GemmArgs g) {
  __builtin_amdgcn_s_setprio(AGP_CHAIN_PRIO);
  g.C += (long long)blockIdx.y * g.batch_C;
  g.A += (long long)blockIdx.y * g.batch_A;
  g.B += (long long)blockIdx.y * g.batch_B;
  __shared__ double lds[2 * 2 * GK * TLD];
  int bj = 0;
  long long id = blockIdx.x;
  while (true) {
    const int cnt = g.tri ? (g.ntr - bj) : g.ntr;
    if (id < cnt) break;
    id -= cnt;
    ++bj;
  }
  const int bi = (g.tri ? bj : 0) + (int)id;
  gemm32_body(g, (long long)bi * TT, (long long)bj * TT, lds);
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
  if (g.a_kmajor) gemm64_body<true, true>(g, (long long)bi * ST, (long long)bj * ST, lds);
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

static void launch_trailing_update_planned(int variant, hipStream_t s, GemmArgs &g, BulkTiming *timing);
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
  g.batch_C = batch_C; g.batch_A = batch_A; g.batch_B = batch_B;
  g.ntr = (int)((M + GT - 1) / GT);
  g.ntc = (int)((N + GT - 1) / GT);
  if (tri && g.ntc > g.ntr) g.ntc = g.ntr;
  const long long tiles = count_tiles(g.ntr, g.ntc, g.tri);
  if (tiles <= 0) return;
  // fewer 128-tiles than workgroup slots (2 per CU): use 64 x 64 tiles instead (128 ... 512: flat, profiles/r03)
  static const int small_limit = [] { const char *e = getenv("AGP_GEMM_SMALL_LIMIT"); return e && e[0] ? atoi(e) : 512; }();
  if (b_kmajor && !tri && tiles * count < small_limit) {
    // launches that cannot fill the chip with 128 x 128 tiles (few right-hand sides, or the inner updates of a
    // substitution with few rows): 64 x 64 tiles with the transposed-operand loader(s).  N = 16384, predict
    // marginal: M = 64: 10.5 -> 5.0 ms, M = 1024: 12.3 -> 8.9 ms, M = 4096: 25.8 -> 24.8 ms.
    GemmArgs h = g;
    h.a_kmajor = a_kmajor ? 1 : 0;
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
    if (t64 * count < 256 && K >= 256) {
      // fewer 64-tiles than CUs and a deep product: every wave would carry K / 4 x 4 MFMAs one behind the other on a
      // mostly idle chip - 32 x 32 tiles (1536 x 512 x 512: 58 -> 2x us next to a bulk update)
      h.ntr = (int)((M + TT - 1) / TT);
      h.ntc = (int)((N + TT - 1) / TT);
      if (tri && h.ntc > h.ntr) h.ntc = h.ntr;
      const long long t32 = count_tiles(h.ntr, h.ntc, h.tri);
      hipLaunchKernelGGL(gemm32_nt_sub_kernel, dim3((unsigned)t32, (unsigned)count), dim3(GEMM_THREADS), 0, s, h);
      return;
    }
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
  g.ntr = (int)((M + GT - 1) / GT);
  g.ntc = (int)((N + GT - 1) / GT);
  g.stair = 1; g.st_world = world; g.st_rank = rank; g.st_tpb = (int)(block / GT);
  g.st_lb0 = lb0; g.st_c0t = c0 / GT;
  // Few rounds of 128 x 128 tiles (a rank's share of a sharded fit: 12 tile rows): the last round is mostly idle slots -
  // 64 x 64 tiles balance the CUs better (the single-GPU bulk update makes the same switch below four rounds)
  static int slots = 0;
  if (slots == 0) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    slots = 2 * cus;
  }
  if ((long long)g.ntr * g.ntc / 2 < 4LL * slots) {
    g.ntr = (int)((M + ST - 1) / ST);
    g.ntc = (int)((N + ST - 1) / ST);
    g.st_tpb = (int)(block / ST);
    g.st_c0t = c0 / ST;
    hipLaunchKernelGGL(gemm64_nt_sub_kernel, dim3((unsigned)((long long)g.ntr * g.ntc)), dim3(GEMM_THREADS), 0, s, g);
    return;
  }
  hipLaunchKernelGGL((gemm_nt_sub_kernel<false, false>), dim3((unsigned)((long long)g.ntr * g.ntc)), dim3(GEMM_THREADS), 0, s, g);
}

// C (M x N, ldc) = Cin (ldcin) - A B^T, or C = A B^T if Cin == nullptr;  A (M x K): element (i, k) at A[i + k lda],
// B (N x K): element (j, k) at B[k + j ldb]
void launch_gemm_nt_ext(hipStream_t s, double *C, long long ldc, const double *Cin, long long ldcin, const double *A, long long lda,
                        const double *B, long long ldb, long long M, long long N, long long K) {
  if (M <= 0 || N <= 0 || K <= 0) return;
  GemmArgs g;
  g.C = C; g.ldc = ldc; g.A = A; g.lda = lda; g.B = B; g.ldb = ldb;
  g.M = M; g.N = N; g.K = K; g.tri = 0;
  g.ntr = (int)((M + GT - 1) / GT);
  g.ntc = (int)((N + GT - 1) / GT);
  g.Cin = Cin; g.ldcin = ldcin; g.assign = Cin ? 0 : 1;
  hipLaunchKernelGGL(gemm_nt_ext_kernel, dim3((unsigned)((long long)g.ntr * g.ntc)), dim3(GEMM_THREADS), 0, s, g);
}

void launch_gemm_nt_sub(hipStream_t s, double *C, long long ldc, const double *A, long long lda,
                        bool a_kmajor, const double *B, long long ldb, bool b_kmajor, long long M,
                        long long N, long long K, bool tri) {
  launch_gemm_nt_sub_batched(s, C, ldc, 0, A, lda, a_kmajor, 0, B, ldb, b_kmajor, 0, M, N, K, tri, 1);
}

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

// The order in which the whole tiles of a bulk update are handed out, for a part whose workgroup i runs on XCD i % 8:
// the lower-triangular tile grid is cut into 8 x 8 blocks of tiles; every XCD gets whole blocks (longest first to the
// least loaded XCD, then single tiles moved until the XCDs differ by at most one tile), so that the 64 workgroups resident
// on an XCD at a time read 16 panel strips between them instead of 65 - the strips are what the update fetches from the
// Infinity Cache / HBM (measured: FETCH_SIZE 5.8 -> 2.95 GB per launch at M = 15872, the launch 2 % faster;
// profiles/r04/bulk_update_vs_k.txt).
// Tables are cached per (tile rows, tile count) and device; they are a few KB each.
struct XcdOrder { int *dev = nullptr; long long len = 0; };
static XcdOrder xcd_order(int ntr, long long full) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, long long>, XcdOrder> cache;
  int device = 0;
  (void)hipGetDevice(&device);
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_tuple(device, ntr, full);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  constexpr int SB = 8, XCDS = 8;  // (blocks of 4 ... 16 tiles a side measure the same)
  // column-major index of tile (bi, bj) in the lower-triangular order: tiles before column bj + (bi - bj)
  auto index_of = [&](int bi, int bj) { return (long long)bj * ntr - (long long)bj * (bj - 1) / 2 + (bi - bj); };
  const int nsb = (ntr + SB - 1) / SB;
  std::vector<std::vector<int>> blocks;
  for (int SJ = 0; SJ < nsb; ++SJ)
    for (int SI = SJ; SI < nsb; ++SI) {
      std::vector<int> t;
      for (int tj = 0; tj < SB; ++tj)
        for (int ti = 0; ti < SB; ++ti) {
          const int bi = SI * SB + ti, bj = SJ * SB + tj;
          if (bi < ntr && bj <= bi && index_of(bi, bj) < full) t.push_back((bi << 16) | bj);
        }
      if (!t.empty()) blocks.push_back(std::move(t));
    }
  std::stable_sort(blocks.begin(), blocks.end(), [](const std::vector<int> &a, const std::vector<int> &b) { return a.size() > b.size(); });
  std::vector<std::vector<int>> per(XCDS);
  for (auto &b : blocks) {
    int best = 0;
    for (int x = 1; x < XCDS; ++x) if (per[x].size() < per[best].size()) best = x;
    per[best].insert(per[best].end(), b.begin(), b.end());
  }
  while (true) {  // single tiles from the longest list to the shortest
    int lo = 0, hi = 0;
    for (int x = 1; x < XCDS; ++x) { if (per[x].size() < per[lo].size()) lo = x; if (per[x].size() > per[hi].size()) hi = x; }
    if (per[hi].size() <= per[lo].size() + 1) break;
    per[lo].push_back(per[hi].back());
    per[hi].pop_back();
  }
  size_t L = 0;
  for (auto &v : per) L = std::max(L, v.size());
  std::vector<int> table(L * XCDS, -1);
  for (int x = 0; x < XCDS; ++x)
    for (size_t q = 0; q < per[x].size(); ++q) table[q * XCDS + x] = per[x][q];
  XcdOrder o;
  o.len = (long long)table.size();
  if (hipMalloc(&o.dev, sizeof(int) * table.size()) != hipSuccess ||
      hipMemcpy(o.dev, table.data(), sizeof(int) * table.size(), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    o.dev = nullptr; o.len = 0;
  }
  cache[key] = o;
  return o;
}

const int *bulk_tile_order(int ntr, long long tiles, long long *len) {
  const XcdOrder o = xcd_order(ntr, tiles);
  *len = o.dev ? o.len : 0;
  return o.dev;
}

// variant 0: fp64 MFMA kernel, 3: fp32-product MFMA kernel (mixed precision)
void launch_trailing_update_as(int variant, hipStream_t s, double *C, long long ldc, const double *P,
                               const double *Q, long long ldp, long long M, long long K, BulkTiming *timing, const float *P32,
                               const float *Q32, long long ld32) {
  if (timing) timing->flops = 0.;
  if (M <= 0 || K <= 0) return;
  GemmArgs g;
  g.C = C; g.ldc = ldc; g.A = P; g.lda = ldp; g.B = Q; g.ldb = ldp;
  g.A32 = P32; g.B32 = Q32; g.ld32 = ld32;
  g.M = M; g.N = M; g.K = K; g.tri = 1;
  g.ntr = (int)((M + GT - 1) / GT);
  g.ntc = g.ntr;
  launch_trailing_update_planned(variant, s, g, timing);
}

// fp64 only: C (M x M lower, the WHOLE trailing matrix of an outer step) -= P P^T in one launch whose first workgroups are
// the 128 x 128 tiles of the first head_cols tile columns (GemmArgs::head_cols); *head_tiles = how many *head_done will count
void launch_trailing_update_merged(hipStream_t s, double *C, long long ldc, const double *P, long long ldp, long long M, long long K,
                                   int head_cols, unsigned long long *head_done, long long *head_tiles, BulkTiming *timing) {
  if (timing) timing->flops = 0.;
  *head_tiles = 0;
  if (M <= 0 || K <= 0) return;
  GemmArgs g;
  g.C = C; g.ldc = ldc; g.A = P; g.lda = ldp; g.B = P; g.ldb = ldp;
  g.M = M; g.N = M; g.K = K; g.tri = 1;
  const int nth = (int)((M + GT - 1) / GT);
  if (head_cols > nth) head_cols = nth;
  g.head_cols = head_cols;
  g.head_done = head_done;
  for (int c = 0; c < head_cols; ++c) g.head_count += nth - c;
  g.ntr = nth - head_cols;  // the sub-triangle right of the head
  g.ntc = g.ntr;
  *head_tiles = g.head_count;
  launch_trailing_update_planned(0, s, g, timing);
}

static void launch_trailing_update_planned(int variant, hipStream_t s, GemmArgs &g, BulkTiming *timing) {
  const long long M = g.M, K = g.K;
  const long long tiles = count_tiles(g.ntr, g.ntc, 1);
  if (variant == 3) {
    long long wgs = tiles;
    if (tiles >= 1024) {  // (the same XCD-aware order as the fp64 kernel's whole tiles; all tiles are whole here)
      const XcdOrder o = xcd_order(g.ntr, tiles);
      if (o.dev) { g.order = o.dev; wgs = o.len; }
    }
    if (g.A32) hipLaunchKernelGGL(trailing_update_f32_kernel<true>, dim3((unsigned)wgs), dim3(GEMM_THREADS), 0, s, g);
    else hipLaunchKernelGGL(trailing_update_f32_kernel<false>, dim3((unsigned)wgs), dim3(GEMM_THREADS), 0, s, g);
    return;
  }
  // Tail split: a launch of T tiles runs floor(T / slots) full rounds of 128 x 128 workgroups (slots = 2 per
  // CU); the T mod slots tiles left over would occupy a fraction of the chip for a whole further round.
  // They go to trailing_update_tail_kernel as four 64 x 64 workgroups each.
  static int slots = 0;
  if (slots == 0) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    slots = 2 * cus;
  }
  long long full, rem;
  if (g.head_cols > 0) {
    // merged update: the head's tiles are the first round(s); behind them whole rounds of large tiles, the remainder as
    // 64 x 64 quadrants right after the head (the launch still ends with full rounds)
    if (tiles < 2LL * slots) { full = 0; rem = tiles; }
    else {
      full = (tiles / slots) * slots;
      rem = tiles - full;
      if (rem * 4 >= 3LL * slots) { full = tiles; rem = 0; }
    }
    if (timing && timing->e0) (void)hipEventRecord(timing->e0, s);
    g.small_first = full;
    g.small_count = (int)rem;
    long long big = full;
    if (full > 0) {
      const XcdOrder o = xcd_order(g.ntr, full);
      if (o.dev) { g.order = o.dev; big = o.len; }
    }
    hipLaunchKernelGGL(trailing_update_kernel, dim3((unsigned)(g.head_count + big + 4 * rem)), dim3(GEMM_THREADS), 0, s, g);
    if (timing && timing->e1) {
      (void)hipEventRecord(timing->e1, s);
      timing->flops = 2. * (double)K * lower_entries(M, g.ntr + g.head_cols, (long long)(g.ntr + g.head_cols) * (g.ntr + g.head_cols + 1) / 2);
    }
    return;
  }
  if (tiles < 2LL * slots) {
    // fewer than two rounds of large tiles: 64 x 64 workgroups throughout balance the CUs better (M = 4096, K = 512:
    // 0.196 ms instead of 0.272; with the store-only epilogue of the large tiles two rounds are enough: M = 6144 448 -> 424 us,
    // M = 7680 707 -> 670 us)
    full = 0; rem = tiles;
  } else {
    full = (tiles / slots) * slots;
    rem = tiles - full;
    if (rem * 4 >= 3LL * slots) { full = tiles; rem = 0; }  // an almost full round: leave it to the large tiles
  }
  if (full > 0) {
    if (timing && timing->e0) (void)hipEventRecord(timing->e0, s);
    g.small_first = full;
    g.small_count = (int)rem;
    long long big = full;
    {  // XCD-aware order of the whole tiles (halves the launch's fetch traffic: 5.8 -> 2.95 GB at M = 15872; fit 30.4 -> 30.05 ms)
      const XcdOrder o = xcd_order(g.ntr, full);
      if (o.dev) { g.order = o.dev; big = o.len; }
    }
    hipLaunchKernelGGL(trailing_update_kernel, dim3((unsigned)(big + 4 * rem)), dim3(GEMM_THREADS), 0, s, g);
    if (timing && timing->e1) {
      (void)hipEventRecord(timing->e1, s);
      timing->flops = 2. * (double)K * lower_entries(M, g.ntr, tiles);
    }
    return;
  }
  if (rem > 0) {
    GemmArgs h = g;
    h.tile_first = full;
    hipLaunchKernelGGL(trailing_update_tail_kernel, dim3((unsigned)(4 * rem)), dim3(GEMM_THREADS), 0, s, h);
  }
}

// C (M x N, lower tiles: C(0, 0) sits on the matrix diagonal) -= P Q^T with fp32-rounded panels on the fp32 MFMA path
// (trailing_update_f32_kernel), the result subtracted from the fp64 matrix: the next-block-column update U1 of the
// mixed-precision factorisation (agp_fit_create_mixed)
void launch_update_f32(hipStream_t s, double *C, long long ldc, const double *P, const double *Q, long long ldp, long long M,
                       long long N, long long K, const float *P32, const float *Q32, long long ld32) {
  if (M <= 0 || N <= 0 || K <= 0) return;
  GemmArgs g;
  g.C = C; g.ldc = ldc; g.A = P; g.lda = ldp; g.B = Q; g.ldb = ldp;
  g.A32 = P32; g.B32 = Q32; g.ld32 = ld32;
  g.M = M; g.N = N; g.K = K; g.tri = 1;
  g.ntr = (int)((M + GT - 1) / GT);
  g.ntc = (int)((N + GT - 1) / GT);
  if (g.ntc > g.ntr) g.ntc = g.ntr;
  const long long tiles = count_tiles(g.ntr, g.ntc, 1);
  if (tiles <= 0) return;
  if (g.A32) hipLaunchKernelGGL(trailing_update_f32_kernel<true>, dim3((unsigned)tiles), dim3(GEMM_THREADS), 0, s, g);
  else hipLaunchKernelGGL(trailing_update_f32_kernel<false>, dim3((unsigned)tiles), dim3(GEMM_THREADS), 0, s, g);
}

void launch_trailing_update(hipStream_t s, double *C, long long ldc, const double *P, const double *Q,
                            long long ldp, long long M, long long K, BulkTiming *timing) {
  launch_trailing_update_as(0, s, C, ldc, P, Q, ldp, M, K, timing);
}

}  // namespace agp
