// The tile bodies of the update kernels, C -= A B^T on the fp64 MFMA (128 x 128 and 64 x 64 tiles per workgroup), and
// their argument block: shared by gemm.hip (the update launches) and chol.hip (the trailing-update workgroups of the
// panel step kernel).
#pragma once
#include "common.h"
#include "mfma_f64.h"

namespace agp {

constexpr int GT = 128;        // C tile edge
constexpr int GK = 16;         // K chunk
constexpr int GLD = GT + 16;   // LDS row pitch in doubles
constexpr int GEMM_THREADS = 256;

struct GemmArgs {
  double *C;
  long long ldc;
  const double *A;  // operand indexed by C row i
  long long lda;
  const double *B;  // operand indexed by C col j
  long long ldb;
  long long M, N, K;
  int tri;  // skip tiles strictly above the diagonal of C
  int ntr, ntc;
  int a_kmajor = 0;  // gemm64_nt_sub_bk_kernel only: the A operand is stored transposed too
  // batched launches (blockIdx.y = batch entry): element offsets per entry; 0 = not batched
  long long batch_C = 0, batch_A = 0, batch_B = 0;
  long long tile_first = 0;  // first tile (in column-major tile order) of this launch: split bulk updates
  // "staircase" launches of the row-block-sharded fit (shard.h): C = the stacked local row blocks of one rank, the row
  // tile bi belongs to local block st_lb0 + bi / st_tpb = global block gi (snake deal over st_world ranks) and owns the
  // tile columns up to its own diagonal tile: bj <= gi * st_tpb - st_c0t + bi % st_tpb.  stair == 0: off.
  int stair = 0, st_world = 1, st_rank = 0, st_tpb = 4;
  long long st_lb0 = 0, st_c0t = 0;
  // gemm_nt_ext_kernel only: C = Cin - A B^T (Cin may be another matrix, leading dimension ldcin), or - assign != 0 -
  // C = + A B^T without reading C
  const double *Cin = nullptr;
  long long ldcin = 0;
  int assign = 0;
  // trailing_update_kernel only: its first 4 * small_count workgroups compute the 128-tiles small_first .. (column-major
  // lower-tile order) as 64 x 64 quadrants, the others the tiles 0 .. small_first - 1 whole
  long long small_first = 0;
  int small_count = 0;
  // ... and, when set, the whole tiles in the order of this table (entry = bi << 16 | bj, -1: no tile): workgroup i runs on
  // XCD i % 8, and the table gives every XCD compact 8 x 8 blocks of tiles (launch_trailing_update_as)
  const int *order = nullptr;
  // trailing_update_f32_kernel only: fp32 copies of the two operands (element (row, k) at X32[row + k * ld32]; made once per
  // panel by launch_convert_panel_f32) - nullptr: the kernel rounds the fp64 operands itself while it stages them
  const float *A32 = nullptr, *B32 = nullptr;
  long long ld32 = 0;
  // trailing_update_kernel only, the MERGED update of factor_lower (chol.hip): C is the whole trailing matrix of an outer
  // step and its first head_cols tile columns - the NEXT block column, which the panel chain needs first - are the first
  // head_count workgroups of the launch (column by column, top to bottom, 128 x 128 tiles); each counts itself in *head_done
  // (RELEASE, device scope) when its tile is in memory, and the chain stream's gate kernel lets the next panel phase start
  // at head_count - while the rest of the launch (tiles with bj >= head_cols; ntr / small_* / order describe THAT
  // sub-triangle, in its own frame) is still running.  head_cols == 0: off.
  int head_cols = 0;
  long long head_count = 0;
  unsigned long long *head_done = nullptr;
};


// Load this thread's 8 doubles of a 128 x 16 operand chunk.
//   !KMAJOR: element (row, k) at P[row + k * ld]   (panel stored like the matrix)
//    KMAJOR: element (row, k) at P[k + row * ld]   (transposed access)
template <bool KMAJOR>
__device__ __forceinline__ void load_chunk(const double *__restrict__ P, long long ld, long long row0,
                                           long long nrows, long long k0, long long K, bool vec_ok,
                                           double (&r)[8]) {
  const int t = threadIdx.x;
  // wave-uniform: interior tile and full chunk -> unguarded 16-B loads
  const bool fast = vec_ok && (row0 + GT <= nrows) && (k0 + GK <= K);
  if (!KMAJOR) {
    // thread t: row pairs 2 (t & 15) + 32 j (j = 0 .. 3) of k-row t >> 4 - the sixteen lanes of a k-row move 256
    // contiguous bytes per instruction, and so do their ds_write_b128 (store_chunk): eight lanes = 128 B = every bank
    // once (a lane's own 64 contiguous bytes put lanes l and l + 2 on the same banks: 4-way)
    const int kk = t >> 4, seg = (t & 15) * 2;
    const long long row = row0 + seg, k = k0 + kk;
    const double *p = P + row + k * ld;
    if (fast) {
      const double2 a = *reinterpret_cast<const double2 *>(p);
      const double2 b = *reinterpret_cast<const double2 *>(p + 32);
      const double2 c = *reinterpret_cast<const double2 *>(p + 64);
      const double2 d = *reinterpret_cast<const double2 *>(p + 96);
      r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y;
      r[4] = c.x; r[5] = c.y; r[6] = d.x; r[7] = d.y;
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int off = 32 * (q >> 1) + (q & 1);
        r[q] = (k < K && row + off < nrows) ? p[off] : 0.;
      }
    }
  } else {
    const int j = t >> 1, kh = (t & 1) * 8;
    const long long row = row0 + j, k = k0 + kh;
    const double *p = P + k + row * ld;
    if (fast) {
      const double2 a = *reinterpret_cast<const double2 *>(p);
      const double2 b = *reinterpret_cast<const double2 *>(p + 2);
      const double2 c = *reinterpret_cast<const double2 *>(p + 4);
      const double2 d = *reinterpret_cast<const double2 *>(p + 6);
      r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y;
      r[4] = c.x; r[5] = c.y; r[6] = d.x; r[7] = d.y;
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) r[q] = (row < nrows && k + q < K) ? p[q] : 0.;
    }
  }
}

// the same for an interior tile, a full chunk and 16-B aligned operands: no guards
template <bool KMAJOR>
__device__ __forceinline__ void load_chunk_interior(const double *__restrict__ P, long long ld, long long row0, long long k0,
                                                    double (&r)[8]) {
  const int t = threadIdx.x;
  const double *p = KMAJOR ? P + (k0 + (t & 1) * 8) + (row0 + (t >> 1)) * ld : P + (row0 + (t & 15) * 2) + (k0 + (t >> 4)) * ld;
  constexpr int step = KMAJOR ? 2 : 32;  // (!KMAJOR: row pairs 2 (t & 15) + 32 j, see load_chunk)
  const double2 a = *reinterpret_cast<const double2 *>(p);
  const double2 b = *reinterpret_cast<const double2 *>(p + step);
  const double2 c = *reinterpret_cast<const double2 *>(p + 2 * step);
  const double2 d = *reinterpret_cast<const double2 *>(p + 3 * step);
  r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y;
  r[4] = c.x; r[5] = c.y; r[6] = d.x; r[7] = d.y;
}

template <bool KMAJOR, bool NEGATE>
__device__ __forceinline__ void store_chunk(double *__restrict__ Ls, const double (&r)[8]) {
  const int t = threadIdx.x;
  if (!KMAJOR) {
    const int kk = t >> 4, seg = (t & 15) * 2;
    double2 *dst = reinterpret_cast<double2 *>(Ls + kk * GLD + seg);  // (double2 index 16 j = row pair + 32 j)
    if (NEGATE) {
      dst[0] = make_double2(-r[0], -r[1]);
      dst[16] = make_double2(-r[2], -r[3]);
      dst[32] = make_double2(-r[4], -r[5]);
      dst[48] = make_double2(-r[6], -r[7]);
    } else {
      dst[0] = make_double2(r[0], r[1]);
      dst[16] = make_double2(r[2], r[3]);
      dst[32] = make_double2(r[4], r[5]);
      dst[48] = make_double2(r[6], r[7]);
    }
  } else {
    const int j = t >> 1, kh = (t & 1) * 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) Ls[(kh + q) * GLD + j] = NEGATE ? -r[q] : r[q];
  }
}

// One 128 x 128 tile (bi, bj) of C by one workgroup of 256 threads; lds: 2 * 2 * GK * GLD doubles.
template <bool A_KMAJOR, bool B_KMAJOR, bool EXT = false>
__device__ __forceinline__ void gemm_nt_sub_tile(const GemmArgs &g, const int bi, const int bj, double *lds) {
  const long long i0 = (long long)bi * GT, j0 = (long long)bj * GT;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int ln = lane & 15, lg = lane >> 4;

  const bool a_vec = (((reinterpret_cast<uintptr_t>(g.A)) & 15) == 0) && ((g.lda & 1) == 0);
  const bool b_vec = (((reinterpret_cast<uintptr_t>(g.B)) & 15) == 0) && ((g.ldb & 1) == 0);

  v4d acc[4][4];  // [tj][ti]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = v4zero();

  double ra[8], rb[8];
  const long long nk = (g.K + GK - 1) / GK;
  load_chunk<A_KMAJOR>(g.A, g.lda, i0, g.M, 0, g.K, a_vec, ra);
  load_chunk<B_KMAJOR>(g.B, g.ldb, j0, g.N, 0, g.K, b_vec, rb);
  store_chunk<A_KMAJOR, false>(lds, ra);
  store_chunk<B_KMAJOR, true>(lds + GK * GLD, rb);
  __syncthreads();

  for (long long kc = 0; kc < nk; ++kc) {
    const int cur = (int)(kc & 1);
    const double *As = lds + cur * (2 * GK * GLD);
    const double *Bs = As + GK * GLD;
    const bool more = kc + 1 < nk;
    if (more) {
      load_chunk<A_KMAJOR>(g.A, g.lda, i0, g.M, (kc + 1) * GK, g.K, a_vec, ra);
      load_chunk<B_KMAJOR>(g.B, g.ldb, j0, g.N, (kc + 1) * GK, g.K, b_vec, rb);
    }
#pragma unroll
    for (int s = 0; s < GK / 4; ++s) {
      double fa[4], fb[4];
      const int krow = (4 * s + lg) * GLD;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t] = Bs[krow + 64 * wc + 16 * t + ln];  // MFMA A operand: C-column panel (negated)
        fb[t] = As[krow + 64 * wr + 16 * t + ln];  // MFMA B operand: C-row panel
      }
      if (s == GK / 4 - 1 && more) {  // (the next chunk's LDS stores in front of the last k step's MFMAs: see gemm_nt_sub_tile_cpf)
        double *An = lds + (cur ^ 1) * (2 * GK * GLD);
        store_chunk<A_KMAJOR, false>(An, ra);
        store_chunk<B_KMAJOR, true>(An + GK * GLD, rb);
      }
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) acc[tj][ti] = mfma16(fa[tj], fb[ti], acc[tj][ti]);
    }
    __syncthreads();
  }

  // ---- epilogue: C += acc (acc already holds -A B^T) ----
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      const long long row = i0 + 64 * wr + 16 * ti + ln;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long col = j0 + 64 * wc + 16 * tj + lg + 4 * r;
        if (row < g.M && col < g.N) {
          double *c = g.C + row + col * g.ldc;
          if (EXT) {
            if (g.assign) *c = -acc[tj][ti][r];
            else *c = g.Cin[row + col * g.ldcin] + acc[tj][ti][r];
          } else {
            *c = *c + acc[tj][ti][r];
          }
        }
      }
    }
}

// The same tile for INTERIOR tiles of products at least 8 chunks deep, without the read-modify-write of C behind the K loop.
// gemm_nt_sub_tile reads, adds and writes C after its loop - where every workgroup of a round does it at the same
// moment: a burst of 2 x 128 KB per workgroup that the matrix pipe waits for (45 us per round of 512 tiles against
// 131 us of MFMA work at K = 512: the rate of the bulk update was a function of K, 47.6 TFLOP/s at 512, 60 at 2048;
// profiles/r04/bulk_update_vs_k.txt).  Here C is added into the accumulators in EIGHT parts while the loop runs: part p
// (accumulators acc[p >> 1][2 (p & 1) + 0 / 1], 8 values per lane) is loaded at the first chunk of the p-th eighth of
// the K loop and added after its last chunk, an eighth of the loop later - nothing waits for it -, and the epilogue only
// stores.  EXT: C = Cin - A B^T / C = A B^T as in gemm_nt_sub_tile.
template <bool A_KMAJOR, bool B_KMAJOR, bool EXT = false>
__device__ __forceinline__ void gemm_nt_sub_tile_cpf(const GemmArgs &g, const int bi, const int bj, double *lds) {
  const long long i0 = (long long)bi * GT, j0 = (long long)bj * GT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int ln = lane & 15, lg = lane >> 4;

  v4d acc[4][4];  // [tj][ti]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = v4zero();

  double ra[8], rb[8];
  const long long nk = g.K / GK, nq = nk / 8;
  load_chunk_interior<A_KMAJOR>(g.A, g.lda, i0, 0, ra);
  load_chunk_interior<B_KMAJOR>(g.B, g.ldb, j0, 0, rb);
  store_chunk<A_KMAJOR, false>(lds, ra);
  store_chunk<B_KMAJOR, true>(lds + GK * GLD, rb);
  __syncthreads();

  const long long coff = (i0 + 64 * wr + ln) + (j0 + 64 * wc + lg) * g.ldc;
  double *const cbase = g.C + coff;
  const bool add_c = !EXT || !g.assign;
  const double *const cin = (EXT && add_c) ? g.Cin + (i0 + 64 * wr + ln) + (j0 + 64 * wc + lg) * g.ldcin : cbase;
  const long long ldcin = (EXT && add_c) ? g.ldcin : g.ldc;
  long long kc = 0;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    v4d ct[2];
    const long long k_end = (p == 7) ? nk : (p + 1) * nq;
    bool first = add_c;
    for (; kc < k_end; ++kc) {
      const int cur = (int)(kc & 1);
      const double *As = lds + cur * (2 * GK * GLD);
      const double *Bs = As + GK * GLD;
      const bool more = kc + 1 < nk;
      if (more) {
        load_chunk_interior<A_KMAJOR>(g.A, g.lda, i0, (kc + 1) * GK, ra);
        load_chunk_interior<B_KMAJOR>(g.B, g.ldb, j0, (kc + 1) * GK, rb);
      }
      if (first) {  // part p of C: columns 16 (p >> 1) + lg + 4 r of this wave's 64, rows 16 (2 (p & 1) + t) + ln
        first = false;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) ct[t][r] = __builtin_nontemporal_load(&cin[16 * (2 * (p & 1) + t) + (long long)(16 * (p >> 1) + 4 * r) * ldcin]);
      }
#pragma unroll
      for (int s = 0; s < GK / 4; ++s) {
        double fa[4], fb[4];
        const int krow = (4 * s + lg) * GLD;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          fa[t] = Bs[krow + 64 * wc + 16 * t + ln];
          fb[t] = As[krow + 64 * wr + 16 * t + ln];
        }
        if (s == GK / 4 - 1 && more) {
          // the next chunk into the OTHER stage in front of the last sixteen MFMAs (nobody reads that stage before the
          // barrier below; its loads were issued three k steps ago): the stores used to follow the last MFMA - eight
          // ds_write_b128 and their vmcnt waits with an idle matrix pipe, once per chunk
          double *An = lds + (cur ^ 1) * (2 * GK * GLD);
          store_chunk<A_KMAJOR, false>(An, ra);
          store_chunk<B_KMAJOR, true>(An + GK * GLD, rb);
        }
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
          for (int ti = 0; ti < 4; ++ti) acc[tj][ti] = mfma16(fa[tj], fb[ti], acc[tj][ti]);
      }
      __syncthreads();
    }
    if (add_c) {
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[p >> 1][2 * (p & 1) + t] += ct[t];
    }
  }
  const double sgn = add_c ? 1. : -1.;  // (the accumulators hold -A B^T)
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(sgn * acc[tj][ti][r], &cbase[16 * ti + (long long)(16 * tj + 4 * r) * g.ldc]);
}

// interior tile of a product deep enough for the eight parts?
__device__ __forceinline__ bool tile_takes_cpf(const GemmArgs &g, const int bi, const int bj) {
  const bool aligned = ((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.B)) & 15) == 0 && ((g.lda | g.ldb) & 1) == 0;
  return aligned && g.K >= 8 * GK && g.K % GK == 0 && (long long)(bi + 1) * GT <= g.M && (long long)(bj + 1) * GT <= g.N;
}

// ---------------------------------------------------------------------------
// 64 x 64-tile variant for launches too small to fill the chip with 128 x 128
// tiles (the next-panel and inner updates of the serial panel chain): four
// times as many, four times shorter workgroups, 3-4 of them per CU.
// Panel-major operands only.
// ---------------------------------------------------------------------------
constexpr int ST = 64;         // small tile edge
constexpr int SLD = ST + 16;   // LDS pitch (pitch mod 32 == 16: conflict-free fragment reads)

__device__ __forceinline__ void load_chunk64(const double *__restrict__ P, long long ld, long long row0,
                                             long long nrows, long long k0, long long K, bool vec_ok,
                                             double (&r)[4]) {
  const int t = threadIdx.x;
  // (row pairs 2 (t & 15) and 2 (t & 15) + 32: contiguous across lanes, as load_chunk)
  const int kk = t >> 4, seg = (t & 15) * 2;
  const long long row = row0 + seg, k = k0 + kk;
  const double *p = P + row + k * ld;
  const bool fast = vec_ok && (row0 + ST <= nrows) && (k0 + GK <= K);
  if (fast) {
    const double2 a = *reinterpret_cast<const double2 *>(p);
    const double2 b = *reinterpret_cast<const double2 *>(p + 32);
    r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y;
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int off = 32 * (q >> 1) + (q & 1);
      r[q] = (k < K && row + off < nrows) ? p[off] : 0.;
    }
  }
}

template <bool NEGATE>
__device__ __forceinline__ void store_chunk64(double *__restrict__ Ls, const double (&r)[4]) {
  const int t = threadIdx.x;
  const int kk = t >> 4, seg = (t & 15) * 2;
  double2 *dst = reinterpret_cast<double2 *>(Ls + kk * SLD + seg);
  dst[0] = NEGATE ? make_double2(-r[0], -r[1]) : make_double2(r[0], r[1]);
  dst[16] = NEGATE ? make_double2(-r[2], -r[3]) : make_double2(r[2], r[3]);
}

// transposed operand storage (element (row, k) at P[k + row * ld]): thread t holds 4 consecutive k of row t >> 2
__device__ __forceinline__ void load_chunk64_kmajor(const double *__restrict__ P, long long ld, long long row0,
                                                    long long nrows, long long k0, long long K, bool vec_ok,
                                                    double (&r)[4]) {
  const int t = threadIdx.x;
  const int j = t >> 2, kq = (t & 3) * 4;
  const long long row = row0 + j, k = k0 + kq;
  const double *p = P + k + row * ld;
  if (vec_ok && row < nrows && k + 4 <= K) {
    const double2 a = *reinterpret_cast<const double2 *>(p);
    const double2 b = *reinterpret_cast<const double2 *>(p + 2);
    r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y;
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = (row < nrows && k + q < K) ? p[q] : 0.;
  }
}

template <bool NEGATE>
__device__ __forceinline__ void store_chunk64_kmajor(double *__restrict__ Ls, const double (&r)[4]) {
  const int t = threadIdx.x;
  const int j = t >> 2, kq = (t & 3) * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) Ls[(kq + q) * SLD + j] = NEGATE ? -r[q] : r[q];
}

template <bool B_KMAJOR = false, bool A_KMAJOR = false>
__device__ __forceinline__ void gemm64_body(const GemmArgs &g, const long long i0, const long long j0, double *lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int ln = lane & 15, lg = lane >> 4;
  const bool a_vec = (((reinterpret_cast<uintptr_t>(g.A)) & 15) == 0) && ((g.lda & 1) == 0);
  const bool b_vec = (((reinterpret_cast<uintptr_t>(g.B)) & 15) == 0) && ((g.ldb & 1) == 0);

  // These launches sit on the serial panel chain and are latency-bound (a 64 x 64 x K tile is a
  // few microseconds of MFMA work behind K / 16 global-load round trips), so the operand stream
  // runs TWO chunks ahead of the MFMAs (two register stages + the two LDS buffers) and the
  // accumulators start from C, loaded while the first chunks are in flight: no read-modify-write
  // at the end.
  double ra[2][4], rb[2][4];
  const long long nk = (g.K + GK - 1) / GK;
  if (A_KMAJOR) load_chunk64_kmajor(g.A, g.lda, i0, g.M, 0, g.K, a_vec, ra[0]);
  else load_chunk64(g.A, g.lda, i0, g.M, 0, g.K, a_vec, ra[0]);
  if (B_KMAJOR) load_chunk64_kmajor(g.B, g.ldb, j0, g.N, 0, g.K, b_vec, rb[0]);
  else load_chunk64(g.B, g.ldb, j0, g.N, 0, g.K, b_vec, rb[0]);
  if (nk > 1) {
    if (A_KMAJOR) load_chunk64_kmajor(g.A, g.lda, i0, g.M, GK, g.K, a_vec, ra[1]);
    else load_chunk64(g.A, g.lda, i0, g.M, GK, g.K, a_vec, ra[1]);
    if (B_KMAJOR) load_chunk64_kmajor(g.B, g.ldb, j0, g.N, GK, g.K, b_vec, rb[1]);
    else load_chunk64(g.B, g.ldb, j0, g.N, GK, g.K, b_vec, rb[1]);
  }
  v4d acc[2][2];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      const long long row = i0 + 32 * wr + 16 * ti + ln;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long col = j0 + 32 * wc + 16 * tj + lg + 4 * r;
        acc[tj][ti][r] = (row < g.M && col < g.N) ? g.C[row + col * g.ldc] : 0.;
      }
    }
  if (A_KMAJOR) store_chunk64_kmajor<false>(lds, ra[0]);
  else store_chunk64<false>(lds, ra[0]);
  if (B_KMAJOR) store_chunk64_kmajor<true>(lds + GK * SLD, rb[0]);
  else store_chunk64<true>(lds + GK * SLD, rb[0]);
  __syncthreads();
  for (long long kc = 0; kc < nk; kc += 2) {
    // two chunks per trip so that the register stages are compile-time indices
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const long long k = kc + half;
      if (k >= nk) break;
      const double *As = lds + half * (2 * GK * SLD);
      const double *Bs = As + GK * SLD;
      if (k + 2 < nk) {  // stage `half` was stored to LDS one trip ago: refill it with chunk k + 2
        if (A_KMAJOR) load_chunk64_kmajor(g.A, g.lda, i0, g.M, (k + 2) * GK, g.K, a_vec, ra[half]);
        else load_chunk64(g.A, g.lda, i0, g.M, (k + 2) * GK, g.K, a_vec, ra[half]);
        if (B_KMAJOR) load_chunk64_kmajor(g.B, g.ldb, j0, g.N, (k + 2) * GK, g.K, b_vec, rb[half]);
        else load_chunk64(g.B, g.ldb, j0, g.N, (k + 2) * GK, g.K, b_vec, rb[half]);
      }
#pragma unroll
      for (int s = 0; s < GK / 4; ++s) {
        const int krow = (4 * s + lg) * SLD;
        double fa[2], fb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          fa[t] = Bs[krow + 32 * wc + 16 * t + ln];
          fb[t] = As[krow + 32 * wr + 16 * t + ln];
        }
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
          for (int ti = 0; ti < 2; ++ti) acc[tj][ti] = mfma16(fa[tj], fb[ti], acc[tj][ti]);
      }
      // (the stores IN FRONT of the MFMAs - their data has been in registers for a trip - measured 1 % slower at
      // N = 8192 / 16384: behind the MFMAs they overlap with the tail of the matrix pipe)
      if (k + 1 < nk) {  // chunk k + 1 (register stage half ^ 1, loaded a trip ago) -> the other LDS buffer
        double *An = lds + (half ^ 1) * (2 * GK * SLD);
        if (A_KMAJOR) store_chunk64_kmajor<false>(An, ra[half ^ 1]);
        else store_chunk64<false>(An, ra[half ^ 1]);
        if (B_KMAJOR) store_chunk64_kmajor<true>(An + GK * SLD, rb[half ^ 1]);
        else store_chunk64<true>(An + GK * SLD, rb[half ^ 1]);
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      const long long row = i0 + 32 * wr + 16 * ti + ln;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long col = j0 + 32 * wc + 16 * tj + lg + 4 * r;
        if (row < g.M && col < g.N) g.C[row + col * g.ldc] = acc[tj][ti][r];
      }
    }
}


// ---------------------------------------------------------------------------
// 32 x 32-tile variant for the SMALLEST launches of the panel chain - the next-block-column update of a sharded fit's
// rank (1536 x 512 x 512), the update of a 512 x 512 diagonal block - where even 64 x 64 tiles leave most of the chip
// idle and every wave carries 512 MFMAs one behind the other: four times as many workgroups, a 16 x 16 tile (two
// accumulators over alternating k-steps) per wave.  Panel-major operands only.
// ---------------------------------------------------------------------------
constexpr int TT = 32;          // tiny tile edge
constexpr int TLD = TT + 16;    // LDS pitch (pitch mod 32 == 16: conflict-free fragment reads)

__device__ __forceinline__ void load_chunk32(const double *__restrict__ P, long long ld, long long row0, long long nrows,
                                             long long k0, long long K, bool vec_ok, double (&r)[2]) {
  const int t = threadIdx.x;
  const int kk = t >> 4, seg = (t & 15) * 2;
  const long long row = row0 + seg, k = k0 + kk;
  const double *p = P + row + k * ld;
  if (vec_ok && row0 + TT <= nrows && k0 + GK <= K) {
    const double2 a = *reinterpret_cast<const double2 *>(p);
    r[0] = a.x; r[1] = a.y;
  } else {
    r[0] = (k < K && row < nrows) ? p[0] : 0.;
    r[1] = (k < K && row + 1 < nrows) ? p[1] : 0.;
  }
}

template <bool NEGATE>
__device__ __forceinline__ void store_chunk32(double *__restrict__ Ls, const double (&r)[2]) {
  const int t = threadIdx.x;
  const int kk = t >> 4, seg = (t & 15) * 2;
  *reinterpret_cast<double2 *>(Ls + kk * TLD + seg) = NEGATE ? make_double2(-r[0], -r[1]) : make_double2(r[0], r[1]);
}

// lds: 2 * 2 * GK * TLD doubles
__device__ __forceinline__ void gemm32_body(const GemmArgs &g, const long long i0, const long long j0, double *lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int ln = lane & 15, lg = lane >> 4;
  const bool a_vec = (((reinterpret_cast<uintptr_t>(g.A)) & 15) == 0) && ((g.lda & 1) == 0);
  const bool b_vec = (((reinterpret_cast<uintptr_t>(g.B)) & 15) == 0) && ((g.ldb & 1) == 0);
  // (the operand stream runs two chunks ahead of the MFMAs, the accumulator starts from C: see gemm64_body)
  double ra[2][2], rb[2][2];
  const long long nk = (g.K + GK - 1) / GK;
  load_chunk32(g.A, g.lda, i0, g.M, 0, g.K, a_vec, ra[0]);
  load_chunk32(g.B, g.ldb, j0, g.N, 0, g.K, b_vec, rb[0]);
  if (nk > 1) {
    load_chunk32(g.A, g.lda, i0, g.M, GK, g.K, a_vec, ra[1]);
    load_chunk32(g.B, g.ldb, j0, g.N, GK, g.K, b_vec, rb[1]);
  }
  v4d acc0, acc1 = v4zero();
  const long long row = i0 + 16 * wr + ln;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long long col = j0 + 16 * wc + lg + 4 * r;
    acc0[r] = (row < g.M && col < g.N) ? g.C[row + col * g.ldc] : 0.;
  }
  store_chunk32<false>(lds, ra[0]);
  store_chunk32<true>(lds + GK * TLD, rb[0]);
  __syncthreads();
  for (long long kc = 0; kc < nk; kc += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const long long k = kc + half;
      if (k >= nk) break;
      const double *As = lds + half * (2 * GK * TLD);
      const double *Bs = As + GK * TLD;
      if (k + 2 < nk) {
        load_chunk32(g.A, g.lda, i0, g.M, (k + 2) * GK, g.K, a_vec, ra[half]);
        load_chunk32(g.B, g.ldb, j0, g.N, (k + 2) * GK, g.K, b_vec, rb[half]);
      }
#pragma unroll
      for (int s2 = 0; s2 < GK / 4; s2 += 2) {
        const int k0r = (4 * s2 + lg) * TLD, k1r = (4 * (s2 + 1) + lg) * TLD;
        acc0 = mfma16(Bs[k0r + 16 * wc + ln], As[k0r + 16 * wr + ln], acc0);
        acc1 = mfma16(Bs[k1r + 16 * wc + ln], As[k1r + 16 * wr + ln], acc1);
      }
      if (k + 1 < nk) {
        double *An = lds + (half ^ 1) * (2 * GK * TLD);
        store_chunk32<false>(An, ra[half ^ 1]);
        store_chunk32<true>(An + GK * TLD, rb[half ^ 1]);
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const long long col = j0 + 16 * wc + lg + 4 * r;
    if (row < g.M && col < g.N) g.C[row + col * g.ldc] = acc0[r] + acc1[r];
  }
}

}  // namespace agp
