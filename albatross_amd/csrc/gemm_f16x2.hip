// gemm_f16x2.hip - the bulk trailing update of the MIXED-precision factorisation from TWO fp16 planes per panel (round 6).
//
// BASELINE config 4 ("fp32 ... MFMA f32 Gram + mixed-precision Cholesky", examples/temperature_example/temperature_example.cc:34-85
// at N = 32768): agp_fit_create_mixed keeps the matrix, the panel chain and every accumulation between outer steps in fp64
// and forms the K <= 512 products of one outer step with fp32 ACCUMULATION.  Round 5 formed them from
// three bf16 planes (gemm_bf16x3.hip: six matrix instructions per 16 x 16 x 32 block).  fp16 has 11 significant bits against
// bf16's 8, so TWO planes carry 22 bits, and the whole product of two such numbers,
//     a b ~ (h1 + h2)(h1 + h2) = h2 h2 + (h2 h1 + h1 h2) + h1 h1,
// is FOUR instructions of the same rate (v_mfma_f32_16x16x32_f16) against six, on two thirds of the plane traffic; every
// partial product is exact in fp32 (22 bits), so all the error is in the split (<= 2^-22 |x|) and in the fp32 accumulation.
// Three products (AGP_F16X2_TERMS=3, without h2 h2) are 3 % faster in the fit and leave the diagonal of the Schur complement
// systematically too large (sum_k h2[i, k]^2 is never subtracted): log|K| of config 4 at N = 32768 off by +0.055 against -0.017
// with four (bf16 x 3: +0.027), config 3's kernel 0.15 against 0.06 (bf16 x 3: 0.14) - four is the default.
// What fp16 lacks is RANGE (2^-24 .. 65504), so every row is scaled by a power of two first:
//     |L[i, k]| <= sqrt(A[i, i])  (sum_k L[i, k]^2 = A[i, i]),    r_i = 2^(14 - e_i),  sqrt(A[i, i]) < 2^e_i,
// the planes hold h1 = rn16(r_i x), h2 = rn16(r_i x - h1) (|r_i x| < 2^14; the residual is a normal fp16 number down to
// |r_i x| ~ 2^-2 and has an ABSOLUTE error <= 2^-25 below that, i.e. 2^-39 of the row's scale), and the epilogue undoes the
// scales exactly: C[i, j] -= (double)acc / (r_i r_j).  The diagonal is read once, before the first panel
// (launch_f16x2_row_scales); rows whose diagonal is not a positive finite number get r = 1 (the factorisation reports them).
//
// Measured (profiles/r06/time_bf16x3.txt, time_mixed.txt; M = 15872 / 30720, K = 512): 175 / 189 TFLOP/s of fp32-equivalent
// products against 125 / 147 of the bf16 x 3 kernel; config 4 (N = 32768) 111.9 -> 94.1 ms.
//
//   convert_panel_f16x2    fp64 panel of one outer step -> two fp16 planes [plane][k / 32][row][k % 32] (as the bf16 planes)
//   trailing_update_f16x2_kernel   128 x 128 tile of C per workgroup, 64 x 64 per wave (16 accumulators), K in chunks of 32
//                          through one LDS stage (unpadded 64-B rows, 16-B pieces swizzled as in gemm_bf16x3.hip), 32 KB of LDS;
//                          the fp64 C of the tile comes in one accumulator column (16 doubles per lane) at a time, the first
//                          requested in front of the last chunk's matrix instructions, each next one before the previous is
//                          stored (+1 % on the kernel, -1 % on the fit against two columns loaded and stored in turn)
#include "common.h"
#include "gemm_tiles.h"

namespace agp {

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f32 __attribute__((ext_vector_type(4)));

namespace {
constexpr int MK = 32;                // k of one v_mfma_f32_16x16x32_f16
constexpr int SCALE_EXP = 14;         // |r_i x| < 2^14 (fp16 overflows at 65504 ~ 2^16)
// LDS image of one plane tile: 128 rows of CH halves (64 or 128 B, no padding), the 16-B pieces of a row permuted so that
// every lane group of a ds_read_b128 (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27}, ...) covers the 256-B bank row once and
// the eight lanes of a ds_write_b128 group cover 128 contiguous bytes.  CH = 32: piece kg of row r at slot
// kg ^ ((-(r >> 2)) & 3) (gemm_bf16x3.hip); CH = 64: at slot kg ^ ((r >> 1) & 7) (found by enumeration).
template <int CH>
__device__ __forceinline__ int sw_piece(int kg, int row) {
  return CH == 32 ? (kg ^ ((4 - ((row >> 2) & 3)) & 3)) : (kg ^ ((row >> 1) & 7));
}
int f16x2_chunk = 32;                 // k per LDS stage and per plane chunk (AGP_F16X2_CHUNK: 32 or 64)
}  // namespace

__global__ __launch_bounds__(256) void f16x2_row_scales_kernel(const double *__restrict__ A, long long lda, long long n, double *__restrict__ rs,
                                                               double *__restrict__ irs) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double d = A[i * (lda + 1)];
  double r = 1., ir = 1.;
  if (d > 0. && d < 1e300) {
    int e = 0;
    (void)frexp(sqrt(d), &e);  // sqrt(d) = m 2^e, 0.5 <= m < 1
    e = e < -200 ? -200 : (e > 200 ? 200 : e);
    r = ldexp(1., SCALE_EXP - e);
    ir = ldexp(1., e - SCALE_EXP);
  }
  rs[i] = r;
  irs[i] = ir;
}

void launch_f16x2_row_scales(hipStream_t s, const double *A, long long lda, long long n, double *rs, double *irs) {
  if (n <= 0) return;
  hipLaunchKernelGGL(f16x2_row_scales_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, A, lda, n, rs, irs);
}

// planes: 2 x [K / CH][rows_pad][CH] fp16; thread = (row, chunk)
template <int CH>
__global__ __launch_bounds__(256) void convert_panel_f16x2_kernel(const double *__restrict__ P, long long ldp, long long rows, long long rows_pad,
                                                                  const double *__restrict__ rs, unsigned short *__restrict__ planes,
                                                                  long long plane_stride) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long c = blockIdx.y;
  if (row >= rows_pad) return;
  const double r = row < rows ? rs[row] : 0.;
  unsigned short *dst = planes + c * rows_pad * CH + row * CH;
#pragma unroll
  for (int q = 0; q < CH / 8; ++q) {
    v8h h1, h2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double x = row < rows ? r * P[row + (c * CH + 8 * q + j) * ldp] : 0.;
      const _Float16 a = (_Float16)x;
      const _Float16 b = (_Float16)(x - (double)a);
      h1[j] = a;
      h2[j] = b;
    }
    *reinterpret_cast<v8h *>(dst + 8 * q) = h1;
    *reinterpret_cast<v8h *>(dst + plane_stride + 8 * q) = h2;
  }
}

long long f16x2_rows_pad(long long rows) { return (rows + GT - 1) / GT * GT + GT; }  // (+ one tile of zero rows: a tile may start anywhere below `rows`)
size_t f16x2_bytes(long long rows, long long K) { return sizeof(unsigned short) * 2 * (size_t)f16x2_rows_pad(rows) * (size_t)((K + 63) / 64 * 64); }
bool f16x2_depth_ok(long long K) { return K > 0 && K % f16x2_chunk == 0; }

// rs: the scales of the panel's rows (rs[0] = the scale of panel row 0)
void launch_convert_panel_f16x2(hipStream_t s, const double *P, long long ldp, long long rows, long long K, const double *rs,
                                unsigned short *planes) {
  if (rows <= 0 || !f16x2_depth_ok(K)) return;
  const long long rows_pad = f16x2_rows_pad(rows);
  const dim3 grid((unsigned)((rows_pad + 255) / 256), (unsigned)(K / f16x2_chunk));
  if (f16x2_chunk == 64) hipLaunchKernelGGL(convert_panel_f16x2_kernel<64>, grid, dim3(256), 0, s, P, ldp, rows, rows_pad, rs, planes, rows_pad * K);
  else hipLaunchKernelGGL(convert_panel_f16x2_kernel<32>, grid, dim3(256), 0, s, P, ldp, rows, rows_pad, rs, planes, rows_pad * K);
}

struct F16Args {
  double *C;
  long long ldc;
  const unsigned short *planes;  // of the panel both operands come from
  long long rows_pad, plane_stride;
  long long row_a, row_b;        // panel row of C's row 0 / of C's column 0
  const double *irs_a, *irs_b;   // 1 / r of C's rows / of C's columns
  long long M, N, K;
  int ntr, ntc;
  const int *order;              // XCD-aware tile order (gemm.hip: xcd_order) or nullptr
};

// Staging of one K chunk: piece q (16 B) of a plane tile = row q / (CH / 8), k group q % (CH / 8); a thread moves the pieces
// tid + 256 u of the two planes of both operands.
template <int TERMS, int CH>  // TERMS 3: h2 h1 + h1 h2 + h1 h1; 4: + h2 h2 (the term of weight 2^-22).  CH: k per LDS stage
__global__ __launch_bounds__(256, 2) void trailing_update_f16x2_kernel(F16Args g) {
  constexpr int PIECES = CH / 8;              // 16-B pieces per row
  constexpr int NP = GT * PIECES / 256;       // pieces per thread, plane and operand
  constexpr int PLANE = GT * CH;              // one plane of one operand, in halves
  __shared__ unsigned short lds[4 * PLANE];   // [operand A: h1 h2 | operand B: h1 h2][128 rows][CH], 16-B pieces swizzled
  int bi, bj;
  if (g.order) {
    const int packed = g.order[blockIdx.x];
    if (packed < 0) return;
    bi = packed >> 16;
    bj = packed & 0xffff;
  } else {  // tile (bi, bj) of the lower-triangular grid in column-major order
    long long id = blockIdx.x;
    bj = 0;
    while (bj < g.ntc && id >= g.ntr - bj) { id -= g.ntr - bj; ++bj; }
    if (bj >= g.ntc) return;
    bi = bj + (int)id;
  }
  const long long i0 = (long long)bi * GT, j0 = (long long)bj * GT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int ln = lane & 15, lg = lane >> 4;

  // A operand = the C-COLUMN panel (rows j0 ..), B operand = the C-ROW panel (rows i0 ..), as in gemm_tiles.h
  const unsigned short *srcA = g.planes + (g.row_b + j0) * CH + (long long)tid * 8, *srcB = g.planes + (g.row_a + i0) * CH + (long long)tid * 8;
  const long long chunk_stride = g.rows_pad * CH;
  // (named registers and macros: an array of them, or a lambda over them, lives in scratch memory)
  uint4 sa00, sa01, sa02, sa03, sa10, sa11, sa12, sa13, sb00, sb01, sb02, sb03, sb10, sb11, sb12, sb13;  // s<operand><plane><piece>
  const int d0 = (tid / PIECES) * CH + sw_piece<CH>(tid % PIECES, tid / PIECES) * 8;  // (row + 256 / PIECES: the same swizzle)
#define AGP_H_LOAD1(U, OFF)                                                                           \
  do {                                                                                                \
    sa0##U = *reinterpret_cast<const uint4 *>(srcA + (OFF) + (U) * 256 * 8);                          \
    sb0##U = *reinterpret_cast<const uint4 *>(srcB + (OFF) + (U) * 256 * 8);                          \
    sa1##U = *reinterpret_cast<const uint4 *>(srcA + g.plane_stride + (OFF) + (U) * 256 * 8);         \
    sb1##U = *reinterpret_cast<const uint4 *>(srcB + g.plane_stride + (OFF) + (U) * 256 * 8);         \
  } while (0)
#define AGP_H_STORE1(U)                                                                               \
  do {                                                                                                \
    *reinterpret_cast<uint4 *>(lds + 0 * PLANE + d0 + (U) * (256 / PIECES) * CH) = sa0##U;            \
    *reinterpret_cast<uint4 *>(lds + 1 * PLANE + d0 + (U) * (256 / PIECES) * CH) = sa1##U;            \
    *reinterpret_cast<uint4 *>(lds + 2 * PLANE + d0 + (U) * (256 / PIECES) * CH) = sb0##U;            \
    *reinterpret_cast<uint4 *>(lds + 3 * PLANE + d0 + (U) * (256 / PIECES) * CH) = sb1##U;            \
  } while (0)
#define AGP_H_LOAD(OFF)                                                   \
  do {                                                                    \
    const long long off_ = (OFF);                                         \
    AGP_H_LOAD1(0, off_); AGP_H_LOAD1(1, off_);                           \
    if constexpr (NP == 4) { AGP_H_LOAD1(2, off_); AGP_H_LOAD1(3, off_); } \
  } while (0)
#define AGP_H_STORE()                                                     \
  do {                                                                    \
    AGP_H_STORE1(0); AGP_H_STORE1(1);                                     \
    if constexpr (NP == 4) { AGP_H_STORE1(2); AGP_H_STORE1(3); }          \
  } while (0)

  v4f32 acc[4][4];  // [tj][ti]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = v4f32{0.f, 0.f, 0.f, 0.f};

  const long long nk = g.K / CH;
#define AGP_H_COMPUTE()                                                                                                                  \
  do {                                                                                                                                   \
    _Pragma("unroll") for (int t = 0; t < CH / MK; ++t) { /* the matrix instruction's k steps inside the stage */                         \
      v8h fa[2][4], fb[2][4];                                                                                                            \
      _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                                                      \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                                  \
          fa[p][q] = *reinterpret_cast<const v8h *>(lds + p * PLANE + (64 * wc + 16 * q + ln) * CH + 8 * sw_piece<CH>(4 * t + lg, ln));  \
          fb[p][q] = *reinterpret_cast<const v8h *>(lds + (2 + p) * PLANE + (64 * wr + 16 * q + ln) * CH + 8 * sw_piece<CH>(4 * t + lg, ln)); \
        }                                                                                                                                \
      _Pragma("unroll") for (int tj = 0; tj < 4; ++tj)                                                                                   \
        _Pragma("unroll") for (int ti = 0; ti < 4; ++ti) {                                                                               \
          v4f32 a = acc[tj][ti];                                                                                                         \
          if (TERMS == 4) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[1][tj], fb[1][ti], a, 0, 0, 0); /* h2 h2 */                       \
          a = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[1][tj], fb[0][ti], a, 0, 0, 0); /* h2 h1 (smallest terms first) */               \
          a = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][tj], fb[1][ti], a, 0, 0, 0); /* h1 h2 */                                      \
          a = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][tj], fb[0][ti], a, 0, 0, 0); /* h1 h1 */                                      \
          acc[tj][ti] = a;                                                                                                               \
        }                                                                                                                                \
    }                                                                                                                                    \
  } while (0)
  // C of this wave's quadrant (register r of accumulator (tj, ti): row 16 ti + ln of the quadrant, column 16 tj + 4 lg + r)
  const long long rbase = i0 + 64 * wr + ln, cbase_col = j0 + 64 * wc + 4 * lg;
  const bool interior = i0 + GT <= g.M && j0 + GT <= g.N;
  double *const cbase = g.C + rbase + cbase_col * g.ldc;
  AGP_H_LOAD(0);
  for (long long kc = 0; kc + 1 < nk; ++kc) {
    if (kc > 0) __syncthreads();  // every wave has read chunk kc - 1 out of the stage
#if defined(AGP_DIAG_F16_NOSTORE)  // diagnostic builds (wrong results): which part of the loop the time is in
    if (kc == 0) AGP_H_STORE();
#else
    AGP_H_STORE();
#endif
    __syncthreads();
#if defined(AGP_DIAG_F16_NOLOAD) || defined(AGP_DIAG_F16_NOSTORE)
    asm volatile("" ::: "memory");
#else
    AGP_H_LOAD((kc + 1) * chunk_stride);
#endif
    AGP_H_COMPUTE();
  }
  // the last chunk: nothing left to load for the loop - the first batch of C (one accumulator column: 16 doubles) is requested in
  // front of its matrix instructions instead, and the epilogue below keeps one batch in flight behind the one it stores
  if (nk > 1) __syncthreads();
#if defined(AGP_DIAG_F16_NOSTORE)
  if (nk == 1) AGP_H_STORE();
#else
  AGP_H_STORE();
#endif
  __syncthreads();
  double cv0[4][4], cv1[4][4];
  if (interior) {
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) cv0[ti][r] = __builtin_nontemporal_load(&cbase[16 * ti + (long long)r * g.ldc]);
  }
  AGP_H_COMPUTE();
#undef AGP_H_COMPUTE
#undef AGP_H_LOAD
#undef AGP_H_STORE
#undef AGP_H_LOAD1
#undef AGP_H_STORE1
  // C -= acc / (r_row r_col)
  double ir_row[4];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) ir_row[ti] = (rbase + 16 * ti < g.M) ? g.irs_a[rbase + 16 * ti] : 0.;
  if (interior) {
    // software pipeline over the four accumulator columns: column tj + 1 is requested before column tj is stored
#define AGP_H_EPI_LOAD(DST, TJ)                                                                                          \
  _Pragma("unroll") for (int ti = 0; ti < 4; ++ti)                                                                       \
    _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                                        \
      DST[ti][r] = __builtin_nontemporal_load(&cbase[16 * ti + (long long)(16 * (TJ) + r) * g.ldc])
#define AGP_H_EPI_STORE(SRC, TJ)                                                                                         \
  do {                                                                                                                   \
    double ic_[4];                                                                                                       \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) ic_[r] = g.irs_b[cbase_col + 16 * (TJ) + r];                           \
    _Pragma("unroll") for (int ti = 0; ti < 4; ++ti)                                                                     \
      _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                                      \
        __builtin_nontemporal_store(SRC[ti][r] - (double)acc[TJ][ti][r] * (ir_row[ti] * ic_[r]),                         \
                                    &cbase[16 * ti + (long long)(16 * (TJ) + r) * g.ldc]);                               \
  } while (0)
    AGP_H_EPI_LOAD(cv1, 1);
    AGP_H_EPI_STORE(cv0, 0);
    AGP_H_EPI_LOAD(cv0, 2);
    AGP_H_EPI_STORE(cv1, 1);
    AGP_H_EPI_LOAD(cv1, 3);
    AGP_H_EPI_STORE(cv0, 2);
    AGP_H_EPI_STORE(cv1, 3);
#undef AGP_H_EPI_LOAD
#undef AGP_H_EPI_STORE
    return;
  }
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {
      const long long row = rbase + 16 * ti;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long col = cbase_col + 16 * tj + r;
        if (row < g.M && col < g.N) {
          double *c = g.C + row + col * g.ldc;
          *c = *c - (double)acc[tj][ti][r] * (ir_row[ti] * g.irs_b[col]);
        }
      }
    }
}

// AGP_F16X2_LDS_PAD: extra dynamic LDS per workgroup (bytes), which sets how many workgroups share a CU next to the panel
// kernels of the chain stream (32 KB static; registers allow two)
static int f16x2_lds_pad = 8192;
static int f16x2_terms = 4;
void set_f16x2_kernel(int lds_pad, int terms, int chunk) {
  f16x2_lds_pad = lds_pad;
  f16x2_terms = terms == 3 ? 3 : 4;
  f16x2_chunk = chunk == 64 ? 64 : 32;
}

// C (M x N, lower tiles, C(0, 0) on the matrix diagonal) -= P[row_a ..] P[row_b ..]^T from the fp16 planes of ONE panel
// (launch_convert_panel_f16x2).  irs: 1 / r of the PANEL's rows (irs[0] belongs to panel row 0).  order / order_len: the
// XCD-aware tile order of gemm.hip (nullptr: column-major tiles).
void launch_update_f16x2(hipStream_t s, double *C, long long ldc, const unsigned short *planes, long long panel_rows, long long row_a,
                         long long row_b, const double *irs, long long M, long long N, long long K, const int *order, long long order_len) {
  if (M <= 0 || N <= 0 || !f16x2_depth_ok(K)) return;
  F16Args g;
  g.C = C; g.ldc = ldc; g.planes = planes;
  g.rows_pad = f16x2_rows_pad(panel_rows);
  g.plane_stride = g.rows_pad * K;
  g.row_a = row_a; g.row_b = row_b;
  g.irs_a = irs + row_a; g.irs_b = irs + row_b;
  g.M = M; g.N = N; g.K = K;
  g.ntr = (int)((M + GT - 1) / GT);
  g.ntc = (int)((N + GT - 1) / GT);
  if (g.ntc > g.ntr) g.ntc = g.ntr;
  g.order = order;
  long long tiles = 0;
  for (int bj = 0; bj < g.ntc; ++bj) tiles += g.ntr - bj;
  const long long wgs = order ? order_len : tiles;
  if (wgs <= 0) return;
  const dim3 grid((unsigned)wgs), block(256);
  const size_t pad = (size_t)f16x2_lds_pad;
  if (f16x2_chunk == 64) {
    if (f16x2_terms == 4) hipLaunchKernelGGL((trailing_update_f16x2_kernel<4, 64>), grid, block, pad, s, g);
    else hipLaunchKernelGGL((trailing_update_f16x2_kernel<3, 64>), grid, block, pad, s, g);
  } else {
    if (f16x2_terms == 4) hipLaunchKernelGGL((trailing_update_f16x2_kernel<4, 32>), grid, block, pad, s, g);
    else hipLaunchKernelGGL((trailing_update_f16x2_kernel<3, 32>), grid, block, pad, s, g);
  }
}

}  // namespace agp
