// gemm_bf16x3.hip - the bulk trailing update of the MIXED-precision factorisation on the BF16 matrix pipe (round 5).
//
// BASELINE config 4 ("fp32 ... MFMA f32 Gram + mixed-precision Cholesky",
// examples/temperature_example/temperature_example.cc:34-85 at N = 32768): agp_fit_create_mixed keeps the matrix, the
// panel chain and every accumulation between outer steps in fp64 and forms the K <= 512 products of one outer step at
// fp32 accuracy.  Rounds 1-4 did that with v_mfma_f32_16x16x4_f32 (157 TFLOP/s peak, 103 reached).
// gfx950 has no faster fp32 matrix instruction, but its BF16 pipe is 16x that rate, and an fp32 number IS three
// bf16 numbers: x = hi + mid + lo with 8 significant bits each (24 = the fp32 significand).  A product of two such
// numbers to fp32 accuracy needs the six partial products whose weight is above 2^-24,
//     a b ~ hi hi + (hi mid + mid hi) + (hi lo + lo hi + mid mid),
// all accumulated in the fp32 accumulators of v_mfma_f32_16x16x32_bf16: six instructions of 16 cycles for a
// 16 x 16 x 32 block = 171 flop per clock and SIMD, 419 TFLOP/s of fp32-equivalent peak against 157.
//
//   convert_panel_bf16x3   the fp64 panel of one outer step -> three bf16 planes, ONCE for all the tiles that read it
//                          (hi = rn(x), mid = rn(x - hi), lo = rn(x - hi - mid); |x - hi - mid - lo| <= 2^-25 |x|),
//                          laid out [plane][k / 32][row][k % 32]: the 128 rows x 32 k of a tile's chunk are 8 KB
//                          contiguous per plane
//   trailing_update_bf16x3_pair_kernel   128 x 128 tile of C per workgroup, 64 x 64 per wave (16 accumulators), K in
//                          chunks of 32 through ONE LDS stage (unpadded 64-B rows, the four 16-B pieces of a row
//                          permuted so that every ds_read_b128 lane group covers the 256-B bank row once), two
//                          workgroups per CU; C (fp64) read and written in the epilogue:
//                          C -= (double)(fp32 sum of the step's products) - the same contract as
//                          trailing_update_f32_kernel (gemm.hip), whose place it takes.
//   trailing_update_bf16x3_kernel        the first version (one workgroup per CU, two stages at an 80-B pitch, C
//                          prefetched during the loop), kept behind AGP_BF16X3_KERNEL=1 for comparison.
//
// Measured (profiles/r05/time_bf16x3.txt, pmc_bulk_kernels_sq_counters.txt; N = 32768 trailing shapes, fp32-equivalent
// TFLOP/s at M = 8192 / 15872 / 30720; fp32 kernel: 87-89 / - / 105-107):
//   first version    one workgroup per CU, two LDS stages at an 80-B pitch, C prefetched into 128 registers:
//                    97-99 / 98 / 115-118.  PMC: the MFMA pipe busy 30 % of the time, the (single) wave per SIMD at an
//                    s_waitcnt 52 % of its cycles - 9 % of them for LDS, the rest for the global loads of the next
//                    chunk -, and HALF of the LDS cycles bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
//   pair kernel      one stage, no prefetched C, 163 registers: two workgroups per CU, one's MFMAs in the other's
//                    waits: 124-128 / 118 / 141.
//   + swizzled LDS   conflict-free reads AND writes (below), 48 KB per workgroup: 144-146 / 125-127 / 150.
// In the fit the bulk stream runs it back to back at 142 TFLOP/s (80 of the factorisation's 87 ms at N = 32768): what
// is left is the memory system - 768 KB of planes per tile, 37 % of them past the L2, plus 256 KB of fp64 C read and
// written: ~4.9 TB/s beyond the L2 at 150 TFLOP/s.  (A second register stage - the loads of chunk k + 2 in flight while
// chunk k is multiplied, 212 registers - measured 141 against 144.5 TFLOP/s on one box: it is throughput, not latency.)
#include "common.h"
#include "gemm_tiles.h"

namespace agp {

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f32 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;                 // k per chunk = one v_mfma_f32_16x16x32_bf16
constexpr int BPITCH = 40;             // LDS row pitch in bf16 (80 B)
constexpr int BPLANE = GT * BPITCH;    // one plane of one operand of one stage, in bf16
// The 80-B pitch is NOT conflict-free: ds_read_b128 is served in four NON-contiguous groups of 16 lanes (lanes {0-3,
// 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS), in which rows 4-11 read k group lg + 1 while rows 0-3 / 12-15 read lg, and
// three pairs of them share a 16-B slot (PMC: SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE).  No pitch repairs that -
// a row's four pieces must be PERMUTED: unpadded 64-B rows, piece kg of row r at slot kg ^ ((-(r >> 2)) & 3).  Every
// group then covers the sixteen slots of the 256-B bank row once, and the eight lanes of a ds_write_b128 group (two
// whole rows) cover 128 contiguous bytes.
constexpr int SPITCH = 32;             // swizzled layout: row pitch in bf16 (64 B, no padding)
constexpr int SPLANE = GT * SPITCH;
__device__ __forceinline__ int sw_piece(int kg, int row) { return kg ^ ((4 - ((row >> 2) & 3)) & 3); }

__device__ __forceinline__ unsigned short bf16_rn(float f) {
  unsigned int u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even (finite inputs: the panel of a factorisation)
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_val(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }

// planes: 3 x [K / 32][rows_pad][32] bf16; thread = (row, chunk)
__global__ __launch_bounds__(256) void convert_panel_bf16x3_kernel(const double *__restrict__ P, long long ldp, long long rows, long long rows_pad,
                                                                   unsigned short *__restrict__ planes, long long plane_stride) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long c = blockIdx.y;
  if (row >= rows_pad) return;
  unsigned short *dst = planes + c * rows_pad * BK + row * BK;
#pragma unroll
  for (int q = 0; q < BK / 8; ++q) {
    unsigned short h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double x = row < rows ? P[row + (c * BK + 8 * q + j) * ldp] : 0.;
      h[j] = bf16_rn((float)x);
      const double r1 = x - (double)bf16_val(h[j]);
      m[j] = bf16_rn((float)r1);
      const double r2 = r1 - (double)bf16_val(m[j]);
      l[j] = bf16_rn((float)r2);
    }
    auto pack = [](const unsigned short (&v)[8]) {
      uint4 o;
      o.x = v[0] | ((unsigned int)v[1] << 16); o.y = v[2] | ((unsigned int)v[3] << 16);
      o.z = v[4] | ((unsigned int)v[5] << 16); o.w = v[6] | ((unsigned int)v[7] << 16);
      return o;
    };
    *reinterpret_cast<uint4 *>(dst + 8 * q) = pack(h);
    *reinterpret_cast<uint4 *>(dst + plane_stride + 8 * q) = pack(m);
    *reinterpret_cast<uint4 *>(dst + 2 * plane_stride + 8 * q) = pack(l);
  }
}

long long bf16x3_rows_pad(long long rows) { return (rows + GT - 1) / GT * GT + GT; }  // (+ one tile of zero rows: a tile may start anywhere below `rows`)
size_t bf16x3_bytes(long long rows, long long K) { return sizeof(unsigned short) * 3 * (size_t)bf16x3_rows_pad(rows) * (size_t)((K + BK - 1) / BK * BK); }

void launch_convert_panel_bf16x3(hipStream_t s, const double *P, long long ldp, long long rows, long long K, unsigned short *planes) {
  if (rows <= 0 || K <= 0 || K % BK) return;
  const long long rows_pad = bf16x3_rows_pad(rows);
  hipLaunchKernelGGL(convert_panel_bf16x3_kernel, dim3((unsigned)((rows_pad + 255) / 256), (unsigned)(K / BK)), dim3(256), 0, s, P, ldp, rows,
                     rows_pad, planes, rows_pad * K);
}

struct Bf16Args {
  double *C;
  long long ldc;
  const unsigned short *planes;  // of the panel both operands come from
  long long rows_pad, plane_stride;
  long long row_a, row_b;        // panel row of C's row 0 / of C's column 0
  long long M, N, K;
  int ntr, ntc;
  const int *order;              // XCD-aware tile order (gemm.hip: xcd_order) or nullptr
};

// tile (bi, bj) of the lower-triangular grid in column-major order (as gemm.hip's tile_of_block with tri = 1)
__device__ __forceinline__ bool bf16_tile_of_block(const Bf16Args &g, int &bi, int &bj) {
  long long id = blockIdx.x;
  bj = 0;
  while (bj < g.ntc && id >= g.ntr - bj) { id -= g.ntr - bj; ++bj; }
  if (bj >= g.ntc) return false;
  bi = bj + (int)id;
  return true;
}

// staging of one K chunk: piece q (16 B) of a plane tile = row q >> 2, k group q & 3; a thread moves pieces tid and tid + 256
// of the three planes of both operands - twelve NAMED registers and macros: an array or a struct of them passed to helper
// functions stayed in scratch once the loads were pinned in front of the MFMAs
#define AGP_BF_LOAD(OFF)                                                                                   \
  do {                                                                                                     \
    const long long o0_ = (OFF) + (long long)tid * 8, o1_ = o0_ + 256 * 8;                                 \
    sa00 = *reinterpret_cast<const uint4 *>(srcA + o0_);                    sa01 = *reinterpret_cast<const uint4 *>(srcA + o1_);                    \
    sb00 = *reinterpret_cast<const uint4 *>(srcB + o0_);                    sb01 = *reinterpret_cast<const uint4 *>(srcB + o1_);                    \
    sa10 = *reinterpret_cast<const uint4 *>(srcA + g.plane_stride + o0_);     sa11 = *reinterpret_cast<const uint4 *>(srcA + g.plane_stride + o1_);     \
    sb10 = *reinterpret_cast<const uint4 *>(srcB + g.plane_stride + o0_);     sb11 = *reinterpret_cast<const uint4 *>(srcB + g.plane_stride + o1_);     \
    sa20 = *reinterpret_cast<const uint4 *>(srcA + 2 * g.plane_stride + o0_); sa21 = *reinterpret_cast<const uint4 *>(srcA + 2 * g.plane_stride + o1_); \
    sb20 = *reinterpret_cast<const uint4 *>(srcB + 2 * g.plane_stride + o0_); sb21 = *reinterpret_cast<const uint4 *>(srcB + 2 * g.plane_stride + o1_); \
  } while (0)
#define AGP_BF_STORE_P(BASE, PL)                                                                            \
  do {                                                                                                     \
    unsigned short *b_ = (BASE);                                                                           \
    *reinterpret_cast<uint4 *>(b_ + 0 * (PL) + d0) = sa00; *reinterpret_cast<uint4 *>(b_ + 0 * (PL) + d1) = sa01; \
    *reinterpret_cast<uint4 *>(b_ + 1 * (PL) + d0) = sa10; *reinterpret_cast<uint4 *>(b_ + 1 * (PL) + d1) = sa11; \
    *reinterpret_cast<uint4 *>(b_ + 2 * (PL) + d0) = sa20; *reinterpret_cast<uint4 *>(b_ + 2 * (PL) + d1) = sa21; \
    *reinterpret_cast<uint4 *>(b_ + 3 * (PL) + d0) = sb00; *reinterpret_cast<uint4 *>(b_ + 3 * (PL) + d1) = sb01; \
    *reinterpret_cast<uint4 *>(b_ + 4 * (PL) + d0) = sb10; *reinterpret_cast<uint4 *>(b_ + 4 * (PL) + d1) = sb11; \
    *reinterpret_cast<uint4 *>(b_ + 5 * (PL) + d0) = sb20; *reinterpret_cast<uint4 *>(b_ + 5 * (PL) + d1) = sb21; \
  } while (0)
#define AGP_BF_STORE(BASE) AGP_BF_STORE_P(BASE, BPLANE)

__global__ __launch_bounds__(256, 1) void trailing_update_bf16x3_kernel(Bf16Args g) {
  __shared__ unsigned short lds[2 * 6 * BPLANE];  // [stage][operand A: hi mid lo | operand B: hi mid lo][128 rows][40]
  int bi, bj;
  if (g.order) {
    const int packed = g.order[blockIdx.x];
    if (packed < 0) return;
    bi = packed >> 16;
    bj = packed & 0xffff;
  } else if (!bf16_tile_of_block(g, bi, bj)) return;
  const long long i0 = (long long)bi * GT, j0 = (long long)bj * GT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int ln = lane & 15, lg = lane >> 4;

  const unsigned short *srcA = g.planes + (g.row_a + i0) * BK, *srcB = g.planes + (g.row_b + j0) * BK;
  const long long chunk_stride = g.rows_pad * BK;
  uint4 sa00, sa01, sa10, sa11, sa20, sa21, sb00, sb01, sb10, sb11, sb20, sb21;
  const int d0 = (tid >> 2) * BPITCH + (tid & 3) * 8, d1 = ((tid + 256) >> 2) * BPITCH + (tid & 3) * 8;

  v4f32 acc[4][4];  // [tj][ti]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = v4f32{0.f, 0.f, 0.f, 0.f};

  const long long nk = g.K / BK;
  AGP_BF_LOAD(0);
  AGP_BF_STORE(lds);
  __syncthreads();

  // C of this wave's 64 x 64 quadrant, fetched while the loop runs (register r of accumulator (tj, ti): row
  // 16 ti + ln of the quadrant, column 16 tj + 4 lg + r - the C/D map of every non-f64 MFMA)
  const bool interior = i0 + GT <= g.M && j0 + GT <= g.N;
  double *const cbase = g.C + (i0 + 64 * wr + ln) + (j0 + 64 * wc + 4 * lg) * g.ldc;
  double cpre[4][4][4];
  double cnew[4] = {0., 0., 0., 0.};
  bool fresh = false;
  const bool prefetch_c = interior && nk >= 16;
  const long long nq = nk / 16;
  long long kc = 0;
  // (sixteen sections so that the part of C a section fetches is a compile-time index: one accumulator tile each)
#pragma unroll
  for (int part = 0; part < 16; ++part) {
    const long long k_end = (part == 15 || !prefetch_c) ? nk : (part + 1) * nq;
    bool first = prefetch_c;
    for (; kc < k_end; ++kc) {
      const int cur = (int)(kc & 1);
      const unsigned short *S = lds + cur * (6 * BPLANE);
      const bool more = kc + 1 < nk;
      // (pinning these loads in front of the MFMAs with a scheduling barrier, and parking C in accumulation registers to make
      // room for that, were measured: 97-99 / 115-116 TFLOP/s at M = 15872 / 30720 in all four combinations - not the bound)
      AGP_BF_LOAD((more ? kc + 1 : kc) * chunk_stride);
      if (first) {
        first = false;
        fresh = true;
#pragma unroll
        for (int r = 0; r < 4; ++r) cnew[r] = __builtin_nontemporal_load(&cbase[16 * (part & 3) + (long long)(16 * (part >> 2) + r) * g.ldc]);
      }
      // fragments: A operand = the C-COLUMN panel (rows j0 ..), B operand = the C-ROW panel (rows i0 ..), as in gemm_tiles.h
      v8bf fa[3][4], fb[3][4];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          fa[p][t] = *reinterpret_cast<const v8bf *>(S + (3 + p) * BPLANE + (64 * wc + 16 * t + ln) * BPITCH + 8 * lg);
          fb[p][t] = *reinterpret_cast<const v8bf *>(S + p * BPLANE + (64 * wr + 16 * t + ln) * BPITCH + 8 * lg);
        }
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
          v4f32 a = acc[tj][ti];
          // smallest terms first
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2][tj], fb[0][ti], a, 0, 0, 0);  // lo hi
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][tj], fb[2][ti], a, 0, 0, 0);  // hi lo
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][tj], fb[1][ti], a, 0, 0, 0);  // mid mid
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][tj], fb[0][ti], a, 0, 0, 0);  // mid hi
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][tj], fb[1][ti], a, 0, 0, 0);  // hi mid
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][tj], fb[0][ti], a, 0, 0, 0);  // hi hi
          acc[tj][ti] = a;
        }
      if (fresh) {  // the part of C requested at the top of this chunk has arrived behind its MFMAs: park it
        fresh = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          cpre[part >> 2][part & 3][r] = cnew[r];
        }
      }
      if (more) AGP_BF_STORE(lds + (cur ^ 1) * (6 * BPLANE));
      __syncthreads();
    }
  }
  if (prefetch_c) {
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double cv = cpre[tj][ti][r];
          __builtin_nontemporal_store(cv - (double)acc[tj][ti][r], &cbase[16 * ti + (long long)(16 * tj + r) * g.ldc]);
        }
    return;
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
          *c = *c - (double)acc[tj][ti][r];
        }
      }
    }
}

// Cycle probe of the pair kernel (a -DAGP_BF16_STAMPS build only: scripts/build_variant.sh bf16_stamps -DAGP_BF16_STAMPS;
// read through agp_debug_bf16_probe by scripts/probe_bf16x3.py).  Wave 0 of the workgroup in the middle of the grid sums,
// over its K chunks, the shader cycles (s_memtime) of the four phases of a chunk - [0] barrier "stage free", [1] wait for
// the global loads + the twelve LDS stores, [2] barrier "stage full", [3] fragment reads + 96 MFMAs (+ the issue of the next
// chunk's loads) - and leaves [4] = cycles and [5] = 100 MHz ticks (s_memrealtime) of the whole loop, [6] = chunks, [7] = cycles of the epilogue (C read, subtract, write): the clock
// the chip held is 100 MHz x [4] / [5].  Every stamp sits where the kernel waits for lgkmcnt(0) anyway.
#ifdef AGP_BF16_STAMPS
__device__ unsigned long long g_bf16_probe[8];
#define AGP_BF_STAMP(var)                              \
  do {                                                 \
    __builtin_amdgcn_sched_barrier(0);                 \
    var = __builtin_amdgcn_s_memtime();                \
    __builtin_amdgcn_sched_barrier(0);                 \
  } while (0)
void read_bf16_probe(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bf16_probe), sizeof(unsigned long long) * 8); }
#else
#define AGP_BF_STAMP(var) do { } while (0)
void read_bf16_probe(unsigned long long *out) { for (int i = 0; i < 8; ++i) out[i] = 0; }
#endif

// ---- the same tile with TWO workgroups per CU ------------------------------------------------------------------
// The kernel above keeps one workgroup per CU (two LDS stages, 122 KB; 128 registers of prefetched C): ONE wave per
// SIMD, so every wait of that wave - the fragment reads in front of the MFMAs, the staging stores behind the global
// loads, the barrier - is a hole in the matrix pipe (PMC: the MFMA pipe is busy 28 % of the time).  Here a workgroup has
// ONE stage (61 KB) and no prefetched C (<= 256 registers): two workgroups share a CU and one's MFMAs run in the
// other's holes; C is read and written in the epilogue, behind the other workgroup's loop.
__global__ __launch_bounds__(256, 2) void trailing_update_bf16x3_pair_kernel(Bf16Args g) {
  __shared__ unsigned short lds[6 * SPLANE];  // [operand A: hi mid lo | operand B: hi mid lo][128 rows][32], 16-B pieces swizzled
  int bi, bj;
  if (g.order) {
    const int packed = g.order[blockIdx.x];
    if (packed < 0) return;
    bi = packed >> 16;
    bj = packed & 0xffff;
  } else if (!bf16_tile_of_block(g, bi, bj)) return;
  const long long i0 = (long long)bi * GT, j0 = (long long)bj * GT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int ln = lane & 15, lg = lane >> 4;

  const unsigned short *srcA = g.planes + (g.row_a + i0) * BK, *srcB = g.planes + (g.row_b + j0) * BK;
  const long long chunk_stride = g.rows_pad * BK;
  uint4 sa00, sa01, sa10, sa11, sa20, sa21, sb00, sb01, sb10, sb11, sb20, sb21;
  const int d0 = (tid >> 2) * SPITCH + sw_piece(tid & 3, tid >> 2) * 8, d1 = d0 + 64 * SPITCH;  // (row + 64: the same swizzle)

  v4f32 acc[4][4];  // [tj][ti]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = v4f32{0.f, 0.f, 0.f, 0.f};

  const long long nk = g.K / BK;
  AGP_BF_LOAD(0);
#ifdef AGP_BF16_STAMPS
  unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0, ta = 0, tb = 0, tc = 0, td = 0, te = 0;
  const unsigned long long loop_c0 = __builtin_amdgcn_s_memtime(), loop_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (long long kc = 0; kc < nk; ++kc) {
    AGP_BF_STAMP(ta);
    if (kc > 0) __syncthreads();  // every wave has read chunk kc - 1 out of the stage
    AGP_BF_STAMP(tb);
    AGP_BF_STORE_P(lds, SPLANE);
#ifdef AGP_BF16_STAMPS
    __builtin_amdgcn_s_waitcnt(0);  // (the barrier below waits for the stores anyway)
#endif
    AGP_BF_STAMP(tc);
    __syncthreads();
    AGP_BF_STAMP(td);
#ifdef AGP_DIAG_BF16_NOMEM  // diagnostic build: every chunk re-reads chunk 0 (L2 hits) - the ceiling without the memory side
    AGP_BF_LOAD((kc + 1 < nk ? 1 : 0) * chunk_stride);
#else
    AGP_BF_LOAD((kc + 1 < nk ? kc + 1 : kc) * chunk_stride);  // (unconditional: a guarded load kept the staging registers in scratch)
#endif
#ifdef AGP_BF16_EARLY_LOADS  // (measured, round 6: 198 registers, the waiting phase 1790 -> 1000 cycles, the matrix phase longer by as much)
    __builtin_amdgcn_sched_barrier(0);  // the loads of the next chunk stay in front of this chunk's MFMAs
#endif
    v8bf fa[3][4], fb[3][4];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[p][t] = *reinterpret_cast<const v8bf *>(lds + (3 + p) * SPLANE + (64 * wc + 16 * t + ln) * SPITCH + 8 * sw_piece(lg, ln));
        fb[p][t] = *reinterpret_cast<const v8bf *>(lds + p * SPLANE + (64 * wr + 16 * t + ln) * SPITCH + 8 * sw_piece(lg, ln));
      }
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        v4f32 a = acc[tj][ti];
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2][tj], fb[0][ti], a, 0, 0, 0);  // lo hi (smallest terms first)
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][tj], fb[2][ti], a, 0, 0, 0);  // hi lo
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][tj], fb[1][ti], a, 0, 0, 0);  // mid mid
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1][tj], fb[0][ti], a, 0, 0, 0);  // mid hi
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][tj], fb[1][ti], a, 0, 0, 0);  // hi mid
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][tj], fb[0][ti], a, 0, 0, 0);  // hi hi
        acc[tj][ti] = a;
      }
#ifdef AGP_BF16_STAMPS
    AGP_BF_STAMP(te);
    ph0 += tb - ta; ph1 += tc - tb; ph2 += td - tc; ph3 += te - td;
#endif
  }
#ifdef AGP_BF16_STAMPS
  if (blockIdx.x == gridDim.x / 2 && tid == 0) {
    g_bf16_probe[0] = ph0; g_bf16_probe[1] = ph1; g_bf16_probe[2] = ph2; g_bf16_probe[3] = ph3;
    g_bf16_probe[4] = __builtin_amdgcn_s_memtime() - loop_c0;
    g_bf16_probe[5] = __builtin_amdgcn_s_memrealtime() - loop_r0;
    g_bf16_probe[6] = (unsigned long long)nk;
  }
  const unsigned long long epi_c0 = __builtin_amdgcn_s_memtime();
#endif
  // C -= acc (register r of accumulator (tj, ti): row 16 ti + ln of the quadrant, column 16 tj + 4 lg + r)
  if (i0 + GT <= g.M && j0 + GT <= g.N) {
    double *const cbase = g.C + (i0 + 64 * wr + ln) + (j0 + 64 * wc + 4 * lg) * g.ldc;
#ifndef AGP_BF16_EPI
#define AGP_BF16_EPI 2  // (1: 163 registers, the epilogue 22 % longer and the kernel 1 % slower; 4: spills)
#endif
    constexpr int EPI = AGP_BF16_EPI;  // accumulator columns (of four) whose C is in flight at a time
#pragma unroll
    for (int t0 = 0; t0 < 4; t0 += EPI) {
      double cv[EPI][4][4];
#pragma unroll
      for (int e = 0; e < EPI; ++e)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
          for (int r = 0; r < 4; ++r) cv[e][ti][r] = __builtin_nontemporal_load(&cbase[16 * ti + (long long)(16 * (t0 + e) + r) * g.ldc]);
#pragma unroll
      for (int e = 0; e < EPI; ++e)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __builtin_nontemporal_store(cv[e][ti][r] - (double)acc[t0 + e][ti][r], &cbase[16 * ti + (long long)(16 * (t0 + e) + r) * g.ldc]);
    }
#ifdef AGP_BF16_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    if (blockIdx.x == gridDim.x / 2 && tid == 0) g_bf16_probe[7] = __builtin_amdgcn_s_memtime() - epi_c0;  // [7] = epilogue cycles
#endif
    return;
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
          *c = *c - (double)acc[tj][ti][r];
        }
      }
    }
}

#undef AGP_BF_LOAD
#undef AGP_BF_STORE
#undef AGP_BF_STORE_P

// AGP_BF16X3_KERNEL (api.hip): 1 = one workgroup per CU with prefetched C, otherwise two workgroups per CU
static int bf16x3_kernel_choice = 2;
// AGP_BF16X3_LDS_PAD: extra dynamic LDS per workgroup (bytes): 0 = three workgroups per CU (3 x 48 KB), >= 6 KB = two.
// Three are the faster KERNEL (+2 %) and the slower FIT (126.8 against 123.0 ms at N = 32768, same box): the panel
// kernels of the chain stream need LDS next to the bulk update.
static int bf16x3_lds_pad = 8192;

// C (M x N, lower tiles, C(0, 0) on the matrix diagonal) -= P[row_a ..] P[row_b ..]^T from the bf16 planes of ONE panel
// (launch_convert_panel_bf16x3).  order / order_len: the XCD-aware tile order of gemm.hip (nullptr: column-major tiles).
void launch_update_bf16x3(hipStream_t s, double *C, long long ldc, const unsigned short *planes, long long panel_rows, long long row_a,
                          long long row_b, long long M, long long N, long long K, const int *order, long long order_len) {
  if (M <= 0 || N <= 0 || K <= 0 || K % BK) return;
  Bf16Args g;
  g.C = C; g.ldc = ldc; g.planes = planes;
  g.rows_pad = bf16x3_rows_pad(panel_rows);
  g.plane_stride = g.rows_pad * K;
  g.row_a = row_a; g.row_b = row_b;
  g.M = M; g.N = N; g.K = K;
  g.ntr = (int)((M + GT - 1) / GT);
  g.ntc = (int)((N + GT - 1) / GT);
  if (g.ntc > g.ntr) g.ntc = g.ntr;
  g.order = order;
  long long tiles = 0;
  for (int bj = 0; bj < g.ntc; ++bj) tiles += g.ntr - bj;
  const long long wgs = order ? order_len : tiles;
  if (wgs <= 0) return;
  if (bf16x3_kernel_choice == 1) hipLaunchKernelGGL(trailing_update_bf16x3_kernel, dim3((unsigned)wgs), dim3(256), 0, s, g);
  else hipLaunchKernelGGL(trailing_update_bf16x3_pair_kernel, dim3((unsigned)wgs), dim3(256), (size_t)bf16x3_lds_pad, s, g);
}

void set_bf16x3_kernel(int choice, int lds_pad) { bf16x3_kernel_choice = choice; bf16x3_lds_pad = lds_pad; }

}  // namespace agp
