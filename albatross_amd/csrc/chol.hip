// chol.hip — K3: blocked right-looking LL^T on fp64 MFMA, with the forward
// substitution z = L^-1 y fused into the panel kernels.
//
// Replaces Eigen::SerializableLDLT(cov) + .solve(y)
// (include/albatross/src/eigen/serializable_ldlt.hpp:27, models/gp.hpp:67-68).
// The reference factor is a pivoted L D L^T; this one is un-pivoted L L^T —
// parity is defined on K^-1 y, log|K| and predictions.
//
// Structure (NB = 128, NBO = 512, micro block MB = 16):
//   for every outer block of NBO columns:
//     for every NB-wide diagonal block inside it:
//       potrf_diag_kernel   one workgroup: 128x128 block in LDS, 8 micro steps
//                           {register POTRF16 + INV16 | MFMA TRSM | MFMA SYRK}
//       trsm_panel_kernel   rows below: X <- X L11^-T by substitution over the
//                           micro blocks, all on MFMA; accumulators of finished
//                           micro columns are reused directly as MFMA operands
//       gemm_nt_sub (K=128) update of the rest of the outer block column
//     gemm_nt_sub (K=512)   trailing update  (the MFMA-bound bulk, gemm.hip)
#include <cstdlib>
#include "common.h"
#include "mfma_f64.h"
#include "gemm_tiles.h"
#include "trsm_kernel.h"
#include "pub.h"

namespace agp {
void launch_set_identity_batched(hipStream_t s, double *B, long long ld, long long stride, long long m, long long count);  // reduce.hip

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// 1/sqrt(x) to full fp64 accuracy: hardware estimate + two Newton steps
// (a short dependent chain: this sits on the serial pivot path)
__device__ __forceinline__ double rsqrt_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = y * (1.5 - h * y * y);
  y = y * (1.5 - h * y * y);
  return y;
}

struct PotrfArgs {
  double *A;        // matrix base
  long long lda;
  long long k0;     // first row/col of the diagonal block
  int nbk;          // valid size of the block (<= NB)
  double *img;      // tile image of this diagonal block (IMG_DOUBLES doubles, see below)
  double *y;        // y + k0 or nullptr
  int *flags;
  double *scalars;
  // batched launches (blockIdx.y = batch entry): element offsets per entry; all 0 = not batched
  long long batch_A = 0, batch_img = 0, batch_y = 0, batch_scalars = 0, batch_flags = 0, batch_zpub = 0;
  // fused panel kernel only: z of this block is also PUBLISHED here (device-scope stores) for the workgroups that solve
  // the rows below in the same launch; `below` = number of those rows
  double *zpub = nullptr;
  long long below = 0;
  // fused panel kernel with the PREVIOUS panel's update folded in (panel_fused_kernel<true>): columns k0 - 128 .. k0 - 1
  // hold the factored previous panel, whose rank-128 update of THIS panel's 128 columns has not been applied yet.
  // dpub: 36 tiles (the LDS tile layout of the diagonal block) through which the workgroups that update the diagonal
  // block hand it to the one that factors it (sentinel-filled, see store_pub)
  double *dpub = nullptr;
  // panel STEP kernel (panel_fused_kernel<true> in the chain-bound tail, see panel_phase): workgroups trail_first ..
  // apply the previous panel's rank-128 update to everything RIGHT of this panel (the `below` x `below` lower triangle,
  // 64 x 64 tiles) while this panel is factored - one launch per panel, no update launch, no second stream.
  unsigned trail_first = 0xffffffffu;
  // workgroup hold_index does nothing but keep its slot until workgroup 0 is done (see panel_phase)
  unsigned hold_index = 0xffffffffu, hold_count = 0;
  // step launches: the previous panel's update of THIS panel's rows below the diagonal block is done by the first
  // 2 x ceil(below / 64) trailing workgroups (64 x 64 tiles, device-scope stores), which count themselves in
  // rowcnt[64-row block] when complete; the workgroup that solves those 64 rows waits for rowcnt_expect there and reads
  // its rows with device-scope loads.  (The counters only grow: every step launch adds two per row block.)
  unsigned long long *rowcnt = nullptr;
  unsigned long long rowcnt_expect = 0;
  long long trail_tiles = 0, trail_workers = 1;  // tiles of the launch (row tiles first) / trailing workgroups that share them
};


// DPP row broadcast on fp64 (gfx90a+ "DPP64": row_newbcast only): every lane reads lane J of its
// own 16-lane row.  One VALU instruction, no SGPR round trip (v_readlane x2 + use).  The leading
// s_nop covers the "VALU write -> DPP read" hazard (2 wait states), which the compiler cannot see
// inside inline asm.
template <int J>
__device__ __forceinline__ double bcast_row(double v) {
  double r;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(J));
  return r;
}

// acc -= (lane J's value of src) * a
template <int J>
__device__ __forceinline__ void fnmac_bcast_row(double &acc, double src, double a) {
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
      : "+v"(acc) : "v"(src), "v"(a), "n"(J));
}

// ---- hand-scheduled 16 x 16 POTRF + inverse ---------------------------------------------------
// One wave issues in order, so the serial pivot chain (broadcast -> rsq -> Newton -> scale ->
// first rank-1 term) decides the run time unless the ~30 independent rank-1 terms of every column
// are slotted into its latency shadows.  The compiler does not do that for inline asm, so the
// arithmetic on the chain is volatile asm too and the issue order below is the source order:
// the terms of column C that are not needed at once ("deferred": a[j] for j > C + 1, the inverse
// sweep w[r] for r > C) are emitted between the chain steps of column C + 1.
__device__ __forceinline__ double vmul(double a, double b) {
  double r;
  asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vfma(double a, double b, double c) {
  double r;
  asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ double vfnma(double a, double b, double c) {  // c - a b
  double r;
  asm volatile("v_fma_f64 %0, -%1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ double vrsq(double a) {
  double r;
  asm volatile("v_rsq_f64 %0, %1" : "=v"(r) : "v"(a));
  return r;
}
template <int J>
__device__ __forceinline__ double vbcast(double v) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(J));
  return r;
}
// acc -= src[lane J of the row] * a.  HAZARD: src may have been written by the previous one or two
// VALU instructions (2 wait states before a DPP read); a whole issue slot, so only where needed.
template <int J, bool HAZARD>
__device__ __forceinline__ void vfnmac_bcast(double &acc, double src, double a) {
  if constexpr (HAZARD)
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(src), "v"(a), "n"(J));
  else
    asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc) : "v"(src), "v"(a), "n"(J));
}

// deferred term I of column K (K < 0: nothing).  Order: w[K] *= inv_K, then alternately the
// a-terms j = K + 2 + i and the w-terms r = K + 1 + i.
template <int K, int I>
__device__ __forceinline__ void potrf16_deferred(double (&a)[MB], double (&w)[MB], const double (&dinv)[MB]) {
  if constexpr (K >= 0) {
    constexpr int NA = MB - 2 - K;      // a-terms: j = K + 2 .. 15
    constexpr int NW = MB - 1 - K;      // w-terms: r = K + 1 .. 15
    if constexpr (I == 0) {
      w[K] = vmul(w[K], dinv[K]);
    } else {
      constexpr int q = (I - 1) / 2;
      if constexpr (((I - 1) & 1) == 0) {
        if constexpr (q < NA) vfnmac_bcast<K + 2 + q, false>(a[K + 2 + q], a[K], a[K]);
      } else {
        if constexpr (q < NW) vfnmac_bcast<K + 1 + q, false>(w[K + 1 + q], a[K], w[K]);
      }
    }
  }
}

template <int K, int I0, int I1>
__device__ __forceinline__ void potrf16_deferred_range(double (&a)[MB], double (&w)[MB], const double (&dinv)[MB]) {
  if constexpr (I0 < I1) {
    potrf16_deferred<K, I0>(a, w, dinv);
    potrf16_deferred_range<K, I0 + 1, I1>(a, w, dinv);
  }
}

constexpr int POTRF16_DEFERRED_MAX = 1 + 2 * MB;  // indices beyond a column's last term are no-ops

template <int C>
__device__ __forceinline__ void potrf16_column(double (&a)[MB], double (&w)[MB], double (&dinv)[MB], double &diag,
                                               int ln, int pivot_base, int &bad_pivot) {
  if constexpr (C < MB) {
    const double piv = vbcast<C>(a[C]);
    potrf16_deferred_range<C - 1, 0, 3>(a, w, dinv);
    const double y0 = vrsq(piv);
    potrf16_deferred_range<C - 1, 3, 8>(a, w, dinv);
    // third-order correction of the hardware estimate: e = 1 - piv y0^2, inv = y0 (1 + e/2 + 3 e^2/8)
    const double t = vmul(y0, y0);
    potrf16_deferred_range<C - 1, 8, 10>(a, w, dinv);
    const double e = vfnma(piv, t, 1.0);
    potrf16_deferred_range<C - 1, 10, 12>(a, w, dinv);
    const double pp = vfma(0.375, e, 0.5);
    const double ye = vmul(y0, e);
    potrf16_deferred_range<C - 1, 12, 13>(a, w, dinv);
    const double inv = vfma(ye, pp, y0);
    potrf16_deferred_range<C - 1, 13, 15>(a, w, dinv);
    a[C] = vmul(a[C], inv);  // lane C now holds piv * inv ~ sqrt(piv); the exact diagonal goes to `diag`
    dinv[C] = inv;
    potrf16_deferred_range<C - 1, 15, 17>(a, w, dinv);
    if constexpr (C + 1 < MB) vfnmac_bcast<C + 1, true>(a[C + 1], a[C], a[C]);  // the term the next pivot waits for
    potrf16_deferred_range<C - 1, 17, POTRF16_DEFERRED_MAX>(a, w, dinv);
    // off the chain: pivot check and the Heron-corrected diagonal entry
    if (!(piv > 0.) && bad_pivot == 0) bad_pivot = pivot_base + C + 1;
    double sq = piv * inv;
    sq = sq + 0.5 * inv * (piv - sq * sq);
    diag = (ln == C) ? sq : diag;
    potrf16_column<C + 1>(a, w, dinv, diag, ln, pivot_base, bad_pivot);
  }
}

// POTRF16 + INV16 of one diagonal micro tile by one wave, in registers.
// Lane ln (= lane & 15) owns row ln of the tile / column ln of the inverse; the four 16-lane
// rows of the wave compute the same thing.
// `a` holds row ln of the (updated) diagonal tile.  Results: the tile of L into D (LDS), its inverse into Wout (LDS) and
// img_diag (the tile image; PUB: published, see above).
template <bool PUB = false>
__device__ __forceinline__ void micro_potrf_inv_regs(double (&a)[MB], double *D, double *Wout, double *img_diag, int lane,
                                                     int ln, int pivot_base, int &bad_pivot) {
  double w[MB], dinv[MB], diag = 0.;
#pragma unroll
  for (int r = 0; r < MB; ++r) w[r] = (ln == r) ? 1. : 0.;
  potrf16_column<0>(a, w, dinv, diag, ln, pivot_base, bad_pivot);
  potrf16_deferred_range<MB - 1, 0, POTRF16_DEFERRED_MAX>(a, w, dinv);  // last column: only w[15] *= inv
  if (lane < MB) {
#pragma unroll
    for (int c = 0; c < MB; ++c) D[c * MB + ln] = (c < ln) ? a[c] : ((c == ln) ? diag : 0.);
#pragma unroll
    for (int r = 0; r < MB; ++r) {
      Wout[ln * MB + r] = w[r];
      if constexpr (PUB) store_pub(img_diag + ln * MB + r, w[r]);
      else img_diag[ln * MB + r] = w[r];
    }
  }
}

// POTRF16 + INV16 of one diagonal micro tile by one wave, in registers.
// Lane ln (= lane & 15) owns row ln of the tile / column ln of the inverse; the four 16-lane
// rows of the wave compute the same thing.
template <bool PUB = false>
__device__ __forceinline__ void micro_potrf_inv(double *D, double *Wout, double *img_diag, int lane, int ln, int pivot_base,
                                                int &bad_pivot) {
  double a[MB];
#pragma unroll
  for (int c = 0; c < MB; ++c) a[c] = D[c * MB + ln];
  micro_potrf_inv_regs<PUB>(a, D, Wout, img_diag, lane, ln, pivot_base, bad_pivot);
}

__device__ __forceinline__ void micro_syrk_tile(double *T, int ib, int kb, int jb, int ln, int lg) {
  double *Cc = T + tile_off(ib, kb);
  const double *Xk = T + tile_off(kb, jb), *Xi = T + tile_off(ib, jb);
  v4d acc0, acc1 = v4zero();
#pragma unroll
  for (int r = 0; r < 4; ++r) acc0[r] = Cc[(lg + 4 * r) * MB + ln];
  acc0 = mfma16(-Xk[(0 + lg) * MB + ln], Xi[(0 + lg) * MB + ln], acc0);
  acc1 = mfma16(-Xk[(4 + lg) * MB + ln], Xi[(4 + lg) * MB + ln], acc1);
  acc0 = mfma16(-Xk[(8 + lg) * MB + ln], Xi[(8 + lg) * MB + ln], acc0);
  acc1 = mfma16(-Xk[(12 + lg) * MB + ln], Xi[(12 + lg) * MB + ln], acc1);
#pragma unroll
  for (int r = 0; r < 4; ++r) Cc[(lg + 4 * r) * MB + ln] = acc0[r] + acc1[r];
}

// two SYRK tiles at once (independent accumulators: the MFMA latencies overlap)
__device__ __forceinline__ void micro_syrk_tile2(double *T, int ib0, int kb0, int ib1, int kb1, int jb, int ln, int lg) {
  double *C0 = T + tile_off(ib0, kb0), *C1 = T + tile_off(ib1, kb1);
  const double *Xk0 = T + tile_off(kb0, jb), *Xi0 = T + tile_off(ib0, jb);
  const double *Xk1 = T + tile_off(kb1, jb), *Xi1 = T + tile_off(ib1, jb);
  v4d a0, a1 = v4zero(), b0, b1 = v4zero();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    a0[r] = C0[(lg + 4 * r) * MB + ln];
    b0[r] = C1[(lg + 4 * r) * MB + ln];
  }
  a0 = mfma16(-Xk0[(0 + lg) * MB + ln], Xi0[(0 + lg) * MB + ln], a0);
  b0 = mfma16(-Xk1[(0 + lg) * MB + ln], Xi1[(0 + lg) * MB + ln], b0);
  a1 = mfma16(-Xk0[(4 + lg) * MB + ln], Xi0[(4 + lg) * MB + ln], a1);
  b1 = mfma16(-Xk1[(4 + lg) * MB + ln], Xi1[(4 + lg) * MB + ln], b1);
  a0 = mfma16(-Xk0[(8 + lg) * MB + ln], Xi0[(8 + lg) * MB + ln], a0);
  b0 = mfma16(-Xk1[(8 + lg) * MB + ln], Xi1[(8 + lg) * MB + ln], b0);
  a1 = mfma16(-Xk0[(12 + lg) * MB + ln], Xi0[(12 + lg) * MB + ln], a1);
  b1 = mfma16(-Xk1[(12 + lg) * MB + ln], Xi1[(12 + lg) * MB + ln], b1);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    C0[(lg + 4 * r) * MB + ln] = a0[r] + a1[r];
    C1[(lg + 4 * r) * MB + ln] = b0[r] + b1[r];
  }
}

constexpr int POTRF_LDS_DOUBLES = IMG_DOUBLES + 2 * MB * MB + NB;  // tiles | two inverse buffers | y

// Cycle probe of the factoring workgroup (a -DAGP_POTRF_TIMING build only: scripts/build_variant.sh probe -DAGP_POTRF_TIMING makes one,
// scripts/probe_potrf.py reads it through agp_debug_potrf_probe).  Slots [wave][stamp]: 0 = entry, 1 = block loaded and the
// first micro tile factored, then per micro step jb = 0 .. 6 two stamps: 2 + 2 jb = this wave's own work of stage B done,
// 3 + 2 jb = behind the barrier that ends the step; the last launch on the device wins.
// Round 6 used it to time a TWO-wave version (POTRF16 chain on wave 0, INV16 + the exact diagonal on wave 3, columns
// handed over through LDS - the split round 5 proposed because ~28 instructions per column at one fp64 issue per 8 cycles
// looked issue-bound): the chain ALONE takes 280-312 cycles per column against 275 with everything on one wave - it is
// bound by the latency of its nine dependent fp64 operations, the other terms already sit in their shadows -, the inverse
// lands ~1.1 k cycles behind it and the SYRK loses a wave: N = 512 0.191 instead of 0.165 ms, N = 4096 1.58 instead of
// 1.41 (profiles/r06/ab_potrf_two_waves.txt, potrf_probe_two_waves.txt).  Not kept; the code is in the git history.
// The same probe on the one-wave body: micro steps 0 and 1 of a block wait for the SYRK waves (27 / 20 tiles over three
// waves, ~1.1 k ticks per tile), the other five for wave 0 (5.5 k ticks each); four SYRK tiles per trip instead of two
// (eight MFMA chains, 48 reads in flight) changed neither (10.4 k ticks for step 0 both ways) and cost 32 registers.
// One dependent operation less per column by folding the pivot's broadcast into `v_rsq_f64_dpp ... row_newbcast` is not
// available: the instruction assembles for gfx950 and returns garbage (scripts/microbench/rsq_dpp_probe.hip) - DPP on
// 64-bit operands works for v_mov_b64 and v_fmac_f64 here, not for the transcendental.
#ifdef AGP_POTRF_TIMING
__device__ unsigned long long g_potrf_probe[4 * 32];
#define AGP_PROBE(slot)                                                                             \
  do {                                                                                              \
    if ((threadIdx.x & 63) == 0) g_potrf_probe[(threadIdx.x >> 6) * 32 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
void read_potrf_probe(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_potrf_probe), sizeof(unsigned long long) * 4 * 32); }
#else
#define AGP_PROBE(slot) do { } while (0)
void read_potrf_probe(unsigned long long *out) { for (int i = 0; i < 4 * 32; ++i) out[i] = 0; }
#endif

// PUB: the fused panel kernel - every tile of the image goes out (store_pub) the moment it is final, z_b too
template <bool PUB, bool UPD = false>
__device__ __forceinline__ void potrf_diag_body(PotrfArgs &p, double *T) {
  double *Wc = T + IMG_DOUBLES;
  double *ys = Wc + 2 * MB * MB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 15, lg = lane >> 4;
  const int nbk = p.nbk;
  AGP_PROBE(0);

  double *Adiag = p.A + p.k0 * p.lda + p.k0;  // element (r, c) of the block at Adiag[c * lda + r]
  if constexpr (UPD) {
    // the block arrives UPDATED (D - X X^T, X = the previous panel's rows of this block) from the update workgroups
    // of the same launch, tile by tile in this very layout: poll until every value has appeared
    double v[NTILE];
    unsigned long long t0 = 0;
    for (int spin = 0;; ++spin) {
      bool ok = true;
#pragma unroll
      for (int t = 0; t < NTILE; ++t) v[t] = load_pub(p.dpub + t * (MB * MB) + tid);
#pragma unroll
      for (int t = 0; t < NTILE; ++t) ok = ok && !is_unpublished(v[t]);
      if (__all(ok)) break;
      if (spin == 0) t0 = __builtin_amdgcn_s_memrealtime();
      else if ((spin & 63) == 0 && poll_expired(t0, p.flags)) break;
      __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
    for (int t = 0; t < NTILE; ++t) T[t * (MB * MB) + tid] = v[t];
  } else {  // thread (r, c) of every tile; all 36 loads in flight
    const int r = tid & 15, c = tid >> 4;
    double v[NTILE];
#pragma unroll
    for (int ib = 0; ib < NMB; ++ib)
#pragma unroll
      for (int kb = 0; kb <= ib; ++kb) {
        const int gr = ib * MB + r, gc = kb * MB + c;
        double x;
        if (gr < nbk && gc < nbk) x = (gr >= gc) ? Adiag[gc * p.lda + gr] : 0.;
        else x = (gr == gc) ? 1. : 0.;  // identity padding of a partial last block
        v[ib * (ib + 1) / 2 + kb] = x;
      }
#pragma unroll
    for (int t = 0; t < NTILE; ++t) T[t * (MB * MB) + c * MB + r] = v[t];
  }
  if (tid < NB) ys[tid] = (p.y && tid < nbk) ? p.y[tid] : 0.;
  __syncthreads();

  int bad_pivot = 0;
  if (wave == 0) micro_potrf_inv<PUB>(T + tile_off(0, 0), Wc, p.img + tile_off(0, 0), lane, ln, 0, bad_pivot);
  __syncthreads();
  AGP_PROBE(1);

#pragma unroll 1
  for (int jb = 0; jb < NMB; ++jb) {
    const int o = jb * MB;
    const double *W = Wc + (jb & 1) * (MB * MB);
    // ---- stage A: micro TRSM of the tiles below, X <- X W^T, one tile per wave ----
    for (int ib = jb + 1 + wave; ib < NMB; ib += 4) {
      double *X = T + tile_off(ib, jb);
      v4d acc0 = v4zero(), acc1 = v4zero();
      acc0 = mfma16(W[(0 + lg) * MB + ln], X[(0 + lg) * MB + ln], acc0);
      acc1 = mfma16(W[(4 + lg) * MB + ln], X[(4 + lg) * MB + ln], acc1);
      acc0 = mfma16(W[(8 + lg) * MB + ln], X[(8 + lg) * MB + ln], acc0);
      acc1 = mfma16(W[(12 + lg) * MB + ln], X[(12 + lg) * MB + ln], acc1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double x = acc0[r] + acc1[r];  // final: element (ln, lg + 4 r) of tile (ib, jb) of L (the image holds -L)
        X[(lg + 4 * r) * MB + ln] = x;
        if constexpr (PUB) store_pub(p.img + tile_off(ib, jb) + (lg + 4 * r) * MB + ln, -x);
        else p.img[tile_off(ib, jb) + (lg + 4 * r) * MB + ln] = -x;
      }
    }
    // z_jb = W y_jb  (wave 3; reads precede the write in program order)
    if (wave == 3) {
      double zz = 0.;
#pragma unroll
      for (int c = 0; c < MB; ++c) zz += W[c * MB + ln] * ys[o + c];
      if (lane < MB) ys[o + ln] = zz;
    }
    __syncthreads();
    if (jb == NMB - 1) break;

    // ---- stage B: wave 0 updates the NEXT diagonal tile and factors it right
    // away (look-ahead) while waves 1-3 run the remaining SYRK tiles + y update
    if (wave == 0) {
      micro_syrk_tile(T, jb + 1, jb + 1, jb, ln, lg);
      micro_potrf_inv<PUB>(T + tile_off(jb + 1, jb + 1), Wc + ((jb + 1) & 1) * (MB * MB), p.img + tile_off(jb + 1, jb + 1),
                           lane, ln, o + MB, bad_pivot);
    } else {
      const int rem = NMB - 1 - jb;
      const int ntile = rem * (rem + 1) / 2;
      auto tile_of = [&](int tix, int &ib, int &kb) {
        int left = tix;
        kb = 0;
        while (left >= rem - kb) { left -= rem - kb; ++kb; }
        ib = jb + 1 + kb + left;
        kb = jb + 1 + kb;
      };
      // tix 0 is the diagonal tile done by wave 0; two tiles per trip: four independent MFMA chains
      int tix = wave;
      for (; tix + 3 < ntile; tix += 6) {
        int i0, k0, i1, k1;
        tile_of(tix, i0, k0);
        tile_of(tix + 3, i1, k1);
        micro_syrk_tile2(T, i0, k0, i1, k1, jb, ln, lg);
      }
      if (tix < ntile) {
        int i0, k0;
        tile_of(tix, i0, k0);
        micro_syrk_tile(T, i0, k0, jb, ln, lg);
      }
      const int row = o + MB + (tid - 64);
      if (row < NB) {
        const double *Xr = T + tile_off(row >> 4, jb) + (row & 15);
        double s = ys[row];
#pragma unroll
        for (int k = 0; k < MB; ++k) s -= Xr[k * MB] * ys[o + k];
        ys[row] = s;
      }
    }
    AGP_PROBE(2 + 2 * jb);
    __syncthreads();
    AGP_PROBE(3 + 2 * jb);
  }
  if constexpr (PUB) {  // z_b first: the workgroups below wait for it, nobody in this launch waits for the write-back of L11
    if (p.zpub && tid < nbk) store_pub(p.zpub + tid, ys[tid]);
  }

  {  // L11 back into the matrix: thread (r, c) of every tile
    const int r = tid & 15, c = tid >> 4;
#pragma unroll
    for (int ib = 0; ib < NMB; ++ib)
#pragma unroll
      for (int kb = 0; kb <= ib; ++kb) {
        const int gr = ib * MB + r, gc = kb * MB + c;
        if (gr < nbk && gc < nbk && gr >= gc) Adiag[gc * p.lda + gr] = T[(ib * (ib + 1) / 2 + kb) * (MB * MB) + c * MB + r];
      }
  }
  if (p.y && tid < nbk) p.y[tid] = ys[tid];
  // sum log L_ii of this block (log_determinant, serializable_ldlt.hpp:128-135):
  // 128 logs in parallel, fixed-order reduction
  double logsum = (tid < nbk) ? log(T[tile_off(tid >> 4, tid >> 4) + (tid & 15) * (MB + 1)]) : 0.;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) logsum += __shfl_down(logsum, off, 64);
  __syncthreads();
  if (lane == 0) ys[wave] = logsum;
  __syncthreads();
  if (tid == 0) {
    p.scalars[0] += (ys[0] + ys[1]) + (ys[2] + ys[3]);
    if (bad_pivot && p.flags[1] == 0) p.flags[1] = (int)(p.k0 + bad_pivot);
  }
}

__global__ __launch_bounds__(256) void potrf_diag_kernel(PotrfArgs p) {
  // serial panel chain: issue ahead of the bulk-update waves that share this CU
  __builtin_amdgcn_s_setprio(AGP_CHAIN_PRIO);
  {
    const long long b = blockIdx.y;
    p.A += b * p.batch_A;
    p.img += b * p.batch_img;
    if (p.y) p.y += b * p.batch_y;
    if (p.scalars) p.scalars += b * p.batch_scalars;
    if (p.flags) p.flags += b * p.batch_flags;
  }
  __shared__ double T[POTRF_LDS_DOUBLES];
  potrf_diag_body<false>(p, T);
}

// ---------------------------------------------------------------------------------------------------------------
// Fused panel kernel: ONE launch per 128-column panel instead of POTRF -> TRSM.
//   workgroup 0        factors the diagonal block (potrf_diag_body<true>) and publishes its tile image as it goes
//   workgroup 1 + i    solves rows [64 i, 64 i + 64) below the block, X <- X L11^-T (+ the fused y -= X z_b), micro
//                      column by micro column, each as soon as the row of image tiles it needs has appeared
// Row jb of the image (tiles (jb, 0 .. jb - 1)) is final after micro step jb - 1's TRSM stage of the producer, the
// inverted diagonal tile (jb, jb) after its look-ahead POTRF16: the consumers run one micro step behind the producer
// and finish ~1 us after it, instead of starting a 12-40 us kernel after a launch gap.  Consumers keep nothing in
// LDS and never synchronise: every wave reads its MFMA A-operand fragments straight from L2.
// Workgroup 0 is dispatched first (workgroups are dispatched in order), so the consumers only ever wait for a
// workgroup that is already running or will be as soon as a slot of its XCD frees up.
// ---------------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void poll_tiles(const double *img_row, int lane, double (&f)[NMB][4], int *flags) {
  // tiles 0 .. NT - 1 of one image row, as A-operand fragments (f[t][s] = element s * 64 + lane of tile t)
  unsigned long long t0 = 0;
  for (int spin = 0;; ++spin) {
    bool ok = true;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) f[t][s] = load_pub(img_row + t * (MB * MB) + s * 64 + lane);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) ok = ok && !is_unpublished(f[t][s]);
    if (__all(ok)) break;
    if (spin == 0) t0 = __builtin_amdgcn_s_memrealtime();
    else if ((spin & 63) == 0 && poll_expired(t0, flags)) break;
    __builtin_amdgcn_s_sleep(4);
  }
}

template <int JB>
__device__ __forceinline__ void trsm_fused_step(const PotrfArgs &p, int lane, v4d (&Y)[NMB], double *base, bool nok,
                                                int lg) {
  if constexpr (JB < NMB) {
    const double *row = p.img + (JB * (JB + 1) / 2) * (MB * MB);
    v4d pa[4] = {Y[JB], v4zero(), v4zero(), v4zero()};
    double f[NMB][4];
    if constexpr (JB > 0) {
      poll_tiles<JB>(row, lane, f, p.flags);
#pragma unroll
      for (int ib = 0; ib < JB; ++ib)
#pragma unroll
        for (int s = 0; s < 4; ++s) pa[s] = mfma16(f[ib][s], Y[ib][s], pa[s]);
    }
    const v4d acc = (pa[0] + pa[1]) + (pa[2] + pa[3]);
    poll_tiles<1>(row + JB * (MB * MB), lane, f, p.flags);  // the inverted diagonal tile: the last thing the producer emits for this row
    v4d po[4] = {v4zero(), v4zero(), v4zero(), v4zero()};
#pragma unroll
    for (int s = 0; s < 4; ++s) po[s] = mfma16(f[0][s], acc[s], po[s]);
    const v4d out = (po[0] + po[1]) + (po[2] + po[3]);
    Y[JB] = out;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = JB * MB + lg + 4 * r;
      if (nok && m < p.nbk) store_pub(base + m * p.lda, out[r]);  // (write-through: read by the next launch, see trail_tile64)
    }
    trsm_fused_step<JB + 1>(p, lane, Y, base, nok, lg);
  }
}

// UPD: one of the 36 micro tiles of the diagonal block per wave: D'(ib, kb) = D(ib, kb) - X_ib X_kb^T with
// X = the previous panel's rows of this block (128 deep: 32 MFMAs in four chains), published to p.dpub in the LDS tile
// layout of potrf_diag_body.  Nine workgroups of four waves cover the block; each finishes in ~2 us.
__device__ __forceinline__ void diag_update_body(const PotrfArgs &p, int tile) {
  const int lane = threadIdx.x & 63, ln = lane & 15, lg = lane >> 4;
  int ib = 0;
  while ((ib + 1) * (ib + 2) / 2 <= tile) ++ib;
  const int kb = tile - ib * (ib + 1) / 2;
  const double *Xd = p.A + (p.k0 - NB) * p.lda + p.k0;  // element (row r of the block, depth k) at Xd[k * lda + r]
  const double *Dd = p.A + p.k0 * p.lda + p.k0;         // element (r, c) of the block at Dd[c * lda + r]
  const int ra = kb * MB + ln, rb = ib * MB + ln;
  const bool oka = ra < p.nbk, okb = rb < p.nbk;
  v4d acc[4] = {v4zero(), v4zero(), v4zero(), v4zero()};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gr = ib * MB + ln, gc = kb * MB + lg + 4 * r;  // D/C layout: register r of lane (ln, lg) = element (row ln, column lg + 4 r)
    double x;
    if (gr < p.nbk && gc < p.nbk) x = (gr >= gc) ? Dd[gc * p.lda + gr] : 0.;
    else x = (gr == gc) ? 1. : 0.;  // identity padding of a partial last block
    acc[0][r] = x;
  }
  {
    // the whole depth (32 k-steps x 2 operands) in flight at once: one load round trip on the launch's critical path
    double av[32], bv[32];
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) {
      const long long k = 4 * s2 + lg;
      av[s2] = oka ? Xd[k * p.lda + ra] : 0.;
      bv[s2] = okb ? Xd[k * p.lda + rb] : 0.;
    }
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) acc[s2 & 3] = mfma16(-av[s2], bv[s2], acc[s2 & 3]);
  }
  const v4d out = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    // the strictly upper part of a diagonal tile stays zero, as the prologue of potrf_diag_body loads it
    const int gr = ib * MB + ln, gc = kb * MB + lg + 4 * r;
    store_pub(p.dpub + tile * (MB * MB) + (lg + 4 * r) * MB + ln, (gr >= gc) ? out[r] : 0.);
  }
}

template <bool UPD>
__device__ __forceinline__ void trsm_fused_body(const PotrfArgs &p, int first_block) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ln = lane & 15, lg = lane >> 4;
  const long long n0 = ((long long)(blockIdx.x - first_block) * 4 + wave) * 16;
  const bool active = n0 < p.below;
  const bool nok = active && n0 + ln < p.below;
  double *base = p.A + p.k0 * p.lda + (p.k0 + p.nbk) + (n0 + ln);  // X[n][m] at base[m * lda]
  const bool handed = UPD && p.rowcnt != nullptr;  // the rows arrive updated from the first trailing workgroups of this launch
  if (handed) {
    if (tid == 0) {
      const unsigned long long *c = p.rowcnt + (blockIdx.x - first_block);
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (int spin = 1; __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p.rowcnt_expect; ++spin) {
        if ((spin & 63) == 0 && poll_expired(t0, p.flags)) break;
        __builtin_amdgcn_s_sleep(8);
      }
      // pairs with the RELEASE of trail_tile64<true>'s count: the rows read below are ordered behind the count in the
      // memory model too (on this hardware they already are - every hand-over load is device-scope, sc1, and cannot
      // hit a stale line of the XCD-private L2 -, so the poll itself stays RELAXED and the fence costs one s_waitcnt)
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
  }
  if (!active) return;  // from here on the waves are independent: no barrier below
  v4d Y[NMB];
#pragma unroll
  for (int jb = 0; jb < NMB; ++jb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = jb * MB + lg + 4 * r;
      Y[jb][r] = (nok && m < p.nbk) ? (handed ? load_pub(base + m * p.lda) : base[m * p.lda]) : 0.;
    }
  trsm_fused_step<0>(p, lane, Y, base, nok, lg);
  if (p.y) {
    // y[n] -= sum_m X[n][m] z[m]: z_b is published when the producer has finished the whole block.  All 32 values of this
    // lane in flight at once and re-read together until none is the sentinel (one element at a time, each poll a
    // dependent L2 round trip, took 6 us - measured with scripts/diag_step.py - and was the last thing the launch waited for)
    double z[NMB][4];
    unsigned long long t0 = 0;
    for (int spin = 0;; ++spin) {
      bool ok = true;
#pragma unroll
      for (int jb = 0; jb < NMB; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = jb * MB + lg + 4 * r;
          z[jb][r] = (m < p.nbk) ? load_pub(p.zpub + m) : 0.;
        }
#pragma unroll
      for (int jb = 0; jb < NMB; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) ok = ok && !is_unpublished(z[jb][r]);
      if (__all(ok)) break;
      if (spin == 0) t0 = __builtin_amdgcn_s_memrealtime();
      else if ((spin & 63) == 0 && poll_expired(t0, p.flags)) break;
      __builtin_amdgcn_s_sleep(2);
    }
    double part = 0.;
#pragma unroll
    for (int jb = 0; jb < NMB; ++jb)
#pragma unroll
      for (int r = 0; r < 4; ++r) part += Y[jb][r] * z[jb][r];
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    if (lg == 0 && nok) p.y[p.nbk + n0 + ln] -= part;
  }
}

constexpr int UPD_BLOCKS = NTILE / 4;  // 9 workgroups x 4 waves = the 36 micro tiles of the diagonal block

// One 64 x 64 tile (lower triangle, column-major tile order) of  C[k0 + NB .., k0 + NB ..] -= X X^T,  X = rows k0 + NB ..
// of the PREVIOUS panel (columns k0 - NB .. k0 - 1, final since the previous launch).  Nothing in this launch reads or
// writes those tiles, so these workgroups never wait: they fill the chip behind the ~35 us of the panel's POTRF.
// 64 x 64 tile, depth exactly NB = 128, in TWO passes of 64: each pass brings both operand blocks (64 rows x 64 deep)
// into LDS with one round of loads per thread - the second pass's loads are in flight while the first one multiplies -
// instead of the generic body's eight 16-deep chunks (eight load / barrier round trips, which is what a workgroup
// with only two resident workgroups per CU waits for).  lds: 2 * 64 * TRP doubles.
// LDS image of one 64-deep pass: [64 k rows][64] doubles, NO padding; element (k row, col) at col ^ ((k row & 1) << 4).
// A ds_read_b64 is served in two groups of 32 lanes = two k rows x 16 consecutive columns: with a 512-byte pitch both
// rows would sit on the same half of the 256-byte bank row - the swap of the two 16-column halves in odd k rows puts
// them on opposite halves.  (The first version padded the pitch to 72 on the belief that any pitch is conflict-free for
// 16 consecutive lanes: 25 % of the step launches' LDS cycles were bank conflicts, profiles/r05/pmc_lds_n4096.txt.)
// Staging: thread t moves the row pairs 2 (t & 15) and 2 (t & 15) + 32 of k row t >> 4 (+ 16 c): sixteen lanes = 256
// contiguous bytes in memory, eight lanes of a ds_write_b128 = 128 contiguous bytes in LDS.
constexpr int TRP = ST;
__device__ __forceinline__ int trail_col(int krow, int col) { return col ^ ((krow & 1) << 4); }

__device__ __forceinline__ void trail_load_pass(const double *__restrict__ P, long long ld, long long row0, long long nrows,
                                                int k0, bool vec_ok, double (&r)[16]) {
  const int t = threadIdx.x, kk = t >> 4, seg = (t & 15) * 2;
  const long long row = row0 + seg;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const double *q = P + row + (long long)(k0 + 16 * c + kk) * ld;
    if (vec_ok && row0 + ST <= nrows) {
      const double2 a = *reinterpret_cast<const double2 *>(q);
      const double2 b = *reinterpret_cast<const double2 *>(q + 32);
      r[4 * c] = a.x; r[4 * c + 1] = a.y; r[4 * c + 2] = b.x; r[4 * c + 3] = b.y;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int off = 32 * (e >> 1) + (e & 1);
        r[4 * c + e] = (row + off < nrows) ? q[off] : 0.;
      }
    }
  }
}

template <bool NEGATE>
__device__ __forceinline__ void trail_store_pass(double *__restrict__ Ls, const double (&r)[16]) {
  const int t = threadIdx.x, kk = t >> 4, seg = (t & 15) * 2;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int krow = 16 * c + kk;
    double *row = Ls + krow * TRP;
    double2 *d0 = reinterpret_cast<double2 *>(row + trail_col(krow, seg)), *d1 = reinterpret_cast<double2 *>(row + trail_col(krow, seg + 32));
    *d0 = NEGATE ? make_double2(-r[4 * c], -r[4 * c + 1]) : make_double2(r[4 * c], r[4 * c + 1]);
    *d1 = NEGATE ? make_double2(-r[4 * c + 2], -r[4 * c + 3]) : make_double2(r[4 * c + 2], r[4 * c + 3]);
  }
}

// Barrier for LDS hand-overs only: waits for this wave's LDS operations, not for its global loads and stores in flight
// (__syncthreads() drains those too: at every tile boundary the trailing workgroups waited for their C stores to be
// acknowledged before the next tile's loads could even be issued).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// K: depth of the product (a multiple of 64; the panel P is K columns wide) - passes of 64 through LDS, the next pass's
// loads in flight while this one multiplies.  WT: write-through stores (see below); false: plain stores.

template <bool PUBLISH = false, bool WT = true>
__device__ __forceinline__ void trail_tile64(double *__restrict__ Cc, const double *__restrict__ P, long long ld, long long M,
                                             long long i0, long long j0, double *lds, unsigned long long *done = nullptr,
                                             int K = NB) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1, ln = lane & 15, lg = lane >> 4;
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(P) & 15) == 0) && ((ld & 1) == 0);
  double *As = lds, *Bs = lds + 64 * TRP;
  double ra[16], rb[16];
  trail_load_pass(P, ld, i0, M, 0, vec_ok, ra);
  trail_load_pass(P, ld, j0, M, 0, vec_ok, rb);
  v4d acc[2][2];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      const long long row = i0 + 32 * wr + 16 * ti + ln;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long col = j0 + 32 * wc + 16 * tj + lg + 4 * r;
        acc[tj][ti][r] = (row < M && col < M) ? Cc[row + col * ld] : 0.;
      }
    }
  const int npass = K / 64;
  for (int pass = 0; pass < npass; ++pass) {
    if (pass) lds_barrier();  // the previous pass's readers are done with the buffers
    trail_store_pass<false>(As, ra);
    trail_store_pass<true>(Bs, rb);
    lds_barrier();
    if (pass + 1 < npass) {
      trail_load_pass(P, ld, i0, M, 64 * (pass + 1), vec_ok, ra);
      trail_load_pass(P, ld, j0, M, 64 * (pass + 1), vec_ok, rb);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int krow = (4 * s + lg) * TRP, swz = (lg & 1) << 4;  // (k row 4 s + lg: odd with lg)
      double fa[2], fb[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fa[t] = Bs[krow + ((32 * wc + 16 * t + ln) ^ swz)];
        fb[t] = As[krow + ((32 * wr + 16 * t + ln) ^ swz)];
      }
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) acc[tj][ti] = mfma16(fa[tj], fb[ti], acc[tj][ti]);
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
        // device-scope (write-through) stores for every tile: what a step launch writes is read by the NEXT launch, on other
        // XCDs - left dirty in this XCD's L2 it is written back at the kernel boundary, R^2 / 2 doubles at once, and the
        // next launch's first loads (the diagonal block's update, on its critical path) queue behind that
        if (row < M && col < M) {
          if (WT) store_pub(Cc + row + col * ld, acc[tj][ti][r]);
          else Cc[row + col * ld] = acc[tj][ti][r];
        }
      }
    }
  if constexpr (PUBLISH) {
    // every store of this tile acknowledged, then one count: a reader that sees the count reads final values
    // (the tile's stores are write-through already; the RELEASE on the count is what orders them before it in the
    // language's memory model too, not only on this hardware)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ void trail_one_tile(const PotrfArgs &p, long long id, double *lds) {
  if (p.rowcnt) {
    // the first 2 x nrb tiles: this panel's own columns, rows below the diagonal block (tile column 0, then 1)
    const long long nrb = (p.below + ST - 1) / ST;
    if (id < 2 * nrb) {
      const long long bj = id / nrb, bi = id % nrb;
      // origin at (k0, k0): rows of the diagonal block = rows 0 .. 127 of the previous panel's rows from k0 on
      trail_tile64<true>(p.A + p.k0 * p.lda + p.k0, p.A + (p.k0 - NB) * p.lda + p.k0, p.lda, p.below + NB, (2 + bi) * ST, bj * ST,
                         lds, p.rowcnt + bi);
      return;
    }
    id -= 2 * nrb;
  }
  const long long t0 = p.k0 + NB;
  double *Cc = p.A + t0 * p.lda + t0;
  const double *P = p.A + (p.k0 - NB) * p.lda + t0;
  const int ntr = (int)((p.below + ST - 1) / ST);
  int bj = 0;
  while (id >= ntr - bj) {
    id -= ntr - bj;
    ++bj;
  }
  trail_tile64(Cc, P, p.lda, p.below, (long long)(bj + (int)id) * ST, (long long)bj * ST, lds);
}

// A trailing workgroup takes the tiles worker, worker + trail_workers, ... : no more workgroups than fit on the chip next
// to the critical ones.  (One workgroup per tile: at 4000 remaining rows the dispatcher starts and retires 2600 of them
// during the launch - 20 ns each even when they do nothing - and every start on a CU disturbs what runs there: with
// empty trailing workgroups the diagonal block reached workgroup 0 after 44 us instead of 6, scripts/diag_step.py.)
__device__ __forceinline__ void trail_update_body(const PotrfArgs &p, long long worker, double *lds) {
  for (long long id = worker; id < p.trail_tiles; id += p.trail_workers) {
    if (id != worker) lds_barrier();  // the previous tile's readers are done with lds
    trail_one_tile(p, id, lds);
  }
}

template <bool UPD>
__global__ __launch_bounds__(256, 2) void panel_fused_kernel(PotrfArgs p) {
  __shared__ double T[POTRF_LDS_DOUBLES];
  static_assert(POTRF_LDS_DOUBLES >= 2 * 64 * TRP, "the trailing-update workgroups stage their operands in T");
  if (!UPD && blockIdx.y > 0) {  // batched fused panels (factor_lower_batched): blockIdx.y = problem
    const long long b = blockIdx.y;
    p.A += b * p.batch_A;
    p.img += b * p.batch_img;
    if (p.y) p.y += b * p.batch_y;
    if (p.zpub) p.zpub += b * p.batch_zpub;
    if (p.scalars) p.scalars += b * p.batch_scalars;
    if (p.flags) p.flags += b * p.batch_flags;
  }
  if (UPD && blockIdx.x >= p.trail_first) {
    if (blockIdx.x >= p.hold_index && blockIdx.x < p.hold_index + p.hold_count) {
      // placeholder: idle in the slot next to workgroup 0 until the last tile of the image is out
      const double *last = p.img + tile_off(NMB - 1, NMB - 1) + MB * MB - 1;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (int spin = 1; is_unpublished(load_pub(last)); ++spin) {
        if ((spin & 63) == 0 && poll_expired(t0, p.flags)) break;
        __builtin_amdgcn_s_sleep(64);
      }
      return;
    }
    trail_update_body(p, (long long)(blockIdx.x - p.trail_first) - (blockIdx.x > p.hold_index ? (long long)p.hold_count : 0), T);
    return;
  }
  __builtin_amdgcn_s_setprio(AGP_CHAIN_PRIO);
  if (blockIdx.x == 0) {
    potrf_diag_body<true, UPD>(p, T);
  } else if (UPD && blockIdx.x <= UPD_BLOCKS) {
    diag_update_body(p, (int)(blockIdx.x - 1) * 4 + (int)(threadIdx.x >> 6));
  } else {
    trsm_fused_body<UPD>(p, UPD ? 1 + UPD_BLOCKS : 1);
  }
}

// In the TRANS staging above the image is indexed t = jb(jb+1)/2 + ib with
// ib <= jb; for the transposed solve the "row block" of the stored pair is jb
// and the "column block" ib, i.e. tile (jb, ib) of L, used when solving micro
// block ib with the already-solved block jb > ib.

static void launch_potrf(hipStream_t s, double *A, long long lda, long long k0, int nbk, double *img,
                         double *y, int *flags, double *scalars) {
  PotrfArgs p;
  p.A = A; p.lda = lda; p.k0 = k0; p.nbk = nbk;
  p.img = img + (k0 / NB) * (long long)IMG_DOUBLES;
  p.y = y ? y + k0 : nullptr;
  p.flags = flags; p.scalars = scalars;
  hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), 0, s, p);
}

// trailing update C -= P P^T (lower tiles) bracketed by a HIP-event pair when
// the caller collects per-launch timings (bench.py's roofline block)
static void timed_gemm(hipStream_t s, FactorTimers *timers, double *C, long long lda, const double *P,
                       const double *Q, long long M, long long N, long long K, bool bulk, int variant = -1,
                       const float *P32 = nullptr, long long ld32 = 0) {
  // only the bulk trailing updates' trailing_update_kernel launches (their own kernel symbol) are
  // event-timed: they run on the second stream, where an event gap is off the critical path
  if (!bulk) {
    launch_gemm_nt_sub(s, C, lda, P, lda, false, Q, lda, false, M, N, K, true);
    return;
  }
  BulkTiming bt;
  const bool timed = timers && timers->ev && timers->used + 2 <= timers->n_ev;
  if (timed) { bt.e0 = timers->ev[timers->used]; bt.e1 = timers->ev[timers->used + 1]; }
  if (variant >= 0) launch_trailing_update_as(variant, s, C, lda, P, Q, lda, M, K, timed ? &bt : nullptr, P32, P32, ld32);
  else launch_trailing_update(s, C, lda, P, Q, lda, M, K, timed ? &bt : nullptr);
  if (timed && bt.flops > 0.) {
    timers->flops[timers->used / 2] = bt.flops;  // algorithmic flop: 2 K per covered C entry on or below the diagonal
    timers->used += 2;
  }
}

// ---- the schedule's switch points (measured on one MI355X; DESIGN.md sections 3 and 8 hold the sweeps) ------------
// The two switches a caller can set - AGP_PANEL_FUSED=0 (POTRF and TRSM as two launches) and AGP_STEP_BELOW=<rows> (0: no
// step launches) - are read ONCE, at agp_context_create, into ctx->tune (api.hip); everything else is a constant.
constexpr long long FUSED_BELOW = 4608;     // remaining rows at or below which POTRF + TRSM are one fused launch (2048 .. 4608 best)
constexpr long long INNER_LEFT_ABOVE = 6144;  // left-looking inside an outer block while more rows than this remain
constexpr long long NBO_512_ABOVE = 2048, NBO_256_ABOVE = 1024;  // outer block width 512 / 256 / 128 by remaining rows
constexpr long long NBO_WIDE_ABOVE = 8192;  // ... and ctx->nbo_wide (mixed precision) above this
constexpr long long THROTTLE_BELOW = 8192;  // bulk updates handed to their stream by the host once their panel is done
constexpr long long U1_F32_ABOVE = 4096;    // mixed precision: U1 on the fp32 MFMA path while the block column is this tall
constexpr long long SINGLE_BELOW = 1536;    // the very end on one stream (when the step launches are off)
constexpr long long MASK_BELOW = 8704;      // bulk updates on the CU-masked stream from here on

// Workgroup slots of panel_fused_kernel<true> on this device: occupancy x CUs, asked of the runtime once per context.
static long long step_slots(agp_context *ctx) {
  if (ctx->step_slots <= 0) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, panel_fused_kernel<true>, 256, 0) != hipSuccess || per_cu <= 0) {
      (void)hipGetLastError();
      per_cu = 1;
    }
    ctx->step_slots = (long long)per_cu * ctx->cus;
  }
  return ctx->step_slots;
}

// May `rows` remaining rows be factored with step launches?  The row workgroups of a step launch wait for trailing
// workgroups that are dispatched AFTER them: all critical workgroups (10 + one per 64 rows) and at least 64 trailing ones
// must be resident at once, or the waiters would sit out their 2 s time-out.  Fewer slots than that (a small partition,
// a CU mask): the two-launch schedule.  Should the hand-over time out all the same (flags[2]), agp_fit_create repeats
// the fit without step launches (api.hip).
static bool step_fits(agp_context *ctx, long long rows) {
  return ctx->tune.step_below > 0 && rows <= ctx->tune.step_below && step_slots(ctx) >= 10 + rows / 64 + 64;
}

__global__ __launch_bounds__(256) void prep_kernel(PrepArgs a) {
  int e = 0;
  while (e + 1 < a.n && (long long)blockIdx.x >= a.first_block[e + 1]) ++e;
  const long long i0 = ((long long)blockIdx.x - a.first_block[e]) * 1024 + threadIdx.x;
  unsigned long long *d = a.dst[e];
  const unsigned long long *sr = a.src[e];
  const unsigned long long pat = a.pattern[e];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long i = i0 + 256 * u;
    if (i < a.count[e]) d[i] = sr ? sr[i] : pat;
  }
}

__global__ __launch_bounds__(256) void copy_table_kernel(const CopyItem *__restrict__ table) {
  const CopyItem it = table[blockIdx.y];
  const long long i0 = (long long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long i = i0 + 256 * u;
    if (i < it.words) it.dst[i] = it.src[i];
  }
}

void launch_copy_table(hipStream_t s, const CopyItem *table_dev, long long count, long long max_words) {
  if (count <= 0 || max_words <= 0) return;
  hipLaunchKernelGGL(copy_table_kernel, dim3((unsigned)((max_words + 1023) / 1024), (unsigned)count), dim3(256), 0, s, table_dev);
}

void launch_prep(hipStream_t s, const PrepArgs &a) {
  if (a.n <= 0) return;
  hipLaunchKernelGGL(prep_kernel, dim3((unsigned)a.first_block[a.n]), dim3(256), 0, s, a);
}

// Before the fused panel kernels of one factorisation run: the tile images of the diagonal blocks [k_begin, k_end) and
// the z slots of those rows are sentinel-filled (and, for step launches, the second image and the row counters).
// panel_fused_plan makes sure the buffers exist and APPENDS the fills to `prep`, which the caller launches (stream-ordered
// before the first panel launch) together with whatever else it has to fill or copy; panel_fused_prepare is plan + launch.
// Not planned (and the two-launch path is used) if the fused kernel is off or a buffer cannot be allocated.
bool panel_fused_plan(agp_context *ctx, double *invd, long long k_begin, long long k_end, bool want_step, PrepArgs *prep) {
  ctx->img_ready = nullptr;
  ctx->headcnt_ready = false;  // (set below, only together with the fill that zeroes the counters)
  if (!ctx->tune.panel_fused || k_end <= k_begin) return false;
  if (ctx->zpub_cap < k_end) {
    // grow; the old buffer may still be read by kernels in flight on this context's streams: drain them first
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_zpub) (void)hipFree(ctx->d_zpub);
    ctx->d_zpub = nullptr;
    ctx->zpub_cap = 0;
    const long long cap = (k_end + 4095) / 4096 * 4096;
    if (hipMalloc(&ctx->d_zpub, sizeof(double) * (size_t)cap) != hipSuccess) { (void)hipGetLastError(); return false; }
    ctx->zpub_cap = cap;
  }
  const long long b0 = k_begin / NB, b1 = (k_end + NB - 1) / NB;
  if (want_step && ctx->tune.step_below > 0 && ctx->dpub_cap < b1) {
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_dpub) (void)hipFree(ctx->d_dpub);
    ctx->d_dpub = nullptr;
    ctx->d_rowcnt = nullptr;
    ctx->dpub_cap = 0;
    const long long cap = (b1 + 31) / 32 * 32;
    // (+ two 64-row counters per diagonal block behind the images: the hand-over of the step launches' row updates)
    if (hipMalloc(&ctx->d_dpub, sizeof(double) * (size_t)cap * (IMG_DOUBLES + 2)) == hipSuccess) {
      ctx->dpub_cap = cap;
      ctx->d_rowcnt = reinterpret_cast<unsigned long long *>(ctx->d_dpub + cap * (long long)IMG_DOUBLES);
    }
    else (void)hipGetLastError();
  }
  const long long cnt_img = (b1 - b0) * (long long)IMG_DOUBLES, cnt_z = k_end - k_begin;
  if (ctx->d_dpub && ctx->dpub_cap >= b1) {
    prep->fill(ctx->d_rowcnt + 2 * b0, 0ull, 2 * (b1 - b0));
    prep->sentinel(ctx->d_dpub + b0 * (long long)IMG_DOUBLES, cnt_img);
  }
  prep->sentinel(invd + b0 * (long long)IMG_DOUBLES, cnt_img);
  prep->sentinel(ctx->d_zpub + k_begin, cnt_z);
  // (the counters of the merged bulk updates: factor_lower of a whole matrix large enough to have any)
  if (want_step && k_begin == 0 && ctx->d_headcnt && ctx->tune.merge_above > 0 && k_end > ctx->tune.merge_above && !prep->full()) {
    prep->fill(ctx->d_headcnt, 0ull, agp_context::HEADCNT_WORDS);
    ctx->headcnt_ready = true;
  }
  ctx->zpub_ready_n = k_end;
  ctx->img_ready = invd;
  return true;
}

void panel_fused_prepare(agp_context *ctx, hipStream_t s, double *invd, long long k_begin, long long k_end, bool want_step) {
  PrepArgs prep;
  if (panel_fused_plan(ctx, invd, k_begin, k_end, want_step, &prep)) launch_prep(s, prep);
}

// Panel phase of one outer block [K0, kend): for every NB-wide diagonal block
// POTRF, panel TRSM (with the fused forward substitution on y) and the update
// of the remaining columns of the outer block.  Everything on stream s.
static void panel_phase(agp_context *ctx, hipStream_t s, double *A, long long n, long long lda, double *invd,
                        double *y, long long K0, long long kend, FactorTimers *timers, bool step_mode = false,
                        long long mark_after = -1, hipEvent_t mark_event = nullptr) {
  // step_mode (the chain-bound tail, factor_lower): ONE launch per panel.  The first panel is a plain fused launch;
  // every later one is panel_fused_kernel<true> - the previous panel's update of this panel's 128 columns on the
  // critical workgroups - plus trailing workgroups that apply the previous panel to everything further right while
  // this panel's POTRF runs.  No update launches, nothing leaves stream s.
  // Otherwise, while more than INNER_LEFT_ABOVE rows remain: left-looking inside the outer block - panel k is brought
  // up to date with the panels [K0, k) of this outer block in ONE product of depth k - K0 just before it is factored,
  // instead of every panel updating all later columns of the outer block with depth 128 (same flop, a third of the C
  // traffic: 35.2 -> 35.0 ms at N = 16384; where the chain is the critical path the deeper product costs 1-3 %).
  if (kend != n) step_mode = false;  // (a step launch updates ALL columns right of its panel)
  const bool inner_left = !step_mode && (n - K0) > INNER_LEFT_ABOVE;
  // The consumers of the fused kernel hold their slots for the whole POTRF (~30 us): while the bulk update fills the
  // chip that costs it more than the saved launch (measured: 43.3 -> 41.8 TFLOP/s), so the fused kernel takes over
  // where the panel chain is the critical path
  const bool fused = ctx->tune.panel_fused && ctx->d_zpub && ctx->zpub_ready_n >= kend && ctx->img_ready == invd &&
                     ((n - K0) <= FUSED_BELOW || step_mode);
  for (long long k = K0; k < kend; k += NB) {
    const int nbk = (int)((n - k < NB) ? n - k : NB);
    if (mark_event && k == mark_after + NB) (void)hipEventRecord(mark_event, s);  // everything up to panel mark_after is enqueued
    if (inner_left && k > K0) {
      const double *P = A + K0 * lda + k;  // rows k.., columns K0..k
      timed_gemm(s, timers, A + k * lda + k, lda, P, P, n - k, nbk, k - K0, false);
    }
    const long long below = n - (k + nbk);
    if (fused) {
      PotrfArgs pa;
      pa.A = A; pa.lda = lda; pa.k0 = k; pa.nbk = nbk;
      pa.img = invd + (k / NB) * (long long)IMG_DOUBLES;
      pa.y = y ? y + k : nullptr;
      pa.flags = ctx->d_flags; pa.scalars = ctx->d_scalars;
      pa.zpub = y ? ctx->d_zpub + k : nullptr;
      pa.below = below > 0 ? below : 0;
      if (step_mode && k > K0) {
        // the panel before this one (columns k - 128 .. k - 1) has not been applied to these columns - nor to anything
        // right of them - yet
        pa.dpub = ctx->d_dpub + (k / NB) * (long long)IMG_DOUBLES;
        unsigned grid = (unsigned)(1 + UPD_BLOCKS + (pa.below + 63) / 64);
        if (pa.below > 0) {
          const long long nt = (pa.below + 63) / 64;  // 64-row blocks below = tile rows of the trailing triangle
          pa.trail_first = grid;
          pa.rowcnt = ctx->d_rowcnt + (k + NB) / 64;
          pa.rowcnt_expect = 2ull * (unsigned long long)((k - K0) / NB);
          const long long tiles = nt * (nt + 1) / 2 + 2 * nt;
          // Workgroups go round-robin over the XCDs and, inside one, to its CUs in turn: the first `cus` of a launch get a
          // CU each, number cus + i lands next to number i (scripts/microbench/hwid_probe.hip).  The critical workgroups -
          // the factoring one, the nine that update its block, the row workgroups - keep their CUs to themselves:
          // workgroups cus .. cus + (their number) are PLACEHOLDERS that idle until the image is complete (next to a
          // trailing workgroup the POTRF takes 40-47 us instead of 27-30 and the diagonal block arrives after 10-13 us
          // instead of 6).  The trailing workgroups take the tiles worker, worker + workers, ...: as many as fit on the
          // CUs left, not one per tile (2600 starts and exits per launch at 4000 remaining rows cost ~10 %).  If the
          // dispatch order ever differs this costs idle slots and nothing else.
          const long long slots = step_slots(ctx), first_round = ctx->cus, ncrit = grid;
          long long workers = tiles, nhold = 0;
          if (ncrit + workers > first_round && ncrit < first_round) {
            // all critical workgroups while the trailing update is short; from ~2900 remaining rows on it is what the
            // launch takes and needs the slots: only the factoring workgroup and the nine that feed it
            nhold = tiles <= 1200 ? ncrit : 1 + UPD_BLOCKS;
            const long long cap = slots - ncrit - nhold;
            if (workers > cap) workers = cap;
            if (ncrit + workers <= first_round) nhold = 0;
          } else if (workers > slots - ncrit) {
            workers = slots - ncrit;
          }
          // (the row tiles come first in the tile order and every one needs a workgroup of its own up front: the row
          // workgroups wait for them)
          if (workers < 2 * nt) workers = 2 * nt < tiles ? 2 * nt : tiles;
          pa.trail_tiles = tiles;
          pa.trail_workers = workers;
          grid += (unsigned)workers;
          if (nhold > 0) {
            pa.hold_index = (unsigned)first_round;
            pa.hold_count = (unsigned)nhold;
            grid += (unsigned)nhold;
          }
        }
        hipLaunchKernelGGL(panel_fused_kernel<true>, dim3(grid), dim3(256), 0, s, pa);
      } else {
        hipLaunchKernelGGL(panel_fused_kernel<false>, dim3((unsigned)(1 + (pa.below + 63) / 64)), dim3(256), 0, s, pa);
      }
      if (below <= 0) continue;
    } else {
      launch_potrf(s, A, lda, k, nbk, invd, y, ctx->d_flags, ctx->d_scalars);
      if (below <= 0) continue;
      TrsmArgs t;
      t.img = invd + (k / NB) * (long long)IMG_DOUBLES;
      t.nbk = nbk;
      t.Y = A + k * lda + (k + nbk);
      t.stride_m = lda; t.stride_n = 1;
      t.ncols = below;
      t.z = y ? y + k : nullptr;
      t.yrest = y ? y + k + nbk : nullptr;
      t.batch_img = t.batch_Y = 0; t.n_total = 0;
      const unsigned grid = (unsigned)((below + 63) / 64);
      if (y) hipLaunchKernelGGL((trsm_micro_kernel<false, true>), dim3(grid), dim3(256), 0, s, t);
      else hipLaunchKernelGGL((trsm_micro_kernel<false, false>), dim3(grid), dim3(256), 0, s, t);
    }
    const long long width = kend - (k + nbk);
    if (width > 0 && !inner_left && !(step_mode && fused)) {
      const double *P = A + k * lda + (k + nbk);
      timed_gemm(s, timers, A + (k + nbk) * lda + (k + nbk), lda, P, P, below, width, nbk, false);
    }
  }
}

// The gate of a merged bulk update (factor_lower): ONE wave on the chain stream waits until the tiles of the next block
// column - the first workgroups of the bulk launch that runs on the other stream - have all counted themselves.  The
// kernels behind it on the stream (the next panel phase) then start while the rest of the bulk launch is still running;
// their start is the ACQUIRE that pairs with the tiles' RELEASE.  The wave holds no LDS and four registers; it gives up
// after 2 s like every hand-over of this library (flags[2] -> AGP_ERR_HIP, and the fit is repeated without merged launches).
__global__ __launch_bounds__(64) void head_gate_kernel(const unsigned long long *done, unsigned long long expect, int *flags) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int spin = 1; __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect; ++spin) {
      if ((spin & 63) == 0 && poll_expired(t0, flags)) break;
      __builtin_amdgcn_s_sleep(16);
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
}

void launch_head_gate(hipStream_t s, const unsigned long long *done, unsigned long long expect, int *flags) {
  hipLaunchKernelGGL(head_gate_kernel, dim3(1), dim3(64), 0, s, done, expect, flags);
}

// Host-side wait for an event WITHOUT parking a stream at a hipStreamWaitEvent (section 8: a stream that sits at an
// unsatisfied wait slows the launches of the others).  Polls with a pause instruction between queries (the core is
// shared with the launch threads of the other ranks of a box); true only if the event completed - any error ends the
// wait and the caller skips what depended on it.
static bool host_wait_event(hipEvent_t ev) {
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return true;
    if (e != hipErrorNotReady) { (void)hipGetLastError(); return false; }
    for (int i = 0; i < 32; ++i) __builtin_ia32_pause();
  }
}

static bool step_ready(agp_context *ctx, const double *invd, long long kend) {
  return ctx->tune.panel_fused && ctx->d_dpub && ctx->dpub_cap * NB >= kend && ctx->d_rowcnt && ctx->d_zpub &&
         ctx->img_ready == invd && ctx->zpub_ready_n >= kend;
}

void panel_phase_public(agp_context *ctx, hipStream_t s, double *A, long long n, long long lda, double *img,
                        double *y, long long K0, long long kend) {
  // (one block column of a sharded fit, or a probe: the block's own panels as step launches - one launch per panel, no
  // update launches - while the second image and the counters, which are indexed by the GLOBAL block number, stay small)
  const bool step = kend == n && kend <= 65536 && step_fits(ctx, kend - K0);
  panel_fused_prepare(ctx, s, img, K0, kend, step);
  panel_phase(ctx, s, A, n, lda, img, y, K0, kend, nullptr, step && step_ready(ctx, img, kend));
  ctx->img_ready = nullptr;
}

// X (nrows x w, ld) <- X L^-T against an ALREADY FACTORED w x w diagonal block (w <= 512) given by its lower
// triangle Lkk (leading dimension ldl) and the tile images of its 128 x 128 diagonal sub-blocks; yrows (optional)
// receives the fused forward substitution y -= X z.  The panel TRSM of the row-block-sharded fit (shard.h): every
// rank runs it on its own rows against the diagonal block the owner broadcast.  Right-looking over the sub-blocks:
// the same trsm_micro_kernel / MFMA update launches as panel_phase, minus the POTRFs.
void trsm_rows_wide(hipStream_t s, double *X, long long ld, long long nrows, long long w, const double *Lkk,
                    long long ldl, const double *img, const double *z, double *yrows) {
  if (nrows <= 0 || w <= 0) return;
  for (long long c = 0; c < w; c += NB) {
    const int nbk = (int)((w - c < NB) ? w - c : NB);
    TrsmArgs t;
    t.img = img + (c / NB) * (long long)IMG_DOUBLES;
    t.nbk = nbk;
    t.Y = X + c * ld;
    t.stride_m = ld; t.stride_n = 1;
    t.ncols = nrows;
    t.z = (z && yrows) ? z + c : nullptr;
    t.yrest = (z && yrows) ? yrows : nullptr;
    t.batch_img = t.batch_Y = 0; t.n_total = 0;
    const unsigned grid = (unsigned)((nrows + 63) / 64);
    if (t.z) hipLaunchKernelGGL((trsm_micro_kernel<false, true>), dim3(grid), dim3(256), 0, s, t);
    else hipLaunchKernelGGL((trsm_micro_kernel<false, false>), dim3(grid), dim3(256), 0, s, t);
    const long long rest = w - (c + nbk);
    if (rest > 0)  // X[:, c + nbk :] -= X[:, c : c + nbk] L[c + nbk :, c : c + nbk]^T
      launch_gemm_nt_sub(s, X + (c + nbk) * ld, ld, X + c * ld, ld, false, Lkk + (c + nbk) + c * ldl, ldl, false, nrows, rest,
                         nbk, false);
  }
}

// Outer block width as a function of the remaining (trailing) size.  Wide blocks (K = 512) keep
// the bulk update's C traffic off the HBM roofline and its MFMA efficiency up; narrow blocks
// shorten the serial panel chain per step.  With the current panel kernels the choice barely
// matters at N = 16384 (scripts/sweep_nbo.sh: every split at or below 4096 within 1 %).
static long long pick_nbo(long long remaining, long long override_nbo = 0, long long wide = 0) {
  if (override_nbo > 0) return override_nbo;
  // (mixed precision, bf16 x 3: at 150 TFLOP/s the fp64 C read-modify-write of a K = 512 update is 2.3 TB/s next to
  // 2.6 TB/s of operand strips beyond the L2 - a deeper outer block halves the former while many rows remain)
  if (wide > 512 && remaining > NBO_WIDE_ABOVE) return wide;
  if (remaining > NBO_512_ABOVE) return 512;
  if (remaining > NBO_256_ABOVE) return 256;
  return NB;
}

// Right-looking LL^T with one outer block of look-ahead on two streams:
//   stream  (high priority): panel phase P(j), then U1(j) = update of the
//            NEXT outer block column, then P(j + 1) ...
//   stream2: U2(j) = update of everything right of the next block column
//            (the MFMA-bound bulk), overlapping P(j + 1).
// Dependencies: U1(j), U2(j) after P(j) and after U2(j - 1).
void factor_lower(agp_context *ctx, double *A, long long n, long long lda, double *invd, double *y,
                  FactorTimers *timers) {
  hipStream_t sa = ctx->stream, sb = ctx->stream2;
  hipStream_t sb_prev = sb;
  bool have_u2 = false, p32_flip = false;
  long long K0 = 0, step_index = 0;
  const long long nbo_fixed = ctx->nbo_override;
  int variant = ctx->update_variant;
  // variant 5 (fp16 x 2 products, gemm_f16x2.hip): power-of-two row scales from the diagonal, BEFORE the first panel overwrites it
  const double *rs16 = nullptr, *irs16 = nullptr;
  if (variant == 5) {
    if (ctx->f16_scales && ctx->f16_scales_n >= n) {
      launch_f16x2_row_scales(sa, A, lda, n, ctx->f16_scales, ctx->f16_scales + ctx->f16_scales_n);
      rs16 = ctx->f16_scales;
      irs16 = ctx->f16_scales + ctx->f16_scales_n;
    } else {
      variant = 4;
    }
  }
  const long long nbo_wide = (variant == 4 || variant == 5) ? ctx->nbo_wide : ctx->tune.fp64_nbo;  // (fp64: AGP_FP64_NBO, measurement switch)
  long long kend = K0 + pick_nbo(n, nbo_fixed, nbo_wide);
  if (kend > n) kend = n;
  // (agp_fit_create has already planned and launched the fills together with its own: prep_external)
  if (!(ctx->prep_external && ctx->img_ready == invd && ctx->zpub_ready_n >= n)) panel_fused_prepare(ctx, sa, invd, 0, n, true);
  ctx->prep_external = false;
  // the chain-bound tail as one launch per panel on this stream (panel_phase step_mode) once few enough rows are left
  auto step_ok = [&](long long remaining) { return nbo_fixed == 0 && step_fits(ctx, remaining) && step_ready(ctx, invd, n); };
  const bool step_all = step_ok(n);  // small matrix: every panel
  if (step_all) kend = n;
  // A fit that is ALL step launches (n <= 4608) never reaches the two-stream tail below, where the wide diagonal blocks
  // of the back substitution get inverted on the idle second stream: here the host marks the panel after which all but
  // the last wide block are final, and - the stream must not sit at a hipStreamWaitEvent next to the chain (section 8) -
  // hands the inversion to the second stream itself once that mark has passed; the last four panels run meanwhile.
  const bool early_inv = step_all && ctx->bs_W && ctx->bs_done == 0 && ctx->ev_inv && ctx->stream2 && ctx->bs_BW > 0 &&
                         n % ctx->bs_BW == 0 && n / ctx->bs_BW >= 2;
  panel_phase(ctx, sa, A, n, lda, invd, y, K0, kend, timers, step_all, early_inv ? (n / ctx->bs_BW - 1) * ctx->bs_BW - NB : -1,
              early_inv ? ctx->ev_c : nullptr);
  if (early_inv) {
    const long long BW = ctx->bs_BW, done = n / BW - 1;
    if (host_wait_event(ctx->ev_c)) {  // (otherwise bs_done stays 0: the substitution inverts behind the factorisation)
      hipStream_t si = ctx->stream2;
      launch_set_identity_batched(si, ctx->bs_W, BW, BW * BW, BW, done);
      forward_solve_mat_batched(si, A, BW * (lda + 1), BW, lda, invd, (BW / NB) * (long long)IMG_DOUBLES, ctx->bs_W, BW * BW, BW, BW,
                                /*rhs_lower=*/true, done);
      (void)hipEventRecord(ctx->ev_inv, si);
      ctx->bs_done = done;
    }
  }
  while (kend < n) {
    long long next_end = kend + pick_nbo(n - kend, nbo_fixed, nbo_wide);
    if (next_end > n) next_end = n;
    // The very end runs on ONE stream: the last outer block spans all remaining columns - as step launches (U1 below
    // becomes the hand-over update of the whole trailing matrix), or, where those are off, with <= SINGLE_BELOW rows left:
    // there the bulk updates are 10-20 us launches, less than the ~10 us of event record / wait packets that hand each
    // of them to the second stream and back
    const long long K = kend - K0;
    const bool step = step_ok(n - kend);
    if (step || (nbo_fixed == 0 && n - kend <= SINGLE_BELOW)) next_end = n;
    if (ctx->bs_W && ctx->bs_done == 0 && next_end == n && ctx->ev_inv && ctx->stream2) {
      // Last step: everything left of kend is final and the second stream has nothing more to do - it inverts the wide
      // diagonal blocks the backward substitution of the fit will need (all but the last ones), off the chain.
      const long long BW = ctx->bs_BW, done = kend / BW;
      if (done > 0) {
        hipStream_t si = ctx->stream2;
        (void)hipEventRecord(ctx->ev_c, sa);  // columns < kend final
        (void)hipStreamWaitEvent(si, ctx->ev_c, 0);
        if (have_u2) (void)hipStreamWaitEvent(si, ctx->ev_b, 0);
        launch_set_identity_batched(si, ctx->bs_W, BW, BW * BW, BW, done);
        forward_solve_mat_batched(si, A, BW * (lda + 1), BW, lda, invd, (BW / NB) * (long long)IMG_DOUBLES, ctx->bs_W, BW * BW, BW, BW,
                                  /*rhs_lower=*/true, done);
        (void)hipEventRecord(ctx->ev_inv, si);
        ctx->bs_done = done;
      }
    }
    const double *P = A + K0 * lda + kend;  // panel rows kend.., columns K0..kend
    // mixed precision: ONE fp32 copy of the panel for all the tiles of U1 and of the bulk update (two alternating
    // buffers: the bulk update of step j still reads its copy while step j + 1 makes the next one; the copy of step
    // j + 2 is written after U2(j) has been waited for below).  Without it every tile rounds the fp64 operands itself
    // while it stages them - twice the operand traffic of the fp32 kernel, which is what it waits for.
    const float *P32 = nullptr;
    long long ld32 = 0;
    const unsigned short *P16 = nullptr;  // variant 4: the panel as three bf16 planes (gemm_bf16x3.hip), same two alternating buffers
    if (variant == 3 && ctx->p32 && (n - kend) >= U1_F32_ABOVE) {
      ld32 = (n - kend + 7) / 8 * 8;
      if (ld32 % 512 == 0) ld32 += 8;
      if (sizeof(float) * (size_t)ld32 * (size_t)K * 2 <= ctx->p32_bytes) {
        float *dst = ctx->p32 + (size_t)(p32_flip ? 1 : 0) * (ctx->p32_bytes / sizeof(float) / 2);
        p32_flip = !p32_flip;
        launch_convert_panel_f32(sa, P, lda, n - kend, K, dst, ld32);
        P32 = dst;
      }
    } else if (variant == 4 && ctx->p32 && (n - kend) >= U1_F32_ABOVE && K % 32 == 0 && 2 * bf16x3_bytes(n - kend, K) <= ctx->p32_bytes) {
      unsigned short *dst = reinterpret_cast<unsigned short *>(ctx->p32) + (size_t)(p32_flip ? 1 : 0) * (ctx->p32_bytes / sizeof(unsigned short) / 2);
      p32_flip = !p32_flip;
      launch_convert_panel_bf16x3(sa, P, lda, n - kend, K, dst);
      P16 = dst;
    } else if (variant == 5 && ctx->p32 && (n - kend) >= U1_F32_ABOVE && f16x2_depth_ok(K) && 2 * f16x2_bytes(n - kend, K) <= ctx->p32_bytes) {
      unsigned short *dst = reinterpret_cast<unsigned short *>(ctx->p32) + (size_t)(p32_flip ? 1 : 0) * (ctx->p32_bytes / sizeof(unsigned short) / 2);
      p32_flip = !p32_flip;
      launch_convert_panel_f16x2(sa, P, lda, n - kend, K, rs16 + kend, dst);
      P16 = dst;
    }
    (void)hipEventRecord(ctx->ev_a, sa);                       // P(j) done
    // Bulk-bound phase, fp64: U1(j) and U2(j) are ONE launch on the bulk stream - the whole trailing matrix, the tiles of
    // the next block column FIRST (at the bulk kernel's efficiency: as a launch of its own on the chain stream U1 ran
    // 64 x 64 tiles at MfmaUtil 0.38 next to the bulk update, waited for U2(j - 1) through an event, and sat on the
    // critical path of P(j + 1)) and counted; the chain stream carries a one-wave gate kernel in U1's place, which lets
    // P(j + 1) start when the count is complete, ~one tile time into the launch.  The chain stream never waits for ev_b here:
    // the launch is stream-ordered behind U2(j - 1), and its head is what P(j + 1) needs.
    const bool merged = variant <= 0 && nbo_fixed == 0 && !step && next_end < n && ctx->headcnt_ready && ctx->tune.merge_above > 0 &&
                        (n - kend) > ctx->tune.merge_above && step_index < agp_context::HEADCNT_WORDS && (next_end - kend) % NB == 0;
    if (merged) {
      sb = (ctx->stream_masked && (n - kend) <= MASK_BELOW) ? ctx->stream_masked : ctx->stream2;
      if (sb != sb_prev && have_u2) (void)hipStreamWaitEvent(sb, ctx->ev_b, 0);
      sb_prev = sb;
      (void)hipStreamWaitEvent(sb, ctx->ev_a, 0);
      BulkTiming bt;
      const bool timed = timers && timers->ev && timers->used + 2 <= timers->n_ev;
      if (timed) { bt.e0 = timers->ev[timers->used]; bt.e1 = timers->ev[timers->used + 1]; }
      long long head_tiles = 0;
      unsigned long long *cnt = ctx->d_headcnt + step_index;
      launch_trailing_update_merged(sb, A + kend * lda + kend, lda, P, lda, n - kend, K, (int)((next_end - kend) / NB), cnt, &head_tiles,
                                    timed ? &bt : nullptr);
      if (timed && bt.flops > 0.) {
        timers->flops[timers->used / 2] = bt.flops;
        timers->used += 2;
      }
      (void)hipEventRecord(ctx->ev_b, sb);
      have_u2 = true;
      launch_head_gate(sa, cnt, (unsigned long long)head_tiles, ctx->d_flags);
      panel_phase(ctx, sa, A, n, lda, invd, y, kend, next_end, timers, false);
      K0 = kend;
      kend = next_end;
      ++step_index;
      continue;
    }
    ++step_index;
    // U2(j - 1) must be done before anything of step j touches the next block column
    if (have_u2) (void)hipStreamWaitEvent(sa, ctx->ev_b, 0);
    // U1: block column [kend, next_end), all rows below its diagonal
    if (P16 && variant == 5) {
      // mixed precision, fp16 x 2: U1 and the bulk update from the planes of this step's panel
      launch_update_f16x2(sa, A + kend * lda + kend, lda, P16, n - kend, 0, 0, irs16 + kend, n - kend, next_end - kend, K);
    } else if (P16) {
      // mixed precision, bf16 x 3: U1 and the bulk update from the planes of this step's panel
      launch_update_bf16x3(sa, A + kend * lda + kend, lda, P16, n - kend, 0, 0, n - kend, next_end - kend, K);
    } else if ((variant == 3 || variant == 4 || variant == 5) && (n - kend) >= U1_F32_ABOVE) {
      // mixed precision: U1 on the fp32 MFMA path like the bulk update (products of fp32-rounded panels, fp64
      // subtraction) while the block column is tall enough for 128 x 128 tiles to fill the chip
      launch_update_f32(sa, A + kend * lda + kend, lda, P, P, lda, n - kend, next_end - kend, K, P32, P32, ld32);
    } else if (step && n - kend > 1536) {
      // hand-over to the step tail: the whole trailing matrix, on the chain stream, alone on the chip - the bulk kernel
      timed_gemm(sa, timers, A + kend * lda + kend, lda, P, P, n - kend, n - kend, K, true, variant >= 4 ? 3 : variant);
    } else {
      timed_gemm(sa, timers, A + kend * lda + kend, lda, P, P, n - kend, next_end - kend, K, false);
    }
    // Chain-bound phase (few rows left): a stream that sits at an UNSATISFIED hipStreamWaitEvent slows every dependent
    // launch of the other streams (measured, scripts/probe_chain.py: the panel chain takes 121 instead of 44 us per 128
    // columns while another stream waits on an event) - and here the bulk stream would wait for the panel chain all the
    // time.  So the host enqueues the next panel phase first and hands U2(j) to the bulk stream only once P(j) HAS
    // finished (hipEventQuery): the bulk stream is idle instead of blocked.  In the bulk-bound phase the host must run
    // far ahead, and the waits of the bulk stream are satisfied by the time it reaches them.
    // End phase: the bulk updates move to the CU-masked stream.  A bulk update that fills every CU (the 64 x 64-tile
    // kernel takes all 160 KB of LDS) keeps the panel kernels - 75-79 KB of LDS per workgroup - out until its grid has
    // drained: POTRF took 400-500 us instead of 30 (profiles/r02), panel chain and bulk update ran one after the
    // other.  With a few CUs per XCD left free the two overlap.
    sb = (ctx->stream_masked && (n - kend) <= MASK_BELOW) ? ctx->stream_masked : ctx->stream2;
    if (sb != sb_prev && have_u2) (void)hipStreamWaitEvent(sb, ctx->ev_b, 0);  // U2(j - 1) ran on the other bulk stream
    sb_prev = sb;
    const bool throttle = next_end < n && (n - kend) <= THROTTLE_BELOW;
    if (throttle) panel_phase(ctx, sa, A, n, lda, invd, y, kend, next_end, timers, step);
    if (next_end < n) {
      if (throttle) {
        if (!host_wait_event(ctx->ev_a)) (void)hipStreamWaitEvent(sb, ctx->ev_a, 0);
      } else {
        (void)hipStreamWaitEvent(sb, ctx->ev_a, 0);
      }
      const double *Q = A + K0 * lda + next_end;
      if (P16) {
        const long long M2 = n - next_end;
        const int ntr = (int)((M2 + 127) / 128);
        const long long tiles = (long long)ntr * (ntr + 1) / 2;
        long long olen = 0;
        const int *order = tiles >= 1024 ? bulk_tile_order(ntr, tiles, &olen) : nullptr;
        if (variant == 5)
          launch_update_f16x2(sb, A + next_end * lda + next_end, lda, P16, n - kend, next_end - kend, next_end - kend, irs16 + kend, M2, M2, K,
                              order, olen);
        else
          launch_update_bf16x3(sb, A + next_end * lda + next_end, lda, P16, n - kend, next_end - kend, next_end - kend, M2, M2, K, order, olen);
      } else
      timed_gemm(sb, timers, A + next_end * lda + next_end, lda, Q, Q, n - next_end, n - next_end, K, true, variant >= 4 ? 3 : variant,
                 P32 ? P32 + (next_end - kend) : nullptr, ld32);
      (void)hipEventRecord(ctx->ev_b, sb);
      have_u2 = true;
    } else {
      have_u2 = false;
    }
    // The step tail of a large fit, like the all-step fit above: the wide diagonal blocks that become final INSIDE the tail
    // (all but the last one) are inverted on the idle second stream under the tail's last four panels - the back
    // substitution of the fit used to invert them behind the factorisation (~0.15 ms of a 29 ms fit at N = 16384).
    const long long bw = ctx->bs_BW;
    const bool tail_inv = !throttle && step && next_end == n && ctx->bs_W && bw > 0 && ctx->ev_inv && ctx->stream2 && n % bw == 0 &&
                          ctx->bs_done > 0 && ctx->bs_done == kend / bw && n / bw - 1 > ctx->bs_done && (n / bw - 1) * bw - NB >= kend;
    if (!throttle) panel_phase(ctx, sa, A, n, lda, invd, y, kend, next_end, timers, step, tail_inv ? (n / bw - 1) * bw - NB : -1,
                               tail_inv ? ctx->ev_c : nullptr);
    if (tail_inv) {
      const long long first = ctx->bs_done, last = n / bw - 1, cnt = last - first;
      if (host_wait_event(ctx->ev_c)) {
        // (ev_inv already carries the first `first` inverses: same stream, so the new record covers both)
        hipStream_t si = ctx->stream2;
        launch_set_identity_batched(si, ctx->bs_W + first * bw * bw, bw, bw * bw, bw, cnt);
        forward_solve_mat_batched(si, A + first * bw * (lda + 1), bw * (lda + 1), bw, lda, invd + first * (bw / NB) * (long long)IMG_DOUBLES,
                                  (bw / NB) * (long long)IMG_DOUBLES, ctx->bs_W + first * bw * bw, bw * bw, bw, bw, /*rhs_lower=*/true, cnt);
        (void)hipEventRecord(ctx->ev_inv, si);
        ctx->bs_done = last;
      }
    }
    K0 = kend;
    kend = next_end;
  }
  // the panel stream ran last (its final panel depends on every update)
  ctx->img_ready = nullptr;
  ctx->headcnt_ready = false;
}

// `count` independent n x n factorisations in lock step (blockIdx.y = problem): the blocks of a sparse GP's A, the
// parameter vectors of agp_nll_batch, the fits of agp_fit_create_batch.  Problem b lives at A + b * stride_A (leading
// dimension lda), its tile images at invd + b * stride_invd, its right-hand side (optional, fused forward
// substitution) at y + b * stride_y; logsum[b] receives sum log L_ii; its flags at flags + b * stride_flags (0: shared).
// The blocking of factor_lower without its streams: outer blocks of 512 columns whose panels are left-looking (one
// product of depth <= 384 brings a panel up to date), then ONE trailing update of depth 512 for all problems - with a
// handful of problems that launch fills the chip, and the 27-30 us of a POTRF are shared by all of them.
// zpub (optional, with y): count x stride_zpub doubles; together with the tile images it must enter SENTINEL-filled - then
// every panel is ONE fused launch (panel_fused_kernel<false>, blockIdx.y = problem: the rows below are solved as the image
// appears) instead of POTRF -> TRSM.  For batches whose workgroups fit on the chip at once (batched_fused_fits): the row
// workgroups hold their slots for the whole POTRF.
bool batched_fused_fits(agp_context *ctx, long long n, long long count) {
  if (!ctx->tune.panel_fused || n <= NB) return false;
  const long long per_problem = 1 + (n - NB + 63) / 64;
  return count * per_problem <= step_slots(ctx);
}

void factor_lower_batched(hipStream_t s, double *A, long long stride_A, long long n, long long lda, double *invd,
                          long long stride_invd, double *y, long long stride_y, long long count, int *flags,
                          double *logsum, long long stride_flags, double *zpub, long long stride_zpub) {
  if (count <= 0 || n <= 0) return;
  const bool fused = zpub != nullptr && y != nullptr;
  for (long long K0 = 0; K0 < n; K0 += NBO) {
    const long long kend = (K0 + NBO < n) ? K0 + NBO : n;
    for (long long k = K0; k < kend; k += NB) {
      const int nbk = (int)((n - k < NB) ? n - k : NB);
      if (k > K0) {  // columns [k, k + nbk), rows k.. -= (rows k.. of the panels [K0, k)) (rows k..k+nbk of them)^T
        const double *P = A + K0 * lda + k;
        launch_gemm_nt_sub_batched(s, A + k * lda + k, lda, stride_A, P, lda, false, stride_A, P, lda, false, stride_A, n - k, nbk,
                                   k - K0, true, count);
      }
      PotrfArgs p;
      p.A = A; p.lda = lda; p.k0 = k; p.nbk = nbk;
      p.img = invd + (k / NB) * (long long)IMG_DOUBLES;
      p.y = y ? y + k : nullptr;
      p.flags = flags; p.scalars = logsum;
      p.batch_A = stride_A; p.batch_img = stride_invd; p.batch_y = stride_y; p.batch_scalars = 1; p.batch_flags = stride_flags;
      const long long below = n - (k + nbk);
      if (fused) {
        p.zpub = zpub + k;
        p.batch_zpub = stride_zpub;
        p.below = below > 0 ? below : 0;
        hipLaunchKernelGGL(panel_fused_kernel<false>, dim3((unsigned)(1 + (p.below + 63) / 64), (unsigned)count), dim3(256), 0, s, p);
        continue;
      }
      hipLaunchKernelGGL(potrf_diag_kernel, dim3(1, (unsigned)count), dim3(256), 0, s, p);
      if (below <= 0) continue;
      TrsmArgs t;
      t.img = p.img;
      t.nbk = nbk;
      t.Y = A + k * lda + (k + nbk);
      t.stride_m = lda; t.stride_n = 1;
      t.ncols = below;
      t.z = y ? y + k : nullptr;
      t.yrest = y ? y + k + nbk : nullptr;
      t.batch_img = stride_invd; t.batch_Y = stride_A; t.n_total = 0; t.batch_z = stride_y;
      const dim3 grid((unsigned)((below + 63) / 64), (unsigned)count);
      if (y) hipLaunchKernelGGL((trsm_micro_kernel<false, true>), grid, dim3(256), 0, s, t);
      else hipLaunchKernelGGL((trsm_micro_kernel<false, false>), grid, dim3(256), 0, s, t);
    }
    const long long rest = n - kend;
    if (rest > 0) {  // everything right of the outer block -= (its rows below) (its rows below)^T, depth kend - K0
      const double *P = A + K0 * lda + kend;
      launch_gemm_nt_sub_batched(s, A + kend * lda + kend, lda, stride_A, P, lda, false, stride_A, P, lda, false, stride_A, rest, rest,
                                 kend - K0, true, count);
    }
  }
}
// The same with the two-stream look-ahead of factor_lower (without its end-game): the panels of outer block j + 1 run on
// the context's chain stream while the trailing update of everything right of it (depth 512, all problems: the launch
// that fills the chip) runs on the bulk stream.  For batches whose updates are long enough to hide the chain behind.
void factor_lower_batched_lookahead(agp_context *ctx, double *A, long long stride_A, long long n, long long lda, double *invd,
                                    long long stride_invd, double *y, long long stride_y, long long count, int *flags,
                                    double *logsum, long long stride_flags) {
  if (count <= 0 || n <= 0) return;
  hipStream_t sa = ctx->stream, sb = ctx->stream2;
  bool have_u2 = false;
  auto panels = [&](long long K0, long long kend) {
    for (long long k = K0; k < kend; k += NB) {
      const int nbk = (int)((n - k < NB) ? n - k : NB);
      if (k > K0) {
        const double *P = A + K0 * lda + k;
        launch_gemm_nt_sub_batched(sa, A + k * lda + k, lda, stride_A, P, lda, false, stride_A, P, lda, false, stride_A, n - k, nbk,
                                   k - K0, true, count);
      }
      PotrfArgs p;
      p.A = A; p.lda = lda; p.k0 = k; p.nbk = nbk;
      p.img = invd + (k / NB) * (long long)IMG_DOUBLES;
      p.y = y ? y + k : nullptr;
      p.flags = flags; p.scalars = logsum;
      p.batch_A = stride_A; p.batch_img = stride_invd; p.batch_y = stride_y; p.batch_scalars = 1; p.batch_flags = stride_flags;
      hipLaunchKernelGGL(potrf_diag_kernel, dim3(1, (unsigned)count), dim3(256), 0, sa, p);
      const long long below = n - (k + nbk);
      if (below <= 0) continue;
      TrsmArgs t;
      t.img = p.img;
      t.nbk = nbk;
      t.Y = A + k * lda + (k + nbk);
      t.stride_m = lda; t.stride_n = 1;
      t.ncols = below;
      t.z = y ? y + k : nullptr;
      t.yrest = y ? y + k + nbk : nullptr;
      t.batch_img = stride_invd; t.batch_Y = stride_A; t.n_total = 0; t.batch_z = stride_y;
      const dim3 grid((unsigned)((below + 63) / 64), (unsigned)count);
      if (y) hipLaunchKernelGGL((trsm_micro_kernel<false, true>), grid, dim3(256), 0, sa, t);
      else hipLaunchKernelGGL((trsm_micro_kernel<false, false>), grid, dim3(256), 0, sa, t);
    }
  };
  long long K0 = 0, kend = NBO < n ? NBO : n;
  panels(K0, kend);
  while (kend < n) {
    const long long next_end = (kend + NBO < n) ? kend + NBO : n, K = kend - K0;
    (void)hipEventRecord(ctx->ev_a, sa);                        // P(j) done
    if (have_u2) (void)hipStreamWaitEvent(sa, ctx->ev_b, 0);    // U2(j - 1) wrote the next block column too
    const double *P = A + K0 * lda + kend;                      // panel rows kend.., columns K0 .. kend
    launch_gemm_nt_sub_batched(sa, A + kend * lda + kend, lda, stride_A, P, lda, false, stride_A, P, lda, false, stride_A, n - kend,
                               next_end - kend, K, true, count);  // U1: the next block column
    if (next_end < n) {
      (void)hipStreamWaitEvent(sb, ctx->ev_a, 0);
      const double *Q = A + K0 * lda + next_end;
      launch_gemm_nt_sub_batched(sb, A + next_end * lda + next_end, lda, stride_A, Q, lda, false, stride_A, Q, lda, false, stride_A,
                                 n - next_end, n - next_end, K, true, count);  // U2: everything right of it
      (void)hipEventRecord(ctx->ev_b, sb);
      have_u2 = true;
    } else {
      have_u2 = false;
    }
    panels(kend, next_end);
    K0 = kend;
    kend = next_end;
  }
}
}  // namespace agp
