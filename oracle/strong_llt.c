/*
 * strong_llt.c — the "strong CPU" context baseline of bench.py (flavour 2): a blocked,
 * pthread-parallel, un-pivoted LL^T with an AVX2 / FMA register-blocked product kernel, own code.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY, like oracle.c: nothing under albatross_amd/ may use it.  It is NOT the
 * reference's algorithm - albatross factors with Eigen 3.3's unblocked, single-threaded, pivoted LDL^T
 * (eigen/serializable_ldlt.hpp:27), which oracle.c restates and bench.py's `cpu_baseline.value` times - it answers the
 * other question: what would a competent multi-core CPU implementation of the same fit do on this host?  Rounds 1-5 used
 * scipy's LAPACK for that and measured 49 GFLOP/s on a 64-core EPYC (a BLAS pool that does not scale inside the box's
 * container); this file is checked against oracle.c's orc_llt by tests/test_oracle_golden.py.
 *
 * Layout: column-major, lower triangle, like the rest of the oracle.  Right-looking over NB-wide block columns:
 *   1. diagonal block: the same routine one level down (nb = 32 panels, unblocked base), one thread;
 *   2. panel: rows below, X <- X L11^-T, row chunks over the threads (blocked the same way against the finished block);
 *   3. trailing update C -= X X^T on the lower tiles (TS x TS), tiles dealt dynamically (atomic counter), each tile a
 *      packed 8 x 6 micro-kernel product.
 * A persistent thread pool (one barrier per parallel region) runs stages 2 and 3.
 */
#include <immintrin.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define ORC_API __attribute__((visibility("default")))

enum { NR = 6, KC = 256, TS = 192 }; /* micro tile columns, depth of one packed pass = outer block width, trailing tile edge */
/* micro tile rows: 8 (two ymm, AVX2 + FMA) or 16 (two zmm, where the host has AVX-512F: Zen 4 / 5, Intel server parts);
 * chosen once per process */
static int MR = 8;
static int g_avx512 = -1;

/* C (mr x nr, ld) -= Ap (MR x k packed: MR doubles per k) * Bp (NR x k packed: NR doubles per k)^T */
static void micro_8x6(int64_t k, const double *Ap, const double *Bp, double *C, int64_t ld, int mr, int nr) {
  enum { MR = 8 };
  __m256d c[NR][2];
  for (int j = 0; j < NR; ++j) c[j][0] = c[j][1] = _mm256_setzero_pd();
  for (int64_t p = 0; p < k; ++p) {
    const __m256d a0 = _mm256_loadu_pd(Ap + MR * p), a1 = _mm256_loadu_pd(Ap + MR * p + 4);
    for (int j = 0; j < NR; ++j) {
      const __m256d b = _mm256_broadcast_sd(Bp + NR * p + j);
      c[j][0] = _mm256_fmadd_pd(a0, b, c[j][0]);
      c[j][1] = _mm256_fmadd_pd(a1, b, c[j][1]);
    }
  }
  if (mr == MR && nr == NR) {
    for (int j = 0; j < NR; ++j) {
      double *cj = C + j * ld;
      _mm256_storeu_pd(cj, _mm256_sub_pd(_mm256_loadu_pd(cj), c[j][0]));
      _mm256_storeu_pd(cj + 4, _mm256_sub_pd(_mm256_loadu_pd(cj + 4), c[j][1]));
    }
  } else {
    double t[NR][MR];
    for (int j = 0; j < NR; ++j) {
      _mm256_storeu_pd(t[j], c[j][0]);
      _mm256_storeu_pd(t[j] + 4, c[j][1]);
    }
    for (int j = 0; j < nr; ++j)
      for (int i = 0; i < mr; ++i) C[i + j * ld] -= t[j][i];
  }
}

/* the same with sixteen rows per micro tile (zmm) */
__attribute__((target("avx512f"))) static void micro_16x6(int64_t k, const double *Ap, const double *Bp, double *C, int64_t ld, int mr,
                                                         int nr) {
  __m512d c[NR][2];
  for (int j = 0; j < NR; ++j) c[j][0] = c[j][1] = _mm512_setzero_pd();
  for (int64_t p = 0; p < k; ++p) {
    const __m512d a0 = _mm512_loadu_pd(Ap + 16 * p), a1 = _mm512_loadu_pd(Ap + 16 * p + 8);
    for (int j = 0; j < NR; ++j) {
      const __m512d b = _mm512_set1_pd(Bp[NR * p + j]);
      c[j][0] = _mm512_fmadd_pd(a0, b, c[j][0]);
      c[j][1] = _mm512_fmadd_pd(a1, b, c[j][1]);
    }
  }
  if (mr == 16 && nr == NR) {
    for (int j = 0; j < NR; ++j) {
      double *cj = C + j * ld;
      _mm512_storeu_pd(cj, _mm512_sub_pd(_mm512_loadu_pd(cj), c[j][0]));
      _mm512_storeu_pd(cj + 8, _mm512_sub_pd(_mm512_loadu_pd(cj + 8), c[j][1]));
    }
  } else {
    double t[NR][16];
    for (int j = 0; j < NR; ++j) {
      _mm512_storeu_pd(t[j], c[j][0]);
      _mm512_storeu_pd(t[j] + 8, c[j][1]);
    }
    for (int j = 0; j < nr; ++j)
      for (int i = 0; i < mr; ++i) C[i + j * ld] -= t[j][i];
  }
}

/* rows [0, m) of X (ld), depth k, into R-row micro panels (R = MR or NR), zero padded */
static void pack_rows(const double *X, int64_t ld, int64_t m, int64_t k, int R, double *out) {
  for (int64_t i0 = 0; i0 < m; i0 += R) {
    const int r = (int)(m - i0 < R ? m - i0 : R);
    for (int64_t p = 0; p < k; ++p) {
      const double *src = X + i0 + p * ld;
      for (int i = 0; i < r; ++i) out[i] = src[i];
      for (int i = r; i < R; ++i) out[i] = 0.;
      out += R;
    }
  }
}

/* C (m x n, ldc) -= A (m x k, lda) B (n x k, ldb)^T; lower != 0: C is a diagonal tile, only micro tiles that touch the lower
 * triangle are computed.  ws: (m/MR+1) MR k + (n/NR+1) NR k doubles */
static void gemm_nt_sub(double *C, int64_t ldc, const double *A, int64_t lda, const double *B, int64_t ldb, int64_t m, int64_t n,
                        int64_t k, int lower, double *ws) {
  double *Ap = ws, *Bp = ws + ((m + MR - 1) / MR) * MR * k;
  pack_rows(A, lda, m, k, MR, Ap);
  pack_rows(B, ldb, n, k, NR, Bp);
  for (int64_t j0 = 0; j0 < n; j0 += NR) {
    const int nr = (int)(n - j0 < NR ? n - j0 : NR);
    for (int64_t i0 = 0; i0 < m; i0 += MR) {
      if (lower && i0 + MR <= j0) continue; /* strictly above the diagonal */
      const int mr = (int)(m - i0 < MR ? m - i0 : MR);
      if (MR == 16) micro_16x6(k, Ap + (i0 / MR) * MR * k, Bp + (j0 / NR) * NR * k, C + i0 + j0 * ldc, ldc, mr, nr);
      else micro_8x6(k, Ap + (i0 / MR) * MR * k, Bp + (j0 / NR) * NR * k, C + i0 + j0 * ldc, ldc, mr, nr);
    }
  }
}

/* the same product from panels packed ONCE per outer step (pack_rows of the whole panel): Ap = MR-row micro panels of the
 * tile's rows, Bp = NR-row micro panels of its columns */
static void gemm_packed(double *C, int64_t ldc, const double *Ap, const double *Bp, int64_t m, int64_t n, int64_t k, int lower) {
  for (int64_t j0 = 0; j0 < n; j0 += NR) {
    const int nr = (int)(n - j0 < NR ? n - j0 : NR);
    for (int64_t i0 = 0; i0 < m; i0 += MR) {
      if (lower && i0 + MR <= j0) continue;
      const int mr = (int)(m - i0 < MR ? m - i0 : MR);
      if (MR == 16) micro_16x6(k, Ap + (i0 / MR) * MR * k, Bp + (j0 / NR) * NR * k, C + i0 + j0 * ldc, ldc, mr, nr);
      else micro_8x6(k, Ap + (i0 / MR) * MR * k, Bp + (j0 / NR) * NR * k, C + i0 + j0 * ldc, ldc, mr, nr);
    }
  }
}

static int64_t potrf_unblocked(double *A, int64_t n, int64_t ld) {
  for (int64_t j = 0; j < n; ++j) {
    double *cj = A + j * ld;
    const double d = cj[j];
    if (!(d > 0.)) return j + 1;
    const double s = sqrt(d), inv = 1. / s;
    cj[j] = s;
    for (int64_t r = j + 1; r < n; ++r) cj[r] *= inv;
    for (int64_t c = j + 1; c < n; ++c) {
      double *cc = A + c * ld;
      const double t = cj[c];
      for (int64_t r = c; r < n; ++r) cc[r] -= cj[r] * t;
    }
  }
  return 0;
}

/* X (m x n, ld) <- X L^-T for a finished lower-triangular L (n x n, ldl): column blocks of 32 */
static void trsm_rows(double *X, int64_t ld, int64_t m, const double *L, int64_t ldl, int64_t n, double *ws) {
  for (int64_t c0 = 0; c0 < n; c0 += 32) {
    const int64_t w = n - c0 < 32 ? n - c0 : 32;
    for (int64_t c = c0; c < c0 + w; ++c) { /* the block's own columns, column by column */
      double *xc = X + c * ld;
      for (int64_t q = c0; q < c; ++q) {
        const double l = L[c + q * ldl], *xq = X + q * ld;
        for (int64_t i = 0; i < m; ++i) xc[i] -= xq[i] * l;
      }
      const double inv = 1. / L[c + c * ldl];
      for (int64_t i = 0; i < m; ++i) xc[i] *= inv;
    }
    const int64_t rest = n - (c0 + w);
    if (rest > 0) /* X[:, c0 + w :] -= X[:, c0 : c0 + w] L[c0 + w :, c0 : c0 + w]^T */
      gemm_nt_sub(X + (c0 + w) * ld, ld, X + c0 * ld, ld, L + (c0 + w) + c0 * ldl, ldl, m, rest, w, 0, ws);
  }
}

/* serial blocked LL^T of a small block (the diagonal block of one outer step) */
static int64_t potrf_small(double *A, int64_t n, int64_t ld, double *ws) {
  for (int64_t k = 0; k < n; k += 32) {
    const int64_t w = n - k < 32 ? n - k : 32;
    const int64_t bad = potrf_unblocked(A + k + k * ld, w, ld);
    if (bad) return k + bad;
    const int64_t below = n - (k + w);
    if (below <= 0) continue;
    trsm_rows(A + (k + w) + k * ld, ld, below, A + k + k * ld, ld, w, ws);
    const double *P = A + (k + w) + k * ld;
    gemm_nt_sub(A + (k + w) + (k + w) * ld, ld, P, ld, P, ld, below, below, w, 1, ws);
  }
  return 0;
}

/* ---- thread pool: every worker runs job(arg, tid) between two barriers ---- */
typedef struct {
  int threads;
  pthread_t *tid;
  pthread_barrier_t bar;
  void (*job)(void *, int);
  void *arg;
  int stop;
} pool_t;
typedef struct { pool_t *p; int id; } pool_arg;

static void *pool_main(void *v) {
  pool_arg *pa = (pool_arg *)v;
  pool_t *p = pa->p;
  for (;;) {
    pthread_barrier_wait(&p->bar);
    if (p->stop) break;
    p->job(p->arg, pa->id);
    pthread_barrier_wait(&p->bar);
  }
  return NULL;
}
static void pool_run(pool_t *p, void (*job)(void *, int), void *arg) {
  p->job = job;
  p->arg = arg;
  pthread_barrier_wait(&p->bar); /* release the workers */
  job(arg, 0);                   /* the caller is worker 0 */
  pthread_barrier_wait(&p->bar); /* all done */
}

typedef struct {
  double *A;
  int64_t n, ld, k, w; /* current block column [k, k + w) */
  int threads;
  size_t ws_per_thread;
  double *ws;
  double *packA, *packB; /* the panel of the current step in MR- / NR-row micro panels (whole tiles of TS rows each) */
  atomic_long next;
} step_t;

static void job_panel(void *v, int t) {
  step_t *s = (step_t *)v;
  const int64_t below = s->n - (s->k + s->w);
  const int64_t chunk = ((below + s->threads - 1) / s->threads + MR - 1) / MR * MR;
  const int64_t r0 = (int64_t)t * chunk, r1 = r0 + chunk < below ? r0 + chunk : below;
  if (r0 >= r1) return;
  trsm_rows(s->A + (s->k + s->w + r0) + s->k * s->ld, s->ld, r1 - r0, s->A + s->k + s->k * s->ld, s->ld, s->w,
            s->ws + (size_t)t * s->ws_per_thread);
}

/* pack the solved panel once for all the tiles that read it: tile row b (TS rows) -> packA + b TS w and packB + b TS w */
static void job_pack(void *v, int t) {
  step_t *s = (step_t *)v;
  const int64_t o = s->k + s->w, below = s->n - o, nt = (below + TS - 1) / TS;
  for (int64_t b = t; b < nt; b += s->threads) {
    const int64_t i0 = b * TS, m = below - i0 < TS ? below - i0 : TS;
    const double *P = s->A + (o + i0) + s->k * s->ld;
    pack_rows(P, s->ld, m, s->w, MR, s->packA + b * TS * s->w);
    pack_rows(P, s->ld, m, s->w, NR, s->packB + b * TS * s->w);
  }
}

static void job_update(void *v, int t) {
  step_t *s = (step_t *)v;
  const int64_t o = s->k + s->w, below = s->n - o, nt = (below + TS - 1) / TS;
  const long tiles = (long)(nt * (nt + 1) / 2);
  (void)t;
  for (;;) {
    long id = atomic_fetch_add(&s->next, 1);
    if (id >= tiles) break;
    /* tile id -> (bi, bj), bj <= bi, column-major over the lower tile triangle; the tall first columns go first */
    int64_t bj = 0;
    while (id >= nt - bj) { id -= (long)(nt - bj); ++bj; }
    const int64_t bi = bj + id;
    const int64_t i0 = bi * TS, j0 = bj * TS;
    const int64_t m = below - i0 < TS ? below - i0 : TS, nn = below - j0 < TS ? below - j0 : TS;
    gemm_packed(s->A + (o + i0) + (o + j0) * s->ld, s->ld, s->packA + bi * TS * s->w, s->packB + bj * TS * s->w, m, nn, s->w, bi == bj);
  }
}

/* A (n x n, ld, lower triangle) <- L with A = L L^T; 0 on success, k + 1 if pivot k is not positive */
ORC_API int orc_llt_blocked_isa(void) {  /* 512 / 256: the vector width the product kernel runs at on this host */
  if (g_avx512 < 0) {
    __builtin_cpu_init();
    g_avx512 = __builtin_cpu_supports("avx512f") ? 1 : 0;
    MR = g_avx512 ? 16 : 8;
  }
  return g_avx512 ? 512 : 256;
}

ORC_API int64_t orc_llt_blocked(double *A, int64_t n, int64_t ld, int threads) {
  if (threads < 1) threads = 1;
  (void)orc_llt_blocked_isa();
  pool_t pool;
  memset(&pool, 0, sizeof(pool));
  pool.threads = threads;
  pool.tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
  pool_arg *pa = (pool_arg *)malloc(sizeof(pool_arg) * (size_t)threads);
  pthread_barrier_init(&pool.bar, NULL, (unsigned)threads);
  for (int t = 1; t < threads; ++t) {
    pa[t].p = &pool;
    pa[t].id = t;
    pthread_create(&pool.tid[t], NULL, pool_main, &pa[t]);
  }
  step_t s;
  memset(&s, 0, sizeof(s));
  s.A = A; s.n = n; s.ld = ld; s.threads = threads;
  const int64_t rows_max = n / threads + 2 * TS + 64;
  s.ws_per_thread = (size_t)((rows_max / 16 + 2) * 16 + (TS / NR + 2) * NR + 2 * KC) * KC;
  s.ws = (double *)malloc(sizeof(double) * s.ws_per_thread * (size_t)threads);
  const size_t pack_doubles = (size_t)((n + TS - 1) / TS + 1) * TS * KC;
  s.packA = (double *)malloc(sizeof(double) * pack_doubles);
  s.packB = (double *)malloc(sizeof(double) * pack_doubles);
  int64_t bad = 0;
  for (int64_t k = 0; k < n && !bad; k += KC) {
    const int64_t w = n - k < KC ? n - k : KC;
    s.k = k; s.w = w;
    const int64_t b = potrf_small(A + k + k * ld, w, ld, s.ws);
    if (b) { bad = k + b; break; }
    if (n - (k + w) <= 0) break;
    pool_run(&pool, job_panel, &s);
    pool_run(&pool, job_pack, &s);
    atomic_store(&s.next, 0);
    pool_run(&pool, job_update, &s);
  }
  pool.stop = 1;
  pthread_barrier_wait(&pool.bar);
  for (int t = 1; t < threads; ++t) pthread_join(pool.tid[t], NULL);
  pthread_barrier_destroy(&pool.bar);
  free(s.ws); free(s.packA); free(s.packB); free(pa); free(pool.tid);
  return bad;
}

/* ---- the Gram matrix of the strong-CPU row: SquaredExponential(l, sigma) + IndependentNoise on row-major n x dim points,
 * LOWER triangle only (all orc_llt_blocked and the substitutions read), columns dealt cyclically to the threads - the CPU
 * counterpart of the library's fast path (csrc/gram.hip: gram_fast_kernel), not the reference's generic caller chain ---- */
typedef struct {
  const double *x;
  int64_t n, ld;
  int dim, threads;
  double inv_l2, sigma2, noise_var;
  double *K;
} gram_job_t;
typedef struct { gram_job_t *g; int id; } gram_arg_t;

static void *gram_se_worker(void *v) {
  gram_arg_t *a = (gram_arg_t *)v;
  const gram_job_t *g = a->g;
  for (int64_t j = a->id; j < g->n; j += g->threads) {
    const double *xj = g->x + j * g->dim;
    double *col = g->K + j * g->ld;
    for (int64_t i = j; i < g->n; ++i) {
      const double *xi = g->x + i * g->dim;
      double d2 = 0.;
      for (int d = 0; d < g->dim; ++d) {
        const double t = xi[d] - xj[d];
        d2 += t * t;
      }
      col[i] = g->sigma2 * exp(-d2 * g->inv_l2);
    }
    col[j] += g->noise_var;
  }
  return NULL;
}

ORC_API void orc_strong_gram_se(const double *x, int64_t n, int dim, double length_scale, double sigma, double noise_sigma, double *K,
                                int64_t ld, int threads) {
  if (threads < 1) threads = 1;
  gram_job_t g = {x, n, ld, dim, threads, 1. / (length_scale * length_scale), sigma * sigma, noise_sigma * noise_sigma, K};
  pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
  gram_arg_t *args = (gram_arg_t *)malloc(sizeof(gram_arg_t) * (size_t)threads);
  for (int t = 0; t < threads; ++t) {
    args[t].g = &g;
    args[t].id = t;
    pthread_create(&tid[t], NULL, gram_se_worker, &args[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
  free(args);
  free(tid);
}
