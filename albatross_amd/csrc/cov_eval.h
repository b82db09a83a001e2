// cov_eval.h — device-side evaluation of one k(x, y) for a composed
// covariance function given as a postfix agp_kernel_node program.
//
// Follows, term by term, include/albatross/src/covariance_functions/
//   radial.hpp:25-33,191-198,289-297,461-470   (radial kernels)
//   distance_metrics.hpp:30-90                  (metrics)
//   noise.hpp:37-43, nugget.hpp:40-48, polynomials.hpp:56-60,78-86
//   scaling_function.hpp:79-83, measurement.hpp:87-102
//   covariance_function.hpp:266-272,357-367     (sum, product short-circuit)
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/albatross_amd.h"

namespace agp {

// The program lives in device memory and is passed by pointer: every index
// into it is wave-uniform, so the nodes are fetched with scalar loads through
// the constant cache (a by-value copy would be demoted to a per-thread LDS
// array by the compiler because of the dynamic node index).
struct DevProgram {
  int n_nodes;
  int metric_mask;   // bit m set: some radial leaf uses metric m
  int uses_equality; // any INDEPENDENT_NOISE / NUGGET leaf
  int pad;
  agp_kernel_node nodes[AGP_MAX_KERNEL_NODES];
};

// One point as the pair evaluator sees it.  DIMP = padded dimension
// (1, 2, 3, 4 or 8); unused trailing coordinates are zero, which leaves every
// distance / dot product bit-identical.
template <int DIMP>
struct Point {
  double c[DIMP];
  double norm;                       // ||x||, only valid if metric_mask needs it
  double s[AGP_MAX_SCALE_COLUMNS];   // ScalingTerm values f(x)
  long long id;                      // equality id (only when ids are supplied)
};

// exp(-t) for t >= 0 (NaN in -> NaN out, t = +inf -> 0).  Every radial kernel of the reference ends in such an exp
// (radial.hpp:25-33,191-198,289-297,461-470) and the Gram kernels are VALU-bound on it: the library exp is ~35
// instructions per call, half of them v_mov of 64-bit literals feeding v_fmac.  This one is 22: Cody-Waite reduction
// r = -t - k ln2 (ln2 split so that k ln2_hi is exact), degree-13 Taylor polynomial in Horner form with the
// coefficients held in SCALAR registers (one v_fma_f64 per step; truncation 4e-18, rounding ~1 ulp), v_ldexp_f64.
// Accuracy: <= 1.5 ulp against the correctly rounded value (tests/test_gram_gpu.py::test_exp_neg_accuracy).
__device__ __forceinline__ double horner_step(double p, double r, double c) {
  double o;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(p), "v"(r), "s"(c));
  return o;
}

__device__ __forceinline__ double exp_neg(double t) {
  t = (t > 1100.) ? 1100. : t;  // exp(-1100) = 0 in fp64; keeps +inf out of the reduction, lets NaN through
  const double kf = __builtin_rint(t * -1.4426950408889634074);
  double r = __builtin_fma(kf, -6.93147180369123816490e-01, -t);
  r = __builtin_fma(kf, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;             // 1/13!
  p = horner_step(p, r, 2.0876756987868100e-09);  // 1/12!
  p = horner_step(p, r, 2.5052108385441720e-08);  // 1/11!
  p = horner_step(p, r, 2.7557319223985888e-07);  // 1/10!
  p = horner_step(p, r, 2.7557319223985893e-06);  // 1/9!
  p = horner_step(p, r, 2.4801587301587302e-05);  // 1/8!
  p = horner_step(p, r, 1.9841269841269841e-04);  // 1/7!
  p = horner_step(p, r, 1.3888888888888889e-03);  // 1/6!
  p = horner_step(p, r, 8.3333333333333332e-03);  // 1/5!
  p = horner_step(p, r, 4.1666666666666664e-02);  // 1/4!
  p = horner_step(p, r, 1.6666666666666666e-01);  // 1/3!
  p = horner_step(p, r, 0.5);
  p = horner_step(p, r, 1.0);
  p = horner_step(p, r, 1.0);
  return __builtin_ldexp(p, (int)kf);
}

// NT independent exp_neg in lock step: the Horner steps of the NT arguments are issued round-robin (volatile asm keeps the
// source order), so that a wave covers the latency of one dependent v_fma_f64 with the steps of the others instead of
// waiting - the compiler's scheduler leaves NT separate exp_neg calls one behind the other.  Same operations per
// argument as exp_neg: bit-identical values.
__device__ __forceinline__ double horner_step_ordered(double p, double r, double c) {
  double o;
  asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(p), "v"(r), "s"(c));
  return o;
}

template <int NT>
__device__ __forceinline__ void exp_neg_n(const double (&tin)[NT], double (&out)[NT]) {
  double kf[NT], r[NT], p[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const double t = (tin[i] > 1100.) ? 1100. : tin[i];
    kf[i] = __builtin_rint(t * -1.4426950408889634074);
    r[i] = __builtin_fma(kf[i], -6.93147180369123816490e-01, -t);
    r[i] = __builtin_fma(kf[i], -1.90821492927058770002e-10, r[i]);
    p[i] = 1.6059043836821613e-10;  // 1/13!
  }
#define AGP_EXPN_STEP(C)                                                \
  _Pragma("unroll") for (int i = 0; i < NT; ++i) p[i] = horner_step_ordered(p[i], r[i], C);
  AGP_EXPN_STEP(2.0876756987868100e-09)  // 1/12!
  AGP_EXPN_STEP(2.5052108385441720e-08)
  AGP_EXPN_STEP(2.7557319223985888e-07)
  AGP_EXPN_STEP(2.7557319223985893e-06)
  AGP_EXPN_STEP(2.4801587301587302e-05)
  AGP_EXPN_STEP(1.9841269841269841e-04)
  AGP_EXPN_STEP(1.3888888888888889e-03)
  AGP_EXPN_STEP(8.3333333333333332e-03)
  AGP_EXPN_STEP(4.1666666666666664e-02)
  AGP_EXPN_STEP(1.6666666666666666e-01)
  AGP_EXPN_STEP(0.5)
  AGP_EXPN_STEP(1.0)
  AGP_EXPN_STEP(1.0)
#undef AGP_EXPN_STEP
#pragma unroll
  for (int i = 0; i < NT; ++i) out[i] = __builtin_ldexp(p[i], (int)kf[i]);
}

// acos(x) for the angular metric (distance_metrics.hpp:64-90).  The library acos is 93 VALU instructions, 26 of them
// v_mov of literals; this one is ~45: branch-free argument reduction to z in [0, 1/4] (|x| >= 1/2: z = (1 - |x|) / 2,
// exact by Sterbenz, acos = 2 asin(sqrt z) or pi - that; |x| < 1/2: z = x^2, acos = pi/2 - asin x), asin(sqrt z) =
// sqrt z (1 + z R(z)) with a degree-12 polynomial R (Chebyshev interpolant of (asin(sqrt z) / sqrt z - 1) / z on
// [0, 1/4], relative error 5e-18; coefficients in SCALAR registers, one v_fma_f64 per step) and a correctly rounded
// sqrt.  <= 1.1 ulp against the correctly rounded value (tests/test_gram_gpu.py::test_acos_fast_accuracy against
// mpmath).  Inputs outside [-1, 1] and NaN give NaN like acos.
__device__ __forceinline__ double acos_fast(double x) {
  const double ax = fabs(x);
  const bool big = ax >= 0.5;
  const double z = big ? (1.0 - ax) * 0.5 : x * x;
  double p = 0.028757851367421566;
  p = horner_step(p, z, -0.014851887071247204);
  p = horner_step(p, z, 0.01740087944269402);
  p = horner_step(p, z, 0.005457506718640358);
  p = horner_step(p, z, 0.01032281435018578);
  p = horner_step(p, z, 0.011479177415184906);
  p = horner_step(p, z, 0.013971212973552933);
  p = horner_step(p, z, 0.017352392720869973);
  p = horner_step(p, z, 0.02237217294214989);
  p = horner_step(p, z, 0.030381944138531247);
  p = horner_step(p, z, 0.04464285714635543);
  p = horner_step(p, z, 0.07499999999998433);
  p = horner_step(p, z, 0.16666666666666669);
  const double r = z * p;
  const double s = sqrt(z);                      // NaN for |x| > 1
  const double t = __builtin_fma(s, r, s);       // asin(sqrt z)
  const double two_t = t + t;
  const double res_big = (x > 0.) ? two_t : 3.14159265358979311600e+00 - (two_t - 1.22464679914735317720e-16);
  const double res_small = 1.57079632679489655800e+00 - (x - (6.12323399573676603587e-17 - x * r));
  return big ? res_big : res_small;
}

// The evaluation stack is an 8-wide fp64 vector indexed by the wave-uniform
// stack pointer: the backend lowers that to VGPR-indexed moves (no scratch,
// no LDS), which a plain `double st[8]` does not get.
typedef double stack_t __attribute__((ext_vector_type(AGP_MAX_STACK)));

__device__ __forceinline__ double stack_get(const stack_t &st, int i) { return st[i]; }
__device__ __forceinline__ void stack_set(stack_t &st, int i, double v) { st[i] = v; }

template <int DIMP>
__device__ __forceinline__ double eval_pair(const DevProgram *__restrict__ Pp, const Point<DIMP> &x,
                                            const Point<DIMP> &y, bool swapped, bool have_ids,
                                            bool both_measurement) {
  // `swapped`: evaluate k(y, x) instead of k(x, y).  Every term except the
  // polynomial's left-to-right product is bitwise symmetric in its arguments.
  const DevProgram &P = *Pp;
  // ---- distances, once per pair per metric in use ----
  double d_euclid = 0., d_radial = 0., d_angular = 0.;
  if (P.metric_mask & (1 << AGP_METRIC_EUCLIDEAN)) {
    if (DIMP == 1) {
      d_euclid = fabs(x.c[0] - y.c[0]);  // distance_metrics.hpp:34-36
    } else {
      double s = 0.;
#pragma unroll
      for (int d = 0; d < DIMP; ++d) {
        const double t = x.c[d] - y.c[d];
        s += t * t;
      }
      d_euclid = sqrt(s);  // (x - y).norm()
    }
  }
  if (P.metric_mask & (1 << AGP_METRIC_RADIAL)) d_radial = fabs(x.norm - y.norm);
  if (P.metric_mask & (1 << AGP_METRIC_ANGULAR)) {
    double dot = 0.;
#pragma unroll
    for (int d = 0; d < DIMP; ++d) dot += x.c[d] * y.c[d];
    const double c = dot / (x.norm * y.norm);
    const double eps = 1e-16;  // EPSILON, distance_metrics.hpp:18
    d_angular = (c > 1. - eps) ? 0. : ((c < -1. + eps) ? M_PI : acos_fast(c));
  }
  bool equal = false;
  if (P.uses_equality) {
    if (have_ids) {
      equal = (x.id == y.id);
    } else {
      equal = true;
#pragma unroll
      for (int d = 0; d < DIMP; ++d) equal = equal && (x.c[d] == y.c[d]);
    }
  }

  stack_t st = (stack_t)(0.);
  int sp = 0;
  // bit i: stack slot i holds a DEFINED value.  A term gated to another pair of variant alternatives is
  // undefined for this pair (no _call_impl overload): a sum or product then keeps its other side alone
  // (covariance_function.hpp:266-294, 357-389) and an undefined final result is 0 (VariantForwarder).
  unsigned defined = 0u;
  for (int t = 0; t < P.n_nodes; ++t) {
    const agp_kernel_node &nd = P.nodes[t];
    const int op = nd.op;
    if (op <= AGP_OP_MATERN52) {
      const double dist = nd.metric == AGP_METRIC_EUCLIDEAN ? d_euclid
                          : (nd.metric == AGP_METRIC_RADIAL ? d_radial : d_angular);
      const double l = nd.params[0], sigma = nd.params[1];
      double v;
      if (l <= 0.) {
        v = 0.;
      } else if (op == AGP_OP_SQUARED_EXPONENTIAL) {
        const double q = dist / l;
        v = sigma * sigma * exp_neg(q * q);  // exp(-pow(d/l, 2))
      } else if (op == AGP_OP_EXPONENTIAL) {
        v = sigma * sigma * exp_neg(fabs(dist / l));
      } else if (op == AGP_OP_MATERN32) {
        const double q = sqrt(3.) * dist / l;
        v = sigma * sigma * (1 + q) * exp_neg(q);
      } else {
        const double q = sqrt(5.) * dist / l;
        v = sigma * sigma * (1 + q + q * q * (1. / 3.)) * exp_neg(q);
      }
      stack_set(st, sp, v);
      ++sp;
    } else if (op == AGP_OP_CONSTANT) {
      stack_set(st, sp, nd.params[0] * nd.params[0]);
      ++sp;
    } else if (op == AGP_OP_INDEPENDENT_NOISE || op == AGP_OP_NUGGET) {
      stack_set(st, sp, equal ? nd.params[0] * nd.params[0] : 0.);
      ++sp;
    } else if (op == AGP_OP_POLYNOMIAL) {
      double cov = 0., xp = 1., yp = 1.;
      for (int q = 0; q <= nd.order; ++q) {
        const double s = nd.params[q];
        cov += swapped ? s * s * yp * xp : s * s * xp * yp;  // sigma^2 pow(x,q) pow(y,q)
        xp *= x.c[0];
        yp *= y.c[0];
      }
      stack_set(st, sp, cov);
      ++sp;
    } else if (op == AGP_OP_SCALING) {
      double fx = x.s[0], fy = y.s[0];
#pragma unroll
      for (int k = 1; k < AGP_MAX_SCALE_COLUMNS; ++k) {
        fx = (nd.column == k) ? x.s[k] : fx;
        fy = (nd.column == k) ? y.s[k] : fy;
      }
      stack_set(st, sp, fx * fy);
      ++sp;
    } else if (op == AGP_OP_SUM) {
      const double r = stack_get(st, sp - 1), l = stack_get(st, sp - 2);
      const bool dl = (defined >> (sp - 2)) & 1u, dr = (defined >> (sp - 1)) & 1u;
      stack_set(st, sp - 2, l + r);  // an undefined side holds 0
      defined = (defined & ~(3u << (sp - 2))) | ((unsigned)(dl || dr) << (sp - 2));
      --sp;
    } else if (op == AGP_OP_PRODUCT) {
      const double r = stack_get(st, sp - 1), l = stack_get(st, sp - 2);
      const bool dl = (defined >> (sp - 2)) & 1u, dr = (defined >> (sp - 1)) & 1u;
      // both sides defined: lhs * rhs, rhs skipped when lhs == 0; one side only: that side; none: undefined
      stack_set(st, sp - 2, (dl && dr) ? ((l != 0.) ? l * r : l) : (dl ? l : (dr ? r : 0.)));
      defined = (defined & ~(3u << (sp - 2))) | ((unsigned)(dl || dr) << (sp - 2));
      --sp;
    } else if (op == AGP_OP_MEASUREMENT_ONLY) {
      if (!both_measurement) stack_set(st, sp - 1, 0.);
    } else if (op == AGP_OP_TYPE_PAIR) {  // VariantForwarder, callers.hpp:419-544: undefined pair of alternatives -> 0
      double tx = x.s[0], ty = y.s[0];
#pragma unroll
      for (int k = 1; k < AGP_MAX_SCALE_COLUMNS; ++k) {
        tx = (nd.column == k) ? x.s[k] : tx;
        ty = (nd.column == k) ? y.s[k] : ty;
      }
      const double a = nd.params[0], b = nd.params[1];
      if (!((tx == a && ty == b) || (tx == b && ty == a))) {
        stack_set(st, sp - 1, 0.);
        defined &= ~(1u << (sp - 1));
      }
    }
    if (op <= AGP_OP_SCALING) defined |= 1u << (sp - 1);  // a leaf was pushed: defined for every pair
  }
  return st[0];
}

// ---------------------------------------------------------------------------------------------------------------
// Sum-of-products form.  Most covariance functions in use are sums of products of leaves, possibly wrapped in
// MeasurementOnly (the temperature example: scaling * constant + noise + exponential<angular> * squared_exponential
// <radial>; sinc: polynomial + squared_exponential + measurement_only(noise)).  For those the postfix interpreter above
// - a VGPR-indexed evaluation stack walked node by node - is replaced by a flat loop over terms and factors, and the
// radial factors of one product share ONE exp: prod_i sigma_i^2 p_i(q_i) exp(-e_i) = (prod ...) exp(-sum_i e_i).
// Differences to the reference's operation sequence: the exponents of a product are summed before the exp, and
// distances are multiplied by 1/l instead of divided by l - a few ulp of the exponent argument (the parity bar of
// tests/test_gram_gpu.py holds with a margin).  The product's `lhs != 0` short circuit (covariance_function.hpp:
// 362-366) is kept in effect: a zero factor makes the term zero.  Trees that are not sums of products (a sum inside a
// product, variant alternatives) keep the interpreter.
// ---------------------------------------------------------------------------------------------------------------
constexpr int SOP_MAX_TERMS = 6;
constexpr int SOP_MAX_FACTORS = 4;

struct SopFactor {
  // op | metric << 8 | column << 16 | order << 24 in ONE word: the device reads a factor with one scalar load and one
  // wait (field-by-field loads behind the branches on `op` cost ~5 dependent scalar-cache round trips per factor)
  int packed;
  int pad;
  double a;     // radial: sigma^2;  constant / noise / nugget: sigma^2;  polynomial: sigma_0
  double b;     // radial: SE 1 / l, EXP 1 / l, M32 sqrt(3) / l, M52 sqrt(5) / l  (0 when l <= 0: the leaf is 0)
  double c, d, e;  // polynomial: sigma_1..sigma_3
  __host__ __device__ static int pack(int op, int metric, int column, int order) {
    return (op & 0xff) | ((metric & 0xff) << 8) | ((column & 0xff) << 16) | ((order & 0xff) << 24);
  }
};

struct SopTerm {
  int n_factors;
  int measurement_only;  // the term (or one of its factors) is wrapped in MeasurementOnly: 0 unless both are measurements
  SopFactor f[SOP_MAX_FACTORS];
};

struct SopProgram {
  int n_terms;
  int metric_mask;
  int uses_equality;
  int pad;
  SopTerm t[SOP_MAX_TERMS];
};

// NP pairs (xs[p], ys[p]) per walk of the program: the term / factor loop is wave-uniform scalar work (loads of the
// program from the kernel arguments, branches on the leaf kind) and costs as much as the arithmetic of one pair, so the
// Gram kernel amortises it over the 2 rows x 2 columns a thread has in hand.
template <int DIMP, int NP>
__device__ __forceinline__ void eval_sop_n(const SopProgram &P, const Point<DIMP> *const (&xs)[NP], const Point<DIMP> *const (&ys)[NP],
                                           const bool (&swapped)[NP], bool have_ids, bool both_measurement, double (&out)[NP]) {
  double d_euclid[NP], d_radial[NP], d_angular[NP];
  bool equal[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const Point<DIMP> &x = *xs[p], &y = *ys[p];
    d_euclid[p] = 0.; d_radial[p] = 0.; d_angular[p] = 0.;
    if (P.metric_mask & (1 << AGP_METRIC_EUCLIDEAN)) {
      if (DIMP == 1) {
        d_euclid[p] = fabs(x.c[0] - y.c[0]);
      } else {
        double s = 0.;
#pragma unroll
        for (int d = 0; d < DIMP; ++d) {
          const double t = x.c[d] - y.c[d];
          s += t * t;
        }
        d_euclid[p] = sqrt(s);
      }
    }
    if (P.metric_mask & (1 << AGP_METRIC_RADIAL)) d_radial[p] = fabs(x.norm - y.norm);
    if (P.metric_mask & (1 << AGP_METRIC_ANGULAR)) {
      double dot = 0.;
#pragma unroll
      for (int d = 0; d < DIMP; ++d) dot += x.c[d] * y.c[d];
      const double c = dot / (x.norm * y.norm);
      const double eps = 1e-16;  // EPSILON, distance_metrics.hpp:18
      d_angular[p] = (c > 1. - eps) ? 0. : ((c < -1. + eps) ? M_PI : acos_fast(c));
    }
    equal[p] = false;
    if (P.uses_equality) {
      if (have_ids) {
        equal[p] = (x.id == y.id);
      } else {
        bool e = true;
#pragma unroll
        for (int d = 0; d < DIMP; ++d) e = e && (x.c[d] == y.c[d]);
        equal[p] = e;
      }
    }
    out[p] = 0.;
  }
  for (int ti = 0; ti < P.n_terms; ++ti) {
    const SopTerm &T = P.t[ti];
    if (T.measurement_only && !both_measurement) continue;  // measurement.hpp:87-102
    double v[NP], expo[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) { v[p] = 1.; expo[p] = 0.; }
    bool any_exp = false;
    for (int fi = 0; fi < T.n_factors; ++fi) {
      // the whole factor up front (the asm pins the loads here: one wait instead of one per branch)
      const SopFactor &F = T.f[fi];
      int packed = F.packed;
      double fa = F.a, fb = F.b, fc = F.c, fd = F.d, fe = F.e;
      asm volatile("" : "+s"(packed), "+s"(fa), "+s"(fb), "+s"(fc), "+s"(fd), "+s"(fe));
      const int op = packed & 0xff, metric = (packed >> 8) & 0xff, column = (packed >> 16) & 0xff, order = (packed >> 24) & 0xff;
      // lhs * rhs with rhs skipped once lhs == 0 (covariance_function.hpp:362-366): 0 * inf stays 0
      if (op <= AGP_OP_MATERN52) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const double dist = metric == AGP_METRIC_EUCLIDEAN ? d_euclid[p] : (metric == AGP_METRIC_RADIAL ? d_radial[p] : d_angular[p]);
          const double q = dist * fb;
          double coef = fa;
          if (op == AGP_OP_SQUARED_EXPONENTIAL) expo[p] += q * q;
          else if (op == AGP_OP_EXPONENTIAL) expo[p] += fabs(q);
          else if (op == AGP_OP_MATERN32) { coef = coef * (1 + q); expo[p] += q; }
          else { coef = coef * (1 + q + q * q * (1. / 3.)); expo[p] += q; }
          const double f = (fb > 0.) ? coef : 0.;  // length_scale <= 0: the leaf is 0 (radial.hpp:26-28)
          v[p] = (v[p] != 0.) ? v[p] * f : v[p];
        }
        any_exp = true;
      } else if (op == AGP_OP_CONSTANT) {
#pragma unroll
        for (int p = 0; p < NP; ++p) v[p] = (v[p] != 0.) ? v[p] * fa : v[p];
      } else if (op == AGP_OP_INDEPENDENT_NOISE || op == AGP_OP_NUGGET) {
#pragma unroll
        for (int p = 0; p < NP; ++p) v[p] = (v[p] != 0.) ? v[p] * (equal[p] ? fa : 0.) : v[p];
      } else if (op == AGP_OP_SCALING) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          double fx = xs[p]->s[0], fy = ys[p]->s[0];
#pragma unroll
          for (int k = 1; k < AGP_MAX_SCALE_COLUMNS; ++k) {
            fx = (column == k) ? xs[p]->s[k] : fx;
            fy = (column == k) ? ys[p]->s[k] : fy;
          }
          v[p] = (v[p] != 0.) ? v[p] * (fx * fy) : v[p];
        }
      } else {  // AGP_OP_POLYNOMIAL, polynomials.hpp:78-86
        const double sg[4] = {fa, fc, fd, fe};
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          double cov = 0., xp = 1., yp = 1.;
          for (int q = 0; q <= order; ++q) {
            cov += swapped[p] ? sg[q] * sg[q] * yp * xp : sg[q] * sg[q] * xp * yp;
            xp *= xs[p]->c[0];
            yp *= ys[p]->c[0];
          }
          v[p] = (v[p] != 0.) ? v[p] * cov : v[p];
        }
      }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (any_exp) v[p] = (v[p] != 0.) ? v[p] * exp_neg(expo[p]) : v[p];
      out[p] += v[p];
    }
  }
}

template <int DIMP>
__device__ __forceinline__ double eval_sop(const SopProgram &P, const Point<DIMP> &x, const Point<DIMP> &y, bool swapped,
                                           bool have_ids, bool both_measurement) {
  const Point<DIMP> *const xs[1] = {&x};
  const Point<DIMP> *const ys[1] = {&y};
  const bool sw[1] = {swapped};
  double out[1];
  eval_sop_n<DIMP, 1>(P, xs, ys, sw, have_ids, both_measurement, out);
  return out[0];
}

}  // namespace agp
