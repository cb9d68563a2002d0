// rmp2_solve.h -- the resolve step  qdd = pinv(M) f  in fp64, one robot per lane, all in VGPRs.
//
// Reference: rmp.py:153-154 -- tf.linalg.pinv (SVD, cutoff 10*n*eps*sigma_max) of the fp64
// combined metric, then a mat-vec (quirk Q1).  M may be non-symmetric (JointLimitAvoidance,
// quirk Q2), indefinite (JointVelocityCap, Q4) or rank deficient (target-only sets, Q3).
//
//  * lu_solve:   Gaussian elimination with threshold partial pivoting.  Row swaps are rare
//                for the metrics RMP sets produce, so they sit behind a wave-uniform
//                `__any(need_swap)` branch; in the common case the elimination is pure
//                straight-line fp64 FMA on statically indexed registers.  A pivot column
//                whose largest entry is below 1e-11 * max|M| flags the lane "singular".
//  * pinv_solve: one-sided (Hestenes) Jacobi on the ROWS of [M | f]:  G M = W with mutually
//                orthogonal rows  =>  pinv(M) f = sum_i W_i^T (G f)_i / |W_i|^2 over the rows
//                with |W_i| > cutoff.  No V matrix is needed, so the whole iteration lives in
//                N*(N+1) fp64 registers.
#pragma once
#include "rmp2_device.h"

namespace rmp2 {

// Fast path: elimination WITHOUT row exchanges -- pure straight-line fp64 FMA on statically
// indexed registers (no select chains, no branches).  RMP metrics are sums of PSD pull-backs
// plus positive diagonal inertia terms, for which this is backward stable; the lane is
// flagged (return value) whenever that cannot be certified on the fly:
//   - a pivot is below 1e-11 * max|M|          (numerically singular  -> pseudo-inverse), or
//   - a multiplier exceeds 1e4 in magnitude    (possible element growth -> pivoted solve), or
//   - the input is all-zero / not finite.
// Flagged lanes are re-resolved by resolve_compact() below; x is then not used.
template <int N>
__device__ __forceinline__ bool lu_solve(double (&A)[N][N], double (&b)[N], double (&x)[N]) {
  double scale = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) scale = fmax(scale, fabs(A[i][j]));
  const double tiny = 1e-11 * scale;
  bool flagged = !(scale > 0.0) || !(scale < 1.7e308);
  double lmax = 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const bool bad = !(fabs(A[k][k]) > tiny);
    flagged = flagged || bad;
    const double inv = bad ? 0.0 : 1.0 / A[k][k];
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const double l = A[i][k] * inv;
      lmax = fmax(lmax, fabs(l));
#pragma unroll
      for (int j = k + 1; j < N; ++j) A[i][j] = fma(-l, A[k][j], A[i][j]);
      b[i] = fma(-l, b[k], b[i]);
    }
    A[k][k] = inv;  // keep the reciprocal pivot for the back substitution
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    double s = b[i];
#pragma unroll
    for (int j = i + 1; j < N; ++j) s = fma(-A[i][j], x[j], s);
    x[i] = s * A[i][i];
  }
  return flagged || !(lmax <= 1e4);
}

// x = pinv(M) f with TensorFlow's default cutoff; rows >= n_dof are padding (identity) and are
// excluded from sigma_max.  Returns the number of dropped singular values.
template <int N>
__device__ __forceinline__ int pinv_solve(double (&A)[N][N], double (&b)[N], int n_dof, double (&x)[N]) {
  for (int sweep = 0; sweep < 30; ++sweep) {
    bool rotated = false;
    // squared row norms: formed once per sweep, then carried through the rotations (|w_p'|^2 = |w_p|^2 - t <w_p, w_q>,
    // |w_q'|^2 = |w_q|^2 + t <w_p, w_q> for the rotation below): one dot product per pair instead of three
    double nrm[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      nrm[i] = 0.0;
#pragma unroll
      for (int j = 0; j < N; ++j) nrm[i] = fma(A[i][j], A[i][j], nrm[i]);
    }
#pragma unroll
    for (int p = 0; p < N - 1; ++p) {
#pragma unroll
      for (int q = p + 1; q < N; ++q) {
        const double al = nrm[p], be = nrm[q];
        double ga = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) ga = fma(A[p][j], A[q][j], ga);
        // converged pair: |<w_p, w_q>| <= 4 eps |w_p| |w_q|  (a tighter bound than eps can never be met
        // and only burns sweeps)
        const bool rot = (fabs(ga) > 1e-300) && (ga * ga > 1e-30 * (al * be));
        rotated = rotated || rot;
        // the rotation that makes the two rows orthogonal: zeta = (be - al) / (2 ga), t = sign(zeta) / (|zeta| + sqrt(1 +
        // zeta^2)), c = 1 / sqrt(1 + t^2), s = c t -- with numerator and denominator of t multiplied through by 2 |ga|:
        // one square root, one division and one reciprocal square root per rotation instead of three and three (the fp64
        // forms of those are ~25 instructions each: they were two thirds of the kernel that runs this for a whole fleet)
        const double dd = be - al;
        const double num = 2.0 * (dd == 0.0 ? fabs(ga) : (dd > 0.0 ? ga : -ga));
        const double t = num / (fabs(dd) + sqrt(fma(dd, dd, 4.0 * ga * ga)));
        double c = rsqrt(fma(t, t, 1.0)), s = c * t;
        c = rot ? c : 1.0;
        s = rot ? s : 0.0;
        const double tg = rot ? t * ga : 0.0;
        nrm[p] = al - tg;
        nrm[q] = be + tg;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const double wp = A[p][j], wq = A[q][j];
          A[p][j] = c * wp - s * wq;
          A[q][j] = s * wp + c * wq;
        }
        const double bp = b[p], bq = b[q];
        b[p] = c * bp - s * bq;
        b[q] = s * bp + c * bq;
      }
    }
    if (!__any(rotated)) break;
  }
  double s2[N], smax = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    s2[i] = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) s2[i] = fma(A[i][j], A[i][j], s2[i]);
    if (i < n_dof) smax = fmax(smax, sqrt(s2[i]));
  }
  const double cutoff = 10.0 * (double)n_dof * 2.220446049250313e-16 * smax;
  int dropped = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) x[j] = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const bool keep = sqrt(s2[i]) > cutoff;
    if (i < n_dof && !keep) ++dropped;
    const double coef = keep ? b[i] / s2[i] : 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] = fma(A[i][j], coef, x[j]);
  }
  return dropped;
}

// 2 x 2 systems: x = pinv(M) f in closed form (rmp.py:153-154 with TensorFlow's cutoff 10 * n * eps * sigma_max), for EVERY
// robot -- no elimination, no flag, no second pass.  sigma_1^2 = (|M|_F^2 + sqrt(|M|_F^4 - 4 det^2)) / 2 and
// sigma_2 = |det| / sigma_1 (no cancellation); both singular values kept: x = adj(M) f / det (differences formed with
// Kahan's fma trick); sigma_2 dropped: pinv(M) = M^T / sigma_1^2 up to a relative sigma_2 / sigma_1 <= 4.4e-15.
// Returns RMP2_STATUS_* bits.
__device__ __forceinline__ uint32_t pinv_solve_2x2(double a, double b, double c, double d, double f0, double f1,
                                                   double& x0, double& x1) {
  auto diff = [](double p, double q, double r, double s) {  // p q - r s, ~1.5 ulp
    const double w = r * s;
    const double e = fma(-r, s, w);
    return fma(p, q, -w) + e;
  };
  const double scale = fmax(fmax(fabs(a), fabs(b)), fmax(fabs(c), fabs(d)));
  if (!(scale < 1.7e308) || !(fabs(f0) < 1.7e308) || !(fabs(f1) < 1.7e308)) {  // NaN / Inf in: NaN out, as the reference's pinv
    x0 = x1 = __builtin_nan("");
    return RMP2_STATUS_NONFINITE | RMP2_STATUS_PINV_PATH;
  }
  if (!(scale > 0.0)) {
    x0 = x1 = 0.0;
    return RMP2_STATUS_RANK_DROP | RMP2_STATUS_PINV_PATH;
  }
  const double is = 1.0 / scale;
  a *= is, b *= is, c *= is, d *= is;  // entries in [-1, 1]: nothing below can overflow; x = pinv(M / s) f / s
  const double fro2 = fma(a, a, fma(b, b, fma(c, c, d * d)));
  const double det = diff(a, d, b, c);
  const double disc = sqrt(fmax(fma(fro2, fro2, -4.0 * det * det), 0.0));
  const double s1sq = 0.5 * (fro2 + disc);
  const double s1 = sqrt(s1sq);
  const double s2 = fabs(det) / s1;
  const double cutoff = 10.0 * 2.0 * 2.220446049250313e-16 * s1;
  if (s2 > cutoff) {
    const double id = is / det;
    x0 = diff(d, f0, b, f1) * id;
    x1 = diff(a, f1, c, f0) * id;
    return 0u;
  }
  const double iw = is / s1sq;
  x0 = fma(a, f0, c * f1) * iw;
  x1 = fma(b, f0, d * f1) * iw;
  return RMP2_STATUS_RANK_DROP | RMP2_STATUS_PINV_PATH;
}

// ---- rare path of the AUTO build (run-time indices, scratch memory, loops not unrolled) ----
// Gaussian elimination with partial pivoting on a COPY; returns false (x untouched) when a
// pivot column is below 1e-11 * max|M| -> the caller then takes the pseudo-inverse.
__device__ __forceinline__ bool lu_pivot_compact(const double* W, double* T, int n, double* x) {
  const int ld = n + 1;
  double scale = 0.0;
#pragma nounroll
  for (int i = 0; i < n; ++i)
#pragma nounroll
    for (int j = 0; j <= n; ++j) {
      T[i * ld + j] = W[i * ld + j];
      if (j < n) scale = fmax(scale, fabs(W[i * ld + j]));
    }
  if (!(scale > 0.0) || !(scale < 1.7e308)) return false;
  const double tiny = 1e-11 * scale;
#pragma nounroll
  for (int k = 0; k < n; ++k) {
    int p = k;
    double amax = fabs(T[k * ld + k]);
#pragma nounroll
    for (int i = k + 1; i < n; ++i)
      if (fabs(T[i * ld + k]) > amax) {
        amax = fabs(T[i * ld + k]);
        p = i;
      }
    if (!(amax > tiny)) return false;
    if (p != k) {
#pragma nounroll
      for (int j = k; j <= n; ++j) {
        const double u = T[k * ld + j];
        T[k * ld + j] = T[p * ld + j];
        T[p * ld + j] = u;
      }
    }
    const double inv = 1.0 / T[k * ld + k];
#pragma nounroll
    for (int i = k + 1; i < n; ++i) {
      const double l = T[i * ld + k] * inv;
#pragma nounroll
      for (int j = k + 1; j <= n; ++j) T[i * ld + j] = fma(-l, T[k * ld + j], T[i * ld + j]);
    }
  }
#pragma nounroll
  for (int i = n - 1; i >= 0; --i) {
    double s = T[i * ld + n];
#pragma nounroll
    for (int j = i + 1; j < n; ++j) s = fma(-T[i * ld + j], x[j], s);
    x[i] = s / T[i * ld + i];
  }
  return true;
}

// One-sided Jacobi pseudo-inverse, same algorithm as pinv_solve<N> on a flat [n x (n+1)] array.
__device__ __forceinline__ int pinv_solve_compact(double* W, int n, int n_dof, double* x) {
  const int ld = n + 1;
  for (int sweep = 0; sweep < 30; ++sweep) {
    bool rotated = false;
#pragma nounroll
    for (int p = 0; p < n - 1; ++p) {
#pragma nounroll
      for (int q = p + 1; q < n; ++q) {
        double al = 0.0, be = 0.0, ga = 0.0;
#pragma nounroll
        for (int j = 0; j < n; ++j) {
          const double wp = W[p * ld + j], wq = W[q * ld + j];
          al = fma(wp, wp, al);
          be = fma(wq, wq, be);
          ga = fma(wp, wq, ga);
        }
        if (!((fabs(ga) > 1e-300) && (fabs(ga) > 1e-15 * sqrt(al * be)))) continue;
        rotated = true;
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma nounroll
        for (int j = 0; j <= n; ++j) {
          const double wp = W[p * ld + j], wq = W[q * ld + j];
          W[p * ld + j] = c * wp - s * wq;
          W[q * ld + j] = s * wp + c * wq;
        }
      }
    }
    if (!rotated) break;
  }
  double smax = 0.0;
#pragma nounroll
  for (int i = 0; i < n_dof; ++i) {
    double s2 = 0.0;
#pragma nounroll
    for (int j = 0; j < n; ++j) s2 = fma(W[i * ld + j], W[i * ld + j], s2);
    smax = fmax(smax, sqrt(s2));
  }
  const double cutoff = 10.0 * (double)n_dof * 2.220446049250313e-16 * smax;
  int dropped = 0;
#pragma nounroll
  for (int j = 0; j < n; ++j) x[j] = 0.0;
#pragma nounroll
  for (int i = 0; i < n; ++i) {
    double s2 = 0.0;
#pragma nounroll
    for (int j = 0; j < n; ++j) s2 = fma(W[i * ld + j], W[i * ld + j], s2);
    const bool keep = sqrt(s2) > cutoff;
    if (i < n_dof && !keep) ++dropped;
    if (!keep) continue;
    const double coef = W[i * ld + n] / s2;
#pragma nounroll
    for (int j = 0; j < n; ++j) x[j] = fma(W[i * ld + j], coef, x[j]);
  }
  return dropped;
}

}  // namespace rmp2
