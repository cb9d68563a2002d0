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

// returns true when this lane's matrix is numerically singular (x is then not valid)
template <int N>
__device__ __forceinline__ bool lu_solve(double (&A)[N][N], double (&b)[N], double (&x)[N]) {
  double scale = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) scale = fmax(scale, fabs(A[i][j]));
  const double tiny = 1e-11 * scale;
  bool singular = !(scale > 0.0) || !(scale < 1.7e308);  // all-zero, Inf or NaN input
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double amax = fabs(A[k][k]);
#pragma unroll
    for (int i = k + 1; i < N; ++i) amax = fmax(amax, fabs(A[i][k]));
    const bool need_swap = fabs(A[k][k]) < 0.1 * amax;
    if (__any(need_swap)) {  // wave-uniform slow path
      bool done = !need_swap;
#pragma unroll
      for (int i = k + 1; i < N; ++i) {
        const bool sel = !done && (fabs(A[i][k]) == amax);
        done = done || sel;
#pragma unroll
        for (int j = k; j < N; ++j) {
          const double u = A[k][j], l = A[i][j];
          A[k][j] = sel ? l : u;
          A[i][j] = sel ? u : l;
        }
        const double u = b[k], l = b[i];
        b[k] = sel ? l : u;
        b[i] = sel ? u : l;
      }
    }
    const bool bad = !(amax > tiny);
    singular = singular || bad;
    const double inv = bad ? 0.0 : 1.0 / A[k][k];
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const double l = A[i][k] * inv;
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
  return singular;
}

// x = pinv(M) f with TensorFlow's default cutoff; rows >= n_dof are padding (identity) and are
// excluded from sigma_max.  Returns the number of dropped singular values.
template <int N>
__device__ __forceinline__ int pinv_solve(double (&A)[N][N], double (&b)[N], int n_dof, double (&x)[N]) {
  for (int sweep = 0; sweep < 40; ++sweep) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < N - 1; ++p) {
#pragma unroll
      for (int q = p + 1; q < N; ++q) {
        double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          al = fma(A[p][j], A[p][j], al);
          be = fma(A[q][j], A[q][j], be);
          ga = fma(A[p][j], A[q][j], ga);
        }
        const bool rot = (fabs(ga) > 1e-300) && (fabs(ga) > 1e-17 * sqrt(al * be));
        rotated = rotated || rot;
        const double zeta = (be - al) / (2.0 * (rot ? ga : 1.0));
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        c = rot ? c : 1.0;
        s = rot ? s : 0.0;
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

}  // namespace rmp2
