/*
 * rmp2_oracle.c -- CPU restatement of the reference's RMP control step.
 *
 * >>> TEST INFRASTRUCTURE.  NOT PART OF THE PRODUCT. <<<
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product library
 * (librmp2_hip.so) never links, loads or calls anything in oracle/.
 *
 * Parity status: the reference hot path needs TensorFlow 2.10 + PyBullet, neither of
 * which is installed here (ordinary ModuleNotFoundError), and the
 * reference ships no golden vectors for q-double-dot.  Against TensorFlow itself this
 * oracle is therefore **parity unpinned**.  It is pinned instead by
 *   (1) oracle/torch_autodiff_oracle.py -- an op-for-op PyTorch-autograd restatement of the
 *       reference's nested-tape differentiation (independent derivation of x, xd, J, c),
 *   (2) SciPy rotation known answers, the closed-form planar 2-link arm, Panda FK known
 *       answers and fp64 central finite differences (tests/test_oracle_*.py),
 *   (3) the reference's own URDF parser output (tests/golden/kinematic_tables.json).
 *
 * What is restated (reference file:line):
 *   forward kinematics      kinematics.py:212-247  (T = T_constant @ T_variable, ordered
 *                                                   chain product :12-20, Rodrigues :99-121)
 *   FK differentiation      kinematics.py:250-270  x = vec(T), xd = J qd, J, c = Jdot qd.
 *                           The reference obtains these by nested GradientTapes
 *                           (helper/rmp_helper.py:50-60); here they are written analytically
 *                           (geometric Jacobian + bias-acceleration recursion).
 *   task maps               taskmap.py:13-20 (identity), :45-54 (4x4 -> position),
 *                           :115-138 (4x4 -> distance, incl. stop_gradient quirk Q5),
 *                           :142-168 (chain rule J = J2 J1, c = c2 + J2 c1)
 *   leaves                  rmp2.py:31-226, rmp.py:226-382 (all quirks Q2, Q4, Q8 kept)
 *   pull-back               rmp.py:157-180   f = (J^T A)(xdd - c),  M = (J^T A) J   in fp32
 *   sum + resolve           rmp.py:133-155   fp32 sum over the pairs of ONE rmp, fp64 sum over
 *                                            rmps, qdd = pinv(M) f in fp64 (quirk Q1)
 *
 * The file is compiled twice: ORC_REAL=float (the reference's working precision) gives
 * orc_*_f32, ORC_REAL=double gives orc_*_f64 (an "exact arithmetic" yardstick used to
 * measure the fp32 noise floor of the reference algorithm itself).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rmp2.h"

#ifdef ORC_DOUBLE
typedef double real;
#define ORC_NAME(x) x##_f64
#define R_SIN sin
#define R_COS cos
#define R_EXP exp
#define R_LOG log
#define R_SQRT sqrt
#define R_FABS fabs
#define R_POW pow
#else
typedef float real;
#define ORC_NAME(x) x##_f32
#define R_SIN sinf
#define R_COS cosf
#define R_EXP expf
#define R_LOG logf
#define R_SQRT sqrtf
#define R_FABS fabsf
#define R_POW powf
#endif

#define NMAX RMP2_MAX_DOF
#define FMAX RMP2_MAX_FRAMES

/* ------------------------------------------------------------------------------------ */
/* small helpers                                                                         */
static void cross3(const real a[3], const real b[3], real o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
static real dot3(const real a[3], const real b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* affine 3x4 product  C = A @ B  with the implicit bottom row [0 0 0 1]
 * (the reference multiplies full 4x4 matrices, kinematics.py:19,240) */
static void mat34_mul(const real A[12], const real B[12], real C[12]) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      C[4 * i + j] = A[4 * i + 0] * B[0 + j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j];
    C[4 * i + 3] = A[4 * i + 0] * B[3] + A[4 * i + 1] * B[7] + A[4 * i + 2] * B[11] + A[4 * i + 3];
  }
}

typedef struct {
  real T[FMAX][12]; /* world transform of every frame (rows 0..2)                         */
  real w[FMAX][3];  /* angular velocity                                                   */
  real al[FMAX][3]; /* angular bias acceleration  (qdd = 0)                                */
  real v[FMAX][3];  /* linear velocity of the frame origin                                 */
  real a[FMAX][3];  /* linear bias acceleration of the frame origin                        */
  real z[FMAX][3];  /* world joint axis                                                    */
} kin_state;

/* kinematics.py:214-247 for all frames + the velocity / bias-acceleration recursion that
 * the reference gets from jacobian_vector_product (kinematics.py:265,267).               */
static void kinematics_all(const rmp2_robot *rb, const float *q, const float *qd, kin_state *ks) {
  const int F = rb->n_frames;
  for (int i = 0; i < F; ++i) {
    const int qi = rb->q_index[i];
    /* q' = gather([q, 0], reorder)   kinematics.py:218-219 */
    const real qv = (qi >= 0) ? (real)q[qi] : (real)0;
    const real qdv = (qi >= 0 && qd) ? (real)qd[qi] : (real)0;
    const real ax[3] = {(real)rb->axis[i][0], (real)rb->axis[i][1], (real)rb->axis[i][2]};
    /* T_variable   kinematics.py:222-237 */
    real Tv[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    if (rb->joint_type[i] == RMP2_JOINT_REVOLUTE) {
      /* Rodrigues: cos*I + sin*[u]x + (1-cos)*u u^T   kinematics.py:103-121 */
      const real c = R_COS(qv), s = R_SIN(qv), omc = (real)1 - c;
      const real ut[3][3] = {{0, -ax[2], ax[1]}, {ax[2], 0, -ax[0]}, {-ax[1], ax[0], 0}};
      for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k)
          Tv[4 * r + k] = c * (r == k ? (real)1 : (real)0) + s * ut[r][k] + omc * (ax[r] * ax[k]);
    } else if (rb->joint_type[i] == RMP2_JOINT_PRISMATIC) {
      for (int r = 0; r < 3; ++r) Tv[4 * r + 3] = qv * ax[r];
    }
    real Tc[12], Tl[12];
    for (int k = 0; k < 12; ++k) Tc[k] = (real)rb->T_const[i][k];
    mat34_mul(Tc, Tv, Tl); /* T = T_constant @ T_variable   kinematics.py:240 */

    const int p = rb->parent[i];
    real wp[3] = {0, 0, 0}, alp[3] = {0, 0, 0}, vp[3] = {0, 0, 0}, ap[3] = {0, 0, 0}, pp[3] = {0, 0, 0};
    if (p < 0) {
      memcpy(ks->T[i], Tl, sizeof(Tl)); /* eye @ T == T exactly */
    } else {
      mat34_mul(ks->T[p], Tl, ks->T[i]); /* m @ all_T[i], left to right   kinematics.py:19 */
      for (int k = 0; k < 3; ++k) {
        wp[k] = ks->w[p][k];
        alp[k] = ks->al[p][k];
        vp[k] = ks->v[p][k];
        ap[k] = ks->a[p][k];
        pp[k] = ks->T[p][4 * k + 3];
      }
    }
    const real *Ti = ks->T[i];
    real zi[3], r[3], t1[3], t2[3], t3[3];
    for (int k = 0; k < 3; ++k) {
      zi[k] = Ti[4 * k + 0] * ax[0] + Ti[4 * k + 1] * ax[1] + Ti[4 * k + 2] * ax[2];
      r[k] = Ti[4 * k + 3] - pp[k];
    }
    cross3(wp, r, t1);  /* w_p x r           */
    cross3(alp, r, t2); /* alpha_p x r       */
    cross3(wp, t1, t3); /* w_p x (w_p x r)   */
    for (int k = 0; k < 3; ++k) {
      ks->z[i][k] = zi[k];
      ks->w[i][k] = wp[k];
      ks->al[i][k] = alp[k];
      ks->v[i][k] = vp[k] + t1[k];
      ks->a[i][k] = ap[k] + t2[k] + t3[k];
    }
    if (rb->joint_type[i] == RMP2_JOINT_REVOLUTE) {
      real zq[3] = {zi[0] * qdv, zi[1] * qdv, zi[2] * qdv}, t4[3];
      cross3(wp, zq, t4);
      for (int k = 0; k < 3; ++k) {
        ks->w[i][k] += zq[k];
        ks->al[i][k] += t4[k];
      }
    } else if (rb->joint_type[i] == RMP2_JOINT_PRISMATIC) {
      real zq[3] = {zi[0] * qdv, zi[1] * qdv, zi[2] * qdv}, t4[3];
      cross3(wp, zq, t4);
      for (int k = 0; k < 3; ++k) {
        ks->v[i][k] += zq[k];
        ks->a[i][k] += (real)2 * t4[k];
      }
    }
  }
}

static int is_ancestor_or_self(const rmp2_robot *rb, int anc, int frame) {
  for (int j = frame; j >= 0; j = rb->parent[j])
    if (j == anc) return 1;
  return 0;
}

/* Structural zeros of the position Jacobian.  The reference obtains d p_frame / d q_j by differentiating the product of LOCAL
 * transforms (kinematics.py:240-247, 265-266): where the origin of `frame`, expressed in the frame of a revolute ancestor joint
 * j, has exact zeros off joint j's axis -- the joint's own origin; a child joint with <origin xyz="0 0 0"> (Panda joints 1/2,
 * 5/6); a tool frame straight up the last joint's axis -- every product in that derivative has an exact zero factor and the
 * column is EXACTLY zero, whatever q.  The world-frame lever z_j x (p_frame - p_j) used below is rounding noise there (1e-8): a
 * tiny but "real" column, which the pseudo-inverse of a set that gives dof j no other metric keeps now and then (rmp.py:153-154).
 * So the zero pattern is taken the reference's way -- the chain of local transforms below joint j, multiplied left to right in
 * `real` at one configuration, lever tested for EXACT zeros against the joint axis -- once per call (the pattern does not depend
 * on q); lz[frame][j] != 0: the column of joint-frame j is exactly zero.  (Pinned against the autodiff restatement,
 * torch_autodiff_oracle.py, in tests/test_oracle_pins.py.) */
static void local_transform(const rmp2_robot *rb, int i, real qv, real Tl[12]) {
  const real ax[3] = {(real)rb->axis[i][0], (real)rb->axis[i][1], (real)rb->axis[i][2]};
  real Tv[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  if (rb->joint_type[i] == RMP2_JOINT_REVOLUTE) {
    const real c = R_COS(qv), s = R_SIN(qv), omc = (real)1 - c;
    const real ut[3][3] = {{0, -ax[2], ax[1]}, {ax[2], 0, -ax[0]}, {-ax[1], ax[0], 0}};
    for (int r = 0; r < 3; ++r)
      for (int k = 0; k < 3; ++k) Tv[4 * r + k] = c * (r == k ? (real)1 : (real)0) + s * ut[r][k] + omc * (ax[r] * ax[k]);
  } else if (rb->joint_type[i] == RMP2_JOINT_PRISMATIC) {
    for (int r = 0; r < 3; ++r) Tv[4 * r + 3] = qv * ax[r];
  }
  real Tc[12];
  for (int k = 0; k < 12; ++k) Tc[k] = (real)rb->T_const[i][k];
  mat34_mul(Tc, Tv, Tl);
}

static void lever_zero_table(const rmp2_robot *rb, const float *q0, unsigned char lz[FMAX][FMAX]) {
  const int F = rb->n_frames;
  static const real I34[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  real Tl[FMAX][12];
  for (int i = 0; i < F; ++i) local_transform(rb, i, rb->q_index[i] >= 0 ? (real)q0[rb->q_index[i]] : (real)0, Tl[i]);
  for (int f = 0; f < F; ++f) {
    int path[FMAX], len = 0;
    for (int j = f; j >= 0; j = rb->parent[j]) path[len++] = j; /* f, parent(f), ..., root */
    for (int j = 0; j < F; ++j) lz[f][j] = 0;
    for (int a = 0; a < len; ++a) {
      const int j = path[a];
      if (rb->joint_type[j] != RMP2_JOINT_REVOLUTE || rb->q_index[j] < 0) continue;
      real Mx[12], tmp[12];
      memcpy(Mx, I34, sizeof(Mx));
      for (int b = a - 1; b >= 0; --b) { /* the transforms below joint j, down to the frame */
        mat34_mul(Mx, Tl[path[b]], tmp);
        memcpy(Mx, tmp, sizeof(Mx));
      }
      const real ax[3] = {(real)rb->axis[j][0], (real)rb->axis[j][1], (real)rb->axis[j][2]};
      const real r[3] = {Mx[3], Mx[7], Mx[11]};
      real c[3];
      cross3(ax, r, c);
      lz[f][j] = (c[0] == 0 && c[1] == 0 && c[2] == 0);
    }
  }
}

/* translational Jacobian of the origin of `frame`  (rows 3,7,11 of the 16 x n Jacobian of
 * kinematics.py:266, i.e. what TaskmapFrom4x4ToPosition selects, taskmap.py:45-54) */
static void jacobian_pos(const rmp2_robot *rb, const kin_state *ks, int frame, real J[3][NMAX], unsigned char lz[FMAX][FMAX]) {
  const int n = rb->n_dof;
  for (int k = 0; k < 3; ++k)
    for (int d = 0; d < n; ++d) J[k][d] = 0;
  for (int j = 0; j < rb->n_frames; ++j) {
    const int d = rb->q_index[j];
    if (d < 0 || rb->joint_type[j] == RMP2_JOINT_FIXED || !is_ancestor_or_self(rb, j, frame)) continue;
    if (rb->joint_type[j] == RMP2_JOINT_REVOLUTE) {
      real r[3], col[3];
      for (int k = 0; k < 3; ++k) r[k] = ks->T[frame][4 * k + 3] - ks->T[j][4 * k + 3];
      cross3(ks->z[j], r, col);
      if (lz && lz[frame][j]) col[0] = col[1] = col[2] = 0; /* exactly zero in the reference (lever_zero_table) */
      for (int k = 0; k < 3; ++k) J[k][d] = col[k];
    } else {
      for (int k = 0; k < 3; ++k) J[k][d] = ks->z[j][k];
    }
  }
}

/* ------------------------------------------------------------------------------------ */
/* leaf policies: (x, xd) -> (xdd_des[k], A[k][k])                                         */

/* helper/rmp_helper.py:62-65 */
static void soft_norm(const real *v, int k, real c, real *out) {
  real s = 0;
  for (int i = 0; i < k; ++i) s += v[i] * v[i];
  const real nrm = R_SQRT(s);
  const real h = nrm + (real)1 / c * R_LOG((real)1 + R_EXP((real)-2 * c * nrm));
  for (int i = 0; i < k; ++i) out[i] = v[i] / h;
}

/* helper/rmp_helper.py:67-74 */
static void stretched_metric(const real *v, int k, real beta, real c, real H[NMAX][NMAX]) {
  real zeta[NMAX];
  soft_norm(v, k, c, zeta);
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) H[i][j] = beta * (zeta[i] * zeta[j]) + ((real)1 - beta) * (i == j ? (real)1 : (real)0);
}

/* rmp2.py:52-83 */
static void leaf_target_attractor(const float *P, const real x[3], const real xd[3], const real g[3], real xdd[3],
                                  real A[NMAX][NMAX]) {
  const real kp = P[0], kd = P[1], eps = P[2], ell = P[3], amin = P[4], smax = P[5], smin = P[6], sb = P[7],
             ellb = P[8];
  real delta[3], dhat[3];
  for (int i = 0; i < 3; ++i) delta[i] = g[i] - x[i];
  const real dn = R_SQRT(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
  const real soft = dn > eps / (real)10 ? dn : eps / (real)10;
  for (int i = 0; i < 3; ++i) dhat[i] = delta[i] / soft;
  for (int i = 0; i < 3; ++i) xdd[i] = kp * delta[i] / (dn + eps) - kd * xd[i];
  const real sd = dn / ell;
  const real a = ((real)1 - amin) * R_EXP((real)-.5 * sd * sd) + amin;
  const real bsd = dn / ellb;
  const real ba = R_EXP((real)-.5 * bsd * bsd);
  const real boost = ba * sb + ((real)1 - ba) * (real)1;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const real m = a * smax * (i == j ? (real)1 : (real)0) + ((real)1 - a) * smin * (dhat[i] * dhat[j]);
      A[i][j] = boost * m;
    }
}

/* rmp.py:241-260 (quirk Q8: h uses c*log, soft_norm uses 1/c*log) */
static void leaf_target_policy(const float *P, int k, const real *x, const real *xd, const real *g, real *xdd,
                               real A[NMAX][NMAX]) {
  const real alpha = P[0], beta_d = P[1], c = P[2];
  real v[NMAX], s2 = 0;
  for (int i = 0; i < k; ++i) {
    v[i] = g[i] - x[i];
    s2 += v[i] * v[i];
  }
  const real vn = R_SQRT(s2);
  const real h = vn + c * R_LOG((real)1 + R_EXP((real)-2 * c * vn));
  const real inv_h = (real)1 / h;
  for (int i = 0; i < k; ++i) xdd[i] = alpha * (inv_h * v[i]) - beta_d * xd[i];
  const real sigma_H = 1, sigma_w = 3;
  const real beta = (real)1 - R_EXP((real)-0.5 * (vn * vn) / (sigma_H * sigma_H));
  real H[NMAX][NMAX];
  stretched_metric(xdd, k, beta, c, H);
  const real w = R_EXP(-vn / sigma_w);
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) A[i][j] = w * H[i][j];
}

/* rmp2.py:100-112 (quirk Q4: metric = w / (1 - diag(ratio^2)) on the FULL matrix) */
static void leaf_joint_velocity_cap(const float *P, int n, const real *xd, real *xdd, real A[NMAX][NMAX]) {
  const real vmax = P[0], region = P[1], gain = P[2], wgt = P[3];
  const real cutoff = vmax - region;
  for (int i = 0; i < n; ++i) {
    const real dv = R_FABS(xd[i]) - cutoff;
    const real sgn = (xd[i] > 0) ? (real)1 : (xd[i] < 0 ? (real)-1 : (real)0);
    const real acc = -R_FABS(gain * dv) * sgn;
    xdd[i] = (R_FABS(xd[i]) < cutoff) ? (real)0 : acc;
    const real clipped = dv < (region - (real)1e-6) ? dv : (region - (real)1e-6);
    const real ratio = clipped / region;
    for (int j = 0; j < n; ++j) A[i][j] = wgt / ((real)1 - (i == j ? ratio * ratio : (real)0));
  }
}

/* rmp2.py:127-137 */
static void leaf_joint_damping(const float *P, int n, const real *xd, real *xdd, real A[NMAX][NMAX]) {
  const real kd = P[0], ms = P[1], inertia = P[2];
  real s2 = 0;
  for (int i = 0; i < n; ++i) s2 += xd[i] * xd[i];
  const real nrm = R_SQRT(s2);
  for (int i = 0; i < n; ++i) {
    xdd[i] = -(kd * nrm) * xd[i];
    for (int j = 0; j < n; ++j) A[i][j] = (i == j ? (real)1 : (real)0) * (ms * nrm + inertia);
  }
}

/* rmp2.py:183-196 on ONE (x, xd) pair */
static void leaf_obstacle_avoidance(const float *P, real x, real xd, real *accel, real *metric) {
  const real margin = P[0], dgain = P[1], dstd = P[2], deps = P[3], gate_len = P[4], rgain = P[5], rstd = P[6],
             radius = P[7], mscal = P[8], estd = P[9], eeps = P[10];
  x = x - margin;
  x = x > 0 ? x : (real)0;
  const real base = mscal / (x / estd + eeps);
  real gate = x * x / (radius * radius) - (real)2 * x / radius + (real)1;
  if (x > radius) gate = 0;
  real m = base * gate;
  const real repel = rgain * R_EXP(-(x / rstd));
  const real sig = (real)1 / ((real)1 + R_EXP(-(xd / gate_len)));
  const real damp = -((real)1 - sig) * dgain * xd / (x / dstd + deps);
  *accel = repel + damp;
  *metric = (x > radius) ? (real)0 : ((real)1 - sig) * m;
}

/* rmp2.py:212-226 */
static void leaf_cspace_biasing(const float *P, const float *goal, int n, const real *q, const real *qd, real *xdd,
                                real A[NMAX][NMAX]) {
  const real ms = P[0], kp = P[1], kd = P[2], thresh = P[3], inertia = P[4];
  real e[NMAX], s2 = 0;
  for (int i = 0; i < n; ++i) {
    e[i] = q[i] - (real)goal[i];
    s2 += e[i] * e[i];
  }
  const real en = R_SQRT(s2);
  for (int i = 0; i < n; ++i) {
    const real pos = (en < thresh) ? (-e[i] * kp) : (-thresh * (e[i] / en) * kp);
    xdd[i] = pos + (-kd * qd[i]);
    for (int j = 0; j < n; ++j) A[i][j] = (i == j ? (real)1 : (real)0) * (ms + inertia);
  }
}

/* rmp.py:357-382 (quirk Q2: A = w * H broadcasts over the LAST axis -> column scaling) */
static void leaf_joint_limit_avoidance(const float *P, const float *lo, const float *hi, int n, const real *q,
                                       const real *qd, real *xdd, real A[NMAX][NMAX]) {
  const real gp = P[0], gd = P[1];
  const real r = (real)0.15;
  const real c2 = (real)(-3.0 / (0.15 * 0.15)), c3 = (real)(2.0 / (0.15 * 0.15 * 0.15));
  real w[NMAX], v[NMAX];
  const real qd_max = (real)(20.0 * (2.0 * M_PI) / 60.0);
  for (int i = 0; i < n; ++i) {
    const real range = (real)hi[i] - (real)lo[i];
    const real du = ((real)hi[i] - q[i]) / range;
    const real dl = (q[i] - (real)lo[i]) / range;
    const real d = du < dl ? du : dl;
    const real spline = c3 * (d * d * d) + c2 * (d * d) + (real)0 * d + (real)1;
    w[i] = d > r ? (real)0 : spline;
    v[i] = qd[i] / qd_max;
    xdd[i] = -gp * q[i] - gd * qd[i];
  }
  real H[NMAX][NMAX];
  stretched_metric(v, n, (real)0.9, (real)5, H);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) A[i][j] = w[j] * H[i][j];
}

/* rmp.py:330-347 */
static void leaf_config_space_biasing(const float *P, const float *q0, int n, const real *q, const real *qd,
                                      real *xdd, real A[NMAX][NMAX]) {
  const real gp = P[0], gd = P[1], w = P[2];
  for (int i = 0; i < n; ++i) {
    xdd[i] = gp * ((real)q0[i] - q[i]) - gd * qd[i];
    for (int j = 0; j < n; ++j) A[i][j] = w * (i == j ? (real)1 : (real)0);
  }
}

/* rmp.py:264-315 CollisionAvoidance: data-fed distance d and unit normal nv of one pair; xd = velocity of the
 * attached point.  Metric w(d) * H with H = beta * zeta zeta^T + (1 - beta) * I at beta = 0 (rmp.py:311,
 * helper/rmp_helper.py:67-74) = w(d) * I. */
static void leaf_collision_avoidance(const float *P, real d, const real nv[3], const real xd[3], real xdd[3],
                                     real *w_out) {
  const real eta_rep = P[0], nu_rep = P[1], eta_damp = P[2], nu_damp = P[3], r = P[4];
  const real alpha_rep = eta_rep * R_EXP(-d / nu_rep);                       /* rmp.py:284 */
  const real alpha_damp = eta_damp / (d / nu_damp + (real)1e-6);            /* rmp.py:288-289 */
  real s = -(xd[0] * nv[0] + xd[1] * nv[1] + xd[2] * nv[2]);                /* rmp.py:290 */
  if (s < 0) s = 0;
  const real nxd = nv[0] * xd[0] + nv[1] * xd[1] + nv[2] * xd[2];
  for (int k = 0; k < 3; ++k) {
    const real f_rep = alpha_rep * nv[k];
    const real f_damp = alpha_damp * (s * nv[k] * nxd);                     /* P_obs xd, rmp.py:291-292 */
    xdd[k] = f_rep - f_damp;
  }
  const real c2 = (real)-3 / (r * r), c3 = (real)2 / (r * r * r);           /* rmp.py:300-304 */
  const real spline = c3 * d * d * d + c2 * d * d + (real)1;
  *w_out = d > r ? (real)0 : spline;
}

/* ------------------------------------------------------------------------------------ */
/* pull-back of one (J[k][n], A[k][k], xdd[k], c[k]) -> f[n], M[n][n]   rmp.py:165-167     */
static void pullback(int k, int n, real J[][NMAX], real A[NMAX][NMAX], const real *xdd, const real *c, real *f,
                     real M[NMAX][NMAX]) {
  real JtA[NMAX][NMAX]; /* n x k */
  for (int i = 0; i < n; ++i)
    for (int b = 0; b < k; ++b) {
      real s = 0;
      for (int a = 0; a < k; ++a) s += J[a][i] * A[a][b];
      JtA[i][b] = s;
    }
  for (int i = 0; i < n; ++i) {
    real s = 0;
    for (int b = 0; b < k; ++b) s += JtA[i][b] * (xdd[b] - c[b]);
    f[i] = s;
    for (int j = 0; j < n; ++j) {
      real m = 0;
      for (int b = 0; b < k; ++b) m += JtA[i][b] * J[b][j];
      M[i][j] = m;
    }
  }
}

/* ------------------------------------------------------------------------------------ */
/* fp64 Moore-Penrose solve  x = pinv(M) f  with TensorFlow's default cutoff
 * rcond = 10 * n * eps (tf.linalg.pinv, called at rmp.py:153).  One-sided (Hestenes)
 * Jacobi on the rows of [M | f]:  G M = W with orthogonal rows  =>  pinv(M) f =
 * sum_i W_i^T (G f)_i / |W_i|^2 over the rows with |W_i| > cutoff.                       */
static int pinv_solve(int n, const double *M, const double *f, double *x) {
  double W[NMAX][NMAX + 1];
  int finite = 1;
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) {
      W[i][j] = M[i * n + j];
      finite &= isfinite(W[i][j]) != 0;
    }
    W[i][n] = f[i];
    finite &= isfinite(f[i]) != 0; /* pinv(M) @ f with a NaN / Inf in f: every sum holds 0 * NaN or x * Inf -- non-finite throughout */
  }
  if (!finite) { /* tf.linalg.pinv of a matrix holding NaN / Inf: its SVD is NaN, and so is every entry of the result  [TF-doc]
                  * (found by tools/fuzz_parity.py: the comparisons below are all false for NaN, which read as "every singular
                  * value dropped" and answered 0) */
    for (int j = 0; j < n; ++j) x[j] = NAN;
    return 0;
  }
  for (int sweep = 0; sweep < 60; ++sweep) {
    int rotated = 0;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int j = 0; j < n; ++j) {
          al += W[p][j] * W[p][j];
          be += W[q][j] * W[q][j];
          ga += W[p][j] * W[q][j];
        }
        if (fabs(ga) <= 1e-300 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
        rotated = 1;
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int j = 0; j <= n; ++j) {
          const double wp = W[p][j], wq = W[q][j];
          W[p][j] = c * wp - s * wq;
          W[q][j] = s * wp + c * wq;
        }
      }
    if (!rotated) break;
  }
  double s2[NMAX], smax = 0;
  for (int i = 0; i < n; ++i) {
    s2[i] = 0;
    for (int j = 0; j < n; ++j) s2[i] += W[i][j] * W[i][j];
    if (sqrt(s2[i]) > smax) smax = sqrt(s2[i]);
  }
  const double cutoff = 10.0 * n * 2.220446049250313e-16 * smax;
  int dropped = 0;
  for (int j = 0; j < n; ++j) x[j] = 0;
  for (int i = 0; i < n; ++i) {
    if (!(sqrt(s2[i]) > cutoff)) {
      ++dropped;
      continue;
    }
    const double coef = W[i][n] / s2[i];
    for (int j = 0; j < n; ++j) x[j] += W[i][j] * coef;
  }
  return dropped;
}

/* ------------------------------------------------------------------------------------ */
/* One robot: RmpCore.evaluate   rmp.py:133-155                                            */
/* Finite cylinder with flat caps, record (centre xyz, radius, unit axis xyz, half height): nearest point Y of its SURFACE to p, the
 * outward unit normal n there and the signed distance sd (p = Y + sd n; negative inside).  Outside: the nearest point of the solid
 * (axial coordinate clamped to [-h, h], radial to [0, r]: side, cap or rim); inside: the nearer of side and cap.  On the axis the
 * radial direction is a fixed perpendicular of the axis (rmp2_device.h point_cylinder takes the same one). */
static void point_cylinder(const float *rec, const real *p, real *Y, real *n, real *sd) {
  const real c[3] = {(real)rec[0], (real)rec[1], (real)rec[2]}, r = (real)rec[3];
  const real u[3] = {(real)rec[4], (real)rec[5], (real)rec[6]}, h = (real)rec[7];
  real w[3], rv[3], e[3];
  for (int k = 0; k < 3; ++k) w[k] = p[k] - c[k];
  const real a = dot3(w, u);
  for (int k = 0; k < 3; ++k) rv[k] = w[k] - a * u[k];
  const real rho = R_SQRT(dot3(rv, rv));
  if (rho > 0) {
    for (int k = 0; k < 3; ++k) e[k] = rv[k] / rho;
  } else {
    const real ax = u[0] < 0 ? -u[0] : u[0], ay = u[1] < 0 ? -u[1] : u[1], az = u[2] < 0 ? -u[2] : u[2];
    const int kx = ax <= ay && ax <= az, ky = !kx && ay <= az;
    const real t[3] = {kx ? 1 : 0, ky ? 1 : 0, (!kx && !ky) ? 1 : 0};
    real cr[3];
    cross3(u, t, cr);
    const real cn = R_SQRT(dot3(cr, cr));
    for (int k = 0; k < 3; ++k) e[k] = cr[k] / cn;
  }
  const real sa = a < 0 ? -1 : 1;
  const real da = (a < 0 ? -a : a) - h, dr = rho - r;
  real ac, rc;
  if (da <= 0 && dr <= 0) {
    if (dr >= da) {
      ac = a, rc = r, *sd = dr;
      for (int k = 0; k < 3; ++k) n[k] = e[k];
    } else {
      ac = sa * h, rc = rho, *sd = da;
      for (int k = 0; k < 3; ++k) n[k] = sa * u[k];
    }
  } else {
    ac = a < -h ? -h : (a > h ? h : a);
    rc = rho < r ? rho : r;
    const real ga = a - ac, gr = rho - rc;
    *sd = R_SQRT(ga * ga + gr * gr);
    for (int k = 0; k < 3; ++k) n[k] = (ga * u[k] + gr * e[k]) / *sd;
  }
  for (int k = 0; k < 3; ++k) Y[k] = c[k] + ac * u[k] + rc * e[k];
}

static void step_one(const rmp2_desc *desc, const float *q32, const float *qd32, const float *goal,
                     const rmp2_obstacles *obs, const float *p_link, const float *p_obs, const float *dist,
                     const int32_t *csr_idx, int csr_n, double *Mc, double *fc, unsigned char lz[FMAX][FMAX]) {
  const rmp2_robot *rb = &desc->robot;
  const int n = rb->n_dof;
  kin_state ks;
  kinematics_all(rb, q32, qd32, &ks);
  real q[NMAX], qd[NMAX];
  for (int i = 0; i < n; ++i) {
    q[i] = (real)q32[i];
    qd[i] = (real)qd32[i];
  }
  for (int i = 0; i < n * n; ++i) Mc[i] = 0; /* np.zeros fp64   rmp.py:136-137 */
  for (int i = 0; i < n; ++i) fc[i] = 0;

  for (int l = 0; l < desc->n_leaves; ++l) { /* for rmp in self.rmps.values()   rmp.py:142 */
    const rmp2_leaf *lf = &desc->leaves[l];
    real f[NMAX], M[NMAX][NMAX];
    if (lf->taskmap == RMP2_TASKMAP_IDENTITY) {
      /* x = q, xd = qd, J = I, c = 0   taskmap.py:13-20 */
      real xdd[NMAX], A[NMAX][NMAX], J[NMAX][NMAX], c0[NMAX], g[NMAX];
      for (int i = 0; i < n; ++i) {
        c0[i] = 0;
        for (int j = 0; j < n; ++j) J[i][j] = (i == j);
      }
      switch (lf->kind) {
        case RMP2_LEAF_JOINT_VELOCITY_CAP: leaf_joint_velocity_cap(lf->params, n, qd, xdd, A); break;
        case RMP2_LEAF_JOINT_DAMPING: leaf_joint_damping(lf->params, n, qd, xdd, A); break;
        case RMP2_LEAF_CSPACE_BIASING: leaf_cspace_biasing(lf->params, lf->vec_a, n, q, qd, xdd, A); break;
        case RMP2_LEAF_JOINT_LIMIT_AVOIDANCE:
          leaf_joint_limit_avoidance(lf->params, lf->vec_a, lf->vec_b, n, q, qd, xdd, A);
          break;
        case RMP2_LEAF_CONFIG_SPACE_BIASING: leaf_config_space_biasing(lf->params, lf->vec_a, n, q, qd, xdd, A); break;
        case RMP2_LEAF_TARGET_POLICY:
          for (int i = 0; i < n; ++i) g[i] = (real)goal[lf->goal_offset + i];
          leaf_target_policy(lf->params, n, q, qd, g, xdd, A);
          break;
        default: continue;
      }
      pullback(n, n, J, A, xdd, c0, f, M);
    } else if (lf->taskmap == RMP2_TASKMAP_FK_POSITION) {
      /* chain [FK(frame), 4x4->pos]: x = p, xd = v, J = J_pos, c = a_bias   taskmap.py:150-160 */
      real J[3][NMAX], x[3], xd[3], c[3], g[3], xdd[3], A[NMAX][NMAX];
      jacobian_pos(rb, &ks, lf->frame, J, lz);
      for (int k = 0; k < 3; ++k) {
        x[k] = ks.T[lf->frame][4 * k + 3];
        xd[k] = ks.v[lf->frame][k];
        c[k] = ks.a[lf->frame][k];
        g[k] = (real)goal[lf->goal_offset + k];
      }
      if (lf->kind == RMP2_LEAF_TARGET_ATTRACTOR)
        leaf_target_attractor(lf->params, x, xd, g, xdd, A);
      else if (lf->kind == RMP2_LEAF_TARGET_POLICY)
        leaf_target_policy(lf->params, 3, x, xd, g, xdd, A);
      else
        continue;
      pullback(3, n, J, A, xdd, c, f, M);
    } else if (lf->taskmap == RMP2_TASKMAP_FK_DISTANCE) {
      /* chain [FK(frame), 4x4->distance] over B pairs; fp32 reduce_sum over the pairs
       * (rmp.py:149-150), one rmp per frame */
      real Jp[3][NMAX];
      jacobian_pos(rb, &ks, lf->frame, Jp, lz);
      const real *Ti = ks.T[lf->frame];
      const real pj[3] = {Ti[3], Ti[7], Ti[11]};
      const real *v = ks.v[lf->frame], *ab = ks.a[lf->frame];
      int B = 0;
      if (obs && obs->mode == RMP2_OBS_EXPLICIT_PAIRS)
        B = obs->pair_begin[l + 1] - obs->pair_begin[l];
      else if (obs && obs->mode == RMP2_OBS_SHARED_SPHERES)
        B = obs->n_spheres;
      else if (obs && obs->mode == RMP2_OBS_RAGGED_SPHERES)
        B = csr_n;
      for (int i = 0; i < n; ++i) {
        f[i] = 0;
        for (int j = 0; j < n; ++j) M[i][j] = 0;
      }
      for (int b = 0; b < B; ++b) {
        real diff[3], nh[3], d;
        if (obs->mode == RMP2_OBS_EXPLICIT_PAIRS) {
          /* taskmap.py:124-129: rel = stop_gradient(p_link - p_joint); crit = p_joint + rel;
           * distance = |crit - p_obs|; its gradient w.r.t. the frame origin is diff / d */
          const float *pl = p_link + 3 * (obs->pair_begin[l] + b), *po = p_obs + 3 * (obs->pair_begin[l] + b);
          for (int k = 0; k < 3; ++k) {
            const real rel = (real)pl[k] - pj[k];
            const real crit = pj[k] + rel;
            diff[k] = crit - (real)po[k];
          }
          d = R_SQRT(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);
          for (int k = 0; k < 3; ++k) nh[k] = diff[k] / d;
        } else {
          /* sphere form == EXPLICIT_PAIRS with p_link = frame origin and p_obs = nearest point
           * on the sphere surface: distance = |p - c| - radius, direction (p - c)/|p - c|; the
           * curvature term below keeps the reference's "p_obs is a fixed point" semantics
           * (it divides by the SURFACE distance d) */
          const int s = (obs->mode == RMP2_OBS_RAGGED_SPHERES) ? csr_idx[b] : b;
          const int cap = obs->primitive == RMP2_PRIM_CAPSULE;
          const float *sp = obs->spheres + (obs->primitive == RMP2_PRIM_SPHERE ? 4 : 8) * s;
          if (obs->primitive == RMP2_PRIM_CYLINDER) {
            /* the reference's own obstacle primitive (simulation.py:245-261): nearest point of the cylinder's surface to the control
             * point, signed distance along the outward normal there -- the same semantics as the sphere form */
            real Ys[3];
            point_cylinder(sp, pj, Ys, nh, &d);
          } else {
          real ctr[3] = {(real)sp[0], (real)sp[1], (real)sp[2]};
          if (cap) { /* nearest point of the segment a-b to the control point (calculate_distances
                      * stage, simulation.py:462-484, for a point-vs-capsule pair) */
            real u[3], w[3];
            for (int k = 0; k < 3; ++k) {
              u[k] = (real)sp[4 + k] - (real)sp[k];
              w[k] = pj[k] - (real)sp[k];
            }
            const real uu = dot3(u, u);
            real t = uu > 0 ? dot3(w, u) / uu : 0;
            t = t < 0 ? 0 : (t > 1 ? 1 : t);
            for (int k = 0; k < 3; ++k) ctr[k] += t * u[k];
          }
          for (int k = 0; k < 3; ++k) diff[k] = pj[k] - ctr[k];
          const real dc = R_SQRT(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);
          d = dc - (real)sp[3];
          for (int k = 0; k < 3; ++k) nh[k] = diff[k] / dc;
          }
        }
        real Jd[1][NMAX];
        for (int j = 0; j < n; ++j) Jd[0][j] = nh[0] * Jp[0][j] + nh[1] * Jp[1][j] + nh[2] * Jp[2][j];
        const real xd = dot3(nh, v);
        const real vv = dot3(v, v);
        const real c = (vv - xd * xd) / d + dot3(nh, ab); /* c2 + J2 c1   taskmap.py:159 */
        real acc, met;
        leaf_obstacle_avoidance(lf->params, d, xd, &acc, &met);
        real A1[NMAX][NMAX], fb[NMAX], Mb[NMAX][NMAX];
        A1[0][0] = met;
        pullback(1, n, Jd, A1, &acc, &c, fb, Mb);
        for (int i = 0; i < n; ++i) { /* tf.reduce_sum(..., axis=0) in fp32 */
          f[i] += fb[i];
          for (int j = 0; j < n; ++j) M[i][j] += Mb[i][j];
        }
      }
    } else if (lf->taskmap == RMP2_TASKMAP_FK_POINT) {
      /* chain [FK(frame), TaskmapRelative4x4(relative_pos), 4x4->pos]  (taskmap.py:79-99,150-160,
       * experiments/two_joint_robot/05_obstacle_avoidance.py:51-61): one point per pair, rigidly attached to
       * the frame at rel (joint frame):  x = p + R rel,  xd = v + w x r,  c = a + al x r + w x (w x r),
       * J = geometric Jacobian at x.  fp32 reduce_sum over the pairs (rmp.py:149-150). */
      if (lf->kind != RMP2_LEAF_COLLISION_AVOIDANCE) continue;
      const real *Ti = ks.T[lf->frame];
      const real *v = ks.v[lf->frame], *ab = ks.a[lf->frame], *w = ks.w[lf->frame], *al = ks.al[lf->frame];
      const int B = (obs && obs->mode == RMP2_OBS_EXPLICIT_PAIRS) ? obs->pair_begin[l + 1] - obs->pair_begin[l] : 0;
      for (int i = 0; i < n; ++i) {
        f[i] = 0;
        for (int j = 0; j < n; ++j) M[i][j] = 0;
      }
      for (int b = 0; b < B; ++b) {
        const int pi = obs->pair_begin[l] + b;
        const float *rel = p_link + 3 * pi, *nv32 = p_obs + 3 * pi;
        const real d = (real)dist[pi];
        real r[3], x[3], xd[3], c[3], t1[3], t2[3], nv[3];
        for (int k = 0; k < 3; ++k) {
          r[k] = Ti[4 * k] * (real)rel[0] + Ti[4 * k + 1] * (real)rel[1] + Ti[4 * k + 2] * (real)rel[2];
          x[k] = Ti[4 * k + 3] + r[k];
          nv[k] = (real)nv32[k];
        }
        cross3(w, r, t1);
        for (int k = 0; k < 3; ++k) xd[k] = v[k] + t1[k];
        cross3(w, t1, t2);
        cross3(al, r, t1);
        for (int k = 0; k < 3; ++k) c[k] = ab[k] + t1[k] + t2[k];
        real Jp[3][NMAX];
        for (int k = 0; k < 3; ++k)
          for (int dd = 0; dd < n; ++dd) Jp[k][dd] = 0;
        for (int j = 0; j < rb->n_frames; ++j) {
          const int dd = rb->q_index[j];
          if (dd < 0 || rb->joint_type[j] == RMP2_JOINT_FIXED || !is_ancestor_or_self(rb, j, lf->frame)) continue;
          if (rb->joint_type[j] == RMP2_JOINT_REVOLUTE) {
            real arm[3], col[3];
            /* (a structural joint: the frame's own origin is the point of its axis -- the lever of the attached point is
             *  then its offset R rel, as the reference's local-frame derivative has it) */
            for (int k = 0; k < 3; ++k) arm[k] = (lz && lz[lf->frame][j]) ? r[k] : x[k] - ks.T[j][4 * k + 3];
            cross3(ks.z[j], arm, col);
            for (int k = 0; k < 3; ++k) Jp[k][dd] = col[k];
          } else {
            for (int k = 0; k < 3; ++k) Jp[k][dd] = ks.z[j][k];
          }
        }
        real xdd[3], wgt, A3[NMAX][NMAX], fb[NMAX], Mb[NMAX][NMAX];
        leaf_collision_avoidance(lf->params, d, nv, xd, xdd, &wgt);
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) A3[i][j] = (i == j) ? wgt : (real)0;
        pullback(3, n, Jp, A3, xdd, c, fb, Mb);
        for (int i = 0; i < n; ++i) {
          f[i] += fb[i];
          for (int j = 0; j < n; ++j) M[i][j] += Mb[i][j];
        }
      }
    } else {
      continue;
    }
    for (int i = 0; i < n; ++i) { /* f_combined += ..., M_combined += ...   fp64   rmp.py:149-150 */
      fc[i] += (double)f[i];
      for (int j = 0; j < n; ++j) Mc[i * n + j] += (double)M[i][j];
    }
  }
}

/* ------------------------------------------------------------------------------------ */
/* exported entry points (host pointers everywhere)                                        */

int ORC_NAME(orc_step)(const rmp2_desc *desc, const float *q, const float *qd, const float *goal, int goal_stride,
                       const rmp2_obstacles *obs, float *qdd, double *qdd64, double *M_out, double *f_out,
                       uint32_t *status, int R) {
  const int n = desc->robot.n_dof;
  if (n > NMAX || desc->robot.n_frames > FMAX) return -1;
  unsigned char lz[FMAX][FMAX];
  if (R > 0) lever_zero_table(&desc->robot, q, lz); /* once per call: the zero pattern does not depend on q */
#pragma omp parallel for schedule(static)
  for (int r = 0; r < R; ++r) {
    double Mc[NMAX * NMAX], fc[NMAX], x[NMAX];
    const float *pl = NULL, *po = NULL, *pd = NULL;
    const int32_t *ci = NULL;
    int cn = 0;
    if (obs && obs->mode == RMP2_OBS_EXPLICIT_PAIRS) {
      pl = obs->p_link + (size_t)r * obs->n_pairs * 3;
      po = obs->p_obs + (size_t)r * obs->n_pairs * 3;
      if (obs->dist) pd = obs->dist + (size_t)r * obs->n_pairs;
    } else if (obs && obs->mode == RMP2_OBS_RAGGED_SPHERES) {
      ci = obs->csr_index + obs->csr_offset[r];
      cn = obs->csr_offset[r + 1] - obs->csr_offset[r];
    }
    step_one(desc, q + (size_t)r * n, qd + (size_t)r * n, goal ? goal + (size_t)r * goal_stride : NULL, obs, pl, po,
             pd, ci, cn, Mc, fc, lz);
    const int dropped = pinv_solve(n, Mc, fc, x); /* rmp.py:153-154 */
    uint32_t st = dropped ? RMP2_STATUS_RANK_DROP : 0u;
    for (int i = 0; i < n; ++i) {
      if (!isfinite(x[i])) st |= RMP2_STATUS_NONFINITE;
      if (qdd) qdd[(size_t)r * n + i] = (float)x[i];
      if (qdd64) qdd64[(size_t)r * n + i] = x[i];
      if (f_out) f_out[(size_t)r * n + i] = fc[i];
    }
    if (M_out) memcpy(M_out + (size_t)r * n * n, Mc, sizeof(double) * n * n);
    if (status) status[r] = st;
  }
  return 0;
}

/* UrdfForwardKinematic.forward for all frames: T[R][F][16]   kinematics.py:212-247 */
int ORC_NAME(orc_forward_kinematics)(const rmp2_desc *desc, const float *q, real *T, int R) {
  const rmp2_robot *rb = &desc->robot;
  const int n = rb->n_dof, F = rb->n_frames;
  for (int r = 0; r < R; ++r) {
    kin_state ks;
    kinematics_all(rb, q + (size_t)r * n, NULL, &ks);
    for (int i = 0; i < F; ++i) {
      real *o = T + ((size_t)r * F + i) * 16;
      memcpy(o, ks.T[i], sizeof(real) * 12);
      o[12] = o[13] = o[14] = 0;
      o[15] = 1;
    }
  }
  return 0;
}

/* UrdfForwardKinematic.differentiate   kinematics.py:250-270:
 * x[R][16], xd[R][16], J[R][16][n], c[R][16] of vec(T_frame)                              */
int ORC_NAME(orc_differentiate)(const rmp2_desc *desc, const float *q, const float *qd, int frame, real *x, real *xd,
                                real *J, real *c, int R) {
  const rmp2_robot *rb = &desc->robot;
  const int n = rb->n_dof;
  for (int r = 0; r < R; ++r) {
    kin_state ks;
    kinematics_all(rb, q + (size_t)r * n, qd + (size_t)r * n, &ks);
    const real *T = ks.T[frame];
    real *xo = x + (size_t)r * 16, *xdo = xd + (size_t)r * 16, *co = c + (size_t)r * 16, *Jo = J + (size_t)r * 16 * n;
    for (int k = 0; k < 16 * n; ++k) Jo[k] = 0;
    for (int k = 0; k < 16; ++k) xo[k] = xdo[k] = co[k] = 0;
    memcpy(xo, T, sizeof(real) * 12);
    xo[15] = 1;
    real Jp[3][NMAX];
    unsigned char lz[FMAX][FMAX];
    lever_zero_table(rb, q + (size_t)r * n, lz);
    jacobian_pos(rb, &ks, frame, Jp, lz);
    const real *w = ks.w[frame], *al = ks.al[frame];
    for (int col = 0; col < 3; ++col) {
      const real Rc[3] = {T[col], T[4 + col], T[8 + col]};
      real wxR[3], alxR[3], wwR[3];
      cross3(w, Rc, wxR);
      cross3(al, Rc, alxR);
      cross3(w, wxR, wwR);
      for (int row = 0; row < 3; ++row) {
        xdo[4 * row + col] = wxR[row];
        co[4 * row + col] = alxR[row] + wwR[row];
      }
      for (int j = 0; j < rb->n_frames; ++j) {
        const int d = rb->q_index[j];
        if (d < 0 || rb->joint_type[j] != RMP2_JOINT_REVOLUTE || !is_ancestor_or_self(rb, j, frame)) continue;
        real zxR[3];
        cross3(ks.z[j], Rc, zxR);
        for (int row = 0; row < 3; ++row) Jo[(4 * row + col) * n + d] = zxR[row];
      }
    }
    for (int row = 0; row < 3; ++row) {
      xdo[4 * row + 3] = ks.v[frame][row];
      co[4 * row + 3] = ks.a[frame][row];
      for (int d = 0; d < n; ++d) Jo[(4 * row + 3) * n + d] = Jp[row][d];
    }
  }
  return 0;
}

/* x = pinv(M) f for one n x n fp64 system (exported for the solver tests) */
int ORC_NAME(orc_pinv_solve)(int n, const double *M, const double *f, double *x) { return pinv_solve(n, M, f, x); }

size_t ORC_NAME(orc_sizeof_desc)(void) { return sizeof(rmp2_desc); }

/* chain [FK(frame), TaskmapFrom4x4ToEuler]  (taskmap.py:57-67, kinematics.py:74-96): x[R][3] = (theta_x, theta_y,
 * theta_z) with R = Rz Ry Rx; analytic derivatives through the angular velocity: w = H(e) ed,
 * H = [Rz Ry ex | Rz ey | ez]  =>  xd = H^-1 w,  J = H^-1 J_w,  c = H^-1 (alpha - Hdot xd).            */
int ORC_NAME(orc_differentiate_euler)(const rmp2_desc *desc, const float *q, const float *qd, int frame, real *x,
                                      real *xd, real *J, real *c, int R) {
  const rmp2_robot *rb = &desc->robot;
  const int n = rb->n_dof;
  for (int r = 0; r < R; ++r) {
    kin_state ks;
    kinematics_all(rb, q + (size_t)r * n, qd + (size_t)r * n, &ks);
    const real *T = ks.T[frame];
    const real r00 = T[0], r10 = T[4], r20 = T[8], r21 = T[9], r22 = T[10];
#ifdef ORC_DOUBLE
    const real ty = -asin(r20), cy = cos(ty);
    const real safe = fabs(cy) < 1e-6 ? 1.0 : cy;
    const real tz = atan2(r10 / safe, r00 / safe), tx = atan2(r21 / safe, r22 / safe);
    const real cz = cos(tz), sz = sin(tz);
#else
    const real ty = -asinf(r20), cy = cosf(ty);
    const real safe = fabsf(cy) < 1e-6f ? 1.0f : cy;
    const real tz = atan2f(r10 / safe, r00 / safe), tx = atan2f(r21 / safe, r22 / safe);
    const real cz = cosf(tz), sz = sinf(tz);
#endif
    const real sy = -r20;
    real *xo = x + (size_t)r * 3, *xdo = xd + (size_t)r * 3, *co = c + (size_t)r * 3, *Jo = J + (size_t)r * 3 * n;
    xo[0] = tx, xo[1] = ty, xo[2] = tz;
#define HINV(u, o)                                   \
  do {                                               \
    const real a_ = (cz * (u)[0] + sz * (u)[1]) / cy; \
    (o)[0] = a_;                                     \
    (o)[1] = -sz * (u)[0] + cz * (u)[1];             \
    (o)[2] = (u)[2] + sy * a_;                       \
  } while (0)
    real ed[3];
    HINV(ks.w[frame], ed);
    for (int i = 0; i < 3; ++i) xdo[i] = ed[i];
    for (int k = 0; k < 3 * n; ++k) Jo[k] = 0;
    for (int j = 0; j < rb->n_frames; ++j) {
      const int d = rb->q_index[j];
      if (d < 0 || rb->joint_type[j] != RMP2_JOINT_REVOLUTE || !is_ancestor_or_self(rb, j, frame)) continue;
      real col[3];
      HINV(ks.z[j], col);
      for (int i = 0; i < 3; ++i) Jo[i * n + d] = col[i];
    }
    const real h0[3] = {-sz * cy * ed[2] - cz * sy * ed[1], cz * cy * ed[2] - sz * sy * ed[1], -cy * ed[1]};
    const real h1[3] = {-cz * ed[2], -sz * ed[2], 0};
    real rhs[3], cc[3];
    for (int i = 0; i < 3; ++i) rhs[i] = ks.al[frame][i] - (h0[i] * ed[0] + h1[i] * ed[1]);
    HINV(rhs, cc);
    for (int i = 0; i < 3; ++i) co[i] = cc[i];
#undef HINV
  }
  return 0;
}
