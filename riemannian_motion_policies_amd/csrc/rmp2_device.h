// rmp2_device.h -- device-side building blocks of the RMP2 control step (gfx950).
//
// Shared by the three mappings of robots to lanes (rmp2_hex.h: 16 lanes per robot; rmp2_quad.h: 4; the
// lane-per-robot kernels in rmp2_hip.hip): the compiled program records, the kernel argument blocks, the
// lane-per-robot frame visit and the accurate (libm) leaf functions.  Everything that is identical for all robots
// of the fleet (the program, leaf parameters, the shared obstacle table) is wave-uniform: the lane / quad kernels
// fetch it through the scalar cache (s_load), the hex kernel stages a packed copy in LDS.
//
// Reference restated (file:line) -- see also include/rmp2.h:
//   local/world transforms   kinematics.py:214-247 (T_constant @ T_variable, ordered product)
//   x, xd, J, c of FK maps   kinematics.py:250-270, helper/rmp_helper.py:50-60 (autodiff there,
//                            geometric Jacobian + bias-acceleration recursion here)
//   distance task map        taskmap.py:115-138 (quirk Q5: derivative of the FRAME ORIGIN)
//   leaves                   rmp2.py:31-226, rmp.py:226-382
//   pull-back / sum          rmp.py:142-150,165-167
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/rmp2.h"

namespace rmp2 {

constexpr int kWave = 64;
constexpr int kMaxOps = RMP2_MAX_FRAMES;

// ---- the compiled program (device memory, read with wave-uniform addresses) ------------
struct DevOp {
  int32_t frame;    // reference frame index
  int32_t restore;  // -2: start from the base, -1: continue from the previous frame, >=0: slot
  int32_t save;     // -1 or slot to save the state into after the visit
  int32_t jtype;    // RMP2_JOINT_*
  int32_t qidx;     // index into q or -1 (evaluated at 0)
  uint32_t anc_mask;  // bit j: dof j moves this frame
  int32_t leaf_begin, leaf_count;  // range in DevProgram::fk_leaves
  float axis[3];    // joint axis in the joint frame; {axis, ctl} is one aligned 16-byte record (one s_load_dwordx4)
  int32_t ctl;      // the walk's control word: (restore + 2) | (save + 1) << 2 | jtype << 4 | (qidx + 1) << 6 | has_leaf << 11
  float Tc[12];     // 96 bytes: 16-byte multiples so that the program can be staged with dwordx4 copies
};
static_assert(offsetof(DevOp, axis) % 16 == 0 && offsetof(DevOp, Tc) % 16 == 0, "DevOp: {axis, ctl} and Tc are fetched as float4");

struct DevLeaf {
  // head: fetched with ONE 64-byte scalar load (s_load_dwordx16)
  int32_t kind, taskmap, frame, goal_offset;
  float P[RMP2_MAX_PARAMS];
  float va[RMP2_MAX_DOF];
  float vb[RMP2_MAX_DOF];
  int32_t index;  // leaf index in the caller's descriptor (pair_begin is indexed by it)
  int32_t dist_ordinal;  // FK_DISTANCE leaves: how many distance leaves precede it (row of rmp2_obstacles.link_capsules)
  int32_t next_pair_leaf;  // exec_leaves only: descriptor index of the NEXT FK_DISTANCE leaf in execution order (whose explicit
                           // pairs the quad kernel prefetches while this leaf is evaluated), -1: none
  int32_t pad_[1];  // 208 bytes
};
// the 64-byte head of a leaf / the 32-byte control block of an op, as value types: copying
// them makes the compiler issue one wide scalar load instead of one dependent s_load per field
struct LeafHead {
  int32_t kind, taskmap, frame, goal_offset;
  float P[RMP2_MAX_PARAMS];
};
struct OpCtl {
  int32_t frame, restore, save, jtype, qidx;
  uint32_t anc_mask;
  int32_t leaf_begin, leaf_count;
};
static_assert(sizeof(DevOp) % 16 == 0 && sizeof(DevLeaf) % 16 == 0, "program records must be 16-byte multiples");

struct DevProgram {
  int32_t n_ops, n_dof, n_frames, n_leaves;
  int32_t n_fk_leaves, n_id_leaves, goal_floats, solve_mode;
  uint32_t rev_mask;  // bit j: dof j is revolute (else prismatic)
  uint32_t dof_ops[3];  // op (schedule position) of the joint that owns dof j: 5 bits each, 6 dofs per word
  DevOp ops[kMaxOps];
  DevLeaf leaves[RMP2_MAX_LEAVES];
  int32_t fk_leaves[RMP2_MAX_LEAVES];  // leaf ids grouped by op
  int32_t id_leaves[RMP2_MAX_LEAVES];  // identity-task-map leaves, caller's order
  int32_t n_leaf_ops;
  int32_t leaf_ops[kMaxOps];  // schedule positions of the frames that carry leaves
  // tables of the 16-lanes-per-robot kernel (rmp2_hex.h): the tree as parent pointers instead of a walk order
  struct Hex {
    int32_t n_levels;                 // pointer-jumping rounds: smallest L with 2^L >= deepest chain
    int32_t is_chain;                 // 1: op k's parent is op k - 1 for every k (a serial chain in program order)
    int32_t pad_[2];
    int32_t jump[5][kMaxOps];         // jump[l][k]: the op 2^l levels above op k, -1 = above the base
    uint32_t op_anc[kMaxOps];         // bit j: op j is op k itself or one of its ancestors
  } hex;
  // The leaf phase of the scalar-cache program walk (quad kernel, large fleets), flattened so that a frame costs ONE
  // dependent scalar fetch instead of four (leaf_ops[t] -> ops[k] -> fk_leaves[i] -> leaves[id]): one 16-byte record
  // per leaf-bearing frame, fetched a frame ahead, and the FK-map leaves in execution order (followed by the
  // identity-map leaves in theirs: exec_leaves[n_fk_leaves + i]).
  struct LeafFrame {
    int32_t op;         // schedule position of the frame
    uint32_t anc_mask;  // dofs that move it
    int32_t leaf_begin, leaf_count;  // range in exec_leaves
  } leaf_frames[kMaxOps];
  DevLeaf exec_leaves[RMP2_MAX_LEAVES];
};

// Local transform of one frame as an affine function of (cos q, sin q, q), precomputed on the host in fp64:
//   R_local = A0 + cos(q) A1 + sin(q) A2 ,  t_local = tc + q tu
// (revolute: A0 = Rc u u^T, A1 = Rc (I - u u^T), A2 = Rc [u]x -- Rodrigues' formula, kinematics.py:103-121,
//  multiplied through T_constant; prismatic: A0 = Rc, tu = Rc u; fixed: A0 = Rc)
struct HexOp {
  float A0[9], A1[9], A2[9], tc[3], tu[3], pad_[3];  // 144 bytes
};
static_assert(sizeof(HexOp) % 16 == 0, "staged with dwordx4 copies");
// the fields of DevOp the hex kernel reads, packed (the staged program is fetched by every wave of the grid at the
// same time: every byte of it is L2 hot-spot traffic)
struct HexCtl {
  int32_t jtype, qidx;
  uint32_t anc_mask;
  int32_t leaf_begin, leaf_count;
  float axis[3];
};
static_assert(sizeof(HexCtl) == 32, "two dwordx4");

struct ObsArgs {
  int32_t mode, n_spheres, n_pairs;
  int32_t capsule;  // 1: table records are 8-float capsules, 0: 4-float spheres
  const float* __restrict__ spheres;
  const float* __restrict__ p_link;
  const float* __restrict__ p_obs;
  const int32_t* __restrict__ csr_offset;
  const int32_t* __restrict__ csr_index;
  const int32_t* __restrict__ pair_begin;  // device copy, [RMP2_MAX_LEAVES + 1]
  const float* __restrict__ dist;          // [R][P] distances of the attached-point leaves
  const float* __restrict__ link_caps;     // [n_distance_leaves][8] link capsules in frame coordinates (table modes), or null
  int32_t cylinder;  // 1: the 8-float records are finite CYLINDERS (centre, radius, unit axis, half height: RMP2_PRIM_CYLINDER) -- `capsule`
                     // is then 1 as well (the 8-float-record builds serve both; the closed form per pair is a wave-uniform branch)
  int32_t glds;  // EXPLICIT_PAIRS, plain two-wave quad build: the pair arrays are streamed half a leaf ahead by LDS-DMA (set by
                 // launch_quad when every leaf segment is 16-byte aligned; the launch then carries 6 KiB more LDS per wave)
};

// The reference's obstacle primitive: a finite CYLINDER with flat caps (simulation.py:245-261: pybullet.GEOM_CYLINDER of radius r and
// height 2 h at a pose; seven of them in experiments/franka_panda/06_cluttered_environment.py:39-52).  Record (8 floats):
// ra = (centre xyz, radius), rb = (unit axis xyz, half height).  Nearest point Y of the cylinder's SURFACE to the point p, the
// outward unit normal n of the surface there and the signed distance sd (p = Y + sd n; negative inside) -- what PyBullet's closest
// points give the reference for a point-like link shape (simulation.py:462-484): outside, the nearest point of the solid
// (clamp the axial coordinate to [-h, h] and the radial one to [0, r]: side, cap or rim); inside, the nearer of the side and the
// cap.  On the axis the radial direction is undefined: a fixed perpendicular of the axis is taken (only the rim and the inside-side
// cases use it there; distance and position are unaffected).
__device__ __forceinline__ void point_cylinder(const float4 ra, const float4 rb, const float p[3], float Y[3], float n[3], float& sd) {
  const float u[3] = {rb.x, rb.y, rb.z};
  const float r = ra.w, h = rb.w;
  const float w[3] = {p[0] - ra.x, p[1] - ra.y, p[2] - ra.z};
  const float a = w[0] * u[0] + w[1] * u[1] + w[2] * u[2];
  float rv[3] = {w[0] - a * u[0], w[1] - a * u[1], w[2] - a * u[2]};
  const float rho2 = rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2];
  const float rho = sqrtf(rho2);
  float e[3];  // unit radial direction
  if (rho > 0.f) {
    const float inv = 1.0f / rho;
    e[0] = rv[0] * inv, e[1] = rv[1] * inv, e[2] = rv[2] * inv;
  } else {  // on the axis: any unit vector perpendicular to u (cross with the coordinate axis u is least aligned with)
    const float ax = fabsf(u[0]), ay = fabsf(u[1]), az = fabsf(u[2]);
    const bool kx = ax <= ay && ax <= az, ky = !kx && ay <= az;
    const float t[3] = {kx ? 1.f : 0.f, ky ? 1.f : 0.f, (!kx && !ky) ? 1.f : 0.f};
    const float c[3] = {u[1] * t[2] - u[2] * t[1], u[2] * t[0] - u[0] * t[2], u[0] * t[1] - u[1] * t[0]};
    const float inv = 1.0f / sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    e[0] = c[0] * inv, e[1] = c[1] * inv, e[2] = c[2] * inv;
  }
  const float sa = a < 0.f ? -1.f : 1.f;
  const float da = fabsf(a) - h, dr = rho - r;  // signed distances to the cap plane and to the lateral surface
  float ac, rc;                                  // axial and radial coordinate of Y
  if (da <= 0.f && dr <= 0.f) {                  // inside (or on the surface): the nearer face
    if (dr >= da) {                              // lateral surface
      ac = a, rc = r, sd = dr;
      n[0] = e[0], n[1] = e[1], n[2] = e[2];
    } else {                                     // cap
      ac = sa * h, rc = rho, sd = da;
      n[0] = sa * u[0], n[1] = sa * u[1], n[2] = sa * u[2];
    }
  } else {
    ac = fminf(fmaxf(a, -h), h), rc = fminf(rho, r);
    const float ga = a - ac, gr = rho - rc;      // the gap's axial and radial component (one of them 0 off the rim)
    sd = sqrtf(ga * ga + gr * gr);
    const float inv = 1.0f / sd;
    n[0] = (ga * u[0] + gr * e[0]) * inv, n[1] = (ga * u[1] + gr * e[1]) * inv, n[2] = (ga * u[2] + gr * e[2]) * inv;
  }
  Y[0] = ra.x + ac * u[0] + rc * e[0], Y[1] = ra.y + ac * u[1] + rc * e[1], Y[2] = ra.z + ac * u[2] + rc * e[2];
}

// Nearest points of a link segment A + s D (s in [0, 1]) and a cylinder: the signed distance of a point to a convex body is a convex
// function of the point, so it is convex in s, and its derivative along the segment is n(s) . D with n the outward normal of the
// nearest surface point -- monotone in s.  Bisection on its sign (24 halvings: the parameter to 6e-8) after the two end tests; every
// step is one point_cylinder.  (No closed form: the rim case is a quartic.)
__device__ __forceinline__ void segment_cylinder(const float4 ra, const float4 rb, const float A[3], const float D[3], float X[3],
                                                 float Y[3], float n[3], float& sd) {
  auto at = [&](float s) {
    X[0] = fmaf(s, D[0], A[0]), X[1] = fmaf(s, D[1], A[1]), X[2] = fmaf(s, D[2], A[2]);
    point_cylinder(ra, rb, X, Y, n, sd);
    return n[0] * D[0] + n[1] * D[1] + n[2] * D[2];
  };
  if (!(at(0.f) < 0.f)) return;       // the distance grows from A on (or the segment is a point): A it is
  if (!(at(1.f) > 0.f)) return;       // still falling at B
  float lo = 0.f, hi = 1.f;
  for (int it = 0; it < 24; ++it) {
    const float mid = 0.5f * (lo + hi);
    if (at(mid) < 0.f) lo = mid; else hi = mid;
  }
  (void)at(0.5f * (lo + hi));
}

// Nearest point of the segment a-b (capsule axis) to the control point p: the point-vs-capsule case of the
// reference's CPU closest-point stage (simulation.py:462-484).  rec = (a.xyz, radius, b.xyz, unused).
__device__ __forceinline__ void capsule_centre(const float4 ra, const float4 rb, const float p[3], float ctr[3]) {
  const float u[3] = {rb.x - ra.x, rb.y - ra.y, rb.z - ra.z};
  const float w[3] = {p[0] - ra.x, p[1] - ra.y, p[2] - ra.z};
  const float uu = u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
  float t = uu > 0.f ? (w[0] * u[0] + w[1] * u[1] + w[2] * u[2]) / uu : 0.f;
  t = fminf(fmaxf(t, 0.f), 1.f);
  ctr[0] = ra.x + t * u[0];
  ctr[1] = ra.y + t * u[1];
  ctr[2] = ra.z + t * u[2];
}

// Closest points of two segments p1-q1 and p2-q2 (the axes of two capsules): parameters s, t in [0, 1] of the nearest
// points c1 = p1 + s (q1 - p1), c2 = p2 + t (q2 - p2).  The standard clamped solution of the 2 x 2 normal equations
// (degenerate segments -- points -- included); the link-capsule case of the reference's closest-point stage, which
// PyBullet answers for the link's collision shape (simulation.py:462-484).
__device__ __forceinline__ void segment_segment(const float p1[3], const float q1[3], const float p2[3], const float q2[3],
                                                float& s, float& t) {
  const float d1[3] = {q1[0] - p1[0], q1[1] - p1[1], q1[2] - p1[2]};
  const float d2[3] = {q2[0] - p2[0], q2[1] - p2[1], q2[2] - p2[2]};
  const float r[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  const float a = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2];
  const float e = d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2];
  const float f = d2[0] * r[0] + d2[1] * r[1] + d2[2] * r[2];
  const float c = d1[0] * r[0] + d1[1] * r[1] + d1[2] * r[2];
  const float b = d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2];
  if (!(a > 0.f) && !(e > 0.f)) {  // both are points
    s = t = 0.f;
    return;
  }
  if (!(a > 0.f)) {  // the first is a point
    s = 0.f;
    t = fminf(fmaxf(f / e, 0.f), 1.f);
    return;
  }
  if (!(e > 0.f)) {  // the second is a point
    t = 0.f;
    s = fminf(fmaxf(-c / a, 0.f), 1.f);
    return;
  }
  const float denom = a * e - b * b;  // >= 0; 0 for parallel axes: any s does, 0 is taken
  s = denom > 0.f ? fminf(fmaxf((b * f - c * e) / denom, 0.f), 1.f) : 0.f;
  t = (b * s + f) / e;
  if (t < 0.f) {
    t = 0.f;
    s = fminf(fmaxf(-c / a, 0.f), 1.f);
  } else if (t > 1.f) {
    t = 1.f;
    s = fminf(fmaxf((b - c) / a, 0.f), 1.f);
  }
}

// A joint position that is not finite (a robot that diverged in a rollout, a NaN fed by the caller) resolves to q'' = NaN for
// the whole robot, in the reference (NaN FK -> NaN metric -> pinv of a NaN matrix, rmp.py:153) and here.  Left in place it also
// makes every range test of the robot's control points come out "in range" (NaN compares false): 256 pairs instead of ~55, and
// the robot's whole wave waits for it -- a fleet with 6 % dead robots stepped 25 % slower.  The non-finiteness is therefore
// moved from the position to the velocity of the same joint: FK and the range tests see q_i = 0, and the kernels make the FORCE of
// a dof with a non-finite velocity non-finite by construction before any export or resolve (rmp2_quad.h / rmp2_hex.h behind the
// identity leaves): the resolve settles the robot to NaN with RMP2_STATUS_NONFINITE whatever the leaves see of the joint.  (Until
// round 4 the NaN velocity was left to reach the system through the velocity-dependent leaves -- which a set of distance leaves
// with every obstacle out of range does not have: tools/fuzz_parity.py seed 504944.)
// (A plant tick turns a NaN velocity into a NaN position, so "velocity NaN" identifies the quarantined joints when the state
// is written back.)
__device__ __forceinline__ void quarantine(float& qv, float& qdv) {
  const bool bad = !(fabsf(qv) < 3.0e38f);
  qv = bad ? 0.f : qv;
  qdv = bad ? __builtin_nanf("") : qdv;
}

struct OutArgs {
  float* __restrict__ qdd;
  uint32_t* __restrict__ status;
  double* __restrict__ M;
  double* __restrict__ f;
};

// ---- small vector helpers ------------------------------------------------------------
__device__ __forceinline__ void cross3(const float a[3], const float b[3], float o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ float dot3(const float a[3], const float b[3]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

// running state of the frame being visited (world coordinates)
struct FrameState {
  float R[9];   // rotation, row-major
  float p[3];   // origin
  float w[3];   // angular velocity
  float al[3];  // angular bias acceleration (qdd = 0)
  float v[3];   // linear velocity of the origin
  float a[3];   // linear bias acceleration of the origin
};

// Visit one frame: s (parent state, ignored when from_base) -> s (this frame's state).
// Returns the joint's world axis in z.   kinematics.py:214-247 + the analytic counterpart of
// the two jacobian_vector_product calls at kinematics.py:265,267.
template <bool kWithVelocity>
__device__ __forceinline__ void visit_frame(FrameState& s, const DevOp& op, float qv, float qdv, bool from_base,
                                            float z[3]) {
  const float ax[3] = {op.axis[0], op.axis[1], op.axis[2]};
  float Rl[9], tl[3];
  if (op.jtype == RMP2_JOINT_REVOLUTE) {
    // T_variable = [Rodrigues(axis, q) | 0]; Rodrigues = cos*I + sin*[u]x + (1-cos)*u u^T
    float sn, cs;
    sincosf(qv, &sn, &cs);
    const float omc = 1.0f - cs;
    const float ut[9] = {0.f, -ax[2], ax[1], ax[2], 0.f, -ax[0], -ax[1], ax[0], 0.f};
    float Rv[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k)
        Rv[3 * r + k] = cs * (r == k ? 1.f : 0.f) + sn * ut[3 * r + k] + omc * (ax[r] * ax[k]);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k)
        Rl[3 * r + k] = op.Tc[4 * r + 0] * Rv[k] + op.Tc[4 * r + 1] * Rv[3 + k] + op.Tc[4 * r + 2] * Rv[6 + k];
#pragma unroll
    for (int r = 0; r < 3; ++r) tl[r] = op.Tc[4 * r + 3];
  } else {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int k = 0; k < 3; ++k) Rl[3 * r + k] = op.Tc[4 * r + k];
      tl[r] = op.Tc[4 * r + 3];
    }
    if (op.jtype == RMP2_JOINT_PRISMATIC) {
      const float tv[3] = {qv * ax[0], qv * ax[1], qv * ax[2]};
#pragma unroll
      for (int r = 0; r < 3; ++r)
        tl[r] = op.Tc[4 * r + 0] * tv[0] + op.Tc[4 * r + 1] * tv[1] + op.Tc[4 * r + 2] * tv[2] + op.Tc[4 * r + 3];
    }
  }
  float wp[3] = {0.f, 0.f, 0.f}, alp[3] = {0.f, 0.f, 0.f}, vp[3] = {0.f, 0.f, 0.f}, ap[3] = {0.f, 0.f, 0.f};
  float r[3];
  if (from_base) {
#pragma unroll
    for (int k = 0; k < 9; ++k) s.R[k] = Rl[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      s.p[k] = tl[k];
      r[k] = tl[k];
    }
  } else {
    float Rn[9], pn[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int k = 0; k < 3; ++k)
        Rn[3 * i + k] = s.R[3 * i + 0] * Rl[k] + s.R[3 * i + 1] * Rl[3 + k] + s.R[3 * i + 2] * Rl[6 + k];
      pn[i] = s.R[3 * i + 0] * tl[0] + s.R[3 * i + 1] * tl[1] + s.R[3 * i + 2] * tl[2] + s.p[i];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      r[k] = pn[k] - s.p[k];
      s.p[k] = pn[k];
      if (kWithVelocity) {
        wp[k] = s.w[k];
        alp[k] = s.al[k];
        vp[k] = s.v[k];
        ap[k] = s.a[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) s.R[k] = Rn[k];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) z[k] = s.R[3 * k + 0] * ax[0] + s.R[3 * k + 1] * ax[1] + s.R[3 * k + 2] * ax[2];
  if (kWithVelocity) {
    float t1[3], t2[3], t3[3];
    cross3(wp, r, t1);
    cross3(alp, r, t2);
    cross3(wp, t1, t3);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      s.w[k] = wp[k];
      s.al[k] = alp[k];
      s.v[k] = vp[k] + t1[k];
      s.a[k] = ap[k] + t2[k] + t3[k];
    }
    if (op.jtype != RMP2_JOINT_FIXED) {
      const float zq[3] = {z[0] * qdv, z[1] * qdv, z[2] * qdv};
      float t4[3];
      cross3(wp, zq, t4);
      if (op.jtype == RMP2_JOINT_REVOLUTE) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          s.w[k] += zq[k];
          s.al[k] += t4[k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          s.v[k] += zq[k];
          s.a[k] += 2.0f * t4[k];
        }
      }
    }
  }
}

// ---- leaves on a 3-d position task space ----------------------------------------------
// symmetric 3x3 stored as {xx, xy, xz, yy, yz, zz}

// rmp2.py:52-83
__device__ __forceinline__ void leaf_target_attractor(const float* __restrict__ P, const float x[3], const float xd[3],
                                                      const float g[3], float xdd[3], float A[6]) {
  const float kp = P[0], kd = P[1], eps = P[2], ell = P[3], amin = P[4], smax = P[5], smin = P[6], sb = P[7],
              ellb = P[8];
  float delta[3], dhat[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) delta[i] = g[i] - x[i];
  const float dn = sqrtf(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
  const float soft = fmaxf(dn, eps / 10.0f);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    dhat[i] = delta[i] / soft;
    xdd[i] = kp * delta[i] / (dn + eps) - kd * xd[i];
  }
  const float sd = dn / ell;
  const float a = (1.0f - amin) * expf(-0.5f * sd * sd) + amin;
  const float bsd = dn / ellb;
  const float ba = expf(-0.5f * bsd * bsd);
  const float boost = ba * sb + (1.0f - ba) * 1.0f;
  const float wI = a * smax, wS = (1.0f - a) * smin;
  A[0] = boost * (wI + wS * (dhat[0] * dhat[0]));
  A[1] = boost * (wS * (dhat[0] * dhat[1]));
  A[2] = boost * (wS * (dhat[0] * dhat[2]));
  A[3] = boost * (wI + wS * (dhat[1] * dhat[1]));
  A[4] = boost * (wS * (dhat[1] * dhat[2]));
  A[5] = boost * (wI + wS * (dhat[2] * dhat[2]));
}

// rmp.py:241-260 on a 3-d task space (quirk Q8 kept: c*log in h, (1/c)*log in soft_norm)
__device__ __forceinline__ void leaf_target_policy3(const float* __restrict__ P, const float x[3], const float xd[3],
                                                    const float g[3], float xdd[3], float A[6]) {
  const float alpha = P[0], beta_d = P[1], c = P[2];
  float v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] = g[i] - x[i];
  const float vn = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const float h = vn + c * logf(1.0f + expf(-2.0f * c * vn));
  const float inv_h = 1.0f / h;
#pragma unroll
  for (int i = 0; i < 3; ++i) xdd[i] = alpha * (inv_h * v[i]) - beta_d * xd[i];
  const float beta = 1.0f - expf(-0.5f * (vn * vn) / 1.0f);
  const float fn = sqrtf(xdd[0] * xdd[0] + xdd[1] * xdd[1] + xdd[2] * xdd[2]);
  const float hs = fn + 1.0f / c * logf(1.0f + expf(-2.0f * c * fn));
  const float ze[3] = {xdd[0] / hs, xdd[1] / hs, xdd[2] / hs};
  const float w = expf(-vn / 3.0f);
  const float omb = 1.0f - beta;
  A[0] = w * (beta * (ze[0] * ze[0]) + omb);
  A[1] = w * (beta * (ze[0] * ze[1]));
  A[2] = w * (beta * (ze[0] * ze[2]));
  A[3] = w * (beta * (ze[1] * ze[1]) + omb);
  A[4] = w * (beta * (ze[1] * ze[2]));
  A[5] = w * (beta * (ze[2] * ze[2]) + omb);
}

// rmp2.py:183-196 on one (distance, distance-rate) pair
__device__ __forceinline__ void leaf_obstacle_avoidance(const float* __restrict__ P, float x, float xd, float& accel,
                                                        float& metric) {
  const float margin = P[0], dgain = P[1], dstd = P[2], deps = P[3], gate_len = P[4], rgain = P[5], rstd = P[6],
              radius = P[7], mscal = P[8], estd = P[9], eeps = P[10];
  x = fmaxf(x - margin, 0.0f);
  const float base = mscal / (x / estd + eeps);
  const float gate = x * x / (radius * radius) - 2.0f * x / radius + 1.0f;
  const float repel = rgain * expf(-(x / rstd));
  // 1 - sigmoid(z) without the cancellation of `1. - tf.sigmoid(z)` for z > 0 (rmp2_quad.h obstacle_pair)
  const float zg = xd / gate_len;
  const float ez = expf(-fabsf(zg));
  const float oms = (zg > 0.0f ? (zg > 17.328680f ? 0.0f : ez) : 1.0f) / (1.0f + ez);   // (25 ln 2: an fp32 sigmoid is 1 beyond, the gate an exact 0)
  const float damp = -oms * dgain * xd / (x / dstd + deps);
  accel = repel + damp;
  metric = (x > radius) ? 0.0f : oms * (base * gate);
}

// ---- pull-back of a position-type leaf into the fp64 accumulators ------------------------
// col[j] = J[:, j] (3-vector per dof), S = leaf metric (sym 3x3), h = S (xdd - c)  [or the
// pair sum  sum_b a_b (xdd_b - c_b) n_b  for distance leaves].
//   f += J^T h ,  M += J^T S J      (rmp.py:165-167; fp32 products, fp64 accumulation :149-150)
// Ms holds the upper triangle, row-major: idx(i,j) = i*N - i*(i-1)/2 + (j-i), i <= j.
template <int N>
__device__ __forceinline__ constexpr int sym_idx(int i, int j) {
  return i * N - (i * (i - 1)) / 2 + (j - i);
}

// rmp.py:264-315 CollisionAvoidance on one attached point: data-fed distance d and unit normal nv; metric
// w(d) * I (beta = 0, rmp.py:311).
__device__ __forceinline__ void leaf_collision_avoidance(const float* P, float d, const float nv[3], const float xd[3],
                                                         float xdd[3], float& wgt) {
  const float eta_rep = P[0], nu_rep = P[1], eta_damp = P[2], nu_damp = P[3], r = P[4];
  const float alpha_rep = eta_rep * expf(-d / nu_rep);
  const float alpha_damp = eta_damp / (d / nu_damp + 1e-6f);
  const float nxd = nv[0] * xd[0] + nv[1] * xd[1] + nv[2] * xd[2];
  const float s = fmaxf(-nxd, 0.f);
#pragma unroll
  for (int k = 0; k < 3; ++k) xdd[k] = alpha_rep * nv[k] - alpha_damp * (s * nv[k] * nxd);
  const float c2 = -3.f / (r * r), c3 = 2.f / (r * r * r);
  const float spline = c3 * d * d * d + c2 * d * d + 1.f;
  wgt = d > r ? 0.f : spline;
}

// Datamanager fields of ONE (link capsule, primitive) pair for the attached-point leaves, formed on the device (the reference
// takes them from PyBullet's closest points every control step, simulation.py:462-484 / data_management.py:22-53): world link
// axis LA + s LD (laa = |LD|^2, inv_laa its reciprocal or 0, radius lrad), primitive record ca (a.xyz, radius) and cb (b.xyz;
// = ca for a sphere).  X, Y = nearest points of the two axes (the clamped 2 x 2 normal equations, segment_segment's cases,
// branch-free); p_link = X - lrad n, p_obs = Y + r n with n = (X - Y) / |X - Y|; out:
//   dd = |p_link - p_obs|, nv = (p_link - p_obs) / dd, r = p_link - P3 (= R relative_position, the attached point's lever arm).
// Overlapping shapes: the points have crossed, the distance reads positive and the normal flips (as the explicit arrays would).
__device__ __forceinline__ void link_pair_fields(const float LA[3], const float LD[3], float laa, float inv_laa, float lrad,
                                                 const float4 ca, const float4 cb, const float P3[3], float r[3], float nv[3],
                                                 float& dd) {
  const float d2v[3] = {cb.x - ca.x, cb.y - ca.y, cb.z - ca.z};
  const float rr[3] = {LA[0] - ca.x, LA[1] - ca.y, LA[2] - ca.z};
  const float ee = dot3(d2v, d2v), ff = dot3(d2v, rr), cc = dot3(LD, rr), bb = dot3(LD, d2v);
  const float inv_e = ee > 0.f ? 1.0f / ee : 0.f;
  const float den = fmaf(laa, ee, -bb * bb);
  const float s0 = (den > 0.f && laa > 0.f) ? fminf(fmaxf(fmaf(bb, ff, -cc * ee) / den, 0.f), 1.f) : 0.f;
  const float t0 = fmaf(bb, s0, ff) * inv_e;
  const float s_lo = fminf(fmaxf(-cc * inv_laa, 0.f), 1.f), s_hi = fminf(fmaxf((bb - cc) * inv_laa, 0.f), 1.f);
  const float sl = (t0 < 0.f || !(ee > 0.f)) ? s_lo : (t0 > 1.f ? s_hi : s0);
  const float to = fminf(fmaxf(t0, 0.f), 1.f);
  float X[3], diff[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) X[c] = fmaf(sl, LD[c], LA[c]);
  diff[0] = X[0] - fmaf(to, d2v[0], ca.x), diff[1] = X[1] - fmaf(to, d2v[1], ca.y), diff[2] = X[2] - fmaf(to, d2v[2], ca.z);
  const float dn = sqrtf(dot3(diff, diff));
  // the axes INTERSECT (a sphere centred on the link's axis, two crossing segments): there is no common normal.  The shapes
  // overlap by the sum of their radii whatever direction is taken; a fixed one (+z, as configs.pairs_from_link_capsules) keeps the
  // fields finite and the same on every evaluation -- 1 / 0 would make normal and lever arm NaN (round-4 advisor finding).  A NaN
  // distance is not this case (dn == 0 is false for it) and propagates as before.
  const bool crossing = dn == 0.f;
  diff[2] = crossing ? 1.f : diff[2];
  const float inv0 = crossing ? 1.f : 1.0f / dn;
  const float sgap = dn - ca.w - lrad;
  dd = fabsf(sgap);
  const float inv = copysignf(inv0, sgap);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    nv[c] = diff[c] * inv;
    r[c] = (X[c] - lrad * (diff[c] * inv0)) - P3[c];
  }
}

// Rank-one form of a leaf's summed metric.  A distance leaf pulls back S = sum_k m_k n_k n_k^T (one 3 x 3 per frame instead of
// one scalar per pair: the reason the pair loop is cheap) as J^T S J, accurate to eps32 |S| |J_j|^2 ABSOLUTELY -- backward stable
// in S, not componentwise: where a column is nearly perpendicular to the pair direction (rho = |n . J_j| / |J_j| below
// sqrt(eps32)) the entry m rho^2 |J_j|^2 comes out off by eps32 / rho^2 RELATIVE, while the reference, which squares the projected
// scalar n . J_j (taskmap.py:150-160, rmp.py:165-167), has it to 2 eps32 / rho.  Beside any other metric on the dof that is
// 6e-8 of the total; in a set without an inertia leaf it can be the dof's whole answer (tools/fuzz_parity.py seed 402914: 22 %).
// When S is rank one to fp32 resolution -- one pair in range, the case that matters -- its factor is recovered,
// n_i = S_ik / sqrt(S_kk tr S) from the dominant column k, m = tr S, and S c is formed as m (n . c) n: the projection is then
// computed ONCE and squared, as the reference does.  Used where the set has no positive-definite identity leaf (QuadHdr::rank1)
// and by the lane-per-robot kernel.
struct RankOne {
  float n[3], tr;
  bool on;
};
__device__ __forceinline__ RankOne rank_one_of(const float S[6]) {
  RankOne r;
  const float tr = S[0] + S[3] + S[5];
  const float f2 = S[0] * S[0] + S[3] * S[3] + S[5] * S[5] + 2.f * (S[1] * S[1] + S[2] * S[2] + S[4] * S[4]);
  // 2 lambda_1 lambda_2 <= 4e-7 tr^2 (NaN compares false).  tr > 1e-15: below that tr^2 and dk tr leave fp32's normal range -- the
  // test would read 0 <= 0, the normalisation rsqrt(0) (fuzz seeds 33 / 200016 / 300032 once the velocity gate 1 - sigmoid stopped
  // rounding to an exact 0: a pair's weight of 1e-24 turned its robot into NaN) -- and a metric that small takes the general form
  r.on = tr > 1e-15f && (tr * tr - f2) <= 4e-7f * (tr * tr);
  const bool k0 = S[0] >= S[3] && S[0] >= S[5], k1 = !k0 && S[3] >= S[5];
  const float dk = k0 ? S[0] : (k1 ? S[3] : S[5]);
  const float inv = r.on ? rsqrtf(dk * tr) : 0.f;
  r.n[0] = (k0 ? S[0] : (k1 ? S[1] : S[2])) * inv;
  r.n[1] = (k0 ? S[1] : (k1 ? S[3] : S[4])) * inv;
  r.n[2] = (k0 ? S[2] : (k1 ? S[4] : S[5])) * inv;
  r.tr = tr;
  return r;
}
// u = S c, in the rank-one form where it applies
__device__ __forceinline__ void metric_times_column(const float S[6], const RankOne& r1, const float c[3], float u[3]) {
  u[0] = S[0] * c[0] + S[1] * c[1] + S[2] * c[2];
  u[1] = S[1] * c[0] + S[3] * c[1] + S[4] * c[2];
  u[2] = S[2] * c[0] + S[4] * c[1] + S[5] * c[2];
  const float a = r1.tr * (r1.n[0] * c[0] + r1.n[1] * c[1] + r1.n[2] * c[2]);
  u[0] = r1.on ? a * r1.n[0] : u[0];
  u[1] = r1.on ? a * r1.n[1] : u[1];
  u[2] = r1.on ? a * r1.n[2] : u[2];
}

template <int N>
__device__ __forceinline__ void pull_position(const float (&col)[N][3], uint32_t active, const float S[6],
                                              const float h[3], double (&Ms)[N * (N + 1) / 2], double (&fv)[N]) {
  const RankOne r1 = rank_one_of(S);  // (the lane-per-robot kernel: always)
#pragma unroll
  for (int j = 0; j < N; ++j) {
    if (!((active >> j) & 1u)) continue;  // wave-uniform: dof j does not move this frame
    fv[j] += (double)dot3(col[j], h);
    float t[3];
    metric_times_column(S, r1, col[j], t);
#pragma unroll
    for (int i = 0; i <= j; ++i)
      if ((active >> i) & 1u) Ms[sym_idx<N>(i, j)] += (double)dot3(col[i], t);
  }
}

}  // namespace rmp2
