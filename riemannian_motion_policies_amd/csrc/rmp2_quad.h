// rmp2_quad.h -- the throughput control-step kernel: FOUR LANES (one DPP quad) PER ROBOT (fleets beyond 8 192 robots;
// rmp2_hex.h below that, see dispatch_solve in rmp2_hip.hip).
//
// Why a quad: at the fleet sizes that matter (8 192 .. 65 536+ robots per GPU) a
// lane-per-robot mapping gives at most one wave per SIMD and every step is one long dependent
// instruction stream (measured: 22 us for the 3-leaf Panda set, 183 us for the cluttered set,
// identical at R = 4 096 and R = 65 536 -- pure latency).  Spreading one robot over a quad
//   * shortens that stream 3-4x (each lane does a quarter of the pair loop, a third of the
//     vector algebra of the tree walk, a third of the rows of the metric and of the LU),
//   * gives 4x more waves to fill the 1024 SIMDs, and
//   * keeps all cross-lane traffic on DPP quad_perm moves (full-rate VALU, no LDS, no
//     ds_bpermute): broadcasts of a pivot row, 3-vector gathers, 4-lane butterfly sums.
//
// Lane roles inside a quad (sub = lane & 3):
//   tree walk      sub 0..2 carry the x/y/z COMPONENT of every world vector (p, w, alpha, v, a,
//                  joint axis) and ROW sub of the world rotation; cross products fetch the two
//                  other components with quad rotations.  sub 3 idles through the walk.
//   pair loop      sphere / pair b is handled by sub = b & 3; the 3x3 metric sum S and the force
//                  sum h are combined with a 2-step xor butterfly.
//   metric, LU     row i of the n x n fp64 system lives in lane sub = i & 3 (local row i >> 2);
//                  elimination step k broadcasts row k from its owner.
//
// Numerics: fp32 leaves / Jacobians / pull-back products, fp64 accumulation and resolve, as in
// the reference (rmp.py:136-154).  Reciprocals, rsqrt and exp in the per-pair code use the
// hardware approximations refined to <= 1 ulp (Newton step / compensated argument); sin/cos
// use a Cody-Waite + minimax kernel (<= 1 ulp for |q| <= 8192, ocml beyond).
#pragma once
#include <type_traits>
#include "rmp2_device.h"
#include "rmp2_solve.h"

#ifndef RMP2_EXPLICIT_WINDOW
#define RMP2_EXPLICIT_WINDOW 2  // explicit pairs, register-capped builds: slots loaded this many ahead of their evaluation
#endif
#ifndef RMP2_EXPLICIT_LOCAL
#define RMP2_EXPLICIT_LOCAL 1  // explicit pairs: every lane evaluates the pairs it loaded (0: compacted + re-fetched; A/B)
#endif
#ifndef RMP2_IDENT_FIRST
#define RMP2_IDENT_FIRST 0  // measured: no gain (43.7 vs 43.6 us at 65 536 robots, 163.1 vs 160.2 at 262 144; tools/experiments/README.md)
#endif

namespace rmp2 {

constexpr int kQuad = 4;
constexpr int kRobotsPerWave = kWave / kQuad;  // 16
constexpr int kLdsSpheres = 256;               // sphere table staged in LDS up to this size

// ---- DPP quad permutes -----------------------------------------------------------------
constexpr int quad_perm(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
constexpr int kBcast0 = quad_perm(0, 0, 0, 0), kBcast1 = quad_perm(1, 1, 1, 1), kBcast2 = quad_perm(2, 2, 2, 2),
              kBcast3 = quad_perm(3, 3, 3, 3);
constexpr int kRot1 = quad_perm(1, 2, 0, 3);  // lane i reads component (i+1) % 3
constexpr int kRot2 = quad_perm(2, 0, 1, 3);  // lane i reads component (i+2) % 3
constexpr int kXor1 = quad_perm(1, 0, 3, 2), kXor2 = quad_perm(2, 3, 0, 1);

// Every control used through these helpers (quad_perm, row_mirror, row_half_mirror) reads an IN-BOUNDS lane for every
// lane, so "old" and bound_ctrl never decide a value -- but they decide the code: with old = the source and
// bound_ctrl = 0 the compiler must keep `old` alive in the destination (v_mov + v_mov_dpp + the consuming op, three
// instructions per use); with old = 0 and bound_ctrl = 1 it folds the permute into the consumer (v_add_f32_dpp,
// v_max_f32_dpp, v_or_b32_dpp ...: ONE instruction), or emits the bare v_mov_dpp where it cannot.
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, i, CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dppd(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int S>
__device__ __forceinline__ float bcast(float v) {
  return dpp<S * 0x55>(v);
}
template <int S>
__device__ __forceinline__ double bcastd(double v) {
  return dppd<S * 0x55>(v);
}
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp<kXor1>(v);
  v += dpp<kXor2>(v);
  return v;
}

// ---- refined hardware approximations ------------------------------------------------------
__device__ __forceinline__ float rcp1(float x) {  // 1/x, one Newton step on v_rcp_f32
  float r = __builtin_amdgcn_rcpf(x);
  return fmaf(fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ float rsq1(float x) {  // 1/sqrt(x), one Newton step on v_rsq_f32
  float r = __builtin_amdgcn_rsqf(x);
  const float e = fmaf(-x * r, r, 1.0f);
  return fmaf(0.5f * e, r, r);
}
__device__ __forceinline__ float rcp0(float x) { return __builtin_amdgcn_rcpf(x); }   // 1 ulp
__device__ __forceinline__ float rsq0(float x) { return __builtin_amdgcn_rsqf(x); }   // 1 ulp
// |v| and 1 / |v| from d2 = |v|^2 where the norm goes into a CANCELLATION: x = |p - c| - r at millimetre clearances is 1e-3 of |p - c|,
// so the ~1.5 ulp of d2 * v_rsq_f32(d2) are 1e-4 of x -- and the leaf differentiates exp(-x / 0.01) and 1 / x^2 of it.  One Newton
// step on the square root (the residual d2 - d0^2 is exact in an fma) brings the norm to the correctly rounded one within a few
// hundredths of an ulp -- what sqrtf gives the reference.  Measured on the BASELINE perf fleets (profiles/r05_accuracy_survey.txt):
// the engine's error on near-contact robots was 1.4-1.9x that of an fp32 evaluation with sqrtf at the 90th percentile; three
// more VALU instructions per in-range pair.  The reciprocal stays the 1-ulp one: it only scales the unit normal.
__device__ __forceinline__ void norm_and_inverse(float d2, float& d, float& inv) {
  inv = __builtin_amdgcn_rsqf(d2);
  const float d0 = d2 * inv;
  const float e = fmaf(-d0, d0, d2);
  d = fmaf(0.5f * inv, e, d0);  // (d2 = 0: inv = inf, d0 = NaN -- as d2 * rsq(d2) always was: a point ON the centre has no normal)
}
__device__ __forceinline__ float exp1(float x) {  // e^x: v_exp_f32 on a compensated x*log2(e)
  const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.92596299112661746e-8f;
  const float ph = x * L2E_HI;
  const float pl = fmaf(x, L2E_LO, fmaf(x, L2E_HI, -ph));
  const float e = __builtin_amdgcn_exp2f(ph);
  return fmaf(e, pl * 0.693147182464599609375f, e);
}
// sin and cos, |x| <= 8192: 3-term Cody-Waite reduction by pi/2 + degree-7/6 minimax kernels
__device__ __forceinline__ void sincos1(float x, float& sn, float& cs) {
  const float k = rintf(x * 0.636619746685028076171875f);
  float r = fmaf(-k, 1.57079637050628662109375f, x);
  r = fmaf(-k, -4.37113900018624283e-8f, r);
  r = fmaf(-k, -1.71512449810525e-15f, r);
  const float z = r * r;
  const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
  const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                        z * z, fmaf(-0.5f, z, 1.0f));
  const int n = (int)k;
  const float s0 = (n & 1) ? pc : ps;
  const float c0 = (n & 1) ? ps : pc;
  sn = (n & 2) ? -s0 : s0;
  cs = ((n + 1) & 2) ? -c0 : c0;
}

// 1/x in fp64 for the pivots of the resolve: v_rcp_f64 (measured on gfx950: 4.3e-8 relative) + ONE Newton step
// (1.9e-15 relative, ~49 bits).  The second step would buy the last 3 bits at two more dependent fp64 FMAs on the
// elimination's critical path, 1e10 below the 1e-5 parity tolerance.
__device__ __forceinline__ double rcpd(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// ---- leaves, fast-math flavour (same formulas as rmp2_device.h) --------------------------------
// rmp2.py:183-196; IP = host-precomputed {1/estd, log2(e)/rstd, 1/dstd, log2(e)/gate_len, 1/radius^2, 2/radius}
// (formed in fp64, rounded once): exp(-x/rstd) = exp2(-x * IP[1]) costs one multiply + v_exp_f32 and
// rounds the argument once, like the reference's own x / rstd.
__device__ __forceinline__ void obstacle_pair(const float* P, const float* IP, float x, float xd, float& accel,
                                              float& metric) {
  x = fmaxf(x - P[0], 0.0f);
  const float base = P[8] * rcp0(fmaf(x, IP[0], P[10]));
  const float gate = fmaf(x * x, IP[4], fmaf(-x, IP[5], 1.0f));
  const float repel = P[5] * __builtin_amdgcn_exp2f(-(x * IP[1]));
  // 1 - sigmoid(z), z = xd / gate_len, as e^-|z| / (1 + e^-|z|) for z > 0 and 1 / (1 + e^-|z|) otherwise: the reference's
  // `1. - tf.sigmoid(z)` (rmp2.py:189-194) cancels for a control point that moves AWAY from the obstacle -- at z = 7.5 the 1e-7 of a
  // rounded sigmoid are 2e-4 of the gate, which scales the pair's metric and damping (fuzz seed 2000473: a point inside a capsule, its
  // leaf's weight in M and f off by 1.1e-4 while an fp32 evaluation is only off by what its roundings happen to be); this form is
  // good to a few ulp of the gate itself, i.e. closer to the exact value of the reference's formula than its own fp32 arithmetic
  // Beyond z = 25 ln 2 = 17.33 an fp32 sigmoid IS 1 and the reference's gate an exact 0 (a robot whose only metric is such a pair
  // rests: pinv(0) 0 = 0 -- fuzz seed 300057): kept, so that a weight of 1e-30 never turns into an acceleration.
  const float z2 = xd * IP[3];                              // z log2(e)
  const float ez = __builtin_amdgcn_exp2f(-fabsf(z2));
  const float rz = rcp0(1.0f + ez);
  const float oms = (z2 > 0.0f ? (z2 > 25.0f ? 0.0f : ez) : 1.0f) * rz;
  const float damp = -oms * P[1] * xd * rcp0(fmaf(x, IP[2], P[3]));
  accel = repel + damp;
  metric = (x > P[7]) ? 0.0f : oms * (base * gate);
}

// rmp2.py:52-83 with refined hardware reciprocals / exp
__device__ __forceinline__ void target_attractor_fast(const float* P, const float x[3], const float xd[3],
                                                      const float g[3], float xdd[3], float A[6]) {
  const float kp = P[0], kd = P[1], eps = P[2], ell = P[3], amin = P[4], smax = P[5], smin = P[6], sb = P[7],
              ellb = P[8];
  float delta[3], dhat[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) delta[i] = g[i] - x[i];
  const float d2 = delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2];
  const float dn = d2 > 0.f ? d2 * rsq1(d2) : 0.f;
  const float isoft = rcp1(fmaxf(dn, eps / 10.0f));
  const float ipe = rcp1(dn + eps);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    dhat[i] = delta[i] * isoft;
    xdd[i] = kp * delta[i] * ipe - kd * xd[i];
  }
  const float sd = dn * rcp1(ell);
  const float a = (1.0f - amin) * exp1(-0.5f * sd * sd) + amin;
  const float bsd = dn * rcp1(ellb);
  const float ba = exp1(-0.5f * bsd * bsd);
  const float boost = ba * sb + (1.0f - ba) * 1.0f;
  const float wI = a * smax, wS = (1.0f - a) * smin;
  A[0] = boost * (wI + wS * (dhat[0] * dhat[0]));
  A[1] = boost * (wS * (dhat[0] * dhat[1]));
  A[2] = boost * (wS * (dhat[0] * dhat[2]));
  A[3] = boost * (wI + wS * (dhat[1] * dhat[1]));
  A[4] = boost * (wS * (dhat[1] * dhat[2]));
  A[5] = boost * (wI + wS * (dhat[2] * dhat[2]));
}

// ---- the pair loop of one distance leaf, one instantiation per obstacle mode ----------------------
constexpr int kPairsExplicit = 0, kPairsSharedLds = 1, kPairsSharedGlobal = 2, kPairsRaggedLds = 3,
              kPairsRaggedGlobal = 4;

// W = lanes that share one robot's pairs (4: quad kernel, 16: hex kernel); lane `sub` takes pairs sub, sub + W, ...
template <int MODE, bool CAP, int W = kQuad>
__device__ __forceinline__ void pair_loop(const float* sph, const float* pl, const float* po, const int32_t* ci,
                                          int count, int max_count, int sub, const float P3[3],
                                          const float V3[3], const float A3[3], const float* P, const float* IP,
                                          float S[6], float h[3], bool cyl = false) {
  // cyl (CAP builds, wave-uniform): the 8-float records are finite cylinders (rmp2_device.h point_cylinder), not capsules
  const float vv = dot3(V3, V3);
  const int trips = (max_count + W - 1) / W;  // wave-uniform trip count
  // the obstacle record of the NEXT trip is fetched while the current one is evaluated
  auto fetch = [&](int t, float4& a, float4& b2) {
    const int b_raw = W * t + sub;
    const int b = b_raw < count ? b_raw : 0;  // masked-off lanes re-read pair 0 (in bounds whenever count > 0)
    if (MODE == kPairsExplicit) {
      if (count > 0) {
        a = make_float4(pl[3 * b], pl[3 * b + 1], pl[3 * b + 2], 0.f);
        b2 = make_float4(po[3 * b], po[3 * b + 1], po[3 * b + 2], 0.f);
      } else {
        a = make_float4(1.f, 1.f, 1.f, 0.f);
        b2 = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      int sidx = b;
      if (MODE == kPairsRaggedLds || MODE == kPairsRaggedGlobal) sidx = (b_raw < count) ? ci[b] : 0;
      if (CAP) {  // 8-float capsule records
        a = reinterpret_cast<const float4*>(sph)[2 * sidx];
        b2 = reinterpret_cast<const float4*>(sph)[2 * sidx + 1];
      } else {
        a = reinterpret_cast<const float4*>(sph)[sidx];
      }
    }
  };
  float4 na, nb;
  fetch(0, na, nb);
#pragma unroll 1
  for (int t = 0; t < trips; ++t) {
    float nh[3], d;
    const bool on = W * t + sub < count;
    const float4 ca = na, cb = nb;
    fetch(min(t + 1, trips - 1), na, nb);
    if (MODE == kPairsExplicit) {
      // taskmap.py:124-129: rel = stop_gradient(p_link - p_joint); crit = p_joint + rel
      const float plk[3] = {ca.x, ca.y, ca.z}, pob[3] = {cb.x, cb.y, cb.z};
      float diff[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float rel = plk[c] - P3[c];
        const float crit = P3[c] + rel;
        diff[c] = crit - pob[c];
      }
      const float d2 = diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2];
      const float inv = rsq0(d2);
      d = d2 * inv;
#pragma unroll
      for (int c = 0; c < 3; ++c) nh[c] = diff[c] * inv;
    } else if (CAP && cyl) {  // (wave-uniform) the reference's flat-capped cylinder: nearest surface point, outward normal, signed distance
      float Yc[3];
      point_cylinder(ca, cb, P3, Yc, nh, d);
    } else {
      float ctr[3] = {ca.x, ca.y, ca.z};
      if (CAP) {  // nearest point of the capsule axis a-b to the control point (rmp2_device.h capsule_centre)
        const float u[3] = {cb.x - ca.x, cb.y - ca.y, cb.z - ca.z};
        const float w[3] = {P3[0] - ca.x, P3[1] - ca.y, P3[2] - ca.z};
        const float uu = dot3(u, u);
        float t = uu > 0.f ? dot3(w, u) * rcp1(uu) : 0.f;
        t = fminf(fmaxf(t, 0.f), 1.f);
#pragma unroll
        for (int c = 0; c < 3; ++c) ctr[c] = fmaf(t, u[c], ctr[c]);
      }
      const float diff[3] = {P3[0] - ctr[0], P3[1] - ctr[1], P3[2] - ctr[2]};
      const float d2 = diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2];
      float dn, inv;
      norm_and_inverse(d2, dn, inv);
      d = dn - ca.w;
#pragma unroll
      for (int c = 0; c < 3; ++c) nh[c] = diff[c] * inv;
    }
    const float xdot = dot3(nh, V3);
    const float cd = fmaf(-xdot, xdot, vv) * rcp0(d) + dot3(nh, A3);  // c2 + J2 c1 (taskmap.py:159)
    float acc, met;
    obstacle_pair(P, IP, d, xdot, acc, met);
    if (!on) met = 0.f;
    const float wgt = met * (acc - cd);
    const float mn[3] = {met * nh[0], met * nh[1], met * nh[2]};
    S[0] = fmaf(mn[0], nh[0], S[0]);
    S[1] = fmaf(mn[0], nh[1], S[1]);
    S[2] = fmaf(mn[0], nh[2], S[2]);
    S[3] = fmaf(mn[1], nh[1], S[3]);
    S[4] = fmaf(mn[1], nh[2], S[4]);
    S[5] = fmaf(mn[2], nh[2], S[5]);
    h[0] = fmaf(wgt, nh[0], h[0]);
    h[1] = fmaf(wgt, nh[1], h[1]);
    h[2] = fmaf(wgt, nh[2], h[2]);
  }
}

// ---- per-robot culling of the distance leaves' pairs ------------------------------------------------------------
// ObstacleAvoidance's metric is EXACTLY zero beyond metric_modulation_radius (rmp2.py:191-195: tf.where(x > r, 0, ..)),
// and a pair with zero metric contributes exactly nothing to (S, h).  In the cluttered scene 79 % of the (control
// point, sphere) pairs are out of range, yet the full pair costs 2 exp + 4 rcp + rsq + ~60 FMAs.  So:
//   pass 1  every lane tests its share of a 32-sphere chunk against the frame origin -- 3 FMAs and a compare per
//           sphere on a staged table {-2 c, |c|^2 - thr^2}: |p - c|^2 <= thr^2  <=>  p.(-2c) + (|c|^2 - thr^2) <= -|p|^2,
//           thr = (sphere radius + modulation radius + margin) with 2e-4 of slack (pairs inside the slack are
//           evaluated and zeroed by the exact test; NaNs compare "in range" and propagate as before);
//           the lanes of the robot OR their bits together: a per-(robot, frame) 32-bit mask, no LDS, no cross-robot
//           traffic (the pooled variant of round 1 died on exactly that);
//   pass 2  the set bits are dealt round-robin to the W lanes of the robot (lane s takes the s-th, (s+W)-th, ... set
//           bit) and only those pairs run the transcendental chain.
// Table image in LDS for K spheres: K x float4 {-2cx, -2cy, -2cz, w} followed by K radii.
constexpr float kCullSlack = 1.0002f;
#ifndef RMP2_TRIP_AHEAD
#define RMP2_TRIP_AHEAD 0   // measured (profiles/r05_pair_loop_ab.txt): 44.1 us against 43.8 for config 3 -- four waves per SIMD cover the read already
#endif
constexpr bool kTripAhead = RMP2_TRIP_AHEAD != 0;
#ifndef RMP2_BATCHED_TESTS
#define RMP2_BATCHED_TESTS 1
#endif
constexpr bool kBatchedTests = RMP2_BATCHED_TESTS != 0;
#ifndef RMP2_TEST_BATCH
#define RMP2_TEST_BATCH 4
#endif
#ifndef RMP2_LINK_TRIP_AHEAD
#define RMP2_LINK_TRIP_AHEAD 0   // measured with batches of eight: 79.2 against 78.7 us on config 3l -- off
#endif
constexpr bool kLinkTripAhead = RMP2_LINK_TRIP_AHEAD != 0;   // (the link-geometry loop runs at two or three waves per SIMD)
#ifndef RMP2_LINK_TEST_BATCH
#define RMP2_LINK_TEST_BATCH 4
#endif
#ifndef RMP2_BATCHED_ROW_RECS
#define RMP2_BATCHED_ROW_RECS 0   // measured: 41.9 against 41.7 us on config 3 (with the identity leaves' tile rows batched as well) -- no gain, off
#endif
constexpr bool kBatchedRowRecs = RMP2_BATCHED_ROW_RECS != 0;
__host__ __device__ constexpr int sphere_lds_floats(bool cap, int k) { return cap ? 8 * k : ((5 * k + 3) & ~3); }
// quad mapping: capsule tables keep only the four-float range-test records in LDS (bounding sphere of the capsule); the
// capsule itself (8 floats) is fetched from global memory by the lanes that evaluate an in-range pair -- 512 B for 32
// capsules instead of 1 024: the wave's LDS stays within the 10 240 B that let sixteen waves share a CU
__host__ __device__ constexpr int quad_table_floats(bool cap, int k) { return cap ? 4 * k : ((5 * k + 3) & ~3); }

__device__ __forceinline__ float4 sphere_aux(const float4 sp, float c0) {
  const float thr = fmaxf(sp.w + c0, 0.f);
  const float cc = sp.x * sp.x + sp.y * sp.y + sp.z * sp.z;
  return make_float4(-2.f * sp.x, -2.f * sp.y, -2.f * sp.z, fmaf(-thr * thr, kCullSlack, cc));
}

__device__ __forceinline__ uint32_t dpp_or(uint32_t v, uint32_t w) { return v | w; }
template <int CTRL>
__device__ __forceinline__ uint32_t dppu(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

// position of the r-th (0-based) set bit of m; r < popcount(m) required
__device__ __forceinline__ int select_bit(uint32_t m, int r) {
  int pos = 0;
  int c = __builtin_popcount(m & 0xffffu);
  if (r >= c) { pos = 16; r -= c; }
  c = __builtin_popcount((m >> pos) & 0xffu);
  if (r >= c) { pos += 8; r -= c; }
  c = __builtin_popcount((m >> pos) & 0xfu);
  if (r >= c) { pos += 4; r -= c; }
  c = __builtin_popcount((m >> pos) & 0x3u);
  if (r >= c) { pos += 2; r -= c; }
  c = (int)((m >> pos) & 1u);
  if (r >= c) pos += 1;
  return pos;
}

// RAGGED: positions index the robot's CSR list ci[0 .. count); otherwise positions ARE sphere indices
// SKIP: leave out the test slots no robot of the wave fills (a wave-uniform branch per slot: pays where waves share a SIMD,
// costs where a lone wave pays ~28 cycles per branch)
// MEMBER: the robot sees only the spheres whose bit is set in (member_lo, member_hi) -- a ragged list over a table of at
// most 64 spheres IS a membership mask: the loop then runs in the dense form (positions are sphere indices read from
// LDS) and the in-range mask is ANDed with it; no dependent global load of a list entry per test slot and per trip.
// CAPS: the primitives are capsules (`caps`: the caller's [K][8] table in global memory = (a, radius, b, -)); the staged
// records test the capsule's bounding sphere (centre = midpoint of the axis, radius = half length + capsule radius), the
// exact nearest point of the axis is formed for the in-range pairs only (one trip ahead of its use).
template <bool RAGGED, int W, bool SKIP = false, bool MEMBER = false, bool CAPS = false>
__device__ __forceinline__ void pair_loop_culled(const float* tab, int n_tab, const int32_t* ci, int count, int max_count,
                                                 int sub, const float P3[3], const float V3[3], const float A3[3],
                                                 const float* P, const float* IP, float S[6], float h[3],
                                                 unsigned long long* dbg = nullptr, uint32_t member_lo = 0u,
                                                 uint32_t member_hi = 0u, const float* caps = nullptr, bool cyl = false) {
  static_assert(!(CAPS && W != 4), "capsule culling exists in the quad mapping");
  static_assert(!(RAGGED && MEMBER), "a membership mask replaces the list");
  static_assert(W == 4 || W == 16, "quad or hex");
  const float4* aux = reinterpret_cast<const float4*>(tab);
  const float* rad = tab + 4 * n_tab;
  const float vv = dot3(V3, V3);
  const float npp = -dot3(P3, P3);
  constexpr int kTests = 32 / W;
  for (int base = 0; base < max_count; base += 32) {  // wave-uniform
    // ---- pass 1: in-range mask of this robot's chunk ----
    uint32_t m = 0u;
    if (kBatchedTests && !RAGGED && base + 32 <= count) {
      // a full chunk of a table (count is the table's size here: wave-uniform): all records read, THEN all tests -- one LDS round
      // trip per chunk instead of one per slot (the slot-wise form below puts every read in its own basic block behind a branch)
      constexpr int kBatch = kTests < RMP2_TEST_BATCH ? kTests : RMP2_TEST_BATCH;   // records in flight at a time (4 registers each)
#pragma unroll
      for (int i0 = 0; i0 < kTests; i0 += kBatch) {
        if (i0 > 0) asm volatile("" ::: "memory");   // (this batch's reads stay behind the previous batch's tests: kBatch records in flight)
        float4 a[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i) a[i] = aux[base + sub + W * (i0 + i)];
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
          const float t = fmaf(P3[0], a[i].x, fmaf(P3[1], a[i].y, fmaf(P3[2], a[i].z, a[i].w)));
          m |= !(t > npp) ? (1u << (W * (i0 + i))) : 0u;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < kTests; ++i) {
        if (SKIP && base + W * i >= max_count) continue;  // wave-uniform: no robot of the wave has a sphere in this slot
        const int pos = base + sub + W * i;
        const bool valid = pos < count;
        int sidx = valid ? pos : 0;
        if (RAGGED) sidx = valid ? ci[pos] : 0;
        const float4 a = aux[sidx];
        const float t = fmaf(P3[0], a.x, fmaf(P3[1], a.y, fmaf(P3[2], a.z, a.w)));
        const bool keep = valid && !(t > npp);
        m |= keep ? (1u << (W * i)) : 0u;
      }
    }
    m <<= sub;
    m |= dppu<kXor1>(m);
    m |= dppu<kXor2>(m);
    if (W == 16) {
      m |= dppu<0x141>(m);  // row_half_mirror
      m |= dppu<0x140>(m);  // row_mirror
    }
    if (MEMBER) m &= (base == 0 ? member_lo : member_hi);
    // ---- pass 2: lane `sub` evaluates the set bits of rank sub, sub + W, ... ----
    uint32_t rem = m;
    int rank = sub;                      // W == 16: rank of the bit this lane takes next
    const int total = __builtin_popcount(m);
    if (W == 4) {  // drop the `sub` lowest set bits; every trip then takes the lowest and drops four
      rem = sub > 0 ? (rem & (rem - 1u)) : rem;
      rem = sub > 1 ? (rem & (rem - 1u)) : rem;
      rem = sub > 2 ? (rem & (rem - 1u)) : rem;
    }
#ifdef RMP2_STAMPS
    if (dbg) dbg[1] += (unsigned long long)total;  // in-range pairs of the first robot of the wave
#endif
    // next pair of this lane: (on, index into the table); CAPS: its capsule record, fetched one trip ahead
    auto take = [&](bool& on_, int& sidx_) __attribute__((always_inline)) {
      on_ = (W == 4) ? (rem != 0u) : (rank < total);
      int j = 0;
      if (W == 4) {
        j = on_ ? (__builtin_ffs((int)rem) - 1) : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) rem &= rem - 1u;   // (0 & anything stays 0)
      } else {
        j = on_ ? select_bit(m, rank) : 0;
        rank += W;
      }
      const int pos = base + j;
      sidx_ = on_ ? pos : 0;
      if (RAGGED) sidx_ = on_ ? ci[pos] : 0;
    };
    // (sphere tables, kTripAhead -- off, measured no gain --: the staged record {-2c, w} and the radius of the NEXT trip are read from
    //  LDS before this trip's chain starts; capsule records, which come from global memory, are always fetched a trip ahead)
    bool on_n = false;
    int sidx_n = 0;
    float4 ca_n = make_float4(0.f, 0.f, 0.f, 0.f), cb_n = ca_n;
    if (CAPS) {
      take(on_n, sidx_n);
      ca_n = reinterpret_cast<const float4*>(caps)[2 * sidx_n];
      cb_n = reinterpret_cast<const float4*>(caps)[2 * sidx_n + 1];
    } else if (kTripAhead) {
      take(on_n, sidx_n);
      ca_n = aux[sidx_n];
      cb_n.x = rad[sidx_n];
    }
    while (true) {
      bool on;
      int sidx;
      float4 ca, cb;
      if (CAPS || kTripAhead) {
        on = on_n, sidx = sidx_n, ca = ca_n, cb = cb_n;
      } else {
        take(on, sidx);
      }
      if (!__any(on)) break;
#ifdef RMP2_STAMPS
      if (dbg) dbg[0] += 1ull;  // trips of the wave
#endif
      float diff[3], r;
      float d_cyl = 0.f, n_cyl[3] = {0.f, 0.f, 0.f};
      if (!CAPS && kTripAhead) {
        take(on_n, sidx_n);
        ca_n = aux[sidx_n];
        cb_n.x = rad[sidx_n];
      }
      if (CAPS) {
        take(on_n, sidx_n);
        ca_n = reinterpret_cast<const float4*>(caps)[2 * sidx_n];
        cb_n = reinterpret_cast<const float4*>(caps)[2 * sidx_n + 1];
        if (cyl) {  // (wave-uniform) finite cylinder: nearest surface point, outward normal, signed distance
          float Yc[3];
          point_cylinder(ca, cb, P3, Yc, n_cyl, d_cyl);
        }
        // nearest point of the capsule axis a-b to the control point (rmp2_device.h capsule_centre; same arithmetic as the
        // un-culled loop above)
        const float u[3] = {cb.x - ca.x, cb.y - ca.y, cb.z - ca.z};
        const float w[3] = {P3[0] - ca.x, P3[1] - ca.y, P3[2] - ca.z};
        const float uu = dot3(u, u);
        float t = uu > 0.f ? dot3(w, u) * rcp1(uu) : 0.f;
        t = fminf(fmaxf(t, 0.f), 1.f);
        const float ctr[3] = {fmaf(t, u[0], ca.x), fmaf(t, u[1], ca.y), fmaf(t, u[2], ca.z)};
        diff[0] = P3[0] - ctr[0], diff[1] = P3[1] - ctr[1], diff[2] = P3[2] - ctr[2];
        r = ca.w;
      } else {
        const float4 a = kTripAhead ? ca : aux[sidx];
        r = kTripAhead ? cb.x : rad[sidx];
        diff[0] = fmaf(0.5f, a.x, P3[0]), diff[1] = fmaf(0.5f, a.y, P3[1]), diff[2] = fmaf(0.5f, a.z, P3[2]);  // p - c, exactly
      }
      const float d2 = diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2];
      float dn, inv;
      norm_and_inverse(d2, dn, inv);
      const bool use_cyl = CAPS && cyl;
      const float d = use_cyl ? d_cyl : dn - r;
      const float nh[3] = {use_cyl ? n_cyl[0] : diff[0] * inv, use_cyl ? n_cyl[1] : diff[1] * inv, use_cyl ? n_cyl[2] : diff[2] * inv};
      const float xdot = dot3(nh, V3);
      const float cd = fmaf(-xdot, xdot, vv) * rcp0(d) + dot3(nh, A3);  // c2 + J2 c1 (taskmap.py:159)
      float acc, met;
      obstacle_pair(P, IP, d, xdot, acc, met);
      if (!on) met = 0.f;
      const float wgt = met * (acc - cd);
      const float mn[3] = {met * nh[0], met * nh[1], met * nh[2]};
      S[0] = fmaf(mn[0], nh[0], S[0]);
      S[1] = fmaf(mn[0], nh[1], S[1]);
      S[2] = fmaf(mn[0], nh[2], S[2]);
      S[3] = fmaf(mn[1], nh[1], S[3]);
      S[4] = fmaf(mn[1], nh[2], S[4]);
      S[5] = fmaf(mn[2], nh[2], S[5]);
      h[0] = fmaf(wgt, nh[0], h[0]);
      h[1] = fmaf(wgt, nh[1], h[1]);
      h[2] = fmaf(wgt, nh[2], h[2]);
    }
  }
}

// ---- link geometry in the table modes -----------------------------------------------------------------------------------
// The control point of a pair is the nearest point of the LINK's capsule (world axis LA-LB, radius lr) to the obstacle, as
// PyBullet reports it for the link's collision shape (simulation.py:462-484) -- what rmp2_closest_points_links writes out as
// p_link / p_obs for the explicit-pair step, formed here per in-range pair instead: x = |X - Y| - lr - r, n = (X - Y) / |X - Y|
// with X, Y the nearest points of the two axes (the obstacle's axis is a point for a sphere).  The derivative is the explicit
// form's: the point moves with the frame ORIGIN (V3, A3; taskmap.py:124-129).
// Range test: the staged records hold every primitive's bounding sphere (centre c = -xyz / 2; w = |c|^2 - slack (r + c0)^2);
// the link enters as the SEGMENT it is: the nearest point X of the axis to c, then |X - c| <= r + c0 + lr.  Exact for sphere
// tables, conservative for capsules (their bounding spheres): a pair beyond it is beyond metric_modulation_radius, where the
// leaf's metric is exactly 0 (rmp2.py:191-195).  (The link's own bounding sphere instead of the segment -- the cheaper test --
// keeps 2.7x the pairs: (0.5 + 0.2)^3 / 0.5^3 of the volume at the Panda's link lengths.)
// masked: the robot's ragged list as a membership mask over a table of <= 64 primitives (mem_lo: primitives 0..31, mem_hi: 32..63);
// the in-range mask of a chunk is ANDed with it -- the dense loop, masked, as pair_loop_culled<MEMBER> does for the sphere modes.
template <bool SKIP, bool CAPS>
__device__ __forceinline__ void pair_loop_link(const float* tab, int n_tab, int count, int sub, const float LA[3], const float LB[3],
                                               float lr, float c0, const float V3[3], const float A3[3], const float* P,
                                               const float* IP, float S[6], float h[3], const float* caps, bool masked = false,
                                               uint32_t mem_lo = 0u, uint32_t mem_hi = 0u) {
  constexpr int W = 4, kTests = 32 / W;
  const float4* aux = reinterpret_cast<const float4*>(tab);
  const float* rad = tab + 4 * n_tab;  // (sphere tables; a capsule table's records carry no radii)
  const float vv = dot3(V3, V3);
  const float d1[3] = {LB[0] - LA[0], LB[1] - LA[1], LB[2] - LA[2]};
  const float aa = dot3(d1, d1);
  const float inv_aa = aa > 0.f ? 1.0f / aa : 0.f;
  for (int base = 0; base < count; base += 32) {  // wave-uniform (shared table)
    uint32_t m = 0u;
    auto in_range = [&](const float4 a, const float rad_c0) __attribute__((always_inline)) {
      const float c[3] = {-0.5f * a.x, -0.5f * a.y, -0.5f * a.z};
      // r + c0 of the record: kept next to it for spheres, recovered from the record for capsules' bounding spheres
      float thr;
      if (CAPS) {
        const float cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
        thr = sqrtf(fmaxf(cc - a.w, 0.f)) * 1.00001f;  // (= sqrt(slack) (r + c0) up to rounding: widened, never under-covers)
      } else {
        thr = fmaxf(rad_c0, 0.f);
      }
      const float w[3] = {c[0] - LA[0], c[1] - LA[1], c[2] - LA[2]};
      const float sl = fminf(fmaxf(dot3(w, d1) * inv_aa, 0.f), 1.f);
      const float xc[3] = {fmaf(-sl, d1[0], w[0]), fmaf(-sl, d1[1], w[1]), fmaf(-sl, d1[2], w[2])};  // c - X
      const float lim = thr + lr;
      return !(dot3(xc, xc) > kCullSlack * lim * lim);
    };
    if (kBatchedTests && base + 32 <= count) {
      // a full chunk: the records (and radii) of four slots read together, then their tests -- one LDS round trip per batch instead
      // of one per slot behind a branch each (pair_loop_culled, where this was measured)
      constexpr int kBatch = kTests < RMP2_LINK_TEST_BATCH ? kTests : RMP2_LINK_TEST_BATCH;
#pragma unroll
      for (int i0 = 0; i0 < kTests; i0 += kBatch) {
        if (i0 > 0) asm volatile("" ::: "memory");
        float4 a[kBatch];
        float rc[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
          a[i] = aux[base + sub + W * (i0 + i)];
          rc[i] = CAPS ? 0.f : rad[base + sub + W * (i0 + i)] + c0;
        }
#pragma unroll
        for (int i = 0; i < kBatch; ++i) m |= in_range(a[i], rc[i]) ? (1u << (W * (i0 + i))) : 0u;
      }
    } else {
#pragma unroll
      for (int i = 0; i < kTests; ++i) {
        if (SKIP && base + W * i >= count) continue;
        const int pos = base + sub + W * i;
        const bool valid = pos < count;
        const int sidx = valid ? pos : 0;
        const bool keep = valid && in_range(aux[sidx], CAPS ? 0.f : rad[sidx] + c0);
        m |= keep ? (1u << (W * i)) : 0u;
      }
    }
    m <<= sub;
    m |= dppu<kXor1>(m);
    m |= dppu<kXor2>(m);
    if (masked) m &= base == 0 ? mem_lo : mem_hi;  // (tables of at most 64 primitives: two chunks)
    uint32_t rem = m;
    rem = sub > 0 ? (rem & (rem - 1u)) : rem;
    rem = sub > 1 ? (rem & (rem - 1u)) : rem;
    rem = sub > 2 ? (rem & (rem - 1u)) : rem;
    // next pair of this lane; a capsule's record comes from global memory and is fetched one trip ahead of its use
    auto take = [&](bool& on_, int& sidx_) __attribute__((always_inline)) {
      on_ = rem != 0u;
      const int j = on_ ? (__builtin_ffs((int)rem) - 1) : 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) rem &= rem - 1u;
      sidx_ = on_ ? base + j : 0;
    };
    bool on_n = false;
    int sidx_n = 0;
    float4 ca_n = make_float4(0.f, 0.f, 0.f, 0.f), cb_n = ca_n;
    if (CAPS) {
      take(on_n, sidx_n);
      ca_n = reinterpret_cast<const float4*>(caps)[2 * sidx_n];
      cb_n = reinterpret_cast<const float4*>(caps)[2 * sidx_n + 1];
    } else if (kLinkTripAhead) {  // (sphere tables: the staged record and radius of the next trip, read from LDS a trip ahead)
      take(on_n, sidx_n);
      ca_n = aux[sidx_n];
      cb_n.x = rad[sidx_n];
    }
    while (true) {
      bool on;
      int sidx;
      float4 ca, cb;
      if (CAPS || kLinkTripAhead) {
        on = on_n, sidx = sidx_n, ca = ca_n, cb = cb_n;
      } else {
        take(on, sidx);
      }
      if (!__any(on)) break;
      float X[3], Y[3], r;
      if (!CAPS && kLinkTripAhead) {
        take(on_n, sidx_n);
        ca_n = aux[sidx_n];
        cb_n.x = rad[sidx_n];
      }
      if (CAPS) {
        take(on_n, sidx_n);
        ca_n = reinterpret_cast<const float4*>(caps)[2 * sidx_n];
        cb_n = reinterpret_cast<const float4*>(caps)[2 * sidx_n + 1];
        // nearest points of the two axes: the clamped solution of the 2 x 2 normal equations (rmp2_device.h segment_segment,
        // same cases), branch-free, on refined reciprocals
        const float d2v[3] = {cb.x - ca.x, cb.y - ca.y, cb.z - ca.z};
        const float rr[3] = {LA[0] - ca.x, LA[1] - ca.y, LA[2] - ca.z};
        const float ee = dot3(d2v, d2v), ff = dot3(d2v, rr), cc = dot3(d1, rr), bb = dot3(d1, d2v);
        const float inv_e = ee > 0.f ? rcp1(ee) : 0.f;
        const float den = fmaf(aa, ee, -bb * bb);
        const float s0 = (den > 0.f && aa > 0.f) ? fminf(fmaxf(fmaf(bb, ff, -cc * ee) * rcp1(den), 0.f), 1.f) : 0.f;
        const float t0 = fmaf(bb, s0, ff) * inv_e;
        const float s_lo = fminf(fmaxf(-cc * inv_aa, 0.f), 1.f), s_hi = fminf(fmaxf((bb - cc) * inv_aa, 0.f), 1.f);
        const float sl = (t0 < 0.f || !(ee > 0.f)) ? s_lo : (t0 > 1.f ? s_hi : s0);
        const float to = fminf(fmaxf(t0, 0.f), 1.f);
#pragma unroll
        for (int c = 0; c < 3; ++c) X[c] = fmaf(sl, d1[c], LA[c]);
        Y[0] = fmaf(to, d2v[0], ca.x), Y[1] = fmaf(to, d2v[1], ca.y), Y[2] = fmaf(to, d2v[2], ca.z);
        r = ca.w;
      } else {
        const float4 a = kLinkTripAhead ? ca : aux[sidx];
        r = kLinkTripAhead ? cb.x : rad[sidx];
        Y[0] = -0.5f * a.x, Y[1] = -0.5f * a.y, Y[2] = -0.5f * a.z;  // the centre, exactly
        const float w[3] = {Y[0] - LA[0], Y[1] - LA[1], Y[2] - LA[2]};
        const float sl = fminf(fmaxf(dot3(w, d1) * inv_aa, 0.f), 1.f);
#pragma unroll
        for (int c = 0; c < 3; ++c) X[c] = fmaf(sl, d1[c], LA[c]);
      }
      const float diff[3] = {X[0] - Y[0], X[1] - Y[1], X[2] - Y[2]};
      const float d2 = diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2];
      float dn, inv0;
      norm_and_inverse(d2, dn, inv0);
      // surface to surface along the axes' common normal.  The explicit form sees the two surface points only:
      // x = |p_link - p_obs| and n = (p_link - p_obs) / x -- for overlapping capsules the points have crossed, the distance
      // reads positive and the normal points the other way (taskmap.py:126-129 on PyBullet's points); same here.
      const float sgap = dn - r - lr;
      const float d = fabsf(sgap);
      const float inv = copysignf(inv0, sgap);
      const float nh[3] = {diff[0] * inv, diff[1] * inv, diff[2] * inv};
      const float xdot = dot3(nh, V3);
      const float cd = fmaf(-xdot, xdot, vv) * rcp0(d) + dot3(nh, A3);  // c2 + J2 c1 (taskmap.py:159)
      float acc, met;
      obstacle_pair(P, IP, d, xdot, acc, met);
      if (!on) met = 0.f;
      const float wgt = met * (acc - cd);
      const float mn[3] = {met * nh[0], met * nh[1], met * nh[2]};
      S[0] = fmaf(mn[0], nh[0], S[0]);
      S[1] = fmaf(mn[0], nh[1], S[1]);
      S[2] = fmaf(mn[0], nh[2], S[2]);
      S[3] = fmaf(mn[1], nh[1], S[3]);
      S[4] = fmaf(mn[1], nh[2], S[4]);
      S[5] = fmaf(mn[2], nh[2], S[5]);
      h[0] = fmaf(wgt, nh[0], h[0]);
      h[1] = fmaf(wgt, nh[1], h[1]);
      h[2] = fmaf(wgt, nh[2], h[2]);
    }
  }
}

// Link geometry over a ragged list that cannot be a membership mask (a repeated index -- the reference would count that
// obstacle twice -- or a table beyond 64 primitives): the list walk, every listed primitive evaluated (no culling: a pair beyond
// metric_modulation_radius contributes an exact 0 through the leaf's own where()), records from global memory.
__device__ __forceinline__ void pair_loop_link_list(const float* table, bool capsule, const int32_t* ci, int count, int max_count,
                                                    int sub, const float LA[3], const float LB[3], float lr, const float V3[3],
                                                    const float A3[3], const float* P, const float* IP, float S[6], float h[3]) {
  const float vv = dot3(V3, V3);
  const float LD[3] = {LB[0] - LA[0], LB[1] - LA[1], LB[2] - LA[2]};
  const float laa = dot3(LD, LD);
  const float inv_laa = laa > 0.f ? 1.0f / laa : 0.f;
  const float zero3[3] = {0.f, 0.f, 0.f};
  for (int t = sub; t - sub < max_count; t += kQuad) {  // (wave-uniform trip count)
    const bool on = t < count;
    const int bi = on ? ci[t] : 0;
    const float4* rec = reinterpret_cast<const float4*>(table) + (capsule ? 2 * bi : bi);
    const float4 ca = rec[0];
    const float4 cb = capsule ? rec[1] : ca;
    float r_unused[3], nh[3], d;
    link_pair_fields(LA, LD, laa, inv_laa, lr, ca, cb, zero3, r_unused, nh, d);
    const float xdot = dot3(nh, V3);
    const float cd = fmaf(-xdot, xdot, vv) * rcp0(d) + dot3(nh, A3);  // c2 + J2 c1 (taskmap.py:159)
    float acc, met;
    obstacle_pair(P, IP, d, xdot, acc, met);
    if (!on) met = 0.f;
    const float wgt = met * (acc - cd);
    const float mn[3] = {met * nh[0], met * nh[1], met * nh[2]};
    S[0] = fmaf(mn[0], nh[0], S[0]);
    S[1] = fmaf(mn[0], nh[1], S[1]);
    S[2] = fmaf(mn[0], nh[2], S[2]);
    S[3] = fmaf(mn[1], nh[1], S[3]);
    S[4] = fmaf(mn[1], nh[2], S[4]);
    S[5] = fmaf(mn[2], nh[2], S[5]);
    h[0] = fmaf(wgt, nh[0], h[0]);
    h[1] = fmaf(wgt, nh[1], h[1]);
    h[2] = fmaf(wgt, nh[2], h[2]);
  }
}

// ---- explicit closest-point pairs (the reference's Datamanager layout, data_management.py:8-37), culled ---------------
// Interface B is the HBM-bound variant of the step (24 B per pair: 6 264 B per robot-step against 120 B with a shared
// table), so what the lanes do per byte decides whether the loads or the ALUs set the time.  ObstacleAvoidance's metric is
// exactly 0 beyond `thr` = margin + metric_modulation_radius (rmp2.py:191-195), as in the sphere modes:
//   pass 1  every lane loads ITS share of the leaf's pairs (pair sub, sub + 4, ...: a quad reads 48 contiguous bytes of
//           each array per slot, twelve bytes per lane) and tests |crit - p_obs|^2 against thr^2 (with slack; pairs inside
//           the slack band run the exact formula, NaNs count as in range); the quad ORs the bits into one mask per chunk of
//           32 pairs;
//   pass 2  the set bits are dealt round-robin to the quad's lanes, which fetch their pair again (an L1 / L2 hit: the
//           lines were read a few hundred cycles ago) and run the transcendental chain on it.
// FULL CHUNKS (32 pairs, the usual leaf) do not take the two passes: every lane evaluates the eight pairs it loaded, slot by
// slot, out of its own registers (RMP2_EXPLICIT_LOCAL) -- the dealt pair's second fetch is an L2 round trip per trip on the
// wave's critical path, and this mode is bound by latency, not by issue slots: 132 -> 104 us per step at 65 536 robots.
struct F3 { float x, y, z; };
// WIN > 0 (the register-capped builds): the leaf's eight slots are not loaded up front (48 registers) but in a rolling window
// -- the loads of slot i + WIN go out before slot i is evaluated, compiler barriers keep them from being hoisted --, so that the
// explicit-pair step fits the 168- / 128-register builds that put three / four waves on a SIMD.
template <int WIN = 0>
__device__ __forceinline__ void pair_loop_explicit_culled(const float* pl, const float* po, int count, int sub,
                                                          const float P3[3], const float V3[3], const float A3[3],
                                                          const float* P, const float* IP, float thr2, float S[6], float h[3]) {
  const float vv = dot3(V3, V3);
  auto diff_of = [&](int b, float diff[3]) {
    const F3 a = *reinterpret_cast<const F3*>(pl + 3 * b);
    const F3 o = *reinterpret_cast<const F3*>(po + 3 * b);
    // taskmap.py:124-129: rel = stop_gradient(p_link - p_joint); crit = p_joint + rel
    diff[0] = (P3[0] + (a.x - P3[0])) - o.x;
    diff[1] = (P3[1] + (a.y - P3[1])) - o.y;
    diff[2] = (P3[2] + (a.z - P3[2])) - o.z;
  };
  for (int base = 0; base < count; base += 32) {  // wave-uniform (the pair layout is shared by the fleet)
    uint32_t m = 0u;
    if (base + 32 <= count) {
      // full chunk (the usual case: 32 pairs per leaf): all sixteen loads are issued before the first test -- a guarded
      // loop pays one memory round trip per slot (measured: 163 us per step at 65 536 robots, latency bound)
      // (loading the NEXT leaf's pairs early -- 48 registers held through the quad sums and the pull-back -- was tried twice,
      // on the compacted loop and on this one: 184 us against 124 and 171 against 104 per step; tools/experiments/README.md)
      F3 a[8], o[8];
      if (WIN > 0) {
        auto load = [&](int i) __attribute__((always_inline)) {
          const int pos = base + sub + kQuad * i;
          a[i] = *reinterpret_cast<const F3*>(pl + 3 * pos);
          o[i] = *reinterpret_cast<const F3*>(po + 3 * pos);
        };
#pragma unroll
        for (int i = 0; i < WIN; ++i) load(i);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          asm volatile("" ::: "memory");  // (the loads of slot i + WIN stay HERE: hoisted, they would all be live at once)
          if (i + WIN < 8) load(i + WIN);
          const float diff[3] = {(P3[0] + (a[i].x - P3[0])) - o[i].x, (P3[1] + (a[i].y - P3[1])) - o[i].y,
                                 (P3[2] + (a[i].z - P3[2])) - o[i].z};
          const float d2 = diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2];
          const bool on = !(d2 > thr2);
          if (!__any(on)) continue;
          const float inv = rsq0(d2);
          const float d = d2 * inv;
          const float nh[3] = {diff[0] * inv, diff[1] * inv, diff[2] * inv};
          const float xdot = dot3(nh, V3);
          const float cd = fmaf(-xdot, xdot, vv) * rcp0(d) + dot3(nh, A3);  // c2 + J2 c1 (taskmap.py:159)
          float acc, met;
          obstacle_pair(P, IP, d, xdot, acc, met);
          if (!on) met = 0.f;
          const float wgt = met * (acc - cd);
          const float mn[3] = {met * nh[0], met * nh[1], met * nh[2]};
          S[0] = fmaf(mn[0], nh[0], S[0]);
          S[1] = fmaf(mn[0], nh[1], S[1]);
          S[2] = fmaf(mn[0], nh[2], S[2]);
          S[3] = fmaf(mn[1], nh[1], S[3]);
          S[4] = fmaf(mn[1], nh[2], S[4]);
          S[5] = fmaf(mn[2], nh[2], S[5]);
          h[0] = fmaf(wgt, nh[0], h[0]);
          h[1] = fmaf(wgt, nh[1], h[1]);
          h[2] = fmaf(wgt, nh[2], h[2]);
        }
        continue;  // (next chunk)
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int pos = base + sub + kQuad * i;
        a[i] = *reinterpret_cast<const F3*>(pl + 3 * pos);
        o[i] = *reinterpret_cast<const F3*>(po + 3 * pos);
      }
      if (RMP2_EXPLICIT_LOCAL) {
        // Every lane evaluates the pairs it LOADED, slot by slot, out of its own registers: no dealing across the quad and
        // no second fetch.  More instructions (eight masked trips per chunk instead of ~2.5 compacted ones; a slot no lane
        // of the wave has in range is skipped), but none of them waits for memory: the compacted loop's trips each paid an
        // L2 round trip for the pair they were dealt, and this mode is bound by latency, not by issue slots.
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float diff[3] = {(P3[0] + (a[i].x - P3[0])) - o[i].x, (P3[1] + (a[i].y - P3[1])) - o[i].y,
                                 (P3[2] + (a[i].z - P3[2])) - o[i].z};
          const float d2 = diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2];
          const bool on = !(d2 > thr2);
          if (!__any(on)) continue;
          const float inv = rsq0(d2);
          const float d = d2 * inv;
          const float nh[3] = {diff[0] * inv, diff[1] * inv, diff[2] * inv};
          const float xdot = dot3(nh, V3);
          const float cd = fmaf(-xdot, xdot, vv) * rcp0(d) + dot3(nh, A3);  // c2 + J2 c1 (taskmap.py:159)
          float acc, met;
          obstacle_pair(P, IP, d, xdot, acc, met);
          if (!on) met = 0.f;
          const float wgt = met * (acc - cd);
          const float mn[3] = {met * nh[0], met * nh[1], met * nh[2]};
          S[0] = fmaf(mn[0], nh[0], S[0]);
          S[1] = fmaf(mn[0], nh[1], S[1]);
          S[2] = fmaf(mn[0], nh[2], S[2]);
          S[3] = fmaf(mn[1], nh[1], S[3]);
          S[4] = fmaf(mn[1], nh[2], S[4]);
          S[5] = fmaf(mn[2], nh[2], S[5]);
          h[0] = fmaf(wgt, nh[0], h[0]);
          h[1] = fmaf(wgt, nh[1], h[1]);
          h[2] = fmaf(wgt, nh[2], h[2]);
        }
        continue;  // (next chunk)
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float dx = (P3[0] + (a[i].x - P3[0])) - o[i].x;
        const float dy = (P3[1] + (a[i].y - P3[1])) - o[i].y;
        const float dz = (P3[2] + (a[i].z - P3[2])) - o[i].z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        m |= !(d2 > thr2) ? (1u << (kQuad * i)) : 0u;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (base + kQuad * i >= count) continue;  // wave-uniform
        const int pos = base + sub + kQuad * i;
        const bool valid = pos < count;
        float diff[3];
        diff_of(valid ? pos : 0, diff);
        const float d2 = diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2];
        m |= (valid && !(d2 > thr2)) ? (1u << (kQuad * i)) : 0u;
      }
    }
    m <<= sub;
    m |= dppu<kXor1>(m);
    m |= dppu<kXor2>(m);
    uint32_t rem = m;
    rem = sub > 0 ? (rem & (rem - 1u)) : rem;
    rem = sub > 1 ? (rem & (rem - 1u)) : rem;
    rem = sub > 2 ? (rem & (rem - 1u)) : rem;
    // (the pair of the NEXT trip is fetched while the current one is evaluated: the re-fetch is an L2 hit at best)
    auto take = [&](bool& on_, float diff_[3]) {
      on_ = rem != 0u;
      const int j = on_ ? (__builtin_ffs((int)rem) - 1) : 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) rem &= rem - 1u;
      diff_of(base + j, diff_);
    };
    bool on_next;
    float diff_next[3];
    take(on_next, diff_next);
    while (true) {
      const bool on = on_next;
      if (!__any(on)) break;
      const float diff[3] = {diff_next[0], diff_next[1], diff_next[2]};
      take(on_next, diff_next);
      const float d2 = diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2];
      const float inv = rsq0(d2);
      const float d = d2 * inv;
      const float nh[3] = {diff[0] * inv, diff[1] * inv, diff[2] * inv};
      const float xdot = dot3(nh, V3);
      const float cd = fmaf(-xdot, xdot, vv) * rcp0(d) + dot3(nh, A3);  // c2 + J2 c1 (taskmap.py:159)
      float acc, met;
      obstacle_pair(P, IP, d, xdot, acc, met);
      if (!on) met = 0.f;
      const float wgt = met * (acc - cd);
      const float mn[3] = {met * nh[0], met * nh[1], met * nh[2]};
      S[0] = fmaf(mn[0], nh[0], S[0]);
      S[1] = fmaf(mn[0], nh[1], S[1]);
      S[2] = fmaf(mn[0], nh[2], S[2]);
      S[3] = fmaf(mn[1], nh[1], S[3]);
      S[4] = fmaf(mn[1], nh[2], S[4]);
      S[5] = fmaf(mn[2], nh[2], S[5]);
      h[0] = fmaf(wgt, nh[0], h[0]);
      h[1] = fmaf(wgt, nh[1], h[1]);
      h[2] = fmaf(wgt, nh[2], h[2]);
    }
  }
}

// ---- explicit pairs streamed by LDS-DMA (gfx950 global_load_lds_dwordx4), half a leaf ahead ------------------------------
// Interface B is bound by memory LATENCY at two waves per SIMD: a wave issues a leaf's sixteen loads, waits for them, then
// computes for thousands of cycles with nothing in flight (round 3: three attempts to prefetch the next leaf into REGISTERS
// spilled).  Here the next HALF leaf (16 pairs per robot: 6 KiB per wave, both arrays) flows into LDS while the current half
// is evaluated out of registers: the loads hold no VGPRs, and every wave has a chunk in flight through the pair trips, the
// quad sums and the pull-back.  One wave-instruction moves 64 x 16 B into 1 KiB of LDS, lane-linear; the source address is per
// lane: chunk c = 64 i + lane of the (robot-major) half-leaf image belongs to robot c / 12, bytes 16 (c % 12) .. +16 of its
// 192-byte segment -- every 64-byte sector of the arrays is fetched exactly once, by twelve adjacent lanes.
// Image: [16 robots][16 pairs][3 floats] of p_link at buf, the same of p_obs at buf + 768 floats.
__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// What the stream buys is NOT hidden latency -- the mode is bound by instruction issue at two waves per SIMD, not by memory
// (tools/stream_pairs.hip: the same access pattern with 110 dependent FMAs per pair streams at 5.1-5.7 TB/s on the same LDS
// footprint; the step ran at 3.9) -- but the COMPACTION it makes affordable: with a half leaf in registers and the next one in
// flight, every lane tests its four pairs, the quad ORs a 16-bit mask, and the in-range pairs (21 % in the cluttered scene) are
// written, compacted, into a small per-quad LDS list (8 entries of 16 B: the difference vector and its square; kGldsList floats per
// wave = 2 KiB), from which the quad's lanes
// take them four per trip: ~2.2 trips per half leaf instead of 4 masked slots.  (Round 3's compacted loop re-fetched the dealt
// pair from L2 -- one memory round trip per trip --, which is why the eight-slot form had won.)  A half in which some quad has
// more than 8 pairs in range (contact clusters) evaluates its four slots masked, as before.
constexpr int kGldsBuf = 1536;                       // floats: the half-leaf image, both arrays
constexpr int kGldsListCap = 8;                      // entries per quad and half leaf
constexpr int kGldsList = kRobotsPerWave * kGldsListCap * 4;  // floats: [16 quads][8 entries][diff.xyz, |diff|^2]
// pf_pb: pair_begin of the leaf whose FIRST half is already in the buffer / in flight (-1: none); updated for the next leaf.
__device__ __forceinline__ void pair_loop_explicit_glds(const float* PL, const float* PO, int n_pairs, int r0, int R, int pb,
                                                        int pb_next, int& pf_pb, float* buf, int lane, int g, int sub,
                                                        const float P3[3], const float V3[3], const float A3[3], const float* P,
                                                        const float* IP, float thr2, float S[6], float h[3], int dbg = 0) {
  // dbg (tuning builds' A/B only, 0 in the shipped paths): bit 0 = no pair is in range (the stream without the pair arithmetic),
  // bit 1 = no DMA is issued (the arithmetic on whatever the buffer holds, without the stream)
  const float vv = dot3(V3, V3);
  float* const list = buf + kGldsBuf + g * (kGldsListCap * 4);  // this quad's list
  auto issue = [&](int pbx, int half) __attribute__((always_inline)) {
    if (dbg & 2) return;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = 64 * i + lane;
      const int rr = (c * 5462) >> 16;  // c / 12 for c < 192
      const int piece = c - 12 * rr;
      const int robot = min(r0 + rr, R - 1);  // (quads beyond the fleet's tail re-read the last robot: in bounds, discarded)
      const size_t off = ((size_t)robot * n_pairs + pbx + 16 * half) * 3 + 4 * piece;
      glds16(PL + off, buf + 256 * i);
      glds16(PO + off, buf + 768 + 256 * i);
    }
  };
  auto evaluate = [&](const float diff[3], float d2, bool on) __attribute__((always_inline)) {
    const float inv = rsq0(d2);
    const float d = d2 * inv;
    const float nh[3] = {diff[0] * inv, diff[1] * inv, diff[2] * inv};
    const float xdot = dot3(nh, V3);
    const float cd = fmaf(-xdot, xdot, vv) * rcp0(d) + dot3(nh, A3);  // c2 + J2 c1 (taskmap.py:159)
    float acc, met;
    obstacle_pair(P, IP, d, xdot, acc, met);
    if (!on) met = 0.f;
    const float wgt = met * (acc - cd);
    const float mn[3] = {met * nh[0], met * nh[1], met * nh[2]};
    S[0] = fmaf(mn[0], nh[0], S[0]);
    S[1] = fmaf(mn[0], nh[1], S[1]);
    S[2] = fmaf(mn[0], nh[2], S[2]);
    S[3] = fmaf(mn[1], nh[1], S[3]);
    S[4] = fmaf(mn[1], nh[2], S[4]);
    S[5] = fmaf(mn[2], nh[2], S[5]);
    h[0] = fmaf(wgt, nh[0], h[0]);
    h[1] = fmaf(wgt, nh[1], h[1]);
    h[2] = fmaf(wgt, nh[2], h[2]);
  };
  if (pf_pb != pb) issue(pb, 0);  // (wave-uniform) first distance leaf of the step: nothing was prefetched for it
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the chunk has landed (LDS-DMA counts on vmcnt)
    F3 a[4], o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* src = buf + g * 48 + 3 * (sub + kQuad * i);
      a[i] = *reinterpret_cast<const F3*>(src);
      o[i] = *reinterpret_cast<const F3*>(src + 768);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and is in registers: the buffer is free for the next chunk
    if (half == 0)
      issue(pb, 1);
    else if (pb_next >= 0)
      issue(pb_next, 0);
    // range tests of my four pairs (taskmap.py:124-129: rel = stop_gradient(p_link - p_joint); crit = p_joint + rel)
    float df[4][3], d2s[4];
    uint32_t mine = 0u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      df[i][0] = (P3[0] + (a[i].x - P3[0])) - o[i].x;
      df[i][1] = (P3[1] + (a[i].y - P3[1])) - o[i].y;
      df[i][2] = (P3[2] + (a[i].z - P3[2])) - o[i].z;
      d2s[i] = df[i][0] * df[i][0] + df[i][1] * df[i][1] + df[i][2] * df[i][2];
      mine |= !(d2s[i] > thr2) ? (1u << i) : 0u;  // (NaN compares "in range")
    }
    if (dbg & 1) mine = 0u;
    uint32_t m = mine << (kQuad * sub);  // the quad's 16-bit mask: nibble `sub` = lane sub's four slots
    m |= dppu<kXor1>(m);
    m |= dppu<kXor2>(m);
    const int count = __builtin_popcount(m);
    if (__any(count > kGldsListCap)) {  // (wave-uniform) a contact cluster: the four slots masked, out of the registers
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool on = (mine >> i) & 1u;
        if (!__any(on)) continue;
        evaluate(df[i], d2s[i], on);
      }
      continue;
    }
    // my in-range pairs go to the quad's list at their rank in the mask (the differences, ready to use: 3 floats + d2)
    const uint32_t below = m & ((1u << (kQuad * sub)) - 1u);
    int rank = __builtin_popcount(below);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if ((mine >> i) & 1u) {
        float4* dst = reinterpret_cast<float4*>(list) + rank;  // (16-byte entries: diff.xyz, d2 -- 8 x 16 B per quad)
        *dst = make_float4(df[i][0], df[i][1], df[i][2], d2s[i]);
        ++rank;
      }
    }
    asm volatile("" ::: "memory");  // (same wave: LDS executes its accesses in order; the compiler must not reorder them)
    for (int t = 0; __any(kQuad * t < count); ++t) {
      const int e = kQuad * t + sub;
      const bool on = e < count;
      const float4 v = reinterpret_cast<const float4*>(list)[on ? e : 0];
      const float diff[3] = {v.x, v.y, v.z};
      evaluate(diff, on ? v.w : 1.0f, on);
    }
    asm volatile("" ::: "memory");  // (the list is rewritten by the next half)
  }
  pf_pb = pb_next;
}

// ---- explicit pairs streamed a QUARTER leaf at a time (the four-wave form, kObsExplicitStream) -----------------------------------
// One chunk = 8 pairs per robot of both arrays = 2 x 16 robots x 96 B = 3 KiB = exactly three wave-instructions of 64 x 16 B: piece
// c = 64 i + lane is array c / 96, robot (c % 96) / 6, bytes 16 ((c % 96) % 6) .. +16 of that robot's 96-byte segment.  Image: p_link
// [16 robots][8 pairs][3 floats] at buf, p_obs the same at buf + 384 floats.  One buffer: a chunk is read into registers (two pairs per
// lane: pair sub and sub + 4 of the eight), the NEXT chunk is issued into the same buffer at once, and the two slots are evaluated
// masked while it flies -- through the quad sums, the Jacobian columns and the pull-back of the frame too when it was the leaf's last
// chunk.  No per-quad compaction: the mode is bound by the stream, its arithmetic hides under it (measured, round 5).
constexpr int kQdmaBuf = 768;  // floats: one chunk, both arrays
// pf_pb: pair_begin of the leaf whose FIRST chunk is already in the buffer / in flight (-1: none); updated for the next leaf.
__device__ __forceinline__ void pair_loop_explicit_qdma(const float* PL, const float* PO, int n_pairs, int r0, int R, int pb, int pb_next,
                                                        int& pf_pb, float* buf, int lane, int g, int sub, const float P3[3],
                                                        const float V3[3], const float A3[3], const float* P, const float* IP,
                                                        float thr2, float S[6], float h[3], int dbg = 0) {
  const float vv = dot3(V3, V3);
  auto issue = [&](int pbx, int quarter) __attribute__((always_inline)) {
    if (dbg & 2) return;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = 64 * i + lane;
      const int arr = c >= 96 ? 1 : 0;
      const int cc = c - 96 * arr;
      const int rr = (cc * 10923) >> 16;  // cc / 6 for cc < 96
      const int piece = cc - 6 * rr;
      const int robot = min(r0 + rr, R - 1);  // (quads beyond the fleet's tail re-read the last robot: in bounds, discarded)
      const size_t off = ((size_t)robot * n_pairs + pbx + 8 * quarter) * 3 + 4 * piece;
      glds16((arr ? PO : PL) + off, buf + 256 * i);
    }
  };
  if (pf_pb != pb) issue(pb, 0);  // (wave-uniform) first distance leaf of the pass: nothing was prefetched for it
#pragma unroll
  for (int quarter = 0; quarter < 4; ++quarter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the chunk has landed (LDS-DMA counts on vmcnt)
    F3 a[2], o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float* src = buf + g * 24 + 3 * (sub + kQuad * i);
      a[i] = *reinterpret_cast<const F3*>(src);
      o[i] = *reinterpret_cast<const F3*>(src + 384);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and is in registers: the buffer is free for the next chunk
    if (quarter < 3)
      issue(pb, quarter + 1);
    else if (pb_next >= 0)
      issue(pb_next, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // taskmap.py:124-129: rel = stop_gradient(p_link - p_joint); crit = p_joint + rel
      const float diff[3] = {(P3[0] + (a[i].x - P3[0])) - o[i].x, (P3[1] + (a[i].y - P3[1])) - o[i].y, (P3[2] + (a[i].z - P3[2])) - o[i].z};
      const float d2 = diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2];
      const bool on = !(d2 > thr2) && !(dbg & 1);  // (NaN compares "in range")
      if (!__any(on)) continue;
      const float inv = rsq0(d2);
      const float d = d2 * inv;
      const float nh[3] = {diff[0] * inv, diff[1] * inv, diff[2] * inv};
      const float xdot = dot3(nh, V3);
      const float cd = fmaf(-xdot, xdot, vv) * rcp0(d) + dot3(nh, A3);  // c2 + J2 c1 (taskmap.py:159)
      float acc, met;
      obstacle_pair(P, IP, d, xdot, acc, met);
      if (!on) met = 0.f;
      const float wgt = met * (acc - cd);
      const float mn[3] = {met * nh[0], met * nh[1], met * nh[2]};
      S[0] = fmaf(mn[0], nh[0], S[0]);
      S[1] = fmaf(mn[0], nh[1], S[1]);
      S[2] = fmaf(mn[0], nh[2], S[2]);
      S[3] = fmaf(mn[1], nh[1], S[3]);
      S[4] = fmaf(mn[1], nh[2], S[4]);
      S[5] = fmaf(mn[2], nh[2], S[5]);
      h[0] = fmaf(wgt, nh[0], h[0]);
      h[1] = fmaf(wgt, nh[1], h[1]);
      h[2] = fmaf(wgt, nh[2], h[2]);
    }
  }
  pf_pb = pb_next;
}

// v[4 m + sub] of a wave-uniform per-dof vector of the program (leaf va / vb): four scalar-cache words and three
// selects instead of a lane-indexed vector load from global memory (a full memory latency in the middle of a leaf)
__device__ __forceinline__ float pick4(const float* v, int m, int sub) {
  const float a0 = v[4 * m], a1 = v[4 * m + 1], a2 = v[4 * m + 2], a3 = v[4 * m + 3];
  return sub == 0 ? a0 : (sub == 1 ? a1 : (sub == 2 ? a2 : a3));
}

// ---- the kernel -------------------------------------------------------------------------------
// LDS per wave (16 robots).  The budget that matters: 160 KiB per CU / 16 waves = 10 240 B -- above it the 16 waves
// a CU owes to a 65 536-robot fleet are not resident together (measured before the diet: 16.9 KB, 9 of 16 resident,
// the other 7 ran as a second, half-empty round).  Per robot: the q / qd rows and ONE 12-float slot per frame of the
// program.  A slot first carries the frame's local transform (phase 1 -> walk), then the frame's world quantities
// [p, v, a, z] (walk -> leaf phase): its origin with J qd and Jdot qd, and the world axis of its joint -- which IS the
// (o_j, z_j) record the Jacobian columns of dof j need (QuadHdr::dof_ops maps dof -> frame).  After the resolve slot 0
// of the robot receives qdd (nothing reads the slots of a resolved robot again).
constexpr int kSlot = 12;
template <int N>
struct QuadLds {
  static constexpr int kQ = 0;                                 // [16][N]
  static constexpr int kQd = kRobotsPerWave * N;               // [16][N]
  static constexpr int kLoc = 2 * kRobotsPerWave * N;          // [16 robots][max(n_ops, 1)][12]
  static constexpr int kFloats = kLoc;                         // + dynamic: slots, sphere table, staged program
};
__host__ __device__ constexpr int quad_slots(int n_ops) { return n_ops > 0 ? n_ops : 1; }

// header of the program, passed BY VALUE as a kernel argument (lands in SGPRs with the kernarg
// preload: the prologue needs no dependent round trip before it can issue the tile loads)
// closed-loop rollout: n_iters control steps inside ONE launch; after each resolve the
// state is advanced by `substeps` semi-implicit Euler ticks of length dt with qdd held (the reference's
// 10 Hz control / 100 Hz plant loop, 06_cluttered_environment.py:120-131).  A plain control step is
// {1, 0, 0, nullptr, nullptr}.
struct RolloutArgs {
  int32_t n_iters, substeps;
  float dt;
  float* q_out;
  float* qd_out;
  // obstacle motion inside the rollout (06_cluttered_environment.py:120-131 re-reads the obstacle data every control step):
  // floats between the tables of consecutive control steps ([n_iters][K][4 or 8] behind obs.spheres); 0 = one table
  int32_t table_stride = 0;
};

struct QuadHdr {
  int32_t n_ops, n_dof, n_id, n_leaves, goal_floats, n_leaf_ops;
  uint32_t rev_mask;
  int32_t n_levels;  // pointer-jumping rounds (rmp2_hex.h only)
  int32_t n_fk;      // leaves on FK task maps (rmp2_hex.h only)
  int32_t is_chain;  // every frame's parent is the previous frame of the program (rmp2_hex.h only)
  uint32_t dof_ops[3];  // op that owns dof j, 5 bits each, 6 dofs per word (rmp2_quad.h only)
  float cull_c0;        // max over the distance leaves of (metric_modulation_radius + margin): cull threshold
  int32_t strict;       // 1: solve = PINV.  rmp2_hex.h: the pseudo-inverse on every robot; rmp2_quad.h: the elimination's result for the
                        // robots it certifies as full rank (pinv = inv there), the Jacobi pseudo-inverse for the rest
  int32_t prio_tail;    // wave priority of the phases after the frame loop (rmp2_quad.h only): 0; RMP2_PRIO_TAIL pins another
                        // value for A/B runs (with the kernels of the middle of round 2 a fleet of many rounds preferred 2,
                        // with the final ones 0 wins at every size: tools/gpu_calls_r02/r02_run33_prio_tail.sh)
  int32_t skip_resolve; // 1 (rmp2_quad.h, general flavour): the step stops behind the combined metric / force (out.M, out.f);
                        // rmp2_pinv_kernel resolves every robot by the pseudo-inverse (solve = PINV, rank-deficient sets)
  int32_t has_point;    // the set carries attached-point leaves (full 16-float rotation records even when link geometry is given)
  int32_t rank1;        // the set has no positive-definite identity leaf: a leaf metric that is rank one is pulled back in its
                        // rank-one form (rmp2_device.h rank_one_of)
  int32_t stagger;      // streamed explicit pairs (kObsExplicitStream): the wave in slot s of its SIMD starts (s & 3) * stagger * 3.4 us
                        // late (s_sleep), so that the four waves of a SIMD are not all in their streaming phase at once
};

__device__ __forceinline__ int gi_loc(int g, int n_ops) { return g * kSlot * quad_slots(n_ops); }

// walk state of one lane: component `sub` of the world vectors, row `sub` of the rotation
struct QuadState {
  float R[3];
  float p, w, al, v, a;
};

// MINW = minimum waves per SIMD the register allocator must leave room for: 1 (up to 512
// registers, no spills) for fleets that cannot put two waves on a SIMD anyway; 2 or 3 (256 / 168 registers) for larger
// ones, chosen from the fleet size in launch_quad (rmp2_hip.hip); 4 (128 registers) is an A/B build only.
// STAGE = true: the whole program (ops, leaf records, lists) and the goal rows are copied into LDS
// in the SAME burst of loads that brings the q / qd tile on chip.  At the kernel boundary every
// XCD's L2 starts cold, so each DEPENDENT scalar/global load round costs ~1000 cycles (measured);
// a 3-leaf step has ~11 such rounds when the program is walked through the scalar cache.  With
// STAGE the step pays one round, later accesses are LDS reads (~100 cycles, prefetchable).  For
// big grids (many waves per CU share the scalar cache, latency is hidden by other waves) the
// scalar path costs fewer instructions: STAGE = false.
template <bool STAGE>
__device__ __forceinline__ int uni(int v) {
  return STAGE ? __builtin_amdgcn_readfirstlane(v) : v;
}

// SYM = every leaf of the set has a symmetric metric (no JointLimitAvoidance, quirk Q2; decided at rmp2_create): the
// system stays in block-upper form through the identity leaves and the elimination.
// OBS = the obstacle mode the instantiation is compiled for (kObsAny: resolved at run time); PLAIN = a single control step
// without the debug outputs (M, f) and without the rollout loop.  The specialised throughput builds carry none of the
// other modes' code: no per-lane 64-bit addresses of pair arrays, CSR lists, debug rows or rollout outputs for the
// compiler to hoist into the prologue and park in scratch (what the 128-register build spilled in round 2).
constexpr int kObsAny = -1;
// OBS = kObsSharedLink (plain builds only): a shared primitive table with LINK GEOMETRY for the distance leaves
// (rmp2_obstacles.link_capsules) in a lean build -- no rotation record per frame (the attached-record builds: 12 floats x every
// frame, 256 registers, two waves per SIMD), but the link's world SEGMENT, formed in the walk for the frames that carry a distance
// leaf (8 floats per leaf frame): 15.6 KB of LDS per wave for the Panda, ten waves per CU at 168 registers.
constexpr int kObsSharedLink = 4;
constexpr int kLinkSeg = 8;  // floats per (robot, leaf frame): A.xyz, -, B.xyz, -
// OBS = kObsExplicitStream (plain builds, four waves per SIMD): interface B -- explicit closest-point pairs, 6 264 B per robot-step,
// the one HBM-bound form of the step -- with the pair arrays STREAMED through LDS beside a four-wave working set.  The two-wave form keeps
// the frame records (8.4 KB of a wave's LDS) and 256 registers through the frame loop; at two waves per SIMD the step is bound by the
// latencies two waves cannot cover (round 4: 104.5 us at 65 536 robots = 49 % of the HBM roof; six loader variants, none faster).  Here,
// after the walk, what the frame loop needs of the records leaves their LDS: the (z, o) records of the joints that own a lane's rows go
// to registers (18), [p v a] of the leaf-bearing frames to a compact image (nine floats per robot and frame: 5.2 KB), and the rest of the
// region becomes a 3 KiB LDS-DMA buffer through which every leaf's pairs arrive a quarter leaf at a time (pair_loop_explicit_qdma):
// the next chunk is in flight while the current one is evaluated, summed, and -- at a leaf's end -- pulled back.  128 registers and 9.6 KB
// of LDS per wave: sixteen waves per CU in ONE round at 65 536 robots, each with a chunk in flight from its first leaf to its last.
// (Round 5 first built the two-phase form -- all pair phases, then all pull-backs: tools/experiments/r05_explicit_two_phase.patch --,
// whose stream ran at 6 TB/s but in series with 56 us of walk and pull-back: every wave of a one-round fleet is in the same phase.)
constexpr int kObsExplicitStream = 5;
constexpr int kStreamMaxFrames = 9;  // leaf-bearing frames of the compact [p v a] image (the Panda's cluttered set: 8 + 1)
constexpr int kStreamPva = kRobotsPerWave * kStreamMaxFrames * 9;  // floats of that image; the chunk buffer (kQdmaBuf) sits behind it
// FLAVOR: kGeneral = everything at run time (debug outputs M / f, rollout loop, any obstacle mode); kPlainStep = one control
// step, no debug outputs, OBS fixed; kPlainRollout = the fused rollout loop, no debug outputs, OBS fixed (sphere-table modes).
constexpr int kGeneral = 0, kPlainStep = 1, kPlainRollout = 2;
// PT: the set carries attached-point leaves ([FK, TaskmapRelative4x4, 4x4 -> position] + CollisionAvoidance, taskmap.py:79-99,
// rmp.py:264-315): every frame additionally leaves its world rotation, angular velocity and angular bias acceleration in LDS
// (16 floats per robot and frame behind the other regions).
constexpr int kPtSlot = 16;      // per-frame record of the attached-point builds: rows of the world rotation (9), w (3), alpha (3)
constexpr int kPtSlotLink = 12;  // link geometry needs the rotation only: 9 + 3 of padding -- with it a wave's LDS stays under
                                 // 20 KB for the Panda and eight waves share a CU (16-float records: seven, i.e. three rounds
                                 // instead of two for 65 536 robots)
template <int N, int SLOTS, int MINW, bool STAGE, bool CAP, bool SYM = false, int OBS = kObsAny, int FLAVOR = kGeneral,
          bool PT = false>
__device__ __forceinline__ void quad_step_body(const DevProgram* __restrict__ prog, const QuadHdr& hdr,
                                               const float* __restrict__ q, const float* __restrict__ qd,
                                               const float* __restrict__ goal, int goal_stride, const ObsArgs& obs, OutArgs out,
                                               const RolloutArgs& ro_arg, int R, const int block_idx) {
  constexpr bool LINKSEG = OBS == kObsSharedLink;
  static_assert(!LINKSEG || (FLAVOR == kPlainStep && !PT && !STAGE), "link segments: the lean plain builds");
  constexpr bool STREAM = OBS == kObsExplicitStream;
  static_assert(!STREAM || (FLAVOR == kPlainStep && !PT && !STAGE && !CAP && N == 9 && MINW == 4), "streamed explicit pairs: the plain four-wave build");
  const int obs_mode = OBS == kObsAny ? obs.mode : (LINKSEG ? RMP2_OBS_SHARED_SPHERES : (STREAM ? RMP2_OBS_EXPLICIT_PAIRS : OBS));
  constexpr bool PLAIN = FLAVOR == kPlainStep;   // no rollout loop
  constexpr bool LEAN = FLAVOR != kGeneral;      // no debug outputs
  // the rank-one pull-back of sets without an inertia leaf (QuadHdr::rank1) is compiled into every build such a set can reach:
  // the general flavour (two-kernel step, debug outputs), the rollout builds, the 2-dof builds -- not into the plain step of the
  // 3..9-dof template, which dispatch_solve never gives them (measured: the two wave-uniform branches per leaf frame and the
  // longer live ranges cost the headline kernel 2.5 %)
  constexpr bool kRank1 = FLAVOR != kPlainStep || N == 2;
  const RolloutArgs ro = PLAIN ? RolloutArgs{1, 0, 0.f, nullptr, nullptr, 0} : ro_arg;
#ifdef RMP2_STAMPS
  if (LEAN) out.M = nullptr;  // (diagnostic build: the stamps travel behind the f rows, so the plain build keeps that pointer)
#else
  if (LEAN) out.M = nullptr, out.f = nullptr;
#endif
  constexpr int ROWS = (N + kQuad - 1) / kQuad;  // local rows of the n x n system per lane
  static_assert(!PT || (MINW < 4 && FLAVOR == kGeneral), "attached-point leaves / link geometry: general flavour, row records in registers");
  constexpr bool kIdentFirst = SYM && PLAIN && MINW >= 3 && RMP2_IDENT_FIRST;  // (see "Phase order per wave" below)
  // dynamic LDS: [QuadLds<N>::kFloats floats | frame slots 16 robots x max(n_ops, 1) x 12 floats |
  //               sphere table min(K, 256) x 4 |
  //               STAGE only: ops[n_ops] | leaves[n_leaves] | fk list | id list | goal tile 16 x 16 floats]
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef RMP2_STAMPS
  // diagnostic build only: shader-clock stamps per phase, written to a buffer nothing else reads
  unsigned long long st_[8];
  int st_n = 0;
#define RMP2_STAMP() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); st_[st_n++] = __builtin_amdgcn_s_memtime(); } while (0)
  // ... and accumulated time per segment of the FK-leaf loop (row entries 8..15 of the stamp buffer)
  unsigned long long seg_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, seg_t = 0;
#define RMP2_SEG_BEGIN() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); seg_t = __builtin_amdgcn_s_memtime(); } while (0)
#define RMP2_SEG(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg_[i] += t_ - seg_t; seg_t = t_; } while (0)
#else
#define RMP2_STAMP() do {} while (0)
#define RMP2_SEG_BEGIN() do {} while (0)
#define RMP2_SEG(i) do {} while (0)
#endif
  RMP2_STAMP();
  // Wave priority falls as the step progresses: s_setprio 3 through the walk and the first third of the leaf frames, 2 in
  // the second third, 1 in the last, 0 after the frame loop (hdr.prio_tail).  The SIMD's arbiter otherwise serves its oldest
  // wave first: waves that started together finish staggered and the last one runs alone, at a fraction of the issue
  // rate.  With the lagging wave preferred they finish together (measured when it was introduced: 68.5 -> 66.9 us at
  // 65 536 robots with two rounds of two waves, 51.5 -> 49.1 at 49 152 with one round of three, 73.0 -> 68.5 for the
  // 128-register build; thirds instead of halves of the frame loop: another 52.5 -> 51.5).
  __builtin_amdgcn_s_setprio(3);
  if constexpr (OBS == kObsExplicitStream) {
    // All 4 096 waves of a 65 536-robot fleet are resident at once and would walk, stream and pull back IN STEP: the memory system
    // idle through everybody's walk, saturated through everybody's pair phase (measured: 53 % of a wave's life in s_waitcnt with
    // 25 MB in flight across the chip, 116 us per step), idle again through the pull-backs.  The four waves of a SIMD start a
    // quarter of a streaming phase apart instead: while one streams, its neighbours walk or pull back.
    const int slot = __builtin_amdgcn_s_getreg(0x1804) & 3;  // HW_ID[3:0]: the wave's slot in its SIMD
    for (int i = 0; i < slot * (hdr.stagger & 255); ++i) __builtin_amdgcn_s_sleep(127);
  }
  if constexpr (OBS == RMP2_OBS_EXPLICIT_PAIRS && FLAVOR == kPlainStep && MINW == 2) {
    // The same question for the two-wave form (two ROUNDS at 65 536 robots): the odd slot of every SIMD starts its first wave late,
    // so that one wave of a SIMD streams while the other walks or pulls back, and the offset carries into the second round by itself
    // (a slot's next wave starts when its first ends).  Only the first round sleeps.  RMP2_STREAM_STAGGER=n, opt-in.
    if ((hdr.stagger & 255) != 0 && block_idx < MINW * 1024 && (__builtin_amdgcn_s_getreg(0x1804) & 1))
      for (int i = 0; i < (hdr.stagger & 255); ++i) __builtin_amdgcn_s_sleep(127);
  }
  const int lane = threadIdx.x;
  const int sub = lane & 3;
  const int g = lane >> 2;
  const int r0 = block_idx * kRobotsPerWave;
  const int robot = r0 + g;
  const bool live = robot < R;
  const int n_dof = hdr.n_dof;

  // ---- stage the q / qd tile (coalesced) and the sphere table in LDS --------------------------
  const int n_ops = hdr.n_ops, n_id = hdr.n_id;
  const int n_sph_lds = (obs_mode == RMP2_OBS_SHARED_SPHERES || obs_mode == RMP2_OBS_RAGGED_SPHERES)
                            ? min(obs.n_spheres, kLdsSpheres) : 0;
  float* const sph_lds_base = lds + QuadLds<N>::kFloats + kSlot * kRobotsPerWave * quad_slots(n_ops);  // 16-byte aligned
  const uint32_t rev_mask = hdr.rev_mask;
  // staged copies (STAGE) live behind the local-transform records
  float* const stage_base = sph_lds_base + quad_table_floats(CAP, n_sph_lds);
  DevOp* const s_ops = reinterpret_cast<DevOp*>(stage_base);
  DevLeaf* const s_leaves = reinterpret_cast<DevLeaf*>(s_ops + n_ops);
  int32_t* const s_fk = reinterpret_cast<int32_t*>(s_leaves + hdr.n_leaves);
  int32_t* const s_id = s_fk + RMP2_MAX_LEAVES;
  int32_t* const s_lo = s_id + RMP2_MAX_LEAVES;
  float* const s_goal = reinterpret_cast<float*>(s_lo + kMaxOps);
  // attached-point builds: [16 robots][n_ops][16] = rows of the world rotation (9), w (3), alpha (3), behind everything else
  float* const pt_base = STAGE ? s_goal + 16 * kRobotsPerWave : stage_base;
  const int pt_slot = (PT && obs.link_caps && !hdr.has_point) ? kPtSlotLink : kPtSlot;  // (wave-uniform)
  const DevOp* const ops = STAGE ? s_ops : prog->ops;
  const DevLeaf* const leaves = STAGE ? s_leaves : prog->leaves;
  const int32_t* const fk_list = STAGE ? s_fk : prog->fk_leaves;
  const int32_t* const id_list = STAGE ? s_id : prog->id_leaves;
  const int32_t* const leaf_ops = STAGE ? s_lo : prog->leaf_ops;
  const int n_live = min(kRobotsPerWave, R - r0);
  // the obstacle table's LDS image (no barrier inside: the caller owns the ordering against its readers)
  auto stage_table = [&](const float* table) __attribute__((always_inline)) {
    if (CAP) {  // range-test records of the capsules' bounding spheres (the capsules themselves stay in global memory)
      for (int i = lane; i < n_sph_lds; i += kWave) {
        const float4 ca = reinterpret_cast<const float4*>(table)[2 * i];
        const float4 cb = reinterpret_cast<const float4*>(table)[2 * i + 1];
        if (obs.cylinder) {  // (wave-uniform) finite cylinder (centre, radius, axis, half height): centre, sqrt(r^2 + h^2)
          const float br = sqrtf(ca.w * ca.w + cb.w * cb.w) * 1.000001f;
          reinterpret_cast<float4*>(sph_lds_base)[i] = sphere_aux(make_float4(ca.x, ca.y, ca.z, br), hdr.cull_c0);
          continue;
        }
        const float hx = 0.5f * (cb.x - ca.x), hy = 0.5f * (cb.y - ca.y), hz = 0.5f * (cb.z - ca.z);
        const float hl = sqrtf(hx * hx + hy * hy + hz * hz) * 1.000001f;  // (rounded up: the test must never under-cover)
        reinterpret_cast<float4*>(sph_lds_base)[i] =
            sphere_aux(make_float4(ca.x + hx, ca.y + hy, ca.z + hz, ca.w + hl), hdr.cull_c0);
      }
    } else {  // image for the culled pair loop: {-2c, |c|^2 - thr^2} records, then the radii
      for (int i = lane; i < n_sph_lds; i += kWave) {
        const float4 sp = reinterpret_cast<const float4*>(table)[i];
        reinterpret_cast<float4*>(sph_lds_base)[i] = sphere_aux(sp, hdr.cull_c0);
        sph_lds_base[4 * n_sph_lds + i] = sp.w;
      }
    }
  };
  {
    const int tile = n_live * n_dof;
    const float* gq = q + (size_t)r0 * n_dof;
    const float* gqd = qd + (size_t)r0 * n_dof;
    for (int i = lane; i < tile; i += kWave) {
      const int rr = i / n_dof, jj = i - rr * n_dof;
      float qv = gq[i], qdv = gqd[i];
      quarantine(qv, qdv);
      lds[QuadLds<N>::kQ + rr * N + jj] = qv;
      lds[QuadLds<N>::kQd + rr * N + jj] = qdv;
    }
    if (n_dof < N) {  // padding dofs of the template read as q = qd = 0
      const int pad = N - n_dof;
      for (int i = lane; i < kRobotsPerWave * pad; i += kWave) {
        const int rr = i / pad, jj = n_dof + (i - rr * pad);
        lds[QuadLds<N>::kQ + rr * N + jj] = 0.f;
        lds[QuadLds<N>::kQd + rr * N + jj] = 0.f;
      }
    }
    if (obs_mode == RMP2_OBS_SHARED_SPHERES || obs_mode == RMP2_OBS_RAGGED_SPHERES) stage_table(obs.spheres);
    if (STAGE) {
      const uint4* src = reinterpret_cast<const uint4*>(prog->ops);
      uint4* dst = reinterpret_cast<uint4*>(s_ops);
      for (int i = lane; i < n_ops * (int)(sizeof(DevOp) / 16); i += kWave) dst[i] = src[i];
      src = reinterpret_cast<const uint4*>(prog->leaves);
      dst = reinterpret_cast<uint4*>(s_leaves);
      for (int i = lane; i < hdr.n_leaves * (int)(sizeof(DevLeaf) / 16); i += kWave) dst[i] = src[i];
      if (lane < RMP2_MAX_LEAVES) {
        s_fk[lane] = prog->fk_leaves[lane];
        s_id[lane] = prog->id_leaves[lane];
      }
      if (lane < kMaxOps) s_lo[lane] = prog->leaf_ops[lane];
      if (goal) {
        const int gf = hdr.goal_floats;  // <= 16 (checked on the host)
        for (int i = lane; i < n_live * gf; i += kWave) {
          const int rr = i / gf, jj = i - rr * gf;
          s_goal[rr * 16 + jj] = goal[(size_t)(r0 + rr) * goal_stride + jj];
        }
      }
    }
    __syncthreads();
  }
  RMP2_STAMP();  // 1: prologue done
  // quads beyond the fleet's tail re-use the last live robot's inputs (results are discarded)
  const int gi = min(g, n_live - 1);
  const bool spheres_in_lds = obs.n_spheres <= kLdsSpheres;
  const float* my_q = &lds[QuadLds<N>::kQ + gi * N];
  const float* my_qd = &lds[QuadLds<N>::kQd + gi * N];
  float* const loc = &lds[QuadLds<N>::kLoc + gi_loc(g, n_ops)];  // this robot's frame slots
  float* my_out = loc;                                            // qdd lands in slot 0 once the robot is resolved
  const int out_stride = kSlot * quad_slots(n_ops);
  const float* my_goal = !goal ? nullptr : (STAGE ? s_goal + gi * 16 : goal + (size_t)(live ? robot : 0) * goal_stride);
  uint32_t status = 0u;
  bool flagged = false;
  // explicit pairs by LDS-DMA (pair_loop_explicit_glds): the plain two-wave build of the explicit-pair mode only -- its launch
  // carries the 6 KiB chunk buffer behind the other regions (stage_base: no staged program, no sphere table in this build)
  constexpr bool kGlds = OBS == RMP2_OBS_EXPLICIT_PAIRS && FLAVOR == kPlainStep && MINW == 2 && !STAGE && !PT;
  int pf_pb = -1;  // (wave-uniform) pair_begin of the leaf whose first half is in the chunk buffer / in flight

  // ---- ragged lists over a small table: the robot's list as a membership mask (built once, by its quad) -----------------
  // Each lane reads every fourth entry of the robot's list (the wave's 16 lists are contiguous in the CSR array), sets the
  // bits and the quad ORs them.  A list with a repeated index cannot be a mask (the reference would count that obstacle
  // twice): such a wave keeps the list walk.
  uint32_t member_lo = 0u, member_hi = 0u;
  bool use_member = false;
  if (obs_mode == RMP2_OBS_RAGGED_SPHERES && obs.n_spheres <= 64) {
    const int rr_ = live ? robot : 0;
    const int b0 = obs.csr_offset[rr_];
    const int count = live ? obs.csr_offset[rr_ + 1] - b0 : 0;
    int max_count = count;
#pragma unroll
    for (int o = 32; o >= kQuad; o >>= 1) max_count = max(max_count, __shfl_xor(max_count, o));
    bool bad = false;
    for (int t = sub; t - sub < max_count; t += kQuad) {  // (wave-uniform trip count)
      if (t < count) {
        const int idx = obs.csr_index[b0 + t];
        bad = bad || idx < 0 || idx >= obs.n_spheres;
        if (idx >= 0 && idx < 32) member_lo |= 1u << idx;
        if (idx >= 32 && idx < 64) member_hi |= 1u << (idx - 32);
      }
    }
    member_lo |= dppu<kXor1>(member_lo);
    member_lo |= dppu<kXor2>(member_lo);
    member_hi |= dppu<kXor1>(member_hi);
    member_hi |= dppu<kXor2>(member_hi);
    bad = bad || (__builtin_popcount(member_lo) + __builtin_popcount(member_hi) != count);
    use_member = !__any(bad);
  }

#pragma nounroll
  for (int it = 0; it < ro.n_iters; ++it) {
  __builtin_amdgcn_s_setprio(3);  // (every control step of a fused rollout starts over)
  flagged = false;
  // moving obstacles: control step `it` reads table `it` (one wave per block: the LDS image is the wave's own; its readers of
  // the previous step are done -- same wave, program order -- and the barrier of phase 1 below publishes the new image)
  const float* const step_table = obs.spheres + (size_t)it * (size_t)(!PLAIN ? ro.table_stride : 0);
  if (!PLAIN && ro.table_stride != 0 && it > 0 &&
      (obs_mode == RMP2_OBS_SHARED_SPHERES || obs_mode == RMP2_OBS_RAGGED_SPHERES))
    stage_table(step_table);
  // ---- phase 1: local transforms T_constant @ T_variable(q) of ALL frames, in parallel -------
  // (kinematics.py:222-240).  They do not depend on the chain, so lane `sub` of the quad builds
  // the frames k = sub, sub+4, ... (branch-free: the joint type selects by arithmetic) and
  // leaves Rl (9), tl (3) and Rl @ axis (3) in LDS; the serial walk below only multiplies.
  for (int k = sub; k < n_ops; k += kQuad) {
    const DevOp& opg = ops[k];  // lane-dependent record (LDS copy when staged, else vector loads)
    const int jt = opg.jtype, qi = opg.qidx;
    const float qv = qi >= 0 ? my_q[qi] : 0.f;
    const float ax[3] = {opg.axis[0], opg.axis[1], opg.axis[2]};
    float sn = 0.f, cs = 1.f;
    if (jt == RMP2_JOINT_REVOLUTE) {
      if (fabsf(qv) <= 8192.0f)
        sincos1(qv, sn, cs);
      else
        sincosf(qv, &sn, &cs);
    }
    const float omc = 1.0f - cs;
    const bool rev = jt == RMP2_JOINT_REVOLUTE;
    // Rodrigues: cos*I + sin*[u]x + (1-cos)*u u^T ; for non-revolute joints T_variable's rotation is I exactly
    const float ut[9] = {0.f, -ax[2], ax[1], ax[2], 0.f, -ax[0], -ax[1], ax[0], 0.f};
    float Rv[9], Tc[12], rec[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) Tc[c] = opg.Tc[c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float e = (r == c) ? 1.f : 0.f;
        Rv[3 * r + c] = rev ? cs * e + sn * ut[3 * r + c] + omc * (ax[r] * ax[c]) : e;
      }
    const float tq = (jt == RMP2_JOINT_PRISMATIC) ? qv : 0.f;
    const float tv[3] = {tq * ax[0], tq * ax[1], tq * ax[2]};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        rec[3 * r + c] = Tc[4 * r + 0] * Rv[c] + Tc[4 * r + 1] * Rv[3 + c] + Tc[4 * r + 2] * Rv[6 + c];
      rec[9 + r] = Tc[4 * r + 0] * tv[0] + Tc[4 * r + 1] * tv[1] + Tc[4 * r + 2] * tv[2] + Tc[4 * r + 3];
    }
    float4* dst = reinterpret_cast<float4*>(loc + kSlot * k);
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[c] = make_float4(rec[4 * c], rec[4 * c + 1], rec[4 * c + 2], rec[4 * c + 3]);
  }
  __syncthreads();

  // fp64 system, row-distributed: local row m holds global row i = sub + 4 m
  double A[ROWS][N];
  double fv[ROWS];
  auto zero_system = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < ROWS; ++m) {
      fv[m] = 0.0;
#pragma unroll
      for (int j = 0; j < N; ++j) A[m][j] = 0.0;
    }
  };
  constexpr bool sym = SYM;
  // ---- identity-task-map leaves (row layout): they need q and qd only, not the kinematics ------------------------
  auto identity_leaves = [&]() __attribute__((always_inline)) {
      for (int li = 0; li < n_id; ++li) {
        // (scalar-cache walk: the identity leaves follow the FK leaves in exec_leaves[], in execution order -- no index fetch)
        const DevLeaf& lfr = STAGE ? leaves[uni<STAGE>(id_list[li])] : prog->exec_leaves[hdr.n_fk + li];
        const LeafHead lh = *reinterpret_cast<const LeafHead*>(&lfr);  // one 64-byte load
        struct {
          int kind, goal_offset;
          const float* va;
          const float* vb;
        } lf = {uni<STAGE>(lh.kind), uni<STAGE>(lh.goal_offset), lfr.va, lfr.vb};
        const float* P = lh.P;
        if (lf.kind == RMP2_LEAF_JOINT_DAMPING || lf.kind == RMP2_LEAF_CSPACE_BIASING ||
            lf.kind == RMP2_LEAF_CONFIG_SPACE_BIASING) {
          // diagonal metrics m * I:  A_ii += m, f_i += m * xdd_i
          float mdiag, nrm = 0.f;
          if (lf.kind == RMP2_LEAF_JOINT_DAMPING) {  // rmp2.py:127-137
            float s2 = 0.f;
  #pragma unroll
            for (int j = 0; j < N; ++j) s2 += my_qd[j] * my_qd[j];
            nrm = s2 > 0.f ? s2 * rsq1(s2) : 0.f;
            mdiag = P[1] * nrm + P[2];
          } else if (lf.kind == RMP2_LEAF_CSPACE_BIASING) {  // rmp2.py:212-226
            float s2 = 0.f;
  #pragma unroll
            for (int j = 0; j < N; ++j) {
              const float e = my_q[j] - lf.va[j];
              s2 += e * e;
            }
            nrm = sqrtf(s2);
            mdiag = P[0] + P[4];
          } else {  // rmp.py:330-347
            mdiag = P[2];
          }
  #pragma unroll
          for (int m = 0; m < ROWS; ++m) {
            const int i = sub + kQuad * m;
            const int ii = i < N ? i : 0;
            const float qi_ = my_q[ii], qdi = my_qd[ii];
            float acc;
            if (lf.kind == RMP2_LEAF_JOINT_DAMPING) {
              acc = -(P[0] * nrm) * qdi;
            } else if (lf.kind == RMP2_LEAF_CSPACE_BIASING) {
              const float e = qi_ - (STAGE ? lf.va[ii] : pick4(lf.va, m, sub));
              const float pos = (nrm < P[3]) ? (-e * P[1]) : (-P[3] * (e / nrm) * P[1]);
              acc = pos + (-P[2] * qdi);
            } else {
              acc = P[0] * ((STAGE ? lf.va[ii] : pick4(lf.va, m, sub)) - qi_) - P[1] * qdi;
            }
            fv[m] += (double)(mdiag * acc);
          }
          // A_jj += m: column j's diagonal lives in local row j >> 2 of lane sub == (j & 3)
          const double dm = (double)mdiag;
  #pragma unroll
          for (int j = 0; j < N; ++j) A[j >> 2][j] += (sub == (j & 3)) ? dm : 0.0;
        } else if (lf.kind == RMP2_LEAF_JOINT_VELOCITY_CAP) {
          // rmp2.py:100-112: metric = w / (1 - diag(ratio^2)) evaluated on the FULL matrix (quirk Q4): every off-diagonal
          // entry is w, the diagonal is w / (1 - ratio_i^2).  A constant plus a diagonal: no n x n loop of products --
          //   A_ij += w ,  A_ii += d_i - w ,  f_i += w sum_j xdd_j + (d_i - w) xdd_i
          const float cutoff = P[0] - P[1];
          const float w = P[3] / 1.0f;
          float xo[ROWS], dg[ROWS], sx = 0.f;
  #pragma unroll
          for (int m = 0; m < ROWS; ++m) {
            const int i = sub + kQuad * m;
            const float qdj = my_qd[i < N ? i : 0];
            const float dv = fabsf(qdj) - cutoff;
            const float sgn = (qdj > 0.f) ? 1.f : (qdj < 0.f ? -1.f : 0.f);
            const float acc = -fabsf(P[2] * dv) * sgn;
            xo[m] = (i < n_dof && !(fabsf(qdj) < cutoff)) ? acc : 0.f;
            const float ratio = fminf(dv, P[1] - 1e-6f) / P[1];
            dg[m] = P[3] / (1.0f - ratio * ratio);
            sx += xo[m];
          }
          sx = quad_sum(sx);
          const double wd = (double)w;
  #pragma unroll
          for (int m = 0; m < ROWS; ++m) {
            const int i = sub + kQuad * m;
            const bool row_ok = i < n_dof;
            const double wr = row_ok ? wd : 0.0;
            const double dd = row_ok ? (double)dg[m] - wd : 0.0;  // exact: A_ii = w + (d_i - w) = d_i
  #pragma unroll
            for (int j = 0; j < N; ++j)
              if (j < n_dof && (j >= kQuad * m || !sym)) A[m][j] += wr;  // (sym: the blocks below the diagonal are not kept)
  #pragma unroll
            for (int c = 0; c < kQuad; ++c)
              if (kQuad * m + c < N) A[m][kQuad * m + c] += (sub == c) ? dd : 0.0;
            fv[m] += (double)(row_ok ? fmaf(w, sx, (dg[m] - w) * xo[m]) : 0.f);
          }
        } else {
          // dense metrics  A_ij = cw_j * w * (beta zeta_i zeta_j + (1 - beta) delta_ij)
          float zeta[N], xdd[N], cw[N], beta, wsc;
          // each lane forms the terms of ITS dofs (i = sub + 4 m) once; the quad then broadcasts them (three DPP moves per
          // dof and array) instead of every lane recomputing all n of them -- the divisions below are IEEE sequences
          auto expand = [&](const float (&own)[ROWS], float (&full)[N]) __attribute__((always_inline)) {
  #pragma unroll
            for (int j = 0; j < N; ++j) {
              const float v = own[j >> 2];
              full[j] = (j & 3) == 0 ? bcast<0>(v) : (j & 3) == 1 ? bcast<1>(v) : (j & 3) == 2 ? bcast<2>(v) : bcast<3>(v);
            }
          };
          if (lf.kind == RMP2_LEAF_JOINT_LIMIT_AVOIDANCE) {
            // rmp.py:357-382; A = w * H broadcasts over the LAST axis: column scaling (quirk Q2)
            const float rr_ = 0.15f;
            const float c2 = (float)(-3.0 / (0.15 * 0.15)), c3 = (float)(2.0 / (0.15 * 0.15 * 0.15));
            const float iqd_max = (float)(60.0 / (20.0 * (2.0 * 3.14159265358979323846)));
            float cw_o[ROWS], zeta_o[ROWS], xdd_o[ROWS], s2 = 0.f;
  #pragma unroll
            for (int m = 0; m < ROWS; ++m) {
              const int i = sub + kQuad * m;
              const int ii = i < N ? i : 0;
              const float qj = my_q[ii], qdj = my_qd[ii];
              const float lo_ = STAGE ? lf.va[ii] : pick4(lf.va, m, sub), hi_ = STAGE ? lf.vb[ii] : pick4(lf.vb, m, sub);
              const float irange = rcp1(hi_ - lo_);
              const float du = (hi_ - qj) * irange;
              const float dl = (qj - lo_) * irange;
              const float d = fminf(du, dl);
              const float spline = c3 * (d * d * d) + c2 * (d * d) + 0.f * d + 1.0f;
              cw_o[m] = (i < n_dof) ? (d > rr_ ? 0.f : spline) : 0.f;
              zeta_o[m] = (i < N) ? qdj * iqd_max : 0.f;
              s2 += zeta_o[m] * zeta_o[m];
              xdd_o[m] = -P[0] * qj - P[1] * qdj;
            }
            s2 = quad_sum(s2);
            const float nrm = s2 > 0.f ? s2 * rsq1(s2) : 0.f;
            // soft norm h = |v| + (1/c) log(1 + exp(-2 c |v|)), c = 5   (helper/rmp_helper.py:62-65)
            const float hh = nrm + 0.2f * (0.693147182464599609375f * __builtin_amdgcn_logf(1.0f + exp1(-10.0f * nrm)));
            const float ihh = rcp1(hh);
  #pragma unroll
            for (int m = 0; m < ROWS; ++m) zeta_o[m] *= ihh;
            expand(cw_o, cw);
            expand(zeta_o, zeta);
            expand(xdd_o, xdd);
            beta = 0.9f;
            wsc = 1.0f;
          } else {
            // TargetPolicy on the identity map, rmp.py:241-260 (goal is an n-vector)
            const float alpha = P[0], beta_d = P[1], c = P[2];
            float v[N], s2 = 0.f;
  #pragma unroll
            for (int j = 0; j < N; ++j) {
              v[j] = (j < n_dof) ? my_goal[lf.goal_offset + j] - my_q[j] : 0.f;
              s2 += v[j] * v[j];
            }
            const float vn = sqrtf(s2);
            const float hq = vn + c * logf(1.0f + expf(-2.0f * c * vn));
            const float inv_h = 1.0f / hq;
            float f2 = 0.f;
  #pragma unroll
            for (int j = 0; j < N; ++j) {
              xdd[j] = alpha * (inv_h * v[j]) - beta_d * my_qd[j];
              f2 += xdd[j] * xdd[j];
              cw[j] = 1.0f;
            }
            const float fn = sqrtf(f2);
            const float hs = fn + 1.0f / c * logf(1.0f + expf(-2.0f * c * fn));
  #pragma unroll
            for (int j = 0; j < N; ++j) zeta[j] = xdd[j] / hs;
            beta = 1.0f - expf(-0.5f * (vn * vn) / 1.0f);
            wsc = expf(-vn / 3.0f);
          }
          const float omb = 1.0f - beta;
          const bool is_jla = lf.kind == RMP2_LEAF_JOINT_LIMIT_AVOIDANCE;
          int sd = sub;  // (opaque copy, as for the walk's base row)
          if (MINW >= 3) asm volatile("" : "+v"(sd));
  #pragma unroll
          for (int m = 0; m < ROWS; ++m) {
            const int i = sd + kQuad * m;
            const bool row_ok = i < n_dof;
            // zeta_i of MY row: select from the statically indexed vector
            float zi = 0.f;
  #pragma unroll
            for (int j = 0; j < N; ++j) zi = (j == i) ? zeta[j] : zi;
            float fi = 0.f;
  #pragma unroll
            for (int j = 0; j < N; ++j) {
              if (j >= n_dof) continue;
              // JointLimitAvoidance scales COLUMN j by the limit weight of joint j, which is exactly 0 unless
              // that joint is inside its limit band: skip the column when that holds for the whole wave
              if (is_jla && !__any(cw[j] != 0.f)) continue;
              const float Hij = beta * (zi * zeta[j]) + omb * (j == i ? 1.f : 0.f);
              float a = is_jla ? cw[j] * Hij : wsc * Hij;
              a = row_ok ? a : 0.f;
              if (!SYM || j >= kQuad * m) A[m][j] += (double)a;  // (sym: the blocks below the diagonal are not kept)
              fi += a * xdd[j];
            }
            fv[m] += (double)fi;
          }
        }
      }
  };
  // Phase order per wave.  The four waves of a SIMD start together and would walk their kinematic trees together -- the
  // one phase that is a dependent chain (latency bound, the SIMD issues next to nothing) -- and then crowd the issue-bound
  // leaf phases together.  The identity-map leaves depend on nothing the walk produces, so the waves in the ODD wave slots
  // of a SIMD run them BEFORE the walk (symmetric form only: the general form's mirror step would overwrite what they add
  // below the diagonal blocks): their issue-dense work fills the slots the even waves' walks leave empty, and their walks
  // run beside the even waves' frame loops.
  const bool ident_first = kIdentFirst && ((__builtin_amdgcn_s_getreg(0x1804) & 1) != 0);  // HW_ID[3:0] = wave slot in its SIMD
  if (ident_first) {
    zero_system();
    identity_leaves();
  }

  // ---- phase 2: serial tree walk (component layout), ONCE per step ------------------------------
  // Deliberately a tiny loop with nothing else in it: the walk is one dependent chain, so every
  // instruction in its body is latency.  The frame's control word travels inside its LDS record;
  // frames that carry leaves drop (p, v, a) into LDS for the leaf phase below.
  if (n_ops > 0) {
    // zero-initialised on purpose: left uninitialised, the compiler closes the undefined value at loop entry with the
    // previous control step's state, i.e. carries 8 (1 + SLOTS) registers across the WHOLE step for nothing (16 fewer
    // registers in the 256-register build; in the 128-register build the walk's state otherwise lives in scratch)
    QuadState cur = {};
    QuadState slot[SLOTS > 0 ? SLOTS : 1] = {};
    // loop-local copies of the long-lived LDS addresses: their own short live ranges -- the 128-register build otherwise
    // reloads the spilled originals inside the loop (4 scratch loads per frame on the kernel's one serial chain)
    // (the OFFSETS are laundered, not the pointers: an opaque pointer loses its LDS address space and every access through
    // it becomes a 64-bit flat_load / flat_store instead of a ds_read / ds_write)
    int wloc_off = QuadLds<N>::kLoc + gi_loc(g, n_ops), wqd_off = QuadLds<N>::kQd + gi * N;
    if (MINW >= 3) asm volatile("" : "+v"(wloc_off), "+v"(wqd_off));  // (the builds with registers to spare lose 3 % to it)
    float* wloc = lds + wloc_off;
    const float* wqd = lds + wqd_off;
    const float4* rec4n = reinterpret_cast<const float4*>(wloc);
    float4 n0 = rec4n[0], n1 = rec4n[1], n2 = rec4n[2];
    // {axis, ctl} of the frame: wave-uniform, one 16-byte fetch per frame (scalar cache, or the staged copy), one ahead
    float4 nac = *reinterpret_cast<const float4*>(ops[0].axis);
    // (link segments: the capsule of the frame's distance leaf, fetched with the control word a frame ahead; ctl bits 12..17 =
    // its row of link_capsules + 1, 0 = the frame carries no distance leaf)
    float4 nla = make_float4(0.f, 0.f, 0.f, 0.f), nlb = nla;
    if (LINKSEG) {
      const int row0 = ((__float_as_int(nac.w) >> 12) & 63) - 1;
      const float4* lc0 = reinterpret_cast<const float4*>(obs.link_caps) + 2 * max(row0, 0);
      nla = lc0[0], nlb = lc0[1];
    }
    int t_leaf = 0;  // (wave-uniform) ordinal of the frame among the leaf-bearing frames: its segment slot
    for (int k = 0; k < n_ops; ++k) {
      const float4 r0_ = n0, r1_ = n1, r2_ = n2, ac = nac;
      const float4 la4 = nla, lb4 = nlb;
      const int ctl = uni<STAGE>(__float_as_int(ac.w));
      const int c_restore = (ctl & 3) - 2, c_save = ((ctl >> 2) & 3) - 1, c_jtype = (ctl >> 4) & 3;
      const int qi = ((ctl >> 6) & 31) - 1;
      // branch-free joint handling (a taken scalar branch costs a two-wave SIMD the work of ~5 instructions): the joint
      // velocity is read unconditionally and the joint type enters as two wave-uniform 0 / 1 factors
      const float qdv = wqd[max(qi, 0)];
      const float f_rev = (c_jtype == RMP2_JOINT_REVOLUTE && qi >= 0) ? 1.0f : 0.0f;
      const float f_pri = (c_jtype == RMP2_JOINT_PRISMATIC && qi >= 0) ? 1.0f : 0.0f;
      {  // next frame's record: issued now, consumed one iteration later
        const int kn = min(k + 1, n_ops - 1);
        rec4n = reinterpret_cast<const float4*>(wloc + kSlot * kn);
        n0 = rec4n[0];
        n1 = rec4n[1];
        n2 = rec4n[2];
        nac = *reinterpret_cast<const float4*>(ops[kn].axis);
        if (LINKSEG) {
          const int rown = ((__float_as_int(nac.w) >> 12) & 63) - 1;
          const float4* lcn = reinterpret_cast<const float4*>(obs.link_caps) + 2 * max(rown, 0);
          nla = lcn[0], nlb = lcn[1];
        }
      }
      const float Rl[9] = {r0_.x, r0_.y, r0_.z, r0_.w, r1_.x, r1_.y, r1_.z, r1_.w, r2_.x};
      const float tl[3] = {r2_.y, r2_.z, r2_.w};
      if (c_restore != -1) {  // (one test for the common case: the frame continues from its predecessor)
        if (c_restore == -2) {
          // base: identity rotation row, zero vectors.  (e_sub @ Rl == row sub of Rl exactly.)
          int sb = sub;  // (opaque copy: the three 0 / 1 values are not worth a register across the whole step)
          if (MINW >= 3) asm volatile("" : "+v"(sb));
#pragma unroll
          for (int m = 0; m < 3; ++m) cur.R[m] = (m == sb) ? 1.0f : 0.0f;
          cur.p = cur.w = cur.al = cur.v = cur.a = 0.f;
        } else if (SLOTS > 0) {
#pragma unroll
          for (int s2 = 0; s2 < SLOTS; ++s2)
            if (c_restore == s2) cur = slot[s2];
        }
      }
      // world: my row of  R_parent @ R_local, my component of  R_parent @ t_local + p_parent
      float Rn[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) Rn[c] = cur.R[0] * Rl[c] + cur.R[1] * Rl[3 + c] + cur.R[2] * Rl[6 + c];
      const float pn = cur.R[0] * tl[0] + cur.R[1] * tl[1] + cur.R[2] * tl[2] + cur.p;
      const float rr = pn - cur.p;
      // world joint axis: my component of R_world @ axis (my row of the new world rotation)
      const float z = Rn[0] * ac.x + Rn[1] * ac.y + Rn[2] * ac.z;
      // velocity / bias-acceleration recursion, one component per lane; (a x b)_i = a_{i+1} b_{i+2} - a_{i+2} b_{i+1}
      const float w1 = dpp<kRot1>(cur.w), w2 = dpp<kRot2>(cur.w);
      const float r1 = dpp<kRot1>(rr), r2 = dpp<kRot2>(rr);
      const float a1 = dpp<kRot1>(cur.al), a2 = dpp<kRot2>(cur.al);
      const float t1 = w1 * r2 - w2 * r1;  // w_p x r
      const float t2 = a1 * r2 - a2 * r1;  // alpha_p x r
      const float t1a = dpp<kRot1>(t1), t1b = dpp<kRot2>(t1);
      const float t3 = w1 * t1b - w2 * t1a;  // w_p x (w_p x r)
      const float zq = z * qdv;
      const float zq1 = dpp<kRot1>(zq), zq2 = dpp<kRot2>(zq);
      const float t4 = w1 * zq2 - w2 * zq1;  // w_p x (z qd)
      // (x + 1 * y and x + 0 * y are exact: same values as the branched form)
      const float wn = fmaf(f_rev, zq, cur.w), aln = fmaf(f_rev, t4, cur.al);
      const float vn = fmaf(f_pri, zq, cur.v + t1), an = fmaf(2.0f * f_pri, t4, cur.a + t2 + t3);
#pragma unroll
      for (int c = 0; c < 3; ++c) cur.R[c] = Rn[c];
      cur.p = pn;
      cur.w = wn;
      cur.al = aln;
      cur.v = vn;
      cur.a = an;
      if (sub < 3) {  // the frame's world record [p, v, a, z] replaces its local transform (consumed by every lane of
        float* fr = wloc + kSlot * k;  // the quad one iteration ago)
        fr[sub] = pn;
        fr[3 + sub] = vn;
        fr[6 + sub] = an;
        fr[9 + sub] = z;
        if (LINKSEG && ((ctl >> 12) & 63) != 0) {  // (wave-uniform) my component of the link's world segment A, B
          float* sg = stage_base + (g * hdr.n_leaf_ops + t_leaf) * kLinkSeg;
          sg[sub] = pn + Rn[0] * la4.x + Rn[1] * la4.y + Rn[2] * la4.z;
          sg[4 + sub] = pn + Rn[0] * lb4.x + Rn[1] * lb4.y + Rn[2] * lb4.z;
        }
        if (PT) {  // row `sub` of the world rotation and component `sub` of w and alpha, for the attached-point leaves
          float* pr = pt_base + (g * n_ops + k) * pt_slot;
          pr[3 * sub] = Rn[0];
          pr[3 * sub + 1] = Rn[1];
          pr[3 * sub + 2] = Rn[2];
          if (pt_slot == kPtSlot) {
            pr[9 + sub] = wn;
            pr[12 + sub] = aln;
          }
        }
      }
      if (SLOTS > 0 && c_save >= 0) {
#pragma unroll
        for (int s2 = 0; s2 < SLOTS; ++s2)
          if (c_save == s2) slot[s2] = cur;
      }
      if (LINKSEG) t_leaf += (ctl >> 11) & 1;  // (the frame carries leaves: the next leaf-bearing frame takes the next slot)
    }
  }
  RMP2_STAMP();  // 2: walk done

  // ---- explicit pairs, streamed (OBS = kObsExplicitStream): what the frame loop needs of the frame records leaves their LDS ---------
  // (see kObsExplicitStream above).  The (z, o) records of the joints that own my rows go to registers; [p v a] of every leaf-bearing
  // frame goes to a compact image at the START of the records' region -- nine floats per robot and frame, kStreamMaxFrames frames per
  // robot: 5 184 B --, and the rest of the region (from kStreamPva floats on) becomes the LDS-DMA chunk buffer of the pair loops.
  float rjz_s[STREAM ? ROWS : 1][3], rjo_s[STREAM ? ROWS : 1][3];
  if constexpr (STREAM) {
#pragma unroll
    for (int m = 0; m < ROWS; ++m) {
      const int i = sub + kQuad * m;
      const int ii = i < N ? i : 0;
      const uint32_t dw = ii < 6 ? hdr.dof_ops[0] : (ii < 12 ? hdr.dof_ops[1] : hdr.dof_ops[2]);
      const int fo = (int)((dw >> (5 * (ii - 6 * (ii / 6)))) & 31u);
      const float4* rj = reinterpret_cast<const float4*>(loc + kSlot * fo);
      const float4 j0 = rj[0], j2 = rj[2];
      rjz_s[m][0] = j2.y, rjz_s[m][1] = j2.z, rjz_s[m][2] = j2.w;
      rjo_s[m][0] = j0.x, rjo_s[m][1] = j0.y, rjo_s[m][2] = j0.z;
    }
    float pva_[kStreamMaxFrames][3];  // my floats {sub, 4 + sub, 8 + sub} of the twelve [p v a -] of leaf frame t
#pragma unroll
    for (int t = 0; t < kStreamMaxFrames; ++t) {
      pva_[t][0] = pva_[t][1] = pva_[t][2] = 0.f;
      if (t < hdr.n_leaf_ops) {  // (wave-uniform)
        const float* fr = loc + kSlot * prog->leaf_frames[t].op;
        pva_[t][0] = fr[sub], pva_[t][1] = fr[4 + sub], pva_[t][2] = fr[8 + sub];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every lane of the wave has its copies: the records may be overwritten
    float* const pva = lds + QuadLds<N>::kLoc + g * (kStreamMaxFrames * 9);
#pragma unroll
    for (int t = 0; t < kStreamMaxFrames; ++t) {
      if (t < hdr.n_leaf_ops) {
        pva[9 * t + sub] = pva_[t][0];                    // floats 0 .. 3
        pva[9 * t + 4 + sub] = pva_[t][1];                // floats 4 .. 7
        if (sub == 0) pva[9 * t + 8] = pva_[t][2];        // float 8 (9 .. 11 of the record: the joint axis, not needed here)
      }
    }
  }

  // One accumulation + resolve pass over the wave's robots.  PASS 0 is the fast path (elimination without row exchanges);
  // PASS 1 re-accumulates and runs the careful solver for the robots PASS 0 flagged.  Two INSTANTIATIONS of one body, not a
  // two-trip loop: inside a loop every lane predicate of the resolve (sub == c, i < n_dof, ...) is loop invariant, gets
  // hoisted in front of the loop and stays live -- as an SGPR pair parked in a VGPR lane -- through the whole frame loop
  // (round 2: ~100 v_writelane before the loop, ~250 v_readlane inside it).  The second copy is cold code.
  auto run_pass = [&](auto pass_c) __attribute__((always_inline)) {
    constexpr int pass = decltype(pass_c)::value;
    if (!(ident_first && pass == 0)) zero_system();

    // row-joint records: world axis z_i and origin o_i of the joints that own MY rows (slot [p = o | v | a | z] of the
    // joint's frame), read once per pass instead of once per leaf frame
    // (the 128-register build keeps only the records' LDS addresses and re-reads them per frame: 18 registers)
    constexpr bool kRowRecsInRegs = MINW < 4 || STREAM;  // (streamed explicit pairs: taken before the DMA overwrote the records)
    float rjz[ROWS][3], rjo[ROWS][3];
    const float4* rjs[ROWS];
    bool rjrev[ROWS];
#pragma unroll
    for (int m = 0; m < ROWS; ++m) {
      const int i = sub + kQuad * m;
      const int ii = i < N ? i : 0;
      const uint32_t dw = ii < 6 ? hdr.dof_ops[0] : (ii < 12 ? hdr.dof_ops[1] : hdr.dof_ops[2]);
      const int fo = (int)((dw >> (5 * (ii - 6 * (ii / 6)))) & 31u);
      rjs[m] = reinterpret_cast<const float4*>(loc + kSlot * fo);
      if constexpr (STREAM) {
#pragma unroll
        for (int c = 0; c < 3; ++c) rjz[m][c] = rjz_s[m][c], rjo[m][c] = rjo_s[m][c];
      } else if (kRowRecsInRegs) {
        const float4 j0 = rjs[m][0], j2 = rjs[m][2];
        rjz[m][0] = j2.y, rjz[m][1] = j2.z, rjz[m][2] = j2.w;
        rjo[m][0] = j0.x, rjo[m][1] = j0.y, rjo[m][2] = j0.z;
      }
      rjrev[m] = (rev_mask >> ii) & 1u;
    }
    // ---- leaves on FK task maps, frame by frame (only the frames that carry leaves) --------------
    // (scalar-cache walk: the frame's flattened record is fetched one frame ahead -- a dependent scalar fetch costs
    // ~200 cycles and the old chain leaf_ops[t] -> ops[k] -> fk_leaves[i] -> leaves[id] had four of them per frame)
    int floc_off = QuadLds<N>::kLoc + gi_loc(g, n_ops);  // (loop-local copy, as in the walk: the offset, not the pointer)
    if (MINW >= 3) asm volatile("" : "+v"(floc_off));
    const float* floc = lds + floc_off;
    int4 lfr_next = STAGE ? make_int4(0, 0, 0, 0) : *reinterpret_cast<const int4*>(&prog->leaf_frames[0]);
    for (int t = 0; t < hdr.n_leaf_ops; ++t) {
      if (MINW >= 3) {  // thirds of the frame loop at 3 / 2 / 1
        if (3 * t >= hdr.n_leaf_ops && 3 * (t - 1) < hdr.n_leaf_ops) __builtin_amdgcn_s_setprio(2);
        if (3 * t >= 2 * hdr.n_leaf_ops && 3 * (t - 1) < 2 * hdr.n_leaf_ops) __builtin_amdgcn_s_setprio(1);
      } else {          // two waves per SIMD (and the lone wave of the latency build): halves at 2 / 1 measured better
        if (t == 0) __builtin_amdgcn_s_setprio(2);
        if (2 * t == hdr.n_leaf_ops) __builtin_amdgcn_s_setprio(1);
      }
      int k;
      OpCtl op;
      if (STAGE) {
        k = uni<STAGE>(leaf_ops[t]);
        op = *reinterpret_cast<const OpCtl*>(&ops[k]);
        op.anc_mask = (uint32_t)uni<STAGE>((int)op.anc_mask);
        op.leaf_begin = uni<STAGE>(op.leaf_begin);
        op.leaf_count = uni<STAGE>(op.leaf_count);
      } else {
        const int4 lfr = lfr_next;
        lfr_next = *reinterpret_cast<const int4*>(&prog->leaf_frames[min(t + 1, hdr.n_leaf_ops - 1)]);
        k = lfr.x;
        op.anc_mask = (uint32_t)lfr.y;
        op.leaf_begin = lfr.z;
        op.leaf_count = lfr.w;
      }
      RMP2_SEG_BEGIN();
      // full 3-vectors of the frame in every lane (written by the walk; broadcast reads)
      float P3[3], V3[3], A3[3];
      if constexpr (STREAM) {  // the compact [p v a] image (nine floats per robot and leaf frame)
        const float* pv = lds + QuadLds<N>::kLoc + g * (kStreamMaxFrames * 9) + 9 * t;
#pragma unroll
        for (int c = 0; c < 3; ++c) P3[c] = pv[c], V3[c] = pv[3 + c], A3[c] = pv[6 + c];
      } else {
        const float4* fr4 = reinterpret_cast<const float4*>(floc + kSlot * k);
        const float4 f0 = fr4[0], f1 = fr4[1], f2 = fr4[2];
        P3[0] = f0.x, P3[1] = f0.y, P3[2] = f0.z, V3[0] = f0.w, V3[1] = f1.x, V3[2] = f1.y, A3[0] = f1.z, A3[1] = f1.w, A3[2] = f2.x;
      }
      // Jacobian columns of the frame: formed AFTER the first leaf's (S, h) -- the pair loop is where the time goes and
      // it runs with 36 fewer live registers this way (the kernel must fit 128 for four waves per SIMD)
      float mycol[ROWS][3];
      if (MINW >= 4) {  // (as for the walk's state: an undefined value at loop entry is closed with the previous frame's columns,
#pragma unroll       //  nine registers live through the pair loop; the 256-register build has them to spare)
        for (int m = 0; m < ROWS; ++m) mycol[m][0] = mycol[m][1] = mycol[m][2] = 0.f;
      }
      for (int li = 0; li < op.leaf_count; ++li) {
        float S[6], h[3];
        // attached-point leaf: the sums over its pairs that the pull-back below needs (see there)
        bool pt_leaf = false;
        float ptW = 0.f, ptRho[3] = {0.f, 0.f, 0.f}, ptQ[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ptTau[3] = {0.f, 0.f, 0.f};
        {
          const DevLeaf& lf = STAGE ? leaves[uni<STAGE>(fk_list[op.leaf_begin + li])] : prog->exec_leaves[op.leaf_begin + li];
          LeafHead lh = *reinterpret_cast<const LeafHead*>(&lf);  // one 64-byte load
          lh.kind = uni<STAGE>(lh.kind);
          lh.taskmap = uni<STAGE>(lh.taskmap);
          lh.goal_offset = uni<STAGE>(lh.goal_offset);
          const int lf_kind = lh.kind;
          RMP2_SEG(0);  // frame record + leaf head on chip
          if (PT && lh.taskmap == RMP2_TASKMAP_FK_POINT) {
            // chain [FK(frame), TaskmapRelative4x4(rel), 4x4 -> position] + CollisionAvoidance (metric w(d) I), pair b:
            //   r = R rel, x = p + r, xd = v + w x r, c = a + al x r + w x (w x r), J_b = J_p + [z_j x r]_j (revolute dofs).
            // Every pair has its OWN Jacobian, but it differs from the frame origin's by a term linear in r, so the sums over
            // the pairs collapse into 16 numbers and ONE pull-back per frame:
            //   M += W J_p^T J_p - J_p^T [rho]x Z - (..)^T + Z^T Q Z ,   f += J_p^T h + Z^T tau
            //   W = sum w_b, rho = sum w_b r_b, Q = sum w_b (|r_b|^2 I - r_b r_b^T), h = sum w_b e_b, tau = sum w_b r_b x e_b,
            //   e_b = xdd_b - c_b, Z = the revolute dofs' world axes (zero columns for prismatic dofs and non-ancestors).
            // (The hex and lane mappings pull every pair back through its own columns; same sums to fp32 rounding.)
            pt_leaf = true;
            const float4* pr4 = reinterpret_cast<const float4*>(pt_base + (g * n_ops + k) * kPtSlot);
            const float4 q0 = pr4[0], q1 = pr4[1], q2 = pr4[2], q3 = pr4[3];
            const float Rm[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x};
            const float W3[3] = {q2.y, q2.z, q2.w}, AL3[3] = {q3.x, q3.y, q3.z};
            // The leaf's pairs: (relative_position, normal_vec, distance) per pair from the caller's arrays (the reference's
            // Datamanager fields, data_management.py:22-53) -- or, with a primitive table and link capsules (obs.link_caps), formed
            // HERE per control step: the closest points of the leaf's link capsule and every primitive, as PyBullet reports them
            // to the reference's loop each step (simulation.py:462-484; 05_obstacle_avoidance.py:51-72 re-feeds them); this is
            // what lets such a set roll out inside one launch.
            const bool from_table = obs.link_caps != nullptr;  // (wave-uniform)
            int count;
            size_t pbase = 0;
            float LA[3] = {0.f, 0.f, 0.f}, LD[3] = {0.f, 0.f, 0.f}, lrad = 0.f, laa = 0.f, inv_laa = 0.f;
            if (from_table) {
              count = obs.n_spheres;
              const float* lc = obs.link_caps + 8 * uni<STAGE>(lf.dist_ordinal);
              lrad = lc[3];
  #pragma unroll
              for (int c = 0; c < 3; ++c) {
                LA[c] = P3[c] + Rm[3 * c] * lc[0] + Rm[3 * c + 1] * lc[1] + Rm[3 * c + 2] * lc[2];
                LD[c] = Rm[3 * c] * (lc[4] - lc[0]) + Rm[3 * c + 1] * (lc[5] - lc[1]) + Rm[3 * c + 2] * (lc[6] - lc[2]);
              }
              laa = dot3(LD, LD);
              inv_laa = laa > 0.f ? 1.0f / laa : 0.f;
            } else {
              const int lidx = uni<STAGE>(lf.index);
              const int pb = obs.pair_begin[lidx];
              count = obs.pair_begin[lidx + 1] - pb;
              pbase = (size_t)(live ? robot : 0) * obs.n_pairs + pb;
            }
            h[0] = h[1] = h[2] = 0.f;
            for (int t = 0; kQuad * t < count; ++t) {  // (the pair layout is shared by the fleet: wave-uniform trip count)
              const int b_raw = kQuad * t + sub;
              const bool on = b_raw < count;
              float r[3], nv[3], dd, t1[3], t2[3], xdp[3], cp[3];
              if (from_table) {
                const int bi = on ? b_raw : 0;
                const float4* rec = reinterpret_cast<const float4*>(step_table) + (obs.capsule ? 2 * bi : bi);
                const float4 ca = rec[0];
                const float4 cb = obs.capsule ? rec[1] : ca;
                link_pair_fields(LA, LD, laa, inv_laa, lrad, ca, cb, P3, r, nv, dd);
              } else {
                const size_t b = pbase + (on ? b_raw : 0);
                const float rel[3] = {obs.p_link[3 * b], obs.p_link[3 * b + 1], obs.p_link[3 * b + 2]};
                nv[0] = obs.p_obs[3 * b], nv[1] = obs.p_obs[3 * b + 1], nv[2] = obs.p_obs[3 * b + 2];
                dd = obs.dist[b];
  #pragma unroll
                for (int c = 0; c < 3; ++c) r[c] = Rm[3 * c] * rel[0] + Rm[3 * c + 1] * rel[1] + Rm[3 * c + 2] * rel[2];
              }
              cross3(W3, r, t1);
  #pragma unroll
              for (int c = 0; c < 3; ++c) xdp[c] = V3[c] + t1[c];
              cross3(W3, t1, t2);
              cross3(AL3, r, t1);
  #pragma unroll
              for (int c = 0; c < 3; ++c) cp[c] = A3[c] + t1[c] + t2[c];
              float xdd[3], wgt;
              leaf_collision_avoidance(lh.P, dd, nv, xdp, xdd, wgt);
              if (!on) wgt = 0.f;
              const float e[3] = {xdd[0] - cp[0], xdd[1] - cp[1], xdd[2] - cp[2]};
              float re[3];
              cross3(r, e, re);
              const float rr = dot3(r, r);
              ptW += wgt;
  #pragma unroll
              for (int c = 0; c < 3; ++c) {
                ptRho[c] = fmaf(wgt, r[c], ptRho[c]);
                h[c] = fmaf(wgt, e[c], h[c]);
                ptTau[c] = fmaf(wgt, re[c], ptTau[c]);
              }
              ptQ[0] = fmaf(wgt, rr - r[0] * r[0], ptQ[0]);
              ptQ[1] = fmaf(wgt, -r[0] * r[1], ptQ[1]);
              ptQ[2] = fmaf(wgt, -r[0] * r[2], ptQ[2]);
              ptQ[3] = fmaf(wgt, rr - r[1] * r[1], ptQ[3]);
              ptQ[4] = fmaf(wgt, -r[1] * r[2], ptQ[4]);
              ptQ[5] = fmaf(wgt, rr - r[2] * r[2], ptQ[5]);
            }
            ptW = quad_sum(ptW);
  #pragma unroll
            for (int c = 0; c < 3; ++c) {
              ptRho[c] = quad_sum(ptRho[c]);
              h[c] = quad_sum(h[c]);
              ptTau[c] = quad_sum(ptTau[c]);
            }
  #pragma unroll
            for (int c = 0; c < 6; ++c) ptQ[c] = quad_sum(ptQ[c]);
            S[0] = S[3] = S[5] = ptW;   // (the metric of the origin term: W I)
            S[1] = S[2] = S[4] = 0.f;
          } else if (lh.taskmap == RMP2_TASKMAP_FK_POSITION) {
            float gl[3], xdd[3];
  #pragma unroll
            for (int c = 0; c < 3; ++c) gl[c] = my_goal[lh.goal_offset + c];
            if (lf_kind == RMP2_LEAF_TARGET_ATTRACTOR)
              target_attractor_fast(lh.P, P3, V3, gl, xdd, S);
            else
              leaf_target_policy3(lh.P, P3, V3, gl, xdd, S);
            const float e[3] = {xdd[0] - A3[0], xdd[1] - A3[1], xdd[2] - A3[2]};
            h[0] = S[0] * e[0] + S[1] * e[1] + S[2] * e[2];
            h[1] = S[1] * e[0] + S[3] * e[1] + S[4] * e[2];
            h[2] = S[2] * e[0] + S[4] * e[1] + S[5] * e[2];
          } else {
            // distance leaf: this lane takes pairs b = sub, sub+4, ...; S and h are butterfly-summed.
            // The obstacle mode is resolved OUTSIDE the loop (one straight-line loop body per mode).
  #pragma unroll
            for (int c = 0; c < 6; ++c) S[c] = 0.f;
            h[0] = h[1] = h[2] = 0.f;
            const float IP[6] = {lf.vb[0], lf.vb[1], lf.vb[2], lf.vb[3], lf.vb[4], lf.vb[5]};
            const float* sph_lds = sph_lds_base;
            if (LINKSEG) {  // lean link build: the world segment was formed in the walk (slot t of this robot)
              const float4* sg4 = reinterpret_cast<const float4*>(stage_base + (g * hdr.n_leaf_ops + t) * kLinkSeg);
              const float4 sa = sg4[0], sb = sg4[1];
              const float LA[3] = {sa.x, sa.y, sa.z}, LB[3] = {sb.x, sb.y, sb.z};
              const float lrad = obs.link_caps[8 * lf.dist_ordinal + 3];
              pair_loop_link<true, CAP>(sph_lds, n_sph_lds, obs.n_spheres, sub, LA, LB, lrad, hdr.cull_c0, V3, A3, lh.P, IP, S, h,
                                        step_table);
            } else if (PT && obs.link_caps) {  // (wave-uniform) link geometry: SHARED_SPHERES, table in LDS (checked on the host)
              const float4* pr4 = reinterpret_cast<const float4*>(pt_base + (g * n_ops + k) * pt_slot);
              const float4 q0 = pr4[0], q1 = pr4[1], q2 = pr4[2];
              const float Rm[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x};
              const float* lc = obs.link_caps + 8 * uni<STAGE>(lf.dist_ordinal);
              const float la[3] = {lc[0], lc[1], lc[2]}, lb[3] = {lc[4], lc[5], lc[6]};
              float LA[3], LB[3];
  #pragma unroll
              for (int c = 0; c < 3; ++c) {
                LA[c] = P3[c] + Rm[3 * c] * la[0] + Rm[3 * c + 1] * la[1] + Rm[3 * c + 2] * la[2];
                LB[c] = P3[c] + Rm[3 * c] * lb[0] + Rm[3 * c + 1] * lb[1] + Rm[3 * c + 2] * lb[2];
              }
              if (obs_mode == RMP2_OBS_RAGGED_SPHERES && !use_member) {  // (wave-uniform) a list that is not a mask: the list walk
                int rr_ = live ? robot : 0;
                const int b0 = obs.csr_offset[rr_];
                const int count = live ? obs.csr_offset[rr_ + 1] - b0 : 0;
                int max_count = count;
  #pragma unroll
                for (int o = 32; o >= kQuad; o >>= 1) max_count = max(max_count, __shfl_xor(max_count, o));
                pair_loop_link_list(step_table, obs.capsule != 0, obs.csr_index + b0, count, max_count, sub, LA, LB, lc[3], V3, A3, lh.P,
                                    IP, S, h);
              } else
              pair_loop_link<(MINW >= 2), CAP>(sph_lds, n_sph_lds, obs.n_spheres, sub, LA, LB, lc[3], hdr.cull_c0, V3, A3, lh.P, IP, S, h,
                                               step_table, obs_mode == RMP2_OBS_RAGGED_SPHERES, member_lo, member_hi);
            } else if (obs_mode == RMP2_OBS_SHARED_SPHERES) {
              if (spheres_in_lds)
  #ifdef RMP2_STAMPS
                pair_loop_culled<false, kQuad, (MINW >= 2), false, CAP>(sph_lds, n_sph_lds, nullptr, obs.n_spheres, obs.n_spheres, sub, P3,
                                                                       V3, A3, lh.P, IP, S, h, &seg_[5], 0u, 0u, step_table, obs.cylinder != 0);
  #else
                pair_loop_culled<false, kQuad, (MINW >= 2), false, CAP>(sph_lds, n_sph_lds, nullptr, obs.n_spheres, obs.n_spheres, sub, P3,
                                                                       V3, A3, lh.P, IP, S, h, nullptr, 0u, 0u, step_table, obs.cylinder != 0);
  #endif
              else
                pair_loop<kPairsSharedGlobal, CAP>(step_table, nullptr, nullptr, nullptr, obs.n_spheres, obs.n_spheres, sub,
                                              P3, V3, A3, lh.P, IP, S, h, obs.cylinder != 0);
            } else if (obs_mode == RMP2_OBS_EXPLICIT_PAIRS) {
              const int lidx = uni<STAGE>(lf.index);
              const int pb = obs.pair_begin[lidx];
              const int count = obs.pair_begin[lidx + 1] - pb;
              const size_t base = ((size_t)(live ? robot : 0) * obs.n_pairs + pb) * 3;
              // (cull threshold of THIS leaf: x = max(d - margin, 0) > metric_modulation_radius  <=>  d > margin + radius)
              const float thr = fmaxf(lh.P[0] + lh.P[7], 0.f);
              if constexpr (STREAM) {  // a quarter leaf at a time through the chunk buffer behind the compact [p v a] image (32 pairs per leaf: host)
              const int nxt = lf.next_pair_leaf;
              // (the buffer's offset is a compile-time constant: laundered, because the compiler folds the generic-to-LDS cast of a
              //  CONSTANT address inside global_load_lds into an instruction the assembler rejects -- V_CMP with src_shared_base)
              int dma_off = QuadLds<N>::kLoc + kStreamPva;
              asm volatile("" : "+s"(dma_off));
              pair_loop_explicit_qdma(obs.p_link, obs.p_obs, obs.n_pairs, r0, R, pb, nxt >= 0 ? obs.pair_begin[nxt] : -1, pf_pb,
                                      lds + dma_off, lane, g, sub, P3, V3, A3, lh.P, IP, thr * thr * kCullSlack, S, h, hdr.stagger >> 8);
            } else if (kGlds && obs.glds && count == 32) {  // (wave-uniform) streamed half a leaf ahead by LDS-DMA
                const int nxt = lf.next_pair_leaf;
                int pb_next = -1;
                if (nxt >= 0) {
                  const int nb = obs.pair_begin[nxt];
                  pb_next = (obs.pair_begin[nxt + 1] - nb == 32) ? nb : -1;
                }
                pair_loop_explicit_glds(obs.p_link, obs.p_obs, obs.n_pairs, r0, R, pb, pb_next, pf_pb, stage_base, lane, g, sub, P3,
                                        V3, A3, lh.P, IP, thr * thr * kCullSlack, S, h);
              } else
              pair_loop_explicit_culled<(MINW >= 3 ? RMP2_EXPLICIT_WINDOW : 0)>(obs.p_link + base, obs.p_obs + base, count, sub, P3, V3, A3,
                                                                              lh.P, IP, thr * thr * kCullSlack, S, h);
            } else if (use_member) {  // (wave-uniform) ragged list as a membership mask: the dense loop, masked
              pair_loop_culled<false, kQuad, (MINW >= 2), true, CAP>(sph_lds, n_sph_lds, nullptr, obs.n_spheres, obs.n_spheres, sub, P3,
                                                                    V3, A3, lh.P, IP, S, h, nullptr, member_lo, member_hi, step_table, obs.cylinder != 0);
            } else {
              int rr_ = live ? robot : 0;  // (opaque copy: the two 64-bit addresses are formed here, not in the prologue)
              if (MINW >= 3) asm volatile("" : "+v"(rr_));
              const int b0 = obs.csr_offset[rr_];
              const int count = live ? obs.csr_offset[rr_ + 1] - b0 : 0;
              int max_count = count;
  #pragma unroll
              for (int o = 32; o >= kQuad; o >>= 1) max_count = max(max_count, __shfl_xor(max_count, o));
              if (spheres_in_lds)
                pair_loop_culled<true, kQuad, (MINW >= 2), false, CAP>(sph_lds, n_sph_lds, obs.csr_index + b0, count, max_count, sub, P3,
                                                                      V3, A3, lh.P, IP, S, h, nullptr, 0u, 0u, step_table, obs.cylinder != 0);
              else
                pair_loop<kPairsRaggedGlobal, CAP>(step_table, nullptr, nullptr, obs.csr_index + b0, count, max_count, sub,
                                              P3, V3, A3, lh.P, IP, S, h, obs.cylinder != 0);
            }
            RMP2_SEG(2);  // distance leaf: cull + pair trips
  #pragma unroll
            for (int c = 0; c < 6; ++c) S[c] = quad_sum(S[c]);
  #pragma unroll
            for (int c = 0; c < 3; ++c) h[c] = quad_sum(h[c]);
          }
        }
        RMP2_SEG(1);  // target leaf (S, h) / the quad sums of a distance leaf
        if (li == 0) {
          // the columns of MY rows, z_i x (p - o_i) resp. z_i, from the row-joint records read once per pass; rows
          // whose dof does not move the frame get a zero column, so everything below is branch-free
          if (!kRowRecsInRegs && kBatchedRowRecs) {
            // (the 128-register build re-reads the row-joint records per frame: ALL rows' records first, one LDS round trip -- inside
            //  the per-row branches below every read sits in its own basic block with its own wait; a row block that does not move
            //  the frame reads a valid record it does not use)
#pragma unroll
            for (int m = 0; m < ROWS; ++m) {
              const float4 j0 = rjs[m][0], j2 = rjs[m][2];
              rjz[m][0] = j2.y, rjz[m][1] = j2.z, rjz[m][2] = j2.w;
              rjo[m][0] = j0.x, rjo[m][1] = j0.y, rjo[m][2] = j0.z;
            }
          }
#pragma unroll
          for (int m = 0; m < ROWS; ++m) {
            if (MINW >= 2 && ((op.anc_mask >> (kQuad * m)) & 0xfu) == 0u) {  // wave-uniform: no dof of this row block moves
              mycol[m][0] = mycol[m][1] = mycol[m][2] = 0.f;                 // the frame (nor is its record used)
              continue;
            }
            // (bits >= n_dof are never set; bit 16 + j: the frame's origin lies on joint j's axis for every q, its column is
            //  exactly zero in the reference -- rmp2_hip.hip structural_lever_zeros -- and here too, instead of rounding noise)
            const bool act = ((op.anc_mask & ~(op.anc_mask >> 16)) >> (sub + kQuad * m)) & 1u;
            if (!kRowRecsInRegs && !kBatchedRowRecs) {
              const float4 j0 = rjs[m][0], j2 = rjs[m][2];
              rjz[m][0] = j2.y, rjz[m][1] = j2.z, rjz[m][2] = j2.w;
              rjo[m][0] = j0.x, rjo[m][1] = j0.y, rjo[m][2] = j0.z;
            }
            const float d[3] = {P3[0] - rjo[m][0], P3[1] - rjo[m][1], P3[2] - rjo[m][2]};
            float cr[3];
            cross3(rjz[m], d, cr);
#pragma unroll
            for (int c = 0; c < 3; ++c) mycol[m][c] = act ? (rjrev[m] ? cr[c] : rjz[m][c]) : 0.f;
          }
        }
        // pull-back into MY rows:  f_i += col_i . h ;  A[i][j] += (S col_i) . col_j   (rmp.py:165-167).  J^T S J is
        // symmetric (S is): only the BLOCK-UPPER part (row block m, columns j >= 4 m) is accumulated -- 15 of a lane's 27
        // entries for n = 9, so 24 fewer live registers through the pair loops and 45 % fewer products; mirrored once
        // below.  Column j lives in lane (j & 3) as its local row j >> 2: it is broadcast right where it is consumed.
        RMP2_SEG(3);  // Jacobian columns of my rows
        RankOne r1;
        r1.on = false, r1.tr = 0.f, r1.n[0] = r1.n[1] = r1.n[2] = 0.f;
        if (kRank1 && hdr.rank1) r1 = rank_one_of(S);
        float u[ROWS][3];
        float myz[PT ? ROWS : 1][3], tz[PT ? ROWS : 1][3];   // (attached-point leaves: my rows' revolute axes, and Q z_i - c_i x rho)
        if (PT) {
#pragma unroll
          for (int m = 0; m < ROWS; ++m) myz[m][0] = myz[m][1] = myz[m][2] = tz[m][0] = tz[m][1] = tz[m][2] = 0.f;
        }
#pragma unroll
        for (int m = 0; m < ROWS; ++m) {
          u[m][0] = u[m][1] = u[m][2] = 0.f;
          if (((op.anc_mask >> (kQuad * m)) & 0xfu) == 0u) continue;  // wave-uniform: no row of this block moves the frame
          if (kRank1 && hdr.rank1) {  // (wave-uniform) sets without an inertia leaf: componentwise accuracy matters there
            metric_times_column(S, r1, mycol[m], u[m]);
          } else {
            u[m][0] = S[0] * mycol[m][0] + S[1] * mycol[m][1] + S[2] * mycol[m][2];
            u[m][1] = S[1] * mycol[m][0] + S[3] * mycol[m][1] + S[4] * mycol[m][2];
            u[m][2] = S[2] * mycol[m][0] + S[4] * mycol[m][1] + S[5] * mycol[m][2];
          }
          fv[m] += (double)dot3(mycol[m], h);
          if (PT) {  // attached-point leaf:  A_ij += c_j . (W c_i - rho x z_i) + z_j . (Q z_i - c_i x rho),  f_i += z_i . tau
            const bool actz = pt_leaf && rjrev[m] && ((op.anc_mask >> (sub + kQuad * m)) & 1u);
#pragma unroll
            for (int c = 0; c < 3; ++c) myz[m][c] = actz ? rjz[m][c] : 0.f;
            float rz[3], cr2[3];
            cross3(ptRho, myz[m], rz);
            cross3(mycol[m], ptRho, cr2);
            u[m][0] -= rz[0], u[m][1] -= rz[1], u[m][2] -= rz[2];
            tz[m][0] = ptQ[0] * myz[m][0] + ptQ[1] * myz[m][1] + ptQ[2] * myz[m][2] - cr2[0];
            tz[m][1] = ptQ[1] * myz[m][0] + ptQ[3] * myz[m][1] + ptQ[4] * myz[m][2] - cr2[1];
            tz[m][2] = ptQ[2] * myz[m][0] + ptQ[4] * myz[m][1] + ptQ[5] * myz[m][2] - cr2[2];
            fv[m] += (double)dot3(myz[m], ptTau);
          }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
          if (!((op.anc_mask >> j) & 1u)) continue;  // wave-uniform: dof j does not move this frame (its column is 0)
          float cj[3];
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) {
            const float v = mycol[j >> 2][cc];
            cj[cc] = (j & 3) == 0 ? bcast<0>(v) : (j & 3) == 1 ? bcast<1>(v) : (j & 3) == 2 ? bcast<2>(v) : bcast<3>(v);
          }
          float zj[3] = {0.f, 0.f, 0.f};
          if (PT) {
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
              const float v = myz[j >> 2][cc];
              zj[cc] = (j & 3) == 0 ? bcast<0>(v) : (j & 3) == 1 ? bcast<1>(v) : (j & 3) == 2 ? bcast<2>(v) : bcast<3>(v);
            }
          }
#pragma unroll
          for (int m = 0; m < ROWS; ++m)
            if (kQuad * m <= j)  // (a row block without ancestors adds exact zeros)
              A[m][j] += (double)(PT ? dot3(u[m], cj) + dot3(tz[m], zj) : dot3(u[m], cj));
        }
        RMP2_SEG(4);  // pull-back
      }
    }
    // ---- mirror the block-upper part: M[i][j] = M[j][i] for the blocks below the diagonal ------------------------
    // lane s, local row m (global row i = s + 4 m), column 4 b + c (b < m):  the value sits in lane c as A[b][s + 4 m]
    // Sets whose leaves are all symmetric (hdr.sym) never need the lower blocks on the fast path: the identity leaves
    // add symmetric terms and the elimination takes its multipliers from the broadcast pivot row (M[i][k] = M[k][i]);
    // the mirror is then only run for the debug output of M and for the careful solver of a flagged robot.
    auto mirror = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int m = 1; m < ROWS; ++m) {
#pragma unroll
        for (int b = 0; b < m; ++b) {
#pragma unroll
          for (int c = 0; c < kQuad; ++c) {
            double w = 0.0;
#pragma unroll
            for (int s2 = 0; s2 < kQuad; ++s2) {
              if (s2 + kQuad * m < N) {  // compile time: the source column exists
                const double v = A[b][s2 + kQuad * m];
                const double t = c == 0 ? bcastd<0>(v) : c == 1 ? bcastd<1>(v) : c == 2 ? bcastd<2>(v) : bcastd<3>(v);
                w = (sub == s2) ? t : w;
              }
            }
            A[m][kQuad * b + c] = w;
          }
        }
      }
    };
    if (!sym) mirror();
    if (sym) {
      // Symmetric sets: make the kept part EXACTLY symmetric.  Inside a diagonal block both (i, j) and (j, i) are accumulated,
      // as (S c_i) . c_j and (S c_j) . c_i -- equal up to fp32 rounding (1e-7 of terms that reach 1e2..1e3 near contact) --
      // while the blocks below the diagonal exist only as mirrors.  The elimination reads M[i][k] from the pivot row there,
      // i.e. it assumes the Schur complements stay symmetric; with the rounding-level asymmetry of the diagonal blocks it
      // solved a matrix that differs from the stored one by that asymmetry (cond(M) x 1e-7 in qdd: invisible to the 1e-5
      // tolerance away from contact, 1e-4 relative for near-contact robots -- found when the certified strict step was
      // compared with the Jacobi pseudo-inverse of the same stored system).  The upper triangle is the matrix.
#pragma unroll
      for (int m = 0; m < ROWS; ++m) {
#pragma unroll
        for (int c1 = 0; c1 < kQuad; ++c1) {
#pragma unroll
          for (int c2 = c1 + 1; c2 < kQuad; ++c2) {
            if (kQuad * m + c2 < N) {  // compile time: both rows exist
              const double v = A[m][kQuad * m + c2];  // lane c1 holds (4m + c1, 4m + c2)
              const double t = c1 == 0 ? bcastd<0>(v) : c1 == 1 ? bcastd<1>(v) : bcastd<2>(v);
              A[m][kQuad * m + c1] = (sub == c2) ? t : A[m][kQuad * m + c1];  // lane c2: (4m + c2, 4m + c1)
            }
          }
        }
      }
    }

    switch (hdr.prio_tail) {  // (s_setprio takes an immediate)
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      default: __builtin_amdgcn_s_setprio(3); break;
    }
    RMP2_STAMP();  // 3: FK leaves done
    // ---- identity-task-map leaves (row layout) -------------------------------------------------
    if (!(ident_first && pass == 0)) identity_leaves();

    RMP2_STAMP();  // 4: identity leaves done
    // A robot whose state is not finite resolves to NaN (status NONFINITE) whatever its leaves see of it.  The reference gets
    // there by arithmetic -- NaN FK, or 0 * NaN of an out-of-range pair's (metric 0, acceleration NaN), rmp.py:165-167 -- which
    // the quarantine (rmp2_device.h: a non-finite position becomes q = 0 with a NaN velocity) and the culling (an out-of-range
    // pair is never evaluated) can both cut short: a set of distance leaves only, every obstacle out of range, answered 0 for a
    // NaN joint (tools/fuzz_parity.py, seed 504944).  So the non-finiteness is put into the force of the dof it sits on: the
    // velocity tile holds it after the quarantine (a fed Inf / NaN velocity too), the system is then non-finite by construction
    // and every resolve -- pass 0's settle, the closed-form 2 x 2, rmp2_pinv_kernel behind the exported system -- answers NaN.
#pragma unroll
    for (int m = 0; m < ROWS; ++m) {
      const int i = sub + kQuad * m;
      if (i < N) {  // (padding dofs read as 0)
        const float qdv = my_qd[i];
        fv[m] = (fabsf(qdv) < 3.0e38f) ? fv[m] : (double)__builtin_nanf("");
      }
    }
    // optional debug outputs: the combined metric / force before the resolve
    if (!LEAN && pass == 0 && live && (out.M || out.f)) {
      // (the 64-bit row addresses are formed HERE, from an opaque copy of the robot index: hoisted to the prologue -- where
      // the compiler otherwise puts them -- they are spilled by every wave of the register-capped builds and read back
      // only when the debug outputs are asked for)
      int rb = robot;
      if (MINW >= 3) asm volatile("" : "+v"(rb));
      // (sym: the full rows exist only here -- the mirror sits in the block that reads them, so that the lower blocks are
      // dead everywhere else; `live` is the same for the four lanes of a quad, which is all its quad permutes need)
      if (sym && out.M) mirror();
#pragma unroll
      for (int m = 0; m < ROWS; ++m) {
        const int i = sub + kQuad * m;
        if (i < n_dof) {
          // (skip_resolve: the handle's own buffer between this kernel and rmp2_pinv_kernel, robot index FASTEST -- the
          // sixteen quads of the wave write 128 contiguous bytes per entry, the resolve kernel reads a lane per robot)
          const bool soa = hdr.skip_resolve != 0;
          if (out.M) {
#pragma unroll
            for (int j = 0; j < N; ++j)
              if (j < n_dof) out.M[soa ? (size_t)(i * n_dof + j) * R + rb : ((size_t)rb * n_dof + i) * n_dof + j] = A[m][j];
          }
          if (out.f) out.f[soa ? (size_t)i * R + rb : (size_t)rb * n_dof + i] = fv[m];
        }
      }
    }
    if (!LEAN && hdr.skip_resolve) {  // (wave-uniform) the pseudo-inverse of every robot follows in rmp2_pinv_kernel
      flagged = false;
      return;
    }
    // padding dofs of the template: identity rows so that they resolve to qdd = 0
#pragma unroll
    for (int m = 0; m < ROWS; ++m) {
      const int i = sub + kQuad * m;
#pragma unroll
      for (int j = 0; j < N; ++j)
        if (i >= n_dof && j == i) A[m][j] = 1.0;
    }

    if constexpr (N == 2) {
      // ---- 2-dof robots: the closed-form pseudo-inverse for every robot (rmp2_solve.h) ----------------------------
      // No elimination and no careful second pass: sets without an inertia leaf (the TwoJoint half of the mixed fleet)
      // put a few robots per ten thousand on the rank-deficient path, and ONE such robot used to cost its wave a second
      // trip through the frame loop plus a Jacobi iteration in scratch memory -- the launch waits for its slowest wave
      // (measured: 13.7 us for a 32 400-robot shard without such a robot, 24 .. 41 us with two or three).
      double x0, x1;
      if (n_dof == 2) {
        status |= pinv_solve_2x2(bcastd<0>(A[0][0]), bcastd<0>(A[0][1]), bcastd<1>(A[0][0]), bcastd<1>(A[0][1]),
                                 bcastd<0>(fv[0]), bcastd<1>(fv[0]), x0, x1);
      } else {  // one dof on the 2-dof template: 1 x 1
        const double a = bcastd<0>(A[0][0]), f0 = bcastd<0>(fv[0]);
        const bool fin = fabs(a) < 1.7e308 && fabs(f0) < 1.7e308;
        x0 = !fin ? __builtin_nan("") : (a != 0.0 ? f0 / a : 0.0);
        x1 = 0.0;
        if (fin && a == 0.0) status |= RMP2_STATUS_RANK_DROP | RMP2_STATUS_PINV_PATH;
      }
      if (!(fabs(x0) < 1.7e308) || !(fabs(x1) < 1.7e308)) status |= RMP2_STATUS_NONFINITE;
      if (sub == 0) {
        my_out[0] = (float)x0;
        if (n_dof > 1) my_out[1] = (float)x1;
      }
      flagged = false;
      return;
    } else if (pass == 0) {
      // ---- resolve: row-distributed fp64 elimination without row exchanges -------------------
      // (certification and fall-through exactly as lu_solve<N>, rmp2_solve.h)
      double lmax = 0.0;
      double inv_piv[N];
      // symmetric sets: only the block-upper part (row block m, columns >= 4 m) is read and updated;
      // a multiplier whose entry lies below the diagonal blocks is taken from the broadcast pivot row, M[i][k] = M[k][i]
      bool nonfinite_matrix = false;
      double scale = 0.0;          // max |M_ij| of the robot's system (quad-uniform after the two permutes below)
      bool pivots_positive = true; // (quad-uniform) every pivot > 0: with a symmetric M the elimination is an LDL^T of an SPD matrix
      auto eliminate = [&](auto symc) __attribute__((always_inline)) {
        constexpr bool SYME = decltype(symc)::value;
#pragma unroll
        for (int m = 0; m < ROWS; ++m)
#pragma unroll
          for (int j = SYME ? kQuad * m : 0; j < N; ++j) scale = fmax(scale, fabs(A[m][j]));
        scale = fmax(scale, dppd<kXor1>(scale));
        scale = fmax(scale, dppd<kXor2>(scale));
        const double tiny = 1e-11 * scale;
        flagged = !(scale > 0.0) || !(scale < 1.7e308);
        nonfinite_matrix = !(scale < 1.7e308);  // (NaN compares false: NaN or Inf somewhere in the metric)
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const int ks = k & 3, km = k >> 2;
          // broadcast row k (columns k..N-1) and b_k from its owner
          double rowk[N], bk;
#pragma unroll
          for (int j = k; j < N; ++j) {
            const double v = A[km][j];
            rowk[j] = ks == 0 ? bcastd<0>(v) : ks == 1 ? bcastd<1>(v) : ks == 2 ? bcastd<2>(v) : bcastd<3>(v);
          }
          {
            const double v = fv[km];
            bk = ks == 0 ? bcastd<0>(v) : ks == 1 ? bcastd<1>(v) : ks == 2 ? bcastd<2>(v) : bcastd<3>(v);
          }
          const bool bad = !(fabs(rowk[k]) > tiny);
          flagged = flagged || bad;
          pivots_positive = pivots_positive && rowk[k] > 0.0;
          const double inv = bad ? 0.0 : rcpd(rowk[k]);
          inv_piv[k] = inv;
#pragma unroll
          for (int m = 0; m < ROWS; ++m) {
            if (kQuad * m + 3 <= k) continue;  // no lane has a row i = sub + 4m > k in this block
            const int i = sub + kQuad * m;
            double aik;
            if (!SYME || kQuad * m <= k) {
              aik = A[m][k];  // stored: general form, or column k inside / right of the diagonal block
            } else {          // below the diagonal blocks: M[i][k] = M[k][i], element i = 4 m + sub of the pivot row
              aik = rowk[kQuad * m];
#pragma unroll
              for (int c = 1; c < kQuad; ++c)
                if (kQuad * m + c < N) aik = (sub == c) ? rowk[kQuad * m + c] : aik;
            }
            const double l = (i > k) ? aik * inv : 0.0;
            lmax = fmax(lmax, fabs(l));
#pragma unroll
            for (int j = k + 1; j < N; ++j)
              if (!SYME || j >= kQuad * m) A[m][j] = fma(-l, rowk[j], A[m][j]);
            fv[m] = fma(-l, bk, fv[m]);
          }
        }
      };
      eliminate(std::integral_constant<bool, SYM>{});
      lmax = fmax(lmax, dppd<kXor1>(lmax));
      lmax = fmax(lmax, dppd<kXor2>(lmax));
      flagged = flagged || !(lmax <= 1e4);
      if (hdr.strict) {  // (wave-uniform) solve = PINV: the reference's only resolve (rmp.py:153-154)
        // qdd = pinv(M) f with TensorFlow's cutoff 10 n eps sigma_max.  Where EVERY singular value lies above the cutoff the
        // pseudo-inverse IS the inverse, and the elimination above has just computed inv(M) f; so the strict step only has to
        // CERTIFY full rank per robot -- the rest (uncertified, flagged) goes to the Jacobi pseudo-inverse in the careful pass.
        // Certificate (symmetric sets, all pivots > 0, i.e. M = U^T D^-1 U is SPD and the elimination is growth-free):
        //   W = D^-1/2 U,  M = W^T W,  sigma_min(M) = 1 / |W^-1|_2^2 >= 1 / (n |W^-1|_inf^2),
        //   |W^-1|_inf <= |C(W)^-1 e|_inf  with the comparison matrix C(W) = diag|w_ii| - offdiag|w_ij|   (Higham, ASNA 8.2):
        //   one back substitution on absolute values, t_i = (scale sqrt(d_i / scale) + sum_{j > i} |u_ij| t_j) / d_i  (t in units
        //   that make M / scale the matrix: its sigma_max <= n).  Certified:  n t_max^2 * 10 n eps * n * 16 < 1  -- a factor 16
        //   above the cutoff, far beyond what rounding in U (backward error ~ n eps |M| for SPD) or in TF's own SVD can move.
        // (negative pivots: M = W^T sign(D) W, the bound is the same with |d_i|; the elimination without row exchanges is then
        // only trusted with bounded growth -- max |u_ij| <= 4 max |m_ij| and multipliers <= 64 -- while for SPD it is
        // backward stable unconditionally)
        // General (non-symmetric) sets -- JointLimitAvoidance scales columns, quirk Q2 --: M = L U with unit lower L,
        //   |M^-1|_inf <= |U^-1|_inf |L^-1|_inf,  |U^-1|_inf <= |C(U)^-1 e|_inf (the same back substitution, right-hand side e),
        //   |L^-1|_inf <= (1 + l_max)^(n - 1)  (l_max = the largest multiplier; the elimination keeps no L),
        //   sigma_min(M) >= 1 / (sqrt(n) |M^-1|_inf);  certified:  sqrt(n) t_max (1 + l_max)^(n-1) * 10 n eps * n * 16 < 1,
        // with growth <= 4 and multipliers <= 4 (so that the elimination itself is trustworthy without row exchanges).
        double umax = 0.0;
#pragma unroll
        for (int m = 0; m < ROWS; ++m)
#pragma unroll
          for (int j = kQuad * m; j < N; ++j) umax = fmax(umax, fabs(A[m][j]));  // (general form: up to three stale entries left of the
        umax = fmax(umax, dppd<kXor1>(umax));                                    //  diagonal are included: only ever conservative)
        umax = fmax(umax, dppd<kXor2>(umax));
        bool certified = !flagged && (SYM ? (pivots_positive || (umax <= 4.0 * scale && lmax <= 64.0))
                                          : (umax <= 4.0 * scale && lmax <= 4.0));
        {
          double tr[ROWS], tmax = 0.0;
#pragma unroll
          for (int m = 0; m < ROWS; ++m) {
            if (SYM) {
              double dm = 1.0;  // my row's pivot u_ii, i = sub + 4 m (rows beyond N: unused)
#pragma unroll
              for (int c = 0; c < kQuad; ++c)
                if (kQuad * m + c < N) dm = (sub == c) ? A[m][kQuad * m + c] : dm;
              // an UPPER bound of sqrt(d / scale) from the fp32 square root (argument in [1e-11, ~1]: no over- / underflow)
              const float rt = __builtin_sqrtf((float)(fabs(dm) / scale)) * 1.000001f + 1e-30f;
              tr[m] = scale * (double)rt;
            } else {
              tr[m] = scale;
            }
          }
#pragma unroll
          for (int i = N - 1; i >= 0; --i) {
            const int is = i & 3, im = i >> 2;
            const double sv = tr[im] * fabs(inv_piv[i]);
            const double ti = is == 0 ? bcastd<0>(sv) : is == 1 ? bcastd<1>(sv) : is == 2 ? bcastd<2>(sv) : bcastd<3>(sv);
            tmax = fmax(tmax, ti);
#pragma unroll
            for (int m = 0; m < ROWS; ++m) {
              if (kQuad * m >= i) continue;  // (rows at or below i of the same block add into right-hand sides already consumed)
              tr[m] = fma(fabs(A[m][i]), ti, tr[m]);
            }
          }
          constexpr double kEps = 2.220446049250313e-16;
          if (SYM) {
            // n^3 * 160 eps * t_max^2 < 1   (n = N: the padding rows are identity rows of the same system)
            certified = certified && (tmax * tmax < 1.0 / (160.0 * (double)(N * N * N) * kEps));
          } else {
            const double l1 = 1.0 + lmax, l2 = l1 * l1, l4 = l2 * l2, l8 = l4 * l4;  // (1 + l_max)^8 >= (1 + l_max)^(N - 1), N <= 9
            static_assert(N <= 9, "the bound on |L^-1| is written for n <= 9");
            // n^2.5 * 160 eps * t_max (1 + l_max)^(n-1) < 1;  sqrt(9) = 3 bounds sqrt(n)
            certified = certified && (tmax * l8 < 1.0 / (160.0 * 3.0 * (double)(N * N) * kEps));
          }
        }
        flagged = flagged || !certified;
      }
      // back substitution, column oriented: the owner finishes x_i and broadcasts it, then every
      // lane retires column i from the right-hand sides of ITS rows (independent FMAs: the
      // dependent chain per unknown is one multiply + one broadcast)
      double x[N];
#pragma unroll
      for (int i = N - 1; i >= 0; --i) {
        const int is = i & 3, im = i >> 2;
        const double s = fv[im] * inv_piv[i];
        x[i] = is == 0 ? bcastd<0>(s) : is == 1 ? bcastd<1>(s) : is == 2 ? bcastd<2>(s) : bcastd<3>(s);
#pragma unroll
        for (int m = 0; m < ROWS; ++m) {
          if (kQuad * m >= i) continue;  // rows of this block are all >= i: nothing above the diagonal
          fv[m] = fma(-A[m][i], x[i], fv[m]);
        }
      }
      bool finite = true;
#pragma unroll
      for (int i = 0; i < N; ++i)
        if (i < n_dof) finite = finite && (fabs(x[i]) < 1.7e308);
      flagged = flagged || !finite;
      // A system that is not finite to begin with (NaN / Inf state, e.g. a robot past the JointVelocityCap pole, quirk Q4)
      // resolves to NaN in the reference (tf.linalg.pinv of a NaN matrix) and in the careful solver; it is settled HERE --
      // q-double-dot = NaN, status NONFINITE -- instead of sending its whole wave through the second pass: in a closed-loop
      // rollout such a robot stays non-finite for the rest of the horizon and would cost its wave a second frame loop
      // every control step (the launch waits for its slowest wave).
      {
        double fs = 0.0;
#pragma unroll
        for (int m = 0; m < ROWS; ++m) fs = fmax(fs, fabs(fv[m]) < 1.7e308 ? 0.0 : 1.0);
        fs = fmax(fs, dppd<kXor1>(fs));
        fs = fmax(fs, dppd<kXor2>(fs));
        const bool sys_nonfinite = nonfinite_matrix || fs > 0.0;
        if (sys_nonfinite) {
#pragma unroll
          for (int i = 0; i < N; ++i) x[i] = __builtin_nan("");
          status |= RMP2_STATUS_NONFINITE | RMP2_STATUS_PINV_PATH;
          flagged = false;
        }
      }
      // slot 0 doubles as the qdd tile: a flagged robot keeps its slots for the careful pass below
      if (sub == 0 && !flagged) {
        static_assert(N <= kSlot, "the qdd tile of a robot is its first frame slot");
        // (streamed explicit pairs: the frame slots hold the [p v a] image and the chunk buffer -- the robot's own q row takes its qdd)
        int mo_off = STREAM ? QuadLds<N>::kQ + g * N : QuadLds<N>::kLoc + gi_loc(g, n_ops);  // (formed here: hoisted to the prologue it is spilled and reloaded per dof)
        if (MINW >= 3) asm volatile("" : "+v"(mo_off));
        float* mo = lds + mo_off;
#pragma unroll
        for (int i = 0; i < N; ++i) mo[i] = (float)x[i];  // (padding dofs resolve to 0; only n_dof values are stored to HBM)
      }
      RMP2_STAMP();  // 5: LU done
      return;  // (the caller runs the careful pass if any robot of the wave was flagged)
    } else if (flagged) {
      // ---- rare path: gather the whole system into every lane of the quad, careful solve -------
      if (sym) mirror();  // (the fast path of a symmetric set keeps the block-upper part only)
      double W[N * (N + 1)], T[N * (N + 1)], xp[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int is = i & 3, im = i >> 2;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const double v = A[im][j];
          W[i * (N + 1) + j] = is == 0 ? bcastd<0>(v) : is == 1 ? bcastd<1>(v) : is == 2 ? bcastd<2>(v) : bcastd<3>(v);
        }
        const double v = fv[im];
        W[i * (N + 1) + N] = is == 0 ? bcastd<0>(v) : is == 1 ? bcastd<1>(v) : is == 2 ? bcastd<2>(v) : bcastd<3>(v);
      }
      // (strict: the pseudo-inverse is the asked-for resolve, not a fall-back; the robot is marked "not certified full rank")
      status |= hdr.strict ? RMP2_STATUS_JACOBI : RMP2_STATUS_PINV_PATH;
      bool finite_in = true;  // a metric / force with NaN or Inf resolves to NaN (as the reference's pinv does):
      for (int i = 0; i < N * (N + 1); ++i) finite_in = finite_in && (fabs(W[i]) < 1.7e308);  // no point iterating on it
      if (!finite_in) {
        for (int i = 0; i < N; ++i) xp[i] = __builtin_nan("");
      } else if (hdr.strict || !lu_pivot_compact(W, T, N, xp)) {  // (strict: no elimination for an uncertified robot)
        const int dropped = pinv_solve_compact(W, N, n_dof, xp);
        if (dropped) status |= RMP2_STATUS_RANK_DROP;
      }
      bool finite = true;
      int co_off = STREAM ? QuadLds<N>::kQ + g * N : QuadLds<N>::kLoc + gi_loc(g, n_ops);
      if (MINW >= 3) asm volatile("" : "+v"(co_off));
      float* co = lds + co_off;
      for (int i = 0; i < n_dof; ++i) {
        finite = finite && (fabs(xp[i]) < 1.7e308);
        if (sub == 0) co[i] = (float)xp[i];
      }
      if (!finite) status |= RMP2_STATUS_NONFINITE;
    }
  };
  run_pass(std::integral_constant<int, 0>{});
  if (N != 2 && __any(flagged && live)) run_pass(std::integral_constant<int, 1>{});
  if (ro.substeps > 0) {
    // plant: qdd held, semi-implicit Euler (qd += dt qdd; q += dt qd); every lane advances ITS dofs
    int gio = gi;
    if (MINW >= 3) asm volatile("" : "+v"(gio));
    float* qw = &lds[QuadLds<N>::kQ + gio * N];
    float* qdw = &lds[QuadLds<N>::kQd + gio * N];
#pragma unroll
    for (int m = 0; m < ROWS; ++m) {
      const int i = sub + kQuad * m;
      // (live: the quads beyond the fleet's tail alias the LAST live robot's tile row (gi) -- with ITS inputs only in the staged
      //  builds; in the scalar-cache builds they read robot 0's goal and an empty list, resolve to another qdd, and -- the highest
      //  lane winning an LDS write -- advanced the last robot of a partial wave with it.  Found when the latency build left the
      //  dispatch (round 5): rollout == step loop is tested on small fleets, where it had always run.)
      if (i < n_dof && live) {
        const float acc = my_out[i];
        float qi_ = qw[i], qdi = qdw[i];
        for (int t = 0; t < ro.substeps; ++t) {
          qdi = fmaf(ro.dt, acc, qdi);
          qi_ = fmaf(ro.dt, qdi, qi_);
        }
        quarantine(qi_, qdi);
        qw[i] = qi_;
        qdw[i] = qdi;
      }
    }
  }
  }  // rollout iterations

  // ---- coalesced store of the qdd tile (and of the advanced state after a rollout) ---------------------
  __syncthreads();
  int lane_o = lane;  // (opaque copy: keeps the store addresses from being formed in the prologue and spilled)
  if (MINW >= 3) asm volatile("" : "+v"(lane_o));
  if (ro.q_out) {
    const int count = n_live * n_dof;
    for (int i = lane_o; i < count; i += kWave) {
      const int rr = i / n_dof, jj = i - rr * n_dof;
      const float qdv = lds[QuadLds<N>::kQd + rr * N + jj];
      ro.q_out[(size_t)r0 * n_dof + i] = (qdv != qdv) ? qdv : lds[QuadLds<N>::kQ + rr * N + jj];  // (quarantine, rmp2_device.h)
      ro.qd_out[(size_t)r0 * n_dof + i] = qdv;
    }
  }
  {
    const float* tile = &lds[STREAM ? QuadLds<N>::kQ : QuadLds<N>::kLoc];
    const int tile_stride = STREAM ? N : out_stride;
    const int count = min(kRobotsPerWave, R - r0) * n_dof;
    float* go = out.qdd + (size_t)r0 * n_dof;
    for (int i = lane_o; i < count; i += kWave) {
      const int rr = i / n_dof, jj = i - rr * n_dof;
      go[i] = tile[rr * tile_stride + jj];
    }
  }
  if (out.status && live && sub == 0) out.status[robot] = status;
#ifdef RMP2_STAMPS
  RMP2_STAMP();  // 6: stored
  if (lane == 0 && out.M == nullptr && out.f != nullptr) {  // diagnostic convention: the stamps follow the f rows
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(out.f) + (size_t)R * n_dof + (size_t)block_idx * 16;
    for (int i = 0; i < 8; ++i) dst[i] = i < st_n ? st_[i] : 0ull;
    for (int i = 0; i < 8; ++i) dst[8 + i] = seg_[i];
  }
#endif
}

// The control step of ONE engine: one wave (16 robots) per block.
template <int N, int SLOTS, int MINW, bool STAGE, bool CAP, bool SYM = false, int OBS = kObsAny, int FLAVOR = kGeneral,
          bool PT = false>
__global__ void __launch_bounds__(kWave, MINW)
rmp2_step_quad_kernel(const DevProgram* __restrict__ prog, QuadHdr hdr, const float* __restrict__ q,
                      const float* __restrict__ qd, const float* __restrict__ goal, int goal_stride, ObsArgs obs,
                      OutArgs out, RolloutArgs ro_arg, int R) {
  quad_step_body<N, SLOTS, MINW, STAGE, CAP, SYM, OBS, FLAVOR, PT>(prog, hdr, q, qd, goal, goal_stride, obs, out, ro_arg, R,
                                                                    (int)blockIdx.x);
}

// The control steps of TWO engines in ONE launch (a fleet shard that holds two robot types, BASELINE config 5): the first
// `split` blocks run engine A's program on A's robots, the rest engine B's.  Two launches on two streams need a fork and a
// join fence (~5 us each on this runtime) and two launches on one stream serialise; one grid needs neither, and the short
// program's waves fill the slots the long one's leave.  Wavefronts stay type-homogeneous (a block is one wave).
struct QuadCall {
  const DevProgram* prog;
  QuadHdr hdr;
  const float* q;
  const float* qd;
  const float* goal;
  int goal_stride;
  ObsArgs obs;
  OutArgs out;
  int R;
};
template <int NA, int SA, bool SYMA, int NB, int SB, bool SYMB, int MINW, int OBS>
__global__ void __launch_bounds__(kWave, MINW)
rmp2_step_quad_pair_kernel(QuadCall a, QuadCall b, int split) {
  const RolloutArgs ro{1, 0, 0.f, nullptr, nullptr, 0};
  if ((int)blockIdx.x < split)
    quad_step_body<NA, SA, MINW, false, false, SYMA, OBS, kPlainStep, false>(a.prog, a.hdr, a.q, a.qd, a.goal, a.goal_stride,
                                                                            a.obs, a.out, ro, a.R, (int)blockIdx.x);
  else
    quad_step_body<NB, SB, MINW, false, false, SYMB, OBS, kPlainStep, false>(b.prog, b.hdr, b.q, b.qd, b.goal, b.goal_stride,
                                                                            b.obs, b.out, ro, b.R, (int)blockIdx.x - split);
}

}  // namespace rmp2
