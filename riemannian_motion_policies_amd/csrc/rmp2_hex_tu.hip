// rmp2_hex_tu.hip -- instantiations of the hex mapping (rmp2_hex.h) for ONE template size: compiled once per RMP2_TU_N
// (2, 9, 16) by __graft_entry__.build_hip.
#include "rmp2_host.h"

namespace rmp2 {
namespace {

constexpr size_t kLdsDefault = 64 * 1024;  // dynamic LDS a launch may ask for without raising the function attribute

template <int N, int WAVES, bool ROLL>
// false: the launch could not be prepared (the function attribute for more than 64 KiB of dynamic LDS was refused) -- nothing was
// launched and the caller falls back to the quad mapping instead of meeting a generic launch error later
bool launch_hex_w(const rmp2_handle* h, const float* q, const float* qd, const float* goal, int gs, const ObsArgs& o,
                  const OutArgs& out, const RolloutArgs& ro, int R, hipStream_t s) {
  const int per_block = kHexRobots * WAVES;
  const int blocks = (R + per_block - 1) / per_block;
  const int n_sph_lds = (o.mode == RMP2_OBS_SHARED_SPHERES || o.mode == RMP2_OBS_RAGGED_SPHERES)
                            ? std::min(o.n_spheres, kLdsSpheres) : 0;
  const size_t bytes = hex_lds_bytes<N>(WAVES, h->n_ops_step, h->hex_blob16, sphere_lds_floats(o.capsule, n_sph_lds),
                                        h->has_point);
  const QuadHdr hdr{h->n_ops_step, h->n_dof, h->n_id_leaves, h->n_leaves, h->goal_floats, h->n_leaf_ops, h->rev_mask,
                    h->hex_levels, h->n_fk_leaves, h->hex_is_chain, {h->dof_ops[0], h->dof_ops[1], h->dof_ops[2]}, h->cull_c0,
                    // 0 AUTO; 1 the pseudo-inverse on every robot; 2 / 3 (symmetric set) the certifying elimination (rmp2_hex.h)
                    !h->strict ? 0 : (hex_certifies_strict(h) ? (h->symmetric ? 3 : 2) : (h->symmetric ? 5 : 1)),
                    /*prio_tail*/ 0, /*skip_resolve*/ 0, /*has_point*/ h->has_point ? 1 : 0, /*rank1*/ h->likely_singular ? 1 : 0};
  const uint4* blob = static_cast<const uint4*>(h->d_hex_blob);
  h->last_kernel = h->strict ? (hex_certifies_strict(h)
                                    ? "rmp2_step_hex_kernel (16 lanes per robot; strict: full rank certified per robot, Jacobi pseudo-inverse for the rest)"
                                    : "rmp2_step_hex_kernel (16 lanes per robot, strict pseudo-inverse)")
                             : "rmp2_step_hex_kernel (16 lanes per robot)";
#define RMP2_HEX_LAUNCH(CAP, PT)                                                                                          \
  do {                                                                                                                    \
    auto kern = rmp2_step_hex_kernel<N, CAP, WAVES, ROLL, PT>;                                                            \
    if (bytes > kLdsDefault &&                                                                                             \
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) { \
      (void)hipGetLastError();                                                                                             \
      return false;                                                                                                        \
    }                                                                                                                      \
    RMP2_STEP_LAUNCH(h, kern, dim3(blocks), dim3(kWave * WAVES), bytes, s, blob, h->hex_blob16, hdr, q, qd, goal, gs, o, out, \
                     ro, R);                                                                                              \
  } while (0)
  if (h->has_point)  // attached-point leaves (they roll out when their pairs come from a table + link capsules, which they read
    RMP2_HEX_LAUNCH(false, true);  // from global memory: spheres or capsules alike, so the sphere-mode build serves both)
  else if (o.capsule)
    RMP2_HEX_LAUNCH(true, false);
  else
    RMP2_HEX_LAUNCH(false, false);
#undef RMP2_HEX_LAUNCH
  return true;
}

// A launch may ask for up to 64 KiB of dynamic LDS as it is; beyond that (up to the CU's 160 KiB) the function attribute is
// raised first (launch_hex_w).
constexpr size_t kLdsLimit = 160 * 1024;

template <int N>
size_t hex_bytes(const rmp2_handle* h, const ObsArgs& o, int waves) {
  const int n_sph_lds = (o.mode == RMP2_OBS_SHARED_SPHERES || o.mode == RMP2_OBS_RAGGED_SPHERES)
                            ? std::min(o.n_spheres, kLdsSpheres) : 0;
  return hex_lds_bytes<N>(waves, h->n_ops_step, h->hex_blob16, sphere_lds_floats(o.capsule, n_sph_lds), h->has_point);
}

// big programs (many frames / leaves) do not fit four waves' working sets into one block's LDS: one wave per
// block then; if even that does not fit the caller falls back to the quad kernel
template <int N>
bool launch_hex(const rmp2_handle* h, const float* q, const float* qd, const float* goal, int gs, const ObsArgs& o,
                const OutArgs& out, const RolloutArgs& ro, int R, hipStream_t s) {
  const bool rollout = ro.n_iters != 1 || ro.substeps != 0;
  // four-wave blocks while at least two of them fit a CU (the block shares one staged program); one-wave blocks beyond
  if (h->hex_waves != 1 && hex_bytes<N>(h, o, 4) <= kLdsLimit / 2) {
    if (rollout)
      return launch_hex_w<N, 4, true>(h, q, qd, goal, gs, o, out, ro, R, s);
    else
      return launch_hex_w<N, 4, false>(h, q, qd, goal, gs, o, out, ro, R, s);
  }
  if (!rollout && hex_bytes<N>(h, o, 1) <= kLdsLimit) {  // (the rollout build exists for four-wave blocks only)
    return launch_hex_w<N, 1, false>(h, q, qd, goal, gs, o, out, ro, R, s);
  }
  return false;
}

}  // namespace

#define RMP2_CAT_(a, b) a##b
#define RMP2_CAT(a, b) RMP2_CAT_(a, b)
bool RMP2_CAT(launch_hex_n, RMP2_TU_N)(const rmp2_handle* h, const float* q, const float* qd, const float* goal, int gs, const ObsArgs& o,
                                       const OutArgs& out, const RolloutArgs& ro, int R, hipStream_t s) {
  return launch_hex<RMP2_TU_N>(h, q, qd, goal, gs, o, out, ro, R, s);
}

}  // namespace rmp2
