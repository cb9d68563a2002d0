// rmp2_quad_pair_tu.hip -- the fused two-engine grid (rmp2_quad.h rmp2_step_quad_pair_kernel): a 2-dof robot type without
// branch points (the TwoJoint) beside a 3..9-dof type with one (the Panda), plain control steps on shared or ragged sphere
// tables -- what a shard of BASELINE config 5 holds.  Other pairs fall back to two launches (rmp2_step_pair).
#include "rmp2_host.h"

namespace rmp2 {
namespace {

template <int N>
size_t quad_lds_bytes(const rmp2_handle* h, const ObsArgs& o) {
  const int n_sph_lds = std::min(o.n_spheres, kLdsSpheres);
  return sizeof(float) * (QuadLds<N>::kFloats + kSlot * kRobotsPerWave * quad_slots(h->n_ops_step) +
                          quad_table_floats(false, n_sph_lds));
}

}  // namespace

bool launch_quad_pair(const rmp2_handle* ha, const float* qa, const float* qda, const float* goala, int gsa, const ObsArgs& oa,
                      const OutArgs& outa, int Ra, const rmp2_handle* hb, const float* qb, const float* qdb, const float* goalb,
                      int gsb, const ObsArgs& ob, const OutArgs& outb, int Rb, hipStream_t s) {
  // A = the 2-dof engine, B = the 3..9-dof engine (either order is accepted)
  if (ha->n_template == 9 && hb->n_template == 2)
    return launch_quad_pair(hb, qb, qdb, goalb, gsb, ob, outb, Rb, ha, qa, qda, goala, gsa, oa, outa, Ra, s);
  const auto lean = [](const rmp2_handle* h, const ObsArgs& o, const OutArgs& out) {
    // (solve = PINV: the 2-dof part's closed form is the pseudo-inverse; the 3..9-dof part where its step certifies full rank)
    return (!h->strict || h->n_template == 2 || quad_certifies_strict(h)) && !h->has_point && h->goal_floats <= 16 && !out.M && !out.f && !o.capsule && !o.link_caps &&
           (o.mode == RMP2_OBS_SHARED_SPHERES || o.mode == RMP2_OBS_RAGGED_SPHERES) && o.n_spheres <= kLdsSpheres &&
           h->kernel_choice == 0;
  };
  if (!(ha->n_template == 2 && ha->n_slots == 0 && hb->n_template == 9 && hb->n_slots == 1)) return false;
  if (!lean(ha, oa, outa) || !lean(hb, ob, outb) || oa.mode != ob.mode || hb->likely_singular) return false;
  // the fused grid is a throughput build (scalar-cache program walk, two waves per SIMD and more): fleets that the
  // dispatcher would give to the 16-lanes-per-robot mapping (<= 8 192 robots) keep their two launches
  if (Ra <= 8192 || Rb <= 8192) return false;
  const int blocks_a = (Ra + kRobotsPerWave - 1) / kRobotsPerWave, blocks_b = (Rb + kRobotsPerWave - 1) / kRobotsPerWave;
  const size_t bytes = std::max(quad_lds_bytes<2>(ha, oa), quad_lds_bytes<9>(hb, ob));
  const QuadCall a{ha->d_prog, make_quad_hdr(ha), qa, qda, goala, gsa, oa, outa, Ra};
  const QuadCall b{hb->d_prog, make_quad_hdr(hb), qb, qdb, goalb, gsb, ob, outb, Rb};
  ha->last_kernel = hb->last_kernel = "rmp2_step_quad_pair_kernel (4 lanes per robot, two engines in one grid)";
#define RMP2_PAIR_LAUNCH(SYMB, OBS)                                                                                      \
  hipLaunchKernelGGL((rmp2_step_quad_pair_kernel<2, 0, false, 9, 1, SYMB, 2, OBS>), dim3(blocks_a + blocks_b), dim3(kWave), \
                     bytes, s, a, b, blocks_a)
  const bool symb = hb->symmetric;
  if (oa.mode == RMP2_OBS_RAGGED_SPHERES) {
    if (symb) RMP2_PAIR_LAUNCH(true, RMP2_OBS_RAGGED_SPHERES); else RMP2_PAIR_LAUNCH(false, RMP2_OBS_RAGGED_SPHERES);
  } else {
    if (symb) RMP2_PAIR_LAUNCH(true, RMP2_OBS_SHARED_SPHERES); else RMP2_PAIR_LAUNCH(false, RMP2_OBS_SHARED_SPHERES);
  }
#undef RMP2_PAIR_LAUNCH
  return true;
}

}  // namespace rmp2
