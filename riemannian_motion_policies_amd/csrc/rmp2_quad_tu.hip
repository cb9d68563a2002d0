// rmp2_quad_tu.hip -- instantiations of the quad mapping (rmp2_quad.h) for ONE template size and slot count:
// compiled once per (RMP2_TU_N, RMP2_TU_SLOTS) pair by __graft_entry__.build_hip, e.g. -DRMP2_TU_N=9 -DRMP2_TU_SLOTS=1.
#include "rmp2_host.h"

namespace rmp2 {
namespace {

template <int N, int SLOTS>
void launch_quad(const rmp2_handle* h, const float* q, const float* qd, const float* goal, int gs, const ObsArgs& o,
                 const OutArgs& out, const RolloutArgs& ro, int R, hipStream_t s) {
  const int blocks = (R + kRobotsPerWave - 1) / kRobotsPerWave;
  const int n_sph_lds = (o.mode == RMP2_OBS_SHARED_SPHERES || o.mode == RMP2_OBS_RAGGED_SPHERES)
                            ? std::min(o.n_spheres, kLdsSpheres) : 0;
  const size_t lds_bytes = sizeof(float) * (QuadLds<N>::kFloats + kSlot * kRobotsPerWave * quad_slots(h->n_ops_step) +
                                            quad_table_floats(o.capsule, n_sph_lds));
  const bool with_records = h->has_point || o.link_caps;  // attached-point leaves, link geometry: the builds with rotation records
  const size_t pt_bytes = with_records ? sizeof(float) * ((o.link_caps && !h->has_point) ? kPtSlotLink : kPtSlot) * kRobotsPerWave * quad_slots(h->n_ops_step)
                                       : 0;  // (rotation, w, alpha per frame; link geometry: the rotation)
  const size_t stage_bytes = sizeof(DevOp) * h->n_ops_step + sizeof(DevLeaf) * h->n_leaves +
                             sizeof(int32_t) * (2 * RMP2_MAX_LEAVES + kMaxOps) + sizeof(float) * 16 * kRobotsPerWave;
  const QuadHdr hdr = make_quad_hdr(h);
  // latency build for grids that put at most one wave on a SIMD (256 CUs x 4): program staged in LDS, all 512 registers;
  // throughput builds beyond: scalar-cache program walk, capped at 256, 168 or 128 registers (two, three or four waves per
  // SIMD).  More waves retire the leaf phases faster per robot, but a SIMD's share of the fleet has to divide into rounds.
  // With the per-mode plain builds of round 3 (no spills at two and three waves, 16 spilled dwords at four in the
  // symmetric form) the three-wave build no longer wins anywhere; config 3, us per step (profiles/r03_quad_minw_ab.txt),
  // b = waves a SIMD owes to the fleet:
  //       b      1     1.5    2    2.5    3    3.5    4     5     6     8     12     16
  //   2 waves  30.2  33.6  33.5  37.7  38.1  59.4  60.6  65.9  73.3   99.3  140.2  190.1
  //   3 waves  30.3  34.6  34.4  38.9  39.2  62.2  63.6  67.9  74.5  102.2  142.7  200.4
  //   4 waves  30.5  36.2  36.7  40.7  41.5  46.5  50.1  73.3  76.6   90.5  130.3  170.6
  // two waves up to ceil(b) = 3 and for 5 and 6, four for 4 and from 7 on.  Sets with a JointLimitAvoidance leaf (general
  // form: the elimination works on full rows) follow the same rule since round 3 -- config 3 + JointLimitAvoidance:
  //       b      2     3    3.5    4     6     8     16
  //   2 waves  37.5  43.5  65.0  65.4  80.6  107.7  205.2
  //   4 waves  41.8  47.7  52.5  57.3  88.5  103.4  193.8
  const bool latency = blocks <= h->quad_latency_blocks && h->goal_floats <= 16;
  const bool symk = h->symmetric && N == 9;
  int minw = h->quad_minw;
  if (minw == 0) {
    const int bc = (blocks + h->n_simd - 1) / h->n_simd;  // ceil(b)
    minw = (bc == 4 || bc >= 7) ? 4 : 2;
    // explicit pairs (interface B) are bound by memory latency, not by issue slots: the two-wave build keeps a whole leaf's
    // loads (and the next leaf's prefetch) in flight per wave -- 123.7 us per step at 65 536 robots against 148.7 with four
    // (round 4: with the leaf's slots loaded in a rolling window the 168-register build is spill-free too; it wins where it
    // turns two rounds into one -- 49 152 robots: 80.5 against 89.5 us --, nowhere else: 65 536 robots 129.8 / 118.3 us with
    // three / four waves against 104.6 with two)
    if (o.mode == RMP2_OBS_EXPLICIT_PAIRS) minw = bc == 3 ? 3 : 2;
  }
  size_t bytes = (latency ? lds_bytes + stage_bytes : lds_bytes) + pt_bytes;
  h->last_kernel = quad_certifies_strict(h)
                       ? "rmp2_step_quad_kernel (4 lanes per robot; strict: full rank certified per robot, Jacobi pseudo-inverse for the rest)"
                       : "rmp2_step_quad_kernel (4 lanes per robot)";
  // explicit pairs (interface B) in the plain two-wave build: streamed half a leaf ahead by LDS-DMA when every leaf segment of
  // the two arrays is 16-byte aligned (global_load_lds_dwordx4): 6 KiB more LDS per wave (16.5 KB: eight waves still share a CU)
  ObsArgs og = o;
  {
    const bool plain_explicit = o.mode == RMP2_OBS_EXPLICIT_PAIRS && !latency && minw == 2 && !with_records && ro.n_iters == 1 &&
                                ro.substeps == 0 && !ro.q_out && !out.M && !out.f && !o.capsule;
    bool aligned = (o.n_pairs % 4) == 0 && (reinterpret_cast<uintptr_t>(o.p_link) % 16) == 0 &&
                   (reinterpret_cast<uintptr_t>(o.p_obs) % 16) == 0 && h->explicit_glds;
    for (int l = 0; l <= h->n_leaves && aligned; ++l) aligned = (h->h_pair_begin[l] % 4) == 0;
    if (plain_explicit && aligned) {
      og.glds = 1;
      bytes += sizeof(float) * (kGldsBuf + kGldsList);  // chunk image + the quads' in-range lists: 8 KiB, 18.4 KB per wave
    }
  }
#define RMP2_QUAD_LAUNCH(MINW, STAGE, CAP, SYM, OBS, FLAVOR)                                                            \
  RMP2_STEP_LAUNCH(h, (rmp2_step_quad_kernel<N, SLOTS, MINW, STAGE, CAP, SYM, OBS, FLAVOR>), dim3(blocks), dim3(kWave), bytes, s, \
                   h->d_prog, hdr, q, qd, goal, gs, og, out, ro, R)
  // the symmetric form (block-upper system through the identity leaves and the elimination) exists for the 3..9-dof
  // template with sphere tables (symk above); everything else takes the general form
#define RMP2_QUAD_BY_CAP(MINW, STAGE)                                                                                   \
  do {                                                                                                                  \
    if (o.capsule && symk) RMP2_QUAD_LAUNCH(MINW, STAGE, true, (N == 9), kObsAny, kGeneral);                               \
    else if (o.capsule) RMP2_QUAD_LAUNCH(MINW, STAGE, true, false, kObsAny, kGeneral);                                     \
    else if (symk) RMP2_QUAD_LAUNCH(MINW, STAGE, false, (N == 9), kObsAny, kGeneral);                                      \
    else RMP2_QUAD_LAUNCH(MINW, STAGE, false, false, kObsAny, kGeneral);                                                   \
  } while (0)
  // plain control steps of the throughput builds (no debug outputs, no rollout, sphere primitives): one instantiation per
  // obstacle mode
#define RMP2_QUAD_PLAIN_SYM(MINW, SYM)                                                                                  \
  do {                                                                                                                  \
    switch (o.mode) {                                                                                                   \
      case RMP2_OBS_SHARED_SPHERES: RMP2_QUAD_LAUNCH(MINW, false, false, SYM, RMP2_OBS_SHARED_SPHERES, kPlainStep); break;    \
      case RMP2_OBS_RAGGED_SPHERES: RMP2_QUAD_LAUNCH(MINW, false, false, SYM, RMP2_OBS_RAGGED_SPHERES, kPlainStep); break;    \
      case RMP2_OBS_EXPLICIT_PAIRS: RMP2_QUAD_LAUNCH(MINW, false, false, SYM, RMP2_OBS_EXPLICIT_PAIRS, kPlainStep); break;    \
      default: RMP2_QUAD_LAUNCH(MINW, false, false, SYM, RMP2_OBS_NONE, kPlainStep); break;                                   \
    }                                                                                                                   \
  } while (0)
#define RMP2_QUAD_PLAIN(MINW)                                                                                           \
  do {                                                                                                                  \
    if (o.capsule && symk) RMP2_QUAD_LAUNCH(MINW, false, true, (N == 9), RMP2_OBS_SHARED_SPHERES, kPlainStep);                \
    else if (o.capsule) RMP2_QUAD_LAUNCH(MINW, false, true, false, RMP2_OBS_SHARED_SPHERES, kPlainStep);                      \
    else if (symk) RMP2_QUAD_PLAIN_SYM(MINW, (N == 9));                                                                 \
    else RMP2_QUAD_PLAIN_SYM(MINW, false);                                                                              \
  } while (0)
#ifdef RMP2_STAMPS
  const bool plain = ro.n_iters == 1 && ro.substeps == 0 && !ro.q_out && !out.M && (!o.capsule || o.mode == RMP2_OBS_SHARED_SPHERES);  // (stamps ride behind out.f)
#else
  const bool plain = ro.n_iters == 1 && ro.substeps == 0 && !ro.q_out && !out.M && !out.f &&
                     (!o.capsule || o.mode == RMP2_OBS_SHARED_SPHERES);  // (capsule tables: the shared-table plain build only)
#endif
  // fused rollouts of sphere-table / obstacle-free sets (no debug outputs): the lean rollout builds -- the general build
  // spills 76 dwords at 128 registers (88 us per control step at 65 536 robots against 41 for a plain step)
#define RMP2_QUAD_ROLL_SYM(MINW, SYM)                                                                                   \
  do {                                                                                                                  \
    switch (o.mode) {                                                                                                   \
      case RMP2_OBS_SHARED_SPHERES: RMP2_QUAD_LAUNCH(MINW, false, false, SYM, RMP2_OBS_SHARED_SPHERES, kPlainRollout); break; \
      case RMP2_OBS_RAGGED_SPHERES: RMP2_QUAD_LAUNCH(MINW, false, false, SYM, RMP2_OBS_RAGGED_SPHERES, kPlainRollout); break; \
      default: RMP2_QUAD_LAUNCH(MINW, false, false, SYM, RMP2_OBS_NONE, kPlainRollout); break;                          \
    }                                                                                                                   \
  } while (0)
  const bool lean_rollout = !plain && !out.M && !out.f && !o.capsule && o.mode != RMP2_OBS_EXPLICIT_PAIRS && !latency && !with_records;
  if (lean_rollout) {
    if (minw == 3) minw = 2;  // (the lean rollout exists at two and four waves per SIMD)
    if (minw == 4) {
      if (symk) RMP2_QUAD_ROLL_SYM(4, (N == 9)); else RMP2_QUAD_ROLL_SYM(4, false);
    } else {
      if (symk) RMP2_QUAD_ROLL_SYM(2, (N == 9)); else RMP2_QUAD_ROLL_SYM(2, false);
    }
    return;
  }
#undef RMP2_QUAD_ROLL_SYM
  if constexpr (N == 9) {
    // link geometry of the distance leaves over a SHARED table, plain control step, throughput grid: the lean builds (OBS =
    // kObsSharedLink: the link's world segment formed in the walk, 8 floats per leaf-bearing frame -- no rotation records, no
    // rollout loop, no debug outputs): three waves per SIMD by registers, ten waves per CU by LDS for the Panda
    // (sets without an inertia leaf keep the general build: it carries the rank-one pull-back, rmp2_quad.h kRank1)
    if (o.link_caps && !h->has_point && !h->likely_singular && h->link_rows_ok && o.mode == RMP2_OBS_SHARED_SPHERES && !latency && ro.n_iters == 1 &&
        ro.substeps == 0 && !ro.q_out && !out.M && !out.f) {
      bytes = lds_bytes + sizeof(float) * kLinkSeg * kRobotsPerWave * h->n_leaf_ops;
      const int lw = h->quad_minw == 2 ? 2 : 3;
#define RMP2_QUAD_LINK(MINW, CAP, SYM)                                                                                   \
      do {                                                                                                              \
        auto kern = rmp2_step_quad_kernel<N, SLOTS, MINW, false, CAP, SYM, kObsSharedLink, kPlainStep>;                  \
        if (bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); \
        RMP2_STEP_LAUNCH(h, kern, dim3(blocks), dim3(kWave), bytes, s, h->d_prog, hdr, q, qd, goal, gs, o, out, ro, R);  \
      } while (0)
#define RMP2_QUAD_LINK_SYM(MINW, CAP)                                                                                    \
      do {                                                                                                              \
        if (symk) RMP2_QUAD_LINK(MINW, CAP, true); else RMP2_QUAD_LINK(MINW, CAP, false);                               \
      } while (0)
      if (o.capsule) {
        if (lw == 2) RMP2_QUAD_LINK_SYM(2, true); else RMP2_QUAD_LINK_SYM(3, true);
      } else {
        if (lw == 2) RMP2_QUAD_LINK_SYM(2, false); else RMP2_QUAD_LINK_SYM(3, false);
      }
#undef RMP2_QUAD_LINK_SYM
#undef RMP2_QUAD_LINK
      h->last_kernel = quad_certifies_strict(h)
                           ? "rmp2_step_quad_kernel (4 lanes per robot; link geometry, lean build: segments formed in the walk; strict: full rank "
                             "certified per robot, Jacobi pseudo-inverse for the rest)"
                           : "rmp2_step_quad_kernel (4 lanes per robot; link geometry, lean build: segments formed in the walk)";
      return;
    }
  }
  if constexpr (N == 9) {
    // interface B (explicit closest-point pairs), plain control step: the STREAMED form (rmp2_quad.h kObsExplicitStream) -- the pair
    // arrays by LDS-DMA a quarter leaf at a time beside a four-wave working set (128 registers / 9.6 KB per wave: sixteen waves per
    // CU).  Needs: 32 pairs per distance leaf and 16-byte aligned leaf segments (the DMA moves 16-byte pieces), at most
    // kStreamMaxFrames leaf-bearing frames, frame slots that hold the compact [p v a] image and the chunk buffer, an inertia leaf (the
    // rank-one pull-back lives in the general builds).  RMP2_EXPLICIT_STREAM=0 | 1 forces the single-loop two-wave form / this one.
    const int stream_env = h->explicit_stream;
    bool stream = o.mode == RMP2_OBS_EXPLICIT_PAIRS && !latency && !with_records && plain && !o.capsule && !o.dist &&
                  !h->likely_singular && stream_env == 1 &&
                  h->n_leaf_ops <= kStreamMaxFrames && kSlot * kRobotsPerWave * quad_slots(h->n_ops_step) >= kStreamPva + kQdmaBuf &&
                  (o.n_pairs % 4) == 0 && (reinterpret_cast<uintptr_t>(o.p_link) % 16) == 0 && (reinterpret_cast<uintptr_t>(o.p_obs) % 16) == 0;
    for (int l : h->distance_leaves)
      stream = stream && (h->h_pair_begin[l] % 4) == 0 && h->h_pair_begin[l + 1] - h->h_pair_begin[l] == 32;
    if (stream) {
      if (symk) RMP2_STEP_LAUNCH(h, (rmp2_step_quad_kernel<N, SLOTS, 4, false, false, true, kObsExplicitStream, kPlainStep>), dim3(blocks),
                                 dim3(kWave), lds_bytes, s, h->d_prog, hdr, q, qd, goal, gs, o, out, ro, R);
      else RMP2_STEP_LAUNCH(h, (rmp2_step_quad_kernel<N, SLOTS, 4, false, false, false, kObsExplicitStream, kPlainStep>), dim3(blocks),
                            dim3(kWave), lds_bytes, s, h->d_prog, hdr, q, qd, goal, gs, o, out, ro, R);
      h->last_kernel = quad_certifies_strict(h)
                           ? "rmp2_step_quad_kernel (4 lanes per robot; explicit pairs streamed by LDS-DMA, pair phase before the pull-back; "
                             "strict: full rank certified per robot, Jacobi pseudo-inverse for the rest)"
                           : "rmp2_step_quad_kernel (4 lanes per robot; explicit pairs streamed by LDS-DMA, pair phase before the pull-back)";
      return;
    }
  }
  if (with_records) {
    // attached-point leaves (TaskmapRelative4x4 + CollisionAvoidance) and link geometry in the table modes: the general
    // flavour with the extra per-frame records (27 floats per frame instead of 12: two waves per SIMD at most)
    auto raise = [&](const void* kern) {
      if (bytes > 64 * 1024) (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    };
#define RMP2_QUAD_PT(MINW, STAGE, CAP, SYM)                                                                             \
    do {                                                                                                                \
      raise(reinterpret_cast<const void*>(rmp2_step_quad_kernel<N, SLOTS, MINW, STAGE, CAP, SYM, kObsAny, kGeneral, true>)); \
      RMP2_STEP_LAUNCH(h, (rmp2_step_quad_kernel<N, SLOTS, MINW, STAGE, CAP, SYM, kObsAny, kGeneral, true>), dim3(blocks), \
                       dim3(kWave), bytes, s, h->d_prog, hdr, q, qd, goal, gs, o, out, ro, R);                          \
    } while (0)
#define RMP2_QUAD_PT_SYM(MINW, STAGE, CAP)                                                                              \
    do {                                                                                                                \
      if (symk) RMP2_QUAD_PT(MINW, STAGE, CAP, (N == 9)); else RMP2_QUAD_PT(MINW, STAGE, CAP, false);                   \
    } while (0)
    if (o.capsule) {  // (capsule tables reach here with link geometry only: attached-point leaves take explicit pairs)
      if (latency) RMP2_QUAD_PT_SYM(1, true, true); else RMP2_QUAD_PT_SYM(2, false, true);
    } else {
      if (latency) RMP2_QUAD_PT_SYM(1, true, false); else RMP2_QUAD_PT_SYM(2, false, false);
    }
#undef RMP2_QUAD_PT_SYM
#undef RMP2_QUAD_PT
    return;
  }
  if (latency) RMP2_QUAD_BY_CAP(1, true);
  else if (plain && minw == 4) RMP2_QUAD_PLAIN(4);
  else if (plain && minw == 3) RMP2_QUAD_PLAIN(3);
  else if (plain) RMP2_QUAD_PLAIN(2);
  else if (minw == 4) RMP2_QUAD_BY_CAP(4, false);  // 128 registers, four waves per SIMD
  else if (minw == 3) RMP2_QUAD_BY_CAP(3, false);  // 168 registers, three waves per SIMD
  else RMP2_QUAD_BY_CAP(2, false);
#undef RMP2_QUAD_PLAIN
#undef RMP2_QUAD_PLAIN_SYM
#undef RMP2_QUAD_BY_CAP
#undef RMP2_QUAD_LAUNCH
}

}  // namespace

#define RMP2_CAT_(a, b, c, d) a##b##c##d
#define RMP2_CAT(a, b, c, d) RMP2_CAT_(a, b, c, d)
bool RMP2_CAT(launch_quad_n, RMP2_TU_N, _s, RMP2_TU_SLOTS)(const rmp2_handle* h, const float* q, const float* qd, const float* goal, int gs,
                                                            const ObsArgs& o, const OutArgs& out, const RolloutArgs& ro, int R, hipStream_t s) {
  launch_quad<RMP2_TU_N, RMP2_TU_SLOTS>(h, q, qd, goal, gs, o, out, ro, R, s);
  return true;
}

}  // namespace rmp2
