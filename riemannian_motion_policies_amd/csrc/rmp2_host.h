// rmp2_host.h -- host-side pieces shared by the translation units of librmp2_hip.so: the engine handle, the step
// launch macro, and the launcher entry points of the per-mapping translation units (rmp2_quad_tu.hip, rmp2_hex_tu.hip).
// The kernels are templates in headers; each mapping is INSTANTIATED in its own translation unit so that the library
// builds in parallel (the quad kernel alone has ~70 instantiations) and a change to one mapping recompiles only it.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "rmp2_device.h"
#include "rmp2_solve.h"
#include "rmp2_quad.h"
#include "rmp2_hex.h"

struct rmp2_handle {
  int device = 0;
  int n_dof = 0, n_frames = 0, n_slots = 0, n_leaves = 0, goal_floats = 0;
  int n_slots_full = 0;  // slots of the unpruned program (FK / differentiate entry points)
  int n_ops_step = 0;    // frames the control-step kernels visit (pruned + folded program)
  rmp2::DevProgram* d_prog_full = nullptr;
  int n_template = 0;  // N of the kernel instantiation
  bool has_distance = false;
  bool has_point = false;  // attached-point leaves (CollisionAvoidance): hex and lane-per-robot kernels
  int n_id_leaves = 0;
  int n_leaf_ops = 0;
  uint32_t rev_mask = 0;
  float cull_c0 = 0.f;  // max over the distance leaves of (metric_modulation_radius + margin): beyond it a pair is culled
  uint32_t dof_ops[3] = {0u, 0u, 0u};  // op that owns each dof (quad kernel: Jacobian columns come from the frame slots)
  bool strict = false;  // solve_mode == RMP2_SOLVE_PINV
  bool strict_certify = true;  // RMP2_STRICT_CERTIFY=0: the Jacobi pseudo-inverse on every robot (A/B)
  bool link_rows_ok = false;   // <= 1 distance leaf per frame: link geometry may take the lean builds (segments formed in the walk)
  bool explicit_glds = false;  // RMP2_EXPLICIT_GLDS=1: explicit pairs streamed half a leaf ahead by LDS-DMA (measured: no gain)
  int stream_stagger = 0;    // RMP2_STREAM_STAGGER=n: start offset between the four waves of a SIMD in the streamed explicit-pair step, units of 3.4 us
  int explicit_stream = -1;  // RMP2_EXPLICIT_STREAM=0|1: never / at any throughput grid take the streamed explicit-pair step (A/B; -1: by grid size)
  bool likely_singular = false;  // no positive-definite identity leaf in the set
  int kernel_choice = 0;  // 0 auto, 1 lane-per-robot, 2 quad-per-robot, 3 hex (env RMP2_KERNEL=lane|quad|hex, A/B only)
  int hex_levels = 0;
  int hex_waves = 4;  // waves per block of the hex kernel (env RMP2_HEX_WAVES=1|4, A/B only)
  int quad_minw = 0;  // register cap of the throughput quad build: 0 = by fleet size (2, 3 or 4 waves per SIMD, launch_quad);
                      // env RMP2_QUAD_MINW=2|3|4 pins it (A/B only)
  int n_simd = 1024;  // SIMDs of the device (4 per CU)
  void* step_fence = nullptr;  // rmp2_set_step_fence: completion fence of the step launches (nullptr: none)
  bool symmetric = false;      // no leaf with a non-symmetric metric (JointLimitAvoidance, quirk Q2) in the set
  int prio_tail = -1;          // env RMP2_PRIO_TAIL=0..3 pins the priority of the phases after the frame loop (A/B only)
  bool quad_latency_set = false;   // RMP2_QUAD_LATENCY_BLOCKS given (tuning builds)
  int quad_latency_blocks = 0;     // grids up to this many waves take the latency build: none since round 5 (tuning builds: env RMP2_QUAD_LATENCY_BLOCKS)
  int n_fk_leaves = 0;
  int hex_is_chain = 0;
  void* d_hex_blob = nullptr;  // the staged program of the hex kernel, laid out exactly as it sits in LDS
  int hex_blob16 = 0;          // its size in 16-byte units
  std::vector<int> distance_leaves;
  rmp2::DevProgram* d_prog = nullptr;
  int32_t* d_pair_begin = nullptr;
  int32_t h_pair_begin[RMP2_MAX_LEAVES + 1];
  bool pair_begin_valid = false;
  float* d_scratch = nullptr;  // rmp2_differentiate scratch
  size_t scratch_robots = 0;
  float* d_pairs = nullptr;    // p_link | p_obs of the closest-point stage when a step with link geometry runs as stage + explicit-pair step
  size_t pairs_floats = 0;     // (floats per array)
  double* d_system = nullptr;  // [robots][n_dof * (n_dof + 1)] combined metric and force between the quad step and rmp2_pinv_kernel
  size_t system_robots = 0;
  mutable bool quad_skip_resolve = false;  // set around that quad launch (dispatch_solve)
  mutable const char* last_kernel = "none";  // mapping the last control step / rollout was launched with (rmp2_last_kernel)
  std::string error;
};

// A step launch: plain, or -- when a completion fence is attached to the handle (rmp2_set_step_fence) -- with the
// fence as the dispatch's own completion signal (hipExtLaunchKernelGGL stop event): no separate packet behind the kernel.
#define RMP2_STEP_LAUNCH(h_, kern_, grid_, block_, bytes_, stream_, ...)                                            \
  do {                                                                                                              \
    if ((h_)->step_fence)                                                                                           \
      hipExtLaunchKernelGGL(kern_, grid_, block_, bytes_, stream_, nullptr, static_cast<hipEvent_t>((h_)->step_fence), 0, \
                            __VA_ARGS__);                                                                           \
    else                                                                                                            \
      hipLaunchKernelGGL(kern_, grid_, block_, bytes_, stream_, __VA_ARGS__);                                       \
  } while (0)

namespace rmp2 {
// rmp2_quad_tu.hip: one object per template size N and save/restore slot count of the program
#define RMP2_DECL_QUAD(NAME)                                                                                            \
  bool NAME(const rmp2_handle* h, const float* q, const float* qd, const float* goal, int gs, const ObsArgs& o,         \
            const OutArgs& out, const RolloutArgs& ro, int R, hipStream_t s)
RMP2_DECL_QUAD(launch_quad_n2_s0);
RMP2_DECL_QUAD(launch_quad_n2_s1);
RMP2_DECL_QUAD(launch_quad_n2_s2);
RMP2_DECL_QUAD(launch_quad_n9_s0);
RMP2_DECL_QUAD(launch_quad_n9_s1);
RMP2_DECL_QUAD(launch_quad_n9_s2);
#undef RMP2_DECL_QUAD
// rmp2_quad_pair_tu.hip: the control steps of two engines as ONE grid (a 2-dof and a 3..9-dof robot type on shared / ragged
// sphere tables: a mixed fleet shard); false = no fused instantiation for this pair, nothing was launched
bool launch_quad_pair(const rmp2_handle* ha, const float* qa, const float* qda, const float* goala, int gsa, const ObsArgs& oa,
                      const OutArgs& outa, int Ra, const rmp2_handle* hb, const float* qb, const float* qdb, const float* goalb,
                      int gsb, const ObsArgs& ob, const OutArgs& outb, int Rb, hipStream_t s);
// solve = PINV handles whose strict step is ONE quad launch: an inertia leaf (else every robot is rank deficient and would take
// the careful pass), <= 16 goal floats; symmetric sets are certified from the LDL^T of the elimination, the others from U and
// the largest multiplier (rmp2_quad.h)
inline bool quad_certifies_strict(const rmp2_handle* h) {
  return h->strict && h->strict_certify && !h->likely_singular && h->n_template == 9 && h->goal_floats <= 16;
}
// ... and in the hex mapping (any template size: its Gauss-Jordan keeps the pivot rows for the same certificate)
inline bool hex_certifies_strict(const rmp2_handle* h) { return h->strict && h->strict_certify && !h->likely_singular; }
inline QuadHdr make_quad_hdr(const rmp2_handle* h) {
  return QuadHdr{h->n_ops_step, h->n_dof, h->n_id_leaves, h->n_leaves, h->goal_floats, h->n_leaf_ops, h->rev_mask,
                 h->hex_levels, h->n_fk_leaves, h->hex_is_chain, {h->dof_ops[0], h->dof_ops[1], h->dof_ops[2]}, h->cull_c0, h->strict ? 1 : 0,
                 h->prio_tail >= 0 ? h->prio_tail : 0, h->quad_skip_resolve ? 1 : 0, h->has_point ? 1 : 0, h->likely_singular ? 1 : 0,
                 h->stream_stagger};
}
// rmp2_hex_tu.hip (false: the working set does not fit the CU's LDS -- the caller falls back to the quad mapping)
bool launch_hex_n2(const rmp2_handle* h, const float* q, const float* qd, const float* goal, int gs, const ObsArgs& o,
                   const OutArgs& out, const RolloutArgs& ro, int R, hipStream_t s);
bool launch_hex_n9(const rmp2_handle* h, const float* q, const float* qd, const float* goal, int gs, const ObsArgs& o,
                   const OutArgs& out, const RolloutArgs& ro, int R, hipStream_t s);
bool launch_hex_n16(const rmp2_handle* h, const float* q, const float* qd, const float* goal, int gs, const ObsArgs& o,
                    const OutArgs& out, const RolloutArgs& ro, int R, hipStream_t s);
}  // namespace rmp2
