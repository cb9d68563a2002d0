/*
 * rmp2.h -- C ABI of the MI355X RMP2 evaluation-and-pullback engine (librmp2_hip.so).
 *
 * The reference (TomGoesGitHub/Riemannian-Motion-Policies) has no FFI: its seam is the
 * Python protocol  RmpCore.evaluate(q, qd) -> qdd  (rmp.py:133-155).  This header is the
 * C boundary that replaces everything below that call -- task-map differentiation
 * (taskmap.py:13-168, kinematics.py:212-270, helper/rmp_helper.py:3-60), leaf evaluation
 * (rmp2.py:31-226, rmp.py:226-382), the pull-back and sum (rmp.py:157-180, :142-150) and the
 * resolve step (rmp.py:153-154) -- for a BATCH of R independent robots of one type.
 *
 * Rules of the boundary
 *   - plain C, plain pointers and sizes; no C++/torch types.
 *   - the caller owns every buffer; the engine owns its handle, its constant tables and
 *     nothing else.  rmp2_step() allocates nothing and never synchronises the host -- one exception: a handle
 *     whose every robot is resolved by the pseudo-inverse (solve_mode = PINV, or a set without a positive-definite
 *     identity-map leaf) on a 3..9-dof robot keeps the combined systems of the fleet between its two kernels
 *     (8 n (n + 1) bytes per robot) and grows that buffer, synchronising, the first time a larger fleet is stepped.
 *   - all array arguments of rmp2_step / rmp2_forward_kinematics / rmp2_differentiate are
 *     DEVICE pointers (HBM), row-major, robot index slowest: q[R][n_dof] etc.
 *   - every call returns 0 on success or a negative RMP2_ERR_* code; the message is
 *     available from rmp2_last_error().
 *   - a handle is bound to one device; calls on one handle must not race (thread-compatible).
 *     rmp2_create and the launching calls make that device the calling thread's current HIP device
 *     (as hipSetDevice would) and leave it so; rmp2_destroy restores the caller's.
 *   - there is NO host/CPU execution path in this library.  If no HIP device is usable,
 *     rmp2_create() fails with RMP2_ERR_NO_DEVICE.
 */
#ifndef RMP2_H
#define RMP2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RMP2_ABI_VERSION 4 /* 4: rmp2_reserve, rmp2_exchange_nranks, RMP2_STATUS_JACOBI; attached-point leaves take a table + link_capsules */

#define RMP2_MAX_FRAMES 32  /* frames (= URDF joints) per robot type                    */
#define RMP2_MAX_DOF 16     /* actuated joints per robot type                            */
#define RMP2_MAX_LEAVES 48  /* leaf RMPs per set                                          */
#define RMP2_MAX_PARAMS 12  /* scalar parameters per leaf                                 */

/* ---- error codes --------------------------------------------------------------------- */
#define RMP2_OK 0
#define RMP2_ERR_INVALID_ARGUMENT (-1)
#define RMP2_ERR_UNSUPPORTED (-2) /* no kernel for this combination (e.g. > 2 open branch points; solve = PINV,
                                     sets without an inertia leaf or attached-point leaves beyond 9 dofs) */
#define RMP2_ERR_NO_DEVICE (-3)
#define RMP2_ERR_HIP (-4)         /* a HIP runtime call failed; see rmp2_last_error()      */
#define RMP2_ERR_ABI_MISMATCH (-5)

/* ---- robot table (output of the URDF "kinematic compiler") ---------------------------
 * Replaces the tensors built by UrdfForwardKinematic._build (kinematics.py:163-209):
 * kinematic_chains -> parent[], _q_reordering -> q_index[], T_constant, axis,
 * is_revolute/is_prismatic/is_fixed -> joint_type[].  Frame order = reference frame order
 * (breadth-first, helper/urdf_parsing.py:74-97). */
#define RMP2_JOINT_FIXED 0
#define RMP2_JOINT_REVOLUTE 1
#define RMP2_JOINT_PRISMATIC 2

typedef struct rmp2_robot {
  int32_t n_frames;
  int32_t n_dof;
  int32_t parent[RMP2_MAX_FRAMES];     /* parent frame, -1 = attached to the base link      */
  int32_t joint_type[RMP2_MAX_FRAMES]; /* RMP2_JOINT_*                                      */
  int32_t q_index[RMP2_MAX_FRAMES];    /* index into q, -1 = evaluated at q = 0             */
  float axis[RMP2_MAX_FRAMES][3];      /* joint axis in the joint frame                     */
  float T_const[RMP2_MAX_FRAMES][12];  /* rows 0..2 of the 4x4 T_constant, row-major        */
} rmp2_robot;

/* ---- leaf policies -------------------------------------------------------------------
 * kind: which (xdd_des, A) formula; each cites the reference class it reproduces,
 * quirks included.  params[] order is the reference
 * constructor's keyword order. */
#define RMP2_LEAF_TARGET_ATTRACTOR 1 /* rmp2.py:31-83   params: accel_p_gain, accel_d_gain, accel_norm_eps,
                                        metric_alpha_length_scale, min_metric_alpha, max_metric_scalar,
                                        min_metric_scalar, proximity_metric_boost_scalar,
                                        proximity_metric_boost_length_scale ; goal: 3 floats   */
#define RMP2_LEAF_JOINT_VELOCITY_CAP 2 /* rmp2.py:86-112  params: max_velocity, velocity_damping_region,
                                        damping_gain, metric_weight                          */
#define RMP2_LEAF_JOINT_DAMPING 3    /* rmp2.py:115-137 params: accel_d_gain, metric_scalar, inertia */
#define RMP2_LEAF_OBSTACLE_AVOIDANCE 4 /* rmp2.py:140-196 params: margin, damping_gain, damping_std_dev,
                                        damping_robustness_eps, damping_velocity_gate_length_scale,
                                        repulsion_gain, repulsion_std_dev, metric_modulation_radius,
                                        metric_scalar, metric_exploder_std_dev, metric_exploder_eps */
#define RMP2_LEAF_CSPACE_BIASING 5   /* rmp2.py:198-226 params: metric_scalar, position_gain, damping_gain,
                                        robust_position_term_thresh, inertia ; vec_a = goal q0 */
#define RMP2_LEAF_TARGET_POLICY 6    /* rmp.py:226-260  params: alpha, beta, c ; goal: k floats
                                        (k = 3 on an FK position map, n_dof on the identity map) */
#define RMP2_LEAF_JOINT_LIMIT_AVOIDANCE 7 /* rmp.py:349-382 params: gamma_p, gamma_d ;
                                        vec_a = lower limits, vec_b = upper limits             */
#define RMP2_LEAF_CONFIG_SPACE_BIASING 8 /* rmp.py:318-347 params: gamma_p, gamma_d, w ; vec_a = q0 */
#define RMP2_LEAF_COLLISION_AVOIDANCE 9 /* rmp.py:264-315 params: eta_rep, nu_rep, eta_damp, nu_damp, r, c ;
                                        per-pair data d, n (rmp2_obstacles.dist / p_obs), FK_POINT map only.
                                        beta = 0 in the reference (rmp.py:311) => metric w(d) * I          */

/* task map of a leaf (the chains the reference's experiments build with chain_taskmaps,
 * taskmap.py:142-168) */
#define RMP2_TASKMAP_IDENTITY 0    /* IdentityTaskmap                  taskmap.py:13-20        */
#define RMP2_TASKMAP_FK_POSITION 1 /* FK(frame) -> 4x4 -> position     taskmap.py:22-31,45-54  */
#define RMP2_TASKMAP_FK_DISTANCE 2 /* FK(frame) -> 4x4 -> distance     taskmap.py:22-31,115-138 */
#define RMP2_TASKMAP_FK_POINT 3    /* FK(frame) -> relative 4x4 -> position   taskmap.py:22-31,79-99,45-54:
                                      a point rigidly attached to the frame, one per pair (position given in the
                                      JOINT frame, rmp2_obstacles.p_link); lever arm included, unlike FK_DISTANCE */

typedef struct rmp2_leaf {
  int32_t kind;        /* RMP2_LEAF_*                                                     */
  int32_t taskmap;     /* RMP2_TASKMAP_*                                                  */
  int32_t frame;       /* frame index for FK task maps, else -1                           */
  int32_t goal_offset; /* float offset of this leaf's goal inside one robot's goal row, -1 = none */
  float params[RMP2_MAX_PARAMS];
  float vec_a[RMP2_MAX_DOF];
  float vec_b[RMP2_MAX_DOF];
} rmp2_leaf;

/* ---- resolve step ------------------------------------------------------------------ */
#define RMP2_SOLVE_AUTO 0 /* fp64 LU with threshold pivoting; robots whose metric is (numerically)
                             singular fall through to the PINV path.  Same result as PINV to fp64
                             rounding whenever M is well conditioned.                         */
#define RMP2_SOLVE_PINV 1 /* reference-faithful: qdd = pinv(M) f, the fp64 Moore-Penrose pseudo-inverse with
                             TensorFlow's cutoff 10*n*eps*sigma_max (rmp.py:153).  Where the elimination can CERTIFY
                             that every singular value of a robot's M lies above the cutoff (sets with an inertia
                             leaf: a bound on |M^-1| from its own triangular factor), pinv(M) = inv(M) and the
                             elimination's result stands; every other robot -- and every robot of a set without
                             an inertia leaf -- is resolved by a one-sided Jacobi SVD (status RMP2_STATUS_JACOBI /
                             RANK_DROP).  Same numbers either way to fp64 rounding; the certifying step costs
                             what the AUTO step costs.                                                      */

typedef struct rmp2_desc {
  int32_t abi_version; /* must be RMP2_ABI_VERSION */
  int32_t solve_mode;  /* RMP2_SOLVE_* */
  int32_t n_leaves;
  int32_t goal_floats; /* floats per robot in the goal array (sum over goal-bearing leaves) */
  rmp2_robot robot;
  rmp2_leaf leaves[RMP2_MAX_LEAVES];
} rmp2_desc;

/* ---- per-step obstacle input (data of the distance task maps) -----------------------
 * The reference feeds closest-point pairs through Datamanager tf.Variables
 * (data_management.py:8-17, taskmap.py:115-138).  Two array-backed forms:
 *   EXPLICIT_PAIRS (reference-faithful): for every FK_DISTANCE leaf l a block of pairs
 *     p_link[r][pair_begin[l] .. pair_begin[l+1])[3], p_obs[...] in the robot base frame.
 *     value d = |p_link - p_obs|; derivative w.r.t. the FRAME ORIGIN only (quirk Q5).
 *     For an FK_POINT leaf (CollisionAvoidance) the same pair range carries the Datamanager's
 *     other three fields (data_management.py:14-16): p_link = 'relative_position' in the joint
 *     frame, p_obs = 'normal_vec' in the base frame, dist[r][pair] = 'distance'.
 *   SHARED_SPHERES: one table spheres[K][4] = (cx, cy, cz, radius) for the whole fleet;
 *     every FK_DISTANCE leaf sees K pairs: control point = its frame origin,
 *     d = |origin - c| - radius, direction (origin - c)/|origin - c|.
 *   RAGGED_SPHERES: as SHARED_SPHERES but robot r only sees the spheres
 *     csr_index[csr_offset[r] .. csr_offset[r+1]) .
 * Both table modes take either primitive:
 *   PRIM_SPHERE  record = 4 floats (cx, cy, cz, radius)
 *   PRIM_CAPSULE record = 8 floats (ax, ay, az, radius, bx, by, bz, unused): a segment a-b swept by a
 *     sphere (the reference's cylinder obstacles, simulation.py:245-261, :495-500).  The engine forms the
 *     closest point of the segment to the control point, c* = a + clamp((p-a).(b-a)/|b-a|^2, 0, 1)(b-a),
 *     and proceeds as for a sphere centred at c*: this IS the closest-point preprocessing the reference
 *     runs on the CPU before every step (simulation.py:462-484 calculate_distances -> Datamanager),
 *     fused into the step.  rmp2_closest_points() below materialises the same pairs as arrays.
 */
#define RMP2_OBS_NONE 0
#define RMP2_OBS_EXPLICIT_PAIRS 1
#define RMP2_OBS_SHARED_SPHERES 2
#define RMP2_OBS_RAGGED_SPHERES 3

#define RMP2_PRIM_SPHERE 0
#define RMP2_PRIM_CAPSULE 1
#define RMP2_PRIM_CYLINDER 2 /* record = 8 floats (cx, cy, cz, radius, ux, uy, uz, half_height): a finite cylinder with FLAT caps, centre c, unit
                              * axis u -- the reference's own obstacle primitive (simulation.py:245-261: pybullet.GEOM_CYLINDER).  Table modes:
                              * x = signed distance of the control point to the cylinder's surface, direction = its outward normal there
                              * (side, cap or rim).  rmp2_closest_points[_links]: the nearest points of the link (its capsule, or the frame
                              * origin) and the cylinder's surface.  Not with link_capsules fused into the step (that closed form is an
                              * iteration: run the stage and feed EXPLICIT_PAIRS). */

typedef struct rmp2_obstacles {
  int32_t mode;
  int32_t n_spheres;                       /* K = records in the primitive table             */
  int32_t n_pairs;                         /* P = pairs per robot, EXPLICIT_PAIRS            */
  int32_t primitive;                       /* RMP2_PRIM_* (table modes)                      */
  int32_t pair_begin[RMP2_MAX_LEAVES + 1]; /* indexed by LEAF index; non-distance leaves: empty range */
  const float *spheres;                    /* device [K][4] spheres or [K][8] capsules       */
  const float *p_link;                     /* device [R][P][3]                               */
  const float *p_obs;                      /* device [R][P][3]                               */
  const int32_t *csr_offset;               /* device [R+1]                                   */
  const int32_t *csr_index;                /* device [csr_offset[R]]; non-null even when every list is empty */
  const float *dist;                       /* device [R][P], FK_POINT leaves only (else NULL) */
  const float *link_capsules;              /* device [n_distance_leaves][8] = (a, radius, b, -) in each distance leaf's FRAME
                                              coordinates (leaf order), or NULL.  Table modes: the control point of a pair is
                                              the nearest point of the LINK's capsule to the obstacle -- the values of
                                              rmp2_closest_points_links + EXPLICIT_PAIRS (d = |p_link - p_obs| and unit normal; as
                                              there, the derivative moves the point with the frame origin, taskmap.py:124-129).
                                              FUSED into the step for tables of at most 256 spheres / capsules, robots with at
                                              most 9 dofs and an inertia leaf, solve = AUTO or a certifying PINV, and 2-dof robots
                                              with either resolve (their closed-form 2 x 2 resolve is the pseudo-inverse) -- shared
                                              tables, ragged lists, rollouts; a plain step beyond those limits -- more dofs, the
                                              all-Jacobi PINV, bigger tables, CYLINDER tables -- runs as the stage into a buffer of
                                              the handle followed by the explicit-pair step (two launches, same numbers as calling
                                              the two entry points).  RAGGED lists take that route with one more launch (one pair per
                                              list entry, a repeated index counted twice as in the fused form, filler pairs 1e9 m
                                              away up to the fleet's longest list); the list lengths are read back from csr_offset,
                                              so that form synchronises the stream and is refused inside a stream capture.
                                              Rollouts beyond the fused limits, and sets with attached-point leaves over ragged
                                              lists: RMP2_ERR_UNSUPPORTED. */
} rmp2_obstacles;

/* ---- outputs ----------------------------------------------------------------------- */
#define RMP2_STATUS_NONFINITE 1u /* qdd contains NaN/Inf (e.g. JointVelocityCap pole, quirk Q4).  A robot fed a NaN / Inf in q or qd
                                  * resolves to NaN on EVERY joint with this bit: the non-finite dof's force is made non-finite by
                                  * construction, so that neither the culling (an out-of-range pair -- metric 0, acceleration NaN in
                                  * the reference: 0 * NaN -- is never evaluated here) nor a set that never reads the joint can return
                                  * a finite answer for it.  This is the reference's result wherever its own arithmetic carries the
                                  * value into the system (tf.linalg.pinv of a non-finite system is NaN), and stricter where it does
                                  * not (a joint that moves no leaf frame in a set without identity-map leaves). */
#define RMP2_STATUS_RANK_DROP 2u /* the pseudo-inverse dropped at least one singular value  */
#define RMP2_STATUS_PINV_PATH 4u /* AUTO mode: this robot was resolved on the PINV path      */
#define RMP2_STATUS_JACOBI 8u    /* PINV mode, certifying step (symmetric sets with an inertia leaf): the elimination could not
                                  * certify this robot's metric as full rank above TensorFlow's cutoff, so it was resolved by the
                                  * Jacobi pseudo-inverse instead of by the elimination (same result where both apply; diagnostic) */

typedef struct rmp2_outputs {
  float *qdd;       /* device [R][n_dof]                      required                     */
  uint32_t *status; /* device [R]                             optional (NULL)              */
  double *M;        /* device [R][n_dof][n_dof] combined metric,   optional (NULL)         */
  double *f;        /* device [R][n_dof]        combined force,    optional (NULL)         */
} rmp2_outputs;

typedef struct rmp2_handle rmp2_handle;

/* Library/ABI identification (bindings check these before anything else). */
int rmp2_abi_version(void);
size_t rmp2_sizeof_desc(void);
size_t rmp2_sizeof_obstacles(void);

/* The reference's LEAF protocol on its own:  rmp.evaluate(x, xd) -> (xdd_des, A)  (rmp2.py:25-29, rmp.py:202-206) for a
 * batch of B task-space points, outside any RmpCore (rmp2_step evaluates the leaves inside the fused control step and
 * never calls this).  `leaf`: kind + params (+ vec_a / vec_b) as in rmp2_desc; taskmap / frame / goal_offset ignored.
 *   k           task-space dimension: 3 (TargetAttractor, CollisionAvoidance), 1 (ObstacleAvoidance), 1 .. RMP2_MAX_DOF
 *               for the identity-map leaves and TargetPolicy
 *   x, xd       device [B][k];   goal  device [k] (TargetAttractor, TargetPolicy) else NULL
 *   dist, nvec  device [B], [B][3]: CollisionAvoidance's data-fed distance / normal (else NULL)
 *   xdd, A      device [B][k], [B][k][k] (row major)
 * TargetPolicy's norms are global in the reference (rmp.py:243: B = 1 there); here every row is its own evaluation. */
int rmp2_leaf_evaluate(int device, const rmp2_leaf *leaf, int32_t k, const float *x, const float *xd, const float *goal,
                       const float *dist, const float *nvec, float *xdd, float *A, int32_t B, void *stream);

/* Dry run of rmp2_create's host-side program compiler (descriptor validation, depth-first schedule with save / restore
 * slots, pruning and folding of leaf-less fixed frames, ancestor / dof tables): RMP2_OK or the error rmp2_create would
 * return for this descriptor, message via rmp2_last_error(NULL).  Needs no HIP device -- it is what a host-side tool
 * (and the sanitizer build, tools/asan_compile_program.sh) can exercise without a GPU. */
int rmp2_validate(const rmp2_desc *desc);

/* Environment variables read by rmp2_create (and by nothing else).  They are DIAGNOSTIC overrides of the per-call kernel
 * dispatch, used by the parity tests and the profiling tools to force every mapping over the same inputs; all choices
 * produce the same numbers to fp32 rounding, none is needed for correctness, and an unset variable means "by fleet size":
 *   RMP2_KERNEL     = hex | quad | lane   mapping of robots to lanes (16 / 4 / 1 lanes per robot);
 *                                         lane also selects the lane-per-robot form of the closest-point stage
 *   RMP2_QUAD_MINW  = 2 | 3 | 4           register cap of the quad mapping's throughput build (waves per SIMD it leaves room for)
 *   RMP2_QUAD_SYM   = 0                   general (full-matrix) form of the quad mapping for sets that qualify for the symmetric one
 *   RMP2_EXCHANGE_THROTTLE_US = n         (rmp2_exchange_create) bound of the host throttle of rmp2_exchange_step, 0 = free-running
 *   RMP2_EXCHANGE_BUFFERS = 2 | 3         (rmp2_exchange_create) table buffers in rotation: 3 (default), or depth + 1
 *   RMP2_EXPLICIT_GLDS = 1                (rmp2_create) EXPLICIT_PAIRS: the pair arrays streamed half a leaf ahead by LDS-DMA with per-quad
 *                                         compaction of the in-range pairs (built and parity-tested in round 4; measured no faster than
 *                                         the register loads -- the mode is bound by the bytes a CU can keep in flight beside the frame
 *                                         records in LDS --, so it is off by default)
 *   RMP2_EXPLICIT_STREAM = 1              (rmp2_create) EXPLICIT_PAIRS, plain control step: the streamed form (pair phase of all leaf
 *                                         frames before the pull-back, pair arrays by LDS-DMA through the frame records' LDS, four
 *                                         waves per SIMD) -- built, parity-tested and measured in round 5: not faster while the whole
 *                                         fleet is one round of waves, so it is off by default
 *   RMP2_STREAM_STAGGER = n               (rmp2_create) with it: start offset between the four waves of a SIMD (units of 3.4 us; A/B)
 *   RMP2_STRICT_CERTIFY = 0               (rmp2_create) solve = PINV: the Jacobi pseudo-inverse on EVERY robot (two kernels) instead of
 *                                         the certifying one-launch step -- the A/B the equality test of the two is built on
 * Further A/B knobs (RMP2_PRIO_TAIL, RMP2_HEX_WAVES, RMP2_QUAD_LATENCY_BLOCKS) exist only in builds compiled with
 * -DRMP2_TUNING (tools/); the shipped library ignores them.  A deployment should leave all of them unset. */

/* Build an engine for one robot type + one RMP set on HIP device `device`.
 * Replaces: UrdfForwardKinematic.__init__ tables (kinematics.py:157-209) + the RmpCore
 * registry contents (rmp.py:114-131) as a flat, immutable "program".               */
int rmp2_create(const rmp2_desc *desc, int device, rmp2_handle **out);
int rmp2_destroy(rmp2_handle *h);

/* Last error message of `h` (or of the last failed rmp2_create when h == NULL). */
const char *rmp2_last_error(const rmp2_handle *h);

/* Name of the kernel (mapping of robots to lanes) the last rmp2_step / rmp2_rollout on `h` launched: which of the
 * three mappings runs is decided per call from the fleet size and the RMP set.  Diagnostic:
 * benchmarks and profiles name the kernel they measured from this instead of restating the dispatch rule. */
const char *rmp2_last_kernel(const rmp2_handle *h);

/* Stream-ordering fences for the obstacle exchange of a sharded fleet (fleet.py ObstacleExchange; the reference has
 * no counterpart -- its environment is a host-side list, rmp.py:264-315 reads it in Python).  A fence is a HIP event
 * WITHOUT timing and WITHOUT the system-scope release a default event carries: between two kernels of one stream a
 * default event's record costs ~7 us on MI355X (L2 write-back + the next kernel's cold start), a device-scope one
 * orders the same work for ~1 us.  Valid for work that stays on one GPU (the step kernel reading a table the RCCL
 * kernel of the same device wrote, and the write-after-read the other way); host-visible results still need an
 * ordinary event or a stream synchronisation.  `stream` is a hipStream_t (NULL = default stream). */
int rmp2_fence_create(int device, void **fence);
int rmp2_fence_record(void *fence, void *stream);          /* fence = everything enqueued on `stream` so far */
int rmp2_fence_wait(void *fence, void *stream);            /* later work on `stream` waits for the fence      */
int rmp2_fence_destroy(void *fence);
/* Attach `fence` to the handle: every later rmp2_step / rmp2_rollout on `h` signals it when its kernel completes -- the
 * effect of rmp2_fence_record right behind the launch, but carried by the dispatch itself (no extra packet between two
 * steps; ~3 us per step in the exchange loop).  NULL detaches.  A launch with a fence attached is not stream-capturable.
 * While a fence is attached, rmp2_step / rmp2_rollout on that handle carry per-handle state (the attachment): they must
 * then be issued from ONE thread at a time (handles stay independent of each other).  rmp2_fence_destroy refuses a fence
 * that is still attached (RMP2_ERR_INVALID_ARGUMENT); rmp2_destroy detaches. */
int rmp2_set_step_fence(rmp2_handle *h, void *fence);

/* Native obstacle exchange for a fleet sharded over the GPUs of a node (the `rmp2_comm_init` of SURVEY 8(b)):
 * every rank owns `spheres_per_rank` rows of the shared sphere table, the table [nranks * spheres_per_rank][4] is
 * all-gathered once per control step with RCCL on the exchange's own stream, double-buffered and one step ahead of the
 * kernel that consumes it; gather, stream orderings and the step launch are ONE call per control step (the same loop
 * driven from Python through torch.distributed costs ~50 us of host time per step, more than the step kernel takes).
 * RCCL is bound at run time: `rccl_library` is the path of the librccl.so the process uses (e.g. PyTorch's own copy), so
 * this library has no link-time dependency on it.  Rank 0 calls rmp2_exchange_unique_id and ships the 128 bytes to the
 * other ranks by any means (bench.py: torch.distributed broadcast); every rank then calls rmp2_exchange_create
 * (collective: it returns when all `nranks` ranks have joined).  One issuing thread per exchange. */
typedef struct rmp2_rccl_uid { char bytes[128]; } rmp2_rccl_uid;  /* ncclUniqueId */
typedef struct rmp2_exchange rmp2_exchange;
int rmp2_exchange_unique_id(const char *rccl_library, rmp2_rccl_uid *uid);
int rmp2_exchange_create(const char *rccl_library, const rmp2_rccl_uid *uid, int rank, int nranks, int device,
                         int spheres_per_rank, rmp2_exchange **out);
int rmp2_exchange_destroy(rmp2_exchange *x);
const char *rmp2_exchange_last_error(const rmp2_exchange *x);
int rmp2_exchange_pending(const rmp2_exchange *x);
/* Ranks of the communicator this exchange joined: ncclCommCount of the communicator itself, queried at create time -- which
 * also REFUSES a communicator whose ncclCommCount / ncclCommUserRank differ from the (rank, nranks) it was asked to join with --;
 * the `nranks` of the call only where the collective library does not export the two queries; 0 for NULL. */
int rmp2_exchange_nranks(const rmp2_exchange *x);
/* Pipeline depth (before the first rmp2_exchange_start): depth + 1 gathers may be outstanding.  1 (default): the table of step
 * k is gathered from slices produced before step k - 1 was issued -- the obstacles a step sees are ONE control step old, as in
 * the reference's loop, which re-reads the obstacle data every control step (06_cluttered_environment.py:120-131).  2: one more
 * control step of staleness and one more step of slack for the gather.  THREE table buffers rotate at either depth: the gather
 * for step k + 1, issued with the launch of step k, lands in a buffer whose last reader was step k - 2 -- long complete -- and so
 * starts at once and has a whole step to find a CU beside the running kernel (which fills every SIMD; with two buffers it had
 * to wait for step k - 1 and regularly made step k + 1 wait: 49.7 against 41.6 us per step at 65 536 robots,
 * profiles/r04_exchange_timing.txt).  RMP2_EXCHANGE_BUFFERS=2 (diagnostic) restores depth + 1 buffers. */
int rmp2_exchange_set_depth(rmp2_exchange *x, int32_t depth);   /* gathers started and not yet consumed by a step (0 .. 2) */
/* on != 0: a one-rank exchange orders its steps as an N-rank one does (GPU-side wait on the gathered table kept): what a
 * single-GPU EMULATION of an N-rank run must time.  No effect at nranks > 1 (the wait is always kept there). */
int rmp2_exchange_set_peer_wait(rmp2_exchange *x, int32_t on);
/* Issue the all-gather of `local` (device [spheres_per_rank][4]) into the free table buffer; it waits for the last step
 * that read the buffer it overwrites and -- local_is_ready == 0 -- for everything enqueued on `stream` so far (the producer
 * of `local`).  local_is_ready != 0: the caller guarantees that `local` is complete when this call is made (produced by an
 * earlier, already synchronised step of its pipeline, or static): no event is put on `stream`.  At most depth + 1 gathers may
 * be outstanding. */
int rmp2_exchange_start(rmp2_exchange *x, const float *local, int32_t local_is_ready, void *stream);
/* One control step (as rmp2_step with SHARED_SPHERES) on the OLDEST outstanding table.  next_local != NULL: the gather of
 * the next table is issued before the launch (same as rmp2_exchange_start(x, next_local, next_local_is_ready, stream)) --
 * or right behind it, ordered after this step's read, when all depth + 1 buffers were outstanding and the gather therefore
 * lands in the buffer this very step reads.
 * table_out (optional): device pointer of the table this step reads. */
int rmp2_exchange_step(rmp2_exchange *x, rmp2_handle *h, const float *q, const float *qd, const float *goal,
                       int32_t goal_stride, const float *next_local, int32_t next_local_is_ready, const rmp2_outputs *out,
                       int32_t R, void *stream, const float **table_out);

/* Pre-size the per-handle device buffers a step of up to R robots needs, so that rmp2_step never allocates.  Only handles whose
 * step is TWO kernels own such a buffer: solve = PINV sets without an inertia leaf and rank-deficient sets under AUTO (the
 * combined metric / force of every robot between the quad mapping and rmp2_pinv_kernel, 8 n (n + 1) bytes per robot); every
 * other handle: no-op.  Without it the first step of a larger fleet grows the buffer (hipFree + hipMalloc: a device
 * synchronisation, refused with RMP2_ERR_UNSUPPORTED while the stream is being captured).  Such a handle carries that buffer as
 * per-handle state: its steps must be issued on ONE stream at a time. */
int rmp2_reserve(rmp2_handle *h, int32_t R);

/* One control step for R robots: qdd = resolve(sum_i pullback(leaf_i))   (rmp.py:133-155).
 *   q, qd       device [R][n_dof] fp32
 *   goal        device [R][goal_floats] (goal_stride = goal_floats) or one shared row
 *               (goal_stride = 0); may be NULL when no leaf has a goal.
 *   obs         obstacle data (host struct holding device pointers); NULL = RMP2_OBS_NONE
 *   stream      hipStream_t (NULL = default stream); the call is asynchronous.          */
int rmp2_step(rmp2_handle *h, const float *q, const float *qd, const float *goal, int32_t goal_stride,
              const rmp2_obstacles *obs, const rmp2_outputs *out, int32_t R, void *stream);

/* Closed-loop rollout of the fleet inside ONE launch (the reference's control loop
 * experiments/franka_panda/06_cluttered_environment.py:120-131 with simulation.step tracking qdd):
 *   repeat n_control_steps times:  qdd = control step(q, qd);
 *                                  repeat substeps times:  qd += dt * qdd;  q += dt * qd;
 * q and qd (device, [R][n_dof]) are advanced IN PLACE; out->qdd receives the last qdd, out->status the
 * OR of the per-step status words.  Goals and the sphere table are constant during the rollout;
 * RMP2_OBS_EXPLICIT_PAIRS is rejected (closest-point pairs are only valid for the state they were
 * computed at; sets with attached-point leaves roll out when their pairs come from a SHARED_SPHERES table with
 * link_capsules -- formed anew every control step inside the launch -- instead of explicit arrays).  A handle created with
 * RMP2_SOLVE_PINV rolls out with the pseudo-inverse semantics on every robot and step (certifying elimination + Jacobi for the
 * rest where the set has an inertia leaf; otherwise the 16-lanes-per-robot mapping's Jacobi at any fleet size;
 * RMP2_ERR_UNSUPPORTED where that mapping cannot hold the program) -- never resolved differently from what was asked for. */
typedef struct rmp2_rollout_cfg {
  int32_t n_control_steps;
  int32_t substeps;
  float dt;
  int32_t table_steps; /* obstacle motion inside the rollout (the reference re-reads the obstacle data every control step,
                          06_cluttered_environment.py:120-131): 0 or 1 = one table for all control steps; n_control_steps =
                          one table per control step, obs->spheres then holds [n_control_steps][n_spheres][4 (spheres) or 8
                          (capsules)] and control step k reads table k (sphere modes only; ragged lists index every table) */
} rmp2_rollout_cfg;

int rmp2_rollout(rmp2_handle *h, float *q, float *qd, const float *goal, int32_t goal_stride,
                 const rmp2_obstacles *obs, const rmp2_rollout_cfg *cfg, const rmp2_outputs *out, int32_t R,
                 void *stream);

/* Closest-point preprocessing as a stand-alone stage (the reference's calculate_distances,
 * simulation.py:462-484, which fills the Datamanager arrays read by taskmap.py:115-138).
 * `table` must be a SHARED_SPHERES table (either primitive) with K records.  For the i-th
 * FK_DISTANCE leaf (leaf order) and record k, pair index i*K + k:
 *   p_link[r][i*K + k][3] = origin of the leaf's frame (the control point),
 *   p_obs [r][i*K + k][3] = nearest point on the surface of primitive k.
 * The two arrays are a valid EXPLICIT_PAIRS input (pair_begin[l] = i*K); feeding them back
 * reproduces the fused table mode.  p_link / p_obs: device [R][n_distance_leaves*K][3].   */
int rmp2_closest_points(rmp2_handle *h, const float *q, const rmp2_obstacles *table, float *p_link, float *p_obs,
                        int32_t R, void *stream);
/* The same stage with LINK GEOMETRY: the reference's control points are PyBullet's closest points on the link's collision
 * SHAPE, different for every (link, obstacle) pair (simulation.py:462-484 -> data_management.py:22-37), not the frame
 * origin.  link_capsules: device [n_distance_leaves][8] = (a.xyz, radius, b.xyz, unused), the link of the i-th distance leaf
 * as a capsule in that leaf's FRAME coordinates (urdf.py link_capsules: from the URDF's primitive collision geometry, or
 * supplied by the caller for mesh links).  Per pair: the nearest points of the link capsule's and the obstacle primitive's
 * surfaces (capsule-vs-sphere, capsule-vs-capsule).  link_capsules == NULL: rmp2_closest_points. */
int rmp2_closest_points_links(rmp2_handle *h, const float *q, const rmp2_obstacles *table, const float *link_capsules,
                              float *p_link, float *p_obs, int32_t R, void *stream);

/* The control steps of TWO engines (two robot types of one fleet shard: BASELINE config 5) issued together: arguments as two
 * rmp2_step calls, `stream` shared.  Where a fused instantiation exists for the pair -- a 2-dof and a 3..9-dof robot type,
 * plain steps on shared or ragged sphere tables, both fleets beyond 8 192 robots -- the two steps are ONE grid (the first
 * blocks run A's program, the rest B's; wavefronts stay type-homogeneous); otherwise two launches on `stream`.  Results are
 * those of the two rmp2_step calls either way; rmp2_last_kernel names what ran. */
int rmp2_step_pair(rmp2_handle *ha, const float *qa, const float *qda, const float *goala, int32_t goal_stride_a,
                   const rmp2_obstacles *obsa, const rmp2_outputs *outa, int32_t Ra, rmp2_handle *hb, const float *qb,
                   const float *qdb, const float *goalb, int32_t goal_stride_b, const rmp2_obstacles *obsb,
                   const rmp2_outputs *outb, int32_t Rb, void *stream);

/* Forward kinematics of every frame: T[R][n_frames][16] row-major 4x4
 * (UrdfForwardKinematic.forward, kinematics.py:212-247, for all frames at once).      */
int rmp2_forward_kinematics(rmp2_handle *h, const float *q, float *T, int32_t R, void *stream);

/* Task-map differentiation of the FK map of `frame`
 * (UrdfForwardKinematic.differentiate, kinematics.py:250-270):
 *   x[R][16] = vec(T), xd[R][16] = J qd, J[R][16][n_dof], c[R][16] = Jdot qd.
 * The two differentiate entry points are debug / test entries: they use a per-robot scratch buffer owned by the handle
 * (grown, with a device synchronisation, on the first call at a larger R), so calls on ONE handle must be issued on one
 * stream at a time; rmp2_step / rmp2_rollout / rmp2_forward_kinematics have no such state -- except while a step fence is
 * attached (rmp2_set_step_fence: one issuing thread per handle) and for the cached pair_begin table of EXPLICIT_PAIRS,
 * which is refreshed on the call's stream when the caller's pair layout changes (keep one layout per handle, or one
 * stream). */
int rmp2_differentiate(rmp2_handle *h, const float *q, const float *qd, int32_t frame, float *x, float *xd,
                       float *J, float *c, int32_t R, void *stream);

/* Task-map differentiation of the chain [FK(frame), TaskmapFrom4x4ToEuler] (taskmap.py:57-67 with
 * euler_from_rotation_matrix, kinematics.py:74-96: theta_y = -asin(r20), theta_z = atan2(r10, r00),
 * theta_x = atan2(r21, r22), i.e. R = Rz Ry Rx), used by the reference's tests/test_taskmaps.py:42-44:
 *   x[R][3] = (theta_x, theta_y, theta_z), xd[R][3] = J qd, J[R][3][n_dof], c[R][3] = Jdot qd,
 * analytically: xd = H^-1 w, J = H^-1 J_w, c = H^-1 (alpha - Hdot xd) with w = H(x) xd.  At gimbal lock
 * (|cos theta_y| < 1e-6, where the reference substitutes 1 for the cosine) the outputs are not finite. */
int rmp2_differentiate_euler(rmp2_handle *h, const float *q, const float *qd, int32_t frame, float *x, float *xd,
                             float *J, float *c, int32_t R, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RMP2_H */
