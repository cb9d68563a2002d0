// rmp2_hex.h -- the latency build of the control step: SIXTEEN LANES PER ROBOT, four robots per wave.
//
// Why: at the headline fleet size (4 096 robots) the quad kernel (rmp2_quad.h) puts 256 waves on a chip with
// 1 024 SIMDs -- three SIMDs in four idle -- and each wave runs one long dependent instruction stream.  With 16
// lanes per robot the same fleet is 1 024 waves, one per SIMD, and the per-robot work is re-cut so that its
// critical path shrinks instead of merely being shared:
//   * kinematics is no longer a serial walk.  All frames build their local transform at once (one frame per
//     lane), world transforms come from ceil(log2(depth)) rounds of POINTER JUMPING over the parent links
//     (T[k] <- T[anc_{2^l}(k)] T[k]), and velocities / bias accelerations are closed-form sums over the
//     ancestor joints evaluated by one lane per joint / per leaf frame:
//         v_k = sum_m qd_m c_m(p_k),          c_m(p) = z_m x (p - o_m)   (revolute),  z_m (prismatic)
//         a_k = sum_m qd_m d/dt c_m(p_k)  =  sum_m qd_m [ zd_m x (p_k - o_m) + z_m x (v_k - v_om) ]   (rev.),  zd_m (pris.)
//     with zd_m = w_m x z_m (w_m: angular velocity above joint m) and v_om the velocity of joint m's origin --
//     the same J qd and Jdot qd the reference gets from two jacobian_vector_product calls
//     (kinematics.py:265,267), without the chain dependence of a recursion;
//   * lane i owns ROW i of the n x n metric: the pull-back, the identity-map leaves and the fp64 resolve touch one
//     row per lane (the quad kernel: three); the resolve is Gauss-Jordan (no back substitution), one pivot-row
//     broadcast through LDS per step;
//   * a distance leaf's pairs are split 16 ways.
// gfx950's DPP has no row_share / row_xmask: sums over the 16 lanes use quad xor + row_half_mirror + row_mirror
// (4 full-rate steps); broadcasts go through LDS (all lanes of a robot sit in the same wave, LDS executes a wave's
// accesses in order).
//
// Numerics as in rmp2_quad.h: fp32 kinematics / leaves / pull-back products, fp64 accumulation and resolve,
// certification + careful fall-through (pivoted LU, then the pseudo-inverse) for (near-)singular robots.
#pragma once
#include <utility>

#include "rmp2_quad.h"

namespace rmp2 {

constexpr int kHex = 16;
constexpr int kHexRobots = kWave / kHex;  // 4
constexpr int kRowMirror = 0x140, kRowHalfMirror = 0x141;

__device__ __forceinline__ float hex_sum(float v) {
  v += dpp<kXor1>(v);
  v += dpp<kXor2>(v);
  v += dpp<kRowHalfMirror>(v);
  v += dpp<kRowMirror>(v);
  return v;
}
// lane i of a 16-lane row reads lane i - SH of the same row, 0 below the row's first lane (DPP row_shr)
template <int SH>
__device__ __forceinline__ float row_shr(float v) {
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, i, 0x110 + SH, 0xF, 0xF, true));
}
// as row_shr, but lanes without a source lane receive `fill`
template <int SH>
__device__ __forceinline__ float row_shr_or(float v, float fill) {
  const int i = __builtin_bit_cast(int, v), f = __builtin_bit_cast(int, fill);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(f, i, 0x110 + SH, 0xF, 0xF, false));
}
__device__ __forceinline__ int hex_maxi(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(0, v, kXor1, 0xF, 0xF, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, kXor2, 0xF, 0xF, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, kRowHalfMirror, 0xF, 0xF, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, kRowMirror, 0xF, 0xF, true));
  return v;
}
__device__ __forceinline__ double hex_maxd(double v) {
  v = fmax(v, dppd<kXor1>(v));
  v = fmax(v, dppd<kXor2>(v));
  v = fmax(v, dppd<kRowHalfMirror>(v));
  v = fmax(v, dppd<kRowMirror>(v));
  return v;
}
__device__ __forceinline__ bool hex_any(bool f, int g) {
  return ((__ballot(f) >> (kHex * g)) & 0xffffull) != 0ull;
}
// LDS exchange point between lanes of one robot.  The block IS one wave and the LDS unit executes a wave's
// accesses in program order, so a value written by one lane is visible to a later read of any other lane without a
// hardware barrier or a wait: all that is needed is that the COMPILER keeps the accesses in program order.
__device__ __forceinline__ void hex_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A robot's sixteen lanes are one DPP row: lane K's value to every lane of the row (v_mov_b32_dpp row_newbcast:K), two moves per
// double -- the pivot rows of the resolve travel this way, without an LDS round trip (RMP2_HEX_DPP_PIVOTS; measured against the LDS
// exchange, profiles/r05_hex_resolve_ab.txt).
#ifndef RMP2_HEX_DPP_PIVOTS
#define RMP2_HEX_DPP_PIVOTS 1
#endif
template <int K>
__device__ __forceinline__ double hex_row_bcast(double v) {
  static_assert(K >= 0 && K < 16, "a lane of the robot's row");
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + K, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + K, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int K>
__device__ __forceinline__ float hex_row_bcastf(float v) {
  static_assert(K >= 0 && K < 16, "a lane of the robot's row");
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + K, 0xF, 0xF, false));
}
template <int... Is, class F>
__device__ __forceinline__ void hex_static_for(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

// Per-robot LDS regions are padded to a stride of 8 (mod 32) floats: the four robots of a wave touch the same element
// of their own region in the same instruction, and with strides that are multiples of 32 floats (T: 12 n_ops, SC / VA:
// 8 n_ops, COL: 64) those four 16-byte accesses would queue on the same banks.
__host__ __device__ constexpr int hex_pad(int floats) { return floats + ((8 - floats % 32) + 32) % 32; }

template <int N>
struct HexLds {
  static constexpr int kQ = 0;                                          // [4][N]
  static constexpr int kQd = kHexRobots * N;                            // [4][N]
  static constexpr int kOut = 2 * kHexRobots * N;                       // [4][N]
  static constexpr int kDof = (3 * kHexRobots * N + 3) & ~3;            // [4][N][8]: world axis z_j _, joint origin o_j _
  static constexpr int kCol = kDof + kHexRobots * N * 8;                // [4][16][4]: column of dof s, current frame
  static constexpr int kXch = kCol;                                     // [4][16][4]: identity-leaf exchange (the FK-leaf phase is over)
  static constexpr int kRowStride = (2 * (N + 1) + 3) & ~3;             // one pivot row [A_k | f_k] as doubles
  static constexpr int kColStride = hex_pad(kHex * 4);                  // per-robot stride of COL / XCH
  static constexpr int kRow = kXch + kHexRobots * kColStride;           // [4][2][kRowStride]: two pivot rows per exchange
  static constexpr int kSysStride = 2 * (2 * N * (N + 1) + N);          // the whole system [N][N+1] as doubles, a working
                                                                        // copy of it and x[N] for the careful solver
  static constexpr int kSys = kRow + kHexRobots * 2 * kRowStride;       // [4][kSysStride]
  static constexpr int kFloats = (kSys + kHexRobots * kSysStride + 3) & ~3;
  // dynamic: T [4][n_ops][12] | SC [4][n_ops][8] | VA [4][n_ops][8] | sphere table | staged program:
  //   HexCtl | HexOp | leaves (execution order) | leaf-frame records | jump | op_anc
};

// bytes of dynamic LDS a launch needs (host side)
template <int N>
inline size_t hex_lds_bytes(int waves, int n_ops, int blob16, int n_sphere_floats, bool with_wa = false) {
  const size_t per_wave = HexLds<N>::kFloats + kHexRobots * hex_pad(n_ops * 12) +
                          (with_wa ? 3 : 2) * kHexRobots * hex_pad(n_ops * 8) + 16 * kHexRobots +
                          kHexRobots * RMP2_MAX_DOF;
  return sizeof(float) * (waves * per_wave + n_sphere_floats) + 16 * (size_t)blob16;
}

// global -> LDS copy with up to four loads per lane in flight before the first LDS write
template <int THREADS, typename T>
__device__ __forceinline__ void stage_copy(T* dst, const T* __restrict__ src, int n, int tid) {
  for (int base = 0; base < n; base += 4 * THREADS) {
    T v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = src[min(base + u * THREADS + tid, n - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + u * THREADS + tid;
      if (i < n) dst[i] = v[u];
    }
  }
}

#ifdef RMP2_STAMPS_KIN  // diagnostic: spend the stamps inside the kinematics phase instead of after the later phases
#define RMP2_KSTAMP() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (stp) stp[(*stn)++] = __builtin_amdgcn_s_memtime(); } while (0)
#define RMP2_LSTAMP() do {} while (0)
#else
#define RMP2_KSTAMP() do {} while (0)
#define RMP2_LSTAMP() RMP2_STAMP()
#endif

// ---- kinematics of all frames, SLOTS frames per lane (k = s, s + 16) ---------------------------------------------
// Written as straight-line code: every LDS read that does not depend on another read's result is issued
// unconditionally (clamped indices, 0/1 weights instead of branches) so that the reads of one stage are in flight
// together -- the wave is alone on its SIMD, nothing else hides an LDS round trip.
//   stage 1  local transforms (kinematics.py:222-240) from the host-folded form R = A0 + cos(q) A1 + sin(q) A2
//   stage 2  world transforms by pointer jumping (ordered product of kinematics.py:243-246): after round l,
//            T[k] is the product of the last min(2^(l+1), depth) local transforms ending at k
//   stage 3  velocity and bias acceleration of every frame origin from two sums over the ancestors.  With
//            w_m = qd_m z_m, b_m = -qd_m z_m x o_m (revolute) or w_m = 0, b_m = qd_m z_m (prismatic) and W_k, B_k
//            their sums over the joints at or above frame k:
//                v_k = W_k x p_k + B_k                                   ( = J_k qd,    kinematics.py:265 )
//            and with zd_m = W_m x z_m, al_m = qd_m zd_m, c_m = -qd_m (zd_m x o_m + z_m x v_m) (revolute) or
//            al_m = 0, c_m = qd_m zd_m (prismatic) and AL_k, C_k their sums:
//                a_k = AL_k x p_k + W_k x v_k + C_k                      ( = Jdot_k qd, kinematics.py:267 )
// Leaves world transforms in the returned buffer [n_ops][12], (v, a) in VA [n_ops][8], (z_j, o_j) in DOF [N][8].
template <int N, int SLOTS>
__device__ __forceinline__ const float* hex_kinematics(const QuadHdr& hdr, int s, int g, const float* my_q,
                                                       const float* my_qd, float* T0, float* SCb, float* VAb,
                                                       float* DOF, const HexCtl* s_ctl, const HexOp* s_hops,
                                                       const int32_t* s_jump, const uint32_t* s_op_anc, float* WAb = nullptr,
                                                       unsigned long long* stp = nullptr, int* stn = nullptr) {
  const int n_ops = hdr.n_ops;
  float* const Tb0 = T0 + g * hex_pad(n_ops * 12);
  float4* const SC = reinterpret_cast<float4*>(SCb + g * hex_pad(n_ops * 8));
  float4* const VA = reinterpret_cast<float4*>(VAb + g * hex_pad(n_ops * 8));
  bool on[SLOTS], revk[SLOTS], prik[SLOTS];
  int kk[SLOTS], jmp[SLOTS][5], qi[SLOTS];
  uint32_t ancm[SLOTS];
  float Tm[SLOTS][12], ax[SLOTS][3], qv[SLOTS], qdk[SLOTS];
  // ---- stage 1 ----
#pragma unroll
  for (int slot = 0; slot < SLOTS; ++slot) {
    const int k = s + kHex * slot;
    on[slot] = k < n_ops;
    kk[slot] = on[slot] ? k : 0;
  }
  float4 hop[SLOTS][9];
#pragma unroll
  for (int slot = 0; slot < SLOTS; ++slot) {
    const int4* c4 = reinterpret_cast<const int4*>(&s_ctl[kk[slot]]);
    const int4 c0 = c4[0];                                            // jtype, qidx, anc_mask, leaf_begin
    const float4 c1 = reinterpret_cast<const float4*>(c4)[1];         // (leaf_count), axis
    const int jt = c0.x;
    qi[slot] = on[slot] ? c0.y : -1;  // lanes without a frame read record 0 (or, with no frames at all, whatever follows)
    ax[slot][0] = c1.y, ax[slot][1] = c1.z, ax[slot][2] = c1.w;
    revk[slot] = on[slot] && jt == RMP2_JOINT_REVOLUTE;
    prik[slot] = on[slot] && jt == RMP2_JOINT_PRISMATIC;
    const float4* h4 = reinterpret_cast<const float4*>(&s_hops[kk[slot]]);
#pragma unroll
    for (int c = 0; c < 9; ++c) hop[slot][c] = h4[c];
#pragma unroll
    for (int l = 0; l < 5; ++l) jmp[slot][l] = (on[slot] && l < hdr.n_levels) ? s_jump[l * n_ops + kk[slot]] : -1;
    ancm[slot] = on[slot] ? s_op_anc[kk[slot]] : 0u;
  }
#pragma unroll
  for (int slot = 0; slot < SLOTS; ++slot) {
    const int qc = qi[slot] >= 0 ? qi[slot] : 0;
    const float a = my_q[qc], b = my_qd[qc];
    qv[slot] = qi[slot] >= 0 ? a : 0.f;
    qdk[slot] = (qi[slot] >= 0 && (revk[slot] || prik[slot])) ? b : 0.f;
  }
#pragma unroll
  for (int slot = 0; slot < SLOTS; ++slot) {
    float sn = 0.f, cs = 1.f;
    if (revk[slot]) {
      if (fabsf(qv[slot]) <= 8192.0f)
        sincos1(qv[slot], sn, cs);
      else
        sincosf(qv[slot], &sn, &cs);
    }
    const float tq = prik[slot] ? qv[slot] : 0.f;
    const float* hf = reinterpret_cast<const float*>(hop[slot]);  // A0[9] A1[9] A2[9] tc[3] tu[3]
#pragma unroll
    for (int c = 0; c < 9; ++c) Tm[slot][c] = fmaf(sn, hf[18 + c], fmaf(cs, hf[9 + c], hf[c]));
#pragma unroll
    for (int r = 0; r < 3; ++r) Tm[slot][9 + r] = fmaf(tq, hf[30 + r], hf[27 + r]);
  }
  RMP2_KSTAMP();  // K: local transforms done
  // ---- stage 2 ----
  // serial chains in program order (frame k in lane k, parent in lane k - 1, at most 16 frames): the ancestor 2^l
  // levels up sits 2^l lanes down the row -- its transform arrives through DPP row_shr operands (lanes without
  // such an ancestor receive the identity), no LDS round trip
  const bool dpp_chain = SLOTS == 1 && hdr.is_chain;  // wave-uniform
  if (dpp_chain) {
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      if (l < hdr.n_levels) {
        float Pr[12];
#pragma unroll
        for (int c = 0; c < 12; ++c)
          Pr[c] = (l == 0)   ? row_shr_or<1>(Tm[0][c], (c == 0 || c == 4 || c == 8) ? 1.f : 0.f)
                  : (l == 1) ? row_shr_or<2>(Tm[0][c], (c == 0 || c == 4 || c == 8) ? 1.f : 0.f)
                  : (l == 2) ? row_shr_or<4>(Tm[0][c], (c == 0 || c == 4 || c == 8) ? 1.f : 0.f)
                             : row_shr_or<8>(Tm[0][c], (c == 0 || c == 4 || c == 8) ? 1.f : 0.f);
        float Tn[12];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
          for (int c = 0; c < 3; ++c)
            Tn[3 * r + c] = Pr[3 * r] * Tm[0][c] + Pr[3 * r + 1] * Tm[0][3 + c] + Pr[3 * r + 2] * Tm[0][6 + c];
          Tn[9 + r] = Pr[3 * r] * Tm[0][9] + Pr[3 * r + 1] * Tm[0][10] + Pr[3 * r + 2] * Tm[0][11] + Pr[9 + r];
        }
#pragma unroll
        for (int c = 0; c < 12; ++c) Tm[0][c] = Tn[c];
      }
    }
  }
#pragma unroll
  for (int l = 0; l < 5; ++l) {
    if (!dpp_chain && l < hdr.n_levels) {  // wave-uniform
      float* const src = Tb0;  // one buffer: a round's reads are issued before the next round's writes (same wave)
#pragma unroll
      for (int slot = 0; slot < SLOTS; ++slot) {
        if (on[slot]) {
          float4* dst = reinterpret_cast<float4*>(src + kk[slot] * 12);
#pragma unroll
          for (int c = 0; c < 3; ++c)
            dst[c] = make_float4(Tm[slot][4 * c], Tm[slot][4 * c + 1], Tm[slot][4 * c + 2], Tm[slot][4 * c + 3]);
        }
      }
      hex_sync();
      float4 pa[SLOTS][3];
#pragma unroll
      for (int slot = 0; slot < SLOTS; ++slot) {
        const int j = jmp[slot][l] >= 0 ? jmp[slot][l] : kk[slot];
        const float4* p4 = reinterpret_cast<const float4*>(src + j * 12);
        pa[slot][0] = p4[0], pa[slot][1] = p4[1], pa[slot][2] = p4[2];
      }
      hex_sync();
#pragma unroll
      for (int slot = 0; slot < SLOTS; ++slot) {
        if (jmp[slot][l] >= 0 && on[slot]) {
          const float* Pr = reinterpret_cast<const float*>(pa[slot]);
          float Tn[12];
#pragma unroll
          for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
              Tn[3 * r + c] = Pr[3 * r] * Tm[slot][c] + Pr[3 * r + 1] * Tm[slot][3 + c] + Pr[3 * r + 2] * Tm[slot][6 + c];
            Tn[9 + r] = Pr[3 * r] * Tm[slot][9] + Pr[3 * r + 1] * Tm[slot][10] + Pr[3 * r + 2] * Tm[slot][11] + Pr[9 + r];
          }
#pragma unroll
          for (int c = 0; c < 12; ++c) Tm[slot][c] = Tn[c];
        }
      }
    }
  }
  float* const TW = Tb0;
#pragma unroll
  for (int slot = 0; slot < SLOTS; ++slot) {
    if (on[slot]) {
      float4* dst = reinterpret_cast<float4*>(TW + kk[slot] * 12);
#pragma unroll
      for (int c = 0; c < 3; ++c)
        dst[c] = make_float4(Tm[slot][4 * c], Tm[slot][4 * c + 1], Tm[slot][4 * c + 2], Tm[slot][4 * c + 3]);
    }
  }
  RMP2_KSTAMP();  // K: world transforms done
  // ---- stage 3 ----
  float pk[SLOTS][3], zk[SLOTS][3], X[SLOTS][6];
#pragma unroll
  for (int slot = 0; slot < SLOTS; ++slot) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      zk[slot][r] = Tm[slot][3 * r] * ax[slot][0] + Tm[slot][3 * r + 1] * ax[slot][1] + Tm[slot][3 * r + 2] * ax[slot][2];
      pk[slot][r] = Tm[slot][9 + r];
    }
    if ((revk[slot] || prik[slot]) && qi[slot] >= 0) {  // per-dof table for the Jacobian columns
      float4* d4 = reinterpret_cast<float4*>(DOF + qi[slot] * 8);
      d4[0] = make_float4(zk[slot][0], zk[slot][1], zk[slot][2], 0.f);
      d4[1] = make_float4(pk[slot][0], pk[slot][1], pk[slot][2], 0.f);
    }
    float zxo[3];
    cross3(zk[slot], pk[slot], zxo);
    const float qdm = qdk[slot];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      X[slot][r] = revk[slot] ? qdm * zk[slot][r] : 0.f;
      X[slot][3 + r] = revk[slot] ? -qdm * zxo[r] : (prik[slot] ? qdm * zk[slot][r] : 0.f);
    }
  }
  // inclusive sum of x over op k and its ancestors: every lane adds up the published terms of all ops, weighted
  // by the 0/1 membership bit of its ancestor mask
  auto ancestor_sum = [&](float (&x)[SLOTS][6]) {
#pragma unroll
    for (int slot = 0; slot < SLOTS; ++slot) {
      if (on[slot]) {
        SC[2 * kk[slot]] = make_float4(x[slot][0], x[slot][1], x[slot][2], x[slot][3]);
        SC[2 * kk[slot] + 1] = make_float4(x[slot][4], x[slot][5], 0.f, 0.f);
      }
    }
    hex_sync();
    float acc[SLOTS][6];
#pragma unroll
    for (int slot = 0; slot < SLOTS; ++slot)
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[slot][c] = 0.f;
    for (int j0 = 0; j0 < n_ops; j0 += 8) {
      float4 u0[8], u1[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int j = min(j0 + jj, n_ops - 1);  // clamped: the repeated op carries weight 0
        u0[jj] = SC[2 * j];
        u1[jj] = SC[2 * j + 1];
      }
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
#pragma unroll
        for (int slot = 0; slot < SLOTS; ++slot) {
          const float f = (j0 + jj < n_ops && ((ancm[slot] >> (j0 + jj)) & 1u)) ? 1.f : 0.f;
          acc[slot][0] = fmaf(f, u0[jj].x, acc[slot][0]);
          acc[slot][1] = fmaf(f, u0[jj].y, acc[slot][1]);
          acc[slot][2] = fmaf(f, u0[jj].z, acc[slot][2]);
          acc[slot][3] = fmaf(f, u0[jj].w, acc[slot][3]);
          acc[slot][4] = fmaf(f, u1[jj].x, acc[slot][4]);
          acc[slot][5] = fmaf(f, u1[jj].y, acc[slot][5]);
        }
      }
    }
#pragma unroll
    for (int slot = 0; slot < SLOTS; ++slot)
#pragma unroll
      for (int c = 0; c < 6; ++c) x[slot][c] = acc[slot][c];
    hex_sync();
  };
  // serial chains in program order (frame k in lane k, parent in lane k - 1, at most 16 frames): the ancestor sum is
  // an inclusive prefix sum along the lanes of the row -- four DPP row_shr steps per component, no LDS round trip
  auto chain_prefix = [&](float (&x)[SLOTS][6]) {
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      float v = x[0][c];
      v += row_shr<1>(v);
      v += row_shr<2>(v);
      v += row_shr<4>(v);
      v += row_shr<8>(v);
      x[0][c] = v;
    }
  };
  RMP2_KSTAMP();  // K: per-op terms formed
  if (dpp_chain)
    chain_prefix(X);
  else
    ancestor_sum(X);  // X = (W_k, B_k)
  RMP2_KSTAMP();  // K: first sum done
  float vk[SLOTS][3], Y[SLOTS][6];
#pragma unroll
  for (int slot = 0; slot < SLOTS; ++slot) {
    const float Wk[3] = {X[slot][0], X[slot][1], X[slot][2]};
    float t1[3], zd[3], t2[3], t3[3];
    cross3(Wk, pk[slot], t1);
#pragma unroll
    for (int r = 0; r < 3; ++r) vk[slot][r] = t1[r] + X[slot][3 + r];
    cross3(Wk, zk[slot], zd);
    cross3(zd, pk[slot], t2);
    cross3(zk[slot], vk[slot], t3);
    const float qdm = qdk[slot];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      Y[slot][r] = revk[slot] ? qdm * zd[r] : 0.f;
      Y[slot][3 + r] = revk[slot] ? -qdm * (t2[r] + t3[r]) : (prik[slot] ? qdm * zd[r] : 0.f);
    }
  }
  RMP2_KSTAMP();  // K: second terms formed
  if (dpp_chain)
    chain_prefix(Y);
  else
    ancestor_sum(Y);  // Y = (AL_k, C_k)
#pragma unroll
  for (int slot = 0; slot < SLOTS; ++slot) {
    if (on[slot]) {
      const float Wk[3] = {X[slot][0], X[slot][1], X[slot][2]}, ALk[3] = {Y[slot][0], Y[slot][1], Y[slot][2]};
      float t1[3], t2[3], ak[3];
      cross3(ALk, pk[slot], t1);
      cross3(Wk, vk[slot], t2);
#pragma unroll
      for (int r = 0; r < 3; ++r) ak[r] = t1[r] + t2[r] + Y[slot][3 + r];
      VA[2 * kk[slot]] = make_float4(vk[slot][0], vk[slot][1], vk[slot][2], ak[0]);
      VA[2 * kk[slot] + 1] = make_float4(ak[1], ak[2], 0.f, 0.f);
      if (WAb) {  // angular velocity and angular bias acceleration of the frame (attached-point leaves)
        float4* const WA = reinterpret_cast<float4*>(WAb + g * hex_pad(n_ops * 8));
        WA[2 * kk[slot]] = make_float4(Wk[0], Wk[1], Wk[2], ALk[0]);
        WA[2 * kk[slot] + 1] = make_float4(ALk[1], ALk[2], 0.f, 0.f);
      }
    }
  }
  hex_sync();
  return TW;
}

// WAVES = waves per block.  Each wave owns four robots and its own LDS working set; the waves of a block share one
// staged copy of the program and of the obstacle table (loaded cooperatively, one real barrier after the prologue).
// ROLL = build with the fused closed-loop rollout (control-step loop + plant ticks); the plain step carries none of it.
// PT = build with the attached-point leaves (CollisionAvoidance on [FK, TaskmapRelative4x4, 4x4 -> position], taskmap.py:79-99,
// rmp.py:264-315): every pair carries its own Jacobian (the point is offset from the frame origin); the plain build has
// none of that code.
template <int N, bool CAP, int WAVES, bool ROLL, bool PT = false>
__global__ void __launch_bounds__(kWave * WAVES, 1)
rmp2_step_hex_kernel(const uint4* __restrict__ blob, int blob16, QuadHdr hdr, const float* __restrict__ q,
                     const float* __restrict__ qd, const float* __restrict__ goal, int goal_stride, ObsArgs obs,
                     OutArgs out, RolloutArgs ro, int R) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef RMP2_STAMPS
  unsigned long long st_[8];
  int st_n = 0;
#endif
#if defined(RMP2_STAMPS) && defined(RMP2_STAMPS_KIN)
#define RMP2_KARGS , st_, &st_n
#else
#define RMP2_KARGS
#endif
  RMP2_STAMP();
  const int tid = threadIdx.x;
  const int wv = tid >> 6;
  const int lane = tid & (kWave - 1);
  const int s = lane & (kHex - 1);
  const int g = lane >> 4;
  const int r0 = (blockIdx.x * WAVES + wv) * kHexRobots;
  const int robot = r0 + g;
  const bool live = robot < R;
  const int n_dof = hdr.n_dof, n_ops = hdr.n_ops, n_id = hdr.n_id, n_lo = hdr.n_leaf_ops;
  const uint32_t rev_mask = hdr.rev_mask;
  const int n_live = min(kHexRobots, R - r0);
  const int gi = min(g, n_live - 1);  // groups beyond the fleet's tail re-use the last live robot's inputs

  // ---- LDS carve-up -----------------------------------------------------------------------------------
  // [per-wave static x WAVES | per-wave T SC VA x WAVES | per-wave goal tile x WAVES | block qdd tile | sphere table | program]
  float* const wl = lds + wv * HexLds<N>::kFloats;                       // this wave's static region
  const int dyn_per_wave = kHexRobots * hex_pad(n_ops * 12) + (PT ? 3 : 2) * kHexRobots * hex_pad(n_ops * 8);
  float* const T0 = lds + WAVES * HexLds<N>::kFloats + wv * dyn_per_wave;
  float* const SCb = T0 + kHexRobots * hex_pad(n_ops * 12);   // [4][n_ops][8] prefix-sum exchange
  float* const VAb = SCb + kHexRobots * hex_pad(n_ops * 8);   // [4][n_ops][8] (v, a) of every frame origin
  float* const WAb = PT ? VAb + kHexRobots * hex_pad(n_ops * 8) : nullptr;  // [4][n_ops][8] (w, alpha), PT builds only
  const int n_sph_lds = (obs.mode == RMP2_OBS_SHARED_SPHERES || obs.mode == RMP2_OBS_RAGGED_SPHERES)
                            ? min(obs.n_spheres, kLdsSpheres) : 0;
  float* const goal_base = lds + WAVES * (HexLds<N>::kFloats + dyn_per_wave);
  float* const s_goal = goal_base + wv * 16 * kHexRobots;
  float* const blk_out = goal_base + WAVES * 16 * kHexRobots;             // [WAVES * 4][n_dof] qdd tile of the block
  float* const sph_lds_base = blk_out + WAVES * kHexRobots * RMP2_MAX_DOF;
  float* const stage_base = sph_lds_base + sphere_lds_floats(CAP, n_sph_lds);
  // the staged program: same layout as the host's blob
  HexCtl* const s_ctl = reinterpret_cast<HexCtl*>(stage_base);
  HexOp* const s_hops = reinterpret_cast<HexOp*>(s_ctl + n_ops);
  DevLeaf* const s_leaves = reinterpret_cast<DevLeaf*>(s_hops + n_ops);
  const int4* const s_lo = reinterpret_cast<const int4*>(s_leaves + hdr.n_leaves);  // (op, anc dofs, first leaf, count)
  int32_t* const s_jump = reinterpret_cast<int32_t*>(const_cast<int4*>(s_lo) + n_lo);
  uint32_t* const s_op_anc = reinterpret_cast<uint32_t*>(s_jump + hdr.n_levels * n_ops);

  // ---- prologue: one burst of loads brings the state tile, the obstacle table and the program on chip ----
  {
    const int tile = max(n_live, 0) * n_dof;
    const float* gq = q + (size_t)r0 * n_dof;
    const float* gqd = qd + (size_t)r0 * n_dof;
    // state tile: with n_dof == N rows are contiguous in HBM and in LDS and the tile (<= 4 N <= 64 floats) is one
    // load per lane, issued together with the loads below; other cases take the generic path after them
    const bool fastq = n_dof == N && tile <= kWave;
    float qa = 0.f, qb = 0.f;
    if (fastq && lane < tile) {
      qa = gq[lane];
      qb = gqd[lane];
    }
    // the program blob, the obstacle table and the goals: loads first, LDS writes after
    const int nf4 = (CAP ? 2 : 1) * n_sph_lds;
    const int ng = goal ? max(n_live, 0) * hdr.goal_floats : 0;  // goal_floats <= 16 (checked on the host): ng <= 64
    constexpr int kBlk = kWave * WAVES;  // the block's threads share the staging of the program and the table
    uint4 bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) bv[u] = blob[min(u * kBlk + tid, blob16 - 1)];  // clamped: no branch around the load
    float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < nf4) sv = reinterpret_cast<const float4*>(obs.spheres)[tid];
    float gv = 0.f;
    int g_rr = 0, g_jj = 0;
    if (lane < ng) {
      g_rr = lane / hdr.goal_floats;
      g_jj = lane - g_rr * hdr.goal_floats;
      gv = goal[(size_t)(r0 + g_rr) * goal_stride + g_jj];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = u * kBlk + tid;
      if (i < blob16) reinterpret_cast<uint4*>(stage_base)[i] = bv[u];
    }
    if (tid < nf4) {
      if (CAP) {
        reinterpret_cast<float4*>(sph_lds_base)[tid] = sv;
      } else {  // image for the culled pair loop (rmp2_quad.h): {-2c, |c|^2 - thr^2} records, then the radii
        reinterpret_cast<float4*>(sph_lds_base)[tid] = sphere_aux(sv, hdr.cull_c0);
        sph_lds_base[4 * n_sph_lds + tid] = sv.w;
      }
    }
    if (lane < ng) s_goal[g_rr * 16 + g_jj] = gv;
    if (fastq) {
      if (lane < tile) {
        quarantine(qa, qb);  // (rmp2_device.h: a non-finite joint position moves into the joint's velocity)
        wl[HexLds<N>::kQ + lane] = qa;
        wl[HexLds<N>::kQd + lane] = qb;
      }
    } else {
      for (int i = lane; i < tile; i += kWave) {
        const int rr = i / n_dof, jj = i - rr * n_dof;
        float qv = gq[i], qdv = gqd[i];
        quarantine(qv, qdv);
        wl[HexLds<N>::kQ + rr * N + jj] = qv;
        wl[HexLds<N>::kQd + rr * N + jj] = qdv;
      }
      if (n_dof < N) {  // padding dofs of the template read as q = qd = 0
        const int pad = N - n_dof;
        for (int i = lane; i < kHexRobots * pad; i += kWave) {
          const int rr = i / pad, jj = n_dof + (i - rr * pad);
          wl[HexLds<N>::kQ + rr * N + jj] = 0.f;
          wl[HexLds<N>::kQd + rr * N + jj] = 0.f;
        }
      }
    }
    // leftovers of big programs / big obstacle tables
    if (blob16 > 4 * kBlk)
      stage_copy<kBlk>(reinterpret_cast<uint4*>(stage_base) + 4 * kBlk, blob + 4 * kBlk, blob16 - 4 * kBlk, tid);
    if (nf4 > kBlk) {
      if (CAP) {
        stage_copy<kBlk>(reinterpret_cast<float4*>(sph_lds_base) + kBlk, reinterpret_cast<const float4*>(obs.spheres) + kBlk,
                         nf4 - kBlk, tid);
      } else {
        for (int i = kBlk + tid; i < nf4; i += kBlk) {
          const float4 sp = reinterpret_cast<const float4*>(obs.spheres)[i];
          reinterpret_cast<float4*>(sph_lds_base)[i] = sphere_aux(sp, hdr.cull_c0);
          sph_lds_base[4 * n_sph_lds + i] = sp.w;
        }
      }
    }
    if (WAVES > 1)
      __syncthreads();  // the only cross-wave dependence: the shared staged program / obstacle table
    else
      hex_sync();
  }
  // A wave past the fleet's tail has done its share of the staging and has no robot: it runs ZERO control steps below
  // and meets the others at the block's final barrier (a barrier some waves never reach is undefined in the HIP model,
  // even though gfx9's s_barrier happens to drop ended waves).
  RMP2_STAMP();  // 1: prologue done
  const bool spheres_in_lds = obs.n_spheres <= kLdsSpheres;
  const float* my_q = &wl[HexLds<N>::kQ + gi * N];
  const float* my_qd = &wl[HexLds<N>::kQd + gi * N];
  float* my_out = blk_out + (wv * kHexRobots + g) * n_dof;
  const float* my_goal = goal ? s_goal + gi * 16 : nullptr;
  float* const DOF = &wl[HexLds<N>::kDof + g * N * 8];
  float4* const COL = reinterpret_cast<float4*>(&wl[HexLds<N>::kCol + g * HexLds<N>::kColStride]);
  float4* const XCH = reinterpret_cast<float4*>(&wl[HexLds<N>::kXch + g * HexLds<N>::kColStride]);
  uint32_t status = 0u;

  // closed-loop rollout (rmp2_rollout): n_iters control steps inside this launch; a plain step is one iteration
  const int n_iters = n_live <= 0 ? 0 : (ROLL ? ro.n_iters : 1);
  // (the plain step's condition below is visibly "at most once": the compiler emits a branch, not a loop -- a run-time
  // trip count costs the step ~3 k cycles of loop-carried state)
  // ragged lists over a table of at most 64 spheres: the robot's list as a membership mask, built once by its 16 lanes
  // (rmp2_quad.h, pair_loop_culled<MEMBER>); a list with a repeated or out-of-range index keeps the list walk
  uint32_t member_lo = 0u, member_hi = 0u;
  bool use_member = false;
  if (obs.mode == RMP2_OBS_RAGGED_SPHERES && !CAP && obs.n_spheres <= 64 && n_live > 0) {
    const int rr_ = live ? robot : 0;
    const int b0 = obs.csr_offset[rr_];
    const int count = live ? obs.csr_offset[rr_ + 1] - b0 : 0;
    int max_count = count;
#pragma unroll
    for (int o = 32; o >= kHex; o >>= 1) max_count = max(max_count, __shfl_xor(max_count, o));
    bool bad = false;
    for (int t = s; t - s < max_count; t += kHex) {  // (wave-uniform trip count)
      if (t < count) {
        const int idx = obs.csr_index[b0 + t];
        bad = bad || idx < 0 || idx >= obs.n_spheres;
        if (idx >= 0 && idx < 32) member_lo |= 1u << idx;
        if (idx >= 32 && idx < 64) member_hi |= 1u << (idx - 32);
      }
    }
    member_lo |= dppu<kXor1>(member_lo);
    member_lo |= dppu<kXor2>(member_lo);
    member_lo |= dppu<0x141>(member_lo);  // row_half_mirror
    member_lo |= dppu<0x140>(member_lo);  // row_mirror
    member_hi |= dppu<kXor1>(member_hi);
    member_hi |= dppu<kXor2>(member_hi);
    member_hi |= dppu<0x141>(member_hi);
    member_hi |= dppu<0x140>(member_hi);
    bad = bad || (__builtin_popcount(member_lo) + __builtin_popcount(member_hi) != count);
    use_member = !__any(bad);
  }
  const int n_iters_blk = ROLL ? ro.n_iters : 1;  // (uniform over the block: tail waves meet the barriers below without stepping)
#pragma nounroll
  for (int it = 0; ROLL ? (it < n_iters_blk) : (it == 0 && n_live > 0); ++it) {
  // moving obstacles (RolloutArgs::table_stride): control step `it` reads table `it`.  The LDS image is shared by the
  // block's waves: every wave must be through with table it - 1 before it is overwritten, and see table it before reading
  const float* const step_table = obs.spheres + (size_t)it * (size_t)(ROLL ? ro.table_stride : 0);
  if (ROLL && ro.table_stride != 0 && it > 0 && n_sph_lds > 0) {
    if (WAVES > 1) __syncthreads(); else hex_sync();
    constexpr int kBlk2 = kWave * WAVES;
    const int tid2 = threadIdx.x;
    for (int i = tid2; i < (CAP ? 2 : 1) * n_sph_lds; i += kBlk2) {
      const float4 sv2 = reinterpret_cast<const float4*>(step_table)[i];
      if (CAP) {
        reinterpret_cast<float4*>(sph_lds_base)[i] = sv2;
      } else {
        reinterpret_cast<float4*>(sph_lds_base)[i] = sphere_aux(sv2, hdr.cull_c0);
        sph_lds_base[4 * n_sph_lds + i] = sv2.w;
      }
    }
    if (WAVES > 1) __syncthreads(); else hex_sync();
  }
  if (ROLL && it >= n_iters) continue;
  // ---- kinematics of all frames (hex_kinematics below): world transforms, J qd and Jdot qd of every origin ------
  const float* TW;
  if (n_ops <= kHex)
    TW = hex_kinematics<N, 1>(hdr, s, g, my_q, my_qd, T0, SCb, VAb, DOF, s_ctl, s_hops, s_jump, s_op_anc, WAb RMP2_KARGS);
  else
    TW = hex_kinematics<N, 2>(hdr, s, g, my_q, my_qd, T0, SCb, VAb, DOF, s_ctl, s_hops, s_jump, s_op_anc, WAb RMP2_KARGS);
  RMP2_STAMP();  // 2: kinematics done

  // ---- the fp64 system, one row per lane: A[j] = M[s][j], fv = f[s] -------------------------------------
  double A[N];
  double fv = 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j) A[j] = 0.0;
  const bool my_rev = (rev_mask >> (s < N ? s : 0)) & 1u;

  // ---- leaves on FK task maps, frame by frame -----------------------------------------------------------
  for (int t = 0; t < n_lo; ++t) {
    const int4 lo = s_lo[t];
    const int k = uni<true>(lo.x);
    struct {
      uint32_t anc_mask;
      int leaf_begin, leaf_count;
    } op = {(uint32_t)uni<true>(lo.y), uni<true>(lo.z), uni<true>(lo.w)};
    const float4 tp = reinterpret_cast<const float4*>(TW + k * 12)[2];
    const float4* va4 = reinterpret_cast<const float4*>(VAb + g * hex_pad(n_ops * 8) + k * 8);
    const float4 f0 = va4[0], f1 = va4[1];
    const float P3[3] = {tp.y, tp.z, tp.w}, V3[3] = {f0.x, f0.y, f0.z}, A3[3] = {f0.w, f1.x, f1.y};
    // Jacobian column of MY dof at a point moving with this frame (its origin, or an attached point); all columns
    // through LDS.  Columns of dofs that do not move the frame are written as zeros: no branches in the pull-back.
    float mycol[3];
    float col[N][3];
    auto set_cols = [&](const float pt[3]) {
      mycol[0] = mycol[1] = mycol[2] = 0.f;
      if (s < N && ((op.anc_mask >> s) & 1u)) {
        const float4* d4 = reinterpret_cast<const float4*>(DOF + s * 8);
        const float4 z4 = d4[0], o4 = d4[1];
        const float zj[3] = {z4.x, z4.y, z4.z};
        // (bit 16 + s: the frame's origin P3 lies on my joint's axis for every q -- rmp2_hip.hip structural_lever_zeros: it is
        //  the axis point then, the origin's lever exactly zero as in the reference, an attached point's its offset from P3)
        const bool on_axis = (op.anc_mask >> (16 + s)) & 1u;
        const float dd[3] = {pt[0] - (on_axis ? P3[0] : o4.x), pt[1] - (on_axis ? P3[1] : o4.y), pt[2] - (on_axis ? P3[2] : o4.z)};
        float cr[3];
        cross3(zj, dd, cr);
#pragma unroll
        for (int c = 0; c < 3; ++c) mycol[c] = my_rev ? cr[c] : zj[c];
      }
      COL[s] = make_float4(mycol[0], mycol[1], mycol[2], 0.f);
      hex_sync();
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float4 c4 = COL[j];
        col[j][0] = c4.x, col[j][1] = c4.y, col[j][2] = c4.z;
      }
    };
    // pull-back into MY row:  f_s += col_s . h ;  A[s][j] += (S col_s) . col_j   (rmp.py:165-167)
    auto pull_back = [&](const float S[6], const float h[3]) {
      float u[3];
      if (hdr.rank1) {  // (wave-uniform) sets without an inertia leaf: the rank-one form of the leaf metric (rmp2_device.h)
        metric_times_column(S, rank_one_of(S), mycol, u);
      } else {
        u[0] = S[0] * mycol[0] + S[1] * mycol[1] + S[2] * mycol[2];
        u[1] = S[1] * mycol[0] + S[3] * mycol[1] + S[4] * mycol[2];
        u[2] = S[2] * mycol[0] + S[4] * mycol[1] + S[5] * mycol[2];
      }
      fv += (double)dot3(mycol, h);
#pragma unroll
      for (int j = 0; j < N; ++j) A[j] += (double)dot3(u, col[j]);
    };
    set_cols(P3);
    for (int li = 0; li < op.leaf_count; ++li) {
      const DevLeaf& lf = s_leaves[op.leaf_begin + li];  // staged in execution order
      LeafHead lh = *reinterpret_cast<const LeafHead*>(&lf);
      lh.kind = uni<true>(lh.kind);
      lh.taskmap = uni<true>(lh.taskmap);
      lh.goal_offset = uni<true>(lh.goal_offset);
      float S[6], h[3];
      if (PT && lh.taskmap == RMP2_TASKMAP_FK_POINT) {
        // chain [FK(frame), TaskmapRelative4x4(rel), 4x4 -> position], one trip per pair (taskmap.py:79-99):
        //   r = R rel, x = p + r, xd = v + w x r, c = a + al x r + w x (w x r), J = Jacobian at x;  pulled back pair by
        //   pair (rmp.py:165-167 with B pairs on the batch axis); every lane of the robot forms the small vectors
        // (the pairs' fields from the caller's arrays -- or formed here, per control step, from the primitive table and the
        // leaf's link capsule: obs.link_caps, as in rmp2_quad.h; that form can roll out)
        const bool from_table = obs.link_caps != nullptr;  // (wave-uniform)
        const int lidx = uni<true>(lf.index);
        const int pb = from_table ? 0 : obs.pair_begin[lidx];
        const int cnt = from_table ? obs.n_spheres : obs.pair_begin[lidx + 1] - pb;
        const size_t pbase = (size_t)(live ? robot : 0) * obs.n_pairs + pb;
        const float4* t4 = reinterpret_cast<const float4*>(TW + k * 12);
        const float4 r0 = t4[0], r1 = t4[1];
        const float Rw[9] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, tp.x};
        const float4* wa4 = reinterpret_cast<const float4*>(WAb + g * hex_pad(n_ops * 8) + k * 8);
        const float4 w0 = wa4[0], w1 = wa4[1];
        const float Wf[3] = {w0.x, w0.y, w0.z}, ALf[3] = {w0.w, w1.x, w1.y};
        float LA[3] = {0.f, 0.f, 0.f}, LD[3] = {0.f, 0.f, 0.f}, lrad = 0.f, laa = 0.f, inv_laa = 0.f;
        if (from_table) {
          const float* lc = obs.link_caps + 8 * uni<true>(lf.dist_ordinal);
          lrad = lc[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            LA[c] = P3[c] + Rw[3 * c] * lc[0] + Rw[3 * c + 1] * lc[1] + Rw[3 * c + 2] * lc[2];
            LD[c] = Rw[3 * c] * (lc[4] - lc[0]) + Rw[3 * c + 1] * (lc[5] - lc[1]) + Rw[3 * c + 2] * (lc[6] - lc[2]);
          }
          laa = dot3(LD, LD);
          inv_laa = laa > 0.f ? 1.0f / laa : 0.f;
        }
#pragma nounroll
        for (int trip = 0; trip < cnt; ++trip) {
          float r[3], nv[3], dd, pt[3], t1[3], t2[3], xdp[3], cp[3];
          if (from_table) {
            const float4* rec = reinterpret_cast<const float4*>(step_table) + (obs.capsule ? 2 * trip : trip);
            const float4 ca = rec[0];
            const float4 cb = obs.capsule ? rec[1] : ca;
            link_pair_fields(LA, LD, laa, inv_laa, lrad, ca, cb, P3, r, nv, dd);
          } else {
            const float* rel = obs.p_link + (pbase + trip) * 3;
            const float* nvp = obs.p_obs + (pbase + trip) * 3;
            dd = obs.dist[pbase + trip];
            nv[0] = nvp[0], nv[1] = nvp[1], nv[2] = nvp[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) r[c] = Rw[3 * c] * rel[0] + Rw[3 * c + 1] * rel[1] + Rw[3 * c + 2] * rel[2];
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) pt[c] = P3[c] + r[c];
          cross3(Wf, r, t1);
#pragma unroll
          for (int c = 0; c < 3; ++c) xdp[c] = V3[c] + t1[c];
          cross3(Wf, t1, t2);
          cross3(ALf, r, t1);
#pragma unroll
          for (int c = 0; c < 3; ++c) cp[c] = A3[c] + t1[c] + t2[c];
          float xdd[3], wgt;
          leaf_collision_avoidance(lh.P, dd, nv, xdp, xdd, wgt);
          S[0] = S[3] = S[5] = wgt;
          S[1] = S[2] = S[4] = 0.f;
#pragma unroll
          for (int c = 0; c < 3; ++c) h[c] = wgt * (xdd[c] - cp[c]);
          hex_sync();  // every lane has taken its copy of the previous columns
          set_cols(pt);
          pull_back(S, h);
        }
        hex_sync();
        set_cols(P3);  // back to the frame origin for the leaves that follow
        continue;
      } else if (lh.taskmap == RMP2_TASKMAP_FK_POSITION) {
        float gl[3], xdd[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) gl[c] = my_goal[lh.goal_offset + c];
        if (lh.kind == RMP2_LEAF_TARGET_ATTRACTOR)
          target_attractor_fast(lh.P, P3, V3, gl, xdd, S);
        else
          leaf_target_policy3(lh.P, P3, V3, gl, xdd, S);
        const float e[3] = {xdd[0] - A3[0], xdd[1] - A3[1], xdd[2] - A3[2]};
        h[0] = S[0] * e[0] + S[1] * e[1] + S[2] * e[2];
        h[1] = S[1] * e[0] + S[3] * e[1] + S[4] * e[2];
        h[2] = S[2] * e[0] + S[4] * e[1] + S[5] * e[2];
      } else {
        // distance leaf: this lane takes pairs b = s, s + 16, ...; S and h are summed over the 16 lanes
#pragma unroll
        for (int c = 0; c < 6; ++c) S[c] = 0.f;
        h[0] = h[1] = h[2] = 0.f;
        const float IP[6] = {lf.vb[0], lf.vb[1], lf.vb[2], lf.vb[3], lf.vb[4], lf.vb[5]};
        if (obs.mode == RMP2_OBS_SHARED_SPHERES) {
          if (spheres_in_lds && !CAP)
            pair_loop_culled<false, kHex>(sph_lds_base, n_sph_lds, nullptr, obs.n_spheres, obs.n_spheres, s, P3, V3, A3, lh.P,
                                          IP, S, h);
          else if (spheres_in_lds)
            pair_loop<kPairsSharedLds, CAP, kHex>(sph_lds_base, nullptr, nullptr, nullptr, obs.n_spheres, obs.n_spheres, s,
                                                  P3, V3, A3, lh.P, IP, S, h, obs.cylinder != 0);
          else
            pair_loop<kPairsSharedGlobal, CAP, kHex>(step_table, nullptr, nullptr, nullptr, obs.n_spheres, obs.n_spheres,
                                                     s, P3, V3, A3, lh.P, IP, S, h, obs.cylinder != 0);
        } else if (obs.mode == RMP2_OBS_EXPLICIT_PAIRS) {
          const int lidx = uni<true>(lf.index);
          const int pb = obs.pair_begin[lidx];
          const int count = obs.pair_begin[lidx + 1] - pb;
          const size_t base = ((size_t)(live ? robot : 0) * obs.n_pairs + pb) * 3;
          pair_loop<kPairsExplicit, false, kHex>(nullptr, obs.p_link + base, obs.p_obs + base, nullptr, count, count, s, P3,
                                                 V3, A3, lh.P, IP, S, h);
        } else if (use_member) {  // (wave-uniform) ragged list as a membership mask: the dense loop, masked
          pair_loop_culled<false, kHex, false, true>(sph_lds_base, n_sph_lds, nullptr, obs.n_spheres, obs.n_spheres, s, P3, V3, A3,
                                                     lh.P, IP, S, h, nullptr, member_lo, member_hi);
        } else {
          const int b0 = obs.csr_offset[live ? robot : 0];
          const int count = live ? obs.csr_offset[robot + 1] - b0 : 0;
          int max_count = count;
#pragma unroll
          for (int o = 32; o >= kHex; o >>= 1) max_count = max(max_count, __shfl_xor(max_count, o));
          if (spheres_in_lds && !CAP)
            pair_loop_culled<true, kHex>(sph_lds_base, n_sph_lds, obs.csr_index + b0, count, max_count, s, P3, V3, A3, lh.P,
                                         IP, S, h);
          else if (spheres_in_lds)
            pair_loop<kPairsRaggedLds, CAP, kHex>(sph_lds_base, nullptr, nullptr, obs.csr_index + b0, count, max_count, s,
                                                  P3, V3, A3, lh.P, IP, S, h, obs.cylinder != 0);
          else
            pair_loop<kPairsRaggedGlobal, CAP, kHex>(step_table, nullptr, nullptr, obs.csr_index + b0, count, max_count, s,
                                                     P3, V3, A3, lh.P, IP, S, h, obs.cylinder != 0);
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) S[c] = hex_sum(S[c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) h[c] = hex_sum(h[c]);
      }
      pull_back(S, h);
    }
    hex_sync();  // COL is rewritten for the next frame
  }
  RMP2_LSTAMP();  // 3: FK leaves done

  // ---- identity-task-map leaves: x = q, xd = qd, J = I (taskmap.py:13-20); lane s owns row s -------------
  {
    const bool row_ok = s < n_dof;
    const int ii = (s < N) ? s : 0;
    const float qi_ = my_q[ii], qdi = (s < N) ? my_qd[ii] : 0.f;
    for (int li = 0; li < n_id; ++li) {
      const DevLeaf& lfr = s_leaves[hdr.n_fk + li];  // identity-map leaves follow the FK leaves
      const LeafHead lh = *reinterpret_cast<const LeafHead*>(&lfr);
      const int kind = uni<true>(lh.kind), goal_offset = uni<true>(lh.goal_offset);
      const float* P = lh.P;
      const float va_i = lfr.va[ii], vb_i = lfr.vb[ii];
      if (kind == RMP2_LEAF_JOINT_DAMPING || kind == RMP2_LEAF_CSPACE_BIASING || kind == RMP2_LEAF_CONFIG_SPACE_BIASING) {
        // diagonal metrics m * I:  A_ss += m, f_s += m * xdd_s
        float mdiag, acc;
        if (kind == RMP2_LEAF_JOINT_DAMPING) {  // rmp2.py:127-137
          const float s2 = hex_sum(qdi * qdi);
          const float nrm = s2 > 0.f ? s2 * rsq1(s2) : 0.f;
          mdiag = P[1] * nrm + P[2];
          acc = -(P[0] * nrm) * qdi;
        } else if (kind == RMP2_LEAF_CSPACE_BIASING) {  // rmp2.py:212-226
          const float e = (s < N) ? qi_ - va_i : 0.f;
          const float nrm = sqrtf(hex_sum(e * e));
          mdiag = P[0] + P[4];
          const float pos = (nrm < P[3]) ? (-e * P[1]) : (-P[3] * (e / nrm) * P[1]);
          acc = pos + (-P[2] * qdi);
        } else {  // rmp.py:330-347
          mdiag = P[2];
          acc = P[0] * (va_i - qi_) - P[1] * qdi;
        }
        if (s < N) fv += (double)(mdiag * acc);
        const double dm = (double)mdiag;
#pragma unroll
        for (int j = 0; j < N; ++j) A[j] += (s == j) ? dm : 0.0;
      } else {
        // dense metrics  A_ij = cw_j * w * (beta zeta_i zeta_j + (1 - beta) delta_ij): every lane forms the
        // (zeta, xdd, cw) of ITS dof, the triples are exchanged through LDS
        float zeta_i = 0.f, xdd_i = 0.f, cw_i = 0.f, beta, wsc;
        if (kind == RMP2_LEAF_JOINT_VELOCITY_CAP) {
          // rmp2.py:100-112: metric = w / (1 - diag(ratio^2)) on the FULL matrix (quirk Q4)
          const float cutoff = P[0] - P[1];
          const float dv = fabsf(qdi) - cutoff;
          const float sgn = (qdi > 0.f) ? 1.f : (qdi < 0.f ? -1.f : 0.f);
          const float acc = -fabsf(P[2] * dv) * sgn;
          xdd_i = (fabsf(qdi) < cutoff) ? 0.f : acc;
          const float ratio = fminf(dv, P[1] - 1e-6f) / P[1];
          zeta_i = P[3] / (1.0f - ratio * ratio);  // diagonal entry
          cw_i = P[3] / 1.0f;                      // off-diagonal entry
          beta = 0.f;
          wsc = 0.f;
        } else if (kind == RMP2_LEAF_JOINT_LIMIT_AVOIDANCE) {
          // rmp.py:357-382; A = w * H broadcasts over the LAST axis: column scaling (quirk Q2)
          const float rr_ = 0.15f;
          const float c2 = (float)(-3.0 / (0.15 * 0.15)), c3 = (float)(2.0 / (0.15 * 0.15 * 0.15));
          const float iqd_max = (float)(60.0 / (20.0 * (2.0 * 3.14159265358979323846)));
          const float irange = rcp1(vb_i - va_i);
          const float du = (vb_i - qi_) * irange;
          const float dl = (qi_ - va_i) * irange;
          const float d = fminf(du, dl);
          const float spline = c3 * (d * d * d) + c2 * (d * d) + 0.f * d + 1.0f;
          cw_i = row_ok ? (d > rr_ ? 0.f : spline) : 0.f;
          const float zraw = row_ok ? qdi * iqd_max : 0.f;
          const float s2 = hex_sum(zraw * zraw);
          xdd_i = -P[0] * qi_ - P[1] * qdi;
          const float nrm = s2 > 0.f ? s2 * rsq1(s2) : 0.f;
          // soft norm h = |v| + (1/c) log(1 + exp(-2 c |v|)), c = 5   (helper/rmp_helper.py:62-65)
          const float hh = nrm + 0.2f * (0.693147182464599609375f * __builtin_amdgcn_logf(1.0f + exp1(-10.0f * nrm)));
          zeta_i = zraw * rcp1(hh);
          beta = 0.9f;
          wsc = 1.0f;
        } else {
          // TargetPolicy on the identity map, rmp.py:241-260 (goal is an n-vector)
          const float alpha = P[0], beta_d = P[1], c = P[2];
          const float v_i = row_ok ? my_goal[goal_offset + ii] - qi_ : 0.f;
          const float vn = sqrtf(hex_sum(v_i * v_i));
          const float hq = vn + c * logf(1.0f + expf(-2.0f * c * vn));
          const float inv_h = 1.0f / hq;
          xdd_i = row_ok ? alpha * (inv_h * v_i) - beta_d * qdi : 0.f;
          const float fn = sqrtf(hex_sum(xdd_i * xdd_i));
          const float hs = fn + 1.0f / c * logf(1.0f + expf(-2.0f * c * fn));
          zeta_i = xdd_i / hs;
          cw_i = 1.0f;
          beta = 1.0f - expf(-0.5f * (vn * vn) / 1.0f);
          wsc = expf(-vn / 3.0f);
        }
        if (!row_ok) zeta_i = xdd_i = cw_i = 0.f;
        const float omb = 1.0f - beta;
        float4 xch[N];  // all triples first, then one branch-free loop per leaf kind
        if (RMP2_HEX_DPP_PIVOTS) {  // (dof j's triple from lane j of the robot's DPP row: no LDS round trip)
          hex_static_for(std::make_integer_sequence<int, N>{}, [&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            xch[j] = make_float4(hex_row_bcastf<j>(zeta_i), hex_row_bcastf<j>(xdd_i), hex_row_bcastf<j>(cw_i), 0.f);
          });
        } else {
          XCH[s] = make_float4(zeta_i, xdd_i, cw_i, 0.f);
          hex_sync();
#pragma unroll
          for (int j = 0; j < N; ++j) xch[j] = XCH[j];  // padded dofs hold (0, 0, 0): their columns come out as exact zeros
        }
        float fi = 0.f;
        if (kind == RMP2_LEAF_JOINT_VELOCITY_CAP) {
#pragma unroll
          for (int j = 0; j < N; ++j) {
            float a = (j == s) ? zeta_i : xch[j].z;
            a = row_ok ? a : 0.f;
            A[j] += (double)a;
            fi = fmaf(a, xch[j].y, fi);
          }
        } else if (kind == RMP2_LEAF_JOINT_LIMIT_AVOIDANCE) {
#pragma unroll
          for (int j = 0; j < N; ++j) {
            const float Hij = beta * (zeta_i * xch[j].x) + ((j == s) ? omb : 0.f);
            float a = xch[j].z * Hij;
            a = row_ok ? a : 0.f;
            A[j] += (double)a;
            fi = fmaf(a, xch[j].y, fi);
          }
        } else {
#pragma unroll
          for (int j = 0; j < N; ++j) {
            const float Hij = beta * (zeta_i * xch[j].x) + ((j == s) ? omb : 0.f);
            float a = wsc * Hij;
            a = row_ok ? a : 0.f;
            A[j] += (double)a;
            fi = fmaf(a, xch[j].y, fi);
          }
        }
        fv += (double)fi;
        if (!RMP2_HEX_DPP_PIVOTS) hex_sync();  // XCH is rewritten by the next dense leaf
      }
    }
  }
  RMP2_LSTAMP();  // 4: identity leaves done

  // a dof whose state is not finite (after the quarantine: its velocity) makes the robot's system non-finite by construction:
  // every resolve then answers NaN with RMP2_STATUS_NONFINITE, also where no leaf would have carried the value into the system
  // (rmp2_quad.h, same place; tools/fuzz_parity.py seed 504944)
  if (s < N) fv = (fabsf(my_qd[s]) < 3.0e38f) ? fv : (double)__builtin_nanf("");

  // padding dofs of the template: identity rows so that they resolve to qdd = 0
#pragma unroll
  for (int j = 0; j < N; ++j)
    if (s >= n_dof && j == s) A[j] = 1.0;
  // keep the untouched system for the careful path
  double* const SYS = reinterpret_cast<double*>(&wl[HexLds<N>::kSys + g * HexLds<N>::kSysStride]);
  if (s < N) {
#pragma unroll
    for (int j = 0; j < N; ++j) SYS[s * (N + 1) + j] = A[j];
    SYS[s * (N + 1) + N] = fv;
  }
  if (hdr.strict == 3 || hdr.strict == 5) {  // (wave-uniform) solve = PINV on a symmetric set (3: certified from the LDL^T form;
                                             //  5: the pseudo-inverse on every robot -- the same stored system, RMP2_STRICT_CERTIFY=0)
    // This mapping accumulates FULL rows: A[s][j] = (S c_s) . c_j and A[j][s] = (S c_j) . c_s are rounded independently, equal
    // only to ~1e-7 |M| -- while the symmetric certificate below bounds sigma_min of M = U^T D^-1 U, i.e. it assumes the matrix it
    // eliminates IS symmetric (round-4 advisor finding: the slack of that assumption, 1e-7 sigma_max, is far above the
    // certificate's margin).  The upper triangle is the matrix (as in the quad mapping's symmetric form): every lane takes the
    // part of its row left of the diagonal from the other lanes' parked rows, and the parked (and exported) system is the
    // symmetric one the elimination then factors.
    hex_sync();
    if (s < N) {
#pragma unroll
      for (int j = 0; j < N; ++j) A[j] = (j < s) ? SYS[j * (N + 1) + s] : A[j];
    }
    hex_sync();  // every lane has read the rows above its own before they are rewritten
    if (s < N) {
#pragma unroll
      for (int j = 0; j < N; ++j) SYS[s * (N + 1) + j] = A[j];
    }
  }
  // optional outputs: the combined metric / force before the resolve
  if (live && s < n_dof) {
    if (out.M) {
#pragma unroll
      for (int j = 0; j < N; ++j)
        if (j < n_dof) out.M[((size_t)robot * n_dof + s) * n_dof + j] = A[j];
    }
    if (out.f) out.f[(size_t)robot * n_dof + s] = fv;
  }

  // ---- resolve: Gauss-Jordan in fp64 without row exchanges, one row per lane ------------------------------
  // (certification as lu_solve<N>, rmp2_solve.h: tiny pivot, multiplier growth, non-finite result -> careful path)
  bool flagged = true;
  // hdr.strict: 0 AUTO; 1 / 5 solve = PINV, the pseudo-inverse on every robot (no elimination; 5: symmetric set); 2 / 3 solve = PINV with the
  // round-4 certificate (3: symmetric set): the elimination's result stands for the robots it certifies as full rank above
  // TensorFlow's cutoff -- pinv = inv there, rmp2_quad.h has the derivation --, the careful path's Jacobi for the rest
  const bool certify = hdr.strict == 2 || hdr.strict == 3;
  if (!hdr.strict || certify) {
    // magnitudes are compared on the HIGH WORD of the doubles (monotone for non-negative values, NaN / Inf on top):
    // integer max at fp32 rate instead of dependent fp64 max chains.  scale = max |M_ij| rounded down to its high word.
    int scale_hi = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) scale_hi = max(scale_hi, __double2hiint(A[j]) & 0x7fffffff);
    scale_hi = hex_maxi(scale_hi);
    const double scale = __hiloint2double(scale_hi, 0);
    const double tiny = 1e-11 * scale;
    flagged = !(scale > 0.0) || !(scale < 1.7e308);
    int lmax_hi = 0, umax_hi = 0;
    double inv_own = 0.0;
    double Urow[N];  // (certificate) my row of U: pivot row s as it was when it was used -- columns >= s
    bool piv_pos = true;
#pragma unroll
    for (int j = 0; j < N; ++j) Urow[j] = 0.0;
    // Pivots are taken TWO per LDS exchange: lanes k and k + 1 publish their rows together, every lane applies step k
    // to the copy of row k + 1 itself (n - k fp64 FMAs, redundantly) and then eliminates both columns from its own row
    // -- half the dependent LDS round trips of one pivot per exchange.
    double* const ROW0 = reinterpret_cast<double*>(&wl[HexLds<N>::kRow + g * 2 * HexLds<N>::kRowStride]);
    double* const ROW1 = ROW0 + HexLds<N>::kRowStride / 2;
    hex_static_for(std::make_integer_sequence<int, N / 2>{}, [&](auto kc) __attribute__((always_inline)) {
      constexpr int k = 2 * decltype(kc)::value;
      constexpr int k1 = k + 1;
      double r0[N], r1[N], b0, b1;
      if (RMP2_HEX_DPP_PIVOTS) {
        // rows k and k + 1 as their owners hold them NOW (row k + 1 before step k: every lane applies it to its copy, below)
#pragma unroll
        for (int j = k; j < N; ++j) {
          r0[j] = hex_row_bcast<k>(A[j]);
          r1[j] = hex_row_bcast<k1>(A[j]);
        }
        b0 = hex_row_bcast<k>(fv);
        b1 = hex_row_bcast<k1>(fv);
      } else {
        if (s == k) {
#pragma unroll
          for (int j = k; j < N; ++j) ROW0[j] = A[j];
          ROW0[N] = fv;
        }
        if (s == k1) {
#pragma unroll
          for (int j = k; j < N; ++j) ROW1[j] = A[j];
          ROW1[N] = fv;
        }
        hex_sync();
#pragma unroll
        for (int j = k; j < N; ++j) {
          r0[j] = ROW0[j];
          r1[j] = ROW1[j];
        }
        b0 = ROW0[N];
        b1 = ROW1[N];
      }
      const bool bad0 = !(fabs(r0[k]) > tiny);
      const double inv0 = bad0 ? 0.0 : rcpd(r0[k]);
      const double m10 = r1[k] * inv0;  // step k applied to the published copy of row k + 1
#pragma unroll
      for (int j = k1; j < N; ++j) r1[j] = fma(-m10, r0[j], r1[j]);
      b1 = fma(-m10, b0, b1);
      const bool bad1 = !(fabs(r1[k1]) > tiny);
      const double inv1 = bad1 ? 0.0 : rcpd(r1[k1]);
      flagged = flagged || bad0 || bad1;
      inv_own = (s == k) ? inv0 : ((s == k1) ? inv1 : inv_own);
      if (certify) {
        piv_pos = piv_pos && r0[k] > 0.0 && r1[k1] > 0.0;
#pragma unroll
        for (int j = k; j < N; ++j) {
          Urow[j] = (s == k) ? r0[j] : Urow[j];
          if (j >= k1) Urow[j] = (s == k1) ? r1[j] : Urow[j];
          umax_hi = max(umax_hi, max(__double2hiint(r0[j]) & 0x7fffffff, j >= k1 ? (__double2hiint(r1[j]) & 0x7fffffff) : 0));
        }
      }
      const double l0 = (s != k) ? A[k] * inv0 : 0.0;
#pragma unroll
      for (int j = k1; j < N; ++j) A[j] = fma(-l0, r0[j], A[j]);
      fv = fma(-l0, b0, fv);
      const double l1 = (s != k1) ? A[k1] * inv1 : 0.0;
#pragma unroll
      for (int j = k1 + 1; j < N; ++j) A[j] = fma(-l1, r1[j], A[j]);
      fv = fma(-l1, b1, fv);
      lmax_hi = max(lmax_hi, max((s > k) ? (__double2hiint(l0) & 0x7fffffff) : 0, (s > k1) ? (__double2hiint(l1) & 0x7fffffff) : 0));
      if (!RMP2_HEX_DPP_PIVOTS) hex_sync();  // the next pair of pivot rows overwrites ROW0 / ROW1
    });
    if (N & 1) {  // the last pivot of an odd system on its own
      constexpr int k = N - 1;
      double pk, bk;
      if (RMP2_HEX_DPP_PIVOTS) {
        pk = hex_row_bcast<k>(A[k]);
        bk = hex_row_bcast<k>(fv);
      } else {
        if (s == k) {
          ROW0[k] = A[k];
          ROW0[N] = fv;
        }
        hex_sync();
        pk = ROW0[k], bk = ROW0[N];
      }
      const bool bad = !(fabs(pk) > tiny);
      flagged = flagged || bad;
      const double inv = bad ? 0.0 : rcpd(pk);
      inv_own = (s == k) ? inv : inv_own;
      if (certify) {
        piv_pos = piv_pos && pk > 0.0;
        Urow[k] = (s == k) ? pk : Urow[k];
        umax_hi = max(umax_hi, __double2hiint(pk) & 0x7fffffff);
      }
      const double l = (s != k) ? A[k] * inv : 0.0;
      fv = fma(-l, bk, fv);
      if (!RMP2_HEX_DPP_PIVOTS) hex_sync();
    }
    lmax_hi = hex_maxi(lmax_hi);
    flagged = flagged || !(lmax_hi <= __double2hiint(1e4));  // multiplier growth > 1e4 (NaN / Inf compare above it)
    if (certify) {  // (wave-uniform)
      // |U^-1|_inf (general sets) or |W^-1|_inf, W = |D|^-1/2 U (symmetric sets) by ONE back substitution on absolute values (the
      // comparison matrix of a triangular factor): lane i finishes t_i and hands it to the robot's other lanes, which retire
      // column i from the right-hand sides of their rows -- in units that make M / max|M| the matrix
      const bool symset = hdr.strict == 3;
      const double d_own = fabs(Urow[s < N ? s : 0]);
      const float rt = __builtin_sqrtf((float)(d_own / scale)) * 1.000001f + 1e-30f;  // an UPPER bound of sqrt(d / scale)
      double rhs = symset ? scale * (double)rt : scale;
      double tmax = 0.0;
      if (RMP2_HEX_DPP_PIVOTS) {  // (lane i's t_i to the robot's row by DPP: nine dependent steps without an LDS round trip each)
        hex_static_for(std::make_integer_sequence<int, N>{}, [&](auto ic) __attribute__((always_inline)) {
          constexpr int i = N - 1 - decltype(ic)::value;
          const double ti = hex_row_bcast<i>(rhs * fabs(inv_own));
          tmax = fmax(tmax, ti);
          rhs = (s < i) ? fma(fabs(Urow[i]), ti, rhs) : rhs;
        });
      } else {
#pragma unroll
        for (int i = N - 1; i >= 0; --i) {
          const double ti = __shfl(rhs * fabs(inv_own), i, kHex);
          tmax = fmax(tmax, ti);
          rhs = (s < i) ? fma(fabs(Urow[i]), ti, rhs) : rhs;
        }
      }
      const double lmax = __hiloint2double(lmax_hi, 0) * 1.0000002;  // (the high word rounds down: nudged back up)
      const double umax = __hiloint2double(hex_maxi(umax_hi), 0) * 1.0000002;
      constexpr double kEps = 2.220446049250313e-16;
      bool certified;
      if (symset) {
        certified = (piv_pos || (umax <= 4.0 * scale && lmax <= 64.0)) && tmax * tmax < 1.0 / (160.0 * (double)(N * N * N) * kEps);
      } else {
        double lb = 1.0;
#pragma unroll
        for (int i = 1; i < N; ++i) lb *= 1.0 + lmax;  // |L^-1|_inf <= (1 + l_max)^(n - 1)
        certified = umax <= 4.0 * scale && lmax <= 4.0 && tmax * lb < 1.0 / (160.0 * sqrt((double)N) * (double)(N * N) * kEps);
      }
      flagged = flagged || !certified;
    }
    const double x = fv * inv_own;
    const bool finite = (s >= n_dof) || (fabs(x) < 1.7e308);
    flagged = flagged || hex_any(!finite, g);
    if (s < n_dof) my_out[s] = (float)x;
  }
  RMP2_LSTAMP();  // 5: resolve done

  if (__any(flagged && live)) {
    hex_sync();  // SYS rows of all lanes are in LDS
    if (flagged && s == 0) {
      // ---- rare path: lane 0 of the robot runs the careful solve IN LDS, on the parked system (pivoted LU on a working
      // copy, then the pseudo-inverse in place): no private arrays, so the kernel carries no scratch segment
      double* const W = SYS;
      double* const T = SYS + N * (N + 1);
      double* const xp = T + N * (N + 1);
      if (!hdr.strict) status |= RMP2_STATUS_PINV_PATH;
      if (certify) status |= RMP2_STATUS_JACOBI;  // (not certified full rank: resolved by the Jacobi pseudo-inverse; diagnostic)
      bool finite_in = true;  // a metric / force with NaN or Inf resolves to NaN (as the reference's pinv does)
      for (int i = 0; i < N * (N + 1); ++i) finite_in = finite_in && (fabs(W[i]) < 1.7e308);
      if (!finite_in) {
        for (int i = 0; i < N; ++i) xp[i] = __builtin_nan("");
      } else if (hdr.strict || !lu_pivot_compact(W, T, N, xp)) {
        const int dropped = pinv_solve_compact(W, N, n_dof, xp);
        if (dropped) status |= RMP2_STATUS_RANK_DROP;
      }
      bool finite = true;
      for (int i = 0; i < n_dof; ++i) {
        finite = finite && (fabs(xp[i]) < 1.7e308);
        my_out[i] = (float)xp[i];
      }
      if (!finite) status |= RMP2_STATUS_NONFINITE;
    }
    hex_sync();
  }

  if (ROLL && ro.substeps > 0) {
    // plant: qdd held, semi-implicit Euler (qd += dt qdd; q += dt qd), lane s advances dof s of the LDS-resident state
    hex_sync();
    if (live && s < n_dof) {
      float* qw = &wl[HexLds<N>::kQ + g * N + s];
      float* qdw = &wl[HexLds<N>::kQd + g * N + s];
      const float acc = my_out[s];
      float qi2 = *qw, qdi2 = *qdw;
      for (int t = 0; t < ro.substeps; ++t) {
        qdi2 = fmaf(ro.dt, acc, qdi2);
        qi2 = fmaf(ro.dt, qdi2, qi2);
      }
      quarantine(qi2, qdi2);
      *qw = qi2;
      *qdw = qdi2;
    }
    hex_sync();
  }
  }  // control steps

  // ---- coalesced store of the BLOCK's qdd tile (16 robots, contiguous in HBM): whole cache lines instead of one
  // partial-line write per wave ----------------------------------------------------------------------------
  if (WAVES > 1)
    __syncthreads();  // every wave of the block, tail waves included
  else
    hex_sync();
  {
    const int rb = blockIdx.x * WAVES * kHexRobots;
    const int count = min(WAVES * kHexRobots, R - rb) * n_dof;
    float* go = out.qdd + (size_t)rb * n_dof;
    for (int i = tid; i < count; i += kWave * WAVES) go[i] = blk_out[i];
  }
  if (ROLL && ro.q_out) {  // the advanced state of this wave's robots
    const int count = n_live * n_dof;
    for (int i = lane; i < count; i += kWave) {
      const int rr = i / n_dof, jj = i - rr * n_dof;
      const float qdv = wl[HexLds<N>::kQd + rr * N + jj];
      ro.q_out[(size_t)r0 * n_dof + i] = (qdv != qdv) ? qdv : wl[HexLds<N>::kQ + rr * N + jj];  // (quarantined joints)
      ro.qd_out[(size_t)r0 * n_dof + i] = qdv;
    }
  }
  if (out.status && live && s == 0) out.status[robot] = status;
#ifdef RMP2_STAMPS
  RMP2_STAMP();  // 6: stored
  if (lane == 0 && out.M == nullptr && out.f != nullptr) {  // diagnostic convention: the stamps follow the f rows
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(out.f) + (size_t)R * n_dof + (size_t)blockIdx.x * 16;
    for (int i = 0; i < 8; ++i) dst[i] = i < st_n ? st_[i] : 0ull;
  }
#endif
}

}  // namespace rmp2
