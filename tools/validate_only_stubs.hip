// Stand-ins for the kernel-launching translation units (rmp2_quad_tu.hip, rmp2_quad_pair_tu.hip, rmp2_hex_tu.hip) in the
// host-only sanitizer build of tools/asan_compile_program.sh: that build exercises rmp2_validate -- the descriptor checks and
// the program compiler of csrc/rmp2_hip.hip -- without a GPU, and launches nothing.  Never part of librmp2_hip.so.
#include "../riemannian_motion_policies_amd/csrc/rmp2_host.h"
namespace rmp2 {
#define RMP2_STUB_QUAD(NAME)                                                                                            \
  bool NAME(const rmp2_handle*, const float*, const float*, const float*, int, const ObsArgs&, const OutArgs&,          \
            const RolloutArgs&, int, hipStream_t) { return false; }
RMP2_STUB_QUAD(launch_quad_n2_s0)
RMP2_STUB_QUAD(launch_quad_n2_s1)
RMP2_STUB_QUAD(launch_quad_n2_s2)
RMP2_STUB_QUAD(launch_quad_n9_s0)
RMP2_STUB_QUAD(launch_quad_n9_s1)
RMP2_STUB_QUAD(launch_quad_n9_s2)
RMP2_STUB_QUAD(launch_hex_n2)
RMP2_STUB_QUAD(launch_hex_n9)
RMP2_STUB_QUAD(launch_hex_n16)
bool launch_quad_pair(const rmp2_handle*, const float*, const float*, const float*, int, const ObsArgs&, const OutArgs&, int,
                      const rmp2_handle*, const float*, const float*, const float*, int, const ObsArgs&, const OutArgs&, int,
                      hipStream_t) { return false; }
}  // namespace rmp2
