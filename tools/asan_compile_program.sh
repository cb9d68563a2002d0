#!/bin/bash
# Host-side AddressSanitizer + UBSan build of the engine library, driven WITHOUT a GPU: rmp2_validate() runs the program
# compiler (compile_program: schedule, slots, pruning / folding, ancestor and dof tables, leaf bucketing) on thousands of
# random descriptors -- random tree robots up to the ABI's limits, random leaf sets, and deliberately broken descriptors.
# (GPU ASan is not available on this pool; this covers the host half of the library.  The kernel-launching translation units
# are replaced by tools/validate_only_stubs.hip: the build launches nothing.)
#   bash tools/asan_compile_program.sh [n_descriptors]
set -e
cd "$(dirname "$0")/.."
N=${1:-3000}
mkdir -p tools/diag
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fno-slp-vectorize -fsanitize=address,undefined -fno-omit-frame-pointer \
      -shared -fPIC -o tools/diag/librmp2_asan.so riemannian_motion_policies_amd/csrc/rmp2_hip.hip tools/validate_only_stubs.hip 2>&1 | grep -v "warning" | head -5 || true
ASAN_RT=$(hipcc -print-file-name=libclang_rt.asan-x86_64.so 2>/dev/null || true)
[ -f "$ASAN_RT" ] || ASAN_RT=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.asan-x86_64.so" | head -1)
echo "asan runtime: $ASAN_RT"
LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  RMP2_LIB=$PWD/tools/diag/librmp2_asan.so python3 tools/fuzz_validate.py $N
