#!/bin/bash
# Developer check: the host side of libblinkyhip (script front-end, code generator, context / lens / multi-GPU host code) built with
# AddressSanitizer + UndefinedBehaviorSanitizer - device code untouched (-fno-gpu-sanitize; the .hip objects are taken from the normal build) -
# and the CPU test suite plus the front-end fuzz run against it (BLINKY_HIP_LIB picks the library; no GPU needed).
#   tools/sanitize_host.sh [pytest arguments, default: the whole CPU suite]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${BLINKY_SAN_DIR:-/tmp/blinky_san}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p "$OUT"
make -C "$ROOT/blinky_amd/csrc" ARCH=gfx950 >/dev/null
cd "$ROOT/blinky_amd/csrc"
for f in bk_api.cpp bk_lens.cpp bk_lua.cpp bk_emit.cpp bk_embed.cpp bk_comm.cpp; do
    [ "$OUT/$f.o" -nt "$f" ] && [ "$OUT/$f.o" -nt bk_lua.h ] && [ "$OUT/$f.o" -nt bk_internal.h ] && continue
    $HIPCC --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-gpu-sanitize \
        -fno-omit-frame-pointer -DBK_DEBUG_API=1 -w -c $f -o "$OUT/$f.o"
done
cp build/bk_apply.hip.o build/bk_apply_coop.hip.o build/bk_probe.hip.o "$OUT/"
$HIPCC --offload-arch=gfx950 -shared -fsanitize=address,undefined -fno-gpu-sanitize -o "$OUT/libblinkyhip.so" "$OUT"/*.o -lhiprtc -ldl -lpthread
RT=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan-x86_64.so" | head -1)
cd "$ROOT"
export ASAN_OPTIONS=detect_leaks=0:detect_odr_violation=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0
export LD_PRELOAD=$RT BLINKY_HIP_LIB="$OUT/libblinkyhip.so"
if [ $# -gt 0 ]; then python -m pytest -p no:cacheprovider "$@"; else python -m pytest -p no:cacheprovider tests -q -m "not gpu"; fi
python tests/fuzz_frontend.py 0 400
