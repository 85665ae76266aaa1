#!/bin/bash
# dev: an A/B build of the library under extra hipcc flags -> tulip_amd/libtulip_hip_<name>.so (select it with TULIP_HIP_LIB)
# usage: [ONLY="swinw swin96"] tools/build_variant.sh <name> <flags...>     (ONLY: the sources the flags apply to; default all)
set -e; fail=0
NAME="$1"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OBJ="/tmp/tulip_objs_$NAME"; mkdir -p "$OBJ"
pids=()
for s in gemm norm attention elementwise tail prep evalpost swin96 expand swinw swind glue; do
  F=("$@"); if [ -n "$ONLY" ] && ! [[ " $ONLY " == *" $s "* ]]; then F=(); fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I "$ROOT/include" -I "$ROOT/tulip_amd/csrc" "${F[@]}" \
    -c "$ROOT/tulip_amd/csrc/$s.hip" -o "$OBJ/$s.o" & pids+=($!)
done
for p in "${pids[@]}"; do wait $p || fail=1; done; [ $fail = 0 ] || { echo "compile failed"; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tulip_amd/libtulip_hip_$NAME.so" "$OBJ"/*.o
echo "$ROOT/tulip_amd/libtulip_hip_$NAME.so"
