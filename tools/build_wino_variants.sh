#!/bin/bash
# builds tools/bin/conv_wino_bench[_<suffix>] for the ablation variants given as "suffix:flags" arguments
# usage: tools/build_wino_variants.sh base: trace:-DPOCR_BF16X3_TRACE d1:-DPOCR_WINO_DBG=1 ...
cd "$(dirname "$0")/.." && mkdir -p tools/bin
pids=()
for v in "$@"; do
  name="${v%%:*}"; flags="${v#*:}"
  out=tools/bin/conv_wino_bench; [ "$name" != base ] && out="${out}_$name"
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $flags -o "$out" tools/conv_wino_bench.hip 2> "/tmp/wino_build_$name.log" &
  pids+=($!)
done
rc=0; for p in "${pids[@]}"; do wait "$p" || rc=1; done
for v in "$@"; do name="${v%%:*}"; [ -s "/tmp/wino_build_$name.log" ] && { echo "== $name"; head -20 "/tmp/wino_build_$name.log"; }; done
exit $rc
