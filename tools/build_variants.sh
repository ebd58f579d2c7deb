#!/bin/bash
# Builds libgpt_hip.so and named A/B variants of one source file ON THE BOX, so that every library of a session comes
# from the sources of the same snapshot.  usage: tools/build_variants.sh [name:file:"-DFLAG ..."] ...
#   e.g. tools/build_variants.sh kv_old:gpt_predict:"-DGPT_DIAG_RING=2" vtrace:gpt_predict:"-DGPT_VAR_TRACE"
# -> csrc/build/libgpt_<name>.so (select with GPT_HIP_LIB)
set -u
C=gaussian_process_transportation_amd/csrc
make -C $C -j8 all > $C/build/make_all.log 2>&1 || { tail -20 $C/build/make_all.log; exit 1; }
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -fno-gpu-rdc"
pids=()
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; file=${rest%%:*}; defs=${rest#*:}
  (
    /opt/rocm/bin/hipcc $FLAGS $defs -c $C/$file.hip -o $C/build/${file}_$name.o || exit 1
    objs=""
    for f in gpt_api gpt_fit gpt_predict; do
      if [ "$f" = "$file" ]; then objs="$objs $C/build/${file}_$name.o"; else objs="$objs $C/build/$f.o"; fi
    done
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $C/build/libgpt_$name.so $objs
  ) > $C/build/variant_$name.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
ls -la $C/../libgpt_hip.so $C/build/libgpt_*.so | awk '{print $5, $9}'
exit $rc
