#!/bin/bash
# On the GPU box: one PMC pass of a single-stream frame for a variant library.
# usage: tools/pmc_variant.sh <variant name or "current"> "<counters>" "<bench args>"
name="$1"; counters="$2"; args="$3"
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
cp raytracer_amd/librt_hip.so /tmp/librt_hip_orig.so
[ "$name" != current ] && cp variants/librt_hip_$name.so raytracer_amd/librt_hip.so
out=gpurun_out/pmc_$name
rm -rf $out
timeout -s KILL 120 rocprofv3 --pmc $counters --output-format csv -d $out -- python3 bench.py $args --no-cpu-baseline --no-host-frames --streams 1 > /dev/null 2>&1
rc=$?
cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so
csv=$(find $out -name "*counter_collection.csv" | head -1); python3 tools/pmc_summary.py "$csv" > gpurun_out/pmc_$name.txt 2>&1
rm -rf $out
exit $rc
