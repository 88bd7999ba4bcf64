#!/bin/bash
# On the GPU box: kernel timelines of the last frame of the two mesh workloads (and their bench lines).
export TMPDIR=/tmp
out=gpurun_out/mesh_tl
rm -rf $out; mkdir -p $out
run() {  # name, bench args
  local name="$1"; shift
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $out/$name -o kt -- python3 bench.py "$@" --no-cpu-baseline --no-host-frames > $out/$name.json 2> $out/$name.err || return 1
  python3 tools/kernel_timeline.py $out/$name/kt_kernel_trace.csv > $out/$name.timeline.txt
  python3 tools/bench_line.py < $out/$name.json >> $out/lines.txt
}
run dragon --workload chapter15-dragon-1000x1000 && run bbox3 --workload bounding-box-1000x400 --scale 3 && run bbox1 --workload bounding-box-1000x400
cat $out/lines.txt
