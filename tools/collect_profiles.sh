#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace statistics and PMC passes of bench.py, each pass in
# its own rocprofv3 run, written under gpurun_out/profiles_<tag>/.  Copy the summaries you want to
# keep into profiles/ afterwards (tools/pmc_summary.py condenses the counter CSVs).
# usage: tools/collect_profiles.sh <tag>
set -e
# every pass is one frame or a few; a pass that hangs is killed by its own time limit and ends the script
tag="${1:-r01}"
out="gpurun_out/profiles_$tag"
mkdir -p "$out"
export TMPDIR=/tmp
run() {  # name, rocprofv3 options..., then bench options after "--"
  local name="$1"; shift
  timeout -s KILL 300 rocprofv3 "$@" > "$out/$name.log" 2>&1
}
run kt_default --kernel-trace --stats --output-format csv -d "$out/kt_default" -o kt -- python3 bench.py --no-cpu-baseline --no-host-frames
run kt_streams1 --kernel-trace --stats --output-format csv -d "$out/kt_streams1" -o kt -- python3 bench.py --no-cpu-baseline --no-host-frames --streams 1
run pmc_sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$out/pmc_sq1" -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-frames --streams 1
run pmc_sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES --output-format csv -d "$out/pmc_sq2" -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-frames --streams 1
run pmc_fetch --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-frames --streams 1
run pmc_write --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-frames --streams 1
# the same two counters over a stream of known size at the record columns' access width (8 bytes per lane)
if [ -x tools/bin/calib_fetch ]; then
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/calib_fetch" -o pmc -- tools/bin/calib_fetch > "$out/calib_fetch.log" 2>&1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/calib_write" -o pmc -- tools/bin/calib_fetch > "$out/calib_write.log" 2>&1
fi
echo "profiles written to $out"
