#!/bin/bash
exec < /dev/null
# Round 3: the MFMA filler probe (tools/probes/mfma_filler_probe.hip) on the silicon.
# usage: gpurun --timeout 300 -- bash tools/gpu_r03_probe.sh [tag]
tag=${1:-r03k}
out=$PWD/gpurun_out
mkdir -p $out
hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/probes/mfma_filler_probe.hip -o /tmp/probe 2> $out/${tag}_probe_build.err || { tail -5 $out/${tag}_probe_build.err; exit 1; }
timeout 200 /tmp/probe > $out/${tag}_mfma_filler_probe.jsonl 2> $out/${tag}_probe.err
echo "probe exit $?"; cat $out/${tag}_mfma_filler_probe.jsonl | cut -c1-200
