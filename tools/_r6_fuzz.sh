export MELD_DEV=1
commit=${1:-unknown}; out=gpurun_out/fuzz_r06; mkdir -p $out
{ echo "# commit $commit, $(date -u +%Y-%m-%dT%H:%MZ): tools/fuzz_partial.py 200 11 -- the graph without the test / test forced / principal frame / two-pass form, bit for bit"; timeout 1500 python tools/fuzz_partial.py 200 11 2>&1 | grep -v amdgpu.ids | grep -v "^ok" | tail -20; } > $out/fuzz_partial_long.txt
{ echo "# commit $commit, $(date -u +%Y-%m-%dT%H:%MZ): tools/fuzz_graph.py against the oracle, graph options drawn"; FUZZ_OPTIONS=1 FUZZ_N_MAX=12000 timeout 1500 python tools/fuzz_graph.py 150 ${FUZZ_SEED:-23} 2>&1 | grep -v amdgpu.ids | tail -160; } > $out/fuzz_long.txt
{ echo "# commit $commit: tools/stress_partial.py"; timeout 900 python tools/stress_partial.py 2>&1 | grep -v amdgpu.ids | tail -12; } > $out/stress_partial.txt
tail -3 $out/fuzz_partial_long.txt; tail -3 $out/fuzz_long.txt; tail -3 $out/stress_partial.txt
