O=gpurun_out/r04/fine; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRTMODT_STAMP -DPP_FINE -o /tmp/pp_probe_fine tools/probes/pp_probe.hip 2>/dev/null || exit 1
for f in "4.m.0.cv1" "6.m.0.cv1" "22.s1 P3 "; do timeout -k 10 120 /tmp/pp_probe_fine "$f" 3 >> $O/fine.txt 2>&1 || exit 1; done
cat $O/fine.txt | cut -c1-150
