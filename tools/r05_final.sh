cd /root/repo
timeout 1200 bash tools/profile_round.sh r05 2>&1 | tail -8
timeout 600 bash tools/traffic_round.sh r05 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/r05_bench_default_run.json 2> gpurun_out/r05_bench_default_run.err; tail -c 600 gpurun_out/r05_bench_default_run.json
timeout 900 bash tools/levels_round.sh > gpurun_out/r05_levels.txt 2>&1; tail -12 gpurun_out/r05_levels.txt
timeout 600 python tools/kind_probe.py > gpurun_out/r05_kind_probe.txt 2>/dev/null; tail -8 gpurun_out/r05_kind_probe.txt
timeout 600 bash tools/stream_shapes.sh > /dev/null 2>&1; cat gpurun_out/r05_stream_shapes.txt | grep "streams of" 
