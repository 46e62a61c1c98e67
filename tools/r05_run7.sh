cd /root/repo
for i in 1 2 3 4 5; do echo "== e2e_probe run $i"; python tools/e2e_probe.py 1024 2>/dev/null; done > gpurun_out/r05_e2e_rect.txt 2>&1
cat gpurun_out/r05_e2e_rect.txt
