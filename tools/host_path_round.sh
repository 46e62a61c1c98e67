#!/bin/bash
# What the host link gives and what the host-buffer path makes of it: three runs each, kept whole.
# usage (GPU box): bash tools/host_path_round.sh > gpurun_out/r05_host_path.txt
echo "# bash tools/host_path_round.sh  (three processes of tools/pcie_probe.py, then three of tools/e2e_probe.py 1024)"
for i in 1 2 3; do echo "== pcie_probe run $i"; python tools/pcie_probe.py 2>/dev/null; done
for i in 1 2 3; do echo "== e2e_probe run $i"; python tools/e2e_probe.py 1024 2>/dev/null; done
