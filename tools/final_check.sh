cd /root/repo
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -3
