#!/bin/bash
# quick loop for k_inflate work (GPU box): the inflate tests, then the decompress leg of the headline bench
timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_inflate_spans.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['value'], d['decompress'])"
