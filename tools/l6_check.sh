#!/bin/bash
# quick loop for level-6 tokenizer work (GPU box): tokenizer parity tests, then the headline compress leg
timeout 900 python -m pytest tests/test_gpu_compress.py -m gpu -x -q -k "tokenizer or bytes_match or runny" 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-decompress 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
