#!/bin/bash
# Repeat one GPU test N times and print each outcome (a flake hunt runs ONE process at a time, in sequence).  usage: flake_loop.sh <pytest -k expression> <n> [file]
for i in $(seq 1 $2); do
  timeout -k 10 300 python -m pytest ${3:-tests/test_gpu_ngcf.py} -q -m gpu -k "$1" 2>&1 | grep -E "passed|failed|^E   +assert [0-9.e-]+ <=" | head -3
done
