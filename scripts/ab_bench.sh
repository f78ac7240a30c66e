#!/bin/bash
# A/B on one box: the in-tree library (A) against scripts/dbg/libliverrt_b.so (B), alternating; args: configs...
for rep in 1 2; do
  for c in "$@"; do
    a=$(python3 bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    b=$(LRT_LIBRARY=$PWD/scripts/dbg/libliverrt_b.so python3 bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$c rep $rep: A $a ms   B $b ms"
  done
done
