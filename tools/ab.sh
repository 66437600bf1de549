#!/bin/bash
# A/B timing of library builds inside ONE gpurun call (box-to-box variance is larger than most kernel changes):
#   gpurun -- 'bash tools/ab.sh 3 "" gpurun_exp/libA.so gpurun_exp/libB.so'
# usage: tools/ab.sh ROUNDS "extra bench.py args" lib1.so lib2.so ...   (libs are alternated ROUNDS times)
# Each line: library, ms/step, per-stage ms (HIP events inside the library, see bench.py).
R=$1; shift; ARGS=$1; shift
for i in $(seq "$R"); do
  for L in "$@"; do
    SPLATRASTER_LIB=$(realpath "$L") python bench.py --cpu-baseline none --steps 30 $ARGS | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('%-28s'%'$L', '%.4f'%d['ms_per_step'], ' '.join('%s=%.4f'%(k[:14],v) for k,v in s.items()))"
  done
done
