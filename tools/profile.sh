#!/bin/bash
# Reproduces the artefacts under profiles/ on an MI355X box (run from the repository root, e.g. through
#   gpurun --timeout 1500 -- 'bash tools/profile.sh gpurun_out/prof'
# then copy the summaries you want judged into profiles/).  Counters are collected in their own passes
# (rocprofv3 --pmc is never combined with trace domains other than the kernel trace).
set -e
OUT=${1:-gpurun_out/prof}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
python "$ROOT/bench.py" > "$ROOT/$OUT/bench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats" -o r -- python "$ROOT/bench.py" --steps 20 --warmup 5 --cpu-baseline none --extra-workloads none > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$ROOT/$OUT/pmc_fetch" -o r -- python "$ROOT/bench.py" --steps 3 --warmup 1 --cpu-baseline none --extra-workloads none > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$ROOT/$OUT/pmc_write" -o r -- python "$ROOT/bench.py" --steps 3 --warmup 1 --cpu-baseline none --extra-workloads none > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$ROOT/$OUT/pmc_sq" -o r -- python "$ROOT/bench.py" --steps 3 --warmup 1 --cpu-baseline none --extra-workloads none > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d "$ROOT/$OUT/pmc_sq2" -o r -- python "$ROOT/bench.py" --steps 3 --warmup 1 --cpu-baseline none --extra-workloads none > /dev/null 2>&1
head -12 "$ROOT/$OUT/stats/r_kernel_stats.csv" | cut -c1-160
