#!/bin/bash
# round 6, GPU call 3: soak (with the blinker phase) on the pre-fix library and on HEAD; the VALU counter pass; the new bench line
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r6_run3
mkdir -p $O
( cd .r6_old && for r in 1 2; do timeout 300 python -m pytest tests/test_gpu_soak.py -x -q -s -k k19 2>&1 | tail -6; done ) > $O/soak_old.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_soak.py -x -q -s > $O/soak_new.txt 2>&1
bash tools/pmc_valu.sh r6 > $O/pmc_valu.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
for f in soak_old.txt soak_new.txt pmc_valu.txt bench.err; do echo "== $f"; tail -4 $O/$f; done
cut -c1-600 $O/bench.json
