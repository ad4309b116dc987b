#!/bin/bash
# round 6, GPU call 6: the whole -m gpu suite on HEAD, the round's profile passes, the bench line
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r6_run6
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1
tail -4 $O/gpu_tests.txt
bash tools/profile_round.sh r6 > $O/profile_round.txt 2>&1
tail -3 $O/profile_round.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cut -c1-400 $O/bench.json; tail -3 $O/bench.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
