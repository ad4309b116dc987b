#!/bin/bash
# round 6, session 3: closing evidence at the round's last library — shape sweeps with the three-coset quotient forced on every shape
# it applies to (against the oracle's provers), and the determinism soak over the stream pool / activity hold / three-coset defaults
cd "$(dirname "$0")/.."
O=gpurun_out/r6s3_final; mkdir -p $O
(OPTS=13=2 timeout 500 python tests/shape_sweep.py 7401 100 > $O/sweep_small_c3.txt 2>&1; echo rc $? >> $O/sweep_small_c3.txt)
(OPTS=13=2 timeout 700 python tests/shape_sweep.py 7402 20 mid > $O/sweep_mid_c3.txt 2>&1; echo rc $? >> $O/sweep_mid_c3.txt)
(timeout 400 python tools/soak.py 700 19 4 > $O/soak_k19.txt 2>&1; echo rc $? >> $O/soak_k19.txt)
(timeout 300 python tools/soak.py 1500 17 4 > $O/soak_k17.txt 2>&1; echo rc $? >> $O/soak_k17.txt)
(timeout 300 python tools/soak.py 120 17 2 4 > $O/soak_k17_lockstep.txt 2>&1; echo rc $? >> $O/soak_k17_lockstep.txt)
(timeout 300 python tools/soak.py 300 19 1 > $O/soak_k19_lone.txt 2>&1; echo rc $? >> $O/soak_k19_lone.txt)
for f in $O/*.txt; do echo "== $f"; tail -n 3 $f; done
