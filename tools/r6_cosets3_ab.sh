#!/bin/bash
# round 6, session 3: the three-coset route of the quotient (ZK_OPT_QUOTIENT_DOMAIN: 0 = three cosets where h has three pieces, 1 = the
# whole extended domain) — the proving server's k = 17 shape (EVM + GWC) with 1 and 4 pipelines, and the rows of bench_ecdsa.config
# with two or more advice columns, alternating on one box.  Usage: gpurun -- 'bash tools/r6_cosets3_ab.sh > gpurun_out/r6_cosets3_ab.txt 2>&1'
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for o in 13=1 13=0; do
    OPTS=$o python tools/inflight_k17.py 1 4 2>&1 | grep "proofs/s"
  done
done
for rep in 1 2; do
  for o in 13=1 13=0; do
    echo "bench_rows OPTS=$o"
    OPTS=$o ROWS=18,17,16,15,14,13,12,11 python tools/bench_rows.py 7 2>&1 | grep -v "^degree"
  done
done
