# per-kernel durations WITHOUT overlapping pipelines: tools/prof_solo.sh <tag>
#   <tag>_proof_k19_solo_kernel_stats.csv      bench.py --inflight 1 (one zk_prove at a time)
#   <tag>_proof_k19_lockstep8_kernel_stats.csv bench.py --inflight 1 --lockstep 8 (one context, eight proofs in lock-step)
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$tag
mkdir -p $O
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/solo -- python $R/bench.py --no-cpu-baseline --k17-steps 0 --steps 16 --inflight 1 > $O/solo.log 2>&1
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ls8 -- python $R/bench.py --no-cpu-baseline --k17-steps 0 --steps 16 --inflight 1 --lockstep 8 > $O/ls8.log 2>&1
cd $R
python tools/summarize_prof.py ${tag}_proof_k19_solo $(dirname $(find $O/solo -name "*kernel_stats.csv" | head -1))
python tools/summarize_prof.py ${tag}_proof_k19_lockstep8 $(dirname $(find $O/ls8 -name "*kernel_stats.csv" | head -1))
mkdir -p gpurun_out/${tag}_profiles && cp profiles/${tag}_proof_k19_solo* profiles/${tag}_proof_k19_lockstep8* gpurun_out/${tag}_profiles/
tail -1 $O/solo.log | cut -c1-200; tail -1 $O/ls8.log | cut -c1-200
rm -rf $O/solo $O/ls8
