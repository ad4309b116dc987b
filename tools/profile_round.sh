# Round profile on the GPU box: tools/profile_round.sh <tag>   (outputs under gpurun_out/<tag>/, summaries via tools/summarize_prof.py)
# Counter passes carry --kernel-trace only (no other trace domain), one counter per pass (FETCH_SIZE and WRITE_SIZE do not fit one pass).
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$tag
mkdir -p $O
BENCH="python $R/bench.py --no-cpu-baseline --k17-steps 0 --steps 20"
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- $BENCH > $O/bench_stats.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/bench_fetch -- $BENCH > $O/bench_fetch.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/bench_write -- $BENCH > $O/bench_write.log 2>&1
K21="python $R/tools/stress_k21.py 5"
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k21_stats -- $K21 > $O/k21_stats.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/k21_fetch -- $K21 --no-check > $O/k21_fetch.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/k21_write -- $K21 --no-check > $O/k21_write.log 2>&1
cd $R
python tools/summarize_prof.py ${tag}_proof_k19 $(dirname $(find $O/bench_stats -name "*kernel_stats.csv" | head -1)) $(dirname $(find $O/bench_fetch -name "*counter_collection.csv" | head -1)) $(dirname $(find $O/bench_write -name "*counter_collection.csv" | head -1))
python tools/summarize_prof.py ${tag}_k21_stress $(dirname $(find $O/k21_stats -name "*kernel_stats.csv" | head -1)) $(dirname $(find $O/k21_fetch -name "*counter_collection.csv" | head -1)) $(dirname $(find $O/k21_write -name "*counter_collection.csv" | head -1))
mkdir -p gpurun_out/${tag}_profiles && cp profiles/${tag}_* gpurun_out/${tag}_profiles/
tail -1 $O/bench_stats.log | cut -c1-300; tail -1 $O/k21_stats.log
rm -rf $O/bench_stats $O/bench_fetch $O/bench_write $O/k21_stats $O/k21_fetch $O/k21_write
