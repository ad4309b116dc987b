# GPU clock / power while the bench runs: tools/clock_watch.sh (samples rocm-smi as fast as it answers during `bench.py --steps 400`;
# prints the samples taken while the GPU was clocked up)
python bench.py --no-cpu-baseline --steps 400 > gpurun_out/clock_bench.log 2>&1 &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|Power \(W\)|GPU use" | sed 's/.*: //' | tr '\n' ' '; echo
done > gpurun_out/clock_samples.txt
grep -vE "^\((1[0-9][0-9]|[0-9][0-9])Mhz" gpurun_out/clock_samples.txt | head -40
tail -1 gpurun_out/clock_bench.log | cut -c1-160
