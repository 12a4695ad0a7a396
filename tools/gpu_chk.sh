set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for m in combsub sins; do python bench.py --model $m --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', round(d['ms_per_step'],4), 'fir avg_ms', round(d['roofline']['avg_ms'],4))"; done
cd /tmp
DDSP_HIP_ONE_STREAM=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pq -o pq -- python $R/bench.py --steps 20 --warmup 3 --prewarm-seconds 0.2 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/pq -name "*.db" | head -1) 2>&1 | head -11; rm -rf $R/gpurun_out/pq
