cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=/tmp/prof_tl; rm -rf $P; mkdir -p $P gpurun_out
rocprofv3 --kernel-trace --output-format csv -d $P/t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/tl_bench.log 2>&1
python profiles/summarize.py timeline $P/t gpurun_out/timeline.csv
tail -1 gpurun_out/timeline.csv
