set -e
# The round's closing measurements on the FINAL sources: perf guards, kernel trace + stats, FETCH_SIZE and WRITE_SIZE in
# separate --pmc passes (MI355X_MICROARCH.md, HBM section).  usage (on the GPU box): bash tools/final_prof.sh r04z
TAG=${1:-r04z}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R && (timeout -k 10 280 python3 -m pytest tests -m perf -q > $O/perf_guards.log 2>&1; echo "pytest -m perf rc=$?" >> $O/perf_guards.log; tail -3 $O/perf_guards.log)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-leg > $O/bench_under_rocprof.json 2> $O/trace.err
echo trace done
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg > $O/fetch.json 2> $O/fetch.err
echo fetch done
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg > $O/write.json 2> $O/write.err
echo write done
F=$(find $O/fetch -name "*counter_collection.csv" | head -1); W=$(find $O/write -name "*counter_collection.csv" | head -1)
cd $R && python3 tools/pmc_summary.py $F $W 1048576 $O/traffic.json > /dev/null && echo summary done
S=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp $S $O/kernel_stats.csv
rm -rf $O/trace $O/fetch/*/*kernel_trace* 2>/dev/null || true
ls $O
