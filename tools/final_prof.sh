set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r03h
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03h/trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-leg > $R/gpurun_out/r03h/bench_under_rocprof.json 2> $R/gpurun_out/r03h/trace.err
echo trace done
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r03h/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg > $R/gpurun_out/r03h/fetch.json 2> $R/gpurun_out/r03h/fetch.err
echo fetch done
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r03h/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg > $R/gpurun_out/r03h/write.json 2> $R/gpurun_out/r03h/write.err
echo write done
find $R/gpurun_out/r03h -name "*.csv" | head -20
