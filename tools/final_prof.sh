set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r02g
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02g/trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-leg > $R/gpurun_out/r02g/bench_under_rocprof.json 2> $R/gpurun_out/r02g/trace.err
echo trace done
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02g/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg > $R/gpurun_out/r02g/fetch.json 2> $R/gpurun_out/r02g/fetch.err
echo fetch done
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r02g/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg > $R/gpurun_out/r02g/write.json 2> $R/gpurun_out/r02g/write.err
echo write done
find $R/gpurun_out/r02g -name "*.csv" | head -20
