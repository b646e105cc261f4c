#!/bin/bash
# The windowed two-kernel chain (NYQ_OPT_CHAIN_WINDOW) against the one-window chain, product library, one process per shape,
# variants interleaved inside it.  Writes one JSON line per shape to $1.
out=${1:-gpurun_out/r3_window_sweep.jsonl}
: > "$out"
for shape in "1024 256 mix 0,64,128" "1024 512 mix 0,16,32,64,128,256" "1024 512 real 0,16,32,64,128" "512 512 mix 0,16,32,64,128" "256 512 mix 0,16,32,64,128,256" "128 1024 mix 0,16,32,64,128,256,512"; do
    set -- $shape
    CHAIN_WINDOWS=$4 timeout -k 10 240 python tools/chain_time.py $1 $2 $3 >> "$out" || exit 1
done
