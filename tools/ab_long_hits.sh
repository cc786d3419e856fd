#!/bin/bash
# The four ways through a batch of long unpaired reads (tools/bench_long_hits.py): hit lists or not, segment or wave kernel.
export SLK_BENCH_BASES=${SLK_BENCH_BASES:-1e9}
mkdir -p gpurun_out
SLK_SEG_HITS=1 timeout -k 10 280 python tools/bench_long_hits.py 2>gpurun_out/lh1.err && \
timeout -k 10 280 python tools/bench_long_hits.py 2>gpurun_out/lh2.err && \
SLK_BENCH_NO_HITS=1 timeout -k 10 280 python tools/bench_long_hits.py 2>gpurun_out/lh3.err && \
SLK_BENCH_NO_HITS=1 SLK_SEG_MIN_LEN=100000000 timeout -k 10 280 python tools/bench_long_hits.py 2>gpurun_out/lh4.err
