#!/bin/bash
# tools/round_benches.sh <dir> -- the secondary measurements of a round in one go on the GPU box (each prints one JSON object):
# the default bench line, host entry / other lengths, long reads, mixed long reads, the structured library on a flat and on a deep
# taxonomy, the CLI end to end.  Copy what should be judged into profiles/.
D=${1:-gpurun_out/round}
mkdir -p $D
python3 bench.py > $D/bench_default.json 2> $D/bench_default.err || exit 1
timeout -k 10 500 python3 tools/bench_extras.py > $D/extras.json 2> $D/extras.err || exit 1
timeout -k 10 300 python3 tools/bench_long.py > $D/long.json 2> $D/long.err || exit 1
timeout -k 10 300 python3 tools/bench_long_mixed.py > $D/long_mixed.json 2> $D/long_mixed.err || exit 1
for c in 0 3; do CHAIN=$c PAD=1e10 timeout -k 10 400 python3 tools/bench_phylo.py 2> $D/phylo_chain$c.err | tail -1 > $D/phylo_chain$c.json || exit 1; done
R=10000000 timeout -k 10 800 python3 tools/bench_cli.py > $D/cli.json 2> $D/cli.err || exit 1
tail -c 600 $D/bench_default.json
