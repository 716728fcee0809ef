#!/bin/bash
# Round-3 measurement session (run on the GPU box from the repo root): everything DESIGN.md section 7 quotes.
# Usage: bash tools/r03_measure.sh <commit>     -> gpurun_out/r3z/
set -u
C=${1:-unknown}
O=${GLC_MEASURE_OUT:-gpurun_out/r3z}
R=$(pwd)
mkdir -p $O
python tools/dump_d1_rows.py > $O/dump.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench_a.json 2> $O/bench_a.err
python bench.py --steps 20 --warmup 5 > $O/bench_b.json 2> $O/bench_b.err
timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_n2.json 2> $O/bench_n2.err
echo "bench done: $(date)"
for f in chord random zero; do timeout -k 10 200 build/k1_tune 4096 2 40 $f > $O/k1_tune_$f.txt 2>&1; done
timeout -k 10 120 build/k1_tune 4096 2 30 chord timeline > $O/k1_timeline.txt 2>&1
timeout -k 10 100 build/k1_tune 86 2 40 chord > $O/k1_tune_86.txt 2>&1
timeout -k 10 200 build/d1_tune > $O/d1_tune.txt 2>&1; echo "d1_tune rc $?" >> $O/d1_tune.txt
build/bridge_bench 4096 2 48000 > $O/bridge_cfg2.txt 2>&1
build/bridge_bench 86 2 44100 > $O/bridge_cfg1.txt 2>&1
build/encode_breakdown 4096 2 > $O/encode_breakdown_cfg2.txt 2>&1
build/encode_breakdown 86 2 > $O/encode_breakdown_cfg1.txt 2>&1
build/encode_breakdown 28125 2 > $O/encode_breakdown_10min.txt 2>&1
build/decode_breakdown > $O/decode_breakdown.txt 2>&1
echo "tools done: $(date)"
KMS=$(python3 -c "import json;print(json.loads(open('$O/bench_a.json').read().strip().splitlines()[-1])['roofline']['ms_per_launch'])")
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python3 $R/bench.py --steps 100 --warmup 5 --lean > $R/$O/prof_stdout.txt 2>&1
echo "kernel trace done: $(date)"
python3 $R/tools/pmc_k1.py --tag r03 --kernel-ms $KMS > $R/$O/pmc_k1_stdout.txt 2>&1
echo "pmc k1 done: $(date)"
python3 $R/tools/pmc_traffic.py --commit $C --tag r03 > $R/$O/pmc_traffic_stdout.txt 2>&1
echo "pmc traffic done: $(date)"
cd $R
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
ls $O
