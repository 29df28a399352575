# Evidence of one round, on the GPU box:  bash tools/collect_profiles.sh <tag>   -> gpurun_out/<tag>/
#   kernel stats (rocprofv3 --kernel-trace --stats) of the default bench, PMC HBM traffic passes, the default bench line,
#   and one line per BASELINE config.
TAG=${1:-final}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/$TAG/bench_under_rocprof.json 2> gpurun_out/$TAG/stats.err
bash tools/pmc_traffic.sh $TAG/traffic > /dev/null 2>&1
python3 tools/make_traffic_json.py gpurun_out/$TAG/traffic gpurun_out/$TAG/traffic.json > gpurun_out/$TAG/traffic.txt
python3 bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
: > gpurun_out/$TAG/configs.jsonl
for w in cora pubmed arxiv; do
  python3 bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline --graph >> gpurun_out/$TAG/configs.jsonl 2>/dev/null
  python3 bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline >> gpurun_out/$TAG/configs.jsonl 2>/dev/null
done
python3 bench.py --workload products --dtype bf16 --no-cpu-baseline >> gpurun_out/$TAG/configs.jsonl 2>/dev/null
python3 bench.py --workload pl10m --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline >> gpurun_out/$TAG/configs.jsonl 2>/dev/null
ls gpurun_out/$TAG
