# Round-4 evidence set, on the GPU box:  bash tools/collect_profiles_r4.sh <tag> [parts]   -> gpurun_out/<tag>/
#   parts (default all): stats traffic mfma bench configs cfgstats shards shardstats
# stats    kernel stats (rocprofv3 --kernel-trace --stats) of the default bench
# traffic  PMC L2-to-fabric traffic passes of the default bench (tools/pmc_traffic.sh) -> traffic.json
# mfma     matrix-pipe / VALU busy counters of the dense kernels
# bench    the default bench line (with cpu_baseline)
# configs  one bench line per BASELINE config (eager and hipGraph replay for the small ones)
# cfgstats kernel stats per BASELINE config (Cora / Pubmed / Arxiv shapes under hipGraph replay, pl10m bf16, Products bf16)
# shards   per-rank compute at the shard shapes of N = 4 / 8 ranks, both layer-0 plans (tools/shard_profile.py)
TAG=${1:-final}; PARTS=${2:-"stats traffic mfma bench configs cfgstats shards shardstats"}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$TAG
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
if has stats; then
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/$TAG/bench_under_rocprof.json 2> gpurun_out/$TAG/stats.err
fi
if has traffic; then
  bash tools/pmc_traffic.sh $TAG/traffic > /dev/null 2>&1
  python3 tools/make_traffic_json.py gpurun_out/$TAG/traffic gpurun_out/$TAG/traffic.json > gpurun_out/$TAG/traffic.txt
fi
if has mfma; then
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$TAG/mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/mfma.json 2> gpurun_out/$TAG/mfma.err
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$TAG/mfma_pubmed -- python3 bench.py --workload pubmed --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/mfma_pubmed.json 2> gpurun_out/$TAG/mfma_pubmed.err
fi
if has bench; then
  python3 bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
fi
if has configs; then
  : > gpurun_out/$TAG/configs.jsonl
  for w in cora pubmed arxiv; do
    python3 bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline --graph >> gpurun_out/$TAG/configs.jsonl 2>/dev/null
    python3 bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline >> gpurun_out/$TAG/configs.jsonl 2>/dev/null
  done
  python3 bench.py --workload products --dtype bf16 --no-cpu-baseline >> gpurun_out/$TAG/configs.jsonl 2>/dev/null
  python3 bench.py --workload pl10m --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline >> gpurun_out/$TAG/configs.jsonl 2>/dev/null
fi
if has cfgstats; then
  for w in cora pubmed arxiv; do
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats_$w -- python3 bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --graph > gpurun_out/$TAG/stats_$w.json 2> gpurun_out/$TAG/stats_$w.err
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats_pl10m_bf16 -- python3 bench.py --workload pl10m --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/$TAG/stats_pl10m_bf16.json 2> gpurun_out/$TAG/stats_pl10m_bf16.err
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats_products_bf16 -- python3 bench.py --workload products --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/$TAG/stats_products_bf16.json 2> gpurun_out/$TAG/stats_products_bf16.err
fi
if has shardstats; then      # per-kernel table of the P = 8 shard (which pull kernel, how long the fix-up kernels take)
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats_shard8 -- python3 tools/shard_profile.py --world 8 --rank 3 --exchange-layer0 > gpurun_out/$TAG/stats_shard8.json 2> gpurun_out/$TAG/stats_shard8.err
fi
if has shards; then
  : > gpurun_out/$TAG/shards.jsonl
  python3 tools/shard_profile.py --world 8 --rank 3 >> gpurun_out/$TAG/shards.jsonl 2>> gpurun_out/$TAG/shards.err
  python3 tools/shard_profile.py --world 8 --rank 3 --exchange-layer0 >> gpurun_out/$TAG/shards.jsonl 2>> gpurun_out/$TAG/shards.err
  python3 tools/shard_profile.py --world 4 --rank 1 >> gpurun_out/$TAG/shards.jsonl 2>> gpurun_out/$TAG/shards.err
  python3 tools/shard_profile.py --world 4 --rank 1 --exchange-layer0 >> gpurun_out/$TAG/shards.jsonl 2>> gpurun_out/$TAG/shards.err
  python3 tools/shard_profile.py --world 2 --rank 1 >> gpurun_out/$TAG/shards.jsonl 2>> gpurun_out/$TAG/shards.err
  python3 tools/shard_profile.py --world 2 --rank 1 --exchange-layer0 >> gpurun_out/$TAG/shards.jsonl 2>> gpurun_out/$TAG/shards.err
  GAT_PULL_RUNS=0 python3 tools/shard_profile.py --world 8 --rank 3 --exchange-layer0 2>> gpurun_out/$TAG/shards.err | sed 's/^{/{"label": "GAT_PULL_RUNS=0 (list-per-group pull pass, round 3)", /' >> gpurun_out/$TAG/shards.jsonl
  python3 tools/shard_profile.py --workload pl10m --dtype bf16 --world 8 --rank 3 --steps 5 >> gpurun_out/$TAG/shards.jsonl 2>> gpurun_out/$TAG/shards.err
  python3 tools/shard_profile.py --workload pl10m --dtype bf16 --world 8 --rank 3 --steps 5 --exchange-layer0 >> gpurun_out/$TAG/shards.jsonl 2>> gpurun_out/$TAG/shards.err
fi
ls gpurun_out/$TAG
