#!/bin/bash
# usage: tools/exp.sh <tag> <env assignments...> -- <bench args...>
# runs bench.py with the given env and prints a one-line summary
tag=$1; shift
envs=()
while [ "$1" != "--" ]; do envs+=("$1"); shift; done
shift
out=gpurun_out/exp_$tag.json
env "${envs[@]}" timeout -k 10 500 python bench.py "$@" > $out 2> gpurun_out/exp_$tag.log || { echo "$tag FAILED"; tail -5 gpurun_out/exp_$tag.log; exit 1; }
python - "$tag" "$out" <<'PY'
import json,sys
tag,f=sys.argv[1:3]
d=json.loads(open(f).read().strip().splitlines()[-1])
r=d["roofline"]; ph=d["phases_ms_per_step"]
print(f"{tag:28s} seeds/s {d['value']:9.1f}  spmm_avg_ms {r['avg_launch_ms']:8.3f}  alg_GB/s {r['achieved']:7.1f}  G {d['config']['tile_seeds']} TG {d['config']['tile_group']}  spmm {ph['spmm']:.1f} seedrow {ph['seed_row']:.1f} span {ph['iterate_span']:.1f} rank {ph['rank']:.1f} wall {ph['call_wall']:.1f}" + (f"  cpu_ok ids={d['cpu_baseline']['gpu_topk_ids_identical']} bits={d['cpu_baseline']['gpu_scores_bitwise_equal']}" if 'cpu_baseline' in d else ""))
PY
