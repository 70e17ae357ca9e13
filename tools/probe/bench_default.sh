# usage (on the GPU box): bash tools/probe/bench_default.sh <tag> [bench args...]  -> gpurun_out/r06/bench_<tag>.json
tag=$1; shift
mkdir -p gpurun_out/r06
python3 bench.py "$@" > gpurun_out/r06/bench_$tag.json 2> gpurun_out/r06/bench_$tag.err
tail -3 gpurun_out/r06/bench_$tag.err
python3 - <<PY
import json
j = json.load(open('gpurun_out/r06/bench_$tag.json'))
print('value', j['value'], 'contract_complete', j.get('value_contract_complete'), 'bf16x3', j.get('value_bf16x3'), 'frac', (j.get('roofline') or {}).get('frac'))
print('rd_rm_k5000', j.get('value_rd_rm_k5000'), j.get('value_rd_rm_k5000_contract_complete'))
c = j['config']
print(json.dumps(c.get('rd_rm_k5000'), indent=1))
print(json.dumps(j.get('roofline_rd_rm'), indent=1))
for k in ('rd_rm_leg_pairs_per_s', 'rd_rm_leg_k5000_pairs_per_s', 'rd_rm_leg_k5000_sinkhorn_ms_per_pair', 'dropin_leg', 'yohoc_leg'):
    print(k, c.get(k))
PY
