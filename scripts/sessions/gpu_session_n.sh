#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py smoke 2>&1 | tail -2
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > gpurun_out/t_all.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/t_all.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["frac"], "build", d["roofline_kernel_build"]["achieved"])
print("parity", {k:(v["rel"] if isinstance(v,dict) else v) for k,v in d["parity"].items()})
print("16k", d["config"]["also_configs1_N16384"]["seconds_per_step"], "C4", d["config"]["also_C4"]["seconds_per_step"], d["config"]["also_C4"]["rank_per_level"])
c5=d["config"]["also_C5"]; print("C5", c5["compute_loglike_s"], c5["predict_var_s"], c5["grad_s"], c5["fused_nll_and_grad_s"])
print("public", d["public_api"]["seconds_per_step"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["seconds"])
PY
tail -3 gpurun_out/bench_default.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --size 8192 --backend gloo --share-gpu 2>&1 | tail -1 | cut -c1-900
