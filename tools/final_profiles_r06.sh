#!/bin/bash
# Round-6 refresh on the GPU box: traffic JSON + TA evidence (rocprofv3 --pmc passes), the full GPU suite, smoke, the default
# bench line (with its flavours), the rocprofv3 kernel statistics of the bench command, a 2-rank line (gloo, one GPU: the N > 1 keys).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout -k 5 1500 python tools/measure_traffic.py > gpurun_out/r06_traffic.log 2>&1; tail -2 gpurun_out/r06_traffic.log | cut -c1-200
[ -f gpurun_out/r06_pmc_traffic.json ] && cp gpurun_out/r06_pmc_traffic.json profiles/r06_pmc_traffic.json
timeout -k 5 900 bash tools/r06_fwd_ta_evidence.sh > gpurun_out/r06_ta.log 2>&1; tail -2 gpurun_out/r06_ta.log | cut -c1-200
cp gpurun_out/r06_fwd_enc_TA.json gpurun_out/r06_fwd_enc_TA.txt profiles/ 2>/dev/null
timeout -k 5 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r06_final_tests.log
cat gpurun_out/r06_final_tests.log
timeout -k 5 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 5 900 python bench.py > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_default.err
tail -2 gpurun_out/r06_bench_default.err
SEMIDETR_BENCH_SHARE_GPU=1 SEMIDETR_DIST_BACKEND=gloo timeout -k 5 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --backbone-ms 8 > gpurun_out/r06_bench_line_2ranks_gloo.json 2> gpurun_out/r06_bench_2ranks.err
tail -1 gpurun_out/r06_bench_line_2ranks_gloo.json | cut -c1-100
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r06_prof_bench
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_prof_bench -- python $R/bench.py --steps 10 --warmup 3 --no-micro --no-flavours --no-cpu-baseline > $R/gpurun_out/r06_prof_bench.json 2> $R/gpurun_out/r06_prof_bench.err
cd $R
python tools/summarize_prof.py stats gpurun_out/r06_prof_bench > gpurun_out/r06_bench_kernel_stats.txt
python tools/summarize_prof.py bygrid gpurun_out/r06_prof_bench > gpurun_out/r06_bench_kernel_by_grid.txt
head -26 gpurun_out/r06_bench_kernel_by_grid.txt
# (what goes back is capped at 64 MiB: the raw traces and counter dumps stay on the box)
rm -rf gpurun_out/traffic gpurun_out/pmcset_* gpurun_out/ta_stats gpurun_out/r06_prof_bench gpurun_out/abk_* gpurun_out/r06_prof_first
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_line.json").read().strip().splitlines()[-1])
print({k: round(d[k], 2) for k in ("value", "ms_per_step")}, "frac", round(d["roofline"]["frac"], 4), d["roofline"]["kernel"], d["roofline"]["kernel_symbols"], "traffic", d["roofline"]["traffic"])
print({k: round(v, 3) for k, v in d["breakdown_ms_per_step"].items()})
print({k: (round(v["ms_per_step"], 2), round(v["images_per_s"], 1)) for k, v in d["flavours"].items() if isinstance(v, dict)})
print({k: round(v, 3) for k, v in d.items() if k.startswith("microbench_cold") and isinstance(v, float)})
print("sup", d["supervised_dino_bs2"]["images_per_s"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["enc_fwd_s"])
try:
    c = json.loads(open("gpurun_out/r06_bench_line_2ranks_gloo.json").read().strip().splitlines()[-1])["collectives"]
    print("2 ranks (gloo, one GPU):", {k: c[k] for k in ("per_step_all_reduce", "exposed_ms_per_step", "bucket_ready_to_done_ms")})
except Exception as e:
    print("2-rank line missing", e)
PY
