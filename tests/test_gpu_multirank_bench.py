"""GPU: the N > 1 path of bench.py end to end on a one-GPU box -- two ranks share cuda:0 over gloo (control flow, bucket
bookkeeping, collective count; the RCCL transport itself needs a multi-GPU node, which only the driver has).  VERDICT r02 #8:
a step issues <= 4 bucket all-reduces + <= 2 small collectives.  VERDICT r03 #7: the line carries the exposed communication
time and the per-bucket latency, and takes a stand-in for the backbone's backward (--backbone-ms)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(900)
def test_bench_two_ranks_gloo_end_to_end():
    env = dict(os.environ, SEMIDETR_BENCH_SHARE_GPU="1", SEMIDETR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--backbone-ms", "2.0"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["images_per_gpu"] == 5
    c = d["collectives"]
    assert c["world_size"] == 2 and c["backend"] == "gloo"
    assert 1 <= c["per_step_all_reduce"] <= 4, c           # 240 MB arena in 64 MiB buckets
    assert c["per_step_other"] <= 2, c
    assert abs(c["all_reduce_mb_per_step"] - 240.0) < 1.0, c      # the whole fp32 gradient arena, once
    # what the end of backward waited for, and launch -> complete per bucket, are IN the line
    assert c["exposed_ms_per_step"] is not None and c["exposed_ms_per_step"] >= 0.0, c
    assert len(c["bucket_ready_to_done_ms"]) == round(c["per_step_all_reduce"]) and all(x > 0 for x in c["bucket_ready_to_done_ms"]), c
    assert d["config"]["backbone_ms"] == 2.0
    assert "roofline" in d and "cpu_baseline" not in d and "microbench" not in d      # single-GPU extras stay out at N > 1
