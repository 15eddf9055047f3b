"""bench.py checks the run it times: the exact call pattern of the driver's invocation (`--steps 20 --warmup 5`: warm-up,
serialised kernel table, 64 ramp steps, the timed lyra_hip_run_steps_dev region, latency steps, one context, input ring
cycled) is replayed by the CPU oracle and the GPU's final packets / PCM must be bit-equal -- for every BASELINE config the
bench offers (lyra_benchmark_lib.cc:121-160 is the loop being replaced).  A negative control shows the check has teeth."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(argv):
    import bench
    args = bench.parse(argv + ["--no-cpu-baseline", "--verify-streams", "64"])
    wl = bench.resolve_workload(args, 1)
    sh = bench.Shard(0, 0, wl, args)
    res = bench.run_shard(sh, args, wl, lambda: None)
    return bench, sh, args, wl, res


@pytest.mark.parametrize("cfg", [3, 2, 5, 4])
def test_bench_driver_form_is_verified_against_the_oracle(cfg):
    bench, sh, args, wl, res = _run(["--config", str(cfg), "--steps", "20", "--warmup", "5", "--latency-steps", "7"])
    v = res["verify"]
    assert v["verified"] is True, v
    assert v["streams"] == 64
    n_regions = 2 if cfg == 4 else 1          # config #4 times lyra_hip_generate_dev and lyra_hip_decode_dev
    assert v["steps_replayed"] >= n_regions * (20 + 64) + 5
    line = bench.result_line(args, wl, 1, res["seconds"], wl["B"] * args.steps, res, "test")
    assert line["verified"] is True and json.dumps(line)
    sh.ctx.close()


def test_bench_verify_has_teeth():
    """The same run checked against the WRONG arithmetic flavour must fail (the three modes are three different codecs)."""
    bench, sh, args, wl, res = _run(["--config", "2", "--steps", "6", "--warmup", "2", "--latency-steps", "0", "--ramp-steps", "4",
                                     "--no-kernel-table"])
    assert res["verify"]["verified"] is True
    sh.args.requant = "exact"
    assert sh.verify(64)["verified"] is False
    sh.ctx.close()


def test_bench_rccl_path_with_one_rank():
    """The N > 1 code path of bench.py has only ever met gloo on CPU (no multi-GPU box was available to this build):
    `--force-dist` brings the RCCL process group up with ONE rank, so init_process_group("nccl"), the weight broadcast,
    the barriers around the timed region, the max / sum all-reduce on device tensors, all_gather_object and the teardown
    run on a real GPU -- launched exactly as the driver launches N ranks (torch.distributed.run, 127.0.0.1)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29531", os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--bcast-weights",
           "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--latency-steps", "0", "--verify-streams", "16"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["ranks"] == 1 and "RCCL" in r["process_group"]
    assert r["verified"] is True and r["value"] > 1e6 and r["roofline"]["frac"] > 0.1


def test_bench_two_ranks_sharing_the_one_gpu():
    """The N-rank job itself on real hardware, as far as a 1-GPU box allows: two processes launched exactly as the driver
    launches them (torch.distributed.run, 127.0.0.1), each with its own context, CPU share, barriers around its timed
    region, the max / sum reduction, the rank gather and the weight broadcast -- both driving GPU 0
    (`--oversubscribe-device 0`, process group gloo: RCCL refuses two ranks on one device).  Every rank must verify its own
    run against the oracle; the line must say that it is no scaling measurement."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29532", os.path.join(root, "bench.py"), "--gpus", "2", "--oversubscribe-device", "0", "--bcast-weights",
           "--config", "2", "--steps", "12", "--warmup", "3", "--no-cpu-baseline", "--latency-steps", "0", "--verify-streams", "16"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["ranks"] == 2 and len(r["per_rank"]) == 2 and len(r["per_rank_ms_per_step"]) == 2
    assert r["verified"] is True and all(p["verified"] is True for p in r["per_rank"])
    assert r["scaling"] == "oversubscribed" and "not a scaling measurement" in r["oversubscribed"]
    assert r["config"]["total_streams"] == 2 * r["config"]["streams_per_gpu"]
    assert r["ms_per_step"] == max(r["per_rank_ms_per_step"])
