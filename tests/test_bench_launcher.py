"""CPU tests of bench.py's own launcher: `python bench.py --gpus N` must start N ranks by itself (one process per GPU through
torch.distributed.run, rendezvous on 127.0.0.1), must refuse a launcher that started a different number of ranks, and must
not silently run one rank when N GPUs were asked for.  --launch-only lets the ranks meet over gloo without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env.update(extra)
    return env


def _json_line(out: str):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_by_itself():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-only"], env=_env(), capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    line = _json_line(p.stdout)
    assert line["launch_only"] is True and line["n_gpus"] == 2
    assert [r["rank"] for r in line["ranks"]] == [0, 1]
    assert [r["local_rank"] for r in line["ranks"]] == [0, 1]
    assert len({r["pid"] for r in line["ranks"]}) == 2           # two processes, not two threads
    assert all(r["gpus_arg"] == 2 for r in line["ranks"])        # every rank got the same command line
    assert line["launcher"] == "torch.distributed.run"


def test_external_launcher_with_matching_world_is_used_as_is():
    # the driver's own command: it starts the ranks, bench.py must not start more
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29611", BENCH, "--gpus", "2", "--launch-only"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    line = _json_line(p.stdout)
    assert line["n_gpus"] == 2 and len(line["ranks"]) == 2


def test_world_size_mismatch_fails_loudly():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-only"],
                       env=_env(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29612"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert "WORLD_SIZE=4" in p.stderr and "--gpus 2" in p.stderr


def test_more_gpus_than_the_node_has_is_an_error_not_one_rank():
    import torch

    have = max(torch.cuda.device_count(), 1)
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(have + 1), "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 2, p.stdout + p.stderr
    assert f"--gpus {have + 1}" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]   # no benchmark line was printed
