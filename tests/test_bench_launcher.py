"""`python bench.py --gpus N` typed plainly starts its N ranks itself: one fresh process per GPU with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rank 0's line relayed on stdout, a failing rank fails the run.  --dry-launch
rehearses that without a GPU: every rank prints the environment it was started with."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=120)


def test_plain_command_starts_one_rank_per_gpu():
    p = run(["--gpus", "4", "--steps", "2", "--dry-launch"])
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "the contract is ONE line on stdout"
    zero = json.loads(lines[0])
    assert zero["dry_launch"] and zero["gpus"] == 4 and zero["env"]["RANK"] == "0" and zero["env"]["WORLD_SIZE"] == "4"
    assert zero["env"]["MASTER_ADDR"] == "127.0.0.1" and int(zero["env"]["MASTER_PORT"]) > 0
    others = [json.loads(l) for l in p.stderr.splitlines() if l.startswith("{")]
    assert sorted(o["env"]["RANK"] for o in others) == ["1", "2", "3"]
    assert all(o["env"]["LOCAL_RANK"] == o["env"]["RANK"] and o["env"]["MASTER_PORT"] == zero["env"]["MASTER_PORT"] for o in others)


def test_a_failing_rank_fails_the_run_and_nothing_is_printed():
    p = run(["--gpus", "3", "--dry-launch"], RCX_BENCH_DRY_FAIL_RANK="1")
    assert p.returncode == 3
    assert p.stdout.strip() == "", "no line when a rank failed"
    assert "rank 1 exited with 3" in p.stderr


def test_under_a_launcher_the_environment_counts():
    """Started by torch.distributed.run (WORLD_SIZE set): no second level of processes."""
    p = run(["--gpus", "2", "--dry-launch"], RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    assert p.returncode == 0 and p.stdout.strip() == ""
    mine = json.loads([l for l in p.stderr.splitlines() if l.startswith("{")][0])
    assert mine["env"]["RANK"] == "1" and mine["env"]["MASTER_PORT"] == "29999"
