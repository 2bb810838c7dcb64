"""`python bench.py --gpus N` (the driver's command, no launcher around it) must start N ranks itself or fail loudly -- never print a
1-GPU line labelled otherwise (VERDICT r3 missing #2).  No GPU needed: the launcher decides before it touches HIP."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CZ_BENCH_FORCE_MULTI", "CZ_BENCH_LAUNCH_DRY")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=120, env=e, cwd=ROOT)


def test_gpus_n_builds_the_torchrun_command():
    p = _run(["--gpus", "4", "--steps", "5", "--warmup", "2"], CZ_BENCH_LAUNCH_DRY="1")
    assert p.returncode == 0, p.stderr
    cmd = json.loads(p.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"] and cmd[-7].endswith("bench.py")


def test_more_gpus_than_visible_fails_loudly():
    p = _run(["--gpus", "2"])  # this container has no GPU at all
    assert p.returncode != 0
    assert "2 GPUs requested" in p.stderr and "visible" in p.stderr
    assert p.stdout.strip() == ""  # and above all: no JSON line


def test_world_size_must_equal_gpus():
    p = _run(["--gpus", "2"], WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr and p.stdout.strip() == ""
    p = _run(["--gpus", "8"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")  # a 1-rank launcher around --gpus 8: refused as well
    assert p.returncode != 0 and p.stdout.strip() == ""


def test_boxstate_reads_what_is_there_and_never_raises():
    sys.path.insert(0, ROOT)
    import boxstate
    s = boxstate.static_state()
    assert isinstance(s, dict) and "sysfs" in s
    with boxstate.Sampler(boxstate.device_sysfs(), 0.001) as smp:
        pass
    assert smp.summary()["samples"] >= 0
    assert boxstate._active_level("0: 500Mhz\n1: 2400Mhz *\n") == 2400
