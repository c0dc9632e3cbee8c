"""`python bench.py --gpus N` must itself start N rank processes (the driver calls it that way for N = 1 and may for N > 1):
a plain child process per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, the parent never importing torch or
touching HIP (a process that has initialised the GPU must not be replaced), rank 0's line relayed, a failing rank = non-zero
exit.  `--launch-dry-run` lets the children report their environment and leave before anything heavy is imported."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_2_spawns_two_ranks_with_the_right_environment():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-dry-run"], capture_output=True, text=True, timeout=120, env=_env())
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1  # ONE JSON line on stdout
    out = json.loads(lines[0])
    assert out["gpus"] == 2 and out["parent_imported_torch"] is False
    kids = out["children"]
    assert [k["rank"] for k in kids] == [0, 1]
    for r_, k in enumerate(kids):
        assert k["local_rank"] == str(r_) and k["world_size"] == "2" and k["gpus_flag"] == 2
        assert k["master_addr"] == "127.0.0.1" and k["imported_torch"] is False
    assert kids[0]["master_port"] == kids[1]["master_port"] and 1024 < int(kids[0]["master_port"]) < 65536


def test_world_size_in_the_environment_means_this_process_is_a_rank():
    # the torch.distributed.run form: the launcher must not start ranks of its own
    env = dict(_env(), RANK="1", LOCAL_RANK="1", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--launch-dry-run"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr
    k = json.loads(r.stdout)
    assert k["rank"] == 1 and k["world_size"] == "4" and "children" not in k


def test_a_failing_rank_fails_the_launch():
    # without a GPU every rank fails at context creation (there is no CPU fallback): the launcher must pass that on
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--degree-bits", "6"],
                       capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode != 0
    assert "launcher: rank" in r.stderr
    assert not r.stdout.strip()


def test_watchdog_reports_once_and_exits_when_the_work_hangs():
    """bench.guarded: the sharded side measurement under its watchdog (LCP2_SHARDED_TIMEOUT_S).  A work function that outlives the
    limit: the timeout line goes out once and the process exits with code 3, without waiting for the work."""
    code = ("import os, sys, time; sys.path.insert(0, %r); import bench\n"
            "bench.guarded(lambda: time.sleep(30), 0.05, lambda why: print('LINE ' + why, flush=True))\n"
            "print('not reached')\n") % ROOT
    t0 = __import__("time").time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, env=_env())
    assert r.returncode == 3 and __import__("time").time() - t0 < 20
    assert r.stdout.count("LINE ") == 1 and "not reached" not in r.stdout and "did not finish within 0.05 s" in r.stdout


def test_watchdog_does_nothing_when_the_work_finishes_first():
    sys.path.insert(0, ROOT)
    import time
    import bench
    events = []
    assert bench.guarded(lambda: "done", 0.2, lambda why: events.append(("line", why)), exit_fn=lambda c: events.append(("exit", c))) == "done"
    time.sleep(0.4)  # a timer that was not cancelled would fire here
    assert events == []
    # the timeout wins the race: its thread reports exactly once, and the caller waits for it instead of running on
    # (with the real exit_fn the process is gone at that point)
    out = bench.guarded(lambda: time.sleep(0.3) or "late", 0.05, lambda why: events.append(("line", why)), exit_fn=lambda c: events.append(("exit", c)))
    assert out == "late" and [e[0] for e in events] == ["line", "exit"] and events[1][1] == 3
    # bench.py's default for its side measurement: the line carries the error, the replica result stands, the exit code is 0
    events.clear()
    bench.guarded(lambda: time.sleep(0.3), 0.05, lambda why: events.append(("line", why)), exit_fn=lambda c: events.append(("exit", c)), exit_code=0)
    assert events[1] == ("exit", 0)
    # an exception of the work is the caller's to handle; the watchdog stays quiet
    events.clear()
    try:
        bench.guarded(lambda: 1 / 0, 0.2, lambda why: events.append(why), exit_fn=lambda c: events.append(c))
    except ZeroDivisionError:
        pass
    time.sleep(0.4)
    assert events == []
