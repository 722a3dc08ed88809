"""`bench.py --gpus N` without an external launcher: rayca_amd/launcher.py starts the ranks as child processes with the
environment torch.distributed.run would give them.  Covered here with a tiny gloo program (no GPU in this container)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_PROGRAM = textwrap.dedent("""
    import json, os, sys
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if len(sys.argv) > 1 and sys.argv[1] == "fail" and rank == 1:
        sys.exit(7)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"n_gpus": world, "sum": float(t[0])}), flush=True)
    dist.destroy_process_group()
""")

DRIVER = textwrap.dedent("""
    import sys
    sys.path.insert(0, {root!r})
    from rayca_amd.launcher import spawn_ranks
    assert "torch" not in sys.modules          # the parent never imports torch, let alone touches a device
    sys.exit(spawn_ranks(int(sys.argv[1]), [sys.executable, sys.argv[2]] + sys.argv[3:], timeout_s=120))
""")


def _run(tmp_path, world, *extra):
    prog = tmp_path / "rank_program.py"
    prog.write_text(RANK_PROGRAM)
    drv = tmp_path / "driver.py"
    drv.write_text(DRIVER.format(root=ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, str(drv), str(world), str(prog), *extra], capture_output=True, text=True, env=env, timeout=300)


def test_spawned_ranks_rendezvous_and_rank0_prints_one_line(tmp_path):
    r = _run(tmp_path, 2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"n_gpus": 2, "sum": 3.0}


def test_a_failing_rank_fails_the_launch(tmp_path):
    r = _run(tmp_path, 2, "fail")
    assert r.returncode == 7
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_relaunches_itself_before_importing_torch():
    """bench.py's own entry: with --gpus 2 and no WORLD_SIZE it must go to the launcher first (checked statically: the
    launcher call precedes the first torch import in main())."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def main():"):]
    assert body.index("relaunch_self(args.gpus)") < body.index("import torch")
    assert '"WORLD_SIZE" not in os.environ' in body


SLEEPER = textwrap.dedent("""
    import os, sys, time
    open(sys.argv[1] + "." + os.environ["RANK"], "w").write(str(os.getpid()))
    time.sleep(600)
""")


def _pids_gone(pids, within_s=15.0):
    import time
    t_end = time.monotonic() + within_s
    while time.monotonic() < t_end:
        live = []
        for pid in pids:
            try:
                os.kill(pid, 0)
                # (a zombie of our own driver's child has been reaped by the driver; anything still signalable is alive)
                live.append(pid)
            except ProcessLookupError:
                pass
        if not live:
            return True
        time.sleep(0.1)
    return False


def _start_sleepers(tmp_path, world=2):
    import time
    prog = tmp_path / "sleeper.py"
    prog.write_text(SLEEPER)
    drv = tmp_path / "driver.py"
    drv.write_text(DRIVER.format(root=ROOT))
    stamp = str(tmp_path / "pid")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    parent = subprocess.Popen([sys.executable, str(drv), str(world), str(prog), stamp], env=env)
    t_end = time.monotonic() + 60
    while time.monotonic() < t_end and not all(os.path.exists(f"{stamp}.{r}") and open(f"{stamp}.{r}").read() for r in range(world)):
        time.sleep(0.05)
    pids = [int(open(f"{stamp}.{r}").read()) for r in range(world)]
    return parent, pids


def test_ranks_do_not_outlive_a_terminated_launcher(tmp_path):
    """SIGTERM to the launching process (a harness `timeout`): its handler raises into the try/finally that stops the
    exact child PIDs; exit code 128 + SIGTERM."""
    import signal
    parent, pids = _start_sleepers(tmp_path)
    parent.send_signal(signal.SIGTERM)
    assert parent.wait(timeout=30) == 128 + signal.SIGTERM
    assert _pids_gone(pids)


def test_ranks_do_not_outlive_a_killed_launcher(tmp_path):
    """SIGKILL leaves the launcher no chance to clean up: the ranks go through the parent-death signal."""
    parent, pids = _start_sleepers(tmp_path)
    parent.kill()
    parent.wait(timeout=30)
    assert _pids_gone(pids)
