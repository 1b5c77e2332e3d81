"""bench.py's rank launcher (``--gpus N`` without a launcher): the ranks run in their own sessions, so a SIGTERM to the
parent (``timeout 900 python bench.py --gpus 8``) or a dying rank has to take every rank down - none may survive as an
orphan inside a collective, holding its GPU."""
import os
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PARENT = """
import sys
sys.path.insert(0, {root!r})
import bench
bench.__file__ = {child!r}
sys.argv = ["bench.py", {mode!r}, {piddir!r}]
sys.exit(bench.launch_ranks(3))
"""

CHILD = """
import os, sys, time
mode, piddir = sys.argv[1], sys.argv[2]
open(os.path.join(piddir, os.environ["RANK"]), "w").write(str(os.getpid()))
if mode == "rank1_dies" and os.environ["RANK"] == "1":
    time.sleep(0.5)
    sys.exit(7)
time.sleep(120)
"""


def _alive(pid):
    try:
        os.kill(pid, 0)
    except OSError:
        return False
    try:            # a zombie still answers kill(0)
        with open(f"/proc/{pid}/stat") as f:
            return f.read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def _start(tmp_path, mode):
    child = tmp_path / "rank.py"
    child.write_text(CHILD)
    piddir = tmp_path / "pids"
    piddir.mkdir()
    env = dict(os.environ, BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    parent = subprocess.Popen([sys.executable, "-c", PARENT.format(root=ROOT, child=str(child), mode=mode, piddir=str(piddir))],
                              env=env)
    end = time.time() + 60
    while time.time() < end and len(list(piddir.iterdir())) < 3:
        time.sleep(0.1)
    time.sleep(0.3)
    pids = [int(p.read_text()) for p in piddir.iterdir() if p.read_text()]
    assert len(pids) == 3
    return parent, pids


def test_sigterm_to_the_parent_stops_every_rank(tmp_path):
    parent, pids = _start(tmp_path, "sleep")
    assert all(_alive(p) for p in pids)
    parent.send_signal(signal.SIGTERM)
    assert parent.wait(timeout=30) == 128 + signal.SIGTERM
    end = time.time() + 10
    while time.time() < end and any(_alive(p) for p in pids):
        time.sleep(0.1)
    assert not any(_alive(p) for p in pids)


def test_a_dying_rank_takes_the_others_down(tmp_path):
    parent, pids = _start(tmp_path, "rank1_dies")
    assert parent.wait(timeout=30) == 7
    end = time.time() + 10
    while time.time() < end and any(_alive(p) for p in pids):
        time.sleep(0.1)
    assert not any(_alive(p) for p in pids)


def test_stdout_carries_nothing_but_the_line():
    """bench.hide_stdout / show_stdout: what libraries write to file descriptor 1 from C (RCCL's banner, gloo's connection
    lines) goes to stderr; only what is printed between show_stdout(True) and show_stdout(False) reaches stdout."""
    code = ("import os, sys; sys.path.insert(0, %r); import bench\n"
            "bench.hide_stdout(); os.write(1, b'library noise\\n')\n"
            "bench.show_stdout(True); print('{\"metric\": 1}', flush=True); bench.show_stdout(False)\n"
            "os.write(1, b'more noise\\n')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"metric": 1}\n'
    assert "library noise" in r.stderr and "more noise" in r.stderr
