"""CPU-only: the launcher and the watchdog of flashe_amd.dist (no GPU, no RCCL, no torch): a rank that dies takes the others with
it, a deadline or an abort raised by any rank ends every rank with the fallback hook run, and nothing is left behind."""
import os
import subprocess
import sys
import time

from conftest import ROOT

WORKER = r'''
import os, sys, time
sys.path.insert(0, %r)
mode = sys.argv[1]
rank = int(os.environ["RANK"])
if mode == "one_dies":
    if rank == 1:
        time.sleep(0.5); sys.exit(7)
    time.sleep(120)
elif mode == "all_ok":
    time.sleep(0.2)
elif mode == "sleep":
    time.sleep(120)
elif mode in ("deadline", "abort"):
    from flashe_amd.dist import Watchdog
    wd = Watchdog(rank, int(os.environ["WORLD_SIZE"]), on_fire=lambda why: print("FIRED", rank, why, flush=True))
    wd.exit_code = 0
    wd.arm(1.0 if mode == "deadline" else 60.0, "test phase")
    if mode == "abort" and rank == 2:
        time.sleep(0.5)
        wd.abort("rank 2 gives up")
    time.sleep(120)
elif mode == "finish":
    from flashe_amd.dist import Watchdog
    wd = Watchdog(rank, int(os.environ["WORLD_SIZE"]), on_fire=lambda why: print("FIRED", rank, why, flush=True))
    wd.arm(0.5, "x")
    assert wd.finish()
    time.sleep(1.2)
    print("DONE", rank, flush=True)
''' % ROOT

LAUNCH = r'''
import sys
sys.path.insert(0, %r)
from flashe_amd.dist import spawn
sys.exit(spawn(3, ["-c", %r, sys.argv[1]], master_port=29777))
'''


def run(mode, tmp_path, timeout=60):
    env = dict(os.environ, FLASHE_RDZV_DIR=str(tmp_path))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", LAUNCH % (ROOT, WORKER), mode], capture_output=True, text=True, timeout=timeout, env=env)
    return r, time.time() - t0


def test_spawn_stops_the_other_ranks_when_one_fails(tmp_path):
    r, took = run("one_dies", tmp_path)
    assert r.returncode == 7 and took < 20, (r.returncode, took, r.stderr[-500:])


def test_spawn_returns_zero_when_all_ranks_do(tmp_path):
    r, took = run("all_ok", tmp_path)
    assert r.returncode == 0, r.stderr[-500:]


def test_watchdog_deadline_fires_on_every_rank(tmp_path):
    r, took = run("deadline", tmp_path)
    assert r.returncode == 0 and took < 20, (r.returncode, took, r.stderr[-500:])
    assert sorted(l.split()[1] for l in r.stdout.splitlines() if l.startswith("FIRED")) == ["0", "1", "2"] and "deadline of phase 'test phase'" in r.stdout


def test_watchdog_abort_by_one_rank_reaches_all(tmp_path):
    r, took = run("abort", tmp_path)
    assert r.returncode == 0 and took < 20, (r.returncode, took, r.stderr[-500:])
    fired = [l for l in r.stdout.splitlines() if l.startswith("FIRED")]
    assert len(fired) == 3 and all("rank 2 gives up" in l for l in fired), r.stdout


def test_watchdog_finish_disarms(tmp_path):
    r, took = run("finish", tmp_path)
    assert r.returncode == 0 and "FIRED" not in r.stdout and r.stdout.count("DONE") == 3, r.stdout + r.stderr[-500:]


def test_launcher_sigterm_reaches_the_ranks(tmp_path):
    """`timeout` (or a driver) that kills the launcher must not leave rank processes behind holding GPUs."""
    import signal
    env = dict(os.environ, FLASHE_RDZV_DIR=str(tmp_path))
    p = subprocess.Popen([sys.executable, "-c", LAUNCH % (ROOT, WORKER), "sleep"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    time.sleep(1.5)
    kids = subprocess.run(["pgrep", "-P", str(p.pid)], capture_output=True, text=True).stdout.split()
    assert len(kids) == 3, kids
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=20) == 128 + signal.SIGTERM
    time.sleep(0.3)
    for k in kids:
        assert not os.path.exists(f"/proc/{k}") or open(f"/proc/{k}/stat").read().split()[2] == "Z", k
