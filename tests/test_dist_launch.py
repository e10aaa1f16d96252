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


def test_watchdog_flag_in_a_private_directory_and_a_squatted_name(tmp_path):
    """the abort flag had a predictable name in a shared directory.  It now lives in a 0700 directory of this user; when
    that NAME is taken by something else (here: a plain file squatting on it) the watchdog falls back to the shared directory, and an
    abort() that finds a flag it cannot trust fires LOCALLY instead of silently relying on peers that will never see it."""
    import stat
    code = r'''
import os, sys, time
sys.path.insert(0, %r)
from flashe_amd.dist import Watchdog
wd = Watchdog(0, 1, on_fire=lambda why: print("FIRED", why, flush=True))
wd.exit_code = 0
print("DIR", wd.dir, flush=True)
if sys.argv[1] == "squat":
    assert wd.dir is None
    # a flag this launch did not write, which a watcher would not trust: a FIFO nobody writes to cannot even be opened for reading
    # without blocking -- use a directory instead: open() fails, so the flag is "not ours"
    os.mkdir(wd.path)
    wd.abort("give up")
    time.sleep(30)
else:
    st = os.lstat(wd.dir)
    assert (st.st_mode & 0o777) == 0o700 and st.st_uid == os.getuid()
    wd.abort("give up")
    time.sleep(30)
''' % ROOT
    env = dict(os.environ, FLASHE_RDZV_DIR=str(tmp_path), MASTER_PORT="29888", FLASHE_RUN_ID="t1")
    r = subprocess.run([sys.executable, "-c", code, "normal"], capture_output=True, text=True, timeout=60, env=env)
    assert r.returncode == 0 and "FIRED rank 0: give up" in r.stdout, r.stdout + r.stderr[-800:]
    private = [l.split(" ", 1)[1] for l in r.stdout.splitlines() if l.startswith("DIR ")][0]
    assert private.startswith(str(tmp_path)) and stat.S_ISDIR(os.lstat(private).st_mode)
    # the same launch name again, the directory's name squatted by a regular file
    env2 = dict(env, FLASHE_RUN_ID="t2")
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import os; from flashe_amd.dist import run_tag, rdzv_dir; "
                            "print(os.path.join(rdzv_dir(), f'flashe_run_{os.getuid()}_{run_tag(1)}'))" % ROOT],
                           capture_output=True, text=True, env=env2)
    # (run_tag contains the parent pid: compute the name the child will use by making this test process the parent of both)
    name = probe.stdout.strip()
    open(name, "w").close()
    r = subprocess.run([sys.executable, "-c", code, "squat"], capture_output=True, text=True, timeout=60, env=env2)
    assert r.returncode == 0 and "DIR None" in r.stdout and "FIRED" in r.stdout and "foreign file" in r.stdout, r.stdout + r.stderr[-800:]


def _bench_lines(cmd, tmp_path, env_extra=None, timeout=120):
    import json
    env = dict(os.environ, FLASHE_RDZV_DIR=str(tmp_path), BENCH_SHM_DIR=str(tmp_path))
    env.update(env_extra or {})
    r = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, timeout=timeout, env=env)
    return r, [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{"metric"')]


def test_bench_preflight_refuses_a_machine_with_too_few_devices(tmp_path):
    """`bench.py --gpus N` on a machine that cannot carry N ranks must fail LEGIBLY -- every rank runs the preflight
    before it creates an engine (device count >= WORLD_SIZE, LOCAL_RANK in range, peer access, RCCL), the first failure raises the
    abort flag, rank 0 prints ONE JSON line {"value": null, "error": "preflight: ..."} with what it established in `config`, and every
    rank leaves non-zero.  Through the comm double with an injected count ("2 devices, world 3"), and bare on this box (no device)."""
    r, lines = _bench_lines([os.path.join(ROOT, "tests", "bench_shm.py"), "--gpus", "3", "--n", "1000", "--steps", "1", "--warmup", "0"],
                            tmp_path, {"BENCH_SHM_DEVICES": "2"})
    assert r.returncode != 0, r.stdout[-1000:] + r.stderr[-2000:]
    assert len(lines) == 1, r.stdout[-1000:] + r.stderr[-2000:]
    d = lines[0]
    assert d["value"] is None and d["error"] == "preflight: 2 devices visible, 3 ranks requested" and d["n_gpus"] == 3
    cfg = d["config"]
    assert cfg["devices_visible"] == 2 and cfg["library"] == "libflashe_hip.so" and len(cfg["library_sha256_16"]) == 16
    assert "Traceback" not in r.stdout


def test_bench_preflight_line_on_a_box_without_devices(tmp_path):
    """The real launch path (bench.py --gpus 2 spawning its ranks, no double): wherever fewer devices are visible than ranks were
    asked for, the outcome is the preflight line, not a traceback from rank k.  Skipped where two devices exist."""
    import ctypes
    from flashe_amd import _lib
    n = ctypes.c_int(0)
    _lib.load().flashe_device_count(ctypes.byref(n))
    if n.value >= 2:
        import pytest
        pytest.skip("two devices visible: the launch would run")
    r, lines = _bench_lines([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n", "1000", "--steps", "1", "--warmup", "0"], tmp_path)
    assert r.returncode != 0 and len(lines) == 1, r.stdout[-1000:] + r.stderr[-2000:]
    d = lines[0]
    assert d["value"] is None and d["error"] == f"preflight: {n.value} devices visible, 2 ranks requested", d
    assert d["config"]["devices_visible"] == n.value and d["config"]["abi_version"] == 4
    assert "Traceback" not in r.stderr, r.stderr[-1500:]                  # a refusal, not a crash
