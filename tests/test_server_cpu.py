"""The resident per-GPU server behind the compiled executables (pyp_amd/csrc/dropin_server.h, bin/ppm_server; SURVEY.md 8b "multiplex via a
daemon"): life cycle and protocol on a host without a GPU.  The compute paths are covered by tests/test_gpu_dropin.py on the GPU box."""
import os
import subprocess
import time

import numpy as np
import pytest

from pyp_amd.formats import cistem, mrc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SERVER = os.path.join(ROOT, "bin", "ppm_server")

SOCK = "pyp_amd_gpu0.u%d.sock" % os.getuid()          # the server's socket carries the user id (pyp_amd/csrc/dropin_server.h)


def _built(prog):
    exe = os.path.join(ROOT, "bin", prog)
    if not os.path.exists(exe) or open(exe, "rb").read(4) != b"\x7fELF":
        pytest.skip(f"bin/{prog} is built by __graft_entry__.build()")
    return exe


@pytest.fixture()
def lockdir(tmp_path, monkeypatch):
    monkeypatch.setenv("PPM_LOCK_DIR", str(tmp_path / "lock"))
    (tmp_path / "lock").mkdir()
    yield tmp_path / "lock"
    subprocess.run([SERVER, "--stop"], capture_output=True, timeout=30)          # whatever a test left running


def test_server_starts_reports_and_stops(lockdir):
    _built("ppm_server")
    assert subprocess.run([SERVER, "--stats"], capture_output=True, text=True).returncode != 0           # nothing runs yet
    assert subprocess.run([SERVER, "--daemon"], timeout=30).returncode == 0                                  # returns at once
    for _ in range(100):
        r = subprocess.run([SERVER, "--stats"], capture_output=True, text=True)
        if r.returncode == 0:
            break
        time.sleep(0.05)
    assert r.returncode == 0 and "served 0 calls" in r.stdout and (lockdir / SOCK).exists()
    assert oct((lockdir / SOCK).stat().st_mode & 0o777) == "0o700"                           # the owner's only
    # a second server for the same device steps back
    r2 = subprocess.run([SERVER], capture_output=True, text=True, timeout=30)
    assert r2.returncode == 0 and "already running" in r2.stdout
    r = subprocess.run([SERVER, "--stop"], capture_output=True, text=True, timeout=30)
    assert r.returncode == 0 and "stopping" in r.stdout
    for _ in range(100):
        if not (lockdir / SOCK).exists():
            break
        time.sleep(0.05)
    assert not (lockdir / SOCK).exists()


def test_servers_started_at_the_same_moment_leave_one(lockdir):
    """Two clients may both find no server and start one: the lifetime lock next to the socket lets exactly one serve; the others step
    back without touching its socket, and after --stop nobody holds the lock any more."""
    import fcntl
    _built("ppm_server")
    procs = [subprocess.Popen([SERVER, "--daemon"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for _ in range(4)]
    assert all(p.wait(timeout=30) == 0 for p in procs)
    for _ in range(100):
        r = subprocess.run([SERVER, "--stats"], capture_output=True, text=True)
        if r.returncode == 0:
            break
        time.sleep(0.05)
    assert r.returncode == 0 and "served 0 calls" in r.stdout
    time.sleep(0.3)                                       # the losers are gone by now; the socket must still answer
    assert subprocess.run([SERVER, "--stats"], capture_output=True, text=True).returncode == 0
    with open(lockdir / (SOCK + ".lock"), "r+") as f:
        with pytest.raises(OSError):
            fcntl.flock(f, fcntl.LOCK_EX | fcntl.LOCK_NB)            # held by the one server
    assert subprocess.run([SERVER, "--stop"], capture_output=True, text=True, timeout=30).returncode == 0
    with open(lockdir / (SOCK + ".lock"), "r+") as f:
        for _ in range(100):
            try:
                fcntl.flock(f, fcntl.LOCK_EX | fcntl.LOCK_NB)
                break
            except OSError:
                time.sleep(0.05)
        else:
            pytest.fail("a server survived --stop")


def _gpu_present():
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:          # noqa: BLE001
        return False


@pytest.mark.skipif(_gpu_present(), reason="the no-device message needs a host without a GPU")
def test_executable_through_the_server_fails_loudly_without_a_device(lockdir, tmp_path, monkeypatch):
    """PPM_STACK_CACHE=1: the executable starts the server on demand, sends it the call and passes on its answer - here the library's
    ERROR (no device), a non-zero exit and no output file; a call outside the fast path never reaches the server."""
    exe = _built("reconstruct3d")
    _built("ppm_server")
    monkeypatch.setenv("PPM_STACK_CACHE", "1")
    monkeypatch.setenv("PPM_STACK_CACHE_IDLE_S", "60")
    rows = cistem.default_rows(6, 2.0, 300.0, 2.7, 0.07)
    cistem.write_parameters(str(tmp_path / "p.cistem"), rows)
    mrc.write(np.zeros((6, 32, 32), np.float32), str(tmp_path / "s.mrc"), pixel_size=2.0)
    lines = ["s.mrc", "p.cistem", "null", "ref.mrc", "m1.mrc", "m2.mrc", "out.mrc", "r.res", "C1", 1, 6, 2.0, 300, 0, 30.0, 4.0, 0, 2.0, "no", 0, -1, "no", 0, 1, 1,
             "yes", "no", "no", "no", "no", "yes", "no", "no", "no", "no", "yes", "d1.mrc", "d2.mrc", 1]
    r = subprocess.run([exe], input="\n".join(str(x) for x in lines) + "\n", cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "ERROR" in r.stdout and "no HIP device" in r.stdout and not (tmp_path / "d1.mrc").exists()
    st = subprocess.run([SERVER, "--stats"], capture_output=True, text=True)
    assert st.returncode == 0 and "served 1 calls" in st.stdout
    bad = list(lines); bad[23] = 2.0            # smoothing: the Python implementation's refusal, the server is not asked
    r = subprocess.run([exe], input="\n".join(str(x) for x in bad) + "\n", cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "smoothing" in r.stdout and "resident server" not in r.stdout
    assert "served 1 calls" in subprocess.run([SERVER, "--stats"], capture_output=True, text=True).stdout
