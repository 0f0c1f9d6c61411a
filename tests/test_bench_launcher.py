"""`python bench.py --gpus N` started plainly (the form the driver records for N = 1) must start its own ranks: the parent
never imports torch or maps the HIP library, the ranks are fresh child processes with the env:// rendezvous variables, and
the parent's exit code is the ranks' largest.  No GPU here, so the ranks themselves stop at "no HIP device visible" -- which
is the behaviour under test on the CPU: the failure of a rank reaches the caller as a non-zero exit code, nothing hangs, and
no JSON line is printed.  The GPU half (two gloo ranks on one card through the plain form) is in
tests/test_a_bench_multirank.py."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env_without_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["HIP_VISIBLE_DEVICES"] = ""         # also on a GPU box this test is about the no-device exit
    env["CUDA_VISIBLE_DEVICES"] = ""
    return env


def test_plain_multi_gpu_invocation_starts_its_own_ranks_and_returns_their_exit_code():
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--batch", "256", "--steps", "1",
                          "--warmup", "0"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=_env_without_launcher())
    assert out.returncode == 1, (out.returncode, out.stderr[-2000:])
    assert out.stderr.count("no HIP device visible") == 2, out.stderr[-2000:]          # one per rank
    assert "launcher: rank 0 exited with code 1" in out.stderr and "launcher: rank 1 exited with code 1" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_the_launching_process_stays_clear_of_torch_and_hip():
    """What makes the plain form safe on the GPU box: the parent decides from the raw arguments and starts children before
    `import torch` / before libfrw.so is mapped.  Run the decision with a stub in place of the children."""
    code = r"""
import os, sys, runpy
os.environ.pop("WORLD_SIZE", None)
sys.argv = ["bench.py", "--gpus=4", "--steps", "1"]
import subprocess
started = []
class FakeProc:
    def __init__(self, cmd, env=None, preexec_fn=None):
        assert callable(preexec_fn)                    # the ranks ask for SIGTERM on the launcher's death (PR_SET_PDEATHSIG)
        started.append((cmd, env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_ADDR"], env["MASTER_PORT"]))
        self.pid = 1
    def poll(self):
        return 0
    def wait(self, timeout=None):
        return 0
subprocess.Popen = FakeProc
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit as ex:
    assert ex.code == 0, ex.code
assert "torch" not in sys.modules and "falcon_r1cs_amd" not in sys.modules and "numpy" not in sys.modules, sorted(sys.modules)
assert [s[1] for s in started] == ["0", "1", "2", "3"] and all(s[3] == "4" and s[4] == "127.0.0.1" for s in started), started
assert len({s[5] for s in started}) == 1 and all(s[0][0] == sys.executable and s[0][2:] == sys.argv[1:] for s in started)
print("ok")
"""
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120,
                         env=_env_without_launcher())
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr[-3000:]


def test_under_a_launcher_nothing_is_started():
    """WORLD_SIZE present (torch.distributed.run, or this file's own children): no second generation of ranks."""
    env = dict(_env_without_launcher(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1"], cwd=ROOT, capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode == 1 and "launcher" not in out.stderr and "no HIP device visible" in out.stderr, out.stderr[-2000:]


def test_ranks_do_not_outlive_an_interrupted_launcher(tmp_path):
    """a launcher that is terminated (a harness timeout, Ctrl-C) must take its ranks with it -- the signal is passed on, and
    whatever ends the wait terminates those still running.  The "ranks" here are bench.py itself started under a name that makes it sleep:
    a stand-in script, so no GPU is involved."""
    import signal
    import time
    stub = tmp_path / "bench_stub.py"
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("import numpy as np")]
    # the launcher part of bench.py verbatim; as a RANK (WORLD_SIZE set) the stub records its pid and sleeps
    stub.write_text(head.replace('if __name__ == "__main__" and "WORLD_SIZE" not in os.environ',
                                 'if "WORLD_SIZE" in os.environ:\n    open(%r + os.environ["RANK"], "w").write(str(os.getpid()))\n    time.sleep(120)\n    sys.exit(0)\n'
                                 'if __name__ == "__main__" and "WORLD_SIZE" not in os.environ' % str(tmp_path / "pid")))
    p = subprocess.Popen([sys.executable, str(stub), "--gpus", "2"], cwd=ROOT, env=_env_without_launcher())
    deadline = time.time() + 30
    while time.time() < deadline and not all(os.path.exists(str(tmp_path / ("pid%d" % r))) for r in range(2)):
        time.sleep(0.1)
    pids = [int(open(str(tmp_path / ("pid%d" % r))).read()) for r in range(2)]
    p.send_signal(signal.SIGTERM)
    assert p.wait(30) != 0
    time.sleep(0.5)
    for pid in pids:
        alive = True
        try:
            os.kill(pid, 0)
            # a zombie of a reparented child counts as gone once it has been reaped; give init a moment
            alive = open("/proc/%d/stat" % pid).read().split()[2] not in ("Z", "X")
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive, "rank %d outlived its launcher" % pid
