import os
import sys
import time

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


BREADCRUMB = os.environ.get("BNERV_TEST_TRAIL", os.path.join(ROOT, "gpurun_out", "test_trail.txt"))
_trail = None


def _crumb(text, to_stderr=True):
    """One line per test phase, written and flushed BEFORE the phase runs: to stderr (pytest's capture is bypassed through the
    saved real descriptor) and to a file, so that a native abort (SIGABRT / SIGSEGV inside a kernel launch, RCCL or a graph
    capture) still leaves the name of the test it died in -- the round-4 driver run died with nothing but dots on record."""
    global _trail
    line = f"{text}  t={time.monotonic() - _T0:.1f}s\n"
    if _crash is not None:
        _crash.bnerv_crashtrace_note(text.encode()[:250])
    if to_stderr:
        try:
            os.write(_REAL_STDERR, line.encode())
        except OSError:
            pass
    try:
        if _trail is None:
            os.makedirs(os.path.dirname(BREADCRUMB), exist_ok=True)
            _trail = open(BREADCRUMB, "a", buffering=1)
        _trail.write(line)
        _trail.flush()
        os.fsync(_trail.fileno())
    except OSError:
        pass


_REAL_STDERR = os.dup(2)
_T0 = time.monotonic()
_crash = None
_fault_file = None


def _install_crashtrace():
    """tests/native/crashtrace.c: native backtrace of the faulting thread on SIGABRT / SIGSEGV / SIGBUS, written to the real stderr
    and chained in front of faulthandler.  Built here with gcc (test infrastructure; absent compiler => breadcrumbs only)."""
    global _crash
    import ctypes
    import subprocess
    src = os.path.join(ROOT, "tests", "native", "crashtrace.c")
    so = os.path.join(ROOT, "tests", "native", "_crashtrace.so")
    try:
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", so + f".{os.getpid()}", src, "-ldl"])
            os.replace(so + f".{os.getpid()}", so)
        lib = ctypes.CDLL(so)
        lib.bnerv_crashtrace_note.argtypes = [ctypes.c_char_p]
        if lib.bnerv_crashtrace_install(os.dup(_REAL_STDERR)) == 0:
            _crash = lib
    except (OSError, subprocess.CalledProcessError):
        _crash = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "experimental: exercises a kernel form that is not in the default library")
    config.addinivalue_line("markers", "isolated: run the test body in a child pytest process (a native crash becomes an ordinary failure)")
    # conftest is imported AFTER pytest's fd-level capture replaced descriptor 2: the terminal's stderr is the descriptor the capture
    # manager saved (private attribute, hence the guarded lookup; without it the crumbs still reach the trail file)
    global _REAL_STDERR
    try:
        capman = config.pluginmanager.getplugin("capturemanager")
        _REAL_STDERR = os.dup(capman._global_capturing.err.targetfd_save)
    except Exception:       # noqa: BLE001
        pass
    # order matters: the native tracer first, faulthandler on top of it -- on a fatal signal faulthandler dumps the Python stacks and
    # re-raises into the tracer, which prints the native frames and, as the LAST line of the log, the test that was running
    import faulthandler
    global _fault_file
    faulthandler.disable()
    if os.environ.get("BNERV_CRASHTRACE", "1") != "0":
        _install_crashtrace()
    _fault_file = os.fdopen(os.dup(_REAL_STDERR), "w")
    faulthandler.enable(file=_fault_file, all_threads=True)


def pytest_runtest_logstart(nodeid, location):
    _crumb(f"[bnerv-trail] START {nodeid}")


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    """A GPU test ends with a device synchronisation inside ITS OWN call phase, so an asynchronous fault (an out-of-bounds access
    of a kernel the test enqueued) is reported against that test and not against whichever test synchronises next."""
    outcome = yield
    if "gpu" in item.keywords and torch.cuda.is_available():
        try:
            torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001 - a HIP error here must become THIS test's failure
            if outcome.excinfo is None:
                raise AssertionError(f"device fault surfaced at the end of {item.nodeid}: {e}") from e


def pytest_runtest_logfinish(nodeid, location):
    _crumb(f"[bnerv-trail] END   {nodeid}", to_stderr=False)


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


_session_status = [None]


def pytest_sessionfinish(session, exitstatus):
    _session_status[0] = int(exitstatus)


@pytest.hookimpl(trylast=True)
def pytest_unconfigure(config):
    """A GPU session quiesces torch's autograd worker before the interpreter goes down (boosting_nerv_amd.runtime.quiesce_autograd: the
    root cause of round 5's exit-time abort, profiles/r06_exit_abort.md) and then leaves the REGULAR way.  BNERV_HARD_EXIT=1 restores
    round 5's behaviour -- no interpreter teardown at all (atexit callbacks, flush, os._exit with pytest's own status); =force applies that
    without a GPU (the CPU test of the hook)."""
    mode = os.environ.get("BNERV_HARD_EXIT", "0")
    if _session_status[0] is None or not (torch.cuda.is_available() or mode == "force"):
        return
    from boosting_nerv_amd.runtime import hard_exit, quiesce_autograd, tool_attached
    quiesce_autograd()
    if mode not in ("1", "force"):
        return
    try:
        tr = config.pluginmanager.getplugin("terminalreporter")
        if tr is not None:
            tr._tw.flush()
    except Exception:       # noqa: BLE001
        pass
    if tool_attached():      # (a profiler writes its output from a C-level exit handler: leave the regular way)
        return
    os.environ["BNERV_HARD_EXIT"] = "1"
    hard_exit(_session_status[0])


ISOLATED_CHILD = "BNERV_ISOLATED_CHILD"


def run_isolated(nodeid, timeout=900, marker="gpu"):
    """Run ONE test in a child pytest process and turn whatever happens to the child into an ordinary outcome of the calling test:
    exit 0 -> passed (or skipped, if the child skipped), anything else -- an assertion, a timeout, a native crash (SIGABRT / SIGSEGV
    inside a kernel launch, RCCL, a graph capture) -- -> pytest.fail with the tail of the child's output, which ends with the crash
    tracer's native frames and its "died in" line.  The suite's main process survives and every other test keeps its evidence
    (round 4 lost 183 results to one abort).  Used through @pytest.mark.isolated, see pytest_pyfunc_call below."""
    import subprocess
    env = dict(os.environ)
    env[ISOLATED_CHILD] = "1"
    env["BNERV_TEST_TRAIL"] = BREADCRUMB + ".child"
    cmd = [sys.executable, "-m", "pytest", nodeid, "-x", "-q", "-p", "no:cacheprovider", "-m", marker]
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
        out, rc = r.stdout.decode("utf-8", "replace"), r.returncode
    except subprocess.TimeoutExpired as e:
        out, rc = (e.stdout or b"").decode("utf-8", "replace"), "timeout"
    if rc == 0:
        tail = out[-600:]
        if " skipped" in tail and " passed" not in tail:
            pytest.skip(f"isolated child skipped {nodeid}")
        return out
    how = f"killed by signal {-rc}" if isinstance(rc, int) and rc < 0 else f"exit status {rc}"
    keep = [ln for ln in out.splitlines() if not ln.startswith("  File ") and not ln.startswith("Extension modules:")]
    pytest.fail(f"isolated child for {nodeid}: {how}\n" + "\n".join(keep[-70:])[-6000:], pytrace=False)


@pytest.hookimpl(tryfirst=True)
def pytest_pyfunc_call(pyfuncitem):
    """@pytest.mark.isolated: the test body runs in a child process (run_isolated); inside that child it runs normally."""
    if pyfuncitem.get_closest_marker("isolated") is None or os.environ.get(ISOLATED_CHILD) == "1":
        return None
    if "gpu" in pyfuncitem.keywords and not torch.cuda.is_available():
        return None
    run_isolated(pyfuncitem.nodeid, marker="gpu" if "gpu" in pyfuncitem.keywords else "not gpu")
    return True


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def group(npz, prefix):
    """Sub-dict of an npz whose keys start with prefix (prefix stripped), as torch tensors."""
    out = {}
    for k in npz.files:
        if k.startswith(prefix):
            v = npz[k]
            out[k[len(prefix):]] = torch.from_numpy(v) if v.dtype.kind in "fiu" and v.ndim > 0 else v
    return out


def check_summary(t, npz, name, rtol=1e-3, atol=1e-5):
    f = t.detach().flatten().cpu()
    idx = torch.from_numpy(npz[f"{name}.idx"])
    ref = torch.from_numpy(npz[f"{name}.val"])
    assert tuple(t.shape) == tuple(npz[f"{name}.shape"]), (t.shape, npz[f"{name}.shape"])
    torch.testing.assert_close(f[idx].float(), ref, rtol=rtol, atol=atol)
    assert abs(f.double().mean().item() - float(npz[f"{name}.mean"])) <= atol + rtol * abs(float(npz[f"{name}.mean"])) + 1e-6


@pytest.fixture(scope="session")
def golden():
    return load_golden
