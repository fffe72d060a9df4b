"""Hunt for the exit-time abort of finished torch-ROCm processes: run a payload script N times with the REGULAR interpreter
teardown (BNERV_HARD_EXIT=0) under the native crash tracer (tests/native/crashtrace.c) and keep the log of every child that does not
exit with status 0.  usage: python tools/exit_hunt.py N [jobs] [payload.py args...]      (default payload: tools/split_contract.py)
A child is:  python tools/exit_hunt.py --child <payload> ...   -- installs the tracer, then runs the payload as __main__."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(argv):
    import ctypes
    import runpy
    so = os.path.join(ROOT, "tests", "native", "_crashtrace.so")
    src = os.path.join(ROOT, "tests", "native", "crashtrace.c")
    if not os.path.exists(so):
        subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", so, src, "-ldl"])
    lib = ctypes.CDLL(so)
    lib.bnerv_crashtrace_note.argtypes = [ctypes.c_char_p]
    lib.bnerv_crashtrace_install(os.dup(2))
    lib.bnerv_crashtrace_note(("exit-hunt payload " + " ".join(argv)).encode()[:250])
    import atexit
    atexit.register(lambda: lib.bnerv_crashtrace_note(b"exit-hunt: INTERPRETER TEARDOWN (payload finished, atexit reached)"))
    sys.argv = argv
    sys.path.insert(0, ROOT)
    runpy.run_path(argv[0], run_name="__main__")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2:])
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    payload = sys.argv[3:] or [os.path.join(ROOT, "tools", "split_contract.py")]
    out = os.path.join(ROOT, "gpurun_out", "exit_hunt")
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, BNERV_HARD_EXIT="0", PYTHONFAULTHANDLER="0")
    running, done, bad, i = [], 0, [], 0
    t0 = time.time()
    while done < n:
        while len(running) < jobs and i < n:
            log = open(os.path.join(out, f"child_{i}.log"), "w")
            p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child"] + payload, env=env, stdout=log, stderr=subprocess.STDOUT, cwd=ROOT)
            running.append((i, p, log))
            i += 1
        time.sleep(0.2)
        for item in list(running):
            k, p, log = item
            rc = p.poll()
            if rc is None:
                continue
            log.close()
            running.remove(item)
            done += 1
            path = os.path.join(out, f"child_{k}.log")
            if rc != 0:
                bad.append((k, rc))
                os.replace(path, os.path.join(out, f"BAD_{k}_rc{rc}.log"))
            else:
                os.remove(path)
    print(f"exit hunt: {n} children of `{' '.join(os.path.basename(x) for x in payload)}` with the regular teardown, {jobs} at a time, {time.time() - t0:.0f} s: "
          f"{len(bad)} abnormal exits {bad}")


if __name__ == "__main__":
    main()
