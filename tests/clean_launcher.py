"""A child process that never touches the GPU and starts other programs on request.

On the GPU boxes a process that has initialised the GPU must not exec another program (and a forked child of such
a process carries the same state).  tests/conftest.py starts this helper in pytest_configure -- before any test module
touched the GPU -- and the `clean_launcher` fixture sends it command lines: one JSON object per line on stdin
({"argv": [...], "env": {...}, "cwd": "...", "timeout": seconds}), one JSON object per line back
({"rc": int, "stdout": "...", "stderr": "..."}).  Test infrastructure only.
"""
import json
import os
import subprocess
import sys


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        req = json.loads(line)
        env = dict(os.environ)
        env.update(req.get("env") or {})
        try:
            p = subprocess.run(req["argv"], env=env, cwd=req.get("cwd"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               text=True, timeout=req.get("timeout", 600))
            rep = {"rc": p.returncode, "stdout": p.stdout[-20000:], "stderr": p.stderr[-20000:]}
        except subprocess.TimeoutExpired as e:
            rep = {"rc": -9, "stdout": (e.stdout or "")[-20000:] if isinstance(e.stdout, str) else "",
                   "stderr": "timeout after %s s" % req.get("timeout", 600)}
        sys.stdout.write(json.dumps(rep) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
