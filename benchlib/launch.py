"""benchlib.launch — how `bench.py --gpus N` survives its own native exchange: the per-rank GPU-free supervisor process and the self-launcher."""
import json
import os
import sys


def supervisor_verdict(records, rc):
    """What a rank's supervisor does with what its worker reported (raw pipe lines) and how the worker ended: (the line to print or None, the exit status, whether the
    worker got past the safe legs).  The last line record wins; a line that is not the worker's final one is marked "native_exchange": "crashed"; a worker that died
    after the torch.distributed legs were in does not fail the rank."""
    last, final, safe = None, False, False
    for raw in records:
        try:
            rec = json.loads(raw)
        except ValueError:
            continue                           # (a record cut short by the worker's death)
        if not isinstance(rec, dict):
            continue
        if "marker" in rec:
            safe = safe or rec["marker"] == "safe"
        elif rec.get("line") is not None:
            last, final = rec["line"], bool(rec.get("final"))
    if last is not None and not final:
        last["native_exchange"] = "crashed"
        last["fallback"] = (f"the rank's worker process ended (status {rc}) before its final line: this is the fastest verified leg among those that had finished — "
                            "printed by the rank's supervisor process")
    # a worker that died after the safe legs does not fail the rank (the first multi-GPU run is one shot and its line must come out of a launcher that kills every rank
    # when one exits non-zero); PQ_BENCH_STRICT_EXIT=1 (CI) turns that case into the distinct status 17 — the line is printed either way
    if rc != 0 and safe and os.environ.get("PQ_BENCH_STRICT_EXIT") == "1":
        return last, 17, safe
    return last, (0 if (rc == 0 or safe) else (rc if rc > 0 else 1)), safe


def supervise(args, script):
    """tp over more than one rank: THIS process (one per rank, started by torch.distributed.run) never touches the GPU.  It starts the real rank as a child — same
    command, same environment, plus a pipe — and relays what the child reports: rank 0's worker sends every line it could print so far (after the torch.distributed
    legs, after each native leg, the final one), every worker sends a marker once the torch.distributed legs are in.  However the child ends — normally, by its
    watchdog, or KILLED by a fault inside a native collective (a segfault or a GPU memory fault cannot be caught inside the process) — rank 0's supervisor prints the
    last line it holds ("native_exchange": "crashed" when the child died before its final line) and every supervisor whose child got as far as the safe legs exits 0:
    the first multi-GPU run is one shot, and a verified torch.distributed line must survive anything the native exchange does."""
    import subprocess
    rfd, wfd = os.pipe()
    env = dict(os.environ, PQ_BENCH_WORKER="1", PQ_BENCH_PIPE=str(wfd))
    child = subprocess.Popen([sys.executable, script, *sys.argv[1:]], env=env, pass_fds=(wfd,))
    os.close(wfd)
    with os.fdopen(rfd, "r") as pipe:
        records = list(pipe)                   # ends when the child (and everything that inherited the pipe) is gone
    rc = child.wait()
    rank = int(os.environ.get("RANK", "0"))
    line, code, safe = supervisor_verdict(records, rc)
    if rank == 0 and line is not None:
        sys.stdout.write(json.dumps(line) + "\n"); sys.stdout.flush()
    if rc != 0:
        print(f"[bench] rank {rank}: worker ended with status {rc}" + ("; the line measured before it is kept" if safe else ""), file=sys.stderr)
    sys.exit(code)


def relay_launch(returncode, stdout, ngpus):
    """What self_launch does with what the launcher left (pure: tested on the CPU): (the line to print or None, the exit status).  Exactly one JSON line is printed WHATEVER the
    launcher's status — a supervisor's survivor line comes out of a launcher that ended non-zero (torch.distributed.run reports a rank's failure that way), and losing it
    there would undo the supervisor; only an absent (or ambiguous) line is fatal.  The exit status is the launcher's."""
    lines = [l for l in stdout.splitlines() if l.strip().startswith("{")]
    if len(lines) != 1:
        print(f"[bench] the {ngpus}-rank launch failed (exit {returncode}, {len(lines)} JSON lines)", file=sys.stderr)
        return None, (returncode or 1)
    if returncode != 0:
        print(f"[bench] the {ngpus}-rank launcher ended with status {returncode}; rank 0's line came back and is printed", file=sys.stderr)
    return lines[0], returncode


def self_launch(args, script, launcher_cmd=None):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE / RANK in the environment): start the N ranks ourselves, exactly as the
    driver would (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same args>`),
    relay rank 0's ONE JSON line and exit with the launcher's status.  Runs BEFORE anything in this process has touched the GPU (`import torch`
    does not), and starts CHILD processes — never a re-exec of a process that initialised HIP."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = launcher_cmd or [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), script, *sys.argv[1:]]
    print(f"[bench] no launcher in the environment: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)          # stderr passes through
    for l in r.stdout.splitlines():
        if not l.strip().startswith("{"):
            print(l, file=sys.stderr)
    line, code = relay_launch(r.returncode, r.stdout, args.gpus)
    if line is not None:
        sys.stdout.write(line + "\n"); sys.stdout.flush()
    sys.exit(code)

