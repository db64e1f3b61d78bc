"""Launch plumbing of bench.py: the ONE JSON line on file descriptor 1, self-launch of N ranks, the dry launch."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_JSON_FD = None


def _claim_stdout():
    """Keep file descriptor 1 for the ONE JSON line: everything else that writes to stdout (RCCL prints its library
    path there from C, after Python's own buffers are gone) is sent to stderr."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def _emit(obj):
    os.write(_JSON_FD if _JSON_FD is not None else 1, (json.dumps(obj) + "\n").encode())


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _self_launch(argv, n):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): this process -- which has made NO GPU call and
    makes none -- starts N fresh children, one rank per GPU, with the same environment contract torch.distributed.run
    would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT), forwards rank 0's stdout
    (the ONE JSON line) to its own stdout, sends every other rank's stdout to stderr and exits with the worst child
    return code.  Children are new processes (subprocess, not exec): nothing that has initialised the GPU is replaced."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, MMTG_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else 2))
    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)   # rank 0's stdout: the JSON line
    reader.start()
    worst, deadline = 0, None
    pending = list(procs)
    while pending:
        for p in list(pending):
            rc = p.poll()
            if rc is None:
                continue
            pending.remove(p)
            if rc != 0:
                worst = worst or rc
                if deadline is None:                # a rank died: the others would wait in a collective for ever
                    deadline = time.time() + float(os.environ.get("MMTG_BENCH_KILL_GRACE", "30"))
        if deadline is not None and time.time() > deadline:
            for p in pending:
                p.kill()                            # exactly the PIDs this process started
        time.sleep(0.05)
    reader.join(timeout=5.0)
    if worst == 0:
        for raw in lines:
            os.write(_JSON_FD if _JSON_FD is not None else 1, raw)
    return worst


def _dry_launch(args):
    """--dry-launch: prove the launch contract without a GPU -- every rank joins a gloo group over the rendezvous the
    launcher handed it, ranks are all-gathered, rank 0 prints the ONE JSON line."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    seen = [rank]
    if os.environ.get("MMTG_DRY_FAIL_RANK") == str(rank):      # test hook: a rank that dies before the rendezvous
        raise SystemExit(7)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        got = [None] * world
        dist.all_gather_object(got, (rank, int(os.environ.get("LOCAL_RANK", "0")), os.getpid()))
        seen = got
        dist.barrier()
        dist.destroy_process_group()
    print("[bench dry-launch] rank %d of %d pid %d" % (rank, world, os.getpid()), file=sys.stderr)
    if rank == 0:
        _emit({"dry_launch": True, "n_gpus": args.gpus, "world": world, "ranks": seen,
               "self_launched": bool(os.environ.get("MMTG_BENCH_CHILD"))})
