"""Single-node multi-rank launcher: one child process per GPU, torchrun's environment contract.

``bench.py --gpus N`` (and any other entry point that wants it) calls :func:`spawn_ranks` when it
is started WITHOUT a launcher (``WORLD_SIZE`` unset) and N > 1: the parent -- before anything has
touched HIP, so no GPU state is inherited or exec'd over -- starts N copies of the same command
with ``RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT`` set, relays
their output (rank 0's stdout verbatim, everything else to stderr with a rank prefix) and returns
the worst exit code.  The reference is single-process, one GPU (practical_slurm_launcher.sh:8-11,
test_mod_siren.py:90-93); this is the scale-out of SURVEY.md §8(e).

Nothing here imports torch or the HIP library.
"""

from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time

ENV_KEYS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def free_port(addr: str = "127.0.0.1") -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind((addr, 0))
        return int(s.getsockname()[1])


def under_launcher(env=None) -> bool:
    """True when a launcher (torchrun or :func:`spawn_ranks`) has already set up this process."""
    env = os.environ if env is None else env
    return "WORLD_SIZE" in env and "RANK" in env


def rank_env(rank: int, world: int, port: int, addr: str = "127.0.0.1", base=None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update({
        "RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
        "MASTER_ADDR": addr, "MASTER_PORT": str(port),
        # the host driver only supports dmabuf IPC (RCCL / device-memory sharing across processes)
        "HSA_ENABLE_IPC_MODE_LEGACY": env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
    })
    return env


def _pump(stream, sink, prefix: str, keep: list | None):
    for line in iter(stream.readline, ""):
        if keep is not None:
            keep.append(line)
        sink.write(prefix + line)
        sink.flush()
    stream.close()


def spawn_ranks(argv: list[str], world: int, *, timeout: float | None = None, env=None, port: int | None = None,
                stdout=None, stderr=None) -> tuple[int, str]:
    """Run ``argv`` as ``world`` rank processes; returns (worst exit code, rank 0's stdout).

    A rank that fails takes the others down (they would otherwise wait in a collective for ever);
    ``timeout`` bounds the whole job.  The children are direct subprocesses: no exec over a process
    that has initialised the GPU, no shell.
    """
    if world < 1:
        raise ValueError(f"world size must be positive, got {world}")
    stdout = sys.stdout if stdout is None else stdout
    stderr = sys.stderr if stderr is None else stderr
    port = free_port() if port is None else port
    procs, pumps, rank0_out = [], [], []
    for r in range(world):
        p = subprocess.Popen(argv, env=rank_env(r, world, port, base=env), stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, bufsize=1)
        procs.append(p)
        stderr.write(f"[launch] rank {r} of {world} started (pid {p.pid})\n")
        stderr.flush()
        pumps.append(threading.Thread(target=_pump, daemon=True,
                                      args=(p.stdout, stdout if r == 0 else stderr, "" if r == 0 else f"[rank {r}] ",
                                            rank0_out if r == 0 else None)))
        pumps.append(threading.Thread(target=_pump, args=(p.stderr, stderr, f"[rank {r}] ", None), daemon=True))
    for t in pumps:
        t.start()
    t0, worst, failed = time.monotonic(), 0, False
    live = set(range(world))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                worst = rc if worst == 0 or abs(rc) > abs(worst) else worst
                failed = True
        timed_out = timeout is not None and time.monotonic() - t0 > timeout
        if (failed or timed_out) and live:
            for r in live:
                procs[r].terminate()
            deadline = time.monotonic() + 10
            for r in list(live):
                try:
                    procs[r].wait(max(0.1, deadline - time.monotonic()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            live.clear()
            if timed_out and worst == 0:
                worst = 124
        if live:
            time.sleep(0.05)
    for t in pumps:
        t.join(5)
    return worst, "".join(rank0_out)


# ---- rendezvous of a small blob (the 128-byte RCCL unique id) ------------------------------------------------

def _recv_exact(conn, n: int) -> bytes:
    buf = b""
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise OSError("peer closed the connection")
        buf += chunk
    return buf


def _exchange_socket(payload: bytes | None, rank: int, world: int, addr: str, port: int, timeout: float) -> bytes:
    """Rank 0 serves ``payload`` on (addr, port) until every one of the other world-1 ranks has ACKNOWLEDGED it; they
    connect with retry.  Wire format: peer -> its rank (4 bytes); rank 0 -> length (4 bytes) + payload; peer -> one
    ack byte.  A peer whose read came up short simply connects again: rank 0 counts acknowledged ranks, not accepted
    connections, so a retry never uses up another rank's turn."""
    deadline = time.monotonic() + timeout
    if rank == 0:
        assert payload is not None
        served = set()
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(max(world, 8))
            while len(served) < world - 1:
                left = deadline - time.monotonic()
                if left <= 0:
                    raise TimeoutError(f"rank 0: only ranks {sorted(served)} of {world - 1} peers fetched the rendezvous blob "
                                       f"within {timeout:.0f} s")
                srv.settimeout(max(0.1, left))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                with conn:
                    try:
                        conn.settimeout(10.0)
                        peer = int.from_bytes(_recv_exact(conn, 4), "little")
                        conn.sendall(len(payload).to_bytes(4, "little") + payload)
                        if _recv_exact(conn, 1) == b"\x06" and 0 < peer < world:
                            served.add(peer)
                    except OSError:
                        pass  # that peer will come back
        return payload
    last = None
    while time.monotonic() < deadline:
        try:
            with socket.create_connection((addr, port), timeout=max(0.1, min(10.0, deadline - time.monotonic()))) as c:
                c.sendall(int(rank).to_bytes(4, "little"))
                n = int.from_bytes(_recv_exact(c, 4), "little")
                buf = _recv_exact(c, n)
                c.sendall(b"\x06")
                return buf
        except OSError as e:  # rank 0 is not listening yet, or a short read: try again
            last = e
        time.sleep(0.05)
    raise TimeoutError(f"rank {rank}: no rendezvous with rank 0 at {addr}:{port} within {timeout:.0f} s ({last})")


def exchange_from_rank0(payload: bytes | None, *, key: str = "msiren_comm_id", timeout: float = 300.0, env=None) -> bytes:
    """Every rank returns rank 0's ``payload``.  Uses the environment a launcher set up (module docstring).

    Under torchrun the agent already serves a TCP store on MASTER_PORT (``TORCHELASTIC_USE_AGENT_STORE``): the blob
    goes through it (torch.distributed.TCPStore as a client; no GPU runtime is touched).  Under
    :func:`spawn_ranks` MASTER_PORT is ours: rank 0 serves the blob itself over a plain socket -- no torch at all.
    """
    env = os.environ if env is None else env
    rank, world = int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1"))
    if world == 1:
        assert payload is not None
        return payload
    addr, port = env.get("MASTER_ADDR", "127.0.0.1"), int(env["MASTER_PORT"])
    if env.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true":
        from datetime import timedelta

        from torch.distributed import TCPStore

        store = TCPStore(addr, port, world, is_master=False, timeout=timedelta(seconds=timeout), wait_for_workers=False)
        run_key = f"{key}/{env.get('TORCHELASTIC_RESTART_COUNT', '0')}"
        if rank == 0:
            store.set(run_key, payload)
            return payload
        return bytes(store.get(run_key))
    return _exchange_socket(payload, rank, world, addr, port, timeout)
