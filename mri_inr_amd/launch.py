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


# ---- optional: rank -> CPU affinity by the GPU's NUMA node (off by default) -------------------------------------------------
# One process per GPU; on a two-socket node a rank whose host threads (launches, pageable copies, the rendezvous) run on the
# other socket pays a cross-socket hop for every doorbell and every staged copy.  `bench.py --numa-pin` pins each rank, BEFORE it
# touches HIP (threads the runtime starts later inherit the mask), to the CPUs of the NUMA node its GPU hangs off:
#   PCI bus id of local rank r  <-  MSIREN_RANK_PCI_BUSIDS (comma-separated, by local rank) or `rocm-smi --showbus --json`
#   NUMA node                   <-  /sys/bus/pci/devices/<busid>/numa_node        (-1: unknown -> no pinning)
#   its CPUs                    <-  /sys/devices/system/node/node<N>/cpulist, intersected with the mask the process already has
# Nothing here imports torch or the HIP library; the sysfs root and the smi runner are parameters so that the CPU tests can
# hand in a fake tree.  No scaling run on a multi-GPU node has exercised this: it is reported in config.ranks[], not claimed.

def parse_cpulist(text: str) -> list[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    cpus = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def format_cpulist(cpus) -> str:
    cpus = sorted(set(cpus))
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def gpu_busids_from_smi(runner=None) -> list[str]:
    """PCI bus ids of the GPUs in device order, from `rocm-smi --showbus --json` (a subprocess: this process stays off the GPU).
    Empty list when the tool is missing or its output is not understood."""
    import json
    import re

    runner = runner or (lambda: subprocess.run(["rocm-smi", "--showbus", "--json"], capture_output=True, text=True, timeout=30).stdout)
    try:
        data = json.loads(runner())
    except Exception:  # noqa: BLE001 -- optional feature: any failure means "unknown"
        return []
    ids = []
    for key in sorted((k for k in data if re.fullmatch(r"card\d+", k)), key=lambda k: int(k[4:])):
        val = next((v for kk, v in data[key].items() if "bus" in kk.lower()), None)
        if not isinstance(val, str) or not re.fullmatch(r"[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-7]", val.strip()):
            return []
        ids.append(val.strip().lower())
    return ids


def numa_node_of(busid: str, sysfs: str = "/sys") -> int:
    try:
        with open(os.path.join(sysfs, "bus", "pci", "devices", busid.lower(), "numa_node")) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return -1


def cpus_of_node(node: int, sysfs: str = "/sys") -> list[int]:
    try:
        with open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")) as f:
            return parse_cpulist(f.read())
    except (OSError, ValueError):
        return []


def rank_affinity(local_rank: int, *, busids=None, sysfs: str = "/sys", allowed=None, env=None) -> dict | None:
    """{'pci_bus_id', 'numa_node', 'cpus'} for a local rank, or None when anything is unknown (then nothing is pinned)."""
    env = os.environ if env is None else env
    if busids is None:
        listed = [b.strip() for b in env.get("MSIREN_RANK_PCI_BUSIDS", "").split(",") if b.strip()]
        busids = listed or gpu_busids_from_smi()
    vis = env.get("HIP_VISIBLE_DEVICES") or env.get("ROCR_VISIBLE_DEVICES")
    if vis and all(t.strip().isdigit() for t in vis.split(",")):   # the runtime renumbers the visible devices
        order = [int(t) for t in vis.split(",")]
        busids = [busids[i] for i in order if i < len(busids)]
    if not 0 <= local_rank < len(busids):
        return None
    node = numa_node_of(busids[local_rank], sysfs)
    if node < 0:
        return None
    cpus = set(cpus_of_node(node, sysfs))
    if allowed is None and hasattr(os, "sched_getaffinity"):
        allowed = os.sched_getaffinity(0)
    if allowed is not None:
        cpus &= set(allowed)
    if not cpus:
        return None
    return {"pci_bus_id": busids[local_rank], "numa_node": node, "cpus": format_cpulist(cpus)}


def pin_rank(local_rank: int, *, setter=None, **kw) -> dict | None:
    """Pin the calling process to its GPU's NUMA node (call BEFORE the HIP runtime starts its threads).  Returns what was
    applied (for config.ranks[]) or None."""
    info = rank_affinity(local_rank, **kw)
    if info is None:
        return None
    setter = setter or (lambda cpus: os.sched_setaffinity(0, cpus))
    try:
        setter(set(parse_cpulist(info["cpus"])))
    except OSError:
        return None
    return info
