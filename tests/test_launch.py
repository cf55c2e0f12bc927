"""bench.py's own multi-rank launcher (mri_inr_amd/launch.py) on the CPU, with a stand-in worker:
environment contract, relay of rank 0's stdout, failure propagation, the unique-id rendezvous, and
the rule that WORLD_SIZE != --gpus is an error in every case."""
import io
import json
import os
import subprocess
import sys
import textwrap
import time

import pytest

from mri_inr_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    from mri_inr_amd.launch import exchange_from_rank0
    r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    blob = exchange_from_rank0(bytes(range(128)) if r == 0 else None, timeout=60)
    assert blob == bytes(range(128))
    print(json.dumps({k: os.environ.get(k) for k in %r} | {"argv": sys.argv[1:]}), flush=True)
    print("diagnostics of rank", r, file=sys.stderr, flush=True)
    sys.exit(int(os.environ.get("FAIL_RANK", "-1")) == r and 7 or 0)
""") % (ROOT, list(launch.ENV_KEYS))


@pytest.fixture()
def worker(tmp_path):
    p = tmp_path / "worker.py"
    p.write_text(WORKER)
    return str(p)


def test_spawn_ranks_sets_the_torchrun_environment_and_relays_rank0(worker):
    out, err = io.StringIO(), io.StringIO()
    rc, rank0 = launch.spawn_ranks([sys.executable, worker, "--gpus", "3"], 3, timeout=120, stdout=out, stderr=err)
    assert rc == 0
    line = json.loads(rank0.strip())
    assert out.getvalue() == rank0  # rank 0's stdout verbatim, nothing else on stdout
    assert line["RANK"] == "0" and line["LOCAL_RANK"] == "0" and line["WORLD_SIZE"] == "3"
    assert line["MASTER_ADDR"] == "127.0.0.1" and int(line["MASTER_PORT"]) > 0 and line["argv"] == ["--gpus", "3"]
    e = err.getvalue()
    for r in (1, 2):  # the other ranks' output goes to stderr, prefixed
        rec = [json.loads(l.split("] ", 1)[1]) for l in e.splitlines() if l.startswith(f"[rank {r}] {{")]
        assert len(rec) == 1 and rec[0]["RANK"] == str(r) and rec[0]["MASTER_PORT"] == line["MASTER_PORT"]
    assert "[rank 0] diagnostics of rank 0" in e


def test_spawn_ranks_propagates_the_worst_exit_code(worker):
    env = dict(os.environ, FAIL_RANK="1")
    rc, _ = launch.spawn_ranks([sys.executable, worker], 2, timeout=120, env=env, stdout=io.StringIO(), stderr=io.StringIO())
    assert rc == 7


def test_spawn_ranks_kills_survivors_of_a_failed_rank(tmp_path):
    p = tmp_path / "hang.py"
    p.write_text("import os, sys, time\nif os.environ['RANK'] == '0': sys.exit(3)\ntime.sleep(600)\n")
    rc, _ = launch.spawn_ranks([sys.executable, str(p)], 2, timeout=120, stdout=io.StringIO(), stderr=io.StringIO())
    assert rc == 3  # returned promptly: the sleeping rank was terminated


def test_rank_dying_inside_the_broadcast_takes_the_job_down_within_30_s(tmp_path):
    """After the bootstrap every rank sits in the weight broadcast -- a native call that returns only when all ranks have
    joined.  If one dies there (a GPU fault, an OOM kill), the others would wait for ever: the launcher must end the job,
    also when a survivor does not react to SIGTERM (blocked in native code with the signal ignored -> SIGKILL after 10 s)."""
    p = tmp_path / "bcast.py"
    p.write_text(textwrap.dedent("""
        import ctypes, os, signal, sys, time
        sys.path.insert(0, %r)
        from mri_inr_amd.launch import exchange_from_rank0
        r = int(os.environ["RANK"])
        exchange_from_rank0(b"id" if r == 0 else None, timeout=60)      # the bootstrap succeeds on every rank
        if r == 1:
            signal.signal(signal.SIGTERM, signal.SIG_IGN)               # a survivor that cannot be asked nicely
        print("rank", r, "enters the broadcast", file=sys.stderr, flush=True)
        open(os.path.join(%r, "in.%%d" %% r), "w").close()
        if r == 2:
            # dies inside the collective -- once the other two are in it as well (on a busy box the launcher, rightly, takes the
            # job down the moment this rank is gone: a rank that has not printed yet would never be seen entering)
            t0 = time.monotonic()
            while not all(os.path.exists(os.path.join(%r, "in.%%d" %% k)) for k in (0, 1)) and time.monotonic() - t0 < 60:
                time.sleep(0.01)
            os.kill(os.getpid(), signal.SIGKILL)
        ctypes.CDLL(None).sleep(600)                                    # blocked in native code, like ncclBroadcast
    """) % (ROOT, str(tmp_path), str(tmp_path)))
    err = io.StringIO()
    t0 = time.monotonic()
    rc, _ = launch.spawn_ranks([sys.executable, str(p)], 3, timeout=300, stdout=io.StringIO(), stderr=err)
    took = time.monotonic() - t0
    assert rc != 0 and took < 30, (rc, took)
    assert err.getvalue().count("enters the broadcast") == 3
    # nothing is left behind
    time.sleep(0.2)
    out = subprocess.run(["pgrep", "-f", str(p)], capture_output=True, text=True).stdout.split()
    assert out == [], out


def test_under_launcher_and_single_rank_exchange():
    assert not launch.under_launcher({})
    assert launch.under_launcher({"RANK": "0", "WORLD_SIZE": "2"})
    assert launch.exchange_from_rank0(b"abc", env={}) == b"abc"


def _bench(args, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in launch.ENV_KEYS}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    # under a launcher (WORLD_SIZE set) the rank count must match --gpus: no silent single-GPU run
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    r = _bench(["--gpus", "1", "--steps", "1", "--warmup", "0"], env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_gpus2_without_a_launcher_starts_two_ranks():
    """No GPU in the CPU container: both ranks must come up (and say why they stop) -- the point is that the
    parent spawned TWO of them instead of measuring one GPU, and that it reports their failure."""
    import mri_inr_amd._lib as L

    if L.device_count() > 0:
        pytest.skip("a GPU is visible: the real run is covered by the GPU tests")
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0
    # both ranks were started (whichever fails first takes the other down, so only one message is guaranteed)
    assert "[launch] rank 0 of 2 started" in r.stderr and "[launch] rank 1 of 2 started" in r.stderr
    assert "needs a gfx950 GPU" in r.stderr
    assert "n_gpus" not in r.stdout


def test_rendezvous_serves_until_every_rank_has_acknowledged():
    """A peer whose connection dies mid-payload comes back: rank 0 counts acknowledged RANKS, not accepted connections,
    so the retry does not use up the turn of the last rank (round-2 advice: it used to time out after 300 s)."""
    import socket
    import threading

    port = launch.free_port()
    payload = bytes(range(200))
    got = {}

    def serve():
        got[0] = launch._exchange_socket(payload, 0, 3, "127.0.0.1", port, 30.0)

    def peer(rank):
        got[rank] = launch._exchange_socket(None, rank, 3, "127.0.0.1", port, 30.0)

    t0 = threading.Thread(target=serve)
    t0.start()
    # rank 1's first attempt: connects, announces itself, reads 10 bytes and hangs up without the ack
    deadline = time.monotonic() + 20
    while True:
        try:
            with socket.create_connection(("127.0.0.1", port), timeout=2) as c:
                c.sendall((1).to_bytes(4, "little"))
                c.recv(10)
            break
        except OSError:
            assert time.monotonic() < deadline
            time.sleep(0.05)
    peers = [threading.Thread(target=peer, args=(r,)) for r in (1, 2)]
    for t in peers:
        t.start()
    for t in peers + [t0]:
        t.join(40)
        assert not t.is_alive()
    assert got == {0: payload, 1: payload, 2: payload}
