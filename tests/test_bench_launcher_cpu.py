"""``python bench.py --gpus N`` launches itself: the parent starts N one-GPU child processes with
the environment ``torch.distributed.run`` would give them, relays rank 0's single JSON line and returns the worst
exit code; ``--gpus N`` beyond the visible devices is refused instead of silently measuring one GPU.  The ranks here
run ``tests/tools/bench_double.py`` = bench.py on the oracle-backed device double, shortlists over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DOUBLE = os.path.join(ROOT, "tests", "tools", "bench_double.py")
SMALL = ["--N", "96", "--d", "3", "--M", "3000", "--steps", "1", "--warmup", "1", "--extras", "off",
         "--cpu-baseline", "off"]


def _launch(n, extra_args=(), env_extra=None, timeout=240):
    """Run ``bench.launch_ranks`` in a fresh interpreter (it installs signal handlers and prints the relayed line)."""
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.launch_ranks(%d, %r, script=%r, timeout=%r))"
            % (ROOT, n, ["--gpus", str(n)] + SMALL + list(extra_args), DOUBLE, timeout))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=timeout + 120)


@pytest.mark.parametrize("n", [2, 3])
def test_self_launched_ranks_print_one_line_for_n_gpus(n):
    out = _launch(n, ["--allow-gloo"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == n and res["config"]["comm"] == "gloo-fallback" and res["config"]["rccl_ranks"] == 0
    assert res["config"]["sharding"] == f"candidates x{n}" and res["config"]["M_total"] == 3000
    assert res["config"]["M_per_gpu"] == -(-3000 // n) and res["config"]["N_train_per_step"] == [96]
    assert res["value"] > 0 and res["scaling"] == "strong"


def test_without_rccl_the_ranks_exit_3_unless_gloo_is_allowed():
    out = _launch(2)
    assert out.returncode == 3, (out.returncode, out.stderr[-2000:])
    assert not [ln for ln in out.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert "no RCCL communicator" in out.stderr


def test_a_failing_rank_ends_the_group_with_its_code():
    out = _launch(2, ["--allow-gloo"], {"GPRY_BENCH_DOUBLE_FAIL_RANK": "1", "GPRY_BENCH_FAIL_GRACE": "3"})
    assert out.returncode == 7, (out.returncode, out.stderr[-2000:])
    assert not out.stdout.strip()


def test_a_rank_that_fails_late_leaves_rank_0s_line_with_an_error_object():
    """rank 0 prints before the closing barriers, so a peer that dies behind the timed region cannot take
    the line with it; the launcher adds which rank failed, the exit codes and the tail of that rank's stderr."""
    out = _launch(2, ["--allow-gloo"], {"GPRY_BENCH_DOUBLE_FAIL_LATE_RANK": "1", "GPRY_BENCH_FAIL_GRACE": "3"})
    assert out.returncode == 9, (out.returncode, out.stderr[-2000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["value"] > 0 and res["n_gpus"] == 2 and res["config"]["rccl_ranks"] == 0 and res["config"]["M_per_gpu"] == 1500
    assert res["error"]["failed_ranks"] == [1] and res["error"]["exit_codes"][1] == 9
    assert any("fails behind the timed region" in ln for ln in res["error"]["stderr_tail"]["1"])


def test_a_rank_that_fails_after_the_rendezvous_leaves_rank_0s_snapshot():
    """... and one that dies in its first step, when rank 0 has no result yet: the launcher prints rank 0's last snapshot
    (transport, ``rccl_ranks``, the shard of every rank) marked ``partial`` with the same error object."""
    out = _launch(2, ["--allow-gloo"], {"GPRY_BENCH_DOUBLE_FAIL_MID_RANK": "1", "GPRY_BENCH_FAIL_GRACE": "3"})
    assert out.returncode == 8, (out.returncode, out.stderr[-2000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["partial"] == "after the rendezvous" and res["value"] is None
    assert res["config"]["M_per_gpu_by_rank"] == [1500, 1500] and res["config"]["comm"] == "gloo-fallback"
    assert res["error"]["failed_ranks"] == [1] and res["error"]["exit_codes"][1] == 8


def test_the_watchdog_ends_a_hanging_run():
    out = _launch(2, ["--allow-gloo"], {"GPRY_BENCH_DOUBLE_HANG_RANK": "1"}, timeout=12)
    assert out.returncode != 0 and "time limit" in out.stderr
    assert not out.stdout.strip()


def test_more_gpus_than_visible_is_refused(monkeypatch, capsys):
    import bench
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GPRY_HIP_DEVICE_WRAP"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(bench, "visible_gpus", lambda: 1)
    called = []
    monkeypatch.setattr(bench, "launch_ranks", lambda n, argv, **kw: called.append((n, argv)) or 0)
    for wl in (["--workload", "cycle"], ["--workload", "farm"], ["--workload", "farm", "--mode", "group"]):
        with pytest.raises(SystemExit) as e:
            bench.main(["--gpus", "8"] + wl)
        assert e.value.code == 2 and "only 1 GPU(s) are visible" in capsys.readouterr().err
    assert not called
    # development boxes: ranks may share a device when asked to
    monkeypatch.setenv("GPRY_HIP_DEVICE_WRAP", "1")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2", "--steps", "3"])
    assert e.value.code == 0 and called == [(2, ["--gpus", "2", "--steps", "3"])]
    # a launcher that disagrees with --gpus is an error, not a silent 1-GPU line
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4"])
    assert e.value.code == 2


def _store_world(world, port, occupy=None):
    """the ranks of a ``bench._TcpStore`` as threads of this process: what every rank saw"""
    import socket
    import threading
    import bench
    foreign = None
    if occupy is not None:          # some other service on the first candidate port: it answers, but not with the handshake word
        foreign = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        foreign.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        foreign.bind(("127.0.0.1", port + occupy))
        foreign.listen(8)

        def chatter():
            while True:
                try:
                    c, _ = foreign.accept()
                except OSError:
                    return
                try:
                    c.sendall(b"\x05\x00\x00\x00hello")
                finally:
                    c.close()
        threading.Thread(target=chatter, daemon=True).start()
    seen, errors = {}, []

    def run(rank):
        try:
            st = bench._TcpStore(rank, world, addr="127.0.0.1", port=port, timeout=30)
            uid = st.broadcast(bytes(range(128)) if rank == 0 else b"")
            low = st.all_min(1.0 if rank != 1 else 0.0)          # rank 1 "has no communicator"
            st.barrier()
            high = st.all_min(5.0 + rank)
            st.barrier()
            st.destroy_process_group()
            seen[rank] = (uid, low, high)
        except Exception as e:      # pragma: no cover
            errors.append((rank, repr(e)))

    ths = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ths[::-1]:             # the server last: the others retry until it listens
        t.start()
    for t in ths:
        t.join(60)
    if foreign is not None:
        foreign.close()
    assert not errors, errors
    return seen


def _free_port_block():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_the_tcp_store_of_the_rccl_rendezvous_needs_no_torch():
    """the 128-byte RCCL id, the agreement on the transport and the barriers of ``bench.py --gpus N`` go
    over a small TCP store next to MASTER_PORT -- no ``torch`` on the RCCL path (gloo only behind ``--allow-gloo``)."""
    import bench
    for world, occupy in ((3, None), (2, 1)):
        seen = _store_world(world, _free_port_block(), occupy)
        assert sorted(seen) == list(range(world))
        for r in range(world):
            uid, low, high = seen[r]
            assert uid == bytes(range(128)) and low == (0.0 if world > 1 else 1.0) and high == 5.0
    # the RCCL path of the bench imports no torch: `connect` reaches for it only behind --allow-gloo
    import inspect
    src = inspect.getsource(bench.connect)
    assert src.count("import torch") == 1 and src.index("import torch") > src.index("using gloo (--allow-gloo)")
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "st = bench._TcpStore(0, 1, addr='127.0.0.1', port=%d, timeout=5); st.barrier(); st.destroy_process_group(); "
            "assert 'torch' not in sys.modules, 'torch was imported'" % (ROOT, _free_port_block()))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1500:]
