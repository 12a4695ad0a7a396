"""bench.py's RANK LOGIC on a host without GPUs: ``setup_ranks``' refusals, the barrier + max-over-ranks
timing of ``main`` (``--only-steps``), ``cfg4_line``'s ``reduce_max`` and gather, and the rank-0-only JSON line, under a
2-rank ``gloo`` group.  bench.py talks to the device through one object (``bench.RT``: HIP runtime + RCCL); the workers
below swap in an emulator-backed one -- CPU tensors, the .hip sources compiled against tests/hipemu, ``gloo`` -- so the code
that a real ``--gpus 8`` launch executes first runs here at toy shapes.  No timing claim is made with it."""
import io
import json
import os
import time
from contextlib import redirect_stdout

import pytest
import torch
import torch.multiprocessing as mp


class _EmuEvent:
    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class _EmuRuntime:
    """CPU tensors are the emulated device; 8 'devices' so that any LOCAL_RANK the test uses exists"""
    backend = "gloo"

    def available(self):
        return True

    def device_count(self):
        return 8

    def device(self, local_rank):
        return torch.device("cpu")

    def synchronize(self):
        pass

    def event(self):
        return _EmuEvent()

    def init_process_group(self, rank, world, device):
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)


def _install_emulator():
    import ctypes
    import bench
    from ddsp_svc_amd import _ffi
    from tests.hipemu import build as emu_build
    _ffi._LIB = _ffi.bind(ctypes.CDLL(emu_build.build()))
    _ffi.check_device = lambda *t: None
    bench.RT = _EmuRuntime()
    return bench


def _rank_worker(rank, world, port, port2, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": str(rank)})
    try:
        bench = _install_emulator()
        argv = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--batch-per-gpu", "1", "--seconds", "0.03", "--bins", "65",
                "--only-steps", "--cfg4-batch", "2"]
        # (1) main(): every rank steps, fences, all-reduces its times with MAX; only rank 0 queues a line
        buf = io.StringIO()
        with redirect_stdout(buf):
            bench.main(argv)
            queued = list(bench._PENDING)
            bench.flush_emit()
        lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("{")]
        # (2) cfg4_line on a fresh group (its own rendezvous port: the first group's store may still be closing): reduce_max +
        # the gather of every step's waveforms to rank 0
        os.environ["MASTER_PORT"] = str(port2)
        a = _parse(bench, argv)
        r, w, device, comm = bench.setup_ranks(a)
        F = int(a.seconds * bench.SR) // bench.HOP + 1
        res = bench.cfg4_line(a, r, w, device, F, a.bins, comm)
        bench.finish_ranks()
        q.put((rank, "ok", {"queued": len(queued), "lines": lines, "cfg4": res, "comm": comm}))
    except BaseException as e:                       # SystemExit included: report, the parent asserts
        import traceback
        q.put((rank, "error", "%s: %s\n%s" % (type(e).__name__, e, traceback.format_exc())))


def _parse(bench, argv):
    """bench.main's own parser, without running main: cfg4_line and setup_ranks take its namespace"""
    import argparse
    captured = {}
    orig = argparse.ArgumentParser.parse_args

    def grab(self, args=None, namespace=None):
        captured["a"] = orig(self, args, namespace)
        raise _Stop()
    argparse.ArgumentParser.parse_args = grab
    try:
        bench.main(argv)
    except _Stop:
        pass
    finally:
        argparse.ArgumentParser.parse_args = orig
    return captured["a"]


class _Stop(Exception):
    pass


def test_two_ranks_step_reduce_and_emit():
    import socket
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk, socket.socket() as sk2:   # two ports that are free NOW (xdist workers run other rendezvous beside this one)
        sk.bind(("127.0.0.1", 0))
        sk2.bind(("127.0.0.1", 0))
        port, port2 = sk.getsockname()[1], sk2.getsockname()[1]
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, port2, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in procs:
        rank, status, payload = q.get(timeout=300)
        res[rank] = (status, payload)
    for p in procs:
        p.join(timeout=60)
    assert all(s == "ok" for s, _ in res.values()), res
    r0, r1 = res[0][1], res[1][1]
    # rank 0 alone emits, exactly one line, and it reports the WORLD's size
    assert r0["queued"] == 1 and r1["queued"] == 0 and len(r0["lines"]) == 1 and not r1["lines"]
    line = json.loads(r0["lines"][0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["only_steps"] is True and line["ms_per_step"] > 0
    # the communicator was proven with an all-reduce over both ranks
    assert r0["comm"]["rccl_ranks"] == 2 and r1["comm"]["rccl_ranks"] == 2
    # cfg 4: both ranks hold the SAME (max-reduced) times, and rank 0 checked the gathered batch of 2 x 2 utterances
    c0, c1 = r0["cfg4"], r1["cfg4"]
    for k in ("ms_per_step", "ms_per_step_with_gather", "gather_ms", "value", "value_with_gather"):
        assert c0[k] == c1[k] and c0[k] > 0, (k, c0[k], c1[k])
    assert c0["n_gpus"] == 2 and c0["batch_per_gpu"] == 2 and c0.get("gather_checked_rows") == 4 and "gather_checked_rows" not in c1
    # ... and the gather without a copy of the local shard ran on both ranks (rank 0's synthesis wrote into its slice of the result)
    assert "cfg4_in_place_error" not in r0["comm"] and "cfg4_in_place_error" not in r1["comm"], (r0["comm"], r1["comm"])
    assert c0["ms_per_step_with_gather_in_place"] == c1["ms_per_step_with_gather_in_place"] and c0["ms_per_step_with_gather_in_place"] > 0
    T = (int(0.03 * 44100) // 512 + 1) * 512
    assert abs(c0["value"] - 2 * 2 * T * c0["steps"] / (c0["ms_per_step"] * 1e-3 * c0["steps"])) <= 1e-6 * c0["value"]


def _refusal_worker(env, argv, q):
    os.environ.update(env)
    try:
        bench = _install_emulator()
        with redirect_stdout(io.StringIO()):
            bench.main(argv)
        q.put(("returned", ""))
    except SystemExit as e:
        q.put(("exit", str(e)))
    except BaseException as e:
        q.put(("error", "%s: %s" % (type(e).__name__, e)))


@pytest.mark.parametrize("env,argv,needle", [
    ({"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"}, ["--gpus", "4", "--only-steps"], "refusing to report a different GPU count"),
    ({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"}, ["--gpus", "0"], "--gpus must be >= 1"),
    ({"RANK": "9", "WORLD_SIZE": "16", "LOCAL_RANK": "9"}, ["--gpus", "16", "--only-steps"], "needs 16 devices"),
])
def test_setup_ranks_refuses(env, argv, needle):
    """a world size that is not --gpus, a GPU count below one, a local rank without a device: SystemExit with the reason,
    never a silent run at another size"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_refusal_worker, args=(env, argv, q))
    p.start()
    status, msg = q.get(timeout=300)
    p.join(timeout=60)
    assert status == "exit" and needle in msg, (status, msg)


def test_no_gpu_no_run():
    """the product's own runtime object: on a host without an MI355X bench.py refuses (no CPU path of its own)"""
    if torch.cuda.is_available():
        pytest.skip("this host has a GPU")
    import bench
    assert isinstance(bench.RT, bench._HipRuntime) and bench.RT.backend == "nccl"
    env = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK") if k in os.environ}
    try:
        with pytest.raises(SystemExit) as e:
            bench.main(["--gpus", "1", "--only-steps"])
        assert "no CPU fallback" in str(e.value)
    finally:
        os.environ.update(env)


def _gate_worker(q):
    try:
        bench = _install_emulator()
        B, F, n = 3, 9, 65
        step, inp = bench.build_step("combsub", B, F, n, torch.device("cpu"), seed=5)
        out = step()
        rec = bench.parity_gate("combsub", inp["f0"], inp["ctrls"], inp["noise"], out)
        broken = out.clone()
        broken[-1, 100:] = out[-1, :-100]                # the last utterance shifted by 100 samples: a wrong row offset
        try:
            bench.parity_gate("combsub", inp["f0"], inp["ctrls"], inp["noise"], broken)
            refused = False
        except SystemExit as e:
            refused = "NOT the reference's" in str(e)
        q.put(("ok", rec, refused))
    except BaseException as e:
        import traceback
        q.put(("error", "%s: %s\n%s" % (type(e).__name__, e, traceback.format_exc()), False))


def test_parity_gate_refuses_a_wrong_output():
    """bench.py prints no timing for an output that is not the reference's: the gate passes the emulated step
    and raises SystemExit for the same output with its last utterance shifted"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_gate_worker, args=(q,))
    p.start()
    status, rec, refused = q.get(timeout=300)
    p.join(timeout=60)
    assert status == "ok", rec
    assert rec["rows"] == [0, 2] and rec["rms_abs"] <= 1e-4 and rec["rms_rel"] <= 1e-5
    assert refused
