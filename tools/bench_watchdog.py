"""Watchdog of bench.py's N > 1 runs: every rank the launcher starts is a SUPERVISOR that never touches the GPU; the
measuring process is its child.  A hang inside the library's own RCCL calls (which torch's NCCL watchdog does not see)
therefore ends as a killed child and a fresh one at the next stage of a fallback ladder -- and the run still prints
ONE JSON line, labelled with the stage it came from, instead of running into the driver's time limit with nothing.

    stage 1   library driver (dsea_pop_*), slab exchange overlapped, on its own communicator and side stream
    stage 2   library driver, exchange after the correction pass, ONE communicator for exchange and all-reduces
    stage 3   Python driver over the torch process group, pairwise exchange, no overlap

How a stall is seen: the child appends a line to its progress file after every phase and every step (bench.py
``progress()``); a rank whose file has not grown for ``stall_s`` (``startup_s`` before the first line) -- or whose child
exited non-zero -- reports failure.  The supervisors agree once a second through a two-element gloo all-reduce
(CPU tensors), so all of them kill their children (whole process groups) and move on together.  If rank 0's child had
already recorded the timed point ("provisional" line: the extras beside it were still running), that line is printed
with a note and no further stage is started.

Deadlines: ``stall_s`` = max(150 s, 20 x the longest one-GPU anchor step) -- an N-GPU step is shorter than the one-GPU
step of the same point; ``startup_s`` = 420 s (first ``import torch`` on a fresh box takes up to two minutes, RCCL
initialisation tens of seconds); the whole ladder must fit ``budget_s`` = 1500 s (the driver allows 1800 s): a stage is
only started while startup_s + stall_s still fit.  DSEA_BENCH_STALL_S / _STARTUP_S / _BUDGET_S override (tests)."""
import datetime
import json
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time

STAGES = (
    {"name": "library driver, overlapped slab exchange on its own communicator and stream", "env": {}},
    {"name": "library driver, exchange after the correction pass, one communicator for exchange and all-reduces",
     "env": {"DSEA_COMM_SINGLE": "1", "DSEA_BENCH_OVERLAP": "off", "DSEA_BENCH_REDUCED": "1"}},
    {"name": "Python driver over the torch process group, pairwise slab exchange, no overlap",
     "env": {"DSEA_DRIVER": "python", "DSEA_BENCH_OVERLAP": "off", "DSEA_BENCH_PAIRWISE": "1", "DSEA_BENCH_REDUCED": "1"}},
)
# DSEA_BENCH_INJECT_HANG (tests of this file): which stages the injected hang hits
#   exchange   -- the second communicator's point-to-point traffic: stage 1 only
#   allreduce  -- the library's all-reduce: stages 1 and 2 (stage 3 does not use the library's communicators)
#   extras     -- (dry run only) a stall AFTER the timed point, in the points reported beside it: stage 1's line is kept
#   crash      -- (dry run only) rank 1's child raises in stage 1
INJECT_STAGES = {"exchange": (1,), "allreduce": (1, 2), "extras": (1,), "crash": (1,)}
INJECT_FAKE_RCCL = {"exchange": "p2p@comm1:40", "allreduce": "allreduce:60"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _kill_group(proc):
    if proc is None or proc.poll() is not None:
        return
    try:
        os.killpg(proc.pid, signal.SIGKILL)
    except (ProcessLookupError, PermissionError):
        try:
            proc.kill()
        except ProcessLookupError:
            pass
    try:
        proc.wait(timeout=30)
    except subprocess.TimeoutExpired:
        pass


def _read_progress(path):
    events = []
    try:
        with open(path) as f:
            for ln in f:
                try:
                    events.append(json.loads(ln))
                except ValueError:
                    pass
    except OSError:
        pass
    return events


def deadlines(anchors):
    """(startup_s, stall_s, budget_s, how) -- see the module docstring"""
    longest = 0.0
    for a in (anchors or {}).values():
        if isinstance(a, dict) and a.get("ms_per_step"):
            longest = max(longest, float(a["ms_per_step"]) * 1e-3)
    stall = max(150.0, 20.0 * longest)
    how = "stall = max(150 s, 20 x longest one-GPU anchor step %.2f s)" % longest
    startup, budget = 420.0, 1500.0
    if os.environ.get("DSEA_BENCH_STALL_S"):
        stall, how = float(os.environ["DSEA_BENCH_STALL_S"]), "DSEA_BENCH_STALL_S"
    if os.environ.get("DSEA_BENCH_STARTUP_S"):
        startup = float(os.environ["DSEA_BENCH_STARTUP_S"])
    if os.environ.get("DSEA_BENCH_BUDGET_S"):
        budget = float(os.environ["DSEA_BENCH_BUDGET_S"])
    return startup, stall, budget, how


def supervise(script, argv, anchors, describe):
    """Runs in every rank the launcher started (RANK / WORLD_SIZE / MASTER_* in the environment).  Does not return:
    exits with 0 after rank 0 has printed the final JSON line, 1 when no stage produced one.
    ``describe``: dict with the keys of a failure line (metric, unit, n_gpus, steps, warmup ...)."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    t_start = time.time()
    startup_s, stall_s, budget_s, how = deadlines(anchors)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=budget_s + 600))
    inject = os.environ.get("DSEA_BENCH_INJECT_HANG", "")
    tmpdir = tempfile.mkdtemp(prefix="dsea_bench_r%d_" % rank)
    log, final, used_stage, reason = [], None, None, None
    for s, stage in enumerate(STAGES, 1):
        elapsed = time.time() - t_start
        go = torch.tensor([1.0 if elapsed + startup_s + stall_s <= budget_s else 0.0])
        dist.all_reduce(go, op=dist.ReduceOp.MIN)
        if go.item() == 0.0:
            log.append({"stage": s, "outcome": "not started: %.0f s of the %.0f s budget used" % (elapsed, budget_s)})
            break
        port = [_free_port() if rank == 0 else None]
        dist.broadcast_object_list(port, src=0)
        prog = os.path.join(tmpdir, "progress_stage%d.jsonl" % s)
        out_path = os.path.join(tmpdir, "stdout_stage%d.txt" % s)
        env = dict(os.environ)
        env.update(stage["env"])
        env.update(MASTER_PORT=str(port[0]), DSEA_BENCH_CHILD="1", DSEA_BENCH_STAGE=str(s), DSEA_BENCH_PROGRESS=prog,
                   TORCHELASTIC_USE_AGENT_STORE="False")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if inject:
            if s in INJECT_STAGES.get(inject, ()):
                if inject in INJECT_FAKE_RCCL:
                    env["FAKE_RCCL_HANG"] = INJECT_FAKE_RCCL[inject]     # the stand-in RCCL of the rehearsal
                env["DSEA_BENCH_SIMULATE_HANG"] = inject                  # the dry run (no communicator library at all)
            else:
                env.pop("FAKE_RCCL_HANG", None)
                env.pop("DSEA_BENCH_SIMULATE_HANG", None)
        t_stage = time.time()
        with open(out_path, "w") as fout:
            proc = subprocess.Popen([sys.executable, script] + list(argv), env=env, start_new_session=True,
                                    stdout=fout if rank == 0 else None)
        last_size, last_change, seen_first = -1, time.time(), False
        outcome = None
        while True:
            time.sleep(1.0 if stall_s >= 30 else 0.25)
            rc = proc.poll()
            try:
                size = os.path.getsize(prog)
            except OSError:
                size = 0
            if size != last_size:
                last_size, last_change = size, time.time()
                seen_first = seen_first or size > 0
            mine_ok, mine_bad, why = 0.0, 0.0, None
            if rc is not None:
                if rc == 0:
                    mine_ok = 1.0
                else:
                    mine_bad, why = 1.0, "rank %d: child exited with code %d" % (rank, rc)
            else:
                limit = stall_s if seen_first else startup_s
                if time.time() - last_change > limit:
                    ev = _read_progress(prog)
                    where = ev[-1].get("event", "?") if ev else "start-up (no progress line yet)"
                    mine_bad, why = 1.0, "rank %d: no progress for %.0f s after '%s'" % (rank, limit, where)
            t = torch.tensor([mine_ok, mine_bad])
            dist.all_reduce(t)
            if t[1].item() > 0:
                whys = [None] * world
                dist.all_gather_object(whys, why)
                _kill_group(proc)
                outcome = "; ".join(w for w in whys if w)
                break
            if t[0].item() == world:
                outcome = "completed"
                break
        events = _read_progress(prog)
        rec = {"stage": s, "what": stage["name"], "outcome": outcome, "wall_s": round(time.time() - t_stage, 1),
               "last_event_rank0": None}
        have = [None]
        if rank == 0:
            rec["last_event_rank0"] = events[-1].get("event") if events else None
            text = open(out_path).read()
            lines = [ln for ln in text.splitlines() if ln.startswith("{") and '"metric"' in ln]
            for ln in text.splitlines():
                if not (ln.startswith("{") and '"metric"' in ln):
                    print(ln)
            if lines and outcome == "completed":
                have[0] = json.loads(lines[-1])
            else:
                prov = [e for e in events if e.get("event") == "provisional_line"]
                if prov:
                    have[0] = prov[-1]["line"]
                    have[0].setdefault("config", {})["extras_incomplete"] = \
                        "the timed point was complete; what ran beside it was cut short (%s)" % outcome
        log.append(rec)
        dist.broadcast_object_list(have, src=0)
        if have[0] is not None:
            final, used_stage = have[0], s
            break
        reason = outcome
    rc_exit = 0
    if rank == 0:
        if final is None:
            final = dict(describe)
            final.update(value=None, ms_per_step=None, error="no stage of the N > 1 ladder produced a timed point")
            final.setdefault("config", {})
            rc_exit = 1
        cfg = final.setdefault("config", {})
        cfg["fallback_stage"] = used_stage
        cfg["fallback_stage_is"] = STAGES[used_stage - 1]["name"] if used_stage else None
        cfg["fallback_reason"] = reason if (used_stage or 0) > 1 or final.get("value") is None else None
        cfg["watchdog"] = {"stages": log, "startup_s": startup_s, "stall_s": stall_s, "budget_s": budget_s, "deadline_rule": how,
                           "total_wall_s": round(time.time() - t_start, 1)}
        sys.stdout.flush()
        print(json.dumps(final), flush=True)
    import shutil
    shutil.rmtree(tmpdir, ignore_errors=True)
    code = torch.tensor([float(rc_exit)])
    dist.broadcast(code, src=0)
    try:
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        pass
    raise SystemExit(int(code.item()))
