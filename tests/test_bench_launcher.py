"""bench.py --gpus N without a launcher around it: bench.py starts the N ranks itself.  Driven here with a stub child
(AMV_BENCH_CHILD) -- no GPU, no torch.distributed: what is checked is the environment every rank gets, that rank 0's one
line is relayed alone on stdout, and that the worst exit code comes back."""
import json
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _stub(tmp_path, body):
    path = tmp_path / "child.py"
    path.write_text(textwrap.dedent(body))
    return "%s %s" % (sys.executable, path)


def _run(args, child=None, extra_env=None, timeout=120):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    if child:
        env["AMV_BENCH_CHILD"] = child
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_launcher_sets_ranks_relays_the_line_and_the_arguments(tmp_path):
    child = _stub(tmp_path, """
        import json, os, sys
        rank = int(os.environ["RANK"])
        rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}
        open(os.path.join(%r, "rank%%d.json" %% rank), "w").write(json.dumps({"env": rec, "argv": sys.argv[1:]}))
        print("noise from rank %%d" %% rank, file=sys.stderr)
        if rank == 0:
            print(json.dumps({"metric": "stub", "n_gpus": int(os.environ["WORLD_SIZE"])}))
        else:
            print("a stray stdout line of rank %%d" %% rank)
    """ % str(tmp_path))
    r = _run(["--gpus", "3", "--steps", "2", "--warmup", "1"], child)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout) == {"metric": "stub", "n_gpus": 3}          # rank 0's line, alone
    assert "a stray stdout line of rank 1" in r.stderr                      # the others' stdout never reaches ours
    seen = [json.load(open(tmp_path / ("rank%d.json" % k))) for k in range(3)]
    ports = {s["env"]["MASTER_PORT"] for s in seen}
    assert len(ports) == 1 and int(ports.pop()) > 0
    for k, s in enumerate(seen):
        assert s["env"]["RANK"] == s["env"]["LOCAL_RANK"] == str(k)
        assert s["env"]["WORLD_SIZE"] == "3" and s["env"]["MASTER_ADDR"] == "127.0.0.1"
        assert s["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert s["argv"] == ["--gpus", "3", "--steps", "2", "--warmup", "1"]


def test_launcher_returns_the_worst_exit_code_and_still_relays_the_line(tmp_path):
    child = _stub(tmp_path, """
        import os, sys
        rank = int(os.environ["RANK"])
        if rank == 0:
            print('{"metric": "stub", "strong10k_status": "hung in timed exchange"}')
        sys.exit({0: 3, 1: 0, 2: 4}[rank])
    """)
    r = _run(["--gpus", "3"], child)
    assert r.returncode == 4
    assert json.loads(r.stdout)["strong10k_status"].startswith("hung")
    assert "rank 0 left with code 3" in r.stderr and "rank 2 left with code 4" in r.stderr


def test_launcher_ends_ranks_that_outlive_a_failed_one(tmp_path):
    child = _stub(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)          # a rank waiting for a peer that is gone
    """)
    r = _run(["--gpus", "2"], child, {"AMV_BENCH_LAUNCH_GRACE": "1"}, timeout=60)
    assert r.returncode == 7
    assert "killing pid" in r.stderr and r.stdout == ""


def test_more_gpus_than_devices_is_refused_not_measured_on_fewer():
    # this container has no GPU at all: --gpus 2 must not fall back to whatever is there
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and r.stdout == ""
    assert "2 devices needed" in r.stderr


def test_gpus_must_agree_with_world_size():
    r = _run(["--gpus", "8"], extra_env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and r.stdout == ""
    assert "--gpus 8 but WORLD_SIZE=2" in r.stderr


def test_config_head_fits_the_window_that_records_keep():
    sys.path.insert(0, ROOT)
    import bench
    # a single-GPU line has no strong leg: the other four BASELINE configs (rate, fraction, CPU path) + the headline's own
    # diagnostics must sit inside the first 24 scalars of `config`
    assert not [k for k in bench.CONFIG_HEAD if k.startswith("strong10k_") or k in ("rccl_ranks", "run_status")]
    first = bench.CONFIG_HEAD[:24]
    for p in ("c320_decode", "enc320", "stream10k", "coresident"):
        assert {p + "_fps", p + "_frac", p + "_cpu_fps"} <= set(first)
    assert {"adpcm_sps", "adpcm_frac", "adpcm_cpu_sps", "coresident_adpcm_sps", "mixed160_fps", "mixed160_handed_to_serial",
            "decode_traffic_ratio", "handed_to_serial", "workload"} <= set(first)
    cfg = bench.ordered_config({"zzz": 1, "secondary": {}, "enc320_fps": 2.0, "workload": "w"})
    assert list(cfg) == ["workload", "enc320_fps", "zzz", "secondary"]
    flat = bench.strong_flat(None, "hung in setup on rank 0")
    assert flat["strong10k_status"].startswith("hung") and flat["strong10k_ms"] is None and "rccl_ranks" in flat
    # the line of a run WITH ranks: the communicator's size, both verdicts, configs[3] as stated and the same sixteen
    # scalars -- all five BASELINE configs with their CPU path -- inside the first 24
    ranks = bench.CONFIG_HEAD_RANKS[:24]
    assert len(set(bench.CONFIG_HEAD_RANKS)) == len(bench.CONFIG_HEAD_RANKS) and set(bench.CONFIG_HEAD) <= set(bench.CONFIG_HEAD_RANKS)
    assert {"workload", "rccl_ranks", "run_status", "strong10k_status", "strong10k_ms", "strong10k_fps"} <= set(ranks)
    for p in ("c320_decode", "enc320", "stream10k", "coresident"):
        assert {p + "_fps", p + "_frac", p + "_cpu_fps"} <= set(ranks)
    assert {"adpcm_sps", "adpcm_frac", "adpcm_cpu_sps", "coresident_adpcm_sps"} <= set(ranks)
    cfg = bench.ordered_config({"zzz": 1, "parallelism": "x2", "enc320_fps": 2.0, "strong10k_status": "ok", "rccl_ranks": 2, "workload": "w"})
    assert list(cfg) == ["workload", "rccl_ranks", "strong10k_status", "enc320_fps", "parallelism", "zzz"]


def test_the_line_goes_out_once_whichever_thread_writes_it(tmp_path, monkeypatch):
    """emit_line: the main thread, a leg's watchdog and the peers' watcher may each come to write the line -- one line goes
    out; a watchdog that meets the main thread adding keys (json.dumps raises "dictionary changed size during iteration")
    tries again, and in the worst case the line's scalars leave with the reason in run_status"""
    sys.path.insert(0, ROOT)
    import bench
    out = tmp_path / "line.json"
    E = bench.Env()
    E.json_fd = os.open(str(out), os.O_WRONLY | os.O_CREAT)
    E.result = {"metric": "m", "value": 1.0, "config": {"zzz": 1, "workload": "w"}}
    real, calls = bench.json.dumps, []

    def flaky(obj, *a, **k):
        calls.append(1)
        if len(calls) <= 2:
            raise RuntimeError("dictionary changed size during iteration")
        return real(obj, *a, **k)

    monkeypatch.setattr(bench.json, "dumps", flaky)
    bench.write_line_now(E, "hung in leg adpcm on rank 0")
    monkeypatch.setattr(bench.json, "dumps", real)
    bench.write_line_now(E, "error: a second writer")
    bench.emit_line(E, E.result)
    lines = out.read_text().splitlines()
    assert len(lines) == 1 and len(calls) == 3
    line = json.loads(lines[0])
    assert line["config"]["run_status"].startswith("hung in leg adpcm") and list(line["config"])[0] == "workload"
    # ... and when it never settles: the scalars, with the reason
    E2 = bench.Env()
    out2 = tmp_path / "line2.json"
    E2.json_fd = os.open(str(out2), os.O_WRONLY | os.O_CREAT)
    E2.result = {"metric": "m", "value": 2.0, "config": {"workload": "w"}, "roofline": {"frac": 0.3}}
    state = {"n": 0}

    def never(obj, *a, **k):
        state["n"] += 1
        if state["n"] <= 8:
            raise RuntimeError("dictionary changed size during iteration")
        return real(obj, *a, **k)

    monkeypatch.setattr(bench.json, "dumps", never)
    monkeypatch.setattr(bench.time, "sleep", lambda s: None)
    bench.write_line_now(E2, "hung")
    line2 = json.loads(out2.read_text())
    assert line2["metric"] == "m" and line2["value"] == 2.0 and line2["config"]["run_status"].startswith("error: the line was being written")


def test_balanced_source_share_model():
    """the share of configs[3]'s stream the source keeps in the second strong leg: none without peers, nearly everything with
    one peer (whose one link carries all the rest), falling with the ranks, and never below an equal range"""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.balanced_src_share(1) is None
    shares = [bench.balanced_src_share(w) for w in (2, 3, 4, 8)]
    assert shares == sorted(shares, reverse=True) and 0.85 <= shares[0] <= 0.95 and 0.5 <= shares[-1] <= 0.65
    assert all(s > 1.0 / w for s, w in zip(shares, (2, 3, 4, 8)))
    flat = bench.share_flat(None, "hung in timed exchange on rank 1")
    assert flat["strong10k_srcshare_ms"] is None and flat["strong10k_srcshare_status"].startswith("hung")
    assert set(flat) <= set(bench.CONFIG_HEAD_RANKS)


# A rank of an N-rank run with everything that needs a GPU replaced: gloo instead of RCCL, the three workload functions
# replaced by stand-ins that go through the SAME reductions, barriers, CPU-leg gate and result builders (base_result,
# roofline, cpu_leg / cpu_leg_done, guarded_strong) as the real ones.  What runs for real: launch_ranks (the store it hosts,
# the environment), the ranks' side channel (Peers), run_workload / run_secondary / flat_secondary / ordered_config /
# write_line_and_leave -- the shape of the record an N-rank run leaves.
RANK_STUB = """
    import json, os, sys, time
    sys.path.insert(0, %(root)r)
    import torch, torch.distributed as dist
    import bench
    args = bench.parse_args(sys.argv[1:])
    E = bench.Env()
    E.world, E.rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    E.json_fd = os.dup(1)
    os.dup2(2, 1)
    E.dist = True
    dist.init_process_group("gloo", rank=E.rank, world_size=E.world)
    E.peers = bench.Peers(E, dist.distributed_c10d._get_default_store(), period=0.1)
    E.dev = torch.device("cpu")
    E.sh = bench.entry._load(bench.entry.PKG_NAME + ".sharding", os.path.join(bench.entry.PKG_DIR, "sharding.py"))
    torch.cuda.synchronize = lambda *a: None
    torch.cuda.empty_cache = lambda: None
    FAIL = os.environ.get("STUB_FAIL", "")          # "<rank>:<leg>:<raise|hang>"

    def leg(metric, unit, units, cfg):
        def run(E, args, **kw):
            name = E.strong_phase
            if FAIL and FAIL.split(":")[0] == str(E.rank) and FAIL.split(":")[1] in name:
                if FAIL.split(":")[2] == "hang":
                    time.sleep(600)
                raise RuntimeError("HIP error: stand-in fault")
            elapsed = E.sh.max_over_ranks(0.01 * (1 + E.rank), E.dev)
            r = bench.base_result(E, args, metric, unit, units, elapsed)
            r["config"] = dict({"workload": metric, "parallelism": "frame-range x%%d" %% E.world, "frames_per_gpu": units,
                                "mean_chunk_bytes": 3500.0, "handed_to_serial": 0}, **cfg)
            r["roofline"] = bench.roofline({"k": {"launches": 1, "ms_per_step": 1.0, "ms_per_launch": 1.0}}, 1e6, 0.01, None)
            if metric.startswith("160x120") and E.dist and getattr(args, "strong_leg", True):
                bench.guarded_strong(E, r, lambda: {"ms_per_step": 0.9, "frames_per_s": 1.1e7, "rccl_ranks": dist.get_world_size(),
                                                    "phase_ms_max_over_ranks": {"scatter": 0.1, "decode": 0.5, "gather": 0.3}})
                share = bench.balanced_src_share(E.world)
                bench.guarded_strong(E, r, lambda: {"ms_per_step": 0.6, "frames_per_s": 1.6e7, "src_share": share},
                                     flat=bench.share_flat, key="config4_strong_10k_src_share", what="strong-scaling leg, source share")
            if bench.cpu_leg(E, args):
                time.sleep(0.2)
                r["cpu_baseline"] = {"value": 5.0, "unit": unit, "cores": 1, "kind": "port", "sample": "stand-in"}
            bench.cpu_leg_done(E, args)
            return r
        return run

    def run_decode(E, args):
        if (args.width, args.frames, args.stream) == (None, None, "synthetic"):
            E.strong_phase = E.strong_phase or "headline"
            return leg("160x120 decode", "frames/s", 160000, {})(E, args)
        return leg("%%sx decode %%s %%s" %% (args.width, args.frames, args.stream), "frames/s", args.frames or 160000, {})(E, args)

    bench.run_decode = run_decode
    bench.run_encode = leg("encode", "frames/s", 8000, {"round_trip_psnr_db": 30.0})
    bench.run_adpcm = lambda E, args, with_video=False: leg("coresident" if with_video else "adpcm",
        "frames/s" if with_video else "samples/s", 64000, {"adpcm_samples_per_s": 1e9} if with_video else {})(E, args)
    result = bench.run_workload(E, args)
    bench.write_line_and_leave(E, result)
    E.peers.stop.set()
    dist.destroy_process_group()
"""


def test_two_rank_line_carries_every_config_and_the_cpu_path(tmp_path):
    """WORLD_SIZE = 2: the record an N-rank run leaves -- n_gpus, the weak value summed over the ranks, cpu_baseline (rank 0's
    CPU leg, the other rank waiting in the barrier), rccl_ranks / run_status / strong10k_status and all five BASELINE configs
    with their CPU path inside the first 24 scalars of `config`"""
    child = _stub(tmp_path, RANK_STUB % {"root": ROOT})
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], child)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["cpu_baseline"]["value"] == 5.0
    assert abs(line["value"] - 2 * 160000 * 2 / 0.02) < 1e-3             # both ranks' frames over the slower rank's time
    first = list(line["config"])[:24]
    assert first[:6] == ["workload", "rccl_ranks", "run_status", "strong10k_status", "strong10k_ms", "strong10k_fps"]
    assert line["config"]["rccl_ranks"] == 2 and line["config"]["run_status"] == "ok" and line["config"]["strong10k_status"] == "ok"
    for p in ("c320_decode", "enc320", "stream10k", "coresident"):
        assert {p + "_fps", p + "_frac", p + "_cpu_fps"} <= set(first), (p, first)
        assert line["config"][p + "_cpu_fps"] == 5.0
    assert {"adpcm_sps", "adpcm_frac", "adpcm_cpu_sps", "coresident_adpcm_sps"} <= set(first)
    assert line["config"]["c320_decode_fps"] == 2 * 128000 * 2 / 0.02     # summed over the ranks like the headline
    assert set(line["config"]["secondary"]) == {"decode_320x240", "decode_160x120_10k_stream", "decode_160x120_mixed",
                                                "decode_amv1_looped", "encode_320x240", "coresident_320x240_adpcm", "adpcm"}
    assert list(line["config"])[-1] == "secondary"
    # the second strong leg (the source keeps the balanced share): flat scalars behind the head, its dictionary beside the first's
    sys.path.insert(0, ROOT)
    import bench
    cfg = line["config"]
    assert cfg["strong10k_srcshare"] == bench.balanced_src_share(2) and cfg["strong10k_srcshare_status"] == "ok"
    assert cfg["strong10k_srcshare_ms"] == 0.6 and cfg["strong10k_ms"] == 0.9
    assert cfg["config4_strong_10k_src_share"]["src_share"] == cfg["strong10k_srcshare"]
    assert list(cfg).index("strong10k_scatter_ms") < list(cfg).index("strong10k_srcshare") < list(cfg).index("parallelism")


def test_a_rank_that_fails_in_a_leg_tells_its_peers_and_the_line_still_goes_out(tmp_path):
    """rank 1 raises inside the co-resident leg: it sets the flag in the store and leaves with EXIT_FAILED; rank 0 -- blocked
    in that leg's reduction -- learns it from the store (or from the backend), writes the line with what had finished and
    run_status = "error: ...", and leaves non-zero too.  Seconds, not the leg's watchdog."""
    child = _stub(tmp_path, RANK_STUB % {"root": ROOT})
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], child, {"STUB_FAIL": "1:coresident:raise"}, timeout=100)
    assert r.returncode == 4, r.stderr[-3000:]
    line = json.loads(r.stdout)
    assert line["config"]["run_status"].startswith("error:") and line["config"]["strong10k_status"] == "ok"
    assert "enc320_fps" in line["config"] and "coresident_fps" not in line["config"] and "adpcm_sps" not in line["config"]
    assert line["value"] > 0 and "cpu_baseline" in line


def test_a_rank_that_hangs_in_a_leg_ends_the_run_with_the_line_written(tmp_path):
    """rank 1 never comes back from the encode leg: every rank's watchdog fires after AMV_BENCH_LEG_SECONDS, rank 0 writes
    the line with run_status = "hung in leg encode_320x240 on rank 0" and both leave with EXIT_HUNG"""
    child = _stub(tmp_path, RANK_STUB % {"root": ROOT})
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], child, {"STUB_FAIL": "1:encode:hang", "AMV_BENCH_LEG_SECONDS": "3"},
             timeout=100)
    assert r.returncode == 3, r.stderr[-3000:]
    line = json.loads(r.stdout)
    assert line["config"]["run_status"].startswith("hung in leg encode_320x240 on rank 0")
    assert "stream10k_fps" in line["config"] and "enc320_fps" not in line["config"]


def test_launcher_limit_ends_ranks_that_hang_together_and_an_empty_line_is_not_clean(tmp_path):
    child = _stub(tmp_path, """
        import time
        time.sleep(600)          # every rank waiting: a rendezvous that never completes
    """)
    r = _run(["--gpus", "2"], child, {"AMV_BENCH_LAUNCH_LIMIT": "1"}, timeout=60)
    assert r.returncode == 3 and r.stdout == "" and "AMV_BENCH_LAUNCH_LIMIT" in r.stderr
    child = _stub(tmp_path, "pass")
    r = _run(["--gpus", "2"], child, timeout=60)
    assert r.returncode == 4 and r.stdout == "" and "wrote no line" in r.stderr


def _guard_script(tmp_path, body):
    path = tmp_path / "guard.py"
    path.write_text(textwrap.dedent("""
        import json, os, sys, time
        sys.path.insert(0, %r)
        import bench
        E = bench.Env()
        E.rank, E.world, E.json_fd = int(os.environ.get("RANK", "0")), 2, 1
        result = {"metric": "weak line", "value": 1.0, "config": {"zzz": 1, "workload": "w"}}
    """ % ROOT) + textwrap.dedent(body))
    return str(path)


def test_strong_leg_error_is_a_flat_status_and_a_nonzero_exit(tmp_path):
    """an exception of the exchange (how a device or RCCL fault surfaces): the weak line still goes out, with
    strong10k_status = "error: ..." as a flat scalar in front of it, and the process leaves with EXIT_FAILED -- not 0"""
    script = _guard_script(tmp_path, """
        def body():
            E.strong_phase = "timed exchange"
            raise RuntimeError("HIP error: an illegal memory access was encountered")
        bench.guarded_strong(E, result, body)
        bench.write_line_and_leave(E, result)
        print("not reached")
    """)
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=120)
    assert r.returncode == 4, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert "not reached" not in r.stdout
    assert line["metric"] == "weak line" and line["value"] == 1.0
    assert line["config"]["strong10k_status"].startswith("error: RuntimeError in timed exchange: HIP error")
    assert line["config"]["strong10k_ms"] is None and list(line["config"])[0] == "workload"
    assert list(line["config"]).index("strong10k_status") < list(line["config"]).index("zzz")


def test_strong_leg_hang_writes_the_line_and_leaves_with_its_own_code(tmp_path):
    """a send nobody receives: the watchdog writes the line (rank 0) with strong10k_status = "hung in <phase> on rank r" and
    leaves with EXIT_HUNG; a rank other than 0 leaves with the same code and writes nothing"""
    script = _guard_script(tmp_path, """
        def body():
            E.strong_phase = "first exchange (scatter / decode / gather)"
            time.sleep(600)
        bench.guarded_strong(E, result, body, seconds=1.0)
    """)
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["config"]["strong10k_status"].startswith("hung in first exchange (scatter / decode / gather) on rank 0")
    assert line["value"] == 1.0
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=120, env=dict(os.environ, RANK="1"))
    assert r.returncode == 3 and r.stdout.strip() == ""


def test_strong_leg_ok_path_keeps_going(tmp_path):
    script = _guard_script(tmp_path, """
        bench.guarded_strong(E, result, lambda: {"ms_per_step": 0.9, "frames_per_s": 1.1e7, "rccl_ranks": 2,
                                                   "phase_ms_max_over_ranks": {"scatter": 0.1, "decode": 0.5, "gather": 0.3}})
        bench.write_line_and_leave(E, result)
        print("reached")
    """)
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "reached"
    cfg = json.loads(r.stdout.strip().splitlines()[0])["config"]
    assert cfg["strong10k_status"] == "ok" and cfg["rccl_ranks"] == 2 and cfg["strong10k_gather_ms"] == 0.3
