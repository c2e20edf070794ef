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
    single = [k for k in bench.CONFIG_HEAD if not (k.startswith("strong10k_") or k == "rccl_ranks")]
    first = single[:24]
    for p in ("c320_decode", "enc320", "stream10k", "coresident"):
        assert {p + "_fps", p + "_frac", p + "_cpu_fps"} <= set(first)
    assert {"adpcm_sps", "adpcm_frac", "adpcm_cpu_sps", "coresident_adpcm_sps", "mixed160_fps", "mixed160_handed_to_serial",
            "decode_traffic_ratio", "handed_to_serial", "workload"} <= set(first)
    cfg = bench.ordered_config({"zzz": 1, "secondary": {}, "enc320_fps": 2.0, "workload": "w"})
    assert list(cfg) == ["workload", "enc320_fps", "zzz", "secondary"]
    flat = bench.strong_flat(None, "hung in setup on rank 0")
    assert flat["strong10k_status"].startswith("hung") and flat["strong10k_ms"] is None and "rccl_ranks" in flat


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
