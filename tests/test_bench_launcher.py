"""CPU: ``python bench.py --gpus N`` must end as N ranks or as an error, never as a silent single-GPU number
(VERDICT r02 weak #9).  No GPU is touched: the launcher decides before anything initialises the device."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH, *args], env=e, capture_output=True, text=True, timeout=300)


def test_gpus_n_without_torchrun_env_starts_the_one_rank_per_gpu_launcher():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], SPV_BENCH_LAUNCH_DRYRUN="1")
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", "2", "--steps", "3", "--warmup", "1"]   # the ranks get the caller's own arguments


def test_gpus_n_on_a_node_with_fewer_gpus_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("node has the GPUs")
    r = _run(["--gpus", "2"])
    assert r.returncode != 0
    # either the launcher's own sysfs count says so, or (no KFD topology in sysfs: nothing to count without the runtime) the ranks do
    assert ("exposes" in r.stderr or "needs an MI355X" in r.stderr) and not r.stdout.strip()   # no JSON line


def test_world_size_that_differs_from_gpus_is_an_error():
    for world, gpus in (("1", "2"), ("4", "8"), ("2", "1")):
        r = _run(["--gpus", gpus], WORLD_SIZE=world, RANK="0", LOCAL_RANK="0")
        assert r.returncode != 0 and "WORLD_SIZE" in r.stderr, (world, gpus, r.stderr)
        assert not r.stdout.strip()


def test_default_preset_follows_the_rank_count():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.parse([]).config == "c2" and bench.parse(["--gpus", "1"]).config == "c2"
    a = bench.parse(["--gpus", "8"])
    assert a.config == "c3" and a.genes == 20_000 and a.batch_size == 4096 and a.cells == 25_000
    assert bench.parse(["--gpus", "8", "--config", "c2"]).config == "c2"


def test_bench_line_carries_every_key_of_the_contract():
    """the JSON line is assembled from one dict literal in bench.main: a key lost to an editing accident (a trailing
    comment swallowed `"data"`) must fail here, not on the GPU box"""
    import ast
    tree = ast.parse(open(BENCH).read())
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    lits = [n.value for n in ast.walk(main) if isinstance(n, ast.Assign) and isinstance(n.value, ast.Dict)
            and any(isinstance(t, ast.Name) and t.id == "out" for t in n.targets)]
    assert len(lits) == 1
    keys = {k.value for k in lits[0].keys if isinstance(k, ast.Constant)}
    assert {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "elbo_delta"} <= keys
    cfg = next(v for k, v in zip(lits[0].keys, lits[0].values) if isinstance(k, ast.Constant) and k.value == "config")
    assert {"workload", "parallelism"} <= {k.value for k in cfg.keys if isinstance(k, ast.Constant)}
    src = open(BENCH).read()
    assert 'out["cpu_baseline"]' in src and '"cores"' in src and '"cpu_model"' in src


def test_host_cpu_probe_reports_usable_threads_and_the_model_string():
    sys.path.insert(0, ROOT)
    import bench
    usable, logical, model = bench.host_cpus()
    assert 1 <= usable <= logical and isinstance(model, str) and model
