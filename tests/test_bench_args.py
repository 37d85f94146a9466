"""bench.py's launch contract (no GPU needed): `python bench.py --gpus N` starts its N ranks itself through
torch.distributed.run on 127.0.0.1; under a launcher --gpus must equal WORLD_SIZE (a mismatch would print a one-GPU line
under an eight-GPU label)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_self_launch_command_line():
    cmd = bench.self_launch_command(4, {}, ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    assert os.path.isabs(cmd[cmd.index("--master-port") + 2]) and cmd[cmd.index("--master-port") + 2].endswith("bench.py")
    # the port can be pinned from outside
    cmd = bench.self_launch_command(2, {"MASTER_PORT": "29777"}, ["bench.py", "--gpus", "2"])
    assert cmd[cmd.index("--master-port") + 1] == "29777"


def test_no_self_launch_when_ranks_exist_or_single():
    assert bench.self_launch_command(1, {}, ["bench.py"]) is None
    assert bench.self_launch_command(8, {"WORLD_SIZE": "8", "RANK": "0"}, ["bench.py", "--gpus", "8"]) is None
    # the in-library sharded transform drives all GPUs from ONE process
    assert bench.self_launch_command(8, {}, ["bench.py", "--gpus", "8", "--workload", "sharded"]) is None
    assert bench.self_launch_command(8, {}, ["bench.py", "--gpus", "8", "--workload=sharded"]) is None


def test_gpus_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr


def test_free_port_and_default_gpus():
    """no MASTER_PORT: the port comes from the kernel (bind to port 0), not from a formula; `torchrun ... bench.py` without
    --gpus takes the launcher's WORLD_SIZE instead of aborting"""
    cmd = bench.self_launch_command(2, {}, ["bench.py", "--gpus", "2"])
    port = int(cmd[cmd.index("--master-port") + 1])
    assert 1024 < port < 65536
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'args.gpus = int(os.environ.get("WORLD_SIZE", "1"))' in src


def test_committed_counter_files_match_the_kernel_sources():
    """bench.py prints `roofline.traffic` / `valu` only from counter files taken on the kernel sources as they are now (their
    hash is stored beside the counters): an edit to one of those sources without a refresh on the GPU (tools/profile_refresh.sh)
    would silently turn the driver's line into `traffic: null`, so the CPU suite says it first.  Also: the regime the default
    line reports carries the two kernels of a 2^22 step with traffic between 2x and 3x the algorithmic bytes."""
    import json
    for key, wl in (("ntt22", None), ("ntt22_1stream", None), ("ntt22_mont", None), ("batch16", "batch16"), ("open22", "open22"), ("eval22", "eval22")):
        d, why = bench.load_if_current("profiles/latest_pmc_%s.json" % key, wl)
        assert d is not None, (key, why)
    with open(os.path.join(ROOT, "profiles", "latest_census.json")) as f:
        assert json.load(f).get("kernel_source_hash") == bench.kernel_source_hash()
    line = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_driver_args_c.json")))
    alg = line["roofline"]["algorithmic_bytes_per_step"]
    assert alg == 16 * (1 << 22) and 2.0 * alg < line["roofline"]["traffic"] < 3.0 * alg
    assert line["roofline"]["frac"] == line["roofline"]["achieved"] / line["roofline"]["peak"]


def test_every_cited_profile_file_exists():
    """DESIGN.md / README.md / HISTORY.md / profiles/README.md name their evidence as `rNN_*` / `latest_*` / `profiles/...`: every
    such name (brace lists and globs expanded) is a file under profiles/"""
    import glob
    import re
    missing = []
    for doc in ("DESIGN.md", "README.md", "HISTORY.md", "INTEGRATION.md", os.path.join("profiles", "README.md")):
        text = open(os.path.join(ROOT, doc)).read()
        for m in re.finditer(r"`([^`\s]+)`", text):
            t = m.group(1)
            if t.startswith("profiles/"):
                name = t
            elif re.match(r"^(r0\d_|latest_)[\w.\-{},*]+$", t):
                name = "profiles/" + t
            else:
                continue
            b = re.search(r"\{([^}]*)\}", name)
            for nm in ([name[:b.start()] + alt + name[b.end():] for alt in b.group(1).split(",")] if b else [name]):
                if not glob.glob(os.path.join(ROOT, nm)):
                    missing.append((doc, nm))
    assert not missing, missing
