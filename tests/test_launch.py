"""`python bench.py --gpus N` by itself starts its own ranks (raytracer_amd/launch.py); on CPU with a stub worker."""
import json
import os
import subprocess
import sys
import textwrap

from raytracer_amd.launch import launch_ranks, rank_env

STUB = textwrap.dedent("""
    import json, os, sys
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
    assert int(os.environ["MASTER_PORT"]) > 0 and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    fail = int(sys.argv[1])
    if rank == fail:
        sys.exit(3)
    if fail >= 0 and rank != fail:
        import time
        time.sleep(30)  # a rank stuck in a collective: the launcher must end it
    print(json.dumps({"rank": rank, "world": world}))  # every rank prints; only rank 0's line may come through
""")


def test_ranks_get_their_environment_and_rank0_speaks(tmp_path, capfd):
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    assert launch_ranks(3, [sys.executable, str(stub), "-1"], timeout=60) == 0
    out = capfd.readouterr().out.strip().splitlines()
    assert [json.loads(ln) for ln in out] == [{"rank": 0, "world": 3}]


def test_a_failing_rank_ends_the_job(tmp_path):
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    import time
    t0 = time.time()
    assert launch_ranks(3, [sys.executable, str(stub), "1"], timeout=60) == 3
    assert time.time() - t0 < 20  # the sleeping ranks were terminated, not waited for


def test_rank_env_is_complete():
    env = rank_env(2, 4, 12345, base={})
    assert env == {"RANK": "2", "LOCAL_RANK": "2", "WORLD_SIZE": "4", "LOCAL_WORLD_SIZE": "4",
                   "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "12345", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}


def test_bench_starts_its_own_ranks(root):
    """No WORLD_SIZE in the environment and --gpus 2: bench.py becomes the launcher.  Without a GPU both ranks stop at
    the product's "needs a GPU" check, which shows they were started with a world of 2 (a world mismatch exits earlier)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "WORLD_SIZE" not in r.stderr and r.stderr.count("needs a GPU") >= 1
