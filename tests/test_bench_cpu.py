"""bench.py's N>1 control path on CPU (no GPU in this container): how ranks are started, what happens when the node
has fewer devices than `--gpus`, and that the product-side workload builders agree with the oracle's loader."""
import json
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT, read_case

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra, timeout=300):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_plain_invocation_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: two rank processes rendezvous over gloo, rank 0 prints one line
    that says n_gpus == 2 and that two ranks took part (EZPZ_BENCH_DRY: the control path with an empty step)."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"EZPZ_BENCH_DRY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["world_size_seen"] == 2 and line["steps"] == 3 and line["dry_run"] is True
    # max over ranks: rank 1 sleeps 2 ms per step, rank 0 1 ms
    assert line["ms_per_step"] >= 1.9


def test_launcher_invocation_world_2():
    """The driver's form: torch.distributed.run starts the ranks; the script must not spawn again."""
    env = dict(os.environ, EZPZ_BENCH_DRY="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    import socket

    with socket.socket() as s:  # a port that is free now (two test sessions on one host used to collide on a fixed one)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["world_size_seen"] == 2


def test_refuses_more_gpus_than_devices():
    """No GPU here: `--gpus 8` must exit non-zero and print no result line (never a 1-GPU number labelled as 8)."""
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "--gpus 8" in r.stderr


def test_gpus_must_match_world_size():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0",
                                                               "EZPZ_BENCH_DRY": "1"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_workload_builders_match_the_oracle_loader():
    """bench.py builds its workloads with the product's front end; same records / guesses as the oracle's loader."""
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle as O
    from oracle import textual as T

    for name, text in (("massive50", T.gen_big_problem(50)), ("massive20o", T.gen_big_problem(20, True))):
        desc, recs, guesses, jitter, expect = bench.make_workload(name)
        ref = T.load(text)
        assert np.array_equal(guesses, ref.guesses)
        assert recs.tobytes() == O.stack(ref.constraints).tobytes(), name
    for name in ("circle_tangent", "square", "arc_radius"):
        desc, recs, guesses, jitter, expect = bench.make_workload(name)
        ref = T.load(read_case(name))
        want = O.stack([O.set_from_initial_values(c, ref.guesses) for c in ref.constraints])
        assert recs.tobytes() == want.tobytes(), name
