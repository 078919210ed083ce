"""CPU tests of bench.py's N-rank launch path (`--dry-run`: gloo, a tiny model without HIP layers).

What the driver relies on: `python bench.py --gpus N ...` yields an N-rank run (bench.py starts the ranks itself when no
launcher did), `n_gpus` in the JSON line is the size of the process group, and a rank count that does not match --gpus is
an error, never a silently smaller run.  The dry run also checks the data-parallel bookkeeping with real collectives:
replicas stay identical after all-reduced updates and every rank puts its gradient all-reduces at the same places (the
cut points of GanTrainer.capture_segments)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_and_keeps_replicas_consistent():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1", "--training-ratio", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["steps"] == 2
    assert out["replicas_identical"] and out["finite"]
    # training_ratio critic all-reduces, then the generator's -- the same on both ranks
    assert out["allreduce_order"] == [["d", "d", "g"], ["d", "d", "g"]]
    # the self-diagnosis of an N > 1 line: group size as the backend sees it, every rank's own step time,
    # the collectives' share of a step, the agreed launch mode and why a graph mode fell back (None here: eager was asked for)
    mg = out["multi_gpu"]
    assert mg["ranks_seen_by_backend"] == 2
    assert 0 < mg["ms_per_step_rank_min"] <= mg["ms_per_step_rank_max"]
    assert mg["allreduce_calls_per_step"] == 3 and mg["allreduce_ms_per_step"] > 0 and mg["allreduce_bytes_per_step"] > 0
    assert out["config"]["launch"] == "eager" and out["config"]["launch_fallback"] is None
    # a secondary measurement that throws on rank 0 (here: the roofline leg, which needs a GPU) leaves an error
    # note in its place and the line is printed all the same -- the first 8-GPU run cannot come back empty because of it
    assert "error" in out["roofline"] and "WcHipError" in out["roofline"]["error"]


def test_single_rank_dry_run_needs_no_process_group():
    r = _run(["--dry-run", "--steps", "1", "--warmup", "0", "--training-ratio", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 1 and out["allreduce_order"] == [["d", "g"]]


def test_world_size_that_contradicts_gpus_is_refused():
    # a launcher started ONE rank but the command line says 8 GPUs: exit non-zero, print no result line
    r = _run(["--gpus", "8", "--dry-run", "--steps", "1"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "refusing" in r.stderr


def test_trace_boundaries_matches_segment_layout():
    """The cut points capture_segments() uses (one per critic update, then the generator's) on a CPU trainer."""
    import torch
    from wc_gan_amd.discriminator import make_discriminator
    from wc_gan_amd.generator import make_generator
    from wc_gan_amd.train import GanTrainer
    torch.manual_seed(0)
    G = make_generator(block_sizes=(8,), resamples=("UP",), first_block_shape=(4, 4, 8), block_norm='b',
                       block_after_norm='ucs', last_norm='b', last_after_norm='ucs')
    D = make_discriminator(input_image_shape=(8, 8, 3), block_sizes=(8, 8), resamples=('DOWN', 'SAME'), type=None,
                           spectral=False, sum_pool=True)
    with torch.no_grad():
        G(torch.zeros(2, 128), torch.zeros(2, 1, dtype=torch.int32))
    tr = GanTrainer(G, D, batch_size=2, training_ratio=3, flat_buckets=True)
    reals = [torch.rand(2, 8, 8, 3) * 2 - 1 for _ in range(3)]
    order, (d_loss, g_loss) = tr.trace_boundaries(reals)
    assert order == ['d', 'd', 'd', 'g']
    assert torch.isfinite(d_loss) and torch.isfinite(g_loss)
    assert tr._boundary is None
