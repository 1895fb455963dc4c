"""The multi-GPU path is region sharding with no data-path collective; ranks meet only for the barrier, the max-over-ranks
clock and the host-side gather of their outputs.  Rehearsed here on CPU with gloo, world_size 2, on real data: both ranks plan the
shards of a BAM's tile list, each runs the chain on its own tiles (on the oracle -- no GPU here), rank 0 joins the outputs in shard order
and finds exactly the single-process output (tests/shard_worker.py)."""
import json
import os
import subprocess
import sys

import numpy as np

import bamwriter
from uvc_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_shard_a_bam_and_rank0_joins_the_outputs(tmp_path):
    reads = synth.generate_region(seed=77, region_len=60000, depth=20, beg=35000, snv_every=1500, somatic_every=4000, indel_every=2500)
    recs = bamwriter.records_from_reads(reads, tid=0)
    chrom_len = reads["end"] + 20000
    rng = np.random.default_rng(5)
    seq = "".join("ACGT"[i] for i in rng.integers(0, 4, chrom_len))
    seq = seq[:reads["beg"]] + reads["refseq"] + seq[reads["end"]:]
    bam, fa = str(tmp_path / "s.bam"), str(tmp_path / "s.fa")
    bamwriter.write_bam(bam, [("chrS", chrom_len)], recs)
    bamwriter.write_fasta(fa, [("chrS", seq)])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "tests", "shard_worker.py"), bam, fa, "chrS", "8192"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # only rank 0 prints
    j = json.loads(lines[0])
    assert j["world"] == 2 and j["tiles"] == 15 and sum(j["tiles_per_rank"]) == 15
    assert j["equal_to_serial"] and j["lines"] >= 10
    assert min(j["called_per_rank"]) >= 2        # both ranks had tiles with reads: the cut falls inside the covered stretch
    assert max(j["cost_per_rank"]) <= 0.75 * sum(j["cost_per_rank"])


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher spawns two ranks itself (children first, nothing in the parent touches a GPU) and
    rank 0 reports n_gpus = 2; --dry-run keeps it on the CPU (gloo barrier + max-over-ranks clock, no kernels)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0", "--tile-kb", "10", "--dry-run"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # only rank 0 prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == 3
    assert j["ms_per_step"] >= 19.0              # max over ranks: rank 1 sleeps 20 ms per step, rank 0 only 10 ms
    assert abs(j["value"] - 2 * 10000 * 3 / (j["ms_per_step"] * 3 / 1e3)) / j["value"] < 1e-6


def test_bench_strong_scaling_cut_over_three_ranks():
    """`--total-tiles M` (config 3's shape: one job, N shards): three ranks cut seven tiles into contiguous shards that cover the list once,
    `steps` of a rank = its shard, the line says `scaling: strong` and prices the whole job against the slowest rank; fewer tiles than ranks
    is refused before anything is timed.  --dry-run: the cut and the clock protocol on the CPU, no kernels."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--total-tiles", "7", "--warmup", "0", "--tile-kb", "10", "--dry-run"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    job = j["config"]["strong_scaling_job"]
    assert j["n_gpus"] == 3 and j["scaling"] == "strong"
    assert job["total_tiles"] == 7 and job["tiles_owned_sum"] == 7 and job["tiles_of_a_rank_min"] >= 2 and job["tiles_of_a_rank_max"] <= 3
    assert abs(j["value"] - 7 * 10000 / (j["ms_per_step"] * j["steps"] / 1e3)) / j["value"] < 1e-6      # whole job / slowest rank (rank 0 prints its own `steps`)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--total-tiles", "2", "--dry-run"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "fewer tiles than ranks" in (bad.stderr + bad.stdout)
