"""One rank of the region-shard rehearsal (started by tests/test_shard_gloo.py through torch.distributed.run, backend gloo).

Every rank plans the same contiguous shards over the tile list of a BAM (uvc_amd.shard: BAI-byte cost + length, uvcio_plan_shards), runs
the chain of uvc_amd/pipeline.py on ITS tiles -- on the CPU oracle, there is no GPU here -- and hands its lines to rank 0, which joins
them in shard order and compares them with its own single-process run.  No data-path collective: ranks meet for the barrier, the clock
and the final gather of the text."""
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from uvc_amd import _ffi, io as uio, pipeline, shard   # noqa: E402


def lines_of(lib, bam, fa, chrom, tile, only=None):
    out = io.StringIO()
    n = 0
    for res in pipeline.call_contig(lib, bam, fa, chrom, tile=tile, only=only):
        pipeline.write_tsv(res, out, header=False); n += 1
    return out.getvalue(), n


def main():
    bam_path, fa_path, chrom, tile = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    clock = shard.Clock(backend="gloo")
    rank, world = clock.rank, clock.world
    lib = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_")
    bam, fa = uio.Bam(bam_path), uio.Fasta(fa_path)
    tid = bam.tid(chrom)
    tiles = pipeline.contig_tiles(0, bam.refs[tid][1], tile)
    costs = shard.tile_costs(bam, [(tid, t["beg"], t["end"]) for t in tiles])
    plan = shard.plan_contiguous(costs, world)
    mine = [i for i in range(len(tiles)) if plan[i] == rank]
    clock.barrier(); t0 = time.perf_counter()
    text, n_called = lines_of(lib, bam, fa, chrom, tile, only=mine)
    clock.barrier(); dt = clock.max_over_ranks(time.perf_counter() - t0)
    parts = clock.gather_objects((rank, text, len(mine), n_called))
    if rank == 0:
        parts.sort()
        joined = "".join(p[1] for p in parts)
        serial, _ = lines_of(lib, bam, fa, chrom, tile)
        print(json.dumps({"world": world, "tiles": len(tiles), "tiles_per_rank": [p[2] for p in parts], "called_per_rank": [p[3] for p in parts],
                          "lines": joined.count("\n"), "equal_to_serial": joined == serial, "seconds": dt, "cost_per_rank": [int(sum(c for c, s in zip(costs, plan) if s == r)) for r in range(world)]}))
    clock.close()


if __name__ == "__main__":
    main()
