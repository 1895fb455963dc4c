#!/usr/bin/env python3
"""bench.py -- pileup positions scored per second at 300x depth (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

A step = one pass of the whole hot path (accumulate P1..P5b + default-gate scoring, scored records
copied back to the host) over one synthetic chr20-shaped 300x non-UMI tile that is already resident
in HBM.  Every rank owns one GPU and its own tile (different seed): regions shard with no data-path
collective, so scaling is weak.  value = ranks * tile positions * K / max-over-ranks wall time.

Extra objects on the JSON line:
  roofline      dominant kernel of the timed steps: algorithmic bytes per launch
                / its HIP-event duration on the library's own stream, vs the 8 TB/s HBM peak.
  cpu_baseline  the CPU oracle (a scalar port of the reference algorithm) on a bounded sample of the
                same workload shape on the host cores (rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

READ_LEN = 150
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_position(kernel, depth):
    """Per-position algorithmic HBM bytes of each kernel.  Built from the SURVEY 8(d)
    components: 1 B base + 1 B qual per read-base, one per-read record, 5 544 B of accumulator records."""
    d = float(depth)
    reads = 2.0 * d + 48.0 * d / READ_LEN           # bases + quals + per-read record
    table = {
        "k_prep_fast": reads + 1 + 8 + 208,                         # + ref, baq ; writes SegFormatPrepSet
        "k_thres": 208 + 28 + 72 + 4,                                # prep + rtr -> thres + indelphred
        "k_p2_fast_link": 48.0 * d / READ_LEN + 72 + 16 + 8 + (152 + 16 + 4),        # read records, thres, 2 baq, 2 indelphred ; writes LINK_M seg info + a1/a2 BQ + bqsum
        "k_p2_fast_base": reads + 1 + 72 + 16 + 13 * (152 + 16 + 4),                 # + ref, thres, 2 baq ; writes the base symbols' seg info
        "k_frag": reads + 48.0 * d / (2 * READ_LEN) + 1 + 8 + 14 * 20 + 14 * 4 * (6 + 6 + 4),   # + fragment records, avgBQ inputs ; writes frag, fam(3), VQ(4)
        "k_p5b": 2 * 14 * 4 + 14 * 4 * 6,
    }
    return table.get(kernel, reads)


# bench kernel label -> name(s) in the rocprofv3 output (template instantiations; the default workload runs the PLAIN ones)
PROFILE_NAMES = {"k_p2_fast_link": ("k_p2_fast<true, false, true>", "k_p2_fast<true, false, false>"), "k_p2_fast_base": ("k_p2_fast<false, true, true>", "k_p2_fast<false, true, false>"),
                 "k_p2_slow_walk": ("k_p2_slow<true>",), "k_p2_slow_table": ("k_p2_slow<false>",), "k_frag": ("k_frag16<true>", "k_frag16<false>", "k_frag<true>", "k_frag<false>")}


def measured_traffic(kernel, tile_kb, depth):
    """HBM-side bytes per launch of `kernel` from the committed PMC passes (profiles/traffic_latest.json, made by
    scripts/gpu_round_profile.sh: FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes), or None when the passes were taken on
    another workload."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if not os.path.exists(path):
        return None
    try:
        t = json.load(open(path))
        if t.get("_workload") != {"tile_kb": tile_kb, "depth": depth}:
            return None
        for name in PROFILE_NAMES.get(kernel, (kernel,)):
            if name in t:
                return t[name]["traffic_bytes_per_launch"]
        return None
    except (KeyError, ValueError):
        return None


def run_cpu_baseline(depth, n_regions=16, region_len=20000):
    """Times the oracle (test infrastructure) on `n_regions` independent regions, one thread each."""
    from concurrent.futures import ThreadPoolExecutor
    from uvc_amd import _ffi, region, synth
    lib = _ffi.Lib(_ffi.oracle_library_path(), "uvc_oracle_")
    params = region.default_params(lib)
    cores = min(16, os.cpu_count() or 1)
    regs = []
    for i in range(n_regions):
        r = synth.generate_region(seed=777 + i, region_len=region_len, depth=depth)
        R = region.Region(lib, params, r["tid"], r["beg"], r["end"], r["refseq"])
        R.set_reads(r)
        regs.append(R)

    def work(R):
        R.accumulate()
        return len(R.score()["refpos"])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(work, regs))
    dt = time.perf_counter() - t0
    return {"value": n_regions * region_len / dt, "unit": "positions/s", "cores": cores, "kind": "port",
            "sample": "%d regions x %d bp at %dx, oracle (scalar C++ port), one region per thread, %.1f s wall" % (n_regions, region_len, depth, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tile-kb", type=int, default=1000, help="tile length in kb (chr20 is processed as 1 Mb tiles)")
    ap.add_argument("--depth", type=int, default=300)
    ap.add_argument("--streams", type=int, default=1, help="split the tile into this many regions, each on its own HIP stream, accumulated concurrently")
    ap.add_argument("--umi", action="store_true", help="duplex-UMI families (BASELINE config 4 shape when combined with --depth 2000 --tile-kb 200)")
    ap.add_argument("--serial", action="store_true", help="(the default) one resident tile, accumulate then score, strictly one after the other")
    ap.add_argument("--pipeline", action="store_true", help="time the streamed mode instead: two resident tiles, the accumulate of tile k+1 is enqueued before the (synchronous) score of tile k; every step still does one full accumulate + score of a whole tile.  bench.py --streamed reports this mode beside the timed one")
    ap.add_argument("--streamed", action="store_true", help="behind the timed region, also measure the streamed mode (two handles) and report it as \"streamed\"; off by default so that a profiler run of the default command sees the timed launches only")
    ap.add_argument("--no-streamed", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--all-out", action="store_true", help="second series of SURVEY 8(d): score every symbol of every position (-A), not only the default-gate candidates")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="CPU-only rehearsal of the multi-rank protocol (no kernels, used by the gloo tests)")
    args = ap.parse_args()

    from uvc_amd import shard
    rank, local_rank, world = shard.dist_env()
    if world != args.gpus and world > 1:
        args.gpus = world
    region_len = args.tile_kb * 1000

    if args.dry_run:
        clock = shard.Clock(backend="gloo")
        clock.barrier(); t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(0.01 * (1 + rank))
        clock.barrier(); dt = clock.max_over_ranks(time.perf_counter() - t0)
        if rank == 0:
            print(json.dumps({"metric": "pileup positions scored/sec at 300x depth", "value": world * region_len * args.steps / dt, "unit": "positions/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic", "config": {"workload": "dry-run"}, "dry_run": True}))
        clock.close()
        return

    import torch
    torch.cuda.set_device(local_rank)
    clock = shard.Clock(backend="nccl")
    from uvc_amd import region, synth
    lib = region.gpu_lib()
    rc = lib.dll.uvcgpu_init(local_rank)
    if rc != 0:
        raise RuntimeError(lib.last_error())
    lib.dll.uvcgpu_region_set_profiling.argtypes = [C.c_void_p, C.c_int]
    lib.dll.uvcgpu_region_kernel_times.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_float), C.c_int]
    params = region.default_params(lib)
    t_gen = time.perf_counter()
    sub_len = region_len // args.streams
    args.pipeline = bool(args.pipeline) and not args.serial and args.streams == 1
    if args.pipeline:
        assert args.streams == 1, "--pipeline uses its own two handles"
    n_tiles = 2 if args.pipeline else args.streams
    tiles = [synth.generate_region(seed=12345 + rank + 1000 * i, region_len=sub_len, depth=args.depth, beg=1000000 + i * (sub_len + 1000), umi=args.umi) for i in range(n_tiles)]
    reads = tiles[0]
    t_gen = time.perf_counter() - t_gen
    Rs = [region.Region(lib, params, t["tid"], t["beg"], t["end"], t["refseq"]) for t in tiles]
    R = Rs[0]
    t_h2d = time.perf_counter()
    for Ri, t in zip(Rs, tiles):
        Ri.set_reads(t)
    t_h2d = time.perf_counter() - t_h2d
    n_reads_total = sum(int(t["n_reads"]) for t in tiles)
    n_read_bases = n_reads_total * READ_LEN

    def step():
        for Ri in Rs:          # enqueue only: each region has its own stream
            Ri.accumulate()
        recs = [Ri.score(all_out=args.all_out, capacity=(15 * (sub_len + 2) if args.all_out else max(65536, sub_len // 4)), copy=False, release_state=True) for Ri in Rs]
        return {"refpos": np.concatenate([r["refpos"] for r in recs])} if len(recs) > 1 else recs[0]

    def score_one(Ri):
        return Ri.score(all_out=args.all_out, capacity=(15 * (sub_len + 2) if args.all_out else max(65536, sub_len // 4)), copy=False, release_state=True)

    lib.dll.uvcgpu_region_set_profiling(R.h, 1)          # HIP events around the kernels of handle 0 (the roofline leg reads them)
    if args.pipeline:
        n_reads_total //= 2; n_read_bases //= 2          # per step: one tile
        k_state = [0]
        Rs[0].accumulate()                                # prologue: the pipeline is primed outside the timed region ...

        def step():                                       # ... and every step enqueues the next tile's accumulate, then scores the current one
            k = k_state[0]; k_state[0] += 1
            Rs[(k + 1) % 2].accumulate()
            return score_one(Rs[k % 2])

    for _ in range(args.warmup):
        step()
    ktimes = {}
    names_buf = C.create_string_buffer(1024)
    ms_buf = (C.c_float * 32)()
    n_rec = 0
    torch.cuda.synchronize(); clock.barrier()
    t0 = time.perf_counter()
    for it in range(args.steps):
        rec = step()
        n_rec = len(rec["refpos"])
        if args.pipeline and (args.warmup + it) % 2 == 1:
            continue   # handle 0 has its next accumulate in flight: asking for its kernel times would wait for it and stall the pipeline
        n = lib.dll.uvcgpu_region_kernel_times(R.h, names_buf, 1024, ms_buf, 32)   # the handle's stream is idle here: its score() was synchronous
        for nm, ms in zip(names_buf.value.decode().split(";")[:n], list(ms_buf)[:n]):
            ktimes.setdefault(nm, []).append(ms)
    torch.cuda.synchronize(); clock.barrier()
    dt = clock.max_over_ranks(time.perf_counter() - t0)
    total_positions = clock.sum_over_ranks(float(region_len)) * args.steps

    # outside the timed region: the same step on one handle, strictly serial, so that the kernel durations are also known without
    # the other tile's scoring kernels running beside them (reported as "serial", never as `value`)
    serial = None
    if args.pipeline:
        torch.cuda.synchronize()
        score_one(Rs[k_state[0] % 2])                      # drain the accumulate the last step left in flight
        sk = {}
        ts = time.perf_counter()
        for _ in range(2):
            R.accumulate(); score_one(R)
            n = lib.dll.uvcgpu_region_kernel_times(R.h, names_buf, 1024, ms_buf, 32)
            for nm, ms in zip(names_buf.value.decode().split(";")[:n], list(ms_buf)[:n]):
                sk.setdefault(nm, []).append(ms)
        serial = {"ms_per_step": 1e3 * (time.perf_counter() - ts) / 2, "kernel_ms": {k: sum(v) / len(v) for k, v in sk.items()}}

    # outside the timed region as well: what a caller gets that streams tiles through two handles (the same tile in both here)
    streamed = None
    if args.streamed and (not args.pipeline) and args.streams == 1 and not args.all_out:
        R2 = region.Region(lib, params, reads["tid"], reads["beg"], reads["end"], reads["refseq"])
        R2.set_reads(reads)
        pair = [R, R2]
        pair[0].accumulate()
        def sstep(k):
            pair[(k + 1) % 2].accumulate()
            return score_one(pair[k % 2])
        sstep(0)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for k in range(1, 1 + args.steps):
            sstep(k)
        torch.cuda.synchronize()
        sdt = clock.max_over_ranks(time.perf_counter() - ts)
        streamed = {"ms_per_step": 1e3 * sdt / args.steps, "value": clock.sum_over_ranks(float(region_len)) * args.steps / sdt}
        score_one(pair[(1 + args.steps) % 2])
        R2.close()

    if rank == 0:
        avg = {k: sum(v) / len(v) for k, v in ktimes.items()}
        dom = max(avg, key=avg.get)
        abytes = algorithmic_bytes_per_position(dom, args.depth) * R.npos   # per launch: one region
        achieved = abytes / (avg[dom] * 1e-3) / 1e9
        out = {
            "metric": "pileup positions scored/sec at 300x depth", "value": total_positions / dt, "unit": "positions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "chr20-shaped tumor-only non-UMI tile: %d kb at %dx, 150 bp paired-end, resident in HBM; step = accumulate P1..P5b + default-gate scoring / calling + D2H of the records of one tile (planes released with the score call: they are zeroed for the next tile under the D2H)%s" % (args.tile_kb, args.depth, "; tiles stream through two handles (the accumulate of tile k+1 is enqueued before the synchronous score of tile k)" if args.pipeline else ""),
                       "tile_positions": region_len, "streams": args.streams, "pipeline": bool(args.pipeline), "all_out": bool(args.all_out), "umi": bool(args.umi), "reads_per_tile": n_reads_total, "read_bases_per_tile": n_read_bases, "scored_records_per_tile": n_rec,
                       "parallelism": "region-shard x%d (no collective on the data path)" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(dom, args.tile_kb, args.depth),
                         "algorithmic_bytes_per_launch": abytes, "kernel_ms": avg[dom]},
            "kernel_ms": {k: round(v, 4) for k, v in sorted(avg.items(), key=lambda kv: -kv[1])},
            "read_bases_per_s": n_read_bases * world * args.steps / dt,
            "host_prep_s": {"generate": round(t_gen, 2), "pack_and_h2d": round(t_h2d, 3)},
        }
        if serial:
            sd = serial["kernel_ms"].get(dom)
            out["serial"] = {"ms_per_step": round(serial["ms_per_step"], 3), "kernel": dom, "kernel_ms": round(sd, 4) if sd else None,
                             "roofline_frac": (abytes / (sd * 1e-3) / 1e9 / HBM_PEAK_GBS) if sd else None,
                             "note": "one handle, accumulate then score with nothing overlapped; measured after the timed region"}
        if streamed:
            out["streamed"] = {"ms_per_step": round(streamed["ms_per_step"], 3), "value": streamed["value"], "unit": "positions/s",
                               "note": "two handles, the accumulate of tile k+1 enqueued before the synchronous score of tile k (bench.py --pipeline times this mode); measured after the timed region"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = run_cpu_baseline(args.depth)
        print(json.dumps(out))
    for Ri in Rs:
        Ri.close()
    clock.close()


if __name__ == "__main__":
    main()
