#!/usr/bin/env python3
"""bench.py -- pileup positions scored per second at 300x depth (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

A step = ONE TILE through the whole per-tile hot path, on a stream of DISTINCT synthetic chr20-shaped 300x non-UMI tiles (1 Mb each,
`--tiles` of them, visited round-robin) whose raw UvcReadSoA columns are resident in HBM when the timed region starts:

    uvcgpu_region_reset              the region side arrays of the tile's own reference on the device (uvc_rtr.hip)
    uvcgpu_region_set_reads_device   the device half of set_reads: CIGAR facts + family / fragment nesting (uvc_prep.hip), the three
                                     radix orders, k_pack_bq, k_aln_prelude (updateByAln's per-read prelude), k_build_p2list
    uvcgpu_region_accumulate         P1 .. P5b
    uvcgpu_region_score              default-gate scoring + calling, D2H of the records

Tiles are software-pipelined over their handles (the next tile's preparation and accumulate are enqueued before the synchronous score of
the current one; UVC_BENCH_VALUE_THREADS=n deals the tiles to n host threads instead), which is how a caller streams chr20 through the
library.  Every rank owns one GPU and its own tiles (different seeds):
regions shard with no data-path collective, scaling is weak.  value = ranks * tile positions * K / max-over-ranks wall time.
`--gpus N` without a launcher starts its own N ranks (children are spawned before anything touches a GPU).

Extra objects on the JSON line:
  roofline        dominant kernel of the timed steps: algorithmic bytes per launch / its HIP-event duration on the
                  library's own stream, vs the 8 TB/s HBM peak; `traffic` is REPLAYED from the committed PMC passes (profiles/).
  cpu_baseline    the CPU oracle (a scalar port of the reference algorithm) on a bounded sample of the same workload shape on the host
                  cores (rank 0, N = 1 only).
  pcie_inclusive  the same stream with the columns starting in (pinned) host memory: uvcgpu_region_set_reads copies them (650 MB per tile)
                  before the device half runs.  Reported beside `value`, never as `value` (inputs resident in HBM is the metric's contract).
  resident        one prepared tile accumulated + scored again and again with nothing overlapped (the round-1 figure): clean per-kernel
                  durations without another tile's kernels beside them.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# A region handle has three streams and several handles are in flight: with the HIP runtime's default of four hardware queues the streams of
# different handles share a queue and a kernel waits behind another handle's 10 ms copy (pcie_inclusive: 16.0 ms per tile with 4 queues,
# 13.5 with 16; 24 was slower again).  Read by the runtime when it starts, so it is set before anything imports it; an explicit setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

READ_LEN = 150
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
METRIC = "pileup positions scored/sec at 300x depth"


def algorithmic_bytes_per_position(kernel, depth):
    """Per-position algorithmic HBM bytes of each kernel.  Built from the SURVEY 8(d)
    components: 1 B base + 1 B qual per read-base, one per-read record, 5 544 B of accumulator records."""
    d = float(depth)
    reads = 2.0 * d + 48.0 * d / READ_LEN           # bases + quals + per-read record
    table = {
        "k_prep_fast": reads + 1 + 8 + 208,                         # + ref, baq ; writes SegFormatPrepSet
        "k_thres": 208 + 28 + 72 + 4,                                # prep + rtr -> thres + indelphred
        "k_p2_fast_link": 48.0 * d / READ_LEN + 72 + 16 + 8 + (152 + 16 + 4),        # read records, thres, 2 baq, 2 indelphred ; writes LINK_M seg info + a1/a2 BQ + bqsum
        "k_p2_fast_base": reads + 1 + 72 + 16 + (152 + 16 + 4),                      # + ref, thres, 2 baq ; writes the reference base's seg info (the other symbols' cells: k_p2_mism)
        "k_frag": reads + 48.0 * d / (2 * READ_LEN) + 1 + 8 + 14 * 20 + 14 * 4 * (6 + 6 + 4),   # + fragment records, avgBQ inputs ; writes frag, fam(3), VQ(4)
        "k_p5b": 2 * 14 * 4 + 14 * 4 * 6,
    }
    return table.get(kernel, reads)


# bench kernel label -> name(s) in the rocprofv3 output (template instantiations; the default workload runs the PLAIN ones)
PROFILE_NAMES = {"k_p2_fast_link": ("k_p2_fast<true, false, true>", "k_p2_fast<true, false, false>"), "k_p2_fast_base": ("k_p2_fast<false, true, true>", "k_p2_fast<false, true, false>"),
                 "k_prep_fast": ("k_prep_fast<false>", "k_prep_fast<true>"), "k_p2_slow_walk": ("k_p2_slow<true>",), "k_p2_slow_table": ("k_p2_slow<false>",), "k_frag": ("k_frag16<true>", "k_frag16<false>", "k_frag<true>", "k_frag<false>")}


def kernel_source_hash():
    """sha256 over the device sources the PMC passes were taken on (uvc_amd/csrc/*.hip and the headers they include, comments and white space
    removed): the replayed counters are only quoted while the kernels are the ones that were counted."""
    import glob
    import hashlib
    import re
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "uvc_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "uvc_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        text = open(f, "r", errors="replace").read()
        # comments and white space do not make another kernel: a reworded comment must not declare the counters stale
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        text = re.sub(r"\s+", " ", text)
        h.update(os.path.basename(f).encode()); h.update(text.encode())
    return h.hexdigest()[:16]


_TRAFFIC = {}


def _traffic_table(tile_kb, depth):
    """profiles/traffic_latest.json when it was counted on this workload AND on these kernel sources, else None (`_why` says which)."""
    if "t" not in _TRAFFIC:
        _TRAFFIC["t"], _TRAFFIC["why"] = None, "profiles/traffic_latest.json is missing"
        path = os.path.join(ROOT, "profiles", "traffic_latest.json")
        try:
            t = json.load(open(path))
            if t.get("_workload") != {"tile_kb": tile_kb, "depth": depth}:
                _TRAFFIC["why"] = "the committed PMC passes were taken on another workload"
            elif t.get("_source_hash") != kernel_source_hash():
                _TRAFFIC["why"] = "stale: the kernel sources changed since the committed PMC passes (source hash %s, counted on %s); rerun scripts/gpu_round_profile.sh" % (kernel_source_hash(), t.get("_source_hash"))
            else:
                _TRAFFIC["t"], _TRAFFIC["why"] = t, "replayed from profiles/traffic_latest.json (separate rocprofv3 --pmc passes on these kernel sources, hash %s), not measured in this run" % t["_source_hash"]
        except (OSError, ValueError):
            pass
    return _TRAFFIC["t"]


SCORE_KERNELS = ("k_gate_scan", "k_enum", "k_gather", "k_dpv_pre", "k_dp4", "k_dpv_post", "k_qual", "k_call_group", "k_call_rec", "k_keep_scan", "k_keep_copy")


def replayed_traffic(kernel, tile_kb, depth):
    """HBM-side bytes per launch of `kernel`, REPLAYED from the committed PMC passes (profiles/traffic_latest.json, made by
    scripts/gpu_round_profile.sh: FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes) -- not measured in this run.  None when the passes were
    taken on another workload or on other kernel sources."""
    t = _traffic_table(tile_kb, depth)
    if t is None:
        return None
    try:
        for name in PROFILE_NAMES.get(kernel, (kernel,)):
            if name in t:
                return t[name]["traffic_bytes_per_launch"]
        return None
    except (KeyError, ValueError):
        return None


def valu_issue(kernel, kernel_ms, args):
    """What actually bounds the kernel: the share of the chip's VALU issue rate it uses, from the replayed SQ_INSTS_VALU count and this
    run's duration.  Two denominators: the hardware peak (1024 SIMD-32s, a wave64 VALU instruction every 2 cycles, 2.4 GHz max clock --
    MI355X_MICROARCH.md) and what scripts/ubench/issue_model.hip sustains on this chip with >= 2 waves per SIMD (3.0 cycles per
    instruction at the nominal clock; the scalar instructions of the same loops, ~0.6 per vector one, issue beside them at 4.35)."""
    n = replayed_valu(kernel, args.tile_kb, args.depth)
    if not n or not kernel_ms:
        return None
    simd_cycles = 1024 * (kernel_ms * 1e-3) * 2.4e9
    return {"valu_wave_instructions_per_launch": n, "frac_of_peak_issue_rate": n * 2.0 / simd_cycles, "frac_of_measured_issue_rate": n * 3.0 / simd_cycles,
            "clock_assumed_ghz": 2.4,
            "source": "SQ_INSTS_VALU replayed from profiles/traffic_latest.json; the kernel is bound by instruction issue (vector + scalar pipes), "
                      "not by HBM bytes (DESIGN.md section 6a)"}


def replayed_valu(kernel, tile_kb, depth):
    """VALU wave-instructions per launch of `kernel`, REPLAYED from the committed PMC pass (SQ_INSTS_VALU in profiles/traffic_latest.json)."""
    t = _traffic_table(tile_kb, depth)
    if t is None:
        return None
    for name in PROFILE_NAMES.get(kernel, (kernel,)):
        if name in t and "SQ_INSTS_VALU_per_launch" in t[name]:
            return t[name]["SQ_INSTS_VALU_per_launch"]
    return None


def _cpu_region(args):
    from uvc_amd import synth
    seed, region_len, depth, umi = args
    return synth.generate_region(seed=seed, region_len=region_len, depth=depth, umi=umi)


def run_cpu_baseline(depth, n_regions=64, region_len=20000, umi=False, all_out=False, what="accumulate + default-gate scoring"):
    """Times the oracle (test infrastructure) on `n_regions` independent regions, one thread each: accumulate + scoring."""
    from concurrent.futures import ThreadPoolExecutor
    from uvc_amd import _ffi, region
    lib = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_")
    params = region.default_params(lib)
    cores = min(16, os.cpu_count() or 1)
    regs = []
    for i in range(n_regions):
        r = _cpu_region((777 + i, region_len, depth, umi))
        R = region.Region(lib, params, r["tid"], r["beg"], r["end"], r["refseq"])
        R.set_reads(r)
        regs.append(R)

    def work(R):
        R.accumulate()
        return len(R.score(all_out=all_out)["refpos"])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(work, regs))
    dt = time.perf_counter() - t0
    for R in regs:
        R.close()
    return {"value": n_regions * region_len / dt, "unit": "positions/s", "cores": cores, "kind": "port",
            "sample": "%d regions x %d bp = %.2f Mb at %dx%s, oracle (scalar C++ port of the reference algorithm), one region per thread, %s, %.1f s wall"
                      % (n_regions, region_len, n_regions * region_len / 1e6, depth, " duplex-UMI" if umi else "", what, dt)}


def run_cpu_baseline_tn(depth_t, depth_n, n_regions=32, region_len=20000):
    """The T/N flow of config 5 on the oracle: tumor pass, keys from its written records, normal pass on the keys; one region pair per thread."""
    from concurrent.futures import ThreadPoolExecutor
    from uvc_amd import _ffi, region, synth
    lib = _ffi.Lib(__import__("oracle").library_path(), "uvc_oracle_")
    pt, pn = region.default_params(lib), region.default_params(lib)
    pn.tumor_vcf_is_provided = 1
    cores = min(16, os.cpu_count() or 1)
    pairs = []
    for i in range(n_regions):
        rt = synth.generate_region(seed=877 + i, region_len=region_len, depth=depth_t)
        rn = synth.generate_region(seed=877 + i, region_len=region_len, depth=depth_n, somatic_every=10 ** 9)
        Rt = region.Region(lib, pt, rt["tid"], rt["beg"], rt["end"], rt["refseq"]); Rt.set_reads(rt)
        Rn = region.Region(lib, pn, rn["tid"], rn["beg"], rn["end"], rn["refseq"]); Rn.set_reads(rn)
        pairs.append((Rt, Rn))

    def work(pr):
        Rt, Rn = pr
        Rt.accumulate()
        keys = tumor_keys_of(Rt.score())
        Rn.accumulate()
        return len(Rn.score(tumor_keys=keys)["refpos"])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(work, pairs))
    dt = time.perf_counter() - t0
    for Rt, Rn in pairs:
        Rt.close(); Rn.close()
    return {"value": n_regions * region_len / dt, "unit": "positions/s", "cores": cores, "kind": "port",
            "sample": "%d region pairs x %d bp at %dx / %dx, oracle, one pair per thread: tumor pass, keys from its written records, normal pass on the keys, %.1f s wall" % (n_regions, region_len, depth_t, depth_n, dt)}


def tumor_keys_of(rec):
    """The tumor records that reach the VCF (keep && out) as UvcTumorKey tuples (TumorKeyInfo, main_conversion.hpp:490-529; the fields
    rescue_variants_from_vcf reads back, main.cpp:183-398): what the normal pass of uvcTN.sh receives through --tumor-vcf."""
    w = np.nonzero((rec["keep"] != 0) & (rec["out"] != 0))[0]
    keys = set()
    for i in w:
        sym = int(rec["symbol"][i])
        keys.add((int(rec["refpos"][i]), sym, int(rec["cDP1x"][i]), int(rec["CDP1x0"][i]), int(rec["bAD"][i]), int(rec["bDP"][i]), int(rec["tier2"][i]),
                  int(rec["gapSa_len"][i]) if sym in (7, 8, 9, 10, 11, 12) else 0, int(rec["cVQ1"][i]), int(rec["cPCQ1"][i]), int(rec["cDP2x"][i]), int(rec["CDP2x0"][i]),
                  int(rec["cVQ2"][i]), int(rec["cPCQ2"][i]), int(rec["bNMQ"][i]), int(rec["vHGQ"][i]), int(rec["DP"][i])))
    first = {}
    for k in sorted(keys):   # one record per (refpos, symbol, InDel length): the key of the tumor map
        first.setdefault((k[0], k[1], k[7]), k)
    return sorted(first.values(), key=lambda k: (k[0], k[1]))


def _gen_tile(a):
    from uvc_amd import synth
    seed, region_len, depth, beg, umi = a[:5]
    extra = dict(a[5]) if len(a) > 5 else {}
    if os.environ.get("UVC_BENCH_INDEL_EVERY"):   # experiment knob: spacing of the synthetic InDels (0 = none); the default workload does not set it
        extra["indel_every"] = int(os.environ["UVC_BENCH_INDEL_EVERY"])
    return synth.generate_region(seed=seed, region_len=region_len, depth=depth, beg=beg, umi=umi, **extra)


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (this process has not touched a GPU and never will),
    wait for them, pass rank 0's JSON line through."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=(None if rank == 0 else subprocess.DEVNULL)))
    rc = 0
    for p in procs:
        rc = rc or p.wait()
    return rc


class Leg:
    """One stream of tiles through region handles: reset (side arrays of the tile's own reference) + set_reads + accumulate, then score."""
    def __init__(self, lib, region, params, tiles, dev, all_out=False, all_records=False):
        self.lib, self.region, self.tiles, self.all_out, self.all_records = lib, region, tiles, all_out, all_records
        self.T = len(tiles)
        self.dreads = [region.device_reads(t, dev) for t in tiles]
        self.Rs = [region.Region(lib, params, t["tid"], t["beg"], t["end"], t["refseq"]) for t in tiles]
        for R in self.Rs:
            lib.dll.uvcgpu_region_set_profiling(R.h, 1)          # HIP events around the kernels of every accumulate / score (the roofline legs read them)
        self.refs = [t["refseq"].encode() if isinstance(t["refseq"], str) else bytes(t["refseq"]) for t in tiles]
        rl = tiles[0]["end"] - tiles[0]["beg"]
        self.cap = 15 * (rl + 2) if all_out else max(65536, rl // 4)
        self.names_buf = C.create_string_buffer(4096)
        self.ms_buf = (C.c_float * 64)()

    def prepare(self, k, host=False):
        R, t = self.Rs[k % self.T], self.tiles[k % self.T]
        if not os.environ.get("UVC_BENCH_NO_RESET"):
            # the tile's own reference: CHAR_TO_SYMBOL, refstring2repeatvec and the two BAQ prefix-sum arrays on the device (uvc_rtr.hip)
            R.reset(t["tid"], t["beg"], t["end"], self.refs[k % self.T])
        if host:
            R.set_reads(t)                                     # PCIe-inclusive variant: the columns start in (pinned) host memory
        else:
            R.set_reads_device(self.dreads[k % self.T])
        R.accumulate()

    def finish(self, k):
        R = self.Rs[k % self.T]
        rec = R.score(all_out=self.all_out, capacity=self.cap, copy=False, release_state=True, kept_only=not self.all_records)
        sc = C.c_int64()
        self.lib.dll.uvcgpu_region_last_score_counts(R.h, C.byref(sc), None)
        self.scored = sc.value                                 # records scored (the kept-only form returns fewer)
        return rec

    def kernel_times(self, R, into):
        n = self.lib.dll.uvcgpu_region_kernel_times(R.h, self.names_buf, 4096, self.ms_buf, 64)   # the handle's stream is idle here: its score() was synchronous
        for nm, ms in zip(self.names_buf.value.decode().split(";")[:n], list(self.ms_buf)[:n]):
            into.setdefault(nm, []).append(ms)

    def run_stream(self, k0, n_steps, host=False, ktimes=None, serial=False):
        """n_steps tiles, software-pipelined: tile k + 1 is prepared and its accumulate enqueued before tile k is scored."""
        n_rec, T = 0, self.T
        if serial or T < 2:   # one handle cannot hold the next tile's reads while the current one is still being scored
            for k in range(k0, k0 + n_steps):
                self.prepare(k, host); n_rec = len(self.finish(k)["refpos"])
                if ktimes is not None:
                    self.kernel_times(self.Rs[k % T], ktimes)
            return n_rec
        # tiles prepared (reset + set_reads + accumulate enqueued) beyond the one being scored; two ahead = three tiles in flight measured 9.10-9.13 ms
        # per step against 9.30-9.43 with one and 9.24-9.32 with three (round 4, one box)
        ahead = max(1, min(int(os.environ.get("UVC_BENCH_AHEAD", "2")), T - 1))
        for k in range(k0, min(k0 + ahead, k0 + n_steps)):
            self.prepare(k, host)
        for k in range(k0, k0 + n_steps):
            if k + ahead < k0 + n_steps:
                self.prepare(k + ahead, host)
            n_rec = len(self.finish(k)["refpos"])
            if ktimes is not None:
                self.kernel_times(self.Rs[k % T], ktimes)
        return n_rec

    def close(self):
        for R in self.Rs:
            R.close()
        self.Rs, self.dreads = [], []


def side_leg(lib, region, params, tiles, dev, torch, steps, all_out, depth, what):
    """A short timed stream of other tiles behind the main legs (config 4 shape, all-out scoring): ms per tile, per-kernel times, the
    scoring kernels' own roofline.  Inputs resident in HBM, same step as `value`."""
    leg = Leg(lib, region, params, tiles, dev, all_out=all_out, all_records=all_out)
    rl = tiles[0]["end"] - tiles[0]["beg"]
    leg.run_stream(0, leg.T); leg.run_stream(0, leg.T)                 # priming: every handle twice (buffers sized, records buffer page-locked)
    kt = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n_rec = leg.run_stream(0, steps, ktimes=kt)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    avg = {k: sum(v) / len(v) for k, v in kt.items()}
    out = {"workload": what, "tile_positions": rl, "tiles": leg.T, "steps": steps, "ms_per_step": 1e3 * dt / steps, "value": rl * steps / dt, "unit": "positions/s",
           "records_last_tile": n_rec, "reads_per_tile": int(tiles[0]["n_reads"]),
           "kernel_ms": {k: round(v, 4) for k, v in sorted(avg.items(), key=lambda kv: -kv[1])}}
    if "k_score_all" in avg:
        out["roofline_score"] = score_roofline(avg["k_score_all"], rl + 1, leg.scored, all_out)
    b_d = 2.0 * depth + 48.0 * depth / READ_LEN + 48 + 5544                  # B(D), whole path
    out["path_roofline_frac"] = b_d * rl / (1e-3 * out["ms_per_step"]) / 1e9 / HBM_PEAK_GBS
    leg.close()
    return out


def config5_leg(lib, region, params, tumor, normal, dev, torch, repeats=3):
    """BASELINE config 5 shape on one GPU: the two passes of uvcTN.sh:120-127 on one tile pair -- tumor pass (300x: reset, set_reads, accumulate,
    default-gate scoring, records to the host), the written records as UvcTumorKey[] (what --tumor-vcf carries, main.cpp:183-398), normal
    pass (100x, tumor_vcf_is_provided: the keyed positions are scored, every symbol of them, NLODQ / SomaticQ through the T/N arms).
    Positions per second of the PAIR; columns resident in HBM; strictly one pass after the other (the normal pass needs the tumor's keys)."""
    pn = region.default_params(lib)
    pn.tumor_vcf_is_provided = 1
    rl = tumor["end"] - tumor["beg"]
    dt_, dn_ = region.device_reads(tumor, dev), region.device_reads(normal, dev)
    Rt = region.Region(lib, params, tumor["tid"], tumor["beg"], tumor["end"], tumor["refseq"])
    Rn = region.Region(lib, pn, normal["tid"], normal["beg"], normal["end"], normal["refseq"])
    ref_t = tumor["refseq"].encode() if isinstance(tumor["refseq"], str) else bytes(tumor["refseq"])
    ref_n = normal["refseq"].encode() if isinstance(normal["refseq"], str) else bytes(normal["refseq"])
    cap = max(65536, rl // 4)
    times, n_keys, n_rec = [], 0, 0
    for rep_ in range(repeats + 1):   # the first pair primes the handles
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Rt.reset(tumor["tid"], tumor["beg"], tumor["end"], ref_t); Rt.set_reads_device(dt_); Rt.accumulate()
        rec = Rt.score(capacity=cap, copy=False, release_state=True, kept_only=True)
        t1 = time.perf_counter()
        keys = tumor_keys_of(rec)
        t2 = time.perf_counter()
        Rn.reset(normal["tid"], normal["beg"], normal["end"], ref_n); Rn.set_reads_device(dn_); Rn.accumulate()
        recn = Rn.score(capacity=cap, copy=False, release_state=True, kept_only=True, tumor_keys=keys)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        if rep_:
            times.append((t3 - t0, t1 - t0, t2 - t1, t3 - t2))
        n_keys, n_rec = len(keys), len(recn["refpos"])
    Rt.close(); Rn.close()
    times.sort()
    med = times[len(times) // 2]
    return {"workload": "BASELINE config 5 shape: one tumor (300x) / normal (100x) tile pair of %d kb, non-UMI, columns resident in HBM; tumor pass -> its written records as UvcTumorKey[] -> normal pass "
                        "(uvcTN.sh:120-127, main.cpp:183-398, 1073-1169); the two passes strictly one after the other" % (rl // 1000),
            "tile_positions": rl, "repeats": len(times), "ms_per_pair": 1e3 * med[0], "ms_tumor_pass": 1e3 * med[1], "ms_keys_host": 1e3 * med[2], "ms_normal_pass": 1e3 * med[3],
            "ms_per_pair_min_max": [1e3 * times[0][0], 1e3 * times[-1][0]], "value": rl / med[0], "unit": "positions/s (of the pair)", "tumor_keys": n_keys, "normal_records_returned": n_rec,
            "reads_tumor": int(tumor["n_reads"]), "reads_normal": int(normal["n_reads"])}


def score_roofline(ms, npos, n_records, all_out):
    """The scoring kernels (candidate gate + scan + k_score + k_call + kept-groups copy; HIP events around uvc_launch_score) against HBM.
    `nominal`: SURVEY 8(d)'s per-position figure for the scoring kernel (5.6 KB default gate: the whole accumulator record read once;
    8.3 KB all-out).  `needed`: what the kernels have to move at the default gate -- the gate reads the 2 x 14 fragment-depth cells of every
    position (112 B), and only a scored record reads its ~5.5 KB of state and writes its fields."""
    nominal = (8300.0 if all_out else 5600.0) * npos
    needed = nominal if all_out else 112.0 * npos + n_records * (5544.0 + 4.0 * 160)
    return {"bound": "hbm", "kernels": "k_gate_scan + k_enum + k_gather + k_dpv_pre + k_dp4 + k_dpv_post + k_qual + k_call (+ k_keep_scan + k_keep_copy): HIP events around uvc_launch_score", "kernel_ms": ms, "records": n_records,
            "algorithmic_bytes_nominal": nominal, "achieved_nominal": nominal / (ms * 1e-3) / 1e9, "frac_nominal": nominal / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes": needed, "achieved": needed / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": needed / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--repeats", type=int, default=3, help="how many times the timed region of K steps is run; `value` is the median")
    ap.add_argument("--tile-kb", type=int, default=1000, help="tile length in kb (chr20 is processed as 1 Mb tiles)")
    ap.add_argument("--tiles", type=int, default=8, help="number of distinct tiles per rank in the stream")
    ap.add_argument("--total-tiles", type=int, default=0, help="STRONG scaling: one job of this many distinct tiles (config 3's shape: a tile list cut into contiguous region shards, "
                                                                "uvcio_plan_shards) dealt to the N ranks; each rank streams its own shard once; value = all positions / max-over-ranks time")
    ap.add_argument("--depth", type=int, default=300)
    ap.add_argument("--umi", action="store_true", help="duplex-UMI families (BASELINE config 4 shape when combined with --depth 2000 --tile-kb 200)")
    ap.add_argument("--all-out", action="store_true", help="second series of SURVEY 8(d): score every symbol of every position (-A), not only the default-gate candidates")
    ap.add_argument("--all-records", action="store_true", help="D2H of every scored record (the round-1 form) instead of only the record groups the VCF writer reads (UvcScoreRequest::kept_only)")
    ap.add_argument("--serial", action="store_true", help="no pipelining across tiles: preparation, accumulate and score of a tile strictly one after the other")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the pcie_inclusive and resident measurements behind the timed region (profiler runs)")
    ap.add_argument("--no-side", action="store_true", help="skip the config-4 and all-out side objects (they need ~1 min of synthetic data generation)")
    ap.add_argument("--dry-run", action="store_true", help="CPU-only rehearsal of the launch protocol (no kernels): ranks, barrier, max-over-ranks clock")
    args = ap.parse_args()
    # the validator of the test suite (a sweep over every plane behind each accumulate, ~3 ms per 1 Mb tile) has no place inside a timed step
    if os.environ.pop("UVCGPU_CHECK_PRESENCE", None):
        print("bench.py: UVCGPU_CHECK_PRESENCE is a test switch; ignored here", file=sys.stderr)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    from uvc_amd import shard
    rank, local_rank, world = shard.dist_env()
    args.gpus = world
    region_len = args.tile_kb * 1000

    if args.total_tiles > 0 and args.total_tiles < world:
        sys.exit("--total-tiles %d: fewer tiles than ranks (%d): a rank would have nothing to time" % (args.total_tiles, world))
    if args.dry_run:
        clock = shard.Clock(backend="gloo")
        strong_job = None
        if args.total_tiles > 0:   # the strong-scaling cut, as the real run makes it: every tile owned once, every rank owns some
            cut = shard.plan_contiguous([1000] * args.total_tiles, world)
            mine = [i for i in range(args.total_tiles) if int(cut[i]) == rank]
            assert mine == list(range(mine[0], mine[-1] + 1)), "a shard is a contiguous run of the tile list"
            args.steps = len(mine)
            strong_job = {"total_tiles": args.total_tiles, "tiles_owned_sum": int(clock.sum_over_ranks(float(len(mine)))), "tiles_of_a_rank_min": int(clock.min_over_ranks(float(len(mine)))),
                          "tiles_of_a_rank_max": int(clock.max_over_ranks(float(len(mine))))}
        clock.barrier(); t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(0.01 * (1 + rank))
        clock.barrier(); own = time.perf_counter() - t0; dt = clock.max_over_ranks(own)
        total_positions = (clock.sum_over_ranks(float(region_len) * args.steps) if strong_job else float(world * region_len * args.steps))
        own_min = clock.min_over_ranks(own)
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": total_positions / dt, "unit": "positions/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / max(1, args.steps), "higher_is_better": True,
                              "scaling": ("strong" if strong_job else "weak"), "vs_baseline": None, "dtype": "int32", "data": "synthetic",
                              "config": {"workload": "dry-run", "strong_scaling_job": strong_job}, "dry_run": True,
                              "per_rank_ms_per_step": {"min": 1e3 * own_min / max(1, args.steps), "max": 1e3 * dt / max(1, args.steps)}}))
        clock.close()
        return

    # ---- synthetic tiles (host, before any GPU call so that the generator may fork) ----
    side = (world == 1) and not (args.no_side or args.no_extras or args.umi or args.all_out or args.tile_kb != 1000 or args.depth != 300)
    t_gen = time.perf_counter()
    # every rank keeps to its own share of the host cores (the generator's workers, the launch thread, the library's pools): N ranks of one node
    # must not oversubscribe each other
    try:
        cores = sorted(os.sched_getaffinity(0))
        mine = cores[rank * len(cores) // world:(rank + 1) * len(cores) // world]
        if world > 1 and mine:
            os.sched_setaffinity(0, mine)
    except (AttributeError, OSError):
        pass
    strong = args.total_tiles > 0
    if strong:
        # the same job for every N: tile i of the list has seed 12345 + 1000 i whoever owns it; rank r owns the r-th of N contiguous runs of equal cost
        # (equal tiles: the cut of uvcio_plan_shards / shard.plan_contiguous)
        cut = shard.plan_contiguous([1000] * args.total_tiles, world)
        own = [i for i in range(args.total_tiles) if int(cut[i]) == rank]
        specs = [(12345 + 1000 * i, region_len, args.depth, 1000000 + i * (region_len + 1000), args.umi) for i in own]
        args.tiles = len(specs)
        args.steps = len(specs)          # the timed region of a rank: its shard, every tile once
        args.warmup = min(args.warmup, len(specs))
    else:
        specs = [(12345 + rank + 1000 * i, region_len, args.depth, 1000000 + i * (region_len + 1000), args.umi) for i in range(args.tiles)]
    n_main = len(specs)
    if side:
        specs += [(4000 + i, 200000, 2000, 1000000 + i * 201000, True) for i in range(2)]      # BASELINE config 4 shape: 200 kb duplex-UMI panel tiles at 2000x
        specs += [(5000, 200000, args.depth, 1000000, False)]                                   # all-out scoring (-A) tile
        # BASELINE config 5 shape: one tumor (300x) / normal (100x) tile pair over the same reference (same seed: the reference is drawn first), no somatic variants in the normal
        specs += [(6000, region_len, 300, 1000000, False), (6000, region_len, 100, 1000000, False, {"somatic_every": 10 ** 9})]
    workers = max(1, min(len(specs), 6 if world == 1 else 2, max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 2) // max(1, world))))
    if workers > 1:
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=workers) as ex:
            tiles = list(ex.map(_gen_tile, specs))
    else:
        tiles = [_gen_tile(sp) for sp in specs]
    t_gen = time.perf_counter() - t_gen

    # the form the columns are handed over in: BAM's own 4-bit bases and no offset columns (UvcReadSoA::bases4, seq_off = cigar_off = NULL:
    # 530 instead of 712 MB per tile; the library unpacks and scans on the device) unless UVC_BENCH_PLAIN asks for one byte per base + offsets
    COLS = ("pos", "mpos", "isize", "flag", "mapq", "nm", "l_qseq", "seq_off", "cigar_off", "n_cigar", "frag_id", "fam_id", "fam_strand", "bases", "bases4", "quals", "cigars")
    if not os.environ.get("UVC_BENCH_PLAIN"):
        from uvc_amd import region as _region
        tiles = [_region.compact_form(t) for t in tiles]
    side_tiles = tiles[n_main:]
    tiles = tiles[:n_main]
    import torch
    # rehearsal of the N-rank path on a one-GPU box: UVC_BENCH_DEVICE puts every rank on one device, UVC_BENCH_CLOCK_BACKEND=gloo keeps the
    # barrier / max-over-ranks off RCCL (which refuses two ranks on one device); a real run sets neither
    if os.environ.get("UVC_BENCH_DEVICE"):
        local_rank = int(os.environ["UVC_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    clock = shard.Clock(backend=os.environ.get("UVC_BENCH_CLOCK_BACKEND", "nccl"))
    from uvc_amd import region
    lib = region.gpu_lib()
    rc = lib.dll.uvcgpu_init(local_rank)
    if rc != 0:
        raise RuntimeError(lib.last_error())
    lib.dll.uvcgpu_region_last_score_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.dll.uvcgpu_region_set_profiling.argtypes = [C.c_void_p, C.c_int]
    lib.dll.uvcgpu_region_kernel_times.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_float), C.c_int]
    params = region.default_params(lib)

    # inputs resident in HBM: the columns of every tile + one handle per tile (planes, scratch); the region side arrays are rebuilt from the
    # tile's own reference inside every step (uvcgpu_region_reset)
    t_h2d = time.perf_counter()
    leg = Leg(lib, region, params, tiles, dev, all_out=args.all_out, all_records=args.all_records)
    t_h2d = time.perf_counter() - t_h2d
    Rs, T = leg.Rs, leg.T
    n_reads_tile = int(np.mean([int(t["n_reads"]) for t in tiles]))
    input_bytes_tile = int(np.mean([sum(int(np.asarray(t[k]).nbytes) for k in COLS if t.get(k) is not None) for t in tiles]))
    prepare, finish, kernel_times = leg.prepare, leg.finish, leg.kernel_times

    # UVC_BENCH_VALUE_THREADS=n: the timed stream dealt to n host threads, one tile each at a time, as uvc1-mi355x deals tiles to its workers,
    # instead of the one-thread software pipeline.  One host thread issues ~400 launches per tile (reset, read preparation with three count
    # read-backs, accumulate, scoring) and sits in the preparation's read-backs, so the preparations of successive tiles never overlap: on one
    # box 9.67 ms per tile with one thread, 9.2-9.35 with two, 9.15 with three, 9.25 with four.  The default stays the one-thread pipeline:
    # there the heavy kernels of successive tiles seldom meet on the device (k_p2_fast<base>: 1.61 ms +- 0.03 over the launches of a rocprofv3
    # trace, the same as on an otherwise idle device), so `roofline`'s in-stream duration is the kernel's own and agrees with the committed
    # trace; with two threads the same kernel spreads over 1.6 .. 3.3 ms.  The four-thread rate is reported beside `value` as `resident_in_flight4`.
    n_thr_value = max(1, min(int(os.environ.get("UVC_BENCH_VALUE_THREADS", "1")), T))
    def run_in_flight(k0, n_steps, n_thr, host=False, ktimes=None):
        import threading
        locks = [threading.Lock() for _ in range(T)]
        out = [0]
        def work(w):
            lib.dll.uvcgpu_init(local_rank)                      # hipSetDevice is per host thread
            for k in range(k0 + w, k0 + n_steps, n_thr):
                with locks[k % T]:                               # a handle serves one tile at a time
                    prepare(k, host); out[0] = len(finish(k)["refpos"])
                    if ktimes is not None:
                        kernel_times(leg.Rs[k % T], ktimes)
        th = [threading.Thread(target=work, args=(w,)) for w in range(n_thr)]
        for t in th: t.start()
        for t in th: t.join()
        return out[0]

    def run_stream(k0, n_steps, host=False, ktimes=None):
        if n_thr_value > 1 and not args.serial:
            return run_in_flight(k0, n_steps, n_thr_value, host=host, ktimes=ktimes)
        return leg.run_stream(k0, n_steps, host=host, ktimes=ktimes, serial=args.serial)

    run_stream(0, T)              # untimed priming: every handle once (its first set_reads sizes the cached device blocks, its first score page-locks its records buffer)
    run_stream(0, args.warmup)    # the W warmup steps of the contract
    # the timed region, `--repeats` times (default 3): exactly K steps each, bracketed by barrier + synchronize on both sides, max over ranks;
    # `value` is the MEDIAN repeat (boxes and runs differ by a few per cent: one 0.2 s region is a sample, not a measurement), all of them are on the line
    ktimes = {}
    reps = []
    for rep_ in range(max(1, args.repeats)):
        torch.cuda.synchronize(); clock.barrier()
        t0 = time.perf_counter()
        n_rec = run_stream(args.warmup + rep_ * args.steps, args.steps, ktimes=ktimes)
        torch.cuda.synchronize(); clock.barrier()
        own_dt = time.perf_counter() - t0
        reps.append((clock.max_over_ranks(own_dt), clock.min_over_ranks(own_dt)))
    scored_main = leg.scored
    dt, dt_min = sorted(reps)[len(reps) // 2]
    total_positions = (clock.sum_over_ranks(float(region_len) * args.steps) if strong else clock.sum_over_ranks(float(region_len)) * args.steps)

    pcie = resident = in_flight4 = None
    if not args.no_extras and not args.serial and T >= 4:
        # (0) the timed stream again with four tiles in flight on four host threads (inputs resident, same step)
        nf = max(8, min(args.steps, 2 * T))
        run_in_flight(0, 4, 4)
        torch.cuda.synchronize(); clock.barrier(); ts = time.perf_counter()
        run_in_flight(4, nf, 4)
        torch.cuda.synchronize(); clock.barrier(); fdt = clock.max_over_ranks(time.perf_counter() - ts)
        in_flight4 = {"value": clock.sum_over_ranks(float(region_len)) * nf / fdt, "unit": "positions/s", "ms_per_step": 1e3 * fdt / nf, "steps": nf, "tiles_in_flight": 4,
                      "note": "the step of `value` (inputs resident in HBM), the tiles dealt to four host threads as uvc1-mi355x deals them to its workers: the host's launch and read-back "
                              "time of one tile hides under the other tiles' kernels; whole-job aggregate over all ranks, measured behind the timed region"}
    if not args.no_extras:
        # (1) the same stream with the columns in pinned host memory: H2D of every column inside the step
        # the columns move into page-locked memory of the library's own (uvcgpu_host_alloc = hipHostMalloc): a numpy array page-locked in place
        # (uvcgpu_pin_host_buffer = hipHostRegister over 4 KiB heap pages) is copied at 29 GB/s on this box, library-owned memory at 57 GB/s
        # (gpurun_out/trace_pcie, scripts/gpu_trace_pcie.sh) -- a binding that fills the UvcReadSoA columns anyway fills them there
        pinned, host_bufs = [], []
        lib.dll.uvcgpu_host_alloc.restype, lib.dll.uvcgpu_host_alloc.argtypes = C.c_int, [C.POINTER(C.c_void_p), C.c_int64]
        lib.dll.uvcgpu_host_free.restype, lib.dll.uvcgpu_host_free.argtypes = C.c_int, [C.c_void_p]
        for t in tiles:
            for k in COLS:
                if t.get(k) is None:
                    continue
                a = np.ascontiguousarray(t[k])
                hp = C.c_void_p()
                if a.nbytes and not os.environ.get("UVC_BENCH_REGISTER") and lib.dll.uvcgpu_host_alloc(C.byref(hp), C.c_int64(a.nbytes)) == 0 and hp.value:
                    host_bufs.append(hp)
                    b = np.ctypeslib.as_array((C.c_uint8 * a.nbytes).from_address(hp.value)).view(a.dtype).reshape(a.shape)
                    b[...] = a
                    t[k] = b
                    continue
                t[k] = a
                if a.nbytes and lib.dll.uvcgpu_pin_host_buffer(C.c_void_p(a.ctypes.data), C.c_int64(a.nbytes)) == 0:
                    pinned.append(a)
        n_extra = max(4, min(max(args.steps, 12), 2 * T))
        n_thr = max(1, min(int(os.environ.get("UVC_BENCH_THREADS", "4")), T))
        import threading
        handle_locks = [threading.Lock() for _ in range(T)]

        def run_threads(k0, n_steps):
            """Tiles in flight on host threads, as uvc1-mi355x runs them: thread w takes tiles k0 + w, k0 + w + n_thr, ...; the copy of one tile
            travels while the kernels of another run."""
            import threading
            def work(w):
                lib.dll.uvcgpu_init(local_rank)                 # hipSetDevice is per host thread
                for k in range(k0 + w, k0 + n_steps, n_thr):
                    with handle_locks[k % T]:                    # a handle serves one tile at a time: a thread that runs ahead waits for it
                        prepare(k, True); finish(k)
            th = [threading.Thread(target=work, args=(w,)) for w in range(n_thr)]
            for t in th: t.start()
            for t in th: t.join()
        run_threads(0, n_thr)
        torch.cuda.synchronize(); clock.barrier(); ts = time.perf_counter()
        run_threads(n_thr, n_extra)
        torch.cuda.synchronize(); clock.barrier(); own_sdt = time.perf_counter() - ts; sdt = clock.max_over_ranks(own_sdt); sdt_min = clock.min_over_ranks(own_sdt)
        pcie = {"value": clock.sum_over_ranks(float(region_len)) * n_extra / sdt, "unit": "positions/s", "ms_per_step": 1e3 * sdt / n_extra, "steps": n_extra,
                "h2d_bytes_per_tile": input_bytes_tile, "tiles_in_flight": n_thr, "per_rank_ms_per_step": {"min": 1e3 * sdt_min / n_extra, "max": 1e3 * sdt / n_extra},
                "note": "THE UNIT OF SURVEY 8(d): every tile's columns start in page-locked host memory (uvcgpu_host_alloc) and the step is region reset (side arrays on the device) + H2D of the "
                        "columns + set_reads + accumulate + scoring + D2H of the records; %d tiles in flight on host threads so that one tile's copy runs under another's kernels; whole-job aggregate "
                        "over all ranks, measured behind the timed region.  `value` keeps the harness contract (inputs resident in HBM when the clock starts)" % n_thr}
        for a in pinned:
            lib.dll.uvcgpu_unpin_host_buffer(C.c_void_p(a.ctypes.data))
        torch.cuda.synchronize()
        for t in tiles:   # the views die with their memory
            for k in COLS:
                t[k] = None
        for hp in host_bufs:
            lib.dll.uvcgpu_host_free(hp)
        # (2) one prepared tile, accumulate + score again and again, nothing overlapped
        R = Rs[0]
        R.set_reads_device(leg.dreads[0])
        sk = {}
        R.accumulate(); finish(0)
        torch.cuda.synchronize(); ts = time.perf_counter()
        n_res = 8
        for _ in range(n_res):
            R.accumulate(); finish(0)
            kernel_times(R, sk)
        torch.cuda.synchronize(); rdt = clock.max_over_ranks(time.perf_counter() - ts)
        resident = {"value": clock.sum_over_ranks(float(region_len)) * n_res / rdt, "unit": "positions/s", "ms_per_step": 1e3 * rdt / n_res,
                    "kernel_ms": {k: round(sum(v) / len(v), 4) for k, v in sorted(sk.items(), key=lambda kv: -sum(kv[1]))},
                    "note": "one tile whose reads are already prepared (set_reads done once): accumulate P1..P5b + scoring + D2H per step, one handle, nothing overlapped; measured behind the timed region"}

    # side objects (rank 0 of a one-GPU run): BASELINE config 4 shape and the all-out series of SURVEY 8(d); the main handles give their HBM back first
    side_out = {}
    if side and side_tiles:
        leg.close()
        side_out["config4"] = side_leg(lib, region, params, side_tiles[:2], dev, torch, 4, False, 2000,
                                       "BASELINE config 4 shape: 2 distinct 200 kb duplex-UMI panel tiles at 2000x, pipelined over two handles, inputs resident in HBM, same step as `value`")
        side_out["all_out"] = side_leg(lib, region, params, side_tiles[2:3], dev, torch, 3, True, args.depth,
                                       "second series of SURVEY 8(d): one 200 kb non-UMI tile at %dx with -A (every symbol of every position scored, 14 records per position, every record returned)" % args.depth)
        side_out["config5"] = config5_leg(lib, region, params, side_tiles[3], side_tiles[4], dev, torch)
        if not args.no_cpu_baseline:   # the oracle on bounded samples of the same shapes, beside each side leg (both series, every config)
            side_out["config4"]["cpu_baseline"] = run_cpu_baseline(2000, n_regions=16, region_len=4000, umi=True, what="accumulate + default-gate scoring")
            side_out["all_out"]["cpu_baseline"] = run_cpu_baseline(args.depth, n_regions=32, region_len=10000, all_out=True, what="accumulate + all-out scoring (-A)")
            side_out["config5"]["cpu_baseline"] = run_cpu_baseline_tn(300, 100, n_regions=32, region_len=10000)

    if rank == 0:
        avg = {k: sum(v) / len(v) for k, v in ktimes.items()}
        score_ms = avg.pop("k_score_all", None)
        dom = max(avg, key=avg.get)
        npos_tile = region_len + 1
        abytes = algorithmic_bytes_per_position(dom, args.depth) * npos_tile   # per launch: one region
        achieved = abytes / (avg[dom] * 1e-3) / 1e9
        out = {
            "metric": METRIC, "value": total_positions / dt, "unit": "positions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": ("strong" if strong else "weak"), "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "per_rank_ms_per_step": {"min": 1e3 * dt_min / args.steps, "max": 1e3 * dt / args.steps},
            "repeats": {"n": len(reps), "ms_per_step": [round(1e3 * r[0] / args.steps, 4) for r in reps], "ms_per_step_min": 1e3 * min(r[0] for r in reps) / args.steps,
                        "ms_per_step_max": 1e3 * max(r[0] for r in reps) / args.steps, "value_is": "the median repeat (each repeat: exactly `steps` steps between barrier + synchronize, max over ranks)"},
            "config": {"workload": "stream of %d DISTINCT chr20-shaped tumor-only %s tiles per GPU (%d kb at %dx, 150 bp paired-end), UvcReadSoA columns (BAM 4-bit bases, qualities, per-read fields) resident in HBM; "
                                   "step = one tile: region reset (CHAR_TO_SYMBOL, refstring2repeatvec, BAQ prefix sums of the tile's own reference on the device) + set_reads(offset scans, base unpack, CIGAR facts + "
                                   "family/fragment nesting, radix orders, k_aln_prelude, k_build_p2list) + accumulate P1..P5b + "
                                   "default-gate scoring / calling + D2H of %s%s" % (T, "duplex-UMI" if args.umi else "non-UMI", args.tile_kb, args.depth,
                                   "every scored record" if args.all_records else "the record groups the VCF writer reads (kept_only: written records + the REF / genotype records of their positions)",
                                   "; tiles strictly one after the other" if args.serial else ("; %d tiles in flight, one host thread each, every tile on its own handle" % n_thr_value if n_thr_value > 1 else "; tiles software-pipelined over their handles (tiles k+1 and k+2 are prepared and accumulating while tile k is scored)")),
                       "tile_positions": region_len, "distinct_tiles": T, "pipelined": (not args.serial) and T >= 2, "tiles_in_flight": (1 if args.serial else (n_thr_value if n_thr_value > 1 else 1 + max(1, min(int(os.environ.get("UVC_BENCH_AHEAD", "2")), T - 1)))), "all_out": bool(args.all_out), "umi": bool(args.umi), "reads_per_tile": n_reads_tile,
                       "read_bases_per_tile": n_reads_tile * READ_LEN, "input_bytes_per_tile": input_bytes_tile, "returned_records_last_tile": n_rec, "kept_only": not args.all_records,
                       "parallelism": "region-shard x%d (no collective on the data path)" % world,
                       "strong_scaling_job": ({"total_tiles": args.total_tiles, "tiles_of_rank0": T, "note": "--total-tiles: one tile list cut into N contiguous shards of equal cost; `steps` / `ms_per_step` are rank 0's; "
                                               "value = positions of the whole job / the slowest rank's time"} if strong else None)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": replayed_traffic(dom, args.tile_kb, args.depth), "traffic_source": _TRAFFIC.get("why"),
                         "algorithmic_bytes_per_launch": abytes, "kernel_ms": avg[dom],
                         "valu_issue": valu_issue(dom, avg[dom], args),
                         "note": "HIP events on the library's stream over the timed steps; with pipelining another tile's preparation kernels may run beside the kernel (see resident.kernel_ms for the undisturbed durations)",
                         "dominant_by": "the largest average HIP-event duration inside the pipelined stream of THIS run.  k_frag16 and k_p2_fast<base> are within a few per cent of each other (1.52 / 1.54 ms "
                                        "undisturbed) and swap places between runs and between this figure and the committed rocprofv3 summary (profiles/): rocprofv3 sums a shorter profiled command in which "
                                        "the two tiles in flight overlap differently.  resident.roofline_by_kernel prices both on their undisturbed durations."},
            "kernel_ms": {k: round(v, 4) for k, v in sorted(avg.items(), key=lambda kv: -kv[1])},
            "read_bases_per_s": n_reads_tile * READ_LEN * world * args.steps / dt,
            "host_prep_s": {"generate": round(t_gen, 2), "columns_to_hbm": round(t_h2d, 3)},
        }
        if score_ms:
            out["roofline_score"] = score_roofline(score_ms, npos_tile, scored_main, args.all_out)
            # counter traffic of the scoring launches (replayed like roofline.traffic: only for the workload and the kernel sources the PMC passes saw)
            parts = {k: replayed_traffic(k, args.tile_kb, args.depth) for k in SCORE_KERNELS}
            if not args.all_out and all(v is not None for v in parts.values()):
                tot = float(sum(parts.values()))
                out["roofline_score"]["traffic"] = tot
                out["roofline_score"]["traffic_over_algorithmic"] = round(tot / out["roofline_score"]["algorithmic_bytes"], 3)
                out["roofline_score"]["traffic_by_kernel"] = {k: v for k, v in sorted(parts.items(), key=lambda kv: -kv[1])}
                out["roofline_score"]["traffic_note"] = ("FETCH_SIZE x 2 + WRITE_SIZE per launch, summed over the scoring kernels; k_gather's share is 128-byte lines fetched for 4-byte cells "
                                                         "(profiles/r04_fetch_calibration.txt: every fabric read is a 128-byte request)")
            else:
                out["roofline_score"]["traffic"] = None
        out.update(side_out)
        if in_flight4:
            out["resident_in_flight4"] = in_flight4
        if pcie:
            out["pcie_inclusive"] = pcie
        if resident:
            rd = resident["kernel_ms"].get(dom)
            if rd:
                resident["roofline_frac"] = abytes / (rd * 1e-3) / 1e9 / HBM_PEAK_GBS
            # the same figure for every kernel that has an entry in the algorithmic-bytes table, on its undisturbed duration (a small kernel whose
            # inputs the kernel before it has just written reads them from the 256 MB Infinity Cache: k_thres' counter traffic is below its
            # algorithmic bytes and its fraction of the HBM peak says little)
            resident["roofline_by_kernel"] = {
                k: {"kernel_ms": ms, "algorithmic_bytes_per_launch": algorithmic_bytes_per_position(k, args.depth) * npos_tile,
                    "frac": round(algorithmic_bytes_per_position(k, args.depth) * npos_tile / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "traffic": replayed_traffic(k, args.tile_kb, args.depth)}
                for k, ms in resident["kernel_ms"].items() if k in ("k_prep_fast", "k_thres", "k_p2_fast_link", "k_p2_fast_base", "k_frag", "k_p5b") and ms}
            out["resident"] = resident
            if resident["kernel_ms"].get("k_score_all") and "roofline_score" in out:
                out["roofline_score"]["undisturbed"] = score_roofline(resident["kernel_ms"]["k_score_all"], npos_tile, scored_main, args.all_out)
                out["roofline_score"]["note"] = ("kernel_ms = HIP events around the scoring kernels inside the pipelined stream (eleven short launches that queue behind the other tile's kernels: a span, "
                                                 "not a cost); `undisturbed` = the same events with one tile alone on the device (the resident leg)")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = run_cpu_baseline(args.depth)
        print(json.dumps(out))
    if leg.Rs:
        leg.close()
    clock.close()


if __name__ == "__main__":
    main()
