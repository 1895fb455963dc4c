// uvc1-mi355x -- BAM + FASTA -> block-gzipped VCF: the host chain of the reference's uvc1 (main.cpp:1196-1603, process_batch :458-1193)
// in C++ on the three C ABIs of this repository (uvcio.h readers / writer, uvcgroup.h family assignment, uvcgpu.h hot path + record text).
// The frequently used options keep the reference's names (CmdLineArgs.cpp:188-262): inputBAM -f -o -s --targets -R -t -A -q --outvar-flag
// --tumor-vcf --tn-is-paired --bed-out-fname --bed-in-fname.
//
// Region shards.  The reference fans its regions out over threads with `schedule(dynamic, 1)` and writes the chunk outputs in
// order (main.cpp:1478-1551); uvcTN.sh:92-101 adds one process per chromosome and `bcftools concat -n`.  Here:
//   * one process, several GPUs: --devices 0,1,.. (default: all visible).  Worker thread w binds to devices[w % n]; the workers pull tiles
//     from one queue (dynamic balance: a tile costs what its reads cost), each owns its file handles and one region handle that is reset
//     from tile to tile; the lines are written in tile order and a worker never runs more than 4 * threads tiles ahead of the writer.
//   * several processes: --shard i/n takes the i-th of n contiguous runs of the tile list, balanced by the compressed bytes the BAM index
//     attributes to each tile plus its length (reads and positions, main.cpp:1390-1392); shard 0 writes the header; `--concat out in..`
//     joins the shard outputs like bcftools concat -n.  No GPU ever talks to another one: there is no collective on this path.
// Regions are fixed tiles (--tile); every zerobased_pos has exactly one owner (UvcScoreRequest::base_at_pos_beg), so the tiles of one
// covered stretch write the records of one uncut region.
#include "uvcgpu.h"
#include "uvcgroup.h"
#include "uvcio.h"
#include "uvc_cpus.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {
const int32_t MAX_INSERT_SIZE = 2000, MAX_STR_N_BASES = 100;   // common.hpp:63-64

struct Opts {
    std::string bam, fasta, out, sample = "-", targets, bed, tumor_vcf, bed_out, bed_in, umi_struct;
    std::vector<int> devices;
    int threads = 0, outvar_flag = -1, repeat = 1, shard = 0, n_shards = 1, tn_is_paired = 0, tumor_format = 1;
    int64_t tile = 0;            // 0 = no fixed tiles: the regions are the reference's own cuts (uvcio_plan_regions = SamIter::iternext); --tile N overrides
    int64_t mem_per_thread = 1536;   // --mem-per-thread (MB), CmdLineArgs.hpp:33: enters the reference's region cuts
    bool all_out = false, timing = false, no_header = false, device_inflate = false;
    double vqual = -1e9;
};
[[noreturn]] void die(const std::string &m) { fprintf(stderr, "uvc1-mi355x: %s\n", m.c_str()); exit(2); }
void usage() {
    fprintf(stderr, "usage: uvc1-mi355x inputBAM -f ref.fa -o out.vcf.gz [-s sample] [--targets chr[:beg-end] | -R regions.bed] [-t threads] [-A] [-q vqual]\n"
                    "                   [--outvar-flag bits] [--tile bp (default: the reference's region cuts)] [--mem-per-thread MB] [--devices 0,1,..] [--shard i/n] [--no-header] [--timing] [--device-inflate]\n"
                    "                   [--tn-is-paired 0|1] [--tumor-vcf tumor.vcf.gz] [--is-tumor-format-retrieved 0|1] [--bed-out-fname f] [--bed-in-fname f]\n"
                    "       uvc1-mi355x --concat out.vcf.gz shard0.vcf.gz shard1.vcf.gz ...\n");
}
Opts parse(int argc, char **argv) {
    Opts o;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) die("missing value of " + a); return argv[++i]; };
        if (a == "-f" || a == "--fasta") o.fasta = val();
        else if (a == "-o" || a == "--output") o.out = val();
        else if (a == "-s" || a == "--sample") o.sample = val();
        else if (a == "--targets") o.targets = val();
        else if (a == "-R" || a == "--regions-file") o.bed = val();
        else if (a == "-t" || a == "--threads") o.threads = std::max(1, atoi(val().c_str()));
        else if (a == "-A" || a == "--all-out") o.all_out = true;
        else if (a == "-q" || a == "--vqual") o.vqual = atof(val().c_str());
        else if (a == "--outvar-flag") o.outvar_flag = atoi(val().c_str());
        else if (a == "--tile") o.tile = std::max<int64_t>(100, atoll(val().c_str()));
        else if (a == "--mem-per-thread") o.mem_per_thread = std::max<int64_t>(1, atoll(val().c_str()));
        else if (a == "--device" || a == "--devices") {   // comma-separated HIP device ids; an id may repeat (two workers sets on one GPU)
            o.devices.clear();
            const std::string v = val(); size_t at = 0;
            while (at <= v.size()) { size_t c = v.find(',', at); if (c == std::string::npos) c = v.size(); if (c > at) o.devices.push_back(atoi(v.substr(at, c - at).c_str())); at = c + 1; }
            if (o.devices.empty()) die("--devices needs at least one id");
        }
        else if (a == "--shard") { const std::string v = val(); if (sscanf(v.c_str(), "%d/%d", &o.shard, &o.n_shards) != 2 || o.n_shards < 1 || o.shard < 0 || o.shard >= o.n_shards) die("--shard takes i/n with 0 <= i < n"); }
        else if (a == "--no-header") o.no_header = true;
        else if (a == "--timing") o.timing = true;
        else if (a == "--device-inflate") o.device_inflate = true;   // the BGZF blocks of the BAM inflated by the GPU (uvcgpu_bgzf_inflate) instead of the host cores
        else if (a == "--tumor-vcf") o.tumor_vcf = val();
        else if (a == "--tn-is-paired") o.tn_is_paired = atoi(val().c_str());
        else if (a == "--is-tumor-format-retrieved") o.tumor_format = atoi(val().c_str());
        else if (a == "--bed-out-fname") o.bed_out = val();
        else if (a == "--bed-in-fname") o.bed_in = val();
        else if (a == "--repeat") o.repeat = std::max(1, atoi(val().c_str()));   // benchmark aid: the tile list n times (steady state on a small file)
        else if (a == "-h" || a == "--help") { usage(); exit(0); }
        else if (!a.empty() && a[0] == '-') die("unknown option " + a + " (the hot-path parameters keep the reference's defaults)");
        else if (o.bam.empty()) o.bam = a;
        else die("more than one inputBAM");
    }
    if (o.bam.empty() || o.fasta.empty() || o.out.empty()) { usage(); exit(2); }
    if (const char *us = getenv("ONE_STEP_UMI_STRUCT")) o.umi_struct = us;   // the reference takes the in-read UMI pattern from the environment (main.cpp:1224-1225)
    return o;
}

// A tile of a run of adjacent tiles.  The reference scores zerobased_pos rpos_beg .. rpos_end inclusive without the BASE sub-position of
// the first (main.cpp:608, 643): adjacent regions both write the LINK records of their shared end point.  Here a tile owns the positions
// [beg, end): `continues` (a tile ends where this one begins) = it scores `beg` completely, `has_next` = it leaves `end` to the next one.
// `run_beg` = begin of the run (incluBegPosition of the BED line the run came from, main.cpp:655-656).
struct Tile { int32_t tid; std::string chrom; int64_t beg, end; bool continues, has_next; int64_t run_beg; };

// one worker: its own handles, one region handle for all of its tiles
struct Worker {
    uvcio_bam_t *bam = nullptr; uvcio_fasta_t *fa = nullptr; uvcgpu_region_t *reg = nullptr;
    std::vector<uint64_t> h31, h17, u31, u17; std::vector<uint8_t> kind;
    std::vector<int32_t> filt, isz, order, fam, frag; std::vector<uint8_t> fstrand, dflag, idflag;
    std::vector<int32_t> pos, mpos, isize, nm, lq, ncig, fragp, famp; std::vector<uint16_t> flag; std::vector<uint8_t> mapq, strandp; std::vector<int64_t> soff, coff;
    std::vector<int32_t> fields; std::string ref;
    double t_fetch = 0, t_group = 0, t_region = 0, t_reads = 0, t_gpu = 0, t_text = 0;
    int64_t n_tiles = 0, score_cap = 0, text_cap = 0;   // what the last tiles needed: the next call asks for it at once
};
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// process_batch for one tile; appends the record lines to `lines`; false = nothing to call there.  *n_kept_reads: reads that passed the filters.
bool call_tile(Worker &w, const Opts &o, const UvcParams &P, const Tile &t, int64_t tlen, const uvcio_tumor_vcf_t *tvcf, std::string &lines, int64_t *n_kept_reads) {
    double t0 = now();
    *n_kept_reads = 0;
    UvcBamBatch b;
    if (uvcio_bam_fetch(w.bam, t.tid, std::max<int64_t>(0, t.beg - MAX_INSERT_SIZE), t.end + MAX_INSERT_SIZE, &b)) die(uvcio_last_error());
    w.t_fetch += now() - t0; t0 = now();
    const int64_t n = b.n_alns;
    if (n == 0) return false;
    w.h31.resize(n); w.h17.resize(n); w.u31.resize(n); w.u17.resize(n); w.kind.resize(n);
    uvcgpu_qname_digest_batch(b.qnames, b.qname_off, n, 0, 0, w.h31.data(), w.h17.data(), w.u31.data(), w.u17.data(), w.kind.data());
    if (!o.umi_struct.empty() && uvcgpu_umi_in_read_batch(o.umi_struct.c_str(), b.bases, b.seq_off, b.l_qseq, b.flag, n, w.kind.data(), nullptr)) die(uvcgpu_last_error());   // grouping.cpp:787-792
    UvcGroupParams gp; uvcgpu_group_params_default(&gp);
    gp.fetch_tbeg = (int32_t)t.beg; gp.fetch_tend = (int32_t)t.end; gp.inferred_sequencing_platform = P.inferred_sequencing_platform;
    UvcGroupInput gi; memset(&gi, 0, sizeof(gi));
    gi.n_alns = n; gi.tid = b.tid; gi.pos = b.pos; gi.endpos = b.endpos; gi.mtid = b.mtid; gi.mpos = b.mpos; gi.isize = b.isize; gi.flag = b.flag; gi.mapq = b.mapq;
    gi.qname_hash31 = w.h31.data(); gi.qname_hash17 = w.h17.data(); gi.umi_hash31 = w.u31.data(); gi.umi_hash17 = w.u17.data(); gi.umi_kind = w.kind.data();
    w.filt.resize(n); w.isz.resize(n); w.order.resize(n); w.fam.resize(n); w.frag.resize(n); w.fstrand.resize(n); w.dflag.resize(n); w.idflag.resize(n);
    UvcGroupOut go; memset(&go, 0, sizeof(go));
    go.filter_reason = w.filt.data(); go.isize_norm = w.isz.data(); go.order = w.order.data(); go.fam_id = w.fam.data(); go.frag_id = w.frag.data();
    go.fam_strand = w.fstrand.data(); go.fam_dflag = w.dflag.data(); go.fam_idflag = w.idflag.data();
    if (uvcgpu_group_families(&gp, &gi, &go)) die(uvcgpu_last_error());
    w.t_group += now() - t0; t0 = now();
    const int64_t k = go.n_kept;
    *n_kept_reads = k;
    if (k == 0) return false;
    // region bounds and reference, main.cpp:523-552
    const int64_t bam_beg = go.extended_inclu_beg_pos, bam_end = go.extended_exclu_end_pos;
    const int64_t rpos_beg = std::max(t.beg, bam_beg), rpos_end = std::min(t.end, bam_end);
    const int64_t ext_beg = std::max<int64_t>(0, std::min(t.beg, bam_beg) - MAX_STR_N_BASES), ext_end = std::min(tlen, std::max(t.end, bam_end) + MAX_STR_N_BASES);
    const int64_t first = rpos_beg;
    const int64_t last_excl = t.has_next ? std::min(t.end, bam_end + 1) : std::min(rpos_end + 1, ext_end);   // zerobased_pos t.end belongs to the next tile
    if (last_excl <= first) return false;
    w.ref.resize((size_t)(ext_end - ext_beg));
    if (uvcio_fasta_fetch(w.fa, t.chrom.c_str(), ext_beg, ext_end, &w.ref[0])) die(uvcio_last_error());
    int rc = w.reg ? uvcgpu_region_reset(w.reg, t.tid, (int32_t)ext_beg, (int32_t)ext_end, w.ref.c_str())
                   : uvcgpu_region_create(&w.reg, &P, t.tid, (int32_t)ext_beg, (int32_t)ext_end, w.ref.c_str());
    if (rc) die(uvcgpu_last_error());
    w.t_region += now() - t0; t0 = now();
    // the kept alignments in alns3 order
    auto gather = [&](auto &dst, const auto *src) { dst.resize((size_t)k); for (int64_t i = 0; i < k; i++) dst[(size_t)i] = src[w.order[(size_t)i]]; };
    gather(w.pos, b.pos); gather(w.mpos, b.mpos); gather(w.isize, w.isz.data()); gather(w.flag, b.flag); gather(w.mapq, b.mapq); gather(w.nm, b.nm);
    gather(w.lq, b.l_qseq); gather(w.soff, b.seq_off); gather(w.coff, b.cigar_off); gather(w.ncig, b.n_cigar);
    UvcReadSoA rs; memset(&rs, 0, sizeof(rs)); rs.struct_size = (int32_t)sizeof(rs);
    rs.n_reads = k; rs.pos = w.pos.data(); rs.mpos = w.mpos.data(); rs.isize = w.isize.data(); rs.flag = w.flag.data(); rs.mapq = w.mapq.data(); rs.nm = w.nm.data();
    rs.l_qseq = w.lq.data(); rs.seq_off = w.soff.data(); rs.cigar_off = w.coff.data(); rs.n_cigar = w.ncig.data();
    rs.frag_id = w.frag.data(); rs.fam_id = w.fam.data(); rs.fam_strand = w.fstrand.data();
    rs.n_bases = b.n_bases; rs.bases = b.bases; rs.quals = b.quals; rs.n_cigar_ops = b.n_cigar_ops; rs.cigars = b.cigars;
    rs.n_fams = go.n_fams; rs.fam_dflag = w.dflag.data();
    if (uvcgpu_region_set_reads(w.reg, &rs)) die(uvcgpu_last_error());
    w.t_reads += now() - t0; t0 = now();
    if (uvcgpu_region_correct_bq(w.reg) || uvcgpu_region_accumulate(w.reg)) die(uvcgpu_last_error());
    UvcScoreRequest rq; memset(&rq, 0, sizeof(rq));
    rq.pos_beg = (int32_t)first; rq.pos_end = (int32_t)last_excl; rq.all_out = o.all_out; rq.is_amplicon = (go.n_amplicon * 2 > k);
    rq.base_at_pos_beg = (t.continues && first == t.beg && t.beg > ext_beg) ? 1 : 0; rq.region_beg = (int32_t)t.run_beg;
    if (tvcf) {   // normal sample of a T/N pair: the tumor records of this region (tkis_beg .. tkis_end, main.cpp:532-533)
        const UvcTumorKey *keys = nullptr; const char *const *cols = nullptr, *const *ras = nullptr; int64_t nk = 0;
        if (uvcio_tumor_vcf_fetch(tvcf, t.tid, (int32_t)ext_beg, (int32_t)ext_end, &keys, &cols, &ras, &nk)) die(uvcio_last_error());
        rq.tumor_keys = keys; rq.n_tumor_keys = nk; rq.tumor_sample_columns = (o.tumor_format ? cols : nullptr); rq.tumor_ref_alt = ras;
    }
    rq.kept_only = 1;   // only the record groups that are written travel to the host
    int64_t cap = std::max<int64_t>(std::max<int64_t>(4096, w.score_cap), uvcgpu_region_score_size(w.reg, &rq) / (o.all_out ? 1 : 64));
    UvcScoreOut so;
    for (;;) {
        w.fields.resize((size_t)UVC_NUM_SCORE_FIELDS * (size_t)cap);
        so.capacity = cap; so.n_records = 0; so.fields = w.fields.data();
        rc = uvcgpu_region_score(w.reg, &rq, &so);
        if (rc == UVCGPU_ENOMEM && so.n_records > cap) { cap = so.n_records + so.n_records / 4; continue; }
        if (rc) die(uvcgpu_last_error());
        break;
    }
    w.score_cap = cap;
    w.t_gpu += now() - t0; t0 = now();
    // one call with a buffer as large as the last tile needed (+ slack); a second one only when the text did not fit
    const size_t at = lines.size();
    int64_t len = 0, room = std::max<int64_t>(1 << 16, w.text_cap);
    for (;;) {
        lines.resize(at + (size_t)room);
        rc = uvcgpu_region_vcf_records(w.reg, t.chrom.c_str(), &so, &rq, &lines[at], room, &len);
        if (rc == UVCGPU_ENOMEM && len > room) { room = len + len / 4; continue; }
        if (rc) die(uvcgpu_last_error());
        break;
    }
    lines.resize(at + (size_t)len);
    w.text_cap = std::max<int64_t>(w.text_cap, len + len / 4);
    w.t_text += now() - t0;
    return true;
}
}   // namespace

int main(int argc, char **argv) {
    if (argc >= 3 && !strcmp(argv[1], "--concat")) {   // bcftools concat -n (uvcTN.sh:100)
        std::vector<const char *> in; for (int i = 3; i < argc; i++) in.push_back(argv[i]);
        if (uvcio_bgzf_concat(argv[2], in.data(), (int32_t)in.size())) die(uvcio_last_error());
        return 0;
    }
    Opts o = parse(argc, argv);
    // Each region handle has three streams and a worker's copies should run under another worker's kernels: with the runtime's default of four
    // hardware queues the streams of different handles share queues, and a kernel then waits behind another handle's 10 ms copy (bench.py's
    // pcie_inclusive leg: 16.0 ms per tile with 4 queues, 13.5 with 16).  Read by the HIP runtime when it starts; an explicit setting wins.
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    if (o.devices.empty()) { const int nd = uvcgpu_device_count(); if (nd <= 0) die("no HIP device: uvc1-mi355x has no CPU path"); for (int d = 0; d < nd; d++) o.devices.push_back(d); }
    // tiles in flight: the host stages of a tile (inflate above all) cost ~1.3 core-seconds per 1 Mb x 300x, the device ~10 ms: as many workers
    // as half the cores this process may use keep the cores busy (measured on a 16-core quota: 4 -> 6.7, 8 -> 8-11, 12-16 -> 8-9.6 M positions/s),
    // never more than 8 per device (a region handle holds ~7 GB of planes) and at least one per device
    if (o.threads <= 0) { const int nd = (int)o.devices.size(); o.threads = std::max(nd, std::min(8 * nd, uvc_effective_cpus() / 2)); }
    // the readers of all tiles in flight share one pool of inflate / decode threads inside libuvcio (as many as this process has cores:
    // quota- and affinity-aware, uvc_cpus.h); UVCIO_THREADS overrides
    if (uvcgpu_init(o.devices[0])) die(uvcgpu_last_error());
    uvcio_bam_t *bam0 = nullptr;
    if (uvcio_bam_open(&bam0, o.bam.c_str())) die(uvcio_last_error());
    if (!uvcio_bam_has_index(bam0)) fprintf(stderr, "uvc1-mi355x: no .bai next to %s, every tile scans the file\n", o.bam.c_str());
    const int32_t nref = uvcio_bam_n_refs(bam0);
    std::vector<std::string> names; std::vector<int64_t> lens; std::vector<const char *> cnames;
    for (int32_t i = 0; i < nref; i++) { names.push_back(uvcio_bam_ref_name(bam0, i)); lens.push_back(uvcio_bam_ref_len(bam0, i)); }
    for (auto &s : names) cnames.push_back(s.c_str());
    // the tiles: --bed-in-fname / -R regions, --targets "chr" or "chr:beg-end" (1-based inclusive as in samtools), else every contig
    std::vector<Tile> tiles;
    const std::string bed_path = (!o.bed_in.empty() ? o.bed_in : o.bed);
    // Without --tile and without a BED file the regions are the ones the reference itself would hand to process_batch: one pass over the
    // alignments of the targets (as SamIter::iternext makes it, grouping.cpp:225-312) through uvcio_plan_regions -- cuts at contig changes, at
    // gaps of more than 200 bp and where the per-thread memory model says so (grouping.cpp:28-67: ~20-40 kb at 300x).  Every such region is
    // then processed exactly like a process_batch call (its own reference window, repeat tracks and BAQ sums from its own start, zerobased_pos
    // beg .. end inclusive), so the records equal the reference's also next to a cut.  --tile N trades that for long regions (faster on the
    // device; qualities within one unit next to a cut).
    const bool ref_cuts = (o.tile <= 0 && bed_path.empty());
    if (o.tile <= 0) o.tile = 1000000;
    // the planning pass is a stream: every 4 Mb window of alignments goes straight into the planner, nothing per alignment is kept
    uvcio_planner_t *planner = nullptr; int64_t n_planned = 0; const double tp = now();
    if (ref_cuts && uvcio_planner_open(&planner, lens.data(), nref, (o.threads > 0 ? o.threads : 8) /* the reference's -t default (CmdLineArgs.hpp:34): enters only where a batch of regions ends */, o.mem_per_thread)) die(uvcio_last_error());
    std::vector<int32_t> pl_tid, pl_pos, pl_end; std::vector<uint16_t> pl_flag;   // one window's columns
    auto take_cuts = [&]() { UvcRegionCut c[256]; int64_t k; while ((k = uvcio_planner_take(planner, c, 256)) > 0) for (int64_t q = 0; q < k; q++) tiles.push_back(Tile{ c[q].tid, names[(size_t)c[q].tid], c[q].beg, c[q].end, false, false, c[q].beg }); };
    auto add = [&](int32_t tid, int64_t beg, int64_t end) {
        if (!ref_cuts) { for (int64_t b = beg; b < end; b += o.tile) tiles.push_back(Tile{ tid, names[(size_t)tid], b, std::min(b + o.tile, end), false, false, b }); return; }
        const int64_t W = 4000000;   // the planning pass reads the span window by window; an alignment is taken by the window it starts in (the first window also takes those that reach into it)
        for (int64_t wb = beg; wb < end; wb += W) {
            UvcBamBatch b;
            if (uvcio_bam_fetch(bam0, tid, wb, std::min(wb + W, end), &b)) die(uvcio_last_error());
            pl_tid.clear(); pl_pos.clear(); pl_end.clear(); pl_flag.clear();
            for (int64_t i = 0; i < b.n_alns; i++) if (b.pos[i] >= wb || wb == beg) { pl_tid.push_back(b.tid[i]); pl_pos.push_back(b.pos[i]); pl_end.push_back(b.endpos[i]); pl_flag.push_back(b.flag[i]); }
            if (uvcio_planner_feed(planner, pl_tid.data(), pl_pos.data(), pl_end.data(), pl_flag.data(), (int64_t)pl_tid.size())) die(uvcio_last_error());
            n_planned += (int64_t)pl_tid.size();
            take_cuts();
        }
    };
    if (!bed_path.empty()) {   // one region per BED line (0-based, half-open), cut into tiles; overrides --targets as in the reference
        FILE *fb = fopen(bed_path.c_str(), "r");
        if (!fb) die("cannot open " + bed_path);
        char line[4096], chrom[1024]; long long b = 0, e = 0;
        while (fgets(line, sizeof(line), fb)) {
            if (line[0] == '#' || !strncmp(line, "track", 5) || !strncmp(line, "browser", 7)) continue;
            if (sscanf(line, "%1023s %lld %lld", chrom, &b, &e) != 3) continue;
            int32_t tid = -1;
            for (int32_t i = 0; i < nref; i++) if (names[(size_t)i] == chrom) tid = i;
            if (tid < 0) die(std::string("the BED file names a contig that is not in the BAM header: ") + chrom);
            add(tid, std::max<long long>(0, b), std::min<long long>(e, lens[(size_t)tid]));
        }
        fclose(fb);
    } else if (!o.targets.empty()) {
        std::string chrom = o.targets; int64_t beg = 0, end = -1;
        const size_t c = o.targets.rfind(':');
        if (c != std::string::npos && o.targets.find('-', c) != std::string::npos) {
            chrom = o.targets.substr(0, c);
            std::string rng = o.targets.substr(c + 1); rng.erase(std::remove(rng.begin(), rng.end(), ','), rng.end());
            beg = std::max<int64_t>(0, atoll(rng.c_str()) - 1); end = atoll(rng.substr(rng.find('-') + 1).c_str());
        }
        int32_t tid = -1;
        for (int32_t i = 0; i < nref; i++) if (names[(size_t)i] == chrom) tid = i;
        if (tid < 0) die("--targets names a contig that is not in the BAM header: " + chrom);
        add(tid, beg, end < 0 ? lens[(size_t)tid] : std::min(end, lens[(size_t)tid]));
    } else for (int32_t i = 0; i < nref; i++) add(i, 0, lens[(size_t)i]);
    if (ref_cuts) {
        if (uvcio_planner_finish(planner)) die(uvcio_last_error());
        take_cuts();
        uvcio_planner_close(planner); planner = nullptr;
        fprintf(stderr, "uvc1-mi355x: %zu regions from the reference's cuts over %lld alignments (planning pass %.2f s)\n", tiles.size(), (long long)n_planned, now() - tp);
        std::vector<int32_t>().swap(pl_tid); std::vector<int32_t>().swap(pl_pos); std::vector<int32_t>().swap(pl_end); std::vector<uint16_t>().swap(pl_flag);
    }
    // ownership of the shared end points: a tile whose predecessor ends where it begins continues that one's run (fixed tiles only: the
    // reference's own regions each write both end points, main.cpp:608, 643)
    if (!ref_cuts)
    for (size_t q = 1; q < tiles.size(); q++) if (tiles[q].tid == tiles[q - 1].tid && tiles[q].beg == tiles[q - 1].end) { tiles[q].continues = true; tiles[q - 1].has_next = true; tiles[q].run_beg = tiles[q - 1].run_beg; }
    // --shard i/n: the i-th of n contiguous runs of the list, balanced by index bytes + positions
    if (o.n_shards > 1) {
        std::vector<int64_t> cost(tiles.size()); std::vector<int32_t> shard_of(tiles.size());
        for (size_t q = 0; q < tiles.size(); q++) cost[q] = uvcio_bam_region_bytes(bam0, tiles[q].tid, tiles[q].beg, tiles[q].end) + (tiles[q].end - tiles[q].beg) / 8 + 1;
        if (uvcio_plan_shards(cost.data(), (int64_t)cost.size(), o.n_shards, shard_of.data())) die(uvcio_last_error());
        std::vector<Tile> mine;
        for (size_t q = 0; q < tiles.size(); q++) if (shard_of[q] == o.shard) mine.push_back(tiles[q]);
        fprintf(stderr, "uvc1-mi355x: shard %d of %d takes %zu of %zu tiles\n", o.shard, o.n_shards, mine.size(), tiles.size());
        tiles.swap(mine);
        if (o.shard > 0) o.no_header = true;
    }
    const size_t tiles_per_pass = tiles.size();
    for (int rep = 1; rep < o.repeat; rep++) for (size_t q = 0; q < tiles_per_pass; q++) tiles.push_back(tiles[q]);

    // parameters: the reference's defaults; platform and read length inferred from the first alignments that are seen (CmdLineArgs.cpp:34-111
    // reads the first 5000 records of the file; here: of the first tile that has any)
    UvcParams P; uvcgpu_params_default(&P);
    if (o.vqual > -1e8) P.vqual = o.vqual;
    if (o.outvar_flag >= 0) P.outvar_flag = o.outvar_flag;
    P.should_output_all = o.all_out;
    P.tn_is_paired = o.tn_is_paired;
    {
        int platform = UVC_PLATFORM_ILLUMINA, readlen = 150, maxmq = 0; bool seen = false;
        for (size_t ti = 0; ti < tiles.size() && !seen; ti++) {
            UvcBamBatch b;
            if (uvcio_bam_fetch(bam0, tiles[ti].tid, tiles[ti].beg, tiles[ti].end, &b)) die(uvcio_last_error());
            if (b.n_alns == 0) continue;
            seen = true;
            const int64_t m = std::min<int64_t>(b.n_alns, 5000);
            std::vector<int32_t> ql{ 150 }; uint64_t pe = 0, q20f = 0, q30f = 0, q30p = 0;
            for (int64_t i = 0; i < m; i++) {
                maxmq = std::max<int>(maxmq, b.mapq[i]); pe += (b.flag[i] & 1); ql.push_back(b.l_qseq[i]);
                for (int32_t q = 0; q < b.l_qseq[i]; q++) { const uint8_t bq = b.quals[b.seq_off[i] + q]; if (bq < 30) q30f++; else q30p++; if (bq < 20) q20f++; }
            }
            std::sort(ql.begin(), ql.end());
            readlen = ql[ql.size() / 2];
            const bool fix = ((int64_t)ql[ql.size() / 2] * 100 > (int64_t)ql.back() * 95);
            if (!(pe > 0 || 4 * (q30f - q20f) < q30p || (2 * (q30f - q20f) < q30p && fix))) platform = UVC_PLATFORM_IONTORRENT;
        }
        uvcgpu_params_apply_platform(&P, platform, readlen, maxmq);
    }
    uvcio_bam_close(bam0);
    // UVC1_PINNED=1: the workers' base / quality columns live in page-locked memory of the GPU library from here on, so that set_reads copies
    // them by DMA.  Off by default: on the boxes measured the files -> VCF rate did not move with it (scripts/bench_cli.py; the chain is not
    // bound by that copy) and it locks ~ 800 MB of host memory per worker.
    // --device-inflate (or UVC1_DEVICE_INFLATE=1): the inflate is ~half of the host's work per tile and the host's cores bound files -> VCF
    // (DESIGN.md 6b).  One device call per batch of blocks costs about as much for 500 blocks as for 8 000 (every block is a wave of its own):
    // the reader takes the compressed bytes of a whole tile as one batch.
    if (o.device_inflate || getenv("UVC1_DEVICE_INFLATE")) {
        const char *mb = getenv("UVC1_DEVICE_INFLATE_MIN");   // batches with fewer blocks stay on the host (tests: 1)
        uvcio_set_inflate(uvcgpu_bgzf_inflate, nullptr, mb ? atoi(mb) : 256);
        setenv("UVCIO_BATCH_BYTES", "402653184", 0);
    }
    if (getenv("UVC1_PINNED"))
        uvcio_set_column_allocator([](size_t n) -> void * { void *q = nullptr; return uvcgpu_host_alloc(&q, (int64_t)n) == 0 ? q : nullptr; }, [](void *q) { (void)uvcgpu_host_free(q); });
    // T/N: the tumor pass's records (rescue_variants_from_vcf, main.cpp:183-398)
    uvcio_tumor_vcf_t *tvcf = nullptr;
    if (!o.tumor_vcf.empty()) {
        if (uvcio_tumor_vcf_open(&tvcf, o.tumor_vcf.c_str(), cnames.data(), nref, o.tumor_format)) die(uvcio_last_error());
        P.tumor_vcf_is_provided = 1;
        fprintf(stderr, "uvc1-mi355x: %lld tumor records from %s\n", (long long)uvcio_tumor_vcf_n_records(tvcf), o.tumor_vcf.c_str());
    }

    // output: header, then the lines of every tile in tile order
    uvcio_bgzf_writer_t *zw = nullptr;
    if (uvcio_bgzf_write_open(&zw, o.out.c_str(), 6)) die(uvcio_last_error());
    if (!o.no_header) {
        int64_t len = 0;
        const char *tsample = (tvcf && o.tumor_format) ? uvcio_tumor_vcf_sample_name(tvcf) : nullptr;
        // ##fileDate / ##reference / ##variantCallerCommand as generate_vcf_header prints them (main.hpp:5788-5794, 5870-5874)
        char date[80]; { time_t raw; time(&raw); strftime(date, sizeof(date), "%F %T", localtime(&raw)); }
        std::string cmd; for (int i = 0; i < argc; i++) { cmd += argv[i]; cmd += "  "; }
        uvcgpu_vcf_header_ex(&P, o.sample.c_str(), tsample, cnames.data(), lens.data(), nref, date, o.fasta.c_str(), cmd.c_str(), nullptr, 0, &len);
        std::string h((size_t)len, '\0');
        if (uvcgpu_vcf_header_ex(&P, o.sample.c_str(), tsample, cnames.data(), lens.data(), nref, date, o.fasta.c_str(), cmd.c_str(), &h[0], len, &len)) die(uvcgpu_last_error());
        if (uvcio_bgzf_write(zw, h.data(), (int64_t)h.size())) die(uvcio_last_error());
    }
    const double t_start = now();
    std::vector<std::string> done(tiles.size()); std::vector<char> ready(tiles.size(), 0); std::vector<int64_t> tile_reads(tiles.size(), 0);
    std::mutex mu; std::condition_variable cv; std::atomic<size_t> next{ 0 };
    size_t written = 0;   // tiles the writer has taken (guarded by mu)
    const int nthreads = (int)std::min<size_t>((size_t)o.threads, std::max<size_t>(tiles.size(), 1));
    const size_t max_ahead = (size_t)4 * (size_t)nthreads;
    std::vector<Worker> workers((size_t)nthreads);
    std::vector<std::thread> th;
    for (int wi = 0; wi < nthreads; wi++) th.emplace_back([&, wi]() {
        Worker &w = workers[(size_t)wi];
        if (uvcgpu_init(o.devices[(size_t)wi % o.devices.size()])) die(uvcgpu_last_error());   // binds this host thread to its device
        if (uvcio_bam_open(&w.bam, o.bam.c_str()) || uvcio_fasta_open(&w.fa, o.fasta.c_str())) die(uvcio_last_error());
        for (;;) {
            const size_t ti = next.fetch_add(1);
            if (ti >= tiles.size()) break;
            // bounded run-ahead: finished tiles wait in `done` for the in-order writer; a worker does not start a tile more than
            // 4 * threads in front of it, so a slow early tile cannot make the rest of the genome pile up in memory
            { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return ti < written + max_ahead; }); }
            std::string lines; int64_t nk = 0;
            call_tile(w, o, P, tiles[ti], lens[(size_t)tiles[ti].tid], tvcf, lines, &nk);
            w.n_tiles++;
            { std::lock_guard<std::mutex> g(mu); done[ti].swap(lines); ready[ti] = 1; tile_reads[ti] = nk; }
            cv.notify_all();
        }
        if (w.reg) uvcgpu_region_destroy(w.reg);
        uvcio_bam_close(w.bam); uvcio_fasta_close(w.fa);
    });
    int64_t n_lines = 0, n_pos = 0; double t_first_pass = 0; int64_t pos_first_pass = 0;
    for (size_t ti = 0; ti < tiles.size(); ti++) {
        if (ti == tiles_per_pass) { t_first_pass = now() - t_start; pos_first_pass = n_pos; }
        std::string lines;
        { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return ready[ti] != 0; }); lines.swap(done[ti]); written = ti + 1; }
        cv.notify_all();
        n_lines += std::count(lines.begin(), lines.end(), '\n'); n_pos += tiles[ti].end - tiles[ti].beg;
        if (!lines.empty() && uvcio_bgzf_write(zw, lines.data(), (int64_t)lines.size())) die(uvcio_last_error());
    }
    for (auto &t : th) t.join();
    if (uvcio_bgzf_write_close(zw)) die(uvcio_last_error());
    if (tvcf) uvcio_tumor_vcf_close(tvcf);
    if (!o.bed_out.empty()) {   // the region table of main.cpp:1415-1436 (--bed-out-fname): the shard manifest of the normal pass of a T/N pair
        FILE *fo = fopen(o.bed_out.c_str(), "w");
        if (!fo) die("cannot create " + o.bed_out);
        for (size_t ti = 0; ti < tiles_per_pass; ti++)
            fprintf(fo, "%s\t%lld\t%lld\tBedLineFlag\t%d\tNumberOfReadsInThisInterval\t%lld\tNumberOfRefBasesInThisInterval\t%lld\tTier1regionIndex\t0\tTier2regionIndex\t%d\tTier3regionIndex\t%zu\n",
                    tiles[ti].chrom.c_str(), (long long)tiles[ti].beg, (long long)tiles[ti].end, tiles[ti].continues ? 4 : 16, (long long)tile_reads[ti], (long long)(tiles[ti].end - tiles[ti].beg), o.shard, ti);
        fclose(fo);
    }
    const double dt = now() - t_start;
    fprintf(stderr, "uvc1-mi355x: %lld record lines from %zu tiles (%lld positions) in %.2f s = %.2f M positions/s, %d tiles in flight on %zu device(s)\n",
            (long long)n_lines, tiles.size(), (long long)n_pos, dt, n_pos / dt / 1e6, nthreads, o.devices.size());
    if (o.repeat > 1) fprintf(stderr, "  passes 2..%d (steady state): %.2f M positions/s\n", o.repeat, (n_pos - pos_first_pass) / (dt - t_first_pass) / 1e6);
    if (o.timing) {
        double f = 0, g = 0, r = 0, s = 0, k = 0, x = 0;
        for (auto &w : workers) { f += w.t_fetch; g += w.t_group; r += w.t_region; s += w.t_reads; k += w.t_gpu; x += w.t_text; }
        fprintf(stderr, "  thread-seconds: fetch %.2f, digest+group %.2f, reference+region %.2f, set_reads %.2f, bq+accumulate+score %.2f, record text %.2f\n", f, g, r, s, k, x);
        for (size_t wi = 0; wi < workers.size(); wi++) fprintf(stderr, "  worker %zu on device %d: %lld tiles\n", wi, o.devices[wi % o.devices.size()], (long long)workers[wi].n_tiles);
    }
    return 0;
}
