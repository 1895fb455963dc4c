// uvc_conblock.cpp -- insertion / soft-clip consensus blocks (include/uvcconsensus.h), host C++.
//
// The reference keeps one std::map<position, vector<BaseToCount>> per block type inside every temporary Symbol2CountCoverage of P4
// (main_consensus.hpp:116-225) and folds fragment maps into the family's map.  Here the inserted / clipped stretches of all reads become
// one flat event list that is sorted once by (family, strand, type, position, fragment): a fragment's block is the run of its events
// (per base the maximum quality per base symbol), a family's block is the fold of its fragments' runs.  No per-object maps.
#include "uvcconsensus.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

extern "C" int uvcgpu_set_error(int code, const char *msg);   // uvc_host.cpp

namespace {
struct Event {
    int32_t fam, strand, type, refpos, frag_rank;   // frag_rank: fragments of a unit numbered in read order
    int32_t unit;                                     // index of the (family, strand) unit
    int32_t len; bool reversed;
    int64_t first_base;                               // index into bases[] / quals[] of the op's first query base
};
struct UnitInfo { int32_t fam, strand, n_fragments, beg, end; };

inline bool consumes_query(int op) { return op == 0 || op == 1 || op == 4 || op == 7 || op == 8; }   // M I S = X
inline bool consumes_ref(int op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }     // M D N = X

// events of read i (the INS and soft-clip arms of updateByAln, main.hpp:2009, 2100-2116, 2259-2279)
void read_events(const UvcParams &P, const UvcReadSoA &R, int64_t i, int32_t frag_rank, int32_t unit, std::vector<Event> &ev) {
    const uint32_t *cig = R.cigars + R.cigar_off[i];
    const int n_cigar = R.n_cigar[i];
    int32_t rend = R.pos[i];
    for (int c = 0; c < n_cigar; c++) if (consumes_ref(cig[c] & 0xF)) rend += (int32_t)(cig[c] >> 4);
    const int flag = R.flag[i];
    const bool isrc = (flag & 0x10) != 0;
    const uint8_t dflag = (R.fam_dflag ? R.fam_dflag[R.fam_id[i]] : 0);
    const bool is_assay_amplicon = ((dflag & 0x4) || ((P.primerlen > 0) && !(0x2 & P.primer_flag)));
    const bool normal_filters_primers = (P.tn_is_paired && (0x1 & P.primer_flag));
    // the interval between the two primers of the insert (main.hpp:1872-1875)
    const int64_t lo = std::min(R.pos[i], R.mpos[i]);
    const bool single_rc = (isrc && !(flag & 0x1));
    const int64_t ibeg = (R.isize[i] != 0 ? lo + P.primerlen : (single_rc ? 0 : (int64_t)R.pos[i] + P.primerlen));
    const int64_t iend = (R.isize[i] != 0 ? std::max<int64_t>(lo + std::abs(R.isize[i]) - P.primerlen, 0)
                                          : (single_rc ? std::max<int64_t>((int64_t)rend - P.primerlen, 0) : (int64_t)INT32_MAX));
    int32_t rpos = R.pos[i]; int64_t qpos = 0;
    for (int c = 0; c < n_cigar; c++) {
        const int op = (int)(cig[c] & 0xF); const int32_t len = (int32_t)(cig[c] >> 4);
        if (op == 1 /* I */) {
            if ((normal_filters_primers || !is_assay_amplicon) || (ibeg <= rpos && rpos < iend))
                ev.push_back(Event{ R.fam_id[i], (int32_t)R.fam_strand[i], UVC_CONBLOCK_INS, rpos, frag_rank, unit, len, false, R.seq_off[i] + qpos });
        } else if (op == 4 /* S */) {
            const bool first_op = (c == 0);   // "fixed right, variable left": stored from the aligned end outwards
            ev.push_back(Event{ R.fam_id[i], (int32_t)R.fam_strand[i], first_op ? UVC_CONBLOCK_SOFTCLIP_RIGHT_TO_LEFT : UVC_CONBLOCK_SOFTCLIP_LEFT_TO_RIGHT, rpos, frag_rank, unit, len, first_op, R.seq_off[i] + qpos });
        }
        if (consumes_query(op)) qpos += len;
        if (consumes_ref(op)) rpos += len;
    }
}

// incByPosSeqQual of one event into a fragment-level block (rows grown by the caller)
void add_event(const UvcReadSoA &R, const Event &e, int32_t *rows /* [>= e.len][8] */) {
    for (int32_t k = 0; k < e.len; k++) {
        const int64_t q = e.first_base + (e.reversed ? (e.len - 1 - k) : k);
        const int b = std::min<int>(R.bases[q], 4);            // CHAR_TO_SYMBOL: everything that is not A C G T is BASE_N
        const int32_t qual = (int32_t)(int8_t)R.quals[q];      // the reference carries the quality as int8_t
        int32_t *row = rows + (size_t)k * UVC_CONBLOCK_ROW;
        row[b] = std::max(row[b], qual);
        row[UVC_CONBLOCK_BQ_SUM] = std::max(row[UVC_CONBLOCK_BQ_SUM], qual);
        row[UVC_CONBLOCK_NFRAGS] = 1;
    }
}
// incByMajorMinusMinor: one fragment-level block folded into the family's
void fold_fragment(const int32_t *frag_rows, int32_t len, int32_t *fam_rows) {
    for (int32_t k = 0; k < len; k++) {
        const int32_t *f = frag_rows + (size_t)k * UVC_CONBLOCK_ROW;
        int con = 5 /* BASE_NN */; int32_t concount = 0, tot = 0;
        for (int b = 0; b < 5; b++) { if (f[b] > concount) { con = b; concount = f[b]; } tot += f[b]; }
        int32_t *o = fam_rows + (size_t)k * UVC_CONBLOCK_ROW;
        o[con] += 1;
        o[UVC_CONBLOCK_BQ_SUM] += (concount * 2 > tot ? concount * 2 - tot : 0);
        o[UVC_CONBLOCK_NFRAGS] += 1;
    }
}
bool key_less(const Event &a, const Event &b) {
    if (a.fam != b.fam) return a.fam < b.fam;
    if (a.strand != b.strand) return a.strand < b.strand;
    if (a.type != b.type) return a.type < b.type;
    if (a.refpos != b.refpos) return a.refpos < b.refpos;
    return a.frag_rank < b.frag_rank;
}
bool same_block(const Event &a, const Event &b) { return a.fam == b.fam && a.strand == b.strand && a.type == b.type && a.refpos == b.refpos; }

int finish(std::vector<UvcConBlock> &B, std::vector<int32_t> &rows, UvcConBlock *blocks, int64_t block_capacity, int64_t *n_blocks, int32_t *out_rows, int64_t row_capacity, int64_t *n_rows) {
    if (n_blocks) *n_blocks = (int64_t)B.size();
    if (n_rows) *n_rows = (int64_t)rows.size();
    if ((int64_t)B.size() > block_capacity || (int64_t)rows.size() > row_capacity || (!blocks && !B.empty()) || (!out_rows && !rows.empty())) return uvcgpu_set_error(UVCGPU_ENOMEM, "consensus blocks: destination too small");
    if (!B.empty()) memcpy(blocks, B.data(), sizeof(UvcConBlock) * B.size());
    if (!rows.empty()) memcpy(out_rows, rows.data(), sizeof(int32_t) * rows.size());
    return 0;
}
int check_reads(const UvcReadSoA *R) {
    if (!R || R->n_reads < 0) return uvcgpu_set_error(UVCGPU_EINVAL, "consensus blocks: bad argument");
    if (R->struct_size != (int32_t)sizeof(UvcReadSoA)) return uvcgpu_set_error(UVCGPU_EINVAL, "UvcReadSoA::struct_size mismatch");
    if (R->n_reads > 0 && (!R->pos || !R->mpos || !R->isize || !R->flag || !R->seq_off || !R->cigar_off || !R->n_cigar || !R->frag_id || !R->fam_id || !R->fam_strand || !R->bases || !R->quals || !R->cigars))
        return uvcgpu_set_error(UVCGPU_EINVAL, "consensus blocks: a read column is NULL");
    return 0;
}
}   // namespace

extern "C" int uvcgpu_consensus_blocks_of_fragment(const UvcParams *P, const UvcReadSoA *R, int64_t first_read, int64_t n, UvcConBlock *blocks, int64_t block_capacity, int64_t *n_blocks,
                                                   int32_t *out_rows, int64_t row_capacity, int64_t *n_rows) {
    try {
        if (!P) return uvcgpu_set_error(UVCGPU_EINVAL, "consensus blocks: bad argument");
        { const int rc = check_reads(R); if (rc) return rc; }
        if (first_read < 0 || n < 0 || first_read + n > R->n_reads) return uvcgpu_set_error(UVCGPU_EINVAL, "consensus blocks: fragment outside the reads");
        std::vector<Event> ev;
        for (int64_t i = first_read; i < first_read + n; i++) read_events(*P, *R, i, 0, 0, ev);
        std::stable_sort(ev.begin(), ev.end(), key_less);
        std::vector<UvcConBlock> B; std::vector<int32_t> rows;
        for (size_t a = 0; a < ev.size();) {
            size_t b = a; int32_t len = 0;
            while (b < ev.size() && same_block(ev[a], ev[b])) { len = std::max(len, ev[b].len); b++; }
            const size_t off = rows.size();
            rows.resize(off + (size_t)len * UVC_CONBLOCK_ROW, 0);
            for (size_t k = a; k < b; k++) add_event(*R, ev[k], rows.data() + off);
            B.push_back(UvcConBlock{ ev[a].fam, ev[a].strand, ev[a].type, ev[a].refpos, len, 1, (int64_t)(off / UVC_CONBLOCK_ROW) });
            a = b;
        }
        return finish(B, rows, blocks, block_capacity, n_blocks, out_rows, row_capacity, n_rows);
    } catch (const std::bad_alloc &) { return uvcgpu_set_error(UVCGPU_ENOMEM, "consensus blocks: out of host memory"); }
}

extern "C" int uvcgpu_consensus_blocks(const UvcParams *P, const UvcReadSoA *R, const UvcConBlockRequest *req, UvcConBlock *blocks, int64_t block_capacity, int64_t *n_blocks,
                                       int32_t *out_rows, int64_t row_capacity, int64_t *n_rows) {
    try {
        if (!P || !req) return uvcgpu_set_error(UVCGPU_EINVAL, "consensus blocks: bad argument");
        { const int rc = check_reads(R); if (rc) return rc; }
        // units (family x strand) in read order: fragment count and the span of fillTidBegEndFromAlns2 (every alignment adds one to the end
        // it has reached, main.hpp:658-688)
        std::vector<UnitInfo> units;
        std::vector<int32_t> unit_of((size_t)R->n_reads), frag_rank((size_t)R->n_reads);
        for (int64_t i = 0; i < R->n_reads; i++) {
            const bool new_unit = (i == 0 || R->fam_id[i] != R->fam_id[i - 1] || R->fam_strand[i] != R->fam_strand[i - 1]);
            if (new_unit) units.push_back(UnitInfo{ R->fam_id[i], (int32_t)R->fam_strand[i], 0, INT32_MAX, 0 });
            UnitInfo &u = units.back();
            if (new_unit || R->frag_id[i] != R->frag_id[i - 1]) u.n_fragments++;
            int32_t rend = R->pos[i];
            const uint32_t *cig = R->cigars + R->cigar_off[i];
            for (int c = 0; c < R->n_cigar[i]; c++) if (consumes_ref(cig[c] & 0xF)) rend += (int32_t)(cig[c] >> 4);
            u.beg = std::min(u.beg, R->pos[i]); u.end = std::max(u.end, rend) + 1;
            unit_of[(size_t)i] = (int32_t)units.size() - 1; frag_rank[(size_t)i] = u.n_fragments - 1;
        }
        auto overlap = [](int64_t a0, int64_t a1, int64_t b0, int64_t b1) { return !(a1 <= b0 || b1 <= a0); };   // ARE_INTERVALS_OVERLAPPING, common.hpp:92
        std::vector<uint8_t> unit_done(units.size());
        for (size_t u = 0; u < units.size(); u++) {
            const bool applicable = (units[u].n_fragments >= req->min_fragments);
            const bool only_here = ((req->prev_tid != req->tid) || !overlap(req->prev_beg, req->prev_end, units[u].beg, units[u].end)) && overlap(req->curr_beg, req->curr_end, units[u].beg, units[u].end);
            unit_done[u] = (applicable && only_here) ? 1 : 0;
        }
        std::vector<Event> ev;
        for (int64_t i = 0; i < R->n_reads; i++) if (unit_done[(size_t)unit_of[(size_t)i]]) read_events(*P, *R, i, frag_rank[(size_t)i], unit_of[(size_t)i], ev);
        std::stable_sort(ev.begin(), ev.end(), key_less);
        std::vector<UvcConBlock> B; std::vector<int32_t> rows, frag_rows;
        for (size_t a = 0; a < ev.size();) {
            size_t b = a; int32_t len = 0;
            while (b < ev.size() && same_block(ev[a], ev[b])) { len = std::max(len, ev[b].len); b++; }
            const size_t off = rows.size();
            rows.resize(off + (size_t)len * UVC_CONBLOCK_ROW, 0);
            for (size_t f0 = a; f0 < b;) {   // one fragment of the block: its own maxima first, then one vote per base
                size_t f1 = f0; int32_t flen = 0;
                while (f1 < b && ev[f1].frag_rank == ev[f0].frag_rank) { flen = std::max(flen, ev[f1].len); f1++; }
                frag_rows.assign((size_t)flen * UVC_CONBLOCK_ROW, 0);
                for (size_t k = f0; k < f1; k++) add_event(*R, ev[k], frag_rows.data());
                fold_fragment(frag_rows.data(), flen, rows.data() + off);
                f0 = f1;
            }
            const int32_t nf = units[(size_t)ev[a].unit].n_fragments;
            B.push_back(UvcConBlock{ ev[a].fam, ev[a].strand, ev[a].type, ev[a].refpos, len, nf, (int64_t)(off / UVC_CONBLOCK_ROW) });
            a = b;
        }
        return finish(B, rows, blocks, block_capacity, n_blocks, out_rows, row_capacity, n_rows);
    } catch (const std::bad_alloc &) { return uvcgpu_set_error(UVCGPU_ENOMEM, "consensus blocks: out of host memory"); }
}

extern "C" int uvcgpu_consensus_block_to_seq(const int32_t *rows, int32_t len, int32_t right_to_left, int32_t trim_perc_dp, int32_t trim_n_consec, UvcConBase *out, int32_t *out_len) {
    if (len < 0 || (len > 0 && (!rows || !out)) || !out_len) return uvcgpu_set_error(UVCGPU_EINVAL, "consensus block: bad argument");
    auto depth = [&](int32_t k) { int32_t d = 0; for (int b = 0; b < 5; b++) d += rows[(size_t)k * UVC_CONBLOCK_ROW + b]; return d; };
    int32_t kept = len;
    if (trim_perc_dp >= 0) {   // ConsensusBlock_trim: cut at the trim_n_consec-th low-depth row (the counter never restarts: every row is "consecutive")
        int32_t max_dp = 0;
        for (int32_t k = 0; k < len; k++) max_dp = std::max(max_dp, depth(k));
        int32_t n_low = 0;
        for (int32_t k = 0; k < len; k++) {
            if ((int64_t)depth(k) * 100 < (int64_t)max_dp * trim_perc_dp) {
                n_low++;
                if (n_low >= trim_n_consec) { kept = k - (n_low - 1); break; }   // this row is not kept and n_low - 1 kept rows are dropped again
            }
        }
        if (kept < 0) kept = 0;
    }
    for (int32_t k1 = 0; k1 < kept; k1++) {
        const int32_t k = (right_to_left ? kept - 1 - k1 : k1);
        const int32_t *row = rows + (size_t)k * UVC_CONBLOCK_ROW;
        int con = 5; int32_t concount = 0, tot = 0;
        for (int b = 0; b < 5; b++) { if (row[b] > concount) { con = b; concount = row[b]; } tot += row[b]; }
        UvcConBase o; memset(&o, 0, sizeof(o));
        o.base = "ACGTN*"[con];
        o.quality = (int8_t)(row[UVC_CONBLOCK_BQ_SUM] / std::max(row[UVC_CONBLOCK_NFRAGS], 1));
        o.family_size = tot;
        o.family_identity = (int32_t)((double)concount / (double)std::max(tot, 1));
        out[k1] = o;
    }
    *out_len = kept;
    return 0;
}
