// TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT (see oracle_common.hpp).
//
// Restatement of the reference's insertion / soft-clip consensus blocks in the reference's own object structure:
//   ConsensusBlockSet (one map position -> block per block type)      main_consensus.hpp:116-225
//   ConsensusBlock_trim, consensusBlockToSeqQual                       main_consensus.hpp:52-114
//   the calls that fill them: updateByAln's INS / soft-clip arms        main.hpp:2009, 2100-2116, 2259-2279
//   the family fold of P4: updateByMajorMinusMinor<true>                main.hpp:1722, 2875-2911
// over the boundary types of include/uvcconsensus.h.  PARITY STATUS: "parity unpinned" (the header includes htslib types through
// main_conversion.hpp and cannot be compiled here); checked by a hand-built example and an independent Python restatement in
// tests/test_conblock.py.
#include "oracle_common.hpp"
#include "uvcconsensus.h"

namespace uvco {

typedef std::array<i32, UVC_CONBLOCK_ROW> BaseToCount;        // main_consensus.hpp:41
typedef std::vector<BaseToCount> ConBlock;                    // :42

struct ConBlockSet {                                          // :116
    std::map<i32, ConBlock> pos2conblock;
    void inc_by_pos_seq_qual(i32 pos, const std::vector<int> &seq /* symbols */, const std::vector<int8_t> &qual) {   // :121-135
        ConBlock &cb2 = pos2conblock[pos];
        while (cb2.size() < seq.size()) cb2.push_back(BaseToCount{{ 0 }});
        for (size_t k = 0; k < seq.size(); k++) {
            cb2[k][(size_t)seq[k]] = max_(cb2[k][(size_t)seq[k]], (i32)qual[k]);
            cb2[k][UVC_CONBLOCK_BQ_SUM] = max_(cb2[k][UVC_CONBLOCK_BQ_SUM], (i32)qual[k]);
            cb2[k][UVC_CONBLOCK_NFRAGS] = 1;
        }
    }
    void inc_by_major_minus_minor(const ConBlockSet &other) {                                                            // :178-203
        for (const auto &kv : other.pos2conblock) {
            const ConBlock &cb1 = kv.second;
            ConBlock &cb2 = pos2conblock[kv.first];
            while (cb2.size() < cb1.size()) cb2.push_back(BaseToCount{{ 0 }});
            for (size_t k = 0; k < cb1.size(); k++) {
                int conbase = UVC_BASE_NN; i32 concount = 0, totcount = 0;
                for (int b = UVC_BASE_A; b <= UVC_BASE_N; b++) { if (cb1[k][(size_t)b] > concount) { conbase = b; concount = cb1[k][(size_t)b]; } totcount += cb1[k][(size_t)b]; }
                cb2[k][(size_t)conbase] += 1;
                cb2[k][UVC_CONBLOCK_BQ_SUM] += (i32)nnminus((i64)concount * 2, totcount);
                cb2[k][UVC_CONBLOCK_NFRAGS] += 1;
            }
        }
    }
};

static ConBlock conblock_trim(const ConBlock &cb, i32 perc_dp_thres, i32 n_consec_thres) {                                // :52-86
    i32 max_dp = 0;
    for (const auto &row : cb) { i32 d = 0; for (int b = UVC_BASE_A; b <= UVC_BASE_N; b++) d += row[(size_t)b]; max_dp = max_(max_dp, d); }
    ConBlock ret;
    i32 prev_pos = 0, curr_pos = 0, n_consec = 0;
    for (const auto &row : cb) {
        curr_pos++;
        i32 d = 0; for (int b = UVC_BASE_A; b <= UVC_BASE_N; b++) d += row[(size_t)b];
        if ((i64)d * 100 < (i64)max_dp * perc_dp_thres) {
            if (prev_pos + 1 == curr_pos) n_consec++; else n_consec = 1;
            if (n_consec >= n_consec_thres) { for (i32 i = 1; i < n_consec; i++) ret.pop_back(); return ret; }
        }
        prev_pos = curr_pos;
        ret.push_back(row);
    }
    return ret;
}
static std::vector<UvcConBase> conblock_to_seq(const ConBlock &cb, bool right2left) {                                     // :88-114
    static const char DESC[] = "ACGTN*";
    std::vector<UvcConBase> ret;
    for (size_t k1 = 0; k1 < cb.size(); k1++) {
        const size_t k = (right2left ? cb.size() - k1 - 1 : k1);
        int conbase = UVC_BASE_NN; i32 concount = 0, totcount = 0;
        for (int b = UVC_BASE_A; b <= UVC_BASE_N; b++) { if (cb[k][(size_t)b] > concount) { conbase = b; concount = cb[k][(size_t)b]; } totcount += cb[k][(size_t)b]; }
        UvcConBase o; memset(&o, 0, sizeof(o));
        o.base = DESC[conbase];
        o.quality = (int8_t)(cb[k][UVC_CONBLOCK_BQ_SUM] / max_(cb[k][UVC_CONBLOCK_NFRAGS], 1));
        o.family_size = totcount;
        o.family_identity = (i32)((double)concount / (double)max_(totcount, 1));
        ret.push_back(o);
    }
    return ret;
}

// the INS / soft-clip arms of updateByAln for one alignment into the three block sets of its fragment-level object
static void aln_to_conblocks(const UvcParams &P, const UvcReadSoA &R, i64 i, std::array<ConBlockSet, UVC_NUM_CONBLOCK_TYPES> &sets) {
    const u32 *cigar = R.cigars + R.cigar_off[i];
    const int n_cigar = R.n_cigar[i];
    i32 rend = R.pos[i];
    for (int c = 0; c < n_cigar; c++) { const int op = cigar[c] & 0xF; if (op == C_MATCH || op == C_DEL || op == C_REF_SKIP || op == C_EQUAL || op == C_DIFF) rend += (i32)(cigar[c] >> 4); }
    const bool isrc = ((R.flag[i] & 0x10) == 0x10);
    const u8 dflag = (R.fam_dflag ? R.fam_dflag[R.fam_id[i]] : 0);
    const bool is_assay_amplicon = ((dflag & 0x4) || ((P.primerlen > 0) && !(0x2 & P.primer_flag)));                    // main.hpp:1805-1806
    const bool normal_filter_primers = (P.tn_is_paired && (0x1 & P.primer_flag));                                        // :1867
    const i64 ibeg = ((R.isize[i] != 0) ? (min_(R.pos[i], R.mpos[i]) + P.primerlen) : ((isrc && (0x0 == (0x1 & R.flag[i]))) ? 0 : (R.pos[i] + P.primerlen)));
    const i64 iend = ((R.isize[i] != 0) ? nnminus((i64)min_(R.pos[i], R.mpos[i]) + abs(R.isize[i]), P.primerlen)
                                        : ((isrc && (0x0 == (0x1 & R.flag[i]))) ? nnminus(rend, P.primerlen) : (i64)INT32_MAX));   // :1872-1875
    i64 qpos = 0; i32 rpos = R.pos[i];
    const u8 *bases = R.bases + R.seq_off[i], *quals = R.quals + R.seq_off[i];
    for (int c = 0; c < n_cigar; c++) {
        const int op = cigar[c] & 0xF; const i32 oplen = (i32)(cigar[c] >> 4);
        if (op == C_MATCH || op == C_EQUAL || op == C_DIFF) { qpos += oplen; rpos += oplen; }
        else if (op == C_INS) {
            if ((normal_filter_primers || !is_assay_amplicon) || (ibeg <= rpos && rpos < iend)) {                        // :2009
                std::vector<int> iseq; std::vector<int8_t> iqual;
                for (i32 k = 0; k < oplen; k++) { iseq.push_back(min_((int)bases[qpos + k], (int)UVC_BASE_N)); iqual.push_back((int8_t)quals[qpos + k]); }
                sets[UVC_CONBLOCK_INS].inc_by_pos_seq_qual(rpos, iseq, iqual);                                           // :2114-2116
            }
            qpos += oplen;
        } else if (op == C_DEL) { rpos += oplen; }
        else {
            if (op == C_SOFT_CLIP) {                                                                                     // :2259-2279
                std::vector<int> iseq; std::vector<int8_t> iqual;
                for (i32 k = 0; k < oplen; k++) { iseq.push_back(min_((int)bases[qpos + k], (int)UVC_BASE_N)); iqual.push_back((int8_t)quals[qpos + k]); }
                const int type = ((0 == c) ? UVC_CONBLOCK_SOFTCLIP_RIGHT_TO_LEFT : UVC_CONBLOCK_SOFTCLIP_LEFT_TO_RIGHT);
                if (type == UVC_CONBLOCK_SOFTCLIP_RIGHT_TO_LEFT) { std::reverse(iseq.begin(), iseq.end()); std::reverse(iqual.begin(), iqual.end()); }
                sets[(size_t)type].inc_by_pos_seq_qual(rpos, iseq, iqual);
            }
            if (op == C_SOFT_CLIP) qpos += oplen;                                                                        // process_cigar
            else if (op == C_REF_SKIP) rpos += oplen;
        }
    }
}

static int emit(const std::vector<UvcConBlock> &B, const std::vector<i32> &rows, UvcConBlock *blocks, int64_t bcap, int64_t *nb, int32_t *out_rows, int64_t rcap, int64_t *nr) {
    if (nb) *nb = (int64_t)B.size();
    if (nr) *nr = (int64_t)rows.size();
    if ((int64_t)B.size() > bcap || (int64_t)rows.size() > rcap) return UVCGPU_ENOMEM;
    if (!B.empty()) memcpy(blocks, B.data(), sizeof(UvcConBlock) * B.size());
    if (!rows.empty()) memcpy(out_rows, rows.data(), sizeof(i32) * rows.size());
    return 0;
}
static void sets_to_rows(const std::array<ConBlockSet, UVC_NUM_CONBLOCK_TYPES> &sets, i32 fam, i32 strand, i32 n_frag, std::vector<UvcConBlock> &B, std::vector<i32> &rows) {
    for (int type = 0; type < UVC_NUM_CONBLOCK_TYPES; type++) for (const auto &kv : sets[(size_t)type].pos2conblock) {
        B.push_back(UvcConBlock{ fam, strand, type, kv.first, (i32)kv.second.size(), n_frag, (int64_t)(rows.size() / UVC_CONBLOCK_ROW) });
        for (const auto &row : kv.second) rows.insert(rows.end(), row.begin(), row.end());
    }
}
}   // namespace uvco

using namespace uvco;
extern "C" {
int uvc_oracle_consensus_blocks_of_fragment(const UvcParams *P, const UvcReadSoA *R, int64_t first_read, int64_t n, UvcConBlock *blocks, int64_t bcap, int64_t *nb, int32_t *rows, int64_t rcap, int64_t *nr) {
    std::array<ConBlockSet, UVC_NUM_CONBLOCK_TYPES> sets;
    for (i64 i = first_read; i < first_read + n; i++) aln_to_conblocks(*P, *R, i, sets);
    std::vector<UvcConBlock> B; std::vector<i32> out;
    sets_to_rows(sets, n ? R->fam_id[first_read] : 0, n ? R->fam_strand[first_read] : 0, 1, B, out);
    return emit(B, out, blocks, bcap, nb, rows, rcap, nr);
}
int uvc_oracle_consensus_blocks(const UvcParams *P, const UvcReadSoA *R, const UvcConBlockRequest *req, UvcConBlock *blocks, int64_t bcap, int64_t *nb, int32_t *rows, int64_t rcap, int64_t *nr) {
    std::vector<UvcConBlock> B; std::vector<i32> out;
    for (i64 u0 = 0; u0 < R->n_reads;) {   // one family-strand unit (alns2, main.hpp:2865-2911)
        i64 u1 = u0;
        while (u1 < R->n_reads && R->fam_id[u1] == R->fam_id[u0] && R->fam_strand[u1] == R->fam_strand[u0]) u1++;
        // fillTidBegEndFromAlns2 (main.hpp:658-688): every alignment adds one to the end reached so far
        i32 beg2 = INT32_MAX, end2 = 0, n_frag = 0;
        for (i64 i = u0; i < u1; i++) {
            i32 rend = R->pos[i];
            const u32 *cigar = R->cigars + R->cigar_off[i];
            for (int c = 0; c < R->n_cigar[i]; c++) { const int op = cigar[c] & 0xF; if (op == C_MATCH || op == C_DEL || op == C_REF_SKIP || op == C_EQUAL || op == C_DIFF) rend += (i32)(cigar[c] >> 4); }
            beg2 = min_(beg2, R->pos[i]); end2 = max_(end2, rend) + 1;
            if (i == u0 || R->frag_id[i] != R->frag_id[i - 1]) n_frag++;
        }
        const bool applicable = (req->min_fragments <= n_frag);                                                           // :2875
        auto overlapping = [](i64 a0, i64 a1, i64 b0, i64 b1) { return !((a1 <= b0) || (b1 <= a0)); };                   // common.hpp:92
        const bool only_done_here = ((req->prev_tid != req->tid) || !overlapping(req->prev_beg, req->prev_end, beg2, end2)) && overlapping(req->curr_beg, req->curr_end, beg2, end2);   // :2876-2878
        if (applicable && only_done_here) {
            std::array<ConBlockSet, UVC_NUM_CONBLOCK_TYPES> fam_sets;
            for (i64 f0 = u0; f0 < u1;) {
                i64 f1 = f0;
                while (f1 < u1 && R->frag_id[f1] == R->frag_id[f0]) f1++;
                std::array<ConBlockSet, UVC_NUM_CONBLOCK_TYPES> frag_sets;
                for (i64 i = f0; i < f1; i++) aln_to_conblocks(*P, *R, i, frag_sets);
                for (int type = 0; type < UVC_NUM_CONBLOCK_TYPES; type++) fam_sets[(size_t)type].inc_by_major_minus_minor(frag_sets[(size_t)type]);   // :1722, 2909-2911
                f0 = f1;
            }
            sets_to_rows(fam_sets, R->fam_id[u0], R->fam_strand[u0], n_frag, B, out);
        }
        u0 = u1;
    }
    return emit(B, out, blocks, bcap, nb, rows, rcap, nr);
}
int uvc_oracle_consensus_block_to_seq(const int32_t *rows, int32_t len, int32_t right_to_left, int32_t trim_perc_dp, int32_t trim_n_consec, UvcConBase *out, int32_t *out_len) {
    ConBlock cb((size_t)len);
    for (int32_t k = 0; k < len; k++) for (int q = 0; q < UVC_CONBLOCK_ROW; q++) cb[(size_t)k][(size_t)q] = rows[(size_t)k * UVC_CONBLOCK_ROW + q];
    if (trim_perc_dp >= 0) cb = conblock_trim(cb, trim_perc_dp, trim_n_consec);
    const std::vector<UvcConBase> v = conblock_to_seq(cb, right_to_left != 0);
    for (size_t k = 0; k < v.size(); k++) out[k] = v[k];
    *out_len = (int32_t)v.size();
    return 0;
}
}
