/* uvcconsensus.h -- C ABI of the insertion / soft-clip consensus blocks.
 *
 * Replaces ConsensusBlockSet and its helpers of the reference (main_consensus.hpp:52-225) as P4 fills and reads them:
 *   per fragment   updateByRead1Aln<BASE_QUALITY_MAX, false, TIsBlockConsensus = true> calls incByPosSeqQual for every insertion
 *                  (main.hpp:2100-2116, behind the amplicon primer gate :2009) and every soft clip (:2259-2279; the clip at CIGAR index 0 is
 *                  stored reversed, "fixed right, variable left"),
 *   per family     updateByMajorMinusMinor<true> -> incByMajorMinusMinor adds each fragment's block set (main.hpp:1722, 2909-2911),
 *   on output      consensusBlockToSeqQual (and ConsensusBlock_trim inside returnSeqQualVec) turn a block into bases / qualities.
 * The reference reads the result only with --fam-consensus-out-fastq (main.hpp:2947-2954, 3056-3133); the FASTQ text itself is outside
 * this library (string formatting, off by default).
 *
 * Host code: the sequences are short, rare and keyed by (family, position) -- there is nothing here for the device, and no GPU is
 * needed to call these functions.  The read columns are host pointers, the same UvcReadSoA uvcgpu_region_set_reads takes.
 */
#ifndef UVCCONSENSUS_H_INCLUDED
#define UVCCONSENSUS_H_INCLUDED

#include "uvcgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ConsensusBlockCigarType, main_consensus.hpp:13-18 */
enum UvcConBlockType {
    UVC_CONBLOCK_SOFTCLIP_LEFT_TO_RIGHT = 0,   /* CONSENSUS_BLOCK_CSOFT_CLIP_FIXED_LEFT_TO_VAR_RIGHT: a soft clip that is not the first CIGAR op */
    UVC_CONBLOCK_INS = 1,                      /* CONSENSUS_BLOCK_CINS */
    UVC_CONBLOCK_SOFTCLIP_RIGHT_TO_LEFT = 2,   /* CONSENSUS_BLOCK_CSOFT_CLIP_FIXED_RIGHT_TO_VAR_LEFT: the soft clip at CIGAR index 0, stored reversed */
    UVC_NUM_CONBLOCK_TYPES = 3
};
/* One row of a ConsensusBlock = BaseToCount (main_consensus.hpp:41): A C G T N, BASE_NN (unused), BASE2COUNT_BQ_SUM_IDX, BASE2COUNT_NFRAGS_IDX */
enum { UVC_CONBLOCK_ROW = 8, UVC_CONBLOCK_BQ_SUM = 6, UVC_CONBLOCK_NFRAGS = 7 };

/* One (family, strand, block type, anchor position) entry of read_family_mmm_ampl's ConsensusBlockSets */
typedef struct UvcConBlock {
    int32_t fam_id, strand;      /* UvcReadSoA::fam_id / fam_strand of the unit */
    int32_t type;                /* UvcConBlockType */
    int32_t refpos;              /* key of pos2conblock: rpos of the CIGAR op */
    int32_t len;                 /* rows (inserted / clipped bases; the longest fragment-level block of this key) */
    int32_t n_fragments;         /* fragments of the unit (alns2.size()) */
    int64_t row_off;             /* first row in the rows array: len rows of UVC_CONBLOCK_ROW int32 */
} UvcConBlock;

/* FastqConsensusBase, main_consensus.hpp:33-38 */
typedef struct UvcConBase {
    char base;
    int8_t quality;
    int16_t pad_;
    int32_t family_size;
    int32_t family_identity;     /* the reference stores the ratio concount / totcount in an integer: 0 or 1 */
} UvcConBase;

typedef struct UvcConBlockRequest {
    int32_t min_fragments;       /* paramset.fam_consensus_out_fastq_thres_dup1add: units with fewer fragments have no family-level blocks (main.hpp:2875) */
    int32_t tid;                 /* contig of the reads */
    int32_t curr_beg, curr_end;  /* curr_bedline [beg, end) -- only units whose span overlaps it ... */
    int32_t prev_tid, prev_beg, prev_end;   /* ... and not prev_bedline (prev_tid = -1: none) are done here (is_consensus_only_done_here, main.hpp:2876-2878) */
    int32_t reserved_;
} UvcConBlockRequest;

/* Family-level blocks of every unit of `reads`, ordered by (fam_id, strand, type, refpos) -- the iteration order of the reference (units in
 * alns3 order, ALL_CONSENSUS_BLOCK_CIGAR_TYPES, std::map by position).  P: primerlen / primer_flag / tn_is_paired gate the insertions of
 * amplicon families.  blocks / rows may be NULL with capacity 0 to ask for the sizes (UVCGPU_ENOMEM, counts set). */
int uvcgpu_consensus_blocks(const UvcParams *P, const UvcReadSoA *reads, const UvcConBlockRequest *req,
                            UvcConBlock *blocks, int64_t block_capacity, int64_t *n_blocks, int32_t *rows, int64_t row_capacity, int64_t *n_rows);

/* The same for ONE fragment (its reads contiguous in `reads`, [first_read, first_read + n_reads_of_fragment)): read_ampBQerr_fragWithR1R2's
 * block sets after updateByRead1Aln, i.e. what incByPosSeqQual left (per base the maximum quality per base symbol, NFRAGS = 1). */
int uvcgpu_consensus_blocks_of_fragment(const UvcParams *P, const UvcReadSoA *reads, int64_t first_read, int64_t n_reads_of_fragment,
                                        UvcConBlock *blocks, int64_t block_capacity, int64_t *n_blocks, int32_t *rows, int64_t row_capacity, int64_t *n_rows);

/* consensusBlockToSeqQual (main_consensus.hpp:88-114), optionally after ConsensusBlock_trim (:52-86; trim_perc_dp < 0: no trimming;
 * returnSeqQualVec's defaults are 20 and 3).  out must hold `len` elements; *out_len = elements written. */
int uvcgpu_consensus_block_to_seq(const int32_t *rows, int32_t len, int32_t right_to_left, int32_t trim_perc_dp, int32_t trim_n_consec_positions,
                                  UvcConBase *out, int32_t *out_len);

#ifdef __cplusplus
}
#endif
#endif
