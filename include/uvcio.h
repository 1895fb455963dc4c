/* uvcio.h -- C ABI of the file readers in front of the MI355X UVC hot path ("next" row N3).
 *
 * Replaces the htslib calls of the reference's ingest: sam_open / sam_hdr_read / sam_index_load / sam_itr_queryi / sam_itr_next
 * (grouping.cpp:157-314, 617-731) and fai_load / faidx_fetch_seq (main.cpp:99-130, 529-531) -- BGZF + BAM + BAI and FASTA + .fai,
 * written against the SAM/BAM specification (SAMv1.pdf sections 4 and 5) on zlib only.  Host code, no GPU: the columns it returns are
 * exactly what include/uvcgroup.h (family assignment) and UvcReadSoA (include/uvcgpu.h) take.
 *
 * Return codes: 0 or a negative UVCGPU_E* value of uvcgpu.h; uvcio_last_error() has the text.
 */
#ifndef UVCIO_H
#define UVCIO_H
#include <stddef.h>
#include <stdint.h>
#include "uvcgpu.h"   /* UvcTumorKey, UVCGPU_E* */
#ifdef __cplusplus
extern "C" {
#endif

typedef struct uvcio_bam uvcio_bam_t;
typedef struct uvcio_fasta uvcio_fasta_t;

/* Alignments of one query, in file order, as columns.  The arrays belong to the handle and stay valid until its next fetch / close. */
typedef struct UvcBamBatch {
    int64_t n_alns;
    const int32_t *tid, *pos, *endpos /* bam_endpos: pos + reference-consuming CIGAR lengths, pos + 1 if there are none */;
    const int32_t *mtid, *mpos, *isize;
    const uint16_t *flag;
    const uint8_t *mapq;
    const int32_t *nm;          /* NM aux tag (bam_aux2i), -1 if absent */
    const int32_t *l_qseq, *n_cigar;
    const int64_t *seq_off, *cigar_off, *qname_off;
    int64_t n_bases;
    const uint8_t *bases;       /* seq_nt16_int codes: A C G T -> 0..3, everything else 4 */
    const uint8_t *quals;
    int64_t n_cigar_ops;
    const uint32_t *cigars;     /* BAM-packed: len << 4 | op */
    int64_t n_qname_bytes;
    const char *qnames;         /* NUL-terminated names, qname_off[i] is the start of the i-th */
} UvcBamBatch;

const char *uvcio_last_error(void);

/* sam_open + sam_hdr_read + sam_index_load (<path>.bai or <path minus .bam>.bai; without an index every fetch scans the file) */
int uvcio_bam_open(uvcio_bam_t **out, const char *path);
int32_t uvcio_bam_n_refs(const uvcio_bam_t *b);
const char *uvcio_bam_ref_name(const uvcio_bam_t *b, int32_t tid);
int64_t uvcio_bam_ref_len(const uvcio_bam_t *b, int32_t tid);
int uvcio_bam_has_index(const uvcio_bam_t *b);
/* sam_itr_queryi(idx, tid, beg, end) + the sam_itr_next loop: every alignment of `tid` with pos < end and bam_endpos > beg (0-based,
 * half-open), in file order; unmapped reads placed on `tid` take part with endpos = pos + 1, as in htslib */
int uvcio_bam_fetch(uvcio_bam_t *b, int32_t tid, int64_t beg, int64_t end, UvcBamBatch *out);
void uvcio_bam_close(uvcio_bam_t *b);

/* fai_load + faidx_fetch_seq: needs <path>.fai; the sequence comes back upper-cased (main.cpp:124-127 does the same) */
int uvcio_fasta_open(uvcio_fasta_t **out, const char *path);
int64_t uvcio_fasta_seq_len(const uvcio_fasta_t *f, const char *name);      /* -1 if unknown */
int uvcio_fasta_fetch(uvcio_fasta_t *f, const char *name, int64_t beg, int64_t end, char *dst /* [end - beg] */);
void uvcio_fasta_close(uvcio_fasta_t *f);

/* BGZF writer: the block-gzipped stream the reference writes its VCF through (bgzf_open / bgzf_write / bgzf_close, main.cpp:1196-1215,
 * 1571-1583): blocks of at most 0xff00 input bytes and the 28-byte end-of-file marker.  level 0..9, anything else = 6. */
typedef struct uvcio_bgzf_writer uvcio_bgzf_writer_t;
int uvcio_bgzf_write_open(uvcio_bgzf_writer_t **out, const char *path, int32_t level);
int uvcio_bgzf_write(uvcio_bgzf_writer_t *w, const void *data, int64_t n);
int uvcio_bgzf_write_close(uvcio_bgzf_writer_t *w);

/* The region planner: SamIter::iternext without a BED file (grouping.cpp:225-312; its memory model grouping.cpp:28-67) over the
 * alignment columns of a file or query in file order.  One cut per block the reference would hand to process_batch: `flag` = 16 contig
 * changed | 8 gap of more than 200 bp | 4 per-thread memory budget | 2 end of file, `batch` = the iternext() call that returns it. */
typedef struct UvcRegionCut { int32_t tid, beg, end, flag, batch; int64_t n_reads; } UvcRegionCut;
/* The same walk as a stream (it keeps a handful of scalars between alignments): feed the alignment columns piece by piece in file order, take the
 * finished cuts as they appear, finish at the end of the file.  Memory is bounded by the piece, not by the file. */
typedef struct uvcio_planner uvcio_planner_t;
int uvcio_planner_open(uvcio_planner_t **out, const int64_t *target_len, int32_t n_targets, int32_t nthreads, int64_t mem_per_thread_mb);
int uvcio_planner_feed(uvcio_planner_t *p, const int32_t *tid, const int32_t *pos, const int32_t *endpos, const uint16_t *flag, int64_t n);
int uvcio_planner_finish(uvcio_planner_t *p);
int64_t uvcio_planner_take(uvcio_planner_t *p, UvcRegionCut *out, int64_t capacity);
int64_t uvcio_planner_pending(const uvcio_planner_t *p);
void uvcio_planner_close(uvcio_planner_t *p);
int uvcio_plan_regions(const int32_t *tid, const int32_t *pos, const int32_t *endpos, const uint16_t *flag, int64_t n,
                       const int64_t *target_len, int32_t n_targets, int32_t nthreads, int64_t mem_per_thread_mb,
                       UvcRegionCut *out, int64_t capacity, int64_t *n_out);

/* ---- the tumor VCF of a T/N pair: rescue_variants_from_vcf, main.cpp:183-398 ----
 * Reads the (block-gzipped or plain) VCF the tumor pass wrote and turns every record into the integers the normal pass reads:
 * key (contig, symbolpos, VTI[1]) with symbolpos = POS - 1 for substitutions / MGVCF block / ADDITIONAL_INDEL_CANDIDATE lines and POS for
 * InDels (main.cpp:279), FORMAT BDPb, bDPf, bDPr, CDP1x, cDP1x, cVQ1, cPCQ1, CDP2x, cDP2x, cVQ2, cPCQ2, bNMQ, vHGQ, CDP1b, cDP1f, cDP1r,
 * CDP2b (main.cpp:294-372), the presence of _C2XP (:386-388) and the length of the inserted / deleted string (REF / ALT, main.cpp:867-880).
 * Lines with a symbolic ALT other than <NON_REF> / <ADDITIONAL_INDEL_CANDIDATE> are skipped, and those two as well unless
 * is_tumor_format_retrieved (main.cpp:265-272); lines without FORMAT/VTI are skipped (:275).  The reference reads through htslib's synced
 * reader restricted to the regions of the batch; here the file is read once and queried per region.
 * `contig_names` maps CHROM to tid (the BAM header's order). */
typedef struct uvcio_tumor_vcf uvcio_tumor_vcf_t;
int uvcio_tumor_vcf_open(uvcio_tumor_vcf_t **out, const char *path, const char *const *contig_names, int32_t n_contigs, int32_t is_tumor_format_retrieved);
const char *uvcio_tumor_vcf_sample_name(const uvcio_tumor_vcf_t *v);   /* last column of the #CHROM line ("" if there is none) */
int64_t uvcio_tumor_vcf_n_records(const uvcio_tumor_vcf_t *v);
/* The records of `tid` with pos_beg <= symbolpos <= pos_end, sorted by (symbolpos, symbol) -- tkis_beg .. tkis_end of main.cpp:532-533 --
 * as UvcScoreRequest::tumor_keys / tumor_sample_columns / tumor_ref_alt take them.  The arrays belong to the handle (valid until it is closed). */
int uvcio_tumor_vcf_fetch(const uvcio_tumor_vcf_t *v, int32_t tid, int32_t pos_beg, int32_t pos_end, const UvcTumorKey **keys, const char *const **sample_columns,
                          const char *const **ref_alts, int64_t *n);
void uvcio_tumor_vcf_close(uvcio_tumor_vcf_t *v);

/* CRC-32 (the zlib / BGZF footer polynomial) as the reader and the writer compute it: carry-less multiplication on CPUs that have it. */
/* Where the two large columns of a batch (UvcBamBatch::bases, ::quals) live: NULL, NULL = the C heap.  A caller that hands the batch to
 * uvcgpu_region_set_reads passes page-locked memory of the GPU library here (wrappers of uvcgpu_host_alloc / uvcgpu_host_free), so that the
 * 2 x 300 MB of a 1 Mb x 300x tile travel by DMA.  Set it before the first uvcio_bam_fetch; buffers are kept and grown per BAM handle. */
void uvcio_set_column_allocator(void *(*alloc_fn)(size_t), void (*free_fn)(void *));

/* The BGZF blocks of a batch inflated somewhere else than on the host's cores.  fn gets the raw DEFLATE payloads of n blocks (in_off / in_len
 * inside comp) and the place of each block's output (out_off / out_len = the block's ISIZE) and returns 0 when every byte of every block is
 * in place; the reader then checks each block's CRC-32 itself.  Anything else (non-zero return, a CRC mismatch) and the reader inflates the
 * batch on the host as if no function had been set.  uvcgpu_bgzf_inflate (uvcgpu.h) has this signature: uvc1-mi355x --device-inflate sets
 * it (the calling thread must have selected its device, uvcgpu_init).  Batches with fewer than min_blocks blocks stay on the host.
 * Replaces bgzf_read's inflate of the reference's reader (htslib behind grouping.cpp:617-731).  Process-wide; set it before the first fetch.
 * The reader hands over whole batches: the outputs of the n blocks tile one span of `out` in block order (out_off[i + 1] == out_off[i] + out_len[i]). */
typedef int (*uvcio_inflate_fn)(void *ctx, const uint8_t *comp, int64_t comp_bytes, const int64_t *in_off, const int32_t *in_len,
                                const int64_t *out_off, const int32_t *out_len, int64_t n, uint8_t *out, int64_t out_bytes);
void uvcio_set_inflate(uvcio_inflate_fn fn, void *ctx, int32_t min_blocks);
uint32_t uvcio_crc32(const void *p, int64_t n);
/* Test hook: one raw DEFLATE stream of known output size through the library's own decoder (uvc_inflate_fast.h), which the BGZF reader tries
 * before zlib: 1 = decoded (out holds out_len bytes), 0 = declined (the reader would hand the block to zlib). */
int uvcio_inflate_raw_fast(const void *in, int64_t in_len, void *out, int64_t out_len);

/* ---- region shards ----
 * The reference balances its chunks by reads and positions (main.cpp:1380-1400).  Without reading the alignments the cost of a tile is
 * estimated from the BAI linear index: the compressed bytes between the 16 kb windows that hold its two ends (0 without an index). */
int64_t uvcio_bam_region_bytes(const uvcio_bam_t *b, int32_t tid, int64_t beg, int64_t end);
/* Cuts an ordered list of n tiles with the given costs into n_shards contiguous runs of about equal total cost: shard_of[i] is
 * non-decreasing in i, so that the outputs of the shards, concatenated in shard order, are in tile order (uvcTN.sh:92-101 does the
 * same per chromosome with bcftools concat).  A tile goes to the shard that holds the midpoint of its cost interval. */
int uvcio_plan_shards(const int64_t *cost, int64_t n, int32_t n_shards, int32_t *shard_of);
/* bcftools concat -n (uvcTN.sh:100): the BGZF files one after the other, the 28-byte end-of-file marker of all but the last dropped. */
int uvcio_bgzf_concat(const char *out_path, const char *const *in_paths, int32_t n_in);
/* The whole text of a (block-)gzipped or plain file (the tumor VCF of a T/N pair); *buf is malloc'ed, the caller frees it. */
int uvcio_read_text_file(const char *path, char **buf, int64_t *len);

#ifdef __cplusplus
}
#endif
#endif
