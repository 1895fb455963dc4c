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
int uvcio_plan_regions(const int32_t *tid, const int32_t *pos, const int32_t *endpos, const uint16_t *flag, int64_t n,
                       const int64_t *target_len, int32_t n_targets, int32_t nthreads, int64_t mem_per_thread_mb,
                       UvcRegionCut *out, int64_t capacity, int64_t *n_out);

#ifdef __cplusplus
}
#endif
#endif
