/* uvcgroup.h -- C ABI of the family-assignment pass of genetronhealth/uvc on MI355X ("next" N4).
 *
 * Replaces the arithmetic of bamfname_to_strand_to_familyuid_to_reads (grouping.cpp:608-997): the per-alignment filter
 * (fill_isrc_isr2_beg_end_with_aln, :347-415), the end-position histograms and their prefix sums (:650-699), the peak snapping
 * (poscounter_to_pos2pcenter, :422-442), the amplicon classification and dedup_idflag choice (:793-878), the MolecularBarcode
 * key (MolecularID.hpp:20-69) and the grouping into family -> strand -> fragment -> alignment that process_batch receives as
 * alns3.  BAM decoding stays with the caller (htslib); it hands over plain columns.
 *
 * Strings never reach the device: a read name is represented by strhash(qname, 31) and strhash(qname, 17) (Hash.hpp:6-31;
 * the reference itself keys fragments by the base-17 hash, grouping.cpp:766, 950), a UMI by the same two hashes of the text
 * between the '#' marks (grouping.cpp:767-786).  uvcgpu_qname_digest computes all of them on the host.  Two different
 * names / UMIs are treated as equal only if both 64-bit hashes collide.
 * The reference orders families with MolecularBarcode::operator< (positions, then the strings); here families come out
 * ordered by a 64-bit mix of the key -- the order of families does not enter any result of the hot path (integer sums).
 */
#ifndef UVCGROUP_H
#define UVCGROUP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct UvcGroupParams {
    int32_t struct_size;            /* sizeof(UvcGroupParams) */
    int32_t fetch_tbeg, fetch_tend; /* the region being called (grouping.cpp:614-615) */
    int32_t end2end;                /* BED_END_TO_END_BIT of the region flag (iohts.hpp:12) */
    int32_t inferred_sequencing_platform;   /* as UvcParams */
#define UVC_GI(name, dflt) int32_t name;
#define UVC_GD(name, dflt)
#include "uvc_group_params.def"
#undef UVC_GI
#undef UVC_GD
    int32_t pad_;
#define UVC_GI(name, dflt)
#define UVC_GD(name, dflt) double name;
#include "uvc_group_params.def"
#undef UVC_GI
#undef UVC_GD
} UvcGroupParams;

/* FilterReason, grouping.cpp:329-345 */
enum { UVC_FR_NOT_FILTERED = 0, UVC_FR_NOT_MAPPED, UVC_FR_NOT_PRIMARY_ALN, UVC_FR_LOW_MAPQ, UVC_FR_LOW_ALN_LEN, UVC_FR_LOW_ISIZE, UVC_FR_HIGH_ISIZE,
       UVC_FR_ZERO_ISIZE, UVC_FR_OUT_OF_RANGE, UVC_FR_NOT_END_TO_END,
       UVC_FR_NOT_IN_WINDOW = 100,   /* second scan: outside [fetch_tbeg - 2001, fetch_tend + 2001], grouping.cpp:733-735 */
       UVC_FR_QNAME_NOT_VISITED = 101 /* second scan: no alignment of this read name overlapped the region in the first scan, :736-738 */ };

typedef struct UvcGroupInput {      /* every alignment sam_itr_queryi(tid, fetch_tbeg - 2000, fetch_tend + 2000) returns, in file order */
    int64_t n_alns;
    const int32_t *tid, *pos, *endpos /* bam_endpos */, *mtid, *mpos, *isize;
    const uint16_t *flag;
    const uint8_t *mapq;
    const uint64_t *qname_hash31, *qname_hash17;   /* strhash(qname, 31), strhash(qname, 17) */
    const uint64_t *umi_hash31, *umi_hash17;       /* the same hashes of the UMI text; 0 when umi_kind == 0 */
    const uint8_t *umi_kind;                       /* bit0: UMI found (>= 1 letter, molecule_tag != NONE); bit1: duplex-structured "alpha+beta" */
} UvcGroupInput;

typedef struct UvcGroupOut {        /* caller-allocated arrays of n_alns elements */
    int32_t *filter_reason;         /* per input alignment */
    int32_t *isize_norm;            /* per input alignment: NORM_INSERT_SIZE applied (common.hpp:75) -- what the hot path must see */
    int32_t *order;                 /* [n_kept] input indices of the kept alignments in alns3 order: family, strand, fragment, file order */
    int32_t *fam_id, *frag_id;      /* [n_kept], non-decreasing along `order` */
    uint8_t *fam_strand;            /* [n_kept] bam_get_strand */
    uint8_t *fam_dflag;             /* [n_fams] MolecularBarcode::duplexflag: 1 UMI, 2 duplex, 4 amplicon, 8 borders preserved */
    uint8_t *fam_idflag;            /* [n_fams] dedup_idflag: 1 beg, 2 end, 4 qname, 8 UMI */
    int64_t n_kept; int32_t n_fams, n_frags;
    int32_t extended_inclu_beg_pos, extended_exclu_end_pos;   /* grouping.cpp:760-761 */
    int64_t n_amplicon;             /* pcrpassed */
    int64_t n_visited_qnames;
} UvcGroupOut;

void uvcgpu_group_params_default(UvcGroupParams *p);
/* Hash.hpp:6-39 */
uint64_t uvcgpu_strnhash(const char *s, size_t n, uint64_t base);
uint64_t uvcgpu_hash2hash(uint64_t h1, uint64_t h2);
/* grouping.cpp:763-786: hashes of the read name and of its UMI ("name#UMI" or "name#UMI#..."); returns umi_kind */
int uvcgpu_qname_digest(const char *qname, int molecule_tag, int disable_duplex, uint64_t *qname_hash31, uint64_t *qname_hash17, uint64_t *umi_hash31, uint64_t *umi_hash17);
/* the same for n NUL-terminated names stored back to back (names + off[i]), e.g. UvcBamBatch::qnames / qname_off of uvcio.h */
int uvcgpu_qname_digest_batch(const char *names, const int64_t *off, int64_t n, int molecule_tag, int disable_duplex,
                              uint64_t *qname_hash31, uint64_t *qname_hash17, uint64_t *umi_hash31, uint64_t *umi_hash17, uint8_t *umi_kind);
/* bam2umihash (grouping.cpp:569-606, 787-792): the in-read UMI pattern of single-end (IonTorrent) reads.  `umi_struct` is the pattern the
 * reference takes from the environment variable ONE_STEP_UMI_STRUCT (main.cpp:1224-1225), e.g. "NNNACTNNNTGA": N = a UMI letter, anything
 * else must match; it is looked for at the first five offsets of the read and, failing that, reverse-complemented from its end.  For
 * every unpaired alignment whose umi_kind has no UMI from the name, a hit sets bit 0 of umi_kind[i] (MolecularBarcode::duplexflag 0x1 and
 * the UMI arm of the dedup_idflag choice follow from it); umi_hash (optional) receives the base-16 hash of the UMI letters, which the
 * reference computes and never reads.  The UMI hash pair of such a read stays 0, 0: its key carries an empty umistring (grouping.cpp:929). */
int uvcgpu_umi_in_read_batch(const char *umi_struct, const uint8_t *bases, const int64_t *seq_off, const int32_t *l_qseq, const uint16_t *flag, int64_t n,
                             uint8_t *umi_kind, uint64_t *umi_hash);
/* 0 or a negative UVCGPU_E* code (uvcgpu_last_error() has the text) */
int uvcgpu_group_families(const UvcGroupParams *params, const UvcGroupInput *in, UvcGroupOut *out);

#ifdef __cplusplus
}
#endif
#endif
