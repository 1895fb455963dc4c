// uvc_prep.h -- interface between uvc_host.cpp and uvc_prep.hip (the device-side preparation of uvcgpu_region_set_reads).
#ifndef UVC_PREP_H
#define UVC_PREP_H
#include "uvc_device.h"

// the UvcReadSoA columns as device pointers
struct UvcPrepIn {
    int64_t n_reads, n_bases, n_cigar_ops; int32_t n_fams;
    const int32_t *pos, *mpos, *isize, *nm, *l_qseq, *n_cigar, *frag_id, *fam_id;
    const uint16_t *flag; const uint8_t *mapq, *fam_strand, *fam_dflag;
    const int64_t *seq_off, *cigar_off;
    const uint32_t *cigars;
};
// device arrays (owned by the allocator's context) and host-side totals
struct UvcPrepOut {
    int32_t *endpos, *kind, *dflag_of, *frag_of, *fs_of, *p2_first, *complex_ids, *is_complex;
    int32_t *frag_beg, *frag_strand;   // FragRec::beg / strand as columns (keys of the fragment order)
    int64_t *table_off, *item_off, *gap_off;
    FragRec *frags; FsRec *fss;
    int32_t *generic_fs, *generic_sorted, *sweep_frags, *dup_units; int64_t *dup_off;
    int32_t *p2_aln, *p2_beg, *p2_end, *p2_qb, *p2_cls;   // P2 work-list entries in read order (the caller sorts them by (class, begin))
    int32_t n_frags, n_fs, n_complex, n_simple, n_generic, n_dup, n_sweep, n_frag_strand0;
    int32_t max_aln_span, max_frag_span, max_unit_span, max_unit_frags, max_p2_span, max_frag_depth, any_amplicon;
    int32_t p2_off[5];
    int64_t n_p2, table_rows, item_slots, gap_slots, ins_total, work, dup_work;
};
typedef void *(*UvcPrepAlloc)(void *ctx, size_t bytes, int zero);   // device memory that lives as long as the reads of the handle; NULL on failure
extern "C" int uvc_prep_reads(const UvcPrepIn *in, const UvcParams *P, int32_t rbeg, int32_t rend, int64_t npos, UvcPrepAlloc alloc, void *ctx, hipStream_t s,
                              UvcPrepOut *out, char *errmsg, int errcap);
#endif
