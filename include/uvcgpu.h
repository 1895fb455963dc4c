/* uvcgpu.h -- C ABI of the MI355X-native UVC hot path (libuvcgpu.so).
 *
 * The reference (genetronhealth/uvc v0.15.1) has no FFI/plugin layer; the seam this library
 * replaces is the pair of C++ call groups inside `process_batch` (main.cpp:458-1193):
 *
 *   accumulate:  Symbol2CountCoverageSet S(tid, beg, end+1)                 main.cpp:569
 *                S.updateByRegion3Aln(fq3, hap_bq, hap_fq, hap_f2q, alns3,
 *                    refstring, region_repeatvec, baq, baq2, prev, cur, params)
 *                                                   main.cpp:575-591 -> main.hpp:3665-3742
 *   score:       BcfFormat_symboltype_init   main.cpp:648 -> main.hpp:3889
 *                BcfFormat_symbol_init       main.cpp:911 -> main.hpp:4094
 *                BcfFormat_symbol_calc_DPv   main.cpp:931 -> main.hpp:4274
 *                BcfFormat_symbol_sum_DPv    main.cpp:957 -> main.hpp:4888
 *                BcfFormat_symbol_calc_qual  main.cpp:967 -> main.hpp:4908
 *
 * Everything crosses the boundary as plain pointers + sizes (no C++/torch types).  All entry
 * points return 0 on success and a negative UVCGPU_E* code on failure; uvcgpu_last_error()
 * gives the text.  Library code never aborts (the reference abort()s / exit()s instead,
 * e.g. grouping.cpp:59-87).  A handle is confined to one host thread at a time, which is how
 * the reference calls process_batch from its OpenMP loop (main.cpp:1478-1520).
 *
 * Coordinates are 0-based reference positions, as in the reference (uvc1_refgpos_t).
 */
#ifndef UVCGPU_H_INCLUDED
#define UVCGPU_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- alphabet (a1) ------------ */
/* AlignmentSymbol, main_conversion.hpp:316-334.  Numeric values are part of the ABI. */
enum {
    UVC_BASE_A = 0, UVC_BASE_C = 1, UVC_BASE_G = 2, UVC_BASE_T = 3, UVC_BASE_N = 4, UVC_BASE_NN = 5,
    UVC_LINK_M = 6, UVC_LINK_D3P = 7, UVC_LINK_D2 = 8, UVC_LINK_D1 = 9,
    UVC_LINK_I3P = 10, UVC_LINK_I2 = 11, UVC_LINK_I1 = 12, UVC_LINK_NN = 13,
    UVC_NUM_SYMBOLS = 14
};
enum { UVC_BASE_SYMBOL = 0, UVC_LINK_SYMBOL = 1 };       /* SymbolType, main_conversion.hpp:375-379 */
enum { UVC_PLATFORM_AUTO = 0, UVC_PLATFORM_ILLUMINA = 1, UVC_PLATFORM_IONTORRENT = 2, UVC_PLATFORM_OTHER = 3 };

/* ---------------------------------------------------------------- errors ------------------- */
enum {
    UVCGPU_OK = 0,
    UVCGPU_ENOREADS = -1,      /* region has no reads: process_batch returns -1 (main.cpp:520-523) */
    UVCGPU_EINVAL = -2,        /* malformed argument (bad offsets, read outside region, ...) */
    UVCGPU_EUNSUPPORTED = -3,  /* CIGAR op the reference itself throws on (process_cigar, main_conversion.hpp:902-916) or a shape beyond the documented limits
                                  (2^29 positions, 2^31 read bases, 2^27 InDel ops per region; bias_thres_interfering_indel above 10000) */
    UVCGPU_EDEVICE = -4,       /* HIP runtime failure / no gfx950 device */
    UVCGPU_ESTATE = -5,        /* call order violated (e.g. score before accumulate) */
    UVCGPU_ENOMEM = -6
};

/* ---------------------------------------------------------------- parameters (C15) --------- */
/* POD mirror of the hot-path fields of `struct CommandLineArgs` (CmdLineArgs.hpp:20-438).
 * `struct_size` versions the struct (set by uvcgpu_params_default). */
typedef struct UvcParams {
    int32_t struct_size;
    int32_t reserved_;
#define UVC_PI(name, dflt) int32_t name;
#define UVC_PD(name, dflt)
#include "uvc_params.def"
#undef UVC_PI
#undef UVC_PD
    int32_t pad_to_8_;
#define UVC_PI(name, dflt)
#define UVC_PD(name, dflt) double name;
#include "uvc_params.def"
#undef UVC_PI
#undef UVC_PD
} UvcParams;

/* Fills *p with the reference defaults (CmdLineArgs.hpp:43-417). */
void uvcgpu_params_default(UvcParams *p);
/* Applies the platform deltas of CommandLineArgs::selfUpdateByPlatform (CmdLineArgs.cpp:113-134)
 * and sets inferred_sequencing_platform / central_readlen / inferred_maxMQ, which the reference
 * infers from the first 5000 BAM records (CmdLineArgs.cpp:50-92). */
void uvcgpu_params_apply_platform(UvcParams *p, int32_t platform, int32_t central_readlen, int32_t max_mapq);

/* ---------------------------------------------------------------- reads (alns3) ------------ */
/* SoA image of `alns3` = vector<pair<array<vector<vector<bam1_t*>>,2>, MolecularBarcode>>
 * (main.hpp:3672): family -> strand(2) -> fragment (qname) -> alignment (R1, R2).
 * Reads must be grouped: all reads of one (fam_id, fam_strand, frag_id) contiguous, fragments
 * of one (fam_id, fam_strand) contiguous, both strands of one family contiguous.
 * Per read (htslib bam1_core_t fields, BAM spec):                                              */
typedef struct UvcReadSoA {
    int32_t struct_size;        /* sizeof(UvcReadSoA) of the header the caller was built with: the library refuses any other (fields were
                                   added to the end of this struct before, and will be again)                                         */
    int32_t reserved_;
    int64_t n_reads;
    const int32_t  *pos;        /* core.pos  (0-based leftmost)                                  */
    const int32_t  *mpos;       /* core.mpos                                                    */
    const int32_t  *isize;      /* core.isize, already NORM_INSERT_SIZE'd (common.hpp:73)       */
    const uint16_t *flag;       /* core.flag                                                    */
    const uint8_t  *mapq;       /* core.qual                                                    */
    const int32_t  *nm;         /* NM aux tag, -1 when absent (main.hpp:980-981)                */
    const int32_t  *l_qseq;     /* core.l_qseq                                                  */
    const int64_t  *seq_off;    /* offset of the read's first base in bases[] / quals[]; NULL = the reads lie back to back in read
                                   order (seq_off[i] = sum of l_qseq[0..i)): the library derives the offsets on the device            */
    const int64_t  *cigar_off;  /* offset of the read's first op in cigars[]; NULL = back to back in read order                     */
    const int32_t  *n_cigar;    /* core.n_cigar                                                 */
    const int32_t  *frag_id;    /* fragment (qname group) id, unique within (fam_id, strand)    */
    const int32_t  *fam_id;     /* family index, 0..n_fams-1                                    */
    const uint8_t  *fam_strand; /* 0/1 = bam_get_strand(aln) slot of alns3 (grouping.cpp:926)   */
    int64_t n_bases;
    const uint8_t  *bases;      /* 1 B/base = seq_nt16_int[bam_seqi()]: 0..3 = ACGT, 4 = other  */
    const uint8_t  *quals;      /* 1 B/base phred (bam_get_qual), after apply_bq_err_correction3 unless uvcgpu_region_correct_bq() is used */
    int64_t n_cigar_ops;
    const uint32_t *cigars;     /* BAM-packed ops: len<<4 | op                                  */
    int32_t n_fams;
    const uint8_t  *fam_dflag;  /* MolecularBarcode::duplexflag per family (grouping.cpp:931):
                                   0x1 UMI found, 0x2 duplex found, 0x4 amplicon, 0x8 borders preserved */
    /* Alternative to `bases` (then bases == NULL): the bases as the BAM record holds them (bam_get_seq: 4-bit codes "=ACMGRSVTWYHKDBN",
     * two per byte, high nibble first, every read starting on a byte boundary), the reads back to back in read order -- read i's bytes
     * start at sum of (l_qseq[j] + 1) / 2 over j < i.  The library applies seq_nt16_int[] on the device.  A quarter less to copy than
     * one byte per base + qual. */
    const uint8_t  *bases4;
    int64_t n_bases4_bytes;
} UvcReadSoA;

/* ---------------------------------------------------------------- per-position state (a2) -- */
/* Plane groups readable with uvcgpu_region_fetch().  Every group is a dense array
 * [n_planes][n_positions] (position fastest), position index = refpos - region begin.
 * "per symbol" groups are [field][symbol][pos]. */
enum UvcField {
    UVC_F_PREP32 = 0,   /* int32 [UVC_NPREP32][npos]          SegFormatPrepSet i32 fields, main_conversion.hpp:541-605 */
    UVC_F_PREP64 = 1,   /* int64 [UVC_NPREP64][npos]          SegFormatPrepSet i64 fields                             */
    UVC_F_THRES  = 2,   /* int32 [UVC_NTHRES][npos]           SegFormatThresSet, main_conversion.hpp:614-643          */
    UVC_F_SEG32  = 3,   /* int32 [UVC_NSEG32][14][npos]       SegFormatInfoSet i32 fields, main_conversion.hpp:645-691 */
    UVC_F_SEG64  = 4,   /* int64 [UVC_NSEG64][14][npos]       SegFormatInfoSet i64 fields                             */
    UVC_F_VQ     = 5,   /* int32 [UVC_NVQ][14][npos]          VQFormatTagSet stored slots, main_conversion.hpp:743-763 */
    UVC_F_BQSUM  = 6,   /* int32 [14][npos]                   bg_seg_bqsum_conslogo, main.hpp:2566                    */
    UVC_F_FRAG   = 7,   /* int32 [2][UVC_NFRAG][14][npos]     FragFormatDepthSet x strand, main_conversion.hpp:693-699 */
    UVC_F_FAM    = 8,   /* int32 [2][UVC_NFAM][14][npos]      FamFormatDepthSet x strand, main_conversion.hpp:722-734 */
    UVC_F_FAMINFO32 = 9,/* int32 [UVC_NFAMINFO32][14][npos]   FamFormatInfoSet i32 fields, main_conversion.hpp:701-720 */
    UVC_F_FAMINFO64 = 10,/* int64 [UVC_NFAMINFO64][14][npos]  FamFormatInfoSet i64 fields                             */
    UVC_F_DUPLEX = 11,  /* int32 [UVC_NDUPLEX][14][npos]      DuplexFormatDepthSet, main_conversion.hpp:736-741       */
    UVC_F_RTR    = 12,  /* int32 [UVC_NRTR][npos]             RegionalTandemRepeat AFTER P1b edits indelphred, common.hpp:150-160 */
    UVC_F_BAQ    = 13,  /* int64 [2][npos]                    baq_offsetarr, baq_offsetarr2, main.cpp:400-429         */
    UVC_NUM_FIELD_GROUPS = 14
};

/* plane indices inside the groups (order = declaration order in the reference structs) */
enum { /* UVC_F_PREP32 */
    UVC_P_a_dp = 0, UVC_P_a_near_ins_dp, UVC_P_a_near_del_dp, UVC_P_a_near_RTR_ins_dp, UVC_P_a_near_RTR_del_dp,
    UVC_P_a_pcr_dp, UVC_P_a_umi_dp, UVC_P_a_snv_dp, UVC_P_a_dnv_dp, UVC_P_a_highBQ_dp,
    UVC_P_a_near_pcr_clip_dp, UVC_P_a_near_long_clip_dp, UVC_P_a_at_ins_dp, UVC_P_a_at_del_dp,
    UVC_P_a_XM1500, UVC_P_a_GO1500, UVC_P_a_GAPLEN, UVC_P_a_qlen,
    UVC_P_a_near_ins_inv100len, UVC_P_a_near_del_inv100len,
    UVC_P_a_LIDP, UVC_P_a_RIDP,
    UVC_P_a_l_dist_sum, UVC_P_a_r_dist_sum, UVC_P_a_inslen_sum, UVC_P_a_dellen_sum,
    UVC_NPREP32
};
enum { /* UVC_F_PREP64 */
    UVC_P_a_near_ins_pow2len = 0, UVC_P_a_near_del_pow2len,
    UVC_P_a_near_ins_l_pow2len, UVC_P_a_near_ins_r_pow2len, UVC_P_a_near_del_l_pow2len, UVC_P_a_near_del_r_pow2len,
    UVC_P_a_LI, UVC_P_a_RI,
    UVC_P_a_l_BAQ_sum, UVC_P_a_r_BAQ_sum, UVC_P_a_insBAQ_sum, UVC_P_a_delBAQ_sum,
    UVC_NPREP64
};
enum { /* UVC_F_THRES */
    UVC_T_aLPxT = 0, UVC_T_aRPxT,
    UVC_T_aLI1T, UVC_T_aLI2T, UVC_T_aRI1T, UVC_T_aRI2T, UVC_T_aLI1t, UVC_T_aLI2t, UVC_T_aRI1t, UVC_T_aRI2t,
    UVC_T_aLP1t, UVC_T_aLP2t, UVC_T_aRP1t, UVC_T_aRP2t,
    UVC_T_aLB1t, UVC_T_aLB2t, UVC_T_aRB1t, UVC_T_aRB2t,
    UVC_NTHRES
};
enum { /* UVC_F_SEG32 */
    UVC_S_a2XM2 = 0, UVC_S_a2BM2, UVC_S_aPF1, UVC_S_aPF2, UVC_S_aBQ2, UVC_S_aMQs,
    UVC_S_aP1, UVC_S_aP2, UVC_S_aP3, UVC_S_aNC,
    UVC_S_aDPff, UVC_S_aDPfr, UVC_S_aDPrf, UVC_S_aDPrr,
    UVC_S_aLP1, UVC_S_aLP2, UVC_S_aLPL, UVC_S_aRP1, UVC_S_aRP2, UVC_S_aRPL,
    UVC_S_aLB1, UVC_S_aLB2, UVC_S_aRB1, UVC_S_aRB2,
    UVC_S_aLI1, UVC_S_aLI2, UVC_S_aRI1, UVC_S_aRI2, UVC_S_aRIf, UVC_S_aLIr,
    UVC_NSEG32
};
enum { UVC_S64_aLBL = 0, UVC_S64_aRBL, UVC_S64_aLIT, UVC_S64_aRIT, UVC_NSEG64 };   /* UVC_F_SEG64 */
enum { /* UVC_F_VQ: the 14 stored VQFormatTagSet slots */
    UVC_VQ_a1BQf = 0, UVC_VQ_a1BQr, UVC_VQ_a2BQf, UVC_VQ_a2BQr, UVC_VQ_bMQ,
    UVC_VQ_bIAQb, UVC_VQ_bIADb, UVC_VQ_bIDQb,
    UVC_VQ_cIAQf, UVC_VQ_cIADf, UVC_VQ_cIDQf, UVC_VQ_cIAQr, UVC_VQ_cIADr, UVC_VQ_cIDQr,
    UVC_NVQ
};
enum { UVC_FRAG_bDP = 0, UVC_FRAG_bTA, UVC_FRAG_bTB, UVC_NFRAG };                  /* UVC_F_FRAG */
enum { UVC_FAM_cDP1 = 0, UVC_FAM_cDP12, UVC_FAM_cDP2, UVC_FAM_cDP3, UVC_FAM_cDPM, UVC_FAM_cDPm, UVC_FAM_cDP21, UVC_FAM_cDPD, UVC_NFAM };
enum { /* UVC_F_FAMINFO32 */
    UVC_FI_c2LP1 = 0, UVC_FI_c2LP2, UVC_FI_c2LPL, UVC_FI_c2RP1, UVC_FI_c2RP2, UVC_FI_c2RPL, UVC_FI_c2LP0, UVC_FI_c2RP0,
    UVC_FI_c2LB1, UVC_FI_c2LB2, UVC_FI_c2RB1, UVC_FI_c2RB2, UVC_FI_c2BQ2,
    UVC_NFAMINFO32
};
enum { UVC_FI64_c2LBL = 0, UVC_FI64_c2RBL, UVC_NFAMINFO64 };
enum { UVC_DUPLEX_dDP1 = 0, UVC_DUPLEX_dDP2, UVC_NDUPLEX };
enum { UVC_RTR_begpos = 0, UVC_RTR_tracklen, UVC_RTR_unitlen, UVC_RTR_indelphred, UVC_RTR_anyTR_begpos, UVC_RTR_anyTR_tracklen, UVC_RTR_anyTR_unitlen, UVC_NRTR };

/* ---------------------------------------------------------------- scoring (a13-a17) -------- */
/* One scored allele = one (refpos, symbol[, indel string]) the reference would carry as a
 * bcfrec::BcfFormat through symbol_init -> calc_DPv -> sum_DPv -> calc_qual.  Integer FORMAT
 * fields only (bcf_formats_generator1.cpp:135-527); string fields stay host side. */
enum UvcScoreField {
    /* identity + depths (bit-exact class) */
    UVC_O_refpos = 0, UVC_O_symbol, UVC_O_refsymbol,
    UVC_O_DP, UVC_O_AD, UVC_O_bDP, UVC_O_bAD, UVC_O_c2DP, UVC_O_c2AD, UVC_O_bDPa, UVC_O_cDP0a,
    /* fill_symbol_VQ_fmts, main.hpp:3820-3887 */
    UVC_O_a2BQf, UVC_O_a2BQr, UVC_O_aBQ, UVC_O_aBQQ, UVC_O_bMQ,
    /* calc_DPv, main.hpp:4274-4844 */
    UVC_O_nPF0, UVC_O_nPF1, UVC_O_bNMa, UVC_O_bNMb, UVC_O_bNMQ,
    UVC_O_nNFA0, UVC_O_nNFA1, UVC_O_nNFA2, UVC_O_nNFA3, UVC_O_nNFA4, UVC_O_nNFA5,
    UVC_O_nAFA0, UVC_O_nAFA1, UVC_O_nAFA2, UVC_O_nAFA3, UVC_O_nAFA4, UVC_O_nAFA5, UVC_O_nAFA6, UVC_O_nAFA7, UVC_O_nAFA8,
    UVC_O_nBCFA0, UVC_O_nBCFA1, UVC_O_nBCFA2, UVC_O_nBCFA3, UVC_O_nBCFA4, UVC_O_nBCFA5, UVC_O_nBCFA6, UVC_O_nBCFA7, UVC_O_nBCFA8, UVC_O_nBCFA9,
    UVC_O_FTS,          /* bit i set <=> i-th fmt_bias_push (main.hpp:4745-4769) fired; 0 <=> "PASS" */
    UVC_O_tier2,        /* enable_tier2_consensus_format_tags */
    UVC_O_cDP1v, UVC_O_cDP1w, UVC_O_cDP1x, UVC_O_cDP2v, UVC_O_cDP2w, UVC_O_cDP2x,
    /* sum_DPv, main.hpp:4888-4906: [0] = sum over alleles, [1] = the NN allele */
    UVC_O_CDP1v0, UVC_O_CDP1v1, UVC_O_CDP1w0, UVC_O_CDP1w1, UVC_O_CDP1x0, UVC_O_CDP1x1,
    UVC_O_CDP2v0, UVC_O_CDP2v1, UVC_O_CDP2w0, UVC_O_CDP2w1, UVC_O_CDP2x0, UVC_O_CDP2x1,
    /* calc_qual, main.hpp:4908-5343 */
    UVC_O_cMmQ, UVC_O_aAaMQ, UVC_O_bMQQ, UVC_O_bIAQ, UVC_O_cIAQ,
    UVC_O_cPCQ1, UVC_O_cPLQ1, UVC_O_cPCQ2, UVC_O_cPLQ2, UVC_O_bTINQ, UVC_O_cTINQ,
    UVC_O_gVQ1, UVC_O_cVQ1, UVC_O_dVQinc, UVC_O_cVQ2, UVC_O_CONTQ,
    /* InDel alleles: gapSa = index (into uvcgpu_region_indel_alleles' rows) of the first row that carries this record's InDel string,
     * or -1 (not an InDel, or the string came from the caller); gapSa_len = indelstring.size() (main.cpp:907) */
    UVC_O_gapSa, UVC_O_gapSa_len,
    UVC_O_tkey,                         /* index of the tumor record (UvcScoreRequest::tumor_keys) this allele was paired with, or -1 */
    /* ---- the calling step behind calc_qual (main.cpp:990-1168): same value in every record of a (position, symbol type) group ---- */
    /* the two best non-reference alleles of the group by (max(cVQ1, cVQ2), cVQ1, cVQ2, symbol, InDel string), main.cpp:996-1016:
     * cVQ1M / cVQ2M, cVQAM as the symbol (14 = none), cVQSM as a row of uvcgpu_region_indel_alleles (-1 = no string) */
    UVC_O_cVQ1M0, UVC_O_cVQ1M1, UVC_O_cVQ2M0, UVC_O_cVQ2M1, UVC_O_cVQAM0, UVC_O_cVQAM1, UVC_O_cVQSM0, UVC_O_cVQSM1,
    UVC_O_vAC0, UVC_O_vAC1,             /* alleles at or above germ_phred_het3al per symbol type at this zerobased_pos, main.cpp:994-997, 1088 */
    /* output_germline, main.hpp:5483-5775: vNLODQ[own symbol type] = GL4raw[0] - max(GL4raw[1..3]); GL4raw; GST = a0..a3 LODQ + the four
     * het LODQs; best genotype index (0 "0/0", 1 "0/1", 2 "1/1", 3 "1/2"), its GQ; whether a GERMLINE line is written (needs
     * OUTVAR_GERMLINE); the records chosen as ref / alt1 / alt2, as record indices (-1 = the padding allele) */
    UVC_O_vNLODQ, UVC_O_GL4_0, UVC_O_GL4_1, UVC_O_GL4_2, UVC_O_GL4_3,
    UVC_O_GST0, UVC_O_GST1, UVC_O_GST2, UVC_O_GST3, UVC_O_GST4, UVC_O_GST5, UVC_O_GST6, UVC_O_GST7,
    UVC_O_germ_GT, UVC_O_germ_GQ, UVC_O_germ_emit, UVC_O_germ_ref, UVC_O_germ_alt1, UVC_O_germ_alt2,
    /* ---- per record: main.cpp:1081-1147 and the arithmetic of append_vcf_record (main.hpp:6027-6272) ---- */
    UVC_O_out,                          /* will_generate_out && !is_out_blocked: append_vcf_record is called for this record */
    UVC_O_vHGQ, UVC_O_NLODQ, UVC_O_NLODV /* argmin_nlodq_symbol, 14 = none */, UVC_O_TLODQ, UVC_O_SomaticQ,
    UVC_O_TNBQF0, UVC_O_TNBQF1, UVC_O_TNBQF2, UVC_O_TNBQF3, UVC_O_TNCQF0, UVC_O_TNCQF1, UVC_O_TNCQF2, UVC_O_TNCQF3,
    UVC_O_QUAL,                         /* vcfqual: the bit pattern of the 32-bit float the reference prints with std::to_string */
    UVC_O_FILTER,                       /* 0..5 = Q10..Q60 (bcfrec::FILTER_IDS), 6 = PASS */
    UVC_O_keep,                         /* the record is written: keep_var && tki.bDP >= min_ad, main.hpp:6253-6262 */
    /* FORMAT/FTS prints "<bias name>-<round(100 * biasFA / refFA)>" for every bias that fired (fmt_bias_push, main.hpp:4266-4269): those
     * percentages, 8 bits each (bias i of UVC_O_FTS in byte i % 4 of field i / 4, 0 where the bias did not fire, capped at 255) */
    UVC_O_FTSpct0, UVC_O_FTSpct1, UVC_O_FTSpct2, UVC_O_FTSpct3, UVC_O_FTSpct4,
    UVC_NUM_SCORE_FIELDS
};

/* Caller-supplied InDel alleles (optional override).  By default the library derives the alleles of every scored InDel symbol
 * itself, the way fill_by_indel_info / indel_get_majority do (main.hpp:5350-5455, instcode.hpp, main.cpp:853-896): one record per
 * distinct inserted sequence / deleted length whose fragment support is at least a quarter of the best one, with
 * bDPa / cDP0a = that allele's fragment / family support summed over both strands.  If the request lists alleles for a
 * (refpos, symbol), those are scored instead. */
typedef struct UvcIndelAllele {
    int32_t refpos;
    int32_t symbol;
    int32_t bDPa;       /* std::get<0>(bcad0a_indelstring_tki), main.cpp:923 */
    int32_t cDP0a;      /* std::get<1>(...),                    main.cpp:924 */
    int32_t indel_len;  /* indelstring.size(),                  main.cpp:907 */
} UvcIndelAllele;

enum { UVC_MGVCF_SYMBOL = 15, UVC_ADDITIONAL_INDEL_CANDIDATE_SYMBOL = 16 };   /* main_conversion.hpp: the VTI of the two position-level line types */
/* One tumor-sample record of the T/N channel (TumorKeyInfo, main_conversion.hpp:490-529), reduced to what the scoring functions read:
 * tpfa of calc_DPv = (cDP1x + 1) / (CDP1x + 2) (main.cpp:935), tpfa of calc_qual = (bDP + 0.5) / (BDP + 1) (main.cpp:985-986),
 * enable_tier2_consensus_format_tags (main.hpp:4475), and for InDels the length of the tumor record's inserted / deleted string
 * (main.cpp:867-880).  Only read when UvcParams::tumor_vcf_is_provided. */
typedef struct UvcTumorKey {
    int32_t refpos, symbol;     /* key (the VTI of the tumor record) */
    int32_t cDP1x, CDP1x, bDP, BDP;
    int32_t tier2;
    int32_t indel_len;
    /* the rest of TumorKeyInfo that the somatic quality of the normal-sample record reads (main.cpp:1104-1147, main.hpp:6095-6206) */
    int32_t cVQ1, cPCQ1, cDP2x, CDP2x, cVQ2, cPCQ2, bNMQ, vHGQ, tDP;
    /* INFO values the record of the normal sample repeats from the tumor record (main.cpp:366-372, main.hpp:6214-6218): tADR, tDPC */
    int32_t tAD0, tAD1, t2DP;
} UvcTumorKey;

/* One row of the per-strand InDel allele tables that fill_by_indel_info pushes into gapSeq / gapbAD1 / gapcAD1 / gc2AD / gc2dAD
 * (instcode.hpp:44-83): the allele counters kept beside FRAG_bDP, FAM_cDP12-after-filtering, FAM_cDP2 and FAM_cDPD / DUPLEX_dDP2
 * (main.hpp:2710-2717, 3327-3336, 3196-3206, 3458-3469, 3535-3546).  Rows are ordered by (refpos, symbol, strand) and inside a
 * group as the reference sorts them (descending (cAD1, bAD1, c2AD, c2dAD, sequence)). */
typedef struct UvcGapRow {
    int32_t refpos, symbol, strand;
    int32_t len;                /* inserted / deleted length */
    int64_t seq_off;            /* insertions: offset of `len` base codes (0..4 = ACGTN) in the sequence buffer; deletions: -1 (the deleted
                                 * bases are refseq[refpos - beg, +len)) */
    int32_t bAD1, cAD1, c2AD, c2dAD;
} UvcGapRow;

typedef struct UvcScoreRequest {
    int32_t pos_beg;            /* first zerobased_pos scored (rpos_inclu_beg, main.cpp:527); -1 = whole region core */
    int32_t pos_end;            /* exclusive */
    int32_t all_out;            /* paramset.should_output_all (-A), main.cpp:835 */
    int32_t is_amplicon;        /* ASSAY_TYPE_AMPLICON == inferred_assay_type, main.cpp:510-525 */
    int64_t n_indel_alleles;
    const UvcIndelAllele *indel_alleles;
    int64_t n_tumor_keys;       /* normal sample of a T/N pair: the tumor records of this region, sorted by (refpos, symbol); a position is
                                 * scored iff it has a record (extended_posidx_to_is_rescued, main.cpp:532-538), every symbol of it */
    const UvcTumorKey *tumor_keys;
    int32_t release_state;      /* 1: the caller is done with the planes of this region after this call (no further score / fetch until the next
                                 * accumulate, which return UVCGPU_ESTATE): the library zeroes them for the next accumulate while the records
                                 * travel to the host, instead of in front of the next accumulate's kernels */
    int32_t base_at_pos_beg;    /* 0: as process_batch at the start of a region, the BASE sub-position of pos_beg (refpos pos_beg - 1) is not
                                 * scored (main.cpp:643).  1: it is -- for a region that continues an adjacent one whose last zerobased_pos was
                                 * pos_beg - 1 (fixed tiles of one covered stretch): a run of such tiles then yields each (position, symbol type)
                                 * exactly once, the records of one uncut region.  Needs pos_beg > region begin */
    int32_t region_beg;         /* incluBegPosition of the BED line this region belongs to (main.cpp:655-656): besides every refpos that is a
                                 * multiple of 1000 an MGVCF block also opens at refpos == region_beg.  0 = nothing beyond the multiples */
    int32_t kept_only;          /* 1: return only the (zerobased_pos, symbol type) groups the record writer reads -- those with a written record
                                 * (keep && out) or a GERMLINE line (germ_emit) -- all records of such a group, in the usual order, germ_ref /
                                 * germ_alt1 / germ_alt2 re-based to the returned array.  UvcScoreOut::capacity then only has to hold these (a
                                 * few thousand per Mb at the default gate instead of ~54 k): the D2H shrinks by an order of magnitude.
                                 * uvcgpu_region_vcf_records writes the same text from either form */
    const char *const *tumor_sample_columns;   /* [n_tumor_keys] or NULL: the sample column of each tumor record as text.  Only the record writer reads
                                 * it: with is_tumor_format_retrieved the normal-sample line ends with the tumor's column (bcf1_to_string,
                                 * main.hpp:5897-5910, 6269; MGVCF / ADDITIONAL_INDEL_CANDIDATE lines: main.cpp:739-757, 784-798) */
    const char *const *tumor_ref_alt;          /* [n_tumor_keys] or NULL: "REF\tALT" of each tumor record (TumorKeyInfo::ref_alt, main.cpp:380).  Record writer
                                 * only: the InDel string of a rescued InDel record is the tumor's (main.cpp:867-880); without it such a record
                                 * is written with its symbolic allele */
} UvcScoreRequest;

typedef struct UvcScoreOut {
    int64_t capacity;           /* in: records the planes can hold */
    int64_t n_records;          /* out: records produced (may exceed capacity => UVCGPU_ENOMEM, nothing written) */
    int32_t *fields;            /* [UVC_NUM_SCORE_FIELDS][capacity], record index fastest */
} UvcScoreOut;

/* ---------------------------------------------------------------- entry points ------------- */
typedef struct uvcgpu_region uvcgpu_region_t;

/* Once per host thread / process.  Fails with UVCGPU_EDEVICE when there is no gfx950 device:
 * there is no CPU fallback inside this library. */
int uvcgpu_init(int device_id);
/* Number of HIP devices visible to the process (0 when there is none); the region-shard dispatchers spread their workers over them. */
int uvcgpu_device_count(void);
const char *uvcgpu_last_error(void);
const char *uvcgpu_version(void);

/* Replaces `Symbol2CountCoverageSet(tid, ext_beg, ext_end+1)` (main.cpp:569) together with the
 * region side arrays built just before it (main.cpp:553-563): refstring -> refstring2repeatvec
 * (main.hpp:803-874) -> the two BAQ prefix-sum arrays (main.cpp:400-429).
 * refseq = ASCII reference of [beg, end) (the caller's load_refstring result, +-MAX_STR_N_BASES halo), beg/end =
 * extended_inclu_beg_pos / extended_exclu_end_pos of main.cpp:529-530.  The per-position state then covers
 * [beg, end + 1), i.e. npos = end - beg + 1 positions, exactly like Symbol2CountCoverageSet(tid, beg, end + 1). */
int uvcgpu_region_create(uvcgpu_region_t **out, const UvcParams *params,
                         int32_t tid, int32_t beg, int32_t end, const char *refseq);
/* Re-binds the handle to another region (the next tile of a stream of tiles): as destroy + create, but the streams and, when the new
 * region is not longer than the longest one the handle has held, the device buffers are kept -- no hipMalloc / hipFree per tile. */
int uvcgpu_region_reset(uvcgpu_region_t *r, int32_t tid, int32_t beg, int32_t end, const char *refseq);
/* Copies the reads to the device (the caller keeps ownership of its buffers). */
int uvcgpu_region_set_reads(uvcgpu_region_t *r, const UvcReadSoA *reads);
/* The same for columns that are already in HBM: every pointer of `reads` is a device pointer (on the handle's device), nothing is copied.
 * The arrays must stay valid and unchanged until the handle gets other reads, is reset or destroyed; uvcgpu_region_correct_bq then edits
 * `quals` in place, as the reference edits its bam1_t (grouping.cpp:459-543).  For callers whose decoder already writes to the device, and
 * for measuring the path without the PCIe copy. */
int uvcgpu_region_set_reads_device(uvcgpu_region_t *r, const UvcReadSoA *reads);
/* Optional: apply_bq_err_correction3 (grouping.cpp:459-543) on the device copy of quals. */
int uvcgpu_region_correct_bq(uvcgpu_region_t *r);
/* The base qualities as they are on the device now (after uvcgpu_region_correct_bq if it was called); n must equal
 * UvcReadSoA::n_bases of the last set_reads.  For callers that write the corrected qualities back (the reference edits the
 * bam1_t in place, grouping.cpp:459-543) and for tests. */
int uvcgpu_region_read_quals(uvcgpu_region_t *r, uint8_t *dst, int64_t n);
/* Replaces updateByRegion3Aln (main.hpp:3665-3742): passes P1..P5b. Asynchronous on the handle's stream. */
int uvcgpu_region_accumulate(uvcgpu_region_t *r);
/* Upper bound of records a request can produce (2 symbol types x <= 8 symbols x positions). */
int64_t uvcgpu_region_score_size(const uvcgpu_region_t *r, const UvcScoreRequest *req);
/* Replaces the BcfFormat_symbol* call group.  Synchronous: returns after D2H of the records. */
int uvcgpu_region_score(uvcgpu_region_t *r, const UvcScoreRequest *req, UvcScoreOut *out);
/* Raw state access (the reference reads members directly, main.cpp:682-688, 759-760, 801-816). */
int64_t uvcgpu_region_field_bytes(const uvcgpu_region_t *r, int32_t field_group);
int uvcgpu_region_fetch(uvcgpu_region_t *r, int32_t field_group, void *dst, int64_t dst_bytes);
/* The InDel allele tables of the accumulated region (see UvcGapRow).  *n_rows / *seq_bytes receive the sizes needed; rows / seq are
 * filled when the capacities suffice, else UVCGPU_ENOMEM is returned and nothing is written.  Replaces the direct reads of
 * getPosToIseqToData / getPosToDlenToData / pos2iseq2data_cDP2 / pos2iseq2data_c2dDP in main.hpp:5350-5376. */
int uvcgpu_region_indel_alleles(uvcgpu_region_t *r, UvcGapRow *rows, int64_t row_capacity, int64_t *n_rows,
                                uint8_t *seq, int64_t seq_capacity, int64_t *seq_bytes);
/* The haplotype links of the accumulated region: the three std::vector<HapLink> that updateByRegion3Aln hands back (hap_bq, hap_fq,
 * hap_f2q, main.hpp:3665-3670) -- per link the mutated (refpos, symbol) pairs that one fragment (bq) / one UMI family (fq; f2q: confirmed by
 * the family vote) carried together, how many fragments / families on each strand carried exactly this set, and, for the three most
 * frequent sets, how many carried a superset (updateHapMap, main.hpp:3596-3663).  Links come grouped by `which` in the reference's order;
 * FORMAT/bHap, cHap, c2Hap of a record list the links that contain its (refpos, symbol) (main.cpp:82-97, main.hpp:5380-5404).
 * Sizes first (UVCGPU_ENOMEM with *n_links / *n_mut_ints set), then the data, as for uvcgpu_region_indel_alleles.  Needs the planes. */
typedef struct UvcHapLink {
    int32_t which;              /* 0 bq (fragments, duplicates kept), 1 fq (families), 2 f2q (families, tier-2 consensus) */
    int32_t n_muts;             /* pairs of this link */
    int64_t mut_off;            /* index of its first int in `muts`: refpos, symbol, refpos, symbol, ... */
    int32_t fr_cnt[2];          /* HapLink::fr_cnts */
    int32_t other_cnt[2];       /* HapLink::other_hap_cnts, -1 -1 beyond the phasing_haplotype_max_detail_cnt most frequent links */
} UvcHapLink;
int uvcgpu_region_hap_links(uvcgpu_region_t *r, UvcHapLink *links, int64_t link_capacity, int64_t *n_links,
                            int32_t *muts, int64_t mut_capacity, int64_t *n_mut_ints);
/* Every plane value of chosen positions, one row of uvcgpu_region_n_columns() int64 values per position: the groups in UvcField order
 * (PREP32 .. DUPLEX; RTR and BAQ are not part of a row), the planes of a group in the group's array order, i.e. column =
 * uvcgpu_region_column_base(group) + plane.  Positions outside the region give a row of zeros.  Replaces the getByPos() reads of
 * BcfFormat_symboltype_init / BcfFormat_symbol_init for the few positions whose records are written (main.hpp:3889-4251). */
int32_t uvcgpu_region_n_columns(void);
int32_t uvcgpu_region_column_base(int32_t field_group);   /* -1 for RTR / BAQ / out of range */
int uvcgpu_region_fetch_columns(uvcgpu_region_t *r, const int32_t *refpos, int64_t n, int64_t *dst /* [n][n_columns] */);

/* ---------------------------------------------------------------- VCF text ------ */
/* The header and the record lines of the reference's output for the scored records of a region: generate_vcf_header (the ##FILTER /
 * ##FORMAT / ##INFO / ##contig lines and the #CHROM line) and append_vcf_record + bcfrec::streamAppendBcfFormat (main.hpp:6027-6272,
 * bcf_formats_generator1.cpp:135-527, 643-690): CHROM POS ID REF ALT QUAL FILTER INFO FORMAT and the sample column with every FORMAT tag
 * of FORMAT_STRING_PER_REC(_WITHOUT_SSCS), in the reference's order and separators, bHap / cHap / c2Hap from uvcgpu_region_hap_links.
 * Not produced here: FORMAT/note (should_add_note) and the GERMLINE lines of output_germline.
 * Both return UVCGPU_ENOMEM with *len = the size needed when `capacity` is too small. */
const char *uvcgpu_vcf_format_keys(int32_t with_tier2_consensus_tags);
/* tumor_sample_name: NULL, or -- for the normal sample of a T/N pair with is_tumor_format_retrieved -- the name of the tumor VCF's sample,
 * which becomes a second sample column of the #CHROM line (generate_vcf_header, main.hpp:5785, 5881) */
int uvcgpu_vcf_header(const UvcParams *params, const char *sample_name, const char *tumor_sample_name, const char *const *contig_names,
                      const int64_t *contig_lens, int32_t n_contigs, char *dst, int64_t capacity, int64_t *len);
/* The same with the three lines that state facts about the caller's run, in the reference's places (main.hpp:5792-5794, 5870-5874):
 * ##fileDate=<file_date> (the reference prints strftime("%F %T")), ##reference=<reference_fname>, ##variantCallerCommand=<command_line>
 * (the reference joins argv with two blanks behind each word).  NULL leaves a line out.  Every other line -- ##ALT, ##FILTER, ##INFO,
 * ##FORMAT with their Description texts, ##phasing, ##variantCallerInferredParameters, #CHROM -- is the reference's, byte for byte;
 * ##variantCallerVersion names this library. */
int uvcgpu_vcf_header_ex(const UvcParams *params, const char *sample_name, const char *tumor_sample_name, const char *const *contig_names,
                         const int64_t *contig_lens, int32_t n_contigs, const char *file_date, const char *reference_fname,
                         const char *command_line, char *dst, int64_t capacity, int64_t *len);
/* `scored` is what uvcgpu_region_score filled for this region (all records, in order: the REF record of a position supplies the first
 * value of every Number=R tag); the lines of the records with out != 0 and keep != 0 are written.  The planes must not have been
 * released (UvcScoreRequest::release_state = 0).  `req` is the request of that score call (NULL = its defaults): its tumor_keys (normal
 * sample of a T/N pair), and its zerobased_pos range [pos_beg, pos_end), base_at_pos_beg and region_beg for the MGVCF block lines
 * (OUTVAR_MGVCF, main.cpp:655-735) and the ADDITIONAL_INDEL_CANDIDATE lines (main.cpp:759-799), which are written in front of the
 * records of their position. */
int uvcgpu_region_vcf_records(uvcgpu_region_t *r, const char *contig_name, const UvcScoreOut *scored, const UvcScoreRequest *req,
                              char *dst, int64_t capacity, int64_t *len);
/* Optional: page-lock a caller buffer that is handed to the library again and again (the records buffer of uvcgpu_region_score, read
 * arrays of uvcgpu_region_set_reads): copies then run at PCIe speed.  Unpin before freeing the buffer. */
int uvcgpu_pin_host_buffer(void *p, int64_t bytes);
int uvcgpu_unpin_host_buffer(void *p);
/* Page-locked host memory owned by the library (hipHostMalloc): the safe home of a records buffer that lives as long as a handle.  Pinning
 * a heap buffer in place works, but heap pages change roles -- memory that once was the (read-only mapped) source of a pageable upload and
 * was freed can come back as the buffer the records are written to, and the copy then faults ("write access to a read-only page"). */
int uvcgpu_host_alloc(void **p, int64_t bytes);
int uvcgpu_host_free(void *p);
int uvcgpu_region_sync(uvcgpu_region_t *r);
/* Record counts of the last uvcgpu_region_score on this handle: scored in all, and returned (fewer with UvcScoreRequest::kept_only). */
int uvcgpu_region_last_score_counts(const uvcgpu_region_t *r, int64_t *scored, int64_t *returned);
/* Self-check (a TEST entry point) of the three statements about ALL planes that scoring and the zero fill of a reused handle rely on:
 * (1) a cell of a symbol's planes at a position can be non-zero only if the symbol is the position's reference base / LINK_M or is marked
 * in the position's occupancy word (every writer of any other cell marks it: uvc_device.h occ_mark; uvc_kernels_score.hip type_mask);
 * (2) a (plane family, symbol, 4 096-position block) nobody marked holds only zeros (the fill skips it, the gather skips its FAMINFO /
 * DUPLEX planes); (3) cIAQ / cIAD / cIDQ of a strand are non-zero only behind a P5 bucket of that (strand, position).
 * Sweeps every plane of the last accumulate; *n_violations = cells that contradict one of them (0 expected).
 * The test suite and the soak scripts run it behind every accumulate (UVCGPU_CHECK_PRESENCE=1 in uvc_amd/region.py). */
int uvcgpu_region_check_presence(uvcgpu_region_t *r, int64_t *n_violations);
/* Measurement hooks (bench.py): HIP-event timing of every kernel of the LAST accumulate, recorded on the handle's stream.
 * kernel_times returns the number of kernels; `names` receives their names separated by ';'. */
int uvcgpu_region_set_profiling(uvcgpu_region_t *r, int on);
int uvcgpu_region_kernel_times(uvcgpu_region_t *r, char *names, int names_bytes, float *ms, int capacity);
void uvcgpu_region_destroy(uvcgpu_region_t *r);

/* ---------------------------------------------------------------- BGZF inflate on the device -- */
/* Replaces the zlib inflate behind htslib's bgzf_read for a batch of BGZF blocks (grouping.cpp:617-731 reads every alignment through it):
 * n raw DEFLATE payloads at comp + in_off[i] (in_len[i] bytes: the block without its 12 + XLEN header bytes and its 8 footer bytes) are
 * inflated to out + out_off[i] (out_len[i] = ISIZE bytes).  comp and out are HOST buffers of comp_bytes / out_bytes bytes.  The outputs
 * must tile one span of `out` in block order (out_off[i + 1] == out_off[i] + out_len[i], as a reader's batch does; UVCGPU_EINVAL otherwise):
 * only that span is written.  One lane per block, the Huffman tables of a wave's 64 decoders in LDS (uvc_inflate.hip).  The
 * CRC-32 of the footer is left to the caller (uvcio checks it on the returned bytes).  The signature is uvcio_inflate_fn of uvcio.h: pass
 * the function to uvcio_set_inflater.  Needs uvcgpu_init on the calling thread; UVCGPU_EINVAL names the first block whose stream is corrupt. */
int uvcgpu_bgzf_inflate(void *ctx, const uint8_t *comp, int64_t comp_bytes, const int64_t *in_off, const int32_t *in_len,
                        const int64_t *out_off, const int32_t *out_len, int64_t n_blocks, uint8_t *out, int64_t out_bytes);

#ifdef __cplusplus
}
#endif
#endif /* UVCGPU_H_INCLUDED */
