// uvc_vcf.cpp -- VCF text of the scored records of a region: the host half of append_vcf_record and of the generated
// bcfrec::streamAppendBcfFormat (main.hpp:6027-6272, bcf_formats_generator1.cpp:135-527, 643-690), on top of the public C ABI:
// the score records (uvcgpu_region_score), the plane columns of the positions that are written (uvcgpu_region_fetch_columns, one small
// gather kernel) and the InDel allele rows (uvcgpu_region_indel_alleles).  O(emitted records); nothing here is on the hot path.
#include "uvcgpu.h"
#include "uvc_hap.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

extern "C" const char *uvcgpu_region_refseq(const uvcgpu_region_t *r, int32_t *beg, int32_t *end);   // uvc_host.cpp
extern "C" const int32_t *uvcgpu_region_repeat_tracks(const uvcgpu_region_t *r, int64_t *npos);    // host copy, [UVC_NRTR][npos]
extern "C" const UvcParams *uvcgpu_region_params(const uvcgpu_region_t *r);
extern "C" const std::vector<UvcHapLinkHost> *uvcgpu_region_hap_(uvcgpu_region_t *r);   // the three link vectors (uvc_host.cpp), NULL on error
extern "C" int uvcgpu_fail_(int code, const char *msg);
extern "C" int uvcgpu_region_block_stats_(uvcgpu_region_t *r, int32_t refpos_beg, int32_t refpos_end, int32_t *dst);   // 10 ints per position, k_block_stats

namespace {
const int NSYM = 14;
const char *const SYMBOL_DESC[NSYM + 1] = { "A", "C", "G", "T", "N", "*", "<LR>", "<LD3P>", "<LD2>", "<LD1>", "<LI3P>", "<LI2>", "<LI1>", "*", "<NONE>" };   // main_conversion.hpp:336-346
const char *const FILTER_Q[7] = { "Q10", "Q20", "Q30", "Q40", "Q50", "Q60", "PASS" };
// the FORMAT/FTS names in the order BcfFormat_symbol_calc_DPv pushes them (main.hpp:4745-4769): 9 against nAFA, 10 against nBCFA
const char *const FTS_NAMES[19] = { "aStrand", "aBQXM", "aInsertSize", "aAlignL", "aAlignR", "aPositionL", "aPositionR", "abPositionL", "abPositionR",
                                    "bcDup", "cbDup", "c0Orientation", "c2Orientation", "c2PositionL", "c2PositionR", "c2AlignL", "c2AlignR", "c2StrictPosL", "c2StrictPosR" };

inline bool is_base(int s) { return s <= UVC_BASE_NN; }
inline bool is_ins(int s) { return s == UVC_LINK_I1 || s == UVC_LINK_I2 || s == UVC_LINK_I3P; }
inline bool is_del(int s) { return s == UVC_LINK_D1 || s == UVC_LINK_D2 || s == UVC_LINK_D3P; }

// ---- the FORMAT tags, in the order of the reference's FORMAT_VEC ----
enum Kind {
    K_SEP, K_SPECIAL,
    K_R_S32, K_R_S64, K_R_VQ, K_R_FRf, K_R_FRr, K_R_FAf, K_R_FAr, K_R_FI, K_R_FI64, K_R_DU,   // Number=R from the planes of the REF and ALT symbol
    K_T1_S32, K_T1L_S32, K_T1_S64, K_T1_VQ, K_T1_FI,   // "2,1": the sum over the symbols of the type (int32 tags truncate it)
    K_T2_S32, K_T2_DU,                                   // "2": that sum and the value of the type's NN symbol
    K_FR2_FR, K_FR2_FA, K_NN2_FA, K_ZERO2,               // "2": forward and reverse sums; the NN symbol twice (CDP1d); never filled (BDPd, CDP2d)
    K_POS,                                               // position-level lists
    K_R_REC, K_1_REC, K_N_REC,                           // from the score record(s): Number=R, one value, n consecutive fields
};
struct Tag { const char *name; const char *number; const char *type; int kind; int a; int n; bool sscs; };   // the ##FORMAT lines themselves (with the reference's Description texts): uvc_vcf_header_lines.inc
#define SEG "sequenced-segment (read) statistic"
#define FRA "fragment statistic, duplicates kept"
#define FAM "de-duplicated fragment / UMI-family statistic"
#define SSC "tier-2 single-strand consensus (SSCS) statistic"
const Tag TAGS[] = {
    { "GT", "1", "String", K_SPECIAL, 0, 0, false },
    { "GQ", "1", "Integer", K_SPECIAL, 0, 0, false },
    { "HQ", "2", "Integer", K_SPECIAL, 0, 0, false },
    { "FT", "1", "String", K_SPECIAL, 0, 0, false },
    { "FTS", "A", "String", K_SPECIAL, 0, 0, false },
    { "_A_", "1", "String", K_SEP, 0, 0, false },
    { "DP", "1", "Integer", K_1_REC, UVC_O_DP, 1, false },
    { "AD", "R", "Integer", K_R_REC, UVC_O_AD, 1, false },
    { "bDP", "1", "Integer", K_1_REC, UVC_O_bDP, 1, false },
    { "bAD", "R", "Integer", K_R_REC, UVC_O_bAD, 1, false },
    { "c2DP", "1", "Integer", K_1_REC, UVC_O_c2DP, 1, false },
    { "c2AD", "R", "Integer", K_R_REC, UVC_O_c2AD, 1, false },
    { "_Aa", "1", "String", K_SEP, 0, 0, false },
    { "APDP", "12", "Integer", K_POS, 0, 12, false },
    { "APXM", "8", "Integer", K_POS, 1, 8, false },
    { "_Ab", "1", "String", K_SEP, 0, 0, false },
    { "APLRID", "4", "Integer", K_POS, 2, 4, false },
    { "APLRI", "4", "Integer", K_POS, 3, 4, false },
    { "APLRP", "4", "Integer", K_POS, 4, 4, false },
    { "_Ac", "1", "String", K_SEP, 0, 0, false },
    { "ALRPxT", "2", "Integer", K_POS, 5, 2, false },
    { "ALRIT", "4", "Integer", K_POS, 6, 4, false },
    { "ALRIt", "4", "Integer", K_POS, 7, 4, false },
    { "ALRPt", "4", "Integer", K_POS, 8, 4, false },
    { "ALRBt", "4", "Integer", K_POS, 9, 4, false },
    { "_AQ", "1", "String", K_SEP, 0, 0, false },
    { "aMQs", "R", "Integer", K_R_S32, UVC_S_aMQs, 1, false },
    { "AMQs", "1", "Integer", K_T1_S32, UVC_S_aMQs, 1, false },
    { "a1BQf", "R", "Integer", K_R_VQ, UVC_VQ_a1BQf, 1, false },
    { "A1BQf", "1", "Integer", K_T1_VQ, UVC_VQ_a1BQf, 1, false },
    { "a1BQr", "R", "Integer", K_R_VQ, UVC_VQ_a1BQr, 1, false },
    { "A1BQr", "1", "Integer", K_T1_VQ, UVC_VQ_a1BQr, 1, false },
    { "_ADPf", "1", "String", K_SEP, 0, 0, false },
    { "aDPff", "R", "Integer", K_R_S32, UVC_S_aDPff, 1, false },
    { "ADPff", "2", "Integer", K_T2_S32, UVC_S_aDPff, 1, false },
    { "aDPfr", "R", "Integer", K_R_S32, UVC_S_aDPfr, 1, false },
    { "ADPfr", "2", "Integer", K_T2_S32, UVC_S_aDPfr, 1, false },
    { "_ADPr", "1", "String", K_SEP, 0, 0, false },
    { "aDPrf", "R", "Integer", K_R_S32, UVC_S_aDPrf, 1, false },
    { "ADPrf", "2", "Integer", K_T2_S32, UVC_S_aDPrf, 1, false },
    { "aDPrr", "R", "Integer", K_R_S32, UVC_S_aDPrr, 1, false },
    { "ADPrr", "2", "Integer", K_T2_S32, UVC_S_aDPrr, 1, false },
    { "_ALP", "1", "String", K_SEP, 0, 0, false },
    { "aLP1", "R", "Integer", K_R_S32, UVC_S_aLP1, 1, false },
    { "ALP1", "1", "Integer", K_T1_S32, UVC_S_aLP1, 1, false },
    { "aLP2", "R", "Integer", K_R_S32, UVC_S_aLP2, 1, false },
    { "ALP2", "1", "Integer", K_T1_S32, UVC_S_aLP2, 1, false },
    { "aLPL", "R", "Integer", K_R_S32, UVC_S_aLPL, 1, false },
    { "ALPL", "1", "Integer", K_T1L_S32, UVC_S_aLPL, 1, false },
    { "_ARP", "1", "String", K_SEP, 0, 0, false },
    { "aRP1", "R", "Integer", K_R_S32, UVC_S_aRP1, 1, false },
    { "ARP1", "1", "Integer", K_T1_S32, UVC_S_aRP1, 1, false },
    { "aRP2", "R", "Integer", K_R_S32, UVC_S_aRP2, 1, false },
    { "ARP2", "1", "Integer", K_T1_S32, UVC_S_aRP2, 1, false },
    { "aRPL", "R", "Integer", K_R_S32, UVC_S_aRPL, 1, false },
    { "ARPL", "1", "Integer", K_T1L_S32, UVC_S_aRPL, 1, false },
    { "_ALB", "1", "String", K_SEP, 0, 0, false },
    { "aLB1", "R", "Integer", K_R_S32, UVC_S_aLB1, 1, false },
    { "aLB2", "R", "Integer", K_R_S32, UVC_S_aLB2, 1, false },
    { "ALB2", "1", "Integer", K_T1_S32, UVC_S_aLB2, 1, false },
    { "aLBL", "R", "Integer", K_R_S64, UVC_S64_aLBL, 1, false },
    { "ALBL", "1", "Integer", K_T1_S64, UVC_S64_aLBL, 1, false },
    { "_ARB", "1", "String", K_SEP, 0, 0, false },
    { "aRB1", "R", "Integer", K_R_S32, UVC_S_aRB1, 1, false },
    { "aRB2", "R", "Integer", K_R_S32, UVC_S_aRB2, 1, false },
    { "ARB2", "1", "Integer", K_T1_S32, UVC_S_aRB2, 1, false },
    { "aRBL", "R", "Integer", K_R_S64, UVC_S64_aRBL, 1, false },
    { "ARBL", "1", "Integer", K_T1_S64, UVC_S64_aRBL, 1, false },
    { "_ALI", "1", "String", K_SEP, 0, 0, false },
    { "aLI1", "R", "Integer", K_R_S32, UVC_S_aLI1, 1, false },
    { "aLI2", "R", "Integer", K_R_S32, UVC_S_aLI2, 1, false },
    { "ALI2", "1", "Integer", K_T1_S32, UVC_S_aLI2, 1, false },
    { "aLIr", "R", "Integer", K_R_S32, UVC_S_aLIr, 1, false },
    { "ALIr", "1", "Integer", K_T1_S32, UVC_S_aLIr, 1, false },
    { "_ARI", "1", "String", K_SEP, 0, 0, false },
    { "aRI1", "R", "Integer", K_R_S32, UVC_S_aRI1, 1, false },
    { "aRI2", "R", "Integer", K_R_S32, UVC_S_aRI2, 1, false },
    { "ARI2", "1", "Integer", K_T1_S32, UVC_S_aRI2, 1, false },
    { "aRIf", "R", "Integer", K_R_S32, UVC_S_aRIf, 1, false },
    { "ARIf", "1", "Integer", K_T1_S32, UVC_S_aRIf, 1, false },
    { "_AX", "1", "String", K_SEP, 0, 0, false },
    { "aBQ2", "R", "Integer", K_R_S32, UVC_S_aBQ2, 1, false },
    { "ABQ2", "1", "Integer", K_T1_S32, UVC_S_aBQ2, 1, false },
    { "aPF2", "R", "Integer", K_R_S32, UVC_S_aPF2, 1, false },
    { "APF2", "1", "Integer", K_T1_S32, UVC_S_aPF2, 1, false },
    { "aP1", "R", "Integer", K_R_S32, UVC_S_aP1, 1, false },
    { "AP1", "1", "Integer", K_T1_S32, UVC_S_aP1, 1, false },
    { "aP2", "R", "Integer", K_R_S32, UVC_S_aP2, 1, false },
    { "AP2", "1", "Integer", K_T1_S32, UVC_S_aP2, 1, false },
    { "_Ax", "1", "String", K_SEP, 0, 0, false },
    { "aPF1", "R", "Integer", K_R_S32, UVC_S_aPF1, 1, false },
    { "aLIT", "R", "Integer", K_R_S64, UVC_S64_aLIT, 1, false },
    { "aRIT", "R", "Integer", K_R_S64, UVC_S64_aRIT, 1, false },
    { "aP3", "R", "Integer", K_R_S32, UVC_S_aP3, 1, false },
    { "aNC", "R", "Integer", K_R_S32, UVC_S_aNC, 1, false },
    { "_BDP", "1", "String", K_SEP, 0, 0, false },
    { "bDPf", "R", "Integer", K_R_FRf, UVC_FRAG_bDP, 1, false },
    { "bDPr", "R", "Integer", K_R_FRr, UVC_FRAG_bDP, 1, false },
    { "BDPb", "2", "Integer", K_FR2_FR, UVC_FRAG_bDP, 1, false },
    { "BDPd", "2", "Integer", K_ZERO2, 0, 0, false },
    { "bTAf", "R", "Integer", K_R_FRf, UVC_FRAG_bTA, 1, false },
    { "bTAr", "R", "Integer", K_R_FRr, UVC_FRAG_bTA, 1, false },
    { "BTAb", "2", "Integer", K_FR2_FR, UVC_FRAG_bTA, 1, false },
    { "bTBf", "R", "Integer", K_R_FRf, UVC_FRAG_bTB, 1, false },
    { "bTBr", "R", "Integer", K_R_FRr, UVC_FRAG_bTB, 1, false },
    { "BTBb", "2", "Integer", K_FR2_FR, UVC_FRAG_bTB, 1, false },
    { "_CDP1", "1", "String", K_SEP, 0, 0, false },
    { "cDP1f", "R", "Integer", K_R_FAf, UVC_FAM_cDP1, 1, false },
    { "cDP1r", "R", "Integer", K_R_FAr, UVC_FAM_cDP1, 1, false },
    { "CDP1b", "2", "Integer", K_FR2_FA, UVC_FAM_cDP1, 1, false },
    { "CDP1d", "2", "Integer", K_NN2_FA, UVC_FAM_cDP1, 1, false },
    { "cDP12f", "R", "Integer", K_R_FAf, UVC_FAM_cDP12, 1, false },
    { "cDP12r", "R", "Integer", K_R_FAr, UVC_FAM_cDP12, 1, false },
    { "CDP12b", "2", "Integer", K_FR2_FA, UVC_FAM_cDP12, 1, false },
    { "_CDP2", "1", "String", K_SEP, 0, 0, false },
    { "cDP2f", "R", "Integer", K_R_FAf, UVC_FAM_cDP2, 1, false },
    { "cDP2r", "R", "Integer", K_R_FAr, UVC_FAM_cDP2, 1, false },
    { "CDP2b", "2", "Integer", K_FR2_FA, UVC_FAM_cDP2, 1, false },
    { "CDP2d", "2", "Integer", K_ZERO2, 0, 0, false },
    { "c2BQ2", "R", "Integer", K_R_FI, UVC_FI_c2BQ2, 1, true },
    { "C2BQ2", "1", "Integer", K_T1_FI, UVC_FI_c2BQ2, 1, true },
    { "c2LP0", "R", "Integer", K_R_FI, UVC_FI_c2LP0, 1, true },
    { "C2LP0", "1", "Integer", K_T1_FI, UVC_FI_c2LP0, 1, true },
    { "c2RP0", "R", "Integer", K_R_FI, UVC_FI_c2RP0, 1, true },
    { "C2RP0", "1", "Integer", K_T1_FI, UVC_FI_c2RP0, 1, true },
    { "_C2XP", "1", "String", K_SEP, 0, 0, true },
    { "c2LP1", "R", "Integer", K_R_FI, UVC_FI_c2LP1, 1, true },
    { "c2LP2", "R", "Integer", K_R_FI, UVC_FI_c2LP2, 1, true },
    { "c2RP1", "R", "Integer", K_R_FI, UVC_FI_c2RP1, 1, true },
    { "c2RP2", "R", "Integer", K_R_FI, UVC_FI_c2RP2, 1, true },
    { "c2LPL", "R", "Integer", K_R_FI, UVC_FI_c2LPL, 1, true },
    { "c2RPL", "R", "Integer", K_R_FI, UVC_FI_c2RPL, 1, true },
    { "_C2XB", "1", "String", K_SEP, 0, 0, true },
    { "c2LB1", "R", "Integer", K_R_FI, UVC_FI_c2LB1, 1, true },
    { "c2LB2", "R", "Integer", K_R_FI, UVC_FI_c2LB2, 1, true },
    { "c2RB1", "R", "Integer", K_R_FI, UVC_FI_c2RB1, 1, true },
    { "c2RB2", "R", "Integer", K_R_FI, UVC_FI_c2RB2, 1, true },
    { "c2LBL", "R", "Integer", K_R_FI64, UVC_FI64_c2LBL, 1, true },
    { "c2RBL", "R", "Integer", K_R_FI64, UVC_FI64_c2RBL, 1, true },
    { "_CDPx", "1", "String", K_SEP, 0, 0, true },
    { "cDP3f", "R", "Integer", K_R_FAf, UVC_FAM_cDP3, 1, true },
    { "cDP3r", "R", "Integer", K_R_FAr, UVC_FAM_cDP3, 1, true },
    { "CDP3b", "2", "Integer", K_FR2_FA, UVC_FAM_cDP3, 1, true },
    { "cDP21f", "R", "Integer", K_R_FAf, UVC_FAM_cDP21, 1, true },
    { "cDP21r", "R", "Integer", K_R_FAr, UVC_FAM_cDP21, 1, true },
    { "CDP21b", "2", "Integer", K_FR2_FA, UVC_FAM_cDP21, 1, true },
    { "_cDPMm", "1", "String", K_SEP, 0, 0, true },
    { "cDPMf", "R", "Integer", K_R_FAf, UVC_FAM_cDPM, 1, true },
    { "cDPMr", "R", "Integer", K_R_FAr, UVC_FAM_cDPM, 1, true },
    { "CDPMb", "2", "Integer", K_FR2_FA, UVC_FAM_cDPM, 1, true },
    { "cDPmf", "R", "Integer", K_R_FAf, UVC_FAM_cDPm, 1, true },
    { "cDPmr", "R", "Integer", K_R_FAr, UVC_FAM_cDPm, 1, true },
    { "CDPmb", "2", "Integer", K_FR2_FA, UVC_FAM_cDPm, 1, true },
    { "CDPDb", "2", "Integer", K_FR2_FA, UVC_FAM_cDPD, 1, false },
    { "cDPDf", "R", "Integer", K_R_FAf, UVC_FAM_cDPD, 1, false },
    { "cDPDr", "R", "Integer", K_R_FAr, UVC_FAM_cDPD, 1, false },
    { "_DDP", "1", "String", K_SEP, 0, 0, false },
    { "DDP1", "2", "Integer", K_T2_DU, UVC_DUPLEX_dDP1, 1, false },
    { "dDP1", "R", "Integer", K_R_DU, UVC_DUPLEX_dDP1, 1, false },
    { "DDP2", "2", "Integer", K_T2_DU, UVC_DUPLEX_dDP2, 1, false },
    { "dDP2", "R", "Integer", K_R_DU, UVC_DUPLEX_dDP2, 1, false },
    { "_ea", "1", "String", K_SEP, 0, 0, false },
    { "aBQ", "R", "Integer", K_R_REC, UVC_O_aBQ, 1, false },
    { "a2BQf", "R", "Integer", K_R_REC, UVC_O_a2BQf, 1, false },
    { "a2BQr", "R", "Integer", K_R_REC, UVC_O_a2BQr, 1, false },
    { "a2XM2", "R", "Integer", K_R_S32, UVC_S_a2XM2, 1, false },
    { "a2BM2", "R", "Integer", K_R_S32, UVC_S_a2BM2, 1, false },
    { "aBQQ", "R", "Integer", K_R_REC, UVC_O_aBQQ, 1, false },
    { "_eb", "1", "String", K_SEP, 0, 0, false },
    { "bMQ", "R", "Integer", K_R_REC, UVC_O_bMQ, 1, false },
    { "aAaMQ", "R", "Integer", K_R_REC, UVC_O_aAaMQ, 1, false },
    { "bNMQ", "R", "Integer", K_R_REC, UVC_O_bNMQ, 1, false },
    { "bNMa", "R", "Integer", K_R_REC, UVC_O_bNMa, 1, false },
    { "bNMb", "R", "Integer", K_R_REC, UVC_O_bNMb, 1, false },
    { "bMQQ", "R", "Integer", K_R_REC, UVC_O_bMQQ, 1, false },
    { "_eB", "1", "String", K_SEP, 0, 0, false },
    { "bIAQb", "R", "Integer", K_R_VQ, UVC_VQ_bIAQb, 1, false },
    { "bIADb", "R", "Integer", K_R_VQ, UVC_VQ_bIADb, 1, false },
    { "bIDQb", "R", "Integer", K_R_VQ, UVC_VQ_bIDQb, 1, false },
    { "_eC", "1", "String", K_SEP, 0, 0, false },
    { "cIAQf", "R", "Integer", K_R_VQ, UVC_VQ_cIAQf, 1, false },
    { "cIADf", "R", "Integer", K_R_VQ, UVC_VQ_cIADf, 1, false },
    { "cIDQf", "R", "Integer", K_R_VQ, UVC_VQ_cIDQf, 1, false },
    { "cIAQr", "R", "Integer", K_R_VQ, UVC_VQ_cIAQr, 1, false },
    { "cIADr", "R", "Integer", K_R_VQ, UVC_VQ_cIADr, 1, false },
    { "cIDQr", "R", "Integer", K_R_VQ, UVC_VQ_cIDQr, 1, false },
    { "_eE", "1", "String", K_SEP, 0, 0, false },
    { "bIAQ", "R", "Integer", K_R_REC, UVC_O_bIAQ, 1, false },
    { "cIAQ", "R", "Integer", K_R_REC, UVC_O_cIAQ, 1, false },
    { "bTINQ", "R", "Integer", K_R_REC, UVC_O_bTINQ, 1, false },
    { "cTINQ", "R", "Integer", K_R_REC, UVC_O_cTINQ, 1, false },
    { "_eQ1", "1", "String", K_SEP, 0, 0, false },
    { "cPCQ1", "R", "Integer", K_R_REC, UVC_O_cPCQ1, 1, false },
    { "cPLQ1", "R", "Integer", K_R_REC, UVC_O_cPLQ1, 1, false },
    { "cVQ1", "R", "Integer", K_R_REC, UVC_O_cVQ1, 1, false },
    { "gVQ1", "R", "Integer", K_R_REC, UVC_O_gVQ1, 1, false },
    { "_eQ2", "1", "String", K_SEP, 0, 0, false },
    { "cPCQ2", "R", "Integer", K_R_REC, UVC_O_cPCQ2, 1, false },
    { "cPLQ2", "R", "Integer", K_R_REC, UVC_O_cPLQ2, 1, false },
    { "cVQ2", "R", "Integer", K_R_REC, UVC_O_cVQ2, 1, false },
    { "cMmQ", "R", "Integer", K_R_REC, UVC_O_cMmQ, 1, false },
    { "dVQinc", "R", "Integer", K_R_REC, UVC_O_dVQinc, 1, false },
    { "_CDP1vx", "1", "String", K_SEP, 0, 0, false },
    { "cDP1v", "R", "Integer", K_R_REC, UVC_O_cDP1v, 1, false },
    { "CDP1v", "2", "Integer", K_N_REC, UVC_O_CDP1v0, 2, false },
    { "cDP1w", "R", "Integer", K_R_REC, UVC_O_cDP1w, 1, false },
    { "CDP1w", "1", "Integer", K_N_REC, UVC_O_CDP1w0, 1, false },
    { "cDP1x", "R", "Integer", K_R_REC, UVC_O_cDP1x, 1, false },
    { "CDP1x", "1", "Integer", K_N_REC, UVC_O_CDP1x0, 1, false },
    { "_CDP2vx", "1", "String", K_SEP, 0, 0, false },
    { "cDP2v", "R", "Integer", K_R_REC, UVC_O_cDP2v, 1, false },
    { "CDP2v", "2", "Integer", K_N_REC, UVC_O_CDP2v0, 2, false },
    { "cDP2w", "R", "Integer", K_R_REC, UVC_O_cDP2w, 1, false },
    { "CDP2w", "1", "Integer", K_N_REC, UVC_O_CDP2w0, 1, false },
    { "cDP2x", "R", "Integer", K_R_REC, UVC_O_cDP2x, 1, false },
    { "CDP2x", "1", "Integer", K_N_REC, UVC_O_CDP2x0, 1, false },
    { "_f1", "1", "String", K_SEP, 0, 0, false },
    { "CONTQ", "R", "Integer", K_R_REC, UVC_O_CONTQ, 1, false },
    { "nPF", ".", "Integer", K_N_REC, UVC_O_nPF0, 2, false },
    { "nNFA", ".", "Integer", K_N_REC, UVC_O_nNFA0, 6, false },
    { "nAFA", ".", "Integer", K_N_REC, UVC_O_nAFA0, 9, false },
    { "nBCFA", ".", "Integer", K_N_REC, UVC_O_nBCFA0, 10, false },
    { "_g1", "1", "String", K_SEP, 0, 0, false },
    { "VTI", "R", "Integer", K_SPECIAL, 0, 0, false },
    { "VTD", "R", "String", K_SPECIAL, 0, 0, false },
    { "cVQ1M", "2", "Integer", K_N_REC, UVC_O_cVQ1M0, 2, false },
    { "cVQ2M", "2", "Integer", K_N_REC, UVC_O_cVQ2M0, 2, false },
    { "cVQAM", "2", "String", K_SPECIAL, 0, 0, false },
    { "cVQSM", "2", "String", K_SPECIAL, 0, 0, false },
    { "_g2", "1", "String", K_SEP, 0, 0, false },
    { "gapNf", ".", "Integer", K_SPECIAL, 0, 0, false },
    { "gapNr", ".", "Integer", K_SPECIAL, 0, 0, false },
    { "gapSeq", ".", "String", K_SPECIAL, 0, 0, false },
    { "gapbAD1", ".", "Integer", K_SPECIAL, 0, 0, false },
    { "gapcAD1", ".", "Integer", K_SPECIAL, 0, 0, false },
    { "gc2AD", ".", "Integer", K_SPECIAL, 0, 0, false },
    { "gc2dAD", ".", "Integer", K_SPECIAL, 0, 0, false },
    { "_g3", "1", "String", K_SEP, 0, 0, false },
    { "bDPa", "R", "Integer", K_R_REC, UVC_O_bDPa, 1, false },
    { "cDP0a", "R", "Integer", K_R_REC, UVC_O_cDP0a, 1, false },
    { "gapSa", "R", "String", K_SPECIAL, 0, 0, false },
    { "_h1", "1", "String", K_SEP, 0, 0, false },
    { "bHap", "1", "String", K_SPECIAL, 0, 0, false },
    { "cHap", "1", "String", K_SPECIAL, 0, 0, false },
    { "c2Hap", "1", "String", K_SPECIAL, 0, 0, false },
    { "_i1", "1", "String", K_SEP, 0, 0, false },
    { "vHGQ", "1", "Integer", K_1_REC, UVC_O_vHGQ, 1, false },
    { "vAC", "2", "Integer", K_N_REC, UVC_O_vAC0, 2, false },
    { "vNLODQ", "2", "Integer", K_SPECIAL, 0, 0, false },
    { "note", "1", "String", K_SPECIAL, 0, 0, false },
};
const int N_TAGS = (int)(sizeof(TAGS) / sizeof(TAGS[0]));

// position-level lists (BcfFormat_symboltype_init, main.hpp:3897-3962): (group, plane) pairs
struct PosSrc { int g, p; };
const PosSrc POS_LISTS[10][12] = {
    { { UVC_F_PREP32, UVC_P_a_dp }, { UVC_F_PREP32, UVC_P_a_near_ins_dp }, { UVC_F_PREP32, UVC_P_a_near_del_dp }, { UVC_F_PREP32, UVC_P_a_near_RTR_ins_dp },
      { UVC_F_PREP32, UVC_P_a_near_RTR_del_dp }, { UVC_F_PREP32, UVC_P_a_pcr_dp }, { UVC_F_PREP32, UVC_P_a_snv_dp }, { UVC_F_PREP32, UVC_P_a_dnv_dp },
      { UVC_F_PREP32, UVC_P_a_highBQ_dp }, { UVC_F_PREP32, UVC_P_a_near_pcr_clip_dp }, { UVC_F_PREP32, UVC_P_a_near_long_clip_dp }, { UVC_F_PREP32, UVC_P_a_umi_dp } },
    { { UVC_F_PREP32, UVC_P_a_XM1500 }, { UVC_F_PREP32, UVC_P_a_GO1500 }, { UVC_F_PREP32, UVC_P_a_qlen }, { UVC_F_PREP32, UVC_P_a_GAPLEN },
      { UVC_F_PREP64, UVC_P_a_near_ins_pow2len }, { UVC_F_PREP64, UVC_P_a_near_del_pow2len }, { UVC_F_PREP32, UVC_P_a_near_ins_inv100len }, { UVC_F_PREP32, UVC_P_a_near_del_inv100len } },
    { { UVC_F_PREP64, UVC_P_a_near_ins_l_pow2len }, { UVC_F_PREP64, UVC_P_a_near_ins_r_pow2len }, { UVC_F_PREP64, UVC_P_a_near_del_l_pow2len }, { UVC_F_PREP64, UVC_P_a_near_del_r_pow2len } },
    { { UVC_F_PREP64, UVC_P_a_LI }, { UVC_F_PREP32, UVC_P_a_LIDP }, { UVC_F_PREP64, UVC_P_a_RI }, { UVC_F_PREP32, UVC_P_a_RIDP } },
    { { UVC_F_PREP32, UVC_P_a_l_dist_sum }, { UVC_F_PREP32, UVC_P_a_r_dist_sum }, { UVC_F_PREP32, UVC_P_a_inslen_sum }, { UVC_F_PREP32, UVC_P_a_dellen_sum } },
    { { UVC_F_THRES, UVC_T_aLPxT }, { UVC_F_THRES, UVC_T_aRPxT } },
    { { UVC_F_THRES, UVC_T_aLI1T }, { UVC_F_THRES, UVC_T_aLI2T }, { UVC_F_THRES, UVC_T_aRI1T }, { UVC_F_THRES, UVC_T_aRI2T } },
    { { UVC_F_THRES, UVC_T_aLI1t }, { UVC_F_THRES, UVC_T_aLI2t }, { UVC_F_THRES, UVC_T_aRI1t }, { UVC_F_THRES, UVC_T_aRI2t } },
    { { UVC_F_THRES, UVC_T_aLP1t }, { UVC_F_THRES, UVC_T_aLP2t }, { UVC_F_THRES, UVC_T_aRP1t }, { UVC_F_THRES, UVC_T_aRP2t } },
    { { UVC_F_THRES, UVC_T_aLB1t }, { UVC_F_THRES, UVC_T_aLB2t }, { UVC_F_THRES, UVC_T_aRB1t }, { UVC_F_THRES, UVC_T_aRB2t } },
};

std::string g_keys[2];
struct KeysInit { KeysInit() { for (int t2 = 0; t2 < 2; t2++) for (int i = 0; i < N_TAGS; i++) { if (TAGS[i].sscs && !t2) continue; if (!g_keys[t2].empty()) g_keys[t2] += ":"; g_keys[t2] += TAGS[i].name; } } } g_keys_init;

// one row of uvcgpu_region_fetch_columns
struct Cols {
    const int64_t *v; int base[UVC_NUM_FIELD_GROUPS];
    int64_t pos(int g, int p) const { return v[base[g] + p]; }
    int64_t sym(int g, int f, int s) const { return v[base[g] + f * NSYM + s]; }                      // [field][symbol]
    int64_t fr(int g, int nf, int sd, int f, int s) const { return v[base[g] + (sd * nf + f) * NSYM + s]; }   // [strand][field][symbol]
};
inline void st_symbols(int st, int &first, int &count, int &nn) { if (st == UVC_BASE_SYMBOL) { first = UVC_BASE_A; count = 6; nn = UVC_BASE_NN; } else { first = UVC_LINK_M; count = 8; nn = UVC_LINK_NN; } }

void put(std::string &o, int64_t v) {   // decimal text; snprintf was most of the record writer's time (two million integers per Mb)
    char b[24]; char *const e = b + sizeof(b); char *p = e;
    uint64_t u = (v < 0 ? 0 - (uint64_t)v : (uint64_t)v);
    do { *--p = (char)('0' + (int)(u % 10)); u /= 10; } while (u);
    if (v < 0) *--p = '-';
    o.append(p, (size_t)(e - p));
}
void put2(std::string &o, int64_t a, int64_t b) { put(o, a); o += ','; put(o, b); }

// UVCGPU_TIMING=1: where the record writer spends its time (stderr)
struct VcfTimer {
    bool on; std::chrono::steady_clock::time_point t;
    VcfTimer() : on(getenv("UVCGPU_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
    void lap(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[uvcgpu vcf_records] %-36s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

// indelpos_to_context, main.hpp:733-755
void repeat_context(const std::string &ref, int32_t at, int32_t smax, std::string &unit, int32_t &num) {
    num = 0; unit.clear();
    const int32_t n = (int32_t)ref.size();
    if (at >= n || at < 0) return;
    int32_t best_size = 0;
    for (int32_t u = 1; u <= smax; u++) {
        int32_t q = at;
        while (q + u < n && ref[(size_t)q] == ref[(size_t)(q + u)]) q++;
        const int32_t c = (q - at) / u + 1;
        bool better;   // is_indel_context_more_STR(u, c, best_size, num, smax), main.hpp:699-721
        if (best_size * num == 0) better = true;
        else if (u > smax || best_size > smax) better = (u < best_size || (u == best_size && c > num));
        else {
            int r1 = (c <= 1 ? (-c * u) : ((c - 1) * u)), r2 = (num <= 1 ? (-num * u) : ((num - 1) * best_size));
            if (0 == c) r1 = -100;
            if (0 == num || 0 == best_size) r2 = -100;
            better = r1 > r2;
        }
        if (better) { num = c; best_size = u; }
    }
    unit = ref.substr((size_t)at, (size_t)best_size);
}

struct Rows {   // uvcgpu_region_indel_alleles
    std::vector<UvcGapRow> rows; std::vector<uint8_t> seq;
    std::string text(const UvcGapRow &g, const std::string &ref, int32_t beg) const {
        std::string s;
        if (g.seq_off >= 0) { for (int i = 0; i < g.len; i++) s += "ACGTN"[seq[(size_t)g.seq_off + i] > 4 ? 4 : seq[(size_t)g.seq_off + i]]; }
        else { const int64_t a = (int64_t)g.refpos - beg; if (a >= 0 && a + g.len <= (int64_t)ref.size()) s = ref.substr((size_t)a, (size_t)g.len); }
        return s;
    }
    // the rows of (refpos, symbol): forward rows first, then reverse (main.cpp:858-869)
    void range(int32_t refpos, int32_t symbol, size_t &lo, size_t &hi) const {
        lo = std::lower_bound(rows.begin(), rows.end(), std::make_pair(refpos, symbol), [](const UvcGapRow &a, const std::pair<int32_t, int32_t> &k) {
            return a.refpos < k.first || (a.refpos == k.first && a.symbol < k.second); }) - rows.begin();
        hi = lo;
        while (hi < rows.size() && rows[hi].refpos == refpos && rows[hi].symbol == symbol) hi++;
    }
};
}   // namespace

extern "C" const char *uvcgpu_vcf_format_keys(int32_t with_tier2) { return g_keys[with_tier2 ? 1 : 0].c_str(); }

// generate_vcf_header, main.hpp:5778-5883.  The texts are the reference's (a VCF consumer sees them: "bcf_formats output unchanged"); the
// ##FILTER / ##FORMAT lines come from its own generator (uvc_vcf_header_lines.inc).  Where the reference prints facts about its run --
// ##fileDate, ##reference, ##variantCallerCommand -- the caller hands them in (uvcgpu_vcf_header_ex; NULL leaves the line out, which is
// what uvcgpu_vcf_header does); ##variantCallerVersion names this library, not uvc.
#include "uvc_vcf_header_lines.inc"
extern "C" int uvcgpu_vcf_header_ex(const UvcParams *P, const char *sample, const char *tumor_sample, const char *const *names, const int64_t *lens, int32_t n_contigs,
                                    const char *file_date, const char *reference_fname, const char *command_line, char *dst, int64_t cap, int64_t *len) {
    if (!P || !len || (n_contigs > 0 && (!names || !lens))) return uvcgpu_fail_(UVCGPU_EINVAL, "bad argument");
    std::string h = "##fileformat=VCFv4.2\n";
    if (file_date) h += std::string("##fileDate=") + file_date + "\n";
    if (reference_fname) h += std::string("##reference=") + reference_fname + "\n";
    for (int i = 0; i < n_contigs; i++) h += std::string("##contig=<ID=") + names[i] + ",length=" + std::to_string((long long)lens[i]) + ">\n";
    h += "##ALT=<ID=NON_REF,Description=\"Represents any possible alternative allele at this location, where POS (start position) is one-based inclusive. "
         "CAVEAT: this VCF line record is similar to a GVCF block but does not conform to the GVCF specifications. \">\n";
    for (const char *l : REF_FILTER_LINES) { h += l; h += '\n'; }
    h += "##INFO=<ID=ANY_VAR,Number=0,Type=Flag,Description=\"Any type of variant which may be caused by germline polymorphism and/or somatic mutation\">\n";
    h += "##INFO=<ID=GERMLINE,Number=0,Type=Flag,Description=\"germline variant\">\n";
    h += "##INFO=<ID=SOMATIC,Number=0,Type=Flag,Description=\"Somatic variant\">\n";
    h += "##INFO=<ID=MGVCF_BLOCK,Number=0,Type=Flag,Description=\"Multi-sample GVCF-like genomic regions consisting of " + std::to_string(1000 /* MGVCF_REGION_MAX_SIZE, common.hpp:44 */) + " consecutive positions. "
         "MGVCF is modified from GVCF to allow for easy comparison of sequencing depths of multiple samples at any arbitrary position. "
         "More detail is described in FORMAT/POS_VT_BDP_CDP_HomRefQ. \">\n";
    h += "##INFO=<ID=ADDITIONAL_INDEL_CANDIDATE,Number=0,Type=Flag,Description=\"Position with an abnormally high number of (soft/hard)-clipped sequences adjacent to this position (which can be caused by long InDel, copy-number variation (CNV), structural variation (SV), etc.) or with a high STR track length after it\">\n";
    h += "##INFO=<ID=SomaticQ,Number=A,Type=Float,Description=\"Somatic quality of the variant, the Phred-scaled odds that this variant is not somatic. "
         "CAVEAT: if only tumor bam file is provided, then this quality usually cannot reach 60 even with the help of a very big germline database because "
         "germline and somatic variants share similar characteristics in the tumor. "
         "Therefore, a matched normal is absolutely required to confidently determine the germline-vs-somatic origin of a biological variant. \">\n";
    h += "##INFO=<ID=TLODQ,Number=A,Type=Float,Description=\"Tumor log-of-data-likelihood quality, the Phred-scaled odds that this variant is not of biological origin (i.e., artifactual). \">\n";
    h += "##INFO=<ID=NLODQ,Number=A,Type=Float,Description=\"Normal log-of-data-likelihood quality, the Phred-scaled odds that this variant is of germline origin. \">\n";
    h += "##INFO=<ID=NLODV,Number=A,Type=String,Description=\"The variant symbol that minimizes NLODQ. \">\n";
    h += "##INFO=<ID=TNBQF,Number=4,Type=Float,Description=\"Binomial reward, power-law reward, systematic-error penalty, and normal-adjusted tumor variant quality computed using deduplicated read fragments. \">\n";
    h += "##INFO=<ID=TNCQF,Number=4,Type=Float,Description=\"Binomial reward, power-law reward, systematic-error penalty, and normal-adjusted tumor variant quality computed using consensus families of read fragments. \">\n";
    h += "##INFO=<ID=tbDP,Number=1,Type=Integer,Description=\"Tumor total non-deduped depth (deprecated, please see BDPb (previously named as BDPf and BDPr)). \">\n";
    h += "##INFO=<ID=tDP,Number=1,Type=Integer,Description=\"Tumor total deduped depth (deprecated, please see CDP1b (previously named as CDP1f and CDP1r)). \">\n";
    h += "##INFO=<ID=tAD,Number=R,Type=Integer,Description=\"Tumor deduped depth of each allele (deprecated, please see cDP1f and cDP1r). \">\n";
    h += "##INFO=<ID=t2DP,Number=1,Type=Integer,Description=\"Tumor total UMI-barcoded-family depth for duplex-rescued SSCS (CDP2b + DDP2 (previously used CDP2f and CDP2r)). \">\n";
    h += "##INFO=<ID=t2AD,Number=R,Type=Integer,Description=\"Tumor UMI-barcoded-family depth of each allele for duplex-rescued SSCS (cDP2b + dDP2 (previously used cDP2f and cDP2r)). \">\n";
    h += "##INFO=<ID=nDP,Number=1,Type=Integer,Description=\"Normal total deduped depth (deprecated, please see CDP1b (previously named as CDP1f and CDP1r)). \">\n";
    h += "##INFO=<ID=nAD,Number=R,Type=Integer,Description=\"Normal deduped depth of each allele (deprecated, please see cDP1f and cDP1r). \">\n";
    h += "##INFO=<ID=n2AD,Number=R,Type=Integer,Description=\"Normal UMI-barcoded-family depth of each allele (deprecated, please see cDP2f and cDP2r). \">\n";
    h += "##INFO=<ID=RU,Number=1,Type=String,Description=\"The shortest repeating unit in the reference\">\n";
    h += "##INFO=<ID=RC,Number=1,Type=Integer,Description=\"The number of non-interrupted RUs in the reference\">\n";
    h += "##INFO=<ID=R3X2,Number=6,Type=Integer,Description=\"Repeat start position, repeat track length, and repeat unit size at the two positions before and after this VCF position. \">\n";
    for (const char *l : REF_FORMAT_LINES) { h += l; h += '\n'; }
    h += "##FORMAT=<ID=GL4,Number=4,Type=Integer,Description=\"The four genotype likelihoods for 0/0, 0/1, 1/1, and 1/2\">\n";
    h += "##FORMAT=<ID=GST,Number=.,Type=Integer,Description=\"The genotype statistics\">\n";
    h += "##FORMAT=<ID=CDP1,Number=2,Type=Integer,Description=\"(CDP1f + CDP1r) for all alleles by sum and for the padded deletion allele\">\n";
    h += "##FORMAT=<ID=cDP1,Number=2,Type=Integer,Description=\"(cDP1f + cDP1r)\">\n";
    h += "##FORMAT=<ID=POS_VT_BDP_CDP_HomRefQ,Number=.,Type=Integer,Description=\"Summary of multiple GVCF regions in a line with INFO/MGVCF. "
         "This field conforms to the following regular expression: ((<pos>,<postype>,<.>,<dup>,<dedup>,<dedupBQ>,<homrefQ>,<.>)+<endpos>) "
         "where (x)+ means one or more occurrence of the expression x. "
         "The integer <pos> denotes position (coordinate on the reference sequence) that separates adjacent regions on the reference sequence. "
         "The integer <postype> denotes position type, where 1 and 2 mean SNV and InDel sub-positions, respectively. "
         "The missing integer represented by the dot symbol <.> is a sentinel value that delimits region separators (aka positions) and region information. "
         "The integer <dup> is the minimum non-deduplicated fragment depth of the region. "
         "The integer <dedup> is the minimum deduplicated fragment depth (with duplicated fragments counted only once). "
         "The integer <dedupBQ> is similar to <dedup> but is computed using only support with R1R2-adjusted BQ passing the threshold set by the command-line parameter --fam-thres-highBQ. "
         "The integer <homrefQ> is the minimum likelihood of the homozygous-reference (homref) genotype (GT) in this region. "
         "The integer <endpos> denotes the SNV ending sub-position of the set of regions on this VCF line, and <endpos> is the last number in this field. "
         "The (inclusive) begin position of the current region is the (exclusive) end position of the previous region. "
         "Each genomic position (e.g., chr1:99) is divided into (a) one SNV sub-position and (b) one InDel sub-position that is right after the SNV sub-position. "
         "The SNV prior of homref GT is used here. "
         "Thus, the actual InDel likelihood of homref GT is the one shown here plus "
         + std::to_string(P->germ_phred_hetero_indel - P->germ_phred_hetero_snp) + ". "
         "CAVEAT: HomRefQ is computed by a very fast but imprecise algorithm, so it is not as accurate as GQ. \">\n";
    h += std::string("##FORMAT=<ID=clipDP,Number=2,Type=Integer,Description=\"Total segment depth and segment depth with adjacent long clips "
         "(for the ") + "<ADDITIONAL_INDEL_CANDIDATE>" /* SYMBOL_TO_DESC_ARR[ADDITIONAL_INDEL_CANDIDATE_SYMBOL], main_conversion.hpp:345 */ + " symbolic ALT allele indicating that this position has a lot of long (soft/hard) clips nearby) or that this position is at the beginning of a long STR track\">\n";
    h += "##phasing=partial\n";
    h += std::string("##variantCallerVersion=") + uvcgpu_version() + " (MI355X-native implementation of the uvc 0.15.1 hot path)\n";
    if (command_line) h += std::string("##variantCallerCommand=") + command_line + "\n";
    const char *plat = (P->inferred_sequencing_platform == UVC_PLATFORM_ILLUMINA ? "Illumina/BGI" : P->inferred_sequencing_platform == UVC_PLATFORM_IONTORRENT ? "IonTorrent/LifeTechnologies/ThermoFisher"
                        : P->inferred_sequencing_platform == UVC_PLATFORM_OTHER ? "OtherSequencingPlatform" : "AUTO");   // SEQUENCING_PLATFORM_TO_NAME, common.cpp:26-32
    h += std::string("##variantCallerInferredParameters=(inferred_sequencing_platform=") + plat + ",central_readlen=" + std::to_string(P->central_readlen) + ")\n";
    h += std::string("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t") + (sample ? sample : "SAMPLE") + ((tumor_sample && *tumor_sample) ? std::string("\t") + tumor_sample : std::string()) + "\n";
    *len = (int64_t)h.size();
    if (!dst || cap < (int64_t)h.size()) return uvcgpu_fail_(UVCGPU_ENOMEM, "destination too small");
    memcpy(dst, h.data(), h.size());
    return 0;
}
extern "C" int uvcgpu_vcf_header(const UvcParams *P, const char *sample, const char *tumor_sample, const char *const *names, const int64_t *lens, int32_t n_contigs, char *dst, int64_t cap, int64_t *len) {
    return uvcgpu_vcf_header_ex(P, sample, tumor_sample, names, lens, n_contigs, nullptr, nullptr, nullptr, dst, cap, len);
}

extern "C" int uvcgpu_region_vcf_records(uvcgpu_region_t *r, const char *tname, const UvcScoreOut *scored, const UvcScoreRequest *req,
                                         char *dst, int64_t cap, int64_t *len) {
    if (!r || !tname || !len || !scored || (scored->n_records > 0 && !scored->fields) || scored->n_records > scored->capacity) return uvcgpu_fail_(UVCGPU_EINVAL, "bad argument");
    int32_t pos_beg = (req ? req->pos_beg : -1), pos_end = (req ? req->pos_end : -1);
    const bool base_at_beg = (req && req->base_at_pos_beg && pos_beg >= 0);
    const int32_t region_beg = (req ? req->region_beg : 0);
    const UvcTumorKey *tkeys = (req ? req->tumor_keys : nullptr); const int64_t n_tkeys = (req ? req->n_tumor_keys : 0);
    const char *const *tcols = (req && tkeys ? req->tumor_sample_columns : nullptr);
    const char *const *tref_alt = (req && tkeys ? req->tumor_ref_alt : nullptr);
    const int32_t *recs = scored->fields; const int64_t n = scored->n_records, stride = scored->capacity;
    { const int rc0 = uvcgpu_region_fetch_columns(r, nullptr, 0, nullptr); if (rc0) return rc0; }   // accumulated, planes not released
    const UvcParams &P = *uvcgpu_region_params(r);
    int32_t beg = 0, end = 0;
    const char *refp = uvcgpu_region_refseq(r, &beg, &end);
    const std::string ref(refp, (size_t)(end - beg));
    int64_t npos = 0;
    const int32_t *rtr = uvcgpu_region_repeat_tracks(r, &npos);
    if (!rtr) return uvcgpu_fail_(UVCGPU_EDEVICE, "vcf_records: the repeat tracks could not be fetched from the device");
    auto F = [&](int64_t i, int f) { return recs[(int64_t)f * stride + i]; };
    // the InDel string a rescued record of the normal sample takes from its tumor record: REF / ALT without their common head (main.cpp:867-880)
    auto rescued_text = [&](int64_t i) -> std::string {
        const int32_t tk0 = F(i, UVC_O_tkey);
        if (!(tref_alt && tkeys && tk0 >= 0 && tk0 < n_tkeys && tref_alt[tk0])) return std::string();
        const std::string ra = tref_alt[tk0];
        const size_t tab = ra.find('\t');
        if (tab == std::string::npos) return std::string();
        const std::string vr = ra.substr(0, tab), va = ra.substr(tab + 1);
        return vr.size() > va.size() ? vr.substr(va.size()) : (va.size() > vr.size() ? va.substr(vr.size()) : std::string());
    };
    // the records that are written, and the ref record of each
    std::vector<int64_t> kept, refrec;
    for (int64_t i = 0; i < n; i++) {
        if (!F(i, UVC_O_keep) || !F(i, UVC_O_out)) continue;
        const int st = is_base(F(i, UVC_O_symbol)) ? UVC_BASE_SYMBOL : UVC_LINK_SYMBOL;
        int64_t j = i, found = -1;
        while (j > 0 && F(j - 1, UVC_O_refpos) == F(i, UVC_O_refpos)) j--;
        for (; j < n && F(j, UVC_O_refpos) == F(i, UVC_O_refpos); j++)
            if ((is_base(F(j, UVC_O_symbol)) ? UVC_BASE_SYMBOL : UVC_LINK_SYMBOL) == st && F(j, UVC_O_symbol) == F(i, UVC_O_refsymbol)) { found = j; break; }
        if (found < 0) return uvcgpu_fail_(UVCGPU_EINVAL, "a kept record has no REF record at its position (pass all records of the score call)");
        kept.push_back(i); refrec.push_back(found);
    }
    std::vector<std::pair<int32_t, std::string>> rec_lines;   // (zerobased_pos of the iteration that writes it, line)
    const std::vector<UvcHapLinkHost> *hap = nullptr;
    VcfTimer timer;
    timer.lap("written records + their REF records");
    if (!kept.empty()) { hap = uvcgpu_region_hap_(r); if (!hap) return UVCGPU_EDEVICE; }
    timer.lap("haplotype links");
    if (!kept.empty()) {
        const int32_t ncol = uvcgpu_region_n_columns();
        std::vector<int32_t> where(kept.size());
        for (size_t k = 0; k < kept.size(); k++) where[k] = F(kept[k], UVC_O_refpos);
        std::vector<int64_t> cols((size_t)ncol * kept.size());
        int rc = uvcgpu_region_fetch_columns(r, where.data(), (int64_t)where.size(), cols.data());
        if (rc) return rc;
        timer.lap("plane columns of the written records");
        Rows G; int64_t nr = 0, nb = 0;
        rc = uvcgpu_region_indel_alleles(r, nullptr, 0, &nr, nullptr, 0, &nb);
        if (rc && rc != UVCGPU_ENOMEM) return rc;
        G.rows.resize((size_t)nr); G.seq.resize((size_t)nb + 1);
        if (nr) { rc = uvcgpu_region_indel_alleles(r, G.rows.data(), nr, &nr, G.seq.data(), nb + 1, &nb); if (rc) return rc; }
        timer.lap("InDel allele rows");
        Cols C; for (int g = 0; g < UVC_NUM_FIELD_GROUPS; g++) C.base[g] = uvcgpu_region_column_base(g);
        const bool tprov = (P.tumor_vcf_is_provided != 0);
        for (size_t k = 0; k < kept.size(); k++) {
            const int64_t ia = kept[k], ir = refrec[k];
            std::string out;
            C.v = cols.data() + (size_t)ncol * k;
            const int32_t refpos = F(ia, UVC_O_refpos), symbol = F(ia, UVC_O_symbol), refsymbol = F(ia, UVC_O_refsymbol);
            const int st = is_base(symbol) ? UVC_BASE_SYMBOL : UVC_LINK_SYMBOL;
            int s0, sn, nn; st_symbols(st, s0, sn, nn);
            const int64_t regionpos = (int64_t)refpos - beg;
            // the InDel string of this record
            std::string indel;
            const int32_t garow = F(ia, UVC_O_gapSa);
            if (garow >= 0 && garow < (int32_t)G.rows.size()) indel = G.text(G.rows[(size_t)garow], ref, beg);
            if (indel.empty() && (is_ins(symbol) || is_del(symbol))) indel = rescued_text(ia);
            // an InDel record without an allele row -- a symbol nobody carries, scored under -A -- has the symbol's description as its string
            // (indel_get_majority's fallback, main.hpp:5417-5424): in ALT and in gapSa.  (An InDel whose length came from the caller without text
            // -- UvcIndelAllele, or UvcTumorKey without tumor_ref_alt, a state the reference cannot be in -- is written with the symbolic ALT and
            // an empty gapSa.)
            if (indel.empty() && (is_ins(symbol) || is_del(symbol)) && F(ia, UVC_O_tkey) < 0 && !(req && req->n_indel_alleles > 0)) indel = SYMBOL_DESC[symbol];
            // CHROM POS ID REF ALT
            int64_t vcfpos; std::string vref, valt;
            auto ref_at = [&](int64_t p) { return (p >= 0 && p < (int64_t)ref.size()) ? std::string(1, ref[(size_t)p]) : std::string("n"); };
            if (!indel.empty()) {
                vcfpos = refpos; vref = (regionpos > 0 ? ref_at(regionpos - 1) : std::string("n")); valt = vref;
                if (indel[0] == '<') valt = indel; else if (is_ins(symbol)) valt += indel; else vref += indel;
            } else if (is_base(symbol)) { vcfpos = (int64_t)refpos + 1; vref = ref_at(regionpos); valt = SYMBOL_DESC[symbol]; }
            else { vcfpos = refpos; vref = (regionpos > 0 ? ref_at(regionpos - 1) : std::string("n")); valt = SYMBOL_DESC[symbol]; }
            float qual; { int32_t b = F(ia, UVC_O_QUAL); memcpy(&qual, &b, 4); }
            out += tname; out += '\t'; put(out, vcfpos); out += "\t.\t"; out += vref; out += '\t'; out += valt; out += '\t';
            out += std::to_string(qual); out += '\t'; out += FILTER_Q[std::min(std::max(F(ia, UVC_O_FILTER), 0), 6)]; out += '\t';
            // INFO (main.hpp:6206-6235)
            const int32_t tk = F(ia, UVC_O_tkey);
            const UvcTumorKey *T = (tprov && tkeys && tk >= 0 && tk < n_tkeys) ? &tkeys[tk] : nullptr;
            int64_t cdpd_all = 0, ddp2_all = 0;
            for (int s = s0; s < s0 + sn; s++) { cdpd_all += C.fr(UVC_F_FAM, UVC_NFAM, 0, UVC_FAM_cDPD, s) + C.fr(UVC_F_FAM, UVC_NFAM, 1, UVC_FAM_cDPD, s); ddp2_all += C.sym(UVC_F_DUPLEX, UVC_DUPLEX_dDP2, s); }
            const int64_t t2dp_own = (int32_t)cdpd_all + ((int32_t)ddp2_all + C.sym(UVC_F_DUPLEX, UVC_DUPLEX_dDP2, nn));   // SUMPAIR(CDPDb) + SUMPAIR(DDP2)
            auto adc = [&](int s) { return C.fr(UVC_F_FAM, UVC_NFAM, 0, UVC_FAM_cDPD, s) + C.fr(UVC_F_FAM, UVC_NFAM, 1, UVC_FAM_cDPD, s) + C.sym(UVC_F_DUPLEX, UVC_DUPLEX_dDP2, s); };
            size_t glo = 0, ghi = 0;
            if (is_ins(symbol) || is_del(symbol)) G.range(refpos, symbol, glo, ghi);
            int64_t alt_adc;
            if (is_ins(symbol) || is_del(symbol)) { alt_adc = 0; for (size_t g = glo; g < ghi; g++) if (G.text(G.rows[g], ref, beg) == indel) alt_adc += G.rows[g].c2dAD; }
            else alt_adc = adc(symbol);
            out += (T ? "SOMATIC" : "ANY_VAR");
            out += ";SomaticQ="; put(out, F(ia, UVC_O_SomaticQ)); out += ";TLODQ="; put(out, F(ia, UVC_O_TLODQ)); out += ";NLODQ="; put(out, F(ia, UVC_O_NLODQ));
            out += ";NLODV="; out += SYMBOL_DESC[std::min(std::max(F(ia, UVC_O_NLODV), 0), NSYM)];
            out += ";TNBQF="; for (int q = 0; q < 4; q++) { if (q) out += ','; put(out, F(ia, UVC_O_TNBQF0 + q)); }
            out += ";TNCQF="; for (int q = 0; q < 4; q++) { if (q) out += ','; put(out, F(ia, UVC_O_TNCQF0 + q)); }
            if (T) {
                out += ";tbDP="; put(out, T->BDP); out += ";tDP="; put(out, T->tDP); out += ";tAD="; put2(out, T->tAD0, T->tAD1);
                out += ";t2DP="; put(out, T->t2DP); out += ";t2AD="; put2(out, adc(refsymbol), alt_adc);   // fill_conditional_tki<false> stores the normal's counts in tADCR
                out += ";nDP="; put(out, F(ia, UVC_O_DP)); out += ";nAD="; put2(out, F(ir, UVC_O_AD), F(ia, UVC_O_AD)); out += ";n2AD=0,0";
            } else {
                out += ";tbDP="; put(out, F(ia, UVC_O_bDP)); out += ";tDP="; put(out, F(ia, UVC_O_DP)); out += ";tAD="; put2(out, F(ir, UVC_O_AD), F(ia, UVC_O_AD));
                out += ";t2DP="; put(out, t2dp_own); out += ";t2AD="; put2(out, adc(refsymbol), alt_adc);
            }
            std::string ru; int32_t rcn = 0;
            repeat_context(ref, (int32_t)(regionpos + (st == UVC_BASE_SYMBOL ? 1 : 0)), P.indel_str_repeatsize_max, ru, rcn);
            out += ";RU="; out += ru; out += ";RC="; put(out, rcn);
            {   // R3X2: the repeat tracks indel_adj_tracklen_dist before and behind (main.hpp:6100-6103)
                const int64_t d = P.indel_adj_tracklen_dist;
                const int64_t i1 = std::max(regionpos, d) - d, i2 = std::min(regionpos + d, npos - d);   // region_repeatvec has npos entries (refstring2repeatvec repeats the last one)
                auto tr = [&](int f, int64_t i) { return (i >= 0 && i < npos) ? rtr[(size_t)f * npos + i] : 0; };
                const int32_t l1 = tr(UVC_RTR_tracklen, i1), l2 = tr(UVC_RTR_tracklen, i2);
                out += ";R3X2="; put(out, l1 ? beg + tr(UVC_RTR_begpos, i1) : 0); out += ','; put(out, l1); out += ','; put(out, tr(UVC_RTR_unitlen, i1)); out += ',';
                put(out, l2 ? beg + tr(UVC_RTR_begpos, i2) : 0); out += ','; put(out, l2); out += ','; put(out, tr(UVC_RTR_unitlen, i2));
            }
            const bool tier2 = F(ia, UVC_O_tier2) != 0;
            out += '\t'; out += g_keys[tier2 ? 1 : 0]; out += '\t';
            // the sample column (streamAppendBcfFormat)
            bool first = true;
            for (int t = 0; t < N_TAGS; t++) {
                const Tag &tg = TAGS[t];
                if (tg.sscs && !tier2) continue;
                if (!first) out += ':';
                first = false;
                auto sumst = [&](auto fn) { int64_t acc = 0; for (int s = s0; s < s0 + sn; s++) acc += fn(s); return acc; };
                switch (tg.kind) {
                case K_SEP: out += tg.name; break;
                case K_R_S32: put2(out, C.sym(UVC_F_SEG32, tg.a, refsymbol), C.sym(UVC_F_SEG32, tg.a, symbol)); break;
                case K_R_S64: put2(out, C.sym(UVC_F_SEG64, tg.a, refsymbol), C.sym(UVC_F_SEG64, tg.a, symbol)); break;
                case K_R_VQ: put2(out, C.sym(UVC_F_VQ, tg.a, refsymbol), C.sym(UVC_F_VQ, tg.a, symbol)); break;
                case K_R_FRf: put2(out, C.fr(UVC_F_FRAG, UVC_NFRAG, 0, tg.a, refsymbol), C.fr(UVC_F_FRAG, UVC_NFRAG, 0, tg.a, symbol)); break;
                case K_R_FRr: put2(out, C.fr(UVC_F_FRAG, UVC_NFRAG, 1, tg.a, refsymbol), C.fr(UVC_F_FRAG, UVC_NFRAG, 1, tg.a, symbol)); break;
                case K_R_FAf: put2(out, C.fr(UVC_F_FAM, UVC_NFAM, 0, tg.a, refsymbol), C.fr(UVC_F_FAM, UVC_NFAM, 0, tg.a, symbol)); break;
                case K_R_FAr: put2(out, C.fr(UVC_F_FAM, UVC_NFAM, 1, tg.a, refsymbol), C.fr(UVC_F_FAM, UVC_NFAM, 1, tg.a, symbol)); break;
                case K_R_FI: put2(out, C.sym(UVC_F_FAMINFO32, tg.a, refsymbol), C.sym(UVC_F_FAMINFO32, tg.a, symbol)); break;
                case K_R_FI64: put2(out, C.sym(UVC_F_FAMINFO64, tg.a, refsymbol), C.sym(UVC_F_FAMINFO64, tg.a, symbol)); break;
                case K_R_DU: put2(out, C.sym(UVC_F_DUPLEX, tg.a, refsymbol), C.sym(UVC_F_DUPLEX, tg.a, symbol)); break;
                case K_T1_S32: put(out, (int32_t)sumst([&](int s) { return C.sym(UVC_F_SEG32, tg.a, s); })); break;
                case K_T1L_S32: put(out, sumst([&](int s) { return C.sym(UVC_F_SEG32, tg.a, s); })); break;
                case K_T1_S64: put(out, sumst([&](int s) { return C.sym(UVC_F_SEG64, tg.a, s); })); break;
                case K_T1_VQ: put(out, (int32_t)sumst([&](int s) { return C.sym(UVC_F_VQ, tg.a, s); })); break;
                case K_T1_FI: put(out, (int32_t)sumst([&](int s) { return C.sym(UVC_F_FAMINFO32, tg.a, s); })); break;
                case K_T2_S32: put2(out, (int32_t)sumst([&](int s) { return C.sym(UVC_F_SEG32, tg.a, s); }), C.sym(UVC_F_SEG32, tg.a, nn)); break;
                case K_T2_DU: put2(out, (int32_t)sumst([&](int s) { return C.sym(UVC_F_DUPLEX, tg.a, s); }), C.sym(UVC_F_DUPLEX, tg.a, nn)); break;
                case K_FR2_FR: put2(out, (int32_t)sumst([&](int s) { return C.fr(UVC_F_FRAG, UVC_NFRAG, 0, tg.a, s); }), (int32_t)sumst([&](int s) { return C.fr(UVC_F_FRAG, UVC_NFRAG, 1, tg.a, s); })); break;
                case K_FR2_FA: put2(out, (int32_t)sumst([&](int s) { return C.fr(UVC_F_FAM, UVC_NFAM, 0, tg.a, s); }), (int32_t)sumst([&](int s) { return C.fr(UVC_F_FAM, UVC_NFAM, 1, tg.a, s); })); break;
                case K_NN2_FA: put2(out, C.fr(UVC_F_FAM, UVC_NFAM, 0, tg.a, nn), C.fr(UVC_F_FAM, UVC_NFAM, 0, tg.a, nn)); break;   // fill_symboltype_nn_fmt reads strand 0 twice
                case K_ZERO2: out += "0,0"; break;
                case K_POS: for (int q = 0; q < tg.n; q++) { if (q) out += ','; const PosSrc &ps = POS_LISTS[tg.a][q]; int64_t v = C.pos(ps.g, ps.p); if (!strcmp(tg.name, "APDP") || !strcmp(tg.name, "APLRP")) v = (int32_t)v; put(out, v); } break;
                case K_R_REC: put2(out, F(ir, tg.a), F(ia, tg.a)); break;
                case K_1_REC: put(out, F(ia, tg.a)); break;
                case K_N_REC: for (int q = 0; q < tg.n; q++) { if (q) out += ','; put(out, F(ia, tg.a + q)); } break;
                case K_SPECIAL: {
                    const std::string nm = tg.name;
                    if (nm == "GT") out += "./1";
                    else if (nm == "GQ") out += "0";
                    else if (nm == "HQ") out += "0,0";
                    else if (nm == "bHap" || nm == "cHap" || nm == "c2Hap") {   // main.hpp:4242-4244
                        const std::string ph = uvc_hap_phase_string(hap[nm == "bHap" ? 0 : nm == "cHap" ? 1 : 2], refpos, symbol);
                        out += (ph.empty() ? std::string(".") : ph);
                    }
                    else if (nm == "FT" || nm == "note") out += ".";
                    else if (nm == "FTS") {
                        // fmt_bias_push appends "<name>-<round(100 * biasFA / refFA)>" for each bias that fired
                        const uint32_t bits = (uint32_t)F(ia, UVC_O_FTS);
                        std::string s;
                        for (int b = 0; b < 19; b++) if (bits & (1u << b)) {
                            const uint32_t pct = ((uint32_t)F(ia, UVC_O_FTSpct0 + b / 4) >> (8 * (b % 4))) & 0xFF;
                            if (!s.empty()) s += '|';
                            s += FTS_NAMES[b]; s += '-'; s += std::to_string(pct);
                        }
                        out += (s.empty() ? std::string("PASS") : s);
                    }
                    else if (nm == "VTI") put2(out, refsymbol, symbol);
                    else if (nm == "VTD") { out += SYMBOL_DESC[refsymbol]; out += ','; out += SYMBOL_DESC[symbol]; }
                    else if (nm == "cVQAM") { out += SYMBOL_DESC[std::min(std::max(F(ia, UVC_O_cVQAM0), 0), NSYM)]; out += ','; out += SYMBOL_DESC[std::min(std::max(F(ia, UVC_O_cVQAM1), 0), NSYM)]; }
                    else if (nm == "cVQSM") { for (int q = 0; q < 2; q++) { if (q) out += ','; const int32_t row = F(ia, UVC_O_cVQSM0 + q); if (row >= 0 && row < (int32_t)G.rows.size()) out += G.text(G.rows[(size_t)row], ref, beg); } }
                    else if (nm == "gapNf" || nm == "gapNr") {
                        const int sd = (nm == "gapNf" ? 0 : 1); int cnt = 0;
                        for (size_t g = glo; g < ghi; g++) if (G.rows[g].strand == sd) cnt++;
                        // the count is pushed when the symbol has fragments on the strand (main.cpp:859-869), even if none of them kept an allele string
                        const bool pushed = (is_ins(symbol) || is_del(symbol)) && C.fr(UVC_F_FRAG, UVC_NFRAG, sd, UVC_FRAG_bDP, symbol) > 0;
                        if (pushed) put(out, cnt); else out += ".";
                    }
                    else if (nm == "gapSeq" || nm == "gapbAD1" || nm == "gapcAD1" || nm == "gc2AD" || nm == "gc2dAD") {
                        if (glo == ghi) out += ".";
                        for (size_t g = glo; g < ghi; g++) {
                            if (g > glo) out += ',';
                            const UvcGapRow &gr = G.rows[g];
                            if (nm == "gapSeq") out += G.text(gr, ref, beg);
                            else put(out, nm == "gapbAD1" ? gr.bAD1 : nm == "gapcAD1" ? gr.cAD1 : nm == "gc2AD" ? gr.c2AD : gr.c2dAD);
                        }
                    }
                    else if (nm == "gapSa") { out += ','; out += indel; }   // the REF allele has no string
                    else if (nm == "vNLODQ") { if (st == UVC_BASE_SYMBOL) put2(out, F(ia, UVC_O_vNLODQ), 0); else put2(out, 0, F(ia, UVC_O_vNLODQ)); }
                    break; }
                }
            }
            if (T && tcols && tcols[tk]) { out += '\t'; out += tcols[tk]; }   // is_tumor_format_retrieved: bcf1_to_string of the tumor record, main.hpp:6269
            out += '\n';
            rec_lines.emplace_back(F(ia, UVC_O_refpos) + (is_base(F(ia, UVC_O_symbol)) ? 1 : 0), std::move(out));
        }
    }
    // ---- GERMLINE lines (OUTVAR_GERMLINE, output_germline: main.hpp:5612-5775): one per (zerobased_pos, symbol type) whose genotype call is
    // written, in front of the records of the position (main.cpp:1049-1066 runs before the record loop :1073-1168), BASE before LINK ----
    std::vector<std::pair<int32_t, std::string>> germ_lines;
    if (P.outvar_flag & 0x1) {
        std::vector<int64_t> heads;   // first record of every group with a written genotype
        for (int64_t i = 0; i < n; i++) {
            const bool bs = is_base(F(i, UVC_O_symbol));
            const bool head = (i == 0 || F(i - 1, UVC_O_refpos) != F(i, UVC_O_refpos) || is_base(F(i - 1, UVC_O_symbol)) != bs);
            if (head && F(i, UVC_O_germ_emit)) heads.push_back(i);
        }
        if (!heads.empty()) {
            const int32_t ncol = uvcgpu_region_n_columns();
            std::vector<int32_t> where(heads.size());
            for (size_t k = 0; k < heads.size(); k++) where[k] = F(heads[k], UVC_O_refpos);
            std::vector<int64_t> cols((size_t)ncol * heads.size());
            int rc = uvcgpu_region_fetch_columns(r, where.data(), (int64_t)where.size(), cols.data());
            if (rc) return rc;
            Rows G; int64_t nr = 0, nb = 0;
            rc = uvcgpu_region_indel_alleles(r, nullptr, 0, &nr, nullptr, 0, &nb);
            if (rc && rc != UVCGPU_ENOMEM) return rc;
            G.rows.resize((size_t)nr); G.seq.resize((size_t)nb + 1);
            if (nr) { rc = uvcgpu_region_indel_alleles(r, G.rows.data(), nr, &nr, G.seq.data(), nb + 1, &nb); if (rc) return rc; }
            Cols C; for (int g = 0; g < UVC_NUM_FIELD_GROUPS; g++) C.base[g] = uvcgpu_region_column_base(g);
            static const char *const GT4[4] = { "0/0", "0/1", "1/1", "1/2" };
            for (size_t k = 0; k < heads.size(); k++) {
                const int64_t i0 = heads[k];
                C.v = cols.data() + (size_t)ncol * k;
                const int32_t refpos = F(i0, UVC_O_refpos), refsymbol = F(i0, UVC_O_refsymbol);
                const bool subst = is_base(refsymbol);
                int64_t i1 = i0; while (i1 < n && F(i1, UVC_O_refpos) == refpos && is_base(F(i1, UVC_O_symbol)) == subst) i1++;
                const int64_t regionpos = (int64_t)refpos - beg;
                const int GLidx = F(i0, UVC_O_germ_GT);
                const int32_t sel[3] = { F(i0, UVC_O_germ_ref), F(i0, UVC_O_germ_alt1), F(i0, UVC_O_germ_alt2) };
                auto sym_of = [&](int32_t rec) { return rec >= 0 ? F(rec, UVC_O_symbol) : NSYM; };   // the padding allele: END_ALIGNMENT_SYMBOLS, "<NONE>"
                auto cdp0a = [&](int32_t rec) { return rec >= 0 ? F(rec, UVC_O_cDP0a) : 0; };
                // the q-th allele string of (refpos, symbol) in indel_get_majority order = the order of its records
                auto allele_text = [&](int symbol, int q) -> std::string {
                    int seen = 0;
                    for (int64_t j = i0; j < i1; j++) if (F(j, UVC_O_symbol) == symbol) {
                        if (seen++ == q) {
                            const int32_t row = F(j, UVC_O_gapSa);
                            if (row >= 0 && row < (int32_t)G.rows.size()) return G.text(G.rows[(size_t)row], ref, beg);
                            const std::string rt = rescued_text(j);   // LAST(fmt.gapSa) of a rescued record is the tumor record's string
                            return rt.empty() ? std::string(SYMBOL_DESC[symbol]) : rt;
                        }
                    }
                    return std::string();
                };
                auto ref_at = [&](int64_t q) { return (q >= 0 && q < (int64_t)ref.size()) ? std::string(1, ref[(size_t)q]) : std::string("n"); };
                const int s0 = sym_of(sel[0]), s1 = sym_of(sel[1]), s2 = sym_of(sel[2]);
                std::string vref, valt;
                if (subst) {
                    vref = SYMBOL_DESC[std::min(s0, NSYM)]; valt = SYMBOL_DESC[std::min(s1, NSYM)];
                    if (3 == GLidx) { valt += ","; valt += SYMBOL_DESC[std::min(s2, NSYM)]; }
                } else {
                    const std::string vref1 = (regionpos > 0 ? ref_at(regionpos - 1) : std::string("n"));
                    const std::string str1 = (s1 < NSYM ? allele_text(s1, 0) : std::string());
                    vref = vref1;
                    if (3 != GLidx) {
                        if (str1.empty() || str1[0] == '<') valt = SYMBOL_DESC[std::min(s1, NSYM)];
                        else { valt = vref; if (is_ins(s1)) valt += str1; else if (is_del(s1)) vref += str1; else valt = SYMBOL_DESC[std::min(s1, NSYM)]; }
                    } else {
                        const std::string str2 = (s2 < NSYM ? allele_text(s2, s2 == s1 ? 1 : 0) : std::string());
                        valt = vref1;
                        if (str1.empty() || str1[0] == '<' || str2.empty() || str2[0] == '<') valt = std::string(SYMBOL_DESC[std::min(s1, NSYM)]) + "," + SYMBOL_DESC[std::min(s2, NSYM)];
                        else if (is_ins(s1) && is_ins(s2)) valt = vref1 + str1 + "," + vref1 + str2;
                        else if (is_del(s1) && is_del(s2)) {
                            if (str1.size() > str2.size()) { vref = vref1 + str1; valt = vref1 + "," + vref1 + str1.substr(str2.size()); }
                            else { vref = vref1 + str2; valt = vref1 + str2.substr(str1.size()) + "," + vref1; }
                        }
                        else if (is_ins(s1) && is_del(s2)) { valt = vref1 + str1 + str2 + "," + vref1; vref = vref1 + str2; }
                        else if (is_del(s1) && is_ins(s2)) { valt = vref1 + "," + vref1 + str2 + str1; vref = vref1 + str1; }
                        else valt = std::string(SYMBOL_DESC[std::min(s1, NSYM)]) + "," + SYMBOL_DESC[std::min(s2, NSYM)];
                    }
                }
                const int nn = (subst ? UVC_BASE_NN : UVC_LINK_NN);
                const int64_t cdp1d = 2 * C.fr(UVC_F_FAM, UVC_NFAM, 0, UVC_FAM_cDP1, nn);   // SUMPAIR(CDP1d): fill_symboltype_nn_fmt pushes the forward value twice (main.hpp:3774-3786)
                std::string line = tname; line += '\t'; put(line, (int64_t)refpos + (subst ? 1 : 0)); line += "\t.\t"; line += vref; line += '\t'; line += valt; line += '\t';
                put(line, F(i0, UVC_O_germ_GQ)); line += "\tPASS\tGERMLINE\tGT:GQ:HQ:FT:CDP1:cDP1:GL4:GST:note\t";
                line += GT4[std::min(std::max(GLidx, 0), 3)]; line += ':'; put(line, F(i0, UVC_O_germ_GQ)); line += ":0,0:PASS:";
                put2(line, F(i0, UVC_O_DP), cdp1d); line += ':';
                put2(line, cdp0a(sel[0]), cdp0a(sel[1])); if (3 == GLidx) { line += ','; put(line, cdp0a(sel[2])); }
                line += ':';
                for (int q = 0; q < 4; q++) { if (q) line += ','; put(line, F(i0, UVC_O_GL4_0 + q)); }
                line += ':';
                for (int q = 0; q < 8; q++) { if (q) line += ','; put(line, F(i0, UVC_O_GST0 + q)); }
                line += ":\n";   // FORMAT/note of the reference allele: empty unless should_add_note
                germ_lines.emplace_back(refpos + (subst ? 1 : 0), std::move(line));
            }
        }
    }
    std::string out;
    timer.lap("record + GERMLINE lines");
// ---- the position-level lines, in front of the records of their zerobased_pos (main.cpp:607-799) ----
    {
        if (pos_beg < 0) { pos_beg = beg + 1; pos_end = end; }   // the default range of uvcgpu_region_score
        const bool want_block = (P.outvar_flag & 0x8) != 0, want_cand = (P.outvar_flag & 0x10) != 0;
        std::vector<std::pair<int32_t, std::string>> pos_lines;
        if ((want_block || want_cand) && pos_end > pos_beg + (base_at_beg ? 0 : 1)) {
            const int32_t state_end = end + 1;                                   // getUnifiedExcluEndPosition (main.cpp:569)
            const int32_t s_beg = pos_beg - 1, s_end = std::min<int64_t>((int64_t)pos_end - 1 + 1001, state_end);
            std::vector<int32_t> st((size_t)10 * (size_t)std::max(0, s_end - s_beg));
            if (s_end > s_beg) { const int rc = uvcgpu_region_block_stats_(r, s_beg, s_end, st.data()); if (rc) return rc; }
            timer.lap("block statistics (kernel + D2H)");
            auto S = [&](int32_t refpos, int q) { return st[(size_t)10 * (size_t)(refpos - s_beg) + (size_t)q]; };
            auto refchar = [&](int64_t off) { return (off >= 0 && off < (int64_t)ref.size()) ? ref[(size_t)off] : 'N'; };
            auto code_of = [](char c) { switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; case 'I': case 'i': return 6; case '-': case '_': return 9; default: return 4; } };
            // normal sample with the tumor's FORMAT carried over: the tumor's own line of this position, if it has exactly one
            const bool tcols_pos = (P.tumor_vcf_is_provided && tcols != nullptr);
            auto tumor_column = [&](int32_t refpos, int32_t symbol, const char *if_many, const char *if_none) -> std::string {
                int64_t lo = 0, hi = n_tkeys;
                while (lo < hi) { const int64_t mid = (lo + hi) >> 1; const UvcTumorKey &a = tkeys[mid]; if (a.refpos < refpos || (a.refpos == refpos && a.symbol < symbol)) lo = mid + 1; else hi = mid; }
                int64_t e = lo; while (e < n_tkeys && tkeys[e].refpos == refpos && tkeys[e].symbol == symbol) e++;
                if (e == lo) return if_none;
                if (e - lo != 1 || !tcols[lo]) return if_many;
                return std::string("\t") + tcols[lo];
            };
            // indelpos_to_context at every position of the range: instead of six scans per position, one backward pass per unit size u
            // (run = how many q' >= q in a row have ref[q'] == ref[q' + u]; the repeat count of u at q is run / u + 1) that keeps the
            // better (unit size, count) per position as is_indel_context_more_STR ranks them (main.hpp:699-721)
            const int32_t smax = std::max(0, std::min(P.indel_str_repeatsize_max, 64));
            const int64_t c_lo = std::max<int64_t>((int64_t)pos_beg - 1 - beg, 0), c_hi = std::min<int64_t>((int64_t)pos_end - beg, (int64_t)ref.size());   // offsets asked for
            std::vector<int32_t> ctx_size, ctx_num;
            if (want_cand && c_hi > c_lo) {
                ctx_size.assign((size_t)(c_hi - c_lo), 0); ctx_num.assign((size_t)(c_hi - c_lo), 0);
                const int64_t n = (int64_t)ref.size();
                const char *rs = ref.data();
                for (int32_t u = 1; u <= smax; u++) {
                    int32_t carry = 0; { int64_t q = c_hi; while (q + u < n && rs[q] == rs[q + u]) { q++; carry++; } }   // the run at c_hi - 1 may reach past the range
                    for (int64_t q = c_hi - 1; q >= c_lo; q--) {
                        carry = (q + u < n && rs[q] == rs[q + u]) ? carry + 1 : 0;
                        const int32_t c = carry / u + 1;
                        int32_t &bs = ctx_size[(size_t)(q - c_lo)], &num = ctx_num[(size_t)(q - c_lo)];
                        bool better;
                        if (bs * num == 0) better = true;
                        else { const int r1 = (c <= 1 ? (-c * u) : ((c - 1) * u)), r2 = (num <= 1 ? (-num * u) : ((num - 1) * bs)); better = r1 > r2; }
                        if (better) { num = c; bs = u; }
                    }
                }
            }
            auto context_at = [&](int64_t at, int32_t &best_size, int32_t &num) {
                if (at < c_lo || at >= c_hi || ctx_size.empty()) { best_size = 0; num = 0; return; }
                best_size = ctx_size[(size_t)(at - c_lo)]; num = ctx_num[(size_t)(at - c_lo)];
            };
            int32_t prev_track = 0;
            if (base_at_beg && want_cand) { int32_t bs0 = 0, rcn0 = 0; context_at((int64_t)pos_beg - 1 - beg, bs0, rcn0); prev_track = rcn0 * bs0; }   // the track of the zerobased_pos in front, which the adjacent region iterated
            for (int32_t z = pos_beg; z < pos_end; z++) {
                int32_t best_size = 0, rcn = 0;
                if (want_cand) context_at((int64_t)z - beg, best_size, rcn);
                const int32_t curr_track = rcn * best_size;
                if (z != pos_beg || base_at_beg) {
                    const int32_t refpos = z - 1;
                    std::string line;
                    if (want_block && ((refpos % 1000) == 0 || refpos == region_beg)) {   // main.cpp:655-656: refpos == incluBegPosition
                        // runs of similar depth and hom-ref quality over the next <= 1001 positions, LINK then BASE sub-position
                        const int32_t rp2end = std::min<int64_t>((int64_t)refpos + 1001, state_end);
                        int32_t pb = 0, pc = 0, p12 = 0, pq = INT32_MAX / 2 + 1; const int32_t init_q = INT32_MAX / 2 + 1;
                        auto differ = [](int32_t a, int32_t b) { const int32_t lo = std::min(a, b), hi = std::max(a, b); return !((int64_t)lo * 130 >= (int64_t)hi * 100) && !(lo + 3 >= hi); };
                        std::string body;
                        for (int32_t rp2 = refpos; rp2 < rp2end; rp2++) for (int t = 0; t < 2; t++) {
                            const int32_t b = S(rp2, t * 4), c = S(rp2, t * 4 + 1), c12 = S(rp2, t * 4 + 2), q = S(rp2, t * 4 + 3);
                            if (pq == init_q || std::abs((int64_t)q - pq) > 10 || differ(b, pb) || differ(c, pc) || differ(c12, p12)) {
                                put(body, rp2 + (t == 1 ? 1 : 0)); body += ','; put(body, 1 + (t == 0 ? UVC_LINK_SYMBOL : UVC_BASE_SYMBOL)); body += ",.,";
                                put(body, b); body += ','; put(body, c); body += ','; put(body, c12); body += ','; put(body, q); body += ",.,";
                                pb = b; pc = c; p12 = c12; pq = q;
                            }
                        }
                        const char rc1 = refchar((int64_t)refpos - beg);
                        line += tname; line += '\t'; put(line, (int64_t)refpos + 1); line += "\t.\t"; line += rc1; line += "\t<NON_REF>\t.\t.\tMGVCF_BLOCK\tGT:VTI:POS_VT_BDP_CDP_HomRefQ\t.:";
                        put(line, code_of(rc1)); line += ",15:"; line += body; put(line, rp2end);
                        if (tcols_pos) line += tumor_column(refpos, UVC_MGVCF_SYMBOL, "\t.:.,.:-1", "\t.:.,.:.");   // main.cpp:739-757
                        line += '\n';
                    }
                    if (want_cand) {
                        const int32_t ADP = S(refpos, 8), aCDP = S(refpos, 9);
                        const bool long_track = (curr_track > std::max(P.microadjust_alignment_tracklen_min - 1, prev_track));
                        const bool clip_region = (aCDP >= P.microadjust_alignment_clip_min_count) && (aCDP >= ADP * (P.microadjust_alignment_clip_min_frac - 2.220446049250313e-16));
                        if ((long_track || clip_region) && ADP >= 2 * P.microadjust_alignment_clip_min_count) {
                            const char rc1 = refchar((int64_t)refpos - beg);
                            line += tname; line += '\t'; put(line, (int64_t)refpos + 1); line += "\t.\t"; line += rc1; line += "\t<ADDITIONAL_INDEL_CANDIDATE>\t.\t.\tADDITIONAL_INDEL_CANDIDATE;RU=";
                            if ((int64_t)z - beg < (int64_t)ref.size()) line += ref.substr((size_t)((int64_t)z - beg), (size_t)best_size);
                            line += ";RC="; put(line, rcn); line += "\tGT:VTI:clipDP\t.:"; put(line, code_of(rc1)); line += ",16:"; put(line, ADP); line += ','; put(line, aCDP);
                            if (tcols_pos) line += tumor_column(refpos, UVC_ADDITIONAL_INDEL_CANDIDATE_SYMBOL, "\t.:-1,-1:-1,-1", "\t.:.,.:.,.");   // main.cpp:784-798
                            line += '\n';
                        }
                    }
                    if (!line.empty()) pos_lines.emplace_back(z, std::move(line));
                }
                prev_track = curr_track;
            }
        }
        timer.lap("MGVCF / indel-candidate lines");
        size_t a = 0, b = 0, g = 0;   // at one zerobased_pos: block / candidate lines, GERMLINE lines, records
        while (a < pos_lines.size() || b < rec_lines.size() || g < germ_lines.size()) {
            const int32_t za = (a < pos_lines.size() ? pos_lines[a].first : INT32_MAX), zg = (g < germ_lines.size() ? germ_lines[g].first : INT32_MAX), zb = (b < rec_lines.size() ? rec_lines[b].first : INT32_MAX);
            if (za <= zg && za <= zb) out += pos_lines[a++].second;
            else if (zg <= zb) out += germ_lines[g++].second;
            else out += rec_lines[b++].second;
        }
    }
    *len = (int64_t)out.size();
    if (!dst || cap < (int64_t)out.size()) return uvcgpu_fail_(UVCGPU_ENOMEM, "destination too small");
    memcpy(dst, out.data(), out.size());
    return 0;
}
