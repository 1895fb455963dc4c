// uvc_kernels_score.hip -- per-position Bayesian / power-law scoring on gfx950 (fp64 VALU, no MFMA).
//
// Replaces the BcfFormat_symbol* call group of process_batch (main.cpp:608-1000):
//   BcfFormat_symboltype_init  main.hpp:3889   -> k_gather_tot (the symbol-type totals, once per (position, symbol type) group)
//   BcfFormat_symbol_init      main.hpp:4094   -> k_gather_al + the head of k_dpv_pre (fill_symbol_VQ_fmts, main.hpp:3820)
//   BcfFormat_symbol_calc_DPv  main.hpp:4274   -> k_dpv_pre, k_dp4, k_dpv_post
//   BcfFormat_symbol_sum_DPv   main.hpp:4888   -> head of k_qual
//   BcfFormat_symbol_calc_qual main.hpp:4908   -> k_qual
// Tumor-only, and with UvcTumorKey records in the request the normal sample of a T/N pair (SURVEY next-row N2).
#include <algorithm>
#include "uvc_device.h"

#define DBL_EPS 2.220446049250313e-16
#define FLT_EPS 1.1920928955078125e-07

DEV double dmin(double a, double b) { return a < b ? a : b; }
DEV double dmax(double a, double b) { return a > b ? a : b; }
DEV double dbetween(double v, double a, double b) { return dmin(dmax(a, v), b); }
DEV double nnminus_d(double a, double b) { return a > b ? a - b : 0.0; }
DEV double phred2nat(double x) { return (log(10.0) / 10.0) * x; }                       // common.hpp:81
DEV double numstates2phred(double x) { return (10.0 / log(10.0)) * log(x); }            // common.hpp:85
DEV int numstates2deciphred(double x) { return (int)round((100.0 / log(10.0)) * log(x)); }   // common.hpp:87
DEV double prob2odds(double p) { return p / (1.0 - p); }
DEV double logit2(double a, double b) { return log(prob2odds((a + DBL_EPS) / (a + b + 2.0 * DBL_EPS))); }   // main_conversion.hpp:216-219

// calc_binom_10log10_likeratio<false,false>, main_conversion.hpp:222-237
DEV double binom_llr(double prob, double a, double b) {
    prob = (prob + DBL_EPS) / (1.0 + (2.0 * DBL_EPS));
    a += DBL_EPS; b += DBL_EPS;
    const double A = (prob) * (a + b), B = (1.0 - prob) * (a + b);
    if (a > A) return 10.0 / log(10.0) * (a * log(a / A) + b * log(b / B));
    return 0.0;
}

// dp4_to_pcFA<TBidirectional, TIsOverseqFracDisabled>, main_conversion.hpp:798-849
DEV void dp4(double out[2], bool bidir, bool overseq_disabled, double overseq_frac, double aADpass, double aADfail, double aDPpass, double aDPfail,
             double pl_exponent, double n_nats, double aADavgKeyVal = -1, double aDPavgKeyVal = -1, double priorAD = 0.5, double priorDP = 1.0) {
    if (!overseq_disabled) { aDPfail *= overseq_frac; aDPpass *= overseq_frac; aADfail *= overseq_frac; aADpass *= overseq_frac; }
    aDPfail += priorDP; aDPpass += priorDP; aADfail += priorAD; aADpass += priorAD;
    const double nobiasFA = (aADfail + aADpass) / (aDPfail + aDPpass);
    if ((aADpass / aDPpass) >= (aADfail / aDPfail)) {
        if (bidir) { double t = aDPfail; aDPfail = aDPpass; aDPpass = t; t = aADfail; aADfail = aADpass; aADpass = t; }
        else { out[0] = (aADpass / aDPpass); out[1] = nobiasFA; return; }
    }
    const double aBDfail = aDPfail * 2 - aADfail * 1, aBDpass = aDPpass * 2 - aADpass * 1;
    double aADpassfrac = aADpass / (aADpass + aADfail);
    double aBDpassfrac = aBDpass / (aBDpass + aBDfail);
    if ((!bidir) && (aADavgKeyVal >= 0) && (aDPavgKeyVal >= 0)) { aADpassfrac = aADavgKeyVal / (aADavgKeyVal + aDPavgKeyVal * 0.9); aBDpassfrac = 1.0 - aADpassfrac; }
    double infogain = aADfail * log((1.0 - aADpassfrac) / (1.0 - aBDpassfrac));
    if (bidir) infogain += aADpass * log(aADpassfrac / aBDpassfrac);
    if (infogain <= n_nats) { out[0] = aADfail / aDPfail; out[1] = nobiasFA; }
    else { out[0] = dmax(aADpass / aDPpass, (aADfail / aDPfail) * exp((n_nats - infogain) / pl_exponent)); out[1] = nobiasFA; }
}

DEV int indel_len_rusize_phred_s(int indel_len, int repeatunit_size) {   // main.hpp:757-790
    const int t[19] = { 0, 0, 3, 5, 6, 7, 8, 8, 9, 10, 10, 10, 11, 11, 11, 12, 12, 12, 13 };
    if (0 == (indel_len % repeatunit_size)) return t[imin(indel_len / repeatunit_size, 18)];
    return t[imin(indel_len, 18)];
}
DEV int indel_phred_s(double ampfact, int rs, int rn) {   // main.hpp:794-801
    const int region_size = rs * rn;
    const double num_slips = (region_size > 64 ? (double)(region_size - 8) : log1p(exp((double)region_size - (double)8))) * ampfact / ((double)(rs * rs));
    return (int)floor(-10 * log((1.0 - DBL_EPS) / (num_slips + 1.0)) / log(10.0));
}
DEV bool more_STR_s(int rulen1, int rc1, int rulen2, int rc2, int strmax) {   // main.hpp:699-721
    if (rulen2 * rc2 == 0) return true;
    if (rulen1 > strmax || rulen2 > strmax) return (rulen1 < rulen2 || (rulen1 == rulen2 && rc1 > rc2));
    int rank1 = (rc1 <= 1 ? (-rc1 * rulen1) : ((rc1 - 1) * rulen1));
    int rank2 = (rc2 <= 1 ? (-rc2 * rulen1) : ((rc2 - 1) * rulen2));
    if (0 == rc1 || 0 == rulen1) rank1 = -100;
    if (0 == rc2 || 0 == rulen2) rank2 = -100;
    return rank1 > rank2;
}
// indelpos_to_context, main.hpp:733-755 -> (repeat unit length, repeat count)
DEV void indel_context(const RegionDev &R, int refidx, int strmax, int &unit_len, int &repeatnum) {
    const int n = (int)R.npos - 1;
    repeatnum = 0; unit_len = 0;
    if (refidx >= n) return;
    int rs_at_max = 0;
    for (int rs = 1; rs <= strmax; rs++) {
        int q = refidx;
        while ((q + rs < n) && R.refsym[q] == R.refsym[q + rs]) q++;
        const int rn = (q - refidx) / rs + 1;
        if (more_STR_s(rs, rn, rs_at_max, repeatnum, strmax)) { repeatnum = rn; rs_at_max = rs; }
    }
    unit_len = imin(rs_at_max, n - refidx);   // refstring.substr(refpos, n).size()
}

// symbol iteration order of SYMBOL_TYPE_TO_SYMBOLS (main_conversion.hpp:397-400)
DEV int st_symbol(int st, int k) {
    if (st == UVC_BASE_SYMBOL) return k;   // A C G T N NN
    const int link[8] = { UVC_LINK_M, UVC_LINK_I1, UVC_LINK_I2, UVC_LINK_I3P, UVC_LINK_D1, UVC_LINK_D2, UVC_LINK_D3P, UVC_LINK_NN };
    return link[k];
}
DEV int st_count(int st) { return st == UVC_BASE_SYMBOL ? 6 : 8; }
// index of symbol s in the iteration order of its type (inverse of st_symbol)
DEV int st_index(int st, int s) {
    if (st == UVC_BASE_SYMBOL) return s;
    return s == UVC_LINK_M ? 0 : s == UVC_LINK_I1 ? 1 : s == UVC_LINK_I2 ? 2 : s == UVC_LINK_I3P ? 3 : s == UVC_LINK_D1 ? 4 : s == UVC_LINK_D2 ? 5 : s == UVC_LINK_D3P ? 6 : 7;
}
DEV int st_symbol_at(int st, int k) { return st == UVC_BASE_SYMBOL ? k : (k == 0 ? UVC_LINK_M : (k == 7 ? UVC_LINK_NN : UVC_LINK_NN - k)); }   // st_symbol for an index that is not a constant (I1 I2 I3P D1 D2 D3P = 12 .. 7)
DEV double norm_fa(double FA, double refbias) { return (FA + FA * refbias) / (FA + (1.0 - FA) / (1.0 + refbias) + FA * refbias); }           // main.hpp:4253-4256

struct RtrLite { int tracklen, unitlen, anyTR_tracklen; };
DEV RtrLite load_rtr(const RegionDev &R, int idx) { RtrLite r; r.tracklen = RTRP(R, UVC_RTR_tracklen, idx); r.unitlen = RTRP(R, UVC_RTR_unitlen, idx); r.anyTR_tracklen = RTRP(R, UVC_RTR_anyTR_tracklen, idx); return r; }

// (row base = uniform pointer arithmetic, element = 32-bit byte offset: the loads take the scalar-base + vector-offset form instead of a 64-bit address pair per field)
#define ROW_(T, base, row, cap_, idx) (*(T *)((char *)((base) + (size_t)(row) * (size_t)(cap_)) + (unsigned)(idx) * (unsigned)sizeof(T)))
#define OUT(fld, v) ROW_(int32_t, fields, fld, capacity, rec) = (v)

// ------------------------------------------------------------------------------------------------
struct ScoreCtx {
    int pos_beg, pos_end, all_out, is_amplicon, base_at_beg;
    const UvcIndelAllele *alleles; long long n_alleles;   // sorted by (refpos, symbol): the region's own InDel alleles, or the caller's where it listed any
    const UvcGapRow *gap_rows; const uint8_t *gap_seq;   // the allele table rows (text order of InDel strings in k_call)
    const int32_t *allele_rows;                           // parallel: row of uvcgpu_region_indel_alleles that carries the allele's string, or -1
    const UvcTumorKey *tkeys; long long n_tkeys;          // sorted by (refpos, symbol); only read when tumor_vcf_is_provided
    int32_t *fields; long long capacity;
    long long *offsets;   // exclusive prefix of per-group packed (flag << 32 | allele count), [2 * (pos_end - pos_beg) + 1]
    int *active;          // groups with at least one allele, ascending
};
#define PK_COUNT(v) ((long long)((v) & 0xFFFFFFFFLL))
#define PK_FLAGS(v) ((long long)((v) >> 32))

DEV long long allele_lower_bound(const ScoreCtx &C, int refpos, int symbol) {
    long long lo = 0, hi = C.n_alleles;
    while (lo < hi) { long long mid = (lo + hi) >> 1; const UvcIndelAllele &a = C.alleles[mid]; if (a.refpos < refpos || (a.refpos == refpos && a.symbol < symbol)) lo = mid + 1; else hi = mid; }
    return lo;
}
DEV int allele_multiplicity(const ScoreCtx &C, int refpos, int symbol, long long &first) {
    first = -1;
    if (!(is_ins(symbol) || is_del(symbol)) || C.n_alleles == 0) return 1;
    const long long lo = allele_lower_bound(C, refpos, symbol);
    long long hi = lo;
    while (hi < C.n_alleles && C.alleles[hi].refpos == refpos && C.alleles[hi].symbol == symbol) hi++;
    if (hi == lo) return 1;
    first = lo;
    return (int)(hi - lo);
}

DEV long long tkey_lower_bound(const ScoreCtx &C, int refpos, int symbol) {
    long long lo = 0, hi = C.n_tkeys;
    while (lo < hi) { long long mid = (lo + hi) >> 1; const UvcTumorKey &a = C.tkeys[mid]; if (a.refpos < refpos || (a.refpos == refpos && a.symbol < symbol)) lo = mid + 1; else hi = mid; }
    return lo;
}
// alleles of one (position, symbol): the tumor records when there are any (src = 2, is_var_rescued, main.cpp:806, 864-900), else the
// host-supplied InDel alleles (src = 1), else the single default allele (src = 0)
DEV int allele_source(const ScoreCtx &C, bool tprov, int refpos, int symbol, long long &first, int &src) {
    if (tprov && C.n_tkeys) {
        const long long lo = tkey_lower_bound(C, refpos, symbol);
        long long hi = lo;
        while (hi < C.n_tkeys && C.tkeys[hi].refpos == refpos && C.tkeys[hi].symbol == symbol) hi++;
        if (hi > lo) { first = lo; src = 2; return (int)(hi - lo); }
    }
    const int m = allele_multiplicity(C, refpos, symbol, first);
    src = (first >= 0 ? 1 : 0);
    return m;
}
DEV int group_refsymbol(const RegionDev &R, int zpos, int st) {   // symboltype_to_refsymbol, main.cpp:616-620
    if (st == UVC_LINK_SYMBOL) return UVC_LINK_M;
    const int refidx = zpos - R.beg, refsize = (int)R.npos - 1;
    return ((refsize == (refidx - 1) || (-1 == (refidx - 1))) ? UVC_BASE_NN : (int)R.refsym[refidx - 1]);
}
// ================================================================================================
// The staged form of the scoring path (round 4).  Scoring reads ~6 KB of plane cells per record, scattered over ~460 cache lines; the old
// k_score did those loads from inside the fp64 arithmetic, one lane per record at one wave per SIMD.  Now:
//   k_gate_scan   dense pass over the fragment depths (28 cells per position): candidate gate, record slots by a chained scan, active-group list
//   k_enum        per active group: which symbols have anything at this position (mask), the group's scalars, one header per record
//   k_gather      one thread per (group, plane): every plane cell the arithmetic needs is fetched ONCE into compact rows
//                 [field][group] / [field][record] -- memory-parallel, tens of thousands of independent waves
//   k_dpv_pre     per record: fill_symbol_VQ_fmts + calc_DPv up to the contingency tests
//   k_dp4         per (record, test): the 14 dp4_to_pcFA evaluations of a record are independent of each other
//   k_dpv_post    per record: the minima, FTS, cDP1v .. cDP2x
//   k_qual        per record: sum_DPv over the group's records + calc_qual
// The arithmetic kernels read coalesced rows, hold no plane pointers and fit four waves per SIMD.
// ================================================================================================
enum { SG_PREP32 = 0, SG_PREP64, SG_SEG32, SG_SEG64, SG_VQ, SG_FRAG, SG_FAM, SG_FI32, SG_FI64, SG_DUP };
struct StageDesc { unsigned char grp, trunc; unsigned short plane; };

// symbol-type totals (fill_symboltype_fmt, main.hpp:3745-3793): X(name, plane group, plane, the int32 FORMAT field truncates the sum)
#define TOT_LIST(X) \
    X(APDP0, SG_PREP32, UVC_P_a_dp, 0) X(APDP1, SG_PREP32, UVC_P_a_near_ins_dp, 0) X(APDP2, SG_PREP32, UVC_P_a_near_del_dp, 0) X(APDP3, SG_PREP32, UVC_P_a_near_RTR_ins_dp, 0) \
    X(APDP4, SG_PREP32, UVC_P_a_near_RTR_del_dp, 0) X(APDP5, SG_PREP32, UVC_P_a_pcr_dp, 0) X(APDP6, SG_PREP32, UVC_P_a_snv_dp, 0) X(APDP7, SG_PREP32, UVC_P_a_dnv_dp, 0) \
    X(APDP8, SG_PREP32, UVC_P_a_highBQ_dp, 0) X(APDP9, SG_PREP32, UVC_P_a_near_pcr_clip_dp, 0) X(APDP10, SG_PREP32, UVC_P_a_near_long_clip_dp, 0) X(APDP11, SG_PREP32, UVC_P_a_umi_dp, 0) \
    X(APXM0, SG_PREP32, UVC_P_a_XM1500, 0) X(APXM1, SG_PREP32, UVC_P_a_GO1500, 0) X(APXM2, SG_PREP32, UVC_P_a_qlen, 0) X(APXM3, SG_PREP32, UVC_P_a_GAPLEN, 0) \
    X(APXM4, SG_PREP64, UVC_P_a_near_ins_pow2len, 0) X(APXM5, SG_PREP64, UVC_P_a_near_del_pow2len, 0) X(APXM6, SG_PREP32, UVC_P_a_near_ins_inv100len, 0) X(APXM7, SG_PREP32, UVC_P_a_near_del_inv100len, 0) \
    X(APLRI0, SG_PREP64, UVC_P_a_LI, 0) X(APLRI1, SG_PREP32, UVC_P_a_LIDP, 0) X(APLRI2, SG_PREP64, UVC_P_a_RI, 0) X(APLRI3, SG_PREP32, UVC_P_a_RIDP, 0) \
    X(A1BQf0, SG_VQ, UVC_VQ_a1BQf, 1) X(A1BQr0, SG_VQ, UVC_VQ_a1BQr, 1) X(AMQs0, SG_SEG32, UVC_S_aMQs, 1) X(AP10, SG_SEG32, UVC_S_aP1, 1) X(AP20, SG_SEG32, UVC_S_aP2, 1) \
    X(ADPff0, SG_SEG32, UVC_S_aDPff, 1) X(ADPfr0, SG_SEG32, UVC_S_aDPfr, 1) X(ADPrf0, SG_SEG32, UVC_S_aDPrf, 1) X(ADPrr0, SG_SEG32, UVC_S_aDPrr, 1) \
    X(ALP10, SG_SEG32, UVC_S_aLP1, 1) X(ALP20, SG_SEG32, UVC_S_aLP2, 1) X(ALPL0, SG_SEG32, UVC_S_aLPL, 0) X(ARP20, SG_SEG32, UVC_S_aRP2, 1) X(ARPL0, SG_SEG32, UVC_S_aRPL, 0) \
    X(ALB20, SG_SEG32, UVC_S_aLB2, 1) X(ALBL0, SG_SEG64, UVC_S64_aLBL, 0) X(ARB20, SG_SEG32, UVC_S_aRB2, 1) X(ARBL0, SG_SEG64, UVC_S64_aRBL, 0) \
    X(ABQ20, SG_SEG32, UVC_S_aBQ2, 1) X(APF20, SG_SEG32, UVC_S_aPF2, 1) X(ALI20, SG_SEG32, UVC_S_aLI2, 1) X(ARIf0, SG_SEG32, UVC_S_aRIf, 1) X(ARI20, SG_SEG32, UVC_S_aRI2, 1) X(ALIr0, SG_SEG32, UVC_S_aLIr, 1) \
    X(BDPb0, SG_FRAG, 0 * UVC_NFRAG + UVC_FRAG_bDP, 1) X(BDPb1, SG_FRAG, 1 * UVC_NFRAG + UVC_FRAG_bDP, 1) X(BTAb0, SG_FRAG, 0 * UVC_NFRAG + UVC_FRAG_bTA, 1) X(BTAb1, SG_FRAG, 1 * UVC_NFRAG + UVC_FRAG_bTA, 1) \
    X(BTBb0, SG_FRAG, 0 * UVC_NFRAG + UVC_FRAG_bTB, 1) X(BTBb1, SG_FRAG, 1 * UVC_NFRAG + UVC_FRAG_bTB, 1) \
    X(CDP1b0, SG_FAM, 0 * UVC_NFAM + UVC_FAM_cDP1, 1) X(CDP1b1, SG_FAM, 1 * UVC_NFAM + UVC_FAM_cDP1, 1) X(CDP12b0, SG_FAM, 0 * UVC_NFAM + UVC_FAM_cDP12, 1) X(CDP12b1, SG_FAM, 1 * UVC_NFAM + UVC_FAM_cDP12, 1) \
    X(CDP2b0, SG_FAM, 0 * UVC_NFAM + UVC_FAM_cDP2, 1) X(CDP2b1, SG_FAM, 1 * UVC_NFAM + UVC_FAM_cDP2, 1) X(CDP3b0, SG_FAM, 0 * UVC_NFAM + UVC_FAM_cDP3, 1) X(CDP3b1, SG_FAM, 1 * UVC_NFAM + UVC_FAM_cDP3, 1) \
    X(C2LP20, SG_FI32, UVC_FI_c2LP2, 1) X(C2LPL0, SG_FI32, UVC_FI_c2LPL, 0) X(C2RP20, SG_FI32, UVC_FI_c2RP2, 1) X(C2RPL0, SG_FI32, UVC_FI_c2RPL, 0) \
    X(C2LB20, SG_FI32, UVC_FI_c2LB2, 1) X(C2LBL0, SG_FI64, UVC_FI64_c2LBL, 0) X(C2RB20, SG_FI32, UVC_FI_c2RB2, 1) X(C2RBL0, SG_FI64, UVC_FI64_c2RBL, 0) \
    X(C2BQ20, SG_FI32, UVC_FI_c2BQ2, 1) X(C2LP00, SG_FI32, UVC_FI_c2LP0, 1) X(C2RP00, SG_FI32, UVC_FI_c2RP0, 1) X(DDP10, SG_DUP, UVC_DUPLEX_dDP1, 1)
// one allele's own cells (BcfFormat_symbol_init, main.hpp:4094-4251): X(name, plane group, plane); the 64-bit planes last
#define AL_LIST(X) \
    X(a1BQf, SG_VQ, UVC_VQ_a1BQf) X(a1BQr, SG_VQ, UVC_VQ_a1BQr) X(v2BQf, SG_VQ, UVC_VQ_a2BQf) X(v2BQr, SG_VQ, UVC_VQ_a2BQr) X(vbMQ, SG_VQ, UVC_VQ_bMQ) \
    X(bIAQb, SG_VQ, UVC_VQ_bIAQb) X(bIADb, SG_VQ, UVC_VQ_bIADb) X(cIAQf, SG_VQ, UVC_VQ_cIAQf) X(cIADf, SG_VQ, UVC_VQ_cIADf) X(cIDQf, SG_VQ, UVC_VQ_cIDQf) \
    X(cIAQr, SG_VQ, UVC_VQ_cIAQr) X(cIADr, SG_VQ, UVC_VQ_cIADr) X(cIDQr, SG_VQ, UVC_VQ_cIDQr) \
    X(aMQs, SG_SEG32, UVC_S_aMQs) X(aP1, SG_SEG32, UVC_S_aP1) X(aP2, SG_SEG32, UVC_S_aP2) X(aDPff, SG_SEG32, UVC_S_aDPff) X(aDPfr, SG_SEG32, UVC_S_aDPfr) X(aDPrf, SG_SEG32, UVC_S_aDPrf) X(aDPrr, SG_SEG32, UVC_S_aDPrr) \
    X(aLP1, SG_SEG32, UVC_S_aLP1) X(aLP2, SG_SEG32, UVC_S_aLP2) X(aLPL, SG_SEG32, UVC_S_aLPL) X(aRP1, SG_SEG32, UVC_S_aRP1) X(aRP2, SG_SEG32, UVC_S_aRP2) X(aRPL, SG_SEG32, UVC_S_aRPL) \
    X(aLB1, SG_SEG32, UVC_S_aLB1) X(aLB2, SG_SEG32, UVC_S_aLB2) X(aRB1, SG_SEG32, UVC_S_aRB1) X(aRB2, SG_SEG32, UVC_S_aRB2) X(a2XM2, SG_SEG32, UVC_S_a2XM2) X(a2BM2, SG_SEG32, UVC_S_a2BM2) X(aBQ2, SG_SEG32, UVC_S_aBQ2) \
    X(aPF1, SG_SEG32, UVC_S_aPF1) X(aPF2, SG_SEG32, UVC_S_aPF2) X(aLI1, SG_SEG32, UVC_S_aLI1) X(aLI2, SG_SEG32, UVC_S_aLI2) X(aLIr, SG_SEG32, UVC_S_aLIr) X(aRI1, SG_SEG32, UVC_S_aRI1) X(aRI2, SG_SEG32, UVC_S_aRI2) X(aRIf, SG_SEG32, UVC_S_aRIf) \
    X(aP3, SG_SEG32, UVC_S_aP3) X(aNC, SG_SEG32, UVC_S_aNC) \
    X(bDPf, SG_FRAG, 0 * UVC_NFRAG + UVC_FRAG_bDP) X(bTAf, SG_FRAG, 0 * UVC_NFRAG + UVC_FRAG_bTA) X(bTBf, SG_FRAG, 0 * UVC_NFRAG + UVC_FRAG_bTB) \
    X(bDPr, SG_FRAG, 1 * UVC_NFRAG + UVC_FRAG_bDP) X(bTAr, SG_FRAG, 1 * UVC_NFRAG + UVC_FRAG_bTA) X(bTBr, SG_FRAG, 1 * UVC_NFRAG + UVC_FRAG_bTB) \
    X(cDP1f, SG_FAM, 0 * UVC_NFAM + UVC_FAM_cDP1) X(cDP12f, SG_FAM, 0 * UVC_NFAM + UVC_FAM_cDP12) X(cDP2f, SG_FAM, 0 * UVC_NFAM + UVC_FAM_cDP2) X(cDP3f, SG_FAM, 0 * UVC_NFAM + UVC_FAM_cDP3) \
    X(cDPMf, SG_FAM, 0 * UVC_NFAM + UVC_FAM_cDPM) X(cDPmf, SG_FAM, 0 * UVC_NFAM + UVC_FAM_cDPm) \
    X(cDP1r, SG_FAM, 1 * UVC_NFAM + UVC_FAM_cDP1) X(cDP12r, SG_FAM, 1 * UVC_NFAM + UVC_FAM_cDP12) X(cDP2r, SG_FAM, 1 * UVC_NFAM + UVC_FAM_cDP2) X(cDP3r, SG_FAM, 1 * UVC_NFAM + UVC_FAM_cDP3) \
    X(cDPMr, SG_FAM, 1 * UVC_NFAM + UVC_FAM_cDPM) X(cDPmr, SG_FAM, 1 * UVC_NFAM + UVC_FAM_cDPm) \
    X(c2LP1, SG_FI32, UVC_FI_c2LP1) X(c2LP2, SG_FI32, UVC_FI_c2LP2) X(c2LPL, SG_FI32, UVC_FI_c2LPL) X(c2RP1, SG_FI32, UVC_FI_c2RP1) X(c2RP2, SG_FI32, UVC_FI_c2RP2) X(c2RPL, SG_FI32, UVC_FI_c2RPL) \
    X(c2LB1, SG_FI32, UVC_FI_c2LB1) X(c2LB2, SG_FI32, UVC_FI_c2LB2) X(c2RB1, SG_FI32, UVC_FI_c2RB1) X(c2RB2, SG_FI32, UVC_FI_c2RB2) X(c2BQ2, SG_FI32, UVC_FI_c2BQ2) X(c2LP0, SG_FI32, UVC_FI_c2LP0) X(c2RP0, SG_FI32, UVC_FI_c2RP0) \
    X(dDP1, SG_DUP, UVC_DUPLEX_dDP1) X(dDP2, SG_DUP, UVC_DUPLEX_dDP2) \
    X(aLBL, SG_SEG64, UVC_S64_aLBL) X(aRBL, SG_SEG64, UVC_S64_aRBL) X(aLIT, SG_SEG64, UVC_S64_aLIT) X(aRIT, SG_SEG64, UVC_S64_aRIT) X(c2LBL, SG_FI64, UVC_FI64_c2LBL) X(c2RBL, SG_FI64, UVC_FI64_c2RBL)
#define NAL64 6
#define X(n, g, p, t) TOT_##n,
enum { TOT_LIST(X) NTOT };
#undef X
#define X(n, g, p) AL_##n,
enum { AL_LIST(X) NAL };
#undef X
#define NAL32 (NAL - NAL64)
// One table of the distinct planes the two lists read (every symbol plane of TOT_LIST is also an allele cell): the gather fetches a plane's
// cells of a group once and serves the group's total and its records' own cells from them.
struct GatherDesc { unsigned char grp, trunc; unsigned short plane; short tot, al; };
// bits of a group's mask word above the eight symbol bits: plane groups that can hold something at this position (k_enum)
enum { GM_FI = 1u << 8, GM_DUP = 1u << 9, GM_P5F = 1u << 10, GM_P5R = 1u << 11 };
// the bit a plane waits for (0: always fetched)
constexpr unsigned gather_need(int grp, int plane) {
    return (grp == SG_FI32 || grp == SG_FI64) ? (unsigned)GM_FI : grp == SG_DUP ? (unsigned)GM_DUP
         : (grp == SG_VQ && (plane == UVC_VQ_cIAQf || plane == UVC_VQ_cIADf || plane == UVC_VQ_cIDQf)) ? (unsigned)GM_P5F
         : (grp == SG_VQ && (plane == UVC_VQ_cIAQr || plane == UVC_VQ_cIADr || plane == UVC_VQ_cIDQr)) ? (unsigned)GM_P5R : 0u;
}
#define NGATHER (24 + NAL)   // the 24 per-position planes of TOT_LIST (APDP, APXM, APLRI) + the allele cells
struct GatherTab { GatherDesc d[NGATHER]; };
constexpr GatherTab make_gather_tab() {
    GatherTab g{};
#define X(n, gr, p, t) { (unsigned char)(gr), (unsigned char)(t), (unsigned short)(p) },
    const StageDesc tot[NTOT] = { TOT_LIST(X) };
#undef X
#define X(n, gr, p) { (unsigned char)(gr), 0, (unsigned short)(p) },
    const StageDesc al[NAL] = { AL_LIST(X) };
#undef X
    int n = 0;
    for (int i = 0; i < NTOT; i++) if (tot[i].grp <= SG_PREP64) { g.d[n].grp = tot[i].grp; g.d[n].trunc = 0; g.d[n].plane = tot[i].plane; g.d[n].tot = (short)i; g.d[n].al = -1; n++; }
    for (int a = 0; a < NAL; a++) {
        g.d[n].grp = al[a].grp; g.d[n].plane = al[a].plane; g.d[n].al = (short)a; g.d[n].tot = -1; g.d[n].trunc = 0;
        for (int i = 0; i < NTOT; i++) if (tot[i].grp == al[a].grp && tot[i].plane == al[a].plane) { g.d[n].tot = (short)i; g.d[n].trunc = tot[i].trunc; }
        n++;
    }
    return g;
}
__constant__ GatherTab c_gather = make_gather_tab();
// per active group
enum { GR_x = 0, GR_zpos, GR_st, GR_refsym, GR_mask, GR_hp, GR_r1t, GR_r1u, GR_r1a, GR_r2t, GR_r2u, GR_r2a, GR_insc, GR_delc, GR_ins1c, GR_del1c, GR_rusize, GR_repnum, GR_rec0, GR_nrec, GR_refbdp, GR_vAC, GR_gemit, GR_kept, NGR };
// per record
enum { RH_gi = 0, RH_symbol, RH_src, RH_idx, RH_bdepth, RH_cdepth, NRH };
// what k_dpv_pre hands to k_dp4 / k_dpv_post
enum { MID_cbP = 0, MID_cbBQ, MID_dirdiv, MID_aDPFA, MID_cFA2L, MID_cFA2R, MID_dff, MID_pcread, MID_aPprior, MID_aBprior, MID_aIprior, MID_aSBprior, MID_oriall, NMID };
#define NDP4 14
enum { CNT_nvalid = 0, CNT_ticket1, CNT_ticket2, CNT_kept, NCNT = 8 };

struct Stage {
    int32_t *grp;        // [NGR][cap]
    long long *tot;      // [NTOT][cap]
    int32_t *rh;         // [NRH][cap]
    int32_t *al;         // [NAL32][cap]
    long long *al64;     // [NAL64][cap]
    double *mid;         // [NMID][cap]
    double *d4;          // [NDP4][2][cap]
    long long cap;       // records (and therefore active groups) the rows can hold
    unsigned int *cnt;   // [NCNT] counters, zeroed in front of every call: records of the groups that fit, tickets of the two chained scans
    unsigned long long *status1, *status2;   // tile states of the two chained scans (zeroed with the counters)
    int32_t *keptoff;    // [cap] kept_only: first record of an active group in the kept array, or -1
};
#define GR_(f, gi_) ROW_(int32_t, S.grp, GR_##f, S.cap, gi_)
#define RH_(f, r_) ROW_(int32_t, S.rh, RH_##f, S.cap, r_)
#define MID_(f, r_) ROW_(double, S.mid, MID_##f, S.cap, r_)
#define D4_(t, k, r_) ROW_(double, S.d4, (t) * 2 + (k), S.cap, r_)

DEV long long stage_cell(const RegionDev &R, int grp, int plane, int s, int64_t x) {
    const size_t np = (size_t)R.npos;
    switch (grp) {
        case SG_PREP32: return R.prep32[(size_t)plane * np + x];
        case SG_PREP64: return R.prep64[(size_t)plane * np + x];
        case SG_SEG32: return R.seg32[((size_t)plane * NSYM + s) * np + x];
        case SG_SEG64: return R.seg64[((size_t)plane * NSYM + s) * np + x];
        case SG_VQ: return R.vq[((size_t)plane * NSYM + s) * np + x];
        case SG_FRAG: return R.frag[((size_t)plane * NSYM + s) * np + x];
        case SG_FAM: return R.fam[((size_t)plane * NSYM + s) * np + x];
        case SG_FI32: return R.faminfo32[((size_t)plane * NSYM + s) * np + x];
        case SG_FI64: return R.faminfo64[((size_t)plane * NSYM + s) * np + x];
        default: return R.duplex[((size_t)plane * NSYM + s) * np + x];
    }
}
// staged-row accessors: the names of the old per-thread structs (Tot T, Al f) as loads at the point of use; `gi` / `rec` are the thread's group / record
#define TT_(n) ROW_(long long, S.tot, TOT_##n, S.cap, gi)
#define AL_(n) ROW_(int32_t, S.al, AL_##n, S.cap, rec)
#define AL64_(n) ROW_(long long, S.al64, AL_##n - NAL32, S.cap, rec)
#define T_APDP0 TT_(APDP0)
#define T_APDP1 TT_(APDP1)
#define T_APDP2 TT_(APDP2)
#define T_APDP3 TT_(APDP3)
#define T_APDP4 TT_(APDP4)
#define T_APDP5 TT_(APDP5)
#define T_APDP6 TT_(APDP6)
#define T_APDP7 TT_(APDP7)
#define T_APDP8 TT_(APDP8)
#define T_APDP9 TT_(APDP9)
#define T_APDP10 TT_(APDP10)
#define T_APDP11 TT_(APDP11)
#define T_APXM0 TT_(APXM0)
#define T_APXM1 TT_(APXM1)
#define T_APXM2 TT_(APXM2)
#define T_APXM3 TT_(APXM3)
#define T_APXM4 TT_(APXM4)
#define T_APXM5 TT_(APXM5)
#define T_APXM6 TT_(APXM6)
#define T_APXM7 TT_(APXM7)
#define T_APLRI0 TT_(APLRI0)
#define T_APLRI1 TT_(APLRI1)
#define T_APLRI2 TT_(APLRI2)
#define T_APLRI3 TT_(APLRI3)
#define T_A1BQf0 TT_(A1BQf0)
#define T_A1BQr0 TT_(A1BQr0)
#define T_AMQs0 TT_(AMQs0)
#define T_AP10 TT_(AP10)
#define T_AP20 TT_(AP20)
#define T_ADPff0 TT_(ADPff0)
#define T_ADPfr0 TT_(ADPfr0)
#define T_ADPrf0 TT_(ADPrf0)
#define T_ADPrr0 TT_(ADPrr0)
#define T_ALP10 TT_(ALP10)
#define T_ALP20 TT_(ALP20)
#define T_ALPL0 TT_(ALPL0)
#define T_ARP20 TT_(ARP20)
#define T_ARPL0 TT_(ARPL0)
#define T_ALB20 TT_(ALB20)
#define T_ALBL0 TT_(ALBL0)
#define T_ARB20 TT_(ARB20)
#define T_ARBL0 TT_(ARBL0)
#define T_ABQ20 TT_(ABQ20)
#define T_APF20 TT_(APF20)
#define T_ALI20 TT_(ALI20)
#define T_ARIf0 TT_(ARIf0)
#define T_ARI20 TT_(ARI20)
#define T_ALIr0 TT_(ALIr0)
#define T_C2LP20 TT_(C2LP20)
#define T_C2LPL0 TT_(C2LPL0)
#define T_C2RP20 TT_(C2RP20)
#define T_C2RPL0 TT_(C2RPL0)
#define T_C2LB20 TT_(C2LB20)
#define T_C2LBL0 TT_(C2LBL0)
#define T_C2RB20 TT_(C2RB20)
#define T_C2RBL0 TT_(C2RBL0)
#define T_C2BQ20 TT_(C2BQ20)
#define T_C2LP00 TT_(C2LP00)
#define T_C2RP00 TT_(C2RP00)
#define T_BDPb0 ((int)TT_(BDPb0))
#define T_BDPb1 ((int)TT_(BDPb1))
#define T_BTAb0 ((int)TT_(BTAb0))
#define T_BTAb1 ((int)TT_(BTAb1))
#define T_BTBb0 ((int)TT_(BTBb0))
#define T_BTBb1 ((int)TT_(BTBb1))
#define T_CDP1b0 ((int)TT_(CDP1b0))
#define T_CDP1b1 ((int)TT_(CDP1b1))
#define T_CDP12b0 ((int)TT_(CDP12b0))
#define T_CDP12b1 ((int)TT_(CDP12b1))
#define T_CDP2b0 ((int)TT_(CDP2b0))
#define T_CDP2b1 ((int)TT_(CDP2b1))
#define T_CDP3b0 ((int)TT_(CDP3b0))
#define T_CDP3b1 ((int)TT_(CDP3b1))
#define T_DDP10 ((int)TT_(DDP10))
#define f_a1BQf AL_(a1BQf)
#define f_a1BQr AL_(a1BQr)
#define f_bIAQb AL_(bIAQb)
#define f_bIADb AL_(bIADb)
#define f_cIAQf AL_(cIAQf)
#define f_cIADf AL_(cIADf)
#define f_cIDQf AL_(cIDQf)
#define f_cIAQr AL_(cIAQr)
#define f_cIADr AL_(cIADr)
#define f_cIDQr AL_(cIDQr)
#define f_aMQs AL_(aMQs)
#define f_aP1 AL_(aP1)
#define f_aP2 AL_(aP2)
#define f_aDPff AL_(aDPff)
#define f_aDPfr AL_(aDPfr)
#define f_aDPrf AL_(aDPrf)
#define f_aDPrr AL_(aDPrr)
#define f_aLP1 AL_(aLP1)
#define f_aLP2 AL_(aLP2)
#define f_aRP1 AL_(aRP1)
#define f_aRP2 AL_(aRP2)
#define f_aLB1 AL_(aLB1)
#define f_aLB2 AL_(aLB2)
#define f_aRB1 AL_(aRB1)
#define f_aRB2 AL_(aRB2)
#define f_a2XM2 AL_(a2XM2)
#define f_a2BM2 AL_(a2BM2)
#define f_aBQ2 AL_(aBQ2)
#define f_aPF1 AL_(aPF1)
#define f_aPF2 AL_(aPF2)
#define f_aLI1 AL_(aLI1)
#define f_aLI2 AL_(aLI2)
#define f_aLIr AL_(aLIr)
#define f_aRI1 AL_(aRI1)
#define f_aRI2 AL_(aRI2)
#define f_aRIf AL_(aRIf)
#define f_aP3 AL_(aP3)
#define f_aNC AL_(aNC)
#define f_bDPf AL_(bDPf)
#define f_bTAf AL_(bTAf)
#define f_bTBf AL_(bTBf)
#define f_bDPr AL_(bDPr)
#define f_bTAr AL_(bTAr)
#define f_bTBr AL_(bTBr)
#define f_cDP1f AL_(cDP1f)
#define f_cDP12f AL_(cDP12f)
#define f_cDP2f AL_(cDP2f)
#define f_cDP3f AL_(cDP3f)
#define f_cDPMf AL_(cDPMf)
#define f_cDPmf AL_(cDPmf)
#define f_cDP1r AL_(cDP1r)
#define f_cDP12r AL_(cDP12r)
#define f_cDP2r AL_(cDP2r)
#define f_cDP3r AL_(cDP3r)
#define f_cDPMr AL_(cDPMr)
#define f_cDPmr AL_(cDPmr)
#define f_c2LP1 AL_(c2LP1)
#define f_c2LP2 AL_(c2LP2)
#define f_c2RP1 AL_(c2RP1)
#define f_c2RP2 AL_(c2RP2)
#define f_c2LB1 AL_(c2LB1)
#define f_c2LB2 AL_(c2LB2)
#define f_c2RB1 AL_(c2RB1)
#define f_c2RB2 AL_(c2RB2)
#define f_c2BQ2 AL_(c2BQ2)
#define f_c2LP0 AL_(c2LP0)
#define f_c2RP0 AL_(c2RP0)
#define f_dDP1 AL_(dDP1)
#define f_dDP2 AL_(dDP2)
#define f_aLPL ((long long)AL_(aLPL))
#define f_aRPL ((long long)AL_(aRPL))
#define f_c2LPL ((long long)AL_(c2LPL))
#define f_c2RPL ((long long)AL_(c2RPL))
#define f_aLBL AL64_(aLBL)
#define f_aRBL AL64_(aRBL)
#define f_aLIT AL64_(aLIT)
#define f_aRIT AL64_(aRIT)
#define f_c2LBL AL64_(c2LBL)
#define f_c2RBL AL64_(c2RBL)
// ---- chained scan (one launch): every block takes the next tile by ticket, publishes the tile's sum, and adds up its predecessors'.
// A tile's state is ONE naturally aligned 8-byte word written by one agent-scope store and polled by agent-scope loads (nothing else is
// handed between blocks), bits 63..62: 1 = sum of this tile, 2 = sum of all tiles up to and including this one; bits 61..0 the packed sums.
// Tickets are dealt in order, so the block of tile t - 1 is resident or done whenever tile t looks back: the spin is bounded all the same.
#define CS_VALUE(w) ((w) & 0x3FFFFFFFFFFFFFFFull)
DEV unsigned long long cs_load(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void cs_store(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// wave 0 of the block calls this (all 64 lanes); returns the exclusive prefix of `tile` in every lane
DEV unsigned long long cs_lookback(unsigned long long *status, int tile, unsigned long long aggregate, int lane, int *err) {
    if (tile == 0) { if (lane == 0) cs_store(&status[0], (2ull << 62) | aggregate); return 0; }
    if (lane == 0) cs_store(&status[tile], (1ull << 62) | aggregate);
    unsigned long long run = 0;
    int j = tile - 1;   // lane l looks at tile j - l
    for (unsigned spins = 0; ; ) {
        const int t = j - lane;
        const unsigned long long w = (t >= 0 ? cs_load(&status[t]) : (2ull << 62));   // in front of tile 0: a prefix of zero
        const unsigned long long incl = __ballot((w >> 62) == 2), ready = __ballot((w >> 62) != 0);
        const int k = (incl ? __builtin_ctzll(incl) : 64);                              // the nearest tile that already knows its prefix
        const unsigned long long need = (k >= 63 ? ~0ull : ((1ull << (k + 1)) - 1));
        if ((ready & need) == need) {
            unsigned long long v = (lane <= k ? CS_VALUE(w) : 0ull);
#pragma unroll
            for (int d = 32; d; d >>= 1) v += __shfl_xor(v, d);
            run += v;
            if (k < 64) break;
            j -= 64;
            continue;
        }
        if (++spins > (1u << 24)) { if (lane == 0) atomicExch(err, UVCGPU_EDEVICE); break; }   // cannot happen (see above); never hang the device
        __builtin_amdgcn_s_sleep(2);
    }
    if (lane == 0) cs_store(&status[tile], (2ull << 62) | CS_VALUE(run + aggregate));
    return run;
}
// exclusive scan of one value per thread over the block (256 threads), the block's total in `total`
DEV unsigned long long block_excl_scan256(unsigned long long v, unsigned long long *sh_wave /* [4] */, unsigned long long &total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(inc, d); if (lane >= d) inc += o; }
    if (lane == 63) sh_wave[wv] = inc;
    __syncthreads();
    unsigned long long base = 0;
    for (int q = 0; q < 4; q++) { if (q < wv) base += sh_wave[q]; }
    total = sh_wave[0] + sh_wave[1] + sh_wave[2] + sh_wave[3];
    return base + inc - v;
}

// candidate gate, main.cpp:801-840, on the fragment depth alone (cdepth is an output of the gate, not part of the decision)
DEV bool gate_b(const UvcParams &P, int symbol, int refsymbol, int bdepth, int totBDP, bool all_out, bool pos_rescued) {
    if (P.tumor_vcf_is_provided) return pos_rescued;   // normal sample: every symbol of a position the tumor has a record at, nothing else (main.cpp:832-840)
    if (all_out) return true;
    if (refsymbol != symbol) return !(bdepth < P.min_altdp_thres);
    return !(totBDP - bdepth < P.min_altdp_thres);
}
// The symbols of a type that can have a non-zero cell at position x, as bits in the type's iteration order (st_symbol): the position's dense
// symbol of the type (reference base / LINK_M: P2 and the fragment kernel store them without saying so) and what RegionDev::occ names --
// every other writer of a cell marks its symbol there (occ_mark).  uvcgpu_region_check_presence holds all planes against this statement.
DEV unsigned type_mask(const RegionDev &R, int st, int64_t x, int refsymbol) {
    const unsigned occ = R.occ[x];
    unsigned m;
    if (st == UVC_BASE_SYMBOL) m = (occ & 0x3Fu) | (1u << (int)R.refsym[x < R.npos - 1 ? x : R.npos - 1]);   // what the dense kernels call the reference base of x
    else { const unsigned o = (occ >> UVC_LINK_M) & 0xFFu; m = (o & 0x81u) | (__brev(o & 0x7Eu) >> 24) | 1u; }   // M D3P D2 D1 I3P I2 I1 NN -> M I1 I2 I3P D1 D2 D3P NN
    return m | (1u << st_index(st, refsymbol));
}
// fragment depth of the marked symbols (the others have none), by iteration index
DEV int masked_bdepths(const RegionDev &R, int st, int64_t x, unsigned mask, int bd[8]) {
    int tot = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) bd[k] = 0;
    for (unsigned m = mask & 0xFFu; m; m &= m - 1u) {
        const int k = __builtin_ctz(m), s = st_symbol_at(st, k);
        const int v = FRP(R, 0, UVC_FRAG_bDP, s, x) + FRP(R, 1, UVC_FRAG_bDP, s, x);
        tot += v;
#pragma unroll
        for (int kk = 0; kk < 8; kk++) bd[kk] = (kk == k ? v : bd[kk]);
    }
    return tot;
}
DEV int gate_count(const RegionDev &R, const UvcParams &P, const ScoreCtx &C, long long g) {
    const int zpos = C.pos_beg + (int)(g >> 1), st = (int)(g & 1);
    if (zpos == C.pos_beg && st == UVC_BASE_SYMBOL && !C.base_at_beg) return 0;   // main.cpp:643
    const int refpos = (st == UVC_BASE_SYMBOL ? zpos - 1 : zpos);
    const int64_t x = refpos - R.beg;
    const int refsymbol = group_refsymbol(R, zpos, st), nsym = st_count(st);
    int bd[8], totBDP = 0;
    if (C.all_out || P.tumor_vcf_is_provided) {   // the gate does not look at the depths
#pragma unroll
        for (int k = 0; k < 8; k++) bd[k] = 0;
    } else {
        const unsigned mask = type_mask(R, st, x, refsymbol);
        // nothing but the reference symbol at this position: an ALT needs min_altdp_thres fragments, the REF as many beside it (main.cpp:832-837)
        if (P.min_altdp_thres > 0 && !(mask & (mask - 1u))) return 0;
        totBDP = masked_bdepths(R, st, x, mask, bd);
    }
    bool pos_rescued = false;
    if (P.tumor_vcf_is_provided && C.n_tkeys) { const long long q = tkey_lower_bound(C, refpos, 0); pos_rescued = (q < C.n_tkeys && C.tkeys[q].refpos == refpos); }
    int n = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k >= nsym) break;
        const int s = st_symbol(st, k);
        if (gate_b(P, s, refsymbol, bd[k], totBDP, C.all_out, pos_rescued)) { long long first; int src; n += allele_source(C, P.tumor_vcf_is_provided, refpos, s, first, src); }
    }
    return n;
}

#define GS_BLOCK 256
#define GS_ITEMS 8
#define GS_TILE (GS_BLOCK * GS_ITEMS)
// groups per thread of the gate (measured on the 1 Mb tile: 8 -> 58 us, 4 -> 69 us, 2 -> 84 us: more tiles cost more in tickets and look-back
// than the shorter chains of dependent loads per thread give back)
#ifndef GATE_ITEMS
#define GATE_ITEMS 8
#endif
#define GATE_TILE (GS_BLOCK * GATE_ITEMS)
// gate + record slots + active list in one launch.  offsets[g] = packed (number of active groups << 32 | number of records) in front of g.
__global__ void __launch_bounds__(GS_BLOCK) k_gate_scan(RegionDev R, UvcParams P, ScoreCtx C, Stage S, long long *total_records) {
    __shared__ long long sh_cnt[GATE_TILE + GATE_TILE / 8];   // one pad word per 8: thread t reads 8 consecutive words, 9 apart from its neighbour's
    __shared__ unsigned long long sh_wave[4], sh_prefix;
    __shared__ int sh_tile;
    if (threadIdx.x == 0) sh_tile = (int)atomicAdd(&S.cnt[CNT_ticket1], 1u);
    __syncthreads();
    const int tile = sh_tile;
    const long long ngroups = 2LL * (C.pos_end - C.pos_beg);
    const long long base = (long long)tile * GATE_TILE;
    // counted striped (consecutive lanes = consecutive groups = consecutive positions of the depth planes), scanned blocked
#pragma unroll 1
    for (int i = 0; i < GATE_ITEMS; i++) {
        const int j = i * GS_BLOCK + (int)threadIdx.x;
        const long long g = base + j;
        const long long n = (g < ngroups ? (long long)gate_count(R, P, C, g) : 0LL);
        sh_cnt[j + (j >> 3)] = n | (n > 0 ? (1LL << 32) : 0LL);
    }
    __syncthreads();
    long long v[GATE_ITEMS]; unsigned long long s = 0;
    { const int j0 = (int)threadIdx.x * GATE_ITEMS;
#pragma unroll
      for (int i = 0; i < GATE_ITEMS; i++) { v[i] = sh_cnt[j0 + i + (j0 >> 3)]; s += (unsigned long long)v[i]; } }
    unsigned long long total = 0;
    const unsigned long long excl = block_excl_scan256(s, sh_wave, total);
    if (threadIdx.x < 64) { const unsigned long long p = cs_lookback(S.status1, tile, total, (int)threadIdx.x, R.err); if (threadIdx.x == 0) sh_prefix = p; }
    __syncthreads();
    long long run = (long long)(sh_prefix + excl);
    const long long g0 = base + (long long)threadIdx.x * GATE_ITEMS;
#pragma unroll
    for (int i = 0; i < GATE_ITEMS; i++) {
        const long long g = g0 + i;
        if (g < ngroups) { C.offsets[g] = run; if (PK_FLAGS(v[i])) C.active[PK_FLAGS(run)] = (int)g; }
        run += v[i];
    }
    if (g0 <= ngroups - 1 && ngroups - 1 < g0 + GATE_ITEMS) { C.offsets[ngroups] = (long long)(sh_prefix + total); *total_records = PK_COUNT((long long)(sh_prefix + total)); }   // the thread that owns the last group
}

// One thread per active group: the group's scalars, the symbols that have anything at this position (type_mask), one header per record.
__global__ void __launch_bounds__(128) k_enum(RegionDev R, UvcParams P, ScoreCtx C, Stage S) {
    const long long ngroups = 2LL * (C.pos_end - C.pos_beg);
    const long long n_active = PK_FLAGS(C.offsets[ngroups]);
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= n_active || gi >= S.cap) return;
    const long long g = C.active[gi];
    const long long rec0 = PK_COUNT(C.offsets[g]), nrec = PK_COUNT(C.offsets[g + 1]) - rec0;
    if (nrec <= 0 || rec0 + nrec > C.capacity || rec0 + nrec > S.cap) { GR_(nrec, gi) = 0; return; }   // does not fit: the host sees the count and comes back
    const int zpos = C.pos_beg + (int)(g >> 1), st = (int)(g & 1);
    const int refpos = (st == UVC_BASE_SYMBOL ? zpos - 1 : zpos);
    const int64_t x = refpos - R.beg;
    const int refidx = zpos - R.beg, refsize = (int)R.npos - 1;
    const int refsymbol = group_refsymbol(R, zpos, st), nsym = st_count(st);
    unsigned mask = type_mask(R, st, x, refsymbol);
    int bd[8], cd[8];
    const int totBDP = masked_bdepths(R, st, x, mask, bd);
    // deduplicated depth of the marked symbols (main.cpp:806-812)
#pragma unroll
    for (int k = 0; k < 8; k++) cd[k] = 0;
    for (unsigned m = mask & 0xFFu; m; m &= m - 1u) {
        const int k = __builtin_ctz(m), s = st_symbol_at(st, k);
        const int v = imax(FAP(R, 0, UVC_FAM_cDP1, s, x), FAP(R, 0, UVC_FAM_cDP12, s, x)) + imax(FAP(R, 1, UVC_FAM_cDP1, s, x), FAP(R, 1, UVC_FAM_cDP12, s, x));
#pragma unroll
        for (int kk = 0; kk < 8; kk++) cd[kk] = (kk == k ? v : cd[kk]);
    }
    { const int kr = st_index(st, refsymbol); int rb = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) rb = (k == kr ? bd[k] : rb);
      GR_(refbdp, gi) = rb; }   // fragment depth of the reference symbol (main.cpp:1099)
    // Plane groups the gather need not touch at all (bits 8.. of the mask word): the FAMINFO / DUPLEX cells of a symbol are zero throughout a
    // 4 096-position block nobody marked (RegionDev::dirty -- what the zero fill relies on too), and cIAQ / cIAD / cIDQ of a strand exist only
    // behind a P5 bucket of that (strand, position) (k_p5b).  uvcgpu_region_check_presence holds the planes against both statements.
    { const size_t blk = (size_t)(x >> UVC_DIRTY_SHIFT);
      unsigned fi = 0, du = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
          const int s = st_symbol(st, k < nsym ? k : nsym - 1);
          const unsigned f1 = R.dirty[((size_t)NSYM + s) * R.ndblk + blk], f2 = R.dirty[((size_t)2 * NSYM + s) * R.ndblk + blk];
          if ((mask >> k) & 1u) { fi |= f1; du |= f2; }
      }
      if (fi) mask |= GM_FI;
      if (du) mask |= GM_DUP;
      if (R.p5flag[x]) mask |= GM_P5F;
      if (R.p5flag[(size_t)R.npos + x]) mask |= GM_P5R; }
    GR_(x, gi) = (int)x; GR_(zpos, gi) = zpos; GR_(st, gi) = st; GR_(refsym, gi) = refsymbol; GR_(mask, gi) = (int)mask; GR_(rec0, gi) = (int)rec0; GR_(nrec, gi) = (int)nrec;
    // homopolymer context for minABQ (main.cpp:623-626, 909-928)
    const int prev1 = ((refidx >= 2) ? (int)R.refsym[refidx - 2] : UVC_BASE_NN), prev2 = ((refidx >= 3) ? (int)R.refsym[refidx - 3] : UVC_BASE_NN);
    const int next1 = ((refidx < refsize) ? (int)R.refsym[refidx] : UVC_BASE_NN), next2 = ((refidx + 1 < refsize) ? (int)R.refsym[refidx + 1] : UVC_BASE_NN);
    const bool hp1 = (prev1 == refsymbol && next1 == refsymbol), hp2 = (prev2 == refsymbol && next2 == refsymbol);
    GR_(hp, gi) = (hp1 ? (hp2 ? 20 : 10) : 0);
    const int nrtr = (int)R.npos;
    const RtrLite rtr1 = load_rtr(R, imax(refpos - R.beg, 3) - 3), rtr2 = load_rtr(R, imin(refpos - R.beg + 3, nrtr - 1));
    GR_(r1t, gi) = rtr1.tracklen; GR_(r1u, gi) = rtr1.unitlen; GR_(r1a, gi) = rtr1.anyTR_tracklen; GR_(r2t, gi) = rtr2.tracklen; GR_(r2u, gi) = rtr2.unitlen; GR_(r2a, gi) = rtr2.anyTR_tracklen;
    // the records of the group, in the order of k_gate_scan's count (symbols in SYMBOL_TYPE_TO_SYMBOLS order, alleles in list order)
    long long rec = rec0;
    bool indel_rec = false;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k >= nsym) break;
        const int s = st_symbol(st, k);
        if (!gate_b(P, s, refsymbol, bd[k], totBDP, C.all_out, true /* an active group of a normal sample is a rescued position */)) continue;
        long long first; int src;
        const int m = allele_source(C, P.tumor_vcf_is_provided, refpos, s, first, src);
        indel_rec = indel_rec || is_ins(s) || is_del(s);
        for (int ai = 0; ai < m && rec < rec0 + nrec; ai++, rec++) {
            RH_(gi, rec) = (int)gi; RH_(symbol, rec) = s; RH_(src, rec) = src; RH_(idx, rec) = (src ? (int)(first + ai) : -1); RH_(bdepth, rec) = bd[k]; RH_(cdepth, rec) = cd[k];
        }
    }
    // InDel depths of the LINK group at zerobased_pos (main.cpp:817-831), shared by both groups of this zerobased_pos
    // (calc_qual reads them, and the repeat context below, for InDel alleles only: main.hpp:5113-5190, 5228-5236, 5292)
    int ins_cdepth = 0, del_cdepth = 0, ins1_cdepth = 0, del1_cdepth = 0;
    if (indel_rec) {
        const int64_t xz = zpos - R.beg;
#pragma unroll
        for (int k = 1; k < 7; k++) {
            const int s = st_symbol(UVC_LINK_SYMBOL, k);
            const int cdz = imax(FAP(R, 0, UVC_FAM_cDP1, s, xz), FAP(R, 0, UVC_FAM_cDP12, s, xz)) + imax(FAP(R, 1, UVC_FAM_cDP1, s, xz), FAP(R, 1, UVC_FAM_cDP12, s, xz));
            if (is_ins(s)) { ins_cdepth += cdz; if (UVC_LINK_I1 == s) ins1_cdepth += cdz; } else { del_cdepth += cdz; if (UVC_LINK_D1 == s) del1_cdepth += cdz; }
        }
    }
    GR_(insc, gi) = ins_cdepth; GR_(delc, gi) = del_cdepth; GR_(ins1c, gi) = ins1_cdepth; GR_(del1c, gi) = del1_cdepth;
    int ru_size = 0, repeatnum = 0;
    if (indel_rec) indel_context(R, refidx, P.indel_str_repeatsize_max, ru_size, repeatnum);
    GR_(rusize, gi) = ru_size; GR_(repnum, gi) = repeatnum;
    if (rec != rec0 + nrec) atomicExch(R.err, UVCGPU_EDEVICE);   // the two enumerations disagree: cannot happen
    atomicMax(&S.cnt[CNT_nvalid], (unsigned)(rec0 + nrec));
}

// One thread per (active group, plane) x GATHER_PLANES planes: the plane's cells of the symbols that can be non-zero at this position are
// fetched once; their sum is the symbol-type total (fill_symboltype_fmt, main.hpp:3745-3793), each record of the group takes its own symbol's
// cell (BcfFormat_symbol_init, main.hpp:4094-4251; a symbol outside the mask has nothing here: 0 without a load).  All loads of a thread
// are independent (a symbol outside the mask re-reads the first one's cell, same line, instead of branching around a load).
#define GATHER_PLANES 4
__global__ void __launch_bounds__(256) k_gather(RegionDev R, ScoreCtx C, Stage S) {
    const long long ngroups = 2LL * (C.pos_end - C.pos_beg);
    const long long n_active = PK_FLAGS(C.offsets[ngroups]);
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= n_active || gi >= S.cap) return;
    const int nrec = GR_(nrec, gi);
    if (nrec == 0) return;
    const int64_t x = GR_(x, gi);
    const int st = GR_(st, gi), rec0 = GR_(rec0, gi);
    const unsigned mask = (unsigned)GR_(mask, gi);
    const int kf = __builtin_ctz(mask);   // (never empty: the dense symbol is in it)
    long long c[GATHER_PLANES][8];
#pragma unroll
    for (int q = 0; q < GATHER_PLANES; q++) {
        const int u = (int)blockIdx.y * GATHER_PLANES + q;
        if (u >= NGATHER) break;
        const GatherDesc d = c_gather.d[u];
        if (d.grp <= SG_PREP64) { c[q][0] = stage_cell(R, d.grp, d.plane, 0, x); continue; }
        const unsigned need = gather_need(d.grp, d.plane);
        if (need && !(mask & need)) {   // nothing of this plane group at this position: zeros without a load
#pragma unroll
            for (int k = 0; k < 8; k++) c[q][k] = 0;
            continue;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) c[q][k] = stage_cell(R, d.grp, d.plane, st_symbol(st, ((mask >> k) & 1u) ? k : kf), x);
    }
#pragma unroll
    for (int q = 0; q < GATHER_PLANES; q++) {
        const int u = (int)blockIdx.y * GATHER_PLANES + q;
        if (u >= NGATHER) break;
        const GatherDesc d = c_gather.d[u];
        if (d.grp <= SG_PREP64) { ROW_(long long, S.tot, d.tot, S.cap, gi) = c[q][0]; continue; }
        if (d.tot >= 0) {
            long long v = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) v += (((mask >> k) & 1u) ? c[q][k] : 0LL);
            if (d.trunc) v = (long long)(int)v;   // an int32 FORMAT field truncates the int64 sum on assignment (bcf_formats_generator1.cpp:220-245)
            ROW_(long long, S.tot, d.tot, S.cap, gi) = v;
        }
        for (int j = 0; j < nrec; j++) {
            const int k = st_index(st, RH_(symbol, rec0 + j));
            long long v = 0;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) v = (kk == k && ((mask >> kk) & 1u)) ? c[q][kk] : v;
            if (d.al < NAL32) ROW_(int32_t, S.al, d.al, S.cap, rec0 + j) = (int32_t)v; else ROW_(long long, S.al64, d.al - NAL32, S.cap, rec0 + j) = v;
        }
    }
}
#define FO_(fld) ROW_(int32_t, fields, UVC_O_##fld, capacity, rec)
// what every per-record kernel starts with
#define REC_PROLOGUE \
    const long long rec = (long long)blockIdx.x * blockDim.x + threadIdx.x; \
    if (rec >= (long long)S.cnt[CNT_nvalid]) return; \
    const long long gi = RH_(gi, rec); \
    int32_t *fields = C.fields; const long long capacity = C.capacity; \
    const int symbol = RH_(symbol, rec), refsymbol = GR_(refsym, gi); \
    RtrLite rtr1, rtr2; rtr1.tracklen = GR_(r1t, gi); rtr1.unitlen = GR_(r1u, gi); rtr1.anyTR_tracklen = GR_(r1a, gi); rtr2.tracklen = GR_(r2t, gi); rtr2.unitlen = GR_(r2u, gi); rtr2.anyTR_tracklen = GR_(r2a, gi);
// the tumor record of a rescued allele: tpfa of calc_DPv (main.cpp:935)
DEV double rec_tpfa_dpv(const ScoreCtx &C, int src, int idx) { if (src != 2) return -1.0; const UvcTumorKey &tk = C.tkeys[idx]; return (double)(tk.cDP1x + 1) / (double)(tk.CDP1x + 2); }
// the integer context of calc_DPv that both halves need (cheap: no division)
#define DPV_COMMON \
    const bool tprov = P.tumor_vcf_is_provided; \
    const double unbias_ratio = (!tprov ? 1.0 : sqrt(2.0)); \
    const int pcr_dp = (int)T_APDP5, a_dp = (int)T_APDP0, near_pcr_clip = (int)T_APDP9; \
    const bool strong_amp = (pcr_dp * 100 > a_dp * 50), weak_amp = (pcr_dp * 100 > a_dp * 30); \
    const bool is_rescued = (tpfa >= 0);   /* main.hpp:4297-4298 */ \
    const double pfa = (is_rescued ? tpfa : 0.5), c2altpc = 0.025; \
    const int ADP1 = (int)(T_ADPff0 + T_ADPfr0 + T_ADPrf0 + T_ADPrr0); \
    const int aDP = (f_aDPff + f_aDPfr + f_aDPrf + f_aDPrr); \
    const int ADP = imax(ADP1, near_pcr_clip); \
    const int cDP1 = f_cDP1f + f_cDP1r, CDP1 = T_CDP1b0 + T_CDP1b1; \
    const int sumCDP2 = T_CDP2b0 + T_CDP2b1, sumCDP1 = CDP1; \
    const bool nmore_amp = (!tprov ? strong_amp : weak_amp), tmore_amp = (!tprov ? weak_amp : strong_amp); \
    const int normCDP1 = (T_CDP12b0 + T_CDP12b1) + 1, normBDP = (T_BDPb0 + T_BDPb1) + 1; \
    const int c2DP = f_cDP2f + f_cDP2r;

// BcfFormat_symbol_init's derived values (fill_symbol_VQ_fmts, main.hpp:3820-3887) + BcfFormat_symbol_calc_DPv (main.hpp:4274-4844) up to its contingency tests
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 8))) k_dpv_pre(UvcParams P, ScoreCtx C, Stage S) {
    REC_PROLOGUE
    const int src = RH_(src, rec), idx = RH_(idx, rec);
    const int st = GR_(st, gi), zpos = GR_(zpos, gi);
    const int refpos = (st == UVC_BASE_SYMBOL ? zpos - 1 : zpos);
    // which allele this record is (main.cpp:853-935)
    int bDPa = RH_(bdepth, rec), cDP0a = RH_(cdepth, rec), glen = 0, tki_tier2 = 0, gap_row = -1, tkey_idx = -1;
    if (src == 2) {   // tumor record: main.cpp:935, 985-986
        const UvcTumorKey &tk = C.tkeys[idx];
        tkey_idx = idx; tki_tier2 = tk.tier2;
        if (is_ins(symbol) || is_del(symbol)) glen = tk.indel_len;
    } else if (is_ins(symbol) || is_del(symbol)) {
        if (src == 1) { const UvcIndelAllele &al = C.alleles[idx]; bDPa = al.bDPa; cDP0a = al.cDP0a; glen = al.indel_len; gap_row = C.allele_rows[idx]; }
        else {   // no fragment carries this symbol here: "Invalid indel detected", the allele is the symbol's description text (main.hpp:5415-5423)
            bDPa = 0; cDP0a = 0;
            glen = ((symbol == UVC_LINK_D3P || symbol == UVC_LINK_I3P) ? 6 : 5);   // strlen("<LD3P>") / strlen("<LD2>") etc., main_conversion.hpp:336-346
        }
    }
    const double tpfa = rec_tpfa_dpv(C, src, idx);
    const int minABQ_snv = (C.is_amplicon ? P.syserr_minABQ_pcr_snv : P.syserr_minABQ_cap_snv), minABQ_indel = (C.is_amplicon ? P.syserr_minABQ_pcr_indel : P.syserr_minABQ_cap_indel);
    const int minABQ = (is_subst(symbol) ? (int)nnminus(minABQ_snv, GR_(hp, gi)) : minABQ_indel);
    const int f_gap_len = glen;
    int f_bMQ, f_aBQQ, f_aBQ;
    {   // fill_symbol_VQ_fmts
        const int a2BQf = AL_(v2BQf), a2BQr = AL_(v2BQr);
    const int aDPf = f_aDPff + f_aDPrf, aDPr = f_aDPfr + f_aDPrr;
    const int ADP = (int)(T_ADPff0 + T_ADPrf0 + T_ADPfr0 + T_ADPrr0);
    const int rssf = (int)(aDPf * sqrt((double)(((long long)a2BQf * SQR_QUAL_DIV) / imax(1, aDPf))));
    const int rssr = (int)(aDPr * sqrt((double)(((long long)a2BQr * SQR_QUAL_DIV) / imax(1, aDPr))));
    const int rssb = (int)((aDPf + aDPr) * sqrt((double)((a2BQf + a2BQr) * SQR_QUAL_DIV / imax(1, aDPf + aDPr))));
    const double t = dmax(0.0, ((aDPf + aDPr + 0.5) * 2.0 / (ADP + 1.0) - 1.0));
    int minABQa = minABQ - (int)(5 * 10.0 * (t * t));
    const double sbratio = (double)(imax(aDPf, aDPr) * 10 + 10) / (double)(imin(aDPf, aDPr) * 10 + 10);
    minABQa += ibetween((int)(sbratio * sbratio) - P.syserr_BQ_sbratio_q_add, 0, P.syserr_BQ_sbratio_q_max);
    const int xmratio = (P.syserr_BQ_xmratio_q_max * 10 * (aDPf + aDPr) / imax(1, f_a2XM2));
    const int bmratio = (P.syserr_BQ_bmratio_q_max * 10 * (aDPf + aDPr) / imax(1, f_a2BM2));
    minABQa += ibetween(xmratio - P.syserr_BQ_xmratio_q_add, 0, P.syserr_BQ_xmratio_q_max) + ibetween(bmratio - P.syserr_BQ_bmratio_q_add, 0, P.syserr_BQ_bmratio_q_max);
    const int m = P.syserr_BQ_strand_favor_mul;
    const int q_fw = (rssf * m - minABQa * aDPf * m / 10 + rssr - minABQa * aDPr / 10) / m;
    const int q_rv = (rssr * m - minABQa * aDPr * m / 10 + rssf - minABQa * aDPf / 10) / m;
    const int q_2d = rssb - minABQa * (aDPf + aDPr) / 10;
    const int a_rmsBQ = rssb / imax(1, aDPf + aDPr);
        const int bMQraw = AL_(vbMQ);
    f_bMQ = (int)round(sqrt((double)(((long long)bMQraw * SQR_QUAL_DIV) / imax(f_bDPf + f_bDPr, 1))) + (double)(1.0 - FLT_EPS));
    f_aBQQ = imax(a_rmsBQ, P.syserr_BQ_prior + imax(q_2d, imax(q_fw, q_rv)));
        f_aBQ = a_rmsBQ;
        FO_(a2BQf) = rssf; FO_(a2BQr) = rssr; FO_(aBQ) = a_rmsBQ; FO_(aBQQ) = f_aBQQ; FO_(bMQ) = f_bMQ;
    }
    FO_(refpos) = refpos; FO_(symbol) = symbol; FO_(refsymbol) = refsymbol;
    FO_(DP) = T_CDP1b0 + T_CDP1b1; FO_(bDP) = T_BDPb0 + T_BDPb1; FO_(c2DP) = T_CDP2b0 + T_CDP2b1; FO_(c2AD) = f_cDP2f + f_cDP2r;
    FO_(bDPa) = bDPa; FO_(cDP0a) = cDP0a; FO_(gapSa) = gap_row; FO_(gapSa_len) = glen; FO_(tkey) = tkey_idx;
    // ---- calc_DPv ----
    DPV_COMMON
    const double unbias_qualadd = (!tprov ? 0 : 3);
    const int allprior = (!tprov ? 0 : 31);
    double cbP = 1e-9, cbBQ = 1e-9, dir_bias_div = 1.0;
    if ((nmore_amp && (0x2 == (0x2 & P.nobias_flag))) || ((!nmore_amp) && (0x1 == (0x1 & P.nobias_flag)))) {
        const double oddsA_bias = prob2odds((aDP - f_aP1 + 0.5) / (ADP - T_AP10 + 1.0));
        const double oddsA_nobias = prob2odds((f_aP1 + 0.5) / (T_AP10 + 1.0));
        const bool pos_cb = ((oddsA_bias * P.microadjust_counterbias_pos_odds_ratio < oddsA_nobias * (unbias_ratio - DBL_EPS))
                && (f_aP1 * (unbias_ratio - DBL_EPS) > aDP - f_aP1)
                && ((ADP - T_AP10) * P.microadjust_counterbias_pos_fold_ratio * (unbias_ratio - DBL_EPS) > T_AP10)
                && ((0 == P.primerlen && 0 != P.primerlen2) || !is_subst(symbol)));
        if (pos_cb) cbP = dmax(cbP, (f_aP1 + 0.5) / (lmax(T_AP10, (long long)near_pcr_clip) + 1.0)); else cbP = dmax(cbP, 2e-9);
        if (is_subst(symbol)) {
            const bool f_good = ((T_ADPfr0 + T_ADPrr0) + 150 <= (T_ADPff0 + T_ADPrf0) * 5 * unbias_ratio);
            const bool r_good = ((T_ADPff0 + T_ADPrf0) + 150 <= (T_ADPfr0 + T_ADPrr0) * 5 * unbias_ratio);
            const int avg_f_aBQ = (f_a1BQf / imax(1, f_aDPff + f_aDPrf)), avg_r_aBQ = (f_a1BQr / imax(1, f_aDPfr + f_aDPrr));
            const int avg_f_ABQ = (int)(T_A1BQf0 / lmax(1, T_ADPff0 + T_ADPrf0)), avg_r_ABQ = (int)(T_A1BQr0 / lmax(1, T_ADPfr0 + T_ADPrr0));
            if ((f_a1BQf >= f_a1BQr) && (f_good && r_good) && (avg_f_aBQ + unbias_qualadd >= avg_r_ABQ + 14) && (avg_r_ABQ <= 14 + unbias_qualadd))
                cbBQ = dmax(cbBQ, (f_aDPff + f_aDPrf + 0.5) / (T_ADPff0 + T_ADPrf0 + 1.0));
            if ((f_a1BQr >= f_a1BQf) && (f_good && r_good) && (avg_r_aBQ + unbias_qualadd >= avg_f_ABQ + 14) && (avg_f_ABQ <= 14 + unbias_qualadd))
                cbBQ = dmax(cbBQ, (f_aDPfr + f_aDPrr + 0.5) / (T_ADPfr0 + T_ADPrr0 + 1.0));
        } else dir_bias_div = (1.0 + (unsigned)f_gap_len / (unsigned)P.indel_str_repeatsize_max);
    }
    const long long aDPgap = nnminus(lmax(T_APDP1, T_APDP2), f_aP3);
    const double aDPFAgap = ((rtr1.tracklen + rtr2.tracklen < P.indel_str_repeatsize_max) ? 1.0 : ((f_aP3 + pfa) / (aDPgap + 1.0)));
    const double aDPFA1 = ((aDP + pfa) / (ADP + 1.0));
    const double labelFA = (f_aP2 + 1.5 + f_aP2) / (T_AP20 + 2.0 + f_aP2);
    const double aDPFA = dmin((is_subst(symbol) ? dmin(aDPFA1, dmax(aDPFA1 / 3, aDPFAgap)) : aDPFA1), labelFA * (ADP + 1.0) / (T_AP20 + 0.5) * unbias_ratio);
    const int aDPplus = (is_subst(symbol) ? 0 : ((aDP + 1) * P.bias_prior_DPadd_perc / 100));
    const double dp_coef = ((symbol == UVC_LINK_M) ? dmax(P.contam_any_mul_frac, 1.0 - imax(rtr1.tracklen, rtr2.tracklen) / (lmax(1, lmax(T_ALPL0, T_ARPL0)) / dmax(1.0 / 150.0, (double)T_ABQ20))) : 1.0);
    double aPprior = P.bias_priorfreq_pos, aBprior = P.bias_priorfreq_pos;
    const bool in_indel_read = ((T_APXM1) / 15.0 * P.microadjust_bias_pos_indel_fold * (P.bias_prior_var_DP_mul) > (aDP + aDPplus) * dp_coef);
    const bool in_indel_len = (lmax(T_APDP1, T_APDP2) * (P.bias_prior_var_DP_mul) > (aDP + aDPplus) * dp_coef);
    const bool in_indel_rtr = (lmax(T_APDP3, T_APDP4) * (P.bias_prior_var_DP_mul) > (aDP + aDPplus) * dp_coef);
    const bool in_rtr = (imax(rtr1.tracklen, rtr2.tracklen) > round(P.indel_polymerase_size));
    const bool in_dnv_read = ((UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) && (T_APDP7 * 2 > T_APDP6));
    if (in_indel_read || in_dnv_read || ((is_ins(symbol) || is_del(symbol)) && (T_APXM0 > T_APXM1 * P.microadjust_bias_pos_indel_misma_to_indel_ratio))) {
        aPprior -= P.bias_priorfreq_indel_in_read_div; aBprior -= P.bias_priorfreq_indel_in_read_div;
    }
    if (UVC_LINK_M != symbol && UVC_LINK_NN != symbol) {
        double maxpf = 0;
        if (in_indel_len) maxpf = dmax(maxpf, P.bias_priorfreq_indel_in_var_div2);
        if (in_indel_rtr) maxpf = dmax(maxpf, P.bias_priorfreq_indel_in_str_div2);
        if (in_rtr) maxpf = dmax(maxpf, P.bias_priorfreq_var_in_str_div2);
        aBprior -= maxpf; aPprior -= maxpf;
    }
    aPprior += allprior; aBprior += allprior;
    FO_(nPF0) = (int)round(aPprior); FO_(nPF1) = (int)round(aBprior);
    const double aIprior = (is_subst(symbol) ? P.bias_priorfreq_ipos_snv : P.bias_priorfreq_ipos_indel) + allprior;
    const int homopol_len = ((1 == rtr1.unitlen) ? rtr1.tracklen : 0) + ((1 == rtr2.unitlen) ? rtr2.tracklen : 0);
    const double aSBprior = (is_subst(symbol)
            ? (imin((int)nnminus(f_aBQ, (((UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) && (homopol_len > 0)) ? imin(5 * homopol_len, 20) : 0)), f_bMQ) + P.bias_priorfreq_strand_snv_base)
            : (P.bias_priorfreq_strand_indel)) + allprior;
    const double dedup_A2C1 = dmin(1.0, (double)imax(CDP1, P.bias_reduction_by_high_sequencingDP_min_n_totDepth) / (double)imax(ADP1, 1));
    const double dedup_a2c1 = dmin(1.0, (double)imax(cDP1, P.bias_reduction_by_high_sequencingDP_min_n_altDepth) / (double)imax(aDP, 1));
    const double dff = dmax(dedup_A2C1, dedup_a2c1);
    const double pc_read = (in_indel_read ? P.bias_FA_pseudocount_indel_in_read : 0.5);
    const int f_tier2 = (is_rescued ? (tki_tier2 ? 1 : 0) : (((c2DP >= 2) && (normBDP * P.fam_bias_overseq_perc >= normCDP1 * 100) && (T_APDP11 * 100 > (long long)a_dp * 50)) ? 1 : 0));   // main.hpp:4475
    FO_(tier2) = f_tier2;
    const double cFA2L = (f_tier2 ? (((double)(((long long)f_c2LP0 * f_c2LP0) * 2 / lmax(1, (long long)imin(c2DP, f_c2LP0 * 4))) + c2altpc) / (T_C2LP00 + 1.0)) : 1.0);
    const double cFA2R = (f_tier2 ? (((double)(((long long)f_c2RP0 * f_c2RP0) * 2 / lmax(1, (long long)imin(c2DP, f_c2RP0 * 4))) + c2altpc) / (T_C2RP00 + 1.0)) : 1.0);
    const double ori_base = (is_subst(symbol) ? P.bias_priorfreq_orientation_snv_base : P.bias_priorfreq_orientation_indel_base) + allprior;
    const double te = dmax(aDPFA, P.bias_orientation_min_effective_allelefrac);
    const double ori_all = log(te * te) + phred2nat(ori_base);
    MID_(cbP, rec) = cbP; MID_(cbBQ, rec) = cbBQ; MID_(dirdiv, rec) = dir_bias_div; MID_(aDPFA, rec) = aDPFA; MID_(cFA2L, rec) = cFA2L; MID_(cFA2R, rec) = cFA2R;
    MID_(dff, rec) = dff; MID_(pcread, rec) = pc_read; MID_(aPprior, rec) = aPprior; MID_(aBprior, rec) = aBprior; MID_(aIprior, rec) = aIprior; MID_(aSBprior, rec) = aSBprior; MID_(oriall, rec) = ori_all;
    (void)sumCDP1; (void)sumCDP2; (void)cDP1; (void)tmore_amp;
}

// The 14 dp4_to_pcFA evaluations of a record (main.hpp:4409-4600) are independent of each other: one thread per (record, test).  A block
// holds 64 records; wave w runs test w + 7 * blockIdx.y of them, so a wave's test is uniform and its loads are rows of 64 consecutive records.
#define DP4_WAVES 7
__global__ void __launch_bounds__(64 * DP4_WAVES) __attribute__((amdgpu_waves_per_eu(4, 8))) k_dp4(UvcParams P, ScoreCtx C, Stage S) {
    const long long rec = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    if (rec >= (long long)S.cnt[CNT_nvalid]) return;
    const long long gi = RH_(gi, rec);
    int32_t *fields = C.fields; const long long capacity = C.capacity;
    const int aDP = (f_aDPff + f_aDPfr + f_aDPrf + f_aDPrr);
    const int ADP = imax((int)(T_ADPff0 + T_ADPfr0 + T_ADPrf0 + T_ADPrr0), (int)T_APDP9);
    const double dff = MID_(dff, rec), pl = P.powlaw_exponent, c2altpc = 0.025;
    {   // (no loop over the wave's two tests -- blockIdx.y picks the half: hoisted per-field addresses of both iterations spilled 106 registers)
        const int t = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + DP4_WAVES * (int)blockIdx.y;
        // the arguments of test t (the wave's t is uniform), then ONE dp4_to_pcFA body for all fourteen
        bool bidir = false, noseq = false, skip = false;
        double a1 = 0, a2 = 0, d1 = 0, d2 = 0, nats = 0, k1 = -1, k2 = -1, pa = 0.5, pd = 1.0;
        switch (t) {
        case 0: case 1: case 2: case 3: {   // position / BAQ bias of the read ends
            const double aBQ2d = (double)imax(1, f_aBQ2), ABQ2d = (double)lmax(1, T_ABQ20);
            a2 = aDP; d2 = ADP; pa = MID_(pcread, rec);
            if (t == 0) { a1 = f_aLP1; d1 = T_ALP20 + f_aLP1 - f_aLP2; nats = phred2nat(MID_(aPprior, rec)); k1 = lmax(1, f_aLPL) / aBQ2d; k2 = lmax(1, T_ALPL0) / ABQ2d; }
            else if (t == 1) { a1 = f_aRP1; d1 = T_ARP20 + f_aRP1 - f_aRP2; nats = phred2nat(MID_(aPprior, rec)); k1 = lmax(1, f_aRPL) / aBQ2d; k2 = lmax(1, T_ARPL0) / ABQ2d; }
            else if (t == 2) { a1 = f_aLB1; d1 = T_ALB20 + f_aLB1 - f_aLB2; nats = phred2nat(MID_(aBprior, rec)); k1 = lmax(1, f_aLBL) / aBQ2d; k2 = lmax(1, T_ALBL0) / ABQ2d; }
            else { a1 = f_aRB1; d1 = T_ARB20 + f_aRB1 - f_aRB2; nats = phred2nat(MID_(aBprior, rec)); k1 = lmax(1, f_aRBL) / aBQ2d; k2 = lmax(1, T_ARBL0) / ABQ2d; }
        } break;
        case 4: case 5: case 6: case 7: {   // the same on the tier-2 families
            if (!FO_(tier2)) { skip = true; break; }   // (the results stay 1.0)
            const int c2DP = f_cDP2f + f_cDP2r, sumCDP2 = T_CDP2b0 + T_CDP2b1;
            const double c2Pp = dmax(0.0, MID_(aPprior, rec)), c2Bp = dmax(0.0, MID_(aBprior, rec));
            const double cb = (double)imax(1, f_c2BQ2), CB = (double)lmax(1, T_C2BQ20);
            noseq = true; a2 = c2DP; d2 = sumCDP2; pa = c2altpc; pd = 1.0;
            if (t == 4) { a1 = f_c2LP1; d1 = T_C2LP20 + f_c2LP1 - f_c2LP2; nats = phred2nat(c2Pp); k1 = lmax(1, f_c2LPL) / cb; k2 = lmax(1, T_C2LPL0) / CB; }
            else if (t == 5) { a1 = f_c2RP1; d1 = T_C2RP20 + f_c2RP1 - f_c2RP2; nats = phred2nat(c2Pp); k1 = lmax(1, f_c2RPL) / cb; k2 = lmax(1, T_C2RPL0) / CB; }
            else if (t == 6) { a1 = f_c2LB1; d1 = T_C2LB20 + f_c2LB1 - f_c2LB2; nats = phred2nat(c2Bp); k1 = lmax(1, f_c2LBL) / cb; k2 = lmax(1, T_C2LBL0) / CB; }
            else { a1 = f_c2RB1; d1 = T_C2RB20 + f_c2RB1 - f_c2RB2; nats = phred2nat(c2Bp); k1 = lmax(1, f_c2RBL) / cb; k2 = lmax(1, T_C2RBL0) / CB; }
        } break;
        case 8: {
            const double ALpd = (T_ALI20 + 0.5) / (T_ADPfr0 + T_ADPrr0 - T_ALI20 + 0.5);
            const double aLpd = (f_aLI1 + ALpd / (1.0 + ALpd)) / (f_aDPfr + f_aDPrr - f_aLI1 + 1.0 / (1.0 + ALpd));
            a1 = f_aLI1; a2 = (f_aDPfr + f_aDPrr); d1 = (T_ALI20 + f_aLI1 - f_aLI2); d2 = (T_ADPfr0 + T_ADPrr0); nats = phred2nat(MID_(aIprior, rec)); k1 = aLpd; k2 = ALpd; pa = 0.25; pd = 0.5;
        } break;
        case 9: {
            const double ARpd = (T_ARI20 + 0.5) / (T_ADPff0 + T_ADPrf0 - T_ARI20 + 0.5);
            const double aRpd = (f_aRI1 + ARpd / (1.0 + ARpd)) / (f_aDPff + f_aDPrf - f_aRI1 + 1.0 / (1.0 + ARpd));
            a1 = f_aRI1; a2 = (f_aDPff + f_aDPrf); d1 = (T_ARI20 + f_aRI1 - f_aRI2); d2 = (T_ADPff0 + T_ADPrf0); nats = phred2nat(MID_(aIprior, rec)); k1 = aRpd; k2 = ARpd; pa = 0.25; pd = 0.5;
        } break;
        case 10: bidir = true; a1 = f_aRIf; a2 = f_aLIr; d1 = T_ARIf0; d2 = T_ALIr0; nats = phred2nat(MID_(aSBprior, rec)); break;
        case 11: bidir = true; a1 = f_cDP1f; a2 = f_cDP1r; d1 = T_CDP1b0; d2 = T_CDP1b1; nats = MID_(oriall, rec); break;
        case 12: if (!P.bias_is_orientation_artifact_mixed_with_sequencing_error) { skip = true; break; }
                 bidir = true; a1 = f_cDP12f; a2 = f_cDP12r; d1 = T_CDP12b0; d2 = T_CDP12b1; nats = MID_(oriall, rec); break;
        default: bidir = true; noseq = true; a1 = f_cDP2f; a2 = f_cDP2r; d1 = T_CDP2b0; d2 = T_CDP2b1; nats = MID_(oriall, rec); pa = c2altpc; pd = 1.0; break;
        }
        double r2[2] = { 1.0, 1.0 };
        if (!skip) dp4(r2, bidir, noseq, dff, a1, a2, d1, d2, pl, nats, k1, k2, pa, pd);
        D4_(t, 0, rec) = r2[0]; D4_(t, 1, rec) = r2[1];
    }
}

#define SHORT_FRAG(wgs_min) ((T_APLRI0 + T_APLRI2) < (T_APLRI1 + T_APLRI3) * (long long)(wgs_min))   // does_fmt_imply_short_frag, main.hpp:169-174
// BcfFormat_symbol_calc_DPv behind its contingency tests: the bias-reduced allele fractions, their minima, FTS, cDP1v .. cDP2x (main.hpp:4409-4844)
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 8))) k_dpv_post(UvcParams P, ScoreCtx C, Stage S) {
    REC_PROLOGUE
    const double tpfa = rec_tpfa_dpv(C, RH_(src, rec), RH_(idx, rec));
    DPV_COMMON
    const int f_symbol = symbol, f_gap_len = FO_(gapSa_len), f_bDPa = FO_(bDPa), f_cDP0a = FO_(cDP0a), f_tier2 = FO_(tier2), f_bMQ = FO_(bMQ);
    int f_AD = f_cDP1f + f_cDP1r, f_bAD = f_bDPf + f_bDPr;
    int f_bNMQ, f_cDP1v, f_cDP1w, f_cDP1x, f_cDP2v, f_cDP2w, f_cDP2x;
    const double cFA2 = (f_cDP2f + f_cDP2r + c2altpc) / (sumCDP2 + 1.0);
    const double cFA3 = (f_cDP3f + f_cDP3r + c2altpc) / ((T_CDP3b0 + T_CDP3b1) + 1.0);
    const double cbP = MID_(cbP, rec), cbBQ = MID_(cbBQ, rec), dir_bias_div = MID_(dirdiv, rec), aDPFA = MID_(aDPFA, rec), cFA2L = MID_(cFA2L, rec), cFA2R = MID_(cFA2R, rec);
    double aLPFA = D4_(0, 0, rec), aRPFA = D4_(1, 0, rec), aLBFA = D4_(2, 0, rec), aRBFA = D4_(3, 0, rec);
    double c2LPFA = D4_(4, 0, rec), c2RPFA = D4_(5, 0, rec), c2LBFA = D4_(6, 0, rec), c2RBFA = D4_(7, 0, rec);   // 1.0 without tier 2
    const double LI2[2] = { D4_(8, 0, rec), D4_(8, 1, rec) }, RI2[2] = { D4_(9, 0, rec), D4_(9, 1, rec) }, SS2[2] = { D4_(10, 0, rec), D4_(10, 1, rec) };
    double RO1[2] = { D4_(11, 0, rec), D4_(11, 1, rec) };
    const double RO2[2] = { D4_(13, 0, rec), D4_(13, 1, rec) };
    if (P.bias_is_orientation_artifact_mixed_with_sequencing_error) {
        if ((T_ADPff0 * 8 >= ADP) && (T_ADPfr0 * 8 >= ADP) && (T_ADPrf0 * 8 >= ADP) && (T_ADPrr0 * 8 >= ADP)) { RO1[0] = D4_(12, 0, rec); RO1[1] = D4_(12, 1, rec); }
    }
    double aLIFA = LI2[0] * (tmore_amp ? dir_bias_div : dmax(dir_bias_div, aDPFA / LI2[1]));
    double aRIFA = RI2[0] * (tmore_amp ? dir_bias_div : dmax(dir_bias_div, aDPFA / RI2[1]));
    const double aSIFA = dmax((f_aLI1 + 0.5) / (T_ALI20 + f_aLI1 - f_aLI2 + 1.0), (f_aRI1 + 0.5) / (T_ARI20 + f_aRI1 - f_aRI2 + 1.0));
    const int indel_size = f_gap_len;
    if (is_ins(symbol) || is_del(symbol)) {
        const double coef = imax(1, f_bDPa) / (double)imax(1, f_bDPf + f_bDPr);
        const bool major_reg = ((lmax(T_APDP1, T_APDP3) + lmax(T_APDP2, T_APDP4)) * 0.5 * (1.0 + (double)FLT_EPS) < aDP * coef);
        if ((imin(indel_size, P.microadjust_nobias_pos_indel_maxlen) * aDPFA * coef >= P.nobias_pos_indel_lenfrac_thres) ||
            (imax(rtr1.tracklen, rtr2.tracklen) >= P.nobias_pos_indel_str_track_len && major_reg && !(T_APXM0 > T_APXM1 * P.microadjust_nobias_pos_indel_misma_to_indel_ratio))) {
            aLPFA += 2.0; aRPFA += 2.0; aLBFA += 2.0; aRBFA += 2.0;
            if (f_tier2) { c2LPFA += 2.0; c2RPFA += 2.0; c2LBFA += 2.0; c2RBFA += 2.0; }
        }
        if (f_bMQ >= P.microadjust_nobias_pos_indel_bMQ && f_a2XM2 * 100 >= aDP * 100 * P.microadjust_nobias_pos_indel_perc) { aLIFA += 2.0; aRIFA += 2.0; }
    } else if (UVC_LINK_M == symbol || UVC_LINK_NN == symbol) {
        const double pc = P.bias_FA_pseudocount_indel_in_read;
        aLBFA = dmin(aLBFA, (pc + f_aLB1) / (double)(pc * 2 + ADP));
        aRBFA = dmin(aRBFA, (pc + f_aRB1) / (double)(pc * 2 + ADP));
    } else if (refsymbol == symbol) { aLIFA = aRIFA = dmax(aLIFA, aRIFA); }
    const long long avg_sqr = lmax(T_APXM4 / lmax(1, T_APDP1), T_APXM5 / lmax(1, T_APDP2));
    if ((!is_subst(symbol)) && ((long long)P.microadjust_nobias_pos_indel_maxlen * P.microadjust_nobias_pos_indel_maxlen < avg_sqr)
        && (UVC_LINK_M == symbol || UVC_LINK_NN == symbol || ((long long)(indel_size * 2) * (indel_size * 2) < avg_sqr))) {
        const double pc = P.bias_FA_pseudocount_indel_in_read;
        const double aLmin = (pc + f_aLP1) / (double)(pc * 2 + T_ALP10), aRmin = (pc + f_aRP1) / (double)(pc * 2 + T_ALP10);   // sic: ALP1 in both (main.hpp:4575-4576)
        aLPFA = dmin(aLPFA, aLmin); aRPFA = dmin(aRPFA, aRmin);
        if (f_tier2) { c2LPFA = dmin(c2LPFA, aLmin); c2RPFA = dmin(c2RPFA, aRmin); }
    }
    if (tprov || (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform)) aLIFA = aRIFA = dmax(aLIFA, aRIFA);
    const double aPFFA = (f_aPF1 + pfa * 100.0) / (T_APF20 + (f_aPF1 - f_aPF2) + 100.0);
    double aSSFA = SS2[0] * dir_bias_div, cROFA1 = RO1[0] * dir_bias_div, cROFA2 = RO2[0] * dir_bias_div;
    if (is_ins(symbol) || is_del(symbol)) { f_bAD = imin(f_bAD, f_bDPa); f_AD = imin(f_AD, f_cDP0a); }
    const double bFA = (f_bDPa + pfa) / ((T_BDPb0 + T_BDPb1) + 1.0);
    const double cFA0 = (f_cDP0a + pfa * (SHORT_FRAG(P.lib_wgs_min_avg_fraglen) ? P.lib_nonwgs_ad_pseudocount : 1.0)) / (sumCDP1 + 1.0);
    if ((T_ADPfr0 + T_ADPrr0) * P.microadjust_nobias_strand_all_fold < (T_ADPff0 + T_ADPrf0) * unbias_ratio) { aLIFA += 4.0; aSSFA += 4.0; }
    if ((T_ADPff0 + T_ADPrf0) * P.microadjust_nobias_strand_all_fold < (T_ADPfr0 + T_ADPrr0) * unbias_ratio) { aRIFA += 4.0; aSSFA += 4.0; }
    const double aLPFA2 = dmax(aDPFA * 0.01, aLPFA), aRPFA2 = dmax(aDPFA * 0.01, aRPFA), aLBFA2 = dmax(aDPFA * 0.01, aLBFA), aRBFA2 = dmax(aDPFA * 0.01, aRBFA);
    const double c2LPFA2 = dmax(cFA2 * 0.01, c2LPFA), c2RPFA2 = dmax(cFA2 * 0.01, c2RPFA), c2LBFA2 = dmax(cFA2 * 0.01, c2LBFA), c2RBFA2 = dmax(cFA2 * 0.01, c2RBFA);
    const double aLIFA2 = dmax(aDPFA * 0.01, aLIFA), aRIFA2 = dmax(aDPFA * 0.01, aRIFA), aSSFA2 = dmax(aDPFA * 0.05, aSSFA);
    cROFA1 = dmax(aDPFA * 1e-4, cROFA1); cROFA2 = dmax(aDPFA * 1e-4, cROFA2);
    const double fBTA = (double)((T_BTAb0 + T_BTAb1) + 200), fBTB = (double)((T_BTBb0 + T_BTBb1) + 6);
    const double fbTA = (double)(f_bTAf + f_bTAr + 100), fbTB = (double)(f_bTBf + f_bTBr + 3);
    const long long sl = lmin(
            lmin(lmax(0, f_aLIT / lmax(1, (long long)(f_aDPfr + f_aDPrr)) - P.microadjust_longfrag_sidelength_min), (long long)P.microadjust_longfrag_sidelength_max),
            lmin(lmax(0, f_aRIT / lmax(1, (long long)(f_aDPff + f_aDPrf)) - P.microadjust_longfrag_sidelength_min), (long long)P.microadjust_longfrag_sidelength_max));
    const double sidelen_frac = 1.0 - sl / P.microadjust_longfrag_sidelength_zeroMQpenalty;
    const double _alt_frac = fbTB / fbTA;
    const double alt_frac = (nmore_amp ? (dmax(0.0, _alt_frac - 0.2) * 1.25) : _alt_frac);
    const double nonalt_frac = (fBTB + P.contam_any_mul_frac * fbTB - fbTB) / (fBTA + P.contam_any_mul_frac * fbTA - fbTA);
    const double frac_mut = dmax(P.syserr_MQ_NMR_expfrac, P.syserr_MQ_NMR_altfrac_coef * alt_frac * sidelen_frac - P.syserr_MQ_NMR_nonaltfrac_coef * nonalt_frac);
    f_bNMQ = (int)round(numstates2phred(pow(frac_mut / P.syserr_MQ_NMR_expfrac, (P.syserr_MQ_NMR_pl_exponent))) * (frac_mut));
    FO_(bNMQ) = f_bNMQ; FO_(bNMa) = (int)round(100 * alt_frac); FO_(bNMb) = (int)round(100 * nonalt_frac);
    const bool tmore_primer = (tmore_amp || ((P.primerlen > 0) && !(0x4 & P.primer_flag)));
    double t1only = dmin(cROFA1, dmin(aLPFA2, dmin(aRPFA2, dmin(aLBFA2, dmin(aRBFA2, cFA0)))));
    t1only = dmin(t1only, dmin(aDPFA * dbetween(1.0 + aDPFA - alt_frac, 0.1, 1.0), aPFFA * aSSFA2 / dmax(aSSFA2, SS2[1])));
    const double t1plus = dmin(aSSFA2, dmin(aLIFA2, dmin(aRIFA2, dmin(dmax(aDPFA * 0.01, aSIFA), bFA))));
    const double cFA2a = ((tmore_primer && !is_rescued) ? (cFA2 * (P.powlaw_amplicon_allele_fraction_coef)) : cFA2);
    const double cFA3a = ((normBDP * 100 > normCDP1 * ((P.fam_tier3DP_bias_overseq_perc - 100) / (is_rescued ? 2 : 1) + 100)) ? cFA3 : 1.0);
    const double c23FA = cFA2a;
    const double t2only = dmin(cROFA2, dmin(c2LPFA2, dmin(c2RPFA2, dmin(c2LBFA2, dmin(c2RBFA2, dmin(cFA2a, dmin(cFA3a, dmin(cFA2L, cFA2R))))))));
    OUT(UVC_O_nNFA0, -numstates2deciphred(cbP)); OUT(UVC_O_nNFA1, -numstates2deciphred(cbBQ)); OUT(UVC_O_nNFA2, -numstates2deciphred(aDPFA));
    OUT(UVC_O_nNFA3, -numstates2deciphred(bFA)); OUT(UVC_O_nNFA4, -numstates2deciphred(cFA0)); OUT(UVC_O_nNFA5, -numstates2deciphred(cFA2));
    int FTS = 0, bit = 0;
    unsigned pct0 = 0, pct1 = 0, pct2 = 0, pct3 = 0, pct4 = 0;
    auto push = [&](int fld, double refFA, double biasFA) {   // fmt_bias_push, main.hpp:4258-4272
        OUT(fld, -numstates2deciphred(biasFA));
        if (biasFA < refFA * P.bias_thres_FTS_FA) {
            FTS |= (1 << bit);
            const unsigned v = (unsigned)imin(imax((int)round(100.0 * biasFA / refFA), 0), 255) << (8 * (bit & 3));
            const int q = bit >> 2;
            if (q == 0) pct0 |= v; else if (q == 1) pct1 |= v; else if (q == 2) pct2 |= v; else if (q == 3) pct3 |= v; else pct4 |= v;
        }
        bit++;
    };
    push(UVC_O_nAFA0, aDPFA, aSSFA2); push(UVC_O_nAFA1, aDPFA, aPFFA); push(UVC_O_nAFA2, aDPFA, aSIFA); push(UVC_O_nAFA3, aDPFA, aLBFA2); push(UVC_O_nAFA4, aDPFA, aRBFA2);
    push(UVC_O_nAFA5, aDPFA, aLPFA2); push(UVC_O_nAFA6, aDPFA, aRPFA2); push(UVC_O_nAFA7, aDPFA, aLIFA2); push(UVC_O_nAFA8, aDPFA, aRIFA2);
    push(UVC_O_nBCFA0, bFA, cFA0); push(UVC_O_nBCFA1, cFA0, bFA); push(UVC_O_nBCFA2, cFA0, cROFA1); push(UVC_O_nBCFA3, cFA2, cROFA2);
    push(UVC_O_nBCFA4, cFA2, c2LPFA2); push(UVC_O_nBCFA5, cFA2, c2RPFA2); push(UVC_O_nBCFA6, cFA2, c2LBFA2); push(UVC_O_nBCFA7, cFA2, c2RBFA2);
    push(UVC_O_nBCFA8, cFA2, cFA2L); push(UVC_O_nBCFA9, cFA2, cFA2R);
    OUT(UVC_O_FTS, FTS);
    OUT(UVC_O_FTSpct0, (int)pct0); OUT(UVC_O_FTSpct1, (int)pct1); OUT(UVC_O_FTSpct2, (int)pct2); OUT(UVC_O_FTSpct3, (int)pct3); OUT(UVC_O_FTSpct4, (int)pct4);
    const double aNCFA = ((!tprov && SHORT_FRAG(P.lib_wgs_min_avg_fraglen) && (is_ins(symbol) || is_del(symbol)) && indel_size >= P.lib_nonwgs_clip_penal_min_indelsize)
            ? dmax((f_aNC + 0.5) / (ADP + 1.0), dbetween((f_cDP1f + f_cDP1r) / 300.0, 1.0 / 3.0, 2.0 / 3.0) * aDPFA) : 2.0);
    const double cb_normalgerm = ((!tprov || !SHORT_FRAG(P.lib_wgs_min_avg_fraglen)) ? 1e-9
            : dbetween(aPFFA * aPFFA * (1.0 / P.lib_nonwgs_normal_full_self_rescue_fa), aPFFA * P.lib_nonwgs_normal_min_self_rescue_fa_ratio, aPFFA));
    const double cbFA = dmax(cbP, dmax(cbBQ, cb_normalgerm));
    const double dedup_FA = (!tprov ? dmin(bFA, cFA0) : dmax(bFA, cFA0));
    const double frac_umi2seg = dmin(1.0, dmin(c23FA / aDPFA, aDPFA / c23FA));
    double refbias = 0;
    if ((is_ins(f_symbol) || is_del(f_symbol)) && is_rescued) {   // main.hpp:4804-4810
        const int isz = f_gap_len;
        const int noinfo = (isz * (is_ins(f_symbol) ? 2 : 1) + imax(isz, imax(rtr1.tracklen, rtr2.anyTR_tracklen)));
        refbias = (double)(noinfo) / ((double)(lmin(T_ALPL0, T_ARPL0) * 2 + noinfo) / (double)(T_ABQ20 + 0.5));
        refbias = dmin(refbias, P.microadjust_refbias_indel_max);
    }
    f_cDP1v = (int)(norm_fa(dmax(dmin(dmin(t1plus, t1only), aNCFA), cbFA), refbias) * sumCDP1 * 100);
    f_cDP1w = (int)(norm_fa(dmax(dmin(aLPFA2, dmin(aRPFA2, dmin(aLBFA2, dmin(aRBFA2, dmin(bFA, aNCFA))))), cbFA), refbias) * sumCDP1 * 100);
    double abc_x = dmin(aPFFA, dedup_FA);
    if (tprov) abc_x = dmax(abc_x, cbFA);
    f_cDP1x = 1 + (int)(abc_x * sumCDP1 * 100);
    const double cFA2c = cFA2 * cFA2 * cFA2;
    const double c2XB = dbetween(3.0 * c2LBFA2 * c2RBFA2 * aSSFA2 / cFA2c, dmin(c2LBFA2, c2RBFA2) / 8.0, dmin(c2LBFA2, c2RBFA2));
    const double c2XP = dbetween(3.0 * c2LPFA2 * c2RPFA2 * aSSFA2 / cFA2c, dmin(c2LPFA2, c2RPFA2) / 8.0, dmin(c2LPFA2, c2RPFA2));
    const double c2XX = dmin(c2XB, c2XP);
    f_cDP2v = (int)(norm_fa(dmax(dmin(dmin(t1plus, dmin(t2only, c2XX)), aNCFA), cbFA * frac_umi2seg), refbias) * sumCDP2 * 100);
    f_cDP2w = (int)(norm_fa(dmax(dmin(c2LPFA2, dmin(c2RPFA2, dmin(c2XX, dmin(c2LBFA2, dmin(c2RBFA2, dmin(cFA2, aNCFA)))))), cbFA * frac_umi2seg), refbias) * sumCDP2 * 100);
    f_cDP2x = 1 + (int)(dmin(aPFFA, c23FA) * sumCDP2 * 100);
    OUT(UVC_O_cDP1v, f_cDP1v); OUT(UVC_O_cDP1w, f_cDP1w); OUT(UVC_O_cDP1x, f_cDP1x); OUT(UVC_O_cDP2v, f_cDP2v); OUT(UVC_O_cDP2w, f_cDP2w); OUT(UVC_O_cDP2x, f_cDP2x);
    OUT(UVC_O_AD, f_AD); OUT(UVC_O_bAD, f_bAD);
    (void)cDP1; (void)RO1[1]; (void)RO2[1]; (void)a_dp; (void)c2DP;
}

// BcfFormat_symbol_sum_DPv (main.hpp:4888-4906) over the records of the group + BcfFormat_symbol_calc_qual (main.hpp:4908-5343)
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 8))) k_qual(UvcParams P, ScoreCtx C, Stage S) {
    REC_PROLOGUE
    // sum_DPv: [0] = the sum over the group's alleles, [1] = the NN allele's value
    int CDP1v0 = 0, CDP1x0 = 0;
    {
        const long long r0 = GR_(rec0, gi), r1 = r0 + GR_(nrec, gi);
        int s1[6] = { 0, 0, 0, 0, 0, 0 }, s2[6] = { 0, 0, 0, 0, 0, 0 };
        for (long long r = r0; r < r1; r++) {
            const int sy = fields[(size_t)UVC_O_symbol * (size_t)capacity + (size_t)r];
#pragma unroll
            for (int i = 0; i < 6; i++) { const int v = fields[(size_t)(UVC_O_cDP1v + i) * (size_t)capacity + (size_t)r]; s1[i] += v; if (UVC_BASE_NN == sy || UVC_LINK_NN == sy) s2[i] = v; }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) { OUT(UVC_O_CDP1v0 + 2 * i, s1[i]); OUT(UVC_O_CDP1v0 + 2 * i + 1, s2[i]); }
        CDP1v0 = s1[0]; CDP1x0 = s1[2];
    }
    const int src = RH_(src, rec);
    double tpfa = -1.0;   // tpfa of calc_qual (main.cpp:985-986)
    if (src == 2) { const UvcTumorKey &tk = C.tkeys[RH_(idx, rec)]; tpfa = (double)(tk.bDP + 0.5) / (double)(tk.BDP + 1.0); }
    const int ins_cdepth = GR_(insc, gi), del_cdepth = GR_(delc, gi), ins1_cdepth = GR_(ins1c, gi), del1_cdepth = GR_(del1c, gi), ru_size = GR_(rusize, gi), repeatnum = GR_(repnum, gi);
    const int f_gap_len = FO_(gapSa_len), f_cDP0a = FO_(cDP0a), f_tier2 = FO_(tier2), f_bMQ = FO_(bMQ), f_aBQQ = FO_(aBQQ), f_bNMQ = FO_(bNMQ);
    const int f_cDP1v = FO_(cDP1v), f_cDP1w = FO_(cDP1w), f_cDP1x = FO_(cDP1x), f_cDP2v = FO_(cDP2v), f_cDP2w = FO_(cDP2w);
    const bool tprov = P.tumor_vcf_is_provided, is_rescued = tprov;   // the caller passes IS_PROVIDED(vcf_tumor_fname), main.cpp:979
    const int indel_size = f_gap_len;
    const int sumCDP1 = T_CDP1b0 + T_CDP1b1, sumCDP2 = T_CDP2b0 + T_CDP2b1, sumBDP = T_BDPb0 + T_BDPb1, sumCDP12 = T_CDP12b0 + T_CDP12b1;
    const double cFA2 = (f_cDP2f + f_cDP2r + 0.5) / (sumCDP2 + 1.0);
    const int phrederr = sscs_phred(P, refsymbol, symbol) + (!tprov ? 0 : 4);
    const double umi_cFA = (((double)(f_cDP2v) + 0.5) / ((double)(sumCDP2 * 100 + 1.0)));
    const double umi_cFA_w = (((double)(f_cDP2w) + 0.5) / ((double)(sumCDP2 * 100 + 1.0)));
    const int inc1 = (int)(phrederr - (is_subst(symbol)
            ? (((UVC_BASE_A == refsymbol && UVC_BASE_T == symbol) || (UVC_BASE_T == refsymbol && UVC_BASE_A == symbol)) ? (double)P.fam_phred_pow_sscs_transversion_AT_TA_origin : P.fam_phred_pow_sscs_snv_origin)
            : P.fam_phred_pow_sscs_indel_origin));
    int inc4tn = (is_subst(symbol)
            ? (int)(imax(imax(P.fam_phred_sscs_transition_CG_TA, P.fam_phred_sscs_transition_AT_GC), imax(P.fam_phred_sscs_transversion_CG_AT, P.fam_phred_sscs_transversion_other)) - (P.fam_phred_pow_sscs_snv_origin))
            : inc1);
    const bool oxid = ((UVC_BASE_C == refsymbol && UVC_BASE_A == symbol) || (UVC_BASE_G == refsymbol && UVC_BASE_T == symbol));
    inc4tn += (oxid ? P.tn_q_inc_max_sscs_CG_AT : P.tn_q_inc_max_sscs_other);
    const double t2n = (tpfa > 0 ? tpfa : 0) * P.contam_t2n_mul_frac;
    const double contamfrac = P.contam_any_mul_frac + (1.0 - P.contam_any_mul_frac) * t2n;
    const int aDP = (f_aDPff + f_aDPfr + f_aDPrf + f_aDPrr);
    const int ADP = (int)(T_ADPff0 + T_ADPrf0 + T_ADPfr0 + T_ADPrr0);
    const int cDP0 = (f_cDP1f + f_cDP1r), CDP0 = sumCDP1, cDP2 = (f_cDP2f + f_cDP2r), CDP2 = sumCDP2;
    const int aavgMQ = (int)(f_aMQs / imax(1, aDP));
    const int diffAaMQs = (int)((T_AMQs0 - f_aMQs) / imax(1, ADP - aDP)) - aavgMQ;
    const int noUMI_inc = imin(P.bias_FA_powerlaw_noUMI_phred_inc_snv, aDP / 2);
    const double pl_noUMI = P.powlaw_anyvar_base + (is_subst(symbol) ? noUMI_inc : P.bias_FA_powerlaw_noUMI_phred_inc_indel);
    const int withUMI_inc = imin(P.bias_FA_powerlaw_withUMI_phred_inc_snv - P.bias_FA_powerlaw_noUMI_phred_inc_snv, cDP2 / 2) + noUMI_inc;
    const double pl_withUMI = P.powlaw_anyvar_base + (is_subst(symbol) ? withUMI_inc : P.bias_FA_powerlaw_withUMI_phred_inc_indel);
    const double prior_weight = 1.0 / (f_cDPmf + f_cDPmr + 1.0);
    const int thres_highBQ = (is_subst(symbol) ? P.fam_thres_highBQ_snv : P.fam_thres_highBQ_indel);
    const int cMmQ = (int)round(numstates2phred((f_cDPMf + f_cDPmf + f_cDPMr + f_cDPmr + pow(10.0, thres_highBQ / 10.0) * prior_weight) / (f_cDPmf + f_cDPmr + prior_weight)));
    const int nb1 = f_bIADb * 100 + 1, nb2 = imin(nb1, f_cDP1v + 1);
    const long long pq1 = 10 * f_bIAQb / imax(1, f_bIADb);
    const long long pq2 = pq1 + (long long)round(10 * numstates2phred((double)nb2 / (double)nb1));
    long long duped_binom = ((is_ins(symbol) || is_del(symbol)) ? pq1 : pq2) * nb2 / (10 * 100);
    const long long contam_frag_q = (long long)round(binom_llr(t2n, cDP0, CDP0 - cDP0)) + 9 - 3;
    const int h_snp = imax(0, 2 * P.germ_phred_hetero_snp - P.germ_phred_het3al_snp), h_indel = imax(0, 2 * P.germ_phred_hetero_indel - P.germ_phred_het3al_indel);
    int het3al_inc = (is_subst(symbol) ? h_snp : h_indel);
    if (is_ins(symbol) || is_del(symbol)) het3al_inc = (int)nnminus(h_indel + 1, indel_size);
    const int normcDP1 = (f_cDP12f + f_cDP12r + 1), normCDP1 = sumCDP12 + 1, normBDP = sumBDP + 1;
    const int ddiv = (is_rescued ? 2 : 1);
    const long long dec1a = (((P.fam_min_n_copies / ddiv <= normCDP1) || (P.fam_min_n_copies_DPxAD / ddiv <= (long long)normCDP1 * normcDP1)) ? 0 : (inc1 + 3));
    const long long dec1b = (((long long)((P.fam_min_overseq_perc - 100) / ddiv + 100) * normCDP1 <= (long long)100 * normBDP) ? 0 : (inc1 + 3));
    const long long dec1 = lmax(dec1a, dec1b);
    const long long dec2 = nnminus(thres_highBQ, cMmQ);
    const long long cIADnorm = (long long)(f_cIADf + f_cIADr) * 100 + 1;
    const long long cIADmin = lmin(cIADnorm, (long long)f_cDP2v + 1);
    const long long bq_fw = f_cIAQf + ((long long)f_cIAQr * imin(P.fam_phred_dscs_all - f_cIDQf, f_cIDQr)) / imax(f_cIDQr, 1);
    const long long bq_rv = f_cIAQr + ((long long)f_cIAQf * imin(P.fam_phred_dscs_all - f_cIDQr, f_cIDQf)) / imax(f_cIDQf, 1);
    const long long contam_sscs_q = (long long)round(binom_llr(t2n, cDP2, CDP2 - cDP2)) + 9 - 3;
    long long sscs_binom = ((long long)nnminus_d((double)lmax(bq_fw, bq_rv), numstates2phred(cIADnorm / (double)cIADmin) * cIADnorm / 100.0) * cIADmin) / (cIADnorm);
    if (lmax(bq_fw, bq_rv) > P.microadjust_fam_binom_qual_halving_thres && is_subst(symbol))
        sscs_binom = lmin(sscs_binom, P.microadjust_fam_binom_qual_halving_thres + (lmax(bq_fw, bq_rv) - P.microadjust_fam_binom_qual_halving_thres) / 2);
    sscs_binom -= dec1 + dec2;
    const double bcFA_v = (((double)(f_cDP1v) + 0.5) / (double)(sumCDP1 * 100 + 1.0));
    int pl_v = (int)round(P.powlaw_exponent * numstates2phred(bcFA_v) + (pl_noUMI));
    const double bcFA_w = (((double)(f_cDP1w) + 0.5) / (double)(sumCDP1 * 100 + 1.0));
    int pl_w = (int)round(P.powlaw_exponent * numstates2phred(bcFA_w) + (pl_noUMI) + P.tn_q_inc_max);
    const int ds_pl = (int)round(10 / log(10.0) * dmin(log((f_cDP12f + 0.5) / (T_CDP12b0 + 1.0)), log((f_cDP12r + 0.5) / (T_CDP12b1 + 1.0)))) + (phrederr);
    const int ds_binom = 3 * imin(f_cDP2f, f_cDP2r);
    const long long m5 = lmin(lmin(bq_fw, bq_rv), (long long)imin(ds_pl, imin(ds_binom, 3)));
    const int inc2 = (int)lmax(0, m5) * ((cFA2 > 0.002) ? 1 : 0);
    const int dec3 = (is_rescued ? (-3) : ((cFA2 >= 0.003) ? 0 : 5));
    const int base_2 = (int)(pl_withUMI + inc1 + inc2 - dec1 - dec2 - dec3);
    const int base_2tn = (int)(pl_withUMI + inc4tn + inc2 - dec1 - dec2 - dec3);
    int sscs_pl_v = (int)round((P.powlaw_exponent * numstates2phred(umi_cFA) + base_2));
    int sscs_pl_w = (int)round((P.powlaw_exponent * numstates2phred(umi_cFA_w) + base_2tn));
    const double dFA = (double)(f_dDP2 + 0.5) / (double)(T_DDP10 + 1.0);
    const double dSNR = (double)(f_dDP2 + 0.5) / (double)(f_dDP1 + 1.0);
    const double dnormFA = dFA * pow(dSNR, 1.0 / P.powlaw_exponent);
    const long long dscs_est = (long long)round((P.fam_phred_dscs_max + phrederr) / 2.0);
    const long long dFA_binom = (dscs_est - (long long)round(numstates2phred(1.0 / (dnormFA)))) * (long long)f_dDP2 * cIADmin / cIADnorm;
    const int dFA_pl = (int)(P.powlaw_anyvar_base + (dscs_est - P.fam_phred_pow_dscs_all_origin)
            + (int)round(numstates2phred((dnormFA) * dmin(1.0, (double)((f_cDP1v) + 0.5) / (double)(sumCDP1 * 100 + 1.0)))));
    OUT(UVC_O_cMmQ, cMmQ);
    const double eps = (double)FLT_EPS;
    const bool penal_applied = ((UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) && !tprov);
    const int penal_base = (penal_applied ? ((int)round(P.indel_multiallele_samepos_penal / log(2.0) * log((double)dmax(aDP + eps, (double)lmax(T_APDP1, T_APDP2)) / (double)(aDP + eps)))) : 0);
    int penal4multi = 0, penal4multi_g = 0, penal4multi_soma = 0, indel_UMI_penal = 0;
    if (indel_size > 0 && f_cDP0a > 0) {
        const double indel_pq = (double)imin(indel_phred_s(P.indel_polymerase_slip_rate, ru_size, repeatnum), 24) + 2 - (double)10;
        const int eff1 = (ru_size * imax(1, repeatnum) - ru_size);
        const int eff2 = (imax(rtr1.tracklen - rtr1.unitlen, rtr2.tracklen - rtr2.unitlen) / 3);
        const int effm = imax(eff1, eff2);
        const double indel_ic = numstates2phred((double)imax(indel_size + (is_ins(symbol) ? 1 : 0), 1) / (double)(effm + 1))
                + (is_ins(symbol) ? (numstates2phred(P.indel_del_to_ins_err_ratio) * imin(200, f_cDP0a) / 200) : 0);
        int ic = (is_ins(symbol) ? ins_cdepth : del_cdepth);
        if (UVC_LINK_D1 == symbol) ic += ins1_cdepth;
        if (UVC_LINK_I1 == symbol) ic = (int)(ic + del1_cdepth / P.indel_del_to_ins_err_ratio);
        const int nearInDelDP = (int)(is_ins(symbol) ? T_APDP1 : T_APDP2);
        int penal1 = (int)round(P.indel_multiallele_samepos_penal / log(2.0) * log((double)(ic + eps) / (double)(f_cDP0a + eps)));
        if (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) penal1 = (int)nnminus_d(penal1, P.indel_multiallele_samepos_penal);
        const int penal2 = (int)round(P.indel_multiallele_diffpos_penal / log(2.0) * log((double)(nearInDelDP + eps) / (double)(imax(aDP, nearInDelDP) + eps)));
        penal4multi_g = (int)((int)round(P.indel_tetraallele_germline_penal_value / log(2.0) * log((double)(ins_cdepth + del_cdepth + eps) / (double)(f_cDP0a + eps))) - P.indel_tetraallele_germline_penal_thres);
        if (is_ins(symbol)) { penal4multi = (penal1 * P.indel_ins_penal_pseudocount / (P.indel_ins_penal_pseudocount + indel_size)); penal4multi_soma = penal4multi; }
        else { penal4multi = imax(penal1, penal2); penal4multi_soma = penal1; }
        pl_v += (int)round(indel_ic); pl_w += (int)round(indel_ic);
        duped_binom += (long long)round(indel_pq);
        const long long sz = imax(indel_size, 1);
        const double sscs_ic = numstates2phred((double)(sz * sz) / (double)(effm + 1));
        const int ivd = (int)round(P.powlaw_exponent * numstates2phred(P.indel_del_to_ins_err_ratio));
        const int extra = (int)(nnminus_d(ivd, sscs_ic * (is_ins(symbol) ? 0 : effm) / round(P.indel_polymerase_size)) - (double)(ivd / 2));
        sscs_pl_v += (int)round(sscs_ic) + extra; sscs_pl_w += (int)round(sscs_ic) + extra;
        sscs_binom += (long long)round(indel_pq) + extra;
        if (f_tier2) indel_UMI_penal = (int)nnminus_d((sumBDP + 1.0) / (double)(sumCDP1 + 1.0) * P.fam_indel_nonUMI_phred_dec_per_fold_overseq,
                                                      (P.fam_thres_emperr_all_flat_indel + 1) * P.fam_indel_nonUMI_phred_dec_per_fold_overseq);
    }
    if (oxid && tprov) sscs_binom = lmax(sscs_binom, (long long)imin(aDP, 3));
    OUT(UVC_O_aAaMQ, diffAaMQs);
    const int readlenMQcap = (int)((T_APXM2) / lmax(1, T_APDP0) - 17);
    const int diffMQ = imax(0, diffAaMQs);
    const bool extra_accurate = (P.inferred_maxMQ > 60);
    const int MQVQadd = ((symbol == refsymbol) ? 0 : (imin(P.germ_phred_homalt_snp, ADP * 3)));
    const int MQVQadd_soma = ((symbol != refsymbol) ? 0 : (imin(P.germ_phred_homalt_snp, ADP * 3)));
    const bool MQ_unadj = (extra_accurate || (!is_subst(symbol)) || (aDP > ADP * 3 / 4));
    const int MQVQminus = (MQ_unadj ? 0 : ((int)nnminus((60 - 30), aavgMQ) * 2 / 5)) + ((MQ_unadj || (refsymbol != symbol)) ? 0 : (int)nnminus(imin(15, diffMQ), aavgMQ));
    int diffMQ2 = diffMQ;
    if (f_bMQ < 20 && !tprov) {
        const double axf = (f_aDPff + f_aDPrf + 0.5), axr = (f_aDPfr + f_aDPrr + 0.5), Axf = (T_ADPff0 + T_ADPrf0 + 1.0), Axr = (T_ADPfr0 + T_ADPrr0 + 1.0);
        if ((axr / Axr) * 2 < (axf / Axf) || (axf / Axf) * 2 < (axr / Axr)
            || (f_aLI1 + 0.5) / (T_ALI20 + 1.0) * (2 * (1.0 + DBL_EPS)) < (axr) / (Axr) || (f_aRI1 + 0.5) / (T_ARI20 + 1.0) * (2 * (1.0 + DBL_EPS)) < (axf) / (Axf)) diffMQ2 = imax(diffMQ2, 20 - imin(f_bMQ, 20));
    }
    const double MQ_base = ((f_bMQ * (P.syserr_MQ_max - P.syserr_MQ_nonref_base) / P.syserr_MQ_max + P.syserr_MQ_nonref_base)) - (int)(diffMQ2) - (int)(f_bNMQ);
    const int sysMQ = (((refsymbol == symbol) && (ADP > aDP * 2)) ? f_bMQ : (int)(MQ_base - (int)(numstates2phred((ADP + 1.0) / (aDP + 0.5)))));
    const bool nonWGS = SHORT_FRAG(P.lib_wgs_min_avg_fraglen);
    const int rescued_MQ = imin((int)nnminus(readlenMQcap, 60), (nonWGS ? P.lib_nonwgs_normal_max_rescued_MQ : P.lib_wgs_normal_max_rescued_MQ));
    int sysMQVQ1 = imin((imax(sysMQ, P.syserr_MQ_min) + MQVQadd), readlenMQcap);
    const int sysBQVQ = (((UVC_PLATFORM_IONTORRENT != P.inferred_sequencing_platform) && is_subst(symbol)) ? f_aBQQ : 200);
    const int pcr_dp = (int)T_APDP5;
    const bool strong_amp = ((pcr_dp * 100) > T_APDP0 * 50), weak_amp = ((pcr_dp * 100) > T_APDP0 * 30);
    const bool tmore_amp = (!tprov ? weak_amp : strong_amp);
    if (tmore_amp && (is_ins(symbol) || is_del(symbol)) && (sysMQVQ1 > 70) && (T_APXM1 / lmax(T_APDP0, 1) > 20))
        sysMQVQ1 = (int)(70 + ((sysMQVQ1 - 70) * 5 / (T_APXM1 / lmax(T_APDP0, 1) - 15)));
    int penal_add = 0;
    if (!tprov) {
        const long long delAPDP = lmax(T_APDP2, T_APDP4);
        const long long snv_dp = T_APDP6;
        if ((T_APDP0 < 3 * delAPDP) && (T_APDP0 < 3 * snv_dp) && (aDP * 3 < delAPDP) && (aDP * 3 < snv_dp) && is_subst(symbol) && (rtr2.tracklen >= 8 * rtr2.unitlen))
            penal_add = P.microadjust_germline_mix_with_del_snv_penalty;
        if (tmore_amp && is_del(symbol)) {
            if (aDP * 4 < T_APDP2) penal_add = imax(penal_add, 5);
            else if (f_cDP0a * 3 < 2 * (del_cdepth)) penal_add = imax(penal_add, 2);
        }
    }
    const int sysMQVQ = imax(0, sysMQVQ1);
    const int penal_base2 = penal_base + penal_add;
    const long long fx = T_ADPff0 + T_ADPfr0, rx = T_ADPrf0 + T_ADPrr0, xf = T_ADPff0 + T_ADPrf0, xr = T_ADPfr0 + T_ADPrr0;
    const bool frx = (lmax(fx, rx) > P.microadjust_strand_orientation_absence_DP_fold * (lmin(fx, rx) + 1));
    const bool xfr = (lmax(xf, xr) > P.microadjust_strand_orientation_absence_DP_fold * (lmin(xf, xr) + 1));
    const int v_minus = (is_subst(symbol) ? ((frx ? P.microadjust_orientation_absence_snv_penalty : 0) + (xfr ? P.microadjust_strand_absence_snv_penalty : 0)) : (tmore_amp ? P.microadjust_dedup_absence_indel_penalty : 0));
    const int tn_syserr_q = sysMQVQ + P.tn_q_inc_max + rescued_MQ;
    const int bIAQ = (int)(duped_binom - penal_base2), cIAQ = (int)(sscs_binom - penal_base);
    const int cPCQ1 = imin(pl_w - penal_base2, tn_syserr_q), cPLQ1 = pl_v - penal_base2 - v_minus;
    const int cPCQ2 = imin(sscs_pl_w - penal_base, tn_syserr_q), cPLQ2 = sscs_pl_v - penal_base;
    const int bTINQ = (int)(contam_frag_q + het3al_inc), cTINQ = (int)(contam_sscs_q + het3al_inc);
    OUT(UVC_O_bMQQ, sysMQVQ); OUT(UVC_O_bIAQ, bIAQ); OUT(UVC_O_cIAQ, cIAQ); OUT(UVC_O_cPCQ1, cPCQ1); OUT(UVC_O_cPLQ1, cPLQ1); OUT(UVC_O_cPCQ2, cPCQ2); OUT(UVC_O_cPLQ2, cPLQ2);
    OUT(UVC_O_bTINQ, bTINQ); OUT(UVC_O_cTINQ, cTINQ);
    const int aDPpc = ((refsymbol == symbol) ? 1 : 0);
    const long long d_ = imax(1, aDP + aDPpc);
    const int penal4BQerr = (is_subst(symbol) ? (5 + (int)(((long long)P.penal4lowdep) / (d_ * d_))) : 0);
    const int indel_q_inc = ((((!is_ins(symbol)) && (!is_del(symbol))) || is_rescued) ? 0 : indel_len_rusize_phred_s(indel_size, repeatnum));
    const double m3 = dmax(0.0, dmax(penal4multi - P.indel_multiallele_soma_penal_thres, (double)penal4multi_g));
    OUT(UVC_O_gVQ1, (int)dmax(0.0, indel_q_inc + imin(imin(sysBQVQ, (int)nnminus(sysMQVQ, MQVQminus)), imin(bIAQ - penal4BQerr, cPLQ1)) - 2 * m3));
    const int soma_minus = (is_rescued ? 0 : (15 - imin(ADP * 15 / 100, imin(aDP, 15))));
    const int sysVQsoma = (int)nnminus(imin(sysBQVQ, sysMQVQ + MQVQadd_soma), soma_minus);
    const int bcVQ1 = imin(sysVQsoma, imin(bIAQ - (is_rescued ? 0 : penal4BQerr), cPLQ1)) - penal4multi_soma;
    OUT(UVC_O_cVQ1, imax(0, imin(bcVQ1, bTINQ) - indel_UMI_penal));
    int mincVQ2 = 0;
    if (is_ins(symbol) || is_del(symbol)) {
        const int floor_v = (int)(dmin(P.germ_phred_homalt_indel + numstates2phred(umi_cFA), (double)(f_cDP2v * 3 / 100)) + (double)(((is_ins(symbol) ? 1 : 0) - 1) * 3));
        mincVQ2 = imax(mincVQ2, floor_v);
    }
    const long long dVQinc = lmin(lmin(dFA_binom, (long long)dFA_pl) - imax(0, imin(cIAQ, cPLQ2)), (long long)P.fam_phred_dscs_inc_max);
    OUT(UVC_O_dVQinc, (int)dVQinc);
    const int cVQ2 = (int)lmin((long long)sysVQsoma, lmin(cIAQ + lmax(0, dVQinc), cPLQ2 + lmax(0, dVQinc))) - penal4multi;
    OUT(UVC_O_cVQ2, imax(mincVQ2, imin(cVQ2, cTINQ)));
    const int cDP1y = (is_rescued ? f_cDP1x : f_cDP1v), CDP1y0 = (is_rescued ? CDP1x0 : CDP1v0);
    const double binom_contam = binom_llr(contamfrac, cDP1y, CDP1y0);
    const double power_contam = round(10.0 / log(10.0) * P.powlaw_exponent * dmax(logit2((cDP1y + 1) / (double)(CDP1y0 + 1), contamfrac), 0.0));
    OUT(UVC_O_CONTQ, (int)dmin(binom_contam, power_contam));
    (void)f_cDP1w; (void)f_cDP1x;
}
// ------------------------------------------------------------------------------------------------
// The calling step behind calc_qual -- main.cpp:990-1168, output_germline (main.hpp:5483-5775) and the arithmetic of append_vcf_record
// (main.hpp:6027-6272) -- on the records k_qual finished and the staged rows (no plane is read here):
//   k_call_group  one thread per active (zerobased_pos, symbol type) group: vAC, the two best non-reference alleles, the genotype
//   k_call_rec    one thread per record: NLODQ / TLODQ / SomaticQ / QUAL / FILTER / keep; vAC and "a GERMLINE line was written here" cross the
//                 two symbol types of a zerobased_pos: the other type's group is the neighbour in the active list
// ------------------------------------------------------------------------------------------------
#define FLD(fld, rec) fields[(size_t)(fld) * capacity + (rec)]
#define SYM_END UVC_NUM_SYMBOLS
DEV int het_lodq(double a1, double a2, double expfrac, double powlaw_exponent) {   // hetLODQ, main.hpp:5457-5462
    const int binomLODQ = (int)binom_llr(expfrac, a1, a2);
    const int powerLODQ = (int)round(10.0 / log(10.0) * powlaw_exponent * dmax(logit2((a1 + 0.5) * 0.5 / expfrac, (a2 + 0.5) * 0.5 / (1.0 - expfrac)), 0.0));
    return imin(binomLODQ, powerLODQ);
}
DEV int indel_n_units(int s) {   // SYMBOL_TO_INDEL_N_UNITS, main.hpp:271-279
    return s == UVC_LINK_D3P ? -3 : s == UVC_LINK_D2 ? -2 : s == UVC_LINK_D1 ? -1 : s == UVC_LINK_I3P ? 3 : s == UVC_LINK_I2 ? 2 : s == UVC_LINK_I1 ? 1 : 0;
}
// order of two InDel strings given as rows of the allele table (-1: no text, compares equal to everything)
DEV int gap_row_cmp(const ScoreCtx &C, int ra, int rb) {
    if (ra < 0 || rb < 0 || ra == rb) return 0;
    const UvcGapRow &a = C.gap_rows[ra], &b = C.gap_rows[rb];
    if (a.seq_off < 0 || b.seq_off < 0) return (a.len > b.len) - (a.len < b.len);   // deleted reference text from one start: the longer is the larger
    const int n = imin(a.len, b.len);
    for (int i = 0; i < n; i++) {
        const int ca = C.gap_seq[a.seq_off + i], cb = C.gap_seq[b.seq_off + i];
        const int xa = (ca == 4 ? 3 : (ca == 3 ? 4 : ca)), xb = (cb == 4 ? 3 : (cb == 3 ? 4 : cb));   // "ACGTN": A < C < G < N < T
        if (xa != xb) return xa < xb ? -1 : 1;
    }
    return (a.len > b.len) - (a.len < b.len);
}
DEV void normv_quals(int out[4], double tAD, double tDP, int tVQ, int tnVQcap, double nAD, double nDP, int nVQ, double penal_dimret_coef, int prior_phred, int tn_dec_by_xm, double powlaw_exponent) {   // main.hpp:5982-6009
    const int binom = (int)binom_llr((tDP - tAD) / (tDP), nDP - nAD, nAD);
    const double nADplus = nAD * dbetween(nDP / tDP - 1.0, 0.0, 1.0);
    const double bjpfrac = ((tAD + 0.5) / (tDP + 1.0)) / ((nAD + 0.5 + nADplus) / (nDP + 1.0 + nADplus));
    const int powlaw = (int)round(powlaw_exponent * numstates2phred(bjpfrac));
    const int tnVQinc = imax(-prior_phred, imax((-(int)nAD) * 3, imin(binom - prior_phred, powlaw - prior_phred)));
    const double l2 = log(dmax(bjpfrac, 1.001)) / log(2.0);
    int tnVQdec = imax(0, nVQ - imax(0, imin(binom - prior_phred, (int)(l2 * l2 * penal_dimret_coef))));
    tnVQdec = imax(tnVQdec, imin(nVQ + 9, tn_dec_by_xm));
    out[0] = binom; out[1] = powlaw; out[2] = tnVQdec; out[3] = imin(tnVQcap, tVQ + tnVQinc) - tnVQdec;
}
DEV void normv_quals2(int out[4], double tAD, double tDP, int tVQ, int tnVQcap, double nAD, double nDP, int nVQ) {   // main.hpp:6011-6025
    const int binom = (int)binom_llr((tDP - tAD) / (tDP), nDP - nAD, nAD);
    const int powlaw = (nAD <= 3 ? binom : (int)round(binom * 3 / nAD));
    const double m = dmax((double)(imin(binom, powlaw) - 3), dmax(-3 * nAD, -3.0));
    out[0] = binom; out[1] = powlaw; out[2] = nVQ; out[3] = (int)dbetween((double)tVQ + m - (double)nVQ, 0.0, (double)tnVQcap);
}

__global__ void __launch_bounds__(128) k_call_group(UvcParams P, ScoreCtx C, Stage S) {
    const long long ngroups = 2LL * (C.pos_end - C.pos_beg);
    const long long n_active = PK_FLAGS(C.offsets[ngroups]);
    const long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= n_active || gi >= S.cap) return;
    const long long nrec_ = GR_(nrec, gi);
    GR_(vAC, gi) = 0; GR_(gemit, gi) = 0; GR_(kept, gi) = 0;
    if (nrec_ == 0) return;
    int32_t *fields = C.fields; const long long capacity = C.capacity;
    const bool tprov = (P.tumor_vcf_is_provided != 0);
    const int st = GR_(st, gi), refsymbol = GR_(refsym, gi);
    const long long r0 = GR_(rec0, gi), r1 = r0 + nrec_;
    int vAC = 0;
    const int het3al = ((UVC_BASE_SYMBOL == st) ? P.germ_phred_het3al_snp : P.germ_phred_het3al_indel);
    // ---- vAC and the two best non-reference alleles (main.cpp:990-1016) ----
    long long top[2] = { -1, -1 };
    for (long long r = r0; r < r1; r++) {
        if (FLD(UVC_O_symbol, r) == refsymbol) continue;
        if (imax(FLD(UVC_O_cVQ1, r), FLD(UVC_O_cVQ2, r)) >= het3al) vAC += 1;
    }
    for (int k = 0; k < 2; k++) {
        long long best = -1;
        for (long long r = r0; r < r1; r++) {
            if (FLD(UVC_O_symbol, r) == refsymbol || r == top[0]) continue;
            if (best < 0) { best = r; continue; }
            const int v1 = FLD(UVC_O_cVQ1, r), v2 = FLD(UVC_O_cVQ2, r), b1 = FLD(UVC_O_cVQ1, best), b2 = FLD(UVC_O_cVQ2, best);
            const int s = FLD(UVC_O_symbol, r), bs = FLD(UVC_O_symbol, best);
            bool gt;   // tuple (max, VQ1, VQ2, symbol, string) greater than the best so far
            if (imax(v1, v2) != imax(b1, b2)) gt = imax(v1, v2) > imax(b1, b2);
            else if (v1 != b1) gt = v1 > b1;
            else if (v2 != b2) gt = v2 > b2;
            else if (s != bs) gt = s > bs;
            else gt = (gap_row_cmp(C, FLD(UVC_O_gapSa, r), FLD(UVC_O_gapSa, best)) > 0);
            if (gt) best = r;
        }
        top[k] = best;
    }
    // ---- output_germline ----
    // symbol_format_vec = the records except BASE_NN, padded with init_fmt to five entries, in descending gVQ1 (equal values keep
    // their order); ref = the best of {refsymbol, NN}, alt1..3 = the next three others
    int n_entries = 0;
    for (long long r = r0; r < r1; r++) if (FLD(UVC_O_symbol, r) != UVC_BASE_NN) n_entries++;
    const int n_pad = imax(0, 5 - n_entries);
    long long sel[4] = { -1, -1, -1, -1 };   // record, or -2 - k for the k-th padding allele
    bool have[4] = { false, false, false, false };
    for (long long r = r0; r < r1; r++) {
        const int s = FLD(UVC_O_symbol, r);
        if (s == UVC_BASE_NN || !(s == refsymbol || s == UVC_LINK_NN)) continue;
        if (!have[0] || FLD(UVC_O_gVQ1, r) > FLD(UVC_O_gVQ1, sel[0])) { sel[0] = r; have[0] = true; }
    }
    for (int k = 1; k <= 3; k++) {
        long long best = -1; int best_q = 0; bool found = false;
        for (long long r = r0; r < r1; r++) {
            const int s = FLD(UVC_O_symbol, r);
            if (s == UVC_BASE_NN || s == refsymbol || s == UVC_LINK_NN || r == sel[1] || r == sel[2]) continue;
            const int q = FLD(UVC_O_gVQ1, r);
            if (!found || q > best_q) { best = r; best_q = q; found = true; }
        }
        int pads_used = 0;
        for (int j = 1; j < k; j++) if (sel[j] <= -2) pads_used++;
        if (pads_used < n_pad && (!found || 0 > best_q)) { best = -2 - pads_used; found = true; }   // a padding allele (gVQ1 = 0) sorts behind the records with gVQ1 >= 0
        sel[k] = best; have[k] = found;
    }
    auto gq = [&](long long r, int fld, int pad) { return r >= 0 ? FLD(fld, r) : pad; };
    int a0 = gq(sel[0], UVC_O_gVQ1, 0), a1 = gq(sel[1], UVC_O_gVQ1, 0), a2 = gq(sel[2], UVC_O_gVQ1, 0), a3 = gq(sel[3], UVC_O_gVQ1, 0);
    const bool isSubst = is_subst(refsymbol);
    const int symbolNN = ((isSubst || !tprov) ? UVC_BASE_NN : UVC_LINK_NN);
    const int symb1 = gq(sel[1], UVC_O_symbol, SYM_END), symb2 = gq(sel[2], UVC_O_symbol, SYM_END);
    double ad0 = gq(sel[0], UVC_O_cDP1v, 50) / 100.0, ad1 = gq(sel[1], UVC_O_cDP1v, 50) / 100.0, ad2 = gq(sel[2], UVC_O_cDP1v, 50) / 100.0;
    if (symbolNN == symb1) { ad0 += ad1; ad1 = 0; }
    if (symbolNN == symb2) { ad0 += ad2; ad2 = 0; }
    const int a0a1 = het_lodq(ad0, ad1, 1.0 - P.germ_hetero_FA, P.powlaw_exponent), a1a0 = het_lodq(ad1, ad0, P.germ_hetero_FA, P.powlaw_exponent);
    const int a1a2 = het_lodq(ad1, ad2, 0.5, P.powlaw_exponent), a2a1 = het_lodq(ad2, ad1, 0.5, P.powlaw_exponent);
    const int phred_hetero = (isSubst ? P.germ_phred_hetero_snp : P.germ_phred_hetero_indel), phred_homalt = (isSubst ? P.germ_phred_homalt_snp : P.germ_phred_homalt_indel);
    const int phred_tri_al = (isSubst ? P.germ_phred_het3al_snp : P.germ_phred_het3al_indel);
    if (tprov) { a0 = imin(a0, gq(sel[0], UVC_O_CONTQ, 0)); a1 = imin(a1, gq(sel[1], UVC_O_CONTQ, 0)); a2 = imin(a2, gq(sel[2], UVC_O_CONTQ, 0)); a3 = imin(a3, gq(sel[3], UVC_O_CONTQ, 0)); }
    else a0 = imin(a0, gq(sel[0], UVC_O_CONTQ, 0));
    const int a2penal = imax(a2 - (phred_tri_al - phred_hetero), 0), a3penal = imax(a3 - phred_hetero, 0);
    const int a01hetp = imax(imax(a0a1, a1a0), 0), a12hetp = imax(imax(a1a2, a2a1) - 3, 0), a03trip = imax(a0, a3);
    int tri_al_penal = 0;
    if (is_ins(symb1) && is_ins(symb2)) { tri_al_penal += 3; if (symb1 == symb2) { tri_al_penal += 3; if (UVC_LINK_I3P == symb1) tri_al_penal += 3; } }
    { const int n1 = indel_n_units(symb1), n2 = indel_n_units(symb2); if (n1 != 0 && n2 != 0) tri_al_penal -= ibetween(abs(n1 - n2) * 3 - 5, 0, 9); }
    int GL4[4];
    GL4[0] = (0 - a1 - a2penal - a3penal);
    GL4[1] = (-phred_hetero - imax(a01hetp, a2) - imax(imin(a01hetp, a2) - phred_hetero, 0) - a3penal);
    GL4[2] = (-phred_homalt - imax(a0, a2) - imax(imin(a0, a2) - phred_hetero, 0) - a3penal);
    GL4[3] = (-phred_tri_al - imax(a12hetp, a03trip) - imax(imin(a12hetp, a03trip) - phred_hetero, 0) - imax(imin(a12hetp, imin(a0, a3)) - phred_hetero, 0) - tri_al_penal);
    const int ret = GL4[0] - imax(GL4[1], imax(GL4[2], GL4[3]));
    int i_best = 0, i_second = -1;   // descending (value, index): PairSecondLess over reverse iterators, main.hpp:5464-5469
    for (int i = 1; i < 4; i++) if (GL4[i] >= GL4[i_best]) i_best = i;
    for (int i = 0; i < 4; i++) if (i != i_best && (i_second < 0 || GL4[i] >= GL4[i_second])) i_second = i;
    const int germ_GQ = GL4[i_best] - GL4[i_second];
    int emit = ((0x1 & P.outvar_flag) ? 1 : 0);
    if (emit && 0 == i_best && (!P.should_output_all_germline) && imax(gq(sel[1], UVC_O_cDP0a, 0), gq(sel[2], UVC_O_cDP0a, 0)) <= 2) emit = 0;
    
    for (long long rec = r0; rec < r1; rec++) {
        for (int k = 0; k < 2; k++) {
            OUT(UVC_O_cVQ1M0 + k, top[k] >= 0 ? FLD(UVC_O_cVQ1, top[k]) : -999); OUT(UVC_O_cVQ2M0 + k, top[k] >= 0 ? FLD(UVC_O_cVQ2, top[k]) : -999);
            OUT(UVC_O_cVQAM0 + k, top[k] >= 0 ? FLD(UVC_O_symbol, top[k]) : SYM_END); OUT(UVC_O_cVQSM0 + k, top[k] >= 0 ? FLD(UVC_O_gapSa, top[k]) : -1);
        }
        OUT(UVC_O_vNLODQ, ret);
        for (int i = 0; i < 4; i++) OUT(UVC_O_GL4_0 + i, GL4[i]);
        OUT(UVC_O_GST0, a0); OUT(UVC_O_GST1, a1); OUT(UVC_O_GST2, a2); OUT(UVC_O_GST3, a3); OUT(UVC_O_GST4, a0a1); OUT(UVC_O_GST5, a1a0); OUT(UVC_O_GST6, a1a2); OUT(UVC_O_GST7, a2a1);
        OUT(UVC_O_germ_GT, i_best); OUT(UVC_O_germ_GQ, germ_GQ); OUT(UVC_O_germ_emit, emit);
        OUT(UVC_O_germ_ref, (int)(sel[0] >= 0 ? sel[0] : -1)); OUT(UVC_O_germ_alt1, (int)(sel[1] >= 0 ? sel[1] : -1)); OUT(UVC_O_germ_alt2, (int)(sel[2] >= 0 ? sel[2] : -1));
    }
    GR_(vAC, gi) = vAC; GR_(gemit, gi) = emit; GR_(kept, gi) = emit;   // a GERMLINE line keeps the group (k_keep_scan); k_call_rec adds the written records
}

__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 8))) k_call_rec(UvcParams P, ScoreCtx C, Stage S) {
    REC_PROLOGUE
    const bool tprov = (P.tumor_vcf_is_provided != 0);
    const int st = GR_(st, gi);
    // the other symbol type's group of this zerobased_pos, if it is active: BASE (even group index) sits right in front of LINK in the list
    const long long g = C.active[gi];
    long long gj = -1;
    {
        const long long ngroups = 2LL * (C.pos_end - C.pos_beg);
        const long long n_active = PK_FLAGS(C.offsets[ngroups]);
        if (st == UVC_BASE_SYMBOL) { if (gi + 1 < n_active && gi + 1 < S.cap && C.active[gi + 1] == g + 1) gj = gi + 1; }
        else if (gi > 0 && C.active[gi - 1] == g - 1) gj = gi - 1;
    }
    const int vA_own = GR_(vAC, gi), vA_oth = (gj >= 0 ? GR_(vAC, gj) : 0);
    const int vAC0 = (st == UVC_BASE_SYMBOL ? vA_own : vA_oth), vAC1 = (st == UVC_BASE_SYMBOL ? vA_oth : vA_own);
    const bool germ_any = (GR_(gemit, gi) != 0) || (gj >= 0 && GR_(gemit, gj) != 0);
    const int ref_bDP = GR_(refbdp, gi);
    const bool should_output_ref_allele = (C.all_out || germ_any);
    OUT(UVC_O_vAC0, vAC0); OUT(UVC_O_vAC1, vAC1);
    const int tki = FLD(UVC_O_tkey, rec);
    const bool will_generate_out = (!tprov ? ((P.outvar_flag & 0x4) != 0) : (tki >= 0 && (P.outvar_flag & 0x2)));
    const bool is_out_blocked = (((UVC_BASE_NN == symbol) && !(0x20 & P.outvar_flag)) || ((UVC_LINK_NN == symbol) && !(0x40 & P.outvar_flag)));
    int o_out = 0, o_vHGQ = 0, o_NLODQ = 0, o_NLODV = SYM_END, o_TLODQ = 0, o_SQ = 0, o_QUAL = 0, o_FILTER = 0, o_keep = 0, bq4[4] = { 0, 0, 0, 0 }, cq4[4] = { 0, 0, 0, 0 };
    if (will_generate_out && !is_out_blocked) {
        o_out = 1;
        const int germ_phred = (is_subst(symbol) ? P.germ_phred_hetero_snp : P.germ_phred_hetero_indel);
        const int nlodq_singlesite = FLD(UVC_O_vNLODQ, rec);
        const int nlodq_singlesample = nlodq_singlesite - 3 + germ_phred;
        int nlodq1;
        const int totBDP = FLD(UVC_O_bDP, rec);
        const int own_bDP = RH_(bdepth, rec);
        int t_BDP, t_bDP, t_CDP1x, t_cDP1x, t_cVQ1, t_cPCQ1, t_CDP2x, t_cDP2x, t_cVQ2, t_cPCQ2, t_bNMQ, t_tDP = 0;
        if (tprov) {
            const UvcTumorKey &tk = C.tkeys[tki];
            int nlodq_inc = 999;
            const int ptr[2] = { FLD(UVC_O_germ_alt1, rec), FLD(UVC_O_germ_alt2, rec) };
            for (int k = 0; k < 2; k++) {
                const int normsymbol = (ptr[k] >= 0 ? FLD(UVC_O_symbol, ptr[k]) : SYM_END);
                const int bgerr_norm_max_ad = (ptr[k] >= 0 ? FLD(UVC_O_cDP1x, ptr[k]) : 50);
                const double tAD = (tk.cDP1x + 1 * 50) / 100.0, tDP = (tk.CDP1x + 2 * 50) / 100.0;
                const double nAD = (bgerr_norm_max_ad + 1 * 50) / 100.0, nDP = ((ptr[k] >= 0 ? FLD(UVC_O_CDP1x0, ptr[k]) : 0) + 2 * 50) / 100.0;
                const double bjpfrac = ((tAD) / (tDP)) / ((nAD) / (nDP));
                const int binom = (int)binom_llr((tDP - tAD) / (tDP), nDP - nAD, nAD);
                const int powlaw = (int)(P.powlaw_exponent * 10 / log(10.0) * log(bjpfrac));
                const int inc_snp = 2 * P.germ_phred_hetero_snp - P.germ_phred_het3al_snp, inc_indel = 2 * P.germ_phred_hetero_indel - P.germ_phred_het3al_indel;
                const int triallele_inc = ((normsymbol != symbol) ? (is_subst(symbol) ? inc_snp : inc_indel) : 0);
                const int new_inc = (int)dbetween((double)imin(binom, powlaw), -3.0, P.powlaw_anyvar_base) + triallele_inc;
                if (nlodq_inc > new_inc) { nlodq_inc = new_inc; o_NLODV = normsymbol; }
            }
            const int n_norm_alts = (totBDP - ref_bDP) + own_bDP;
            nlodq1 = imax(imax(nlodq_singlesite, germ_phred + nlodq_inc), tk.vHGQ + imin(3, totBDP - n_norm_alts * (int)round(0.5 / P.contam_any_mul_frac)));
            t_BDP = tk.BDP; t_bDP = tk.bDP; t_CDP1x = tk.CDP1x; t_cDP1x = tk.cDP1x; t_cVQ1 = tk.cVQ1; t_cPCQ1 = tk.cPCQ1;
            t_CDP2x = tk.CDP2x; t_cDP2x = tk.cDP2x; t_cVQ2 = tk.cVQ2; t_cPCQ2 = tk.cPCQ2; t_bNMQ = tk.bNMQ; t_tDP = tk.tDP;
        } else {
            nlodq1 = nlodq_singlesample;
            t_BDP = totBDP; t_bDP = own_bDP; t_CDP1x = FLD(UVC_O_CDP1x0, rec); t_cDP1x = FLD(UVC_O_cDP1x, rec); t_cVQ1 = FLD(UVC_O_cVQ1, rec); t_cPCQ1 = FLD(UVC_O_cPCQ1, rec);
            t_CDP2x = FLD(UVC_O_CDP2x0, rec); t_cDP2x = FLD(UVC_O_cDP2x, rec); t_cVQ2 = FLD(UVC_O_cVQ2, rec); t_cPCQ2 = FLD(UVC_O_cPCQ2, rec); t_bNMQ = FLD(UVC_O_bNMQ, rec);
        }
        o_vHGQ = nlodq_singlesample;
        const bool normal = tprov;
        const int nfm_cDP1x = (normal ? FLD(UVC_O_cDP1x, rec) : 0), nfm_CDP1x = (normal ? FLD(UVC_O_CDP1x0, rec) : 0), nfm_cDP2x = (normal ? FLD(UVC_O_cDP2x, rec) : 0), nfm_CDP2x = (normal ? FLD(UVC_O_CDP2x0, rec) : 0);
        const int nfm_cVQ1 = (normal ? FLD(UVC_O_cVQ1, rec) : 0), nfm_cVQ2 = (normal ? FLD(UVC_O_cVQ2, rec) : 0), nfm_BDP = (normal ? totBDP : 0), nfm_CDP1 = (normal ? FLD(UVC_O_DP, rec) : 0);
        const int inc_snp = imax(0, 2 * P.germ_phred_hetero_snp - P.germ_phred_het3al_snp), inc_indel = imax(0, 2 * P.germ_phred_hetero_indel - P.germ_phred_het3al_indel);
        int het3al_inc = (is_subst(symbol) ? inc_snp : inc_indel);
        if (is_ins(symbol) || is_del(symbol)) het3al_inc = (int)nnminus(inc_indel + 1, FLD(UVC_O_gapSa_len, rec));
        const int qmin = P.microadjust_syserr_MQ_NMR_tn_syserr_no_penal_qual_min, qmax = P.microadjust_syserr_MQ_NMR_tn_syserr_no_penal_qual_max;
        const int tn_dec_by_xm = ibetween(imin(FLD(UVC_O_bNMQ, rec), t_bNMQ), qmin, qmax) - qmin;
        double add1 = 0, add2 = 0;
        int tn_dec_both = 0;
        if (normal) {
            const long long LI = T_APLRI0 + T_APLRI2, LIDP = T_APLRI1 + T_APLRI3;
            if (LI < LIDP * (long long)P.lib_wgs_min_avg_fraglen) { add1 = P.lib_nonwgs_normal_add_mul_ad * nfm_cDP1x / 100.0; add2 = P.lib_nonwgs_normal_add_mul_ad * nfm_cDP2x / 100.0; }
            if (t_tDP > 500 && FLD(UVC_O_DP, rec) > 500 && is_del(symbol) && T_APDP2 * 3 > T_APDP0) tn_dec_both = imin((int)nnminus(nfm_cVQ1, 31), 9);
        }
        const int prior_phred = ((UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) ? 11 : 3);
        if (P.tn_syserr_norm_devqual >= 0) normv_quals(bq4, (t_cDP1x + 0.5) / 100.0 + 0.0, (t_CDP1x + 1.0) / 100.0 + 0.0, t_cVQ1, t_cPCQ1, (nfm_cDP1x + 0.5) / 100.0 + 0.0 + add1, (nfm_CDP1x + 1.0) / 100.0 + 0.0 + add1,
                                                   (int)nnminus(nfm_cVQ1, het3al_inc), P.tn_syserr_norm_devqual, prior_phred, tn_dec_by_xm, P.powlaw_exponent);
        else normv_quals2(bq4, (t_cDP1x + 0.5) / 100.0 + 0.0, (t_CDP1x + 1.0) / 100.0 + 0.0, t_cVQ1, t_cPCQ1, (nfm_cDP1x + 0.5) / 100.0 + 0.0 + add1, (nfm_CDP1x + 1.0) / 100.0 + 0.0 + add1, (int)nnminus(nfm_cVQ1, het3al_inc));
        const int converted_nfm_cVQ2 = nfm_cVQ1 - (3 * (nfm_BDP + 1) / (nfm_CDP1 + 1));
        const int norm_norm_vq = (int)nnminus(nfm_cVQ2, imax(het3al_inc, 3) - 3);
        if (P.tn_syserr_norm_devqual >= 0) normv_quals(cq4, (t_cDP2x + 0.5) / 100.0 + 0.0, (t_CDP2x + 1.0) / 100.0 + 0.0, t_cVQ2, t_cPCQ2, (nfm_cDP2x + 0.5) / 100.0 + 0.0 + add2, (nfm_CDP2x + 1.0) / 100.0 + 0.0 + add2,
                                                   norm_norm_vq, P.tn_syserr_norm_devqual, prior_phred, imax(tn_dec_by_xm, imin(imax(nfm_cVQ2, converted_nfm_cVQ2), 12)), P.powlaw_exponent);
        else normv_quals2(cq4, (t_cDP2x + 0.5) / 100.0 + 0.0, (t_CDP2x + 1.0) / 100.0 + 0.0, t_cVQ2, t_cPCQ2, (nfm_cDP2x + 0.0) / 100.0 + 0.5 + add2, (nfm_CDP2x + 0.0) / 100.0 + 1.0 + add2, norm_norm_vq);
        const int tlodq1 = imax(bq4[3], cq4[3]);
        const bool deanim = ((UVC_BASE_C == refsymbol && UVC_BASE_T == symbol) || (UVC_BASE_G == refsymbol && UVC_BASE_A == symbol));
        const double b_min_tlodq = 2 + 3 - (-10 * log((t_bDP + 1e-3) / (t_BDP + 1)) / log(10.0)) / 10.0;
        const double c2v_min_tlodq = 2 + 5 - (-10 * log((t_cDP2x * 0.01 + 1e-5) / (t_CDP2x * 0.01 + 1) / (deanim ? 5 : 1)) / log(10.0)) / 10.0;
        const float lowestVAQ = (float)dmax(b_min_tlodq, c2v_min_tlodq);
        const int tlodq = ((tlodq1 >= 10) ? tlodq1 : (tlodq1 * 3 - 20)) - tn_dec_both;
        const int nlodq = nlodq1 - tn_dec_both;
        const int somaticq = imin(tlodq, nlodq);
        float v = (normal ? ((float)somaticq) : fmaxf((float)tlodq, lowestVAQ));
        { const float base = (float)pow(10.0, 0.1); if (v < 10.0f) v = log1pf(powf(base, v)) / logf(base); }   // calc_non_negative<float>
        o_TLODQ = tlodq; o_NLODQ = nlodq; o_SQ = somaticq; o_QUAL = __float_as_int(v);
        o_FILTER = (v < 10 ? 0 : v < 20 ? 1 : v < 30 ? 2 : v < 40 ? 3 : v < 50 ? 4 : v < 60 ? 5 : 6);
        const int vad1 = f_aBQ2; const long long vdp1 = T_ABQ20;
        const bool keep_var = ((((double)v >= P.vqual) || ((!tprov) && ((vad1 >= P.vad1 && vdp1 >= P.vdp1 && (vdp1 * P.vfa1) <= vad1) || (t_bDP >= P.vad2 && t_BDP >= P.vdp2 && (t_BDP * P.vfa2) <= t_bDP))))
                               && (symbol != refsymbol || should_output_ref_allele));
        o_keep = (keep_var && t_bDP >= ((symbol == refsymbol) ? P.min_r_ad : P.min_a_ad)) ? 1 : 0;
    }
    OUT(UVC_O_out, o_out); OUT(UVC_O_vHGQ, o_vHGQ); OUT(UVC_O_NLODQ, o_NLODQ); OUT(UVC_O_NLODV, o_NLODV); OUT(UVC_O_TLODQ, o_TLODQ); OUT(UVC_O_SomaticQ, o_SQ);
    for (int i = 0; i < 4; i++) { OUT(UVC_O_TNBQF0 + i, bq4[i]); OUT(UVC_O_TNCQF0 + i, cq4[i]); }
    OUT(UVC_O_QUAL, o_QUAL); OUT(UVC_O_FILTER, o_FILTER); OUT(UVC_O_keep, o_keep);
    if (o_keep && o_out) GR_(kept, gi) = 1;   // (every writer stores the same 1; nobody reads it before k_keep_scan)
}


// ---- UvcScoreRequest::kept_only: the groups that are written travel, nothing else ----
// A (zerobased_pos, symbol type) group is kept iff one of its records is written (keep && out) or it has a GERMLINE line (germ_emit): the
// record writer reads the REF record and the genotype's records of such a group and nothing of the others.  Kept groups keep their order;
// germ_ref / germ_alt1 / germ_alt2 (record indices inside the group) move with it.  One chained scan over the ACTIVE groups (a few ten
// thousand at the default gate), blocks take tiles of the list by ticket.
__global__ void __launch_bounds__(GS_BLOCK) k_keep_scan(ScoreCtx C, Stage S, long long ngroups, long long *total_kept, int *err) {
    __shared__ unsigned long long sh_wave[4], sh_prefix;
    __shared__ int sh_tile;
    const long long n_active = PK_FLAGS(C.offsets[ngroups]);
    const int ntiles = (int)((n_active + GS_TILE - 1) / GS_TILE);
    for (;;) {
        if (threadIdx.x == 0) sh_tile = (int)atomicAdd(&S.cnt[CNT_ticket2], 1u);
        __syncthreads();
        const int tile = sh_tile;
        if (tile >= ntiles) break;
        const long long a0 = (long long)tile * GS_TILE + (long long)threadIdx.x * GS_ITEMS;
        long long v[GS_ITEMS]; unsigned long long s = 0;
#pragma unroll
        for (int i = 0; i < GS_ITEMS; i++) {
            const long long ai = a0 + i;
            long long n = 0;
            if (ai < n_active && ai < S.cap && GR_(kept, ai)) n = GR_(nrec, ai);   // (a group that did not fit has nrec 0)
            v[i] = n; s += (unsigned long long)n;
        }
        unsigned long long total = 0;
        const unsigned long long excl = block_excl_scan256(s, sh_wave, total);
        if (threadIdx.x < 64) { const unsigned long long p = cs_lookback(S.status2, tile, total, (int)threadIdx.x, err); if (threadIdx.x == 0) sh_prefix = p; }
        __syncthreads();
        long long run = (long long)(sh_prefix + excl);
#pragma unroll
        for (int i = 0; i < GS_ITEMS; i++) { const long long ai = a0 + i; if (ai < n_active && ai < S.cap) S.keptoff[ai] = (v[i] ? (int32_t)run : -1); run += v[i]; }
        if (tile == ntiles - 1 && threadIdx.x == 0) *total_kept = (long long)(sh_prefix + total);
        __syncthreads();   // sh_tile / sh_prefix / sh_wave are reused by the next tile
    }
}
#define KEEP_LANES 8
__global__ void __launch_bounds__(256) k_keep_copy(ScoreCtx C, Stage S, long long ngroups, int32_t *fields2) {
    const long long n_active = PK_FLAGS(C.offsets[ngroups]);
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long ai = t / KEEP_LANES; const int sub = (int)(t % KEEP_LANES);
    if (ai >= n_active || ai >= S.cap) return;
    const long long q0 = S.keptoff[ai];
    if (q0 < 0) return;
    const long long r0 = GR_(rec0, ai), n = GR_(nrec, ai);
    const int32_t *fields = C.fields; const long long capacity = C.capacity;
    for (int f = sub; f < UVC_NUM_SCORE_FIELDS; f += KEEP_LANES) {
        const bool is_index = (f == UVC_O_germ_ref || f == UVC_O_germ_alt1 || f == UVC_O_germ_alt2);
        for (long long k = 0; k < n; k++) {
            int32_t v = fields[(size_t)f * capacity + r0 + k];
            if (is_index && v >= 0) v = (int32_t)(v - r0 + q0);
            fields2[(size_t)f * capacity + q0 + k] = v;
        }
    }
}

// scratch layout of one score call.  The head [record counts (2 x int64)] [counters] [tile states of the two scans] is zeroed in front of every call.
static size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
struct ScratchLayout { size_t zero_bytes, cnt, status1, status2, offsets, active, grp, tot, rh, al, al64, mid, d4, keptoff, total; };
static ScratchLayout scratch_layout(int64_t npos_scored, int64_t cap) {
    const size_t ngroups = (size_t)(2 * npos_scored), c = (size_t)(cap > 0 ? cap : 1);
    const size_t ntiles1 = (ngroups + GATE_TILE - 1) / GATE_TILE + 1, ntiles2 = (std::min(ngroups, c) + GS_TILE - 1) / GS_TILE + 1;
    ScratchLayout L; size_t o = 16;
    L.cnt = o; o += NCNT * 4;
    L.status1 = o; o += ntiles1 * 8;
    L.status2 = o; o += ntiles2 * 8;
    o = align16(o); L.zero_bytes = o;
    L.offsets = o; o = align16(o + (ngroups + 1) * 8);
    L.active = o; o = align16(o + ngroups * 4);
    L.grp = o; o = align16(o + (size_t)NGR * c * 4);
    L.tot = o; o = align16(o + (size_t)NTOT * c * 8);
    L.rh = o; o = align16(o + (size_t)NRH * c * 4);
    L.al = o; o = align16(o + (size_t)NAL32 * c * 4);
    L.al64 = o; o = align16(o + (size_t)NAL64 * c * 8);
    L.mid = o; o = align16(o + (size_t)NMID * c * 8);
    L.d4 = o; o = align16(o + (size_t)NDP4 * 2 * c * 8);
    L.keptoff = o; o = align16(o + c * 4);
    L.total = o;
    return L;
}
extern "C" size_t uvc_score_scratch_bytes(int64_t npos_scored, int64_t capacity) { return scratch_layout(npos_scored, capacity).total; }
extern "C" size_t uvc_score_scratch_zero_bytes(int64_t npos_scored, int64_t capacity) { return scratch_layout(npos_scored, capacity).zero_bytes; }

// The caller zeroes the first uvc_score_scratch_zero_bytes of `scratch` on the stream in front of this; the record counts (all records, kept
// records) are its first two int64.
extern "C" int uvc_launch_score(const RegionDev *R, const UvcParams *P, const UvcScoreRequest *req, const UvcIndelAllele *d_alleles, const int32_t *d_allele_rows, int64_t n_alleles,
                                const UvcGapRow *d_gap_rows, const uint8_t *d_gap_seq, const UvcTumorKey *d_tkeys, int32_t *d_fields, int64_t capacity,
                                char *scratch /* uvc_score_scratch_bytes */, int32_t *d_fields_kept /* kept_only: a second [fields][capacity] array */, hipStream_t s) {
    ScoreCtx C;
    C.pos_beg = req->pos_beg; C.pos_end = req->pos_end; C.all_out = (req->all_out || P->should_output_all) ? 1 : 0; C.is_amplicon = req->is_amplicon; C.base_at_beg = req->base_at_pos_beg ? 1 : 0;
    C.alleles = d_alleles; C.allele_rows = d_allele_rows; C.n_alleles = n_alleles; C.gap_rows = d_gap_rows; C.gap_seq = d_gap_seq; C.tkeys = d_tkeys; C.n_tkeys = (d_tkeys ? req->n_tumor_keys : 0); C.fields = d_fields; C.capacity = capacity;
    const long long npos_scored = C.pos_end - C.pos_beg, ngroups = 2LL * npos_scored;
    if (ngroups <= 0) return 0;
    const ScratchLayout L = scratch_layout(npos_scored, capacity);
    long long *d_count = (long long *)scratch;
    Stage S;
    S.cnt = (unsigned int *)(scratch + L.cnt); S.status1 = (unsigned long long *)(scratch + L.status1); S.status2 = (unsigned long long *)(scratch + L.status2);
    C.offsets = (long long *)(scratch + L.offsets); C.active = (int *)(scratch + L.active);
    S.grp = (int32_t *)(scratch + L.grp); S.tot = (long long *)(scratch + L.tot); S.rh = (int32_t *)(scratch + L.rh); S.al = (int32_t *)(scratch + L.al); S.al64 = (long long *)(scratch + L.al64);
    S.mid = (double *)(scratch + L.mid); S.d4 = (double *)(scratch + L.d4); S.keptoff = (int32_t *)(scratch + L.keptoff); S.cap = capacity;
    const unsigned ntiles = (unsigned)((ngroups + GATE_TILE - 1) / GATE_TILE);
    hipLaunchKernelGGL(k_gate_scan, dim3(ntiles), dim3(GS_BLOCK), 0, s, *R, *P, C, S, d_count);
    // every kernel below works on a list whose length lives on the device: grids cover what the rows can hold, the threads beyond the length leave
    const long long max_groups = (ngroups < capacity ? ngroups : capacity), max_recs = capacity;
    hipLaunchKernelGGL(k_enum, dim3((unsigned)((max_groups + 127) / 128)), dim3(128), 0, s, *R, *P, C, S);
    hipLaunchKernelGGL(k_gather, dim3((unsigned)((max_groups + 255) / 256), (NGATHER + GATHER_PLANES - 1) / GATHER_PLANES), dim3(256), 0, s, *R, C, S);
    hipLaunchKernelGGL(k_dpv_pre, dim3((unsigned)((max_recs + 127) / 128)), dim3(128), 0, s, *P, C, S);
    hipLaunchKernelGGL(k_dp4, dim3((unsigned)((max_recs + 63) / 64), NDP4 / DP4_WAVES), dim3(64 * DP4_WAVES), 0, s, *P, C, S);
    hipLaunchKernelGGL(k_dpv_post, dim3((unsigned)((max_recs + 127) / 128)), dim3(128), 0, s, *P, C, S);
    hipLaunchKernelGGL(k_qual, dim3((unsigned)((max_recs + 127) / 128)), dim3(128), 0, s, *P, C, S);
    hipLaunchKernelGGL(k_call_group, dim3((unsigned)((max_groups + 127) / 128)), dim3(128), 0, s, *P, C, S);
    hipLaunchKernelGGL(k_call_rec, dim3((unsigned)((max_recs + 127) / 128)), dim3(128), 0, s, *P, C, S);
    if (req->kept_only && d_fields_kept) {
        const long long tiles2 = (max_groups + GS_TILE - 1) / GS_TILE;
        hipLaunchKernelGGL(k_keep_scan, dim3((unsigned)(tiles2 < 1024 ? (tiles2 > 0 ? tiles2 : 1) : 1024)), dim3(GS_BLOCK), 0, s, C, S, ngroups, d_count + 1, R->err);
        hipLaunchKernelGGL(k_keep_copy, dim3((unsigned)((max_groups * KEEP_LANES + 255) / 256)), dim3(256), 0, s, C, S, ngroups, d_fields_kept);
    }
    return 0;
}

// ---- uvcgpu_region_check_presence: the planes against the presence statement of k_enum, every (position, symbol) ----
__global__ void __launch_bounds__(256) k_check_presence(RegionDev R, unsigned long long *n_bad) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = (int)blockIdx.y;
    if (x >= R.npos) return;
    long long any = 0;
    for (int a = 0; a < NAL; a++) { const GatherDesc d = c_gather.d[24 + a]; any |= stage_cell(R, d.grp, d.plane, s, x); }
    if (!any) return;
    const bool dense = (s == UVC_LINK_M || (s <= UVC_BASE_NN && s == (int)R.refsym[x]));
    if (!(dense || ((R.occ[x] >> s) & 1u))) atomicAdd(n_bad, 1ull);   // type_mask
    // cIAQ / cIAD / cIDQ of a strand only behind a P5 bucket of that (strand, position) (GM_P5F / GM_P5R of k_enum)
    if ((VQP(R, UVC_VQ_cIAQf, s, x) | VQP(R, UVC_VQ_cIADf, s, x) | VQP(R, UVC_VQ_cIDQf, s, x)) != 0 && !R.p5flag[x]) atomicAdd(n_bad, 1ull);
    if ((VQP(R, UVC_VQ_cIAQr, s, x) | VQP(R, UVC_VQ_cIADr, s, x) | VQP(R, UVC_VQ_cIDQr, s, x)) != 0 && !R.p5flag[(size_t)R.npos + x]) atomicAdd(n_bad, 1ull);
}
extern "C" void uvc_launch_check_presence(const RegionDev *R, unsigned long long *d_n_bad, hipStream_t s) {
    hipLaunchKernelGGL(k_check_presence, dim3((unsigned)((R->npos + 255) / 256), NSYM), dim3(256), 0, s, *R, d_n_bad);
}

// ---- position-level numbers of the VCF writer: the MGVCF block lines (main.cpp:655-735) and ADDITIONAL_INDEL_CANDIDATE (main.cpp:759-799) ----
// Per position 10 ints: for LINK then BASE (SYMBOL_TYPES_IN_VCF_ORDER) the total fragment depth, the de-duplicated depth, the BQ-filtered
// de-duplicated depth and the homozygous-reference quality; then segprep_a_dp and segprep_a_near_long_clip_dp.
__global__ void __launch_bounds__(256) k_block_stats(RegionDev R, UvcParams P, long long x0, long long n, int *out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t x = x0 + i;
    int *o = out + i * 10;
    if (x < 0 || x >= R.npos) { for (int q = 0; q < 10; q++) o[q] = 0; return; }
    for (int t = 0; t < 2; t++) {
        const int st = (t == 0 ? UVC_LINK_SYMBOL : UVC_BASE_SYMBOL);
        const int nsym = st_count(st);
        const int refsymbol = (st == UVC_BASE_SYMBOL ? (x < R.npos - 1 ? (int)R.refsym[x] : UVC_BASE_N) : UVC_LINK_M);
        int bdepth = 0, cdepth = 0, cdep12 = 0;
        for (int sd = 0; sd < 2; sd++) for (int k = 0; k < nsym; k++) { const int s = st_symbol(st, k); bdepth += FRP(R, sd, UVC_FRAG_bDP, s, x); cdepth += FAP(R, sd, UVC_FAM_cDP1, s, x); cdep12 += FAP(R, sd, UVC_FAM_cDP12, s, x); }
        const int ref_c = FAP(R, 0, UVC_FAM_cDP12, refsymbol, x) + FAP(R, 1, UVC_FAM_cDP12, refsymbol, x);
        const int nonref_c = cdep12 - ref_c;
        const double k10 = 10.0 / log(10.0);
        const double rb = -binom_llr(P.contam_any_mul_frac, nonref_c + 0.5, cdepth + 1.0);
        const double rp = -dmax(0.0, P.powlaw_exponent * k10 * logit2((nonref_c + 0.5) / (cdepth + 1.0), P.contam_any_mul_frac));
        const double nb = -binom_llr(P.germ_hetero_FA, ref_c + 0.5, cdepth + 1.0);
        const double np = -dmax(0.0, P.powlaw_exponent * k10 * logit2((ref_c + 0.5) / (cdepth + 1.0), P.germ_hetero_FA));
        o[t * 4 + 0] = bdepth; o[t * 4 + 1] = cdepth; o[t * 4 + 2] = cdep12;
        o[t * 4 + 3] = P.germ_phred_hetero_snp + (int)round(dmax(rb, rp) - (double)(int)round(dmax(nb, np)));
    }
    o[8] = P32(R, UVC_P_a_dp, x); o[9] = P32(R, UVC_P_a_near_long_clip_dp, x);
}
extern "C" void uvc_launch_block_stats(const RegionDev *R, const UvcParams *P, int64_t x0, int64_t n, int32_t *d_out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_block_stats, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, *R, *P, (long long)x0, (long long)n, d_out);
}
