// uvc_kernels_score.hip -- per-position Bayesian / power-law scoring on gfx950 (fp64 VALU, no MFMA).
//
// Replaces the BcfFormat_symbol* call group of process_batch (main.cpp:608-1000):
//   BcfFormat_symboltype_init  main.hpp:3889   -> group_totals()
//   BcfFormat_symbol_init      main.hpp:4094   -> allele_load() (+ fill_symbol_VQ_fmts, main.hpp:3820)
//   BcfFormat_symbol_calc_DPv  main.hpp:4274   -> calc_dpv()
//   BcfFormat_symbol_sum_DPv   main.hpp:4888   -> in-thread reduction (one thread owns a (position, symbol type) group)
//   BcfFormat_symbol_calc_qual main.hpp:4908   -> calc_qual()
// Tumor-only, and with UvcTumorKey records in the request the normal sample of a T/N pair (SURVEY next-row N2).
//
// Launch shape: k_score_count  one thread per (zerobased_pos, symbol type): number of emitted alleles (candidate gate, main.cpp:832-837)
//               k_scan_*       exclusive prefix sum -> deterministic record slots, in the reference's emission order
//               k_score        one thread per (zerobased_pos, symbol type): pass 1 = init + calc_DPv per allele and the
//                              cross-allele sums, pass 2 = calc_qual per allele; records are written SoA [field][record].
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

struct Tot {   // symbol-type totals: [0] = sum over the type's symbols, [1] = the NN symbol (fill_symboltype_fmt, main.hpp:3745-3793)
    long long APDP[12], APXM[8], APLRI[4];
    long long A1BQf0, A1BQr0, AMQs0, AP10, AP20, ADPff0, ADPfr0, ADPrf0, ADPrr0, ALP10, ALP20, ALPL0, ARP20, ARPL0, ALB20, ALBL0, ARB20, ARBL0, ABQ20, APF20, ALI20, ARIf0, ARI20, ALIr0;
    int BDPb[2], BTAb[2], BTBb[2], CDP1b[2], CDP12b[2], CDP2b[2], CDP3b[2];
    long long C2LP20, C2LPL0, C2RP20, C2RPL0, C2LB20, C2LBL0, C2RB20, C2RBL0, C2BQ20, C2LP00, C2RP00;
    int DDP10;
};

struct Al {   // one allele (index a = 0 everywhere in the reference)
    int symbol;
    int a1BQf, a1BQr, aMQs, aP1, aP2, aDPff, aDPfr, aDPrf, aDPrr, aLP1, aLP2, aRP1, aRP2, aLB1, aLB2, aRB1, aRB2;
    long long aLPL, aRPL, aLBL, aRBL, aLIT, aRIT;
    int a2XM2, a2BM2, aBQ2, aPF1, aPF2, aLI1, aLI2, aLIr, aRI1, aRI2, aRIf, aP3, aNC;
    int bDPf, bTAf, bTBf, bDPr, bTAr, bTBr;
    int cDP1f, cDP12f, cDP2f, cDP3f, cDPMf, cDPmf, cDP1r, cDP12r, cDP2r, cDP3r, cDPMr, cDPmr;
    int c2LP1, c2LP2, c2RP1, c2RP2, c2LP0, c2RP0, c2LB1, c2LB2, c2RB1, c2RB2, c2BQ2;
    long long c2LPL, c2RPL, c2LBL, c2RBL;
    int dDP1, dDP2;
    int AD, bAD;
    int bMQ, a2BQf, a2BQr, aBQ, aBQQ, bIAQb, bIADb, cIAQf, cIADf, cIDQf, cIAQr, cIADr, cIDQr;
    int bDPa, cDP0a, gap_len;
    int tier2, bNMQ;
    int cDP1v, cDP1w, cDP1x, cDP2v, cDP2w, cDP2x;
};

// sum of a plane value over the symbols of the type (integer: any order).  Unrolled over the eight possible symbols; the guard is a SELECT on the
// loaded value, not a branch around the load (a BASE group loads its last symbol twice more and drops it): with `if (k_ < nsym)` around each
// load the compiler emitted 332 s_cbranch_execz and a s_waitcnt vmcnt(0) behind every load.
#define SUMSYM(expr) ({ long long r_ = 0; const int sb_ = (st == UVC_BASE_SYMBOL ? UVC_BASE_A : UVC_LINK_M); _Pragma("unroll") for (int k_ = 0; k_ < 8; k_++) { const int s = sb_ + (k_ < nsym ? k_ : nsym - 1); const long long v_ = (long long)(expr); r_ += (k_ < nsym ? v_ : 0LL); } r_; })

// The plane pointers group_totals reads, by value (members named like RegionDev's so the plane macros work on it).
struct TotSrc { const int32_t *prep32, *seg32, *vq, *frag, *fam, *faminfo32, *duplex; const int64_t *prep64, *seg64, *faminfo64; int64_t npos; };

// Out of line on purpose: inlined into k_score (512 registers, spilling) the scheduler ran in its register-saving mode and put a
// s_waitcnt vmcnt(0) behind every pair of these ~440 loads (288 waits).  As its own function -- pointers by value, totals built in
// registers and stored once, so no store can alias a later load -- the loads go out in batches (78 waits, most of them partial).
// Measured (1 Mb x 300x tile, 54 k records, kernel alone): 548 -> 470 us.  The kernel stays latency-bound at one wave per SIMD: PMC
// SQ_WAIT_INST_ANY 34 %, SQ_ACTIVE_INST_ANY 15 % of SQ_WAVE_CYCLES, ~51 k VALU instructions per wave (fp64 divisions and log / exp).
__device__ __attribute__((noinline)) void group_totals(const TotSrc R, int64_t x, int st, Tot *out) {
    Tot f;
    const int nsym = st_count(st);
    const int pidx[12] = { UVC_P_a_dp, UVC_P_a_near_ins_dp, UVC_P_a_near_del_dp, UVC_P_a_near_RTR_ins_dp, UVC_P_a_near_RTR_del_dp, UVC_P_a_pcr_dp,
                           UVC_P_a_snv_dp, UVC_P_a_dnv_dp, UVC_P_a_highBQ_dp, UVC_P_a_near_pcr_clip_dp, UVC_P_a_near_long_clip_dp, UVC_P_a_umi_dp };
    _Pragma("unroll") for (int i = 0; i < 12; i++) f.APDP[i] = P32(R, pidx[i], x);
    f.APXM[0] = P32(R, UVC_P_a_XM1500, x); f.APXM[1] = P32(R, UVC_P_a_GO1500, x); f.APXM[2] = P32(R, UVC_P_a_qlen, x); f.APXM[3] = P32(R, UVC_P_a_GAPLEN, x);
    f.APXM[4] = P64(R, UVC_P_a_near_ins_pow2len, x); f.APXM[5] = P64(R, UVC_P_a_near_del_pow2len, x);
    f.APXM[6] = P32(R, UVC_P_a_near_ins_inv100len, x); f.APXM[7] = P32(R, UVC_P_a_near_del_inv100len, x);
    f.APLRI[0] = P64(R, UVC_P_a_LI, x); f.APLRI[1] = P32(R, UVC_P_a_LIDP, x); f.APLRI[2] = P64(R, UVC_P_a_RI, x); f.APLRI[3] = P32(R, UVC_P_a_RIDP, x);
    // int32 FORMAT fields truncate the int64 sum on assignment; the *L fields are int64 (bcf_formats_generator1.cpp:220-245)
    f.A1BQf0 = (int)SUMSYM(VQP(R, UVC_VQ_a1BQf, s, x)); f.A1BQr0 = (int)SUMSYM(VQP(R, UVC_VQ_a1BQr, s, x));
    f.AMQs0 = (int)SUMSYM(S32(R, UVC_S_aMQs, s, x)); f.AP10 = (int)SUMSYM(S32(R, UVC_S_aP1, s, x)); f.AP20 = (int)SUMSYM(S32(R, UVC_S_aP2, s, x));
    f.ADPff0 = (int)SUMSYM(S32(R, UVC_S_aDPff, s, x)); f.ADPfr0 = (int)SUMSYM(S32(R, UVC_S_aDPfr, s, x)); f.ADPrf0 = (int)SUMSYM(S32(R, UVC_S_aDPrf, s, x)); f.ADPrr0 = (int)SUMSYM(S32(R, UVC_S_aDPrr, s, x));
    f.ALP10 = (int)SUMSYM(S32(R, UVC_S_aLP1, s, x)); f.ALP20 = (int)SUMSYM(S32(R, UVC_S_aLP2, s, x)); f.ALPL0 = SUMSYM(S32(R, UVC_S_aLPL, s, x));
    f.ARP20 = (int)SUMSYM(S32(R, UVC_S_aRP2, s, x)); f.ARPL0 = SUMSYM(S32(R, UVC_S_aRPL, s, x));
    f.ALB20 = (int)SUMSYM(S32(R, UVC_S_aLB2, s, x)); f.ALBL0 = SUMSYM(S64(R, UVC_S64_aLBL, s, x));
    f.ARB20 = (int)SUMSYM(S32(R, UVC_S_aRB2, s, x)); f.ARBL0 = SUMSYM(S64(R, UVC_S64_aRBL, s, x));
    f.ABQ20 = (int)SUMSYM(S32(R, UVC_S_aBQ2, s, x)); f.APF20 = (int)SUMSYM(S32(R, UVC_S_aPF2, s, x));
    f.ALI20 = (int)SUMSYM(S32(R, UVC_S_aLI2, s, x)); f.ARIf0 = (int)SUMSYM(S32(R, UVC_S_aRIf, s, x)); f.ARI20 = (int)SUMSYM(S32(R, UVC_S_aRI2, s, x)); f.ALIr0 = (int)SUMSYM(S32(R, UVC_S_aLIr, s, x));
    _Pragma("unroll") for (int sd = 0; sd < 2; sd++) {
        f.BDPb[sd] = (int)SUMSYM(FRP(R, sd, UVC_FRAG_bDP, s, x)); f.BTAb[sd] = (int)SUMSYM(FRP(R, sd, UVC_FRAG_bTA, s, x)); f.BTBb[sd] = (int)SUMSYM(FRP(R, sd, UVC_FRAG_bTB, s, x));
        f.CDP1b[sd] = (int)SUMSYM(FAP(R, sd, UVC_FAM_cDP1, s, x)); f.CDP12b[sd] = (int)SUMSYM(FAP(R, sd, UVC_FAM_cDP12, s, x));
        f.CDP2b[sd] = (int)SUMSYM(FAP(R, sd, UVC_FAM_cDP2, s, x)); f.CDP3b[sd] = (int)SUMSYM(FAP(R, sd, UVC_FAM_cDP3, s, x));
    }
    f.C2LP20 = (int)SUMSYM(FIP(R, UVC_FI_c2LP2, s, x)); f.C2LPL0 = SUMSYM(FIP(R, UVC_FI_c2LPL, s, x)); f.C2RP20 = (int)SUMSYM(FIP(R, UVC_FI_c2RP2, s, x)); f.C2RPL0 = SUMSYM(FIP(R, UVC_FI_c2RPL, s, x));
    f.C2LB20 = (int)SUMSYM(FIP(R, UVC_FI_c2LB2, s, x)); f.C2LBL0 = SUMSYM(FI64P(R, UVC_FI64_c2LBL, s, x)); f.C2RB20 = (int)SUMSYM(FIP(R, UVC_FI_c2RB2, s, x)); f.C2RBL0 = SUMSYM(FI64P(R, UVC_FI64_c2RBL, s, x));
    f.C2BQ20 = (int)SUMSYM(FIP(R, UVC_FI_c2BQ2, s, x)); f.C2LP00 = (int)SUMSYM(FIP(R, UVC_FI_c2LP0, s, x)); f.C2RP00 = (int)SUMSYM(FIP(R, UVC_FI_c2RP0, s, x));
    f.DDP10 = (int)SUMSYM(DUP(R, UVC_DUPLEX_dDP1, s, x));
    *out = f;
}

// BcfFormat_symbol_init + fill_symbol_VQ_fmts, main.hpp:4094-4251, 3820-3887
DEV void allele_load(const RegionDev &R, const UvcParams &P, int64_t x, int sym, const Tot &T, int bDPa, int cDP0a, int gap_len, int minABQ, Al &f) {
    f.symbol = sym;
    f.a1BQf = VQP(R, UVC_VQ_a1BQf, sym, x); f.a1BQr = VQP(R, UVC_VQ_a1BQr, sym, x);
    f.aMQs = S32(R, UVC_S_aMQs, sym, x); f.aP1 = S32(R, UVC_S_aP1, sym, x); f.aP2 = S32(R, UVC_S_aP2, sym, x);
    f.aDPff = S32(R, UVC_S_aDPff, sym, x); f.aDPfr = S32(R, UVC_S_aDPfr, sym, x); f.aDPrf = S32(R, UVC_S_aDPrf, sym, x); f.aDPrr = S32(R, UVC_S_aDPrr, sym, x);
    f.aLP1 = S32(R, UVC_S_aLP1, sym, x); f.aLP2 = S32(R, UVC_S_aLP2, sym, x); f.aLPL = S32(R, UVC_S_aLPL, sym, x);
    f.aRP1 = S32(R, UVC_S_aRP1, sym, x); f.aRP2 = S32(R, UVC_S_aRP2, sym, x); f.aRPL = S32(R, UVC_S_aRPL, sym, x);
    f.aLB1 = S32(R, UVC_S_aLB1, sym, x); f.aLB2 = S32(R, UVC_S_aLB2, sym, x); f.aLBL = S64(R, UVC_S64_aLBL, sym, x);
    f.aRB1 = S32(R, UVC_S_aRB1, sym, x); f.aRB2 = S32(R, UVC_S_aRB2, sym, x); f.aRBL = S64(R, UVC_S64_aRBL, sym, x);
    f.a2XM2 = S32(R, UVC_S_a2XM2, sym, x); f.a2BM2 = S32(R, UVC_S_a2BM2, sym, x); f.aBQ2 = S32(R, UVC_S_aBQ2, sym, x);
    f.aPF1 = S32(R, UVC_S_aPF1, sym, x); f.aPF2 = S32(R, UVC_S_aPF2, sym, x);
    f.aLI1 = S32(R, UVC_S_aLI1, sym, x); f.aLI2 = S32(R, UVC_S_aLI2, sym, x); f.aLIr = S32(R, UVC_S_aLIr, sym, x);
    f.aRI1 = S32(R, UVC_S_aRI1, sym, x); f.aRI2 = S32(R, UVC_S_aRI2, sym, x); f.aRIf = S32(R, UVC_S_aRIf, sym, x);
    f.bDPf = FRP(R, 0, UVC_FRAG_bDP, sym, x); f.bTAf = FRP(R, 0, UVC_FRAG_bTA, sym, x); f.bTBf = FRP(R, 0, UVC_FRAG_bTB, sym, x);
    f.bDPr = FRP(R, 1, UVC_FRAG_bDP, sym, x); f.bTAr = FRP(R, 1, UVC_FRAG_bTA, sym, x); f.bTBr = FRP(R, 1, UVC_FRAG_bTB, sym, x);
    f.cDP1f = FAP(R, 0, UVC_FAM_cDP1, sym, x); f.cDP12f = FAP(R, 0, UVC_FAM_cDP12, sym, x); f.cDP2f = FAP(R, 0, UVC_FAM_cDP2, sym, x); f.cDP3f = FAP(R, 0, UVC_FAM_cDP3, sym, x);
    f.cDPMf = FAP(R, 0, UVC_FAM_cDPM, sym, x); f.cDPmf = FAP(R, 0, UVC_FAM_cDPm, sym, x);
    f.cDP1r = FAP(R, 1, UVC_FAM_cDP1, sym, x); f.cDP12r = FAP(R, 1, UVC_FAM_cDP12, sym, x); f.cDP2r = FAP(R, 1, UVC_FAM_cDP2, sym, x); f.cDP3r = FAP(R, 1, UVC_FAM_cDP3, sym, x);
    f.cDPMr = FAP(R, 1, UVC_FAM_cDPM, sym, x); f.cDPmr = FAP(R, 1, UVC_FAM_cDPm, sym, x);
    f.c2LP1 = FIP(R, UVC_FI_c2LP1, sym, x); f.c2LP2 = FIP(R, UVC_FI_c2LP2, sym, x); f.c2LPL = FIP(R, UVC_FI_c2LPL, sym, x);
    f.c2RP1 = FIP(R, UVC_FI_c2RP1, sym, x); f.c2RP2 = FIP(R, UVC_FI_c2RP2, sym, x); f.c2RPL = FIP(R, UVC_FI_c2RPL, sym, x);
    f.c2LB1 = FIP(R, UVC_FI_c2LB1, sym, x); f.c2LB2 = FIP(R, UVC_FI_c2LB2, sym, x); f.c2LBL = FI64P(R, UVC_FI64_c2LBL, sym, x);
    f.c2RB1 = FIP(R, UVC_FI_c2RB1, sym, x); f.c2RB2 = FIP(R, UVC_FI_c2RB2, sym, x); f.c2RBL = FI64P(R, UVC_FI64_c2RBL, sym, x);
    f.c2BQ2 = FIP(R, UVC_FI_c2BQ2, sym, x); f.c2LP0 = FIP(R, UVC_FI_c2LP0, sym, x); f.c2RP0 = FIP(R, UVC_FI_c2RP0, sym, x);
    f.dDP1 = DUP(R, UVC_DUPLEX_dDP1, sym, x); f.dDP2 = DUP(R, UVC_DUPLEX_dDP2, sym, x);
    f.aLIT = S64(R, UVC_S64_aLIT, sym, x); f.aRIT = S64(R, UVC_S64_aRIT, sym, x); f.aP3 = S32(R, UVC_S_aP3, sym, x); f.aNC = S32(R, UVC_S_aNC, sym, x);
    f.AD = f.cDP1f + f.cDP1r; f.bAD = f.bDPf + f.bDPr;
    const int a2BQf = VQP(R, UVC_VQ_a2BQf, sym, x), a2BQr = VQP(R, UVC_VQ_a2BQr, sym, x);
    const int aDPf = f.aDPff + f.aDPrf, aDPr = f.aDPfr + f.aDPrr;
    const int ADP = (int)(T.ADPff0 + T.ADPrf0 + T.ADPfr0 + T.ADPrr0);
    const int rssf = (int)(aDPf * sqrt((double)(((long long)a2BQf * SQR_QUAL_DIV) / imax(1, aDPf))));
    const int rssr = (int)(aDPr * sqrt((double)(((long long)a2BQr * SQR_QUAL_DIV) / imax(1, aDPr))));
    const int rssb = (int)((aDPf + aDPr) * sqrt((double)((a2BQf + a2BQr) * SQR_QUAL_DIV / imax(1, aDPf + aDPr))));
    const double t = dmax(0.0, ((aDPf + aDPr + 0.5) * 2.0 / (ADP + 1.0) - 1.0));
    int minABQa = minABQ - (int)(5 * 10.0 * (t * t));
    const double sbratio = (double)(imax(aDPf, aDPr) * 10 + 10) / (double)(imin(aDPf, aDPr) * 10 + 10);
    minABQa += ibetween((int)(sbratio * sbratio) - P.syserr_BQ_sbratio_q_add, 0, P.syserr_BQ_sbratio_q_max);
    const int xmratio = (P.syserr_BQ_xmratio_q_max * 10 * (aDPf + aDPr) / imax(1, f.a2XM2));
    const int bmratio = (P.syserr_BQ_bmratio_q_max * 10 * (aDPf + aDPr) / imax(1, f.a2BM2));
    minABQa += ibetween(xmratio - P.syserr_BQ_xmratio_q_add, 0, P.syserr_BQ_xmratio_q_max) + ibetween(bmratio - P.syserr_BQ_bmratio_q_add, 0, P.syserr_BQ_bmratio_q_max);
    const int m = P.syserr_BQ_strand_favor_mul;
    const int q_fw = (rssf * m - minABQa * aDPf * m / 10 + rssr - minABQa * aDPr / 10) / m;
    const int q_rv = (rssr * m - minABQa * aDPr * m / 10 + rssf - minABQa * aDPf / 10) / m;
    const int q_2d = rssb - minABQa * (aDPf + aDPr) / 10;
    const int a_rmsBQ = rssb / imax(1, aDPf + aDPr);
    const int bMQraw = VQP(R, UVC_VQ_bMQ, sym, x);
    f.bMQ = (int)round(sqrt((double)(((long long)bMQraw * SQR_QUAL_DIV) / imax(f.bDPf + f.bDPr, 1))) + (double)(1.0 - FLT_EPS));
    f.aBQQ = imax(a_rmsBQ, P.syserr_BQ_prior + imax(q_2d, imax(q_fw, q_rv)));
    f.a2BQf = rssf; f.a2BQr = rssr; f.aBQ = a_rmsBQ;
    f.bIAQb = VQP(R, UVC_VQ_bIAQb, sym, x); f.bIADb = VQP(R, UVC_VQ_bIADb, sym, x);
    f.cIAQf = VQP(R, UVC_VQ_cIAQf, sym, x); f.cIADf = VQP(R, UVC_VQ_cIADf, sym, x); f.cIDQf = VQP(R, UVC_VQ_cIDQf, sym, x);
    f.cIAQr = VQP(R, UVC_VQ_cIAQr, sym, x); f.cIADr = VQP(R, UVC_VQ_cIADr, sym, x); f.cIDQr = VQP(R, UVC_VQ_cIDQr, sym, x);
    f.bDPa = bDPa; f.cDP0a = cDP0a; f.gap_len = gap_len;
}

DEV bool short_frag(const Tot &T, int wgs_min) { return (T.APLRI[0] + T.APLRI[2]) < (T.APLRI[1] + T.APLRI[3]) * (long long)wgs_min; }   // does_fmt_imply_short_frag, main.hpp:169-174
DEV double norm_fa(double FA, double refbias) { return (FA + FA * refbias) / (FA + (1.0 - FA) / (1.0 + refbias) + FA * refbias); }           // main.hpp:4253-4256

struct RtrLite { int tracklen, unitlen, anyTR_tracklen; };
DEV RtrLite load_rtr(const RegionDev &R, int idx) { RtrLite r; r.tracklen = RTRP(R, UVC_RTR_tracklen, idx); r.unitlen = RTRP(R, UVC_RTR_unitlen, idx); r.anyTR_tracklen = RTRP(R, UVC_RTR_anyTR_tracklen, idx); return r; }

#define OUT(fld, v) fields[(size_t)(fld) * capacity + rec] = (v)

// BcfFormat_symbol_calc_DPv, main.hpp:4274-4844
DEV void calc_dpv(const RegionDev &R, const UvcParams &P, int64_t x, const Tot &T, Al &f, const RtrLite &rtr1, const RtrLite &rtr2, int refsymbol,
                  double tpfa, int tki_tier2, int32_t *fields, int64_t capacity, int64_t rec) {
    const bool tprov = P.tumor_vcf_is_provided;
    const double unbias_ratio = (!tprov ? 1.0 : sqrt(2.0));
    const double unbias_qualadd = (!tprov ? 0 : 3);
    const int allprior = (!tprov ? 0 : 31);
    const int pcr_dp = (int)T.APDP[5], a_dp = (int)T.APDP[0], near_pcr_clip = (int)T.APDP[9];
    const bool strong_amp = (pcr_dp * 100 > a_dp * 50), weak_amp = (pcr_dp * 100 > a_dp * 30);
    const bool is_rescued = (tpfa >= 0);   // main.hpp:4297-4298
    const double pfa = (is_rescued ? tpfa : 0.5), c2altpc = 0.025;
    const int ADP1 = (int)(T.ADPff0 + T.ADPfr0 + T.ADPrf0 + T.ADPrr0);
    const int aDP = (f.aDPff + f.aDPfr + f.aDPrf + f.aDPrr);
    const int ADP = imax(ADP1, near_pcr_clip);
    const int cDP1 = f.cDP1f + f.cDP1r, CDP1 = T.CDP1b[0] + T.CDP1b[1];
    const int sumCDP2 = T.CDP2b[0] + T.CDP2b[1], sumCDP1 = CDP1;
    const double cFA2 = (f.cDP2f + f.cDP2r + c2altpc) / (sumCDP2 + 1.0);
    const double cFA3 = (f.cDP3f + f.cDP3r + c2altpc) / ((T.CDP3b[0] + T.CDP3b[1]) + 1.0);
    const int symbol = f.symbol;
    double cbP = 1e-9, cbBQ = 1e-9, dir_bias_div = 1.0;
    const bool nmore_amp = (!tprov ? strong_amp : weak_amp);
    if ((nmore_amp && (0x2 == (0x2 & P.nobias_flag))) || ((!nmore_amp) && (0x1 == (0x1 & P.nobias_flag)))) {
        const double oddsA_bias = prob2odds((aDP - f.aP1 + 0.5) / (ADP - T.AP10 + 1.0));
        const double oddsA_nobias = prob2odds((f.aP1 + 0.5) / (T.AP10 + 1.0));
        const bool pos_cb = ((oddsA_bias * P.microadjust_counterbias_pos_odds_ratio < oddsA_nobias * (unbias_ratio - DBL_EPS))
                && (f.aP1 * (unbias_ratio - DBL_EPS) > aDP - f.aP1)
                && ((ADP - T.AP10) * P.microadjust_counterbias_pos_fold_ratio * (unbias_ratio - DBL_EPS) > T.AP10)
                && ((0 == P.primerlen && 0 != P.primerlen2) || !is_subst(symbol)));
        if (pos_cb) cbP = dmax(cbP, (f.aP1 + 0.5) / (lmax(T.AP10, (long long)near_pcr_clip) + 1.0)); else cbP = dmax(cbP, 2e-9);
        if (is_subst(symbol)) {
            const bool f_good = ((T.ADPfr0 + T.ADPrr0) + 150 <= (T.ADPff0 + T.ADPrf0) * 5 * unbias_ratio);
            const bool r_good = ((T.ADPff0 + T.ADPrf0) + 150 <= (T.ADPfr0 + T.ADPrr0) * 5 * unbias_ratio);
            const int avg_f_aBQ = (f.a1BQf / imax(1, f.aDPff + f.aDPrf)), avg_r_aBQ = (f.a1BQr / imax(1, f.aDPfr + f.aDPrr));
            const int avg_f_ABQ = (int)(T.A1BQf0 / lmax(1, T.ADPff0 + T.ADPrf0)), avg_r_ABQ = (int)(T.A1BQr0 / lmax(1, T.ADPfr0 + T.ADPrr0));
            if ((f.a1BQf >= f.a1BQr) && (f_good && r_good) && (avg_f_aBQ + unbias_qualadd >= avg_r_ABQ + 14) && (avg_r_ABQ <= 14 + unbias_qualadd))
                cbBQ = dmax(cbBQ, (f.aDPff + f.aDPrf + 0.5) / (T.ADPff0 + T.ADPrf0 + 1.0));
            if ((f.a1BQr >= f.a1BQf) && (f_good && r_good) && (avg_r_aBQ + unbias_qualadd >= avg_f_ABQ + 14) && (avg_f_ABQ <= 14 + unbias_qualadd))
                cbBQ = dmax(cbBQ, (f.aDPfr + f.aDPrr + 0.5) / (T.ADPfr0 + T.ADPrr0 + 1.0));
        } else dir_bias_div = (1.0 + (unsigned)f.gap_len / (unsigned)P.indel_str_repeatsize_max);
    }
    const long long aDPgap = nnminus(lmax(T.APDP[1], T.APDP[2]), f.aP3);
    const double aDPFAgap = ((rtr1.tracklen + rtr2.tracklen < P.indel_str_repeatsize_max) ? 1.0 : ((f.aP3 + pfa) / (aDPgap + 1.0)));
    const double aDPFA1 = ((aDP + pfa) / (ADP + 1.0));
    const double labelFA = (f.aP2 + 1.5 + f.aP2) / (T.AP20 + 2.0 + f.aP2);
    const double aDPFA = dmin((is_subst(symbol) ? dmin(aDPFA1, dmax(aDPFA1 / 3, aDPFAgap)) : aDPFA1), labelFA * (ADP + 1.0) / (T.AP20 + 0.5) * unbias_ratio);
    const int aDPplus = (is_subst(symbol) ? 0 : ((aDP + 1) * P.bias_prior_DPadd_perc / 100));
    const double dp_coef = ((symbol == UVC_LINK_M) ? dmax(P.contam_any_mul_frac, 1.0 - imax(rtr1.tracklen, rtr2.tracklen) / (lmax(1, lmax(T.ALPL0, T.ARPL0)) / dmax(1.0 / 150.0, (double)T.ABQ20))) : 1.0);
    double aPprior = P.bias_priorfreq_pos, aBprior = P.bias_priorfreq_pos;
    const bool in_indel_read = ((T.APXM[1]) / 15.0 * P.microadjust_bias_pos_indel_fold * (P.bias_prior_var_DP_mul) > (aDP + aDPplus) * dp_coef);
    const bool in_indel_len = (lmax(T.APDP[1], T.APDP[2]) * (P.bias_prior_var_DP_mul) > (aDP + aDPplus) * dp_coef);
    const bool in_indel_rtr = (lmax(T.APDP[3], T.APDP[4]) * (P.bias_prior_var_DP_mul) > (aDP + aDPplus) * dp_coef);
    const bool in_rtr = (imax(rtr1.tracklen, rtr2.tracklen) > round(P.indel_polymerase_size));
    const bool in_dnv_read = ((UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) && (T.APDP[7] * 2 > T.APDP[6]));
    if (in_indel_read || in_dnv_read || ((is_ins(symbol) || is_del(symbol)) && (T.APXM[0] > T.APXM[1] * P.microadjust_bias_pos_indel_misma_to_indel_ratio))) {
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
    OUT(UVC_O_nPF0, (int)round(aPprior)); OUT(UVC_O_nPF1, (int)round(aBprior));
    const double aIprior = (is_subst(symbol) ? P.bias_priorfreq_ipos_snv : P.bias_priorfreq_ipos_indel) + allprior;
    const int homopol_len = ((1 == rtr1.unitlen) ? rtr1.tracklen : 0) + ((1 == rtr2.unitlen) ? rtr2.tracklen : 0);
    const double aSBprior = (is_subst(symbol)
            ? (imin((int)nnminus(f.aBQ, (((UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) && (homopol_len > 0)) ? imin(5 * homopol_len, 20) : 0)), f.bMQ) + P.bias_priorfreq_strand_snv_base)
            : (P.bias_priorfreq_strand_indel)) + allprior;
    const double dedup_A2C1 = dmin(1.0, (double)imax(CDP1, P.bias_reduction_by_high_sequencingDP_min_n_totDepth) / (double)imax(ADP1, 1));
    const double dedup_a2c1 = dmin(1.0, (double)imax(cDP1, P.bias_reduction_by_high_sequencingDP_min_n_altDepth) / (double)imax(aDP, 1));
    const double dff = dmax(dedup_A2C1, dedup_a2c1);
    const double pc_read = (in_indel_read ? P.bias_FA_pseudocount_indel_in_read : 0.5);
    const double aBQ2d = (double)imax(1, f.aBQ2), ABQ2d = (double)lmax(1, T.ABQ20);
    double r2[2];
    dp4(r2, false, false, dff, f.aLP1, aDP, T.ALP20 + f.aLP1 - f.aLP2, ADP, P.powlaw_exponent, phred2nat(aPprior), lmax(1, f.aLPL) / aBQ2d, lmax(1, T.ALPL0) / ABQ2d, pc_read); double aLPFA = r2[0];
    dp4(r2, false, false, dff, f.aRP1, aDP, T.ARP20 + f.aRP1 - f.aRP2, ADP, P.powlaw_exponent, phred2nat(aPprior), lmax(1, f.aRPL) / aBQ2d, lmax(1, T.ARPL0) / ABQ2d, pc_read); double aRPFA = r2[0];
    dp4(r2, false, false, dff, f.aLB1, aDP, T.ALB20 + f.aLB1 - f.aLB2, ADP, P.powlaw_exponent, phred2nat(aBprior), lmax(1, f.aLBL) / aBQ2d, lmax(1, T.ALBL0) / ABQ2d, pc_read); double aLBFA = r2[0];
    dp4(r2, false, false, dff, f.aRB1, aDP, T.ARB20 + f.aRB1 - f.aRB2, ADP, P.powlaw_exponent, phred2nat(aBprior), lmax(1, f.aRBL) / aBQ2d, lmax(1, T.ARBL0) / ABQ2d, pc_read); double aRBFA = r2[0];
    const bool tmore_amp = (!tprov ? weak_amp : strong_amp);
    const int normCDP1 = (T.CDP12b[0] + T.CDP12b[1]) + 1, normBDP = (T.BDPb[0] + T.BDPb[1]) + 1;
    const int c2DP = f.cDP2f + f.cDP2r;
    f.tier2 = (is_rescued ? (tki_tier2 ? 1 : 0) : (((c2DP >= 2) && (normBDP * P.fam_bias_overseq_perc >= normCDP1 * 100) && (T.APDP[11] * 100 > (long long)a_dp * 50)) ? 1 : 0));   // main.hpp:4475
    OUT(UVC_O_tier2, f.tier2);
    const double cFA2L = (f.tier2 ? (((double)(((long long)f.c2LP0 * f.c2LP0) * 2 / lmax(1, (long long)imin(c2DP, f.c2LP0 * 4))) + c2altpc) / (T.C2LP00 + 1.0)) : 1.0);
    const double cFA2R = (f.tier2 ? (((double)(((long long)f.c2RP0 * f.c2RP0) * 2 / lmax(1, (long long)imin(c2DP, f.c2RP0 * 4))) + c2altpc) / (T.C2RP00 + 1.0)) : 1.0);
    double c2LPFA = 1.0, c2RPFA = 1.0, c2LBFA = 1.0, c2RBFA = 1.0;
    if (f.tier2) {
        const double c2Pp = dmax(0.0, aPprior), c2Bp = dmax(0.0, aBprior);
        const double cb = (double)imax(1, f.c2BQ2), CB = (double)lmax(1, T.C2BQ20);
        dp4(r2, false, true, -1, f.c2LP1, c2DP, T.C2LP20 + f.c2LP1 - f.c2LP2, sumCDP2, P.powlaw_exponent, phred2nat(c2Pp), lmax(1, f.c2LPL) / cb, lmax(1, T.C2LPL0) / CB, c2altpc, 1.0); c2LPFA = r2[0];
        dp4(r2, false, true, -1, f.c2RP1, c2DP, T.C2RP20 + f.c2RP1 - f.c2RP2, sumCDP2, P.powlaw_exponent, phred2nat(c2Pp), lmax(1, f.c2RPL) / cb, lmax(1, T.C2RPL0) / CB, c2altpc, 1.0); c2RPFA = r2[0];
        dp4(r2, false, true, -1, f.c2LB1, c2DP, T.C2LB20 + f.c2LB1 - f.c2LB2, sumCDP2, P.powlaw_exponent, phred2nat(c2Bp), lmax(1, f.c2LBL) / cb, lmax(1, T.C2LBL0) / CB, c2altpc, 1.0); c2LBFA = r2[0];
        dp4(r2, false, true, -1, f.c2RB1, c2DP, T.C2RB20 + f.c2RB1 - f.c2RB2, sumCDP2, P.powlaw_exponent, phred2nat(c2Bp), lmax(1, f.c2RBL) / cb, lmax(1, T.C2RBL0) / CB, c2altpc, 1.0); c2RBFA = r2[0];
    }
    double LI2[2], RI2[2];
    {
        const double ALpd = (T.ALI20 + 0.5) / (T.ADPfr0 + T.ADPrr0 - T.ALI20 + 0.5);
        const double aLpd = (f.aLI1 + ALpd / (1.0 + ALpd)) / (f.aDPfr + f.aDPrr - f.aLI1 + 1.0 / (1.0 + ALpd));
        dp4(LI2, false, false, dff, f.aLI1, (f.aDPfr + f.aDPrr), (T.ALI20 + f.aLI1 - f.aLI2), (T.ADPfr0 + T.ADPrr0), P.powlaw_exponent, phred2nat(aIprior), aLpd, ALpd, 0.25, 0.5);
        const double ARpd = (T.ARI20 + 0.5) / (T.ADPff0 + T.ADPrf0 - T.ARI20 + 0.5);
        const double aRpd = (f.aRI1 + ARpd / (1.0 + ARpd)) / (f.aDPff + f.aDPrf - f.aRI1 + 1.0 / (1.0 + ARpd));
        dp4(RI2, false, false, dff, f.aRI1, (f.aDPff + f.aDPrf), (T.ARI20 + f.aRI1 - f.aRI2), (T.ADPff0 + T.ADPrf0), P.powlaw_exponent, phred2nat(aIprior), aRpd, ARpd, 0.25, 0.5);
    }
    double aLIFA = LI2[0] * (tmore_amp ? dir_bias_div : dmax(dir_bias_div, aDPFA / LI2[1]));
    double aRIFA = RI2[0] * (tmore_amp ? dir_bias_div : dmax(dir_bias_div, aDPFA / RI2[1]));
    const double aSIFA = dmax((f.aLI1 + 0.5) / (T.ALI20 + f.aLI1 - f.aLI2 + 1.0), (f.aRI1 + 0.5) / (T.ARI20 + f.aRI1 - f.aRI2 + 1.0));
    const int indel_size = f.gap_len;
    if (is_ins(symbol) || is_del(symbol)) {
        const double coef = imax(1, f.bDPa) / (double)imax(1, f.bDPf + f.bDPr);
        const bool major_reg = ((lmax(T.APDP[1], T.APDP[3]) + lmax(T.APDP[2], T.APDP[4])) * 0.5 * (1.0 + (double)FLT_EPS) < aDP * coef);
        if ((imin(indel_size, P.microadjust_nobias_pos_indel_maxlen) * aDPFA * coef >= P.nobias_pos_indel_lenfrac_thres) ||
            (imax(rtr1.tracklen, rtr2.tracklen) >= P.nobias_pos_indel_str_track_len && major_reg && !(T.APXM[0] > T.APXM[1] * P.microadjust_nobias_pos_indel_misma_to_indel_ratio))) {
            aLPFA += 2.0; aRPFA += 2.0; aLBFA += 2.0; aRBFA += 2.0;
            if (f.tier2) { c2LPFA += 2.0; c2RPFA += 2.0; c2LBFA += 2.0; c2RBFA += 2.0; }
        }
        if (f.bMQ >= P.microadjust_nobias_pos_indel_bMQ && f.a2XM2 * 100 >= aDP * 100 * P.microadjust_nobias_pos_indel_perc) { aLIFA += 2.0; aRIFA += 2.0; }
    } else if (UVC_LINK_M == symbol || UVC_LINK_NN == symbol) {
        const double pc = P.bias_FA_pseudocount_indel_in_read;
        aLBFA = dmin(aLBFA, (pc + f.aLB1) / (double)(pc * 2 + ADP));
        aRBFA = dmin(aRBFA, (pc + f.aRB1) / (double)(pc * 2 + ADP));
    } else if (refsymbol == symbol) { aLIFA = aRIFA = dmax(aLIFA, aRIFA); }
    const long long avg_sqr = lmax(T.APXM[4] / lmax(1, T.APDP[1]), T.APXM[5] / lmax(1, T.APDP[2]));
    if ((!is_subst(symbol)) && ((long long)P.microadjust_nobias_pos_indel_maxlen * P.microadjust_nobias_pos_indel_maxlen < avg_sqr)
        && (UVC_LINK_M == symbol || UVC_LINK_NN == symbol || ((long long)(indel_size * 2) * (indel_size * 2) < avg_sqr))) {
        const double pc = P.bias_FA_pseudocount_indel_in_read;
        const double aLmin = (pc + f.aLP1) / (double)(pc * 2 + T.ALP10), aRmin = (pc + f.aRP1) / (double)(pc * 2 + T.ALP10);   // sic: ALP1 in both (main.hpp:4575-4576)
        aLPFA = dmin(aLPFA, aLmin); aRPFA = dmin(aRPFA, aRmin);
        if (f.tier2) { c2LPFA = dmin(c2LPFA, aLmin); c2RPFA = dmin(c2RPFA, aRmin); }
    }
    if (tprov || (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform)) aLIFA = aRIFA = dmax(aLIFA, aRIFA);
    const double aPFFA = (f.aPF1 + pfa * 100.0) / (T.APF20 + (f.aPF1 - f.aPF2) + 100.0);
    double SS2[2];
    dp4(SS2, true, false, dff, f.aRIf, f.aLIr, T.ARIf0, T.ALIr0, P.powlaw_exponent, phred2nat(aSBprior));
    const double ori_base = (is_subst(symbol) ? P.bias_priorfreq_orientation_snv_base : P.bias_priorfreq_orientation_indel_base) + allprior;
    const double te = dmax(aDPFA, P.bias_orientation_min_effective_allelefrac);
    const double ori_all = log(te * te) + phred2nat(ori_base);
    double RO1[2], RO2[2];
    dp4(RO1, true, false, dff, f.cDP1f, f.cDP1r, T.CDP1b[0], T.CDP1b[1], P.powlaw_exponent, ori_all);
    if (P.bias_is_orientation_artifact_mixed_with_sequencing_error) {
        double c12[2];
        dp4(c12, true, false, dff, f.cDP12f, f.cDP12r, T.CDP12b[0], T.CDP12b[1], P.powlaw_exponent, ori_all);
        if ((T.ADPff0 * 8 >= ADP) && (T.ADPfr0 * 8 >= ADP) && (T.ADPrf0 * 8 >= ADP) && (T.ADPrr0 * 8 >= ADP)) { RO1[0] = c12[0]; RO1[1] = c12[1]; }
    }
    dp4(RO2, true, true, -1, f.cDP2f, f.cDP2r, T.CDP2b[0], T.CDP2b[1], P.powlaw_exponent, ori_all, -1, -1, c2altpc, 1.0);
    double aSSFA = SS2[0] * dir_bias_div, cROFA1 = RO1[0] * dir_bias_div, cROFA2 = RO2[0] * dir_bias_div;
    if (is_ins(symbol) || is_del(symbol)) { f.bAD = imin(f.bAD, f.bDPa); f.AD = imin(f.AD, f.cDP0a); }
    const double bFA = (f.bDPa + pfa) / ((T.BDPb[0] + T.BDPb[1]) + 1.0);
    const double cFA0 = (f.cDP0a + pfa * (short_frag(T, P.lib_wgs_min_avg_fraglen) ? P.lib_nonwgs_ad_pseudocount : 1.0)) / (sumCDP1 + 1.0);
    if ((T.ADPfr0 + T.ADPrr0) * P.microadjust_nobias_strand_all_fold < (T.ADPff0 + T.ADPrf0) * unbias_ratio) { aLIFA += 4.0; aSSFA += 4.0; }
    if ((T.ADPff0 + T.ADPrf0) * P.microadjust_nobias_strand_all_fold < (T.ADPfr0 + T.ADPrr0) * unbias_ratio) { aRIFA += 4.0; aSSFA += 4.0; }
    const double aLPFA2 = dmax(aDPFA * 0.01, aLPFA), aRPFA2 = dmax(aDPFA * 0.01, aRPFA), aLBFA2 = dmax(aDPFA * 0.01, aLBFA), aRBFA2 = dmax(aDPFA * 0.01, aRBFA);
    const double c2LPFA2 = dmax(cFA2 * 0.01, c2LPFA), c2RPFA2 = dmax(cFA2 * 0.01, c2RPFA), c2LBFA2 = dmax(cFA2 * 0.01, c2LBFA), c2RBFA2 = dmax(cFA2 * 0.01, c2RBFA);
    const double aLIFA2 = dmax(aDPFA * 0.01, aLIFA), aRIFA2 = dmax(aDPFA * 0.01, aRIFA), aSSFA2 = dmax(aDPFA * 0.05, aSSFA);
    cROFA1 = dmax(aDPFA * 1e-4, cROFA1); cROFA2 = dmax(aDPFA * 1e-4, cROFA2);
    const double fBTA = (double)((T.BTAb[0] + T.BTAb[1]) + 200), fBTB = (double)((T.BTBb[0] + T.BTBb[1]) + 6);
    const double fbTA = (double)(f.bTAf + f.bTAr + 100), fbTB = (double)(f.bTBf + f.bTBr + 3);
    const long long sl = lmin(
            lmin(lmax(0, f.aLIT / lmax(1, (long long)(f.aDPfr + f.aDPrr)) - P.microadjust_longfrag_sidelength_min), (long long)P.microadjust_longfrag_sidelength_max),
            lmin(lmax(0, f.aRIT / lmax(1, (long long)(f.aDPff + f.aDPrf)) - P.microadjust_longfrag_sidelength_min), (long long)P.microadjust_longfrag_sidelength_max));
    const double sidelen_frac = 1.0 - sl / P.microadjust_longfrag_sidelength_zeroMQpenalty;
    const double _alt_frac = fbTB / fbTA;
    const double alt_frac = (nmore_amp ? (dmax(0.0, _alt_frac - 0.2) * 1.25) : _alt_frac);
    const double nonalt_frac = (fBTB + P.contam_any_mul_frac * fbTB - fbTB) / (fBTA + P.contam_any_mul_frac * fbTA - fbTA);
    const double frac_mut = dmax(P.syserr_MQ_NMR_expfrac, P.syserr_MQ_NMR_altfrac_coef * alt_frac * sidelen_frac - P.syserr_MQ_NMR_nonaltfrac_coef * nonalt_frac);
    f.bNMQ = (int)round(numstates2phred(pow(frac_mut / P.syserr_MQ_NMR_expfrac, (P.syserr_MQ_NMR_pl_exponent))) * (frac_mut));
    OUT(UVC_O_bNMQ, f.bNMQ); OUT(UVC_O_bNMa, (int)round(100 * alt_frac)); OUT(UVC_O_bNMb, (int)round(100 * nonalt_frac));
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
    const double aNCFA = ((!tprov && short_frag(T, P.lib_wgs_min_avg_fraglen) && (is_ins(symbol) || is_del(symbol)) && indel_size >= P.lib_nonwgs_clip_penal_min_indelsize)
            ? dmax((f.aNC + 0.5) / (ADP + 1.0), dbetween((f.cDP1f + f.cDP1r) / 300.0, 1.0 / 3.0, 2.0 / 3.0) * aDPFA) : 2.0);
    const double cb_normalgerm = ((!tprov || !short_frag(T, P.lib_wgs_min_avg_fraglen)) ? 1e-9
            : dbetween(aPFFA * aPFFA * (1.0 / P.lib_nonwgs_normal_full_self_rescue_fa), aPFFA * P.lib_nonwgs_normal_min_self_rescue_fa_ratio, aPFFA));
    const double cbFA = dmax(cbP, dmax(cbBQ, cb_normalgerm));
    const double dedup_FA = (!tprov ? dmin(bFA, cFA0) : dmax(bFA, cFA0));
    const double frac_umi2seg = dmin(1.0, dmin(c23FA / aDPFA, aDPFA / c23FA));
    double refbias = 0;
    if ((is_ins(f.symbol) || is_del(f.symbol)) && is_rescued) {   // main.hpp:4804-4810
        const int isz = f.gap_len;
        const int noinfo = (isz * (is_ins(f.symbol) ? 2 : 1) + imax(isz, imax(rtr1.tracklen, rtr2.anyTR_tracklen)));
        refbias = (double)(noinfo) / ((double)(lmin(T.ALPL0, T.ARPL0) * 2 + noinfo) / (double)(T.ABQ20 + 0.5));
        refbias = dmin(refbias, P.microadjust_refbias_indel_max);
    }
    f.cDP1v = (int)(norm_fa(dmax(dmin(dmin(t1plus, t1only), aNCFA), cbFA), refbias) * sumCDP1 * 100);
    f.cDP1w = (int)(norm_fa(dmax(dmin(aLPFA2, dmin(aRPFA2, dmin(aLBFA2, dmin(aRBFA2, dmin(bFA, aNCFA))))), cbFA), refbias) * sumCDP1 * 100);
    double abc_x = dmin(aPFFA, dedup_FA);
    if (tprov) abc_x = dmax(abc_x, cbFA);
    f.cDP1x = 1 + (int)(abc_x * sumCDP1 * 100);
    const double cFA2c = cFA2 * cFA2 * cFA2;
    const double c2XB = dbetween(3.0 * c2LBFA2 * c2RBFA2 * aSSFA2 / cFA2c, dmin(c2LBFA2, c2RBFA2) / 8.0, dmin(c2LBFA2, c2RBFA2));
    const double c2XP = dbetween(3.0 * c2LPFA2 * c2RPFA2 * aSSFA2 / cFA2c, dmin(c2LPFA2, c2RPFA2) / 8.0, dmin(c2LPFA2, c2RPFA2));
    const double c2XX = dmin(c2XB, c2XP);
    f.cDP2v = (int)(norm_fa(dmax(dmin(dmin(t1plus, dmin(t2only, c2XX)), aNCFA), cbFA * frac_umi2seg), refbias) * sumCDP2 * 100);
    f.cDP2w = (int)(norm_fa(dmax(dmin(c2LPFA2, dmin(c2RPFA2, dmin(c2XX, dmin(c2LBFA2, dmin(c2RBFA2, dmin(cFA2, aNCFA)))))), cbFA * frac_umi2seg), refbias) * sumCDP2 * 100);
    f.cDP2x = 1 + (int)(dmin(aPFFA, c23FA) * sumCDP2 * 100);
    OUT(UVC_O_cDP1v, f.cDP1v); OUT(UVC_O_cDP1w, f.cDP1w); OUT(UVC_O_cDP1x, f.cDP1x); OUT(UVC_O_cDP2v, f.cDP2v); OUT(UVC_O_cDP2w, f.cDP2w); OUT(UVC_O_cDP2x, f.cDP2x);
    OUT(UVC_O_AD, f.AD); OUT(UVC_O_bAD, f.bAD);
}

// BcfFormat_symbol_calc_qual, main.hpp:4908-5343
DEV void calc_qual(const RegionDev &R, const UvcParams &P, const Tot &T, const Al &f, const int CDP1v0, const int CDP1x0,
                   int ins_cdepth, int del_cdepth, int ins1_cdepth, int del1_cdepth, int ru_size, int repeatnum, const RtrLite &rtr1, const RtrLite &rtr2, int refsymbol,
                   double tpfa, int32_t *fields, int64_t capacity, int64_t rec) {
    const bool tprov = P.tumor_vcf_is_provided, is_rescued = tprov;   // the caller passes IS_PROVIDED(vcf_tumor_fname), main.cpp:979
    const int symbol = f.symbol, indel_size = f.gap_len;
    const int sumCDP1 = T.CDP1b[0] + T.CDP1b[1], sumCDP2 = T.CDP2b[0] + T.CDP2b[1], sumBDP = T.BDPb[0] + T.BDPb[1], sumCDP12 = T.CDP12b[0] + T.CDP12b[1];
    const double cFA2 = (f.cDP2f + f.cDP2r + 0.5) / (sumCDP2 + 1.0);
    const int phrederr = sscs_phred(P, refsymbol, symbol) + (!tprov ? 0 : 4);
    const double umi_cFA = (((double)(f.cDP2v) + 0.5) / ((double)(sumCDP2 * 100 + 1.0)));
    const double umi_cFA_w = (((double)(f.cDP2w) + 0.5) / ((double)(sumCDP2 * 100 + 1.0)));
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
    const int aDP = (f.aDPff + f.aDPfr + f.aDPrf + f.aDPrr);
    const int ADP = (int)(T.ADPff0 + T.ADPrf0 + T.ADPfr0 + T.ADPrr0);
    const int cDP0 = (f.cDP1f + f.cDP1r), CDP0 = sumCDP1, cDP2 = (f.cDP2f + f.cDP2r), CDP2 = sumCDP2;
    const int aavgMQ = (int)(f.aMQs / imax(1, aDP));
    const int diffAaMQs = (int)((T.AMQs0 - f.aMQs) / imax(1, ADP - aDP)) - aavgMQ;
    const int noUMI_inc = imin(P.bias_FA_powerlaw_noUMI_phred_inc_snv, aDP / 2);
    const double pl_noUMI = P.powlaw_anyvar_base + (is_subst(symbol) ? noUMI_inc : P.bias_FA_powerlaw_noUMI_phred_inc_indel);
    const int withUMI_inc = imin(P.bias_FA_powerlaw_withUMI_phred_inc_snv - P.bias_FA_powerlaw_noUMI_phred_inc_snv, cDP2 / 2) + noUMI_inc;
    const double pl_withUMI = P.powlaw_anyvar_base + (is_subst(symbol) ? withUMI_inc : P.bias_FA_powerlaw_withUMI_phred_inc_indel);
    const double prior_weight = 1.0 / (f.cDPmf + f.cDPmr + 1.0);
    const int thres_highBQ = (is_subst(symbol) ? P.fam_thres_highBQ_snv : P.fam_thres_highBQ_indel);
    const int cMmQ = (int)round(numstates2phred((f.cDPMf + f.cDPmf + f.cDPMr + f.cDPmr + pow(10.0, thres_highBQ / 10.0) * prior_weight) / (f.cDPmf + f.cDPmr + prior_weight)));
    const int nb1 = f.bIADb * 100 + 1, nb2 = imin(nb1, f.cDP1v + 1);
    const long long pq1 = 10 * f.bIAQb / imax(1, f.bIADb);
    const long long pq2 = pq1 + (long long)round(10 * numstates2phred((double)nb2 / (double)nb1));
    long long duped_binom = ((is_ins(symbol) || is_del(symbol)) ? pq1 : pq2) * nb2 / (10 * 100);
    const long long contam_frag_q = (long long)round(binom_llr(t2n, cDP0, CDP0 - cDP0)) + 9 - 3;
    const int h_snp = imax(0, 2 * P.germ_phred_hetero_snp - P.germ_phred_het3al_snp), h_indel = imax(0, 2 * P.germ_phred_hetero_indel - P.germ_phred_het3al_indel);
    int het3al_inc = (is_subst(symbol) ? h_snp : h_indel);
    if (is_ins(symbol) || is_del(symbol)) het3al_inc = (int)nnminus(h_indel + 1, indel_size);
    const int normcDP1 = (f.cDP12f + f.cDP12r + 1), normCDP1 = sumCDP12 + 1, normBDP = sumBDP + 1;
    const int ddiv = (is_rescued ? 2 : 1);
    const long long dec1a = (((P.fam_min_n_copies / ddiv <= normCDP1) || (P.fam_min_n_copies_DPxAD / ddiv <= (long long)normCDP1 * normcDP1)) ? 0 : (inc1 + 3));
    const long long dec1b = (((long long)((P.fam_min_overseq_perc - 100) / ddiv + 100) * normCDP1 <= (long long)100 * normBDP) ? 0 : (inc1 + 3));
    const long long dec1 = lmax(dec1a, dec1b);
    const long long dec2 = nnminus(thres_highBQ, cMmQ);
    const long long cIADnorm = (long long)(f.cIADf + f.cIADr) * 100 + 1;
    const long long cIADmin = lmin(cIADnorm, (long long)f.cDP2v + 1);
    const long long bq_fw = f.cIAQf + ((long long)f.cIAQr * imin(P.fam_phred_dscs_all - f.cIDQf, f.cIDQr)) / imax(f.cIDQr, 1);
    const long long bq_rv = f.cIAQr + ((long long)f.cIAQf * imin(P.fam_phred_dscs_all - f.cIDQr, f.cIDQf)) / imax(f.cIDQf, 1);
    const long long contam_sscs_q = (long long)round(binom_llr(t2n, cDP2, CDP2 - cDP2)) + 9 - 3;
    long long sscs_binom = ((long long)nnminus_d((double)lmax(bq_fw, bq_rv), numstates2phred(cIADnorm / (double)cIADmin) * cIADnorm / 100.0) * cIADmin) / (cIADnorm);
    if (lmax(bq_fw, bq_rv) > P.microadjust_fam_binom_qual_halving_thres && is_subst(symbol))
        sscs_binom = lmin(sscs_binom, P.microadjust_fam_binom_qual_halving_thres + (lmax(bq_fw, bq_rv) - P.microadjust_fam_binom_qual_halving_thres) / 2);
    sscs_binom -= dec1 + dec2;
    const double bcFA_v = (((double)(f.cDP1v) + 0.5) / (double)(sumCDP1 * 100 + 1.0));
    int pl_v = (int)round(P.powlaw_exponent * numstates2phred(bcFA_v) + (pl_noUMI));
    const double bcFA_w = (((double)(f.cDP1w) + 0.5) / (double)(sumCDP1 * 100 + 1.0));
    int pl_w = (int)round(P.powlaw_exponent * numstates2phred(bcFA_w) + (pl_noUMI) + P.tn_q_inc_max);
    const int ds_pl = (int)round(10 / log(10.0) * dmin(log((f.cDP12f + 0.5) / (T.CDP12b[0] + 1.0)), log((f.cDP12r + 0.5) / (T.CDP12b[1] + 1.0)))) + (phrederr);
    const int ds_binom = 3 * imin(f.cDP2f, f.cDP2r);
    const long long m5 = lmin(lmin(bq_fw, bq_rv), (long long)imin(ds_pl, imin(ds_binom, 3)));
    const int inc2 = (int)lmax(0, m5) * ((cFA2 > 0.002) ? 1 : 0);
    const int dec3 = (is_rescued ? (-3) : ((cFA2 >= 0.003) ? 0 : 5));
    const int base_2 = (int)(pl_withUMI + inc1 + inc2 - dec1 - dec2 - dec3);
    const int base_2tn = (int)(pl_withUMI + inc4tn + inc2 - dec1 - dec2 - dec3);
    int sscs_pl_v = (int)round((P.powlaw_exponent * numstates2phred(umi_cFA) + base_2));
    int sscs_pl_w = (int)round((P.powlaw_exponent * numstates2phred(umi_cFA_w) + base_2tn));
    const double dFA = (double)(f.dDP2 + 0.5) / (double)(T.DDP10 + 1.0);
    const double dSNR = (double)(f.dDP2 + 0.5) / (double)(f.dDP1 + 1.0);
    const double dnormFA = dFA * pow(dSNR, 1.0 / P.powlaw_exponent);
    const long long dscs_est = (long long)round((P.fam_phred_dscs_max + phrederr) / 2.0);
    const long long dFA_binom = (dscs_est - (long long)round(numstates2phred(1.0 / (dnormFA)))) * (long long)f.dDP2 * cIADmin / cIADnorm;
    const int dFA_pl = (int)(P.powlaw_anyvar_base + (dscs_est - P.fam_phred_pow_dscs_all_origin)
            + (int)round(numstates2phred((dnormFA) * dmin(1.0, (double)((f.cDP1v) + 0.5) / (double)(sumCDP1 * 100 + 1.0)))));
    OUT(UVC_O_cMmQ, cMmQ);
    const double eps = (double)FLT_EPS;
    const bool penal_applied = ((UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) && !tprov);
    const int penal_base = (penal_applied ? ((int)round(P.indel_multiallele_samepos_penal / log(2.0) * log((double)dmax(aDP + eps, (double)lmax(T.APDP[1], T.APDP[2])) / (double)(aDP + eps)))) : 0);
    int penal4multi = 0, penal4multi_g = 0, penal4multi_soma = 0, indel_UMI_penal = 0;
    if (indel_size > 0 && f.cDP0a > 0) {
        const double indel_pq = (double)imin(indel_phred_s(P.indel_polymerase_slip_rate, ru_size, repeatnum), 24) + 2 - (double)10;
        const int eff1 = (ru_size * imax(1, repeatnum) - ru_size);
        const int eff2 = (imax(rtr1.tracklen - rtr1.unitlen, rtr2.tracklen - rtr2.unitlen) / 3);
        const int effm = imax(eff1, eff2);
        const double indel_ic = numstates2phred((double)imax(indel_size + (is_ins(symbol) ? 1 : 0), 1) / (double)(effm + 1))
                + (is_ins(symbol) ? (numstates2phred(P.indel_del_to_ins_err_ratio) * imin(200, f.cDP0a) / 200) : 0);
        int ic = (is_ins(symbol) ? ins_cdepth : del_cdepth);
        if (UVC_LINK_D1 == symbol) ic += ins1_cdepth;
        if (UVC_LINK_I1 == symbol) ic = (int)(ic + del1_cdepth / P.indel_del_to_ins_err_ratio);
        const int nearInDelDP = (int)(is_ins(symbol) ? T.APDP[1] : T.APDP[2]);
        int penal1 = (int)round(P.indel_multiallele_samepos_penal / log(2.0) * log((double)(ic + eps) / (double)(f.cDP0a + eps)));
        if (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) penal1 = (int)nnminus_d(penal1, P.indel_multiallele_samepos_penal);
        const int penal2 = (int)round(P.indel_multiallele_diffpos_penal / log(2.0) * log((double)(nearInDelDP + eps) / (double)(imax(aDP, nearInDelDP) + eps)));
        penal4multi_g = (int)((int)round(P.indel_tetraallele_germline_penal_value / log(2.0) * log((double)(ins_cdepth + del_cdepth + eps) / (double)(f.cDP0a + eps))) - P.indel_tetraallele_germline_penal_thres);
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
        if (f.tier2) indel_UMI_penal = (int)nnminus_d((sumBDP + 1.0) / (double)(sumCDP1 + 1.0) * P.fam_indel_nonUMI_phred_dec_per_fold_overseq,
                                                      (P.fam_thres_emperr_all_flat_indel + 1) * P.fam_indel_nonUMI_phred_dec_per_fold_overseq);
    }
    if (oxid && tprov) sscs_binom = lmax(sscs_binom, (long long)imin(aDP, 3));
    OUT(UVC_O_aAaMQ, diffAaMQs);
    const int readlenMQcap = (int)((T.APXM[2]) / lmax(1, T.APDP[0]) - 17);
    const int diffMQ = imax(0, diffAaMQs);
    const bool extra_accurate = (P.inferred_maxMQ > 60);
    const int MQVQadd = ((symbol == refsymbol) ? 0 : (imin(P.germ_phred_homalt_snp, ADP * 3)));
    const int MQVQadd_soma = ((symbol != refsymbol) ? 0 : (imin(P.germ_phred_homalt_snp, ADP * 3)));
    const bool MQ_unadj = (extra_accurate || (!is_subst(symbol)) || (aDP > ADP * 3 / 4));
    const int MQVQminus = (MQ_unadj ? 0 : ((int)nnminus((60 - 30), aavgMQ) * 2 / 5)) + ((MQ_unadj || (refsymbol != symbol)) ? 0 : (int)nnminus(imin(15, diffMQ), aavgMQ));
    int diffMQ2 = diffMQ;
    if (f.bMQ < 20 && !tprov) {
        const double axf = (f.aDPff + f.aDPrf + 0.5), axr = (f.aDPfr + f.aDPrr + 0.5), Axf = (T.ADPff0 + T.ADPrf0 + 1.0), Axr = (T.ADPfr0 + T.ADPrr0 + 1.0);
        if ((axr / Axr) * 2 < (axf / Axf) || (axf / Axf) * 2 < (axr / Axr)
            || (f.aLI1 + 0.5) / (T.ALI20 + 1.0) * (2 * (1.0 + DBL_EPS)) < (axr) / (Axr) || (f.aRI1 + 0.5) / (T.ARI20 + 1.0) * (2 * (1.0 + DBL_EPS)) < (axf) / (Axf)) diffMQ2 = imax(diffMQ2, 20 - imin(f.bMQ, 20));
    }
    const double MQ_base = ((f.bMQ * (P.syserr_MQ_max - P.syserr_MQ_nonref_base) / P.syserr_MQ_max + P.syserr_MQ_nonref_base)) - (int)(diffMQ2) - (int)(f.bNMQ);
    const int sysMQ = (((refsymbol == symbol) && (ADP > aDP * 2)) ? f.bMQ : (int)(MQ_base - (int)(numstates2phred((ADP + 1.0) / (aDP + 0.5)))));
    const bool nonWGS = short_frag(T, P.lib_wgs_min_avg_fraglen);
    const int rescued_MQ = imin((int)nnminus(readlenMQcap, 60), (nonWGS ? P.lib_nonwgs_normal_max_rescued_MQ : P.lib_wgs_normal_max_rescued_MQ));
    int sysMQVQ1 = imin((imax(sysMQ, P.syserr_MQ_min) + MQVQadd), readlenMQcap);
    const int sysBQVQ = (((UVC_PLATFORM_IONTORRENT != P.inferred_sequencing_platform) && is_subst(symbol)) ? f.aBQQ : 200);
    const int pcr_dp = (int)T.APDP[5];
    const bool strong_amp = ((pcr_dp * 100) > T.APDP[0] * 50), weak_amp = ((pcr_dp * 100) > T.APDP[0] * 30);
    const bool tmore_amp = (!tprov ? weak_amp : strong_amp);
    if (tmore_amp && (is_ins(symbol) || is_del(symbol)) && (sysMQVQ1 > 70) && (T.APXM[1] / lmax(T.APDP[0], 1) > 20))
        sysMQVQ1 = (int)(70 + ((sysMQVQ1 - 70) * 5 / (T.APXM[1] / lmax(T.APDP[0], 1) - 15)));
    int penal_add = 0;
    if (!tprov) {
        const long long delAPDP = lmax(T.APDP[2], T.APDP[4]);
        const long long snv_dp = T.APDP[6];
        if ((T.APDP[0] < 3 * delAPDP) && (T.APDP[0] < 3 * snv_dp) && (aDP * 3 < delAPDP) && (aDP * 3 < snv_dp) && is_subst(symbol) && (rtr2.tracklen >= 8 * rtr2.unitlen))
            penal_add = P.microadjust_germline_mix_with_del_snv_penalty;
        if (tmore_amp && is_del(symbol)) {
            if (aDP * 4 < T.APDP[2]) penal_add = imax(penal_add, 5);
            else if (f.cDP0a * 3 < 2 * (del_cdepth)) penal_add = imax(penal_add, 2);
        }
    }
    const int sysMQVQ = imax(0, sysMQVQ1);
    const int penal_base2 = penal_base + penal_add;
    const long long fx = T.ADPff0 + T.ADPfr0, rx = T.ADPrf0 + T.ADPrr0, xf = T.ADPff0 + T.ADPrf0, xr = T.ADPfr0 + T.ADPrr0;
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
        const int floor_v = (int)(dmin(P.germ_phred_homalt_indel + numstates2phred(umi_cFA), (double)(f.cDP2v * 3 / 100)) + (double)(((is_ins(symbol) ? 1 : 0) - 1) * 3));
        mincVQ2 = imax(mincVQ2, floor_v);
    }
    const long long dVQinc = lmin(lmin(dFA_binom, (long long)dFA_pl) - imax(0, imin(cIAQ, cPLQ2)), (long long)P.fam_phred_dscs_inc_max);
    OUT(UVC_O_dVQinc, (int)dVQinc);
    const int cVQ2 = (int)lmin((long long)sysVQsoma, lmin(cIAQ + lmax(0, dVQinc), cPLQ2 + lmax(0, dVQinc))) - penal4multi;
    OUT(UVC_O_cVQ2, imax(mincVQ2, imin(cVQ2, cTINQ)));
    const int cDP1y = (is_rescued ? f.cDP1x : f.cDP1v), CDP1y0 = (is_rescued ? CDP1x0 : CDP1v0);
    const double binom_contam = binom_llr(contamfrac, cDP1y, CDP1y0);
    const double power_contam = round(10.0 / log(10.0) * P.powlaw_exponent * dmax(logit2((cDP1y + 1) / (double)(CDP1y0 + 1), contamfrac), 0.0));
    OUT(UVC_O_CONTQ, (int)dmin(binom_contam, power_contam));
}

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

// candidate gate, main.cpp:801-840
DEV bool gate(const RegionDev &R, const UvcParams &P, int64_t x, int st, int symbol, int refsymbol, int totBDP, bool all_out, bool pos_rescued, int &bdepth, int &cdepth) {
    bdepth = FRP(R, 0, UVC_FRAG_bDP, symbol, x) + FRP(R, 1, UVC_FRAG_bDP, symbol, x);
    cdepth = imax(FAP(R, 0, UVC_FAM_cDP1, symbol, x), FAP(R, 0, UVC_FAM_cDP12, symbol, x)) + imax(FAP(R, 1, UVC_FAM_cDP1, symbol, x), FAP(R, 1, UVC_FAM_cDP12, symbol, x));
    if (P.tumor_vcf_is_provided) return pos_rescued;   // normal sample: every symbol of a position the tumor has a record at, nothing else (main.cpp:832-840)
    if (all_out) return true;
    if (refsymbol != symbol) return !(bdepth < P.min_altdp_thres);
    return !(totBDP - bdepth < P.min_altdp_thres);
}

DEV int group_refsymbol(const RegionDev &R, int zpos, int st) {   // symboltype_to_refsymbol, main.cpp:616-620
    if (st == UVC_LINK_SYMBOL) return UVC_LINK_M;
    const int refidx = zpos - R.beg, refsize = (int)R.npos - 1;
    return ((refsize == (refidx - 1) || (-1 == (refidx - 1))) ? UVC_BASE_NN : (int)R.refsym[refidx - 1]);
}

__global__ void __launch_bounds__(256) k_score_count(RegionDev R, UvcParams P, ScoreCtx C, long long *counts) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long ngroups = 2LL * (C.pos_end - C.pos_beg);
    if (g >= ngroups) return;
    const int zpos = C.pos_beg + (int)(g >> 1), st = (int)(g & 1);
    int n = 0;
    if (!(zpos == C.pos_beg && st == UVC_BASE_SYMBOL && !C.base_at_beg)) {   // main.cpp:643
        const int refpos = (st == UVC_BASE_SYMBOL ? zpos - 1 : zpos);
        const int64_t x = refpos - R.beg;
        const int refsymbol = group_refsymbol(R, zpos, st);
        int totBDP = 0;
        for (int k = 0; k < st_count(st); k++) { const int s = st_symbol(st, k); totBDP += FRP(R, 0, UVC_FRAG_bDP, s, x) + FRP(R, 1, UVC_FRAG_bDP, s, x); }
        bool pos_rescued = false;
        if (P.tumor_vcf_is_provided && C.n_tkeys) { const long long q = tkey_lower_bound(C, refpos, 0); pos_rescued = (q < C.n_tkeys && C.tkeys[q].refpos == refpos); }
        for (int k = 0; k < st_count(st); k++) {
            const int s = st_symbol(st, k);
            int bd, cd;
            if (gate(R, P, x, st, s, refsymbol, totBDP, C.all_out, pos_rescued, bd, cd)) { long long first; int src; n += allele_source(C, P.tumor_vcf_is_provided, refpos, s, first, src); }
        }
    }
    counts[g] = (long long)n | (n > 0 ? (1LL << 32) : 0LL);
}

// three-kernel exclusive scan of the packed (flag, count) words: block-local scan, scan of the block sums, add-back.
#define SCAN_ITEMS 8
#define SCAN_BLOCK 256
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_local(const long long *in, long long *out, long long *block_sums, long long n) {
    __shared__ long long sh[SCAN_BLOCK];
    const long long base = ((long long)blockIdx.x * SCAN_BLOCK + threadIdx.x) * SCAN_ITEMS;
    long long v[SCAN_ITEMS], s = 0;
    for (int i = 0; i < SCAN_ITEMS; i++) { v[i] = (base + i < n ? in[base + i] : 0); s += v[i]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < SCAN_BLOCK; d <<= 1) {   // Hillis-Steele inclusive scan of the per-thread sums
        const long long t = (threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0);
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    long long run = sh[threadIdx.x] - s;
    for (int i = 0; i < SCAN_ITEMS; i++) { if (base + i < n) out[base + i] = run; run += v[i]; }
    if (threadIdx.x == SCAN_BLOCK - 1) block_sums[blockIdx.x] = sh[threadIdx.x];
}
__global__ void __launch_bounds__(1024) k_scan_tops(long long *block_sums, int nblocks, long long *offsets, long long n, long long *total_records) {
    __shared__ long long sh[1024];
    long long carry = 0;
    for (int b0 = 0; b0 < nblocks; b0 += 1024) {
        const int i = b0 + threadIdx.x;
        const long long v = (i < nblocks ? block_sums[i] : 0);
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const long long t = (threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0);
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblocks) block_sums[i] = carry + sh[threadIdx.x] - v;
        const long long tot = sh[1023];
        __syncthreads();
        carry += tot;
    }
    if (threadIdx.x == 0) { offsets[n] = carry; *total_records = PK_COUNT(carry); }
}
__global__ void __launch_bounds__(256) k_scan_add(const long long *in, long long *offsets, const long long *block_sums, int *active, long long n) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const long long o = offsets[g] + block_sums[g / (SCAN_BLOCK * SCAN_ITEMS)];
    offsets[g] = o;
    if (PK_FLAGS(in[g])) active[PK_FLAGS(o)] = (int)g;
}

#define SCORE_LPG 2
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) k_score(RegionDev R, UvcParams P, ScoreCtx C) {
    __shared__ Tot lds_T[128 / SCORE_LPG];
    const long long ngroups = 2LL * (C.pos_end - C.pos_beg);
    const long long n_active = PK_FLAGS(C.offsets[ngroups]);
    // SCORE_LPG adjacent lanes share one (position, symbol type) group: its records are dealt to them round-robin, the cross-allele sums
    // of BcfFormat_symbol_sum_DPv are combined with lane shuffles.  A default-gate group has two records (REF + one ALT): the serial
    // chain of a lane is halved.
    const int sub = (int)(threadIdx.x % SCORE_LPG);
    for (long long ai_ = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / SCORE_LPG; ai_ < n_active; ai_ += ((long long)gridDim.x * blockDim.x) / SCORE_LPG) {
    const long long g = C.active[ai_];
    const long long rec0 = PK_COUNT(C.offsets[g]), nrec = PK_COUNT(C.offsets[g + 1]) - rec0;
    if (nrec == 0 || rec0 + nrec > C.capacity) continue;
    const int zpos = C.pos_beg + (int)(g >> 1), st = (int)(g & 1);
    const int refpos = (st == UVC_BASE_SYMBOL ? zpos - 1 : zpos);
    const int64_t x = refpos - R.beg;
    const int refidx = zpos - R.beg, refsize = (int)R.npos - 1;
    const int refsymbol = group_refsymbol(R, zpos, st);
    int32_t *fields = C.fields; const long long capacity = C.capacity;
    // the symbol-type totals live in LDS, one set per group (the SCORE_LPG lanes of a group compute the same values and store them twice):
    // as a local struct of k_score they went to scratch memory.
    Tot &T = lds_T[threadIdx.x / SCORE_LPG];
    { TotSrc ts; ts.prep32 = R.prep32; ts.seg32 = R.seg32; ts.vq = R.vq; ts.frag = R.frag; ts.fam = R.fam; ts.faminfo32 = R.faminfo32; ts.duplex = R.duplex;
      ts.prep64 = R.prep64; ts.seg64 = R.seg64; ts.faminfo64 = R.faminfo64; ts.npos = R.npos;
      group_totals(ts, x, st, &T); }
    const int totBDP = T.BDPb[0] + T.BDPb[1];
    // homopolymer context for minABQ (main.cpp:623-626, 909-928)
    const int prev1 = ((refidx >= 2) ? (int)R.refsym[refidx - 2] : UVC_BASE_NN), prev2 = ((refidx >= 3) ? (int)R.refsym[refidx - 3] : UVC_BASE_NN);
    const int next1 = ((refidx < refsize) ? (int)R.refsym[refidx] : UVC_BASE_NN), next2 = ((refidx + 1 < refsize) ? (int)R.refsym[refidx + 1] : UVC_BASE_NN);
    const bool hp1 = (prev1 == refsymbol && next1 == refsymbol), hp2 = (prev2 == refsymbol && next2 == refsymbol);
    const int minABQ_snv = (C.is_amplicon ? P.syserr_minABQ_pcr_snv : P.syserr_minABQ_cap_snv), minABQ_indel = (C.is_amplicon ? P.syserr_minABQ_pcr_indel : P.syserr_minABQ_cap_indel);
    const int nrtr = (int)R.npos;
    const RtrLite rtr1 = load_rtr(R, imax(refpos - R.beg, 3) - 3), rtr2 = load_rtr(R, imin(refpos - R.beg + 3, nrtr - 1));
    // InDel depths of the LINK group at zerobased_pos (main.cpp:817-831), shared by both groups of this zerobased_pos
    int ins_cdepth = 0, del_cdepth = 0, ins1_cdepth = 0, del1_cdepth = 0;
    {
        const int64_t xz = zpos - R.beg;
        for (int k = 1; k < 7; k++) {
            const int s = st_symbol(UVC_LINK_SYMBOL, k);
            const int cd = imax(FAP(R, 0, UVC_FAM_cDP1, s, xz), FAP(R, 0, UVC_FAM_cDP12, s, xz)) + imax(FAP(R, 1, UVC_FAM_cDP1, s, xz), FAP(R, 1, UVC_FAM_cDP12, s, xz));
            if (is_ins(s)) { ins_cdepth += cd; if (UVC_LINK_I1 == s) ins1_cdepth += cd; } else { del_cdepth += cd; if (UVC_LINK_D1 == s) del1_cdepth += cd; }
        }
    }
    int ru_size, repeatnum;
    indel_context(R, refidx, P.indel_str_repeatsize_max, ru_size, repeatnum);
    // pass 1: init + calc_DPv, cross-allele sums (BcfFormat_symbol_sum_DPv, main.hpp:4888-4906)
    int s1[6] = { 0, 0, 0, 0, 0, 0 }, s2[6] = { 0, 0, 0, 0, 0, 0 };
    long long rec = rec0;
    for (int pass = 0; pass < 2; pass++) {
        for (long long my = sub; my < nrec; my += SCORE_LPG) {   // this lane's records; all lanes run the body below at the same time
            rec = rec0 + my;
            int symbol = -1, ai = 0, src = 0, bdepth = 0, cdepth = 0; long long first = -1;
            {   // which (symbol, allele) is record `my` of the group: the enumeration of k_score_count again
                long long cnt = 0;
                for (int k = 0; k < st_count(st) && symbol < 0; k++) {
                    const int s2_ = st_symbol(st, k);
                    int bd, cd;
                    if (!gate(R, P, x, st, s2_, refsymbol, totBDP, C.all_out, true /* an active group of a normal sample is a rescued position */, bd, cd)) continue;
                    long long f1; int sr;
                    const int m = allele_source(C, P.tumor_vcf_is_provided, refpos, s2_, f1, sr);
                    if (my < cnt + m) { symbol = s2_; ai = (int)(my - cnt); first = f1; src = sr; bdepth = bd; cdepth = cd; }
                    cnt += m;
                }
            }
            if (symbol < 0) continue;   // cannot happen: nrec is the count of this enumeration
            {
                int bDPa = bdepth, cDP0a = cdepth, glen = 0, tki_tier2 = 0, gap_row = -1;
                double tpfa_dpv = -1.0, tpfa_qual = -1.0;
                int tkey_idx = -1;
                if (src == 2) {   // tumor record: main.cpp:935, 985-986
                    const UvcTumorKey &tk = C.tkeys[first + ai];
                    tkey_idx = (int)(first + ai);
                    tpfa_dpv = (double)(tk.cDP1x + 1) / (double)(tk.CDP1x + 2); tpfa_qual = (double)(tk.bDP + 0.5) / (double)(tk.BDP + 1.0); tki_tier2 = tk.tier2;
                    if (is_ins(symbol) || is_del(symbol)) glen = tk.indel_len;
                } else if (is_ins(symbol) || is_del(symbol)) {
                    if (src == 1) { const UvcIndelAllele &al = C.alleles[first + ai]; bDPa = al.bDPa; cDP0a = al.cDP0a; glen = al.indel_len; gap_row = C.allele_rows[first + ai]; }
                    else {   // no fragment carries this symbol here: "Invalid indel detected", the allele is the symbol's description text (main.hpp:5415-5423)
                        bDPa = 0; cDP0a = 0;
                        glen = ((symbol == UVC_LINK_D3P || symbol == UVC_LINK_I3P) ? 6 : 5);   // strlen("<LD3P>") / strlen("<LD2>") etc., main_conversion.hpp:336-346
                    }
                }
                const int minABQ = (is_subst(symbol) ? (int)nnminus(minABQ_snv, (hp1 ? (hp2 ? 20 : 10) : 0)) : minABQ_indel);
                Al f;
                allele_load(R, P, x, symbol, T, bDPa, cDP0a, glen, minABQ, f);
                if (pass == 0) {
                    OUT(UVC_O_refpos, refpos); OUT(UVC_O_symbol, symbol); OUT(UVC_O_refsymbol, refsymbol);
                    OUT(UVC_O_DP, T.CDP1b[0] + T.CDP1b[1]); OUT(UVC_O_bDP, T.BDPb[0] + T.BDPb[1]); OUT(UVC_O_c2DP, T.CDP2b[0] + T.CDP2b[1]); OUT(UVC_O_c2AD, f.cDP2f + f.cDP2r);
                    OUT(UVC_O_bDPa, bDPa); OUT(UVC_O_cDP0a, cDP0a); OUT(UVC_O_gapSa, gap_row); OUT(UVC_O_gapSa_len, glen); OUT(UVC_O_tkey, tkey_idx);
                    OUT(UVC_O_a2BQf, f.a2BQf); OUT(UVC_O_a2BQr, f.a2BQr); OUT(UVC_O_aBQ, f.aBQ); OUT(UVC_O_aBQQ, f.aBQQ); OUT(UVC_O_bMQ, f.bMQ);
                    calc_dpv(R, P, x, T, f, rtr1, rtr2, refsymbol, tpfa_dpv, tki_tier2, fields, capacity, rec);
                    const int v[6] = { f.cDP1v, f.cDP1w, f.cDP1x, f.cDP2v, f.cDP2w, f.cDP2x };
                    for (int i = 0; i < 6; i++) s1[i] += v[i];
                    if (UVC_BASE_NN == symbol || UVC_LINK_NN == symbol) for (int i = 0; i < 6; i++) s2[i] = v[i];
                } else {
                    // restore what calc_DPv produced for this allele
                    f.tier2 = fields[(size_t)UVC_O_tier2 * capacity + rec]; f.bNMQ = fields[(size_t)UVC_O_bNMQ * capacity + rec];
                    f.cDP1v = fields[(size_t)UVC_O_cDP1v * capacity + rec]; f.cDP1w = fields[(size_t)UVC_O_cDP1w * capacity + rec]; f.cDP1x = fields[(size_t)UVC_O_cDP1x * capacity + rec];
                    f.cDP2v = fields[(size_t)UVC_O_cDP2v * capacity + rec]; f.cDP2w = fields[(size_t)UVC_O_cDP2w * capacity + rec]; f.cDP2x = fields[(size_t)UVC_O_cDP2x * capacity + rec];
                    for (int i = 0; i < 6; i++) { OUT(UVC_O_CDP1v0 + 2 * i, s1[i]); OUT(UVC_O_CDP1v0 + 2 * i + 1, s2[i]); }
                    calc_qual(R, P, T, f, s1[0], s1[2], ins_cdepth, del_cdepth, ins1_cdepth, del1_cdepth, ru_size, repeatnum, rtr1, rtr2, refsymbol, tpfa_qual, fields, capacity, rec);
                }
            }
        }
        if (pass == 0) {   // the lanes of the group add up their parts of the sums (the NN allele is with exactly one of them)
#pragma unroll
            for (int d = 1; d < SCORE_LPG; d <<= 1) for (int i = 0; i < 6; i++) { s1[i] += __shfl_xor(s1[i], d); s2[i] += __shfl_xor(s2[i], d); }
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------
// k_call: the calling step behind calc_qual -- main.cpp:990-1168, output_germline (main.hpp:5483-5775) and the arithmetic of
// append_vcf_record (main.hpp:6027-6272).  One thread per zerobased_pos: both symbol-type groups of it, because vAC and
// "a GERMLINE line was written here" cross the two.  Works on the records k_score wrote plus a few plane reads.
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

__global__ void __launch_bounds__(128) k_call(RegionDev R, UvcParams P, ScoreCtx C) {
    const long long npos = C.pos_end - C.pos_beg;
    int32_t *fields = C.fields; const long long capacity = C.capacity;
    const bool tprov = (P.tumor_vcf_is_provided != 0);
    // one thread per zerobased_pos that has records: the first active group of the position stands for it
    const long long n_active = PK_FLAGS(C.offsets[2 * npos]);
    for (long long ai_ = (long long)blockIdx.x * blockDim.x + threadIdx.x; ai_ < n_active; ai_ += (long long)gridDim.x * blockDim.x) {
        const long long g0 = C.active[ai_];
        if ((g0 & 1) && ai_ > 0 && C.active[ai_ - 1] == g0 - 1) continue;   // the BASE group of this position is active too and does the work
        const long long zi = g0 >> 1;
        const int zpos = C.pos_beg + (int)zi;
        long long rec0[2], nrec[2];
        for (int st = 0; st < 2; st++) { const long long g = 2 * zi + st; rec0[st] = PK_COUNT(C.offsets[g]); nrec[st] = PK_COUNT(C.offsets[g + 1]) - rec0[st]; if (rec0[st] + nrec[st] > capacity) nrec[st] = 0; }
        if (nrec[0] + nrec[1] == 0) continue;
        int vAC[2] = { 0, 0 };
        bool germ_any = false;
        for (int st = 0; st < 2; st++) {
            if (nrec[st] == 0) continue;
            const long long r0 = rec0[st], r1 = rec0[st] + nrec[st];
            const int refsymbol = group_refsymbol(R, zpos, st);
            const int het3al = ((UVC_BASE_SYMBOL == st) ? P.germ_phred_het3al_snp : P.germ_phred_het3al_indel);
            // ---- vAC and the two best non-reference alleles (main.cpp:990-1016) ----
            long long top[2] = { -1, -1 };
            for (long long r = r0; r < r1; r++) {
                if (FLD(UVC_O_symbol, r) == refsymbol) continue;
                if (imax(FLD(UVC_O_cVQ1, r), FLD(UVC_O_cVQ2, r)) >= het3al) vAC[st] += 1;
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
            germ_any = germ_any || (emit != 0);
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
        }
        // ---- per record: main.cpp:1081-1147 + append_vcf_record ----
        for (int st = 0; st < 2; st++) {
            if (nrec[st] == 0) continue;
            const long long r0 = rec0[st], r1 = rec0[st] + nrec[st];
            const int refsymbol = group_refsymbol(R, zpos, st);
            const int refpos = (st == UVC_BASE_SYMBOL ? zpos - 1 : zpos);
            const int64_t x = refpos - R.beg;
            const int ref_bDP = FRP(R, 0, UVC_FRAG_bDP, refsymbol, x) + FRP(R, 1, UVC_FRAG_bDP, refsymbol, x);
            long long ABQ2_0 = 0;
            for (int k = 0; k < st_count(st); k++) ABQ2_0 += S32(R, UVC_S_aBQ2, st_symbol(st, k), x);
            ABQ2_0 = (int)ABQ2_0;   // the int32 FORMAT field truncates the sum
            const bool should_output_ref_allele = (C.all_out || germ_any);
            for (long long rec = r0; rec < r1; rec++) {
                const int symbol = FLD(UVC_O_symbol, rec);
                OUT(UVC_O_vAC0, vAC[0]); OUT(UVC_O_vAC1, vAC[1]);
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
                    const int own_bDP = FRP(R, 0, UVC_FRAG_bDP, symbol, x) + FRP(R, 1, UVC_FRAG_bDP, symbol, x);
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
                        const long long LI = P64(R, UVC_P_a_LI, x) + P64(R, UVC_P_a_RI, x), LIDP = (long long)P32(R, UVC_P_a_LIDP, x) + P32(R, UVC_P_a_RIDP, x);
                        if (LI < LIDP * (long long)P.lib_wgs_min_avg_fraglen) { add1 = P.lib_nonwgs_normal_add_mul_ad * nfm_cDP1x / 100.0; add2 = P.lib_nonwgs_normal_add_mul_ad * nfm_cDP2x / 100.0; }
                        if (t_tDP > 500 && FLD(UVC_O_DP, rec) > 500 && is_del(symbol) && (long long)P32(R, UVC_P_a_near_del_dp, x) * 3 > (long long)P32(R, UVC_P_a_dp, x)) tn_dec_both = imin((int)nnminus(nfm_cVQ1, 31), 9);
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
                    const int vad1 = S32(R, UVC_S_aBQ2, symbol, x); const long long vdp1 = ABQ2_0;
                    const bool keep_var = ((((double)v >= P.vqual) || ((!tprov) && ((vad1 >= P.vad1 && vdp1 >= P.vdp1 && (vdp1 * P.vfa1) <= vad1) || (t_bDP >= P.vad2 && t_BDP >= P.vdp2 && (t_BDP * P.vfa2) <= t_bDP))))
                                           && (symbol != refsymbol || should_output_ref_allele));
                    o_keep = (keep_var && t_bDP >= ((symbol == refsymbol) ? P.min_r_ad : P.min_a_ad)) ? 1 : 0;
                }
                OUT(UVC_O_out, o_out); OUT(UVC_O_vHGQ, o_vHGQ); OUT(UVC_O_NLODQ, o_NLODQ); OUT(UVC_O_NLODV, o_NLODV); OUT(UVC_O_TLODQ, o_TLODQ); OUT(UVC_O_SomaticQ, o_SQ);
                for (int i = 0; i < 4; i++) { OUT(UVC_O_TNBQF0 + i, bq4[i]); OUT(UVC_O_TNCQF0 + i, cq4[i]); }
                OUT(UVC_O_QUAL, o_QUAL); OUT(UVC_O_FILTER, o_FILTER); OUT(UVC_O_keep, o_keep);
            }
        }
    }
}

// ---- UvcScoreRequest::kept_only: the groups that are written travel, nothing else ----
// A (zerobased_pos, symbol type) group is kept iff one of its records is written (keep && out) or it has a GERMLINE line (germ_emit): the
// record writer reads the REF record and the genotype's records of such a group and nothing of the others.  Kept groups keep their order;
// germ_ref / germ_alt1 / germ_alt2 (record indices inside the group) move with it.
__global__ void __launch_bounds__(256) k_keep_count(ScoreCtx C, long long ngroups, long long *counts2) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const long long r0 = PK_COUNT(C.offsets[g]), r1 = PK_COUNT(C.offsets[g + 1]);
    const int32_t *fields = C.fields; const long long capacity = C.capacity;
    long long n = 0;
    if (r1 <= capacity) for (long long r = r0; r < r1; r++) if ((FLD(UVC_O_keep, r) && FLD(UVC_O_out, r)) || FLD(UVC_O_germ_emit, r)) { n = r1 - r0; break; }
    counts2[g] = n;
}
#define KEEP_LANES 8
__global__ void __launch_bounds__(256) k_keep_copy(ScoreCtx C, long long ngroups, const long long *counts2, const long long *offsets2, int32_t *fields2) {
    const long long n_active = PK_FLAGS(C.offsets[ngroups]);
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long ai = t / KEEP_LANES; const int sub = (int)(t % KEEP_LANES);
    if (ai >= n_active) return;
    const long long g = C.active[ai];
    const long long n = counts2[g];
    if (n == 0) return;
    const long long r0 = PK_COUNT(C.offsets[g]), q0 = PK_COUNT(offsets2[g]);
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

extern "C" int uvc_launch_score(const RegionDev *R, const UvcParams *P, const UvcScoreRequest *req, const UvcIndelAllele *d_alleles, const int32_t *d_allele_rows, int64_t n_alleles,
                                const UvcGapRow *d_gap_rows, const uint8_t *d_gap_seq, const UvcTumorKey *d_tkeys, int32_t *d_fields, int64_t capacity, int64_t *d_count /* [2]: all records, kept records */,
                                long long *scratch /* uvc_score_scratch_bytes */, int32_t *d_fields_kept /* kept_only: a second [fields][capacity] array */, hipStream_t s) {
    ScoreCtx C;
    C.pos_beg = req->pos_beg; C.pos_end = req->pos_end; C.all_out = (req->all_out || P->should_output_all) ? 1 : 0; C.is_amplicon = req->is_amplicon; C.base_at_beg = req->base_at_pos_beg ? 1 : 0;
    C.alleles = d_alleles; C.allele_rows = d_allele_rows; C.n_alleles = n_alleles; C.gap_rows = d_gap_rows; C.gap_seq = d_gap_seq; C.tkeys = d_tkeys; C.n_tkeys = (d_tkeys ? req->n_tumor_keys : 0); C.fields = d_fields; C.capacity = capacity;
    const long long ngroups = 2LL * (C.pos_end - C.pos_beg);
    if (ngroups <= 0) return 0;
    const int nblocks = (int)((ngroups + SCAN_BLOCK * SCAN_ITEMS - 1) / (SCAN_BLOCK * SCAN_ITEMS));
    long long *counts = scratch, *offsets = scratch + ngroups, *block_sums = offsets + ngroups + 1;
    C.offsets = offsets; C.active = (int *)(block_sums + nblocks + 1);
    hipLaunchKernelGGL(k_score_count, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, s, *R, *P, C, counts);
    hipLaunchKernelGGL(k_scan_local, dim3(nblocks), dim3(SCAN_BLOCK), 0, s, counts, offsets, block_sums, ngroups);
    hipLaunchKernelGGL(k_scan_tops, dim3(1), dim3(1024), 0, s, block_sums, nblocks, offsets, ngroups, (long long *)d_count);
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, s, counts, offsets, block_sums, C.active, ngroups);
    // k_score holds one wave per SIMD (512 registers): 512 blocks of two waves are one wave on every SIMD of the 256 CUs.  The list length lives on
    // the device; with the default gate most blocks find nothing and leave, with -A (every group active) 512 blocks is the smallest grid
    // that leaves no SIMD idle (a 200 kb tile asked for 391).
    const long long want_blocks = (ngroups / 16 * SCORE_LPG + 127) / 128;
    const unsigned grid = (unsigned)(want_blocks < 4096 ? (want_blocks > 512 ? want_blocks : 512) : 4096);
    hipLaunchKernelGGL(k_score, dim3(grid), dim3(128), 0, s, *R, *P, C);
    const long long npos_scored = C.pos_end - C.pos_beg;
    hipLaunchKernelGGL(k_call, dim3((unsigned)((npos_scored / 8 + 127) / 128 < 2048 ? (npos_scored / 8 + 127) / 128 + 1 : 2048)), dim3(128), 0, s, *R, *P, C);
    if (req->kept_only && d_fields_kept) {
        // second half of the scratch: counts2 [ngroups], offsets2 [ngroups + 1], block sums [nblocks + 1]
        long long *counts2 = (long long *)((char *)scratch + (((size_t)((2 * ngroups + 1 + nblocks + 1) * 8 + ngroups * 4 + 64) + 7) & ~(size_t)7));
        long long *offsets2 = counts2 + ngroups, *block_sums2 = offsets2 + ngroups + 1;
        hipLaunchKernelGGL(k_keep_count, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, s, C, ngroups, counts2);
        hipLaunchKernelGGL(k_scan_local, dim3(nblocks), dim3(SCAN_BLOCK), 0, s, counts2, offsets2, block_sums2, ngroups);
        hipLaunchKernelGGL(k_scan_tops, dim3(1), dim3(1024), 0, s, block_sums2, nblocks, offsets2, ngroups, (long long *)d_count + 1);
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, s, counts2, offsets2, block_sums2, (int *)nullptr, ngroups);   // counts2 carries no flag bits: nothing is listed
        // one thread per (active group, field lane); the grid covers every group that could be active (the list length lives on the device)
        const long long max_active = ngroups;
        const long long threads = max_active * KEEP_LANES;
        hipLaunchKernelGGL(k_keep_copy, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, C, ngroups, counts2, offsets2, d_fields_kept);
    }
    return 0;
}
extern "C" size_t uvc_score_scratch_bytes(int64_t npos_scored) {
    const long long ngroups = 2LL * npos_scored;
    const long long nblocks = (ngroups + SCAN_BLOCK * SCAN_ITEMS - 1) / (SCAN_BLOCK * SCAN_ITEMS);
    const size_t first = (size_t)((2 * ngroups + 1 + nblocks + 1) * 8 + ngroups * 4 + 64);
    return first + (size_t)((2 * ngroups + 1 + nblocks + 1) * 8 + 64);   // + the kept_only scan
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
