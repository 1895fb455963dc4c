// TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT (see oracle_common.hpp header).
//
// CPU restatement of the scoring half of the hot path (tumor-only and, with UvcTumorKey records, the normal sample of a T/N pair):
//   per-position driver loop             main.cpp:608-1000
//   BcfFormat_symboltype_init            main.hpp:3889-4081
//   BcfFormat_symbol_init + VQ fmts      main.hpp:3820-3887, 4094-4251
//   BcfFormat_symbol_calc_DPv            main.hpp:4274-4844  (dp4_to_pcFA main_conversion.hpp:798-849)
//   BcfFormat_symbol_sum_DPv             main.hpp:4888-4906
//   BcfFormat_symbol_calc_qual           main.hpp:4908-5343  (calc_binom_10log10_likeratio main_conversion.hpp:222-237)
// The T/N rescue branches (tki, IS_PROVIDED(vcf_tumor_fname)) read their tumor records from UvcScoreRequest::tumor_keys.
#include "oracle_common.hpp"
#include <array>
#include <map>

namespace uvco {

i32 oracle_sscs_phred(const UvcParams &P, int con_symbol, int alt_symbol);

// calc_binom_10log10_likeratio<TIsBiDirectional, TSetMaxProbToOne>, main_conversion.hpp:222-237
double calc_binom_10log10_likeratio(double prob, double a, double b, bool bidirectional, bool set_max_prob_to_one) {
    if (set_max_prob_to_one) prob = min_(1.0, prob);
    prob = (prob + DBL_EPSILON) / (1.0 + (2.0 * DBL_EPSILON));
    a += DBL_EPSILON; b += DBL_EPSILON;
    double A = (prob) * (a + b);
    double B = (1.0 - prob) * (a + b);
    if (bidirectional || a > A) return 10.0 / log(10.0) * (a * log(a / A) + b * log(b / B));
    return 0.0;
}
static inline double prob2odds(double p) { return p / (1.0 - p); }                                  // main_conversion.hpp:191-196
static inline double logit2(double a, double b) { return log(prob2odds((a + DBL_EPSILON) / (a + b + 2.0 * DBL_EPSILON))); }  // :211-219

// dp4_to_pcFA<TBidirectional, TIsOverseqFracDisabled>, main_conversion.hpp:798-849
void dp4_to_pcFA(double out[2], bool bidir, bool overseq_disabled, double overseq_frac, double aADpass, double aADfail, double aDPpass, double aDPfail,
                 double pl_exponent, double n_nats, double aADavgKeyVal, double aDPavgKeyVal, double priorAD, double priorDP) {
    if (!overseq_disabled) { aDPfail *= overseq_frac; aDPpass *= overseq_frac; aADfail *= overseq_frac; aADpass *= overseq_frac; }
    aDPfail += priorDP; aDPpass += priorDP; aADfail += priorAD; aADpass += priorAD;
    const double nobiasFA = (aADfail + aADpass) / (aDPfail + aDPpass);
    if ((aADpass / aDPpass) >= (aADfail / aDPfail)) {
        if (bidir) { std::swap(aDPfail, aDPpass); std::swap(aADfail, aADpass); }
        else { out[0] = (aADpass / aDPpass); out[1] = nobiasFA; return; }
    }
    double aBDfail = aDPfail * 2 - aADfail * 1;
    double aBDpass = aDPpass * 2 - aADpass * 1;
    double aADpassfrac = aADpass / (aADpass + aADfail);
    double aBDpassfrac = aBDpass / (aBDpass + aBDfail);
    if ((!bidir) && (aADavgKeyVal >= 0) && (aDPavgKeyVal >= 0)) {
        aADpassfrac = aADavgKeyVal / (aADavgKeyVal + aDPavgKeyVal * 0.9);
        aBDpassfrac = 1.0 - aADpassfrac;
    }
    double infogain = aADfail * log((1.0 - aADpassfrac) / (1.0 - aBDpassfrac));
    if (bidir) infogain += aADpass * log(aADpassfrac / aBDpassfrac);
    if (infogain <= n_nats) { out[0] = aADfail / aDPfail; out[1] = nobiasFA; }
    else { out[0] = max_(aADpass / aDPpass, (aADfail / aDPfail) * exp((n_nats - infogain) / pl_exponent)); out[1] = nobiasFA; }
}

// the slice of bcfrec::BcfFormat the scoring functions read and write (allele index a = 0 everywhere)
struct Fmt {
    // symbol-type totals: [0] = sum over the type's symbols, [1] = the NN symbol (fill_symboltype_fmt, main.hpp:3745-3793)
    i64 APDP[12], APXM[8], APLRI[4];
    i64 A1BQf[2], A1BQr[2], AMQs[2], AP1[2], AP2[2], ADPff[2], ADPfr[2], ADPrf[2], ADPrr[2];
    double tpfa_dpv, tpfa_qual; int tki_tier2;   // T/N channel inputs of this allele (main.cpp:935, 985-986; main.hpp:4475); -1 / -1 / 0 without a tumor record
    i64 ALP1[2], ALP2[2], ALPL[2], ARP1[2], ARP2[2], ARPL[2], ALB2[2], ALBL[2], ARB2[2], ARBL[2], ABQ2[2], APF2[2], ALI2[2], ARIf[2], ARI2[2], ALIr[2];
    i32 BDPb[2], BTAb[2], BTBb[2], CDP1b[2], CDP12b[2], CDP2b[2], CDP3b[2];
    i64 C2LP2[2], C2LPL[2], C2RP2[2], C2RPL[2], C2LB2[2], C2LBL[2], C2RB2[2], C2RBL[2], C2BQ2[2], C2LP0[2], C2RP0[2];
    i32 DDP1[2], DDP2[2];
    // per allele
    int symbol;
    i32 a1BQf, a1BQr, aMQs, aP1, aP2, aDPff, aDPfr, aDPrf, aDPrr, aLP1, aLP2, aRP1, aRP2, aLB1, aLB2, aRB1, aRB2;
    i64 aLPL, aRPL, aLBL, aRBL, aLIT, aRIT;
    i32 a2XM2, a2BM2, aBQ2, aPF1, aPF2, aLI1, aLI2, aLIr, aRI1, aRI2, aRIf, aP3, aNC;
    i32 bDPf, bTAf, bTBf, bDPr, bTAr, bTBr;
    i32 cDP1f, cDP12f, cDP2f, cDP3f, cDP21f, cDPMf, cDPmf, cDPDf, cDP1r, cDP12r, cDP2r, cDP3r, cDP21r, cDPMr, cDPmr, cDPDr;
    i32 c2LP1, c2LP2, c2RP1, c2RP2, c2LP0, c2RP0, c2LB1, c2LB2, c2RB1, c2RB2, c2BQ2;
    i64 c2LPL, c2RPL, c2LBL, c2RBL;
    i32 dDP1, dDP2;
    i32 DP, AD, bDP, bAD, c2DP, c2AD;
    i32 bMQ, a2BQf, a2BQr, aBQ, aBQQ, bIAQb, bIADb, bIDQb, cIAQf, cIADf, cIDQf, cIAQr, cIADr, cIDQr;
    i32 bDPa, cDP0a, gapSa_len, gapSa_row;
    // outputs of calc_DPv
    i32 nPF[2], bNMa, bNMb, bNMQ, nNFA[6], nAFA[9], nBCFA[10], FTS, tier2;
    i32 FTSpct[19];   // round(100 * biasFA / refFA) of each fmt_bias_push, the number FORMAT/FTS prints behind the bias name (main.hpp:4268)
    i32 cDP1v, cDP1w, cDP1x, cDP2v, cDP2w, cDP2x;
    i32 CDPv[6][2];
    // outputs of calc_qual
    i32 cMmQ, aAaMQ, bMQQ, bIAQ, cIAQ, cPCQ1, cPLQ1, cPCQ2, cPLQ2, bTINQ, cTINQ, gVQ1, cVQ1, dVQinc, cVQ2, CONTQ;
    i32 refpos, refsymbol;
    // the calling step (main.cpp:990-1168, output_germline, append_vcf_record)
    i32 cVQ1M[2], cVQ2M[2], cVQAM[2], cVQSM[2], vAC[2], vNLODQ, GL4[4], GST[8], germ_GT, germ_GQ, germ_emit, germ_ref, germ_alt1, germ_alt2;
    i32 out, vHGQ, NLODQ, NLODV, TLODQ, SomaticQ, TNBQF[4], TNCQF[4], QUALbits, FILTER, keep;
    int g_ref_rel, g_alt1_rel, g_alt2_rel;   // output_germline's ref / alt1 / alt2 as indices into the group (-1 = the padding allele)
    const UvcTumorKey *tkey; i32 tkey_idx;   // the tumor record of this allele (normal sample of a T/N pair) and its index in the request, or NULL / -1
};

// BcfFormat_symboltype_init, main.hpp:3889-4081
static void symboltype_init(Fmt &f, State &S, i32 refpos, int st, i32 bDPcDP[2]) {
    const i64 x = refpos - S.beg;
    const int nn = (st == UVC_BASE_SYMBOL ? UVC_BASE_NN : UVC_LINK_NN);
    const int pidx[12] = { UVC_P_a_dp, UVC_P_a_near_ins_dp, UVC_P_a_near_del_dp, UVC_P_a_near_RTR_ins_dp, UVC_P_a_near_RTR_del_dp, UVC_P_a_pcr_dp,
                           UVC_P_a_snv_dp, UVC_P_a_dnv_dp, UVC_P_a_highBQ_dp, UVC_P_a_near_pcr_clip_dp, UVC_P_a_near_long_clip_dp, UVC_P_a_umi_dp };
    for (int i = 0; i < 12; i++) f.APDP[i] = S.p32(pidx[i], x);
    f.APXM[0] = S.p32(UVC_P_a_XM1500, x); f.APXM[1] = S.p32(UVC_P_a_GO1500, x); f.APXM[2] = S.p32(UVC_P_a_qlen, x); f.APXM[3] = S.p32(UVC_P_a_GAPLEN, x);
    f.APXM[4] = S.p64(UVC_P_a_near_ins_pow2len, x); f.APXM[5] = S.p64(UVC_P_a_near_del_pow2len, x);
    f.APXM[6] = S.p32(UVC_P_a_near_ins_inv100len, x); f.APXM[7] = S.p32(UVC_P_a_near_del_inv100len, x);
    f.APLRI[0] = S.p64(UVC_P_a_LI, x); f.APLRI[1] = S.p32(UVC_P_a_LIDP, x); f.APLRI[2] = S.p64(UVC_P_a_RI, x); f.APLRI[3] = S.p32(UVC_P_a_RIDP, x);
    auto sum_vq = [&](i64 out[2], int fld) { i64 r = 0; for (int k = 0; k < ST_NSYMBOLS[st]; k++) r += S.VQ(fld, ST_SYMBOLS[st][k], x); out[0] = (i32)r; out[1] = S.VQ(fld, nn, x); };
    auto sum_s32 = [&](i64 out[2], int fld) { i64 r = 0; for (int k = 0; k < ST_NSYMBOLS[st]; k++) r += (i64)S.s32(fld, ST_SYMBOLS[st][k], x); out[0] = r; out[1] = S.s32(fld, nn, x); };
    auto sum_s64 = [&](i64 out[2], int fld) { i64 r = 0; for (int k = 0; k < ST_NSYMBOLS[st]; k++) r += S.s64(fld, ST_SYMBOLS[st][k], x); out[0] = r; out[1] = S.s64(fld, nn, x); };
    sum_vq(f.A1BQf, UVC_VQ_a1BQf); sum_vq(f.A1BQr, UVC_VQ_a1BQr);
    // the int32 FORMAT fields truncate the int64 sum on assignment (fmtDP[0] = ret)
    auto t32 = [&](i64 v[2]) { v[0] = (i32)v[0]; v[1] = (i32)v[1]; };
    sum_s32(f.AMQs, UVC_S_aMQs); t32(f.AMQs); sum_s32(f.AP1, UVC_S_aP1); t32(f.AP1); sum_s32(f.AP2, UVC_S_aP2); t32(f.AP2);
    sum_s32(f.ADPff, UVC_S_aDPff); t32(f.ADPff); sum_s32(f.ADPfr, UVC_S_aDPfr); t32(f.ADPfr); sum_s32(f.ADPrf, UVC_S_aDPrf); t32(f.ADPrf); sum_s32(f.ADPrr, UVC_S_aDPrr); t32(f.ADPrr);
    sum_s32(f.ALP1, UVC_S_aLP1); t32(f.ALP1); sum_s32(f.ALP2, UVC_S_aLP2); t32(f.ALP2); sum_s32(f.ALPL, UVC_S_aLPL);
    sum_s32(f.ARP1, UVC_S_aRP1); t32(f.ARP1); sum_s32(f.ARP2, UVC_S_aRP2); t32(f.ARP2); sum_s32(f.ARPL, UVC_S_aRPL);
    sum_s32(f.ALB2, UVC_S_aLB2); t32(f.ALB2); sum_s64(f.ALBL, UVC_S64_aLBL);
    sum_s32(f.ARB2, UVC_S_aRB2); t32(f.ARB2); sum_s64(f.ARBL, UVC_S64_aRBL);
    sum_s32(f.ABQ2, UVC_S_aBQ2); t32(f.ABQ2); sum_s32(f.APF2, UVC_S_aPF2); t32(f.APF2);
    sum_s32(f.ALI2, UVC_S_aLI2); t32(f.ALI2); sum_s32(f.ARIf, UVC_S_aRIf); t32(f.ARIf); sum_s32(f.ARI2, UVC_S_aRI2); t32(f.ARI2); sum_s32(f.ALIr, UVC_S_aLIr); t32(f.ALIr);
    auto sum_fr = [&](i32 out[2], int fld, bool isfam) {   // fill_symboltype_fr_fmt: [0] = strand 0, [1] = strand 1
        for (int s = 0; s < 2; s++) { int r = 0; for (int k = 0; k < ST_NSYMBOLS[st]; k++) r += (isfam ? S.FA(s, fld, ST_SYMBOLS[st][k], x) : S.FR(s, fld, ST_SYMBOLS[st][k], x)); out[s] = r; }
    };
    sum_fr(f.BDPb, UVC_FRAG_bDP, false); sum_fr(f.BTAb, UVC_FRAG_bTA, false); sum_fr(f.BTBb, UVC_FRAG_bTB, false);
    sum_fr(f.CDP1b, UVC_FAM_cDP1, true); sum_fr(f.CDP12b, UVC_FAM_cDP12, true); sum_fr(f.CDP2b, UVC_FAM_cDP2, true); sum_fr(f.CDP3b, UVC_FAM_cDP3, true);
    auto sum_fi = [&](i64 out[2], int fld) { i64 r = 0; for (int k = 0; k < ST_NSYMBOLS[st]; k++) r += (i64)S.FI(fld, ST_SYMBOLS[st][k], x); out[0] = r; out[1] = S.FI(fld, nn, x); };
    auto sum_fi64 = [&](i64 out[2], int fld) { i64 r = 0; for (int k = 0; k < ST_NSYMBOLS[st]; k++) r += S.FI64(fld, ST_SYMBOLS[st][k], x); out[0] = r; out[1] = S.FI64(fld, nn, x); };
    sum_fi(f.C2LP2, UVC_FI_c2LP2); t32(f.C2LP2); sum_fi(f.C2LPL, UVC_FI_c2LPL); sum_fi(f.C2RP2, UVC_FI_c2RP2); t32(f.C2RP2); sum_fi(f.C2RPL, UVC_FI_c2RPL);
    sum_fi(f.C2LB2, UVC_FI_c2LB2); t32(f.C2LB2); sum_fi64(f.C2LBL, UVC_FI64_c2LBL); sum_fi(f.C2RB2, UVC_FI_c2RB2); t32(f.C2RB2); sum_fi64(f.C2RBL, UVC_FI64_c2RBL);
    sum_fi(f.C2BQ2, UVC_FI_c2BQ2); t32(f.C2BQ2); sum_fi(f.C2LP0, UVC_FI_c2LP0); t32(f.C2LP0); sum_fi(f.C2RP0, UVC_FI_c2RP0); t32(f.C2RP0);
    { int r1 = 0, r2 = 0; for (int k = 0; k < ST_NSYMBOLS[st]; k++) { r1 += S.DU(UVC_DUPLEX_dDP1, ST_SYMBOLS[st][k], x); r2 += S.DU(UVC_DUPLEX_dDP2, ST_SYMBOLS[st][k], x); }
      f.DDP1[0] = r1; f.DDP1[1] = S.DU(UVC_DUPLEX_dDP1, nn, x); f.DDP2[0] = r2; f.DDP2[1] = S.DU(UVC_DUPLEX_dDP2, nn, x); }
    bDPcDP[0] = f.BDPb[0] + f.BDPb[1];
    bDPcDP[1] = max_(f.CDP1b[0], f.CDP12b[0]) + max_(f.CDP1b[1], f.CDP12b[1]);
}

// BcfFormat_symbol_init + fill_symbol_VQ_fmts, main.hpp:4094-4251, 3820-3887
static void symbol_init(Fmt &f, State &S, i32 refpos, int sym, i32 bDPa, i32 cDP0a, i32 gapSa_len, i32 minABQ) {
    const UvcParams &P = S.P;
    const i64 x = refpos - S.beg;
    f.symbol = sym;
    f.a1BQf = S.VQ(UVC_VQ_a1BQf, sym, x); f.a1BQr = S.VQ(UVC_VQ_a1BQr, sym, x);
    f.aMQs = S.s32(UVC_S_aMQs, sym, x); f.aP1 = S.s32(UVC_S_aP1, sym, x); f.aP2 = S.s32(UVC_S_aP2, sym, x);
    f.aDPff = S.s32(UVC_S_aDPff, sym, x); f.aDPfr = S.s32(UVC_S_aDPfr, sym, x); f.aDPrf = S.s32(UVC_S_aDPrf, sym, x); f.aDPrr = S.s32(UVC_S_aDPrr, sym, x);
    f.aLP1 = S.s32(UVC_S_aLP1, sym, x); f.aLP2 = S.s32(UVC_S_aLP2, sym, x); f.aLPL = S.s32(UVC_S_aLPL, sym, x);
    f.aRP1 = S.s32(UVC_S_aRP1, sym, x); f.aRP2 = S.s32(UVC_S_aRP2, sym, x); f.aRPL = S.s32(UVC_S_aRPL, sym, x);
    f.aLB1 = S.s32(UVC_S_aLB1, sym, x); f.aLB2 = S.s32(UVC_S_aLB2, sym, x); f.aLBL = S.s64(UVC_S64_aLBL, sym, x);
    f.aRB1 = S.s32(UVC_S_aRB1, sym, x); f.aRB2 = S.s32(UVC_S_aRB2, sym, x); f.aRBL = S.s64(UVC_S64_aRBL, sym, x);
    f.a2XM2 = S.s32(UVC_S_a2XM2, sym, x); f.a2BM2 = S.s32(UVC_S_a2BM2, sym, x); f.aBQ2 = S.s32(UVC_S_aBQ2, sym, x);
    f.aPF1 = S.s32(UVC_S_aPF1, sym, x); f.aPF2 = S.s32(UVC_S_aPF2, sym, x);
    f.aLI1 = S.s32(UVC_S_aLI1, sym, x); f.aLI2 = S.s32(UVC_S_aLI2, sym, x); f.aLIr = S.s32(UVC_S_aLIr, sym, x);
    f.aRI1 = S.s32(UVC_S_aRI1, sym, x); f.aRI2 = S.s32(UVC_S_aRI2, sym, x); f.aRIf = S.s32(UVC_S_aRIf, sym, x);
    f.bDPf = S.FR(0, UVC_FRAG_bDP, sym, x); f.bTAf = S.FR(0, UVC_FRAG_bTA, sym, x); f.bTBf = S.FR(0, UVC_FRAG_bTB, sym, x);
    f.bDPr = S.FR(1, UVC_FRAG_bDP, sym, x); f.bTAr = S.FR(1, UVC_FRAG_bTA, sym, x); f.bTBr = S.FR(1, UVC_FRAG_bTB, sym, x);
    f.cDP1f = S.FA(0, UVC_FAM_cDP1, sym, x); f.cDP12f = S.FA(0, UVC_FAM_cDP12, sym, x); f.cDP2f = S.FA(0, UVC_FAM_cDP2, sym, x); f.cDP3f = S.FA(0, UVC_FAM_cDP3, sym, x);
    f.cDP21f = S.FA(0, UVC_FAM_cDP21, sym, x); f.cDPMf = S.FA(0, UVC_FAM_cDPM, sym, x); f.cDPmf = S.FA(0, UVC_FAM_cDPm, sym, x); f.cDPDf = S.FA(0, UVC_FAM_cDPD, sym, x);
    f.cDP1r = S.FA(1, UVC_FAM_cDP1, sym, x); f.cDP12r = S.FA(1, UVC_FAM_cDP12, sym, x); f.cDP2r = S.FA(1, UVC_FAM_cDP2, sym, x); f.cDP3r = S.FA(1, UVC_FAM_cDP3, sym, x);
    f.cDP21r = S.FA(1, UVC_FAM_cDP21, sym, x); f.cDPMr = S.FA(1, UVC_FAM_cDPM, sym, x); f.cDPmr = S.FA(1, UVC_FAM_cDPm, sym, x); f.cDPDr = S.FA(1, UVC_FAM_cDPD, sym, x);
    f.c2LP1 = S.FI(UVC_FI_c2LP1, sym, x); f.c2LP2 = S.FI(UVC_FI_c2LP2, sym, x); f.c2LPL = S.FI(UVC_FI_c2LPL, sym, x);
    f.c2RP1 = S.FI(UVC_FI_c2RP1, sym, x); f.c2RP2 = S.FI(UVC_FI_c2RP2, sym, x); f.c2RPL = S.FI(UVC_FI_c2RPL, sym, x);
    f.c2LB1 = S.FI(UVC_FI_c2LB1, sym, x); f.c2LB2 = S.FI(UVC_FI_c2LB2, sym, x); f.c2LBL = S.FI64(UVC_FI64_c2LBL, sym, x);
    f.c2RB1 = S.FI(UVC_FI_c2RB1, sym, x); f.c2RB2 = S.FI(UVC_FI_c2RB2, sym, x); f.c2RBL = S.FI64(UVC_FI64_c2RBL, sym, x);
    f.c2BQ2 = S.FI(UVC_FI_c2BQ2, sym, x); f.c2LP0 = S.FI(UVC_FI_c2LP0, sym, x); f.c2RP0 = S.FI(UVC_FI_c2RP0, sym, x);
    f.dDP1 = S.DU(UVC_DUPLEX_dDP1, sym, x); f.dDP2 = S.DU(UVC_DUPLEX_dDP2, sym, x);
    f.aLIT = S.s64(UVC_S64_aLIT, sym, x); f.aRIT = S.s64(UVC_S64_aRIT, sym, x); f.aP3 = S.s32(UVC_S_aP3, sym, x); f.aNC = S.s32(UVC_S_aNC, sym, x);
    f.DP = f.CDP1b[0] + f.CDP1b[1]; f.AD = f.cDP1f + f.cDP1r;
    f.bDP = f.BDPb[0] + f.BDPb[1]; f.bAD = f.bDPf + f.bDPr;
    f.c2DP = f.CDP2b[0] + f.CDP2b[1]; f.c2AD = f.cDP2f + f.cDP2r;
    // fill_symbol_VQ_fmts
    const i32 a2BQf = S.VQ(UVC_VQ_a2BQf, sym, x), a2BQr = S.VQ(UVC_VQ_a2BQr, sym, x);
    const i32 aDPf = f.aDPff + f.aDPrf, aDPr = f.aDPfr + f.aDPrr;
    const i32 ADP = (i32)(f.ADPff[0] + f.ADPrf[0] + f.ADPfr[0] + f.ADPrr[0]);
    const i32 rssDPfBQ = (i32)(aDPf * sqrt((double)(((i64)a2BQf * SQR_QUAL_DIV_) / max_(1, aDPf))));
    const i32 rssDPrBQ = (i32)(aDPr * sqrt((double)(((i64)a2BQr * SQR_QUAL_DIV_) / max_(1, aDPr))));
    const i32 rssDPbBQ = (i32)((aDPf + aDPr) * sqrt((double)((a2BQf + a2BQr) * SQR_QUAL_DIV_ / max_(1, aDPf + aDPr))));
    const double t = max_(0.0, ((aDPf + aDPr + 0.5) * 2.0 / (ADP + 1.0) - 1.0));
    i32 minABQa = minABQ - (i32)(5 * 10.0 * (t * t));
    const double sbratio = (double)(max_(aDPf, aDPr) * 10 + 10) / (double)(min_(aDPf, aDPr) * 10 + 10);
    minABQa += between_((i32)(sbratio * sbratio) - P.syserr_BQ_sbratio_q_add, 0, P.syserr_BQ_sbratio_q_max);
    const i32 xmratio = (P.syserr_BQ_xmratio_q_max * 10 * (aDPf + aDPr) / max_(1, f.a2XM2));
    const i32 bmratio = (P.syserr_BQ_bmratio_q_max * 10 * (aDPf + aDPr) / max_(1, f.a2BM2));
    minABQa += between_(xmratio - P.syserr_BQ_xmratio_q_add, 0, P.syserr_BQ_xmratio_q_max) + between_(bmratio - P.syserr_BQ_bmratio_q_add, 0, P.syserr_BQ_bmratio_q_max);
    const i32 m = P.syserr_BQ_strand_favor_mul;
    const i32 q_fw = (rssDPfBQ * m - minABQa * aDPf * m / 10 + rssDPrBQ - minABQa * aDPr / 10) / m;
    const i32 q_rv = (rssDPrBQ * m - minABQa * aDPr * m / 10 + rssDPfBQ - minABQa * aDPf / 10) / m;
    const i32 q_2d = (rssDPbBQ) - minABQa * (aDPf + aDPr) / 10;
    const i32 a_rmsBQ = (rssDPbBQ) / max_(1, aDPf + aDPr);
    const i32 bMQraw = S.VQ(UVC_VQ_bMQ, sym, x);
    f.bMQ = (i32)round(sqrt((double)(((i64)bMQraw * SQR_QUAL_DIV_) / max_(f.bDPf + f.bDPr, 1))) + (double)(1.0 - FLT_EPSILON));
    f.aBQQ = max_(a_rmsBQ, P.syserr_BQ_prior + max_(q_2d, max_(q_fw, q_rv)));
    f.a2BQf = rssDPfBQ; f.a2BQr = rssDPrBQ; f.aBQ = a_rmsBQ;
    f.bIAQb = S.VQ(UVC_VQ_bIAQb, sym, x); f.bIADb = S.VQ(UVC_VQ_bIADb, sym, x); f.bIDQb = S.VQ(UVC_VQ_bIDQb, sym, x);
    f.cIAQf = S.VQ(UVC_VQ_cIAQf, sym, x); f.cIADf = S.VQ(UVC_VQ_cIADf, sym, x); f.cIDQf = S.VQ(UVC_VQ_cIDQf, sym, x);
    f.cIAQr = S.VQ(UVC_VQ_cIAQr, sym, x); f.cIADr = S.VQ(UVC_VQ_cIADr, sym, x); f.cIDQr = S.VQ(UVC_VQ_cIDQr, sym, x);
    f.bDPa = bDPa; f.cDP0a = cDP0a; f.gapSa_len = gapSa_len;
}

// does_fmt_imply_short_frag, main.hpp:169-174
static inline bool implies_short_frag(const Fmt &f, i32 wgs_min_avg_fragsize) { return (f.APLRI[0] + f.APLRI[2]) < (f.APLRI[1] + f.APLRI[3]) * (i64)wgs_min_avg_fragsize; }
static inline double norm_fa(double FA, double refbias) { return (FA + FA * refbias) / (FA + (1.0 - FA) / (1.0 + refbias) + FA * refbias); }   // main.hpp:4253-4256

// ---- test hook: what BcfFormat_symboltype_init / BcfFormat_symbol_init gathered for one record, as named numbers, so that an independent
// restatement of calc_DPv / calc_qual (tests/score_restatement.py) starts from the same inputs
#define TRACE_ARR(X) X(APDP, 12) X(APXM, 8) X(APLRI, 4) X(A1BQf, 2) X(A1BQr, 2) X(AMQs, 2) X(AP1, 2) X(AP2, 2) X(ADPff, 2) X(ADPfr, 2) X(ADPrf, 2) X(ADPrr, 2) \
    X(ALP1, 2) X(ALP2, 2) X(ALPL, 2) X(ARP1, 2) X(ARP2, 2) X(ARPL, 2) X(ALB2, 2) X(ALBL, 2) X(ARB2, 2) X(ARBL, 2) X(ABQ2, 2) X(APF2, 2) X(ALI2, 2) X(ARIf, 2) X(ARI2, 2) X(ALIr, 2) \
    X(BDPb, 2) X(BTAb, 2) X(BTBb, 2) X(CDP1b, 2) X(CDP12b, 2) X(CDP2b, 2) X(CDP3b, 2) \
    X(C2LP2, 2) X(C2LPL, 2) X(C2RP2, 2) X(C2RPL, 2) X(C2LB2, 2) X(C2LBL, 2) X(C2RB2, 2) X(C2RBL, 2) X(C2BQ2, 2) X(C2LP0, 2) X(C2RP0, 2) X(DDP1, 2) X(DDP2, 2)
#define TRACE_SCA(X) X(symbol) X(a1BQf) X(a1BQr) X(aMQs) X(aP1) X(aP2) X(aDPff) X(aDPfr) X(aDPrf) X(aDPrr) X(aLP1) X(aLP2) X(aRP1) X(aRP2) X(aLB1) X(aLB2) X(aRB1) X(aRB2) \
    X(aLPL) X(aRPL) X(aLBL) X(aRBL) X(aLIT) X(aRIT) X(a2XM2) X(a2BM2) X(aBQ2) X(aPF1) X(aPF2) X(aLI1) X(aLI2) X(aLIr) X(aRI1) X(aRI2) X(aRIf) X(aP3) X(aNC) \
    X(bDPf) X(bTAf) X(bTBf) X(bDPr) X(bTAr) X(bTBr) X(cDP1f) X(cDP12f) X(cDP2f) X(cDP3f) X(cDP21f) X(cDPMf) X(cDPmf) X(cDPDf) X(cDP1r) X(cDP12r) X(cDP2r) X(cDP3r) X(cDP21r) X(cDPMr) X(cDPmr) X(cDPDr) \
    X(c2LP1) X(c2LP2) X(c2RP1) X(c2RP2) X(c2LP0) X(c2RP0) X(c2LB1) X(c2LB2) X(c2RB1) X(c2RB2) X(c2BQ2) X(c2LPL) X(c2RPL) X(c2LBL) X(c2RBL) X(dDP1) X(dDP2) \
    X(DP) X(AD) X(bDP) X(bAD) X(c2DP) X(c2AD) X(bMQ) X(a2BQf) X(a2BQr) X(aBQ) X(aBQQ) X(bIAQb) X(bIADb) X(bIDQb) X(cIAQf) X(cIADf) X(cIDQf) X(cIAQr) X(cIADr) X(cIDQr) \
    X(bDPa) X(cDP0a) X(gapSa_len) X(refpos) X(refsymbol) X(tki_tier2) X(tpfa_dpv) X(tpfa_qual)
static const char *trace_names() {
    static std::string s;
    if (s.empty()) {
#define X(n, k) for (int i = 0; i < k; i++) { s += #n; s += "["; s += std::to_string(i); s += "];"; }
        TRACE_ARR(X)
#undef X
#define X(n) s += #n ";";
        TRACE_SCA(X)
#undef X
        s += "rtr1_tracklen;rtr1_unitlen;rtr1_anyTR_tracklen;rtr2_tracklen;rtr2_unitlen;rtr2_anyTR_tracklen;";
    }
    return s.c_str();
}
static void trace_record(std::vector<double> &t, const Fmt &f, const Rtr &rtr1, const Rtr &rtr2) {
#define X(n, k) for (int i = 0; i < k; i++) t.push_back((double)f.n[i]);
    TRACE_ARR(X)
#undef X
#define X(n) t.push_back((double)f.n);
    TRACE_SCA(X)
#undef X
    t.push_back(rtr1.tracklen); t.push_back(rtr1.unitlen); t.push_back(rtr1.anyTR_tracklen); t.push_back(rtr2.tracklen); t.push_back(rtr2.unitlen); t.push_back(rtr2.anyTR_tracklen);
}
const char *score_trace_names() { return trace_names(); }

// BcfFormat_symbol_calc_DPv, main.hpp:4274-4844
static void calc_DPv(Fmt &fmt, const Rtr &rtr1, const Rtr &rtr2, int refsymbol, State &S, i32 refpos) {
    const UvcParams &P = S.P;
    const i64 x = refpos - S.beg;
    const Fmt &f = fmt;
    const bool tprov = P.tumor_vcf_is_provided;
    const double unbias_ratio = (!tprov ? 1.0 : sqrt(2.0));
    const double unbias_qualadd = (!tprov ? 0 : 3);
    const i32 allbias_allprior = (!tprov ? 0 : 31);
    const i32 pcr_dp = S.p32(UVC_P_a_pcr_dp, x), a_dp = S.p32(UVC_P_a_dp, x), near_pcr_clip = S.p32(UVC_P_a_near_pcr_clip_dp, x);
    const bool is_strong_amplicon = (pcr_dp * 100 > a_dp * 50);
    const bool is_weak_amplicon = (pcr_dp * 100 > a_dp * 30);
    const double tpfa = fmt.tpfa_dpv;
    const bool is_rescued = (tpfa >= 0);
    const double pfa = (is_rescued ? tpfa : 0.5);
    const double c2altpc = 0.025;
    const i32 ADP1 = (i32)(f.ADPff[0] + f.ADPfr[0] + f.ADPrf[0] + f.ADPrr[0]);
    const i32 aDP1 = (f.aDPff + f.aDPfr + f.aDPrf + f.aDPrr);
    const i32 aDP = aDP1;
    const i32 ADP = max_(ADP1, near_pcr_clip);
    const i32 cDP1 = f.cDP1f + f.cDP1r;
    const i32 CDP1 = f.CDP1b[0] + f.CDP1b[1];
    const double cFA2 = (f.cDP2f + f.cDP2r + c2altpc) / ((f.CDP2b[0] + f.CDP2b[1]) + 1.0);
    const double cFA3 = (f.cDP3f + f.cDP3r + c2altpc) / ((f.CDP3b[0] + f.CDP3b[1]) + 1.0);
    const int symbol = f.symbol;
    double _cb_P_FA = 1e-9, _cb_BQ_FA = 1e-9, _dir_bias_div = 1.0;
    const bool is_nmore_amplicon = (!tprov ? is_strong_amplicon : is_weak_amplicon);
    if ((is_nmore_amplicon && (0x2 == (0x2 & P.nobias_flag))) || ((!is_nmore_amplicon) && (0x1 == (0x1 & P.nobias_flag)))) {
        const double oddsA_bias = prob2odds((aDP - f.aP1 + 0.5) / (ADP - f.AP1[0] + 1.0));
        const double oddsA_nobias = prob2odds((f.aP1 + 0.5) / (f.AP1[0] + 1.0));
        const bool is_pos_counterbias = ((oddsA_bias * P.microadjust_counterbias_pos_odds_ratio < oddsA_nobias * (unbias_ratio - DBL_EPSILON))
                && (f.aP1 * (unbias_ratio - DBL_EPSILON) > aDP - f.aP1)
                && ((ADP - f.AP1[0]) * P.microadjust_counterbias_pos_fold_ratio * (unbias_ratio - DBL_EPSILON) > f.AP1[0])
                && ((0 == P.primerlen && 0 != P.primerlen2) || !is_subst(symbol)));
        if (is_pos_counterbias) _cb_P_FA = max_(_cb_P_FA, (f.aP1 + 0.5) / (max_((i64)f.AP1[0], (i64)near_pcr_clip) + 1.0));
        else _cb_P_FA = max_(_cb_P_FA, 2e-9);
        if (is_subst(symbol)) {
            const bool f_good = ((f.ADPfr[0] + f.ADPrr[0]) + 150 <= (f.ADPff[0] + f.ADPrf[0]) * 5 * unbias_ratio);
            const bool r_good = ((f.ADPff[0] + f.ADPrf[0]) + 150 <= (f.ADPfr[0] + f.ADPrr[0]) * 5 * unbias_ratio);
            const i32 avg_f_aBQ = (f.a1BQf / max_(1, f.aDPff + f.aDPrf));
            const i32 avg_r_aBQ = (f.a1BQr / max_(1, f.aDPfr + f.aDPrr));
            const i32 avg_f_ABQ = (i32)(f.A1BQf[0] / max_((i64)1, f.ADPff[0] + f.ADPrf[0]));
            const i32 avg_r_ABQ = (i32)(f.A1BQr[0] / max_((i64)1, f.ADPfr[0] + f.ADPrr[0]));
            if ((f.a1BQf >= f.a1BQr) && (f_good && r_good) && (avg_f_aBQ + unbias_qualadd >= avg_r_ABQ + 14) && (avg_r_ABQ <= 14 + unbias_qualadd))
                _cb_BQ_FA = max_(_cb_BQ_FA, (f.aDPff + f.aDPrf + 0.5) / (f.ADPff[0] + f.ADPrf[0] + 1.0));
            if ((f.a1BQr >= f.a1BQf) && (f_good && r_good) && (avg_r_aBQ + unbias_qualadd >= avg_f_ABQ + 14) && (avg_f_ABQ <= 14 + unbias_qualadd))
                _cb_BQ_FA = max_(_cb_BQ_FA, (f.aDPfr + f.aDPrr + 0.5) / (f.ADPfr[0] + f.ADPrr[0] + 1.0));
        } else {
            _dir_bias_div = (1.0 + (u32)f.gapSa_len / (u32)P.indel_str_repeatsize_max);   // size_t / int -> integer division (main.hpp:4372)
        }
    }
    const double counterbias_P_FA = _cb_P_FA, counterbias_BQ_FA = _cb_BQ_FA, dir_bias_div = _dir_bias_div;
    const i64 aDPgap = nnminus(max_(f.APDP[1], f.APDP[2]), f.aP3);
    const double aDPFAgap = ((rtr1.tracklen + rtr2.tracklen < P.indel_str_repeatsize_max) ? 1.0 : ((f.aP3 + pfa) / (aDPgap + 1.0)));
    const double aDPFA1 = ((aDP + pfa) / (ADP + 1.0));
    const double labelFA = (f.aP2 + 1.5 + f.aP2) / (f.AP2[0] + 2.0 + f.aP2);
    const double aDPFA = min_((is_subst(symbol) ? min_(aDPFA1, max_(aDPFA1 / 3, aDPFAgap)) : aDPFA1), labelFA * (ADP + 1.0) / (f.AP2[0] + 0.5) * unbias_ratio);
    const i32 aDPplus = (is_subst(symbol) ? 0 : ((aDP + 1) * P.bias_prior_DPadd_perc / 100));
    const double dp_coef = ((symbol == UVC_LINK_M) ? max_(P.contam_any_mul_frac, 1.0 - max_(rtr1.tracklen, rtr2.tracklen) / (max_((i64)1, max_(f.ALPL[0], f.ARPL[0])) / max_(1.0 / 150.0, (double)f.ABQ2[0]))) : 1.0);
    double _aPpriorfreq = P.bias_priorfreq_pos, _aBpriorfreq = P.bias_priorfreq_pos;
    const bool is_in_indel_read = ((f.APXM[1]) / 15.0 * P.microadjust_bias_pos_indel_fold * (P.bias_prior_var_DP_mul) > (aDP + aDPplus) * dp_coef);
    const bool is_in_indel_len = (max_(f.APDP[1], f.APDP[2]) * (P.bias_prior_var_DP_mul) > (aDP + aDPplus) * dp_coef);
    const bool is_in_indel_rtr = (max_(f.APDP[3], f.APDP[4]) * (P.bias_prior_var_DP_mul) > (aDP + aDPplus) * dp_coef);
    const bool is_in_rtr = (max_(rtr1.tracklen, rtr2.tracklen) > round(P.indel_polymerase_size));
    const bool is_in_dnv_read = ((UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) && (S.p32(UVC_P_a_dnv_dp, x) * 2 > S.p32(UVC_P_a_snv_dp, x)));
    if (is_in_indel_read || is_in_dnv_read || ((is_ins(symbol) || is_del(symbol)) && (f.APXM[0] > f.APXM[1] * P.microadjust_bias_pos_indel_misma_to_indel_ratio))) {
        _aPpriorfreq -= P.bias_priorfreq_indel_in_read_div; _aBpriorfreq -= P.bias_priorfreq_indel_in_read_div;
    }
    if (UVC_LINK_M != symbol && UVC_LINK_NN != symbol) {
        double maxpf = 0;
        if (is_in_indel_len) maxpf = max_(maxpf, P.bias_priorfreq_indel_in_var_div2);
        if (is_in_indel_rtr) maxpf = max_(maxpf, P.bias_priorfreq_indel_in_str_div2);
        if (is_in_rtr) maxpf = max_(maxpf, P.bias_priorfreq_var_in_str_div2);
        _aBpriorfreq -= maxpf; _aPpriorfreq -= maxpf;
    }
    const double aPpriorfreq = _aPpriorfreq + allbias_allprior, aBpriorfreq = _aBpriorfreq + allbias_allprior;
    fmt.nPF[0] = (i32)round(aPpriorfreq); fmt.nPF[1] = (i32)round(aBpriorfreq);
    const double aIpriorfreq = (is_subst(symbol) ? P.bias_priorfreq_ipos_snv : P.bias_priorfreq_ipos_indel) + allbias_allprior;
    const i32 homopol_len = ((1 == rtr1.unitlen) ? rtr1.tracklen : 0) + ((1 == rtr2.unitlen) ? rtr2.tracklen : 0);
    const double aSBpriorfreq = (is_subst(symbol)
            ? (min_((i32)nnminus(f.aBQ, (((UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) && (homopol_len > 0)) ? min_(5 * homopol_len, 20) : 0)), f.bMQ) + P.bias_priorfreq_strand_snv_base)
            : (P.bias_priorfreq_strand_indel)) + allbias_allprior;
    const double dedup_A2C1_frac = min_(1.0, (double)max_(CDP1, P.bias_reduction_by_high_sequencingDP_min_n_totDepth) / (double)max_(ADP1, 1));
    const double dedup_a2c1_frac = min_(1.0, (double)max_(cDP1, P.bias_reduction_by_high_sequencingDP_min_n_altDepth) / (double)max_(aDP1, 1));
    const double dedup_frag_frac = max_(dedup_A2C1_frac, dedup_a2c1_frac);
    const double pc_read = (is_in_indel_read ? P.bias_FA_pseudocount_indel_in_read : 0.5);
    double r2[2];
    dp4_to_pcFA(r2, false, false, dedup_frag_frac, f.aLP1, aDP, f.ALP2[0] + f.aLP1 - f.aLP2, ADP, P.powlaw_exponent, phred2nat(aPpriorfreq),
                max_((i64)1, f.aLPL) / (double)max_(1, f.aBQ2), max_((i64)1, f.ALPL[0]) / (double)max_((i64)1, f.ABQ2[0]), pc_read);
    double aLPFA = r2[0];
    dp4_to_pcFA(r2, false, false, dedup_frag_frac, f.aRP1, aDP, f.ARP2[0] + f.aRP1 - f.aRP2, ADP, P.powlaw_exponent, phred2nat(aPpriorfreq),
                max_((i64)1, f.aRPL) / (double)max_(1, f.aBQ2), max_((i64)1, f.ARPL[0]) / (double)max_((i64)1, f.ABQ2[0]), pc_read);
    double aRPFA = r2[0];
    dp4_to_pcFA(r2, false, false, dedup_frag_frac, f.aLB1, aDP, f.ALB2[0] + f.aLB1 - f.aLB2, ADP, P.powlaw_exponent, phred2nat(aBpriorfreq),
                max_((i64)1, f.aLBL) / (double)max_(1, f.aBQ2), max_((i64)1, f.ALBL[0]) / (double)max_((i64)1, f.ABQ2[0]), pc_read);
    double aLBFA = r2[0];
    dp4_to_pcFA(r2, false, false, dedup_frag_frac, f.aRB1, aDP, f.ARB2[0] + f.aRB1 - f.aRB2, ADP, P.powlaw_exponent, phred2nat(aBpriorfreq),
                max_((i64)1, f.aRBL) / (double)max_(1, f.aBQ2), max_((i64)1, f.ARBL[0]) / (double)max_((i64)1, f.ABQ2[0]), pc_read);
    double aRBFA = r2[0];
    const bool is_tmore_amplicon = (!tprov ? is_weak_amplicon : is_strong_amplicon);
    const i32 normCDP1 = (f.CDP12b[0] + f.CDP12b[1]) + 1;
    const i32 normBDP = (f.BDPb[0] + f.BDPb[1]) + 1;
    const i32 c2DP = f.cDP2f + f.cDP2r;
    const bool try_tier2 = ((c2DP >= 2) && (normBDP * P.fam_bias_overseq_perc >= normCDP1 * 100) && (S.p32(UVC_P_a_umi_dp, x) * 100 > a_dp * 50));
    fmt.tier2 = (is_rescued ? (fmt.tki_tier2 ? 1 : 0) : (try_tier2 ? 1 : 0));
    // sic: "fmt.c2LP0[0]" and "fmt.c2LP0[a]" are the same element (a = 0) in the MIN() of main.hpp:4477-4478
    const double cFA2L = (fmt.tier2 ? (((double)(((i64)f.c2LP0 * (i64)f.c2LP0) * 2 / max_((i64)1, (i64)min_(c2DP, f.c2LP0 * 4))) + c2altpc) / (f.C2LP0[0] + 1.0)) : 1.0);
    const double cFA2R = (fmt.tier2 ? (((double)(((i64)f.c2RP0 * (i64)f.c2RP0) * 2 / max_((i64)1, (i64)min_(c2DP, f.c2RP0 * 4))) + c2altpc) / (f.C2RP0[0] + 1.0)) : 1.0);
    double c2LPFA = 1.0, c2RPFA = 1.0, c2LBFA = 1.0, c2RBFA = 1.0;
    if (fmt.tier2) {
        const i32 C2DP = f.CDP2b[0] + f.CDP2b[1];
        const double c2Pprior = max_(0.0, aPpriorfreq), c2Bprior = max_(0.0, aBpriorfreq);
        dp4_to_pcFA(r2, false, true, -1, f.c2LP1, c2DP, f.C2LP2[0] + f.c2LP1 - f.c2LP2, C2DP, P.powlaw_exponent, phred2nat(c2Pprior),
                    max_((i64)1, f.c2LPL) / (double)max_(1, f.c2BQ2), max_((i64)1, f.C2LPL[0]) / (double)max_((i64)1, f.C2BQ2[0]), c2altpc, 1.0); c2LPFA = r2[0];
        dp4_to_pcFA(r2, false, true, -1, f.c2RP1, c2DP, f.C2RP2[0] + f.c2RP1 - f.c2RP2, C2DP, P.powlaw_exponent, phred2nat(c2Pprior),
                    max_((i64)1, f.c2RPL) / (double)max_(1, f.c2BQ2), max_((i64)1, f.C2RPL[0]) / (double)max_((i64)1, f.C2BQ2[0]), c2altpc, 1.0); c2RPFA = r2[0];
        dp4_to_pcFA(r2, false, true, -1, f.c2LB1, c2DP, f.C2LB2[0] + f.c2LB1 - f.c2LB2, C2DP, P.powlaw_exponent, phred2nat(c2Bprior),
                    max_((i64)1, f.c2LBL) / (double)max_(1, f.c2BQ2), max_((i64)1, f.C2LBL[0]) / (double)max_((i64)1, f.C2BQ2[0]), c2altpc, 1.0); c2LBFA = r2[0];
        dp4_to_pcFA(r2, false, true, -1, f.c2RB1, c2DP, f.C2RB2[0] + f.c2RB1 - f.c2RB2, C2DP, P.powlaw_exponent, phred2nat(c2Bprior),
                    max_((i64)1, f.c2RBL) / (double)max_(1, f.c2BQ2), max_((i64)1, f.C2RBL[0]) / (double)max_((i64)1, f.C2BQ2[0]), c2altpc, 1.0); c2RBFA = r2[0];
    }
    double _aLIFAx2[2], _aRIFAx2[2];
    {
        double ALpd = (f.ALI2[0] + 0.5) / (f.ADPfr[0] + f.ADPrr[0] - f.ALI2[0] + 0.5);
        double aLpd = (f.aLI1 + ALpd / (1.0 + ALpd)) / (f.aDPfr + f.aDPrr - f.aLI1 + 1.0 / (1.0 + ALpd));
        dp4_to_pcFA(_aLIFAx2, false, false, dedup_frag_frac, f.aLI1, (f.aDPfr + f.aDPrr), (f.ALI2[0] + f.aLI1 - f.aLI2), (f.ADPfr[0] + f.ADPrr[0]),
                    P.powlaw_exponent, phred2nat(aIpriorfreq), aLpd, ALpd, 0.25, 0.5);
    }
    double aLIFA = _aLIFAx2[0] * (is_tmore_amplicon ? dir_bias_div : max_(dir_bias_div, aDPFA / _aLIFAx2[1]));
    {
        double ARpd = (f.ARI2[0] + 0.5) / (f.ADPff[0] + f.ADPrf[0] - f.ARI2[0] + 0.5);
        double aRpd = (f.aRI1 + ARpd / (1.0 + ARpd)) / (f.aDPff + f.aDPrf - f.aRI1 + 1.0 / (1.0 + ARpd));
        dp4_to_pcFA(_aRIFAx2, false, false, dedup_frag_frac, f.aRI1, (f.aDPff + f.aDPrf), (f.ARI2[0] + f.aRI1 - f.aRI2), (f.ADPff[0] + f.ADPrf[0]),
                    P.powlaw_exponent, phred2nat(aIpriorfreq), aRpd, ARpd, 0.25, 0.5);
    }
    double aRIFA = _aRIFAx2[0] * (is_tmore_amplicon ? dir_bias_div : max_(dir_bias_div, aDPFA / _aRIFAx2[1]));
    const double aSIFA = max_((f.aLI1 + 0.5) / (f.ALI2[0] + f.aLI1 - f.aLI2 + 1.0), (f.aRI1 + 0.5) / (f.ARI2[0] + f.aRI1 - f.aRI2 + 1.0));
    const i32 indel_size = f.gapSa_len;
    if (is_ins(symbol) || is_del(symbol)) {
        const double coef = max_(1, f.bDPa) / (double)max_(1, f.bDPf + f.bDPr);
        const bool major_reg = ((max_(f.APDP[1], f.APDP[3]) + max_(f.APDP[2], f.APDP[4])) * 0.5 * (1.0 + (double)FLT_EPSILON) < aDP * coef);
        if ((min_(indel_size, P.microadjust_nobias_pos_indel_maxlen) * aDPFA * coef >= P.nobias_pos_indel_lenfrac_thres) ||
            (max_(rtr1.tracklen, rtr2.tracklen) >= P.nobias_pos_indel_str_track_len && major_reg
             && !(f.APXM[0] > f.APXM[1] * P.microadjust_nobias_pos_indel_misma_to_indel_ratio))) {
            aLPFA += 2.0; aRPFA += 2.0; aLBFA += 2.0; aRBFA += 2.0;
            if (fmt.tier2) { c2LPFA += 2.0; c2RPFA += 2.0; c2LBFA += 2.0; c2RBFA += 2.0; }
        }
        if (f.bMQ >= P.microadjust_nobias_pos_indel_bMQ && f.a2XM2 * 100 >= aDP * 100 * P.microadjust_nobias_pos_indel_perc) { aLIFA += 2.0; aRIFA += 2.0; }
    } else if (UVC_LINK_M == symbol || UVC_LINK_NN == symbol) {
        const double pc = P.bias_FA_pseudocount_indel_in_read;
        aLBFA = min_(aLBFA, (pc + f.aLB1) / (double)(pc * 2 + ADP));
        aRBFA = min_(aRBFA, (pc + f.aRB1) / (double)(pc * 2 + ADP));
    } else if (refsymbol == symbol) {
        aLIFA = aRIFA = max_(aLIFA, aRIFA);
    }
    const i64 avg_sqr_indel_len = max_(f.APXM[4] / max_((i64)1, f.APDP[1]), f.APXM[5] / max_((i64)1, f.APDP[2]));
    if ((!is_subst(symbol)) && ((i64)P.microadjust_nobias_pos_indel_maxlen * P.microadjust_nobias_pos_indel_maxlen < avg_sqr_indel_len)
        && (UVC_LINK_M == symbol || UVC_LINK_NN == symbol || ((i64)(indel_size * 2) * (indel_size * 2) < avg_sqr_indel_len))) {
        const double pc = P.bias_FA_pseudocount_indel_in_read;
        const double aLPFA_minA = (pc + f.aLP1) / (double)(pc * 2 + f.ALP1[0]);
        const double aRPFA_minA = (pc + f.aRP1) / (double)(pc * 2 + f.ALP1[0]);   // sic: ALP1 in both (main.hpp:4575-4576)
        aLPFA = min_(aLPFA, aLPFA_minA); aRPFA = min_(aRPFA, aRPFA_minA);
        if (fmt.tier2) { c2LPFA = min_(c2LPFA, aLPFA_minA); c2RPFA = min_(c2RPFA, aRPFA_minA); }
    }
    if (tprov || (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform)) aLIFA = aRIFA = max_(aLIFA, aRIFA);
    const double aPFFA = (f.aPF1 + pfa * 100.0) / (f.APF2[0] + (f.aPF1 - f.aPF2) + 100.0);
    double aSSFAx2[2];
    dp4_to_pcFA(aSSFAx2, true, false, dedup_frag_frac, f.aRIf, f.aLIr, f.ARIf[0], f.ALIr[0], P.powlaw_exponent, phred2nat(aSBpriorfreq));
    const double ori_base = (is_subst(symbol) ? P.bias_priorfreq_orientation_snv_base : P.bias_priorfreq_orientation_indel_base) + allbias_allprior;
    const double t_ = max_(aDPFA, P.bias_orientation_min_effective_allelefrac);
    const double ori_all = log(t_ * t_) + phred2nat(ori_base);
    double cROFA1x2[2], cROFA2x2[2];
    dp4_to_pcFA(cROFA1x2, true, false, dedup_frag_frac, f.cDP1f, f.cDP1r, f.CDP1b[0], f.CDP1b[1], P.powlaw_exponent, ori_all);
    if (P.bias_is_orientation_artifact_mixed_with_sequencing_error) {
        double c12[2];
        dp4_to_pcFA(c12, true, false, dedup_frag_frac, f.cDP12f, f.cDP12r, f.CDP12b[0], f.CDP12b[1], P.powlaw_exponent, ori_all);
        if ((f.ADPff[0] * 8 >= ADP) && (f.ADPfr[0] * 8 >= ADP) && (f.ADPrf[0] * 8 >= ADP) && (f.ADPrr[0] * 8 >= ADP)) { cROFA1x2[0] = c12[0]; cROFA1x2[1] = c12[1]; }
    }
    dp4_to_pcFA(cROFA2x2, true, true, -1, f.cDP2f, f.cDP2r, f.CDP2b[0], f.CDP2b[1], P.powlaw_exponent, ori_all, -1, -1, c2altpc, 1.0);
    double aSSFA = aSSFAx2[0] * dir_bias_div, cROFA1 = cROFA1x2[0] * dir_bias_div, cROFA2 = cROFA2x2[0] * dir_bias_div;
    if (is_ins(symbol) || is_del(symbol)) { fmt.bAD = min_(fmt.bAD, fmt.bDPa); fmt.AD = min_(fmt.AD, fmt.cDP0a); }
    const double bFA = (f.bDPa + pfa) / ((f.BDPb[0] + f.BDPb[1]) + 1.0);
    const double cFA0 = (f.cDP0a + pfa * (implies_short_frag(f, P.lib_wgs_min_avg_fraglen) ? P.lib_nonwgs_ad_pseudocount : 1.0)) / ((f.CDP1b[0] + f.CDP1b[1]) + 1.0);
    const bool strand_r_weak = ((f.ADPfr[0] + f.ADPrr[0]) * P.microadjust_nobias_strand_all_fold < (f.ADPff[0] + f.ADPrf[0]) * unbias_ratio);
    const bool strand_f_weak = ((f.ADPff[0] + f.ADPrf[0]) * P.microadjust_nobias_strand_all_fold < (f.ADPfr[0] + f.ADPrr[0]) * unbias_ratio);
    if (strand_r_weak) { aLIFA += 4.0; aSSFA += 4.0; }
    if (strand_f_weak) { aRIFA += 4.0; aSSFA += 4.0; }
    const double aLPFA2 = max_(aDPFA * 0.01, aLPFA), aRPFA2 = max_(aDPFA * 0.01, aRPFA), aLBFA2 = max_(aDPFA * 0.01, aLBFA), aRBFA2 = max_(aDPFA * 0.01, aRBFA);
    const double c2LPFA2 = max_(cFA2 * 0.01, c2LPFA), c2RPFA2 = max_(cFA2 * 0.01, c2RPFA), c2LBFA2 = max_(cFA2 * 0.01, c2LBFA), c2RBFA2 = max_(cFA2 * 0.01, c2RBFA);
    const double aLIFA2 = max_(aDPFA * 0.01, aLIFA), aRIFA2 = max_(aDPFA * 0.01, aRIFA), aSSFA2 = max_(aDPFA * 0.05, aSSFA);
    cROFA1 = max_(aDPFA * 1e-4, cROFA1); cROFA2 = max_(aDPFA * 1e-4, cROFA2);
    const double fBTA = (double)((f.BTAb[0] + f.BTAb[1]) + 200), fBTB = (double)((f.BTBb[0] + f.BTBb[1]) + 6);
    const double fbTA = (double)(f.bTAf + f.bTAr + 100), fbTB = (double)(f.bTBf + f.bTBr + 3);
    const double frag_sidelen_frac = 1.0 - min_(
            between_(f.aLIT / max_((i64)1, (i64)(f.aDPfr + f.aDPrr)) - P.microadjust_longfrag_sidelength_min, (i64)0, (i64)P.microadjust_longfrag_sidelength_max),
            between_(f.aRIT / max_((i64)1, (i64)(f.aDPff + f.aDPrf)) - P.microadjust_longfrag_sidelength_min, (i64)0, (i64)P.microadjust_longfrag_sidelength_max))
            / P.microadjust_longfrag_sidelength_zeroMQpenalty;
    const double _alt_frac = fbTB / fbTA;
    const double alt_frac = (is_nmore_amplicon ? (max_(0.0, _alt_frac - 0.2) * 1.25) : _alt_frac);
    const double nonalt_frac = (fBTB + P.contam_any_mul_frac * fbTB - fbTB) / (fBTA + P.contam_any_mul_frac * fbTA - fbTA);
    const double frac_mut = max_(P.syserr_MQ_NMR_expfrac, P.syserr_MQ_NMR_altfrac_coef * alt_frac * frag_sidelen_frac - P.syserr_MQ_NMR_nonaltfrac_coef * nonalt_frac);
    fmt.bNMQ = (i32)round(numstates2phred(pow(frac_mut / P.syserr_MQ_NMR_expfrac, (P.syserr_MQ_NMR_pl_exponent))) * (frac_mut));
    fmt.bNMa = (i32)round(100 * alt_frac); fmt.bNMb = (i32)round(100 * nonalt_frac);
    const bool tmore_primer = (is_tmore_amplicon || ((P.primerlen > 0) && !(0x4 & P.primer_flag)));
    const double bFAa = bFA;
    const double t1only[8] = { cROFA1, aLPFA2, aRPFA2, aLBFA2, aRBFA2, cFA0, aDPFA * between_(1.0 + aDPFA - alt_frac, 0.1, 1.0), aPFFA * aSSFA2 / max_(aSSFA2, aSSFAx2[1]) };
    double t1only_min = t1only[0]; for (int i = 0; i < 8; i++) t1only_min = min_(t1only_min, t1only[i]);
    const double t1plus[5] = { aSSFA2, aLIFA2, aRIFA2, max_(aDPFA * 0.01, aSIFA), bFAa };
    double t1plus_min = t1plus[0]; for (int i = 0; i < 5; i++) t1plus_min = min_(t1plus_min, t1plus[i]);
    const double cFA2a = ((tmore_primer && !is_rescued) ? (cFA2 * (P.powlaw_amplicon_allele_fraction_coef)) : cFA2);
    const double cFA3a = ((normBDP * 100 > normCDP1 * ((P.fam_tier3DP_bias_overseq_perc - 100) / (is_rescued ? 2 : 1) + 100)) ? cFA3 : 1.0);
    const double c23FA = cFA2a;
    const double t2only[9] = { cROFA2, c2LPFA2, c2RPFA2, c2LBFA2, c2RBFA2, cFA2a, cFA3a, cFA2L, cFA2R };
    double t2only_min = t2only[0]; for (int i = 0; i < 9; i++) t2only_min = min_(t2only_min, t2only[i]);
    fmt.nNFA[0] = -numstates2deciphred(counterbias_P_FA); fmt.nNFA[1] = -numstates2deciphred(counterbias_BQ_FA);
    fmt.nNFA[2] = -numstates2deciphred(aDPFA); fmt.nNFA[3] = -numstates2deciphred(bFA); fmt.nNFA[4] = -numstates2deciphred(cFA0); fmt.nNFA[5] = -numstates2deciphred(cFA2);
    fmt.FTS = 0;
    int bit = 0;
    auto push = [&](i32 *vec, int idx, double refFA, double biasFA) {   // fmt_bias_push, main.hpp:4258-4272
        vec[idx] = -numstates2deciphred(biasFA);
        if (biasFA < refFA * P.bias_thres_FTS_FA) fmt.FTS |= (1 << bit);
        fmt.FTSpct[bit] = (i32)round(100.0 * biasFA / refFA);
        bit++;
    };
    push(fmt.nAFA, 0, aDPFA, aSSFA2); push(fmt.nAFA, 1, aDPFA, aPFFA); push(fmt.nAFA, 2, aDPFA, aSIFA);
    push(fmt.nAFA, 3, aDPFA, aLBFA2); push(fmt.nAFA, 4, aDPFA, aRBFA2); push(fmt.nAFA, 5, aDPFA, aLPFA2); push(fmt.nAFA, 6, aDPFA, aRPFA2);
    push(fmt.nAFA, 7, aDPFA, aLIFA2); push(fmt.nAFA, 8, aDPFA, aRIFA2);
    push(fmt.nBCFA, 0, bFA, cFA0); push(fmt.nBCFA, 1, cFA0, bFA); push(fmt.nBCFA, 2, cFA0, cROFA1); push(fmt.nBCFA, 3, cFA2, cROFA2);
    push(fmt.nBCFA, 4, cFA2, c2LPFA2); push(fmt.nBCFA, 5, cFA2, c2RPFA2); push(fmt.nBCFA, 6, cFA2, c2LBFA2); push(fmt.nBCFA, 7, cFA2, c2RBFA2);
    push(fmt.nBCFA, 8, cFA2, cFA2L); push(fmt.nBCFA, 9, cFA2, cFA2R);
    const double aNCFA = ((!tprov && implies_short_frag(f, P.lib_wgs_min_avg_fraglen) && (is_ins(symbol) || is_del(symbol)) && indel_size >= P.lib_nonwgs_clip_penal_min_indelsize)
            ? max_((f.aNC + 0.5) / (ADP + 1.0), between_((f.cDP1f + f.cDP1r) / 300.0, 1.0 / 3.0, 2.0 / 3.0) * aDPFA) : 2.0);
    const double cb_normalgerm = ((!tprov || !implies_short_frag(f, P.lib_wgs_min_avg_fraglen)) ? 1e-9
            : between_(aPFFA * aPFFA * (1.0 / P.lib_nonwgs_normal_full_self_rescue_fa), aPFFA * P.lib_nonwgs_normal_min_self_rescue_fa_ratio, aPFFA));
    const double counterbias_FA = max_(counterbias_P_FA, max_(counterbias_BQ_FA, cb_normalgerm));
    const double dedup_FA = (!tprov ? min_(bFA, cFA0) : max_(bFA, cFA0));
    const double frac_umi2seg = min_(1.0, min_(c23FA / aDPFA, aDPFA / c23FA));
    double refbias = 0;
    if ((is_ins(symbol) || is_del(symbol)) && is_rescued) {   // main.hpp:4804-4810
        const i32 isz = fmt.gapSa_len;
        const i32 indel_noinfo_nbases = (isz * (is_ins(symbol) ? 2 : 1) + max_(isz, max_(rtr1.tracklen, rtr2.anyTR_tracklen)));
        refbias = (double)(indel_noinfo_nbases) / ((double)(min_(f.ALPL[0], f.ARPL[0]) * 2 + indel_noinfo_nbases) / (double)(f.ABQ2[0] + 0.5));
        refbias = min_(refbias, P.microadjust_refbias_indel_max);
    }
    const i32 sumCDP1 = f.CDP1b[0] + f.CDP1b[1], sumCDP2 = f.CDP2b[0] + f.CDP2b[1];
    const double min_abcFA_v = max_(min_(min_(t1plus_min, t1only_min), aNCFA), counterbias_FA);
    fmt.cDP1v = (i32)(norm_fa(min_abcFA_v, refbias) * sumCDP1 * 100);
    const double w6[6] = { aLPFA2, aRPFA2, aLBFA2, aRBFA2, bFA, aNCFA };
    double w6min = w6[0]; for (int i = 0; i < 6; i++) w6min = min_(w6min, w6[i]);
    const double min_abcFA_w = max_(w6min, counterbias_FA);
    fmt.cDP1w = (i32)(norm_fa(min_abcFA_w, refbias) * sumCDP1 * 100);
    double min_abcFA_x = min_(aPFFA, dedup_FA);
    if (tprov) min_abcFA_x = max_(min_abcFA_x, counterbias_FA);
    fmt.cDP1x = 1 + (i32)(min_abcFA_x * sumCDP1 * 100);
    const double cFA2c = cFA2 * cFA2 * cFA2;
    const double c2XBFA2 = between_(3.0 * c2LBFA2 * c2RBFA2 * aSSFA2 / cFA2c, min_(c2LBFA2, c2RBFA2) / 8.0, min_(c2LBFA2, c2RBFA2));
    const double c2XPFA2 = between_(3.0 * c2LPFA2 * c2RPFA2 * aSSFA2 / cFA2c, min_(c2LPFA2, c2RPFA2) / 8.0, min_(c2LPFA2, c2RPFA2));
    const double c2XXFA2 = min_(c2XBFA2, c2XPFA2);
    const double min_c23FA_v = max_(min_(min_(t1plus_min, min_(t2only_min, c2XXFA2)), aNCFA), counterbias_FA * frac_umi2seg);
    fmt.cDP2v = (i32)(norm_fa(min_c23FA_v, refbias) * sumCDP2 * 100);
    const double w7[7] = { c2LPFA2, c2RPFA2, c2XXFA2, c2LBFA2, c2RBFA2, cFA2, aNCFA };
    double w7min = w7[0]; for (int i = 0; i < 7; i++) w7min = min_(w7min, w7[i]);
    const double min_c23FA_w = max_(w7min, counterbias_FA * frac_umi2seg);
    fmt.cDP2w = (i32)(norm_fa(min_c23FA_w, refbias) * sumCDP2 * 100);
    const double min_c23FA_x = min_(aPFFA, c23FA);
    fmt.cDP2x = 1 + (i32)(min_c23FA_x * sumCDP2 * 100);
}

// BcfFormat_symbol_calc_qual, main.hpp:4908-5343; is_rescued = IS_PROVIDED(vcf_tumor_fname) as the caller passes it (main.cpp:979)
static void calc_qual(Fmt &fmt, i32 ins_cdepth, i32 del_cdepth, i32 ins1_cdepth, i32 del1_cdepth, i32 repeatunit_size, i32 repeatnum,
                      const Rtr &rtr1, const Rtr &rtr2, i32 refpos, int refsymbol, State &S) {
    const UvcParams &P = S.P;
    const i64 x = refpos - S.beg;
    const bool tprov = P.tumor_vcf_is_provided;
    const bool is_rescued = tprov;
    const double tpfa = fmt.tpfa_qual;
    const int symbol = fmt.symbol;
    const i32 indel_size = fmt.gapSa_len;
    const i32 sumCDP1 = fmt.CDP1b[0] + fmt.CDP1b[1], sumCDP2 = fmt.CDP2b[0] + fmt.CDP2b[1], sumBDP = fmt.BDPb[0] + fmt.BDPb[1], sumCDP12 = fmt.CDP12b[0] + fmt.CDP12b[1];
    const double cFA2 = (fmt.cDP2f + fmt.cDP2r + 0.5) / (sumCDP2 + 1.0);
    const i32 sscs_phrederr = oracle_sscs_phred(P, refsymbol, symbol) + (!tprov ? 0 : 4);
    const double umi_cFA = (((double)(fmt.cDP2v) + 0.5) / ((double)(sumCDP2 * 100 + 1.0)));
    const double umi_cFA_w = (((double)(fmt.cDP2w) + 0.5) / ((double)(sumCDP2 * 100 + 1.0)));
    // sic: int - double -> truncated into uvc1_qual_t (main.hpp:4951-4954)
    const i32 sscs_inc1 = (i32)(sscs_phrederr - (is_subst(symbol)
            ? (((UVC_BASE_A == refsymbol && UVC_BASE_T == symbol) || (UVC_BASE_T == refsymbol && UVC_BASE_A == symbol))
                ? (double)P.fam_phred_pow_sscs_transversion_AT_TA_origin : P.fam_phred_pow_sscs_snv_origin)
            : P.fam_phred_pow_sscs_indel_origin));
    i32 sscs_inc4tn = (is_subst(symbol)
            ? (i32)(max_(max_(P.fam_phred_sscs_transition_CG_TA, P.fam_phred_sscs_transition_AT_GC), max_(P.fam_phred_sscs_transversion_CG_AT, P.fam_phred_sscs_transversion_other)) - (P.fam_phred_pow_sscs_snv_origin))
            : sscs_inc1);
    const bool is_oxidation = ((UVC_BASE_C == refsymbol && UVC_BASE_A == symbol) || (UVC_BASE_G == refsymbol && UVC_BASE_T == symbol));
    sscs_inc4tn += (is_oxidation ? P.tn_q_inc_max_sscs_CG_AT : P.tn_q_inc_max_sscs_other);
    const double t2n_contam_frac = (tpfa > 0 ? tpfa : 0) * P.contam_t2n_mul_frac;
    const double contamfrac = P.contam_any_mul_frac + (1.0 - P.contam_any_mul_frac) * t2n_contam_frac;
    const i32 aDP = (fmt.aDPff + fmt.aDPfr + fmt.aDPrf + fmt.aDPrr);
    const i32 ADP = (i32)(fmt.ADPff[0] + fmt.ADPrf[0] + fmt.ADPfr[0] + fmt.ADPrr[0]);
    const i32 cDP0 = (fmt.cDP1f + fmt.cDP1r), CDP0 = sumCDP1, cDP2 = (fmt.cDP2f + fmt.cDP2r), CDP2 = sumCDP2;
    const i32 aavgMQ = (i32)(fmt.aMQs / max_(1, aDP));
    const i32 diffAaMQs = (i32)((fmt.AMQs[0] - fmt.aMQs) / max_(1, ADP - aDP)) - aavgMQ;
    const i32 noUMI_bias_inc = min_(P.bias_FA_powerlaw_noUMI_phred_inc_snv, aDP / 2);
    const double pl_noUMI_phred_inc = P.powlaw_anyvar_base + (is_subst(symbol) ? noUMI_bias_inc : P.bias_FA_powerlaw_noUMI_phred_inc_indel);
    const i32 withUMI_bias_inc = min_(P.bias_FA_powerlaw_withUMI_phred_inc_snv - P.bias_FA_powerlaw_noUMI_phred_inc_snv, cDP2 / 2) + noUMI_bias_inc;
    const double pl_withUMI_phred_inc = P.powlaw_anyvar_base + (is_subst(symbol) ? withUMI_bias_inc : P.bias_FA_powerlaw_withUMI_phred_inc_indel);
    const double prior_weight = 1.0 / (fmt.cDPmf + fmt.cDPmr + 1.0);
    const i32 fam_thres_highBQ = (is_subst(symbol) ? P.fam_thres_highBQ_snv : P.fam_thres_highBQ_indel);
    const i32 cMmQ = (i32)round(numstates2phred((fmt.cDPMf + fmt.cDPmf + fmt.cDPMr + fmt.cDPmr + pow(10, fam_thres_highBQ / 10.0) * prior_weight) / (fmt.cDPmf + fmt.cDPmr + prior_weight)));
    const i32 nbases_x100_1 = fmt.bIADb * 100 + 1;
    const i32 nbases_x100_2 = min_(nbases_x100_1, fmt.cDP1v + 1);
    const i64 perbase_q_x10_1 = 10 * fmt.bIAQb / max_(1, fmt.bIADb);
    const i64 perbase_q_x10_2 = perbase_q_x10_1 + (i64)round(10 * numstates2phred((double)nbases_x100_2 / (double)nbases_x100_1));
    i64 duped_frag_binom_qual = ((is_ins(symbol) || is_del(symbol)) ? perbase_q_x10_1 : perbase_q_x10_2) * nbases_x100_2 / (10 * 100);
    const i64 contam_frag_withmin_qual = (i64)round(calc_binom_10log10_likeratio(t2n_contam_frac, cDP0, CDP0 - cDP0)) + 9 - 3;
    const i32 het3al_snp = max_(0, 2 * P.germ_phred_hetero_snp - P.germ_phred_het3al_snp - 0);
    const i32 het3al_indel = max_(0, 2 * P.germ_phred_hetero_indel - P.germ_phred_het3al_indel - 0);
    i32 het3al_inc = (is_subst(symbol) ? het3al_snp : het3al_indel);
    if (is_ins(symbol) || is_del(symbol)) het3al_inc = (i32)nnminus(het3al_indel + 1, indel_size);
    const i32 contam_syserr_phred_bypassed = het3al_inc;
    const i32 normcDP1 = (fmt.cDP12f + fmt.cDP12r + 1), normCDP1 = sumCDP12 + 1, normBDP = sumBDP + 1;
    const i32 sscs_dec1_div = (is_rescued ? 2 : 1);
    const i64 sscs_dec1a = (((P.fam_min_n_copies / sscs_dec1_div <= normCDP1) || (P.fam_min_n_copies_DPxAD / sscs_dec1_div <= (i64)normCDP1 * normcDP1)) ? 0 : (sscs_inc1 + 3));
    const i64 sscs_dec1b = (((i64)((P.fam_min_overseq_perc - 100) / sscs_dec1_div + 100) * normCDP1 <= (i64)100 * normBDP) ? 0 : (sscs_inc1 + 3));
    const i64 sscs_dec1 = max_(sscs_dec1a, sscs_dec1b);
    const i64 sscs_dec2 = nnminus(fam_thres_highBQ, cMmQ);
    const i64 cIADnormcnt = (i64)(fmt.cIADf + fmt.cIADr) * 100 + 1;
    const i64 cIADmincnt = min_(cIADnormcnt, (i64)fmt.cDP2v + 1);
    const i64 sscs_bq_fw = fmt.cIAQf + ((i64)fmt.cIAQr * min_(P.fam_phred_dscs_all - fmt.cIDQf, fmt.cIDQr)) / max_(fmt.cIDQr, 1);
    const i64 sscs_bq_rv = fmt.cIAQr + ((i64)fmt.cIAQf * min_(P.fam_phred_dscs_all - fmt.cIDQr, fmt.cIDQf)) / max_(fmt.cIDQf, 1);
    const i64 contam_sscs_withmin_qual = (i64)round(calc_binom_10log10_likeratio(t2n_contam_frac, cDP2, CDP2 - cDP2)) + 9 - 3;
    // int64mul(non_neg_minus(int64, double) -> double, int64): the double is converted to int64 by int64mul's cast (main.hpp:5052-5055)
    i64 sscs_binom_qual = ((i64)nnminus_d((double)max_(sscs_bq_fw, sscs_bq_rv), numstates2phred(cIADnormcnt / (double)cIADmincnt) * cIADnormcnt / 100.0) * cIADmincnt) / (cIADnormcnt);
    if (max_(sscs_bq_fw, sscs_bq_rv) > P.microadjust_fam_binom_qual_halving_thres && is_subst(symbol))
        sscs_binom_qual = min_(sscs_binom_qual, P.microadjust_fam_binom_qual_halving_thres + (max_(sscs_bq_fw, sscs_bq_rv) - P.microadjust_fam_binom_qual_halving_thres) / 2);
    sscs_binom_qual -= sscs_dec1 + sscs_dec2;
    const double min_bcFA_v = (((double)(fmt.cDP1v) + 0.5) / (double)(sumCDP1 * 100 + 1.0));
    i32 dedup_powlaw_qual_v = (i32)round(P.powlaw_exponent * numstates2phred(min_bcFA_v) + (pl_noUMI_phred_inc));
    const double min_bcFA_w = (((double)(fmt.cDP1w) + 0.5) / (double)(sumCDP1 * 100 + 1.0));
    i32 dedup_powlaw_qual_w = (i32)round(P.powlaw_exponent * numstates2phred(min_bcFA_w) + (pl_noUMI_phred_inc) + P.tn_q_inc_max);
    const i32 ds_vq_inc_powlaw = (i32)round(10 / log(10) * min_(log((fmt.cDP12f + 0.5) / (fmt.CDP12b[0] + 1.0)), log((fmt.cDP12r + 0.5) / (fmt.CDP12b[1] + 1.0)))) + (sscs_phrederr);
    const i32 ds_vq_inc_binom = 3 * min_(fmt.cDP2f, fmt.cDP2r);
    const i64 m5 = min_(min_(sscs_bq_fw, sscs_bq_rv), (i64)min_(ds_vq_inc_powlaw, min_(ds_vq_inc_binom, 3)));
    const i32 sscs_inc2 = (i32)max_((i64)0, m5) * ((cFA2 > 0.002) ? 1 : 0);
    const i32 sscs_dec3 = (is_rescued ? (-3) : ((cFA2 >= 0.003) ? 0 : 5));
    // uvc1_qual_t = double + ints: truncation toward zero on assignment (main.hpp:5079-5080)
    const i32 sscs_base_2 = (i32)(pl_withUMI_phred_inc + sscs_inc1 + sscs_inc2 - sscs_dec1 - sscs_dec2 - sscs_dec3);
    const i32 sscs_base_2tn = (i32)(pl_withUMI_phred_inc + sscs_inc4tn + sscs_inc2 - sscs_dec1 - sscs_dec2 - sscs_dec3);
    i32 sscs_powlaw_qual_v = (i32)round((P.powlaw_exponent * numstates2phred(umi_cFA) + sscs_base_2));
    i32 sscs_powlaw_qual_w = (i32)round((P.powlaw_exponent * numstates2phred(umi_cFA_w) + sscs_base_2tn));
    const double dFA = (double)(fmt.dDP2 + 0.5) / (double)(fmt.DDP1[0] + 1.0);
    const double dSNR = (double)(fmt.dDP2 + 0.5) / (double)(fmt.dDP1 + 1.0);
    const double dnormFA = dFA * pow(dSNR, 1.0 / P.powlaw_exponent);
    const i64 dscs_est = (i64)round((P.fam_phred_dscs_max + sscs_phrederr) / 2.0);
    const i64 dFA_vq_binom = (dscs_est - (i64)round(numstates2phred(1.0 / (dnormFA)))) * (i64)fmt.dDP2 * (i64)cIADmincnt / (i64)cIADnormcnt;
    const i32 dFA_vq_powlaw = (i32)(P.powlaw_anyvar_base + (dscs_est - P.fam_phred_pow_dscs_all_origin)
            + (i32)round(numstates2phred((dnormFA) * min_(1.0, (double)((fmt.cDP1v) + 0.5) / (double)(sumCDP1 * 100 + 1.0)))));
    fmt.cMmQ = cMmQ;
    const double eps = (double)FLT_EPSILON;
    const bool is_indel_penal_applied = ((UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) && !tprov);
    const i32 indel_penal_base = (is_indel_penal_applied
            ? ((i32)round(P.indel_multiallele_samepos_penal / log(2) * log((double)max_(aDP + eps, (double)max_(fmt.APDP[1], fmt.APDP[2])) / (double)(aDP + eps)))) : 0);
    i32 penal4multi = 0, penal4multi_g = 0, penal4multi_soma = 0, indel_UMI_penal = 0;
    if (indel_size > 0 && fmt.cDP0a > 0) {
        const double indel_pq = (double)min_(indel_phred(P.indel_polymerase_slip_rate, repeatunit_size, repeatnum), 24) + 2 - (double)10;
        const i32 eff_tracklen1 = (repeatunit_size * max_(1, repeatnum) - repeatunit_size);
        const i32 eff_tracklen2 = (max_(rtr1.tracklen - rtr1.unitlen, rtr2.tracklen - rtr2.unitlen) / 3);
        const double indel_ic = numstates2phred((double)max_((size_t)(indel_size + (is_ins(symbol) ? 1 : 0)), (size_t)1) / (double)(max_(eff_tracklen1, eff_tracklen2) + 1))
                + (is_ins(symbol) ? (numstates2phred(P.indel_del_to_ins_err_ratio) * min_(200, fmt.cDP0a) / 200) : 0);
        double indelcdepth;
        {
            // indelcdepth is `auto` = uvc1_readnum_t; "indelcdepth += del1_cdepth / ratio" converts back to int (main.hpp:5131-5137)
            i32 ic = (is_ins(symbol) ? ins_cdepth : del_cdepth);
            if (UVC_LINK_D1 == symbol) ic += ins1_cdepth;
            if (UVC_LINK_I1 == symbol) ic = (i32)(ic + del1_cdepth / P.indel_del_to_ins_err_ratio);
            indelcdepth = ic;
        }
        const i32 nearInDelDP = (i32)(is_ins(symbol) ? fmt.APDP[1] : fmt.APDP[2]);
        i32 penal1 = (i32)round(P.indel_multiallele_samepos_penal / log(2.0) * log((double)(indelcdepth + eps) / (double)(fmt.cDP0a + eps)));
        if (UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) penal1 = (i32)nnminus_d(penal1, P.indel_multiallele_samepos_penal);
        const i32 penal2 = (i32)round(P.indel_multiallele_diffpos_penal / log(2.0) * log((double)(nearInDelDP + eps) / (double)(max_(aDP, nearInDelDP) + eps)));
        penal4multi_g = (i32)((i32)round(P.indel_tetraallele_germline_penal_value / log(2.0) * log((double)(ins_cdepth + del_cdepth + eps) / (double)(fmt.cDP0a + eps))) - P.indel_tetraallele_germline_penal_thres);
        if (is_ins(symbol)) {
            penal4multi = (penal1 * P.indel_ins_penal_pseudocount / (i32)(P.indel_ins_penal_pseudocount + indel_size));
            penal4multi_soma = penal4multi;
        } else { penal4multi = max_(penal1, penal2); penal4multi_soma = penal1; }
        dedup_powlaw_qual_v += (i32)round(indel_ic);
        dedup_powlaw_qual_w += (i32)round(indel_ic);
        duped_frag_binom_qual += (i64)round(indel_pq);
        const i64 sz = max_((u32)indel_size, 1U);
        const double sscs_indel_ic = numstates2phred((double)(sz * sz) / (double)(max_(eff_tracklen1, eff_tracklen2) + 1));
        const i32 ins_vs_del_inc = (i32)round(P.powlaw_exponent * numstates2phred(P.indel_del_to_ins_err_ratio));
        // double - int, truncated once on assignment to uvc1_qual_t (main.hpp:5170-5173)
        const i32 extra_reward = (i32)(nnminus_d(ins_vs_del_inc, sscs_indel_ic * (is_ins(symbol) ? 0 : max_(eff_tracklen1, eff_tracklen2)) / round(P.indel_polymerase_size)) - (double)(ins_vs_del_inc / 2));
        sscs_powlaw_qual_v += (i32)round(sscs_indel_ic) + extra_reward;
        sscs_powlaw_qual_w += (i32)round(sscs_indel_ic) + extra_reward;
        sscs_binom_qual += (i64)round(indel_pq) + extra_reward;
        if (fmt.tier2) indel_UMI_penal = (i32)nnminus_d((sumBDP + 1.0) / (double)(sumCDP1 + 1.0) * P.fam_indel_nonUMI_phred_dec_per_fold_overseq,
                                                        (P.fam_thres_emperr_all_flat_indel + 1) * P.fam_indel_nonUMI_phred_dec_per_fold_overseq);
    }
    if (is_oxidation && tprov) sscs_binom_qual = max_(sscs_binom_qual, (i64)min_(aDP, 3));
    fmt.aAaMQ = diffAaMQs;
    const i32 readlenMQcap = (i32)((fmt.APXM[2]) / max_((i64)1, fmt.APDP[0]) - 17);
    const i32 diffMQ = max_(0, diffAaMQs);
    const bool aln_extra_accurate = (P.inferred_maxMQ > 60);
    const i32 MQVQadd = (i32)((symbol == refsymbol) ? 0 : (min_(P.germ_phred_homalt_snp, ADP * 3)));
    const i32 MQVQadd_somatic = (i32)((symbol != refsymbol) ? 0 : (min_(P.germ_phred_homalt_snp, ADP * 3)));
    const bool MQ_unadjusted = (aln_extra_accurate || (!is_subst(symbol)) || (aDP > ADP * 3 / 4));
    const i32 MQVQminus = (MQ_unadjusted ? 0 : ((i32)nnminus((60 - 30), aavgMQ) * 2 / 5)) + ((MQ_unadjusted || (refsymbol != symbol)) ? 0 : (i32)nnminus(min_(15, diffMQ), aavgMQ));
    i32 diffMQ2 = diffMQ;
    if (fmt.bMQ < 20 && !tprov) {
        const double aDPxf = (fmt.aDPff + fmt.aDPrf + 0.5), aDPxr = (fmt.aDPfr + fmt.aDPrr + 0.5);
        const double ADPxf = (fmt.ADPff[0] + fmt.ADPrf[0] + 1.0), ADPxr = (fmt.ADPfr[0] + fmt.ADPrr[0] + 1.0);
        if ((aDPxr / ADPxr) * 2 < (aDPxf / ADPxf) || (aDPxf / ADPxf) * 2 < (aDPxr / ADPxr)
            || (fmt.aLI1 + 0.5) / (fmt.ALI2[0] + 1.0) * (2 * (1.0 + DBL_EPSILON)) < (aDPxr) / (ADPxr)
            || (fmt.aRI1 + 0.5) / (fmt.ARI2[0] + 1.0) * (2 * (1.0 + DBL_EPSILON)) < (aDPxf) / (ADPxf)) diffMQ2 = max_(diffMQ2, 20 - min_(fmt.bMQ, 20));
    }
    // `auto` = double here: int * double / int + double - int - int (main.hpp:5215-5217)
    const double MQ_base = ((fmt.bMQ * (P.syserr_MQ_max - P.syserr_MQ_nonref_base) / P.syserr_MQ_max + P.syserr_MQ_nonref_base)) - (i32)(diffMQ2) - (i32)(fmt.bNMQ);
    const i32 systematicMQ = (((refsymbol == symbol) && (ADP > aDP * 2)) ? fmt.bMQ : (i32)(MQ_base - (i32)(numstates2phred((ADP + 1.0) / (aDP + 0.5)))));
    const bool is_nonWGS = implies_short_frag(fmt, P.lib_wgs_min_avg_fraglen);
    const i32 normal_rescued_MQ = min_((i32)nnminus(readlenMQcap, 60), (is_nonWGS ? P.lib_nonwgs_normal_max_rescued_MQ : P.lib_wgs_normal_max_rescued_MQ));
    i32 systematicMQVQ1 = min_((max_(systematicMQ, P.syserr_MQ_min) + MQVQadd), readlenMQcap);
    const i32 systematicBQVQ = (((UVC_PLATFORM_IONTORRENT != P.inferred_sequencing_platform) && is_subst(symbol)) ? fmt.aBQQ : 200);
    const i32 pcr_dp = S.p32(UVC_P_a_pcr_dp, x);
    const bool strong_amp = ((pcr_dp * 100) > fmt.APDP[0] * 50), weak_amp = ((pcr_dp * 100) > fmt.APDP[0] * 30);
    const bool tmore_amp = (!tprov ? weak_amp : strong_amp);
    if (tmore_amp && (is_ins(symbol) || is_del(symbol)) && (systematicMQVQ1 > 70) && (fmt.APXM[1] / max_(fmt.APDP[0], (i64)1) > 20))
        systematicMQVQ1 = (i32)(70 + ((systematicMQVQ1 - 70) * 5 / (fmt.APXM[1] / max_(fmt.APDP[0], (i64)1) - 15)));
    i32 indel_penal_base_add = 0;
    if (!tprov) {
        const i64 delAPDP = max_(fmt.APDP[2], fmt.APDP[4]);
        const i32 snv_dp = S.p32(UVC_P_a_snv_dp, x);
        if ((fmt.APDP[0] < 3 * delAPDP) && (fmt.APDP[0] < 3 * snv_dp) && (aDP * 3 < delAPDP) && (aDP * 3 < snv_dp) && is_subst(symbol) && (rtr2.tracklen >= 8 * rtr2.unitlen))
            indel_penal_base_add = P.microadjust_germline_mix_with_del_snv_penalty;
        if (tmore_amp && is_del(symbol)) {
            if (aDP * 4 < fmt.APDP[2]) indel_penal_base_add = max_(indel_penal_base_add, 5);
            else if (fmt.cDP0a * 3 < 2 * (del_cdepth)) indel_penal_base_add = max_(indel_penal_base_add, 2);
        }
    }
    const i32 systematicMQVQ = max_(0, systematicMQVQ1);
    const i32 indel_penal_base2 = indel_penal_base + indel_penal_base_add;
    const i64 fx = fmt.ADPff[0] + fmt.ADPfr[0], rx = fmt.ADPrf[0] + fmt.ADPrr[0], xf = fmt.ADPff[0] + fmt.ADPrf[0], xr = fmt.ADPfr[0] + fmt.ADPrr[0];
    const bool frx_imba = (max_(fx, rx) > P.microadjust_strand_orientation_absence_DP_fold * (min_(fx, rx) + 1));
    const bool xfr_imba = (max_(xf, xr) > P.microadjust_strand_orientation_absence_DP_fold * (min_(xf, xr) + 1));
    const i32 v_minus = (is_subst(symbol) ? ((frx_imba ? P.microadjust_orientation_absence_snv_penalty : 0) + (xfr_imba ? P.microadjust_strand_absence_snv_penalty : 0))
                                          : (tmore_amp ? P.microadjust_dedup_absence_indel_penalty : 0));
    const i32 tn_syserr_q = systematicMQVQ + P.tn_q_inc_max + normal_rescued_MQ;
    fmt.bMQQ = systematicMQVQ;
    fmt.bIAQ = (i32)(duped_frag_binom_qual - indel_penal_base2);
    fmt.cIAQ = (i32)(sscs_binom_qual - indel_penal_base);
    fmt.cPCQ1 = min_(dedup_powlaw_qual_w - indel_penal_base2, tn_syserr_q);
    fmt.cPLQ1 = dedup_powlaw_qual_v - indel_penal_base2 - v_minus;
    fmt.cPCQ2 = min_(sscs_powlaw_qual_w - indel_penal_base, tn_syserr_q);
    fmt.cPLQ2 = sscs_powlaw_qual_v - indel_penal_base;
    fmt.bTINQ = (i32)(contam_frag_withmin_qual + contam_syserr_phred_bypassed);
    fmt.cTINQ = (i32)(contam_sscs_withmin_qual + contam_syserr_phred_bypassed);
    const i32 aDPpc = ((refsymbol == symbol) ? 1 : 0);
    const i64 d_ = max_(1, aDP + aDPpc);
    const i32 penal4BQerr = (is_subst(symbol) ? (5 + (i32)(((i64)P.penal4lowdep) / (d_ * d_))) : 0);
    const i32 indel_q_inc = ((((!is_ins(symbol)) && (!is_del(symbol))) || is_rescued) ? 0 : indel_len_rusize_phred(indel_size, repeatnum));
    // MAX3(0, int - double, int - double) is evaluated in double (main.hpp:5298-5301)
    const double m3 = max_(0.0, max_(penal4multi - P.indel_multiallele_soma_penal_thres, (double)penal4multi_g));
    fmt.gVQ1 = (i32)max_(0.0, indel_q_inc + min_(min_(systematicBQVQ, (i32)nnminus(systematicMQVQ, MQVQminus)), min_(fmt.bIAQ - penal4BQerr, fmt.cPLQ1)) - 2 * m3);
    const i32 somatic_minus = (is_rescued ? 0 : (15 - min_(ADP * 15 / 100, min_(aDP, 15))));
    const i32 systematicVQsomatic = (i32)nnminus(min_(systematicBQVQ, systematicMQVQ + MQVQadd_somatic), somatic_minus);
    const i32 bcVQ1 = min_(systematicVQsomatic, min_(fmt.bIAQ - (is_rescued ? 0 : penal4BQerr), fmt.cPLQ1)) - penal4multi_soma;
    fmt.cVQ1 = max_(0, min_(bcVQ1, fmt.bTINQ) - indel_UMI_penal);
    i32 mincVQ2 = 0;
    if (is_ins(symbol) || is_del(symbol)) {
        const i32 floor_v = (i32)(min_(P.germ_phred_homalt_indel + numstates2phred(umi_cFA), (double)(fmt.cDP2v * 3 / 100)) + (double)(((is_ins(symbol) ? 1 : 0) - 1) * 3));
        mincVQ2 = max_(mincVQ2, floor_v);
    }
    const i64 dVQinc = min_(min_(dFA_vq_binom, (i64)dFA_vq_powlaw) - max_(0, min_(fmt.cIAQ, fmt.cPLQ2)), (i64)P.fam_phred_dscs_inc_max);
    fmt.dVQinc = (i32)dVQinc;
    const i32 cVQ2 = (i32)min_((i64)systematicVQsomatic, min_(fmt.cIAQ + max_((i64)0, dVQinc), fmt.cPLQ2 + max_((i64)0, dVQinc))) - penal4multi;
    fmt.cVQ2 = max_(mincVQ2, min_(cVQ2, fmt.cTINQ));
    const i32 cDP1y = (is_rescued ? fmt.cDP1x : fmt.cDP1v);
    const i32 CDP1y0 = (is_rescued ? fmt.CDPv[2][0] : fmt.CDPv[0][0]);
    const double binom_contam_LODQ = calc_binom_10log10_likeratio(contamfrac, cDP1y, CDP1y0);
    const double power_contam_LODQ = round(10.0 / log(10.0) * P.powlaw_exponent * max_(logit2((cDP1y + 1) / (double)(CDP1y0 + 1), contamfrac), 0.0));
    fmt.CONTQ = (i32)min_(binom_contam_LODQ, power_contam_LODQ);
}

static void emit(const Fmt &f, std::vector<i32> &r) {
    r.assign(UVC_NUM_SCORE_FIELDS, 0);
    r[UVC_O_refpos] = f.refpos; r[UVC_O_symbol] = f.symbol; r[UVC_O_refsymbol] = f.refsymbol;
    r[UVC_O_DP] = f.DP; r[UVC_O_AD] = f.AD; r[UVC_O_bDP] = f.bDP; r[UVC_O_bAD] = f.bAD; r[UVC_O_c2DP] = f.c2DP; r[UVC_O_c2AD] = f.c2AD; r[UVC_O_bDPa] = f.bDPa; r[UVC_O_cDP0a] = f.cDP0a;
    r[UVC_O_a2BQf] = f.a2BQf; r[UVC_O_a2BQr] = f.a2BQr; r[UVC_O_aBQ] = f.aBQ; r[UVC_O_aBQQ] = f.aBQQ; r[UVC_O_bMQ] = f.bMQ;
    r[UVC_O_nPF0] = f.nPF[0]; r[UVC_O_nPF1] = f.nPF[1]; r[UVC_O_bNMa] = f.bNMa; r[UVC_O_bNMb] = f.bNMb; r[UVC_O_bNMQ] = f.bNMQ;
    for (int i = 0; i < 6; i++) r[UVC_O_nNFA0 + i] = f.nNFA[i];
    for (int i = 0; i < 9; i++) r[UVC_O_nAFA0 + i] = f.nAFA[i];
    for (int i = 0; i < 10; i++) r[UVC_O_nBCFA0 + i] = f.nBCFA[i];
    r[UVC_O_FTS] = f.FTS; r[UVC_O_tier2] = f.tier2;
    r[UVC_O_cDP1v] = f.cDP1v; r[UVC_O_cDP1w] = f.cDP1w; r[UVC_O_cDP1x] = f.cDP1x; r[UVC_O_cDP2v] = f.cDP2v; r[UVC_O_cDP2w] = f.cDP2w; r[UVC_O_cDP2x] = f.cDP2x;
    for (int i = 0; i < 6; i++) { r[UVC_O_CDP1v0 + 2 * i] = f.CDPv[i][0]; r[UVC_O_CDP1v0 + 2 * i + 1] = f.CDPv[i][1]; }
    r[UVC_O_cMmQ] = f.cMmQ; r[UVC_O_aAaMQ] = f.aAaMQ; r[UVC_O_bMQQ] = f.bMQQ; r[UVC_O_bIAQ] = f.bIAQ; r[UVC_O_cIAQ] = f.cIAQ;
    r[UVC_O_cPCQ1] = f.cPCQ1; r[UVC_O_cPLQ1] = f.cPLQ1; r[UVC_O_cPCQ2] = f.cPCQ2; r[UVC_O_cPLQ2] = f.cPLQ2; r[UVC_O_bTINQ] = f.bTINQ; r[UVC_O_cTINQ] = f.cTINQ;
    r[UVC_O_gVQ1] = f.gVQ1; r[UVC_O_cVQ1] = f.cVQ1; r[UVC_O_dVQinc] = f.dVQinc; r[UVC_O_cVQ2] = f.cVQ2; r[UVC_O_CONTQ] = f.CONTQ;
    r[UVC_O_gapSa] = f.gapSa_row; r[UVC_O_gapSa_len] = f.gapSa_len; r[UVC_O_tkey] = f.tkey_idx;
    for (int i = 0; i < 2; i++) { r[UVC_O_cVQ1M0 + i] = f.cVQ1M[i]; r[UVC_O_cVQ2M0 + i] = f.cVQ2M[i]; r[UVC_O_cVQAM0 + i] = f.cVQAM[i]; r[UVC_O_cVQSM0 + i] = f.cVQSM[i]; r[UVC_O_vAC0 + i] = f.vAC[i]; }
    r[UVC_O_vNLODQ] = f.vNLODQ;
    for (int i = 0; i < 4; i++) { r[UVC_O_GL4_0 + i] = f.GL4[i]; r[UVC_O_TNBQF0 + i] = f.TNBQF[i]; r[UVC_O_TNCQF0 + i] = f.TNCQF[i]; }
    for (int i = 0; i < 8; i++) r[UVC_O_GST0 + i] = f.GST[i];
    r[UVC_O_germ_GT] = f.germ_GT; r[UVC_O_germ_GQ] = f.germ_GQ; r[UVC_O_germ_emit] = f.germ_emit; r[UVC_O_germ_ref] = f.germ_ref; r[UVC_O_germ_alt1] = f.germ_alt1; r[UVC_O_germ_alt2] = f.germ_alt2;
    r[UVC_O_out] = f.out; r[UVC_O_vHGQ] = f.vHGQ; r[UVC_O_NLODQ] = f.NLODQ; r[UVC_O_NLODV] = f.NLODV; r[UVC_O_TLODQ] = f.TLODQ; r[UVC_O_SomaticQ] = f.SomaticQ;
    r[UVC_O_QUAL] = f.QUALbits; r[UVC_O_FILTER] = f.FILTER; r[UVC_O_keep] = f.keep;
    for (int b = 0; b < 19; b++) if (f.FTS & (1 << b)) r[UVC_O_FTSpct0 + b / 4] |= (i32)((u32)min_(max_(f.FTSpct[b], 0), 255) << (8 * (b % 4)));
}

// ------------------------------------------------------------------------------------------------
// InDel alleles: fill_by_indel_info (instcode.hpp, main.hpp:5350-5376) and indel_get_majority (main.hpp:5406-5455)
// ------------------------------------------------------------------------------------------------
static inline int ins_idx_(int s) { return (UVC_LINK_I1 == s ? 0 : ((UVC_LINK_I2 == s) ? 1 : 2)); }
static inline int del_idx_(int s) { return (UVC_LINK_D1 == s ? 0 : ((UVC_LINK_D2 == s) ? 1 : 2)); }
struct GapTuple { i32 fq, bq, c2, c2d; std::string s; };
static bool gap_tuple_less(const GapTuple &a, const GapTuple &b) {
    if (a.fq != b.fq) return a.fq < b.fq;
    if (a.bq != b.bq) return a.bq < b.bq;
    if (a.c2 != b.c2) return a.c2 < b.c2;
    if (a.c2d != b.c2d) return a.c2d < b.c2d;
    return a.s < b.s;
}
template <class K> static i32 gap_get(const std::map<i32, std::map<K, i32>> &m, i32 refpos, const K &k) {   // posToIndelToData_get, main.hpp:65-71
    auto it = m.find(refpos);
    if (it == m.end()) return 0;
    auto jt = it->second.find(k);
    return (jt == it->second.end() ? 0 : jt->second);
}
// the tuples one strand pushes into gapSeq / gapbAD1 / gapcAD1 / gc2AD / gc2dAD, in the pushed order (instcode.hpp:44-83);
// insertion strings are "ACGTN" text, deletion strings the deleted reference characters
static void gap_tuples(State &S, int strand, i32 refpos, int symbol, std::vector<GapTuple> &out) {
    out.clear();
    if (is_ins(symbol)) {
        const int k = ins_idx_(symbol);
        auto it = S.gap_frag[strand].iseq[k].find(refpos);
        if (it == S.gap_frag[strand].iseq[k].end()) return;
        for (const auto &kv : it->second) {
            if (kv.first.empty()) continue;
            GapTuple t = { gap_get(S.gap_fam[strand].iseq[k], refpos, kv.first), kv.second, gap_get(S.gap_c2[strand].iseq[k], refpos, kv.first), gap_get(S.gap_c2d[strand].iseq[k], refpos, kv.first), kv.first };
            out.push_back(t);
        }
    } else {
        const int k = del_idx_(symbol);
        auto it = S.gap_frag[strand].dlen[k].find(refpos);
        if (it == S.gap_frag[strand].dlen[k].end()) return;
        for (const auto &kv : it->second) {
            const std::string str = S.refstring.substr((size_t)(refpos - S.beg), (size_t)kv.first);
            if (str.empty()) continue;
            GapTuple t = { gap_get(S.gap_fam[strand].dlen[k], refpos, kv.first), kv.second, gap_get(S.gap_c2[strand].dlen[k], refpos, kv.first), gap_get(S.gap_c2d[strand].dlen[k], refpos, kv.first), str };
            out.push_back(t);
        }
    }
    std::sort(out.rbegin(), out.rend(), gap_tuple_less);
}

void indel_allele_rows(State &S, std::vector<UvcGapRow> &rows, std::vector<u8> &seq) {
    rows.clear(); seq.clear();
    std::map<std::pair<i32, int>, int> keys;   // (refpos, symbol) that have a row on either strand
    for (int strand = 0; strand < 2; strand++) for (int k = 0; k < 3; k++) {
        const int isym[3] = { UVC_LINK_I1, UVC_LINK_I2, UVC_LINK_I3P }, dsym[3] = { UVC_LINK_D1, UVC_LINK_D2, UVC_LINK_D3P };
        for (const auto &pv : S.gap_frag[strand].iseq[k]) keys[std::make_pair(pv.first, isym[k])] = 1;
        for (const auto &pv : S.gap_frag[strand].dlen[k]) keys[std::make_pair(pv.first, dsym[k])] = 1;
    }
    std::vector<GapTuple> t;
    for (const auto &kv : keys) for (int strand = 0; strand < 2; strand++) {
        gap_tuples(S, strand, kv.first.first, kv.first.second, t);
        for (const GapTuple &g : t) {
            UvcGapRow r; memset(&r, 0, sizeof(r));
            r.refpos = kv.first.first; r.symbol = kv.first.second; r.strand = strand; r.len = (i32)g.s.size();
            r.seq_off = -1;
            if (is_ins(r.symbol)) { r.seq_off = (i64)seq.size(); for (char c : g.s) seq.push_back((u8)(c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4)); }
            r.bAD1 = g.bq; r.cAD1 = g.fq; r.c2AD = g.c2; r.c2dAD = g.c2d;
            rows.push_back(r);
        }
    }
}

struct GapAllele { i32 bAD1, cAD1; std::string s; };
static const char *SYMBOL_DESC[NSYM] = { "A", "C", "G", "T", "N", "*", "<LR>", "<LD3P>", "<LD2>", "<LD1>", "<LI3P>", "<LI2>", "<LI1>", "*" };   // main_conversion.hpp:336-346
// The alleles the reference scores for one InDel symbol (main.cpp:853-896): strands with FRAG_bDP > 0 contribute their tuples, equal
// strings are merged, alleles below a quarter of the best fragment support are dropped, the rest is ordered by descending
// bAD1^2 * length.  The reference's std::sort over reverse iterators leaves ties in ascending string order for the small
// vectors that occur (insertion sort); that order is taken as the definition here and in the HIP path.
static void indel_majority(State &S, i32 refpos, int symbol, std::vector<GapAllele> &out) {
    out.clear();
    std::map<std::string, std::array<i32, 2>> indelmap;
    std::vector<GapTuple> t;
    size_t n_rows = 0;
    for (int strand = 0; strand < 2; strand++) {
        if (!(0 < S.FR(strand, UVC_FRAG_bDP, symbol, refpos - S.beg))) continue;
        gap_tuples(S, strand, refpos, symbol, t);
        n_rows += t.size();
        for (const GapTuple &g : t) { auto &v = indelmap[g.s]; v[0] += g.bq; v[1] += g.fq; }
    }
    if (0 == n_rows) { GapAllele a = { 0, 0, SYMBOL_DESC[symbol] }; out.push_back(a); return; }   // "Invalid indel detected", main.hpp:5415-5423
    i32 max_bAD1 = 0;
    for (const auto &kv : indelmap) max_bAD1 = max_(max_bAD1, kv.second[0]);
    for (const auto &kv : indelmap) if (kv.second[0] >= (max_bAD1 + 3) / 4) { GapAllele a = { kv.second[0], kv.second[1], kv.first }; out.push_back(a); }
    std::stable_sort(out.begin(), out.end(), [](const GapAllele &x, const GapAllele &y) {
        return ((i64)x.bAD1 * x.bAD1 * (i64)x.s.size()) > ((i64)y.bAD1 * y.bAD1 * (i64)y.s.size()); });
}

// ------------------------------------------------------------------------------------------------
// The calling step behind calc_qual: main.cpp:990-1168, output_germline (main.hpp:5483-5775) and the arithmetic of
// append_vcf_record (main.hpp:6027-6272).  Strings (GT text, REF / ALT, INFO, FORMAT) stay with the caller.
// ------------------------------------------------------------------------------------------------
static const int SYM_END = UVC_NUM_SYMBOLS;   // END_ALIGNMENT_SYMBOLS
static i32 hetLODQ(double a1, double a2, double expfrac, double powlaw_exponent) {   // main.hpp:5457-5462
    const i32 binomLODQ = (i32)calc_binom_10log10_likeratio(expfrac, a1, a2);
    const i32 powerLODQ = (i32)round(10.0 / log(10.0) * powlaw_exponent * max_(logit2((a1 + 0.5) * 0.5 / expfrac, (a2 + 0.5) * 0.5 / (1.0 - expfrac)), 0.0));
    return min_(binomLODQ, powerLODQ);
}
static inline int indel_n_units(int s) {   // SYMBOL_TO_INDEL_N_UNITS, main.hpp:271-279
    switch (s) { case UVC_LINK_D3P: return -3; case UVC_LINK_D2: return -2; case UVC_LINK_D1: return -1; case UVC_LINK_I3P: return 3; case UVC_LINK_I2: return 2; case UVC_LINK_I1: return 1; default: return 0; }
}
// one entry of symbol_format_vec: a scored allele, or the padding allele init_fmt (main.cpp:1043-1053)
struct GermAl { int symbol; i32 gVQ1, CONTQ, cDP0a, cDP1v; int rec; };
static void call_germline(const UvcParams &P, int refsymbol, std::vector<Fmt> &fm, i64 rec0) {
    const bool is_rescued = P.tumor_vcf_is_provided;
    std::vector<GermAl> v;
    for (size_t i = 0; i < fm.size(); i++) if (fm[i].symbol != UVC_BASE_NN) { GermAl a = { fm[i].symbol, fm[i].gVQ1, fm[i].CONTQ, fm[i].cDP0a, fm[i].cDP1v, (int)i }; v.push_back(a); }
    while (v.size() <= 4) { GermAl a = { SYM_END, 0, 0, 0, 50, -1 }; v.push_back(a); }
    // std::sort over reverse iterators with a comparator on ALODQ alone: descending gVQ1, equal values keep their order (insertion sort
    // of the short vectors that occur)
    std::stable_sort(v.begin(), v.end(), [](const GermAl &a, const GermAl &b) { return a.gVQ1 > b.gVQ1; });
    const GermAl *sel[4] = { NULL, NULL, NULL, NULL };
    int allele_idx = 1; i32 ref_alodq = INT32_MIN;
    for (const GermAl &a : v) {
        const bool isref = (refsymbol == a.symbol || UVC_BASE_NN == a.symbol || UVC_LINK_NN == a.symbol);
        if (isref && a.gVQ1 > ref_alodq) { sel[0] = &a; ref_alodq = a.gVQ1; }
        if ((!isref) && allele_idx <= 3) { sel[allele_idx] = &a; allele_idx++; }
    }
    i32 a0 = sel[0]->gVQ1, a1 = sel[1]->gVQ1, a2 = sel[2]->gVQ1, a3 = sel[3]->gVQ1;
    const bool isSubst = is_subst(refsymbol);
    const int symbolNN = ((isSubst || !is_rescued) ? UVC_BASE_NN : UVC_LINK_NN);
    double ad0 = sel[0]->cDP1v / 100.0, ad1 = sel[1]->cDP1v / 100.0, ad2 = sel[2]->cDP1v / 100.0;   // compute_norm_ad
    if (symbolNN == sel[1]->symbol) { ad0 += ad1; ad1 = 0; }
    if (symbolNN == sel[2]->symbol) { ad0 += ad2; ad2 = 0; }
    const i32 a0a1 = hetLODQ(ad0, ad1, 1.0 - P.germ_hetero_FA, P.powlaw_exponent), a1a0 = hetLODQ(ad1, ad0, P.germ_hetero_FA, P.powlaw_exponent);
    const i32 a1a2 = hetLODQ(ad1, ad2, 0.5, P.powlaw_exponent), a2a1 = hetLODQ(ad2, ad1, 0.5, P.powlaw_exponent);
    const i32 phred_homref = 0;
    const i32 phred_hetero = (isSubst ? P.germ_phred_hetero_snp : P.germ_phred_hetero_indel), phred_homalt = (isSubst ? P.germ_phred_homalt_snp : P.germ_phred_homalt_indel);
    const i32 phred_tri_al = (isSubst ? P.germ_phred_het3al_snp : P.germ_phred_het3al_indel);
    // CONTQ[1] of the padding allele is read out of bounds by the reference in T/N mode; it is taken as 0 here
    if (is_rescued) { a0 = min_(a0, sel[0]->CONTQ); a1 = min_(a1, sel[1]->CONTQ); a2 = min_(a2, sel[2]->CONTQ); a3 = min_(a3, sel[3]->CONTQ); }
    else a0 = min_(a0, sel[0]->CONTQ);
    const i32 a2penal = max_(a2 - (phred_tri_al - phred_hetero), 0), a3penal = max_(a3 - phred_hetero, 0);
    const i32 a01hetp = max_(max_(a0a1, a1a0) - (0 - 0), 0), a12hetp = max_(max_(a1a2, a2a1) - (3 - 0), 0), a03trip = max_(a0, a3);
    i32 tri_al_penal = 0;
    const int symb1 = sel[1]->symbol, symb2 = sel[2]->symbol;
    if (is_ins(symb1) && is_ins(symb2)) { tri_al_penal += 3; if (symb1 == symb2) { tri_al_penal += 3; if (UVC_LINK_I3P == symb1) tri_al_penal += 3; } }
    { const int n1 = indel_n_units(symb1), n2 = indel_n_units(symb2); if (n1 != 0 && n2 != 0) tri_al_penal -= between_(abs(n1 - n2) * 3 - 5, 0, 9); }
    const i32 GL4raw[4] = {
        (-phred_homref - a1 - a2penal - a3penal),
        (-phred_hetero - max_(a01hetp, a2) - max_(min_(a01hetp, a2) - phred_hetero, 0) - a3penal),
        (-phred_homalt - max_(a0, a2) - max_(min_(a0, a2) - phred_hetero, 0) - a3penal),
        (-phred_tri_al - max_(a12hetp, a03trip) - max_(min_(a12hetp, a03trip) - phred_hetero, 0) - max_(min_(a12hetp, min_(a0, a3)) - phred_hetero, 0) - tri_al_penal) };
    const i32 ret = GL4raw[0] - max_(GL4raw[1], max_(GL4raw[2], GL4raw[3]));
    // sort by (value, index) descending (PairSecondLess over reverse iterators, main.hpp:5464-5469): best genotype and the runner-up
    int order[4] = { 0, 1, 2, 3 };
    std::sort(order, order + 4, [&](int x, int y) { return GL4raw[x] > GL4raw[y] || (GL4raw[x] == GL4raw[y] && x > y); });
    const int GLidx = order[0];
    const i32 germ_GQ = GL4raw[order[0]] - GL4raw[order[1]];
    int emit = ((0x1 /* OUTVAR_GERMLINE */ & P.outvar_flag) ? 1 : 0);
    if (emit && 0 == GLidx && (!P.should_output_all_germline) && max_(sel[1]->cDP0a, sel[2]->cDP0a) <= 2) emit = 0;
    for (Fmt &f : fm) {
        f.vNLODQ = ret;
        for (int i = 0; i < 4; i++) f.GL4[i] = GL4raw[i];
        const i32 gst[8] = { a0, a1, a2, a3, a0a1, a1a0, a1a2, a2a1 };
        for (int i = 0; i < 8; i++) f.GST[i] = gst[i];
        f.germ_GT = GLidx; f.germ_GQ = germ_GQ; f.germ_emit = emit;
        f.g_ref_rel = sel[0]->rec; f.g_alt1_rel = sel[1]->rec; f.g_alt2_rel = sel[2]->rec;
        f.germ_ref = (sel[0]->rec < 0 ? -1 : (i32)(rec0 + sel[0]->rec)); f.germ_alt1 = (sel[1]->rec < 0 ? -1 : (i32)(rec0 + sel[1]->rec)); f.germ_alt2 = (sel[2]->rec < 0 ? -1 : (i32)(rec0 + sel[2]->rec));
    }
}

// calc_binom_powlaw_syserr_normv_quals, main.hpp:5982-6009
static void normv_quals(i32 out[4], double tAD, double tDP, i32 tVQ, i32 tnVQcap, double nAD, double nDP, i32 nVQ, double penal_dimret_coef, i32 prior_phred, i32 tn_dec_by_xm, double powlaw_exponent) {
    const i32 binom_b10log10like = (i32)calc_binom_10log10_likeratio((tDP - tAD) / (tDP), nDP - nAD, nAD);
    const double nADplus = nAD * between_(nDP / tDP - 1.0, 0.0, 1.0);
    const double bjpfrac = ((tAD + 0.5) / (tDP + 1.0)) / ((nAD + 0.5 + nADplus) / (nDP + 1.0 + nADplus));
    const i32 powlaw_b10log10like = (i32)round(powlaw_exponent * numstates2phred(bjpfrac));
    const i32 tnVQinc = max_(-prior_phred, max_((-(i32)nAD) * 3, min_(binom_b10log10like - prior_phred, powlaw_b10log10like - prior_phred)));
    const double l2 = log(max_(bjpfrac, 1.001)) / log(2);
    i32 tnVQdec = max_(0, nVQ - max_(0, min_(binom_b10log10like - prior_phred, (i32)(l2 * l2 * penal_dimret_coef))));
    tnVQdec = max_(tnVQdec, min_(nVQ + 9, tn_dec_by_xm));
    out[0] = binom_b10log10like; out[1] = powlaw_b10log10like; out[2] = tnVQdec; out[3] = min_(tnVQcap, tVQ + tnVQinc) - tnVQdec;
}
// calc_binom_powlaw_syserr_normv_quals2, main.hpp:6011-6025
static void normv_quals2(i32 out[4], double tAD, double tDP, i32 tVQ, i32 tnVQcap, double nAD, double nDP, i32 nVQ) {
    const i32 binom = (i32)calc_binom_10log10_likeratio((tDP - tAD) / (tDP), nDP - nAD, nAD);
    const i32 powlaw = (nAD <= 3 ? binom : (i32)round(binom * 3 / nAD));
    const double m = max_((double)(min_(binom, powlaw) - 3), max_(-3 * nAD, -3.0));   // TVN_MICRO_VQ_DELTA = 3; MAX3 promotes to double
    out[0] = binom; out[1] = powlaw; out[2] = nVQ; out[3] = (i32)between_((double)tVQ + m - (double)nVQ, 0.0, (double)tnVQcap);
}

// main.cpp:1081-1147 + append_vcf_record for one record; `fm` are the records of its group (front-pushed: reffmt = the reference allele)
static void call_record(const UvcParams &P, Fmt &fmt, const Fmt &reffmt, const std::vector<Fmt> &fm, int st, int refsymbol, bool should_output_ref_allele) {
    const int symbol = fmt.symbol;
    const bool tprov = P.tumor_vcf_is_provided;
    const UvcTumorKey *tk = fmt.tkey;
    fmt.out = 0; fmt.vHGQ = 0; fmt.NLODQ = 0; fmt.NLODV = SYM_END; fmt.TLODQ = 0; fmt.SomaticQ = 0; fmt.QUALbits = 0; fmt.FILTER = 0; fmt.keep = 0;
    for (int i = 0; i < 4; i++) { fmt.TNBQF[i] = 0; fmt.TNCQF[i] = 0; }
    const bool will_generate_out = (!tprov ? ((P.outvar_flag & 0x4 /* OUTVAR_ANY */) != 0) : (tk != NULL && (P.outvar_flag & 0x2 /* OUTVAR_SOMATIC */)));
    const bool is_out_blocked = (((UVC_BASE_NN == symbol) && !(0x20 & P.outvar_flag)) || ((UVC_LINK_NN == symbol) && !(0x40 & P.outvar_flag)));
    if (!(will_generate_out && !is_out_blocked)) return;
    fmt.out = 1;
    const i32 germ_phred = (is_subst(symbol) ? P.germ_phred_hetero_snp : P.germ_phred_hetero_indel);
    const i32 nlodq_singlesite = fmt.vNLODQ;
    const i32 nlodq_singlesample = nlodq_singlesite - 3 + germ_phred;
    i32 nlodq1 = ((UVC_BASE_SYMBOL == st) ? P.germ_phred_hetero_snp : P.germ_phred_hetero_indel);
    int argmin_nlodq_symbol = SYM_END;
    const i32 totBDP = fmt.BDPb[0] + fmt.BDPb[1];
    // the TumorKeyInfo the arithmetic reads: the tumor record (normal sample) or this record itself (fill_tki, main.hpp:5912-5930)
    i32 t_BDP, t_bDP, t_CDP1x, t_cDP1x, t_cVQ1, t_cPCQ1, t_CDP2x, t_cDP2x, t_cVQ2, t_cPCQ2, t_bNMQ, t_tDP = 0;
    if (tprov) {
        i32 nlodq_inc = 999;
        const int ptr[2] = { fmt.g_alt1_rel, fmt.g_alt2_rel };
        for (int k = 0; k < 2; k++) {
            // fmtptr1 / fmtptr2 of output_germline; the padding allele has cDP1x = {} (collectget default 50), CDP1x[0] = 0 and VTI = END
            const Fmt *fp = (ptr[k] >= 0 ? &fm[(size_t)ptr[k]] : NULL);
            const int normsymbol = (fp ? fp->symbol : SYM_END);
            const i32 bgerr_norm_max_ad = (fp ? fp->cDP1x : 50);
            const double tAD = (tk->cDP1x + 1 * 50) / 100.0, tDP = (tk->CDP1x + 2 * 50) / 100.0;
            const double nAD = (bgerr_norm_max_ad + 1 * 50) / 100.0, nDP = ((fp ? fp->CDPv[2][0] : 0) + 2 * 50) / 100.0;
            const double bjpfrac = ((tAD) / (tDP)) / ((nAD) / (nDP));
            const i32 binom_b10log10like = (i32)calc_binom_10log10_likeratio((tDP - tAD) / (tDP), nDP - nAD, nAD);
            const i32 powlaw_b10log10like = (i32)(P.powlaw_exponent * 10 / log(10) * log(bjpfrac));
            const i32 inc_snp = 2 * P.germ_phred_hetero_snp - P.germ_phred_het3al_snp, inc_indel = 2 * P.germ_phred_hetero_indel - P.germ_phred_het3al_indel;
            const i32 triallele_inc = ((normsymbol != symbol) ? (is_subst(symbol) ? inc_snp : inc_indel) : 0);
            const i32 new_nlodq_inc = (i32)between_((double)min_(binom_b10log10like, powlaw_b10log10like), -3.0, P.powlaw_anyvar_base) + triallele_inc;
            if (nlodq_inc > new_nlodq_inc) { nlodq_inc = new_nlodq_inc; argmin_nlodq_symbol = normsymbol; }
        }
        const i32 n_norm_alts = (totBDP - (reffmt.bDPf + reffmt.bDPr)) + (fmt.bDPf + fmt.bDPr);
        nlodq1 = max_(max_(nlodq_singlesite, germ_phred + nlodq_inc), tk->vHGQ + min_(3, totBDP - n_norm_alts * (i32)round(0.5 / P.contam_any_mul_frac)));
        t_BDP = tk->BDP; t_bDP = tk->bDP; t_CDP1x = tk->CDP1x; t_cDP1x = tk->cDP1x; t_cVQ1 = tk->cVQ1; t_cPCQ1 = tk->cPCQ1;
        t_CDP2x = tk->CDP2x; t_cDP2x = tk->cDP2x; t_cVQ2 = tk->cVQ2; t_cPCQ2 = tk->cPCQ2; t_bNMQ = tk->bNMQ; t_tDP = tk->tDP;
    } else {
        nlodq1 = nlodq_singlesample;
        t_BDP = totBDP; t_bDP = fmt.bDPf + fmt.bDPr; t_CDP1x = fmt.CDPv[2][0]; t_cDP1x = fmt.cDP1x; t_cVQ1 = fmt.cVQ1; t_cPCQ1 = fmt.cPCQ1;
        t_CDP2x = fmt.CDPv[5][0]; t_cDP2x = fmt.cDP2x; t_cVQ2 = fmt.cVQ2; t_cPCQ2 = fmt.cPCQ2; t_bNMQ = fmt.bNMQ;
    }
    fmt.vHGQ = nlodq_singlesample; fmt.NLODV = argmin_nlodq_symbol;
    // ---- append_vcf_record ----
    const bool is_processing_normal = tprov;   // tki.ref_alt.size() > 0
    // nfm = the normal's record, or FORMAT_UNCOV (all zero / empty) for a single sample
    const i32 nfm_cDP1x = (is_processing_normal ? fmt.cDP1x : 0), nfm_CDP1x = (is_processing_normal ? fmt.CDPv[2][0] : 0);
    const i32 nfm_cDP2x = (is_processing_normal ? fmt.cDP2x : 0), nfm_CDP2x = (is_processing_normal ? fmt.CDPv[5][0] : 0);
    const i32 nfm_cVQ1 = (is_processing_normal ? fmt.cVQ1 : 0), nfm_cVQ2 = (is_processing_normal ? fmt.cVQ2 : 0);
    const i32 nfm_BDP = (is_processing_normal ? totBDP : 0), nfm_CDP1 = (is_processing_normal ? fmt.CDP1b[0] + fmt.CDP1b[1] : 0);
    const i32 inc_snp = max_(0, 2 * P.germ_phred_hetero_snp - P.germ_phred_het3al_snp - 0), inc_indel = max_(0, 2 * P.germ_phred_hetero_indel - P.germ_phred_het3al_indel - 0);   // TIN_CONTAM_MICRO_VQ_DELTA = 0
    i32 phred_het3al_chance_inc = (is_subst(symbol) ? inc_snp : inc_indel);
    if (is_ins(symbol) || is_del(symbol)) phred_het3al_chance_inc = (i32)nnminus(inc_indel + 1, fmt.gapSa_len);
    const i32 qmin = P.microadjust_syserr_MQ_NMR_tn_syserr_no_penal_qual_min, qmax = P.microadjust_syserr_MQ_NMR_tn_syserr_no_penal_qual_max;
    const i32 tn_dec_by_xm = between_(min_(fmt.bNMQ, t_bNMQ), qmin, qmax) - qmin;
    double add1 = 0, add2 = 0;
    if (is_processing_normal && implies_short_frag(fmt, P.lib_wgs_min_avg_fraglen)) { add1 = P.lib_nonwgs_normal_add_mul_ad * nfm_cDP1x / 100.0; add2 = P.lib_nonwgs_normal_add_mul_ad * nfm_cDP2x / 100.0; }
    i32 tn_dec_both = 0;
    if (is_processing_normal && t_tDP > 500 && fmt.DP > 500 && is_del(symbol) && fmt.APDP[2] * 3 > fmt.APDP[0]) tn_dec_both = min_((i32)nnminus(nfm_cVQ1, 31), 9);
    const i32 prior_phred = ((UVC_PLATFORM_IONTORRENT == P.inferred_sequencing_platform) ? (3 + 8) : 3);
    i32 bq4[4], cq4[4];
    if (P.tn_syserr_norm_devqual >= 0) normv_quals(bq4, (t_cDP1x + 0.5) / 100.0 + 0.0, (t_CDP1x + 1.0) / 100.0 + 0.0, t_cVQ1, t_cPCQ1, (nfm_cDP1x + 0.5) / 100.0 + 0.0 + add1, (nfm_CDP1x + 1.0) / 100.0 + 0.0 + add1,
                                               (i32)nnminus(nfm_cVQ1, phred_het3al_chance_inc), P.tn_syserr_norm_devqual, prior_phred, tn_dec_by_xm, P.powlaw_exponent);
    else normv_quals2(bq4, (t_cDP1x + 0.5) / 100.0 + 0.0, (t_CDP1x + 1.0) / 100.0 + 0.0, t_cVQ1, t_cPCQ1, (nfm_cDP1x + 0.5) / 100.0 + 0.0 + add1, (nfm_CDP1x + 1.0) / 100.0 + 0.0 + add1, (i32)nnminus(nfm_cVQ1, phred_het3al_chance_inc));
    const i32 converted_nfm_cVQ2 = nfm_cVQ1 - (3 * (nfm_BDP + 1) / (nfm_CDP1 + 1));
    const i32 norm_norm_vq = (i32)nnminus(nfm_cVQ2, max_(phred_het3al_chance_inc, 3) - 3);
    if (P.tn_syserr_norm_devqual >= 0) normv_quals(cq4, (t_cDP2x + 0.5) / 100.0 + 0.0, (t_CDP2x + 1.0) / 100.0 + 0.0, t_cVQ2, t_cPCQ2, (nfm_cDP2x + 0.5) / 100.0 + 0.0 + add2, (nfm_CDP2x + 1.0) / 100.0 + 0.0 + add2,
                                               norm_norm_vq, P.tn_syserr_norm_devqual, prior_phred, max_(tn_dec_by_xm, min_(max_(nfm_cVQ2, converted_nfm_cVQ2), 8 + 4)), P.powlaw_exponent);
    else normv_quals2(cq4, (t_cDP2x + 0.5) / 100.0 + 0.0, (t_CDP2x + 1.0) / 100.0 + 0.0, t_cVQ2, t_cPCQ2, (nfm_cDP2x + 0.0) / 100.0 + 0.5 + add2, (nfm_CDP2x + 0.0) / 100.0 + 1.0 + add2, norm_norm_vq);
    const i32 tlodq1 = max_(bq4[3], cq4[3] + 0);
    const bool is_cytosine_deanim_CT = ((UVC_BASE_C == refsymbol && UVC_BASE_T == symbol) || (UVC_BASE_G == refsymbol && UVC_BASE_A == symbol));
    auto prob2realphred = [](double p) { return -10 * log(p) / log(10); };
    const double b_min_tlodq = 2 + 3 - prob2realphred((t_bDP + 1e-3) / (t_BDP + 1)) / 10.0;
    const double c2v_min_tlodq = 2 + 5 - prob2realphred((t_cDP2x * 0.01 + 1e-5) / (t_CDP2x * 0.01 + 1) / (is_cytosine_deanim_CT ? 5 : 1)) / 10.0;
    const float lowestVAQ = (float)max_(b_min_tlodq, c2v_min_tlodq);
    const i32 tlodq = ((tlodq1 >= 10) ? tlodq1 : (tlodq1 * 3 - 20)) - tn_dec_both;
    const i32 nlodq = nlodq1 - tn_dec_both;
    const i32 somaticq = min_(tlodq, nlodq);
    float v = (is_processing_normal ? ((float)somaticq) : max_((float)tlodq, lowestVAQ));
    { const float base = (float)pow(10.0, 0.1); if (v < 10.0f) v = log1pf(powf(base, v)) / logf(base); }   // calc_non_negative<float>, main_conversion.hpp:163-171
    const float vcfqual = v;
    fmt.TLODQ = tlodq; fmt.NLODQ = nlodq; fmt.SomaticQ = somaticq;
    for (int i = 0; i < 4; i++) { fmt.TNBQF[i] = bq4[i]; fmt.TNCQF[i] = cq4[i]; }
    memcpy(&fmt.QUALbits, &vcfqual, 4);
    fmt.FILTER = (vcfqual < 10 ? 0 : vcfqual < 20 ? 1 : vcfqual < 30 ? 2 : vcfqual < 40 ? 3 : vcfqual < 50 ? 4 : vcfqual < 60 ? 5 : 6);
    const i32 vad1curr = fmt.aBQ2; const i64 vdp1curr = fmt.ABQ2[0];
    const i32 vad2curr = t_bDP, vdp2curr = t_BDP;
    const bool keep_var = (((vcfqual >= P.vqual) || ((!tprov) && ((vad1curr >= P.vad1 && vdp1curr >= P.vdp1 && (vdp1curr * P.vfa1) <= vad1curr) || (vad2curr >= P.vad2 && vdp2curr >= P.vdp2 && (vdp2curr * P.vfa2) <= vad2curr))))
                           && (symbol != refsymbol || should_output_ref_allele));
    const i32 min_ad = ((symbol == refsymbol) ? P.min_r_ad : P.min_a_ad);
    fmt.keep = (keep_var && t_bDP >= min_ad) ? 1 : 0;
}

// per-position driver, main.cpp:608-1000
// ---- VCF text of one written record: the string half of append_vcf_record (main.hpp:6050-6067, 6206-6270) and, for the sample column,
// the field values of the bcfrec::BcfFormat it streams, as "TAG \t N \t v1 \x1f v2 ..." lines.  The test feeds those lines to the
// REFERENCE's own streamAppendBcfFormat (oracle/ref_vcf_driver.cpp over the header its generator prints), so the layout of the sample
// column is the reference's, not a restatement.  Not carried: note (left at the reference's default).
static void vcf_emit(State &S, const Fmt &f, const Fmt &rf, const std::string &indelstring, const std::vector<UvcGapRow> &gap_rows, const std::vector<u8> &gap_seq, i32 zpos) {
    const UvcParams &P = S.P;
    const int symbol = f.symbol, refsymbol = f.refsymbol;
    const int st = (symbol <= UVC_BASE_NN ? UVC_BASE_SYMBOL : UVC_LINK_SYMBOL), nn = (st == UVC_BASE_SYMBOL ? UVC_BASE_NN : UVC_LINK_NN);
    const i32 refpos = f.refpos; const i64 x = refpos - S.beg;
    const i32 regionpos = refpos - S.beg;
    auto row_text = [&](const UvcGapRow &g) { std::string t; if (g.seq_off >= 0) for (int c = 0; c < g.len; c++) t += "ACGTN"[gap_seq[(size_t)g.seq_off + c]]; else t = S.refstring.substr((size_t)(g.refpos - S.beg), (size_t)g.len); return t; };
    // ---- CHROM .. INFO ----
    i32 vcfpos; std::string vcfref, vcfalt;
    if (indelstring.size() > 0) {
        vcfpos = refpos; vcfref = (regionpos > 0 ? S.refstring.substr((size_t)regionpos - 1, 1) : "n"); vcfalt = vcfref;
        if ('<' == indelstring[0]) vcfalt = indelstring; else if (is_ins(symbol)) vcfalt += indelstring; else vcfref += indelstring;
    } else {
        if (is_subst(symbol)) { vcfpos = refpos + 1; vcfref = S.refstring.substr((size_t)regionpos, 1); }
        else { vcfpos = refpos; vcfref = (regionpos > 0 ? S.refstring.substr((size_t)regionpos - 1, 1) : "n"); }
        vcfalt = (symbol < NSYM ? SYMBOL_DESC[symbol] : "<NONE>");
    }
    float vcfqual; memcpy(&vcfqual, &f.QUALbits, 4);
    const char *const FILTERS[7] = { "Q10", "Q20", "Q30", "Q40", "Q50", "Q60", "PASS" };
    i32 cdpd_b[2] = { 0, 0 }, ddp2_sum = 0;
    for (int k = 0; k < ST_NSYMBOLS[st]; k++) { const int sy = ST_SYMBOLS[st][k]; cdpd_b[0] += S.FA(0, UVC_FAM_cDPD, sy, x); cdpd_b[1] += S.FA(1, UVC_FAM_cDPD, sy, x); ddp2_sum += S.DU(UVC_DUPLEX_dDP2, sy, x); }
    const i32 DDP2[2] = { ddp2_sum, S.DU(UVC_DUPLEX_dDP2, nn, x) };
    std::vector<size_t> own_rows;   // fill_by_indel_info: forward rows, then reverse rows (main.cpp:858-869)
    if (is_ins(symbol) || is_del(symbol)) for (size_t q = 0; q < gap_rows.size(); q++) if (gap_rows[q].refpos == refpos && gap_rows[q].symbol == symbol) own_rows.push_back(q);
    i32 cond_altDP;   // fill_conditional_tki, main.hpp:5943-5979
    if (is_ins(symbol) || is_del(symbol)) { cond_altDP = 0; for (size_t q : own_rows) if (row_text(gap_rows[q]) == indelstring) cond_altDP += gap_rows[q].c2dAD; }
    else cond_altDP = f.cDPDf + f.cDPDr + f.dDP2;
    const i32 tADCR[2] = { rf.cDPDf + rf.cDPDr + rf.dDP2, cond_altDP };
    const bool normal = (f.tkey != NULL);
    std::string info = (normal ? "SOMATIC" : "ANY_VAR");
    info += ";SomaticQ=" + std::to_string(f.SomaticQ) + ";TLODQ=" + std::to_string(f.TLODQ) + ";NLODQ=" + std::to_string(f.NLODQ);
    info += std::string(";NLODV=") + (f.NLODV < NSYM ? SYMBOL_DESC[f.NLODV] : "<NONE>");
    info += ";TNBQF=" + std::to_string(f.TNBQF[0]) + "," + std::to_string(f.TNBQF[1]) + "," + std::to_string(f.TNBQF[2]) + "," + std::to_string(f.TNBQF[3]);
    info += ";TNCQF=" + std::to_string(f.TNCQF[0]) + "," + std::to_string(f.TNCQF[1]) + "," + std::to_string(f.TNCQF[2]) + "," + std::to_string(f.TNCQF[3]);
    if (!normal) {
        info += ";tbDP=" + std::to_string(f.BDPb[0] + f.BDPb[1]) + ";tDP=" + std::to_string(f.DP) + ";tAD=" + std::to_string(rf.AD) + "," + std::to_string(f.AD);
        info += ";t2DP=" + std::to_string((cdpd_b[0] + cdpd_b[1]) + (DDP2[0] + DDP2[1])) + ";t2AD=" + std::to_string(tADCR[0]) + "," + std::to_string(tADCR[1]);
    } else {
        const UvcTumorKey &k = *f.tkey;
        info += ";tbDP=" + std::to_string(k.BDP) + ";tDP=" + std::to_string(k.tDP) + ";tAD=" + std::to_string(k.tAD0) + "," + std::to_string(k.tAD1);
        info += ";t2DP=" + std::to_string(k.t2DP) + ";t2AD=" + std::to_string(tADCR[0]) + "," + std::to_string(tADCR[1]);
        info += ";nDP=" + std::to_string(f.DP) + ";nAD=" + std::to_string(rf.AD) + "," + std::to_string(f.AD) + ";n2AD=0,0";
    }
    {   // RU / RC: indelpos_to_context at this zerobased_pos (main.cpp:609-613)
        i32 us = 0, rn = 0;
        indelpos_to_context(us, rn, S.refstring, zpos - S.beg, P.indel_str_repeatsize_max);
        const i32 at = zpos - S.beg;
        info += ";RU=" + ((at >= 0 && at < (i32)S.refstring.size()) ? S.refstring.substr((size_t)at, (size_t)us) : std::string()) + ";RC=" + std::to_string(rn);
    }
    {
        const i32 d = P.indel_adj_tracklen_dist, nr = (i32)S.rtr.size();
        const Rtr &r1 = S.rtr[(size_t)(max_(regionpos, d) - d)], &r2 = S.rtr[(size_t)min_(regionpos + d, nr - d)];
        info += ";R3X2=" + std::to_string(r1.tracklen ? S.beg + r1.begpos : 0) + "," + std::to_string(r1.tracklen) + "," + std::to_string(r1.unitlen) + ","
              + std::to_string(r2.tracklen ? S.beg + r2.begpos : 0) + "," + std::to_string(r2.tracklen) + "," + std::to_string(r2.unitlen);
    }
    std::string fixed = S.vcf_sink->tname + "\t" + std::to_string(vcfpos) + "\t.\t" + vcfref + "\t" + vcfalt + "\t" + std::to_string(vcfqual) + "\t" + FILTERS[min_(max_(f.FILTER, 0), 6)] + "\t" + info;
    // ---- the bcfrec::BcfFormat of this record ----
    std::string spec;
    auto L = [&](const char *tag, std::initializer_list<i64> v) { spec += tag; spec += '\t'; spec += std::to_string(v.size()); spec += '\t'; bool first = true; for (i64 e : v) { if (!first) spec += '\x1f'; first = false; spec += std::to_string(e); } spec += '\n'; };
    auto LS = [&](const char *tag, const std::vector<std::string> &v) { spec += tag; spec += '\t'; spec += std::to_string(v.size()); spec += '\t'; for (size_t q = 0; q < v.size(); q++) { if (q) spec += '\x1f'; spec += v[q]; } spec += '\n'; };
    auto LV = [&](const char *tag, const std::vector<i64> &v) { spec += tag; spec += '\t'; spec += std::to_string(v.size()); spec += '\t'; for (size_t q = 0; q < v.size(); q++) { if (q) spec += '\x1f'; spec += std::to_string(v[q]); } spec += '\n'; };
    L("enable_tier2_consensus_format_tags", { f.tier2 });
    LS("GT", { "./1" });   // main.cpp:1097; GQ / HQ / FT keep the defaults
    {
        static const char *const NAMES[19] = { "aStrand", "aBQXM", "aInsertSize", "aAlignL", "aAlignR", "aPositionL", "aPositionR", "abPositionL", "abPositionR",
                                               "bcDup", "cbDup", "c0Orientation", "c2Orientation", "c2PositionL", "c2PositionR", "c2AlignL", "c2AlignR", "c2StrictPosL", "c2StrictPosR" };
        std::string fts;
        for (int b = 0; b < 19; b++) if (f.FTS & (1 << b)) { if (!fts.empty()) fts += "|"; fts += std::string(NAMES[b]) + "-" + std::to_string(f.FTSpct[b]); }
        LS("FTS", { fts.empty() ? std::string("PASS") : fts });
    }
    L("DP", { f.DP }); L("AD", { rf.AD, f.AD }); L("bDP", { f.bDP }); L("bAD", { rf.bAD, f.bAD }); L("c2DP", { f.c2DP }); L("c2AD", { rf.c2AD, f.c2AD });
    LV("APDP", std::vector<i64>(f.APDP, f.APDP + 12)); LV("APXM", std::vector<i64>(f.APXM, f.APXM + 8));
    L("APLRID", { S.p64(UVC_P_a_near_ins_l_pow2len, x), S.p64(UVC_P_a_near_ins_r_pow2len, x), S.p64(UVC_P_a_near_del_l_pow2len, x), S.p64(UVC_P_a_near_del_r_pow2len, x) });
    LV("APLRI", std::vector<i64>(f.APLRI, f.APLRI + 4));
    L("APLRP", { S.p32(UVC_P_a_l_dist_sum, x), S.p32(UVC_P_a_r_dist_sum, x), S.p32(UVC_P_a_inslen_sum, x), S.p32(UVC_P_a_dellen_sum, x) });
    L("ALRPxT", { S.th(UVC_T_aLPxT, x), S.th(UVC_T_aRPxT, x) });
    L("ALRIT", { S.th(UVC_T_aLI1T, x), S.th(UVC_T_aLI2T, x), S.th(UVC_T_aRI1T, x), S.th(UVC_T_aRI2T, x) });
    L("ALRIt", { S.th(UVC_T_aLI1t, x), S.th(UVC_T_aLI2t, x), S.th(UVC_T_aRI1t, x), S.th(UVC_T_aRI2t, x) });
    L("ALRPt", { S.th(UVC_T_aLP1t, x), S.th(UVC_T_aLP2t, x), S.th(UVC_T_aRP1t, x), S.th(UVC_T_aRP2t, x) });
    L("ALRBt", { S.th(UVC_T_aLB1t, x), S.th(UVC_T_aLB2t, x), S.th(UVC_T_aRB1t, x), S.th(UVC_T_aRB2t, x) });
#define RR(tag) L(#tag, { rf.tag, f.tag })
#define T2(tag) L(#tag, { f.tag[0], f.tag[1] })
    RR(aMQs); T2(AMQs); RR(a1BQf); T2(A1BQf); RR(a1BQr); T2(A1BQr);
    RR(aDPff); T2(ADPff); RR(aDPfr); T2(ADPfr); RR(aDPrf); T2(ADPrf); RR(aDPrr); T2(ADPrr);
    RR(aLP1); T2(ALP1); RR(aLP2); T2(ALP2); RR(aLPL); T2(ALPL);
    RR(aRP1); T2(ARP1);
    RR(aRP2); T2(ARP2); RR(aRPL); T2(ARPL);
    RR(aLB1); RR(aLB2); T2(ALB2); RR(aLBL); T2(ALBL); RR(aRB1); RR(aRB2); T2(ARB2); RR(aRBL); T2(ARBL);
    RR(aLI1); RR(aLI2); T2(ALI2); RR(aLIr); T2(ALIr); RR(aRI1); RR(aRI2); T2(ARI2); RR(aRIf); T2(ARIf);
    RR(aBQ2); T2(ABQ2); RR(aPF2); T2(APF2); RR(aP1); T2(AP1); RR(aP2); T2(AP2);
    RR(aPF1); RR(aLIT); RR(aRIT); RR(aP3); RR(aNC);
    RR(bDPf); RR(bDPr); T2(BDPb); RR(bTAf); RR(bTAr); T2(BTAb); RR(bTBf); RR(bTBr); T2(BTBb);   // BDPd is never filled
    RR(cDP1f); RR(cDP1r); T2(CDP1b); L("CDP1d", { S.FA(0, UVC_FAM_cDP1, nn, x), S.FA(0, UVC_FAM_cDP1, nn, x) });   // fill_symboltype_nn_fmt, main.hpp:3774-3786
    RR(cDP12f); RR(cDP12r); T2(CDP12b); RR(cDP2f); RR(cDP2r); T2(CDP2b);
    RR(c2BQ2); T2(C2BQ2); RR(c2LP0); T2(C2LP0); RR(c2RP0); T2(C2RP0);
    RR(c2LP1); RR(c2LP2); T2(C2LP2); RR(c2RP1); RR(c2RP2); T2(C2RP2); RR(c2LPL); T2(C2LPL); RR(c2RPL); T2(C2RPL);
    RR(c2LB1); RR(c2LB2); T2(C2LB2); RR(c2RB1); RR(c2RB2); T2(C2RB2); RR(c2LBL); T2(C2LBL); RR(c2RBL); T2(C2RBL);
    RR(cDP3f); RR(cDP3r); T2(CDP3b); RR(cDP21f); RR(cDP21r); RR(cDPMf); RR(cDPMr); RR(cDPmf); RR(cDPmr); RR(cDPDf); RR(cDPDr);
    {
        auto frsum = [&](int fld, i64 out2[2]) { for (int sd = 0; sd < 2; sd++) { i32 acc = 0; for (int k = 0; k < ST_NSYMBOLS[st]; k++) acc += S.FA(sd, fld, ST_SYMBOLS[st][k], x); out2[sd] = acc; } };
        i64 v2[2];
        frsum(UVC_FAM_cDP21, v2); L("CDP21b", { v2[0], v2[1] }); frsum(UVC_FAM_cDPM, v2); L("CDPMb", { v2[0], v2[1] });
        frsum(UVC_FAM_cDPm, v2); L("CDPmb", { v2[0], v2[1] }); frsum(UVC_FAM_cDPD, v2); L("CDPDb", { v2[0], v2[1] });
    }
    T2(DDP1); RR(dDP1); L("DDP2", { DDP2[0], DDP2[1] }); RR(dDP2);
    RR(aBQ); RR(a2BQf); RR(a2BQr); RR(a2XM2); RR(a2BM2); RR(aBQQ);
    RR(bMQ); RR(aAaMQ); RR(bNMQ); RR(bNMa); RR(bNMb); RR(bMQQ);
    RR(bIAQb); RR(bIADb); RR(bIDQb); RR(cIAQf); RR(cIADf); RR(cIDQf); RR(cIAQr); RR(cIADr); RR(cIDQr);
    RR(bIAQ); RR(cIAQ); RR(bTINQ); RR(cTINQ); RR(cPCQ1); RR(cPLQ1); RR(cVQ1); RR(gVQ1); RR(cPCQ2); RR(cPLQ2); RR(cVQ2); RR(cMmQ); RR(dVQinc);
    RR(cDP1v); RR(cDP1w); RR(cDP1x); RR(cDP2v); RR(cDP2w); RR(cDP2x);
    { static const char *const NM[6] = { "CDP1v", "CDP1w", "CDP1x", "CDP2v", "CDP2w", "CDP2x" }; for (int q = 0; q < 6; q++) L(NM[q], { f.CDPv[q][0], f.CDPv[q][1] }); }
    RR(CONTQ);
    LV("nPF", std::vector<i64>(f.nPF, f.nPF + 2)); LV("nNFA", std::vector<i64>(f.nNFA, f.nNFA + 6)); LV("nAFA", std::vector<i64>(f.nAFA, f.nAFA + 9)); LV("nBCFA", std::vector<i64>(f.nBCFA, f.nBCFA + 10));
    L("VTI", { refsymbol, symbol }); LS("VTD", { SYMBOL_DESC[refsymbol], SYMBOL_DESC[symbol] });
    T2(cVQ1M); T2(cVQ2M);
    {
        std::vector<std::string> am, sm;
        for (int q = 0; q < 2; q++) { am.push_back(f.cVQAM[q] < NSYM ? SYMBOL_DESC[f.cVQAM[q]] : "<NONE>"); sm.push_back((f.cVQSM[q] >= 0 && f.cVQSM[q] < (i32)gap_rows.size()) ? row_text(gap_rows[(size_t)f.cVQSM[q]]) : std::string()); }
        LS("cVQAM", am); LS("cVQSM", sm);
    }
    {
        std::vector<i64> nf, nr, b1, c1, c2, c2d; std::vector<std::string> seqs;
        if (is_ins(symbol) || is_del(symbol)) for (int sd = 0; sd < 2; sd++) if (0 < S.FR(sd, UVC_FRAG_bDP, symbol, x)) {
            i64 cnt = 0;
            for (size_t q : own_rows) if (gap_rows[q].strand == sd) { cnt++; seqs.push_back(row_text(gap_rows[q])); b1.push_back(gap_rows[q].bAD1); c1.push_back(gap_rows[q].cAD1); c2.push_back(gap_rows[q].c2AD); c2d.push_back(gap_rows[q].c2dAD); }
            (sd ? nr : nf).push_back(cnt);
        }
        LV("gapNf", nf); LV("gapNr", nr); LS("gapSeq", seqs); LV("gapbAD1", b1); LV("gapcAD1", c1); LV("gc2AD", c2); LV("gc2dAD", c2d);
    }
    RR(bDPa); RR(cDP0a); LS("gapSa", { std::string(), indelstring });
    LS("bHap", { hap_phase_string(S, 0, f.refpos, f.symbol) }); LS("cHap", { hap_phase_string(S, 1, f.refpos, f.symbol) }); LS("c2Hap", { hap_phase_string(S, 2, f.refpos, f.symbol) });   // main.hpp:4242-4244
    L("vHGQ", { f.vHGQ }); T2(vAC);
    if (st == UVC_BASE_SYMBOL) L("vNLODQ", { f.vNLODQ, 0 }); else L("vNLODQ", { 0, f.vNLODQ });
#undef RR
#undef T2
    S.vcf_sink->fixed.push_back(fixed); S.vcf_sink->spec.push_back(spec); S.vcf_sink->tier2.push_back(f.tier2);
}

// The two position-level lines written in front of the records of a zerobased_pos: the MGVCF block (main.cpp:655-735) and
// ADDITIONAL_INDEL_CANDIDATE (main.cpp:759-799).  Whole lines; the sink marks them with tier2 = -1.
static void position_lines(State &S, i32 region_beg, i32 refpos, i32 zpos, i32 prev_tracklen, i32 curr_tracklen, i32 repeatunit_size, i32 repeatnum) {
    const UvcParams &P = S.P;
    const std::string &tname = S.vcf_sink->tname;
    auto sym_of = [](char c) { switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; case 'I': case 'i': return 6; case '-': case '_': return 9; default: return 4; } };
    auto put_line = [&](const std::string &l) { S.vcf_sink->fixed.push_back(l); S.vcf_sink->spec.push_back(std::string()); S.vcf_sink->tier2.push_back(-1); };
    if ((P.outvar_flag & 0x8) && (((refpos % 1000) == 0) || (refpos == region_beg))) {   // main.cpp:655-656: refpos == incluBegPosition, the begin of the BED line
        const i32 init_refQ = (INT_MAX / 2 + 1);
        i32 prev_b = 0, prev_c = 0, prev_c12 = 0, prev_q = init_refQ;
        auto depths_diff = [](i32 cur, i32 prev, i32 mul, i32 add) { const i32 lo = min_(cur, prev), hi = max_(cur, prev); if ((i64)lo * mul >= (i64)hi * 100) return false; if (lo + add >= hi) return false; return true; };
        std::vector<i64> v;
        const i32 rp2end = min_(refpos + 1000 + 1, S.end);
        for (i32 rp2 = refpos; rp2 < rp2end; rp2++) {
            const int stypes[2] = { UVC_LINK_SYMBOL, UVC_BASE_SYMBOL };   // SYMBOL_TYPES_IN_VCF_ORDER
            for (int stype : stypes) {
                const i64 x = rp2 - S.beg;
                const i32 off = rp2 - S.beg;
                const int base_m = (off < (i32)S.refstring.size()) ? sym_of(S.refstring[(size_t)off]) : UVC_BASE_N;
                const int refsymbol = (UVC_BASE_SYMBOL == stype ? base_m : UVC_LINK_M);
                i32 b = 0, c = 0, c12 = 0;
                for (int sd = 0; sd < 2; sd++) for (int k = 0; k < ST_NSYMBOLS[stype]; k++) { const int sy = ST_SYMBOLS[stype][k]; b += S.FR(sd, UVC_FRAG_bDP, sy, x); c += S.FA(sd, UVC_FAM_cDP1, sy, x); c12 += S.FA(sd, UVC_FAM_cDP12, sy, x); }
                const i32 ref_c = S.FA(0, UVC_FAM_cDP12, refsymbol, x) + S.FA(1, UVC_FAM_cDP12, refsymbol, x);
                const i32 nonref_c = c12 - ref_c;
                const double ref_like_binom = -calc_binom_10log10_likeratio(P.contam_any_mul_frac, nonref_c + 0.5, c + 1.0);
                const double ref_like_powlaw = -max_(0.0, P.powlaw_exponent * (10 / log(10)) * logit2((nonref_c + 0.5) / (c + 1.0), P.contam_any_mul_frac));
                const double nonref_like_binom = -calc_binom_10log10_likeratio(P.germ_hetero_FA, ref_c + 0.5, c + 1.0);
                const double nonref_like_powlaw = -max_(0.0, P.powlaw_exponent * (10 / log(10)) * logit2((ref_c + 0.5) / (c + 1.0), P.germ_hetero_FA));
                const i32 q = P.germ_phred_hetero_snp + (i32)round(max_(ref_like_binom, ref_like_powlaw) - (i32)round(max_(nonref_like_binom, nonref_like_powlaw)));
                if ((init_refQ == prev_q) || (abs(q - prev_q) > 10) || depths_diff(b, prev_b, 130, 3) || depths_diff(c, prev_c, 130, 3) || depths_diff(c12, prev_c12, 130, 3)) {
                    const i64 e[8] = { rp2 + ((UVC_BASE_SYMBOL == stype) ? 1 : 0), 1 + stype, INT32_MIN, b, c, c12, q, INT32_MIN };
                    v.insert(v.end(), e, e + 8);
                    prev_b = b; prev_c = c; prev_c12 = c12; prev_q = q;
                }
            }
        }
        std::string joined;
        for (i64 e : v) joined += (e == INT32_MIN ? std::string(".") : std::to_string(e)) + ",";
        const std::string vcfREF = S.refstring.substr((size_t)(refpos - S.beg), 1);
        put_line(tname + "\t" + std::to_string(refpos + 1) + "\t.\t" + vcfREF + "\t<NON_REF>\t.\t.\tMGVCF_BLOCK\tGT:VTI:POS_VT_BDP_CDP_HomRefQ\t.:" + std::to_string(sym_of(vcfREF[0])) + ",15:" + joined + std::to_string(rp2end));
    }
    const i64 x = refpos - S.beg;
    const i32 aCDP = S.p32(UVC_P_a_near_long_clip_dp, x), ADP = S.p32(UVC_P_a_dp, x);
    const bool is_in_long_track = (curr_tracklen > max_(P.microadjust_alignment_tracklen_min - 1, prev_tracklen));
    const bool is_in_clip_region = ((aCDP >= P.microadjust_alignment_clip_min_count) && (aCDP >= ADP * (P.microadjust_alignment_clip_min_frac - DBL_EPSILON)));
    if ((0x10 & P.outvar_flag) && (is_in_long_track || is_in_clip_region) && (ADP >= 2 * P.microadjust_alignment_clip_min_count)) {
        const std::string vcfREF = S.refstring.substr((size_t)(refpos - S.beg), 1);
        const i32 at = zpos - S.beg;
        const std::string ru = ((at >= 0 && at < (i32)S.refstring.size()) ? S.refstring.substr((size_t)at, (size_t)repeatunit_size) : std::string());
        put_line(tname + "\t" + std::to_string(refpos + 1) + "\t.\t" + vcfREF + "\t<ADDITIONAL_INDEL_CANDIDATE>\t.\t.\tADDITIONAL_INDEL_CANDIDATE;RU=" + ru + ";RC=" + std::to_string(repeatnum)
                 + "\tGT:VTI:clipDP\t.:" + std::to_string(sym_of(vcfREF[0])) + ",16:" + std::to_string(ADP) + "," + std::to_string(aCDP));
    }
}

int score(State &S, const UvcScoreRequest *req, std::vector<std::vector<i32>> &records, std::string &err) {
    if (!S.accumulated) { err = "score before accumulate"; return UVCGPU_ESTATE; }
    const UvcParams &P = S.P;
    const bool tprov = P.tumor_vcf_is_provided;
    const UvcTumorKey *tk = (req ? req->tumor_keys : NULL); const i64 ntk = (req ? req->n_tumor_keys : 0);
    auto pos_rescued = [&](i32 refpos) { for (i64 q = 0; q < ntk; q++) if (tk[q].refpos == refpos) return true; return false; };   // extended_posidx_to_is_rescued
    const i32 ext_beg = S.beg;
    const i32 refsize = (i32)S.refstring.size();
    i32 pos_beg = (req && req->pos_beg >= 0) ? req->pos_beg : S.beg + 1;
    i32 pos_end = (req && req->pos_beg >= 0) ? req->pos_end : S.end - 1;
    const bool all_out = (req && req->all_out) || P.should_output_all;
    const bool amplicon = (req && req->is_amplicon);
    const i32 minABQ_snv = (amplicon ? P.syserr_minABQ_pcr_snv : P.syserr_minABQ_cap_snv);
    const i32 minABQ_indel = (amplicon ? P.syserr_minABQ_pcr_indel : P.syserr_minABQ_cap_indel);
    const i32 nrtr = (i32)S.rtr.size();
    records.clear();
    std::vector<UvcGapRow> gap_rows; std::vector<u8> gap_seq;
    indel_allele_rows(S, gap_rows, gap_seq);
    std::map<std::pair<i32, int>, size_t> gap_first;   // (refpos, symbol) -> first row
    for (size_t q = gap_rows.size(); q-- > 0;) gap_first[std::make_pair(gap_rows[q].refpos, gap_rows[q].symbol)] = q;
    const bool base_at_beg = (req && req->base_at_pos_beg && req->pos_beg >= 0);   // the region continues an adjacent one (include/uvcgpu.h)
    const i32 region_beg = (req ? req->region_beg : 0);
    if (pos_beg < S.beg + (base_at_beg ? 1 : 0) || pos_end > S.end - 1 || pos_end < pos_beg) { err = "score range outside the region"; return UVCGPU_EINVAL; }
    i32 prev_tracklen = 0, curr_tracklen = 0;
    if (base_at_beg) { i32 us = 0, rn = 0; indelpos_to_context(us, rn, S.refstring, pos_beg - 1 - ext_beg, P.indel_str_repeatsize_max); prev_tracklen = rn * us; }   // what the adjacent region's last iteration left
    for (i32 zpos = pos_beg; zpos < pos_end; zpos++, prev_tracklen = curr_tracklen) {
        i32 repeatunit_size = 0, repeatnum = 0;
        indelpos_to_context(repeatunit_size, repeatnum, S.refstring, zpos - ext_beg, P.indel_str_repeatsize_max);
        curr_tracklen = repeatnum * repeatunit_size;   // main.cpp:614
        const i32 refidx = zpos - ext_beg;
        auto symat = [&](i32 i) -> int { return (int)S.refsym[i]; };
        const int st_refsymbol[2] = { ((refsize == (refidx - 1) || (-1 == (refidx - 1))) ? UVC_BASE_NN : symat(refidx - 1)), UVC_LINK_M };
        const int prev_base1 = ((refidx >= 2) ? symat(refidx - 2) : UVC_BASE_NN), prev_base2 = ((refidx >= 3) ? symat(refidx - 3) : UVC_BASE_NN);
        const int next_base1 = ((refidx < refsize) ? symat(refidx) : UVC_BASE_NN), next_base2 = ((refidx + 1 < refsize) ? symat(refidx + 1) : UVC_BASE_NN);
        std::vector<Fmt> fmts[2];
        std::vector<std::string> texts[2];
        i32 curr_vAC[2] = { 0, 0 };
        i32 ins_cdepth = 0, del_cdepth = 0, ins1_cdepth = 0, del1_cdepth = 0;
        for (int st = 0; st < 2; st++) {
            if (zpos == pos_beg && UVC_BASE_SYMBOL == st && !base_at_beg) continue;
            const i32 refpos = (UVC_BASE_SYMBOL == st ? (zpos - 1) : zpos);
            const i64 x = refpos - S.beg;
            const int refsymbol = st_refsymbol[st];
            Fmt init; memset(&init, 0, sizeof(init));
            i32 bDPcDP[2];
            symboltype_init(init, S, refpos, st, bDPcDP);
            if (S.vcf_sink && UVC_BASE_SYMBOL == st) position_lines(S, region_beg, refpos, zpos, prev_tracklen, curr_tracklen, repeatunit_size, repeatnum);
            const i32 ref_bdepth = S.FR(0, UVC_FRAG_bDP, refsymbol, x) + S.FR(1, UVC_FRAG_bDP, refsymbol, x);
            for (int k = 0; k < ST_NSYMBOLS[st]; k++) {
                const int symbol = ST_SYMBOLS[st][k];
                const i32 bdepth = S.FR(0, UVC_FRAG_bDP, symbol, x) + S.FR(1, UVC_FRAG_bDP, symbol, x);
                const i32 cdepth = max_(S.FA(0, UVC_FAM_cDP1, symbol, x), S.FA(0, UVC_FAM_cDP12, symbol, x)) + max_(S.FA(1, UVC_FAM_cDP1, symbol, x), S.FA(1, UVC_FAM_cDP12, symbol, x));
                if (is_ins(symbol)) { ins_cdepth += cdepth; if (UVC_LINK_I1 == symbol) ins1_cdepth += cdepth; }
                else if (is_del(symbol)) { del_cdepth += cdepth; if (UVC_LINK_D1 == symbol) del1_cdepth += cdepth; }
                if ((!tprov) && (((refsymbol != symbol) && (bdepth < P.min_altdp_thres)) || ((refsymbol == symbol) && (bDPcDP[0] - ref_bdepth < P.min_altdp_thres))) && (!all_out)) continue;
                if (tprov && !pos_rescued(refpos)) continue;                      // main.cpp:838-840
                // allele list: the tumor records of this (position, symbol) if any (is_var_rescued, main.cpp:806, 864-900), else the
                // host-supplied InDel alleles, else the single default allele (see UvcIndelAllele)
                std::vector<UvcIndelAllele> alleles;
                std::vector<i32> allele_rows;   // parallel to `alleles` when they were derived from the region's own allele tables
                std::vector<std::string> allele_texts;
                std::vector<const UvcTumorKey *> akeys;
                if (tprov) for (i64 q = 0; q < ntk; q++) if (tk[q].refpos == refpos && tk[q].symbol == symbol) {
                    UvcIndelAllele d = { refpos, symbol, bdepth, cdepth, (is_ins(symbol) || is_del(symbol)) ? tk[q].indel_len : 0 };
                    alleles.push_back(d); akeys.push_back(&tk[q]);
                    // the InDel string of a rescued record is the tumor record's: REF / ALT without their common head (main.cpp:867-880); without the
                    // strings (UvcScoreRequest::tumor_ref_alt == NULL, a state the reference cannot be in) the record keeps its symbolic allele
                    std::string text;
                    if ((is_ins(symbol) || is_del(symbol)) && req && req->tumor_ref_alt && req->tumor_ref_alt[q]) {
                        const std::string ra = req->tumor_ref_alt[q];
                        const size_t tab = ra.find('\t');
                        if (tab != std::string::npos) {
                            const std::string vr = ra.substr(0, tab), va = ra.substr(tab + 1);
                            if (vr.size() > va.size()) text = vr.substr(va.size()); else if (va.size() > vr.size()) text = va.substr(vr.size());
                        }
                    }
                    allele_rows.push_back(-1); allele_texts.push_back(text);
                }
                if (!alleles.empty()) {}
                else if (is_ins(symbol) || is_del(symbol)) {
                    if (req) for (i64 q = 0; q < req->n_indel_alleles; q++) if (req->indel_alleles[q].refpos == refpos && req->indel_alleles[q].symbol == symbol) alleles.push_back(req->indel_alleles[q]);
                    if (alleles.empty()) {   // fill_by_indel_info + indel_get_majority, main.cpp:853-896
                        std::vector<GapAllele> ga;
                        indel_majority(S, refpos, symbol, ga);
                        for (const GapAllele &g : ga) {
                            UvcIndelAllele d = { refpos, symbol, g.bAD1, g.cAD1, (i32)g.s.size() };
                            alleles.push_back(d);
                            i32 row = -1;
                            auto gf = gap_first.find(std::make_pair(refpos, symbol));
                            for (size_t q = (gf == gap_first.end() ? gap_rows.size() : gf->second); q < gap_rows.size() && row < 0 && gap_rows[q].refpos == refpos && gap_rows[q].symbol == symbol; q++) if (gap_rows[q].refpos == refpos && gap_rows[q].symbol == symbol && gap_rows[q].len == (i32)g.s.size()) {
                                bool same = true;
                                if (is_ins(symbol)) for (size_t c = 0; c < g.s.size(); c++) if ("ACGTN"[gap_seq[(size_t)gap_rows[q].seq_off + c]] != g.s[c]) same = false;
                                if (same) row = (i32)q;
                            }
                            allele_rows.push_back(row); allele_texts.push_back(g.s);
                        }
                    }
                } else { UvcIndelAllele d = { refpos, symbol, bdepth, cdepth, 0 }; alleles.push_back(d); }
                for (size_t ai = 0; ai < alleles.size(); ai++) {
                    const UvcIndelAllele &al = alleles[ai];
                    Fmt f = init;
                    f.refpos = refpos; f.refsymbol = refsymbol;
                    f.tpfa_dpv = f.tpfa_qual = -1.0; f.tki_tier2 = 0; f.tkey = NULL; f.tkey_idx = -1;
                    if (ai < akeys.size()) {
                        const UvcTumorKey &k = *akeys[ai];
                        f.tkey = &k; f.tkey_idx = (i32)(&k - tk);
                        f.tpfa_dpv = (double)(k.cDP1x + 1) / (double)(k.CDP1x + 2);          // main.cpp:935
                        f.tpfa_qual = (double)(k.bDP + 0.5) / (double)(k.BDP + 1.0);         // main.cpp:985-986
                        f.tki_tier2 = k.tier2;
                    }
                    const bool homopol_1bp = (prev_base1 == refsymbol && next_base1 == refsymbol);
                    const bool homopol_2bp = (prev_base2 == refsymbol && next_base2 == refsymbol);
                    const i32 minABQ = (is_subst(symbol) ? (i32)nnminus(minABQ_snv, (homopol_1bp ? (homopol_2bp ? 20 : 10) : 0)) : minABQ_indel);
                    symbol_init(f, S, refpos, symbol, al.bDPa, al.cDP0a, al.indel_len, minABQ);
                    f.gapSa_row = (ai < allele_rows.size() ? allele_rows[ai] : -1);
                    if (S.trace) trace_record(*S.trace, f, S.rtr[max_(refpos - ext_beg, 3) - 3], S.rtr[min_(refpos - ext_beg + 3, nrtr - 1)]);
                    calc_DPv(f, S.rtr[max_(refpos - ext_beg, 3) - 3], S.rtr[min_(refpos - ext_beg + 3, nrtr - 1)], refsymbol, S, refpos);
                    fmts[st].push_back(f); texts[st].push_back(ai < allele_texts.size() ? allele_texts[ai] : std::string());   // InDel string as text, for the order of cVQSM
                }
            }
        }
        for (int st = 0; st < 2; st++) {
            if (zpos == pos_beg && UVC_BASE_SYMBOL == st && !base_at_beg) continue;
            if (fmts[st].empty()) continue;
            const i32 refpos = (UVC_BASE_SYMBOL == st ? (zpos - 1) : zpos);
            // BcfFormat_symbol_sum_DPv, main.hpp:4888-4906
            i32 s1[6] = { 0 }, s2[6] = { 0 };
            for (Fmt &f : fmts[st]) {
                const i32 v[6] = { f.cDP1v, f.cDP1w, f.cDP1x, f.cDP2v, f.cDP2w, f.cDP2x };
                for (int i = 0; i < 6; i++) s1[i] += v[i];
                if (UVC_BASE_NN == f.symbol || UVC_LINK_NN == f.symbol) for (int i = 0; i < 6; i++) s2[i] = v[i];
            }
            for (Fmt &f : fmts[st]) for (int i = 0; i < 6; i++) { f.CDPv[i][0] = s1[i]; f.CDPv[i][1] = s2[i]; }
            struct Top { i32 mx, v1, v2; int symbol; std::string str; i32 row; };
            std::vector<Top> tops;
            for (size_t fi = 0; fi < fmts[st].size(); fi++) {
                Fmt &f = fmts[st][fi];
                if (S.trace2) { const double v6[6] = { (double)ins_cdepth, (double)del_cdepth, (double)ins1_cdepth, (double)del1_cdepth, (double)repeatunit_size, (double)repeatnum }; S.trace2->insert(S.trace2->end(), v6, v6 + 6); }
                calc_qual(f, ins_cdepth, del_cdepth, ins1_cdepth, del1_cdepth, repeatunit_size, repeatnum,
                          S.rtr[max_(refpos - ext_beg, 3) - 3], S.rtr[min_(refpos - ext_beg + 3, nrtr - 1)], refpos, st_refsymbol[st], S);
                if (st_refsymbol[st] != f.symbol) {   // main.cpp:990-998
                    Top t = { max_(f.cVQ1, f.cVQ2), f.cVQ1, f.cVQ2, f.symbol, texts[st][fi], f.gapSa_row }; tops.push_back(t);
                    if (max_(f.cVQ1, f.cVQ2) >= ((UVC_BASE_SYMBOL == st) ? P.germ_phred_het3al_snp : P.germ_phred_het3al_indel)) curr_vAC[st] += 1;
                }
            }
            std::sort(tops.begin(), tops.end(), [](const Top &a, const Top &b) {   // descending tuple order, main.cpp:1000
                if (a.mx != b.mx) return a.mx > b.mx;
                if (a.v1 != b.v1) return a.v1 > b.v1;
                if (a.v2 != b.v2) return a.v2 > b.v2;
                if (a.symbol != b.symbol) return a.symbol > b.symbol;
                return a.str > b.str; });
            for (Fmt &f : fmts[st]) for (int i = 0; i < 2; i++) {   // cVQ1M / cVQ2M / cVQAM / cVQSM, main.cpp:1001-1016
                const bool has = ((size_t)i < tops.size());
                f.cVQ1M[i] = (has ? tops[i].v1 : -999); f.cVQ2M[i] = (has ? tops[i].v2 : -999); f.cVQAM[i] = (has ? tops[i].symbol : SYM_END); f.cVQSM[i] = (has ? tops[i].row : -1);
            }
            i64 rec0 = (i64)records.size();
            for (int t = 0; t < st; t++) if (!(zpos == pos_beg && UVC_BASE_SYMBOL == t && !base_at_beg)) rec0 += (i64)fmts[t].size();
            call_germline(P, st_refsymbol[st], fmts[st], rec0);
            if (S.vcf_sink && fmts[st][0].germ_emit) {   // the GERMLINE line of output_germline (main.hpp:5612-5775)
                static const char *const DESC[] = { "A", "C", "G", "T", "N", "*", "<LR>", "<LD3P>", "<LD2>", "<LD1>", "<LI3P>", "<LI2>", "<LI1>", "*", "<NONE>" };
                static const char *const GT4[4] = { "0/0", "0/1", "1/1", "1/2" };
                const Fmt &f0 = fmts[st][0];
                const bool subst = (UVC_BASE_SYMBOL == st);
                const int rel[3] = { f0.g_ref_rel, f0.g_alt1_rel, f0.g_alt2_rel };
                auto sym_of = [&](int q) { return rel[q] >= 0 ? fmts[st][(size_t)rel[q]].symbol : (int)SYM_END; };
                auto cdp0a = [&](int q) { return rel[q] >= 0 ? fmts[st][(size_t)rel[q]].cDP0a : 0; };
                auto allele_text = [&](int symbol, int q) -> std::string {
                    int seen = 0;
                    for (size_t j = 0; j < fmts[st].size(); j++) if (fmts[st][j].symbol == symbol) { if (seen++ == q) return texts[st][j].empty() ? std::string(DESC[symbol]) : texts[st][j]; }
                    return std::string();
                };
                const i32 regionpos = refpos - ext_beg;
                const int s0 = sym_of(0), s1 = sym_of(1), s2 = sym_of(2), GLidx = f0.germ_GT;
                std::string vref, valt;
                if (subst) { vref = DESC[s0]; valt = DESC[s1]; if (3 == GLidx) valt += std::string(",") + DESC[s2]; }
                else {
                    const std::string vref1 = (regionpos > 0 ? S.refstring.substr((size_t)regionpos - 1, 1) : std::string("n"));
                    const std::string str1 = (s1 < (int)SYM_END ? allele_text(s1, 0) : std::string());
                    vref = vref1;
                    if (3 != GLidx) {
                        if (str1.empty() || str1[0] == '<') valt = DESC[s1];
                        else { valt = vref; if (is_ins(s1)) valt += str1; else if (is_del(s1)) vref += str1; else valt = DESC[s1]; }
                    } else {
                        const std::string str2 = (s2 < (int)SYM_END ? allele_text(s2, s2 == s1 ? 1 : 0) : std::string());
                        valt = vref1;
                        if (str1.empty() || str1[0] == '<' || str2.empty() || str2[0] == '<') valt = std::string(DESC[s1]) + "," + DESC[s2];
                        else if (is_ins(s1) && is_ins(s2)) valt = vref1 + str1 + "," + vref1 + str2;
                        else if (is_del(s1) && is_del(s2)) {
                            if (str1.size() > str2.size()) { vref = vref1 + str1; valt = vref1 + "," + vref1 + str1.substr(str2.size()); }
                            else { vref = vref1 + str2; valt = vref1 + str2.substr(str1.size()) + "," + vref1; }
                        }
                        else if (is_ins(s1) && is_del(s2)) { valt = vref1 + str1 + str2 + "," + vref1; vref = vref1 + str2; }
                        else if (is_del(s1) && is_ins(s2)) { valt = vref1 + "," + vref1 + str2 + str1; vref = vref1 + str1; }
                        else valt = std::string(DESC[s1]) + "," + DESC[s2];
                    }
                }
                const int nn = (subst ? UVC_BASE_NN : UVC_LINK_NN);
                const i64 x = refpos - S.beg;
                std::string l = S.vcf_sink->tname + "\t" + std::to_string(refpos + (subst ? 1 : 0)) + "\t.\t" + vref + "\t" + valt + "\t" + std::to_string(f0.germ_GQ) + "\tPASS\tGERMLINE\tGT:GQ:HQ:FT:CDP1:cDP1:GL4:GST:note\t"
                    + GT4[GLidx] + ":" + std::to_string(f0.germ_GQ) + ":0,0:PASS:" + std::to_string(f0.DP) + "," + std::to_string(2 * S.FA(0, UVC_FAM_cDP1, nn, x)) + ":"
                    + std::to_string(cdp0a(0)) + "," + std::to_string(cdp0a(1)) + (3 == GLidx ? "," + std::to_string(cdp0a(2)) : std::string()) + ":";
                for (int q = 0; q < 4; q++) l += (q ? "," : "") + std::to_string(f0.GL4[q]);
                l += ":";
                for (int q = 0; q < 8; q++) l += (q ? "," : "") + std::to_string(f0.GST[q]);
                l += ":";
                S.vcf_sink->fixed.push_back(l); S.vcf_sink->spec.push_back(std::string()); S.vcf_sink->tier2.push_back(-1);
            }
        }
        const bool is_germline_var_generated = ((!fmts[0].empty() && fmts[0][0].germ_emit) || (!fmts[1].empty() && fmts[1][0].germ_emit));
        for (int st = 0; st < 2; st++) {   // main.cpp:1073-1168
            if (zpos == pos_beg && UVC_BASE_SYMBOL == st && !base_at_beg) continue;
            const Fmt *reffmt = NULL;
            for (const Fmt &f : fmts[st]) if (f.symbol == st_refsymbol[st]) reffmt = &f;
            for (Fmt &f : fmts[st]) {
                f.vAC[0] = curr_vAC[0]; f.vAC[1] = curr_vAC[1];
                if (!reffmt) { err = "a scored position has no REF allele"; return UVCGPU_ESTATE; }   // the reference aborts, main.cpp:1025-1029
                call_record(P, f, *reffmt, fmts[st], st, st_refsymbol[st], (all_out || is_germline_var_generated));
                std::vector<i32> r; emit(f, r); records.push_back(r);
                if (S.vcf_sink && f.keep && f.out) vcf_emit(S, f, *reffmt, texts[st][(size_t)(&f - &fmts[st][0])], gap_rows, gap_seq, zpos);
            }
        }
    }
    return 0;
}

}  // namespace uvco
