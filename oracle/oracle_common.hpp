// TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
//
// oracle/: a scalar CPU restatement of the reference's hot path, written from scratch over the
// SoA boundary types of include/uvcgpu.h.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it.  The product (uvc_amd/csrc, libuvcgpu.so) never links or calls it.
//
// PARITY STATUS: "parity unpinned" for everything except
//   * the parameter defaults (pinned against the reference's own CmdLineArgs.hpp, compiled as-is:
//     oracle/ref_params_dump.cpp -> tests/golden/params_default.json), and
//   * calc_binom_10log10_likeratio / prob2odds / odds2prob, pinned by the reference's own
//     known-answer static_asserts (main_conversion.hpp:205-209, 251-254) in tests/test_oracle_math.py, and
//   * the layout of the VCF sample column, which is not restated at all: oracle/ref_vcf_driver.cpp streams the oracle's values through
//     the reference's own generated bcfrec::streamAppendBcfFormat (oracle/_ref/, built by `make ref_formats` from bcf_formats_generator1.cpp).
// The reference's hot-path headers #include htslib, which is absent here and may not be
// stood-in for, so the reference itself cannot be compiled in this environment.
#ifndef UVC_ORACLE_COMMON_HPP
#define UVC_ORACLE_COMMON_HPP

#include "uvcgpu.h"

#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <map>
#include <string>
#include <vector>
#include <algorithm>

namespace uvco {

typedef int32_t i32;
typedef int64_t i64;
typedef uint32_t u32;
typedef uint8_t u8;

static const int NSYM = UVC_NUM_SYMBOLS;
static const int NBUCKETS = 16;            // NUM_BUCKETS, main_conversion.hpp:920
static const int SQR_QUAL_DIV_ = 32;       // SQR_QUAL_DIV, main_conversion.hpp:20
static const int MAX_INSERT_SIZE_ = 2000;  // common.hpp:64
static const int MAX_STR_N_BASES_ = 100;   // common.hpp:63

// BAM CIGAR op codes (SAM/BAM spec section 4.2)
enum { C_MATCH = 0, C_INS = 1, C_DEL = 2, C_REF_SKIP = 3, C_SOFT_CLIP = 4, C_HARD_CLIP = 5, C_PAD = 6, C_EQUAL = 7, C_DIFF = 8, C_BACK = 9 };

template <class T> static inline T min_(T a, T b) { return a < b ? a : b; }
template <class T> static inline T max_(T a, T b) { return a > b ? a : b; }
template <class T> static inline T between_(T v, T a, T b) { return min_(max_(a, v), b); }  // BETWEEN, main_conversion.hpp:124-128
static inline i64 nnminus(i64 a, i64 b) { return a > b ? a - b : 0; }                       // non_neg_minus, common.hpp:195-200
static inline double nnminus_d(double a, double b) { return a > b ? a - b : 0.0; }

static inline bool is_ins(int s) { return s == UVC_LINK_I3P || s == UVC_LINK_I2 || s == UVC_LINK_I1; }   // main_conversion.hpp:417-424
static inline bool is_del(int s) { return s == UVC_LINK_D3P || s == UVC_LINK_D2 || s == UVC_LINK_D1; }   // main_conversion.hpp:426-433
static inline bool is_subst(int s) { return s >= UVC_BASE_A && s <= UVC_BASE_NN; }                        // isSymbolSubstitution, :463-466
static inline int ins_len_to_symbol(int len) { return len == 1 ? UVC_LINK_I1 : (len == 2 ? UVC_LINK_I2 : UVC_LINK_I3P); }  // :435-440
static inline int del_len_to_symbol(int len) { return len == 1 ? UVC_LINK_D1 : (len == 2 ? UVC_LINK_D2 : UVC_LINK_D3P); }  // :442-447
// areSymbolsMutated, main_conversion.hpp:364-371
static inline bool symbols_mutated(int ref, int alt) {
    if (alt <= UVC_BASE_NN) return ref != alt && ref < UVC_BASE_N && alt < UVC_BASE_N;
    return alt != UVC_LINK_M && alt != UVC_LINK_NN;
}
// symbol ranges per type: BASE = [BASE_A, BASE_NN], LINK = [LINK_M, LINK_NN]  (main_conversion.hpp:388-410)
static inline int st_beg(int st) { return st == UVC_BASE_SYMBOL ? UVC_BASE_A : UVC_LINK_M; }
static inline int st_end(int st) { return st == UVC_BASE_SYMBOL ? UVC_BASE_NN : UVC_LINK_NN; }
// SYMBOL_TYPE_TO_SYMBOLS iteration order (main_conversion.hpp:397-400)
static const int ST_SYMBOLS[2][8] = {
    { UVC_BASE_A, UVC_BASE_C, UVC_BASE_G, UVC_BASE_T, UVC_BASE_N, UVC_BASE_NN, -1, -1 },
    { UVC_LINK_M, UVC_LINK_I1, UVC_LINK_I2, UVC_LINK_I3P, UVC_LINK_D1, UVC_LINK_D2, UVC_LINK_D3P, UVC_LINK_NN } };
static const int ST_NSYMBOLS[2] = { 6, 8 };

// phred helpers, common.hpp:81-87
static inline double phred2nat(double x) { return (log(10.0) / 10.0) * x; }
static inline double numstates2phred(double x) { return (10.0 / log(10.0)) * log(x); }
static inline i32 numstates2deciphred(double x) { return (i32)round((100.0 / log(10.0)) * log(x)); }

struct Rtr { i32 begpos = 0, tracklen = 0, unitlen = 0, indelphred = 43, anyTR_begpos = 0, anyTR_tracklen = 0, anyTR_unitlen = 0; };  // common.hpp:150-160

// one alignment record (view into the SoA)
struct Aln {
    i32 pos, endpos, mpos, isize, flag, mapq, nm, l_qseq, n_cigar;
    const u8 *bases; const u8 *quals; const u32 *cigar;
    i32 frag, fam, strand;
    bool isrc() const { return (flag & 0x10) != 0; }
    int bam_strand() const { return ((flag & 0x81) == 0x81) ? ((flag & 0x20) ? 1 : 0) : ((flag & 0x10) ? 1 : 0); }  // bam_get_strand, common.hpp:89
};
static inline int cig_op(u32 c) { return (int)(c & 0xF); }
static inline u32 cig_len(u32 c) { return c >> 4; }

// nested alns3 view: family -> strand -> fragment -> alignment
struct Frag { int aln_beg, aln_end; };
struct FamStrand { int frag_beg, frag_end; };
struct Family { FamStrand fs[2]; int dflag; };

// dense per-position state, same plane layout as the uvcgpu fetch groups
struct VcfSink { std::string tname; std::vector<std::string> fixed, spec; std::vector<int> tier2; };   // text of the written records (vcf_emit, oracle_score.cpp)

struct State {
    VcfSink *vcf_sink = nullptr;
    std::vector<double> *trace2 = nullptr;  // ... and the six position-level arguments of calc_qual (ins / del depths, repeat unit size and count), per record
    std::vector<double> *trace = nullptr;   // test hook (uvc_oracle_score_trace): the BcfFormat inputs of every record in front of calc_DPv / calc_qual
    i32 tid, beg, end;      // state covers [beg, end): end = caller's `end` + 1 (main.cpp:569)
    i64 npos;
    std::vector<u8> refsym;            // region_symbolvec (string2symbolseq, main_conversion.hpp:531-539), npos-1 entries + 1 pad
    std::string refstring;
    std::vector<Rtr> rtr;              // npos entries (refstring.size()+1, main.hpp:872)
    std::vector<i64> baq, baq2;        // npos entries each
    std::vector<i32> prep32, thres, seg32, vq, bqsum, frag, fam, faminfo32, duplex;
    std::vector<i64> prep64, seg64, faminfo64;
    // transient bucket histograms dedup_ampDistr[2] (main.hpp:2377)
    std::vector<i32> bucket[2];
    UvcParams P;
    std::vector<Aln> alns;
    std::vector<Frag> frags;
    std::vector<Family> fams;
    std::vector<u8> bases, quals; std::vector<u32> cigars;
    bool accumulated = false;
    // InDel allele counters beside the depth planes: index I1/D1 = 0, I2/D2 = 1, I3P/D3P = 2 (main.hpp:574-583)
    struct GapMaps {
        std::map<i32, std::map<std::string, i32>> iseq[3];
        std::map<i32, std::map<i32, i32>> dlen[3];
        void clear() { for (int i = 0; i < 3; i++) { iseq[i].clear(); dlen[i].clear(); } }
    };
    // [strand]: symbol_to_frag_format_depth_sets / symbol_to_fam_format_depth_sets_2strand side maps (main.hpp:529-530),
    // pos2iseq2data_cDP2 / pos2dlen2data_cDP2, pos2iseq2data_c2dDP / pos2dlen2data_c2dDP (main.hpp:2380-2383)
    GapMaps gap_frag[2], gap_fam[2], gap_c2[2], gap_c2d[2];
    // haplotype links (a12): the mutated (position, symbol) strings of fragments / families -> [forward, reverse] counts
    // (mutform2count4map_bq / _fq / _f2q, main.hpp:3685-3687) and what updateHapMap makes of them (main.hpp:3596-3663)
    typedef std::vector<std::pair<i32, int>> MutForm;
    struct HapLink { MutForm form; i32 fr[2]; i32 other[2]; };
    std::map<MutForm, std::array<i32, 2>> hapmap[3];     // 0 bq, 1 fq, 2 f2q
    std::vector<HapLink> haplinks[3];

    // Internal storage is position-major (AoS, like the reference's std::vector<struct> per kind, main.hpp:523-604) so that
    // the per-read scatter touches one or two cache lines per position; fetch() transposes to the plane layout of uvcgpu.h.
    inline i32 &p32(int f, i64 i) { return prep32[(size_t)i * UVC_NPREP32 + f]; }
    inline i64 &p64(int f, i64 i) { return prep64[(size_t)i * UVC_NPREP64 + f]; }
    inline i32 &th(int f, i64 i) { return thres[(size_t)i * UVC_NTHRES + f]; }
    inline i32 &s32(int f, int s, i64 i) { return seg32[((size_t)i * NSYM + s) * UVC_NSEG32 + f]; }
    inline i64 &s64(int f, int s, i64 i) { return seg64[((size_t)i * NSYM + s) * UVC_NSEG64 + f]; }
    inline i32 &VQ(int f, int s, i64 i) { return vq[((size_t)i * NSYM + s) * UVC_NVQ + f]; }
    inline i32 &BQS(int s, i64 i) { return bqsum[(size_t)i * NSYM + s]; }
    inline i32 &FR(int strand, int f, int s, i64 i) { return frag[(((size_t)i * 2 + strand) * NSYM + s) * UVC_NFRAG + f]; }
    inline i32 &FA(int strand, int f, int s, i64 i) { return fam[(((size_t)i * 2 + strand) * NSYM + s) * UVC_NFAM + f]; }
    inline i32 &FI(int f, int s, i64 i) { return faminfo32[((size_t)i * NSYM + s) * UVC_NFAMINFO32 + f]; }
    inline i64 &FI64(int f, int s, i64 i) { return faminfo64[((size_t)i * NSYM + s) * UVC_NFAMINFO64 + f]; }
    inline i32 &DU(int f, int s, i64 i) { return duplex[((size_t)i * NSYM + s) * UVC_NDUPLEX + f]; }
    inline i32 &BK(int strand, int s, int b, i64 i) { return bucket[strand][((size_t)i * NSYM + s) * NBUCKETS + b]; }
    inline i32 seg_ad(int s, i64 i) {  // seg_format_get_ad, main_conversion.hpp:785-789
        return s32(UVC_S_aDPff, s, i) + s32(UVC_S_aDPfr, s, i) + s32(UVC_S_aDPrf, s, i) + s32(UVC_S_aDPrr, s, i);
    }
};

// side arrays (C10)
void build_side_arrays(State &S);
// accumulate passes
int accumulate(State &S, std::string &err);
std::string hap_phase_string(const State &S, int which, i32 refpos, int symbol);
// InDel allele rows (fill_by_indel_info, instcode.hpp) in the order documented at UvcGapRow
void indel_allele_rows(State &S, std::vector<UvcGapRow> &rows, std::vector<u8> &seq);
// scoring
int score(State &S, const UvcScoreRequest *req, std::vector<std::vector<i32>> &records, std::string &err);

// math primitives exported for unit tests
double calc_binom_10log10_likeratio(double prob, double a, double b, bool bidirectional = false, bool set_max_prob_to_one = false);
void dp4_to_pcFA(double out[2], bool bidirectional, bool overseq_disabled, double overseq_frac, double aADpass, double aADfail, double aDPpass, double aDPfail,
                 double pl_exponent, double n_nats, double aADavgKeyVal = -1, double aDPavgKeyVal = -1, double priorAD = 0.5, double priorDP = 1.0);
void infer_max_qual_assuming_independence(i32 &maxvqual, i32 &argmaxAD, i32 &argmaxBQ, i32 max_qual, i32 dec_qual, const i32 *qual_distr, i32 totDP);
i32 indel_phred(double ampfact, i32 repeatsize_at_max_repeatnum, i32 max_repeatnum);
i32 indel_len_rusize_phred(i32 indel_len, i32 repeatunit_size);
bool is_indel_context_more_STR(i32 rulen1, i32 rc1, i32 rulen2, i32 rc2, i32 indel_str_repeatsize_max);
void indelpos_to_context(i32 &repeatunit_len, i32 &max_repeatnum, const std::string &refstring, i32 refpos, i32 indel_str_repeatsize_max);

}  // namespace uvco
#endif
