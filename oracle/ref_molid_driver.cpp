// TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
// C wrapper around the reference's own MolecularBarcode (MolecularID.hpp / MolecularID.cpp) and Hash.hpp, which compile without htslib.
// Built by `make -C oracle ref_molid` with -I/root/reference from the sources where they lie, into oracle/_ref/libref_molid.so:
// everything reached through the functions below is REFERENCE code (createKey, operator<, calcHash, strnhash, strhash, hash2hash).
// tests/test_ref_molid.py checks the oracle's (and through it the HIP library's) hashes and family-key equivalence against it.
#include "MolecularID.hpp"
#include "Hash.hpp"

#include <cstdint>
#include <cstring>

static MolecularBarcode make(int begtid, int beg, int endtid, int end, const char *qname, const char *umi, int dflag, int idflag) {
    MolecularBarcode mb;
    mb.beg_tidpos_pair = std::make_pair((uvc1_refgpos_t)begtid, (uvc1_refgpos_t)beg);
    mb.end_tidpos_pair = std::make_pair((uvc1_refgpos_t)endtid, (uvc1_refgpos_t)end);
    mb.qnamestring = qname;
    mb.umistring = umi;
    mb.duplexflag = (uvc1_flag_t)dflag;
    mb.dedup_idflag = (uvc1_flag_t)idflag;
    MolecularBarcode key = mb.createKey();          // grouping.cpp:937-938
    key.hashvalue = key.calcHash();
    return key;
}

extern "C" {
uint64_t ref_strnhash(const char *s, size_t n, uint64_t base) { return (uint64_t)strnhash(s, n, (uvc1_hash_t)base); }
uint64_t ref_strhash(const char *s, uint64_t base) { return (uint64_t)strhash(s, (uvc1_hash_t)base); }
uint64_t ref_hash2hash(uint64_t a, uint64_t b) { return (uint64_t)hash2hash((uvc1_hash_t)a, (uvc1_hash_t)b); }

// the key createKey() makes of a barcode: out4 = beg (tid, pos), end (tid, pos); *qlen / *ulen = length of the strings kept in the key;
// returns calcHash() of the key
uint64_t ref_molid_key(int begtid, int beg, int endtid, int end, const char *qname, const char *umi, int dflag, int idflag, int32_t *out4, int32_t *qlen, int32_t *ulen) {
    const MolecularBarcode k = make(begtid, beg, endtid, end, qname, umi, dflag, idflag);
    out4[0] = (int32_t)k.beg_tidpos_pair.first; out4[1] = (int32_t)k.beg_tidpos_pair.second;
    out4[2] = (int32_t)k.end_tidpos_pair.first; out4[3] = (int32_t)k.end_tidpos_pair.second;
    *qlen = (int32_t)k.qnamestring.size(); *ulen = (int32_t)k.umistring.size();
    return (uint64_t)k.hashvalue;
}
// MolecularBarcode::operator< between the keys of two barcodes (std::map order of umi_to_strand_to_reads, grouping.cpp:939)
int ref_molid_less(int begtid1, int beg1, int endtid1, int end1, const char *qname1, const char *umi1, int dflag1, int idflag1,
                   int begtid2, int beg2, int endtid2, int end2, const char *qname2, const char *umi2, int dflag2, int idflag2) {
    return make(begtid1, beg1, endtid1, end1, qname1, umi1, dflag1, idflag1) < make(begtid2, beg2, endtid2, end2, qname2, umi2, dflag2, idflag2) ? 1 : 0;
}
}
