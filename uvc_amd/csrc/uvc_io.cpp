// uvc_io.cpp -- libuvcio.so: BGZF / BAM / BAI and FASTA / .fai readers on zlib (include/uvcio.h).
// Written against the SAM/BAM specification (SAMv1.pdf: section 4.1 BGZF, 4.2 BAM, 5.1-5.3 BAI binning scheme); no htslib.
// Host code only: it feeds the family assignment (include/uvcgroup.h) and uvcgpu_region_set_reads.
#include "uvcio.h"
#include "uvcgpu.h"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
extern "C" const char *uvcio_last_error(void) { return g_err.c_str(); }

namespace {

inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t le64(const uint8_t *p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

// ---- BGZF: a series of gzip members, each with a 'BC' extra subfield that holds the member size (SAMv1 4.1) ----
struct Bgzf {
    FILE *fp = nullptr;
    int64_t block_addr = -1;        // file offset of the block in `buf`
    int64_t next_addr = 0;          // file offset of the next block
    std::vector<uint8_t> buf;       // inflated block
    size_t off = 0;                 // read position inside buf
    bool eof = false;

    bool load(int64_t addr) {       // reads and inflates the block at file offset addr
        if (fseeko(fp, (off_t)addr, SEEK_SET) != 0) return false;
        uint8_t h[18];
        const size_t got = fread(h, 1, 18, fp);
        if (got == 0) { eof = true; buf.clear(); off = 0; block_addr = addr; next_addr = addr; return true; }
        if (got < 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return false;
        const int xlen = le16(h + 10);
        std::vector<uint8_t> extra((size_t)xlen);
        memcpy(extra.data(), h + 12, std::min(6, xlen));
        if (xlen > 6 && fread(extra.data() + 6, 1, (size_t)xlen - 6, fp) != (size_t)xlen - 6) return false;
        int bsize = -1;
        for (int i = 0; i + 4 <= xlen;) {
            const int slen = le16(extra.data() + i + 2);
            if (extra[i] == 'B' && extra[i + 1] == 'C' && slen == 2) bsize = le16(extra.data() + i + 4);
            i += 4 + slen;
        }
        if (bsize < 0) return false;
        const int cdata = bsize - xlen - 19;   // BSIZE = total block size - 1
        if (cdata < 0) return false;
        if (xlen < 6 && fseeko(fp, (off_t)(addr + 12 + xlen), SEEK_SET) != 0) return false;
        std::vector<uint8_t> comp((size_t)cdata + 8);
        if (fread(comp.data(), 1, comp.size(), fp) != comp.size()) return false;
        const uint32_t isize = le32(comp.data() + cdata + 4);
        buf.assign(isize, 0);
        if (isize) {
            z_stream zs; memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) return false;
            zs.next_in = comp.data(); zs.avail_in = (uInt)cdata; zs.next_out = buf.data(); zs.avail_out = isize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END || zs.total_out != isize) return false;
            if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), buf.data(), isize) != le32(comp.data() + cdata)) return false;
        }
        block_addr = addr; next_addr = addr + bsize + 1; off = 0; eof = false;
        return true;
    }
    bool seek(uint64_t voffset) {   // virtual file offset: coffset << 16 | uoffset
        const int64_t c = (int64_t)(voffset >> 16); const size_t u = (size_t)(voffset & 0xFFFF);
        if (c != block_addr && !load(c)) return false;
        if (u > buf.size()) return false;
        off = u;
        return true;
    }
    uint64_t tell() const { return off < buf.size() || buf.empty() ? (((uint64_t)block_addr << 16) | off) : ((uint64_t)next_addr << 16); }
    // reads n bytes across block boundaries; returns the number read (less than n only at the end of the file)
    size_t read(void *dst, size_t n) {
        size_t done = 0;
        while (done < n) {
            if (off >= buf.size()) {
                if (eof) break;
                if (!load(next_addr)) { eof = true; break; }
                if (buf.empty()) { if (eof) break; continue; }   // an empty block (e.g. the EOF marker): keep going
            }
            const size_t k = std::min(n - done, buf.size() - off);
            memcpy((uint8_t *)dst + done, buf.data() + off, k);
            off += k; done += k;
        }
        return done;
    }
};

struct Chunk { uint64_t beg, end; };
struct RefIndex { std::map<uint32_t, std::vector<Chunk>> bins; std::vector<uint64_t> linear; };

// SAMv1 5.3: bins that may hold alignments overlapping [beg, end)
void reg2bins(int64_t beg, int64_t end, std::vector<uint32_t> &list) {
    --end;
    list.push_back(0);
    for (int64_t k = 1 + (beg >> 26); k <= 1 + (end >> 26); ++k) list.push_back((uint32_t)k);
    for (int64_t k = 9 + (beg >> 23); k <= 9 + (end >> 23); ++k) list.push_back((uint32_t)k);
    for (int64_t k = 73 + (beg >> 20); k <= 73 + (end >> 20); ++k) list.push_back((uint32_t)k);
    for (int64_t k = 585 + (beg >> 17); k <= 585 + (end >> 17); ++k) list.push_back((uint32_t)k);
    for (int64_t k = 4681 + (beg >> 14); k <= 4681 + (end >> 14); ++k) list.push_back((uint32_t)k);
}

}  // namespace

struct uvcio_bam {
    Bgzf z;
    std::vector<std::string> ref_names; std::vector<int64_t> ref_lens;
    uint64_t first_record = 0;      // virtual offset behind the header
    bool has_index = false;
    std::vector<RefIndex> idx;
    // batch storage
    std::vector<int32_t> tid, pos, endpos, mtid, mpos, isize, nm, l_qseq, n_cigar;
    std::vector<uint16_t> flag; std::vector<uint8_t> mapq;
    std::vector<int64_t> seq_off, cigar_off, qname_off;
    std::vector<uint8_t> bases, quals; std::vector<uint32_t> cigars; std::vector<char> qnames;
    std::vector<uint8_t> rec;
};

static bool load_bai(uvcio_bam *b, const std::string &path) {
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) return false;
    std::vector<uint8_t> d;
    { uint8_t tmp[1 << 16]; size_t k; while ((k = fread(tmp, 1, sizeof(tmp), fp)) > 0) d.insert(d.end(), tmp, tmp + k); }
    fclose(fp);
    size_t o = 0;
    auto need = [&](size_t n) { return o + n <= d.size(); };
    if (!need(8) || memcmp(d.data(), "BAI\1", 4) != 0) return false;
    const uint32_t n_ref = le32(d.data() + 4); o = 8;
    b->idx.assign(n_ref, RefIndex());
    for (uint32_t r = 0; r < n_ref; r++) {
        if (!need(4)) return false;
        const uint32_t n_bin = le32(d.data() + o); o += 4;
        for (uint32_t i = 0; i < n_bin; i++) {
            if (!need(8)) return false;
            const uint32_t bin = le32(d.data() + o), n_chunk = le32(d.data() + o + 4); o += 8;
            if (!need((size_t)n_chunk * 16)) return false;
            std::vector<Chunk> &v = b->idx[r].bins[bin];
            for (uint32_t c = 0; c < n_chunk; c++) { v.push_back(Chunk{ le64(d.data() + o), le64(d.data() + o + 8) }); o += 16; }
        }
        if (!need(4)) return false;
        const uint32_t n_intv = le32(d.data() + o); o += 4;
        if (!need((size_t)n_intv * 8)) return false;
        for (uint32_t i = 0; i < n_intv; i++) { b->idx[r].linear.push_back(le64(d.data() + o)); o += 8; }
    }
    return true;
}

extern "C" int uvcio_bam_open(uvcio_bam_t **out, const char *path) {
    if (!out || !path) return fail(UVCGPU_EINVAL, "null argument");
    uvcio_bam *b = new uvcio_bam();
    b->z.fp = fopen(path, "rb");
    if (!b->z.fp) { delete b; return fail(UVCGPU_EINVAL, std::string("cannot open ") + path); }
    if (!b->z.load(0) || b->z.eof) { fclose(b->z.fp); delete b; return fail(UVCGPU_EINVAL, std::string(path) + " is not a BGZF file"); }
    uint8_t h[8];
    auto bad = [&](const char *why) { fclose(b->z.fp); delete b; return fail(UVCGPU_EINVAL, std::string(path) + ": " + why); };
    if (b->z.read(h, 8) != 8 || memcmp(h, "BAM\1", 4) != 0) return bad("not a BAM file");
    const uint32_t l_text = le32(h + 4);
    std::vector<uint8_t> text(l_text);
    if (b->z.read(text.data(), l_text) != l_text) return bad("truncated header");
    if (b->z.read(h, 4) != 4) return bad("truncated header");
    const uint32_t n_ref = le32(h);
    for (uint32_t i = 0; i < n_ref; i++) {
        if (b->z.read(h, 4) != 4) return bad("truncated reference list");
        const uint32_t l_name = le32(h);
        std::vector<char> nm(l_name);
        if (b->z.read(nm.data(), l_name) != l_name || b->z.read(h, 4) != 4) return bad("truncated reference list");
        b->ref_names.push_back(std::string(nm.data(), l_name ? l_name - 1 : 0)); b->ref_lens.push_back((int64_t)le32(h));
    }
    b->first_record = b->z.tell();
    std::string p(path);
    b->has_index = load_bai(b, p + ".bai");
    if (!b->has_index && p.size() > 4 && p.substr(p.size() - 4) == ".bam") b->has_index = load_bai(b, p.substr(0, p.size() - 4) + ".bai");
    if (b->has_index && b->idx.size() != b->ref_names.size()) { b->has_index = false; b->idx.clear(); }
    *out = b;
    return 0;
}
extern "C" int32_t uvcio_bam_n_refs(const uvcio_bam_t *b) { return b ? (int32_t)b->ref_names.size() : 0; }
extern "C" const char *uvcio_bam_ref_name(const uvcio_bam_t *b, int32_t tid) { return (b && tid >= 0 && tid < (int32_t)b->ref_names.size()) ? b->ref_names[tid].c_str() : nullptr; }
extern "C" int64_t uvcio_bam_ref_len(const uvcio_bam_t *b, int32_t tid) { return (b && tid >= 0 && tid < (int32_t)b->ref_lens.size()) ? b->ref_lens[tid] : -1; }
extern "C" int uvcio_bam_has_index(const uvcio_bam_t *b) { return b && b->has_index; }
extern "C" void uvcio_bam_close(uvcio_bam_t *b) { if (!b) return; if (b->z.fp) fclose(b->z.fp); delete b; }

// reads the alignment at the current position into b->rec; 1 = got one, 0 = end of file, < 0 = error
static int next_record(uvcio_bam *b) {
    uint8_t h[4];
    const size_t got = b->z.read(h, 4);
    if (got == 0) return 0;
    if (got != 4) return fail(UVCGPU_EINVAL, "truncated BAM record");
    const uint32_t bs = le32(h);
    if (bs < 32 || bs > (1u << 28)) return fail(UVCGPU_EINVAL, "implausible BAM record size");
    b->rec.resize(bs);
    if (b->z.read(b->rec.data(), bs) != bs) return fail(UVCGPU_EINVAL, "truncated BAM record");
    return 1;
}

// appends b->rec to the batch when it overlaps [beg, end) of tid; *past = the record starts at or behind `end` (or on a later reference)
static int take_record(uvcio_bam *b, int32_t want_tid, int64_t beg, int64_t end, bool *past) {
    static const uint8_t nt16_int[16] = { 4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4 };   // seq_nt16_int of htslib: =ACMGRSVTWYHKDBN
    const uint8_t *r = b->rec.data();
    const size_t bs = b->rec.size();
    const int32_t tid = (int32_t)le32(r), pos = (int32_t)le32(r + 4);
    const int l_name = r[8], mq = r[9];
    const int n_cig = le16(r + 12), flg = le16(r + 14);
    const int32_t l_seq = (int32_t)le32(r + 16), mtid = (int32_t)le32(r + 20), mpos = (int32_t)le32(r + 24), tlen = (int32_t)le32(r + 28);
    *past = (tid > want_tid || tid < 0 || (tid == want_tid && pos >= end));
    if (tid != want_tid) return 0;
    const size_t o_cig = 32 + (size_t)l_name, o_seq = o_cig + 4 * (size_t)n_cig, o_qual = o_seq + ((size_t)l_seq + 1) / 2, o_aux = o_qual + (size_t)l_seq;
    if (l_seq < 0 || o_aux > bs) return fail(UVCGPU_EINVAL, "corrupt BAM record");
    int64_t e = pos;
    for (int k = 0; k < n_cig; k++) {
        const uint32_t c = le32(r + o_cig + 4 * (size_t)k); const int op = (int)(c & 0xF);
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) e += (int64_t)(c >> 4);   // M D N = X consume the reference
    }
    if (e == pos) e = pos + 1;
    if (!(pos < end && e > beg)) return 0;
    // NM:  aux = tag[2] type value ...
    int32_t nm = -1;
    for (size_t o = o_aux; o + 3 <= bs;) {
        const char t0 = (char)r[o], t1 = (char)r[o + 1], ty = (char)r[o + 2];
        o += 3;
        size_t sz = 0; long long val = 0; bool is_int = true;
        switch (ty) {
            case 'A': sz = 1; is_int = false; break;
            case 'c': sz = 1; if (o + 1 <= bs) val = (int8_t)r[o]; break;
            case 'C': sz = 1; if (o + 1 <= bs) val = r[o]; break;
            case 's': sz = 2; if (o + 2 <= bs) val = (int16_t)le16(r + o); break;
            case 'S': sz = 2; if (o + 2 <= bs) val = le16(r + o); break;
            case 'i': sz = 4; if (o + 4 <= bs) val = (int32_t)le32(r + o); break;
            case 'I': sz = 4; if (o + 4 <= bs) val = le32(r + o); break;
            case 'f': sz = 4; is_int = false; break;
            case 'Z': case 'H': { size_t k = o; while (k < bs && r[k]) k++; sz = k - o + 1; is_int = false; break; }
            case 'B': {
                if (o + 5 > bs) return fail(UVCGPU_EINVAL, "corrupt aux array");
                const char sub = (char)r[o]; const uint32_t cnt = le32(r + o + 1);
                const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                sz = 5 + es * (size_t)cnt; is_int = false; break;
            }
            default: return fail(UVCGPU_EINVAL, "unknown aux type in BAM record");
        }
        if (o + sz > bs) return fail(UVCGPU_EINVAL, "corrupt aux field");
        if (t0 == 'N' && t1 == 'M' && is_int) nm = (int32_t)val;
        o += sz;
    }
    b->tid.push_back(tid); b->pos.push_back(pos); b->endpos.push_back((int32_t)e); b->mtid.push_back(mtid); b->mpos.push_back(mpos); b->isize.push_back(tlen);
    b->flag.push_back((uint16_t)flg); b->mapq.push_back((uint8_t)mq); b->nm.push_back(nm); b->l_qseq.push_back(l_seq); b->n_cigar.push_back(n_cig);
    b->seq_off.push_back((int64_t)b->bases.size()); b->cigar_off.push_back((int64_t)b->cigars.size()); b->qname_off.push_back((int64_t)b->qnames.size());
    for (int k = 0; k < n_cig; k++) b->cigars.push_back(le32(r + o_cig + 4 * (size_t)k));
    for (int32_t i = 0; i < l_seq; i++) { const uint8_t by = r[o_seq + (size_t)(i >> 1)]; b->bases.push_back(nt16_int[(i & 1) ? (by & 0xF) : (by >> 4)]); }
    b->quals.insert(b->quals.end(), r + o_qual, r + o_qual + l_seq);
    b->qnames.insert(b->qnames.end(), (const char *)r + 32, (const char *)r + 32 + l_name);
    if (l_name == 0 || r[32 + l_name - 1] != 0) b->qnames.push_back('\0');
    return 0;
}

extern "C" int uvcio_bam_fetch(uvcio_bam_t *b, int32_t tid, int64_t beg, int64_t end, UvcBamBatch *out) {
    if (!b || !out) return fail(UVCGPU_EINVAL, "null argument");
    if (tid < 0 || tid >= (int32_t)b->ref_names.size()) return fail(UVCGPU_EINVAL, "tid out of range");
    if (beg < 0) beg = 0;
    if (end > b->ref_lens[tid]) end = b->ref_lens[tid];
    b->tid.clear(); b->pos.clear(); b->endpos.clear(); b->mtid.clear(); b->mpos.clear(); b->isize.clear(); b->nm.clear(); b->l_qseq.clear(); b->n_cigar.clear();
    b->flag.clear(); b->mapq.clear(); b->seq_off.clear(); b->cigar_off.clear(); b->qname_off.clear(); b->bases.clear(); b->quals.clear(); b->cigars.clear(); b->qnames.clear();
    int rc = 0;
    if (end > beg) {
        std::vector<Chunk> chunks;
        if (b->has_index) {
            const RefIndex &ri = b->idx[tid];
            const size_t w = (size_t)(beg >> 14);
            const uint64_t min_off = ri.linear.empty() ? 0 : ri.linear[std::min(w, ri.linear.size() - 1)];
            std::vector<uint32_t> bins; reg2bins(beg, end, bins);
            for (uint32_t bin : bins) { auto it = ri.bins.find(bin); if (it != ri.bins.end()) for (const Chunk &c : it->second) if (c.end > min_off) chunks.push_back(Chunk{ std::max(c.beg, min_off), c.end }); }
            std::sort(chunks.begin(), chunks.end(), [](const Chunk &a, const Chunk &c) { return a.beg < c.beg; });
            std::vector<Chunk> merged;
            for (const Chunk &c : chunks) { if (!merged.empty() && c.beg <= merged.back().end) merged.back().end = std::max(merged.back().end, c.end); else merged.push_back(c); }
            chunks.swap(merged);
        } else chunks.push_back(Chunk{ b->first_record, ~0ull });
        bool past = false;
        for (const Chunk &c : chunks) {
            if (past) break;
            if (!b->z.seek(c.beg)) return fail(UVCGPU_EINVAL, "bad virtual offset in the index");
            while (b->z.tell() < c.end) {
                rc = next_record(b);
                if (rc <= 0) break;
                rc = take_record(b, tid, beg, end, &past);
                if (rc < 0 || past) break;
            }
            if (rc < 0) return rc;
        }
    }
    out->n_alns = (int64_t)b->pos.size();
    out->tid = b->tid.data(); out->pos = b->pos.data(); out->endpos = b->endpos.data(); out->mtid = b->mtid.data(); out->mpos = b->mpos.data(); out->isize = b->isize.data();
    out->flag = b->flag.data(); out->mapq = b->mapq.data(); out->nm = b->nm.data(); out->l_qseq = b->l_qseq.data(); out->n_cigar = b->n_cigar.data();
    out->seq_off = b->seq_off.data(); out->cigar_off = b->cigar_off.data(); out->qname_off = b->qname_off.data();
    out->n_bases = (int64_t)b->bases.size(); out->bases = b->bases.data(); out->quals = b->quals.data();
    out->n_cigar_ops = (int64_t)b->cigars.size(); out->cigars = b->cigars.data(); out->n_qname_bytes = (int64_t)b->qnames.size(); out->qnames = b->qnames.data();
    return 0;
}

// ---------------------------------------------------------------- FASTA + .fai ----------------
struct FaiEntry { int64_t len, offset, linebases, linewidth; };
struct uvcio_fasta { FILE *fp = nullptr; std::map<std::string, FaiEntry> seqs; };

extern "C" int uvcio_fasta_open(uvcio_fasta_t **out, const char *path) {
    if (!out || !path) return fail(UVCGPU_EINVAL, "null argument");
    FILE *fi = fopen((std::string(path) + ".fai").c_str(), "r");
    if (!fi) return fail(UVCGPU_EINVAL, std::string("cannot open ") + path + ".fai (samtools faidx layout: name, length, offset, linebases, linewidth)");
    uvcio_fasta *f = new uvcio_fasta();
    char name[4096]; long long a, o, lb, lw;
    char line[8192];
    while (fgets(line, sizeof(line), fi)) if (sscanf(line, "%4095s %lld %lld %lld %lld", name, &a, &o, &lb, &lw) == 5 && lb > 0 && lw >= lb) f->seqs[name] = FaiEntry{ a, o, lb, lw };
    fclose(fi);
    f->fp = fopen(path, "rb");
    if (!f->fp || f->seqs.empty()) { if (f->fp) fclose(f->fp); delete f; return fail(UVCGPU_EINVAL, std::string("cannot open ") + path + " or empty .fai"); }
    *out = f;
    return 0;
}
extern "C" int64_t uvcio_fasta_seq_len(const uvcio_fasta_t *f, const char *name) {
    if (!f || !name) return -1;
    auto it = f->seqs.find(name);
    return it == f->seqs.end() ? -1 : it->second.len;
}
extern "C" int uvcio_fasta_fetch(uvcio_fasta_t *f, const char *name, int64_t beg, int64_t end, char *dst) {
    if (!f || !name || !dst) return fail(UVCGPU_EINVAL, "null argument");
    auto it = f->seqs.find(name);
    if (it == f->seqs.end()) return fail(UVCGPU_EINVAL, std::string("sequence not in the .fai: ") + name);
    const FaiEntry &e = it->second;
    if (beg < 0 || end > e.len || end < beg) return fail(UVCGPU_EINVAL, "FASTA range outside the sequence");
    int64_t i = beg;
    while (i < end) {
        const int64_t line = i / e.linebases, col = i % e.linebases;
        const int64_t k = std::min(end - i, e.linebases - col);
        if (fseeko(f->fp, (off_t)(e.offset + line * e.linewidth + col), SEEK_SET) != 0 || fread(dst + (i - beg), 1, (size_t)k, f->fp) != (size_t)k) return fail(UVCGPU_EINVAL, "short read from the FASTA file");
        i += k;
    }
    for (int64_t k = 0; k < end - beg; k++) if (dst[k] >= 'a' && dst[k] <= 'z') dst[k] = (char)(dst[k] - 32);
    return 0;
}
extern "C" void uvcio_fasta_close(uvcio_fasta_t *f) { if (!f) return; if (f->fp) fclose(f->fp); delete f; }
