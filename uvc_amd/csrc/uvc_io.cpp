// uvc_io.cpp -- libuvcio.so: BGZF / BAM / BAI and FASTA / .fai readers on zlib (include/uvcio.h).
// Written against the SAM/BAM specification (SAMv1.pdf: section 4.1 BGZF, 4.2 BAM, 5.1-5.3 BAI binning scheme); no htslib.
// Host code only: it feeds the family assignment (include/uvcgroup.h) and uvcgpu_region_set_reads.
#include "uvcio.h"
#include "uvcgpu.h"
#include "uvc_cpus.h"
#include "uvc_inflate_fast.h"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <vector>

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
extern "C" const char *uvcio_last_error(void) { return g_err.c_str(); }
namespace { uint32_t block_crc32(const uint8_t *p, size_t n); }
// test entry: one raw DEFLATE stream of known output size through the fast decoder (1 = decoded, 0 = declined)
extern "C" int uvcio_inflate_raw_fast(const void *in, int64_t in_len, void *out, int64_t out_len) {
    return (in && out && in_len >= 0 && out_len >= 0 && uvc_fast_inflate::inflate((const uint8_t *)in, (size_t)in_len, (uint8_t *)out, (size_t)out_len)) ? 1 : 0;
}
namespace { extern void *(*g_col_alloc)(size_t); extern void (*g_col_free)(void *); }
namespace { uvcio_inflate_fn g_inflate_fn = nullptr; void *g_inflate_ctx = nullptr; int32_t g_inflate_min = 0; }
extern "C" void uvcio_set_inflate(uvcio_inflate_fn fn, void *ctx, int32_t min_blocks) { g_inflate_fn = fn; g_inflate_ctx = ctx; g_inflate_min = min_blocks; }
extern "C" void uvcio_set_column_allocator(void *(*alloc_fn)(size_t), void (*free_fn)(void *)) { g_col_alloc = alloc_fn; g_col_free = free_fn; }
extern "C" uint32_t uvcio_crc32(const void *p, int64_t n) { return (p && n > 0) ? block_crc32((const uint8_t *)p, (size_t)n) : 0u; }

namespace {
uint32_t block_crc32(const uint8_t *p, size_t n);   // CRC-32 of a BGZF block (carry-less multiplication where the CPU has it), defined below

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
            if (block_crc32(buf.data(), isize) != le32(comp.data() + cdata)) return false;
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

// grow-only byte buffer without value initialisation (a fresh std::vector of tens of MB costs its page faults on every query)
struct RawBuf {
    uint8_t *p = nullptr; size_t cap = 0;
    // nullptr when the memory is not there (the old block stays valid and is freed by the destructor)
    uint8_t *need(size_t n) { if (n > cap) { const size_t c2 = n + n / 4 + 4096; uint8_t *q = (uint8_t *)realloc(p, c2); if (!q) return nullptr; p = q; cap = c2; } return p; }
    ~RawBuf() { free(p); }
};
// The two large columns of a batch (one byte per base each) live in memory from the caller's allocator when one is set
// (uvcio_set_column_allocator: uvc1-mi355x hands in uvcgpu_host_alloc, so that uvcgpu_region_set_reads copies them at DMA speed instead of
// through the runtime's pageable staging path, which serialises the worker threads).  Grow-only, no value initialisation, contents kept.
void *(*g_col_alloc)(size_t) = nullptr; void (*g_col_free)(void *) = nullptr;
struct ColBuf {
    uint8_t *p = nullptr; size_t n = 0, cap = 0; bool hooked = false;
    uint8_t *data() { return p; }
    size_t size() const { return n; }
    void clear() { n = 0; }
    void resize(size_t m) {
        if (m > cap) {
            const size_t ncap = m + m / 4 + 65536;
            const bool use_hook = (g_col_alloc && g_col_free);
            uint8_t *q = (uint8_t *)(use_hook ? g_col_alloc(ncap) : malloc(ncap));
            if (!q) throw std::bad_alloc();
            if (n) memcpy(q, p, n);
            release();
            p = q; cap = ncap; hooked = use_hook;
        }
        n = m;
    }
    void release() { if (p) { if (hooked) g_col_free(p); else free(p); } p = nullptr; cap = 0; }
    ~ColBuf() { release(); }
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
    ColBuf bases, quals; std::vector<uint32_t> cigars; std::vector<char> qnames;
    RawBuf comp, infl;             // compressed / inflated bytes of the current batch (kept between queries)
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

// ---- region query: compressed blocks are read in batches, inflated in parallel (BGZF blocks are independent), the records of a
// batch are located by one sequential walk over their size fields and decoded in parallel into the output columns ----
namespace {

struct BlockRef { int64_t addr; uint32_t csize /* whole block */, isize; size_t in_off /* in the batch's compressed buffer */, out_off /* in its inflated buffer */; };
struct RecRef { const uint8_t *r; uint32_t size; int32_t endpos; int64_t base_off, cig_off, name_off; };

int n_threads() {
    static const int n = [] { const char *e = getenv("UVCIO_THREADS"); int v = e ? atoi(e) : uvc_effective_cpus(); return std::max(1, std::min(v, 64)); }();
    return n;
}
// One pool of helper threads for the whole process (UVCIO_THREADS of them, default: the cores the process may use, uvc_cpus.h).
// The command line keeps several tiles in flight, one reader each: with a pool per call every reader owned cores/readers threads and
// those cores idled while its worker was in a single-threaded stage; with one queue the slices of every reader go to whichever core is
// free.  The calling thread works on the queue too until its own slices are done, so a call never waits for a busy pool (and a
// process that forked after the pool was made still finishes, on its own thread).
struct SliceGroup { std::mutex m; std::condition_variable cv; size_t left = 0; };
struct Slice { const std::function<void(size_t, size_t)> *f; size_t a, b; SliceGroup *g; };
struct Pool {
    std::mutex m; std::condition_variable cv; std::deque<Slice> q;
    explicit Pool(int n) { for (int i = 0; i < n; i++) std::thread([this] { for (;;) { Slice s; { std::unique_lock<std::mutex> l(m); cv.wait(l, [this] { return !q.empty(); }); s = q.front(); q.pop_front(); } run(s); } }).detach(); }
    static void run(const Slice &s) { (*s.f)(s.a, s.b); std::lock_guard<std::mutex> l(s.g->m); if (--s.g->left == 0) s.g->cv.notify_all(); }
    bool try_pop(Slice &s) { std::lock_guard<std::mutex> l(m); if (q.empty()) return false; s = q.front(); q.pop_front(); return true; }
};
Pool *pool() { static Pool *p = new Pool(std::max(0, n_threads() - 1)); return p; }   // never destroyed: its threads sleep on it until the process ends
template <class F> void parallel_for(size_t n, F f) {   // f(first, last) on contiguous slices
    const size_t nt = std::min<size_t>((size_t)n_threads() * 2, std::max<size_t>(n / 64, 1));
    if (nt <= 1 || n_threads() <= 1) { f((size_t)0, n); return; }
    const std::function<void(size_t, size_t)> fn = f;
    SliceGroup g; g.left = nt;
    Pool *P = pool();
    { std::lock_guard<std::mutex> l(P->m); for (size_t t = 1; t < nt; t++) P->q.push_back(Slice{ &fn, n * t / nt, n * (t + 1) / nt, &g }); }
    P->cv.notify_all();
    Pool::run(Slice{ &fn, 0, n / nt, &g });
    for (;;) {
        { std::lock_guard<std::mutex> l(g.m); if (g.left == 0) break; }
        Slice s;
        if (P->try_pop(s)) { Pool::run(s); continue; }   // ours or another reader's: either way a core does useful work
        std::unique_lock<std::mutex> l(g.m); g.cv.wait(l, [&] { return g.left == 0; }); break;
    }
}
// CRC-32 of a block with carry-less multiplication (the folding scheme of Intel's "Fast CRC computation using PCLMULQDQ", constants
// for the reflected polynomial 0xEDB88320): zlib 1.2.11's table walk does 1.1 GB/s, a quarter of the time of inflating a BGZF block.
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_fold(const uint8_t *buf, size_t len, uint32_t crc) {   // len >= 64, multiple of 16; crc = ~running value
    alignas(16) static const uint64_t k1k2[2] = { 0x0154442bd4ULL, 0x01c6e41596ULL };
    alignas(16) static const uint64_t k3k4[2] = { 0x01751997d0ULL, 0x00ccaa009eULL };
    alignas(16) static const uint64_t k5k0[2] = { 0x0163cd6124ULL, 0x0000000000ULL };
    alignas(16) static const uint64_t poly[2] = { 0x01db710641ULL, 0x01f7011641ULL };
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i *)k1k2);
    buf += 64; len -= 64;
    while (len >= 64) {   // four lanes of 16 bytes folded over 64 bytes
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00); x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11); x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = _mm_load_si128((const __m128i *)k3k4);   // the four lanes into one
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);   // 128 -> 64 bits
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8); x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i *)k5k0);
    x2 = _mm_srli_si128(x1, 4); x1 = _mm_and_si128(x1, x3); x1 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i *)poly);   // Barrett reduction to 32 bits
    x2 = _mm_and_si128(x1, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x10); x2 = _mm_and_si128(x2, x3); x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
static const bool g_have_pclmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
#else
static const bool g_have_pclmul = false;
static uint32_t crc32_fold(const uint8_t *, size_t, uint32_t c) { return c; }
#endif
uint32_t block_crc32(const uint8_t *p, size_t n) {
    if (!g_have_pclmul || n < 64) return (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n);
    const size_t body = n & ~(size_t)15;
    const uint32_t c = ~crc32_fold(p, body, ~0u);
    return (n > body) ? (uint32_t)crc32(c, p + body, (uInt)(n - body)) : c;
}
const bool g_fast_inflate = (getenv("UVCIO_ZLIB") == nullptr);   // UVCIO_ZLIB=1: every block through zlib (A/B)
bool inflate_block(const uint8_t *blk, uint32_t csize, uint8_t *dst, uint32_t isize) {
    const int xlen = le16(blk + 10);
    const uint8_t *cdata = blk + 12 + xlen;
    const int clen = (int)csize - xlen - 20;
    if (clen < 0) return false;
    if (isize == 0) return true;
    // the decoder of uvc_inflate_fast.h first; zlib looks at whatever it declines or gets wrong (the CRC-32 of the block decides)
    if (g_fast_inflate && uvc_fast_inflate::inflate(cdata, (size_t)clen, dst, isize) && block_crc32(dst, isize) == le32(blk + csize - 8)) return true;
    z_stream zs; memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    zs.next_in = const_cast<uint8_t *>(cdata); zs.avail_in = (uInt)clen; zs.next_out = dst; zs.avail_out = isize;
    const int rc = inflate(&zs, Z_FINISH);
    inflateEnd(&zs);
    return rc == Z_STREAM_END && zs.total_out == isize && block_crc32(dst, isize) == le32(blk + csize - 8);
}
const uint8_t NT16_INT[16] = { 4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4 };   // seq_nt16_int of htslib: =ACMGRSVTWYHKDBN

// NM aux tag of the record at r (bam_aux_get + bam_aux2i), -1 if absent; false on a corrupt aux area
bool aux_nm(const uint8_t *r, size_t bs, size_t o, int32_t &nm) {
    nm = -1;
    while (o + 3 <= bs) {
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
                if (o + 5 > bs) return false;
                const char sub = (char)r[o]; const uint32_t cnt = le32(r + o + 1);
                const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                sz = 5 + es * (size_t)cnt; is_int = false; break;
            }
            default: return false;
        }
        if (o + sz > bs) return false;
        if (t0 == 'N' && t1 == 'M' && is_int) nm = (int32_t)val;
        o += sz;
    }
    return true;
}

}  // namespace

static int bam_fetch_impl(uvcio_bam_t *b, int32_t tid, int64_t beg, int64_t end, UvcBamBatch *out);
// no exception crosses the C boundary: a column that cannot grow is UVCGPU_ENOMEM
extern "C" int uvcio_bam_fetch(uvcio_bam_t *b, int32_t tid, int64_t beg, int64_t end, UvcBamBatch *out) {
    try { return bam_fetch_impl(b, tid, beg, end, out); }
    catch (const std::bad_alloc &) { return fail(UVCGPU_ENOMEM, "uvcio_bam_fetch: out of host memory"); }
    catch (const std::exception &e) { return fail(UVCGPU_EINVAL, std::string("uvcio_bam_fetch: ") + e.what()); }
}
static int bam_fetch_impl(uvcio_bam_t *b, int32_t tid, int64_t beg, int64_t end, UvcBamBatch *out) {
    if (!b || !out) return fail(UVCGPU_EINVAL, "null argument");
    if (tid < 0 || tid >= (int32_t)b->ref_names.size()) return fail(UVCGPU_EINVAL, "tid out of range");
    if (beg < 0) beg = 0;
    if (end > b->ref_lens[tid]) end = b->ref_lens[tid];
    b->tid.clear(); b->pos.clear(); b->endpos.clear(); b->mtid.clear(); b->mpos.clear(); b->isize.clear(); b->nm.clear(); b->l_qseq.clear(); b->n_cigar.clear();
    b->flag.clear(); b->mapq.clear(); b->seq_off.clear(); b->cigar_off.clear(); b->qname_off.clear(); b->bases.clear(); b->quals.clear(); b->cigars.clear(); b->qnames.clear();
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
            // chunks that continue in the block behind the previous one are read as one run (an index written record by record ends every
            // chunk at its block's end; what lies between is at most one block of records, which the overlap test drops again)
            for (const Chunk &c : chunks) { if (!merged.empty() && (int64_t)(c.beg >> 16) <= (int64_t)(merged.back().end >> 16) + 65536) merged.back().end = std::max(merged.back().end, c.end); else merged.push_back(c); }
            chunks.swap(merged);
        } else chunks.push_back(Chunk{ b->first_record, ~0ull });
        // compressed bytes per batch (UVCIO_BATCH_BYTES: tests use small batches to exercise records that straddle two)
        const char *be = getenv("UVCIO_BATCH_BYTES");
        const size_t BATCH = (be && atol(be) >= (1 << 16)) ? (size_t)atol(be) : ((size_t)64 << 20);
        const bool timing = (getenv("UVCIO_TIMING") != nullptr);
        double t_read = 0, t_inf = 0, t_walk = 0, t_dec = 0;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        std::vector<uint8_t> carry;
        int64_t file_size = -1;
        if (fseeko(b->z.fp, 0, SEEK_END) == 0) file_size = (int64_t)ftello(b->z.fp);
        std::vector<BlockRef> blocks;
        std::vector<RecRef> recs;
        bool past = false;
        for (const Chunk &c : chunks) {
            if (past) break;
            int64_t addr = (int64_t)(c.beg >> 16);
            size_t skip = (size_t)(c.beg & 0xFFFF);            // bytes of the first block that lie in front of the chunk
            const int64_t last_addr = (c.end == ~0ull ? INT64_MAX : (int64_t)(c.end >> 16));
            const size_t last_u = (size_t)(c.end & 0xFFFF);
            carry.clear();
            bool chunk_done = false;
            while (!chunk_done && !past) {
                // 1. read whole blocks of this batch
                double t0 = now();
                if (fseeko(b->z.fp, (off_t)addr, SEEK_SET) != 0) return fail(UVCGPU_EINVAL, "seek failed");
                // no more than the chunk (plus its last block) or the rest of the file
                size_t want = BATCH + (1 << 16);
                if (last_addr != INT64_MAX) want = std::min<size_t>(want, (size_t)(last_addr - addr) + (1 << 16) + 18);
                if (file_size >= 0) want = std::min<size_t>(want, (size_t)std::max<int64_t>(file_size - addr, 0));
                uint8_t *const cbuf = b->comp.need(std::max<size_t>(want, 1));
                if (!cbuf) return fail(UVCGPU_ENOMEM, "uvcio_bam_fetch: no memory for the compressed batch");
                const size_t got = fread(cbuf, 1, want, b->z.fp);
                const bool at_eof = (file_size >= 0 ? (addr + (int64_t)got >= file_size) : (got < want));
                blocks.clear();
                size_t o = 0, out_bytes = 0;
                bool hit_last = false;
                while (o + 18 <= got && !hit_last) {
                    const uint8_t *h = cbuf + o;
                    if (h[0] != 31 || h[1] != 139 || !(h[3] & 4)) return fail(UVCGPU_EINVAL, "not a BGZF block where the index points");
                    const int xlen = le16(h + 10);
                    if (o + 12 + (size_t)xlen > got) break;
                    int bsize = -1;
                    for (int i = 0; i + 4 <= xlen;) { const int slen = le16(h + 12 + i + 2); if (h[12 + i] == 'B' && h[12 + i + 1] == 'C' && slen == 2) bsize = le16(h + 12 + i + 4); i += 4 + slen; }
                    if (bsize < 0) return fail(UVCGPU_EINVAL, "BGZF block without a BC field");
                    const size_t csize = (size_t)bsize + 1;
                    if (o + csize > got) break;                 // incomplete block: next batch starts here
                    const uint32_t isize = le32(h + csize - 4);
                    blocks.push_back(BlockRef{ addr + (int64_t)o, (uint32_t)csize, isize, o, out_bytes });
                    out_bytes += isize; o += csize;
                    if (addr + (int64_t)(o - csize) >= last_addr) hit_last = true;   // the block that holds the chunk end
                    if (o >= BATCH) break;
                }
                if (blocks.empty()) { if (got >= 18 && o == 0 && !at_eof && got == want) return fail(UVCGPU_EINVAL, "BGZF block larger than the batch"); chunk_done = true; break; }
                t_read += now() - t0; t0 = now();
                // 2. inflate in parallel
                const size_t infl_size = carry.size() + out_bytes;
                uint8_t *const ibuf = b->infl.need(std::max<size_t>(infl_size, 1));
                if (!ibuf) return fail(UVCGPU_ENOMEM, "uvcio_bam_fetch: no memory for the inflated batch");
                if (!carry.empty()) memcpy(ibuf, carry.data(), carry.size());
                std::atomic<bool> ok{ true }; bool done = false;   // (flags written from the pool's threads are atomics)
                if (g_inflate_fn && (int64_t)blocks.size() >= (int64_t)g_inflate_min) {   // somewhere else (the device); the CRC-32 of every block is checked here
                    const size_t nb2 = blocks.size();
                    std::vector<int64_t> in_off(nb2), out_off(nb2); std::vector<int32_t> in_len(nb2), out_len(nb2);
                    bool sane = true;
                    for (size_t i = 0; i < nb2; i++) {
                        const uint8_t *blk = cbuf + blocks[i].in_off;
                        const int xlen = le16(blk + 10), clen = (int)blocks[i].csize - xlen - 20;
                        if (clen < 0) { sane = false; break; }
                        in_off[i] = (int64_t)blocks[i].in_off + 12 + xlen; in_len[i] = clen; out_off[i] = (int64_t)(carry.size() + blocks[i].out_off); out_len[i] = (int32_t)blocks[i].isize;
                    }
                    if (sane && g_inflate_fn(g_inflate_ctx, cbuf, (int64_t)o, in_off.data(), in_len.data(), out_off.data(), out_len.data(), (int64_t)nb2, ibuf, (int64_t)infl_size) == 0) {
                        std::atomic<bool> crc_ok{ true };
                        parallel_for(nb2, [&](size_t i0, size_t i1) { for (size_t i = i0; i < i1; i++) if (blocks[i].isize && block_crc32(ibuf + out_off[i], blocks[i].isize) != le32(cbuf + blocks[i].in_off + blocks[i].csize - 8)) crc_ok = false; });
                        done = crc_ok;
                    }
                }
                if (!done) parallel_for(blocks.size(), [&](size_t i0, size_t i1) { for (size_t i = i0; i < i1; i++) if (!inflate_block(cbuf + blocks[i].in_off, blocks[i].csize, ibuf + carry.size() + blocks[i].out_off, blocks[i].isize)) ok = false; });
                if (!ok) return fail(UVCGPU_EINVAL, "corrupt BGZF block (inflate / CRC)");
                t_inf += now() - t0; t0 = now();
                // the usable byte range of this batch
                size_t lo = (carry.empty() ? skip : 0), hi = infl_size;
                skip = 0;
                if (hit_last) { const BlockRef &L = blocks.back(); if (L.addr == last_addr) hi = carry.size() + L.out_off + std::min<size_t>(last_u, L.isize); chunk_done = true; }
                if (at_eof && o >= got - std::min<size_t>(got, 17)) chunk_done = true;   // end of file
                // 3. Which records overlap, where their outputs go.  The chain of record sizes is one dependent load per record through
                // memory that other cores have just written (~100 ns each: as long as the parallel inflate when walked by one thread).
                // htslib starts a new BGZF block rather than let a record straddle two (bgzf_flush_try in bam_write1), so in the files it
                // wrote every block begins with a record: the blocks are walked in parallel, twice (sizes, then offsets).  A block whose
                // walk does not end on its last byte disproves that for this file, and the batch is walked the sequential way.
                recs.clear();
                size_t p = lo;
                int64_t nb = (int64_t)b->bases.size(), nc = (int64_t)b->cigars.size(), nq = (int64_t)b->qnames.size();
                // one record at ibuf + q: 0 = behind the query (stop), 1 = skip, 2 = keep; sizes of a kept record in *e .. *ln
                auto classify = [&](size_t q, uint32_t bs, int32_t *e_out, int32_t *lseq, int32_t *ncig, int32_t *ln, bool *bad) -> int {
                    const uint8_t *r = ibuf + q + 4;
                    const int32_t rt = (int32_t)le32(r), rp = (int32_t)le32(r + 4);
                    const int l_name = r[8], n_cig = le16(r + 12); const int32_t l_seq = (int32_t)le32(r + 16);
                    if (rt > tid || rt < 0 || (rt == tid && rp >= end)) return 0;
                    if (rt != tid) return 1;
                    const size_t o_cig = 32 + (size_t)l_name;
                    if (l_seq < 0 || o_cig + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq > bs) { *bad = true; return 0; }
                    int64_t e = rp;
                    for (int k = 0; k < n_cig; k++) { const uint32_t cg = le32(r + o_cig + 4 * (size_t)k); const int op = (int)(cg & 0xF); if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) e += (int64_t)(cg >> 4); }
                    if (e == rp) e = rp + 1;
                    if (!(rp < end && e > beg)) return 1;
                    *e_out = (int32_t)e; *lseq = l_seq; *ncig = n_cig; *ln = l_name + ((l_name == 0 || r[32 + l_name - 1] != 0) ? 1 : 0);
                    return 2;
                };
                bool walked = false;
                if (carry.empty() && blocks.size() >= 4 && n_threads() > 1 && !getenv("UVCIO_SERIAL_WALK")) {
                    struct BlockWalk { size_t stop; int64_t n, sb, sc, sq; bool aligned, past, bad, size_bad; };
                    const int32_t n_refs = (int32_t)b->ref_names.size();
                    std::vector<BlockWalk> bw(blocks.size());
                    auto walk_block = [&](size_t k, RecRef *dst, int64_t ob, int64_t oc, int64_t oq) {   // dst == nullptr: count only
                        const size_t s0 = std::max(blocks[k].out_off, lo), e0 = std::min(blocks[k].out_off + (size_t)blocks[k].isize, hi);
                        BlockWalk w = { s0, 0, 0, 0, 0, true, false, false, false };
                        size_t q = s0;
                        while (q + 4 <= e0) {
                            const uint32_t bs = le32(ibuf + q);
                            if (bs < 32 || bs > (1u << 28)) { w.size_bad = true; break; }
                            if (q + 4 + bs > e0) break;
                            {   // bytes that only look like a record (a block that does begin inside one) must not pass: what every record has
                                const uint8_t *r = ibuf + q + 4;
                                const int32_t rt = (int32_t)le32(r), rp = (int32_t)le32(r + 4); const int l_name = r[8];
                                if (rt < -1 || rt >= n_refs || rp < -1 || l_name < 1 || 32 + (uint32_t)l_name > bs || r[32 + l_name - 1] != 0) { w.size_bad = true; break; }
                            }
                            int32_t e = 0, ls = 0, ncg = 0, ln = 0;
                            const int c = classify(q, bs, &e, &ls, &ncg, &ln, &w.bad);
                            if (c == 0) { w.past = !w.bad; break; }
                            if (c == 2) {
                                if (dst) dst[w.n] = RecRef{ ibuf + q + 4, bs, e, ob + w.sb, oc + w.sc, oq + w.sq };
                                w.n++; w.sb += ls; w.sc += ncg; w.sq += ln;
                            }
                            q += 4 + bs;
                        }
                        w.stop = q;
                        // the walk must end on the block's last byte (the last block of the query may end at the chunk's end instead)
                        w.aligned = (w.past || w.bad || w.size_bad || q == e0 || s0 >= e0);
                        if (!dst) bw[k] = w;
                    };
                    parallel_for(blocks.size(), [&](size_t k0, size_t k1) { for (size_t k = k0; k < k1; k++) walk_block(k, nullptr, 0, 0, 0); });
                    bool all_aligned = true;
                    for (size_t k = 0; k + 1 < blocks.size(); k++) if (!bw[k].aligned) { all_aligned = false; break; }   // (the last block may hold the first part of a record)
                    for (const BlockWalk &w : bw) if (w.size_bad || w.bad) all_aligned = false;   // not records there: the sequential walk decides what the file is
                    if (all_aligned) {
                        // up to the first block that reaches the end of the query
                        size_t k_end = blocks.size();
                        for (size_t k = 0; k < blocks.size(); k++) {
                            if (bw[k].past) { k_end = k + 1; past = true; break; }
                        }
                        std::vector<int64_t> off_n(k_end + 1, 0), off_b(k_end + 1, nb), off_c(k_end + 1, nc), off_q(k_end + 1, nq);
                        for (size_t k = 0; k < k_end; k++) { off_n[k + 1] = off_n[k] + bw[k].n; off_b[k + 1] = off_b[k] + bw[k].sb; off_c[k + 1] = off_c[k] + bw[k].sc; off_q[k + 1] = off_q[k] + bw[k].sq; }
                        recs.resize((size_t)off_n[k_end]);
                        parallel_for(k_end, [&](size_t k0, size_t k1) { for (size_t k = k0; k < k1; k++) walk_block(k, recs.data() + off_n[k], off_b[k], off_c[k], off_q[k]); });
                        nb = off_b[k_end]; nc = off_c[k_end]; nq = off_q[k_end];
                        p = bw[k_end - 1].stop;
                        walked = true;
                    }
                }
                if (!walked) while (p + 4 <= hi) {
                    const uint32_t bs = le32(ibuf + p);
                    if (bs < 32 || bs > (1u << 28)) return fail(UVCGPU_EINVAL, "implausible BAM record size");
                    if (p + 4 + bs > hi) break;
                    int32_t e = 0, ls = 0, ncg = 0, ln = 0; bool bad = false;
                    const int c = classify(p, bs, &e, &ls, &ncg, &ln, &bad);
                    if (bad) return fail(UVCGPU_EINVAL, "corrupt BAM record");
                    if (c == 0) { past = true; break; }
                    if (c == 2) { recs.push_back(RecRef{ ibuf + p + 4, bs, e, nb, nc, nq }); nb += ls; nc += ncg; nq += ln; }
                    p += 4 + bs;
                }
                t_walk += now() - t0; t0 = now();
                // 4. decode in parallel
                const size_t r0 = b->pos.size(), nr = recs.size();
                b->tid.resize(r0 + nr); b->pos.resize(r0 + nr); b->endpos.resize(r0 + nr); b->mtid.resize(r0 + nr); b->mpos.resize(r0 + nr); b->isize.resize(r0 + nr); b->nm.resize(r0 + nr);
                b->l_qseq.resize(r0 + nr); b->n_cigar.resize(r0 + nr); b->flag.resize(r0 + nr); b->mapq.resize(r0 + nr); b->seq_off.resize(r0 + nr); b->cigar_off.resize(r0 + nr); b->qname_off.resize(r0 + nr);
                b->bases.resize((size_t)nb); b->quals.resize((size_t)nb); b->cigars.resize((size_t)nc); b->qnames.resize((size_t)nq);
                std::atomic<bool> aux_ok{ true };
                parallel_for(nr, [&](size_t i0, size_t i1) {
                    for (size_t i = i0; i < i1; i++) {
                        const RecRef &q = recs[i]; const uint8_t *r = q.r; const size_t k = r0 + i;
                        const int l_name = r[8], n_cig = le16(r + 12); const int32_t l_seq = (int32_t)le32(r + 16);
                        const size_t o_cig = 32 + (size_t)l_name, o_seq = o_cig + 4 * (size_t)n_cig, o_qual = o_seq + ((size_t)l_seq + 1) / 2, o_aux = o_qual + (size_t)l_seq;
                        b->tid[k] = (int32_t)le32(r); b->pos[k] = (int32_t)le32(r + 4); b->endpos[k] = q.endpos; b->mapq[k] = r[9]; b->flag[k] = le16(r + 14);
                        b->mtid[k] = (int32_t)le32(r + 20); b->mpos[k] = (int32_t)le32(r + 24); b->isize[k] = (int32_t)le32(r + 28);
                        b->l_qseq[k] = l_seq; b->n_cigar[k] = n_cig; b->seq_off[k] = q.base_off; b->cigar_off[k] = q.cig_off; b->qname_off[k] = q.name_off;
                        int32_t nm; if (!aux_nm(r, q.size, o_aux, nm)) aux_ok = false;
                        b->nm[k] = nm;
                        for (int c = 0; c < n_cig; c++) b->cigars[(size_t)q.cig_off + c] = le32(r + o_cig + 4 * (size_t)c);
                        uint8_t *bd = b->bases.data() + q.base_off;
                        for (int32_t j = 0; j < l_seq; j++) { const uint8_t by = r[o_seq + (size_t)(j >> 1)]; bd[j] = NT16_INT[(j & 1) ? (by & 0xF) : (by >> 4)]; }
                        memcpy(b->quals.data() + q.base_off, r + o_qual, (size_t)l_seq);
                        memcpy(b->qnames.data() + q.name_off, r + 32, (size_t)l_name);
                        if (l_name == 0 || r[32 + l_name - 1] != 0) b->qnames[(size_t)q.name_off + l_name] = '\0';
                    }
                });
                if (!aux_ok) return fail(UVCGPU_EINVAL, "corrupt aux fields in a BAM record");
                t_dec += now() - t0;
                // 5. what is left of a record that continues in the next batch
                if (!chunk_done && !past) { std::vector<uint8_t> rest(ibuf + p, ibuf + hi); carry.swap(rest); addr = blocks.back().addr + blocks.back().csize; }
                else if (!past && p < hi && hit_last && blocks.back().addr == last_addr) { /* the chunk ends inside a record only in a broken index */ }
            }
        }
        if (timing) fprintf(stderr, "[uvcio fetch] threads %d: read+headers %.3f s, inflate %.3f s, record walk %.3f s, decode %.3f s\n", n_threads(), t_read, t_inf, t_walk, t_dec);
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

// ---- BGZF writer (SAMv1 section 4.1): what the reference writes its VCF through (bgzf_open / bgzf_write / bgzf_flush, main.cpp:1196-1215) ----
struct uvcio_bgzf_writer { FILE *fp = nullptr; std::vector<uint8_t> pending; int level = 6; };
namespace {
const size_t BGZF_INPUT_BLOCK = 0xff00;   // uncompressed bytes per block, as htslib
int bgzf_put_block(uvcio_bgzf_writer *w, const uint8_t *src, size_t n) {
    uint8_t out[65536 + 64];
    z_stream zs; memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, w->level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return fail(UVCGPU_EINVAL, "deflateInit2 failed");
    zs.next_in = const_cast<uint8_t *>(src); zs.avail_in = (uInt)n; zs.next_out = out + 18; zs.avail_out = (uInt)(sizeof(out) - 18 - 8);
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END || clen + 26 > 65536) return fail(UVCGPU_EINVAL, "BGZF block does not fit 64 KiB");
    const uint8_t head[18] = { 31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, 0, 0 };
    memcpy(out, head, 18);
    const uint32_t bsize = (uint32_t)(clen + 26 - 1);
    out[16] = (uint8_t)(bsize & 0xFF); out[17] = (uint8_t)(bsize >> 8);
    const uint32_t crc = block_crc32(src, n), isize = (uint32_t)n;
    uint8_t *t = out + 18 + clen;
    for (int i = 0; i < 4; i++) { t[i] = (uint8_t)(crc >> (8 * i)); t[4 + i] = (uint8_t)(isize >> (8 * i)); }
    if (fwrite(out, 1, clen + 26, w->fp) != clen + 26) return fail(UVCGPU_EINVAL, "short write");
    return 0;
}
}
extern "C" int uvcio_bgzf_write_open(uvcio_bgzf_writer_t **out, const char *path, int32_t level) {
    if (!out || !path) return fail(UVCGPU_EINVAL, "null argument");
    uvcio_bgzf_writer *w = new uvcio_bgzf_writer();
    w->fp = fopen(path, "wb");
    if (!w->fp) { delete w; return fail(UVCGPU_EINVAL, std::string("cannot create ") + path); }
    w->level = (level < 0 || level > 9) ? 6 : level;
    *out = w;
    return 0;
}
extern "C" int uvcio_bgzf_write(uvcio_bgzf_writer_t *w, const void *data, int64_t n) {
    if (!w || (n > 0 && !data) || n < 0) return fail(UVCGPU_EINVAL, "bad argument");
    const uint8_t *p = (const uint8_t *)data;
    w->pending.insert(w->pending.end(), p, p + n);
    size_t at = 0;
    while (w->pending.size() - at >= BGZF_INPUT_BLOCK) { const int rc = bgzf_put_block(w, w->pending.data() + at, BGZF_INPUT_BLOCK); if (rc) return rc; at += BGZF_INPUT_BLOCK; }
    w->pending.erase(w->pending.begin(), w->pending.begin() + (ptrdiff_t)at);
    return 0;
}
// flushes the pending bytes as a (short) block and appends the 28-byte end-of-file marker block
extern "C" int uvcio_bgzf_write_close(uvcio_bgzf_writer_t *w) {
    if (!w) return 0;
    int rc = 0;
    if (!w->pending.empty()) rc = bgzf_put_block(w, w->pending.data(), w->pending.size());
    if (!rc) rc = bgzf_put_block(w, nullptr, 0);
    if (fclose(w->fp) != 0 && !rc) rc = fail(UVCGPU_EINVAL, "close failed");
    delete w;
    return rc;
}

// ---- the region planner: SamIter::iternext without a BED file (grouping.cpp:225-312) over the alignment columns in file order ----
// A block of the reference is cut when the contig changes (flag 16), the next read starts more than 2 * MAX_STR_N_BASES behind the
// running end (8), the block's estimated memory exceeds the per-thread budget (4, check_if_sub_is_over_mem_lim, grouping.cpp:49-67) or
// the file ends (2); check_if_is_over_mem_lim (grouping.cpp:28-47) closes a batch of blocks, which drops the read that triggered the
// cut from the running end of the next block exactly as the reference's early return does.
namespace {
const int64_t PLAN_BYTES_PER_REF_POS = 1024 * 8, PLAN_BYTES_PER_READ = 512, PLAN_MAX_STR_N_BASES = 100, PLAN_UNITS_PER_THREAD = 8;
bool plan_batch_over(int64_t n_reads, int64_t reads_sq, int64_t n_rposs, int64_t rposs_sq, int64_t nthreads, int64_t mem_mb) {
    const uint64_t by_reads = (uint64_t)(std::min<uint64_t>((uint64_t)(reads_sq / std::max<int64_t>(1, n_reads)) * (uint64_t)nthreads, (uint64_t)n_reads) * (uint64_t)PLAN_BYTES_PER_READ);
    const uint64_t by_rposs = (uint64_t)((std::min<uint64_t>((uint64_t)(rposs_sq / std::max<int64_t>(1, n_rposs)) * (uint64_t)nthreads, (uint64_t)n_rposs) + (uint64_t)(2 * PLAN_MAX_STR_N_BASES * nthreads)) * (uint64_t)PLAN_BYTES_PER_REF_POS);
    const uint64_t by_vcf = (uint64_t)n_rposs * 1024;
    return (by_reads + by_rposs + by_vcf) > (uint64_t)(1024 * 1024) * (uint64_t)mem_mb * (uint64_t)nthreads;
}
bool plan_block_over(int64_t n_reads, int64_t n_rposs, int64_t mem_mb, int64_t curr_beg, int64_t running_end) {
    const uint64_t used = (uint64_t)(n_reads * PLAN_BYTES_PER_READ) + (uint64_t)(n_rposs * (PLAN_BYTES_PER_REF_POS + 1024));
    const uint64_t memfree = (uint64_t)((1024 * 1024) / PLAN_UNITS_PER_THREAD) * (uint64_t)mem_mb;
    const uint64_t ovl = (uint64_t)std::min<int64_t>(running_end > curr_beg ? running_end - curr_beg : 0, 150);   // size_t arithmetic in the reference: both are non-negative here
    return used > memfree + memfree * ovl / 150;
}
}
// The planner as a stream: the walk keeps a handful of scalars between alignments, so a caller that reads a whole genome feeds it window by
// window and holds no per-alignment column (uvc1-mi355x; uvcio_plan_regions below is the one-call form of the same code).
struct uvcio_planner {
    int64_t nthreads, mem_mb; std::vector<int64_t> target_len;
    int64_t block_tid = -1, block_beg = -1, block_running_end = -1;   // last_it_* (grouping.hpp)
    int64_t total_reads = 0, total_rposs = 0, total_reads_sq = 0, total_rposs_sq = 0;   // of the current iternext() call
    int64_t region_reads = 0, region_rposs = 0, region_rposs_add = 0;
    int32_t batch = 0;
    bool any = false, last_unmapped = false, done = false;
    bool cur_valid = false;   // this iternext() call has read a record (the end of the file is only processed on a record the call itself holds)
    int64_t last_tid = 0, last_pos = 0, last_end = 0;
    std::vector<UvcRegionCut> cuts;   // cuts that have not been taken yet
    void new_call() { cur_valid = false; total_reads = total_rposs = total_reads_sq = total_rposs_sq = 0; region_reads = region_rposs = region_rposs_add = 0; batch++; }
    // the body of iternext's read loop for one record (ret = 0) or for the end of the file (ret = -1, on the last record read)
    void step(int64_t curr_tid, int64_t curr_beg, int64_t curr_end, int ret) {
        const bool sub_over = plan_block_over(region_reads, region_rposs + region_rposs_add, mem_mb, curr_beg, block_running_end < 0 ? 0 : block_running_end);
        const bool tmpl_changed = (curr_tid != block_tid);
        const bool far_jumped = ((curr_tid == block_tid) && (block_running_end + (PLAN_MAX_STR_N_BASES * 2) < curr_beg));
        const int32_t rflag = (tmpl_changed ? 16 : 0) + (far_jumped ? 8 : 0) + (sub_over ? 4 : 0) + ((-1 == ret) ? 2 : 0);
        if (rflag) {
            const bool first = (-1 == block_tid);
            const int64_t tlen = first ? (int64_t)INT32_MAX : ((block_tid < (int64_t)target_len.size()) ? target_len[(size_t)block_tid] : (int64_t)INT32_MAX);
            const int64_t norm_end = std::min(block_running_end, tlen);
            const bool zero = (block_beg >= norm_end);
            if (!first && !zero) {
                cuts.push_back(UvcRegionCut{ (int32_t)block_tid, (int32_t)block_beg, (int32_t)norm_end, rflag, batch, region_reads });
                const int64_t s_rposs = region_rposs + region_rposs_add;
                total_reads += region_reads; total_rposs += s_rposs; total_reads_sq += region_reads * region_reads; total_rposs_sq += s_rposs * s_rposs;
                region_rposs = 0; region_rposs_add = 0; region_reads = 0;
            }
            block_tid = curr_tid;
            const int64_t new_beg = std::max(block_beg, curr_beg);
            block_beg = (tmpl_changed ? curr_beg : std::max(new_beg, norm_end));
            if (plan_batch_over(total_reads, total_reads_sq, total_rposs, total_rposs_sq, nthreads, mem_mb)) {
                block_running_end = std::max(block_beg, norm_end);
                new_call();   // iternext returns here: the record that closed the batch is not counted (the reference drops it the same way)
                return;
            }
        }
        if (tmpl_changed) { block_beg = curr_beg; block_running_end = curr_end; region_rposs_add += region_rposs; }
        else block_running_end = std::max(block_running_end, curr_end);
        region_reads++;
        region_rposs = block_running_end - block_beg;
    }
};
extern "C" int uvcio_planner_open(uvcio_planner_t **out, const int64_t *target_len, int32_t n_targets, int32_t nthreads, int64_t mem_per_thread_mb) {
    if (!out || nthreads < 1 || mem_per_thread_mb < 1 || n_targets < 0) return fail(UVCGPU_EINVAL, "bad argument");
    uvcio_planner *p = new uvcio_planner();
    p->nthreads = nthreads; p->mem_mb = mem_per_thread_mb;
    if (target_len) p->target_len.assign(target_len, target_len + n_targets);
    *out = p;
    return 0;
}
extern "C" int uvcio_planner_feed(uvcio_planner_t *p, const int32_t *tid, const int32_t *pos, const int32_t *endpos, const uint16_t *flag, int64_t n) {
    if (!p || n < 0 || (n > 0 && (!tid || !pos || !endpos || !flag)) || p->done) return fail(UVCGPU_EINVAL, "bad argument");
    for (int64_t i = 0; i < n; i++) {
        p->any = true; p->cur_valid = true; p->last_unmapped = (flag[i] & 0x4) != 0; p->last_tid = tid[i]; p->last_pos = pos[i]; p->last_end = endpos[i];
        if (flag[i] & 0x4) continue;   // BAM_FUNMAP
        p->step(tid[i], pos[i], endpos[i], 0);
    }
    return 0;
}
extern "C" int uvcio_planner_finish(uvcio_planner_t *p) {   // the end of the file: the read call fails and the loop body runs once more on the record it still holds
    if (!p) return fail(UVCGPU_EINVAL, "bad argument");
    if (!p->done && p->any && p->cur_valid && !p->last_unmapped) p->step(p->last_tid, p->last_pos, p->last_end, -1);
    p->done = true;
    return 0;
}
extern "C" int64_t uvcio_planner_take(uvcio_planner_t *p, UvcRegionCut *out, int64_t capacity) {   // moves up to `capacity` finished cuts out, in order
    if (!p || capacity < 0 || (capacity > 0 && !out)) return 0;
    const int64_t n = std::min<int64_t>(capacity, (int64_t)p->cuts.size());
    for (int64_t k = 0; k < n; k++) out[k] = p->cuts[(size_t)k];
    p->cuts.erase(p->cuts.begin(), p->cuts.begin() + n);
    return n;
}
extern "C" int64_t uvcio_planner_pending(const uvcio_planner_t *p) { return p ? (int64_t)p->cuts.size() : 0; }
extern "C" void uvcio_planner_close(uvcio_planner_t *p) { delete p; }

extern "C" int uvcio_plan_regions(const int32_t *tid, const int32_t *pos, const int32_t *endpos, const uint16_t *flag, int64_t n,
                                  const int64_t *target_len, int32_t n_targets, int32_t nthreads, int64_t mem_per_thread_mb,
                                  UvcRegionCut *out, int64_t capacity, int64_t *n_out) {
    if (!n_out || n < 0 || (n > 0 && (!tid || !pos || !endpos || !flag)) || nthreads < 1 || mem_per_thread_mb < 1) return fail(UVCGPU_EINVAL, "bad argument");
    uvcio_planner_t *p = nullptr;
    if (uvcio_planner_open(&p, target_len, target_len ? n_targets : 0, nthreads, mem_per_thread_mb)) return UVCGPU_EINVAL;
    uvcio_planner_feed(p, tid, pos, endpos, flag, n);
    uvcio_planner_finish(p);
    *n_out = uvcio_planner_pending(p);
    int rc = 0;
    if (*n_out > capacity || (!out && *n_out > 0)) rc = fail(UVCGPU_ENOMEM, "destination too small");
    else uvcio_planner_take(p, out, *n_out);
    uvcio_planner_close(p);
    return rc;
}

// ---- region shards: cost estimate, contiguous balanced partition, concatenation of the shard outputs ----
extern "C" int64_t uvcio_bam_region_bytes(const uvcio_bam_t *b, int32_t tid, int64_t beg, int64_t end) {
    if (!b || !b->has_index || tid < 0 || tid >= (int32_t)b->idx.size() || end <= beg) return 0;
    const std::vector<uint64_t> &lin = b->idx[(size_t)tid].linear;
    if (lin.empty()) return 0;
    // a virtual offset = compressed offset of the block << 16 | offset inside the inflated block; inside one block only the second part
    // moves (small files), so it counts too, at the usual ~4:1 ratio.  Windows without alignments carry offset 0 in files written by
    // some tools: take the nearest filled window in front.
    auto filled = [&](size_t w) { if (w >= lin.size()) w = lin.size() - 1; while (w > 0 && lin[w] == 0) w--; return (int64_t)(lin[w] >> 16) + (int64_t)(lin[w] & 0xFFFF) / 4; };
    // the index knows the offset of the first alignment that overlaps each 16 kb window: between two windows interpolate
    auto at = [&](int64_t p) { p = std::max<int64_t>(p, 0); const size_t w = (size_t)(p >> 14); const int64_t lo = filled(w), hi = std::max(lo, filled(w + 1)); return lo + (hi - lo) * (p & 16383) / 16384; };
    const int64_t a = at(beg), z = at(end);
    return z > a ? z - a : 0;
}
extern "C" int uvcio_plan_shards(const int64_t *cost, int64_t n, int32_t n_shards, int32_t *shard_of) {
    if (n < 0 || n_shards < 1 || (n > 0 && (!cost || !shard_of))) return fail(UVCGPU_EINVAL, "bad argument");
    long double total = 0;
    for (int64_t i = 0; i < n; i++) { if (cost[i] < 0) return fail(UVCGPU_EINVAL, "negative cost"); total += (long double)cost[i]; }
    long double run = 0;
    int32_t prev = 0;
    for (int64_t i = 0; i < n; i++) {
        int32_t s = (total > 0) ? (int32_t)(((run + (long double)cost[i] / 2) * n_shards) / total) : (int32_t)((i * n_shards) / std::max<int64_t>(n, 1));
        s = std::min(std::max(s, prev), n_shards - 1);
        shard_of[i] = s; prev = s; run += (long double)cost[i];
    }
    return 0;
}
extern "C" int uvcio_bgzf_concat(const char *out_path, const char *const *in_paths, int32_t n_in) {
    if (!out_path || n_in < 0 || (n_in > 0 && !in_paths)) return fail(UVCGPU_EINVAL, "bad argument");
    static const uint8_t eof_marker[28] = { 0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    FILE *fo = fopen(out_path, "wb");
    if (!fo) return fail(UVCGPU_EINVAL, std::string("cannot create ") + out_path);
    std::vector<uint8_t> buf;
    for (int32_t k = 0; k < n_in; k++) {
        FILE *fi = fopen(in_paths[k], "rb");
        if (!fi) { fclose(fo); return fail(UVCGPU_EINVAL, std::string("cannot open ") + in_paths[k]); }
        fseeko(fi, 0, SEEK_END); const int64_t sz = (int64_t)ftello(fi); fseeko(fi, 0, SEEK_SET);
        buf.resize((size_t)sz);
        if (sz && fread(buf.data(), 1, (size_t)sz, fi) != (size_t)sz) { fclose(fi); fclose(fo); return fail(UVCGPU_EINVAL, std::string("short read of ") + in_paths[k]); }
        fclose(fi);
        int64_t keep = sz;
        if (sz >= 28 && !memcmp(buf.data() + sz - 28, eof_marker, 28)) keep = sz - 28;
        if (keep && fwrite(buf.data(), 1, (size_t)keep, fo) != (size_t)keep) { fclose(fo); return fail(UVCGPU_EINVAL, "write failed"); }
    }
    const bool ok = (fwrite(eof_marker, 1, 28, fo) == 28);
    if (fclose(fo) != 0 || !ok) return fail(UVCGPU_EINVAL, "write failed");
    return 0;
}
extern "C" int uvcio_read_text_file(const char *path, char **out, int64_t *len) {
    if (!path || !out || !len) return fail(UVCGPU_EINVAL, "bad argument");
    gzFile g = gzopen(path, "rb");   // zlib reads multi-member gzip (= BGZF) and plain files alike
    if (!g) return fail(UVCGPU_EINVAL, std::string("cannot open ") + path);
    gzbuffer(g, 1 << 20);
    size_t cap = (size_t)1 << 22, n = 0;
    char *b = (char *)malloc(cap);
    if (!b) { gzclose(g); return fail(UVCGPU_ENOMEM, "malloc"); }
    for (;;) {
        if (cap - n < ((size_t)1 << 20)) { cap *= 2; char *nb = (char *)realloc(b, cap); if (!nb) { free(b); gzclose(g); return fail(UVCGPU_ENOMEM, "realloc"); } b = nb; }
        const int got = gzread(g, b + n, (unsigned)std::min<size_t>(cap - n - 1, (size_t)1 << 30));
        if (got < 0) { free(b); gzclose(g); return fail(UVCGPU_EINVAL, std::string("cannot inflate ") + path); }
        if (got == 0) break;
        n += (size_t)got;
    }
    gzclose(g);
    b[n] = 0; *out = b; *len = (int64_t)n;
    return 0;
}

// ---- the tumor VCF of a T/N pair: rescue_variants_from_vcf (main.cpp:183-398) on plain text ----
struct uvcio_tumor_vcf {
    std::string sample;
    std::vector<int32_t> tid;                   // per record, parallel to keys
    std::vector<UvcTumorKey> keys;              // sorted by (tid, refpos, symbol); records of one key keep their file order
    std::vector<std::string> cols_text;         // sample column of each record
    std::vector<const char *> cols;             // c_str() of the above
    std::vector<std::string> ra_text;           // "REF\tALT" of each record (TumorKeyInfo::ref_alt)
    std::vector<const char *> ras;
    std::vector<int64_t> tid_first;             // [n_contigs + 1] first record of each tid
};
namespace {
// the comma-separated integers of one FORMAT value ("." = missing -> none)
int parse_ints(const char *p, const char *e, int32_t *out, int cap) {
    int n = 0;
    while (p < e) {
        const char *q = p; while (q < e && *q != ',') q++;
        if (q - p == 1 && *p == '.') return -1;
        if (n < cap) out[n] = (int32_t)strtol(std::string(p, q).c_str(), nullptr, 10);
        n++;
        p = (q < e ? q + 1 : q);
    }
    return n;
}
}
extern "C" int uvcio_tumor_vcf_open(uvcio_tumor_vcf_t **out, const char *path, const char *const *contig_names, int32_t n_contigs, int32_t is_tumor_format_retrieved) {
    if (!out || !path || n_contigs < 0 || (n_contigs > 0 && !contig_names)) return fail(UVCGPU_EINVAL, "bad argument");
    char *text = nullptr; int64_t len = 0;
    int rc = uvcio_read_text_file(path, &text, &len);
    if (rc) return rc;
    std::map<std::string, int32_t> tid_of;
    for (int32_t i = 0; i < n_contigs; i++) tid_of[contig_names[i]] = i;
    uvcio_tumor_vcf *v = new uvcio_tumor_vcf();
    struct Rec { int32_t tid; UvcTumorKey k; std::string col, ra; int64_t ord; };
    std::vector<Rec> recs;
    std::string err;
    const char *p = text, *end = text + len;
    int64_t lineno = 0;
    while (p < end && err.empty()) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        const char *line = p; p = nl ? nl + 1 : end; lineno++;
        if (le > line && le[-1] == '\r') le--;
        if (le == line) continue;
        if (line[0] == '#') {
            if (le - line > 6 && !strncmp(line, "#CHROM", 6)) { const char *t = le; int tabs = 0; for (const char *q = line; q < le; q++) if (*q == '\t') tabs++; while (t > line && t[-1] != '\t') t--; v->sample = (tabs >= 9 ? std::string(t, le) : std::string()); }
            continue;
        }
        const char *f[11]; int nf = 0; f[nf++] = line;
        for (const char *q = line; q < le && nf < 11; q++) if (*q == '\t') f[nf++] = q + 1;
        if (nf < 10) continue;   // no sample column: nothing to rescue
        auto fld = [&](int i) { return std::string(f[i], (i + 1 < nf ? f[i + 1] - 1 : le)); };
        const std::string chrom = fld(0), ref = fld(3), alt = fld(4), fmt = fld(8), smp = fld(9);
        auto ti = tid_of.find(chrom);
        if (ti == tid_of.end()) continue;   // a contig the BAM does not have can never be asked for
        // symbolic alleles: only <NON_REF> and <ADDITIONAL_INDEL_CANDIDATE> pass, and only when the tumor FORMAT is carried over (main.cpp:265-272)
        bool skip = false;
        {   size_t a = 0;
            while (a <= alt.size()) { size_t c = alt.find(',', a); if (c == std::string::npos) c = alt.size();
                const std::string one = alt.substr(a, c - a);
                if (!one.empty() && one[0] == '<' && ((one != "<NON_REF>" && one != "<ADDITIONAL_INDEL_CANDIDATE>") || !is_tumor_format_retrieved)) skip = true;
                a = c + 1; } }
        if (skip) continue;
        // FORMAT keys -> sample values
        std::vector<std::pair<const char *, const char *>> keys, vals;
        { const char *a = fmt.data(), *z = a + fmt.size(); while (a <= z) { const char *c = a; while (c < z && *c != ':') c++; keys.emplace_back(a, c); a = c + 1; } }
        { const char *a = smp.data(), *z = a + smp.size(); while (a <= z) { const char *c = a; while (c < z && *c != ':') c++; vals.emplace_back(a, c); a = c + 1; } }
        auto get = [&](const char *name, int32_t *dst, int cap) -> int {   // number of values, 0 = tag absent, -1 = "."
            const size_t nl2 = strlen(name);
            for (size_t k = 0; k < keys.size() && k < vals.size(); k++) if ((size_t)(keys[k].second - keys[k].first) == nl2 && !memcmp(keys[k].first, name, nl2)) return parse_ints(vals[k].first, vals[k].second, dst, cap);
            return 0;
        };
        int32_t a2[4];
        if (get("VTI", a2, 4) != 2) continue;   // valsize <= 0 -> continue (main.cpp:275)
        Rec r; memset(&r.k, 0, sizeof(r.k));
        r.tid = ti->second; r.ord = (int64_t)recs.size(); r.col = smp; r.ra = ref + "\t" + alt;
        const int symbol = a2[1];
        const long pos0 = strtol(fld(1).c_str(), nullptr, 10) - 1;   // line->pos
        const bool at_pos = (symbol <= UVC_BASE_NN || symbol == UVC_MGVCF_SYMBOL || symbol == UVC_ADDITIONAL_INDEL_CANDIDATE_SYMBOL);   // isSymbolSubstitution covers BASE_A .. BASE_NN
        r.k.refpos = (int32_t)(at_pos ? pos0 : pos0 + 1); r.k.symbol = symbol;
        if (symbol != UVC_MGVCF_SYMBOL && symbol != UVC_ADDITIONAL_INDEL_CANDIDATE_SYMBOL) {
            auto need = [&](const char *name, int n_expect, int32_t *dst) { if (get(name, dst, 4) != n_expect && err.empty()) err = std::string("FORMAT/") + name + " of line " + std::to_string(lineno) + " does not have " + std::to_string(n_expect) + " integers (main.cpp:294-372 asserts it)"; };
            need("BDPb", 2, a2); r.k.BDP = a2[0] + a2[1];
            need("bDPf", 2, a2); r.k.bDP = a2[1]; need("bDPr", 2, a2); r.k.bDP += a2[1];
            need("CDP1x", 1, a2); r.k.CDP1x = a2[0]; need("cDP1x", 2, a2); r.k.cDP1x = a2[1];
            need("cVQ1", 2, a2); r.k.cVQ1 = a2[1]; need("cPCQ1", 2, a2); r.k.cPCQ1 = a2[1];
            need("CDP2x", 1, a2); r.k.CDP2x = a2[0]; need("cDP2x", 2, a2); r.k.cDP2x = a2[1];
            need("cVQ2", 2, a2); r.k.cVQ2 = a2[1]; need("cPCQ2", 2, a2); r.k.cPCQ2 = a2[1];
            need("bNMQ", 2, a2); r.k.bNMQ = a2[1]; need("vHGQ", 1, a2); r.k.vHGQ = a2[0];
            need("CDP1b", 2, a2); r.k.tDP = a2[0] + a2[1];
            need("cDP1f", 2, a2); r.k.tAD0 = a2[0]; r.k.tAD1 = a2[1]; need("cDP1r", 2, a2); r.k.tAD0 += a2[0]; r.k.tAD1 += a2[1];
            need("CDP2b", 2, a2); r.k.t2DP = a2[0] + a2[1];
            if (symbol >= UVC_LINK_D3P && symbol <= UVC_LINK_I1)   // the InDel string of the record: REF / ALT without their common head (main.cpp:867-880)
                r.k.indel_len = (int32_t)(ref.size() > alt.size() ? ref.size() - alt.size() : alt.size() - ref.size());
        }
        for (size_t k = 0; k < keys.size(); k++) if ((keys[k].second - keys[k].first) == 5 && !memcmp(keys[k].first, "_C2XP", 5)) r.k.tier2 = 1;   // enable_tier2_consensus_format_tags, main.cpp:386-388
        recs.push_back(std::move(r));
    }
    free(text);
    if (!err.empty()) { delete v; return fail(UVCGPU_EINVAL, err); }
    std::stable_sort(recs.begin(), recs.end(), [](const Rec &a, const Rec &b) {
        if (a.tid != b.tid) return a.tid < b.tid;
        if (a.k.refpos != b.k.refpos) return a.k.refpos < b.k.refpos;
        return a.k.symbol < b.k.symbol; });
    v->tid_first.assign((size_t)n_contigs + 1, 0);
    for (const Rec &r : recs) { v->tid.push_back(r.tid); v->keys.push_back(r.k); v->cols_text.push_back(r.col); v->ra_text.push_back(r.ra); v->tid_first[(size_t)r.tid + 1]++; }
    for (int32_t i = 0; i < n_contigs; i++) v->tid_first[(size_t)i + 1] += v->tid_first[(size_t)i];
    for (const std::string &s : v->cols_text) v->cols.push_back(s.c_str());
    for (const std::string &s : v->ra_text) v->ras.push_back(s.c_str());
    *out = v;
    return 0;
}
extern "C" const char *uvcio_tumor_vcf_sample_name(const uvcio_tumor_vcf_t *v) { return v ? v->sample.c_str() : ""; }
extern "C" int64_t uvcio_tumor_vcf_n_records(const uvcio_tumor_vcf_t *v) { return v ? (int64_t)v->keys.size() : 0; }
extern "C" int uvcio_tumor_vcf_fetch(const uvcio_tumor_vcf_t *v, int32_t tid, int32_t pos_beg, int32_t pos_end, const UvcTumorKey **keys, const char *const **cols, const char *const **ref_alts, int64_t *n) {
    if (!v || !n) return fail(UVCGPU_EINVAL, "bad argument");
    *n = 0; if (keys) *keys = nullptr; if (cols) *cols = nullptr; if (ref_alts) *ref_alts = nullptr;
    if (tid < 0 || (size_t)tid + 1 >= v->tid_first.size()) return 0;
    const int64_t lo0 = v->tid_first[(size_t)tid], hi0 = v->tid_first[(size_t)tid + 1];
    const auto b = v->keys.begin();
    const int64_t lo = std::lower_bound(b + lo0, b + hi0, pos_beg, [](const UvcTumorKey &k, int32_t p) { return k.refpos < p; }) - b;
    const int64_t hi = std::upper_bound(b + lo, b + hi0, pos_end, [](int32_t p, const UvcTumorKey &k) { return p < k.refpos; }) - b;
    *n = hi - lo;
    if (*n > 0) { if (keys) *keys = v->keys.data() + lo; if (cols) *cols = v->cols.data() + lo; if (ref_alts) *ref_alts = v->ras.data() + lo; }
    return 0;
}
extern "C" void uvcio_tumor_vcf_close(uvcio_tumor_vcf_t *v) { delete v; }
