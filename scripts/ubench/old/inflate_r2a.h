// uvc_inflate_fast.h -- raw DEFLATE (RFC 1951) decoder for BGZF blocks on the host, written for throughput.
//
// The reference reads its BAM through htslib's bgzf_read -> zlib inflate (grouping.cpp:617-731).  Behind the GPU path the host's inflate is
// what bounds files -> VCF: zlib's inflate decodes a symbol per table probe with a byte-wise bit buffer.  This
// decoder keeps 56+ bits in a 64-bit buffer (one unaligned 8-byte load per refill), resolves a literal / length code in one probe of an
// 11-bit table (longer codes through sub-tables), takes up to three literals per refill and copies matches eight bytes at a time.
// It decodes ONE complete stream of known output size (a BGZF block: <= 64 KiB) and returns false on anything it does not like -- the
// caller then lets zlib look at the block, and checks the CRC-32 of the result in either case (uvc_io.cpp: inflate_block).
#ifndef UVC_INFLATE_FAST_H
#define UVC_INFLATE_FAST_H
#include <cstdint>
#include <cstring>

namespace uvc_fast_inflate {

enum { LL_BITS = 11, D_BITS = 8, PRE_BITS = 7 };
enum { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3 };
// entry: bits 0..7 code length to consume (for K_SUB: the primary bits), 8..9 kind, 10..15 number of extra bits (K_SUB: log2 of the sub-table),
//        16..31 literal / base value / first index of the sub-table
static inline uint32_t mk(int len, int kind, int extra, int val) { return (uint32_t)len | ((uint32_t)kind << 8) | ((uint32_t)extra << 10) | ((uint32_t)val << 16); }

struct Tables {
    uint32_t ll[(1 << LL_BITS) + 1024];    // 288 symbols, codes <= 15 bits
    uint32_t d[(1 << D_BITS) + 512];       // 30 symbols
    uint32_t pre[1 << PRE_BITS];
};

static const uint16_t LEN_BASE[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
static const uint8_t LEN_EXTRA[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
static const uint16_t DIST_BASE[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
static const uint8_t DIST_EXTRA[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

static inline uint32_t bitrev(uint32_t code, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; i++) { r = (r << 1) | (code & 1); code >>= 1; }
    return r;
}

// which: 0 literal/length alphabet, 1 distance alphabet, 2 code-length alphabet.  A canonical code from the lengths, entries indexed by the
// bit-reversed code (DEFLATE sends codes most significant bit first into an LSB-first stream).  Returns false for an over-subscribed code, or
// an incomplete one other than the single-code distance / the empty distance alphabet (zlib accepts a few more; the caller falls back).
static bool build(uint32_t *tab, int tab_cap, int primary_bits, const uint8_t *lens, int n, int which, bool fixed = false) {
    int count[16] = { 0 };
    for (int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    int left = 1, used = 0;
    for (int l = 1; l <= 15; l++) { left = (left << 1) - count[l]; if (left < 0) return false; used += count[l]; }
    const int psize = 1 << primary_bits;
    if (used == 0) { if (which != 1) return false; for (int i = 0; i < psize; i++) tab[i] = 0; return true; }   // no distance codes: any use is an error (entry 0 = length 0)
    if (left > 0 && !(which == 1 && (used == 1 || fixed))) return false;   // (the fixed distance code has 30 of its 32 codes)
    int next[16]; next[1] = 0;
    for (int l = 1; l < 15; l++) next[l + 1] = (next[l] + count[l]) << 1;
    for (int i = 0; i < psize; i++) tab[i] = 0;
    // sub-tables: for every primary prefix the longest code that starts with it
    uint8_t sub_bits[1 << LL_BITS];
    memset(sub_bits, 0, (size_t)psize);
    {
        int nx[16]; memcpy(nx, next, sizeof(nx));
        for (int s = 0; s < n; s++) {
            const int l = lens[s];
            if (l <= primary_bits) { if (l) nx[l]++; continue; }
            const uint32_t r = bitrev((uint32_t)nx[l]++, l);
            const uint32_t pfx = r & (uint32_t)(psize - 1);
            if (l - primary_bits > sub_bits[pfx]) sub_bits[pfx] = (uint8_t)(l - primary_bits);
        }
    }
    int free_at = psize;
    for (int pfx = 0; pfx < psize; pfx++) if (sub_bits[pfx]) {
        const int sz = 1 << sub_bits[pfx];
        if (free_at + sz > tab_cap) return false;
        tab[pfx] = mk(primary_bits, K_SUB, sub_bits[pfx], free_at);
        for (int i = 0; i < sz; i++) tab[free_at + i] = 0;
        free_at += sz;
    }
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t r = bitrev((uint32_t)next[l]++, l);
        uint32_t e;
        if (which == 0) {
            if (s < 256) e = mk(l, K_LIT, 0, s);
            else if (s == 256) e = mk(l, K_EOB, 0, 0);
            else if (s <= 285) e = mk(l, K_LEN, LEN_EXTRA[s - 257], LEN_BASE[s - 257]);
            else e = 0;   // 286, 287: never valid in a stream
        } else if (which == 1) e = (s < 30 ? mk(l, K_LEN, DIST_EXTRA[s], DIST_BASE[s]) : 0);
        else e = mk(l, K_LIT, 0, s);
        if (l <= primary_bits) {
            if (tab[r] != 0 && ((tab[r] >> 8) & 3) == K_SUB) return false;   // cannot happen in a prefix code
            for (uint32_t i = r; i < (uint32_t)psize; i += 1u << l) tab[i] = e;
        } else {
            const uint32_t pfx = r & (uint32_t)(psize - 1), head = tab[pfx];
            const int sb = (int)((head >> 10) & 63), base = (int)(head >> 16), sl = l - primary_bits;
            e = (e & ~0xFFu) | (uint32_t)sl;   // bits to consume after the primary ones
            for (uint32_t i = r >> primary_bits; i < (1u << sb); i += 1u << sl) tab[base + i] = e;
        }
    }
    return true;
}

static inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }   // little-endian host (x86-64)

// one stream, all of `in`, exactly out_len bytes of output
static bool inflate(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
    static thread_local Tables tls_tables;
    Tables &T = tls_tables;   // one TLS address computation, not one per table probe (this is a shared library)
    const uint8_t *ip = in, *const in_end = in + in_len;
    uint8_t *op = out, *const out_end = out + out_len;
    uint64_t bb = 0; int bc = 0;
    // refill: at least 56 bits while 8 input bytes are left, whatever is left otherwise (zeros behind the end: a code that needs them fails the
    // length checks below because `over` counts the bits taken from behind the end)
    int over = 0;
#define REFILL() do { if (in_end - ip >= 8) { bb |= load64(ip) << bc; ip += (63 - bc) >> 3; bc |= 56; } \
                      else { while (bc <= 56) { if (ip < in_end) bb |= (uint64_t)*ip++ << bc; else over += 8; bc += 8; } } } while (0)
#define TAKE(n) do { bb >>= (n); bc -= (n); } while (0)
    for (;;) {
        REFILL();
        const int final = (int)(bb & 1), type = (int)((bb >> 1) & 3);
        TAKE(3);
        if (type == 0) {   // stored
            TAKE(bc & 7);
            REFILL();
            const uint32_t len = (uint32_t)(bb & 0xFFFF), nlen = (uint32_t)((bb >> 16) & 0xFFFF);
            TAKE(32);
            if ((len ^ 0xFFFF) != nlen) return false;
            // give the whole bytes of the buffer back to the input
            if (over > bc) return false;   // LEN / NLEN came partly from behind the end
            ip -= (bc - over) >> 3; bb = 0; bc = 0; over = 0;
            if ((size_t)(in_end - ip) < len || (size_t)(out_end - op) < len) return false;
            memcpy(op, ip, len); op += len; ip += len;
        } else if (type == 1 || type == 2) {
            uint8_t lens[288 + 32];
            int hlit = 288, hdist = 30;
            if (type == 1) {
                for (int i = 0; i < 144; i++) lens[i] = 8;
                for (int i = 144; i < 256; i++) lens[i] = 9;
                for (int i = 256; i < 280; i++) lens[i] = 7;
                for (int i = 280; i < 288; i++) lens[i] = 8;
                for (int i = 0; i < 30; i++) lens[288 + i] = 5;
            } else {
                hlit = (int)(bb & 31) + 257; hdist = (int)((bb >> 5) & 31) + 1;
                const int hclen = (int)((bb >> 10) & 15) + 4;
                TAKE(14);
                if (hlit > 286 || hdist > 30) return false;
                static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
                uint8_t pl[19] = { 0 };
                REFILL();
                for (int i = 0; i < hclen; i++) { if (bc < 3) REFILL(); pl[order[i]] = (uint8_t)(bb & 7); TAKE(3); }
                if (!build(T.pre, 1 << PRE_BITS, PRE_BITS, pl, 19, 2)) return false;
                int i = 0;
                while (i < hlit + hdist) {
                    REFILL();
                    const uint32_t e = T.pre[bb & ((1 << PRE_BITS) - 1)];
                    const int l = (int)(e & 0xFF), sym = (int)(e >> 16);
                    if (!l) return false;
                    TAKE(l);
                    if (sym < 16) lens[i++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (!i) return false; val = lens[i - 1]; rep = 3 + (int)(bb & 3); TAKE(2); }
                        else if (sym == 17) { rep = 3 + (int)(bb & 7); TAKE(3); }
                        else { rep = 11 + (int)(bb & 127); TAKE(7); }
                        if (i + rep > hlit + hdist) return false;
                        while (rep--) lens[i++] = (uint8_t)val;
                    }
                }
                if (lens[256] == 0) return false;
                // the two alphabets behind each other: move the distance lengths to their own place
                memmove(lens + 288, lens + hlit, (size_t)hdist);
                memset(lens + hlit, 0, (size_t)(288 - hlit));
            }
            if (!build(T.ll, (int)(sizeof(T.ll) / 4), LL_BITS, lens, 288, 0)) return false;
            { uint8_t dl[30] = { 0 }; memcpy(dl, lens + 288, (size_t)hdist); if (!build(T.d, (int)(sizeof(T.d) / 4), D_BITS, dl, 30, 1, type == 1)) return false; }
            const uint32_t *const ll = T.ll, *const dt = T.d;
            // Fast loop: while 16 input bytes and 3 literals + the longest match + the copy's overrun fit, nothing is bounds-checked per
            // symbol, the refill has no branch, and the next table entry is fetched before the current literal is stored.
            while (in_end - ip >= 16 && out_end - op >= 3 + 258 + 8) {
                bb |= load64(ip) << bc; ip += (63 - bc) >> 3; bc |= 56;
                uint32_t e = ll[bb & ((1 << LL_BITS) - 1)];
                if ((e & 0x300) == 0 && (e & 0xFF)) {            // literal
                    TAKE((int)(e & 0xFF)); const uint32_t e2 = ll[bb & ((1 << LL_BITS) - 1)]; *op++ = (uint8_t)(e >> 16);
                    if ((e2 & 0x300) == 0 && (e2 & 0xFF)) {
                        TAKE((int)(e2 & 0xFF)); const uint32_t e3 = ll[bb & ((1 << LL_BITS) - 1)]; *op++ = (uint8_t)(e2 >> 16);
                        if ((e3 & 0x300) == 0 && (e3 & 0xFF)) { TAKE((int)(e3 & 0xFF)); *op++ = (uint8_t)(e3 >> 16); continue; }
                        e = e3;
                    } else e = e2;
                    // >= 56 - 30 = 26 bits left: enough for a length code with its extra bits (20) or a sub-table walk (15)
                }
                if (((e >> 8) & 3) == K_SUB) { TAKE(LL_BITS); e = ll[(e >> 16) + (bb & ((1u << ((e >> 10) & 63)) - 1))]; }
                const int l = (int)(e & 0xFF), kind = (int)((e >> 8) & 3);
                if (!l) return false;
                TAKE(l);
                if (kind == K_LIT) { *op++ = (uint8_t)(e >> 16); continue; }   // (a literal with a code longer than the primary table)
                if (kind == K_EOB) goto block_done;
                const int xl = (int)((e >> 10) & 63);
                const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1u << xl) - 1));
                TAKE(xl);
                bb |= load64(ip) << bc; ip += (63 - bc) >> 3; bc |= 56;
                uint32_t de = dt[bb & ((1 << D_BITS) - 1)];
                if (((de >> 8) & 3) == K_SUB) { TAKE(D_BITS); de = dt[(de >> 16) + (bb & ((1u << ((de >> 10) & 63)) - 1))]; }
                const int dl = (int)(de & 0xFF);
                if (!dl) return false;
                TAKE(dl);
                const int dx = (int)((de >> 10) & 63);
                const uint32_t dist = (de >> 16) + (uint32_t)(bb & ((1u << dx) - 1));
                TAKE(dx);
                if (dist > (size_t)(op - out)) return false;
                const uint8_t *src = op - dist;
                if (dist >= 8) { uint8_t *dst = op; uint8_t *const e8 = op + len; do { memcpy(dst, src, 8); dst += 8; src += 8; } while (dst < e8); }
                else if (dist == 1) memset(op, *src, len);
                else for (uint32_t i = 0; i < len; i++) op[i] = src[i];
                op += len;
            }
            for (;;) {
                REFILL();
                uint32_t e = T.ll[bb & ((1 << LL_BITS) - 1)];
                if (((e >> 8) & 3) == K_SUB) { TAKE(LL_BITS); e = T.ll[(e >> 16) + (bb & ((1u << ((e >> 10) & 63)) - 1))]; }
                int l = (int)(e & 0xFF);
                if (!l) return false;
                int kind = (int)((e >> 8) & 3);
                if (kind == K_LIT) {
                    // up to three literals per refill (3 x 15 bits of 56)
                    if (op >= out_end) return false;
                    TAKE(l); *op++ = (uint8_t)(e >> 16);
                    e = T.ll[bb & ((1 << LL_BITS) - 1)];
                    if (((e >> 8) & 3) != K_LIT || !(e & 0xFF) || op >= out_end) continue;
                    TAKE((int)(e & 0xFF)); *op++ = (uint8_t)(e >> 16);
                    e = T.ll[bb & ((1 << LL_BITS) - 1)];
                    if (((e >> 8) & 3) != K_LIT || !(e & 0xFF) || op >= out_end) continue;
                    TAKE((int)(e & 0xFF)); *op++ = (uint8_t)(e >> 16);
                    continue;
                }
                TAKE(l);
                if (kind == K_EOB) break;
                const int xl = (int)((e >> 10) & 63);
                const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1u << xl) - 1));
                TAKE(xl);
                if (bc < 32) REFILL();
                uint32_t de = T.d[bb & ((1 << D_BITS) - 1)];
                if (((de >> 8) & 3) == K_SUB) { TAKE(D_BITS); de = T.d[(de >> 16) + (bb & ((1u << ((de >> 10) & 63)) - 1))]; }
                const int dl = (int)(de & 0xFF);
                if (!dl) return false;
                TAKE(dl);
                const int dx = (int)((de >> 10) & 63);
                const uint32_t dist = (de >> 16) + (uint32_t)(bb & ((1u << dx) - 1));
                TAKE(dx);
                if (dist > (size_t)(op - out) || len > (size_t)(out_end - op)) return false;
                const uint8_t *src = op - dist;
                if (dist >= 8 && (size_t)(out_end - op) >= len + 8) {   // eight bytes at a time; the overrun stays inside this block's output
                    uint8_t *dst = op; const uint8_t *s = src; uint8_t *const e8 = op + len;
                    do { memcpy(dst, s, 8); dst += 8; s += 8; } while (dst < e8);
                } else if (dist == 1) memset(op, *src, len);
                else for (uint32_t i = 0; i < len; i++) op[i] = src[i];
                op += len;
            }
            block_done:;
        } else return false;
        if (final) break;
    }
#undef REFILL
#undef TAKE
    // every output byte written, no input bit invented, nothing but padding left (a BGZF payload ends with its last block)
    return op == out_end && over <= bc;
}

}   // namespace uvc_fast_inflate
#endif
