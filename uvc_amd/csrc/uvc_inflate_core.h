// uvc_inflate_core.h -- raw DEFLATE (RFC 1951) decoder for one BGZF block, written so that the same code compiles for the host (the
// CPU test compares it with zlib on real BGZF blocks) and as the body of a GPU thread (uvc_inflate.hip: one lane per block).
//
// Shape for SIMT execution: the decoder is ONE loop whose iterations each do one bounded piece of work -- copy at most 8 bytes of a pending
// match, or decode one symbol, or read one code length of a dynamic header -- so that the 64 blocks of a wave advance together instead of
// every lane waiting for the longest match copy.  Tables (2.3 KB per block) live in the caller's InflState (LDS on the GPU: one wave per CU).
// Every read and write is bounds-checked against the block's compressed / uncompressed sizes from the BGZF header / footer: a corrupt block
// ends with an error code, never with an out-of-range access, and the loop ends after at most in_len * 8 + out_len + a constant iterations.
#ifndef UVC_INFLATE_CORE_H_INCLUDED
#define UVC_INFLATE_CORE_H_INCLUDED

#include <stdint.h>

#if defined(__HIPCC__)
#define UVC_HD __host__ __device__ inline
#else
#define UVC_HD inline
#endif

#define UVC_INFL_LBITS 9    // literal / length codes up to this long resolve with one table read
#define UVC_INFL_DBITS 7    // (2.3 KB of tables per block: 64 decoders fit the 160 KB of LDS of one CU)

// LB / DB: index bits of the two fast tables.  The lane-per-block kernel and the host build use 9 / 7; the wave-per-block kernels have one
// table set per wave and can afford 10 / 9 (3.9 KB), which takes most distance codes of BAM data off the bit-by-bit path.
template <int LB, int DB>
struct InflStateT {
    uint16_t lit_fast[1 << LB];                // len << 9 | symbol, 0 = longer code
    uint16_t dist_fast[1 << DB];               // len << 5 | symbol, 0 = longer code
    uint16_t lit_count[16], dist_count[16];    // canonical form for the longer codes: codes per length, symbols in code order
    uint16_t lit_sym[288], dist_sym[32];
    uint8_t lens[320];                         // code lengths of a dynamic header while they are read
};
typedef InflStateT<UVC_INFL_LBITS, UVC_INFL_DBITS> InflState;

enum { UVC_INFL_OK = 0, UVC_INFL_EINPUT = -1 /* ran out of input */, UVC_INFL_EOUTPUT = -2 /* more output than ISIZE */, UVC_INFL_ECODE = -3 /* invalid code / header */,
       UVC_INFL_ESHORT = -4 /* stream ended before ISIZE bytes */, UVC_INFL_EDIST = -5 /* distance before the start of the block */ };

// canonical Huffman: counts, symbol order and the fast table of `n` code lengths; returns 0, or -1 for an over-subscribed set
// (pointer types are template parameters: on the GPU the tables are address-space-3 pointers, so that a lookup is a ds_read and not a
// FLAT load that has to wait for every global store in flight)
template <class PL, class PC, class PS, class PF>
UVC_HD int uvc_infl_build(PL lens, int n, PC count, PS sym, PF fast, int fast_bits, int sym_shift) {
    for (int l = 0; l < 16; l++) count[l] = 0;
    for (int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    int left = 1;
    for (int l = 1; l < 16; l++) { left <<= 1; left -= count[l]; if (left < 0) return -1; }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
    for (int i = 0; i < n; i++) if (lens[i]) sym[offs[lens[i]]++] = (uint16_t)i;
    for (int i = 0; i < (1 << fast_bits); i++) fast[i] = 0;
    // codes in canonical order; the bit stream carries them most significant bit first, the table is indexed by the next bits least significant first
    int code = 0, index = 0;
    for (int l = 1; l <= fast_bits; l++) {
        for (int k = 0; k < count[l]; k++) {
            int rev = 0;
            for (int b = 0; b < l; b++) rev |= ((code >> b) & 1) << (l - 1 - b);
            const uint16_t e = (uint16_t)((l << sym_shift) | sym[index]);
            for (int f = rev; f < (1 << fast_bits); f += (1 << l)) fast[f] = e;
            code++; index++;
        }
        code <<= 1;
    }
    return 0;
}

// Inflates in[0, in_len) into out[0, out_len); returns UVC_INFL_OK only if the stream ends with its final block exactly at out_len bytes.
// COOP (device only): the 64 lanes of a wave run this function together on ONE block, every value below identical in all of them (one table
// set per wave, input words read by broadcast); lane 0 stores literals, and a match is copied by the lanes side by side -- one step per
// match whatever its length, behind a wavefront-scope fence that orders the wave's earlier stores before the loads of the copy.
template <bool COOP, int LB, int DB, class ST>
UVC_HD int uvc_inflate_block_t(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len, ST &S, const uint32_t lane) {
    uint64_t bitbuf = 0; int bitcnt = 0; uint32_t ip = 0, op = 0;
    // phases of the one loop
    enum { PH_HEADER, PH_STORED, PH_LENS, PH_SYMBOL, PH_COPY, PH_DONE };
    int phase = PH_HEADER, last = 0, err = UVC_INFL_OK;
    uint32_t copy_len = 0, copy_dist = 0, stored_left = 0;
    int nlen = 0, ndist = 0, lens_at = 0;
    uint16_t cl_count[16], cl_sym[19];
    // base value and number of extra bits of a length symbol (257 + s) and of a distance symbol, RFC 1951 3.2.5, as arithmetic: indexed
    // loads of four little constant tables were four global-memory round trips per match on the GPU
    auto len_ext = [](int s_) -> int { return (s_ < 8 || s_ == 28) ? 0 : ((s_ - 4) >> 2); };
    auto len_base = [&](int s_) -> uint32_t { return s_ < 8 ? (uint32_t)(3 + s_) : (s_ == 28 ? 258u : (uint32_t)(3 + ((4 + (s_ & 3)) << len_ext(s_)))); };
    auto dist_ext = [](int d_) -> int { return d_ < 4 ? 0 : ((d_ - 2) >> 1); };
    auto dist_base = [&](int d_) -> uint32_t { return d_ < 4 ? (uint32_t)(d_ + 1) : (uint32_t)(1 + ((2 + (d_ & 1)) << dist_ext(d_))); };
    static constexpr uint8_t CLORDER[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
    // a length code with its extra bits takes at most 20 bits, a distance code 28: one refill in front of each
    // keeps at least 32 bits in the buffer while input lasts.  The input comes through a 64-bit look-ahead word that is re-loaded (one
    // unaligned 8-byte load, single bytes only at the tail) as soon as it is empty, i.e. a few symbols before its bits are needed: the
    // load's latency is not on the decode path
    uint64_t ahead = 0; int ahead_bits = 0;
#define UVC_INFL_LOAD_AHEAD() do { \
        if (ahead_bits == 0 && ip < in_len) { \
            if (ip + 8 <= in_len) { __builtin_memcpy(&ahead, in + ip, 8); ip += 8; ahead_bits = 64; } \
            else { ahead = 0; while (ip < in_len) { ahead |= (uint64_t)in[ip++] << ahead_bits; ahead_bits += 8; } } \
        } } while (0)
#define UVC_INFL_REFILL() do { \
        for (int r_ = 0; r_ < 2 && bitcnt <= 32; r_++) { \
            UVC_INFL_LOAD_AHEAD(); \
            const int t_ = ahead_bits < 32 ? ahead_bits : 32; \
            if (t_ == 0) break; \
            bitbuf |= (ahead & ((1ull << t_) - 1)) << bitcnt; ahead >>= t_; ahead_bits -= t_; bitcnt += t_; \
            UVC_INFL_LOAD_AHEAD(); \
        } } while (0)
#define UVC_INFL_TAKE(n) do { bitbuf >>= (n); bitcnt -= (n); } while (0)
    // a symbol of a canonical code the fast table does not resolve: bit by bit from length 1 (puff's decode)
    auto slow = [&](auto count, auto sym, int &used) -> int {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; l++) {
            code |= (int)((bitbuf >> (l - 1)) & 1);
            const int c = count[l];
            if (code - c < first) { used = l; return sym[index + (code - first)]; }
            index += c; first += c; first <<= 1; code <<= 1;
        }
        used = 0; return -1;
    };
    const uint64_t max_iter = (uint64_t)in_len * 8 + (uint64_t)out_len + 1024;   // every iteration consumes a bit, produces a byte, or ends a block
    for (uint64_t it = 0; phase != PH_DONE && it < max_iter; it++) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (COOP) {   // tell the compiler what the launch guarantees: the decoder's state is the same in every lane -> scalar registers, scalar branches
            auto u32 = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
            auto u64 = [&](uint64_t v) -> uint64_t { return (uint64_t)u32((uint32_t)v) | ((uint64_t)u32((uint32_t)(v >> 32)) << 32); };
            bitbuf = u64(bitbuf); ahead = u64(ahead); bitcnt = (int)u32((uint32_t)bitcnt); ahead_bits = (int)u32((uint32_t)ahead_bits);
            ip = u32(ip); op = u32(op); phase = (int)u32((uint32_t)phase); last = (int)u32((uint32_t)last); stored_left = u32(stored_left);
            lens_at = (int)u32((uint32_t)lens_at); nlen = (int)u32((uint32_t)nlen); ndist = (int)u32((uint32_t)ndist);
        }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
        if (COOP && phase == PH_SYMBOL) {
            // The symbols of one Huffman block in a loop of their own.  The state machine around it costs ~300 instructions per symbol, and with
            // every block of a tile resident (8 waves per SIMD) the kernel is bound by instruction issue, not by latency.  Here the bit reader is
            // a 64-bit scalar buffer topped up 32 bits at a time from a word loaded one step ahead; table entries and input words come back
            // through v_readfirstlane, everything else is scalar.  Bits behind the end of the input read as zeros and are caught at the end.
            auto u32 = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
            // (in has 8 bytes of padding behind the last block.)  wordv leaves the loaded word in a vector register: it is read (u32) at the NEXT
            // top-up, two or three symbols later, so the memory round trip of the look-ahead word is never waited for
            auto wordv = [&](uint32_t at) -> uint32_t { uint32_t w = 0; if (at < in_len) __builtin_memcpy(&w, in + at, 4); return w; };
            auto word = [&](uint32_t at) -> uint32_t { return u32(wordv(at)); };
            const uint32_t bitpos = ip * 8 - (uint32_t)ahead_bits - (uint32_t)bitcnt;   // bits consumed so far
            uint32_t wp = bitpos >> 3;
            uint64_t bb = ((uint64_t)word(wp) | ((uint64_t)word(wp + 4) << 32)) >> (bitpos & 7);
            int bc = 64 - (int)(bitpos & 7);
            wp += 8;
            uint32_t nextw = wordv(wp);
            // a match of <= 64 bytes is loaded at once and stored when the next match (or the end of the block) comes: its round trip runs beside
            // the decoding of the symbols in between
            uint32_t pend_len = 0, pend_op = 0; uint8_t pend_byte = 0;
            for (;;) {
                if (bc <= 32) { bb |= (uint64_t)u32(nextw) << bc; bc += 32; wp += 4; nextw = wordv(wp); }
                int sym, used;
                const uint32_t e = u32(S.lit_fast[bb & ((1u << LB) - 1)]);
                if (e) { used = (int)(e >> 9); sym = (int)(e & 511); }
                else {   // longer than the fast table: bit by bit (puff's decode)
                    int code = 0, first = 0, index = 0; sym = -1; used = 0;
                    for (int l = 1; l < 16; l++) {
                        code |= (int)((bb >> (l - 1)) & 1);
                        const int c = (int)u32(S.lit_count[l]);
                        if (code - c < first) { used = l; sym = (int)u32(S.lit_sym[index + (code - first)]); break; }
                        index += c; first += c; first <<= 1; code <<= 1;
                    }
                }
                if (sym < 0) { err = UVC_INFL_ECODE; break; }
                bb >>= used; bc -= used;
                if (sym < 256) { if (op >= out_len) { err = UVC_INFL_EOUTPUT; break; } if (lane == 0) out[op] = (uint8_t)sym; op++; continue; }
                if (sym == 256) { phase = last ? PH_DONE : PH_HEADER; break; }
                sym -= 257;
                if (sym >= 29) { err = UVC_INFL_ECODE; break; }
                const int lext = len_ext(sym);
                const uint32_t len = len_base(sym) + (uint32_t)(bb & ((1u << lext) - 1));
                bb >>= lext; bc -= lext;
                if (bc <= 32) { bb |= (uint64_t)u32(nextw) << bc; bc += 32; wp += 4; nextw = wordv(wp); }
                int dsym, dused;
                const uint32_t de = u32(S.dist_fast[bb & ((1u << DB) - 1)]);
                if (de) { dused = (int)(de >> 5); dsym = (int)(de & 31); }
                else {
                    int code = 0, first = 0, index = 0; dsym = -1; dused = 0;
                    for (int l = 1; l < 16; l++) {
                        code |= (int)((bb >> (l - 1)) & 1);
                        const int c = (int)u32(S.dist_count[l]);
                        if (code - c < first) { dused = l; dsym = (int)u32(S.dist_sym[index + (code - first)]); break; }
                        index += c; first += c; first <<= 1; code <<= 1;
                    }
                }
                if (dsym < 0 || dsym >= 30) { err = UVC_INFL_ECODE; break; }
                bb >>= dused; bc -= dused;
                const int dext = dist_ext(dsym);
                const uint32_t dist = dist_base(dsym) + (uint32_t)(bb & ((1u << dext) - 1));
                bb >>= dext; bc -= dext;
                if (dist > op) { err = UVC_INFL_EDIST; break; }
                if (len > out_len - op) { err = UVC_INFL_EOUTPUT; break; }
                if (pend_len) { if (lane < pend_len) out[pend_op + lane] = pend_byte; pend_len = 0; }   // (this match may read those bytes)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint8_t *const src = out + op - dist;
                if (len <= 64) { if (lane < len) pend_byte = src[dist >= len ? lane : lane % dist]; pend_op = op; pend_len = len; }
                else if (dist >= len) { for (uint32_t k = lane; k < len; k += 64) out[op + k] = src[k]; }
                else { for (uint32_t k = lane; k < len; k += 64) out[op + k] = src[k % dist]; }
                op += len;
            }
            if (pend_len && lane < pend_len) out[pend_op + lane] = pend_byte;
            if (err) break;
            const uint32_t endpos = wp * 8 - (uint32_t)bc;   // first unread bit
            if (endpos > in_len * 8) { err = UVC_INFL_EINPUT; break; }
            // hand the position back to the reader of the state machine (headers, stored blocks)
            ip = endpos >> 3; ahead = 0; ahead_bits = 0; bitbuf = 0; bitcnt = 0;
            UVC_INFL_REFILL();
            UVC_INFL_TAKE((int)(endpos & 7));   // (endpos <= in_len * 8: the byte that holds these bits was there to load)
            continue;
        }
#endif
        if (phase == PH_COPY) {   // at most 8 bytes of the pending match
            uint32_t n = copy_len < 8 ? copy_len : 8;
            if (n == 8 && copy_dist >= 8) { uint64_t w_; __builtin_memcpy(&w_, out + op - copy_dist, 8); __builtin_memcpy(out + op, &w_, 8); op += 8; }   // source and destination do not overlap
            else for (uint32_t k = 0; k < n; k++) { out[op] = out[op - copy_dist]; op++; }
            copy_len -= n;
            if (copy_len == 0) phase = PH_SYMBOL;
            continue;
        }
        UVC_INFL_REFILL();
        if (phase == PH_SYMBOL) {
            int sym, used;
            const uint16_t e = S.lit_fast[bitbuf & ((1u << LB) - 1)];
            if (e) { used = e >> 9; sym = e & 511; } else sym = slow(&S.lit_count[0], &S.lit_sym[0], used);
            if (sym < 0 || used > bitcnt) { err = (sym < 0 ? UVC_INFL_ECODE : UVC_INFL_EINPUT); break; }
            UVC_INFL_TAKE(used);
            if (sym < 256) { if (op >= out_len) { err = UVC_INFL_EOUTPUT; break; } if (!COOP || lane == 0) out[op] = (uint8_t)sym; op++; continue; }
            if (sym == 256) { phase = last ? PH_DONE : PH_HEADER; continue; }
            sym -= 257;
            if (sym >= 29) { err = UVC_INFL_ECODE; break; }
            const int lext = len_ext(sym);
            uint32_t len = len_base(sym) + (uint32_t)(bitbuf & ((1u << lext) - 1));
            UVC_INFL_TAKE(lext);
            UVC_INFL_REFILL();
            int dsym, dused;
            const uint16_t de = S.dist_fast[bitbuf & ((1u << DB) - 1)];
            if (de) { dused = de >> 5; dsym = de & 31; } else dsym = slow(&S.dist_count[0], &S.dist_sym[0], dused);
            if (dsym < 0 || dsym >= 30) { err = UVC_INFL_ECODE; break; }
            UVC_INFL_TAKE(dused);
            const int dext = dist_ext(dsym);
            const uint32_t dist = dist_base(dsym) + (uint32_t)(bitbuf & ((1u << dext) - 1));
            UVC_INFL_TAKE(dext);
            if (bitcnt < 0) { err = UVC_INFL_EINPUT; break; }
            if (dist > op) { err = UVC_INFL_EDIST; break; }
            if (len > out_len - op) { err = UVC_INFL_EOUTPUT; break; }
            if (COOP) {
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
                const uint8_t *const src = out + op - dist;   // everything in [op - dist, op) was written by earlier steps
                if (dist >= len) { for (uint32_t k = lane; k < len; k += 64) out[op + k] = src[k]; }
                else { for (uint32_t k = lane; k < len; k += 64) out[op + k] = src[k % dist]; }
                op += len;
                continue;
            }
            copy_len = len; copy_dist = dist; phase = PH_COPY;
            continue;
        }
        if (phase == PH_HEADER) {
            if (bitcnt < 3) { err = UVC_INFL_EINPUT; break; }
            last = (int)(bitbuf & 1); const int type = (int)((bitbuf >> 1) & 3);
            UVC_INFL_TAKE(3);
            if (type == 0) {   // stored: to the byte boundary, LEN, NLEN
                UVC_INFL_TAKE(bitcnt & 7);
                UVC_INFL_REFILL();
                if (bitcnt < 32) { err = UVC_INFL_EINPUT; break; }
                const uint32_t len = (uint32_t)(bitbuf & 0xFFFF), nlen_ = (uint32_t)((bitbuf >> 16) & 0xFFFF);
                UVC_INFL_TAKE(32);
                if (len != (~nlen_ & 0xFFFF)) { err = UVC_INFL_ECODE; break; }
                stored_left = len; phase = (len ? PH_STORED : (last ? PH_DONE : PH_HEADER));
            } else if (type == 1) {   // fixed codes
                for (int i = 0; i < 144; i++) S.lens[i] = 8;
                for (int i = 144; i < 256; i++) S.lens[i] = 9;
                for (int i = 256; i < 280; i++) S.lens[i] = 7;
                for (int i = 280; i < 288; i++) S.lens[i] = 8;
                uvc_infl_build(&S.lens[0], 288, &S.lit_count[0], &S.lit_sym[0], &S.lit_fast[0], LB, 9);
                for (int i = 0; i < 30; i++) S.lens[i] = 5;
                uvc_infl_build(&S.lens[0], 30, &S.dist_count[0], &S.dist_sym[0], &S.dist_fast[0], DB, 5);
                phase = PH_SYMBOL;
            } else if (type == 2) {   // dynamic codes: HLIT, HDIST, HCLEN and the code-length code
                if (bitcnt < 14) { err = UVC_INFL_EINPUT; break; }
                nlen = (int)(bitbuf & 31) + 257; ndist = (int)((bitbuf >> 5) & 31) + 1; const int ncode = (int)((bitbuf >> 10) & 15) + 4;
                UVC_INFL_TAKE(14);
                if (nlen > 286 || ndist > 30) { err = UVC_INFL_ECODE; break; }
                uint8_t cl[19];
                for (int i = 0; i < 19; i++) cl[i] = 0;
                for (int i = 0; i < ncode; i++) { UVC_INFL_REFILL(); if (bitcnt < 3) { err = UVC_INFL_EINPUT; break; } cl[CLORDER[i]] = (uint8_t)(bitbuf & 7); UVC_INFL_TAKE(3); }
                if (err) break;
                uint16_t dummy_fast[2];
                if (uvc_infl_build(&cl[0], 19, &cl_count[0], &cl_sym[0], &dummy_fast[0], 0, 0)) { err = UVC_INFL_ECODE; break; }
                lens_at = 0; phase = PH_LENS;
            } else { err = UVC_INFL_ECODE; break; }
            continue;
        }
        if (phase == PH_STORED) {   // up to 4 bytes of a stored block (the bit buffer is byte-aligned here and holds at least 32 bits while input lasts)
            uint32_t n = stored_left < 4 ? stored_left : 4;
            if (n > out_len - op) { err = UVC_INFL_EOUTPUT; break; }
            if ((int)(n * 8) > bitcnt) { err = UVC_INFL_EINPUT; break; }
            for (uint32_t k = 0; k < n; k++) { if (!COOP || lane == 0) out[op] = (uint8_t)(bitbuf & 0xFF); op++; UVC_INFL_TAKE(8); }
            stored_left -= n;
            if (stored_left == 0) phase = last ? PH_DONE : PH_HEADER;
            continue;
        }
        if (phase == PH_LENS) {   // one code-length symbol of the dynamic header
            int used; const int sym = slow(&cl_count[0], &cl_sym[0], used);
            if (sym < 0 || used > bitcnt) { err = (sym < 0 ? UVC_INFL_ECODE : UVC_INFL_EINPUT); break; }
            UVC_INFL_TAKE(used);
            if (sym < 16) S.lens[lens_at++] = (uint8_t)sym;
            else {
                int prev = 0, rep;
                if (sym == 16) { if (lens_at == 0) { err = UVC_INFL_ECODE; break; } prev = S.lens[lens_at - 1]; rep = 3 + (int)(bitbuf & 3); UVC_INFL_TAKE(2); }
                else if (sym == 17) { rep = 3 + (int)(bitbuf & 7); UVC_INFL_TAKE(3); }
                else { rep = 11 + (int)(bitbuf & 127); UVC_INFL_TAKE(7); }
                if (bitcnt < 0) { err = UVC_INFL_EINPUT; break; }
                if (lens_at + rep > nlen + ndist) { err = UVC_INFL_ECODE; break; }
                for (int k = 0; k < rep; k++) S.lens[lens_at++] = (uint8_t)prev;
            }
            if (lens_at == nlen + ndist) {
                if (S.lens[256] == 0) { err = UVC_INFL_ECODE; break; }   // no end-of-block code
                if (uvc_infl_build(&S.lens[0], nlen, &S.lit_count[0], &S.lit_sym[0], &S.lit_fast[0], LB, 9) || uvc_infl_build(&S.lens[0] + nlen, ndist, &S.dist_count[0], &S.dist_sym[0], &S.dist_fast[0], DB, 5)) { err = UVC_INFL_ECODE; break; }
                phase = PH_SYMBOL;
            }
            continue;
        }
    }
#undef UVC_INFL_REFILL
#undef UVC_INFL_LOAD_AHEAD
#undef UVC_INFL_TAKE
    if (err) return err;
    if (phase != PH_DONE) return UVC_INFL_EINPUT;     // (the iteration bound: cannot be reached by a well-formed stream)
    return op == out_len ? UVC_INFL_OK : UVC_INFL_ESHORT;
}
template <class ST>
UVC_HD int uvc_inflate_block(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_len, ST &S) { return uvc_inflate_block_t<false, UVC_INFL_LBITS, UVC_INFL_DBITS>(in, in_len, out, out_len, S, 0u); }

#endif
