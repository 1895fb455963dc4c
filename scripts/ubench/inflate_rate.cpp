// One core: the library's DEFLATE decoder (uvc_inflate_fast.h, or another version of it: -DHDR='"path"') against zlib on the BGZF blocks of a
// BAM file, with a byte comparison of every block first.
//   g++ -O3 -std=c++17 -Iuvc_amd/csrc -o /tmp/inflate_rate scripts/ubench/inflate_rate.cpp -lz && /tmp/inflate_rate file.bam
#ifndef HDR
#define HDR "uvc_inflate_fast.h"
#endif
#include HDR
#include <zlib.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: inflate_rate file.bam\n"); return 2; }
    FILE *f = fopen(argv[1], "rb"); if (!f) { perror(argv[1]); return 2; }
    std::vector<uint8_t> raw; { static uint8_t b[1 << 16]; size_t k; while ((k = fread(b, 1, sizeof b, f)) > 0) raw.insert(raw.end(), b, b + k); }
    fclose(f);
    const size_t n = raw.size(); raw.resize(n + 64);
    struct B { size_t off, clen, isize; }; std::vector<B> bl; size_t o = 0, tot = 0;
    while (o + 18 <= n) {
        const size_t xlen = raw[o + 10] | (raw[o + 11] << 8), bs = (size_t)(raw[o + 16] | (raw[o + 17] << 8)) + 1;
        const size_t is = raw[o + bs - 4] | (raw[o + bs - 3] << 8) | (raw[o + bs - 2] << 16) | ((size_t)raw[o + bs - 1] << 24);
        bl.push_back({ o + 12 + xlen, bs - 12 - xlen - 8, is }); tot += is; o += bs;
    }
    std::vector<uint8_t> out(70000), ref(70000);
    auto zl = [&](const B &b, uint8_t *dst) { z_stream z{}; inflateInit2(&z, -15); z.next_in = raw.data() + b.off; z.avail_in = (uInt)b.clen; z.next_out = dst; z.avail_out = 70000; const int rc = inflate(&z, Z_FINISH); const size_t got = z.total_out; inflateEnd(&z); return rc == Z_STREAM_END && got == b.isize; };
    size_t bad = 0;
    for (auto &b : bl) { const bool ok = uvc_fast_inflate::inflate(raw.data() + b.off, b.clen, out.data(), b.isize); if (!ok || !zl(b, ref.data()) || memcmp(out.data(), ref.data(), b.isize)) bad++; }
    double best = 1e9, bestz = 1e9;
    for (int rep = 0; rep < 7; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        for (auto &b : bl) uvc_fast_inflate::inflate(raw.data() + b.off, b.clen, out.data(), b.isize);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); if (dt < best) best = dt;
        t0 = std::chrono::steady_clock::now();
        for (auto &b : bl) zl(b, ref.data());
        dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); if (dt < bestz) bestz = dt;
    }
    printf("%s: %zu blocks, %.1f MB; differing from zlib: %zu; own %.3f s = %.0f MB/s, zlib %.3f s = %.0f MB/s (%.2f x)\n", HDR, bl.size(), tot / 1e6, bad, best, tot / best / 1e6, bestz, tot / bestz / 1e6, bestz / best);
    return bad != 0;
}
