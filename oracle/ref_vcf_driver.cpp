// TEST INFRASTRUCTURE (checker only).  A thin C ABI over the REFERENCE's own generated record layout: oracle/_ref/bcf_formats.step1.hpp is
// the output of the reference's bcf_formats_generator1.cpp, compiled from where it lies and run as the reference's Makefile:55-59 does.
// bcfrec::BcfFormat, bcfrec::streamAppendBcfFormat, FORMAT_STRING_PER_REC*, FORMAT_LINES and FILTER_* below are therefore reference code,
// not a restatement: what this library prints is what the reference prints for the same field values.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "_ref/bcf_formats.step1.hpp"
#include "_ref/ref_vcf_setters.inc"

extern "C" {
const char *uvc_ref_format_string(int tier2) { return tier2 ? bcfrec::FORMAT_STRING_PER_REC : bcfrec::FORMAT_STRING_PER_REC_WITHOUT_SSCS; }
int uvc_ref_n_format(void) { return (int)bcfrec::FORMAT_NUM; }
const char *uvc_ref_format_id(int i) { return bcfrec::FORMAT_IDS[i]; }
const char *uvc_ref_format_line(int i) { return bcfrec::FORMAT_LINES[i]; }
int uvc_ref_n_filter(void) { return (int)bcfrec::FILTER_NUM; }
const char *uvc_ref_filter_id(int i) { return bcfrec::FILTER_IDS[i]; }
const char *uvc_ref_filter_line(int i) { return bcfrec::FILTER_LINES[i]; }

// spec: one line per tag, "TAG \t N \t v1 \x1f v2 ... \n" (N values; N = 0 leaves the field at the reference's default),
// plus the pseudo tag "enable_tier2_consensus_format_tags".  Returns the length of the text, -1 on an unknown tag, -2 if `out` is too small.
int64_t uvc_ref_stream_format(const char *spec, char *out, int64_t cap) {
    bcfrec::BcfFormat f;
    const char *p = spec;
    while (*p) {
        const char *e = strchr(p, '\n'); if (!e) e = p + strlen(p);
        std::string line(p, e); p = (*e ? e + 1 : e);
        if (line.empty()) continue;
        const size_t t1 = line.find('\t'), t2 = line.find('\t', t1 + 1);
        if (t1 == std::string::npos || t2 == std::string::npos) return -1;
        const std::string tag = line.substr(0, t1);
        const long nv = strtol(line.substr(t1 + 1, t2 - t1 - 1).c_str(), nullptr, 10);
        std::vector<std::string> v; std::string rest = line.substr(t2 + 1); size_t at = 0;
        for (long i = 0; i < nv; i++) { size_t s = rest.find('\x1f', at); if (s == std::string::npos) s = rest.size(); v.push_back(rest.substr(at, s - at)); at = std::min(rest.size(), s + 1); }
        if (tag == "enable_tier2_consensus_format_tags") { f.enable_tier2_consensus_format_tags = (nv > 0 && v[0] == "1"); continue; }
        if (!set_field(f, tag, v)) return -1;
    }
    std::string s;
    bcfrec::streamAppendBcfFormat(s, f);
    if ((int64_t)s.size() + 1 > cap) return -2;
    memcpy(out, s.c_str(), s.size() + 1);
    return (int64_t)s.size();
}
}
