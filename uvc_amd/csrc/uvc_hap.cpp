// uvc_hap.cpp -- the host half of the haplotype links: from the per-fragment / per-unit event lists the device makes
// (k_hap_frags / k_hap_units) to the HapLink vectors of updateByRegion3Aln (main.hpp:3665-3742): the three mutform -> [forward, reverse]
// maps (mutform2count4map_bq / _fq / _f2q, main.hpp:2734-2737, 3514-3521) and updateHapMap (main.hpp:3596-3663), then per record the
// phase strings of FORMAT/bHap, cHap, c2Hap (mutform2count4vec_to_simplemut2indices, main.cpp:82-97; mutform2count4map_to_phase,
// main.hpp:5380-5404).  A few thousand short lists per region: plain C++ containers, as in the reference.
#include "uvc_hap.h"

#include <algorithm>
#include <array>
#include <map>
#include <tuple>

typedef std::vector<std::pair<int32_t, int32_t>> MutForm;   // (refpos, symbol): compares like the reference's basic_string of pairs

void uvc_hap_build(const int32_t *events, int64_t n_ints, int32_t beg, int64_t npos, int32_t max_count, int32_t min_ad, int32_t max_detail_cnt, std::vector<UvcHapLinkHost> out[3]) {
    std::map<MutForm, std::array<int32_t, 2>> maps[3];
    for (int64_t at = 0; at + 2 <= n_ints;) {   // [strand | kind << 1, count, events...]; the slot is as long as the object's bound, the walk follows the slots
        const int32_t head = events[at], count = events[at + 1];
        if (head < 0) break;                     // (untouched tail)
        const int kind = (head >> 1) & 3, strand = head & 1;
        if (count > 1 && kind < 3) {
            MutForm f; f.reserve((size_t)count);
            for (int32_t k = 0; k < count; k++) { const int32_t v = events[at + 2 + k]; f.emplace_back(beg + (v >> 4), v & 15); }
            maps[kind][f][strand]++;
        }
        at += 2 + (head >> 8);                   // the slot length rides in the header's upper bits
    }
    for (int w = 0; w < 3; w++) {
        out[w].clear();
        typedef std::tuple<int32_t, MutForm, std::array<int32_t, 2>> Row;
        std::vector<Row> v;
        for (const auto &it : maps[w]) v.push_back(Row(it.second[0] + it.second[1], it.first, it.second));
        std::sort(v.rbegin(), v.rend());
        const size_t num_dst = std::min((size_t)std::max(max_detail_cnt, 0), v.size());
        std::vector<int32_t> inc_fw(num_dst, 0), inc_rv(num_dst, 0);
        for (size_t i = 0; i < num_dst; i++) {
            const MutForm &dst = std::get<1>(v[i]);
            for (size_t j = i + 1; j < v.size(); j++) {
                const MutForm &src = std::get<1>(v[j]);
                bool skipped = false;
                for (const auto &al : dst) if (std::find(src.begin(), src.end(), al) == src.end()) { skipped = true; break; }
                if (!skipped) { inc_fw[i] += std::get<2>(v[j])[0]; inc_rv[i] += std::get<2>(v[j])[1]; }
            }
        }
        std::vector<int32_t> tsum((size_t)npos + 1, 0);
        for (size_t i = 0; i < v.size(); i++) {
            const MutForm &form = std::get<1>(v[i]);
            const std::array<int32_t, 2> &cnt = std::get<2>(v[i]);
            if ((cnt[0] + cnt[1]) < (min_ad + (int32_t)form.size())) continue;
            int32_t tot = 0;
            for (const auto &sm : form) { const size_t x = (size_t)(sm.first - beg); if (x < tsum.size()) { tsum[x] += 1; tot += tsum[x]; } }
            if ((int64_t)tot > (int64_t)max_count * (int64_t)form.size()) continue;
            UvcHapLinkHost h; h.form = form; h.fr[0] = cnt[0]; h.fr[1] = cnt[1];
            h.other[0] = (i >= num_dst ? -1 : inc_fw[i]); h.other[1] = (i >= num_dst ? -1 : inc_rv[i]);
            out[w].push_back(std::move(h));
        }
    }
}

std::string uvc_hap_phase_string(const std::vector<UvcHapLinkHost> &links, int32_t refpos, int32_t symbol) {
    static const char *const DESC[] = { "A", "C", "G", "T", "N", "*", "<LR>", "<LD3P>", "<LD2>", "<LD1>", "<LI3P>", "<LI2>", "<LI1>", "*" };
    std::string out;
    for (const UvcHapLinkHost &h : links) {
        if (h.fr[0] + h.fr[1] < 2) continue;
        if (std::find(h.form.begin(), h.form.end(), std::make_pair(refpos, symbol)) == h.form.end()) continue;
        out += "(";
        for (const auto &ps : h.form) { out += "("; out += std::to_string(ps.first + (ps.second <= 5 ? 1 : 0)); out += "&"; out += DESC[ps.second]; out += ")"; }
        out += "&"; out += std::to_string(h.fr[0]); out += "&"; out += std::to_string(h.fr[1]);
        if (-1 < h.other[0]) { out += "&&"; out += std::to_string(h.other[0] + h.fr[0]); out += "&"; out += std::to_string(h.other[1] + h.fr[1]); }
        out += ")";
    }
    return out;
}
