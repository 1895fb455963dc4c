// uvc_hap.cpp -- the host half of the haplotype links: from the per-fragment / per-unit event lists the device makes
// (k_hap_frags / k_hap_units) to the HapLink vectors of updateByRegion3Aln (main.hpp:3665-3742): the three mutform -> [forward, reverse]
// maps (mutform2count4map_bq / _fq / _f2q, main.hpp:2734-2737, 3514-3521) and updateHapMap (main.hpp:3596-3663), then per record the
// phase strings of FORMAT/bHap, cHap, c2Hap (mutform2count4vec_to_simplemut2indices, main.cpp:82-97; mutform2count4map_to_phase,
// main.hpp:5380-5404).  A few thousand short lists per region: plain C++ containers, as in the reference.
#include "uvc_hap.h"

#include <algorithm>
#include <array>
#include <cstddef>
#include <cstring>
#include <map>
#include <tuple>

typedef std::vector<std::pair<int32_t, int32_t>> MutForm;   // (refpos, symbol): compares like the reference's basic_string of pairs

void uvc_hap_build(const int32_t *events, int64_t n_ints, int32_t beg, int64_t npos, int32_t max_count, int32_t min_ad, int32_t max_detail_cnt, std::vector<UvcHapLinkHost> out[3]) {
    // A list is the run of event words (position offset << 4 | symbol, ascending): comparing the words compares the (refpos, symbol) pairs, so
    // the reference's ordered maps and its sort of (count, list, [forward, reverse]) rows can work on the raw words.  Most lists occur once (two
    // sequencing errors in one fragment) and never reach the output: they are counted in an open-addressing table and only looked at again
    // as "other" support of the few detailed links; no per-list containers.
    struct Slot { int64_t at; int32_t cnt[2]; };
    size_t n_lists = 0;
    for (int64_t at = 0; at + 2 <= n_ints;) { const int32_t head = events[at]; if (head < 0) break; if (events[at + 1] > 1 && ((head >> 1) & 3) < 3) n_lists++; at += 2 + (head >> 8); }
    size_t cap = 64; while (cap < 2 * n_lists + 2) cap <<= 1;
    std::vector<Slot> table(cap, Slot{ -1, { 0, 0 } });
    for (int64_t at = 0; at + 2 <= n_ints;) {   // [strand | kind << 1, count, events...]; the slot is as long as the object's bound, the walk follows the slots
        const int32_t head = events[at], count = events[at + 1];
        if (head < 0) break;                     // (untouched tail)
        const int kind = (head >> 1) & 3, strand = head & 1;
        if (count > 1 && kind < 3) {
            uint64_t h = 1469598103934665603ULL ^ (uint64_t)kind;
            for (int32_t k = 0; k < count; k++) { h ^= (uint32_t)events[at + 2 + k]; h *= 1099511628211ULL; }
            size_t i = (size_t)(h ^ (h >> 29)) & (cap - 1);
            for (;; i = (i + 1) & (cap - 1)) {
                Slot &sl = table[i];
                if (sl.at < 0) { sl.at = at; sl.cnt[strand] = 1; break; }
                const int32_t head2 = events[sl.at];
                if (((head2 >> 1) & 3) == kind && events[sl.at + 1] == count && !memcmp(&events[sl.at + 2], &events[at + 2], sizeof(int32_t) * (size_t)count)) { sl.cnt[strand]++; break; }
            }
        }
        at += 2 + (head >> 8);                   // the slot length rides in the header's upper bits
    }
    std::vector<Slot> rows[3];
    for (const Slot &sl : table) if (sl.at >= 0) rows[(events[sl.at] >> 1) & 3].push_back(sl);
    auto n_of = [&](const Slot &a) { return events[a.at + 1]; };
    auto form_cmp = [&](const Slot &a, const Slot &b) {   // lexicographic, like operator< of the reference's basic_string of pairs
        const int32_t na = n_of(a), nb = n_of(b);
        for (int32_t k = 0; k < std::min(na, nb); k++) { const int32_t va = events[a.at + 2 + k], vb = events[b.at + 2 + k]; if (va != vb) return va < vb ? -1 : 1; }
        return na < nb ? -1 : (na > nb ? 1 : 0);
    };
    auto before = [&](const Slot &a, const Slot &b) {     // descending (count, list, [forward, reverse]): std::sort over reverse iterators, main.hpp:3607-3613
        const int32_t ta = a.cnt[0] + a.cnt[1], tb = b.cnt[0] + b.cnt[1];
        if (ta != tb) return ta > tb;
        const int c = form_cmp(a, b);
        if (c) return c > 0;
        if (a.cnt[0] != b.cnt[0]) return a.cnt[0] > b.cnt[0];
        return a.cnt[1] > b.cnt[1];
    };
    auto contains = [&](const Slot &src, int32_t word) { const int32_t n = n_of(src); for (int32_t k = 0; k < n; k++) if (events[src.at + 2 + k] == word) return true; return false; };
    for (int w = 0; w < 3; w++) {
        out[w].clear();
        std::vector<Slot> &v = rows[w];
        const size_t num_dst = std::min((size_t)std::max(max_detail_cnt, 0), v.size());
        std::partial_sort(v.begin(), v.begin() + (std::ptrdiff_t)num_dst, v.end(), before);   // the detailed links are the first num_dst rows of the full order
        std::vector<int32_t> inc_fw(num_dst, 0), inc_rv(num_dst, 0);
        for (size_t i = 0; i < num_dst; i++) {   // support from the lists behind it that contain all of its mutations
            const int32_t nd = n_of(v[i]);
            for (size_t j = i + 1; j < v.size(); j++) {
                bool all = true;
                for (int32_t k = 0; k < nd && all; k++) all = contains(v[j], events[v[i].at + 2 + k]);
                if (all) { inc_fw[i] += v[j].cnt[0]; inc_rv[i] += v[j].cnt[1]; }
            }
        }
        // the rows that can be written at all, in the full order; the per-position budget (tsum) only counts them
        std::vector<std::pair<Slot, int32_t>> pass;   // (row, index among the detailed ones or -1)
        for (size_t i = 0; i < v.size(); i++) if ((v[i].cnt[0] + v[i].cnt[1]) >= (min_ad + n_of(v[i]))) pass.emplace_back(v[i], i < num_dst ? (int32_t)i : -1);
        std::sort(pass.begin(), pass.end(), [&](const std::pair<Slot, int32_t> &a, const std::pair<Slot, int32_t> &b) { return before(a.first, b.first); });
        std::map<int32_t, int32_t> tsum;
        for (const auto &pr : pass) {
            const Slot &sl = pr.first; const int32_t n = n_of(sl);
            int32_t tot = 0;
            for (int32_t k = 0; k < n; k++) { const int64_t x = (int64_t)(events[sl.at + 2 + k] >> 4); if (x >= 0 && x < npos + 1) { int32_t &t = tsum[(int32_t)x]; t += 1; tot += t; } }
            if ((int64_t)tot > (int64_t)max_count * (int64_t)n) continue;
            UvcHapLinkHost h; h.form.reserve((size_t)n);
            for (int32_t k = 0; k < n; k++) { const int32_t word = events[sl.at + 2 + k]; h.form.emplace_back(beg + (word >> 4), word & 15); }
            h.fr[0] = sl.cnt[0]; h.fr[1] = sl.cnt[1];
            h.other[0] = (pr.second < 0 ? -1 : inc_fw[(size_t)pr.second]); h.other[1] = (pr.second < 0 ? -1 : inc_rv[(size_t)pr.second]);
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
