// uvc_hap.h -- host side of the haplotype links (uvc_hap.cpp); shared by uvc_host.cpp (builds them) and uvc_vcf.cpp (prints them)
#ifndef UVC_HAP_H
#define UVC_HAP_H
#include <stdint.h>
#include <string>
#include <utility>
#include <vector>
struct UvcHapLinkHost { std::vector<std::pair<int32_t, int32_t>> form; int32_t fr[2]; int32_t other[2]; };   // HapLink, main.hpp:33-46
// events: the device's lists back to back -- [strand | kind << 1 | slot_len << 8, count, (x << 4 | symbol) x slot_len]
void uvc_hap_build(const int32_t *events, int64_t n_ints, int32_t beg, int64_t npos, int32_t max_count, int32_t min_ad, int32_t max_detail_cnt, std::vector<UvcHapLinkHost> out[3]);
std::string uvc_hap_phase_string(const std::vector<UvcHapLinkHost> &links, int32_t refpos, int32_t symbol);
#endif
