// uvc_host.cpp -- C ABI of libuvcgpu.so (include/uvcgpu.h): region handles, host-side packing of the
// alns3-equivalent SoA into device records, kernel sequencing on the handle's HIP stream.
// There is no CPU compute fallback in this library: every entry point that computes launches HIP kernels.
#include "uvc_device.h"
#include "uvc_alloc.h"
#include "uvc_prep.h"
#include "uvc_rtr.h"
#include "uvc_hap.h"
#include "uvc_cpus.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>
// device memory through the caching allocator (uvc_alloc.h): freed blocks are reused without the device-wide synchronisation of hipFree
#define hipMalloc(p, n) uvc_dev_malloc((void **)(p), (n))
#define hipFree(p) uvc_dev_free((void *)(p))

struct RawReads {
    const int32_t *pos, *endpos, *mpos, *isize, *nm, *l_qseq, *n_cigar, *frag, *fs, *dflag, *kind, *fast_rank;
    const uint16_t *flag; const uint8_t *mapq;
    const int64_t *seq_off, *cigar_off, *table_off, *item_off, *gap_off;
};
extern "C" void uvc_launch_correct_bq(const RegionDev *R, int bq_max, int bq_inc, hipStream_t s);
extern "C" void uvc_launch_pack_bq(const uint8_t *bases, const uint8_t *quals, uint16_t *bq, int64_t n, int32_t *bad, hipStream_t s);
extern "C" void uvc_launch_build_p2list(const RegionDev *R, const int32_t *fast_rank, const int32_t *aln, const int32_t *cbeg, const int32_t *cend, const int32_t *qb, hipStream_t s);
extern "C" void uvc_launch_prelude(const RegionDev *R, const RawReads *W, const UvcParams *P, hipStream_t s);
struct UvcProf { int on; int n; const char *name[32]; hipEvent_t ev[32][2]; };
extern "C" void uvc_launch_accumulate(const RegionDev *R, const UvcParams *P, int half_ratio_phred,
                                      const int32_t *dup_units, int n_dup, const int64_t *dup_off, int64_t n_dup_work, hipStream_t s, UvcProf *prof,
                                      hipStream_t side, hipEvent_t e_fork, hipEvent_t e_join, hipEvent_t e_fork2, hipStream_t side3, hipEvent_t e_join3, hipEvent_t e_stat, hipEvent_t e_alleles);
extern "C" int uvc_launch_score(const RegionDev *R, const UvcParams *P, const UvcScoreRequest *req, const UvcIndelAllele *d_alleles, const int32_t *d_allele_rows, int64_t n_alleles,
                                const UvcGapRow *d_gap_rows, const uint8_t *d_gap_seq, const UvcTumorKey *d_tkeys, int32_t *d_fields, int64_t capacity, char *scratch, int32_t *d_fields_kept, hipStream_t s);
extern "C" size_t uvc_score_scratch_bytes(int64_t npos_scored, int64_t capacity);
extern "C" size_t uvc_score_scratch_zero_bytes(int64_t npos_scored, int64_t capacity);
extern "C" void uvc_launch_zero_state(char *slab, const void *planes, int n_planes, uint8_t *dirty, int ndblk, int64_t npos, hipStream_t s);
extern "C" void uvc_launch_check_dirty(const RegionDev *R, unsigned long long *d_n_bad, hipStream_t s);
extern "C" void uvc_launch_check_presence(const RegionDev *R, unsigned long long *d_n_bad, hipStream_t s);
extern "C" void uvc_launch_block_stats(const RegionDev *R, const UvcParams *P, int64_t x0, int64_t n, int32_t *d_out, hipStream_t s);
extern "C" size_t uvc_gap_sort_tmp_bytes(size_t n);
extern "C" void uvc_launch_hap_cand(const RegionDev *R, const HapWork *H, int units, hipStream_t s);
extern "C" void uvc_launch_hap_events(const RegionDev *R, const UvcParams *P, const HapWork *H, int units, int n_cand, hipStream_t s);
extern "C" void uvc_launch_gather_columns(const char *const *base, const int32_t *first_col, const int32_t *elem, int64_t npos, const int32_t *d_xs, int64_t n, long long *d_out, hipStream_t s);
extern "C" int uvc_sort_by_pos_cls(const int32_t *d_pos, const int32_t *d_cls, int32_t beg, int pos_bits, int cls_bits, int64_t n, uint32_t *work, void *tmp, size_t tmp_bytes, hipStream_t s);
extern "C" size_t uvc_sort32_tmp_bytes(size_t n);
extern "C" size_t uvc_prep_compact_tmp_bytes(int64_t n);
extern "C" int uvc_prep_compact(const int32_t *l_qseq, const int32_t *n_cigar, int64_t n, int64_t n_bases, const uint8_t *bases4, int64_t n_b4, const uint8_t *quals, int64_t *seq_off_out, const int64_t *seq_off_in,
                                int64_t *cigar_off_out, int64_t *b4_off, uint8_t *bases_out, uint16_t *bq_out, int32_t *bad, void *tmp, size_t tmp_bytes, hipStream_t s);
extern "C" void uvc_launch_gather4(const uint32_t *perm, int64_t n, const int32_t *a0, const int32_t *a1, const int32_t *a2, const int32_t *a3, int32_t *o0, int32_t *o1, int32_t *o2, int32_t *o3, hipStream_t s);
extern "C" void uvc_launch_rank_from_sorted(const uint32_t *perm, int64_t n, int64_t n_first, int32_t *out_ids, int32_t *rank, hipStream_t s);

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
// C++ exceptions must not cross the C boundary (std::terminate would take the caller's process down): every entry point that allocates
// host memory reports them as an error code instead
template <class F> static int guarded(const char *what, F &&f) {
    try {
        const int rc = f();
        // a HIP call whose failure was handled (or did not matter) still leaves its code in the thread's "last error", where the launch checks of
        // rocPRIM in a later call would find it and refuse: it ends here (UVCGPU_DEBUG names the entry point that left one)
        const hipError_t stale = hipGetLastError();
        if (stale != hipSuccess && getenv("UVCGPU_DEBUG")) fprintf(stderr, "[uvcgpu] %s (rc %d) left HIP error: %s\n", what, rc, hipGetErrorString(stale));
        return rc;
    }
    catch (const std::bad_alloc &) { return fail(UVCGPU_ENOMEM, std::string(what) + ": out of host memory"); }
    catch (const std::exception &e) { return fail(UVCGPU_EDEVICE, std::string(what) + ": " + e.what()); }
}
extern "C" int uvcgpu_set_error(int code, const char *msg) { return fail(code, msg ? msg : ""); }   // for the other translation units of the library
#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(UVCGPU_EDEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

struct uvcgpu_region {
    UvcParams P;
    int32_t tid, beg, end;     // state covers [beg, end): end = caller's end + 1 (main.cpp:569)
    int64_t npos;
    std::string refstring;
    std::vector<int32_t> h_rtr; bool h_rtr_valid = false;   // host copy of the repeat tracks as built (begpos / tracklen / unitlen planes), fetched when the record writer first asks
    UvcRtrWork rw; char *d_rtrwork = nullptr; bool thr_ready = false;   // scratch of the side-array kernels (uvc_rtr.hip)
    char *h_ref = nullptr; size_t h_ref_cap = 0;                        // page-locked staging of the reference characters
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr, side3 = nullptr; hipEvent_t e_fork = nullptr, e_join = nullptr, e_fork2 = nullptr, e_join3 = nullptr, e_stat = nullptr, e_alleles = nullptr;   // fork/join inside accumulate (see uvc_launch_accumulate)
    // device buffers
    uint8_t *d_refsym = nullptr; int32_t *d_rtr = nullptr, *d_rtr0 = nullptr, *d_fsum = nullptr, *d_win = nullptr; int64_t *d_baq = nullptr;
    char *d_state = nullptr; size_t state_bytes = 0;
    int64_t npos_cap = 0;         // positions the side arrays and the slab were allocated for (uvcgpu_region_reset reuses them)
    size_t bucket_off = 0; bool buckets_clean = false;   // the transient bucket planes (tail of the slab) are left zero by P3b / P5b
    std::vector<void *> owned;   // read-dependent device allocations
    RegionDev R;
    int32_t *d_dup_units = nullptr; int64_t *d_dup_off = nullptr; int n_dup = 0; int64_t n_dup_work = 0;
    size_t off[UVC_NUM_FIELD_GROUPS + 1];
    bool has_reads = false, accumulated = false;
    size_t p5flag_off = 0, occ_off = 0;
    // zero fill without what the last accumulate left untouched (RegionDev::dirty, uvc_launch_zero_state)
    uint8_t *d_dirty = nullptr; void *d_zero_planes = nullptr; int n_zero_planes = 0; int64_t zero_planes_npos = 0;
    int64_t dirty_npos = 0;      // the region length under which the slab's contents and the marks were written; 0: unknown, fill everything
    bool state_released = false, state_zeroed = false;   // UvcScoreRequest::release_state: planes given up / already zeroed on the side stream (e_join marks the end)
    int64_t last_scored = 0, last_returned = 0;   // record counts of the last score call (uvcgpu_region_last_score_counts)
    size_t zeroed_bytes = 0;     // with state_zeroed: the slab is zero from its start up to here (a rebind to a region that fits keeps the benefit)
    RawReads W;                  // per-read input columns on the device (kept: uvcgpu_region_correct_bq re-derives the per-read records)
    int32_t *d_p2[4] = { nullptr, nullptr, nullptr, nullptr };   // P2 work list: alignment, begin, end, query offset
    int64_t n_bases = 0;
    UvcProf prof;
    // persistent scoring buffers (grown on demand)
    char *d_score_scratch = nullptr; size_t score_scratch_bytes = 0;
    int32_t *d_score_fields = nullptr; int64_t score_capacity = 0; int64_t *d_score_count = nullptr;
    uint8_t *h_stage = nullptr; size_t h_stage_cap = 0;   // page-locked staging for the small per-call uploads of score (tumor keys, caller's alleles): never the caller's own pages
    int32_t *d_score_kept = nullptr; int64_t score_kept_capacity = 0;   // UvcScoreRequest::kept_only: the compacted copy, same pitch as d_score_fields
    // InDel allele tables of the last accumulate (built on first use by gap_tables)
    bool gap_ready = false;
    std::vector<UvcGapRow> gap_rows; std::vector<uint8_t> gap_seq;
    std::vector<UvcIndelAllele> gap_alleles; std::vector<int32_t> gap_allele_row;   // what indel_get_majority yields, sorted by (refpos, symbol)
    UvcIndelAllele *d_gap_alleles = nullptr; int32_t *d_gap_allele_row = nullptr; int64_t gap_alleles_cap = 0;
    UvcGapRow *d_gap_rows = nullptr; int64_t gap_rows_cap = 0; uint8_t *d_gap_seq = nullptr; int64_t gap_seq_cap = 0;   // device copies for k_call
    // haplotype links of the last accumulate (built on first use by hap_tables): hap_bq, hap_fq, hap_f2q of updateByRegion3Aln
    bool hap_ready = false;
    std::vector<UvcHapLinkHost> hap[3];
};

static size_t group_bytes(const uvcgpu_region *r, int g) {
    const size_t n = (size_t)r->npos;
    switch (g) {
        case UVC_F_PREP32: return 4 * n * UVC_NPREP32;
        case UVC_F_PREP64: return 8 * n * UVC_NPREP64;
        case UVC_F_THRES: return 4 * n * UVC_NTHRES;
        case UVC_F_SEG32: return 4 * n * UVC_NSEG32 * NSYM;
        case UVC_F_SEG64: return 8 * n * UVC_NSEG64 * NSYM;
        case UVC_F_VQ: return 4 * n * UVC_NVQ * NSYM;
        case UVC_F_BQSUM: return 4 * n * NSYM;
        case UVC_F_FRAG: return 4 * n * 2 * UVC_NFRAG * NSYM;
        case UVC_F_FAM: return 4 * n * 2 * UVC_NFAM * NSYM;
        case UVC_F_FAMINFO32: return 4 * n * UVC_NFAMINFO32 * NSYM;
        case UVC_F_FAMINFO64: return 8 * n * UVC_NFAMINFO64 * NSYM;
        case UVC_F_DUPLEX: return 4 * n * UVC_NDUPLEX * NSYM;
        case UVC_F_RTR: return 4 * n * UVC_NRTR;
        case UVC_F_BAQ: return 8 * n * 2;
    }
    return 0;
}

// ---------------------------------------------------------------- region side arrays (a3) ---
// refstring2repeatvec (main.hpp:803-874) + region_repeatvec_to_baq_offsetarr (main.cpp:400-429) run on the device (uvc_rtr.hip); the host
// only stages the reference characters and, once per handle, turns indel_phred (main.hpp:794-801) into its threshold table.
namespace {
template <class T> int upload(uvcgpu_region *r, const std::vector<T> &v, T **out, bool owned = true) {
    *out = nullptr;
    const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    hipError_t e = hipMalloc((void **)out, bytes);
    if (e != hipSuccess) return fail(UVCGPU_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    if (owned) r->owned.push_back(*out);
    if (!v.empty()) { e = hipMemcpyAsync(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, r->stream); if (e != hipSuccess) return fail(UVCGPU_EDEVICE, hipGetErrorString(e)); }
    return 0;
}
// device allocation without a host image (the prelude kernels write every element); optionally zero-filled on the stream
template <class T> int dev_alloc(uvcgpu_region *r, size_t count, T **out, bool zero = false) {
    *out = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc((void **)out, bytes);
    if (e != hipSuccess) return fail(UVCGPU_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    r->owned.push_back(*out);
    if (zero && hipMemsetAsync(*out, 0, bytes, r->stream) != hipSuccess) return fail(UVCGPU_EDEVICE, "hipMemsetAsync");
    return 0;
}
// straight from the caller's array (it stays valid until set_reads returns, which synchronises the stream)
template <class T> int upload_raw(uvcgpu_region *r, const T *src, size_t count, T **out) {
    int rc = dev_alloc(r, count, out);
    if (rc) return rc;
    if (count && hipMemcpyAsync(*out, src, count * sizeof(T), hipMemcpyHostToDevice, r->stream) != hipSuccess) return fail(UVCGPU_EDEVICE, "hipMemcpyAsync(H2D)");
    return 0;
}
// (the cache hands freed blocks to other handles at once: nothing of this handle may still be running on them)
void quiesce(uvcgpu_region *r) { if (r->stream) (void)hipStreamSynchronize(r->stream); if (r->side) (void)hipStreamSynchronize(r->side); if (r->side3) (void)hipStreamSynchronize(r->side3); }
void free_reads(uvcgpu_region *r) { if (!r->owned.empty()) quiesce(r); for (void *p : r->owned) hipFree(p); r->owned.clear(); r->has_reads = false; r->accumulated = false; }
}  // namespace

extern "C" {

const char *uvcgpu_last_error(void) { return g_err.c_str(); }
const char *uvcgpu_version(void) { return "uvcgpu 0.1 (gfx950)"; }

int uvcgpu_init(int device_id) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(UVCGPU_EDEVICE, "no HIP device: libuvcgpu has no CPU fallback");
    HIP_OK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, device_id));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) return fail(UVCGPU_EDEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    return 0;
}

int uvcgpu_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; } return n; }

void uvcgpu_params_default(UvcParams *p) {
    memset(p, 0, sizeof(*p));
    p->struct_size = (int32_t)sizeof(UvcParams);
#define UVC_PI(name, dflt) p->name = (int32_t)(dflt);
#define UVC_PD(name, dflt) p->name = (double)(dflt);
#include "uvc_params.def"
#undef UVC_PI
#undef UVC_PD
}

void uvcgpu_params_apply_platform(UvcParams *p, int32_t platform, int32_t central_readlen, int32_t max_mapq) {   // CmdLineArgs.cpp:13-15, 50-134
    p->inferred_sequencing_platform = platform;
    if (0 == p->central_readlen) p->central_readlen = central_readlen;
    p->inferred_maxMQ = std::max(p->inferred_maxMQ, max_mapq);
    auto dec = [](int32_t &a, int32_t b) { a = a - std::min(a, b); };
    if (platform == UVC_PLATFORM_IONTORRENT) {
        p->bq_phred_added_misma += 8;
        dec(p->fam_thres_highBQ_snv, 30); dec(p->fam_thres_highBQ_indel, 30); dec(p->bias_thres_PFBQ1, 30); dec(p->bias_thres_PFBQ2, 30); dec(p->bias_thres_highBQ, 13);
    } else if (platform == UVC_PLATFORM_ILLUMINA) {
        p->syserr_minABQ_pcr_snv += 200; p->syserr_minABQ_pcr_indel += 100; p->syserr_minABQ_cap_snv += 200; p->syserr_minABQ_cap_indel += 100;
    }
}

// (Re)binds a handle to a region: side arrays of the reference (C10) and the plane layout.  Device buffers are kept when they are large
// enough, so that a caller that streams tiles of one size through a handle pays hipMalloc / hipFree once (uvcgpu_region_reset).
static int configure_region(uvcgpu_region *r, int32_t tid, int32_t beg, int32_t end, const char *refseq) {
    r->tid = tid; r->beg = beg; r->end = end + 1; r->npos = (int64_t)end - beg + 1;
    r->refstring.assign(refseq, (size_t)(end - beg));
    r->accumulated = false; r->gap_ready = false; r->hap_ready = false; r->state_released = false;
    r->h_rtr_valid = false;
    // one slab for all per-position planes (+ the transient bucket planes), 8-byte groups first
    const int order[] = { UVC_F_PREP64, UVC_F_SEG64, UVC_F_FAMINFO64, UVC_F_PREP32, UVC_F_THRES, UVC_F_SEG32, UVC_F_VQ, UVC_F_BQSUM, UVC_F_FRAG, UVC_F_FAM, UVC_F_FAMINFO32, UVC_F_DUPLEX };
    size_t o = 0;
    for (int g : order) { r->off[g] = o; o += group_bytes(r, g); }
    r->p5flag_off = o; o += ((size_t)2 * r->npos + 255) & ~(size_t)255;   // one byte per (strand, position): a P5 bucket was filled (k_p5b skips the others); zeroed with the planes
    r->occ_off = o; o += ((size_t)4 * r->npos + 255) & ~(size_t)255;   // RegionDev::occ, zeroed with the planes
    r->bucket_off = o; o += (size_t)4 * r->npos * 2 * NSYM * NBUCKETS;
    r->state_bytes = o;
    // planes that the last score zeroed on the side stream (release_state) stay zero under the new layout when it fits into what was zeroed
    r->state_zeroed = r->state_zeroed && r->npos <= r->npos_cap && r->state_bytes <= r->zeroed_bytes;
    r->buckets_clean = r->state_zeroed;
    const int vmax = r->P.indel_vntr_repeatsize_max, smax = r->P.indel_str_repeatsize_max, bqm = r->P.indel_BQ_max;
    const size_t n_rtr = (size_t)UVC_NRTR * r->npos;
    if (r->npos > r->npos_cap) {
        quiesce(r);
        for (void *p : { (void *)r->d_refsym, (void *)r->d_rtr, (void *)r->d_rtr0, (void *)r->d_fsum, (void *)r->d_win, (void *)r->d_baq, (void *)r->d_state, (void *)r->d_rtrwork, (void *)r->d_dirty }) if (p) hipFree(p);
        r->d_dirty = nullptr; r->dirty_npos = 0;
        r->d_refsym = nullptr; r->d_rtr = r->d_rtr0 = r->d_fsum = r->d_win = nullptr; r->d_baq = nullptr; r->d_state = nullptr; r->d_rtrwork = nullptr; r->npos_cap = 0; r->thr_ready = false;
        size_t scan_tmp = 0;
        const size_t work_bytes = uvc_rtr_work_bytes(r->npos, vmax, smax, bqm, &scan_tmp);
        if (hipMalloc((void **)&r->d_refsym, (size_t)r->npos + 1) != hipSuccess || hipMalloc((void **)&r->d_rtr, sizeof(int32_t) * n_rtr) != hipSuccess
            || hipMalloc((void **)&r->d_rtr0, sizeof(int32_t) * n_rtr) != hipSuccess || hipMalloc((void **)&r->d_fsum, sizeof(int32_t) * 2 * UVC_FSUM_N * (size_t)r->npos) != hipSuccess || hipMalloc((void **)&r->d_win, sizeof(int32_t) * 16 * (size_t)((r->npos + 63) / 64)) != hipSuccess || hipMalloc((void **)&r->d_baq, sizeof(int64_t) * 2 * (size_t)r->npos) != hipSuccess
            || hipMalloc((void **)&r->d_state, r->state_bytes) != hipSuccess || hipMalloc((void **)&r->d_rtrwork, work_bytes) != hipSuccess
            || hipMalloc((void **)&r->d_dirty, (size_t)3 * NSYM * (size_t)((r->npos >> UVC_DIRTY_SHIFT) + 2)) != hipSuccess) return fail(UVCGPU_ENOMEM, "hipMalloc(region planes) failed");
        uvc_rtr_bind(&r->rw, r->d_rtrwork, r->npos, vmax, smax, bqm, scan_tmp);
        r->npos_cap = r->npos;
    }
    if (!r->thr_ready) {   // parameters only: once per handle (and again when the scratch moved)
        std::vector<int32_t> thr((size_t)smax * bqm);
        uvc_rtr_thresholds(&r->P, thr.data());
        HIP_OK(hipMemcpy(r->rw.thr, thr.data(), sizeof(int32_t) * thr.size(), hipMemcpyHostToDevice));
        r->thr_ready = true;
    }
    // the reference characters travel from page-locked memory of the handle, so that the copy is a stream operation like the kernels behind it
    const size_t n_ref = (size_t)(end - beg);
    if (n_ref > r->h_ref_cap) {
        if (r->h_ref) (void)hipHostFree(r->h_ref);
        r->h_ref = nullptr; r->h_ref_cap = 0;
        if (hipHostMalloc((void **)&r->h_ref, n_ref, hipHostMallocDefault) != hipSuccess) return fail(UVCGPU_ENOMEM, "hipHostMalloc(reference staging) failed");
        r->h_ref_cap = n_ref;
    }
    memcpy(r->h_ref, refseq, n_ref);
    HIP_OK(hipMemcpyAsync(r->rw.refchar, r->h_ref, n_ref, hipMemcpyHostToDevice, r->stream));
    {
        const int e = uvc_launch_region_tracks(&r->rw, &r->P, r->npos, r->d_refsym, r->d_rtr0, r->d_baq, r->stream);
        if (e) return fail(UVCGPU_EDEVICE, std::string("side-array kernels: ") + hipGetErrorString((hipError_t)e));
    }
    int32_t *d_err = r->R.err;
    RegionDev &R = r->R;
    memset(&R, 0, sizeof(R));
    R.beg = r->beg; R.end = r->end; R.npos = r->npos; R.refsym = r->d_refsym; R.rtr = r->d_rtr; R.baq = r->d_baq; R.fsum = r->d_fsum; R.win = r->d_win; R.nwin = (int32_t)((r->npos + 63) / 64);
    char *b = r->d_state;
    R.prep64 = (int64_t *)(b + r->off[UVC_F_PREP64]); R.seg64 = (int64_t *)(b + r->off[UVC_F_SEG64]); R.faminfo64 = (int64_t *)(b + r->off[UVC_F_FAMINFO64]);
    R.prep32 = (int32_t *)(b + r->off[UVC_F_PREP32]); R.thres = (int32_t *)(b + r->off[UVC_F_THRES]); R.seg32 = (int32_t *)(b + r->off[UVC_F_SEG32]);
    R.vq = (int32_t *)(b + r->off[UVC_F_VQ]); R.bqsum = (int32_t *)(b + r->off[UVC_F_BQSUM]); R.frag = (int32_t *)(b + r->off[UVC_F_FRAG]);
    R.fam = (int32_t *)(b + r->off[UVC_F_FAM]); R.faminfo32 = (int32_t *)(b + r->off[UVC_F_FAMINFO32]); R.duplex = (int32_t *)(b + r->off[UVC_F_DUPLEX]);
    R.bucket = (int32_t *)(b + r->bucket_off); R.p5flag = (uint8_t *)(b + r->p5flag_off); R.occ = (uint32_t *)(b + r->occ_off);
    R.dirty = r->d_dirty; R.ndblk = (int32_t)((r->npos + ((int64_t)1 << UVC_DIRTY_SHIFT) - 1) >> UVC_DIRTY_SHIFT);
    R.err = d_err;
    HIP_OK(hipMemsetAsync(d_err, 0, 4, r->stream));
    return 0;   // nothing to wait for: the staging memory belongs to the handle and a handle is rebound only when its streams are idle
}

static int uvcgpu_region_create_impl(uvcgpu_region_t **out, const UvcParams *params, int32_t tid, int32_t beg, int32_t end, const char *refseq) {
    if (!out || !params || !refseq || end <= beg) return fail(UVCGPU_EINVAL, "bad argument");
    if (params->struct_size != (int32_t)sizeof(UvcParams)) return fail(UVCGPU_EINVAL, "UvcParams::struct_size mismatch");
    if (params->indel_str_repeatsize_max < 1 || params->indel_vntr_repeatsize_max < params->indel_str_repeatsize_max) return fail(UVCGPU_EINVAL, "bad repeat-size parameters");
    if (params->indel_vntr_repeatsize_max > 255 || params->indel_BQ_max < 1 || params->indel_BQ_max > 32767) return fail(UVCGPU_EUNSUPPORTED, "indel_vntr_repeatsize_max > 255 or indel_BQ_max outside 1..32767");
    // dist_to_interfering_indel is 10000 where a read has no low-quality InDel (main.hpp:1897) and a difference of GENOME coordinates next to
    // the sentinels of its InDel list otherwise: a threshold above 10000 compares with those.  The kernels carry the distance in 16 bits with
    // "10000 or more" as one value, which is exact for every threshold up to 10000 (the default is 5) and wrong beyond: refused, not approximated.
    if (params->bias_thres_interfering_indel > 10000) return fail(UVCGPU_EUNSUPPORTED, "bias_thres_interfering_indel above 10000");
    uvcgpu_region *r = new uvcgpu_region();
    memset(&r->prof, 0, sizeof(r->prof));
    memset(&r->R, 0, sizeof(r->R));
    r->P = *params;
    if (hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess) { delete r; return fail(UVCGPU_EDEVICE, "hipStreamCreate failed (no GPU?)"); }
    // UVCGPU_ONE_STREAM=1 (diagnosis): no side streams, every kernel of a handle runs alone, so that UVCGPU_TIMING shows undisturbed durations
    if (getenv("UVCGPU_ONE_STREAM")) { /* side stays null: uvc_launch_accumulate and score run everything on the main stream */ }
    else if (hipStreamCreateWithFlags(&r->side, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&r->side3, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&r->e_join3, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&r->e_stat, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&r->e_alleles, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&r->e_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&r->e_join, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&r->e_fork2, hipEventDisableTiming) != hipSuccess) {
        uvcgpu_region_destroy(r); return fail(UVCGPU_EDEVICE, "hipStreamCreate / hipEventCreate failed");
    }
    if (hipMalloc((void **)&r->R.err, 4) != hipSuccess) { uvcgpu_region_destroy(r); return fail(UVCGPU_ENOMEM, "hipMalloc failed"); }
    const int rc = configure_region(r, tid, beg, end, refseq);
    if (rc) { uvcgpu_region_destroy(r); return rc; }
    *out = r;
    return 0;
}

// The same handle for another region (the next tile): reads, results and the plane contents of the previous region are dropped, streams,
// events and -- when the new region is not longer than the longest one the handle has seen -- the device buffers are kept.
static int uvcgpu_region_reset_impl(uvcgpu_region_t *r, int32_t tid, int32_t beg, int32_t end, const char *refseq) {
    if (!r || !refseq || end <= beg) return fail(UVCGPU_EINVAL, "bad argument");
    HIP_OK(hipStreamSynchronize(r->stream));
    if (r->side) HIP_OK(hipStreamSynchronize(r->side));
    if (r->side3) HIP_OK(hipStreamSynchronize(r->side3));
    free_reads(r);
    return configure_region(r, tid, beg, end, refseq);
}

// The read-dependent preparation runs on the device (uvc_prep.hip) over the caller's columns; `d` holds DEVICE pointers.
static void *prep_alloc(void *ctx, size_t bytes, int zero) {
    uvcgpu_region *r = (uvcgpu_region *)ctx;
    char *p = nullptr;
    if (dev_alloc(r, bytes, &p, zero != 0)) return nullptr;
    return p;
}
static int set_reads_on_device(uvcgpu_region_t *r, const UvcReadSoA *d, bool timing, std::chrono::steady_clock::time_point t_prev) {
    auto lap = [&](const char *what) { if (!timing) return; hipStreamSynchronize(r->stream); const auto t = std::chrono::steady_clock::now();
                                       fprintf(stderr, "[uvcgpu set_reads] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count()); t_prev = t; };
    const int64_t n = d->n_reads;
    if (r->npos >= ((int64_t)1 << 29)) return fail(UVCGPU_EUNSUPPORTED, "region longer than 2^29");
    if (d->n_bases >= ((int64_t)1 << 31)) return fail(UVCGPU_EUNSUPPORTED, "more than 2^31 read bases in one region: split the region");
    if (!d->bases && !d->bases4 && d->n_bases > 0) return fail(UVCGPU_EINVAL, "neither bases nor bases4");
    // compact input forms: offsets as prefix sums of l_qseq / n_cigar, BAM's 4-bit base codes unpacked on the device
    UvcReadSoA dd = *d;
    int32_t *d_bad = nullptr; uint16_t *bq_done = nullptr;
    if (!d->seq_off || !d->cigar_off || !d->bases) {
        int rc0;
        int64_t *so = nullptr, *co = nullptr, *b4o = nullptr; uint8_t *ub = nullptr; char *tmp = nullptr;
        const size_t tb = uvc_prep_compact_tmp_bytes(n);
        if ((rc0 = dev_alloc(r, tb, &tmp)) || (rc0 = dev_alloc(r, 1, &d_bad, true))) return rc0;
        if (!d->seq_off && (rc0 = dev_alloc(r, (size_t)n, &so))) return rc0;
        if (!d->cigar_off && (rc0 = dev_alloc(r, (size_t)n, &co))) return rc0;
        if (!d->bases && ((rc0 = dev_alloc(r, (size_t)n, &b4o)) || (rc0 = dev_alloc(r, (size_t)std::max<int64_t>(d->n_bases, 1), &ub)) || (rc0 = dev_alloc(r, (size_t)std::max<int64_t>(d->n_bases, 1), &bq_done)))) return rc0;
        const int e = uvc_prep_compact(d->l_qseq, d->n_cigar, n, d->n_bases, d->bases ? nullptr : d->bases4, d->n_bases4_bytes, d->quals, so, d->seq_off, co, b4o, ub, bq_done, d_bad, tmp, tb, r->stream);
        if (e) return fail(UVCGPU_EDEVICE, std::string("offset scans: ") + hipGetErrorString((hipError_t)e));
        if (so) dd.seq_off = so;
        if (co) dd.cigar_off = co;
        if (ub) dd.bases = ub;
        d = &dd;
    }
    UvcPrepIn in; memset(&in, 0, sizeof(in));
    in.n_reads = n; in.n_bases = d->n_bases; in.n_cigar_ops = d->n_cigar_ops; in.n_fams = d->n_fams;
    in.pos = d->pos; in.mpos = d->mpos; in.isize = d->isize; in.nm = d->nm; in.l_qseq = d->l_qseq; in.n_cigar = d->n_cigar; in.frag_id = d->frag_id; in.fam_id = d->fam_id;
    in.flag = d->flag; in.mapq = d->mapq; in.fam_strand = d->fam_strand; in.fam_dflag = d->fam_dflag; in.seq_off = d->seq_off; in.cigar_off = d->cigar_off; in.cigars = d->cigars;
    UvcPrepOut o; char msg[256];
    int rc = uvc_prep_reads(&in, &r->P, r->beg, r->end, r->npos, prep_alloc, r, r->stream, &o, msg, (int)sizeof(msg));
    if (rc) return fail(rc, msg);
    lap("nesting + CIGAR facts (device)");
    RegionDev &R = r->R;
    RawReads &W = r->W;
    W.pos = d->pos; W.endpos = o.endpos; W.mpos = d->mpos; W.isize = d->isize; W.flag = d->flag; W.mapq = d->mapq; W.nm = d->nm; W.l_qseq = d->l_qseq; W.n_cigar = d->n_cigar;
    W.frag = o.frag_of; W.fs = o.fs_of; W.dflag = o.dflag_of; W.kind = o.kind; W.seq_off = d->seq_off; W.cigar_off = d->cigar_off; W.table_off = o.table_off; W.item_off = o.item_off; W.gap_off = o.gap_off;
    R.bases = d->bases; R.quals = d->quals; R.cigars = d->cigars;
    R.frag_off[0] = 0; R.frag_off[1] = o.n_frag_strand0; R.frag_off[2] = o.n_frags;
    const int64_t n_simple = o.n_simple; const size_t nf = (size_t)o.n_frags;
    if (bq_done) { R.bq = bq_done; R.bq_bytes = (uint32_t)(d->n_bases * 2); }   // packed while the 4-bit bases were unpacked
    else { uint16_t *b = nullptr; if ((rc = dev_alloc(r, (size_t)std::max<int64_t>(d->n_bases, 1), &b))) return rc; R.bq = b; R.bq_bytes = (uint32_t)(d->n_bases * 2);
      if (!d_bad && (rc = dev_alloc(r, 1, &d_bad, true))) return rc;
      uvc_launch_pack_bq(R.bases, R.quals, b, d->n_bases, d_bad, r->stream); }
    { AlnRec *a; if ((rc = dev_alloc(r, (size_t)n, &a))) return rc; R.alns = a; R.n_alns = (int32_t)n; }
    R.n_fast = (int32_t)n_simple;
    { FastRec *f; if ((rc = dev_alloc(r, (size_t)n_simple, &f))) return rc; R.frec = f; }
    R.complex_ids = o.complex_ids; R.n_complex = o.n_complex;
    R.frags = o.frags; R.n_frags = o.n_frags;
    int pos_bits = 1; while (((int64_t)1 << pos_bits) < r->npos + 1) pos_bits++;   // every sorted position offset is < npos: the radix passes stop there
    {   // stable device sorts: simple alignments by begin (the others go behind them), fragments by (strand, begin) -- k_frag walks two
        // beg-sorted sub-lists, one per strand, so that the strand-specific accumulators are fixed registers
        const size_t nmax = std::max<size_t>(std::max<size_t>((size_t)n, nf), 1);
        int32_t *d_rank, *d_fsorted, *d_frank; uint32_t *work; uint8_t *tmp;
        const size_t tmp_bytes = uvc_sort32_tmp_bytes(nmax);
        if ((rc = dev_alloc(r, (size_t)n, &d_rank)) || (rc = dev_alloc(r, nf, &d_fsorted)) || (rc = dev_alloc(r, nf, &d_frank)) || (rc = dev_alloc(r, 4 * nmax, &work)) || (rc = dev_alloc(r, tmp_bytes + 16, &tmp))) return rc;
        if (uvc_sort_by_pos_cls(W.pos, o.is_complex, r->beg, pos_bits, 1, n, work, tmp, tmp_bytes, r->stream) != 0) return fail(UVCGPU_EDEVICE, "device sort of the alignments failed");
        uvc_launch_rank_from_sorted(work + 3 * n, n, n_simple, nullptr, d_rank, r->stream);
        W.fast_rank = d_rank;
        if (uvc_sort_by_pos_cls(o.frag_beg, o.frag_strand, r->beg, pos_bits, 1, (int64_t)nf, work, tmp, tmp_bytes, r->stream) != 0) return fail(UVCGPU_EDEVICE, "device sort of the fragments failed");
        uvc_launch_rank_from_sorted(work + 3 * nf, (int64_t)nf, (int64_t)nf, d_fsorted, d_frank, r->stream);
        R.frag_sorted = d_fsorted; R.frag_rank = d_frank;
    }
    { FragFast *f; if ((rc = dev_alloc(r, nf, &f, true))) return rc; R.ffast = f; }
    { FragUnit *f; if ((rc = dev_alloc(r, nf, &f, true))) return rc; R.ffast_u = f; }
    R.sweep_frags = o.sweep_frags; R.n_sweep = o.n_sweep;
    { int32_t *q; if ((rc = dev_alloc(r, nf * (size_t)(UVC_MAXEV + 2) + 1, &q, true))) return rc;
      R.frag_nmut = q; R.frag_mut = q + nf; R.overflow_frags = q + nf * (size_t)(UVC_MAXEV + 1); R.n_overflow = q + nf * (size_t)(UVC_MAXEV + 2); }
    R.fss = o.fss; R.n_fs = o.n_fs;
    R.generic_fs = o.generic_fs; R.n_generic_fs = o.n_generic; R.n_generic_work = o.work;
    R.generic_sorted = o.generic_sorted; R.max_unit_span = o.max_unit_span;
    R.fam_digest = nullptr;
    {   // UVCGPU_FAM_PATH=generic | window | (unset: by the data): which form of the family kernels runs -- the three give identical planes (tests/test_gpu_fullsize.py)
        const char *fp = getenv("UVCGPU_FAM_PATH");
        R.fam_path = (fp && !strcmp(fp, "generic")) ? 1 : ((fp && !strcmp(fp, "window")) ? 2 : 0);
        R.frag32 = (getenv("UVCGPU_FRAG32") != nullptr);
    }
    if (R.fam_path == 0 && o.work > 8 * r->npos && (size_t)o.work * 32 <= ((size_t)48 << 30) && o.max_unit_frags < 16384) {   // (the digest packs vote counts in 14 bits)
        // deep data (the window family kernels): 32 B per (unit, position) so that P5 and the duplex pass do not walk the fragments again
        uint32_t *q = nullptr; if ((rc = dev_alloc(r, (size_t)o.work * 8, &q))) return rc; R.fam_digest = q;
    }
    if (timing) fprintf(stderr, "[uvcgpu set_reads] %d fragments, %d units, %d generic units, %lld (unit, position) cells, longest unit %d, %lld positions\n", o.n_frags, o.n_fs, o.n_generic, (long long)o.work, o.max_unit_span, (long long)r->npos);
    { Contrib *t; if ((rc = dev_alloc(r, (size_t)std::max<int64_t>(o.table_rows, 1), &t))) return rc; R.table = t; }
    { int32_t *t; if ((rc = dev_alloc(r, (size_t)(o.gap_slots + 2 * (int64_t)o.n_complex + 4), &t))) return rc; R.ir_list = t; }
    { Item *t; if ((rc = dev_alloc(r, (size_t)std::max<int64_t>(o.item_slots, 1), &t))) return rc; R.items = t;
      int32_t *c; if ((rc = dev_alloc(r, (size_t)o.n_complex + 1, &c, true))) return rc; R.item_cnt = c; }
    {   // InDel allele pipeline (k_gap_*): events, two sort stages, rows
        const int64_t gap_slots = o.gap_slots, ins_total = o.ins_total;
        if (gap_slots >= ((int64_t)1 << 27)) return fail(UVCGPU_EUNSUPPORTED, "more than 2^27 InDel ops in one region");
        if (gap_slots > 0 && r->npos >= ((int64_t)1 << 26)) return fail(UVCGPU_EUNSUPPORTED, "region longer than 2^26 positions: split it (the InDel allele keys hold 26 position bits)");
        GapWork &G = R.gap; memset(&G, 0, sizeof(G));
        G.n_ev = (int32_t)gap_slots; G.inc_cap = (int32_t)(7 * gap_slots + 8); G.seq_cap = ins_total + 8;
        const size_t ne = (size_t)std::max<int64_t>(gap_slots, 1), ni = (size_t)G.inc_cap;
        unsigned long long *k8; int32_t *c4;
        if ((rc = dev_alloc(r, ne, &G.ev)) || (rc = dev_alloc(r, 4 * ne + 4 * ni, &k8)) || (rc = dev_alloc(r, (size_t)4, &c4, true)) || (rc = dev_alloc(r, ne, &G.rows)) || (rc = dev_alloc(r, (size_t)G.seq_cap, &G.seq)) || (rc = dev_alloc(r, 2 * ne, &G.maj))) return rc;
        G.ckey = k8; G.ckey_s = k8 + ne; G.cval = k8 + 2 * ne; G.cval_s = k8 + 3 * ne;
        G.ikey = k8 + 4 * ne; G.ikey_s = G.ikey + ni; G.ival = G.ikey + 2 * ni; G.ival_s = G.ikey + 3 * ni;
        G.n_inc = c4; G.n_rows = c4 + 1; G.seq_len = (unsigned long long *)(c4 + 2);
        G.sort_tmp_bytes = uvc_gap_sort_tmp_bytes(std::max(ne, ni));
        uint8_t *tmp; if ((rc = dev_alloc(r, G.sort_tmp_bytes + 16, &tmp))) return rc; G.sort_tmp = tmp;
    }
    { int32_t *c; if ((rc = dev_alloc(r, (size_t)4, &c, true))) return rc; R.mis_cnt = c; R.mis_total = (unsigned long long *)(c + 2); R.mis = nullptr; R.mis_cap = 0; }
    r->d_dup_units = o.dup_units; r->d_dup_off = o.dup_off; r->n_dup = o.n_dup; r->n_dup_work = o.dup_work;
    R.max_aln_span = o.max_aln_span; R.max_frag_span = o.max_frag_span;
    R.any_amplicon = o.any_amplicon;
    R.max_frag_depth = o.max_frag_depth;   // k_frag packs two 16-bit bucket counters per LDS word when it is below 65 536
    // table rows are written by k_p2_slow<false>; mark all slots empty (0xFF)
    HIP_OK(hipMemsetAsync(R.table, 0, (size_t)std::max<int64_t>(o.table_rows, 1) * sizeof(Contrib), r->stream));   // no base, no LINK symbol
    lap("allocations + orders");
    uvc_launch_prelude(&R, &W, &r->P, r->stream);
    lap("prelude kernel");
    {   // the P2 work list: stable order by (class, begin) on the device -- begin - region begin < 2^29 and the class takes the two bits above
        const size_t np2 = (size_t)o.n_p2;
        for (int c = 0; c <= 4; c++) R.p2_off[c] = o.p2_off[c];
        uint32_t *work; uint8_t *tmp;
        const size_t tmp_bytes = uvc_sort32_tmp_bytes(std::max<size_t>(np2, 1));
        if ((rc = dev_alloc(r, 4 * np2, &work)) || (rc = dev_alloc(r, tmp_bytes + 16, &tmp))) return rc;
        for (int k = 0; k < 4; k++) if ((rc = dev_alloc(r, np2, &r->d_p2[k]))) return rc;
        if (uvc_sort_by_pos_cls(o.p2_beg, o.p2_cls, r->beg, pos_bits, 2, (int64_t)np2, work, tmp, tmp_bytes, r->stream) != 0) return fail(UVCGPU_EDEVICE, "device sort of the P2 work list failed");
        uvc_launch_gather4(work + 3 * np2, (int64_t)np2, o.p2_aln, o.p2_beg, o.p2_end, o.p2_qb, r->d_p2[0], r->d_p2[1], r->d_p2[2], r->d_p2[3], r->stream);
        { FastRec *f; if ((rc = dev_alloc(r, np2, &f))) return rc; R.frec2 = f; R.n_fast2 = (int32_t)np2; R.max_p2_span = o.max_p2_span; }
        uvc_launch_build_p2list(&R, W.fast_rank, r->d_p2[0], r->d_p2[1], r->d_p2[2], r->d_p2[3], r->stream);
    }
    HIP_OK(hipGetLastError());
    {   // the queue of mismatching bases (k_p2_fast -> k_p2_mism) is sized from the count the prelude made
        unsigned long long total = 0; int32_t bad4 = 0;
        HIP_OK(hipMemcpyAsync(&total, R.mis_total, sizeof(total), hipMemcpyDeviceToHost, r->stream));
        if (d_bad) HIP_OK(hipMemcpyAsync(&bad4, d_bad, sizeof(bad4), hipMemcpyDeviceToHost, r->stream));
        HIP_OK(hipStreamSynchronize(r->stream));
        if (bad4 == 2) return fail(UVCGPU_EINVAL, "a base code outside 0..4 in UvcReadSoA::bases");
        if (bad4) return fail(UVCGPU_EINVAL, "bases4 / l_qseq / n_bases do not fit together");
        if (total > ((unsigned long long)1 << 30)) return fail(UVCGPU_EUNSUPPORTED, "more than 2^30 mismatching bases in one region");
        MisItem *q = nullptr;
        if ((rc = dev_alloc(r, (size_t)(total + 64), &q))) return rc;
        R.mis = q; R.mis_cap = (int32_t)(total + 64);
    }
    lap("P2 list sort + build");
    r->n_bases = d->n_bases;
    r->has_reads = true;
    return 0;
}

// Host columns: they are copied to the device as they are (no host pass over the reads), then prepared there.
static int uvcgpu_region_set_reads_impl(uvcgpu_region_t *r, const UvcReadSoA *in) {
    if (!r || !in) return fail(UVCGPU_EINVAL, "bad reads");
    if (in->struct_size != (int32_t)sizeof(UvcReadSoA)) return fail(UVCGPU_EINVAL, "UvcReadSoA::struct_size mismatch (set it to sizeof(UvcReadSoA))");
    if (in->n_reads < 0 || in->n_fams < 0) return fail(UVCGPU_EINVAL, "bad reads");
    const bool timing = (getenv("UVCGPU_TIMING") != nullptr);   // stderr breakdown of the ingest, for tuning
    auto t_prev = std::chrono::steady_clock::now();
    free_reads(r);
    struct SyncOnExit { hipStream_t s, s2; ~SyncOnExit() { hipStreamSynchronize(s); if (s2) hipStreamSynchronize(s2); } } sync_on_exit = { r->stream, r->side };   // no copy may outlive the caller's arrays, on any return path
    const int64_t n = in->n_reads;
    if (n == 0) return 0;
    if (n > INT32_MAX / 2) return fail(UVCGPU_EUNSUPPORTED, "more than 2^30 reads in one region");
    if (in->n_bases < 0 || in->n_cigar_ops < 0) return fail(UVCGPU_EINVAL, "bad reads");
    UvcReadSoA d = *in;
    int rc;
    // The two large columns travel on the handle's main stream, the dozen small ones (8 MB each at 2 M reads; a copy costs ~0.13 ms of set-up
    // whatever its size) on a side stream beside them: their set-up times hide under the large transfers instead of adding up in front.
    hipStream_t small_stream = (r->side ? r->side : r->stream);
    auto up_on = [&](hipStream_t st, const void *src, size_t bytes, void **out) -> int {
        char *q = nullptr;
        int rc2 = dev_alloc(r, bytes, &q);
        if (rc2) return rc2;
        if (bytes && hipMemcpyAsync(q, src, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return fail(UVCGPU_EDEVICE, "hipMemcpyAsync(H2D)");
        *out = q;
        return 0;
    };
#define UP(field, T, count) { void *q; if ((rc = up_on(small_stream, in->field, sizeof(T) * (size_t)(count), &q))) return rc; d.field = (const T *)q; }
#define UPBIG(field, T, count) { void *q; if ((rc = up_on(r->stream, in->field, sizeof(T) * (size_t)(count), &q))) return rc; d.field = (const T *)q; }
    if (in->bases) UPBIG(bases, uint8_t, in->n_bases)
    else if (in->bases4) { if (in->n_bases4_bytes < 0) return fail(UVCGPU_EINVAL, "bad reads"); UPBIG(bases4, uint8_t, in->n_bases4_bytes) }
    UPBIG(quals, uint8_t, in->n_bases)
    UP(pos, int32_t, n) UP(mpos, int32_t, n) UP(isize, int32_t, n) UP(flag, uint16_t, n) UP(mapq, uint8_t, n) UP(nm, int32_t, n) UP(l_qseq, int32_t, n)
    UP(n_cigar, int32_t, n) UP(frag_id, int32_t, n) UP(fam_id, int32_t, n) UP(fam_strand, uint8_t, n)
    if (in->seq_off) UP(seq_off, int64_t, n)
    if (in->cigar_off) UP(cigar_off, int64_t, n)
    UP(cigars, uint32_t, in->n_cigar_ops) UP(fam_dflag, uint8_t, in->n_fams)
    if (small_stream != r->stream) {   // the preparation kernels read every column: the main stream waits for the side stream's copies
        if (hipEventRecord(r->e_fork2, small_stream) != hipSuccess || hipStreamWaitEvent(r->stream, r->e_fork2, 0) != hipSuccess) return fail(UVCGPU_EDEVICE, "hipEventRecord / hipStreamWaitEvent");
    }
#undef UPBIG
#undef UP
    if (timing) { hipStreamSynchronize(r->stream); const auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[uvcgpu set_reads] %-28s %8.2f ms\n", "H2D of the columns", std::chrono::duration<double, std::milli>(t - t_prev).count()); t_prev = t; }
    return set_reads_on_device(r, &d, timing, t_prev);
}

// The same with the columns already in HBM (every pointer of `in` is a device pointer on the handle's device): nothing is copied.  The
// arrays must stay valid and unchanged until the handle gets other reads, is reset or destroyed -- the kernels of every accumulate
// read bases / quals / cigars in place -- and uvcgpu_region_correct_bq edits `quals` in place, as the reference edits its bam1_t.
static int uvcgpu_region_set_reads_device_impl(uvcgpu_region_t *r, const UvcReadSoA *in) {
    if (!r || !in) return fail(UVCGPU_EINVAL, "bad reads");
    if (in->struct_size != (int32_t)sizeof(UvcReadSoA)) return fail(UVCGPU_EINVAL, "UvcReadSoA::struct_size mismatch (set it to sizeof(UvcReadSoA))");
    if (in->n_reads < 0 || in->n_fams < 0 || in->n_bases < 0 || in->n_cigar_ops < 0) return fail(UVCGPU_EINVAL, "bad reads");
    const bool timing = (getenv("UVCGPU_TIMING") != nullptr);
    free_reads(r);
    if (in->n_reads == 0) return 0;
    if (in->n_reads > INT32_MAX / 2) return fail(UVCGPU_EUNSUPPORTED, "more than 2^30 reads in one region");
    return set_reads_on_device(r, in, timing, std::chrono::steady_clock::now());
}

// apply_bq_err_correction3 (grouping.cpp:459-543) on the resident reads, then everything derived from the base qualities again:
// the packed base|qual array and the per-read records (the low-quality-InDel test of k_aln_prelude reads them)
static int uvcgpu_region_correct_bq_impl(uvcgpu_region_t *r) {
    if (!r) return fail(UVCGPU_EINVAL, "null region");
    if (!r->has_reads) return fail(UVCGPU_ENOREADS, "no reads");
    uvc_launch_correct_bq(&r->R, r->P.assay_sequencing_BQ_max, r->P.assay_sequencing_BQ_inc, r->stream);
    uvc_launch_pack_bq(r->R.bases, r->R.quals, (uint16_t *)r->R.bq, r->n_bases, nullptr, r->stream);
    HIP_OK(hipMemsetAsync(r->R.mis_total, 0, sizeof(unsigned long long), r->stream));
    uvc_launch_prelude(&r->R, &r->W, &r->P, r->stream);
    uvc_launch_build_p2list(&r->R, r->W.fast_rank, r->d_p2[0], r->d_p2[1], r->d_p2[2], r->d_p2[3], r->stream);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(r->stream));
    r->accumulated = false;
    return 0;
}

// the base qualities as they are on the device (after uvcgpu_region_correct_bq, if it was called)
int uvcgpu_region_read_quals(uvcgpu_region_t *r, uint8_t *dst, int64_t n) {
    if (!r || !dst) return fail(UVCGPU_EINVAL, "null argument");
    if (!r->has_reads) return fail(UVCGPU_ENOREADS, "no reads");
    if (n != r->n_bases) return fail(UVCGPU_EINVAL, "n must equal UvcReadSoA::n_bases");
    HIP_OK(hipStreamSynchronize(r->stream));
    if (n > 0) HIP_OK(hipMemcpy(dst, r->R.quals, (size_t)n, hipMemcpyDeviceToHost));
    return 0;
}

// Zero fill of the plane slab (everything in front of the bucket planes, and the bucket planes too unless P3b / P5b left them clean) on
// stream `s`.  When the slab's contents were written under the present layout, only what every accumulate writes and what the last one marked
// (RegionDev::dirty) is filled; otherwise (first use, the handle was rebound to a region of another length) the whole slab.
struct ZeroPlaneHost { unsigned long long off; int32_t elem; int16_t fam, sym; };
static int zero_state(uvcgpu_region *r, hipStream_t s) {
    const size_t n_dirty = (size_t)3 * NSYM * (size_t)r->R.ndblk;
    if (r->dirty_npos == r->npos && !getenv("UVCGPU_FILL_ALL")) {
        if (r->zero_planes_npos != r->npos) {   // the plane table of this layout (once per region length)
            std::vector<ZeroPlaneHost> v;
            const size_t n = (size_t)r->npos;
            auto sym_planes = [&](int g, int nplanes, int elem, int fam) {
                for (int pl = 0; pl < nplanes; pl++) for (int sy = 0; sy < NSYM; sy++) {
                    const bool always = (fam == 0 && (sy < UVC_BASE_NN || sy == UVC_LINK_M));
                    v.push_back({ (unsigned long long)(r->off[g] + ((size_t)pl * NSYM + sy) * n * elem), elem, (int16_t)(always ? -1 : fam), (int16_t)sy });
                }
            };
            for (int pl = 0; pl < UVC_NPREP32; pl++) v.push_back({ (unsigned long long)(r->off[UVC_F_PREP32] + (size_t)pl * n * 4), 4, -1, 0 });
            for (int pl = 0; pl < UVC_NPREP64; pl++) v.push_back({ (unsigned long long)(r->off[UVC_F_PREP64] + (size_t)pl * n * 8), 8, -1, 0 });
            // (k_thres stores every threshold of every position, but a handle that is rebound to a region of another length relies on an all-zero slab)
            for (int pl = 0; pl < UVC_NTHRES; pl++) v.push_back({ (unsigned long long)(r->off[UVC_F_THRES] + (size_t)pl * n * 4), 4, -1, 0 });
            sym_planes(UVC_F_SEG32, UVC_NSEG32, 4, 0); sym_planes(UVC_F_SEG64, UVC_NSEG64, 8, 0); sym_planes(UVC_F_VQ, UVC_NVQ, 4, 0); sym_planes(UVC_F_BQSUM, 1, 4, 0);
            sym_planes(UVC_F_FRAG, 2 * UVC_NFRAG, 4, 0); sym_planes(UVC_F_FAM, 2 * UVC_NFAM, 4, 0);
            sym_planes(UVC_F_FAMINFO32, UVC_NFAMINFO32, 4, 1); sym_planes(UVC_F_FAMINFO64, UVC_NFAMINFO64, 8, 1); sym_planes(UVC_F_DUPLEX, UVC_NDUPLEX, 4, 2);
            v.push_back({ (unsigned long long)r->p5flag_off, 1, -1, 0 }); v.push_back({ (unsigned long long)(r->p5flag_off + n), 1, -1, 0 });
            v.push_back({ (unsigned long long)r->occ_off, 4, -1, 0 });
            if (r->d_zero_planes) { quiesce(r); hipFree(r->d_zero_planes); r->d_zero_planes = nullptr; }
            HIP_OK(hipMalloc(&r->d_zero_planes, v.size() * sizeof(ZeroPlaneHost)));
            HIP_OK(hipMemcpy(r->d_zero_planes, v.data(), v.size() * sizeof(ZeroPlaneHost), hipMemcpyHostToDevice));
            r->n_zero_planes = (int)v.size(); r->zero_planes_npos = r->npos;
        }
        uvc_launch_zero_state(r->d_state, r->d_zero_planes, r->n_zero_planes, r->d_dirty, r->R.ndblk, r->npos, s);
        if (hipGetLastError() != hipSuccess) return fail(UVCGPU_EDEVICE, "zero fill of the planes failed");
        if (!r->buckets_clean) HIP_OK(hipMemsetAsync(r->d_state + r->bucket_off, 0, r->state_bytes - r->bucket_off, s));
    } else {
        HIP_OK(hipMemsetAsync(r->d_state, 0, r->buckets_clean ? r->bucket_off : r->state_bytes, s));
    }
    HIP_OK(hipMemsetAsync(r->d_dirty, 0, n_dirty, s));
    return 0;
}

static int uvcgpu_region_accumulate_impl(uvcgpu_region_t *r) {
    if (!r) return fail(UVCGPU_EINVAL, "null region");
    if (!r->has_reads) return fail(UVCGPU_ENOREADS, "no reads");   // process_batch returns -1, main.cpp:520-523
    if (r->state_zeroed) HIP_OK(hipStreamWaitEvent(r->stream, r->e_join, 0));   // zeroed behind the last score (release_state)
    else { const int rcz = zero_state(r, r->stream); if (rcz) return rcz; }
    r->state_zeroed = false; r->state_released = false;
    r->buckets_clean = false;
    HIP_OK(hipMemcpyAsync(r->d_rtr, r->d_rtr0, (size_t)4 * UVC_NRTR * r->npos, hipMemcpyDeviceToDevice, r->stream));   // P1b edits indelphred in place
    HIP_OK(hipMemsetAsync(r->R.frag_nmut, 0, sizeof(int32_t) * (size_t)r->R.n_frags, r->stream));
    HIP_OK(hipMemsetAsync(r->R.n_overflow, 0, sizeof(int32_t), r->stream));
    HIP_OK(hipMemsetAsync(r->R.mis_cnt, 0, sizeof(int32_t), r->stream));
    // the table is rebuilt by k_p2_slow<false>: MAX-merge needs empty slots
    if (r->R.n_complex) {
        // rows were set to 0xFF in set_reads and k_p2_slow<false> is idempotent under MAX, so no reset is needed
    }
    const int half = (int)std::round((10.0 / std::log(10.0)) * std::log(r->P.indel_del_to_ins_err_ratio)) / 2;   // main.hpp:1244
    uvc_launch_accumulate(&r->R, &r->P, half, r->d_dup_units, r->n_dup, r->d_dup_off, r->n_dup_work, r->stream, &r->prof, r->side, r->e_fork, r->e_join, r->e_fork2, r->side3, r->e_join3, r->e_stat, r->e_alleles);
    HIP_OK(hipGetLastError());
    r->buckets_clean = (r->P.inferred_is_vcf_generated != 0);   // k_frag (P3b) and k_p5b cleared every bucket they consumed
    r->dirty_npos = r->npos;   // the slab and the marks now describe this layout
    r->accumulated = true; r->gap_ready = false; r->hap_ready = false;
    return 0;
}

int uvcgpu_region_check_presence(uvcgpu_region_t *r, int64_t *n_violations) {
    if (!r || !n_violations) return fail(UVCGPU_EINVAL, "bad argument");
    if (!r->accumulated || r->state_released) return fail(UVCGPU_ESTATE, "check_presence needs the planes of an accumulate");
    unsigned long long *d = nullptr, h = 0;
    HIP_OK(hipMalloc((void **)&d, 8));
    int rc = 0;
    if (hipMemsetAsync(d, 0, 8, r->stream) != hipSuccess) rc = fail(UVCGPU_EDEVICE, "hipMemsetAsync");
    if (!rc) { uvc_launch_check_presence(&r->R, d, r->stream); uvc_launch_check_dirty(&r->R, d, r->stream); if (hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, r->stream) != hipSuccess || hipStreamSynchronize(r->stream) != hipSuccess) rc = fail(UVCGPU_EDEVICE, "check_presence"); }
    hipFree(d);
    *n_violations = (int64_t)h;
    return rc;
}

// Per-kernel timing of the LAST accumulate, measured with HIP events on the handle's own stream.
int uvcgpu_region_set_profiling(uvcgpu_region_t *r, int on) { if (!r) return fail(UVCGPU_EINVAL, "null region"); r->prof.on = on ? 1 : 0; return 0; }
int uvcgpu_region_kernel_times(uvcgpu_region_t *r, char *names, int names_bytes, float *ms, int capacity) {
    if (!r || !names || !ms) return fail(UVCGPU_EINVAL, "bad argument");
    HIP_OK(hipStreamSynchronize(r->stream));
    std::string all;
    int n = 0;
    for (int i = 0; i < r->prof.n && n < capacity; i++, n++) {
        float t = 0;
        HIP_OK(hipEventElapsedTime(&t, r->prof.ev[i][0], r->prof.ev[i][1]));
        ms[n] = t; all += r->prof.name[i]; all += ";";
    }
    if ((int)all.size() + 1 > names_bytes) return fail(UVCGPU_EINVAL, "names buffer too small");
    memcpy(names, all.c_str(), all.size() + 1);
    return n;
}

// how many records the last uvcgpu_region_score scored, and how many it returned (fewer with UvcScoreRequest::kept_only)
int uvcgpu_region_last_score_counts(const uvcgpu_region_t *r, int64_t *scored, int64_t *returned) {
    if (!r) return fail(UVCGPU_EINVAL, "null region");
    if (scored) *scored = r->last_scored;
    if (returned) *returned = r->last_returned;
    return 0;
}

int uvcgpu_region_sync(uvcgpu_region_t *r) {
    if (!r) return fail(UVCGPU_EINVAL, "null region");
    HIP_OK(hipStreamSynchronize(r->stream));
    int32_t e = 0;
    HIP_OK(hipMemcpy(&e, r->R.err, 4, hipMemcpyDeviceToHost));
    if (e) return fail(e, "a kernel flagged an unsupported read shape (a CIGAR op the reference itself throws on, process_cigar main_conversion.hpp:902-916)");
    return 0;
}

int64_t uvcgpu_region_field_bytes(const uvcgpu_region_t *r, int32_t g) {
    if (!r || g < 0 || g >= UVC_NUM_FIELD_GROUPS) return -1;
    if (g != UVC_F_RTR && g != UVC_F_BAQ && !r->accumulated) return -1;
    return (int64_t)group_bytes(r, g);
}

static int uvcgpu_region_fetch_impl(uvcgpu_region_t *r, int32_t g, void *dst, int64_t dst_bytes) {
    if (!r || !dst || g < 0 || g >= UVC_NUM_FIELD_GROUPS) return fail(UVCGPU_EINVAL, "bad argument");
    if (g != UVC_F_RTR && g != UVC_F_BAQ && !r->accumulated) return fail(UVCGPU_ESTATE, "fetch before accumulate");
    if (g != UVC_F_RTR && g != UVC_F_BAQ && r->state_released) return fail(UVCGPU_ESTATE, "the planes were released by the last score (UvcScoreRequest::release_state)");
    if ((int64_t)group_bytes(r, g) != dst_bytes) return fail(UVCGPU_EINVAL, "bad destination size");
    int rc = uvcgpu_region_sync(r);
    if (rc) return rc;
    const void *src = (g == UVC_F_RTR) ? (const void *)(r->accumulated ? r->d_rtr : r->d_rtr0) /* P1b's edit of indelphred exists after accumulate only */ : (g == UVC_F_BAQ) ? (const void *)r->d_baq : (const void *)(r->d_state + r->off[g]);
    HIP_OK(hipMemcpy(dst, src, (size_t)dst_bytes, hipMemcpyDeviceToHost));
    return 0;
}

// ---- columns of chosen positions (the read side of the record writer, uvc_vcf.cpp) ----
namespace {
int group_elem(int g) { return (g == UVC_F_PREP64 || g == UVC_F_SEG64 || g == UVC_F_FAMINFO64 || g == UVC_F_BAQ) ? 8 : 4; }
int group_planes(int g) {
    switch (g) {
        case UVC_F_PREP32: return UVC_NPREP32; case UVC_F_PREP64: return UVC_NPREP64; case UVC_F_THRES: return UVC_NTHRES; case UVC_F_SEG32: return UVC_NSEG32 * NSYM;
        case UVC_F_SEG64: return UVC_NSEG64 * NSYM; case UVC_F_VQ: return UVC_NVQ * NSYM; case UVC_F_BQSUM: return NSYM; case UVC_F_FRAG: return 2 * UVC_NFRAG * NSYM;
        case UVC_F_FAM: return 2 * UVC_NFAM * NSYM; case UVC_F_FAMINFO32: return UVC_NFAMINFO32 * NSYM; case UVC_F_FAMINFO64: return UVC_NFAMINFO64 * NSYM; case UVC_F_DUPLEX: return UVC_NDUPLEX * NSYM;
    }
    return 0;   // RTR / BAQ are not part of a row
}
}
int32_t uvcgpu_region_column_base(int32_t g) {
    if (g < 0 || g >= UVC_NUM_FIELD_GROUPS || group_planes(g) == 0) return -1;
    int32_t at = 0;
    for (int q = 0; q < g; q++) at += group_planes(q);
    return at;
}
int32_t uvcgpu_region_n_columns(void) { int32_t at = 0; for (int q = 0; q < UVC_NUM_FIELD_GROUPS; q++) at += group_planes(q); return at; }
static int uvcgpu_region_fetch_columns_impl(uvcgpu_region_t *r, const int32_t *refpos, int64_t n, int64_t *dst) {
    if (!r || n < 0 || (n > 0 && (!refpos || !dst))) return fail(UVCGPU_EINVAL, "bad argument");
    if (!r->accumulated) return fail(UVCGPU_ESTATE, "fetch before accumulate");
    if (r->state_released) return fail(UVCGPU_ESTATE, "the planes were released by the last score (UvcScoreRequest::release_state)");
    if (n == 0) return 0;
    int rc = uvcgpu_region_sync(r);
    if (rc) return rc;
    const int32_t ncol = uvcgpu_region_n_columns();
    const char *base[UVC_NUM_FIELD_GROUPS]; int32_t first[UVC_NUM_FIELD_GROUPS + 1], elem[UVC_NUM_FIELD_GROUPS];
    int32_t at = 0;
    for (int g = 0; g < UVC_NUM_FIELD_GROUPS; g++) { first[g] = at; at += group_planes(g); elem[g] = group_elem(g); base[g] = (group_planes(g) ? r->d_state + r->off[g] : r->d_state); }
    first[UVC_NUM_FIELD_GROUPS] = at;
    std::vector<int32_t> xs((size_t)n);
    for (int64_t i = 0; i < n; i++) xs[(size_t)i] = refpos[i] - r->beg;   // out of range -> a row of zeros (checked in the kernel)
    int32_t *d_xs = nullptr; long long *d_out = nullptr;
    if (hipMalloc((void **)&d_xs, sizeof(int32_t) * (size_t)n) != hipSuccess) return fail(UVCGPU_ENOMEM, "hipMalloc(column positions) failed");
    if (hipMalloc((void **)&d_out, sizeof(long long) * (size_t)n * ncol) != hipSuccess) { hipFree(d_xs); return fail(UVCGPU_ENOMEM, "hipMalloc(columns) failed"); }
    hipError_t e = hipMemcpyAsync(d_xs, xs.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, r->stream);
    if (e == hipSuccess) { uvc_launch_gather_columns(base, first, elem, r->npos, d_xs, n, d_out, r->stream); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(dst, d_out, sizeof(long long) * (size_t)n * ncol, hipMemcpyDeviceToHost, r->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
    hipFree(d_xs); hipFree(d_out);
    if (e != hipSuccess) return fail(UVCGPU_EDEVICE, hipGetErrorString(e));
    return 0;
}
// position-level numbers for the MGVCF block and ADDITIONAL_INDEL_CANDIDATE lines of the record writer: 10 ints per position of
// [refpos_beg, refpos_end), see k_block_stats
int uvcgpu_region_block_stats_(uvcgpu_region_t *r, int32_t refpos_beg, int32_t refpos_end, int32_t *dst) {
    if (!r || !dst || refpos_end < refpos_beg) return fail(UVCGPU_EINVAL, "bad argument");
    if (!r->accumulated) return fail(UVCGPU_ESTATE, "fetch before accumulate");
    if (r->state_released) return fail(UVCGPU_ESTATE, "the planes were released by the last score (UvcScoreRequest::release_state)");
    const int64_t n = (int64_t)refpos_end - refpos_beg;
    if (n == 0) return 0;
    int rc = uvcgpu_region_sync(r);
    if (rc) return rc;
    int32_t *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(int32_t) * 10 * (size_t)n) != hipSuccess) return fail(UVCGPU_ENOMEM, "hipMalloc(block stats) failed");
    uvc_launch_block_stats(&r->R, &r->P, (int64_t)refpos_beg - r->beg, n, d, r->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(dst, d, sizeof(int32_t) * 10 * (size_t)n, hipMemcpyDeviceToHost, r->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
    hipFree(d);
    if (e != hipSuccess) return fail(UVCGPU_EDEVICE, hipGetErrorString(e));
    return 0;
}
// Page-locks a caller buffer (the records buffer of uvcgpu_region_score, the read arrays of uvcgpu_region_set_reads) so that copies to
// and from it run at PCIe speed instead of through the runtime's staging buffers.  Optional; the buffer must be unpinned before it is freed.
int uvcgpu_pin_host_buffer(void *p, int64_t bytes) {
    if (!p || bytes <= 0) return fail(UVCGPU_EINVAL, "bad argument");
    hipError_t e = hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(UVCGPU_EDEVICE, std::string("hipHostRegister: ") + hipGetErrorString(e)); }
    return 0;
}
int uvcgpu_host_alloc(void **p, int64_t bytes) {
    if (!p || bytes <= 0) return fail(UVCGPU_EINVAL, "bad argument");
    *p = nullptr;
    hipError_t e = hipHostMalloc(p, (size_t)bytes, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return fail(UVCGPU_ENOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
    return 0;
}
int uvcgpu_host_free(void *p) {
    if (!p) return 0;
    hipError_t e = hipHostFree(p);
    if (e != hipSuccess) { (void)hipGetLastError(); if (getenv("UVCGPU_DEBUG")) fprintf(stderr, "[uvcgpu] hipHostFree: %s\n", hipGetErrorString(e)); return fail(UVCGPU_EDEVICE, std::string("hipHostFree: ") + hipGetErrorString(e)); }
    return 0;
}
int uvcgpu_unpin_host_buffer(void *p) {
    if (!p) return 0;
    hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(UVCGPU_EDEVICE, std::string("hipHostUnregister: ") + hipGetErrorString(e)); }
    return 0;
}

// what uvc_vcf.cpp reads of a handle besides the public calls
const char *uvcgpu_region_refseq(const uvcgpu_region_t *r, int32_t *beg, int32_t *end) { if (beg) *beg = r->beg; if (end) *end = r->end - 1; return r->refstring.c_str(); }
// begpos / tracklen / unitlen of every position as refstring2repeatvec built them (the first three planes of UVC_F_RTR before P1b): the
// record writer's INFO/R3X2.  Fetched from the device on first use after a (re)bind; NULL when the copy fails.
const int32_t *uvcgpu_region_repeat_tracks(const uvcgpu_region_t *cr, int64_t *npos) {
    uvcgpu_region_t *r = const_cast<uvcgpu_region_t *>(cr);
    if (npos) *npos = r->npos;
    if (!r->h_rtr_valid) {
        r->h_rtr.resize((size_t)3 * r->npos);
        // on the handle's own stream: a null-stream copy would also wait for every other handle's work
        if (hipMemcpyAsync(r->h_rtr.data(), r->d_rtr0, sizeof(int32_t) * r->h_rtr.size(), hipMemcpyDeviceToHost, r->stream) != hipSuccess || hipStreamSynchronize(r->stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        r->h_rtr_valid = true;
    }
    return r->h_rtr.data();
}
const UvcParams *uvcgpu_region_params(const uvcgpu_region_t *r) { return &r->P; }
int uvcgpu_fail_(int code, const char *msg) { return fail(code, msg); }

// ---- InDel allele tables: the host half of fill_by_indel_info / indel_get_majority (instcode.hpp, main.hpp:5350-5455) ----
// The device reduces the allele-keyed counters to one GapRow per (position, symbol, allele) (k_gap_rows); what is left is per InDel
// site: splitting by strand, the reference's two sorts, and the quarter-of-the-best filter.
namespace {
struct GapAl { int32_t len; const uint8_t *seq; bool del; };   // allele text: inserted bases (codes 0..4) or a deleted length
inline int text_rank(int b) { return b == 4 ? 3 : (b == 3 ? 4 : b); }   // "ACGTN": A < C < G < N < T
int gap_al_cmp(const GapAl &a, const GapAl &b) {   // std::string order of the reference's keys
    if (a.del) return (a.len > b.len) - (a.len < b.len);   // prefixes of the same reference text
    const int n = std::min(a.len, b.len);
    for (int i = 0; i < n; i++) { const int ra = text_rank(a.seq[i]), rb = text_rank(b.seq[i]); if (ra != rb) return ra < rb ? -1 : 1; }
    return (a.len > b.len) - (a.len < b.len);
}
}
static int gap_tables(uvcgpu_region_t *r) {
    if (r->gap_ready) return 0;
    // The allele pipeline runs on the side stream and ends long before the main stream does (uvc_launch_accumulate): wait for it alone,
    // so that this host step and the launches of the scoring kernels behind it stay hidden under the fragment / family kernels.
    hipStream_t cs = (r->side ? r->side : r->stream);
    if (r->side) HIP_OK(hipEventSynchronize(r->e_fork2)); else HIP_OK(hipStreamSynchronize(r->stream));
    r->gap_rows.clear(); r->gap_seq.clear(); r->gap_alleles.clear(); r->gap_allele_row.clear();
    const GapWork &G = r->R.gap;
    int32_t cnt[4] = { 0, 0, 0, 0 };
    if (G.n_inc) { HIP_OK(hipMemcpyAsync(cnt, G.n_inc, 16, hipMemcpyDeviceToHost, cs)); HIP_OK(hipStreamSynchronize(cs)); }
    const int32_t n_rows = cnt[1];
    unsigned long long seq_len = 0; memcpy(&seq_len, &cnt[2], 8);
    if (n_rows < 0 || n_rows > G.n_ev || seq_len > (unsigned long long)G.seq_cap) return fail(UVCGPU_EDEVICE, "allele table overflow");   // before anything is sized from them
    std::vector<GapRow> dev((size_t)n_rows);
    std::vector<uint8_t> dseq((size_t)seq_len);
    if (n_rows) HIP_OK(hipMemcpyAsync(dev.data(), G.rows, sizeof(GapRow) * (size_t)n_rows, hipMemcpyDeviceToHost, cs));
    if (seq_len) HIP_OK(hipMemcpyAsync(dseq.data(), G.seq, (size_t)seq_len, hipMemcpyDeviceToHost, cs));
    if (n_rows || seq_len) HIP_OK(hipStreamSynchronize(cs));
    auto al_of = [&](const GapRow &g) { GapAl a; a.len = g.len; a.del = (g.seq_off < 0); a.seq = (a.del ? nullptr : dseq.data() + g.seq_off); return a; };
    std::sort(dev.begin(), dev.end(), [&](const GapRow &a, const GapRow &b) {
        if (a.x != b.x) return a.x < b.x;
        if (a.sym != b.sym) return a.sym < b.sym;
        return gap_al_cmp(al_of(a), al_of(b)) < 0; });   // ascending allele text: the iteration order of the reference's maps
    struct Tup { int32_t fq, bq, c2, c2d; const GapRow *g; };
    for (size_t g0 = 0; g0 < dev.size();) {
        size_t g1 = g0;
        while (g1 < dev.size() && dev[g1].x == dev[g0].x && dev[g1].sym == dev[g0].sym) g1++;
        const int32_t refpos = r->beg + dev[g0].x, symbol = dev[g0].sym;
        const size_t first_row = r->gap_rows.size();
        std::vector<const GapRow *> row_allele;   // parallel to the rows pushed for this site
        for (int strand = 0; strand < 2; strand++) {   // fill_by_indel_info2_{1,2}: the strand's tuples, sorted descending (instcode.hpp:44-62)
            std::vector<Tup> t;
            for (size_t g = g0; g < g1; g++) if (dev[g].cnt[strand * 4] > 0) t.push_back(Tup{ dev[g].cnt[strand * 4 + 1], dev[g].cnt[strand * 4], dev[g].cnt[strand * 4 + 2], dev[g].cnt[strand * 4 + 3], &dev[g] });
            std::sort(t.begin(), t.end(), [&](const Tup &a, const Tup &b) {
                if (a.fq != b.fq) return a.fq > b.fq;
                if (a.bq != b.bq) return a.bq > b.bq;
                if (a.c2 != b.c2) return a.c2 > b.c2;
                if (a.c2d != b.c2d) return a.c2d > b.c2d;
                return gap_al_cmp(al_of(*a.g), al_of(*b.g)) > 0; });
            for (const Tup &u : t) {
                UvcGapRow o; memset(&o, 0, sizeof(o));
                o.refpos = refpos; o.symbol = symbol; o.strand = strand; o.len = u.g->len; o.seq_off = -1;
                if (u.g->seq_off >= 0) { o.seq_off = (int64_t)r->gap_seq.size(); r->gap_seq.insert(r->gap_seq.end(), dseq.begin() + u.g->seq_off, dseq.begin() + u.g->seq_off + u.g->len); }
                o.bAD1 = u.bq; o.cAD1 = u.fq; o.c2AD = u.c2; o.c2dAD = u.c2d;
                r->gap_rows.push_back(o); row_allele.push_back(u.g);
            }
        }
        // indel_get_majority (main.hpp:5406-5455): merge the strands per allele, keep those with at least a quarter of the best fragment
        // support, order by descending bAD1^2 * length (ties: ascending text -- what the reference's std::sort over
        // reverse iterators leaves for the short vectors that occur)
        struct Maj { int32_t b, c; const GapRow *g; };
        std::vector<Maj> m;
        int32_t max_b = 0;
        for (size_t g = g0; g < g1; g++) {
            Maj a = { 0, 0, &dev[g] };
            for (int strand = 0; strand < 2; strand++) if (dev[g].cnt[strand * 4] > 0) { a.b += dev[g].cnt[strand * 4]; a.c += dev[g].cnt[strand * 4 + 1]; }
            if (a.b > 0) { m.push_back(a); max_b = std::max(max_b, a.b); }
        }
        std::vector<Maj> kept;
        for (const Maj &a : m) if (a.b >= (max_b + 3) / 4) kept.push_back(a);
        std::stable_sort(kept.begin(), kept.end(), [](const Maj &a, const Maj &b) { return (int64_t)a.b * a.b * (int64_t)a.g->len > (int64_t)b.b * b.b * (int64_t)b.g->len; });
        for (const Maj &a : kept) {
            r->gap_alleles.push_back(UvcIndelAllele{ refpos, symbol, a.b, a.c, a.g->len });
            int32_t row = -1;
            for (size_t q = 0; q < row_allele.size() && row < 0; q++) if (row_allele[q] == a.g) row = (int32_t)(first_row + q);
            r->gap_allele_row.push_back(row);
        }
        g0 = g1;
    }
    const int64_t na = (int64_t)r->gap_alleles.size();
    if (na > r->gap_alleles_cap) {
        if (r->d_gap_alleles) hipFree(r->d_gap_alleles);
        if (r->d_gap_allele_row) hipFree(r->d_gap_allele_row);
        r->d_gap_alleles = nullptr; r->d_gap_allele_row = nullptr; r->gap_alleles_cap = 0;
        HIP_OK(hipMalloc((void **)&r->d_gap_alleles, sizeof(UvcIndelAllele) * (size_t)(na + 64)));
        HIP_OK(hipMalloc((void **)&r->d_gap_allele_row, sizeof(int32_t) * (size_t)(na + 64)));
        r->gap_alleles_cap = na + 64;
    }
    const int64_t nr = (int64_t)r->gap_rows.size(), ns = (int64_t)r->gap_seq.size();
    if (nr > r->gap_rows_cap) { if (r->d_gap_rows) hipFree(r->d_gap_rows); r->d_gap_rows = nullptr; r->gap_rows_cap = 0; HIP_OK(hipMalloc((void **)&r->d_gap_rows, sizeof(UvcGapRow) * (size_t)(nr + 64))); r->gap_rows_cap = nr + 64; }
    if (ns > r->gap_seq_cap) { if (r->d_gap_seq) hipFree(r->d_gap_seq); r->d_gap_seq = nullptr; r->gap_seq_cap = 0; HIP_OK(hipMalloc((void **)&r->d_gap_seq, (size_t)(ns + 64))); r->gap_seq_cap = ns + 64; }
    if (nr) HIP_OK(hipMemcpyAsync(r->d_gap_rows, r->gap_rows.data(), sizeof(UvcGapRow) * (size_t)nr, hipMemcpyHostToDevice, cs));
    if (ns) HIP_OK(hipMemcpyAsync(r->d_gap_seq, r->gap_seq.data(), (size_t)ns, hipMemcpyHostToDevice, cs));
    if (nr || ns) HIP_OK(hipStreamSynchronize(cs));
    if (na) {
        HIP_OK(hipMemcpyAsync(r->d_gap_alleles, r->gap_alleles.data(), sizeof(UvcIndelAllele) * (size_t)na, hipMemcpyHostToDevice, cs));
        HIP_OK(hipMemcpyAsync(r->d_gap_allele_row, r->gap_allele_row.data(), sizeof(int32_t) * (size_t)na, hipMemcpyHostToDevice, cs));
        HIP_OK(hipStreamSynchronize(cs));
    }
    r->gap_ready = true;
    return 0;
}

static int uvcgpu_region_indel_alleles_impl(uvcgpu_region_t *r, UvcGapRow *rows, int64_t row_capacity, int64_t *n_rows, uint8_t *seq, int64_t seq_capacity, int64_t *seq_bytes) {
    if (!r) return fail(UVCGPU_EINVAL, "null region");
    if (!r->accumulated) return fail(UVCGPU_ESTATE, "indel_alleles before accumulate");
    int rc = gap_tables(r);
    if (rc) return rc;
    if (n_rows) *n_rows = (int64_t)r->gap_rows.size();
    if (seq_bytes) *seq_bytes = (int64_t)r->gap_seq.size();
    if ((int64_t)r->gap_rows.size() > row_capacity || (int64_t)r->gap_seq.size() > seq_capacity) return fail(UVCGPU_ENOMEM, "allele table capacity too small");
    if (!r->gap_rows.empty()) memcpy(rows, r->gap_rows.data(), r->gap_rows.size() * sizeof(UvcGapRow));
    if (!r->gap_seq.empty()) memcpy(seq, r->gap_seq.data(), r->gap_seq.size());
    return 0;
}

// ---- haplotype links: hap_bq / hap_fq / hap_f2q of updateByRegion3Aln (main.hpp:3665-3742) ----
// Built on request (the record writer and uvcgpu_region_hap_links ask): candidate fragments / units -> their mutated (position, symbol)
// lists on the device (k_hap_*), maps + updateHapMap on the host (uvc_hap.cpp).  Needs the planes (P5's cDPM / cDPm).
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int hap_tables(uvcgpu_region_t *r) {
    if (r->hap_ready) return 0;
    if (!r->accumulated) return fail(UVCGPU_ESTATE, "haplotype links before accumulate");
    if (r->state_released) return fail(UVCGPU_ESTATE, "the planes were released by the last score (UvcScoreRequest::release_state)");
    const RegionDev &R = r->R;
    std::vector<int32_t> events;
    const bool timing = (getenv("UVCGPU_TIMING") != nullptr);
    double t_prev = now_s();
    auto lap = [&](const char *what, long long n) { if (timing) { const double t = now_s(); fprintf(stderr, "[uvcgpu hap_links] %-34s %8.2f ms (%lld)\n", what, 1e3 * (t - t_prev), n); t_prev = t; } };
    struct Tmp { std::vector<void *> p; hipStream_t s; ~Tmp() { (void)hipStreamSynchronize(s); for (void *q : p) hipFree(q); } } tmp; tmp.s = r->stream;
    auto get = [&](size_t bytes) -> void * { void *q = nullptr; if (hipMalloc(&q, std::max<size_t>(bytes, 8)) != hipSuccess) return nullptr; tmp.p.push_back(q); return q; };
    for (int units = 0; units < 2; units++) {
        if (!units && !r->P.inferred_is_vcf_generated) continue;   // P3 belongs to updateByAlns3UsingBQ (main.hpp:3691)
        const size_t n = (size_t)(units ? R.n_fs : R.n_frags);
        if (n == 0) continue;
        HapWork H; memset(&H, 0, sizeof(H));
        H.cand = (int32_t *)get(4 * n); H.cand_off = (int32_t *)get(4 * n); H.cand_cap = (int32_t *)get(4 * n);
        char *ctr = (char *)get(16);
        if (!H.cand || !H.cand_off || !H.cand_cap || !ctr) return fail(UVCGPU_ENOMEM, "hipMalloc(haplotype candidates)");
        H.n_cand = (int32_t *)ctr; H.total = (unsigned long long *)(ctr + 8);
        HIP_OK(hipMemsetAsync(ctr, 0, 16, r->stream));
        uvc_launch_hap_cand(&R, &H, units, r->stream);
        struct { int32_t n_cand, pad; unsigned long long total; } c;
        HIP_OK(hipMemcpyAsync(&c, ctr, 16, hipMemcpyDeviceToHost, r->stream));
        HIP_OK(hipStreamSynchronize(r->stream));
        lap(units ? "candidates of units" : "candidates of fragments", (long long)c.n_cand);
        if (c.n_cand == 0) continue;
        if (c.total >= ((unsigned long long)1 << 31)) return fail(UVCGPU_EUNSUPPORTED, "more than 2^31 haplotype event slots in one region");
        H.events = (int32_t *)get(4 * (size_t)c.total);
        if (!H.events) return fail(UVCGPU_ENOMEM, "hipMalloc(haplotype events)");
        HIP_OK(hipMemsetAsync(H.events, 0xFF, 4 * (size_t)c.total, r->stream));
        uvc_launch_hap_events(&R, &r->P, &H, units, c.n_cand, r->stream);
        HIP_OK(hipGetLastError());
        const size_t at = events.size();
        events.resize(at + (size_t)c.total);
        HIP_OK(hipMemcpyAsync(events.data() + at, H.events, 4 * (size_t)c.total, hipMemcpyDeviceToHost, r->stream));
        HIP_OK(hipStreamSynchronize(r->stream));
        lap(units ? "events of units + D2H" : "events of fragments + D2H", (long long)c.total);
    }
    { const int rc = uvcgpu_region_sync(r); if (rc) return rc; }
    uvc_hap_build(events.data(), (int64_t)events.size(), r->beg, r->npos, r->P.phasing_haplotype_max_count, r->P.phasing_haplotype_min_ad, r->P.phasing_haplotype_max_detail_cnt, r->hap);
    lap("maps + links (host)", (long long)(r->hap[0].size() + r->hap[1].size() + r->hap[2].size()));
    r->hap_ready = true;
    return 0;
}
const std::vector<UvcHapLinkHost> *uvcgpu_region_hap_(uvcgpu_region_t *r) { return hap_tables(r) ? nullptr : r->hap; }   // for uvc_vcf.cpp

static int uvcgpu_region_hap_links_impl(uvcgpu_region_t *r, UvcHapLink *links, int64_t link_capacity, int64_t *n_links, int32_t *muts, int64_t mut_capacity, int64_t *n_mut_ints) {
    if (!r) return fail(UVCGPU_EINVAL, "null region");
    int rc = hap_tables(r);
    if (rc) return rc;
    int64_t nl = 0, nm = 0;
    for (int w = 0; w < 3; w++) for (const UvcHapLinkHost &h : r->hap[w]) { nl++; nm += 2 * (int64_t)h.form.size(); }
    if (n_links) *n_links = nl;
    if (n_mut_ints) *n_mut_ints = nm;
    if (nl > link_capacity || nm > mut_capacity || (nl && !links) || (nm && !muts)) return fail(UVCGPU_ENOMEM, "haplotype link capacity too small");
    int64_t li = 0, mi = 0;
    for (int w = 0; w < 3; w++) for (const UvcHapLinkHost &h : r->hap[w]) {
        UvcHapLink &o = links[li++];
        o.which = w; o.n_muts = (int32_t)h.form.size(); o.mut_off = mi; o.fr_cnt[0] = h.fr[0]; o.fr_cnt[1] = h.fr[1]; o.other_cnt[0] = h.other[0]; o.other_cnt[1] = h.other[1];
        for (const auto &ps : h.form) { muts[mi++] = ps.first; muts[mi++] = ps.second; }
    }
    return 0;
}

int64_t uvcgpu_region_score_size(const uvcgpu_region_t *r, const UvcScoreRequest *req) {
    if (!r) return -1;
    const int64_t np = (req && req->pos_beg >= 0) ? (req->pos_end - req->pos_beg) : r->npos;
    return NSYM * (np + 1) + (req ? req->n_indel_alleles + req->n_tumor_keys : 0) + (int64_t)r->gap_alleles.size();
}

// Small host arrays of a score call go to the device through the handle's own page-locked staging buffer: an asynchronous copy straight from
// the caller's pageable memory lets the runtime map those heap pages for the GPU (read-only, as a copy source), and heap pages come back to the
// caller in other roles -- as a records buffer the next copy writes to.  `at` advances through the staging buffer within one call.
static int stage_upload(uvcgpu_region_t *r, void *dst, const void *src, size_t bytes, size_t &at, size_t total) {
    if (total > r->h_stage_cap) {
        if (r->h_stage) { (void)hipStreamSynchronize(r->stream); (void)hipHostFree(r->h_stage); }
        r->h_stage = nullptr; r->h_stage_cap = 0;
        const size_t want = total + total / 2 + 4096;
        if (hipHostMalloc((void **)&r->h_stage, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fail(UVCGPU_ENOMEM, "hipHostMalloc(staging)"); }
        r->h_stage_cap = want;
    }
    memcpy(r->h_stage + at, src, bytes);
    if (hipMemcpyAsync(dst, r->h_stage + at, bytes, hipMemcpyHostToDevice, r->stream) != hipSuccess) return fail(UVCGPU_EDEVICE, "hipMemcpyAsync(staging)");
    at += (bytes + 63) & ~(size_t)63;
    return 0;
}

static int uvcgpu_region_score_impl(uvcgpu_region_t *r, const UvcScoreRequest *req, UvcScoreOut *out) {
    if (!r || !out || !out->fields) return fail(UVCGPU_EINVAL, "bad argument");
    if (!r->accumulated) return fail(UVCGPU_ESTATE, "score before accumulate");
    if (r->state_released) return fail(UVCGPU_ESTATE, "the planes were released by the last score (UvcScoreRequest::release_state)");
    UvcScoreRequest rq; memset(&rq, 0, sizeof(rq)); rq.pos_beg = -1;
    if (req) rq = *req;
    if (rq.pos_beg < 0) { rq.pos_beg = r->beg + 1; rq.pos_end = r->end - 1; }
    // zerobased_pos == region begin is legal (a contig that starts inside the region: main.cpp:617-619 guards its "base in front" with BASE_NN);
    // with base_at_pos_beg the BASE sub-position of pos_beg reads refpos pos_beg - 1, which must be inside the region
    if (rq.pos_beg < r->beg + (rq.base_at_pos_beg ? 1 : 0) || rq.pos_end > r->end - 1 || rq.pos_end < rq.pos_beg) return fail(UVCGPU_EINVAL, "score range outside the region");
    // InDel alleles: the region's own tables (fill_by_indel_info / indel_get_majority); a (refpos, symbol) the caller lists is overridden
    { int rc0 = gap_tables(r); if (rc0) return rc0; }
    UvcIndelAllele *d_al = nullptr; int32_t *d_al_row = nullptr; UvcTumorKey *d_tk = nullptr;
    std::vector<UvcIndelAllele> merged; std::vector<int32_t> merged_row;
    // the temporaries go back to the caching allocator, which hands them to other handles at once: on every return path the stream is
    // drained first (async copies from `merged` / the caller's keys, kernels that read the blocks), then they are freed
    struct Temps { uvcgpu_region *r; UvcIndelAllele *&a; int32_t *&b; UvcTumorKey *&c;
                   ~Temps() { if (a || b || c) (void)hipStreamSynchronize(r->stream); if (a) hipFree(a); if (b) hipFree(b); if (c) hipFree(c); } } temps = { r, d_al, d_al_row, d_tk };
    const UvcIndelAllele *use_al = r->d_gap_alleles; const int32_t *use_row = r->d_gap_allele_row; int64_t n_al = (int64_t)r->gap_alleles.size();
    // one staging layout per call (an earlier call's copies are complete: score synchronises before it returns)
    size_t stage_at = 0;
    const size_t stage_total = (sizeof(UvcIndelAllele) + sizeof(int32_t)) * (size_t)(r->gap_alleles.size() + (size_t)std::max<int64_t>(rq.n_indel_alleles, 0)) + sizeof(UvcTumorKey) * (size_t)std::max<int64_t>(rq.n_tumor_keys, 0) + 256;
    if (rq.n_indel_alleles > 0) {
        auto less = [](const UvcIndelAllele &a, const UvcIndelAllele &b) { return a.refpos < b.refpos || (a.refpos == b.refpos && a.symbol < b.symbol); };
        for (int64_t q = 1; q < rq.n_indel_alleles; q++) if (less(rq.indel_alleles[q], rq.indel_alleles[q - 1])) return fail(UVCGPU_EINVAL, "indel_alleles must be sorted by (refpos, symbol)");
        size_t i = 0; int64_t j = 0;
        while (i < r->gap_alleles.size() || j < rq.n_indel_alleles) {
            const bool take_own = (j >= rq.n_indel_alleles) || (i < r->gap_alleles.size() && less(r->gap_alleles[i], rq.indel_alleles[j]));
            if (take_own) { merged.push_back(r->gap_alleles[i]); merged_row.push_back(r->gap_allele_row[i]); i++; continue; }
            const UvcIndelAllele key = rq.indel_alleles[j];
            while (i < r->gap_alleles.size() && !less(key, r->gap_alleles[i])) i++;   // drop the library's alleles of this (refpos, symbol)
            while (j < rq.n_indel_alleles && !less(key, rq.indel_alleles[j])) { merged.push_back(rq.indel_alleles[j]); merged_row.push_back(-1); j++; }
        }
        n_al = (int64_t)merged.size();
        HIP_OK(hipMalloc((void **)&d_al, sizeof(UvcIndelAllele) * (size_t)n_al));
        HIP_OK(hipMalloc((void **)&d_al_row, sizeof(int32_t) * (size_t)n_al));
        { int rc1 = stage_upload(r, d_al, merged.data(), sizeof(UvcIndelAllele) * (size_t)n_al, stage_at, stage_total); if (rc1) return rc1; }
        { int rc1 = stage_upload(r, d_al_row, merged_row.data(), sizeof(int32_t) * (size_t)n_al, stage_at, stage_total); if (rc1) return rc1; }
        use_al = d_al; use_row = d_al_row;
    }
    if (r->P.tumor_vcf_is_provided && rq.n_tumor_keys > 0) {   // normal sample of a T/N pair: the tumor records, sorted by (refpos, symbol)
        for (int64_t q = 1; q < rq.n_tumor_keys; q++) {
            const UvcTumorKey &a = rq.tumor_keys[q - 1], &b = rq.tumor_keys[q];
            if (a.refpos > b.refpos || (a.refpos == b.refpos && a.symbol > b.symbol)) return fail(UVCGPU_EINVAL, "tumor_keys must be sorted by (refpos, symbol)");
        }
        HIP_OK(hipMalloc((void **)&d_tk, sizeof(UvcTumorKey) * rq.n_tumor_keys));
        { int rc1 = stage_upload(r, d_tk, rq.tumor_keys, sizeof(UvcTumorKey) * (size_t)rq.n_tumor_keys, stage_at, stage_total); if (rc1) return rc1; }
    }
    const bool kept_only = (rq.kept_only != 0);
    // device capacity: the caller's in the plain form; with kept_only the caller's buffer only has to hold the kept groups, the device
    // array every record -- start from a guess and grow once if the count says so
    int64_t cap = std::max<int64_t>(out->capacity, 1);
    if (kept_only) cap = std::max<int64_t>(std::max<int64_t>(cap, r->score_capacity), (int64_t)(rq.pos_end - rq.pos_beg) / 8 + 4096);
    int rc = 0;
    int64_t cnt[2] = { 0, 0 };
    for (int attempt = 0; attempt < 2 && !rc; attempt++) {
        if (cap > r->score_capacity) {
            if (r->d_score_fields) hipFree(r->d_score_fields);
            r->d_score_fields = nullptr; r->score_capacity = 0;
            HIP_OK(hipMalloc((void **)&r->d_score_fields, sizeof(int32_t) * UVC_NUM_SCORE_FIELDS * cap));
            r->score_capacity = cap;
        }
        if (kept_only && r->score_kept_capacity != r->score_capacity) {
            if (r->d_score_kept) hipFree(r->d_score_kept);
            r->d_score_kept = nullptr; r->score_kept_capacity = 0;
            HIP_OK(hipMalloc((void **)&r->d_score_kept, sizeof(int32_t) * UVC_NUM_SCORE_FIELDS * r->score_capacity));
            r->score_kept_capacity = r->score_capacity;
        }
        {   // the staged rows of the scoring kernels are sized by the record capacity
            const size_t need = uvc_score_scratch_bytes(rq.pos_end - rq.pos_beg, r->score_capacity);
            if (need > r->score_scratch_bytes) {
                if (r->d_score_scratch) hipFree(r->d_score_scratch);
                r->d_score_scratch = nullptr; r->score_scratch_bytes = 0;
                HIP_OK(hipMalloc((void **)&r->d_score_scratch, need + need / 8));
                r->score_scratch_bytes = need + need / 8;
            }
            r->d_score_count = (int64_t *)r->d_score_scratch;   // the record counts head the scratch (zeroed with the scan states)
        }
        HIP_OK(hipMemsetAsync(r->d_score_scratch, 0, uvc_score_scratch_zero_bytes(rq.pos_end - rq.pos_beg, r->score_capacity), r->stream));
        int pi = -1;   // with profiling on, the scoring kernels (gate + scan + k_score + k_call + the kept-groups copy) as one more entry of uvcgpu_region_kernel_times
        if (r->prof.on && r->prof.n < 32) { pi = r->prof.n++; r->prof.name[pi] = "k_score_all"; if (!r->prof.ev[pi][0]) { hipEventCreate(&r->prof.ev[pi][0]); hipEventCreate(&r->prof.ev[pi][1]); } hipEventRecord(r->prof.ev[pi][0], r->stream); }
        rc = uvc_launch_score(&r->R, &r->P, &rq, use_al, use_row, n_al, r->d_gap_rows, r->d_gap_seq, d_tk, r->d_score_fields, r->score_capacity, r->d_score_scratch,
                              kept_only ? r->d_score_kept : nullptr, r->stream);
        if (pi >= 0) hipEventRecord(r->prof.ev[pi][1], r->stream);
        if (!rc && hipGetLastError() != hipSuccess) rc = fail(UVCGPU_EDEVICE, "score kernel launch failed");
        if (!rc) rc = uvcgpu_region_sync(r);
        // copies on the handle's own stream: a null-stream hipMemcpy would also wait for every other handle's work
        if (!rc && (hipMemcpyAsync(cnt, r->d_score_count, 16, hipMemcpyDeviceToHost, r->stream) != hipSuccess || hipStreamSynchronize(r->stream) != hipSuccess)) rc = fail(UVCGPU_EDEVICE, "hipMemcpy(count)");
        if (rc || !kept_only || cnt[0] <= r->score_capacity) break;
        cap = cnt[0] + cnt[0] / 8;   // the guess was too small: nothing was written, once more with room for every record
    }
    if (!rc) {
        const int64_t n_out = (kept_only ? cnt[1] : cnt[0]);
        r->last_scored = cnt[0]; r->last_returned = n_out;
        const int32_t *src = (kept_only ? r->d_score_kept : r->d_score_fields);
        out->n_records = n_out;
        if (kept_only && cnt[0] > r->score_capacity) rc = fail(UVCGPU_EDEVICE, "score: the record count changed between two passes");
        else if (n_out > out->capacity) rc = fail(UVCGPU_ENOMEM, "score output capacity too small");   // the planes stay: the caller comes back with a larger buffer
        else if (rq.release_state && r->side) {   // the scoring kernels are done: zero the planes on the side stream under the D2H of the records
            if (hipEventRecord(r->e_fork, r->stream) == hipSuccess && hipStreamWaitEvent(r->side, r->e_fork, 0) == hipSuccess
                && zero_state(r, r->side) == 0 && hipEventRecord(r->e_join, r->side) == hipSuccess) {
                r->state_released = true; r->state_zeroed = true; r->zeroed_bytes = r->state_bytes;
            }
        }
        if (!rc && n_out > 0 && (hipMemcpy2DAsync(out->fields, sizeof(int32_t) * out->capacity, src, sizeof(int32_t) * r->score_capacity, sizeof(int32_t) * n_out, UVC_NUM_SCORE_FIELDS, hipMemcpyDeviceToHost, r->stream) != hipSuccess
                             || hipStreamSynchronize(r->stream) != hipSuccess))
            rc = fail(UVCGPU_EDEVICE, "hipMemcpy2D(records)");
    }
    return rc;   // ~Temps frees the temporaries
}

int uvcgpu_region_create(uvcgpu_region_t **out, const UvcParams *params, int32_t tid, int32_t beg, int32_t end, const char *refseq) { return guarded("uvcgpu_region_create", [&] { return uvcgpu_region_create_impl(out, params, tid, beg, end, refseq); }); }
int uvcgpu_region_reset(uvcgpu_region_t *r, int32_t tid, int32_t beg, int32_t end, const char *refseq) { return guarded("uvcgpu_region_reset", [&] { return uvcgpu_region_reset_impl(r, tid, beg, end, refseq); }); }
int uvcgpu_region_set_reads(uvcgpu_region_t *r, const UvcReadSoA *in) { return guarded("uvcgpu_region_set_reads", [&] { return uvcgpu_region_set_reads_impl(r, in); }); }
int uvcgpu_region_set_reads_device(uvcgpu_region_t *r, const UvcReadSoA *in) { return guarded("uvcgpu_region_set_reads_device", [&] { return uvcgpu_region_set_reads_device_impl(r, in); }); }
int uvcgpu_region_correct_bq(uvcgpu_region_t *r) { return guarded("uvcgpu_region_correct_bq", [&] { return uvcgpu_region_correct_bq_impl(r); }); }
int uvcgpu_region_accumulate(uvcgpu_region_t *r) { return guarded("uvcgpu_region_accumulate", [&] { return uvcgpu_region_accumulate_impl(r); }); }
int uvcgpu_region_fetch(uvcgpu_region_t *r, int32_t g, void *dst, int64_t dst_bytes) { return guarded("uvcgpu_region_fetch", [&] { return uvcgpu_region_fetch_impl(r, g, dst, dst_bytes); }); }
int uvcgpu_region_fetch_columns(uvcgpu_region_t *r, const int32_t *refpos, int64_t n, int64_t *dst) { return guarded("uvcgpu_region_fetch_columns", [&] { return uvcgpu_region_fetch_columns_impl(r, refpos, n, dst); }); }
int uvcgpu_region_indel_alleles(uvcgpu_region_t *r, UvcGapRow *rows, int64_t row_capacity, int64_t *n_rows, uint8_t *seq, int64_t seq_capacity, int64_t *seq_bytes) { return guarded("uvcgpu_region_indel_alleles", [&] { return uvcgpu_region_indel_alleles_impl(r, rows, row_capacity, n_rows, seq, seq_capacity, seq_bytes); }); }
int uvcgpu_region_hap_links(uvcgpu_region_t *r, UvcHapLink *links, int64_t link_capacity, int64_t *n_links, int32_t *muts, int64_t mut_capacity, int64_t *n_mut_ints) { return guarded("uvcgpu_region_hap_links", [&] { return uvcgpu_region_hap_links_impl(r, links, link_capacity, n_links, muts, mut_capacity, n_mut_ints); }); }
int uvcgpu_region_score(uvcgpu_region_t *r, const UvcScoreRequest *req, UvcScoreOut *out) { return guarded("uvcgpu_region_score", [&] { return uvcgpu_region_score_impl(r, req, out); }); }

void uvcgpu_region_destroy(uvcgpu_region_t *r) {
    if (!r) return;
    quiesce(r);
    free_reads(r);
    if (r->d_refsym) hipFree(r->d_refsym);
    if (r->d_rtr) hipFree(r->d_rtr);
    if (r->d_rtr0) hipFree(r->d_rtr0);
    if (r->d_baq) hipFree(r->d_baq);
    if (r->d_rtrwork) hipFree(r->d_rtrwork);
    if (r->h_ref) (void)hipHostFree(r->h_ref);
    if (r->d_fsum) hipFree(r->d_fsum);
    if (r->d_win) hipFree(r->d_win);
    if (r->side) hipStreamDestroy(r->side);
    if (r->side3) hipStreamDestroy(r->side3);
    if (r->e_join3) hipEventDestroy(r->e_join3);
    if (r->e_stat) hipEventDestroy(r->e_stat);
    if (r->e_alleles) hipEventDestroy(r->e_alleles);
    if (r->e_fork) hipEventDestroy(r->e_fork);
    if (r->e_join) hipEventDestroy(r->e_join);
    if (r->e_fork2) hipEventDestroy(r->e_fork2);
    if (r->d_state) hipFree(r->d_state);
    if (r->d_dirty) hipFree(r->d_dirty);
    if (r->d_zero_planes) hipFree(r->d_zero_planes);
    if (r->R.err) hipFree(r->R.err);
    if (r->d_score_scratch) hipFree(r->d_score_scratch);
    if (r->d_score_fields) hipFree(r->d_score_fields);
    if (r->d_score_kept) hipFree(r->d_score_kept);
    if (r->h_stage) (void)hipHostFree(r->h_stage);
    if (r->d_gap_alleles) hipFree(r->d_gap_alleles);
    if (r->d_gap_allele_row) hipFree(r->d_gap_allele_row);
    if (r->d_gap_rows) hipFree(r->d_gap_rows);
    if (r->d_gap_seq) hipFree(r->d_gap_seq);
    if (r->stream) hipStreamDestroy(r->stream);
    delete r;
    { const hipError_t stale = hipGetLastError(); if (stale != hipSuccess && getenv("UVCGPU_DEBUG")) fprintf(stderr, "[uvcgpu] uvcgpu_region_destroy left HIP error: %s\n", hipGetErrorString(stale)); }
}

}  // extern "C"
