// Micro-benchmark (VERDICT r1 #4a): the two wave layouts for the per-position accumulators of P2, on the same synthetic pileup.
//
//   A  lanes = 64 consecutive positions, the wave loops over the reads that cover them (what k_p2_fast / k_frag do): per-read quantities
//      are wave-uniform (scalar), the base|qual byte pair of (read, position) is a coalesced 128-byte load, a counter whose condition is
//      uniform costs one masked add, a per-lane condition a compare + add-with-carry, sums are plain per-lane adds.
//   B  lanes = 64 reads, the wave walks the positions of an LDS-staged [64 reads x 64 positions] tile (BASELINE.json north_star: "LDS-staged
//      per-position read tiles + wavefront reductions across reads"): every condition is a per-lane compare whose SGPR mask is counted
//      with s_bcnt1 (the "ballot + popcount" counter), sums are DPP / shuffle reductions, and the per-position totals of a tile have to be
//      added to the running totals of the position (LDS) because 64 positions x 14 counters do not fit in SGPRs.
//
// Both count the same 12 conditional counters + 2 sums per (read, position) cell -- a third of what dealwith_segbias does, in the same
// proportions of uniform / per-lane conditions -- and both results are checked against each other.  Output: ns per cell and instructions
// the compiler emitted per 64 cells (read from the disassembly by hand: see DESIGN.md section 6).
// Build: hipcc --offload-arch=gfx950 -O3 layout_ab.hip -o layout_ab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define NPOS 65536          // positions
#define DEPTH 320           // reads covering every position (5 read tiles of 64)
#define NC 14               // 12 counters + 2 sums
struct ReadRec { int thr_q, strand, isize_ok, clip, mapq_ok, pad0, pad1, pad2; };   // the per-read scalars the conditions look at

// cell(r, p): base | qual << 8 at bq[r * NPOS + p] (reads laid out like reads of equal start: every read covers every position; the real
// kernels pay a binary search and a range test on top, the same in both layouts)
__device__ __forceinline__ void cell_counts(int bqv, const ReadRec &R, int ref, int c[NC]) {
    const int b = bqv & 0xFF, q = bqv >> 8;
    const bool match = (b == ref);
    c[0] += 1;                                   // depth
    c[1] += R.strand;                            // uniform in A
    c[2] += R.isize_ok;                          // uniform in A
    c[3] += R.mapq_ok;                           // uniform in A
    c[4] += (R.clip > 0);                        // uniform in A
    c[5] += (R.strand & R.isize_ok);             // uniform in A
    c[6] += (q >= R.thr_q);                      // per lane in both
    c[7] += (q >= 20);
    c[8] += match;
    c[9] += (match && q >= 30);
    c[10] += (!match && q >= R.thr_q);
    c[11] += (q < 10);
    c[12] += q;                                  // sums
    c[13] += (q * q) >> 5;
}

// ---- layout A: lanes = positions ----
__global__ void __launch_bounds__(256) k_lanes_positions(const unsigned short *bq, const ReadRec *reads, const unsigned char *ref, int *out) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    int c[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) c[k] = 0;
    const int rf = ref[p];
    for (int r = 0; r < DEPTH; r++) {
        const ReadRec R = reads[r];              // wave-uniform address: scalar loads
        cell_counts(bq[(size_t)r * NPOS + p], R, rf, c);
    }
#pragma unroll
    for (int k = 0; k < NC; k++) out[(size_t)k * NPOS + p] = c[k];
}

// ---- layout B: lanes = reads over an LDS-staged tile ----
__device__ __forceinline__ int wave_sum(int v) { for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d); return v; }
__global__ void __launch_bounds__(64) k_lanes_reads(const unsigned short *bq, const ReadRec *reads, const unsigned char *ref, int *out) {
    __shared__ unsigned short tile[64][64 + 2];   // [read][position], padded: a column read is conflict-free
    __shared__ int tot[NC][64];                   // running totals of the tile's 64 positions
    const int lane = threadIdx.x, p0 = blockIdx.x * 64;
    for (int k = 0; k < NC; k++) tot[k][lane] = 0;
    for (int r0 = 0; r0 < DEPTH; r0 += 64) {
        // stage: 64 reads x 64 positions, each row a coalesced 128-byte load
        for (int rr = 0; rr < 64; rr++) tile[rr][lane] = bq[(size_t)(r0 + rr) * NPOS + p0 + lane];
        __syncthreads();
        const ReadRec R = reads[r0 + lane];       // per-lane read scalars
        for (int pp = 0; pp < 64; pp++) {
            int c[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) c[k] = 0;
            cell_counts(tile[lane][pp], R, ref[p0 + pp], c);
            // counters: ballot + popcount; sums: wave reduction
#pragma unroll
            for (int k = 0; k < 12; k++) { const int n = __popcll(__ballot(c[k] != 0)); if (lane == 0) tot[k][pp] += n; }
            const int s0 = wave_sum(c[12]), s1 = wave_sum(c[13]);
            if (lane == 0) { tot[12][pp] += s0; tot[13][pp] += s1; }
        }
        __syncthreads();
    }
    for (int k = 0; k < NC; k++) out[(size_t)k * NPOS + p0 + lane] = tot[k][lane];
}

int main() {
    std::vector<unsigned short> h_bq((size_t)DEPTH * NPOS); std::vector<ReadRec> h_r(DEPTH); std::vector<unsigned char> h_ref(NPOS);
    srand(7);
    for (auto &v : h_ref) v = rand() & 3;
    for (size_t i = 0; i < h_bq.size(); i++) { const int p = (int)(i % NPOS); const int b = (rand() % 100 < 2) ? (rand() & 3) : h_ref[p]; h_bq[i] = (unsigned short)(b | ((2 + rand() % 40) << 8)); }
    for (auto &r : h_r) { r.thr_q = 15 + rand() % 10; r.strand = rand() & 1; r.isize_ok = rand() % 10 != 0; r.clip = rand() % 20 == 0; r.mapq_ok = rand() % 15 != 0; r.pad0 = r.pad1 = r.pad2 = 0; }
    unsigned short *d_bq; ReadRec *d_r; unsigned char *d_ref; int *d_a, *d_b;
    hipMalloc(&d_bq, h_bq.size() * 2); hipMalloc(&d_r, h_r.size() * sizeof(ReadRec)); hipMalloc(&d_ref, NPOS); hipMalloc(&d_a, (size_t)NC * NPOS * 4); hipMalloc(&d_b, (size_t)NC * NPOS * 4);
    hipMemcpy(d_bq, h_bq.data(), h_bq.size() * 2, hipMemcpyHostToDevice); hipMemcpy(d_r, h_r.data(), h_r.size() * sizeof(ReadRec), hipMemcpyHostToDevice); hipMemcpy(d_ref, h_ref.data(), NPOS, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double cells = (double)NPOS * DEPTH;
    float ms[2] = { 0, 0 };
    for (int which = 0; which < 2; which++) {
        for (int rep = 0; rep < 6; rep++) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_lanes_positions, dim3(NPOS / 256), dim3(256), 0, 0, d_bq, d_r, d_ref, d_a);
            else hipLaunchKernelGGL(k_lanes_reads, dim3(NPOS / 64), dim3(64), 0, 0, d_bq, d_r, d_ref, d_b);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1);
            if (rep == 0 || t < ms[which]) ms[which] = t;
        }
    }
    std::vector<int> a((size_t)NC * NPOS), b((size_t)NC * NPOS);
    hipMemcpy(a.data(), d_a, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d_b, b.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < a.size(); i++) bad += (a[i] != b[i]);
    printf("layout A (lanes = positions): %.3f ms, %.4f ns per (read, position) cell, %.1f G cells/s\n", ms[0], ms[0] * 1e6 / cells, cells / ms[0] / 1e6);
    printf("layout B (lanes = reads, LDS tile, ballot + s_bcnt1 counters, wave-reduced sums): %.3f ms, %.4f ns per cell, %.1f G cells/s\n", ms[1], ms[1] * 1e6 / cells, cells / ms[1] / 1e6);
    printf("B / A = %.2f; results %s (%zu of %zu values differ)\n", ms[1] / ms[0], bad ? "DIFFER" : "identical", bad, a.size());
    return bad ? 1 : 0;
}
