// scan_kernels.h -- device code of the MI355X (gfx950) read-to-whitelist scan.
//
// What it replaces: the `for i in whitelist: sub_edit_distance(i, bc, best_ed)` loop of
// /root/reference/blaze/read_assignment.py:66-85 (N_whitelist third-party calls per read)
// and the exact-match shortcut of :61-62.
//
// Design (CDNA4, wave64, integer VALU bound -- no MFMA, this is bit-vector work):
//   * lane = read.  The read's <=20-nt extension lives in 40 VGPRs as per-column
//     bit-plane masks (all-ones / all-zeros), constant over the whole scan.
//   * the whitelist is packed 2 bits/base into two 16-bit planes per barcode and TWO
//     barcodes share one 32-bit word per plane (low/high half) -> every VALU op scores two
//     (read, barcode) pairs.  Wave-uniform operands are fetched with vector loads into VGPRs:
//     an SGPR source halves the issue rate of a VALU instruction on gfx950 (measured).
//   * per pair the kernels run the k-error Shift-And automaton (Wu-Manber form of the
//     Myers/Hyyro family: pure and/or/shift, no carries, so two 16-bit automata run in one
//     32-bit register with no cross-talk: the only bit that leaks from the low half's shift
//     lands on the high half's always-one bit).  The automaton accepts iff the semi-global edit
//     distance of the barcode inside the extension is <= K = max_ed.
//   * accepting pairs are rare (~1 per read).  They take a divergent, exact path that
//     re-runs the automaton un-packed, records the per-column distance, derives
//     (ed, end_idx) under the configured tie rule and feeds the reference's
//     ORDER-DEPENDENT fold (ed<best / ed==best / 'ambiguous' / best-=1), always in whitelist order.
//   * three scans share that core: k_scan_seeded (pigeonhole seeds pick the candidates, product
//     path), k_scan_filtered (3-gram set filter over the whole whitelist, in order) and
//     k_scan_bitparallel (every pair, the all-pairs roofline leg).  All give identical results.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace bz {

constexpr int kBcLen = 16;        // config.py:34
constexpr int kTextCols = 20;     // 2 + 16 + 2 (config.py:75, read_assignment.py:70)
constexpr int kMaxK = 8;          // the packed automaton is instantiated for max_ed 0..8 (seeded forms: 0..3, all pairs above)
constexpr uint32_t kOne2 = 0x00010001u;
constexpr int kHitNone = -1;
constexpr int kHitAmbiguous = -2;

constexpr unsigned kFlagExact = 1u, kFlagBlank = 2u, kFlagAmbig = 4u, kFlagGeneric = 8u, kFlagInvalid = 16u;

__device__ __forceinline__ uint32_t base_code(uint32_t c) { return (c >> 1) & 3u; }   // A0 C1 T2 G3
__device__ __forceinline__ bool is_acgt(uint32_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

// First 20 bytes of a row as five little-endian words.  Rows of the packed layout (stride a multiple of 8, >= 24, base
// 8-byte aligned: 24 is the tightest stride that holds a 20-nt extension, 32 what the file pipeline uses for barcodes of
// up to 28 nt) take three naturally aligned 8-byte loads; anything else is read byte by byte.
struct RowBytes {
    uint32_t w[8];
    __device__ __forceinline__ uint32_t at(int j) const { return (w[j >> 2] >> (8 * (j & 3))) & 0xFFu; }
};
__host__ __device__ constexpr bool packed_stride(int stride) { return (stride & 7) == 0 && stride >= 24; }
__device__ __forceinline__ RowBytes load_row_packed(const uint8_t *row) {      // packed layout only (host checks stride and alignment)
    RowBytes b;
    const uint2 a = *(const uint2 *)row, c = *(const uint2 *)(row + 8), d = *(const uint2 *)(row + 16);
    b.w[0] = a.x; b.w[1] = a.y; b.w[2] = c.x; b.w[3] = c.y; b.w[4] = d.x; b.w[5] = d.y; b.w[6] = 0; b.w[7] = 0;
    return b;
}
__device__ __forceinline__ RowBytes load_row(const uint8_t *row, int stride) {
    if (packed_stride(stride)) return load_row_packed(row);   // wave-uniform
    RowBytes b;
#pragma unroll
    for (int k = 0; k < 8; ++k) b.w[k] = 0;
    const int nb = stride < kTextCols ? stride : kTextCols;
    for (int j = 0; j < nb; ++j) b.w[j >> 2] |= (uint32_t)row[j] << (8 * (j & 3));
    return b;
}

// ------------------------------------------------------------------------------------------
// k-error Shift-And state for one 32-bit word (two 16-bit automata when ONE = 0x00010001).
// R[d] bit i  <=>  barcode[0..i] matches some suffix of the text read so far with <= d errors.
// Q[d] = (R[d] << 1) | ONE is kept beside it so that a column costs 2 + 3K ops.
template <int K, uint32_t ONE>
struct ShiftAnd {
    uint32_t R[K + 1], Q[K + 1];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int d = 0; d <= K; ++d) {
            R[d] = ((1u << d) - 1u) * ONE;            // first d barcode bases deleted
            Q[d] = ((1u << (d + 1)) - 1u) * ONE;
        }
    }
    __device__ __forceinline__ void step(uint32_t eq) {
        uint32_t oldR = R[0], oldQ = Q[0];
        R[0] = oldQ & eq;                             // v_and
        Q[0] = (R[0] << 1) | ONE;                     // v_lshl_or
#pragma unroll
        for (int d = 1; d <= K; ++d) {
            uint32_t r = R[d], q = Q[d];
            uint32_t t = (q & eq) | oldR;             // v_and_or : match | insertion
            R[d] = t | oldQ | Q[d - 1];               // v_or3    : substitution | deletion
            Q[d] = (R[d] << 1) | ONE;                 // v_lshl_or
            oldR = r; oldQ = q;
        }
    }
};

// Exact (ed, end) of ONE barcode (16-bit planes bh/bl) inside the lane's text, by re-running the
// automaton un-packed and reading the accept bit of every level at every real column.
// cnt = number of levels whose accept bit is set = K+1-D[16][j] (levels are nested).
template <int K>
__device__ __forceinline__ void exact_pair(uint32_t bh, uint32_t bl, const uint32_t (&XH)[kTextCols],
                                           const uint32_t (&XL)[kTextCols], int len, int tie, int &ed, int &end) {
    ShiftAnd<K, 1u> A;
    A.init();
    int best_cnt = 0, best_end = -1;
#pragma unroll
    for (int j = 0; j < kTextCols; ++j) {
        uint32_t eq = (bh ^ XH[j]) & (bl ^ XL[j]);
        A.step(eq);
        if (j >= kBcLen - K - 1) {
            int cnt = 0;
#pragma unroll
            for (int d = 0; d <= K; ++d) cnt += (A.R[d] >> (kBcLen - 1)) & 1u;
            bool better = (cnt > best_cnt) || (tie && cnt == best_cnt && cnt > 0);
            if (j < len && better) { best_cnt = cnt; best_end = j; }
        }
    }
    ed = K + 1 - best_cnt;
    end = best_end;
}

// The reference's fold, literally (read_assignment.py:73-85).  hit: -1 '' / -2 'ambiguous' / index.
__device__ __forceinline__ void fold_update(int ed, int end_idx, int idx, int &best, int &hit, int &hit_end, bool &done) {
    if (done) return;
    if (ed < best) { best = ed; hit = idx; hit_end = end_idx; }
    else if (ed == best) {
        if (hit == kHitNone) { hit = idx; hit_end = end_idx; }
        else { hit = kHitAmbiguous; best -= 1; if (best < 0) done = true; }
    }
}

// The same with an EMPTY whitelist entry in play (a blank line of --known-bc-list: '' is within distance 0 of every
// text, and as `bc_hit` it is falsy, so `if not bc_hit` of :76 lets the next tie replace it and :87 reports no hit).
__device__ __forceinline__ void fold_update_falsy(int ed, int end_idx, int idx, bool empty, int &best, int &hit, int &hit_end,
                                                  bool &hit_empty, bool &done) {
    if (done) return;
    if (ed < best) { best = ed; hit = idx; hit_end = end_idx; hit_empty = empty; }
    else if (ed == best) {
        if (hit == kHitNone || hit_empty) { hit = idx; hit_end = end_idx; hit_empty = empty; }
        else { hit = kHitAmbiguous; hit_empty = false; best -= 1; if (best < 0) done = true; }
    }
}

// ------------------------------------------------------------------------------------------
// Totals: [assigned, rows seen] go to kTotStripes striped slots behind counts[n_wl + 2] (one atomic pair per
// workgroup) and are folded into counts[n_wl], counts[n_wl+1] by k_fold_totals before anyone reads them.
// Per-barcode counts of one run.  Two forms:
//  * whitelists up to kHistLdsBins barcodes: kHistBlocks workgroups each keep a private histogram of ALL bins in LDS
//    (ds_add, no global atomics), write it to partial[block][bin], and k_hist_reduce adds the partials into counts[]
//    (one owner per bin).  Global atomics from 8 XCDs on a 40 KB table ran at ~13 G/s = 55 us per 1M rows; this form
//    takes a pass over hit[] at streaming rate.
//  * larger whitelists: one global atomic per assigned row (no contention to speak of on millions of bins).
// Both add [assigned, rows seen] to the striped totals.
constexpr int kTotStripes = 64;
constexpr int kHistLdsBins = 32768;
constexpr int kHistBlocks = 64;
constexpr int kHistThreads = 1024;
// A run's per-barcode counts are not needed before somebody asks for them, and the two histogram kernels were 12 us of a
// 94 us step that does nothing else meanwhile.  The engine therefore DEFERS them (blaze_hip.hip: Slot::pend): the next run
// on the slot carries them along -- workgroups in front of k_classify_packed's grid take the private histograms of the
// previous run's hits (its results sit in the slot's other hit buffer), workgroups in front of k_pair_candidates' grid (or
// k_scan_seeded's: max_ed 0, 1, 3) add them into counts[] -- and whoever reads counts, list counters or timings first has the pending histogram run on its own.
// Whitelists up to kPiggyBins barcodes (the histogram rides in k_classify_packed's dynamic LDS).
constexpr int kPiggyBins = 12287;

// one workgroup's share of k_hist_partial (kHistThreads threads, (n_wl + 1) words of dynamic LDS)
__device__ __forceinline__ void hist_partial_block(int block, const int32_t *__restrict__ hit, int n, int n_wl,
                                                   uint32_t *__restrict__ partial, unsigned long long *__restrict__ counts) {
    extern __shared__ uint32_t s_bins[];                          // n_wl bins + 1 (assigned)
    for (int b = threadIdx.x; b <= n_wl; b += kHistThreads) s_bins[b] = 0;
    __syncthreads();
    const int per = (((n + kHistBlocks - 1) / kHistBlocks) + 3) & ~3;     // rows per block, multiple of 4 (16-byte loads)
    const int lo = block * per, hi = min(n, lo + per);
    int assigned = 0;
    // four 16-byte loads in flight per thread before their LDS atomics (one after another, a block's 15.6K rows of a 1M-row
    // step were four dependent load -> atomics rounds per thread)
    constexpr int kU = 4;
    for (int i0 = lo + 4 * threadIdx.x; i0 < hi; i0 += kU * 4 * kHistThreads) {
        int4 h[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int i = i0 + u * 4 * kHistThreads;
            h[u] = make_int4(-1, -1, -1, -1);
            if (i + 4 <= hi) h[u] = *(const int4 *)(hit + i);
            else if (i < hi) {                                    // the block's ragged end (at most once per block)
                h[u].x = hit[i];
                if (i + 1 < hi) h[u].y = hit[i + 1];
                if (i + 2 < hi) h[u].z = hit[i + 2];
            }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (h[u].x >= 0) { atomicAdd(&s_bins[h[u].x], 1u); ++assigned; }
            if (h[u].y >= 0) { atomicAdd(&s_bins[h[u].y], 1u); ++assigned; }
            if (h[u].z >= 0) { atomicAdd(&s_bins[h[u].z], 1u); ++assigned; }
            if (h[u].w >= 0) { atomicAdd(&s_bins[h[u].w], 1u); ++assigned; }
        }
    }
    if (assigned) atomicAdd(&s_bins[n_wl], (uint32_t)assigned);
    __syncthreads();
    uint32_t *out = partial + (size_t)block * n_wl;
    for (int b = threadIdx.x; b < n_wl; b += kHistThreads) out[b] = s_bins[b];
    if (threadIdx.x == 0) {
        unsigned long long *tot = counts + n_wl + 2 + 2 * (block & (kTotStripes - 1));
        if (s_bins[n_wl]) atomicAdd(&tot[0], (unsigned long long)s_bins[n_wl]);
        if (hi > lo) atomicAdd(&tot[1], (unsigned long long)(hi - lo));
    }
}

__global__ __launch_bounds__(kHistThreads) void k_hist_partial(const int32_t *__restrict__ hit, int n, int n_wl,
                                                              uint32_t *__restrict__ partial, unsigned long long *__restrict__ counts) {
    hist_partial_block((int)blockIdx.x, hit, n, n_wl, partial, counts);
}

// End-of-run duties of the kernel that finishes a run's histogram: publish the run's list counters to the (host-mapped)
// pinned buffer the timing call reads, and zero them: the slot's runs alternate between two counter sets, and a set is
// clean again before the run after next appends to it (so a run needs neither a memset nor a device-to-host copy dispatch).
__device__ __forceinline__ void finish_run(bool first_block, int32_t *cur, int32_t *host_out) {
    if (first_block && threadIdx.x < 4) { host_out[threadIdx.x] = cur[threadIdx.x]; cur[threadIdx.x] = 0; }
}

// bins [first_bin, first_bin + n_threads): the kHistBlocks private histograms added into counts[]
__device__ __forceinline__ void hist_reduce_bin(int b, const uint32_t *__restrict__ partial, int n_wl, unsigned long long *__restrict__ counts) {
    if (b >= n_wl) return;
    unsigned long long s = 0;
#pragma unroll
    for (int k0 = 0; k0 < kHistBlocks; k0 += 16) {               // sixteen loads in flight (all 64 cost 130 registers)
        uint32_t v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = partial[(size_t)(k0 + k) * n_wl + b];
#pragma unroll
        for (int k = 0; k < 16; ++k) s += v[k];
    }
    if (s) atomicAdd(&counts[b], s);      // the other slot's run may be adding to the same bin on its own stream
}

__global__ __launch_bounds__(256) void k_hist_reduce(const uint32_t *__restrict__ partial, int n_wl, unsigned long long *__restrict__ counts,
                                                    int32_t *cur, int32_t *host_out) {
    finish_run(blockIdx.x == 0, cur, host_out);
    hist_reduce_bin((int)(blockIdx.x * blockDim.x + threadIdx.x), partial, n_wl, counts);
}

__global__ __launch_bounds__(256) void k_histogram(const int32_t *__restrict__ hit, int n, int n_wl,
                                                  unsigned long long *__restrict__ counts,
                                                  int32_t *cur, int32_t *host_out) {
    __shared__ unsigned int s_assigned[4];
    finish_run(blockIdx.x == 0, cur, host_out);
    int assigned = 0;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const int h = hit[r];
        if (h >= 0) { atomicAdd(&counts[h], 1ull); ++assigned; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) assigned += __shfl_down(assigned, o, 64);
    if ((threadIdx.x & 63) == 0) s_assigned[threadIdx.x >> 6] = (unsigned int)assigned;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long a = (unsigned long long)s_assigned[0] + s_assigned[1] + s_assigned[2] + s_assigned[3];
        const long long first = (long long)blockIdx.x * blockDim.x;
        long long rows = 0;                                   // rows this block visited (grid-stride)
        const long long step = (long long)gridDim.x * blockDim.x;
        for (long long b = first; b < n; b += step) rows += min((long long)blockDim.x, (long long)n - b);
        unsigned long long *tot = counts + n_wl + 2 + 2 * (blockIdx.x & (kTotStripes - 1));
        if (a) atomicAdd(&tot[0], a);
        if (rows) atomicAdd(&tot[1], (unsigned long long)rows);
    }
}

// ------------------------------------------------------------------------------------------
// Full-rate instruction selection (measured on MI355X, tools/valu_microbench.hip): with all-VGPR
// operands v_xor/v_and/v_or/v_add_u32 and the 3-input v_bitop3_b32 issue at the SIMD-32 rate
// (~2 cycles per wave64 instruction); v_lshlrev, v_and_or, v_or3, v_lshl_or, every v_pk_* and ANY
// instruction with an SGPR source run at half that.  So the hot loop below is written as
//   - x+x (v_add_u32) instead of x<<1,
//   - explicit v_bitop3_b32 truth tables instead of and_or / or3,
//   - barcode planes and the constant 0x00010001 held in VGPRs (broadcast vector loads),
// which makes one text column cost 2 (match mask) + 2 + 4K full-rate ops for TWO pairs.
constexpr uint32_t kTA = 0xF0, kTB = 0xCC, kTC = 0xAA;           // LUT3 operand patterns
#define BZ_LUT3(a, b, c, expr) __builtin_amdgcn_bitop3_b32((a), (b), (c), (uint32_t)((expr) & 0xFF))

__device__ __forceinline__ uint32_t dbl(uint32_t x) {            // x << 1 at the full VALU rate
    uint32_t s;
    asm("v_add_u32 %0, %1, %1" : "=v"(s) : "v"(x));
    return s;
}
__device__ __forceinline__ uint32_t in_vgpr(uint32_t x) {        // pin a (uniform) value into a VGPR
    asm volatile("" : "+v"(x));
    return x;
}

__device__ __forceinline__ uint32_t popc_add(uint32_t x, uint32_t acc) {   // popcount(x) + acc in ONE instruction
    uint32_t d;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
    return d;
}

// Packed k-error Shift-And in the "S = R + R" form: Q = S | ONE is never materialised, ONE rides
// along as a bitop3 operand.  R[d]' = ((S[d]|ONE) & eq) | R[d-1] | S[d-1] | S[d-1]' | ONE.
template <int K>
struct PackedScan {
    uint32_t R[K + 1], S[K + 1];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int d = 0; d <= K; ++d) { R[d] = ((1u << d) - 1u) * kOne2; S[d] = 2u * R[d]; }
    }
    template <bool LAST>
    __device__ __forceinline__ void step(uint32_t eq, uint32_t one) {
        uint32_t oldR = R[0], oldS = S[0];
        R[0] = BZ_LUT3(oldS, one, eq, (kTA | kTB) & kTC);
        if (!LAST || K > 0) S[0] = dbl(R[0]);
#pragma unroll
        for (int d = 1; d <= K; ++d) {
            const uint32_t r = R[d], sd = S[d];
            uint32_t t = BZ_LUT3(sd, one, eq, (kTA | kTB) & kTC);
            t = BZ_LUT3(t, oldR, oldS, kTA | kTB | kTC);
            R[d] = BZ_LUT3(t, S[d - 1], one, kTA | kTB | kTC);
            if (!LAST || d < K) S[d] = dbl(R[d]);
            oldR = r; oldS = sd;
        }
    }
};

// Bit-parallel scan: one lane = one read of `list`, every lane walks the whole whitelist in order.
// planes[p] = {hi-bit plane, lo-bit plane} of barcodes 2p (bits 0..15) and 2p+1 (bits 16..31).
// n_pairs is a multiple of 2 (host pads); barcodes with index >= n_wl are padding and ignored.
// The planes are wave-uniform but are fetched with a (laundered) vector load so they arrive in
// VGPRs; the next iteration's 16 bytes are requested before the current one is scored.
template <int K>
__global__ __launch_bounds__(256) void k_scan_bitparallel(
    const uint4 *__restrict__ planes, int n_pairs, int n_wl, const uint8_t *__restrict__ ext, int ext_stride,
    const uint8_t *__restrict__ ext_len, const int32_t *__restrict__ list, const int32_t *__restrict__ n_list_ptr, int tie,
    int32_t *__restrict__ hit_out, int8_t *__restrict__ ed_out, int8_t *__restrict__ end_out, uint8_t *__restrict__ flags_out) {
    const int n_list = *n_list_ptr;
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if ((slot & ~63) >= n_list) return;                       // whole wave idle (wave-uniform)
    const bool active = slot < n_list;
    const int r = list[active ? slot : n_list - 1];
    const int len = ext_len[r];

    // text bit-plane masks, inverted so that eq = (BH ^ XH) & (BL ^ XL)
    uint32_t XH[kTextCols], XL[kTextCols];
    const RowBytes rb = load_row(ext + (size_t)r * ext_stride, ext_stride);
#pragma unroll
    for (int j = 0; j < kTextCols; ++j) {
        uint32_t c = (j < len) ? rb.at(j) : (uint32_t)'A';   // columns past the text are padding
        uint32_t code = base_code(c);
        XH[j] = (code & 2u) ? 0u : 0xFFFFFFFFu;
        XL[j] = (code & 1u) ? 0u : 0xFFFFFFFFu;
    }
    const uint32_t one = in_vgpr(kOne2);

    int best = K, hit = kHitNone, hit_end = -1;
    bool done = !active;

    const int n_quads = n_pairs >> 1;                         // one uint4 = two pairs = four barcodes
    uint32_t q_idx = in_vgpr(0u);                             // divergent-looking index -> vector load
    uint4 cur = planes[q_idx];
    constexpr int kFirst = kBcLen - K - 1;                    // first column a hit can end in
    for (int q = 0; q < n_quads; ++q) {
        const uint4 pl4 = cur;
        q_idx = in_vgpr((uint32_t)min(q + 1, n_quads - 1));
        cur = planes[q_idx];                                  // prefetch the next four barcodes
        PackedScan<K> A, B;
        A.init();
        B.init();
        uint32_t HA = 0, HB = 0, PA = 0, PB = 0;
#pragma unroll
        for (int j = 0; j < kTextCols; ++j) {
            const uint32_t eqa = BZ_LUT3(pl4.x ^ XH[j], pl4.y, XL[j], kTA & (kTB ^ kTC));
            const uint32_t eqb = BZ_LUT3(pl4.z ^ XH[j], pl4.w, XL[j], kTA & (kTB ^ kTC));
            if (j == kTextCols - 1) { A.template step<true>(eqa, one); B.template step<true>(eqb, one); }
            else { A.template step<false>(eqa, one); B.template step<false>(eqb, one); }
            if (j >= kFirst) {                                // accumulate accept bits two columns per op
                if (((j - kFirst) & 1) == 0) { PA = A.R[K]; PB = B.R[K]; }
                else { HA = BZ_LUT3(HA, PA, A.R[K], kTA | kTB | kTC); HB = BZ_LUT3(HB, PB, B.R[K], kTA | kTB | kTC); }
            }
        }
        if (((kTextCols - 1 - kFirst) & 1) == 0) { HA |= PA; HB |= PB; }
        // accept bits: bit15 = even barcode, bit31 = odd barcode of the pair
        uint32_t cand = ((HA >> 15) & 1u) | ((HA >> 30) & 2u) | ((HB >> 13) & 4u) | ((HB >> 28) & 8u);
        if (done) cand = 0;
        while (cand) {                                        // rare, divergent, exact
            const int k = __ffs(cand) - 1;
            cand &= cand - 1;
            const uint32_t ph = (k & 2) ? pl4.z : pl4.x, plo = (k & 2) ? pl4.w : pl4.y;
            const uint32_t sh = (k & 1) ? 16u : 0u;
            const int idx = 4 * q + k;
            int ed, end;
            exact_pair<K>((ph >> sh) & 0xFFFFu, (plo >> sh) & 0xFFFFu, XH, XL, len, tie, ed, end);
            if (ed <= K && idx < n_wl) fold_update(ed, end, idx, best, hit, hit_end, done);
        }
    }
    if (active) {
        hit_out[r] = hit >= 0 ? hit : -1;
        ed_out[r] = (int8_t)best;
        end_out[r] = (int8_t)(hit >= 0 ? hit_end : -1);
        flags_out[r] = (uint8_t)(hit == kHitAmbiguous ? kFlagAmbig : 0u);
    }
}

// ------------------------------------------------------------------------------------------
// Filtered scan (the product path for max_ed <= 2): an EXACT q-gram pre-filter in front of the
// same packed automaton.
//
// Lemma (Jokinen & Ukkonen 1991, for approximate substring matching): if the 16-nt barcode P
// occurs in the text T with e <= K edit operations, every operation spoils at most 3 of P's 14
// 3-gram positions, so at least 14-3K of them occur verbatim in T.  Hence the number of DISTINCT
// 3-grams of P that are absent from T is <= 3K.  With the 64 possible 3-grams as one 64-bit set
// per barcode (W, wave-uniform) and per read (E, per lane):   popcount(W & ~E) <= 3K
// is necessary for sub_edit_distance(P, T) <= K.  Pairs that fail it can never enter the
// reference's fold (read_assignment.py:73-85 ignores ed > best_ed), so skipping them is exact.
// For uniform 16-mers ~2.3 % of the pairs pass at K = 2.
//
// Structure per lane (= one read), no cross-lane traffic, no barriers:
//   filter : 32 barcodes per block, 2 v_and + 2 v_bcnt + 1 v_alignbit per pair -> a 32-bit pass
//            mask (MSB = first barcode of the block);
//   push   : mask bits -> barcode indices into the lane's own FIFO in LDS (whitelist order kept);
//   verify : when some lane's FIFO is full (or the whitelist is exhausted) every lane pops two
//            candidates, packs their planes into one register pair and runs the packed K-error
//            automaton over the 20 text columns; accepting candidates take the rare exact path
//            and the reference's order-dependent fold, in FIFO (= whitelist) order.
constexpr int kQCap = 32;                                         // FIFO entries per lane (power of two)
constexpr int kFilterBlock = 32;                                  // barcodes per filter block

template <int K>
__device__ __forceinline__ void scan_filtered_block(
    int block, int n_list, uint32_t (&fifo)[kQCap][256],
    const uint4 *__restrict__ grams, const uint32_t *__restrict__ planes32, int n_blk, int n_wl,
    const uint8_t *__restrict__ ext, int ext_stride, const uint8_t *__restrict__ ext_len,
    const int32_t *__restrict__ list, int tie,
    int32_t *__restrict__ hit_out, int8_t *__restrict__ ed_out, int8_t *__restrict__ end_out, uint8_t *__restrict__ flags_out) {
    const int tid = threadIdx.x;
    const int slot = block * 256 + tid;
    if ((slot & ~63) >= n_list) return;                           // whole wave idle (wave-uniform)
    const bool active = slot < n_list;
    const int r = list[active ? slot : n_list - 1];
    const int len = ext_len[r];

    uint32_t XH[kTextCols], XL[kTextCols];
    uint32_t nelo = 0xFFFFFFFFu, nehi = 0xFFFFFFFFu;              // ~E: 3-grams NOT in the real text
    {
        const RowBytes rb = load_row(ext + (size_t)r * ext_stride, ext_stride);
        uint32_t g = 0;
#pragma unroll
        for (int j = 0; j < kTextCols; ++j) {
            const uint32_t c = (j < len) ? rb.at(j) : (uint32_t)'A';
            const uint32_t code = base_code(c);
            XH[j] = (code & 2u) ? 0u : 0xFFFFFFFFu;
            XL[j] = (code & 1u) ? 0u : 0xFFFFFFFFu;
            g = (g >> 2) | (code << 4);                           // code[j-2] | code[j-1]<<2 | code[j]<<4
            if (j >= 2 && j < len) {
                if (g & 32u) nehi &= ~(1u << (g & 31u)); else nelo &= ~(1u << (g & 31u));
            }
        }
    }
    const uint32_t one = in_vgpr(kOne2);
    const uint32_t thr = in_vgpr((uint32_t)(-(3 * K + 1)));       // diff = missing - (3K+1) < 0  <=>  pass

    int best = K, hit = kHitNone, hit_end = -1;
    bool done = !active;
    uint32_t head = 0, tail = 0, mask = 0, base = 0;
    int blk = 0;
    constexpr int kFirst = kBcLen - K - 1;

    for (;;) {
        if (__builtin_amdgcn_ballot_w64(mask != 0) != 0) {
            const bool full = mask != 0 && (tail - head) == (uint32_t)kQCap;
            if (__builtin_amdgcn_ballot_w64(full) == 0) {         // push one candidate per lane
                if (mask != 0) {
                    const uint32_t b = (uint32_t)__builtin_clz(mask);
                    mask &= ~(0x80000000u >> b);
                    fifo[tail & (kQCap - 1)][tid] = base + b;
                    ++tail;
                }
                continue;
            }
        } else if (blk < n_blk) {                                 // filter the next 32 barcodes
            base = (uint32_t)blk * kFilterBlock;
            uint32_t m = 0;
            uint32_t goff = in_vgpr((uint32_t)blk * (kFilterBlock / 2) * (uint32_t)sizeof(uint4));   // wave-uniform, vector loads
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int i = 0; i < kFilterBlock / 4; ++i) {
                    const uint4 w = *(const uint4 *)((const char *)grams + goff + i * sizeof(uint4));
                    const uint32_t d0 = popc_add(w.y & nehi, popc_add(w.x & nelo, thr));
                    const uint32_t d1 = popc_add(w.w & nehi, popc_add(w.z & nelo, thr));
                    m = __builtin_amdgcn_alignbit(m, d0, 31);     // m = m << 1 | sign(d0)
                    m = __builtin_amdgcn_alignbit(m, d1, 31);
                }
                goff += (kFilterBlock / 4) * (uint32_t)sizeof(uint4);
            }
            mask = done ? 0u : m;
            ++blk;
            continue;
        } else if (__builtin_amdgcn_ballot_w64(tail != head) == 0) {
            break;
        }
        // ---- verify round: pop up to two candidates per lane, oldest first
        const uint32_t cnt = tail - head;
        const bool have = cnt != 0;
        uint32_t i0 = 0, i1 = 0;
        if (have) {
            i0 = fifo[head & (kQCap - 1)][tid];
            i1 = cnt > 1 ? fifo[(head + 1) & (kQCap - 1)][tid] : i0;
            head += cnt > 1 ? 2u : 1u;
        }
        const uint32_t p0 = planes32[i0], p1 = planes32[i1];
        const uint32_t bh = (p0 & 0xFFFFu) | (p1 << 16), bl = (p0 >> 16) | (p1 & 0xFFFF0000u);
        PackedScan<K> A;
        A.init();
        uint32_t H = 0, P = 0;
#pragma unroll
        for (int j = 0; j < kTextCols; ++j) {
            const uint32_t eq = BZ_LUT3(bh ^ XH[j], bl, XL[j], kTA & (kTB ^ kTC));
            if (j == kTextCols - 1) A.template step<true>(eq, one);
            else A.template step<false>(eq, one);
            if (j >= kFirst) {
                if (((j - kFirst) & 1) == 0) P = A.R[K];
                else H = BZ_LUT3(H, P, A.R[K], kTA | kTB | kTC);
            }
        }
        if (((kTextCols - 1 - kFirst) & 1) == 0) H |= P;
        uint32_t cand = ((H >> 15) & 1u) | ((H >> 30) & 2u);
        if (!have || done) cand = 0;
        if (cnt < 2) cand &= 1u;
        while (cand) {                                            // rare, divergent, exact, in order
            const int k = __ffs(cand) - 1;
            cand &= cand - 1;
            const uint32_t p = k ? p1 : p0;
            int ed, end;
            exact_pair<K>(p & 0xFFFFu, p >> 16, XH, XL, len, tie, ed, end);
            if (ed <= K) fold_update(ed, end, (int)(k ? i1 : i0), best, hit, hit_end, done);
        }
    }
    if (active) {
        hit_out[r] = hit >= 0 ? hit : -1;
        ed_out[r] = (int8_t)best;
        end_out[r] = (int8_t)(hit >= 0 ? hit_end : -1);
        flags_out[r] = (uint8_t)(hit == kHitAmbiguous ? kFlagAmbig : 0u);
    }
}

template <int K>
__global__ __launch_bounds__(256) void k_scan_filtered(
    const uint4 *__restrict__ grams, const uint32_t *__restrict__ planes32, int n_blk, int n_wl,
    const uint8_t *__restrict__ ext, int ext_stride, const uint8_t *__restrict__ ext_len,
    const int32_t *__restrict__ list, const int32_t *__restrict__ n_list_ptr, int tie,
    int32_t *__restrict__ hit_out, int8_t *__restrict__ ed_out, int8_t *__restrict__ end_out, uint8_t *__restrict__ flags_out) {
    __shared__ uint32_t fifo[kQCap][256];                         // [entry][thread]: bank = thread, conflict free
    scan_filtered_block<K>(blockIdx.x, *n_list_ptr, fifo, grams, planes32, n_blk, n_wl, ext, ext_stride, ext_len, list, tie,
                           hit_out, ed_out, end_out, flags_out);
}

// ------------------------------------------------------------------------------------------
// Seeded scan (the product path for max_ed <= 2): pigeonhole seeds + the same packed verify.
//
// Lemma (pigeonhole): cut the 16-nt barcode P into K+1 disjoint pieces.  An alignment of P inside
// the text T with <= K edit operations leaves at least one piece untouched, and that piece then
// occurs verbatim in T.  With s the start of the matched substring (0 <= s, the text has only
// len-16+K spare columns) and d in [-K, K] the net indel shift in front of the piece, a piece at
// barcode offset o starts at text column j = s + o + d; the loose window j in [o-2, o+6]
// (clipped to the text) covers every case for len <= 20 and K <= 2.  So the set of barcodes that
// share one piece with one of those <= 9 text windows is a SUPERSET of {w : sub_ed(w, T) <= K}; a
// barcode outside it can never enter the reference's fold (read_assignment.py:73-85).
//
// Plan per K (SeedPlan): K = 3 -> 4|4|4|4 nt (256 buckets each, below); K = 2 -> pieces 5|5|6 nt (1024 / 1024 / 4096 buckets); K <= 1 -> 8|8 nt
// (65536 buckets each; sound for K = 0 too).  The host builds, per piece, a bucket table over the
// piece's 2-bit code (CSR: tab[code] .. tab[code+1] into `members`, barcode indices).  For a 5k
// whitelist a read meets ~100 candidates instead of 5000 pairs.
//
// Per lane (= read): for each piece the 9 bucket ranges are looked up at once (independent
// gathers, one latency), their members are streamed two per range per step into the lane's FIFO in
// LDS, and whenever a lane's FIFO cannot take another step every lane pops two candidates and runs
// the packed K-error automaton on them (same verify as k_scan_filtered).  Candidates arrive in
// index order only within a bucket, so accepted ones (rare) are recorded as EVENTS (index, ed, end)
// and the reference's order-dependent fold is replayed over the lane's events sorted by whitelist
// index (duplicates dropped) at the end.  A lane with more than kMaxEvents events is not folded
// here: its row goes to `redo_list` and the order-preserving k_scan_filtered finishes it.
// K = 3 (--max-edit-distance 3, parser.py:100): four 4-nt pieces, one of them untouched by <= 3 operations.  The window
// bound generalises: a piece at barcode offset o starts at text column j = s + o + d with s >= 0 the start of the match and
// d >= -K the net indel shift in front of the piece, so j >= o - K; the match (length 16 + ins - del) must end inside 20
// columns, s <= 4 - ins + del, hence j <= 4 + o + (deletions behind the piece) <= o + 4 + K.  K = 3: 11 windows per piece,
// 44 probes of 256-bucket tables: ~860 candidates per read of a 5k-cell list instead of 5000 pairs.
constexpr int kMaxEvents = 8;

template <int K> struct SeedPlan;
template <> struct SeedPlan<2> {
    static constexpr int n = 3, shifts = 9, first = -2;
    static constexpr int len(int p) { return p == 2 ? 6 : 5; }
    static constexpr int off(int p) { return 5 * p; }
};
template <> struct SeedPlan<1> {
    static constexpr int n = 2, shifts = 9, first = -2;
    static constexpr int len(int) { return 8; }
    static constexpr int off(int p) { return 8 * p; }
};
template <> struct SeedPlan<0> : SeedPlan<1> {};
template <> struct SeedPlan<3> {
    static constexpr int n = 4, shifts = 11, first = -3;
    static constexpr int len(int) { return 4; }
    static constexpr int off(int p) { return 4 * p; }
};
template <class Plan> constexpr int seed_tab_base(int p) {       // where piece p's CSR table starts inside the plan's table
    int b = 0;
    for (int q = 0; q < p; ++q) b += (1 << (2 * Plan::len(q))) + 1;
    return b;
}

template <int K>
__global__ __launch_bounds__(256) void k_scan_seeded(
    const int32_t *__restrict__ tab, const int32_t *__restrict__ members, const uint32_t *__restrict__ planes32, int n_wl,
    const uint8_t *__restrict__ ext, int ext_stride, const uint8_t *__restrict__ ext_len,
    const int32_t *__restrict__ list, const int32_t *__restrict__ n_list_ptr, int tie,
    int32_t *__restrict__ hit_out, int8_t *__restrict__ ed_out, int8_t *__restrict__ end_out, uint8_t *__restrict__ flags_out,
    const uint32_t *__restrict__ red_partial, int red_n_wl, unsigned long long *__restrict__ red_counts,
    int32_t *red_cur, int32_t *red_host) {
    // the second half of the slot's deferred histogram (see kPiggyBins), as in front of k_pair_candidates' grid
    const int red_blocks = red_partial ? (red_n_wl + 255) / 256 : 0;
    if ((int)blockIdx.x < red_blocks) {                           // workgroup-uniform
        finish_run(blockIdx.x == 0, red_cur, red_host);
        hist_reduce_bin((int)blockIdx.x * 256 + (int)threadIdx.x, red_partial, red_n_wl, red_counts);
        return;
    }
    using Plan = SeedPlan<K>;
    __shared__ uint32_t fifo[kQCap][256];
    // The fold (read_assignment.py:73-85) consumes at most TWO events of each distance class, always the first two in
    // whitelist order (a class is dead for good once best_ed has dropped below it): slots 2c and 2c + 1 keep the two
    // smallest events of class c (event word = index << 7 | ed << 5 | end, so words of one class order by index).
    // However many near matches a row has -- at max_ed 3 a 5k-cell list gives most rows several -- 2 (K + 1) slots
    // decide it, and nothing overflows.  (The first form kept the first 8 events met and handed the row to the ordered
    // scan at the ninth: at max_ed 3 that was most rows, each a whole-whitelist pass.)
    constexpr int kSlots = 2 * (K + 1);
    static_assert(kSlots <= kMaxEvents, "event slots");
    __shared__ uint32_t events[kMaxEvents][256];
    const int n_list = *n_list_ptr;
    const int tid = threadIdx.x;
    const int slot = ((int)blockIdx.x - red_blocks) * 256 + tid;
    if ((slot & ~63) >= n_list) return;                           // whole wave idle (wave-uniform)
    const bool active = slot < n_list;
    const int r = list[active ? slot : n_list - 1];
    const int len = ext_len[r];
#pragma unroll
    for (int a = 0; a < kSlots; ++a) events[a][tid] = 0xFFFFFFFFu;

    uint32_t XH[kTextCols], XL[kTextCols];
    uint64_t tc = 0;                                              // 2-bit codes of the text, column j at bits 2j
    {
        const RowBytes rb = load_row(ext + (size_t)r * ext_stride, ext_stride);
#pragma unroll
        for (int j = 0; j < kTextCols; ++j) {
            const uint32_t c = (j < len) ? rb.at(j) : (uint32_t)'A';
            const uint32_t code = base_code(c);
            XH[j] = (code & 2u) ? 0u : 0xFFFFFFFFu;
            XL[j] = (code & 1u) ? 0u : 0xFFFFFFFFu;
            tc |= (uint64_t)code << (2 * j);
        }
    }
    const uint32_t one = in_vgpr(kOne2);
    uint32_t head = 0, tail = 0;
    constexpr int kFirst = kBcLen - K - 1;
    const bool full_text = len == kTextCols;

    // one verify round: every lane pops up to two candidates (oldest first) and runs the packed automaton
    auto verify_round = [&]() {
        const uint32_t cnt = tail - head;
        const bool have = cnt != 0;
        uint32_t i0 = 0, i1 = 0;
        if (have) {
            i0 = fifo[head & (kQCap - 1)][tid];
            i1 = cnt > 1 ? fifo[(head + 1) & (kQCap - 1)][tid] : i0;
            head += cnt > 1 ? 2u : 1u;
        }
        const uint32_t p0 = planes32[i0], p1 = planes32[i1];
        const uint32_t bh = (p0 & 0xFFFFu) | (p1 << 16), bl = (p0 >> 16) | (p1 & 0xFFFF0000u);
        PackedScan<K> A;
        A.init();
        uint32_t H[K + 1], P[K + 1];                              // accept bits per level, two columns per op
#pragma unroll
        for (int d = 0; d <= K; ++d) { H[d] = 0; P[d] = 0; }
#pragma unroll
        for (int j = 0; j < kTextCols; ++j) {
            const uint32_t eq = BZ_LUT3(bh ^ XH[j], bl, XL[j], kTA & (kTB ^ kTC));
            if (j == kTextCols - 1) A.template step<true>(eq, one);
            else A.template step<false>(eq, one);
            if (j >= kFirst) {
#pragma unroll
                for (int d = 0; d <= K; ++d) {
                    if (((j - kFirst) & 1) == 0) P[d] = A.R[d];
                    else H[d] = BZ_LUT3(H[d], P[d], A.R[d], kTA | kTB | kTC);
                }
            }
        }
        if (((kTextCols - 1 - kFirst) & 1) == 0) {
#pragma unroll
            for (int d = 0; d <= K; ++d) H[d] |= P[d];
        }
        uint32_t cand = ((H[K] >> 15) & 1u) | ((H[K] >> 30) & 2u);
        if (!have || !active) cand = 0;
        if (cnt < 2) cand &= 1u;
        while (cand) {                                            // accepted candidates: a few per row at most
            const int k = __ffs(cand) - 1;
            cand &= cand - 1;
            const uint32_t sh = k ? 31u : 15u;
            int ed = K, end = 0;
            if (full_text) {
                // on a 20-column text the lowest level that accepts in any column IS the edit distance (the minimum over
                // all end positions); the end index is needed for the fold's winner only and is computed after the fold
#pragma unroll
                for (int d = K - 1; d >= 0; --d) if ((H[d] >> sh) & 1u) ed = d;
            } else {                                              // padding columns could lower the level: exact, on the real columns
                const uint32_t p = k ? p1 : p0;
                exact_pair<K>(p & 0xFFFFu, p >> 16, XH, XL, len, tie, ed, end);
            }
            if (ed <= K) {                                        // a barcode met through several seeds is recorded once
                const uint32_t ev = ((k ? i1 : i0) << 7) | ((uint32_t)ed << 5) | (uint32_t)end;
                const uint32_t lo = events[2 * ed][tid], hi = events[2 * ed + 1][tid];
                if (ev < lo) { events[2 * ed][tid] = ev; events[2 * ed + 1][tid] = lo; }
                else if (ev > lo && ev < hi) events[2 * ed + 1][tid] = ev;
            }
        }
    };

    constexpr int kSeedShifts = Plan::shifts;                     // text windows per piece
    constexpr int kSeedBurst = 2 * kSeedShifts;                   // most FIFO pushes per step
    static_assert(kSeedBurst < kQCap, "a step's pushes must fit the FIFO");
#pragma unroll 1
    for (int p = 0; p < Plan::n; ++p) {
        const int lp = Plan::len(p);
        const int op = Plan::off(p);
        const int32_t *tp = tab + seed_tab_base<Plan>(p);
        const int32_t *mp = members + (size_t)p * n_wl;
        int cur[kSeedShifts], end[kSeedShifts];
#pragma unroll
        for (int d = 0; d < kSeedShifts; ++d) {                   // all windows' bucket bounds: independent gathers
            const int j = op + Plan::first + d;
            cur[d] = 0; end[d] = 0;
            if (active && j >= 0 && j + lp <= len) {              // window inside the real text
                const uint32_t code = (uint32_t)(tc >> (2 * j)) & ((1u << (2 * lp)) - 1u);
                cur[d] = tp[code]; end[d] = tp[code + 1];
            }
        }
        for (;;) {
            bool any = false;
#pragma unroll
            for (int d = 0; d < kSeedShifts; ++d) any = any || cur[d] < end[d];
            if (__builtin_amdgcn_ballot_w64(any) == 0) break;
            while (__builtin_amdgcn_ballot_w64((tail - head) > (uint32_t)(kQCap - kSeedBurst)) != 0) verify_round();
            int32_t m0[kSeedShifts], m1[kSeedShifts];
#pragma unroll
            for (int d = 0; d < kSeedShifts; ++d) {               // up to two members per range, all loads in flight
                m0[d] = cur[d] < end[d] ? mp[cur[d]] : -1;
                m1[d] = cur[d] + 1 < end[d] ? mp[cur[d] + 1] : -1;
            }
#pragma unroll
            for (int d = 0; d < kSeedShifts; ++d) {
                if (m0[d] >= 0) { fifo[tail & (kQCap - 1)][tid] = (uint32_t)m0[d]; ++tail; }
                if (m1[d] >= 0) { fifo[tail & (kQCap - 1)][tid] = (uint32_t)m1[d]; ++tail; }
                cur[d] += 2;
            }
        }
    }
    while (__builtin_amdgcn_ballot_w64(tail != head) != 0) verify_round();

    if (!active) return;
    // replay the reference's fold over the kept events in whitelist order (insertion sort; empty slots sort last)
    for (int a = 1; a < kSlots; ++a) {
        const uint32_t v = events[a][tid];
        int b = a - 1;
        while (b >= 0 && events[b][tid] > v) { events[b + 1][tid] = events[b][tid]; --b; }
        events[b + 1][tid] = v;
    }
    int best = K, hit = kHitNone, hit_end = -1;
    bool done = false;
    for (int a = 0; a < kSlots; ++a) {
        const uint32_t v = events[a][tid];
        if (v == 0xFFFFFFFFu) break;
        fold_update((int)((v >> 5) & 3u), (int)(v & 31u), (int)(v >> 7), best, hit, hit_end, done);
    }
    if (full_text && hit >= 0) {                                  // the winner's end index (its ed is `best`)
        const uint32_t p = planes32[hit];
        int ed_w;
        exact_pair<K>(p & 0xFFFFu, p >> 16, XH, XL, len, tie, ed_w, hit_end);
    }
    hit_out[r] = hit >= 0 ? hit : -1;
    ed_out[r] = (int8_t)best;
    end_out[r] = (int8_t)(hit >= 0 ? hit_end : -1);
    flags_out[r] = (uint8_t)(hit == kHitAmbiguous ? kFlagAmbig : 0u);
}

// ------------------------------------------------------------------------------------------
// Pair-seeded scan (the product path for max_ed == 2): "2 untouched pieces of 4" + packed verify.
//
// Lemma: cut the 16-nt barcode P into FOUR 4-nt pieces.  An edit operation touches at most one piece
// (an insertion between two pieces touches none), so an alignment of P inside the text T with <= 2
// operations leaves at least TWO pieces a < b untouched, both verbatim in T: piece a at column
// ja = s + 4a + d_a and piece b at jb = ja + 4(b-a) + delta, where s >= 0 is the start of the matched
// substring, d_a the net indel shift in front of piece a (|d_a| <= 2) and delta the net shift between
// the two pieces (|delta| <= 2; between ADJACENT pieces only insertions fit, delta >= 0).  The end of the
// alignment must fit a text of <= 20 columns, which bounds ja - 4a to [-2, 6] (same derivation as the
// single-piece windows above).  So the barcodes whose pieces (a, b) equal the text windows (ja, jb) for
// one of the 6 pairs x 9 x 5 (minus the statically impossible ones: 162 probes) are a SUPERSET of
// {w : sub_ed(w, T) <= 2}; a barcode outside it can never enter the reference's fold (:73-85).
//
// The key of a probe is 16 bits (two 4-nt codes), so per pair a 65536-bit occupancy bitmap (8 KB; all
// six = 48 KB) lives in LDS and ~92 % of the probes of a 5k-cell list end there; the rest look up a CSR
// table (tab2 / members2) in L2.  A read meets ~18 candidates (5k cells) instead of ~88 with the three
// single-piece seeds, and ~1/330 of a 6.8M list instead of 1/38.
//
// Two forms below: candidate lists + verify as two kernels (small whitelists), and k_pair_stream, which verifies
// straight out of the buckets (any whitelist size).  Events and the replay of the ordered fold are those of k_scan_seeded.
constexpr int kPairCount = 6;
// Is the probe (pieces a < b, window ja = 4a - 2 + xi, jb = ja + 4(b - a) + di - 2) admissible?  With x = xi - 2 the
// shift of piece a and delta = di - 2 the shift between the pieces, an alignment needs u >= max(0, -x) operations in
// front of piece a (the match cannot start before column 0), v >= |delta| between the pieces (adjacent pieces: only
// insertions, delta >= 0) and w >= x + delta - 4 behind piece b (the match must end inside <= 20 columns), and
// u + v + w <= 2.  131 of the 6 x 9 x 5 combinations remain (162 with the window bounds alone).
constexpr bool pair_probe_ok(int a, int b, int xi, int di) {
    const int ja = 4 * a - 2 + xi, jb = ja + 4 * (b - a) + di - 2;
    if (ja < 0 || ja + 4 > kTextCols || jb < 0 || jb + 4 > kTextCols) return false;
    const int x = xi - 2, d = di - 2;
    if (b == a + 1 && d < 0) return false;
    const int u = x < 0 ? -x : 0, v = d < 0 ? -d : d, w = x + d - 4 > 0 ? x + d - 4 : 0;
    return u + v + w <= 2;
}
constexpr int kPairBmWords = 2048;                                // 65536 bits per pair
constexpr int kPairTab = 65537;

// ------------------------------------------------------------------------------------------
// The pair-seeded scan as two kernels (product path for max_ed == 2 while the candidate lists fit,
// i.e. whitelists up to ~45k barcodes; above that k_pair_stream verifies straight out of the buckets):
//
//   k_pair_candidates : one wave per (64 rows of the work list, pair).  Needs the text's 2-bit codes only (2 VGPRs,
//       no bit-plane masks).  Probes its pair's LDS bitmap, turns occupied buckets into barcode indices through `tab8`
//       (entry = {size | offset << 8, first member}: 96 % of the occupied buckets of a 5k list hold one barcode and
//       need no second gather) and lists them in LDS (s-th candidate of a row, s handed out by a per-row LDS counter
//       shared by the row's six waves); the lists go to cand[slot / 64][s][slot % 64] (tiles of 64 slots x (cap + 1)
//       rows) 256 contiguous bytes per store, and k_pair_verify reads them the same way.
//   k_pair_verify<K>  : lane = row.  Two candidates per round through the packed K-error automaton; of the accepting
//       candidates only the two smallest indices per distance class are kept (all the ordered fold can consume, see
//       k_pair_stream) and get their exact (ed, end) in at most six rounds at the end (a true match is met through 3-6
//       pairs, and the divergent exact path used to run in almost every round of a wave); then the events are sorted
//       by whitelist index and the reference's order-dependent fold is replayed.  Rows with more candidates than `cap`
//       go to redo_list (k_scan_filtered finishes them in whitelist order).
constexpr uint32_t kCandOverflow = 0xFFFFFFFFu;
constexpr int kCandPop = 8;                                       // bucket look-ups a lane keeps in flight per trip

// Candidate generation with one WAVE per (64 rows, pair): a workgroup of 12 waves = 2 x 64 rows x 6 pairs shares
// the six bitmaps in LDS, every wave runs ONE pair's straight-line code (wave-uniform switch), and a row's candidate
// slots are handed out by an LDS counter (the order of a row's candidates does not matter: k_pair_verify keeps the two
// smallest accepted indices per class).  With lane = row the kernel was one round of 4.5 waves per SIMD whose length was
// the six pairs back to back (51 us per 293K rows; half the rows took as long); this form has six times as many waves,
// each a sixth as long, six resident per SIMD: 43 us.  Persistent workgroups: the 48 KB of bitmaps are loaded once per
// workgroup.
constexpr int kCandWaves = 12;
constexpr int kCandWThreads = kCandWaves * 64;
constexpr int kCandRowsPerBlock = (kCandWaves / kPairCount) * 64;
constexpr int kCandSeen = 16;                                     // memo entries per row (power of two)

__global__ __launch_bounds__(kCandWThreads) void k_pair_candidates(
    const uint32_t *__restrict__ bitmap, const uint2 *__restrict__ tab8, const uint2 *__restrict__ members8, int n_wl,
    const uint2 *__restrict__ list_tc, const int32_t *__restrict__ n_list_ptr,
    uint32_t *__restrict__ cand, int cap, uint32_t *__restrict__ cand_cnt,
    const uint32_t *__restrict__ red_partial, int red_n_wl, unsigned long long *__restrict__ red_counts,
    int32_t *red_cur, int32_t *red_host) {
    // the deferred histogram of the slot's previous run (see kPiggyBins): red_blocks workgroups in front of the grid add
    // its private histograms -- k_classify_packed of THIS run took them -- into counts[] and finish that run
    const int red_blocks = red_partial ? (red_n_wl + kCandWThreads - 1) / kCandWThreads : 0;
    if ((int)blockIdx.x < red_blocks) {                           // workgroup-uniform
        finish_run(blockIdx.x == 0, red_cur, red_host);
        hist_reduce_bin((int)blockIdx.x * kCandWThreads + (int)threadIdx.x, red_partial, red_n_wl, red_counts);
        return;
    }
    const int bid = (int)blockIdx.x - red_blocks, nblk = (int)gridDim.x - red_blocks;
    __shared__ uint32_t bm[kPairCount * kPairBmWords];
    __shared__ uint32_t row_cnt[2][kCandRowsPerBlock];            // by trip parity: a trip's counters are cleared while the trip before it writes out
    __shared__ uint32_t seen[kCandSeen][kCandRowsPerBlock];       // per row: the last index listed per hash class (bank = row), tagged with the trip
    extern __shared__ uint16_t lst[];                             // [cap + 1][128]: the trip's candidate lists (host: n_wl < 65536), row `cap` = dump
    const int n_list = *n_list_ptr;
    const int tid = threadIdx.x;
    if (bid * kCandRowsPerBlock >= n_list) return;    // workgroup-uniform
    for (int i = tid; i < kPairCount * kPairBmWords / 4; i += kCandWThreads) ((uint4 *)bm)[i] = ((const uint4 *)bitmap)[i];
    if (tid < 2 * kCandRowsPerBlock) (&row_cnt[0][0])[tid] = 0;
    for (int i = tid; i < kCandSeen * kCandRowsPerBlock; i += kCandWThreads) (&seen[0][0])[i] = 0xFFFFFFFFu;
    __syncthreads();                                              // bitmaps loaded, counters and memo cleared
    const int wave = tid >> 6, lane = tid & 63;
    const int p_mine = wave % kPairCount, rg = wave / kPairCount; // this wave's pair and which 64 rows of the workgroup's 128
    uint32_t trip = 0;
    // the text's codes come from k_classify_packed (list_tc[slot], one coalesced 8-byte load instead of list -> row);
    // the next trip's are requested a trip ahead
    uint2 t_next = list_tc[min(bid * kCandRowsPerBlock + rg * 64 + lane, n_list - 1)];
    for (int g = bid; g * kCandRowsPerBlock < n_list; g += nblk, ++trip) {      // workgroup-uniform trip count
        // The memo is never cleared between trips: an entry carries the trip that wrote it in its upper half (indices
        // are below 65536), so what an earlier trip left cannot match.  (Cleared again before the tag could repeat.)
        if ((trip & 0x7FFFu) == 0x7FFFu) {                        // workgroup-uniform, every 32K trips of a workgroup
            __syncthreads();
            for (int i = tid; i < kCandSeen * kCandRowsPerBlock; i += kCandWThreads) (&seen[0][0])[i] = 0xFFFFFFFFu;
            __syncthreads();
        }
        const uint32_t tag = (trip & 0x7FFFu) << 16;
        uint32_t *const my_cnt = &row_cnt[trip & 1u][rg * 64 + lane];
        const int slot = g * kCandRowsPerBlock + rg * 64 + lane;
        const bool active = slot < n_list;
        const uint32_t tlo = t_next.x, thi = t_next.y & 0xFFu;    // 2-bit codes of the text, column j at bits 2j; 'A' padding past
        const uint64_t tc = (uint64_t)tlo | ((uint64_t)thi << 32);   // its end, which can only add candidates
        t_next = list_tc[min(slot + nblk * kCandRowsPerBlock, n_list - 1)];
        uint16_t *const out = lst + rg * 64 + lane;               // the row's column of the trip's lists in LDS

        // one bucket entry -> candidate slots (e.x = bucket size (saturating) | offset << 8, e.y = first member)
        // A true barcode is met through 3-6 of the pairs, i.e. by several of this row's waves: `seen` is a direct-mapped
        // memo of the indices already listed for the row (exact compare, so it can only drop real repeats; a repeat that
        // slips through -- two waves listing the same index at the same moment, or an evicted entry -- is harmless:
        // k_pair_verify ignores an index it already keeps).  ~14 -> ~10 candidates per row of a 5k-cell list.
        auto emit = [&](uint32_t idx) {
            uint32_t *const sp = &seen[idx & (kCandSeen - 1)][rg * 64 + lane];
            if (atomicExch(sp, tag | idx) == (tag | idx)) return;  // one LDS exchange: of two waves listing an index at the same moment one sees the other's
            const uint32_t base = atomicAdd(my_cnt, 1u);
            out[min(base, (uint32_t)cap) * kCandRowsPerBlock] = (uint16_t)idx;  // row `cap` is a dump row
        };
        auto consume = [&](uint2 e, bool h, const uint2 *mp) {
            const uint32_t n_mem = h ? (e.x & 0xFFu) : 0u;
            if (n_mem) {
                if (n_mem == 0xFFu) { atomicAdd(my_cnt, 0x40000000u); return; }   // saturated size: let the ordered scan do this row
                emit(e.y);
                for (uint32_t k = 1; k < n_mem; ++k) emit(mp[(e.x >> 8) + k].x);  // rare for small whitelists
            }
        };
        auto do_pair = [&](auto pa_c, auto pb_c, auto p_c) {
            constexpr int a = decltype(pa_c)::value, b = decltype(pb_c)::value, p = decltype(p_c)::value;
            uint32_t mlo = 0, mhi = 0;                            // bit (5 xi + di) <=> windows (ja, jb) name an occupied bucket
#pragma unroll
            for (int xi = 0; xi < 9; ++xi) {
#pragma unroll
                for (int di = 0; di < 5; ++di) {
                    const int ja = 4 * a - 2 + xi, jb = ja + 4 * (b - a) + di - 2;
                    if (pair_probe_ok(a, b, xi, di)) {
                        const uint32_t ca = 2 * ja + 8 <= 32 ? (tlo >> (2 * ja)) : 2 * ja >= 32 ? (thi >> ((2 * ja) & 31)) : __builtin_amdgcn_alignbit(thi, tlo, (2 * ja) & 31);
                        const uint32_t cb = 2 * jb + 8 <= 32 ? (tlo >> (2 * jb)) : 2 * jb >= 32 ? (thi >> ((2 * jb) & 31)) : __builtin_amdgcn_alignbit(thi, tlo, (2 * jb) & 31);
                        const uint32_t key = ((ca & 0xFFu) << 8) | (cb & 0xFFu);
                        const uint32_t w = bm[p * kPairBmWords + (key >> 5)];
                        const uint32_t bit = (w >> (key & 31u)) & 1u;
                        const int pos = 5 * xi + di;
                        if (pos < 32) mlo |= bit << (pos & 31); else mhi |= bit << ((pos - 32) & 31);
                    }
                }
            }
            if (!active) { mlo = 0; mhi = 0; }
            const uint2 *tp = tab8 + (size_t)p * 65536;
            const uint2 *mp = members8 + (size_t)p * n_wl;
            auto pop = [&](bool &h) -> uint32_t {                 // next hit of this lane -> its 16-bit key
                const bool lo = mlo != 0;
                const uint32_t w = lo ? mlo : mhi;
                h = w != 0;
                const int pos = (int)__builtin_ctz(w | 0x80000000u) + (lo ? 0 : 32);
                const uint32_t w2 = w & (w - 1u);
                mlo = lo ? w2 : mlo;
                mhi = lo ? mhi : w2;
                const int xi = (pos * 13) >> 6, di = pos - 5 * xi;    // pos / 5, pos % 5 for pos < 64
                const int ja = 4 * a - 2 + xi, jb = ja + 4 * (b - a) + di - 2;
                return (((uint32_t)(tc >> ((2 * ja) & 63)) & 0xFFu) << 8) | ((uint32_t)(tc >> ((2 * jb) & 63)) & 0xFFu);
            };
            while (__builtin_amdgcn_ballot_w64((mlo | mhi) != 0) != 0) {             // usually one trip: <= kCandPop hits per lane
                uint2 e[kCandPop];
                bool h[kCandPop];
#pragma unroll
                for (int q = 0; q < kCandPop; ++q) {              // all look-ups of the trip in flight together
                    const uint32_t key = pop(h[q]);
                    e[q] = tp[h[q] ? key : 0u];
                }
#pragma unroll
                for (int q = 0; q < kCandPop; ++q) consume(e[q], h[q], mp);
            }
        };
#define BZ_PAIR(A, B, P) do_pair(std::integral_constant<int, A>{}, std::integral_constant<int, B>{}, std::integral_constant<int, P>{})
        switch (p_mine) {                                         // wave-uniform
            case 0: BZ_PAIR(0, 1, 0); break;
            case 1: BZ_PAIR(0, 2, 1); break;
            case 2: BZ_PAIR(0, 3, 2); break;
            case 3: BZ_PAIR(1, 2, 3); break;
            case 4: BZ_PAIR(1, 3, 4); break;
            default: BZ_PAIR(2, 3, 5); break;
        }
#undef BZ_PAIR
        __syncthreads();                                          // the six waves of a row group are done with its lists
        // Write-out: the lists go to the tile row by row, 64 slots x 4 bytes = 256 contiguous bytes per store (written
        // straight from the lanes that found them, every store instruction touched up to 64 different lines).
        // Wave w takes rows w / 2, w / 2 + 6, ... of tile w % 2.
        {
            const int t = wave & 1, row = t * 64 + lane, wslot = g * kCandRowsPerBlock + row;
            const uint32_t c = row_cnt[trip & 1u][row];
            if (tid < kCandRowsPerBlock) row_cnt[(trip & 1u) ^ 1u][tid] = 0;   // the next trip's counters (last read before this trip's first barrier)
            const uint32_t c_ok = (wslot < n_list && c <= (uint32_t)cap) ? c : 0u;
            uint32_t *const tile = cand + (size_t)(wslot >> 6) * ((size_t)(cap + 1) * 64) + lane;
            for (uint32_t q = (uint32_t)(wave >> 1); __builtin_amdgcn_ballot_w64(q < c_ok) != 0; q += kCandWaves / 2)
                if (q < c_ok) tile[q << 6] = lst[q * kCandRowsPerBlock + row];
            if (wave < 2 && wslot < n_list) cand_cnt[wslot] = c > (uint32_t)cap ? kCandOverflow : c;
        }
        __syncthreads();                                          // ... lists and counters read before the next trip reuses them
    }
}

constexpr int kVerifyThreads = 256;                               // measured per 293K rows: 128 lanes 37.4 us, 256 35.0, 512 38.6, 1024 50.1
// (.., 5): five waves per SIMD (91 VGPRs; left alone the compiler takes 97 = four).  A 1M-row step of BASELINE's workload
// puts 4.5 workgroups on every CU: with four resident the last ones wait for a whole wave-lifetime.
template <int K>
__global__ __launch_bounds__(kVerifyThreads, 5) void k_pair_verify(
    const uint32_t *__restrict__ cand, int cap, const uint32_t *__restrict__ cand_cnt,
    const uint32_t *__restrict__ planes32, const uint2 *__restrict__ list_tc,
    const int32_t *__restrict__ list, const int32_t *__restrict__ n_list_ptr, int tie,
    int32_t *__restrict__ redo_list, int32_t *__restrict__ n_redo,
    int32_t *__restrict__ hit_out, int8_t *__restrict__ ed_out, int8_t *__restrict__ end_out, uint8_t *__restrict__ flags_out) {
    __shared__ uint32_t s_own[6][kVerifyThreads];                 // codes (2) + first four candidates of slot tid (see below)
    __shared__ uint32_t s_cnt[kVerifyThreads], s_bin[64];
    __shared__ uint16_t s_perm[kVerifyThreads];
    const int n_list = *n_list_ptr;
    const int tid = threadIdx.x;
    if ((int)(blockIdx.x * blockDim.x) >= n_list) return;         // workgroup-uniform
    // A wave runs as many rounds as its fullest lane: the workgroup's 256 slots are dealt to the lanes in order of
    // their round counts (counting sort in LDS), so each wave holds one quartile; the quartile -> wave map rotates with
    // the workgroup so that no SIMD collects the long waves.
    // What a lane needs of its slot at once (codes, the first two rounds' candidates) is requested for slot TID together
    // with the count, before the sort knows which slot the lane will take, and changes hands through LDS afterwards:
    // the loads that used to wait for the permutation (one more memory round trip in a kernel that is about one
    // wave-lifetime long) are in flight while the sort runs.
    int slot, mine;
    uint32_t cnt_raw;
    {
        const int s0 = blockIdx.x * blockDim.x + tid;
        const uint32_t c = s0 < n_list ? cand_cnt[s0] : 0u;
        const int s0c = min(s0, n_list - 1);
        const uint2 t_own = list_tc[s0c];
        const uint32_t *cp_own = cand + (size_t)(s0c >> 6) * ((size_t)(cap + 1) * 64) + (s0c & 63);
        uint32_t x_own[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x_own[q] = cp_own[min(q, cap) << 6];   // rows 0..cap of the tile exist (row `cap` = dump)
        s_cnt[tid] = c;
        if (tid < 64) s_bin[tid] = 0;
        __syncthreads();
        const uint32_t rounds = c == kCandOverflow ? 0u : min((c + 1u) >> 1, 63u);
        const uint32_t pos = atomicAdd(&s_bin[rounds], 1u);
        __syncthreads();
        if (tid < 64) {                                           // exclusive prefix over the 64 bins (one wave)
            const uint32_t v = s_bin[tid];
            uint32_t x = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d, 64); if (tid >= d) x += y; }
            s_bin[tid] = x - v;
        }
        __syncthreads();
        s_perm[s_bin[rounds] + pos] = (uint16_t)tid;
        s_own[0][tid] = t_own.x; s_own[1][tid] = t_own.y;
#pragma unroll
        for (int q = 0; q < 4; ++q) s_own[2 + q][tid] = x_own[q];
        __syncthreads();
        mine = s_perm[(((tid >> 6) + blockIdx.x) & (kVerifyThreads / 64 - 1)) * 64 + (tid & 63)];
        slot = blockIdx.x * blockDim.x + mine;
        cnt_raw = s_cnt[mine];
    }
    const bool active = slot < n_list;
    const int r = list[active ? slot : n_list - 1];              // only the result stores need it
    const uint2 t = make_uint2(s_own[0][mine], s_own[1][mine]);  // the text's codes + length, from k_classify_packed
    const uint32_t x0 = s_own[2][mine], x1 = s_own[3][mine], x2 = s_own[4][mine], x3 = s_own[5][mine];
    const int len = (int)((t.y >> 8) & 0xFFu);
    bool overflow = cnt_raw == kCandOverflow;
    const uint32_t cnt = overflow ? 0u : cnt_raw;

    uint32_t XH[kTextCols], XL[kTextCols];
    {
        const uint32_t nlo = ~t.x, nhi = ~t.y;                    // a sign-extended 1-bit field of the inverted codes = the mask
#pragma unroll
        for (int j = 0; j < kTextCols; ++j) {
            const uint32_t w = j < 16 ? nlo : nhi;
            const int b = 2 * (j & 15);
            XH[j] = (uint32_t)__builtin_amdgcn_sbfe((int)w, b + 1, 1);
            XL[j] = (uint32_t)__builtin_amdgcn_sbfe((int)w, b, 1);
        }
    }
    const uint32_t one = in_vgpr(kOne2);
    constexpr int kFirst = kBcLen - K - 1;
    uint32_t keep[3][2];                                          // two smallest accepted indices per distance class (see k_pair_stream)
#pragma unroll
    for (int c = 0; c < 3; ++c) { keep[c][0] = 0xFFFFFFFFu; keep[c][1] = 0xFFFFFFFFu; }
    const bool full_text = len == kTextCols;
    const uint32_t *cp = cand + (size_t)(slot >> 6) * ((size_t)(cap + 1) * 64) + (slot & 63);

    // candidates of round k: slots 2k and 2k+1 (an odd tail repeats the first).  Two-deep pipeline: the candidate
    // indices of round k+2 and the bit planes of round k+1 are requested before round k's automaton runs, so no
    // round waits on a load it issued itself.
    auto load_idx = [&](uint32_t k, uint32_t &a0, uint32_t &a1) {     // unconditional loads (clamped slot): no wait, no branch
        const bool v0 = 2 * k < cnt, v1 = 2 * k + 1 < cnt;
        const uint32_t x0 = cp[(v0 ? 2 * k : 0u) << 6];
        const uint32_t x1 = cp[(v1 ? 2 * k + 1 : 0u) << 6];
        a0 = v0 ? x0 : 0u;
        a1 = v1 ? x1 : a0;
    };
    uint32_t i0, i1, n0, n1;
    {                                                             // rounds 0 and 1: candidates 0..3, handed over through LDS
        i0 = 0 < cnt ? x0 : 0u;
        i1 = 1 < cnt ? x1 : i0;
        n0 = 2 < cnt ? x2 : 0u;
        n1 = 3 < cnt ? x3 : n0;
    }
    uint32_t p0 = planes32[i0], p1 = planes32[i1];
    for (uint32_t k = 0; __builtin_amdgcn_ballot_w64(2 * k < cnt) != 0; ++k) {
        const uint32_t c0 = i0, c1 = i1, q0 = p0, q1 = p1;
        i0 = n0; i1 = n1;
        p0 = planes32[i0]; p1 = planes32[i1];                     // round k+1's planes
        load_idx(k + 2, n0, n1);                                  // round k+2's indices
        const uint32_t bh = (q0 & 0xFFFFu) | (q1 << 16), bl = (q0 >> 16) | (q1 & 0xFFFF0000u);
        PackedScan<K> A;
        A.init();
        uint32_t H[K + 1], P[K + 1];                              // accept bits per level, two columns per op
#pragma unroll
        for (int d = 0; d <= K; ++d) { H[d] = 0; P[d] = 0; }
#pragma unroll
        for (int j = 0; j < kTextCols; ++j) {
            const uint32_t eq = BZ_LUT3(bh ^ XH[j], bl, XL[j], kTA & (kTB ^ kTC));
            if (j == kTextCols - 1) A.template step<true>(eq, one);
            else A.template step<false>(eq, one);
            if (j >= kFirst) {
#pragma unroll
                for (int d = 0; d <= K; ++d) {
                    if (((j - kFirst) & 1) == 0) P[d] = A.R[d];
                    else H[d] = BZ_LUT3(H[d], P[d], A.R[d], kTA | kTB | kTC);
                }
            }
        }
        if (((kTextCols - 1 - kFirst) & 1) == 0) {
#pragma unroll
            for (int d = 0; d <= K; ++d) H[d] |= P[d];
        }
        uint32_t acc = ((H[K] >> 15) & 1u) | ((H[K] >> 30) & 2u);
        if (2 * k >= cnt) acc = 0;
        if (2 * k + 1 >= cnt) acc &= 1u;
        while (acc) {                                             // keep the two smallest indices of the candidate's class
            const int b = __ffs(acc) - 1;
            acc &= acc - 1;
            const uint32_t idx = b ? c1 : c0;
            const uint32_t sh = b ? 31u : 15u;
            int cls = ((H[0] >> sh) & 1u) ? 0 : (K >= 1 && ((H[K >= 1 ? 1 : 0] >> sh) & 1u)) ? 1 : 2;
            if (!full_text) cls = 2;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (c != cls) continue;
                if (idx == keep[c][0] || idx == keep[c][1]) continue;
                if (!full_text && keep[c][1] != 0xFFFFFFFFu) overflow = true;   // a third index in the short-text pool
                if (idx < keep[c][0]) { keep[c][1] = keep[c][0]; keep[c][0] = idx; }
                else if (idx < keep[c][1]) keep[c][1] = idx;
            }
        }
    }
    // Events of the kept barcodes.  On a 20-column text the class IS the edit distance (the lowest level that accepts in
    // any column = the minimum over all end positions), and the fold needs the end index of the WINNER only: the kept
    // indices become events as they are and one exact pass after the fold gives the winner's end.  (Every kept barcode
    // used to get its exact (ed, end) first: as many wave-wide passes as the fullest lane kept, ~2.5 per wave.)  Shorter
    // texts keep that path: their pool has no classes, only slots 4 and 5 (class 2) are in use.
    // The at most six events (index << 7 | ed << 5 | end; 0xFFFFFFFF = none) stay in registers: a fixed sorting network
    // and an unrolled fold instead of an insertion sort over per-lane LDS columns.
    uint32_t ev[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint32_t idx = keep[c][q];
            ev[2 * c + q] = idx == 0xFFFFFFFFu ? 0xFFFFFFFFu : full_text ? ((idx << 7) | ((uint32_t)c << 5)) : idx;
        }
    }
    if (__builtin_amdgcn_ballot_w64(!full_text && active && !overflow) != 0) {
#pragma unroll 1
        for (int a = 4; a < 6; ++a) {
            const uint32_t idx = a == 4 ? ev[4] : ev[5];
            const bool want = !full_text && idx != 0xFFFFFFFFu;
            if (__builtin_amdgcn_ballot_w64(want) == 0) continue;
            const uint32_t p = planes32[want ? idx : 0u];
            int ed, end;
            exact_pair<K>(p & 0xFFFFu, p >> 16, XH, XL, len, tie, ed, end);
            const uint32_t v = (want && ed <= K) ? ((idx << 7) | ((uint32_t)ed << 5) | (uint32_t)end) : 0xFFFFFFFFu;
            if (!full_text) { if (a == 4) ev[4] = v; else ev[5] = v; }
        }
    }
    // whitelist order: the 12-comparator network for six inputs (none sorts last)
    auto cx = [&](uint32_t &a, uint32_t &b) { const uint32_t lo = min(a, b), hi = max(a, b); a = lo; b = hi; };
    cx(ev[0], ev[5]); cx(ev[1], ev[3]); cx(ev[2], ev[4]);
    cx(ev[1], ev[2]); cx(ev[3], ev[4]);
    cx(ev[0], ev[3]); cx(ev[2], ev[5]);
    cx(ev[0], ev[1]); cx(ev[2], ev[3]); cx(ev[4], ev[5]);
    cx(ev[1], ev[2]); cx(ev[3], ev[4]);
    int best = K, hit = kHitNone, hit_end = -1;
    bool done = false;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const uint32_t v = ev[a];
        if (v != 0xFFFFFFFFu) fold_update((int)((v >> 5) & 3u), (int)(v & 31u), (int)(v >> 7), best, hit, hit_end, done);
    }
    if (__builtin_amdgcn_ballot_w64(full_text && hit >= 0 && active && !overflow) != 0) {     // the winner's end index (its ed is `best`)
        const bool want = full_text && hit >= 0;
        const uint32_t p = planes32[want ? hit : 0];
        int ed_w, end_w;
        exact_pair<K>(p & 0xFFFFu, p >> 16, XH, XL, len, tie, ed_w, end_w);
        if (want) hit_end = end_w;
    }
    if (!active) return;
    if (overflow) {                                               // too many candidates / near matches: finish in whitelist order
        redo_list[atomicAdd(n_redo, 1)] = r;
        return;
    }
    hit_out[r] = hit >= 0 ? hit : -1;
    ed_out[r] = (int8_t)best;
    end_out[r] = (int8_t)(hit >= 0 ? hit_end : -1);
    flags_out[r] = (uint8_t)(hit == kHitAmbiguous ? kFlagAmbig : 0u);
}

// ------------------------------------------------------------------------------------------
// k_pair_stream: the pair-seeded scan for whitelists of ANY size (BASELINE's stress case: the full 10x list, 6.8M
// barcodes, ~104 per bucket, ~13,600 candidates per row -- far too many to write down as lists).
// lane = row.  Each lane walks its 131 admissible probes; a probe names a bucket = a contiguous run of
// members8[] = {barcode index, planes32}, and the lane verifies the run two barcodes per round through the packed
// automaton: one 8-byte gather per candidate, bit planes included, consecutive candidates of a bucket in the same
// cache lines.  Pipeline per round: this round's two members were requested last round, the next bucket's bounds
// two probes ahead.  Of the accepted candidates only the two smallest indices per distance class are kept (that is all
// the ordered fold can consume, see below); they get their exact (ed, end) at the end, then the fold is replayed.
constexpr int kStreamProbes = 136;                                // >= the 131 admissible probes

struct PairProbeTable {                                           // entry = p | ja << 3 | jb << 8, flattened at compile time
    uint32_t e[kStreamProbes];
    int n;
    constexpr PairProbeTable() : e(), n(0) {
        const int pa[kPairCount] = {0, 0, 0, 1, 1, 2}, pb[kPairCount] = {1, 2, 3, 2, 3, 3};
        for (int p = 0; p < kPairCount; ++p)
            for (int xi = 0; xi < 9; ++xi)
                for (int di = 0; di < 5; ++di)
                    if (pair_probe_ok(pa[p], pb[p], xi, di)) {
                        const int ja = 4 * pa[p] - 2 + xi, jb = ja + 4 * (pb[p] - pa[p]) + di - 2;
                        e[n++] = (uint32_t)p | ((uint32_t)ja << 3) | ((uint32_t)jb << 8);
                    }
        for (int i = n; i < kStreamProbes; ++i) e[i] = 0;
    }
};
__device__ const PairProbeTable kPairProbes{};
static_assert(PairProbeTable().n <= kStreamProbes, "probe table too small");

template <int K>
__global__ __launch_bounds__(256) void k_pair_stream(
    const int32_t *__restrict__ tab2, const uint2 *__restrict__ members8, const uint32_t *__restrict__ planes_of, int n_wl,
    const uint8_t *__restrict__ ext, int ext_stride, const uint8_t *__restrict__ ext_len,
    const int32_t *__restrict__ list, const int32_t *__restrict__ n_list_ptr, int tie,
    int32_t *__restrict__ redo_list, int32_t *__restrict__ n_redo,
    int32_t *__restrict__ hit_out, int8_t *__restrict__ ed_out, int8_t *__restrict__ end_out, uint8_t *__restrict__ flags_out) {
    static_assert(K == 2, "the pair lemma is stated for two edit operations");
    __shared__ uint32_t probes[kStreamProbes];
    __shared__ uint32_t events[6][256];
    const int n_list = *n_list_ptr;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x * 256 >= n_list) return;                  // workgroup-uniform
    if (tid < kStreamProbes) probes[tid] = kPairProbes.e[tid];
    __syncthreads();
    const int n_probes = kPairProbes.n;
    const int slot = blockIdx.x * 256 + tid;
    if ((slot & ~63) >= n_list) return;                           // whole wave idle (wave-uniform)
    const bool active = slot < n_list;
    const int r = list[active ? slot : n_list - 1];
    const int len = ext_len[r];

    uint32_t XH[kTextCols], XL[kTextCols];
    uint64_t tc = 0;                                              // 2-bit codes of the text, column j at bits 2j
    {
        const RowBytes rb = load_row_packed(ext + (size_t)r * ext_stride);
#pragma unroll
        for (int j = 0; j < kTextCols; ++j) {
            const uint32_t c = (j < len) ? rb.at(j) : (uint32_t)'A';   // padding can only add candidates
            const uint32_t code = base_code(c);
            XH[j] = (code & 2u) ? 0u : 0xFFFFFFFFu;
            XL[j] = (code & 1u) ? 0u : 0xFFFFFFFFu;
            tc |= (uint64_t)code << (2 * j);
        }
    }
    const uint32_t one = in_vgpr(kOne2);
    constexpr int kFirst = kBcLen - K - 1;
    // The fold (read_assignment.py:73-85) consumes at most TWO events of each distance class, and always the first two
    // in whitelist order (a class is ignored for good once best_ed has dropped below it).  So the two smallest accepted
    // indices per class decide the row, however many near matches a 6.8M-barcode list holds (~55 per row).  The class of
    // a candidate = the lowest automaton level that accepts, exact for 20-column texts; shorter texts (padding columns
    // can lower the level) keep every accepted index in one pool of two and hand the row on if a third shows up.
    uint32_t keep[3][2];
#pragma unroll
    for (int c = 0; c < 3; ++c) { keep[c][0] = 0xFFFFFFFFu; keep[c][1] = 0xFFFFFFFFu; }
    const bool full_text = len == kTextCols;
    bool overflow = false;

    // bucket of probe i as a [start, end) range of the flat members8 array (empty past the last probe / idle lanes)
    auto bucket = [&](int i, uint32_t &bs, uint32_t &be) {
        const bool ok = active && i < n_probes;
        const uint32_t pe = probes[ok ? i : 0];
        const uint32_t p = pe & 7u;
        const uint32_t key = (((uint32_t)(tc >> (2 * ((pe >> 3) & 31u))) & 0xFFu) << 8) | ((uint32_t)(tc >> (2 * ((pe >> 8) & 31u))) & 0xFFu);
        const int32_t *t = tab2 + (size_t)p * kPairTab + key;
        const uint32_t s0 = (uint32_t)t[0], s1 = (uint32_t)t[1];
        bs = ok ? p * (uint32_t)n_wl + s0 : 0u;
        be = ok ? p * (uint32_t)n_wl + s1 : 0u;
    };
    uint32_t cur = 0, end = 0;                                    // rest of the current bucket
    uint32_t nbs, nbe;                                            // the NEXT bucket's bounds (probe pi), already requested
    int pi = 0;
    bucket(pi, nbs, nbe);
    // step to the next pair of members: inside the bucket, else into the next non-exhausted probe
    auto advance = [&]() {
        if (cur >= end) {                                         // (an empty bucket costs one idle round: rare on big lists)
            cur = nbs; end = nbe;
            ++pi;
            bucket(pi, nbs, nbe);
        }
    };
    advance();
    uint2 m0 = members8[cur < end ? cur : 0], m1 = members8[cur + 1 < end ? cur + 1 : (cur < end ? cur : 0)];
    bool v0 = cur < end, v1 = cur + 1 < end;
    for (;;) {
        const bool more = v0 || pi <= n_probes;                   // this lane still has members in hand or probes to open
        if (__builtin_amdgcn_ballot_w64(active && more) == 0) break;
        const uint2 c0 = m0, c1 = m1;
        const bool a0 = v0, a1 = v1;
        cur += 2;
        advance();
        m0 = members8[cur < end ? cur : 0];                       // next round's members fly under this round's automaton
        m1 = members8[cur + 1 < end ? cur + 1 : (cur < end ? cur : 0)];
        v0 = cur < end; v1 = cur + 1 < end;

        const uint32_t q0 = c0.y, q1 = a1 ? c1.y : c0.y;
        const uint32_t bh = (q0 & 0xFFFFu) | (q1 << 16), bl = (q0 >> 16) | (q1 & 0xFFFF0000u);
        PackedScan<K> A;
        A.init();
        uint32_t H[K + 1], P[K + 1];                              // accept bits per level, two columns per op
#pragma unroll
        for (int d = 0; d <= K; ++d) { H[d] = 0; P[d] = 0; }
#pragma unroll
        for (int j = 0; j < kTextCols; ++j) {
            const uint32_t eq = BZ_LUT3(bh ^ XH[j], bl, XL[j], kTA & (kTB ^ kTC));
            if (j == kTextCols - 1) A.template step<true>(eq, one);
            else A.template step<false>(eq, one);
            if (j >= kFirst) {
#pragma unroll
                for (int d = 0; d <= K; ++d) {
                    if (((j - kFirst) & 1) == 0) P[d] = A.R[d];
                    else H[d] = BZ_LUT3(H[d], P[d], A.R[d], kTA | kTB | kTC);
                }
            }
        }
        if (((kTextCols - 1 - kFirst) & 1) == 0) {
#pragma unroll
            for (int d = 0; d <= K; ++d) H[d] |= P[d];
        }
        uint32_t acc = ((H[K] >> 15) & 1u) | ((H[K] >> 30) & 2u);
        if (!a0) acc = 0;
        if (!a1) acc &= 1u;
        while (acc) {                                             // keep the two smallest indices of the candidate's class
            const int b = __ffs(acc) - 1;
            acc &= acc - 1;
            const uint32_t idx = b ? c1.x : c0.x;
            const uint32_t sh = b ? 31u : 15u;
            int cls = ((H[0] >> sh) & 1u) ? 0 : ((H[1] >> sh) & 1u) ? 1 : 2;
            if (!full_text) cls = 2;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (c != cls) continue;
                if (idx == keep[c][0] || idx == keep[c][1]) continue;
                if (!full_text && keep[c][1] != 0xFFFFFFFFu) overflow = true;   // a third index in the short-text pool
                if (idx < keep[c][0]) { keep[c][1] = keep[c][0]; keep[c][0] = idx; }
                else if (idx < keep[c][1]) keep[c][1] = idx;
            }
        }
    }
    // Events of the kept barcodes.  On a 20-column text the class IS the edit distance (the lowest level that accepts in
    // any column = the minimum over all end positions), and the fold needs the end index of the WINNER only: the kept
    // indices become events as they are and one exact pass after the fold gives the winner's end.  (Every kept barcode
    // used to get its exact (ed, end) first: as many wave-wide passes as the fullest lane kept, ~2.5 per wave.)  Shorter
    // texts keep that path: their pool has no classes.
    int nkept = 0, nacc = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (keep[c][t] == 0xFFFFFFFFu) continue;
            if (full_text) events[nacc++][tid] = (keep[c][t] << 7) | ((uint32_t)c << 5);
            else events[nkept++][tid] = keep[c][t];
        }
    }
    for (int a = 0; __builtin_amdgcn_ballot_w64(a < nkept) != 0; ++a) {
        if (a < nkept) {
            const uint32_t idx = events[a][tid];
            const uint32_t p = planes_of[idx];
            int ed, end_i;
            exact_pair<K>(p & 0xFFFFu, p >> 16, XH, XL, len, tie, ed, end_i);
            if (ed <= K) events[nacc++][tid] = (idx << 7) | ((uint32_t)ed << 5) | (uint32_t)end_i;   // nacc <= a: in place
        }
    }
    if (!active) return;
    if (overflow) {
        redo_list[atomicAdd(n_redo, 1)] = r;
        return;
    }
    for (int a = 1; a < nacc; ++a) {                              // events in whitelist order (insertion sort)
        const uint32_t v = events[a][tid];
        int b = a - 1;
        while (b >= 0 && events[b][tid] > v) { events[b + 1][tid] = events[b][tid]; --b; }
        events[b + 1][tid] = v;
    }
    int best = K, hit = kHitNone, hit_end = -1;
    bool done = false;
    for (int a = 0; a < nacc; ++a) {
        const uint32_t v = events[a][tid];
        fold_update((int)((v >> 5) & 3u), (int)(v & 31u), (int)(v >> 7), best, hit, hit_end, done);
    }
    if (full_text && hit >= 0) {                                  // the winner's end index (its ed is `best`)
        const uint32_t p = planes_of[hit];
        int ed_w;
        exact_pair<K>(p & 0xFFFFu, p >> 16, XH, XL, len, tie, ed_w, hit_end);
    }
    hit_out[r] = hit >= 0 ? hit : -1;
    ed_out[r] = (int8_t)best;
    end_out[r] = (int8_t)(hit >= 0 ? hit_end : -1);
    flags_out[r] = (uint8_t)(hit == kHitAmbiguous ? kFlagAmbig : 0u);
}

// ------------------------------------------------------------------------------------------
// Generic byte-wise semi-global DP (any alphabet, any lengths; pattern = the shorter operand,
// the first one when equal).  Used for irregular rows / whitelists, max_ed > kMaxK and the
// single-pair entry points.  D[0][j] = 0, D[i][0] = i, answer = min_j D[m][j], end = j-1.
constexpr int kGenericMax = 64;   // BZ_EXT_MAX

__device__ inline int dev_sub_edit_distance(const uint8_t *a, int la, const uint8_t *b, int lb, int max_ed, int tie, int &end_idx) {
    const uint8_t *p, *t;
    int m, n;
    if (la <= lb) { p = a; m = la; t = b; n = lb; } else { p = b; m = lb; t = a; n = la; }
    uint8_t col[kGenericMax + 1];
    for (int i = 0; i <= m; ++i) col[i] = (uint8_t)i;
    int best = m, best_end = -1;
    for (int j = 1; j <= n; ++j) {
        int diag = col[0];
        col[0] = 0;
        const uint8_t tc = t[j - 1];
        for (int i = 1; i <= m; ++i) {
            int left = col[i];
            int v = diag + (p[i - 1] != tc);
            v = min(v, (int)col[i - 1] + 1);
            v = min(v, left + 1);
            diag = left;
            col[i] = (uint8_t)v;
        }
        int s = col[m];
        if (s < best || (tie && s == best)) { best = s; best_end = j - 1; }
    }
    if (best > max_ed) { end_idx = -1; return max_ed + 1; }
    end_idx = best_end;
    return best;
}

// Global Levenshtein distance with cut-off; the shorter operand is the DP column (<= kGenericMax).
__device__ inline int dev_edit_distance(const uint8_t *a, int la, const uint8_t *b, int lb, int max_ed) {
    const uint8_t *p = a, *t = b;
    int m = la, n = lb;
    if (la > lb) { p = b; m = lb; t = a; n = la; }
    uint8_t col[kGenericMax + 1];
    for (int i = 0; i <= m; ++i) col[i] = (uint8_t)i;
    for (int j = 1; j <= n; ++j) {
        int diag = col[0];
        col[0] = (uint8_t)min(j, 255);
        const uint8_t tc = t[j - 1];
        for (int i = 1; i <= m; ++i) {
            int left = col[i];
            int v = diag + (p[i - 1] != tc);
            v = min(v, (int)col[i - 1] + 1);
            v = min(v, left + 1);
            diag = left;
            col[i] = (uint8_t)min(v, 255);
        }
    }
    int d = col[m];
    return d > max_ed ? max_ed + 1 : d;
}

__device__ __forceinline__ void scan_generic_rows(int first, int step, int n_list,
                               const uint8_t *__restrict__ wl_bytes, const int64_t *__restrict__ wl_off, int n_wl,
                               const uint8_t *__restrict__ ext, int ext_stride, const uint8_t *__restrict__ ext_len,
                               const uint8_t *__restrict__ bc_off, const uint8_t *__restrict__ bc_len,
                               const int32_t *__restrict__ list, int max_ed, int tie,
                               int shortcut, int32_t *__restrict__ hit_out, int8_t *__restrict__ ed_out,
                               int8_t *__restrict__ end_out, uint8_t *__restrict__ flags_out) {
    for (int slot = first; slot < n_list; slot += step) {         // small persistent grid
    const int r = list[slot];
    const uint8_t *row = ext + (size_t)r * ext_stride;
    const int len = ext_len[r], off = bc_off[r], bl = bc_len[r];

    bool exact = false;
    if (shortcut) {                                           // `row.putative_bc in whitelist` (:61)
        for (int k = 0; k < n_wl; ++k) {
            const int64_t o = wl_off[k];
            if ((int)(wl_off[k + 1] - o) != bl) continue;
            bool same = true;
            for (int i = 0; i < bl && same; ++i) same = wl_bytes[o + i] == row[off + i];
            if (same) {
                hit_out[r] = k; ed_out[r] = 0; end_out[r] = -1; flags_out[r] = (uint8_t)(kFlagExact | kFlagGeneric);
                exact = true;
                break;
            }
        }
    }
    if (exact) continue;
    int best = max_ed, hit = kHitNone, hit_end = -1;
    bool hit_empty = false;                                   // bc_hit is the empty string (falsy in Python)
    bool done = false;
    for (int k = 0; k < n_wl && !done; ++k) {
        const int64_t o = wl_off[k];
        const int lw = (int)(wl_off[k + 1] - o);
        int end;
        const int ed = dev_sub_edit_distance(wl_bytes + o, lw, row, len, best, tie, end);
        if (ed < best) { best = ed; hit = k; hit_end = end; hit_empty = (lw == 0); }
        else if (ed == best) {
            if (hit == kHitNone || hit_empty) { hit = k; hit_end = end; hit_empty = (lw == 0); }
            else { hit = kHitAmbiguous; hit_empty = false; best -= 1; if (best < 0) done = true; }
        }
    }
    const bool ok = hit >= 0 && !hit_empty;
    hit_out[r] = ok ? hit : -1;
    ed_out[r] = (int8_t)best;
    end_out[r] = (int8_t)(ok ? hit_end : -1);
    flags_out[r] = (uint8_t)(kFlagGeneric | (hit == kHitAmbiguous ? kFlagAmbig : 0u));
    }
}

__global__ void k_scan_generic(const uint8_t *__restrict__ wl_bytes, const int64_t *__restrict__ wl_off, int n_wl,
                               const uint8_t *__restrict__ ext, int ext_stride, const uint8_t *__restrict__ ext_len,
                               const uint8_t *__restrict__ bc_off, const uint8_t *__restrict__ bc_len,
                               const int32_t *__restrict__ list, const int32_t *__restrict__ n_list_ptr, int max_ed, int tie,
                               int shortcut, int32_t *__restrict__ hit_out, int8_t *__restrict__ ed_out,
                               int8_t *__restrict__ end_out, uint8_t *__restrict__ flags_out) {
    scan_generic_rows(blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, *n_list_ptr, wl_bytes, wl_off, n_wl, ext, ext_stride,
                      ext_len, bc_off, bc_len, list, max_ed, tie, shortcut, hit_out, ed_out, end_out, flags_out);
}

// ------------------------------------------------------------------------------------------
// Irregular ROWS (N or lower case somewhere in the extension, a putative barcode that is not 16 nt, an extension longer
// than 20; every row when max_ed is 4..8) against a whitelist that has the fast-path structures (all regular, or mixed).
// k_scan_generic above gives such a row ONE lane that walks the whole list with the byte-wise DP: ~0.24 ms per pair with
// the DP column in scratch memory, 1.2 s for a 5,000-entry list, and a step lasts as long as its slowest lane -- ten such
// rows in a batch of a million made the step 12,000 times longer.  Here a WORKGROUP takes the row: its lanes stride over
// the entries; a regular entry is scored bit-parallel in registers -- the text's per-column masks come from LDS, a
// column that is not ACGT matches nothing (exactly what the byte compare of the reference's library does) -- with the
// roles the replaced FFI takes: the barcode is the pattern unless the text is shorter than 16, then the text is the
// pattern and the barcode's 16 characters the columns.  Irregular entries (mixed lists) take the byte-wise DP,
// one per lane.  The exact-match shortcut (`row.putative_bc in whitelist`, :61) is a byte compare over the list, smallest
// index wins.  As everywhere, the two smallest events per distance class decide the fold: lanes keep two per class,
// 64-bit LDS atomicMin picks the workgroup's (word = index << 16 | ed << 8 | end + 1), one lane replays the fold.
// One irregular whitelist entry against a plain-ACGT text of <= 32 columns (tc: its 2-bit codes): bit-parallel through the
// entry's four match masks when the entry is the pattern (1..32 characters, not longer than the text); the byte-wise DP
// otherwise (the text then is the pattern).  See the mixed-whitelist kernels below.
template <int K>
__device__ __forceinline__ void irr_exact(const uint4 peq, int m, const uint8_t *__restrict__ entry, const uint8_t *__restrict__ row,
                                          uint64_t tc, int len, int tie, int &ed, int &end) {
    if (m >= 1 && m <= 32 && m <= len) {
        ShiftAnd<K, 1u> A;
        A.init();
        const uint32_t acc = 1u << (m - 1);
        int best_cnt = m <= K ? K + 1 - m : 0, best_end = -1;     // before any column: every character deleted, D[m][0] = m
        for (int j = 0; j < len; ++j) {
            const uint32_t code = (uint32_t)(tc >> (2 * j)) & 3u;
            const uint32_t eq = code == 0u ? peq.x : code == 1u ? peq.y : code == 2u ? peq.z : peq.w;
            A.step(eq);
            int cnt = 0;
#pragma unroll
            for (int d = 0; d <= K; ++d) cnt += (A.R[d] & acc) != 0u;
            if (cnt > best_cnt || (tie && cnt == best_cnt && cnt > 0)) { best_cnt = cnt; best_end = j; }
        }
        ed = K + 1 - best_cnt;
        end = ed <= K ? best_end : -1;
    } else {
        ed = dev_sub_edit_distance(entry, m, row, len, K, tie, end);
    }
}

struct GenericArgs {
    const uint8_t *wl_bytes; const int64_t *wl_off; int n_wl;            // the whole list in the caller's order (shortcut compare)
    const uint32_t *planes32; int n_fast; const int32_t *reg_to_orig;   // its regular entries (reg_to_orig null: all of it)
    const int2 *irr_meta; const uint8_t *irr_bytes; const int64_t *irr_off; const uint4 *irr_peq; int n_irr;
    const uint8_t *ext; int ext_stride; const uint8_t *ext_len, *bc_off, *bc_len;
    int tie, shortcut;
    int32_t *hit; int8_t *ed; int8_t *end; uint8_t *flags;
};
constexpr int kGenericMaxK = 8;                                   // generic_row_wg is instantiated for max_ed 0..8
struct GenericShared {
    unsigned long long mn[2][kGenericMaxK + 1];
    uint32_t xh[kGenericMax], xl[kGenericMax], v[kGenericMax], peq[4];
    int exact;
    int odd_columns;                                              // columns of the text that are not ACGT
    int empty;                                                    // smallest index of an empty entry (mixed lists), or INT_MAX
    uint8_t text[kGenericMax];
};
constexpr int kNoIndex = 0x7FFFFFFF;

// Which EMPTY entries can reach the fold.  Distance-0 events in whitelist order: a run of empties is one falsy hit (each
// replaces the one before it), the first non-empty entry n1 replaces that, and whatever distance-0 event comes next --
// empty or not -- makes the read ambiguous and ends the fold.  So beside the two smallest non-empty events per class only
// the first empty entry (if in front of n1) and the first empty entry behind n1 matter.
__device__ inline void relevant_empties(const int2 *__restrict__ irr_meta, int n_irr, int first_empty, int n1, int &e_a, int &e_b) {
    e_a = first_empty < n1 ? first_empty : kNoIndex;
    e_b = kNoIndex;
    if (first_empty != kNoIndex && n1 != kNoIndex)
        for (int k = 0; k < n_irr; ++k)
            if (irr_meta[k].x == 0 && irr_meta[k].y > n1 && irr_meta[k].y < e_b) e_b = irr_meta[k].y;
}

template <int K>
__device__ __forceinline__ void generic_row_wg(const GenericArgs &a, int r, GenericShared &sh) {
    const int tid = threadIdx.x;
    const int len = a.ext_len[r], off = a.bc_off[r], bl = a.bc_len[r];
    const uint8_t *row = a.ext + (size_t)r * a.ext_stride;
    if (tid < kGenericMax) {
        const uint32_t c = tid < len ? row[tid] : 0u;
        const uint32_t code = base_code(c);
        sh.text[tid] = (uint8_t)c;
        sh.xh[tid] = (code & 2u) ? 0u : 0xFFFFu;
        sh.xl[tid] = (code & 1u) ? 0u : 0xFFFFu;
        sh.v[tid] = (tid < len && is_acgt(c)) ? 0xFFFFu : 0u;
    }
    if (tid < 4) sh.peq[tid] = 0u;
    if (tid == 4) sh.exact = 0x7FFFFFFF;
    if (tid == 5) sh.empty = kNoIndex;
    if (tid == 6) sh.odd_columns = 0;
    if (tid < 2 * (kGenericMaxK + 1)) (&sh.mn[0][0])[tid] = ~0ull;
    __syncthreads();
    if (len < kBcLen && tid < len && sh.v[tid]) atomicOr(&sh.peq[base_code(sh.text[tid])], 1u << tid);   // the text as the pattern
    if (tid < len && !sh.v[tid]) atomicAdd(&sh.odd_columns, 1);
    if (a.shortcut) {
        for (int k = tid; k < a.n_wl; k += 256) {
            const int64_t o = a.wl_off[k];
            if ((int)(a.wl_off[k + 1] - o) != bl) continue;
            bool same = true;
            for (int i = 0; i < bl && same; ++i) same = a.wl_bytes[o + i] == sh.text[off + i];
            if (same) atomicMin(&sh.exact, k);
        }
    }
    __syncthreads();
    if (sh.exact != 0x7FFFFFFF) {                                 // workgroup-uniform
        if (tid == 0) { a.hit[r] = sh.exact; a.ed[r] = 0; a.end[r] = -1; a.flags[r] = (uint8_t)(kFlagExact | kFlagGeneric); }
        __syncthreads();
        return;
    }
    unsigned long long keep[K + 1][2];
#pragma unroll
    for (int c = 0; c <= K; ++c) { keep[c][0] = ~0ull; keep[c][1] = ~0ull; }
    auto note = [&](int idx, int ed, int end) {
        const unsigned long long ev = ((unsigned long long)(uint32_t)idx << 16) | ((unsigned long long)ed << 8) | (unsigned long long)(end + 1);
#pragma unroll
        for (int c = 0; c <= K; ++c) {
            if (c != ed) continue;
            if (ev < keep[c][0]) { keep[c][1] = keep[c][0]; keep[c][0] = ev; }
            else if (ev < keep[c][1]) keep[c][1] = ev;
        }
    };
    for (int sub = tid; sub < a.n_fast; sub += 256) {
        const uint32_t p = a.planes32[sub];
        const uint32_t bh = p & 0xFFFFu, blo = p >> 16;
        ShiftAnd<K, 1u> A;
        A.init();
        int best_cnt = 0, best_end = -1;
        if (len >= kBcLen) {                                      // the barcode is the pattern (first argument when the lengths are equal)
            for (int j = 0; j < len; ++j) {
                A.step((bh ^ sh.xh[j]) & (blo ^ sh.xl[j]) & sh.v[j]);
                int cnt = 0;
#pragma unroll
                for (int d = 0; d <= K; ++d) cnt += (A.R[d] >> (kBcLen - 1)) & 1u;
                if (cnt > best_cnt || (a.tie && cnt == best_cnt && cnt > 0)) { best_cnt = cnt; best_end = j; }
            }
        } else {                                                  // the text is shorter: it is the pattern, the barcode's characters the columns
            const uint32_t acc = 1u << (len - 1);
            best_cnt = len <= K ? K + 1 - len : 0;
            for (int j = 0; j < kBcLen; ++j) {
                A.step(sh.peq[(((bh >> j) & 1u) << 1) | ((blo >> j) & 1u)]);
                int cnt = 0;
#pragma unroll
                for (int d = 0; d <= K; ++d) cnt += (A.R[d] & acc) != 0u;
                if (cnt > best_cnt || (a.tie && cnt == best_cnt && cnt > 0)) { best_cnt = cnt; best_end = j; }
            }
        }
        const int ed = K + 1 - best_cnt;
        if (ed <= K) note(a.reg_to_orig ? a.reg_to_orig[sub] : sub, ed, best_end);
    }
    // irregular entries: bit-parallel through their four match masks when the text is plain ACGT of <= 32 columns (what a
    // list without regular entries meets on every ordinary row); the byte-wise DP otherwise
    const bool plain_text = sh.odd_columns == 0 && len >= 1 && len <= 32;
    uint64_t tc = 0;
    if (plain_text && a.n_irr > 0)
        for (int j = 0; j < len; ++j) tc |= (uint64_t)base_code(sh.text[j]) << (2 * j);
    for (int k = tid; k < a.n_irr; k += 256) {
        if (a.irr_meta[k].x == 0) { atomicMin(&sh.empty, a.irr_meta[k].y); continue; }   // '' : distance 0 from any text
        int ed, end;
        if (plain_text) irr_exact<K>(a.irr_peq[k], a.irr_meta[k].x, a.irr_bytes + a.irr_off[k], row, tc, len, a.tie, ed, end);
        else ed = dev_sub_edit_distance(a.irr_bytes + a.irr_off[k], a.irr_meta[k].x, row, len, K, a.tie, end);
        if (ed <= K) note(a.irr_meta[k].y, ed, end);
    }
#pragma unroll
    for (int c = 0; c <= K; ++c) if (keep[c][0] != ~0ull) atomicMin(&sh.mn[0][c], keep[c][0]);
    __syncthreads();
#pragma unroll
    for (int c = 0; c <= K; ++c) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (keep[c][t] != ~0ull && keep[c][t] != sh.mn[0][c]) atomicMin(&sh.mn[1][c], keep[c][t]);
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long ev[2 * (K + 1) + 2];
        int nev = 0;
#pragma unroll
        for (int c = 0; c <= K; ++c) {
            if (sh.mn[0][c] != ~0ull) ev[nev++] = sh.mn[0][c];
            if (sh.mn[1][c] != ~0ull) ev[nev++] = sh.mn[1][c];
        }
        // empty entries: events of distance 0 whose end field (255) marks them as falsy
        int e_a, e_b;
        relevant_empties(a.irr_meta, a.n_irr, sh.empty, sh.mn[0][0] != ~0ull ? (int)(sh.mn[0][0] >> 16) : kNoIndex, e_a, e_b);
        if (e_a != kNoIndex) ev[nev++] = ((unsigned long long)(uint32_t)e_a << 16) | 255ull;
        if (e_b != kNoIndex) ev[nev++] = ((unsigned long long)(uint32_t)e_b << 16) | 255ull;
        for (int x = 1; x < nev; ++x) {                           // whitelist order = order of the words (index in the top bits)
            const unsigned long long v = ev[x];
            int y = x - 1;
            while (y >= 0 && ev[y] > v) { ev[y + 1] = ev[y]; --y; }
            ev[y + 1] = v;
        }
        int best = K, hit = kHitNone, hit_end = -1;
        bool done = false, hit_empty = false;
        for (int x = 0; x < nev; ++x) {
            const int e = (int)(ev[x] & 0xFFu);
            fold_update_falsy((int)((ev[x] >> 8) & 0xFFu), e == 255 ? -1 : e - 1, (int)(ev[x] >> 16), e == 255, best, hit, hit_end, hit_empty, done);
        }
        const bool ok = hit >= 0 && !hit_empty;
        a.hit[r] = ok ? hit : -1;
        a.ed[r] = (int8_t)best;
        a.end[r] = (int8_t)(ok ? hit_end : -1);
        a.flags[r] = (uint8_t)(kFlagGeneric | (hit == kHitAmbiguous ? kFlagAmbig : 0u));
    }
    __syncthreads();                                              // the shared state is reused by the workgroup's next row
}

template <int K>
__global__ __launch_bounds__(256) void k_generic_rows(GenericArgs a, const int32_t *__restrict__ list, const int32_t *__restrict__ n_list_ptr) {
    __shared__ GenericShared sh;
    const int n_list = *n_list_ptr;
    for (int g = blockIdx.x; g < n_list; g += gridDim.x) generic_row_wg<K>(a, list[g], sh);
}

// The two rare-row scans behind the pair-seeded kernels as ONE launch: hand-over rows (redo_list: candidate-list
// overflow, short-text pools) through the ordered filtered scan, then irregular rows (gen_list) through the byte-wise
// scan (a workgroup per row, generic_row_wg).  The two lists are disjoint and both are almost always empty, so what this saves is a launch and the idle gap
// in front of it (~3-4 us of a ~100 us step).  Small persistent grid: workgroups stride over both lists.
__global__ __launch_bounds__(256) void k_pair_tail(
    const uint4 *__restrict__ grams, const uint32_t *__restrict__ planes32, int n_blk, int n_wl,
    const int32_t *__restrict__ redo_list, const int32_t *__restrict__ n_redo_ptr,
    const int32_t *__restrict__ gen_list, const int32_t *__restrict__ n_gen_ptr, GenericArgs a) {
    __shared__ uint32_t fifo[kQCap][256];
    __shared__ GenericShared sh;
    const int n_redo = *n_redo_ptr, n_gen = *n_gen_ptr;
    for (int b = blockIdx.x; b * 256 < n_redo; b += gridDim.x)
        scan_filtered_block<2>(b, n_redo, fifo, grams, planes32, n_blk, n_wl, a.ext, a.ext_stride, a.ext_len, redo_list, a.tie,
                               a.hit, a.ed, a.end, a.flags);
    for (int g = blockIdx.x; g < n_gen; g += gridDim.x) generic_row_wg<2>(a, gen_list[g], sh);
}

// Mixed whitelists: a list with a FEW entries outside the packed shape (not 16 nt, or not all ACGT) keeps the fast path.
// The seed indices, planes and the shortcut table are built over the regular entries only (the sub-list keeps their
// relative order), so every result of the fast kernels is an index into the sub-list.  Behind the scan,
//   k_mixed_fix<K>  (1) maps those indices back to the caller's order (reg_to_orig), and (2) tests every scored row
//       against the irregular entries: a row within K = max_ed of one of them is listed in fix_list.  Every other row is
//       final: the reference's fold (read_assignment.py:73-85) ignores an entry whose distance exceeds best_ed <= max_ed,
//       so leaving the irregular entries out changes nothing for it, and the relative order of the entries that do take
//       part is kept.  Rows that took the exact-match shortcut are final too (:61-62 returns before the fold), and a
//       regular putative barcode can only equal a regular entry.
//   k_fix_rows<K>   one workgroup per listed row re-scores it against the WHOLE list: the regular entries by the exact
//       bit-parallel pass on their planes, the irregular ones as below, lanes striding over the entries; only the two
//       smallest events of each distance class (by index in the caller's order) can reach the fold, so the lanes keep
//       two per class, LDS atomics pick the workgroup's two, and one lane replays the fold.
// An irregular entry of 1..32 characters that is not longer than the text is scored bit-parallel too: the text of a
// scored row is all ACGT, so four match masks per entry (bit i: entry[i] is that letter; any other character matches
// nothing) drive the same K-error Shift-And; its accept bit sits at the entry's last character.  Longer entries (the text
// then is the pattern, as in the replaced FFI) take the byte-wise DP.  An EMPTY entry (a blank line of the list) is at
// distance 0 from every text -- every scored row is re-scored -- and a falsy hit in the reference's fold:
// fold_update_falsy / relevant_empties.

// the text of a regular row: 2-bit codes (column j at bits 2j) from its bytes
__device__ __forceinline__ uint64_t row_codes(const RowBytes &rb, int len) {
    uint64_t tc = 0;
#pragma unroll
    for (int j = 0; j < kTextCols; ++j) tc |= (uint64_t)(j < len ? base_code(rb.at(j)) : 0u) << (2 * j);
    return tc;
}

template <int K>
__global__ __launch_bounds__(256) void k_mixed_fix(
    const int32_t *__restrict__ reg_to_orig, const uint4 *__restrict__ irr_peq, const int2 *__restrict__ irr_meta,
    const uint8_t *__restrict__ irr_bytes, const int64_t *__restrict__ irr_off, int n_irr,
    const uint8_t *__restrict__ ext, int ext_stride, const uint8_t *__restrict__ ext_len,
    const int32_t *__restrict__ fast_list, const int32_t *__restrict__ n_fast_ptr, int n, int tie,
    int32_t *__restrict__ fix_list, int32_t *__restrict__ n_fix, int32_t *__restrict__ hit, const uint8_t *__restrict__ flags) {
    const int first = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
    for (int r = first; r < n; r += step) {
        const int h = hit[r];
        if (h >= 0 && !(flags[r] & kFlagGeneric)) hit[r] = reg_to_orig[h];
    }
    const int n_list = *n_fast_ptr;
    for (int slot = first; slot < n_list; slot += step) {
        const int r = fast_list[slot];
        const uint8_t *row = ext + (size_t)r * ext_stride;
        const int len = ext_len[r];
        const uint64_t tc = row_codes(load_row(row, ext_stride), len);
        bool near = false;
        for (int k = 0; k < n_irr && !near; ++k) {
            if (irr_meta[k].x == 0) { near = true; break; }      // '' is at distance 0 from every text
            int ed, end;
            irr_exact<K>(irr_peq[k], irr_meta[k].x, irr_bytes + irr_off[k], row, tc, len, tie, ed, end);
            near = ed <= K;
        }
        if (near) fix_list[atomicAdd(n_fix, 1)] = r;
    }
}

template <int K>
__global__ __launch_bounds__(256) void k_fix_rows(
    const int32_t *__restrict__ reg_to_orig, const uint32_t *__restrict__ planes32, int n_fast,
    const uint4 *__restrict__ irr_peq, const int2 *__restrict__ irr_meta, const uint8_t *__restrict__ irr_bytes,
    const int64_t *__restrict__ irr_off, int n_irr,
    const uint8_t *__restrict__ ext, int ext_stride, const uint8_t *__restrict__ ext_len,
    const int32_t *__restrict__ fix_list, const int32_t *__restrict__ n_fix_ptr, int tie,
    int32_t *__restrict__ hit_out, int8_t *__restrict__ ed_out, int8_t *__restrict__ end_out, uint8_t *__restrict__ flags_out) {
    __shared__ unsigned long long s_min[2][K + 1];               // event word: index << 16 | ed << 8 | end + 1 (an entry longer than the row: end up to 126)
    __shared__ int s_empty;                                       // smallest index of an empty entry, or kNoIndex
    constexpr unsigned long long kNoEvent = ~0ull, kEndEmpty = 255ull;   // end field of the empty entry's event (a real end + 1 is <= 127)
    const int n_fix = *n_fix_ptr;
    const int tid = threadIdx.x;
    for (int f = blockIdx.x; f < n_fix; f += gridDim.x) {         // workgroup-uniform
        const int r = fix_list[f];
        const uint8_t *row = ext + (size_t)r * ext_stride;
        const int len = ext_len[r];
        const RowBytes rb = load_row(row, ext_stride);
        uint32_t XH[kTextCols], XL[kTextCols];
#pragma unroll
        for (int j = 0; j < kTextCols; ++j) {
            const uint32_t code = base_code(j < len ? rb.at(j) : (uint32_t)'A');
            XH[j] = (code & 2u) ? 0u : 0xFFFFFFFFu;
            XL[j] = (code & 1u) ? 0u : 0xFFFFFFFFu;
        }
        const uint64_t tc = row_codes(rb, len);
        unsigned long long keep[K + 1][2];
#pragma unroll
        for (int c = 0; c <= K; ++c) { keep[c][0] = kNoEvent; keep[c][1] = kNoEvent; }
        auto note = [&](int idx, int ed, int end) {
            const unsigned long long ev = ((unsigned long long)(uint32_t)idx << 16) | ((unsigned long long)ed << 8) | (unsigned long long)(end + 1);
#pragma unroll
            for (int c = 0; c <= K; ++c) {
                if (c != ed) continue;
                if (ev < keep[c][0]) { keep[c][1] = keep[c][0]; keep[c][0] = ev; }
                else if (ev < keep[c][1]) keep[c][1] = ev;
            }
        };
        for (int sub = tid; sub < n_fast; sub += 256) {
            const uint32_t p = planes32[sub];
            int ed, end;
            exact_pair<K>(p & 0xFFFFu, p >> 16, XH, XL, len, tie, ed, end);
            if (ed <= K) note(reg_to_orig[sub], ed, end);
        }
        int my_empty = kNoIndex;
        for (int k = tid; k < n_irr; k += 256) {
            if (irr_meta[k].x == 0) { my_empty = min(my_empty, irr_meta[k].y); continue; }   // '' (see generic_row_wg)
            int ed, end;
            irr_exact<K>(irr_peq[k], irr_meta[k].x, irr_bytes + irr_off[k], row, tc, len, tie, ed, end);
            if (ed <= K) note(irr_meta[k].y, ed, end);
        }
        if (tid <= K) { s_min[0][tid] = kNoEvent; s_min[1][tid] = kNoEvent; }
        if (tid == 0) s_empty = kNoIndex;
        __syncthreads();
        if (my_empty != kNoIndex) atomicMin(&s_empty, my_empty);
#pragma unroll
        for (int c = 0; c <= K; ++c) if (keep[c][0] != kNoEvent) atomicMin(&s_min[0][c], keep[c][0]);
        __syncthreads();
#pragma unroll
        for (int c = 0; c <= K; ++c) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (keep[c][t] != kNoEvent && keep[c][t] != s_min[0][c]) atomicMin(&s_min[1][c], keep[c][t]);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned long long ev[2 * (K + 1) + 2];
            int nev = 0;
#pragma unroll
            for (int c = 0; c <= K; ++c) {
                if (s_min[0][c] != kNoEvent) ev[nev++] = s_min[0][c];
                if (s_min[1][c] != kNoEvent) ev[nev++] = s_min[1][c];
            }
            int e_a, e_b;
            relevant_empties(irr_meta, n_irr, s_empty, s_min[0][0] != kNoEvent ? (int)(s_min[0][0] >> 16) : kNoIndex, e_a, e_b);
            if (e_a != kNoIndex) ev[nev++] = ((unsigned long long)(uint32_t)e_a << 16) | kEndEmpty;
            if (e_b != kNoIndex) ev[nev++] = ((unsigned long long)(uint32_t)e_b << 16) | kEndEmpty;
            for (int a = 1; a < nev; ++a) {                       // whitelist order = order of the words (index in the top bits)
                const unsigned long long v = ev[a];
                int b = a - 1;
                while (b >= 0 && ev[b] > v) { ev[b + 1] = ev[b]; --b; }
                ev[b + 1] = v;
            }
            int best = K, hit = kHitNone, hit_end = -1;
            bool done = false, hit_empty = false;
            for (int a = 0; a < nev; ++a) {
                const int e = (int)(ev[a] & 0xFFu);
                fold_update_falsy((int)((ev[a] >> 8) & 0xFFu), e == (int)kEndEmpty ? -1 : e - 1,
                                  (int)(ev[a] >> 16), e == (int)kEndEmpty, best, hit, hit_end, hit_empty, done);
            }
            const bool ok = hit >= 0 && !hit_empty;
            hit_out[r] = ok ? hit : -1;
            ed_out[r] = (int8_t)best;
            end_out[r] = (int8_t)(ok ? hit_end : -1);
            flags_out[r] = (uint8_t)(hit == kHitAmbiguous ? kFlagAmbig : 0u);
        }
        __syncthreads();                                          // s_min is reused by the workgroup's next row
    }
}

// far[i] = 1 iff every string of set B is at global edit distance >= min_ed from A[i]: the test of the empty-drop list,
// `min([edit_distance(k, x, max_ed=5) for x in cells_bc.keys()]) >= 5` (main.py:249-250, :274-275).  lane = A[i]; the B
// strings are wave-uniform (scalar loads), a lane stops at the first close one.  Lengths <= kGenericMax (host checks).
__global__ __launch_bounds__(256) void k_far_from_all(const uint8_t *__restrict__ a_bytes, const int64_t *__restrict__ a_off, int n_a,
                                                     const uint8_t *__restrict__ b_bytes, const int64_t *__restrict__ b_off, int n_b,
                                                     int min_ed, uint8_t *__restrict__ far) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_a) return;
    const uint8_t *a = a_bytes + a_off[i];
    const int la = (int)(a_off[i + 1] - a_off[i]);
    uint8_t mine[kGenericMax];
    for (int k = 0; k < la; ++k) mine[k] = a[k];
    bool is_far = true;
    for (int k = 0; k < n_b && is_far; ++k) {
        const int64_t o = b_off[k];
        const int lb = (int)(b_off[k + 1] - o);
        if (abs(la - lb) >= min_ed) continue;                  // the length gap alone is a lower bound
        if (dev_edit_distance(mine, la, b_bytes + o, lb, min_ed - 1) < min_ed) is_far = false;
    }
    far[i] = is_far ? 1 : 0;
}

// The same test for 16-nt ACGT strings on both sides (the real case: 10x barcodes), bit-parallel: lane = A[i] as the
// TEXT (16 columns as bit-plane masks, as in the scan kernels), each B string the 16-bit PATTERN of Myers' / Hyyro's
// global edit-distance recurrence (first row D[0][j] = j: the shifted-in horizontal delta is +1), ~17 ops per column.
// blockIdx.y splits B into chunks so that a block of a few thousand candidates still fills the chip; far[] must be
// preset to 1 and any lane that meets a close string clears its entry.
constexpr int kFarChunk = 256;
__global__ __launch_bounds__(256) void k_far_from_all_16(const uint32_t *__restrict__ a_codes, int n_a,
                                                        const uint32_t *__restrict__ b_planes, int n_b, int min_ed,
                                                        uint8_t *__restrict__ far) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_a) return;
    const uint32_t code = a_codes[i];
    uint32_t XH[kBcLen], XL[kBcLen];
#pragma unroll
    for (int j = 0; j < kBcLen; ++j) {
        const uint32_t c = (code >> (2 * j)) & 3u;
        XH[j] = (c & 2u) ? 0u : 0xFFFFFFFFu;
        XL[j] = (c & 1u) ? 0u : 0xFFFFFFFFu;
    }
    const int k0 = blockIdx.y * kFarChunk, k1 = min(n_b, k0 + kFarChunk);
    for (int k = k0; k < k1; ++k) {
        const uint32_t p = b_planes[k];                          // wave-uniform
        const uint32_t bh = p & 0xFFFFu, bl = p >> 16;
        uint32_t Pv = 0xFFFFu, Mv = 0u;
        int score = kBcLen;
#pragma unroll
        for (int j = 0; j < kBcLen; ++j) {
            const uint32_t Eq = (bh ^ XH[j]) & (bl ^ XL[j]) & 0xFFFFu;
            const uint32_t Xv = Eq | Mv;
            const uint32_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
            uint32_t Ph = Mv | ~(Xh | Pv);
            uint32_t Mh = Pv & Xh;
            score += (int)((Ph >> (kBcLen - 1)) & 1u) - (int)((Mh >> (kBcLen - 1)) & 1u);
            Ph = (Ph << 1) | 1u;
            Mh <<= 1;
            Pv = (Mh | ~(Xv | Ph)) & 0xFFFFu;
            Mv = Ph & Xv & 0xFFFFu;
        }
        if (score < min_ed) { far[i] = 0; return; }
    }
}

__global__ void k_pair(const uint8_t *a, int la, const uint8_t *b, int lb, int max_ed, int tie, int global_mode, int32_t *out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (global_mode) { out[0] = dev_edit_distance(a, la, b, lb, max_ed); out[1] = -1; }
    else { int end; out[0] = dev_sub_edit_distance(a, la, b, lb, max_ed, tie, end); out[1] = end; }
}

// ------------------------------------------------------------------------------------------
// Row classification + the exact-match shortcut (read_assignment.py:61-62) for regular rows.
// `row.putative_bc in whitelist` = look-up of the 2-bit code in the sorted whitelist: a bucket table
// over the code's top `bucket_bits` bits gives the (short) range, a lower-bound search inside it the
// entry (duplicates -> first index in caller order).  2-3 dependent loads instead of log2(n_wl).
constexpr int kClassifyRows = 2;                              // rows per thread
constexpr int kClassifyThreads = 1024;                        // 2048 rows per workgroup and ONE returned list atomic pair for them:
                                                              // same-address returned atomics serialise at ~13 ns each.  Measured per
                                                              // 1M rows (rows x threads): 4 x 1024 20.0 us, 2 x 1024 19.6, 1 x 1024
                                                              // 23.4, 2 x 512 24.5, 2 x 256 35.6 (1953 workgroups x 2 atomics)
constexpr int kClassifyWaves = kClassifyThreads / 64;

// Staged form for the packed row layout: the rows of a thread go through every step together (row bytes ->
// bucket bounds -> search -> (code, index) pair -> stores), so a thread waits ~6 memory latencies instead of ~24.
// The first form ran the rows one after another through divergent branches: 32 us per 1M rows with every wave
// resident and nothing saturated.  sorted_pairs[i] = {code, original index}, sorted by code (stable).
__global__ __launch_bounds__(kClassifyThreads) void k_classify_packed(
    const uint8_t *__restrict__ ext, int ext_stride, const uint8_t *__restrict__ ext_len,
    const uint8_t *__restrict__ bc_off, const uint8_t *__restrict__ bc_len, int n, int wl_regular,
    const uint2 *__restrict__ sorted_pairs, int n_wl, const int32_t *__restrict__ bucket_start, const uint2 *__restrict__ bucket8,
    int bucket_bits, int shortcut, int force_generic, int max_ed, int32_t *__restrict__ fast_list, uint2 *__restrict__ fast_tc,
    int32_t *__restrict__ gen_list, int32_t *__restrict__ counters, int32_t *__restrict__ hit_out,
    int8_t *__restrict__ ed_out, int8_t *__restrict__ end_out, uint8_t *__restrict__ flags_out,
    const int32_t *__restrict__ prev_hit, int prev_n, int prev_n_wl, uint32_t *__restrict__ prev_partial,
    unsigned long long *__restrict__ prev_counts) {
    // the deferred histogram of the slot's previous run (see kPiggyBins): kHistBlocks workgroups in front of the grid
    static_assert(kHistThreads == kClassifyThreads, "the histogram workgroups ride in this kernel's grid");
    const int hist_blocks = prev_hit ? kHistBlocks : 0;
    if ((int)blockIdx.x < hist_blocks) {                      // workgroup-uniform
        hist_partial_block((int)blockIdx.x, prev_hit, prev_n, prev_n_wl, prev_partial, prev_counts);
        return;
    }
    const int bid = (int)blockIdx.x - hist_blocks;
    __shared__ int s_cnt[2][kClassifyRows][kClassifyWaves];   // [list][k][wave] -> exclusive offsets
    __shared__ int s_base[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    int r[kClassifyRows], len[kClassifyRows], off[kClassifyRows], bl[kClassifyRows];
    bool in[kClassifyRows];
    uint32_t bad_row[kClassifyRows];                          // flags of a row without a barcode: blank, or blank | invalid
#pragma unroll
    for (int k = 0; k < kClassifyRows; ++k) {
        r[k] = (bid * kClassifyRows + k) * kClassifyThreads + threadIdx.x;
        in[k] = r[k] < n;
        const int rr = in[k] ? r[k] : 0;
        len[k] = ext_len[rr]; off[k] = bc_off[rr]; bl[k] = bc_len[rr];
        // a row whose lengths do not fit its stride is never handed to a scan (the host entry points reject such a
        // batch before any launch; rows that arrive in device memory are checked here): blank + BZ_F_INVALID
        bad_row[k] = (in[k] && (len[k] > ext_stride || off[k] + bl[k] > len[k])) ? (kFlagBlank | kFlagInvalid) : kFlagBlank;
        if (!in[k] || bad_row[k] != kFlagBlank) bl[k] = 0;
    }
    RowBytes rb[kClassifyRows];
#pragma unroll
    for (int k = 0; k < kClassifyRows; ++k) rb[k] = load_row_packed(ext + (size_t)(in[k] ? r[k] : 0) * ext_stride);
    const bool search = wl_regular && shortcut;               // wave-uniform
    bool regular[kClassifyRows];
    uint32_t code[kClassifyRows];
    uint2 tc[kClassifyRows];                                  // the text's 2-bit codes, column j at bits 2j ('A' past its end), len in y's bits 8..15
    int lo[kClassifyRows], hi[kClassifyRows];
#pragma unroll
    for (int k = 0; k < kClassifyRows; ++k) {
        bool ok = wl_regular && !force_generic && bl[k] == kBcLen && len[k] >= kBcLen && len[k] <= kTextCols && max_ed <= kMaxK;
        // Four characters per step: sel = their 2-bit codes, one per byte; v_perm turns the codes back into the letters
        // they stand for ("ACTG"[code]) and the word is all ACGT iff that equals the word itself.  The codes of a word
        // are squeezed into 8 bits with two shift-ors.  (The per-character form cost ~15 instructions per character,
        // most of them half-rate compares; this kernel was issue bound on them.)
        uint32_t bad = 0, tlo = 0, thi = 0;
        const int over = len[k] - 16;                         // columns 16.. that belong to the text (0..4 for a regular row)
#pragma unroll
        for (int w = 0; w < kTextCols / 4; ++w) {
            const uint32_t x = rb[k].w[w];
            const uint32_t sel = (x >> 1) & 0x03030303u;
            uint32_t diff = x ^ __builtin_amdgcn_perm(0x47544341u, 0x47544341u, sel);
            uint32_t c = sel | (sel >> 6);
            c = (c | (c >> 12)) & 0xFFu;
            if (w < 4) tlo |= c << (8 * w);
            else {
                diff &= over >= 4 ? 0xFFFFFFFFu : ((1u << ((8 * over) & 31)) - 1u);
                thi = c & ((1u << ((2 * over) & 31)) - 1u);
            }
            bad |= diff;
        }
        ok = ok && bad == 0;
        // the barcode's 16 columns start at off (host-checked: off + 16 <= len, so off <= 4 on a regular row)
        const uint32_t c16 = (uint32_t)((((uint64_t)thi << 32) | tlo) >> ((2 * off[k]) & 63));
        regular[k] = ok; code[k] = c16;
        tc[k] = make_uint2(tlo, thi | ((uint32_t)len[k] << 8));
        lo[k] = 0; hi[k] = 0;
    }
    // bucket8[b] = {code, index | multi << 31} of the bucket's first barcode (index 0xFFFFFFFF: empty bucket).  With ~4-8
    // buckets per barcode almost every look-up ends at this ONE gather; only a row whose bucket holds several barcodes
    // and whose code is not the first of them takes the lower-bound search (bucket bounds, then 1-2 probes).
    int found[kClassifyRows];
    if (search) {
        uint2 e[kClassifyRows];
        bool need = false;
#pragma unroll
        for (int k = 0; k < kClassifyRows; ++k) e[k] = bucket8[(regular[k] && bucket_bits) ? code[k] >> (32 - bucket_bits) : 0u];
#pragma unroll
        for (int k = 0; k < kClassifyRows; ++k) {
            const bool occupied = regular[k] && e[k].y != 0xFFFFFFFFu;
            found[k] = (occupied && e[k].x == code[k]) ? (int)(e[k].y & 0x7FFFFFFFu) : -1;
            const bool more = occupied && found[k] < 0 && (e[k].y >> 31);
            lo[k] = more ? 1 : 0; hi[k] = 0;
            need = need || more;
        }
        if (__builtin_amdgcn_ballot_w64(need) != 0) {
#pragma unroll
            for (int k = 0; k < kClassifyRows; ++k) {
                const bool more = lo[k] != 0;
                const uint32_t b = (more && bucket_bits) ? code[k] >> (32 - bucket_bits) : 0u;
                const int l = bucket_start[b], h = bucket_start[b + 1];
                lo[k] = more ? l : 0; hi[k] = more ? h : 0;
            }
            bool searched[kClassifyRows];
#pragma unroll
            for (int k = 0; k < kClassifyRows; ++k) searched[k] = lo[k] < hi[k];
            for (;;) {                                        // lower bound inside the (tiny) buckets, four rows per step
                bool any = false;
#pragma unroll
                for (int k = 0; k < kClassifyRows; ++k) any = any || lo[k] < hi[k];
                if (__builtin_amdgcn_ballot_w64(any) == 0) break;
                uint32_t v[kClassifyRows];
#pragma unroll
                for (int k = 0; k < kClassifyRows; ++k) v[k] = sorted_pairs[lo[k] < hi[k] ? (lo[k] + hi[k]) >> 1 : 0].x;
#pragma unroll
                for (int k = 0; k < kClassifyRows; ++k) {
                    const int mid = (lo[k] + hi[k]) >> 1;
                    if (lo[k] < hi[k]) { if (v[k] < code[k]) lo[k] = mid + 1; else hi[k] = mid; }
                }
            }
#pragma unroll
            for (int k = 0; k < kClassifyRows; ++k) {
                const uint2 pr = sorted_pairs[min(lo[k], n_wl - 1)];
                if (searched[k] && lo[k] < n_wl && pr.x == code[k]) found[k] = (int)pr.y;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < kClassifyRows; ++k) found[k] = -1;
    }
    int route[kClassifyRows], pos[kClassifyRows];
#pragma unroll
    for (int k = 0; k < kClassifyRows; ++k) {
        int hit = -1, rt = 0;
        uint32_t flags = 0;
        int ed = min(max_ed, 127);
        if (bl[k] == 0) flags = bad_row[k];
        else if (!regular[k]) rt = 2;
        else {
            rt = 1;
            if (found[k] >= 0) { hit = found[k]; ed = 0; flags = kFlagExact; rt = 0; }
        }
        if (in[k]) { hit_out[r[k]] = hit; ed_out[r[k]] = (int8_t)ed; end_out[r[k]] = -1; flags_out[r[k]] = (uint8_t)flags; }
        const unsigned long long m1 = __builtin_amdgcn_ballot_w64(rt == 1), m2 = __builtin_amdgcn_ballot_w64(rt == 2);
        route[k] = rt;
        pos[k] = rt == 1 ? __popcll(m1 & below) : __popcll(m2 & below);
        if (lane == 0) { s_cnt[0][k][wave] = __popcll(m1); s_cnt[1][k][wave] = __popcll(m2); }
    }
    __syncthreads();
    if (threadIdx.x < 2) {                                    // one thread per list: exclusive scan of 64 counts, one atomic
        int *c = &s_cnt[threadIdx.x][0][0];
        int tot = 0;
        for (int i = 0; i < kClassifyRows * kClassifyWaves; ++i) { const int v = c[i]; c[i] = tot; tot += v; }
        s_base[threadIdx.x] = tot ? atomicAdd(&counters[threadIdx.x], tot) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kClassifyRows; ++k) {
        if (route[k] == 1) { const int at = s_base[0] + s_cnt[0][k][wave] + pos[k]; fast_list[at] = r[k]; fast_tc[at] = tc[k]; }
        else if (route[k] == 2) gen_list[s_base[1] + s_cnt[1][k][wave] + pos[k]] = r[k];
    }
}

// General form (any row stride).
__global__ __launch_bounds__(kClassifyThreads) void k_classify(
    const uint8_t *__restrict__ ext, int ext_stride, const uint8_t *__restrict__ ext_len,
    const uint8_t *__restrict__ bc_off, const uint8_t *__restrict__ bc_len, int n, int wl_regular,
    const uint32_t *__restrict__ sorted_codes, const int32_t *__restrict__ sorted_idx, int n_wl,
    const int32_t *__restrict__ bucket_start, int bucket_bits,
    int shortcut, int force_generic, int max_ed, int32_t *__restrict__ fast_list,
    int32_t *__restrict__ gen_list, int32_t *__restrict__ counters, int32_t *__restrict__ hit_out,
    int8_t *__restrict__ ed_out, int8_t *__restrict__ end_out, uint8_t *__restrict__ flags_out) {
    __shared__ int s_cnt[2][kClassifyRows][kClassifyWaves];                // [list][k][wave] -> exclusive offsets
    __shared__ int s_base[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    int route[kClassifyRows];                                 // 0 none, 1 bit-parallel list, 2 generic list
    int pos[kClassifyRows];                                   // rank inside (wave, k)
#pragma unroll
    for (int k = 0; k < kClassifyRows; ++k) {
        const int r = (blockIdx.x * kClassifyRows + k) * kClassifyThreads + threadIdx.x;
        const bool in = r < n;
        const int len = in ? ext_len[r] : 0, off = in ? bc_off[r] : 0;
        int bl = in ? bc_len[r] : 0;
        const bool bad_row = len > ext_stride || off + bl > len;   // see k_classify_packed
        if (bad_row) bl = 0;
        int hit = -1;
        uint32_t flags = 0;
        int ed = min(max_ed, 127);
        int rt = 0;
        if (bl == 0) {
            flags = bad_row ? (kFlagBlank | kFlagInvalid) : kFlagBlank;
        } else {
            bool regular = wl_regular && !force_generic && bl == kBcLen && len >= kBcLen && len <= kTextCols && max_ed <= kMaxK;
            uint32_t code = 0;
            if (regular) {
                const RowBytes rb = load_row(ext + (size_t)r * ext_stride, ext_stride);
                bool ok = true;
#pragma unroll
                for (int j = 0; j < kTextCols; ++j) {
                    const uint32_t c = rb.at(j);
                    ok = ok && (j >= len || is_acgt(c));
                    const int i = j - off;
                    if (i >= 0 && i < kBcLen && j < len) code |= base_code(c) << (2 * i);
                }
                regular = ok;
            }
            if (!regular) rt = 2;
            else {
                rt = 1;
                if (shortcut) {
                    const uint32_t b = bucket_bits ? code >> (32 - bucket_bits) : 0u;
                    int lo = bucket_start[b], hi = bucket_start[b + 1];
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (sorted_codes[mid] < code) lo = mid + 1; else hi = mid; }
                    if (lo < n_wl && sorted_codes[lo] == code) { hit = sorted_idx[lo]; ed = 0; flags = kFlagExact; rt = 0; }
                }
            }
        }
        if (in) { hit_out[r] = hit; ed_out[r] = (int8_t)ed; end_out[r] = -1; flags_out[r] = (uint8_t)flags; }
        const unsigned long long m1 = __builtin_amdgcn_ballot_w64(rt == 1), m2 = __builtin_amdgcn_ballot_w64(rt == 2);
        route[k] = rt;
        pos[k] = rt == 1 ? __popcll(m1 & below) : __popcll(m2 & below);
        if (lane == 0) { s_cnt[0][k][wave] = __popcll(m1); s_cnt[1][k][wave] = __popcll(m2); }
    }
    __syncthreads();
    if (threadIdx.x < 2) {                                    // one thread per list: exclusive scan of 64 counts, one atomic
        int *c = &s_cnt[threadIdx.x][0][0];
        int tot = 0;
        for (int i = 0; i < kClassifyRows * kClassifyWaves; ++i) { const int v = c[i]; c[i] = tot; tot += v; }
        s_base[threadIdx.x] = tot ? atomicAdd(&counters[threadIdx.x], tot) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kClassifyRows; ++k) {
        const int r = (blockIdx.x * kClassifyRows + k) * kClassifyThreads + threadIdx.x;
        if (route[k] == 1) fast_list[s_base[0] + s_cnt[0][k][wave] + pos[k]] = r;
        else if (route[k] == 2) gen_list[s_base[1] + s_cnt[1][k][wave] + pos[k]] = r;
    }
}

// the striped totals folded (see the histogram kernels)
__global__ void k_fold_totals(unsigned long long *__restrict__ counts, int n_wl) {
    if (blockIdx.x != 0 || threadIdx.x >= 2) return;
    unsigned long long s = 0;
    unsigned long long *tot = counts + n_wl + 2;
    for (int k = 0; k < kTotStripes; ++k) { s += tot[2 * k + threadIdx.x]; tot[2 * k + threadIdx.x] = 0; }
    counts[n_wl + threadIdx.x] += s;
}

// Integer-VALU ceiling probe: 8 independent chains of the scan kernel's own full-rate op mix
// (v_bitop3_b32 / v_add_u32 / v_xor_b32, all-VGPR operands), written in asm so that the
// instruction count is exact: 32 VALU instructions per loop iteration.
__global__ __launch_bounds__(256) void k_valu_probe(uint32_t *out, int iters, uint32_t seed) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = tid * 2654435761u + k * 40503u + seed;
    uint32_t b = ~tid ^ seed, c = tid * 97u + 13u;
#define BZ_P4(k)                                              \
    "v_bitop3_b32 %" #k ", %" #k ", %8, %9 bitop3:0xa8\n"      \
    "v_add_u32 %" #k ", %" #k ", %" #k "\n"                      \
    "v_xor_b32 %" #k ", %" #k ", %8\n"                          \
    "v_bitop3_b32 %" #k ", %" #k ", %9, %8 bitop3:0x60\n"
    for (int i = 0; i < iters; ++i) {
        asm volatile(BZ_P4(0) BZ_P4(1) BZ_P4(2) BZ_P4(3) BZ_P4(4) BZ_P4(5) BZ_P4(6) BZ_P4(7)
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                     : "v"(b), "v"(c));
    }
#undef BZ_P4
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s ^= r[k];
    out[tid] = s;
}
constexpr int kProbeOpsPerIter = 32;

}  // namespace bz
