// stage1_kernels.h -- device code of BLAZE's stage 1 ("get putative barcodes") per read.
//
// What it replaces (file:line under /root/reference/blaze): class Read of polyT_adaptor_finder.py --
//   find_adaptor_3_prime :85-164   poly-T positions in the first 200 nt of each strand view, search windows 20..50 nt
//                                  upstream, sub_edit_distance(window, 'CTTCCGATCT', 2) per window
//   find_adapter_5_prime :43-83    sub_edit_distance(first 200 nt, TSO, 2), then the adaptor 20..50 nt upstream of the TSO
//   get_strand_and_raw_bc :198-281 strand / ambiguity codes 0, 1, 2, 10, raw_bc_start, minimum barcode quality
//   polyT_trimming_idx :344-379    first 'TTTT' behind the UMI, extended while >= 7 of the next 10 bases are T
// lane = read.  Integer / byte work; the DP is the byte-wise dev_sub_edit_distance of scan_kernels.h (pattern = the
// shorter operand, the window when the lengths are equal; end index under the configured tie rule).  Outputs are the
// five numbers every CSV field of main.py:96-105 is a slice of; the host does the slicing.
#pragma once
#include "scan_kernels.h"

namespace bz {

constexpr int kS1Win = 200;       // config.py:13,17
constexpr int kS1D1 = 20, kS1D2 = 50;   // config.py:19
constexpr int kS1MaxEd = 2;       // config.py:14
constexpr int kS1Flank = 5;
constexpr int kS1PassError = -1;  // the reference raises here (min() of an empty quality slice, :278)

__device__ __forceinline__ uint8_t s1_comp(uint8_t c) {            // helper.reverse_complement (helper.py:14-25)
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        default: return c;
    }
}

struct S1View {                                                     // a read as one strand sees it
    const uint8_t *s;
    int len;
    bool rc;
    __device__ __forceinline__ uint8_t at(int i) const { return rc ? s1_comp(s[len - 1 - i]) : s[i]; }
};

// adaptor end positions found on one strand view: n = 0, 1 or 2 (= more than one distinct), first = the value when n == 1
struct S1Ends { int n = 0, first = -1; __device__ void add(int v) { if (n == 0) { n = 1; first = v; } else if (v != first) n = 2; } };

constexpr int kS1MaxWins = 16;    // search windows per strand view (a new one needs a gap of > 30 columns: at most 6 in 200)
constexpr int kS1Row = 204 + 2 * kS1MaxWins;   // bytes of LDS per lane: the 200-nt window + its search windows; 59 dwords: an odd number of banks

// sub_edit_distance(text, pattern) for a FIXED pattern of M characters and a text of n > M characters that sits in LDS:
// the DP column lives in registers (static unroll over the pattern), one LDS byte per text column.  Same recurrence, same
// end-index tie rule and same cut-off result as dev_sub_edit_distance (which keeps its column in scratch memory: a
// loop-carried store -> load chain there cost ~12 ms per read before this form).
template <int M>
__device__ __forceinline__ int s1_sub_ed_fixed(const uint8_t *text, int n, const uint8_t (&pat)[M], int max_ed, int tie, int &end_idx) {
    int col[M + 1];
#pragma unroll
    for (int i = 0; i <= M; ++i) col[i] = i;
    int best = M, best_end = -1;
    for (int j = 1; j <= n; ++j) {
        const int tc = text[j - 1];
        int diag = col[0];
        col[0] = 0;
#pragma unroll
        for (int i = 1; i <= M; ++i) {
            const int left = col[i];
            int v = diag + (pat[i - 1] != tc);
            v = min(v, col[i - 1] + 1);
            v = min(v, left + 1);
            diag = left;
            col[i] = v;
        }
        const int sc = col[M];
        if (sc < best || (tie && sc == best)) { best = sc; best_end = j - 1; }
    }
    if (best > max_ed) { end_idx = -1; return max_ed + 1; }
    end_idx = best_end;
    return best;
}

// sub_edit_distance(window, pattern) with the reference's role rule: the window is the text only when it is LONGER
// than the pattern (fast path); otherwise it is the pattern (rare: windows clipped at the read start) and the generic
// byte-wise DP runs on a copy.
__device__ __noinline__ int s1_sub_ed_small(const uint8_t *win, int n, const uint8_t *pat, int m, int tie, int *end_idx) {
    uint8_t w[16], q[16];                                           // out of line: rare, and its scratch arrays stay out of the hot code
    for (int i = 0; i < 16; ++i) { w[i] = i < n ? win[i] : 0; q[i] = i < m ? pat[i] : 0; }
    int end;
    const int ed = dev_sub_edit_distance(w, n, q, m, kS1MaxEd, tie, end);
    *end_idx = end;
    return ed;
}

template <int M>
__device__ __forceinline__ int s1_sub_ed(const uint8_t *win, int n, const uint8_t (&pat)[M], int tie, int &end_idx) {
    static_assert(M <= 16, "pattern longer than the small-window buffer");
    if (n > M) return s1_sub_ed_fixed<M>(win, n, pat, kS1MaxEd, tie, end_idx);
    int end;
    const int ed = s1_sub_ed_small(win, n, pat, M, tie, &end);
    end_idx = end;
    return ed;
}

// adaptor ends on one strand view whose first min(len, 200) characters are v[0 .. n) in LDS
__device__ inline S1Ends s1_adaptor_ends(uint8_t *v, int n, bool kit5, int tie) {
    const uint8_t adpt[10] = {'C', 'T', 'T', 'C', 'C', 'G', 'A', 'T', 'C', 'T'};                         // config.py:12
    const uint8_t tso[13] = {'T', 'T', 'T', 'C', 'T', 'T', 'A', 'T', 'A', 'T', 'G', 'G', 'G'};          // config.py:24
    S1Ends ends;
    int end;
    if (kit5) {                                                     // :50-63
        s1_sub_ed<13>(v, n, tso, tie, end);
        if (end > 0) {
            const int tso_start = max(0, end - 13);
            const int lo = max(0, tso_start - kS1D2), hi = max(0, tso_start - kS1D1);
            s1_sub_ed<10>(v + lo, hi - lo, adpt, tie, end);
            if (end > 0) ends.add(lo + end + 1);
        }
        return ends;
    }
    // :118-144.  T_prop[i] covers v[i+1 .. i+4] (cumsum differences, helper.py:61-63); positions >= d1 open windows.
    // Two phases, so that the lanes of a wave run their DPs together: first every lane lists its (merged) search
    // windows behind its text in LDS -- at most 6 fit 200 columns --, then the w-th windows of all lanes are searched
    // in the same pass (with the search inside the position loop a wave ran one DP per DISTINCT window end among its
    // 64 reads, ~0.7 ms per wave).
    uint8_t *wins = v + kS1Win + 4;                                  // [2w] = start (clipped at 0), [2w+1] = end
    int nw = 0;
    bool open = false;
    int ws = 0, we = 0;
    int run = 0;                                                    // T's in a row ending at v[i+4]
    for (int k = kS1D1 + 1; k < min(n, kS1D1 + 4); ++k) run = v[k] == 'T' ? run + 1 : 0;
    for (int i = kS1D1; i < n - 4; ++i) {
        run = v[i + 4] == 'T' ? run + 1 : 0;
        if (run < 4) continue;
        if (!open) { ws = i - kS1D2; we = i - kS1D1; open = true; }
        else if (i - kS1D2 > we) {
            wins[2 * nw] = (uint8_t)max(ws, 0); wins[2 * nw + 1] = (uint8_t)we; ++nw;
            ws = i - kS1D2; we = i - kS1D1;
        } else we = i - kS1D1;
    }
    if (open) { wins[2 * nw] = (uint8_t)max(ws, 0); wins[2 * nw + 1] = (uint8_t)we; ++nw; }
    for (int w = 0; w < nw; ++w) {
        const int lo = wins[2 * w], hi = wins[2 * w + 1];
        s1_sub_ed<10>(v + lo, hi - lo, adpt, tie, end);
        if (end > 0) ends.add(lo + end + 1);
    }
    return ends;
}

// 'T' flags of view[base .. base + 40) as a bit mask (positions past the end read as 0), 8 byte loads in flight; out of
// line: called a few times per read, and inlined thrice it cost the kernel its registers (256 VGPRs + 2.4 KB of spills)
__device__ __noinline__ uint64_t s1_tbits(const S1View &view, int base) {
    uint64_t m = 0;
#pragma unroll 8
    for (int k = 0; k < 40; ++k) {
        const int i = base + k;
        const uint8_t c = view.at(min(max(i, 0), max(view.len - 1, 0)));
        m |= (uint64_t)((i < view.len) & (c == 'T')) << k;
    }
    return m;
}

// one wave = 64 reads.  Per strand view the wave copies every read's first 200 characters into LDS together (coalesced
// byte loads, complemented on the way for the reverse view), then each lane searches its own row.
__global__ __launch_bounds__(64) void k_stage1(const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                              const int64_t *__restrict__ off, int n_reads, int kit5, int umi_len, int tie,
                                              int32_t *__restrict__ pass_out, int8_t *__restrict__ strand_out,
                                              int32_t *__restrict__ start_out, int32_t *__restrict__ minq_out,
                                              int32_t *__restrict__ polyt_out) {
    __shared__ uint8_t s_text[64 * kS1Row];
    const int lane = threadIdx.x;
    const int r = blockIdx.x * 64 + lane;
    const bool active = r < n_reads;
    const long long o = active ? (long long)off[r] : 0;
    const int len = active ? (int)(off[r + 1] - off[r]) : 0;
    const uint8_t *s = seq + o, *q = qual + o;
    S1Ends found[2];                                              // [0] strand '-': the read as it is, [1] strand '+'
#pragma unroll 1
    for (int view = 0; view < 2; ++view) {
        __syncthreads();                                          // the previous view's rows are no longer read
#pragma unroll 4
        for (int k = 0; k < 64; ++k) {                            // read k of the wave, all lanes together (4 reads' loads in flight)
            const long long ok = __shfl(o, k);
            const int lk = __shfl(len, k);
            const int nk = min(lk, kS1Win);
            const uint8_t *sk = seq + ok;
#pragma unroll
            for (int c = 0; c < (kS1Win + 63) / 64; ++c) {
                const int i = c * 64 + lane;
                if (i < nk) s_text[k * kS1Row + i] = view ? s1_comp(sk[lk - 1 - i]) : sk[i];
            }
        }
        __syncthreads();
        if (active) found[view] = s1_adaptor_ends(s_text + lane * kS1Row, min(len, kS1Win), kit5 != 0, tie);
    }
    if (!active) return;
    const S1Ends em = found[0], ep = found[1];
    int code = 0;
    const int n_strands = (em.n > 0) + (ep.n > 0);
    if (n_strands == 0) code = 1;
    else if (n_strands == 2) code = 2;
    else if ((em.n > 0 ? em.n : ep.n) > 1) code = 10;
    int strand = 0, start = -1, minq = -1, polyt = 0;
    if (code == 0) {
        const bool plus = ep.n > 0;
        strand = plus ? '+' : '-';
        start = plus ? ep.first : em.first;
        const S1View view{s, len, plus};
        // :270-278 minimum quality of the 16 barcode bases (the reversed quality string on '+')
        int mq = 1 << 30;
#pragma unroll
        for (int k = 0; k < kBcLen; ++k) {
            const int i = start + k;
            if (i < len) mq = min(mq, (int)(plus ? q[len - 1 - i] : q[i]));
        }
        if (mq == 1 << 30) code = kS1PassError;
        else minq = mq - 33;
        // :344-379 on a stream of 'T' flags, 32 positions per refill
        const int umi_end = start + kBcLen + umi_len;
        int p = -1;
        for (int base = umi_end; base + 4 <= len; base += 32) {    // first 'TTTT' at or behind the UMI's end
            const uint64_t m = s1_tbits(view, base);
            const uint32_t f = (uint32_t)(m & (m >> 1) & (m >> 2) & (m >> 3));
            if (f) { p = base + __builtin_ctz(f); break; }
        }
        if (p < 0) polyt = umi_end;                                 // unsigned on BOTH strands (:368-369)
        else {
            int base = p;
            uint64_t m = s1_tbits(view, base);
            while (p < len) {
                if (p - base > 24) { base = p; m = s1_tbits(view, base); }
                const uint32_t w = (uint32_t)(m >> (p - base));
                if ((w & 1u) || __popc(w & 0x3FFu) >= 7) ++p; else break;
            }
            polyt = plus ? -p : p;
        }
    }
    pass_out[r] = code;
    strand_out[r] = (int8_t)strand;
    start_out[r] = start;
    minq_out[r] = minq;
    polyt_out[r] = polyt;
}

}  // namespace bz
