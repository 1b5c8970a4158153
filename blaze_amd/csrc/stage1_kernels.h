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

__device__ inline S1Ends s1_adaptor_ends(const S1View &view, bool kit5, int tie) {
    const uint8_t adpt[10] = {'C', 'T', 'T', 'C', 'C', 'G', 'A', 'T', 'C', 'T'};                         // config.py:12
    const uint8_t tso[13] = {'T', 'T', 'T', 'C', 'T', 'T', 'A', 'T', 'A', 'T', 'G', 'G', 'G'};          // config.py:24
    uint8_t v[kS1Win];
    const int n = min(view.len, kS1Win);
    for (int i = 0; i < n; ++i) v[i] = view.at(i);
    S1Ends ends;
    int end;
    if (kit5) {                                                     // :50-63
        dev_sub_edit_distance(v, n, tso, 13, kS1MaxEd, tie, end);
        if (end > 0) {
            const int tso_start = max(0, end - 13);
            const int lo = max(0, tso_start - kS1D2), hi = max(0, tso_start - kS1D1);
            dev_sub_edit_distance(v + lo, hi - lo, adpt, 10, kS1MaxEd, tie, end);
            if (end > 0) ends.add(lo + end + 1);
        }
        return ends;
    }
    // :118-144.  T_prop[i] covers v[i+1 .. i+4] (cumsum differences, helper.py:61-63); positions >= d1 open windows
    bool open = false;
    int ws = 0, we = 0;
    for (int i = kS1D1; i < n - 4; ++i) {
        if (!(v[i + 1] == 'T' && v[i + 2] == 'T' && v[i + 3] == 'T' && v[i + 4] == 'T')) continue;
        if (!open) { ws = i - kS1D2; we = i - kS1D1; open = true; }
        else if (i - kS1D2 > we) {
            dev_sub_edit_distance(v + max(ws, 0), we - max(ws, 0), adpt, 10, kS1MaxEd, tie, end);
            if (end > 0) ends.add(max(ws, 0) + end + 1);
            ws = i - kS1D2; we = i - kS1D1;
        } else we = i - kS1D1;
    }
    if (open) {
        dev_sub_edit_distance(v + max(ws, 0), we - max(ws, 0), adpt, 10, kS1MaxEd, tie, end);
        if (end > 0) ends.add(max(ws, 0) + end + 1);
    }
    return ends;
}

__global__ __launch_bounds__(64) void k_stage1(const uint8_t *__restrict__ seq, const uint8_t *__restrict__ qual,
                                              const int64_t *__restrict__ off, int n_reads, int kit5, int umi_len, int tie,
                                              int32_t *__restrict__ pass_out, int8_t *__restrict__ strand_out,
                                              int32_t *__restrict__ start_out, int32_t *__restrict__ minq_out,
                                              int32_t *__restrict__ polyt_out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint8_t *s = seq + off[r], *q = qual + off[r];
    const int len = (int)(off[r + 1] - off[r]);
    S1View fwd{s, len, false}, rev{s, len, true};
    const S1Ends em = s1_adaptor_ends(fwd, kit5 != 0, tie);       // strand '-': the read as it is
    const S1Ends ep = s1_adaptor_ends(rev, kit5 != 0, tie);       // strand '+': reverse complement of the last 200 nt
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
        const S1View view = plus ? rev : fwd;
        // :270-278 minimum quality of the 16 barcode bases (the reversed quality string on '+')
        int mq = 1 << 30;
        for (int i = start; i < min(start + kBcLen, len); ++i) mq = min(mq, (int)(plus ? q[len - 1 - i] : q[i]));
        if (mq == 1 << 30) code = kS1PassError;
        else minq = mq - 33;
        // :344-379
        const int umi_end = start + kBcLen + umi_len;
        int p = -1;
        for (int i = umi_end; i + 4 <= len; ++i)
            if (view.at(i) == 'T' && view.at(i + 1) == 'T' && view.at(i + 2) == 'T' && view.at(i + 3) == 'T') { p = i; break; }
        if (p < 0) polyt = umi_end;                                 // unsigned on BOTH strands (:368-369)
        else {
            while (p < len) {
                if (view.at(p) == 'T') { ++p; continue; }
                int t = 0;
                for (int k = p; k < min(p + 10, len); ++k) t += view.at(k) == 'T';
                if (t >= 7) ++p; else break;
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
