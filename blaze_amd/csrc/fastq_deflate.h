// fastq_deflate.h -- a deflate encoder for the `.gz` members of the demultiplexed FASTQ (host code, no HIP).
//
// What it replaces: `gzip.compress(out_buffer.encode())`, one gzip member per batch
// (/root/reference/blaze/read_assignment.py:244-245; zlib level 9).  The contract is the DECOMPRESSED stream (the
// member headers carry a wall-clock mtime), so any valid deflate stream of the same bytes is the same output.
//
// Why not zlib: on FASTQ text zlib spends its time in LZ77 match search that finds next to nothing (random ACGT, noisy
// qualities), and then codes sequence and quality characters with ONE Huffman tree per block, which costs about one bit
// per character ("which of the two alphabets is this").  A FASTQ record has structure zlib cannot see:
//     @header \n SEQUENCE \n + \n QUALITIES \n
// This encoder cuts the text at line boundaries into alternating segments -- S = a sequence line, O = everything between
// two sequence lines ("+", the quality line, the next record's header) -- and writes every segment as its OWN dynamic
// Huffman block of literals: a 4-symbol tree for the bases (~2.2 bits per base), a tree fitted to the quality
// distribution for the rest.  No match search at all.  Measured (tools/gz_perf.cpp, profiles/): about the size of zlib
// level 4-6 at several times the speed of level 1.
//
// Output is a complete gzip member (RFC 1952 header, RFC 1951 blocks, CRC-32 + ISIZE).  Any byte string is accepted:
// the segmentation is only a heuristic for where to put block boundaries, every cut is valid.
#pragma once
#include <stdint.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <string>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace fqz {

// CRC-32 (IEEE, reflected: the one gzip uses).  zlib 1.2.11's table-driven crc32 runs at ~1 GB/s per thread, a sixth of this
// encoder's time; carry-less multiplication folds 64 bytes per step (Gopal et al., "Fast CRC computation for generic
// polynomials using PCLMULQDQ"; constants for the reflected polynomial 0xEDB88320).  Whole 16-byte chunks of at least 64
// bytes go through it, the tail through zlib; hosts without PCLMULQDQ use zlib for everything.  Checked against zlib
// on random lengths in tests/test_host_cpu.py (through bz_gzip_fastq_member: a wrong CRC makes gzip refuse the member).
// Where the routine's shape comes from: it is the paper's folding scheme laid out the way its public implementations lay it
// out (Intel's sample code for the paper, and after it zlib-ng's and Chromium zlib's crc32_sse42_simd_: four 128-bit lanes
// folded by k1|k2, reduced by k3|k4, then k5 and the Barrett step with the same published constants); not from the reference,
// which has no native code.
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_clmul(const uint8_t *buf, size_t len, uint32_t crc) {   // len >= 64, len % 16 == 0
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
    alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i *)k1k2);
    buf += 64; len -= 64;
    while (len >= 64) {                            // fold 4 x 128 bits over the next 64 bytes
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00); x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(buf + 0x00)); y6 = _mm_loadu_si128((const __m128i *)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(buf + 0x20)); y8 = _mm_loadu_si128((const __m128i *)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5); x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7); x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64; len -= 64;
    }
    x0 = _mm_load_si128((const __m128i *)k3k4);    // 4 x 128 -> 128
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);       // 128 -> 64
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i *)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i *)poly);    // Barrett reduction 64 -> 32
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

inline uint32_t crc32_of(const uint8_t *buf, size_t n) {
    uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
#if defined(__x86_64__)
    static const bool has_clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (has_clmul && n >= 64) {
        const size_t chunk = n & ~(size_t)15;
        crc = ~crc32_clmul(buf, chunk, ~crc);
        buf += chunk; n -= chunk;
    }
#endif
    while (n) {                                    // zlib's crc32 takes a uInt length
        const size_t chunk = std::min<size_t>(n, (size_t)1 << 30);
        crc = (uint32_t)crc32(crc, (const Bytef *)buf, (uInt)chunk);
        buf += chunk; n -= chunk;
    }
    return crc;
}


constexpr int kLitCodes = 257;                    // literals 0..255 + end-of-block (no length codes: no matches are emitted)
constexpr int kMinSeqSegment = 48;                // a sequence line shorter than this stays in the surrounding O segment

struct BitWriter {                                // deflate packs bits LSB first
    uint8_t *p;
    uint64_t buf = 0;
    int n = 0;
    explicit BitWriter(uint8_t *out) : p(out) {}
    inline void put(uint32_t v, int len) {        // len <= 24, n < 32 on entry
        buf |= (uint64_t)v << n;
        n += len;
        if (n >= 32) { memcpy(p, &buf, 4); p += 4; buf >>= 32; n -= 32; }
    }
    inline void align_byte() {
        while (n > 0) { *p++ = (uint8_t)buf; buf >>= 8; n -= 8; }
        n = 0; buf = 0;
    }
};

// Huffman code lengths (<= max_bits) for m >= 1 symbols that are all in use (f[i] > 0).  The leaves are ranked by
// (frequency, position); the lengths come from Moffat & Katajainen's in-place construction ("In-place calculation of
// minimum-redundancy codes", 1995: three linear passes over one array, no tree nodes).  If the code is deeper than
// max_bits the frequencies are halved (never below 1) and the code rebuilt: a few dozen symbols, a few iterations at most.
constexpr int kMaxSyms = 288;
inline void huffman_lengths_used(const uint32_t *f_in, int m, int max_bits, uint8_t *len) {
    if (m == 1) { len[0] = 1; return; }
    if (m == 2) { len[0] = len[1] = 1; return; }
    uint32_t f[kMaxSyms];
    for (int i = 0; i < m; ++i) f[i] = f_in[i];
    for (;;) {
        // leaves in ascending (frequency, position) order: ord[r] = the r-th smallest, A[r] its weight
        uint16_t ord[kMaxSyms];
        uint64_t A[kMaxSyms];
        uint32_t fmax = 0;
        for (int i = 0; i < m; ++i) fmax = std::max(fmax, f[i]);
        if (m <= 96 && fmax < (1u << 23)) {
            // a few dozen symbols (a quality alphabet): rank of a key = how many keys are smaller, m^2 comparisons that the
            // compiler turns into 8-wide vector code -- no data-dependent branch, no calls
            uint32_t key[96];
            for (int i = 0; i < m; ++i) key[i] = (f[i] << 9) | (uint32_t)i;
            for (int i = 0; i < m; ++i) {
                const uint32_t v = key[i];
                int r = 0;
                for (int j = 0; j < m; ++j) r += key[j] < v;
                ord[r] = (uint16_t)i;
                A[r] = f[i];
            }
        } else {
            uint64_t key[kMaxSyms];                 // frequency above the symbol's position: plain integers sort fastest
            for (int i = 0; i < m; ++i) key[i] = ((uint64_t)f[i] << 16) | (uint64_t)i;
            std::sort(key, key + m);
            for (int i = 0; i < m; ++i) { ord[i] = (uint16_t)(key[i] & 0xFFFFu); A[i] = key[i] >> 16; }
        }
        // pass 1: pair the two smallest of (remaining leaves, finished internal nodes); A[next] = weight of the new
        // internal node, a used internal node's slot keeps the index of its parent
        const int n = m;
        A[0] += A[1];
        int root = 0, leaf = 2;
        for (int next = 1; next < n - 1; ++next) {
            if (leaf >= n || A[root] < A[leaf]) { A[next] = A[root]; A[root++] = (uint64_t)next; }
            else A[next] = A[leaf++];
            if (leaf >= n || (root < next && A[root] < A[leaf])) { A[next] += A[root]; A[root++] = (uint64_t)next; }
            else A[next] += A[leaf++];
        }
        // pass 2: parent indices -> depths of the internal nodes
        A[n - 2] = 0;
        for (int next = n - 3; next >= 0; --next) A[next] = A[A[next]] + 1;
        // pass 3: depths of the internal nodes -> depths of the leaves (deepest = rarest first)
        int avbl = 1, used = 0, dpth = 0, next = n - 1;
        root = n - 2;
        while (avbl > 0) {
            while (root >= 0 && (int)A[root] == dpth) { ++used; --root; }
            while (avbl > used) { A[next--] = (uint64_t)dpth; --avbl; }
            avbl = 2 * used; ++dpth; used = 0;
        }
        if ((int)A[0] <= max_bits) {                // A[0] is the rarest symbol's length: the longest
            for (int i = 0; i < m; ++i) len[ord[i]] = (uint8_t)A[i];
            return;
        }
        for (int i = 0; i < m; ++i) f[i] = (f[i] + 1) >> 1;
    }
}

// the same for an alphabet of n symbols of which some are unused (frequency 0 -> length 0)
inline void huffman_lengths(const uint32_t *freq, int n, int max_bits, uint8_t *len) {
    uint32_t f[kMaxSyms];
    int sym[kMaxSyms], m = 0;
    for (int i = 0; i < n; ++i) { len[i] = 0; if (freq[i]) { sym[m] = i; f[m++] = freq[i]; } }
    if (m == 0) return;
    uint8_t l[kMaxSyms];
    huffman_lengths_used(f, m, max_bits, l);
    for (int i = 0; i < m; ++i) len[sym[i]] = l[i];
}

// canonical codes of RFC 1951 3.2.2 for the symbols sym[0..m) (ascending) with lengths l[0..m), bit-reversed so that
// BitWriter::put sends them most significant bit first
inline uint32_t reverse16(uint32_t v) {            // bit reversal of a 16-bit value
    v = ((v >> 1) & 0x5555u) | ((v & 0x5555u) << 1);
    v = ((v >> 2) & 0x3333u) | ((v & 0x3333u) << 2);
    v = ((v >> 4) & 0x0F0Fu) | ((v & 0x0F0Fu) << 4);
    return ((v >> 8) & 0x00FFu) | ((v & 0x00FFu) << 8);
}
inline void canonical_codes_used(const uint8_t *l, int m, uint16_t *code) {
    int count[16] = {0}, next[16];
    for (int i = 0; i < m; ++i) ++count[l[i]];
    int c = 0;
    count[0] = 0;
    next[0] = 0;
    for (int b = 1; b < 16; ++b) { c = (c + count[b - 1]) << 1; next[b] = c; }
    for (int i = 0; i < m; ++i) code[i] = (uint16_t)(reverse16((uint32_t)next[l[i]]++) >> (16 - l[i]));
}

// The block header of a dynamic-Huffman block of literals (after the BFINAL bit): BTYPE, HLIT / HDIST / HCLEN, the
// code-length code and the run-length coded lengths of the 257 literal codes + the one (unused) distance code.
// sym[0..m) ascending with sym[m-1] == 256 (end of block), l[k] = code length of sym[k]; every other literal has length 0.
// The run-length tokens are made from the SHORT list of symbols in use (the gaps between them are the zero runs), not by
// walking all 258 lengths.
inline void put_tree_header(BitWriter &bw, const int *sym, const uint8_t *l, int m) {
    struct Tok { uint8_t sym, extra_bits; uint16_t extra; };
    Tok tok[kLitCodes + 8];
    int nt = 0;
    auto zeros = [&](int left) {
        while (left >= 11) { const int r = std::min(left, 138); tok[nt++] = {18, 7, (uint16_t)(r - 11)}; left -= r; }
        if (left >= 3) { tok[nt++] = {17, 3, (uint16_t)(left - 3)}; left = 0; }
        while (left-- > 0) tok[nt++] = {0, 0, 0};
    };
    int pos = 0;                                   // next literal whose length has not been coded yet
    for (int k = 0; k < m;) {
        zeros(sym[k] - pos);
        const int v = l[k];
        int run = 1;
        while (k + run < m && sym[k + run] == sym[k] + run && l[k + run] == v) ++run;
        tok[nt++] = {(uint8_t)v, 0, 0};
        int left = run - 1;
        while (left >= 3) { const int r = std::min(left, 6); tok[nt++] = {16, 2, (uint16_t)(r - 3)}; left -= r; }
        while (left-- > 0) tok[nt++] = {(uint8_t)v, 0, 0};
        pos = sym[k] + run;
        k += run;
    }
    zeros(kLitCodes + 1 - pos);                    // [257] = the one distance code, of zero bits: "no distance codes are used" (RFC 1951 3.2.7)
    uint32_t cl_freq[19] = {0};
    for (int t = 0; t < nt; ++t) ++cl_freq[tok[t].sym];
    uint8_t cl_len[19];
    huffman_lengths(cl_freq, 19, 7, cl_len);
    {   // inflate rejects an incomplete code-length code, even one of a single symbol: give a second symbol a code then
        int used = 0;
        for (int k = 0; k < 19; ++k) used += cl_len[k] != 0;
        if (used == 1) cl_len[cl_len[0] ? 1 : 0] = 1;
    }
    uint16_t cl_code[19];
    {
        uint8_t ul[19];
        uint16_t uc[19];
        int us[19], um = 0;
        for (int k = 0; k < 19; ++k) if (cl_len[k]) { us[um] = k; ul[um++] = cl_len[k]; }
        canonical_codes_used(ul, um, uc);
        for (int k = 0; k < 19; ++k) cl_code[k] = 0;
        for (int k = 0; k < um; ++k) cl_code[us[k]] = uc[k];
    }
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cl_len[order[hclen - 1]] == 0) --hclen;
    bw.put(2u, 2);                                 // BTYPE 10: dynamic Huffman
    bw.put(0u, 5);                                 // HLIT: 257 literal/length codes
    bw.put(0u, 5);                                 // HDIST: 1 distance code
    bw.put((uint32_t)(hclen - 4), 4);
    for (int k = 0; k < hclen; ++k) bw.put(cl_len[order[k]], 3);
    for (int t = 0; t < nt; ++t) {
        bw.put(cl_code[tok[t].sym], cl_len[tok[t].sym]);
        if (tok[t].extra_bits) bw.put(tok[t].extra, tok[t].extra_bits);
    }
}

// A literal tree ready for use: the codes of the bytes it covers, and its block header as bits (made once, replayed in
// front of every block that uses the tree).
struct LitTree {
    uint32_t cl[256];                              // code | length << 16; only the entries of covered bytes are ever read
    uint32_t eob;                                  // the same for end-of-block
    int longest = 0;                               // longest literal code
    int hdr_bits = 0;
    uint8_t hdr[640];                              // <= 266 tokens of <= 14 bits + 71 bits, and the writer's 4-byte stores
};

// byte histogram of s[0..n) ADDED to h: four tables so that runs of one character do not serialise on a single counter
inline void histogram_add(const uint8_t *s, size_t n, uint32_t (&h)[4][256]) {
    size_t i = 0;
    for (; i + 4 <= n; i += 4) { ++h[0][s[i]]; ++h[1][s[i + 1]]; ++h[2][s[i + 2]]; ++h[3][s[i + 3]]; }
    for (; i < n; ++i) ++h[0][s[i]];
}

// the Huffman tree of a histogram (+ one end-of-block)
inline void build_lit_tree(uint32_t (&h)[4][256], LitTree &t) {
    for (int c = 0; c < 256; ++c) h[0][c] += h[1][c] + h[2][c] + h[3][c];      // (vectorised by the compiler)
    // the symbols in use, ascending (end-of-block = 256 last): everything below works on this short list
    int sym[kLitCodes], m = 0;
    uint32_t f[kLitCodes];
    for (int c = 0; c < 256; ++c)
        if (h[0][c]) { sym[m] = c; f[m++] = h[0][c]; }
    sym[m] = 256; f[m++] = 1;                      // end of block
    uint8_t l[kLitCodes];
    huffman_lengths_used(f, m, 15, l);
    uint16_t cd[kLitCodes];
    canonical_codes_used(l, m, cd);
    t.longest = 0;
    for (int k = 0; k + 1 < m; ++k) { t.cl[sym[k]] = (uint32_t)cd[k] | ((uint32_t)l[k] << 16); t.longest = std::max<int>(t.longest, l[k]); }
    t.eob = (uint32_t)cd[m - 1] | ((uint32_t)l[m - 1] << 16);
    BitWriter hw(t.hdr);
    put_tree_header(hw, sym, l, m);
    t.hdr_bits = (int)(hw.p - t.hdr) * 8 + hw.n;
    hw.align_byte();
}

// s[0..n) as one block of literals coded with `t` (which covers every byte of it).  Upper bound of its size: block_bound().
inline void put_block(BitWriter &bw, const uint8_t *s, size_t n, bool last, const LitTree &t) {
    bw.put(last ? 1u : 0u, 1);
    {
        int left = t.hdr_bits;
        const uint8_t *h = t.hdr;
        for (; left >= 24; left -= 24, h += 3) bw.put((uint32_t)h[0] | ((uint32_t)h[1] << 8) | ((uint32_t)h[2] << 16), 24);
        if (left > 0) bw.put(((uint32_t)h[0] | ((uint32_t)h[1] << 8) | ((uint32_t)h[2] << 16)) & ((1u << left) - 1u), left);
    }
    // the literals: code and length of a byte fetched as one 32-bit entry
    const uint32_t *cl = t.cl;
    uint64_t buf = bw.buf;
    int nb = bw.n;
    uint8_t *p = bw.p;
    size_t k = 0;
    if (t.longest <= 14 && n >= 8) {
        // Four literals per step, joined pairwise BEFORE they meet the bit buffer (the only serial chain is one shift-or
        // and one add per step), and a flush without a branch: store eight bytes, keep the 0..7 bits that did not fill a
        // byte.  nb <= 7 on entry of a step, 4 x 14 = 56 bits added at most.
        if (nb >= 8) { memcpy(p, &buf, 4); const int bytes = nb >> 3; p += bytes; buf >>= 8 * bytes; nb &= 7; }
        for (; k + 4 <= n; k += 4) {
            const uint32_t e0 = cl[s[k]], e1 = cl[s[k + 1]], e2 = cl[s[k + 2]], e3 = cl[s[k + 3]];
            const uint32_t l0 = e0 >> 16, l2 = e2 >> 16;
            const uint64_t c01 = (uint64_t)(e0 & 0xFFFFu) | ((uint64_t)(e1 & 0xFFFFu) << l0);
            const uint64_t c23 = (uint64_t)(e2 & 0xFFFFu) | ((uint64_t)(e3 & 0xFFFFu) << l2);
            const uint32_t l01 = l0 + (e1 >> 16), l23 = l2 + (e3 >> 16);
            buf |= (c01 | (c23 << l01)) << nb;
            nb += (int)(l01 + l23);
            memcpy(p, &buf, 8);
            p += nb >> 3;
            buf >>= nb & ~7;
            nb &= 7;
        }
    }
    for (; k + 2 <= n; k += 2) {                   // two literals (<= 30 bits) per flush test: nb < 32 on entry, < 62 before it
        const uint32_t e0 = cl[s[k]], e1 = cl[s[k + 1]];
        buf |= (uint64_t)(e0 & 0xFFFFu) << nb;
        nb += (int)(e0 >> 16);
        buf |= (uint64_t)(e1 & 0xFFFFu) << nb;
        nb += (int)(e1 >> 16);
        if (nb >= 32) { memcpy(p, &buf, 4); p += 4; buf >>= 32; nb -= 32; }
    }
    if (k < n) {
        const uint32_t e = cl[s[k]];
        buf |= (uint64_t)(e & 0xFFFFu) << nb;
        nb += (int)(e >> 16);
        if (nb >= 32) { memcpy(p, &buf, 4); p += 4; buf >>= 32; nb -= 32; }
    }
    bw.buf = buf; bw.n = nb; bw.p = p;
    bw.put(t.eob & 0xFFFFu, (int)(t.eob >> 16));   // end of block
}

// One segment as a block with a tree of its own.
inline void put_literal_block(BitWriter &bw, const uint8_t *s, size_t n, bool last) {
    uint32_t h[4][256];
    memset(h, 0, sizeof h);
    histogram_add(s, n, h);
    LitTree t;
    build_lit_tree(h, t);
    put_block(bw, s, n, last, t);
}

// ---- sequence lines of plain ACGT: no histogram table, no tree construction, four bases per table look-up -------------
// A line of A/C/G/T + '\n' whose base counts are balanced (the two rarest weights together at least the most frequent
// one: any real read) has ONE optimal Huffman shape: three bases 2 bits, the rarest base 3 bits, '\n' and end-of-block
// 4 bits.  That leaves four possible trees (which base is the rare one); their block headers, per-base codes and the
// codes of all 256 base quadruples are built once, with the same routines as every other block.
struct SeqTree {
    uint32_t quad[256];                            // code bits | total length << 16 of four bases; index b0 | b1 << 2 | b2 << 4 | b3 << 6
    uint32_t base[4];                              // code | length << 16 by base index (ascii >> 1) & 3: A 0, C 1, T 2, G 3
    uint32_t nl, eob;
    uint16_t hdr[16];                              // the block header after BFINAL, 16 bits per entry
    int hdr_bits;
};
inline const SeqTree *seq_trees() {
    static const SeqTree *const trees = [] {
        static SeqTree t[4];
        static const int sym[6] = {'\n', 'A', 'C', 'G', 'T', 256};
        static const int base_of_sym[4] = {0, 1, 3, 2};               // A C G T (ascii order) -> base index
        for (int rare = 0; rare < 4; ++rare) {                        // rare = position in A C G T
            uint8_t l[6] = {4, 2, 2, 2, 2, 4};
            l[1 + rare] = 3;
            uint16_t cd[6];
            canonical_codes_used(l, 6, cd);
            SeqTree &o = t[rare];
            for (int b = 0; b < 4; ++b) o.base[base_of_sym[b]] = (uint32_t)cd[1 + b] | ((uint32_t)l[1 + b] << 16);
            o.nl = (uint32_t)cd[0] | (4u << 16);
            o.eob = (uint32_t)cd[5] | (4u << 16);
            for (int q = 0; q < 256; ++q) {
                uint32_t bits = 0, len = 0;
                for (int j = 0; j < 4; ++j) {
                    const uint32_t e = o.base[(q >> (2 * j)) & 3];
                    bits |= (e & 0xFFFFu) << len;
                    len += e >> 16;
                }
                o.quad[q] = bits | (len << 16);
            }
            uint8_t tmp[64] = {0};
            BitWriter bw(tmp);
            put_tree_header(bw, sym, l, 6);
            o.hdr_bits = (int)(bw.p - tmp) * 8 + bw.n;
            bw.align_byte();
            for (int k = 0; k < 16; ++k) o.hdr[k] = (uint16_t)(tmp[2 * k] | (tmp[2 * k + 1] << 8));
        }
        return t;
    }();
    return trees;
}

#if defined(__x86_64__)
// counts of A, C, G, T in s[0..n) (index: ascii order A C G T); true when every byte is one of them
__attribute__((target("avx2,popcnt"))) inline bool acgt_counts(const uint8_t *s, size_t n, uint32_t cnt[4]) {
    const __m256i va = _mm256_set1_epi8('A'), vc = _mm256_set1_epi8('C'), vg = _mm256_set1_epi8('G'), vt = _mm256_set1_epi8('T');
    uint32_t a = 0, c = 0, g = 0, t = 0;
    size_t k = 0;
    for (; k + 32 <= n; k += 32) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)(s + k));
        a += (uint32_t)__builtin_popcount((uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, va)));
        c += (uint32_t)__builtin_popcount((uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, vc)));
        g += (uint32_t)__builtin_popcount((uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, vg)));
        t += (uint32_t)__builtin_popcount((uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, vt)));
    }
    for (; k < n; ++k) { a += s[k] == 'A'; c += s[k] == 'C'; g += s[k] == 'G'; t += s[k] == 'T'; }
    cnt[0] = a; cnt[1] = c; cnt[2] = g; cnt[3] = t;
    return (size_t)a + c + g + t == n;
}
#endif

// The sequence line s[0..n) (its last byte is '\n') as a block with one of the four fixed trees; false = not such a line
// (other characters, lopsided base counts, no AVX2): the caller writes it as an ordinary block.
inline bool put_sequence_block(BitWriter &bw, const uint8_t *s, size_t n, bool last) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("popcnt");
    if (!avx2 || n < 17 || s[n - 1] != '\n') return false;
    const size_t nb_bases = n - 1;
    uint32_t cnt[4];
    if (!acgt_counts(s, nb_bases, cnt)) return false;
    int rare = 0;
    for (int b = 1; b < 4; ++b) if (cnt[b] < cnt[rare]) rare = b;
    uint32_t w[4] = {cnt[0], cnt[1], cnt[2], cnt[3]};
    w[rare] += 2;                                   // the rare base carries '\n' and end-of-block below it
    std::sort(w, w + 4);
    if (w[0] + w[1] < w[3]) return false;           // a lopsided line: its own Huffman tree is not the balanced one
    const SeqTree &tr = seq_trees()[rare];
    bw.put(last ? 1u : 0u, 1);
    for (int k = 0, left = tr.hdr_bits; left > 0; ++k, left -= 16) bw.put(left >= 16 ? tr.hdr[k] : (tr.hdr[k] & ((1u << left) - 1u)), std::min(left, 16));
    uint64_t buf = bw.buf;
    int nb = bw.n;
    uint8_t *p = bw.p;
    size_t k = 0;
    if (nb >= 8) { memcpy(p, &buf, 4); const int bytes = nb >> 3; p += bytes; buf >>= 8 * bytes; nb &= 7; }
    for (; k + 16 <= nb_bases; k += 16) {           // sixteen bases = four table entries (<= 48 bits) per step, branch-free flush
        uint64_t v0, v1;
        memcpy(&v0, s + k, 8);
        memcpy(&v1, s + k + 8, 8);
        v0 = (v0 >> 1) & 0x0303030303030303ull;     // base indices, one per byte
        v1 = (v1 >> 1) & 0x0303030303030303ull;
        // b0 | b1 << 2 | b2 << 4 | b3 << 6 of each four bytes: the multiplication adds the shifted copies, the top byte has the sum
        const uint32_t e0 = tr.quad[((uint32_t)v0 * 0x01041040u) >> 24], e1 = tr.quad[((uint32_t)(v0 >> 32) * 0x01041040u) >> 24];
        const uint32_t e2 = tr.quad[((uint32_t)v1 * 0x01041040u) >> 24], e3 = tr.quad[((uint32_t)(v1 >> 32) * 0x01041040u) >> 24];
        const uint32_t l0 = e0 >> 16, l2 = e2 >> 16;
        const uint64_t c01 = (uint64_t)(e0 & 0xFFFFu) | ((uint64_t)(e1 & 0xFFFFu) << l0);
        const uint64_t c23 = (uint64_t)(e2 & 0xFFFFu) | ((uint64_t)(e3 & 0xFFFFu) << l2);
        const uint32_t l01 = l0 + (e1 >> 16), l23 = l2 + (e3 >> 16);
        buf |= (c01 | (c23 << l01)) << nb;
        nb += (int)(l01 + l23);
        memcpy(p, &buf, 8);
        p += nb >> 3;
        buf >>= nb & ~7;
        nb &= 7;
    }
    for (; k < nb_bases; ++k) {
        const uint32_t e = tr.base[(s[k] >> 1) & 3];
        buf |= (uint64_t)(e & 0xFFFFu) << nb; nb += (int)(e >> 16);
        if (nb >= 32) { memcpy(p, &buf, 4); p += 4; buf >>= 32; nb -= 32; }
    }
    bw.buf = buf; bw.n = nb; bw.p = p;
    bw.put(tr.nl & 0xFFFFu, 4);
    bw.put(tr.eob & 0xFFFFu, 4);
    return true;
#else
    (void)bw; (void)s; (void)n; (void)last;
    return false;
#endif
}

// worst case of a block: 15 bits per literal never happens for a real histogram, but it is the only bound that is cheap to state
inline size_t block_bound(size_t n) { return n * 2 + 512; }

// member text -> segments [cuts[k], cuts[k + 1]) covering [0, n), seq[k] = 1 for a sequence line (lines 1, 5, 9, ... 0-based)
// that is long enough for a block of its own
inline void fastq_segments(const char *t, size_t n, std::vector<size_t> &cuts, std::vector<uint8_t> &seq) {
    cuts.clear();
    seq.clear();
    cuts.push_back(0);
    size_t pos = 0;
    unsigned line = 0;
    while (pos < n) {
        const char *nl = (const char *)memchr(t + pos, '\n', n - pos);
        const size_t end = nl ? (size_t)(nl - t) + 1 : n;
        if ((line & 3u) == 1u && end - pos >= (size_t)kMinSeqSegment) {      // a sequence line worth a block of its own
            if (pos != cuts.back()) { cuts.push_back(pos); seq.push_back(0); }
            cuts.push_back(end);
            seq.push_back(1);
        }
        pos = end;
        ++line;
    }
    if (cuts.back() != n) { cuts.push_back(n); seq.push_back(0); }
}

// How many consecutive "other" segments (quality line + the next header) share one Huffman tree, fitted to their summed
// histogram.  1 = every segment its own tree.  Building a tree costs about as much as coding 1.5 KB with it, so sharing is
// faster (4 segments per tree: ~1.3x), but quality distributions differ from read to read: with per-read mean qualities
// spread as in a nanopore run (sigma 6) a tree for 2 / 4 / 8 reads makes the member 2.4 / 4.0 / 5.0 % larger, and 0.7 %
// even when all reads have the same distribution (tools/gz_perf.cpp).  Kept at 1.
constexpr int kTreeShare = 1;

// `text` -> one complete gzip member in `out` (resized to fit).
inline void gzip_fastq_member(const char *text, size_t n, std::string &out, int tree_share = kTreeShare) {
    static thread_local std::vector<size_t> cuts;
    static thread_local std::vector<uint8_t> seq;
    fastq_segments(text, n, cuts, seq);
    const size_t n_seg = seq.size();
    size_t bound = 10 + 8 + 16;
    for (size_t k = 0; k < n_seg; ++k) bound += block_bound(cuts[k + 1] - cuts[k]);
    // the worst-case buffer is ~5x the output: a per-thread scratch that is never value-initialised again, not `out`
    static thread_local std::vector<uint8_t> scratch;
    if (scratch.size() < bound) scratch.resize(bound + bound / 4);
    uint8_t *o = scratch.data();
    static const uint8_t hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};      // deflate, no flags, mtime 0, unix
    memcpy(o, hdr, 10);
    BitWriter bw(o + 10);
    if (n == 0) {                                  // gzip.compress(b''): one empty final block (fixed Huffman: just the 7-bit end-of-block code 0)
        bw.put(1u, 1); bw.put(1u, 2); bw.put(0u, 7);
    } else {
        const uint8_t *t8 = (const uint8_t *)text;
        LitTree tree;
        size_t k = 0;
        while (k < n_seg) {
            // a group: segments [k, e) holding at most tree_share "other" segments, which share `tree`
            size_t e = k;
            int others = 0;
            uint32_t h[4][256];
            memset(h, 0, sizeof h);
            while (e < n_seg && (seq[e] || others < tree_share)) {
                if (!seq[e]) { histogram_add(t8 + cuts[e], cuts[e + 1] - cuts[e], h); ++others; }
                ++e;
            }
            if (others) build_lit_tree(h, tree);
            for (; k < e; ++k) {
                const uint8_t *seg = t8 + cuts[k];
                const size_t len = cuts[k + 1] - cuts[k];
                const bool last = k + 1 == n_seg;
                if (!seq[k]) put_block(bw, seg, len, last, tree);
                else if (!put_sequence_block(bw, seg, len, last)) put_literal_block(bw, seg, len, last);   // N, lower case, lopsided counts
            }
        }
    }
    bw.align_byte();
    uint8_t *p = bw.p;
    const uint32_t crc = crc32_of((const uint8_t *)text, n);
    const uint32_t isize = (uint32_t)n;
    memcpy(p, &crc, 4); memcpy(p + 4, &isize, 4);
    out.assign((const char *)o, (size_t)(p + 8 - o));
}

}  // namespace fqz
