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

// Huffman code lengths (<= max_bits) for m >= 1 symbols that are all in use (f[i] > 0).  Two-queue construction over the
// sorted leaves; if the tree is deeper than max_bits the frequencies are halved (never below 1) and the tree rebuilt:
// a few dozen symbols, a few iterations at most.
constexpr int kMaxSyms = 288;
inline void huffman_lengths_used(const uint32_t *f_in, int m, int max_bits, uint8_t *len) {
    if (m == 1) { len[0] = 1; return; }
    uint32_t f[kMaxSyms];
    for (int i = 0; i < m; ++i) f[i] = f_in[i];
    for (;;) {
        uint64_t key[kMaxSyms];                     // frequency above the symbol's position: plain integers sort fastest
        for (int i = 0; i < m; ++i) key[i] = ((uint64_t)f[i] << 16) | (uint64_t)i;
        std::sort(key, key + m);
        int ord[kMaxSyms];
        uint64_t w[2 * kMaxSyms];
        int parent[2 * kMaxSyms], depth[2 * kMaxSyms];
        for (int i = 0; i < m; ++i) { ord[i] = (int)(key[i] & 0xFFFFu); w[i] = key[i] >> 16; }
        int leaf = 0, inode = m, next = m;          // leaves [0, m), internal nodes [m, 2m-1) created in weight order
        auto take = [&]() {
            if (leaf < m && (inode >= next || w[leaf] <= w[inode])) return leaf++;
            return inode++;
        };
        while (next < 2 * m - 1) {
            const int a = take(), b = take();
            w[next] = w[a] + w[b];
            parent[a] = parent[b] = next;
            ++next;
        }
        depth[2 * m - 2] = 0;
        int deepest = 0;
        for (int i = 2 * m - 3; i >= 0; --i) { depth[i] = depth[parent[i]] + 1; if (i < m && depth[i] > deepest) deepest = depth[i]; }
        if (deepest <= max_bits) {
            for (int i = 0; i < m; ++i) len[ord[i]] = (uint8_t)depth[i];
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
inline void canonical_codes_used(const uint8_t *l, int m, uint16_t *code) {
    int count[16] = {0}, next[16];
    for (int i = 0; i < m; ++i) ++count[l[i]];
    int c = 0;
    count[0] = 0;
    for (int b = 1; b < 16; ++b) { c = (c + count[b - 1]) << 1; next[b] = c; }
    for (int i = 0; i < m; ++i) {
        uint32_t v = (uint32_t)next[l[i]]++, r = 0;
        for (int b = 0; b < l[i]; ++b) { r = (r << 1) | (v & 1u); v >>= 1; }
        code[i] = (uint16_t)r;
    }
}

// One segment as a dynamic-Huffman block of literals.  Upper bound of its size: see block_bound().
inline void put_literal_block(BitWriter &bw, const uint8_t *s, size_t n, bool last) {
    // histogram, four tables so that runs of one character do not serialise on a single counter
    uint32_t h[4][256];
    memset(h, 0, sizeof h);
    size_t i = 0;
    for (; i + 4 <= n; i += 4) { ++h[0][s[i]]; ++h[1][s[i + 1]]; ++h[2][s[i + 2]]; ++h[3][s[i + 3]]; }
    for (; i < n; ++i) ++h[0][s[i]];
    // the symbols in use, ascending (end-of-block = 256 last): everything below works on this short list
    int sym[kLitCodes], m = 0;
    uint32_t f[kLitCodes];
    for (int c = 0; c < 256; ++c) {
        const uint32_t v = h[0][c] + h[1][c] + h[2][c] + h[3][c];
        if (v) { sym[m] = c; f[m++] = v; }
    }
    sym[m] = 256; f[m++] = 1;                      // end of block
    uint8_t l[kLitCodes];
    huffman_lengths_used(f, m, 15, l);
    uint16_t cd[kLitCodes];
    canonical_codes_used(l, m, cd);
    uint8_t len[kLitCodes + 1];
    memset(len, 0, sizeof len);                    // [257] = the one distance code, of zero bits: "no distance codes are used" (RFC 1951 3.2.7)
    for (int k = 0; k < m; ++k) len[sym[k]] = l[k];

    // run-length code of the 258 code lengths with the code-length alphabet (16: repeat previous 3-6, 17: zeros 3-10, 18: zeros 11-138)
    struct Tok { uint8_t sym, extra_bits; uint16_t extra; };
    Tok tok[kLitCodes + 1];
    int nt = 0;
    uint32_t cl_freq[19] = {0};
    const int total = kLitCodes + 1;
    for (int k = 0; k < total;) {
        const int v = len[k];
        int run = 1;
        while (k + run < total && len[k + run] == v) ++run;
        int left = run;
        if (v == 0) {
            while (left >= 11) { const int r = std::min(left, 138); tok[nt++] = {18, 7, (uint16_t)(r - 11)}; left -= r; }
            if (left >= 3) { tok[nt++] = {17, 3, (uint16_t)(left - 3)}; left = 0; }
            while (left-- > 0) tok[nt++] = {0, 0, 0};
        } else {
            tok[nt++] = {(uint8_t)v, 0, 0};
            --left;
            while (left >= 3) { const int r = std::min(left, 6); tok[nt++] = {16, 2, (uint16_t)(r - 3)}; left -= r; }
            while (left-- > 0) tok[nt++] = {(uint8_t)v, 0, 0};
        }
        k += run;
    }
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

    bw.put(last ? 1u : 0u, 1);
    bw.put(2u, 2);                                 // BTYPE 10: dynamic Huffman
    bw.put(0u, 5);                                 // HLIT: 257 literal/length codes
    bw.put(0u, 5);                                 // HDIST: 1 distance code
    bw.put((uint32_t)(hclen - 4), 4);
    for (int k = 0; k < hclen; ++k) bw.put(cl_len[order[k]], 3);
    for (int t = 0; t < nt; ++t) {
        bw.put(cl_code[tok[t].sym], cl_len[tok[t].sym]);
        if (tok[t].extra_bits) bw.put(tok[t].extra, tok[t].extra_bits);
    }
    // the literals: code and length of a byte fetched as one 32-bit entry
    uint32_t cl[256];                              // only the entries of bytes that occur are ever read
    for (int k = 0; k + 1 < m; ++k) cl[sym[k]] = (uint32_t)cd[k] | ((uint32_t)l[k] << 16);
    uint64_t buf = bw.buf;
    int nb = bw.n;
    uint8_t *p = bw.p;
    size_t k = 0;
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
    bw.put(cd[m - 1], l[m - 1]);                   // end of block
}

// worst case of a block: 15 bits per literal never happens for a real histogram, but it is the only bound that is cheap to state
inline size_t block_bound(size_t n) { return n * 2 + 512; }

// member text -> [(start, end, is_sequence_line)] segments covering [0, n): lines 1, 5, 9, ... (0-based) are sequence lines
inline void fastq_segments(const char *t, size_t n, std::vector<size_t> &cuts) {
    cuts.clear();
    cuts.push_back(0);
    size_t pos = 0;
    unsigned line = 0;
    while (pos < n) {
        const char *nl = (const char *)memchr(t + pos, '\n', n - pos);
        const size_t end = nl ? (size_t)(nl - t) + 1 : n;
        if ((line & 3u) == 1u && end - pos >= (size_t)kMinSeqSegment) {      // a sequence line worth a block of its own
            if (pos != cuts.back()) cuts.push_back(pos);
            cuts.push_back(end);
        }
        pos = end;
        ++line;
    }
    if (cuts.back() != n) cuts.push_back(n);
}

// `text` -> one complete gzip member in `out` (resized to fit).
inline void gzip_fastq_member(const char *text, size_t n, std::string &out) {
    static thread_local std::vector<size_t> cuts;
    fastq_segments(text, n, cuts);
    size_t bound = 10 + 8 + 16;
    for (size_t k = 0; k + 1 < cuts.size(); ++k) bound += block_bound(cuts[k + 1] - cuts[k]);
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
        for (size_t k = 0; k + 1 < cuts.size(); ++k)
            put_literal_block(bw, (const uint8_t *)text + cuts[k], cuts[k + 1] - cuts[k], k + 2 == cuts.size());
    }
    bw.align_byte();
    uint8_t *p = bw.p;
    const uint32_t crc = crc32_of((const uint8_t *)text, n);
    const uint32_t isize = (uint32_t)n;
    memcpy(p, &crc, 4); memcpy(p + 4, &isize, 4);
    out.assign((const char *)o, (size_t)(p + 8 - o));
}

}  // namespace fqz
