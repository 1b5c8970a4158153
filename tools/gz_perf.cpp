// gz_perf.cpp -- what bounds `.gz` output (read_assignment.py:244-245: one gzip member per batch, level 9): zlib's
// deflate on FASTQ text.  Host only.  Builds ~1000-read members of demultiplexed-FASTQ-shaped text (header with barcode /
// UMI tags, uniform ACGT sequence, nanopore-like qualities) and times deflate at several levels / match-chain limits
// / strategies, on 1..T threads.     g++ -O2 -pthread tools/gz_perf.cpp -lz -o /tmp/gz_perf && /tmp/gz_perf [threads...]
#include <zlib.h>
#include "../blaze_amd/csrc/fastq_deflate.h"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

static std::string make_member(int reads, uint64_t seed) {
    std::mt19937_64 rng(seed);
    std::lognormal_distribution<double> len(std::log(900.0), 0.6);
    std::normal_distribution<double> q(18.0, 7.0);
    std::string t;
    const char *acgt = "ACGT";
    for (int r = 0; r < reads; ++r) {
        char bc[17], umi[13], id[37];
        for (int i = 0; i < 16; ++i) bc[i] = acgt[rng() & 3]; bc[16] = 0;
        for (int i = 0; i < 12; ++i) umi[i] = acgt[rng() & 3]; umi[12] = 0;
        for (int i = 0; i < 36; ++i) id[i] = "0123456789abcdef-"[(i == 8 || i == 13 || i == 18 || i == 23) ? 16 : rng() & 15]; id[36] = 0;
        const int L = std::max(150, (int)len(rng));
        t += "@"; t += bc; t += "_"; t += umi; t += "#"; t += id; t += "_+\tCB:Z:"; t += bc; t += "\tUB:Z:"; t += umi; t += "\n";
        for (int i = 0; i < L; ++i) t.push_back(acgt[rng() & 3]);
        t += "\n+\n";
        double cur = q(rng);
        for (int i = 0; i < L; ++i) {                       // qualities wander (auto-correlated), as basecaller output does
            cur = 0.8 * cur + 0.2 * q(rng) + ((rng() & 7) == 0 ? -6.0 : 0.5);
            t.push_back((char)(33 + std::min(60, std::max(1, (int)cur))));
        }
        t += "\n";
    }
    return t;
}

struct Cfg { const char *name; int level, chain, strategy; };

static bool roundtrip_ok(const std::string &gzipped, const std::string &want) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    inflateInit2(&zs, 15 + 16);
    std::string got(want.size() + 64, 0);
    zs.next_in = (Bytef *)gzipped.data(); zs.avail_in = (uInt)gzipped.size();
    zs.next_out = (Bytef *)&got[0]; zs.avail_out = (uInt)got.size();
    const int rc = inflate(&zs, Z_FINISH);
    const size_t n = zs.total_out;
    inflateEnd(&zs);
    return rc == Z_STREAM_END && n == want.size() && memcmp(got.data(), want.data(), n) == 0;
}

static size_t gz(const std::string &in, std::string &out, const Cfg &c) {
    if (c.level < 0) {                              // the FASTQ-aware block encoder (blaze_amd/csrc/fastq_deflate.h)
        fqz::gzip_fastq_member(in.data(), in.size(), out);
        return out.size();
    }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, c.level, Z_DEFLATED, 15 + 16, 8, c.strategy);
    if (c.chain > 0) deflateTune(&zs, 32, 258, 258, c.chain);
    out.resize(deflateBound(&zs, in.size()) + 32);
    zs.next_in = (Bytef *)in.data(); zs.avail_in = (uInt)in.size();
    zs.next_out = (Bytef *)out.data(); zs.avail_out = (uInt)out.size();
    deflate(&zs, Z_FINISH);
    const size_t n = zs.total_out;
    deflateEnd(&zs);
    return n;
}

int main(int argc, char **argv) {
    std::vector<int> threads;
    for (int i = 1; i < argc; ++i) threads.push_back(atoi(argv[i]));
    if (threads.empty()) threads = {1, 8, 16};
    const int n_members = 64;
    std::vector<std::string> members;
    size_t total = 0;
    for (int m = 0; m < n_members; ++m) { members.push_back(make_member(1000, 1000 + m)); total += members.back().size(); }
    printf("%d members of 1000 reads, %.1f MB of FASTQ text\n", n_members, total / 1e6);
    {
        std::string o;
        int bad = 0;
        for (const std::string &m : members) { fqz::gzip_fastq_member(m.data(), m.size(), o); bad += !roundtrip_ok(o, m); }
        const std::string odd[] = {"", "\n", "A", "@x\nACGT\n+\nIIII\n", std::string(100000, 'A'), std::string("@r\n") + std::string(70000, 'N') + "\n+\n" + std::string(70000, '#') + "\n"};
        for (const std::string &m : odd) { fqz::gzip_fastq_member(m.data(), m.size(), o); bad += !roundtrip_ok(o, m); }
        printf("fastq blocks: zlib inflates every member back to its text: %s\n", bad ? "NO" : "yes");
        if (bad) return 1;
    }
    const Cfg cfgs[] = {{"level 9 (reference)", 9, 0, Z_DEFAULT_STRATEGY}, {"level 9, chain 256", 9, 256, Z_DEFAULT_STRATEGY},
                        {"level 9, chain 64", 9, 64, Z_DEFAULT_STRATEGY}, {"level 9, chain 16", 9, 16, Z_DEFAULT_STRATEGY},
                        {"level 6", 6, 0, Z_DEFAULT_STRATEGY}, {"level 5", 5, 0, Z_DEFAULT_STRATEGY}, {"level 4", 4, 0, Z_DEFAULT_STRATEGY},
                        {"level 3", 3, 0, Z_DEFAULT_STRATEGY}, {"level 2", 2, 0, Z_DEFAULT_STRATEGY}, {"level 1", 1, 0, Z_DEFAULT_STRATEGY},
                        {"huffman only", 6, 0, Z_HUFFMAN_ONLY}, {"fastq blocks (ours)", -1, 0, 0}, {"level 6 filtered", 6, 0, Z_FILTERED}};
    for (const Cfg &c : cfgs) {
        for (int T : threads) {
            std::vector<size_t> outb(T, 0);
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] { std::string o; for (int m = t; m < n_members; m += T) outb[t] += gz(members[m], o, c); });
            for (auto &x : th) x.join();
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            size_t ob = 0; for (size_t v : outb) ob += v;
            printf("%-22s threads %2d: %7.1f MB/s in (%6.1f per thread), ratio %.3f (%.2f bits/char), %.0f K reads/s\n", c.name, T, total / 1e6 / s,
                   total / 1e6 / s / T, (double)ob / total, 8.0 * ob / total, n_members * 1000.0 / s / 1e3);
        }
    }
    return 0;
}
