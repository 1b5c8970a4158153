// host_asan.cpp -- AddressSanitizer / UBSan harness for the host side of the file pipelines (host only, no HIP):
// the lock-step reader, parse_batch, format_batch (plain and .gz, restrand on / off) and the stage-1 reader / prepare /
// copy / format, fed with quirky inputs (CRLF, no final newline, quoted and NA cells, blank rows, empty and one-letter
// reads, titles with tabs) and with MADE-UP device results that cover every value the device can return (any whitelist
// index, any end index, every flag; stage 1: any start, either strand).  Gzipped inputs on purpose: their lines live on
// the heap (the chunk's arena), where the sanitizer sees every byte read past a line.
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -std=c++17 -pthread tools/host_asan.cpp -lz -ldl -o /tmp/host_asan && /tmp/host_asan <scratch dir>
#include "../blaze_amd/csrc/host_io.h"
#include <cstdio>
#include <cstdlib>
#include <random>

using namespace bzio;

static std::string rnd_seq(std::mt19937 &rng, int n, bool odd) {
    std::string s(n, 'A');
    for (int i = 0; i < n; ++i) s[i] = odd && rng() % 11 == 0 ? "NnacgtRY-.*"[rng() % 11] : "ACGT"[rng() & 3];
    return s;
}

int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp/host_asan_d";
    if (system(("mkdir -p " + dir).c_str()) != 0) return 2;
    std::mt19937 rng(12345);
    long total_out = 0;
    for (int variant = 0; variant < 6; ++variant) {
        const bool crlf = variant == 1, gz_in = variant != 2, no_final_nl = variant == 3;
        const std::string nl = crlf ? "\r\n" : "\n";
        // ---- inputs
        const int n_files = 3, per_file[3] = {257, 0, 1200};
        std::vector<std::string> fns;
        std::string csv = "read_id,putative_bc,putative_bc_min_q,putative_umi,polyT_end,pre_bc_flanking,post_umi_flanking" + nl;
        int rid = 0;
        for (int f = 0; f < n_files; ++f) {
            std::string p = dir + "/v" + std::to_string(variant) + "_" + std::to_string(f) + ".fastq";
            std::string body;
            for (int r = 0; r < per_file[f]; ++r, ++rid) {
                const int L = rid % 97 == 0 ? 0 : rid % 89 == 0 ? 1 : 20 + (int)(rng() % 400);
                const std::string id = "read" + std::to_string(rid) + (rid % 13 == 0 ? "@" : "");
                body += "@" + id + (rid % 7 == 0 ? "\trunid=1 x" : " ch=2") + nl + rnd_seq(rng, L, rid % 5 == 0) + nl + "+" + (rid % 9 == 0 ? id : "") + nl;
                for (int i = 0; i < L; ++i) body.push_back((char)(33 + rng() % 60));
                body += nl;
                // the csv row of this read
                std::string bc = rnd_seq(rng, rid % 53 == 0 ? 29 + (int)(rng() % 32) : rid % 31 == 0 ? 14 : 16, rid % 17 == 0), umi = rnd_seq(rng, rid % 23 == 0 ? 1 : 12, false);
                std::string pre = rid % 19 == 0 ? "NA" : rid % 29 == 0 ? "" : rnd_seq(rng, 5, false), post = rid % 37 == 0 ? "null" : rnd_seq(rng, rid % 41 == 0 ? 2 : 5, false);
                char pt[32];
                const int L1 = L > 0 ? L : 1;
                snprintf(pt, sizeof pt, rid % 3 == 0 ? "%d.0" : rid % 3 == 1 ? "%d" : "%d.7", (rid & 1 ? 1 : -1) * (int)(rng() % (2 * L1 + 5)));
                std::string idc = id;
                while (!idc.empty() && idc.back() == '@') idc.pop_back();
                if (rid % 6 == 0) csv += idc + ",,,,,," + nl;                     // blank row
                else csv += idc + "," + (rid % 43 == 0 ? "\"" + bc + "\"" : bc) + ",15," + umi + "," + (rid % 47 == 0 ? "0" : pt) + "," + pre + "," + post + nl;
            }
            if (no_final_nl && body.size() >= nl.size()) body.resize(body.size() - nl.size());
            FILE *fo = fopen(p.c_str(), "wb");
            fwrite(body.data(), 1, body.size(), fo);
            fclose(fo);
            if (gz_in) { if (system(("gzip -1 -f " + p).c_str()) != 0) return 2; p += ".gz"; }
            fns.push_back(p);
        }
        if (no_final_nl) csv.resize(csv.size() - nl.size());
        const std::string csv_path = dir + "/v" + std::to_string(variant) + ".csv" + (variant == 4 ? ".gz" : "");
        {
            const std::string plain = dir + "/v" + std::to_string(variant) + ".csv";
            FILE *fo = fopen(plain.c_str(), "wb");
            fwrite(csv.data(), 1, csv.size(), fo);
            fclose(fo);
            if (variant == 4 && system(("gzip -1 -f " + plain).c_str()) != 0) return 2;
        }
        // a whitelist of 50 entries, some irregular
        std::vector<char> wl_bytes;
        std::vector<int64_t> wl_off{0};
        for (int k = 0; k < 50; ++k) {
            const std::string w = k == 7 ? "" : rnd_seq(rng, k == 9 ? 18 : 16, k == 11);
            wl_bytes.insert(wl_bytes.end(), w.begin(), w.end());
            wl_off.push_back((int64_t)wl_bytes.size());
        }
        // ---- step 3 host side
        for (int mode = 0; mode < 4; ++mode) {
            Job job;
            job.fastq_fns = fns; job.csv = csv_path; job.gz = mode & 1; job.restrand = !(mode & 2); job.batch_size = mode == 3 ? 7 : 100;
            job.n_threads = 4; job.device_batch_rows = 300; job.gz_level = mode == 3 ? 1 : 0;
            Pool pool(job.n_threads);
            PoolScope scope(&pool);
            ChunkReader rd(job);
            while (auto ch = rd.next()) {
                const size_t n = ch->reads.size();
                ch->bc.resize(n); ch->umi.resize(n); ch->post.resize(n); ch->rid.resize(n); ch->polyT.resize(n);
                ch->ext.resize(n * kExtStride); ch->ext_len.resize(n); ch->bc_off.resize(n); ch->bc_len.resize(n);
                Chunk &c = *ch;
                const int nb = (int)c.batch_start.size() - 1;
                parallel_for(nb, job.n_threads, [&](int b) { parse_batch(c, b); });
                if (!c.wide.empty()) {                         // barcode cells wider than the 32-byte row: the chunk again at 64
                    widen_rows(c);
                    for (size_t r = 0; r < n; ++r) total_out += c.ext_wide[r * kExtWide + (c.ext_len[r] ? c.ext_len[r] - 1 : 0)];
                }
                c.hit.resize(n); c.ed.resize(n); c.end_idx.resize(n); c.flags.resize(n);
                for (size_t r = 0; r < n; ++r) {               // any answer the device may give
                    c.hit[r] = (int32_t)(rng() % 52) - 2;
                    if (c.hit[r] >= 50) c.hit[r] = 49;
                    c.ed[r] = (int8_t)(rng() % 4);
                    c.end_idx[r] = (int8_t)((int)(rng() % 34) - 1);
                    c.flags[r] = (uint8_t)(rng() % 8 == 0 ? 1 : 0);
                    if (c.bc_len[r] == 0) { c.hit[r] = -1; c.flags[r] = 2; }
                }
                c.out.resize(nb); c.demul.assign(nb, 0);
                parallel_for(nb, job.n_threads, [&](int b) { format_batch(c, b, job, wl_bytes, wl_off); });
                for (auto &o : c.out) total_out += (long)o.size();
            }
        }
        // ---- what a rank of a sharded run asks of its files: read counts, first ids, where a read's row starts in the table
        {
            int64_t reads = 0;
            for (const std::string &f : fns) reads += count_fastq_reads(f);
            if (reads != 257 + 1200) { printf("count_fastq_reads: %lld\n", (long long)reads); return 1; }
            const std::string first = first_fastq_id(fns[2]);
            if (first != "read257") { printf("first_fastq_id: %s\n", first.c_str()); return 1; }
            if (!first_fastq_id(fns[1]).empty()) return 1;
            if (variant != 4)                                       // a gzipped table cannot be searched: -1
                for (double hint : {0.0, 0.17, 0.5, 1.0}) {
                    const int64_t at = csv_locate_row(csv_path, first, hint);
                    if (at <= 0 || csv.compare((size_t)at, first.size() + 1, first + ",") != 0) { printf("csv_locate_row(%g) = %lld\n", hint, (long long)at); return 1; }
                    if (csv_locate_row(csv_path, "no_such_read", hint) != -1) return 1;
                }
            else if (csv_locate_row(csv_path, first, 0.5) != -1) return 1;
        }
        // ---- stage 1 host side
        for (int mode = 0; mode < 2; ++mode) {
            S1Job job;
            job.fastq_fns = fns; job.batch_size = mode ? 3 : 500; job.n_threads = 4; job.device_batch_reads = 400; job.kit5 = mode; job.umi_len = mode ? 10 : 12;
            Pool pool(job.n_threads);
            PoolScope scope(&pool);
            FastqChunkReader rd(job);
            bool first = true;
            while (auto ch = rd.next()) {
                S1Chunk &c = *ch;
                c.first = first; first = false;
                s1_prepare(c, job.n_threads);
                const size_t n = c.reads.size();
                std::vector<uint8_t> seq((size_t)c.off[n] + 1), qual((size_t)c.off[n] + 1);
                std::vector<int64_t> off(n + 1);
                s1_copy(c, seq.data(), qual.data(), off.data(), job.n_threads);
                c.code.resize(n); c.start.resize(n); c.minq.resize(n); c.polyt.resize(n); c.strand.resize(n);
                for (size_t r = 0; r < n; ++r) {
                    const int L = (int)c.reads[r].seq.len;
                    c.code[r] = rng() % 4 == 0 ? (int)(rng() % 3 == 0 ? 1 : rng() % 2 ? 2 : 10) : 0;
                    c.start[r] = (int32_t)(rng() % (unsigned)(L + 12)) - 3;
                    c.minq[r] = (int32_t)(rng() % 40);
                    c.polyt[r] = (int32_t)(rng() % (unsigned)(L + 40));
                    c.strand[r] = rng() & 1 ? '+' : '-';
                }
                const int nb = (int)c.batch_start.size() - 1;
                c.out.resize(nb); c.pass.resize(nb);
                parallel_for(nb, job.n_threads, [&](int b) { s1_format_batch(c, b, job); });
                for (auto &o : c.out) total_out += (long)o.size();
            }
        }
    }
    // ---- the row locator and the counters on the same files
    printf("bytes formatted %ld\n", total_out);
    return 0;
}
