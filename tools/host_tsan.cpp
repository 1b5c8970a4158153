// host_tsan.cpp -- ThreadSanitizer harness for the host-side threading of the file pipelines (host only, no HIP):
// the job-long Pool behind parallel_for (many short loops, exceptions), the member encoder's thread_local buffers under
// it, and the FastqFiles pool that inflates gzipped inputs ahead of the reader with a one-megabyte budget.
//   g++ -O1 -g -fsanitize=thread -std=c++17 -pthread tools/host_tsan.cpp -lz -ldl -o /tmp/host_tsan && /tmp/host_tsan <scratch dir>
// tests/test_host_cpu.py builds and runs it; any ThreadSanitizer report fails the test.
#include "../blaze_amd/csrc/host_io.h"
#include <cstdio>
#include <cstdlib>
int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp/host_tsan_fq";
    if (system(("mkdir -p " + dir).c_str()) != 0) return 2;
    using namespace bzio;
    // 1. Pool: many short loops, nested scopes, exceptions
    {
        Pool pool(8);
        PoolScope sc(&pool);
        std::atomic<long> sum{0};
        for (int rep = 0; rep < 2000; ++rep) {
            parallel_for(rep % 37, 8, [&](int i) { sum += i; });
        }
        int caught = 0;
        for (int rep = 0; rep < 50; ++rep) {
            try { parallel_for(100, 8, [&](int i) { if (i == 57) throw std::runtime_error("x"); sum += 1; }); } catch (const std::exception &) { ++caught; }
        }
        printf("pool sum %ld caught %d\n", sum.load(), caught);
        // thread_local text reuse through format-like work
        std::vector<std::string> outs(64);
        for (int rep = 0; rep < 20; ++rep)
            parallel_for(64, 8, [&](int i) {
                std::string t;
                for (int k = 0; k < 50; ++k) { t += "@r\n"; t.append(200 + i, "ACGT"[(i + k) & 3]); t += "\n+\n"; t.append(200 + i, 'I'); t += "\n"; }
                fqz::gzip_fastq_member(t.data(), t.size(), outs[i]);
            });
        size_t z = 0; for (auto &o : outs) z += o.size();
        printf("members %zu bytes\n", z);
    }
    // 1b. the ordered reader -> submit -> finish -> writer frame with a pretend device: output in order, and every way of failing
    {
        struct Ck { int k = 0; std::vector<int> v; };
        for (int fail_at = -1; fail_at < 4; ++fail_at) {          // -1 = none; 0 reader, 1 submit, 2 finish, 3 unwritable output
            Pool pool(4);
            PoolScope sc(&pool);
            OrderedPipeline<Ck> pipe;
            pipe.out_path = fail_at == 3 ? dir + "/no/such/dir/out.txt" : dir + "/pipe_out.txt";
            pipe.create_output_first = fail_at != 3;
            pipe.max_inflight = 3;
            std::atomic<int> abandoned{0};
            pipe.open_reader = [&]() {
                auto k = std::make_shared<int>(0);
                return std::function<std::unique_ptr<Ck>()>([k, fail_at]() -> std::unique_ptr<Ck> {
                    if (*k == 40) return nullptr;
                    if (fail_at == 0 && *k == 17) throw std::runtime_error("reader");
                    auto c = std::make_unique<Ck>();
                    c->k = (*k)++;
                    c->v.assign(100, c->k);
                    return c;
                });
            };
            pipe.submit = [&](Ck &c, int64_t k) { if (fail_at == 1 && k == 9) throw std::runtime_error("submit"); parallel_for(8, 4, [&](int i) { c.v[i] += 1; }); };
            pipe.finish = [&](Ck &c, std::vector<std::string> &parts) {
                if (fail_at == 2 && c.k == 21) throw MismatchError("finish");
                parts.assign(3, "");
                parallel_for(3, 4, [&](int i) { parts[i] = std::to_string(c.k) + "." + std::to_string(i) + "\n"; });
            };
            pipe.abandon = [&](Ck &) { ++abandoned; };
            std::string what = "ok";
            try { pipe.run(); } catch (const MismatchError &e) { what = std::string("mismatch:") + e.what(); } catch (const std::exception &e) { what = e.what(); }
            std::string got;
            if (FILE *f = fopen((dir + "/pipe_out.txt").c_str(), "rb")) { char buf[4096]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) got.append(buf, n); fclose(f); }
            std::string want;
            if (fail_at == -1) for (int k = 0; k < 40; ++k) for (int i = 0; i < 3; ++i) want += std::to_string(k) + "." + std::to_string(i) + "\n";
            const bool ok = fail_at == -1 ? (what == "ok" && got == want && pipe.n_chunks == 40)
                                          : (what != "ok" && (fail_at != 2 || what == "mismatch:finish"));
            printf("pipeline fail_at %d: %s, abandoned %d, %s\n", fail_at, what.substr(0, 40).c_str(), abandoned.load(), ok ? "as expected" : "WRONG");
            if (!ok) return 1;
        }
    }
    // 2. FastqFiles prefetch pool with a tight budget
    {
        std::vector<std::string> paths;
        if (system(("mkdir -p " + dir).c_str()) != 0) return 2;
        for (int f = 0; f < 24; ++f) {
            std::string p = dir + "/r" + std::to_string(f) + ".fastq";
            FILE *fo = fopen(p.c_str(), "w");
            for (int r = 0; r < (f % 5 == 3 ? 0 : 300 + 50 * f); ++r) fprintf(fo, "@f%d_%d x\nACGTACGTACGTACGTACGTTTGACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIII\n", f, r);
            fclose(fo);
            if (f % 4 != 1) { std::string cmd = "gzip -1 -f " + p; if (system(cmd.c_str()) != 0) return 2; p += ".gz"; }
            paths.push_back(p);
        }
        setenv("BLAZE_AMD_GZ_PREFETCH_MB", "1", 1);
        long lines = 0;
        {
            FastqFiles files(paths, 6);
            std::vector<std::shared_ptr<LineSource>> keep;
            for (size_t k = 0; k < paths.size(); ++k) {
                auto src = files.open(k);
                Arena a; Span s;
                while (src->next(a, s)) ++lines;
                keep.push_back(src);
                if (keep.size() > 3) keep.erase(keep.begin());
            }
        }
        printf("lines %ld\n", lines);
    }
    return 0;
}
