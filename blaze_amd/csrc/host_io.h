// host_io.h -- native host side of step 3 around the device scan: bz_assign_files().
//
// What it replaces (file:line under /root/reference/blaze):
//   _read_and_bc_batch_generator_with_idx   read_assignment.py:138-181  lock-step FASTQ + putative_bc.csv batches
//   helper.fastq_parser / batch_iterator     helper.py:275-301           4-line records, batches per input file
//   _assign_read_batches, second loop        read_assignment.py:214-247  id check, trim, restrand, record, gzip member
//   _match_bc_row epilogue                   read_assignment.py:51-62,87-99  strand, shortcut, UMI start correction
//   assign_read's ordered byte concatenation read_assignment.py:293-327
// The per-batch `pd.read_csv` of :160-164 is restated for the seven columns of main.py:96-105:
//   * string cells: pandas' default NA tokens ('', 'NA', 'NaN', 'null', ... ) become NaN and `fillna('')` (:205) turns
//     them into '' -- so a 2-nt flank spelled "NA" reads as empty in the reference, and does here;
//   * polyT_end: float64/int64 in pandas, only truth-tested, sign-tested and int()-ed (:51-56, :227-232);
//   * a batch whose read_id cells are ALL numeric is inferred as a numeric column by pandas and then fails the
//     reference's `bc.read_id != r.id` check (:220) -- reproduced as the same "different order" error.
// Files are read as bytes; Python's universal-newline translation is reproduced for "\r\n" (a lone '\r' as a line
// terminator is not).  Output bytes equal the reference's for plain FASTQ; for .gz the DECOMPRESSED stream is equal and
// the member boundaries are the reference's (one member per --batch-size reads per input file), the compressed bytes
// themselves are not reproducible even in the reference (gzip header mtime).
#pragma once
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <dlfcn.h>
#include <errno.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include "fastq_deflate.h"

#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace bzio {

struct Span { char *p = nullptr; uint32_t len = 0; };          // bytes live in a file mapping or in the chunk's arena

inline bool py_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }

// ---- stable storage for lines that have to be copied (gzip input): blocks never move ----------------------------
// "cannot open <path>[ for writing] [errno N: text]": the Python layer raises the OSError subclass of N (the reference's own
// open() calls raise FileNotFoundError / IsADirectoryError / PermissionError for the same paths)
inline std::runtime_error open_error(const std::string &what, int err) {
    return std::runtime_error(what + " [errno " + std::to_string(err) + ": " + strerror(err) + "]");
}

class Arena {
   public:
    char *alloc(size_t n) {
#ifdef BZ_ARENA_EXACT                                  // sanitizer builds (tools/host_asan.cpp): every line its own heap block, so a byte read past it is seen
        blocks_.emplace_back(new char[std::max<size_t>(n, 1)]);
        return blocks_.back().get();
#endif
        if (n > room_) {
            const size_t sz = std::max<size_t>(n, (size_t)4 << 20);
            blocks_.emplace_back(new char[sz]);
            cur_ = blocks_.back().get(); room_ = sz;
        }
        char *p = cur_;
        cur_ += n; room_ -= n;
        return p;
    }
    void give_back(size_t n) {                                   // undo the tail of the last alloc
#ifdef BZ_ARENA_EXACT
        (void)n;
        return;
#endif
        cur_ -= n; room_ += n;
    }
   private:
    std::vector<std::unique_ptr<char[]>> blocks_;
    char *cur_ = nullptr;
    size_t room_ = 0;
};

// ---- sequential line sources: next line without its terminator ("\n" or "\r\n"); false at end of file ----------
class LineSource {
   public:
    virtual ~LineSource() {}
    virtual bool next(Arena &arena, Span &out) = 0;
};

// plain file: private writable mapping (quoted csv cells are unescaped in place), lines are NOT copied; the mapping
// lives as long as the source object, which the job keeps until its last chunk is written
class MmapLines : public LineSource {
   public:
    explicit MmapLines(const std::string &path, size_t start = 0) {
        fd_ = open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw open_error("cannot open " + path, errno);
        struct stat st;
        if (fstat(fd_, &st) != 0) { const int e = errno; close(fd_); fd_ = -1; throw open_error("cannot map " + path, e); }
        if (!S_ISREG(st.st_mode)) { close(fd_); fd_ = -1; throw open_error("cannot map " + path, S_ISDIR(st.st_mode) ? EISDIR : EINVAL); }
        size_ = (size_t)st.st_size;
        if (size_) {
            void *m = mmap(nullptr, size_, PROT_READ | PROT_WRITE, MAP_PRIVATE, fd_, 0);
            if (m == MAP_FAILED) { const int e = errno; close(fd_); fd_ = -1; throw open_error("cannot map " + path, e); }
            base_ = (char *)m;
            start_ = std::min(start, size_);                      // a rank of a sharded run starts at ITS first row of the table
            at_.store(start_, std::memory_order_relaxed);
            madvise(base_, size_, MADV_SEQUENTIAL);
            // The scan below runs at memchr speed only over pages that are already mapped: taking the page faults itself
            // (one per 64 KB of page cache) was most of the reader thread's time.  A helper maps the pages ahead of the
            // scan, kPopulateLead bytes at most, in kPopulateStep pieces (MADV_POPULATE_READ; touching a byte per page
            // where the kernel does not know it).
            if (size_ >= (size_t)4 * kPopulateStep) populator_ = std::thread([this] { populate(); });
        }
    }
    ~MmapLines() override {
        stop_.store(true, std::memory_order_relaxed);
        if (populator_.joinable()) populator_.join();
        if (base_) munmap(base_, size_);
        if (fd_ >= 0) close(fd_);
    }
    bool next(Arena &, Span &out) override {
        const size_t pos_ = at_.load(std::memory_order_relaxed);
        if (pos_ >= size_) return false;
        char *p = base_ + pos_;
        char *nl = (char *)memchr(p, '\n', size_ - pos_);
        size_t len = nl ? (size_t)(nl - p) : size_ - pos_;
        at_.store(pos_ + len + (nl ? 1 : 0), std::memory_order_relaxed);
        if (len && p[len - 1] == '\r') --len;
        if (len > 0xFFFFFFFFu) throw std::runtime_error("line longer than 4 GiB");
        out.p = p; out.len = (uint32_t)len;
        return true;
    }
   private:
    static constexpr size_t kPopulateStep = (size_t)8 << 20, kPopulateLead = (size_t)256 << 20;
    void populate() {
        bool by_madvise = true;
        for (size_t off = start_ / kPopulateStep * kPopulateStep; off < size_ && !stop_.load(std::memory_order_relaxed); off += kPopulateStep) {
            while (off > at_.load(std::memory_order_relaxed) + kPopulateLead) {          // far enough ahead of the scan
                if (stop_.load(std::memory_order_relaxed)) return;
                usleep(200);
            }
            const size_t n = std::min(kPopulateStep, size_ - off);
            if (by_madvise && madvise(base_ + off, n, 22 /* MADV_POPULATE_READ, Linux 5.14 */) != 0) by_madvise = false;
            if (!by_madvise) {
                volatile const char *q = base_ + off;
                for (size_t k = 0; k < n; k += 4096) (void)q[k];
            }
        }
    }
    int fd_ = -1;
    char *base_ = nullptr;
    size_t size_ = 0, start_ = 0;
    std::atomic<size_t> at_{0};                                   // scan position (the helper stays kPopulateLead ahead at most)
    std::atomic<bool> stop_{false};
    std::thread populator_;
};

// gzip (or anything zlib's gzread reads): lines are copied into the chunk's arena
class GzLines : public LineSource {
   public:
    explicit GzLines(const std::string &path) : path_(path) {
        f_ = gzopen(path.c_str(), "rb");
        if (!f_) throw open_error("cannot open " + path, errno ? errno : ENOENT);
        gzbuffer(f_, 1 << 20);
        buf_.resize(8 << 20);
    }
    ~GzLines() override { if (f_) gzclose(f_); }
    bool next(Arena &arena, Span &out) override {
        tmp_.clear();
        bool any = false;
        for (;;) {
            if (pos_ == end_) {
                if (eof_) break;
                int got = gzread(f_, buf_.data(), (unsigned)buf_.size());
                if (got < 0) throw std::runtime_error("read error in " + path_);
                if (got == 0) { eof_ = true; break; }
                pos_ = 0; end_ = (size_t)got;
            }
            const char *p = buf_.data() + pos_;
            const char *nl = (const char *)memchr(p, '\n', end_ - pos_);
            any = true;
            if (nl) {
                tmp_.insert(tmp_.end(), p, nl);
                pos_ = (size_t)(nl - buf_.data()) + 1;
                if (!tmp_.empty() && tmp_.back() == '\r') tmp_.pop_back();
                return emit(arena, out);
            }
            tmp_.insert(tmp_.end(), p, (const char *)buf_.data() + end_);
            pos_ = end_;
        }
        if (!any) return false;
        return emit(arena, out);                                  // last line without a newline
    }
   private:
    bool emit(Arena &arena, Span &out) {
        if (tmp_.size() > 0xFFFFFFFFu) throw std::runtime_error("line longer than 4 GiB");
        out.p = arena.alloc(tmp_.size() + 1);
        memcpy(out.p, tmp_.data(), tmp_.size());
        out.len = (uint32_t)tmp_.size();
        return true;
    }
    std::string path_;
    gzFile f_ = nullptr;
    std::vector<char> buf_, tmp_;
    size_t pos_ = 0, end_ = 0;
    bool eof_ = false;
};

// a whole decompressed file in memory: lines are NOT copied (as with a mapping); the buffer lives as long as the source
// object, which every chunk that holds lines of it keeps alive
class MemLines : public LineSource {
   public:
    MemLines(std::unique_ptr<char[]> &&buf, size_t size, std::function<void()> on_free)
        : buf_(std::move(buf)), size_(size), on_free_(std::move(on_free)) {}
    ~MemLines() override { buf_.reset(); if (on_free_) on_free_(); }
    bool next(Arena &, Span &out) override {
        if (pos_ >= size_) return false;
        char *p = buf_.get() + pos_;
        char *nl = (char *)memchr(p, '\n', size_ - pos_);
        size_t len = nl ? (size_t)(nl - p) : size_ - pos_;
        pos_ += len + (nl ? 1 : 0);
        if (len && p[len - 1] == '\r') --len;
        if (len > 0xFFFFFFFFu) throw std::runtime_error("line longer than 4 GiB");
        out.p = p; out.len = (uint32_t)len;
        return true;
    }
   private:
    std::unique_ptr<char[]> buf_;
    size_t size_ = 0, pos_ = 0;
    std::function<void()> on_free_;
};

inline bool ends_with_gz(const std::string &p) { return p.size() >= 3 && p.compare(p.size() - 3, 3, ".gz") == 0; }
inline std::shared_ptr<LineSource> open_lines(const std::string &path) {      // '.gz' decides, as read_assignment.py:151
    if (ends_with_gz(path)) return std::make_shared<GzLines>(path);
    struct stat st;
    if (stat(path.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) throw open_error("cannot open " + path, EISDIR);   // Python's open(): IsADirectoryError
    if (stat(path.c_str(), &st) == 0 && !S_ISREG(st.st_mode)) return std::make_shared<GzLines>(path);   // pipes etc.: streamed
    return std::make_shared<MmapLines>(path);
}

// ---- the job's FASTQ files, opened in order; gzipped ones are inflated AHEAD of the reader by a pool of threads --------
// A nanopore run is thousands of `.fastq.gz` files of a few thousand reads each (the reference opens them one after
// another with gzip.open, read_assignment.py:151-153, main.py:311-315).  zlib inflates ~0.3-0.5 GB/s per thread, a
// twentieth of what the reader's line scan does on a mapping, so with one reader thread gzipped input ran at a twentieth
// of the plain rate.  Whole files are independent: `threads` workers inflate the next files into memory while the reader
// consumes the current one; files are handed out strictly in order; at most `budget` bytes of inflated text (counted
// from when a file is complete until the last chunk holding its lines is written) and `threads + 2` files are ahead.
// A gzipped file larger than kStreamAbove compressed bytes is streamed by the reader itself (GzLines) as before: it
// may not fit in memory.  BLAZE_AMD_GZ_PREFETCH=0 turns the pool off.
class FastqFiles {
   public:
    static constexpr int64_t kStreamAbove = (int64_t)512 << 20;
    FastqFiles(const std::vector<std::string> &paths, int threads) : paths_(paths), slots_(paths.size()) {
        const char *e = getenv("BLAZE_AMD_GZ_PREFETCH");
        const bool enabled = !(e && e[0] == '0');
        for (size_t k = 0; k < paths_.size(); ++k) {
            struct stat st;
            slots_[k].prefetch = enabled && ends_with_gz(paths_[k]) && stat(paths_[k].c_str(), &st) == 0 && S_ISREG(st.st_mode) &&
                                 (int64_t)st.st_size <= kStreamAbove;
            // what the file will hold once inflated, as far as its size tells (FASTQ deflates ~4x): reserved against the budget
            // from the moment a worker takes the file, corrected when it is done -- a budget that only saw finished files let
            // every worker start a 512 MiB file at once and hold tens of GB before the first one was counted
            if (slots_[k].prefetch) slots_[k].est = (int64_t)st.st_size * 4;
            n_prefetch_ += slots_[k].prefetch;
        }
        if (const char *b = getenv("BLAZE_AMD_GZ_PREFETCH_MB")) budget_ = std::max<int64_t>(1, atoll(b)) << 20;
        const int nt = n_prefetch_ ? std::max(1, std::min<int>(threads, (int)n_prefetch_)) : 0;
        ahead_ = (size_t)nt + 2;
        for (int t = 0; t < nt; ++t) workers_.emplace_back([this] { work(); });
    }
    ~FastqFiles() {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    size_t size() const { return paths_.size(); }
    // the k-th file's lines; k must go up by one per call
    std::shared_ptr<LineSource> open(size_t k) {
        Slot &sl = slots_[k];
        if (!sl.prefetch) {
            { std::lock_guard<std::mutex> g(mu_); next_needed_ = k + 1; }
            cv_.notify_all();
            return open_lines(paths_[k]);
        }
        std::unique_lock<std::mutex> lk(mu_);
        next_needed_ = k;
        cv_.notify_all();
        cv_.wait(lk, [&] { return sl.done; });
        next_needed_ = k + 1;
        cv_.notify_all();
        if (sl.err) std::rethrow_exception(sl.err);
        const int64_t n = (int64_t)sl.size;
        auto state = state_;
        return std::make_shared<MemLines>(std::move(sl.buf), sl.size, [state, n] {
            { std::lock_guard<std::mutex> g(state->mu); state->bytes -= n; }
            state->cv.notify_all();
        });
    }

   private:
    struct Slot { bool prefetch = false, started = false, done = false; std::unique_ptr<char[]> buf; size_t size = 0; int64_t est = 0; std::exception_ptr err; };
    struct Shared { std::mutex mu; std::condition_variable cv; int64_t bytes = 0; };     // outlives this object: chunks free buffers later
    void work() {
        for (;;) {
            size_t k;
            {
                std::unique_lock<std::mutex> lk(mu_);
                for (;;) {
                    if (stop_) return;
                    k = scan_;
                    while (k < slots_.size() && (!slots_[k].prefetch || slots_[k].started)) ++k;
                    if (k >= slots_.size()) return;
                    // the file the reader is waiting for is always taken; others only inside the window and the budget
                    // (bytes = inflated text held + the estimates of the files being inflated right now)
                    bool room;
                    { std::lock_guard<std::mutex> g(state_->mu); room = state_->bytes + slots_[k].est <= budget_ || state_->bytes == 0; }
                    if (k <= next_needed_ || (k < next_needed_ + ahead_ && room)) break;
                    cv_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(20));   // (system clock: pthread_cond_timedwait, which ThreadSanitizer understands)
                }
                slots_[k].started = true;
                scan_ = k + 1;
                { std::lock_guard<std::mutex> g(state_->mu); state_->bytes += slots_[k].est; }
            }
            std::unique_ptr<char[]> buf;
            size_t size = 0;
            std::exception_ptr err;
            try {
                size = inflate_file(paths_[k], buf);
            } catch (...) {
                err = std::current_exception();
            }
            {
                std::lock_guard<std::mutex> g(mu_);
                { std::lock_guard<std::mutex> g2(state_->mu); state_->bytes += (int64_t)size - slots_[k].est; }   // the estimate becomes the real size
                slots_[k].buf = std::move(buf);
                slots_[k].size = size;
                slots_[k].err = err;
                slots_[k].done = true;
            }
            cv_.notify_all();
        }
    }
    // Whole-buffer inflate through the system's libdeflate when the host has one (dlopen, no header needed: four C symbols
    // with a stable ABI): 2-3x zlib's rate on gzipped FASTQ.  Member after member (`_ex` reports the input it consumed);
    // anything it does not take at face value (not gzip, trailing bytes, damaged data, > 2 GiB members) is left to the
    // zlib path below, whose behaviour is then what it always was.  BLAZE_AMD_LIBDEFLATE=0 turns it off.
    struct LibDeflate {
        void *(*alloc)() = nullptr;
        void (*release)(void *) = nullptr;
        int (*gunzip)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;
        static const LibDeflate *get() {
            static const LibDeflate *inst = []() -> const LibDeflate * {
                void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
                if (!h) return nullptr;
                static LibDeflate l;
                l.alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
                l.release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
                l.gunzip = (int (*)(void *, const void *, size_t, void *, size_t, size_t *, size_t *))dlsym(h, "libdeflate_gzip_decompress_ex");
                return (l.alloc && l.release && l.gunzip) ? &l : nullptr;
            }();
            return inst;
        }
    };
    static bool inflate_whole(const std::string &path, std::unique_ptr<char[]> &out, size_t &size) {
        const char *e = getenv("BLAZE_AMD_LIBDEFLATE");
        const LibDeflate *ld = (e && e[0] == '0') ? nullptr : LibDeflate::get();
        if (!ld) return false;
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size < 18) { close(fd); return false; }
        const size_t in_size = (size_t)st.st_size;
        void *m = mmap(nullptr, in_size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (m == MAP_FAILED) return false;
        const unsigned char *in = (const unsigned char *)m;
        void *d = ld->alloc();
        bool ok = d != nullptr;
        // ISIZE of the last member is the whole size for the usual one-member file; several members grow the buffer
        const size_t isize = (size_t)in[in_size - 4] | (size_t)in[in_size - 3] << 8 | (size_t)in[in_size - 2] << 16 | (size_t)in[in_size - 1] << 24;
        size_t cap = std::max<size_t>(isize, in_size * 3) + 4096, n = 0, pos = 0;
        std::unique_ptr<char[]> buf;
        if (ok) buf.reset(new char[cap]);
        while (ok && pos < in_size) {
            size_t used = 0, made = 0;
            const int rc = ld->gunzip(d, in + pos, in_size - pos, buf.get() + n, cap - n, &used, &made);
            if (rc == 3) {                                        // LIBDEFLATE_INSUFFICIENT_SPACE: the member starts over in a larger buffer
                std::unique_ptr<char[]> bigger(new char[cap * 2]);
                memcpy(bigger.get(), buf.get(), n);
                buf.swap(bigger);
                cap *= 2;
                continue;
            }
            if (rc != 0 || used == 0) { ok = false; break; }
            n += made; pos += used;
        }
        if (d) ld->release(d);
        munmap(m, in_size);
        if (!ok) return false;
        out = std::move(buf);
        size = n;
        return true;
    }
    static size_t inflate_file(const std::string &path, std::unique_ptr<char[]> &out) {
        size_t whole = 0;
        if (inflate_whole(path, out, whole)) return whole;
        gzFile f = gzopen(path.c_str(), "rb");                  // multi-member files and plain text behind a .gz name included
        if (!f) throw open_error("cannot open " + path, errno ? errno : ENOENT);
        gzbuffer(f, 1 << 20);
        struct stat st;
        size_t cap = (size_t)4 << 20;
        if (stat(path.c_str(), &st) == 0) cap = std::max<size_t>(cap, (size_t)st.st_size * 4);
        size_t n = 0;
        std::unique_ptr<char[]> buf(new char[cap]);             // not value-initialised
        for (;;) {
            if (n == cap) {
                std::unique_ptr<char[]> bigger(new char[cap * 2]);
                memcpy(bigger.get(), buf.get(), n);
                buf.swap(bigger);
                cap *= 2;
            }
            const int got = gzread(f, buf.get() + n, (unsigned)std::min<size_t>(cap - n, (size_t)1 << 30));
            if (got < 0) { gzclose(f); throw std::runtime_error("read error in " + path); }
            if (got == 0) break;
            n += (size_t)got;
        }
        gzclose(f);
        out = std::move(buf);
        return n;
    }
    const std::vector<std::string> &paths_;
    std::vector<Slot> slots_;
    std::shared_ptr<Shared> state_ = std::make_shared<Shared>();
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::thread> workers_;
    size_t scan_ = 0, next_needed_ = 0, ahead_ = 2, n_prefetch_ = 0;
    int64_t budget_ = (int64_t)2 << 30;
    bool stop_ = false;
};

// reads in a FASTQ file as helper.fastq_parser would yield them: one per four lines (a trailing partial record counts,
// the reader raises on it later)
inline int64_t count_fastq_reads(const std::string &path) {
    auto src = open_lines(path);
    int64_t lines = 0;
    Span s;
    for (;;) {
        Arena scratch;
        bool more = true;
        for (int k = 0; k < 4096 && (more = src->next(scratch, s)); ++k) ++lines;
        if (!more) break;
    }
    return (lines + 3) / 4;
}

// ---- a rank of a file-sharded run finds ITS rows of putative_bc.csv without counting anybody's reads ---------------
// The table has one row per read in the order of the FASTQ files (parser.py:35-47, main.py:361-364), so the rows of a
// rank's files start at the row whose read_id is the id of the first read of its first non-empty file.  Counting the
// reads of all earlier files finds that row too, but reads every FASTQ byte a second time (gzipped files: inflates
// them twice); a search for "\n<id>," in the mapped table, outward from where the row would be if reads were spread
// evenly over the bytes, touches a few megabytes.
inline Span fastq_id(Span line);
inline std::string first_fastq_id(const std::string &path) {
    auto src = open_lines(path);
    Arena a;
    Span s;
    if (!src->next(a, s)) return std::string();
    const Span id = fastq_id(s);
    return std::string(id.p, id.len);
}

// byte offset of the row whose first cell is `id` in the PLAIN csv at `path`, or -1 (not found, or the table is gzipped)
inline int64_t csv_locate_row(const std::string &path, const std::string &id, double hint) {
    if (ends_with_gz(path) || id.empty()) return -1;
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return -1;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size == 0) { close(fd); return -1; }
    const size_t size = (size_t)st.st_size;
    void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { close(fd); return -1; }
    const char *base = (const char *)m;
    const std::string needle = "\n" + id + ",";
    const size_t window = (size_t)16 << 20, overlap = needle.size();
    const size_t mid = (size_t)(std::min(1.0, std::max(0.0, hint)) * (double)size);
    int64_t found = -1;
    auto look = [&](size_t lo, size_t hi) {                       // [lo, hi) plus the overlap behind it
        hi = std::min(size, hi + overlap);
        if (lo >= hi) return false;
        const void *hit = memmem(base + lo, hi - lo, needle.data(), needle.size());
        if (hit) found = (int64_t)((const char *)hit - base) + 1;
        return hit != nullptr;
    };
    for (size_t k = 0;; ++k) {
        const size_t up_lo = mid + k * window, dn_hi = mid > k * window ? mid - k * window : 0;
        const bool up = up_lo < size, dn = dn_hi > 0;
        if (!up && !dn) break;
        if (up && look(up_lo, up_lo + window)) break;
        if (dn && look(dn_hi > window ? dn_hi - window : 0, dn_hi)) break;
    }
    munmap(m, size);
    close(fd);
    return found;
}

// seconds spent per stage of a file job (BLAZE_AMD_IO_TIMING=1 prints them)
struct StageClock {
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};

// ---- one device batch worth of input ------------------------------------------------------------------------
struct Read { Span id, seq, qual, csv; };
struct Chunk {
    Arena arena;                             // copied lines (gzip input)
    std::vector<std::shared_ptr<LineSource>> sources;   // keeps the file mappings the spans point into alive
    std::vector<Read> reads;
    std::vector<uint32_t> batch_start;       // reference batches (gzip member / pandas inference units), + end sentinel
    // filled by parse
    std::vector<Span> bc, umi, post, rid;
    std::vector<double> polyT;               // NaN = blank cell
    std::vector<uint8_t> ext, ext_len, bc_off, bc_len;
    // rows whose extension does not fit kExtStride (a hand-made table: stage 1 only writes 16-nt barcodes): their
    // pre_bc_flanking cell, kept so that the chunk can be packed again at kExtWide bytes per row (widen_rows)
    std::mutex wide_mu;
    std::vector<std::pair<uint32_t, Span>> wide;
    std::vector<uint8_t> ext_wide;
    // filled by the device
    std::vector<int32_t> hit;
    std::vector<int8_t> ed, end_idx;
    std::vector<uint8_t> flags;
    // filled by format
    std::vector<std::string> out;            // per reference batch
    std::vector<int64_t> demul;
    int engine = 0, slot = 0;
};

constexpr int kExtStride = 32;
constexpr int kExtWide = 64;                 // BZ_EXT_MAX: the device's longest row, and the pandas host's limit (_pack_rows)

inline bool is_na_token(const char *s, uint32_t n) {   // pandas' default na_values (parsers STR_NA_VALUES)
    static const char *const toks[] = {"", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN",
                                       "<NA>", "N/A", "NA", "NULL", "NaN", "None", "n/a", "nan", "null"};
    for (const char *t : toks)
        if (strlen(t) == n && memcmp(t, s, n) == 0) return true;
    return false;
}

inline bool looks_numeric(const char *s, uint32_t n) {  // would pandas parse this cell as a number?
    if (n == 0 || n > 63) return false;
    char tmp[64];
    memcpy(tmp, s, n); tmp[n] = 0;
    char *e = nullptr;
    (void)strtod(tmp, &e);
    if (e != tmp + n) return false;
    for (uint32_t i = 0; i < n; ++i) {                  // strtod also takes inf/nan/hex: pandas' tokenizer does not infer those
        const char c = tmp[i];
        if (!((c >= '0' && c <= '9') || c == '+' || c == '-' || c == '.' || c == 'e' || c == 'E')) return false;
    }
    return true;
}

// Split one CSV line into up to 7 cells (RFC-4180 quotes as pandas' C parser reads them); quoted cells are unescaped
// in place.
struct Cells { Span c[7]; int n = 0; };
inline void split_csv(Span line, Cells &out, std::string &err) {
    out.n = 0;
    char *i = line.p;
    char *const end = line.p + line.len;
    for (;;) {
        if (out.n == 7) { err = "putative bc csv: more than 7 cells in a row"; return; }
        Span cell;
        if (i < end && *i == '"') {                     // quoted: unescape in place (result is never longer)
            char *w = i, *r = i + 1;
            cell.p = w;
            while (r < end) {
                if (*r == '"') {
                    if (r + 1 < end && r[1] == '"') { *w++ = '"'; r += 2; continue; }
                    ++r; break;
                }
                *w++ = *r++;
            }
            while (r < end && *r != ',') *w++ = *r++;   // junk after the closing quote is kept, as pandas does
            cell.len = (uint32_t)(w - cell.p);
            i = r;
        } else {
            cell.p = i;
            while (i < end && *i != ',') ++i;
            cell.len = (uint32_t)(i - cell.p);
        }
        out.c[out.n++] = cell;
        if (i >= end) break;
        ++i;                                            // the comma
        if (i == end) { if (out.n < 7) { Span e; e.p = end; e.len = 0; out.c[out.n++] = e; } break; }
    }
}

struct Job {
    std::vector<std::string> fastq_fns;
    std::string csv, out_path;
    bool gz = false, restrand = true;
    int max_ed = 2, batch_size = 1000, n_threads = 8, gz_level = 0;   // gz_level: 0 = fastq_deflate.h, 1..9 = zlib
    int64_t device_batch_rows = 1 << 16;
    int64_t csv_skip_rows = 0;               // rows of the table that belong to reads of earlier FASTQ files (another rank's)
    int64_t csv_byte_offset = -1;            // or: where this rank's first row starts in the (plain) table; -1 = header + csv_skip_rows
};

struct MismatchError : std::runtime_error { using std::runtime_error::runtime_error; };

// ---- the host threads of one file job ---------------------------------------------------------------------------
// A job runs a few dozen fork-join loops (parse, format, pack) per second.  Threads that live as long as the job keep
// their thread_local buffers (the formatter's text, the member encoder's scratch) from one loop to the next and cost no
// creation; threads made per loop paid for both every time.  The caller of run() works too: n_threads - 1 workers.
class Pool {
   public:
    explicit Pool(int n_threads) {
        for (int t = 1; t < std::max(1, n_threads); ++t) workers_.emplace_back([this] { work(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    int size() const { return (int)workers_.size() + 1; }
    // fn(i) for every i in [0, n_items), items handed out one at a time; returns when all are done; the first exception is rethrown
    void run(int n_items, const std::function<void(int)> &fn) {
        if (n_items <= 0) return;
        {
            std::lock_guard<std::mutex> g(mu_);
            fn_ = &fn; n_items_ = n_items; next_.store(0); err_ = nullptr;
            busy_ = (int)workers_.size();
            ++generation_;
        }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return busy_ == 0; });
        fn_ = nullptr;
        if (err_) std::rethrow_exception(err_);
    }
   private:
    void drain() {
        try {
            for (int i; (i = next_.fetch_add(1)) < n_items_;) (*fn_)(i);
        } catch (...) {
            std::lock_guard<std::mutex> g(mu_);
            if (!err_) err_ = std::current_exception();
            next_.store(n_items_);
        }
    }
    void work() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
            }
            drain();
            {
                std::lock_guard<std::mutex> g(mu_);
                --busy_;
            }
            done_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int)> *fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_items_ = 0, busy_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
    std::exception_ptr err_;
};

// the pool of the job this thread drives (set for the lifetime of a PoolScope); parallel_for uses it when it has the asked size
inline Pool *&current_pool() { static thread_local Pool *p = nullptr; return p; }
struct PoolScope {
    explicit PoolScope(Pool *p) : prev_(current_pool()) { current_pool() = p; }
    ~PoolScope() { current_pool() = prev_; }
    Pool *prev_;
};

template <class F>
void parallel_for(int n_items, int n_threads, F &&fn) {
    if (n_items <= 0) return;
    if (Pool *pool = current_pool()) {
        if (pool->size() == std::max(1, n_threads)) { pool->run(n_items, std::function<void(int)>(std::ref(fn))); return; }
    }
    const int nt = std::max(1, std::min(n_threads, n_items));
    std::atomic<int> next{0};
    std::exception_ptr err;
    std::mutex mu;
    auto body = [&]() {
        try {
            for (int i; (i = next.fetch_add(1)) < n_items;) fn(i);
        } catch (...) {
            std::lock_guard<std::mutex> g(mu);
            if (!err) err = std::current_exception();
            next.store(n_items);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(body);
    body();
    for (auto &t : th) t.join();
    if (err) std::rethrow_exception(err);
}

// ---- the frame both file jobs run in ------------------------------------------------------------------------------
// reader thread -> queue (depth 2) -> the CALLING thread, which submits chunk k to its device slot and, once
// `max_inflight` chunks are with the devices, finishes the oldest (waits for it, formats it) -> queue (depth 4) -> writer
// thread.  Chunks are finished and written in the order they were read: the reference's ordered fan-out
// (helper.py:177-205).  run() returns when everything is written and closed; on an error it stops both threads, lets the
// device work of the chunks still in flight end (`abandon`), closes the file and rethrows.
// StageClock slots it fills: 0 reader thread, 1 wait for the reader, 6 hand-over to the writer, 7 writer thread.
template <class ChunkT>
struct OrderedPipeline {
    std::function<std::function<std::unique_ptr<ChunkT>()>()> open_reader;   // called ON the reader thread; its result yields chunks, nullptr at the end
    std::function<void(ChunkT &, int64_t)> submit;                            // prepare the k-th chunk and start its device work
    std::function<void(ChunkT &, std::vector<std::string> &)> finish;         // wait for the chunk's device work; its output parts, in order
    std::function<void(ChunkT &)> abandon;                                    // wait for the device work of a chunk that will not be finished
    std::string out_path;
    bool create_output_first = true;             // false: the file is created when the first chunk is finished (a job without input leaves none)
    int max_inflight = 2;
    StageClock *clk = nullptr;
    int64_t n_chunks = 0;

    void run() {
        struct { std::mutex mu; std::condition_variable cv; std::deque<std::unique_ptr<ChunkT>> q; bool done = false, stop = false; std::exception_ptr err; } Q;
        struct { std::mutex mu; std::condition_variable cv; std::deque<std::vector<std::string>> q; bool done = false, failed = false; } W;
        std::thread reader, writer;
        std::deque<std::unique_ptr<ChunkT>> inflight;
        FILE *fo = nullptr;
        StageClock local;
        StageClock &c = clk ? *clk : local;
        auto open_out = [&]() {
            if (fo) return;
            fo = fopen(out_path.c_str(), "wb");
            if (!fo) throw open_error("cannot open " + out_path + " for writing", errno);
        };
        auto finish_oldest = [&]() {
            auto old = std::move(inflight.front());
            inflight.pop_front();
            std::vector<std::string> parts;
            finish(*old, parts);
            const double t0 = StageClock::now();
            open_out();
            {
                std::unique_lock<std::mutex> lk(W.mu);
                W.cv.wait(lk, [&] { return W.q.size() < 4 || W.failed; });
                if (W.failed) throw std::runtime_error("write error on " + out_path);
                W.q.push_back(std::move(parts));
            }
            W.cv.notify_all();
            c.t[6] += StageClock::now() - t0;
        };
        try {
            if (create_output_first) open_out();
            reader = std::thread([&]() {
                try {
                    auto next = open_reader();
                    for (;;) {
                        const double t0 = StageClock::now();
                        auto ch = next();
                        c.t[0] += StageClock::now() - t0;
                        std::unique_lock<std::mutex> lk(Q.mu);
                        if (!ch) break;
                        Q.cv.wait(lk, [&] { return Q.q.size() < 2 || Q.stop; });
                        if (Q.stop) return;
                        Q.q.push_back(std::move(ch));
                        Q.cv.notify_all();
                    }
                } catch (...) {
                    std::lock_guard<std::mutex> g(Q.mu);
                    Q.err = std::current_exception();
                }
                { std::lock_guard<std::mutex> g(Q.mu); Q.done = true; }
                Q.cv.notify_all();
            });
            writer = std::thread([&]() {
                for (;;) {
                    std::vector<std::string> parts;
                    {
                        std::unique_lock<std::mutex> lk(W.mu);
                        W.cv.wait(lk, [&] { return !W.q.empty() || W.done; });
                        if (W.q.empty()) return;
                        parts = std::move(W.q.front());
                        W.q.pop_front();
                    }
                    W.cv.notify_all();
                    const double t0 = StageClock::now();
                    for (const std::string &part : parts)
                        if (!part.empty() && !W.failed && fwrite(part.data(), 1, part.size(), fo) != part.size()) {
                            std::lock_guard<std::mutex> g(W.mu);
                            W.failed = true;
                        }
                    c.t[7] += StageClock::now() - t0;
                }
            });
            for (int64_t k = 0;; ++k) {
                std::unique_ptr<ChunkT> ch;
                const double t0 = StageClock::now();
                {
                    std::unique_lock<std::mutex> lk(Q.mu);
                    Q.cv.wait(lk, [&] { return !Q.q.empty() || Q.done; });
                    if (Q.err) std::rethrow_exception(Q.err);
                    if (Q.q.empty()) break;
                    ch = std::move(Q.q.front());
                    Q.q.pop_front();
                }
                Q.cv.notify_all();
                c.t[1] += StageClock::now() - t0;
                if ((int)inflight.size() == max_inflight) finish_oldest();      // the oldest chunk holds the slot this one takes
                submit(*ch, k);
                inflight.push_back(std::move(ch));
                ++n_chunks;
            }
            while (!inflight.empty()) finish_oldest();
            { std::lock_guard<std::mutex> g(W.mu); W.done = true; }
            W.cv.notify_all();
            writer.join();
            if (W.failed) throw std::runtime_error("write error on " + out_path);
            if (fo) {
                FILE *f = fo;
                fo = nullptr;
                if (fclose(f) != 0) throw std::runtime_error("write error on " + out_path);
            }
            reader.join();
        } catch (...) {
            { std::lock_guard<std::mutex> g(Q.mu); Q.stop = true; }
            Q.cv.notify_all();
            if (reader.joinable()) reader.join();
            { std::lock_guard<std::mutex> g(W.mu); W.done = true; }
            W.cv.notify_all();
            if (writer.joinable()) writer.join();
            for (auto &ch : inflight) abandon(*ch);
            inflight.clear();
            if (fo) { fclose(fo); fo = nullptr; }
            throw;
        }
    }
};

// ---- reader: FASTQ records + the same number of csv lines, cut into device batches of whole reference batches --
class ChunkReader {
   public:
    explicit ChunkReader(const Job &job)
        : job_(job), csv_(job.csv_byte_offset >= 0 ? std::shared_ptr<LineSource>(std::make_shared<MmapLines>(job.csv, (size_t)job.csv_byte_offset))
                                                   : open_lines(job.csv)),
          files_(job.fastq_fns, job.n_threads) {
        if (job.csv_byte_offset >= 0) return;                      // positioned on this rank's first row already
        Arena tmp; Span s;
        if (!csv_->next(tmp, s)) throw std::runtime_error("putative bc csv is empty: " + job.csv);   // header (:150 `next(f)`)
        for (int64_t k = 0; k < job.csv_skip_rows; ++k) {
            Arena skip;                                            // copied lines (gzip input) are dropped at once
            if (!csv_->next(skip, s)) throw std::runtime_error("putative bc csv has fewer rows than the FASTQ files have reads");
        }
    }
    // Next chunk or nullptr when every FASTQ file is exhausted.
    std::unique_ptr<Chunk> next() {
        auto ch = std::make_unique<Chunk>();
        ch->sources.push_back(csv_);
        if (fq_) ch->sources.push_back(fq_);
        Arena skip;                                                // '+' lines of gzip input: copied, then dropped
        while ((int64_t)ch->reads.size() < job_.device_batch_rows) {
            if (!fq_) {
                if (file_idx_ >= job_.fastq_fns.size()) break;
                fq_ = files_.open(file_idx_++);
                ch->sources.push_back(fq_);
            }
            // one reference batch: up to batch_size reads of THIS file (helper.batch_iterator is per file, :156-158)
            const uint32_t start = (uint32_t)ch->reads.size();
            int got = 0;
            for (; got < job_.batch_size; ++got) {
                Read r; Span plus;
                if (!fq_->next(ch->arena, r.id)) break;
                if (!fq_->next(ch->arena, r.seq) || !fq_->next(skip, plus) || !fq_->next(ch->arena, r.qual))     // '+' line skipped (helper.py:299)
                    throw std::runtime_error("truncated FASTQ record in " + job_.fastq_fns[file_idx_ - 1]);
                ch->reads.push_back(r);
            }
            if (got == 0) { fq_.reset(); continue; }
            ch->batch_start.push_back(start);
            for (int k = 0; k < got; ++k)
                if (!csv_->next(ch->arena, ch->reads[start + k].csv))
                    throw std::runtime_error("putative bc csv has fewer rows than the FASTQ files have reads");   // StopIteration in the reference
            if (got < job_.batch_size) fq_.reset();
        }
        if (ch->reads.empty()) return nullptr;
        ch->batch_start.push_back((uint32_t)ch->reads.size());
        return ch;
    }

   private:
    const Job &job_;
    std::shared_ptr<LineSource> csv_;
    FastqFiles files_;
    std::shared_ptr<LineSource> fq_;
    size_t file_idx_ = 0;
};

inline Span strip(Span s) {                                           // str.strip()
    while (s.len && py_space((unsigned char)*s.p)) { ++s.p; --s.len; }
    while (s.len && py_space((unsigned char)s.p[s.len - 1])) --s.len;
    return s;
}

// read id: title[1:].split()[0] (helper.py:301) then .split()[0].strip('@') (read_assignment.py:31)
inline Span fastq_id(Span line) {
    if (line.len) { ++line.p; --line.len; }                           // title[1:]
    while (line.len && py_space((unsigned char)*line.p)) { ++line.p; --line.len; }
    uint32_t n = 0;
    while (n < line.len && !py_space((unsigned char)line.p[n])) ++n;
    if (n == 0) throw std::runtime_error("FASTQ title line without a read id");   // IndexError in the reference
    line.len = n;
    while (line.len && *line.p == '@') { ++line.p; --line.len; }
    while (line.len && line.p[line.len - 1] == '@') --line.len;
    return line;
}

// ---- parse one reference batch: cells, NA handling, id check, ext rows --------------------------------------
inline void parse_batch(Chunk &ch, int b) {
    const uint32_t lo = ch.batch_start[b], hi = ch.batch_start[b + 1];
    std::string err;
    bool all_numeric_ids = true;
    for (uint32_t r = lo; r < hi; ++r) {
        Read &rd = ch.reads[r];
        Cells cells;
        split_csv(rd.csv, cells, err);
        if (!err.empty()) throw std::runtime_error(err);
        Span cell[7];
        for (int k = 0; k < 7; ++k) {
            cell[k] = k < cells.n ? cells.c[k] : Span();
            if (is_na_token(cell[k].p, cell[k].len)) cell[k].len = 0;   // NaN -> fillna('')
        }
        // columns: read_id, putative_bc, putative_bc_min_q, putative_umi, polyT_end, pre_bc_flanking, post_umi_flanking
        ch.rid[r] = cell[0]; ch.bc[r] = cell[1]; ch.umi[r] = cell[3]; ch.post[r] = cell[6];
        all_numeric_ids = all_numeric_ids && looks_numeric(cell[0].p, cell[0].len);
        double pt = std::numeric_limits<double>::quiet_NaN();
        if (cell[4].len) {
            char tmp[64];
            const uint32_t n = std::min<uint32_t>(cell[4].len, 63);
            memcpy(tmp, cell[4].p, n); tmp[n] = 0;
            char *e = nullptr;
            pt = strtod(tmp, &e);
            if (e != tmp + n || n != cell[4].len) throw std::runtime_error("putative bc csv: polyT_end is not a number: " + std::string(tmp));
        }
        ch.polyT[r] = pt;
        // FASTQ side
        rd.seq = strip(rd.seq);
        rd.qual = strip(rd.qual);
        rd.id = fastq_id(rd.id);
        if (rd.id.len != cell[0].len || memcmp(rd.id.p, cell[0].p, rd.id.len) != 0)
            throw MismatchError("Different order in putative bc file and input fastq!");
        // ext = pre_bc_flanking[-2:] + putative_bc + putative_umi[:2]   (:70)
        uint8_t *e = ch.ext.data() + (size_t)r * kExtStride;
        memset(e, 0, kExtStride);
        const Span pre = cell[5];
        if (cell[1].len == 0) { ch.ext_len[r] = ch.bc_off[r] = ch.bc_len[r] = 0; continue; }     // blank row (:61)
        const uint32_t npre = std::min<uint32_t>(pre.len, 2), numi = std::min<uint32_t>(cell[3].len, 2);
        const uint32_t total = npre + cell[1].len + numi;
        if (total > kExtStride) {
            if (total > kExtWide || cell[1].len > kExtWide)
                throw std::runtime_error("putative_bc row: extension of " + std::to_string(total) + " nt exceeds the device limit of " + std::to_string(kExtWide));
            ch.ext_len[r] = (uint8_t)total; ch.bc_off[r] = (uint8_t)npre; ch.bc_len[r] = (uint8_t)cell[1].len;
            std::lock_guard<std::mutex> g(ch.wide_mu);
            ch.wide.emplace_back(r, pre);
            continue;
        }
        memcpy(e, pre.p + pre.len - npre, npre);
        memcpy(e + npre, cell[1].p, cell[1].len);
        memcpy(e + npre + cell[1].len, cell[3].p, numi);
        ch.ext_len[r] = (uint8_t)total; ch.bc_off[r] = (uint8_t)npre; ch.bc_len[r] = (uint8_t)cell[1].len;
    }
    if (all_numeric_ids && hi > lo)       // pandas would have made read_id a numeric column: `bc.read_id != r.id` (:220)
        throw MismatchError("Different order in putative bc file and input fastq!");
}

// a chunk with rows wider than kExtStride: every row again at kExtWide bytes (after all its batches are parsed)
inline void widen_rows(Chunk &ch) {
    const size_t n = ch.reads.size();
    ch.ext_wide.assign(n * kExtWide, 0);
    for (size_t r = 0; r < n; ++r)
        if (ch.ext_len[r] <= kExtStride) memcpy(ch.ext_wide.data() + r * kExtWide, ch.ext.data() + r * kExtStride, kExtStride);
    for (const auto &w : ch.wide) {
        const uint32_t r = w.first;
        const Span pre = w.second;
        uint8_t *e = ch.ext_wide.data() + (size_t)r * kExtWide;
        const uint32_t npre = ch.bc_off[r], nbc = ch.bc_len[r], numi = (uint32_t)ch.ext_len[r] - npre - nbc;
        memcpy(e, pre.p + pre.len - npre, npre);
        memcpy(e + npre, ch.bc[r].p, nbc);
        memcpy(e + npre + nbc, ch.umi[r].p, numi);
    }
}

inline void gzip_member(const std::string &in, std::string &out, int level) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2 failed");
    out.resize(deflateBound(&zs, (uLong)in.size()) + 32);
    zs.next_in = (Bytef *)in.data(); zs.avail_in = (uInt)in.size();
    zs.next_out = (Bytef *)out.data(); zs.avail_out = (uInt)out.size();
    const int rc = deflate(&zs, Z_FINISH);
    const size_t n = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) throw std::runtime_error("deflate failed");
    out.resize(n);
}

inline char comp(char c) {                                  // helper.reverse_complement (helper.py:14-25)
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        default: return c;
    }
}

// dst[k] = comp(src[n - 1 - k]) / src[n - 1 - k]: the restrand of read_assignment.py:234-236 for one record.  32 bytes per
// step where the host has AVX2 (byte reversal by shuffle + lane swap; the complement as an XOR: A<->T differ by 0x15,
// C<->G by 0x04, both cases, every other byte unchanged as in helper.py:14-25), bytes through comp() otherwise.
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void reverse_avx2(char *dst, const char *src, size_t n, bool complement) {
    const __m256i rev = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    const __m256i lc = _mm256_set1_epi8(0x20), a = _mm256_set1_epi8('a'), t = _mm256_set1_epi8('t'), c = _mm256_set1_epi8('c'),
                  g = _mm256_set1_epi8('g'), x_at = _mm256_set1_epi8(0x15), x_cg = _mm256_set1_epi8(0x04);
    size_t k = 0;
    for (; k + 32 <= n; k += 32) {
        __m256i v = _mm256_loadu_si256((const __m256i *)(src + n - 32 - k));
        v = _mm256_permute4x64_epi64(_mm256_shuffle_epi8(v, rev), 0x4E);           // bytes reversed across the 32
        if (complement) {
            const __m256i low = _mm256_or_si256(v, lc);
            const __m256i at = _mm256_or_si256(_mm256_cmpeq_epi8(low, a), _mm256_cmpeq_epi8(low, t));
            const __m256i cg = _mm256_or_si256(_mm256_cmpeq_epi8(low, c), _mm256_cmpeq_epi8(low, g));
            v = _mm256_xor_si256(v, _mm256_or_si256(_mm256_and_si256(at, x_at), _mm256_and_si256(cg, x_cg)));
        }
        _mm256_storeu_si256((__m256i *)(dst + k), v);
    }
    for (; k < n; ++k) dst[k] = complement ? comp(src[n - 1 - k]) : src[n - 1 - k];
}
#endif
inline void reverse_into(char *dst, const char *src, size_t n, bool complement) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) { reverse_avx2(dst, src, n, complement); return; }
#endif
    for (size_t k = 0; k < n; ++k) dst[k] = complement ? comp(src[n - 1 - k]) : src[n - 1 - k];
}

// Python slice of a string of length n: [:cut] when neg (cut < 0 given), [cut:] otherwise -> (start, len)
inline void py_cut(uint32_t n, double polyT, uint32_t &start, uint32_t &len) {
    const double tr = polyT < 0 ? -std::floor(-polyT) : std::floor(polyT);       // int(): toward zero
    const int64_t cut = (int64_t)tr;
    if (polyT < 0) {                                         // seq[:cut], cut <= 0 (cut == 0 only for -1 < x < 0 -> '')
        const int64_t endp = cut < 0 ? std::max<int64_t>(0, (int64_t)n + cut) : 0;
        start = 0; len = (uint32_t)endp;
    } else {                                                 // seq[cut:]
        const int64_t s = std::min<int64_t>(cut, n);
        start = (uint32_t)s; len = n - (uint32_t)s;
    }
}

// ---- format one reference batch (read_assignment.py:87-99 epilogue + :214-247) ---------------------------------
inline void format_batch(Chunk &ch, int b, const Job &job, const std::vector<char> &wl_bytes, const std::vector<int64_t> &wl_off) {
    const uint32_t lo = ch.batch_start[b], hi = ch.batch_start[b + 1];
    // .gz: the text only lives until it is compressed -- one buffer per host thread, grown once; plain: the text IS the
    // output and is handed to the writer
    static thread_local std::string tl_text;
    std::string own;
    std::string &text = job.gz ? tl_text : own;
    text.clear();
    {
        size_t bound = 0;                                    // the batch's records at most: no regrowth while appending
        for (uint32_t r = lo; r < hi; ++r)
            if (ch.bc[r].len && ch.hit[r] >= 0) bound += (size_t)ch.reads[r].seq.len + ch.reads[r].qual.len + ch.rid[r].len + 160;
        text.reserve(bound);
    }
    int64_t demul = 0;
    std::string umi;
    for (uint32_t r = lo; r < hi; ++r) {
        if (ch.bc[r].len == 0) continue;                     // blank row: BC_corrected == '' (:61-62)
        const int32_t h = ch.hit[r];
        if (h < 0) continue;                                 // none / ambiguous (:87-88)
        const Read &rd = ch.reads[r];
        const char *bc; uint32_t bcl;
        umi.assign(ch.umi[r].p, ch.umi[r].len);
        if (ch.flags[r] & 1u) { bc = ch.bc[r].p; bcl = ch.bc[r].len; }       // exact shortcut: untouched
        else {
            bc = wl_bytes.data() + wl_off[h]; bcl = (uint32_t)(wl_off[h + 1] - wl_off[h]);
            const int adj = (int)ch.end_idx[r] - ((int)ch.ext_len[r] - 1 - 2);      // :93
            if (adj > 0) {                                   // umi[adj:] + post[:adj]
                const uint32_t a = std::min<uint32_t>((uint32_t)adj, (uint32_t)umi.size());
                umi.erase(0, a);
                umi.append(ch.post[r].p, std::min<uint32_t>((uint32_t)adj, ch.post[r].len));
            } else if (adj < 0) {                            // putative_bc[adj:] + umi[:adj]
                const uint32_t k = (uint32_t)(-adj);
                const Span pb = ch.bc[r];
                const uint32_t take = std::min(k, pb.len);
                std::string nu(pb.p + pb.len - take, take);
                if (umi.size() > k) nu.append(umi, 0, umi.size() - k);
                umi.swap(nu);
            }
        }
        if (bcl == 0) continue;                              // an empty whitelist entry can never be a hit (falsy in :87)
        ++demul;
        const double pt = ch.polyT[r];
        if (pt != pt) throw std::runtime_error("putative bc csv: row with a barcode but no polyT_end");   // TypeError in the reference
        const char strand = pt == 0 ? 0 : (pt > 0 ? '-' : '+');
        uint32_t s0, sl, q0, ql;
        py_cut(rd.seq.len, pt, s0, sl);
        py_cut(rd.qual.len, pt, q0, ql);
        text.push_back('@'); text.append(bc, bcl); text.push_back('_'); text.append(umi); text.push_back('#');
        text.append(ch.rid[r].p, ch.rid[r].len); text.push_back('_');
        if (strand) text.push_back(strand);
        text.append("\tCB:Z:"); text.append(bc, bcl); text.append("\tUB:Z:"); text.append(umi); text.push_back('\n');
        const char *sq = rd.seq.p + s0, *ql_p = rd.qual.p + q0;
        if (job.restrand && strand == '-') {
            const size_t at = text.size();
            text.resize(at + sl);
            reverse_into(&text[at], sq, sl, true);
            text.append("\n+\n");
            const size_t aq = text.size();
            text.resize(aq + ql);
            reverse_into(&text[aq], ql_p, ql, false);
        } else {
            text.append(sq, sl); text.append("\n+\n"); text.append(ql_p, ql);
        }
        text.push_back('\n');
    }
    ch.demul[b] = demul;
    if (job.gz) {                                               // one member per batch, also when empty (gzip.compress(b''))
        if (job.gz_level == 0) fqz::gzip_fastq_member(text.data(), text.size(), ch.out[b]);   // FASTQ-aware blocks (fastq_deflate.h)
        else gzip_member(text, ch.out[b], job.gz_level);                                     // zlib at the given level
    }
    else ch.out[b].swap(text);
}

// =================================================================================================================
// Stage 1 over files: bz_putative_bc_files().  What it replaces (file:line under /root/reference/blaze):
//   read_batch_generator                     main.py:304-323      batches of --batch-size reads, file by file
//   the loop over the pool's results         main.py:355-364      pass-code counts + DataFrame.to_csv per batch
//   get_raw_bc_from_reads' table             main.py:96-105       the seven columns, None for reads without a barcode
//   Read.raw_bc / putative_UMI / flankings   polyT_adaptor_finder.py:263-342   slices of the strand view
// The device gives five numbers per read (k_stage1); the cells are slices of the read at those numbers, cut here with
// Python's slice rules.  `DataFrame.to_csv` is restated for this table: a batch's putative_bc_min_q / polyT_end columns
// are int64 when every read of the batch has a barcode ("15"), float64 as soon as one has none ("15.0", blanks for the
// missing), and all blank when none has; cells holding a comma, a quote or a newline are quoted with doubled quotes
// (csv.QUOTE_MINIMAL); the header goes out with the first batch of the job only.
struct S1Read { Span id, seq, qual; };
struct S1Chunk {
    Arena arena;
    std::vector<std::shared_ptr<LineSource>> sources;
    std::vector<S1Read> reads;
    std::vector<uint32_t> batch_start;       // reference batches + end sentinel
    std::vector<int64_t> off;                // where each read goes in the packed device buffers
    std::vector<int32_t> code, start, minq, polyt;
    std::vector<int8_t> strand;
    std::vector<std::string> out;            // csv text per reference batch
    std::vector<std::array<int64_t, 5>> pass;   // per batch: reads with pass code 0, 1, 2, 10, 100
    int engine = 0, slot = 0;
    bool first = false;                      // first chunk of the job: its first batch carries the header
};

struct S1Job {
    std::vector<std::string> fastq_fns;
    std::string out_path;
    bool kit5 = false;
    int umi_len = 12, min_q = 15, batch_size = 1000, n_threads = 8;
    int64_t device_batch_reads = 1 << 15, device_batch_bytes = (int64_t)192 << 20;
};

class FastqChunkReader {
   public:
    explicit FastqChunkReader(const S1Job &job) : job_(job), files_(job.fastq_fns, job.n_threads) {}
    std::unique_ptr<S1Chunk> next() {
        auto ch = std::make_unique<S1Chunk>();
        if (fq_) ch->sources.push_back(fq_);
        Arena skip;
        int64_t bytes = 0;
        while ((int64_t)ch->reads.size() < job_.device_batch_reads && bytes < job_.device_batch_bytes) {
            if (!fq_) {
                if (file_idx_ >= job_.fastq_fns.size()) break;
                fq_ = files_.open(file_idx_++);
                ch->sources.push_back(fq_);
            }
            const uint32_t start = (uint32_t)ch->reads.size();
            int got = 0;
            for (; got < job_.batch_size; ++got) {
                S1Read r; Span plus;
                if (!fq_->next(ch->arena, r.id)) break;
                if (!fq_->next(ch->arena, r.seq) || !fq_->next(skip, plus) || !fq_->next(ch->arena, r.qual))
                    throw std::runtime_error("truncated FASTQ record in " + job_.fastq_fns[file_idx_ - 1]);
                bytes += r.seq.len;
                ch->reads.push_back(r);
            }
            if (got == 0) { fq_.reset(); continue; }
            ch->batch_start.push_back(start);
            if (got < job_.batch_size) fq_.reset();
        }
        if (ch->reads.empty()) return nullptr;
        ch->batch_start.push_back((uint32_t)ch->reads.size());
        return ch;
    }
   private:
    const S1Job &job_;
    FastqFiles files_;
    std::shared_ptr<LineSource> fq_;
    size_t file_idx_ = 0;
};

// strip the lines, take the ids, compute where each read goes in the packed device buffers
inline void s1_prepare(S1Chunk &ch, int n_threads) {
    const size_t n = ch.reads.size();
    const int parts = std::max(1, std::min<int>(n_threads, (int)(n / 1024) + 1));
    parallel_for(parts, n_threads, [&](int p) {
        const size_t lo = n * p / parts, hi = n * (p + 1) / parts;
        for (size_t r = lo; r < hi; ++r) {
            S1Read &rd = ch.reads[r];
            rd.seq = strip(rd.seq); rd.qual = strip(rd.qual);
            if (rd.seq.len != rd.qual.len) throw std::runtime_error("sequence and quality strings differ in length");
            // id = title[1:].split()[0] (helper.py:301): no '@' stripping here, unlike read_assignment.py:31
            Span t = rd.id;
            if (t.len) { ++t.p; --t.len; }
            while (t.len && py_space((unsigned char)*t.p)) { ++t.p; --t.len; }
            uint32_t k = 0;
            while (k < t.len && !py_space((unsigned char)t.p[k])) ++k;
            if (k == 0) throw std::runtime_error("FASTQ title line without a read id");   // IndexError in the reference
            t.len = k;
            rd.id = t;
        }
    });
    ch.off.assign(n + 1, 0);
    for (size_t r = 0; r < n; ++r) ch.off[r + 1] = ch.off[r] + ch.reads[r].seq.len;
}

// lay sequence and quality bytes out back to back in the engine's (pinned) staging buffers
inline void s1_copy(const S1Chunk &ch, uint8_t *seq, uint8_t *qual, int64_t *off, int n_threads) {
    const size_t n = ch.reads.size();
    memcpy(off, ch.off.data(), (n + 1) * sizeof(int64_t));
    const int parts = std::max(1, std::min<int>(4 * n_threads, (int)(n / 256) + 1));
    parallel_for(parts, n_threads, [&](int p) {
        const size_t lo = n * p / parts, hi = n * (p + 1) / parts;
        for (size_t r = lo; r < hi; ++r) {
            memcpy(seq + ch.off[r], ch.reads[r].seq.p, ch.reads[r].seq.len);
            memcpy(qual + ch.off[r], ch.reads[r].qual.p, ch.reads[r].qual.len);
        }
    });
}

// view[a:b] of a read (Python slice rules; a may be negative, b is never) appended as one csv cell; the view is the
// read itself on strand '-' and its reverse complement on '+'
inline void s1_cell(std::string &text, const Span &seq, bool rc, int64_t a, int64_t b) {
    const int64_t n = seq.len;
    if (a < 0) a = std::max<int64_t>(0, a + n);
    a = std::min(a, n); b = std::min(std::max<int64_t>(b, 0), n);
    if (b <= a) return;
    const size_t at = text.size();
    bool special = false;
    text.resize(at + (size_t)(b - a));
    for (int64_t i = a; i < b; ++i) {
        const char c = rc ? comp(seq.p[n - 1 - i]) : seq.p[i];
        special = special || c == ',' || c == '"' || c == '\n';
        text[at + (size_t)(i - a)] = c;
    }
    if (special) {
        std::string q = "\"";
        for (size_t i = at; i < text.size(); ++i) { if (text[i] == '"') q.push_back('"'); q.push_back(text[i]); }
        q.push_back('"');
        text.resize(at);
        text += q;
    }
}

inline void s1_format_batch(S1Chunk &ch, int b, const S1Job &job) {
    const uint32_t lo = ch.batch_start[b], hi = ch.batch_start[b + 1];
    std::string &text = ch.out[b];
    text.clear();
    if (ch.first && b == 0) text = "read_id,putative_bc,putative_bc_min_q,putative_umi,polyT_end,pre_bc_flanking,post_umi_flanking\n";
    std::array<int64_t, 5> pass = {0, 0, 0, 0, 0};
    bool any_none = false;
    for (uint32_t r = lo; r < hi; ++r) {
        if (ch.code[r] < 0) throw std::runtime_error("min() arg is an empty sequence");   // polyT_adaptor_finder.py:278 raises on this read
        any_none = any_none || ch.code[r] != 0;
    }
    const char *num_tail = any_none ? ".0," : ",";            // float64 column as soon as one cell of the batch is None
    char num[32];
    for (uint32_t r = lo; r < hi; ++r) {
        const S1Read &rd = ch.reads[r];
        const Span id0 = rd.id;
        {                                                     // the id is a cell like any other
            Span whole = id0;
            s1_cell(text, whole, false, 0, whole.len);
        }
        const int code = ch.code[r];
        if (code != 0) {
            text += ",,,,,,\n";
            pass[code == 1 ? 1 : code == 2 ? 2 : 3] += 1;
            continue;
        }
        const bool rc = ch.strand[r] == '+';
        const int64_t s = ch.start[r], u = job.umi_len;
        const int mq = ch.minq[r];
        pass[(mq != 0 && mq < job.min_q) ? 4 : 0] += 1;       // main.py:88-94 (a minimum quality of 0 is falsy there)
        text.push_back(',');
        s1_cell(text, rd.seq, rc, s, s + 16);
        text.push_back(',');
        snprintf(num, sizeof num, "%d", mq); text += num; text += num_tail;
        s1_cell(text, rd.seq, rc, s + 16, s + 16 + u);
        text.push_back(',');
        snprintf(num, sizeof num, "%d", ch.polyt[r]); text += num; text += num_tail;
        s1_cell(text, rd.seq, rc, s - 5, s);
        text.push_back(',');
        s1_cell(text, rd.seq, rc, s + 16 + u, s + 16 + u + 5);
        text.push_back('\n');
    }
    ch.pass[b] = pass;
}

}  // namespace bzio
