// blaze_hip.hip -- host side of libblaze_hip.so: the C ABI declared in include/blaze_hip.h.
// Engine = one GPU + two slots (stream, pinned staging, device buffers) + the packed whitelist
// + the per-barcode histogram + an optional RCCL communicator (dlopen'ed, one process per GPU).
// There is no CPU compute path in this file.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/blaze_hip.h"
#include "scan_kernels.h"
#include "stage1_kernels.h"
#include "host_io.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(BZ_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---- RCCL through dlopen: no link-time dependency, one copy per process --------------------
struct RcclId { char b[128]; };   // ncclUniqueId (rccl.h:40-43), passed by value
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, RcclId, int) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.lib) return BZ_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *lib = nullptr;
    for (const char *n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib) break;
    }
    if (!lib) return fail(BZ_E_RCCL, "cannot dlopen librccl: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(lib, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return fail(BZ_E_RCCL, "librccl lacks an expected symbol");
    g_rccl.lib = lib;
    return BZ_OK;
}
constexpr int kNcclSum = 0, kNcclMax = 2, kNcclInt64 = 4, kNcclFloat64 = 8;   // rccl.h:448-467

struct Slot {
    hipStream_t stream = nullptr;
    hipEvent_t ev[7] = {};      // 0..4 run phases; 5, 6 = after k_pair_candidates / after k_pair_verify
    bool timed_pair = false;
    bool tail_done = false;               // this run's scan already launched the rare-row scans (k_pair_tail)
    int64_t cap = 0;          // rows
    int cap_stride = 0;
    int64_t n = 0;
    int stride = 0;
    bool staged = false, ran = false;
    bool direct_out = false;              // the run's results were copied straight into the caller's pinned arrays (bz_assign_submit_pinned)
    // device
    uint8_t *d_ext = nullptr, *d_len = nullptr, *d_off = nullptr, *d_bcl = nullptr, *d_flags = nullptr;
    int32_t *d_hit = nullptr, *d_fast = nullptr, *d_gen = nullptr, *d_redo = nullptr, *d_counters = nullptr;
    void *d_fast_tc = nullptr;             // uint2 per fast_list slot: the text's 2-bit codes + length (k_classify_packed -> pair kernels)
    int8_t *d_ed = nullptr, *d_end = nullptr;
    uint32_t *d_cand = nullptr, *d_cand_cnt = nullptr;   // candidate lists of the pair-seeded scan: cand[slot / 64][s][slot % 64], tiles of (cap + 1) x 64
    int64_t cand_words = 0;
    int cand_stride = 0;
    // pinned host staging
    uint8_t *h_in = nullptr, *h_out = nullptr;
    int32_t *h_counters = nullptr, *h_counters_dev = nullptr;   // pinned + mapped: the last kernel of a run writes it
    int32_t *d_counters_base = nullptr;   // two sets of 4; d_counters points at the set of the current run
    // deferred histogram (scan_kernels.h: kPiggyBins): the hits of consecutive runs alternate between two buffers, and a
    // run whose per-barcode counts have not been taken yet is `pend`ing: the slot's next run carries them along, or
    // flush_pending() runs them on their own for whoever needs counts, list counters or timings first
    int32_t *d_hit_buf[2] = {nullptr, nullptr};
    int hit_cur = 0;
    bool pend = false;
    int pend_n = 0;
    const int32_t *pend_hit = nullptr;
    int32_t *pend_counters = nullptr;
    bool timed = false;
};

// stage 1: one staging set per slot (pinned host + device), grow-only; the file pipeline packs reads straight into
// the pinned buffers and double-buffers over the two slots' streams
struct S1Buf {
    uint8_t *h_seq = nullptr, *h_qual = nullptr, *d_seq = nullptr, *d_qual = nullptr;
    int64_t *h_off = nullptr, *d_off = nullptr;
    int32_t *h_i32 = nullptr, *d_i32 = nullptr;     // pass_code, raw_bc_start, min_q, polyT_end: 4 x cap_reads
    int8_t *h_strand = nullptr, *d_strand = nullptr;
    int64_t cap_bytes = 0, cap_reads = 0, n = 0;
};

}  // namespace

// ---- scratch of the entry point that allocates nothing (bz_assign_device_noalloc) ---------------------------------
// Work lists + counters of one run over at most BZ_NOALLOC_MAX_ROWS rows, in a file-scope device array: each GPU the
// library runs on holds its own copy until the process ends.  In host code the NAME below is only a stand-in (passing it
// to a kernel compiles and then faults): g_scratch_ptr[device], set once per device by scratch_init() through
// hipGetSymbolAddress with that device current, is the address host code uses.
constexpr size_t kNaRows = BZ_NOALLOC_MAX_ROWS;
constexpr size_t kNaOffFast = 0, kNaOffGen = kNaOffFast + 4 * kNaRows, kNaOffRedo = kNaOffGen + 4 * kNaRows,
                 kNaOffTc = kNaOffRedo + 4 * kNaRows, kNaOffCounters = kNaOffTc + 8 * kNaRows, kNaBytes = kNaOffCounters + 256;
static __device__ __align__(256) unsigned char scratch[kNaBytes];
constexpr int kMaxDevices = 64;
static unsigned char *g_scratch_ptr[kMaxDevices] = {};
static std::mutex g_scratch_mu;

struct bz_engine {
    int device = 0;
    int tie_rule = 0, shortcut = 1, force_generic = 0, scan_mode = 0, pair_fused = 0, profile = 0, gz_level = 0;
    // whitelist
    int64_t n_wl = -1;
    int wl_regular = 0;                 // the packed fast path is usable (every entry 16-nt ACGT, or all but a few: `mixed`)
    int64_t n_fast = 0;                 // entries the fast-path structures below are built over (the regular ones, in caller order)
    int mixed = 0, n_irr = 0;           // irregular entries beside the regular ones (k_mixed_fix)
    int irr_only = 0;                   // a list WITHOUT a regular entry (<= 65,536 entries): every row through the workgroup-per-row scan
    int32_t *d_reg_to_orig = nullptr;   //   sub-list index -> index in the caller's list
    uint8_t *d_irr_bytes = nullptr;     //   the irregular entries, back to back
    int64_t *d_irr_off = nullptr;
    uint4 *d_irr_peq = nullptr;         //   per irregular entry: which positions hold A / C / T / G (code order), bit i = character i
    int2 *d_irr_meta = nullptr;         //   {length, index in the caller's list}
    int n_pairs = 0;
    uint2 *d_planes = nullptr;
    uint4 *d_grams = nullptr;          // 3-gram sets, two barcodes per uint4, padded to 32 barcodes
    uint32_t *d_planes32 = nullptr;    // hi-plane | lo-plane << 16 per barcode (verify gather)
    int32_t *d_bucket = nullptr;       // bucket table over the sorted codes' top bits
    uint2 *d_bucket8 = nullptr;        //   and its one-gather form (k_classify_packed)
    int32_t *d_seed_tab[3] = {nullptr, nullptr, nullptr};      // seed index, plan 0: pieces 5|5|6 (max_ed 2), plan 1: 8|8 (max_ed <= 1), plan 2: 4|4|4|4 (max_ed 3)
    int32_t *d_seed_members[3] = {nullptr, nullptr, nullptr};  // per piece n_fast barcode indices grouped by the piece's code
    uint32_t *d_pair_bitmap = nullptr;  // pair-seed index: 6 x 65536-bit bucket occupancy (k_pair_candidates)
    int32_t *d_pair_tab = nullptr;      //   6 CSR tables over the 16-bit key (piece a code << 8 | piece b code)
    uint2 *d_pair_members = nullptr;    //   6 x n_wl {barcode index, planes32} grouped by key, index order inside a bucket
    uint2 *d_pair_tab8 = nullptr;       //   6 x 65536 {bucket size, the barcode (size 1) or offset into members} (k_pair_candidates)
    int pair_cap = 0;                   //   candidate slots per row for the two-kernel form, 0 = use k_pair_stream
    int n_cu = 0;
    int bucket_bits = 0, n_blk = 0;
    size_t cand_lds_set = 0;            // dynamic LDS bytes k_pair_candidates has been allowed so far
    uint32_t *d_sorted_codes = nullptr;
    uint2 *d_sorted_pairs = nullptr;      // {code, original index} sorted by code (k_classify_packed)
    int32_t *d_sorted_idx = nullptr;
    uint8_t *d_wl_bytes = nullptr;
    int64_t *d_wl_off = nullptr;
    unsigned long long *d_counts = nullptr;
    uint32_t *d_hist_partial = nullptr;   // per slot kHistBlocks x n_wl private histograms (whitelists up to kHistLdsBins), else null:
                                          // the two slots' runs overlap on their streams, so each needs its own
    Slot slot[BZ_SLOTS];
    S1Buf s1[BZ_SLOTS];
    std::vector<char> wl_host;         // host copy of the whitelist (bz_assign_files writes barcodes from it)
    std::vector<int64_t> wl_off_host;
    // single-pair scratch
    uint8_t *d_pair = nullptr;
    int32_t *d_pair_out = nullptr;
    // comm
    void *comm = nullptr;
    double *d_f64 = nullptr;
    int64_t *d_i64 = nullptr;           // staging of bz_comm_allreduce_i64, grow-only
    int64_t i64_cap = 0;
};

namespace {

int use(bz_handle h) {
    if (!h) return fail(BZ_E_ARG, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    return BZ_OK;
}

// once per device, with that device current: the device address of `scratch` on THIS GPU
int scratch_init(int device) {
    if (device < 0 || device >= kMaxDevices) return fail(BZ_E_LIMIT, "device %d beyond the %d this library keeps scratch for", device, kMaxDevices);
    std::lock_guard<std::mutex> g(g_scratch_mu);
    if (g_scratch_ptr[device]) return BZ_OK;
    HIP_TRY(hipSetDevice(device));
    void *ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(BZ_E_HIP, "hipGetSymbolAddress(scratch) returned a null address on device %d", device);
    g_scratch_ptr[device] = (unsigned char *)ptr;
    return BZ_OK;
}

void free_whitelist(bz_handle h) {
    hipFree(h->d_planes); hipFree(h->d_sorted_codes); hipFree(h->d_sorted_idx); hipFree(h->d_sorted_pairs);
    h->d_sorted_pairs = nullptr;
    hipFree(h->d_wl_bytes); hipFree(h->d_wl_off); hipFree(h->d_counts); hipFree(h->d_hist_partial);
    h->d_hist_partial = nullptr;
    hipFree(h->d_grams); hipFree(h->d_planes32); hipFree(h->d_bucket); hipFree(h->d_bucket8);
    h->d_grams = nullptr; h->d_planes32 = nullptr; h->d_bucket = nullptr; h->d_bucket8 = nullptr;
    for (int k = 0; k < 3; ++k) {
        hipFree(h->d_seed_tab[k]); hipFree(h->d_seed_members[k]);
        h->d_seed_tab[k] = nullptr; h->d_seed_members[k] = nullptr;
    }
    hipFree(h->d_pair_bitmap); hipFree(h->d_pair_tab); hipFree(h->d_pair_members); hipFree(h->d_pair_tab8);
    h->d_pair_bitmap = nullptr; h->d_pair_tab = nullptr; h->d_pair_members = nullptr; h->d_pair_tab8 = nullptr;
    h->pair_cap = 0;
    hipFree(h->d_reg_to_orig); hipFree(h->d_irr_bytes); hipFree(h->d_irr_off); hipFree(h->d_irr_peq); hipFree(h->d_irr_meta);
    h->d_reg_to_orig = nullptr; h->d_irr_bytes = nullptr; h->d_irr_off = nullptr; h->d_irr_peq = nullptr; h->d_irr_meta = nullptr;
    h->mixed = 0; h->irr_only = 0; h->n_irr = 0; h->n_fast = 0;
    h->d_planes = nullptr; h->d_sorted_codes = nullptr; h->d_sorted_idx = nullptr;
    h->d_wl_bytes = nullptr; h->d_wl_off = nullptr; h->d_counts = nullptr;
    h->n_wl = -1;
}

void free_slot(Slot &s) {
    hipFree(s.d_ext); hipFree(s.d_len); hipFree(s.d_off); hipFree(s.d_bcl); hipFree(s.d_flags);
    hipFree(s.d_hit_buf[0]); hipFree(s.d_hit_buf[1]); hipFree(s.d_fast); hipFree(s.d_gen); hipFree(s.d_redo); hipFree(s.d_fast_tc); hipFree(s.d_ed); hipFree(s.d_end);
    hipHostFree(s.h_in); hipHostFree(s.h_out);
    hipFree(s.d_cand); hipFree(s.d_cand_cnt);
    s.d_cand = s.d_cand_cnt = nullptr; s.cand_words = 0; s.cand_stride = 0;
    s.d_ext = s.d_len = s.d_off = s.d_bcl = s.d_flags = nullptr;
    s.d_hit = s.d_hit_buf[0] = s.d_hit_buf[1] = s.d_fast = s.d_gen = s.d_redo = nullptr; s.d_fast_tc = nullptr;
    s.hit_cur = 0; s.pend = false;
    s.d_ed = s.d_end = nullptr;
    s.h_in = s.h_out = nullptr;
    s.cap = 0; s.cap_stride = 0;
}

void s1_free(S1Buf &b) {
    hipHostFree(b.h_seq); hipHostFree(b.h_qual); hipHostFree(b.h_off); hipHostFree(b.h_i32); hipHostFree(b.h_strand);
    hipFree(b.d_seq); hipFree(b.d_qual); hipFree(b.d_off); hipFree(b.d_i32); hipFree(b.d_strand);
    b = S1Buf();
}

int s1_ensure(S1Buf &b, hipStream_t st, int64_t bytes, int64_t reads) {
    if (bytes <= b.cap_bytes && reads <= b.cap_reads) return BZ_OK;
    HIP_TRY(hipStreamSynchronize(st));
    const int64_t cb = std::max<int64_t>({bytes + bytes / 4, b.cap_bytes, (int64_t)1 << 20});
    const int64_t cr = std::max<int64_t>({reads + reads / 4, b.cap_reads, (int64_t)1024});
    s1_free(b);
    HIP_TRY(hipHostMalloc(&b.h_seq, (size_t)cb, hipHostMallocDefault)); HIP_TRY(hipHostMalloc(&b.h_qual, (size_t)cb, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(&b.h_off, (size_t)(cr + 1) * sizeof(int64_t), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(&b.h_i32, (size_t)cr * 4 * sizeof(int32_t), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(&b.h_strand, (size_t)cr, hipHostMallocDefault));
    HIP_TRY(hipMalloc(&b.d_seq, (size_t)cb)); HIP_TRY(hipMalloc(&b.d_qual, (size_t)cb));
    HIP_TRY(hipMalloc(&b.d_off, (size_t)(cr + 1) * sizeof(int64_t)));
    HIP_TRY(hipMalloc(&b.d_i32, (size_t)cr * 4 * sizeof(int32_t))); HIP_TRY(hipMalloc(&b.d_strand, (size_t)cr));
    b.cap_bytes = cb; b.cap_reads = cr;
    return BZ_OK;
}

// reads [0, n) packed in the slot's pinned buffers (h_off[0] == 0) -> copies + k_stage1 + copies back, asynchronous
int s1_launch(bz_handle h, int slot, int64_t n, int kit5, int umi_len) {
    S1Buf &b = h->s1[slot];
    hipStream_t st = h->slot[slot].stream;
    b.n = n;
    if (n == 0) return BZ_OK;
    const int64_t total = b.h_off[n];
    if (total) {
        HIP_TRY(hipMemcpyAsync(b.d_seq, b.h_seq, (size_t)total, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(b.d_qual, b.h_qual, (size_t)total, hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipMemcpyAsync(b.d_off, b.h_off, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(bz::k_stage1, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, b.d_seq, b.d_qual, b.d_off, (int)n, kit5 ? 1 : 0, umi_len,
                       h->tie_rule, b.d_i32, b.d_strand, b.d_i32 + n, b.d_i32 + 2 * n, b.d_i32 + 3 * n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(b.h_i32, b.d_i32, (size_t)n * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(b.h_strand, b.d_strand, (size_t)n, hipMemcpyDeviceToHost, st));
    return BZ_OK;
}

// input staging layout: [ext n*stride][len n][off n][bcl n]; output: [hit 4n][ed n][end n][flags n]
int ensure_slot(Slot &s, int64_t n, int stride) {
    if (n <= s.cap && stride <= s.cap_stride) return BZ_OK;
    int64_t cap = std::max<int64_t>(n, s.cap);
    cap = std::max<int64_t>(cap, 1024);
    int cs = std::max(stride, s.cap_stride);
    free_slot(s);
    HIP_TRY(hipMalloc(&s.d_ext, (size_t)cap * cs));
    HIP_TRY(hipMalloc(&s.d_len, cap)); HIP_TRY(hipMalloc(&s.d_off, cap)); HIP_TRY(hipMalloc(&s.d_bcl, cap));
    HIP_TRY(hipMalloc(&s.d_flags, cap)); HIP_TRY(hipMalloc(&s.d_ed, cap)); HIP_TRY(hipMalloc(&s.d_end, cap));
    HIP_TRY(hipMalloc(&s.d_hit_buf[0], cap * 4)); HIP_TRY(hipMalloc(&s.d_hit_buf[1], cap * 4)); s.d_hit = s.d_hit_buf[0]; s.hit_cur = 0;
    HIP_TRY(hipMalloc(&s.d_fast, cap * 4)); HIP_TRY(hipMalloc(&s.d_gen, cap * 4));
    HIP_TRY(hipMalloc(&s.d_redo, cap * 4));
    HIP_TRY(hipMalloc(&s.d_fast_tc, cap * 8));
    HIP_TRY(hipHostMalloc(&s.h_in, (size_t)cap * (cs + 3), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(&s.h_out, (size_t)cap * 7, hipHostMallocDefault));
    s.cap = cap; s.cap_stride = cs;
    return BZ_OK;
}

// shapes the kernels assume, checked before any launch.  One branch-free pass (it vectorises: ~0.05 ms per 1M rows; the
// row-by-row form with its early returns cost ~1 ms and capped the boundary at ~1 G rows/s), rows named only on failure
int validate_rows(const uint8_t *ext_len, const uint8_t *bc_off, const uint8_t *bc_len, int64_t n, int ext_stride) {
    unsigned bad = 0;
    for (int64_t r = 0; r < n; ++r)
        bad |= (unsigned)(ext_len[r] > ext_stride) | (unsigned)((unsigned)bc_off[r] + (unsigned)bc_len[r] > (unsigned)ext_len[r]);
    if (!bad) return BZ_OK;
    for (int64_t r = 0; r < n; ++r) {
        if (ext_len[r] > ext_stride) return fail(BZ_E_ARG, "row %lld: ext_len %d > stride %d", (long long)r, ext_len[r], ext_stride);
        if ((int)bc_off[r] + bc_len[r] > ext_len[r]) return fail(BZ_E_ARG, "row %lld: barcode outside its extension", (long long)r);
    }
    return BZ_OK;
}

// Host copies between the caller's pageable arrays and the pinned staging of a slot (bz_assign_stage / bz_assign_wait):
// one thread moves ~25 GB/s, a 1M-row batch is 34 MB, and that copy -- not PCIe -- was what bounded callers with numpy
// arrays.  Large copies are cut into pieces for a small pool of threads that lives as long as the library; a caller that
// finds the pool busy (another engine's thread) copies by itself.
void par_memcpy(void *dst, const void *src, size_t n) {
    constexpr size_t kPiece = (size_t)2 << 20;
    constexpr int kThreads = 6;
    static bzio::Pool pool(kThreads);
    static std::mutex busy;
    if (n < 4 * kPiece || !busy.try_lock()) { memcpy(dst, src, n); return; }
    std::lock_guard<std::mutex> g(busy, std::adopt_lock);
    const int parts = (int)std::min<size_t>(4 * kThreads, n / kPiece);
    pool.run(parts, [&](int p) {
        const size_t lo = n * (size_t)p / parts, hi = n * (size_t)(p + 1) / parts;
        memcpy((char *)dst + lo, (const char *)src + lo, hi - lo);
    });
}

// is p inside pinned (page-locked, device-visible) host memory: hipHostMalloc / bz_host_alloc / hipHostRegister?
bool is_pinned_host(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

int check_slot(bz_handle h, int slot) {
    if (int rc = use(h)) return rc;
    if (slot < 0 || slot >= BZ_SLOTS) return fail(BZ_E_ARG, "slot %d out of range", slot);
    return BZ_OK;
}

template <int K>
void launch_scan(bz_handle h, Slot &s) {
    const int block = 256;
    const int grid = (int)((s.n + block - 1) / block);
    hipLaunchKernelGGL(bz::k_scan_bitparallel<K>, dim3(grid), dim3(block), 0, s.stream, (const uint4 *)h->d_planes, h->n_pairs, (int)h->n_fast,
                       s.d_ext, s.stride, s.d_len, s.d_fast, s.d_counters, h->tie_rule, s.d_hit, s.d_ed, s.d_end, s.d_flags);
}

template <int K>
void launch_scan_filtered(bz_handle h, Slot &s) {
    const int block = 256;
    const int grid = (int)((s.n + block - 1) / block);
    hipLaunchKernelGGL(bz::k_scan_filtered<K>, dim3(grid), dim3(block), 0, s.stream, (const uint4 *)h->d_grams, h->d_planes32, h->n_blk,
                       (int)h->n_fast, s.d_ext, s.stride, s.d_len, s.d_fast, s.d_counters, h->tie_rule, s.d_hit, s.d_ed, s.d_end, s.d_flags);
}

// seeded scan over fast_list (single-piece seeds); every row is finished here: the lane keeps the two smallest events per
// distance class, which is all the fold consumes, so nothing is handed to the ordered scan any more
template <int K>
void launch_scan_seeded(bz_handle h, Slot &s, int hist_slot, bool piggy, int32_t *publish) {
    const int block = 256;
    const int grid = (int)((s.n + block - 1) / block);
    const int plan = K == 2 ? 0 : K == 3 ? 2 : 1;
    // the pending histogram's second half rides in front of this grid (its first half rode in k_classify_packed's)
    const uint32_t *const red_partial = piggy ? h->d_hist_partial + (size_t)hist_slot * bz::kHistBlocks * h->n_wl : nullptr;
    const int red_blocks = piggy ? ((int)h->n_wl + block - 1) / block : 0;
    hipLaunchKernelGGL(bz::k_scan_seeded<K>, dim3(grid + red_blocks), dim3(block), 0, s.stream, h->d_seed_tab[plan], h->d_seed_members[plan],
                       h->d_planes32, (int)h->n_fast, s.d_ext, s.stride, s.d_len, s.d_fast, s.d_counters, h->tie_rule,
                       s.d_hit, s.d_ed, s.d_end, s.d_flags, red_partial, (int)h->n_wl, h->d_counts, s.pend_counters, publish);
    if (piggy) s.pend = false;
}

// what the workgroup-per-row scan of the irregular rows needs (scan_kernels.h: generic_row_wg)
bz::GenericArgs generic_args(bz_handle h, const Slot &s) {
    bz::GenericArgs a;
    a.wl_bytes = h->d_wl_bytes; a.wl_off = h->d_wl_off; a.n_wl = (int)h->n_wl;
    a.planes32 = h->d_planes32; a.n_fast = (int)h->n_fast; a.reg_to_orig = h->mixed ? h->d_reg_to_orig : nullptr;
    a.irr_meta = h->d_irr_meta; a.irr_bytes = h->d_irr_bytes; a.irr_off = h->d_irr_off; a.irr_peq = h->d_irr_peq;
    a.n_irr = (h->mixed || h->irr_only) ? h->n_irr : 0;
    a.ext = s.d_ext; a.ext_stride = s.stride; a.ext_len = s.d_len; a.bc_off = s.d_off; a.bc_len = s.d_bcl;
    a.tie = h->tie_rule; a.shortcut = h->shortcut;
    a.hit = s.d_hit; a.ed = s.d_ed; a.end = s.d_end; a.flags = s.d_flags;
    return a;
}

// pair-seeded scan, two-kernel form (whitelists whose candidate lists fit `pair_cap` slots per row)
int launch_pair_two_kernels(bz_handle h, Slot &s, int hist_slot, bool piggy, int32_t *publish) {
    const int stride = (int)((s.n + 63) / 64 * 64);
    const int64_t need = (int64_t)(h->pair_cap + 1) * stride;
    if (need > s.cand_words || !s.d_cand_cnt) {
        HIP_TRY(hipStreamSynchronize(s.stream));
        hipFree(s.d_cand); hipFree(s.d_cand_cnt);
        s.d_cand = s.d_cand_cnt = nullptr; s.cand_words = 0;
        const int64_t rows = std::max<int64_t>(s.cap, stride);
        HIP_TRY(hipMalloc(&s.d_cand, (size_t)(h->pair_cap + 1) * rows * sizeof(uint32_t)));   // + the dump row
        HIP_TRY(hipMalloc(&s.d_cand_cnt, (size_t)rows * sizeof(uint32_t)));
        s.cand_words = (int64_t)(h->pair_cap + 1) * rows;
    }
    s.cand_stride = stride;
    const int64_t groups = (s.n + bz::kCandRowsPerBlock - 1) / bz::kCandRowsPerBlock;
    const int ga = (int)std::min<int64_t>(groups, (int64_t)2 * h->n_cu);          // persistent: 2 workgroups of 12 waves per CU (24 of its 32 wave slots)
    const size_t lst_bytes = (size_t)(h->pair_cap + 1) * bz::kCandRowsPerBlock * sizeof(uint16_t);   // the trip's lists in LDS
    if (lst_bytes > h->cand_lds_set) {
        HIP_TRY(hipFuncSetAttribute((const void *)bz::k_pair_candidates, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lst_bytes));
        h->cand_lds_set = lst_bytes;
    }
    // the pending histogram's second half rides in front of this grid (its first half rode in k_classify_packed's)
    const uint32_t *const red_partial = piggy ? h->d_hist_partial + (size_t)hist_slot * bz::kHistBlocks * h->n_wl : nullptr;
    const int red_blocks = piggy ? ((int)h->n_wl + bz::kCandWThreads - 1) / bz::kCandWThreads : 0;
    hipLaunchKernelGGL(bz::k_pair_candidates, dim3(ga + red_blocks), dim3(bz::kCandWThreads), lst_bytes, s.stream, h->d_pair_bitmap, h->d_pair_tab8,
                       h->d_pair_members, (int)h->n_fast, (const uint2 *)s.d_fast_tc, s.d_counters, s.d_cand, h->pair_cap,
                       s.d_cand_cnt, red_partial, (int)h->n_wl, h->d_counts, s.pend_counters, publish);
    if (piggy) s.pend = false;
    if (h->profile) HIP_TRY(hipEventRecord(s.ev[5], s.stream));
    const int block = 256;
    const int grid = (int)((s.n + block - 1) / block);
    hipLaunchKernelGGL(bz::k_pair_verify<2>, dim3((unsigned)((s.n + bz::kVerifyThreads - 1) / bz::kVerifyThreads)), dim3(bz::kVerifyThreads), 0, s.stream, s.d_cand, h->pair_cap, s.d_cand_cnt, h->d_planes32,
                       (const uint2 *)s.d_fast_tc, s.d_fast, s.d_counters, h->tie_rule, s.d_redo, s.d_counters + 2, s.d_hit, s.d_ed,
                       s.d_end, s.d_flags);
    if (h->profile) { HIP_TRY(hipEventRecord(s.ev[6], s.stream)); s.timed_pair = true; }
    // hand-over rows (ordered filtered scan) and irregular rows (byte-wise scan) in ONE launch; bz_assign_run skips its
    // own k_scan_generic launch when tail_done is set
    hipLaunchKernelGGL(bz::k_pair_tail, dim3(std::min(grid, 2 * h->n_cu)), dim3(256), 0, s.stream, (const uint4 *)h->d_grams, h->d_planes32,
                       h->n_blk, (int)h->n_fast, s.d_redo, s.d_counters + 2, s.d_gen, s.d_counters + 1, generic_args(h, s));
    s.tail_done = true;
    return BZ_OK;
}

// pair-seeded scan straight out of the buckets (any whitelist size); hand-over rows as above
void launch_pair_stream(bz_handle h, Slot &s) {
    const int block = 256;
    const int grid = (int)((s.n + block - 1) / block);
    hipLaunchKernelGGL(bz::k_pair_stream<2>, dim3(grid), dim3(block), 0, s.stream, h->d_pair_tab, h->d_pair_members, h->d_planes32,
                       (int)h->n_fast, s.d_ext, s.stride, s.d_len, s.d_fast, s.d_counters, h->tie_rule, s.d_redo, s.d_counters + 2,
                       s.d_hit, s.d_ed, s.d_end, s.d_flags);
    hipLaunchKernelGGL(bz::k_scan_filtered<2>, dim3(grid), dim3(block), 0, s.stream, (const uint4 *)h->d_grams, h->d_planes32, h->n_blk,
                       (int)h->n_fast, s.d_ext, s.stride, s.d_len, s.d_redo, s.d_counters + 2, h->tie_rule, s.d_hit, s.d_ed, s.d_end, s.d_flags);
}

// mixed whitelists: remap + probe of the irregular entries, then the listed rows against the whole list (scan_kernels.h)
template <int K>
void launch_mixed(bz_handle h, Slot &s) {
    const int n = (int)s.n;
    hipLaunchKernelGGL(bz::k_mixed_fix<K>, dim3(std::min((n + 255) / 256, 8 * h->n_cu)), dim3(256), 0, s.stream, h->d_reg_to_orig, h->d_irr_peq,
                       h->d_irr_meta, h->d_irr_bytes, h->d_irr_off, h->n_irr, s.d_ext, s.stride, s.d_len, s.d_fast, s.d_counters, n, h->tie_rule,
                       s.d_redo, s.d_counters + 3, s.d_hit, s.d_flags);
    hipLaunchKernelGGL(bz::k_fix_rows<K>, dim3(2 * h->n_cu), dim3(256), 0, s.stream, h->d_reg_to_orig, h->d_planes32, (int)h->n_fast, h->d_irr_peq,
                       h->d_irr_meta, h->d_irr_bytes, h->d_irr_off, h->n_irr, s.d_ext, s.stride, s.d_len, s.d_redo, s.d_counters + 3, h->tie_rule,
                       s.d_hit, s.d_ed, s.d_end, s.d_flags);
}

// the striped totals behind counts[n_wl+2] -> counts[n_wl], counts[n_wl+1]
int fold_totals(bz_handle h, hipStream_t st) {
    hipLaunchKernelGGL(bz::k_fold_totals, dim3(1), dim3(64), 0, st, h->d_counts, (int)h->n_wl);
    HIP_TRY(hipGetLastError());
    return BZ_OK;
}

// the slot's pending histogram on its own (nobody to carry it: a reader of counts / counters / timings, or a next run
// that takes another path)
int flush_pending(bz_handle h, Slot &s, int hist_slot) {
    if (!s.pend) return BZ_OK;
    s.pend = false;
    uint32_t *const partial = h->d_hist_partial + (size_t)hist_slot * bz::kHistBlocks * h->n_wl;
    hipLaunchKernelGGL(bz::k_hist_partial, dim3(bz::kHistBlocks), dim3(bz::kHistThreads), (size_t)(h->n_wl + 1) * sizeof(uint32_t), s.stream,
                       s.pend_hit, s.pend_n, (int)h->n_wl, partial, h->d_counts);
    hipLaunchKernelGGL(bz::k_hist_reduce, dim3(((int)h->n_wl + 255) / 256), dim3(256), 0, s.stream, partial, (int)h->n_wl, h->d_counts,
                       s.pend_counters, s.h_counters_dev);
    HIP_TRY(hipGetLastError());
    return BZ_OK;
}

int flush_all(bz_handle h) {
    for (int i = 0; i < BZ_SLOTS; ++i)
        if (int rc = flush_pending(h, h->slot[i], i)) return rc;
    return BZ_OK;
}

// One run over the rows a Slot (or a Slot-shaped view) describes: classify + shortcut -> scan by max_ed -> byte-wise
// scan of the irregular rows -> per-barcode counts.  `lists_allowed` = the candidate-list form of the pair-seeded scan and
// the privatised histogram may be used (both need buffers the engine allocates); without it the scan verifies straight
// out of the buckets and the counts take global atomics.  The last kernel publishes the list counters to `publish` and
// zeroes them.  With `may_defer` (the engine's own slots) the counts are left pending for the slot's next run to carry
// along (Slot::pend), and a pending histogram of the previous run is carried along or flushed here.
int run_rows(bz_handle h, Slot &s, int hist_slot, int max_ed, bool lists_allowed, int32_t *publish, bool may_defer = false) {
    const int n = (int)s.n;
    const bool prof = s.timed;
    const bool packed = bz::packed_stride(s.stride);   // device buffers are hipMalloc'ed (256-byte aligned); bz_assign_device_noalloc checks the caller's
    // the paths that can carry a pending histogram: k_classify_packed takes its first half, the second rides in front of
    // k_pair_candidates' grid (the two-kernel pair-seeded scan) or k_scan_seeded's (max_ed 0, 1, 3; scan mode 3)
    const bool regular_path = h->wl_regular && !h->force_generic && max_ed <= bz::kMaxK;
    const bool seeded = (h->scan_mode == 0 || h->scan_mode == 3) && h->n_fast < ((int64_t)1 << 25);
    const bool pair_path = regular_path && seeded && max_ed == 2 && h->scan_mode == 0 && packed;      // pair seeds: lists + verify, or straight out of the buckets
    const bool pair2 = pair_path && h->pair_cap > 0 && !h->pair_fused && lists_allowed;
    const bool seeded_kernel = regular_path && seeded && max_ed <= 3 && !pair_path;                  // k_scan_seeded<max_ed>
    const bool deferrable = may_defer && !prof && h->d_hist_partial && h->n_wl <= bz::kPiggyBins;
    bool piggy = false;
    if (s.pend) {
        if ((pair2 || seeded_kernel) && packed && deferrable) piggy = true;
        else if (int rc = flush_pending(h, s, hist_slot)) return rc;
    }
    if (prof) HIP_TRY(hipEventRecord(s.ev[0], s.stream));
    {
        const int cblocks = (n + bz::kClassifyThreads * bz::kClassifyRows - 1) / (bz::kClassifyThreads * bz::kClassifyRows);
        const dim3 cgrid(cblocks + (piggy ? bz::kHistBlocks : 0));
        if (packed)
            hipLaunchKernelGGL(bz::k_classify_packed, cgrid, dim3(bz::kClassifyThreads), piggy ? (size_t)(h->n_wl + 1) * sizeof(uint32_t) : 0, s.stream,
                               s.d_ext, s.stride, s.d_len, s.d_off,
                               s.d_bcl, n, h->wl_regular, h->d_sorted_pairs, (int)h->n_fast, h->d_bucket, h->d_bucket8, h->bucket_bits, h->shortcut,
                               h->force_generic, max_ed, s.d_fast, (uint2 *)s.d_fast_tc, s.d_gen, s.d_counters, s.d_hit, s.d_ed, s.d_end, s.d_flags,
                               piggy ? s.pend_hit : nullptr, s.pend_n, (int)h->n_wl,
                               piggy ? h->d_hist_partial + (size_t)hist_slot * bz::kHistBlocks * h->n_wl : nullptr, h->d_counts);
        else
            hipLaunchKernelGGL(bz::k_classify, cgrid, dim3(bz::kClassifyThreads), 0, s.stream, s.d_ext, s.stride, s.d_len, s.d_off,
                               s.d_bcl, n, h->wl_regular, h->d_sorted_codes, h->d_sorted_idx, (int)h->n_fast, h->d_bucket, h->bucket_bits, h->shortcut,
                               h->force_generic, max_ed, s.d_fast, s.d_gen, s.d_counters, s.d_hit, s.d_ed, s.d_end, s.d_flags);
        HIP_TRY(hipGetLastError());
    }
    if (prof) HIP_TRY(hipEventRecord(s.ev[1], s.stream));
    if (h->wl_regular && !h->force_generic && max_ed <= bz::kMaxK) {
        // product path: pigeonhole seeds + verify (BZ_OPT_SCAN_MODE 0; max_ed 3: four 4-nt pieces); mode 2 = exact 3-gram
        // filter + verify over the whole whitelist (max_ed <= 2; its sets are too weak for 3); mode 1 = brute force
        // (event words hold the barcode index in 25 bits: larger whitelists take the filtered scan)
        const bool filtered = !seeded && h->scan_mode != 1 && max_ed <= 2;
        switch (max_ed) {
            case 0: if (seeded) launch_scan_seeded<0>(h, s, hist_slot, piggy, publish); else if (filtered) launch_scan_filtered<0>(h, s); else launch_scan<0>(h, s); break;
            case 1: if (seeded) launch_scan_seeded<1>(h, s, hist_slot, piggy, publish); else if (filtered) launch_scan_filtered<1>(h, s); else launch_scan<1>(h, s); break;
            case 2: if (pair_path) {
                        if (pair2) { if (int rc = launch_pair_two_kernels(h, s, hist_slot, piggy, publish)) return rc; }
                        else launch_pair_stream(h, s);
                    } else if (seeded) launch_scan_seeded<2>(h, s, hist_slot, piggy, publish);
                    else if (filtered) launch_scan_filtered<2>(h, s); else launch_scan<2>(h, s); break;
            case 3: if (seeded) launch_scan_seeded<3>(h, s, hist_slot, piggy, publish); else launch_scan<3>(h, s); break;
            // above 3 no seed plan is exact any more (a 16-mer has no piece that survives 4 edits): every pair through the
            // packed automaton, 40 (1 + K) instructions per pair
            case 4: launch_scan<4>(h, s); break;
            case 5: launch_scan<5>(h, s); break;
            case 6: launch_scan<6>(h, s); break;
            case 7: launch_scan<7>(h, s); break;
            default: launch_scan<8>(h, s); break;
        }
        HIP_TRY(hipGetLastError());
    }
    if (h->mixed && max_ed <= bz::kMaxK) {
        // indices of the regular sub-list -> caller's indices; scored rows within max_ed of an irregular entry -> fix list
        // (the hand-over list's buffer: its rows have been consumed by now; counter 3), re-scored against the whole list
        switch (max_ed) {
            case 0: launch_mixed<0>(h, s); break;
            case 1: launch_mixed<1>(h, s); break;
            case 2: launch_mixed<2>(h, s); break;
            case 3: launch_mixed<3>(h, s); break;
            case 4: launch_mixed<4>(h, s); break;
            case 5: launch_mixed<5>(h, s); break;
            case 6: launch_mixed<6>(h, s); break;
            case 7: launch_mixed<7>(h, s); break;
            default: launch_mixed<8>(h, s); break;
        }
        HIP_TRY(hipGetLastError());
    }
    if (prof) HIP_TRY(hipEventRecord(s.ev[2], s.stream));
    if (!s.tail_done) {
        if ((h->wl_regular || h->irr_only) && max_ed <= bz::kGenericMaxK) {
            // irregular rows (every row when max_ed is 4..8) against a list with the fast-path structures: a workgroup per
            // row, bit-parallel per entry
            const dim3 ggrid(std::min(n, 4 * h->n_cu));
            const bz::GenericArgs ga = generic_args(h, s);
#define BZ_GEN(K) case K: hipLaunchKernelGGL(bz::k_generic_rows<K>, ggrid, dim3(256), 0, s.stream, ga, s.d_gen, s.d_counters + 1); break
            switch (max_ed) { BZ_GEN(0); BZ_GEN(1); BZ_GEN(2); BZ_GEN(3); BZ_GEN(4); BZ_GEN(5); BZ_GEN(6); BZ_GEN(7); BZ_GEN(8); }
#undef BZ_GEN
        } else {
            // any list, any max_ed: one lane per row, byte-wise DP over the whole list (slow; irregular lists and max_ed > 8 only)
            hipLaunchKernelGGL(bz::k_scan_generic, dim3(std::min((n + 63) / 64, 16 * h->n_cu)), dim3(64), 0, s.stream, h->d_wl_bytes, h->d_wl_off, (int)h->n_wl,
                               s.d_ext, s.stride, s.d_len, s.d_off, s.d_bcl, s.d_gen, s.d_counters + 1, max_ed, h->tie_rule,
                               h->shortcut, s.d_hit, s.d_ed, s.d_end, s.d_flags);
        }
        HIP_TRY(hipGetLastError());
    }
    if (prof) HIP_TRY(hipEventRecord(s.ev[3], s.stream));
    if (piggy && s.pend) return fail(BZ_E_STATE, "a pending histogram was not carried along");   // (both carrier paths are always reached)
    if (deferrable) {
        s.pend = true; s.pend_hit = s.d_hit; s.pend_n = n; s.pend_counters = s.d_counters;
    } else if (h->d_hist_partial && lists_allowed) {
        uint32_t *const partial = h->d_hist_partial + (size_t)hist_slot * bz::kHistBlocks * h->n_wl;
        hipLaunchKernelGGL(bz::k_hist_partial, dim3(bz::kHistBlocks), dim3(bz::kHistThreads), (size_t)(h->n_wl + 1) * sizeof(uint32_t), s.stream,
                           s.d_hit, n, (int)h->n_wl, partial, h->d_counts);
        hipLaunchKernelGGL(bz::k_hist_reduce, dim3(((int)h->n_wl + 255) / 256), dim3(256), 0, s.stream, partial, (int)h->n_wl, h->d_counts,
                           s.d_counters, publish);
    } else {
        hipLaunchKernelGGL(bz::k_histogram, dim3(std::min((n + 255) / 256, 1024)), dim3(256), 0, s.stream, s.d_hit, n, (int)h->n_wl, h->d_counts,
                           s.d_counters, publish);
    }
    HIP_TRY(hipGetLastError());
    if (prof) HIP_TRY(hipEventRecord(s.ev[4], s.stream));
    return BZ_OK;
}

size_t counts_bytes(int64_t n_wl) { return (size_t)(n_wl + 2 + 2 * bz::kTotStripes) * sizeof(unsigned long long); }

}  // namespace

extern "C" {

int bz_abi_version(void) { return BZ_ABI_VERSION; }
const char *bz_last_error(void) { return g_err.c_str(); }

int bz_device_count(int *count) {
    if (!count) return fail(BZ_E_ARG, "null count");
    HIP_TRY(hipGetDeviceCount(count));
    return BZ_OK;
}

int bz_device_info(int device, int64_t info[8], char *arch, int arch_cap) {
    if (!info) return fail(BZ_E_ARG, "null info");
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    memset(info, 0, 8 * sizeof(int64_t));
    info[0] = p.multiProcessorCount; info[1] = p.clockRate; info[2] = p.warpSize;
    info[3] = (int64_t)(p.totalGlobalMem >> 20); info[4] = (int64_t)p.maxSharedMemoryPerMultiProcessor;
    if (arch && arch_cap > 0) { strncpy(arch, p.gcnArchName, arch_cap - 1); arch[arch_cap - 1] = 0; }
    return BZ_OK;
}

int bz_device_identity(int device, char *out, int cap) {
    if (!out || cap < 64) return fail(BZ_E_ARG, "identity buffer of at least 64 bytes needed");
    char bus[32] = {0};
    HIP_TRY(hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    // PCI address + the UUID bytes as hex: two ranks that print the same string share one physical GPU
    int k = snprintf(out, (size_t)cap, "%s/", bus);
    for (int i = 0; i < 16 && k + 2 < cap; ++i) k += snprintf(out + k, (size_t)(cap - k), "%02x", (unsigned)(uint8_t)p.uuid.bytes[i]);
    return BZ_OK;
}

static int create_buffers(bz_engine *h) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, h->device));
    h->n_cu = std::max(1, prop.multiProcessorCount);
    HIP_TRY(hipFuncSetAttribute((const void *)bz::k_hist_partial, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((bz::kHistLdsBins + 1) * sizeof(uint32_t))));
    for (Slot &s : h->slot) {
        HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        for (hipEvent_t &e : s.ev) HIP_TRY(hipEventCreate(&e));
        HIP_TRY(hipMalloc(&s.d_counters_base, 8 * sizeof(int32_t)));
        HIP_TRY(hipMemset(s.d_counters_base, 0, 8 * sizeof(int32_t)));
        s.d_counters = s.d_counters_base;
        HIP_TRY(hipHostMalloc(&s.h_counters, 4 * sizeof(int32_t), hipHostMallocMapped));
        memset(s.h_counters, 0, 4 * sizeof(int32_t));
        HIP_TRY(hipHostGetDevicePointer((void **)&s.h_counters_dev, s.h_counters, 0));
    }
    HIP_TRY(hipMalloc(&h->d_pair, 4096));
    HIP_TRY(hipMalloc(&h->d_pair_out, 2 * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&h->d_f64, 64 * sizeof(double)));
    return BZ_OK;
}

int bz_create(int device, bz_handle *out) {
    if (!out) return fail(BZ_E_ARG, "null out");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(BZ_E_HIP, "device %d not present (%d visible): no GPU, no compute", device, n);
    HIP_TRY(hipSetDevice(device));
    if (int rc = scratch_init(device)) return rc;
    bz_engine *h = new bz_engine();
    h->device = device;
    if (int rc = create_buffers(h)) {                 // what was made so far goes again; the message of the failed call stays
        const std::string why = g_err;
        bz_destroy(h);
        g_err = why;
        return rc;
    }
    *out = h;
    return BZ_OK;
}

int bz_destroy(bz_handle h) {
    if (!h) return BZ_OK;
    hipSetDevice(h->device);
    hipDeviceSynchronize();
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    free_whitelist(h);
    for (S1Buf &b : h->s1) s1_free(b);
    for (Slot &s : h->slot) {
        free_slot(s);
        hipFree(s.d_counters_base); hipHostFree(s.h_counters);
        for (hipEvent_t &e : s.ev) if (e) hipEventDestroy(e);
        if (s.stream) hipStreamDestroy(s.stream);
    }
    hipFree(h->d_pair); hipFree(h->d_pair_out); hipFree(h->d_f64); hipFree(h->d_i64);
    delete h;
    return BZ_OK;
}

int bz_set_option(bz_handle h, int option, int64_t value) {
    if (!h) return fail(BZ_E_ARG, "null handle");
    switch (option) {
        case BZ_OPT_TIE_RULE: if (value != 0 && value != 1) return fail(BZ_E_ARG, "tie rule must be 0 or 1"); h->tie_rule = (int)value; return BZ_OK;
        case BZ_OPT_SHORTCUT: h->shortcut = value ? 1 : 0; return BZ_OK;
        case BZ_OPT_FORCE_GENERIC: h->force_generic = value ? 1 : 0; return BZ_OK;
        case BZ_OPT_SCAN_MODE: if (value < 0 || value > 3) return fail(BZ_E_ARG, "scan mode must be 0..3"); h->scan_mode = (int)value; return BZ_OK;
        case BZ_OPT_PAIR_FUSED: h->pair_fused = value ? 1 : 0; return BZ_OK;
        case BZ_OPT_PROFILE: h->profile = value ? 1 : 0; return BZ_OK;
        case BZ_OPT_GZ_LEVEL: if (value < 0 || value > 9) return fail(BZ_E_ARG, "gz level must be 0..9"); h->gz_level = (int)value; return BZ_OK;
    }
    return fail(BZ_E_ARG, "unknown option %d", option);
}

int bz_set_whitelist(bz_handle h, const char *bytes, const int64_t *offsets, int64_t n) {
    if (int rc = use(h)) return rc;
    if (!offsets || n < 0 || (n > 0 && !bytes)) return fail(BZ_E_ARG, "bad whitelist arguments");
    if (n > (int64_t)1 << 30) return fail(BZ_E_LIMIT, "whitelist of %lld barcodes exceeds 2^30", (long long)n);
    int64_t longest = 0;
    for (int64_t k = 0; k < n; ++k) {
        if (offsets[k + 1] < offsets[k]) return fail(BZ_E_ARG, "offsets not monotone at %lld", (long long)k);
        longest = std::max(longest, offsets[k + 1] - offsets[k]);
        // an entry longer than the row is the TEXT of the replaced call and the end index a position inside it: int8 results
        if (offsets[k + 1] - offsets[k] > 127) return fail(BZ_E_LIMIT, "whitelist entry %lld has %lld characters: at most 127", (long long)k, (long long)(offsets[k + 1] - offsets[k]));
    }
    if (h->n_wl >= 0) { if (int rc = flush_all(h)) return rc; }     // pending counts belong to the list that is about to go (and their list counters must be zeroed)
    HIP_TRY(hipDeviceSynchronize());
    free_whitelist(h);

    const int64_t total = n ? offsets[n] - offsets[0] : 0;
    // entries of the packed shape (16-nt ACGT) in caller order; a FEW others beside them keep the fast path (k_mixed_fix),
    // more than kMaxIrregular (or no regular entry at all) send every row through the byte-wise scan
    constexpr int64_t kMaxIrregular = 65536;          // (k_mixed_fix probes every scored row against each of them: ~6 us per entry per 1M rows)
    std::vector<int32_t> reg_pos, irr_pos;
    for (int64_t k = 0; k < n; ++k) {
        bool ok = offsets[k + 1] - offsets[k] == bz::kBcLen;
        const char *w = bytes + offsets[k];
        for (int i = 0; ok && i < bz::kBcLen; ++i) ok = w[i] == 'A' || w[i] == 'C' || w[i] == 'G' || w[i] == 'T';
        if (ok) reg_pos.push_back((int32_t)k);
        else if ((int64_t)irr_pos.size() <= kMaxIrregular) irr_pos.push_back((int32_t)k);
    }
    const int64_t nf = (int64_t)reg_pos.size();
    // (an empty entry -- a blank line of the list -- is one of the irregular entries: at distance 0 from every text and a
    // falsy hit in the reference's fold, fold_update_falsy; every scored row is then re-scored by k_fix_rows)
    // (entries are at most 127 characters, checked above: k_fix_rows' event words hold end + 1 in 8 bits; the index bound is
    // kept from the 32-bit event words of earlier builds)
    constexpr int64_t kMaxFastEntry = 127;
    const bool regular = nf > 0 && (nf == n || (n - nf <= kMaxIrregular && n < ((int64_t)1 << 23) && longest <= kMaxFastEntry));
    const bool mixed = regular && nf < n;
    // byte form (generic kernel + histogram size)
    std::vector<int64_t> off(n + 1);
    for (int64_t k = 0; k <= n; ++k) off[k] = offsets[k] - offsets[0];
    HIP_TRY(hipMalloc(&h->d_wl_bytes, std::max<int64_t>(total, 1)));
    HIP_TRY(hipMalloc(&h->d_wl_off, (n + 1) * sizeof(int64_t)));
    if (total) HIP_TRY(hipMemcpy(h->d_wl_bytes, bytes + offsets[0], total, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_wl_off, off.data(), (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&h->d_counts, counts_bytes(n)));
    HIP_TRY(hipMemset(h->d_counts, 0, counts_bytes(n)));
    if (n > 0 && n <= bz::kHistLdsBins) HIP_TRY(hipMalloc(&h->d_hist_partial, (size_t)BZ_SLOTS * bz::kHistBlocks * n * sizeof(uint32_t)));

    h->wl_host.assign(bytes + (n ? offsets[0] : 0), bytes + (n ? offsets[0] : 0) + total);
    h->wl_off_host = off;
    h->wl_regular = regular ? 1 : 0;
    h->n_pairs = 0;
    if (regular) {
        // 2-bit planes, two barcodes per 32-bit word, padded to a multiple of 2 pairs with copies of
        // barcode 0 (indices >= nf are ignored by the kernel).
        const int64_t n_pairs = ((nf + 3) / 4) * 2;
        std::vector<uint2> planes(n_pairs);
        std::vector<uint32_t> codes(nf);
        auto plane16 = [&](int64_t k, uint32_t &hi, uint32_t &lo, uint32_t &code) {
            const char *w = bytes + offsets[reg_pos[k < nf ? k : 0]];
            hi = lo = code = 0;
            for (int i = 0; i < bz::kBcLen; ++i) {
                uint32_t c = ((uint32_t)(unsigned char)w[i] >> 1) & 3u;
                hi |= ((c >> 1) & 1u) << i; lo |= (c & 1u) << i; code |= c << (2 * i);
            }
        };
        for (int64_t p = 0; p < n_pairs; ++p) {
            uint32_t h0, l0, c0, h1, l1, c1;
            plane16(2 * p, h0, l0, c0); plane16(2 * p + 1, h1, l1, c1);
            planes[p] = make_uint2(h0 | (h1 << 16), l0 | (l1 << 16));
            if (2 * p < nf) codes[2 * p] = c0;
            if (2 * p + 1 < nf) codes[2 * p + 1] = c1;
        }
        std::vector<int32_t> idx(nf);
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return codes[a] < codes[b]; });
        std::vector<uint32_t> sorted(nf);
        for (int64_t k = 0; k < nf; ++k) sorted[k] = codes[idx[k]];
        HIP_TRY(hipMalloc(&h->d_planes, n_pairs * sizeof(uint2)));
        HIP_TRY(hipMalloc(&h->d_sorted_codes, nf * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&h->d_sorted_idx, nf * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(h->d_planes, planes.data(), n_pairs * sizeof(uint2), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->d_sorted_codes, sorted.data(), nf * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->d_sorted_idx, idx.data(), nf * sizeof(int32_t), hipMemcpyHostToDevice));
        {
            std::vector<uint2> pairs(nf);
            for (int64_t k = 0; k < nf; ++k) pairs[k] = make_uint2(sorted[k], (uint32_t)idx[k]);
            HIP_TRY(hipMalloc(&h->d_sorted_pairs, nf * sizeof(uint2)));
            HIP_TRY(hipMemcpy(h->d_sorted_pairs, pairs.data(), nf * sizeof(uint2), hipMemcpyHostToDevice));
        }
        h->n_pairs = (int)n_pairs;

        // bucket table for the shortcut look-up: ~4 buckets per barcode, 2^8 .. 2^24 buckets
        int bits = 8;
        while (bits < 24 && ((int64_t)1 << bits) < 4 * nf) ++bits;
        const int64_t nb = (int64_t)1 << bits;
        std::vector<int32_t> bucket(nb + 1);
        {
            int64_t k = 0;
            for (int64_t b = 0; b <= nb; ++b) {
                const uint64_t lim = (uint64_t)b << (32 - bits);     // first code of bucket b
                while (k < nf && (uint64_t)sorted[k] < lim) ++k;
                bucket[b] = (int32_t)k;
            }
        }
        HIP_TRY(hipMalloc(&h->d_bucket, (nb + 1) * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(h->d_bucket, bucket.data(), (nb + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
        h->bucket_bits = bits;
        {   // one-gather form: the bucket's first barcode {code, index | (more than one barcode) << 31}; index all ones = empty
            std::vector<uint2> b8(nb);
            for (int64_t b = 0; b < nb; ++b) {
                const int32_t lo = bucket[b], hi = bucket[b + 1];
                b8[b] = hi == lo ? make_uint2(0u, 0xFFFFFFFFu) : make_uint2(sorted[lo], (uint32_t)idx[lo] | (hi - lo > 1 ? 0x80000000u : 0u));
            }
            HIP_TRY(hipMalloc(&h->d_bucket8, nb * sizeof(uint2)));
            HIP_TRY(hipMemcpy(h->d_bucket8, b8.data(), nb * sizeof(uint2), hipMemcpyHostToDevice));
        }

        // filtered scan: 64-bit 3-gram set per barcode (padding = all ones: never passes), planes32
        const int64_t n_blk = (nf + bz::kFilterBlock - 1) / bz::kFilterBlock;
        std::vector<uint2> grams(n_blk * bz::kFilterBlock, make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu));
        std::vector<uint32_t> planes32(nf);
        for (int64_t k = 0; k < nf; ++k) {
            const uint32_t c = codes[k];
            uint64_t set = 0;
            for (int i = 0; i + 3 <= bz::kBcLen; ++i) set |= (uint64_t)1 << ((c >> (2 * i)) & 63u);
            grams[k] = make_uint2((uint32_t)set, (uint32_t)(set >> 32));
            uint32_t hi = 0, lo = 0;
            for (int i = 0; i < bz::kBcLen; ++i) { hi |= ((c >> (2 * i + 1)) & 1u) << i; lo |= ((c >> (2 * i)) & 1u) << i; }
            planes32[k] = hi | (lo << 16);
        }
        HIP_TRY(hipMalloc(&h->d_grams, grams.size() * sizeof(uint2)));
        HIP_TRY(hipMalloc(&h->d_planes32, nf * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(h->d_grams, grams.data(), grams.size() * sizeof(uint2), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->d_planes32, planes32.data(), nf * sizeof(uint32_t), hipMemcpyHostToDevice));
        h->n_blk = (int)n_blk;

        // seed index (k_scan_seeded): per piece a CSR bucket table over the piece's 2-bit code
        const int plan_n[3] = {3, 2, 4};
        const int plan_len[3][4] = {{5, 5, 6, 0}, {8, 8, 0, 0}, {4, 4, 4, 4}}, plan_off[3][4] = {{0, 5, 10, 0}, {0, 8, 0, 0}, {0, 4, 8, 12}};
        for (int pl = 0; pl < 3; ++pl) {
            size_t tab_size = 0;
            for (int q = 0; q < plan_n[pl]; ++q) tab_size += ((size_t)1 << (2 * plan_len[pl][q])) + 1;
            std::vector<int32_t> tab(tab_size), mem((size_t)plan_n[pl] * nf);
            size_t base = 0;
            for (int q = 0; q < plan_n[pl]; ++q) {
                const size_t nb = (size_t)1 << (2 * plan_len[pl][q]);
                const uint32_t mask = (uint32_t)(nb - 1);
                int32_t *t = tab.data() + base;
                std::fill(t, t + nb + 1, 0);
                for (int64_t k = 0; k < nf; ++k) ++t[((codes[k] >> (2 * plan_off[pl][q])) & mask) + 1];
                for (size_t b = 0; b < nb; ++b) t[b + 1] += t[b];
                std::vector<int32_t> fill(t, t + nb);
                int32_t *m = mem.data() + (size_t)q * nf;
                for (int64_t k = 0; k < nf; ++k) m[fill[(codes[k] >> (2 * plan_off[pl][q])) & mask]++] = (int32_t)k;
                base += nb + 1;
            }
            HIP_TRY(hipMalloc(&h->d_seed_tab[pl], tab.size() * sizeof(int32_t)));
            HIP_TRY(hipMalloc(&h->d_seed_members[pl], mem.size() * sizeof(int32_t)));
            HIP_TRY(hipMemcpy(h->d_seed_tab[pl], tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(h->d_seed_members[pl], mem.data(), mem.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }

        // pair-seed index (max_ed 2): the barcode is cut into four 4-nt pieces; per pair of pieces a
        // CSR table over the 16-bit key + an occupancy bitmap the kernel keeps in LDS
        {
            const int pa[bz::kPairCount] = {0, 0, 0, 1, 1, 2}, pb[bz::kPairCount] = {1, 2, 3, 2, 3, 3};
            std::vector<uint32_t> bitmap((size_t)bz::kPairCount * bz::kPairBmWords, 0u);
            std::vector<int32_t> tab((size_t)bz::kPairCount * bz::kPairTab);
            std::vector<uint2> mem((size_t)bz::kPairCount * nf);
            for (int p = 0; p < bz::kPairCount; ++p) {
                auto key = [&](int64_t k) { return (((codes[k] >> (8 * pa[p])) & 0xFFu) << 8) | ((codes[k] >> (8 * pb[p])) & 0xFFu); };
                int32_t *t = tab.data() + (size_t)p * bz::kPairTab;
                std::fill(t, t + bz::kPairTab, 0);
                for (int64_t k = 0; k < nf; ++k) {
                    const uint32_t ky = key(k);
                    ++t[ky + 1];
                    bitmap[(size_t)p * bz::kPairBmWords + (ky >> 5)] |= 1u << (ky & 31u);
                }
                for (int b = 0; b < bz::kPairTab - 1; ++b) t[b + 1] += t[b];
                std::vector<int32_t> fill(t, t + bz::kPairTab - 1);
                uint2 *m = mem.data() + (size_t)p * nf;
                for (int64_t k = 0; k < nf; ++k) m[fill[key(k)]++] = make_uint2((uint32_t)k, planes32[k]);
            }
            HIP_TRY(hipMalloc(&h->d_pair_bitmap, bitmap.size() * sizeof(uint32_t)));
            HIP_TRY(hipMalloc(&h->d_pair_tab, tab.size() * sizeof(int32_t)));
            HIP_TRY(hipMalloc(&h->d_pair_members, mem.size() * sizeof(uint2)));
            HIP_TRY(hipMemcpy(h->d_pair_bitmap, bitmap.data(), bitmap.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(h->d_pair_tab, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(h->d_pair_members, mem.data(), mem.size() * sizeof(uint2), hipMemcpyHostToDevice));
            // two-kernel form: 8-byte table entries and a per-row candidate capacity of the expected number of candidates
            // (131 probes x nf / 65536 barcodes per bucket by chance, + 6 for a true barcode met through several pairs)
            // + 5 sigma + 8: 48 at 5k cells, 64 at 10k, 88 at 20k, 144 at 40k (6.5 sigma or more once rounded).  A row above
            // it (the rows of low-complexity whitelists) is handed to the ordered scan, which costs a whole-whitelist
            // pass: milliseconds at 40k barcodes, so the capacity must not be tight (120 at 40k cells: 58 of 586K rows
            // over it, step 0.82 -> 4.9 ms).  The capacity also sizes k_pair_candidates' lists in LDS: up to 93 rows two
            // workgroups share a CU (the old 2 x chance + 40 = 136 rows at 20k cells allowed one, and configs[2] ran
            // 17 % slower than with the lists in global memory).  Indices are 16 bits there: nf < 48K.
            const double chance = 131.0 * (double)nf / 65536.0 + 6.0;
            const int cap = ((int)std::ceil(chance + 5.0 * std::sqrt(chance) + 8.0) + 7) / 8 * 8;
            if (nf < 49152) {
                std::vector<uint2> tab8((size_t)bz::kPairCount * 65536);
                for (int p = 0; p < bz::kPairCount; ++p) {
                    const int32_t *t = tab.data() + (size_t)p * bz::kPairTab;
                    const uint2 *m = mem.data() + (size_t)p * nf;
                    for (int b = 0; b < 65536; ++b) {
                        const uint32_t c = (uint32_t)(t[b + 1] - t[b]);       // {size (saturating) | offset << 8, first member}
                        tab8[(size_t)p * 65536 + b] = make_uint2(std::min(c, 255u) | ((uint32_t)t[b] << 8), c ? m[t[b]].x : 0u);
                    }
                }
                HIP_TRY(hipMalloc(&h->d_pair_tab8, tab8.size() * sizeof(uint2)));
                HIP_TRY(hipMemcpy(h->d_pair_tab8, tab8.data(), tab8.size() * sizeof(uint2), hipMemcpyHostToDevice));
                h->pair_cap = cap;
            }
        }
        h->n_fast = nf;
    }
    // the irregular entries: bytes, four match masks and {length, caller's index} each -- beside regular entries (mixed), or
    // the whole of a list that has no regular entry
    const bool irr_only = !regular && nf == 0 && n > 0 && n <= kMaxIrregular;
    if (mixed || irr_only) {
        {
            std::vector<char> ib;
            std::vector<int64_t> io(1, 0);
            std::vector<uint4> peq;
            std::vector<int2> meta;
            for (int32_t k : irr_pos) {
                ib.insert(ib.end(), bytes + offsets[k], bytes + offsets[k + 1]);
                io.push_back((int64_t)ib.size());
                const int m = (int)(offsets[k + 1] - offsets[k]);
                uint32_t q[4] = {0, 0, 0, 0};                       // code order A C T G ((ascii >> 1) & 3)
                for (int i = 0; i < std::min(m, 32); ++i) {
                    const char c = bytes[offsets[k] + i];
                    if (c == 'A' || c == 'C' || c == 'G' || c == 'T') q[((unsigned char)c >> 1) & 3u] |= 1u << i;
                }
                peq.push_back(make_uint4(q[0], q[1], q[2], q[3]));
                meta.push_back(make_int2(m, k));
            }
            HIP_TRY(hipMalloc(&h->d_irr_peq, peq.size() * sizeof(uint4)));
            HIP_TRY(hipMalloc(&h->d_irr_meta, meta.size() * sizeof(int2)));
            HIP_TRY(hipMemcpy(h->d_irr_peq, peq.data(), peq.size() * sizeof(uint4), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(h->d_irr_meta, meta.data(), meta.size() * sizeof(int2), hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc(&h->d_reg_to_orig, std::max<int64_t>(nf, 1) * sizeof(int32_t)));
            HIP_TRY(hipMalloc(&h->d_irr_bytes, std::max<size_t>(ib.size(), 1)));
            HIP_TRY(hipMalloc(&h->d_irr_off, io.size() * sizeof(int64_t)));
            if (nf) HIP_TRY(hipMemcpy(h->d_reg_to_orig, reg_pos.data(), nf * sizeof(int32_t), hipMemcpyHostToDevice));
            if (!ib.empty()) HIP_TRY(hipMemcpy(h->d_irr_bytes, ib.data(), ib.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(h->d_irr_off, io.data(), io.size() * sizeof(int64_t), hipMemcpyHostToDevice));
            h->mixed = mixed ? 1 : 0; h->irr_only = irr_only ? 1 : 0; h->n_irr = (int)irr_pos.size();
        }
    }
    h->n_wl = n;
    return BZ_OK;
}

int bz_assign_stage(bz_handle h, int slot, const uint8_t *ext, int ext_stride, const uint8_t *ext_len,
                    const uint8_t *bc_off, const uint8_t *bc_len, int64_t n) {
    if (int rc = check_slot(h, slot)) return rc;
    if (n < 0 || n > (int64_t)1 << 30) return fail(BZ_E_LIMIT, "row count %lld out of range", (long long)n);
    if (n > 0 && (!ext || !ext_len || !bc_off || !bc_len)) return fail(BZ_E_ARG, "null row arrays");
    if (ext_stride < 1 || ext_stride > BZ_EXT_MAX) return fail(BZ_E_LIMIT, "ext_stride %d outside 1..%d", ext_stride, BZ_EXT_MAX);
    if (int rc = validate_rows(ext_len, bc_off, bc_len, n, ext_stride)) return rc;
    Slot &s = h->slot[slot];
    if (n > s.cap || ext_stride > s.cap_stride) { if (int rc = flush_pending(h, s, slot)) return rc; }   // the buffers are about to be replaced
    HIP_TRY(hipStreamSynchronize(s.stream));
    if (int rc = ensure_slot(s, n, ext_stride)) return rc;
    s.n = n; s.stride = ext_stride; s.staged = false; s.ran = false; s.direct_out = false;
    if (n) {
        uint8_t *p = s.h_in;
        par_memcpy(p, ext, (size_t)n * ext_stride); p += (size_t)n * ext_stride;
        memcpy(p, ext_len, n); p += n;
        memcpy(p, bc_off, n); p += n;
        memcpy(p, bc_len, n);
        p = s.h_in;
        HIP_TRY(hipMemcpyAsync(s.d_ext, p, (size_t)n * ext_stride, hipMemcpyHostToDevice, s.stream)); p += (size_t)n * ext_stride;
        HIP_TRY(hipMemcpyAsync(s.d_len, p, n, hipMemcpyHostToDevice, s.stream)); p += n;
        HIP_TRY(hipMemcpyAsync(s.d_off, p, n, hipMemcpyHostToDevice, s.stream)); p += n;
        HIP_TRY(hipMemcpyAsync(s.d_bcl, p, n, hipMemcpyHostToDevice, s.stream));
    }
    s.staged = true;
    return BZ_OK;
}

int bz_assign_run(bz_handle h, int slot, int max_ed) {
    if (int rc = check_slot(h, slot)) return rc;
    Slot &s = h->slot[slot];
    if (h->n_wl < 0) return fail(BZ_E_STATE, "bz_set_whitelist has not been called");
    if (!s.staged) return fail(BZ_E_STATE, "slot %d has no staged rows", slot);
    if (max_ed < 0 || max_ed > BZ_EXT_MAX) return fail(BZ_E_ARG, "max_ed %d outside 0..%d", max_ed, BZ_EXT_MAX);
    // HIP events between the kernels cost ~6 us of idle GPU each: recorded only when BZ_OPT_PROFILE is on
    s.timed = h->profile != 0;
    s.timed_pair = false;
    s.tail_done = false;
    if (s.n == 0) { memset(s.h_counters, 0, 4 * sizeof(int32_t)); s.timed = false; s.ran = true; return BZ_OK; }
    // this run's hits and list counters go to the buffers the previous run did NOT use: its histogram may still be pending
    s.hit_cur ^= 1;
    s.d_hit = s.d_hit_buf[s.hit_cur];
    s.d_counters = s.d_counters_base + (s.d_counters == s.d_counters_base ? 4 : 0);
    if (int rc = run_rows(h, s, slot, max_ed, /*lists_allowed=*/true, s.h_counters_dev, /*may_defer=*/true)) return rc;
    s.ran = true;
    return BZ_OK;
}

// The run of bz_assign_run over rows that are ALREADY in device memory, results left in device memory, with every work
// list in the library's file-scope scratch array: no hipMalloc / hipFree / allocating call, asynchronous on `stream`.
int bz_assign_device_noalloc(bz_handle h, const uint8_t *d_ext, int ext_stride, const uint8_t *d_ext_len, const uint8_t *d_bc_off,
                             const uint8_t *d_bc_len, int64_t n, int max_ed, int32_t *d_hit, int8_t *d_ed, int8_t *d_end_idx,
                             uint8_t *d_flags, void *stream) {
    if (int rc = use(h)) return rc;
    if (h->n_wl < 0) return fail(BZ_E_STATE, "bz_set_whitelist has not been called");
    if (n < 0) return fail(BZ_E_ARG, "negative row count");
    if (n > (int64_t)BZ_NOALLOC_MAX_ROWS)
        return fail(BZ_E_LIMIT, "bz_assign_device_noalloc takes at most %d rows per call (got %lld): split the batch or use bz_assign_submit",
                    BZ_NOALLOC_MAX_ROWS, (long long)n);
    if (max_ed < 0 || max_ed > BZ_EXT_MAX) return fail(BZ_E_ARG, "max_ed %d outside 0..%d", max_ed, BZ_EXT_MAX);
    if (ext_stride < 1 || ext_stride > BZ_EXT_MAX) return fail(BZ_E_LIMIT, "ext_stride %d outside 1..%d", ext_stride, BZ_EXT_MAX);
    if (n == 0) return BZ_OK;
    if (!d_ext || !d_ext_len || !d_bc_off || !d_bc_len || !d_hit || !d_ed || !d_end_idx || !d_flags) return fail(BZ_E_ARG, "null device array");
    if ((ext_stride & 7) == 0 && ((uintptr_t)d_ext & 7) != 0) return fail(BZ_E_ARG, "d_ext must be 8-byte aligned for a stride that is a multiple of 8");
    if (((uintptr_t)d_hit & 3) != 0) return fail(BZ_E_ARG, "d_hit must be 4-byte aligned");
    unsigned char *const base = g_scratch_ptr[h->device];     // this GPU's copy of `scratch` (scratch_init, bz_create)
    if (!base) return fail(BZ_E_STATE, "no scratch address for device %d", h->device);
    Slot s;                                                    // a view: caller's arrays + the scratch lists, nothing owned
    s.stream = stream ? (hipStream_t)stream : h->slot[0].stream;
    s.n = n; s.stride = ext_stride;
    s.d_ext = const_cast<uint8_t *>(d_ext); s.d_len = const_cast<uint8_t *>(d_ext_len);
    s.d_off = const_cast<uint8_t *>(d_bc_off); s.d_bcl = const_cast<uint8_t *>(d_bc_len);
    s.d_hit = d_hit; s.d_ed = d_ed; s.d_end = d_end_idx; s.d_flags = d_flags;
    s.d_fast = (int32_t *)(base + kNaOffFast); s.d_gen = (int32_t *)(base + kNaOffGen); s.d_redo = (int32_t *)(base + kNaOffRedo);
    s.d_fast_tc = base + kNaOffTc;
    int32_t *const counters = (int32_t *)(base + kNaOffCounters);
    s.d_counters = counters;
    // the array keeps the last call's contents: the list counters must start at zero
    HIP_TRY(hipMemsetAsync(counters, 0, 16 * sizeof(int32_t), s.stream));
    return run_rows(h, s, 0, max_ed, /*lists_allowed=*/false, counters + 8);
}

static int start_fetch(Slot &s) {
    const int64_t n = s.n;
    if (!n) return BZ_OK;
    uint8_t *p = s.h_out;
    HIP_TRY(hipMemcpyAsync(p, s.d_hit, n * 4, hipMemcpyDeviceToHost, s.stream)); p += n * 4;
    HIP_TRY(hipMemcpyAsync(p, s.d_ed, n, hipMemcpyDeviceToHost, s.stream)); p += n;
    HIP_TRY(hipMemcpyAsync(p, s.d_end, n, hipMemcpyDeviceToHost, s.stream)); p += n;
    HIP_TRY(hipMemcpyAsync(p, s.d_flags, n, hipMemcpyDeviceToHost, s.stream));
    return BZ_OK;
}

static int finish_fetch(Slot &s, int32_t *hit, int8_t *ed, int8_t *end_idx, uint8_t *flags) {
    HIP_TRY(hipStreamSynchronize(s.stream));
    const int64_t n = s.n;
    const uint8_t *p = s.h_out;
    if (hit && n) par_memcpy(hit, p, n * 4);
    p += n * 4;
    if (ed && n) memcpy(ed, p, n);
    p += n;
    if (end_idx && n) memcpy(end_idx, p, n);
    p += n;
    if (flags && n) memcpy(flags, p, n);
    return BZ_OK;
}

int bz_assign_submit(bz_handle h, int slot, const uint8_t *ext, int ext_stride, const uint8_t *ext_len,
                     const uint8_t *bc_off, const uint8_t *bc_len, int64_t n, int max_ed) {
    if (int rc = bz_assign_stage(h, slot, ext, ext_stride, ext_len, bc_off, bc_len, n)) return rc;
    if (int rc = bz_assign_run(h, slot, max_ed)) return rc;
    return start_fetch(h->slot[slot]);
}

int bz_assign_wait(bz_handle h, int slot, int32_t *hit, int8_t *ed, int8_t *end_idx, uint8_t *flags) {
    if (int rc = check_slot(h, slot)) return rc;
    Slot &s = h->slot[slot];
    if (!s.ran) return fail(BZ_E_STATE, "slot %d has nothing submitted", slot);
    if (s.direct_out) {                              // bz_assign_submit_pinned: the results are in the caller's arrays already
        HIP_TRY(hipStreamSynchronize(s.stream));
        return BZ_OK;
    }
    return finish_fetch(s, hit, ed, end_idx, flags);
}

// ---- pinned host memory for callers, and the submit that DMAs straight from / into it ---------------------------
int bz_host_alloc(void **out, int64_t bytes) {
    if (!out || bytes < 0) return fail(BZ_E_ARG, "bad host allocation arguments");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, (size_t)std::max<int64_t>(bytes, 1), hipHostMallocPortable));
    return BZ_OK;
}

int bz_host_free(void *p) {
    if (p) HIP_TRY(hipHostFree(p));
    return BZ_OK;
}

int bz_assign_submit_pinned(bz_handle h, int slot, const uint8_t *ext, int ext_stride, const uint8_t *ext_len, const uint8_t *bc_off,
                            const uint8_t *bc_len, int64_t n, int max_ed, int32_t *hit, int8_t *ed, int8_t *end_idx, uint8_t *flags) {
    if (int rc = check_slot(h, slot)) return rc;
    if (h->n_wl < 0) return fail(BZ_E_STATE, "bz_set_whitelist has not been called");
    if (n < 0 || n > (int64_t)1 << 30) return fail(BZ_E_LIMIT, "row count %lld out of range", (long long)n);
    if (n > 0 && (!ext || !ext_len || !bc_off || !bc_len || !hit || !ed || !end_idx || !flags)) return fail(BZ_E_ARG, "null row / result arrays");
    if (ext_stride < 1 || ext_stride > BZ_EXT_MAX) return fail(BZ_E_LIMIT, "ext_stride %d outside 1..%d", ext_stride, BZ_EXT_MAX);
    if (n > 0) {
        const void *all[8] = {ext, ext_len, bc_off, bc_len, hit, ed, end_idx, flags};
        for (const void *p : all)
            if (!is_pinned_host(p)) return fail(BZ_E_ARG, "bz_assign_submit_pinned needs every array in pinned host memory (bz_host_alloc / hipHostRegister); use bz_assign_submit for pageable arrays");
    }
    if (int rc = validate_rows(ext_len, bc_off, bc_len, n, ext_stride)) return rc;
    Slot &s = h->slot[slot];
    if (n > s.cap || ext_stride > s.cap_stride) { if (int rc = flush_pending(h, s, slot)) return rc; }   // the buffers are about to be replaced
    HIP_TRY(hipStreamSynchronize(s.stream));
    if (int rc = ensure_slot(s, n, ext_stride)) return rc;
    s.n = n; s.stride = ext_stride; s.staged = false; s.ran = false; s.direct_out = false;
    if (n) {
        HIP_TRY(hipMemcpyAsync(s.d_ext, ext, (size_t)n * ext_stride, hipMemcpyHostToDevice, s.stream));
        HIP_TRY(hipMemcpyAsync(s.d_len, ext_len, n, hipMemcpyHostToDevice, s.stream));
        HIP_TRY(hipMemcpyAsync(s.d_off, bc_off, n, hipMemcpyHostToDevice, s.stream));
        HIP_TRY(hipMemcpyAsync(s.d_bcl, bc_len, n, hipMemcpyHostToDevice, s.stream));
    }
    s.staged = true;
    if (int rc = bz_assign_run(h, slot, max_ed)) return rc;
    if (n) {
        HIP_TRY(hipMemcpyAsync(hit, s.d_hit, n * 4, hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(hipMemcpyAsync(ed, s.d_ed, n, hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(hipMemcpyAsync(end_idx, s.d_end, n, hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(hipMemcpyAsync(flags, s.d_flags, n, hipMemcpyDeviceToHost, s.stream));
    }
    s.direct_out = true;
    return BZ_OK;
}

int bz_assign_fetch(bz_handle h, int slot, int32_t *hit, int8_t *ed, int8_t *end_idx, uint8_t *flags) {
    if (int rc = check_slot(h, slot)) return rc;
    Slot &s = h->slot[slot];
    if (!s.ran) return fail(BZ_E_STATE, "slot %d has not run", slot);
    if (int rc = start_fetch(s)) return rc;
    return finish_fetch(s, hit, ed, end_idx, flags);
}

int bz_slot_sync(bz_handle h, int slot) {
    if (int rc = check_slot(h, slot)) return rc;
    if (int rc = flush_pending(h, h->slot[slot], slot)) return rc;     // "everything submitted to the slot is done" includes its counts
    HIP_TRY(hipStreamSynchronize(h->slot[slot].stream));
    return BZ_OK;
}

int bz_slot_timing(bz_handle h, int slot, float ms[8], int64_t counts[8]) {
    if (int rc = check_slot(h, slot)) return rc;
    Slot &s = h->slot[slot];
    if (!s.ran) return fail(BZ_E_STATE, "slot %d has not run", slot);
    if (int rc = flush_pending(h, s, slot)) return rc;                 // the list counters are published by the histogram's last kernel
    HIP_TRY(hipStreamSynchronize(s.stream));
    if (ms) {
        memset(ms, 0, 8 * sizeof(float));
        if (s.timed) {                                  // BZ_OPT_PROFILE was on for that run
            HIP_TRY(hipEventElapsedTime(&ms[0], s.ev[0], s.ev[4]));
            for (int i = 1; i <= 4; ++i) HIP_TRY(hipEventElapsedTime(&ms[i], s.ev[i - 1], s.ev[i]));
            if (s.timed_pair) {
                HIP_TRY(hipEventElapsedTime(&ms[5], s.ev[1], s.ev[5]));
                HIP_TRY(hipEventElapsedTime(&ms[6], s.ev[5], s.ev[6]));
            }
        }
    }
    if (counts) {
        memset(counts, 0, 8 * sizeof(int64_t));
        counts[0] = s.n; counts[1] = s.h_counters[0]; counts[2] = s.h_counters[1]; counts[3] = h->n_wl;
        counts[4] = s.h_counters[2]; counts[5] = s.h_counters[3];
    }
    return BZ_OK;
}

static int pair_call(bz_handle h, const char *a, int la, const char *b, int lb, int max_ed, int global_mode, int *end_idx) {
    if (int rc = use(h)) return rc;
    if (la < 0 || lb < 0 || (la && !a) || (lb && !b)) return fail(BZ_E_ARG, "bad operands");
    if (std::min(la, lb) > BZ_EXT_MAX) return fail(BZ_E_LIMIT, "shorter operand longer than %d", BZ_EXT_MAX);
    if (la + lb > 4096) return fail(BZ_E_LIMIT, "operands longer than 4096 bytes in total");
    if (max_ed < 0) return fail(BZ_E_ARG, "negative max_ed");
    hipStream_t st = h->slot[0].stream;
    if (la) HIP_TRY(hipMemcpyAsync(h->d_pair, a, la, hipMemcpyHostToDevice, st));
    if (lb) HIP_TRY(hipMemcpyAsync(h->d_pair + la, b, lb, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(bz::k_pair, dim3(1), dim3(64), 0, st, h->d_pair, la, h->d_pair + la, lb, max_ed, h->tie_rule, global_mode, h->d_pair_out);
    HIP_TRY(hipGetLastError());
    int32_t out[2];
    HIP_TRY(hipMemcpyAsync(out, h->d_pair_out, sizeof out, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (end_idx) *end_idx = out[1];
    return out[0];
}

int bz_sub_edit_distance(bz_handle h, const char *a, int la, const char *b, int lb, int max_ed, int *end_idx) {
    return pair_call(h, a, la, b, lb, max_ed, 0, end_idx);
}
int bz_edit_distance(bz_handle h, const char *a, int la, const char *b, int lb, int max_ed) {
    return pair_call(h, a, la, b, lb, max_ed, 1, nullptr);
}

// ---- stage 1: putative barcode search per read (polyT_adaptor_finder.py class Read) -----------------------------
int bz_putative_bc(bz_handle h, const char *seq, const char *qual, const int64_t *off, int64_t n, int kit_5prime, int umi_len,
                   int32_t *pass_code, int8_t *strand, int32_t *raw_bc_start, int32_t *min_q, int32_t *polyT_end) {
    if (int rc = use(h)) return rc;
    if (n < 0 || !off || (n > 0 && (!seq || !qual || !pass_code || !strand || !raw_bc_start || !min_q || !polyT_end)))
        return fail(BZ_E_ARG, "bad putative-bc arguments");
    if (umi_len < 0 || umi_len > 64) return fail(BZ_E_ARG, "umi_len %d outside 0..64", umi_len);
    if (n > (int64_t)1 << 30) return fail(BZ_E_LIMIT, "more than 2^30 reads");
    for (int64_t k = 0; k < n; ++k) {
        if (off[k + 1] < off[k]) return fail(BZ_E_ARG, "offsets not monotone at %lld", (long long)k);
        if (off[k + 1] - off[k] > (int64_t)1 << 30) return fail(BZ_E_LIMIT, "read %lld longer than 2^30", (long long)k);
    }
    if (n == 0) return BZ_OK;
    const int64_t total = off[n] - off[0];
    S1Buf &b = h->s1[0];
    hipStream_t st = h->slot[0].stream;
    if (int rc = s1_ensure(b, st, total, n)) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    for (int64_t k = 0; k <= n; ++k) b.h_off[k] = off[k] - off[0];
    if (total) { memcpy(b.h_seq, seq + off[0], (size_t)total); memcpy(b.h_qual, qual + off[0], (size_t)total); }
    if (int rc = s1_launch(h, 0, n, kit_5prime, umi_len)) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    memcpy(pass_code, b.h_i32, (size_t)n * sizeof(int32_t));
    memcpy(raw_bc_start, b.h_i32 + n, (size_t)n * sizeof(int32_t));
    memcpy(min_q, b.h_i32 + 2 * n, (size_t)n * sizeof(int32_t));
    memcpy(polyT_end, b.h_i32 + 3 * n, (size_t)n * sizeof(int32_t));
    memcpy(strand, b.h_strand, (size_t)n);
    return BZ_OK;
}

// ---- empty-drop test: which strings of A are far from every string of B (main.py:244-256, :269-281) -------------
int bz_far_from_all(bz_handle h, const char *a_bytes, const int64_t *a_off, int64_t n_a, const char *b_bytes, const int64_t *b_off,
                    int64_t n_b, int min_ed, uint8_t *far_out) {
    if (int rc = use(h)) return rc;
    if (n_a < 0 || n_b < 0 || !a_off || !b_off || (n_a > 0 && (!a_bytes || !far_out)) || (n_b > 0 && !b_bytes))
        return fail(BZ_E_ARG, "bad far-from-all arguments");
    if (min_ed < 1 || min_ed > BZ_EXT_MAX) return fail(BZ_E_ARG, "min_ed %d outside 1..%d", min_ed, BZ_EXT_MAX);
    if (n_a > (int64_t)1 << 30 || n_b > (int64_t)1 << 30) return fail(BZ_E_LIMIT, "more than 2^30 strings");
    for (int64_t k = 0; k < n_a; ++k)
        if (a_off[k + 1] < a_off[k] || a_off[k + 1] - a_off[k] > BZ_EXT_MAX) return fail(BZ_E_LIMIT, "string %lld of A: length outside 0..%d", (long long)k, BZ_EXT_MAX);
    for (int64_t k = 0; k < n_b; ++k)
        if (b_off[k + 1] < b_off[k] || b_off[k + 1] - b_off[k] > BZ_EXT_MAX) return fail(BZ_E_LIMIT, "string %lld of B: length outside 0..%d", (long long)k, BZ_EXT_MAX);
    if (n_a == 0) return BZ_OK;
    // the real case -- 16-nt ACGT strings on both sides -- takes the bit-parallel kernel
    auto regular16 = [](const char *bytes, const int64_t *off, int64_t n) {
        for (int64_t k = 0; k < n; ++k) {
            if (off[k + 1] - off[k] != bz::kBcLen) return false;
            for (int i = 0; i < bz::kBcLen; ++i) { const char c = bytes[off[k] + i]; if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return false; }
        }
        return true;
    };
    if (n_b > 0 && regular16(a_bytes, a_off, n_a) && regular16(b_bytes, b_off, n_b)) {
        std::vector<uint32_t> ac(n_a), bp(n_b);
        for (int64_t k = 0; k < n_a; ++k) {
            uint32_t c = 0;
            for (int i = 0; i < bz::kBcLen; ++i) c |= (((uint32_t)(unsigned char)a_bytes[a_off[k] + i] >> 1) & 3u) << (2 * i);
            ac[k] = c;
        }
        for (int64_t k = 0; k < n_b; ++k) {
            uint32_t hi = 0, lo = 0;
            for (int i = 0; i < bz::kBcLen; ++i) {
                const uint32_t c = ((uint32_t)(unsigned char)b_bytes[b_off[k] + i] >> 1) & 3u;
                hi |= ((c >> 1) & 1u) << i; lo |= (c & 1u) << i;
            }
            bp[k] = hi | (lo << 16);
        }
        uint32_t *d_ac = nullptr, *d_bp = nullptr;
        uint8_t *d_f = nullptr;
        int rc2 = BZ_OK;
        auto step2 = [&](hipError_t e) { if (e != hipSuccess && rc2 == BZ_OK) rc2 = fail(BZ_E_HIP, "%s", hipGetErrorString(e)); };
        step2(hipMalloc(&d_ac, n_a * sizeof(uint32_t))); step2(hipMalloc(&d_bp, n_b * sizeof(uint32_t))); step2(hipMalloc(&d_f, n_a));
        hipStream_t st = h->slot[0].stream;
        if (rc2 == BZ_OK) {
            step2(hipMemcpyAsync(d_ac, ac.data(), n_a * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            step2(hipMemcpyAsync(d_bp, bp.data(), n_b * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            step2(hipMemsetAsync(d_f, 1, n_a, st));
        }
        if (rc2 == BZ_OK) {
            hipLaunchKernelGGL(bz::k_far_from_all_16, dim3((unsigned)((n_a + 255) / 256), (unsigned)((n_b + bz::kFarChunk - 1) / bz::kFarChunk)), dim3(256), 0, st,
                               d_ac, (int)n_a, d_bp, (int)n_b, min_ed, d_f);
            step2(hipGetLastError());
            step2(hipMemcpyAsync(far_out, d_f, n_a, hipMemcpyDeviceToHost, st));
            step2(hipStreamSynchronize(st));
        }
        hipFree(d_ac); hipFree(d_bp); hipFree(d_f);
        return rc2;
    }
    std::vector<int64_t> ao(n_a + 1), bo(n_b + 1);
    for (int64_t k = 0; k <= n_a; ++k) ao[k] = a_off[k] - a_off[0];
    for (int64_t k = 0; k <= n_b; ++k) bo[k] = b_off[k] - b_off[0];
    uint8_t *d_a = nullptr, *d_b = nullptr, *d_far = nullptr;
    int64_t *d_ao = nullptr, *d_bo = nullptr;
    int rc = BZ_OK;
    auto step = [&](hipError_t e) { if (e != hipSuccess && rc == BZ_OK) rc = fail(BZ_E_HIP, "%s", hipGetErrorString(e)); };
    step(hipMalloc(&d_a, std::max<int64_t>(ao[n_a], 1))); step(hipMalloc(&d_b, std::max<int64_t>(bo[n_b], 1)));
    step(hipMalloc(&d_ao, (n_a + 1) * sizeof(int64_t))); step(hipMalloc(&d_bo, (n_b + 1) * sizeof(int64_t)));
    step(hipMalloc(&d_far, n_a));
    if (rc == BZ_OK) {
        if (ao[n_a]) step(hipMemcpy(d_a, a_bytes + a_off[0], ao[n_a], hipMemcpyHostToDevice));
        if (bo[n_b]) step(hipMemcpy(d_b, b_bytes + b_off[0], bo[n_b], hipMemcpyHostToDevice));
        step(hipMemcpy(d_ao, ao.data(), (n_a + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
        step(hipMemcpy(d_bo, bo.data(), (n_b + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    if (rc == BZ_OK) {
        hipLaunchKernelGGL(bz::k_far_from_all, dim3((unsigned)((n_a + 255) / 256)), dim3(256), 0, h->slot[0].stream, d_a, d_ao, (int)n_a, d_b, d_bo,
                           (int)n_b, min_ed, d_far);
        step(hipGetLastError());
        step(hipStreamSynchronize(h->slot[0].stream));
        step(hipMemcpy(far_out, d_far, n_a, hipMemcpyDeviceToHost));
    }
    hipFree(d_a); hipFree(d_b); hipFree(d_ao); hipFree(d_bo); hipFree(d_far);
    return rc;
}

int bz_counts_reset(bz_handle h) {
    if (int rc = use(h)) return rc;
    if (h->n_wl < 0) return fail(BZ_E_STATE, "no whitelist");
    if (int rc = flush_all(h)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(h->d_counts, 0, counts_bytes(h->n_wl)));
    return BZ_OK;
}

int bz_counts_get(bz_handle h, int64_t *out, int64_t n_out) {
    if (int rc = use(h)) return rc;
    if (h->n_wl < 0) return fail(BZ_E_STATE, "no whitelist");
    if (!out || n_out != h->n_wl + 2) return fail(BZ_E_ARG, "n_out must be n_whitelist + 2");
    if (int rc = flush_all(h)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (int rc = fold_totals(h, h->slot[0].stream)) return rc;
    HIP_TRY(hipStreamSynchronize(h->slot[0].stream));
    HIP_TRY(hipMemcpy(out, h->d_counts, n_out * sizeof(int64_t), hipMemcpyDeviceToHost));
    return BZ_OK;
}

int bz_comm_unique_id(uint8_t id[128]) {
    if (!id) return fail(BZ_E_ARG, "null id");
    if (int rc = rccl_load()) return rc;
    int r = g_rccl.GetUniqueId(id);
    if (r) return fail(BZ_E_RCCL, "ncclGetUniqueId: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return BZ_OK;
}

int bz_comm_init(bz_handle h, const uint8_t id[128], int n_ranks, int rank) {
    if (int rc = use(h)) return rc;
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(BZ_E_ARG, "bad communicator arguments");
    if (int rc = rccl_load()) return rc;
    if (h->comm) { g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
    RcclId uid;
    memcpy(uid.b, id, 128);
    int r = g_rccl.CommInitRank(&h->comm, n_ranks, uid, rank);
    if (r) { h->comm = nullptr; return fail(BZ_E_RCCL, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"); }
    return BZ_OK;
}

int bz_comm_allreduce_counts(bz_handle h) {
    if (int rc = use(h)) return rc;
    if (!h->comm) return fail(BZ_E_STATE, "no communicator");
    if (h->n_wl < 0) return fail(BZ_E_STATE, "no whitelist");
    if (int rc = flush_all(h)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    hipStream_t st = h->slot[0].stream;
    if (int rc = fold_totals(h, st)) return rc;
    int r = g_rccl.AllReduce(h->d_counts, h->d_counts, (size_t)h->n_wl + 2, kNcclInt64, kNcclSum, h->comm, st);
    if (r) return fail(BZ_E_RCCL, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    HIP_TRY(hipStreamSynchronize(st));
    return BZ_OK;
}

int bz_comm_allreduce_f64(bz_handle h, double *inout, int n, int op) {
    if (int rc = use(h)) return rc;
    if (!h->comm) return fail(BZ_E_STATE, "no communicator");
    if (!inout || n < 1 || n > 64) return fail(BZ_E_ARG, "n must be 1..64");
    hipStream_t st = h->slot[0].stream;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyAsync(h->d_f64, inout, n * sizeof(double), hipMemcpyHostToDevice, st));
    int r = g_rccl.AllReduce(h->d_f64, h->d_f64, (size_t)n, kNcclFloat64, op == 1 ? kNcclMax : kNcclSum, h->comm, st);
    if (r) return fail(BZ_E_RCCL, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    HIP_TRY(hipMemcpyAsync(inout, h->d_f64, n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return BZ_OK;
}

int bz_comm_allreduce_i64(bz_handle h, int64_t *inout, int64_t n) {
    if (int rc = use(h)) return rc;
    if (!h->comm) return fail(BZ_E_STATE, "no communicator");
    if (!inout || n < 1 || n > (int64_t)1 << 28) return fail(BZ_E_ARG, "n must be 1..2^28");
    hipStream_t st = h->slot[0].stream;
    HIP_TRY(hipDeviceSynchronize());
    if (n > h->i64_cap) {
        hipFree(h->d_i64);
        h->d_i64 = nullptr; h->i64_cap = 0;
        const int64_t cap = std::max<int64_t>(n, 1024);
        HIP_TRY(hipMalloc(&h->d_i64, (size_t)cap * sizeof(int64_t)));
        h->i64_cap = cap;
    }
    HIP_TRY(hipMemcpyAsync(h->d_i64, inout, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, st));
    int r = g_rccl.AllReduce(h->d_i64, h->d_i64, (size_t)n, kNcclInt64, kNcclSum, h->comm, st);
    if (r) return fail(BZ_E_RCCL, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    HIP_TRY(hipMemcpyAsync(inout, h->d_i64, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return BZ_OK;
}

int bz_comm_destroy(bz_handle h) {
    if (!h) return fail(BZ_E_ARG, "null handle");
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    h->comm = nullptr;
    return BZ_OK;
}

// ---- whole step 3 over files (native host pipeline, host_io.h) ------------------------------------------------
int bz_fastq_count_reads(const char *path, int64_t *n_reads) {
    if (!path || !n_reads) return fail(BZ_E_ARG, "bad arguments");
    try {
        *n_reads = bzio::count_fastq_reads(path);
    } catch (const std::exception &e) {
        return fail(BZ_E_ARG, "%s", e.what());
    }
    return BZ_OK;
}

int bz_assign_files(bz_handle *engines, int n_engines, const char *const *fastq_fns, int n_fastq, const char *putative_bc_csv,
                    const char *fastq_out, int gz, int restrand, int max_ed, int batch_size, int64_t device_batch_rows,
                    int n_threads, int64_t out_counts[4]) {
    return bz_assign_files_from(engines, n_engines, fastq_fns, n_fastq, putative_bc_csv, 0, fastq_out, gz, restrand, max_ed, batch_size,
                                device_batch_rows, n_threads, out_counts);
}

static int assign_files_impl(bz_handle *engines, int n_engines, const char *const *fastq_fns, int n_fastq, const char *putative_bc_csv,
                             int64_t csv_skip_rows, int64_t csv_byte_offset, const char *fastq_out, int gz, int restrand, int max_ed,
                             int batch_size, int64_t device_batch_rows, int n_threads, int64_t out_counts[4]);

int bz_assign_files_from(bz_handle *engines, int n_engines, const char *const *fastq_fns, int n_fastq, const char *putative_bc_csv,
                         int64_t csv_skip_rows, const char *fastq_out, int gz, int restrand, int max_ed, int batch_size,
                         int64_t device_batch_rows, int n_threads, int64_t out_counts[4]) {
    return assign_files_impl(engines, n_engines, fastq_fns, n_fastq, putative_bc_csv, csv_skip_rows, -1, fastq_out, gz, restrand, max_ed,
                             batch_size, device_batch_rows, n_threads, out_counts);
}

int bz_assign_files_at(bz_handle *engines, int n_engines, const char *const *fastq_fns, int n_fastq, const char *putative_bc_csv,
                       int64_t csv_byte_offset, const char *fastq_out, int gz, int restrand, int max_ed, int batch_size,
                       int64_t device_batch_rows, int n_threads, int64_t out_counts[4]) {
    if (csv_byte_offset < 0) return fail(BZ_E_ARG, "negative csv_byte_offset");
    return assign_files_impl(engines, n_engines, fastq_fns, n_fastq, putative_bc_csv, 0, csv_byte_offset, fastq_out, gz, restrand, max_ed,
                             batch_size, device_batch_rows, n_threads, out_counts);
}

int bz_fastq_first_read_id(const char *path, char *out, int cap, int *len) {
    if (!path || !out || cap < 1 || !len) return fail(BZ_E_ARG, "bad arguments");
    try {
        const std::string id = bzio::first_fastq_id(path);
        if ((int)id.size() >= cap) return fail(BZ_E_LIMIT, "read id of %zu bytes does not fit %d", id.size(), cap);
        memcpy(out, id.data(), id.size());
        out[id.size()] = 0;
        *len = (int)id.size();
    } catch (const std::exception &e) {
        return fail(BZ_E_ARG, "%s", e.what());
    }
    return BZ_OK;
}

int bz_csv_locate_read(const char *putative_bc_csv, const char *read_id, int id_len, double hint_fraction, int64_t *byte_offset) {
    if (!putative_bc_csv || !read_id || id_len < 0 || !byte_offset) return fail(BZ_E_ARG, "bad arguments");
    *byte_offset = bzio::csv_locate_row(putative_bc_csv, std::string(read_id, (size_t)id_len), hint_fraction);
    return BZ_OK;
}

static int assign_files_impl(bz_handle *engines, int n_engines, const char *const *fastq_fns, int n_fastq, const char *putative_bc_csv,
                             int64_t csv_skip_rows, int64_t csv_byte_offset, const char *fastq_out, int gz, int restrand, int max_ed,
                             int batch_size, int64_t device_batch_rows, int n_threads, int64_t out_counts[4]) {
    if (csv_skip_rows < 0) return fail(BZ_E_ARG, "negative csv_skip_rows");
    if (!engines || n_engines < 1 || !fastq_fns || n_fastq < 0 || !putative_bc_csv || !fastq_out) return fail(BZ_E_ARG, "bad file arguments");
    if (batch_size < 1 || device_batch_rows < 1 || n_threads < 1) return fail(BZ_E_ARG, "batch_size, device_batch_rows and n_threads must be >= 1");
    for (int e = 0; e < n_engines; ++e) {
        if (!engines[e]) return fail(BZ_E_ARG, "null engine");
        if (engines[e]->n_wl < 0) return fail(BZ_E_STATE, "bz_set_whitelist has not been called on engine %d", e);
    }
    bzio::Job job;
    for (int k = 0; k < n_fastq; ++k) job.fastq_fns.emplace_back(fastq_fns[k]);
    job.csv = putative_bc_csv; job.out_path = fastq_out;
    job.gz = gz != 0; job.restrand = restrand != 0; job.max_ed = max_ed; job.batch_size = batch_size;
    job.n_threads = n_threads; job.device_batch_rows = std::min<int64_t>(device_batch_rows, (int64_t)1 << 24);
    {   // a chunk's reference batches are the tasks of the host threads: a chunk of 66 batches on 16 threads takes five
        // rounds of which the last is one task; cut chunks at a multiple of (threads x batch) rows where that fits
        const int64_t round = (int64_t)n_threads * batch_size;
        if (round <= job.device_batch_rows) job.device_batch_rows = job.device_batch_rows / round * round;
    }
    job.csv_skip_rows = csv_skip_rows;
    job.csv_byte_offset = csv_byte_offset;
    // How the `.gz` members are compressed.  The reference's gzip.compress is zlib level 9; the decompressed stream is the
    // contract (the compressed bytes carry a wall-clock mtime and are never reproducible).  Default 0 = the
    // FASTQ-aware block encoder of fastq_deflate.h (1.8 % larger than level 9, ~80x faster per host thread); 1..9 = zlib
    // at that level (BZ_OPT_GZ_LEVEL / BLAZE_AMD_GZ_LEVEL=9 give the reference's setting back).
    job.gz_level = engines[0]->gz_level;
    if (const char *lv = getenv("BLAZE_AMD_GZ_LEVEL")) {
        const int v = atoi(lv);
        if (v >= 0 && v <= 9 && (v > 0 || lv[0] == '0')) job.gz_level = v;
    }

    int64_t demul_tot = 0, reads_tot = 0;
    int rc = BZ_OK;
    bzio::StageClock clk;                            // 0 reader, 1 wait for reader, 2 parse, 3 submit, 4 wait device, 5 format, 6 hand to writer, 7 write
    const double t_job = bzio::StageClock::now();
    bzio::Pool pool(n_threads);                      // the job's host threads (parse, format): alive until the job ends
    bzio::PoolScope pool_scope(&pool);
    const bz_handle e0 = engines[0];
    bzio::OrderedPipeline<bzio::Chunk> pipe;
    pipe.out_path = fastq_out;
    pipe.create_output_first = true;                 // the reference opens its output before the first batch (read_assignment.py:279-283)
    pipe.max_inflight = 2 * n_engines;
    pipe.clk = &clk;
    pipe.open_reader = [&]() {
        auto rd = std::make_shared<bzio::ChunkReader>(job);
        return std::function<std::unique_ptr<bzio::Chunk>()>([rd]() { return rd->next(); });
    };
    pipe.submit = [&](bzio::Chunk &c, int64_t k) {
        double t1 = bzio::StageClock::now();
        const size_t n = c.reads.size();
        c.bc.resize(n); c.umi.resize(n); c.post.resize(n); c.rid.resize(n); c.polyT.resize(n);
        c.ext.resize(n * bzio::kExtStride); c.ext_len.resize(n); c.bc_off.resize(n); c.bc_len.resize(n);
        bzio::parallel_for((int)c.batch_start.size() - 1, job.n_threads, [&](int b) { bzio::parse_batch(c, b); });
        const double t2 = bzio::StageClock::now();
        clk.t[2] += t2 - t1;
        c.engine = (int)(k % n_engines); c.slot = (int)((k / n_engines) % 2);
        const bool wide = !c.wide.empty();
        if (wide) bzio::widen_rows(c);
        if (bz_assign_submit(engines[c.engine], c.slot, wide ? c.ext_wide.data() : c.ext.data(), wide ? bzio::kExtWide : bzio::kExtStride,
                             c.ext_len.data(), c.bc_off.data(), c.bc_len.data(), (int64_t)n, max_ed) != BZ_OK)
            throw std::runtime_error(g_err);
        clk.t[3] += bzio::StageClock::now() - t2;
    };
    pipe.finish = [&](bzio::Chunk &c, std::vector<std::string> &parts) {
        const size_t n = c.reads.size();
        c.hit.resize(n); c.ed.resize(n); c.end_idx.resize(n); c.flags.resize(n);
        const double t0 = bzio::StageClock::now();
        if (bz_assign_wait(engines[c.engine], c.slot, c.hit.data(), c.ed.data(), c.end_idx.data(), c.flags.data()) != BZ_OK)
            throw std::runtime_error(g_err);
        const double t1 = bzio::StageClock::now();
        clk.t[4] += t1 - t0;
        const int nb = (int)c.batch_start.size() - 1;
        c.out.resize(nb); c.demul.assign(nb, 0);
        bzio::parallel_for(nb, job.n_threads, [&](int b) { bzio::format_batch(c, b, job, e0->wl_host, e0->wl_off_host); });
        clk.t[5] += bzio::StageClock::now() - t1;
        for (int b = 0; b < nb; ++b) demul_tot += c.demul[b];
        reads_tot += (int64_t)n;
        parts = std::move(c.out);
    };
    pipe.abandon = [&](bzio::Chunk &c) { hipStreamSynchronize(engines[c.engine]->slot[c.slot].stream); };
    try {
        pipe.run();
    } catch (const bzio::MismatchError &e) {
        rc = fail(BZ_E_MISMATCH, "%s", e.what());
    } catch (const std::exception &e) {
        rc = fail(BZ_E_ARG, "%s", e.what());
    }
    const int64_t n_chunks = pipe.n_chunks;
    if (getenv("BLAZE_AMD_IO_TIMING"))
        fprintf(stderr, "[bz_assign_files] %.3f s total | reader thread %.3f | main: wait reader %.3f parse %.3f submit %.3f wait device %.3f "
                        "format %.3f hand-over %.3f | writer thread %.3f | %lld reads, %lld chunks, %d threads\n",
                bzio::StageClock::now() - t_job, clk.t[0], clk.t[1], clk.t[2], clk.t[3], clk.t[4], clk.t[5], clk.t[6], clk.t[7],
                (long long)reads_tot, (long long)n_chunks, n_threads);
    if (out_counts) { out_counts[0] = demul_tot; out_counts[1] = reads_tot; out_counts[2] = n_chunks; out_counts[3] = 0; }
    return rc;
}

// ---- stage 1 over files (native host pipeline, host_io.h) ------------------------------------------------------
int bz_putative_bc_files(bz_handle *engines, int n_engines, const char *const *fastq_fns, int n_fastq, const char *out_csv,
                         int kit_5prime, int umi_len, int min_q, int batch_size, int64_t device_batch_reads, int n_threads,
                         int64_t out_counts[8]) {
    if (!engines || n_engines < 1 || !fastq_fns || n_fastq < 0 || !out_csv) return fail(BZ_E_ARG, "bad file arguments");
    if (batch_size < 1 || device_batch_reads < 1 || n_threads < 1) return fail(BZ_E_ARG, "batch_size, device_batch_reads and n_threads must be >= 1");
    if (umi_len < 0 || umi_len > 64) return fail(BZ_E_ARG, "umi_len %d outside 0..64", umi_len);
    for (int e = 0; e < n_engines; ++e)
        if (!engines[e]) return fail(BZ_E_ARG, "null engine");
    bzio::S1Job job;
    for (int k = 0; k < n_fastq; ++k) job.fastq_fns.emplace_back(fastq_fns[k]);
    job.out_path = out_csv; job.kit5 = kit_5prime != 0; job.umi_len = umi_len; job.min_q = min_q; job.batch_size = batch_size;
    job.n_threads = n_threads; job.device_batch_reads = std::min<int64_t>(device_batch_reads, (int64_t)1 << 22);
    {   // whole rounds of (threads x batch) reads per chunk, as in bz_assign_files
        const int64_t round = (int64_t)n_threads * batch_size;
        if (round <= job.device_batch_reads) job.device_batch_reads = job.device_batch_reads / round * round;
    }

    int64_t totals[5] = {0, 0, 0, 0, 0}, reads_tot = 0;
    int rc = BZ_OK;
    bzio::StageClock clk;                            // 0 reader, 1 wait for reader, 2 prepare, 3 pack + submit, 4 wait device, 5 format, 6 hand to writer, 7 write
    const double t_job = bzio::StageClock::now();
    bzio::Pool pool(n_threads);                      // the job's host threads (prepare, pack, format): alive until the job ends
    bzio::PoolScope pool_scope(&pool);
    bzio::OrderedPipeline<bzio::S1Chunk> pipe;
    pipe.out_path = out_csv;
    pipe.create_output_first = false;                // the reference creates the table with its first batch (main.py:361-362): no reads, no table
    pipe.max_inflight = BZ_SLOTS * n_engines;
    pipe.clk = &clk;
    pipe.open_reader = [&]() {
        auto rd = std::make_shared<bzio::FastqChunkReader>(job);
        return std::function<std::unique_ptr<bzio::S1Chunk>()>([rd]() { return rd->next(); });
    };
    pipe.submit = [&](bzio::S1Chunk &c, int64_t k) {
        double t1 = bzio::StageClock::now();
        c.first = k == 0;
        c.engine = (int)(k % n_engines); c.slot = (int)((k / n_engines) % BZ_SLOTS);
        bzio::s1_prepare(c, job.n_threads);
        const double t2 = bzio::StageClock::now();
        clk.t[2] += t2 - t1;
        bz_handle e = engines[c.engine];
        if (use(e) != BZ_OK) throw std::runtime_error(g_err);
        const int64_t n = (int64_t)c.reads.size();
        S1Buf &b = e->s1[c.slot];
        if (s1_ensure(b, e->slot[c.slot].stream, c.off[n], n) != BZ_OK) throw std::runtime_error(g_err);
        bzio::s1_copy(c, b.h_seq, b.h_qual, b.h_off, job.n_threads);
        if (s1_launch(e, c.slot, n, job.kit5 ? 1 : 0, job.umi_len) != BZ_OK) throw std::runtime_error(g_err);
        clk.t[3] += bzio::StageClock::now() - t2;
    };
    pipe.finish = [&](bzio::S1Chunk &c, std::vector<std::string> &parts) {
        bz_handle e = engines[c.engine];
        S1Buf &b = e->s1[c.slot];
        const size_t n = c.reads.size();
        const double t0 = bzio::StageClock::now();
        if (hipSetDevice(e->device) != hipSuccess || hipStreamSynchronize(e->slot[c.slot].stream) != hipSuccess)
            throw std::runtime_error("device search failed");
        c.code.assign(b.h_i32, b.h_i32 + n); c.start.assign(b.h_i32 + n, b.h_i32 + 2 * n);
        c.minq.assign(b.h_i32 + 2 * n, b.h_i32 + 3 * n); c.polyt.assign(b.h_i32 + 3 * n, b.h_i32 + 4 * n);
        c.strand.assign(b.h_strand, b.h_strand + n);                // the slot's buffers are free again from here
        const double t1 = bzio::StageClock::now();
        clk.t[4] += t1 - t0;
        const int nb = (int)c.batch_start.size() - 1;
        c.out.resize(nb); c.pass.resize(nb);
        bzio::parallel_for(nb, job.n_threads, [&](int bi) { bzio::s1_format_batch(c, bi, job); });
        clk.t[5] += bzio::StageClock::now() - t1;
        for (int bi = 0; bi < nb; ++bi)
            for (int k = 0; k < 5; ++k) totals[k] += c.pass[bi][k];
        reads_tot += (int64_t)n;
        parts = std::move(c.out);
    };
    pipe.abandon = [&](bzio::S1Chunk &c) { hipStreamSynchronize(engines[c.engine]->slot[c.slot].stream); };
    try {
        pipe.run();
    } catch (const std::exception &e) {
        rc = fail(BZ_E_ARG, "%s", e.what());
    }
    const int64_t n_chunks = pipe.n_chunks;
    if (getenv("BLAZE_AMD_IO_TIMING"))
        fprintf(stderr, "[bz_putative_bc_files] %.3f s total | reader thread %.3f | main: wait reader %.3f prepare %.3f pack+submit %.3f wait device %.3f "
                        "format %.3f hand-over %.3f | writer thread %.3f | %lld reads, %lld chunks, %d threads\n",
                bzio::StageClock::now() - t_job, clk.t[0], clk.t[1], clk.t[2], clk.t[3], clk.t[4], clk.t[5], clk.t[6], clk.t[7],
                (long long)reads_tot, (long long)n_chunks, n_threads);
    if (out_counts) {
        memset(out_counts, 0, 8 * sizeof(int64_t));
        out_counts[0] = reads_tot;
        for (int k = 0; k < 5; ++k) out_counts[1 + k] = totals[k];
        out_counts[6] = n_chunks;
    }
    return rc;
}

// One gzip member of `text` by the FASTQ-aware block encoder (host only; what bz_assign_files writes per batch at gz level 0).
int64_t bz_gzip_fastq_bound(int64_t n) { return n < 0 ? -1 : 2 * n + 1024 + (n / 32) * 520; }

int bz_gzip_fastq_member(const char *text, int64_t n, uint8_t *out, int64_t cap, int64_t *out_len) {
    if (n < 0 || (n > 0 && !text) || !out || !out_len) return fail(BZ_E_ARG, "bad gzip arguments");
    if (n >= (int64_t)1 << 32) return fail(BZ_E_LIMIT, "a gzip member holds less than 4 GiB of text");
    std::string z;
    try {
        fqz::gzip_fastq_member(text, (size_t)n, z);
    } catch (const std::exception &e) {
        return fail(BZ_E_ARG, "%s", e.what());
    }
    if ((int64_t)z.size() > cap) return fail(BZ_E_LIMIT, "output needs %lld bytes, %lld given", (long long)z.size(), (long long)cap);
    memcpy(out, z.data(), z.size());
    *out_len = (int64_t)z.size();
    return BZ_OK;
}

int bz_valu_peak_probe(bz_handle h, int iters, double *lane_ops_per_s, float *ms_out) {
    if (int rc = use(h)) return rc;
    if (iters < 1 || !lane_ops_per_s) return fail(BZ_E_ARG, "bad probe arguments");
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, h->device));
    const int blocks = p.multiProcessorCount * 8, threads = 256;
    uint32_t *d = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct Cleanup { uint32_t *&d; hipEvent_t &e0, &e1; ~Cleanup() { if (e0) hipEventDestroy(e0); if (e1) hipEventDestroy(e1); hipFree(d); } } cleanup{d, e0, e1};
    HIP_TRY(hipMalloc(&d, (size_t)blocks * threads * 4));
    hipStream_t st = h->slot[0].stream;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    hipLaunchKernelGGL(bz::k_valu_probe, dim3(blocks), dim3(threads), 0, st, d, iters / 8 + 1, 1u);   // warm-up
    HIP_TRY(hipEventRecord(e0, st));
    hipLaunchKernelGGL(bz::k_valu_probe, dim3(blocks), dim3(threads), 0, st, d, iters, 2u);
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *lane_ops_per_s = (double)blocks * threads * (double)iters * bz::kProbeOpsPerIter / (ms * 1e-3);
    if (ms_out) *ms_out = ms;
    return BZ_OK;
}

}  // extern "C"
