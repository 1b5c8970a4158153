/*
 * blaze_hip.h -- C ABI of libblaze_hip.so, the MI355X (gfx950) read-to-whitelist
 * assignment engine that replaces the native FFI on BLAZE's step-3 hot path.
 *
 * Boundary (what these entry points replace; file:line under /root/reference):
 *   fast_edit_distance.sub_edit_distance / edit_distance  -- third-party CPython
 *     extension (pyproject.toml:22) called once per (read, barcode) pair from
 *     blaze/read_assignment.py:72 (N_whitelist calls per non-exact read).
 *   The Python loop around it, blaze/read_assignment.py:61-88 (shortcut + fold),
 *     and the process pool that fans batches out, blaze/helper.py:111-205.
 * The Python host (blaze_amd/read_assignment.py) keeps the reference's entry
 * points (assign_read / _assign_read_batches / _match_bc_row) and calls this
 * ABI through ctypes.  INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions: plain pointers and sizes, caller owns every host buffer, every
 * function returns 0 on success or a negative BZ_E* code (text of the last
 * error on the calling thread: bz_last_error()).  No exceptions cross the
 * boundary.  Calls on one handle are not re-entrant; two slots per handle give
 * double buffering (submit slot 1 while slot 0 computes).  There is NO CPU
 * fallback anywhere in this library: without a usable gfx950 device every
 * compute entry point fails with BZ_E_HIP.
 */
#ifndef BLAZE_HIP_H
#define BLAZE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bz_engine *bz_handle;

#define BZ_ABI_VERSION 1
#define BZ_EXT_MAX 64   /* longest supported extension / single-pair operand on the packed path */
#define BZ_SLOTS 2

/* error codes */
#define BZ_OK 0
#define BZ_E_ARG (-1)      /* bad argument (null pointer, size, option, slot) */
#define BZ_E_HIP (-2)      /* HIP runtime error / no device */
#define BZ_E_STATE (-3)    /* call order (no whitelist, slot empty, comm missing) */
#define BZ_E_RCCL (-4)     /* RCCL missing or returned an error */
#define BZ_E_LIMIT (-5)    /* size beyond a documented limit */
#define BZ_E_MISMATCH (-6) /* putative_bc.csv and the FASTQ files disagree on read order (read_assignment.py:219-223) */

/* options for bz_set_option */
#define BZ_OPT_TIE_RULE 1      /* 0 = leftmost end index on ties (default), 1 = rightmost.
                                  The only UNPINNED policy point of sub_edit_distance. */
#define BZ_OPT_SHORTCUT 2      /* 1 (default) = exact-match shortcut of read_assignment.py:61-62;
                                  0 = score every non-blank row (benchmark "all rows scored" mode) */
#define BZ_OPT_FORCE_GENERIC 3 /* 1 = send every row through the byte-wise DP kernel (testing) */
#define BZ_OPT_SCAN_MODE 4     /* 0 (default) = pigeonhole seeds + verify for max_ed <= 3 (max_ed 2: two untouched
                                  4-nt pieces of four; max_ed <= 1: one untouched 8-nt piece of two; max_ed 3: one
                                  untouched 4-nt piece of four), every pair through the packed automaton for max_ed 4..8, byte-wise scan above;
                                  1 = brute force always: every (row, barcode) pair runs the automaton
                                  (the roofline leg of bench.py); 2 = exact 3-gram filter + verify over the
                                  whole whitelist (also finishes the rows mode 0 hands on); 3 = single-piece
                                  seeds 5|5|6 for max_ed 2 (the first seeded scan, kept for comparison).
                                  Results are identical in all modes. */

#define BZ_OPT_PAIR_FUSED 5    /* 1 = run the pair-seeded scan of mode 0 as ONE kernel that verifies straight out of the
                                  buckets (k_pair_stream, the form large whitelists take) even when the whitelist is
                                  small enough for candidate lists + verify (testing; results identical) */

#define BZ_OPT_PROFILE 6       /* 1 = record HIP events between the kernels of every run (bz_slot_timing's ms[]); off by
                                  default: each event costs ~6 us of idle GPU between two kernels */

#define BZ_OPT_GZ_LEVEL 7      /* how bz_assign_files compresses its gzip members (set on engines[0]): 0 (default) = the FASTQ-aware
                                  block encoder (blaze_amd/csrc/fastq_deflate.h: every sequence line and everything between two
                                  sequence lines as its own dynamic-Huffman block, no match search), 1..9 = zlib at that level.
                                  The reference compresses at 9 (gzip.compress, read_assignment.py:244-245): 1.8 % smaller than
                                  level 0, ~30x the CPU time.  Only the decompressed stream is comparable (mtime in the header). */

/* result flags (per row) */
#define BZ_F_EXACT 1u      /* took the exact-match shortcut */
#define BZ_F_BLANK 2u      /* empty putative_bc (`not row.putative_bc`, read_assignment.py:61) */
#define BZ_F_AMBIG 4u      /* fold ended 'ambiguous' (read_assignment.py:82-88) */
#define BZ_F_GENERIC 8u    /* row went through the generic DP kernel, not the bit-parallel one */
#define BZ_F_INVALID 16u   /* row rejected on the device: ext_len > stride or barcode outside its extension (set together
                              with BZ_F_BLANK; only rows handed over in device memory can carry it, host rows are
                              rejected with BZ_E_ARG before any launch) */

/* ---- library / device ------------------------------------------------------------------ */
int bz_abi_version(void);
const char *bz_last_error(void);
int bz_device_count(int *count);
/* info[0]=compute units, [1]=max clock kHz, [2]=wavefront size, [3]=total HBM MiB, [4]=LDS bytes per CU */
int bz_device_info(int device, int64_t info[8], char *arch, int arch_cap);
/* "<PCI bus id>/<uuid as hex>" of a visible device (cap >= 64): what bench.py's ranks exchange to count the DISTINCT GPUs of a
 * multi-process run -- a launcher that narrows every rank's view makes each rank see "device 0" (no reference counterpart:
 * the reference's pool workers share the host's cores, helper.py:137). */
int bz_device_identity(int device, char *out, int cap);

/* ---- engine ---------------------------------------------------------------------------- */
/* One engine per (process, GPU).  Replaces helper.multiprocessing_submit's worker pool
 * (helper.py:135-139): the device, two streams and their staging buffers live here.  A bz_create that fails part way
 * gives back what it had made; *out is written on success only. */
int bz_create(int device, bz_handle *out);
int bz_destroy(bz_handle h);
int bz_set_option(bz_handle h, int option, int64_t value);

/* Whitelist upload, replaces the per-task pickled whitelist + `set(whitelist)` of
 * read_assignment.py:206,314.  `bytes` holds the barcodes back to back, barcode k is
 * bytes[offsets[k] .. offsets[k+1]).  ORDER IS SEMANTIC (the fold of :66-85 is
 * order dependent) and is kept exactly as given; the caller passes list(set(whitelist)).
 * 16-nt ACGT entries are packed 2 bits/base into the bit-plane layout the kernels stream and get the seed indices;
 * any other line the reference would accept (read_assignment.py:286-289: other lengths, other letters, a blank line)
 * is kept as an irregular entry beside them -- up to 65,536 -- and scored exactly (same results, slower rows).  An entry
 * longer than a row is the TEXT of the replaced call and the end index a position inside it: an entry of more than 127
 * characters is BZ_E_LIMIT (the end index would not fit the int8 result). */
int bz_set_whitelist(bz_handle h, const char *bytes, const int64_t *offsets, int64_t n);

/* Rows.  Row r is the extension ext[r*ext_stride .. +ext_len[r]) =
 * pre_bc_flanking[-2:] + putative_bc + putative_umi[:2] (read_assignment.py:70), with
 * putative_bc at [bc_off[r], bc_off[r]+bc_len[r]).  bc_len 0 = blank row.
 * Fastest row layout: ext_stride a multiple of 8 and >= 24 (24 holds the reference's 20-nt extension; rows are then read
 * with aligned 8-byte loads and take the packed kernels); any other stride works and is read byte by byte.
 * Outputs per row: hit = whitelist index or -1 ('' in the reference: none, ambiguous or
 * blank); ed = final best_ed (0 for the shortcut); end_idx = bc_hit_end_idx of :76,80
 * (umi_adj = end_idx - (ext_len-1-2), :93); flags = BZ_F_*.
 *
 * submit = stage + run + start the device->host copy, asynchronous on the slot's stream;
 * wait   = block until that slot is finished and copy the results out (any out pointer may
 *          be NULL).  Results come back in row order (helper.py:191-205's guarantee). */
int bz_assign_submit(bz_handle h, int slot, const uint8_t *ext, int ext_stride, const uint8_t *ext_len,
                     const uint8_t *bc_off, const uint8_t *bc_len, int64_t n, int max_ed);
int bz_assign_wait(bz_handle h, int slot, int32_t *hit, int8_t *ed, int8_t *end_idx, uint8_t *flags);

/* Pinned (page-locked) host memory for callers that want the boundary at PCIe speed: bz_assign_submit copies pageable
 * arrays into the engine's pinned staging first (one extra pass over every byte on the host, shared among a few copy
 * threads of the library when the batch is large); rows and results that
 * already live in pinned memory -- bz_host_alloc, or the caller's own hipHostRegister -- are DMA'd straight from / into
 * the caller's arrays by bz_assign_submit_pinned.  It checks that all eight arrays are pinned (BZ_E_ARG otherwise); the
 * result arrays are given at submit time and are complete when bz_assign_wait(h, slot, NULL, NULL, NULL, NULL) returns.
 * Replaces the pickling of every batch to a pool worker and back (helper.py:159-205, read_assignment.py:309-327). */
int bz_host_alloc(void **out, int64_t bytes);
int bz_host_free(void *p);
int bz_assign_submit_pinned(bz_handle h, int slot, const uint8_t *ext, int ext_stride, const uint8_t *ext_len,
                            const uint8_t *bc_off, const uint8_t *bc_len, int64_t n, int max_ed, int32_t *hit, int8_t *ed,
                            int8_t *end_idx, uint8_t *flags);

/* The same three steps separately (bench.py keeps rows resident in HBM and times `run`). */
int bz_assign_stage(bz_handle h, int slot, const uint8_t *ext, int ext_stride, const uint8_t *ext_len,
                    const uint8_t *bc_off, const uint8_t *bc_len, int64_t n);
int bz_assign_run(bz_handle h, int slot, int max_ed);
int bz_assign_fetch(bz_handle h, int slot, int32_t *hit, int8_t *ed, int8_t *end_idx, uint8_t *flags);
int bz_slot_sync(bz_handle h, int slot);
/* HIP-event durations (ms) of the slot's last run (zeros unless BZ_OPT_PROFILE was on): ms[0] whole run, [1] classify+shortcut,
 * [2] bit-parallel scan, [3] generic scan, [4] histogram, [5] k_pair_candidates and [6] k_pair_verify inside [2] (zero when
 * another scan ran); counts[0] rows, [1] rows scored by the
 * bit-parallel kernels, [2] rows scored by the generic kernel, [3] whitelist size, [4] rows the seeded
 * scan handed to the filtered scan (more than 8 near matches), [5] rows of a MIXED whitelist (regular entries + at most
 * 65,536 irregular ones, a blank line's empty entry included) that came within max_ed of an irregular entry and were re-scored over the whole list. */
int bz_slot_timing(bz_handle h, int slot, float ms[8], int64_t counts[8]);

/* ---- the entry point that allocates nothing ------------------------------------------------------------------
 * NO COUNTERPART IN THE REFERENCE.  For callers that cannot call hipMalloc: code under HIP graph capture, tight
 * serving loops, harnesses that forbid allocation.  It is bz_assign_run for rows that already ARE in device memory
 * (same layout as bz_assign_stage's arguments, but device pointers), results left in device memory (d_hit .. d_flags,
 * n entries each), asynchronous on `stream` (a hipStream_t; NULL = the engine's slot-0 stream, wait with
 * bz_slot_sync(h, 0)).  The kernels' work lists and counters (20 bytes per row) live in a file-scope device array of
 * the library, 5.0 MiB per GPU for BZ_NOALLOC_MAX_ROWS rows, which each GPU holds until the process ends: the call
 * performs no hipMalloc / hipFree / hipHostMalloc and no other allocating routine.  More rows than that:
 * BZ_E_LIMIT, nothing launched.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two
 * streams, two engines, two threads) RACE; calls on one stream are safe.  bz_set_whitelist must have been called (it is
 * the set-up step and does allocate).  Results equal bz_assign_submit/wait's on the same rows (the max_ed 2 scan
 * verifies straight out of the seed buckets instead of writing candidate lists); rows whose lengths do not fit are
 * returned as BZ_F_BLANK | BZ_F_INVALID with hit -1 instead of an error (the host cannot see them).  The
 * per-barcode histogram (bz_counts_get) is updated as by bz_assign_run.  d_ext must be 8-byte aligned when
 * ext_stride is a multiple of 8. */
#define BZ_NOALLOC_MAX_ROWS 262144
int bz_assign_device_noalloc(bz_handle h, const uint8_t *d_ext, int ext_stride, const uint8_t *d_ext_len,
                             const uint8_t *d_bc_off, const uint8_t *d_bc_len, int64_t n, int max_ed, int32_t *d_hit,
                             int8_t *d_ed, int8_t *d_end_idx, uint8_t *d_flags, void *stream);

/* Single pair on the device: the drop-in for fast_edit_distance.sub_edit_distance(a, b, max_ed)
 * -> (ed, end_idx) and edit_distance(a, b, max_ed) -> ed (read_assignment.py:1,72; main.py:24,250).
 * Returns the distance (max_ed+1 above the cut-off, end -1) or a negative error. */
int bz_sub_edit_distance(bz_handle h, const char *a, int la, const char *b, int lb, int max_ed, int *end_idx);
int bz_edit_distance(bz_handle h, const char *a, int la, const char *b, int lb, int max_ed);

/* ---- stage 1: putative barcode search (polyT_adaptor_finder.py class Read, main.py:43-106) ------------------ */
/* Per read r = seq[off[r] .. off[r+1]) with its quality string at the same offsets of `qual`:
 *   pass_code     adaptor_polyT_pass of polyT_adaptor_finder.py:222-244: 0 found, 1 none, 2 both strands, 10 several
 *                 adaptors; -1 where the reference raises (empty quality slice, :278)
 *   strand        '+' / '-' or 0 (:261), raw_bc_start the adaptor end (:260), min_q raw_bc_min_q (:278),
 *   polyT_end     polyT_trimming_idx (:344-379; 0 when pass_code != 0)
 * kit_5prime = 0 for kits 3v2 / 3v3 / 3v4 (find_adaptor_3_prime, :85-164), 1 otherwise (find_adapter_5_prime, :43-83);
 * umi_len as parser.py:180-191.  The end-index tie rule of sub_edit_distance is BZ_OPT_TIE_RULE (it decides
 * raw_bc_start when several adaptor ends tie).  Every CSV field of main.py:96-105 is a slice of the read at these
 * numbers (the host slices).  Synchronous. */
int bz_putative_bc(bz_handle h, const char *seq, const char *qual, const int64_t *off, int64_t n, int kit_5prime, int umi_len,
                   int32_t *pass_code, int8_t *strand, int32_t *raw_bc_start, int32_t *min_q, int32_t *polyT_end);

/* ---- empty-drop test (stage 2, main.py:244-256 and :269-281) --------------------------------------------- */
/* far_out[i] = 1 iff every string of B is at global edit distance >= min_ed from string i of A, i.e.
 * `min([edit_distance(k, x, max_ed=min_ed) for x in cells_bc.keys()]) >= min_ed` of main.py:250,275 for k = A[i]
 * (A = empty-drop candidates, B = the selected cell barcodes, min_ed = DEFAULT_EMPTY_DROP_MIN_ED = 5, config.py:64).
 * Strings are given as in bz_set_whitelist (bytes + n+1 offsets), each at most BZ_EXT_MAX long.  Synchronous. */
int bz_far_from_all(bz_handle h, const char *a_bytes, const int64_t *a_off, int64_t n_a, const char *b_bytes,
                    const int64_t *b_off, int64_t n_b, int min_ed, uint8_t *far_out);

/* ---- whole step 3 over files ------------------------------------------------------------------------ */
/* Native host pipeline = assign_read's batch loop (read_assignment.py:253-330) with its generator
 * (:138-181, helper.py:275-301), per-batch epilogue (:87-99, :214-247) and ordered concatenation
 * (:293-327): FASTQ(.gz) files + putative_bc.csv in lock step -> device batches on the engines' two
 * slots each (round robin, drained in submit order) -> trimmed / re-stranded FASTQ records of the
 * assigned reads, plain or one gzip member per `batch_size` reads per input file (gz != 0).
 * The whitelist must have been set, in the same order, on every engine.  `batch_size` is the
 * reference's --batch-size (parser.py:67), `device_batch_rows` the rows per device submit,
 * `n_threads` the host threads for parsing / formatting / compression.
 * out_counts: [0] reads assigned (demul_count_tot), [1] reads seen (count_tot), [2] device batches.
 * Returns BZ_E_MISMATCH (message = the reference's) when a csv row's read_id differs from its read. */
int bz_assign_files(bz_handle *engines, int n_engines, const char *const *fastq_fns, int n_fastq,
                    const char *putative_bc_csv, const char *fastq_out, int gz, int restrand, int max_ed,
                    int batch_size, int64_t device_batch_rows, int n_threads, int64_t out_counts[4]);

/* The same for a rank that owns a contiguous run of the job's FASTQ files (one process per GPU, reads sharded by
 * file): `csv_skip_rows` = reads in the files before this rank's, whose rows of the table are skipped after the header.
 * Concatenating the ranks' outputs in rank order gives bz_assign_files' output over all the files (reference batches
 * never span files, read_assignment.py:149-158).  blaze_amd.dist.assign_read_sharded drives it. */
int bz_assign_files_from(bz_handle *engines, int n_engines, const char *const *fastq_fns, int n_fastq,
                         const char *putative_bc_csv, int64_t csv_skip_rows, const char *fastq_out, int gz, int restrand,
                         int max_ed, int batch_size, int64_t device_batch_rows, int n_threads, int64_t out_counts[4]);
/* The same without counting anybody's reads: the rank's first row of the (plain) table is the one whose read_id is the id
 * of the first read of its first non-empty file.  bz_fastq_first_read_id gives that id (len 0: the file holds no read),
 * bz_csv_locate_read searches the mapped table for it outward from hint_fraction of its size (bytes of FASTQ before the
 * rank's files / all FASTQ bytes) and gives the row's byte offset (-1: not found, or the table is gzipped: count instead),
 * bz_assign_files_at starts the table there.  A wrong offset cannot go unnoticed: every row's read_id is compared with
 * its read as in the reference (read_assignment.py:219-223, BZ_E_MISMATCH).  Host only but for the last. */
int bz_fastq_first_read_id(const char *path, char *out, int cap, int *len);
int bz_csv_locate_read(const char *putative_bc_csv, const char *read_id, int id_len, double hint_fraction, int64_t *byte_offset);
int bz_assign_files_at(bz_handle *engines, int n_engines, const char *const *fastq_fns, int n_fastq,
                       const char *putative_bc_csv, int64_t csv_byte_offset, const char *fastq_out, int gz, int restrand,
                       int max_ed, int batch_size, int64_t device_batch_rows, int n_threads, int64_t out_counts[4]);
/* Reads in one FASTQ(.gz) file as helper.fastq_parser (helper.py:292-301) would yield them (lines / 4); host only. */
int bz_fastq_count_reads(const char *path, int64_t *n_reads);

/* ---- whole stage 1 over files ----------------------------------------------------------------------- */
/* Native host pipeline = the stage-1 loop of main.py:344-364 with its generator (read_batch_generator,
 * main.py:304-323; helper.py:275-301) and the table of get_raw_bc_from_reads (main.py:96-105): FASTQ(.gz)
 * files -> device batches of whole `batch_size`-read reference batches through bz_putative_bc (one search in
 * flight per engine, round robin, drained in order) -> putative_bc.csv, byte for byte what the reference's
 * `DataFrame.to_csv` calls write (int / float / blank number cells per reference batch, csv quoting).
 * `min_q` is --minQ (it only decides the low-quality tag 100 of main.py:91-92), `device_batch_reads` the reads
 * per device submit, `n_threads` the host threads for packing / formatting.
 * out_counts: [0] reads seen, [1..5] reads with pass code 0 / 1 / 2 / 10 / 100 (raw_bc_pass_count of
 * main.py:355-360), [6] device batches.  A read whose barcode slice is empty makes the reference raise
 * (polyT_adaptor_finder.py:278): BZ_E_ARG with that message.  No file is created when there are no reads. */
int bz_putative_bc_files(bz_handle *engines, int n_engines, const char *const *fastq_fns, int n_fastq, const char *out_csv,
                         int kit_5prime, int umi_len, int min_q, int batch_size, int64_t device_batch_reads, int n_threads,
                         int64_t out_counts[8]);

/* ---- per-barcode counts + the one collective -------------------------------------------- */
/* Device-resident int64 histogram counts[0..n_wl) of assigned rows per barcode, followed by
 * counts[n_wl] = rows assigned, counts[n_wl+1] = rows seen (the two scalars the reference sums
 * at read_assignment.py:299-301,324-326).  Every bz_assign_run adds to it -- not necessarily before it returns control to
 * the stream's next kernel: the counts of a slot's last run may ride along with the slot's next run (two short kernels
 * less per step); bz_counts_get / bz_counts_reset / bz_comm_allreduce_counts / bz_slot_sync / bz_slot_timing and
 * bz_set_whitelist run whatever is still pending first, so every reader sees every run counted exactly once. */
int bz_counts_reset(bz_handle h);
int bz_counts_get(bz_handle h, int64_t *out, int64_t n_out);     /* n_out = n_wl + 2 */
/* RCCL over xGMI, one process per GPU: rank 0 makes the id, the host passes it round,
 * every rank inits, then the all-reduce sums the histogram in place on the device. */
int bz_comm_unique_id(uint8_t id[128]);
int bz_comm_init(bz_handle h, const uint8_t id[128], int n_ranks, int rank);
int bz_comm_allreduce_counts(bz_handle h);
int bz_comm_allreduce_f64(bz_handle h, double *inout, int n, int op); /* op 0 sum, 1 max; doubles as barrier */
/* Exact sum over ranks of a host vector of n int64 (any n; staged through a grow-only device buffer): the per-rank read
 * counts and part sizes of the file-sharded runs (blaze_amd/dist.py), which must not round-trip through doubles. */
int bz_comm_allreduce_i64(bz_handle h, int64_t *inout, int64_t n);
int bz_comm_destroy(bz_handle h);

/* ---- the gzip member writer on its own (host only, no GPU needed) ------------------------------------- */
/* One complete gzip member of `text` as bz_assign_files writes it per batch at gz level 0; replaces
 * gzip.compress(out_buffer.encode()) of read_assignment.py:244-245 (same decompressed bytes).  `out` needs
 * bz_gzip_fastq_bound(n) bytes at most. */
int64_t bz_gzip_fastq_bound(int64_t n);
int bz_gzip_fastq_member(const char *text, int64_t n, uint8_t *out, int64_t cap, int64_t *out_len);

/* ---- measurement helper ------------------------------------------------------------------ */
/* Independent-chain 32-bit integer VALU micro-kernel (v_and_or / v_lshl_or / v_or3 / v_xor mix):
 * measured lane-ops per second = the empirical ceiling the scan kernel is priced against. */
int bz_valu_peak_probe(bz_handle h, int iters, double *lane_ops_per_s, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* BLAZE_HIP_H */
