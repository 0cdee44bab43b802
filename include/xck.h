/* xck.h - C-ABI of the MI355X-native cell x feature / cell x SNP counting engine (libxck.so).
 *
 * This is the drop-in boundary for the ONE hot path of hxj5/xcltk:
 *   - RDR `basefc`  : per-region UMI/read counting        (reference xcltk/rdr/fc/core.py:69-178)
 *   - BAF  pileup   : per-region allele-specific counting (reference xcltk/baf/fc/core.py:42-247)
 * The reference has no FFI; its seams are the Python functions fc_features()/fc_fet1()/plp_snp().
 * Each entry point below names the reference interface it replaces.  A reference-side binding
 * (ctypes) is shown in INTEGRATION.md.
 *
 * Conventions: plain C, plain pointers and sizes, no torch / HIP types.  Every call returns an
 * int status (0 = ok, <0 = error; xck_last_error() gives the text) unless stated otherwise.
 * One engine handle drives one GPU; calls on one handle must be serialised by the caller;
 * different handles may be used from different host threads.  No global state, no callbacks.
 * There is NO CPU fallback: if no HIP device is usable xck_create() fails.
 */
/* Environment.  A handle reads its knobs ONCE, at xck_create() (csrc/api.cpp Knobs::from_env): later changes of the
 * environment do not reach a live handle, and no push / finish call looks at the environment.  None of them changes a result.
 *   XCK_DEBUG_TIMING          (set)       stage timings and path decisions on stderr
 *   XCK_FOLD=sort                         basefc fold by radix sort instead of the partition fold (tests; xck_stats.fold_path says which ran)
 *   XCK_FOLD_C=<keys>                     page size of the partition folds (tests: small pages reach every path with small inputs)
 *   XCK_FOLD_LGG=<l>                      at most 2^l cell groups per row in the basefc fold (default 6, or one per cell up to 512 cells)
 *   XCK_FOLD_COPIES_LG=<l>                2^l copies of the level-1 counters / cursors (default 4)
 *   XCK_FOLD_BUCKET_BLOCKS, XCK_FOLD_OVERLAP=0|1, XCK_FOLD_OVERLAP_BLOCKS      grids of the work-item pass, its second stream
 *   XCK_FULL_SORT=1                       radix-sort fold over all key bits
 *   XCK_PILEUP_SORT=radix                 library radix sorts for both pileup stages (the fallback path, forced)
 *   XCK_PILEUP_HAP=sorted|values          region-level hits: sorted items + k_hap_class, or class in a value word instead of packed bits
 *   XCK_PILEUP_ITEM_SORT=bitonic          LDS item sort by the bitonic network
 *   XCK_PILEUP_LGG=<l>                    at most 2^l cell groups per SNP in the pileup partitions (default 10)
 *   XCK_HIT_CAP0, XCK_HIT_SLACK           first capacity / head room of the hit accumulators (tests: reach the overflow-replay path)
 *   XCK_PUSH_STAGE=0|1, XCK_PUSH_STAGE_BYTES   xck_push_batch: packed one-copy form always / never / below this size (default 2 MB)
 *   XCK_GPU_INFLATE=auto|<percent>|0      share of the BGZF chunks that xck_ingest_bam inflates on the handle's GPU (csrc/inflate_dev.hip; record walk and
 *                                         parse stay on the host).  auto (the default): files of at least XCK_GPU_INFLATE_MIN_MB (96) compressed MB keep
 *                                         XCK_GPU_INFLATE_DEPTH (10) chunks on the device and leave the rest to the host pool, XCK_GPU_INFLATE_RING (12)
 *                                         chunks in flight in all; <percent>: a fixed share; 0 = host only.  XCK_GPU_INFLATE_FREE_CUS (32): CUs the
 *                                         inflate streams never use.  Bit-identical results either way (a block the kernel does not finish, and every
 *                                         chunk after a runtime error, is inflated by the host); off for handles without a device and with XCK_F_VERIFY_CRC.
 * Decoder (read when a BAM is opened or once per process): XCK_THREADS, XCK_NUMA=0, XCK_INFLATE=zlib, XCK_CHUNK_BYTES,
 * XCK_WRITE_THREADS (writer threads of xck_write_mtx), XCK_TEST_INTERN_LIMIT (tests). */
#ifndef XCK_H
#define XCK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XCK_ABI_VERSION 3   /* 3: xck_stats.fold_path / fold_fallbacks / pileup_sort_path / fold_refinements / pileup_sort2_path; 2: xck_config.n_excl_pairs / excl_region / excl_snp, xck_ingest_opts.pause_records (older, shorter xck_config / xck_ingest_opts are still accepted: they carry struct_size.  xck_stats does not: xck_get_stats writes the whole ABI-3 struct, so its caller must be built against this header - check xck_abi_version() first) */

/* status codes */
#define XCK_OK            0
#define XCK_E_ARG        -1   /* invalid argument / configuration            */
#define XCK_E_DEVICE     -2   /* HIP runtime error, or no device             */
#define XCK_E_NOMEM      -3
#define XCK_E_IO         -4   /* file / BGZF / BAM format error              */
#define XCK_E_STATE      -5   /* call sequence error                         */
#define XCK_E_CAPACITY   -6   /* key space exhausted (too many interned UMIs)*/

/* counting modes */
#define XCK_MODE_BASEFC   1   /* feature x cell counts        (xcltk basefc)        */
#define XCK_MODE_BAF      2   /* feature x cell AD / DP / OTH (xcltk baf, step 3)   */
#define XCK_MODE_BOTH     3   /* both from ONE decode of the BAM */

/* UMI / read-name key code meaning "no usable key" (tag missing or empty string;
 * reference: check_read() -12, rdr/fc/mcount.py:41, baf/fc/mcount.py:116-117) */
#define XCK_UMI_NONE  0xFFFFFFFFFFFFFFFFull

/* A region ("feature"): 1-based inclusive start/end exactly as in the region TSV
 * (reference load_region_from_txt, rdr/fc/utils.py:10-45).  `contig` indexes the caller's
 * contig table (names with any leading "chr" stripped, utils/grange.py:263).
 * Output row index of a region == its index in this array. */
typedef struct xck_region {
    int32_t contig;
    int32_t start;      /* 1-based inclusive */
    int32_t end;        /* 1-based inclusive */
} xck_region;

/* A phased het SNP (reference SNP class, baf/fc/gfeature.py:8-39). ref/alt are upper-case
 * ASCII in "ACGTN"; ref_hap/alt_hap in {0,1} are the haplotype index of each allele. */
typedef struct xck_snp {
    int32_t contig;
    int32_t pos;        /* 1-based */
    uint8_t ref, alt;
    uint8_t ref_hap, alt_hap;
} xck_snp;

/* Engine configuration.  Replaces the fields of the reference Config objects that the hot
 * path reads (rdr/fc/config.py:5-41, baf/fc/config.py:8-60). */
typedef struct xck_config {
    uint32_t struct_size;       /* sizeof(xck_config), for ABI checking                      */
    int32_t  mode;              /* XCK_MODE_* (BOTH: every pushed batch feeds both pipelines) */
    int32_t  device;            /* HIP device ordinal                                        */
    /* read filter = check_read(), rdr/fc/core.py:46-62 == baf/fc/core.py:18-34 */
    double   min_mapq;          /* drop if mapq < min_mapq                                   */
    int32_t  min_len;           /* drop if #aligned (M/=/X) bases < min_len                  */
    uint32_t incl_flag;         /* if non-zero: drop unless flag & incl_flag                 */
    uint32_t excl_flag;         /* if non-zero: drop if flag & excl_flag                     */
    int32_t  no_orphan;         /* drop paired reads that are not proper pairs               */
    /* basefc include criterion, rdr/fc/core.py:160-165: 0 < v < 1 -> fraction (IEEE double
     * m/float(n) < v drops), otherwise length (m < v drops).                                */
    double   min_include;
    /* BAF: per-SNP filters (baf/fc/core.py:238-246) and haplotype rule (:181-192) */
    double   min_count;
    double   min_maf;
    int32_t  no_dup_hap;
    /* tables */
    int32_t  n_cells;           /* number of matrix columns (barcodes or sample ids)         */
    int32_t  n_contigs;
    int32_t  n_regions;
    const xck_region* regions;
    int32_t  n_snps;            /* BAF only                                                  */
    const xck_snp*    snps;
    /* decoder side (used only by xck_ingest_bam / xck_bam_next_batch) */
    const char* const* barcodes;  /* n_cells NUL-terminated barcodes, or NULL = well mode
                                     (column = sample index of the BAM)                      */
    char     cell_tag[4];       /* e.g. "CB"; "" when barcodes == NULL                       */
    char     umi_tag[4];        /* e.g. "UB"; "" = key is the read name                      */
    /* sizing */
    int64_t  max_batch_reads;   /* upper bound on reads per xck_push_batch (0 = default)     */
    int32_t  n_threads;         /* host decode threads (0 = the CPUs this process may use)   */
    int32_t  flags;             /* XCK_F_*                                                   */
    /* BAF: (region, SNP) pairs left OUT of the SNP -> region join although start <= pos <= end.  Region-wise
     * local phasing drops the SNPs without coverage in the cellsnp data from that region's list
     * (baf/fc/phasing.py:44-49; the SNP still counts in other regions that contain it).  Indices into
     * regions[] / snps[].  Read only when struct_size covers the fields.                      */
    int32_t  n_excl_pairs;
    const int32_t* excl_region;
    const int32_t* excl_snp;
} xck_config;

#define XCK_F_FORCE_KEY128   1  /* always use 128-bit sort keys (testing)                    */
#define XCK_F_VERIFY_CRC     2  /* verify BGZF CRC32 while decoding                          */
#define XCK_F_LOW_PRIORITY   8  /* run this engine's kernels on a low-priority HIP stream: lets a second
                                   engine fill the GPU while the first one copies results out */
#define XCK_F_DECODE_ONLY    4  /* handle drives the BAM decoder only: no GPU is touched, and
                                   xck_push_batch / xck_finish fail (used to run the host
                                   ingest on machines without a device; NOT a compute path)  */

/* One batch of decoded alignment records, structure-of-arrays, all reads on ONE contig,
 * in file order.  This is what the reference obtains record by record from
 * pysam.AlignmentFile.fetch() (utils/sam.py:85-118).  Host pointers (ideally pinned). */
typedef struct xck_batch {
    int32_t  contig;            /* engine contig id; <0 : batch is skipped                   */
    int32_t  n_reads;
    uint64_t ordinal_base;      /* fetch-order ordinal of read 0: (bam_index << 40) | record#;
                                   read i has ordinal_base + i  (baf/fc/mcount.py:118-119:
                                   first read of a UMI in fetch order decides the allele)    */
    const int32_t*  pos;        /* [n]   0-based leftmost reference coordinate               */
    const uint16_t* flag;       /* [n]   BAM FLAG                                            */
    const uint8_t*  mapq;       /* [n]                                                       */
    const int32_t*  cell;       /* [n]   column index; <0 = tag missing / not in list        */
    const uint64_t* umi;        /* [n]   key code (see xck_umi_bits) or XCK_UMI_NONE         */
    const uint32_t* cig_off;    /* [n+1] offsets into cigar[]                                */
    const uint32_t* cigar;      /* BAM CIGAR words (len << 4 | op)                           */
    const uint32_t* seq_off;    /* [n+1] BYTE offsets into seq[] (BAF only, else NULL)       */
    const uint8_t*  seq;        /* BAM 4-bit packed bases, each read starts on a byte        */
} xck_batch;

/* Sparse result in coordinate form, sorted by (row, col), no zero entries.
 * row = region index (0-based, input order), col = cell index (0-based). Engine-owned
 * host memory, valid until xck_destroy(). */
typedef struct xck_coo {
    int64_t nnz;
    const int32_t* row;
    const int32_t* col;
    const int32_t* val;
} xck_coo;

typedef struct xck_result {
    xck_coo count;              /* XCK_MODE_BASEFC: matrix.mtx  (rdr/fc/core.py:109-116)     */
    xck_coo ad, dp, oth;        /* XCK_MODE_BAF: AD/DP/OTH.mtx  (baf/fc/core.py:84-99)       */
} xck_result;

typedef struct xck_stats {
    int64_t n_batches;
    int64_t n_reads;            /* records pushed (every decoded BAM record counts)          */
    int64_t n_hits;             /* (read,region) or (read,SNP) pairs accepted by the join    */
    int64_t n_hits_unique;      /* keys that reached HBM after the in-LDS de-duplication     */
    double  ms_h2d;             /* host-measured, cumulative                                 */
    double  ms_device;          /* HIP-event time of all kernels, cumulative                 */
    double  ms_join;            /* HIP-event time of the join/pileup kernels only            */
    double  ms_sort;            /* HIP-event time of sort + reduce kernels                   */
    double  ms_d2h;             /* HIP-event time of the result copy-out (copy stream)       */
    int64_t algo_bytes_join;    /* algorithmic bytes of the join kernels         */
    int64_t n_join_launches;    /* fused join kernel launches (device-resident batches are fused) */
    int32_t key_bits;           /* 64 or 128                                                 */
    int32_t umi_bits;
    int32_t fold_path;          /* basefc fold of the last xck_finish: 0 none yet, 1 partition fold (no sort), 2 radix-sort fold */
    int32_t fold_fallbacks;     /* finishes of this handle in which the partition fold handed over to the radix-sort fold   */
    int32_t pileup_sort_path;   /* pileup hits of the last xck_finish: 0 none yet, 1 row partition + LDS sort per item, 2 radix sort */
    int32_t fold_refinements;   /* partition fold of the last xck_finish: times the level-2 geometry had to be refined (uneven cells) */
    int32_t pileup_sort2_path;  /* pileup, region-level hits of the last xck_finish: 0 none, 1 partition + per-item hash classification (no sort), 2 radix sort, 3 partition + LDS sort per item */
    int32_t gpu_inflate_chunks; /* BGZF chunks (~740 blocks each) whose inflate ran on the GPU since the last xck_reset (XCK_GPU_INFLATE; was reserved0) */
} xck_stats;

typedef struct xck_engine xck_engine;     /* opaque: one per GPU */
typedef struct xck_bam    xck_bam;        /* opaque: one open BAM file */

/* -- library ------------------------------------------------------------------------------- */
const char* xck_version(void);
int         xck_abi_version(void);
int         xck_device_count(void);              /* number of usable HIP devices (0 if none) */
const char* xck_last_error(const xck_engine* e); /* e may be NULL: last error of the creating thread */

/* -- engine (replaces fc_features()/fc_fet1()/plp_snp(): the per-region fetch loops) -------- */
int  xck_create(const xck_config* cfg, xck_engine** out);
void xck_destroy(xck_engine* e);
/* number of bits available for a UMI / read-name key code in xck_batch.umi (26..64):
 *   ACGT-only key of L bases with 2L+1 <= bits-1  ->  (1 << 2L) | 2-bit packed bases (A0 C1 G2 T3)
 *   anything else                                   ->  (1 << (bits-1)) | interned id            */
int  xck_umi_bits(const xck_engine* e);
/* Copy one batch to the GPU (async, overlapped with kernels of the previous batch) and run
 * the join / pileup kernels on it.  The batch arrays may be reused once the call returns.
 * Batches of 2 MB and more cross PCIe straight from the caller's arrays (nine DMA copies, pinned or not: 0.86 - 1.05 G reads/s
 * measured); smaller ones are packed into one engine-owned pinned block and cross with ONE copy whose completion is an
 * event, not a wait (XCK_PUSH_STAGE_BYTES / XCK_PUSH_STAGE=0|1 override the rule).
 * The host arrays are checked first (one linear pass): cig_off / seq_off must not run backwards,
 * cell[i] < n_cells, contig < n_contigs, no null column - otherwise XCK_E_ARG and nothing is queued. */
int  xck_push_batch(xck_engine* e, const xck_batch* b);
/* Same, but the arrays are DEVICE pointers already resident in HBM (benchmarks, pipelines
 * that decode on the GPU side); no copy is made and they must stay valid until xck_flush().
 * The contents cannot be checked from the host: the same invariants are the caller's word. */
int  xck_push_batch_device(xck_engine* e, const xck_batch* b);
int  xck_flush(xck_engine* e);                    /* wait for all queued device work */
/* Fold all hits into the final sparse matrices (radix sort + segmented reduce on the GPU) and copy
 * them to engine-owned pinned host memory. */
int  xck_finish(xck_engine* e, xck_result* out);
/* Same fold, but returns as soon as the copy-out of the matrices has been ENQUEUED on the engine's copy
 * stream; a following xck_finish() waits for it and hands out the pointers.  Lets the caller run other
 * GPU work (e.g. a second engine) while the matrix crosses PCIe. */
int  xck_finish_async(xck_engine* e);
/* Same matrices as the last xck_finish(), but the pointers are DEVICE addresses ([row|col|val] in the
 * engine's workspace, valid until the next xck_finish / xck_reset): lets a multi-GPU driver exchange the
 * per-contig sparse blocks GPU-to-GPU (RCCL over xGMI) without a host round trip. */
int  xck_get_result_device(xck_engine* e, xck_result* out);
/* Forget all pushed reads, keep tables and buffers (lets one engine be re-used per step). */
int  xck_reset(xck_engine* e);
int  xck_get_stats(const xck_engine* e, xck_stats* out);

/* -- host ingest (replaces pysam.AlignmentFile + fetch(): own BGZF/BAM reader) --------------- */
/* n_threads = 0: the process's CPU share (affinity and cgroup quota; 1.5 threads per CPU behind a quota).  BAM only: CRAM / SAM text
 * are named in `err`.  On a multi-socket host the reader binds its threads - and, from the first xck_ingest_bam / xck_bam_next_batch
 * call until xck_bam_close, the CALLING thread - to the NUMA node of the engine's GPU (XCK_NUMA=0 in the environment turns that off). */
int  xck_bam_open(const char* path, int n_threads, xck_bam** out, char* err, size_t errlen);
void xck_bam_close(xck_bam* b);
int  xck_bam_n_refs(const xck_bam* b);
const char* xck_bam_ref_name(const xck_bam* b, int tid);
int64_t     xck_bam_ref_len(const xck_bam* b, int tid);
/* records per reference from the .bai next to the file (XCK_E_IO if there is no usable index);
 * used to balance contigs over GPUs */
int  xck_bam_ref_records(xck_bam* b, int tid, int64_t* n_mapped, int64_t* n_unmapped);
/* the .bai linear index of one reference: virtual offset of the first record overlapping every 16 kb window (SAMv1 5.1.3);
 * *n = 0 when the index has none.  The array belongs to the reader.  The compressed-byte distance between two windows
 * weighs the reads between them: used to cut an over-weight contig into position windows of equal work. */
int  xck_bam_linear_index(xck_bam* b, int tid, int64_t* n, const uint64_t** voffsets);

typedef struct xck_ingest_opts {
    uint32_t struct_size;
    int32_t  sample;            /* index of this BAM in the BAM list (ordinal high bits; column
                                   index in well mode)                                       */
    const int32_t* tid_to_contig; /* [n_refs] engine contig id per BAM tid, -1 = not used    */
    int32_t  use_index;         /* 1: decode only the virtual-offset ranges of the wanted tids
                                   (needs PATH.bai; silently decodes everything if absent)   */
    int64_t  max_records;       /* stop after this many records (0 = all)                    */
    int64_t  pause_records;     /* xck_ingest_bam only: return 1 ("paused") after the decode chunk in
                                   which at least this many further records were pushed by THIS call;
                                   the reader keeps its position and the next call continues
                                   (0 = run to the end of the file).  Read only when struct_size covers it. */
    /* Position windows (multi-GPU: ONE over-weight contig split at region boundaries): for BAM tid t only the
     * records from the first one that overlaps 0-based position tid_beg[t] (found through the .bai linear index; use_index = 1)
     * up to the last one that STARTS before tid_end[t] are decoded; records that start earlier but reach into the window are
     * included, so neighbouring windows share the reads that straddle the cut.  NULL = whole references.  tid_end[t] <= 0 =
     * no upper bound.  Read only when struct_size covers them. */
    const int32_t* tid_beg;
    const int32_t* tid_end;
} xck_ingest_opts;

/* Decode the BAM with the engine's decoder settings and push every batch; returns the number
 * of records decoded so far (over all calls on this reader) through *n_records.
 * Returns 0 at the end of the file, 1 when paused by pause_records, <0 on error. */
int  xck_ingest_bam(xck_engine* e, xck_bam* b, const xck_ingest_opts* o, int64_t* n_records);
/* Read ahead: the reader's threads start on the first chunks of the file (scanner, inflate - on the host pool or the GPU) and the call
 * returns at once; a later xck_ingest_bam / xck_bam_next_batch with the SAME options (which must stay valid until then) continues from
 * there.  For callers that count many small files one after the other (a plate of per-cell BAMs, the reference's
 * `for sam_fn in sam_fn_list` loops, rdr/fc/core.py:73-76): prefetch file k + 1 before ingesting file k, and the first inflate of the
 * next file overlaps the parse and the joins of this one.  Nothing of the engine is touched.  Returns 0, or <0 on error. */
int  xck_bam_prefetch(xck_engine* e, xck_bam* b, const xck_ingest_opts* o);
/* Pull-style decoding for tests / other consumers: fills *out with the next batch (arrays are
 * owned by the xck_bam and valid until the next call); returns 1 if a batch was produced,
 * 0 at end of file, <0 on error. */
int  xck_bam_next_batch(xck_engine* e, xck_bam* b, const xck_ingest_opts* o, xck_batch* out);

/* -- phased-SNP lists (fast path of load_snp_from_tsv / load_snp_from_vcf, baf/fc/utils.py:51-110, :114-193) --- */
/* The accepted SNPs of a TSV (header line, then chrom pos ref alt ref_hap alt_hap) or a phased VCF (first sample,
 * GT exactly 0|1, 1|0, 0/1, 1/0; single-base REF/ALT in ACGTN, upper-cased), in file order, chrom without a leading
 * "chr".  gzip / bgzip input is read through zlib.  Returns 0, or 1 when the file is outside what this parser
 * reproduces exactly (non-ASCII bytes, carriage returns, a position that is not plain decimal digits): the caller then
 * uses its generic loader.  <0 on I/O errors. */
typedef struct xck_snp_text {
    int64_t n;
    const int32_t* chrom_id;        /* [n] index into chroms[]                               */
    const int64_t* pos;             /* [n] 1-based                                           */
    const char*    ref;             /* [n]                                                   */
    const char*    alt;             /* [n]                                                   */
    const int8_t*  ref_hap;         /* [n] 0 / 1                                             */
    const int8_t*  alt_hap;         /* [n]                                                   */
    int32_t n_chroms;
    const char* const* chroms;      /* in order of first appearance                          */
    int64_t n_rejected;             /* lines the loaders warn about in verbose mode ...      */
    const int64_t* rej_line;        /* [n_rejected] 1-based line numbers, ascending          */
    const int8_t*  rej_code;        /* [n_rejected] XCK_SNP_REJ_*: the first check that failed */
} xck_snp_text;
#define XCK_SNP_REJ_COLUMNS     1   /* too few columns                                        */
#define XCK_SNP_REJ_REF         2   /* invalid REF base                                       */
#define XCK_SNP_REJ_ALT         3   /* invalid ALT base                                       */
#define XCK_SNP_REJ_NO_GT       4   /* VCF: no GT in FORMAT                                   */
#define XCK_SNP_REJ_FORMAT_LEN  5   /* VCF: FORMAT and sample column differ in length         */
#define XCK_SNP_REJ_DELIMITER   6   /* VCF: GT without | or /                                 */
#define XCK_SNP_REJ_GT          7   /* genotype is not 0|1 / 1|0 (0/1, 1/0)                   */
int  xck_parse_snp_text(const char* path, int is_vcf, xck_snp_text** out);
void xck_free_snp_text(xck_snp_text* t);

/* -- output (replaces merge_mtx(), rdr/fc/utils.py:54-93) ------------------------------------ */
/* Write a MatrixMarket file byte-identical to the reference: header
 * "%%MatrixMarket matrix coordinate integer general\n%%\n{nrow}\t{ncol}\t{nnz}\n" then
 * "row\tcol\tval\n" lines (1-based).  row_map[r] gives the 1-based output row of region r
 * (0 = region not written). */
int  xck_write_mtx(const char* path, const xck_coo* m, const int32_t* row_map,
                   int32_t n_rows_out, int32_t n_cols);
/* The same file written by several processes (multi-GPU run on one node; no reference counterpart - its workers hand their
 * triplets to the parent, rdr/fc/main.py:232-262): every process owns some rows.  xck_mtx_part_size gives the bytes and lines
 * of the "row\tcol\tval\n" text of m (no header); after the sizes have been exchanged, xck_write_mtx_part writes that text at
 * byte `offset` of `path` (created if needed, never truncated).  The header line is the caller's. */
int  xck_mtx_part_size(const xck_coo* m, const int32_t* row_map, int64_t* n_bytes, int64_t* n_lines);
int  xck_write_mtx_part(const char* path, int64_t offset, const xck_coo* m, const int32_t* row_map);

#ifdef __cplusplus
}
#endif
#endif /* XCK_H */
