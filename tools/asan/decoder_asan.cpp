// Host-only AddressSanitizer / UBSan harness for the BAM decoder (csrc/bam.cpp + csrc/api.cpp): the device side is
// stubbed out, the handle is a decode-only one.  Every file given on the command line is opened and decoded to the
// end in both modes (with and without sequences); corrupt input has to end in an error code, never in a bad access.
// Each decode runs twice: batch by batch (xck_bam_next_batch: parse in place) and through xck_ingest_bam in slices with a
// stand-in for engine_push_block - the path a GPU-backed handle takes (parse of a chunk and its push behind the coordinator
// on the push thread, csrc/bam.cpp IngestJob / Pusher); both must see the same records, field for field.
//   make -C tools/asan && tools/asan/decoder_asan FILE.bam...
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/xck.h"
#include "../../xcltk_amd/csrc/xck_internal.h"
#include "../../xcltk_amd/csrc/inflate_dev.h"

namespace xck {   // what engine.hip provides; never reached through a decode-only handle
int  engine_create(const xck_config*, xck_engine*) { return XCK_E_ARG; }
void engine_destroy(xck_engine*) {}
int  engine_push(xck_engine*, const xck_batch*, bool) { return XCK_E_ARG; }
int  engine_flush(xck_engine*) { return XCK_E_ARG; }
int  engine_finish(xck_engine*, xck_result*) { return XCK_E_ARG; }
int  engine_finish_async(xck_engine*) { return XCK_E_ARG; }
int  engine_result_device(xck_engine*, xck_result*) { return XCK_E_ARG; }
int  engine_reset(xck_engine*) { return XCK_E_ARG; }
int  engine_stats(const xck_engine*, xck_stats*) { return XCK_E_ARG; }
int  engine_umi_bits(const xck_engine* e) { return e ? e->umi_bits : 0; }
int  engine_numa_node(const xck_engine*) { return -1; }
int  engine_device(const xck_engine*) { return -1; }                          // no device: the decoder's GPU share of the inflate never starts
GpuInflateSlot* gpu_inflate_slot_create(int, int, bool) { return nullptr; }
void gpu_inflate_slot_destroy(GpuInflateSlot*) {}
bool gpu_inflate_slot_reserve(GpuInflateSlot*, size_t, size_t, size_t) { return false; }
int  gpu_inflate_slot_launch(GpuInflateSlot*, size_t, size_t, size_t) { return -1; }
int  gpu_inflate_slot_wait(GpuInflateSlot*) { return -1; }
bool gpu_inflate_slot_done(GpuInflateSlot*) { return true; }
// stand-in for the engine's block push (called on the decoder's push thread): touches everything the batches claim to own
static unsigned long long g_push_sum = 0; static long g_push_recs = 0;
static unsigned long long batch_sum(const xck_batch& bt) {
    unsigned long long s = 0;
    for (int64_t i = 0; i < bt.n_reads; i++) s += (unsigned)bt.pos[i] + bt.flag[i] + bt.mapq[i] + (unsigned)bt.cell[i] + bt.umi[i] + (bt.cig_off[i] - bt.cig_off[0]);
    if (bt.n_reads) { for (uint32_t c = bt.cig_off[0]; c < bt.cig_off[bt.n_reads]; c++) s += bt.cigar[c];
                      if (bt.seq) for (uint32_t q = bt.seq_off[0]; q < bt.seq_off[bt.n_reads]; q++) s += bt.seq[q]; }
    return s + (unsigned long long)bt.contig * 1000003ull + bt.ordinal_base;
}
int  engine_push_block(xck_engine*, const void* base, size_t bytes, const xck_batch* bts, int n, void**) {
    for (int i = 0; i < n; i++) {
        if (bts[i].n_reads > 0 && ((const char*)bts[i].pos < (const char*)base || (const char*)(bts[i].pos + bts[i].n_reads) > (const char*)base + bytes)) return XCK_E_ARG;
        g_push_sum += batch_sum(bts[i]); g_push_recs += bts[i].n_reads;
    }
    return XCK_OK;
}
void engine_release_staging(xck_engine*) {}
void fence_wait(void*) {}
void fence_destroy(void*) {}
void* pinned_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }       // plain heap: ASAN sees every overrun
void  pinned_free(void* p) { free(p); }
}

int main(int argc, char** argv) {
    std::vector<std::string> bcs;
    for (int i = 0; i < 64; i++) { char b[32]; snprintf(b, sizeof b, "ACGTACGTACGT%04d-1", i); bcs.push_back(b); }
    std::vector<const char*> bp; for (auto& s : bcs) bp.push_back(s.c_str());
    long ok = 0, bad = 0, recs = 0;
    for (int a = 1; a < argc; a++) {
        const size_t al = strlen(argv[a]);
        if (al > 4 && (!strcmp(argv[a] + al - 4, ".tsv") || !strcmp(argv[a] + al - 4, ".vcf"))) {       // SNP lists: the text parser
            for (int vcf = 0; vcf < 2; vcf++) {
                xck_snp_text* t = nullptr;
                const int rc = xck_parse_snp_text(argv[a], vcf, &t);
                if (rc == 0 && t) { unsigned long long s = 0; for (int64_t i = 0; i < t->n; i++) s += (unsigned)t->chrom_id[i] + (unsigned long long)t->pos[i] + t->ref[i] + t->alt[i] + t->ref_hap[i] + t->alt_hap[i];
                                    for (int c = 0; c < t->n_chroms; c++) s += strlen(t->chroms[c]); if (s == 0x123456789abcull) puts(""); recs += t->n; ok++; xck_free_snp_text(t); }
                else bad++;
            }
            continue;
        }
        for (int mode : {XCK_MODE_BASEFC, XCK_MODE_BAF}) {
            const int t_env = getenv("XCK_ASAN_THREADS") ? atoi(getenv("XCK_ASAN_THREADS")) : 0;      // e.g. 24: the thread count of a run to be re-checked
            for (int threads : {t_env > 0 ? t_env : 1, 3}) {
                xck_config cfg; memset(&cfg, 0, sizeof cfg);
                cfg.struct_size = sizeof cfg; cfg.mode = mode; cfg.n_cells = (int)bp.size(); cfg.barcodes = bp.data();
                cfg.cell_tag[0] = 'C'; cfg.cell_tag[1] = 'B'; cfg.umi_tag[0] = 'U'; cfg.umi_tag[1] = 'B';
                cfg.flags = XCK_F_DECODE_ONLY | (threads == 3 ? XCK_F_VERIFY_CRC : 0); cfg.n_threads = threads; cfg.max_batch_reads = 777;
                cfg.n_snps = mode == XCK_MODE_BAF ? 0 : 0;
                xck_engine* e = nullptr;
                if (xck_create(&cfg, &e) != XCK_OK) { fprintf(stderr, "create failed: %s\n", xck_last_error(nullptr)); return 2; }
                xck_bam* b = nullptr; char err[256] = {0};
                int rc = xck_bam_open(argv[a], threads, &b, err, sizeof err);
                if (rc == 0) {
                    const int nref = xck_bam_n_refs(b);
                    std::vector<int32_t> t2c(nref > 0 ? nref : 1);
                    for (int t = 0; t < nref; t++) t2c[t] = t;
                    xck_ingest_opts o; memset(&o, 0, sizeof o); o.struct_size = sizeof o; o.sample = -1; o.tid_to_contig = t2c.data();
                    o.use_index = getenv("XCK_ASAN_INDEX") ? 1 : 0;          // PATH.bai is untrusted input as well
                    if (o.use_index) for (int t = 1; t < nref; t += 2) t2c[t] = -1;
                    xck_batch bt;
                    unsigned long long sum_pull = 0; long recs_pull = 0;
                    while ((rc = xck_bam_next_batch(e, b, &o, &bt)) > 0) {
                        recs += bt.n_reads; recs_pull += bt.n_reads;
                        sum_pull += xck::batch_sum(bt);                  // touches everything the batch claims to own
                    }
                    xck_bam_close(b);
                    // the same file through xck_ingest_bam in slices, as a GPU-backed handle would take it (the stand-in above receives the chunks)
                    b = nullptr;
                    if (xck_bam_open(argv[a], threads, &b, err, sizeof err) == 0) {
                        xck::g_push_sum = 0; xck::g_push_recs = 0;
                        e->n_impl = 1;                                   // (no engine behind it: engine_push_block is the stand-in)
                        if (xck_bam_prefetch(e, b, &o) < 0) { /* the ingest below reports it */ }   // (read ahead first, as Engine.ingest_bams does for the next file of a list)
                        o.pause_records = 3000;
                        int64_t n = 0; int rc2;
                        while ((rc2 = xck_ingest_bam(e, b, &o, &n)) == 1) {}
                        e->n_impl = 0;
                        xck_bam_close(b);
                        if ((rc2 == 0) != (rc == 0) || (rc == 0 && (xck::g_push_recs != recs_pull || xck::g_push_sum != sum_pull))) {
                            fprintf(stderr, "%s: pull decode rc %d, %ld records, sum %llx - ingest rc %d, %ld records, sum %llx\n", argv[a], rc, recs_pull, sum_pull, rc2, xck::g_push_recs, xck::g_push_sum);
                            return 3;
                        }
                    }
                }
                (rc == 0 ? ok : bad)++;
                xck_destroy(e);
            }
        }
    }
    printf("%d files: %ld clean decodes, %ld rejected, %ld records seen\n", argc - 1, ok, bad, recs);
    return 0;
}
