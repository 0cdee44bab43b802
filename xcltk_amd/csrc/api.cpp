// api.cpp - C-ABI entry points of libxck.so (include/xck.h) except the BAM functions (bam.cpp).
#include <hip/hip_runtime_api.h>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <unistd.h>
#include <cerrno>
#include "xck_internal.h"

using namespace xck;

extern "C" {

const char* xck_version(void) { return XCK_VERSION_STR; }
int xck_abi_version(void) { return XCK_ABI_VERSION; }

int xck_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* xck_last_error(const xck_engine* e) { return e ? e->err.c_str() : get_thread_error(); }

// the environment knobs of a handle (include/xck.h "Environment"), read here and nowhere on the push / finish path
Knobs xck::Knobs::from_env() {
    Knobs k;
    auto str = [](const char* n) { const char* e = getenv(n); return e ? e : ""; };
    auto num = [](const char* n, long long d) { const char* e = getenv(n); return e && *e ? atoll(e) : d; };
    k.debug_timing = getenv("XCK_DEBUG_TIMING") != nullptr;
    k.fold_sort = !strcmp(str("XCK_FOLD"), "sort");
    k.fold_c = (int)num("XCK_FOLD_C", 0);
    k.fold_lgg = (int)num("XCK_FOLD_LGG", -1);
    k.fold_copies_lg = (int)num("XCK_FOLD_COPIES_LG", 4);
    k.fold_bucket_blocks = (int)num("XCK_FOLD_BUCKET_BLOCKS", 256 * 4 * 2);
    k.fold_overlap = (int)num("XCK_FOLD_OVERLAP", 1);
    k.fold_overlap_blocks = (int)num("XCK_FOLD_OVERLAP_BLOCKS", 512);
    k.full_sort = num("XCK_FULL_SORT", 0) != 0;
    k.pileup_radix = !strcmp(str("XCK_PILEUP_SORT"), "radix");
    k.pileup_hap = !strcmp(str("XCK_PILEUP_HAP"), "sorted") ? 1 : !strcmp(str("XCK_PILEUP_HAP"), "values") ? 2 : 0;
    k.pileup_bitonic = !strcmp(str("XCK_PILEUP_ITEM_SORT"), "bitonic");
    k.pileup_lgg = (int)num("XCK_PILEUP_LGG", 10);
    k.hit_slack = std::max(0ll, num("XCK_HIT_SLACK", 65536));
    k.hit_cap0 = std::max(64ll, num("XCK_HIT_CAP0", 1ll << 20));
    k.push_stage = (int)num("XCK_PUSH_STAGE", -1);
    k.push_stage_bytes = num("XCK_PUSH_STAGE_BYTES", 2ll << 20);
    k.gpu_inflate_pct = (!*str("XCK_GPU_INFLATE") || !strcmp(str("XCK_GPU_INFLATE"), "auto")) ? -1 : (int)std::max(0ll, std::min(100ll, num("XCK_GPU_INFLATE", 0)));
    k.gpu_inflate_depth = (int)std::max(1ll, std::min(10ll, num("XCK_GPU_INFLATE_DEPTH", 10)));
    k.gpu_inflate_ring = (int)num("XCK_GPU_INFLATE_RING", 12);
    k.gpu_inflate_min_mb = (int)std::max(0ll, num("XCK_GPU_INFLATE_MIN_MB", 96));
    k.gpu_inflate_free_cus = (int)num("XCK_GPU_INFLATE_FREE_CUS", 32);
    return k;
}

int xck_create(const xck_config* cfg_in, xck_engine** out) {
    if (!cfg_in || !out) { set_thread_error("null argument"); return XCK_E_ARG; }
    *out = nullptr;
    // ABI 1 structs (without the exclusion pairs) are accepted: the missing tail reads as zero
    if (cfg_in->struct_size != offsetof(xck_config, n_excl_pairs) && cfg_in->struct_size != sizeof(xck_config)) { set_thread_error("xck_config.struct_size mismatch (ABI)"); return XCK_E_ARG; }
    xck_config cfg_full; memset(&cfg_full, 0, sizeof cfg_full); memcpy(&cfg_full, cfg_in, cfg_in->struct_size); cfg_full.struct_size = sizeof(xck_config);
    const xck_config* cfg = &cfg_full;
    if (cfg->n_excl_pairs < 0 || (cfg->n_excl_pairs > 0 && (!cfg->excl_region || !cfg->excl_snp))) { set_thread_error("invalid exclusion pairs"); return XCK_E_ARG; }
    if (cfg->mode != XCK_MODE_BASEFC && cfg->mode != XCK_MODE_BAF && cfg->mode != XCK_MODE_BOTH) { set_thread_error("invalid mode"); return XCK_E_ARG; }
    if (cfg->n_cells <= 0 || cfg->n_contigs < 0 || cfg->n_regions < 0 || cfg->n_snps < 0) { set_thread_error("invalid table sizes"); return XCK_E_ARG; }
    if ((cfg->n_regions > 0 && !cfg->regions) || (cfg->n_snps > 0 && !cfg->snps)) { set_thread_error("null table pointer"); return XCK_E_ARG; }
    xck_engine* e = new xck_engine();
    e->knobs = Knobs::from_env();
    e->umi_bits = key_layout(cfg).ubits;
    e->mode = cfg->mode;
    e->n_cells = cfg->n_cells; e->n_contigs = cfg->n_contigs;
    if (!(cfg->flags & XCK_F_DECODE_ONLY)) {
        const int modes[2] = { cfg->mode == XCK_MODE_BOTH ? XCK_MODE_BASEFC : cfg->mode, XCK_MODE_BAF };
        const int n = cfg->mode == XCK_MODE_BOTH ? 2 : 1;
        for (int k = 0; k < n; k++) {
            xck_config c = *cfg;
            c.mode = modes[k];
            if (n == 2) c.flags |= XCK_F_LAYOUT_BOTH;          // one key layout -> one decode serves both pipelines
            e->impl = nullptr;
            int rc = engine_create(&c, e);
            e->impls[k] = e->impl; e->n_impl = k + 1;
            if (rc) { set_thread_error(e->err); xck_destroy(e); return rc; }
        }
        e->impl = e->impls[0];
    }
    // decoder settings
    DecodeCfg& d = e->dec;
    d.use_barcodes = cfg->barcodes != nullptr;
    if (d.use_barcodes) { d.build_barcodes(cfg->barcodes, cfg->n_cells); d.cell_tag[0] = cfg->cell_tag[0]; d.cell_tag[1] = cfg->cell_tag[1]; }
    d.use_umi = cfg->umi_tag[0] != 0;
    if (d.use_umi) { d.umi_tag[0] = cfg->umi_tag[0]; d.umi_tag[1] = cfg->umi_tag[1]; }
    d.umi_bits = e->umi_bits;
    d.want_seq = (cfg->mode & XCK_MODE_BAF) != 0;
    d.verify_crc = (cfg->flags & XCK_F_VERIFY_CRC) != 0;
    d.n_threads = cfg->n_threads;
    d.max_batch_reads = cfg->max_batch_reads;
    *out = e;
    return XCK_OK;
}

#define FOR_IMPLS(e, expr) do { if ((e)->n_impl == 0) { (e)->impl = nullptr; int rc_ = (expr); if (rc_) return rc_; }         \
    for (int k_ = 0; k_ < (e)->n_impl; k_++) { (e)->impl = (e)->impls[k_]; int rc_ = (expr); if (rc_) { (e)->impl = (e)->impls[0]; return rc_; } } \
    (e)->impl = (e)->impls[0]; } while (0)

void xck_destroy(xck_engine* e) {
    if (!e) return;
    for (int k = 0; k < 3; k++) {                                        // staging ring of xck_push_batch
        if (e->push_ring.fence[k]) { fence_wait(e->push_ring.fence[k]); fence_destroy(e->push_ring.fence[k]); }
        if (e->push_ring.blk[k]) pinned_free(e->push_ring.blk[k]);
    }
    for (int k = 0; k < e->n_impl; k++) { e->impl = e->impls[k]; engine_destroy(e); }
    if (e->stager) engine_release_staging(e);
    delete e;
}
int xck_umi_bits(const xck_engine* e) { return e ? e->umi_bits : 0; }

// Host arrays handed in through the ABI are checked before any kernel sees them: an offset table that runs backwards or a
// column index outside the cell table would otherwise become an out-of-bounds access on the device or a row outside the
// result.  One linear pass over three of the columns; device-resident batches (xck_push_batch_device) are the caller's word.
static int check_host_batch(xck_engine* e, const xck_batch* b) {
    if (b->n_reads < 0) { e->err = "batch: negative n_reads"; return XCK_E_ARG; }
    if (b->contig < 0 || b->n_reads == 0) return XCK_OK;                       // skipped by the engine
    if (b->contig >= e->n_contigs) { e->err = "batch: contig id outside the configured contigs"; return XCK_E_ARG; }
    const bool seq = (e->mode & XCK_MODE_BAF) != 0;
    if (!b->pos || !b->flag || !b->mapq || !b->cell || !b->umi || !b->cig_off || (!b->cigar && b->cig_off[b->n_reads] != b->cig_off[0]) ||
        (seq && (!b->seq_off || (!b->seq && b->seq_off[b->n_reads] != b->seq_off[0])))) { e->err = "batch: null column"; return XCK_E_ARG; }
    const int64_t n = b->n_reads;
    uint32_t bad = 0;
    for (int64_t i = 0; i < n; i++) bad |= (uint32_t)(b->cig_off[i + 1] < b->cig_off[i]) | (uint32_t)(b->cell[i] >= e->n_cells);
    if (seq) for (int64_t i = 0; i < n; i++) bad |= (uint32_t)(b->seq_off[i + 1] < b->seq_off[i]);
    if (bad) { e->err = "batch: offsets run backwards or a cell index is outside the cell table"; return XCK_E_ARG; }
    return XCK_OK;
}
extern "C++" { namespace xck { int push_trusted(xck_engine* e, const xck_batch* b) { FOR_IMPLS(e, engine_push(e, b, false)); return XCK_OK; } } }

// xck_push_batch, one-copy form for SMALL batches: the caller's nine arrays are packed into ONE engine-owned pinned block (ring of
// three), which crosses PCIe with ONE hipMemcpyAsync into a device staging slot shared by the handle's pipelines
// (engine_push_block - the path the BAM decoder uses).  The call returns as soon as the arrays are packed: the caller may reuse
// them, and the DMA of this batch overlaps the packing of the next one (the block's fence, an event, is waited for only when
// the ring comes round).  Measured on the MI355X box (tools/h2d_bench.py, profiles/r03_a_h2d_*): for batches of 4 M reads the
// direct form (engine_push: nine DMA copies straight from the caller's arrays + one wait) moves 0.86-1.05 G reads/s whether the
// arrays are pinned or plain numpy memory, the packed form 0.6 G reads/s (a host memcpy is slower than the DMA it saves) - so
// the packed form is used where the nine calls and the wait dominate: below XCK_PUSH_STAGE_BYTES (default 2 MB) per batch.
// XCK_PUSH_STAGE=0 / 1 forces one form.
static int push_staged(xck_engine* e, const xck_batch* b) {
    if (e->n_impl <= 0) { e->err = "decode-only handle: no GPU engine behind it"; return XCK_E_STATE; }
    const bool seq = (e->mode & XCK_MODE_BAF) != 0;
    const size_t n = (size_t)b->n_reads;
    const uint32_t c_lo = b->cig_off[0], s_lo = seq ? b->seq_off[0] : 0;
    const size_t n_cig = b->cig_off[n] - c_lo, n_seq = seq ? b->seq_off[n] - s_lo : 0;
    struct Seg { const void* src; size_t bytes, off; } seg[9] = {
        { b->pos, n * 4, 0 }, { b->flag, n * 2, 0 }, { b->mapq, n, 0 }, { b->cell, n * 4, 0 }, { b->umi, n * 8, 0 }, { b->cig_off, (n + 1) * 4, 0 },
        { b->cigar ? b->cigar + c_lo : nullptr, n_cig * 4, 0 }, { seq ? b->seq_off : nullptr, seq ? (n + 1) * 4 : 0, 0 }, { seq && b->seq ? b->seq + s_lo : nullptr, n_seq, 0 } };
    size_t total = 0;
    for (auto& g : seg) { g.off = total; total += (g.bytes + 255) & ~size_t(255); }
    xck_engine::PushRing& ring = e->push_ring;
    const int k = ring.next; ring.next = (ring.next + 1) % 3;
    if (ring.fence[k]) fence_wait(ring.fence[k]);                        // the DMA that last read this block
    if (total > ring.cap[k]) {
        if (ring.blk[k]) pinned_free(ring.blk[k]);
        ring.cap[k] = 0;
        const size_t c = total + total / 4 + (1 << 16);
        ring.blk[k] = pinned_alloc(c);
        if (!ring.blk[k]) { e->err = "out of pinned host memory (xck_push_batch staging)"; return XCK_E_NOMEM; }
        ring.cap[k] = c;
    }
    char* base = (char*)ring.blk[k];
    // pack: the byte range [0, total) of the block is cut into equal parts, one per thread
    auto pack = [&](size_t lo, size_t hi) {
        for (const auto& g : seg) {
            if (!g.bytes || !g.src) continue;
            const size_t a = std::max(lo, g.off), z = std::min(hi, g.off + g.bytes);
            if (a < z) memcpy(base + a, (const char*)g.src + (a - g.off), z - a);
        }
    };
    const int nt = total < (size_t(4) << 20) ? 1 : std::min(4, std::max(1, default_threads()));
    if (nt == 1) pack(0, total);
    else {
        std::vector<std::thread> th;
        const size_t step = ((total + nt - 1) / nt + 63) & ~size_t(63);
        for (int t = 1; t < nt; t++) th.emplace_back(pack, std::min(total, step * t), std::min(total, step * (t + 1)));
        pack(0, std::min(total, step));
        for (auto& t : th) t.join();
    }
    xck_batch sb = *b;
    sb.pos = (const int32_t*)(base + seg[0].off); sb.flag = (const uint16_t*)(base + seg[1].off); sb.mapq = (const uint8_t*)(base + seg[2].off);
    sb.cell = (const int32_t*)(base + seg[3].off); sb.umi = (const uint64_t*)(base + seg[4].off); sb.cig_off = (const uint32_t*)(base + seg[5].off);
    sb.cigar = (const uint32_t*)(base + seg[6].off) - c_lo;               // rebased: never read below c_lo
    sb.seq_off = seq ? (const uint32_t*)(base + seg[7].off) : nullptr;
    sb.seq = seq ? (const uint8_t*)(base + seg[8].off) - s_lo : nullptr;
    return engine_push_block(e, base, total, &sb, 1, &ring.fence[k]);
}
int xck_push_batch(xck_engine* e, const xck_batch* b) {
    if (!e || !b) return XCK_E_ARG;
    if (int rc = check_host_batch(e, b)) return rc;
    if (b->n_reads <= 0 || b->contig < 0 || e->n_impl <= 0) return push_trusted(e, b);     // nothing to copy / decode-only: engine_push reports
    const int force = e->knobs.push_stage;
    const long long small = e->knobs.push_stage_bytes;
    const size_t n = (size_t)b->n_reads;
    const size_t bytes = n * 27 + (size_t)(b->cig_off[n] - b->cig_off[0]) * 4 + ((e->mode & XCK_MODE_BAF) ? n * 4 + (b->seq_off[n] - b->seq_off[0]) : 0);
    const bool stage = force >= 0 ? force != 0 : (long long)bytes < small;
    return stage ? push_staged(e, b) : push_trusted(e, b);
}
int xck_push_batch_device(xck_engine* e, const xck_batch* b) { if (!e || !b) return XCK_E_ARG; FOR_IMPLS(e, engine_push(e, b, true)); return XCK_OK; }
int xck_flush(xck_engine* e) { if (!e) return XCK_E_ARG; FOR_IMPLS(e, engine_flush(e)); return XCK_OK; }
int xck_reset(xck_engine* e) { if (!e) return XCK_E_ARG; FOR_IMPLS(e, engine_reset(e)); e->gpu_inflate_chunks = 0; return XCK_OK; }

int xck_finish_async(xck_engine* e) { if (!e) return XCK_E_ARG; FOR_IMPLS(e, engine_finish_async(e)); return XCK_OK; }

int xck_finish(xck_engine* e, xck_result* out) {
    if (!e || !out) return XCK_E_ARG;
    memset(out, 0, sizeof *out);
    if (e->n_impl == 0) { e->impl = nullptr; return engine_finish(e, out); }
    for (int k = 0; k < e->n_impl; k++) {
        xck_result r;
        e->impl = e->impls[k];
        int rc = engine_finish(e, &r);
        if (rc) { e->impl = e->impls[0]; return rc; }
        if (e->mode == XCK_MODE_BOTH) { if (k == 0) out->count = r.count; else { out->ad = r.ad; out->dp = r.dp; out->oth = r.oth; } }
        else *out = r;
    }
    e->impl = e->impls[0];
    return XCK_OK;
}

int xck_get_result_device(xck_engine* e, xck_result* out) {
    if (!e || !out) return XCK_E_ARG;
    memset(out, 0, sizeof *out);
    if (e->n_impl == 0) { e->impl = nullptr; return engine_result_device(e, out); }
    for (int k = 0; k < e->n_impl; k++) {
        xck_result r;
        e->impl = e->impls[k];
        int rc = engine_result_device(e, &r);
        if (rc) { e->impl = e->impls[0]; return rc; }
        if (e->mode == XCK_MODE_BOTH) { if (k == 0) out->count = r.count; else { out->ad = r.ad; out->dp = r.dp; out->oth = r.oth; } }
        else *out = r;
    }
    e->impl = e->impls[0];
    return XCK_OK;
}

int xck_get_stats(const xck_engine* ce, xck_stats* out) {
    if (!ce || !out) return XCK_E_ARG;
    xck_engine* e = const_cast<xck_engine*>(ce);
    if (e->n_impl == 0) return XCK_E_STATE;
    memset(out, 0, sizeof *out);
    for (int k = 0; k < e->n_impl; k++) {
        xck_stats s; e->impl = e->impls[k];
        int rc = engine_stats(e, &s);
        if (rc) { e->impl = e->impls[0]; return rc; }
        if (k == 0) *out = s;
        else { out->n_hits += s.n_hits; out->n_hits_unique += s.n_hits_unique; out->ms_h2d += s.ms_h2d; out->ms_device += s.ms_device;
               out->ms_join += s.ms_join; out->ms_sort += s.ms_sort; out->ms_d2h += s.ms_d2h; out->algo_bytes_join += s.algo_bytes_join; out->n_join_launches += s.n_join_launches; }
    }
    e->impl = e->impls[0];
    return XCK_OK;
}

// ---- MatrixMarket text (merge_mtx(), rdr/fc/utils.py:54-93: byte-identical) ----
// A 500 M-read run writes ~10^8 lines: the entries are formatted by a few threads, a wave of 1 M-entry chunks at a time; the
// chunks' file offsets follow from their sizes, and the same threads copy their chunks into the file with pwrite.
namespace {
const char* const kDigits2 =
    "0001020304050607080910111213141516171819202122232425262728293031323334353637383940414243444546474849"
    "5051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
// decimal text of an int32-sized number: two digits per step from a table
inline void put_int(char*& p, int64_t v64) {
    if (v64 < 0) { *p++ = '-'; v64 = -v64; }
    uint64_t v = (uint64_t)v64;
    if (v < 10) { *p++ = char('0' + v); return; }
    if (v < 100) { memcpy(p, kDigits2 + 2 * v, 2); p += 2; return; }
    char t[24]; int q = 24;
    while (v >= 100) { const uint64_t r = v % 100; v /= 100; q -= 2; memcpy(t + q, kDigits2 + 2 * r, 2); }
    if (v >= 10) { q -= 2; memcpy(t + q, kDigits2 + 2 * v, 2); } else t[--q] = char('0' + v);
    memcpy(p, t + q, (size_t)(24 - q)); p += 24 - q;
}
inline int n_digits(int64_t v) { int d = v < 0 ? 2 : 1; if (v < 0) v = -v; while (v >= 10) { v /= 10; d++; } return d; }

struct MtxBody {
    const xck_coo* m; const int32_t* row_map; int64_t n, n_chunks; unsigned nt;
    static constexpr int64_t CH = 1 << 20;
    MtxBody(const xck_coo* m_, const int32_t* rm) : m(m_), row_map(rm), n(m_->nnz), n_chunks((m_->nnz + CH - 1) / CH) {
        nt = (unsigned)xck::default_threads(); if (nt > 16) nt = 16;
        if (const char* e = getenv("XCK_WRITE_THREADS")) nt = (unsigned)std::max(1, atoi(e));
        if ((int64_t)nt > n_chunks) nt = (unsigned)std::max<int64_t>(n_chunks, 1);
    }
    void for_chunks(int64_t c0, int64_t c1, const std::function<void(int64_t)>& fn) const {
        if (nt <= 1 || c1 - c0 <= 1) { for (int64_t c = c0; c < c1; c++) fn(c); return; }
        std::atomic<int64_t> next(c0);
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back([&]() { for (int64_t c; (c = next.fetch_add(1)) < c1;) fn(c); });
        for (auto& x : th) x.join();
    }
    // lines (entries whose region is written) and, if asked, the bytes of their text
    void measure(int64_t* lines, int64_t* bytes) const {
        std::vector<int64_t> kl((size_t)n_chunks, 0), kb((size_t)n_chunks, 0);
        for_chunks(0, n_chunks, [&](int64_t c) { int64_t k = 0, by = 0; const int64_t e = std::min(n, (c + 1) * CH);
            for (int64_t i = c * CH; i < e; i++) { const int32_t r = row_map[m->row[i]]; if (r <= 0) continue; k++;
                if (bytes) by += n_digits(r) + n_digits((int64_t)m->col[i] + 1) + n_digits(m->val[i]) + 3; }
            kl[(size_t)c] = k; kb[(size_t)c] = by; });
        int64_t L = 0, B = 0; for (int64_t c = 0; c < n_chunks; c++) { L += kl[(size_t)c]; B += kb[(size_t)c]; }
        *lines = L; if (bytes) *bytes = B;
    }
    // "row\tcol\tval\n" lines into fd from byte `file_off` on; returns the offset past the last byte, or -1
    off_t write(int fd, off_t file_off) const {
        auto put_all = [fd](const char* p, size_t len, off_t at) { while (len) { const ssize_t w = pwrite(fd, p, len, at); if (w < 0) { if (errno == EINTR) continue; return false; }
                                                                                  p += w; len -= (size_t)w; at += w; } return true; };
        const int64_t WAVE = std::max<int64_t>(nt * 2, 1);
        std::vector<std::unique_ptr<char[]>> bufs((size_t)WAVE);       // uninitialised, allocated on first use, reused by every wave
        std::vector<size_t> used((size_t)WAVE, 0);
        std::vector<off_t> at((size_t)WAVE, 0);
        std::atomic<bool> io_ok(true);
        for (int64_t w0 = 0; w0 < n_chunks; w0 += WAVE) {
            const int64_t w1 = std::min(n_chunks, w0 + WAVE);
            for_chunks(w0, w1, [&](int64_t c) {
                std::unique_ptr<char[]>& b = bufs[(size_t)(c - w0)];
                if (!b) b.reset(new char[(size_t)CH * 36 + 64]);       // 3 ints of <= 11 characters + 3 separators per line (+ slack for the 16-byte row copy)
                char* p = b.get();
                const int64_t e = std::min(n, (c + 1) * CH);
                char rtxt[16] = {0}; int rlen = 0; int32_t r_prev = -1;   // entries are sorted by row: its text is built once per run
                for (int64_t i = c * CH; i < e; i++) {
                    const int32_t r = row_map[m->row[i]];
                    if (r <= 0) continue;
                    if (r != r_prev) { char* q = rtxt; put_int(q, r); *q++ = '\t'; rlen = (int)(q - rtxt); r_prev = r; }
                    memcpy(p, rtxt, 16); p += rlen;                    // (16-byte copy, rlen <= 12 of it kept)
                    put_int(p, (int64_t)m->col[i] + 1); *p++ = '\t'; put_int(p, m->val[i]); *p++ = '\n';
                }
                used[(size_t)(c - w0)] = (size_t)(p - b.get());
            });
            for (int64_t c = w0; c < w1; c++) { at[(size_t)(c - w0)] = file_off; file_off += (off_t)used[(size_t)(c - w0)]; }
            for_chunks(w0, w1, [&](int64_t c) { const size_t j = (size_t)(c - w0); if (used[j] && !put_all(bufs[j].get(), used[j], at[j])) io_ok = false; });
            if (!io_ok) return (off_t)-1;
        }
        return file_off;
    }
};
}  // namespace

int xck_write_mtx(const char* path, const xck_coo* m, const int32_t* row_map, int32_t n_rows_out, int32_t n_cols) {
    if (!path || !m || !row_map) return XCK_E_ARG;
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) { set_thread_error(std::string("cannot open ") + path); return XCK_E_IO; }
    MtxBody body(m, row_map);
    int64_t nnz = 0; body.measure(&nnz, nullptr);
    char line[96];
    const int k = snprintf(line, sizeof line, "%%%%MatrixMarket matrix coordinate integer general\n%%%%\n%d\t%d\t%lld\n", n_rows_out, n_cols, (long long)nnz);
    if (pwrite(fd, line, (size_t)k, 0) != (ssize_t)k) { close(fd); return XCK_E_IO; }
    const off_t end = body.write(fd, (off_t)k);
    if (end < 0) { close(fd); return XCK_E_IO; }
    if (close(fd) != 0) return XCK_E_IO;
    return XCK_OK;
}

// One process's piece of a file that several processes write together (multi-GPU run on one node: every rank owns rows and
// writes their lines itself; sizes are exchanged first, so every piece knows its offset).
int xck_mtx_part_size(const xck_coo* m, const int32_t* row_map, int64_t* n_bytes, int64_t* n_lines) {
    if (!m || !row_map || !n_bytes || !n_lines) return XCK_E_ARG;
    MtxBody(m, row_map).measure(n_lines, n_bytes);
    return XCK_OK;
}
int xck_write_mtx_part(const char* path, int64_t offset, const xck_coo* m, const int32_t* row_map) {
    if (!path || !m || !row_map || offset < 0) return XCK_E_ARG;
    const int fd = open(path, O_WRONLY | O_CREAT, 0666);             // never truncated: other pieces may be there already
    if (fd < 0) { set_thread_error(std::string("cannot open ") + path); return XCK_E_IO; }
    const off_t end = MtxBody(m, row_map).write(fd, (off_t)offset);
    if (close(fd) != 0 || end < 0) return XCK_E_IO;
    return XCK_OK;
}

}  // extern "C"
