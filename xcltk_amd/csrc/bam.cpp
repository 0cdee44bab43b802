// bam.cpp - host ingest: multithreaded BGZF inflate + BAM record parse into pinned SoA batches.
//
// Replaces pysam.AlignmentFile / fetch() of the reference (xcltk/rdr/fc/core.py:73-76,
// xcltk/utils/sam.py:85-118): instead of one indexed fetch per region / SNP (every read
// overlapping k features is decompressed k times) the file is decoded ONCE,
// in file order, and handed to the GPU as structure-of-arrays batches (include/xck.h xck_batch).
//
// Written from the SAM/BAM specification (SAMv1 section 4); htslib is not available here.
// Pipeline per chunk of BGZF blocks:  inflate (parallel over blocks) -> record walk (serial,
// touches 24 B per record) -> field/tag parse into SoA (parallel over records).  The inflate of
// chunk i+1 runs while chunk i is walked and parsed.
#include <zlib.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstring>
#include <deque>
#include <functional>
#include <new>
#include <stdexcept>
#include <thread>
#include "xck_internal.h"
#include "inflate_fast.h"
#include "inflate_dev.h"

namespace xck {

// ---------------------------------------------------------------------------------------------
// small utilities
// ---------------------------------------------------------------------------------------------
static thread_local std::string t_err;
void set_thread_error(const std::string& s) { t_err = s; }
const char* get_thread_error() { return t_err.c_str(); }

int default_threads() {
    static const int cached = [] {
        long n = (long)std::thread::hardware_concurrency(); if (n <= 0) n = 4;
        cpu_set_t set; CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0 && c < n) n = c; }
        long long quota = -1, period = 100000;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                                  // cgroup v2: "<quota|max> <period>"
            char q[32] = {0}; if (fscanf(f, "%31s %lld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atoll(q); fclose(f);
        } else if (FILE* f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {              // cgroup v1
            if (fscanf(f1, "%lld", &quota) != 1) quota = -1; fclose(f1);
            if (FILE* f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f2, "%lld", &period) != 1) period = 100000; fclose(f2); }
        }
        // Behind a CPU-time quota smaller than the CPUs the scheduler may use (a GPU box: 256 hardware threads, 16 CPUs' worth of
        // time), 1.5 threads per quota CPU keep the quota spent while the coordinator, the scanner and the GPU runtime's own
        // threads take their turns: fused ingest 36 -> 40-42 M reads/s at 24 threads, no further gain at 32
        // (profiles/r02_b_ingest_thread_scaling_100m.log, r02_d_ingest_thread_scaling_500m.log).
        if (quota > 0 && period > 0) { const long c = (long)((quota + period - 1) / period); if (c > 0 && c < n) n = std::min(n, c + c / 2); }
        if (const char* e = getenv("XCK_THREADS")) { const int v = atoi(e); if (v > 0) n = v; }
        return (int)std::max(1l, n);
    }();
    return cached;
}

uint64_t hash_bytes(const char* s, size_t n) {                       // 8 bytes per step (read names are 20-40 characters)
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xff51afd7ed558ccdull);
    while (n >= 8) { uint64_t w; memcpy(&w, s, 8); h = (h ^ w) * 0xbf58476d1ce4e5b9ull; h ^= h >> 32; s += 8; n -= 8; }
    if (n) { uint64_t w = 0; memcpy(&w, s, n); h = (h ^ w) * 0x94d049bb133111ebull; }
    h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
    return h;
}

void InternTable::Shard::grow() {
    const size_t ncap = slots.empty() ? 1024 : slots.size() * 2;
    std::vector<Slot> ns(ncap, Slot{0, 0, 0, 0, 0});
    for (const Slot& sl : slots) if (sl.epoch == epoch) {
        size_t i = (size_t)(sl.h >> 6) & (ncap - 1);
        while (ns[i].epoch == epoch) i = (i + 1) & (ncap - 1);
        ns[i] = sl;
    }
    slots.swap(ns);
}

// test hook: XCK_TEST_INTERN_LIMIT=<n> makes the id space of 64-bit keys overflow after n ids (tests/test_gpu_frontends.py
// exercises the front-ends' retry with 128-bit keys without 2^25 distinct read names)
static const uint64_t g_test_intern_limit = [] { const char* e = getenv("XCK_TEST_INTERN_LIMIT"); return e ? (uint64_t)strtoull(e, nullptr, 10) : ~0ull; }();
static inline uint64_t intern_id_limit(int umi_bits) {
    return umi_bits >= 64 ? (1ull << 63) - 1 : std::min<uint64_t>((1ull << (umi_bits - 1)) - 1, g_test_intern_limit);
}

uint64_t InternTable::intern(const char* s, size_t n) {
    const uint64_t h = hash_bytes(s, n);
    Shard& sh = shards_[h & (NSHARD - 1)];
    std::lock_guard<std::mutex> lk(sh.mu);
    if ((size_t)(sh.count + 1) * 2 > sh.slots.size()) sh.grow();
    const size_t mask = sh.slots.size() - 1;
    size_t i = (size_t)(h >> 6) & mask;
    for (;;) {
        Slot& sl = sh.slots[i];
        if (sl.epoch != sh.epoch) {
            sl.h = h; sl.off = (uint32_t)sh.arena.size(); sl.len1 = (uint32_t)n + 1; sl.idx = sh.count++; sl.epoch = sh.epoch;
            sh.arena.insert(sh.arena.end(), s, s + n);
            return ((uint64_t)sl.idx << 6) | (h & (NSHARD - 1));
        }
        if (sl.h == h && sl.len1 == (uint32_t)n + 1 && memcmp(sh.arena.data() + sl.off, s, n) == 0) return ((uint64_t)sl.idx << 6) | (h & (NSHARD - 1));
        i = (i + 1) & mask;
    }
}
uint64_t InternTable::hash(const char* s, size_t n) { return hash_bytes(s, n); }

// UMI-less mode interns every read name: per string the shared table costs a lock and two cache misses (the table of a
// 2 M-read BAM is ~100 MB).  A parse task therefore collects its names and hands them over here, sorted by shard: one
// lock per shard, and the probes of a shard (1/64 of the table) hit the cache.
void InternTable::intern_batch(std::vector<Item>& items, bool* overflow) {
    std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return (a.h & (NSHARD - 1)) < (b.h & (NSHARD - 1)); });
    size_t i = 0;
    while (i < items.size()) {
        const unsigned shard = (unsigned)(items[i].h & (NSHARD - 1));
        Shard& sh = shards_[shard];
        std::lock_guard<std::mutex> lk(sh.mu);
        for (; i < items.size() && (items[i].h & (NSHARD - 1)) == shard; i++) {
            const Item& it = items[i];
            if ((size_t)(sh.count + 1) * 2 > sh.slots.size()) sh.grow();
            const size_t mask = sh.slots.size() - 1;
            if (i + 4 < items.size()) __builtin_prefetch(&sh.slots[(size_t)(items[i + 4].h >> 6) & mask]);
            size_t k = (size_t)(it.h >> 6) & mask;
            uint64_t id;
            for (;;) {
                Slot& sl = sh.slots[k];
                if (sl.epoch != sh.epoch) {
                    sl.h = it.h; sl.off = (uint32_t)sh.arena.size(); sl.len1 = it.n + 1; sl.idx = sh.count++; sl.epoch = sh.epoch;
                    sh.arena.insert(sh.arena.end(), it.s, it.s + it.n);
                    id = ((uint64_t)sl.idx << 6) | shard; break;
                }
                if (sl.h == it.h && sl.len1 == it.n + 1 && memcmp(sh.arena.data() + sl.off, it.s, it.n) == 0) { id = ((uint64_t)sl.idx << 6) | shard; break; }
                k = (k + 1) & mask;
            }
            const uint64_t lim = intern_id_limit(it.umi_bits);
            if (id >= lim) { if (overflow) *overflow = true; *it.out = XCK_UMI_NONE; }
            else *it.out = (1ull << (it.umi_bits - 1)) | id;
        }
    }
}
uint64_t InternTable::size() const { uint64_t n = 0; for (auto& s : shards_) n += s.count; return n; }
void InternTable::clear() {
    for (auto& s : shards_) {
        std::lock_guard<std::mutex> lk(s.mu);
        s.arena.clear(); s.count = 0;
        if (++s.epoch == 0) { std::fill(s.slots.begin(), s.slots.end(), Slot{0, 0, 0, 0, 0}); s.epoch = 1; }   // slots of older epochs read as empty
    }
}

// the part of encode_key() that needs no table: empty string -> NONE, short ACGT string -> 2-bit code
static bool encode_key_direct(const char* s, size_t n, int umi_bits, uint64_t* out) {
    if (n == 0) { *out = XCK_UMI_NONE; return true; }
    if (2 * (int)n + 1 <= umi_bits - 1) {
        // A C G T -> 0 1 2 3, anything else 4: one table look-up per character, no branch inside the loop
        static const struct Lut { uint8_t t[256]; Lut() { memset(t, 4, sizeof t); t[(uint8_t)'A'] = 0; t[(uint8_t)'C'] = 1; t[(uint8_t)'G'] = 2; t[(uint8_t)'T'] = 3; } } lut;
        uint64_t v = 1; unsigned bad = 0;
        for (size_t i = 0; i < n; i++) { const unsigned c = lut.t[(uint8_t)s[i]]; bad |= c; v = (v << 2) | (c & 3u); }
        if (!(bad & 4u)) { *out = v; return true; }
    }
    return false;
}

uint64_t encode_key(const char* s, size_t n, int umi_bits, InternTable& tab, bool* overflow) {
    uint64_t direct;
    if (encode_key_direct(s, n, umi_bits, &direct)) return direct;
    uint64_t id = tab.intern(s, n);
    const uint64_t lim = intern_id_limit(umi_bits);
    if (id >= lim) { if (overflow) *overflow = true; return XCK_UMI_NONE; }
    return (1ull << (umi_bits - 1)) | id;
}

// Barcode -> column: open addressing over slots that hold the full 64-bit hash, the column and the barcode's own bytes (up to
// 18, the length of a 10x barcode), so a hit costs ONE cache line and still compares the exact string; longer barcodes compare against
// the string list.  (Slots -> std::string -> heap was three dependent misses per read.)
void DecodeCfg::build_barcodes(const char* const* names, int n) {
    barcodes.clear();
    for (int i = 0; i < n; i++) barcodes.emplace_back(names[i]);
    size_t cap = 16; while (cap < (size_t)n * 2) cap <<= 1;
    BcSlot empty; memset(&empty, 0, sizeof empty); empty.idx = -1;
    bc_slots.assign(cap, empty); bc_mask = cap - 1;
    for (int i = 0; i < n; i++) {
        const std::string& b = barcodes[i];
        const uint64_t hh = hash_bytes(b.data(), b.size());
        uint64_t h = hh & bc_mask;
        while (bc_slots[h].idx >= 0) h = (h + 1) & bc_mask;
        BcSlot& sl = bc_slots[h];
        sl.h = hh; sl.idx = i; sl.len = (uint16_t)std::min<size_t>(b.size(), 0xFFFF);
        memcpy(sl.key, b.data(), std::min<size_t>(b.size(), sizeof sl.key));
    }
}
int32_t DecodeCfg::lookup_cell(const char* s, size_t n) const {
    if (bc_slots.empty()) return -1;
    const uint64_t hh = hash_bytes(s, n);
    uint64_t h = hh & bc_mask;
    while (bc_slots[h].idx >= 0) {
        const BcSlot& sl = bc_slots[h];
        if (sl.h == hh && sl.len == (n > 0xFFFF ? 0xFFFF : (uint16_t)n)) {
            if (n <= sizeof sl.key) { if (memcmp(sl.key, s, n) == 0) return sl.idx; }
            else { const std::string& b = barcodes[sl.idx]; if (b.size() == n && memcmp(b.data(), s, n) == 0) return sl.idx; }
        }
        h = (h + 1) & bc_mask;
    }
    return -1;
}

// ---------------------------------------------------------------------------------------------
// thread pool
// ---------------------------------------------------------------------------------------------
class Pool {
public:
    explicit Pool(int n) { for (int i = 0; i < n; i++) th_.emplace_back([this] { run(); }); }
    ~Pool() { { std::lock_guard<std::mutex> lk(mu_); stop_ = true; } cv_.notify_all(); for (auto& t : th_) t.join(); }
    void submit(std::function<void()> f, bool front = false) { { std::lock_guard<std::mutex> lk(mu_); if (front) q_.push_front(std::move(f)); else q_.push_back(std::move(f)); } cv_.notify_one(); }
    // many tasks under ONE lock and one wake-up: a chunk hands ~100 part tasks to the pool three times (inflate / gather, walk, parse), and a
    // lock + futex wake per task cost the coordinator ~0.7 ms per chunk
    void submit_many(std::vector<std::function<void()>>& fs, bool front = false) {
        if (fs.empty()) return;
        { std::lock_guard<std::mutex> lk(mu_);
          if (front) for (size_t i = fs.size(); i-- > 0;) q_.push_front(std::move(fs[i])); else for (auto& f : fs) q_.push_back(std::move(f)); }
        if (fs.size() == 1) cv_.notify_one(); else cv_.notify_all();
        fs.clear();
    }
    int size() const { return (int)th_.size(); }
    void set_affinity(const cpu_set_t& set) { for (auto& t : th_) pthread_setaffinity_np(t.native_handle(), sizeof set, &set); }
private:
    void run() {
        for (;;) {
            std::function<void()> f;
            { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
              if (q_.empty()) return; f = std::move(q_.front()); q_.pop_front(); }
            f();
        }
    }
    std::vector<std::thread> th_; std::deque<std::function<void()>> q_; std::mutex mu_; std::condition_variable cv_; bool stop_ = false;
};

struct TaskGroup {
    std::mutex mu; std::condition_variable cv; int pending = 0;
    std::atomic<int> thrown{0};          // a task body threw: 1 = std::bad_alloc, 2 = anything else (checked by the waiter: an exception
                                         // that leaves a pool thread would otherwise end the process through std::terminate)
    int take_thrown() { return thrown.exchange(0); }
    void add(Pool& p, std::function<void()> f, bool front = false) {     // front: ahead of what is queued (work the coordinator is waiting for)
        { std::lock_guard<std::mutex> lk(mu); pending++; }
        // The notify happens UNDER the lock: a TaskGroup on the waiter's stack may be destroyed as soon as wait() returns, and
        // wait() cannot return before this task has released the mutex - after which it touches the group no more.  (Notifying
        // after the unlock let a waiter that saw pending == 0 leave first; the late pthread_cond_broadcast then wrote into
        // whatever the coordinator's stack held by then: "free(): invalid pointer", once per ~14 k chunks with more pool threads
        // than CPUs, tools/stress_e2e.py.)
        p.submit([this, f] {
            try { f(); } catch (const std::bad_alloc&) { thrown.store(1); } catch (...) { thrown.store(2); }
            std::lock_guard<std::mutex> lk(mu); if (--pending == 0) cv.notify_all(); }, front);
    }
    // collect tasks with later(), hand them to the pool in one go with flush()
    std::vector<std::function<void()>> batch;
    void later(std::function<void()> f) {
        batch.push_back([this, f] {
            try { f(); } catch (const std::bad_alloc&) { thrown.store(1); } catch (...) { thrown.store(2); }
            std::lock_guard<std::mutex> lk(mu); if (--pending == 0) cv.notify_all(); });
    }
    void flush(Pool& p, bool front = false) {
        if (batch.empty()) return;
        { std::lock_guard<std::mutex> lk(mu); pending += (int)batch.size(); }
        p.submit_many(batch, front);
    }
    void wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this] { return pending == 0; }); }
};

// ---------------------------------------------------------------------------------------------
// BGZF
// ---------------------------------------------------------------------------------------------
static inline uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

struct BlockRef { uint64_t coff; uint32_t clen; uint32_t isize; uint32_t data_off; uint32_t data_len; uint64_t uoff; };

// parse one BGZF block header at file offset `coff`; returns false at EOF / on malformed data
static bool bgzf_peek(const uint8_t* map, uint64_t fsize, uint64_t coff, BlockRef* out, std::string* err) {
    if (coff + 18 > fsize) { if (coff != fsize && err) *err = "truncated BGZF block header"; return false; }
    const uint8_t* p = map + coff;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) { if (err) *err = "not a BGZF block (bad gzip magic)"; return false; }
    uint32_t xlen = le16(p + 10);
    if (coff + 12 + xlen > fsize) { if (err) *err = "truncated BGZF extra field"; return false; }
    const uint8_t* x = p + 12; const uint8_t* xe = x + xlen;
    int64_t bsize = -1;
    while (x + 4 <= xe) { uint32_t sl = le16(x + 2); if (x + 4 + sl > xe) break;          // subfield payload must lie inside the extra field
                          if (x[0] == 'B' && x[1] == 'C' && sl == 2) { bsize = le16(x + 4); } x += 4 + sl; }
    if (bsize < 0) { if (err) *err = "BGZF block without BC subfield"; return false; }
    uint32_t total = (uint32_t)bsize + 1;
    if (coff + total > fsize || total < 12 + xlen + 8) { if (err) *err = "truncated BGZF block"; return false; }
    out->coff = coff; out->clen = total; out->data_off = 12 + xlen; out->data_len = total - (12 + xlen) - 8;
    out->isize = le32(p + total - 4); out->uoff = 0;
    if (out->isize > 65536) { if (err) *err = "BGZF block claims more than 64 KiB of data"; return false; }   // SAMv1 4.1: ISIZE <= 65536
    return true;
}

static std::atomic<long> g_inflate_fallbacks{0};
static const bool g_use_fast_inflate = [] { const char* e = getenv("XCK_INFLATE"); return !(e && strcmp(e, "zlib") == 0); }();
static thread_local InflateTables t_inflate_tables;

static bool bgzf_inflate(const uint8_t* map, const BlockRef& b, uint8_t* dst, bool verify_crc, z_stream* zs, std::string* err) {
    if (b.isize == 0) return true;
    bool done = false;
    if (g_use_fast_inflate) {                                  // own whole-buffer decoder (inflate_fast.h); zlib on any irregularity
        done = inflate_raw(map + b.coff + b.data_off, b.data_len, dst, b.isize, &t_inflate_tables) == 0;
        if (!done) g_inflate_fallbacks.fetch_add(1, std::memory_order_relaxed);
    }
    if (!done) {
        inflateReset(zs);
        zs->next_in = const_cast<Bytef*>(map + b.coff + b.data_off); zs->avail_in = b.data_len;
        zs->next_out = dst; zs->avail_out = b.isize;
        int rc = inflate(zs, Z_FINISH);
        if (rc != Z_STREAM_END || zs->avail_out != 0) { if (err) *err = "BGZF inflate failed"; return false; }
    }
    if (verify_crc) {
        uint32_t want = le32(map + b.coff + b.clen - 8);
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, b.isize) != want) { if (err) *err = "BGZF CRC mismatch"; return false; }
    }
    return true;
}

struct ZStream { z_stream zs; bool ok; ZStream() { memset(&zs, 0, sizeof zs); ok = inflateInit2(&zs, -15) == Z_OK; } ~ZStream() { if (ok) inflateEnd(&zs); } };
static thread_local ZStream t_zs;

}  // namespace xck

using namespace xck;

// ---------------------------------------------------------------------------------------------
// the BAM file object
// ---------------------------------------------------------------------------------------------
// One decoded chunk as structure-of-arrays, all columns carved out of ONE pinned block: the whole chunk crosses PCIe in a
// single hipMemcpyAsync (engine_push_block) and the batches of the chunk are slices of it.  Three blocks rotate, so the
// decoder fills block i + 1 while block i is still being copied; `fence` (an event owned by this block, recorded by the
// engine behind the copy) is waited for before the block is overwritten.
struct HostSoA {
    uint8_t* base = nullptr; size_t cap = 0, used = 0; bool pinned = false; void* fence = nullptr;
    int32_t* pos = nullptr; uint16_t* flag = nullptr; uint8_t* mapq = nullptr; int32_t* cell = nullptr; uint64_t* umi = nullptr;
    uint32_t* cig_off = nullptr; uint32_t* cigar = nullptr; uint32_t* seq_off = nullptr; uint8_t* seq = nullptr;
};

struct RecRef { const uint8_t* p; uint32_t len; int32_t tid; uint32_t l_seq; uint16_t n_cig; };

// Records found by one inflate task in its own byte range, walked right after inflating (data
// still hot in that core's cache) under the SPECULATION that a record starts at the first byte of
// the range - true for BGZF writers that flush a block before a record that would not fit (htslib,
// samtools, this repo's writers).  The task also sums up what its records need in the SoA (kept records, CIGAR words,
// sequence bytes) and notes where the contig changes, so that when every speculation of a chunk holds the output layout
// is a prefix sum over the parts and each part is parsed by its own task; otherwise the stitch re-walks serially.
struct ContigRun { uint32_t first; int32_t contig; };                    // records [first, next run's first) of the part are on `contig` (-1: unused)
struct WalkPart {
    size_t u_begin = 0, u_end = 0, spec_start = 0, stop = 0; std::vector<RecRef> recs;
    size_t n_out = 0, n_cig = 0, n_seq = 0; std::vector<ContigRun> runs;
    size_t out_base = 0, cig_base = 0, seq_base = 0;                      // filled by the coordinator (fast path)
    size_t i0 = 0, i1 = 0;                                                // the part's blocks: [i0, i1) of the chunk
};

struct Chunk {
    std::vector<BlockRef> blocks;
    std::vector<uint8_t> ubuf; size_t usize = 0;
    uint8_t* ubase = nullptr;          // where the chunk's inflated bytes live: ubuf, or the pinned output block of its GPU slot
    std::vector<WalkPart> parts;
    TaskGroup tg; std::atomic<bool> failed{false}; std::string err; std::mutex emu;
    bool valid = false, new_range = false; uint32_t first_skip = 0; int range_id = 0;
    // GPU share of the inflate (csrc/inflate_dev.hip): the pool only gathers the compressed bytes into the slot's pinned block, the task
    // that finishes last enqueues copy-in, kernel and copy-out; walk (and the blocks the kernel left) follow when the chunk is consumed
    bool gpu = false; std::atomic<int> copy_left{0}; std::atomic<int> gpu_rc{0}; size_t in_total = 0;
    std::atomic<int> launched{0};      // the last gather task has handed the chunk to the device (gpu_rc says how that went)
    bool stage2 = false;               // the walk tasks that follow the device's inflate have been queued
};

struct PendingBatch { int32_t contig; int64_t r0, r1; uint64_t ordinal_base; };

// BAM tid (+ record position) -> engine contig: -1 for references that are not wanted and for records that start at or
// beyond the end of the position window of their reference (xck_ingest_opts.tid_end)
struct ContigMap {
    const int32_t* t2c = nullptr; const int32_t* t_end = nullptr; int n_refs = 0;
    int32_t only_tid = -1;               // position windows: the chunk belongs to the index range of ONE reference; the tail block of that
                                         // range also holds the first records of the next reference, which that reference's own range
                                         // delivers again - they are dropped here, so no record reaches the consumer twice
    inline int32_t operator()(int32_t tid, int32_t pos) const {
        if (tid < 0 || tid >= n_refs || !t2c) return -1;
        if (only_tid >= 0 && tid != only_tid) return -1;
        if (t_end && t_end[tid] > 0 && pos >= t_end[tid]) return -1;
        return t2c[tid];
    }
};

// where the ingest time goes (XCK_DEBUG_TIMING=1 prints it when the reader is closed): pool-side CPU time summed over
// threads, and the phases of the coordinating thread
struct DecodeTimes {
    std::atomic<uint64_t> inflate{0}, walk{0}, parse{0};                 // ns, summed over pool threads
    uint64_t wait_inflate = 0, sched = 0, stitch = 0, layout = 0, wait_parse = 0, wait_push = 0;   // ns, coordinating thread
    uint64_t push = 0;                                                   // ns, push thread (engine_push_block)
    uint64_t chunks = 0, slow_chunks = 0; std::chrono::steady_clock::time_point t_open;
};
static inline uint64_t ns_since(std::chrono::steady_clock::time_point t0) { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }

// The BGZF block headers form a chain (every header gives the offset of the next one): a scanner thread walks it ahead of
// the decoder - it also takes the page faults of the mapping - and hands over one plan per chunk.
struct ChunkPlan { std::vector<BlockRef> blocks; size_t usize = 0; uint32_t first_skip = 0; int range_id = 0; bool new_range = false, failed = false, end = false; std::string err; };
struct ScanRange { uint64_t coff; uint32_t skip; uint64_t stop; };
class Scanner {
public:
    Scanner(const uint8_t* map, uint64_t fsize, size_t chunk_target, std::vector<ScanRange> ranges)
        : map_(map), fsize_(fsize), target_(chunk_target), ranges_(std::move(ranges)) { th_ = std::thread([this] { run(); }); }
    ~Scanner() { { std::lock_guard<std::mutex> lk(mu_); stop_ = true; } cv_.notify_all(); if (th_.joinable()) th_.join(); }
    void abandon(int range_id) { abandoned_.store(range_id); }          // the decoder has seen everything it wants of that range
    ChunkPlan next() {                                                  // blocks until a plan is ready; `end` after the last chunk
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !q_.empty(); });
        ChunkPlan p = std::move(q_.front()); q_.pop_front();
        lk.unlock(); cv_.notify_all();
        return p;
    }
private:
    bool put(ChunkPlan&& p) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return stop_ || q_.size() < 3; });
        if (stop_) return false;
        q_.push_back(std::move(p)); lk.unlock(); cv_.notify_all();
        return true;
    }
    void run() {
        // (the scanner's only allocations are the block lists: out of memory ends the stream with a failed plan, not the process)
        try { scan(); }
        catch (...) { ChunkPlan f; f.failed = true; f.err = "out of host memory (BGZF scanner)"; f.blocks.clear(); if (put(std::move(f))) { ChunkPlan z; z.end = true; put(std::move(z)); } }
    }
    void scan() {
        for (size_t ri = 0; ri < ranges_.size(); ri++) {
            const ScanRange& rg = ranges_[ri];
            uint64_t coff = rg.coff; bool first = true, ended = false;
            while (!ended) {
                if (abandoned_.load() == (int)ri) break;
                ChunkPlan p; p.new_range = first; p.first_skip = first ? rg.skip : 0; p.range_id = (int)ri; first = false;
                size_t usz = 0;
                while (usz < target_) {
                    if (coff > rg.stop) { ended = true; break; }          // end of the indexed range
                    BlockRef br; std::string e;
                    if (!bgzf_peek(map_, fsize_, coff, &br, &e)) { ended = true; if (!e.empty()) { p.failed = true; p.err = e; } break; }
                    br.uoff = usz; usz += br.isize; coff += br.clen;
                    p.blocks.push_back(br);
                }
                p.usize = usz;
                if (p.blocks.empty() && !p.failed) { if (p.new_range) { /* an empty range: nothing to hand over */ } break; }
                const bool failed = p.failed;
                if (!put(std::move(p))) return;
                if (failed) { ChunkPlan z; z.end = true; put(std::move(z)); return; }
            }
        }
        ChunkPlan z; z.end = true; put(std::move(z));
    }
    const uint8_t* map_; uint64_t fsize_; size_t target_; std::vector<ScanRange> ranges_;
    std::thread th_; std::mutex mu_; std::condition_variable cv_; std::deque<ChunkPlan> q_; bool stop_ = false;
    std::atomic<int> abandoned_{-1};
};

#ifndef XCK_PUSH_Q
#define XCK_PUSH_Q 2
#endif
constexpr int PUSH_Q = XCK_PUSH_Q;      // chunks the coordinator may run ahead of the push thread (Pusher)
constexpr int N_CHUNK = 3 + PUSH_Q, N_SOA = PUSH_Q + 1;
// What follows a chunk's layout - wait for its parse tasks, then push it (engine_push_block: one H2D copy + the join launches, after
// waiting for the previous chunk's launch to be confirmed) - runs on a thread of its own, in file order, up to two chunks behind the
// coordinator, which meanwhile schedules, waits for and lays out the next chunks.  With the GPU share of the inflate the coordinator
// had become the limiter of the ingest (schedule + wait for parse + push = 4.5 ms per chunk, the pool a third idle).
// A job owns: the chunk's SoA block, its ring chunk (inflated bytes + record lists, until `parsed`), its batch list.
struct IngestJob {
    TaskGroup tg; std::atomic<int> flags{0}; bool has_parse = false; int soa = -1, ring_idx = -1; std::deque<PendingBatch> pending;
    std::atomic<int> parsed{0};        // set by the push thread once the parse tasks have ended: the coordinator may take the ring chunk back
    bool released = true;              // (coordinator) the ring chunk was taken back
};
struct Pusher {
    static constexpr int MAXQ = PUSH_Q; // jobs the coordinator may run ahead
    std::thread th; std::mutex mu; std::condition_variable cv;
    bool stop = false; IngestJob* q[MAXQ] = {}; int qh = 0, qn = 0;
    int rc = 0; std::string err;       // first failure (parse flags or push); later jobs are waited for, not pushed
    xck_engine* e = nullptr; xck_bam* b = nullptr;
    uint64_t push_ns = 0, parse_wait_ns = 0;
    void run();                        // (below: needs xck_bam)
    // blocks while MAXQ jobs are outstanding; returns the first failure so far (the job is queued regardless: its parse must be waited for)
    int submit(IngestJob* j) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this] { return qn < MAXQ; }); q[(qh + qn) % MAXQ] = j; qn++; cv.notify_all(); return rc; }
    int wait_idle() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this] { return qn == 0; }); return rc; }
    ~Pusher() { { std::lock_guard<std::mutex> lk(mu); stop = true; cv.notify_all(); } if (th.joinable()) th.join(); }
};
constexpr int N_CHUNK_GPU = 16;        // ring depth with the GPU share on: its chunks need several in flight on the device (one wave per block)
// XCK_GPU_INFLATE=<percent>: that share of the chunks is inflated on the handle's GPU (0 = off).  State of one reader:
struct GpuShare {
    bool tried = false, on = false, broken = false, verbose = false; int pct = 0, acc = 0, device = -1, free_cus = 32, depth = 4;
    GpuInflateSlot* slot[N_CHUNK_GPU] = {nullptr}; bool inflight[N_CHUNK_GPU] = {false};   // slot[k]: the slot that holds ring chunk k (its inflated bytes live there until the chunk is consumed)
    std::vector<GpuInflateSlot*> free_slots;                             // slots of consumed chunks, reused before a new one is made
    int max_held = 10;                                                   // device chunks in the ring at a time (in flight + inflated, not yet consumed) = slots this reader ever holds
    uint64_t chunks = 0, blocks = 0, left_blocks = 0, wait_ns = 0, copy_ns = 0;
};
struct CallerBinding;
struct xck_bam {
    std::string path; int fd = -1; const uint8_t* map = nullptr; uint64_t fsize = 0;
    std::vector<std::string> ref_names; std::vector<int64_t> ref_lens;
    uint64_t next_coff = 0;            // first BGZF block with alignment records
    // .bai: per reference [beg, end) virtual offsets + record counts (samtools pseudo-bin 37450 or bin chunks)
    bool idx_loaded = false, idx_ok = false;
    std::vector<uint64_t> idx_beg, idx_end; std::vector<int64_t> idx_mapped, idx_unmapped; std::vector<std::vector<uint64_t>> idx_lin;
    std::vector<std::pair<uint64_t, uint64_t>> ranges; bool use_ranges = false, ranges_set = false;
    uint32_t first_skip = 0;           // bytes of the first block that precede the first record
    uint32_t stitch_skip = 0;          // same for the chunk being stitched
    Scanner* scanner = nullptr; bool scan_end = false;
    std::vector<int32_t> range_tid;                       // per_tid_ranges: reference of every range
    bool per_tid_ranges = false; int skip_range = -1;   // position windows: one range per reference; a range is dropped once a record starts beyond its window
    Pool* pool = nullptr; int n_threads = 1;
    Chunk ch[N_CHUNK_GPU]; int n_ring = N_CHUNK, head = 0, n_sched = 0;      // ring of n_ring chunks: ch[head] is decoded next, n_sched chunks are inflating / inflated
    GpuShare gi;
    bool started = false;              // a decode call has been made (xck_bam_prefetch alone does not count)
    bool prefetching = false;          // inside xck_bam_prefetch
    Pusher* pusher = nullptr;          // made at the first push of a GPU-backed ingest
    // xck_ingest_bam on a GPU-backed handle (defer_parse): the parse of a chunk (pool tasks: record fields -> SoA block) and its push run
    // behind the coordinator (Pusher).  jobs[k % (PUSH_Q + 1)] belongs to the k-th chunk handed over; at most Pusher::MAXQ are outstanding.
    IngestJob jobs[PUSH_Q + 1]; uint64_t job_n = 0;
    bool defer_parse = false;          // set by xck_ingest_bam for GPU-backed handles
    std::vector<uint8_t> carry;        // partial record from the previous chunk
    std::vector<uint8_t> stitch;       // boundary record assembled from carry + head of this chunk
    std::vector<RecRef> recs; std::vector<int32_t> rec_contig; std::vector<int64_t> rec_out;   // serial walk output (slow path)
    HostSoA soa[N_SOA]; int soa_i = 0;                 // soa[soa_i] holds the chunk decoded last
    std::deque<PendingBatch> pending;
    int64_t n_records = 0;             // records walked so far (all, including unused contigs)
    bool done = false;
    size_t chunk_target = 48u << 20;   // uncompressed bytes per chunk
    // NUMA binding (bind_to_numa_node): the pool and the scanner sit on the node of the handle's GPU while the reader is open; the
    // CALLING thread (it runs the coordinator, and its first touch places the inflate buffers) only for the duration of a decode
    // call (CallerBinding) - between calls, and on whatever thread closes the reader, the caller's own mask is untouched
    bool numa_bound = false, threads_auto = false; cpu_set_t numa_set, old_affinity;
    struct CallerBinding* cur_binding = nullptr;
    DecodeTimes tm;
    std::string err;
};

static void soa_free(HostSoA& s) {
    if (s.fence) { fence_wait(s.fence); fence_destroy(s.fence); }
    if (s.base) { if (s.pinned) pinned_free(s.base); else free(s.base); }
    s = HostSoA();
}
// carve the columns of a chunk with n_reads kept records out of the block (growing it if needed)
static bool soa_reserve(HostSoA& s, size_t n_reads, size_t n_cig, size_t n_seq) {
    auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
    const size_t o_umi = 0, o_pos = o_umi + up(n_reads * 8), o_cell = o_pos + up(n_reads * 4), o_cgo = o_cell + up(n_reads * 4),
                 o_sqo = o_cgo + up((n_reads + 1) * 4), o_cig = o_sqo + up((n_reads + 1) * 4), o_flag = o_cig + up(n_cig * 4),
                 o_mapq = o_flag + up(n_reads * 2), o_seq = o_mapq + up(n_reads), total = o_seq + up(n_seq);
    if (total > s.cap) {
        if (s.fence) fence_wait(s.fence);
        if (s.base) { if (s.pinned) pinned_free(s.base); else free(s.base); }
        const size_t c = total + total / 4;
        s.base = (uint8_t*)pinned_alloc(c); s.pinned = s.base != nullptr;
        if (!s.base) s.base = (uint8_t*)malloc(c);
        if (!s.base) { s.cap = 0; return false; }
        s.cap = c;
    }
    s.used = total;
    s.umi = (uint64_t*)(s.base + o_umi); s.pos = (int32_t*)(s.base + o_pos); s.cell = (int32_t*)(s.base + o_cell);
    s.cig_off = (uint32_t*)(s.base + o_cgo); s.seq_off = (uint32_t*)(s.base + o_sqo); s.cigar = (uint32_t*)(s.base + o_cig);
    s.flag = (uint16_t*)(s.base + o_flag); s.mapq = s.base + o_mapq; s.seq = s.base + o_seq;
    return true;
}

// sequential reader used only for the header
struct SeqReader {
    xck_bam* b; uint64_t coff = 0; std::vector<uint8_t> buf; size_t pos = 0; uint64_t blk_coff = 0; size_t blk_start = 0; std::string err;
    bool fill(size_t need) {
        while (buf.size() - pos < need) {
            BlockRef br;
            if (!bgzf_peek(b->map, b->fsize, coff, &br, &err)) { if (err.empty()) err = "unexpected end of file in BAM header"; return false; }
            size_t old = buf.size();
            buf.resize(old + br.isize);
            if (!t_zs.ok || !bgzf_inflate(b->map, br, buf.data() + old, false, &t_zs.zs, &err)) { if (err.empty()) err = "inflate init failed"; return false; }
            blk_coff = coff; blk_start = old; coff += br.clen;
        }
        return true;
    }
};

// A multi-BAM run (one file per cell, hundreds of files) opens and closes a reader per file.  The big buffers of a reader -
// pinned SoA arrays (hipHostMalloc pins page by page), the two inflate buffers, the record index, the thread pool - cost
// ~30 ms to set up, twice the decode time of a 500 k-read file: closed readers park them here for the next open.
struct BamScratch {
    HostSoA soa[N_SOA]; std::vector<uint8_t> ubuf[N_CHUNK]; std::vector<RecRef> recs; std::vector<int32_t> rec_contig; std::vector<int64_t> rec_out;
    Pool* pool = nullptr; int n_threads = 0;
};
static std::mutex g_scratch_mu;
static std::vector<BamScratch*> g_scratch;                            // at most 4 parked sets; kept until the process ends
// ... and so are the GPU-inflate slots of closed readers: a slot is ~70 MB of pinned + 50 MB of device memory, and pinning it anew
// for every file cost 0.2 s per open (a 20 M-record file decodes in 0.45 s)
static std::vector<GpuInflateSlot*> g_gpu_slots;                      // at most 16; kept until the process ends
static GpuInflateSlot* take_gpu_slot(int device, int free_cus, bool verbose) {
    { std::lock_guard<std::mutex> lk(g_scratch_mu);
      for (size_t i = 0; i < g_gpu_slots.size(); i++) if (g_gpu_slots[i]->device == device) { GpuInflateSlot* s = g_gpu_slots[i]; g_gpu_slots.erase(g_gpu_slots.begin() + (long)i); return s; } }
    return gpu_inflate_slot_create(device, free_cus, verbose);
}
// (parked slots are destroyed at exit, BEFORE the HIP runtime's own teardown - this handler is registered after the runtime came up, so
// it runs first: streams with a CU mask and mapped host blocks left alive made the process crash at exit under rocprofv3)
static void destroy_parked_gpu_slots() {
    std::vector<GpuInflateSlot*> v;
    { std::lock_guard<std::mutex> lk(g_scratch_mu); v.swap(g_gpu_slots); }
    for (GpuInflateSlot* s : v) gpu_inflate_slot_destroy(s);
}
static void park_gpu_slot(GpuInflateSlot* s) {
    if (!s) return;
    gpu_inflate_slot_wait(s);                                          // nothing in flight on a parked slot
    { std::lock_guard<std::mutex> lk(g_scratch_mu);
      static bool at_exit = false;
      if (!at_exit) { at_exit = true; std::atexit(destroy_parked_gpu_slots); }
      if (g_gpu_slots.size() < 16) { g_gpu_slots.push_back(s); return; } }
    gpu_inflate_slot_destroy(s);
}

extern "C" {

static int bam_open_impl(const char* path, int n_threads, xck_bam** out, char* err, size_t errlen) {
    auto fail = [&](const std::string& m, xck_bam* b) { if (err && errlen) { snprintf(err, errlen, "%s: %s", path ? path : "(null)", m.c_str()); } set_thread_error(m); if (b) xck_bam_close(b); return XCK_E_IO; };
    if (!path || !out) return fail("null argument", nullptr);
    xck_bam* b = new xck_bam();
    b->path = path;
    b->fd = open(path, O_RDONLY);
    if (b->fd < 0) return fail("cannot open file", b);
    struct stat st; if (fstat(b->fd, &st) != 0 || st.st_size < 28) return fail("cannot stat file / file too small", b);
    b->fsize = (uint64_t)st.st_size;
    void* m = mmap(nullptr, b->fsize, PROT_READ, MAP_PRIVATE, b->fd, 0);
    if (m == MAP_FAILED) return fail("mmap failed", b);
    b->map = (const uint8_t*)m;
    madvise(m, b->fsize, MADV_SEQUENTIAL);
    // the reference lets pysam detect SAM / BAM / CRAM; this decoder reads BAM only and says so up front
    if (memcmp(b->map, "CRAM", 4) == 0) return fail("CRAM input is not supported: this decoder reads BAM only (samtools view -b)", b);
    if (b->map[0] == '@' && b->map[3] == '\t') return fail("SAM text input is not supported: this decoder reads BAM only (samtools view -b)", b);
    SeqReader r; r.b = b;
    if (!r.fill(12)) return fail(r.err, b);
    if (memcmp(r.buf.data(), "BAM\1", 4) != 0) return fail("not a BAM file (bad magic)", b);
    uint32_t l_text = le32(r.buf.data() + 4);
    if (!r.fill(12 + (size_t)l_text)) return fail(r.err, b);
    r.pos = 8 + l_text;
    uint32_t n_ref = le32(r.buf.data() + r.pos); r.pos += 4;
    for (uint32_t i = 0; i < n_ref; i++) {
        if (!r.fill(4)) return fail(r.err, b);
        uint32_t l_name = le32(r.buf.data() + r.pos); r.pos += 4;
        if (!r.fill((size_t)l_name + 4)) return fail(r.err, b);
        b->ref_names.emplace_back((const char*)r.buf.data() + r.pos, l_name ? l_name - 1 : 0); r.pos += l_name;
        b->ref_lens.push_back((int64_t)le32(r.buf.data() + r.pos)); r.pos += 4;
    }
    // first alignment record: inside the last inflated block at offset (r.pos - r.blk_start), or at the next block
    if (r.pos == r.buf.size()) { b->next_coff = r.coff; b->first_skip = 0; }
    else { b->next_coff = r.blk_coff; b->first_skip = (uint32_t)(r.pos - r.blk_start); }
    b->threads_auto = n_threads <= 0;
    if (n_threads <= 0) n_threads = default_threads();
    b->n_threads = n_threads;
    b->tm.t_open = std::chrono::steady_clock::now();
    { BamScratch* sc = nullptr;
      { std::lock_guard<std::mutex> lk(g_scratch_mu); if (!g_scratch.empty()) { sc = g_scratch.back(); g_scratch.pop_back(); } }
      if (sc) {
          for (int i = 0; i < N_SOA; i++) b->soa[i] = sc->soa[i];
          for (int i = 0; i < N_CHUNK; i++) b->ch[i].ubuf.swap(sc->ubuf[i]);
          b->recs.swap(sc->recs); b->rec_contig.swap(sc->rec_contig); b->rec_out.swap(sc->rec_out);
          if (sc->pool && sc->n_threads == n_threads) b->pool = sc->pool; else delete sc->pool;
          delete sc;
      } }
    if (!b->pool) b->pool = new Pool(n_threads);
    if (const char* cb = getenv("XCK_CHUNK_BYTES")) { long long v = atoll(cb); if (v >= 1024) b->chunk_target = (size_t)v; }   // tests: force many chunks
    *out = b;
    return XCK_OK;
}

void xck_bam_close(xck_bam* b) {
    if (!b) return;
    if (b->pusher) { b->pusher->wait_idle(); b->tm.push += b->pusher->push_ns; b->tm.wait_parse += b->pusher->parse_wait_ns; delete b->pusher; b->pusher = nullptr; }
    for (auto& j : b->jobs) j.tg.wait();
    delete b->scanner; b->scanner = nullptr;
    for (auto& c : b->ch) c.tg.wait();
    // no H2D copy may still read a block that is parked or freed - and a parked block must not keep its event: the copy stream
    // it was recorded on dies with the engine's staging, and waiting on such an event later fails (hipErrorStreamCaptureUnsupported
    // on ROCm 7: the next reader's first launch check then reported that stale error; tests/test_gpu_random_e2e.py seed 9)
    for (auto& so : b->soa) if (so.fence) { fence_wait(so.fence); fence_destroy(so.fence); so.fence = nullptr; }
    if (getenv("XCK_DEBUG_TIMING") && b->gi.chunks)
        fprintf(stderr, "[xck] ingest %s: GPU share of the inflate: %llu chunks / %llu blocks on the device (%llu of them finished by the host), gather %.0f ms of pool time, coordinator waited %.0f ms%s\n",
                b->path.c_str(), (unsigned long long)b->gi.chunks, (unsigned long long)b->gi.blocks, (unsigned long long)b->gi.left_blocks, b->gi.copy_ns * 1e-6, b->gi.wait_ns * 1e-6,
                b->gi.broken ? "; the device path was given up (runtime error / no memory)" : "");
    for (auto& gs : b->gi.slot) { park_gpu_slot(gs); gs = nullptr; }
    for (auto& gs : b->gi.free_slots) park_gpu_slot(gs);
    b->gi.free_slots.clear();
    if (b->numa_bound && b->pool) b->pool->set_affinity(b->old_affinity);   // a parked pool goes back to the full mask
    if (getenv("XCK_DEBUG_TIMING") && b->tm.chunks) {
        const DecodeTimes& t = b->tm; const double ms = 1e-6;
        fprintf(stderr, "[xck] ingest %s: %lld records, %llu chunks (%llu stitched serially), %.0f ms since open, %d threads | pool CPU ms: inflate %.0f walk %.0f parse %.0f | "
                        "coordinator ms: wait_inflate %.0f schedule %.0f stitch %.0f layout %.0f wait_parse %.0f wait_push %.0f | push thread ms: %.0f\n",
                b->path.c_str(), (long long)b->n_records, (unsigned long long)t.chunks, (unsigned long long)t.slow_chunks, ns_since(t.t_open) * ms, b->n_threads,
                t.inflate.load() * ms, t.walk.load() * ms, t.parse.load() * ms,
                t.wait_inflate * ms, t.sched * ms, t.stitch * ms, t.layout * ms, t.wait_parse * ms, t.wait_push * ms, t.push * ms);
    }
    { BamScratch* sc = new BamScratch();
      for (int i = 0; i < N_SOA; i++) { sc->soa[i] = b->soa[i]; b->soa[i] = HostSoA(); }
      for (int i = 0; i < N_CHUNK; i++) sc->ubuf[i].swap(b->ch[i].ubuf);
      sc->recs.swap(b->recs); sc->rec_contig.swap(b->rec_contig); sc->rec_out.swap(b->rec_out);
      sc->pool = b->pool; sc->n_threads = b->n_threads; b->pool = nullptr;
      std::lock_guard<std::mutex> lk(g_scratch_mu);
      if (g_scratch.size() < 4) g_scratch.push_back(sc);
      else { delete sc->pool; for (auto& so : sc->soa) soa_free(so); delete sc; } }
    if (b->map) munmap((void*)b->map, b->fsize);
    if (b->fd >= 0) close(b->fd);
    delete b;
}

int xck_bam_n_refs(const xck_bam* b) { return b ? (int)b->ref_names.size() : 0; }
const char* xck_bam_ref_name(const xck_bam* b, int tid) { return (b && tid >= 0 && tid < (int)b->ref_names.size()) ? b->ref_names[tid].c_str() : nullptr; }
int64_t xck_bam_ref_len(const xck_bam* b, int tid) { return (b && tid >= 0 && tid < (int)b->ref_lens.size()) ? b->ref_lens[tid] : -1; }

}  // extern "C"

// ---- .bai (SAMv1 section 5.2): only what contig sharding needs: where each reference starts / ends ----
static void load_index(xck_bam* b) {
    if (b->idx_loaded) return;
    b->idx_loaded = true;
    std::string cand[2] = { b->path + ".bai", b->path.size() > 4 ? b->path.substr(0, b->path.size() - 4) + ".bai" : std::string() };
    std::vector<uint8_t> d;
    for (auto& fn : cand) { if (fn.empty()) continue; FILE* fp = fopen(fn.c_str(), "rb"); if (!fp) continue;
        fseek(fp, 0, SEEK_END); long n = ftell(fp); fseek(fp, 0, SEEK_SET); d.resize(n > 0 ? (size_t)n : 0);
        size_t got = d.empty() ? 0 : fread(d.data(), 1, d.size(), fp); fclose(fp); if (got == d.size() && got >= 8) break; d.clear(); }
    if (d.size() < 8 || memcmp(d.data(), "BAI\1", 4) != 0) return;
    size_t o = 4; auto need = [&](size_t k) { return o + k <= d.size(); };
    uint32_t n_ref = le32(d.data() + o); o += 4;
    if (n_ref != b->ref_names.size()) return;
    b->idx_beg.assign(n_ref, ~0ull); b->idx_end.assign(n_ref, 0); b->idx_mapped.assign(n_ref, 0); b->idx_unmapped.assign(n_ref, 0); b->idx_lin.assign(n_ref, {});
    auto le64 = [&](size_t at) { return (uint64_t)le32(d.data() + at) | ((uint64_t)le32(d.data() + at + 4) << 32); };
    for (uint32_t r = 0; r < n_ref; r++) {
        if (!need(4)) return;
        uint32_t n_bin = le32(d.data() + o); o += 4;
        for (uint32_t k = 0; k < n_bin; k++) {
            if (!need(8)) return;
            uint32_t bin = le32(d.data() + o), n_chunk = le32(d.data() + o + 4); o += 8;
            if (!need((size_t)n_chunk * 16)) return;
            if (bin == 37450 && n_chunk == 2) { b->idx_mapped[r] = (int64_t)le64(o + 16); b->idx_unmapped[r] = (int64_t)le64(o + 24); }
            else for (uint32_t c = 0; c < n_chunk; c++) { uint64_t cb = le64(o + c * 16), ce = le64(o + c * 16 + 8);
                if (cb < b->idx_beg[r]) b->idx_beg[r] = cb; if (ce > b->idx_end[r]) b->idx_end[r] = ce; }
            o += (size_t)n_chunk * 16;
        }
        if (!need(4)) return;
        uint32_t n_intv = le32(d.data() + o); o += 4;
        if (!need((size_t)n_intv * 8)) return;
        b->idx_lin[r].resize(n_intv);
        for (uint32_t w = 0; w < n_intv; w++) b->idx_lin[r][w] = le64(o + (size_t)w * 8);
        o += (size_t)n_intv * 8;
    }
    b->idx_ok = true;
}

// restrict decoding to the references wanted by tid_to_contig (virtual-offset ranges from the index)
static void set_ranges(xck_bam* b, const xck_ingest_opts* o) {
    b->ranges_set = true; b->use_ranges = false;
    if (!o->use_index || !o->tid_to_contig) return;
    load_index(b);
    if (!b->idx_ok) return;                                            // no usable index: decode everything
    const bool has_win = o->struct_size >= offsetof(xck_ingest_opts, tid_end) + sizeof(void*) && (o->tid_beg || o->tid_end);
    struct Rg { uint64_t first, second; int32_t tid; bool operator<(const Rg& o) const { return first != o.first ? first < o.first : second < o.second; } };
    std::vector<Rg> rg;
    for (size_t t = 0; t < b->ref_names.size(); t++) {
        if (!(o->tid_to_contig[t] >= 0 && b->idx_beg[t] != ~0ull && b->idx_end[t] > b->idx_beg[t])) continue;
        uint64_t beg = b->idx_beg[t];
        if (has_win && o->tid_beg && o->tid_beg[t] > 0) {                  // first record overlapping the window's first position: linear index
            const std::vector<uint64_t>& lin = b->idx_lin[t];
            const size_t w = (size_t)o->tid_beg[t] >> 14;
            if (w >= lin.size()) continue;                               // nothing overlaps that far right
            if (lin[w] > beg) beg = lin[w];
        }
        if (beg < b->idx_end[t]) rg.push_back({beg, b->idx_end[t], (int32_t)t});
    }
    std::sort(rg.begin(), rg.end());
    b->per_tid_ranges = has_win;
    for (auto& x : rg) { if (!has_win && !b->ranges.empty() && (x.first >> 16) <= (b->ranges.back().second >> 16) + 1) b->ranges.back().second = std::max(b->ranges.back().second, x.second);
                         else { b->ranges.push_back({x.first, x.second}); b->range_tid.push_back(x.tid); } }
    b->use_ranges = true;
}

// take the next plan from the scanner and start inflating it into c (asynchronous); tid_to_contig lets the walk tasks
// prepare the output layout of their own records
// One part of a chunk: inflate its blocks (all of them, or - st given - only those the GPU kernel left: status != 0), then walk the
// records of its byte range speculatively.  Pool task.
static void run_part(Chunk* cp, const uint8_t* map, bool verify_crc, WalkPart* wpp, DecodeTimes* tmp_, const ContigMap cm, bool want_seq, const int32_t* st) {
    if (!t_zs.ok) { cp->failed = true; return; }
    const auto t_a = std::chrono::steady_clock::now();
    struct Acc { DecodeTimes* t; std::chrono::steady_clock::time_point a, b; bool walked = false;
                 ~Acc() { const auto e = std::chrono::steady_clock::now(); if (!walked) b = e;
                          t->inflate += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count();
                          t->walk += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(e - b).count(); } } acc{tmp_, t_a, t_a};
    for (size_t i = wpp->i0; i < wpp->i1; i++) {
        if (st && st[i] == 0) continue;
        std::string e;
        if (!bgzf_inflate(map, cp->blocks[i], cp->ubase + cp->blocks[i].uoff, verify_crc, &t_zs.zs, &e)) {
            std::lock_guard<std::mutex> lk(cp->emu); cp->failed = true; cp->err = e; return;
        }
    }
    // speculative record walk over this range
    acc.b = std::chrono::steady_clock::now(); acc.walked = true;
    const uint8_t* u = cp->ubase; size_t o = wpp->spec_start; const size_t end = wpp->u_end;
    if (o > end) { wpp->stop = o; return; }
    int32_t cur_c = INT32_MIN;
    while (o + 4 <= end) {
        uint32_t bs = le32(u + o);
        if (bs < 32 || o + 4 + (size_t)bs > end) break;
        const uint8_t* r = u + o + 4;
        // (a device chunk's bytes come from the GPU, not from this core's inflate: the hop from header to header is a chain of cache
        // misses unless the lines a few records ahead are asked for early)
        { const uint8_t* pf = r + 4096; __builtin_prefetch(pf); __builtin_prefetch(pf + 64); __builtin_prefetch(pf + 128); __builtin_prefetch(pf + 192); }   // (every line of the stretch ~18 records ahead: a header may sit on any of them)
        const RecRef rr{r, bs, (int32_t)le32(r), le32(r + 16), le16(r + 12)};
        const int32_t ctg = cm(rr.tid, (int32_t)le32(r + 4));
        if (ctg != cur_c) { wpp->runs.push_back({(uint32_t)wpp->recs.size(), ctg}); cur_c = ctg; }
        if (ctg >= 0) { wpp->n_out++; wpp->n_cig += rr.n_cig; if (want_seq) wpp->n_seq += (rr.l_seq + 1) / 2; }
        wpp->recs.push_back(rr);
        o += 4 + (size_t)bs;
    }
    wpp->stop = o;
}

static void schedule_chunk(xck_bam* b, Chunk& c, int ci, bool verify_crc, const ContigMap cm_in, bool want_seq) {
    c.blocks.clear(); c.usize = 0; c.valid = false; c.failed = false; c.err.clear(); c.new_range = false; c.first_skip = 0; c.gpu = false; c.gpu_rc = 0; c.launched = 0; c.stage2 = false;
    if (b->scan_end) return;
    ChunkPlan pl = b->scanner->next();
    if (pl.end) { b->scan_end = true; return; }
    c.blocks.swap(pl.blocks); c.new_range = pl.new_range; c.first_skip = pl.first_skip; c.range_id = pl.range_id;
    if (pl.failed) { c.failed = true; c.err = pl.err; }
    if (c.blocks.empty() && !c.failed) return;
    c.valid = true; c.usize = pl.usize;
    const size_t usz = pl.usize;
    const size_t nb = c.blocks.size();
    // ---- does this chunk go to the GPU?  (a fixed share of the chunks; small chunks - the tail of a range - stay on the host)
    GpuShare& gi = b->gi;
    GpuInflateSlot* gs = nullptr;
    if (gi.on && !gi.broken && !verify_crc && nb >= 64 && usz < (size_t(1) << 31)) {
        bool want;
        if (gi.pct > 0) { gi.acc += gi.pct; want = gi.acc >= 100; if (want) gi.acc -= 100; }       // a fixed share
        else {                                                            // auto: keep gi.depth chunks on the device, the pool takes the rest -
            int busy = 0;                                                 // the shares then follow the two sides' speeds on this file and this box
            int held = 0;
            for (int k = 0; k < b->n_ring; k++) { if (!gi.slot[k]) continue; held++; if (gi.inflight[k] && !gpu_inflate_slot_done(gi.slot[k])) busy++; }
            // ... but never the chunk the coordinator needs next or the one after (a device chunk takes tens of milliseconds, the pool
            // delivers in three: the first chunks of a file, and whatever follows a drained ring, stay on the host)
            want = busy < gi.depth && held < gi.max_held && (b->n_sched >= 2 || b->prefetching);   // (reading ahead: nobody waits for these chunks yet)
        }
        if (want) {
            size_t tin = 0; for (size_t i = 0; i < nb; i++) tin += ((size_t)c.blocks[i].data_len + 3) & ~size_t(3);
            if (!gi.slot[ci]) {
                if (!gi.free_slots.empty()) { gi.slot[ci] = gi.free_slots.back(); gi.free_slots.pop_back(); }
                else gi.slot[ci] = take_gpu_slot(gi.device, gi.free_cus, gi.verbose && !gi.chunks);
            }
            gs = gi.slot[ci];
            if (gs && gi.inflight[ci]) { gpu_inflate_slot_wait(gs); gi.inflight[ci] = false; }   // (a chunk that was dropped unconsumed)
            if (!gs || !gpu_inflate_slot_reserve(gs, tin + 8, usz + 8, nb)) { gi.broken = true; gs = nullptr; }   // no device memory: the host does it all from here on
            else c.in_total = tin;
        }
    }
    if (gs) { c.gpu = true; c.ubase = gs->h_out; gi.inflight[ci] = true; }
    else { if (c.ubuf.size() < usz + 8) c.ubuf.resize(usz + 8); c.ubase = c.ubuf.data(); }
    const size_t per = std::max<size_t>(1, (nb + (size_t)b->n_threads * 4 - 1) / ((size_t)b->n_threads * 4));
    const size_t n_parts = (nb + per - 1) / per;
    c.parts.resize(n_parts);
    const size_t skip0 = c.first_skip;             // bytes before the first record (first chunk of the file / of an index range)
    ContigMap cm = cm_in; cm.only_tid = b->per_tid_ranges && (size_t)c.range_id < b->range_tid.size() ? b->range_tid[c.range_id] : -1;
    if (gs) {                                       // block table: where every block's stream sits in the gathered input, where its bytes go
        size_t at = 0;
        for (size_t i = 0; i < nb; i++) { gs->h_bl[i] = DevBlock{(uint32_t)at, c.blocks[i].data_len, (uint32_t)c.blocks[i].uoff, c.blocks[i].isize}; at += ((size_t)c.blocks[i].data_len + 3) & ~size_t(3); }
        c.copy_left = (int)n_parts; gi.chunks++; gi.blocks += nb;
    }
    for (size_t pi = 0; pi < n_parts; pi++) {
        size_t i0 = pi * per, i1 = std::min(nb, i0 + per);
        WalkPart& wp = c.parts[pi];
        wp.i0 = i0; wp.i1 = i1;
        wp.u_begin = c.blocks[i0].uoff; wp.u_end = c.blocks[i1 - 1].uoff + c.blocks[i1 - 1].isize;
        wp.spec_start = wp.u_begin + (pi == 0 ? skip0 : 0); wp.stop = wp.spec_start; wp.recs.clear();
        wp.n_out = wp.n_cig = wp.n_seq = 0; wp.runs.clear();
        Chunk* cp = &c; const uint8_t* map = b->map; WalkPart* wpp = &wp; DecodeTimes* tmp_ = &b->tm;
        if (!gs) { c.tg.later([cp, map, verify_crc, wpp, tmp_, cm, want_seq] { run_part(cp, map, verify_crc, wpp, tmp_, cm, want_seq, nullptr); }); continue; }
        GpuShare* gip = &gi; const size_t nb_ = nb, usz_ = usz;
        c.tg.later([cp, map, i0, i1, gs, gip, nb_, usz_] {
            const auto t_a = std::chrono::steady_clock::now();
            for (size_t i = i0; i < i1; i++) memcpy(gs->h_in + gs->h_bl[i].in_off, map + cp->blocks[i].coff + cp->blocks[i].data_off, cp->blocks[i].data_len);
            __atomic_fetch_add(&gip->copy_ns, ns_since(t_a), __ATOMIC_RELAXED);
            if (cp->copy_left.fetch_sub(1) == 1) {                         // the chunk's compressed bytes are gathered: hand it to the device
                const int rc = gpu_inflate_slot_launch(gs, cp->in_total, usz_, nb_);
                cp->gpu_rc = rc; cp->launched.store(1, std::memory_order_release);
            }
        });
    }
    c.tg.flush(*b->pool);
}

// ---- NUMA: the decoder stays on ONE socket ---------------------------------------------------------------------------
// A GPU box has two sockets behind a CPU-time quota; left alone, the scheduler spreads the pool over both and every chunk's
// inflated bytes, records and SoA block cross the socket link: 35 M reads/s unpinned against 40.5-41.8 M with the process held
// on either socket (24 threads, same box, same call: profiles/r02_v_numa_affinity.log).  At the first decode call the pool, the
// scanner and the calling thread (whose first touch places the inflate buffers) are bound to the CPUs of the GPU's NUMA node
// (decode-only handle: the node the caller runs on); the caller's mask is restored by xck_bam_close.  XCK_NUMA=0 turns it off.
static bool read_cpulist(const char* path, cpu_set_t* out) {
    FILE* f = fopen(path, "r"); if (!f) return false;
    char buf[4096]; const size_t n = fread(buf, 1, sizeof buf - 1, f); fclose(f); buf[n] = 0;
    CPU_ZERO(out); int any = 0;
    for (char* p = buf; *p;) {
        while (*p == ',' || *p == ' ' || *p == '\n') p++;
        if (!*p) break;
        char* q; long a = strtol(p, &q, 10); if (q == p) break; long z = a; p = q;
        if (*p == '-') { p++; z = strtol(p, &q, 10); if (q == p) break; p = q; }
        for (long c = a; c <= z && c < CPU_SETSIZE; c++) { CPU_SET((int)c, out); any = 1; }
    }
    return any != 0;
}
static int node_of_cpu(int cpu, cpu_set_t* node_set) {
    for (int node = 0; node < 64; node++) {
        char path[96]; snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
        cpu_set_t s; if (!read_cpulist(path, &s)) { if (node > 8) break; continue; }
        if (cpu >= 0 && CPU_ISSET(cpu, &s)) { *node_set = s; return node; }
    }
    return -1;
}
struct CallerBinding {
    xck_bam* b; cpu_set_t saved; bool active = false;
    explicit CallerBinding(xck_bam* b_) : b(b_) { if (b) { b->cur_binding = this; enter(); } }
    void enter() {
        if (!b || !b->numa_bound || active) return;
        if (sched_getaffinity(0, sizeof saved, &saved) != 0) return;
        if (pthread_setaffinity_np(pthread_self(), sizeof b->numa_set, &b->numa_set) == 0) active = true;
    }
    ~CallerBinding() { if (active) pthread_setaffinity_np(pthread_self(), sizeof saved, &saved); if (b) b->cur_binding = nullptr; }
};
static void bind_to_numa_node(xck_engine* e, xck_bam* b) {
    if (const char* v = getenv("XCK_NUMA")) if (atoi(v) == 0) return;
    if (!b->pool || b->n_threads < 2) return;
    cpu_set_t node_set; CPU_ZERO(&node_set);
    int node = e && e->n_impl > 0 ? xck::engine_numa_node(e) : -1;
    if (node >= 0) { char path[96]; snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node); if (!read_cpulist(path, &node_set)) node = -1; }
    if (node < 0) node = node_of_cpu(sched_getcpu(), &node_set);
    if (node < 0) return;
    cpu_set_t cur; CPU_ZERO(&cur);
    if (sched_getaffinity(0, sizeof cur, &cur) != 0) return;
    cpu_set_t both; CPU_AND(&both, &cur, &node_set);
    if (CPU_COUNT(&both) < 2 || CPU_COUNT(&both) == CPU_COUNT(&cur)) return;     // not allowed there, or there is only this node
    // a thread count that was derived from the CPUs of the WHOLE machine (no quota, two sockets) is cut to the node's CPUs
    if (b->threads_auto && b->n_threads > CPU_COUNT(&both)) { delete b->pool; b->n_threads = CPU_COUNT(&both); b->pool = new Pool(b->n_threads); }
    b->old_affinity = cur; b->numa_set = both; b->numa_bound = true;
    b->pool->set_affinity(both);
    if (b->cur_binding) b->cur_binding->enter();                     // the rest of this decode call (the scanner thread made next inherits it)
}

// locate a 2-character aux tag; returns pointer to the type byte or nullptr
static inline const uint8_t* aux_skip(const uint8_t* p, const uint8_t* e) {       // p at type byte; returns next tag or nullptr
    if (p >= e) return nullptr;
    uint8_t t = *p++;
    switch (t) {
        case 'A': case 'c': case 'C': p += 1; break;
        case 's': case 'S': p += 2; break;
        case 'i': case 'I': case 'f': p += 4; break;
        case 'Z': case 'H': { const void* z = memchr(p, 0, (size_t)(e - p)); if (!z) return nullptr; p = (const uint8_t*)z + 1; break; }
        case 'B': { if (p + 5 > e) return nullptr; uint8_t st = *p; uint32_t n = le32(p + 1); p += 5;
                    size_t sz = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4; p += (size_t)n * sz; break; }
        default: return nullptr;
    }
    return p <= e ? p : nullptr;
}

// one record -> SoA slot o (cig_off[o] / seq_off[o] are already set)
static inline void parse_record(const DecodeCfg& dc, xck_engine* e, HostSoA& s, const uint8_t* p, uint32_t len, int64_t o, int32_t sample,
                                std::vector<InternTable::Item>& names, std::atomic<int>* flags) {
        const uint8_t* end = p + len;
        uint32_t l_name = p[8], n_cig = le16(p + 12), l_seq = le32(p + 16);
        s.pos[o] = (int32_t)le32(p + 4);
        s.mapq[o] = p[9];
        s.flag[o] = le16(p + 14);
        const uint8_t* q = p + 32;
        const char* qname = (const char*)q; size_t qlen = l_name ? l_name - 1 : 0;
        q += l_name;
        if (q + (size_t)n_cig * 4 + (l_seq + 1) / 2 + l_seq > end) { flags->fetch_or(1); s.cell[o] = -1; s.umi[o] = XCK_UMI_NONE; return; }
        memcpy(s.cigar + s.cig_off[o], q, (size_t)n_cig * 4);
        q += (size_t)n_cig * 4;
        if (dc.want_seq) memcpy(s.seq + s.seq_off[o], q, (l_seq + 1) / 2);
        q += (l_seq + 1) / 2 + l_seq;
        // aux tags: first occurrence wins (htslib bam_aux_get)
        const uint8_t* cb = nullptr; const uint8_t* ub = nullptr;
        bool need_cb = dc.use_barcodes, need_ub = dc.use_umi;
        const uint8_t* a = q;
        while ((need_cb || need_ub) && a && a + 3 <= end) {
            if (need_cb && a[0] == (uint8_t)dc.cell_tag[0] && a[1] == (uint8_t)dc.cell_tag[1]) { cb = a + 2; need_cb = false; }
            else if (need_ub && a[0] == (uint8_t)dc.umi_tag[0] && a[1] == (uint8_t)dc.umi_tag[1]) { ub = a + 2; need_ub = false; }
            a = aux_skip(a + 2, end);
        }
        int32_t cell = -1;
        if (dc.use_barcodes) {
            if (cb && (*cb == 'Z' || *cb == 'A' || *cb == 'H')) {         // get_tag() returns str only for these
                const char* v = (const char*)cb + 1; size_t n = (*cb == 'A') ? 1 : strnlen(v, (size_t)(end - (cb + 1)));
                cell = dc.lookup_cell(v, n);
            }
        } else cell = sample;
        s.cell[o] = cell;
        uint64_t key = XCK_UMI_NONE; bool ovf = false;
        if (dc.use_umi) {
            if (ub) {
                if (*ub == 'Z' || *ub == 'A' || *ub == 'H') {
                    const char* v = (const char*)ub + 1; size_t n = (*ub == 'A') ? 1 : strnlen(v, (size_t)(end - (ub + 1)));
                    key = encode_key(v, n, dc.umi_bits, e->intern, &ovf);
                } else {                                                   // numeric tag: get_tag() gives a number; the key is its VALUE
                    // (an int 0 / float 0.0 is falsy: `if not umi` skips the read, rdr/fc/mcount.py:41, baf/fc/mcount.py:116-117)
                    const uint8_t* nx = aux_skip(ub, end);
                    bool zero = false; std::string t("\1num:");
                    if (nx) {
                        const uint8_t* v = ub + 1; long long iv = 0; bool is_int = true;
                        switch (*ub) { case 'c': iv = (int8_t)v[0]; break; case 'C': iv = v[0]; break; case 's': iv = (int16_t)le16(v); break; case 'S': iv = le16(v); break;
                                       case 'i': iv = (int32_t)le32(v); break; case 'I': iv = le32(v); break; default: is_int = false; }
                        if (is_int) { zero = iv == 0; t += std::to_string(iv); }          // equal integers are equal keys whatever their storage width
                        else if (*ub == 'f') { float f; memcpy(&f, v, 4); zero = f == 0.0f; t.append("f", 1); t.append((const char*)v, 4); }
                        else t.append((const char*)ub, (size_t)(nx - ub));                 // B arrays: their bytes
                    }
                    if (!zero) key = encode_key(t.data(), t.size(), dc.umi_bits, e->intern, &ovf);
                }
            }
        } else if (encode_key_direct(qname, qlen, dc.umi_bits, &key)) { /* empty or 2-bit codable name */ }
        else { names.push_back(InternTable::Item{InternTable::hash(qname, qlen), qname, (uint32_t)qlen, 0, &s.umi[o], dc.umi_bits}); key = XCK_UMI_NONE; }
        if (ovf) flags->fetch_or(2);
        s.umi[o] = key;
}

// slow path: records [r0, r1) of the serially stitched list (b->recs / b->rec_out)
static void parse_range(xck_bam* b, xck_engine* e, int32_t sample, int64_t r0, int64_t r1, std::atomic<int>* flags) {
    const auto t_parse0 = std::chrono::steady_clock::now();
    struct PAcc { xck_bam* b; std::chrono::steady_clock::time_point t0; ~PAcc() { b->tm.parse += ns_since(t0); } } pacc{b, t_parse0};
    const DecodeCfg& dc = e->dec;
    HostSoA& s = b->soa[b->soa_i];
    // UMI-less mode: every read name goes through the intern table - collected here, interned in one batch at the end
    std::vector<InternTable::Item> names;
    if (!dc.use_umi) names.reserve((size_t)(r1 - r0));
    for (int64_t r = r0; r < r1; r++) {
        const int64_t o = b->rec_out[r];
        if (o < 0) continue;
        parse_record(dc, e, s, b->recs[r].p, b->recs[r].len, o, sample, names, flags);
    }
    if (!names.empty()) { bool ovf = false; e->intern.intern_batch(names, &ovf); if (ovf) flags->fetch_or(2); }
}

// fast path: the records of one walk part, whose output slots start at the part's bases (prefix sums over the parts)
static void parse_part(xck_bam* b, xck_engine* e, HostSoA* sp, int32_t sample, const WalkPart* wp, const ContigMap cm, std::atomic<int>* flags) {
    const auto t_parse0 = std::chrono::steady_clock::now();
    struct PAcc { xck_bam* b; std::chrono::steady_clock::time_point t0; ~PAcc() { b->tm.parse += ns_since(t0); } } pacc{b, t_parse0};
    const DecodeCfg& dc = e->dec;
    HostSoA& s = *sp;
    std::vector<InternTable::Item> names;
    if (!dc.use_umi) names.reserve(wp->n_out);
    int64_t o = (int64_t)wp->out_base; uint32_t co = (uint32_t)wp->cig_base, so = (uint32_t)wp->seq_base;
    const RecRef* const rend = wp->recs.data() + wp->recs.size();
    for (const RecRef& rr : wp->recs) {
        if (&rr + 3 < rend) { const RecRef& nx = (&rr)[3]; __builtin_prefetch(nx.p); __builtin_prefetch(nx.p + 64); __builtin_prefetch(nx.p + nx.len - 64); __builtin_prefetch(nx.p + nx.len - 128); }
        const int32_t ctg = cm(rr.tid, (int32_t)le32(rr.p + 4));
        if (ctg < 0) continue;
        s.cig_off[o] = co; s.seq_off[o] = so;
        parse_record(dc, e, s, rr.p, rr.len, o, sample, names, flags);
        co += rr.n_cig; if (dc.want_seq) so += (rr.l_seq + 1) / 2;
        o++;
    }
    if (!names.empty()) { bool ovf = false; e->intern.intern_batch(names, &ovf); if (ovf) flags->fetch_or(2); }
}

// decode the next chunk into the SoA and fill b->pending. returns 1 (decoded), 0 (eof), <0 error
// the chunk at the head of the ring is consumed (or dropped): its slot, if it had one and nothing is in flight on it, serves the next device chunk
static void release_chunk(xck_bam* b, int ci) {
    GpuShare& gi = b->gi;
    if (gi.slot[ci] && !gi.inflight[ci]) { gi.free_slots.push_back(gi.slot[ci]); gi.slot[ci] = nullptr; }
}
static void advance_head(xck_bam* b, IngestJob* hold = nullptr) {     // hold: that job's parse still reads the chunk (reap_jobs takes it back)
    if (hold) { hold->ring_idx = b->head; hold->released = false; } else release_chunk(b, b->head);
    b->head = (b->head + 1) % b->n_ring; b->n_sched--;
}
// ring chunks whose parse has ended go back to the ring (coordinator only)
static void reap_jobs(xck_bam* b) {
    for (auto& j : b->jobs) if (!j.released && j.parsed.load(std::memory_order_acquire)) { release_chunk(b, j.ring_idx); j.released = true; }
}
static Pusher* get_pusher(xck_engine* e, xck_bam* b) {
    if (!b->pusher) { b->pusher = new Pusher(); Pusher* pp = b->pusher; pp->e = e; pp->b = b; pp->th = std::thread([pp] { pp->run(); }); }
    return b->pusher;
}

void Pusher::run() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
        cv.wait(lk, [this] { return stop || qn > 0; });
        if (qn == 0) return;
        IngestJob* j = q[qh];
        const bool failed = rc != 0;
        lk.unlock();
        int r = 0; std::string msg;
        if (j->has_parse) {
            const auto t0 = std::chrono::steady_clock::now();
            j->tg.wait();
            parse_wait_ns += ns_since(t0);
            int fl = j->flags.load();
            if (const int th = j->tg.take_thrown()) fl |= th == 1 ? 4 : 8;
            if (fl & 4) { msg = "out of host memory (record parse)"; r = XCK_E_NOMEM; }
            else if (fl & 8) { msg = "C++ exception in a parse task"; r = XCK_E_IO; }
            else if (fl & 1) { msg = "corrupt BAM record (fields exceed block_size)"; r = XCK_E_IO; }
            else if (fl & 2) { msg = "too many distinct non-ACGT keys for the key width"; r = XCK_E_CAPACITY; }
        }
        j->parsed.store(1, std::memory_order_release);
        if (!failed && !r && e->n_impl > 0 && !j->pending.empty()) {
            // the whole chunk crosses PCIe as ONE block; its batches are slices of the block (fused handles: both pipelines read the same copy)
            const auto t0 = std::chrono::steady_clock::now();
            HostSoA& s = b->soa[j->soa];
            std::vector<xck_batch> bts; bts.reserve(j->pending.size());
            for (const PendingBatch& pb : j->pending) {
                xck_batch bt; memset(&bt, 0, sizeof bt);
                bt.contig = pb.contig; bt.n_reads = (int32_t)(pb.r1 - pb.r0); bt.ordinal_base = pb.ordinal_base;
                bt.pos = s.pos + pb.r0; bt.flag = s.flag + pb.r0; bt.mapq = s.mapq + pb.r0; bt.cell = s.cell + pb.r0; bt.umi = s.umi + pb.r0;
                bt.cig_off = s.cig_off + pb.r0; bt.cigar = s.cigar;
                if (e->dec.want_seq) { bt.seq_off = s.seq_off + pb.r0; bt.seq = s.seq; }
                bts.push_back(bt);
            }
            try { r = xck::engine_push_block(e, s.base, s.used, bts.data(), (int)bts.size(), &s.fence); } catch (const std::bad_alloc&) { r = XCK_E_NOMEM; } catch (...) { r = XCK_E_IO; }
            if (r) msg = e->err;
            push_ns += ns_since(t0);
        }
        lk.lock();
        if (r && !rc) { rc = r; err = msg; }
        qh = (qh + 1) % MAXQ; qn--;
        cv.notify_all();
    }
}

// A device chunk whose inflate is over: whatever the device did not inflate - single blocks with a non-zero status, or the whole chunk
// after a runtime error - the pool inflates now, and every part walks its records as on the host path.  Called for the chunk the
// coordinator has reached (wait = true: for the device if need be) and, without waiting, for the chunks behind it whose device part
// has already ended - their walk is then over by the time the coordinator gets there (it used to wait ~0.5 ms per chunk for it).
static void start_stage_two(xck_engine* e, xck_bam* b, int ci, ContigMap cm, bool wait) {
    Chunk& c = b->ch[ci];
    if (!c.gpu || c.stage2 || !c.valid || c.failed || (b->skip_range >= 0 && c.range_id == b->skip_range)) return;
    GpuInflateSlot* gs = b->gi.slot[ci];
    if (!wait && !(c.launched.load(std::memory_order_acquire) && (c.gpu_rc.load() != 0 || gpu_inflate_slot_done(gs)))) return;
    if (wait) c.tg.wait();                                               // (the gather tasks: the last of them launches)
    const auto t_w = std::chrono::steady_clock::now();
    const bool dev_ok = c.gpu_rc.load() == 0 && gpu_inflate_slot_wait(gs) == 0;
    b->gi.inflight[ci] = false;
    if (wait) b->gi.wait_ns += ns_since(t_w);
    if (!dev_ok) b->gi.broken = true;                                    // the device path stays off for the rest of this reader
    else e->gpu_inflate_chunks++;
    const int32_t* st = dev_ok ? gs->h_st : nullptr;
    if (dev_ok) for (size_t i = 0; i < c.blocks.size(); i++) b->gi.left_blocks += st[i] != 0;
    else b->gi.left_blocks += c.blocks.size();
    cm.only_tid = b->per_tid_ranges && (size_t)c.range_id < b->range_tid.size() ? b->range_tid[c.range_id] : -1;
    Chunk* cp = &c; const uint8_t* map = b->map; DecodeTimes* tmp_ = &b->tm; const bool want_seq = e->dec.want_seq;
    for (WalkPart& wp : c.parts) { WalkPart* wpp = &wp; c.tg.later([cp, map, wpp, tmp_, cm, want_seq, st] { run_part(cp, map, false, wpp, tmp_, cm, want_seq, st); }); }
    c.tg.flush(*b->pool, true);                                          // (ahead of the later chunks' inflate tasks)
    c.stage2 = true;
}

static int decode_next_chunk(xck_engine* e, xck_bam* b, const xck_ingest_opts* o, bool schedule_only = false) {
    const bool crc = e->dec.verify_crc;
    const int n_refs = (int)b->ref_names.size();
    const bool has_win = o->struct_size >= offsetof(xck_ingest_opts, tid_end) + sizeof(void*);
    ContigMap cm; cm.t2c = o->tid_to_contig; cm.n_refs = n_refs; cm.t_end = has_win ? o->tid_end : nullptr;
    if (!schedule_only && !b->started) {
        b->started = true;
        // well mode without UMIs: the key is the read name and the column is the BAM itself, so names only
        // have to be unique within one file - restart the intern table per BAM (bounded memory for 384 x 2 M reads).
        // (At the first DECODE call, not at xck_bam_prefetch: the file before this one may still be parsing.)
        if (!e->dec.use_barcodes && !e->dec.use_umi) e->intern.clear();
    }
    if (!b->ranges_set) {
        set_ranges(b, o);
        if (!b->gi.tried) {                                    // GPU share of the inflate: a handle with a device, XCK_GPU_INFLATE > 0, no CRC checks asked for
            b->gi.tried = true;
            const int dev = engine_device(e), pct = e->knobs.gpu_inflate_pct;
            // (files of a few chunks - the per-cell BAMs of a well-based run - are done before the device has returned its first chunk)
            uint64_t span = b->fsize;
            if (b->use_ranges) { span = 0; for (auto& r : b->ranges) span += (r.second >> 16) - (r.first >> 16); }
            const bool big = span >= (uint64_t)e->knobs.gpu_inflate_min_mb << 20 || schedule_only;   // (a file that is read ahead has the time a device chunk takes, whatever its size)
            if (pct != 0 && dev >= 0 && !crc && (big || pct > 0)) { b->gi.on = true; b->gi.pct = pct < 0 ? 0 : std::min(pct, 100); b->gi.depth = e->knobs.gpu_inflate_depth; b->gi.max_held = std::min(b->gi.depth + 2, 12); b->gi.device = dev; b->gi.free_cus = e->knobs.gpu_inflate_free_cus; b->gi.verbose = e->knobs.debug_timing; b->n_ring = std::max(N_CHUNK + 1, std::min(N_CHUNK_GPU, e->knobs.gpu_inflate_ring)); }
        }
        bind_to_numa_node(e, b);                               // (before the scanner thread is made: it inherits the mask)
        std::vector<ScanRange> rg;
        if (b->use_ranges) for (auto& r : b->ranges) rg.push_back({r.first >> 16, (uint32_t)(r.first & 0xffff), r.second >> 16});
        else rg.push_back({b->next_coff, b->first_skip, ~0ull});
        b->scanner = new Scanner(b->map, b->fsize, b->chunk_target, rg);
    }
    auto t_ph = std::chrono::steady_clock::now();
    auto phase = [&](uint64_t& acc) { const auto now = std::chrono::steady_clock::now(); acc += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(now - t_ph).count(); t_ph = now; };
    // keep the ring full: while this chunk is stitched and parsed the pool inflates the next two
    reap_jobs(b);
    // (the chunks behind `head` that a parse in flight still reads must not be scheduled over: `back` = how far the oldest of them lies behind)
    int back = 0;
    for (auto& j : b->jobs) if (!j.released) back = std::max(back, (b->head - j.ring_idx + b->n_ring) % b->n_ring);
    while (b->n_sched + back < b->n_ring && !b->scan_end) {
        const int ci = (b->head + b->n_sched) % b->n_ring;
        Chunk& nc = b->ch[ci];
        schedule_chunk(b, nc, ci, crc, cm, e->dec.want_seq);
        if (b->scan_end) break;
        b->n_sched++;
    }
    phase(b->tm.sched);
    if (schedule_only) return 2;                                         // xck_bam_prefetch: the pool is inflating the first chunks now
    if (b->n_sched == 0) {
        if (!b->use_ranges && !b->carry.empty()) { b->err = "truncated BAM file (partial record at end of file)"; return XCK_E_IO; }
        return 0;
    }
    if (b->gi.on) for (int k = 1; k < b->n_sched; k++) start_stage_two(e, b, (b->head + k) % b->n_ring, cm, false);   // device chunks that are back already: walk them now
    Chunk& c = b->ch[b->head];
    cm.only_tid = b->per_tid_ranges && (size_t)c.range_id < b->range_tid.size() ? b->range_tid[c.range_id] : -1;
    c.tg.wait();
    if (const int th = c.tg.take_thrown()) { b->err = th == 1 ? "out of host memory (BGZF inflate / record walk)" : "C++ exception in a decoder task"; return th == 1 ? XCK_E_NOMEM : XCK_E_IO; }
    if (c.gpu && !c.stage2) {
        start_stage_two(e, b, b->head, cm, true);
        c.tg.wait();
        if (const int th = c.tg.take_thrown()) { b->err = th == 1 ? "out of host memory (BGZF inflate / record walk)" : "C++ exception in a decoder task"; return th == 1 ? XCK_E_NOMEM : XCK_E_IO; }
    }
    if (b->skip_range >= 0 && c.range_id == b->skip_range && !c.failed) {   // rest of a reference whose position window is behind us
        if (c.gpu && b->gi.inflight[b->head]) { gpu_inflate_slot_wait(b->gi.slot[b->head]); b->gi.inflight[b->head] = false; }
        advance_head(b);
        b->carry.clear();
        return decode_next_chunk(e, b, o);
    }
    phase(b->tm.wait_inflate); b->tm.chunks++;
    if (c.failed) { b->err = c.err.empty() ? "BGZF decode error" : c.err; return XCK_E_IO; }
    if (c.new_range) {                                                  // start of the file's records / of an index range
        if (!b->use_ranges && b->tm.chunks > 1 && !b->carry.empty()) { b->err = "internal: carry at a range start"; return XCK_E_IO; }
        b->carry.clear(); b->stitch_skip = c.first_skip;
    }
    const uint8_t* u = c.ubase; const size_t usz = c.usize; size_t off = 0;
    if (b->stitch_skip) { off = b->stitch_skip; b->stitch_skip = 0; if (off > usz) { b->err = "corrupt BAM header offset"; return XCK_E_IO; } }
    // ---- fast path: no carried-over bytes, and every part's speculative walk began where the previous one ended ----
    bool fast = b->carry.empty() && !(o->max_records > 0);
    if (fast) {
        size_t at = off;
        for (size_t pi = 0; pi < c.parts.size() && fast; pi++) {
            const WalkPart& wp = c.parts[pi];
            if (wp.spec_start != at) fast = false;
            at = wp.stop;
            if (pi + 1 < c.parts.size() && wp.stop != wp.u_end) fast = false;      // a record straddles two parts
        }
        if (fast && at + 4 <= usz && le32(u + at) < 32) fast = false;              // not a straddling tail but a damaged record: the serial walk reports it
    }
    HostSoA& s = b->soa[(b->soa_i + 1) % N_SOA];
    if (s.fence) fence_wait(s.fence);                                   // the H2D copy that last read this block has completed
    b->soa_i = (b->soa_i + 1) % N_SOA;
    b->pending.clear();
    const uint64_t ord_hi = (uint64_t)(uint32_t)o->sample << ORD_REC_BITS;
    std::atomic<int> flags{0};
    int64_t limit = 0;
    bool hit_limit = false;
    if (fast) {
        size_t n_out = 0, n_cig = 0, n_seq = 0, n_rec = 0;
        for (WalkPart& wp : c.parts) { wp.out_base = n_out; wp.cig_base = n_cig; wp.seq_base = n_seq; n_out += wp.n_out; n_cig += wp.n_cig; n_seq += wp.n_seq; n_rec += wp.recs.size(); }
        const WalkPart& last = c.parts.back();
        if (last.stop < usz) b->carry.insert(b->carry.end(), u + last.stop, u + usz);
        phase(b->tm.stitch);
        if (n_cig >= (size_t(1) << 32) || n_seq >= (size_t(1) << 32)) { b->err = "chunk too large"; return XCK_E_IO; }
        if (!soa_reserve(s, n_out + 1, n_cig + 1, e->dec.want_seq ? n_seq + 1 : 1)) { b->err = "out of host memory"; return XCK_E_NOMEM; }
        s.cig_off[n_out] = (uint32_t)n_cig; s.seq_off[n_out] = (uint32_t)n_seq;
        // batches = runs of records on one contig, across part boundaries
        { int32_t cur_c = -1; int64_t seg0 = 0, seg_r = 0; int64_t rec0 = 0;
          for (const WalkPart& wp : c.parts) {
              int64_t oi = (int64_t)wp.out_base;
              for (size_t ri = 0; ri < wp.runs.size(); ri++) {
                  const ContigRun& rn = wp.runs[ri];
                  const uint32_t nxt = ri + 1 < wp.runs.size() ? wp.runs[ri + 1].first : (uint32_t)wp.recs.size();
                  if (rn.contig != cur_c) {
                      if (cur_c >= 0 && oi > seg0) b->pending.push_back({cur_c, seg0, oi, ord_hi | (uint64_t)(b->n_records + seg_r)});
                      cur_c = rn.contig; seg0 = oi; seg_r = rec0 + rn.first;
                  }
                  if (rn.contig >= 0) oi += nxt - rn.first;
              }
              rec0 += (int64_t)wp.recs.size();
          }
          if (cur_c >= 0 && (int64_t)n_out > seg0) b->pending.push_back({cur_c, seg0, (int64_t)n_out, ord_hi | (uint64_t)(b->n_records + seg_r)}); }
        phase(b->tm.layout);
        limit = (int64_t)n_rec;
        if (b->defer_parse) {
            IngestJob& jb = b->jobs[b->job_n % (PUSH_Q + 1)];           // (free: at most PUSH_Q jobs are outstanding)
            reap_jobs(b);
            const int32_t smp = o->sample; HostSoA* sp = &s; std::atomic<int>* fl = &jb.flags;
            jb.flags.store(0); jb.parsed.store(0); jb.has_parse = true; jb.soa = b->soa_i; jb.pending.swap(b->pending); b->pending.clear();
            for (const WalkPart& wp : c.parts) { if (!wp.n_out) continue; const WalkPart* wpp = &wp;
                jb.tg.later([b, e, sp, smp, wpp, cm, fl] { parse_part(b, e, sp, smp, wpp, cm, fl); }); }
            jb.tg.flush(*b->pool, true);                                // (ahead of the later chunks' inflate tasks)
            if (b->per_tid_ranges && cm.t_end) {                        // (same check as below: it only looks at the record lists)
                const RecRef* last = nullptr;
                for (size_t pi = c.parts.size(); pi-- > 0 && !last;) if (!c.parts[pi].recs.empty()) last = &c.parts[pi].recs.back();
                if (last && last->tid >= 0 && last->tid < n_refs && cm.t_end[last->tid] > 0 && (int32_t)le32(last->p + 4) >= cm.t_end[last->tid]) {
                    b->skip_range = c.range_id; b->scanner->abandon(c.range_id);
                }
            }
            b->n_records += limit;
            advance_head(b, &jb);
            b->job_n++;
            const int prc = get_pusher(e, b)->submit(&jb);              // waits while two chunks are outstanding behind this one
            phase(b->tm.wait_push);
            if (prc) { b->err = b->pusher->err; return prc; }
            return 2;                                                   // parse + push in flight
        }
        { TaskGroup tg; const int32_t smp = o->sample; HostSoA* sp = &s;
          for (const WalkPart& wp : c.parts) { if (!wp.n_out) continue; const WalkPart* wpp = &wp;
              tg.later([b, e, sp, smp, wpp, cm, &flags] { parse_part(b, e, sp, smp, wpp, cm, &flags); }); }
          tg.flush(*b->pool, true);                                     // (ahead of the later chunks' inflate tasks: the coordinator waits for these)
          tg.wait();
          if (const int th = tg.take_thrown()) flags.fetch_or(th == 1 ? 4 : 8); }
        phase(b->tm.wait_parse);
    } else {
    // ---- slow path: stitch the per-task record lists serially; re-walk where the speculation failed ----
    b->tm.slow_chunks++;
    b->recs.clear(); b->rec_contig.clear();
    b->stitch.clear();
    if (!b->carry.empty()) {
        b->stitch = b->carry; b->carry.clear();
        while (b->stitch.size() < 4 && off < usz) b->stitch.push_back(u[off++]);
        if (b->stitch.size() >= 4) {
            uint32_t bs = le32(b->stitch.data());
            size_t need = 4 + (size_t)bs - b->stitch.size();
            if (need <= usz - off) { b->stitch.insert(b->stitch.end(), u + off, u + off + need); off += need; }
            else { b->stitch.insert(b->stitch.end(), u + off, u + usz); off = usz; b->carry.swap(b->stitch); }   // record larger than a chunk
        } else { b->carry.swap(b->stitch); }
    }
    if (!b->stitch.empty()) {
        if (b->stitch.size() < 36) { b->err = "corrupt BAM record"; return XCK_E_IO; }
        const uint8_t* r = b->stitch.data() + 4;
        b->recs.push_back({r, (uint32_t)b->stitch.size() - 4, (int32_t)le32(r), le32(r + 16), le16(r + 12)});
    }
    { size_t pi = 0; bool incomplete = false;
      while (!incomplete) {
          while (pi < c.parts.size() && c.parts[pi].u_end <= off) pi++;
          if (pi == c.parts.size()) break;
          WalkPart& wp = c.parts[pi];
          if (off == wp.spec_start) {                                   // speculation held: take the task's list
              b->recs.insert(b->recs.end(), wp.recs.begin(), wp.recs.end());
              off = wp.stop;
              if (off == wp.u_end) { pi++; continue; }
          }
          while (off < wp.u_end) {                                      // serial walk (record straddles ranges / mis-speculation)
              if (off + 4 > usz) { incomplete = true; break; }
              uint32_t bs = le32(u + off);
              if (off + 4 + (size_t)bs > usz) { incomplete = true; break; }
              if (bs < 32) { b->err = "corrupt BAM record (block_size < 32)"; return XCK_E_IO; }
              const uint8_t* r = u + off + 4;
              b->recs.push_back({r, bs, (int32_t)le32(r), le32(r + 16), le16(r + 12)});
              off += 4 + (size_t)bs;
          }
          pi++;
      } }
    if (off < usz) b->carry.insert(b->carry.end(), u + off, u + usz);
    phase(b->tm.stitch);
    b->rec_contig.resize(b->recs.size());
    for (size_t r = 0; r < b->recs.size(); r++) b->rec_contig[r] = cm(b->recs[r].tid, (int32_t)le32(b->recs[r].p + 4));
    // ---- output layout: prefix sums + batch segmentation ----
    const int64_t nrec = (int64_t)b->recs.size();
    limit = nrec;
    if (o->max_records > 0 && b->n_records + nrec > o->max_records) { limit = std::max<int64_t>(0, o->max_records - b->n_records); }
    hit_limit = limit < nrec;
    b->rec_out.assign(nrec, -1);
    size_t n_out = 0, n_cig = 0, n_seq = 0;
    for (int64_t r = 0; r < limit; r++) if (b->rec_contig[r] >= 0) { n_out++; n_cig += b->recs[r].n_cig; n_seq += (b->recs[r].l_seq + 1) / 2; }
    if (n_cig >= (size_t(1) << 32) || n_seq >= (size_t(1) << 32)) { b->err = "chunk too large"; return XCK_E_IO; }
    if (!soa_reserve(s, n_out + 1, n_cig + 1, e->dec.want_seq ? n_seq + 1 : 1)) { b->err = "out of host memory"; return XCK_E_NOMEM; }
    { size_t oi = 0; uint32_t co = 0, so = 0; int32_t cur_c = -1; int64_t seg0 = 0, seg_r = 0;
      s.cig_off[0] = 0; s.seq_off[0] = 0;
      for (int64_t r = 0; r < limit; r++) {
          int32_t ctg = b->rec_contig[r];
          if (ctg != cur_c) {                                     // a run of records on one contig = one batch
              if (cur_c >= 0 && (int64_t)oi > seg0) b->pending.push_back({cur_c, seg0, (int64_t)oi, ord_hi | (uint64_t)(b->n_records + seg_r)});
              cur_c = ctg; seg0 = (int64_t)oi; seg_r = r;
          }
          if (ctg < 0) continue;
          b->rec_out[r] = (int64_t)oi;
          co += b->recs[r].n_cig; so += e->dec.want_seq ? (b->recs[r].l_seq + 1) / 2 : 0; oi++;
          s.cig_off[oi] = co; s.seq_off[oi] = so;
      }
      if (cur_c >= 0 && (int64_t)oi > seg0) b->pending.push_back({cur_c, seg0, (int64_t)oi, ord_hi | (uint64_t)(b->n_records + seg_r)});
    }
    phase(b->tm.layout);
    // ---- parse (parallel) ----
    { TaskGroup tg; const int64_t per = std::max<int64_t>(4096, (limit + b->n_threads * 4 - 1) / (b->n_threads * 4));
      for (int64_t r0 = 0; r0 < limit; r0 += per) { int64_t r1 = std::min(limit, r0 + per); int32_t smp = o->sample;
          tg.later([b, e, smp, r0, r1, &flags] { parse_range(b, e, smp, r0, r1, &flags); }); }
      tg.flush(*b->pool, true);
      tg.wait();
      if (const int th = tg.take_thrown()) flags.fetch_or(th == 1 ? 4 : 8); }
    phase(b->tm.wait_parse);
    }
    if (flags.load() & 4) { b->err = "out of host memory (record parse)"; return XCK_E_NOMEM; }
    if (flags.load() & 8) { b->err = "C++ exception in a parse task"; return XCK_E_IO; }
    if (flags.load() & 1) { b->err = "corrupt BAM record (fields exceed block_size)"; return XCK_E_IO; }
    if (flags.load() & 2) { b->err = "too many distinct non-ACGT keys for the key width"; return XCK_E_CAPACITY; }
    if (b->per_tid_ranges && cm.t_end) {                               // last record of the chunk starts beyond its reference's window: drop the rest
        const RecRef* last = nullptr;
        if (fast) { for (size_t pi = c.parts.size(); pi-- > 0 && !last;) if (!c.parts[pi].recs.empty()) last = &c.parts[pi].recs.back(); }
        else if (!b->recs.empty()) last = &b->recs.back();
        if (last && last->tid >= 0 && last->tid < n_refs && cm.t_end[last->tid] > 0 && (int32_t)le32(last->p + 4) >= cm.t_end[last->tid]) {
            b->skip_range = c.range_id; b->scanner->abandon(c.range_id);
        }
    }
    b->n_records += limit;
    advance_head(b);
    if (hit_limit) { b->done = true; for (auto& cc : b->ch) cc.tg.wait(); }
    return 1;
}

extern "C" {

int xck_bam_ref_records(xck_bam* b, int tid, int64_t* n_mapped, int64_t* n_unmapped) {
    if (!b || tid < 0 || tid >= (int)b->ref_names.size()) return XCK_E_ARG;
    load_index(b);
    if (!b->idx_ok) return XCK_E_IO;
    if (n_mapped) *n_mapped = b->idx_mapped[tid];
    if (n_unmapped) *n_unmapped = b->idx_unmapped[tid];
    return XCK_OK;
}

int xck_bam_linear_index(xck_bam* b, int tid, int64_t* n, const uint64_t** voffsets) {
    if (!b || !n || !voffsets || tid < 0 || tid >= (int)b->ref_names.size()) return XCK_E_ARG;
    load_index(b);
    if (!b->idx_ok) return XCK_E_IO;
    *n = (int64_t)b->idx_lin[tid].size(); *voffsets = b->idx_lin[tid].data();
    return XCK_OK;
}

static int next_batch_impl(xck_engine* e, xck_bam* b, const xck_ingest_opts* o, xck_batch* out) {
    if (!e || !b || !o || !out) return XCK_E_ARG;
    CallerBinding on_node(b);
    b->defer_parse = false;                                            // (this interface hands out finished batches: parse in place)
    while (b->pending.empty()) {
        if (b->done) return 0;
        int rc = decode_next_chunk(e, b, o);
        if (rc < 0) { e->err = b->path + ": " + b->err; return rc; }
        if (rc == 0) { b->done = true; return 0; }
    }
    PendingBatch pb = b->pending.front(); b->pending.pop_front();
    HostSoA& s = b->soa[b->soa_i];
    memset(out, 0, sizeof *out);
    out->contig = pb.contig; out->n_reads = (int32_t)(pb.r1 - pb.r0); out->ordinal_base = pb.ordinal_base;
    out->pos = s.pos + pb.r0; out->flag = s.flag + pb.r0; out->mapq = s.mapq + pb.r0; out->cell = s.cell + pb.r0; out->umi = s.umi + pb.r0;
    out->cig_off = s.cig_off + pb.r0; out->cigar = s.cigar;
    if (e->dec.want_seq) { out->seq_off = s.seq_off + pb.r0; out->seq = s.seq; }
    return 1;
}

// a chunk that was parsed in place (slow path) -> the push thread, behind the chunks already queued there
static int push_chunk(xck_engine* e, xck_bam* b) {
    if (e->n_impl <= 0 || b->pending.empty()) return XCK_OK;    // (a decode-only handle decodes and discards: host-ingest benchmarks)
    const auto t_p = std::chrono::steady_clock::now();
    IngestJob& jb = b->jobs[b->job_n % (PUSH_Q + 1)];
    reap_jobs(b);
    jb.flags.store(0); jb.parsed.store(0); jb.has_parse = false; jb.soa = b->soa_i; jb.ring_idx = -1; jb.released = true;
    jb.pending.swap(b->pending); b->pending.clear();
    b->job_n++;
    const int prc = get_pusher(e, b)->submit(&jb);
    b->tm.wait_push += ns_since(t_p);
    if (prc) b->err = b->pusher->err;
    return prc;
}

// nothing of this reader is in flight any more: parse tasks ended, their chunks pushed (unless something failed), push thread idle
static int drain_ingest(xck_engine* e, xck_bam* b, int rc) {
    (void)e;
    if (b->pusher) { const int prc = b->pusher->wait_idle(); if (rc >= 0 && prc) { b->err = b->pusher->err; rc = prc; } }
    reap_jobs(b);
    return rc;
}

static int ingest_impl(xck_engine* e, xck_bam* b, const xck_ingest_opts* o, int64_t* n_records) {
    if (!e || !b || !o) return XCK_E_ARG;
    CallerBinding on_node(b);
    const int64_t pause = o->struct_size >= offsetof(xck_ingest_opts, pause_records) + sizeof(int64_t) ? o->pause_records : 0;
    const int64_t start = b->n_records;
    b->defer_parse = e->n_impl > 0;
    auto fail = [&](int rc) { rc = drain_ingest(e, b, rc); e->err = b->path + ": " + b->err; return rc; };
    while (!b->done) {
        int rc = decode_next_chunk(e, b, o);
        if (rc < 0) return fail(rc);
        if (rc == 0) { b->done = true; break; }
        if (rc == 1) { if (const int prc = push_chunk(e, b)) return fail(prc); }   // (parsed in place: the slow path, or no deferral)
        b->pending.clear();
        if (pause > 0 && !b->done && b->n_records - start >= pause) {   // chunk boundary: the reader stays positioned
            if (const int drc = drain_ingest(e, b, 0)) return fail(drc);   // (the caller may talk to the engine now: nothing is in flight)
            if (n_records) *n_records = b->n_records;
            return 1;
        }
    }
    if (const int drc = drain_ingest(e, b, 0)) return fail(drc);
    if (n_records) *n_records = b->n_records;
    return XCK_OK;
}

// C++ exceptions (std::bad_alloc from a buffer that a damaged file made huge, ...) must not unwind through the C ABI into
// ctypes: they become error codes here.
#define XCK_GUARD(e_, expr)                                                                              \
    try { return (expr); }                                                                               \
    catch (const std::bad_alloc&) { if (e_) (e_)->err = "out of host memory"; set_thread_error("out of host memory"); return XCK_E_NOMEM; } \
    catch (const std::exception& x) { if (e_) (e_)->err = x.what(); set_thread_error(x.what()); return XCK_E_IO; }                        \
    catch (...) { if (e_) (e_)->err = "unknown C++ exception"; set_thread_error("unknown C++ exception"); return XCK_E_IO; }
int xck_bam_open(const char* path, int n_threads, xck_bam** out, char* err, size_t errlen) {
    xck_engine* none = nullptr;
    try { return bam_open_impl(path, n_threads, out, err, errlen); }
    catch (const std::bad_alloc&) { if (err && errlen) snprintf(err, errlen, "%s: out of host memory", path ? path : "(null)"); set_thread_error("out of host memory"); (void)none; return XCK_E_NOMEM; }
    catch (const std::exception& x) { if (err && errlen) snprintf(err, errlen, "%s: %s", path ? path : "(null)", x.what()); set_thread_error(x.what()); return XCK_E_IO; }
    catch (...) { if (err && errlen) snprintf(err, errlen, "%s: unknown C++ exception", path ? path : "(null)"); return XCK_E_IO; }
}
static int prefetch_impl(xck_engine* e, xck_bam* b, const xck_ingest_opts* o) {
    if (!e || !b || !o) return XCK_E_ARG;
    if (b->done || b->started) return XCK_OK;
    CallerBinding on_node(b);
    b->prefetching = true;
    const int rc = decode_next_chunk(e, b, o, true);
    b->prefetching = false;
    if (rc < 0) { e->err = b->path + ": " + b->err; return rc; }
    return XCK_OK;
}
int xck_bam_prefetch(xck_engine* e, xck_bam* b, const xck_ingest_opts* o) { XCK_GUARD(e, prefetch_impl(e, b, o)) }
int xck_bam_next_batch(xck_engine* e, xck_bam* b, const xck_ingest_opts* o, xck_batch* out) { XCK_GUARD(e, next_batch_impl(e, b, o, out)) }
int xck_ingest_bam(xck_engine* e, xck_bam* b, const xck_ingest_opts* o, int64_t* n_records) { XCK_GUARD(e, ingest_impl(e, b, o, n_records)) }

}  // extern "C"
