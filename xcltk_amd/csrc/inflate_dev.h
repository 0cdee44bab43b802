// inflate_dev.h - GPU raw-DEFLATE decoder for BGZF blocks (csrc/inflate_dev.hip): one wave per block.
#pragma once
#include <cstdint>
#include <hip/hip_runtime_api.h>

namespace xck {

struct DevBlock { uint32_t in_off, in_len, out_off, out_len; };   // byte ranges of one block's deflate stream / inflated data inside the chunk buffers

// Enqueue the inflate of n_blocks blocks: d_in (compressed bytes of the chunk), d_out (inflated bytes), d_status[b] = 0 when block b
// was inflated to exactly out_len bytes, non-zero when the block is left to the host decoder.  Returns 0 / -1 (launch error).
void dev_inflate_read_prof(unsigned long long out[8]);   // variant 10 (phase clocks): cycles per phase, summed over the blocks
int dev_inflate_launch(hipStream_t stream, const uint8_t* d_in, const DevBlock* d_blocks, int n_blocks, uint8_t* d_out, int32_t* d_status, uint8_t* host_out = nullptr, int variant = 0);
// host_out: device alias of a mapped host block that receives every finished block's bytes too; variant: 0 = the kernel, 10 = the same with
// phase clocks (bench harness only, dev_inflate_read_prof)

// One chunk of BGZF blocks in flight on the GPU (the host decoder keeps a ring of these, csrc/bam.cpp): pinned host buffers for the
// compressed bytes, the block table, the inflated bytes and the per-block status; device twins; a stream and an event of its own,
// so that the chunks of a ring overlap on the device (the decoder needs thousands of blocks in flight to pay: one wave per block).
struct GpuInflateSlot {
    int device = -1; hipStream_t stream = nullptr; hipEvent_t done = nullptr; int variant = 0;   // (see dev_inflate_launch)
    // pinned + mapped host blocks (h_*) with their device aliases (a_*): the kernel reads the compressed bytes and the block table
    // straight from host memory and writes the statuses there; the inflated bytes are produced in HBM (d_out: match copies read
    // them back) and stored to h_out by the wave that made them, block by block - no DMA engine is involved (see gpu_inflate_slot_launch)
    uint8_t *h_in = nullptr, *a_in = nullptr, *h_out = nullptr, *a_out = nullptr, *d_out = nullptr; DevBlock *h_bl = nullptr, *a_bl = nullptr; int32_t *h_st = nullptr, *a_st = nullptr;
    size_t cap_in = 0, cap_out = 0, cap_bl = 0;
};
GpuInflateSlot* gpu_inflate_slot_create(int device, int free_cus, bool verbose);           // nullptr when the device cannot be used; free_cus: CUs its stream never uses
void gpu_inflate_slot_destroy(GpuInflateSlot* s);                                          // waits for whatever is in flight
bool gpu_inflate_slot_reserve(GpuInflateSlot* s, size_t in_bytes, size_t out_bytes, size_t n_blocks);   // (re)allocates; false = out of memory
int  gpu_inflate_slot_launch(GpuInflateSlot* s, size_t in_bytes, size_t out_bytes, size_t n_blocks);    // inflate kernel (every wave also stores its finished block to h_out) + event, asynchronous; 0 = enqueued
bool gpu_inflate_slot_done(GpuInflateSlot* s);                                            // non-blocking: has the last launch finished?
int  gpu_inflate_slot_wait(GpuInflateSlot* s);                                             // 0 = the chunk's bytes and statuses are in h_out / h_st

}  // namespace xck
