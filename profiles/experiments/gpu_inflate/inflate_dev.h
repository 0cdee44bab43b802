// inflate_dev.h - GPU raw-DEFLATE decoder for BGZF blocks (csrc/inflate_dev.hip): one wave per block.
#pragma once
#include <cstdint>
#include <hip/hip_runtime_api.h>

namespace xck {

struct DevBlock { uint32_t in_off, in_len, out_off, out_len; };   // byte ranges of one block's deflate stream / inflated data inside the chunk buffers

// Enqueue the inflate of n_blocks blocks: d_in (compressed bytes of the chunk), d_out (inflated bytes), d_status[b] = 0 when block b
// was inflated to exactly out_len bytes, non-zero when the block is left to the host decoder.  Returns 0 / -1 (launch error).
int dev_inflate_launch(hipStream_t stream, const uint8_t* d_in, const DevBlock* d_blocks, int n_blocks, uint8_t* d_out, int32_t* d_status);

}  // namespace xck
