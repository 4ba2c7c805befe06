// Internal: device-wide exclusive scan used by the binning stage and the radix sort.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

size_t gs_scan_ws_bytes(long long n);
// gather != nullptr: scans in[gather[i]] instead of in[i] and leaves in[gather[i]] in gathered[i]
int gs_exclusive_scan_u32(const uint32_t* in, uint32_t* out, long long n, uint32_t* total_out,
                          const uint32_t* gather, uint32_t* gathered, void* ws, hipStream_t stream);
