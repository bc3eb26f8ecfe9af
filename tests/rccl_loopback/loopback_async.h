// loopback_async.h -- TEST INFRASTRUCTURE: the asynchronous, device-side mode of the loopback transport (rccl_loopback.cpp
// selects it with CAPI_LOOPBACK_MODE=async).  See loopback_async.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <string>

namespace lb_async {

struct AComm;   // one per communicator and process

// `dir`/`name` identify the communicator's control segment (a file all of its ranks map); collective over its ranks
AComm* attach(const std::string& dir, const std::string& name, int rank, int size);
// collective: drains this process's device, meets the peers, closes mappings, frees the rings
void detach(AComm* c);

// stream-ordered, asynchronous on the host: nothing below waits for the device or for a peer's device
// (`accumulate`: the receiver adds the message to `buf` instead of overwriting it: the reductions' building block)
bool send(AComm* c, int peer, const void* buf, size_t bytes, hipStream_t s);
bool recv(AComm* c, int peer, void* buf, size_t bytes, hipStream_t s, bool accumulate);
bool local_copy(void* dst, const void* src, size_t bytes, hipStream_t s, bool accumulate);
// this process's scratch of at least `bytes` (one block per communicator, grown on demand; stream-ordered reuse is the caller's business)
void* scratch(AComm* c, size_t bytes);

}  // namespace lb_async
