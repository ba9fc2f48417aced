// cdae_internal.hpp — what cdae_multi.hip (data-parallel exchange, multi-shard handle) needs from cdae_hip.hip.
// Not part of the C ABI.  The kernels stay in ONE translation unit (cdae_hip.hip); this is a narrow accessor surface.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/cdae_hip.h"

namespace cdae_internal {

int fail(const char* fmt, ...);                 // sets cdae_hip_last_error(), returns 1

int device_of(const cdae_hip_t* h);
hipStream_t main_stream(cdae_hip_t* h);
hipStream_t aux_stream(cdae_hip_t* h);         // second stream of the handle (full-output path: the b recurrence); idle in the sampled path
uint64_t num_users(const cdae_hip_t* h);
uint64_t num_items(const cdae_hip_t* h);
uint32_t batch_users(const cdae_hip_t* h);      // min(batch_users, num_users)
bool ready(const cdae_hip_t* h);                // set_interactions has run

// shared block, staged delta (`send`) and receive buffer of the pipelined exchange; all compact_count() floats long
// (allocated by the first pipe_stage)
float* send_buf(cdae_hip_t* h);
float* recv_buf(cdae_hip_t* h);
size_t compact_count(const cdae_hip_t* h);

// 0.5 * lambda * (|Wu|^2) of this handle's users (0 when !user_factor): the private part of penalty_loss
int private_penalty(cdae_hip_t* h, double* out);
// 0.5 * lambda * (|W|^2 + |V|^2 + |b|^2 + |b'|^2): the shared part
int shared_penalty(cdae_hip_t* h, double* out);

// exchange state owned by cdae_multi.hip, destroyed with the handle
void*& exchange_slot(cdae_hip_t* h);
void set_exchange_deleter(cdae_hip_t* h, void (*deleter)(void*));

}  // namespace cdae_internal
