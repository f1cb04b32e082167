// Internal launch interface between engine.cpp (host runtime, built with g++) and kernels.hip
// (device code, built with hipcc for gfx950).  Not part of the public C ABI.
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stdint.h>

#ifndef GSV_BLOCK_THREADS
#define GSV_BLOCK_THREADS 1024
#endif

namespace gsv {
namespace dev {

// One call of a plan inside a WINDOW launch (schedule.hpp: the calls of a window run as a dataflow, blockIdx.y = call in stream
// order): the call's program image, its gate-id / ciphertext offsets, the base of its own scratch region inside every instance's
// wire file, its wire hand-over lists (global wires -> program inputs before it runs, program outputs -> global wires after) and the
// earlier calls of the window it has to wait for.
struct CallDesc {
  const void* steps;
  const void* ands;
  const void* xors;
  uint64_t gid_off;    // added to KernelArgs::gid_base
  uint64_t ct_off;     // record offset of the call's ciphertext block inside every instance's device stream
  uint32_t w_base;     // first slot of the call's scratch region inside the instance's wire file
  uint32_t n_steps;
  uint32_t pre_off, n_pre;    // KernelArgs::copy_src/copy_dst[pre_off .. +n_pre): absolute slots, globals -> scratch
  uint32_t post_off, n_post;  // scratch -> globals
  uint32_t dep_off, n_deps;   // KernelArgs::deps[dep_off .. +n_deps): window-relative indices (= blockIdx.y) of the calls to wait for
  uint32_t and_terms;         // record form of the call's program: 2 (pack_and) or 4 (pack_and4), program.hpp
  uint32_t pad0_;
  // ciphertext ring (schedule.hpp, SchedParams::ring_ct): the stream position the host's counter (*ct_pos) must have reached before
  // the call may touch its block — garbling: everything its block overwrites has been gathered off the device (0 = nothing to wait
  // for); evaluating: its own segment has been uploaded
  uint64_t ct_need;
  uint64_t ct_ready;
  // Both live in host memory mapped into the device (fine-grained, system scope) and are the same for every launch of the session.
  // They sit here and not in the kernel arguments because the descriptor's address is held across the step loop anyway: as arguments
  // they stayed live in SGPRs and pushed loop-resident scalars into VGPR lanes (58 v_readlane per step in the four-wire kernels).
  const unsigned long long* ct_pos;  // ring mode: the host's stream-position counter, null = no ring
  uint32_t* done_host;               // the call's counter of finished workgroups: how the host follows a RUNNING window (drain segments)
};
static_assert(sizeof(CallDesc) == 112, "CallDesc layout");

struct KernelArgs {
  const void* steps;   // StepDesc[n_steps]   {and_off, and_cnt, xor_off, xor_cnt}
  const void* ands;    // AndRec[]            32 B
  const void* xors;    // XorRec[]            16 B
  uint4* W;            // [n_instances][n_slots] labels
  uint8_t* VB;         // [n_instances][n_slots] plaintext bits (evaluate only)
  uint4* CT;           // [n_instances][ct_stride] ciphertext streams
  const uint4* delta;  // [n_instances] (garble only)
  const uint32_t* te;  // 4*256 T-table words
  const uint32_t* fb_src;
  const uint32_t* fb_dst;
  uint64_t ct_stride;  // records per instance = ct_cap_replays * n_ct
  uint64_t gid_base;   // gate_id of the first gate of replay 0
  uint64_t n_gates;    // gate_ids consumed per replay
  uint64_t n_ct;       // ciphertexts per replay
  uint64_t ct_offset;  // first record of this launch inside every instance's stream (plans: the call's block)
  uint32_t n_steps;
  uint32_t n_slots;
  uint32_t replays;
  uint32_t rep_base;          // index of this launch's first replay inside the instance's whole stream (segmented launches)
  uint32_t ct_cap_replays;
  uint32_t n_fb;
  uint32_t fb_stage_base;
  uint32_t n_instances;
  uint32_t instances_per_wg;  // 1, 2 or 4 (n needs a program compiled for 1/n of the LDS window)
  uint32_t hasher;            // 0 = AesNiHasher, 1 = Blake3Hasher
  uint32_t and_terms;         // record form of the program (program launches; window launches take it from the call descriptor): 2 or 4
  uint32_t any_four_wire;     // host-side kernel choice: some program of this launch is in the four-wire form (the FW instantiations)
  unsigned long long* step_clock;  // diagnostics: workgroup 0 stamps the 100 MHz wall clock at the start of every step of the last replay (null = off)
  const CallDesc* calls;  // non-null: window launch, grid.y = calls; steps / ands / xors / n_steps / ct_offset come from calls[blockIdx.y]
  const uint32_t* copy_src;   // wire hand-over lists of the session (absolute slots inside an instance's wire file)
  const uint32_t* copy_dst;
  const uint32_t* deps;
  uint32_t* flags;            // [gridDim.x][flag_stride] completion flags: flags[x][c] == epoch once call c has finished for instance group x;
                              // flags[x][flag_stride - 1] counts the calls group x has completed (the dependency watchdog's progress counter)
  uint32_t* error;            // set to 1 when a dependency wait gives up (never expected: see schedule.hpp), to 2 when a ciphertext-ring wait does
  uint32_t flag_stride;
  uint32_t epoch;             // launch counter of the session: flags are never reset
  unsigned long long wait_ticks;  // dependency watchdog: give up when the group's progress counter has not moved for this long (100 MHz ticks)
  uint32_t diag;  // timing experiments only (GSV_DIAG env; honoured by a library built with -DGSV_DIAG_BUILD = `build.py --diag`, ignored by
                  // the production build): 1 = skip AES, 4 = skip label loads, 8 = skip stores, 16 = no multi-lane narrow form,
                  // 32 = no step barrier, 64 = no record prefetch, 128 = no load of an AND record's second half
};

}  // namespace dev
}  // namespace gsv

extern "C" {
int gsvk_upload_round_keys(const uint32_t rk[44]);
int gsvk_launch_program(const gsv::dev::KernelArgs* ka, uint32_t n_instances, int evaluate, hipStream_t stream);
// grid = (instance groups, n_calls): ka->calls[0 .. n_calls) run as a dataflow (n_calls <= 65535)
int gsvk_launch_batch(const gsv::dev::KernelArgs* ka, uint32_t n_instances, uint32_t n_calls, int evaluate, hipStream_t stream);
int gsvk_gather_outputs(const void* W, const void* VB, uint32_t n_slots, const uint32_t* slots, uint32_t n_out, uint32_t n_instances,
                        void* out, void* out_bits, hipStream_t stream);
// stage[i] <-> stream[(idx / n_ct) * n_ct + ct_pos[idx % n_ct]] for idx = first + i, i < n  (scatter != 0: stage -> stream)
int gsvk_permute_ciphertexts(void* stream, const void* ct_pos, uint64_t n_ct, uint64_t first, uint64_t n, void* stage, int scatter, hipStream_t s);
// out[inst][r*n_ct + g] = ring[inst][r*n_ct + ct_pos[g]] for r < n_rep, every instance (strides in 16-byte records); scatter != 0: the other way
int gsvk_gather_segment(void* ring, uint64_t ring_stride, const void* ct_pos, uint64_t n_ct, uint32_t n_rep, uint32_t n_instances, void* out,
                        uint64_t out_stride, int scatter, hipStream_t s);
// W[inst][dst[i]] = W[inst][src[i]] (and the plaintext bits when VB != null) for every instance: wire hand-over between the calls of a plan
// word[0] = 0 before the call; a one-thread kernel on `first` waits up to `ticks` (100 MHz) for the one-thread kernel on `second` to set it
// and leaves word[1] = 1 (the two streams run side by side) or 2 (the second launch waited behind the first: one hardware queue)
int gsvk_probe_overlap(void* word, unsigned long long ticks, hipStream_t first, hipStream_t second);
int gsvk_copy_slots(void* W, void* VB, uint32_t n_slots, const uint32_t* src, const uint32_t* dst, uint32_t n, uint32_t n_instances, hipStream_t s);
int gsvk_scatter_bits(void* VB, uint32_t n_slots, uint32_t first_slot, const void* bits, uint32_t n, uint32_t n_instances, hipStream_t stream);
}
