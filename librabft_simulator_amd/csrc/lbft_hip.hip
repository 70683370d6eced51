// lbft_hip.hip -- HIP kernels (gfx950) and the C ABI of include/lbft.h.
//
// Execution model: one lane = one simulated network; a wavefront advances up to 32 (large batches) independent
// discrete-event simulations, two wavefronts per SIMD.  The simulation step itself is lbft_core.h (device build only
// in this library), instantiated per network-size class; this file holds the kernels around it, the LDS / launch
// geometry, and the host side of the C ABI.
// No CPU fallback exists: every entry point fails with LBFT_ERR_HIP if the device is unusable.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lbft.h"
#include "lbft_core.h"
#include "lbft_save_node.h"
#include "lbft_tables.h"

using namespace lbft;

static_assert(LBFT_MAX_NODES == LBFT_MAX_NODES_SUPPORTED, "header mismatch");

// ------------------------------------------------------------------------------------------------
// Kernels
// ------------------------------------------------------------------------------------------------
#define LBFT_BLOCK 64  // one wavefront per workgroup: wavefronts retire independently

// Simulator::new for every instance (simulator.rs:200-250).
__global__ __launch_bounds__(LBFT_BLOCK) void lbft_k_init(Params p, u32* __restrict__ state, const u64* __restrict__ seeds) {
  u32 i = blockIdx.x * p.lpw + threadIdx.x;
  if (threadIdx.x >= p.lpw || i >= p.m) return;
  Sim s(p, state, i);
  s.init(seeds[i]);
}

// Simulator::loop_until for every instance (simulator.rs:380-475).
//
// Workgroup = LBFT_RUN_WAVES (8) wavefronts = both wavefront slots of a CU's four SIMDs (256 registers per lane: two per SIMD);
// wavefront w of workgroup g advances instances [(g * 8 + w) * lpw, +lpw).  LDS (dynamic, up to the CU's whole 160 KiB):
//   [zig_x 257][zig_f 257][exp_tab 256]  u64   read-only tables of the delay sampler, one copy per workgroup
//   [dur 128] i64, [leader 1024] u8            pacemaker duration / leader tables (first rounds)
//   keys  [wave][slot][lane]              u64   event-queue keys, lane-private columns: class 0 one packed word per event
//                                               (time | 3-kind | stamp | node | sender | slot), other classes (time, 3-kind, stamp)
//   metas [wave][slot][lane]              u32   classes 1-2 only: (node, sender, snapshot slot)
//   [diagnostic phase counters], then      --   n > 16: one 128-byte receiver list per instance; class 0 with n <= 4: the nodes'
//                                               hcbr buffers, 32 words per instance ([wave][word][lane])
// A lane only ever touches its own column (address = slot * lpw + lane), so data-dependent slot
// indices are bank-conflict free and no workgroup barrier is needed after the table fill.
// (round 3: 8 instead of 4.  Equal for every batch that fills the chip -- 65 536 x 4: 21.61 vs 21.67 ms, 32 768: 18.4 vs 18.6, 16 384: 15.7 vs
// 15.5 -- but a batch of <= 1 024 networks then packs two wavefronts on every SIMD of half the CUs instead of one on each SIMD of all
// of them, and a wavefront that shares its SIMD runs FASTER per step (phase timers, one network per wavefront: 13.2 k cycles per step
// with a partner, 17.3 k alone): 1 024 x 4 nodes 9.8 -> 7.0 ms.)
#ifndef LBFT_RUN_WAVES
#define LBFT_RUN_WAVES 8
#endif
#define LBFT_RUN_BLOCK (64 * LBFT_RUN_WAVES)
#define LBFT_RUN_WAVES_FULL 4  // the kernels that use the whole register file (lbft_k_run<1>, <2>): one wavefront per SIMD = 4 per workgroup
                               // (a 512-thread launch bound would cap them at 256 registers)
#define LBFT_LDS_HCBR_WORDS 32  // class 0, n <= 4: hcbr[node][2][4] per instance
#ifndef LBFT_PACKED_QL_MAX
#define LBFT_PACKED_QL_MAX 64  // LDS slots per instance of the packed (class 0) queue: the 4-node bench workload peaks at 53 pending events
#endif
#define LBFT_LDS_LEADERS 1024  // rounds of the leader table kept in LDS (bytes)
#define LBFT_LDS_DURS 128      // entries of the duration table kept in LDS (i64)
#define LBFT_LDS_WEIGHTS LBFT_MAX_NODES  // voting rights (u32)
#define LBFT_TABLE_U64 (257 + 257 + 256 + LBFT_LDS_DURS + LBFT_LDS_LEADERS / 8 + LBFT_LDS_WEIGHTS / 2)

// [tables][queue keys][queue metas][diagnostics: LBFT_NPHASES u64 per wavefront][n > 16: one 128-byte receiver list per instance]
// `slot_bytes`: 12 (key + meta) or 8 (packed one-word entries, kernel class 0)
// `hcbr_lds`: class 0 with networks of <= 4 nodes keeps the nodes' hcbr buffers in LDS -- except lbft_k_run0q, which carries them in
// registers with the node burst (LBFT_C0_HCREG)
// the LDS window of block records of the large-network kernels (SimT::attach_blk_window): `entries` records + tags per network
static inline size_t blk_window_bytes(u32 entries, u32 lpw, u32 nwaves) { return (size_t)nwaves * lpw * entries * (1u + BC_WORDS) * 4u; }
static inline size_t run_lds_bytes(u32 ql, u32 lpw, u32 n, u32 slot_bytes, u32 nwaves, bool hcbr_lds = true) {
  return (size_t)LBFT_TABLE_U64 * 8 + (size_t)nwaves * ql * lpw * slot_bytes + (size_t)nwaves * LBFT_NPHASES * 8 + 8 +
         (n > 16 ? (size_t)nwaves * lpw * LBFT_MAX_NODES : 0) +
         (n <= 4 && slot_bytes == 8 && hcbr_lds ? (size_t)nwaves * lpw * LBFT_LDS_HCBR_WORDS * 4 : 0);  // class 0, n <= 4: hcbr buffers
}

__device__ __forceinline__ size_t run_lds_bytes_dev(u32 ql, u32 lpw, u32 slot_bytes, u32 nwaves) {  // = run_lds_bytes(ql, lpw, 0, ..): where the receiver lists start
  return (size_t)LBFT_TABLE_U64 * 8 + (size_t)nwaves * ql * lpw * slot_bytes + (size_t)nwaves * LBFT_NPHASES * 8 + 8;
}
#ifndef LBFT_RUN_WAVES_PER_SIMD
#define LBFT_RUN_WAVES_PER_SIMD 2  // register budget of the class-0 run kernel: 512 / 2 = 256 VGPRs + AGPRs per lane (the
                                   // large-network classes run one 8- or 16-lane wavefront per SIMD and may use all 512)
#endif
template <int CLS>
__device__ __forceinline__ void run_body(const Params& p, u32* __restrict__ state, u32* __restrict__ unfinished) {
  extern __shared__ u64 lds[];
  const u32 nwaves = blockDim.x >> 6;  // wavefronts per workgroup: 8 for the two-wavefronts-per-SIMD kernels, 4 for the full-register ones
  u64* t_zx = lds;
  u64* t_zf = lds + 257;
  u64* t_et = lds + 514;
  for (u32 t = threadIdx.x; t < 257; t += blockDim.x) { t_zx[t] = p.zig_x[t]; t_zf[t] = p.zig_f[t]; }
  for (u32 t = threadIdx.x; t < 256; t += blockDim.x) t_et[t] = p.exp_tab[t];
  i64* t_dur = reinterpret_cast<i64*>(lds + 770);
  u8* t_leader = reinterpret_cast<u8*>(lds + 770 + LBFT_LDS_DURS);
  u32 n_dur = p.dur_len < LBFT_LDS_DURS ? p.dur_len : LBFT_LDS_DURS;
  u32 n_leader = p.leader_len < LBFT_LDS_LEADERS ? p.leader_len : LBFT_LDS_LEADERS;
  for (u32 t = threadIdx.x; t < n_dur; t += blockDim.x) t_dur[t] = p.dur_tab[t];
  for (u32 t = threadIdx.x; t < n_leader; t += blockDim.x) t_leader[t] = p.leader_tab[t];
  u32* t_weights = reinterpret_cast<u32*>(lds + 770 + LBFT_LDS_DURS + LBFT_LDS_LEADERS / 8);
  for (u32 t = threadIdx.x; t < p.n; t += blockDim.x) t_weights[t] = p.weights[t];
  __syncthreads();
  u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u32 qslots = p.ql;  // u64 words per instance in the key area
  const u32 qcols = SimT<CLS>::QS32 ? 32u : p.lpw;  // queue columns per wavefront (lbft_k_run0q: always 32, see LBFT_QUAD_STRIDE32)
  u64* keys = lds + LBFT_TABLE_U64 + (size_t)wave * p.ql * qcols + lane;
  u32* metas = reinterpret_cast<u32*>(lds + LBFT_TABLE_U64 + (size_t)nwaves * qslots * qcols) + (size_t)wave * p.ql * qcols + lane;  // (CLS 0: unused, not allocated)
  const u32 meta_words = SimT<CLS>::C0 ? 0u : nwaves * p.ql * p.lpw;
  // Only the first p.lpw lanes of a wavefront carry an instance (occupancy vs lane-utilisation knob).
  u32 i = (blockIdx.x * nwaves + wave) * p.lpw + lane;
  bool active = lane < p.lpw && i < p.m;
  bool done = true;
  // lpw divides 64, so a wavefront's instances share one tile: its base is wavefront-uniform (SGPRs) and
  // every row access is saddr + 32-bit voffset
  // (tile width tw: 64 for the small-network classes -- two 32-lane wavefronts share a tile --, otherwise tw == lpw: one tile per wavefront)
  const u32 tw = SimT<CLS>::TILE64 ? 64u : SimT<CLS>::IMAJOR ? 1u : p.tw;
  u32 tile_idx = __builtin_amdgcn_readfirstlane(((blockIdx.x * nwaves + wave) * p.lpw) / tw);
  char* tile = reinterpret_cast<char*>(state) + (size_t)tile_idx * p.total_words * ((size_t)4 * tw);
  if constexpr (SimT<CLS>::COOP) {
    // Large networks: EVERY lane of the wavefront runs the event loop; the first lpw lanes carry a network each, all 64
    // cooperate on the bulk sends of those networks (SimT::run_coop / coop_bulk).
    // (tw may be narrower than the lanes that carry a network: lane j's instance then sits j / tw tiles behind the wavefront's
    // first tile -- folded into the lane's 32-bit column offset, the tile base stays wavefront-uniform)
    const u32 li = active ? (i - ((blockIdx.x * nwaves + wave) * p.lpw)) : (lane & (p.lpw - 1u));
    SimT<CLS> s(p, tile, (li / tw) * (p.total_words * 4u * tw) + (li & (tw - 1u)) * 4u, 0);
    bool lead = false;
    if (active) lead = s.ld(I_DONE) == 0;
    s.attach_queue(keys, metas, p.lpw, p.ql);
    s.attach_tables(t_zx, t_zf, t_et);
    s.attach_round_tables(t_leader, n_leader, t_dur, n_dur);
    s.attach_weights(t_weights);
    {  // [receiver lists: nwaves * lpw * LBFT_MAX_NODES bytes][block-record windows: lane-private columns per wavefront]
      u8* lists = reinterpret_cast<u8*>(lds) + run_lds_bytes_dev(p.ql, p.lpw, 12u, nwaves);
      u32* win = reinterpret_cast<u32*>(lists + (size_t)nwaves * p.lpw * LBFT_MAX_NODES) + (size_t)wave * p.lpw * p.blw * (1u + BC_WORDS);
      u32 wsh = 0;
      while ((1u << wsh) < p.lpw) wsh++;
      s.attach_blk_window(win + (lane & (p.lpw - 1u)), p.blw, wsh);
      if (lane < p.lpw) s.blw_reset();
    }
    if (lead) {
      u8* lists = reinterpret_cast<u8*>(lds) + run_lds_bytes_dev(p.ql, p.lpw, 12u, nwaves);
      s.attach_peer_list(lists + ((size_t)wave * p.lpw + lane) * LBFT_MAX_NODES);
      s.load_scalars();
      s.queue_to_lds();
    }
#if defined(LBFT_PHASE_TIMERS)
    u64* wprof = reinterpret_cast<u64*>(reinterpret_cast<u32*>(lds + LBFT_TABLE_U64 + (size_t)nwaves * qslots * qcols) +
                                        (size_t)meta_words + (meta_words & 1u)) + wave * LBFT_NPHASES;
    if (lane == 0) { for (int k = 0; k < LBFT_NPHASES; k++) wprof[k] = 0; wprof[31] = __builtin_readcyclecounter(); }
    s.wprof = wprof;
    u64 t_begin = __builtin_readcyclecounter();
#endif
    bool drained = s.run_coop(lead);
    if (lead) {
      done = drained;
      s.queue_from_lds();
      s.store_scalars(done);
    }
#if defined(LBFT_PHASE_TIMERS)
    if (p.prof && lane == 0) {
      for (int k = 0; k < 31; k++) atomicAdd(&p.prof[k], (unsigned long long)s.wprof[k]);
      atomicAdd(&p.prof[31], (unsigned long long)(__builtin_readcyclecounter() - t_begin));
    }
#endif
  } else if constexpr (SimT<CLS>::WUNI) {
    // ONE network per wavefront as wavefront-uniform code (SimT<K_SMALL_UNIFORM>, lbft_k_run0u; p.lpw == 1): nothing below depends on the lane -- the
    // network's index, rows and LDS columns come from the wavefront's index through readfirstlane -- so all 64 lanes run the event loop
    // with the same values and the compiler keeps the protocol logic on the scalar unit (lbft_core.h, SimT::WUNI); only the pop's scan
    // (coop_find) reads per-lane slots.  Stores / LDS writes: the same address and value in every lane.
    const u32 uwave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u32 ui = __builtin_amdgcn_readfirstlane(blockIdx.x * nwaves + uwave);  // (p.lpw == 1: the wavefront's network)
    u64* ukeys = lds + LBFT_TABLE_U64 + (size_t)uwave * p.ql;
    SimT<CLS> s(p, tile, 0u, 0);
    bool lead = false;
    if (ui < p.m) lead = s.ld(I_DONE) == 0;
    s.attach_queue(ukeys, nullptr, 1u, p.ql);
    s.attach_tables(t_zx, t_zf, t_et);
    s.attach_round_tables(t_leader, n_leader, t_dur, n_dur);
    s.attach_weights(t_weights);
    if (p.n <= 4) {  // the nodes' hcbr buffers
      u32* hcb = reinterpret_cast<u32*>(reinterpret_cast<u8*>(lds) + run_lds_bytes_dev(p.ql, 1u, 8u, nwaves));
      s.attach_hcbr(hcb + (size_t)uwave * LBFT_LDS_HCBR_WORDS);
    }
    s.qlen = 0;
    if (lead) {
      s.load_scalars();
      s.queue_to_lds();
      s.hcbr_to_lds();
    }
#if defined(LBFT_PHASE_TIMERS)
    u64* wprof = reinterpret_cast<u64*>(reinterpret_cast<u32*>(lds + LBFT_TABLE_U64 + (size_t)nwaves * qslots * qcols) +
                                        (size_t)meta_words + (meta_words & 1u)) + uwave * LBFT_NPHASES;
    if (lane == 0) { for (int k = 0; k < LBFT_NPHASES; k++) wprof[k] = 0; wprof[31] = __builtin_readcyclecounter(); }
    s.wprof = wprof;
    u64 t_begin = __builtin_readcyclecounter();
#endif
    bool drained = s.run_popc(lead, ukeys);
    if (lead) {
      done = drained;
      s.queue_from_lds();
      s.hcbr_from_lds();
      s.store_scalars(done);
    }
#if defined(LBFT_PHASE_TIMERS)
    if (p.prof && lane == 0) {
      for (int k = 0; k < 31; k++) atomicAdd(&p.prof[k], (unsigned long long)s.wprof[k]);
      atomicAdd(&p.prof[31], (unsigned long long)(__builtin_readcyclecounter() - t_begin));
    }
#endif
    active = lane == 0 && ui < p.m;  // (one report per wavefront below)
  } else if constexpr (SimT<CLS>::POPC) {
    // Class 0 with the wavefront-wide pop (SimT::run_popc): every lane runs the event loop and scans the wavefront's queue columns;
    // the lanes that carry a network execute its events.
    SimT<CLS> s(p, tile, SimT<CLS>::IMAJOR ? lane * (p.total_words * 4u) : (i & (tw - 1u)) * 4u, 0);
    bool lead = false;
    if (active) lead = s.ld(I_DONE) == 0;
    s.attach_queue(keys, metas, p.lpw, p.ql);
    s.attach_tables(t_zx, t_zf, t_et);
    s.attach_round_tables(t_leader, n_leader, t_dur, n_dur);
    s.attach_weights(t_weights);
    if (p.n <= 4 && !SimT<CLS>::HCREG) {  // the nodes' hcbr buffers
      u32* hcb = reinterpret_cast<u32*>(reinterpret_cast<u8*>(lds) + run_lds_bytes_dev(p.ql, p.lpw, 8u, nwaves));
      s.attach_hcbr(hcb + (size_t)wave * LBFT_LDS_HCBR_WORDS * p.lpw + lane);
    }
    s.qlen = 0;
    if (lead) {
      s.load_scalars();
      s.queue_to_lds();
      s.hcbr_to_lds();
    }
#if defined(LBFT_PHASE_TIMERS)
    u64* wprof = reinterpret_cast<u64*>(reinterpret_cast<u32*>(lds + LBFT_TABLE_U64 + (size_t)nwaves * qslots * qcols) +
                                        (size_t)meta_words + (meta_words & 1u)) + wave * LBFT_NPHASES;
    if (lane == 0) { for (int k = 0; k < LBFT_NPHASES; k++) wprof[k] = 0; wprof[31] = __builtin_readcyclecounter(); }
    s.wprof = wprof;
    u64 t_begin = __builtin_readcyclecounter();
#endif
    bool drained = s.run_popc(lead, keys - lane);
    if (lead) {
      done = drained;
      s.queue_from_lds();
      s.hcbr_from_lds();
      s.store_scalars(done);
    }
#if defined(LBFT_PHASE_TIMERS)
    if (p.prof && lane == 0) {
      for (int k = 0; k < 31; k++) atomicAdd(&p.prof[k], (unsigned long long)s.wprof[k]);
      atomicAdd(&p.prof[31], (unsigned long long)(__builtin_readcyclecounter() - t_begin));
    }
#endif
  } else
  if (active) {
    // (instance-major classes: lane j's instance sits j instances behind the wavefront's first one -- folded into the lane's 32-bit column offset)
    SimT<CLS> s(p, tile, SimT<CLS>::IMAJOR ? lane * (p.total_words * 4u) : (i & (tw - 1u)) * 4u, 0);
    if (s.ld(I_DONE) == 0) {
      s.attach_queue(keys, metas, p.lpw, p.ql);
      s.attach_tables(t_zx, t_zf, t_et);
      s.attach_round_tables(t_leader, n_leader, t_dur, n_dur);
      s.attach_weights(t_weights);
      if (p.n > 16) {  // receiver / sender lists of process_node_actions: LDS instead of HBM rows
        u8* lists = reinterpret_cast<u8*>(lds) + run_lds_bytes_dev(p.ql, p.lpw, SimT<CLS>::C0 ? 8u : 12u, nwaves);
        s.attach_peer_list(lists + ((size_t)wave * p.lpw + lane) * LBFT_MAX_NODES);
      }
      if (SimT<CLS>::C0 && p.n <= 4 && !SimT<CLS>::HCREG) {  // the nodes' hcbr buffers (same place as the receiver lists of large networks)
        u32* hcb = reinterpret_cast<u32*>(reinterpret_cast<u8*>(lds) + run_lds_bytes_dev(p.ql, p.lpw, 8u, nwaves));
        s.attach_hcbr(hcb + (size_t)wave * LBFT_LDS_HCBR_WORDS * p.lpw + lane);
      }
      s.load_scalars();
      s.queue_to_lds();
      s.hcbr_to_lds();
#if defined(LBFT_PHASE_TIMERS)
      // per-wavefront accumulators behind the queue columns (8-byte aligned: the meta area is a multiple of 8 words)
      u64* wprof = reinterpret_cast<u64*>(reinterpret_cast<u32*>(lds + LBFT_TABLE_U64 + (size_t)nwaves * qslots * qcols) +
                                          (size_t)meta_words + (meta_words & 1u)) + wave * LBFT_NPHASES;
      if (lane == 0) { for (int k = 0; k < LBFT_NPHASES; k++) wprof[k] = 0; wprof[31] = __builtin_readcyclecounter(); }
      s.wprof = wprof;
      u64 t_begin = __builtin_readcyclecounter();
#endif
      done = s.run();
      s.queue_from_lds();
      s.hcbr_from_lds();
      s.store_scalars(done);
#if defined(LBFT_PHASE_TIMERS)
      // every lane of a wavefront sees the wavefront's clock: the first active lane reports
      if (p.prof && lane == 0) {
        for (int k = 0; k < 31; k++) atomicAdd(&p.prof[k], (unsigned long long)s.wprof[k]);  // 30 = wavefront loop iterations
        atomicAdd(&p.prof[31], (unsigned long long)(__builtin_readcyclecounter() - t_begin));  // total cycles
      }
#endif
    }
  }
  // one atomic per wavefront: ballot of the lanes that still have pending events
  unsigned long long pending = __ballot(active && !done);
  if (pending && lane == (u32)(__ffsll((long long)pending) - 1)) atomicAdd(unfinished, (u32)__popcll(pending));
}
// (register-budget experiments: -DLBFT_DEV_ONLY_CLASS=k compiles the event loop of class k alone -- seconds instead of minutes; never a product build)
#if defined(LBFT_DEV_ONLY_CLASS)
#define LBFT_DEV_ONLY(k) if ((k) != LBFT_DEV_ONLY_CLASS) return;
#else
#define LBFT_DEV_ONLY(k)
#endif
// Class 0 (the headline small-network path) is compiled for two wavefronts per SIMD; classes 1 and 2 run one 8- or 16-lane
// wavefront per SIMD and may use the whole register file (VGPRs + AGPRs).
__global__ __launch_bounds__(LBFT_RUN_BLOCK) __attribute__((amdgpu_waves_per_eu(LBFT_RUN_WAVES_PER_SIMD, LBFT_RUN_WAVES_PER_SIMD)))
void lbft_k_run0(Params p, u32* __restrict__ state, u32* __restrict__ unfinished) { LBFT_DEV_ONLY(K_SMALL) run_body<K_SMALL>(p, state, unfinished); }
// ... the same kernel with the headline network fixed at compile time (4 nodes, unit voting rights, log-normal delays: SimT<K_HEADLINE>, sim_quad())
__global__ __launch_bounds__(LBFT_RUN_BLOCK) __attribute__((amdgpu_waves_per_eu(LBFT_RUN_WAVES_PER_SIMD, LBFT_RUN_WAVES_PER_SIMD)))
void lbft_k_run0q(Params p, u32* __restrict__ state, u32* __restrict__ unfinished) { LBFT_DEV_ONLY(K_HEADLINE) run_body<K_HEADLINE>(p, state, unfinished); }
// ... and for small batches (at most LBFT_POPC_MAX_LPW networks per wavefront): the pop's scan by all 64 lanes (SimT<K_SMALL_WAVE_POP>)
__global__ __launch_bounds__(LBFT_RUN_BLOCK) __attribute__((amdgpu_waves_per_eu(LBFT_RUN_WAVES_PER_SIMD, LBFT_RUN_WAVES_PER_SIMD)))
void lbft_k_run0s(Params p, u32* __restrict__ state, u32* __restrict__ unfinished) { LBFT_DEV_ONLY(K_SMALL_WAVE_POP) run_body<K_SMALL_WAVE_POP>(p, state, unfinished); }
// ... and for ONE network per wavefront (batches of <= 2 048 networks), as wavefront-uniform code on the scalar unit (SimT<K_SMALL_UNIFORM>; round 5, measured: 256 / 1 024 / 2 048 x 4
// networks 4.77 / 4.88 / 4.91 ms against 5.81 / 5.27 / 5.28 ms on lbft_k_run0s; LBFT_NO_UNI=1 falls back to that kernel)
__global__ __launch_bounds__(LBFT_RUN_BLOCK) __attribute__((amdgpu_waves_per_eu(LBFT_RUN_WAVES_PER_SIMD, LBFT_RUN_WAVES_PER_SIMD)))
void lbft_k_run0u(Params p, u32* __restrict__ state, u32* __restrict__ unfinished) { LBFT_DEV_ONLY(K_SMALL_UNIFORM) run_body<K_SMALL_UNIFORM>(p, state, unfinished); }
// Large networks without record exchange / trace / lossy network (sim_lean()): also two wavefronts per SIMD (4 spilled registers)
__global__ __launch_bounds__(LBFT_RUN_BLOCK) __attribute__((amdgpu_waves_per_eu(LBFT_RUN_WAVES_PER_SIMD, LBFT_RUN_WAVES_PER_SIMD)))
void lbft_k_run2l(Params p, u32* __restrict__ state, u32* __restrict__ unfinished) { LBFT_DEV_ONLY(K_LARGE_LEAN) run_body<K_LARGE_LEAN>(p, state, unfinished); }
// ... and with the record exchange of quirks bit 0 (sim_lean_q1(): requests answered by the peer, responses inserted): 24 spilled registers
__global__ __launch_bounds__(LBFT_RUN_BLOCK) __attribute__((amdgpu_waves_per_eu(LBFT_RUN_WAVES_PER_SIMD, LBFT_RUN_WAVES_PER_SIMD)))
void lbft_k_run2q(Params p, u32* __restrict__ state, u32* __restrict__ unfinished) { LBFT_DEV_ONLY(K_LARGE_EXCHANGE) run_body<K_LARGE_EXCHANGE>(p, state, unfinished); }
// ... and class 1 without them (networks of <= 32 nodes with equivocators, a heap / calendar queue, ...): 22 spilled registers
__global__ __launch_bounds__(LBFT_RUN_BLOCK) __attribute__((amdgpu_waves_per_eu(LBFT_RUN_WAVES_PER_SIMD, LBFT_RUN_WAVES_PER_SIMD)))
void lbft_k_run1l(Params p, u32* __restrict__ state, u32* __restrict__ unfinished) { LBFT_DEV_ONLY(K_MID_LEAN) run_body<K_MID_LEAN>(p, state, unfinished); }
#ifndef LBFT_BIG_WAVES_PER_SIMD
#define LBFT_BIG_WAVES_PER_SIMD 1  // classes 1-2: wavefronts per SIMD the kernels are compiled for (1 = the whole register file;
                                   // measured with 2 -- half the lanes per wavefront, 167 spilled registers: 16384 x 64 nodes
                                   // 1.23 s instead of 1.01 s, 8192 x 100 nodes 7.0 s instead of 5.5 s)
#endif
template <int CLS>
__global__ __launch_bounds__(64 * LBFT_RUN_WAVES_FULL)
#if LBFT_BIG_WAVES_PER_SIMD > 1
__attribute__((amdgpu_waves_per_eu(LBFT_BIG_WAVES_PER_SIMD, LBFT_BIG_WAVES_PER_SIMD)))
#endif
void lbft_k_run(Params p, u32* __restrict__ state, u32* __restrict__ unfinished) { LBFT_DEV_ONLY(CLS) run_body<CLS>(p, state, unfinished); }

__device__ __forceinline__ u64 wave_sum(u64 v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ u64 wave_max(u64 v) {
  for (int off = 32; off > 0; off >>= 1) { u64 o = __shfl_down(v, off, 64); v = o > v ? o : v; }
  return v;
}

enum CounterSlot { C_EV0 = 0, C_EV1, C_EV2, C_EV3, C_DRAWS, C_ROUNDS, C_COMMITS, C_SCHED, C_FAULTED, C_NFOLD, C_NUPD, C_MAXQ, C_MAXSNAP, C_MAXBLK, C_WORDS };

// Per-node State hash (simulated_context.rs:51-55) and the batch counters (wavefront shuffle
// reductions, one atomic per wavefront and counter).
#define LBFT_FINAL_WAVES 4  // wavefronts per workgroup of lbft_k_finalize = the nodes of the SAME 64 instances that are hashed side by side
__global__ __launch_bounds__(LBFT_BLOCK * LBFT_FINAL_WAVES) void lbft_k_finalize(Params p, const u32* __restrict__ state, u64* __restrict__ states_out,
                                                                                 unsigned long long* __restrict__ counters) {
  // grid = (instances / 64, nodes / 4), workgroup = 4 wavefronts: lane = instance, wavefront w of block (x, y) hashes node 4 y + w of the block's 64
  // instances.  The nodes of an instance committed (nearly) the same chain, so the four wavefronts of a workgroup read the same block records at the same
  // time on one CU: three of four reads hit its vector cache / the XCD's L2 (round 4 launched one wavefront per (64 instances, node) -- the four readers of a
  // record ran on different XCDs, 1 024 workgroups apart: 0.25 ms; round 5: EXPERIMENTS.md).  The wavefronts of node 0 also reduce the batch counters.
  u32 i = blockIdx.x * LBFT_BLOCK + (threadIdx.x & 63u);
  u32 n = blockIdx.y * LBFT_FINAL_WAVES + (threadIdx.x >> 6);
  if (n >= p.n) return;
  if (i < p.m) {
    Sim s(p, const_cast<u32*>(state), i);
    u32 nc = s.nfm(n, NF_NCOMMITS);
    Sip13 h;
    h.init();
    h.word(nc);
    for (u32 k0 = 0; k0 < nc; k0 += 4) {  // four commits per round trip: their block ids, then their three fields each
      u32 b[4], link[4], cmd[4], tm[4];
#pragma unroll
      for (u32 j = 0; j < 4; j++) b[j] = k0 + j < nc ? s.ld(p.off_log + n * p.lcap + k0 + j) : 0;
#pragma unroll
      for (u32 j = 0; j < 4; j++) {
        link[j] = b[j] ? s.bf(b[j], B_LINK) : 0; cmd[j] = b[j] ? s.bf(b[j], B_CMD) : 0; tm[j] = b[j] ? s.bf(b[j], B_TIME) : 0;
      }
#pragma unroll
      for (u32 j = 0; j < 4; j++)
        if (k0 + j < nc) { h.word(link[j] >> 16); h.word(cmd[j]); h.word((u64)(i64)(i32)tm[j]); }
    }
    states_out[(size_t)i * p.n + n] = h.finish();
  }
  if (n != 0) return;
  u64 c[C_WORDS];
  for (int k = 0; k < C_WORDS; k++) c[k] = 0;
  if (i < p.m) {
    Sim s(p, const_cast<u32*>(state), i);
    u64 min_round = ~0ULL, min_commits = ~0ULL;
    for (u32 q = 0; q < p.n; q++) {
      u64 nc = s.nfm(q, NF_NCOMMITS), ar = s.nfm(q, NF_PM_ROUND);
      min_round = ar < min_round ? ar : min_round;
      min_commits = nc < min_commits ? nc : min_commits;
    }
    c[C_EV0] = s.ld(I_EV0); c[C_EV1] = s.ld(I_EV1); c[C_EV2] = s.ld(I_EV2); c[C_EV3] = s.ld(I_EV3);
    c[C_DRAWS] = s.ld(I_DRAWS); c[C_ROUNDS] = min_round; c[C_COMMITS] = min_commits; c[C_SCHED] = s.ld(I_STAMP);
    c[C_FAULTED] = s.ld(I_FAULT) ? 1 : 0;
    c[C_NFOLD] = s.ld(I_NFOLD); c[C_NUPD] = s.ld(I_NUPD);
    c[C_MAXQ] = s.ld(I_MAXQ); c[C_MAXSNAP] = s.ld(I_MAXSNAP); c[C_MAXBLK] = s.ld(I_NBLOCKS);
  }
  for (int k = 0; k < C_WORDS; k++) {
    bool is_max = k >= C_MAXQ;
    u64 r = is_max ? wave_max(c[k]) : wave_sum(c[k]);
    if ((threadIdx.x & 63) == 0) {
      if (is_max) atomicMax(&counters[k], (unsigned long long)r);
      else atomicAdd(&counters[k], (unsigned long long)r);
    }
  }
}

// out[inst * n + node] = node row `field` (or instance row when node_field == 0xffffffff).
__global__ void lbft_k_gather_node(Params p, const u32* __restrict__ state, u32 field, u32* __restrict__ out) {
  u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.m * p.n) return;
  u32 n = t / p.m, i = t % p.m;  // consecutive lanes -> consecutive instances (coalesced reads)
  out[(size_t)i * p.n + n] = state[word_offset(p, i, p.off_node + n * p.node_words + field)];
}
__global__ void lbft_k_gather_inst(Params p, const u32* __restrict__ state, u32 row, u32* __restrict__ out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.m) return;
  out[i] = state[word_offset(p, i, row)];
}
// Materialise committed histories: out[(inst * n + node) * cap + k].
__global__ void lbft_k_export_histories(Params p, const u32* __restrict__ state, lbft_commit* __restrict__ out, u32 cap,
                                        u32 first_inst, u32 n_inst) {
  u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_inst * p.n) return;
  u32 n = t / n_inst, i = first_inst + t % n_inst;
  Sim s(p, const_cast<u32*>(state), i);
  u32 nc = s.nfm(n, NF_NCOMMITS);
  lbft_commit* o = out + ((size_t)(i - first_inst) * p.n + n) * cap;
  for (u32 k = 0; k < nc && k < cap; k++) {
    u32 b = s.ld(p.off_log + n * p.lcap + k);
    lbft_commit c;
    c.proposer = s.blk_author(b);
    c.index = s.bf(b, B_CMD);
    c.time = (i64)(i32)s.bf(b, B_TIME);
    o[k] = c;
  }
}

// Node-level interface (include/lbft.h lbft_node_*): one lane applies one trait call to one node.
enum NodeOp : u32 { OP_UPDATE = 0, OP_CREATE_NOTIFICATION, OP_HANDLE_NOTIFICATION, OP_RELEASE_NOTIFICATION, OP_VIEW,
                    OP_CREATE_REQUEST, OP_HANDLE_REQUEST, OP_HANDLE_RESPONSE };
__device__ __forceinline__ void node_op_body(const Params& p, u32* __restrict__ state, u32 op, u32 inst, u32 node, u32 arg0, u32 arg1, i64 node_time,
                                             unsigned long long* __restrict__ out) {
  Sim s(p, state, inst);
  s.load_scalars();
  if (op == OP_UPDATE) {
    s.begin_node(node);
    Actions a = s.update_node(node, node_time);
    s.end_node(node);
    out[0] = (unsigned long long)a.next;
    out[1] = out[2] = 0;
    if (a.send_to >= 0) out[1 + (a.send_to >> 6)] = 1ULL << (a.send_to & 63);
    out[3] = a.broadcast ? 1 : 0;
    out[4] = a.query_all ? 1 : 0;
  } else if (op == OP_CREATE_NOTIFICATION) {
    s.begin_node(node);
    i32 slot = s.snap_alloc();
    if (slot >= 0) { s.write_snapshot(node, (u32)slot); s.snap_set_refs((u32)slot, 1, s.nf(node, NF_EPOCH)); }
    out[0] = (unsigned long long)(long long)slot;
  } else if (op == OP_HANDLE_NOTIFICATION) {
    s.begin_node(node);
    auto sn = s.load_snapshot(arg1);
    bool sync = s.handle_notification(node, arg0, arg1, sn);
    s.end_node(node);
    out[0] = sync ? 1 : 0;
  } else if (op == OP_RELEASE_NOTIFICATION) {
    s.snap_release(arg1);
  } else if (op == OP_CREATE_REQUEST) {  // DataSyncNode::create_request (data_sync.rs:66-71,179-181): epoch + the chains' heads
    s.begin_node(node);
    i32 slot = s.make_request_slot(s.nf(node, NF_EPOCH), s.nf(node, NF_HCC_BLK) | (s.nf(node, NF_HQC_BLK) << 16));
    if (slot >= 0) s.snap_set_refs((u32)slot, 1, s.nf(node, NF_EPOCH));
    out[0] = (unsigned long long)(long long)slot;
  } else if (op == OP_HANDLE_REQUEST) {  // DataSyncNode::handle_request on `node` (data_sync.rs:183-207): its store now + the request
    s.begin_node(node);
    u32 qb = s.sfw(arg1, 0);
    u32 req_epoch = s.ld(qb + S_EPOCH), req_certs = s.ld(qb + S_CERTS);
    if (s.refpack()) req_epoch &= 0xffffu;  // (large networks: the slot's reference count rides in the upper half of this word)
    i32 slot = s.snap_alloc();
    if (slot >= 0) {
      u32 rb = s.sfw((u32)slot, 0);
      s.write_store_snapshot(node, rb);
      s.st(s.sqw(rb, 0), req_epoch); s.st(s.sqw(rb, 1), req_certs);
      s.snap_set_refs((u32)slot, 1, s.nf(node, NF_EPOCH));
    }
    out[0] = (unsigned long long)(long long)slot;
  } else if (op == OP_HANDLE_RESPONSE) {  // DataSyncNode::handle_response(response from peer arg0, clock) (data_sync.rs:209-240)
    s.begin_node(node);
    s.handle_response(node, arg0, arg1, node_time);
    s.end_node(node);
  } else {  // OP_VIEW
    s.begin_node(node);
    out[0] = s.nf(node, NF_EPOCH); out[1] = s.nf(node, NF_CUR_ROUND); out[2] = s.nf(node, NF_HQC_ROUND);
    out[3] = s.nf(node, NF_HTC_ROUND); out[4] = s.nf(node, NF_HC_ROUND); out[5] = s.nf(node, NF_PM_ROUND);
    out[6] = s.nf(node, NF_LVR); out[7] = s.nf(node, NF_LOCKED); out[8] = s.nf(node, NF_NCOMMITS);
    u32 leader = s.nf(node, NF_PM_LEADER);
    out[9] = leader == LBFT_NO_LEADER ? 0xffffffffULL : leader;
    out[10] = s.nf(node, NF_ELECTION) & 0xff;
    u32 nt = 0, nv = 0;
    for (u32 k = 0; k < p.mw; k++) {
      nt += (u32)__popc(s.am_word(node, NF_TO_MASK, k));
      nv += (u32)__popc(s.am_word(node, NF_BAL0_AUTHORS, k)) + (u32)__popc(s.am_word(node, NF_BAL1_AUTHORS, k));
    }
    out[11] = nt; out[12] = nv;
    out[13] = s.nf(node, NF_PROPOSED_BLK) ? 1 : 0;
    out[14] = s.nf(node, NF_HTC_ROUND) ? 1 : 0;
  }
  s.store_scalars(s.ld(I_DONE) != 0);
}
__global__ void lbft_k_node_op(Params p, u32* __restrict__ state, u32 op, u32 inst, u32 node, u32 arg0, u32 arg1, i64 node_time,
                               unsigned long long* __restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  node_op_body(p, state, op, inst, node, arg0, arg1, node_time, out);
}
// The same trait calls for MANY instances in one launch (lbft_node_calls): lane t applies calls[t] -- each on another instance, so the
// calls are independent -- and leaves its 16 result words at out + 16 t.
__global__ __launch_bounds__(64) void lbft_k_node_ops(Params p, u32* __restrict__ state, const lbft_node_call* __restrict__ calls, u32 n,
                                                       unsigned long long* __restrict__ out) {
  u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  lbft_node_call c = calls[t];
  node_op_body(p, state, c.op, c.instance, c.node, c.peer, c.handle, c.node_time, out + (size_t)16 * t);
}

// out[shift * len + r] = PacemakerState::leader(r) under the voting rights shifted by `shift` (blockIdx.y; one table when the
// rights do not rotate)
__global__ void lbft_k_fill_leaders(Params p, u8* __restrict__ out, u32 len) {
  u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  u32 shift = blockIdx.y;
  if (r < len) out[(size_t)shift * len + r] = (u8)compute_leader(p.weights, p.n, p.total_votes, r, shift);
}
__global__ void lbft_k_sample_delays(Params p, u64 seed, i64* __restrict__ out, u32 n) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  u32 dummy[I_WORDS];
  Sim s(p, dummy, 0);
  s.rng.seed(seed);
  for (u32 k = 0; k < n; k++) out[k] = s.sample_delay();
}
__global__ void lbft_k_exp_log(const u64* __restrict__ exp_tab, const double* __restrict__ x, double* __restrict__ e, double* __restrict__ l, u32 n) {
  u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  e[t] = lbft_exp(x[t], exp_tab);
  l[t] = lbft_log(x[t]);
}

// ------------------------------------------------------------------------------------------------
// Host side of the C ABI
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
// Kernel selection.  Large networks without record exchange / round trace / message loss run the two-wavefronts-per-SIMD kernel
// (`lbft_k_run2l`, 8 lanes per wavefront at 16 384 networks) WITHOUT the register-staged node sets and the calendar fetch-ahead: two
// wavefronts overlap each other's dependent round trips, and at 256 registers every staged word is a spilled one (16 384 x 64 nodes:
// 451 ms; with the staging 476-511 ms; the full-register kernel with twice the lanes 503 ms; 8 192 x 100 nodes: 2.49 / 2.75-2.85 / 2.85 s).
// Tuning knobs: LBFT_NO_LEAN=1 = always the full-register kernels; LBFT_LEAN2=0 = the full-register kernel for large networks.
// class-0 batches with few networks per wavefront run lbft_k_run0s (wavefront-wide pop); LBFT_NO_POPC=1: lbft_k_run0 for every batch size
// (one reading of LBFT_NO_UNI for both kernel choices it touches -- round-5 advisor: "LBFT_NO_UNI=0" used to count as set in one of them)
static bool uni_allowed() { const char* e = getenv("LBFT_NO_UNI"); return !(e && atoi(e)); }
static bool quad_eligible(const Params& p) {
  const char* e = getenv("LBFT_NO_QUAD");
  // (its LDS queue columns are 32 lanes apart at compile time, LBFT_QUAD_STRIDE32: 64 networks per wavefront -- batches beyond 131 072
  // networks, or a forced lanes_per_wavefront -- run the generic class-0 kernel)
  // (... except the smallest batches: with one network per wavefront and fewer than 1 024 of them lbft_k_run0u wins -- 256 x 4: 4.77 ms against 7.24 here and
  // 5.92 on lbft_k_run0s; at 1 024 the two tie (4.89 / 4.91), at 2 048 this kernel leads again: profiles/r05/lds_resident_instance_ab.txt)
  // (LBFT_NO_UNI=1 sends those tiny batches to lbft_k_run0s -- the kernel lbft_k_run0u replaced there --, not back here)
  const bool tiny = p.lpw == 1 && p.m < 1024;
  return LBFT_C0_QUAD && sim_quad(p) && !tiny && !(LBFT_QUAD_STRIDE32 && p.lpw > 32) && !(e && atoi(e));
}
// (round 5: since round 4's work on lbft_k_run0q the lane-private kernel beats the wavefront-wide pop at EVERY batch size of the headline network --
// 2 048 / 4 096 / 8 192 / 16 384 x 4: 4.78 / 6.48 / 8.23 / 10.36 ms against 4.90 / 7.41 / 9.42 / 12.19 -- so lbft_k_run0s / lbft_k_run0u now serve the
// class-0 batches that kernel does not take: other sizes, weighted rights, uniform delays -- 4 096 / 8 192 / 16 384 x 4 uniform: 7.23 / 9.40 / 11.94 ms against
// 8.37 / 10.82 / 13.85 on lbft_k_run0; 1 024 x 4 uniform on lbft_k_run0u: 4.71 against 5.11)
static bool small_batch_kernel(const Params& p) {
  const char* e = getenv("LBFT_NO_POPC");
  return LBFT_C0_POPC && sim_class(p) == K_SMALL && p.lpw <= LBFT_POPC_MAX_LPW && !quad_eligible(p) && !(e && atoi(e));
}
// ... and large batches of the headline network (4 nodes, unit rights, log-normal delays) lbft_k_run0q; LBFT_NO_QUAD=1: lbft_k_run0
// LBFT_BLK_WINDOW=n: entries (a power of two, default 32; 0 = off) of the large-network kernels' LDS window of block records
static u32 blk_window_max() {
  const char* e = getenv("LBFT_BLK_WINDOW");
  u32 v = e ? (u32)atoi(e) : 32u;
  while (v & (v - 1)) v &= v - 1;  // round down to a power of two
  return v > 256u ? 256u : v;
}
static bool blk_window_allowed() { return blk_window_max() != 0; }
static bool quad_kernel(const Params& p) { return quad_eligible(p); }
// ... and among them the batches with ONE network per wavefront lbft_k_run0u (wavefront-uniform code on the scalar unit); LBFT_NO_UNI=1: lbft_k_run0s
static bool uni_kernel(const Params& p) { return small_batch_kernel(p) && p.lpw == 1 && uni_allowed(); }
static bool lean_allowed() { const char* e = getenv("LBFT_NO_LEAN"); return !(e && atoi(e)); }
static bool lean2_allowed() { const char* e = getenv("LBFT_LEAN2"); return lean_allowed() && !(e && !atoi(e)); }

static int hip_fail(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return LBFT_ERR_HIP;
}
#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t _e = (expr);                            \
    if (_e != hipSuccess) return hip_fail(_e, #expr);  \
  } while (0)

static const u64 H_ZX[257] = LBFT_ZIG_NORM_X_BITS_INIT;
static const u64 H_ZF[257] = LBFT_ZIG_NORM_F_BITS_INIT;
static const u64 H_ET[256] = LBFT_EXP_TAB_INIT;

#define LBFT_DUR_TABLE_LEN 4096
#define LBFT_LEADER_TABLE_LEN 8192

struct lbft_batch {
  lbft_config cfg;
  std::vector<u64> rights;
  size_t m = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
  u64* d_seeds = nullptr;
  u32* d_state = nullptr;
  size_t state_bytes = 0;
  u64 *d_zx = nullptr, *d_zf = nullptr, *d_et = nullptr;
  i64* d_dur = nullptr;
  u8* d_leaders = nullptr;
  u32* d_weights = nullptr;
  std::vector<u32> weights;
  u32* d_unfinished = nullptr;
  u64* d_states_out = nullptr;
  unsigned long long* d_counters = nullptr;
  mutable std::mutex sn_mutex;  // guards the save_node image cache (sn_*): lbft_batch_save_node takes a const batch but fills the cache
  u32* d_scratch = nullptr;  // m * n words for gathers
  lbft_node_call* d_calls = nullptr;  // lbft_node_calls: the calls of one batch and their result words
  unsigned long long* d_call_out = nullptr;
  size_t calls_cap = 0;
  Params p;
  bool ran = false;
  bool manual = false;  // node-level interface active (lbft_batch_manual_begin)
  bool allow_calendar = true;  // tuning / test knob: 0 forces the heap queue where the calendar would be used
  bool started = false; // lbft_batch_run_steps / checkpoint_load: state initialised, event loop not drained yet
  int64_t started_max_clock = 0;
  u64 step_launches = 0;
  // save_node asked twice in a row (size query, then the fill -- what the Python and Rust callers do) builds the image once: the image of
  // the last (instance, node) is kept until anything writes the state rows (`generation` counts those: init, run, node calls, checkpoint load)
  u64 generation = 0;
  mutable u64 sn_generation = ~0ull;
  mutable size_t sn_inst = 0;
  mutable u32 sn_node = 0;
  mutable std::vector<uint8_t> sn_image;
  u32 max_steps = 0;
  u32 lpw = 0;  // 0 = auto
  int ql = -1;  // LDS queue slots per instance; -1 = auto
  u32 rcap = 0; // round-switch trace capacity (rounds per node); 0 = off
  bool keep_stores = false;  // lbft_batch_keep_retired_stores: the record store a node retires at an epoch change is archived in full
  unsigned long long* d_prof = nullptr;
  unsigned long long* h_counters = nullptr;  // pinned: the counter read-back of a run is one asynchronous copy behind the finalize kernel (round 6)
  size_t lds_bytes = 0;
  u32 run_waves = LBFT_RUN_WAVES;  // wavefronts per workgroup of the run kernel this batch uses (prepare_run)
  float init_ms = 0, run_ms = 0;
  lbft_counters counters;
  size_t table_bytes = 0;
};

static int prepare_run(lbft_batch* b, int64_t max_clock);
static int finalize_run(lbft_batch* b, u32 grid_full, u64 launches);
static int launch_run(lbft_batch* b);
static int zero_calendar(lbft_batch* b);

static int fill_params(const lbft_config* cfg, size_t m, Params& p, std::vector<u32>& weights) {
  memset(&p, 0, sizeof(p));
  p.n = cfg->num_nodes;
  p.m = (u32)m;
  p.stride = (u32)((m + 63) / 64 * 64);
  p.tw = 64; p.rsh = 8;
  p.delay_model = cfg->delay_model;
  // RandomDelay::new (simulator.rs:99-106), computed once on the host with the host libm like the reference
  p.mu = std::log(cfg->mean / std::sqrt(1.0 + cfg->variance / (cfg->mean * cfg->mean)));
  p.sigma = std::sqrt(std::log(1.0 + cfg->variance / (cfg->mean * cfg->mean)));
  p.uni_lo = cfg->uniform_lo;
  p.uni_span = (u64)(cfg->uniform_hi - cfg->uniform_lo) + 1;
  p.cpe = cfg->commands_per_epoch;
  p.tci = cfg->target_commit_interval;
  p.lambda = cfg->lambda;
  p.equiv = cfg->equivocate_every;
  p.quirks = cfg->quirks;
  p.drop_ppm = cfg->drop_per_million;
  p.part_size = cfg->partition_size;
  p.part_start = (i32)(cfg->partition_start < 0 ? 0 : (cfg->partition_start > 0x7fffffff ? 0x7fffffff : cfg->partition_start));
  p.part_end = (i32)(cfg->partition_end < 0 ? 0 : (cfg->partition_end > 0x7fffffff ? 0x7fffffff : cfg->partition_end));
  p.rot = cfg->rights_rotation % (cfg->num_nodes ? cfg->num_nodes : 1);
  p.total_votes = 0;
  weights.assign(p.n, 1);
  p.unit_weights = 1;
  for (u32 i = 0; i < p.n; i++) {
    u64 w = cfg->voting_rights ? cfg->voting_rights[i] : 1;
    if (w > 0x00ffffffu) return LBFT_ERR_INVALID;
    weights[i] = (u32)w;
    p.total_votes += (u32)w;
    if (w != 1) p.unit_weights = 0;
  }
  if (p.total_votes == 0) return LBFT_ERR_INVALID;
  if (p.unit_weights) p.rot = 0;  // rotating equal rights changes nothing
  p.mw = (p.n + 31) / 32;
  p.quorum = 2 * p.total_votes / 3 + 1;  // quorum_threshold (configuration.rs:52-56)
  return LBFT_OK;
}

static int validate(const lbft_config* cfg) {
  if (!cfg) return LBFT_ERR_INVALID;
  if (cfg->num_nodes == 0) return LBFT_ERR_INVALID;
  if (cfg->num_nodes > LBFT_MAX_NODES_SUPPORTED) return LBFT_ERR_UNSUPPORTED;
  if ((cfg->quirks & ~3u) != 0) return LBFT_ERR_UNSUPPORTED;
  if (cfg->delay_model > 1) return LBFT_ERR_INVALID;
  if (cfg->delay_model == 0 && !(cfg->mean > 0.0 && cfg->variance >= 0.0)) return LBFT_ERR_INVALID;
  if (cfg->delay_model == 1 && !(cfg->uniform_lo >= 0 && cfg->uniform_hi >= cfg->uniform_lo)) return LBFT_ERR_INVALID;
  if (cfg->commands_per_epoch == 0) return LBFT_ERR_INVALID;
  // NodeConfig (node.rs:76-81): durations and periods are f64 products truncated to i64; NaN / negative parameters have no meaning
  if (!(cfg->gamma >= 0.0) || !(cfg->lambda >= 0.0) || cfg->delta < 0 || cfg->target_commit_interval < 0) return LBFT_ERR_INVALID;
  if (!std::isfinite(cfg->gamma) || !std::isfinite(cfg->lambda) || !std::isfinite(cfg->mean) || !std::isfinite(cfg->variance)) return LBFT_ERR_INVALID;
  return LBFT_OK;
}

static void free_batch(lbft_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  hipFree(b->d_seeds); hipFree(b->d_state); hipFree(b->d_zx); hipFree(b->d_zf); hipFree(b->d_et); hipFree(b->d_dur);
  hipFree(b->d_leaders); hipFree(b->d_weights); hipFree(b->d_prof); hipFree(b->d_unfinished); hipFree(b->d_states_out); hipFree(b->d_counters); hipFree(b->d_scratch); hipFree(b->d_calls); hipFree(b->d_call_out);
  if (b->h_counters) hipHostFree(b->h_counters);
  if (b->ev0) hipEventDestroy(b->ev0);
  if (b->ev1) hipEventDestroy(b->ev1);
  if (b->ev2) hipEventDestroy(b->ev2);
  if (b->stream) hipStreamDestroy(b->stream);
  delete b;
}

static int upload_tables(lbft_batch* b) {
  HIP_TRY(hipMalloc(&b->d_zx, sizeof(H_ZX)));
  HIP_TRY(hipMalloc(&b->d_zf, sizeof(H_ZF)));
  HIP_TRY(hipMalloc(&b->d_et, sizeof(H_ET)));
  HIP_TRY(hipMemcpy(b->d_zx, H_ZX, sizeof(H_ZX), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b->d_zf, H_ZF, sizeof(H_ZF), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b->d_et, H_ET, sizeof(H_ET), hipMemcpyHostToDevice));
  // PacemakerState::duration (pacemaker.rs:111-124): delta * n^gamma with the host libm's pow, as the reference
  std::vector<i64> dur(LBFT_DUR_TABLE_LEN);
  for (size_t k = 0; k < dur.size(); k++) dur[k] = f64_to_i64_sat((double)b->cfg.delta * std::pow((double)k, b->cfg.gamma));
  HIP_TRY(hipMalloc(&b->d_dur, dur.size() * sizeof(i64)));
  HIP_TRY(hipMemcpy(b->d_dur, dur.data(), dur.size() * sizeof(i64), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&b->d_weights, b->weights.size() * sizeof(u32)));
  HIP_TRY(hipMemcpy(b->d_weights, b->weights.data(), b->weights.size() * sizeof(u32), hipMemcpyHostToDevice));
  b->p.weights = b->d_weights;
  u32 leader_tables = b->p.rot ? b->p.n : 1;  // one per shift of the rotating voting rights
  HIP_TRY(hipMalloc(&b->d_leaders, (size_t)leader_tables * LBFT_LEADER_TABLE_LEN));
  b->p.dur_tab = b->d_dur; b->p.dur_len = LBFT_DUR_TABLE_LEN;
  b->p.leader_tab = b->d_leaders; b->p.leader_len = 0;  // table not valid while it is being filled
  b->p.exp_tab = b->d_et; b->p.zig_x = b->d_zx; b->p.zig_f = b->d_zf;
  lbft_k_fill_leaders<<<dim3((LBFT_LEADER_TABLE_LEN + 255) / 256, leader_tables), 256, 0, b->stream>>>(b->p, b->d_leaders, LBFT_LEADER_TABLE_LEN);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(b->stream));
  b->p.leader_len = LBFT_LEADER_TABLE_LEN;
  b->table_bytes = sizeof(H_ZX) + sizeof(H_ZF) + sizeof(H_ET) + dur.size() * sizeof(i64) + (size_t)leader_tables * LBFT_LEADER_TABLE_LEN;
  return LBFT_OK;
}

extern "C" {

const char* lbft_last_error(void) { return g_err.c_str(); }
#if defined(LBFT_PHASE_TIMERS)
const char* lbft_build_info(void) { return "liblbft_hip gfx950 abi4 lane-per-instance lds-queue coop-bulk-send phase-timers"; }
#else
const char* lbft_build_info(void) { return "liblbft_hip gfx950 abi4 lane-per-instance lds-queue coop-bulk-send"; }
#endif

int lbft_batch_create(const lbft_config* cfg, const uint64_t* seeds, size_t n_instances, int device, lbft_batch** out) {
  if (!out) return LBFT_ERR_INVALID;
  *out = nullptr;
  int rc = validate(cfg);
  if (rc != LBFT_OK) { g_err = "invalid or unsupported configuration"; return rc; }
  if (!seeds || n_instances == 0 || n_instances > 0x7fffffffu / cfg->num_nodes) { g_err = "bad seeds / n_instances"; return LBFT_ERR_INVALID; }
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) { g_err = "no such HIP device"; return LBFT_ERR_HIP; }
  HIP_TRY(hipSetDevice(device));
  lbft_batch* b = new lbft_batch();
  b->cfg = *cfg;
  if (cfg->voting_rights) b->rights.assign(cfg->voting_rights, cfg->voting_rights + cfg->num_nodes);
  b->cfg.voting_rights = b->rights.empty() ? nullptr : b->rights.data();
  b->m = n_instances;
  b->device = device;
  rc = fill_params(&b->cfg, n_instances, b->p, b->weights);
  if (rc != LBFT_OK) { delete b; g_err = "bad voting rights"; return rc; }
#define CREATE_TRY(expr)                                                   \
  do {                                                                     \
    hipError_t _e = (expr);                                                \
    if (_e != hipSuccess) { int r_ = hip_fail(_e, #expr); free_batch(b); return r_; } \
  } while (0)
  CREATE_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
  CREATE_TRY(hipEventCreate(&b->ev0));
  CREATE_TRY(hipEventCreate(&b->ev1));
  CREATE_TRY(hipEventCreate(&b->ev2));
  CREATE_TRY(hipMalloc(&b->d_seeds, n_instances * sizeof(u64)));
  CREATE_TRY(hipMemcpy(b->d_seeds, seeds, n_instances * sizeof(u64), hipMemcpyHostToDevice));
  CREATE_TRY(hipMalloc(&b->d_unfinished, sizeof(u32)));
  CREATE_TRY(hipMalloc(&b->d_prof, LBFT_NPHASES * sizeof(unsigned long long)));
  CREATE_TRY(hipMemset(b->d_prof, 0, LBFT_NPHASES * sizeof(unsigned long long)));  // (zeroed per run only in phase-timer builds)
  CREATE_TRY(hipMalloc(&b->d_states_out, n_instances * cfg->num_nodes * sizeof(u64)));
  CREATE_TRY(hipMalloc(&b->d_counters, C_WORDS * sizeof(unsigned long long)));
  CREATE_TRY(hipMalloc(&b->d_scratch, n_instances * cfg->num_nodes * sizeof(u64) + 256));  // + the node-level calls' result words
  rc = upload_tables(b);
  if (rc != LBFT_OK) { free_batch(b); return rc; }
  *out = b;
  return LBFT_OK;
}

int lbft_batch_set_max_steps(lbft_batch* b, uint32_t max_steps) {
  if (!b) return LBFT_ERR_INVALID;
  b->max_steps = max_steps;
  return LBFT_OK;
}

int lbft_batch_set_lanes_per_wavefront(lbft_batch* b, uint32_t lanes) {
  if (!b || lanes > 64 || (lanes && 64 % lanes != 0)) return LBFT_ERR_INVALID;  // a wavefront's instances must share one tile
  b->lpw = lanes;
  return LBFT_OK;
}

int lbft_batch_manual_begin(lbft_batch* b, int64_t max_clock) {
  if (!b) return LBFT_ERR_INVALID;
  if (b->ran) { g_err = "batch already ran; call lbft_batch_reset first"; return LBFT_ERR_STATE; }
  int rc = prepare_run(b, max_clock);
  if (rc != LBFT_OK) return rc;
  u32 grid_init = (u32)((b->m + b->p.lpw - 1) / b->p.lpw);
  { int zrc = zero_calendar(b); if (zrc != LBFT_OK) return zrc; }
  b->generation++;
  lbft_k_init<<<grid_init, LBFT_BLOCK, 0, b->stream>>>(b->p, b->d_state, b->d_seeds);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(b->stream));
  b->manual = true;
  return LBFT_OK;
}

int lbft_batch_manual_finalize(lbft_batch* b) {
  if (!b) return LBFT_ERR_INVALID;
  if (!b->manual) { g_err = "lbft_batch_manual_begin first"; return LBFT_ERR_STATE; }
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipEventRecord(b->ev0, b->stream));
  HIP_TRY(hipEventRecord(b->ev1, b->stream));
  HIP_TRY(hipEventRecord(b->ev2, b->stream));
  return finalize_run(b, (u32)((b->m + LBFT_BLOCK - 1) / LBFT_BLOCK), 0);
}

static int node_op(lbft_batch* b, u32 op, size_t inst, u32 node, u32 arg0, u32 arg1, i64 node_time, unsigned long long* host_out, int n_out) {
  if (!b || inst >= b->m || node >= b->p.n) return LBFT_ERR_INVALID;
  if (!b->manual) { g_err = "lbft_batch_manual_begin first"; return LBFT_ERR_STATE; }
  HIP_TRY(hipSetDevice(b->device));
  unsigned long long* d_out = reinterpret_cast<unsigned long long*>(b->d_scratch);  // 16 result words
  b->generation++;
  lbft_k_node_op<<<1, 64, 0, b->stream>>>(b->p, b->d_state, op, (u32)inst, node, arg0, arg1, node_time, d_out);
  HIP_TRY(hipGetLastError());
  if (n_out) HIP_TRY(hipMemcpyAsync(host_out, d_out, n_out * sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return LBFT_OK;
}

static bool exchange_layout(const lbft_batch* b);
// Many trait calls in ONE launch and ONE synchronisation (a host that drives thousands of simulators -- the Rust `Simulator<GpuNode, ..>`
// of bindings/rust -- is otherwise bound by ~10-20 us of launch + sync per call).  Every call of a batch must address another
// instance (calls on one instance are ordered by the protocol; they go into successive batches).
int lbft_node_calls(lbft_batch* b, const lbft_node_call* calls, size_t n, lbft_node_result* results) {
  if (!b || (n && (!calls || !results))) return LBFT_ERR_INVALID;
  if (!b->manual) { g_err = "lbft_batch_manual_begin first"; return LBFT_ERR_STATE; }
  if (n == 0) return LBFT_OK;
  if (n > b->m) { g_err = "more calls than instances: two calls of a batch would address one instance"; return LBFT_ERR_INVALID; }
  const bool exchange = exchange_layout(b);
  std::vector<uint8_t> seen(b->m, 0);
  std::vector<lbft_node_call> dev(calls, calls + n);
  for (size_t k = 0; k < n; k++) {
    const lbft_node_call& c = calls[k];
    if (c.instance >= b->m || c.node >= b->p.n) { g_err = "call addresses no such instance / node"; return LBFT_ERR_INVALID; }
    if (seen[c.instance]) { g_err = "two calls of one batch address the same instance"; return LBFT_ERR_INVALID; }
    seen[c.instance] = 1;
    u32 op;
    switch (c.op) {
      case LBFT_CALL_UPDATE_NODE: op = OP_UPDATE; break;
      case LBFT_CALL_CREATE_NOTIFICATION: op = OP_CREATE_NOTIFICATION; break;
      case LBFT_CALL_HANDLE_NOTIFICATION: op = OP_HANDLE_NOTIFICATION; if (c.peer >= b->p.n || c.handle >= b->p.scap) return LBFT_ERR_INVALID; break;
      case LBFT_CALL_RELEASE_NOTIFICATION: op = OP_RELEASE_NOTIFICATION; if (c.handle >= b->p.scap) return LBFT_ERR_INVALID; break;
      case LBFT_CALL_CREATE_REQUEST: op = OP_CREATE_REQUEST; break;
      case LBFT_CALL_HANDLE_REQUEST: op = OP_HANDLE_REQUEST; if (c.handle >= b->p.scap) return LBFT_ERR_INVALID; break;
      case LBFT_CALL_HANDLE_RESPONSE: op = OP_HANDLE_RESPONSE; if (c.peer >= b->p.n || c.handle >= b->p.scap) return LBFT_ERR_INVALID; break;
      default: g_err = "unknown call"; return LBFT_ERR_INVALID;
    }
    if (!exchange && (op == OP_CREATE_REQUEST || op == OP_HANDLE_REQUEST || op == OP_HANDLE_RESPONSE)) {
      g_err = "request / response calls of a batch need the record-exchange layout (quirks bit 0); in reference mode use the single calls (payload-free tokens)";
      return LBFT_ERR_UNSUPPORTED;
    }
    dev[k].op = op;
  }
  HIP_TRY(hipSetDevice(b->device));
  const size_t call_bytes = n * sizeof(lbft_node_call), out_bytes = n * 16 * sizeof(unsigned long long);
  if (b->calls_cap < n) {
    if (b->d_calls) { hipFree(b->d_calls); b->d_calls = nullptr; }
    if (b->d_call_out) { hipFree(b->d_call_out); b->d_call_out = nullptr; }
    b->calls_cap = 0;
    size_t cap = n < 1024 ? 1024 : n;
    HIP_TRY(hipMalloc(&b->d_calls, cap * sizeof(lbft_node_call)));
    HIP_TRY(hipMalloc(&b->d_call_out, cap * 16 * sizeof(unsigned long long)));
    b->calls_cap = cap;
  }
  HIP_TRY(hipMemcpyAsync(b->d_calls, dev.data(), call_bytes, hipMemcpyHostToDevice, b->stream));
  b->generation++;
  lbft_k_node_ops<<<(u32)((n + 63) / 64), 64, 0, b->stream>>>(b->p, b->d_state, b->d_calls, (u32)n, b->d_call_out);
  HIP_TRY(hipGetLastError());
  std::vector<unsigned long long> h(n * 16);
  HIP_TRY(hipMemcpyAsync(h.data(), b->d_call_out, out_bytes, hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  int rc = LBFT_OK;
  for (size_t k = 0; k < n; k++) {
    const unsigned long long* o = &h[k * 16];
    lbft_node_result& r = results[k];
    memset(&r, 0, sizeof(r));
    switch (dev[k].op) {
      case OP_UPDATE:
        r.actions.next_scheduled_update = (int64_t)o[0]; r.actions.should_send[0] = o[1]; r.actions.should_send[1] = o[2];
        r.actions.should_broadcast = (uint32_t)o[3]; r.actions.should_query_all = (uint32_t)o[4];
        break;
      case OP_HANDLE_NOTIFICATION: r.should_sync = (uint32_t)o[0]; break;
      case OP_CREATE_NOTIFICATION: case OP_CREATE_REQUEST: case OP_HANDLE_REQUEST:
        if ((long long)o[0] < 0) { r.status = LBFT_ERR_FAULT; rc = LBFT_ERR_FAULT; g_err = "no free snapshot slot (snapshot_capacity) in at least one call"; }
        else r.handle = (uint32_t)o[0];
        break;
      default: break;
    }
  }
  return rc;
}

int lbft_node_update(lbft_batch* b, size_t inst, uint32_t node, int64_t node_time, lbft_actions* out) {
  if (!out) return LBFT_ERR_INVALID;
  unsigned long long h[5];
  int rc = node_op(b, OP_UPDATE, inst, node, 0, 0, node_time, h, 5);
  if (rc != LBFT_OK) return rc;
  out->next_scheduled_update = (int64_t)h[0];
  out->should_send[0] = h[1]; out->should_send[1] = h[2];
  out->should_broadcast = (uint32_t)h[3]; out->should_query_all = (uint32_t)h[4];
  return LBFT_OK;
}
int lbft_node_create_notification(lbft_batch* b, size_t inst, uint32_t node, uint32_t* handle) {
  if (!handle) return LBFT_ERR_INVALID;
  unsigned long long h[1];
  int rc = node_op(b, OP_CREATE_NOTIFICATION, inst, node, 0, 0, 0, h, 1);
  if (rc != LBFT_OK) return rc;
  if ((long long)h[0] < 0) { g_err = "no free notification snapshot (snapshot_capacity)"; return LBFT_ERR_FAULT; }
  *handle = (uint32_t)h[0];
  return LBFT_OK;
}
int lbft_node_handle_notification(lbft_batch* b, size_t inst, uint32_t receiver, uint32_t sender, uint32_t handle, uint32_t* should_sync) {
  if (b && (sender >= b->p.n || handle >= b->p.scap)) return LBFT_ERR_INVALID;
  unsigned long long h[1];
  int rc = node_op(b, OP_HANDLE_NOTIFICATION, inst, receiver, sender, handle, 0, h, 1);
  if (rc != LBFT_OK) return rc;
  if (should_sync) *should_sync = (uint32_t)h[0];
  return LBFT_OK;
}
// The request / response half of DataSyncNode.  Batches created with quirks bit 0 hold real payloads (request words in the
// snapshot slots, archive of retired record stores).  Without it the batch follows the reference simulator, where a request
// is answered by the node that issued it (simulator.rs:446, quirk Q1): such a response can only name records the node
// already holds, so it inserts nothing (the oracle asserts response_inserts == 0 in every quirks-0 run) and the calls
// exchange payload-free TOKENS: 0xffff0000 | requester for a request, 0xfffe0000 | requester for its response.
#define LBFT_TOKEN_REQUEST 0xffff0000u
#define LBFT_TOKEN_RESPONSE 0xfffe0000u
static bool exchange_layout(const lbft_batch* b) { return b && (b->p.quirks & 1u); }
int lbft_node_create_request(lbft_batch* b, size_t inst, uint32_t node, uint32_t* handle) {
  if (!handle) return LBFT_ERR_INVALID;
  if (!exchange_layout(b)) {
    if (!b || inst >= b->m || node >= b->p.n) return LBFT_ERR_INVALID;
    if (!b->manual) { g_err = "lbft_batch_manual_begin first"; return LBFT_ERR_STATE; }
    *handle = LBFT_TOKEN_REQUEST | node;
    return LBFT_OK;
  }
  unsigned long long h[1];
  int rc = node_op(b, OP_CREATE_REQUEST, inst, node, 0, 0, 0, h, 1);
  if (rc != LBFT_OK) return rc;
  if ((long long)h[0] < 0) { g_err = "no free snapshot slot (snapshot_capacity)"; return LBFT_ERR_FAULT; }
  *handle = (uint32_t)h[0];
  return LBFT_OK;
}
int lbft_node_handle_request(lbft_batch* b, size_t inst, uint32_t node, uint32_t request, uint32_t* response) {
  if (!response) return LBFT_ERR_INVALID;
  if (!exchange_layout(b)) {
    if (!b || inst >= b->m || node >= b->p.n || (request & 0xffff0000u) != LBFT_TOKEN_REQUEST) return LBFT_ERR_INVALID;
    if (!b->manual) { g_err = "lbft_batch_manual_begin first"; return LBFT_ERR_STATE; }
    if ((request & 0xffffu) != node) {
      g_err = "reference mode (quirks bit 0 clear): a request is answered by its own requester (simulator.rs:446); create the batch with quirks bit 0 for peer-answered requests";
      return LBFT_ERR_UNSUPPORTED;
    }
    *response = LBFT_TOKEN_RESPONSE | node;
    return LBFT_OK;
  }
  if (request >= b->p.scap) return LBFT_ERR_INVALID;
  unsigned long long h[1];
  int rc = node_op(b, OP_HANDLE_REQUEST, inst, node, 0, request, 0, h, 1);
  if (rc != LBFT_OK) return rc;
  if ((long long)h[0] < 0) { g_err = "no free snapshot slot (snapshot_capacity)"; return LBFT_ERR_FAULT; }
  *response = (uint32_t)h[0];
  return LBFT_OK;
}
int lbft_node_handle_response(lbft_batch* b, size_t inst, uint32_t node, uint32_t peer, uint32_t response, int64_t node_time) {
  if (!exchange_layout(b)) {
    if (!b || inst >= b->m || node >= b->p.n || peer >= b->p.n || (response & 0xffff0000u) != LBFT_TOKEN_RESPONSE) return LBFT_ERR_INVALID;
    if (!b->manual) { g_err = "lbft_batch_manual_begin first"; return LBFT_ERR_STATE; }
    if ((response & 0xffffu) != node) { g_err = "reference mode: the response to a self-answered request goes back to its requester"; return LBFT_ERR_INVALID; }
    return LBFT_OK;  // data_sync.rs:209-240 over records the node already holds: every insertion is rejected as a duplicate
  }
  if (peer >= b->p.n || response >= b->p.scap) return LBFT_ERR_INVALID;
  return node_op(b, OP_HANDLE_RESPONSE, inst, node, peer, response, node_time, nullptr, 0);
}
int lbft_node_release_notification(lbft_batch* b, size_t inst, uint32_t handle) {
  if (b && !exchange_layout(b) && (handle & 0xfffe0000u) == 0xfffe0000u) return LBFT_OK;  // payload-free request / response token
  if (b && handle >= b->p.scap) return LBFT_ERR_INVALID;
  return node_op(b, OP_RELEASE_NOTIFICATION, inst, 0, 0, handle, 0, nullptr, 0);
}
int lbft_node_view_get(lbft_batch* b, size_t inst, uint32_t node, lbft_node_view* out) {
  if (!out) return LBFT_ERR_INVALID;
  unsigned long long h[15];
  int rc = node_op(b, OP_VIEW, inst, node, 0, 0, 0, h, 15);
  if (rc != LBFT_OK) return rc;
  out->epoch_id = h[0]; out->current_round = h[1]; out->highest_quorum_certificate_round = h[2];
  out->highest_timeout_certificate_round = h[3]; out->highest_committed_round = h[4]; out->active_round = h[5];
  out->latest_voted_round = h[6]; out->locked_round = h[7]; out->commit_count = h[8];
  out->active_leader = (uint32_t)h[9]; out->election = (uint32_t)h[10];
  out->num_current_timeouts = (uint32_t)h[11]; out->num_current_votes = (uint32_t)h[12];
  out->has_proposed_block = (uint32_t)h[13]; out->has_timeout_certificate = (uint32_t)h[14];
  return LBFT_OK;
}

int lbft_batch_enable_round_trace(lbft_batch* b, uint32_t max_rounds) {
  if (!b) return LBFT_ERR_INVALID;
  if (b->ran || b->manual) { g_err = "enable the round trace before running the batch"; return LBFT_ERR_STATE; }
  b->rcap = max_rounds;
  return LBFT_OK;
}

// past_record_stores (node.rs:43,338-340) kept in full on the device, so that lbft_batch_save_node also serves nodes that have changed
// epoch.  Costs num_nodes x epochs x one node's rows of device memory per instance; off by default.
int lbft_batch_keep_retired_stores(lbft_batch* b, int enable) {
  if (!b) return LBFT_ERR_INVALID;
  if (b->ran || b->manual || b->started) { g_err = "ask for the retired record stores before running the batch"; return LBFT_ERR_STATE; }
  b->keep_stores = enable != 0;
  return LBFT_OK;
}

// out[round * num_nodes + node] = GlobalTime at which `node` was first seen in `round` by the reference's DataWriter
// (bft-lib/src/data_writer.rs:34-50), INT64_MIN = empty cell; rows for round < min(max_round, cap_rounds).
int lbft_batch_round_switches(const lbft_batch* b, size_t inst, int64_t* out, size_t cap_rounds, uint64_t* max_round, uint64_t* messages) {
  if (!b || !out || !max_round || inst >= b->m) return LBFT_ERR_INVALID;
  if (!b->ran) { g_err = "run the batch first"; return LBFT_ERR_STATE; }
  if (b->p.rcap == 0) { g_err = "lbft_batch_enable_round_trace was not called"; return LBFT_ERR_STATE; }
  HIP_TRY(hipSetDevice(b->device));
  const Params& p = b->p;
  u32 words = p.n * p.rcap + p.n;
  std::vector<u32> h(words + 3);
  // strided rows of one instance: word w lives at word_offset(p, inst, w); copy as a 2-D memcpy (4 bytes x words, pitch 256)
  HIP_TRY(hipMemcpy2D(h.data(), sizeof(u32), b->d_state + word_offset(p, (u32)inst, p.off_trace), (size_t)4 * p.tw, sizeof(u32), words,
                      hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy2D(h.data() + words, sizeof(u32), b->d_state + word_offset(p, (u32)inst, I_EV0), (size_t)4 * p.tw, sizeof(u32), 3,
                      hipMemcpyDeviceToHost));
  u32 mr = 0;
  for (u32 k = 0; k < p.n; k++) mr = h[p.n * p.rcap + k] > mr ? h[p.n * p.rcap + k] : mr;
  *max_round = mr;
  for (u32 r = 0; r < mr && r < cap_rounds && r < p.rcap; r++)
    for (u32 k = 0; k < p.n; k++) {
      u32 t = h[k * p.rcap + r];
      out[(size_t)r * p.n + k] = t == 0xffffffffu ? INT64_MIN : (int64_t)(i32)t;
    }
  if (messages) *messages = (uint64_t)h[words] + h[words + 1] + h[words + 2];  // DataWriter::add_message_counter: every non-timer event
  return LBFT_OK;
}

int lbft_batch_set_calendar_queue(lbft_batch* b, int enabled) {
  if (!b) return LBFT_ERR_INVALID;
  b->allow_calendar = enabled != 0;
  return LBFT_OK;
}

int lbft_batch_set_lds_queue_slots(lbft_batch* b, int32_t slots) {
  if (!b || slots < -1) return LBFT_ERR_INVALID;
  b->ql = slots;
  return LBFT_OK;
}

int lbft_batch_phase_cycles(const lbft_batch* b, uint64_t* out) {
  if (!b || !out) return LBFT_ERR_INVALID;
#if defined(LBFT_PHASE_TIMERS)
  if (!b->ran) { g_err = "run the batch first"; return LBFT_ERR_STATE; }
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipMemcpy(out, b->d_prof, LBFT_NPHASES * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return LBFT_OK;
#else
  g_err = "this build has no phase timers (compile with -DLBFT_PHASE_TIMERS)";
  return LBFT_ERR_UNSUPPORTED;
#endif
}

int lbft_batch_layout(const lbft_batch* b, uint32_t* out) {
  if (!b || !out) return LBFT_ERR_INVALID;
  if (!b->ran) { g_err = "run the batch first"; return LBFT_ERR_STATE; }
  const Params& p = b->p;
  // what one event moves (the roofline's S_node / S_notif, bench.py, tools/configs.py), not the padded row sizes: a node burst is the
  // fixed words + the extension words of the four author sets (begin_node / end_node); the hcbr buffers (2n words behind them) are
  // only touched where a timeout is inserted or copied -- for networks of <= 4 nodes they live in LDS for the whole launch.  A
  // notification snapshot: its fixed words + set extension words; for <= 4 nodes also its 2n hcbr words (always fetched with it), for
  // larger networks the hcbr words an event happens to carry are NOT counted (the figure is a lower bound there).
  // (lbft_k_run0q, LBFT_C0_HCREG: the node's 2n hcbr words ride in its burst -- hc_load / hc_store -- and are counted)
  out[0] = (NF_FIXED_WORDS + 4 * (p.mw - 1) + ((quad_kernel(p) && LBFT_C0_HCREG) ? 2 * p.n : 0)) * 4;
  out[1] = (p.qpack ? 8 : p.qcal ? 4 : 12);  // (the calendar keeps no key: an entry is the 4-byte meta word, plus 1/31 of a chunk's link word)
  out[2] = (S_FIXED_WORDS + 2 * (p.mw - 1) + (p.n <= 4 ? 2 * p.n : 0)) * 4;
  out[3] = (B_WORDS + 4 * (p.mw - 1)) * 4;   // bytes of one block record
  out[4] = p.total_words * 4; // HBM bytes per instance
  out[5] = p.ql;              // event-queue slots per instance resident in LDS
  out[6] = p.lpw;             // lanes per wavefront carrying an instance
  out[7] = (uint32_t)sim_class(p) | (p.qheap << 8) | (p.qcal << 9) | ((((sim_lean(p) && lean2_allowed()) || (sim_lean1(p) && lean_allowed())) ? 1u : 0u) << 10) | ((p.ring ? 1u : 0u) << 11) |
           (((sim_lean_q1(p) && lean2_allowed()) ? 1u : 0u) << 12) | ((small_batch_kernel(p) ? 1u : 0u) << 13) |
           ((quad_kernel(p) ? 1u : 0u) << 14) | ((uni_kernel(p) ? 1u : 0u) << 15);
  return LBFT_OK;
}

int lbft_batch_reset(lbft_batch* b) {
  if (!b) return LBFT_ERR_INVALID;
  b->ran = false;
  b->manual = false;
  b->started = false;
  b->step_launches = 0;
  return LBFT_OK;
}

// The calendar queue's head / tail / bitmap rows must be zero when Simulator::new runs: they are one contiguous
// range of rows in every 64-instance tile.
static int zero_calendar(lbft_batch* b) {
  const Params& p = b->p;
  if (!p.qcal) return LBFT_OK;
  size_t rows = (size_t)p.off_snap - p.off_cal_head;  // head, tail, bitmap
  size_t tiles = p.stride / p.tw, row_bytes = (size_t)4 * p.tw;
  HIP_TRY(hipMemset2DAsync(reinterpret_cast<char*>(b->d_state) + (size_t)p.off_cal_head * row_bytes, (size_t)p.total_words * row_bytes, 0,
                           rows * row_bytes, tiles, b->stream));
  return LBFT_OK;
}

// Capacities, HBM layout, launch geometry (shared by lbft_batch_run_until and lbft_batch_manual_begin).
static int prepare_run(lbft_batch* b, int64_t max_clock) {
  if (max_clock < 0 || max_clock >= 0x7ffffffeLL) { g_err = "max_clock out of range"; return LBFT_ERR_INVALID; }
  HIP_TRY(hipSetDevice(b->device));
  Params& p = b->p;
  const lbft_config& c = b->cfg;
  u32 n = c.num_nodes;
  // Capacities (0 = auto).  The queue only ever holds events with time <= max_clock.
  // (large networks: ~n^2 messages in flight per round; the heap keeps push/pop logarithmic)
  u32 qauto = n <= 16 ? 16 * n * n : 8 * n * n;
  u32 qcap = c.queue_capacity ? c.queue_capacity : (qauto < 128 ? 128 : qauto);
  // (quirks bit 0: every request and response in flight holds a slot as well)
  // (measured high-water marks with quirks bit 0: ~n^2 -- 400 at n = 20, 1250 at n = 36 -- and flat over the horizon)
  u32 sauto = (c.quirks & 1u) ? (n * n + 8 * n > 64 * n ? n * n + 8 * n : 64 * n) : 8 * n;
  u32 scap = c.snapshot_capacity ? c.snapshot_capacity : (sauto < 32 ? 32 : (sauto > 65535 ? 65535 : sauto));
  // one block per round; a 1- or 2-node network can finish a round per time unit
  u64 bauto = n <= 2 ? (u64)max_clock + 64 : (u64)max_clock / 10 + 64;
  u32 bcap = c.block_capacity ? c.block_capacity : (u32)(bauto > 65534 ? 65534 : bauto);
  if (bcap > 65534 || scap > 65535 || n > 255) { g_err = "capacity out of range"; return LBFT_ERR_INVALID; }
  u32 lcap = c.log_capacity ? c.log_capacity : bcap;
  if (lcap > bcap) lcap = bcap;  // a node commits each block at most once
  // Queue discipline: 4-node honest lossless networks scan an LDS-resident array (kernel class 0); everything else keeps
  // hundreds to tens of thousands of pending events and uses a calendar of (time, kind) FIFOs when max_clock allows it
  // (O(1) push and pop), otherwise a binary heap whose top levels are the LDS-resident slots.
  bool big = qcap > 256 || n > 32;
  u32 qheap = big ? 1u : 0u;
  bool class0 = n <= 16 && !qheap && !p.equiv && !b->rcap && !p.drop_ppm && !p.part_size && !(p.quirks & 1u);
  // epochs a node can go through are bounded by its commits: the archive of retired record stores (quirks bit 0) is exact
  u64 eauto = (u64)bcap / (c.commands_per_epoch ? c.commands_per_epoch : 1) + 2;
  u32 ecap = ((p.quirks & 1u) || b->keep_stores) ? (u32)(eauto > 4096 ? 4096 : eauto) : 0;
  const u32 rarch = b->keep_stores ? 1u : 0u;  // (compute_layout turns the flag into the entry size)
  // (the calendar replaces the HEAP: a small network outside class 0 -- e.g. 4 nodes with an equivocator -- keeps the LDS-fronted
  // array; 65536 x 4 nodes with one equivocator each: 28.6 ms on the array, 43.6 ms on the HBM calendar)
  u32 qcal = (!class0 && big && !b->rcap && b->allow_calendar && max_clock <= LBFT_CAL_MAX_CLOCK) ? 1u : 0u;
  bool relayout = !(p.qcap == qcap && p.scap == scap && p.bcap == bcap && p.lcap == lcap && p.rcap == b->rcap && p.qcal == qcal && p.ecap == ecap &&
                    (p.rarch_words != 0) == (rarch != 0) && p.max_clock == (i32)max_clock && b->d_state);
  p.qcap = qcap; p.scap = scap; p.bcap = bcap; p.lcap = lcap; p.rcap = b->rcap; p.qcal = qcal; p.qheap = qheap; p.ecap = ecap; p.rarch_words = rarch;
  p.max_clock = (i32)max_clock;
  p.max_steps = b->max_steps;
  // Cooperative large-network kernels (class 2 on the calendar queue): ring of pre-generated RNG draws per instance and how far
  // every network's generator runs ahead per step (tuning knobs: LBFT_RING = entries, a power of two, 0 = lane-per-network
  // execution as for the small classes; LBFT_RING_TOPUP = draws per step)
  {
    u32 ring = 0, topup = 0;
    if (n > 32 && qcal) {
      // (round 6, third session: with the runs a large network makes ~15 k loop iterations instead of 215 k, and a bulk send of a 100-node network consumes ~220 draws:
      // a top-up of 4 per iteration left the bulk's leader lane generating them one at a time -- profiles/r06/ring_topup_after_all_runs.txt)
      ring = 512; topup = n > 64 ? 128 : 16;
      if (const char* e = getenv("LBFT_RING")) ring = (u32)atoi(e);
      if (const char* e = getenv("LBFT_RING_TOPUP")) topup = (u32)atoi(e);
      if (ring & (ring - 1)) ring = 512;
      if (ring && ring < 128) ring = 128;
    }
    if (p.ring != ring) relayout = true;
    p.ring = ring; p.ring_topup = ring ? topup : 0;
  }
  u64 words = compute_layout(p);
  if (p.qcal) {  // the calendar's bucket rows grow with the horizon: keep it only while the batch fits comfortably in HBM
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    size_t avail = free_b + (b->d_state ? b->state_bytes : 0);
    if (!layout_fits(words) || (double)words * p.stride * 4.0 > 0.85 * (double)avail) {
      qcal = 0; p.qcal = 0;
      relayout = true;
      words = compute_layout(p);
    }
  }
  if (!layout_fits(words)) {
    p.qcap = 0;  // (forces a relayout next time)
    g_err = "per-instance state exceeds 2^24 rows (64 MiB per instance): lower the capacities / the horizon";
    return LBFT_ERR_INVALID;
  }
  if (relayout) {
    if (b->d_state) { HIP_TRY(hipFree(b->d_state)); b->d_state = nullptr; }
    b->state_bytes = state_words(p) * sizeof(u32);
    HIP_TRY(hipMalloc(&b->d_state, b->state_bytes));
  }
  // Lanes per wavefront that carry an instance.  The LDS queue front makes residency LDS-bound: one CU holds
  // 160 KiB / (bytes per instance) instances however they are spread over wavefronts, so prefer full wavefronts
  // unless the batch is too small to give every SIMD a wavefront.
  u32 lpw = b->lpw;
  if (lpw == 0) {
    // Wavefronts that can be resident at once: 256 CUs x 4 SIMDs, two per SIMD for kernel class 0 (256 registers), one for
    // the large-network classes.  The fewest lanes per wavefront that still fit the batch in one residency win: a
    // wavefront-step costs the union of its lanes' paths (65536 x 4 nodes, r01_s3 build: 27.0 ms at 64 lanes = one wavefront
    // per SIMD, 24.4 ms at 32 = two per SIMD, 40.1 ms at 16 = two rounds; 1024 x 4 nodes: 22.9 ms at 8 lanes, 17.6 at 4,
    // 13.2 at 2, 9.4 ms at ONE network per wavefront; 8192 x 100 nodes: 9.0 s at 16 lanes, 5.9 s at 8, 7.8 s at 4 = two rounds).
    const bool lean2k = sim_lean(p) && lean2_allowed();
    u64 resident = (lean2k || sim_class(p) == K_SMALL || (sim_lean1(p) && lean_allowed())) ? 2048 : 1024 * LBFT_BIG_WAVES_PER_SIMD;
    u64 want = (b->m + resident - 1) / resident;
    lpw = 1;
    while (lpw < want && lpw < 32) lpw <<= 1;
  }
  p.lpw = lpw;
  // Tile width of the HBM layout (lbft_core.h "HBM layout"): 64 for the small-network classes 0 and 1, 1 (instance-major) for large networks
  {
    // Measured (16 384 x 64 nodes / 8 192 x 100 nodes / c4live): tw = 64: 652 ms / 3.63 s / --; tw = lanes per wavefront: 596 / 3.20 / 4.94 s;
    // tw = 4: 555 / 3.17 / 4.43; tw = 2: 550 / 3.10 / 4.28; tw = 1: 541 ms / 3.05 s / 4.14 s -- the large-network kernels address tw = 1 at compile time.
    u32 tw = layout_tile_width(p);
    // (instance-major rows: a lane addresses its instance through a 32-bit offset from the wavefront's first instance)
    if (tw == 1 && (u64)lpw * p.total_words * 4ULL >= (1ULL << 32)) {
      g_err = "lanes per wavefront x per-instance state exceeds 4 GiB: lower the capacities or the lanes per wavefront";
      return LBFT_ERR_INVALID;
    }
    p.tw = tw;
    p.rsh = 2;
    while ((4u << (p.rsh - 2)) < 4u * tw) p.rsh++;
  }
  // wavefronts per workgroup of the kernel this batch runs on: 8 = both wavefront slots of a CU's four SIMDs for the kernels compiled
  // for two wavefronts per SIMD, 4 for the full-register ones
  const bool two_wave_kernel = sim_class(p) == K_SMALL || (sim_lean(p) && lean2_allowed()) || (sim_lean1(p) && lean_allowed());
  const u32 nwaves = two_wave_kernel ? LBFT_RUN_WAVES : LBFT_RUN_WAVES_FULL;
  b->run_waves = nwaves;
  // LDS queue slots per instance: what one CU's LDS affords when it hosts 64/lpw workgroups
  u32 wg_per_cu = (64 / lpw) * 4 / nwaves;  // workgroups that make up a CU's 256 instances
  if (wg_per_cu < 1) wg_per_cu = 1;
  if (wg_per_cu > 4) wg_per_cu = 4;
  // (the kernels compiled for two wavefronts per SIMD run as 8-wavefront workgroups: one of them fills a CU's wavefront slots at 256
  // registers per lane, so the whole LDS is that one workgroup's whatever its lanes per wavefront -- round 4: with the budget of two the
  // 16-lane form of lbft_k_run0q kept 24 queue slots in LDS and spilled the rest to HBM)
  if (two_wave_kernel && nwaves >= 8) wg_per_cu = 1;
  size_t budget = (160u * 1024u) / wg_per_cu;
  // 2 KiB slack per workgroup: with less, two workgroups of 32-lane wavefronts do not become co-resident on a CU
  u32 slot_bytes = p.qpack ? 8u : 12u;  // kernel class 0 keeps one-word entries
  p.lpw = lpw;
  const bool quadk = quad_kernel(p);  // (the kernel choice only depends on the layout and lpw)
  const bool hcbr_lds = !(LBFT_C0_IMAJOR && LBFT_C0_HCREG && quadk);
  const u32 qcols = (quadk && LBFT_QUAD_STRIDE32) ? 32u : lpw;  // queue columns per wavefront in LDS (SimT::QS32)
  u32 ql_auto = (u32)((budget - run_lds_bytes(0, qcols, n, slot_bytes, nwaves, hcbr_lds) - 2048) / (slot_bytes * nwaves * qcols));  // (run_lds_bytes(0, ..) includes the lane padding)
  const u32 ql_max = quadk ? LBFT_PACKED_QL_QUAD : LBFT_PACKED_QL_MAX, pop_batch = quadk ? LBFT_POP_BATCH_QUAD : LBFT_POP_BATCH;
  if (p.qpack && ql_auto > ql_max) ql_auto = ql_max;
  u32 ql = b->ql < 0 ? ql_auto : (u32)b->ql;
  if (ql > qcap) ql = qcap;
  if (p.qpack) ql -= ql % pop_batch;  // scanned in batches (SimT::PB)
  if (p.qcal) ql = 0;  // the calendar lives in HBM rows
  if (run_lds_bytes(ql, qcols, n, slot_bytes, nwaves, hcbr_lds) > 160u * 1024u) { g_err = "LDS queue slots do not fit the CU's 160 KiB"; return LBFT_ERR_INVALID; }
  p.ql = ql;
  b->lds_bytes = run_lds_bytes(ql, qcols, n, slot_bytes, nwaves, hcbr_lds);
  // large networks: the LDS that the calendar queue leaves unused holds a window of block records per network (SimT::attach_blk_window)
  p.blw = 0;
  // (measured, round 4: c4live 2.77 -> 2.76 s, c5live 4.72 -> 4.60 s with 32 entries, 4.58 s with 64; the kernel without the record exchange
  // LOSES with it -- c4 346 -> 357 ms, c5 1.90 -> 1.98 s, its three register records already serve it -- and does not get one)
  if (sim_lean_q1(p) && lean2_allowed() && blk_window_allowed()) {
    u32 e = blk_window_max();
    while (e && b->lds_bytes + blk_window_bytes(e, lpw, nwaves) > 150u * 1024u) e >>= 1;
    p.blw = e;
    b->lds_bytes += blk_window_bytes(e, lpw, nwaves);
  }
  p.prof = b->d_prof;
  return LBFT_OK;
}

int lbft_batch_run_until(lbft_batch* b, int64_t max_clock) {
  if (!b) return LBFT_ERR_INVALID;
  if (b->ran || b->started) { g_err = "batch already ran (or is being stepped); call lbft_batch_reset first"; return LBFT_ERR_STATE; }
  int prc = prepare_run(b, max_clock);
  if (prc != LBFT_OK) return prc;
  Params& p = b->p;
  u32 lpw = p.lpw;
#if defined(LBFT_PHASE_TIMERS)
  HIP_TRY(hipMemsetAsync(b->d_prof, 0, LBFT_NPHASES * sizeof(unsigned long long), b->stream));  // (product builds never write the phase accumulators)
#endif
  u32 grid_full = (u32)((b->m + LBFT_BLOCK - 1) / LBFT_BLOCK);
  u32 grid_init = (u32)((b->m + lpw - 1) / lpw);
  HIP_TRY(hipEventRecord(b->ev0, b->stream));
  { int zrc = zero_calendar(b); if (zrc != LBFT_OK) return zrc; }
  b->generation++;
  lbft_k_init<<<grid_init, LBFT_BLOCK, 0, b->stream>>>(p, b->d_state, b->d_seeds);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(b->ev1, b->stream));
  u64 launches = 0;
  for (;;) {
    int rc = launch_run(b);
    if (rc != LBFT_OK) return rc;
    launches++;
    if (p.max_steps == 0) break;  // whole simulation in one launch
    u32 unfinished = 0;
    HIP_TRY(hipMemcpyAsync(&unfinished, b->d_unfinished, sizeof(u32), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (unfinished == 0) break;
  }
  HIP_TRY(hipEventRecord(b->ev2, b->stream));
  return finalize_run(b, grid_full, launches);
}

static int launch_run(lbft_batch* b) {
  Params& p = b->p;
  int cls = sim_class(p);
  bool lean = sim_lean(p) && lean2_allowed(), lean1 = sim_lean1(p) && lean_allowed();
  const bool leanq = lean && sim_lean_q1(p);
  const bool small0 = small_batch_kernel(p);
  const bool quad0 = quad_kernel(p);
  const bool uni0 = cls == K_SMALL && uni_kernel(p);
  const void* run_fn = leanq ? reinterpret_cast<const void*>(lbft_k_run2q) : lean ? reinterpret_cast<const void*>(lbft_k_run2l) : lean1 ? reinterpret_cast<const void*>(lbft_k_run1l) :
                       uni0 ? reinterpret_cast<const void*>(lbft_k_run0u) :
                       (cls == K_SMALL && small0) ? reinterpret_cast<const void*>(lbft_k_run0s) : (cls == K_SMALL && quad0) ? reinterpret_cast<const void*>(lbft_k_run0q) : cls == K_SMALL ? reinterpret_cast<const void*>(lbft_k_run0)
                     : cls == K_MID ? reinterpret_cast<const void*>(lbft_k_run<K_MID>) : reinterpret_cast<const void*>(lbft_k_run<K_LARGE>);
  HIP_TRY(hipFuncSetAttribute(run_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_bytes));
  const u32 nwaves = b->run_waves, block = 64u * nwaves;
  u32 grid_run = (u32)((b->m + (size_t)nwaves * p.lpw - 1) / ((size_t)nwaves * p.lpw));
  HIP_TRY(hipMemsetAsync(b->d_unfinished, 0, sizeof(u32), b->stream));
  b->generation++;
  if (leanq) lbft_k_run2q<<<grid_run, block, b->lds_bytes, b->stream>>>(p, b->d_state, b->d_unfinished);
  else if (lean) lbft_k_run2l<<<grid_run, block, b->lds_bytes, b->stream>>>(p, b->d_state, b->d_unfinished);
  else if (lean1) lbft_k_run1l<<<grid_run, block, b->lds_bytes, b->stream>>>(p, b->d_state, b->d_unfinished);
  else if (uni0) lbft_k_run0u<<<grid_run, block, b->lds_bytes, b->stream>>>(p, b->d_state, b->d_unfinished);
  else if (cls == K_SMALL && small0) lbft_k_run0s<<<grid_run, block, b->lds_bytes, b->stream>>>(p, b->d_state, b->d_unfinished);
  else if (cls == K_SMALL && quad0) lbft_k_run0q<<<grid_run, block, b->lds_bytes, b->stream>>>(p, b->d_state, b->d_unfinished);
  else if (cls == K_SMALL) lbft_k_run0<<<grid_run, block, b->lds_bytes, b->stream>>>(p, b->d_state, b->d_unfinished);
  else if (cls == K_MID) lbft_k_run<K_MID><<<grid_run, block, b->lds_bytes, b->stream>>>(p, b->d_state, b->d_unfinished);
  else lbft_k_run<K_LARGE><<<grid_run, block, b->lds_bytes, b->stream>>>(p, b->d_state, b->d_unfinished);
  HIP_TRY(hipGetLastError());
  return LBFT_OK;
}

// ---- stepwise execution and checkpoint / resume (the reference's save_node / load_node, node.rs:211-238, at batch
// granularity: the whole SoA state instead of one bincode blob per node per event) ----
int lbft_batch_run_steps(lbft_batch* b, int64_t max_clock, uint32_t steps, uint64_t* unfinished) {
  if (!b || !unfinished) return LBFT_ERR_INVALID;
  if (b->ran || b->manual) { g_err = "batch already ran; call lbft_batch_reset first"; return LBFT_ERR_STATE; }
  if (!b->started) {
    int rc = prepare_run(b, max_clock);
    if (rc != LBFT_OK) return rc;
    u32 grid_init = (u32)((b->m + b->p.lpw - 1) / b->p.lpw);
    HIP_TRY(hipEventRecord(b->ev0, b->stream));
    { int zrc = zero_calendar(b); if (zrc != LBFT_OK) return zrc; }
    b->generation++;
    lbft_k_init<<<grid_init, LBFT_BLOCK, 0, b->stream>>>(b->p, b->d_state, b->d_seeds);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(b->ev1, b->stream));
    b->started = true;
    b->started_max_clock = max_clock;
  } else if (max_clock != b->started_max_clock) {
    g_err = "max_clock differs from the one this run was started with (events past it were already dropped)";
    return LBFT_ERR_INVALID;
  }
  b->p.max_steps = steps;
  int rc = launch_run(b);
  if (rc != LBFT_OK) return rc;
  b->step_launches++;
  u32 left = 0;
  HIP_TRY(hipMemcpyAsync(&left, b->d_unfinished, sizeof(u32), hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  *unfinished = left;
  if (left == 0) {
    HIP_TRY(hipEventRecord(b->ev2, b->stream));
    b->started = false;
    return finalize_run(b, (u32)((b->m + LBFT_BLOCK - 1) / LBFT_BLOCK), b->step_launches);
  }
  return LBFT_OK;
}

struct CheckpointHeader {
  char magic[8];  // "LBFTCKP5"
  u32 n, qcap, scap, bcap, lcap, rcap, total_words, equiv;
  // everything that changes the meaning of the state words without changing their number: the protocol mode, the fault model,
  // the kernel class and the queue discipline / key encoding it implies, the archive capacity of retired record stores
  u32 quirks, drop_ppm, part_size, rot, sim_class, qpack, qcal, qheap, ecap, tw;
  i64 part_start, part_end;
  u64 m, cpe;
  i64 max_clock, tci, delta, uni_lo, uni_hi;
  double mean, variance, gamma, lambda;
  u32 delay_model, weights_hash;
};
static u32 weights_hash(const std::vector<u32>& w) {
  u32 h = 2166136261u;
  for (u32 v : w) { h ^= v; h *= 16777619u; }
  return h;
}
static void fill_header(const lbft_batch* b, CheckpointHeader& h) {
  memset(&h, 0, sizeof(h));
  memcpy(h.magic, "LBFTCKP5", 8);
  const Params& p = b->p; const lbft_config& c = b->cfg;
  h.n = p.n; h.qcap = p.qcap; h.scap = p.scap; h.bcap = p.bcap; h.lcap = p.lcap; h.rcap = p.rcap; h.total_words = p.total_words;
  h.equiv = p.equiv; h.m = b->m; h.cpe = c.commands_per_epoch; h.max_clock = b->started_max_clock; h.tci = c.target_commit_interval;
  h.delta = c.delta; h.uni_lo = c.uniform_lo; h.uni_hi = c.uniform_hi; h.mean = c.mean; h.variance = c.variance; h.gamma = c.gamma;
  h.lambda = c.lambda; h.delay_model = c.delay_model; h.weights_hash = weights_hash(b->weights) ^ (p.rot * 0x9e3779b9u);
  h.quirks = p.quirks; h.drop_ppm = p.drop_ppm; h.part_size = p.part_size; h.rot = p.rot; h.sim_class = (u32)sim_class(p) | (((sim_lean(p) && lean2_allowed()) || (sim_lean1(p) && lean_allowed())) ? 256u : 0u);
  h.qpack = p.qpack; h.qcal = p.qcal; h.qheap = p.qheap; h.ecap = p.ecap; h.tw = p.tw | (p.ring << 8); h.part_start = c.partition_start; h.part_end = c.partition_end;
}
size_t lbft_batch_checkpoint_bytes(const lbft_batch* b) {
  if (!b || !b->started) return 0;
  return sizeof(CheckpointHeader) + b->state_bytes;
}
int lbft_batch_checkpoint_save(const lbft_batch* b, void* buf, size_t cap) {
  if (!b || !buf) return LBFT_ERR_INVALID;
  if (!b->started) { g_err = "nothing to checkpoint: start the run with lbft_batch_run_steps"; return LBFT_ERR_STATE; }
  if (cap < sizeof(CheckpointHeader) + b->state_bytes) { g_err = "checkpoint buffer too small"; return LBFT_ERR_INVALID; }
  HIP_TRY(hipSetDevice(b->device));
  CheckpointHeader h;
  fill_header(b, h);
  memcpy(buf, &h, sizeof(h));
  HIP_TRY(hipMemcpy((char*)buf + sizeof(h), b->d_state, b->state_bytes, hipMemcpyDeviceToHost));
  return LBFT_OK;
}
int lbft_batch_checkpoint_load(lbft_batch* b, const void* buf, size_t len) {
  if (!b || !buf || len < sizeof(CheckpointHeader)) return LBFT_ERR_INVALID;
  if (b->ran || b->manual || b->started) { g_err = "load a checkpoint into a fresh (or reset) batch"; return LBFT_ERR_STATE; }
  CheckpointHeader h;
  memcpy(&h, buf, sizeof(h));
  if (memcmp(h.magic, "LBFTCKP5", 8) != 0) { g_err = "not a checkpoint (or one of an older format)"; return LBFT_ERR_INVALID; }
  // the batch must have been created with the same configuration; capacities come from the checkpoint.  A failed load leaves
  // the batch's own capacities as they were.
  const lbft_config saved_cfg = b->cfg;
  const u32 saved_rcap = b->rcap;
  const int64_t saved_max_clock = b->started_max_clock;
  b->cfg.queue_capacity = h.qcap; b->cfg.snapshot_capacity = h.scap; b->cfg.block_capacity = h.bcap; b->cfg.log_capacity = h.lcap;
  b->rcap = h.rcap;
  int rc = prepare_run(b, h.max_clock);
  if (rc == LBFT_OK) {
    b->started_max_clock = h.max_clock;
    CheckpointHeader mine;
    fill_header(b, mine);
    if (memcmp(&mine, &h, sizeof(h)) != 0) { g_err = "checkpoint was taken from a batch with a different configuration (parameters, capacities, protocol mode -- or a different kernel family: the "
              "LBFT_NO_LEAN / LBFT_LEAN2 / LBFT_RING tuning variables and lbft_batch_keep_retired_stores change what the state words hold)"; rc = LBFT_ERR_INVALID; }
    else if (len != sizeof(h) + b->state_bytes) { g_err = "checkpoint size mismatch"; rc = LBFT_ERR_INVALID; }
  }
  if (rc != LBFT_OK) { b->cfg = saved_cfg; b->rcap = saved_rcap; b->started_max_clock = saved_max_clock; return rc; }
  b->generation++;
  HIP_TRY(hipMemcpy(b->d_state, (const char*)buf + sizeof(h), b->state_bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipEventRecord(b->ev0, b->stream));
  HIP_TRY(hipEventRecord(b->ev1, b->stream));
  b->started = true;
  return LBFT_OK;
}

static int finalize_run(lbft_batch* b, u32 grid_full, u64 launches) {
  Params& p = b->p;
  HIP_TRY(hipMemsetAsync(b->d_counters, 0, C_WORDS * sizeof(unsigned long long), b->stream));
  lbft_k_finalize<<<dim3(grid_full, (p.n + LBFT_FINAL_WAVES - 1) / LBFT_FINAL_WAVES), LBFT_BLOCK * LBFT_FINAL_WAVES, 0, b->stream>>>(p, b->d_state, b->d_states_out, b->d_counters);
  HIP_TRY(hipGetLastError());
  // reset + init + run + finalize + this copy are ONE stream sequence with a single synchronisation at its end (no host round trip in between); the
  // destination is pinned, so the copy is a DMA behind the finalize kernel instead of a staged copy with a synchronisation of its own
  if (!b->h_counters) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&b->h_counters), C_WORDS * sizeof(unsigned long long), hipHostMallocDefault));
  unsigned long long* hc = b->h_counters;
  HIP_TRY(hipMemcpyAsync(hc, b->d_counters, C_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  HIP_TRY(hipEventElapsedTime(&b->init_ms, b->ev0, b->ev1));
  HIP_TRY(hipEventElapsedTime(&b->run_ms, b->ev1, b->ev2));
  lbft_counters& k = b->counters;
  memset(&k, 0, sizeof(k));
  for (int e = 0; e < 4; e++) k.events[e] = hc[C_EV0 + e];
  k.rng_draws = hc[C_DRAWS]; k.rounds = hc[C_ROUNDS]; k.commits = hc[C_COMMITS]; k.events_scheduled = hc[C_SCHED];
  k.faulted_instances = hc[C_FAULTED]; k.max_queue = hc[C_MAXQ]; k.max_snapshots = hc[C_MAXSNAP]; k.max_blocks = hc[C_MAXBLK];
  k.launches = launches;
  k.timers_folded = hc[C_NFOLD]; k.node_updates = hc[C_NUPD];
  b->ran = true;
  if (k.faulted_instances) { g_err = "some instances raised a fault; see lbft_batch_faults"; return LBFT_ERR_FAULT; }
  return LBFT_OK;
}

static int gather_node(const lbft_batch* b, u32 field, u32* host_out) {
  if (!b || !host_out) return LBFT_ERR_INVALID;
  if (!b->ran) { g_err = "run the batch first"; return LBFT_ERR_STATE; }
  HIP_TRY(hipSetDevice(b->device));
  u32 total = (u32)(b->m * b->p.n);
  lbft_k_gather_node<<<(total + 255) / 256, 256, 0, b->stream>>>(b->p, b->d_state, field, b->d_scratch);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(host_out, b->d_scratch, (size_t)total * sizeof(u32), hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return LBFT_OK;
}

int lbft_batch_commit_counts(const lbft_batch* b, uint32_t* out) { return gather_node(b, NF_NCOMMITS, out); }

static int gather_node_u64(const lbft_batch* b, u32 field, bool sign_extend, uint64_t* out) {
  if (!b || !out) return LBFT_ERR_INVALID;
  std::vector<u32> tmp(b->m * b->p.n);
  int rc = gather_node(b, field, tmp.data());
  if (rc != LBFT_OK) return rc;
  for (size_t i = 0; i < tmp.size(); i++) out[i] = sign_extend ? (u64)(i64)(i32)tmp[i] : (u64)tmp[i];
  return LBFT_OK;
}
int lbft_batch_active_rounds(const lbft_batch* b, uint64_t* out) { return gather_node_u64(b, NF_PM_ROUND, false, out); }
int lbft_batch_epochs(const lbft_batch* b, uint64_t* out) { return gather_node_u64(b, NF_EPOCH, false, out); }
int lbft_batch_startup_times(const lbft_batch* b, int64_t* out) { return gather_node_u64(b, NF_STARTUP, true, (uint64_t*)out); }

int lbft_batch_last_committed_states(const lbft_batch* b, uint64_t* out) {
  if (!b || !out) return LBFT_ERR_INVALID;
  if (!b->ran) { g_err = "run the batch first"; return LBFT_ERR_STATE; }
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipMemcpy(out, b->d_states_out, b->m * b->p.n * sizeof(u64), hipMemcpyDeviceToHost));
  return LBFT_OK;
}

// StateFinalizer::last_committed_state() of one node (simulated_context.rs:51-55,194-196)
int lbft_batch_last_committed_state(const lbft_batch* b, size_t inst, uint32_t node, uint64_t* out) {
  if (!b || !out || inst >= b->m || node >= b->p.n) return LBFT_ERR_INVALID;
  if (!b->ran) { g_err = "run the batch first"; return LBFT_ERR_STATE; }
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipMemcpy(out, b->d_states_out + inst * b->p.n + node, sizeof(u64), hipMemcpyDeviceToHost));
  return LBFT_OK;
}

static int export_histories(const lbft_batch* b, size_t first, size_t count, lbft_commit* out, size_t cap) {
  if (cap == 0 || count == 0) return LBFT_OK;
  HIP_TRY(hipSetDevice(b->device));
  // stage through a bounded device buffer (<= 64 MiB per chunk)
  size_t per_inst = (size_t)b->p.n * cap * sizeof(lbft_commit);
  size_t chunk = (64u << 20) / per_inst;
  if (chunk == 0) chunk = 1;
  if (chunk > count) chunk = count;
  lbft_commit* d_buf = nullptr;
  HIP_TRY(hipMalloc(&d_buf, chunk * per_inst));
  int rc = LBFT_OK;
  for (size_t done = 0; done < count && rc == LBFT_OK; done += chunk) {
    size_t cnt = count - done < chunk ? count - done : chunk;
    hipError_t e = hipMemsetAsync(d_buf, 0, cnt * per_inst, b->stream);
    if (e == hipSuccess) {
      u32 total = (u32)(cnt * b->p.n);
      lbft_k_export_histories<<<(total + 63) / 64, 64, 0, b->stream>>>(b->p, b->d_state, d_buf, (u32)cap, (u32)(first + done), (u32)cnt);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync((char*)out + done * per_inst, d_buf, cnt * per_inst, hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (e != hipSuccess) rc = hip_fail(e, "export_histories");
  }
  hipFree(d_buf);
  return rc;
}

int lbft_batch_committed_histories(const lbft_batch* b, lbft_commit* out, size_t cap_per_node) {
  if (!b || !out) return LBFT_ERR_INVALID;
  if (!b->ran) { g_err = "run the batch first"; return LBFT_ERR_STATE; }
  if (cap_per_node > 0xffffffffu) return LBFT_ERR_INVALID;
  return export_histories(b, 0, b->m, out, cap_per_node);
}

int lbft_batch_committed_history(const lbft_batch* b, size_t inst, uint32_t node, lbft_commit* out, size_t cap, size_t* len) {
  if (!b || !len || inst >= b->m || node >= b->p.n) return LBFT_ERR_INVALID;
  if (!b->ran) { g_err = "run the batch first"; return LBFT_ERR_STATE; }
  u32 lcap = b->p.lcap;
  std::vector<lbft_commit> tmp((size_t)b->p.n * lcap);
  int rc = export_histories(b, inst, 1, tmp.data(), lcap);
  if (rc != LBFT_OK) return rc;
  HIP_TRY(hipSetDevice(b->device));
  u32 nc = 0;
  HIP_TRY(hipMemcpy(&nc, b->d_state + word_offset(b->p, (u32)inst, b->p.off_node + node * b->p.node_words + NF_NCOMMITS), sizeof(u32),
                    hipMemcpyDeviceToHost));
  *len = nc;
  for (size_t k = 0; k < nc && k < cap && out; k++) out[k] = tmp[(size_t)node * lcap + k];
  return LBFT_OK;
}

// SimT::committed_record_hashes for one (instance, node): a single lane walks the node's log
__global__ void lbft_k_record_hashes(Params p, const u32* __restrict__ state, u32 inst, u32 node, u64* __restrict__ out, u32 cap) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  Sim s(p, const_cast<u32*>(state), inst);
  s.committed_record_hashes(node, out, cap);
}

int lbft_batch_committed_record_hashes(const lbft_batch* b, size_t inst, uint32_t node, lbft_record_hash* out, size_t cap, size_t* len) {
  if (!b || !len || inst >= b->m || node >= b->p.n) return LBFT_ERR_INVALID;
  if (!b->ran) { g_err = "run the batch first"; return LBFT_ERR_STATE; }
  HIP_TRY(hipSetDevice(b->device));
  u32 nc = 0;
  HIP_TRY(hipMemcpy(&nc, b->d_state + word_offset(b->p, (u32)inst, b->p.off_node + node * b->p.node_words + NF_NCOMMITS), sizeof(u32),
                    hipMemcpyDeviceToHost));
  *len = nc;
  size_t k = nc < cap ? nc : cap;
  if (!k || !out) return LBFT_OK;
  u64* d_out = nullptr;
  HIP_TRY(hipMalloc(&d_out, k * 4 * sizeof(u64)));
  lbft_k_record_hashes<<<1, 64, 0, b->stream>>>(b->p, b->d_state, (u32)inst, node, d_out, (u32)k);
  hipError_t e = hipGetLastError();
  std::vector<u64> h(k * 4);
  if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_out, k * 4 * sizeof(u64), hipMemcpyDeviceToHost, b->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
  hipFree(d_out);
  if (e != hipSuccess) return hip_fail(e, "lbft_batch_committed_record_hashes");
  for (size_t i = 0; i < k; i++)
    out[i] = lbft_record_hash{h[4 * i], h[4 * i + 1], h[4 * i + 2], (uint32_t)h[4 * i + 3], (uint32_t)(h[4 * i + 3] >> 32)};
  return LBFT_OK;
}

// ---- ConsensusNode::save_node (librabft-v2/src/node.rs:233-238): the bincode image of one node's NodeState -------------
// The event loop keeps structural ids, masks and denormalised rounds instead of the reference's records; this is the way
// back: every Block / QuorumCertificate / Vote / Timeout the node's record store holds is rebuilt with the hashes and
// signatures the reference gives it (BCS + SipHash-1-3, as lbft_batch_committed_record_hashes does for the committed chain)
// and written in bincode 1.3's default encoding, HashMaps in ascending key order (oracle/lbft_oracle.cpp
// lbft_oracle_save_node is the format's restatement; the two images are compared byte for byte in tests/).  One instance's
// rows are copied to the host and serialised there -- a read-back format conversion like the history export, not simulation.
int lbft_batch_save_node(const lbft_batch* b, size_t inst, uint32_t node, void* buf, size_t cap, size_t* len) {
  if (!b || !len || inst >= b->m || node >= b->p.n) return LBFT_ERR_INVALID;
  if (!b->ran && !b->manual && !b->started) { g_err = "run the batch (or start a node-level session) first"; return LBFT_ERR_STATE; }
  HIP_TRY(hipSetDevice(b->device));
  const Params& dp = b->p;
  std::lock_guard<std::mutex> sn_lock(b->sn_mutex);  // (two threads saving different nodes of one batch share the one-image cache)
  if (b->sn_generation != b->generation || b->sn_inst != inst || b->sn_node != node) {
    // this instance's rows, contiguous on the host (a tile of width 1)
    std::vector<u32> hw(dp.total_words);
    if (dp.tw == 1)  // instance-major rows: one contiguous copy
      HIP_TRY(hipMemcpy(hw.data(), b->d_state + word_offset(dp, (u32)inst, 0), (size_t)4 * dp.total_words, hipMemcpyDeviceToHost));
    else
      HIP_TRY(hipMemcpy2D(hw.data(), sizeof(u32), b->d_state + word_offset(dp, (u32)inst, 0), (size_t)4 * dp.tw, sizeof(u32), dp.total_words,
                          hipMemcpyDeviceToHost));
    std::string err;
    b->sn_generation = ~0ull;
    b->sn_image.clear();
    int rc = build_node_image(dp, hw.data(), node, b->weights.data(), b->cfg.delta, b->cfg.gamma, b->cfg.lambda, b->cfg.target_commit_interval, b->sn_image, err);
    if (rc != 0) { g_err = err; return LBFT_ERR_UNSUPPORTED; }
    b->sn_generation = b->generation; b->sn_inst = inst; b->sn_node = node;
  }
  const std::vector<uint8_t>& image = b->sn_image;
  *len = image.size();
  if (buf && cap < image.size()) { g_err = "save_node: the buffer is smaller than the image (*len holds the size needed)"; return LBFT_ERR_INVALID; }
  if (buf) memcpy(buf, image.data(), image.size());
  return LBFT_OK;
}

// ---- ConsensusNode::load_node (librabft-v2/src/node.rs:211-231): a bincode NodeState image into one device-resident node -------------
// The inverse of lbft_batch_save_node (lbft_save_node.h load_node_image): the instance's rows are copied to the host, the image's
// records are found by hash among the instance's block pool, the node's rows (record store, pacemaker, voting constraints, tracker,
// retired stores) are rewritten and copied back.  Like save_node a read-back-sized format conversion, not simulation.
int lbft_batch_load_node(lbft_batch* b, size_t inst, uint32_t node, const void* image, size_t len, int64_t node_time) {
  if (!b || !image || inst >= b->m || node >= b->p.n) return LBFT_ERR_INVALID;
  if (!b->ran && !b->manual && !b->started) { g_err = "run the batch (or start a node-level session) first"; return LBFT_ERR_STATE; }
  HIP_TRY(hipSetDevice(b->device));
  const Params& dp = b->p;
  std::lock_guard<std::mutex> sn_lock(b->sn_mutex);
  std::vector<u32> hw(dp.total_words);
  u32* d_rows = b->d_state + word_offset(dp, (u32)inst, 0);
  HIP_TRY(hipStreamSynchronize(b->stream));
  if (dp.tw == 1) HIP_TRY(hipMemcpy(hw.data(), d_rows, (size_t)4 * dp.total_words, hipMemcpyDeviceToHost));
  else HIP_TRY(hipMemcpy2D(hw.data(), sizeof(u32), d_rows, (size_t)4 * dp.tw, sizeof(u32), dp.total_words, hipMemcpyDeviceToHost));
  std::string err;
  int rc = load_node_image(dp, hw.data(), node, b->weights.data(), b->cfg.delta, b->cfg.gamma, b->cfg.lambda, b->cfg.target_commit_interval,
                           static_cast<const uint8_t*>(image), len, node_time, err);
  if (rc != 0) { g_err = err; return rc == -4 ? LBFT_ERR_STATE : rc == -3 ? LBFT_ERR_UNSUPPORTED : LBFT_ERR_INVALID; }
  if (dp.tw == 1) HIP_TRY(hipMemcpy(d_rows, hw.data(), (size_t)4 * dp.total_words, hipMemcpyHostToDevice));
  else HIP_TRY(hipMemcpy2D(d_rows, (size_t)4 * dp.tw, hw.data(), sizeof(u32), sizeof(u32), dp.total_words, hipMemcpyHostToDevice));
  b->generation++;  // (a cached save_node image of this batch is stale now)
  b->sn_generation = ~0ull;
  return LBFT_OK;
}

int lbft_batch_counters(const lbft_batch* b, lbft_counters* out) {
  if (!b || !out) return LBFT_ERR_INVALID;
  if (!b->ran) { g_err = "run the batch first"; return LBFT_ERR_STATE; }
  *out = b->counters;
  return LBFT_OK;
}

// ---- the run's ONE collective, natively: every rank contributes its 14 counter words, RCCL over xGMI ----
// Sums and high-water marks need different reduction operators, which one ncclAllReduce call cannot mix: the collective is ONE
// ncclAllGather of 14 words per rank, reduced locally (exactly what the Python host does through torch.distributed:
// librabft_simulator_amd/distributed.py gather_rows).  librccl is not linked: it is loaded when the first aggregation is asked for (a
// single-GPU user never maps it).
namespace {
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int /*ncclDataType_t*/, void* /*ncclComm_t*/, hipStream_t);
typedef int (*nccl_commcount_fn)(void* /*ncclComm_t*/, int*);
nccl_allgather_fn g_nccl_allgather = nullptr;
nccl_commcount_fn g_nccl_commcount = nullptr;
const int kNcclUint64 = 5;  // rccl.h: ncclUint64 = 5
const size_t kCounterWords = 14;
}
int lbft_batch_counters_allgather_reduce(lbft_batch* b, void* nccl_comm, lbft_counters* out) {
  if (!b || !nccl_comm || !out) return LBFT_ERR_INVALID;
  if (!b->ran) { g_err = "run the batch first"; return LBFT_ERR_STATE; }
  if (!g_nccl_allgather) {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { g_err = std::string("librccl not found: ") + dlerror(); return LBFT_ERR_HIP; }
    g_nccl_commcount = reinterpret_cast<nccl_commcount_fn>(dlsym(h, "ncclCommCount"));
    g_nccl_allgather = reinterpret_cast<nccl_allgather_fn>(dlsym(h, "ncclAllGather"));
    if (!g_nccl_allgather || !g_nccl_commcount) { g_nccl_allgather = nullptr; g_err = "ncclAllGather / ncclCommCount not found in librccl"; return LBFT_ERR_HIP; }
  }
  HIP_TRY(hipSetDevice(b->device));
  int world = 0;
  int rc = g_nccl_commcount(nccl_comm, &world);
  if (rc != 0 || world < 1) { g_err = "ncclCommCount failed with ncclResult_t " + std::to_string(rc); return LBFT_ERR_HIP; }
  const lbft_counters& c = b->counters;
  // [0, 11): summed over the ranks; [11, 14): high-water marks, the largest of any rank
  unsigned long long mine[kCounterWords] = {c.events[0], c.events[1], c.events[2], c.events[3], c.rng_draws, c.rounds, c.commits, c.events_scheduled,
                                            c.faulted_instances, c.timers_folded, c.node_updates, c.max_queue, c.max_snapshots, c.max_blocks};
  const size_t scratch_bytes = b->m * b->cfg.num_nodes * sizeof(u64) + 256;  // (lbft_batch_create)
  const size_t need = ((size_t)world + 1) * sizeof(mine);
  unsigned long long* d = reinterpret_cast<unsigned long long*>(b->d_scratch);
  unsigned long long* d_big = nullptr;
  if (need > scratch_bytes) { HIP_TRY(hipMalloc(&d_big, need)); d = d_big; }
  std::vector<unsigned long long> all((size_t)world * kCounterWords);
  int st = LBFT_OK;
  do {
    hipError_t he = hipMemcpyAsync(d, mine, sizeof(mine), hipMemcpyHostToDevice, b->stream);
    if (he != hipSuccess) { st = hip_fail(he, "hipMemcpyAsync (counters to device)"); break; }
    rc = g_nccl_allgather(d, d + kCounterWords, kCounterWords, kNcclUint64, nccl_comm, b->stream);
    if (rc != 0) { g_err = "ncclAllGather failed with ncclResult_t " + std::to_string(rc); st = LBFT_ERR_HIP; break; }
    he = hipMemcpyAsync(all.data(), d + kCounterWords, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, b->stream);
    if (he != hipSuccess) { st = hip_fail(he, "hipMemcpyAsync (gathered counters to host)"); break; }
    he = hipStreamSynchronize(b->stream);
    if (he != hipSuccess) { st = hip_fail(he, "hipStreamSynchronize (after ncclAllGather)"); break; }
  } while (0);
  if (d_big) hipFree(d_big);
  if (st != LBFT_OK) return st;  // (every failing path above set the error text itself)
  unsigned long long h[kCounterWords] = {0};
  for (int r = 0; r < world; r++)
    for (size_t k = 0; k < kCounterWords; k++) {
      const unsigned long long v = all[(size_t)r * kCounterWords + k];
      if (k < 11) h[k] += v; else h[k] = v > h[k] ? v : h[k];
    }
  *out = c;
  out->events[0] = h[0]; out->events[1] = h[1]; out->events[2] = h[2]; out->events[3] = h[3];
  out->rng_draws = h[4]; out->rounds = h[5]; out->commits = h[6]; out->events_scheduled = h[7]; out->faulted_instances = h[8];
  out->timers_folded = h[9]; out->node_updates = h[10]; out->max_queue = h[11]; out->max_snapshots = h[12]; out->max_blocks = h[13];
  return LBFT_OK;
}
// (the name the entry point had through round 4: it never was an all-reduce -- kept so that existing bindings keep linking)
int lbft_batch_counters_allreduce(lbft_batch* b, void* nccl_comm, lbft_counters* out) { return lbft_batch_counters_allgather_reduce(b, nccl_comm, out); }

int lbft_batch_faults(const lbft_batch* b, uint32_t* out) {
  if (!b || !out) return LBFT_ERR_INVALID;
  if (!b->ran) { g_err = "run the batch first"; return LBFT_ERR_STATE; }
  HIP_TRY(hipSetDevice(b->device));
  lbft_k_gather_inst<<<(u32)((b->m + 255) / 256), 256, 0, b->stream>>>(b->p, b->d_state, I_FAULT, b->d_scratch);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out, b->d_scratch, b->m * sizeof(u32), hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return LBFT_OK;
}

void lbft_batch_destroy(lbft_batch* b) { free_batch(b); }

void* lbft_batch_stream(const lbft_batch* b) { return b ? (void*)b->stream : nullptr; }
int lbft_batch_last_run_ms(const lbft_batch* b, float* init_ms, float* run_ms) {
  if (!b) return LBFT_ERR_INVALID;
  if (!b->ran) return LBFT_ERR_STATE;
  if (init_ms) *init_ms = b->init_ms;
  if (run_ms) *run_ms = b->run_ms;
  return LBFT_OK;
}
size_t lbft_batch_device_bytes(const lbft_batch* b) {
  if (!b) return 0;
  return b->state_bytes + b->table_bytes + b->m * sizeof(u64) + b->m * b->p.n * (sizeof(u64) * 2);
}

// ---- stand-alone device checks ----
static int tiny_setup(int device, const lbft_config* cfg, lbft_batch** out) {
  static const u64 one_seed = 0;
  return lbft_batch_create(cfg, &one_seed, 1, device, out);
}

int lbft_device_leaders(int device, const uint64_t* voting_rights, uint32_t num_nodes, uint8_t* out, uint32_t n_rounds) {
  if (!out || num_nodes == 0) return LBFT_ERR_INVALID;
  lbft_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.num_nodes = num_nodes; cfg.mean = 10; cfg.variance = 4; cfg.commands_per_epoch = 1; cfg.gamma = 2; cfg.lambda = 0.5;
  cfg.delta = 20; cfg.target_commit_interval = 1; cfg.voting_rights = voting_rights;
  lbft_batch* b = nullptr;
  int rc = tiny_setup(device, &cfg, &b);
  if (rc != LBFT_OK) return rc;
  u8* d = nullptr;
  hipError_t e = hipMalloc(&d, n_rounds);
  if (e == hipSuccess) {
    Params p = b->p;
    lbft_k_fill_leaders<<<(n_rounds + 255) / 256, 256, 0, b->stream>>>(p, d, n_rounds);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
  if (e == hipSuccess) e = hipMemcpy(out, d, n_rounds, hipMemcpyDeviceToHost);
  hipFree(d);
  free_batch(b);
  return e == hipSuccess ? LBFT_OK : hip_fail(e, "lbft_device_leaders");
}

int lbft_device_sample_delays(int device, const lbft_config* cfg, uint64_t seed, int64_t* out, size_t n) {
  if (!out || n > 0xffffffffu) return LBFT_ERR_INVALID;
  lbft_batch* b = nullptr;
  int rc = tiny_setup(device, cfg, &b);
  if (rc != LBFT_OK) return rc;
  i64* d = nullptr;
  hipError_t e = hipMalloc(&d, n * sizeof(i64));
  if (e == hipSuccess) {
    lbft_k_sample_delays<<<1, 64, 0, b->stream>>>(b->p, seed, d, (u32)n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
  if (e == hipSuccess) e = hipMemcpy(out, d, n * sizeof(i64), hipMemcpyDeviceToHost);
  hipFree(d);
  free_batch(b);
  return e == hipSuccess ? LBFT_OK : hip_fail(e, "lbft_device_sample_delays");
}

int lbft_device_exp_log(int device, const double* x, double* exp_out, double* log_out, size_t n) {
  if (!x || !exp_out || !log_out || n > 0xffffffffu) return LBFT_ERR_INVALID;
  HIP_TRY(hipSetDevice(device));
  u64* d_et = nullptr;
  double *dx = nullptr, *de = nullptr, *dl = nullptr;
  hipError_t e = hipMalloc(&d_et, sizeof(H_ET));
  if (e == hipSuccess) e = hipMemcpy(d_et, H_ET, sizeof(H_ET), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&dx, n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&de, n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&dl, n * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(dx, x, n * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    lbft_k_exp_log<<<(u32)((n + 255) / 256), 256>>>(d_et, dx, de, dl, (u32)n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(exp_out, de, n * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(log_out, dl, n * sizeof(double), hipMemcpyDeviceToHost);
  hipFree(d_et); hipFree(dx); hipFree(de); hipFree(dl);
  return e == hipSuccess ? LBFT_OK : hip_fail(e, "lbft_device_exp_log");
}

}  // extern "C"
